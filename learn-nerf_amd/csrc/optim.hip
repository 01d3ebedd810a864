// optim.hip — fused Adam over a flat fp32 buffer and squared-norm reduction.
// Reference: optax.adam as constructed at learn_nerf/train.py:59 and tree_norm (train.py:92-97).
#include "common.h"

namespace lnrf {

// NORMS: also accumulate sum g^2 (the gradient as passed in, before grad_scale) and sum p^2 (parameters BEFORE the
// update, train.py:99-104) into sq_norms[0..1] — the step then needs no separate norm launches.
template <bool NORMS>
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                            float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                            float b1, float b2, float eps, float inv_bc1, float inv_bc2,
                            float grad_scale, float* __restrict__ sq_norms) {
  float sg = 0.0f, sp = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    if (NORMS) {
      sg += gg * gg;
      sp += pp * pp;
    }
    gg *= grad_scale;
    mm = b1 * mm + (1.0f - b1) * gg;
    vv = b2 * vv + (1.0f - b2) * gg * gg;
    const float mh = mm * inv_bc1;
    const float vh = vv * inv_bc2;
    pp -= lr * mh / (sqrtf(vh) + eps);
  };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
    upd(pp.x, gg.x, mm.x, vv.x);
    upd(pp.y, gg.y, mm.y, vv.y);
    upd(pp.z, gg.z, mm.z, vv.z);
    upd(pp.w, gg.w, mm.w, vv.w);
    p4[i] = pp;
    m4[i] = mm;
    v4[i] = vv;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    upd(p[i], g[i], m[i], v[i]);
  if (NORMS) {
    __shared__ float s_part[2][16];
    sg = wave_sum(sg);
    sp = wave_sum(sp);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
      s_part[0][wave] = sg;
      s_part[1][wave] = sp;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      float t = 0.0f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_part[threadIdx.x][w];
      atomicAdd(sq_norms + threadIdx.x, t);
    }
  }
}

// Logging scalars of one step (train.py:141-144, 99-104) in one tiny launch instead of five elementwise ones:
// sums = [sum sq err coarse, sum sq err fine, sum g^2, sum p^2] -> out = [coarse loss, fine loss, grad_norm,
// param_norm]; with `clear` the sums are zeroed for the next step.
__global__ void step_log_kernel(float* __restrict__ sums, float inv_count, float grad_scale, int clear,
                                float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const float a = sums[0], b = sums[1], c = sums[2], d = sums[3];
    out[0] = a * inv_count;
    out[1] = b * inv_count;
    out[2] = sqrtf(c) * grad_scale;
    out[3] = sqrtf(d);
    if (clear) sums[0] = sums[1] = sums[2] = sums[3] = 0.0f;
  }
}

__global__ void sq_norm_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float s_part[16];
  float acc = 0.0f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float t = x[i];
    acc += t * t;
  }
  acc = wave_sum(acc);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += s_part[w];
    atomicAdd(out, s);
  }
}

}  // namespace lnrf

using namespace lnrf;

extern "C" int lnrf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                              float b1, float b2, float eps, int32_t step, float grad_scale,
                              lnrf_stream_t stream) {
  LNRF_CHECK_ARG(p && g && m && v, "null pointer");
  LNRF_CHECK_ARG(n >= 0 && step >= 1, "bad n/step");
  LNRF_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                 "buffers must be 16-byte aligned");
  if (n == 0) return LNRF_OK;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v,
                     n, lr, b1, b2, eps, (float)(1.0 / bc1), (float)(1.0 / bc2), grad_scale, (float*)nullptr);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_adam_step_norms(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                    float b1, float b2, float eps, int32_t step, float grad_scale,
                                    float* sq_norms, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(p && g && m && v && sq_norms, "null pointer");
  LNRF_CHECK_ARG(n >= 0 && step >= 1, "bad n/step");
  LNRF_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                 "buffers must be 16-byte aligned");
  if (n == 0) return LNRF_OK;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256) blocks = 256;  // every workgroup ends with two atomics on the same two words (~12 ns each, serialised)
  hipLaunchKernelGGL(adam_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, g, m, v,
                     n, lr, b1, b2, eps, (float)(1.0 / bc1), (float)(1.0 / bc2), grad_scale, sq_norms);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_step_log(float* sums, float inv_count, float grad_scale, int32_t clear, float* out,
                             lnrf_stream_t stream) {
  LNRF_CHECK_ARG(sums && out, "null pointer");
  hipLaunchKernelGGL(step_log_kernel, dim3(1), dim3(64), 0, as_stream(stream), sums, inv_count, grad_scale, (int)clear,
                     out);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_sq_norm(const float* x, int64_t n, float* out, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(x && out, "null pointer");
  LNRF_CHECK_ARG(n >= 0, "bad n");
  if (n == 0) return LNRF_OK;
  int64_t blocks = (n + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(sq_norm_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, n, out);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}
