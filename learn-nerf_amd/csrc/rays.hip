// rays.hip — per-ray fp32 kernels of the NeRF renderer (one wave64 per ray where a scan is
// needed).  Reference: learn_nerf/render.py (file:line cited per kernel).
#include "common.h"
#include "philox.h"

namespace lnrf {

// ---------------------------------------------------------------------------------------
// ray_t_range (render.py:346-389), computed per ray.
struct TRange {
  float t_min, t_max;
  bool mask;
};

__device__ __forceinline__ TRange ray_t_range(const float* __restrict__ ray, const float bmin[3],
                                              const float bmax[3], float min_t_range, float eps) {
  float lo = -INFINITY, hi = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float o = ray[a], d = ray[3 + a];
    const float den = d + eps;                 // render.py:371 (epsilon added, not sign-matched)
    const float t0 = (bmin[a] - o) / den;      // render.py:370-371
    const float t1 = (bmax[a] - o) / den;
    lo = fmaxf(lo, fminf(t0, t1));             // render.py:374-383
    hi = fminf(hi, fmaxf(t0, t1));             // render.py:384
  }
  const float min_t = fmaxf(0.0f, lo);         // render.py:383
  const float max_t = hi;
  const float max_c = fmaxf(max_t, min_t + min_t_range);  // render.py:385
  TRange r;
  r.mask = min_t < max_t;                      // render.py:388
  r.t_min = r.mask ? min_t : 0.0f;             // render.py:389 null_range
  r.t_max = r.mask ? max_c : min_t_range;
  return r;
}

struct BBox {
  float mn[3], mx[3];
};

__global__ void ray_aabb_stratified_kernel(const float* __restrict__ rays, int64_t ray_stride,
                                           int64_t n_rays, BBox bb, float min_t_range, float eps,
                                           int count, const float* __restrict__ u, uint64_t seed,
                                           uint32_t stream_id, int64_t ray_offset,
                                           float* __restrict__ t_min, float* __restrict__ t_max,
                                           uint8_t* __restrict__ mask, float* __restrict__ ts) {
  const int per = count > 0 ? count : 1;
  const int64_t total = n_rays * per;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = e / per;
    const int i = (int)(e - n * per);
    const TRange r = ray_t_range(rays + n * ray_stride, bb.mn, bb.mx, min_t_range, eps);
    if (i == 0) {
      if (t_min) t_min[n] = r.t_min;
      if (t_max) t_max[n] = r.t_max;
      if (mask) mask[n] = r.mask ? 1 : 0;
    }
    if (count > 0) {
      const float uu = u ? u[e] : philox_uniform(seed, stream_id, (uint64_t)(ray_offset + n) * count + i);
      const float bin = (r.t_max - r.t_min) / (float)count;   // render.py:138
      const float start = (float)i * bin + r.t_min;           // render.py:139-141
      ts[e] = start + uu * bin;                               // render.py:142-143
    }
  }
}

__global__ void stratified_kernel(const float* __restrict__ t_min, const float* __restrict__ t_max,
                                  int64_t n_rays, int count, const float* __restrict__ u,
                                  uint64_t seed, uint32_t stream_id, int64_t ray_offset,
                                  float* __restrict__ ts) {
  const int64_t total = n_rays * count;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = e / count;
    const int i = (int)(e - n * count);
    const float uu = u ? u[e] : philox_uniform(seed, stream_id, (uint64_t)(ray_offset + n) * count + i);
    const float lo = t_min[n];
    const float bin = (t_max[n] - lo) / (float)count;
    ts[e] = ((float)i * bin + lo) + uu * bin;
  }
}

__global__ void ray_points_kernel(const float* __restrict__ rays, int64_t ray_stride,
                                  const float* __restrict__ ts, int64_t n_rays, int t,
                                  float* __restrict__ points, float* __restrict__ dirs) {
  const int64_t total = n_rays * t;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = e / t;
    const float* r = rays + n * ray_stride;
    const float tt = ts[e];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (points) points[e * 3 + a] = r[a] + r[3 + a] * tt;   // render.py:153
      if (dirs) dirs[e * 3 + a] = r[3 + a];                   // render.py:319
    }
  }
}

// ---------------------------------------------------------------------------------------
// Quadrature helpers (render.py:259-287). One wave per ray, samples processed in chunks of
// 64 lanes with the running optical depth carried between chunks.
struct Quad {
  float a;      // density * delta                         (render.py:271)
  float incl;   // inclusive cumsum of a up to this sample (render.py:275)
  float prev;   // exclusive cumsum                        (render.py:276-278)
  float t;      // sample position
  float end;    // bin end                                 (render.py:263-265)
};

__device__ __forceinline__ Quad quad_chunk(const float* __restrict__ ts_row,
                                           const float* __restrict__ dens_row, int T, int i,
                                           int lane, float tmin, float tmax, float carry) {
  Quad q;
  const bool valid = i < T;
  const int ic = valid ? i : T - 1;
  const float t_i = ts_row[ic];
  const float t_p = ic > 0 ? ts_row[ic - 1] : 0.0f;
  const float t_n = ic < T - 1 ? ts_row[ic + 1] : 0.0f;
  const float start = ic == 0 ? tmin : (t_i + t_p) / 2.0f;      // render.py:259-261
  const float end = ic == T - 1 ? tmax : (t_n + t_i) / 2.0f;    // render.py:263-265
  const float delta = end - start;                              // render.py:268
  q.a = valid ? dens_row[ic] * delta : 0.0f;
  q.incl = wave_incl_scan(q.a, lane) + carry;
  const float up = __shfl_up(q.incl, 1, 64);
  q.prev = lane == 0 ? carry : up;
  q.t = t_i;
  q.end = end;
  return q;
}

// termination_probs (render.py:270-287) -> probs[N, T+1]
__global__ void termination_probs_kernel(const float* __restrict__ ts, const float* __restrict__ t_min,
                                         const float* __restrict__ t_max,
                                         const float* __restrict__ density, int64_t n_rays, int T,
                                         float* __restrict__ probs) {
  const int lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (n >= n_rays) return;
  const float* ts_row = ts + n * T;
  const float* de_row = density + n * T;
  const float tmin = t_min[n], tmax = t_max[n];
  float carry = 0.0f;
  for (int base = 0; base < T; base += 64) {
    const int i = base + lane;
    const Quad q = quad_chunk(ts_row, de_row, T, i, lane, tmin, tmax, carry);
    if (i < T) probs[n * (T + 1) + i] = expf(-q.prev) * (1.0f - expf(-q.a));  // render.py:279-287
    carry = __shfl(q.incl, 63, 64);
  }
  if (lane == 0) probs[n * (T + 1) + T] = expf(-carry);
}

constexpr int kMaxAux = 4;

// render_rays / render_alpha / average_aux_losses (render.py:155-209) + coords (render.py:331)
__global__ void composite_fwd_kernel(const float* __restrict__ rays, int64_t ray_stride,
                                     const float* __restrict__ ts, const float* __restrict__ t_min,
                                     const float* __restrict__ t_max,
                                     const uint8_t* __restrict__ mask,
                                     const float* __restrict__ density,
                                     const float* __restrict__ rgb, const float* __restrict__ aux,
                                     int n_aux, const float* __restrict__ background,
                                     int64_t n_rays, int T, float* __restrict__ outputs,
                                     float* __restrict__ alphas, float* __restrict__ coords,
                                     float* __restrict__ aux_sum, const float* __restrict__ targets,
                                     int64_t target_stride, float* __restrict__ sq_err) {
  __shared__ float s_err[16];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  const bool live = n < n_rays;
  float err = 0.0f;
  if (live) {
    const float* ts_row = ts + n * T;
    const float* de_row = density + n * T;
    const float tmin = t_min[n], tmax = t_max[n];
    const bool m = mask[n] != 0;
    float o[3] = {0, 0, 0}, d[3] = {0, 0, 0};
    if (rays) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        o[a] = rays[n * ray_stride + a];
        d[a] = rays[n * ray_stride + 3 + a];
      }
    }
    float acc_c[3] = {0, 0, 0}, acc_x[3] = {0, 0, 0}, acc_a[kMaxAux] = {0, 0, 0, 0};
    float carry = 0.0f;
    for (int base = 0; base < T; base += 64) {
      const int i = base + lane;
      const Quad q = quad_chunk(ts_row, de_row, T, i, lane, tmin, tmax, carry);
      if (i < T) {
        const float p = expf(-q.prev) * (1.0f - expf(-q.a));
        const int64_t e = n * T + i;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          acc_c[c] += p * rgb[e * 3 + c];
          acc_x[c] += p * (o[c] + d[c] * q.t);
        }
        for (int k = 0; k < n_aux; ++k) acc_a[k] += p * aux[e * n_aux + k];
      }
      carry = __shfl(q.incl, 63, 64);
    }
    const float p_bg = expf(-carry);  // probs[:, T]
    float out[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float bg = background[c];
      const float s = wave_sum(acc_c[c]) + p_bg * bg;            // render.py:170-176
      out[c] = m ? s : bg;
      const float x = wave_sum(acc_x[c]);
      if (lane == 0) {
        if (outputs) outputs[n * 3 + c] = out[c];
        if (coords) coords[n * 3 + c] = m ? x : 0.0f;            // render.py:331
      }
    }
    for (int k = 0; k < n_aux; ++k) {
      const float s = wave_sum(acc_a[k]);
      if (lane == 0 && aux_sum) aux_sum[n * n_aux + k] = m ? s : 0.0f;  // render.py:205-208
    }
    if (lane == 0 && alphas) alphas[n] = m ? 1.0f - p_bg : 0.0f;        // render.py:189-190
    if (targets && lane == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float df = out[c] - targets[n * target_stride + c];
        err += df * df;
      }
    }
  }
  if (sq_err && targets) {  // block-level partial, one atomic per block (train.py:141-142)
    if (lane == 0) s_err[wave] = err;
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.0f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += s_err[w];
      atomicAdd(sq_err, s);
    }
  }
}

// Backward of the compositing integral.  With F = sum_{i<=T} p_i w_i (w_i = <g, c_i>, w_T = <g, bg>):
//   dF/da_k = S_{k+1} w_k - sum_{i>k} p_i w_i,   S_{k+1} = exp(-A_k);   d sigma_k = delta_k dF/da_k.
__global__ void composite_bwd_kernel(const float* __restrict__ ts, const float* __restrict__ t_min,
                                     const float* __restrict__ t_max,
                                     const uint8_t* __restrict__ mask,
                                     const float* __restrict__ density,
                                     const float* __restrict__ rgb, const float* __restrict__ aux,
                                     int n_aux, const float* __restrict__ background,
                                     int64_t n_rays, int T, const float* __restrict__ g_out,
                                     const float* __restrict__ outputs,
                                     const float* __restrict__ targets, int64_t target_stride,
                                     float out_scale, float gw0, float gw1, float gw2, float gw3,
                                     float* __restrict__ g_density, float* __restrict__ g_rgb,
                                     float* __restrict__ g_aux, float* __restrict__ g_background,
                                     float* __restrict__ bg_parts) {
  __shared__ float s_bg[16][3];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  const bool live = n < n_rays;
  const float gw[kMaxAux] = {gw0, gw1, gw2, gw3};
  float gbg[3] = {0, 0, 0};
  if (live) {
    const float* ts_row = ts + n * T;
    const float* de_row = density + n * T;
    const float tmin = t_min[n], tmax = t_max[n];
    const bool m = mask[n] != 0;
    float g[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
      g[c] = g_out ? g_out[n * 3 + c]
                   : out_scale * (outputs[n * 3 + c] - targets[n * target_stride + c]);
    if (!m) {
      // outputs == background, no dependence on the samples (render.py:174-176)
      for (int i = lane; i < T; i += 64) {
        const int64_t e = n * T + i;
        g_density[e] = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; ++c) g_rgb[e * 3 + c] = 0.0f;
        for (int k = 0; k < n_aux; ++k) g_aux[e * n_aux + k] = 0.0f;
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) gbg[c] = g[c];
    } else {
      // pass 1: total F
      float carry = 0.0f, f_part = 0.0f;
      for (int base = 0; base < T; base += 64) {
        const int i = base + lane;
        const Quad q = quad_chunk(ts_row, de_row, T, i, lane, tmin, tmax, carry);
        if (i < T) {
          const float p = expf(-q.prev) * (1.0f - expf(-q.a));
          const int64_t e = n * T + i;
          float w = 0.0f;
#pragma unroll
          for (int c = 0; c < 3; ++c) w += g[c] * rgb[e * 3 + c];
          for (int k = 0; k < n_aux; ++k) w += gw[k] * aux[e * n_aux + k];
          f_part += p * w;
        }
        carry = __shfl(q.incl, 63, 64);
      }
      const float p_bg = expf(-carry);
      float w_bg = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) w_bg += g[c] * background[c];
      const float F = wave_sum(f_part) + p_bg * w_bg;
      // pass 2: gradients
      carry = 0.0f;
      float pw_carry = 0.0f;
      for (int base = 0; base < T; base += 64) {
        const int i = base + lane;
        const Quad q = quad_chunk(ts_row, de_row, T, i, lane, tmin, tmax, carry);
        const bool valid = i < T;
        const int64_t e = n * T + (valid ? i : T - 1);
        float p = 0.0f, w = 0.0f;
        if (valid) {
          p = expf(-q.prev) * (1.0f - expf(-q.a));
#pragma unroll
          for (int c = 0; c < 3; ++c) w += g[c] * rgb[e * 3 + c];
          for (int k = 0; k < n_aux; ++k) w += gw[k] * aux[e * n_aux + k];
        }
        const float pw_incl = wave_incl_scan(p * w, lane) + pw_carry;
        if (valid) {
          const float s_next = expf(-q.incl);
          const float rest = F - pw_incl;                  // sum_{i>k} p_i w_i incl. background
          const float t_p = i > 0 ? ts_row[i - 1] : 0.0f;
          const float start = i == 0 ? tmin : (q.t + t_p) / 2.0f;
          const float dl = q.end - start;                  // delta_k (density may be 0: recompute)
          g_density[e] = dl * (s_next * w - rest);
#pragma unroll
          for (int c = 0; c < 3; ++c) g_rgb[e * 3 + c] = p * g[c];
          for (int k = 0; k < n_aux; ++k) g_aux[e * n_aux + k] = p * gw[k];
        }
        carry = __shfl(q.incl, 63, 64);
        pw_carry = __shfl(pw_incl, 63, 64);
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) gbg[c] = p_bg * g[c];
    }
  }
  if (g_background) {
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) s_bg[wave][c] = gbg[c];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
      float s = 0.0f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += s_bg[w][threadIdx.x];
      // bg_parts: the block's partial sum goes to its own row and composite_bg_fold_kernel adds the rows in a fixed order
      // (bit-reproducible); without it the blocks meet in fp32 atomics
      if (bg_parts) bg_parts[(int64_t)blockIdx.x * 4 + threadIdx.x] = s;
      else atomicAdd(g_background + threadIdx.x, s);
    }
  }
}

// g_background[c] += sum over blocks of parts[block][c], c < 3: thread t adds the rows t, t + 256, ... in order, then the
// 256 partial sums meet in a fixed binary tree
__global__ __launch_bounds__(256) void composite_bg_fold_kernel(const float* __restrict__ parts, int n_blocks,
                                                                float* __restrict__ g_background) {
  __shared__ float red[256][3];
  float s[3] = {0.0f, 0.0f, 0.0f};
  for (int b = threadIdx.x; b < n_blocks; b += 256) {
    const float4 v = *reinterpret_cast<const float4*>(parts + (int64_t)b * 4);
    s[0] += v.x; s[1] += v.y; s[2] += v.z;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) red[threadIdx.x][c] = s[c];
  __syncthreads();
  for (int half = 128; half > 0; half >>= 1) {
    if ((int)threadIdx.x < half) {
#pragma unroll
      for (int c = 0; c < 3; ++c) red[threadIdx.x][c] += red[threadIdx.x + half][c];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) g_background[threadIdx.x] += red[0][threadIdx.x];
}

// fine_sampling (render.py:211-257). One wave per ray; LDS per wave: xs[tc+1], ys[tc+1],
// comb[tc+tf].  The final jnp.sort is exact for any input (merge ranks when both lists are in order, rank count otherwise).
__global__ void fine_sample_kernel(const float* __restrict__ ts_c, const float* __restrict__ t_min,
                                   const float* __restrict__ t_max,
                                   const float* __restrict__ density_c, int64_t n_rays, int tc,
                                   int tf, float eps, int combine, const float* __restrict__ u,
                                   uint64_t seed, uint32_t stream_id, int64_t ray_offset,
                                   float* __restrict__ ts_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int per_wave = 2 * (tc + 1) + tc + tf;
  float* xs = smem + (size_t)wave * per_wave;
  float* ys = xs + (tc + 1);
  float* comb = ys + (tc + 1);
  int64_t n = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  const bool live = n < n_rays;
  if (!live) n = n_rays - 1;  // keep the wave in the barriers; stores are predicated
  const float* ts_row = ts_c + n * tc;
  const float* de_row = density_c + n * tc;
  const float tmin = t_min[n], tmax = t_max[n];

  // CDF over coarse bins: w = probs + eps (render.py:232), xs = [0, cumsum(w)] / sum (render.py:235-237)
  float carry = 0.0f, wcarry = 0.0f;
  for (int base = 0; base < tc; base += 64) {
    const int i = base + lane;
    const Quad q = quad_chunk(ts_row, de_row, tc, i, lane, tmin, tmax, carry);
    const float p = expf(-q.prev) * (1.0f - expf(-q.a));
    const float w = i < tc ? p + eps : 0.0f;
    const float wincl = wave_incl_scan(w, lane) + wcarry;
    if (i < tc) {
      xs[i + 1] = wincl;
      ys[i + 1] = q.end;             // render.py:238-241
      comb[i] = q.t;
    }
    carry = __shfl(q.incl, 63, 64);
    wcarry = __shfl(wincl, 63, 64);
  }
  if (lane == 0) {
    xs[0] = 0.0f;
    ys[0] = tmin;
  }
  __syncthreads();
  const float total = xs[tc];
  __syncthreads();
  for (int i = lane; i <= tc; i += 64) xs[i] = xs[i] / total;  // render.py:237
  __syncthreads();

  // inverse CDF at stratified points (render.py:244-251)
  const float bin = (1.0f - 0.0f) / (float)tf;
  for (int j = lane; j < tf; j += 64) {
    const float uu = u ? u[n * tf + j] : philox_uniform(seed, stream_id, (uint64_t)(ray_offset + n) * tf + j);
    const float x = ((float)j * bin + 0.0f) + uu * bin;
    // i = clip(searchsorted(xs, x, side="right"), 1, tc)
    int lo = 0, hi = tc + 1;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (xs[mid] <= x) lo = mid + 1; else hi = mid;
    }
    int idx = lo < 1 ? 1 : (lo > tc ? tc : lo);
    const float x0 = xs[idx - 1], x1 = xs[idx], y0 = ys[idx - 1], y1 = ys[idx];
    const float dx = x1 - x0;
    float y = fabsf(dx) <= 1.17549435e-38f ? y0 : y0 + (x - x0) / dx * (y1 - y0);
    if (x < xs[0]) y = ys[0];
    if (x > xs[tc]) y = ys[tc];
    if (combine) comb[tc + j] = y;
    else if (live) ts_out[n * tf + j] = y;
  }
  if (!combine) return;
  __syncthreads();
  // exact stable sort of comb[0..tc+tf) (render.py:253-255).  The coarse ts (stratified) and the new ts (a monotone
  // map of increasing CDF arguments) are each already in order, so an element's rank is its own index plus a binary
  // search in the other list (ties: the coarse element first, as in the stable sort); the wave checks the two orders
  // and falls back to the O(T^2) rank count for anything else (unsorted caller-supplied ts, NaN).
  const int tot = tc + tf;
  bool in_order = true;
  for (int e = lane; e + 1 < tot; e += 64)
    if (e + 1 != tc) in_order &= comb[e] <= comb[e + 1];
  if (__all(in_order)) {
    for (int e = lane; e < tot; e += 64) {
      const float x = comb[e];
      const bool first = e < tc;
      // first list: count of new ts < x (lower bound); second list: count of coarse ts <= x (upper bound)
      const float* other = first ? comb + tc : comb;
      int lo = 0, hi = first ? tf : tc;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float y = other[mid];
        if (first ? (y < x) : (y <= x)) lo = mid + 1; else hi = mid;
      }
      const int rank = (first ? e : e - tc) + lo;
      if (live) ts_out[n * tot + rank] = x;
    }
    return;
  }
  for (int e = lane; e < tot; e += 64) {
    const float x = comb[e];
    int rank = 0;
    for (int k = 0; k < tot; ++k) {
      const float y = comb[k];
      rank += (y < x || (y == x && k < e)) ? 1 : 0;
    }
    if (live) ts_out[n * tot + rank] = x;
  }
}

// RaySamples.starts / ends (render.py:259-265)
__global__ void bin_edges_kernel(const float* __restrict__ ts, const float* __restrict__ t_min,
                                 const float* __restrict__ t_max, int64_t n_rays, int T,
                                 float* __restrict__ starts, float* __restrict__ ends) {
  const int64_t total = n_rays * T;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = e / T;
    const int i = (int)(e - n * T);
    const float t_i = ts[e];
    if (starts) starts[e] = i == 0 ? t_min[n] : (t_i + ts[e - 1]) / 2.0f;
    if (ends) ends[e] = i == T - 1 ? t_max[n] : (ts[e + 1] + t_i) / 2.0f;
  }
}

// CameraView.bare_rays (dataset.py:52-78): pixel (px, py) in raster order ->
// direction = normalize(z + tan(x_fov/2) * lx[px] * x_axis + tan(y_fov/2) * ly[py] * y_axis),
// lx = linspace(-1, 1, W), ly = linspace(-1, 1, H) (end points inclusive); origin = camera origin.
struct Camera {
  float origin[3], x_axis[3], y_axis[3], z_axis[3];
  float tan_x, tan_y;
};
__global__ void camera_rays_kernel(Camera cam, int width, int height, float* __restrict__ rays) {
  const int64_t total = (int64_t)width * height;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int py = (int)(e / width), px = (int)(e - (int64_t)py * width);
    const float lx = width > 1 ? -1.0f + 2.0f * (float)px / (float)(width - 1) : -1.0f;
    const float ly = height > 1 ? -1.0f + 2.0f * (float)py / (float)(height - 1) : -1.0f;
    float d[3];
    float nrm = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      d[a] = (cam.tan_x * lx) * cam.x_axis[a] + (cam.tan_y * ly) * cam.y_axis[a] + cam.z_axis[a];
      nrm += d[a] * d[a];
    }
    nrm = sqrtf(nrm);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      rays[e * 6 + a] = cam.origin[a];
      rays[e * 6 + 3 + a] = d[a] / nrm;
    }
  }
}

// out[i, :] = src[idx[i], :]: one thread per output float, so the stores are fully coalesced and the 36-byte source
// rows are read by 9 neighbouring lanes
__global__ void gather_rows_kernel(const float* __restrict__ src, int64_t n_src, int row_floats,
                                   const int32_t* __restrict__ idx, int64_t n, float* __restrict__ out) {
  const int64_t total = n * row_floats;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / row_floats;
    const int f = (int)(e - i * row_floats);
    const int64_t r = idx[i];
    out[e] = (r >= 0 && r < n_src) ? src[r * row_floats + f] : 0.0f;
  }
}

}  // namespace lnrf

using namespace lnrf;

static inline int grid_for(int64_t total, int block, int cap = 2048 * 8) {
  int64_t g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

extern "C" int lnrf_ray_aabb_stratified(const float* rays, int64_t ray_stride, int64_t n_rays,
                                        const float* bbox_min, const float* bbox_max,
                                        float min_t_range, float epsilon, int32_t count,
                                        const float* u, uint64_t seed, uint32_t stream_id,
                                        int64_t ray_offset, float* t_min, float* t_max,
                                        uint8_t* mask, float* ts, lnrf_stream_t stream) {
  if (n_rays == 0) return LNRF_OK;
  LNRF_CHECK_ARG(rays && bbox_min && bbox_max, "null rays/bbox");
  LNRF_CHECK_ARG(n_rays >= 0 && count >= 0 && ray_stride >= 6, "bad sizes");
  LNRF_CHECK_ARG(count == 0 || ts, "ts is NULL with count > 0");
  if (n_rays == 0) return LNRF_OK;
  BBox bb;
  for (int a = 0; a < 3; ++a) {
    bb.mn[a] = bbox_min[a];
    bb.mx[a] = bbox_max[a];
  }
  const int64_t total = n_rays * (count > 0 ? count : 1);
  hipLaunchKernelGGL(ray_aabb_stratified_kernel, dim3(grid_for(total, 256)), dim3(256), 0,
                     as_stream(stream), rays, ray_stride, n_rays, bb, min_t_range, epsilon, count, u,
                     seed, stream_id, ray_offset, t_min, t_max, mask, ts);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_stratified(const float* t_min, const float* t_max, int64_t n_rays,
                               int32_t count, const float* u, uint64_t seed, uint32_t stream_id,
                               int64_t ray_offset, float* ts, lnrf_stream_t stream) {
  if (n_rays == 0 || count == 0) return LNRF_OK;
  LNRF_CHECK_ARG(t_min && t_max, "null t range");
  LNRF_CHECK_ARG(n_rays >= 0 && count >= 0, "bad sizes");
  if (n_rays == 0 || count == 0) return LNRF_OK;
  LNRF_CHECK_ARG(ts, "null ts");
  hipLaunchKernelGGL(stratified_kernel, dim3(grid_for(n_rays * count, 256)), dim3(256), 0,
                     as_stream(stream), t_min, t_max, n_rays, count, u, seed, stream_id, ray_offset, ts);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_ray_points(const float* rays, int64_t ray_stride, const float* ts,
                               int64_t n_rays, int32_t t, float* points, float* dirs,
                               lnrf_stream_t stream) {
  if (n_rays == 0 || t == 0) return LNRF_OK;
  LNRF_CHECK_ARG(rays && ts, "null rays/ts");
  LNRF_CHECK_ARG(n_rays >= 0 && t >= 0 && ray_stride >= 6, "bad sizes");
  if (n_rays == 0 || t == 0) return LNRF_OK;
  hipLaunchKernelGGL(ray_points_kernel, dim3(grid_for(n_rays * t, 256)), dim3(256), 0,
                     as_stream(stream), rays, ray_stride, ts, n_rays, t, points, dirs);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_termination_probs(const float* ts, const float* t_min, const float* t_max,
                                      const float* density, int64_t n_rays, int32_t t,
                                      float* probs, lnrf_stream_t stream) {
  if (n_rays == 0) return LNRF_OK;
  LNRF_CHECK_ARG(ts && t_min && t_max && density && probs, "null pointer");
  LNRF_CHECK_ARG(n_rays >= 0 && t >= 1, "bad sizes");
  if (n_rays == 0) return LNRF_OK;
  const int wpb = 4;
  hipLaunchKernelGGL(termination_probs_kernel, dim3((unsigned)((n_rays + wpb - 1) / wpb)),
                     dim3(wpb * 64), 0, as_stream(stream), ts, t_min, t_max, density, n_rays, t, probs);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_composite_fwd(const float* rays, int64_t ray_stride, const float* ts,
                                  const float* t_min, const float* t_max, const uint8_t* mask,
                                  const float* density, const float* rgb, const float* aux,
                                  int32_t n_aux, const float* background, int64_t n_rays, int32_t t,
                                  float* outputs, float* alphas, float* coords, float* aux_sum,
                                  const float* targets, int64_t target_stride, float* sq_err,
                                  lnrf_stream_t stream) {
  if (n_rays == 0) return LNRF_OK;
  LNRF_CHECK_ARG(ts && t_min && t_max && mask && density && rgb && background, "null pointer");
  LNRF_CHECK_ARG(n_rays >= 0 && t >= 1, "bad sizes");
  LNRF_CHECK_ARG(n_aux >= 0 && n_aux <= kMaxAux && (n_aux == 0 || aux), "bad aux");
  LNRF_CHECK_ARG(!coords || rays, "coords needs rays");
  if (n_rays == 0) return LNRF_OK;
  // with a squared-error sum every workgroup ends in one atomic on the same word (~12 ns each, serialised): 16 rays per
  // workgroup instead of 4 then (256 instead of 1,024 atomics at 4,096 rays)
  const int wpb = (sq_err && targets) ? 16 : 4;
  hipLaunchKernelGGL(composite_fwd_kernel, dim3((unsigned)((n_rays + wpb - 1) / wpb)),
                     dim3(wpb * 64), 0, as_stream(stream), rays, ray_stride, ts, t_min, t_max, mask,
                     density, rgb, aux, n_aux, background, n_rays, t, outputs, alphas, coords,
                     aux_sum, targets, target_stride, sq_err);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

static int composite_bwd_impl(const float* ts, const float* t_min, const float* t_max,
                                  const uint8_t* mask, const float* density, const float* rgb,
                                  const float* aux, int32_t n_aux, const float* background,
                                  int64_t n_rays, int32_t t, const float* g_out,
                                  const float* outputs, const float* targets,
                                  int64_t target_stride, float out_scale, const float* g_aux_w,
                                  float* g_density, float* g_rgb, float* g_aux,
                                  float* g_background, float* bg_parts, lnrf_stream_t stream) {
  if (n_rays == 0) return LNRF_OK;
  LNRF_CHECK_ARG(ts && t_min && t_max && mask && density && rgb && background, "null pointer");
  LNRF_CHECK_ARG(g_density && g_rgb, "null gradient outputs");
  LNRF_CHECK_ARG(g_out || (outputs && targets), "need g_out or outputs+targets");
  LNRF_CHECK_ARG(n_rays >= 0 && t >= 1, "bad sizes");
  LNRF_CHECK_ARG(n_aux >= 0 && n_aux <= kMaxAux && (n_aux == 0 || (aux && g_aux && g_aux_w)), "bad aux");
  if (n_rays == 0) return LNRF_OK;
  float gw[kMaxAux] = {0, 0, 0, 0};
  for (int k = 0; k < n_aux; ++k) gw[k] = g_aux_w[k];
  const int wpb = 4;
  hipLaunchKernelGGL(composite_bwd_kernel, dim3((unsigned)((n_rays + wpb - 1) / wpb)),
                     dim3(wpb * 64), 0, as_stream(stream), ts, t_min, t_max, mask, density, rgb, aux,
                     n_aux, background, n_rays, t, g_out, outputs, targets, target_stride, out_scale,
                     gw[0], gw[1], gw[2], gw[3], g_density, g_rgb, g_aux, g_background,
                     g_background ? bg_parts : nullptr);
  LNRF_LAUNCH_CHECK();
  if (g_background && bg_parts) {
    hipLaunchKernelGGL(composite_bg_fold_kernel, dim3(1), dim3(256), 0, as_stream(stream), bg_parts,
                       (int)((n_rays + wpb - 1) / wpb), g_background);
    LNRF_LAUNCH_CHECK();
  }
  return LNRF_OK;
}

extern "C" int lnrf_composite_bwd(const float* ts, const float* t_min, const float* t_max,
                                  const uint8_t* mask, const float* density, const float* rgb,
                                  const float* aux, int32_t n_aux, const float* background,
                                  int64_t n_rays, int32_t t, const float* g_out,
                                  const float* outputs, const float* targets,
                                  int64_t target_stride, float out_scale, const float* g_aux_w,
                                  float* g_density, float* g_rgb, float* g_aux,
                                  float* g_background, lnrf_stream_t stream) {
  return composite_bwd_impl(ts, t_min, t_max, mask, density, rgb, aux, n_aux, background, n_rays, t, g_out, outputs,
                            targets, target_stride, out_scale, g_aux_w, g_density, g_rgb, g_aux, g_background, nullptr,
                            stream);
}

extern "C" int64_t lnrf_composite_bwd_scratch_bytes(int64_t n_rays) {
  return n_rays < 0 ? -1 : ((n_rays + 3) / 4) * 4 * (int64_t)sizeof(float) + 16;
}

// lnrf_composite_bwd with a fixed summation order for the background gradient (the only sum over rays the call forms):
// per-workgroup partial sums in `scratch`, folded by a second launch.
extern "C" int lnrf_composite_bwd_det(const float* ts, const float* t_min, const float* t_max,
                                      const uint8_t* mask, const float* density, const float* rgb,
                                      const float* aux, int32_t n_aux, const float* background,
                                      int64_t n_rays, int32_t t, const float* g_out,
                                      const float* outputs, const float* targets,
                                      int64_t target_stride, float out_scale, const float* g_aux_w,
                                      float* g_density, float* g_rgb, float* g_aux,
                                      float* g_background, void* scratch, int64_t scratch_bytes, lnrf_stream_t stream) {
  if (n_rays > 0 && g_background)
    LNRF_CHECK_ARG(scratch && scratch_bytes >= lnrf_composite_bwd_scratch_bytes(n_rays) &&
                       (reinterpret_cast<uintptr_t>(scratch) & 15u) == 0, "scratch too small or unaligned");
  return composite_bwd_impl(ts, t_min, t_max, mask, density, rgb, aux, n_aux, background, n_rays, t, g_out, outputs,
                            targets, target_stride, out_scale, g_aux_w, g_density, g_rgb, g_aux, g_background,
                            reinterpret_cast<float*>(scratch), stream);
}

extern "C" int lnrf_fine_sample(const float* ts_c, const float* t_min, const float* t_max,
                                const float* density_c, int64_t n_rays, int32_t tc, int32_t tf,
                                float eps, int32_t combine, const float* u, uint64_t seed,
                                uint32_t stream_id, int64_t ray_offset, float* ts_out,
                                lnrf_stream_t stream) {
  if (n_rays == 0) return LNRF_OK;
  LNRF_CHECK_ARG(ts_c && t_min && t_max && density_c && ts_out, "null pointer");
  LNRF_CHECK_ARG(n_rays >= 0 && tc >= 1 && tf >= 0, "bad sizes");
  if (n_rays == 0) return LNRF_OK;
  if (tf == 0 && !combine) return LNRF_OK;
  const int wpb = 4;
  const size_t lds = (size_t)wpb * (2 * (tc + 1) + tc + tf) * sizeof(float);
  LNRF_CHECK_ARG(lds <= 160 * 1024, "tc+tf too large for LDS");
  hipLaunchKernelGGL(fine_sample_kernel, dim3((unsigned)((n_rays + wpb - 1) / wpb)), dim3(wpb * 64),
                     lds, as_stream(stream), ts_c, t_min, t_max, density_c, n_rays, tc, tf, eps,
                     combine, u, seed, stream_id, ray_offset, ts_out);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_bin_edges(const float* ts, const float* t_min, const float* t_max, int64_t n_rays,
                              int32_t t, float* starts, float* ends, lnrf_stream_t stream) {
  if (n_rays == 0 || t == 0) return LNRF_OK;
  LNRF_CHECK_ARG(ts && t_min && t_max, "null pointer");
  LNRF_CHECK_ARG(n_rays >= 0 && t >= 0, "bad sizes");
  if (n_rays == 0 || t == 0) return LNRF_OK;
  hipLaunchKernelGGL(bin_edges_kernel, dim3(grid_for(n_rays * t, 256)), dim3(256), 0, as_stream(stream), ts,
                     t_min, t_max, n_rays, t, starts, ends);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_gather_rows(const float* src, int64_t n_src, int32_t row_floats, const int32_t* idx, int64_t n,
                                float* out, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(n >= 0 && n_src >= 0 && row_floats >= 1, "bad sizes");
  if (n == 0) return LNRF_OK;
  LNRF_CHECK_ARG(src && idx && out, "null pointer");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n * row_floats, 256)), dim3(256), 0, as_stream(stream), src,
                     n_src, (int)row_floats, idx, n, out);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_camera_rays(const float* origin, const float* x_axis, const float* y_axis, const float* z_axis,
                                float x_fov, float y_fov, int32_t width, int32_t height, float* rays,
                                lnrf_stream_t stream) {
  LNRF_CHECK_ARG(origin && x_axis && y_axis && z_axis && rays, "null pointer");
  LNRF_CHECK_ARG(width >= 1 && height >= 1, "bad image size");
  Camera cam;
  for (int a = 0; a < 3; ++a) {
    cam.origin[a] = origin[a];
    cam.x_axis[a] = x_axis[a];
    cam.y_axis[a] = y_axis[a];
    cam.z_axis[a] = z_axis[a];
  }
  cam.tan_x = tanf(x_fov * 0.5f);
  cam.tan_y = tanf(y_fov * 0.5f);
  hipLaunchKernelGGL(camera_rays_kernel, dim3(grid_for((int64_t)width * height, 256)), dim3(256), 0,
                     as_stream(stream), cam, (int)width, (int)height, rays);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}
