// mfma_shape_bench.hip — v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 in the operand-feeding pattern of the
// fused forward chain (fused_chain.h chain_layer): 8 waves per workgroup (two per SIMD), a 256 -> 256 layer per
// 32-evaluation tile and wave, the weight (A) fragment of every MFMA read from LDS by ds_read_b128 four fragments ahead,
// the activations (B) in registers, a ReLU + bf16 conversion of the accumulators per out tile.
//   S=0  32x32x16: 8 out tiles x 16 k-steps, one 1 KiB A fragment per MFMA
//   S=1  16x16x32: 16 out tiles x 8 k-steps, one 1 KiB A fragment per TWO MFMAs (the tile's two 16-evaluation halves)
// Same MACs, same LDS bytes, same accumulator count.  Reports wall time (HIP events) and s_memtime cycles per layer:
// MI355X_MICROARCH.md quotes 1.12-1.14x for the 16x16x32 form in LDS-fed loops at equal cycles per FLOP (clock, not issue).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_bench.hip -o /tmp/mfma_shape_bench && /tmp/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) char smem[];

template <int S>
__global__ __launch_bounds__(512) void bench(const uint4* __restrict__ src, float* __restrict__ out,
                                             unsigned long long* __restrict__ cyc, int layers) {
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = src[i & 4095];  // 128 KiB of weights
  __syncthreads();
  bf16x8 act[16], nxt[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) act[i] = __builtin_bit_cast(bf16x8, src[(i * 64 + lane) & 4095]);
  unsigned lbase = lane * 16;  // made opaque per out tile: the weights in LDS are loop-invariant and hipcc would hoist the reads
  auto lda = [&](int f) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + (f & 127) * 1024 + lbase)); };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int l = 0; l < layers; ++l) {
    asm volatile("" : "+v"(lbase));
    bf16x8 aq[4] = {lda(0), lda(1), lda(2), lda(3)};
    if constexpr (S == 0) {
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        asm volatile("" : "+v"(lbase));
        f32x16 acc;
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.01f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const int f = o * 16 + ks;
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq[f & 3], act[ks], acc, 0, 0, 0);
          aq[f & 3] = lda(f + 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {  // ReLU + conversion: 16 floats -> two fragments' worth of bf16
          nxt[2 * o][j] = (__bf16)fmaxf(acc[j], 0.0f);
          nxt[2 * o + 1][j] = (__bf16)fmaxf(acc[8 + j], 0.0f);
        }
      }
    } else {
#pragma unroll
      for (int o = 0; o < 16; ++o) {
        asm volatile("" : "+v"(lbase));
        f32x4 c0 = {0.01f, 0.01f, 0.01f, 0.01f}, c1 = c0;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const int f = o * 8 + ks;
          // the tile's two 16-evaluation halves: B fragments act[2 ks] and act[2 ks + 1] (32 k x 16 evaluations each)
          c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[f & 3], act[2 * ks], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[f & 3], act[2 * ks + 1], c1, 0, 0, 0);
          aq[f & 3] = lda(f + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          nxt[o][j] = (__bf16)fmaxf(c0[j], 0.0f);
          nxt[o][4 + j] = (__bf16)fmaxf(c1[j], 0.0f);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) act[i] = nxt[i];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += (float)act[i][lane & 7];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int S>
static void run(const char* name, int blocks, const uint4* src, float* out, unsigned long long* cyc) {
  const int layers = 4000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(bench<S>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(bench<S>, dim3(blocks), dim3(512), 131072, 0, src, out, cyc, layers);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(bench<S>, dim3(blocks), dim3(512), 131072, 0, src, out, cyc, layers);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double flop = 2.0 * 256 * 256 * 32 * 8.0 * layers * blocks;
  printf("%-10s %4d workgroups: %8.3f ms  %7.1f TFLOP/s  %.0f s_memtime ticks per layer and wave pair (median)\n", name, blocks, ms,
         flop / ms * 1e-9, (double)h[blocks / 2] / layers);
}

int main() {
  uint4* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, 65536); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8);
  std::vector<unsigned short> h(32768);
  for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
  hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep)
    for (int blocks : {1, 256}) {
      run<0>("32x32x16", blocks, src, out, cyc);
      run<1>("16x16x32", blocks, src, out, cyc);
    }
  return 0;
}
