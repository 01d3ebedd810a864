// mfma_issue_bench.hip — what one wave per SIMD sustains on v_mfma_f32_32x32x16_bf16 under the operand-feeding patterns of
// the fused NeRF kernels (cycles per MFMA from s_memtime, median over workgroups).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_issue_bench.hip -o tools/mfma_issue_bench && tools/mfma_issue_bench
// V0 registers only, 2 dependent chains        V1 + one ds_read_b128 (B) per 2 MFMAs, read 3 steps ahead
// V2 V1 with the A operand from 32 different registers (as W^T in nerf_bwd_ls)   V3 V2 + 4 chains instead of 2
// V4 16 independent accumulators (weight-gradient shape), A/B from ds_read_b64_tr_b16 pairs
// V5 V1 but 2 waves per SIMD (512 threads)
// V6 V1 + one 1 KiB LDS-DMA (global_load_lds_dwordx4) per k-step of every second step, all four waves in the same step
// V7 V6 but the waves take turns: wave w issues only in the steps ks = w (mod 4) (same pieces per wave: 4 of 16 steps)
// V8 V6 with the pieces in 4 of 16 steps (same count as V7, all waves in the same steps)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) char smem[];
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

template <int V>
__global__ __launch_bounds__(V == 5 ? 512 : 256) void bench(const uint4* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ cyc, int iters, const char* __restrict__ big) {
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = src[i];
  __syncthreads();
  bf16x8 wa[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) wa[i] = __builtin_bit_cast(bf16x8, src[(i * 64 + lane) & 4095]);
  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
  auto ldb = [&](int k) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(smem + (k & 31) * 1024 + lane * 16)); };
  auto ldtr = [&](int k) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + (k & 31) * 1024 + lane * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(smem + (k & 31) * 1024 + 512 + lane * 8));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (V == 0) {
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        acc[0] = MFMA(wa[0], wa[1], acc[0]);
        acc[1] = MFMA(wa[2], wa[1], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (V == 1 || V == 5) {
      bf16x8 bq[3] = {ldb(0), ldb(1), ldb(2)};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 b = bq[ks % 3];
        acc[0] = MFMA(wa[0], b, acc[0]);
        acc[1] = MFMA(wa[2], b, acc[1]);
        if (ks + 3 < 16) bq[ks % 3] = ldb(ks + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (V == 6 || V == 7 || V == 8) {
      bf16x8 bq[3] = {ldb(0), ldb(1), ldb(2)};
      const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 b = bq[ks % 3];
        acc[0] = MFMA(wa[0], b, acc[0]);
        acc[1] = MFMA(wa[2], b, acc[1]);
        if (ks + 3 < 16) bq[ks % 3] = ldb(ks + 3);
        const bool mine = V == 6 ? (ks & 1) == 0 : (V == 7 ? (ks & 3) == wave : (ks & 3) == 0);
        if (mine) {
          unsigned keep;
          const unsigned voff = (threadIdx.x * 16 + ((it * 16 + ks) & 255) * 4096) & 0xFFFFF;
          const unsigned dst = 32768 + wave * 1024;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(voff), "s"(big), "s"(dst) : "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if constexpr (V == 2) {
      bf16x8 bq[3] = {ldb(0), ldb(1), ldb(2)};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 b = bq[ks % 3];
        acc[0] = MFMA(wa[ks], b, acc[0]);
        acc[1] = MFMA(wa[16 + ks], b, acc[1]);
        if (ks + 3 < 16) bq[ks % 3] = ldb(ks + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (V == 3) {
      bf16x8 bq[3] = {ldb(0), ldb(1), ldb(2)};
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const bf16x8 b = bq[ks % 3];
        acc[(ks & 1) * 2] = MFMA(wa[ks], b, acc[(ks & 1) * 2]);
        acc[(ks & 1) * 2 + 1] = MFMA(wa[16 + ks], b, acc[(ks & 1) * 2 + 1]);
        if (ks + 3 < 16) bq[ks % 3] = ldb(ks + 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (V == 4) {
      bf16x8 a0 = ldtr(0), a1 = ldtr(2);
      bf16x8 fq[3] = {ldtr(4), ldtr(5), ldtr(6)};
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const bf16x8 bf = fq[e % 3];
        acc[e & 7] = MFMA(a0, bf, acc[e & 7]);
        acc[8 + (e & 7)] = MFMA(a1, bf, acc[8 + (e & 7)]);
        if (e + 3 < 16) fq[e % 3] = ldtr(7 + e);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][lane & 15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V>
static void run(const char* name, int threads, int blocks, const uint4* src, float* out, unsigned long long* cyc, const char* big) {
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(bench<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(bench<V>, dim3(blocks), dim3(threads), 65536, 0, src, out, cyc, iters, big);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-58s %4d blocks x %3d threads: %.1f cycles per MFMA (median), min %.1f max %.1f\n", name, blocks, threads,
         (double)h[blocks / 2] / (iters * 32.0), (double)h[0] / (iters * 32.0), (double)h[blocks - 1] / (iters * 32.0));
}

int main() {
  uint4* src; float* out; unsigned long long* cyc;
  char* big; hipMalloc(&big, 1 << 21); hipMemset(big, 0, 1 << 21); hipMalloc(&src, 65536); hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8);
  std::vector<unsigned short> h(32768);
  for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);  // random bf16 values around +-0.01..0.03
  hipMemcpy(src, h.data(), 65536, hipMemcpyHostToDevice);
  for (int blocks : {1, 256}) {
    run<0>("V0 registers only, 2 dependent chains", 256, blocks, src, out, cyc, big);
    run<1>("V1 + ds_read_b128 B per 2 MFMAs, 3 steps ahead", 256, blocks, src, out, cyc, big);
    run<2>("V2 V1 with A from 32 different registers", 256, blocks, src, out, cyc, big);
    run<3>("V3 V2 with 4 chains", 256, blocks, src, out, cyc, big);
    run<4>("V4 16 accumulators, operands by ds_read_b64_tr_b16", 256, blocks, src, out, cyc, big);
    run<5>("V5 V1 with 2 waves per SIMD", 512, blocks, src, out, cyc, big);
    run<6>("V6 V1 + LDS-DMA piece every 2nd step, waves together", 256, blocks, src, out, cyc, big);
    run<8>("V8 V1 + LDS-DMA piece every 4th step, waves together", 256, blocks, src, out, cyc, big);
    run<7>("V7 V1 + LDS-DMA piece every 4th step, waves take turns", 256, blocks, src, out, cyc, big);
  }
  return 0;
}
