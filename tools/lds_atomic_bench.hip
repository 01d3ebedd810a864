// Micro-benchmark: LDS atomic-add throughput on gfx950 (float / u32 / u64), for distinct random addresses,
// for runs of equal addresses in neighbouring lanes, and for all lanes on one address.
// build: hipcc -O3 --offload-arch=gfx950 tools/lds_atomic_bench.hip -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kEntries = 16384;  // 64 KiB of float
constexpr int kThreads = 512;

template <int MODE, int RUN>
__global__ __launch_bounds__(kThreads) void bench(int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < kEntries; i += kThreads) lds[i] = 0.0f;
  __syncthreads();
  unsigned s = (blockIdx.x * kThreads + threadIdx.x / RUN) * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const unsigned idx = (s >> 8) % kEntries;
    if (MODE == 0) atomicAdd(&lds[idx], 1.0f);
    if (MODE == 1) atomicAdd(reinterpret_cast<unsigned*>(lds) + idx, 1u);
    if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(lds) + (idx >> 1), 1ull);
    if (MODE == 3) lds[idx] += 1.0f;  // plain (racy) read-modify-write, for reference
  }
  __syncthreads();
  float acc = 0.0f;
  for (int i = threadIdx.x; i < kEntries; i += kThreads) acc += lds[i];
  if (acc == 123.456f) out[0] = acc;
}

template <int MODE, int RUN>
void run(const char* name, float* out) {
  const int iters = 2000, blocks = 512;
  hipFuncSetAttribute(reinterpret_cast<const void*>(bench<MODE, RUN>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  bench<MODE, RUN><<<blocks, kThreads, 65536>>>(10, out);
  hipEventRecord(a);
  bench<MODE, RUN><<<blocks, kThreads, 65536>>>(iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double ops = (double)blocks * kThreads * iters;
  printf("%-28s run=%2d : %8.3f ms  %8.1f G lane-ops/s  (%.3f per clk per CU at 2.4 GHz, 256 CUs)\n", name, RUN, ms,
         ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  run<0, 1>("ds_add_f32 random", out);
  run<0, 4>("ds_add_f32 runs", out);
  run<0, 64>("ds_add_f32 same addr/wave", out);
  run<1, 1>("ds_add_u32 random", out);
  run<1, 4>("ds_add_u32 runs", out);
  run<1, 64>("ds_add_u32 same addr/wave", out);
  run<2, 1>("ds_add_u64 random", out);
  run<2, 4>("ds_add_u64 runs", out);
  run<3, 1>("plain rmw random", out);
  return 0;
}
