// philox.h — Philox4x32-10 counter RNG, bit-identical to oracle/philox.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lnrf {

struct PhiloxKey {
  uint32_t k0, k1;
};

__host__ __device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
  const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
  const uint32_t n0 = hi1 ^ c[1] ^ k0;
  const uint32_t n2 = hi0 ^ c[3] ^ k1;
  c[0] = n0;
  c[1] = lo1;
  c[2] = n2;
  c[3] = lo0;
}

// uniform in [0,1) for element e of `stream` under `seed`
__host__ __device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t stream, uint64_t e) {
  const uint64_t ctr = e >> 2;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), stream, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const uint32_t sel = (uint32_t)(e & 3);
  const uint32_t w = sel == 0 ? c[0] : sel == 1 ? c[1] : sel == 2 ? c[2] : c[3];
  return (float)(w >> 8) * 5.9604644775390625e-08f;  // 2^-24
}

}  // namespace lnrf
