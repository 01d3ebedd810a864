// fast_math.h — branch-free sin/cos for positional-encoding arguments |r| <= 2^11
// (model.py:72-77: r = 2^f * coordinate, f <= 9).  Cody-Waite reduction by pi/2 with FMA,
// then Cephes single-precision minimax polynomials on [-pi/4, pi/4].  Max abs error measured on
// the host against double precision: < 2e-7 (tests/test_nerf_layout.py).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define FM_HD __host__ __device__ __forceinline__
#else
#define FM_HD inline
#endif

namespace lnrf {

FM_HD void sincos_pe(float r, float* s_out, float* c_out) {
  const float kf = rintf(r * 0.63661977236758134f);  // r * 2/pi
  float y = fmaf(kf, -1.57079637050628662109375f, r);  // fp32(pi/2)
  y = fmaf(kf, 4.37113900018624283e-8f, y);            // pi/2 - fp32(pi/2) = -4.3711e-8
  const int q = (int)kf;
  const float z = y * y;
  float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
  sp = fmaf(sp, z, -1.6666654611e-1f);
  const float sn = fmaf(sp * z, y, y);
  float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
  cp = fmaf(cp, z, 4.166664568298827e-2f);
  const float cs = fmaf(cp, z * z, fmaf(z, -0.5f, 1.0f));
  const bool swap = (q & 1) != 0;
  const float s0 = swap ? cs : sn;
  const float c0 = swap ? sn : cs;
  *s_out = (q & 2) ? -s0 : s0;
  *c_out = ((q + 1) & 2) ? -c0 : c0;
}

}  // namespace lnrf
