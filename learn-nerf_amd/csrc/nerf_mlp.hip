// nerf_mlp.hip — fused NeRFModel (learn_nerf/model.py:43-62) on the bf16 MFMA of gfx950.
//
// Forward  : one workgroup = 8 waves x 32 evaluations.  Activations never leave registers:
//            the f32 accumulator tile of layer l (32 out-features x 32 evaluations) is converted
//            to bf16 in place and is the B operand of layer l+1 (nerf_layout.h).  Weights are
//            pre-packed in MFMA A-fragment order and streamed L2 -> VGPR -> LDS through a
//            two-slot ring of 16 KiB stages shared by the 8 waves (one barrier per stage).
// Backward : (1) the same structure on the transposed weight stream produces the pre-activation
//            gradients dy_l and dumps them, (2) a split-K MFMA kernel reduces
//            dW_l = X_l^T dy_l over all evaluations from the forward/backward dumps.
// Precision: bf16 operands, fp32 accumulate, fp32 bias / activations / positional encoding.
#include "nerf_chain.h"

namespace lnrf {

constexpr int kFusedLds = kRingBytes + round_up(kBiasFloats * 4, 1024);
#ifdef LNRF_TIMELINE
__device__ unsigned long long* g_timeline_buf = nullptr;  // 3 kernels x 8 waves x 1024 stamps (debug build only)
#endif

struct FwdSeq {
  static constexpr int count = kFwdUsed;
  static constexpr int at(int c) { return fwd_seq(c); }
};

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
constexpr int kFwdStages = kFwdFrags / kStageFrags;  // 75

// HMASKS: the ReLU-mask slots of the hidden layers are written too (the chain backward reads them; the layer-stationary
// backward takes the mask from the saved activation itself and needs only the mask of h10, which is always written).
template <bool SAVE, bool FROM_RAYS, bool HMASKS = true>
__global__ __launch_bounds__(kThreads) void nerf_fwd_kernel(
    const char* __restrict__ packed, const float* __restrict__ xin_g, const float* __restrict__ din_g,
    const float* __restrict__ rays, int64_t ray_stride, const float* __restrict__ ts, int T, int64_t M,
    int64_t n_tiles, float* __restrict__ density, float* __restrict__ rgb, char* __restrict__ save) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;

  // biases -> LDS, inputs -> registers (all ordinary loads retire before the first LDS-DMA)
  {
    const float* bias_g = reinterpret_cast<const float*>(packed + kPackBiasOff);
    float* bias_l = reinterpret_cast<float*>(&smem[kBiasLdsOff]);
    for (int i = tid; i < kBiasFloats; i += kThreads) bias_l[i] = bias_g[i];
  }
  float px[3] = {0, 0, 0}, pd[3] = {0, 0, 0};
  if (valid) {
    if (FROM_RAYS) {
      const int64_t n = m / T;
      const float t = ts[m];
      const float* r = rays + n * ray_stride;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        pd[a] = r[3 + a];
        px[a] = r[a] + pd[a] * t;  // render.py:153
      }
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        px[a] = xin_g[m * 3 + a];
        pd[a] = din_g[m * 3 + a];
      }
    }
  }
  __syncthreads();

  Ring<kFwdStages, FwdSeq> ring;
  ring.stream = packed + kPackFwdOff;
  ring.wave = wave;
  ring.lane = lane;
#ifdef LNRF_TIMELINE
  ring.tl.buf = g_timeline_buf + (SAVE ? 0 : 8 * 1024);  // [save fwd | inference fwd | chain] x 8 waves x 1024 stamps
  ring.tl.n = 0;
  ring.tl.on = g_timeline_buf != nullptr && blockIdx.x == gridDim.x / 2;
  ring.tl.stamp();
#endif
  ring.prologue();
  LNRF_TL_STAMP(ring);

  // positional encodings (model.py:65-77) in fp32, rounded to bf16 operands
  bf16x8 xe[4], de[2];
  static_for<4>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int p = 4 * ks + pp;
      float s = 0.0f, co = 0.0f;
      if (p < 15) {
        const int pg = 15 * h + p;
        const int cd = pg / 10, f = pg - 10 * cd;
        const float v = cd == 0 ? px[0] : (cd == 1 ? px[1] : px[2]);
        sincos_pe(v * (float)(1 << f), &s, &co);
      }
      xe[ks][2 * pp] = (__bf16)s;
      xe[ks][2 * pp + 1] = (__bf16)co;
    }
  });
  static_for<2>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int p = 4 * ks + pp;
      float s = 0.0f, co = 0.0f;
      if (p < 6) {
        const int pg = 6 * h + p;
        const int cd = pg >> 2, f = pg & 3;
        const float v = cd == 0 ? pd[0] : (cd == 1 ? pd[1] : pd[2]);
        sincos_pe(v * (float)(1 << f), &s, &co);
      }
      de[ks][2 * pp] = (__bf16)s;
      de[ks][2 * pp + 1] = (__bf16)co;
    }
  });

  DumpAddr dump{save, n_tiles, tile, c, h, kSaveTileSlots};
  auto save_frag = [&](int slot, const bf16x8& f) {
    if (SAVE) dump.store(slot, frag_to_bits(f));
  };
  if (SAVE) {
    static_for<4>([&](auto i) { save_frag(kSaveXin + decltype(i)::value, xe[decltype(i)::value]); });
    static_for<2>([&](auto i) { save_frag(kSaveDin + decltype(i)::value, de[decltype(i)::value]); });
  }

  bf16x8 a0[16], a1[16];
  unsigned mask_bits[4] = {0u, 0u, 0u, 0u};

  // hidden layer: in (16 k-steps, + optional 4 extra), out 256 features
  auto hidden = [&](auto s_, bf16x8(&in)[16], bf16x8(&out)[16], auto relu_, int save_slot) {
    constexpr int S = decltype(s_)::value;
    constexpr bool RELU = decltype(relu_)::value;
    chain_layer<fwd_cons_base(S), fwd_nk(S), fwd_no(S)>(
        ring, [&](auto o_) { return bias_acc(fwd_bias_base(S) + 32 * decltype(o_)::value, h); },
        [&](auto k_) -> bf16x8 {
          constexpr int ks = decltype(k_)::value;
          if constexpr (S == 0) return xe[ks];
          else if constexpr (ks < 16) return in[ks];
          else return xe[ks - 16];
        },
        [&](auto o_, const f32x16& acc) {
          constexpr int o = decltype(o_)::value;
          out[2 * o] = acc_to_frag<0, RELU>(acc);
          out[2 * o + 1] = acc_to_frag<1, RELU>(acc);
          save_frag(save_slot + 2 * o, out[2 * o]);
          save_frag(save_slot + 2 * o + 1, out[2 * o + 1]);
          if constexpr (SAVE && RELU && HMASKS)
            mask_bits[o >> 1] |= relu_bits(out[2 * o], out[2 * o + 1]) << (16 * (o & 1));
        });
    if constexpr (SAVE && RELU && HMASKS) {
      *reinterpret_cast<uint4*>(save + dump_off(kSaveMask + S, tile, n_tiles, kSaveTileSlots) + lane * 16) =
          make_uint4(mask_bits[0], mask_bits[1], mask_bits[2], mask_bits[3]);
      mask_bits[0] = mask_bits[1] = mask_bits[2] = mask_bits[3] = 0u;
    }
  };
  std::true_type relu;
  std::false_type lin;
  hidden(std::integral_constant<int, 0>{}, a1, a0, relu, kSaveH + 0 * 16);  // Dense_0 (in: x_emb)
  hidden(std::integral_constant<int, 1>{}, a0, a1, relu, kSaveH + 1 * 16);
  hidden(std::integral_constant<int, 2>{}, a1, a0, relu, kSaveH + 2 * 16);
  hidden(std::integral_constant<int, 3>{}, a0, a1, relu, kSaveH + 3 * 16);
  hidden(std::integral_constant<int, 4>{}, a1, a0, relu, kSaveH + 4 * 16);
  hidden(std::integral_constant<int, 5>{}, a0, a1, relu, kSaveH + 5 * 16);  // Dense_5 (+x_emb), relu precedes Dense_6
  hidden(std::integral_constant<int, 6>{}, a1, a0, relu, kSaveH + 6 * 16);
  hidden(std::integral_constant<int, 7>{}, a0, a1, relu, kSaveH + 7 * 16);
  hidden(std::integral_constant<int, 8>{}, a1, a0, lin, kSaveZ);            // Dense_8: linear z (model.py:53-56)

  // Dense_10 (+ Dense_9 as row 128): z and d_emb in, relu(h10) and the density logit out
  chain_layer<fwd_cons_base(9), fwd_nk(9), fwd_no(9)>(
      ring, [&](auto o_) { return bias_acc(fwd_bias_base(9) + 32 * decltype(o_)::value, h); },
      [&](auto k_) -> bf16x8 {
        constexpr int ks = decltype(k_)::value;
        if constexpr (ks < 16) return a0[ks];
        else return de[ks - 16];
      },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        if constexpr (o < 4) {
          a1[2 * o] = acc_to_frag<0, true>(acc);
          a1[2 * o + 1] = acc_to_frag<1, true>(acc);
          save_frag(kSaveH10 + 2 * o, a1[2 * o]);
          save_frag(kSaveH10 + 2 * o + 1, a1[2 * o + 1]);
          if constexpr (SAVE) mask_bits[o >> 1] |= relu_bits(a1[2 * o], a1[2 * o + 1]) << (16 * (o & 1));
        } else {
          if constexpr (SAVE) {
            *reinterpret_cast<uint4*>(save + dump_off(kSaveMask + 8, tile, n_tiles, kSaveTileSlots) +
                                      lane * 16) = make_uint4(mask_bits[0], mask_bits[1], 0u, 0u);
          }
          if (h == 0 && valid) {
            const float x = acc[0];
            density[m] = fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));  // softplus (model.py:57)
          }
        }
      });
  // Dense_11 + tanh (model.py:60)
  chain_layer<fwd_cons_base(10), fwd_nk(10), fwd_no(10)>(
      ring, [&](auto) { return bias_acc(fwd_bias_base(10), h); },
      [&](auto k_) -> bf16x8 { return a1[decltype(k_)::value]; },
      [&](auto, const f32x16& acc) {
        if (h == 0 && valid) {
          rgb[m * 3 + 0] = tanhf(acc[0]);
          rgb[m * 3 + 1] = tanhf(acc[1]);
          rgb[m * 3 + 2] = tanhf(acc[2]);
        }
      });
}

// ---------------------------------------------------------------------------------------------
// Split-precision forward ("bf16x3", render / evaluation path; model.py:43-62 is fp32 in the reference).
// Same chain as nerf_fwd_kernel<false>, but every operand is a bf16 pair (hi, lo) with hi + lo equal to the
// fp32 value to 16 significant bits, and every product is hi*hi + hi*lo + lo*hi on the bf16 MFMA with fp32
// accumulation.  The activations of a layer therefore need 2 x 64 VGPRs in and 2 x 64 out: the workgroup is
// 4 waves (one per SIMD) so that each wave may use the whole 512-entry register file.
// ---------------------------------------------------------------------------------------------
constexpr int kFwd3Stages = kFwd3Frags / kStageFrags;  // 149
struct Fwd3Seq {
  static constexpr int count = kFwd3Used;
  static constexpr int at(int c) { return fwd3_seq(c); }
};

template <bool FROM_RAYS>
__global__ __launch_bounds__(kSplitThreads) void nerf_fwd_split_kernel(
    const char* __restrict__ packed, const float* __restrict__ xin_g, const float* __restrict__ din_g,
    const float* __restrict__ rays, int64_t ray_stride, const float* __restrict__ ts, int T, int64_t M,
    float* __restrict__ density, float* __restrict__ rgb) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kSplitWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;

  {
    const float* bias_g = reinterpret_cast<const float*>(packed + kPack3BiasOff);
    float* bias_l = reinterpret_cast<float*>(&smem[kBiasLdsOff]);
    for (int i = tid; i < kBiasFloats; i += kSplitThreads) bias_l[i] = bias_g[i];
  }
  float px[3] = {0, 0, 0}, pd[3] = {0, 0, 0};
  if (valid) {
    if (FROM_RAYS) {
      const int64_t n = m / T;
      const float t = ts[m];
      const float* r = rays + n * ray_stride;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        pd[a] = r[3 + a];
        px[a] = r[a] + pd[a] * t;  // render.py:153
      }
    } else {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        px[a] = xin_g[m * 3 + a];
        pd[a] = din_g[m * 3 + a];
      }
    }
  }
  __syncthreads();

  Ring<kFwd3Stages, Fwd3Seq, kSplitWaves> ring;
  ring.stream = packed;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();

  // positional encodings (model.py:65-77) in fp32, split into bf16 pairs
  bf16x8 xe_hi[4], xe_lo[4], de_hi[2], de_lo[2];
  static_for<4>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int p = 4 * ks + pp;
      float s = 0.0f, co = 0.0f;
      if (p < 15) {
        const int pg = 15 * h + p;
        const int cd = pg / 10, f = pg - 10 * cd;
        const float v = cd == 0 ? px[0] : (cd == 1 ? px[1] : px[2]);
        sincos_pe(v * (float)(1 << f), &s, &co);
      }
      split_store(s, xe_hi[ks], xe_lo[ks], 2 * pp);
      split_store(co, xe_hi[ks], xe_lo[ks], 2 * pp + 1);
    }
  });
  static_for<2>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int p = 4 * ks + pp;
      float s = 0.0f, co = 0.0f;
      if (p < 6) {
        const int pg = 6 * h + p;
        const int cd = pg >> 2, f = pg & 3;
        const float v = cd == 0 ? pd[0] : (cd == 1 ? pd[1] : pd[2]);
        sincos_pe(v * (float)(1 << f), &s, &co);
      }
      split_store(s, de_hi[ks], de_lo[ks], 2 * pp);
      split_store(co, de_hi[ks], de_lo[ks], 2 * pp + 1);
    }
  });

  bf16x8 a0h[16], a0l[16], a1h[16], a1l[16];

  auto hidden = [&](auto s_, bf16x8(&inh)[16], bf16x8(&inl)[16], bf16x8(&outh)[16], bf16x8(&outl)[16], auto relu_) {
    constexpr int S = decltype(s_)::value;
    constexpr bool RELU = decltype(relu_)::value;
    chain_layer_split<fwd_cons_base(S), fwd_nk(S), fwd_no(S)>(
        ring, [&](auto o_) { return bias_acc(fwd_bias_base(S) + 32 * decltype(o_)::value, h); },
        [&](auto k_) -> bf16x8 {
          constexpr int ks = decltype(k_)::value;
          if constexpr (S == 0) return xe_hi[ks];
          else if constexpr (ks < 16) return inh[ks];
          else return xe_hi[ks - 16];
        },
        [&](auto k_) -> bf16x8 {
          constexpr int ks = decltype(k_)::value;
          if constexpr (S == 0) return xe_lo[ks];
          else if constexpr (ks < 16) return inl[ks];
          else return xe_lo[ks - 16];
        },
        [&](auto o_, const f32x16& acc) {
          constexpr int o = decltype(o_)::value;
          acc_to_frag_split<0, RELU>(acc, outh[2 * o], outl[2 * o]);
          acc_to_frag_split<1, RELU>(acc, outh[2 * o + 1], outl[2 * o + 1]);
        });
  };
  std::true_type relu;
  std::false_type lin;
  hidden(std::integral_constant<int, 0>{}, a1h, a1l, a0h, a0l, relu);
  hidden(std::integral_constant<int, 1>{}, a0h, a0l, a1h, a1l, relu);
  hidden(std::integral_constant<int, 2>{}, a1h, a1l, a0h, a0l, relu);
  hidden(std::integral_constant<int, 3>{}, a0h, a0l, a1h, a1l, relu);
  hidden(std::integral_constant<int, 4>{}, a1h, a1l, a0h, a0l, relu);
  hidden(std::integral_constant<int, 5>{}, a0h, a0l, a1h, a1l, relu);
  hidden(std::integral_constant<int, 6>{}, a1h, a1l, a0h, a0l, relu);
  hidden(std::integral_constant<int, 7>{}, a0h, a0l, a1h, a1l, relu);
  hidden(std::integral_constant<int, 8>{}, a1h, a1l, a0h, a0l, lin);

  chain_layer_split<fwd_cons_base(9), fwd_nk(9), fwd_no(9)>(
      ring, [&](auto o_) { return bias_acc(fwd_bias_base(9) + 32 * decltype(o_)::value, h); },
      [&](auto k_) -> bf16x8 {
        constexpr int ks = decltype(k_)::value;
        if constexpr (ks < 16) return a0h[ks];
        else return de_hi[ks - 16];
      },
      [&](auto k_) -> bf16x8 {
        constexpr int ks = decltype(k_)::value;
        if constexpr (ks < 16) return a0l[ks];
        else return de_lo[ks - 16];
      },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        if constexpr (o < 4) {
          acc_to_frag_split<0, true>(acc, a1h[2 * o], a1l[2 * o]);
          acc_to_frag_split<1, true>(acc, a1h[2 * o + 1], a1l[2 * o + 1]);
        } else {
          if (h == 0 && valid) {
            const float x = acc[0];
            density[m] = fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));  // softplus (model.py:57)
          }
        }
      });
  chain_layer_split<fwd_cons_base(10), fwd_nk(10), fwd_no(10)>(
      ring, [&](auto) { return bias_acc(fwd_bias_base(10), h); },
      [&](auto k_) -> bf16x8 { return a1h[decltype(k_)::value]; },
      [&](auto k_) -> bf16x8 { return a1l[decltype(k_)::value]; },
      [&](auto, const f32x16& acc) {
        if (h == 0 && valid) {
          rgb[m * 3 + 0] = tanhf(acc[0]);
          rgb[m * 3 + 1] = tanhf(acc[1]);
          rgb[m * 3 + 2] = tanhf(acc[2]);
        }
      });
}

// ---------------------------------------------------------------------------------------------
// Backward, part 1: input-gradient chain (nerf_chain.h).  Produces dy_l (pre-activation gradients) dumps.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void nerf_bwd_chain_kernel(
    const char* __restrict__ packed, const char* __restrict__ save, const float* __restrict__ density,
    const float* __restrict__ rgb, const float* __restrict__ g_density, const float* __restrict__ g_rgb,
    int64_t M, int64_t n_tiles, char* __restrict__ gdump) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;

  Ring<kBwdStages, BwdSeq> ring;
  ring.stream = packed + kPackBwdOff;
  ring.wave = wave;
  ring.lane = lane;
#ifdef LNRF_TIMELINE
  ring.tl.buf = g_timeline_buf + 16 * 1024;
  ring.tl.n = 0;
  ring.tl.on = g_timeline_buf != nullptr && blockIdx.x == gridDim.x / 2;
  ring.tl.stamp();
#endif
  GlobalDumpSink sink{DumpAddr{gdump, n_tiles, tile, lane & 31, lane >> 5, kGradTileSlots}};
  bwd_chain_tile(ring, sink, save, n_tiles, density, rgb, g_density, g_rgb, M, tile, lane);
}

// ---------------------------------------------------------------------------------------------
// Backward, part 2: weight gradients  dW_l[in][out] += sum_m X_l[m][in] * dy_l[m][out]
// A problem = (X tensor of NXF k-step slots) x (dy tensor of NYF slots) for one Dense kernel (or a
// row block of it).  blockIdx -> (problem, K-slice of 32-evaluation steps).  Per step the workgroup
// stages the X and dy fragments into LDS (global -> VGPR -> LDS, two steps of loads in flight) and
// every wave reads its operand tiles transposed (ds_read_b64_tr_b16: feature on the lane,
// evaluation in the registers) for the 32x32x16 MFMA.  Partial sums leave by fp32 atomics.
// ---------------------------------------------------------------------------------------------
// One launch for every Dense layer of the model: blockIdx -> problem -> operand-shape body.
// Problems are listed heaviest first so that the small ones fill the tail of the launch.
template <bool PLAIN>
__global__ __launch_bounds__(kThreads) void nerf_wgrad_kernel(WgradArgs args, const char* __restrict__ save,
                                                              const char* __restrict__ gdump,
                                                              int64_t n_tiles, float* __restrict__ grads,
                                                              WgLayout lay, float* __restrict__ slabs) {
  WgradProblem pb = args.p[0];
#pragma unroll
  for (int i = 1; i < kMaxProblems; ++i)
    if (i < args.n_problems && (int)blockIdx.x >= args.p[i].first_block) pb = args.p[i];
  switch (pb.shape) {
    // steps per barrier chosen so that every body keeps ~60 KB of loads in flight per workgroup
    case 0: wgrad_body<16, 16, 4, 2, 2, NerfWgradEpi, PLAIN, (kSaveTileSlots > 0)>(pb, save, gdump, n_tiles, grads, lay, slabs); break;
    case 1: wgrad_body<16, 10, 4, 2, 2, NerfWgradEpi, PLAIN, (kSaveTileSlots > 0)>(pb, save, gdump, n_tiles, grads, lay, slabs); break;
    case 2: wgrad_body<4, 16, 2, 4, 3, NerfWgradEpi, PLAIN, (kSaveTileSlots > 0)>(pb, save, gdump, n_tiles, grads, lay, slabs); break;
    case 3: wgrad_body<2, 10, 1, 8, 5, NerfWgradEpi, PLAIN, (kSaveTileSlots > 0)>(pb, save, gdump, n_tiles, grads, lay, slabs); break;
    case 5: wgrad_body<18, 8, 4, 2, 2, NerfWgradEpi, PLAIN, (kSaveTileSlots > 0)>(pb, save, gdump, n_tiles, grads, lay, slabs); break;  // Ref-NeRF Dense_9
    case 6: wgrad_body<18, 10, 4, 2, 2, NerfWgradEpi, PLAIN, (kSaveTileSlots > 0)>(pb, save, gdump, n_tiles, grads, lay, slabs); break;  // [z | d_emb] x dy10m
    case 7: wgrad_body<4, 32, 2, 4, 2, NerfWgradEpi, PLAIN, (kSaveTileSlots > 0)>(pb, save, gdump, n_tiles, grads, lay, slabs); break;   // x_emb x [dy0 | dy5]
    default: wgrad_body<8, 2, 4, 2, 6, NerfWgradEpi, PLAIN, (kSaveTileSlots > 0)>(pb, save, gdump, n_tiles, grads, lay, slabs); break;
  }
}

// Folds the slabs of a nerf_wgrad_kernel launch (slab epilogue): blockIdx.x = (problem * 8 + wave) * kSlabMaxTiles + tile.
__global__ __launch_bounds__(64 * kSlabReduceWaves) void nerf_wgrad_reduce_kernel(WgradArgs args,
                                                                                 const float* __restrict__ slabs,
                                                                                 float* __restrict__ grads) {
  __shared__ float lds[(kSlabReduceWaves - 1) * 17 * 64];
  const int prob = blockIdx.x / (kWaves * kSlabMaxTiles), w = blockIdx.x / kSlabMaxTiles % kWaves, j = blockIdx.x % kSlabMaxTiles;
  const WgradProblem pb = args.p[prob];
  switch (pb.shape) {  // the shapes of nerf_wgrad_kernel
    case 0: wgrad_reduce_tile<16, 16, 4, 2, NerfWgradEpi>(pb, w, j, slabs, grads, lds); break;
    case 1: wgrad_reduce_tile<16, 10, 4, 2, NerfWgradEpi>(pb, w, j, slabs, grads, lds); break;
    case 2: wgrad_reduce_tile<4, 16, 2, 4, NerfWgradEpi>(pb, w, j, slabs, grads, lds); break;
    case 3: wgrad_reduce_tile<2, 10, 1, 8, NerfWgradEpi>(pb, w, j, slabs, grads, lds); break;
    case 5: wgrad_reduce_tile<18, 8, 4, 2, NerfWgradEpi>(pb, w, j, slabs, grads, lds); break;
    case 6: wgrad_reduce_tile<18, 10, 4, 2, NerfWgradEpi>(pb, w, j, slabs, grads, lds); break;
    case 7: wgrad_reduce_tile<4, 32, 2, 4, NerfWgradEpi>(pb, w, j, slabs, grads, lds); break;
    default: wgrad_reduce_tile<8, 2, 4, 2, NerfWgradEpi>(pb, w, j, slabs, grads, lds); break;
  }
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
// One thread per lane slot of a fragment (8 bf16 = one 16-byte store) or per bias float.
__global__ void nerf_pack_kernel(const float* __restrict__ params, char* __restrict__ packed) {
  constexpr int64_t units_f = (int64_t)kFwdFrags * 64, units_b = (int64_t)kBwdFrags * 64;
  constexpr int64_t total = units_f + units_b + kBiasFloats;
  for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
    if (u < units_f + units_b) {
      const bool fwd = u < units_f;
      const int64_t uu = fwd ? u : u - units_f;
      const int g = (int)(uu >> 6), lane = (int)(uu & 63);
      int layer = 0, loc, nk, n_used;
      if (fwd) {
        for (int i = 1; i < kFwdLayers; ++i)
          if (g >= fwd_base(i)) layer = i;
        loc = g - fwd_base(layer); nk = fwd_nk(layer); n_used = nk * fwd_no(layer);
      } else {
        for (int i = 1; i < kBwdLayers; ++i)
          if (g >= bwd_base(i)) layer = i;
        loc = g - bwd_base(layer); nk = bwd_nk(layer); n_used = nk * bwd_no(layer);
      }
      bf16x8 out;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int idx = -1;
        if (loc < n_used) idx = fwd ? fwd_weight_index(layer, loc / nk, loc % nk, lane, j)
                                    : bwd_weight_index(layer, loc / nk, loc % nk, lane, j);
        out[j] = (__bf16)(idx >= 0 ? params[idx] : 0.0f);
      }
      *reinterpret_cast<bf16x8*>(packed + (fwd ? kPackFwdOff : kPackBwdOff) + uu * 16) = out;
    } else {
      const int i = (int)(u - units_f - units_b);
      int s = 0;
      for (int k = 1; k < kFwdLayers; ++k)
        if (i >= fwd_bias_base(k)) s = k;
      const int idx = fwd_bias_index(s, i - fwd_bias_base(s));
      reinterpret_cast<float*>(packed + kPackBiasOff)[i] = idx >= 0 ? params[idx] : 0.0f;
    }
  }
}

// split stream: [hi frag, lo frag] per forward (layer, out-tile, k-step); fp32 bias block behind it
__global__ void nerf_pack_split_kernel(const float* __restrict__ params, char* __restrict__ packed) {
  const int64_t total_w = (int64_t)kFwd3Frags * 512;
  const int64_t total = total_w + kBiasFloats;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    if (e < total_w) {
      const int g = (int)(e >> 9), lane = (int)((e >> 3) & 63), j = (int)(e & 7);
      int s = 0;
      for (int i = 1; i < kFwdLayers; ++i)
        if (g >= fwd3_base(i)) s = i;
      const int loc = g - fwd3_base(s);
      int idx = -1;
      if (loc < 2 * fwd_nk(s) * fwd_no(s)) {
        const int pair = loc >> 1;
        idx = fwd_weight_index(s, pair / fwd_nk(s), pair % fwd_nk(s), lane, j);
      }
      const float w = idx >= 0 ? params[idx] : 0.0f;
      const __bf16 hi = (__bf16)w;
      reinterpret_cast<__bf16*>(packed)[e] = (loc & 1) ? (__bf16)(w - (float)hi) : hi;
    } else {
      const int i = (int)(e - total_w);
      int s = 0;
      for (int k = 1; k < kFwdLayers; ++k)
        if (i >= fwd_bias_base(k)) s = k;
      const int idx = fwd_bias_index(s, i - fwd_bias_base(s));
      reinterpret_cast<float*>(packed + kPack3BiasOff)[i] = idx >= 0 ? params[idx] : 0.0f;
    }
  }
}

}  // namespace lnrf

using namespace lnrf;

static bool shape_supported(const lnrf_nerf_shape* s) { return nerf_shape_fused(s); }
static inline int64_t tiles_for(int64_t m) { return nerf_tiles_for(m); }

extern "C" int64_t lnrf_nerf_param_count(const lnrf_nerf_shape* s) {
  if (!s) return -1;
  const int64_t xe = 6 * s->x_freqs, de = 6 * s->d_freqs, hd = s->hidden_dim, cd = s->color_layer_dim;
  int64_t n = 0, fan = xe;
  for (int i = 0; i < s->input_layers; ++i) { n += fan * hd + hd; fan = hd; }
  fan = hd + xe;
  for (int i = 0; i < s->mid_layers; ++i) { n += fan * hd + hd; fan = hd; }
  n += hd + 1;
  n += (hd + de) * cd + cd;
  n += cd * 3 + 3;
  return n;
}
extern "C" int64_t lnrf_nerf_packed_bytes(const lnrf_nerf_shape* s) {
  return shape_supported(s) ? kPackBytes : -1;
}
extern "C" int64_t lnrf_nerf_save_bytes(const lnrf_nerf_shape* s, int64_t m) {
  return shape_supported(s) ? (int64_t)kSaveSlots * tiles_for(m) * kFragBytes : -1;
}
// weight-gradient workgroups of one NeRFModel launch (sum of the per-problem counts in lnrf_nerf_mlp_bwd_weights)
constexpr int kNerfWgradBlocks = 512;
static int64_t grad_dump_bytes(int64_t m) { return (int64_t)kGradSlots * tiles_for(m) * kFragBytes; }
extern "C" int64_t lnrf_nerf_bwd_scratch_bytes(const lnrf_nerf_shape* s, int64_t m) {
  // the dy dump, then the partial-sum slabs of the weight-gradient launch (fused_chain.h, slab epilogue)
  return shape_supported(s) ? grad_dump_bytes(m) + kNerfWgradBlocks * kSlabBlockBytes : -1;
}

extern "C" int lnrf_nerf_pack_weights(const lnrf_nerf_shape* shape, const float* params, void* packed,
                                      lnrf_stream_t stream) {
  if (!shape_supported(shape)) {
    set_error("lnrf_nerf_pack_weights: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(params && packed, "null pointer");
  hipLaunchKernelGGL(nerf_pack_kernel, dim3(640), dim3(256), 0, as_stream(stream), params, (char*)packed);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

template <class K>
static int ensure_lds(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(max dynamic LDS)");
  return LNRF_OK;
}

static int nerf_fwd_launch(const lnrf_nerf_shape* shape, const void* packed, const float* x, const float* d,
                           const float* rays, int64_t ray_stride, const float* ts, int32_t t, int64_t m,
                           float* density, float* rgb, void* save, bool hidden_masks, lnrf_stream_t stream) {
  if (!shape_supported(shape)) {
    set_error("lnrf_nerf_mlp_fwd: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(packed && density && rgb, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  const bool from_rays = rays != nullptr;
  if (from_rays) LNRF_CHECK_ARG(ts && t >= 1 && ray_stride >= 6 && m % t == 0, "rays mode needs ts, t, m = n*t");
  else LNRF_CHECK_ARG(x && d, "need x and d (or rays and ts)");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = tiles_for(m);
  const dim3 grid((unsigned)((n_tiles + kWaves - 1) / kWaves)), block(kThreads);
  hipStream_t st = as_stream(stream);
  int rc;
#define LAUNCH_FWD(SAVE, RAYS, HM)                                                                        \
  do {                                                                                                    \
    rc = ensure_lds(nerf_fwd_kernel<SAVE, RAYS, HM>, kFusedLds);                                          \
    if (rc) return rc;                                                                                    \
    hipLaunchKernelGGL((nerf_fwd_kernel<SAVE, RAYS, HM>), grid, block, kFusedLds, st, (const char*)packed, \
                       x, d, rays, ray_stride, ts, (int)t, m, n_tiles, density, rgb, (char*)save);        \
  } while (0)
  if (save != nullptr && hidden_masks) {
    if (from_rays) LAUNCH_FWD(true, true, true); else LAUNCH_FWD(true, false, true);
  } else if (save != nullptr) {
    if (from_rays) LAUNCH_FWD(true, true, false); else LAUNCH_FWD(true, false, false);
  } else {
    if (from_rays) LAUNCH_FWD(false, true, true); else LAUNCH_FWD(false, false, true);
  }
#undef LAUNCH_FWD
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_nerf_mlp_fwd(const lnrf_nerf_shape* shape, const void* packed, const float* x,
                                 const float* d, const float* rays, int64_t ray_stride, const float* ts,
                                 int32_t t, int64_t m, float* density, float* rgb, void* save,
                                 lnrf_stream_t stream) {
  return nerf_fwd_launch(shape, packed, x, d, rays, ray_stride, ts, t, m, density, rgb, save, true, stream);
}

extern "C" int lnrf_nerf_mlp_fwd_ls(const lnrf_nerf_shape* shape, const void* packed, const float* x,
                                    const float* d, const float* rays, int64_t ray_stride, const float* ts,
                                    int32_t t, int64_t m, float* density, float* rgb, void* save,
                                    lnrf_stream_t stream) {
  return nerf_fwd_launch(shape, packed, x, d, rays, ray_stride, ts, t, m, density, rgb, save, false, stream);
}

extern "C" int64_t lnrf_nerf_packed_split_bytes(const lnrf_nerf_shape* s) {
  return shape_supported(s) ? kPack3Bytes : -1;
}

extern "C" int lnrf_nerf_pack_weights_split(const lnrf_nerf_shape* shape, const float* params, void* packed,
                                            lnrf_stream_t stream) {
  if (!shape_supported(shape)) {
    set_error("lnrf_nerf_pack_weights_split: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(params && packed, "null pointer");
  hipLaunchKernelGGL(nerf_pack_split_kernel, dim3(1024), dim3(256), 0, as_stream(stream), params, (char*)packed);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_nerf_mlp_fwd_split(const lnrf_nerf_shape* shape, const void* packed_split, const float* x,
                                       const float* d, const float* rays, int64_t ray_stride, const float* ts,
                                       int32_t t, int64_t m, float* density, float* rgb, lnrf_stream_t stream) {
  if (!shape_supported(shape)) {
    set_error("lnrf_nerf_mlp_fwd_split: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(packed_split && density && rgb, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  const bool from_rays = rays != nullptr;
  if (from_rays) LNRF_CHECK_ARG(ts && t >= 1 && ray_stride >= 6 && m % t == 0, "rays mode needs ts, t, m = n*t");
  else LNRF_CHECK_ARG(x && d, "need x and d (or rays and ts)");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = (m + kTileCols - 1) / kTileCols;
  const dim3 grid((unsigned)((n_tiles + kSplitWaves - 1) / kSplitWaves)), block(kSplitThreads);
  hipStream_t st = as_stream(stream);
  int rc;
  if (from_rays) {
    rc = ensure_lds(nerf_fwd_split_kernel<true>, kFusedLds);
    if (rc) return rc;
    hipLaunchKernelGGL((nerf_fwd_split_kernel<true>), grid, block, kFusedLds, st, (const char*)packed_split, x, d,
                       rays, ray_stride, ts, (int)t, m, density, rgb);
  } else {
    rc = ensure_lds(nerf_fwd_split_kernel<false>, kFusedLds);
    if (rc) return rc;
    hipLaunchKernelGGL((nerf_fwd_split_kernel<false>), grid, block, kFusedLds, st, (const char*)packed_split, x, d,
                       rays, ray_stride, ts, (int)t, m, density, rgb);
  }
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_nerf_mlp_bwd_chain(const lnrf_nerf_shape* shape, const void* packed, const void* save,
                                       const float* density, const float* rgb, const float* g_density,
                                       const float* g_rgb, int64_t m, void* scratch, lnrf_stream_t stream) {
  if (!shape_supported(shape)) {
    set_error("lnrf_nerf_mlp_bwd_chain: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(packed && save && density && rgb && g_density && g_rgb && scratch, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = tiles_for(m);
  hipStream_t st = as_stream(stream);
  int rc = ensure_lds(nerf_bwd_chain_kernel, kFusedLds);
  if (rc) return rc;
  hipLaunchKernelGGL(nerf_bwd_chain_kernel, dim3((unsigned)((n_tiles + kWaves - 1) / kWaves)), dim3(kThreads),
                     kFusedLds, st, (const char*)packed, (const char*)save, density, rgb, g_density, g_rgb, m,
                     n_tiles, (char*)scratch);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_nerf_mlp_bwd(const lnrf_nerf_shape* shape, const void* packed, const void* save,
                                 const float* density, const float* rgb, const float* g_density,
                                 const float* g_rgb, int64_t m, void* scratch, float* grads,
                                 lnrf_stream_t stream) {
  int rc = lnrf_nerf_mlp_bwd_chain(shape, packed, save, density, rgb, g_density, g_rgb, m, scratch, stream);
  if (rc) return rc;
  return lnrf_nerf_mlp_bwd_weights(shape, save, scratch, m, grads, stream);
}

int lnrf::launch_nerf_wgrad(const WgradArgs& args, int blocks, const void* xbuf, const void* ybuf, int64_t n_tiles,
                            float* grads, hipStream_t stream, WgLayout lay, float* slabs, bool plain, bool fold) {
  const int lds = 2 * 2 * 36 * kFragBytes;  // largest body: 2 buffers x 2 steps x (4 + 32) fragments
  hipError_t e = hipFuncSetAttribute(plain ? reinterpret_cast<const void*>(nerf_wgrad_kernel<true>)
                                           : reinterpret_cast<const void*>(nerf_wgrad_kernel<false>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(max dynamic LDS)");
  {
    static const int il = exp_env_int("LNRF_WGRAD_INTERLEAVE", -1);  // experiment builds only (common.h): overrides the caller
    if (il >= 0) lay.interleave = il;
  }
  if (plain)
    hipLaunchKernelGGL(nerf_wgrad_kernel<true>, dim3((unsigned)blocks), dim3(kThreads), lds, stream, args,
                       (const char*)xbuf, (const char*)ybuf, n_tiles, grads, lay, slabs);
  else
    hipLaunchKernelGGL(nerf_wgrad_kernel<false>, dim3((unsigned)blocks), dim3(kThreads), lds, stream, args,
                       (const char*)xbuf, (const char*)ybuf, n_tiles, grads, lay, slabs);
  LNRF_LAUNCH_CHECK();
  if (slabs != nullptr && fold) {  // room for `blocks` slabs of kSlabBlockBytes is the caller's business
    hipLaunchKernelGGL(nerf_wgrad_reduce_kernel, dim3((unsigned)(args.n_problems * kWaves * kSlabMaxTiles)), dim3(64 * kSlabReduceWaves), 0, stream, args,
                       (const float*)slabs, grads);
    LNRF_LAUNCH_CHECK();
  }
  return LNRF_OK;
}

extern "C" int lnrf_nerf_mlp_bwd_weights(const lnrf_nerf_shape* shape, const void* save, void* scratch,
                                         int64_t m, float* grads, lnrf_stream_t stream) {
  if (!shape_supported(shape)) {
    set_error("lnrf_nerf_mlp_bwd_weights: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(save && scratch && grads, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = tiles_for(m);
  hipStream_t st = as_stream(stream);
  int rc;
  // weight-gradient problems: ONE launch, heaviest problems first, blocks proportional to bytes
  WgradArgs a;
  int blocks[13] = {48, 48, 48, 48, 47, 47, 47, 47, 39, 30, 30, 18, 15};
  {
    // experiment builds only: LNRF_WGRAD_BLOCK_SCALE=<percent> scales the per-problem workgroup counts (100 = 512 total)
    static const int pct = exp_env_int("LNRF_WGRAD_BLOCK_SCALE", 100);
    if (pct > 0 && pct != 100)
      for (int i = 0; i < 13; ++i) blocks[i] = (blocks[i] * pct + 50) / 100 < 1 ? 1 : (blocks[i] * pct + 50) / 100;
  }
  const int first = build_wgrad_problems(a, blocks, (n_tiles + 5) / 6);
  (void)rc;
  // experiment builds only: LNRF_WGRAD_ATOMICS=1 keeps the older fp32-atomic epilogue (A/B)
  static const bool atomics = exp_env_is("LNRF_WGRAD_ATOMICS", '1');
  float* slabs = (atomics || first > kNerfWgradBlocks)
                     ? nullptr
                     : reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + grad_dump_bytes(m));
  // operand loads (fused_chain.h WgStage::load): ordinary for the big (fine-pass) launch, non-temporal for the small one —
  // measured on this model only (the Ref-NeRF launches are faster with non-temporal loads at every size);
  // (experiment builds: LNRF_WGRAD_PLAIN_TILES overrides the threshold, in tiles of 32 evaluations)
  static const int64_t plain_from = exp_env_int("LNRF_WGRAD_PLAIN_TILES", 16384);
  return launch_nerf_wgrad(a, first, save, scratch, n_tiles, grads, st, WgLayout{kSaveTileSlots, kGradTileSlots}, slabs,
                           n_tiles >= plain_from);
}

#ifdef LNRF_TIMELINE
// debug library only (tools/build_timeline.sh): where the stamped workgroup writes its s_memtime values
extern "C" int lnrf_debug_set_timeline(void* buf) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(lnrf::g_timeline_buf), &buf, sizeof(buf));
  return e == hipSuccess ? LNRF_OK : hip_fail(e, "hipMemcpyToSymbol(g_timeline_buf)");
}
#endif
