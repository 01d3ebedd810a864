// ngp_mlp.hip — fused InstantNGPModel MLP (learn_nerf/instant_ngp.py:38-54) on the bf16 MFMA of gfx950.
//
//   enc (L*F hash-grid features) -> Dense_0 64 relu -> Dense_1 16 (density = exp(out[0]))
//   [d_emb(24), out(16)] -> Dense_2 64 relu -> Dense_3 64 relu -> Dense_4 3 tanh
//
// Same construction as nerf_mlp.hip: one wave owns 32 evaluations, the f32 accumulator tile of a layer is
// converted in place into the B operand of the next one, the packed A-fragment stream (26 + 20 fragments)
// goes through the shared LDS ring.  The network is so small that the backward kernel recomputes the
// forward (26 MFMAs) instead of reading saved activations, then runs the transposed chain, writes
// d loss / d enc feature-major for the hash-grid scatter, and dumps X / dy fragments (about 1 KiB per
// evaluation) for the shared split-K weight-gradient body.
// Precision: bf16 operands, fp32 accumulate, fp32 bias / activations / directional encoding.

#include "fused_chain.h"

namespace lnrf {

constexpr int kNgpLayers = 5;
constexpr int kNgpHidden = 64, kNgpDensityDim = 16, kNgpDembDim = 24;
constexpr int kNgpStreamFrags = 48;                               // 3 ring stages
constexpr int kNgpBiasFloats = 256;
constexpr int kNgpPackBiasOff = kNgpStreamFrags * kFragBytes;
constexpr int kNgpPackBytes = kNgpPackBiasOff + kNgpBiasFloats * 4;
constexpr int kNgpLds = kRingBytes + 1024;

// forward layer l: k-steps, 32-row out tiles, first fragment (consumption order == stream order)
constexpr int ngp_fwd_nk(int l, int ne) { return l == 0 ? ne : (l == 2 ? 3 : 4); }
constexpr int ngp_fwd_no(int l) { return (l == 1 || l == 4) ? 1 : 2; }
constexpr int ngp_fwd_base(int l, int ne) {
  int b = 0;
  for (int i = 0; i < l; ++i) b += ngp_fwd_nk(i, ne) * ngp_fwd_no(i);
  return b;
}
constexpr int ngp_fwd_count(int ne) { return ngp_fwd_base(kNgpLayers, ne); }
// backward step t applies Dense_{4-t}^T
constexpr int ngp_bwd_nk(int t) { return (t == 0 || t == 3) ? 1 : 4; }
constexpr int ngp_bwd_no(int t) { return (t == 2 || t == 4) ? 1 : 2; }
constexpr int ngp_bwd_base(int t, int ne) {
  int b = ngp_fwd_count(ne);
  for (int i = 0; i < t; ++i) b += ngp_bwd_nk(i) * ngp_bwd_no(i);
  return b;
}
constexpr int ngp_total_count(int ne) { return ngp_bwd_base(kNgpLayers, ne); }
constexpr int ngp_bias_base(int l) { return l == 0 ? 0 : (l == 1 ? 64 : (l == 2 ? 96 : (l == 3 ? 160 : 224))); }
__host__ __device__ constexpr int ngp_out_dim(int l) { return l == 1 ? kNgpDensityDim : (l == 4 ? 3 : kNgpHidden); }

// dump slots of the backward scratch ([slot][tile][1 KiB]): X tensors then dy tensors
constexpr int kNgpXEnc = 0, kNgpXH0 = 2, kNgpXCat = 6, kNgpXC1 = 10, kNgpXC2 = 14;
constexpr int kNgpDy0 = 18, kNgpDy1 = 22, kNgpDy2 = 24, kNgpDy3 = 28, kNgpDy4 = 32;
constexpr int kNgpSlots = 34;

template <int NE, bool BWD>
struct NgpSeq {
  static constexpr int count = BWD ? ngp_total_count(NE) : ngp_fwd_count(NE);
  static constexpr int at(int c) { return c; }
};

// dh * relu'(h): keep accumulator registers 8S..8S+7 where the bf16 activation fragment is non-zero
template <int S>
__device__ __forceinline__ bf16x8 masked_by(const f32x16& acc, const bf16x8& ref) {
  const uint4 rb = frag_to_bits(ref);
  const unsigned w[4] = {rb.x, rb.y, rb.z, rb.w};
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned e = (j & 1) ? (w[j >> 1] >> 16) : (w[j >> 1] & 0xFFFFu);
    f[j] = (__bf16)(e != 0u ? acc[8 * S + j] : 0.0f);
  }
  return f;
}

// FUSED (backward only): the weight gradients are formed inside this kernel.  The model has 10 K parameters, so a
// workgroup can keep its share of dW in registers for the whole launch: the workgroups are persistent (one per CU,
// groups of 8 tiles taken round-robin), and after every backward step the 8 waves put the layer's X and dy fragments
// into an LDS staging area (the layout the split-K body of fused_chain.h reads: one 32-evaluation step per wave),
// then every wave accumulates ONE 32x32 tile of that layer's dW over its share of the 8 steps (transposed
// ds_read_b64_tr_b16 reads) and the tiles leave by fp32 atomics once, at the end of the launch.  This removes the
// 34 KiB per tile of X / dy dumps (written and read back: 2.3 GB per step at 4096 rays) and the second launch.
struct NgpWgradProblem {
  int shape, x_slot0, y_slot0, do_bias, first_block, n_blocks;
  int out_dim;             // columns of the Flax kernel (= valid dy features)
  unsigned w_lo, w_hi, b_lo, b_hi;  // float offsets of kernel / bias in the gradient vector (64-bit, split)
  int rb0, rb1, rb2, rb3;  // per X fragment: first kernel row ...
  int rv0, rv1, rv2, rv3;  // ... and how many of its 16 features are real (scalars: keeps the struct in SGPRs)
};
struct NgpWgradArgs {
  NgpWgradProblem p[kNgpLayers];
};
struct NgpWgradEpi {
  static __device__ __forceinline__ void cols(const NgpWgradProblem& pb, int ot, int colr, int& out_idx,
                                              int& out_dim, int64_t& w_off, int64_t& b_off) {
    const int idx = 32 * ot + colr;
    out_idx = idx < pb.out_dim ? idx : -1;
    out_dim = pb.out_dim;
    w_off = (int64_t)(((uint64_t)pb.w_hi << 32) | pb.w_lo);
    b_off = (int64_t)(((uint64_t)pb.b_hi << 32) | pb.b_lo);
  }
  static __device__ __forceinline__ int row(const NgpWgradProblem& pb, int f, int r16) {
    const int base = f == 0 ? pb.rb0 : (f == 1 ? pb.rb1 : (f == 2 ? pb.rb2 : pb.rb3));
    const int nv = f == 0 ? pb.rv0 : (f == 1 ? pb.rv1 : (f == 2 ? pb.rv2 : pb.rv3));
    return r16 < nv ? base + r16 : -1;
  }
  static __device__ __forceinline__ int row_limit(const NgpWgradProblem&, int) { return 0x7FFFFFFF; }
};
// k-parts of the fused mode per weight-gradient problem (host table order: Dense_3, Dense_2, Dense_1, Dense_4,
// Dense_0): a layer with NT < 4 dW tiles is dealt to its four waves as NT tiles x 4 / NT parts of the group's 256
// evaluations, and every part has its own row in the partial-sum buffer
constexpr int kNgpMaxParts = 4;
__host__ __device__ constexpr int ngp_wgrad_parts(int problem) { return (problem == 0 || problem == 1) ? 1 : 2; }
struct NgpPartsPlan {
  int lo[kNgpLayers], hi[kNgpLayers], parts[kNgpLayers];  // float range [lo, hi) relative to dense_offset
};
// grads[dense_off + p] += sum over workgroups and the layer's k-parts of wparts[(wg, part)][p].  One workgroup folds 32
// neighbouring parameters: thread (slice s, parameter) sums the rows of slice s of the workgroups in order, the eight
// slice sums meet in LDS and are added in order by slice 0, which owns the parameter (plain read-modify-write).  Every
// addition has a fixed place, so the Dense gradients of the fused backward are bit-reproducible.
constexpr int kNgpReduceSlices = 8;
__global__ __launch_bounds__(256) void ngp_wparts_reduce_kernel(const float* __restrict__ wparts, int n_wg, int pstride,
                                                                int n_params, NgpPartsPlan plan,
                                                                float* __restrict__ grads_dense) {
  __shared__ float part[kNgpReduceSlices][32];
  const int pl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int p = blockIdx.x * 32 + pl;
  int parts = 0;
  if (p < n_params) {
#pragma unroll
    for (int l = 0; l < kNgpLayers; ++l)
      if (p >= plan.lo[l] && p < plan.hi[l]) parts = plan.parts[l];
  }
  const int w0 = (int)((int64_t)n_wg * sl / kNgpReduceSlices), w1 = (int)((int64_t)n_wg * (sl + 1) / kNgpReduceSlices);
  float acc = 0.0f;
  for (int w = w0; w < w1; ++w)
    for (int q = 0; q < parts; ++q) acc += wparts[((int64_t)w * kNgpMaxParts + q) * pstride + p];
  part[sl][pl] = acc;
  __syncthreads();
  if (sl == 0 && parts > 0) {
    float tot = part[0][pl];
#pragma unroll
    for (int q = 1; q < kNgpReduceSlices; ++q) tot += part[q][pl];
    grads_dense[p] += tot;
  }
}

constexpr int kNgpStageStep = 8 * kFragBytes;                 // fused mode: staging bytes per wave (<= 8 fragments per layer)
constexpr int kNgpGmaxOff = kNgpLds + kWaves * kNgpStageStep;    // fused mode: 8 running maxima per lane (32 B)
constexpr int kNgpFusedLds = kNgpGmaxOff + kThreads * 32;        // 129 KiB: one persistent workgroup per CU

#ifdef LNRF_TIMELINE
__device__ unsigned long long* g_ngp_timeline_buf = nullptr;  // 8 waves x 1024 stamps (debug build only)
#endif

template <int NE, bool BWD, bool FUSED = false>
__global__ __launch_bounds__(kThreads) void ngp_mlp_kernel(
    const char* __restrict__ packed, const float* __restrict__ enc_t, const float* __restrict__ d_g, int lf,
    int64_t M, int64_t n_tiles, float* __restrict__ density, float* __restrict__ rgb,
    const float* __restrict__ g_density, const float* __restrict__ g_rgb, char* __restrict__ scratch,
    float* __restrict__ g_enc_t, float* __restrict__ lmax_parts = nullptr, NgpWgradArgs wargs = NgpWgradArgs{},
    float* __restrict__ wparts = nullptr, int pstride = 0, int64_t dense_off = 0) {
  static_assert(!FUSED || BWD, "FUSED is a backward mode");
  // backward: running max |d loss / d enc| of the rows this lane writes, 8 slots (rows 4h + 8j + {0,1} / + {2,3} are
  // the two features of levels 2h + 4j / 2h + 4j + 1): the fixed-point scale of the scatter pass.  The persistent
  // (FUSED) kernel keeps the slots in LDS (32 bytes per lane, its registers are full), reduces them once at the end
  // and stores one row of 16 per workgroup — plain stores; ngp_level_max_kernel folds the rows.  (Same-address
  // global atomics retire at ~10 ns each: 256 workgroups x 16 of them at the end of the launch measured +30 us,
  // a wave reduction per group +35 us.)
  __shared__ unsigned s_lmax[16];
  float gmax[(BWD && !FUSED) ? 8 : 1];
  if constexpr (BWD && !FUSED) {
#pragma unroll
    for (int i = 0; i < 8; ++i) gmax[i] = 0.0f;
  }
  if constexpr (FUSED) {
    float4* gm = reinterpret_cast<float4*>(smem + kNgpGmaxOff + threadIdx.x * 32);
    gm[0] = gm[1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  {
    const float* bias_g = reinterpret_cast<const float*>(packed + kNgpPackBiasOff);
    float* bias_l = reinterpret_cast<float*>(&smem[kBiasLdsOff]);
    for (int i = tid; i < kNgpBiasFloats; i += kThreads) bias_l[i] = bias_g[i];
    if (tid < 16) s_lmax[tid] = 0u;
  }
  // persistent accumulators of the fused mode.  The layers are dealt to the two halves of the workgroup so that a
  // wave carries at most three dW tiles (five would not fit next to the chain's fragments): waves 0-3 take
  // Dense_3 (slot 0), Dense_4 (slot 1), Dense_0 (slot 2); waves 4-7 take Dense_2 (slot 0), Dense_1 (slot 1).
  constexpr int kWSlots = 3;
  f32x16 wacc[FUSED ? kWSlots : 1];
  float wbias[FUSED ? kWSlots : 1];
  if constexpr (FUSED) {
#pragma unroll
    for (int i = 0; i < kWSlots; ++i) {
      wacc[i] = zero_acc();
      wbias[i] = 0.0f;
    }
  }
  const int64_t n_groups = n_tiles / kWaves;
  for (int64_t group = blockIdx.x; group < n_groups; group += FUSED ? (int64_t)gridDim.x : n_groups) {
  const int64_t tile = group * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  using Seq = NgpSeq<NE, BWD>;
  Ring<(Seq::count + kStageFrags - 1) / kStageFrags, Seq> ring;
#ifdef LNRF_TIMELINE
  ring.tl.buf = g_ngp_timeline_buf;
  ring.tl.n = 0;
  ring.tl.on = FUSED && NE == 2 && g_ngp_timeline_buf != nullptr && blockIdx.x == gridDim.x / 2 &&
               group == (int64_t)blockIdx.x + 2 * (int64_t)gridDim.x;
#endif
  LNRF_TL_STAMP(ring);  // 0: group start
  if constexpr (FUSED) __syncthreads();  // previous group's LDS reads (ring, staging) are finished
  LNRF_TL_STAMP(ring);  // 1: after the top barrier

  // encoding fragments: k slot (ks, h, j) <-> feature 16 ks + 8 (j >> 2) + 4 h + (j & 3) (row of enc_t).
  // All loads are issued before the first use: clamped addresses instead of branches, and the row stride re-read
  // through an opaque copy so that the 16 row addresses are computed here instead of being hoisted out of the group
  // loop and spilled (that version waited for 32 scratch / global round trips in a row: 12 us of a 32 us group).
  int64_t Ms = M;
  if constexpr (FUSED) asm volatile("" : "+s"(Ms));
  const int64_t mm = valid ? m : M - 1;
  float ev[NE][8];
  static_for<NE>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int feat = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
      ev[ks][j] = enc_t[(int64_t)(feat < lf ? feat : lf - 1) * Ms + mm];
    }
  });
  bf16x8 ef[NE];
  static_for<NE>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int feat = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
      ef[ks][j] = (__bf16)((valid && feat < lf) ? ev[ks][j] : 0.0f);
    }
  });
  LNRF_TL_STAMP(ring);  // 2: encoding loaded and converted
  float pd[3] = {0, 0, 0};
  float gy4[3] = {0, 0, 0}, g_dens = 0.0f;
  if (valid) {
#pragma unroll
    for (int a = 0; a < 3; ++a) pd[a] = d_g[m * 3 + a];
    if (BWD && h == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) gy4[k] = g_rgb[m * 3 + k];
      g_dens = g_density[m];
    }
  }
  LNRF_TL_STAMP(ring);  // 3: direction / output gradients requested
  __syncthreads();
  LNRF_TL_STAMP(ring);  // 4: after the barrier

  const char* wstream = packed;
  if constexpr (FUSED) asm volatile("" : "+s"(wstream));  // keep the stage addresses out of the group loop's preheader
  ring.stream = wstream;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();
  LNRF_TL_STAMP(ring);  // 5: ring prologue done

  // sinusoidal direction embedding (model.py:65-77): feature e = 8 coord + 4 is_cos + freq, slot order = e
  bf16x8 de[2];
  static_for<2>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int e = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
      float v = 0.0f;
      if (e < kNgpDembDim) {
        const int cd = e >> 3, fr = e & 3;
        const float x = cd == 0 ? pd[0] : (cd == 1 ? pd[1] : pd[2]);
        float s, co;
        sincos_pe(x * (float)(1 << fr), &s, &co);
        v = (e & 4) ? co : s;
      }
      de[ks][j] = (__bf16)v;
    }
  });

  DumpAddr dump{scratch, n_tiles, tile, c, h};
  auto dump_frag = [&](int slot, const bf16x8& f) {
    if (BWD && !FUSED) stream_store(dump.at(slot), frag_to_bits(f));  // unconditional: tiles are padded to whole workgroups
  };
  // fused mode: stage fragment f (X fragments first, then dy) of this wave's tile; weight-gradient step of layer P
  char* stage = smem + kNgpLds + wave * kNgpStageStep;
  auto stage_frag = [&](int f, const bf16x8& v) {
    *reinterpret_cast<uint4*>(stage + f * kFragBytes + dump_lane_off(f, c, h)) = frag_to_bits(v);
  };
  auto wgrad_layer = [&](auto slot_, auto half_, auto nxf_, auto nyf_) {
    constexpr int SLOT = decltype(slot_)::value, HALF = decltype(half_)::value;
    constexpr int NXF = decltype(nxf_)::value, NYF = decltype(nyf_)::value;
    constexpr int NI = NXF / 2, NO = NYF / 2, NT = NI * NO;  // tiles of this layer's dW
    constexpr int STEPS = 2 * NT;                            // 4 waves = NT tiles x (4 / NT) k-parts of 8 / (4 / NT) steps
    LNRF_TL_STAMP(ring);                                     // staged, arrive
    __syncthreads();                                         // all 8 tiles staged
    LNRF_TL_STAMP(ring);                                     // released
    if ((wave >> 2) == HALF) {
      const int w4 = wave & 3;
      const int tile_id = w4 % NT, part = w4 / NT;
      const int it = tile_id / NO, ot = tile_id % NO;
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        const char* buf = smem + kNgpLds + (part * STEPS + st) * kNgpStageStep;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const bf16x8 bfv = tr_frag(buf + (NXF + 2 * ot) * kFragBytes, lane, 0, q);
          if (it == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) wbias[SLOT] += (float)bfv[j];
          }
          const bf16x8 afv = tr_frag(buf + 2 * it * kFragBytes, lane, 0, q);
          wacc[SLOT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afv, bfv, wacc[SLOT], 0, 0, 0);
        }
      }
    }
    LNRF_TL_STAMP(ring);  // weight-gradient MFMAs issued
    __syncthreads();  // staging area free for the next layer
    LNRF_TL_STAMP(ring);  // released
  };

  bf16x8 h0[4], o16, c1[4], c2[4];
  float logit = 0.0f;
  // Dense_0 + relu
  chain_layer<ngp_fwd_base(0, NE), NE, 2>(
      ring, [&](auto o_) { return bias_acc(ngp_bias_base(0) + 32 * decltype(o_)::value, h); },
      [&](auto k_) -> bf16x8 { return ef[decltype(k_)::value]; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        h0[2 * o] = acc_to_frag<0, true>(acc);
        h0[2 * o + 1] = acc_to_frag<1, true>(acc);
      });
  // Dense_1 (linear): 16 features = rows 0..15 of the tile; density logit = feature 0
  chain_layer<ngp_fwd_base(1, NE), 4, 1>(
      ring, [&](auto) { return bias_acc(ngp_bias_base(1), h); },
      [&](auto k_) -> bf16x8 { return h0[decltype(k_)::value]; },
      [&](auto, const f32x16& acc) {
        o16 = acc_to_frag<0, false>(acc);
        logit = acc[0];
      });
  const float dens = __expf(logit);  // instant_ngp.py:49, valid on lanes h == 0
  // Dense_2 + relu on [d_emb, out]
  chain_layer<ngp_fwd_base(2, NE), 3, 2>(
      ring, [&](auto o_) { return bias_acc(ngp_bias_base(2) + 32 * decltype(o_)::value, h); },
      [&](auto k_) -> bf16x8 {
        constexpr int ks = decltype(k_)::value;
        if constexpr (ks < 2) return de[ks];
        else return o16;
      },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        c1[2 * o] = acc_to_frag<0, true>(acc);
        c1[2 * o + 1] = acc_to_frag<1, true>(acc);
      });
  // Dense_3 + relu
  chain_layer<ngp_fwd_base(3, NE), 4, 2>(
      ring, [&](auto o_) { return bias_acc(ngp_bias_base(3) + 32 * decltype(o_)::value, h); },
      [&](auto k_) -> bf16x8 { return c1[decltype(k_)::value]; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        c2[2 * o] = acc_to_frag<0, true>(acc);
        c2[2 * o + 1] = acc_to_frag<1, true>(acc);
      });
  // Dense_4 + tanh
  float y[3] = {0, 0, 0};
  chain_layer<ngp_fwd_base(4, NE), 4, 1>(
      ring, [&](auto) { return bias_acc(ngp_bias_base(4), h); },
      [&](auto k_) -> bf16x8 { return c2[decltype(k_)::value]; },
      [&](auto, const f32x16& acc) {
        y[0] = tanhf(acc[0]);
        y[1] = tanhf(acc[1]);
        y[2] = tanhf(acc[2]);
      });
  if constexpr (!BWD) {
    if (h == 0 && valid) {
      density[m] = dens;
      rgb[m * 3 + 0] = y[0];
      rgb[m * 3 + 1] = y[1];
      rgb[m * 3 + 2] = y[2];
    }
    return;
  } else {
    // ---- backward: X dumps for the weight gradients
    static_for<2>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      if constexpr (i < NE) dump_frag(kNgpXEnc + i, ef[i]);
      else dump_frag(kNgpXEnc + i, zero_frag());
    });
    static_for<4>([&](auto i_) {
      constexpr int i = decltype(i_)::value;
      dump_frag(kNgpXH0 + i, h0[i]);
      dump_frag(kNgpXC1 + i, c1[i]);
      dump_frag(kNgpXC2 + i, c2[i]);
    });
    dump_frag(kNgpXCat + 0, de[0]);
    dump_frag(kNgpXCat + 1, de[1]);
    dump_frag(kNgpXCat + 2, o16);
    dump_frag(kNgpXCat + 3, zero_frag());

    // head gradients in fp32: tanh' and exp'
    bf16x8 dy4 = zero_frag();
    dy4[0] = (__bf16)(gy4[0] * (1.0f - y[0] * y[0]));
    dy4[1] = (__bf16)(gy4[1] * (1.0f - y[1] * y[1]));
    dy4[2] = (__bf16)(gy4[2] * (1.0f - y[2] * y[2]));
    if (h != 0) dy4 = zero_frag();
    const float g_logit = h == 0 ? g_dens * dens : 0.0f;
    dump_frag(kNgpDy4, dy4);
    dump_frag(kNgpDy4 + 1, zero_frag());
    if constexpr (FUSED) {  // Dense_4: X = c2, dy = dy4
      static_for<4>([&](auto i_) { stage_frag(decltype(i_)::value, c2[decltype(i_)::value]); });
      stage_frag(4, dy4);
      stage_frag(5, zero_frag());
      wgrad_layer(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});  // Dense_4: waves 0-3, slot 1
    }

    bf16x8 dy3[4], dy2[4], dy1, dy0[4];
    // T0: Dense_4^T -> dc2, relu mask of c2
    chain_layer<ngp_bwd_base(0, NE), 1, 2>(
        ring, [&](auto) { return zero_acc(); }, [&](auto) -> bf16x8 { return dy4; },
        [&](auto o_, const f32x16& acc) {
          constexpr int o = decltype(o_)::value;
          dy3[2 * o] = masked_by<0>(acc, c2[2 * o]);
          dy3[2 * o + 1] = masked_by<1>(acc, c2[2 * o + 1]);
          dump_frag(kNgpDy3 + 2 * o, dy3[2 * o]);
          dump_frag(kNgpDy3 + 2 * o + 1, dy3[2 * o + 1]);
        });
    if constexpr (FUSED) {  // Dense_3: X = c1, dy = dy3
      static_for<4>([&](auto i_) {
        stage_frag(decltype(i_)::value, c1[decltype(i_)::value]);
        stage_frag(4 + decltype(i_)::value, dy3[decltype(i_)::value]);
      });
      wgrad_layer(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});  // Dense_3: waves 0-3, slot 0
    }
    // T1: Dense_3^T -> dc1, relu mask of c1
    chain_layer<ngp_bwd_base(1, NE), 4, 2>(
        ring, [&](auto) { return zero_acc(); }, [&](auto k_) -> bf16x8 { return dy3[decltype(k_)::value]; },
        [&](auto o_, const f32x16& acc) {
          constexpr int o = decltype(o_)::value;
          dy2[2 * o] = masked_by<0>(acc, c1[2 * o]);
          dy2[2 * o + 1] = masked_by<1>(acc, c1[2 * o + 1]);
          dump_frag(kNgpDy2 + 2 * o, dy2[2 * o]);
          dump_frag(kNgpDy2 + 2 * o + 1, dy2[2 * o + 1]);
        });
    if constexpr (FUSED) {  // Dense_2: X = [d_emb, out] (24 + 16 features in 4 fragments), dy = dy2
      stage_frag(0, de[0]);
      stage_frag(1, de[1]);
      stage_frag(2, o16);
      stage_frag(3, zero_frag());
      static_for<4>([&](auto i_) { stage_frag(4 + decltype(i_)::value, dy2[decltype(i_)::value]); });
      wgrad_layer(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});  // Dense_2: waves 4-7, slot 0
    }
    // T2: Dense_2^T restricted to the rows of `out` (d_emb has no parameters upstream); the density
    // head adds d exp(out_0) to feature 0 (lane h == 0, register 0)
    chain_layer<ngp_bwd_base(2, NE), 4, 1>(
        ring, [&](auto) { return zero_acc(); }, [&](auto k_) -> bf16x8 { return dy2[decltype(k_)::value]; },
        [&](auto, const f32x16& acc) {
          f32x16 t = acc;
          t[0] += g_logit;
          dy1 = acc_to_frag<0, false>(t);
          dump_frag(kNgpDy1, dy1);
          dump_frag(kNgpDy1 + 1, zero_frag());
        });
    if constexpr (FUSED) {  // Dense_1: X = h0, dy = dy1
      static_for<4>([&](auto i_) { stage_frag(decltype(i_)::value, h0[decltype(i_)::value]); });
      stage_frag(4, dy1);
      stage_frag(5, zero_frag());
      wgrad_layer(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});  // Dense_1: waves 4-7, slot 1
    }
    // T3: Dense_1^T -> dh0, relu mask of h0
    chain_layer<ngp_bwd_base(3, NE), 1, 2>(
        ring, [&](auto) { return zero_acc(); }, [&](auto) -> bf16x8 { return dy1; },
        [&](auto o_, const f32x16& acc) {
          constexpr int o = decltype(o_)::value;
          dy0[2 * o] = masked_by<0>(acc, h0[2 * o]);
          dy0[2 * o + 1] = masked_by<1>(acc, h0[2 * o + 1]);
          dump_frag(kNgpDy0 + 2 * o, dy0[2 * o]);
          dump_frag(kNgpDy0 + 2 * o + 1, dy0[2 * o + 1]);
        });
    if constexpr (FUSED) {  // Dense_0: X = hash-grid encoding, dy = dy0
      stage_frag(0, ef[0]);
      if constexpr (NE > 1) stage_frag(1, ef[1]);
      else stage_frag(1, zero_frag());
      static_for<4>([&](auto i_) { stage_frag(2 + decltype(i_)::value, dy0[decltype(i_)::value]); });
      wgrad_layer(std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});  // Dense_0: waves 0-3, slot 2
    }
    // T4: Dense_0^T -> d loss / d enc, feature-major fp32 for the hash-grid scatter
    chain_layer<ngp_bwd_base(4, NE), 4, 1>(
        ring, [&](auto) { return zero_acc(); }, [&](auto k_) -> bf16x8 { return dy0[decltype(k_)::value]; },
        [&](auto, const f32x16& acc) {
          float tmax[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) tmax[i] = 0.0f;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
            if (valid && row < lf) {
              g_enc_t[(int64_t)row * Ms + m] = acc[q];  // Ms: addresses formed here, not hoisted and spilled
              tmax[q >> 1] = fmaxf(tmax[q >> 1], fabsf(acc[q]));
            }
          }
          if constexpr (FUSED) {
            float4* gm = reinterpret_cast<float4*>(smem + kNgpGmaxOff + tid * 32);
            const float4 a = gm[0], b = gm[1];
            gm[0] = make_float4(fmaxf(a.x, tmax[0]), fmaxf(a.y, tmax[1]), fmaxf(a.z, tmax[2]), fmaxf(a.w, tmax[3]));
            gm[1] = make_float4(fmaxf(b.x, tmax[4]), fmaxf(b.y, tmax[5]), fmaxf(b.z, tmax[6]), fmaxf(b.w, tmax[7]));
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) gmax[i] = fmaxf(gmax[i], tmax[i]);
          }
        });
  LNRF_TL_STAMP(ring);  // group end
  }
  }  // group loop
  if constexpr (BWD) {
    if (lmax_parts) {  // kernel argument: uniform
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float v;
        if constexpr (FUSED) v = reinterpret_cast<const float*>(smem + kNgpGmaxOff + tid * 32)[i];
        else v = gmax[i];
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
        // non-negative floats order like their bit patterns
        if (c == 0 && v > 0.0f) atomicMax(&s_lmax[4 * (i >> 1) + 2 * h + (i & 1)], __float_as_uint(v));
      }
      __syncthreads();
      if (tid < 16) lmax_parts[(int64_t)blockIdx.x * 16 + tid] = __uint_as_float(s_lmax[tid]);
    }
  }
  if constexpr (FUSED) {
    // The launch's share of dW: every wave stores its tile (and bias sums) ONCE, with plain stores, into the row of its
    // (workgroup, k-part) in `wparts`; ngp_wparts_reduce_kernel folds the rows into the gradient vector.  (fp32
    // atomics straight into the gradient — 256 workgroups hitting the same 10 K addresses — cost 76 us per launch,
    // half of the coarse model's backward: same-line memory atomics retire one after the other.)
    const int colr = lane & 31, hh = lane >> 5;
    auto flush = [&](auto p_, auto slot_, auto half_, auto nxf_, auto nyf_) {
      constexpr int P = decltype(p_)::value, SLOT = decltype(slot_)::value, HALF = decltype(half_)::value;
      constexpr int NXF = decltype(nxf_)::value, NYF = decltype(nyf_)::value;
      constexpr int NO = NYF / 2, NT = (NXF / 2) * NO;
      static_assert(4 / NT == ngp_wgrad_parts(P), "k-parts per layer: host table of the reduce launch");
      if ((wave >> 2) != HALF) return;
      const NgpWgradProblem& pb = wargs.p[P];
      const int tile_id = (wave & 3) % NT, part = (wave & 3) / NT;
      const int it = tile_id / NO, ot = tile_id % NO;
      int out_idx = -1, out_dim = 1;
      int64_t w_off = 0, b_off = 0;
      NgpWgradEpi::cols(pb, ot, colr, out_idx, out_dim, w_off, b_off);
      float* __restrict__ row = wparts + ((int64_t)blockIdx.x * kNgpMaxParts + part) * pstride;
      w_off -= dense_off;  // rows hold the Dense parameters only
      b_off -= dense_off;
      if (it == 0) {
        float sacc = wbias[SLOT];
        sacc += __shfl_xor(sacc, 32, 64);
        if (hh == 0 && out_idx >= 0) row[b_off + out_idx] = sacc;
      }
      static_for<16>([&](auto q_) {
        constexpr int qq = decltype(q_)::value;
        const int r = (qq & 3) + 8 * (qq >> 2) + 4 * hh;
        const int f = 2 * it + (r >> 4);
        const int in_idx = NgpWgradEpi::row(pb, f, r & 15);
        if (out_idx >= 0 && in_idx >= 0) row[w_off + (int64_t)in_idx * out_dim + out_idx] = wacc[SLOT][qq];
      });
    };
    // (problem index of the host table, accumulator slot, workgroup half, X fragments, dy fragments)
    flush(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});  // Dense_3
    flush(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});  // Dense_4
    flush(std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});  // Dense_0
    flush(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{});  // Dense_2
    flush(std::integral_constant<int, 2>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});  // Dense_1
  }
}

// level_absmax[l] = max(level_absmax[l], max over the workgroup rows parts[n_parts][16]); one workgroup of 256 threads
__global__ __launch_bounds__(256) void ngp_level_max_kernel(const float* __restrict__ parts, int n_parts, int n_levels,
                                                            unsigned* __restrict__ level_absmax) {
  __shared__ unsigned s_max[16];
  if (threadIdx.x < 16) s_max[threadIdx.x] = 0u;
  __syncthreads();
  const int l = threadIdx.x & 15;
  float v = 0.0f;
  for (int p = threadIdx.x >> 4; p < n_parts; p += 16) v = fmaxf(v, parts[(int64_t)p * 16 + l]);
  if (v > 0.0f) atomicMax(&s_max[l], __float_as_uint(v));
  __syncthreads();
  if ((int)threadIdx.x < n_levels && threadIdx.x < 16 && s_max[threadIdx.x] != 0u)
    atomicMax(&level_absmax[threadIdx.x], s_max[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// weight gradients: the shared split-K body with InstantNGP addressing
// ---------------------------------------------------------------------------------------------
constexpr int kNgpWgSpi = 4;  // steps per barrier: 2 x 4 x 8 KiB = 64 KiB of LDS, two workgroups per CU
constexpr int kNgpWgradLds = 2 * kNgpWgSpi * 8 * kFragBytes;

__global__ __launch_bounds__(kThreads) void ngp_wgrad_kernel(NgpWgradArgs args, const char* __restrict__ scratch,
                                                             int64_t n_tiles, float* __restrict__ grads) {
  NgpWgradProblem pb = args.p[0];
#pragma unroll
  for (int i = 1; i < kNgpLayers; ++i)
    if ((int)blockIdx.x >= args.p[i].first_block) pb = args.p[i];
  switch (pb.shape) {
    case 0: wgrad_body<2, 4, 1, 8, kNgpWgSpi, NgpWgradEpi>(pb, scratch, scratch, n_tiles, grads); break;
    case 1: wgrad_body<4, 2, 2, 4, kNgpWgSpi, NgpWgradEpi>(pb, scratch, scratch, n_tiles, grads); break;
    default: wgrad_body<4, 4, 2, 4, kNgpWgSpi, NgpWgradEpi>(pb, scratch, scratch, n_tiles, grads); break;
  }
}

// ---------------------------------------------------------------------------------------------
// weight packing: flat fp32 parameters -> bf16 A-fragment stream (forward then transposed) + fp32 biases
// ---------------------------------------------------------------------------------------------
struct NgpOffsets {
  int64_t w[kNgpLayers], b[kNgpLayers];
};

// parameter feeding element (lane, j) of forward A-fragment g (stream order = consumption order); -1 = zero padding
__device__ __forceinline__ int64_t ngp_fwd_param_index(int g, int lane, int j, const NgpOffsets& off, int lf, int ne) {
  const int r = lane & 31, hh = lane >> 5;
  const int fo = 8 * (j >> 2) + 4 * hh + (j & 3);  // feature offset of k slot (hh, j) within its k-step
  int l = 0;
  for (int i = 1; i < kNgpLayers; ++i)
    if (g >= ngp_fwd_base(i, ne)) l = i;
  const int loc = g - ngp_fwd_base(l, ne), nk = ngp_fwd_nk(l, ne);
  const int o = loc / nk, ks = loc % nk;
  const int row = 32 * o + r, od = ngp_out_dim(l);
  int k = -1;
  if (l == 0) k = (16 * ks + fo < lf) ? 16 * ks + fo : -1;
  else if (l == 2) k = ks == 0 ? fo : (ks == 1 ? (fo < 8 ? 16 + fo : -1) : kNgpDembDim + fo);
  else k = 16 * ks + fo;
  return (row < od && k >= 0) ? off.w[l] + (int64_t)k * od + row : -1;
}

// ---------------------------------------------------------------------------------------------
// Split-precision forward ("bf16x3", the render / evaluation path; instant_ngp.py:38-54 is fp32 in the reference): every
// fp32 operand — weight, hash-grid feature, direction embedding, activation — is a bf16 pair (hi, lo) with hi + lo equal
// to the value to 16 significant bits and every product is lo*hi + hi*lo + hi*hi on the bf16 MFMA with fp32 accumulation,
// as nerf_fwd_split_kernel does for NeRFModel.  The whole stream ([hi, lo] per forward fragment: at most 52 KiB) is
// staged in LDS once per persistent workgroup.
// ---------------------------------------------------------------------------------------------
constexpr int kNgpSplitMaxFrags = 2 * 26;
constexpr int kNgpSplitBiasOff = kNgpSplitMaxFrags * kFragBytes;
constexpr int kNgpSplitBytes = kNgpSplitBiasOff + kNgpBiasFloats * 4;

__global__ void ngp_pack_split_kernel(const float* __restrict__ params, NgpOffsets off, int lf, int ne,
                                      char* __restrict__ packed) {
  const int total_w = kNgpSplitMaxFrags * 512;
  const int total = total_w + kNgpBiasFloats;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    if (e < total_w) {
      const int gg = e >> 9, lane = (e >> 3) & 63, j = e & 7;
      const int g = gg >> 1;
      const int64_t idx = g < ngp_fwd_count(ne) ? ngp_fwd_param_index(g, lane, j, off, lf, ne) : -1;
      const float w = idx >= 0 ? params[idx] : 0.0f;
      const __bf16 hi = (__bf16)w;
      reinterpret_cast<__bf16*>(packed)[e] = (gg & 1) ? (__bf16)(w - (float)hi) : hi;
    } else {
      const int i = e - total_w;
      int l = 0;
      for (int k = 1; k < kNgpLayers; ++k)
        if (i >= ngp_bias_base(k)) l = k;
      const int loc = i - ngp_bias_base(l);
      reinterpret_cast<float*>(packed + kNgpSplitBiasOff)[i] = loc < ngp_out_dim(l) ? params[off.b[l] + loc] : 0.0f;
    }
  }
}

__device__ __forceinline__ void ngp_split(float v, bf16x8& hi, bf16x8& lo, int j) {
  const __bf16 hb = (__bf16)v;
  hi[j] = hb;
  lo[j] = (__bf16)(v - (float)hb);
}
template <int S, bool RELU>
__device__ __forceinline__ void ngp_acc_split(const f32x16& acc, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = acc[8 * S + j];
    if (RELU) v = __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff());
    ngp_split(v, hi, lo, j);
  }
}

template <int NE>
__global__ __launch_bounds__(kThreads) void ngp_mlp_fwd_split_kernel(
    const char* __restrict__ packed3, const float* __restrict__ enc_t, const float* __restrict__ d_g, int lf, int64_t M,
    int64_t n_tiles, float* __restrict__ density, float* __restrict__ rgb) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  for (int i = tid; i < kNgpSplitBytes / 16; i += kThreads)
    reinterpret_cast<uint4*>(smem)[i] = reinterpret_cast<const uint4*>(packed3)[i];
  __syncthreads();
  const float* bias_l = reinterpret_cast<const float*>(smem + kNgpSplitBiasOff);
  // the per-lane LDS offset is made opaque once per group: the stream does not change, so the compiler would otherwise
  // hoist all 52 fragment reads (208 registers) out of the group loop and spill them
  int lds_lane = lane * 16;
  auto afrag = [&](int g, int part) -> bf16x8 {
    return bits_to_frag(*reinterpret_cast<const uint4*>(smem + (2 * g + part) * kFragBytes + lds_lane));
  };
  auto bias_tile = [&](int row0) -> f32x16 {  // accumulator rows (q & 3) + 8 (q >> 2) + 4 h of the 32-row tile at row0
    f32x16 acc;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 v = *reinterpret_cast<const float4*>(bias_l + row0 + 4 * h + 8 * gq);
      acc[4 * gq + 0] = v.x; acc[4 * gq + 1] = v.y; acc[4 * gq + 2] = v.z; acc[4 * gq + 3] = v.w;
    }
    return acc;
  };
  // one layer: out tiles x k-steps, three MFMAs per product (small terms first)
  auto layer = [&](auto g0_, auto nk_, auto no_, int bias0, auto&& bhi, auto&& blo, auto&& epi) {
    constexpr int G0 = decltype(g0_)::value, NK = decltype(nk_)::value, NO = decltype(no_)::value;
    static_for<NO>([&](auto o_) {
      constexpr int o = decltype(o_)::value;
      f32x16 acc = bias_tile(bias0 + 32 * o);
      static_for<NK>([&](auto k_) {
        constexpr int ks = decltype(k_)::value;
        const bf16x8 ahi = afrag(G0 + o * NK + ks, 0), alo = afrag(G0 + o * NK + ks, 1);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi(k_), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo(k_), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi(k_), acc, 0, 0, 0);
      });
      epi(o_, acc);
    });
  };
  const int64_t n_groups = n_tiles / kWaves;
  for (int64_t group = blockIdx.x; group < n_groups; group += gridDim.x) {
    asm volatile("" : "+v"(lds_lane));
    const int64_t tile = group * kWaves + wave;
    const int64_t m = tile * kTileCols + c;
    const bool valid = m < M;
    const int64_t mm = valid ? m : M - 1;
    bf16x8 ef_hi[NE], ef_lo[NE];
    static_for<NE>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int feat = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
        const float v = (valid && feat < lf) ? enc_t[(int64_t)(feat < lf ? feat : lf - 1) * M + mm] : 0.0f;
        ngp_split(v, ef_hi[ks], ef_lo[ks], j);
      }
    });
    float pd[3] = {0, 0, 0};
    if (valid) {
#pragma unroll
      for (int a = 0; a < 3; ++a) pd[a] = d_g[m * 3 + a];
    }
    bf16x8 de_hi[2], de_lo[2];  // sinusoidal direction embedding (model.py:65-77): feature e = 8 coord + 4 is_cos + freq
    static_for<2>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
        float v = 0.0f;
        if (e < kNgpDembDim) {
          const int cd = e >> 3, fr = e & 3;
          const float x = cd == 0 ? pd[0] : (cd == 1 ? pd[1] : pd[2]);
          float sn, co;
          sincos_pe(x * (float)(1 << fr), &sn, &co);
          v = (e & 4) ? co : sn;
        }
        ngp_split(v, de_hi[ks], de_lo[ks], j);
      }
    });
    bf16x8 h0h[4], h0l[4], o16h, o16l, c1h[4], c1l[4], c2h[4], c2l[4];
    float logit = 0.0f, y[3] = {0, 0, 0};
    using I = std::integral_constant<int, 0>;
    (void)sizeof(I);
    layer(std::integral_constant<int, ngp_fwd_base(0, NE)>{}, std::integral_constant<int, NE>{}, std::integral_constant<int, 2>{},
          ngp_bias_base(0), [&](auto k_) -> bf16x8 { return ef_hi[decltype(k_)::value]; },
          [&](auto k_) -> bf16x8 { return ef_lo[decltype(k_)::value]; },
          [&](auto o_, const f32x16& acc) {
            constexpr int o = decltype(o_)::value;
            ngp_acc_split<0, true>(acc, h0h[2 * o], h0l[2 * o]);
            ngp_acc_split<1, true>(acc, h0h[2 * o + 1], h0l[2 * o + 1]);
          });
    layer(std::integral_constant<int, ngp_fwd_base(1, NE)>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{},
          ngp_bias_base(1), [&](auto k_) -> bf16x8 { return h0h[decltype(k_)::value]; },
          [&](auto k_) -> bf16x8 { return h0l[decltype(k_)::value]; },
          [&](auto, const f32x16& acc) {
            ngp_acc_split<0, false>(acc, o16h, o16l);
            logit = acc[0];
          });
    layer(std::integral_constant<int, ngp_fwd_base(2, NE)>{}, std::integral_constant<int, 3>{}, std::integral_constant<int, 2>{},
          ngp_bias_base(2),
          [&](auto k_) -> bf16x8 {
            constexpr int ks = decltype(k_)::value;
            if constexpr (ks < 2) return de_hi[ks];
            else return o16h;
          },
          [&](auto k_) -> bf16x8 {
            constexpr int ks = decltype(k_)::value;
            if constexpr (ks < 2) return de_lo[ks];
            else return o16l;
          },
          [&](auto o_, const f32x16& acc) {
            constexpr int o = decltype(o_)::value;
            ngp_acc_split<0, true>(acc, c1h[2 * o], c1l[2 * o]);
            ngp_acc_split<1, true>(acc, c1h[2 * o + 1], c1l[2 * o + 1]);
          });
    layer(std::integral_constant<int, ngp_fwd_base(3, NE)>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{},
          ngp_bias_base(3), [&](auto k_) -> bf16x8 { return c1h[decltype(k_)::value]; },
          [&](auto k_) -> bf16x8 { return c1l[decltype(k_)::value]; },
          [&](auto o_, const f32x16& acc) {
            constexpr int o = decltype(o_)::value;
            ngp_acc_split<0, true>(acc, c2h[2 * o], c2l[2 * o]);
            ngp_acc_split<1, true>(acc, c2h[2 * o + 1], c2l[2 * o + 1]);
          });
    layer(std::integral_constant<int, ngp_fwd_base(4, NE)>{}, std::integral_constant<int, 4>{}, std::integral_constant<int, 1>{},
          ngp_bias_base(4), [&](auto k_) -> bf16x8 { return c2h[decltype(k_)::value]; },
          [&](auto k_) -> bf16x8 { return c2l[decltype(k_)::value]; },
          [&](auto, const f32x16& acc) {
            y[0] = tanhf(acc[0]);
            y[1] = tanhf(acc[1]);
            y[2] = tanhf(acc[2]);
          });
    if (h == 0 && valid) {
      density[m] = expf(logit);  // instant_ngp.py:49
      rgb[m * 3 + 0] = y[0];
      rgb[m * 3 + 1] = y[1];
      rgb[m * 3 + 2] = y[2];
    }
  }
}

__global__ void ngp_pack_kernel(const float* __restrict__ params, NgpOffsets off, int lf, int ne,
                                char* __restrict__ packed) {
  const int total_frag_elems = kNgpStreamFrags * 512;
  const int total = total_frag_elems + kNgpBiasFloats;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    if (e < total_frag_elems) {
      const int g = e >> 9, lane = (e >> 3) & 63, j = e & 7;
      const int r = lane & 31, hh = lane >> 5;
      const int fo = 8 * (j >> 2) + 4 * hh + (j & 3);  // feature offset of k slot (hh, j) within its k-step
      int64_t idx = -1;
      const int nfwd = ngp_fwd_count(ne);
      if (g < nfwd) {
        idx = ngp_fwd_param_index(g, lane, j, off, lf, ne);
      } else if (g < ngp_total_count(ne)) {
        int t = 0;
        for (int i = 1; i < kNgpLayers; ++i)
          if (g >= ngp_bwd_base(i, ne)) t = i;
        const int loc = g - ngp_bwd_base(t, ne), nk = ngp_bwd_nk(t);
        const int o = loc / nk, ks = loc % nk;
        const int l = 4 - t, od = ngp_out_dim(l);
        // A[row = input feature of Dense_l][k = output feature of Dense_l] = W_l[row][k]
        int row = 32 * o + r;
        const int k = 16 * ks + fo;
        if (l == 2) row = r < kNgpDensityDim ? kNgpDembDim + r : -1;  // only the rows fed by `out`
        else if (l == 0) row = r < lf ? r : -1;
        else if (row >= kNgpHidden) row = -1;
        if (row >= 0 && k < od) idx = off.w[l] + (int64_t)row * od + k;
      }
      reinterpret_cast<__bf16*>(packed)[e] = (__bf16)(idx >= 0 ? params[idx] : 0.0f);
    } else {
      const int i = e - total_frag_elems;
      int l = 0;
      for (int k = 1; k < kNgpLayers; ++k)
        if (i >= ngp_bias_base(k)) l = k;
      const int loc = i - ngp_bias_base(l);
      reinterpret_cast<float*>(packed + kNgpPackBiasOff)[i] = loc < ngp_out_dim(l) ? params[off.b[l] + loc] : 0.0f;
    }
  }
}

}  // namespace lnrf

using namespace lnrf;

static bool ngp_supported(const lnrf_ngp_mlp_desc* d) {
  return d && d->hidden_dim == kNgpHidden && d->density_dim == kNgpDensityDim && d->density_layers == 1 &&
         d->color_layers == 2 && d->d_freqs == 4 && d->enc_dim >= 1 && d->enc_dim <= 32;
}
// padded to whole workgroups (8 waves) so that the dump stores need no branch (see tiles_for in nerf_mlp.hip)
static inline int64_t ngp_tiles(int64_t m) { return ((m + kTileCols - 1) / kTileCols + kWaves - 1) / kWaves * kWaves; }
static NgpOffsets ngp_offsets(const lnrf_ngp_mlp_desc* d) {
  NgpOffsets o;
  int64_t off = d->dense_offset;
  int fan[kNgpLayers] = {d->enc_dim, kNgpHidden, kNgpDembDim + kNgpDensityDim, kNgpHidden, kNgpHidden};
  for (int l = 0; l < kNgpLayers; ++l) {
    o.w[l] = off;
    off += (int64_t)fan[l] * ngp_out_dim(l);
    o.b[l] = off;
    off += ngp_out_dim(l);
  }
  return o;
}
#define NGP_REQUIRE_SUPPORTED(fn)                                                                          \
  if (!ngp_supported(desc)) {                                                                              \
    set_error(fn ": only InstantNGPModel{hidden 64, density_dim 16, 1 density layer, 2 color layers, "    \
                 "d_freqs 4, L*F <= 32} is fused");                                                        \
    return LNRF_ERR_UNSUPPORTED;                                                                           \
  }

template <class K>
static int ngp_ensure_lds(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(max dynamic LDS)");
  return LNRF_OK;
}

extern "C" int64_t lnrf_ngp_mlp_packed_bytes(const lnrf_ngp_mlp_desc* desc) {
  return ngp_supported(desc) ? kNgpPackBytes : -1;
}
static int64_t ngp_lmax_bytes(int64_t n_tiles) { return (n_tiles + kWaves - 1) / kWaves * 16 * (int64_t)sizeof(float); }

// The two-launch backward (fragment dumps + split-K weight-gradient kernel with fp32 atomics) is the A/B partner of the
// persistent backward; only experiment builds (common.h) can select it, with LNRF_NGP_WGRAD=split.
static bool ngp_fused_wgrad_enabled() {
  static const bool on = !exp_env_is("LNRF_NGP_WGRAD", 's');
  return on;
}
// scratch = [partial dW rows of the persistent backward | (experiment builds: fragment dumps of the two-launch path)]
// then one row of 16 per-level maxima per workgroup.  At most kNgpMaxPersistent workgroups form partial rows.
constexpr int kNgpMaxPersistent = 512;
static int ngp_dense_params(const lnrf_ngp_mlp_desc* d) {
  const NgpOffsets o = ngp_offsets(d);
  return (int)(o.b[kNgpLayers - 1] + ngp_out_dim(kNgpLayers - 1) - d->dense_offset);
}
static int ngp_pstride(const lnrf_ngp_mlp_desc* d) { return (ngp_dense_params(d) + 63) / 64 * 64; }
static int64_t ngp_lmax_off(const lnrf_ngp_mlp_desc* d, int64_t n_tiles) {
  int64_t rows = n_tiles / kWaves;
  if (rows > kNgpMaxPersistent) rows = kNgpMaxPersistent;
  int64_t bytes = rows * kNgpMaxParts * ngp_pstride(d) * (int64_t)sizeof(float);
  if (!ngp_fused_wgrad_enabled()) {
    const int64_t dumps = (int64_t)kNgpSlots * n_tiles * kFragBytes;
    if (dumps > bytes) bytes = dumps;
  }
  return (bytes + 255) / 256 * 256;
}

extern "C" int64_t lnrf_ngp_mlp_scratch_bytes(const lnrf_ngp_mlp_desc* desc, int64_t m) {
  return ngp_supported(desc) ? ngp_lmax_off(desc, ngp_tiles(m)) + ngp_lmax_bytes(ngp_tiles(m)) : -1;
}

extern "C" int lnrf_ngp_mlp_pack(const lnrf_ngp_mlp_desc* desc, const float* params, void* packed,
                                 lnrf_stream_t stream) {
  NGP_REQUIRE_SUPPORTED("lnrf_ngp_mlp_pack");
  LNRF_CHECK_ARG(params && packed, "null pointer");
  LNRF_CHECK_ARG(desc->dense_offset >= 0, "bad dense_offset");
  const int ne = desc->enc_dim <= 16 ? 1 : 2;
  hipLaunchKernelGGL(ngp_pack_kernel, dim3(64), dim3(256), 0, as_stream(stream), params, ngp_offsets(desc),
                     (int)desc->enc_dim, ne, (char*)packed);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int64_t lnrf_ngp_mlp_packed_split_bytes(const lnrf_ngp_mlp_desc* desc) {
  return ngp_supported(desc) ? kNgpSplitBytes : -1;
}

extern "C" int lnrf_ngp_mlp_pack_split(const lnrf_ngp_mlp_desc* desc, const float* params, void* packed_split,
                                       lnrf_stream_t stream) {
  NGP_REQUIRE_SUPPORTED("lnrf_ngp_mlp_pack_split");
  LNRF_CHECK_ARG(params && packed_split, "null pointer");
  LNRF_CHECK_ARG(desc->dense_offset >= 0, "bad dense_offset");
  const int ne = desc->enc_dim <= 16 ? 1 : 2;
  hipLaunchKernelGGL(ngp_pack_split_kernel, dim3(64), dim3(256), 0, as_stream(stream), params, ngp_offsets(desc),
                     (int)desc->enc_dim, ne, (char*)packed_split);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_ngp_mlp_fwd_split(const lnrf_ngp_mlp_desc* desc, const void* packed_split, const float* enc_t,
                                      const float* d, int64_t m, float* density, float* rgb, lnrf_stream_t stream) {
  NGP_REQUIRE_SUPPORTED("lnrf_ngp_mlp_fwd_split");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  LNRF_CHECK_ARG(packed_split && enc_t && d && density && rgb, "null pointer");
  const int64_t n_tiles = ngp_tiles(m);
  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return hip_fail(e, "hipDeviceGetAttribute(multiprocessor count)");
  int64_t nb = n_tiles / kWaves;
  if (nb > 2 * (int64_t)cus) nb = 2 * (int64_t)cus;  // persistent: the stream is staged in LDS once per workgroup
  hipStream_t st = as_stream(stream);
  int rc;
  if (desc->enc_dim <= 16) {
    rc = ngp_ensure_lds(ngp_mlp_fwd_split_kernel<1>, kNgpSplitBytes);
    if (rc) return rc;
    hipLaunchKernelGGL((ngp_mlp_fwd_split_kernel<1>), dim3((unsigned)nb), dim3(kThreads), kNgpSplitBytes, st,
                       (const char*)packed_split, enc_t, d, (int)desc->enc_dim, m, n_tiles, density, rgb);
  } else {
    rc = ngp_ensure_lds(ngp_mlp_fwd_split_kernel<2>, kNgpSplitBytes);
    if (rc) return rc;
    hipLaunchKernelGGL((ngp_mlp_fwd_split_kernel<2>), dim3((unsigned)nb), dim3(kThreads), kNgpSplitBytes, st,
                       (const char*)packed_split, enc_t, d, (int)desc->enc_dim, m, n_tiles, density, rgb);
  }
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_ngp_mlp_fwd(const lnrf_ngp_mlp_desc* desc, const void* packed, const float* enc_t,
                                const float* d, int64_t m, float* density, float* rgb, lnrf_stream_t stream) {
  NGP_REQUIRE_SUPPORTED("lnrf_ngp_mlp_fwd");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  LNRF_CHECK_ARG(packed && enc_t && d && density && rgb, "null pointer");
  const int64_t n_tiles = ngp_tiles(m);
  const dim3 grid((unsigned)((n_tiles + kWaves - 1) / kWaves)), block(kThreads);
  hipStream_t st = as_stream(stream);
  int rc;
  if (desc->enc_dim <= 16) {
    rc = ngp_ensure_lds(ngp_mlp_kernel<1, false>, kNgpLds);
    if (rc) return rc;
    hipLaunchKernelGGL((ngp_mlp_kernel<1, false>), grid, block, kNgpLds, st, (const char*)packed, enc_t, d,
                       (int)desc->enc_dim, m, n_tiles, density, rgb, nullptr, nullptr, nullptr, nullptr);
  } else {
    rc = ngp_ensure_lds(ngp_mlp_kernel<2, false>, kNgpLds);
    if (rc) return rc;
    hipLaunchKernelGGL((ngp_mlp_kernel<2, false>), grid, block, kNgpLds, st, (const char*)packed, enc_t, d,
                       (int)desc->enc_dim, m, n_tiles, density, rgb, nullptr, nullptr, nullptr, nullptr);
  }
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_ngp_mlp_bwd(const lnrf_ngp_mlp_desc* desc, const void* packed, const float* enc_t,
                                const float* d, const float* g_density, const float* g_rgb, int64_t m,
                                void* scratch, float* g_enc_t, float* level_absmax, float* grads,
                                lnrf_stream_t stream) {
  NGP_REQUIRE_SUPPORTED("lnrf_ngp_mlp_bwd");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  LNRF_CHECK_ARG(packed && enc_t && d && g_density && g_rgb && scratch && g_enc_t && grads, "null pointer");
  const int64_t n_tiles = ngp_tiles(m);
  const dim3 grid((unsigned)((n_tiles + kWaves - 1) / kWaves)), block(kThreads);
  hipStream_t st = as_stream(stream);
  int rc;
  // per-workgroup rows of level maxima live behind the dumps in the scratch buffer
  float* lmax_parts = level_absmax ? reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) +
                                                              ngp_lmax_off(desc, n_tiles))
                                   : nullptr;
  const int n_levels = desc->enc_dim / 2;
  // weight-gradient problems (five Dense layers)
  const NgpOffsets off = ngp_offsets(desc);
  const int lf = desc->enc_dim;
  NgpWgradArgs a;
  int first = 0;
  auto add = [&](int i, int shape, int xs, int ys, int layer, int rb0, int rv0, int rb1, int rv1, int rb2, int rv2,
                 int rb3, int rv3) {
    NgpWgradProblem p;
    p.shape = shape; p.x_slot0 = xs; p.y_slot0 = ys; p.do_bias = 1;
    p.out_dim = ngp_out_dim(layer);
    p.w_lo = (unsigned)(off.w[layer] & 0xFFFFFFFFll); p.w_hi = (unsigned)(off.w[layer] >> 32);
    p.b_lo = (unsigned)(off.b[layer] & 0xFFFFFFFFll); p.b_hi = (unsigned)(off.b[layer] >> 32);
    p.rb0 = rb0; p.rv0 = rv0; p.rb1 = rb1; p.rv1 = rv1;
    p.rb2 = rb2; p.rv2 = rv2; p.rb3 = rb3; p.rv3 = rv3;
    int64_t nb = 102;  // 5 problems x 102 = 510 workgroups = two per CU
    const int64_t max_nb = (n_tiles + 2 * kNgpWgSpi - 1) / (2 * kNgpWgSpi);
    if (nb > max_nb) nb = max_nb;
    p.first_block = first;
    p.n_blocks = (int)nb;
    first += (int)nb;
    a.p[i] = p;
  };
  add(0, 2, kNgpXC1, kNgpDy3, 3, 0, 16, 16, 16, 32, 16, 48, 16);
  add(1, 2, kNgpXCat, kNgpDy2, 2, 0, 16, 16, 8, kNgpDembDim, 16, 0, 0);
  add(2, 1, kNgpXH0, kNgpDy1, 1, 0, 16, 16, 16, 32, 16, 48, 16);
  add(3, 1, kNgpXC2, kNgpDy4, 4, 0, 16, 16, 16, 32, 16, 48, 16);
  add(4, 0, kNgpXEnc, kNgpDy0, 0, 0, lf < 16 ? lf : 16, 16, lf > 16 ? lf - 16 : 0, 0, 0, 0, 0);
  if (ngp_fused_wgrad_enabled()) {
    // one persistent workgroup per CU forms the weight gradients itself (no dumps, no second launch)
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return hip_fail(e, "hipDeviceGetAttribute(multiprocessor count)");
    int64_t nb = n_tiles / kWaves;
    if (nb > cus) nb = cus;
    if (nb > kNgpMaxPersistent) nb = kNgpMaxPersistent;
    const dim3 pgrid((unsigned)nb);
    // partial dW rows (workgroup, k-part) live where the two-launch path keeps its dumps: nb <= n_tiles / 8 rows of
    // 4 x pstride floats against 8 x 34 KiB of dump room per workgroup
    NgpPartsPlan plan;
    for (int i = 0; i < kNgpLayers; ++i) {
      const int layer = i == 0 ? 3 : (i == 1 ? 2 : (i == 2 ? 1 : (i == 3 ? 4 : 0)));  // problem i of the table above
      plan.lo[i] = (int)(off.w[layer] - desc->dense_offset);
      plan.hi[i] = (int)(off.b[layer] - desc->dense_offset) + ngp_out_dim(layer);
      plan.parts[i] = ngp_wgrad_parts(i);
    }
    const int n_params = ngp_dense_params(desc);
    const int pstride = ngp_pstride(desc);
    float* wparts = reinterpret_cast<float*>(scratch);
    if (desc->enc_dim <= 16) {
      rc = ngp_ensure_lds(ngp_mlp_kernel<1, true, true>, kNgpFusedLds);
      if (rc) return rc;
      hipLaunchKernelGGL((ngp_mlp_kernel<1, true, true>), pgrid, block, kNgpFusedLds, st, (const char*)packed, enc_t, d,
                         (int)desc->enc_dim, m, n_tiles, nullptr, nullptr, g_density, g_rgb, nullptr, g_enc_t,
                         lmax_parts, a, wparts, pstride, (int64_t)desc->dense_offset);
    } else {
      rc = ngp_ensure_lds(ngp_mlp_kernel<2, true, true>, kNgpFusedLds);
      if (rc) return rc;
      hipLaunchKernelGGL((ngp_mlp_kernel<2, true, true>), pgrid, block, kNgpFusedLds, st, (const char*)packed, enc_t, d,
                         (int)desc->enc_dim, m, n_tiles, nullptr, nullptr, g_density, g_rgb, nullptr, g_enc_t,
                         lmax_parts, a, wparts, pstride, (int64_t)desc->dense_offset);
    }
    LNRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(ngp_wparts_reduce_kernel, dim3((unsigned)((n_params + 31) / 32)), dim3(256), 0, st, wparts,
                       (int)nb, pstride, n_params, plan, grads + desc->dense_offset);
    LNRF_LAUNCH_CHECK();
    if (lmax_parts) {
      hipLaunchKernelGGL(ngp_level_max_kernel, dim3(1), dim3(256), 0, st, lmax_parts, (int)nb, n_levels,
                         reinterpret_cast<unsigned*>(level_absmax));
      LNRF_LAUNCH_CHECK();
    }
    return LNRF_OK;
  }
  if (desc->enc_dim <= 16) {
    rc = ngp_ensure_lds(ngp_mlp_kernel<1, true>, kNgpLds);
    if (rc) return rc;
    hipLaunchKernelGGL((ngp_mlp_kernel<1, true>), grid, block, kNgpLds, st, (const char*)packed, enc_t, d,
                       (int)desc->enc_dim, m, n_tiles, nullptr, nullptr, g_density, g_rgb, (char*)scratch, g_enc_t,
                       lmax_parts);
  } else {
    rc = ngp_ensure_lds(ngp_mlp_kernel<2, true>, kNgpLds);
    if (rc) return rc;
    hipLaunchKernelGGL((ngp_mlp_kernel<2, true>), grid, block, kNgpLds, st, (const char*)packed, enc_t, d,
                       (int)desc->enc_dim, m, n_tiles, nullptr, nullptr, g_density, g_rgb, (char*)scratch, g_enc_t,
                       lmax_parts);
  }
  LNRF_LAUNCH_CHECK();
  if (lmax_parts) {
    hipLaunchKernelGGL(ngp_level_max_kernel, dim3(1), dim3(256), 0, st, lmax_parts, (int)grid.x, n_levels,
                       reinterpret_cast<unsigned*>(level_absmax));
    LNRF_LAUNCH_CHECK();
  }

  rc = ngp_ensure_lds(ngp_wgrad_kernel, kNgpWgradLds);
  if (rc) return rc;
  hipLaunchKernelGGL(ngp_wgrad_kernel, dim3((unsigned)first), dim3(kThreads), kNgpWgradLds, st, a,
                     (const char*)scratch, n_tiles, grads);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

#ifdef LNRF_TIMELINE
// debug library only (tools/build_timeline.sh): where the stamped workgroup of the fused backward writes its stamps
extern "C" int lnrf_debug_set_ngp_timeline(void* buf) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(lnrf::g_ngp_timeline_buf), &buf, sizeof(buf));
  return e == hipSuccess ? LNRF_OK : lnrf::hip_fail(e, "hipMemcpyToSymbol(g_ngp_timeline_buf)");
}
#endif
