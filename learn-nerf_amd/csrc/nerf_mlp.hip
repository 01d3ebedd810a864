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
#include <utility>

#include "common.h"
#include "fast_math.h"
#include "nerf_layout.h"

namespace lnrf {
using namespace nl;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));


template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int kWaves = 8;            // waves per workgroup
constexpr int kThreads = kWaves * 64;
constexpr int kTileCols = 32;        // evaluations per wave
constexpr int kStageBytes = kStageFrags * kFragBytes;  // 16 KiB
constexpr int kSlots = 3;            // LDS stages: being read, published-next, being written
constexpr int kRingBytes = kSlots * kStageBytes;
constexpr int kBiasLdsOff = kRingBytes;
constexpr int kFusedLds = kRingBytes + round_up(kBiasFloats * 4, 1024);
constexpr int kFragAhead = 4;        // A-fragments read from LDS ahead of the MFMA that uses them

extern __shared__ __attribute__((aligned(16))) char smem[];

__device__ __forceinline__ bf16x8 bits_to_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ uint4 frag_to_bits(bf16x8 v) { return __builtin_bit_cast(uint4, v); }
__device__ __forceinline__ bf16x8 zero_frag() { return bits_to_frag(make_uint4(0, 0, 0, 0)); }

struct FwdSeq {
  static constexpr int count = kFwdUsed;
  static constexpr int at(int c) { return fwd_seq(c); }
};
struct BwdSeq {
  static constexpr int count = kBwdUsed;
  static constexpr int at(int c) { return bwd_seq(c); }
};

// The weight ring.  A stage is 16 fragments (16 KiB) shared by the 8 waves; every wave moves 2 of
// them.  Staging is global -> VGPR -> LDS (not LDS-DMA: hipcc drains vmcnt(0) before any ds_read
// while an LDS-DMA is pending, which serialises the pipeline).  Timeline at the barrier that opens
// stage T: stage T+1 is already in LDS and becomes visible (it was written after barrier T-1), stage
// T+2 is written from registers into the slot freed by stage T-1, stage T+4 is requested from L2.
// Because stage T+1 is visible during stage T, the per-wave FIFO of A-fragments (kFragAhead LDS reads
// in flight) runs continuously across stage boundaries.
template <int NSTAGES, class SEQ>
struct Ring {
  const char* stream;  // global, NSTAGES * 16 KiB, fragment order
  int wave, lane;
  uint4 r[2][2];
  bf16x8 fifo[kFragAhead];

  template <int T>
  __device__ __forceinline__ void load() {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = wave + kWaves * q;
      r[T & 1][q] = *reinterpret_cast<const uint4*>(stream + ((int64_t)T * kStageFrags + f) * kFragBytes +
                                                    lane * 16);
    }
  }
  template <int T>
  __device__ __forceinline__ void write() {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = wave + kWaves * q;
      *reinterpret_cast<uint4*>(&smem[(T % kSlots) * kStageBytes + f * kFragBytes + lane * 16]) = r[T & 1][q];
    }
  }
  template <int G>
  __device__ __forceinline__ bf16x8 read_lds() const {
    constexpr int off = ((G / kStageFrags) % kSlots) * kStageBytes + (G % kStageFrags) * kFragBytes;
    return bits_to_frag(*reinterpret_cast<const uint4*>(&smem[off + lane * 16]));
  }
  // loads stages 0..3, publishes stages 0 and 1, fills the fragment FIFO
  __device__ __forceinline__ void prologue() {
    load<0>();
    if constexpr (NSTAGES > 1) load<1>();
    write<0>();
    if constexpr (NSTAGES > 1) write<1>();
    if constexpr (NSTAGES > 2) load<2>();
    if constexpr (NSTAGES > 3) load<3>();
    __syncthreads();
    if constexpr (NSTAGES > 2) write<2>();  // the work of the (implicit) barrier that opens stage 0
    if constexpr (NSTAGES > 4) load<4>();
    static_for<kFragAhead>([&](auto i) {
      constexpr int c = decltype(i)::value;
      if constexpr (c < SEQ::count) fifo[c] = read_lds<SEQ::at(c)>();
    });
  }
  // barrier that opens stage T (T >= 1): frees the slot of stage T-1, publishes stage T+1
  template <int T>
  __device__ __forceinline__ void advance() {
    __syncthreads();
    if constexpr (T + 2 < NSTAGES) write<T + 2>();
    if constexpr (T + 4 < NSTAGES) load<T + 4>();
  }
  // fragment of consumption index C (and request the one kFragAhead later)
  template <int C>
  __device__ __forceinline__ bf16x8 next() {
    constexpr int g = SEQ::at(C);
    if constexpr (g % kStageFrags == 0 && g > 0) advance<g / kStageFrags>();
    const bf16x8 a = fifo[C % kFragAhead];
    if constexpr (C + kFragAhead < SEQ::count) fifo[C % kFragAhead] = read_lds<SEQ::at(C + kFragAhead)>();
    return a;
  }
};

// accumulator initialised with the fp32 bias of rows 32*o.. (LDS block, broadcast reads)
__device__ __forceinline__ f32x16 bias_acc(int bias_row0, int h) {
  f32x16 acc;
  const float* b = reinterpret_cast<const float*>(&smem[kBiasLdsOff]) + bias_row0 + 4 * h;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(b + 8 * g);
    acc[4 * g + 0] = v.x;
    acc[4 * g + 1] = v.y;
    acc[4 * g + 2] = v.z;
    acc[4 * g + 3] = v.w;
  }
  return acc;
}
__device__ __forceinline__ f32x16 zero_acc() {
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
  return acc;
}

// registers 8s..8s+7 of an accumulator tile -> B-frag of k-step s of the next layer
template <int S, bool RELU>
__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& acc) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = acc[8 * S + j];
    if (RELU) v = __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff());  // max(v, 0) in one VALU op
    f[j] = (__bf16)v;
  }
  return f;
}

// ReLU mask of one out-tile from its two bf16 output fragments: bit 8s + j set <=> element j of
// fragment s is non-zero (ReLU output > 0), i.e. accumulator register 8s + j passed the ReLU
__device__ __forceinline__ unsigned relu_bits(const bf16x8& f0, const bf16x8& f1) {
  const uint4 a = frag_to_bits(f0), b = frag_to_bits(f1);
  const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned bits = 0u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    bits |= ((w[i] & 0xFFFFu) != 0u ? 1u : 0u) << (2 * i);
    bits |= ((w[i] >> 16) != 0u ? 1u : 0u) << (2 * i + 1);
  }
  return bits;
}
// dy = dh * mask: registers 8s..8s+7 of the tile whose 16 mask bits start at bit `shift` of `bits`
template <int S>
__device__ __forceinline__ bf16x8 masked_frag(const f32x16& acc, unsigned bits, int shift) {
  bf16x8 f;
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (__bf16)(((bits >> (shift + 8 * S + j)) & 1u) ? acc[8 * S + j] : 0.0f);
  return f;
}

// the activation / gradient dumps are written once and read by a later kernel: non-temporal stores keep
// them from displacing the L2-resident weight stream
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_store(char* p, uint4 v) {
  u32x4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
}

struct DumpAddr {
  char* base;       // [slot][tile][1 KiB]
  int64_t n_tiles;  // tiles in the buffer
  int64_t tile;
  int c, hh;
  __device__ __forceinline__ char* at(int slot) const {
    return base + ((int64_t)slot * n_tiles + tile) * kFragBytes + dump_lane_off(slot, c, hh);
  }
};

// One GEMM layer of the fused chain: for each 32-row out tile, for each k-step, one MFMA.
// C0 = consumption index of the layer's first fragment.
template <int C0, int NK, int NO, class RING, class Init, class GetB, class Epi>
__device__ __forceinline__ void chain_layer(RING& ring, Init init, GetB getb, Epi epi) {
  static_for<NO>([&](auto o_) {
    constexpr int o = decltype(o_)::value;
    f32x16 acc = init(o_);
    static_for<NK>([&](auto k_) {
      constexpr int ks = decltype(k_)::value;
      const bf16x8 a = ring.template next<C0 + o * NK + ks>();
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, getb(k_), acc, 0, 0, 0);
      // Keep the software pipeline the source expresses (one LDS fragment read, for the MFMA kFragAhead
      // steps later, per MFMA) instead of the scheduler's read-wait-use pairs.  A plain scheduling fence
      // per step does it; sched_group_barrier gives the same code but costs ~15 min of compile time here.
      __builtin_amdgcn_sched_barrier(0);
    });
    epi(o_, acc);
  });
}

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
constexpr int kFwdStages = kFwdFrags / kStageFrags;  // 75

template <bool SAVE, bool FROM_RAYS>
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
  const bool tile_ok = tile < n_tiles;

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
  ring.prologue();

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

  DumpAddr dump{save, n_tiles, tile, c, h};
  auto save_frag = [&](int slot, const bf16x8& f) {
    if (SAVE && tile_ok) stream_store(dump.at(slot), frag_to_bits(f));
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
          if constexpr (SAVE && RELU) mask_bits[o >> 1] |= relu_bits(out[2 * o], out[2 * o + 1]) << (16 * (o & 1));
        });
    if constexpr (SAVE && RELU) {
      if (tile_ok)
        *reinterpret_cast<uint4*>(save + ((int64_t)(kSaveMask + S) * n_tiles + tile) * kFragBytes + lane * 16) =
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
            if (tile_ok)
              *reinterpret_cast<uint4*>(save + ((int64_t)(kSaveMask + 8) * n_tiles + tile) * kFragBytes +
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
// Backward, part 1: input-gradient chain.  Produces dy_l (pre-activation gradients) dumps.
// ---------------------------------------------------------------------------------------------
constexpr int kBwdStages = kBwdFrags / kStageFrags;  // 70

__global__ __launch_bounds__(kThreads) void nerf_bwd_chain_kernel(
    const char* __restrict__ packed, const char* __restrict__ save, const float* __restrict__ density,
    const float* __restrict__ rgb, const float* __restrict__ g_density, const float* __restrict__ g_rgb,
    int64_t M, int64_t n_tiles, char* __restrict__ gdump) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  const bool tile_ok = tile < n_tiles;

  // head gradients (fp32): d/d(pre-tanh) and d/d(density logit)
  float gy11[3] = {0, 0, 0}, gy9 = 0.0f;
  if (valid && h == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float y = rgb[m * 3 + k];
      gy11[k] = g_rgb[m * 3 + k] * (1.0f - y * y);             // tanh'
    }
    gy9 = g_density[m] * (1.0f - expf(-density[m]));           // softplus' = sigmoid = 1 - exp(-sp)
  }
  // ReLU masks of h0..h7 and h10 (written by the forward), 16 bytes per lane and layer
  uint4 relu_mask[9];
#pragma unroll
  for (int i = 0; i < 9; ++i)
    relu_mask[i] = tile_ok ? *reinterpret_cast<const uint4*>(save + ((int64_t)(kSaveMask + i) * n_tiles + tile) *
                                                                        kFragBytes + lane * 16)
                           : make_uint4(0, 0, 0, 0);
  __syncthreads();

  Ring<kBwdStages, BwdSeq> ring;
  ring.stream = packed + kPackBwdOff;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();

  DumpAddr gd{gdump, n_tiles, tile, c, h};
  auto dump_frag = [&](int slot, const bf16x8& f) {
    if (tile_ok) stream_store(gd.at(slot), frag_to_bits(f));
  };

  bf16x8 a0[16], a1[16];

  // dy11 fragment: k slot (h=0, j<3) = rgb channel
  bf16x8 dy11 = zero_frag();
  dy11[0] = (__bf16)gy11[0];
  dy11[1] = (__bf16)gy11[1];
  dy11[2] = (__bf16)gy11[2];
  dump_frag(kGradDy11, dy11);
  dump_frag(kGradDy11 + 1, zero_frag());

  // T0: Dense_11^T -> dh10, masked by relu(h10)
  chain_layer<bwd_cons_base(0), bwd_nk(0), bwd_no(0)>(
      ring, [&](auto) { return zero_acc(); }, [&](auto) -> bf16x8 { return dy11; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        const unsigned mb = (o >> 1) == 0 ? relu_mask[8].x : relu_mask[8].y;
        a0[2 * o] = masked_frag<0>(acc, mb, 16 * (o & 1));
        a0[2 * o + 1] = masked_frag<1>(acc, mb, 16 * (o & 1));
        dump_frag(kGradDy10m + 2 * o, a0[2 * o]);
        dump_frag(kGradDy10m + 2 * o + 1, a0[2 * o + 1]);
      });
  // logit-gradient fragment: slot (h=0, j=0)
  bf16x8 dlogit = zero_frag();
  dlogit[0] = (__bf16)gy9;
  dump_frag(kGradDy10m + 8, dlogit);
  dump_frag(kGradDy10m + 9, zero_frag());

  // T1: [Dense_10 | Dense_9]^T (z rows) -> dz = dy8 (Dense_8 output is linear)
  chain_layer<bwd_cons_base(1), bwd_nk(1), bwd_no(1)>(
      ring, [&](auto) { return zero_acc(); },
      [&](auto k_) -> bf16x8 {
        constexpr int ks = decltype(k_)::value;
        if constexpr (ks < 8) return a0[ks];
        else return dlogit;
      },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        a1[2 * o] = acc_to_frag<0, false>(acc);
        a1[2 * o + 1] = acc_to_frag<1, false>(acc);
        dump_frag(grad_dy_slot(8) + 2 * o, a1[2 * o]);
        dump_frag(grad_dy_slot(8) + 2 * o + 1, a1[2 * o + 1]);
      });

  // T2..T9: Dense_l^T for l = 8..1: dy_l (in) -> dh_{l-1}, masked by relu(h_{l-1}) -> dy_{l-1}
  auto back = [&](auto t_, bf16x8(&in)[16], bf16x8(&out)[16]) {
    constexpr int TT = decltype(t_)::value;
    constexpr int l = bwd_dense(TT);  // dense layer whose transpose is applied
    chain_layer<bwd_cons_base(TT), bwd_nk(TT), bwd_no(TT)>(
        ring, [&](auto) { return zero_acc(); },
        [&](auto k_) -> bf16x8 { return in[decltype(k_)::value]; },
        [&](auto o_, const f32x16& acc) {
          constexpr int o = decltype(o_)::value;
          const uint4 mk = relu_mask[l - 1];
          const unsigned mb = (o >> 1) == 0 ? mk.x : ((o >> 1) == 1 ? mk.y : ((o >> 1) == 2 ? mk.z : mk.w));
          out[2 * o] = masked_frag<0>(acc, mb, 16 * (o & 1));
          out[2 * o + 1] = masked_frag<1>(acc, mb, 16 * (o & 1));
          dump_frag(grad_dy_slot(l - 1) + 2 * o, out[2 * o]);
          dump_frag(grad_dy_slot(l - 1) + 2 * o + 1, out[2 * o + 1]);
        });
  };
  back(std::integral_constant<int, 2>{}, a1, a0);  // Dense_8^T: dy8 -> dy7
  back(std::integral_constant<int, 3>{}, a0, a1);  // dy7 -> dy6
  back(std::integral_constant<int, 4>{}, a1, a0);  // dy6 -> dy5
  back(std::integral_constant<int, 5>{}, a0, a1);  // Dense_5^T (h rows): dy5 -> dy4
  back(std::integral_constant<int, 6>{}, a1, a0);  // dy4 -> dy3
  back(std::integral_constant<int, 7>{}, a0, a1);  // dy3 -> dy2
  back(std::integral_constant<int, 8>{}, a1, a0);  // dy2 -> dy1
  back(std::integral_constant<int, 9>{}, a0, a1);  // Dense_1^T: dy1 -> dy0
}

// ---------------------------------------------------------------------------------------------
// Backward, part 2: weight gradients  dW_l[in][out] += sum_m X_l[m][in] * dy_l[m][out]
// A problem = (X tensor of NXF k-step slots) x (dy tensor of NYF slots) for one Dense kernel (or a
// row block of it).  blockIdx -> (problem, K-slice of 32-evaluation steps).  Per step the workgroup
// stages the X and dy fragments into LDS (global -> VGPR -> LDS, two steps of loads in flight) and
// every wave reads its operand tiles transposed (ds_read_b64_tr_b16: feature on the lane,
// evaluation in the registers) for the 32x32x16 MFMA.  Partial sums leave by fp32 atomics.
// ---------------------------------------------------------------------------------------------
enum { ROW_HIDDEN = 0, ROW_XEMB = 1, ROW_DEMB = 2 };
enum { COL_256 = 0, COL_DY10M = 1, COL_DY11 = 2 };
struct WgradProblem {
  int shape;     // operand-shape body, see nerf_wgrad_kernel
  int x_slot0;   // first X slot in the forward save buffer
  int y_slot0;   // first dy slot in the gradient dump
  int dense;     // Flax Dense index (COL_DY10M: Dense_10 with Dense_9 attached as column 128)
  int row_map;   // how X slots map to kernel rows
  int row_off;   // first kernel row of this block
  int col_map;
  int do_bias;
  int first_block, n_blocks;
};
constexpr int kMaxProblems = 13;
struct WgradArgs {
  WgradProblem p[kMaxProblems];
  int n_problems;
};

// operand tile = 32 features (fragment pair starting at frag_even) x 16 evaluations (half q of the
// step).  Lane l receives feature l&31 and evaluations 16q + 8(l>>5) + j, j = 0..7.
__device__ __forceinline__ bf16x8 tr_frag(const char* frag_even, int lane, int parity, int q) {
  const int g = lane >> 4;             // 16-lane group
  const int i = lane & 15;             // supplies block row qp = i>>2 (evaluation), piece p = i&3
  const int frag = g & 1;              // which fragment of the pair (features 0-15 / 16-31)
  const int hk = g >> 1;               // k half
  const int qp = i >> 2, p = i & 3;
  const int slot_par = parity ^ frag;  // slot parity of that fragment (see dump_lane_off)
  const int c0 = 16 * q + 8 * hk;
  const int ca = c0 + qp, cb = c0 + 4 + qp;
  const char* base = frag_even + frag * kFragBytes;
  const int offa = 256 * (ca >> 3) + 128 * (((ca >> 2) & 1) ^ slot_par) + 64 * (p & 1) + 16 * (ca & 3) + 8 * (p >> 1);
  const int offb = 256 * (cb >> 3) + 128 * (((cb >> 2) & 1) ^ slot_par) + 64 * (p & 1) + 16 * (cb & 3) + 8 * (p >> 1);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + offa));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + offb));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

// global -> VGPR -> LDS staging of one iteration = SPI consecutive 32-evaluation steps
// (NXF + NYF fragments each, 8 waves).  The registers hold the iteration that is written to LDS
// after the next barrier; its loads were issued one whole iteration earlier.
template <int NXF, int NYF, int SPI>  // SPI = steps per iteration (per barrier)
struct WgStage {
  static constexpr int kWgSpi = SPI;
  static constexpr int NF = NXF + NYF;
  static constexpr int PER_WAVE = (NF + kWaves - 1) / kWaves;
  static constexpr int STEP_BYTES = NF * kFragBytes;
  static constexpr int ITER_BYTES = kWgSpi * STEP_BYTES;
  const char* x_src;  // slot x_slot0, first tile of this K-slice, + lane*16
  const char* y_src;
  int64_t slot_stride;  // bytes between consecutive slots (= n_tiles KiB)
  int64_t steps;        // steps in this K-slice
  int wave, lane;
  uint4 rr[kWgSpi][PER_WAVE];

  // fragment q of this wave is f = wave + 8q; when NF is not a multiple of 8 the surplus waves of
  // the last round re-load fragment NF-1 (harmless duplicate, keeps the loop branch-free).
  // Steps past the end of the K-slice are staged as zeros (they contribute nothing).
  __device__ __forceinline__ void load(int64_t iter) {
#pragma unroll
    for (int u = 0; u < kWgSpi; ++u) {
      const int64_t step = iter * kWgSpi + u;
      const bool ok = step < steps;
      const int64_t st = ok ? step : 0;
      const unsigned keep = ok ? 0xFFFFFFFFu : 0u;  // branch-free zeroing of out-of-range steps
#pragma unroll
      for (int q = 0; q < PER_WAVE; ++q) {
        int f = wave + kWaves * q;
        if constexpr (NF % kWaves != 0) f = f < NF ? f : NF - 1;
        const char* src = f < NXF ? x_src + (int64_t)f * slot_stride : y_src + (int64_t)(f - NXF) * slot_stride;
        // read-once operands: non-temporal loads leave L2 / Infinity Cache to data that is reused
        const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + st * kFragBytes));
        rr[u][q] = make_uint4(v[0] & keep, v[1] & keep, v[2] & keep, v[3] & keep);
      }
    }
  }
  template <int B>
  __device__ __forceinline__ void write() {
#pragma unroll
    for (int u = 0; u < kWgSpi; ++u) {
#pragma unroll
      for (int q = 0; q < PER_WAVE; ++q) {
        int f = wave + kWaves * q;
        if constexpr (NF % kWaves != 0) f = f < NF ? f : NF - 1;
        *reinterpret_cast<uint4*>(&smem[B * ITER_BYTES + u * STEP_BYTES + f * kFragBytes + lane * 16]) = rr[u][q];
      }
    }
  }
};

template <int NXF, int NYF, int WI, int WO, int SPI>
__device__ __forceinline__ void wgrad_body(const WgradProblem& pb, const char* __restrict__ save,
                                           const char* __restrict__ gdump, int64_t n_tiles,
                                           float* __restrict__ grads) {
  constexpr int NI = NXF / 2, NO = NYF / 2;
  constexpr int TI = (NI + WI - 1) / WI, TO = (NO + WO - 1) / WO;  // tiles per wave
  constexpr bool FULL_I = TI * WI == NI, FULL_O = TO * WO == NO;   // every wave owns TI x TO real tiles
  using Stage = WgStage<NXF, NYF, SPI>;
  constexpr int kWgSpi = SPI;
  static_assert(WI * WO == kWaves, "wave grid");
  static_assert(NXF % 2 == 0 && NYF % 2 == 0, "fragment pairs");

  const int split = blockIdx.x - pb.first_block;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / WO, wo = wave % WO;

  const int64_t per = (n_tiles + pb.n_blocks - 1) / pb.n_blocks;
  const int64_t t0 = (int64_t)split * per;
  const int64_t t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
  const int64_t steps = t1 > t0 ? t1 - t0 : 0;
  const int64_t iters = (steps + kWgSpi - 1) / kWgSpi;

  Stage stg;
  stg.x_src = save + ((int64_t)pb.x_slot0 * n_tiles + t0) * kFragBytes + lane * 16;
  stg.y_src = gdump + ((int64_t)pb.y_slot0 * n_tiles + t0) * kFragBytes + lane * 16;
  stg.slot_stride = n_tiles * kFragBytes;
  stg.steps = steps;
  stg.wave = wave;
  stg.lane = lane;

  f32x16 acc[TI][TO];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TO; ++b) acc[a][b] = zero_acc();
  float bsum[TO];
#pragma unroll
  for (int b = 0; b < TO; ++b) bsum[b] = 0.0f;
  const int ypar = pb.y_slot0 & 1, xpar = pb.x_slot0 & 1;  // slot parity of even fragments

  auto compute = [&](auto buf_) {
    constexpr int bufi = decltype(buf_)::value;
#pragma unroll
    for (int u = 0; u < kWgSpi; ++u) {
      const char* buf = smem + bufi * Stage::ITER_BYTES + u * Stage::STEP_BYTES;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        bf16x8 bf[TO];
#pragma unroll
        for (int b = 0; b < TO; ++b) {
          const int ot = wo + WO * b;
          if (FULL_O || ot < NO) {
            bf[b] = tr_frag(buf + (NXF + 2 * ot) * kFragBytes, lane, ypar, q);
            if (wi == 0) {
#pragma unroll
              for (int j = 0; j < 8; ++j) bsum[b] += (float)bf[b][j];
            }
          }
        }
#pragma unroll
        for (int a = 0; a < TI; ++a) {
          const int it = wi + WI * a;
          if (FULL_I || it < NI) {
            const bf16x8 af = tr_frag(buf + 2 * it * kFragBytes, lane, xpar, q);
#pragma unroll
            for (int b = 0; b < TO; ++b) {
              const int ot = wo + WO * b;
              if (FULL_O || ot < NO)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[b], acc[a][b], 0, 0, 0);
            }
          }
        }
      }
    }
  };
  std::integral_constant<int, 0> c0;
  std::integral_constant<int, 1> c1;
  // iteration i: barrier | write iteration i+1 (registers) to LDS buffer (i+1)&1 | load iteration i+2 |
  // compute iteration i from buffer i&1.  Loads stay in flight for one whole iteration of MFMA work.
  if (iters > 0) {
    stg.load(0);
    stg.template write<0>();
    stg.load(1);
  }
  for (int64_t i = 0; i < iters; i += 2) {
    __syncthreads();
    stg.template write<1>();
    stg.load(i + 2);
    compute(c0);
    __syncthreads();
    stg.template write<0>();
    stg.load(i + 3);
    if (i + 1 < iters) compute(c1);
  }

  // epilogue: atomically add the partial dW tiles / bias sums
  const int colr = lane & 31, hh = lane >> 5;
  static_for<TO>([&](auto b_) {
    constexpr int b = decltype(b_)::value;
    const int ot = wo + WO * b;
    int out_idx = -1, out_dim = 1, dense_w = pb.dense;
    if (ot < NO) {
      if (pb.col_map == COL_DY10M) {  // tiles 0..3 = Dense_10 outputs, tile 4 column 0 = Dense_9
        if (ot < 4) { out_idx = 32 * ot + colr; out_dim = 128; dense_w = 10; }
        else if (colr == 0 && pb.row_map == ROW_HIDDEN) { out_idx = 0; out_dim = 1; dense_w = 9; }
      } else if (pb.col_map == COL_DY11) {
        if (colr < 3) { out_idx = colr; out_dim = 3; }
      } else {
        out_idx = 32 * ot + colr; out_dim = 256;
      }
    }
    if (wi == 0 && pb.do_bias) {
      float sacc = bsum[b];
      sacc += __shfl_xor(sacc, 32, 64);
      if (hh == 0 && out_idx >= 0) atomicAdd(grads + dense_b_off(dense_w) + out_idx, sacc);
    }
    static_for<TI>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      const int it = wi + WI * a;
      static_for<16>([&](auto q_) {
        constexpr int qq = decltype(q_)::value;
        const int r = (qq & 3) + 8 * (qq >> 2) + 4 * hh;  // row in the 32-feature tile
        const int f = 2 * it + (r >> 4);                   // k-step slot within X
        const int r16 = r & 15;
        const int sh = (r16 >> 2) & 1, sj = 4 * (r16 >> 3) + (r16 & 3);  // slot (h, j) of that feature
        int in_idx;
        if (pb.row_map == ROW_HIDDEN) in_idx = 16 * f + r16;
        else if (pb.row_map == ROW_XEMB) in_idx = xemb_feat(f, sh, sj);
        else in_idx = demb_feat(f, sh, sj);
        if (it < NI && out_idx >= 0 && in_idx >= 0)
          atomicAdd(grads + dense_w_off(dense_w) + (int64_t)(in_idx + pb.row_off) * out_dim + out_idx,
                    acc[a][b][qq]);
      });
    });
  });
}

// One launch for every Dense layer of the model: blockIdx -> problem -> operand-shape body.
// Problems are listed heaviest first so that the small ones fill the tail of the launch.
__global__ __launch_bounds__(kThreads) void nerf_wgrad_kernel(WgradArgs args, const char* __restrict__ save,
                                                              const char* __restrict__ gdump,
                                                              int64_t n_tiles, float* __restrict__ grads) {
  WgradProblem pb = args.p[0];
#pragma unroll
  for (int i = 1; i < kMaxProblems; ++i)
    if (i < args.n_problems && (int)blockIdx.x >= args.p[i].first_block) pb = args.p[i];
  switch (pb.shape) {
    // steps per barrier chosen so that every body keeps ~60 KB of loads in flight per workgroup
    case 0: wgrad_body<16, 16, 4, 2, 2>(pb, save, gdump, n_tiles, grads); break;
    case 1: wgrad_body<16, 10, 4, 2, 2>(pb, save, gdump, n_tiles, grads); break;
    case 2: wgrad_body<4, 16, 2, 4, 3>(pb, save, gdump, n_tiles, grads); break;
    case 3: wgrad_body<2, 10, 1, 8, 5>(pb, save, gdump, n_tiles, grads); break;
    default: wgrad_body<8, 2, 4, 2, 6>(pb, save, gdump, n_tiles, grads); break;
  }
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
__global__ void nerf_pack_kernel(const float* __restrict__ params, char* __restrict__ packed) {
  const int64_t total_f = (int64_t)kFwdFrags * 512;
  const int64_t total_b = (int64_t)kBwdFrags * 512;
  const int64_t total = total_f + total_b + kBiasFloats;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    if (e < total_f + total_b) {
      const bool fwd = e < total_f;
      const int64_t ee = fwd ? e : e - total_f;
      const int g = (int)(ee >> 9), lane = (int)((ee >> 3) & 63), j = (int)(ee & 7);
      int idx = -1;
      if (fwd) {
        int s = 0;
        for (int i = 1; i < kFwdLayers; ++i)
          if (g >= fwd_base(i)) s = i;
        const int loc = g - fwd_base(s);
        if (loc < fwd_nk(s) * fwd_no(s)) idx = fwd_weight_index(s, loc / fwd_nk(s), loc % fwd_nk(s), lane, j);
      } else {
        int t = 0;
        for (int i = 1; i < kBwdLayers; ++i)
          if (g >= bwd_base(i)) t = i;
        const int loc = g - bwd_base(t);
        if (loc < bwd_nk(t) * bwd_no(t)) idx = bwd_weight_index(t, loc / bwd_nk(t), loc % bwd_nk(t), lane, j);
      }
      const float v = idx >= 0 ? params[idx] : 0.0f;
      __bf16* dst = reinterpret_cast<__bf16*>(packed + (fwd ? kPackFwdOff : kPackBwdOff));
      dst[ee] = (__bf16)v;
    } else {
      const int i = (int)(e - total_f - total_b);
      int s = 0;
      for (int k = 1; k < kFwdLayers; ++k)
        if (i >= fwd_bias_base(k)) s = k;
      const int idx = fwd_bias_index(s, i - fwd_bias_base(s));
      reinterpret_cast<float*>(packed + kPackBiasOff)[i] = idx >= 0 ? params[idx] : 0.0f;
    }
  }
}

}  // namespace lnrf

using namespace lnrf;

static bool shape_supported(const lnrf_nerf_shape* s) {
  return s && s->input_layers == 5 && s->mid_layers == 4 && s->hidden_dim == 256 &&
         s->color_layer_dim == 128 && s->x_freqs == 10 && s->d_freqs == 4;
}
static inline int64_t tiles_for(int64_t m) { return (m + kTileCols - 1) / kTileCols; }

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
extern "C" int64_t lnrf_nerf_bwd_scratch_bytes(const lnrf_nerf_shape* s, int64_t m) {
  return shape_supported(s) ? (int64_t)kGradSlots * tiles_for(m) * kFragBytes : -1;
}

extern "C" int lnrf_nerf_pack_weights(const lnrf_nerf_shape* shape, const float* params, void* packed,
                                      lnrf_stream_t stream) {
  if (!shape_supported(shape)) {
    set_error("lnrf_nerf_pack_weights: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(params && packed, "null pointer");
  hipLaunchKernelGGL(nerf_pack_kernel, dim3(1024), dim3(256), 0, as_stream(stream), params, (char*)packed);
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

extern "C" int lnrf_nerf_mlp_fwd(const lnrf_nerf_shape* shape, const void* packed, const float* x,
                                 const float* d, const float* rays, int64_t ray_stride, const float* ts,
                                 int32_t t, int64_t m, float* density, float* rgb, void* save,
                                 lnrf_stream_t stream) {
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
#define LAUNCH_FWD(SAVE, RAYS)                                                                       \
  do {                                                                                               \
    rc = ensure_lds(nerf_fwd_kernel<SAVE, RAYS>, kFusedLds);                                         \
    if (rc) return rc;                                                                               \
    hipLaunchKernelGGL((nerf_fwd_kernel<SAVE, RAYS>), grid, block, kFusedLds, st, (const char*)packed, \
                       x, d, rays, ray_stride, ts, (int)t, m, n_tiles, density, rgb, (char*)save);   \
  } while (0)
  if (save) {
    if (from_rays) LAUNCH_FWD(true, true); else LAUNCH_FWD(true, false);
  } else {
    if (from_rays) LAUNCH_FWD(false, true); else LAUNCH_FWD(false, false);
  }
#undef LAUNCH_FWD
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

extern "C" int lnrf_nerf_mlp_bwd_weights(const lnrf_nerf_shape* shape, const void* save, const void* scratch,
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
  // weight-gradient problems: ONE launch, heaviest problems first
  WgradArgs a;
  a.n_problems = 0;
  int first = 0;
  auto add = [&](int shape, int xs, int ys, int dense, int row_map, int row_off, int col_map, int do_bias,
                 int blocks) {
    WgradProblem p;
    p.shape = shape; p.x_slot0 = xs; p.y_slot0 = ys; p.dense = dense; p.row_map = row_map; p.row_off = row_off;
    p.col_map = col_map; p.do_bias = do_bias;
    int64_t nb = blocks;
    const int64_t max_nb = (n_tiles + 5) / 6;
    if (nb > max_nb) nb = max_nb;
    p.first_block = first;
    p.n_blocks = (int)nb;
    first += (int)nb;
    a.p[a.n_problems++] = p;
  };
  // hidden x hidden: Dense_1..8 (Dense_5: rows 0..255)
  for (int l = 1; l <= 8; ++l) add(0, kSaveH + (l - 1) * 16, grad_dy_slot(l), l, ROW_HIDDEN, 0, COL_256, 1, l <= 4 ? 48 : 47);
  // z x dy10m: Dense_10 rows 0..255 and Dense_9
  add(1, kSaveZ, kGradDy10m, 10, ROW_HIDDEN, 0, COL_DY10M, 1, 39);
  // x_emb x dy0 (Dense_0) and x_emb x dy5 (Dense_5 rows 256..315)
  add(2, kSaveXin, grad_dy_slot(0), 0, ROW_XEMB, 0, COL_256, 1, 30);
  add(2, kSaveXin, grad_dy_slot(5), 5, ROW_XEMB, 256, COL_256, 0, 30);
  // d_emb x dy10m: Dense_10 rows 256..279
  add(3, kSaveDin, kGradDy10m, 10, ROW_DEMB, 256, COL_DY10M, 0, 18);
  // h10 x dy11: Dense_11
  add(4, kSaveH10, kGradDy11, 11, ROW_HIDDEN, 0, COL_DY11, 1, 15);
  const int lds = 2 * 2 * 32 * kFragBytes;  // largest body: 2 buffers x 2 steps x (16 + 16) fragments
  rc = ensure_lds(nerf_wgrad_kernel, lds);
  if (rc) return rc;
  hipLaunchKernelGGL(nerf_wgrad_kernel, dim3((unsigned)first), dim3(kThreads), lds, st, a, (const char*)save,
                     (const char*)scratch, n_tiles, grads);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}
