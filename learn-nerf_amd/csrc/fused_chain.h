// fused_chain.h — device building blocks shared by the fused MLP kernels (nerf_mlp.hip, ngp_mlp.hip):
// the weight ring, accumulator <-> B-fragment conversion, dump addressing, the chain layer and the
// split-K weight-gradient body.
#pragma once
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
constexpr int kFragAhead = 4;        // A-fragments read from LDS ahead of the MFMA that uses them

extern __shared__ __attribute__((aligned(16))) char smem[];

// In-kernel timeline (debug builds with -DLNRF_TIMELINE only, see tools/timeline_probe.py): the waves of one
// workgroup write s_memtime stamps to a buffer; compiled out of the product library.
#ifdef LNRF_TIMELINE
struct Timeline {
  unsigned long long* buf = nullptr;  // [8 waves][1024 stamps]
  int n = 0;
  bool on = false;
  __device__ __forceinline__ void stamp() {
    if (on) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0) buf[(threadIdx.x >> 6) * 1024 + n] = t;
      ++n;
    }
  }
};
#define LNRF_TL_STAMP(ring) (ring).tl.stamp()
#else
#define LNRF_TL_STAMP(ring) ((void)0)
#endif

__device__ __forceinline__ bf16x8 bits_to_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ uint4 frag_to_bits(bf16x8 v) { return __builtin_bit_cast(uint4, v); }
__device__ __forceinline__ bf16x8 zero_frag() { return bits_to_frag(make_uint4(0, 0, 0, 0)); }


// The weight ring.  A stage is 16 fragments (16 KiB) shared by the 8 waves; every wave moves 2 of
// them.  Staging is global -> VGPR -> LDS (not LDS-DMA: hipcc drains vmcnt(0) before any ds_read
// while an LDS-DMA is pending, which serialises the pipeline).  Timeline at the barrier that opens
// stage T: stage T+1 is already in LDS and becomes visible (it was written after barrier T-1), stage
// T+2 is written from registers into the slot freed by stage T-1, stage T+4 is requested from L2.
// Because stage T+1 is visible during stage T, the per-wave FIFO of A-fragments (kFragAhead LDS reads
// in flight) runs continuously across stage boundaries.
template <int NSTAGES, class SEQ, int NW = kWaves>
struct Ring {
  static constexpr int kPerWave = kStageFrags / NW;  // fragments of a stage moved by one wave
  const char* stream;  // global, NSTAGES * 16 KiB, fragment order
  int wave, lane;
  uint4 r[2][kPerWave];
  bf16x8 fifo[kFragAhead];
#ifdef LNRF_TIMELINE
  Timeline tl;
#endif

  template <int T>
  __device__ __forceinline__ void load() {
#pragma unroll
    for (int q = 0; q < kPerWave; ++q) {
      const int f = wave + NW * q;
      r[T & 1][q] = *reinterpret_cast<const uint4*>(stream + ((int64_t)T * kStageFrags + f) * kFragBytes +
                                                    lane * 16);
    }
  }
  template <int T>
  __device__ __forceinline__ void write() {
#pragma unroll
    for (int q = 0; q < kPerWave; ++q) {
      const int f = wave + NW * q;
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
#ifdef LNRF_TIMELINE
    tl.stamp();  // arrived at the stage barrier
#endif
    __syncthreads();
#ifdef LNRF_TIMELINE
    tl.stamp();  // released
#endif
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
#ifdef LNRF_PLAIN_DUMP_STORES  // A/B: ordinary write-back stores
  *reinterpret_cast<u32x4*>(p) = t;
#else
  __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
#endif
}

// Byte offset of fragment (slot, tile) in a dump buffer.  n_slots == 0: slot-major [slot][tile][1 KiB] (a slot's tiles
// are contiguous; a wave's dumps of one tile are n_tiles KiB apart).  n_slots > 0: tile-major [tile][slot][1 KiB] (a
// tile's n_slots fragments are one contiguous block: the dumping wave writes through one block front to back, and a
// weight-gradient workgroup reads its layer's 16 + 16 KiB out of consecutive blocks).
// With LNRF_DUMP_GROUP = G > 1 (a power of two dividing the 8 tiles of a workgroup) the blocks hold G tiles each:
// [tile / G][slot][tile % G][1 KiB], i.e. the 8 waves of a dumping workgroup fill G KiB runs per slot.
#ifndef LNRF_DUMP_GROUP
#define LNRF_DUMP_GROUP 1
#endif
constexpr int kDumpGroup = LNRF_DUMP_GROUP;
static_assert(kDumpGroup >= 1 && (kDumpGroup & (kDumpGroup - 1)) == 0 && kDumpGroup <= 8, "dump group");
__device__ __forceinline__ int64_t dump_off(int slot, int64_t tile, int64_t n_tiles, int n_slots) {
  if (n_slots > 0) return (((tile / kDumpGroup) * n_slots + slot) * kDumpGroup + tile % kDumpGroup) * kFragBytes;
  return ((int64_t)slot * n_tiles + tile) * kFragBytes;
}
struct DumpAddr {
  char* base;
  int64_t n_tiles;  // tiles in the buffer
  int64_t tile;
  int c, hh;
  int n_slots = 0;  // 0: slot-major, else tile-major with this many slots per tile
  __device__ __forceinline__ char* at(int slot) const {
    return base + dump_off(slot, tile, n_tiles, n_slots) + dump_lane_off(slot, c, hh);
  }
  // one fragment of this tile, non-temporal.  LNRF_BUFFER_DUMP_STORES (experiment): tile-major buffers take a buffer
  // store — the tile's block as scalar base, the slot as scalar offset, the lane's 16 bytes as the only vector operand
  __device__ __forceinline__ void store(int slot, uint4 v) const {
#ifdef LNRF_BUFFER_DUMP_STORES
    if (n_slots > 0 && kDumpGroup == 1) {
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(base + tile * n_slots * kFragBytes, 0, 0x7FFFFFFF, 0x00020000);
      const u32x4 t = {v.x, v.y, v.z, v.w};
      __builtin_amdgcn_raw_buffer_store_b128(t, rs, dump_lane_off(slot, c, hh), slot * kFragBytes, 2);  // 2 = nt
      return;
    }
#endif
    stream_store(at(slot), v);
  }
};
// which layout the operand buffers of a weight-gradient launch have (slots per tile, 0 = slot-major)
struct WgLayout {
  int x_slots = 0, y_slots = 0;
  int interleave = 0;  // 1: split s of a problem takes tiles (i * n_blocks + s) * SPI + u instead of a contiguous range
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
    LNRF_TL_STAMP(ring);  // MFMAs of the tile issued
    epi(o_, acc);
    LNRF_TL_STAMP(ring);  // epilogue issued
  });
}

// ---- split precision ("bf16x3"): every fp32 operand is a bf16 pair hi = bf16(v), lo = bf16(v - hi) and every product is
// lo*hi + hi*lo + hi*hi on the bf16 MFMA with fp32 accumulation.  The activations of a layer then need 2 x 64 VGPRs in
// and 2 x 64 out, so these kernels run 4 waves per workgroup (one per SIMD, the whole 512-entry register file each).
constexpr int kSplitWaves = 4;
constexpr int kSplitThreads = kSplitWaves * 64;
// v = hi + lo with hi = bf16(v), lo = bf16(v - hi)
template <int S, bool RELU>
__device__ __forceinline__ void acc_to_frag_split(const f32x16& acc, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v = acc[8 * S + j];
    if (RELU) v = __builtin_amdgcn_fmed3f(v, 0.0f, __builtin_inff());
    const __bf16 hb = (__bf16)v;
    hi[j] = hb;
    lo[j] = (__bf16)(v - (float)hb);
  }
}
__device__ __forceinline__ void split_store(float v, bf16x8& hi, bf16x8& lo, int j) {
  const __bf16 hb = (__bf16)v;
  hi[j] = hb;
  lo[j] = (__bf16)(v - (float)hb);
}

// one GEMM layer with split operands; C0 = consumption index (in hi/lo pairs) of the layer's first k-step
template <int C0, int NK, int NO, class RING, class Init, class GetHi, class GetLo, class Epi>
__device__ __forceinline__ void chain_layer_split(RING& ring, Init init, GetHi bhi, GetLo blo, Epi epi) {
  static_for<NO>([&](auto o_) {
    constexpr int o = decltype(o_)::value;
    f32x16 acc = init(o_);
    static_for<NK>([&](auto k_) {
      constexpr int ks = decltype(k_)::value;
      constexpr int c = 2 * (C0 + o * NK + ks);
      const bf16x8 ahi = ring.template next<c>();
      const bf16x8 alo = ring.template next<c + 1>();
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi(k_), acc, 0, 0, 0);  // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo(k_), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi(k_), acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    });
    epi(o_, acc);
  });
}


// ---------------------------------------------------------------------------------------------
// weight gradients
// Backward, part 2: weight gradients  dW_l[in][out] += sum_m X_l[m][in] * dy_l[m][out]
// A problem = (X tensor of NXF k-step slots) x (dy tensor of NYF slots) for one Dense kernel (or a
// row block of it).  blockIdx -> (problem, K-slice of 32-evaluation steps).  Per step the workgroup
// stages the X and dy fragments into LDS (global -> VGPR -> LDS, two steps of loads in flight) and
// every wave reads its operand tiles transposed (ds_read_b64_tr_b16: feature on the lane,
// evaluation in the registers) for the 32x32x16 MFMA.  Partial sums leave by fp32 atomics.
// ---------------------------------------------------------------------------------------------
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
// SPI = steps per iteration (per barrier); PLAIN: ordinary loads instead of non-temporal ones; BUF: buffer loads (scalar
// base re-anchored at every iteration + 32-bit offsets) instead of flat 64-bit addresses — measured -3 % on the fine-pass
// launch of NeRFModel (1.55 -> 1.50 ms, three interleaved runs each); tile-major operand buffers only (small offsets)
template <int NXF, int NYF, int SPI, bool PLAIN = false, bool BUF = false>
struct WgStage {
  static constexpr int kWgSpi = SPI;
  static constexpr int NF = NXF + NYF;
  static constexpr int PER_WAVE = (NF + kWaves - 1) / kWaves;
  static constexpr int STEP_BYTES = NF * kFragBytes;
  static constexpr int ITER_BYTES = kWgSpi * STEP_BYTES;
  const char* x_src;  // buffer base + lane*16
  const char* y_src;
  const char* x_base;  // buffer base (BUF)
  const char* y_base;
  int x_slot0, y_slot0, x_slots, y_slots;  // first slot of the operand; slots per tile of its buffer (0: slot-major)
  int64_t t0, n_tiles;                     // first tile of this K-slice; tiles in the buffers
  int64_t iter_stride, t_end;              // tile of (iteration i, step u) = t0 + i * iter_stride + u, valid below t_end
  int64_t steps;        // steps in this K-slice
  int wave, lane;
  uint4 rr[kWgSpi][PER_WAVE];

  // fragment q of this wave is f = wave + 8q; when NF is not a multiple of 8 the surplus waves of
  // the last round re-load fragment NF-1 (harmless duplicate, keeps the loop branch-free).
  // Steps past the end of the K-slice are staged as zeros (they contribute nothing).
  __device__ __forceinline__ void load(int64_t iter) {
    if constexpr (BUF) {
      const int64_t tb = t0 + iter * iter_stride;      // first tile of the iteration
      const int64_t tbc = tb < t_end ? tb : t0;         // an address that exists, for the steps past the end
      const int64_t xo = dump_off(x_slot0, tbc, n_tiles, x_slots), yo = dump_off(y_slot0, tbc, n_tiles, y_slots);
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(x_base) + xo, 0, 0x7FFFFFFF, 0x00020000);
      const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(y_base) + yo, 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int u = 0; u < kWgSpi; ++u) {
        const int64_t tl = tb + u;
        const bool ok = tl < t_end;
        const int64_t ta = ok ? tl : tbc;
        const unsigned keep = ok ? 0xFFFFFFFFu : 0u;  // branch-free zeroing of out-of-range steps
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) {
          int f = wave + kWaves * q;
          if constexpr (NF % kWaves != 0) f = f < NF ? f : NF - 1;
          const bool isx = f < NXF;
          const int bo = (int)(isx ? dump_off(x_slot0 + f, ta, n_tiles, x_slots) - xo
                                   : dump_off(y_slot0 + f - NXF, ta, n_tiles, y_slots) - yo);
          // cache policy (aux): 0 ordinary, 2 non-temporal
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(isx ? rx : ry, bo + lane * 16, 0, PLAIN ? 0 : 2);
          rr[u][q] = make_uint4(v[0] & keep, v[1] & keep, v[2] & keep, v[3] & keep);
        }
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < kWgSpi; ++u) {
      const int64_t tl = t0 + iter * iter_stride + u;
      const bool ok = tl < t_end;
      const int64_t st = ok ? tl - t0 : 0;
      const unsigned keep = ok ? 0xFFFFFFFFu : 0u;  // branch-free zeroing of out-of-range steps
#pragma unroll
      for (int q = 0; q < PER_WAVE; ++q) {
        int f = wave + kWaves * q;
        if constexpr (NF % kWaves != 0) f = f < NF ? f : NF - 1;
        const char* src = f < NXF ? x_src + dump_off(x_slot0 + f, t0 + st, n_tiles, x_slots)
                                  : y_src + dump_off(y_slot0 + f - NXF, t0 + st, n_tiles, y_slots);
        // Non-temporal loads leave L2 / Infinity Cache to data that is reused — measured better for the coarse pass
        // (1.4 GB of operands: 0.58-0.60 vs 0.61 ms); ordinary loads are better for the fine pass (8.8 GB: 1.53 vs
        // 1.57-1.58 ms), so the launch picks by size.
        u32x4 v;
        if constexpr (PLAIN) v = *reinterpret_cast<const u32x4*>(src);
        else v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
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

// slab epilogue of wgrad_body: floats per (workgroup, wave): up to 9 accumulator tiles of 64 lanes x 16 + 4 bias rows
constexpr int kSlabMaxTiles = 9, kSlabMaxTO = 4;
constexpr int kSlabTileFloats = 64 * 16;
constexpr int kSlabWaveFloats = kSlabMaxTiles * kSlabTileFloats + kSlabMaxTO * 64;
constexpr int64_t kSlabBlockBytes = (int64_t)kWaves * kSlabWaveFloats * (int64_t)sizeof(float);

// PB supplies x_slot0, y_slot0, do_bias, first_block, n_blocks; EPI maps (out tile, column) and
// (X fragment, row) to gradient-vector offsets.
template <int NXF, int NYF, int WI, int WO, int SPI, class EPI, bool PLAIN = false, bool BUF = false, class PB>
__device__ __forceinline__ void wgrad_body(const PB& pb, const char* __restrict__ save,
                                           const char* __restrict__ gdump, int64_t n_tiles,
                                           float* __restrict__ grads, WgLayout lay = WgLayout{},
                                           float* __restrict__ slabs = nullptr) {
  constexpr int NI = NXF / 2, NO = NYF / 2;
  constexpr int TI = (NI + WI - 1) / WI, TO = (NO + WO - 1) / WO;  // tiles per wave
  constexpr bool FULL_I = TI * WI == NI, FULL_O = TO * WO == NO;   // every wave owns TI x TO real tiles
  using Stage = WgStage<NXF, NYF, SPI, PLAIN, BUF>;
  constexpr int kWgSpi = SPI;
  static_assert(WI * WO == kWaves, "wave grid");
  static_assert(NXF % 2 == 0 && NYF % 2 == 0, "fragment pairs");

  const int split = blockIdx.x - pb.first_block;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / WO, wo = wave % WO;

  int64_t per = (n_tiles + pb.n_blocks - 1) / pb.n_blocks;
  per = (per + kDumpGroup - 1) / kDumpGroup * kDumpGroup;  // K-slices start on a tile group of the dump layout
  int64_t t0 = (int64_t)split * per;
  const int64_t t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
  int64_t steps = t1 > t0 ? t1 - t0 : 0;
  int64_t iters = (steps + kWgSpi - 1) / kWgSpi;
  int64_t iter_stride = kWgSpi, t_end = t1;
  if (lay.interleave) {  // the splits of a problem walk the tiles side by side
    t0 = (int64_t)split * kWgSpi;
    iter_stride = (int64_t)pb.n_blocks * kWgSpi;
    t_end = n_tiles;
    iters = t0 < n_tiles ? (n_tiles - t0 + iter_stride - 1) / iter_stride : 0;
    steps = iters * kWgSpi;
  }

  Stage stg;
  stg.x_src = save + lane * 16;
  stg.y_src = gdump + lane * 16;
  stg.x_base = save;
  stg.y_base = gdump;
  stg.x_slot0 = pb.x_slot0;
  stg.y_slot0 = pb.y_slot0;
  stg.x_slots = lay.x_slots;
  stg.y_slots = lay.y_slots;
  stg.t0 = t0;
  stg.n_tiles = n_tiles;
  stg.iter_stride = iter_stride;
  stg.t_end = t_end;
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

#ifdef LNRF_NO_FLUSH
  if (n_tiles >= 0) return;  // timing experiment only (results are wrong): what the epilogue below costs
#endif
  if (slabs != nullptr) {
    // Slab epilogue: the workgroup's accumulators leave as they stand — [workgroup][wave][tile a * TO + b][4][lane][4 f32],
    // 4 KiB per tile written as four 1 KiB store instructions — plus the per-lane bias partial sums of the wi == 0 waves;
    // wgrad_reduce_tile() folds the slabs of a problem in a fixed order and runs the index mapping below once.  Against
    // fp32 atomics (memory-side, ~1.3 TB/s of added bytes chip-wide: 62-67 us for the ~100 MB of one NeRFModel launch)
    // the partial sums move at store / load rates, and the gradient no longer depends on arrival order.
    float* __restrict__ mine = slabs + ((int64_t)blockIdx.x * kWaves + wave) * kSlabWaveFloats;
#pragma unroll
    for (int a = 0; a < TI; ++a)
#pragma unroll
      for (int b = 0; b < TO; ++b) {
        // [tile][v][lane][4 f32]: every store instruction writes 1 KiB contiguous
        float4* dst = reinterpret_cast<float4*>(mine + (a * TO + b) * kSlabTileFloats) + lane;
#pragma unroll
        for (int v = 0; v < 4; ++v)
          dst[64 * v] = make_float4(acc[a][b][4 * v], acc[a][b][4 * v + 1], acc[a][b][4 * v + 2], acc[a][b][4 * v + 3]);
      }
    if (wi == 0) {
#pragma unroll
      for (int b = 0; b < TO; ++b) mine[kSlabTileFloats * kSlabMaxTiles + b * 64 + lane] = bsum[b];
    }
    return;
  }
  // epilogue: atomically add the partial dW tiles / bias sums
  const int colr = lane & 31, hh = lane >> 5;
  static_for<TO>([&](auto b_) {
    constexpr int b = decltype(b_)::value;
    const int ot = wo + WO * b;
    int out_idx = -1, out_dim = 1;
    int64_t w_off = 0, b_off = 0;
    if (ot < NO) EPI::cols(pb, ot, colr, out_idx, out_dim, w_off, b_off);
    if (wi == 0 && pb.do_bias) {
      float sacc = bsum[b];
      sacc += __shfl_xor(sacc, 32, 64);
      if (hh == 0 && out_idx >= 0 && b_off >= 0) atomicAdd(grads + b_off + out_idx, sacc);
    }
    const int row_lim = EPI::row_limit(pb, ot);
    static_for<TI>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      const int it = wi + WI * a;
      static_for<16>([&](auto q_) {
        constexpr int qq = decltype(q_)::value;
        const int r = (qq & 3) + 8 * (qq >> 2) + 4 * hh;  // row in the 32-feature tile
        const int f = 2 * it + (r >> 4);                   // k-step slot within X
        const int r16 = r & 15;
        const int in_idx = EPI::row(pb, f, r16);  // kernel row of that X feature, -1 = padding
        if (it < NI && out_idx >= 0 && in_idx >= 0 && in_idx < row_lim)
          atomicAdd(grads + w_off + (int64_t)in_idx * out_dim + out_idx,
                    acc[a][b][qq]);
      });
    });
  });
}

// A 4-wave workgroup folds ONE accumulator tile (wave w of the producing workgroups, tile j = a * TO + b) of a problem
// over the problem's n_blocks slabs: wave q sums the slabs q, q + 4, ... in order, the four partial sums meet in LDS and
// wave 0 adds them (q = 0..3) and applies the same index mapping as the atomic epilogue of wgrad_body.  The order of
// every addition is fixed and every parameter has exactly one owner tile (plain read-modify-write): bit-reproducible.
constexpr int kSlabReduceWaves = 4;  // 8 measured no faster (finish phase 0.67 vs 0.65 ms, Ref-NeRF step unchanged)
template <int NXF, int NYF, int WI, int WO, class EPI, class PB>
__device__ __forceinline__ void wgrad_reduce_tile(const PB& pb, int w, int j, const float* __restrict__ slabs,
                                                  float* __restrict__ grads, float* lds) {
  constexpr int NI = NXF / 2, NO = NYF / 2;
  constexpr int TI = (NI + WI - 1) / WI, TO = (NO + WO - 1) / WO;
  static_assert(TI * TO <= kSlabMaxTiles && TO <= kSlabMaxTO, "slab layout");
  if (j >= TI * TO) return;  // uniform for the workgroup
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int a = j / TO, b = j % TO;
  const int wi = w / WO, wo = w % WO;
  const int it = wi + WI * a, ot = wo + WO * b;
  if (it >= NI || ot >= NO) return;
  const float* __restrict__ src = slabs + ((int64_t)pb.first_block * kWaves + w) * kSlabWaveFloats;
  const int64_t blk_stride = (int64_t)kWaves * kSlabWaveFloats;
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  const bool bias = wi == 0 && a == 0 && pb.do_bias;
  float bsum = 0.0f;
  constexpr int U = 4;  // slabs in flight per wave
  for (int s0 = q; s0 < pb.n_blocks; s0 += U * kSlabReduceWaves) {
    float4 t[U][4];
    float bb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sb = s0 + u * kSlabReduceWaves;
      const int sblk = sb < pb.n_blocks ? sb : pb.n_blocks - 1;
      const float* base = src + (int64_t)sblk * blk_stride;
      const float4* tp = reinterpret_cast<const float4*>(base + j * kSlabTileFloats) + lane;
#pragma unroll
      for (int v = 0; v < 4; ++v) t[u][v] = tp[64 * v];
      bb[u] = bias ? base[kSlabTileFloats * kSlabMaxTiles + b * 64 + lane] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = s0 + u * kSlabReduceWaves < pb.n_blocks;  // the tail re-reads the last slab and adds zeros
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        acc[4 * v] += ok ? t[u][v].x : 0.0f;
        acc[4 * v + 1] += ok ? t[u][v].y : 0.0f;
        acc[4 * v + 2] += ok ? t[u][v].z : 0.0f;
        acc[4 * v + 3] += ok ? t[u][v].w : 0.0f;
      }
      bsum += ok ? bb[u] : 0.0f;
    }
  }
  // partial sums of waves 1..3 -> LDS [q][17][64]
  if (q > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[((q - 1) * 17 + r) * 64 + lane] = acc[r];
    lds[((q - 1) * 17 + 16) * 64 + lane] = bsum;
  }
  __syncthreads();
  if (q > 0) return;
#pragma unroll
  for (int p = 0; p < kSlabReduceWaves - 1; ++p) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += lds[(p * 17 + r) * 64 + lane];
    bsum += lds[(p * 17 + 16) * 64 + lane];
  }
  const int colr = lane & 31, hh = lane >> 5;
  int out_idx = -1, out_dim = 1;
  int64_t w_off = 0, b_off = 0;
  EPI::cols(pb, ot, colr, out_idx, out_dim, w_off, b_off);
  const int row_lim = EPI::row_limit(pb, ot);
  if (bias) {
    float sacc = bsum;
    sacc += __shfl_xor(sacc, 32, 64);
    if (hh == 0 && out_idx >= 0 && b_off >= 0) grads[b_off + out_idx] += sacc;
  }
#pragma unroll
  for (int qq = 0; qq < 16; ++qq) {
    const int r = (qq & 3) + 8 * (qq >> 2) + 4 * hh;  // row in the 32-feature tile
    const int f = 2 * it + (r >> 4);                   // k-step slot within X
    const int in_idx = EPI::row(pb, f, r & 15);
    if (out_idx >= 0 && in_idx >= 0 && in_idx < row_lim) grads[w_off + (int64_t)in_idx * out_dim + out_idx] += acc[qq];
  }
}

}  // namespace lnrf
