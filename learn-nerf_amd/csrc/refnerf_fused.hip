// refnerf_fused.hip — the spatial block of RefNERFModel (ref_nerf.py:92-99: Dense_0..8, the NeRFModel trunk) and
// everything that differentiates through it, on the fused bf16-MFMA chain of nerf_mlp.hip / nerf_chain.h:
//
//   trunk forward   spatial_out = Dense_8(...)  (fp32, row-major for the head kernels) + saved bf16 activations / masks
//   normal pass     n_raw = -d spatial_out[:, 0] / dx (ref_nerf.py:38-43): the input-gradient chain with the seed
//                   -e_0, continued through the x_emb rows of Dense_5 and through Dense_0 into the positional
//                   embedding, whose Jacobian is applied in registers; dumps the chain states c_l
//   trunk backward  first-order: d L / d spatial_out -> dy_l chain -> dW_l = X_l^T dy_l (nerf_wgrad_kernel)
//   normal backward second-order: the normal enters the loss (normal_mse, ref_nerf.py:73), and with the ReLU masks
//                   fixed the normal pass is multilinear in the kernels: tbar_l = mask_l * (tbar_{l-1} W_l) seeded by
//                   (d emb / dx) u, u = d L / d n_raw; dW_l += tbar_{l-1}^T c_l (the same weight-gradient kernel on
//                   the tangent dump and the c dump)
//
// The head (exp / sigmoid / softplus heads, reflection, integrated directional encoding, aux losses: refnerf.hip) and
// the 273 -> 128 -> 3 directional block (dense.hip) stay separate kernels; they are 6 % of the model's FLOPs.
// Precision: bf16 operands, fp32 accumulate, fp32 bias / embedding / head inputs — the arithmetic of the dense bf16
// path that these kernels replace (precision="fp32" keeps the exact dense path).
#include "nerf_chain.h"

namespace lnrf {

constexpr int kRefLds = kRingBytes + round_up(kBiasFloats * 4, 1024);

template <int COUNT>
struct LinSeq {  // a stream consumed front to back without padding
  static constexpr int count = COUNT;
  static constexpr int at(int c) { return c; }
};
template <int LAYERS>
struct FwdPrefixSeq {  // the forward stream up to (not including) stream layer LAYERS
  static constexpr int count = fwd_cons_base(LAYERS);
  static constexpr int at(int c) { return fwd_seq(c); }
};

__device__ __forceinline__ void load_relu_masks(uint4 (&mask)[8], const char* __restrict__ save, int64_t n_tiles,
                                                int64_t tile, int lane) {
#pragma unroll
  for (int i = 0; i < 8; ++i)
    mask[i] = *reinterpret_cast<const uint4*>(save + dump_off(kSaveMask + i, tile, n_tiles, kSaveTileSlots) + lane * 16);
}
// ReLU mask of h_i (written by the trunk forward), 16 bytes per lane
__device__ __forceinline__ uint4 load_relu_mask(int i, const char* __restrict__ save, int64_t n_tiles, int64_t tile,
                                                int lane) {
  return *reinterpret_cast<const uint4*>(save + dump_off(kSaveMask + i, tile, n_tiles, kSaveTileSlots) + lane * 16);
}
__device__ __forceinline__ unsigned mask_word(const uint4& mk, int o) {
  return (o >> 1) == 0 ? mk.x : ((o >> 1) == 1 ? mk.y : ((o >> 1) == 2 ? mk.z : mk.w));
}

// ---------------------------------------------------------------------------------------------
// trunk forward: Dense_0..8 (model.py:50-56 as used by ref_nerf.py:92-99), saving what the backward passes need
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void refnerf_trunk_fwd_kernel(
    const char* __restrict__ packed, const float* __restrict__ xin_g, int64_t M, int64_t n_tiles,
    char* __restrict__ save, float* __restrict__ zout, int64_t ldz) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  {
    const float* bias_g = reinterpret_cast<const float*>(packed + kPackBiasOff);
    float* bias_l = reinterpret_cast<float*>(&smem[kBiasLdsOff]);
    for (int i = tid; i < kBiasFloats; i += kThreads) bias_l[i] = bias_g[i];
  }
  float px[3] = {0, 0, 0};
  if (valid) {
#pragma unroll
    for (int a = 0; a < 3; ++a) px[a] = xin_g[m * 3 + a];
  }
  __syncthreads();
  Ring<fwd_base(9) / kStageFrags, FwdPrefixSeq<9>> ring;
  ring.stream = packed + kPackFwdOff;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();

  bf16x8 xe[4];
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
  DumpAddr dump{save, n_tiles, tile, c, h, kSaveTileSlots};
  static_for<4>([&](auto i) { stream_store(dump.at(kSaveXin + decltype(i)::value), frag_to_bits(xe[decltype(i)::value])); });

  bf16x8 a0[16], a1[16];
  unsigned mask_bits[4] = {0u, 0u, 0u, 0u};
  auto hidden = [&](auto s_, bf16x8(&in)[16], bf16x8(&out)[16]) {
    constexpr int S = decltype(s_)::value;
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
          out[2 * o] = acc_to_frag<0, true>(acc);
          out[2 * o + 1] = acc_to_frag<1, true>(acc);
          stream_store(dump.at(kSaveH + 16 * S + 2 * o), frag_to_bits(out[2 * o]));
          stream_store(dump.at(kSaveH + 16 * S + 2 * o + 1), frag_to_bits(out[2 * o + 1]));
          mask_bits[o >> 1] |= relu_bits(out[2 * o], out[2 * o + 1]) << (16 * (o & 1));
        });
    *reinterpret_cast<uint4*>(save + dump_off(kSaveMask + S, tile, n_tiles, kSaveTileSlots) + lane * 16) =
        make_uint4(mask_bits[0], mask_bits[1], mask_bits[2], mask_bits[3]);
    mask_bits[0] = mask_bits[1] = mask_bits[2] = mask_bits[3] = 0u;
  };
  hidden(std::integral_constant<int, 0>{}, a1, a0);
  hidden(std::integral_constant<int, 1>{}, a0, a1);
  hidden(std::integral_constant<int, 2>{}, a1, a0);
  hidden(std::integral_constant<int, 3>{}, a0, a1);
  hidden(std::integral_constant<int, 4>{}, a1, a0);
  hidden(std::integral_constant<int, 5>{}, a0, a1);
  hidden(std::integral_constant<int, 6>{}, a1, a0);
  hidden(std::integral_constant<int, 7>{}, a0, a1);
  // Dense_8: linear spatial_out (ref_nerf.py:98-99), fp32, row m of a row-major matrix with ldz floats per row
  chain_layer<fwd_cons_base(8), fwd_nk(8), fwd_no(8)>(
      ring, [&](auto o_) { return bias_acc(fwd_bias_base(8) + 32 * decltype(o_)::value, h); },
      [&](auto k_) -> bf16x8 { return a1[decltype(k_)::value]; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        if (valid) {
          float* zr = zout + m * ldz + 32 * o + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(zr + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
      });
}

// One hidden step of an input-gradient chain: out = relu'(h_{l-1}) * (W_l^T in), dumped as dy_{l-1} / c_{l-1}
template <int C0, int L, class RING>
__device__ __forceinline__ void hidden_back(RING& ring, bf16x8 (&in)[16], bf16x8 (&out)[16], const uint4& mk,
                                            const DumpAddr& gd) {
  chain_layer<C0, 16, 8>(
      ring, [&](auto) { return zero_acc(); }, [&](auto k_) -> bf16x8 { return in[decltype(k_)::value]; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        const unsigned mb = mask_word(mk, o);
        out[2 * o] = masked_frag<0>(acc, mb, 16 * (o & 1));
        out[2 * o + 1] = masked_frag<1>(acc, mb, 16 * (o & 1));
        stream_store(gd.at(grad_dy_slot(L - 1) + 2 * o), frag_to_bits(out[2 * o]));
        stream_store(gd.at(grad_dy_slot(L - 1) + 2 * o + 1), frag_to_bits(out[2 * o + 1]));
      });
}

// ---------------------------------------------------------------------------------------------
// normal pass (ref_nerf.py:38-43)
// ---------------------------------------------------------------------------------------------
// J^T applied to one 32-row tile of the embedding gradient (rows ordered by nrm_x_row: registers 2i, 2i + 1 of a lane
// are the sin and the cos row of pair pg = 16 o + 8 hh + i = 10 a + f):  d sin(2^f x)/dx = 2^f cos, d cos = -2^f sin.
// The 8 pairs of a lane are consecutive frequencies of at most two coordinates: two exact sincos (first pair, and
// frequency 0 of the next coordinate) and angle doubling in between — a dependency chain with two live values instead
// of eight interleaved range reductions (which cost ~100 registers and spilled).  Doubling error <= 2^7 x 1e-7.
template <int O>
__device__ __forceinline__ void emb_bwd_tile(const f32x16& acc, const float (&px)[3], int hh, float (&nr)[3]) {
  const int base = 16 * O + 8 * hh;
  const int a0 = base / 10, f0 = base - 10 * a0;
  const int ri = 10 - f0;  // pair index at which the next coordinate starts (>= 8: not in this lane)
  const float xa0 = a0 == 0 ? px[0] : (a0 == 1 ? px[1] : px[2]);
  const float xa1 = a0 == 0 ? px[1] : px[2];
  float sc = (float)(1 << f0);
  float s, co, s1, c1;
  sincos_pe(xa0 * sc, &s, &co);
  sincos_pe(xa1, &s1, &c1);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i > 0) {
      const bool restart = i == ri;
      const float sd = 2.0f * s * co, cd = co * co - s * s;
      s = restart ? s1 : sd;
      co = restart ? c1 : cd;
      sc = restart ? 1.0f : 2.0f * sc;
    }
    const int a = i >= ri ? a0 + 1 : a0;
    const float t = base + i < 30 ? sc * (co * acc[2 * i] - s * acc[2 * i + 1]) : 0.0f;
    nr[0] += a == 0 ? t : 0.0f;
    nr[1] += a == 1 ? t : 0.0f;
    nr[2] += a == 2 ? t : 0.0f;
  }
}

__global__ __launch_bounds__(kThreads) void refnerf_normal_kernel(
    const char* __restrict__ packed, const char* __restrict__ save, const float* __restrict__ xin_g, int64_t M,
    int64_t n_tiles, char* __restrict__ cdump, float* __restrict__ nraw) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  float px[3] = {0, 0, 0};
  if (valid) {
#pragma unroll
    for (int a = 0; a < 3; ++a) px[a] = xin_g[m * 3 + a];
  }
  // masks are fetched one layer ahead instead of all eight up front: the X layers need the registers
  auto mask_of = [&](int i) { return load_relu_mask(i, save, n_tiles, tile, lane); };
  uint4 mk = mask_of(7), mk_next = mask_of(6);
  __syncthreads();
  Ring<kNrmFrags / kStageFrags, LinSeq<kNrmFrags>> ring;
  ring.stream = packed + kRefPackNrmOff;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();

  DumpAddr gd{cdump, n_tiles, tile, c, h, kGradTileSlots};
  bf16x8 a0[16], a1[16];
  // seed c_8 = -e_0: feature 0 is k slot (ks 0, h 0, j 0)
#pragma unroll
  for (int i = 0; i < 16; ++i) a1[i] = zero_frag();
  if (h == 0) a1[0][0] = (__bf16)(-1.0f);
#pragma unroll
  for (int i = 0; i < 16; ++i) stream_store(gd.at(grad_dy_slot(8) + i), frag_to_bits(a1[i]));

  float nr[3] = {0.0f, 0.0f, 0.0f};
  auto x_layer = [&](auto u_, bf16x8(&in)[16]) {
    constexpr int U = decltype(u_)::value;
    chain_layer<nrm_base(U), 16, 2>(
        ring, [&](auto) { return zero_acc(); }, [&](auto k_) -> bf16x8 { return in[decltype(k_)::value]; },
        [&](auto o_, const f32x16& acc) {
          // The sin / cos values depend on x only, so the compiler would compute them once and keep the 60 of them
          // alive from the first X layer to the second (spills); an empty asm makes each use recompute its own.
          float pl[3] = {px[0], px[1], px[2]};
          asm volatile("" : "+v"(pl[0]), "+v"(pl[1]), "+v"(pl[2]));
          emb_bwd_tile<decltype(o_)::value>(acc, pl, h, nr);
        });
  };
  auto advance_mask = [&](int next) {
    mk = mk_next;
    if (next >= 0) mk_next = mask_of(next);
  };
  hidden_back<nrm_base(0), 8>(ring, a1, a0, mk, gd);  // c_7
  advance_mask(5);
  hidden_back<nrm_base(1), 7>(ring, a0, a1, mk, gd);  // c_6
  advance_mask(4);
  hidden_back<nrm_base(2), 6>(ring, a1, a0, mk, gd);  // c_5
  advance_mask(3);
  x_layer(std::integral_constant<int, 3>{}, a0);      // x_emb rows of Dense_5 (model.py:52 concat) on c_5
  hidden_back<nrm_base(4), 5>(ring, a0, a1, mk, gd);  // c_4
  advance_mask(2);
  hidden_back<nrm_base(5), 4>(ring, a1, a0, mk, gd);  // c_3
  advance_mask(1);
  hidden_back<nrm_base(6), 3>(ring, a0, a1, mk, gd);  // c_2
  advance_mask(0);
  hidden_back<nrm_base(7), 2>(ring, a1, a0, mk, gd);  // c_1
  advance_mask(-1);
  hidden_back<nrm_base(8), 1>(ring, a0, a1, mk, gd);  // c_0
  x_layer(std::integral_constant<int, 9>{}, a1);      // Dense_0^T on c_0
#pragma unroll
  for (int a = 0; a < 3; ++a) nr[a] += __shfl_xor(nr[a], 32, 64);  // the two lane halves hold different rows
  if (h == 0 && valid) {
    nraw[m * 3 + 0] = nr[0];
    nraw[m * 3 + 1] = nr[1];
    nraw[m * 3 + 2] = nr[2];
  }
}

// ---------------------------------------------------------------------------------------------
// trunk backward, chain part: d L / d spatial_out [M, ldg] (fp32) -> dy_8 .. dy_0 dumps
// ---------------------------------------------------------------------------------------------
constexpr int kHiddenBwdOff = bwd_base(2);                 // fragments of T0 / T1 (unused by this model)
constexpr int kHiddenBwdFrags = kBwdFrags - kHiddenBwdOff;  // 8 x 128
static_assert(kHiddenBwdOff % kStageFrags == 0 && kHiddenBwdFrags == 1024, "hidden part of the transposed stream");

__global__ __launch_bounds__(kThreads) void refnerf_trunk_bwd_chain_kernel(
    const char* __restrict__ packed, const char* __restrict__ save, const float* __restrict__ g_z, int64_t ldg,
    int64_t M, int64_t n_tiles, char* __restrict__ gdump) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  uint4 mask[8];
  load_relu_masks(mask, save, n_tiles, tile, lane);
  bf16x8 a0[16], a1[16];
  // seed dy_8 = bf16(d L / d spatial_out): k slot (ks, h, j) <-> feature 16 ks + 8 (j >> 2) + 4 h + (j & 3)
  static_for<16>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    float4 lo = make_float4(0, 0, 0, 0), hi = make_float4(0, 0, 0, 0);
    if (valid) {
      const float* gr = g_z + m * ldg + 16 * ks + 4 * h;
      lo = *reinterpret_cast<const float4*>(gr);
      hi = *reinterpret_cast<const float4*>(gr + 8);
    }
    a1[ks][0] = (__bf16)lo.x; a1[ks][1] = (__bf16)lo.y; a1[ks][2] = (__bf16)lo.z; a1[ks][3] = (__bf16)lo.w;
    a1[ks][4] = (__bf16)hi.x; a1[ks][5] = (__bf16)hi.y; a1[ks][6] = (__bf16)hi.z; a1[ks][7] = (__bf16)hi.w;
  });
  __syncthreads();
  Ring<kHiddenBwdFrags / kStageFrags, LinSeq<kHiddenBwdFrags>> ring;
  ring.stream = packed + kPackBwdOff + (int64_t)kHiddenBwdOff * kFragBytes;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();
  DumpAddr gd{gdump, n_tiles, tile, c, h, kGradTileSlots};
#pragma unroll
  for (int i = 0; i < 16; ++i) stream_store(gd.at(grad_dy_slot(8) + i), frag_to_bits(a1[i]));
  hidden_back<0 * 128, 8>(ring, a1, a0, mask[7], gd);
  hidden_back<1 * 128, 7>(ring, a0, a1, mask[6], gd);
  hidden_back<2 * 128, 6>(ring, a1, a0, mask[5], gd);
  hidden_back<3 * 128, 5>(ring, a0, a1, mask[4], gd);
  hidden_back<4 * 128, 4>(ring, a1, a0, mask[3], gd);
  hidden_back<5 * 128, 3>(ring, a0, a1, mask[2], gd);
  hidden_back<6 * 128, 2>(ring, a1, a0, mask[1], gd);
  hidden_back<7 * 128, 1>(ring, a0, a1, mask[0], gd);
}

// seed of the layer-stationary trunk backward: dy_8 = bf16(d L / d spatial_out) as the 16 fragments of the dump's dy8 slots
__global__ __launch_bounds__(kThreads) void refnerf_dy8_kernel(const float* __restrict__ g_z, int64_t ldg, int64_t M,
                                                               int64_t n_tiles, char* __restrict__ gdump) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  DumpAddr gd{gdump, n_tiles, tile, c, h, kGradTileSlots};
  static_for<16>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    float4 lo = make_float4(0, 0, 0, 0), hi = make_float4(0, 0, 0, 0);
    if (valid) {
      const float* gr = g_z + m * ldg + 16 * ks + 4 * h;
      lo = *reinterpret_cast<const float4*>(gr);
      hi = *reinterpret_cast<const float4*>(gr + 8);
    }
    bf16x8 f;
    f[0] = (__bf16)lo.x; f[1] = (__bf16)lo.y; f[2] = (__bf16)lo.z; f[3] = (__bf16)lo.w;
    f[4] = (__bf16)hi.x; f[5] = (__bf16)hi.y; f[6] = (__bf16)hi.z; f[7] = (__bf16)hi.w;
    stream_store(gd.at(grad_dy_slot(8) + ks), frag_to_bits(f));
  });
}

// ---------------------------------------------------------------------------------------------
// normal backward, tangent chain: ubar_e = (d emb / dx) u, tbar_l = relu'(h_l) * (tbar_{l-1} W_l), l = 0..7
// (forward weight stream, no bias); dumps in the layout of the forward save buffer (x_emb and h slots)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void refnerf_tangent_kernel(
    const char* __restrict__ packed, const char* __restrict__ save, const float* __restrict__ xin_g,
    const float* __restrict__ u_g, int64_t M, int64_t n_tiles, char* __restrict__ tdump) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  float px[3] = {0, 0, 0}, pu[3] = {0, 0, 0};
  if (valid) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      px[a] = xin_g[m * 3 + a];
      pu[a] = u_g[m * 3 + a];
    }
  }
  uint4 mask[8];
  load_relu_masks(mask, save, n_tiles, tile, lane);
  __syncthreads();
  Ring<fwd_base(8) / kStageFrags, FwdPrefixSeq<8>> ring;
  ring.stream = packed + kPackFwdOff;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();

  DumpAddr td{tdump, n_tiles, tile, c, h, kSaveTileSlots};
  bf16x8 xe[4];
  static_for<4>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int p = 4 * ks + pp;
      float ts = 0.0f, tc = 0.0f;
      if (p < 15) {
        const int pg = 15 * h + p;
        const int cd = pg / 10, f = pg - 10 * cd;
        const float v = cd == 0 ? px[0] : (cd == 1 ? px[1] : px[2]);
        const float uu = cd == 0 ? pu[0] : (cd == 1 ? pu[1] : pu[2]);
        const float sc = (float)(1 << f);
        float s, co;
        sincos_pe(v * sc, &s, &co);
        ts = sc * co * uu;   // d sin(2^f x) = 2^f cos(2^f x) dx
        tc = -sc * s * uu;   // d cos(2^f x) = -2^f sin(2^f x) dx
      }
      xe[ks][2 * pp] = (__bf16)ts;
      xe[ks][2 * pp + 1] = (__bf16)tc;
    }
    stream_store(td.at(kSaveXin + ks), frag_to_bits(xe[ks]));
  });
  bf16x8 a0[16], a1[16];
  auto hidden = [&](auto s_, bf16x8(&in)[16], bf16x8(&out)[16]) {
    constexpr int S = decltype(s_)::value;
    chain_layer<fwd_cons_base(S), fwd_nk(S), fwd_no(S)>(
        ring, [&](auto) { return zero_acc(); },
        [&](auto k_) -> bf16x8 {
          constexpr int ks = decltype(k_)::value;
          if constexpr (S == 0) return xe[ks];
          else if constexpr (ks < 16) return in[ks];
          else return xe[ks - 16];
        },
        [&](auto o_, const f32x16& acc) {
          constexpr int o = decltype(o_)::value;
          const unsigned mb = mask_word(mask[S], o);
          out[2 * o] = masked_frag<0>(acc, mb, 16 * (o & 1));
          out[2 * o + 1] = masked_frag<1>(acc, mb, 16 * (o & 1));
          stream_store(td.at(kSaveH + 16 * S + 2 * o), frag_to_bits(out[2 * o]));
          stream_store(td.at(kSaveH + 16 * S + 2 * o + 1), frag_to_bits(out[2 * o + 1]));
        });
  };
  hidden(std::integral_constant<int, 0>{}, a1, a0);
  hidden(std::integral_constant<int, 1>{}, a0, a1);
  hidden(std::integral_constant<int, 2>{}, a1, a0);
  hidden(std::integral_constant<int, 3>{}, a0, a1);
  hidden(std::integral_constant<int, 4>{}, a1, a0);
  hidden(std::integral_constant<int, 5>{}, a0, a1);
  hidden(std::integral_constant<int, 6>{}, a1, a0);
  hidden(std::integral_constant<int, 7>{}, a0, a1);
}


// ---------------------------------------------------------------------------------------------
// directional block (ref_nerf.py:100-107): dir_out = Dense_10(relu(Dense_9([spatial_out, IDE, -d.n])))
// One wave = 32 evaluations; the input row (fp32, written by the trunk forward and the head kernel) is converted to
// bf16 fragments in registers; saves the input fragments, relu(Dense_9) and its mask for the backward.
// ---------------------------------------------------------------------------------------------
constexpr int kDirLds = kRingBytes + 1024;
struct DirFwdSeq {
  static constexpr int count = 80;
  static constexpr int at(int c) { return dir_fwd_seq(c); }
};
struct DirBwdSeq {
  static constexpr int count = 76;
  static constexpr int at(int c) { return dir_bwd_seq(c); }
};

__global__ __launch_bounds__(kThreads) void refnerf_dir_fwd_kernel(
    const char* __restrict__ packed, const float* __restrict__ dir_in, int64_t ld, int64_t M, int64_t n_tiles,
    char* __restrict__ dsave, float* __restrict__ dir_out) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  {
    const float* bias_g = reinterpret_cast<const float*>(packed + kRefPackDirBiasOff);
    float* bias_l = reinterpret_cast<float*>(&smem[kBiasLdsOff]);
    for (int i = tid; i < kDirBiasFloats; i += kThreads) bias_l[i] = bias_g[i];
  }
  // input fragments: k slot (ks, h, j) <-> feature 16 ks + 8 (j >> 2) + 4 h + (j & 3); features >= 273 are zero
  bf16x8 xin[18];
  static_for<18>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    float4 lo = make_float4(0, 0, 0, 0), hi = make_float4(0, 0, 0, 0);
    if (valid) {
      const float* row = dir_in + m * ld + 16 * ks + 4 * h;
      if constexpr (ks < 17) {
        lo = *reinterpret_cast<const float4*>(row);
        hi = *reinterpret_cast<const float4*>(row + 8);
      } else {
        if (h == 0) lo.x = row[0];  // feature 272 = -d.n; 273.. do not exist
      }
    }
    xin[ks][0] = (__bf16)lo.x; xin[ks][1] = (__bf16)lo.y; xin[ks][2] = (__bf16)lo.z; xin[ks][3] = (__bf16)lo.w;
    xin[ks][4] = (__bf16)hi.x; xin[ks][5] = (__bf16)hi.y; xin[ks][6] = (__bf16)hi.z; xin[ks][7] = (__bf16)hi.w;
  });
  __syncthreads();
  Ring<kDirFwdFrags / kStageFrags, DirFwdSeq> ring;
  ring.stream = packed + kRefPackDirFwdOff;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();
  DumpAddr dump{dsave, n_tiles, tile, c, h, kDirSaveTileSlots};
  static_for<18>([&](auto i) { stream_store(dump.at(kDirSaveXin + decltype(i)::value), frag_to_bits(xin[decltype(i)::value])); });
  bf16x8 hcol[8];
  unsigned mask_bits[2] = {0u, 0u};
  chain_layer<0, 18, 4>(
      ring, [&](auto o_) { return bias_acc(32 * decltype(o_)::value, h); },
      [&](auto k_) -> bf16x8 { return xin[decltype(k_)::value]; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        hcol[2 * o] = acc_to_frag<0, true>(acc);
        hcol[2 * o + 1] = acc_to_frag<1, true>(acc);
        stream_store(dump.at(kDirSaveH + 2 * o), frag_to_bits(hcol[2 * o]));
        stream_store(dump.at(kDirSaveH + 2 * o + 1), frag_to_bits(hcol[2 * o + 1]));
        mask_bits[o >> 1] |= relu_bits(hcol[2 * o], hcol[2 * o + 1]) << (16 * (o & 1));
      });
  *reinterpret_cast<uint4*>(dsave + dump_off(kDirSaveMask, tile, n_tiles, kDirSaveTileSlots) + lane * 16) =
      make_uint4(mask_bits[0], mask_bits[1], 0u, 0u);
  chain_layer<72, 8, 1>(
      ring, [&](auto) { return bias_acc(128, h); }, [&](auto k_) -> bf16x8 { return hcol[decltype(k_)::value]; },
      [&](auto, const f32x16& acc) {
        if (h == 0 && valid) {
          dir_out[m * 3 + 0] = acc[0];
          dir_out[m * 3 + 1] = acc[1];
          dir_out[m * 3 + 2] = acc[2];
        }
      });
}

// backward: g_dir_out [M,3] -> dy10, dy9 dumps (weight gradients) and g_dir_in [M, ld] = dy9 Dense_9^T (fp32, all 273 columns)
__global__ __launch_bounds__(kThreads) void refnerf_dir_bwd_kernel(
    const char* __restrict__ packed, const char* __restrict__ dsave, const float* __restrict__ g_do, int64_t M,
    int64_t n_tiles, char* __restrict__ gdump, float* __restrict__ g_dir_in, int64_t ld) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  bf16x8 dy10 = zero_frag();
  if (valid && h == 0) {
    dy10[0] = (__bf16)g_do[m * 3 + 0];
    dy10[1] = (__bf16)g_do[m * 3 + 1];
    dy10[2] = (__bf16)g_do[m * 3 + 2];
  }
  const uint4 mk = *reinterpret_cast<const uint4*>(dsave + dump_off(kDirSaveMask, tile, n_tiles, kDirSaveTileSlots) + lane * 16);
  __syncthreads();
  Ring<kDirBwdFrags / kStageFrags, DirBwdSeq> ring;
  ring.stream = packed + kRefPackDirBwdOff;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();
  DumpAddr gd{gdump, n_tiles, tile, c, h, kDirGradTileSlots};
  stream_store(gd.at(kDirGradDy10), frag_to_bits(dy10));
  stream_store(gd.at(kDirGradDy10 + 1), frag_to_bits(zero_frag()));
  bf16x8 dy9[8];
  chain_layer<0, 1, 4>(
      ring, [&](auto) { return zero_acc(); }, [&](auto) -> bf16x8 { return dy10; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        const unsigned mb = (o >> 1) == 0 ? mk.x : mk.y;
        dy9[2 * o] = masked_frag<0>(acc, mb, 16 * (o & 1));
        dy9[2 * o + 1] = masked_frag<1>(acc, mb, 16 * (o & 1));
        stream_store(gd.at(kDirGradDy9 + 2 * o), frag_to_bits(dy9[2 * o]));
        stream_store(gd.at(kDirGradDy9 + 2 * o + 1), frag_to_bits(dy9[2 * o + 1]));
      });
  chain_layer<4, 8, 9>(
      ring, [&](auto) { return zero_acc(); }, [&](auto k_) -> bf16x8 { return dy9[decltype(k_)::value]; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        if (valid) {
          float* gr = g_dir_in + m * ld + 32 * o + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            // columns 32 o + 8 g + 4 h .. + 3; the last tile ends at column 272 (the row has ld >= 276 floats)
            if (32 * o + 8 * g + 4 * h + 3 < (int)ld)
              *reinterpret_cast<float4*>(gr + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
          }
        }
      });
}

// ---------------------------------------------------------------------------------------------
// Split-precision render path ("bf16x3"; forwards without a backward: rendering, evaluation, model.apply).  The reference
// runs ref_nerf.py:35-77 in fp32; plain bf16 operands leave the rendered colour up to 5e-2 off.  Here the trunk forward,
// the normal pass and the directional block carry every fp32 operand as a bf16 pair (hi, lo) and form lo*hi + hi*lo +
// hi*hi per product (fused_chain.h), as nerf_fwd_split_kernel does for NeRFModel.
//   refnerf_render_split_kernel: trunk forward (spatial_out, fp32) and, with the ReLU masks kept in registers, the
//                                normal pass n_raw = -d spatial_out[:, 0] / dx of the same tile — one launch, nothing saved
//   refnerf_dir_fwd_split_kernel: Dense_10(relu(Dense_9([spatial_out, IDE, -d.n])))
// Blob (lnrf_refnerf_render_pack): [hi, lo] pairs of the forward stream of Dense_0..8 | fp32 biases | pairs of the
// normal-pass stream | pairs of the directional forward stream | its fp32 biases.
// ---------------------------------------------------------------------------------------------
constexpr int kRef3FwdFrags = fwd3_base(9);
constexpr int64_t kRef3FwdOff = 0;
constexpr int64_t kRef3BiasOff = (int64_t)kRef3FwdFrags * kFragBytes;
constexpr int64_t kRef3NrmOff = kRef3BiasOff + round_up(kBiasFloats * 4, 1024);
constexpr int64_t kRef3DirOff = kRef3NrmOff + 2 * (int64_t)kNrmFrags * kFragBytes;
constexpr int64_t kRef3DirBiasOff = kRef3DirOff + 2 * (int64_t)kDirFwdFrags * kFragBytes;
constexpr int64_t kRef3Bytes = kRef3DirBiasOff + 1024;
static_assert(kRef3FwdFrags % kStageFrags == 0 && (2 * kNrmFrags) % kStageFrags == 0 && (2 * kDirFwdFrags) % kStageFrags == 0,
              "split streams are whole stages");
struct RefFwd3Seq {
  static constexpr int count = 2 * fwd_cons_base(9);
  static constexpr int at(int c) { return fwd3_seq(c); }
};
struct DirFwd3Seq {
  static constexpr int count = 160;
  static constexpr int at(int c) { return 2 * dir_fwd_seq(c >> 1) + (c & 1); }
};

// 16 mask bits of one out tile: bit q set <=> accumulator register q is positive (the ReLU passes it)
__device__ __forceinline__ unsigned acc_positive_bits(const f32x16& acc) {
  unsigned bits = 0u;
#pragma unroll
  for (int q = 0; q < 16; ++q) bits |= (acc[q] > 0.0f ? 1u : 0u) << q;
  return bits;
}
// dh * relu'(h) in split precision: registers 8S..8S+7 of the tile whose 16 mask bits start at bit `shift`
template <int S>
__device__ __forceinline__ void masked_frag_split(const f32x16& acc, unsigned bits, int shift, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) split_store(((bits >> (shift + 8 * S + j)) & 1u) ? acc[8 * S + j] : 0.0f, hi, lo, j);
}

__global__ __launch_bounds__(kSplitThreads) void refnerf_render_split_kernel(
    const char* __restrict__ packed3, const float* __restrict__ xin_g, int64_t M, float* __restrict__ zout, int64_t ldz,
    float* __restrict__ nraw, uint4* __restrict__ mask_buf) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kSplitWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  {
    const float* bias_g = reinterpret_cast<const float*>(packed3 + kRef3BiasOff);
    float* bias_l = reinterpret_cast<float*>(&smem[kBiasLdsOff]);
    for (int i = tid; i < kBiasFloats; i += kSplitThreads) bias_l[i] = bias_g[i];
  }
  float px[3] = {0, 0, 0};
  if (valid) {
#pragma unroll
    for (int a = 0; a < 3; ++a) px[a] = xin_g[m * 3 + a];
  }
  __syncthreads();
  bf16x8 a0h[16], a0l[16], a1h[16], a1l[16];
  // ReLU masks of h_0..h_7 of this wave's tile: written by the forward half, read back one layer ahead by the normal pass
  // (8 KiB per tile in mask_buf: kept in registers they tip the kernel into 3.6 KB of scratch per lane)
  uint4* my_masks = mask_buf + tile * (8 * 64) + lane;
  {
    // ---- trunk forward (model.py:50-56 as used by ref_nerf.py:92-99) ----
    Ring<kRef3FwdFrags / kStageFrags, RefFwd3Seq, kSplitWaves> ring;
    ring.stream = packed3 + kRef3FwdOff;
    ring.wave = wave;
    ring.lane = lane;
    ring.prologue();
    bf16x8 xe_hi[4], xe_lo[4];
    static_for<4>([&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int p = 4 * ks + pp;
        float sn = 0.0f, co = 0.0f;
        if (p < 15) {
          const int pg = 15 * h + p;
          const int cd = pg / 10, f = pg - 10 * cd;
          const float v = cd == 0 ? px[0] : (cd == 1 ? px[1] : px[2]);
          sincos_pe(v * (float)(1 << f), &sn, &co);
        }
        split_store(sn, xe_hi[ks], xe_lo[ks], 2 * pp);
        split_store(co, xe_hi[ks], xe_lo[ks], 2 * pp + 1);
      }
    });
    auto hidden = [&](auto s_, bf16x8(&inh)[16], bf16x8(&inl)[16], bf16x8(&outh)[16], bf16x8(&outl)[16]) {
      constexpr int S = decltype(s_)::value;
      unsigned mb[4] = {0u, 0u, 0u, 0u};
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
            acc_to_frag_split<0, true>(acc, outh[2 * o], outl[2 * o]);
            acc_to_frag_split<1, true>(acc, outh[2 * o + 1], outl[2 * o + 1]);
            mb[o >> 1] |= acc_positive_bits(acc) << (16 * (o & 1));
          });
      my_masks[S * 64] = make_uint4(mb[0], mb[1], mb[2], mb[3]);
    };
    hidden(std::integral_constant<int, 0>{}, a1h, a1l, a0h, a0l);
    hidden(std::integral_constant<int, 1>{}, a0h, a0l, a1h, a1l);
    hidden(std::integral_constant<int, 2>{}, a1h, a1l, a0h, a0l);
    hidden(std::integral_constant<int, 3>{}, a0h, a0l, a1h, a1l);
    hidden(std::integral_constant<int, 4>{}, a1h, a1l, a0h, a0l);
    hidden(std::integral_constant<int, 5>{}, a0h, a0l, a1h, a1l);
    hidden(std::integral_constant<int, 6>{}, a1h, a1l, a0h, a0l);
    hidden(std::integral_constant<int, 7>{}, a0h, a0l, a1h, a1l);
    // Dense_8: linear spatial_out (ref_nerf.py:98-99), fp32 rows of ldz floats
    chain_layer_split<fwd_cons_base(8), fwd_nk(8), fwd_no(8)>(
        ring, [&](auto o_) { return bias_acc(fwd_bias_base(8) + 32 * decltype(o_)::value, h); },
        [&](auto k_) -> bf16x8 { return a1h[decltype(k_)::value]; },
        [&](auto k_) -> bf16x8 { return a1l[decltype(k_)::value]; },
        [&](auto o_, const f32x16& acc) {
          constexpr int o = decltype(o_)::value;
          if (valid) {
            float* zr = zout + m * ldz + 32 * o + 4 * h;
#pragma unroll
            for (int g = 0; g < 4; ++g)
              *reinterpret_cast<float4*>(zr + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
          }
        });
  }
  __syncthreads();  // every wave is done with the forward stream's ring slots
  {
    // ---- normal pass (ref_nerf.py:38-43): c_8 = -e_0, c_{l-1} = relu'(h_{l-1}) o (W_l^T c_l), into the embedding ----
    Ring<2 * kNrmFrags / kStageFrags, LinSeq<2 * kNrmFrags>, kSplitWaves> ring;
    ring.stream = packed3 + kRef3NrmOff;
    ring.wave = wave;
    ring.lane = lane;
    ring.prologue();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      a1h[i] = zero_frag();
      a1l[i] = zero_frag();
    }
    if (h == 0) a1h[0][0] = (__bf16)(-1.0f);
    float nr[3] = {0.0f, 0.0f, 0.0f};
    uint4 mk_next = my_masks[7 * 64];
    auto back = [&](auto u_, auto l_, bf16x8(&inh)[16], bf16x8(&inl)[16], bf16x8(&outh)[16], bf16x8(&outl)[16]) {
      constexpr int U = decltype(u_)::value, LL = decltype(l_)::value;  // stream layer, Dense index: out = c_{LL-1}
      const uint4 mk = mk_next;
      if constexpr (LL >= 2) mk_next = my_masks[(LL - 2) * 64];
      chain_layer_split<nrm_base(U), 16, 8>(
          ring, [&](auto) { return zero_acc(); }, [&](auto k_) -> bf16x8 { return inh[decltype(k_)::value]; },
          [&](auto k_) -> bf16x8 { return inl[decltype(k_)::value]; },
          [&](auto o_, const f32x16& acc) {
            constexpr int o = decltype(o_)::value;
            const unsigned mb = mask_word(mk, o);
            masked_frag_split<0>(acc, mb, 16 * (o & 1), outh[2 * o], outl[2 * o]);
            masked_frag_split<1>(acc, mb, 16 * (o & 1), outh[2 * o + 1], outl[2 * o + 1]);
          });
    };
    auto x_layer = [&](auto u_, bf16x8(&inh)[16], bf16x8(&inl)[16]) {
      constexpr int U = decltype(u_)::value;
      chain_layer_split<nrm_base(U), 16, 2>(
          ring, [&](auto) { return zero_acc(); }, [&](auto k_) -> bf16x8 { return inh[decltype(k_)::value]; },
          [&](auto k_) -> bf16x8 { return inl[decltype(k_)::value]; },
          [&](auto o_, const f32x16& acc) {
            float pl[3] = {px[0], px[1], px[2]};
            asm volatile("" : "+v"(pl[0]), "+v"(pl[1]), "+v"(pl[2]));  // see refnerf_normal_kernel
            emb_bwd_tile<decltype(o_)::value>(acc, pl, h, nr);
          });
    };
    using I = std::integral_constant<int, 0>;
    (void)sizeof(I);
    back(std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{}, a1h, a1l, a0h, a0l);  // c_7
    back(std::integral_constant<int, 1>{}, std::integral_constant<int, 7>{}, a0h, a0l, a1h, a1l);  // c_6
    back(std::integral_constant<int, 2>{}, std::integral_constant<int, 6>{}, a1h, a1l, a0h, a0l);  // c_5
    x_layer(std::integral_constant<int, 3>{}, a0h, a0l);                                           // x_emb rows of Dense_5
    back(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{}, a0h, a0l, a1h, a1l);  // c_4
    back(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{}, a1h, a1l, a0h, a0l);  // c_3
    back(std::integral_constant<int, 6>{}, std::integral_constant<int, 3>{}, a0h, a0l, a1h, a1l);  // c_2
    back(std::integral_constant<int, 7>{}, std::integral_constant<int, 2>{}, a1h, a1l, a0h, a0l);  // c_1
    back(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{}, a0h, a0l, a1h, a1l);  // c_0
    x_layer(std::integral_constant<int, 9>{}, a1h, a1l);                                           // Dense_0^T on c_0
#pragma unroll
    for (int a = 0; a < 3; ++a) nr[a] += __shfl_xor(nr[a], 32, 64);
    if (h == 0 && valid) {
      nraw[m * 3 + 0] = nr[0];
      nraw[m * 3 + 1] = nr[1];
      nraw[m * 3 + 2] = nr[2];
    }
  }
}

__global__ __launch_bounds__(kSplitThreads) void refnerf_dir_fwd_split_kernel(
    const char* __restrict__ packed3, const float* __restrict__ dir_in, int64_t ld, int64_t M, float* __restrict__ dir_out) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * kSplitWaves + wave;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;
  {
    const float* bias_g = reinterpret_cast<const float*>(packed3 + kRef3DirBiasOff);
    float* bias_l = reinterpret_cast<float*>(&smem[kBiasLdsOff]);
    for (int i = tid; i < kDirBiasFloats; i += kSplitThreads) bias_l[i] = bias_g[i];
  }
  bf16x8 xh[18], xl[18];
  static_for<18>([&](auto ks_) {
    constexpr int ks = decltype(ks_)::value;
    float4 lo = make_float4(0, 0, 0, 0), hi = make_float4(0, 0, 0, 0);
    if (valid) {
      const float* row = dir_in + m * ld + 16 * ks + 4 * h;
      if constexpr (ks < 17) {
        lo = *reinterpret_cast<const float4*>(row);
        hi = *reinterpret_cast<const float4*>(row + 8);
      } else {
        if (h == 0) lo.x = row[0];  // feature 272 = -d.n; 273.. do not exist
      }
    }
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) split_store(v[j], xh[ks], xl[ks], j);
  });
  __syncthreads();
  Ring<2 * kDirFwdFrags / kStageFrags, DirFwd3Seq, kSplitWaves> ring;
  ring.stream = packed3 + kRef3DirOff;
  ring.wave = wave;
  ring.lane = lane;
  ring.prologue();
  bf16x8 hh_[8], hl_[8];
  chain_layer_split<0, 18, 4>(
      ring, [&](auto o_) { return bias_acc(32 * decltype(o_)::value, h); },
      [&](auto k_) -> bf16x8 { return xh[decltype(k_)::value]; }, [&](auto k_) -> bf16x8 { return xl[decltype(k_)::value]; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        acc_to_frag_split<0, true>(acc, hh_[2 * o], hl_[2 * o]);
        acc_to_frag_split<1, true>(acc, hh_[2 * o + 1], hl_[2 * o + 1]);
      });
  chain_layer_split<72, 8, 1>(
      ring, [&](auto) { return bias_acc(128, h); }, [&](auto k_) -> bf16x8 { return hh_[decltype(k_)::value]; },
      [&](auto k_) -> bf16x8 { return hl_[decltype(k_)::value]; },
      [&](auto, const f32x16& acc) {
        if (h == 0 && valid) {
          dir_out[m * 3 + 0] = acc[0];
          dir_out[m * 3 + 1] = acc[1];
          dir_out[m * 3 + 2] = acc[2];
        }
      });
}

__global__ void refnerf_render_pack_kernel(const float* __restrict__ params, char* __restrict__ packed3) {
  const int64_t n_f = (int64_t)kRef3FwdFrags * 512, n_n = 2 * (int64_t)kNrmFrags * 512, n_d = 2 * (int64_t)kDirFwdFrags * 512;
  const int64_t total = n_f + kBiasFloats + n_n + n_d + kDirBiasFloats;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    int idx = -1;
    bool lo_part = false;
    __bf16* dst = nullptr;
    if (e < n_f) {  // pairs of the forward stream of Dense_0..8 (layout: nerf_layout.h fwd3_base)
      const int g = (int)(e >> 9), lane = (int)((e >> 3) & 63), j = (int)(e & 7);
      int s = 0;
      for (int i = 1; i < 9; ++i)
        if (g >= fwd3_base(i)) s = i;
      const int loc = g - fwd3_base(s);
      if (loc < 2 * fwd_nk(s) * fwd_no(s)) idx = fwd_weight_index(s, (loc >> 1) / fwd_nk(s), (loc >> 1) % fwd_nk(s), lane, j);
      lo_part = loc & 1;
      dst = reinterpret_cast<__bf16*>(packed3 + kRef3FwdOff) + e;
    } else if (e < n_f + kBiasFloats) {
      const int i = (int)(e - n_f);
      int s = 0;
      for (int k = 1; k < kFwdLayers; ++k)
        if (i >= fwd_bias_base(k)) s = k;
      const int bi = s <= 8 ? fwd_bias_index(s, i - fwd_bias_base(s)) : -1;
      reinterpret_cast<float*>(packed3 + kRef3BiasOff)[i] = bi >= 0 ? params[bi] : 0.0f;
      continue;
    } else if (e < n_f + kBiasFloats + n_n) {
      const int64_t ee = e - n_f - kBiasFloats;
      const int gg = (int)(ee >> 9), lane = (int)((ee >> 3) & 63), j = (int)(ee & 7);
      const int g = gg >> 1;
      int u = 0;
      for (int i = 1; i < kNrmLayers; ++i)
        if (g >= nrm_base(i)) u = i;
      const int loc = g - nrm_base(u);
      idx = nrm_weight_index(u, loc / nrm_nk(u), loc % nrm_nk(u), lane, j);
      lo_part = gg & 1;
      dst = reinterpret_cast<__bf16*>(packed3 + kRef3NrmOff) + ee;
    } else if (e < n_f + kBiasFloats + n_n + n_d) {
      const int64_t ee = e - n_f - kBiasFloats - n_n;
      const int gg = (int)(ee >> 9), lane = (int)((ee >> 3) & 63), j = (int)(ee & 7);
      idx = dir_fwd_weight_index(gg >> 1, lane, j);
      lo_part = gg & 1;
      dst = reinterpret_cast<__bf16*>(packed3 + kRef3DirOff) + ee;
    } else {
      const int i = (int)(e - n_f - kBiasFloats - n_n - n_d);
      const int bi = dir_bias_index(i);
      reinterpret_cast<float*>(packed3 + kRef3DirBiasOff)[i] = bi >= 0 ? params[bi] : 0.0f;
      continue;
    }
    const float w = idx >= 0 ? params[idx] : 0.0f;
    const __bf16 hi = (__bf16)w;
    *dst = lo_part ? (__bf16)(w - (float)hi) : hi;
  }
}

// ---------------------------------------------------------------------------------------------
// packing: the NeRFModel streams restricted to the trunk (head fragments zero) + the normal-pass stream
// ---------------------------------------------------------------------------------------------
__global__ void refnerf_pack_kernel(const float* __restrict__ params, char* __restrict__ packed) {
  const int64_t total_f = (int64_t)kFwdFrags * 512;
  const int64_t total_b = (int64_t)kBwdFrags * 512;
  const int64_t total_n = (int64_t)kNrmFrags * 512;
  const int64_t total_d = (int64_t)(kDirFwdFrags + kDirBwdFrags) * 512;
  const int64_t total = total_f + total_b + kBiasFloats + total_n + total_d + kDirBiasFloats;
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
        if (s <= 8 && loc < fwd_nk(s) * fwd_no(s)) idx = fwd_weight_index(s, loc / fwd_nk(s), loc % fwd_nk(s), lane, j);
      } else {
        int t = 0;
        for (int i = 1; i < kBwdLayers; ++i)
          if (g >= bwd_base(i)) t = i;
        const int loc = g - bwd_base(t);
        if (t >= 2 && loc < bwd_nk(t) * bwd_no(t)) idx = bwd_weight_index(t, loc / bwd_nk(t), loc % bwd_nk(t), lane, j);
      }
      const float v = idx >= 0 ? params[idx] : 0.0f;
      reinterpret_cast<__bf16*>(packed + (fwd ? kPackFwdOff : kPackBwdOff))[ee] = (__bf16)v;
    } else if (e < total_f + total_b + kBiasFloats) {
      const int i = (int)(e - total_f - total_b);
      int s = 0;
      for (int k = 1; k < kFwdLayers; ++k)
        if (i >= fwd_bias_base(k)) s = k;
      const int idx = s <= 8 ? fwd_bias_index(s, i - fwd_bias_base(s)) : -1;
      reinterpret_cast<float*>(packed + kPackBiasOff)[i] = idx >= 0 ? params[idx] : 0.0f;
    } else if (e >= total_f + total_b + kBiasFloats + total_n) {  // directional block
      const int64_t ee = e - (total_f + total_b + kBiasFloats + total_n);
      if (ee < total_d) {
        const bool fwd = ee < (int64_t)kDirFwdFrags * 512;
        const int64_t e2 = fwd ? ee : ee - (int64_t)kDirFwdFrags * 512;
        const int g = (int)(e2 >> 9), lane = (int)((e2 >> 3) & 63), j = (int)(e2 & 7);
        const int idx = fwd ? dir_fwd_weight_index(g, lane, j) : dir_bwd_weight_index(g, lane, j);
        reinterpret_cast<__bf16*>(packed + (fwd ? kRefPackDirFwdOff : kRefPackDirBwdOff))[e2] =
            (__bf16)(idx >= 0 ? params[idx] : 0.0f);
      } else {
        const int i = (int)(ee - total_d);
        const int idx = dir_bias_index(i);
        reinterpret_cast<float*>(packed + kRefPackDirBiasOff)[i] = idx >= 0 ? params[idx] : 0.0f;
      }
    } else {
      const int64_t ee = e - total_f - total_b - kBiasFloats;
      const int g = (int)(ee >> 9), lane = (int)((ee >> 3) & 63), j = (int)(ee & 7);
      int u = 0;
      for (int i = 1; i < kNrmLayers; ++i)
        if (g >= nrm_base(i)) u = i;
      const int loc = g - nrm_base(u);
      const int idx = nrm_weight_index(u, loc / nrm_nk(u), loc % nrm_nk(u), lane, j);
      reinterpret_cast<__bf16*>(packed + kRefPackNrmOff)[ee] = (__bf16)(idx >= 0 ? params[idx] : 0.0f);
    }
  }
}

}  // namespace lnrf

using namespace lnrf;

template <class K>
static int set_lds(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     bytes);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(max dynamic LDS)");
  return LNRF_OK;
}
static inline dim3 tile_grid(int64_t n_tiles) { return dim3((unsigned)(n_tiles / kWaves)); }

extern "C" int64_t lnrf_refnerf_render_packed_bytes(void) { return kRef3Bytes; }

extern "C" int lnrf_refnerf_render_pack(const float* params, void* packed_split, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(params && packed_split, "null pointer");
  hipLaunchKernelGGL(refnerf_render_pack_kernel, dim3(1024), dim3(256), 0, as_stream(stream), params, (char*)packed_split);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int64_t lnrf_refnerf_trunk_normal_split_scratch_bytes(int64_t m) {
  if (m < 0) return -1;
  const int64_t tiles = ((m + kTileCols - 1) / kTileCols + kSplitWaves - 1) / kSplitWaves * kSplitWaves;
  return tiles * 8 * 64 * (int64_t)sizeof(uint4);
}

extern "C" int lnrf_refnerf_trunk_normal_split(const void* packed_split, const float* x, int64_t m, float* spatial_out,
                                               int64_t ld, float* n_raw, void* scratch, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(packed_split && x && spatial_out && n_raw && scratch, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && ld >= 256 && ld % 4 == 0 && ((uintptr_t)spatial_out & 15) == 0,
                 "spatial_out rows must be 16-byte aligned (ld a multiple of 4, >= 256)");
  if (m == 0) return LNRF_OK;
  int rc = set_lds(refnerf_render_split_kernel, kRefLds);
  if (rc) return rc;
  const int64_t tiles = (m + kTileCols - 1) / kTileCols;
  hipLaunchKernelGGL(refnerf_render_split_kernel, dim3((unsigned)((tiles + kSplitWaves - 1) / kSplitWaves)),
                     dim3(kSplitThreads), kRefLds, as_stream(stream), (const char*)packed_split, x, m, spatial_out, ld, n_raw,
                     reinterpret_cast<uint4*>(scratch));
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_refnerf_dir_fwd_split(const void* packed_split, const float* dir_in, int64_t ld, int64_t m,
                                          float* dir_out, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(packed_split && dir_in && dir_out, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && ld >= 273 && ld % 4 == 0 && ((uintptr_t)dir_in & 15) == 0, "dir_in rows must be 16-byte aligned");
  if (m == 0) return LNRF_OK;
  int rc = set_lds(refnerf_dir_fwd_split_kernel, kDirLds);
  if (rc) return rc;
  const int64_t tiles = (m + kTileCols - 1) / kTileCols;
  hipLaunchKernelGGL(refnerf_dir_fwd_split_kernel, dim3((unsigned)((tiles + kSplitWaves - 1) / kSplitWaves)),
                     dim3(kSplitThreads), kDirLds, as_stream(stream), (const char*)packed_split, dir_in, ld, m, dir_out);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int64_t lnrf_refnerf_trunk_packed_bytes(void) { return kRefPackBytes; }

extern "C" int lnrf_refnerf_trunk_pack(const float* params, void* packed, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(params && packed, "null pointer");
  hipLaunchKernelGGL(refnerf_pack_kernel, dim3(1024), dim3(256), 0, as_stream(stream), params, (char*)packed);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_refnerf_trunk_fwd(const void* packed, const float* x, int64_t m, void* save, float* spatial_out,
                                      int64_t ld, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(packed && x && save && spatial_out, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && ld >= 256 && ld % 4 == 0 && ((uintptr_t)spatial_out & 15) == 0,
                 "spatial_out rows must be 16-byte aligned (ld a multiple of 4, >= 256)");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = nerf_tiles_for(m);
  int rc = set_lds(refnerf_trunk_fwd_kernel, kRefLds);
  if (rc) return rc;
  hipLaunchKernelGGL(refnerf_trunk_fwd_kernel, tile_grid(n_tiles), dim3(kThreads), kRefLds, as_stream(stream),
                     (const char*)packed, x, m, n_tiles, (char*)save, spatial_out, ld);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_refnerf_normal_pass(const void* packed, const void* save, const float* x, int64_t m, void* cdump,
                                        float* nraw, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(packed && save && x && cdump && nraw, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = nerf_tiles_for(m);
  int rc = set_lds(refnerf_normal_kernel, kRefLds);
  if (rc) return rc;
  hipLaunchKernelGGL(refnerf_normal_kernel, tile_grid(n_tiles), dim3(kThreads), kRefLds, as_stream(stream),
                     (const char*)packed, (const char*)save, x, m, n_tiles, (char*)cdump, nraw);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

// the nine weight-gradient problems of the trunk: Dense_1..8 (hidden x hidden) and x_emb x [dy0 | dy5] (Dense_0 and the
// x_emb rows of Dense_5: the two dumps are neighbours, nerf_layout.h); 256 workgroups = one per CU (two rounds of 512 cost
// twice the partial-sum traffic for the same streaming rate)
// `slabs`: room for 512 workgroups' partial sums (kSlabBlockBytes each) — the deterministic epilogue of fused_chain.h
static int trunk_wgrad(const void* xbuf, const void* ybuf, int64_t n_tiles, int do_bias, float* grads, hipStream_t st,
                       float* slabs) {
  WgradArgs a;
  a.n_problems = 0;
  int first = 0;
  auto add = [&](int shape, int xs, int ys, int dense, int row_map, int row_off, int bias, int blocks, int col_map) {
    WgradProblem p;
    p.shape = shape; p.x_slot0 = xs; p.y_slot0 = ys; p.dense = dense; p.row_map = row_map; p.row_off = row_off;
    p.col_map = col_map; p.do_bias = bias;
    p.w_off = p.b_off = p.out_dim = p.n_rows = 0;
    int64_t nb = blocks;
    const int64_t cap = (n_tiles + 5) / 6;
    if (nb > cap) nb = cap;
    p.first_block = first;
    p.n_blocks = (int)nb;
    first += (int)nb;
    a.p[a.n_problems++] = p;
  };
  for (int l = 1; l <= 8; ++l) add(0, kSaveH + (l - 1) * 16, grad_dy_slot(l), l, ROW_HIDDEN, 0, do_bias, 28, COL_256);
  add(7, kSaveXin, grad_dy_slot(0), 0, ROW_XEMB, 0, do_bias, 32, COL_DY0_DY5);
  return launch_nerf_wgrad(a, first, xbuf, ybuf, n_tiles, grads, st, WgLayout{kSaveTileSlots, kGradTileSlots},
                           first <= 512 ? slabs : nullptr);
}
// the slab region behind a gradient dump that was sized by lnrf_nerf_bwd_scratch_bytes
static float* slabs_behind_dump(const void* dump, int64_t n_tiles) {
  return reinterpret_cast<float*>(const_cast<char*>(reinterpret_cast<const char*>(dump)) + (int64_t)kGradSlots * n_tiles * kFragBytes);
}

extern "C" int lnrf_refnerf_trunk_bwd(const void* packed, const void* save, const float* g_spatial, int64_t ld,
                                      int64_t m, void* scratch, float* grads, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(packed && save && g_spatial && scratch && grads, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && ld >= 256 && ld % 4 == 0 && ((uintptr_t)g_spatial & 15) == 0,
                 "gradient rows must be 16-byte aligned (ld a multiple of 4, >= 256)");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = nerf_tiles_for(m);
  // layer-stationary form (nerf_bwd_ls.hip): seed dy8, the pipeline forms dy7..dy0, dW_1..8 and db_1..8; the two x_emb
  // problems (Dense_0, rows 256.. of Dense_5) stay on the split-K kernel.  scratch: lnrf_refnerf_trunk_bwd_scratch_bytes.
  hipLaunchKernelGGL(refnerf_dy8_kernel, tile_grid(n_tiles), dim3(kThreads), 0, as_stream(stream), g_spatial, ld, m, n_tiles,
                     (char*)scratch);
  LNRF_LAUNCH_CHECK();
  int rc = launch_ls_pipeline(packed, save, scratch, m, grads, as_stream(stream));
  if (rc) return rc;
  WgradArgs a;
  a.n_problems = 0;
  int first = 0;
  {  // x_emb x [dy0 | dy5]: Dense_0 (with its bias) and rows 256.. of Dense_5
    WgradProblem p;
    p.shape = 7; p.x_slot0 = kSaveXin; p.y_slot0 = grad_dy_slot(0); p.dense = 0;
    p.row_map = ROW_XEMB; p.row_off = 0; p.col_map = COL_DY0_DY5; p.do_bias = 1;
    p.w_off = p.b_off = p.out_dim = p.n_rows = 0;
    int64_t nb = 256;
    const int64_t cap = (n_tiles + 5) / 6;
    if (nb > cap) nb = cap;
    p.first_block = first;
    p.n_blocks = (int)nb;
    first += (int)nb;
    a.p[a.n_problems++] = p;
  }
  float* small_slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + ls_small_slab_off(m));
  return launch_nerf_wgrad(a, first, save, scratch, n_tiles, grads, as_stream(stream),
                           WgLayout{kSaveTileSlots, kGradTileSlots}, small_slabs);
}

extern "C" int64_t lnrf_refnerf_trunk_bwd_scratch_bytes(int64_t m) { return m < 0 ? -1 : ls_scratch_bytes(m); }

extern "C" int lnrf_refnerf_normal_bwd(const void* packed, const void* save, const void* cdump, const float* x,
                                       const float* u, int64_t m, void* scratch, float* grads, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(packed && save && cdump && x && u && scratch && grads, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = nerf_tiles_for(m);
  int rc = set_lds(refnerf_tangent_kernel, kRefLds);
  if (rc) return rc;
  hipLaunchKernelGGL(refnerf_tangent_kernel, tile_grid(n_tiles), dim3(kThreads), kRefLds, as_stream(stream),
                     (const char*)packed, (const char*)save, x, u, m, n_tiles, (char*)scratch);
  LNRF_LAUNCH_CHECK();
  // the partial sums go behind the chain-state dump (cdump is sized by lnrf_nerf_bwd_scratch_bytes: dump + slab region)
  return trunk_wgrad(scratch, cdump, n_tiles, 0, grads, as_stream(stream), slabs_behind_dump(cdump, n_tiles));
}

extern "C" int64_t lnrf_refnerf_dir_save_bytes(int64_t m) { return (int64_t)kDirSaveSlots * nerf_tiles_for(m) * kFragBytes; }
extern "C" int64_t lnrf_refnerf_dir_scratch_bytes(int64_t m) {  // gradient dump, then the slab region of the weight-gradient launch
  return (int64_t)kDirGradSlots * nerf_tiles_for(m) * kFragBytes + 512 * kSlabBlockBytes;
}

extern "C" int lnrf_refnerf_dir_fwd(const void* packed, const float* dir_in, int64_t ld, int64_t m, void* dsave,
                                    float* dir_out, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(packed && dir_in && dsave && dir_out, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && ld >= 276 && ld % 4 == 0 && ((uintptr_t)dir_in & 15) == 0,
                 "dir_in rows must be 16-byte aligned (ld a multiple of 4, >= 276)");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = nerf_tiles_for(m);
  int rc = set_lds(refnerf_dir_fwd_kernel, kDirLds);
  if (rc) return rc;
  hipLaunchKernelGGL(refnerf_dir_fwd_kernel, tile_grid(n_tiles), dim3(kThreads), kDirLds, as_stream(stream),
                     (const char*)packed, dir_in, ld, m, n_tiles, (char*)dsave, dir_out);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_refnerf_dir_bwd(const void* packed, const void* dsave, const float* g_dir_out, int64_t m,
                                    void* scratch, float* g_dir_in, int64_t ld, float* grads, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(packed && dsave && g_dir_out && scratch && g_dir_in && grads, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && ld >= 276 && ld % 4 == 0 && ((uintptr_t)g_dir_in & 15) == 0,
                 "g_dir_in rows must be 16-byte aligned (ld a multiple of 4, >= 276)");
  if (m == 0) return LNRF_OK;
  const int64_t n_tiles = nerf_tiles_for(m);
  int rc = set_lds(refnerf_dir_bwd_kernel, kDirLds);
  if (rc) return rc;
  hipLaunchKernelGGL(refnerf_dir_bwd_kernel, tile_grid(n_tiles), dim3(kThreads), kDirLds, as_stream(stream),
                     (const char*)packed, (const char*)dsave, g_dir_out, m, n_tiles, (char*)scratch, g_dir_in, ld);
  LNRF_LAUNCH_CHECK();
  // weight gradients: Dense_9 = [input fragments]^T dy9 (273 x 128 + bias), Dense_10 = relu(Dense_9)^T dy10 (128 x 3 + bias)
  WgradArgs a;
  a.n_problems = 0;
  int first = 0;
  auto add = [&](int shape, int xs, int ys, int out_dim, int n_rows, int w_off, int b_off, int blocks) {
    WgradProblem p;
    p.shape = shape; p.x_slot0 = xs; p.y_slot0 = ys; p.dense = 0; p.row_map = ROW_HIDDEN; p.row_off = 0;
    p.col_map = COL_EXPLICIT; p.do_bias = 1;
    p.w_off = w_off; p.b_off = b_off; p.out_dim = out_dim; p.n_rows = n_rows;
    int64_t nb = blocks;
    const int64_t cap = (n_tiles + 5) / 6;
    if (nb > cap) nb = cap;
    p.first_block = first;
    p.n_blocks = (int)nb;
    first += (int)nb;
    a.p[a.n_problems++] = p;
  };
  add(5, kDirSaveXin, kDirGradDy9, kDirHidden, kDirIn, kDirW9, kDirB9, 200);  // 256 workgroups = one per CU
  add(4, kDirSaveH, kDirGradDy10, 3, kDirHidden, kDirW10, kDirB10, 56);
  float* slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + (int64_t)kDirGradSlots * n_tiles * kFragBytes);
  return launch_nerf_wgrad(a, first, dsave, scratch, n_tiles, grads, as_stream(stream),
                           WgLayout{kDirSaveTileSlots, kDirGradTileSlots}, first <= 512 ? slabs : nullptr);
}
