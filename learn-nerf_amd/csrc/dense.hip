// dense.hip — generic fp32 dense layers on the f32-input MFMA (v_mfma_f32_32x32x2_f32: exact
// fp32 products, fp32 accumulate).  This is the model-agnostic / exact-precision path:
// flax.linen.Dense as used at learn_nerf/model.py:51-60, instant_ngp.py:47-53, ref_nerf.py:97-107,
// plus sinusoidal_emb (model.py:65-77).  The performance path for NeRFModel is nerf_mlp.hip.
#include <cstdint>
#include <type_traits>
#include <utility>

#include "common.h"

namespace lnrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// operand precision of the dense path (lnrf_set_dense_precision): thread-local like lnrf_last_error
static thread_local int g_dense_bf16 = 0;

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case LNRF_ACT_RELU: return fmaxf(x, 0.0f);
    case LNRF_ACT_SOFTPLUS: return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));  // logaddexp(x, 0)
    case LNRF_ACT_TANH: return tanhf(x);
    case LNRF_ACT_EXP: return expf(x);
    case LNRF_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
    default: return x;
  }
}

// derivative of the activation expressed through its OUTPUT y
__device__ __forceinline__ float act_grad_from_output(float y, int act) {
  switch (act) {
    case LNRF_ACT_RELU: return y > 0.0f ? 1.0f : 0.0f;
    case LNRF_ACT_SOFTPLUS: return -expm1f(-y);  // sigmoid(x) = 1 - exp(-softplus(x)), no cancellation
    case LNRF_ACT_TANH: return 1.0f - y * y;
    case LNRF_ACT_EXP: return y;
    case LNRF_ACT_SIGMOID: return y * (1.0f - y);
    default: return 1.0f;
  }
}

constexpr int TI = 64, TJ = 64, RC = 16;

// Optional epilogue gate: C(i, j) *= act'(gate[i][j]) for j < n (activation derivative through its OUTPUT), i.e. the
// activation backward of the layer below fused into the GEMM that produces its input gradient (one pass over a
// 0.8 GB tensor less per layer).  Applied in modes 0 and 1 after bias / activation.
struct Gate {
  const float* y;
  int64_t ld;
  int act;
  int n;
};

// C[i][j] (op)= sum_r A(i,r) * B(r,j),  A(i,r) = a[i*sa_i + r*sa_r],  B(r,j) = b[r*sb_r + j*sb_j].
// One workgroup = 64x64 output tile, 4 waves in a 2x2 grid of 32x32 MFMA tiles.
// mode 0: store act(C + bias);  1: C += result (plain);  2: atomicAdd (split reduction);
// mode 3: split reduction without atomics — split z stores its partial tile at c[(z * I + row) * ldc + col] and
// dense_fold_kernel adds the splits in a fixed order (lnrf_dense_bwd_weight_det).
// BF16 = false: exact fp32 products (v_mfma_f32_32x32x2_f32).  BF16 = true: operands rounded to bf16 while they
// are staged into LDS, one v_mfma_f32_32x32x16_bf16 per 16-deep chunk, fp32 accumulate / bias / activation:
// the arithmetic of the fused kernels for models that have no fused kernel (16x the MFMA rate; the kernel is
// then bound by operand staging and by the fp32 activations in HBM).
template <bool BF16>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ a, int64_t sa_i,
                                                       int64_t sa_r, const float* __restrict__ b,
                                                       int64_t sb_r, int64_t sb_j,
                                                       float* __restrict__ c, int64_t ldc,
                                                       const float* __restrict__ bias, int act,
                                                       int mode, int64_t I, int J, int64_t R,
                                                       int64_t r_per_split, Gate gate) {
  // fp32: As[i][r], Bs[r][j].  bf16: both operands r-contiguous (8 k-values per lane = one 16-byte read)
  __shared__ float As[BF16 ? 1 : TI][RC + 1];
  __shared__ float Bs[BF16 ? 1 : RC][TJ + 4];
  __shared__ __attribute__((aligned(16))) __bf16 Ah[BF16 ? TI : 1][RC + 8];
  __shared__ __attribute__((aligned(16))) __bf16 Bh[BF16 ? TJ : 1][RC + 8];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.x * TI;
  const int j0 = blockIdx.y * TJ;
  const int64_t r_begin = (int64_t)blockIdx.z * r_per_split;
  const int64_t r_end = r_begin + r_per_split < R ? r_begin + r_per_split : R;

  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.0f;

  const bool a_r_fast = (sa_r == 1);
  const bool b_j_fast = (sb_j == 1);

  // software pipeline: the global loads of reduction chunk r0 + RC are in registers while chunk r0 is
  // multiplied out of LDS (the loop was latency-bound on one global round trip per 16-deep chunk before)
  constexpr int NA = (TI * RC) / 256, NB = (RC * TJ) / 256;
  float ra[NA], rb[NB];
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int e = q * 256 + tid;
      int ii, rr;
      if (a_r_fast) { rr = e % RC; ii = e / RC; } else { ii = e % TI; rr = e / TI; }
      const int64_t gi = i0 + ii, gr = r0 + rr;
      ra[q] = (gi < I && gr < r_end) ? a[gi * sa_i + gr * sa_r] : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int e = q * 256 + tid;
      int jj, rr;
      if (b_j_fast) { jj = e % TJ; rr = e / TJ; } else { rr = e % RC; jj = e / RC; }
      const int gj = j0 + jj;
      const int64_t gr = r0 + rr;
      rb[q] = (gj < J && gr < r_end) ? b[gr * sb_r + (int64_t)gj * sb_j] : 0.0f;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int e = q * 256 + tid;
      int ii, rr;
      if (a_r_fast) { rr = e % RC; ii = e / RC; } else { ii = e % TI; rr = e / TI; }
      if constexpr (BF16) Ah[ii][rr] = (__bf16)ra[q];
      else As[ii][rr] = ra[q];
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      const int e = q * 256 + tid;
      int jj, rr;
      if (b_j_fast) { jj = e % TJ; rr = e / TJ; } else { rr = e % RC; jj = e / RC; }
      if constexpr (BF16) Bh[jj][rr] = (__bf16)rb[q];
      else Bs[rr][jj] = rb[q];
    }
  };
  if (r_begin < r_end) fetch(r_begin);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += RC) {
    stage();
    __syncthreads();
    if (r0 + RC < r_end) fetch(r0 + RC);
    if constexpr (BF16) {
      static_assert(RC == 16, "one 32x32x16 MFMA per chunk");
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(&Ah[wr * 32 + (lane & 31)][8 * (lane >> 5)]);
      const bf16x8 bv = *reinterpret_cast<const bf16x8*>(&Bh[wc * 32 + (lane & 31)][8 * (lane >> 5)]);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    } else {
#pragma unroll
      for (int kk = 0; kk < RC / 2; ++kk) {
        const float av = As[wr * 32 + (lane & 31)][kk * 2 + (lane >> 5)];
        const float bv = Bs[kk * 2 + (lane >> 5)][wc * 32 + (lane & 31)];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const int col = j0 + wc * 32 + (lane & 31);
  if (col < J) {
    const float bv = (bias && mode == 0) ? bias[col] : 0.0f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int64_t row = i0 + wr * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
      if (row < I) {
        float* dst = c + (mode == 3 ? (int64_t)blockIdx.z * I + row : row) * ldc + col;
        const float gm = (gate.y && col < gate.n) ? act_grad_from_output(gate.y[row * gate.ld + col], gate.act) : 1.0f;
        if (mode == 0) *dst = act_apply(acc[q] + bv, act) * gm;
        else if (mode == 1) *dst += acc[q] * gm;
        else if (mode == 3) *dst = acc[q];
        else atomicAdd(dst, acc[q]);
      }
    }
  }
}

// Large-problem variant: 128x128 output tile, 4 waves x (2x2) 32x32 MFMA tiles, operands fetched as float4 along
// their contiguous dimension (the caller guarantees 16-byte alignment and sizes that are multiples of 4 along
// it).  A_FAST_R: A(i, r) contiguous in r (activations, row-major) else in i (X^T of the weight gradient);
// B_FAST_R: B(r, j) contiguous in r (W^T of the input gradient) else in j (W, dy).  The generic kernel above
// spends ~15 instructions per operand element on address arithmetic and scalar staging, which caps it at
// ~60 (f32) / ~100 (bf16) TFLOP/s; this one moves 4 elements per load.
constexpr int BI = 128, BJ = 128;
// bf16 LDS image of an operand tile: [row][k] with the 16-byte chunks (8 k) of a row XOR-swizzled by bits 2..3 of the
// row.  The operand that is contiguous along its row dimension in memory arrives as float4 = 4 ROWS at one k, i.e.
// as scalar 2-byte LDS stores 4 rows apart: without the swizzle they fall on 4 banks (8-way conflict).
__device__ __forceinline__ int swz(int row, int k) { return ((((k >> 3) ^ (row >> 2)) & 3) << 3) | (k & 7) | (k & ~31); }
// B_ALIGNED = false: B (a weight matrix inside a flat parameter vector) is only 4-byte aligned.
template <bool BF16, bool A_FAST_R, bool B_FAST_R, bool B_ALIGNED = true>
__global__ __launch_bounds__(256) void gemm_big_kernel(const float* __restrict__ a, int64_t lda,
                                                       const float* __restrict__ b, int64_t ldb,
                                                       float* __restrict__ c, int64_t ldc,
                                                       const float* __restrict__ bias, int act, int mode, int64_t I,
                                                       int J, int64_t R, int64_t r_per_split, Gate gate) {
  constexpr int KC = BF16 ? 32 : 16;                 // reduction depth per chunk
  constexpr int NV = (BI * KC / 4) / 256;            // float4 per thread and operand
  __shared__ __attribute__((aligned(16))) __bf16 Ah[BF16 ? BI : 1][KC + 8];
  __shared__ __attribute__((aligned(16))) __bf16 Bh[BF16 ? BJ : 1][KC + 8];
  __shared__ float As[BF16 ? 1 : BI][KC + 1];
  __shared__ float Bs[BF16 ? 1 : KC][BJ + 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blocks b and b + 8 share an L2), so
  // the column tiles of one row tile are placed 8 blocks apart: the second one finds the A tile in its XCD's L2
  // instead of re-reading it from HBM.  blockIdx.x = (8 nj) g + 8 jt + t  ->  row tile 8 g + t, column tile jt.
  // (Few row tiles, e.g. the split-K weight gradient: plain order, or all the work would land on ni of the 8 XCDs.)
  const int nj = (J + BJ - 1) / BJ;
  const int64_t ni = (I + BI - 1) / BI;
  int64_t it;
  int jt;
  if (ni >= 64) {
    const unsigned grp = blockIdx.x / (8u * nj), rem = blockIdx.x % (8u * nj);
    it = (int64_t)grp * 8 + (rem & 7u);
    jt = (int)(rem >> 3);
  } else {
    it = blockIdx.x % (unsigned)ni;
    jt = (int)(blockIdx.x / (unsigned)ni);
  }
  const int64_t i0 = it * BI;
  if (i0 >= I || jt >= nj) return;
  const int j0 = jt * BJ;
  const int64_t r_begin = (int64_t)blockIdx.z * r_per_split;
  const int64_t r_end = r_begin + r_per_split < R ? r_begin + r_per_split : R;

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[x][y][q] = 0.0f;

  float4 ra[NV], rb[NV];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // element (row, k) of an operand tile: row = i or j (0..127), k = 0..KC-1
  // branch-free: out-of-range vectors read element 0 and are zeroed afterwards (a guarded load makes the
  // compiler wait for every load separately)
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int v = q * 256 + tid;
      int64_t ao, bo;
      bool aok, bok;
      if (A_FAST_R) {
        const int row = v / (KC / 4), k4 = (v % (KC / 4)) * 4;
        const int64_t gi = i0 + row, gr = r0 + k4;
        aok = gi < I && gr < r_end;
        ao = gi * lda + gr;
      } else {
        const int k = v / (BI / 4), row4 = (v % (BI / 4)) * 4;
        const int64_t gi = i0 + row4, gr = r0 + k;
        aok = gi < I && gr < r_end;
        ao = gr * lda + gi;
      }
      if (B_FAST_R) {
        const int row = v / (KC / 4), k4 = (v % (KC / 4)) * 4;
        const int gj = j0 + row;
        const int64_t gr = r0 + k4;
        bok = gj < J && gr < r_end;
        bo = (int64_t)gj * ldb + gr;
      } else {
        const int k = v / (BJ / 4), row4 = (v % (BJ / 4)) * 4;
        const int gj = j0 + row4;
        const int64_t gr = r0 + k;
        bok = gj < J && gr < r_end;
        bo = gr * ldb + gj;
      }
      const float4 av = *reinterpret_cast<const float4*>(a + (aok ? ao : 0));
      float4 bv;
      if constexpr (B_ALIGNED) bv = *reinterpret_cast<const float4*>(b + (bok ? bo : 0));
      else __builtin_memcpy(&bv, b + (bok ? bo : 0), sizeof(float4));  // 4-byte aligned source
      ra[q] = aok ? av : zero4;
      rb[q] = bok ? bv : zero4;
    }
  };
  auto put = [&](auto is_a, int row, int k, float val) {
    if constexpr (decltype(is_a)::value) {
      if constexpr (BF16) Ah[row][swz(row, k)] = (__bf16)val; else As[row][k] = val;
    } else {
      if constexpr (BF16) Bh[row][swz(row, k)] = (__bf16)val; else Bs[k][row] = val;
    }
  };
  auto stage = [&]() {
    std::true_type is_a;
    std::false_type is_b;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int v = q * 256 + tid;
      const float va[4] = {ra[q].x, ra[q].y, ra[q].z, ra[q].w};
      const float vb[4] = {rb[q].x, rb[q].y, rb[q].z, rb[q].w};
      if (A_FAST_R) {
        const int row = v / (KC / 4), k4 = (v % (KC / 4)) * 4;
        if constexpr (BF16) {  // four consecutive k of one row: one 8-byte LDS store
          const bf16x4 hv = {(__bf16)va[0], (__bf16)va[1], (__bf16)va[2], (__bf16)va[3]};
          *reinterpret_cast<bf16x4*>(&Ah[row][swz(row, k4)]) = hv;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) put(is_a, row, k4 + e, va[e]);
        }
      } else {
        const int k = v / (BI / 4), row4 = (v % (BI / 4)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) put(is_a, row4 + e, k, va[e]);
      }
      if (B_FAST_R) {
        const int row = v / (KC / 4), k4 = (v % (KC / 4)) * 4;
        if constexpr (BF16) {
          const bf16x4 hv = {(__bf16)vb[0], (__bf16)vb[1], (__bf16)vb[2], (__bf16)vb[3]};
          *reinterpret_cast<bf16x4*>(&Bh[row][swz(row, k4)]) = hv;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) put(is_b, row, k4 + e, vb[e]);
        }
      } else {
        const int k = v / (BJ / 4), row4 = (v % (BJ / 4)) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) put(is_b, row4 + e, k, vb[e]);
      }
    }
  };
  if (r_begin < r_end) fetch(r_begin);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += KC) {
    stage();
    __syncthreads();
    if (r0 + KC < r_end) fetch(r0 + KC);
    if constexpr (BF16) {
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        bf16x8 av[2], bv[2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
          av[x] = *reinterpret_cast<const bf16x8*>(&Ah[wr * 64 + x * 32 + (lane & 31)][swz(wr * 64 + x * 32 + (lane & 31), 16 * ks + 8 * (lane >> 5))]);
#pragma unroll
        for (int y = 0; y < 2; ++y)
          bv[y] = *reinterpret_cast<const bf16x8*>(&Bh[wc * 64 + y * 32 + (lane & 31)][swz(wc * 64 + y * 32 + (lane & 31), 16 * ks + 8 * (lane >> 5))]);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[x], bv[y], acc[x][y], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < KC / 2; ++kk) {
        float av[2], bv[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) av[x] = As[wr * 64 + x * 32 + (lane & 31)][kk * 2 + (lane >> 5)];
#pragma unroll
        for (int y = 0; y < 2; ++y) bv[y] = Bs[kk * 2 + (lane >> 5)][wc * 64 + y * 32 + (lane & 31)];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x], bv[y], acc[x][y], 0, 0, 0);
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int y = 0; y < 2; ++y) {
    const int col = j0 + wc * 64 + y * 32 + (lane & 31);
    if (col >= J) continue;
    const float bvv = (bias && mode == 0) ? bias[col] : 0.0f;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int64_t row = i0 + wr * 64 + x * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < I) {
          float* dst = c + (mode == 3 ? (int64_t)blockIdx.z * I + row : row) * ldc + col;
          const float gm = (gate.y && col < gate.n) ? act_grad_from_output(gate.y[row * gate.ld + col], gate.act) : 1.0f;
          if (mode == 0) *dst = act_apply(acc[x][y][q] + bvv, act) * gm;
          else if (mode == 1) *dst += acc[x][y][q] * gm;
          else if (mode == 3) *dst = acc[x][y][q];
          else atomicAdd(dst, acc[x][y][q]);
        }
      }
  }
}

__global__ void col_sum_kernel(const float* __restrict__ g, int64_t ldg, int64_t m, int n,
                               float* __restrict__ out) {
  // out[j] += sum_i g[i][j]; grid.x tiles rows (1024 per block), threads stride columns
  const int64_t rows_per_block = 1024;
  const int64_t i0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t i1 = i0 + rows_per_block < m ? i0 + rows_per_block : m;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    float s = 0.0f;
    for (int64_t i = i0; i < i1; ++i) s += g[i * ldg + j];
    atomicAdd(out + j, s);
  }
}

// n % 4 == 0, n <= 256, 16-byte aligned rows: 64 float4 column groups x 4 row lanes per workgroup, 8 independent
// row loads in flight per thread (the scalar kernel above runs one dependent load-add chain per thread)
constexpr int kColSumRows = 512;
__global__ __launch_bounds__(256) void col_sum4_kernel(const float* __restrict__ g, int64_t ldg, int64_t m, int n,
                                                       float* __restrict__ out) {
  __shared__ float4 part[4][64];
  const int cg = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * kColSumRows;
  const int64_t i1 = i0 + kColSumRows < m ? i0 + kColSumRows : m;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (4 * cg < n) {
    for (int64_t i = i0 + rl; i < i1; i += 32) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int64_t r = i + 4 * q;
        const float4 v = r < i1 ? *reinterpret_cast<const float4*>(g + r * ldg + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
  }
  part[rl][cg] = s;
  __syncthreads();
  if (rl == 0 && 4 * cg < n) {
    const float4 a = part[0][cg], b = part[1][cg], c = part[2][cg], d = part[3][cg];
    atomicAdd(out + 4 * cg + 0, a.x + b.x + c.x + d.x);
    atomicAdd(out + 4 * cg + 1, a.y + b.y + c.y + d.y);
    atomicAdd(out + 4 * cg + 2, a.z + b.z + c.z + d.z);
    atomicAdd(out + 4 * cg + 3, a.w + b.w + c.w + d.w);
  }
}

// out[e] += parts[0][e] + parts[1][e] + ... in that order (e < count): the deterministic end of a split reduction
__global__ __launch_bounds__(256) void dense_fold_kernel(const float* __restrict__ parts, int n_parts, int64_t count,
                                                         float* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  float s = 0.0f;
  for (int p = 0; p < n_parts; ++p) s += parts[(int64_t)p * count + e];
  out[e] += s;
}
// parts[block][j] = sum of the block's 512 rows of g[:, j] (any n, any alignment); folded by dense_fold_kernel
__global__ __launch_bounds__(256) void col_sum_parts_kernel(const float* __restrict__ g, int64_t ldg, int64_t m, int n,
                                                            float* __restrict__ parts) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * kColSumRows;
  const int64_t i1 = i0 + kColSumRows < m ? i0 + kColSumRows : m;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + cl;
    float s = 0.0f;
    if (j < n)
      for (int64_t i = i0 + rl; i < i1; i += 4) s += g[i * ldg + j];
    part[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && j < n) parts[(int64_t)blockIdx.x * n + j] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
    __syncthreads();
  }
}

__global__ void act_bwd_kernel(float* __restrict__ g, int64_t ldg, const float* __restrict__ y,
                               int64_t ldy, int act, int64_t m, int n) {
  const int64_t total = m * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = e / n;
    const int j = (int)(e - i * n);
    g[i * ldg + j] *= act_grad_from_output(y[i * ldy + j], act);
  }
}

__global__ void sinusoidal_emb_kernel(const float* __restrict__ x, int64_t ldx, int64_t m, int dims,
                                      int freqs, float* __restrict__ out, int64_t ldo,
                                      int64_t col_off) {
  const int64_t total = m * dims * freqs;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(e % freqs);
    const int64_t t = e / freqs;
    const int cdim = (int)(t % dims);
    const int64_t row = t / dims;
    const float arg = x[row * ldx + cdim] * (float)(1u << f);  // model.py:72-73 (2^f exact)
    float s, c;
    sincosf(arg, &s, &c);
    float* o = out + row * ldo + col_off + (int64_t)cdim * 2 * freqs;
    o[f] = s;           // model.py:74-77: per coordinate [sin block, cos block]
    o[freqs + f] = c;
  }
}

}  // namespace lnrf

using namespace lnrf;

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int launch_gemm(const float* a, int64_t sa_i, int64_t sa_r, const float* b, int64_t sb_r,
                       int64_t sb_j, float* c, int64_t ldc, const float* bias, int act, int mode,
                       int64_t I, int J, int64_t R, int splits, hipStream_t stream,
                       Gate gate = Gate{nullptr, 0, 0, 0}, int* splits_used = nullptr) {
  if (splits_used) *splits_used = 0;
  if (I == 0 || J == 0) return LNRF_OK;
  if (splits < 1) splits = 1;
  // vectorised 128x128 kernel: both operands contiguous along one of their dimensions, 16-byte aligned rows,
  // extents along the contiguous dimensions multiples of 4, and enough work to fill 128-wide tiles
  const bool a_fast_r = sa_r == 1, a_fast_i = sa_i == 1 && !a_fast_r;
  const bool b_fast_r = sb_r == 1, b_fast_j = sb_j == 1;
  const int64_t lda = a_fast_r ? sa_i : sa_r, ldb = b_fast_r ? sb_j : sb_r;
  // A contiguous in i = X^T of the weight gradient: transposing LDS stores (swizzled bf16 image / 17-float rows in f32)
  const bool b_al = aligned16(b);  // weights inside a flat parameter vector may start at any float
  const bool big = (a_fast_r || a_fast_i) && (b_fast_r || b_fast_j) && aligned16(a) &&
                   lda % 4 == 0 && ldb % 4 == 0 && (a_fast_r ? R % 4 == 0 : I % 4 == 0) &&
                   (b_fast_r ? R % 4 == 0 : J % 4 == 0) && I >= 64 && J >= 64 && R >= 32;
  if (big) {
    const int kc = g_dense_bf16 ? 32 : 16;
    int64_t per = (R + splits - 1) / splits;
    per = ((per + kc - 1) / kc) * kc;
    if (per % 4 != 0) per = ((per + 3) / 4) * 4;
    int nsplit = (int)((R + per - 1) / per);
    if (nsplit < 1) nsplit = 1;
    if (splits_used) *splits_used = nsplit;
    const int64_t ni = (I + BI - 1) / BI, njt = (J + BJ - 1) / BJ;
    dim3 grid((unsigned)((ni >= 64 ? ((ni + 7) / 8) * 8 : ni) * njt), 1u, (unsigned)nsplit);  // tile order: see the kernel
#define LNRF_BIG(BF, AR, BR)                                                                                          \
  do {                                                                                                                \
    if (b_al)                                                                                                         \
      hipLaunchKernelGGL((gemm_big_kernel<BF, AR, BR, true>), grid, dim3(256), 0, stream, a, lda, b, ldb, c, ldc, bias, \
                         act, mode, I, J, R, per, gate);                                                              \
    else                                                                                                              \
      hipLaunchKernelGGL((gemm_big_kernel<BF, AR, BR, false>), grid, dim3(256), 0, stream, a, lda, b, ldb, c, ldc,    \
                         bias, act, mode, I, J, R, per, gate);                                                        \
  } while (0)
    const bool ar = a_fast_r, br = !b_fast_j;
    if (g_dense_bf16) {
      if (ar && br) LNRF_BIG(true, true, true); else if (ar) LNRF_BIG(true, true, false);
      else if (br) LNRF_BIG(true, false, true); else LNRF_BIG(true, false, false);
    } else {
      if (ar && br) LNRF_BIG(false, true, true); else if (ar) LNRF_BIG(false, true, false);
      else if (br) LNRF_BIG(false, false, true); else LNRF_BIG(false, false, false);
    }
#undef LNRF_BIG
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gemm_big");
    return LNRF_OK;
  }
  int64_t per = (R + splits - 1) / splits;
  per = ((per + RC - 1) / RC) * RC;
  if (per < RC) per = RC;
  splits = (int)((R + per - 1) / per);
  if (splits < 1) splits = 1;
  if (splits_used) *splits_used = splits;
  dim3 grid((unsigned)((I + TI - 1) / TI), (unsigned)((J + TJ - 1) / TJ), (unsigned)splits);
  if (g_dense_bf16)
    hipLaunchKernelGGL(gemm_f32_kernel<true>, grid, dim3(256), 0, stream, a, sa_i, sa_r, b, sb_r, sb_j, c, ldc,
                       bias, act, mode, I, J, R, per, gate);
  else
    hipLaunchKernelGGL(gemm_f32_kernel<false>, grid, dim3(256), 0, stream, a, sa_i, sa_r, b, sb_r, sb_j, c, ldc,
                       bias, act, mode, I, J, R, per, gate);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "gemm_f32");
  return LNRF_OK;
}

extern "C" int lnrf_set_dense_precision(int32_t precision) {
  LNRF_CHECK_ARG(precision == LNRF_DENSE_FP32 || precision == LNRF_DENSE_BF16, "precision must be LNRF_DENSE_FP32 or LNRF_DENSE_BF16");
  g_dense_bf16 = precision == LNRF_DENSE_BF16;
  return LNRF_OK;
}
extern "C" int32_t lnrf_get_dense_precision(void) { return g_dense_bf16 ? LNRF_DENSE_BF16 : LNRF_DENSE_FP32; }

extern "C" int lnrf_dense_fwd(const float* x, int64_t ldx, const float* w, const float* b,
                              int32_t act, float* y, int64_t ldy, int64_t m, int32_t k, int32_t n,
                              lnrf_stream_t stream) {
  LNRF_CHECK_ARG(x && w && y, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && k >= 1 && n >= 1 && ldx >= k && ldy >= n, "bad sizes");
  LNRF_CHECK_ARG(act >= 0 && act <= LNRF_ACT_SIGMOID, "bad activation");
  return launch_gemm(x, ldx, 1, w, n, 1, y, ldy, b, act, 0, m, n, k, 1, as_stream(stream));
}

extern "C" int lnrf_act_bwd(float* g, int64_t ldg, const float* y, int64_t ldy, int32_t act,
                            int64_t m, int32_t n, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(g && y, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && n >= 1 && ldg >= n && ldy >= n, "bad sizes");
  if (m == 0 || act == LNRF_ACT_NONE) return LNRF_OK;
  int64_t blocks = (m * n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g, ldg, y,
                     ldy, act, m, n);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_dense_bwd_input(const float* gy, int64_t ldgy, const float* w, float* gx,
                                    int64_t ldgx, int32_t accumulate, int64_t m, int32_t k,
                                    int32_t n, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(gy && w && gx, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && k >= 1 && n >= 1 && ldgy >= n && ldgx >= k, "bad sizes");
  // gx[i][kk] = sum_r gy[i][r] * w[kk][r]  ->  B(r, j=kk) = w[kk*n + r]
  return launch_gemm(gy, ldgy, 1, w, 1, n, gx, ldgx, nullptr, 0, accumulate ? 1 : 0, m, k, n, 1,
                     as_stream(stream));
}

extern "C" int lnrf_dense_bwd_input_gated(const float* gy, int64_t ldgy, const float* w, const float* y_below,
                                          int64_t ldy, int32_t act_below, int32_t n_gated, float* gx, int64_t ldgx,
                                          int32_t accumulate, int64_t m, int32_t k, int32_t n, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(gy && w && gx && y_below, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && k >= 1 && n >= 1 && ldgy >= n && ldgx >= k, "bad sizes");
  LNRF_CHECK_ARG(n_gated >= 0 && n_gated <= k && ldy >= n_gated, "bad gate extent");
  LNRF_CHECK_ARG(act_below >= 0 && act_below <= LNRF_ACT_SIGMOID, "bad activation");
  return launch_gemm(gy, ldgy, 1, w, 1, n, gx, ldgx, nullptr, 0, accumulate ? 1 : 0, m, k, n, 1, as_stream(stream),
                     Gate{y_below, ldy, act_below, n_gated});
}

extern "C" int lnrf_dense_fwd_gated(const float* x, int64_t ldx, const float* w, const float* b, int32_t act,
                                    const float* y_gate, int64_t ldg, int32_t act_gate, float* y, int64_t ldy,
                                    int64_t m, int32_t k, int32_t n, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(x && w && y && y_gate, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && k >= 1 && n >= 1 && ldx >= k && ldy >= n && ldg >= n, "bad sizes");
  LNRF_CHECK_ARG(act >= 0 && act <= LNRF_ACT_SIGMOID && act_gate >= 0 && act_gate <= LNRF_ACT_SIGMOID, "bad activation");
  return launch_gemm(x, ldx, 1, w, n, 1, y, ldy, b, act, 0, m, n, k, 1, as_stream(stream), Gate{y_gate, ldg, act_gate, n});
}

extern "C" int lnrf_dense_bwd_weight(const float* x, int64_t ldx, const float* gy, int64_t ldgy,
                                     float* gw, float* gb, int64_t m, int32_t k, int32_t n,
                                     lnrf_stream_t stream) {
  LNRF_CHECK_ARG(gy && ((x && gw) || (!x && !gw && gb)), "null pointer");
  LNRF_CHECK_ARG(m >= 0 && n >= 1 && ldgy >= n && (!x || (k >= 1 && ldx >= k)), "bad sizes");
  if (m == 0) return LNRF_OK;
  if (x) {
    // gw[kk][j] += sum_m x[m][kk] * gy[m][j]: A(i=kk, r=m) = x[m*ldx + kk]
    const int tiles = ((k + TI - 1) / TI) * ((n + TJ - 1) / TJ);
    int splits = (int)((2048 + tiles - 1) / tiles);
    if (splits > 512) splits = 512;  // bounds the atomic adders per output element
    const int64_t max_splits = (m + 255) / 256;
    if (splits > max_splits) splits = (int)max_splits;
    int rc = launch_gemm(x, 1, ldx, gy, ldgy, 1, gw, n, nullptr, 0, 2, k, n, m, splits, as_stream(stream));
    if (rc != LNRF_OK) return rc;
  }
  if (gb) {
    if (n % 4 == 0 && n <= 256 && ldgy % 4 == 0 && aligned16(gy))
      hipLaunchKernelGGL(col_sum4_kernel, dim3((unsigned)((m + kColSumRows - 1) / kColSumRows)), dim3(256), 0,
                         as_stream(stream), gy, ldgy, m, n, gb);
    else
      hipLaunchKernelGGL(col_sum_kernel, dim3((unsigned)((m + 1023) / 1024)), dim3(256), 0,
                         as_stream(stream), gy, ldgy, m, n, gb);
    LNRF_LAUNCH_CHECK();
  }
  return LNRF_OK;
}

static int wgrad_splits(int64_t m, int k, int n) {
  const int tiles = ((k + TI - 1) / TI) * ((n + TJ - 1) / TJ);
  int splits = (int)((2048 + tiles - 1) / tiles);
  if (splits > 512) splits = 512;
  const int64_t max_splits = (m + 255) / 256;
  if (splits > max_splits) splits = (int)max_splits;
  return splits < 1 ? 1 : splits;
}
extern "C" int64_t lnrf_dense_bwd_weight_scratch_bytes(int64_t m, int32_t k, int32_t n) {
  if (m < 0 || k < 0 || n < 1) return -1;
  const int64_t kernel_parts = k > 0 ? (int64_t)wgrad_splits(m, k, n) * k * n : 0;
  const int64_t bias_parts = (m + kColSumRows - 1) / kColSumRows * n;
  return (kernel_parts + bias_parts) * (int64_t)sizeof(float) + 256;
}

// lnrf_dense_bwd_weight with a fixed summation order: every split of the reduction over m stores its partial sums in
// `scratch` and a second launch adds them split by split, so two calls on the same inputs agree bit for bit.
extern "C" int lnrf_dense_bwd_weight_det(const float* x, int64_t ldx, const float* gy, int64_t ldgy, float* gw, float* gb,
                                         int64_t m, int32_t k, int32_t n, void* scratch, int64_t scratch_bytes,
                                         lnrf_stream_t stream) {
  LNRF_CHECK_ARG(gy && ((x && gw) || (!x && !gw && gb)), "null pointer");
  LNRF_CHECK_ARG(m >= 0 && n >= 1 && ldgy >= n && (!x || (k >= 1 && ldx >= k)), "bad sizes");
  if (m == 0) return LNRF_OK;
  LNRF_CHECK_ARG(scratch && scratch_bytes >= lnrf_dense_bwd_weight_scratch_bytes(m, x ? k : 0, n), "scratch too small");
  float* parts = reinterpret_cast<float*>(scratch);
  hipStream_t st = as_stream(stream);
  if (x) {
    int used = 0;
    int rc = launch_gemm(x, 1, ldx, gy, ldgy, 1, parts, n, nullptr, 0, 3, k, n, m, wgrad_splits(m, k, n), st,
                         Gate{nullptr, 0, 0, 0}, &used);
    if (rc != LNRF_OK) return rc;
    const int64_t count = (int64_t)k * n;
    hipLaunchKernelGGL(dense_fold_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, parts, used, count, gw);
    LNRF_LAUNCH_CHECK();
    parts += (int64_t)wgrad_splits(m, k, n) * count;
  }
  if (gb) {
    const int blocks = (int)((m + kColSumRows - 1) / kColSumRows);
    hipLaunchKernelGGL(col_sum_parts_kernel, dim3((unsigned)blocks), dim3(256), 0, st, gy, ldgy, m, n, parts);
    LNRF_LAUNCH_CHECK();
    hipLaunchKernelGGL(dense_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, parts, blocks, (int64_t)n, gb);
    LNRF_LAUNCH_CHECK();
  }
  return LNRF_OK;
}

extern "C" int lnrf_sinusoidal_emb(const float* x, int64_t ldx, int64_t m, int32_t dims,
                                   int32_t freqs, float* out, int64_t ldo, int64_t col_off,
                                   lnrf_stream_t stream) {
  LNRF_CHECK_ARG(x && out, "null pointer");
  LNRF_CHECK_ARG(m >= 0 && dims >= 1 && freqs >= 1 && freqs <= 31 && ldx >= dims, "bad sizes");
  LNRF_CHECK_ARG(ldo >= col_off + (int64_t)dims * 2 * freqs, "output row too short");
  if (m == 0) return LNRF_OK;
  int64_t blocks = (m * dims * freqs + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(sinusoidal_emb_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x,
                     ldx, m, dims, freqs, out, ldo, col_off);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int lnrf_gemm_f32(const float* a, int64_t sa_i, int64_t sa_r, const float* b, int64_t sb_r,
                             int64_t sb_j, float* c, int64_t ldc, const float* bias, int32_t act, int32_t mode,
                             int64_t i_rows, int32_t j_cols, int64_t r_depth, int32_t splits,
                             lnrf_stream_t stream) {
  LNRF_CHECK_ARG(a && b && c, "null pointer");
  LNRF_CHECK_ARG(i_rows >= 0 && j_cols >= 0 && r_depth >= 1 && ldc >= j_cols, "bad sizes");
  LNRF_CHECK_ARG(mode >= 0 && mode <= 2 && act >= 0 && act <= LNRF_ACT_SIGMOID, "bad mode/activation");
  if (mode != 2) splits = 1;
  if (mode == 2 && splits <= 0) {
    const int64_t tiles = ((i_rows + TI - 1) / TI) * ((j_cols + TJ - 1) / TJ);
    splits = (int)((2048 + tiles - 1) / (tiles > 0 ? tiles : 1));
    if (splits > 512) splits = 512;
    const int64_t max_splits = (r_depth + 255) / 256;
    if (splits > max_splits) splits = (int)max_splits;
  }
  return launch_gemm(a, sa_i, sa_r, b, sb_r, sb_j, c, ldc, bias, act, mode, i_rows, j_cols, r_depth, splits,
                     as_stream(stream));
}

// lnrf_gemm_f32 mode 2 with a fixed summation order: C (contiguous rows, ldc == J) += sum over r, the splits of the
// reduction leaving their partial tiles in `scratch` (lnrf_gemm_f32_det_scratch_bytes) and added in order.
static int gemm_det_splits(int64_t I, int J, int64_t R) {
  const int64_t tiles = ((I + TI - 1) / TI) * ((J + TJ - 1) / TJ);
  int64_t splits = (2048 + tiles - 1) / (tiles > 0 ? tiles : 1);
  if (splits > 512) splits = 512;
  const int64_t max_splits = (R + 255) / 256;
  if (splits > max_splits) splits = max_splits;
  return splits < 1 ? 1 : (int)splits;
}
extern "C" int64_t lnrf_gemm_f32_det_scratch_bytes(int64_t i_rows, int32_t j_cols, int64_t r_depth) {
  if (i_rows < 0 || j_cols < 0 || r_depth < 0) return -1;
  return (int64_t)gemm_det_splits(i_rows, j_cols, r_depth) * i_rows * j_cols * (int64_t)sizeof(float) + 256;
}
extern "C" int lnrf_gemm_f32_det(const float* a, int64_t sa_i, int64_t sa_r, const float* b, int64_t sb_r, int64_t sb_j,
                                 float* c, int64_t i_rows, int32_t j_cols, int64_t r_depth, void* scratch,
                                 int64_t scratch_bytes, lnrf_stream_t stream) {
  LNRF_CHECK_ARG(a && b && c, "null pointer");
  LNRF_CHECK_ARG(i_rows >= 0 && j_cols >= 0 && r_depth >= 0, "bad sizes");
  if (i_rows == 0 || j_cols == 0 || r_depth == 0) return LNRF_OK;
  LNRF_CHECK_ARG(scratch && scratch_bytes >= lnrf_gemm_f32_det_scratch_bytes(i_rows, j_cols, r_depth), "scratch too small");
  float* parts = reinterpret_cast<float*>(scratch);
  int used = 0;
  hipStream_t st = as_stream(stream);
  int rc = launch_gemm(a, sa_i, sa_r, b, sb_r, sb_j, parts, j_cols, nullptr, 0, 3, i_rows, j_cols, r_depth,
                       gemm_det_splits(i_rows, j_cols, r_depth), st, Gate{nullptr, 0, 0, 0}, &used);
  if (rc != LNRF_OK) return rc;
  const int64_t count = i_rows * (int64_t)j_cols;
  hipLaunchKernelGGL(dense_fold_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, parts, used, count, c);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}
