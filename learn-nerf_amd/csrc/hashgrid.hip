// hashgrid.hip — multiresolution hash-grid encoding of Instant-NGP as restated by the reference:
// HashTableEncoding.__call__ (learn_nerf/instant_ngp.py:134-208), hash_table_lookup (:211-224),
// MultiresHashTableEncoding (:92-118).  F = 2 features per entry, fp32 tables as in the reference.
//
// MI355X mapping: the gather is bound by L2 / Infinity-Cache / HBM random access, not by FLOPs.  The
// launch is LEVEL-MAJOR: blockIdx.y = level, so all CUs sweep the points of one level at a time and that
// level's table (<= 2 MiB at T = 2^18, 4 MiB at 2^19) stays resident in every XCD's 4 MiB L2 instead of
// 16 tables thrashing it.  The encoding is written feature-major ([L*F][M]) so that both the gather
// kernel's stores and the scatter kernel's gradient loads are fully coalesced; the tiny MLP reads it
// through the strided GEMM.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace lnrf {

constexpr int kMaxLevels = 32;

struct HashGridDesc {  // mirrors lnrf_hashgrid_desc
  int n_levels, feature_dim, smooth, pad_;
  float bbox_min[3], bbox_max[3];
  int grid_size[kMaxLevels];
  int table_size[kMaxLevels];         // entries (rows) of the level's table
  long long table_offset[kMaxLevels]; // first float of the level's table in the flat buffer
  int hashed[kMaxLevels];             // 1: XOR-prime hash, 0: dense x + G (y + G z)
};

struct Corner {
  unsigned base[3];  // floored cell
  float c[3];        // interpolation fraction (after smoothstep if smooth)
  float dc[3];       // d c[a] / d x[a] (0 outside the bounding box: the clip has zero slope there)
};

__device__ __forceinline__ Corner locate(const float x[3], const HashGridDesc& d, int G) {
  Corner r;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float frac = (x[a] - d.bbox_min[a]) / (d.bbox_max[a] - d.bbox_min[a]);   // instant_ngp.py:138-140
    frac = fminf(fmaxf(frac, 0.0f), 1.0f);
    const float fi = d.smooth ? 0.5f + (float)(G - 2) * frac : (float)(G - 1) * frac;  // :141-146
    float fl = floorf(fi);
    fl = fminf(fl, (float)(G - 2));                                           // :150
    float c = fi - fl;                                                        // :152
    const float extent = d.bbox_max[a] - d.bbox_min[a];
    const bool inside = x[a] > d.bbox_min[a] && x[a] < d.bbox_max[a];
    float slope = (d.smooth ? (float)(G - 2) : (float)(G - 1)) / extent;
    if (d.smooth) {
      slope *= 6.0f * c * (1.0f - c);                                         // d/dt of t^2 (3 - 2t)
      c = (c * c) * (3.0f - 2.0f * c);                                        // :153-154
    }
    r.base[a] = (unsigned)fl;                                                 // :156
    r.c[a] = c;
    r.dc[a] = inside ? slope : 0.0f;
  }
  return r;
}

// trilinear weight of corner (xo, yo, zo) and its derivative along direction u: sum_a u_a dw/dx_a
__device__ __forceinline__ float corner_weight(const Corner& k, int xo, int yo, int zo) {
  return (xo ? k.c[0] : 1.0f - k.c[0]) * (yo ? k.c[1] : 1.0f - k.c[1]) * (zo ? k.c[2] : 1.0f - k.c[2]);
}
__device__ __forceinline__ void corner_weight_grad(const Corner& k, int xo, int yo, int zo, float g[3]) {
  const float wx = xo ? k.c[0] : 1.0f - k.c[0], wy = yo ? k.c[1] : 1.0f - k.c[1], wz = zo ? k.c[2] : 1.0f - k.c[2];
  g[0] = (xo ? k.dc[0] : -k.dc[0]) * wy * wz;
  g[1] = (yo ? k.dc[1] : -k.dc[1]) * wx * wz;
  g[2] = (zo ? k.dc[2] : -k.dc[2]) * wx * wy;
}

__device__ __forceinline__ unsigned entry_index(unsigned cx, unsigned cy, unsigned cz, int G, int T, int hashed) {
  if (hashed) {                                                                  // instant_ngp.py:219-223
    const unsigned hv = cx ^ (19349663u * cy) ^ (83492791u * cz);
    // power-of-two tables (the usual case): a mask instead of the ~35-instruction runtime modulo (uniform branch)
    return (T & (T - 1)) == 0 ? (hv & (unsigned)(T - 1)) : hv % (unsigned)T;
  }
  return cx + (unsigned)G * (cy + (unsigned)G * cz);                            // :199-201
}

// one sample of one level: 8 corners x 2 features (u != nullptr: directional derivative weights)
__device__ __forceinline__ void gather_sample(const HashGridDesc& d, int level, int G, int T, int hashed,
                                              const float2* __restrict__ tab, const float* __restrict__ x,
                                              const float* __restrict__ u, int64_t M, float* __restrict__ enc_t,
                                              int64_t m) {
  const float p[3] = {x[m * 3 + 0], x[m * 3 + 1], x[m * 3 + 2]};
  const Corner k = locate(p, d, G);
  float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
  for (int xo = 0; xo < 2; ++xo)
#pragma unroll
    for (int yo = 0; yo < 2; ++yo)
#pragma unroll
      for (int zo = 0; zo < 2; ++zo) {  // instant_ngp.py:160-175: weight = prod(o ? c : 1 - c)
        float w;
        if (u) {
          float gw[3];
          corner_weight_grad(k, xo, yo, zo, gw);
          w = gw[0] * u[m * 3] + gw[1] * u[m * 3 + 1] + gw[2] * u[m * 3 + 2];
        } else {
          w = corner_weight(k, xo, yo, zo);
        }
        const unsigned idx = entry_index(k.base[0] + xo, k.base[1] + yo, k.base[2] + zo, G, T, hashed);
        const float2 v = tab[idx];
        acc.x += w * v.x;
        acc.y += w * v.y;
      }
  enc_t[(int64_t)(2 * level) * M + m] = acc.x;
  enc_t[(int64_t)(2 * level + 1) * M + m] = acc.y;
}

struct LevelList {
  int n;
  int level[kMaxLevels];
};

// enc_t[(2*level + f) * M + m]
// u == nullptr: the encoding.  u != nullptr ([M,3]): its directional derivative (d enc / d x) u.
// STAGE: the level's whole table (dense levels of at most kStageEntries entries, i.e. the 16^3 grids that every
// sample of the batch hits) is first copied into LDS with coalesced 16-byte loads and the 8 corner reads of every
// sample are LDS reads; the launch uses few, long-lived workgroups so that one copy serves thousands of samples.
constexpr int kStageEntries = 8192;  // 64 KiB of float2
template <bool STAGE>
__global__ void hashgrid_fwd_kernel(HashGridDesc d, LevelList ll, const float* __restrict__ tables,
                                    const float* __restrict__ x, const float* __restrict__ u, int64_t M,
                                    float* __restrict__ enc_t) {
  extern __shared__ __attribute__((aligned(16))) float lds_tab[];
  const int level = ll.level[blockIdx.y];
  const int G = d.grid_size[level], T = d.table_size[level], hashed = d.hashed[level];
  const float2* __restrict__ tab = reinterpret_cast<const float2*>(tables + d.table_offset[level]);
  if (STAGE) {
    // table_offset is a multiple of 2 floats (F = 2), i.e. 8-byte aligned: copy as float2, fully coalesced
    float2* lt = reinterpret_cast<float2*>(lds_tab);
    for (int i = threadIdx.x; i < T; i += blockDim.x) lt[i] = tab[i];
    __syncthreads();
    tab = lt;  // generic pointer into LDS: the gathers below become ds_read_b64
  }
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x)
    gather_sample(d, level, G, T, hashed, tab, x, u, M, enc_t, m);
}

// The same gather with the (level, sample chunk) -> workgroup mapping chosen per XCD.  Workgroups are dealt to the
// eight XCDs round-robin (workgroup i runs on XCD i % 8) and every XCD has its own 4 MiB L2, so with a plain
// (chunk, level) grid every level's table is pulled from HBM into all eight L2s.  Here the work of all levels is laid
// out on a line (level after level, chunk after chunk, each chunk weighted by its level's cost) and cut into eight
// equal pieces: an XCD sweeps its piece in order, so a table is fetched by one XCD — two when a cut runs through the
// level — and the pieces end together.  Only speed depends on the dispatch order, not the result.
constexpr int kXcds = 8;
constexpr int kMaxXcdSegs = kMaxLevels + kXcds;
struct XcdPlan {
  int n_segs;
  int first_seg[kXcds + 1];     // segments of XCD x: first_seg[x] .. first_seg[x + 1] - 1
  int seg_level[kMaxXcdSegs];
  int seg_chunk0[kMaxXcdSegs];  // first 256-sample chunk of the segment
  int seg_chunks[kMaxXcdSegs];
};
__global__ __launch_bounds__(256) void hashgrid_fwd_xcd_kernel(HashGridDesc d, XcdPlan plan,
                                                               const float* __restrict__ tables,
                                                               const float* __restrict__ x, const float* __restrict__ u,
                                                               int64_t M, float* __restrict__ enc_t) {
  const int xcd = blockIdx.x % kXcds;
  int slot = blockIdx.x / kXcds;
  int seg = plan.first_seg[xcd];
  const int seg_end = plan.first_seg[xcd + 1];
  while (seg < seg_end && slot >= plan.seg_chunks[seg]) slot -= plan.seg_chunks[seg++];
  if (seg >= seg_end) return;
  const int level = plan.seg_level[seg];
  const int G = d.grid_size[level], T = d.table_size[level], hashed = d.hashed[level];
  const float2* __restrict__ tab = reinterpret_cast<const float2*>(tables + d.table_offset[level]);
  const int64_t m = ((int64_t)plan.seg_chunk0[seg] + slot) * 256 + threadIdx.x;
  if (m < M) gather_sample(d, level, G, T, hashed, tab, x, u, M, enc_t, m);
}

// g_tables[level][idx][f] += w * g_enc_t[(2*level+f)*M + m].  Levels whose whole table fits in LDS
// (G^3 * 8 B <= 64 KiB, i.e. the 16^3 levels that thousands of samples share) are pre-reduced in LDS and
// flushed once per workgroup; the other levels use fp32 atomics directly.

__global__ void hashgrid_bwd_kernel(HashGridDesc d, LevelList ll, const float* __restrict__ x,
                                    const float* __restrict__ u, int64_t M,
                                    const float* __restrict__ g_enc_t, float* __restrict__ g_tables) {
  extern __shared__ __attribute__((aligned(16))) float lds_tab[];
  const int level = ll.level[blockIdx.y];
  const int G = d.grid_size[level], T = d.table_size[level], hashed = d.hashed[level];
  float* __restrict__ gtab = g_tables + d.table_offset[level];
  const bool in_lds = !hashed && (int64_t)T * 2 * 4 <= 64 * 1024;
  if (in_lds) {
    for (int i = threadIdx.x; i < T * 2; i += blockDim.x) lds_tab[i] = 0.0f;
    __syncthreads();
  }
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const float p[3] = {x[m * 3 + 0], x[m * 3 + 1], x[m * 3 + 2]};
    const Corner k = locate(p, d, G);
    const float g0 = g_enc_t[(int64_t)(2 * level) * M + m];
    const float g1 = g_enc_t[(int64_t)(2 * level + 1) * M + m];
#pragma unroll
    for (int xo = 0; xo < 2; ++xo)
#pragma unroll
      for (int yo = 0; yo < 2; ++yo)
#pragma unroll
        for (int zo = 0; zo < 2; ++zo) {
          float w;
          if (u) {
            float gw[3];
            corner_weight_grad(k, xo, yo, zo, gw);
            w = gw[0] * u[m * 3] + gw[1] * u[m * 3 + 1] + gw[2] * u[m * 3 + 2];
          } else {
            w = corner_weight(k, xo, yo, zo);
          }
          const unsigned idx = entry_index(k.base[0] + xo, k.base[1] + yo, k.base[2] + zo, G, T, hashed);
          if (in_lds) {
            atomicAdd(&lds_tab[2 * idx], w * g0);
            atomicAdd(&lds_tab[2 * idx + 1], w * g1);
          } else {
            atomicAdd(gtab + 2 * (int64_t)idx, w * g0);
            atomicAdd(gtab + 2 * (int64_t)idx + 1, w * g1);
          }
        }
  }
  if (in_lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < T * 2; i += blockDim.x) {
      const float v = lds_tab[i];
      if (v != 0.0f) atomicAdd(gtab + i, v);
    }
  }
}

// g_x[m][a] = sum_levels sum_f g_enc_t[(2l+f)*M + m] * sum_corners (d w / d x_a) table[idx][f]
// (transpose of d enc / d x applied to g_enc; the analytic normal of the Ref-NeRF head on a hash grid)
__global__ void hashgrid_input_grad_kernel(HashGridDesc d, const float* __restrict__ tables,
                                           const float* __restrict__ x, int64_t M,
                                           const float* __restrict__ g_enc_t, float* __restrict__ g_x) {
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const float p[3] = {x[m * 3 + 0], x[m * 3 + 1], x[m * 3 + 2]};
    float acc[3] = {0.0f, 0.0f, 0.0f};
    for (int level = 0; level < d.n_levels; ++level) {
      const int G = d.grid_size[level], T = d.table_size[level], hashed = d.hashed[level];
      const float2* __restrict__ tab = reinterpret_cast<const float2*>(tables + d.table_offset[level]);
      const Corner k = locate(p, d, G);
      const float g0 = g_enc_t[(int64_t)(2 * level) * M + m];
      const float g1 = g_enc_t[(int64_t)(2 * level + 1) * M + m];
#pragma unroll
      for (int xo = 0; xo < 2; ++xo)
#pragma unroll
        for (int yo = 0; yo < 2; ++yo)
#pragma unroll
          for (int zo = 0; zo < 2; ++zo) {
            float gw[3];
            corner_weight_grad(k, xo, yo, zo, gw);
            const float2 v = tab[entry_index(k.base[0] + xo, k.base[1] + yo, k.base[2] + zo, G, T, hashed)];
            const float s = g0 * v.x + g1 * v.y;
            acc[0] += gw[0] * s;
            acc[1] += gw[1] * s;
            acc[2] += gw[2] * s;
          }
    }
    g_x[m * 3 + 0] = acc[0];
    g_x[m * 3 + 1] = acc[1];
    g_x[m * 3 + 2] = acc[2];
  }
}

// Dense levels whose table does not fit in LDS (G = 32, 64: thousands of samples per entry, so direct
// atomics contend).  blockIdx.y = 8K-entry slice of the table, blockIdx.z = level in `ll`; every workgroup
// scans its chunk of points, accumulates only the corners that fall into its slice in LDS, and flushes the
// slice with contiguous atomics.  The index arithmetic is recomputed once per slice (cheap next to the
// contended atomics it replaces).
constexpr int kSliceEntries = 8192;  // 64 KiB of float2
__global__ void hashgrid_bwd_sliced_kernel(HashGridDesc d, LevelList ll, const float* __restrict__ x,
                                           const float* __restrict__ u, int64_t M,
                                           const float* __restrict__ g_enc_t, float* __restrict__ g_tables) {
  extern __shared__ __attribute__((aligned(16))) float lds_tab[];
  const int level = ll.level[blockIdx.z];
  const int G = d.grid_size[level], T = d.table_size[level], hashed = d.hashed[level];
  const unsigned slice0 = blockIdx.y * (unsigned)kSliceEntries;
  if (slice0 >= (unsigned)T) return;
  float* __restrict__ gtab = g_tables + d.table_offset[level];
  for (int i = threadIdx.x; i < kSliceEntries * 2; i += blockDim.x) lds_tab[i] = 0.0f;
  __syncthreads();
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    const float p[3] = {x[m * 3 + 0], x[m * 3 + 1], x[m * 3 + 2]};
    const Corner k = locate(p, d, G);
    const float g0 = g_enc_t[(int64_t)(2 * level) * M + m];
    const float g1 = g_enc_t[(int64_t)(2 * level + 1) * M + m];
#pragma unroll
    for (int xo = 0; xo < 2; ++xo)
#pragma unroll
      for (int yo = 0; yo < 2; ++yo)
#pragma unroll
        for (int zo = 0; zo < 2; ++zo) {
          const unsigned idx = entry_index(k.base[0] + xo, k.base[1] + yo, k.base[2] + zo, G, T, hashed);
          const unsigned rel = idx - slice0;
          if (rel < (unsigned)kSliceEntries) {
            float w;
            if (u) {
              float gw[3];
              corner_weight_grad(k, xo, yo, zo, gw);
              w = gw[0] * u[m * 3] + gw[1] * u[m * 3 + 1] + gw[2] * u[m * 3 + 2];
            } else {
              w = corner_weight(k, xo, yo, zo);
            }
            atomicAdd(&lds_tab[2 * rel], w * g0);
            atomicAdd(&lds_tab[2 * rel + 1], w * g1);
          }
        }
  }
  __syncthreads();
  const int n = (T - (int)slice0 < kSliceEntries ? T - (int)slice0 : kSliceEntries) * 2;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = lds_tab[i];
    if (v != 0.0f) atomicAdd(gtab + 2 * (int64_t)slice0 + i, v);
  }
}

// ---------------------------------------------------------------------------------------------------
// Bucketed scatter.  LDS float atomics are slow on gfx950 (ds_add_f32: ~0.3 lanes/clk/CU, measured with
// tools/lds_atomic_bench.hip; ds_add_u64: ~5.6) and memory-side float atomics slower still, so the gradient
// table is reduced with 64-bit INTEGER LDS atomics on fixed-point values:
// Pass A bins the 8 (entry, w*g0, w*g1) contributions of every sample by 4K-entry table slice: a workgroup
// counts its chunk per slice in LDS, reserves room in each slice's global bucket with ONE atomic per
// (workgroup, slice), writes its tuples there and records the level's max |value|.
// Pass B gives every (level, slice) bucket to one workgroup (few-slice levels: several), which converts the
// values to fixed point with the power-of-two scale 2^(62 - lg - e) (level max < 2^e, fewer than 2^lg tuples
// in the bucket: the sum cannot overflow; resolution 2^-46 of the max for a 64K-tuple bucket), accumulates
// them with ds_add_u64 and adds the slice to the gradient table.  Integer accumulation is order-independent, so unsplit slices are bit-reproducible.
// Buckets are sized 1.25x the mean for hashed levels (the hash spreads samples evenly), 4x for dense levels;
// overflowing tuples fall back to global atomics.
struct BucketTuple {
  unsigned rel;  // entry index within the 4K-entry slice
  float v0, v1;
};
// Plain scatter (u == nullptr): the trilinear weights are <= 1, so every contribution is bounded by the level's
// max |d loss / d enc|, which the producer of the gradient rows hands in (lnrf_ngp_mlp_bwd).  The values are then
// quantised right away to 26-bit fixed point relative to that bound (2^-25 of the level maximum, about fp32's own
// resolution at the maximum) and a tuple is 8 bytes: {q0 : 26 | rel[0:6], q1 : 26 | rel[6:12]} — a third less tuple
// traffic than {rel, float, float}, integer sums in the reduce pass without any floating-point conversion.
struct PackedTuple {
  unsigned w0, w1;
};
constexpr int kQuantBits = 26;
__device__ __forceinline__ PackedTuple pack_tuple(unsigned rel, float v0, float v1, float scale) {
  // saturate (a caller-supplied bound that is too small must not wrap through the 26-bit field into the other sign)
  constexpr float kQmax = (float)((1 << (kQuantBits - 1)) - 1);
  const int q0 = __float2int_rn(fminf(fmaxf(v0 * scale, -kQmax), kQmax));
  const int q1 = __float2int_rn(fminf(fmaxf(v1 * scale, -kQmax), kQmax));
  PackedTuple t;
  t.w0 = ((unsigned)q0 & 0x03FFFFFFu) | ((rel & 63u) << 26);
  t.w1 = ((unsigned)q1 & 0x03FFFFFFu) | ((rel >> 6) << 26);
  return t;
}

constexpr int kBinChunk = 2048;  // samples per binning workgroup
constexpr int kBucketEntries = 4096;  // 64 KiB of int64 x 2 features
constexpr int kMaxSlices = 256;       // table <= 1M entries

struct BucketPlan {
  int n;                 // bucketed levels
  int level[kMaxLevels];
  int slices[kMaxLevels];
  int split[kMaxLevels];             // reduce workgroups per bucket
  int wg_off[kMaxLevels + 1];        // first reduce workgroup of the level
  long long tuple_off[kMaxLevels];   // first tuple of the level's buckets
  long long cursor_off[kMaxLevels];  // first cursor of the level
  long long cap[kMaxLevels];         // tuples per bucket of the level
};

// One slot per valid lane in counter[b] with one LDS atomic per RUN of equal b in neighbouring lanes
// (neighbouring lanes = neighbouring samples of a ray: on dense levels they share the slice, and same-address
// LDS atomics serialise at ~0.5 lanes/clk).  Must be called by the whole wave.  Returns the lane's offset.
__device__ __forceinline__ unsigned run_reserve(unsigned* counter, unsigned b, bool valid) {
  const int lane = threadIdx.x & 63;
  const unsigned bb = valid ? b : 0xFFFFFFFFu;
  const unsigned prev = __shfl_up(bb, 1, 64);
  const bool head = lane == 0 || bb != prev;
  const unsigned long long heads = __ballot(head);
  const int hl = 63 - __clzll((long long)(heads & ((2ull << lane) - 1ull)));  // head lane of my run
  const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
  const int len = above ? __ffsll((long long)above) : 64 - lane;              // run length (head lanes)
  unsigned base = 0u;
  if (head && valid) base = atomicAdd(&counter[bb], (unsigned)len);
  base = __shfl(base, hl, 64);
  return base + (unsigned)(lane - hl);
}

template <bool QUANT>
__global__ void hashgrid_bin_kernel(HashGridDesc d, BucketPlan plan, const float* __restrict__ x,
                                    const float* __restrict__ u, int64_t M, const float* __restrict__ g_enc_t,
                                    BucketTuple* __restrict__ tuples, unsigned* __restrict__ cursors,
                                    unsigned* __restrict__ level_max, float* __restrict__ g_tables) {
  __shared__ unsigned s_count[kMaxSlices];
  __shared__ unsigned s_base[kMaxSlices];
  const int li = blockIdx.y;
  const int level = plan.level[li];
  const int S = plan.slices[li];
  const int G = d.grid_size[level], T = d.table_size[level], hashed = d.hashed[level];
  BucketTuple* __restrict__ tup = tuples + plan.tuple_off[li];
  PackedTuple* __restrict__ ptup = reinterpret_cast<PackedTuple*>(tuples) + plan.tuple_off[li];
  float qscale = 0.0f;  // QUANT: 2^(25 - e) with the caller's bound < 2^e
  if (QUANT) {
    int e = 0;
    frexpf(__uint_as_float(level_max[plan.level[li]]), &e);
    qscale = ldexpf(1.0f, kQuantBits - 1 - e);
  }
  unsigned* __restrict__ cur = cursors + plan.cursor_off[li];
  const long long cap = plan.cap[li];
  for (int i = threadIdx.x; i < S; i += blockDim.x) s_count[i] = 0u;
  __syncthreads();
  const int64_t m0 = (int64_t)blockIdx.x * kBinChunk;
  constexpr int kIters = kBinChunk / 256;  // blockDim.x == 256; uniform trip count (run_reserve needs the whole wave)
  // pass 1: count per slice
  for (int it = 0; it < kIters; ++it) {
    const int64_t m = m0 + it * 256 + threadIdx.x;
    const bool valid = m < M;
    const int64_t mm = valid ? m : M - 1;
    const float p[3] = {x[mm * 3 + 0], x[mm * 3 + 1], x[mm * 3 + 2]};
    const Corner k = locate(p, d, G);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const unsigned idx = entry_index(k.base[0] + (c >> 2), k.base[1] + ((c >> 1) & 1), k.base[2] + (c & 1), G, T, hashed);
      // hashed levels spread a wave over many slices (little same-address contention): plain LDS atomics are
      // cheaper than the run detection; dense levels put whole runs of lanes into one slice
      if (hashed) {
        if (valid) atomicAdd(&s_count[idx / kBucketEntries], 1u);
      } else {
        run_reserve(s_count, idx / kBucketEntries, valid);
      }
    }
  }
  __syncthreads();
  // reserve room in the global buckets; s_count becomes the running local offset
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    s_base[i] = s_count[i] ? atomicAdd(&cur[i], s_count[i]) : 0u;
    s_count[i] = 0u;
  }
  __syncthreads();
  // pass 2: write the tuples
  float vmax = 0.0f;
  for (int it = 0; it < kIters; ++it) {
    const int64_t m = m0 + it * 256 + threadIdx.x;
    const bool valid = m < M;
    const int64_t mm = valid ? m : M - 1;
    const float p[3] = {x[mm * 3 + 0], x[mm * 3 + 1], x[mm * 3 + 2]};
    const Corner k = locate(p, d, G);
    const float g0 = valid ? g_enc_t[(int64_t)(2 * level) * M + mm] : 0.0f;
    const float g1 = valid ? g_enc_t[(int64_t)(2 * level + 1) * M + mm] : 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int xo = c >> 2, yo = (c >> 1) & 1, zo = c & 1;
      float w;
      if (u) {
        float gw[3];
        corner_weight_grad(k, xo, yo, zo, gw);
        w = gw[0] * u[mm * 3] + gw[1] * u[mm * 3 + 1] + gw[2] * u[mm * 3 + 2];
      } else {
        w = corner_weight(k, xo, yo, zo);
      }
      const unsigned idx = entry_index(k.base[0] + xo, k.base[1] + yo, k.base[2] + zo, G, T, hashed);
      const unsigned b = idx / kBucketEntries;
      const unsigned off = hashed ? (valid ? atomicAdd(&s_count[b], 1u) : 0u) : run_reserve(s_count, b, valid);
      if (valid) {
        const long long pos = (long long)s_base[b] + off;
        // NaN / Inf contributions cannot be carried in fixed point: they take the float atomics below, which propagate
        // them into the table as the reference's scatter-add would, and their tuple slot carries zeros
        const float v0 = w * g0, v1 = w * g1;
        const bool finite = fabsf(v0) <= 3.0e38f && fabsf(v1) <= 3.0e38f;
        const float t0 = finite ? v0 : 0.0f, t1 = finite ? v1 : 0.0f;
        if (!QUANT) vmax = fmaxf(vmax, fmaxf(fabsf(t0), fabsf(t1)));
        if (pos < cap) {
          if (QUANT) {
            ptup[(long long)b * cap + pos] = pack_tuple(idx - b * kBucketEntries, t0, t1, qscale);
          } else {
            BucketTuple t;
            t.rel = idx - b * kBucketEntries;
            t.v0 = t0;
            t.v1 = t1;
            tup[(long long)b * cap + pos] = t;
          }
        }
        if (pos >= cap || !finite) {  // bucket full (rare), or a non-finite value
          float* gt = g_tables + d.table_offset[level] + 2 * (int64_t)idx;
          atomicAdd(gt, v0);
          atomicAdd(gt + 1, v1);
        }
      }
    }
  }
  if (!QUANT) {
    // level max |value| (fixed-point scale of pass B): non-negative floats order like their bit patterns
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    if ((threadIdx.x & 63) == 0 && vmax > 0.0f) atomicMax(&level_max[plan.level[li]], __float_as_uint(vmax));
  }
}

// Pass B.  A bucket is split among plan.split[level] workgroups (few-slice levels would otherwise leave
// most of the chip idle; their partial slices are flushed with atomics); blockIdx.x -> (level, bucket, part).
// The tuple stream is read kReduceUnroll tuples per thread ahead of the LDS atomics that consume it.
constexpr int kReduceThreads = 512;
constexpr int kReduceUnroll = 4;
template <bool QUANT>
__global__ __launch_bounds__(kReduceThreads) void hashgrid_reduce_kernel(
    HashGridDesc d, BucketPlan plan, const BucketTuple* __restrict__ tuples, const unsigned* __restrict__ cursors,
    const unsigned* __restrict__ level_max, float* __restrict__ g_tables) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds_q[];
  int li = 0;
#pragma unroll 1
  for (int i = 1; i < plan.n; ++i)
    if ((int)blockIdx.x >= plan.wg_off[i]) li = i;
  const int level = plan.level[li];
  const int split = plan.split[li];
  const int local = (int)blockIdx.x - plan.wg_off[li];
  const int b = local / split, part = local - b * split;
  const int T = d.table_size[level];
  for (int i = threadIdx.x; i < kBucketEntries * 2; i += kReduceThreads) lds_q[i] = 0ull;
  __syncthreads();
  long long count = cursors[plan.cursor_off[li] + b];
  if (count > plan.cap[li]) count = plan.cap[li];
  const long long lo = count * part / split, hi = count * (part + 1) / split;
  // power-of-two scale: level max < 2^e and at most n = hi - lo < 2^lg terms per entry, so values scaled by
  // 2^(62 - lg - e) cannot overflow the signed 64-bit sum whatever the sample distribution is
  int e = 0;
  frexpf(__uint_as_float(level_max[plan.level[li]]), &e);
  const int lg = 64 - __clzll((hi - lo) | 1ll);
  const int fixed_bits = 62 - lg;  // 46 for a 64K-tuple bucket
  const double scale = ldexp(1.0, fixed_bits - e);
  const double inv_scale = QUANT ? ldexp(1.0, e - (kQuantBits - 1)) : ldexp(1.0, e - fixed_bits);
  const BucketTuple* __restrict__ tup = tuples + plan.tuple_off[li] + (long long)b * plan.cap[li];
  if (QUANT) {
    // 8-byte tuples: 26-bit values (sign-extended) summed as they are — 2^38 of them fit a signed 64-bit sum
    const uint2* __restrict__ pt = reinterpret_cast<const uint2*>(reinterpret_cast<const PackedTuple*>(tuples) +
                                                                  plan.tuple_off[li] + (long long)b * plan.cap[li]);
    for (long long i0 = lo + threadIdx.x; i0 < hi; i0 += (long long)kReduceUnroll * kReduceThreads) {
      uint2 t[kReduceUnroll];
      bool ok[kReduceUnroll];
#pragma unroll
      for (int q = 0; q < kReduceUnroll; ++q) {
        const long long i = i0 + (long long)q * kReduceThreads;
        ok[q] = i < hi;
        t[q] = pt[ok[q] ? i : hi - 1];
      }
#pragma unroll
      for (int q = 0; q < kReduceUnroll; ++q) {
        const unsigned rel = (t[q].x >> 26) | ((t[q].y >> 26) << 6);
        const long long q0 = ok[q] ? (long long)(((int)(t[q].x << 6)) >> 6) : 0ll;
        const long long q1 = ok[q] ? (long long)(((int)(t[q].y << 6)) >> 6) : 0ll;
        atomicAdd(&lds_q[2 * rel], (unsigned long long)q0);
        atomicAdd(&lds_q[2 * rel + 1], (unsigned long long)q1);
      }
    }
  } else
  for (long long i0 = lo + threadIdx.x; i0 < hi; i0 += (long long)kReduceUnroll * kReduceThreads) {
    BucketTuple t[kReduceUnroll];
    // branch-free: out-of-range slots re-read the last tuple and add zeros (a guarded load makes the
    // compiler wait for every load separately)
#pragma unroll
    for (int q = 0; q < kReduceUnroll; ++q) {
      const long long i = i0 + (long long)q * kReduceThreads;
      const bool ok = i < hi;
      t[q] = tup[ok ? i : hi - 1];
      t[q].v0 = ok ? t[q].v0 : 0.0f;
      t[q].v1 = ok ? t[q].v1 : 0.0f;
    }
#pragma unroll
    for (int q = 0; q < kReduceUnroll; ++q) {
      const long long q0 = __double2ll_rn((double)t[q].v0 * scale);
      const long long q1 = __double2ll_rn((double)t[q].v1 * scale);
      atomicAdd(&lds_q[2 * t[q].rel], (unsigned long long)q0);
      atomicAdd(&lds_q[2 * t[q].rel + 1], (unsigned long long)q1);
    }
  }
  __syncthreads();
  const long long slice0 = (long long)b * kBucketEntries;
  const int n = (int)((T - slice0 < kBucketEntries ? T - slice0 : kBucketEntries) * 2);
  float* __restrict__ gt = g_tables + d.table_offset[level] + 2 * slice0;
  for (int i = threadIdx.x; i < n; i += kReduceThreads) {
    const long long q = (long long)lds_q[i];
    if (q != 0) {
      const float v = (float)((double)q * inv_scale);
      if (split == 1) gt[i] += v;  // this workgroup is the only writer of the slice in this launch
      else atomicAdd(gt + i, v);
    }
  }
}

}  // namespace lnrf

using namespace lnrf;

static int check_desc(const lnrf_hashgrid_desc* d) {
  if (!d) { set_error("hashgrid: null descriptor"); return LNRF_ERR_ARG; }
  if (d->n_levels < 1 || d->n_levels > kMaxLevels) { set_error("hashgrid: n_levels out of range"); return LNRF_ERR_ARG; }
  if (d->feature_dim != 2) {
    set_error("hashgrid: only feature_dim == 2 is implemented (the reference default)");
    return LNRF_ERR_UNSUPPORTED;
  }
  for (int l = 0; l < d->n_levels; ++l)
    if (d->grid_size[l] < 2 || d->table_size[l] < 1) { set_error("hashgrid: bad level"); return LNRF_ERR_ARG; }
  return LNRF_OK;
}

static_assert(sizeof(HashGridDesc) == sizeof(lnrf_hashgrid_desc), "descriptor layout");

// experiment builds only (common.h): LNRF_HASHGRID_LDS=0 turns the LDS-staged gather off, LNRF_HASHGRID_XCD=0 selects the
// plain (chunk, level) grid of the gather
static bool lds_staging_enabled() {
  static const bool on = !exp_env_is("LNRF_HASHGRID_LDS", '0');
  return on;
}
static bool xcd_mapping_enabled() {
  static const bool on = !exp_env_is("LNRF_HASHGRID_XCD", '0');
  return on;
}

// Relative cost of one sample of a level in the gather, measured alone at 786,432 samples (tools/
// hashgrid_level_probe.py): dense levels 12 us, hashed levels 17 us at 4 cells per entry, 27 us at 32, 31 us beyond
// (more and more of the 8 corners fall into different cache lines).
static double gather_cost(const HashGridDesc& d, int l) {
  if (!d.hashed[l]) return 12.0;
  const double cells = (double)d.grid_size[l] * d.grid_size[l] * d.grid_size[l] / (double)d.table_size[l];
  return cells <= 4.0 ? 17.0 : (cells <= 32.0 ? 27.0 : 31.0);
}

static XcdPlan make_xcd_plan(const HashGridDesc& d, const LevelList& levels, int64_t m, unsigned* grid) {
  XcdPlan p;
  const int chunks = (int)((m + 255) / 256);
  double total = 0.0;
  for (int i = 0; i < levels.n; ++i) total += gather_cost(d, levels.level[i]) * chunks;
  const double share = total / kXcds;
  p.n_segs = 0;
  int xcd = 0, li = 0, next_chunk = 0, max_slots = 0, slots = 0;
  double filled = 0.0;
  p.first_seg[0] = 0;
  while (li < levels.n) {
    const double c = gather_cost(d, levels.level[li]);
    // chunks of this level that still fit into the XCD's share (the last XCD takes whatever is left)
    int take = chunks - next_chunk;
    if (xcd < kXcds - 1) {
      const int fit = (int)((share - filled) / c + 0.5);
      if (fit < take) take = fit;
    }
    if (take > 0) {
      p.seg_level[p.n_segs] = levels.level[li];
      p.seg_chunk0[p.n_segs] = next_chunk;
      p.seg_chunks[p.n_segs] = take;
      ++p.n_segs;
      next_chunk += take;
      filled += take * c;
      slots += take;
    }
    if (next_chunk >= chunks) {
      ++li;
      next_chunk = 0;
    } else {  // the XCD is full: the rest of the level goes to the next one
      if (slots > max_slots) max_slots = slots;
      ++xcd;
      p.first_seg[xcd] = p.n_segs;
      filled = 0.0;
      slots = 0;
    }
  }
  if (slots > max_slots) max_slots = slots;
  for (int x2 = xcd + 1; x2 <= kXcds; ++x2) p.first_seg[x2] = p.n_segs;
  *grid = (unsigned)max_slots * kXcds;
  // every chunk of every level exactly once, in order (otherwise the caller uses the plain grid)
  int64_t covered = 0;
  bool ok = p.n_segs <= kMaxXcdSegs;
  for (int i = 0, s = 0; ok && i < levels.n; ++i) {
    int at = 0;
    while (s < p.n_segs && p.seg_level[s] == levels.level[i] && p.seg_chunk0[s] == at) at += p.seg_chunks[s++];
    ok = at == chunks;
    covered += at;
  }
  if (!ok || covered != (int64_t)chunks * levels.n) *grid = 0;
  return p;
}

extern "C" int lnrf_hashgrid_fwd(const lnrf_hashgrid_desc* desc, const float* tables, const float* x, int64_t m,
                                 float* enc_t, lnrf_stream_t stream) {
  return lnrf_hashgrid_jvp(desc, tables, x, nullptr, m, enc_t, stream);
}

extern "C" int lnrf_hashgrid_jvp(const lnrf_hashgrid_desc* desc, const float* tables, const float* x,
                                 const float* u, int64_t m, float* enc_t, lnrf_stream_t stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  LNRF_CHECK_ARG(tables && x && enc_t, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  HashGridDesc d;
  memcpy((void*)&d, (const void*)desc, sizeof(d));
  // dense levels whose table fits in 64 KiB of LDS are staged there (LDS gather); the others gather from L2 / HBM
  LevelList staged, direct;
  staged.n = direct.n = 0;
  for (int l = 0; l < d.n_levels; ++l) {
    if (lds_staging_enabled() && !d.hashed[l] && d.table_size[l] <= kStageEntries && m >= 16384)
      staged.level[staged.n++] = l;
    else direct.level[direct.n++] = l;
  }
  if (direct.n > 0 && xcd_mapping_enabled() && (m + 255) / 256 * direct.n <= (int64_t)1 << 28) {
    unsigned grid = 0;
    const XcdPlan plan = make_xcd_plan(d, direct, m, &grid);
    if (grid == 0) {
      set_error("hashgrid: inconsistent XCD plan");
      return LNRF_ERR_ARG;
    }
    hipLaunchKernelGGL(hashgrid_fwd_xcd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), d, plan, tables, x, u, m,
                       enc_t);
    LNRF_LAUNCH_CHECK();
  } else if (direct.n > 0) {
    int64_t bx = (m + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(hashgrid_fwd_kernel<false>, dim3((unsigned)bx, (unsigned)direct.n), dim3(256), 0,
                       as_stream(stream), d, direct, tables, x, u, m, enc_t);
    LNRF_LAUNCH_CHECK();
  }
  if (staged.n > 0) {
    // one table copy (32 KiB for a 16^3 grid, read from L2) per 512-thread workgroup and >= 1024 samples; LDS is
    // sized to the largest staged table so that four such workgroups share a CU
    int max_rows = 0;
    for (int i = 0; i < staged.n; ++i)
      if (d.table_size[staged.level[i]] > max_rows) max_rows = d.table_size[staged.level[i]];
    int64_t bx = (m + 1023) / 1024;
    if (bx > 1024) bx = 1024;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hashgrid_fwd_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kStageEntries * 8);
    if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(max dynamic LDS)");
    hipLaunchKernelGGL(hashgrid_fwd_kernel<true>, dim3((unsigned)bx, (unsigned)staged.n), dim3(512), max_rows * 8,
                       as_stream(stream), d, staged, tables, x, u, m, enc_t);
    LNRF_LAUNCH_CHECK();
  }
  return LNRF_OK;
}

extern "C" int lnrf_hashgrid_bwd(const lnrf_hashgrid_desc* desc, const float* x, int64_t m, const float* g_enc_t,
                                 float* g_tables, lnrf_stream_t stream) {
  return lnrf_hashgrid_bwd_dir(desc, x, nullptr, m, g_enc_t, g_tables, stream);
}

extern "C" int lnrf_hashgrid_input_grad(const lnrf_hashgrid_desc* desc, const float* tables, const float* x,
                                        int64_t m, const float* g_enc_t, float* g_x, lnrf_stream_t stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  LNRF_CHECK_ARG(tables && x && g_enc_t && g_x, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  HashGridDesc d;
  memcpy((void*)&d, (const void*)desc, sizeof(d));
  int64_t bx = (m + 255) / 256;
  if (bx > 8192) bx = 8192;
  hipLaunchKernelGGL(hashgrid_input_grad_kernel, dim3((unsigned)bx), dim3(256), 0, as_stream(stream), d, tables, x, m,
                     g_enc_t, g_x);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

// Every level whose table has at most kMaxSlices 4K-entry slices is bucketed (hashed or dense).
static int bucket_slices(const HashGridDesc& d, int l) { return (d.table_size[l] + kBucketEntries - 1) / kBucketEntries; }
static bool level_is_bucketed(const HashGridDesc& d, int l) { return bucket_slices(d, l) <= kMaxSlices; }
static BucketPlan make_plan(const HashGridDesc& d, int64_t m, int64_t* tuple_total, int64_t* cursor_total) {
  BucketPlan p;
  p.n = 0;
  int64_t toff = 0, coff = 0;
  int wg = 0;
  for (int l = 0; l < d.n_levels; ++l) {
    if (!level_is_bucketed(d, l)) continue;
    const int slices = bucket_slices(d, l);
    // capacity: the hash spreads samples evenly (1.25x the mean); dense levels follow the scene geometry
    // (4x the mean).  Overflowing tuples fall back to atomics in the bin kernel.
    const double mean = (double)m * 8.0 / (double)slices;
    long long cap = (long long)(mean * (d.hashed[l] ? 1.25 : 4.0)) + 4096;
    if (cap > m * 8) cap = m * 8;
    // One workgroup streams about 1 tuple/ns, so buckets of more than ~64K tuples (few-slice levels) are
    // split; a split bucket is flushed with (contiguous) atomics, an unsplit one with plain stores.
    int split = mean <= 81920.0 ? 1 : (int)((mean + 65535.0) / 65536.0);
    if (split > 64) split = 64;
    p.level[p.n] = l;
    p.slices[p.n] = slices;
    p.split[p.n] = split;
    p.wg_off[p.n] = wg;
    p.cap[p.n] = cap;
    p.tuple_off[p.n] = toff;
    p.cursor_off[p.n] = coff;
    wg += slices * split;
    toff += (int64_t)slices * cap;
    coff += slices;
    ++p.n;
  }
  p.wg_off[p.n] = wg;
  *tuple_total = toff;
  *cursor_total = coff;
  return p;
}

extern "C" int64_t lnrf_hashgrid_bwd_scratch_bytes(const lnrf_hashgrid_desc* desc, int64_t m) {
  if (check_desc(desc)) return -1;
  HashGridDesc d;
  memcpy((void*)&d, (const void*)desc, sizeof(d));
  int64_t tuples = 0, cursors = 0;
  make_plan(d, m, &tuples, &cursors);
  return (((cursors + kMaxLevels) * 4 + 255) / 256) * 256 + tuples * (int64_t)sizeof(BucketTuple);
}

extern "C" int lnrf_hashgrid_bwd_dir(const lnrf_hashgrid_desc* desc, const float* x, const float* u, int64_t m,
                                     const float* g_enc_t, float* g_tables, lnrf_stream_t stream) {
  return lnrf_hashgrid_bwd_bucketed(desc, x, u, m, g_enc_t, nullptr, g_tables, nullptr, 0, stream);
}

extern "C" int lnrf_hashgrid_bwd_bucketed(const lnrf_hashgrid_desc* desc, const float* x, const float* u, int64_t m,
                                          const float* g_enc_t, const float* level_absmax, float* g_tables,
                                          void* scratch, int64_t scratch_bytes, lnrf_stream_t stream) {
  int rc = check_desc(desc);
  if (rc) return rc;
  LNRF_CHECK_ARG(x && g_enc_t && g_tables, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  HashGridDesc d;
  memcpy((void*)&d, (const void*)desc, sizeof(d));
  // bucketed path when the caller provides scratch memory; without it: LDS-sliced / direct atomic kernels
  int64_t tuple_total = 0, cursor_total = 0;
  const BucketPlan plan = make_plan(d, m, &tuple_total, &cursor_total);
  const int64_t cursor_bytes = (((cursor_total + kMaxLevels) * 4 + 255) / 256) * 256;  // cursors + level max
  const bool use_buckets = scratch && plan.n > 0 &&
                           scratch_bytes >= cursor_bytes + tuple_total * (int64_t)sizeof(BucketTuple);
  if (use_buckets) {
    unsigned* cursors = reinterpret_cast<unsigned*>(scratch);
    unsigned* level_max = cursors + cursor_total;
    BucketTuple* tuples = reinterpret_cast<BucketTuple*>(reinterpret_cast<char*>(scratch) + cursor_bytes);
    hipError_t e = hipMemsetAsync(cursors, 0, (size_t)cursor_bytes, as_stream(stream));
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(bucket cursors)");
    const unsigned chunks = (unsigned)((m + kBinChunk - 1) / kBinChunk);
    if (u == nullptr && level_absmax != nullptr) {
      // the caller hands in a bound of |g| per level (lnrf_ngp_mlp_bwd finds it while it writes the rows; weights <= 1):
      // quantised 8-byte tuples, 2^-25 of that bound per contribution.  This is the fused bf16 path's scatter; WITHOUT a
      // bound (the exact-fp32 path, any other caller) the fp32-tuple form below is used: 12-byte tuples, level maximum
      // found while binning, 64-bit fixed point of 2^-46 of the maximum per contribution in the reduce pass.
      level_max = reinterpret_cast<unsigned*>(const_cast<float*>(level_absmax));  // read only from here on
      hipLaunchKernelGGL(hashgrid_bin_kernel<true>, dim3(chunks, (unsigned)plan.n), dim3(256), 0, as_stream(stream), d,
                         plan, x, u, m, g_enc_t, tuples, cursors, level_max, g_tables);
      LNRF_LAUNCH_CHECK();
      hipLaunchKernelGGL(hashgrid_reduce_kernel<true>, dim3((unsigned)plan.wg_off[plan.n]), dim3(kReduceThreads),
                         64 * 1024, as_stream(stream), d, plan, tuples, cursors, level_max, g_tables);
    } else {
      // no bound handed in, or directional-derivative weights (unbounded): fp32 tuples, level maximum found while binning
      hipLaunchKernelGGL(hashgrid_bin_kernel<false>, dim3(chunks, (unsigned)plan.n), dim3(256), 0, as_stream(stream), d,
                         plan, x, u, m, g_enc_t, tuples, cursors, level_max, g_tables);
      LNRF_LAUNCH_CHECK();
      hipLaunchKernelGGL(hashgrid_reduce_kernel<false>, dim3((unsigned)plan.wg_off[plan.n]), dim3(kReduceThreads),
                         64 * 1024, as_stream(stream), d, plan, tuples, cursors, level_max, g_tables);
    }
    LNRF_LAUNCH_CHECK();
  }
  // levels with 8K < entries <= 512K go to the sliced LDS kernel, the rest to the direct kernel
  LevelList direct, sliced;
  direct.n = sliced.n = 0;
  int max_slices = 1;
  for (int l = 0; l < d.n_levels; ++l) {
    if (use_buckets && level_is_bucketed(d, l)) continue;
    const int slices = (d.table_size[l] + kSliceEntries - 1) / kSliceEntries;
    if (slices > 1 && slices <= 64) {
      sliced.level[sliced.n++] = l;
      if (slices > max_slices) max_slices = slices;
    } else {
      direct.level[direct.n++] = l;
    }
  }
  if (direct.n > 0) {
    int64_t bx = (m + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(hashgrid_bwd_kernel, dim3((unsigned)bx, (unsigned)direct.n), dim3(256), 64 * 1024,
                       as_stream(stream), d, direct, x, u, m, g_enc_t, g_tables);
    LNRF_LAUNCH_CHECK();
  }
  if (sliced.n > 0) {
    int64_t bx = 1024 / max_slices;
    if (bx < 8) bx = 8;
    const int64_t max_bx = (m + 255) / 256;
    if (bx > max_bx) bx = max_bx;
    hipLaunchKernelGGL(hashgrid_bwd_sliced_kernel, dim3((unsigned)bx, (unsigned)max_slices, (unsigned)sliced.n),
                       dim3(256), 64 * 1024, as_stream(stream), d, sliced, x, u, m, g_enc_t, g_tables);
  }
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}
