// nerf_bwd_ls.hip — the NeRFModel backward (jax.grad through model.py:43-62) as a LAYER-STATIONARY pipeline.
//
// Why.  With a chain launch and a weight-gradient launch (nerf_mlp.hip) every evaluation and 256-wide layer moves
// 2 KiB through HBM: the forward writes X_l, the chain writes dy_l, the weight-gradient kernel reads both, and that last
// kernel runs at the HBM roof.  dW_l = X_{l-1}^T dy_l is a sum over ALL evaluations, so its
// accumulators (256 KB of fp32 per layer) can only stay on chip if one CU sees a layer's operands for the whole launch.
// Here a CU owns ONE Dense layer: its four waves keep dW_l (64 x 256 each, 256 accumulator registers) AND W_l^T (bf16
// A-fragments, 128 registers) in their register files for the whole launch, and every 32-evaluation tile passes
// through a pipeline of 8 such CUs (Dense_8 ... Dense_1):
//
//     dy_l tile (from the CU of layer l+1)  --+--> dy_{l-1} = (dy_l W_l^T) o relu'(X_{l-1}) --> to the CU of layer l-1
//     X_{l-1} tile (forward save, HBM)      --+--> dW_l += X_{l-1}^T dy_l,  db_l += sum dy_l      (stays in registers)
//
// so dy is written once and read once right behind its producer, X is read once, and the separate weight-gradient pass
// over 8 of the 12 layers (90 % of its bytes) is gone.  The head (Dense_11, Dense_10, Dense_9 -> dz) runs first as a
// truncated chain launch (nerf_bwd_head_kernel); the five small problems (z x dy10m, x_emb x dy0, x_emb x dy5,
// d_emb x dy10m, h10 x dy11) keep the split-K kernel of nerf_mlp.hip, fed by the dumps this pipeline leaves behind.
//
// Pipeline p of a model takes the tiles p, p + P, ...; grid = P pipelines x 8 stages, one 4-wave workgroup per CU
// (P = CUs / 8).  blockIdx = stage * P + pipeline, so a workgroup's producer always has a LOWER block index: blocks are
// dispatched in order, the dumps are full-size (a producer never waits for its consumer), hence every wait is for a
// workgroup that was dispatched earlier and the launch completes even if fewer than all workgroups are resident.
//
// Hand-off (MI355X_MICROARCH.md, visibility, first table row): payload = sc1 (write-through) 16-byte stores, every
// storing wave drains them behind a counted vmcnt, workgroup barrier, ONE lane publishes the number of finished tiles
// with an sc1 flag store; the consumer polls that word (an sc1 LDS-DMA load issued a whole tile ahead, so the poll is
// never waited for in steady state), passes a barrier, and only then issues the sc1 loads (LDS-DMA) of the tile.
// Each dump line is written once and read once per launch, so no L1 copy of it can pre-exist on the reading CU.
// All steady-state vector-memory operations are inline asm (the compiler would drain vmcnt(0) at every LDS read while
// an LDS-DMA is pending) and counted by hand: per tile and wave 8 DMA loads + 4 stores (+ poll and flag on wave 0).
#include "nerf_chain.h"

namespace lnrf {

constexpr int kLsWaves = 4;
constexpr int kLsThreads = kLsWaves * 64;
constexpr int kLsStages = 8;                        // Dense_8 ... Dense_1
constexpr int kLsLag = 2;                           // tiles between issuing a vector-memory op and relying on it
constexpr int kLsDist = kLsLag + 1;                 // prefetch distance (tiles)
constexpr int kLsBufs = kLsDist + 1;                // LDS tile buffers
constexpr int kLsTileBytes = 32 * kFragBytes;       // X_{l-1} (16 fragments) then dy_l (16 fragments)
constexpr int kLsPollOff = kLsBufs * kLsTileBytes;  // poll landing words: 4 rotating slots + 1 for the blocking wait, 256 bytes each
constexpr int kLsLds = kLsPollOff + 5 * 256;
constexpr unsigned kLsMaxSpins = 1u << 18;          // bounded wait (each spin >= ~1 us): give up after a fraction of a second
constexpr int kLsSlabWaveFloats = 16 * kSlabTileFloats + 2 * 64;  // 16 dW tiles + 2 bias rows per wave
constexpr int64_t kLsSlabBlockBytes = (int64_t)kLsWaves * kLsSlabWaveFloats * (int64_t)sizeof(float);
constexpr int kLsCounterStride = 32;                // one 128-byte line per (pipeline, stage) counter

struct LsJob {
  const char* packed;   // nerf_pack_weights blob of this model
  const char* save;     // forward save buffer
  char* gdump;          // gradient dump (dy8 already written by the head kernel)
  unsigned* counters;   // [pipelines][8 stages][32] finished-tile counts, zeroed before the launch; [.. + 0] status word at the end
  float* slabs;         // [pipelines][8 stages] kLsSlabBlockBytes
  unsigned* status;     // != 0: a bounded wait gave up
  int64_t n_tiles;
  int pipelines;
};
struct LsArgs {
  LsJob job[2];
  int n_jobs, total_pipelines;
};

// In-kernel timeline (debug builds with -DLNRF_TIMELINE only, tools/ls_timeline_probe.py): the four waves of pipeline 0's
// stages 0, 3 and 7 stamp s_memtime at 8 points of iterations kLsTlIter0 .. + 15; compiled out of the product library.
#ifdef LNRF_TIMELINE
__device__ unsigned long long* g_ls_timeline_buf = nullptr;  // [3 stages][4 waves][16 iterations][8 stamps]
constexpr int kLsTlIter0 = 200;
#define LS_STAMP(k)                                                                                          \
  do {                                                                                                       \
    if (tl_on && i >= kLsTlIter0 && i < kLsTlIter0 + 16) {                                                   \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                            \
      if (lane == 0) tl_buf[((tl_sel * 4 + wave) * 16 + (i - kLsTlIter0)) * 8 + (k)] = t_;                     \
    }                                                                                                        \
  } while (0)
#else
#define LS_STAMP(k) ((void)0)
#endif

// ---- inline-asm vector memory (hand-counted, see the header) ------------------------------------------------------
// 1 KiB LDS-DMA: lane i fetches 16 bytes at base + voff + IMM and they land at lds_dst + IMM + 16 i (the instruction
// offset applies to the global AND the LDS address)
template <int IMM, bool SC1>
__device__ __forceinline__ void ls_dma16(const char* base, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  if constexpr (SC1)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%c4 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst), "i"(IMM) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%c4 nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst), "i"(IMM) : "memory");
}
// poll: every lane fetches the same counter word (4 bytes), landing at lds_dst + 4 i
__device__ __forceinline__ void ls_poll(const unsigned* word, unsigned lds_dst) {
  unsigned keep;
  unsigned zero = 0u;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(zero), "s"(word), "s"(lds_dst) : "memory");
}
template <int IMM, bool SC1>
__device__ __forceinline__ void ls_store16(char* base, unsigned voff, u32x4 v) {
  // The s_nop covers the hazard the compiler handles for its own stores and cannot see here: a store of more than 64 bits
  // reads its data registers for a few cycles after issue, and the next instruction may be a VALU write of one of them.
  if constexpr (SC1) asm volatile("global_store_dwordx4 %0, %1, %2 offset:%c3 sc1\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(base), "i"(IMM) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, %2 offset:%c3\n\ts_nop 1" : : "v"(voff), "v"(v), "s"(base), "i"(IMM) : "memory");
}
__device__ __forceinline__ void ls_flag_store(unsigned* word, unsigned value) {
  unsigned zero = 0u;
  asm volatile("global_store_dword %0, %1, %2 sc1" : : "v"(zero), "v"(value), "s"(word) : "memory");
}
template <int N>
__device__ __forceinline__ void ls_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%c0)" : : "i"(N) : "memory");
}
__device__ __forceinline__ void ls_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// LDS word (never a generic pointer: a flat load would count on vmcnt)
__device__ __forceinline__ unsigned lds_u32(int off) {
  return *reinterpret_cast<volatile __attribute__((address_space(3))) unsigned*>(
      (__attribute__((address_space(3))) char*)smem + off);
}

// dy pair = bf16 pair `pk`, each half kept where the matching half of the activation dword `x` is non-zero (ReLU output
// > 0) and zeroed elsewhere: min(x, 1) is 0 / 1 per half, times the bf16 bits.  Two packed VALU operations.
__device__ __forceinline__ unsigned relu_gate_pair(unsigned pk, unsigned x) {
  unsigned t, r;
  const unsigned ones = 0x00010001u;
  asm("v_pk_min_u16 %0, %1, %2" : "=v"(t) : "v"(x), "v"(ones));
  asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(t), "v"(pk));
  return r;
}

// The input-gradient MFMAs are inline asm so that their accumulators are VGPR tuples: the 256 accumulator registers hold
// dW, and the compiler would otherwise swap two dW tiles between the files around every tile (64 copies) and read the
// results back one register at a time.  Exactly-same-register accumulate chains need no software wait states; the first
// VALU read of a result is placed (scheduling fences) behind two later MFMAs, i.e. after the producing MFMA has left the
// pipe, plus an explicit s_nop.
// bias column sum: acc += the two bf16 halves of `pair` (dot product with (1, 1)); asm volatile keeps it between the MFMAs
// where it is written (the compiler sinks the builtin below the tile barrier, where nothing overlaps it)
__device__ __forceinline__ void dot2_ones(float& acc, unsigned pair) {
  const unsigned ones = 0x3f803f80u;
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(pair), "v"(ones));
}
__device__ __forceinline__ void mfma_vgpr_first(f32x16& d, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_vgpr(f32x16& d, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

__global__ __launch_bounds__(kLsThreads) void nerf_bwd_ls_kernel(LsArgs args) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int stage = (int)blockIdx.x / args.total_pipelines;
  const int gp = (int)blockIdx.x % args.total_pipelines;
  const bool second = args.n_jobs > 1 && gp >= args.job[0].pipelines;
  const LsJob job = second ? args.job[1] : args.job[0];
  const int pipe = second ? gp - args.job[0].pipelines : gp;
  const int l = 8 - stage;  // Dense layer of this CU

  const int64_t n_tiles = job.n_tiles;
  const int n_my = pipe < n_tiles ? (int)((n_tiles - pipe + job.pipelines - 1) / job.pipelines) : 0;
  unsigned* my_count = job.counters + ((int64_t)pipe * kLsStages + stage) * kLsCounterStride;
  const unsigned* up_count = job.counters + ((int64_t)pipe * kLsStages + (stage > 0 ? stage - 1 : 0)) * kLsCounterStride;

  // ---- per-launch state in registers: W_l^T fragments of this wave's 64 input rows, zero dW accumulators -----------
  // transposed stream (nerf_layout.h): chain layer t = 10 - l, fragment (in-tile o, k-step ks) at bwd_base(t) + 16 o + ks
  bf16x8 wt[2][16];
  {
    const char* wsrc = job.packed + kPackBwdOff + ((int64_t)bwd_base(2) + (int64_t)(l <= 8 ? 8 - l : 0) * 128 + 32 * wave) * kFragBytes +
                       lane * 16;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks)
        wt[a][ks] = bits_to_frag(*reinterpret_cast<const uint4*>(wsrc + (a * 16 + ks) * kFragBytes));
  }
  f32x16 acc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = zero_acc();
  float bsum[2] = {0.0f, 0.0f};

  const int c = lane & 31, hh = lane >> 5;
  const int x_slot0 = kSaveH + (l - 1) * 16, y_slot0 = grad_dy_slot(l), o_slot0 = grad_dy_slot(l - 1);
  const unsigned lane_off0 = dump_lane_off(0, c, hh), lane_off1 = dump_lane_off(1, c, hh);
  const unsigned dma_voff = (unsigned)(lane * 16 + wave * 4 * kFragBytes);  // this wave moves fragments 4w .. 4w+3 of X and of dy
  // output fragments of this wave: slots o_slot0 + 4 wave + {0..3}; per-lane offset inside the tile's dump block
  const unsigned st_voff0 = (unsigned)((o_slot0 + 4 * wave) * kFragBytes) + lane_off0;
  const unsigned st_voff1 = (unsigned)((o_slot0 + 4 * wave) * kFragBytes) + lane_off1;

  auto tile_of = [&](int i) -> int64_t { return (int64_t)pipe + (int64_t)i * job.pipelines; };
  auto issue_tile = [&](int i) {  // DMA of my tile i (clamped: surplus issues re-load the last tile into an idle buffer)
    const int ii = i < n_my ? i : n_my - 1;
    const int64_t t = tile_of(ii);
    const unsigned dst = (unsigned)((i % kLsBufs) * kLsTileBytes + wave * 4 * kFragBytes);
    const char* xb = job.save + (t * kSaveTileSlots + x_slot0) * (int64_t)kFragBytes;
    const char* yb = job.gdump + (t * kGradTileSlots + y_slot0) * (int64_t)kFragBytes;
    ls_dma16<0 * kFragBytes, false>(xb, dma_voff, dst);
    ls_dma16<1 * kFragBytes, false>(xb, dma_voff, dst);
    ls_dma16<2 * kFragBytes, false>(xb, dma_voff, dst);
    ls_dma16<3 * kFragBytes, false>(xb, dma_voff, dst);
    ls_dma16<0 * kFragBytes, true>(yb, dma_voff, dst + 16 * kFragBytes);
    ls_dma16<1 * kFragBytes, true>(yb, dma_voff, dst + 16 * kFragBytes);
    ls_dma16<2 * kFragBytes, true>(yb, dma_voff, dst + 16 * kFragBytes);
    ls_dma16<3 * kFragBytes, true>(yb, dma_voff, dst + 16 * kFragBytes);
  };

  if (n_my == 0) {  // nothing to do (tiny launches): publish "0 tiles" is the initial state already; leave zero slabs
    float* __restrict__ mine = job.slabs + (((int64_t)pipe * kLsStages + stage) * kLsWaves + wave) * kLsSlabWaveFloats;
    for (int i = lane; i < kLsSlabWaveFloats; i += 64) mine[i] = 0.0f;
    return;
  }

#ifdef LNRF_TIMELINE
  unsigned long long* tl_buf = g_ls_timeline_buf;
  const int tl_sel = stage == 0 ? 0 : (stage == 3 ? 1 : 2);
  const bool tl_on = tl_buf != nullptr && gp == 0 && (stage == 0 || stage == 3 || stage == 7);
#endif
  // tiles of my input known to be complete (stage 0 reads the head kernel's dump: all there before the launch)
  int ready = stage == 0 ? n_my : 0;
  bool failed = false;
  // blocking wait until `need` tiles of my input are published (slow path: start-up, or the producer is the bottleneck)
  auto wait_ready = [&](int need) {
    unsigned spins = 0;
    while (ready < need && !failed) {
      if (wave == 0) ls_poll(up_count, kLsPollOff + 4 * 256);
      ls_wait_vm<0>();
      ls_barrier();
      ready = (int)lds_u32(kLsPollOff + 4 * 256);
      ls_barrier();  // everybody has read the word before the next poll overwrites it
      if (ready < need) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kLsMaxSpins) {
          failed = true;
          if (tid == 0) __hip_atomic_store(job.status, 0x100u + (unsigned)stage, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  };

  // ---- prologue: the first kLsDist tiles ------------------------------------------------------------------------
  for (int i = 0; i < kLsDist; ++i) {
    wait_ready(i + 1 < n_my ? i + 1 : n_my);
    issue_tile(i);
  }
  ls_wait_vm<0>();
  ls_barrier();

  // ---- steady state: one tile per iteration ---------------------------------------------------------------------
  // vector-memory operations of iteration i: [wave 0: poll] 8 DMA loads of tile i + kLsDist, 4 stores of dy_{l-1} of tile
  // i, [wave 0, after the barrier: flag store].  The wait at the end of iteration i leaves the operations of
  // the last kLsLag iterations in flight.
  for (int i = 0; i < n_my; ++i) {
    const int buf = i % kLsBufs;
    const char* xbuf = smem + buf * kLsTileBytes;
    const char* ybuf = xbuf + 16 * kFragBytes;

    LS_STAMP(0);
#ifdef LS_STRESS  // hand-off test under uneven load (tools/ls_variants.sh): some stages stall for several tile times now and then
    if ((stage == 2 || stage == 5) && (i % 11) == stage) {
      for (int z = 0; z < (stage == 2 ? 4 : 9); ++z) __builtin_amdgcn_s_sleep(127);
    }
    if (stage == 6 && (i & 1)) __builtin_amdgcn_s_sleep(40);
#endif
    {
      const int want = i + kLsDist < n_my ? i + kLsDist + 1 : n_my;
      if (ready < want) wait_ready(want);  // uniform: `ready` comes from LDS words every wave reads after a barrier
    }
    // DMA of tile i + kLsDist: issued in 8 pieces between the MFMAs of phase A
    const int nxt = i + kLsDist < n_my ? i + kLsDist : n_my - 1;
    const int64_t tn = tile_of(nxt);
    const unsigned ndst = (unsigned)(((i + kLsDist) % kLsBufs) * kLsTileBytes + wave * 4 * kFragBytes);
    const char* nxb = job.save + (tn * kSaveTileSlots + x_slot0) * (int64_t)kFragBytes;
    const char* nyb = job.gdump + (tn * kGradTileSlots + y_slot0) * (int64_t)kFragBytes;
    auto dy_frag = [&](auto k_) -> bf16x8 {  // B operand of k-step ks as the producer stored it
      constexpr int ks = decltype(k_)::value;
      return bits_to_frag(*reinterpret_cast<const uint4*>(ybuf + ks * kFragBytes + ((ks & 1) ? lane_off1 : lane_off0)));
    };

    auto y_tr = [&](auto e_) -> bf16x8 {  // transposed dy operand of phase-B step e = 8 q + pos (rotated out tile, see below)
      constexpr int e = decltype(e_)::value;
      const int bb = ((e & 7) + 2 * wave) & 7;
      return tr_frag(ybuf + 2 * bb * kFragBytes, lane, 0, e >> 3);
    };
    auto x_mask = [&](auto f_) -> uint4 {  // saved X_{l-1} fragment f of this wave's four = the ReLU mask of output fragment f
      constexpr int f = decltype(f_)::value;
      return *reinterpret_cast<const uint4*>(xbuf + (4 * wave + f) * kFragBytes + ((f & 1) ? lane_off1 : lane_off0));
    };
    // poll result of iteration i - kLsLag - 1: it landed before the barrier that closed iteration i - 1 (wave 0's counted
    // wait), so every wave may read it now; it is looked at after this iteration's barrier (no LDS latency on that path)
    const unsigned seen_word = i > kLsLag ? lds_u32(kLsPollOff + ((i - kLsLag - 1) & 3) * 256) : 0u;

    LS_STAMP(1);
    // ---- phase A: input gradient dh[in rows of this wave][32 evaluations] = sum_k W^T[in][k] dy_l[k][eval], 32 MFMAs, with
    //      the B fragments read 3 k-steps ahead, the poll + 8 DMA pieces of the next tile issued between the MFMAs, and the
    //      first operands of phase B read during the last k-steps ----
    f32x16 d0, d1;
    bf16x8 bq[3];
    bf16x8 a0, a1, fq[3];
    uint4 xm;
    bq[0] = dy_frag(std::integral_constant<int, 0>{});
    bq[1] = dy_frag(std::integral_constant<int, 1>{});
    bq[2] = dy_frag(std::integral_constant<int, 2>{});
    static_for<16>([&](auto k_) {
      constexpr int ks = decltype(k_)::value;
      const bf16x8 b = bq[ks % 3];
      if constexpr (ks == 0) {
        mfma_vgpr_first(d0, wt[0][ks], b);
        mfma_vgpr_first(d1, wt[1][ks], b);
      } else {
        mfma_vgpr(d0, wt[0][ks], b);
        mfma_vgpr(d1, wt[1][ks], b);
      }
      if constexpr (ks + 3 < 16) bq[ks % 3] = dy_frag(std::integral_constant<int, ks + 3>{});
      if constexpr (ks == 0) {
        if (wave == 0) ls_poll(up_count, kLsPollOff + (i & 3) * 256);
      }
      if constexpr (ks >= 1 && ks <= 4) ls_dma16<(ks - 1) * kFragBytes, false>(nxb, dma_voff, ndst);
      if constexpr (ks >= 5 && ks <= 8) ls_dma16<(ks - 5) * kFragBytes, true>(nyb, dma_voff, ndst + 16 * kFragBytes);
      if constexpr (ks == 11) a0 = tr_frag(xbuf + (4 * wave) * kFragBytes, lane, 0, 0);
      if constexpr (ks == 12) a1 = tr_frag(xbuf + (4 * wave + 2) * kFragBytes, lane, 0, 0);
      if constexpr (ks == 13) fq[0] = y_tr(std::integral_constant<int, 0>{});
      if constexpr (ks == 14) fq[1] = y_tr(std::integral_constant<int, 1>{});
      if constexpr (ks == 15) {
        fq[2] = y_tr(std::integral_constant<int, 2>{});
        xm = x_mask(std::integral_constant<int, 0>{});
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    LS_STAMP(2);
    // ---- phase B: weight gradient dW[in rows of this wave][256 out] += X_{l-1}^T dy_l (32 MFMAs) with the epilogue of
    //      phase A between them: dy_{l-1} = bf16(dh) o relu'(X_{l-1}) — the saved activation fragment of the same slot is
    //      the mask — in 16 dword pieces, every finished fragment leaves by one sc1 store.  The wave walks the out tiles
    //      in the rotated order b = (pos + 2 wave) mod 8, so that "its" two bias columns (out tiles 2 wave, 2 wave + 1)
    //      are always positions 0 and 1 and no branch depends on the wave; acc[a][pos] is out tile b. ----
    char* ob = job.gdump + tile_of(i) * kGradTileSlots * (int64_t)kFragBytes;
    bf16x8 a0n, a1n;  // operands of the second half (evaluations 16..31), read a few steps ahead
    u32x4 ov;
    static_for<16>([&](auto e_) {
      constexpr int e = decltype(e_)::value;
      constexpr int pos = e & 7;
      const bf16x8 bf = fq[e % 3];
      acc[0][pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bf, acc[0][pos], 0, 0, 0);
      acc[1][pos] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bf, acc[1][pos], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // nothing of the epilogue below may move above these two MFMAs
      if constexpr (e == 0) asm volatile("s_nop 7" ::: "memory");
      if constexpr (e == 3) a0n = tr_frag(xbuf + (4 * wave) * kFragBytes, lane, 0, 1);
      if constexpr (e == 4) a1n = tr_frag(xbuf + (4 * wave + 2) * kFragBytes, lane, 0, 1);
      if constexpr (e == 7) {
        LS_STAMP(3);
        a0 = a0n;
        a1 = a1n;
      }
      if constexpr (pos < 2) {  // bias column sums: 8 bf16 values of the dy operand through the bf16 dot product with ones
        const uint4 bw = frag_to_bits(bf);
        dot2_ones(bsum[pos], bw.x);
        dot2_ones(bsum[pos], bw.y);
        dot2_ones(bsum[pos], bw.z);
        dot2_ones(bsum[pos], bw.w);
      }
      if constexpr (e + 3 < 16) fq[e % 3] = y_tr(std::integral_constant<int, e + 3>{});
      {  // epilogue piece e: dword w of output fragment f = 2 a + s
        constexpr int f = e >> 2, w = e & 3, aa = f >> 1, sh = f & 1;
        float lo, hi;
        if constexpr (aa == 0) { lo = d0[8 * sh + 2 * w]; hi = d0[8 * sh + 2 * w + 1]; }
        else { lo = d1[8 * sh + 2 * w]; hi = d1[8 * sh + 2 * w + 1]; }
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bf16x2 pk = {(__bf16)lo, (__bf16)hi};
        const unsigned xw = w == 0 ? xm.x : (w == 1 ? xm.y : (w == 2 ? xm.z : xm.w));
        ov[w] = relu_gate_pair(__builtin_bit_cast(unsigned, pk), xw);
        if constexpr (w == 3) {
#ifdef LS_PLAIN_DY_STORES
          ls_store16<f * kFragBytes, false>(ob, sh ? st_voff1 : st_voff0, ov);
#else
          ls_store16<f * kFragBytes, true>(ob, sh ? st_voff1 : st_voff0, ov);
#endif
          if constexpr (f < 3) xm = x_mask(std::integral_constant<int, (f < 3 ? f + 1 : 3)>{});
        }
      }
#ifdef LS_DMA_IN_B
      if constexpr ((e & 1) == 0 && e < 8) ls_dma16<(e / 2) * kFragBytes, false>(nxb, dma_voff, ndst);
      if constexpr ((e & 1) == 0 && e >= 8) ls_dma16<((e - 8) / 2) * kFragBytes, true>(nyb, dma_voff, ndst + 16 * kFragBytes);
#endif
#ifndef LS_NO_SB_B
      __builtin_amdgcn_sched_barrier(0);
#endif
    });

    LS_STAMP(4);
    // (4) retire: everything issued before the last kLsLag iterations is complete -> tile i + 1 is in LDS, the stores of
    //     tile i - kLsLag have reached memory, the poll of iteration i - kLsLag has landed
    if (wave == 0) ls_wait_vm<kLsLag * 14 - 1>(); else ls_wait_vm<kLsLag * 12>();
    LS_STAMP(5);
    ls_barrier();
    LS_STAMP(6);
    if (i > kLsLag && stage != 0) ready = (int)seen_word > ready ? (int)seen_word : ready;
    // tile i - kLsLag is published (every iteration issues the store, so that the counts above hold from the start)
    if (wave == 0) ls_flag_store(my_count, (unsigned)(i >= kLsLag ? i - kLsLag + 1 : 0));
  }
  ls_wait_vm<0>();
  ls_barrier();
  if (tid == 0) ls_flag_store(my_count, (unsigned)n_my);

  // ---- epilogue: the accumulators leave as they stand ([wave][tile a * 8 + b][4][lane][4 f32]) + bias partial sums ----
  float* __restrict__ mine = job.slabs + (((int64_t)pipe * kLsStages + stage) * kLsWaves + wave) * kLsSlabWaveFloats;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      float4* dst = reinterpret_cast<float4*>(mine + (a * 8 + b) * kSlabTileFloats) + lane;
#pragma unroll
      for (int v = 0; v < 4; ++v)
        dst[64 * v] = make_float4(acc[a][b][4 * v], acc[a][b][4 * v + 1], acc[a][b][4 * v + 2], acc[a][b][4 * v + 3]);
    }
  mine[16 * kSlabTileFloats + lane] = bsum[0];
  mine[16 * kSlabTileFloats + 64 + lane] = bsum[1];
}

// Folds the slabs of one model: blockIdx.x = stage * 64 + wave * 16 + tile; the four waves of the workgroup sum the
// pipelines q, q + 4, ... in order, meet in LDS and wave 0 adds the total to the gradient vector (fixed order, one owner
// per parameter: bit-reproducible).
template <int NW>  // waves of the folding workgroup; lds: (NW - 1) * 17 * 64 floats
__device__ __forceinline__ void ls_fold_block(int bid, const float* __restrict__ slabs, int pipelines,
                                              float* __restrict__ grads, float* lds) {
  const int stage = bid >> 6, w = (bid >> 4) & 3, j = bid & 15;
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int l = 8 - stage, a = j >> 3, pos = j & 7;
  const int b = (pos + 2 * w) & 7;       // out tile held at position `pos` of wave w (rotated walk of the pipeline kernel)
  const bool bias = a == 0 && pos < 2;   // bias row `pos` of wave w is the column sum of out tile b
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float bs = 0.0f;
  for (int p = q; p < pipelines; p += NW) {
    const float* base = slabs + (((int64_t)p * kLsStages + stage) * kLsWaves + w) * kLsSlabWaveFloats;
    const float4* tp = reinterpret_cast<const float4*>(base + j * kSlabTileFloats) + lane;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float4 t = tp[64 * v];
      acc[4 * v] += t.x; acc[4 * v + 1] += t.y; acc[4 * v + 2] += t.z; acc[4 * v + 3] += t.w;
    }
    if (bias) bs += base[16 * kSlabTileFloats + pos * 64 + lane];
  }
  if (q > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lds[((q - 1) * 17 + r) * 64 + lane] = acc[r];
    lds[((q - 1) * 17 + 16) * 64 + lane] = bs;
  }
  __syncthreads();
  if (q > 0) return;
#pragma unroll
  for (int p = 0; p < NW - 1; ++p) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += lds[(p * 17 + r) * 64 + lane];
    bs += lds[(p * 17 + 16) * 64 + lane];
  }
  const int colr = lane & 31, hh = lane >> 5;
  const int out = 32 * b + colr;
  if (bias) {
    bs += __shfl_xor(bs, 32, 64);
    if (hh == 0) grads[dense_b_off(l) + out] += bs;
  }
  const int it = 2 * w + a;
#pragma unroll
  for (int qq = 0; qq < 16; ++qq) {
    const int r = (qq & 3) + 8 * (qq >> 2) + 4 * hh;  // row of the 32-feature in-tile
    grads[dense_w_off(l) + (int64_t)(32 * it + r) * 256 + out] += acc[qq];
  }
}
__global__ __launch_bounds__(256) void nerf_ls_reduce_kernel(const float* __restrict__ slabs, int pipelines,
                                                             float* __restrict__ grads) {
  __shared__ float lds[3 * 17 * 64];
  ls_fold_block<4>(blockIdx.x, slabs, pipelines, grads, lds);
}

// Every fold of a layer-stationary backward in ONE launch: blockIdx.y = model; blockIdx.x first walks the slabs of the
// model's small problems (nerf_wgrad_reduce_kernel's work: (problem, wave, tile)), then the pipeline slabs
// (nerf_ls_reduce_kernel's work).  The two parts own disjoint parameters (Dense_5: rows 256.. here, rows 0..255 there).
struct LsFoldArgs {
  WgradArgs w[2];
  const float* small_slabs[2];
  const float* ls_slabs[2];
  float* grads[2];
  int pipelines[2];
};
__global__ __launch_bounds__(64 * kSlabReduceWaves) void nerf_ls_fold_kernel(LsFoldArgs a) {
  __shared__ float lds[(kSlabReduceWaves - 1) * 17 * 64];
  const int k = blockIdx.y;
  const int n_small = a.w[k].n_problems * kWaves * kSlabMaxTiles;
  const int bid = blockIdx.x;
  if (bid < n_small) {
    const int prob = bid / (kWaves * kSlabMaxTiles), w = bid / kSlabMaxTiles % kWaves, j = bid % kSlabMaxTiles;
    const WgradProblem pb = a.w[k].p[prob];
    switch (pb.shape) {  // the shapes ls_backward() launches
      case 7: wgrad_reduce_tile<4, 32, 2, 4, NerfWgradEpi>(pb, w, j, a.small_slabs[k], a.grads[k], lds); break;
      case 6: wgrad_reduce_tile<18, 10, 4, 2, NerfWgradEpi>(pb, w, j, a.small_slabs[k], a.grads[k], lds); break;
      default: wgrad_reduce_tile<8, 2, 4, 2, NerfWgradEpi>(pb, w, j, a.small_slabs[k], a.grads[k], lds); break;
    }
  } else if (bid - n_small < kLsStages * 64) {
    ls_fold_block<kSlabReduceWaves>(bid - n_small, a.ls_slabs[k], a.pipelines[k], a.grads[k], lds);
  }
}

// ---- head: the chain truncated after dz (flags 0 and 1 of nerf_chain.h) -----------------------------------------
struct BwdHeadSeq {
  static constexpr int count = bwd_cons_base(2);  // 76 fragments: Dense_11^T and [Dense_10 | Dense_9]^T
  static constexpr int at(int c) { return bwd_seq(c); }
};
constexpr int kBwdHeadStages = bwd_base(2) / kStageFrags;  // 6

__global__ __launch_bounds__(kThreads) void nerf_bwd_head_kernel(
    const char* __restrict__ packed, const char* __restrict__ save, const float* __restrict__ density,
    const float* __restrict__ rgb, const float* __restrict__ g_density, const float* __restrict__ g_rgb,
    int64_t M, int64_t n_tiles, char* __restrict__ gdump, unsigned* __restrict__ zero_words, int n_zero_words) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = (int64_t)blockIdx.x * kWaves + wave;
  // the pipeline launch behind this one starts from zero hand-off counters: cleared here instead of by a memset launch
  for (int i = blockIdx.x * kThreads + tid; i < n_zero_words; i += gridDim.x * kThreads) zero_words[i] = 0u;
  Ring<kBwdHeadStages, BwdHeadSeq> ring;
  ring.stream = packed + kPackBwdOff;
  ring.wave = wave;
  ring.lane = lane;
  GlobalDumpSink sink{DumpAddr{gdump, n_tiles, tile, lane & 31, lane >> 5, kGradTileSlots}};
  bwd_chain_tile<GlobalDumpSink, decltype(ring), true>(ring, sink, save, n_tiles, density, rgb, g_density, g_rgb, M, tile, lane);
}

}  // namespace lnrf

using namespace lnrf;

#ifdef LNRF_TIMELINE
// debug library only (tools/build_timeline.sh): where the stamped workgroups write their s_memtime values
extern "C" int lnrf_debug_set_ls_timeline(void* buf) {
  hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(lnrf::g_ls_timeline_buf), &buf, sizeof(buf));
  return e == hipSuccess ? LNRF_OK : hip_fail(e, "hipMemcpyToSymbol(g_ls_timeline_buf)");
}
#endif

static int ls_pipelines_for_device(int* out) {
  int dev = 0, cus = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return hip_fail(e, "hipDeviceGetAttribute(multiprocessor count)");
  *out = cus / kLsStages;
  return LNRF_OK;
}
constexpr int kLsMaxPipelines = 64;
static int64_t ls_dump_bytes(int64_t m) { return (int64_t)kGradSlots * nerf_tiles_for(m) * kFragBytes; }
// workgroups of the three small problems (x_emb x [dy0 | dy5], [z | d_emb] x dy10m, h10 x dy11), in proportion to the bytes
// they stream per tile (36, 28, 10 KiB)
static int ls_small_blocks(int i) {
  static const int total = exp_env_int("LNRF_LS_SMALL_BLOCKS", 256);  // experiment builds only (common.h)
  const int share[3] = {36, 28, 10};
  return total * share[i] / 74;
}
static int64_t ls_small_slab_bytes() { return 512 * kSlabBlockBytes; }
static int64_t ls_counter_bytes() { return ((int64_t)kLsMaxPipelines * kLsStages * kLsCounterStride + 64) * (int64_t)sizeof(unsigned); }

extern "C" int64_t lnrf_nerf_bwd_ls_scratch_bytes(const lnrf_nerf_shape* s, int64_t m) {
  if (!nerf_shape_fused(s)) return -1;
  return lnrf::ls_scratch_bytes(m);
}

int64_t lnrf::ls_scratch_bytes(int64_t m) {
  return ls_dump_bytes(m) + ls_small_slab_bytes() + ls_counter_bytes() + (int64_t)kLsMaxPipelines * kLsStages * kLsSlabBlockBytes;
}
int64_t lnrf::ls_small_slab_off(int64_t m) { return ls_dump_bytes(m); }

int lnrf::launch_ls_pipeline(const void* packed, const void* save, void* scratch, int64_t m, float* grads, hipStream_t st) {
  int total = 0;
  int rc = ls_pipelines_for_device(&total);
  if (rc) return rc;
  if (total > kLsMaxPipelines) total = kLsMaxPipelines;
  if (total < 1) {
    set_error("layer-stationary backward: the device has fewer than 8 CUs");
    return LNRF_ERR_UNSUPPORTED;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_bwd_ls_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLsLds);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(max dynamic LDS)");
  const int64_t n_tiles = nerf_tiles_for(m);
  char* sc = (char*)scratch;
  unsigned* counters = reinterpret_cast<unsigned*>(sc + ls_dump_bytes(m) + ls_small_slab_bytes());
  float* slabs = reinterpret_cast<float*>(sc + ls_dump_bytes(m) + ls_small_slab_bytes() + ls_counter_bytes());
  e = hipMemsetAsync(counters, 0, ls_counter_bytes(), st);
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(counters)");
  LsArgs a;
  a.n_jobs = 1;
  a.total_pipelines = total;
  LsJob& j = a.job[0];
  j.packed = (const char*)packed;
  j.save = (const char*)save;
  j.gdump = sc;
  j.counters = counters;
  j.slabs = slabs;
  j.status = counters + (int64_t)kLsMaxPipelines * kLsStages * kLsCounterStride;
  j.n_tiles = n_tiles;
  j.pipelines = total;
  a.job[1] = a.job[0];
  hipLaunchKernelGGL(nerf_bwd_ls_kernel, dim3((unsigned)(total * kLsStages)), dim3(kLsThreads), kLsLds, st, a);
  LNRF_LAUNCH_CHECK();
  hipLaunchKernelGGL(nerf_ls_reduce_kernel, dim3(kLsStages * 64), dim3(256), 0, st, (const float*)slabs, total, grads);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}

extern "C" int64_t lnrf_nerf_bwd_ls_status_offset(const lnrf_nerf_shape* s, int64_t m) {
  if (!nerf_shape_fused(s)) return -1;
  return ls_dump_bytes(m) + ls_small_slab_bytes() + (int64_t)kLsMaxPipelines * kLsStages * kLsCounterStride * (int64_t)sizeof(unsigned);
}

namespace {
struct LsModel {
  const void* packed;
  const void* save;
  const float *density, *rgb, *g_density, *g_rgb;
  int64_t m;
  void* scratch;
  float* grads;
};
}  // namespace

// head launch + pipeline launch (one or two models) + small problems + reduce launches
static int ls_backward(const LsModel* mdl, int n_models, int phases, hipStream_t st) {
  int total = 0;
  int rc = ls_pipelines_for_device(&total);
  if (rc) return rc;
  if (total > kLsMaxPipelines) total = kLsMaxPipelines;
  if (total < n_models) {
    set_error("lnrf_nerf_mlp_bwd_ls: the device has too few CUs for a pipeline per model");
    return LNRF_ERR_UNSUPPORTED;
  }
  // pipelines proportional to the evaluations of each model
  int pipes[2] = {total, 0};
  if (n_models == 2) {
    const double share = (double)mdl[0].m / (double)(mdl[0].m + mdl[1].m);
    pipes[0] = (int)(total * share + 0.5);
    if (pipes[0] < 1) pipes[0] = 1;
    if (pipes[0] > total - 1) pipes[0] = total - 1;
    pipes[1] = total - pipes[0];
  }
  LsArgs a;
  a.n_jobs = n_models;
  a.total_pipelines = total;
  rc = [&]() -> int {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_bwd_head_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kRingBytes + round_up(kBiasFloats * 4, 1024));
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_bwd_ls_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLsLds);
    return e == hipSuccess ? LNRF_OK : hip_fail(e, "hipFuncSetAttribute(max dynamic LDS)");
  }();
  if (rc) return rc;
  for (int k = 0; k < n_models; ++k) {
    const LsModel& md = mdl[k];
    const int64_t n_tiles = nerf_tiles_for(md.m);
    char* sc = (char*)md.scratch;
    unsigned* counters = reinterpret_cast<unsigned*>(sc + ls_dump_bytes(md.m) + ls_small_slab_bytes());
    float* slabs = reinterpret_cast<float*>(sc + ls_dump_bytes(md.m) + ls_small_slab_bytes() + ls_counter_bytes());
    if (phases & 1) {
      hipLaunchKernelGGL(nerf_bwd_head_kernel, dim3((unsigned)(n_tiles / kWaves)), dim3(kThreads),
                         kRingBytes + round_up(kBiasFloats * 4, 1024), st, (const char*)md.packed, (const char*)md.save,
                         md.density, md.rgb, md.g_density, md.g_rgb, md.m, n_tiles, sc, counters,
                         (phases & 2) ? (int)(ls_counter_bytes() / sizeof(unsigned)) : 0);
      LNRF_LAUNCH_CHECK();
    }
    if ((phases & 3) == 2) {  // pipeline phase on its own (bench sections): nobody cleared the counters yet
      hipError_t e = hipMemsetAsync(counters, 0, ls_counter_bytes(), st);
      if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(counters)");
    }
    LsJob& j = a.job[k];
    j.packed = (const char*)md.packed;
    j.save = (const char*)md.save;
    j.gdump = sc;
    j.counters = counters;
    j.slabs = slabs;
    j.status = counters + (int64_t)kLsMaxPipelines * kLsStages * kLsCounterStride;
    j.n_tiles = n_tiles;
    j.pipelines = pipes[k];
  }
  if (n_models == 1) a.job[1] = a.job[0];
  if (phases & 2) {
    hipLaunchKernelGGL(nerf_bwd_ls_kernel, dim3((unsigned)(total * kLsStages)), dim3(kLsThreads), kLsLds, st, a);
    LNRF_LAUNCH_CHECK();
  }
  LsFoldArgs fold;
  for (int k = 0; k < n_models && (phases & 4); ++k) {
    const LsModel& md = mdl[k];
    const int64_t n_tiles = nerf_tiles_for(md.m);
    // the five problems that are not a pipeline stage, fed by the dumps the two launches above left behind
    WgradArgs w;
    w.n_problems = 0;
    int first = 0;
    const int64_t cap = (n_tiles + 5) / 6;
    auto add = [&](int nb_want, int shape, int xs, int ys, int dense, int row_map, int row_off, int col_map, int do_bias) {
      WgradProblem p;
      p.shape = shape; p.x_slot0 = xs; p.y_slot0 = ys; p.dense = dense; p.row_map = row_map; p.row_off = row_off;
      p.col_map = col_map; p.do_bias = do_bias;
      p.w_off = p.b_off = p.out_dim = p.n_rows = 0;
      int64_t nb = nb_want;
      if (nb > cap) nb = cap;
      if (nb < 1) nb = 1;
      p.first_block = first;
      p.n_blocks = (int)nb;
      first += (int)nb;
      w.p[w.n_problems++] = p;
    };
    // workgroups in proportion to the bytes a problem streams per tile (36, 28, 10 KiB): 256 = one per CU (384 = one and a half rounds is 10 % slower, 512 equal within the box-to-box spread).
    // [z | d_emb] and [dy0 | dy5] are neighbours in the save / dump layouts (nerf_layout.h), so dy10m and x_emb are read once.
    add(ls_small_blocks(0), 7, kSaveXin, grad_dy_slot(0), 0, ROW_XEMB, 0, COL_DY0_DY5, 1);   // Dense_0 and Dense_5 rows 256..315
    add(ls_small_blocks(1), 6, kSaveZ, kGradDy10m, 10, ROW_Z_DEMB, 0, COL_DY10M, 1);         // Dense_10 (all 280 rows) and Dense_9
    add(ls_small_blocks(2), 4, kSaveH10, kGradDy11, 11, ROW_HIDDEN, 0, COL_DY11, 1);          // Dense_11
    float* small_slabs = reinterpret_cast<float*>((char*)md.scratch + ls_dump_bytes(md.m));
    // interleaved walk: at any moment the 256 workgroups read neighbouring tiles (one 80 MB window moving through the save
    // and the dump) instead of 256 places 1/256 of the buffers apart — measured 0.60 vs 0.68 ms for the finish phase
    rc = launch_nerf_wgrad(w, first, md.save, md.scratch, n_tiles, md.grads, st, WgLayout{kSaveTileSlots, kGradTileSlots, 1},
                           small_slabs, false, false);
    if (rc) return rc;
    fold.w[k] = w;
    fold.small_slabs[k] = small_slabs;
    fold.ls_slabs[k] = a.job[k].slabs;
    fold.grads[k] = md.grads;
    fold.pipelines[k] = pipes[k];
  }
  if (phases & 4) {  // one fold launch for everything both models left in slabs
    if (n_models == 1) {
      fold.w[1] = fold.w[0]; fold.small_slabs[1] = fold.small_slabs[0]; fold.ls_slabs[1] = fold.ls_slabs[0];
      fold.grads[1] = fold.grads[0]; fold.pipelines[1] = fold.pipelines[0];
    }
    const int gx = fold.w[0].n_problems * kWaves * kSlabMaxTiles + kLsStages * 64;  // both models have the same problems
    hipLaunchKernelGGL(nerf_ls_fold_kernel, dim3((unsigned)gx, (unsigned)n_models), dim3(64 * kSlabReduceWaves), 0, st, fold);
    LNRF_LAUNCH_CHECK();
  }
  return LNRF_OK;
}

extern "C" int lnrf_nerf_mlp_bwd_ls(const lnrf_nerf_shape* shape, const void* packed, const void* save,
                                    const float* density, const float* rgb, const float* g_density,
                                    const float* g_rgb, int64_t m, void* scratch, float* grads, int32_t phases,
                                    lnrf_stream_t stream) {
  if (!nerf_shape_fused(shape)) {
    set_error("lnrf_nerf_mlp_bwd_ls: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(packed && save && density && rgb && g_density && g_rgb && scratch && grads, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  LNRF_CHECK_ARG(phases >= 1 && phases <= 7, "phases: bit 0 head, bit 1 pipeline, bit 2 small problems + fold");
  const LsModel md{packed, save, density, rgb, g_density, g_rgb, m, scratch, grads};
  return ls_backward(&md, 1, phases, as_stream(stream));
}

extern "C" int lnrf_nerf_mlp_bwd_ls2(const lnrf_nerf_shape* shape, const void* packed_a, const void* save_a,
                                     const float* density_a, const float* rgb_a, const float* g_density_a,
                                     const float* g_rgb_a, int64_t m_a, void* scratch_a, float* grads_a,
                                     const void* packed_b, const void* save_b, const float* density_b,
                                     const float* rgb_b, const float* g_density_b, const float* g_rgb_b, int64_t m_b,
                                     void* scratch_b, float* grads_b, int32_t phases, lnrf_stream_t stream) {
  if (!nerf_shape_fused(shape)) {
    set_error("lnrf_nerf_mlp_bwd_ls2: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(packed_a && save_a && density_a && rgb_a && g_density_a && g_rgb_a && scratch_a && grads_a, "null pointer");
  LNRF_CHECK_ARG(packed_b && save_b && density_b && rgb_b && g_density_b && g_rgb_b && scratch_b && grads_b, "null pointer");
  LNRF_CHECK_ARG(m_a > 0 && m_b > 0, "both models need evaluations (use lnrf_nerf_mlp_bwd_ls for one)");
  const LsModel md[2] = {{packed_a, save_a, density_a, rgb_a, g_density_a, g_rgb_a, m_a, scratch_a, grads_a},
                         {packed_b, save_b, density_b, rgb_b, g_density_b, g_rgb_b, m_b, scratch_b, grads_b}};
  LNRF_CHECK_ARG(phases >= 1 && phases <= 7, "phases: bit 0 head, bit 1 pipeline, bit 2 small problems + fold");
  return ls_backward(md, 2, phases, as_stream(stream));
}
