// nerf_bwd_fused.hip — the NeRFModel backward (jax.grad through model.py:43-62) as ONE persistent launch.
//
// Why: with separate launches the chain kernel writes the pre-activation gradients dy_l (4.9 KiB per evaluation)
// to HBM and the weight-gradient kernel reads them back together with the saved activations X_l, which makes the
// weight gradients HBM-bound.  Here "chain" workgroups (producers) and "weight-gradient"
// workgroups (consumers) run side by side on the chip, one workgroup per CU, and every dy_l tile is handed over
// layer by layer through small ring buffers that stay in the 256 MiB Infinity Cache: dy never reaches HBM, the
// consumers stream only X from HBM, and their dW accumulators live in registers for the whole pass.
//
//   producer p      : groups g = p, p + P, ... (a group = 8 tiles = 256 evaluations, one tile per wave).  For chain
//                     layer ("flag") f = 0..9 it takes the next ticket s of flag f (device-scope counter), waits
//                     until ring buffer s % NB of that flag has been released by all its consumers, writes the
//                     layer's fragments there with write-through (sc1) stores and, one layer later (so that the
//                     stores have drained behind a counted vmcnt wait), every wave adds 1 to ready[f][s % NB].
//   consumer (p, j) : workgroup j of weight-gradient problem p (13 problems, nerf_chain.h) consumes the items
//                     s = j, j + n_p, ... of its flag IN TICKET ORDER: polls ready == 8 * (s / NB + 1), reads the
//                     group id, streams X (save buffer, HBM) and dy (ring, sc1 loads) through LDS into the MFMA
//                     body of fused_chain.h, then adds 1 to released[f][s % NB].  dW leaves by fp32 atomics at the end.
//
// MEASURED OUTCOME (MI355X, 786,432 evaluations, tools/fused_bwd_probe.py, profiles/r02_fused_bwd_experiment.txt):
// correct (gradients equal to the two-launch path to fp32-atomic noise) but SLOWER — 3.25 ms at best (30 % producers)
// against 2.6-2.9 ms for chain + weight-gradient launches.  The consumers alone stream at 21-22 GB/s per CU whether
// 128 or 230 of them run (registers holding one or two iterations of loads in flight make no difference): that is the
// per-CU miss-path rate (~10 B/clk/CU) at which the two-launch weight-gradient kernel already runs on ALL 256 CUs.  A
// role split leaves the operand stream to half the CUs, so it cannot win however the hand-off is done; the kernel is
// kept as an opt-in experiment (NeRFModel.backward_kernel = "fused") with its parity test.
//
// Hand-off protocol = recipe R1 of the CDNA4 guide (sc1 payload stores drained by every storing wave before its
// counter add; consumer: relaxed poll of the counter, workgroup barrier, sc1 loads of the payload).  Allocation and
// consumption of a flag's buffers both follow ticket order and a producer never waits while it holds an unwritten
// buffer of the flag it waits on, so the wait-for graph only points from flag f to flags > f: no deadlock as long as
// all workgroups are resident (grid <= CUs, checked against the occupancy API).  Every spin is bounded: on a timeout
// the waiter records a code in ctrl.status and all loops run to their fixed trip counts (results are then invalid
// and the host reports the status).
#include "nerf_chain.h"

namespace lnrf {

constexpr int kMaxNB = 128;                                       // ring buffers per flag (upper bound)
constexpr int kBufSlots = 16;                                     // dump slots per ring buffer
constexpr int kBufBytes = kBufSlots * kWaves * kFragBytes;        // 128 KiB: 16 slots x 8 tiles x 1 KiB
constexpr unsigned kMaxSpins = 1u << 21;                          // ~2-4 s of polling before giving up
constexpr int kFusedBwdLds = 2 * 2 * 32 * kFragBytes;             // 128 KiB: the largest consumer body
constexpr int kTicketLdsOff = kRingBytes + round_up(kBiasFloats * 4, 1024);  // producer: ticket broadcast words

typedef __attribute__((address_space(1))) unsigned gu32;
#define LNRF_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct FusedCtrl {  // zeroed by hipMemsetAsync before EVERY launch
  unsigned status[32];                         // [0] != 0: a bounded spin gave up (code); own cache line
  unsigned seq[kChainFlags][32];               // ticket counters, one 128-byte line each
  unsigned ready[kChainFlags][kMaxNB];         // arrivals of storing waves (8 per use)
  unsigned released[kChainFlags][kMaxNB];      // arrivals of consumers (ncons[f] per use)
  unsigned desc[kChainFlags][kMaxNB];          // group id of the item in the buffer
};
constexpr int64_t kCtrlBytes = (sizeof(FusedCtrl) + 4095) / 4096 * 4096;

struct FusedArgs {
  WgradArgs w;                 // consumer problems; first_block / n_blocks count CONSUMER workgroups
  int flag[kMaxProblems];      // chain flag that carries the problem's dy
  int y_local0[kMaxProblems];  // first dy slot inside that flag's buffer
  int ncons[kChainFlags];      // consumers (problems) per flag
  int producers;               // workgroups 0..producers-1 run the chain
  int nb;                      // ring buffers per flag in use (<= kMaxNB)
  int debug;                   // diagnostics (lnrf_nerf_bwd_fused_debug): 1 = consumers do not wait (items = groups in
                               // order, ring contents arbitrary), 2 = producers idle, 4 = producers neither wait nor publish
};

__device__ __forceinline__ gu32* g32(unsigned* p) { return (gu32*)p; }

// ONE lane polls *p until it reaches `target`; gives up when another waiter already failed or after kMaxSpins.
__device__ __forceinline__ bool wait_ge(gu32* p, unsigned target, gu32* status, unsigned code) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(p, LNRF_RLX_AGENT) >= target) return true;
    if ((spins & 255u) == 255u && __hip_atomic_load(status, LNRF_RLX_AGENT) != 0u) return false;
    if (spins > kMaxSpins) {
      __hip_atomic_store(status, code, LNRF_RLX_AGENT);
      return false;
    }
    __builtin_amdgcn_s_sleep(4);
  }
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ring_rsrc(char* base) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, kBufBytes, 0x00020000);
}

// ---------------------------------------------------------------------------------------------
// producer: chain workgroup
// ---------------------------------------------------------------------------------------------
struct RingSink {
  char* ring;        // [flag][buffer][slot 0..15][tile 0..7][1 KiB]
  FusedCtrl* ctrl;
  int ncons[kChainFlags];  // consumers per flag (wave-uniform copies of the kernel argument)
  int nb, wave, lane, c, hh;
  unsigned group;
  bool more_groups;  // this producer runs another group after the current one
  unsigned next_ticket;  // (wave 0, lane 0) ticket taken ahead of time for the next flag
  __amdgpu_buffer_rsrc_t cur;
  int cur_slot0;
  unsigned cur_b, prev_b;
  int lane_off[2];  // dump_lane_off for even / odd slots
  int debug;

  template <int F>
  __device__ __forceinline__ void begin_flag() {
    volatile unsigned* bc = reinterpret_cast<volatile unsigned*>(&smem[kTicketLdsOff]);
    if (wave == 0 && lane == 0) {
      const unsigned t = next_ticket;
      const unsigned b = t % (unsigned)nb;
      // the buffer's previous use (ticket t - nb) must have been consumed by every problem of this flag
      if (!(debug & 4)) wait_ge(g32(&ctrl->released[F][b]), (unsigned)ncons[F] * (t / (unsigned)nb), g32(&ctrl->status[0]), 0x100u + F);
      __hip_atomic_store(g32(&ctrl->desc[F][b]), group, LNRF_RLX_AGENT);
      bc[F] = b;
    }
    __syncthreads();
    cur_b = __builtin_amdgcn_readfirstlane(bc[F]);
    if (wave == 0 && lane == 0) {  // ticket for the flag after this one, in flight while this layer computes
      if (F + 1 < kChainFlags) next_ticket = __hip_atomic_fetch_add(g32(&ctrl->seq[F + 1][0]), 1u, LNRF_RLX_AGENT);
      else if (more_groups) next_ticket = __hip_atomic_fetch_add(g32(&ctrl->seq[0][0]), 1u, LNRF_RLX_AGENT);
    }
    // per-wave descriptor (tile = wave folded into the base): the remaining store offset is a compile-time constant,
    // so there is nothing loop-invariant for the compiler to hoist out of the group loop (156 live scalars otherwise)
    cur = ring_rsrc(ring + ((int64_t)F * nb + cur_b) * kBufBytes + wave * kFragBytes);
    cur_slot0 = flag_slot0(F);
  }
  __device__ __forceinline__ void store(int slot, const bf16x8& f) {
    // per-lane part (two variants, by slot parity) in the vector offset, slot / tile part in the scalar offset
    const int soff = (slot - cur_slot0) * kWaves * kFragBytes;
    __builtin_amdgcn_raw_buffer_store_b128(frag_to_bits_v(f), cur, lane_off[slot & 1], soff, 16);  // 16 = sc1
  }
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ u4v frag_to_bits_v(const bf16x8& f) { return __builtin_bit_cast(u4v, f); }

  template <int F>
  __device__ __forceinline__ void end_flag() {
    if constexpr (F >= 1) {
      // every layer issues 16 dump stores per wave after the previous layer's last one: once at most 16 vector
      // memory operations are outstanding, the previous flag's stores have reached memory (they retire in order)
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(g32(&ctrl->ready[F - 1][prev_b]), 1u, LNRF_RLX_AGENT);
    }
    prev_b = cur_b;
    if constexpr (F == kChainFlags - 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(g32(&ctrl->ready[F][cur_b]), 1u, LNRF_RLX_AGENT);
    }
  }
};

__device__ __forceinline__ void producer_role(RingSink& sink, int producers, const char* __restrict__ packed,
                                              const char* __restrict__ save, int64_t save_tiles,
                                              const float* __restrict__ density, const float* __restrict__ rgb,
                                              const float* __restrict__ g_density, const float* __restrict__ g_rgb,
                                              int64_t M, int64_t n_groups, char* ring, FusedCtrl* ctrl) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  sink.ring = ring;
  sink.ctrl = ctrl;
  sink.wave = wave;
  sink.lane = lane;
  sink.c = lane & 31;
  sink.hh = lane >> 5;
  sink.lane_off[0] = dump_lane_off(0, sink.c, sink.hh);
  sink.lane_off[1] = dump_lane_off(1, sink.c, sink.hh);
  sink.next_ticket = 0;
  sink.prev_b = 0;
  sink.cur_b = 0;
  const int64_t first = blockIdx.x;
  if (first < n_groups && wave == 0 && lane == 0)
    sink.next_ticket = __hip_atomic_fetch_add(g32(&ctrl->seq[0][0]), 1u, LNRF_RLX_AGENT);
  for (int64_t g = first; g < n_groups; g += producers) {
    sink.group = (unsigned)g;
    sink.more_groups = g + producers < n_groups;
    Ring<kBwdStages, BwdSeq> wring;
    // The weight stream does not depend on the group, so the compiler would hoist all 140 stage addresses of the
    // unrolled chain out of this loop (280 scalar registers live across it: spills).  An empty asm makes the base
    // opaque per iteration; the addresses are then formed next to their loads as in the one-tile kernel.
    const char* wstream = packed + kPackBwdOff;
    asm volatile("" : "+s"(wstream));
    wring.stream = wstream;
    wring.wave = wave;
    wring.lane = lane;
    bwd_chain_tile(wring, sink, save, save_tiles, density, rgb, g_density, g_rgb, M, g * kWaves + wave, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// consumer: weight-gradient workgroup (the MFMA body of fused_chain.h::wgrad_body on ring items)
// ---------------------------------------------------------------------------------------------
template <int NXF, int NYF, int WI, int WO, int SPI, class EPI>
__device__ __forceinline__ void wgrad_ring_body(const WgradProblem& pb, int flag, int y_local0, int nb, int debug, int split,
                                                const char* __restrict__ save, int64_t save_tiles, char* ring,
                                                FusedCtrl* ctrl, int64_t n_groups, float* __restrict__ grads) {
  constexpr int NI = NXF / 2, NO = NYF / 2;
  constexpr int TI = (NI + WI - 1) / WI, TO = (NO + WO - 1) / WO;
  constexpr bool FULL_I = TI * WI == NI, FULL_O = TO * WO == NO;
  constexpr int NF = NXF + NYF;
  constexpr int PER_WAVE = (NF + kWaves - 1) / kWaves;
  constexpr int STEP_BYTES = NF * kFragBytes;
  constexpr int ITER_BYTES = SPI * STEP_BYTES;
  constexpr int IPG = kWaves / SPI;  // iterations per group of 8 tiles
  static_assert(WI * WO == kWaves && kWaves % SPI == 0 && 2 * ITER_BYTES <= kFusedBwdLds, "consumer geometry");
  typedef unsigned u4v __attribute__((ext_vector_type(4)));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave / WO, wo = wave % WO;

  const int64_t n_items = split < n_groups ? (n_groups - split + pb.n_blocks - 1) / pb.n_blocks : 0;
  const int64_t iters = n_items * IPG;
  volatile unsigned* bc = reinterpret_cast<volatile unsigned*>(&smem[kFusedBwdLds]);  // 1 word past the stage buffers

  // the forward's save buffer in either layout (fused_chain.h dump_off): slot stride / tile stride
  const char* x_base = save + lane * 16;
  uint4 rr[2][SPI][PER_WAVE];  // two iterations of loads in flight: one iteration per memory latency is not enough
  unsigned cur_group = 0;
  __amdgpu_buffer_rsrc_t yrs = ring_rsrc(ring);

  auto item_buffer = [&](int64_t it) -> unsigned {  // ring buffer of the item that iteration `it` belongs to
    const int64_t s = split + (it / IPG) * pb.n_blocks;
    return (unsigned)(s % nb);
  };
  auto load = [&](auto par_, int64_t it) {
    constexpr int P = decltype(par_)::value;
    if (it >= iters) {
#pragma unroll
      for (int u = 0; u < SPI; ++u)
#pragma unroll
        for (int q = 0; q < PER_WAVE; ++q) rr[P][u][q] = make_uint4(0, 0, 0, 0);
      return;
    }
    const int sub = (int)(it % IPG);
    if (sub == 0) {  // a new item: wait until all 8 storing waves have published it, then learn its group
      const int64_t s = split + (it / IPG) * pb.n_blocks;
      const unsigned b = (unsigned)(s % nb);
      if (tid == 0) {
        if (debug & 1) {
          bc[0] = (unsigned)s;
        } else {
          wait_ge(g32(&ctrl->ready[flag][b]), (unsigned)kWaves * (unsigned)(s / nb + 1), g32(&ctrl->status[0]),
                  0x200u + flag);
          bc[0] = __hip_atomic_load(g32(&ctrl->desc[flag][b]), LNRF_RLX_AGENT);
        }
      }
      __syncthreads();  // the barrier between the poll and EVERY load of the handed-off bytes
      cur_group = __builtin_amdgcn_readfirstlane(bc[0]);
      if (cur_group >= (unsigned)n_groups) cur_group = (unsigned)n_groups - 1u;  // after a timeout: stay in bounds
      yrs = ring_rsrc(ring + ((int64_t)flag * nb + b) * kBufBytes);
    }
#pragma unroll
    for (int u = 0; u < SPI; ++u) {
      const int tg = sub * SPI + u;                          // tile inside the group
      const int64_t tile = (int64_t)cur_group * kWaves + tg;  // tile in the save buffer
#pragma unroll
      for (int q = 0; q < PER_WAVE; ++q) {
        int f = wave + kWaves * q;
        if constexpr (NF % kWaves != 0) f = f < NF ? f : NF - 1;
        if (f < NXF) {
          const u4v v = __builtin_nontemporal_load(
              reinterpret_cast<const u4v*>(x_base + dump_off(pb.x_slot0 + f, tile, save_tiles, kSaveTileSlots)));
          rr[P][u][q] = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
          const int off = ((y_local0 + f - NXF) * kWaves + tg) * kFragBytes + lane * 16;
          const u4v v = __builtin_amdgcn_raw_buffer_load_b128(yrs, off, 0, 16);  // sc1: served past this CU's L1
          rr[P][u][q] = make_uint4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  };
  auto write = [&](auto par_) {  // iteration of parity P: registers rr[P] -> LDS buffer P
    constexpr int P = decltype(par_)::value;
#pragma unroll
    for (int u = 0; u < SPI; ++u)
#pragma unroll
      for (int q = 0; q < PER_WAVE; ++q) {
        int f = wave + kWaves * q;
        if constexpr (NF % kWaves != 0) f = f < NF ? f : NF - 1;
        *reinterpret_cast<uint4*>(&smem[P * ITER_BYTES + u * STEP_BYTES + f * kFragBytes + lane * 16]) = rr[P][u][q];
      }
  };

  f32x16 acc[TI][TO];
#pragma unroll
  for (int a = 0; a < TI; ++a)
#pragma unroll
    for (int b = 0; b < TO; ++b) acc[a][b] = zero_acc();
  float bsum[TO];
#pragma unroll
  for (int b = 0; b < TO; ++b) bsum[b] = 0.0f;
  const int ypar = y_local0 & 1, xpar = pb.x_slot0 & 1;  // slot parity of even fragments (flag bases are even)

  auto compute = [&](int bufi) {
#pragma unroll
    for (int u = 0; u < SPI; ++u) {
      const char* buf = smem + bufi * ITER_BYTES + u * STEP_BYTES;
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
          const int itile = wi + WI * a;
          if (FULL_I || itile < NI) {
            const bf16x8 af = tr_frag(buf + 2 * itile * kFragBytes, lane, xpar, q);
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

  // step `it`: barrier | [release the item whose last iteration is now in LDS] | write iteration it+1 (registers of
  // parity (it+1)&1) to LDS | load iteration it+3 into the registers just freed | compute iteration it.
  // Two iterations of loads (2 x 64 KiB per workgroup) are always in flight.
  std::integral_constant<int, 0> p0;
  std::integral_constant<int, 1> p1;
  auto release = [&](int64_t it) {
    if ((it % IPG) == IPG - 1 && tid == 0)  // every wave has copied its share of the item out of the ring
      __hip_atomic_fetch_add(g32(&ctrl->released[flag][item_buffer(it)]), 1u, LNRF_RLX_AGENT);
  };
  if (iters > 0) {
    load(p0, 0);
    load(p1, 1);
    write(p0);
    load(p0, 2);
  }
  for (int64_t it = 0; it < iters; it += 2) {
    __syncthreads();
    release(it);
    write(p1);
    load(p1, it + 3);
    compute(0);
    __syncthreads();
    if (it + 1 < iters) release(it + 1);
    write(p0);
    load(p0, it + 4);
    if (it + 1 < iters) compute(1);
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
      if (hh == 0 && out_idx >= 0) atomicAdd(grads + b_off + out_idx, sacc);
    }
    static_for<TI>([&](auto a_) {
      constexpr int a = decltype(a_)::value;
      const int itile = wi + WI * a;
      static_for<16>([&](auto q_) {
        constexpr int qq = decltype(q_)::value;
        const int r = (qq & 3) + 8 * (qq >> 2) + 4 * hh;
        const int f = 2 * itile + (r >> 4);
        const int r16 = r & 15;
        const int in_idx = EPI::row(pb, f, r16);
        if (itile < NI && out_idx >= 0 && in_idx >= 0)
          atomicAdd(grads + w_off + (int64_t)in_idx * out_dim + out_idx, acc[a][b][qq]);
      });
    });
  });
}

__global__ __launch_bounds__(kThreads) void nerf_bwd_fused_kernel(
    FusedArgs args, const char* __restrict__ packed, const char* __restrict__ save, int64_t save_tiles,
    const float* __restrict__ density, const float* __restrict__ rgb, const float* __restrict__ g_density,
    const float* __restrict__ g_rgb, int64_t M, int64_t n_groups, char* ring, FusedCtrl* ctrl,
    float* __restrict__ grads) {
  if ((int)blockIdx.x < args.producers) {
    if (args.debug & 2) return;
    RingSink sink;
    sink.debug = args.debug;
#pragma unroll
    for (int f = 0; f < kChainFlags; ++f) sink.ncons[f] = args.ncons[f];
    sink.nb = args.nb;
    producer_role(sink, args.producers, packed, save, save_tiles, density, rgb, g_density, g_rgb, M, n_groups, ring,
                  ctrl);
    return;
  }
  const int cidx = (int)blockIdx.x - args.producers;
  WgradProblem pb = args.w.p[0];
  int flag = args.flag[0], y0 = args.y_local0[0];
#pragma unroll
  for (int i = 1; i < kMaxProblems; ++i)
    if (i < args.w.n_problems && cidx >= args.w.p[i].first_block) {
      pb = args.w.p[i];
      flag = args.flag[i];
      y0 = args.y_local0[i];
    }
  const int split = cidx - pb.first_block;
  switch (pb.shape) {
    case 0: wgrad_ring_body<16, 16, 4, 2, 2, NerfWgradEpi>(pb, flag, y0, args.nb, args.debug, split, save, save_tiles, ring, ctrl, n_groups, grads); break;
    case 1: wgrad_ring_body<16, 10, 4, 2, 2, NerfWgradEpi>(pb, flag, y0, args.nb, args.debug, split, save, save_tiles, ring, ctrl, n_groups, grads); break;
    case 2: wgrad_ring_body<4, 16, 2, 4, 2, NerfWgradEpi>(pb, flag, y0, args.nb, args.debug, split, save, save_tiles, ring, ctrl, n_groups, grads); break;
    case 3: wgrad_ring_body<2, 10, 1, 8, 4, NerfWgradEpi>(pb, flag, y0, args.nb, args.debug, split, save, save_tiles, ring, ctrl, n_groups, grads); break;
    default: wgrad_ring_body<8, 2, 4, 2, 4, NerfWgradEpi>(pb, flag, y0, args.nb, args.debug, split, save, save_tiles, ring, ctrl, n_groups, grads); break;
  }
}

}  // namespace lnrf

using namespace lnrf;

static int g_producer_permille = 550;
static int g_ring_buffers = 64;
static int g_debug = 0;

extern "C" int lnrf_nerf_bwd_fused_debug(int32_t mode) {
  g_debug = mode;
  return LNRF_OK;
}

extern "C" int lnrf_nerf_bwd_fused_tune(int32_t producer_permille, int32_t ring_buffers) {
  LNRF_CHECK_ARG(producer_permille >= 100 && producer_permille <= 900, "producer share must be 100..900 permille");
  LNRF_CHECK_ARG(ring_buffers >= 8 && ring_buffers <= kMaxNB, "ring buffers per flag must be 8..128");
  g_producer_permille = producer_permille;
  g_ring_buffers = ring_buffers;
  return LNRF_OK;
}

extern "C" int64_t lnrf_nerf_bwd_fused_workspace_bytes(const lnrf_nerf_shape* s) {
  return nerf_shape_fused(s) ? kCtrlBytes + (int64_t)kChainFlags * kMaxNB * kBufBytes : -1;
}

extern "C" int lnrf_nerf_mlp_bwd_fused(const lnrf_nerf_shape* shape, const void* packed, const void* save,
                                       const float* density, const float* rgb, const float* g_density,
                                       const float* g_rgb, int64_t m, void* workspace, float* grads,
                                       lnrf_stream_t stream) {
  if (!nerf_shape_fused(shape)) {
    set_error("lnrf_nerf_mlp_bwd_fused: only the default NeRFModel shape {5,4,256,128,10,4} is fused");
    return LNRF_ERR_UNSUPPORTED;
  }
  LNRF_CHECK_ARG(packed && save && density && rgb && g_density && g_rgb && workspace && grads, "null pointer");
  LNRF_CHECK_ARG(m >= 0, "bad m");
  if (m == 0) return LNRF_OK;
  hipStream_t st = as_stream(stream);
  const int64_t n_tiles = nerf_tiles_for(m);
  const int64_t n_groups = n_tiles / kWaves;

  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  int cus = 0;
  e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e != hipSuccess) return hip_fail(e, "hipDeviceGetAttribute(multiprocessor count)");
  const int lds = kFusedBwdLds + 1024;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(nerf_bwd_fused_kernel),
                          hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute(max dynamic LDS)");
  int per_cu = 0;
  e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nerf_bwd_fused_kernel, kThreads, lds);
  if (e != hipSuccess) return hip_fail(e, "hipOccupancyMaxActiveBlocksPerMultiprocessor");
  // every workgroup must be resident at once (producers and consumers wait for each other): one per CU
  const int grid = cus;
  if (per_cu < 1 || grid < 2 * kMaxProblems + 8) {
    set_error("lnrf_nerf_mlp_bwd_fused: device cannot keep %d workgroups resident (use lnrf_nerf_mlp_bwd)", grid);
    return LNRF_ERR_UNSUPPORTED;
  }

  FusedArgs a;
  int producers = (int)((int64_t)grid * g_producer_permille / 1000);
  int consumers = grid - producers;
  if (consumers < kMaxProblems) { consumers = kMaxProblems; producers = grid - consumers; }
  // consumer workgroups per problem, proportional to the operand bytes a problem streams per evaluation
  // (hidden x hidden 32 slots, z x dy10m 26, x_emb 20 + 20, d_emb 12, h10 10), every problem at least one
  const double weight[13] = {32, 32, 32, 32, 32, 32, 32, 32, 26, 20, 20, 12, 10};
  double wsum = 0;
  for (double w : weight) wsum += w;
  int blocks[13], used = 0;
  for (int i = 0; i < 13; ++i) {
    blocks[i] = (int)(consumers * weight[i] / wsum);
    if (blocks[i] < 1) blocks[i] = 1;
    used += blocks[i];
  }
  for (int i = 0; used < consumers; i = (i + 1) % 8) { ++blocks[i]; ++used; }     // leftovers to the big problems
  for (int i = 0; used > consumers; i = (i + 1) % 8) if (blocks[i] > 1) { --blocks[i]; --used; }
  const int c_total = build_wgrad_problems(a.w, blocks, 1 << 30);
  if (c_total != consumers) {
    set_error("lnrf_nerf_mlp_bwd_fused: internal consumer split error");
    return LNRF_ERR_ARG;
  }
  for (int f = 0; f < kChainFlags; ++f) a.ncons[f] = 0;
  for (int i = 0; i < a.w.n_problems; ++i) {
    a.flag[i] = flag_of_slot(a.w.p[i].y_slot0);
    a.y_local0[i] = a.w.p[i].y_slot0 - flag_slot0(a.flag[i]);
    a.ncons[a.flag[i]]++;
  }
  a.producers = producers;
  a.nb = g_ring_buffers;
  a.debug = g_debug;

  char* ws = (char*)workspace;
  e = hipMemsetAsync(ws, 0, kCtrlBytes, st);
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(ctrl)");
  hipLaunchKernelGGL(nerf_bwd_fused_kernel, dim3((unsigned)grid), dim3(kThreads), lds, st, a, (const char*)packed,
                     (const char*)save, n_tiles, density, rgb, g_density, g_rgb, m, n_groups, ws + kCtrlBytes,
                     (FusedCtrl*)ws, grads);
  LNRF_LAUNCH_CHECK();
  return LNRF_OK;
}
