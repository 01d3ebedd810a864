// nerf_chain.h — parts of the NeRFModel backward shared by nerf_mlp.hip (separate chain / weight-gradient
// launches) and nerf_bwd_fused.hip (one persistent launch, chain workgroups feeding weight-gradient workgroups):
// the input-gradient chain of one 32-evaluation tile, written against a "sink" that receives the pre-activation
// gradient fragments, and the weight-gradient problem table.
#pragma once
#include "fused_chain.h"

namespace lnrf {

struct BwdSeq {
  static constexpr int count = kBwdUsed;
  static constexpr int at(int c) { return bwd_seq(c); }
};
constexpr int kBwdStages = kBwdFrags / kStageFrags;  // 70

static inline bool nerf_shape_fused(const lnrf_nerf_shape* s) {
  return s && s->input_layers == 5 && s->mid_layers == 4 && s->hidden_dim == 256 && s->color_layer_dim == 128 &&
         s->x_freqs == 10 && s->d_freqs == 4;
}
// Tiles are padded to whole workgroups (8 waves): every wave then owns a dump slot, so the dump stores need no
// branch — a conditional store makes hipcc lose count of the outstanding VMEM operations and wait vmcnt(0) (= drain
// all dump stores) before every ring write.  Padding tiles hold finite activations and zero gradients.
static inline int64_t nerf_tiles_for(int64_t m) {
  return ((m + kTileCols - 1) / kTileCols + kWaves - 1) / kWaves * kWaves;
}

// The chain publishes its dumps in 10 groups ("flags"), one per chain layer:
//   flag 0  = dy11 (2 slots) + dy10m incl. the density-logit slot (10 slots)   [known after layer T0]
//   flag t  = output of chain layer T_t, t = 1..9: dy8, dy7, ..., dy0 (16 slots each)
constexpr int kChainFlags = 10;
NL_HD constexpr int flag_slot0(int f) { return f == 0 ? 0 : grad_dy_slot(9 - f); }
NL_HD constexpr int flag_of_slot(int slot) {
  for (int f = 1; f < kChainFlags; ++f)
    if (slot >= flag_slot0(f) && slot < flag_slot0(f) + 16) return f;
  return 0;
}
NL_HD constexpr int flag_slots(int f) { return f == 0 ? kGradDy : 16; }

// Sink of the separate-launch path: fragments go to the gradient dump (layout: fused_chain.h dump_off) with non-temporal stores.
struct GlobalDumpSink {
  DumpAddr gd;
  template <int F>
  __device__ __forceinline__ void begin_flag() {}
  template <int F>
  __device__ __forceinline__ void end_flag() {}
  __device__ __forceinline__ void store(int slot, const bf16x8& f) { gd.store(slot, frag_to_bits(f)); }
};

// Input-gradient chain of the tile owned by this wave (what jax.grad does through model.py:49-60 back to front).
// SINK: begin_flag<F>() before the first store of flag F, store(slot, frag), end_flag<F>() after its last store.
// `ring` must be freshly constructed; the caller provides the workgroup barrier that separates two uses of the
// weight ring's LDS.
// HEAD_ONLY: stop after flag 1 (dy11, dy10m, dz): the head launch of the layer-stationary backward (nerf_bwd_ls.hip).
template <class SINK, class RING, bool HEAD_ONLY = false>
__device__ __forceinline__ void bwd_chain_tile(RING& ring, SINK& sink, const char* __restrict__ save,
                                               int64_t save_tiles, const float* __restrict__ density,
                                               const float* __restrict__ rgb, const float* __restrict__ g_density,
                                               const float* __restrict__ g_rgb, int64_t M, int64_t tile, int lane) {
  const int c = lane & 31, h = lane >> 5;
  const int64_t m = tile * kTileCols + c;
  const bool valid = m < M;

  // head gradients (fp32): d/d(pre-tanh) and d/d(density logit)
  float gy11[3] = {0, 0, 0}, gy9 = 0.0f;
  if (valid && h == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float y = rgb[m * 3 + k];
      gy11[k] = g_rgb[m * 3 + k] * (1.0f - y * y);  // tanh'
    }
    gy9 = g_density[m] * -expm1f(-density[m]);  // softplus' = sigmoid = 1 - exp(-sp), no cancellation
  }
  // ReLU masks of h0..h7 and h10 (written by the forward), 16 bytes per lane and layer
  uint4 relu_mask[9];
#pragma unroll
  for (int i = HEAD_ONLY ? 8 : 0; i < 9; ++i)  // the head launch only applies relu'(h10)
    relu_mask[i] = *reinterpret_cast<const uint4*>(save + dump_off(kSaveMask + i, tile, save_tiles, kSaveTileSlots) +
                                                   lane * 16);
  __syncthreads();
  ring.prologue();
  LNRF_TL_STAMP(ring);

  bf16x8 a0[16], a1[16];

  sink.template begin_flag<0>();
  // dy11 fragment: k slot (h=0, j<3) = rgb channel
  bf16x8 dy11 = zero_frag();
  dy11[0] = (__bf16)gy11[0];
  dy11[1] = (__bf16)gy11[1];
  dy11[2] = (__bf16)gy11[2];
  sink.store(kGradDy11, dy11);
  sink.store(kGradDy11 + 1, zero_frag());

  // T0: Dense_11^T -> dh10, masked by relu(h10)
  chain_layer<bwd_cons_base(0), bwd_nk(0), bwd_no(0)>(
      ring, [&](auto) { return zero_acc(); }, [&](auto) -> bf16x8 { return dy11; },
      [&](auto o_, const f32x16& acc) {
        constexpr int o = decltype(o_)::value;
        const unsigned mb = (o >> 1) == 0 ? relu_mask[8].x : relu_mask[8].y;
        a0[2 * o] = masked_frag<0>(acc, mb, 16 * (o & 1));
        a0[2 * o + 1] = masked_frag<1>(acc, mb, 16 * (o & 1));
        sink.store(kGradDy10m + 2 * o, a0[2 * o]);
        sink.store(kGradDy10m + 2 * o + 1, a0[2 * o + 1]);
      });
  // logit-gradient fragment: slot (h=0, j=0)
  bf16x8 dlogit = zero_frag();
  dlogit[0] = (__bf16)gy9;
  sink.store(kGradDy10m + 8, dlogit);
  sink.store(kGradDy10m + 9, zero_frag());
  sink.template end_flag<0>();

  // T1: [Dense_10 | Dense_9]^T (z rows) -> dz = dy8 (Dense_8 output is linear)
  sink.template begin_flag<1>();
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
        sink.store(grad_dy_slot(8) + 2 * o, a1[2 * o]);
        sink.store(grad_dy_slot(8) + 2 * o + 1, a1[2 * o + 1]);
      });
  sink.template end_flag<1>();

  if constexpr (HEAD_ONLY) return;
  // T2..T9: Dense_l^T for l = 8..1: dy_l (in) -> dh_{l-1}, masked by relu(h_{l-1}) -> dy_{l-1}
  auto back = [&](auto t_, bf16x8(&in)[16], bf16x8(&out)[16]) {
    constexpr int TT = decltype(t_)::value;
    constexpr int l = bwd_dense(TT);  // dense layer whose transpose is applied
    sink.template begin_flag<TT>();
    chain_layer<bwd_cons_base(TT), bwd_nk(TT), bwd_no(TT)>(
        ring, [&](auto) { return zero_acc(); },
        [&](auto k_) -> bf16x8 { return in[decltype(k_)::value]; },
        [&](auto o_, const f32x16& acc) {
          constexpr int o = decltype(o_)::value;
          const uint4 mk = relu_mask[l - 1];
          const unsigned mb = (o >> 1) == 0 ? mk.x : ((o >> 1) == 1 ? mk.y : ((o >> 1) == 2 ? mk.z : mk.w));
          out[2 * o] = masked_frag<0>(acc, mb, 16 * (o & 1));
          out[2 * o + 1] = masked_frag<1>(acc, mb, 16 * (o & 1));
          sink.store(grad_dy_slot(l - 1) + 2 * o, out[2 * o]);
          sink.store(grad_dy_slot(l - 1) + 2 * o + 1, out[2 * o + 1]);
        });
    sink.template end_flag<TT>();
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
// weight-gradient problems  dW_l[in][out] += sum_m X_l[m][in] * dy_l[m][out]
// ---------------------------------------------------------------------------------------------
enum { ROW_HIDDEN = 0, ROW_XEMB = 1, ROW_DEMB = 2, ROW_Z_DEMB = 3 };  // ROW_Z_DEMB: 16 slots of z, then 2 of d_emb
enum { COL_256 = 0, COL_DY10M = 1, COL_DY11 = 2, COL_EXPLICIT = 3, COL_DY0_DY5 = 4 };  // COL_DY0_DY5: tiles 0..7 dy0, 8..15 dy5
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
  // COL_EXPLICIT (kernels outside NeRFModel's parameter layout, e.g. RefNERFModel's directional block): float offsets
  // of the kernel / bias in the gradient vector, kernel columns, and number of real kernel rows
  int w_off, b_off, out_dim, n_rows;
};
constexpr int kMaxProblems = 13;
struct WgradArgs {
  WgradProblem p[kMaxProblems];
  int n_problems;
};

// NeRFModel gradient-vector addressing for the shared weight-gradient body
struct NerfWgradEpi {
  static __device__ __forceinline__ void cols(const WgradProblem& pb, int ot, int colr, int& out_idx, int& out_dim,
                                              int64_t& w_off, int64_t& b_off) {
    int dense_w = pb.dense;
    if (pb.col_map == COL_DY10M) {  // tiles 0..3 = Dense_10 outputs, tile 4 column 0 = Dense_9
      if (ot < 4) { out_idx = 32 * ot + colr; out_dim = 128; dense_w = 10; }
      else if (colr == 0 && pb.row_map != ROW_DEMB) { out_idx = 0; out_dim = 1; dense_w = 9; }
    } else if (pb.col_map == COL_DY0_DY5) {  // x_emb rows: Dense_0 (with its bias), then rows 256.. of Dense_5 (no bias)
      out_idx = 32 * (ot & 7) + colr; out_dim = 256;
      if (ot >= 8) {
        w_off = dense_w_off(5) + 256 * 256;
        b_off = -1;
        return;
      }
      dense_w = 0;
    } else if (pb.col_map == COL_DY11) {
      if (colr < 3) { out_idx = colr; out_dim = 3; }
    } else if (pb.col_map == COL_EXPLICIT) {
      if (32 * ot + colr < pb.out_dim) out_idx = 32 * ot + colr;
      out_dim = pb.out_dim;
      w_off = pb.w_off;
      b_off = pb.b_off;
      return;
    } else {
      out_idx = 32 * ot + colr; out_dim = 256;
    }
    w_off = dense_w_off(dense_w);
    b_off = dense_b_off(dense_w);
  }
  static __device__ __forceinline__ int row(const WgradProblem& pb, int f, int r16) {
    const int sh = (r16 >> 2) & 1, sj = 4 * (r16 >> 3) + (r16 & 3);  // slot (h, j) of that feature
    int in_idx;
    if (pb.row_map == ROW_HIDDEN) {
      in_idx = 16 * f + r16;
      if (pb.col_map == COL_EXPLICIT && in_idx >= pb.n_rows) in_idx = -1;
    } else if (pb.row_map == ROW_Z_DEMB) {
      if (f < 16) in_idx = 16 * f + r16;
      else {
        in_idx = demb_feat(f - 16, sh, sj);
        if (in_idx >= 0) in_idx += 256;
      }
    } else if (pb.row_map == ROW_XEMB) in_idx = xemb_feat(f, sh, sj);
    else in_idx = demb_feat(f, sh, sj);
    return in_idx >= 0 ? in_idx + pb.row_off : -1;
  }
  // rows of the kernel behind column tile `ot` (rows at or above it belong to no parameter): Dense_9 takes z only
  static __device__ __forceinline__ int row_limit(const WgradProblem& pb, int ot) {
    return pb.col_map == COL_DY10M && ot >= 4 ? 256 : 0x7FFFFFFF;
  }
};

// launches nerf_wgrad_kernel (nerf_mlp.hip) on `blocks` workgroups: X operands from xbuf, dy operands from ybuf (both
// dumps of n_tiles tiles in the layout `lay` names: slot-major by default, see fused_chain.h dump_off)
int launch_nerf_wgrad(const WgradArgs& args, int blocks, const void* xbuf, const void* ybuf, int64_t n_tiles,
                      float* grads, hipStream_t stream, WgLayout lay = WgLayout{}, float* slabs = nullptr,
                      bool plain_loads = false, bool fold = true);  // fold == false: the caller folds the slabs itself

// Layer-stationary backward of the eight 256 x 256 layers Dense_8 .. Dense_1 of ONE model (nerf_bwd_ls.hip): `scratch` holds
// the gradient dump with dy8 already written (slots grad_dy_slot(8)..), behind it room for ls_scratch_bytes(m); on return
// the launches that add dW_1..8 and db_1..8 to `grads` and leave dy7..dy0 in the dump are enqueued.  Used by the NeRFModel
// backward and by RefNERFModel's first-order trunk backward.  ls_dump_bytes / ls_small_slab_off: layout of `scratch`.
int64_t ls_scratch_bytes(int64_t m);
int64_t ls_small_slab_off(int64_t m);
int launch_ls_pipeline(const void* packed, const void* save, void* scratch, int64_t m, float* grads, hipStream_t stream);

// The 13 problems of one NeRFModel, heaviest first; `blocks[i]` workgroups for problem i (capped by `cap`).
// Block budget per problem in the order: Dense_1..8 (hidden x hidden), z x dy10m, x_emb x dy0, x_emb x dy5,
// d_emb x dy10m, h10 x dy11.
static inline int build_wgrad_problems(WgradArgs& a, const int (&blocks)[13], int64_t cap) {
  a.n_problems = 0;
  int first = 0;
  auto add = [&](int shape, int xs, int ys, int dense, int row_map, int row_off, int col_map, int do_bias) {
    WgradProblem p;
    p.shape = shape; p.x_slot0 = xs; p.y_slot0 = ys; p.dense = dense; p.row_map = row_map; p.row_off = row_off;
    p.col_map = col_map; p.do_bias = do_bias;
    p.w_off = p.b_off = p.out_dim = p.n_rows = 0;
    int64_t nb = blocks[a.n_problems];
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    p.first_block = first;
    p.n_blocks = (int)nb;
    first += (int)nb;
    a.p[a.n_problems++] = p;
  };
  for (int l = 1; l <= 8; ++l) add(0, kSaveH + (l - 1) * 16, grad_dy_slot(l), l, ROW_HIDDEN, 0, COL_256, 1);
  add(1, kSaveZ, kGradDy10m, 10, ROW_HIDDEN, 0, COL_DY10M, 1);           // Dense_10 rows 0..255 and Dense_9
  add(2, kSaveXin, grad_dy_slot(0), 0, ROW_XEMB, 0, COL_256, 1);         // Dense_0
  add(2, kSaveXin, grad_dy_slot(5), 5, ROW_XEMB, 256, COL_256, 0);       // Dense_5 rows 256..315
  add(3, kSaveDin, kGradDy10m, 10, ROW_DEMB, 256, COL_DY10M, 0);         // Dense_10 rows 256..279
  add(4, kSaveH10, kGradDy11, 11, ROW_HIDDEN, 0, COL_DY11, 1);           // Dense_11
  return first;
}

}  // namespace lnrf
