// nerf_layout.h — data layout of the fused NeRFModel kernels (default shape only:
// input_layers=5, mid_layers=4, hidden=256, color=128, x_freqs=10, d_freqs=4; model.py:35-40).
//
// Shared by device code (nerf_mlp.hip) and a host build (layout_host.cpp) so that the index
// maps can be exercised on the CPU by tests/test_nerf_layout.py.
//
// Vocabulary
//   frag   : one MFMA operand for v_mfma_f32_32x32x16_bf16 = 64 lanes x 8 bf16 = 1 KiB.
//            A-frag lane l (r = l&31, h = l>>5), element j = A[row r][k = 8h + j].
//            B-frag lane l (c = l&31, h = l>>5), element j = B[k = 8h + j][col c].
//   k-step : 16 consecutive k of a layer's contraction = one frag per operand.
//   tile   : 32 evaluations (MFMA columns) handled by one wave.
//   The f32 result D[32 out rows][32 cols] keeps column c on lane c + 32*hh and row
//   (q&3) + 8*(q>>2) + 4*hh in register q, so converting registers 8s..8s+7 to bf16 gives the
//   B-frag of k-step s of the next layer with  k-slot (h, j) <-> feature 16s + 8(j>>2) + 4h + (j&3)
//   ("hidden" feature order below).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NL_HD __host__ __device__ __forceinline__
#else
#define NL_HD inline
#endif

namespace lnrf {
namespace nl {

constexpr int kFragBytes = 1024;
constexpr int kStageFrags = 16;  // frags consumed between two workgroup barriers

// ---- Flax parameter vector (Dense_i.kernel[in,out] row-major, then Dense_i.bias) ----------
constexpr int kNumDense = 12;
NL_HD constexpr int dense_in(int l) {
  return l == 0 ? 60 : (l == 5 ? 316 : (l == 10 ? 280 : (l == 11 ? 128 : 256)));
}
NL_HD constexpr int dense_out(int l) { return l == 9 ? 1 : (l == 10 ? 128 : (l == 11 ? 3 : 256)); }
NL_HD constexpr int dense_w_off(int l) {
  int off = 0;
  for (int i = 0; i < l; ++i) off += dense_in(i) * dense_out(i) + dense_out(i);
  return off;
}
NL_HD constexpr int dense_b_off(int l) { return dense_w_off(l) + dense_in(l) * dense_out(l); }
constexpr int kParamCount = dense_w_off(kNumDense);  // 593,924
static_assert(kParamCount == 593924, "NeRFModel parameter count");

// ---- k-slot <-> feature maps --------------------------------------------------------------
// hidden order: slot (ks, h, j) of a tensor produced by the previous MFMA
NL_HD constexpr int hidden_feat(int ks, int h, int j) { return 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3); }

// x embedding (60 features, model.py:72-77 order e = 20c + f [sin] / 20c + 10 + f [cos]) lives in
// 4 k-steps: lane-half h holds 15 (coordinate, frequency) pairs p_global = 15h + p, p = 4ks + (j>>1),
// element j&1 = {sin, cos}; p == 15 is padding.  Returns -1 for a pad slot.
NL_HD constexpr int xemb_feat(int ks, int h, int j) {
  const int p = 4 * ks + (j >> 1);
  if (p >= 15) return -1;
  const int pg = 15 * h + p;
  return 20 * (pg / 10) + (pg % 10) + 10 * (j & 1);
}
// d embedding (24 features, e = 8c + f / 8c + 4 + f) lives in 2 k-steps: 6 pairs per lane-half.
NL_HD constexpr int demb_feat(int ks, int h, int j) {
  const int p = 4 * ks + (j >> 1);
  if (p >= 6) return -1;
  const int pg = 6 * h + p;
  return 8 * (pg / 4) + (pg % 4) + 4 * (j & 1);
}

// ---- forward weight stream ------------------------------------------------------------------
// stream layers: 0..8 = Dense_0..8, 9 = "L10m" (Dense_10 with Dense_9 appended as out row 128),
// 10 = Dense_11.  Each layer: for out-tile o (32 rows), for k-step ks: one A-frag.
constexpr int kFwdLayers = 11;
NL_HD constexpr int fwd_nk(int s) { return s == 0 ? 4 : (s == 5 ? 20 : (s == 9 ? 18 : (s == 10 ? 8 : 16))); }
NL_HD constexpr int fwd_no(int s) { return s == 9 ? 5 : (s == 10 ? 1 : 8); }
NL_HD constexpr int round_up(int x, int m) { return (x + m - 1) / m * m; }
NL_HD constexpr int fwd_base(int s) {
  int b = 0;
  for (int i = 0; i < s; ++i) b += round_up(fwd_nk(i) * fwd_no(i), kStageFrags);
  return b;
}
constexpr int kFwdFrags = fwd_base(kFwdLayers);  // 1200
static_assert(kFwdFrags == 1200, "forward stream length");
// consumption order: fragments are used layer by layer, skipping the stage-alignment padding
NL_HD constexpr int fwd_cons_base(int s) {
  int b = 0;
  for (int i = 0; i < s; ++i) b += fwd_nk(i) * fwd_no(i);
  return b;
}
constexpr int kFwdUsed = fwd_cons_base(kFwdLayers);  // 1186
NL_HD constexpr int fwd_seq(int c) {  // c-th consumed fragment -> stream index
  int s = 0;
  for (int i = 1; i < kFwdLayers; ++i)
    if (c >= fwd_cons_base(i)) s = i;
  return fwd_base(s) + (c - fwd_cons_base(s));
}
// fp32 bias block, [stream layer][32*o + row]
NL_HD constexpr int fwd_bias_base(int s) {
  int b = 0;
  for (int i = 0; i < s; ++i) b += 32 * fwd_no(i);
  return b;
}
constexpr int kBiasFloats = fwd_bias_base(kFwdLayers);  // 2496
static_assert(kBiasFloats == 2496, "bias block");

struct WRef {  // reference to one scalar of the Flax parameter vector; idx < 0 = zero padding
  int idx;
};

// weight feeding A-frag element (lane, j) of frag (stream layer s, out-tile o, k-step ks)
NL_HD constexpr int fwd_weight_index(int s, int o, int ks, int lane, int j) {
  const int r = lane & 31, h = lane >> 5;
  if (s == 0) {
    const int in = xemb_feat(ks, h, j);
    return in < 0 ? -1 : dense_w_off(0) + in * 256 + (32 * o + r);
  }
  if (s <= 8) {
    int in;
    if (s == 5 && ks >= 16) {
      const int e = xemb_feat(ks - 16, h, j);
      if (e < 0) return -1;
      in = 256 + e;  // model.py:52 concat([z, x_emb])
    } else {
      in = hidden_feat(ks, h, j);
    }
    return dense_w_off(s) + in * 256 + (32 * o + r);
  }
  if (s == 9) {
    if (o < 4) {
      int in;
      if (ks >= 16) {
        const int e = demb_feat(ks - 16, h, j);
        if (e < 0) return -1;
        in = 256 + e;  // model.py:58 concat([z, d_emb])
      } else {
        in = hidden_feat(ks, h, j);
      }
      return dense_w_off(10) + in * 128 + (32 * o + r);
    }
    // o == 4: row 0 is the density logit (Dense_9, model.py:57); it sees z only
    if (r != 0 || ks >= 16) return -1;
    return dense_w_off(9) + hidden_feat(ks, h, j);
  }
  // s == 10: Dense_11 (128 -> 3)
  if (r >= 3) return -1;
  return dense_w_off(11) + hidden_feat(ks, h, j) * 3 + r;
}

// parameter index of bias element (stream layer s, row idx = 32*o + r); -1 = zero
NL_HD constexpr int fwd_bias_index(int s, int row) {
  if (s <= 8) return dense_b_off(s) + row;
  if (s == 9) return row < 128 ? dense_b_off(10) + row : (row == 128 ? dense_b_off(9) : -1);
  return row < 3 ? dense_b_off(11) + row : -1;
}

// ---- split-precision forward weight stream ("bf16x3") ------------------------------------------
// Every fp32 weight w is carried as two bf16 values hi = bf16(w), lo = bf16(w - hi); the stream holds, for
// each (layer, out-tile, k-step) of the forward stream above, the pair [hi frag, lo frag].  The render kernel
// forms hi*hi + hi*lo + lo*hi per product with fp32 accumulation (the dropped lo*lo term is ~2^-16 relative).
NL_HD constexpr int fwd3_base(int s) {
  int b = 0;
  for (int i = 0; i < s; ++i) b += round_up(2 * fwd_nk(i) * fwd_no(i), kStageFrags);
  return b;
}
constexpr int kFwd3Frags = fwd3_base(kFwdLayers);  // 2384
static_assert(kFwd3Frags == 2384, "split forward stream length");
constexpr int kFwd3Used = 2 * kFwdUsed;
NL_HD constexpr int fwd3_seq(int c) {  // c-th consumed fragment (2 per MFMA k-step: hi, lo) -> stream index
  int s = 0;
  for (int i = 1; i < kFwdLayers; ++i)
    if (c >= 2 * fwd_cons_base(i)) s = i;
  return fwd3_base(s) + (c - 2 * fwd_cons_base(s));
}
constexpr int64_t kPack3BiasOff = (int64_t)kFwd3Frags * kFragBytes;
constexpr int64_t kPack3Bytes = kPack3BiasOff + round_up(kBiasFloats * 4, 1024);

// ---- backward (input-gradient) weight stream --------------------------------------------------
// stream layers t: 0 = Dense_11^T, 1 = L10m^T (z rows only), 2..4 = Dense_8..6^T,
// 5 = Dense_5^T (h rows only), 6..9 = Dense_4..1^T.   A rows = layer inputs, k = layer outputs.
constexpr int kBwdLayers = 10;
NL_HD constexpr int bwd_nk(int t) { return t == 0 ? 1 : (t == 1 ? 9 : 16); }
NL_HD constexpr int bwd_no(int t) { return t == 0 ? 4 : 8; }
NL_HD constexpr int bwd_base(int t) {
  int b = 0;
  for (int i = 0; i < t; ++i) b += round_up(bwd_nk(i) * bwd_no(i), kStageFrags);
  return b;
}
constexpr int kBwdFrags = bwd_base(kBwdLayers);  // 1120
static_assert(kBwdFrags == 1120, "backward stream length");
NL_HD constexpr int bwd_cons_base(int t) {
  int b = 0;
  for (int i = 0; i < t; ++i) b += bwd_nk(i) * bwd_no(i);
  return b;
}
constexpr int kBwdUsed = bwd_cons_base(kBwdLayers);  // 1100
NL_HD constexpr int bwd_seq(int c) {
  int t = 0;
  for (int i = 1; i < kBwdLayers; ++i)
    if (c >= bwd_cons_base(i)) t = i;
  return bwd_base(t) + (c - bwd_cons_base(t));
}
NL_HD constexpr int bwd_dense(int t) { return t == 0 ? 11 : (t == 1 ? 10 : 10 - t); }  // t=2->8 ... t=9->1

NL_HD constexpr int bwd_weight_index(int t, int o, int ks, int lane, int j) {
  const int r = lane & 31, h = lane >> 5;
  const int in = 32 * o + r;  // row of the Flax kernel = input feature
  if (t == 0) {               // k slot (h=0, j<3) = rgb channel j
    if (h != 0 || j >= 3) return -1;
    return dense_w_off(11) + in * 3 + j;
  }
  if (t == 1) {
    if (ks < 8) return dense_w_off(10) + in * 128 + hidden_feat(ks, h, j);
    if (h == 0 && j == 0) return dense_w_off(9) + in;  // density-logit gradient slot
    return -1;
  }
  const int l = bwd_dense(t);
  return dense_w_off(l) + in * 256 + hidden_feat(ks, h, j);
}

// ---- Ref-NeRF normal-pass weight stream (ref_nerf.py:38-43: n = -d out[:, 0] / dx) ----------------------------
// The analytic normal is an input-gradient chain through the spatial block Dense_8 .. Dense_0 (same trunk as
// NeRFModel) down to the positional embedding.  Stream layers u, in consumption order:
//   0,1,2 = Dense_8,7,6^T | 3 = "X5": the x_emb rows of Dense_5 (316-wide input, rows 256..315) applied to dy5 |
//   4 = Dense_5^T (h rows) | 5..8 = Dense_4..1^T | 9 = "X0": Dense_0^T applied to dy0.
// X layers: k = hidden feature of the layer's output gradient, 16 k-steps x 2 out tiles; the 64 A rows are ordered
// so that every lane of the result tile holds BOTH the sin and the cos row of its (coordinate, frequency) pairs:
// result register q of lane half hh in tile o <-> pair pg = 16 o + 8 hh + (q >> 1) = 10 a + f (30 real pairs),
// row e = 20 a + f + 10 (q & 1) of model.py:72-77 (nrm_x_row); one sincos per pair applies the embedding Jacobian.
constexpr int kNrmLayers = 10;
NL_HD constexpr bool nrm_is_x(int u) { return u == 3 || u == 9; }
NL_HD constexpr int nrm_nk(int) { return 16; }
NL_HD constexpr int nrm_no(int u) { return nrm_is_x(u) ? 2 : 8; }
NL_HD constexpr int nrm_base(int u) {
  int b = 0;
  for (int i = 0; i < u; ++i) b += nrm_nk(i) * nrm_no(i);
  return b;
}
constexpr int kNrmFrags = nrm_base(kNrmLayers);  // 1088, every layer a whole number of 16-fragment stages
static_assert(kNrmFrags == 1088 && kNrmFrags % kStageFrags == 0, "normal-pass stream length");
NL_HD constexpr int nrm_seq(int c) { return c; }
// Dense layer applied by hidden stream layer u (u not an X layer)
NL_HD constexpr int nrm_dense(int u) { return u < 3 ? 8 - u : (u == 4 ? 5 : 9 - u); }  // 0->8 1->7 2->6 4->5 5->4 .. 8->1
NL_HD constexpr int nrm_x_row(int o, int r) {  // result row r = (q & 3) + 8 (q >> 2) + 4 hh  ->  embedding feature
  const int hh = (r >> 2) & 1, q = (r & 3) + 4 * (r >> 3);
  const int pg = 16 * o + 8 * hh + (q >> 1);
  if (pg >= 30) return -1;
  return 20 * (pg / 10) + (pg % 10) + 10 * (q & 1);
}
NL_HD constexpr int nrm_weight_index(int u, int o, int ks, int lane, int j) {
  const int r = lane & 31, h = lane >> 5;
  const int k = hidden_feat(ks, h, j);
  if (nrm_is_x(u)) {
    const int e = nrm_x_row(o, r);  // embedding feature fed by result row r of tile o
    if (e < 0) return -1;
    return u == 3 ? dense_w_off(5) + (256 + e) * 256 + k : dense_w_off(0) + e * 256 + k;
  }
  return dense_w_off(nrm_dense(u)) + (32 * o + r) * 256 + k;  // row of the Flax kernel = input feature
}

// ---- packed parameter blob --------------------------------------------------------------------
constexpr int64_t kPackFwdOff = 0;
constexpr int64_t kPackBwdOff = (int64_t)kFwdFrags * kFragBytes;
constexpr int64_t kPackBiasOff = kPackBwdOff + (int64_t)kBwdFrags * kFragBytes;
constexpr int64_t kPackBytes = kPackBiasOff + round_up(kBiasFloats * 4, 1024);
// ---- Ref-NeRF directional block (ref_nerf.py:100-107: Dense_9 273 -> 128 relu, Dense_10 128 -> 3) -------------------
// Input row = [spatial_out(256), IDE(16), -d.n(1)]; k slot (ks, h, j) of its 18 k-steps <-> input feature
// hidden_feat(ks, h, j) (valid below 273).  Forward stream: Dense_9 (4 out tiles x 18 k-steps, padded to 80
// fragments), Dense_10 (1 x 8, padded to 16).  Transposed stream: Dense_10^T (4 out tiles x 1 k-step, k slot
// (h 0, j < 3) = colour channel; padded to 16), Dense_9^T (9 out tiles = 288 >= 273 input rows, 8 k-steps; 72 -> 80).
constexpr int kDirIn = 273, kDirHidden = 128;
constexpr int kDirW9 = dense_w_off(9);                       // RefNERFModel's Dense_9 starts where NeRFModel's does
constexpr int kDirB9 = kDirW9 + kDirIn * kDirHidden;
constexpr int kDirW10 = kDirB9 + kDirHidden;
constexpr int kDirB10 = kDirW10 + kDirHidden * 3;
constexpr int kDirFwdFrags = 96, kDirBwdFrags = 96, kDirBiasFloats = 160;  // bias: 128 + 32 (3 real)
NL_HD constexpr int dir_fwd_seq(int c) { return c < 72 ? c : 80 + (c - 72); }  // 80 consumed fragments
NL_HD constexpr int dir_bwd_seq(int c) { return c < 4 ? c : 16 + (c - 4); }    // 76 consumed fragments
NL_HD constexpr int dir_fwd_weight_index(int g, int lane, int j) {
  const int r = lane & 31, h = lane >> 5;
  if (g < 72) {  // Dense_9: out tile o, k-step ks
    const int o = g / 18, ks = g % 18;
    const int in = hidden_feat(ks, h, j);
    return in < kDirIn ? kDirW9 + in * kDirHidden + 32 * o + r : -1;
  }
  if (g >= 80 && g < 88) {  // Dense_10
    if (r >= 3) return -1;
    return kDirW10 + hidden_feat(g - 80, h, j) * 3 + r;
  }
  return -1;
}
NL_HD constexpr int dir_bwd_weight_index(int g, int lane, int j) {
  const int r = lane & 31, h = lane >> 5;
  if (g < 4) {  // Dense_10^T: rows = hidden feature 32 g + r, k = colour channel
    if (h != 0 || j >= 3) return -1;
    return kDirW10 + (32 * g + r) * 3 + j;
  }
  if (g >= 16 && g < 88) {  // Dense_9^T: rows = input feature, k = hidden feature
    const int o = (g - 16) / 8, ks = (g - 16) % 8;
    const int in = 32 * o + r;
    return in < kDirIn ? kDirW9 + in * kDirHidden + hidden_feat(ks, h, j) : -1;
  }
  return -1;
}
NL_HD constexpr int dir_bias_index(int i) { return i < 128 ? kDirB9 + i : (i < 131 ? kDirB10 + (i - 128) : -1); }
// saved by the directional forward (1 KiB fragments, layout: fused_chain.h dump_off): the 18 input fragments, relu(Dense_9) (8), its mask (1)
constexpr int kDirSaveXin = 0, kDirSaveH = 18, kDirSaveMask = 26, kDirSaveSlots = 27;
// dumped by the directional backward: dy10 (2 slots, second zero) then dy9 (8 slots)
constexpr int kDirGradDy10 = 0, kDirGradDy9 = 2, kDirGradSlots = 10;

// Ref-NeRF blob: the NeRFModel blob (head layers zero), the normal-pass stream, the directional block
constexpr int64_t kRefPackNrmOff = kPackBytes;
constexpr int64_t kRefPackDirFwdOff = kRefPackNrmOff + (int64_t)kNrmFrags * kFragBytes;
constexpr int64_t kRefPackDirBwdOff = kRefPackDirFwdOff + (int64_t)kDirFwdFrags * kFragBytes;
constexpr int64_t kRefPackDirBiasOff = kRefPackDirBwdOff + (int64_t)kDirBwdFrags * kFragBytes;
constexpr int64_t kRefPackBytes = kRefPackDirBiasOff + 1024;

// ---- saved activations / gradient dumps -------------------------------------------------------
// Both buffers hold 1 KiB fragments addressed by (slot, tile) through dump_off() (tile-major [tile][slot] unless built with
// -DLNRF_DUMP_SLOT_MAJOR); a slot is one k-step (16 features) of one tensor.
// Inside the 1 KiB block lane (c, hh) of frag slot F stores its 16 bytes at dump_lane_off():
// the permutation makes the transposed LDS reads of the weight-gradient kernel conflict-free.
NL_HD constexpr int dump_lane_off(int slot, int c, int hh) {
  return 256 * (c >> 3) + 128 * (((c >> 2) & 1) ^ (slot & 1)) + 64 * hh + 16 * (c & 3);
}
// forward save slots
constexpr int kSaveXin = 0;    // 4 slots
constexpr int kSaveH = 4;      // h0..h7 (relu outputs of Dense_0..7): 16 slots each
constexpr int kSaveZ = 4 + 8 * 16;      // z = Dense_8 output (linear): 16 slots
constexpr int kSaveDin = kSaveZ + 16;   // d_emb, 2 slots, right behind z: [z | d_emb] is the input of Dense_10 and ONE
                                        // operand of its weight-gradient problem (dy10m is then read once, not twice)
constexpr int kSaveH10 = kSaveDin + 2;  // relu(Dense_10): 8 slots
constexpr int kSaveMask = kSaveH10 + 8;   // ReLU masks, one 1 KiB slot per layer: h0..h7, h10 (9 slots);
                                          // lane l keeps a uint4 at l*16: bit 16*o + q of the 128 = acc reg q
                                          // of out-tile o was > 0
constexpr int kSaveSlots = kSaveMask + 9;  // 167
// backward dump slots (pre-activation gradients)
constexpr int kGradDy11 = 0;            // 2 slots (second is zero padding)
constexpr int kGradDy10m = 2;           // 10 slots: 8 for Dense_10 outputs, slot 8 = logit, slot 9 zero
constexpr int kGradDy = 12;             // dy8, dy7, dy6, dy4, dy3, dy2, dy1, dy0, dy5: 16 slots each (grad_dy_slot)
constexpr int kGradSlots = kGradDy + 9 * 16;  // 156
// NeRFModel dump layout: tile-major unless built with -DLNRF_DUMP_SLOT_MAJOR (A/B; see fused_chain.h dump_off)
#ifdef LNRF_DUMP_SLOT_MAJOR
constexpr int kSaveTileSlots = 0, kGradTileSlots = 0, kDirSaveTileSlots = 0, kDirGradTileSlots = 0;
#else
constexpr int kSaveTileSlots = kSaveSlots, kGradTileSlots = kGradSlots;
constexpr int kDirSaveTileSlots = kDirSaveSlots, kDirGradTileSlots = kDirGradSlots;  // Ref-NeRF directional block
#endif
// dy5 sits behind dy0: x_emb^T [dy0 | dy5] (Dense_0 and rows 256.. of Dense_5) is ONE weight-gradient problem
NL_HD constexpr int grad_dy_pos(int l) { return l == 5 ? 8 : (l > 5 ? 8 - l : 7 - l); }
NL_HD constexpr int grad_dy_slot(int l) { return kGradDy + grad_dy_pos(l) * 16; }

}  // namespace nl
}  // namespace lnrf
