// layout_host.cpp — host build of the index maps of the fused NeRF kernels (nerf_layout.h) and of the
// positional-encoding sincos (fast_math.h), so that tests/test_nerf_layout.py can run the exact packing /
// fragment logic on the CPU (no GPU needed) with an MFMA emulator.
#include <stdint.h>

#include "fast_math.h"
#include "nerf_layout.h"

using namespace lnrf::nl;

extern "C" {
int lnrf_host_fwd_frags(void) { return kFwdFrags; }
int lnrf_host_bwd_frags(void) { return kBwdFrags; }
int lnrf_host_bias_floats(void) { return kBiasFloats; }
int lnrf_host_fwd_used(void) { return kFwdUsed; }
int lnrf_host_fwd_seq(int c) { return fwd_seq(c); }
int lnrf_host_fwd3_frags(void) { return kFwd3Frags; }
int lnrf_host_fwd3_used(void) { return kFwd3Used; }
int lnrf_host_fwd3_seq(int c) { return fwd3_seq(c); }
int lnrf_host_fwd3_base(int s) { return fwd3_base(s); }
int lnrf_host_bwd_seq(int c) { return bwd_seq(c); }
int lnrf_host_fwd_layer_info(int s, int what) {
  return what == 0 ? fwd_nk(s) : what == 1 ? fwd_no(s) : what == 2 ? fwd_base(s) : what == 3 ? fwd_bias_base(s)
                                                                                             : fwd_cons_base(s);
}
int lnrf_host_bwd_layer_info(int t, int what) {
  return what == 0 ? bwd_nk(t) : what == 1 ? bwd_no(t) : what == 2 ? bwd_base(t) : bwd_dense(t);
}
// parameter index (or -1) feeding element j of lane `lane` of stream fragment g
int lnrf_host_fwd_weight_index(int g, int lane, int j) {
  int s = 0;
  for (int i = 1; i < kFwdLayers; ++i)
    if (g >= fwd_base(i)) s = i;
  const int loc = g - fwd_base(s);
  if (loc >= fwd_nk(s) * fwd_no(s)) return -1;
  return fwd_weight_index(s, loc / fwd_nk(s), loc % fwd_nk(s), lane, j);
}
int lnrf_host_bwd_weight_index(int g, int lane, int j) {
  int t = 0;
  for (int i = 1; i < kBwdLayers; ++i)
    if (g >= bwd_base(i)) t = i;
  const int loc = g - bwd_base(t);
  if (loc >= bwd_nk(t) * bwd_no(t)) return -1;
  return bwd_weight_index(t, loc / bwd_nk(t), loc % bwd_nk(t), lane, j);
}
int lnrf_host_fwd_bias_index(int i) {
  int s = 0;
  for (int k = 1; k < kFwdLayers; ++k)
    if (i >= fwd_bias_base(k)) s = k;
  return fwd_bias_index(s, i - fwd_bias_base(s));
}
int lnrf_host_nrm_frags(void) { return kNrmFrags; }
int lnrf_host_nrm_weight_index(int g, int lane, int j) {
  int u = 0;
  for (int i = 1; i < kNrmLayers; ++i)
    if (g >= nrm_base(i)) u = i;
  const int loc = g - nrm_base(u);
  return nrm_weight_index(u, loc / nrm_nk(u), loc % nrm_nk(u), lane, j);
}
int lnrf_host_xemb_feat(int ks, int h, int j) { return xemb_feat(ks, h, j); }
// slot orders of the save / gradient-dump tiles: what 0 x_emb, 1 h_l (arg = l), 2 z, 3 d_emb, 4 h10, 5 masks, 6 slots per tile
int lnrf_host_save_slot(int what, int arg) {
  return what == 0 ? kSaveXin : what == 1 ? kSaveH + 16 * arg : what == 2 ? kSaveZ : what == 3 ? kSaveDin
       : what == 4 ? kSaveH10 : what == 5 ? kSaveMask : kSaveSlots;
}
// what 0 dy11, 1 dy10m, 2 dy_l (arg = l), 3 slots per tile
int lnrf_host_grad_slot(int what, int arg) {
  return what == 0 ? kGradDy11 : what == 1 ? kGradDy10m : what == 2 ? grad_dy_slot(arg) : kGradSlots;
}
int lnrf_host_demb_feat(int ks, int h, int j) { return demb_feat(ks, h, j); }
int lnrf_host_dump_lane_off(int slot, int c, int hh) { return dump_lane_off(slot, c, hh); }
void lnrf_host_sincos_pe(float r, float* s, float* c) { lnrf::sincos_pe(r, s, c); }
}
