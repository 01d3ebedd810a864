"""
CPU check of the fused-kernel data layout (csrc/nerf_layout.h, compiled for the host into
liblnrf_layout_host.so): the weight packing, the k-slot <-> feature permutations, the stage/consumption
order and the positional-encoding sincos are exercised with a NumPy emulation of
v_mfma_f32_32x32x16_bf16 (operand / accumulator lane maps from the CDNA4 guide) and must reproduce the
oracle's bf16-operand NeRFModel forward.  No GPU involved.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import model as OM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_LIB = os.path.join(ROOT, "learn-nerf_amd", "lib", "liblnrf_layout_host.so")


@pytest.fixture(scope="module")
def H():
    if not os.path.exists(HOST_LIB):
        pytest.skip("liblnrf_layout_host.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(HOST_LIB)
    lib.lnrf_host_sincos_pe.argtypes = [ctypes.c_float, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    return lib


def bf16(x):
    return torch.from_numpy(np.asarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def mfma_32x32x16(a_frag, b_frag, acc):
    """D = A B + C with the gfx950 lane maps: A/B frag [64 lanes][8], acc [64 lanes][16]."""
    A = np.zeros((32, 16), np.float64)
    B = np.zeros((16, 32), np.float64)
    for lane in range(64):
        r, h = lane & 31, lane >> 5
        A[r, 8 * h:8 * h + 8] = a_frag[lane]
        B[8 * h:8 * h + 8, r] = b_frag[lane]
    D = A @ B
    out = acc.copy()
    for lane in range(64):
        col, hh = lane & 31, lane >> 5
        for q in range(16):
            out[lane, q] += D[(q & 3) + 8 * (q >> 2) + 4 * hh, col]
    return out


def test_stream_sizes_and_consumption_order(H):
    assert H.lnrf_host_fwd_frags() == 1200 and H.lnrf_host_bwd_frags() == 1120 and H.lnrf_host_bias_floats() == 2496
    assert H.lnrf_host_fwd_used() == 1186
    seq = [H.lnrf_host_fwd_seq(c) for c in range(1186)]
    assert seq == sorted(seq) and len(set(seq)) == 1186 and seq[-1] < 1200
    assert seq[1088 - 6 + 0] == 1088 - 6  # no gap before the padded L10m layer ...
    # ... but the 6 padding fragments after its 90 are skipped
    c10 = H.lnrf_host_fwd_layer_info(10, 4)
    assert H.lnrf_host_fwd_seq(c10) == H.lnrf_host_fwd_layer_info(10, 2) == 1184
    bseq = [H.lnrf_host_bwd_seq(c) for c in range(1100)]
    assert bseq == sorted(bseq) and bseq[4] == 16  # Dense_11^T uses 4 fragments of its 16-fragment stage


def test_split_precision_stream_order(H):
    """bf16x3 render stream: [hi, lo] fragment pairs in forward order; the weight ring publishes one 16-fragment
    stage per barrier, so consumption must be monotone and must enter EVERY stage at its first fragment."""
    n, used = H.lnrf_host_fwd3_frags(), H.lnrf_host_fwd3_used()
    assert n == 2384 and n % 16 == 0 and used == 2 * 1186
    seq = [H.lnrf_host_fwd3_seq(c) for c in range(used)]
    assert seq == sorted(seq) and len(set(seq)) == used and seq[-1] < n
    entered = [g // 16 for g in seq if g % 16 == 0]
    assert entered == list(range(n // 16)), "a stage would be skipped or entered mid-way"
    # pair c (hi, lo) of layer s sits at fwd3_base(s) + 2 * (index inside the layer), i.e. mirrors fwd_seq
    for s in range(11):
        c0 = H.lnrf_host_fwd_layer_info(s, 4)
        assert H.lnrf_host_fwd3_seq(2 * c0) == H.lnrf_host_fwd3_base(s)
        assert H.lnrf_host_fwd3_seq(2 * c0 + 1) == H.lnrf_host_fwd3_base(s) + 1


def test_every_weight_is_packed_exactly_once(H):
    counts = np.zeros(593_924, np.int32)
    for g in range(1200):
        for lane in range(64):
            for j in range(8):
                idx = H.lnrf_host_fwd_weight_index(g, lane, j)
                if idx >= 0:
                    counts[idx] += 1
    for i in range(2496):
        idx = H.lnrf_host_fwd_bias_index(i)
        if idx >= 0:
            counts[idx] += 1
    assert counts.min() == 1 and counts.max() == 1, "forward stream + bias block must cover every parameter once"
    # transposed stream: every kernel that needs an input gradient, once; Dense_0 and the x_emb/d_emb rows never
    bcounts = np.zeros(593_924, np.int32)
    for g in range(1120):
        for lane in range(64):
            for j in range(8):
                idx = H.lnrf_host_bwd_weight_index(g, lane, j)
                if idx >= 0:
                    bcounts[idx] += 1
    dims = OM.nerf_layer_dims()
    off = 0
    for l, (fi, fo) in enumerate(dims):
        k = bcounts[off:off + fi * fo].reshape(fi, fo)
        if l == 0:
            assert k.sum() == 0
        elif l in (5, 10):
            assert (k[:256] == 1).all() and k[256:].sum() == 0
        else:
            assert (k == 1).all()
        assert bcounts[off + fi * fo:off + fi * fo + fo].sum() == 0
        off += fi * fo + fo


def test_sincos_pe_accuracy(H):
    rng = np.random.default_rng(0)
    worst = 0.0
    for _ in range(20000):
        r = np.float32(rng.uniform(-1.8, 1.8) * 2 ** rng.integers(0, 10))
        s, c = ctypes.c_float(), ctypes.c_float()
        H.lnrf_host_sincos_pe(r, ctypes.byref(s), ctypes.byref(c))
        worst = max(worst, abs(s.value - np.sin(np.float64(r))), abs(c.value - np.cos(np.float64(r))))
    assert worst < 2e-7


def test_dump_lane_offsets_are_a_permutation_and_conflict_free(H):
    for slot in (0, 1, 7):
        offs = sorted(H.lnrf_host_dump_lane_off(slot, c, hh) for c in range(32) for hh in range(2))
        assert offs == list(range(0, 1024, 16))
    # transposed read of the weight-gradient kernel: 32 lanes (2 fragments x 4 rows x 4 pieces) -> 64 distinct banks
    for c0 in (0, 8, 20):
        banks = set()
        for frag in range(2):
            for qp in range(4):
                for p in range(4):
                    off = H.lnrf_host_dump_lane_off(frag, c0 + qp, p & 1) + 8 * (p >> 1) + 1024 * frag
                    banks.update({(off // 4) % 64, (off // 4 + 1) % 64})
        assert len(banks) == 64


def test_emulated_forward_chain_matches_oracle(H):
    """Run the forward 'program' of nerf_fwd_kernel for one 32-evaluation tile with the real packing maps."""
    gen = torch.Generator().manual_seed(0)
    dims = OM.nerf_layer_dims()
    flat = OM.lecun_normal_init(dims, gen)
    off = 0
    for fi, fo in dims:
        off += fi * fo
        flat[off:off + fo] = torch.randn(fo, generator=gen) * 0.1
        off += fo
    x = (torch.rand(32, 3, generator=gen) * 2 - 1).float()
    d = torch.randn(32, 3, generator=gen)
    d = (d / d.norm(dim=-1, keepdim=True)).float()
    P = flat.numpy()

    def a_frag(g):
        f = np.zeros((64, 8), np.float32)
        for lane in range(64):
            for j in range(8):
                idx = H.lnrf_host_fwd_weight_index(g, lane, j)
                f[lane, j] = P[idx] if idx >= 0 else 0.0
        return bf16(f).astype(np.float64)

    bias = np.array([P[i] if (i := H.lnrf_host_fwd_bias_index(k)) >= 0 else 0.0 for k in range(2496)], np.float64)
    xe = OM.sinusoidal_emb(x.double(), 10).numpy()
    de = OM.sinusoidal_emb(d.double(), 4).numpy()

    def emb_frags(emb, feat, nks):
        fr = np.zeros((nks, 64, 8), np.float32)
        for ks in range(nks):
            for lane in range(64):
                c, h = lane & 31, lane >> 5
                for j in range(8):
                    e = feat(ks, h, j)
                    fr[ks, lane, j] = emb[c, e] if e >= 0 else 0.0
        return bf16(fr).astype(np.float64)

    xin = emb_frags(xe, H.lnrf_host_xemb_feat, 4)
    din = emb_frags(de, H.lnrf_host_demb_feat, 2)

    def layer(s, b_frags, relu):
        nk, no, base, bias0 = (H.lnrf_host_fwd_layer_info(s, w) for w in range(4))
        outs = []
        for o in range(no):
            acc = np.zeros((64, 16), np.float64)
            for lane in range(64):
                hh = lane >> 5
                for q in range(16):
                    acc[lane, q] = bias[bias0 + 32 * o + (q & 3) + 8 * (q >> 2) + 4 * hh]
            for ks in range(nk):
                acc = mfma_32x32x16(a_frag(base + o * nk + ks), b_frags[ks], acc)
            outs.append(acc)
        frags = []
        for acc in outs:  # registers 8s..8s+7 -> B-frag of k-step s of the next layer
            v = np.maximum(acc, 0) if relu else acc
            frags += [bf16(v[:, :8]).astype(np.float64), bf16(v[:, 8:]).astype(np.float64)]
        return outs, frags

    _, act = layer(0, list(xin), True)
    for s in (1, 2, 3, 4):
        _, act = layer(s, act, True)
    _, act = layer(5, act + list(xin), True)
    for s in (6, 7):
        _, act = layer(s, act, True)
    _, z = layer(8, act, False)
    outs, h10 = layer(9, z + list(din), True)
    logit = outs[4][:32, 0]  # row 0 of out-tile 4 lives in register 0 of lanes 0..31
    density = np.logaddexp(logit, 0.0)
    outs11, _ = layer(10, h10[:8], False)
    rgb = np.tanh(outs11[0][:32, :3])
    rd, rr, _ = OM.nerf_mlp(flat.double(), x.double(), d.double(), operand_round=OM.bf16_round)
    assert np.abs(rgb - rr.numpy()).max() < 2e-5
    assert np.abs(density - rd.numpy()[:, 0]).max() < 2e-5


def test_slot_orders_keep_the_merged_weight_gradient_operands_adjacent(H):
    """nerf_layout.h: the weight-gradient problems [z | d_emb] x dy10m and x_emb x [dy0 | dy5] read ONE run of slots per
    operand, and the head-with-weights experiment one run of 26 (z, d_emb, h10): the slot orders must keep those
    neighbours, every tensor on an even slot (fragment pairs / the parity swizzle of dump_lane_off), nothing overlapping."""
    S, G = H.lnrf_host_save_slot, H.lnrf_host_grad_slot
    assert S(3, 0) == S(2, 0) + 16 and S(4, 0) == S(3, 0) + 2          # z, d_emb, h10
    assert S(1, 0) == S(0, 0) + 4 and S(2, 0) == S(1, 7) + 16           # x_emb, h0..h7, z
    assert S(5, 0) == S(4, 0) + 8 and S(6, 0) == S(5, 0) + 9 == 167
    assert all(S(w, a) % 2 == 0 for w, a in [(0, 0), (2, 0), (3, 0), (4, 0)] + [(1, l) for l in range(8)])
    dy = [G(2, l) for l in range(9)]
    assert G(2, 5) == G(2, 0) + 16                                       # dy5 right behind dy0
    assert sorted(dy) == list(range(G(1, 0) + 10, G(3, 0), 16)) and all(v % 2 == 0 for v in dy)
    assert G(0, 0) == 0 and G(1, 0) == 2 and G(3, 0) == 156
