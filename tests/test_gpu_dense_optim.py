"""GPU parity of the exact-fp32 dense path (f32 MFMA) and the fused Adam / norm kernels."""
import pytest
import torch

from oracle import model as OM
from oracle import train as OT

pytestmark = pytest.mark.gpu
F64 = torch.float64


@pytest.mark.parametrize("m,k,n,act", [(1, 3, 1, 0), (100, 60, 256, 1), (257, 316, 256, 0), (70, 256, 1, 2),
                                       (129, 128, 3, 3), (64, 32, 64, 4), (65, 40, 64, 5)])
def test_dense_fwd(m, k, n, act):
    from learn_nerf import ops

    gen = torch.Generator().manual_seed(m + k)
    x = torch.randn(m, k, generator=gen)
    w = torch.randn(k, n, generator=gen) / k ** 0.5
    b = torch.randn(n, generator=gen)
    pre = x.double() @ w.double() + b.double()
    ref = [pre, torch.relu(pre), torch.nn.functional.softplus(pre), torch.tanh(pre), torch.exp(pre),
           torch.sigmoid(pre)][act]
    y = ops.dense_fwd(x.cuda(), w.cuda(), b.cuda(), act)
    assert torch.allclose(y.cpu().double(), ref, atol=2e-5, rtol=2e-5)


def test_dense_strided_views_concat():
    from learn_nerf import ops

    gen = torch.Generator().manual_seed(1)
    m = 90
    buf = torch.zeros(m, 316, device="cuda")
    h = torch.randn(m, 256, generator=gen)
    buf[:, :256] = h.cuda()
    x = torch.rand(m, 3, generator=gen) * 2 - 1
    ops.sinusoidal_emb_into(x.cuda(), 10, buf, col_off=256)
    emb_ref = OM.sinusoidal_emb(x.double(), 10)
    # fp32 argument 2^f*x is exact, sincosf is accurate to ~1 ulp
    assert torch.allclose(buf[:, 256:].cpu().double(), emb_ref, atol=3e-7)
    w = torch.randn(316, 256, generator=gen) / 316 ** 0.5
    y = ops.dense_fwd(buf, w.cuda(), None, 0)
    ref = torch.cat([h.double(), emb_ref], 1) @ w.double()
    assert torch.allclose(y.cpu().double(), ref, atol=3e-5, rtol=3e-5)


@pytest.mark.parametrize("m,k,n", [(300, 60, 256), (1000, 256, 3), (5000, 280, 128), (77, 256, 1)])
def test_dense_bwd(m, k, n):
    from learn_nerf import ops

    gen = torch.Generator().manual_seed(k)
    x = torch.randn(m, k, generator=gen)
    w = torch.randn(k, n, generator=gen) / k ** 0.5
    gy = torch.randn(m, n, generator=gen)
    gx = ops.dense_bwd_input(gy.cuda(), w.cuda())
    assert torch.allclose(gx.cpu().double(), gy.double() @ w.double().T, atol=5e-5, rtol=5e-5)
    gx2 = ops.dense_bwd_input(gy.cuda(), w.cuda(), out=gx.clone(), accumulate=True)
    assert torch.allclose(gx2, 2 * gx, atol=1e-5, rtol=1e-5)
    gw = torch.zeros(k, n, device="cuda")
    gb = torch.zeros(n, device="cuda")
    ops.dense_bwd_weight(x.cuda(), gy.cuda(), gw, gb)
    ref_w = x.double().T @ gy.double()
    scale = ref_w.abs().max().item()
    assert (gw.cpu().double() - ref_w).abs().max().item() <= 2e-5 * scale
    assert torch.allclose(gb.cpu().double(), gy.double().sum(0), atol=1e-3, rtol=1e-4)


def test_act_bwd():
    from learn_nerf import ops

    gen = torch.Generator().manual_seed(0)
    pre = torch.randn(50, 20, generator=gen, dtype=F64)
    g = torch.randn(50, 20, generator=gen)
    fns = {1: torch.relu, 2: torch.nn.functional.softplus, 3: torch.tanh, 4: torch.exp, 5: torch.sigmoid}
    for act, fn in fns.items():
        p = pre.clone().requires_grad_(True)
        y = fn(p)
        (ref,) = torch.autograd.grad(y, p, g.double())
        out = ops.act_bwd_(g.cuda().clone(), y.detach().float().cuda(), act)
        assert torch.allclose(out.cpu().double(), ref, atol=1e-5, rtol=1e-4), act


@pytest.mark.parametrize("n", [3, 1024, 593_924 * 2 + 3])
def test_adam_and_sq_norm(n):
    from learn_nerf import ops

    gen = torch.Generator().manual_seed(n)
    p = torch.randn(n, generator=gen)
    g = torch.randn(n, generator=gen) * 1e-2
    m = torch.zeros(n)
    v = torch.zeros(n)
    pd, md, vd = p.cuda(), m.cuda(), v.cuda()
    pr, mr, vr = p.double(), m.double(), v.double()
    for step in (1, 2, 3):
        ops.adam_step_(pd, (g * step).cuda(), md, vd, 1e-4, 0.9, 0.999, 1e-7, step, grad_scale=0.5)
        pr, mr, vr = OT.adam_update(pr, (g * step).double() * 0.5, mr, vr, step, 1e-4, 0.9, 0.999, 1e-7)
    assert torch.allclose(pd.cpu().double(), pr, atol=1e-6, rtol=1e-6)
    assert torch.allclose(md.cpu().double(), mr, atol=1e-8, rtol=1e-5)
    assert torch.allclose(vd.cpu().double(), vr, atol=1e-10, rtol=1e-5)
    out = torch.zeros(1, device="cuda")
    ops.sq_norm_into(pd, out)
    ref = (pd.cpu().double() ** 2).sum().item()
    assert abs(out.item() - ref) <= 1e-5 * ref
    # fused variant: same update, plus sum g^2 (as passed in) and sum p^2 (before the update) in the same pass
    p2, m2, v2 = pd.clone(), md.clone(), vd.clone()
    sums = torch.zeros(4, device="cuda")
    sums[0], sums[1] = 6.0, 12.0  # stand-ins for the step's squared-error sums
    gg = (g * 4).cuda()
    ops.adam_step_(p2, gg, m2, v2, 1e-4, 0.9, 0.999, 1e-7, 4, grad_scale=0.5, sq_norms=sums[2:4])
    ops.adam_step_(pd, gg, md, vd, 1e-4, 0.9, 0.999, 1e-7, 4, grad_scale=0.5)
    assert torch.equal(p2, pd) and torch.equal(m2, md) and torch.equal(v2, vd)
    log = ops.step_log(sums, 0.5, 0.5, clear=True).cpu().double()
    g_ref = (gg.cpu().double() ** 2).sum().sqrt().item() * 0.5
    assert abs(log[0] - 3.0) < 1e-6 and abs(log[1] - 6.0) < 1e-6
    assert abs(log[2] - g_ref) <= 1e-5 * g_ref and abs(log[3] - ref ** 0.5) <= 1e-5 * ref ** 0.5
    assert (sums == 0).all()  # cleared for the next step


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("m,k,n", [(1000, 256, 256), (777, 316, 256), (4101, 64, 128), (130, 280, 132)])
def test_vectorised_gemm_all_layouts(precision, m, k, n):
    """
    The 128x128 float4 kernel (taken for aligned operands with >= 64 rows / columns): forward (A contiguous in r,
    B in j), input gradient (A in r, B in r), weight gradient (A in i, B in j, split-K atomics), ragged tile
    edges, both operand precisions.  bf16 reference = the same products with bf16-rounded operands.
    """
    from learn_nerf import _lib as L
    from learn_nerf import ops

    gen = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, k, generator=gen)
    w = torch.randn(k, n, generator=gen) / k ** 0.5
    b = torch.randn(n, generator=gen)
    gy = torch.randn(m, n, generator=gen)
    rnd = (lambda t: t.bfloat16().double()) if precision == "bf16" else (lambda t: t.double())
    tol = dict(atol=1e-4, rtol=1e-4)
    with ops.dense_precision(precision):
        y = ops.dense_fwd(x.cuda(), w.cuda(), b.cuda(), L.ACT_RELU)
        gx = ops.dense_bwd_input(gy.cuda(), w.cuda())
        gw = torch.zeros(k, n, device="cuda")
        gb = torch.zeros(n, device="cuda")
        ops.dense_bwd_weight(x.cuda(), gy.cuda(), gw, gb)
    assert torch.allclose(y.cpu().double(), torch.relu(rnd(x) @ rnd(w) + b.double()), **tol)
    assert torch.allclose(gx.cpu().double(), rnd(gy) @ rnd(w).T, **tol)
    ref_gw = rnd(x).T @ rnd(gy)
    assert ((gw.cpu().double() - ref_gw).abs().max() / ref_gw.abs().max()).item() < 1e-5
    assert torch.allclose(gb.cpu().double(), gy.double().sum(0), atol=1e-3, rtol=1e-4)
    if precision == "bf16":  # and it is not the exact product
        assert (y.cpu().double() - torch.relu(x.double() @ w.double() + b.double())).abs().max().item() > 1e-3


@pytest.mark.parametrize("m,k,n,g", [(300, 256, 256, 256), (1000, 316, 256, 256), (77, 40, 64, 40), (513, 64, 3, 64)])
def test_gated_input_gradient_and_gated_forward(m, k, n, g):
    """lnrf_dense_bwd_input_gated / lnrf_dense_fwd_gated = GEMM followed by the activation backward of the layer
    below (first g columns), in one kernel; both GEMM kernels, plain and accumulating."""
    from learn_nerf import _lib as L
    from learn_nerf import ops

    gen = torch.Generator().manual_seed(m + k)
    w = (torch.randn(k, n, generator=gen) / k ** 0.5).cuda()
    gy = torch.randn(m, n, generator=gen).cuda()
    h_below = torch.relu(torch.randn(m, g, generator=gen)).cuda()  # ReLU output: ~half zeros
    want = ops.dense_bwd_input(gy, w)
    want[:, :g] = ops.act_bwd_(want[:, :g].contiguous(), h_below, L.ACT_RELU)
    got = ops.dense_bwd_input(gy, w, gate=h_below)
    assert torch.equal(got, want)
    base = torch.randn(m, k, generator=gen).cuda()
    acc = ops.dense_bwd_input(gy, w, out=base.clone(), accumulate=True, gate=h_below)
    assert torch.allclose(acc, base + want, atol=1e-5, rtol=1e-5)
    # gated forward with a tanh gate: y * (1 - t^2)
    x = torch.randn(m, k, generator=gen).cuda()
    t = torch.tanh(torch.randn(m, n, generator=gen)).cuda()
    b = torch.randn(n, generator=gen).cuda()
    y = ops.dense_fwd(x, w, b, L.ACT_NONE, gate=t, gate_act=L.ACT_TANH)
    assert torch.allclose(y, ops.dense_fwd(x, w, b, L.ACT_NONE) * (1 - t * t), atol=1e-5, rtol=1e-5)
