"""
Tensor-level wrappers over the C ABI (include/lnrf.h).  Each function allocates its outputs
with torch (device memory plumbing only) and enqueues one HIP kernel family on the current
stream.  No arithmetic happens in Python.
"""
import contextlib
import functools
from typing import Optional, Sequence, Tuple

import torch

from . import _lib as L
from . import _ws

F32 = torch.float32


def _dev(t: torch.Tensor):
    return t.device


def _ray_stride(rays: torch.Tensor) -> int:
    # [N,2,3] -> 6, [N,3,3] -> 9 (training batches keep the colour next to the ray)
    assert rays.dim() == 3 and rays.shape[2] == 3 and rays.shape[1] in (2, 3), rays.shape
    return rays.shape[1] * 3


def _seed(seed) -> int:
    return int(seed) & 0xFFFFFFFFFFFFFFFF


def ray_aabb_stratified(rays, bbox_min: Sequence[float], bbox_max: Sequence[float], count: int,
                        u: Optional[torch.Tensor] = None, seed: int = 0, stream_id: int = 0,
                        ray_offset: int = 0, min_t_range: float = 1e-3, epsilon: float = 1e-8):
    """ray_t_range + stratified_sampling (render.py:346-389, 121-143) -> t_min, t_max, mask, ts."""
    n = rays.shape[0]
    dev = _dev(rays)
    t_min = torch.empty(n, dtype=F32, device=dev)
    t_max = torch.empty(n, dtype=F32, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    ts = torch.empty((n, count), dtype=F32, device=dev)
    if u is not None:
        assert u.shape == (n, count)
    L.check(L.lib().lnrf_ray_aabb_stratified(
        L.ptr(rays), _ray_stride(rays), n, L.f3(bbox_min), L.f3(bbox_max), min_t_range, epsilon, count,
        L.ptr(u), _seed(seed), stream_id, ray_offset, L.ptr(t_min), L.ptr(t_max), L.ptr(mask, torch.uint8),
        L.ptr(ts) if count > 0 else None, L.stream()), "ray_aabb_stratified")
    return t_min, t_max, mask, ts


def camera_rays(origin, x_axis, y_axis, z_axis, x_fov: float, y_fov: float, width: int, height: int, device):
    """CameraView.bare_rays on the GPU -> [H*W, 2, 3]."""
    rays = torch.empty((width * height, 2, 3), dtype=F32, device=device)
    L.check(L.lib().lnrf_camera_rays(L.f3(origin), L.f3(x_axis), L.f3(y_axis), L.f3(z_axis), float(x_fov),
                                     float(y_fov), width, height, L.ptr(rays), L.stream()), "camera_rays")
    return rays


def gather_rows_into(src: torch.Tensor, idx: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[i, :] = src[idx[i], :] for fp32 row matrices on the GPU (the shuffled batch iterator, dataset.py:222-229)."""
    if src.dim() != 2 or out.dim() != 2 or src.shape[1] != out.shape[1] or idx.shape[0] != out.shape[0]:
        raise ValueError("gather_rows_into: shape mismatch")
    L.check(L.lib().lnrf_gather_rows(L.ptr(src), src.shape[0], src.shape[1], L.ptr(idx, torch.int32), idx.shape[0],
                                     L.ptr(out), L.stream()), "gather_rows")
    return out


def stratified(t_min, t_max, count: int, u=None, seed: int = 0, stream_id: int = 0, ray_offset: int = 0):
    n = t_min.shape[0]
    ts = torch.empty((n, count), dtype=F32, device=_dev(t_min))
    L.check(L.lib().lnrf_stratified(L.ptr(t_min), L.ptr(t_max), n, count, L.ptr(u), _seed(seed), stream_id,
                                    ray_offset, L.ptr(ts) if count > 0 else None, L.stream()), "stratified")
    return ts


def ray_points(rays, ts, want_dirs: bool = True):
    n, t = ts.shape
    pts = torch.empty((n, t, 3), dtype=F32, device=_dev(ts))
    dirs = torch.empty((n, t, 3), dtype=F32, device=_dev(ts)) if want_dirs else None
    L.check(L.lib().lnrf_ray_points(L.ptr(rays), _ray_stride(rays), L.ptr(ts), n, t, L.ptr(pts), L.ptr(dirs),
                                    L.stream()), "ray_points")
    return pts, dirs


def fine_sample(ts_c, t_min, t_max, density_c, count: int, u=None, seed: int = 0, stream_id: int = 1,
                ray_offset: int = 0, combine: bool = True, eps: float = 1e-8):
    n, tc = ts_c.shape
    width = tc + count if combine else count
    out = torch.empty((n, width), dtype=F32, device=_dev(ts_c))
    if width == 0:
        return out
    if u is not None:
        assert u.shape == (n, count)
    L.check(L.lib().lnrf_fine_sample(L.ptr(ts_c), L.ptr(t_min), L.ptr(t_max), L.ptr(density_c), n, tc, count,
                                     eps, 1 if combine else 0, L.ptr(u), _seed(seed), stream_id, ray_offset,
                                     L.ptr(out), L.stream()), "fine_sample")
    return out


def bin_edges(ts, t_min, t_max):
    n, t = ts.shape
    starts = torch.empty((n, t), dtype=F32, device=_dev(ts))
    ends = torch.empty((n, t), dtype=F32, device=_dev(ts))
    L.check(L.lib().lnrf_bin_edges(L.ptr(ts), L.ptr(t_min), L.ptr(t_max), n, t, L.ptr(starts), L.ptr(ends),
                                   L.stream()), "bin_edges")
    return starts, ends


def termination_probs(ts, t_min, t_max, density):
    n, t = ts.shape
    probs = torch.empty((n, t + 1), dtype=F32, device=_dev(ts))
    L.check(L.lib().lnrf_termination_probs(L.ptr(ts), L.ptr(t_min), L.ptr(t_max), L.ptr(density), n, t,
                                           L.ptr(probs), L.stream()), "termination_probs")
    return probs


def composite_fwd(rays, ts, t_min, t_max, mask, density, rgb, background, aux=None, targets=None,
                  sq_err=None, want_coords: bool = True):
    """-> outputs[N,3], alphas[N], coords[N,3] | None, aux_sum[N,n_aux] | None."""
    n, t = ts.shape
    dev = _dev(ts)
    n_aux = 0 if aux is None else aux.shape[-1]
    outputs = torch.empty((n, 3), dtype=F32, device=dev)
    alphas = torch.empty(n, dtype=F32, device=dev)
    coords = torch.empty((n, 3), dtype=F32, device=dev) if (want_coords and rays is not None) else None
    aux_sum = torch.empty((n, n_aux), dtype=F32, device=dev) if n_aux else None
    tstride = 0
    tptr = None
    if targets is not None:
        # targets may be a strided view batch[:, 2] of an [N,3,3] batch
        assert targets.shape == (n, 3) and targets.stride(1) == 1
        tstride = targets.stride(0)
        tptr = L.c_void_p(targets.data_ptr())
    L.check(L.lib().lnrf_composite_fwd(
        L.ptr(rays) if rays is not None else None, _ray_stride(rays) if rays is not None else 6, L.ptr(ts),
        L.ptr(t_min), L.ptr(t_max), L.ptr(mask, torch.uint8), L.ptr(density), L.ptr(rgb), L.ptr(aux), n_aux,
        L.ptr(background), n, t, L.ptr(outputs), L.ptr(alphas), L.ptr(coords), L.ptr(aux_sum), tptr, tstride,
        L.ptr(sq_err), L.stream()), "composite_fwd")
    return outputs, alphas, coords, aux_sum


def composite_bwd(ts, t_min, t_max, mask, density, rgb, background, g_background, g_out=None, outputs=None,
                  targets=None, out_scale: float = 0.0, aux=None, g_aux_w: Sequence[float] = ()):
    """-> g_density[N,T], g_rgb[N,T,3], g_aux | None; accumulates into g_background[3]."""
    n, t = ts.shape
    dev = _dev(ts)
    n_aux = 0 if aux is None else aux.shape[-1]
    g_density = torch.empty((n, t), dtype=F32, device=dev)
    g_rgb = torch.empty((n, t, 3), dtype=F32, device=dev)
    g_aux = torch.empty((n, t, n_aux), dtype=F32, device=dev) if n_aux else None
    tstride, tptr = 0, None
    if targets is not None:
        assert targets.shape == (n, 3) and targets.stride(1) == 1
        tstride = targets.stride(0)
        tptr = L.c_void_p(targets.data_ptr())
    gw = (L.c_float * 4)(*([float(x) for x in g_aux_w] + [0.0] * (4 - len(g_aux_w))))
    # fixed-order background gradient (bit-reproducible step): per-workgroup partial sums in a leased scratch block
    nbytes = L.lib().lnrf_composite_bwd_scratch_bytes(n)
    lease = _ws.lease("composite_bwd", nbytes, dev)
    L.check(L.lib().lnrf_composite_bwd_det(
        L.ptr(ts), L.ptr(t_min), L.ptr(t_max), L.ptr(mask, torch.uint8), L.ptr(density), L.ptr(rgb),
        L.ptr(aux), n_aux, L.ptr(background), n, t, L.ptr(g_out), L.ptr(outputs), tptr, tstride,
        float(out_scale), gw, L.ptr(g_density), L.ptr(g_rgb), L.ptr(g_aux), L.ptr(g_background),
        L.ptr(lease.buf, torch.uint8), nbytes, L.stream()), "composite_bwd_det")
    lease.release()
    return g_density, g_rgb, g_aux


# ---------------------------------------------------------------- generic dense (exact fp32)

_DENSE_PRECISIONS = {"fp32": 0, "bf16": 1}  # LNRF_DENSE_FP32 / LNRF_DENSE_BF16


@contextlib.contextmanager
def dense_precision(precision: str):
    """Operand precision of the generic dense kernels inside the block (lnrf_set_dense_precision, thread-local)."""
    if precision not in _DENSE_PRECISIONS:
        raise ValueError(f"unknown precision {precision!r}")
    lib = L.lib()
    previous = lib.lnrf_get_dense_precision()
    L.check(lib.lnrf_set_dense_precision(_DENSE_PRECISIONS[precision]), "set_dense_precision")
    try:
        yield
    finally:
        lib.lnrf_set_dense_precision(previous)


def uses_model_precision(method):
    """Run a model method with the dense kernels in the model's `precision` ("bf16" | "fp32")."""
    @functools.wraps(method)
    def wrapped(self, *args, **kwargs):
        with dense_precision(getattr(self, "precision", "fp32")):
            return method(self, *args, **kwargs)
    return wrapped


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "need a row-major 2-D view"
    return t.stride(0)


def _vptr(t: torch.Tensor):
    """pointer of a possibly column-sliced row-major view"""
    assert t.is_cuda and t.dtype == F32
    return L.c_void_p(t.data_ptr())


def sinusoidal_emb_into(x: torch.Tensor, freqs: int, out: torch.Tensor, col_off: int = 0):
    m, dims = x.shape
    L.check(L.lib().lnrf_sinusoidal_emb(_vptr(x), _ld(x), m, dims, freqs, _vptr(out), _ld(out), col_off,
                                        L.stream()), "sinusoidal_emb")
    return out


def dense_fwd(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], act: int,
              out: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
              gate_act: int = L.ACT_RELU) -> torch.Tensor:
    """act(x @ w + b); with `gate` ([m, n], an activation OUTPUT) the result is multiplied by gate_act'(gate)."""
    m, k = x.shape
    k2, n = w.shape
    assert k == k2 and w.is_contiguous()
    if out is None:
        out = torch.empty((m, n), dtype=F32, device=_dev(x))
    if gate is None:
        L.check(L.lib().lnrf_dense_fwd(_vptr(x), _ld(x), L.ptr(w), L.ptr(b), act, _vptr(out), _ld(out), m, k, n,
                                       L.stream()), "dense_fwd")
    else:
        assert gate.shape == (m, n)
        L.check(L.lib().lnrf_dense_fwd_gated(_vptr(x), _ld(x), L.ptr(w), L.ptr(b), act, _vptr(gate), _ld(gate),
                                             gate_act, _vptr(out), _ld(out), m, k, n, L.stream()), "dense_fwd_gated")
    return out


def act_bwd_(g: torch.Tensor, y: torch.Tensor, act: int) -> torch.Tensor:
    m, n = g.shape
    L.check(L.lib().lnrf_act_bwd(_vptr(g), _ld(g), _vptr(y), _ld(y), act, m, n, L.stream()), "act_bwd")
    return g


def dense_bwd_input(gy: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None,
                    accumulate: bool = False, gate: Optional[torch.Tensor] = None,
                    gate_act: int = L.ACT_RELU) -> torch.Tensor:
    """
    gy @ w.T (added to `out` if accumulate).  With `gate` ([m, g <= k], the OUTPUT of the activation that
    produced the first g input columns) those columns are multiplied by gate_act'(gate): the activation backward
    of the layer below fused into the GEMM (lnrf_dense_bwd_input_gated).
    """
    m, n = gy.shape
    k, n2 = w.shape
    assert n == n2
    if out is None:
        out = torch.empty((m, k), dtype=F32, device=_dev(gy))
        accumulate = False
    if gate is None:
        L.check(L.lib().lnrf_dense_bwd_input(_vptr(gy), _ld(gy), L.ptr(w), _vptr(out), _ld(out),
                                             1 if accumulate else 0, m, k, n, L.stream()), "dense_bwd_input")
    else:
        assert gate.shape[0] == m and gate.shape[1] <= k
        L.check(L.lib().lnrf_dense_bwd_input_gated(_vptr(gy), _ld(gy), L.ptr(w), _vptr(gate), _ld(gate), gate_act,
                                                   gate.shape[1], _vptr(out), _ld(out), 1 if accumulate else 0, m,
                                                   k, n, L.stream()), "dense_bwd_input_gated")
    return out


def dense_bwd_weight(x: torch.Tensor, gy: torch.Tensor, gw: torch.Tensor, gb: Optional[torch.Tensor]):
    m, k = x.shape
    m2, n = gy.shape
    assert m == m2 and gw.shape == (k, n) and gw.is_contiguous()
    # fixed-order split reduction (bit-reproducible), partial sums in a leased scratch block
    nbytes = L.lib().lnrf_dense_bwd_weight_scratch_bytes(m, k, n)
    lease = _ws.lease("dense_wgrad", nbytes, _dev(x))
    L.check(L.lib().lnrf_dense_bwd_weight_det(_vptr(x), _ld(x), _vptr(gy), _ld(gy), L.ptr(gw), L.ptr(gb), m, k, n,
                                              L.ptr(lease.buf, torch.uint8), nbytes, L.stream()), "dense_bwd_weight_det")
    lease.release()


def bias_grad(gy: torch.Tensor, gb: torch.Tensor):
    m, n = gy.shape
    nbytes = L.lib().lnrf_dense_bwd_weight_scratch_bytes(m, 0, n)
    lease = _ws.lease("dense_wgrad", nbytes, _dev(gy))
    L.check(L.lib().lnrf_dense_bwd_weight_det(None, 0, _vptr(gy), _ld(gy), None, L.ptr(gb), m, 0, n,
                                              L.ptr(lease.buf, torch.uint8), nbytes, L.stream()), "bias_grad")
    lease.release()


def gemm(a: torch.Tensor, sa_i: int, sa_r: int, b: torch.Tensor, sb_r: int, sb_j: int, c: torch.Tensor, ldc: int,
         i_rows: int, j_cols: int, r_depth: int, bias=None, act: int = 0, mode: int = 0, splits: int = 0):
    """C[i*ldc+j] (op)= sum_r A[i*sa_i + r*sa_r] * B[r*sb_r + j*sb_j] (see lnrf_gemm_f32).  mode 2 (split reduction) with
    contiguous C rows takes the fixed-order form lnrf_gemm_f32_det (bit-reproducible) instead of fp32 atomics."""
    if mode == 2 and ldc == j_cols and bias is None and splits == 0:
        nbytes = L.lib().lnrf_gemm_f32_det_scratch_bytes(i_rows, j_cols, r_depth)
        lease = _ws.lease("gemm_det", nbytes, _dev(a))
        L.check(L.lib().lnrf_gemm_f32_det(_vptr(a), sa_i, sa_r, _vptr(b), sb_r, sb_j, _vptr(c), i_rows, j_cols, r_depth,
                                          L.ptr(lease.buf, torch.uint8), nbytes, L.stream()), "gemm_f32_det")
        lease.release()
        return c
    L.check(L.lib().lnrf_gemm_f32(_vptr(a), sa_i, sa_r, _vptr(b), sb_r, sb_j, _vptr(c), ldc, L.ptr(bias), act, mode,
                                  i_rows, j_cols, r_depth, splits, L.stream()), "gemm_f32")
    return c


def hashgrid_fwd(desc, tables: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    import ctypes

    m = x.shape[0]
    enc_t = torch.empty((desc.n_levels * desc.feature_dim, m), dtype=F32, device=_dev(x))
    L.check(L.lib().lnrf_hashgrid_fwd(ctypes.byref(desc), L.ptr(tables), L.ptr(x), m, L.ptr(enc_t), L.stream()),
            "hashgrid_fwd")
    return enc_t


def hashgrid_jvp(desc, tables: torch.Tensor, x: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """(d enc / d x) u in the feature-major layout [L*F, M]"""
    import ctypes

    m = x.shape[0]
    out = torch.empty((desc.n_levels * desc.feature_dim, m), dtype=F32, device=_dev(x))
    L.check(L.lib().lnrf_hashgrid_jvp(ctypes.byref(desc), L.ptr(tables), L.ptr(x), L.ptr(u), m, L.ptr(out),
                                      L.stream()), "hashgrid_jvp")
    return out


def hashgrid_input_grad(desc, tables: torch.Tensor, x: torch.Tensor, g_enc_t: torch.Tensor) -> torch.Tensor:
    """(d enc / d x)^T g_enc -> [M, 3]"""
    import ctypes

    m = x.shape[0]
    g_x = torch.empty((m, 3), dtype=F32, device=_dev(x))
    L.check(L.lib().lnrf_hashgrid_input_grad(ctypes.byref(desc), L.ptr(tables), L.ptr(x), m, L.ptr(g_enc_t),
                                             L.ptr(g_x), L.stream()), "hashgrid_input_grad")
    return g_x


def hashgrid_bwd_dir(desc, x: torch.Tensor, u: torch.Tensor, g_enc_t: torch.Tensor, g_tables: torch.Tensor):
    import ctypes

    hashgrid_bwd(desc, x, g_enc_t, g_tables, u=u)


def hashgrid_bwd(desc, x: torch.Tensor, g_enc_t: torch.Tensor, g_tables: torch.Tensor, u=None, level_absmax=None):
    """g_tables += scatter(g_enc_t); hashed levels go through the bucketed (atomic-free) path.  level_absmax
    (u is None only): per-level upper bound of |g_enc_t| from the kernel that wrote it (lnrf_ngp_mlp_bwd)."""
    import ctypes

    m = x.shape[0]
    nbytes = L.lib().lnrf_hashgrid_bwd_scratch_bytes(ctypes.byref(desc), m)
    lease = _ws.lease("hashgrid_bwd", max(int(nbytes), 16), _dev(x))
    scratch = lease.buf
    if level_absmax is not None and (u is not None or level_absmax.numel() < desc.n_levels):
        raise ValueError("level_absmax: one float per level, plain scatter only")
    L.check(L.lib().lnrf_hashgrid_bwd_bucketed(ctypes.byref(desc), L.ptr(x), L.ptr(u), m, L.ptr(g_enc_t),
                                               L.ptr(level_absmax), L.ptr(g_tables), L.ptr(scratch, torch.uint8),
                                               int(nbytes), L.stream()),
            "hashgrid_bwd_bucketed")


# ---------------------------------------------------------------- Ref-NeRF pieces

def sinusoidal_emb_bwd(x: torch.Tensor, freqs: int, g_emb: torch.Tensor, col_off: int = 0) -> torch.Tensor:
    m, dims = x.shape
    g_x = torch.empty((m, dims), dtype=F32, device=_dev(x))
    L.check(L.lib().lnrf_sinusoidal_emb_bwd(_vptr(x), _ld(x), m, dims, freqs, _vptr(g_emb), _ld(g_emb), col_off,
                                            L.ptr(g_x), L.stream()), "sinusoidal_emb_bwd")
    return g_x


def sinusoidal_emb_jvp_into(x: torch.Tensor, freqs: int, u: torch.Tensor, out: torch.Tensor, col_off: int = 0):
    m, dims = x.shape
    L.check(L.lib().lnrf_sinusoidal_emb_jvp(_vptr(x), _ld(x), m, dims, freqs, L.ptr(u), _vptr(out), _ld(out), col_off,
                                            L.stream()), "sinusoidal_emb_jvp")
    return out


def integrated_directional_encoding(sh_degree: int, coords: torch.Tensor, roughness) -> torch.Tensor:
    m = coords.shape[0]
    out = torch.empty((m, sh_degree * sh_degree), dtype=F32, device=_dev(coords))
    L.check(L.lib().lnrf_integrated_directional_encoding(sh_degree, L.ptr(coords), L.ptr(roughness), m, L.ptr(out),
                                                         L.stream()), "integrated_directional_encoding")
    return out


def refnerf_head_fwd(spatial: torch.Tensor, nraw, d, sh_degree: int, tail: torch.Tensor):
    m = spatial.shape[0]
    dev = _dev(spatial)
    density = torch.empty(m, dtype=F32, device=dev)
    diffuse = torch.empty((m, 3), dtype=F32, device=dev)
    spectral = torch.empty(m, dtype=F32, device=dev)
    aux = torch.empty((m, 2), dtype=F32, device=dev)
    L.check(L.lib().lnrf_refnerf_head_fwd(_vptr(spatial), _ld(spatial), L.ptr(nraw), L.ptr(d), m, sh_degree,
                                          L.ptr(density), L.ptr(diffuse), L.ptr(spectral), _vptr(tail), _ld(tail),
                                          L.ptr(aux), L.stream()), "refnerf_head_fwd")
    return density, diffuse, spectral, aux


def refnerf_head_bwd(spatial, nraw, d, sh_degree: int, g_density, g_diffuse, g_spectral, g_tail, g_aux, g_spatial):
    m = spatial.shape[0]
    g_nraw = torch.empty((m, 3), dtype=F32, device=_dev(spatial))
    L.check(L.lib().lnrf_refnerf_head_bwd(_vptr(spatial), _ld(spatial), L.ptr(nraw), L.ptr(d), m, sh_degree,
                                          L.ptr(g_density), L.ptr(g_diffuse), L.ptr(g_spectral), _vptr(g_tail),
                                          _ld(g_tail), L.ptr(g_aux), _vptr(g_spatial), _ld(g_spatial), L.ptr(g_nraw),
                                          L.stream()), "refnerf_head_bwd")
    return g_nraw


def refnerf_color_fwd(dir_out, spectral, diffuse):
    m = dir_out.shape[0]
    rgb = torch.empty((m, 3), dtype=F32, device=_dev(dir_out))
    L.check(L.lib().lnrf_refnerf_color_fwd(L.ptr(dir_out), L.ptr(spectral), L.ptr(diffuse), m, L.ptr(rgb), L.stream()),
            "refnerf_color_fwd")
    return rgb


def refnerf_color_bwd(dir_out, spectral, diffuse, g_rgb):
    m = dir_out.shape[0]
    dev = _dev(dir_out)
    g_do = torch.empty((m, 3), dtype=F32, device=dev)
    g_sp = torch.empty(m, dtype=F32, device=dev)
    g_df = torch.empty((m, 3), dtype=F32, device=dev)
    L.check(L.lib().lnrf_refnerf_color_bwd(L.ptr(dir_out), L.ptr(spectral), L.ptr(diffuse), m, L.ptr(g_rgb),
                                           L.ptr(g_do), L.ptr(g_sp), L.ptr(g_df), L.stream()), "refnerf_color_bwd")
    return g_do, g_sp, g_df


# ---------------------------------------------------------------- optimiser

def adam_step_(p, g, m, v, lr, b1, b2, eps, step: int, grad_scale: float = 1.0, sq_norms=None):
    """Fused Adam over flat buffers; with sq_norms (2 floats, pre-zeroed) the same pass accumulates sum g^2 (g as
    passed in) and sum p^2 (before the update) for the step's log."""
    if sq_norms is None:
        L.check(L.lib().lnrf_adam_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), lr, b1, b2, eps, step,
                                       grad_scale, L.stream()), "adam_step")
    else:
        L.check(L.lib().lnrf_adam_step_norms(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), lr, b1, b2, eps, step,
                                             grad_scale, L.ptr(sq_norms), L.stream()), "adam_step_norms")


def step_log(sums: torch.Tensor, inv_count: float, grad_scale: float, clear: bool = True) -> torch.Tensor:
    """[sum sq err coarse, sum sq err fine, sum g^2, sum p^2] -> [coarse loss, fine loss, grad_norm, param_norm]."""
    out = torch.empty(4, dtype=F32, device=_dev(sums))
    L.check(L.lib().lnrf_step_log(L.ptr(sums), inv_count, grad_scale, 1 if clear else 0, L.ptr(out), L.stream()),
            "step_log")
    return out


def sq_norm_into(x: torch.Tensor, out: torch.Tensor):
    """out (1-element fp32, pre-zeroed by the caller) += sum(x^2)"""
    L.check(L.lib().lnrf_sq_norm(L.ptr(x), x.numel(), L.ptr(out), L.stream()), "sq_norm")
    return out
