"""
learn_nerf.instant_ngp — InstantNGPModel, MultiresHashTableEncoding, HashTableEncoding,
hash_table_lookup (reference: learn_nerf/instant_ngp.py).

The multiresolution hash-grid gather / scatter-add are HIP kernels (csrc/hashgrid.hip, level-major
sweep so one level's table stays L2-resident); the tiny MLP is the fused bf16 MFMA kernel of
csrc/ngp_mlp.hip (precision="bf16", default) or the exact-fp32 strided GEMM (precision="fp32").
Parameter tree names follow Flax: MultiresHashTableEncoding_0/HashTableEncoding_{l}/table, Dense_0..4.
"""
from dataclasses import dataclass, field
from typing import Any, Dict, List, Sequence, Tuple

import ctypes
from collections import OrderedDict

import torch

from . import _lib as L
from . import _prof
from . import _ws
from . import ops
from .model import ModelBase
from .params import lecun_normal_

F32 = torch.float32
POISON_SCRATCH = False  # tests set this: scratch buffers are filled with 0xFF bytes (NaN) before the kernels run


def _level_rows(grid_size: int, table_size: int) -> Tuple[int, bool]:
    hashed = grid_size ** 3 > table_size  # instant_ngp.py:178
    return (table_size if hashed else grid_size ** 3), hashed


def hash_table_lookup(table: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """
    Lookup integer coordinates [N x 3] in a hash table [T x F] (instant_ngp.py:211-224); uint32
    wrap-around hash (x ^ 19349663*y ^ 83492791*z) mod T.  Host-level utility (index arithmetic only).
    """
    c = coords.to(torch.int64) & 0xFFFFFFFF
    idx = (c[:, 0] ^ ((19_349_663 * c[:, 1]) & 0xFFFFFFFF) ^ ((83_492_791 * c[:, 2]) & 0xFFFFFFFF)) & 0xFFFFFFFF
    return table[idx % table.shape[0]]


@dataclass
class MultiresHashTableEncoding:
    """
    Encode spatial coordinates with a multiresolution hash table (instant_ngp.py:92-118).
    apply(tables_flat, x[N,3]) -> [N, L*F].
    """

    table_sizes: Sequence[int]
    grid_sizes: Sequence[int]
    bbox_min: Sequence[float]
    bbox_max: Sequence[float]
    feature_dim: int = 2
    smooth: bool = False

    def rows(self) -> List[int]:
        return [_level_rows(g, t)[0] for t, g in zip(self.table_sizes, self.grid_sizes)]

    def num_table_floats(self) -> int:
        return sum(r * self.feature_dim for r in self.rows())

    def desc(self) -> L.HashGridDesc:
        d = L.HashGridDesc()
        d.n_levels, d.feature_dim, d.smooth = len(self.grid_sizes), self.feature_dim, int(self.smooth)
        for a in range(3):
            d.bbox_min[a] = float(self.bbox_min[a])
            d.bbox_max[a] = float(self.bbox_max[a])
        off = 0
        for l, (t, g) in enumerate(zip(self.table_sizes, self.grid_sizes)):
            rows, hashed = _level_rows(g, t)
            d.grid_size[l], d.table_size[l], d.table_offset[l], d.hashed[l] = g, rows, off, int(hashed)
            off += rows * self.feature_dim
        return d

    def encode_t(self, tables_flat: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """feature-major encoding [L*F, N] (the layout the kernels use)"""
        return ops.hashgrid_fwd(self.desc(), tables_flat, x.contiguous())

    def apply(self, tables_flat: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        return self.encode_t(tables_flat, x).t().contiguous()


@dataclass
class HashTableEncoding(MultiresHashTableEncoding):
    """Single-level encoding (instant_ngp.py:121-208): HashTableEncoding(table_size, grid_size, bbox...)."""

    def __init__(self, table_size: int, grid_size: int, bbox_min, bbox_max, feature_dim: int = 2,
                 smooth: bool = False):
        super().__init__([table_size], [grid_size], bbox_min, bbox_max, feature_dim, smooth)


@dataclass
class InstantNGPModel(ModelBase):
    """A NeRF model that utilizes a multilevel hash table (instant_ngp.py:16-54)."""

    # TrainLoop: keep the coarse backward on the main stream — this model's gather / scatter / persistent MLP kernels
    # all contend for the same L2 and LDS, running two of them side by side measured slightly slower
    overlap_backward_hint = False

    table_sizes: Sequence[int] = None
    grid_sizes: Sequence[int] = None
    bbox_min: Sequence[float] = None
    bbox_max: Sequence[float] = None
    table_feature_dim: int = 2
    table_smooth: bool = False
    d_freqs: int = 4
    hidden_dim: int = 64
    density_dim: int = 16
    density_layers: int = 1
    color_layers: int = 2
    precision: str = "bf16"  # "bf16" (fused MFMA MLP, lnrf_ngp_mlp_*) | "fp32" (exact dense path)
    # fused path, forward without a backward (rendering, evaluation, model.apply): "bf16x3" = split-precision kernel
    # (lnrf_ngp_mlp_fwd_split: the reference's fp32 arithmetic to ~1e-5) | "bf16" = the training forward's plain operands
    render_precision: str = "bf16x3"
    tag: str = "ngp"

    _pack_cache: Any = field(default=None, repr=False, compare=False)
    _pack_generation: int = field(default=0, repr=False, compare=False)

    def invalidate_packed(self) -> None:
        """Call after the parameters were modified outside torch (e.g. by lnrf_adam_step)."""
        self._pack_generation += 1

    def encoding(self) -> MultiresHashTableEncoding:
        return MultiresHashTableEncoding(self.table_sizes, self.grid_sizes, self.bbox_min, self.bbox_max,
                                         self.table_feature_dim, self.table_smooth)

    # ---- fused bf16 MLP ---------------------------------------------------------------------------
    def fused_supported(self) -> bool:
        return ((self.hidden_dim, self.density_dim, self.density_layers, self.color_layers, self.d_freqs)
                == (64, 16, 1, 2, 4) and len(self.grid_sizes) * self.table_feature_dim <= 32)

    def _use_fused(self) -> bool:
        if self.precision not in ("bf16", "fp32"):
            raise ValueError(f"unknown precision {self.precision!r}")
        return self.precision == "bf16" and self.fused_supported()

    def _mlp_desc(self) -> L.NgpMlpDesc:
        return L.NgpMlpDesc(len(self.grid_sizes) * self.table_feature_dim, self.hidden_dim, self.density_dim,
                            self.density_layers, self.color_layers, self.d_freqs,
                            self.encoding().num_table_floats())

    def packed_weights(self, flat: torch.Tensor, kind: str = "bf16") -> torch.Tensor:
        """bf16 MFMA-fragment copy of the Dense parameters ("bf16": forward + transposed streams; "split": hi/lo pairs
        of the forward stream for the render kernel); rebuilt when the flat buffer changes.  As in
        NeRFModel.packed_weights: a miss packs into a fresh buffer (a saved backward context may hold the old
        one) and cache entries keep their source tensor alive so that a recycled address cannot hit."""
        if self._pack_cache is None:
            self._pack_cache = OrderedDict()
        key = (kind, flat.data_ptr(), flat.numel(), flat._version, str(flat.device), self._pack_generation)
        hit = self._pack_cache.get(key)
        if hit is not None:
            self._pack_cache.move_to_end(key)
            return hit[1]
        desc = self._mlp_desc()
        if kind == "split":
            nbytes = L.lib().lnrf_ngp_mlp_packed_split_bytes(ctypes.byref(desc))
            packed = torch.empty(nbytes, dtype=torch.uint8, device=flat.device)
            L.check(L.lib().lnrf_ngp_mlp_pack_split(ctypes.byref(desc), L.ptr(flat), L.ptr(packed, torch.uint8),
                                                    L.stream()), "ngp_mlp_pack_split")
        else:
            nbytes = L.lib().lnrf_ngp_mlp_packed_bytes(ctypes.byref(desc))
            packed = torch.empty(nbytes, dtype=torch.uint8, device=flat.device)
            L.check(L.lib().lnrf_ngp_mlp_pack(ctypes.byref(desc), L.ptr(flat), L.ptr(packed, torch.uint8), L.stream()),
                    "ngp_mlp_pack")
        self._pack_cache[key] = (flat, packed)
        while len(self._pack_cache) > 6:
            self._pack_cache.popitem(last=False)
        return packed

    def _fused_forward_points(self, flat, x, d, save: bool):
        enc = self.encoding()
        tables, _ = self._dense_views(flat)
        m, dev = x.shape[0], flat.device
        desc = self._mlp_desc()
        if self.render_precision not in ("bf16x3", "bf16"):
            raise ValueError(f"unknown render_precision {self.render_precision!r}")
        with _prof.section(f"{self.tag}_hashgrid_fwd"):
            enc_t = enc.encode_t(tables, x)  # [L*F, M]
        density = torch.empty(m, dtype=F32, device=dev)
        rgb = torch.empty((m, 3), dtype=F32, device=dev)
        d = d.contiguous()
        if not save and self.render_precision == "bf16x3":
            packed3 = self.packed_weights(flat, "split")
            with _prof.section(f"{self.tag}_mlp_fwd_split"):
                L.check(L.lib().lnrf_ngp_mlp_fwd_split(ctypes.byref(desc), L.ptr(packed3, torch.uint8), L.ptr(enc_t),
                                                       L.ptr(d), m, L.ptr(density), L.ptr(rgb), L.stream()),
                        "ngp_mlp_fwd_split")
            return density, rgb, {}, None
        packed = self.packed_weights(flat)
        with _prof.section(f"{self.tag}_mlp_fwd"):
            L.check(L.lib().lnrf_ngp_mlp_fwd(ctypes.byref(desc), L.ptr(packed, torch.uint8), L.ptr(enc_t), L.ptr(d),
                                             m, L.ptr(density), L.ptr(rgb), L.stream()), "ngp_mlp_fwd")
        ctx = dict(kind="fused", packed=packed, x=x, d=d, enc_t=enc_t) if save else None
        return density, rgb, {}, ctx

    def _fused_backward(self, ctx, g_density, g_rgb, grad_flat):
        enc = self.encoding()
        g_tables, _ = self._dense_views(grad_flat)
        desc = self._mlp_desc()
        m, dev = ctx["x"].shape[0], grad_flat.device
        lf = desc.enc_dim
        with _prof.section(f"{self.tag}_mlp_bwd"):
            lease = _ws.lease("ngp_scratch", L.lib().lnrf_ngp_mlp_scratch_bytes(ctypes.byref(desc), m), dev)
            scratch = lease.buf
            if POISON_SCRATCH:  # tests: every word the kernels read back must have been written by them
                scratch.fill_(0xFF)
            glease = _ws.lease("ngp_g_enc", lf * m * 4, dev)
            g_enc_t = glease.buf.view(F32).view(lf, m)
            level_absmax = torch.zeros(lf // 2, dtype=F32, device=dev)  # the scatter's fixed-point scale per level
            gd = g_density.reshape(-1).contiguous()
            gr = g_rgb.reshape(-1, 3).contiguous()
            L.check(L.lib().lnrf_ngp_mlp_bwd(
                ctypes.byref(desc), L.ptr(ctx["packed"], torch.uint8), L.ptr(ctx["enc_t"]), L.ptr(ctx["d"]),
                L.ptr(gd), L.ptr(gr), m, L.ptr(scratch, torch.uint8), L.ptr(g_enc_t), L.ptr(level_absmax),
                L.ptr(grad_flat), L.stream()), "ngp_mlp_bwd")
        with _prof.section(f"{self.tag}_hashgrid_bwd"):
            ops.hashgrid_bwd(enc.desc(), ctx["x"], g_enc_t, g_tables, level_absmax=level_absmax)

    def dense_dims(self) -> List[Tuple[int, int]]:
        dims, fan = [], len(self.grid_sizes) * self.table_feature_dim
        for _ in range(self.density_layers):
            dims.append((fan, self.hidden_dim))
            fan = self.hidden_dim
        dims.append((fan, self.density_dim))
        fan = 6 * self.d_freqs + self.density_dim
        for _ in range(self.color_layers):
            dims.append((fan, self.hidden_dim))
            fan = self.hidden_dim
        dims.append((fan, 3))
        return dims

    def param_spec(self):
        spec = []
        for l, r in enumerate(self.encoding().rows()):
            spec.append((f"MultiresHashTableEncoding_0/HashTableEncoding_{l}", "table", (r, self.table_feature_dim)))
        for i, (fi, fo) in enumerate(self.dense_dims()):
            spec.append((f"Dense_{i}", "kernel", (fi, fo)))
            spec.append((f"Dense_{i}", "bias", (fo,)))
        return spec

    def init_flat_(self, flat, gen):
        nt = self.encoding().num_table_floats()
        u = torch.rand(nt, generator=gen, dtype=torch.float32)
        flat[:nt] = 1e-4 * (u * 2 - 1)  # instant_ngp.py:181-185
        off = nt
        for fi, fo in self.dense_dims():
            lecun_normal_(flat[off:off + fi * fo].view(fi, fo), fi, gen)
            off += fi * fo + fo

    def _dense_views(self, flat):
        nt = self.encoding().num_table_floats()
        out, off = [], nt
        for fi, fo in self.dense_dims():
            k = flat[off:off + fi * fo].view(fi, fo)
            off += fi * fo
            b = flat[off:off + fo]
            off += fo
            out.append((k, b))
        return flat[:nt], out

    @ops.uses_model_precision
    def forward_points(self, flat, x, d, save: bool):
        if self._use_fused():
            return self._fused_forward_points(flat, x, d, save)
        enc = self.encoding()
        tables, W = self._dense_views(flat)
        m, dev = x.shape[0], flat.device
        lf = len(self.grid_sizes) * self.table_feature_dim
        de_w = 6 * self.d_freqs
        with _prof.section(f"{self.tag}_hashgrid_fwd"):
            enc_t = enc.encode_t(tables, x)  # [L*F, M]
        with _prof.section(f"{self.tag}_mlp_fwd"):
            acts = []
            li = 0
            h = torch.empty((m, self.hidden_dim), dtype=F32, device=dev)
            ops.gemm(enc_t, 1, m, W[0][0], self.hidden_dim, 1, h, self.hidden_dim, m, self.hidden_dim, lf,
                     bias=W[0][1], act=L.ACT_RELU)  # Dense_0 on the feature-major encoding
            acts.append(h)
            li = 1
            for _ in range(self.density_layers - 1):
                h = ops.dense_fwd(h, W[li][0], W[li][1], L.ACT_RELU)
                acts.append(h)
                li += 1
            cat = torch.empty((m, de_w + self.density_dim), dtype=F32, device=dev)  # [d_emb, out] (:50)
            out = ops.dense_fwd(h, W[li][0], W[li][1], L.ACT_NONE, out=cat[:, de_w:])
            li += 1
            ops.sinusoidal_emb_into(d, self.d_freqs, cat, 0)
            e0 = torch.zeros((self.density_dim, 1), dtype=F32, device=dev)
            e0[0, 0] = 1.0
            density = ops.dense_fwd(out, e0, None, L.ACT_EXP)  # exp(out[:, :1]) (:49), unclamped
            c = cat
            cacts = []
            for _ in range(self.color_layers):
                c = ops.dense_fwd(c, W[li][0], W[li][1], L.ACT_RELU)
                cacts.append(c)
                li += 1
            rgb = ops.dense_fwd(c, W[li][0], W[li][1], L.ACT_TANH)
        ctx = None
        if save:
            ctx = dict(flat=flat, x=x, enc_t=enc_t, acts=acts, cat=cat, cacts=cacts, density=density, rgb=rgb, e0=e0)
        return density.view(-1), rgb, {}, ctx

    @ops.uses_model_precision
    def backward(self, ctx, g_density, g_rgb, g_aux, grad_flat):
        if ctx.get("kind") == "fused":
            return self._fused_backward(ctx, g_density, g_rgb, grad_flat)
        enc = self.encoding()
        tables, W = self._dense_views(ctx["flat"])
        g_tables, G = self._dense_views(grad_flat)
        m = ctx["x"].shape[0]
        lf = len(self.grid_sizes) * self.table_feature_dim
        de_w = 6 * self.d_freqs
        cat, cacts, acts = ctx["cat"], ctx["cacts"], ctx["acts"]
        with _prof.section(f"{self.tag}_mlp_bwd"):
            li = len(W) - 1
            gy = ops.act_bwd_(g_rgb.reshape(-1, 3).clone(), ctx["rgb"], L.ACT_TANH)
            for i in reversed(range(self.color_layers)):
                ops.dense_bwd_weight(cacts[i], gy, G[li][0], G[li][1])
                gy = ops.dense_bwd_input(gy, W[li][0], gate=cacts[i])
                li -= 1
            ops.dense_bwd_weight(cat, gy, G[li][0], G[li][1])
            gcat = ops.dense_bwd_input(gy, W[li][0])  # [M, 24 + density_dim]
            g_out = gcat[:, de_w:]
            gd = ops.act_bwd_(g_density.reshape(-1, 1).clone(), ctx["density"], L.ACT_EXP)
            ops.dense_bwd_input(gd, ctx["e0"], out=g_out, accumulate=True)  # d exp(out0) / d out0
            li -= 1
            h = acts[-1]
            ops.dense_bwd_weight(h, g_out, G[li][0], G[li][1])
            gy = ops.dense_bwd_input(g_out, W[li][0], gate=h)
            li -= 1
            for i in reversed(range(1, self.density_layers)):
                ops.dense_bwd_weight(acts[i - 1], gy, G[li][0], G[li][1])
                gy = ops.dense_bwd_input(gy, W[li][0], gate=acts[i - 1])
                li -= 1
            # Dense_0: input is the feature-major encoding
            hd = self.hidden_dim
            ops.gemm(ctx["enc_t"], m, 1, gy, hd, 1, G[0][0], hd, lf, hd, m, mode=2)  # gW0 += enc^T gy
            ops.bias_grad(gy, G[0][1])
            g_enc_t = torch.empty((lf, m), dtype=F32, device=grad_flat.device)
            ops.gemm(W[0][0], hd, 1, gy, 1, hd, g_enc_t, m, lf, m, hd)  # g_enc^T = W0 gy^T
        with _prof.section(f"{self.tag}_hashgrid_bwd"):
            ops.hashgrid_bwd(enc.desc(), ctx["x"], g_enc_t, g_tables)


@dataclass
class InstantNGPRefNERFModel(ModelBase):
    """
    A Ref-NeRF head on a smooth multilevel hash table (instant_ngp.py:57-89).  The spatial block is
    smooth-hash-grid -> Dense(hidden) relu -> Dense(density_dim); its first 9 outputs are the Ref-NeRF heads
    and the whole vector feeds the directional block together with the IDE and -d.n.  The analytic normal
    needs d enc / d x (lnrf_hashgrid_input_grad) and, because it is trained through, the second-order terms
    lnrf_hashgrid_jvp / lnrf_hashgrid_bwd_dir.
    """

    sh_degree: int = 4
    table_sizes: Sequence[int] = None
    grid_sizes: Sequence[int] = None
    bbox_min: Sequence[float] = None
    bbox_max: Sequence[float] = None
    table_feature_dim: int = 2
    d_freqs: int = 4  # unused, as in the reference
    hidden_dim: int = 64
    density_dim: int = 16
    density_layers: int = 1
    color_layers: int = 2
    precision: str = "bf16"  # operands of the Dense layers: "bf16" (MFMA rate) | "fp32" (exact, parity gate)
    tag: str = "ngpref"

    def encoding(self) -> MultiresHashTableEncoding:
        return MultiresHashTableEncoding(self.table_sizes, self.grid_sizes, self.bbox_min, self.bbox_max,
                                         self.table_feature_dim, True)

    def dense_dims(self) -> List[Tuple[int, int]]:
        dims, fan = [], len(self.grid_sizes) * self.table_feature_dim
        for _ in range(self.density_layers):
            dims.append((fan, self.hidden_dim))
            fan = self.hidden_dim
        dims.append((fan, self.density_dim))
        fan = self.density_dim + self.sh_degree ** 2 + 1
        for _ in range(self.color_layers):
            dims.append((fan, self.hidden_dim))
            fan = self.hidden_dim
        dims.append((fan, 3))
        return dims

    param_spec = InstantNGPModel.param_spec
    init_flat_ = InstantNGPModel.init_flat_
    _dense_views = InstantNGPModel._dense_views

    @ops.uses_model_precision
    def forward_points(self, flat, x, d, save: bool):
        if self.density_layers != 1:
            raise NotImplementedError("InstantNGPRefNERFModel: density_layers != 1")
        enc = self.encoding()
        desc = enc.desc()
        tables, W = self._dense_views(flat)
        m, dev, hd, dd = x.shape[0], flat.device, self.hidden_dim, self.density_dim
        lf, ne = len(self.grid_sizes) * self.table_feature_dim, self.sh_degree ** 2
        enc_t = ops.hashgrid_fwd(desc, tables, x)
        h0 = torch.empty((m, hd), dtype=F32, device=dev)
        ops.gemm(enc_t, 1, m, W[0][0], hd, 1, h0, hd, m, hd, lf, bias=W[0][1], act=L.ACT_RELU)
        dir_in = torch.empty((m, dd + ne + 1), dtype=F32, device=dev)
        ops.dense_fwd(h0, W[1][0], W[1][1], L.ACT_NONE, out=dir_in[:, :dd])
        # analytic normal: c1 = -e0, c0 = relu'(h0) * (c1 W1^T), g_enc = W0 c0, nraw = (d enc/dx)^T g_enc
        c1 = torch.zeros((m, dd), dtype=F32, device=dev)
        c1[:, 0] = -1.0
        c0 = ops.dense_bwd_input(c1, W[1][0], gate=h0)
        g_enc_t = torch.empty((lf, m), dtype=F32, device=dev)
        ops.gemm(W[0][0], hd, 1, c0, 1, hd, g_enc_t, m, lf, m, hd)
        nraw = ops.hashgrid_input_grad(desc, tables, x, g_enc_t)
        density, diffuse, spectral, aux2 = ops.refnerf_head_fwd(dir_in, nraw, d, self.sh_degree, dir_in[:, dd:])
        c = dir_in
        cacts = []
        li = 2
        for _ in range(self.color_layers):
            c = ops.dense_fwd(c, W[li][0], W[li][1], L.ACT_RELU)
            cacts.append(c)
            li += 1
        dir_out = ops.dense_fwd(c, W[li][0], W[li][1], L.ACT_NONE)
        rgb = ops.refnerf_color_fwd(dir_out, spectral, diffuse)
        aux = dict(normal_mse=aux2[:, 0], neg_normal=aux2[:, 1])
        ctx = None
        if save:
            ctx = dict(flat=flat, x=x, d=d, enc_t=enc_t, h0=h0, dir_in=dir_in, c0=c0, c1=c1, g_enc_t=g_enc_t,
                       nraw=nraw, density=density, diffuse=diffuse, spectral=spectral, cacts=cacts, dir_out=dir_out)
        return density, rgb, aux, ctx

    @ops.uses_model_precision
    def backward(self, ctx, g_density, g_rgb, g_aux, grad_flat):
        enc = self.encoding()
        desc = enc.desc()
        tables, W = self._dense_views(ctx["flat"])
        g_tables, G = self._dense_views(grad_flat)
        x, d, h0, dir_in, cacts = ctx["x"], ctx["d"], ctx["h0"], ctx["dir_in"], ctx["cacts"]
        m, dev, hd, dd = x.shape[0], grad_flat.device, self.hidden_dim, self.density_dim
        lf = len(self.grid_sizes) * self.table_feature_dim
        if g_aux is None:
            g_aux2 = torch.zeros((m, 2), dtype=F32, device=dev)
        else:
            g_aux2 = torch.stack([g_aux["normal_mse"].reshape(-1), g_aux["neg_normal"].reshape(-1)], 1).contiguous()
        g_do, g_sp, g_df = ops.refnerf_color_bwd(ctx["dir_out"], ctx["spectral"], ctx["diffuse"],
                                                 g_rgb.reshape(-1, 3).contiguous())
        li = len(W) - 1
        gy = g_do
        for i in reversed(range(self.color_layers)):
            ops.dense_bwd_weight(cacts[i], gy, G[li][0], G[li][1])
            gy = ops.dense_bwd_input(gy, W[li][0], gate=cacts[i])
            li -= 1
        ops.dense_bwd_weight(dir_in, gy, G[li][0], G[li][1])
        g_dir_in = ops.dense_bwd_input(gy, W[li][0])
        u = ops.refnerf_head_bwd(dir_in, ctx["nraw"], d, self.sh_degree, g_density.reshape(-1).contiguous(), g_df,
                                 g_sp, g_dir_in[:, dd:], g_aux2, g_dir_in)
        # (i) first-order path through Dense_1, Dense_0 and the tables
        g_out = g_dir_in[:, :dd]
        ops.dense_bwd_weight(h0, g_out, G[1][0], G[1][1])
        gy0 = ops.dense_bwd_input(g_out, W[1][0], gate=h0)
        ops.gemm(ctx["enc_t"], m, 1, gy0, hd, 1, G[0][0], hd, lf, hd, m, mode=2)
        ops.bias_grad(gy0, G[0][1])
        g1_t = torch.empty((lf, m), dtype=F32, device=dev)
        ops.gemm(W[0][0], hd, 1, gy0, 1, hd, g1_t, m, lf, m, hd)
        ops.hashgrid_bwd(desc, x, g1_t, g_tables)
        # (ii) second-order path: nraw = J^T g_enc with J = d enc/dx (linear in the tables), g_enc = W0 c0,
        # c0 = mask(h0) * (c1 W1^T).  Given u = dL/d nraw:
        ops.hashgrid_bwd_dir(desc, x, u, ctx["g_enc_t"], g_tables)          # d/d tables
        gbar_t = ops.hashgrid_jvp(desc, tables, x, u)                        # gbar_enc = J u, [L*F, M]
        ops.gemm(gbar_t, m, 1, ctx["c0"], hd, 1, G[0][0], hd, lf, hd, m, mode=2)  # dW0 += gbar_enc (x) c0
        cbar0 = torch.empty((m, hd), dtype=F32, device=dev)
        ops.gemm(gbar_t, 1, m, W[0][0], hd, 1, cbar0, hd, m, hd, lf)         # cbar0 = gbar_enc^T W0
        tbar0 = ops.act_bwd_(cbar0, h0, L.ACT_RELU)
        ops.dense_bwd_weight(tbar0, ctx["c1"], G[1][0], None)                # dW1 += tbar0^T c1
