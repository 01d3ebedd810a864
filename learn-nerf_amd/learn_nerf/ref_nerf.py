"""
learn_nerf.ref_nerf — RefNERFBase, RefNERFModel, linear_rgb_to_srgb, integrated_directional_encoding,
spherical_harmonic, HARMONIC_COUNTS, REF_NERF_OUT_DIM (reference: learn_nerf/ref_nerf.py).

The per-sample head (normals, reflection, integrated directional encoding, diffuse + specular, sRGB, aux
losses) and the embedding derivative maps are HIP kernels (refnerf.hip); the 273 -> 128 -> 3 directional block
runs on the dense GEMM kernels (dense.hip).  The analytic normal -d out[:,0]/dx (ref_nerf.py:38-43) is an explicit
input-gradient pass; because it enters the training loss (normal_mse, ref_nerf.py:73), backward() also runs the
second-order ("double backward") chain.  The spatial block — 94 % of the FLOPs — has two implementations:
  precision="bf16", default widths: the fused bf16-MFMA chain kernels of refnerf_fused.hip (trunk forward, normal
      pass, first- and second-order backward with the shared weight-gradient kernel);
  precision="fp32" or any other shape: the generic dense GEMM path (exact fp32 / bf16 operands), layer by layer.
"""
import ctypes
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Any, Dict, List, Tuple

import torch

from . import _lib as L
from . import _prof
from . import _ws
from . import ops
from .model import ModelBase
from .params import lecun_normal_

F32 = torch.float32
HARMONIC_COUNTS = [1, 3, 5, 7, 9, 11, 13, 15]
REF_NERF_OUT_DIM = 9


def linear_rgb_to_srgb(colors: torch.Tensor) -> torch.Tensor:
    """Gamma compression (ref_nerf.py:110-118).  Elementwise utility kept for API parity."""
    safe = torch.clamp(colors, min=1e-5)
    return torch.where(colors <= 0.0031308, 12.92 * colors, 1.055 * safe ** (1 / 2.4) - 0.055)


def integrated_directional_encoding(sh_degree: int, coords: torch.Tensor, roughness: torch.Tensor) -> torch.Tensor:
    """[N x 3] normalized coords, [N x 1] roughness -> [N x sh_degree^2] (ref_nerf.py:121-143)."""
    assert roughness.dim() == 2 and roughness.shape[1] == 1 and coords.dim() == 2 and coords.shape[1] == 3
    return ops.integrated_directional_encoding(sh_degree, coords.contiguous(), roughness.reshape(-1).contiguous())


def spherical_harmonic(sh_degree: int, coords: torch.Tensor) -> torch.Tensor:
    """Real spherical harmonics of normalized coords, [N x sh_degree^2] (ref_nerf.py:146-311)."""
    assert 1 <= sh_degree <= 8
    return ops.integrated_directional_encoding(sh_degree, coords.contiguous(), None)


@dataclass
class RefNERFBase(ModelBase):
    """A base class for Ref-NeRF models (ref_nerf.py:19-77)."""

    sh_degree: int = 4


@dataclass
class RefNERFModel(RefNERFBase):
    """A Ref-NeRF model built upon the original NeRF architecture (ref_nerf.py:80-107)."""

    input_layers: int = 5
    mid_layers: int = 4
    hidden_dim: int = 256
    color_layer_dim: int = 128
    x_freqs: int = 10
    d_freqs: int = 4  # unused, as in the reference
    precision: str = "bf16"  # operands of the Dense layers: "bf16" (MFMA rate) | "fp32" (exact, parity gate)
    tag: str = "refnerf"
    spatial_kernel: str = "fused"  # bf16 + default widths: "fused" chain kernels | "dense" GEMM path
    # fused path, forward without a backward (rendering, evaluation, model.apply): "bf16x3" = split-precision kernels
    # (lnrf_refnerf_trunk_normal_split / _dir_fwd_split: the reference's fp32 arithmetic to ~1e-5) | "bf16" = the training
    # forward's plain bf16 operands
    render_precision: str = "bf16x3"

    _pack_cache: Any = field(default=None, repr=False, compare=False)
    _pack_generation: int = field(default=0, repr=False, compare=False)

    def invalidate_packed(self) -> None:
        """Call after the parameters were modified outside torch (e.g. by lnrf_adam_step)."""
        self._pack_generation += 1

    def _use_fused_trunk(self) -> bool:
        return (self.precision == "bf16" and self.spatial_kernel == "fused"
                and (self.input_layers, self.mid_layers, self.hidden_dim, self.x_freqs) == (5, 4, 256, 10))

    def _use_fused_dir(self) -> bool:
        """the fused directional-block kernels cover the reference's widths: 256 + 4^2 + 1 -> 128 -> 3"""
        return self.sh_degree == 4 and self.color_layer_dim == 128

    def packed_trunk(self, flat: torch.Tensor, kind: str = "bf16") -> torch.Tensor:
        """Fragment streams for the fused kernels ("bf16": trunk, normal pass and directional block, forward and transposed;
        "split": hi/lo pairs of the forward-only streams for the render kernels); same cache discipline as
        NeRFModel.packed_weights (fresh buffer per miss, entries keep their source tensor alive)."""
        if self._pack_cache is None:
            self._pack_cache = OrderedDict()
        key = (kind, flat.data_ptr(), flat.numel(), flat._version, str(flat.device), self._pack_generation)
        hit = self._pack_cache.get(key)
        if hit is not None:
            self._pack_cache.move_to_end(key)
            return hit[1]
        if kind == "split":
            packed = torch.empty(L.lib().lnrf_refnerf_render_packed_bytes(), dtype=torch.uint8, device=flat.device)
            L.check(L.lib().lnrf_refnerf_render_pack(L.ptr(flat), L.ptr(packed, torch.uint8), L.stream()),
                    "refnerf_render_pack")
        else:
            packed = torch.empty(L.lib().lnrf_refnerf_trunk_packed_bytes(), dtype=torch.uint8, device=flat.device)
            L.check(L.lib().lnrf_refnerf_trunk_pack(L.ptr(flat), L.ptr(packed, torch.uint8), L.stream()),
                    "refnerf_trunk_pack")
        self._pack_cache[key] = (flat, packed)
        while len(self._pack_cache) > 6:
            self._pack_cache.popitem(last=False)
        return packed

    def layer_dims(self) -> List[Tuple[int, int]]:
        xe = 6 * self.x_freqs
        dims, fan = [], xe
        for _ in range(self.input_layers):
            dims.append((fan, self.hidden_dim))
            fan = self.hidden_dim
        fan = self.hidden_dim + xe
        for _ in range(self.mid_layers):
            dims.append((fan, self.hidden_dim))
            fan = self.hidden_dim
        dims.append((self.hidden_dim + self.sh_degree ** 2 + 1, self.color_layer_dim))
        dims.append((self.color_layer_dim, 3))
        return dims

    def param_spec(self):
        spec = []
        for i, (fi, fo) in enumerate(self.layer_dims()):
            spec.append((f"Dense_{i}", "kernel", (fi, fo)))
            spec.append((f"Dense_{i}", "bias", (fo,)))
        return spec

    def init_flat_(self, flat, gen):
        off = 0
        for fi, fo in self.layer_dims():
            lecun_normal_(flat[off:off + fi * fo].view(fi, fo), fi, gen)
            off += fi * fo + fo

    def _views(self, flat):
        out, off = [], 0
        for fi, fo in self.layer_dims():
            k = flat[off:off + fi * fo].view(fi, fo)
            off += fi * fo
            b = flat[off:off + fo]
            off += fo
            out.append((k, b))
        return out

    # ---- fused spatial block ---------------------------------------------------------------------------
    def _render_forward_points(self, flat, x, d):
        """Forward without a backward on the split-precision kernels (render_precision "bf16x3")."""
        lib = L.lib()
        m, dev, hd = x.shape[0], flat.device, self.hidden_dim
        ne = self.sh_degree ** 2
        packed3 = self.packed_trunk(flat, "split")
        width = hd + ne + 1
        ld = (width + 3) // 4 * 4
        dir_full = torch.empty((m, ld), dtype=F32, device=dev)
        dir_in = dir_full[:, :width]
        nraw = torch.empty((m, 3), dtype=F32, device=dev)
        lease = _ws.lease("ref_masks", lib.lnrf_refnerf_trunk_normal_split_scratch_bytes(m), dev)
        with _prof.section(f"{self.tag}_spatial_fwd_split"):
            L.check(lib.lnrf_refnerf_trunk_normal_split(L.ptr(packed3, torch.uint8), L.ptr(x), m, L.ptr(dir_full), ld,
                                                        L.ptr(nraw), L.ptr(lease.buf, torch.uint8), L.stream()),
                    "refnerf_trunk_normal_split")
        lease.release()
        with _prof.section(f"{self.tag}_head_fwd"):
            density, diffuse, spectral, aux2 = ops.refnerf_head_fwd(dir_in, nraw, d, self.sh_degree, dir_in[:, hd:])
            dir_out = torch.empty((m, 3), dtype=F32, device=dev)
            L.check(lib.lnrf_refnerf_dir_fwd_split(L.ptr(packed3, torch.uint8), L.ptr(dir_full), ld, m, L.ptr(dir_out),
                                                   L.stream()), "refnerf_dir_fwd_split")
            rgb = ops.refnerf_color_fwd(dir_out, spectral, diffuse)
        return density, rgb, dict(normal_mse=aux2[:, 0], neg_normal=aux2[:, 1]), None

    def _fused_forward_points(self, flat, x, d, save: bool):
        if self.render_precision not in ("bf16x3", "bf16"):
            raise ValueError(f"unknown render_precision {self.render_precision!r}")
        if not save and self.render_precision == "bf16x3" and self._use_fused_dir():
            return self._render_forward_points(flat, x, d)
        lib = L.lib()
        W = self._views(flat)
        m, dev, hd = x.shape[0], flat.device, self.hidden_dim
        ne = self.sh_degree ** 2
        ns = self.input_layers + self.mid_layers
        packed = self.packed_trunk(flat)
        shape = L.NerfShape(5, 4, 256, 128, 10, 4)  # the trunk's buffers have NeRFModel's layout
        # GB-sized blocks are leased from step-persistent pools (_ws.py) and stay busy as long as the context lives
        leases = [_ws.lease("ref_save", lib.lnrf_nerf_save_bytes(ctypes.byref(shape), m), dev),
                  _ws.lease("ref_cdump", lib.lnrf_nerf_bwd_scratch_bytes(ctypes.byref(shape), m), dev)]
        save_buf, cdump = leases[0].buf, leases[1].buf
        width = hd + ne + 1
        ld = (width + 3) // 4 * 4  # rows 16-byte aligned for the fused kernels' float4 accesses
        dir_full = torch.empty((m, ld), dtype=F32, device=dev)
        dir_in = dir_full[:, :width]  # [spatial_out, IDE, -d.n] (ref_nerf.py:63)
        nraw = torch.empty((m, 3), dtype=F32, device=dev)
        with _prof.section(f"{self.tag}_spatial_fwd"):
            L.check(lib.lnrf_refnerf_trunk_fwd(L.ptr(packed, torch.uint8), L.ptr(x), m, L.ptr(save_buf, torch.uint8),
                                               L.ptr(dir_full), ld, L.stream()), "refnerf_trunk_fwd")
        with _prof.section(f"{self.tag}_normal_pass"):
            L.check(lib.lnrf_refnerf_normal_pass(L.ptr(packed, torch.uint8), L.ptr(save_buf, torch.uint8), L.ptr(x), m,
                                                 L.ptr(cdump, torch.uint8), L.ptr(nraw), L.stream()),
                    "refnerf_normal_pass")
        hcol = dsave = None
        with _prof.section(f"{self.tag}_head_fwd"), ops.dense_precision(self.precision):
            density, diffuse, spectral, aux2 = ops.refnerf_head_fwd(dir_in, nraw, d, self.sh_degree, dir_in[:, hd:])
            if self._use_fused_dir():  # ref_nerf.py:105-107 on the fused chain
                leases.append(_ws.lease("ref_dsave", lib.lnrf_refnerf_dir_save_bytes(m), dev))
                dsave = leases[-1].buf
                dir_out = torch.empty((m, 3), dtype=F32, device=dev)
                L.check(lib.lnrf_refnerf_dir_fwd(L.ptr(packed, torch.uint8), L.ptr(dir_full), ld, m,
                                                 L.ptr(dsave, torch.uint8), L.ptr(dir_out), L.stream()),
                        "refnerf_dir_fwd")
            else:
                hcol = ops.dense_fwd(dir_in, W[ns][0], W[ns][1], L.ACT_RELU)
                dir_out = ops.dense_fwd(hcol, W[ns + 1][0], W[ns + 1][1], L.ACT_NONE)
            rgb = ops.refnerf_color_fwd(dir_out, spectral, diffuse)
        aux = dict(normal_mse=aux2[:, 0], neg_normal=aux2[:, 1])
        ctx = None
        if save:
            ctx = dict(kind="fused", flat=flat, packed=packed, x=x, d=d, save=save_buf, cdump=cdump, dir_in=dir_in,
                       ld=ld, nraw=nraw, density=density, diffuse=diffuse, spectral=spectral, hcol=hcol,
                       dir_out=dir_out, dsave=dsave, leases=leases)
        return density, rgb, aux, ctx

    def _fused_backward(self, ctx, g_density, g_rgb, g_aux, grad_flat):
        lib = L.lib()
        W = self._views(ctx["flat"])
        G = self._views(grad_flat)
        x, d, dir_in, ld = ctx["x"], ctx["d"], ctx["dir_in"], ctx["ld"]
        m, dev, hd = x.shape[0], grad_flat.device, self.hidden_dim
        ns = self.input_layers + self.mid_layers
        packed, save_buf, cdump = ctx["packed"], ctx["save"], ctx["cdump"]
        if g_aux is None:
            g_aux2 = torch.zeros((m, 2), dtype=F32, device=dev)
        else:
            g_aux2 = torch.stack([g_aux["normal_mse"].reshape(-1), g_aux["neg_normal"].reshape(-1)], 1).contiguous()
        with _prof.section(f"{self.tag}_head_bwd"), ops.dense_precision(self.precision):
            g_do, g_sp, g_df = ops.refnerf_color_bwd(ctx["dir_out"], ctx["spectral"], ctx["diffuse"],
                                                     g_rgb.reshape(-1, 3).contiguous())
            g_full = torch.empty((m, ld), dtype=F32, device=dev)
            if ctx["dsave"] is not None:
                dlease = _ws.lease("ref_dscratch", lib.lnrf_refnerf_dir_scratch_bytes(m), dev)
                dscratch = dlease.buf
                L.check(lib.lnrf_refnerf_dir_bwd(L.ptr(packed, torch.uint8), L.ptr(ctx["dsave"], torch.uint8),
                                                 L.ptr(g_do), m, L.ptr(dscratch, torch.uint8), L.ptr(g_full), ld,
                                                 L.ptr(grad_flat), L.stream()), "refnerf_dir_bwd")
                g_dir_in = g_full[:, :dir_in.shape[1]]
            else:
                ops.dense_bwd_weight(ctx["hcol"], g_do, G[ns + 1][0], G[ns + 1][1])
                gy = ops.dense_bwd_input(g_do, W[ns + 1][0], gate=ctx["hcol"])
                ops.dense_bwd_weight(dir_in, gy, G[ns][0], G[ns][1])
                g_dir_in = ops.dense_bwd_input(gy, W[ns][0], out=g_full[:, :dir_in.shape[1]])
            u = ops.refnerf_head_bwd(dir_in, ctx["nraw"], d, self.sh_degree, g_density.reshape(-1).contiguous(), g_df,
                                     g_sp, g_dir_in[:, hd:], g_aux2, g_dir_in)
        shape = L.NerfShape(5, 4, 256, 128, 10, 4)
        with _prof.section(f"{self.tag}_spatial_bwd"):  # first-order: d L / d spatial_out -> Dense_8 .. Dense_0
            slease = _ws.lease("ref_scratch", lib.lnrf_refnerf_trunk_bwd_scratch_bytes(m), dev)
            scratch = slease.buf
            L.check(lib.lnrf_refnerf_trunk_bwd(L.ptr(packed, torch.uint8), L.ptr(save_buf, torch.uint8), L.ptr(g_full),
                                               ld, m, L.ptr(scratch, torch.uint8), L.ptr(grad_flat), L.stream()),
                    "refnerf_trunk_bwd")
        with _prof.section(f"{self.tag}_normal_bwd"):  # second-order: through n_raw (u = d L / d n_raw)
            tlease = _ws.lease("ref_tscratch", lib.lnrf_nerf_save_bytes(ctypes.byref(shape), m), dev)
            tscratch = tlease.buf
            L.check(lib.lnrf_refnerf_normal_bwd(L.ptr(packed, torch.uint8), L.ptr(save_buf, torch.uint8),
                                                L.ptr(cdump, torch.uint8), L.ptr(x), L.ptr(u), m,
                                                L.ptr(tscratch, torch.uint8), L.ptr(grad_flat), L.stream()),
                    "refnerf_normal_bwd")

    # ---- forward ------------------------------------------------------------------------------------
    def forward_points(self, flat, x, d, save: bool):
        if self._use_fused_trunk():
            return self._fused_forward_points(flat, x, d, save)
        return self._dense_forward_points(flat, x, d, save)

    def backward(self, ctx, g_density, g_rgb, g_aux, grad_flat):
        if ctx.get("kind") == "fused":
            return self._fused_backward(ctx, g_density, g_rgb, g_aux, grad_flat)
        return self._dense_backward(ctx, g_density, g_rgb, g_aux, grad_flat)

    @ops.uses_model_precision
    def _dense_forward_points(self, flat, x, d, save: bool):
        W = self._views(flat)
        m, dev, hd = x.shape[0], flat.device, self.hidden_dim
        xe_w, ne = 6 * self.x_freqs, self.sh_degree ** 2
        ns = self.input_layers + self.mid_layers  # spatial Dense layers
        with _prof.section(f"{self.tag}_spatial_fwd"):
            cat_x = torch.empty((m, hd + xe_w), dtype=F32, device=dev)
            ops.sinusoidal_emb_into(x, self.x_freqs, cat_x, hd)
            dir_in = torch.empty((m, hd + ne + 1), dtype=F32, device=dev)  # [spatial_out, IDE, -d.n] (:63)
            h = []  # relu outputs of Dense_0 .. Dense_{ns-2}; h[input_layers-1] aliases cat_x[:, :hd]
            z = cat_x[:, hd:]
            for i in range(self.input_layers):
                out = cat_x[:, :hd] if i == self.input_layers - 1 else None
                z = ops.dense_fwd(z, W[i][0], W[i][1], L.ACT_RELU, out=out)
                h.append(z)
            z = cat_x
            for i in range(self.mid_layers):
                last = i == self.mid_layers - 1
                z = ops.dense_fwd(z, W[self.input_layers + i][0], W[self.input_layers + i][1],
                                  L.ACT_NONE if last else L.ACT_RELU, out=dir_in[:, :hd] if last else None)
                if not last:
                    h.append(z)
        with _prof.section(f"{self.tag}_normal_pass"):
            # analytic normal: c_l = d(-out0)/d y_l back to the embedding (ref_nerf.py:38-42)
            c = [None] * ns
            top = torch.zeros((m, hd), dtype=F32, device=dev)
            top[:, 0] = -1.0
            c[ns - 1] = top
            g_cat = None
            for l in range(ns - 1, 0, -1):
                # input gradient with the ReLU backward of the layer below fused in (first hd columns)
                gh = ops.dense_bwd_input(c[l], W[l][0], gate=h[l - 1])  # [M, fan_in(l)]
                if l == self.input_layers:  # Dense_5 consumed [h4, x_emb]
                    g_cat = gh
                    gh = gh[:, :hd]
                c[l - 1] = gh
            if g_cat is None:  # no mid layers: should not happen with the reference architecture
                g_cat = torch.zeros((m, hd + xe_w), dtype=F32, device=dev)
            ops.dense_bwd_input(c[0], W[0][0], out=g_cat[:, hd:], accumulate=True)  # x_emb also feeds Dense_0
            nraw = ops.sinusoidal_emb_bwd(x, self.x_freqs, g_cat, col_off=hd)
        with _prof.section(f"{self.tag}_head_fwd"):
            density, diffuse, spectral, aux2 = ops.refnerf_head_fwd(dir_in, nraw, d, self.sh_degree, dir_in[:, hd:])
            hcol = ops.dense_fwd(dir_in, W[ns][0], W[ns][1], L.ACT_RELU)  # ref_nerf.py:105-107
            dir_out = ops.dense_fwd(hcol, W[ns + 1][0], W[ns + 1][1], L.ACT_NONE)
            rgb = ops.refnerf_color_fwd(dir_out, spectral, diffuse)
        aux = dict(normal_mse=aux2[:, 0], neg_normal=aux2[:, 1])
        ctx = None
        if save:
            ctx = dict(flat=flat, x=x, d=d, cat_x=cat_x, dir_in=dir_in, h=h, c=c, nraw=nraw, density=density,
                       diffuse=diffuse, spectral=spectral, hcol=hcol, dir_out=dir_out)
        return density, rgb, aux, ctx

    # ---- backward -------------------------------------------------------------------------------------
    @ops.uses_model_precision
    def _dense_backward(self, ctx, g_density, g_rgb, g_aux, grad_flat):
        W = self._views(ctx["flat"])
        G = self._views(grad_flat)
        x, d, cat_x, dir_in, h, c = ctx["x"], ctx["d"], ctx["cat_x"], ctx["dir_in"], ctx["h"], ctx["c"]
        m, dev, hd = x.shape[0], grad_flat.device, self.hidden_dim
        xe_w = 6 * self.x_freqs
        ns, il = self.input_layers + self.mid_layers, self.input_layers
        if g_aux is None:
            g_aux2 = torch.zeros((m, 2), dtype=F32, device=dev)
        else:
            g_aux2 = torch.stack([g_aux["normal_mse"].reshape(-1), g_aux["neg_normal"].reshape(-1)], 1).contiguous()
        with _prof.section(f"{self.tag}_head_bwd"):
            g_do, g_sp, g_df = ops.refnerf_color_bwd(ctx["dir_out"], ctx["spectral"], ctx["diffuse"],
                                                     g_rgb.reshape(-1, 3).contiguous())
            ops.dense_bwd_weight(ctx["hcol"], g_do, G[ns + 1][0], G[ns + 1][1])
            gy = ops.dense_bwd_input(g_do, W[ns + 1][0], gate=ctx["hcol"])
            ops.dense_bwd_weight(dir_in, gy, G[ns][0], G[ns][1])
            g_dir_in = ops.dense_bwd_input(gy, W[ns][0])  # [M, hd + sh^2 + 1]
            u = ops.refnerf_head_bwd(dir_in, ctx["nraw"], d, self.sh_degree, g_density.reshape(-1).contiguous(), g_df,
                                     g_sp, g_dir_in[:, hd:], g_aux2, g_dir_in)
        with _prof.section(f"{self.tag}_spatial_bwd"):
            # (i) first-order path: d L / d spatial_out back through Dense_{ns-1} .. Dense_0
            gy = g_dir_in[:, :hd]  # Dense_{ns-1} output is linear
            for l in range(ns - 1, -1, -1):
                inp = cat_x if l == il else (cat_x[:, hd:] if l == 0 else h[l - 1])
                ops.dense_bwd_weight(inp, gy, G[l][0], G[l][1])
                if l > 0:
                    gy = ops.dense_bwd_input(gy, W[l][0][:hd], gate=h[l - 1])
        with _prof.section(f"{self.tag}_normal_bwd"):
            # (ii) second-order path: the normal pass is a chain in the SAME kernels with the ReLU masks
            # fixed (ReLU'' = 0 a.e.): ge = W_0 c_0 + W_5[hd:] c_5, c_{l-1} = mask_{l-1} * (W_l c_l).
            # Given u = d L / d nraw:   ubar_e = (d emb/dx) u;  tbar_{l-1} = mask * cbar_{l-1};
            # dW_l += tbar_{l-1}^T c_l;  cbar_l = tbar_{l-1} W_l.
            tc = torch.empty((m, hd + xe_w), dtype=F32, device=dev)  # [tbar_{il-1}, ubar_e]
            ops.sinusoidal_emb_jvp_into(x, self.x_freqs, u, tc, hd)
            ubar_e = tc[:, hd:]
            ops.dense_bwd_weight(ubar_e, c[0], G[0][0], None)
            # tbar_l = relu'(h_l) * (tbar_{l-1} W_l): the mask is applied in the GEMM epilogue (gated forward);
            # the product that feeds Dense_{il} is written straight into its [tbar, ubar_e] buffer
            tbar = ops.dense_fwd(ubar_e, W[0][0], None, L.ACT_NONE, out=tc[:, :hd] if il == 1 else None, gate=h[0])
            for l in range(1, ns):
                tb = tc if l == il else tbar  # Dense_5 sees [tbar_4, ubar_e]
                ops.dense_bwd_weight(tb, c[l], G[l][0], None)
                if l < ns - 1:
                    tbar = ops.dense_fwd(tb, W[l][0], None, L.ACT_NONE, out=tc[:, :hd] if l + 1 == il else None,
                                         gate=h[l])
