"""
learn_nerf.model — ModelBase, NeRFModel, sinusoidal_emb (reference: learn_nerf/model.py).

Same constructor fields and call pattern as the Flax modules:
    params = model.init(dict(params=rng), x, d)["params"]
    density, rgb, aux = model.apply(dict(params=params), x, d)          (render.py:320-324)
Arrays are torch tensors on the GPU.  NeRFModel has two compute paths, both hand-written HIP:
  precision="bf16": fused bf16-MFMA kernels (nerf_mlp.hip) — the performance path.  Training (forward with
      saved activations + backward) runs on plain bf16 operands; every forward WITHOUT a backward (rendering,
      evaluation, model.apply) runs the split-precision kernel (render_precision="bf16x3": bf16 hi+lo pairs,
      three MFMAs per product, fp32 accumulate), which reproduces the reference's fp32 arithmetic
      (model.py:72, render.py:140) to ~1e-5 so that rendered RGB meets the 1e-3 parity gate;
  precision="fp32": exact-fp32 dense kernels on the f32 MFMA (dense.hip) — parity/any shape.
"""
import ctypes
import os
import warnings
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import ClassVar, Any, Dict, List, Optional, Tuple

import torch

from . import _lib as L
from . import _prof
from . import _ws
from . import ops
from .params import ParamTree, as_generator, build_tree, default_device, flat_of, lecun_normal_, spec_size

F32 = torch.float32


class ModelBase:
    """
    Base class used by all NeRF models (model.py:7-27).

    __call__/apply(x[N,3], d[N,3]) -> (density[N,1] >= 0, rgb[N,3] in [-1,1], aux{name: [N]}).
    Subclasses provide param_spec(), _forward(flat, x, d, rays, ts, save) and _backward(...).
    """

    def param_spec(self) -> List[Tuple[str, str, Tuple[int, ...]]]:
        raise NotImplementedError

    def num_params(self) -> int:
        return spec_size(self.param_spec())

    # ---- Flax-like API ------------------------------------------------------------------
    def init(self, rngs, x=None, d=None, device=None) -> Dict[str, ParamTree]:
        """model.init(dict(params=key), x, d) (train.py:49-50): returns {"params": tree}."""
        rng = rngs["params"] if isinstance(rngs, dict) else rngs
        device = device if device is not None else default_device()
        flat = torch.zeros(self.num_params(), dtype=F32)
        self.init_flat_(flat, as_generator(rng))
        return {"params": build_tree(flat.to(device), self.param_spec())}

    def init_flat_(self, flat: torch.Tensor, gen: torch.Generator) -> None:
        raise NotImplementedError

    def tree(self, flat: torch.Tensor) -> ParamTree:
        return build_tree(flat, self.param_spec())

    def flat(self, params) -> torch.Tensor:
        return flat_of(params, self.param_spec())

    def apply(self, variables, x: torch.Tensor, d: torch.Tensor):
        params = variables["params"] if "params" in variables and not _is_leafy(variables) else variables
        density, rgb, aux, _ = self.forward_points(self.flat(params), x.contiguous(), d.contiguous(), save=False)
        return density.reshape(-1, 1), rgb, aux

    def __call__(self, x, d):
        raise NotImplementedError("call model.apply(dict(params=params), x, d)")

    # ---- kernel-facing API (used by render.py / train.py) ------------------------------------
    def forward_points(self, flat, x, d, save: bool):
        """-> density[M], rgb[M,3], aux{name: [M]}, ctx"""
        raise NotImplementedError

    def forward_rays(self, flat, rays, ts, save: bool):
        """Evaluate at x = o + d*ts without the caller materialising points (render.py:318-319).
        -> density[N,T], rgb[N,T,3], aux{name: [N,T]}, ctx"""
        pts, dirs = ops.ray_points(rays, ts)
        n, t = ts.shape
        density, rgb, aux, ctx = self.forward_points(flat, pts.view(-1, 3), dirs.view(-1, 3), save)
        return density.view(n, t), rgb.view(n, t, 3), {k: v.view(n, t) for k, v in aux.items()}, ctx

    def backward(self, ctx, g_density, g_rgb, g_aux, grad_flat) -> None:
        """grad_flat += d L / d params given gradients wrt the forward outputs."""
        raise NotImplementedError


def ls_status(ctx) -> int:
    """Status word of the last layer-stationary backward of this context (synchronises): 0 = every hand-off completed,
    otherwise the code of the bounded wait that gave up (the gradients of that call are invalid)."""
    buf = ctx.get("ls_scratch")
    if buf is None:
        return 0
    shape = L.NerfShape(5, 4, 256, 128, 10, 4)
    off = L.lib().lnrf_nerf_bwd_ls_status_offset(ctypes.byref(shape), ctx["m"])
    return int(buf[off:off + 4].view(torch.int32).item())


def backward_ls_pair(ctx_a, g_density_a, g_rgb_a, grad_a, ctx_b, g_density_b, g_rgb_b, grad_b) -> None:
    """The layer-stationary backward of TWO fused NeRFModel contexts (train.py:141-142: coarse and fine) in one persistent
    launch (lnrf_nerf_mlp_bwd_ls2): the pipelines of the chip are shared in proportion to the evaluations, so the coarse
    pass does not pay a pipeline fill and drain of its own.  grad_a / grad_b += d L / d params."""
    shape = L.NerfShape(5, 4, 256, 128, 10, 4)
    lib = L.lib()
    dev = grad_a.device
    la = _ws.lease("nerf_bwd_ls", lib.lnrf_nerf_bwd_ls_scratch_bytes(ctypes.byref(shape), ctx_a["m"]), dev)
    lb = _ws.lease("nerf_bwd_ls", lib.lnrf_nerf_bwd_ls_scratch_bytes(ctypes.byref(shape), ctx_b["m"]), dev)
    # with the kernel-family timers on (bench.py) the three parts are separate calls so that the persistent pipeline launch
    # — the dominant kernel — has its own HIP-event section
    parts = ((1, "bwd_ls_head"), (2, "bwd_ls_pipeline"), (4, "bwd_ls_finish")) if _prof.enabled() else ((7, "bwd_ls"),)
    for phases, name in parts:
        with _prof.section(name):
            L.check(lib.lnrf_nerf_mlp_bwd_ls2(
                ctypes.byref(shape),
                L.ptr(ctx_a["packed"], torch.uint8), L.ptr(ctx_a["save"], torch.uint8), L.ptr(ctx_a["density"]),
                L.ptr(ctx_a["rgb"]), L.ptr(g_density_a.reshape(-1)), L.ptr(g_rgb_a.reshape(-1, 3)), ctx_a["m"],
                L.ptr(la.buf, torch.uint8), L.ptr(grad_a),
                L.ptr(ctx_b["packed"], torch.uint8), L.ptr(ctx_b["save"], torch.uint8), L.ptr(ctx_b["density"]),
                L.ptr(ctx_b["rgb"]), L.ptr(g_density_b.reshape(-1)), L.ptr(g_rgb_b.reshape(-1, 3)), ctx_b["m"],
                L.ptr(lb.buf, torch.uint8), L.ptr(grad_b), phases, L.stream()), "nerf_mlp_bwd_ls2")
    ctx_a["ls_scratch"], ctx_b["ls_scratch"] = la.buf, lb.buf
    la.release()
    lb.release()


def _is_leafy(d) -> bool:
    return any(isinstance(v, torch.Tensor) for v in d.values())


def sinusoidal_emb(coords: torch.Tensor, freqs: int) -> torch.Tensor:
    """
    Compute sinusoidal embeddings (model.py:65-77): [N x D] -> [N x D*freqs*2], per coordinate
    [sin(2^0 c) .. sin(2^(F-1) c), cos(2^0 c) .. cos(2^(F-1) c)].
    """
    shape = coords.shape
    x = coords.reshape(-1, shape[-1]).contiguous()
    out = torch.empty((x.shape[0], shape[-1] * 2 * freqs), dtype=F32, device=x.device)
    ops.sinusoidal_emb_into(x, freqs, out, 0)
    return out.reshape(shape[:-1] + (shape[-1] * 2 * freqs,))


@dataclass
class NeRFModel(ModelBase):
    """
    A model architecture based directly on Mildenhall et al. (2020) (model.py:30-62).
    """

    input_layers: int = 5
    mid_layers: int = 4
    hidden_dim: int = 256
    color_layer_dim: int = 128
    x_freqs: int = 10
    d_freqs: int = 4
    precision: str = "bf16"  # "bf16" (fused MFMA) | "fp32" (exact dense path)
    render_precision: str = "bf16x3"  # fused path, forward without backward: "bf16x3" (split) | "bf16"
    # fused path, backward: "ls" = layer-stationary pipeline (lnrf_nerf_mlp_bwd_ls: one CU per Dense layer, dW in
    # registers; measured 4.74 vs 5.11 ms per 4096-ray step) | "split" = separate chain and weight-gradient launches through
    # HBM (lnrf_nerf_mlp_bwd_chain / _bwd_weights)
    backward_kernel: str = "ls"
    tag: str = "mlp"  # label used by the optional kernel-family timers (_prof)
    # TrainLoop may run the coarse backward on a second stream beside the fine forward; for this model every kernel
    # already fills all 256 CUs at one workgroup per CU, so the two streams only time-slice (5.012 / 5.020 ms without,
    # 5.001 / 5.003 ms with it, same box) while the per-kernel HIP-event and rocprof durations double.  Off by default
    # so that the kernel timers of bench.py stay meaningful; LNRF_OVERLAP_BACKWARD=1 turns it on.
    overlap_backward_hint: ClassVar[bool] = False

    _pack_cache: Any = field(default=None, repr=False, compare=False)
    _pack_generation: int = field(default=0, repr=False, compare=False)
    _warned_dense: bool = field(default=False, repr=False, compare=False)

    def __post_init__(self):
        # model.py:50-56 builds Dense_0..(input_layers-1) and the mid block; the reference's widths for an empty
        # block (no Dense between the embedding and the heads) are not supported here rather than guessed
        if self.input_layers < 1 or self.mid_layers < 1:
            raise ValueError("NeRFModel needs input_layers >= 1 and mid_layers >= 1")
        if min(self.hidden_dim, self.color_layer_dim, self.x_freqs, self.d_freqs) < 1:
            raise ValueError("NeRFModel widths and frequency counts must be positive")

    def invalidate_packed(self) -> None:
        """Call after the parameters were modified outside torch (e.g. by lnrf_adam_step)."""
        self._pack_generation += 1

    # ---- structure ---------------------------------------------------------------------------
    def layer_dims(self) -> List[Tuple[int, int]]:
        xe, de = 6 * self.x_freqs, 6 * self.d_freqs
        dims, fan = [], xe
        for _ in range(self.input_layers):
            dims.append((fan, self.hidden_dim))
            fan = self.hidden_dim
        fan = self.hidden_dim + xe
        for _ in range(self.mid_layers):
            dims.append((fan, self.hidden_dim))
            fan = self.hidden_dim
        dims.append((self.hidden_dim, 1))
        dims.append((self.hidden_dim + de, self.color_layer_dim))
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
            off += fi * fo + fo  # bias stays zero (Flax default)

    def _shape_struct(self) -> L.NerfShape:
        return L.NerfShape(self.input_layers, self.mid_layers, self.hidden_dim, self.color_layer_dim,
                           self.x_freqs, self.d_freqs)

    def fused_supported(self) -> bool:
        return (self.input_layers, self.mid_layers, self.hidden_dim, self.color_layer_dim, self.x_freqs,
                self.d_freqs) == (5, 4, 256, 128, 10, 4)

    def _use_fused(self) -> bool:
        if self.precision not in ("bf16", "fp32"):
            raise ValueError(f"unknown precision {self.precision!r}")
        if self.render_precision not in ("bf16x3", "bf16"):
            raise ValueError(f"unknown render_precision {self.render_precision!r}")
        fused = self.precision == "bf16" and self.fused_supported()
        if self.precision == "bf16" and not fused and not self._warned_dense:
            self._warned_dense = True
            warnings.warn("NeRFModel: only the default shape (5, 4, 256, 128, 10, 4) has fused kernels; this shape "
                          "runs on the generic dense GEMM path (about 10x slower)", RuntimeWarning, stacklevel=3)
        return fused

    # ---- fused bf16 path ----------------------------------------------------------------------
    def packed_weights(self, flat: torch.Tensor, kind: str = "bf16") -> torch.Tensor:
        """
        MFMA-fragment copy of the parameters: kind "bf16" (forward + transposed streams, training) or
        "split" (bf16 hi/lo pairs, rendering).  A small cache maps (storage address, size, version, generation)
        to the packed buffer.  Every miss packs into a FRESH buffer — a saved backward context may still hold
        the previous one (e.g. one model instance serving as coarse and fine) — and every entry keeps a
        reference to its source tensor, so a freed temporary's address cannot come back as a stale hit.
        """
        if self._pack_cache is None:
            self._pack_cache = OrderedDict()
        key = (kind, flat.data_ptr(), flat.numel(), flat._version, str(flat.device), self._pack_generation)
        hit = self._pack_cache.get(key)
        if hit is not None:
            self._pack_cache.move_to_end(key)
            return hit[1]
        shape = self._shape_struct()
        lib = L.lib()
        if kind == "split":
            nbytes = lib.lnrf_nerf_packed_split_bytes(ctypes.byref(shape))
            packed = torch.empty(nbytes, dtype=torch.uint8, device=flat.device)
            L.check(lib.lnrf_nerf_pack_weights_split(ctypes.byref(shape), L.ptr(flat), L.ptr(packed, torch.uint8),
                                                     L.stream()), "nerf_pack_weights_split")
        else:
            nbytes = lib.lnrf_nerf_packed_bytes(ctypes.byref(shape))
            packed = torch.empty(nbytes, dtype=torch.uint8, device=flat.device)
            L.check(lib.lnrf_nerf_pack_weights(ctypes.byref(shape), L.ptr(flat), L.ptr(packed, torch.uint8),
                                               L.stream()), "nerf_pack_weights")
        self._pack_cache[key] = (flat, packed)
        while len(self._pack_cache) > 4:
            self._pack_cache.popitem(last=False)
        return packed

    def _fused_fwd(self, flat, m, save, x=None, d=None, rays=None, ts=None):
        shape = self._shape_struct()
        dev = flat.device
        density = torch.empty(m, dtype=F32, device=dev)
        rgb = torch.empty((m, 3), dtype=F32, device=dev)
        if rays is not None:
            rstride, t = rays.shape[1] * 3, ts.shape[1]
        else:
            rstride, t = 6, 0
        if not save and self.render_precision == "bf16x3":
            packed3 = self.packed_weights(flat, "split")
            with _prof.section(f"{self.tag}_fwd_split"):
                L.check(L.lib().lnrf_nerf_mlp_fwd_split(
                    ctypes.byref(shape), L.ptr(packed3, torch.uint8), L.ptr(x), L.ptr(d), L.ptr(rays), rstride,
                    L.ptr(ts), t, m, L.ptr(density), L.ptr(rgb), L.stream()), "nerf_mlp_fwd_split")
            return density, rgb, None
        packed = self.packed_weights(flat)
        save_buf, save_lease = None, None
        if save:  # leased for the life of the backward context (a step-persistent block, see _ws.py)
            save_lease = _ws.lease("nerf_save", L.lib().lnrf_nerf_save_bytes(ctypes.byref(shape), m), dev)
            save_buf = save_lease.buf
        # the layer-stationary backward takes the hidden layers' ReLU masks from the saved activations: its forward
        # leaves their mask slots unwritten (lnrf_nerf_mlp_fwd_ls)
        hidden_masks = not (save and self.backward_kernel == "ls")
        fwd = L.lib().lnrf_nerf_mlp_fwd if hidden_masks else L.lib().lnrf_nerf_mlp_fwd_ls
        with _prof.section(f"{self.tag}_fwd"):
            L.check(fwd(
                ctypes.byref(shape), L.ptr(packed, torch.uint8), L.ptr(x), L.ptr(d), L.ptr(rays), rstride,
                L.ptr(ts), t, m, L.ptr(density), L.ptr(rgb), L.ptr(save_buf, torch.uint8), L.stream()),
                "nerf_mlp_fwd")
        ctx = (dict(kind="fused", packed=packed, save=save_buf, save_lease=save_lease, density=density, rgb=rgb, m=m,
                    tag=self.tag, hidden_masks=hidden_masks) if save else None)
        return density, rgb, ctx

    def forward_points(self, flat, x, d, save: bool):
        m = x.shape[0]
        if self._use_fused():
            density, rgb, ctx = self._fused_fwd(flat, m, save, x=x, d=d)
            return density, rgb, {}, ctx
        return self._dense_fwd(flat, x, d, save)

    def forward_rays(self, flat, rays, ts, save: bool):
        if self._use_fused():
            n, t = ts.shape
            density, rgb, ctx = self._fused_fwd(flat, n * t, save, rays=rays, ts=ts)
            return density.view(n, t), rgb.view(n, t, 3), {}, ctx
        return super().forward_rays(flat, rays, ts, save)

    def backward(self, ctx, g_density, g_rgb, g_aux, grad_flat):
        if ctx["kind"] == "fused":
            shape = self._shape_struct()
            m = ctx["m"]
            tag = ctx.get("tag", "mlp")
            if self.backward_kernel == "ls":
                lease = _ws.lease("nerf_bwd_ls", L.lib().lnrf_nerf_bwd_ls_scratch_bytes(ctypes.byref(shape), m),
                                  grad_flat.device)
                with _prof.section(f"{tag}_bwd_ls"):
                    L.check(L.lib().lnrf_nerf_mlp_bwd_ls(
                        ctypes.byref(shape), L.ptr(ctx["packed"], torch.uint8), L.ptr(ctx["save"], torch.uint8),
                        L.ptr(ctx["density"]), L.ptr(ctx["rgb"]), L.ptr(g_density.reshape(-1)),
                        L.ptr(g_rgb.reshape(-1, 3)), m, L.ptr(lease.buf, torch.uint8), L.ptr(grad_flat), 7, L.stream()),
                        "nerf_mlp_bwd_ls")
                ctx["ls_scratch"] = lease.buf  # ls_status(ctx) reads the hand-off status word from it
                lease.release()
                return
            if self.backward_kernel != "split":
                raise ValueError(f"unknown backward_kernel {self.backward_kernel!r}")
            if not ctx.get("hidden_masks", True):
                raise ValueError("this forward was saved for the layer-stationary backward (no hidden ReLU-mask slots): "
                                 "set backward_kernel before the forward")
            lease = _ws.lease("nerf_bwd", L.lib().lnrf_nerf_bwd_scratch_bytes(ctypes.byref(shape), m), grad_flat.device)
            scratch = lease.buf
            with _prof.section(f"{tag}_bwd_chain"):
                L.check(L.lib().lnrf_nerf_mlp_bwd_chain(
                    ctypes.byref(shape), L.ptr(ctx["packed"], torch.uint8), L.ptr(ctx["save"], torch.uint8),
                    L.ptr(ctx["density"]), L.ptr(ctx["rgb"]), L.ptr(g_density.reshape(-1)),
                    L.ptr(g_rgb.reshape(-1, 3)), m, L.ptr(scratch, torch.uint8), L.stream()), "nerf_mlp_bwd_chain")
            with _prof.section(f"{tag}_bwd_weights"):
                L.check(L.lib().lnrf_nerf_mlp_bwd_weights(
                    ctypes.byref(shape), L.ptr(ctx["save"], torch.uint8), L.ptr(scratch, torch.uint8), m,
                    L.ptr(grad_flat), L.stream()), "nerf_mlp_bwd_weights")
            lease.release()
            return
        self._dense_bwd(ctx, g_density, g_rgb, grad_flat)

    # ---- exact fp32 dense path (any shape) ------------------------------------------------------
    @ops.uses_model_precision
    def _dense_fwd(self, flat, x, d, save: bool):
        tree = self.tree(flat)
        W = [(tree[f"Dense_{i}"]["kernel"], tree[f"Dense_{i}"]["bias"]) for i in range(len(self.layer_dims()))]
        m, dev, hd = x.shape[0], flat.device, self.hidden_dim
        xe_w, de_w = 6 * self.x_freqs, 6 * self.d_freqs
        cat_x = torch.empty((m, hd + xe_w), dtype=F32, device=dev)  # [z, x_emb] (model.py:52)
        ops.sinusoidal_emb_into(x, self.x_freqs, cat_x, hd)
        x_emb = cat_x[:, hd:]
        acts = []
        z = x_emb
        li = 0
        for i in range(self.input_layers):  # model.py:50-51
            out = cat_x[:, :hd] if i == self.input_layers - 1 else None
            z = ops.dense_fwd(z, W[li][0], W[li][1], L.ACT_RELU, out=out)
            acts.append(z)
            li += 1
        cat_d = torch.empty((m, hd + de_w), dtype=F32, device=dev)  # [z, d_emb] (model.py:58)
        ops.sinusoidal_emb_into(d, self.d_freqs, cat_d, hd)
        z = cat_x
        for i in range(self.mid_layers):  # model.py:53-56: relu precedes Dense for i > 0
            last = i == self.mid_layers - 1
            z = ops.dense_fwd(z, W[li][0], W[li][1], L.ACT_NONE if last else L.ACT_RELU,
                              out=cat_d[:, :hd] if last else None)
            acts.append(z)
            li += 1
        density = ops.dense_fwd(cat_d[:, :hd], W[li][0], W[li][1], L.ACT_SOFTPLUS)  # model.py:57
        li += 1
        h_col = ops.dense_fwd(cat_d, W[li][0], W[li][1], L.ACT_RELU)  # model.py:59
        li += 1
        rgb = ops.dense_fwd(h_col, W[li][0], W[li][1], L.ACT_TANH)  # model.py:60
        ctx = None
        if save:
            ctx = dict(kind="dense", flat=flat, cat_x=cat_x, cat_d=cat_d, acts=acts, h_col=h_col,
                       density=density, rgb=rgb)
        return density.view(-1), rgb, {}, ctx

    @ops.uses_model_precision
    def _dense_bwd(self, ctx, g_density, g_rgb, grad_flat):
        tree, gtree = self.tree(ctx["flat"]), self.tree(grad_flat)
        nl = len(self.layer_dims())
        W = [tree[f"Dense_{i}"]["kernel"] for i in range(nl)]
        G = [(gtree[f"Dense_{i}"]["kernel"], gtree[f"Dense_{i}"]["bias"]) for i in range(nl)]
        hd = self.hidden_dim
        cat_x, cat_d, acts, h_col = ctx["cat_x"], ctx["cat_d"], ctx["acts"], ctx["h_col"]
        i_rgb, i_col, i_den = nl - 1, nl - 2, nl - 3
        gy = ops.act_bwd_(g_rgb.reshape(-1, 3).clone(), ctx["rgb"], L.ACT_TANH)
        ops.dense_bwd_weight(h_col, gy, *G[i_rgb])
        gy = ops.dense_bwd_input(gy, W[i_rgb], gate=h_col)  # ReLU backward of Dense_10 fused in
        ops.dense_bwd_weight(cat_d, gy, *G[i_col])
        gz = ops.dense_bwd_input(gy, W[i_col][:hd])
        gd = ops.act_bwd_(g_density.reshape(-1, 1).clone(), ctx["density"], L.ACT_SOFTPLUS)
        z_view = cat_d[:, :hd]
        ops.dense_bwd_weight(z_view, gd, *G[i_den])
        ops.dense_bwd_input(gd, W[i_den], out=gz, accumulate=True)
        gy = gz  # Dense_{last mid} output is linear
        li = i_den - 1
        for i in reversed(range(self.mid_layers)):
            inp = cat_x if i == 0 else acts[self.input_layers + i - 1]
            ops.dense_bwd_weight(inp, gy, *G[li])
            prev = cat_x[:, :hd] if i == 0 else acts[self.input_layers + i - 1]
            gy = ops.dense_bwd_input(gy, W[li][:hd], gate=prev)  # input gradient + ReLU backward of the layer below
            li -= 1
        for i in reversed(range(self.input_layers)):
            inp = cat_x[:, hd:] if i == 0 else acts[i - 1]
            ops.dense_bwd_weight(inp, gy, *G[li])
            if i > 0:
                gy = ops.dense_bwd_input(gy, W[li], gate=acts[i - 1])
            li -= 1
