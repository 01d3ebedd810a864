"""
learn_nerf.train — TrainLoop, default_loss_weights (reference: learn_nerf/train.py).

One step = TrainLoop.step_fn's jitted function (train.py:85-106) restated as an explicit
sequence of HIP kernel families on one stream:
  ray/bbox + stratified -> coarse MLP -> composite -> hierarchical resample -> fine MLP ->
  composite -> [loss gradient fused into] composite backward -> MLP backward (both models) ->
  [RCCL all-reduce of the flat gradient when torch.distributed is initialised] ->
  grad/param norms -> fused Adam.
Parameters {coarse, fine, background} are views of ONE flat fp32 buffer; so are the gradients
and the Adam moments.
"""
import os
from dataclasses import dataclass
from typing import Any, Callable, Dict, Optional, Sequence

import numpy as np
import torch

from . import _prof
from . import ops
from . import parallel
from .model import ModelBase
from .params import ParamTree, as_generator, build_tree, default_device, split_seed
from .render import STREAM_COARSE, STREAM_FINE, NeRFRenderer, _vec3
from .rng import Key, KeyLike, Uniforms, as_key, sampler_args, split

F32 = torch.float32


def _dist():
    import torch.distributed as dist

    return dist if (dist.is_available() and dist.is_initialized()) else None


@dataclass
class TrainState:
    """Minimal stand-in for flax.training.train_state.TrainState (train.py:51-60)."""

    params: Dict[str, Any]
    step: int
    opt_m: torch.Tensor
    opt_v: torch.Tensor


class TrainLoop:
    """
    A stateful training loop (train.py:17-184).  Constructor arguments as in the reference;
    ``init_rng`` is an int seed / torch.Generator instead of a jax key.
    """

    def __init__(
        self,
        coarse: ModelBase,
        fine: ModelBase,
        init_rng,
        lr: float,
        coarse_ts: int,
        fine_ts: int,
        adam_b1: float = 0.9,
        adam_b2: float = 0.999,
        adam_eps: float = 1e-7,
        loss_weights: Dict[str, float] = None,
        density_penalty: Optional[float] = None,
        density_penalty_batch_size: int = 128,
        device=None,
    ):
        self.coarse = coarse
        self.fine = fine
        if hasattr(coarse, "tag") and hasattr(fine, "tag") and coarse is not fine:
            coarse.tag, fine.tag = "coarse", "fine"
        self.coarse_ts = coarse_ts
        self.fine_ts = fine_ts
        self.lr, self.adam_b1, self.adam_b2, self.adam_eps = lr, adam_b1, adam_b2, adam_eps
        self.loss_weights = loss_weights if loss_weights is not None else default_loss_weights()
        self.density_penalty = density_penalty
        self.density_penalty_batch_size = density_penalty_batch_size
        self.device = device if device is not None else default_device()

        # train.py:47-58: split the init key, init both models, background = all black (-1,-1,-1)
        seed = as_generator(init_rng).initial_seed()
        coarse_seed, fine_seed = split_seed(seed, 2)
        self.n_coarse, self.n_fine = coarse.num_params(), fine.num_params()
        total = self.n_coarse + self.n_fine + 3
        host = torch.zeros(total, dtype=F32)
        coarse.init_flat_(host[: self.n_coarse], as_generator(coarse_seed))
        fine.init_flat_(host[self.n_coarse: self.n_coarse + self.n_fine], as_generator(fine_seed))
        host[-3:] = -1.0
        self.flat = host.to(self.device)
        self.grad = torch.zeros_like(self.flat)
        self.state = TrainState(params=self._views(self.flat), step=0, opt_m=torch.zeros_like(self.flat),
                                opt_v=torch.zeros_like(self.flat))
        self._scalars = torch.zeros(8, dtype=F32, device=self.device)
        self._scalars_dirty = False  # lnrf_step_log leaves the accumulators zeroed for the next step
        # The coarse model's backward depends on nothing of the fine pass (fine sampling carries no gradient,
        # render.py:211-257), so it runs on a second HIP stream beside the fine forward / backward and is joined
        # before the optimizer (measured: Ref-NeRF 11.03 -> 10.79 ms per step; NeRFModel and the hash-grid model opt
        # out through overlap_backward_hint: 5.015 -> 5.002 ms is within noise and doubles the per-kernel timings,
        # hash grid 2.30 -> 2.33 ms).  LNRF_OVERLAP_BACKWARD=0 / 1 overrides.
        env = os.environ.get("LNRF_OVERLAP_BACKWARD")
        want = (env != "0") if env is not None else bool(getattr(coarse, "overlap_backward_hint", True))
        self.overlap_backward = torch.device(self.device).type == "cuda" and coarse is not fine and want
        self._side_stream = None
        self._early_reduce = (0, None)  # (floats already handed to the all-reduce, its handle) of the running step

    def _params_changed(self):
        # the kernels write through raw pointers, which torch's version counters do not see
        for mdl in (self.coarse, self.fine):
            if hasattr(mdl, "invalidate_packed"):
                mdl.invalidate_packed()

    # ---- parameter views ------------------------------------------------------------------------
    def _slices(self, flat):
        a, b = self.n_coarse, self.n_coarse + self.n_fine
        return flat[:a], flat[a:b], flat[b:b + 3]

    def _views(self, flat) -> Dict[str, Any]:
        c, f, bg = self._slices(flat)
        return dict(coarse=self.coarse.tree(c), fine=self.fine.tree(f), background=bg)

    # ---- checkpointing (train.py:62-76) ------------------------------------------------------------
    def save(self, path: str):
        """
        Save the model parameters to a file (atomic rename like train.py:62-69).  Besides the reference's
        params tree {"coarse", "fine", "background"} the file also keeps the Adam moments and step count
        (the reference drops them, so every resume restarts Adam's bias correction).
        The container is a NumPy .npz archive (arrays only, read back with allow_pickle=False) whatever the
        file is called — the reference pickles a jax pytree, which needs JAX to read and executes code on load.
        Keys: "format", "background", "opt/step", "opt/m", "opt/v" and "<model>/<module path>/<leaf>".
        """
        tmp_path = path + ".tmp"
        c, f, bg = self._slices(self.flat)
        arrays = {"format": np.array("lnrf-params-v3"), "background": bg.detach().cpu().numpy(),
                  "opt/step": np.array(self.state.step, dtype=np.int64),
                  "opt/m": self.state.opt_m.detach().cpu().numpy(), "opt/v": self.state.opt_v.detach().cpu().numpy()}
        for name in ("coarse", "fine"):
            for key, leaf in _flatten_tree(self.state.params[name]):
                arrays[f"{name}/{key}"] = leaf.detach().cpu().numpy()
        with open(tmp_path, "wb") as fh:
            np.savez(fh, **arrays)
        os.rename(tmp_path, path)

    def load(self, path: str, load_optimizer: bool = True):
        """Load parameters (and, when present, the optimiser state) from a file written by save()."""
        with np.load(path, allow_pickle=False) as blob:
            if str(blob["format"]) != "lnrf-params-v3":
                raise ValueError(f"{path}: not an lnrf-params-v3 checkpoint")
            c, f, bg = self._slices(self.flat)
            for name, model, dst in (("coarse", self.coarse, c), ("fine", self.fine, f)):
                leaves = [torch.from_numpy(blob[f"{name}/{module}/{leaf}"]).to(F32).reshape(-1)
                          for module, leaf, _ in model.param_spec()]
                dst.copy_(torch.cat(leaves).to(self.device))
            bg.copy_(torch.from_numpy(blob["background"]).to(F32).to(self.device))
            if load_optimizer and "opt/m" in blob and blob["opt/m"].shape[0] == self.flat.numel():
                self.state.opt_m.copy_(torch.from_numpy(blob["opt/m"]).to(self.device))
                self.state.opt_v.copy_(torch.from_numpy(blob["opt/v"]).to(self.device))
                self.state.step = int(blob["opt/step"])
        self._params_changed()

    # ---- the step --------------------------------------------------------------------------------
    def step_fn(self, bbox_min, bbox_max) -> Callable[[KeyLike, torch.Tensor], Dict[str, torch.Tensor]]:
        """Create a function that steps in place and returns a logging dict (train.py:78-112)."""
        bmin, bmax = _vec3(bbox_min), _vec3(bbox_max)

        def in_place_step(key: KeyLike, batch: torch.Tensor) -> Dict[str, torch.Tensor]:
            return self._step(key, bmin, bmax, batch)

        return in_place_step

    def _forward_backward(self, key, bmin, bmax, batch, params_flat, grad_flat, want_grad: bool):
        """losses (train.py:114-165) and, if want_grad, d total / d params accumulated into grad_flat."""
        n = batch.shape[0]
        dist = _dist()
        world = dist.get_world_size() if dist else 1
        c_flat, f_flat, bg = self._slices(params_flat)
        sc = self._scalars  # [sum sq err coarse, sum sq err fine, sum g^2, sum p^2, ...]: zero on entry
        if self._scalars_dirty:
            sc.zero_()
        self._scalars_dirty = True
        render_key, density_key = split(key, 2)  # train.py:137
        coarse_key, fine_key = split(render_key, 2)  # render.py:55
        targets = batch[:, 2]

        t_min, t_max, mask, ts_c = ops.ray_aabb_stratified(batch, bmin, bmax, self.coarse_ts,
                                                           **sampler_args(coarse_key, STREAM_COARSE))
        dens_c, rgb_c, aux_c, ctx_c = self.coarse.forward_rays(c_flat, batch, ts_c, save=want_grad)
        names_c = list(aux_c.keys())
        auxs_c = torch.stack([aux_c[k] for k in names_c], -1).contiguous() if names_c else None
        out_c, _, _, asum_c = ops.composite_fwd(None, ts_c, t_min, t_max, mask, dens_c, rgb_c, bg, aux=auxs_c,
                                                targets=targets, sq_err=sc[0:1], want_coords=False)
        ts_f = ops.fine_sample(ts_c, t_min, t_max, dens_c, self.fine_ts, **sampler_args(fine_key, STREAM_FINE))

        inv = 1.0 / (3.0 * n)
        out_scale = 2.0 * inv  # d mean((out-t)^2) / d out
        gc, gf, gbg = self._slices(grad_flat) if want_grad else (None, None, None)

        def output_grads(ts, dens, rgb, out, names, auxs):
            gw = [self.loss_weights[k] / n for k in names]
            gd, grgb, gaux = ops.composite_bwd(ts, t_min, t_max, mask, dens, rgb, bg, gbg, outputs=out,
                                               targets=targets, out_scale=out_scale, aux=auxs, g_aux_w=gw)
            g_aux = {k: gaux[..., i] for i, k in enumerate(names)} if names else None
            return gd, grgb, g_aux

        def backward_of(ts, dens, rgb, out, model, ctx, gslice, names, auxs):
            gd, grgb, g_aux = output_grads(ts, dens, rgb, out, names, auxs)
            model.backward(ctx, gd, grgb, g_aux, gslice)

        coarse_args = (ts_c, dens_c, rgb_c, out_c, self.coarse, ctx_c, gc, names_c, auxs_c)
        coarse_done = None
        if want_grad and self.overlap_backward:
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            main = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self._side_stream):
                self._side_stream.wait_event(ready)
                backward_of(*coarse_args)
                coarse_done = torch.cuda.Event()
                coarse_done.record(self._side_stream)
        dens_f, rgb_f, aux_f, ctx_f = self.fine.forward_rays(f_flat, batch, ts_f, save=want_grad)
        names_f = list(aux_f.keys())
        auxs_f = torch.stack([aux_f[k] for k in names_f], -1).contiguous() if names_f else None
        out_f, _, _, asum_f = ops.composite_fwd(None, ts_f, t_min, t_max, mask, dens_f, rgb_f, bg, aux=auxs_f,
                                                targets=targets, sq_err=sc[1:2], want_coords=False)

        # train.py:141-144; in a training step the two means are produced by lnrf_step_log together with the norms
        loss_dict = {} if want_grad else dict(coarse=sc[0] * inv, fine=sc[1] * inv)
        for prefix, names, asum in (("coarse", names_c, asum_c), ("fine", names_f, asum_f)):
            if names:
                means = asum.mean(dim=0)
                for i, k in enumerate(names):
                    loss_dict[f"{prefix}_{k}"] = means[i]  # train.py:146-151

        if want_grad and coarse_done is None and _ls_pair(self.coarse, ctx_c, self.fine, ctx_f):
            # both models are fused NeRFModels on the layer-stationary backward: one persistent launch for the two
            from .model import backward_ls_pair

            gd_c, grgb_c, _ = output_grads(ts_c, dens_c, rgb_c, out_c, names_c, auxs_c)
            gd_f, grgb_f, _ = output_grads(ts_f, dens_f, rgb_f, out_f, names_f, auxs_f)
            backward_ls_pair(ctx_c, gd_c, grgb_c, gc, ctx_f, gd_f, grgb_f, gf)
        elif want_grad:
            if coarse_done is None:
                backward_of(*coarse_args)
                if self.density_penalty is None:
                    # the coarse model's gradient is final: its all-reduce runs under the fine backward (data parallel)
                    self._early_reduce = (self.n_coarse, parallel.begin_reduce_(gc))
            backward_of(ts_f, dens_f, rgb_f, out_f, self.fine, ctx_f, gf, names_f, auxs_f)
            if coarse_done is not None:
                torch.cuda.current_stream(self.device).wait_event(coarse_done)  # join before anything reads the gradient

        if self.density_penalty is not None:  # train.py:153-163
            gsl = self._slices(grad_flat) if want_grad else (None, None, None)
            for prefix, model, pflat, gslice in (("fine", self.fine, f_flat, gsl[1]),
                                                 ("coarse", self.coarse, c_flat, gsl[0])):
                loss_dict[f"{prefix}_density"] = self._average_density(density_key, model, pflat, bmin, bmax,
                                                                       gslice if want_grad else None)
        return loss_dict, world

    def _average_density(self, key, model, pflat, bmin, bmax, gslice):
        """TrainLoop.average_density (train.py:167-184): mean density at random bbox points."""
        b = self.density_penalty_batch_size
        k = as_key(key)
        gen = torch.Generator(device=self.device)
        gen.manual_seed((k.seed if isinstance(k, Key) else 0) & 0x7FFFFFFFFFFFFFFF)
        lo = torch.tensor(bmin, dtype=F32, device=self.device)
        hi = torch.tensor(bmax, dtype=F32, device=self.device)
        coords = torch.rand((b, 3), generator=gen, device=self.device) * (hi - lo) + lo
        dirs = torch.randn((b, 3), generator=gen, device=self.device)
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        dens, rgb, aux, ctx = model.forward_points(pflat, coords.contiguous(), dirs.contiguous(),
                                                   save=gslice is not None)
        if gslice is not None:
            g_d = torch.full_like(dens, self.density_penalty / b)
            g_aux = {kk: torch.zeros_like(v) for kk, v in aux.items()} if aux else None
            model.backward(ctx, g_d, torch.zeros_like(rgb), g_aux, gslice)
        return dens.mean()

    def _step(self, key, bmin, bmax, batch):
        self.grad.zero_()
        self._early_reduce = (0, None)
        aux_losses, _ = self._forward_backward(key, bmin, bmax, batch.contiguous(), self.flat, self.grad, True)
        self.state.step += 1
        sc = self._scalars
        prefix, pending = self._early_reduce
        scale = apply_gradients(self.flat, self.grad, self.state.opt_m, self.state.opt_v, self.state.step, self.lr,
                                self.adam_b1, self.adam_b2, self.adam_eps, sq_norms=sc[2:4],
                                reduced_prefix=prefix, pending=pending)
        log = ops.step_log(sc, 1.0 / (3.0 * batch.shape[0]), scale, clear=True)  # one launch; zeroes sc again
        self._scalars_dirty = False
        self._params_changed()
        loss_dict = dict(coarse=log[0], fine=log[1])  # train.py:141-144
        loss_dict.update(aux_losses)
        loss_dict.update(grad_norm=log[2], param_norm=log[3])  # train.py:99-104
        return loss_dict

    def losses(self, key: KeyLike, bbox_min, bbox_max, batch: torch.Tensor, params=None):
        """
        Compute losses and a logging dict for a given batch (train.py:114-165), no gradient.
        Returns (total_loss, loss_dict) like the reference.
        """
        flat = self.flat if params is None else self._flat_from_params(params)
        loss_dict, _ = self._forward_backward(key, _vec3(bbox_min), _vec3(bbox_max), batch.contiguous(), flat,
                                              None, False)
        total = loss_dict["coarse"] + loss_dict["fine"]
        for k, v in loss_dict.items():
            for prefix in ("coarse_", "fine_"):
                if k.startswith(prefix):
                    name = k[len(prefix):]
                    if name in self.loss_weights:
                        total = total + self.loss_weights[name] * v
                    elif name == "density" and self.density_penalty is not None:
                        total = total + self.density_penalty * v
        return total, loss_dict

    def _flat_from_params(self, params) -> torch.Tensor:
        if params is self.state.params:
            return self.flat
        return torch.cat([self.coarse.flat(params["coarse"]).reshape(-1), self.fine.flat(params["fine"]).reshape(-1),
                          torch.as_tensor(params["background"], dtype=F32, device=self.device).reshape(-1)])

    def average_density(self, key, model, params, bbox_min, bbox_max) -> torch.Tensor:
        return self._average_density(key, model, model.flat(params), _vec3(bbox_min), _vec3(bbox_max), None)


def _ls_pair(coarse, ctx_c, fine, ctx_f) -> bool:
    """Both contexts come from fused NeRFModels that ask for the layer-stationary backward and hold evaluations."""
    from .model import NeRFModel

    return all(type(mdl) is NeRFModel and mdl.backward_kernel == "ls" and isinstance(ctx, dict)
               and ctx.get("kind") == "fused" and ctx["m"] > 0 for mdl, ctx in ((coarse, ctx_c), (fine, ctx_f)))


def apply_gradients(flat, grad, opt_m, opt_v, step: int, lr: float, b1: float, b2: float, eps: float,
                    sq_norms: torch.Tensor, kernels=ops, reduced_prefix: int = 0, pending=None) -> float:
    """
    What follows jax.grad in the reference's step (train.py:99-106), in its data-parallel form:
      1. the all-reduce (sum) of the flat gradient over the process group (parallel.reduce_gradient_): one collective,
         or two when the caller has already started the coarse model's slice with parallel.begin_reduce_ while the fine
         backward was running (`reduced_prefix` floats, handle `pending`) — the same sums either way,
      2. optax.adam as at train.py:59 with the mean over ranks folded in as grad_scale = 1/world, and in the same
         pass over the buffers
      3. the tree_norm numerators (train.py:92-104) of the REDUCED gradient and of the parameters before the update:
         sq_norms[0] += sum g^2 (of the summed gradient), sq_norms[1] += sum p^2.
    Returns scale = 1/world: grad_norm = sqrt(sq_norms[0]) * scale, param_norm = sqrt(sq_norms[1]).
    `kernels` supplies adam_step_(p, g, m, v, lr, b1, b2, eps, step, grad_scale=..., sq_norms=...): learn_nerf.ops
    (HIP) in the product; tests/test_dp_gloo.py passes a CPU implementation to drive this same sequence under gloo.
    """
    with _prof.section("allreduce"):
        scale = parallel.reduce_gradient_(grad, reduced_prefix, pending)
    with _prof.section("norms_adam"):
        kernels.adam_step_(flat, grad, opt_m, opt_v, lr, b1, b2, eps, step, grad_scale=scale, sq_norms=sq_norms)
    return scale


def load_params(path: str, coarse: ModelBase, fine: ModelBase, device) -> Dict[str, Any]:
    """The params tree {"coarse", "fine", "background"} of a checkpoint written by TrainLoop.save, on `device`
    (what scripts/render_nerf.py:56-58 gets from pickle.load in the reference)."""
    out = {}
    with np.load(path, allow_pickle=False) as blob:
        if str(blob["format"]) != "lnrf-params-v3":
            raise ValueError(f"{path}: not an lnrf-params-v3 checkpoint")
        for name, model in (("coarse", coarse), ("fine", fine)):
            leaves = [torch.from_numpy(blob[f"{name}/{module}/{leaf}"]).to(F32).reshape(-1)
                      for module, leaf, _ in model.param_spec()]
            out[name] = model.tree(torch.cat(leaves).to(device))
        out["background"] = torch.from_numpy(blob["background"]).to(F32).to(device)
    return out


def _flatten_tree(tree, prefix=""):
    """(path, leaf) pairs of a nested parameter dict, '/'-joined paths."""
    for k, v in tree.items():
        path = f"{prefix}/{k}" if prefix else k
        if isinstance(v, torch.Tensor):
            yield path, v
        else:
            yield from _flatten_tree(v, path)


def default_loss_weights() -> Dict[str, float]:
    return dict(normal_mse=3e-4, neg_normal=0.1)  # train.py:187-191
