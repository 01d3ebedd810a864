"""
ctypes binding of liblnrf.so (C ABI declared in include/lnrf.h).

The product path has no CPU fallback: if the shared library is missing, or a kernel is
asked to run on a non-GPU tensor, this module raises.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_uint8, c_uint32, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEFAULT = os.path.normpath(os.path.join(_HERE, "..", "lib", "liblnrf.so"))

ERR_UNSUPPORTED = -3
ACT_NONE, ACT_RELU, ACT_SOFTPLUS, ACT_TANH, ACT_EXP, ACT_SIGMOID = range(6)


class HashGridDesc(ctypes.Structure):
    _fields_ = [("n_levels", c_int32), ("feature_dim", c_int32), ("smooth", c_int32), ("pad_", c_int32),
                ("bbox_min", c_float * 3), ("bbox_max", c_float * 3), ("grid_size", c_int32 * 32),
                ("table_size", c_int32 * 32), ("table_offset", c_int64 * 32), ("hashed", c_int32 * 32)]


class NerfShape(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in
                ("input_layers", "mid_layers", "hidden_dim", "color_layer_dim", "x_freqs", "d_freqs")]


class NgpMlpDesc(ctypes.Structure):
    _fields_ = [("enc_dim", c_int32), ("hidden_dim", c_int32), ("density_dim", c_int32),
                ("density_layers", c_int32), ("color_layers", c_int32), ("d_freqs", c_int32),
                ("dense_offset", c_int64)]


_P = c_void_p
_F3 = POINTER(c_float)

# name -> (restype, argtypes); must list every symbol of include/lnrf.h (tests check this)
PROTOTYPES = {
    "lnrf_version": (c_int32, []),
    "lnrf_last_error": (c_char_p, []),
    "lnrf_ray_aabb_stratified": (c_int32, [_P, c_int64, c_int64, _F3, _F3, c_float, c_float, c_int32, _P,
                                           c_uint64, c_uint32, c_int64, _P, _P, _P, _P, _P]),
    "lnrf_camera_rays": (c_int32, [_F3, _F3, _F3, _F3, c_float, c_float, c_int32, c_int32, _P, _P]),
    "lnrf_gather_rows": (c_int32, [_P, c_int64, c_int32, _P, c_int64, _P, _P]),
    "lnrf_stratified": (c_int32, [_P, _P, c_int64, c_int32, _P, c_uint64, c_uint32, c_int64, _P, _P]),
    "lnrf_ray_points": (c_int32, [_P, c_int64, _P, c_int64, c_int32, _P, _P, _P]),
    "lnrf_fine_sample": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, c_int32, c_float, c_int32, _P,
                                   c_uint64, c_uint32, c_int64, _P, _P]),
    "lnrf_bin_edges": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P, _P]),
    "lnrf_termination_probs": (c_int32, [_P, _P, _P, _P, c_int64, c_int32, _P, _P]),
    "lnrf_composite_fwd": (c_int32, [_P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int64, c_int32,
                                     _P, _P, _P, _P, _P, c_int64, _P, _P]),
    "lnrf_composite_bwd": (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int64, c_int32, _P, _P, _P,
                                     c_int64, c_float, _F3, _P, _P, _P, _P, _P]),
    "lnrf_composite_bwd_scratch_bytes": (c_int64, [c_int64]),
    "lnrf_composite_bwd_det": (c_int32, [_P, _P, _P, _P, _P, _P, _P, c_int32, _P, c_int64, c_int32, _P, _P, _P,
                                         c_int64, c_float, _F3, _P, _P, _P, _P, _P, c_int64, _P]),
    "lnrf_sinusoidal_emb": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, c_int64, _P]),
    "lnrf_dense_fwd": (c_int32, [_P, c_int64, _P, _P, c_int32, _P, c_int64, c_int64, c_int32, c_int32, _P]),
    "lnrf_act_bwd": (c_int32, [_P, c_int64, _P, c_int64, c_int32, c_int64, c_int32, _P]),
    "lnrf_dense_bwd_input": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int32, c_int64, c_int32, c_int32, _P]),
    "lnrf_dense_bwd_weight": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int32, c_int32, _P]),
    "lnrf_gemm_f32_det_scratch_bytes": (c_int64, [c_int64, c_int32, c_int64]),
    "lnrf_gemm_f32_det": (c_int32, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int32, c_int64, _P, c_int64, _P]),
    "lnrf_dense_bwd_weight_scratch_bytes": (c_int64, [c_int64, c_int32, c_int32]),
    "lnrf_dense_bwd_weight_det": (c_int32, [_P, c_int64, _P, c_int64, _P, _P, c_int64, c_int32, c_int32, _P, c_int64, _P]),
    "lnrf_gemm_f32": (c_int32, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, _P, c_int32, c_int32,
                                c_int64, c_int32, c_int64, c_int32, _P]),
    "lnrf_hashgrid_fwd": (c_int32, [POINTER(HashGridDesc), _P, _P, c_int64, _P, _P]),
    "lnrf_hashgrid_bwd": (c_int32, [POINTER(HashGridDesc), _P, c_int64, _P, _P, _P]),
    "lnrf_hashgrid_bwd_scratch_bytes": (c_int64, [POINTER(HashGridDesc), c_int64]),
    "lnrf_hashgrid_bwd_bucketed": (c_int32, [POINTER(HashGridDesc), _P, _P, c_int64, _P, _P, _P, _P, c_int64, _P]),
    "lnrf_hashgrid_jvp": (c_int32, [POINTER(HashGridDesc), _P, _P, _P, c_int64, _P, _P]),
    "lnrf_hashgrid_input_grad": (c_int32, [POINTER(HashGridDesc), _P, _P, c_int64, _P, _P, _P]),
    "lnrf_hashgrid_bwd_dir": (c_int32, [POINTER(HashGridDesc), _P, _P, c_int64, _P, _P, _P]),
    "lnrf_sinusoidal_emb_bwd": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, c_int64, c_int64, _P, _P]),
    "lnrf_sinusoidal_emb_jvp": (c_int32, [_P, c_int64, c_int64, c_int32, c_int32, _P, _P, c_int64, c_int64, _P]),
    "lnrf_integrated_directional_encoding": (c_int32, [c_int32, _P, _P, c_int64, _P, _P]),
    "lnrf_refnerf_head_fwd": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int32, _P, _P, _P, _P, c_int64, _P, _P]),
    "lnrf_refnerf_head_bwd": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int32, _P, _P, _P, _P, c_int64, _P, _P,
                                        c_int64, _P, _P]),
    "lnrf_refnerf_color_fwd": (c_int32, [_P, _P, _P, c_int64, _P, _P]),
    "lnrf_refnerf_color_bwd": (c_int32, [_P, _P, _P, c_int64, _P, _P, _P, _P, _P]),
    "lnrf_refnerf_trunk_packed_bytes": (c_int64, []),
    "lnrf_refnerf_trunk_pack": (c_int32, [_P, _P, _P]),
    "lnrf_refnerf_trunk_fwd": (c_int32, [_P, _P, c_int64, _P, _P, c_int64, _P]),
    "lnrf_refnerf_normal_pass": (c_int32, [_P, _P, _P, c_int64, _P, _P, _P]),
    "lnrf_refnerf_trunk_bwd_scratch_bytes": (c_int64, [c_int64]),
    "lnrf_refnerf_trunk_bwd": (c_int32, [_P, _P, _P, c_int64, c_int64, _P, _P, _P]),
    "lnrf_refnerf_normal_bwd": (c_int32, [_P, _P, _P, _P, _P, c_int64, _P, _P, _P]),
    "lnrf_refnerf_dir_save_bytes": (c_int64, [c_int64]),
    "lnrf_refnerf_dir_scratch_bytes": (c_int64, [c_int64]),
    "lnrf_refnerf_dir_fwd": (c_int32, [_P, _P, c_int64, c_int64, _P, _P, _P]),
    "lnrf_refnerf_dir_bwd": (c_int32, [_P, _P, _P, c_int64, _P, _P, c_int64, _P, _P]),
    "lnrf_refnerf_render_packed_bytes": (c_int64, []),
    "lnrf_refnerf_render_pack": (c_int32, [_P, _P, _P]),
    "lnrf_refnerf_trunk_normal_split_scratch_bytes": (c_int64, [c_int64]),
    "lnrf_refnerf_trunk_normal_split": (c_int32, [_P, _P, c_int64, _P, c_int64, _P, _P, _P]),
    "lnrf_refnerf_dir_fwd_split": (c_int32, [_P, _P, c_int64, c_int64, _P, _P]),
    "lnrf_nerf_param_count": (c_int64, [POINTER(NerfShape)]),
    "lnrf_nerf_packed_bytes": (c_int64, [POINTER(NerfShape)]),
    "lnrf_nerf_save_bytes": (c_int64, [POINTER(NerfShape), c_int64]),
    "lnrf_nerf_bwd_scratch_bytes": (c_int64, [POINTER(NerfShape), c_int64]),
    "lnrf_nerf_pack_weights": (c_int32, [POINTER(NerfShape), _P, _P, _P]),
    "lnrf_nerf_mlp_fwd": (c_int32, [POINTER(NerfShape), _P, _P, _P, _P, c_int64, _P, c_int32, c_int64, _P, _P,
                                    _P, _P]),
    "lnrf_nerf_mlp_fwd_ls": (c_int32, [POINTER(NerfShape), _P, _P, _P, _P, c_int64, _P, c_int32, c_int64, _P, _P,
                                       _P, _P]),
    "lnrf_nerf_packed_split_bytes": (c_int64, [POINTER(NerfShape)]),
    "lnrf_nerf_pack_weights_split": (c_int32, [POINTER(NerfShape), _P, _P, _P]),
    "lnrf_nerf_mlp_fwd_split": (c_int32, [POINTER(NerfShape), _P, _P, _P, _P, c_int64, _P, c_int32, c_int64, _P, _P,
                                          _P]),
    "lnrf_nerf_mlp_bwd": (c_int32, [POINTER(NerfShape), _P, _P, _P, _P, _P, _P, c_int64, _P, _P, _P]),
    "lnrf_nerf_mlp_bwd_chain": (c_int32, [POINTER(NerfShape), _P, _P, _P, _P, _P, _P, c_int64, _P, _P]),
    "lnrf_nerf_mlp_bwd_weights": (c_int32, [POINTER(NerfShape), _P, _P, c_int64, _P, _P]),
    "lnrf_nerf_bwd_ls_scratch_bytes": (c_int64, [POINTER(NerfShape), c_int64]),
    "lnrf_nerf_bwd_ls_status_offset": (c_int64, [POINTER(NerfShape), c_int64]),
    "lnrf_nerf_mlp_bwd_ls": (c_int32, [POINTER(NerfShape), _P, _P, _P, _P, _P, _P, c_int64, _P, _P, c_int32, _P]),
    "lnrf_nerf_mlp_bwd_ls2": (c_int32, [POINTER(NerfShape), _P, _P, _P, _P, _P, _P, c_int64, _P, _P,
                                        _P, _P, _P, _P, _P, _P, c_int64, _P, _P, c_int32, _P]),
    "lnrf_ngp_mlp_packed_bytes": (c_int64, [POINTER(NgpMlpDesc)]),
    "lnrf_ngp_mlp_scratch_bytes": (c_int64, [POINTER(NgpMlpDesc), c_int64]),
    "lnrf_ngp_mlp_pack": (c_int32, [POINTER(NgpMlpDesc), _P, _P, _P]),
    "lnrf_ngp_mlp_fwd": (c_int32, [POINTER(NgpMlpDesc), _P, _P, _P, c_int64, _P, _P, _P]),
    "lnrf_ngp_mlp_packed_split_bytes": (c_int64, [POINTER(NgpMlpDesc)]),
    "lnrf_ngp_mlp_pack_split": (c_int32, [POINTER(NgpMlpDesc), _P, _P, _P]),
    "lnrf_ngp_mlp_fwd_split": (c_int32, [POINTER(NgpMlpDesc), _P, _P, _P, c_int64, _P, _P, _P]),
    "lnrf_ngp_mlp_bwd": (c_int32, [POINTER(NgpMlpDesc), _P, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, _P]),
    "lnrf_dense_bwd_input_gated": (c_int32, [_P, c_int64, _P, _P, c_int64, c_int32, c_int32, _P, c_int64, c_int32,
                                             c_int64, c_int32, c_int32, _P]),
    "lnrf_dense_fwd_gated": (c_int32, [_P, c_int64, _P, _P, c_int32, _P, c_int64, c_int32, _P, c_int64, c_int64,
                                       c_int32, c_int32, _P]),
    "lnrf_set_dense_precision": (c_int32, [c_int32]),
    "lnrf_get_dense_precision": (c_int32, []),
    "lnrf_comm_get_unique_id": (c_int32, [_P]),
    "lnrf_comm_init": (c_int32, [_P, c_int32, c_int32, POINTER(c_void_p)]),
    "lnrf_comm_allreduce": (c_int32, [_P, _P, c_int64, _P]),
    "lnrf_comm_info": (c_int32, [_P, POINTER(c_int32), POINTER(c_int32)]),
    "lnrf_comm_destroy": (c_int32, [_P]),
    "lnrf_adam_step": (c_int32, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int32,
                                 c_float, _P]),
    "lnrf_adam_step_norms": (c_int32, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_int32,
                                       c_float, _P, _P]),
    "lnrf_step_log": (c_int32, [_P, c_float, c_float, c_int32, _P, _P]),
    "lnrf_sq_norm": (c_int32, [_P, c_int64, _P, _P]),
}

_lib = None


def library_path() -> str:
    return os.environ.get("LNRF_LIB", _DEFAULT)


def lib() -> ctypes.CDLL:
    """Load liblnrf.so once; raises if it is missing (there is no CPU fallback)."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"liblnrf.so not found at {path}: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (or `make -C learn-nerf_amd/csrc`). The HIP library is required."
            )
        handle = ctypes.CDLL(path)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().lnrf_last_error()
        raise RuntimeError(f"lnrf error {rc} {what}: {msg.decode() if msg else ''}")


def ptr(t, dtype=torch.float32):
    """Device pointer of a contiguous GPU tensor (None -> NULL)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError("lnrf kernels need tensors on a ROCm GPU device (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def f3(values):
    vals = [float(v) for v in values]
    assert len(vals) == 3
    return (c_float * 3)(*vals)
