"""ctypes binding of libnrhints_hip.so (C ABI: include/nrhints_hip.h).

There is deliberately NO fallback: if the library is missing or a call fails, the product path raises.
Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or ``make -C nrhints_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NRHINTS_HIP_LIB") or os.path.join(_HERE, "lib", "libnrhints_hip.so")

NRH_OK = 0
_ERRNAMES = {-1: "NRH_E_INVALID", -2: "NRH_E_LAUNCH", -3: "NRH_E_WORKSPACE", -4: "NRH_E_UNSUPPORTED"}

# every symbol include/nrhints_hip.h declares (tests check the .so exports exactly these)
EXPORTED = ("nrh_version", "nrh_build_info", "nrh_source_hash", "nrh_last_error_string", "nrh_param_sizes", "nrh_mlp_grid",
            "nrh_sdf_eval", "nrh_sampler_step", "nrh_color_eval", "nrh_render_workspace_floats",
            "nrh_render_forward", "nrh_kernel_timing_select", "nrh_kernel_timing_read", "nrh_generate_rays",
            "nrh_sdf_train_forward", "nrh_sdf_train_backward", "nrh_outside_sizes", "nrh_outside_forward",
            "nrh_outside_backward", "nrh_ray_adjoint", "nrh_render_forward_train", "nrh_alpha_train_forward", "nrh_alpha_train_backward",
            "nrh_shadow_alpha_forward", "nrh_shadow_alpha_backward", "nrh_alpha_train_forward_n", "nrh_alpha_train_backward_n",
            "nrh_sample_primary", "nrh_alpha_blend_forward", "nrh_alpha_blend_backward",
            "nrh_color_transposed_floats", "nrh_color_train_forward", "nrh_color_train_forward_grouped", "nrh_color_train_backward",
            "nrh_weight_norm_fold", "nrh_weight_norm_fold_backward", "nrh_sdf_eval_wide", "nrh_sdf_wide_stream_bytes", "nrh_sdf_eval_split", "nrh_sdf_grad_split",
            "nrh_generate_rays_indexed", "nrh_generate_rays_indexed_backward", "nrh_color_wide_stream_bytes", "nrh_color_eval_wide",
            "nrh_alpha_composite", "nrh_visibility", "nrh_color_composite", "nrh_sphere_trace", "nrh_sphere_trace_workspace_floats",
            "nrh_dw_workspace_floats", "nrh_dw_gemm", "nrh_embedding_rows", "nrh_composite_loss", "nrh_loss_finish",
            "nrh_alpha_train_backward_fused", "nrh_variance_grad", "nrh_pack_gather", "nrh_sdf32_tables", "nrh_adam_step", "nrh_fuse_feature_head",
            "nrh_train_arrays_tiled", "nrh_sdf_eval_wide_f16", "nrh_train_half_supported", "nrh_sdf_train_forward_half",
            "nrh_sdf_train_backward_half", "nrh_color_train_forward_half", "nrh_color_train_backward_half", "nrh_step_scalars", "nrh_sampler_fusion")


class NrhNet(Structure):
    _fields_ = [("sdf_w", c_void_p), ("sdf_b", c_void_p), ("sdf_head", c_void_p), ("col_w", c_void_p),
                ("col_b", c_void_p), ("inv_s", c_float), ("precision", c_int), ("hints", c_int),
                ("normal_type", c_int), ("depth_type", c_int), ("dyn_scalars", c_void_p),
                ("sdf_w32", c_void_p), ("sdf_tab32", c_void_p), ("feat_fused", c_int),
                ("col_w32", c_void_p), ("col_tab32", c_void_p), ("shadow_jvp", c_int), ("shadow_clip", c_int),
                ("samples", c_int), ("bg_alpha", c_void_p), ("tail_t", c_void_p), ("sampled_color", c_void_p),
                ("custom_consts", c_int), ("specular_roughness", ctypes.c_double * 4), ("shadow_ray_offset", ctypes.c_double),
                ("n_coarse", c_int), ("n_steps", c_int), ("n_new", c_int), ("s_coarse", c_int), ("s_new", c_int), ("lin_tables", c_void_p)]


class NrhAdamTensor(Structure):
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("step", c_void_p), ("n", c_longlong),
                ("group", c_int), ("reserved", c_int)]


class NrhTrainSaves(Structure):
    _fields_ = [("sdf", c_void_p), ("feat_rows", c_void_p), ("save_h", c_void_p), ("save_s1", c_void_p),
                ("save_t", c_void_p), ("save_ge", c_void_p), ("raymisc", c_void_p), ("shadow_mid_z", c_void_p),
                ("shadow_dists", c_void_p), ("vis_groups", c_void_p), ("save_h16", c_void_p), ("save_t16", c_void_p), ("pts", c_void_p)]


def adjoint_scale(n_rays: int) -> float:
    """``adj_scale`` of nrh_sdf_train_backward / nrh_color_train_backward / nrh_outside_backward for a batch of ``n_rays`` rays: the
    power of two that brings the adjoints of a loss normalised by the ray count (pipelines/base_pipeline.py:57-62) to the magnitude
    they have in a batch of about 8 rays, whatever the batch size.  The f16x3 kernels split every adjoint into fp16 halves with an
    absolute floor of 3e-11 below 6e-5: unscaled, a 1 024-ray step lost up to 6e-3 of a gradient tensor's scale against the
    reference's float64 step; the 40-ray fixtures (magnitudes of an unscaled 40-ray batch) and scales 64 .. 32 768 at 1 024 rays
    were all inside the bounds (profiles/r05/train1024_diag2.log).  8 leaves three more octaves of head-room to fp16's 65 504
    than a scale of n_rays would (the largest seed, d alpha / d sdf <= inv_s / 4 per unit of colour adjoint, grows with the
    trained sharpness).  Ignored by precision f32."""
    import math
    return float(2.0 ** max(0, round(math.log2(max(1, int(n_rays)) / 8.0))))


#: How the generic autograd Functions (SdfValueFeatGradHip, ColorNetHip, OutsideNetHip) choose ``adj_scale``.  False (default): the
#: 1 / rays convention of ``adjoint_scale`` - right for the reference's loss (pipelines/base_pipeline.py:57-62: normalised by the ray
#: count), no host synchronisation, and an eager step equals its hipGraph replay bit for bit.  True: from the incoming adjoints
#: themselves (one host read per backward) - for callers whose loss is NOT ~1 / rays (sum-reduced, custom weights): the f16x3 chains
#: then can neither underflow nor overflow fp16's 65 504 whatever the normalisation.
ADJOINT_SCALE_FROM_SEEDS = False


def adjoint_scale_from_seeds(seeds, n_rays: int) -> float:
    """``adj_scale`` for the generic autograd Functions.  With ``ADJOINT_SCALE_FROM_SEEDS``: the power of two that brings the LARGEST
    incoming adjoint to [8, 16) - what adjoint_range_kernel computes on the device for the 16-bit hand-offs; under hipGraph capture
    (no host read possible) and for empty or non-finite seeds, and by default, the 1 / rays convention of ``adjoint_scale``.
    Ignored by precision f32."""
    import math

    import torch
    if not ADJOINT_SCALE_FROM_SEEDS or torch.cuda.is_current_stream_capturing():
        return adjoint_scale(n_rays)
    mx = 0.0
    for t in seeds:
        if t is not None and t.numel():
            mx = max(mx, float(t.detach().abs().max()))
    if not (0.0 < mx < 3.0e38):          # zero, inf or NaN seeds: nothing to scale (NaN propagates into the gradients, loudly)
        return adjoint_scale(n_rays)
    e = 4 - math.frexp(mx)[1]            # mx = f 2^k, f in [0.5, 1)  ->  mx 2^(4 - k) in [8, 16)
    return float(2.0 ** max(-60, min(60, e)))


class HipExtensionMissing(RuntimeError):
    pass


class NrhError(RuntimeError):
    pass


_lib = None


class StaleLibrary(RuntimeError):
    pass


def _check_provenance(lib) -> None:
    """The binary must have been built from the sources beside it: ``nrh_source_hash()`` (embedded by csrc/Makefile) against
    build_id.source_hash() of the tree.  The .so is git-ignored and reaches the GPU box as a prebuilt file; this is what ties it
    to the tree.  A library named explicitly through NRHINTS_HIP_LIB (the -D experiment variants of ``make variant``) is exempt
    from the comparison - its identity is reported by bench.py - but must still carry a hash."""
    from . import build_id
    try:
        fn = lib.nrh_source_hash
    except AttributeError:
        raise StaleLibrary(f"{LIB_PATH} predates nrh_source_hash(): rebuild (make -C nrhints_amd/csrc)") from None
    fn.restype = c_char_p
    have, want = fn().decode(), build_id.source_hash()
    if have != want and not os.environ.get("NRHINTS_HIP_LIB"):
        raise StaleLibrary(f"{LIB_PATH} was built from other sources (embedded hash {have}, this tree {want}): "
                           "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")


def library_identity() -> dict:
    """{'embedded': hash in the loaded binary, 'tree': hash of the sources beside it, 'build_info': ...} for reports."""
    from . import build_id
    lib = load()
    return {"embedded": lib.nrh_source_hash().decode(), "tree": build_id.source_hash(), "build_info": lib.nrh_build_info().decode(),
            "path": os.path.relpath(LIB_PATH, os.path.dirname(_HERE))}


def load():
    """Load (once) and return the ctypes handle.  Raises HipExtensionMissing if the .so is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: the HIP hot path is not built (run __graft_entry__.build()). "
            "nrhints_amd has no CPU/PyTorch fallback for rendering.")
    lib = ctypes.CDLL(LIB_PATH)
    _check_provenance(lib)
    P = c_void_p
    lib.nrh_version.restype = c_int
    lib.nrh_train_arrays_tiled.restype = c_int
    lib.nrh_train_half_supported.argtypes = [c_int, c_longlong]
    lib.nrh_train_half_supported.restype = c_int
    lib.nrh_build_info.restype = c_char_p
    lib.nrh_source_hash.restype = c_char_p
    lib.nrh_last_error_string.restype = c_char_p
    lib.nrh_param_sizes.argtypes = [POINTER(c_int)]
    lib.nrh_mlp_grid.restype = c_int
    lib.nrh_sdf_eval.argtypes = [c_int, c_int, P, P, P, P, P, P, c_int, c_int, c_longlong, P, c_int, P, P, P, P]
    lib.nrh_sdf_train_forward.argtypes = [c_int, P, P, P, P, P, P, c_int, c_int, c_longlong, P, P, P, P, P, P, P, P]
    lib.nrh_sdf_train_forward_half.argtypes = [c_int, P, P, P, P, P, P, c_int, c_int, c_longlong, P, P, P, P, P, P, P, P, P, P]
    lib.nrh_ray_adjoint.argtypes = [P, P, P, P, P, P, P, P, c_int, P, c_longlong, P, P, P, P]
    lib.nrh_outside_sizes.argtypes = [POINTER(c_int)]
    lib.nrh_outside_forward.argtypes = [c_int, P, P, P, P, P, c_int, c_longlong, P, P, P, P, P, P, P, P]
    lib.nrh_outside_backward.argtypes = [c_int, P, P, P, P, P, P, c_longlong, P, P, P, P, P, c_float, P]
    lib.nrh_sdf_train_backward.argtypes = [c_int, P, P, P, P, P, P, c_int, c_int, c_longlong, P, P, P, P, P, P, P, P, P, P, c_float, P]
    lib.nrh_sdf_train_backward_half.argtypes = [c_int, P, P, P, P, P, P, c_int, c_int, c_longlong, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P]
    lib.nrh_alpha_train_forward.argtypes = [P, P, P, P, c_float, c_float, P, c_longlong, P, P, P]
    lib.nrh_alpha_train_backward.argtypes = [P, P, P, P, c_float, c_float, P, c_longlong, P, P, P, P, P, P, P]
    lib.nrh_alpha_train_forward_n.argtypes = [P, P, P, P, c_float, c_float, P, c_longlong, c_int, P, P, P]
    lib.nrh_alpha_train_backward_n.argtypes = [P, P, P, P, c_float, c_float, P, c_longlong, c_int, P, P, P, P, P, P, P]
    lib.nrh_sample_primary.argtypes = [POINTER(NrhNet), P, P, P, P, c_longlong, P, P, P, P, P, P, P, c_longlong, P]
    lib.nrh_alpha_blend_forward.argtypes = [P, P, P, P, P, P, c_float, c_float, P, c_longlong, P, P, P, P]
    lib.nrh_alpha_blend_backward.argtypes = [P, P, P, P, P, P, c_float, c_float, P, c_longlong, P, P, P, P, P, P, P, P, P]
    lib.nrh_shadow_alpha_forward.argtypes = [P, P, P, P, c_float, c_float, P, c_longlong, c_int, P, P]
    lib.nrh_shadow_alpha_backward.argtypes = [P, P, P, P, c_float, c_float, P, c_longlong, c_int, P, P, P, P, P, P]
    lib.nrh_color_transposed_floats.argtypes = [c_int]
    lib.nrh_color_transposed_floats.restype = c_longlong
    lib.nrh_color_train_forward.argtypes = [c_int, c_int, P, P, P, P, P, P, c_longlong, P, P, P, P]
    lib.nrh_color_train_forward_grouped.argtypes = [c_int, c_int, P, P, P, P, P, P, c_int, c_longlong, P, P, P, P]
    lib.nrh_color_train_backward.argtypes = [c_int, c_int, P, P, P, c_longlong, P, P, P, c_float, P]
    lib.nrh_color_train_forward_half.argtypes = [c_int, c_int, P, P, P, P, P, P, c_int, c_longlong, P, P, P, P, P]
    lib.nrh_color_train_backward_half.argtypes = [c_int, c_int, P, P, P, c_longlong, P, P, P, c_float, P, P, c_float, P]
    PP = POINTER(c_void_p)
    lib.nrh_weight_norm_fold.argtypes = [c_int, POINTER(c_int), POINTER(c_int), PP, PP, PP, P]
    lib.nrh_weight_norm_fold_backward.argtypes = [c_int, POINTER(c_int), POINTER(c_int), PP, PP, PP, PP, PP, P]
    lib.nrh_sdf_eval_wide.argtypes = [c_int, P, P, P, P, P, c_int, c_int, c_longlong, P, c_int, P, P, P, P]
    lib.nrh_sdf_eval_wide_f16.argtypes = lib.nrh_sdf_eval_wide.argtypes
    lib.nrh_sdf_wide_stream_bytes.restype = c_longlong
    lib.nrh_sdf_eval_split.argtypes = [P, P, P, P, P, P, c_int, c_int, c_longlong, P, c_int, c_int, P]
    lib.nrh_sdf_grad_split.argtypes = [P, P, P, P, P, P, c_int, c_int, c_longlong, P, c_int, P, P]
    lib.nrh_sampler_step.argtypes = [P, P, P, P, P, P, P, P, P, P, P, c_float, c_float, c_int, c_int, c_int, c_int,
                                     c_int, c_int, P]
    lib.nrh_color_eval.argtypes = [c_int, c_int, P, P, P, P, P, P, P, P, c_longlong, P, P]
    lib.nrh_render_workspace_floats.argtypes = [c_longlong]
    lib.nrh_render_workspace_floats.restype = c_longlong
    lib.nrh_render_forward.argtypes = [POINTER(NrhNet), P, P, P, P, P, c_longlong, P, c_float, P, P, c_int, P, P,
                                       P, P, P, P, P, P, P, P, P, P, P, P, P, c_longlong, P]
    lib.nrh_render_forward_train.argtypes = [POINTER(NrhNet), P, P, P, P, P, c_longlong, c_float, P, P, c_int, P, P,
                                             P, P, P, P, P, P, P, P, P, POINTER(NrhTrainSaves), P, c_longlong, P]
    lib.nrh_generate_rays.argtypes = [POINTER(c_float), POINTER(c_float), c_float, c_float, c_float, c_float, c_int,
                                      c_int, c_int, P, P, P, P, P, P]
    lib.nrh_generate_rays_indexed.argtypes = [P, P, P, P, c_int, P, c_longlong, P, P, c_int, c_float, c_float, c_float, c_float,
                                              c_int, c_float, c_float, P, P, P, P, P, P]
    lib.nrh_generate_rays_indexed_backward.argtypes = [P, P, P, P, c_int, P, c_longlong, P, P, c_int, c_float, c_float, c_float,
                                                       c_float, c_int, P, P, P, P, P, P, P, P]
    lib.nrh_color_wide_stream_bytes.restype = c_longlong
    lib.nrh_color_eval_wide.argtypes = [P, P, P, P, P, P, P, P, c_longlong, P, P]
    lib.nrh_alpha_composite.argtypes = [P, P, P, P, P, P, P, c_float, c_float, c_int, c_int, P, P, c_longlong] + [P] * 12
    lib.nrh_sphere_trace_workspace_floats.argtypes = [c_longlong]
    lib.nrh_sphere_trace_workspace_floats.restype = c_longlong
    lib.nrh_sphere_trace.argtypes = [POINTER(NrhNet), P, P, c_longlong, c_int, c_float, c_float, P, P, P, c_longlong, P]
    lib.nrh_visibility.argtypes = [P, P, P, P, P, P, P, c_float, c_float, c_int, c_longlong, P, P, P]
    lib.nrh_color_composite.argtypes = [P, P, P, P, P, P, P, c_longlong, P, P, P, P]
    lib.nrh_dw_workspace_floats.argtypes = [P, c_int]
    lib.nrh_dw_workspace_floats.restype = c_longlong
    lib.nrh_dw_gemm.argtypes = [P, c_int, c_longlong, P, c_longlong, P]
    lib.nrh_embedding_rows.argtypes = [P, P, P, c_int, c_int, c_longlong, P, P]
    lib.nrh_composite_loss.argtypes = [P, P, P, P, P, P, c_longlong, P, P, P, P, P]
    lib.nrh_loss_finish.argtypes = [P, c_longlong, c_float, P, c_float, P, P]
    lib.nrh_alpha_train_backward_fused.argtypes = [P, P, P, P, c_float, c_float, P, c_longlong, P, P, c_int, P, P, P, P, P, P, c_int, P]
    lib.nrh_variance_grad.argtypes = [P, c_longlong, c_float, P, P, P]
    lib.nrh_step_scalars.argtypes = [POINTER(c_void_p), POINTER(c_float), c_int, P, P, P]
    lib.nrh_adam_step.argtypes = [P, c_int, P, c_int, c_int, POINTER(ctypes.c_double), POINTER(c_void_p), POINTER(ctypes.c_double),
                                  POINTER(ctypes.c_double), POINTER(ctypes.c_double), P]
    lib.nrh_fuse_feature_head.argtypes = [P, c_int, P, P, P, P, P]
    lib.nrh_pack_gather.argtypes = [P, P, P, c_longlong, c_int, P, P]
    lib.nrh_sdf32_tables.argtypes = [POINTER(c_void_p), POINTER(c_int), P, P, P, P, P]
    lib.nrh_kernel_timing_select.argtypes = [c_int]
    lib.nrh_sampler_fusion.argtypes = [c_int]
    lib.nrh_sampler_fusion.restype = c_int
    lib.nrh_kernel_timing_read.argtypes = [POINTER(ctypes.c_double), POINTER(c_longlong)]
    for name in EXPORTED:
        getattr(lib, name)  # AttributeError if the build is stale
    _lib = lib
    return lib


def step_scalars(pairs=(), variance=None, inv_s_out=None) -> None:
    """nrh_step_scalars: ``pairs`` = up to four (0-dim or 1-element float32 CUDA tensor, python float) written in ONE launch, plus
    - with ``variance`` / ``inv_s_out`` (float32 CUDA tensors) - inv_s_out[0] = clip(exp(10 variance), 1e-6, 1e6)."""
    pairs = list(pairs)
    n = len(pairs)
    dst = (c_void_p * 4)(*([t.data_ptr() for t, _ in pairs] + [None] * (4 - n)))
    val = (c_float * 4)(*([float(v) for _, v in pairs] + [0.0] * (4 - n)))
    check(load().nrh_step_scalars(dst, val, n, ptr(variance), ptr(inv_s_out), stream_handle()), "nrh_step_scalars")


def check(rc: int, what: str) -> None:
    if rc != NRH_OK:
        msg = load().nrh_last_error_string().decode(errors="replace")
        raise NrhError(f"{what} failed: {_ERRNAMES.get(rc, rc)}: {msg}")


def param_sizes():
    out = (c_int * 8)()
    check(load().nrh_param_sizes(out), "nrh_param_sizes")
    return list(out)


# precision name -> how the parameters are PACKED (NrhNet.precision of every call but the evaluation render of "f16": see make_net)
PRECISIONS = {"f32": 0, "f16x3": 1, "f16": 1}


def ptr(t, dtype=None):
    """Device pointer of a contiguous float32 (or ``dtype``) CUDA(HIP) tensor, or None."""
    if t is None:
        return None
    import torch
    dtype = torch.float32 if dtype is None else dtype
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise ValueError(f"expected a contiguous {dtype} tensor on the GPU, got {type(t)} "
                         f"{getattr(t, 'dtype', None)} {getattr(t, 'device', None)} contiguous={getattr(t, 'is_contiguous', lambda: None)()}")
    return c_void_p(t.data_ptr())


def on_tensor_device(fn):
    """Decorator: run ``fn`` with the device of its first GPU tensor argument current, so that the C entry points (which
    launch on the current stream of the CURRENT device and cache their per-device set-up by its id) see the tensors' GPU."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        import torch
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, torch.Tensor) and a.is_cuda:
                with torch.cuda.device(a.device):
                    return fn(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapped


def stream_handle(device=None):
    """Current HIP stream of ``device`` (default: the current device).  The library caches per-device state by the CURRENT
    device id, so callers working on another GPU wrap their calls in ``torch.cuda.device(tensor.device)``."""
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_net(pk, hints, normal_type, depth_type, dyn_scalars=None, wide=True, fused=False, wide_color=True, shadow_jvp=False,
             shadow_clip=-1, samples=128, bg_alpha=None, tail_t=None, sampled_color=None, consts=None, counts=None, single_pass=False):
    """NrhNet from a renderer's packed-parameter dict (nrhints_amd/renderer.py: packed_params).  ``fused``: use the wide streams
    whose feature head is multiplied into the reflectance net's first layer (evaluation renders), if the dict has them;
    ``wide_color``: with them, also the reflectance net's block stream for the wide kernel (col_w32 / col_tab32).
    ``counts``: None (the reference's default sample counts) or (n_coarse, n_steps, n_new, s_coarse, s_new, lin_tables [4,128] on
    the device), see NrhNet in include/nrhints_hip.h.  ``single_pass``: NrhNet.precision 2 - the evaluation render of precision
    "f16": the f16x3 buffers with the wide SDF kernels in their one-term builds."""
    fused = bool(fused and wide and pk.get("sdf_w32f") is not None)
    w32 = (pk.get("sdf_w32f") if fused else pk.get("sdf_w32")) if wide else None
    tab = pk.get("sdf_tab32f") if fused else pk.get("sdf_tab32")
    c32 = pk.get("col_w32") if (fused and wide_color) else None
    # ``consts``: (specular_roughness [4], shadow_ray_offset) when they differ from the reference's defaults
    rough, offs = ((ctypes.c_double * 4)(*[float(x) for x in consts[0]]), float(consts[1])) if consts is not None \
        else ((ctypes.c_double * 4)(), 0.0)
    return NrhNet(ptr(pk["sdf_w"], pk["sdf_w"].dtype), ptr(pk["sdf_b"]), ptr(pk["sdf_head"]),
                  ptr(pk["col_w"], pk["col_w"].dtype), ptr(pk["col_b"]), pk["inv_s"], (2 if (single_pass and w32 is not None) else pk["precision"]),
                  hints, normal_type, depth_type, ptr(dyn_scalars),
                  ptr(w32, w32.dtype) if w32 is not None else None, ptr(tab) if w32 is not None else None, int(fused),
                  ptr(c32, c32.dtype) if c32 is not None else None, ptr(pk.get("col_tab32")) if c32 is not None else None,
                  int(bool(shadow_jvp and w32 is not None)), int(shadow_clip), int(samples), ptr(bg_alpha), ptr(tail_t), ptr(sampled_color),
                  int(consts is not None), rough, offs,
                  *((0, 0, 0, 0, 0, None) if counts is None else (int(counts[0]), int(counts[1]), int(counts[2]), int(counts[3]), int(counts[4]),
                                                                   ptr(counts[5]))))
