"""The training step without autograd: forward, loss and the whole backward as a fixed sequence of HIP launches.

``forward() + loss.backward()`` through the autograd Functions of autograd_core.py stays the drop-in path (it is what the
reference's trainer calls, pipelines/base_pipeline.py:41-69, trainer/trainer.py:269-283).  This module is the same computation
with nothing left to the framework between the kernels: no autograd graph, no library GEMM, no elementwise torch op on the
per-sample arrays.  The loss is part of it (its adjoint seeds are known without any downstream information), so one call
returns the loss dict AND leaves ``.grad`` on all 46 parameter tensors:

    fold weight-norm (1 launch) -> re-pack -> nrh_render_forward_train (samplers, SDF training forward, alpha, shadow march,
    per-ray encodings) -> reflectance forward -> composite + loss + seeds (2) -> reflectance adjoint -> alpha adjoint (+ eikonal
    seed) -> SDF tangent + value sweeps -> positional encoding rows -> ALL weight gradients in one split-K bf16x3 launch
    (csrc/nrh_dw.hip) + its reduction -> d variance -> weight-norm adjoint (1)

Ray gradients (pose / light refinement: the reference's default preset nr-hints-cam-opt, and register_view): when the ray tensors
require grad, one more launch (nrh_ray_adjoint) reduces the sweeps' adjoints to d loss / d (origins, directions, pl_positions),
which are handed to the ray generator's backward (nrh_generate_rays_indexed_backward through its autograd.Function).

Covered since round 5 (each differs from the default model only in what the alpha / colour stages are fed): ``normal_type``
Analytic (the reflectance net reads the raw gradient, its adjoint goes straight into the gradient's), one hint without the other
(the kernels of the full model on a first reflectance layer whose missing columns are zero; their gradient columns are dropped)
and the partial visibility hint (``n_shadow_importance_clip``: one row of per-ray inputs per group of samples).
Fewer than 128 samples per ray (``n_importance_samples = 0``, sample counts off the defaults): the per-sample arrays keep 128
slots, the padded ones have weight and adjoint exactly 0.
Round 6: ``shadow_hint_gradient`` / ``specular_hint_gradient`` (models/neus_hint_model.py:379, :589) - the SDF training forward and
both sweeps a second time at the shadow ray's sections, the shadow alpha stage's kernel pair, and a small autograd island on
per-RAY tensors for the hit normal, the cue and the two encodings (_hint_forward / _hint_backward below).
Networks narrower than the compiled shape (renderer._narrow) run zero-padded; the adjoint of the padding cuts the gradients back.
Fourth session of round 6: ``use_outside_nerf`` (:434-473, :516-519, :630-637) - the background network stays behind its autograd
Function (its kernels, csrc/nrh_outside.hip), the 160-sample composite and the loss are a torch island between the kernels
(_outside_forward / _outside_loss / _outside_backward below).
Restrictions (the autograd path covers the rest): GPU float32 parameters, at most ``max_fused_train_rays`` rays per call; shadow_hint_gradient with ray gradients or with the partial visibility hint is refused on
both paths.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

from . import _lib, dw, ops, packing

LOSS_KEYS = ("loss", "rgb_loss", "eikonal_loss", "s_val", "psnr")


# 16-bit hand-offs of the reflectance net's adjoints (nrh_color_train_backward_half): stored as adjoint_scale(rays) x this x the value.
# |zbar4| <= 1 / (12 rays) whatever the scene, and the measured maxima of the layers' adjoints x rays span 2^-12.6 .. 2^-5.3
# (profiles/r05/dw16_ranges.log): with rays / 8 x 1 024 they land at 2^-5.6 .. 2^1.7 - 13 binades under fp16's largest number,
# 7 above the level (2^-13) where its absolute floor would cost accuracy relative to an array's largest entries.
COLOR_HALF_GAIN = 1024.0


def supported(renderer, ray_bundle) -> Optional[str]:
    """None if the fused step applies, else the reason it does not."""
    rc = renderer.config.renderer
    if rc.shadow_hint_gradient and renderer.has_shadow_hint:
        if int(getattr(renderer, "_shadow_clip", -1)) > 0:
            return "shadow_hint_gradient with the partial visibility hint (not implemented on either path)"
        if any(torch.is_tensor(t) and t.requires_grad for t in (ray_bundle.origins, ray_bundle.directions, ray_bundle.pl_positions)):
            return "shadow_hint_gradient together with ray gradients (not implemented on either path)"
    if ray_bundle.origins.shape[0] > renderer.max_fused_train_rays:
        return "more rays than max_fused_train_rays"
    if ray_bundle.origins.shape[0] == 0:
        return "empty batch"
    ps = list(renderer.parameters())
    if not all(p.is_cuda and p.dtype == torch.float32 for p in ps):
        return "parameters must be float32 on the GPU"
    if not (all(p.requires_grad for p in ps) or not any(p.requires_grad for p in ps)):
        return "partly frozen parameters"
    return None


class _Buffers:
    """Per-(device, rays) arrays of a step, allocated once and reused (a captured graph bakes their addresses in)."""

    def __init__(self, dev, n: int, hints: bool, shapes: Dict[str, tuple], param_layout: Dict[str, tuple], clip: int = 1, narrow: bool = False):
        T, P = 128, n * 128
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.n, self.dev = n, dev
        mw = 128 if hints else 64
        # per-ray inputs of the reflectance net: one row per ray, or per group of 128 / clip samples (partial visibility hint)
        self.raymisc = torch.zeros(n * max(1, clip), packing.RAYMISC_STRIDE, dtype=torch.float32, device=dev)
        self.pts = new(P, 3)
        self.color, self.save_h, self.save_misc = new(P, 3), new(4, P, 256), new(P, mw)
        self.rgb, self.zbar4, self.wbar, self.partial, self.loss8 = new(n, 3), new(P, 3), new(n, T), new(n, 4), new(8)
        self.czbar, self.fbar, self.mbar = new(4, P, 256), new(P, 256), new(P, mw)
        self.sdf_bar, self.grad_bar, self.rd_bar, self.invs_bar = new(P), new(P, 3), new(n, 3), new(n)
        self.emb = new(P, 64)
        self.save_h16 = self.czbar16 = None      # the reflectance net's 16-bit hand-offs (allocated when the step uses them)
        self.g_shadow = self.emb_shadow = None   # shadow_hint_gradient: the SDF net's weight gradients through the visibility
        self.dyn = torch.zeros(4, dtype=torch.float32, device=dev)      # {S, 1 / S, work words} of the 16-bit hand-offs (ops.sdf_train_backward)
        self.o_bar, self.d_bar, self.pl_bar = new(n, 3), new(n, 3), new(n, 3)      # ray adjoints (pose / light refinement)
        self.far_bar = None          # ... and d loss / d far [n,1] when the outside NeRF's far bound carries a graph (train_step_backward)
        # What .grad of the 46 parameter tensors points at: views into ONE flat float32 buffer, laid out in the order of
        # renderer.parameters() (``param_layout``: name -> (offset, shape)).  The kernels that produce parameter gradients write
        # straight into it - the weight-norm adjoint (weight_v / weight_g), the weight-gradient reduction's column sums (biases),
        # nrh_variance_grad - so a data-parallel step all-reduces this buffer IN PLACE (training.FlatGradAllReduce finds it through
        # the views' ``_nrh_flat`` tag: no pack / unpack copies) and an optimiser working from a pointer table (adam.HipAdam) never
        # sees a new address.
        self.flat = torch.zeros(sum(int(torch.Size(shp).numel()) for _, (_, shp) in param_layout.items()), dtype=torch.float32, device=dev)
        self.pgrad = {}
        for name, (off, shp) in param_layout.items():
            v = self.flat[off: off + int(torch.Size(shp).numel())].view(shp)
            v._nrh_flat = (self.flat, off)
            self.pgrad[name] = v
        g = {}
        # a network narrower than the compiled shape (renderer._narrow): the kernels write bias sums of the COMPILED length, so they
        # go to buffers of their own and reach .grad through the adjoint of the zero padding (train_step_backward)
        bias = (lambda name, key: new(*shapes[key])) if narrow else (lambda name, key: self.pgrad[name].view(-1))
        for l in range(8):     # dense (folded) weight gradients are intermediates; the bias gradients are final
            g[f"dW{l}"], g[f"db{l}"] = new(*shapes[f"sdf_w{l}"]), bias(f"sdf_network.lin{l}.bias", f"sdf_b{l}")
        g["ws"], g["bs"] = new(1, 256), bias("sdf_network.out_sdf.bias", "sdf_head_b")
        g["Wf"], g["bf"] = new(256, 256), bias("sdf_network.out_feat.bias", "feat_b")
        for l in range(5):
            g[f"w{l}"], g[f"b{l}"] = new(*shapes[f"col_w{l}"]), bias(f"color_network.lin{l}.bias", f"col_b{l}")
        # one-hint models: the weight-gradient kernel fills the full model's [256, 361] first layer; ``w0_real`` receives the
        # columns that exist in the parameter (``shapes["col_w0_real"]``)
        self.w0_real = new(*shapes["col_w0_real"]) if (tuple(shapes["col_w0_real"]) != tuple(shapes["col_w0"]) and not narrow) else None
        self.g = g
        self.vbars = {"v:" + k: self.pgrad[k + ".weight_v"] for k in packing._FOLD_LAYERS}
        self.gbars = {"g:" + k: self.pgrad[k + ".weight_g"] for k in packing._FOLD_LAYERS}
        self.var_bar = self.pgrad["deviation_network.variance"].view(1)


def train_step_backward(renderer, ray_bundle, rgb_gt: torch.Tensor, background_rgb: Optional[torch.Tensor], global_step: int,
                        igr_weight: Optional[float] = None, t_rand_primary=None, t_rand_shadow=None, is_training: bool = True,
                        ray_grads: Optional[Dict[str, torch.Tensor]] = None, forward_out: Optional[dict] = None,
                        t_rand_outside=None) -> torch.Tensor:
    """Forward + loss + backward of one batch.  Returns the loss vector [8] on the device (``LOSS_KEYS`` = entries 0..4) and sets
    ``.grad`` of every renderer parameter (overwriting, like zero_grad + backward) - unless the renderer's parameters are frozen
    (``requires_grad_(False)``, as in register_view), in which case the weight gradients, the weight-norm adjoint and the variance
    gradient are skipped.

    ``is_training=False``: the evaluation-mode forward of ``register_view`` (pipelines/base_pipeline.py:80-85): no jitter, cosine
    ratio 1, no geometry warm-up; pass ``igr_weight=0`` for its plain L1 loss.
    Ray gradients: if any of the bundle's origins / directions / pl_positions requires grad, their adjoints are computed
    (nrh_ray_adjoint) and - ``ray_grads`` None - pushed into the autograd graph that produced the bundle (the ray generator's
    backward), accumulating ``.grad`` on its parameters like ``loss.backward()``; with a dict ``ray_grads`` they are returned in it
    (keys origins / directions / pl_positions; persistent buffers) and nothing is propagated.
    ``forward_out`` (tests / diagnostics): a dict that receives THIS step's forward products - mid_z, dists [n,128], visibilities
    [n,1], cue [n,128,4], weights, inside, and the SDF network's outputs at the samples (sdf [P,1], normals [n,128,3], feat
    [P,256]) - i.e. the sample placement and linearisation point of the gradients this call leaves in ``.grad``.  (A separate
    _render_train call is NOT guaranteed to place the same samples: this step folds weight-norm with nrh_weight_norm_fold, other
    paths with torch ops, and a last-bit difference in a weight moves importance samples where the pdf sits at its floor.)
    ``t_rand_outside`` [n,32]: the stratified jitter of the samples beyond the sphere (renderer.use_outside_nerf, :689); drawn when None."""
    why = supported(renderer, ray_bundle)
    if why is not None:
        raise ValueError(f"fused training step not applicable: {why}")
    lib = _lib.load()
    P_ = _lib.ptr
    cfg = renderer.config
    igr = float(cfg.igr_weight if igr_weight is None else igr_weight)
    f32 = lambda t: t.detach().to(dtype=torch.float32).contiguous()
    o, d, pl = f32(ray_bundle.origins), f32(ray_bundle.directions), f32(ray_bundle.pl_positions)
    near, far = f32(ray_bundle.nears).reshape(-1), f32(ray_bundle.fars).reshape(-1)
    dev, n = o.device, o.shape[0]
    with torch.cuda.device(dev), torch.no_grad():
        stream = _lib.stream_handle()
        cos_anneal = (min(1.0, global_step / cfg.anneal_end) if cfg.anneal_end > 0 else 1.0) if is_training else 1.0
        zero_hints = 1 if (is_training and global_step < cfg.geometry_warmup_end) else 0
        t_p = t_s = None
        if is_training:
            want_s = not zero_hints and renderer._hints
            srows = n * max(1, int(getattr(renderer, "_shadow_clip", -1)))      # one shadow ray per ray, or per group of samples (:560-568)
            sc = int(getattr(renderer, "_shadow_coarse", 64))                    # renderer.n_shadow_samples: the jitter's row length (:394)
            if t_rand_primary is None and t_rand_shadow is None and want_s:
                n64 = (n + 63) // 64 * 64                        # (the shadow block stays 256-byte aligned)
                r = torch.rand(n64 + srows * sc, device=dev)    # one generator launch: primary jitter [n], then shadow jitter [rows, sc]
                t_p, t_s = r[:n], r[n64:].view(srows, sc)
            else:
                t_p = f32(t_rand_primary).reshape(-1) if t_rand_primary is not None else torch.rand(n, device=dev)
                if want_s:
                    t_s = f32(t_rand_shadow) if t_rand_shadow is not None else torch.rand(srows, sc, device=dev)
                    if tuple(t_s.shape) != (srows, sc):
                        raise ValueError(f"t_rand_shadow must be [{srows}, {sc}]")
        want_rays = any(torch.is_tensor(t) and t.requires_grad for t in (ray_bundle.origins, ray_bundle.directions, ray_bundle.pl_positions))
        want_params = any(p.requires_grad for p in renderer.parameters())
        # ---- parameters: fold weight-norm (one launch), re-pack ----
        named = dict(renderer.named_parameters())
        gs = [named[k + ".weight_g"].detach() for k in packing._FOLD_LAYERS]
        vs = [named[k + ".weight_v"].detach() for k in packing._FOLD_LAYERS]
        ws = [torch.empty_like(v) for v in vs]
        packing.WeightNormFoldHip._call("nrh_weight_norm_fold", vs, gs, ws)
        dense = {}
        for (wk, bk), k, w in zip(packing._FOLD_KEYS, packing._FOLD_LAYERS, ws):
            dense[wk], dense[bk] = w, named[k + ".bias"].detach()
        w0_real_shape = tuple(dense["col_w0"].shape)
        narrow, real, padded = bool(getattr(renderer, "_narrow", False)), None, None
        if narrow and want_params:
            # a network narrower than the compiled shape runs zero-padded (packing.pad_to_compiled); the padding is recorded as a
            # tiny autograd graph on the folded matrices, and its adjoint below cuts the kernels' compiled-shape gradients back
            with torch.enable_grad():
                real = {k: v.detach().requires_grad_(True) for k, v in dense.items()}
                padded = renderer._to_compiled(real)
            dense = {k: v.detach() for k, v in padded.items()}
        else:
            dense = renderer._to_compiled(dense)             # one-hint models: zero columns for the hint that is not there
        pk = renderer.packed_params(dev, dense=dense)
        hints = bool(renderer._hints)
        clip = max(1, int(getattr(renderer, "_shadow_clip", -1)))
        analytic = bool(renderer._normal_type)
        key = (str(dev), n, hints)
        cache = renderer.__dict__.setdefault("_fused_buffers", {})     # per renderer: .grad aliases these arrays
        B = cache.get(key)
        if B is None:
            # one buffer set is ~1.3 GB per 1 024 rays: a loop with varying batch sizes (a last partial batch, a curriculum) must not
            # accumulate one set per distinct n.  Only the most recent set stays, plus whatever a live captured step
            # pinned (GraphedTrainStep bakes the addresses into its hipGraph; it unpins in release()).
            pinned = renderer.__dict__.setdefault("_fused_pinned", set())
            for k in [k for k in cache if k not in pinned]:
                del cache[k]
            shapes = {k: tuple(v.shape) for k, v in dense.items()}
            shapes["col_w0_real"] = w0_real_shape
            shapes.update({"v:" + k: tuple(v.shape) for k, v in zip(packing._FOLD_LAYERS, vs)})
            shapes.update({"g:" + k: tuple(x.shape) for k, x in zip(packing._FOLD_LAYERS, gs)})
            layout, off = {}, 0
            for pname, prm in renderer.named_parameters():
                layout[pname] = (off, tuple(prm.shape))
                off += prm.numel()
            B = cache[key] = _Buffers(dev, n, hints, shapes, layout, clip, narrow)
        # ---- no-grad stages + SDF training forward (one C call) ----
        Pn = n * 128
        # 16-bit hand-offs of the SDF net's weight-gradient operands (f16x3, batches the 8-wave kernels run): the renderer's option
        # ``dw_half`` (NeuSHintRenderer.dw_half; False = float32 hand-offs, the precision-matched form).  Not an environment switch:
        # it changes numerics (11-bit operands of the weight-gradient products).
        half = (pk["precision"] == 1 and bool(getattr(renderer, "dw_half", False)) and bool(lib.nrh_train_half_supported(1, Pn)))
        # (the alpha stage also leaves p = o + d * t - separate roundings, as the SDF kernels form it - in B.pts for the reflectance net)
        rcfg = cfg.renderer
        shadow_grad = bool(rcfg.shadow_hint_gradient and renderer.has_shadow_hint and not zero_hints)
        specular_grad = bool(rcfg.specular_hint_gradient and renderer.has_specular_hint and not zero_hints)
        og = _outside_forward(renderer, lib, pk, o, d, pl, near, far, t_p, t_rand_outside, is_training, want_rays,
                              ray_bundle.fars if (want_rays and torch.is_tensor(ray_bundle.fars) and ray_bundle.fars.requires_grad) else None) \
            if renderer.has_outside_nerf else None
        res = renderer._render_train(o, d, pl, near, far, cos_anneal, t_p, t_s, zero_hints, raymisc=B.raymisc, half_handoffs=half, pts=B.pts,
                                     want_shadow=shadow_grad, extra_net=None if og is None else og["extra"])
        pre, sv = res["pre"], res["pre"]["saves"]
        hg = _hint_forward(renderer, lib, B, pk, res, o, d, pl, cos_anneal, shadow_grad, specular_grad, want_rays) \
            if (shadow_grad or specular_grad) else None
        if forward_out is not None:
            forward_out.update({k: res[k] for k in ("mid_z", "dists", "visibilities", "cue", "weights", "inside", "normals", "depth")})
            forward_out.update(sdf=pre["sdf"], feat=pre["feat"], vis_groups=res.get("vis_groups"))
        # ---- reflectance forward ----
        cw = pk["col_w"]
        # the normal the reflectance net reads: normalised (default) or the raw gradient (normal_type Analytic, :622-623)
        normal_in = (res["normals"] if analytic else res["nhat"]).view(Pn, 3)
        # rows of per-ray inputs: one per ray, or - partial visibility hint outside the geometry warm-up - one per group of samples
        per_row = 128 // clip if (clip > 1 and not zero_hints) else 128
        # the reflectance net's 16-bit hand-offs: with the SDF net's (its kernels have one build and would take them at any size, but
        # at 64 rays - where the SDF net runs on the channel-split kernels - the step was 7 % SLOWER with them: 1.04 against 0.97 ms,
        # profiles/r05/train_half_small_ab.log; the weight-gradient launch is latency-bound there and the half path's deeper pipeline
        # has the longer prologue)
        half_c = half
        if half_c:
            if B.save_h16 is None:
                B.save_h16 = torch.empty(4, Pn, 256, dtype=torch.float16, device=dev)
                B.czbar16 = torch.empty(4, Pn, 256, dtype=torch.float16, device=dev)
            _lib.check(lib.nrh_color_train_forward_half(1, int(hints), P_(cw, cw.dtype), P_(pk["col_b"]), P_(pre["feat"]), P_(B.pts),
                                                        P_(normal_in), P_(B.raymisc), per_row, n, P_(B.color), P_(B.save_h),
                                                        P_(B.save_misc), P_(B.save_h16, torch.float16), stream), "nrh_color_train_forward_half")
        else:
            _lib.check(lib.nrh_color_train_forward_grouped(pk["precision"], int(hints), P_(cw, cw.dtype), P_(pk["col_b"]), P_(pre["feat"]), P_(B.pts),
                                                           P_(normal_in), P_(B.raymisc), per_row, n, P_(B.color), P_(B.save_h),
                                                           P_(B.save_misc), stream), "nrh_color_train_forward")
        # ---- composite, loss, adjoint seeds ----
        bg = f32(background_rgb.to(dev)).reshape(-1) if background_rgb is not None else None
        gt = f32(rgb_gt)
        dyn = renderer.dyn_scalars
        inv_s = pk["inv_s"]
        if og is None:
            _lib.check(lib.nrh_composite_loss(P_(B.color), P_(res["weights"]), P_(gt), P_(bg), P_(res["normals"].view(Pn, 3)), P_(res["inside"]),
                                              n, P_(B.rgb), P_(B.zbar4), P_(B.wbar), P_(B.partial), stream), "nrh_composite_loss")
            _lib.check(lib.nrh_loss_finish(P_(B.partial), n, float(inv_s), P_(dyn), igr, P_(B.loss8), stream), "nrh_loss_finish")
        else:
            _outside_loss(og, B, res, gt, None if bg is None else bg.view(1, 3), igr, n, float(inv_s), dyn)
        # ---- reflectance adjoint sweep ----
        cwt = pk["col_wt"]
        if half_c:
            _lib.check(lib.nrh_color_train_backward_half(1, int(hints), P_(cwt, cwt.dtype), P_(B.zbar4), P_(B.save_h), n, P_(B.czbar), P_(B.fbar),
                                                         P_(B.mbar), _lib.adjoint_scale(n), P_(B.save_h16, torch.float16),
                                                         P_(B.czbar16, torch.float16), COLOR_HALF_GAIN, stream), "nrh_color_train_backward_half")
        else:
            _lib.check(lib.nrh_color_train_backward(pk["precision"], int(hints), P_(cwt, cwt.dtype), P_(B.zbar4), P_(B.save_h), n, P_(B.czbar),
                                                    P_(B.fbar), P_(B.mbar), _lib.adjoint_scale(n), stream), "nrh_color_train_backward")
        # ---- alpha stage adjoint (+ the eikonal seed; the unit normal's adjoint are columns 3..5 of mbar) ----
        mw = B.mbar.shape[1]
        # NormalizedAnalytic: the unit normal's adjoint goes through the normalisation inside the kernel; Analytic: the reflectance
        # net read the gradient itself, its adjoint is added to the gradient's below
        nbar, nbar_stride = (None if analytic else ctypes.c_void_p(B.mbar.data_ptr() + 12)), mw
        if hg is not None:
            # hint gradients (:379, :589): the adjoints of enc(visibility) / enc(cue) - per-ray sums of their mbar columns - go back
            # through the hints: the cue's into the hit normal (unit normals and weights of this ray's samples: added to what the alpha
            # adjoint reads), the visibility's into the shadow ray's alpha stage and from there into the SDF net at the shadow points
            _hint_backward(hg, B, n, mw, analytic)
            if hg.get("nhat_bar") is not None and analytic:      # Analytic: the net's normal input is the raw gradient, but the hit
                nbar, nbar_stride = P_(hg["nhat_bar"]), 3        # normal is built from UNIT normals (:584): their adjoint on its own
        if og is None:
            _lib.check(lib.nrh_alpha_train_backward_fused(P_(pre["sdf"]), P_(res["normals"].view(Pn, 3)), P_(d), P_(res["dists"]), float(inv_s),
                                                          float(cos_anneal), P_(dyn), n, P_(B.wbar), nbar, nbar_stride, P_(res["inside"]),
                                                          ctypes.c_void_p(B.loss8.data_ptr() + 20), P_(B.sdf_bar), P_(B.grad_bar), P_(B.rd_bar),
                                                          P_(B.invs_bar), int(renderer._samples), stream), "nrh_alpha_train_backward_fused")
        else:
            _outside_backward(renderer, og, lib, B, pre, res, d, float(inv_s), float(cos_anneal), dyn, n, analytic, stream,
                              nhat_bar=hg.get("nhat_bar") if (hg is not None and analytic) else None)
        if analytic:
            B.grad_bar.add_(B.mbar[:, 3:6])
        shadow_r = None
        if hg is not None and hg.get("vis_bar") is not None:
            shadow_r = _shadow_backward(hg, lib, B, pk, n, float(inv_s), float(cos_anneal), dyn, int(renderer._shadow_total), stream)
        if want_params:
            _lib.check(lib.nrh_variance_grad(P_(B.invs_bar), n, float(inv_s), P_(dyn), P_(B.var_bar), stream), "nrh_variance_grad")
        # ---- SDF network: tangent + value sweeps ----
        r = ops.sdf_train_backward(pk["sdf_w"], pk["sdf_wt_feat"], pk["sdf_head"], o, d, res["mid_z"], 128, sv, B.sdf_bar, B.fbar, B.grad_bar,
                                   adj_scale=_lib.adjoint_scale(n), half_handoffs=half, dyn=B.dyn)
        # ---- ray adjoints (pose / light refinement): one per-ray reduction of what the sweeps left ----
        if want_rays:
            _lib.check(lib.nrh_ray_adjoint(P_(o), P_(d), P_(pl), P_(res["mid_z"]), P_(r["pbar"]), P_(B.grad_bar), P_(sv["ge"]), P_(B.mbar), mw,
                                           P_(B.rd_bar), n, P_(B.o_bar), P_(B.d_bar), P_(B.pl_bar), stream), "nrh_ray_adjoint")
            if hg is not None and hg.get("d_bar") is not None:       # the cue's own dependence on the view direction and the light (:590-615)
                B.d_bar.add_(hg["d_bar"])
                B.pl_bar.add_(hg["pl_bar"])
            if og is not None and og.get("leaves") is not None:      # the background network's dependence on the rays (_outside_backward)
                lv = og["leaves"]
                B.o_bar.add_(lv["o"].grad); B.d_bar.add_(lv["d"].grad); B.pl_bar.add_(lv["pl"].grad)
                B.far_bar = lv["far"].grad if lv["far"] is not None else None
        if not want_params:
            # frozen renderer (register_view, pipelines/base_pipeline.py:71-91: only the ray generator's deltas step): the reference
            # computes and discards all 46 parameter gradients there; the results are the same without them
            return _finish_rays(B, ray_bundle, want_rays, ray_grads)
        _lib.check(lib.nrh_embedding_rows(P_(o), P_(d), P_(res["mid_z"]), 128, 128, n, P_(B.emb), stream), "nrh_embedding_rows")
        # ---- every weight gradient: one split-K launch + its reduction ----
        shapes = [tuple(dense[f"sdf_w{l}"].shape) for l in range(8)]
        h16 = dict(h16=sv["h16"], t16=sv["t16"], zbar16=r["zbar16"], abar16=r["abar16"], dyn=B.dyn) if half else None
        jobs = dw.sdf_jobs(shapes, sv["h"], sv["t"], r["zbar"], r["abar"], r["gebar"], B.emb, B.sdf_bar, B.fbar, B.g, half=h16) + \
            dw.color_jobs(hints, B.czbar, B.zbar4, B.save_h, pre["feat"], B.save_misc, B.g,
                          half=dict(zbar16=B.czbar16, h16=B.save_h16, inv_scale=1.0 / (_lib.adjoint_scale(n) * COLOR_HALF_GAIN)) if half_c else None)
        if shadow_r is not None:
            # the SDF net's weight gradients through the visibility: the same job table over the shadow points' arrays, into a second
            # set of outputs that is added below (a job's output is written, not accumulated)
            if B.g_shadow is None:
                B.g_shadow = {k: torch.empty_like(B.g[k]) for k in [f"dW{l}" for l in range(8)] + [f"db{l}" for l in range(8)] + ["ws", "bs", "Wf", "bf"]}
                B.emb_shadow = torch.empty_like(B.emb)
            _lib.check(lib.nrh_embedding_rows(P_(pl), P_(hg["srd"]), P_(res["shadow_mid_z"]), 128, 128, n, P_(B.emb_shadow), stream), "nrh_embedding_rows")
            # (its own nrh_dw_gemm call: the two tables together exceed the kernel's 24 jobs)
            dw.run(dw.sdf_jobs(shapes, hg["saves"]["h"], hg["saves"]["t"], shadow_r["zbar"], shadow_r["abar"], shadow_r["gebar"],
                               B.emb_shadow, hg["sdf_bar"], hg["fbar0"], B.g_shadow), Pn)
        dw.run(jobs, Pn)
        if shadow_r is not None:
            keys = list(B.g_shadow)
            torch._foreach_add_([B.g[k] for k in keys], [B.g_shadow[k] for k in keys])
        # ---- weight-norm adjoint -> .grad ----
        g = B.g
        if narrow:
            # the adjoint of the zero padding: compiled-shape gradients -> the folded matrices' own shapes (rows / columns of padded
            # channels and encoding columns are dropped; their entries are whatever the padded channels' constants produced)
            wk = [f"sdf_w{l}" for l in range(8)] + ["sdf_head_w", "feat_w"] + [f"col_w{l}" for l in range(5)]
            bk = [f"sdf_b{l}" for l in range(8)] + ["sdf_head_b", "feat_b"] + [f"col_b{l}" for l in range(5)]
            gw = [g[f"dW{l}"] for l in range(8)] + [g["ws"], g["Wf"]] + [g[f"w{l}"] for l in range(5)]
            gb = [g[f"db{l}"] for l in range(8)] + [g["bs"], g["bf"]] + [g[f"b{l}"] for l in range(5)]
            with torch.enable_grad():
                cut = torch.autograd.grad([padded[k] for k in wk + bk], [real[k] for k in wk + bk],
                                          [t.view(padded[k].shape) for t, k in zip(gw + gb, wk + bk)])
            wbars = [t.contiguous() for t in cut[:len(wk)]]
            for name, t in zip(packing._FOLD_LAYERS, cut[len(wk):]):
                B.pgrad[name + ".bias"].copy_(t)
            vbars = [B.vbars["v:" + k] for k in packing._FOLD_LAYERS]
            gbars = [B.gbars["g:" + k] for k in packing._FOLD_LAYERS]
            packing.WeightNormFoldHip._call("nrh_weight_norm_fold_backward", vs, gs, wbars, vbars, gbars)
            for pname, prm in named.items():
                prm.grad = B.pgrad[pname]
            return _finish_rays(B, ray_bundle, want_rays, ray_grads)
        w0bar = g["w0"]
        if B.w0_real is not None:       # one-hint model: drop the gradient columns of the hint that does not exist (renderer._pad_hint_columns)
            if renderer.has_shadow_hint:
                B.w0_real.copy_(w0bar[:, :B.w0_real.shape[1]])
            else:
                B.w0_real[:, :316].copy_(w0bar[:, :316])
                B.w0_real[:, 316:].copy_(w0bar[:, 325:])
            w0bar = B.w0_real
        wbars = [g[f"dW{l}"] for l in range(8)] + [g["ws"], g["Wf"], w0bar] + [g[f"w{l}"] for l in range(1, 5)]
        bbars = [g[f"db{l}"] for l in range(8)] + [g["bs"], g["bf"]] + [g[f"b{l}"] for l in range(5)]
        vbars = [B.vbars["v:" + k] for k in packing._FOLD_LAYERS]
        gbars = [B.gbars["g:" + k] for k in packing._FOLD_LAYERS]
        packing.WeightNormFoldHip._call("nrh_weight_norm_fold_backward", vs, gs, wbars, vbars, gbars)
        # (the bias / variance gradients ARE the persistent buffers: like backward() after zero_grad(), every call overwrites them)
        for pname, prm in named.items():
            prm.grad = B.pgrad[pname]
        return _finish_rays(B, ray_bundle, want_rays, ray_grads)


def _outside_forward(renderer, lib, pk, o, d, pl, near, far, t_p, t_rand_outside, is_training: bool, want_rays: bool = False, far_g=None) -> dict:
    """renderer.use_outside_nerf on the fused step, part 1 (models/neus_hint_model.py:677-724, :434-473): where the primary ray's
    samples are (nrh_sample_primary), the 32 positions beyond the sphere, and the background network at the merged 160 positions
    (outside.render_outside: csrc/nrh_outside.hip behind its autograd Function, whose graph - background parameters -> alpha,
    colour - is kept for part 3).  -> the NrhNet extras of the render call (bg_alpha in, tail_t out) and the island's tensors."""
    from . import outside
    P_ = _lib.ptr
    dev, n = o.device, o.shape[0]
    rc = renderer.config.renderer
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    z, mid0, dists0 = new(n, 128), new(n, 128), new(n, 128)
    lin64, lin16 = renderer._const(dev)
    net0 = _lib.make_net(pk, renderer._hints, renderer._normal_type, renderer._depth_type, renderer.dyn_scalars, wide=renderer.wide_kernels,
                         consts=renderer._net_consts)
    ws = renderer._workspace(dev, n)
    _lib.check(lib.nrh_sample_primary(net0, P_(o), P_(d), P_(near), P_(far), n, P_(t_p), P_(lin64), P_(lin16), P_(z), P_(mid0), P_(dists0),
                                      P_(ws), ws.numel(), _lib.stream_handle()), "nrh_sample_primary")
    t_o = None
    if is_training:
        t_o = torch.rand(n, outside.N_OUTSIDE, device=dev) if t_rand_outside is None else t_rand_outside.detach().to(dev, torch.float32)
    renderer.outside_nerf.precision = renderer.precision
    # pose / light refinement: the background sees the rays through its sample points (o + d mid), its view / light inputs and - the
    # ray generator's far bound positions the 32 samples beyond the sphere, :689-693 - through ``far``; leaves of the island
    leaves = None
    if want_rays:
        leaves = dict(o=o.detach().clone().requires_grad_(True), d=d.detach().clone().requires_grad_(True), pl=pl.detach().clone().requires_grad_(True),
                      far=(far.detach().reshape(-1, 1).clone().requires_grad_(True) if far_g is not None else None))
    with torch.enable_grad():
        lo, ld_, lpl = (leaves["o"], leaves["d"], leaves["pl"]) if leaves else (o, d, pl)
        lfar = leaves["far"] if (leaves and leaves["far"] is not None) else far.reshape(-1, 1)
        z_out = outside.outside_z(lfar, rc.n_samples, t_o)
        z_feed, _ = torch.sort(torch.cat([z, z_out], dim=-1), dim=-1)
        bg_alpha, bg_col = outside.render_outside(renderer.outside_nerf, lo, ld_, lpl, z_feed, 2.0 / rc.n_samples)
    bg_a, tail_t = bg_alpha.detach().contiguous(), new(n)
    return dict(bg_alpha=bg_alpha, bg_col=bg_col, bg_a=bg_a, tail_t=tail_t, leaves=leaves,
                extra=dict(bg_alpha=bg_a, tail_t=tail_t, sampled_color=None))


def _outside_loss(og: dict, B, res, gt, bgc, igr: float, n: int, inv_s: float, dyn) -> None:
    """Part 2: composite over 128 + 32 samples and the loss (:520-523, :630-637; pipelines/base_pipeline.py:50-69) as a torch island on
    per-sample arrays the kernels produced - leaves: blended weights, the transmittance behind sample 127, the reflectance net's
    colours, the SDF gradients (eikonal term), and the background's alpha / colour as cut points.  Its backward seeds everything
    downstream: B.wbar, B.zbar4 (through the sigmoid, as nrh_composite_loss writes it), the eikonal seed, d loss / d tail_t and the
    background's own adjoints; B.loss8 gets the loss vector."""
    from . import outside
    with torch.enable_grad():
        w_leaf = res["weights"].detach().requires_grad_(True)
        t_leaf = og["tail_t"].detach().view(n, 1).requires_grad_(True)
        c_leaf = B.color.detach().view(n, 128, 3).requires_grad_(True)
        n_leaf = res["normals"].detach().requires_grad_(True)
        ba, bc = og["bg_alpha"].detach().requires_grad_(True), og["bg_col"].detach().requires_grad_(True)
        rgb = outside.composite(w_leaf, t_leaf, res["inside"], c_leaf, ba, bc, bgc)["rgb"]
        rgb_loss = (rgb - gt).abs().sum() / (n + 1e-5)
        mask = res["inside"]
        eik = (mask * (torch.linalg.norm(n_leaf, ord=2, dim=-1) - 1.0) ** 2).sum() / (mask.sum() + 1e-5)
        loss = rgb_loss + eik * igr
        gw, gt_, gc, gn, gba, gbc = torch.autograd.grad(loss, [w_leaf, t_leaf, c_leaf, n_leaf, ba, bc])
    B.wbar.copy_(gw)
    c = B.color.view(n, 128, 3)
    B.zbar4.view(n, 128, 3).copy_(gc * c * (1.0 - c))
    B.rgb.copy_(rgb.detach())
    og.update(tbar=gt_.reshape(n).contiguous(), gn=gn.reshape(-1, 3), gba=gba, gbc=gbc)
    s_val = (1.0 / dyn[0]).reshape(()) if dyn is not None else torch.full((), 1.0 / inv_s, dtype=torch.float32, device=rgb.device)
    psnr = 10.0 * torch.log10(1.0 / torch.mean((rgb.detach() - gt) ** 2))
    B.loss8[:5].copy_(torch.stack([loss.detach(), rgb_loss.detach(), eik.detach(), s_val, psnr]))


def _outside_backward(renderer, og: dict, lib, B, pre, res, d, inv_s: float, cos_anneal: float, dyn, n: int, analytic: bool, stream,
                      nhat_bar=None) -> None:
    """Part 3: the alpha stage's adjoint with the blend (nrh_alpha_blend_backward: also d loss / d bg_alpha[:, :128]), the eikonal
    seed, and the background network's backward - its parameter gradients accumulate into their views of the flat .grad buffer."""
    P_ = _lib.ptr
    Pn = n * 128
    # the unit normals' adjoint: the reflectance net's input columns (NormalizedAnalytic) or - Analytic normals with the specular
    # hint's gradient - the hit normal's own array (``nhat_bar``, _hint_backward)
    nb = nhat_bar if analytic else B.mbar[:, 3:6].contiguous()
    bg128 = torch.empty(n, 128, dtype=torch.float32, device=B.dev)
    _lib.check(lib.nrh_alpha_blend_backward(P_(pre["sdf"]), P_(res["normals"].view(Pn, 3)), P_(d), P_(res["dists"]), P_(res["inside"]), P_(og["bg_a"]),
                                            inv_s, cos_anneal, P_(dyn), n, P_(B.wbar), P_(nb), P_(og["tbar"]), P_(B.sdf_bar), P_(B.grad_bar),
                                            P_(B.rd_bar), P_(B.invs_bar), P_(bg128), stream), "nrh_alpha_blend_backward")
    B.grad_bar.add_(og["gn"])
    if og["bg_alpha"].requires_grad:          # background parameters and / or ray leaves upstream
        gba = og["gba"].clone()
        gba[:, :128] += bg128
        for name, prm in renderer.outside_nerf.named_parameters():
            if prm.requires_grad:
                v = B.pgrad["outside_nerf." + name]
                v.zero_()
                prm.grad = v      # AccumulateGrad adds in place: the gradient lands in the flat buffer's view
        with torch.enable_grad():
            torch.autograd.backward([og["bg_alpha"], og["bg_col"]], [gba, og["gbc"]])


def _hint_forward(renderer, lib, B, pk, res, o, d, pl, cos_anneal, shadow_grad: bool, specular_grad: bool, want_rays: bool) -> dict:
    """renderer.shadow_hint_gradient / specular_hint_gradient on the fused step (models/neus_hint_model.py:379, :589: the hints stay
    inside the graph).  The per-sample work is the HIP kernels' - the SDF training forward at the shadow ray's 128 sections with
    its own saved arrays, the shadow ray's alpha stage (nrh_shadow_alpha_forward) - and what is left is per-RAY arithmetic on
    [n, .] tensors (hit normal = normalised weighted sum of unit normals, the Cook-Torrance cue, the two encodings), which runs as
    a small torch autograd island whose leaves are exactly the quantities the kernels exchange adjoints in: the visibility [n,1],
    the unit normals [n,128,3] and the weights [n,128] (and, under pose / light refinement, the view direction and the light).
    The encodings it produces overwrite the kernel's (constant-hint) columns of the per-ray table before the reflectance forward."""
    from . import autograd_core, ops
    import torch.nn.functional as F
    P_ = _lib.ptr
    n = o.shape[0]
    hgi = renderer._hint_grad_inputs(o, d, res["depth"], res, shadow_grad, specular_grad)
    hit = hgi["hit"]
    out = dict(shadow=shadow_grad, specular=specular_grad)
    with torch.enable_grad():
        encs, leaves = [], []
        if shadow_grad:
            sd = hit - pl
            srd = (sd / torch.linalg.norm(sd, ord=2, dim=-1, keepdim=True)).contiguous()
            mid_s = res["shadow_mid_z"]
            pts_s = (pl[:, None, :] + srd[:, None, :] * mid_s[..., None]).reshape(-1, 3).contiguous()
            sdf_s, _, grad_s, sv_s = ops.sdf_train_forward(pk["sdf_w"], pk["sdf_b"], pk["sdf_head"], pts_s)
            vis = torch.empty(n, 1, dtype=torch.float32, device=o.device)
            _lib.check(lib.nrh_shadow_alpha_forward(P_(sdf_s), P_(grad_s), P_(srd), P_(res["shadow_dists"]), float(pk["inv_s"]), float(cos_anneal),
                                                    P_(renderer.dyn_scalars), n, int(renderer._shadow_total), P_(vis), _lib.stream_handle()),
                       "nrh_shadow_alpha_forward")
            vis_leaf = vis.requires_grad_(True)
            out.update(srd=srd, pts_s=pts_s, sdf_s=sdf_s, grad_s=grad_s, saves=sv_s, vis_leaf=vis_leaf, shadow_dists=res["shadow_dists"])
            enc_v = autograd_core._enc(vis_leaf, 4)
            B.raymisc[:, 54:63].copy_(enc_v.detach())
            encs.append(enc_v)
            leaves.append(vis_leaf)
        if specular_grad:
            nh_leaf = res["nhat"].detach().requires_grad_(True)
            w_leaf = res["weights"].detach().requires_grad_(True)
            d_leaf = d.detach().requires_grad_(bool(want_rays))
            pl_leaf = pl.detach().requires_grad_(bool(want_rays))
            hit_n = F.normalize((nh_leaf * w_leaf[..., None]).sum(1), dim=-1, p=2)                     # :586-587
            cue = autograd_core._specular_cue(hit_n, pl_leaf, hit, d_leaf, hgi["roughness"])
            enc_c = autograd_core._enc(cue, 4)
            B.raymisc[:, 63:99].copy_(enc_c.detach())
            encs.append(enc_c)
            out.update(nh_leaf=nh_leaf, w_leaf=w_leaf, d_leaf=d_leaf if want_rays else None, pl_leaf=pl_leaf if want_rays else None)
        out["encs"] = encs
    return out


def _hint_backward(hg: dict, B, n: int, mw: int, analytic: bool) -> None:
    """The island's backward: per-ray sums of the mbar columns of enc(visibility) [60:69] and enc(cue) [69:105] -> d loss / d
    visibility (hg["vis_bar"]), and the cue's adjoints added to the weights' (B.wbar) and the unit normals' (columns 3..5 of mbar,
    or - Analytic normals - their own array hg["nhat_bar"])."""
    mb = B.mbar.view(n, 128, mw)
    gouts = []
    if hg["shadow"]:
        gouts.append(mb[:, :, 60:69].sum(1))
    if hg["specular"]:
        gouts.append(mb[:, :, 69:105].sum(1))
    with torch.enable_grad():
        torch.autograd.backward(hg["encs"], gouts)
    hg["vis_bar"] = hg["vis_leaf"].grad.contiguous() if hg["shadow"] else None
    hg["nhat_bar"] = None
    if hg["specular"]:
        B.wbar.add_(hg["w_leaf"].grad)
        nb = hg["nh_leaf"].grad.reshape(-1, 3)
        if analytic:
            hg["nhat_bar"] = nb.contiguous()
        else:
            B.mbar[:, 3:6].add_(nb)
        if hg.get("d_leaf") is not None:
            hg["d_bar"], hg["pl_bar"] = hg["d_leaf"].grad, hg["pl_leaf"].grad


def _shadow_backward(hg: dict, lib, B, pk, n: int, inv_s: float, cos_anneal: float, dyn, n_real: int, stream) -> dict:
    """d loss / d visibility -> the shadow ray's alpha stage (nrh_shadow_alpha_backward) -> the SDF net's two sweeps at the shadow
    points; the per-ray 1 / s partials join the primary ray's (B.invs_bar) ahead of nrh_variance_grad."""
    from . import ops
    P_ = _lib.ptr
    dev = B.dev
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    Pn = n * 128
    sdf_bar, grad_bar, rd_bar, invs_bar = new(Pn), new(Pn, 3), new(n, 3), new(n)
    _lib.check(lib.nrh_shadow_alpha_backward(P_(hg["sdf_s"]), P_(hg["grad_s"]), P_(hg["srd"]), P_(hg["shadow_dists"]), inv_s, cos_anneal, P_(dyn), n,
                                             n_real, P_(hg["vis_bar"]), P_(sdf_bar), P_(grad_bar), P_(rd_bar), P_(invs_bar), stream),
               "nrh_shadow_alpha_backward")
    B.invs_bar.add_(invs_bar)
    fbar0 = torch.zeros(Pn, 256, dtype=torch.float32, device=dev)
    sv = hg["saves"]
    r = ops.sdf_train_backward(pk["sdf_w"], pk["sdf_wt_feat"], pk["sdf_head"], hg["pts_s"], sv["zeros3"], sv["zeros1"], 1, sv, sdf_bar, fbar0,
                               grad_bar, adj_scale=_lib.adjoint_scale(n))
    hg["sdf_bar"], hg["fbar0"] = sdf_bar, fbar0
    return r


def _finish_rays(B, ray_bundle, want_rays: bool, ray_grads) -> torch.Tensor:
    """Hand the ray adjoints on: into ``ray_grads`` when given, else into the autograd graph behind the bundle's tensors (the ray
    generator: one nrh_generate_rays_indexed_backward launch + the exponential map's tiny backward), as loss.backward() would."""
    if want_rays:
        pairs = [(t, g) for t, g in ((ray_bundle.origins, B.o_bar), (ray_bundle.directions, B.d_bar), (ray_bundle.pl_positions, B.pl_bar))
                 if torch.is_tensor(t) and t.requires_grad]
        far_bar = getattr(B, "far_bar", None)       # outside NeRF: the far bound positions the samples beyond the sphere
        if far_bar is not None and torch.is_tensor(ray_bundle.fars) and ray_bundle.fars.requires_grad:
            pairs.append((ray_bundle.fars, far_bar))
        if ray_grads is not None:
            ray_grads.update(origins=B.o_bar, directions=B.d_bar, pl_positions=B.pl_bar)
            if far_bar is not None:
                ray_grads["fars"] = far_bar
        elif pairs:
            torch.autograd.backward([t for t, _ in pairs], [g.to(t.dtype).reshape(t.shape) for t, g in pairs])
    return B.loss8


def loss_dict(loss8: torch.Tensor) -> Dict[str, float]:
    """One device-to-host copy -> the reference's loss dict (pipelines/base_pipeline.py:63-69) as python floats."""
    return dict(zip(LOSS_KEYS, loss8[:5].tolist()))
