"""``NeuSHintRenderer`` - drop-in for the reference class of the same name (models/neus_hint_model.py:236).

Same constructor argument (a ``NeuSModelConfig`` tree), same ``forward(ray_bundle, is_training, background_rgb,
global_step) -> RenderOutput`` signature, same parameter / state-dict key names
(``sdf_network.lin{0..7}.{bias,weight_g,weight_v}``, ``sdf_network.out_sdf.*``, ``sdf_network.out_feat.*``,
``deviation_network.variance``, ``color_network.lin{0..4}.*`` - SURVEY.md §5), so released reference checkpoints
load with ``load_state_dict`` and ``pipelines/base_pipeline.py:30`` can construct it unchanged.

All arithmetic of the forward runs in libnrhints_hip.so (hand-written gfx950 kernels) through the C ABI; this
module only owns the parameters, packs them, sizes the workspace and enqueues ONE C call per ray chunk on the
current torch stream.  There is no PyTorch/CPU fallback: without the extension or without a GPU it raises.
"""
from __future__ import annotations

import math
import ctypes
from typing import Optional

import numpy as np
import torch
from torch import nn

from . import _lib, autograd_core, packing, packing32
from .config import DepthComputationType, NeuSModelConfig, NormalComputationType, unsupported_reason
from .containers import RayBundle, RenderOutput

N_SAMPLES_TOTAL = 128


def _apply_wn(lin: nn.Module) -> nn.Module:
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return nn.utils.weight_norm(lin)  # old-style hook: parameters weight_g / weight_v, as in the reference


class SDFNetwork(nn.Module):
    """Parameter container + SAL geometric initialisation of the SDF MLP (fields/sdf_field.py:40-104).

    The initial zero level set is a sphere of radius ``init_bias``.  Layer/in/out dims: 39->256, 256->256 x2,
    256->217, 256->256 x4, heads 256->1 and 256->256.
    """

    def __init__(self, config):
        super().__init__()
        self.config = config
        emb = config.d_in * (2 * config.multi_res + 1)
        dims = [emb] + [config.d_hidden] * config.n_layers + [config.d_out_feat + 1]
        self.num_layers = len(dims)
        for l in range(self.num_layers - 2):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in config.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if config.geometric_init:
                std = math.sqrt(2.0) / math.sqrt(out_dim)
                nn.init.constant_(lin.bias, 0.0)
                if l == 0:
                    # only the raw xyz columns start non-zero: the net initially ignores the sinusoids
                    nn.init.constant_(lin.weight[:, 3:], 0.0)
                    nn.init.normal_(lin.weight[:, :3], 0.0, std)
                else:
                    nn.init.normal_(lin.weight, 0.0, std)
                    if l in config.skip_in:
                        nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
            if config.weight_norm:
                lin = _apply_wn(lin)
            setattr(self, f"lin{l}", lin)
        bias = config.init_bias * config.scale
        for name, out_dim in (("sdf", 1), ("feat", dims[-1] - 1)):
            lin = nn.Linear(dims[-2], out_dim)
            if config.geometric_init:
                sign = -1.0 if config.inside_outside else 1.0
                nn.init.normal_(lin.weight, mean=sign * math.sqrt(math.pi) / math.sqrt(dims[-1]), std=0.0001)
                nn.init.constant_(lin.bias, -sign * bias)
            if config.weight_norm:
                lin = _apply_wn(lin)
            setattr(self, f"out_{name}", lin)


class ReflectanceNetwork(nn.Module):
    """Parameter container of the reflectance MLP (fields/reflectance_network.py:26-66): 361 -> 4x256 -> 3."""

    def __init__(self, d_feature: int, d_in: int, d_out: int, config, n_cue: int, shadow_hint: bool = True):
        super().__init__()
        pe3 = 3 * 2 * config.multi_res          # extra dims of enc(view) / enc(pl) beyond the raw vector
        d0 = d_in + d_feature + 2 * pe3 + (1 if shadow_hint else 0) * 2 * config.multi_res + n_cue * 2 * config.multi_res
        dims = [d0] + [config.d_hidden] * config.n_layers + [d_out]
        for l in range(len(dims) - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if config.weight_norm:
                lin = _apply_wn(lin)
            setattr(self, f"lin{l}", lin)


class SingleVarianceNetwork(nn.Module):
    """NeuS sharpness: inv_s = exp(10 * variance) (models/neus_hint_model.py:104-110)."""

    def __init__(self, init_val: float):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(init_val)))


class NeuSHintRenderer(nn.Module):
    #: rays per C call; bounds the workspace (about 145 KB per ray + 268 MB of gradient scratch)
    max_chunk_rays = 131072    # rays per nrh_render_forward call: 18 GB of workspace (HBM is 288 GB); +1.3 % over 32 768 (profiles/r02/chunk_rays_ab.log)
    #: a batch larger than max_chunk_rays goes through in ONE call when its workspace fits whole_frame_fraction of the device's memory
    #: (an 800 x 800 frame: 640 000 rays, 93 GB of the 288 GB) and can be allocated: 5 x fewer launch boundaries (persistent-grid
    #: drains) per frame, +0.8 % on the headline (profiles/r06/chunk_ab.log); bit-identical by the re-chunking property
    #: (tests/test_gpu_fullsize.py).  0 = never (always max_chunk_rays).
    whole_frame_rays = 655360
    whole_frame_fraction = 0.4
    #: matrix arithmetic of the MLP kernels: "f32" (v_mfma_f32_16x16x4_f32, exact fp32) or "f16x3" (three
    #: v_mfma_f32_16x16x32_f16 per product on fp16 hi/lo splits: fp32-equivalent accuracy at 16/3 the matrix rate), or - a
    #: REDUCED-precision evaluation mode, never the default - "f16": f16x3 in everything (packing, training, the reflectance net)
    #: except that an evaluation render runs the wide SDF kernels in their single-pass builds (one fp16 MFMA per K step: weights
    #: and activations of the SDF network at 11 bits; PSNR against the reference 73-85 dB on the test scenes, CHANGELOG.md section 7h)
    precision = "f16x3"
    wide_kernels = True   # f16x3: evaluate the SDF network with the wide kernels (csrc/nrh_sdf32.hip); False = the 16-point kernels
    shadow_jvp = False         # the shadow march's last SDF evaluation in forward mode (mode 3: derivative along the ray only, no
                               # sigma' scratch): correct and tested, but 1.6 % slower per frame than reverse mode (profiles/r02/shadow_jvp_ab.log)
    wide_color = True          # ... and the reflectance net on the wide machinery as well (csrc/nrh_color32.hip; hinted model only)
    max_eval_rays_while_graphed = 32768   # training.GraphedTrainStep pins the workspace: evaluation chunks while a graph is alive
    fuse_feature_head = True   # evaluation renders with the wide kernels: W0feat * W_feat multiplied at pack time (NrhNet.feat_fused)
    max_fused_train_rays = 8192
    #: OPTION of the fused f16x3 training step (train_fused.py; batches the 8- / 4-wave kernels run): hand the weight-gradient operands
    #: (h, t, abar, zbar, the reflectance net's ReLU outputs / adjoints) to nrh_dw_gemm as fp16 - ONE fp16 MFMA pass on operands
    #: rounded to 11 bits, range-scaled per step - instead of float32 arrays and three-term products.  False (the default) is the
    #: precision-matched form: every product of the step then carries float32-equivalent operands, as the reference's float32 does.
    #: True is ~10 % faster per step and meets the same gradient bounds against the reference's float64 step
    #: (tests/test_gpu_train1024.py incl. the same-forward tests at the unwidened bound; a three-seed fit A/B shows no difference),
    #: but its dW PRODUCTS are narrower than float32, so it is opt-in and every number measured with it is labelled (bench.py:
    #: ``train.handoff16``).  An attribute of the renderer, never an environment switch.
    dw_half = False
    # hipGraph mode (training.GraphedTrainStep): a device tensor [inv_s, cos_anneal] that the kernels read at run time
    # instead of the host floats baked into a captured launch; None = normal (eager) operation
    dyn_scalars = None

    def __init__(self, config: NeuSModelConfig = None, precision: Optional[str] = None):
        super().__init__()
        if precision is not None:
            if precision not in _lib.PRECISIONS:
                raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
            self.precision = precision
        config = NeuSModelConfig() if config is None else config
        why = unsupported_reason(config)
        if why is not None:
            raise ValueError(f"NeuSHintRenderer (MI355X): unsupported configuration: {why}")
        self.config = config
        self.has_shadow_hint = bool(config.renderer.shadow_hint)
        self.has_specular_hint = bool(config.renderer.specular_hint)
        # one hint without the other (shadow_hint != specular_hint) runs on the kernels of the full nr-hints model: the missing
        # hint's columns of the reflectance net's first layer are zero (see _pad_hint_columns), its output is reported as None
        self._hints = 1 if (self.has_shadow_hint or self.has_specular_hint) else 0
        self._mixed_hints = self.has_shadow_hint != self.has_specular_hint
        # partial visibility hint (:553-575): shadow rays per group of samples instead of per ray; -1 = the hit-point mode
        self._shadow_clip = int(config.renderer.n_shadow_importance_clip) if self.has_shadow_hint else -1
        # n_importance_samples = 0 (:696): the 64 coarse samples are final; the kernels keep 128 slots per ray, the upper 64 as
        # padding with weight exactly 0, and the outputs are cut back to 64 here
        from .config import sample_counts
        cnt = sample_counts(config.renderer)
        # sample counts off the defaults (:139-171): kernel parameters carried by NrhNet (n_coarse .. lin_tables); None = the defaults,
        # which includes the n_importance_samples = 0 variant of the default 64 coarse samples (NrhNet.samples = 64)
        self._counts = None if cnt in ((64, 4, 16, 64, 16), (64, 0, 16, 64, 16)) else cnt
        self._samples = cnt[0] + cnt[1] * cnt[2]
        self._shadow_coarse = cnt[3]
        self._shadow_total = cnt[3] + 4 * cnt[4]      # samples of a shadow ray that exist (the shadow march always takes four steps, :373)
        self._normal_type = 1 if config.renderer.normal_type == NormalComputationType.Analytic else 0
        self._depth_type = {DepthComputationType.AlphaBlend: 0, DepthComputationType.MaximalWeightPoint: 1,
                            DepthComputationType.SphereTracing: 2}[config.renderer.depth_type]
        # the two free scalars of the renderer config (:161, :163) travel to the kernels through NrhNet when they are not the defaults
        rough, offs = [float(x) for x in config.renderer.specular_roughness], float(config.renderer.shadow_ray_offset)
        if not self.has_specular_hint:
            rough = [0.02, 0.05, 0.13, 0.34]      # no cue is computed: the values (any number of them) never reach a kernel
        self._net_consts = None if (rough == [0.02, 0.05, 0.13, 0.34] and offs == 1e-2) else (rough, offs)
        self.sdf_network = SDFNetwork(config.sdf_network)
        self.deviation_network = SingleVarianceNetwork(config.deviation_network.init_val)
        n_cue = len(config.renderer.specular_roughness) if self.has_specular_hint else 0
        self.color_network = ReflectanceNetwork(config.sdf_network.d_out_feat, 12 + int(self.has_shadow_hint) + n_cue, 3,
                                                config.reflectance_network, n_cue, self.has_shadow_hint)
        # widths / encoding resolutions BELOW the compiled ones run zero-padded on the compiled kernels (_to_compiled)
        sc_, cc_ = config.sdf_network, config.reflectance_network
        self._narrow = (sc_.d_hidden, sc_.multi_res, sc_.d_out_feat, cc_.d_hidden, cc_.multi_res) != (256, 6, 256, 256, 4)
        self.has_outside_nerf = bool(config.renderer.use_outside_nerf)
        if self.has_outside_nerf:      # constructed after the reflectance net, as the reference does (:260-266)
            from .outside import OutsideNeRF
            nc = config.outside_nerf
            self.outside_nerf = OutsideNeRF(4, 6, nc.d_hidden, nc.n_layers, nc.multi_res, nc.multi_res_view, nc.skips)
        self._packed = None
        self._pack_plan = None
        self._pack_plan32 = None
        self._packed_key = None
        self._ws = {}
        self._consts = {}

    # ---------------------------------------------------------------------------------------------
    def _param_key(self, device):
        # _generation: bumped by training.GraphedTrainStep after every replay (in-place updates inside a hipGraph do not
        # advance tensor version counters)
        return (str(device), self.precision, getattr(self, "_generation", 0)) + tuple((id(p), p._version) for p in self.parameters())

    def _pad_hint_columns(self, dense):
        """Reflectance net with ONE hint (fields/reflectance_network.py:44-52: 316 + 9 visibility or + 36 cue input columns):
        its first layer as the 361-column matrix of the two-hint layout, the missing hint's columns zero - the kernels then add
        exact zeros for that hint.  Differentiable (plain cat), so the autograd training path sees the right gradient."""
        if not self._mixed_hints:
            return dense
        w0 = dense["col_w0"]
        if self.has_shadow_hint:       # [.., vis 9] -> [.., vis 9, cue 36 = 0]
            w0 = torch.cat([w0, w0.new_zeros(w0.shape[0], 36)], dim=1)
        else:                          # [.., cue 36] -> [.., vis 9 = 0, cue 36]
            w0 = torch.cat([w0[:, :316], w0.new_zeros(w0.shape[0], 9), w0[:, 316:]], dim=1)
        return dict(dense, col_w0=w0)

    def _to_compiled(self, dense):
        """The folded matrices in the shape the kernels are compiled for: a narrower network (``_narrow``: widths / encoding
        resolutions below the defaults, config.unsupported_reason) zero-padded by packing.pad_to_compiled, a one-hint model of the
        default widths by _pad_hint_columns; the default nr-hints / pl-naive shapes pass through untouched (same tensors)."""
        if self._narrow:
            return packing.pad_to_compiled(dense, self.has_shadow_hint, self.has_specular_hint, int(self.config.reflectance_network.multi_res))
        return self._pad_hint_columns(dense)

    def packed_params(self, device, dense=None):
        """Fold weight-norm and pack for the kernels; cached until a parameter changes.  ``dense``: the already folded
        matrices of the CURRENT parameters (the training forward folds them once, with autograd history)."""
        key = self._param_key(device)
        stale = self._packed_key != key or (dense is not None and self._packed.get("col_wt") is None)
        if stale:
            with torch.no_grad():
                prec = _lib.PRECISIONS[self.precision]
                hints = bool(self._hints)
                variance = self.deviation_network.variance.detach().to(device=device, dtype=torch.float32)
                if dense is None:
                    state = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in self.state_dict().items()}
                    # ONE fold for every path on the GPU: nrh_weight_norm_fold, the kernel the training paths fold with.  The torch
                    # expression v * (g / |v|) differs from it in last bits, and a last-bit change of a weight moves importance
                    # samples by a whole bin where the pdf sits at its floor - with two folds an evaluation render and a training
                    # forward of the SAME parameters placed 20 % of their samples differently (profiles/train_forward_determinism.py,
                    # round 6).  (CPU tensors - the packing tests of the emulators - keep the torch expression.)
                    d = self._to_compiled(packing.dense_params_device(state))
                    packing.check_default_shapes(d, hints)
                    sw, sb, sh = packing.pack_sdf(d, prec)
                    cw, cb = packing.pack_color(d, prec, hints)
                    bufs = dict(sdf_w=sw, sdf_b=sb, sdf_head=sh, col_w=cw, col_b=cb,
                                sdf_wt_feat=packing.pack_feat_transposed(d, prec), col_wt=None)
                else:
                    # training: re-packed after every optimiser step -> the one-gather plan (packing.PackPlan)
                    d = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in dense.items()}
                    if self._pack_plan is None or not self._pack_plan.matches(d, prec, hints):
                        packing.check_default_shapes(d, hints)
                        self._pack_plan = packing.PackPlan(d, prec, hints)
                    bufs = self._pack_plan.pack(d)
                if prec == 1:
                    # the wide f16x3 evaluation kernels (csrc/nrh_sdf32.hip) read their own streams: every SDF evaluation
                    # of the no-grad stages goes through them (samplers, shadow march, render_core at evaluation)
                    if self._pack_plan32 is None or not self._pack_plan32.matches(d):
                        self._pack_plan32 = packing32.PackPlan32(d)
                    bufs["sdf_w32"], bufs["sdf_tab32"] = self._pack_plan32.pack(d)
                    if dense is None and self.fuse_feature_head:
                        # evaluation: a second set of streams with the feature head multiplied into the reflectance net's
                        # first layer (packing32.fuse_feature_head) - the render call then skips that block
                        bufs["sdf_w32f"], bufs["sdf_tab32f"] = packing32.pack_sdf32_fused(d)
                        if hints and self.wide_color:
                            bufs["col_w32"], bufs["col_tab32"] = packing32.pack_color32(d)
                # 1 / s = clip(exp(10 variance), 1e-6, 1e6) by ONE kernel in both modes (nrh_step_scalars): the host float of the
                # eager mode and the device scalar of the captured mode are the same bits, and a captured step spends one launch
                # on it instead of four pointwise torch kernels
                if self.dyn_scalars is not None:      # no host sync: the kernels read inv_s from the device
                    with torch.cuda.device(device):
                        _lib.step_scalars(variance=variance.reshape(1).contiguous(), inv_s_out=self.dyn_scalars)
                    inv_s = float("nan")
                    # ... so the f16x3 range check of the branch below cannot raise here.  It still runs, without a sync: every
                    # `range_check_every`-th pack enqueues the same test and copies its verdict to pinned memory; a later pack
                    # (or check_weight_range(), _host_inv_s) reads it once its event has passed and raises THEN (a
                    # diverging run in this mode must not render / train on inf weights silently)
                    if prec == 1:
                        self._range_guard_async(bufs, d, device)
                else:
                    # one host read: 1/s, and for f16x3 whether every packed weight survived the fp16 split (|w| times the
                    # kernels' input scaling must stay below 65504 - true for any trained NeuS net by orders of magnitude; a
                    # checkpoint outside that range is refused instead of rendered wrong)
                    ok = self._range_ok(bufs, d) if prec == 1 else torch.ones((), device=device)
                    if torch.device(device).type == "cuda":
                        s_dev = torch.empty(1, dtype=torch.float32, device=device)
                        with torch.cuda.device(device):
                            _lib.step_scalars(variance=variance.reshape(1).contiguous(), inv_s_out=s_dev)
                    else:       # (CPU tensors: host-side packing tests only)
                        s_dev = torch.exp(variance * 10.0).clip(1e-6, 1e6).reshape(1)
                    inv_s, ok = torch.stack([s_dev.reshape(()), ok.reshape(()).to(torch.float32)]).tolist()
                    if not ok:
                        raise ValueError(self._RANGE_MSG)
            self._packed = dict(bufs, inv_s=inv_s, precision=prec, hints=hints)
            self._packed_key = key
        return self._packed

    range_check_every = 256    # dyn-scalar mode (sync-free training): packs between two asynchronous f16x3 weight-range checks

    _RANGE_MSG = ("precision 'f16x3': a network weight is outside the fp16 range of the 3-term split "
                  "(|w| * 144.3 >= 65504); use precision='f32' for this checkpoint")

    @staticmethod
    def _range_ok(bufs, d) -> torch.Tensor:
        """0-dim float32 device tensor, 1 if every packed fp16 half is finite and every packed bias fits fp16"""
        halves = [v for v in bufs.values() if torch.is_tensor(v) and v.dtype == torch.float16]
        return torch.stack([torch.isfinite(v).all() for v in halves] + [packing32.tables_in_f16_range(d)]).all().to(torch.float32)

    def _range_guard_poll(self, wait: bool = False) -> None:
        g = self.__dict__.get("_range_guard")
        if not g or g["pending"] is None:
            return
        host, ev = g["pending"]
        if wait:
            ev.synchronize()
        if ev.query():
            g["pending"] = None
            if float(host) == 0.0:
                raise ValueError(self._RANGE_MSG)

    def _range_guard_async(self, bufs, d, device) -> None:
        g = self.__dict__.setdefault("_range_guard", {"n": 0, "pending": None})
        if torch.cuda.is_current_stream_capturing():
            # inside a capture nothing may be read back and no event may be queried (hipEventQuery from the capturing thread
            # invalidates a global / thread-local capture; a raise in the middle of one would leave it open): no poll, no new
            # verdict.  GraphedTrainStep drains a pending verdict before it starts capturing and calls check_weight_range()
            # between replays.
            return
        self._range_guard_poll()
        g["n"] += 1
        if g["pending"] is not None or (g["n"] - 1) % max(1, int(self.range_check_every)) != 0:
            return
        host = torch.ones((), dtype=torch.float32).pin_memory()
        host.copy_(self._range_ok(bufs, d), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        g["pending"] = (host, ev)

    def check_weight_range(self) -> None:
        """Synchronous form of the f16x3 weight-range guard for the modes that keep 1/s on the device (``train_step(sync=False)``,
        ``GraphedTrainStep``): raises the documented ``ValueError`` if the CURRENT packed weights left the fp16 range of the split."""
        self._range_guard_poll(wait=True)
        pk = self._packed
        if pk is None or pk.get("precision") != 1:
            return
        halves = [v for v in pk.values() if torch.is_tensor(v) and v.dtype == torch.float16]
        tests = [torch.isfinite(v).all() for v in halves]
        tab = pk.get("sdf_tab32")
        if torch.is_tensor(tab):
            # the bias rows of the wide kernels' table are packed fp16 pairs: what the eager pack tests on the dense parameters
            # (packing32.tables_in_f16_range: |b * IK| < 65504) shows up here as an fp16 half at infinity / NaN
            w = tab.reshape(-1, 256)[:9].view(torch.int32)           # rows 0..8: (b_hi | b_lo << 16) of packing32.sdf32_tables
            tests += [((w & 0x7fff) < 0x7c00).all(), (((w >> 16) & 0x7fff) < 0x7c00).all(), torch.isfinite(tab.reshape(-1, 256)[9:]).all()]
        if tests and not bool(torch.stack(tests).all().item()):
            raise ValueError(self._RANGE_MSG)

    def _count_args(self, device):
        """``counts`` of _lib.make_net: None for the default sample counts, else the five counts + the linspace tables"""
        if self._counts is None:
            return None
        k = "counts:" + str(device)
        if k not in self._consts:
            tab = torch.zeros(4, 128, dtype=torch.float32)
            for row, m in enumerate((self._counts[0], self._counts[2], self._counts[3], self._counts[4])):
                if m > 0:
                    tab[row, :m] = torch.linspace(0.0, 1.0, m)      # on the CPU, like lin64 / lin16: bit-identical to the reference's
            self._consts[k] = tab.to(device)
        return tuple(self._counts) + (self._consts[k],)

    def _const(self, device):
        k = str(device)
        if k not in self._consts:
            # linspace evaluated on the CPU so the tables are bit-identical to the reference's CPU/GPU values
            self._consts[k] = (torch.linspace(0.0, 1.0, 64).to(device), torch.linspace(0.0, 1.0, 16).to(device))
        return self._consts[k]

    def _pick_chunk(self, device, n: int) -> int:
        """Rays per nrh_render_forward call for a batch of ``n``: max_chunk_rays, or the whole batch when it is at most
        whole_frame_rays and its workspace is affordable (see whole_frame_rays); an allocation failure falls back."""
        if self.dyn_scalars is not None:
            return max(1, min(self.max_eval_rays_while_graphed, n))
        base = max(1, min(self.max_chunk_rays, n))
        if n <= base or not self.whole_frame_rays or n > self.whole_frame_rays or self.max_chunk_rays != type(self).max_chunk_rays:
            return base            # (an explicitly set max_chunk_rays is a request to chunk: tests of the re-chunking property, A/B runs)
        need = int(_lib.load().nrh_render_workspace_floats(n)) * 4
        ws = self._ws.get(str(device))
        if ws is not None and ws.numel() * 4 >= need:
            return n
        if need > self.whole_frame_fraction * torch.cuda.get_device_properties(device).total_memory:
            return base
        try:
            self._workspace(device, n)
            return n
        except torch.cuda.OutOfMemoryError:
            return base

    def _workspace(self, device, nrays):
        lib = _lib.load()
        need = int(lib.nrh_render_workspace_floats(nrays))
        k = str(device)
        ws = self._ws.get(k)
        if ws is None or ws.numel() < need:
            self._ws[k] = ws = torch.empty(need, dtype=torch.float32, device=device)
        return ws

    # ---------------------------------------------------------------------------------------------
    def forward(self, ray_bundle: RayBundle, is_training: bool = False, background_rgb: Optional[torch.Tensor] = None,
                global_step: int = 0, _t_rand_primary: Optional[torch.Tensor] = None,
                _t_rand_shadow: Optional[torch.Tensor] = None, _t_rand_outside: Optional[torch.Tensor] = None) -> RenderOutput:
        o = ray_bundle.origins
        if self.has_outside_nerf:
            if not (torch.is_tensor(o) and o.is_cuda):
                raise RuntimeError("NeuSHintRenderer (MI355X) runs on the GPU only")
            with torch.cuda.device(o.device):
                return self._forward_outside(ray_bundle, is_training, background_rgb, global_step, _t_rand_primary, _t_rand_shadow,
                                             _t_rand_outside)
        if torch.is_tensor(o) and o.is_cuda:
            # every C call below launches on "the current stream of the current device": make that the rays' device
            with torch.cuda.device(o.device):
                return self._forward(ray_bundle, is_training, background_rgb, global_step, _t_rand_primary, _t_rand_shadow)
        return self._forward(ray_bundle, is_training, background_rgb, global_step, _t_rand_primary, _t_rand_shadow)

    def _forward(self, ray_bundle, is_training, background_rgb, global_step, _t_rand_primary, _t_rand_shadow) -> RenderOutput:
        lib = _lib.load()
        o, d, pl = ray_bundle.origins, ray_bundle.directions, ray_bundle.pl_positions
        near, far = ray_bundle.nears, ray_bundle.fars
        if near is None or far is None:
            raise ValueError("ray_bundle.nears / fars are required")
        if o.dim() != 2 or o.shape[-1] != 3 or d.shape != o.shape or pl.shape != o.shape:
            raise ValueError(f"origins/directions/pl_positions must all be [N,3], got {tuple(o.shape)}, "
                             f"{tuple(d.shape)}, {tuple(pl.shape)}")
        n = o.shape[0]
        if near.numel() != n or far.numel() != n:
            raise ValueError("nears / fars must be [N,1]")
        if not o.is_cuda:
            raise RuntimeError("NeuSHintRenderer (MI355X) runs on the GPU only: move the RayBundle to cuda "
                               "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")
        # A graph is needed exactly when the reference would build one: grad mode on and something upstream of the
        # output requires grad (parameters in training / checkpoint fine-tuning, ray tensors in register_view).
        needs_grad = torch.is_grad_enabled() and (
            any(t.requires_grad for t in (o, d, pl)) or any(p.requires_grad for p in self.parameters()))
        o_g, d_g, pl_g = o, d, pl
        device = o.device
        f32 = lambda t: t.detach().to(dtype=torch.float32).contiguous()
        o, d, pl = f32(o), f32(d), f32(pl)
        near, far = f32(near).reshape(-1), f32(far).reshape(-1)
        cfg = self.config
        cos_anneal = 1.0
        if is_training and cfg.anneal_end > 0:
            cos_anneal = min(1.0, global_step / cfg.anneal_end)
        zero_hints = 1 if (is_training and global_step < cfg.geometry_warmup_end) else 0
        t_rand_p = t_rand_s = None
        if is_training:
            # same draw order as the reference: primary jitter [N,1] (:682), then shadow jitter [N,64] (:394)
            t_rand_p = f32(_t_rand_primary).reshape(-1) if _t_rand_primary is not None else torch.rand(n, device=device)
            if not zero_hints:
                # one row of 64 per shadow ray: per primary ray (:394), or per sample group in the partial mode (same order as the
                # reference's mini-batches over the flattened [ray, group] list, :560-568)
                rows = n * max(1, self._shadow_clip)
                sc = self._shadow_coarse           # (renderer.n_shadow_samples: the jitter has the shape of the coarse shadow z, :394)
                t_rand_s = f32(_t_rand_shadow) if _t_rand_shadow is not None else torch.rand(rows, sc, device=device)
                if t_rand_s.shape != (rows, sc):
                    raise ValueError(f"_t_rand_shadow must be [{rows}, {sc}]")
        bg = None
        if background_rgb is not None:
            bg = f32(background_rgb.to(device)).reshape(-1)
            if bg.numel() != 3:
                raise ValueError("background_rgb must be [1,3]")

        dense = None
        if needs_grad:
            # fold weight-norm once, with autograd history; the kernels' packed copies are cut from the same matrices
            named = dict(self.named_parameters())
            on_gpu_f32 = all(p.is_cuda and p.dtype == torch.float32 for p in named.values())
            dense = self._to_compiled(packing.dense_params_hip(named) if on_gpu_f32 else packing.dense_params(named))
            self.packed_params(device, dense=dense)
        fused_train = needs_grad and n <= self.max_fused_train_rays
        rcfg = cfg.renderer
        shadow_grad = bool(needs_grad and rcfg.shadow_hint_gradient and self.has_shadow_hint and not zero_hints)
        specular_grad = bool(needs_grad and rcfg.specular_hint_gradient and self.has_specular_hint and not zero_hints)
        if self._shadow_clip > 0 and needs_grad and not fused_train:
            raise ValueError(f"training with n_shadow_importance_clip > 0 needs the batch in one call (at most max_fused_train_rays = {self.max_fused_train_rays} rays)")
        if self._shadow_clip > 0 and shadow_grad:
            raise NotImplementedError("shadow_hint_gradient together with n_shadow_importance_clip > 0 is not implemented")
        if (shadow_grad or specular_grad) and not fused_train:
            raise ValueError(f"hint gradients need the batch in one training call (at most max_fused_train_rays = {self.max_fused_train_rays} rays)")
        if shadow_grad and any(t.requires_grad for t in (o_g, d_g, pl_g)):
            # the reference's coarse shadow samples scale with |hit - light| (:386-387), so d visibility / d light has a term through
            # the sample positions; the shadow sections come out of the HIP sampler as constants here
            raise NotImplementedError("shadow_hint_gradient together with ray gradients (pose / light refinement) is not implemented")
        if fused_train:
            res = self._render_train(o, d, pl, near, far, cos_anneal, t_rand_p, t_rand_s, zero_hints, want_shadow=shadow_grad)
        else:
            res = self._render_chunks(o, d, pl, near, far, bg, cos_anneal, t_rand_p, t_rand_s, zero_hints,
                                      want_samples=True, want_maps=False, want_mid=needs_grad, use_dyn=needs_grad and is_training)
        rgb, depth, vis = res.get("rgb"), res["depth"], res["visibilities"]
        weights, inside, normals, nhat, cue = (res[k] for k in ("weights", "inside", "normals", "nhat", "cue"))
        mid_z, dists = res.get("mid_z"), res.get("dists")
        pk = self.packed_params(device)
        T = self._samples
        cut = (lambda t: t) if T == N_SAMPLES_TOTAL else (lambda t: None if t is None else t[:, :T])
        if needs_grad:
            # differentiable part (render_core) over the HIP results; see autograd_core.py
            core = autograd_core.render_core(
                dense, self.deviation_network.variance, o_g.to(torch.float32), d_g.to(torch.float32),
                pl_g.to(torch.float32), mid_z, dists, (res.get("vis_groups", vis) if self._hints else None),
                cue[:, 0, :].contiguous() if self._hints else None, cos_anneal,
                background_rgb.to(device) if background_rgb is not None else None, analytic_normal=bool(self._normal_type),
                packed=pk, pre=res.get("pre"), dyn=self.dyn_scalars if is_training else None,
                hint_grad=self._hint_grad_inputs(o, d, depth, res, shadow_grad, specular_grad), n_real=T)
            return RenderOutput(rgb=core["rgb"], depth=depth, weights=cut(core["weights"]), s_val=cut(core["s_val"]),
                                inside_sphere=cut(inside), relax_inside_sphere=cut(inside),
                                analytic_normals=cut(core["analytic_normals"]),
                                normalized_analytic_normals=cut(core["normalized_analytic_normals"]),
                                visibilities=vis if self.has_shadow_hint else None,
                                specular_cue=cut(cue) if self.has_specular_hint else None)
        s_val = torch.full((1, 1), 1.0 / self._host_inv_s(pk, device), dtype=torch.float32, device=device).expand(n, T)
        return RenderOutput(rgb=rgb, depth=depth, weights=cut(weights), s_val=s_val, inside_sphere=cut(inside),
                            relax_inside_sphere=cut(inside), analytic_normals=cut(normals),
                            normalized_analytic_normals=cut(nhat), visibilities=vis if self.has_shadow_hint else None,
                            specular_cue=cut(cue) if self.has_specular_hint else None)

    # ---------------------------------------------------------------------------------------------
    max_outside_rays = 65536    # rays per pass of the outside-NeRF branch (evaluation; its [N,160] elementwise glue lives in torch)

    def _forward_outside(self, ray_bundle, is_training, background_rgb, global_step, t_p, t_s, t_o) -> RenderOutput:
        """``forward`` with the outside-NeRF background (renderer.use_outside_nerf; see outside.py for who computes what)."""
        from . import outside
        from .containers import td_concat
        n = ray_bundle.origins.shape[0]
        dev = ray_bundle.origins.device
        if is_training:      # the reference's draw order: primary [N,1] (:682), outside [N,32] (:689), shadow [N,64] (:394)
            t_p = torch.rand(n, 1, device=dev) if t_p is None else t_p.to(dev)
            t_o = torch.rand(n, outside.N_OUTSIDE, device=dev) if t_o is None else t_o.to(dev)
            warm = global_step < self.config.geometry_warmup_end
            t_s = (None if warm else torch.rand(n, 64, device=dev)) if t_s is None else t_s.to(dev)
        outs = []
        for i in range(0, n, self.max_outside_rays):
            sl = slice(i, i + self.max_outside_rays)
            rb = RayBundle(origins=ray_bundle.origins[sl], directions=ray_bundle.directions[sl], pl_positions=ray_bundle.pl_positions[sl],
                           nears=ray_bundle.nears[sl], fars=ray_bundle.fars[sl])
            cut = lambda t: None if t is None else t[sl]
            outs.append(self._outside_pass(rb, is_training, background_rgb, global_step, cut(t_p), cut(t_s), cut(t_o)))
        return outs[0] if len(outs) == 1 else td_concat(outs)

    def _outside_pass(self, ray_bundle, is_training, background_rgb, global_step, t_p, t_s, t_o) -> RenderOutput:
        from . import outside
        lib = _lib.load()
        o_g, d_g, pl_g = ray_bundle.origins, ray_bundle.directions, ray_bundle.pl_positions
        device, n = o_g.device, o_g.shape[0]
        needs_grad = torch.is_grad_enabled() and (
            any(t.requires_grad for t in (o_g, d_g, pl_g)) or any(p.requires_grad for p in self.parameters()))
        f32 = lambda t: t.detach().to(dtype=torch.float32).contiguous()
        o, d, pl = f32(o_g), f32(d_g), f32(pl_g)
        near, far = f32(ray_bundle.nears).reshape(-1), f32(ray_bundle.fars).reshape(-1)
        cfg = self.config
        cos_anneal = min(1.0, global_step / cfg.anneal_end) if (is_training and cfg.anneal_end > 0) else 1.0
        zero_hints = 1 if (is_training and global_step < cfg.geometry_warmup_end) else 0
        t_rand_p = f32(t_p).reshape(-1) if is_training else None
        t_rand_s = f32(t_s) if (is_training and not zero_hints and self._hints) else None
        bgc = None if background_rgb is None else f32(background_rgb.to(device)).reshape(1, 3)
        dense = None
        if needs_grad:
            named = dict(self.named_parameters())
            dense = self._to_compiled(packing.dense_params_hip({k: v for k, v in named.items() if not k.startswith("outside_nerf.")}))
            self.packed_params(device, dense=dense)
        pk = self.packed_params(device)
        if self.dyn_scalars is not None:
            raise RuntimeError("the outside-NeRF branch runs eagerly (release the GraphedTrainStep first)")
        # 1. where the primary ray's samples are (the background network needs them before the alpha stage can blend)
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=device)
        z, mid0, dists0 = new(n, 128), new(n, 128), new(n, 128)
        lin64, lin16 = self._const(device)
        net0 = _lib.make_net(pk, self._hints, self._normal_type, self._depth_type, None, wide=self.wide_kernels, consts=self._net_consts)
        ws = self._workspace(device, n)
        P = _lib.ptr
        _lib.check(lib.nrh_sample_primary(net0, P(o), P(d), P(near), P(far), n, P(t_rand_p), P(lin64), P(lin16), P(z), P(mid0), P(dists0),
                                          P(ws), ws.numel(), _lib.stream_handle()), "nrh_sample_primary")
        # 2. the background at the merged positions (:715-724): the network is csrc/nrh_outside.hip (outside.OutsideNetHip)
        self.outside_nerf.precision = self.precision
        with torch.set_grad_enabled(needs_grad):
            far_g = ray_bundle.fars.to(torch.float32).reshape(-1, 1)
            z_out = outside.outside_z(far_g, cfg.renderer.n_samples, f32(t_o) if is_training else None)
            z_feed, _ = torch.sort(torch.cat([z, z_out], dim=-1), dim=-1)
            bg_alpha, bg_col = outside.render_outside(self.outside_nerf, o_g.to(torch.float32), d_g.to(torch.float32), pl_g.to(torch.float32),
                                                      z_feed, 2.0 / cfg.renderer.n_samples)
        bg_a = bg_alpha.detach().contiguous()
        extra = dict(bg_alpha=bg_a, tail_t=new(n), sampled_color=None)
        if not needs_grad:
            extra["sampled_color"] = new(n, 128, 3)
            res = self._render_chunks(o, d, pl, near, far, bgc.reshape(-1) if bgc is not None else None, cos_anneal, t_rand_p, t_rand_s, zero_hints,
                                      want_samples=True, want_maps=False, want_mid=False, extra_net=extra)
            comp = outside.composite(res["weights"], extra["tail_t"].reshape(n, 1), res["inside"], extra["sampled_color"], bg_a, bg_col, bgc)
            s_val = torch.full((1, 1), 1.0 / self._host_inv_s(pk, device), dtype=torch.float32, device=device).expand(n, 128)
            return RenderOutput(rgb=comp["rgb"], depth=res["depth"], weights=comp["weights"], s_val=s_val, inside_sphere=res["inside"],
                                relax_inside_sphere=res["inside"], analytic_normals=res["normals"], normalized_analytic_normals=res["nhat"],
                                visibilities=res["visibilities"] if self.has_shadow_hint else None,
                                specular_cue=res["cue"] if self.has_specular_hint else None)
        if n > self.max_fused_train_rays:
            raise ValueError("training with use_outside_nerf needs at most max_fused_train_rays rays per call")
        rcfg = cfg.renderer
        if not zero_hints and ((rcfg.shadow_hint_gradient and self.has_shadow_hint) or (rcfg.specular_hint_gradient and self.has_specular_hint)):
            raise NotImplementedError("use_outside_nerf together with shadow_hint_gradient / specular_hint_gradient trains on the fused step "
                                      "(training.train_step / GraphedTrainStep / train_fused.train_step_backward), not through forward() + backward()")
        res = self._render_train(o, d, pl, near, far, cos_anneal, t_rand_p, t_rand_s, zero_hints, extra_net=extra)
        mid_z, dists, inside = res["mid_z"], res["dists"], res["inside"]
        pts = (o_g.to(torch.float32)[:, None, :] + d_g.to(torch.float32)[:, None, :] * mid_z[..., None]).reshape(-1, 3)
        sdf, feat, grad = autograd_core.sdf_value_feat_grad(dense, pts, packed=pk, pre=res.get("pre"))
        weights128, n_hat, tail_t = outside.AlphaBlendHip.apply(sdf, grad, d_g.to(torch.float32), dists, inside, bg_alpha,
                                                                self.deviation_network.variance, pk["inv_s"], cos_anneal, None)
        per_ray = [autograd_core._enc(d_g.to(torch.float32), 4), autograd_core._enc(pl_g.to(torch.float32), 4)]
        if self._hints:
            per_ray += [autograd_core._enc(res["visibilities"], 4), autograd_core._enc(res["cue"][:, 0, :].contiguous(), 4)]
        normal = grad if self._normal_type else n_hat
        col = autograd_core.ColorNetHip.apply(feat, pts, normal, torch.cat(per_ray, dim=-1), pk,
                                              *[dense[f"col_w{l}"] for l in range(5)], *[dense[f"col_b{l}"] for l in range(5)]).reshape(n, 128, 3)
        comp = outside.composite(weights128, tail_t, inside, col, bg_alpha, bg_col, bgc)
        inv_s = torch.exp(self.deviation_network.variance * 10.0).clip(1e-6, 1e6)
        return RenderOutput(rgb=comp["rgb"], depth=res["depth"], weights=comp["weights"], s_val=(1.0 / inv_s).expand(n, 128),
                            inside_sphere=inside, relax_inside_sphere=inside, analytic_normals=grad.reshape(n, 128, 3),
                            normalized_analytic_normals=n_hat.reshape(n, 128, 3),
                            visibilities=res["visibilities"] if self.has_shadow_hint else None,
                            specular_cue=res["cue"] if self.has_specular_hint else None)

    def _hint_grad_inputs(self, o, d, depth, res, shadow_grad: bool, specular_grad: bool):
        """What render_core needs to differentiate the hints (renderer.shadow_hint_gradient / specular_hint_gradient,
        models/neus_hint_model.py:379, :589): the graph-less hit point and, for the visibility, the shadow ray's sections."""
        if not (shadow_grad or specular_grad):
            return None
        if self._depth_type == 2:      # SphereTracing: the hit point is the tracer's last point (:527-528), not o + d * depth
            hit = self.sphere_trace(o, d, 2000, 1e-4, 100.0)[0]
        else:
            hit = o + d * depth.reshape(-1, 1)     # :533, :538 (under no_grad there too)
        return dict(hit=hit, specular=specular_grad, shadow=dict(mid_z=res["shadow_mid_z"], dists=res["shadow_dists"], n_real=self._shadow_total) if shadow_grad else None,
                    roughness=[float(r) for r in self.config.renderer.specular_roughness])

    # ---------------------------------------------------------------------------------------------
    def _render_train(self, o, d, pl, near, far, cos_anneal, t_rand_p, t_rand_s, zero_hints, raymisc=None, want_shadow=False,
                      extra_net=None, half_handoffs=False, pts=None):
        """One nrh_render_forward_train call over the whole batch: the no-grad stages (samplers, hit point, shadow march,
        cue) plus the training evaluation of the SDF network at the section mid-points, whose outputs and saved arrays
        feed the backward sweeps directly (no second evaluation).  ``half_handoffs`` (the fused step, f16x3, batches the 8-wave
        kernels run): layers 0..6 of h and a copy of layers 1..7 of t go to float16 arrays ``saves["h16"]`` / ``saves["t16"]`` in the
        half-tiled layout nrh_dw_gemm reads with one fp16 MFMA pass (include/nrhints_hip.h, nrh_sdf_train_forward_half).
        ``pts`` [n*128,3]: receives p = o + d * mid_z from the alpha stage (the reflectance net's point input)."""
        lib = _lib.load()
        device = o.device
        n = o.shape[0]
        pk = self.packed_params(device)
        lin64, lin16 = self._const(device)
        net = _lib.make_net(pk, self._hints, self._normal_type, self._depth_type, self.dyn_scalars, wide=self.wide_kernels,
                            shadow_clip=self._shadow_clip, samples=self._samples, consts=self._net_consts, counts=self._count_args(device),
                            **(extra_net or {}))
        T = N_SAMPLES_TOTAL
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=device)
        out = dict(depth=new(n, 1), visibilities=new(n, 1), weights=new(n, T), inside=new(n, T), normals=new(n, T, 3),
                   nhat=new(n, T, 3), cue=new(n, T, 4), mid_z=new(n, T), dists=new(n, T))
        pre = dict(sdf=new(n * T, 1), feat=new(n * T, 256), grad=out["normals"].reshape(n * T, 3),
                   saves=dict(h=new(8, n * T, 256), s1=new(8, n * T, 256), t=new(8, n * T, 256), ge=new(n * T, 128)),
                   ro=o, rd=d, t=out["mid_z"], n_per_ray=T)
        P = _lib.ptr
        sv = pre["saves"]
        if want_shadow:       # shadow_hint_gradient: the shadow ray's sections, for the differentiable visibility of render_core
            out.update(shadow_mid_z=new(n, T), shadow_dists=new(n, T))
        if self._shadow_clip > 0 and not zero_hints:
            out.update(vis_groups=new(n * self._shadow_clip, 1))
        if half_handoffs:
            sv["h16"] = torch.empty(8, n * T, 256, dtype=torch.float16, device=device)
            sv["t16"] = torch.empty(8, n * T, 256, dtype=torch.float16, device=device)
        saves = _lib.NrhTrainSaves(P(pre["sdf"]), P(pre["feat"]), P(sv["h"]), P(sv["s1"]), P(sv["t"]), P(sv["ge"]), P(raymisc),
                                   P(out.get("shadow_mid_z")), P(out.get("shadow_dists")), P(out.get("vis_groups")),
                                   P(sv.get("h16"), torch.float16), P(sv.get("t16"), torch.float16), P(pts))
        ws = self._workspace(device, n)
        rc = lib.nrh_render_forward_train(
            net, P(o), P(d), P(pl), P(near), P(far), n, cos_anneal, P(t_rand_p) if t_rand_p is not None else None,
            P(t_rand_s) if t_rand_s is not None else None, zero_hints, P(lin64), P(lin16), P(out["depth"]), P(out["weights"]),
            P(out["inside"]), P(out["normals"]), P(out["nhat"]), P(out["visibilities"]), P(out["cue"]), P(out["mid_z"]),
            P(out["dists"]), ctypes.byref(saves), P(ws), ws.numel(), _lib.stream_handle())
        _lib.check(rc, "nrh_render_forward_train")
        out["pre"] = pre
        return out

    def _host_inv_s(self, pk, device) -> float:
        """1/s as a host float also while a captured training graph keeps it on the device (one sync; evaluation only)."""
        if pk["inv_s"] == pk["inv_s"]:      # not NaN
            return pk["inv_s"]
        self._range_guard_poll()            # (this call synchronises anyway: surface a pending range verdict here)
        v = self.deviation_network.variance.detach().to(device=device, dtype=torch.float32)
        return float(torch.exp(v * 10.0).clip(1e-6, 1e6).item())

    def _render_chunks(self, o, d, pl, near, far, bg, cos_anneal, t_rand_p, t_rand_s, zero_hints, want_samples: bool,
                       want_maps: bool, want_mid: bool, use_dyn: bool = False, want_ray_cue: bool = False, extra_net=None):
        """Enqueue nrh_render_forward per chunk of rays; returns a dict of freshly allocated output tensors.
        want_samples: materialise the per-sample RenderOutput fields; want_maps: the per-pixel normal maps of the
        evaluation loop; want_mid: section mid-points / lengths (for the autograd training path); want_ray_cue: the
        specular hint once per ray, [n,4] (the reference's [n,128,4] specular_cue is that row repeated per sample)."""
        lib = _lib.load()
        device = o.device
        n = o.shape[0]
        pk = self.packed_params(device)
        lin64, lin16 = self._const(device)
        # the device-side scalars (1/s, cos-anneal ratio of a captured TRAINING step) are for training calls only: an
        # evaluation render between graph replays uses 1/s of the current variance and the cos_anneal it was given
        if not use_dyn and self.dyn_scalars is not None:
            pk = dict(pk, inv_s=self._host_inv_s(pk, device))
        net = _lib.make_net(pk, self._hints, self._normal_type, self._depth_type, self.dyn_scalars if use_dyn else None,
                            wide=self.wide_kernels, fused=self.fuse_feature_head and not want_mid, wide_color=self.wide_color,
                            shadow_jvp=self.shadow_jvp, shadow_clip=self._shadow_clip, samples=self._samples, consts=self._net_consts,
                            counts=self._count_args(device), single_pass=(self.precision == "f16" and not want_mid),
                            **(extra_net or {}))
        T = N_SAMPLES_TOTAL
        new = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=device)
        out = dict(rgb=new(n, 3), depth=new(n, 1), visibilities=new(n, 1))
        if want_samples:
            out.update(weights=new(n, T), inside=new(n, T), normals=new(n, T, 3), nhat=new(n, T, 3), cue=new(n, T, 4))
        if want_maps:
            out.update(normal_map=new(n, 3), normalized_normal_map=new(n, 3))
        if want_mid:
            out.update(mid_z=new(n, T), dists=new(n, T))
        chunk = self._pick_chunk(device, n)
        cue_scratch = None
        if want_ray_cue and not want_samples:
            out.update(cue_ray=new(n, 4))
            cue_scratch = new(chunk, T, 4)
        ws = self._workspace(device, chunk)
        stream = _lib.stream_handle()
        P = _lib.ptr
        opt = lambda key, sl: P(out[key][sl]) if key in out else None
        rows_per_ray = max(1, self._shadow_clip)      # rows of the shadow jitter per primary ray
        for i in range(0, n, chunk):
            m = min(chunk, n - i)
            sl = slice(i, i + m)
            rc = lib.nrh_render_forward(
                net, P(o[sl]), P(d[sl]), P(pl[sl]), P(near[sl]), P(far[sl]), m, P(bg), cos_anneal,
                P(t_rand_p[sl]) if t_rand_p is not None else None,
                P(t_rand_s[i * rows_per_ray:(i + m) * rows_per_ray]) if t_rand_s is not None else None, zero_hints, P(lin64), P(lin16),
                P(out["rgb"][sl]), P(out["depth"][sl]), opt("weights", sl), opt("inside", sl), opt("normals", sl),
                opt("nhat", sl), P(out["visibilities"][sl]), P(cue_scratch) if cue_scratch is not None else opt("cue", sl),
                opt("mid_z", sl), opt("dists", sl),
                opt("normal_map", sl), opt("normalized_normal_map", sl), P(ws), ws.numel(), stream)
            _lib.check(rc, "nrh_render_forward")
            if cue_scratch is not None:
                out["cue_ray"][sl] = cue_scratch[:m, 0, :]
        if want_ray_cue and want_samples:
            out["cue_ray"] = out["cue"][:, 0, :]
        return out

    @torch.no_grad()
    def render_products(self, ray_bundle: RayBundle, background_rgb: Optional[torch.Tensor] = None,
                        specular_cue: bool = False):
        """Evaluation fast path: per-PIXEL products only - rgb, depth, shadow map (visibilities) and the two weighted
        normal maps in world space - 15 floats per ray instead of the 1 669 of a full RenderOutput
        (what pipelines/base_pipeline.py:110-133 copies to the host per 512-ray chunk and reduces on the CPU)."""
        o, d, pl = ray_bundle.origins, ray_bundle.directions, ray_bundle.pl_positions
        if not o.is_cuda:
            raise RuntimeError("NeuSHintRenderer (MI355X) runs on the GPU only")
        if self.has_outside_nerf:
            # with a background network the per-pixel products are reduced from the full output (the composite happens on the host
            # side of this branch, outside.py); same keys as below
            out = self(ray_bundle, is_training=False, background_rgb=background_rgb)
            w = (out.weights[:, :N_SAMPLES_TOTAL] * out.inside_sphere)[..., None]
            res = dict(rgb=out.rgb, depth=out.depth, visibilities=out.visibilities if out.visibilities is not None else torch.zeros_like(out.depth),
                       normal_map=(out.analytic_normals * w).sum(1), normalized_normal_map=(out.normalized_analytic_normals * w).sum(1))
            if specular_cue and out.specular_cue is not None:
                res["cue_ray"] = out.specular_cue[:, 0, :]
            return res
        f32 = lambda t: t.detach().to(dtype=torch.float32).contiguous()
        bg = f32(background_rgb.to(o.device)).reshape(-1) if background_rgb is not None else None
        with torch.cuda.device(o.device):
            return self._render_chunks(f32(o), f32(d), f32(pl), f32(ray_bundle.nears).reshape(-1),
                                       f32(ray_bundle.fars).reshape(-1), bg, 1.0, None, None, 0,
                                       want_samples=False, want_maps=True, want_mid=False,
                                       want_ray_cue=specular_cue and bool(self._hints))

    # ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def sphere_trace(self, rays_o: torch.Tensor, rays_d: torch.Tensor, num_iterations: int, convergence_threshold: float,
                     far: float):
        """``NeuSHintRenderer.sphere_trace`` (models/neus_hint_model.py:359-372) -> (points [N,3], depths [N,1])."""
        lib = _lib.load()
        if not rays_o.is_cuda:
            raise RuntimeError("NeuSHintRenderer.sphere_trace runs on the GPU only")
        o = rays_o.detach().to(torch.float32).contiguous()
        d = rays_d.detach().to(torch.float32).contiguous()
        n, dev = o.shape[0], o.device
        pk = self.packed_params(dev)
        if self.dyn_scalars is not None:
            pk = dict(pk, inv_s=self._host_inv_s(pk, dev))
        net = _lib.make_net(pk, self._hints, self._normal_type, self._depth_type, None, wide=self.wide_kernels, consts=self._net_consts)
        pts, depth = torch.empty(n, 3, dtype=torch.float32, device=dev), torch.empty(n, 1, dtype=torch.float32, device=dev)
        ws = torch.empty(int(lib.nrh_sphere_trace_workspace_floats(n)), dtype=torch.float32, device=dev)
        P = _lib.ptr
        with torch.cuda.device(dev):
            rc = lib.nrh_sphere_trace(net, P(o), P(d), n, int(num_iterations), float(convergence_threshold), float(far), P(pts), P(depth),
                                      P(ws), ws.numel(), _lib.stream_handle())
        _lib.check(rc, "nrh_sphere_trace")
        return pts, depth

    @torch.no_grad()
    def sdf(self, pts: torch.Tensor) -> torch.Tensor:
        """SDF values at free points [P,3] -> [P,1] (SDFNetwork.sdf; used by extract_fields,
        models/neus_hint_model.py:68-83, 753-758)."""
        lib = _lib.load()
        if not pts.is_cuda:
            raise RuntimeError("NeuSHintRenderer.sdf runs on the GPU only")
        pts = pts.detach().to(torch.float32).contiguous()
        n = pts.shape[0]
        pk = self.packed_params(pts.device)
        zeros3 = torch.zeros_like(pts)
        t = torch.zeros(n, dtype=torch.float32, device=pts.device)
        out = torch.empty(n, 1, dtype=torch.float32, device=pts.device)
        P = _lib.ptr
        with torch.cuda.device(pts.device):
            rc = lib.nrh_sdf_eval(pk["precision"], 0, P(pk["sdf_w"], pk["sdf_w"].dtype), P(pk["sdf_b"]), P(pk["sdf_head"]), P(pts), P(zeros3), P(t), 1, 1, n,
                                  P(out), 1, None, None, None, _lib.stream_handle())
        _lib.check(rc, "nrh_sdf_eval")
        return out

    @torch.no_grad()
    def extract_fields(self, bound_min, bound_max, resolution: int) -> np.ndarray:
        """Dense -sdf grid for marching cubes (models/neus_hint_model.py:68-83); one kernel launch per z-slab."""
        dev = next(self.parameters()).device
        xs = [torch.linspace(float(bound_min[i]), float(bound_max[i]), resolution, device=dev) for i in range(3)]
        u = np.zeros([resolution] * 3, dtype=np.float32)
        slab = max(1, (1 << 22) // (resolution * resolution))
        for x0 in range(0, resolution, slab):
            xx, yy, zz = torch.meshgrid(xs[0][x0:x0 + slab], xs[1], xs[2], indexing="ij")
            pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1)
            u[x0:x0 + slab] = (-self.sdf(pts)).reshape(xx.shape).cpu().numpy()
        return u

    @torch.no_grad()
    def extract_geometry(self, bound_min, bound_max, resolution: int, threshold: float = 0.0):
        """Mesh of the zero level set -> (vertices [V,3] world coordinates, triangles [T,3]) as numpy arrays
        (models/neus_hint_model.py:86-93, 753-758): dense ``-sdf`` grid from the HIP SDF kernel, then PyMCubes if it is
        installed (the reference's triangulation), else the marching-tetrahedra routine of ``nrhints_amd.isosurface``."""
        from .isosurface import extract_surface
        u = self.extract_fields(bound_min, bound_max, resolution)
        vertices, triangles = extract_surface(u, threshold)
        b_min = np.asarray([float(bound_min[i]) for i in range(3)])
        b_max = np.asarray([float(bound_max[i]) for i in range(3)])
        vertices = vertices / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :]
        return vertices, triangles
