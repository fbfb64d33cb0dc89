"""Training step around the hot path: the reference's loss, optimiser / schedule and the
single-node data-parallel gradient exchange, restated for one process per GPU over RCCL.

* loss            L1 colour (sum / (N + 1e-5)) + igr_weight * eikonal on ``relax_inside_sphere`` samples
                  (pipelines/base_pipeline.py:57-69)
* optimiser       Adam over two parameter groups (renderer, ray generator) with the warm-up -> cosine ``LambdaLR``
                  of trainer/trainer.py:99-113
* data parallel   the reference wraps the pipeline in DDP (trainer/trainer.py:88-93): one bucketed mean all-reduce of
                  the 820 923 renderer parameters per step.  ``FlatGradAllReduce`` does the same with ONE collective on
                  one flat buffer (3.3 MB: latency-bound on xGMI, so one call beats per-bucket calls) and is what the
                  training driver uses; plain ``DistributedDataParallel`` also works on the module.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional

import torch
import torch.distributed as dist
from torch import nn


def train_loss_dict(out, rgb_gt: torch.Tensor, igr_weight: float = 0.1) -> Dict[str, torch.Tensor]:
    """``out``: RenderOutput of ``forward(..., is_training=True)``; returns loss / rgb_loss / eikonal_loss / s_val / psnr."""
    n = out.rgb.shape[0]
    rgb_loss = (out.rgb - rgb_gt).abs().sum() / (n + 1e-5)
    grad_err = (torch.linalg.norm(out.analytic_normals, ord=2, dim=-1) - 1.0) ** 2
    mask = out.relax_inside_sphere
    eikonal = (mask * grad_err).sum() / (mask.sum() + 1e-5)
    loss = rgb_loss + eikonal * igr_weight
    with torch.no_grad():
        psnr = 10.0 * torch.log10(1.0 / torch.mean((out.rgb - rgb_gt) ** 2))
    return {"loss": loss, "rgb_loss": rgb_loss, "eikonal_loss": eikonal, "s_val": out.s_val.mean(), "psnr": psnr}


def lr_factor(step: int, warm_up_end: int = 5_000, end_iter: int = 1_000_000, lr_alpha: float = 0.05) -> float:
    """Linear warm-up then cosine decay to ``lr_alpha`` (trainer/trainer.py:103-111)."""
    if step < warm_up_end:
        return step / warm_up_end
    progress = (step - warm_up_end) / (end_iter - warm_up_end)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1.0 - lr_alpha) + lr_alpha


def make_optimizer(renderer: nn.Module, extra_params: Optional[Iterable[nn.Parameter]] = None, lr: float = 5e-4,
                   extra_lr: float = 1e-4, warm_up_end: int = 5_000, end_iter: int = 1_000_000, lr_alpha: float = 0.05,
                   hip: Optional[bool] = None):
    """Adam + LambdaLR as the reference builds them (two groups: renderer, ray-generator deltas).  ``hip``: the optimiser step
    as one HIP launch (adam.HipAdam: the arithmetic of torch's default Adam, state layout of its capturable form); None = whenever the
    renderer lives on the GPU."""
    groups = [{"params": list(renderer.parameters()), "lr": lr}]
    if extra_params is not None:
        # the reference ALWAYS builds the second group, also when the ray generator has no parameters (cam_opt_mode "off"):
        # pass ray_generator.parameters() so that checkpoints load in both directions (trainer/trainer.py:99-102)
        groups.append({"params": list(extra_params), "lr": extra_lr})
    if hip is None:
        hip = all(p.is_cuda and p.dtype == torch.float32 for g in groups for p in g["params"]) and len(groups[0]["params"]) > 0
    if hip:
        from .adam import HipAdam
        opt = HipAdam(groups)
    else:
        opt = torch.optim.Adam(groups)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda s: lr_factor(s, warm_up_end, end_iter, lr_alpha))
    return opt, sched


class FlatGradAllReduce:
    """Mean all-reduce of all gradients through one flat buffer (one RCCL call per step)."""

    def __init__(self, params: Iterable[nn.Parameter], group: Optional[dist.ProcessGroup] = None, always: bool = False):
        self.params: List[nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.always = always      # run the collective at world size 1 as well (tests of the RCCL path on one GPU)
        self._flat: Optional[torch.Tensor] = None          # what reduce() operates on this step
        self._own_flat: Optional[torch.Tensor] = None      # pack scratch of the copy path (never a zero-copy buffer)

    def broadcast_parameters(self, src: int = 0) -> None:
        """Initial parameter sync from rank 0 (what DDP does at wrap time)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        with torch.no_grad():
            flat = torch.cat([p.detach().reshape(-1) for p in self.params])
            dist.broadcast(flat, src=src, group=self.group)
            off = 0
            for p in self.params:
                p.copy_(flat[off: off + p.numel()].view_as(p))
                off += p.numel()

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.always)

    def _zero_copy_flat(self) -> Optional[torch.Tensor]:
        """The gradients ARE one flat buffer already (train_fused._Buffers lays every .grad out as a view of one float32 tensor in
        parameter order and tags the views): return it, else None."""
        flat, off = None, 0
        for p in self.params:
            tag = getattr(p.grad, "_nrh_flat", None) if p.grad is not None else None
            if tag is None or (flat is not None and tag[0] is not flat) or (flat is None and tag[1] != 0) or (flat is not None and tag[1] != off):
                return None
            flat, off = tag[0], tag[1] + p.numel()
        return flat if (flat is not None and off == flat.numel()) else None

    def pack(self) -> None:
        """Gradients -> the flat buffer (graph-capturable: plain device copies; nothing at all when the gradients already live in
        one flat buffer, as the fused training step leaves them)."""
        zc = self._zero_copy_flat()
        self._in_place = zc is not None
        if zc is not None:
            self._flat = zc
            return
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        n = sum(g.numel() for g in grads)
        if self._own_flat is None or self._own_flat.numel() != n or self._own_flat.device != grads[0].device:
            self._own_flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
        self._flat = self._own_flat      # (a zero-copy step's buffer is the live .grad storage of a fused buffer set: never scratch)
        off = 0
        for g in grads:
            self._flat[off: off + g.numel()].copy_(g.reshape(-1))
            off += g.numel()

    def reduce(self) -> None:
        """The one collective of a step (RCCL ring over xGMI; 3.3 MB of fp32 gradients for the default networks).  RCCL averages
        in the collective (ReduceOp.AVG); gloo (CPU tests, one-GPU rehearsal) sums and unpack() divides."""
        self._avg = dist.get_backend(self.group) == "nccl"
        dist.all_reduce(self._flat, op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM, group=self.group)

    def unpack(self) -> None:
        """Flat buffer / world size -> gradients (graph-capturable; in place - at most the division - when pack() found the
        gradients in one flat buffer)."""
        if not getattr(self, "_avg", False):
            self._flat.div_(dist.get_world_size(self.group))
        if getattr(self, "_in_place", False):
            return
        off = 0
        for p in self.params:
            if p.grad is None:
                p.grad = torch.empty_like(p)
            p.grad.copy_(self._flat[off: off + p.numel()].view_as(p))
            off += p.numel()

    def __call__(self) -> None:
        if not self._active():
            return
        self.pack()
        self.reduce()
        self.unpack()


def train_step(renderer, ray_bundle, rgb_gt, background_rgb, global_step: int, optimizer, scheduler=None,
               grad_sync: Optional[FlatGradAllReduce] = None, fused: Optional[bool] = None, sync: bool = True) -> Dict[str, float]:
    """One optimisation step (trainer/trainer.py:269-283): forward (training mode), loss, backward, gradient mean over
    ranks, Adam, schedule.  Returns python floats of the loss dict (one host sync, as the reference's psnr .item()).
    ``fused``: forward + loss + backward as one fixed sequence of HIP launches without autograd (train_fused.py); None = use it
    whenever it applies (renderer parameters only, no ray gradients), False = the autograd path.  ``sync=False``: return the loss
    dict as 0-dim DEVICE tensors instead of python floats - no host synchronisation, so the host can enqueue the next step while
    this one runs (an eager step is ~70 launches: 1.1 ms of a 7.8 ms step at 1 024 rays is the host catching up after the
    read-back, profiles/r03/train_bench_modes.log)."""
    from . import train_fused
    if fused is None:
        fused = train_fused.supported(renderer, ray_bundle) is None
    if (not sync and fused) or renderer.dyn_scalars is not None:
        # 1/s and the cos-anneal ratio on the device (as in a captured step): the re-pack after every optimiser step then needs no
        # host read of 1/s.  Evaluation renders in between read 1/s back themselves (renderer._host_inv_s); release_device_scalars()
        # returns to host-side scalars.  (Once the device scalars exist the kernels read THEM, so this step's ratio goes there
        # whichever way the step was asked for.)
        if renderer.dyn_scalars is None:
            renderer.dyn_scalars = torch.zeros(2, dtype=torch.float32, device=ray_bundle.origins.device)
            renderer._packed_key = None
        cfg = renderer.config
        renderer.dyn_scalars[1:2].fill_(min(1.0, global_step / cfg.anneal_end) if cfg.anneal_end > 0 else 1.0)
    if fused:
        optimizer.zero_grad(set_to_none=True)
        loss8 = train_fused.train_step_backward(renderer, ray_bundle, rgb_gt, background_rgb, global_step)
        keys, vec = list(train_fused.LOSS_KEYS), loss8[:5]
    else:
        out = renderer(ray_bundle, is_training=True, background_rgb=background_rgb, global_step=global_step)
        losses = train_loss_dict(out, rgb_gt, renderer.config.igr_weight)
        optimizer.zero_grad(set_to_none=True)
        losses["loss"].backward()
        keys = list(losses)
        vec = torch.stack([losses[k].detach().float().reshape(()) for k in keys])
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    if not sync:
        vec = vec.clone()                     # the fused step's loss vector is a persistent buffer
        return {k: vec[i] for i, k in enumerate(keys)}
    return dict(zip(keys, vec.tolist()))      # one device-to-host copy


def release_device_scalars(renderer) -> None:
    """Undo what ``train_step(..., sync=False)`` switched on: 1/s and the cos-anneal ratio back on the host.  Raises the f16x3
    weight-range ``ValueError`` here if the sync-free steps drove a weight out of range (the guard is asynchronous in that mode)."""
    try:
        renderer.check_weight_range()
    finally:
        # also when the guard raises: the renderer must not stay in device-scalar mode with a NaN host 1/s (the caller may
        # switch to precision "f32" and go on)
        renderer.dyn_scalars = None
        renderer._packed_key = None
        renderer._generation = getattr(renderer, "_generation", 0) + 1


class GraphedTrainStep:
    """The whole optimisation step - forward, loss, backward, Adam - captured ONCE into a hipGraph and replayed: one graph
    launch per step instead of ~230 kernel launches issued from Python, which is what bounds small batches (the reference's
    default is 512 rays per step, 64 per rank under 8-way DDP).  With a gradient exchange (``grad_sync`` and more than one
    rank) the step is two graphs around one eager RCCL all-reduce of the flat gradient buffer.

    What changes between steps is read from device memory at run time: the batch (static ray / pixel buffers that
    ``__call__`` copies into), 1/s and the cos-anneal ratio (``renderer.dyn_scalars``, see NrhNet.dyn_scalars) and the
    learning rates (tensor-valued ``lr`` of adam.HipAdam).  The jitter comes from the graph-safe device generator.
    Frozen at capture time: the batch size, the precision / model configuration, whether the geometry warm-up is active
    (re-create the object when ``global_step`` crosses ``geometry_warmup_end``).

        step = GraphedTrainStep(renderer, batch_rays=512, background_rgb=bg, lr=5e-4)
        for it in range(...): losses = step(ray_bundle, rgb_gt, global_step=it)     # dict of python floats

    With ``ray_generator`` the graph starts one stage earlier, at the data loader's ``RawPixelBundle`` (view index, pixel, pose,
    light): ray generation and - when pose / light refinement is on - its adjoint are captured too, and Adam gets the reference's
    SECOND parameter group (the ray generator's deltas at ``ray_lr``, trainer/trainer.py:99-102; both groups follow the same
    warm-up / cosine factor, as ``LambdaLR`` scales every group).  The group exists even when the ray generator has no
    parameters, so optimiser checkpoints have the reference's two-group layout either way.

        step = GraphedTrainStep(renderer, 512, bg, ray_generator=rg, ray_lr=rg.config.opt_lr)
        for it in range(...): losses = step(pixel_bundle, pixel_bundle.rgb_gt, global_step=it)
    """

    def __init__(self, renderer, batch_rays: int, background_rgb: torch.Tensor, lr: float = 5e-4, warm_up_end: int = 5_000,
                 end_iter: int = 1_000_000, lr_alpha: float = 0.05, global_step: int = 0,
                 grad_sync: Optional["FlatGradAllReduce"] = None, warmup_steps: int = 3,
                 optimizer_state: Optional[Dict] = None, jitter: Optional[tuple] = None, fused: Optional[bool] = None,
                 ray_generator: Optional[nn.Module] = None, ray_lr: float = 1e-4, collective_in_graph: bool = False):
        """``collective_in_graph`` (with a gradient exchange): capture the flat all-reduce INSIDE the one hipGraph of the step -
        one launch per step instead of graph | eager all-reduce | graph.  Opt-in: a captured collective ties the replay to the
        communicator's lifetime and needs the backend's capture support (RCCL: yes, tested with one rank; gloo stages through
        the host and cannot be captured); the two-graph form is the default because it is correct by construction.
        ``jitter``: optional static buffers ``(t_rand_primary [n,1], t_rand_shadow [n,64])`` read by every replay instead
        of the device generator's draws (reproducible runs; the parity test against the eager step overwrites them)."""
        dev = next(renderer.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs the renderer on the GPU")
        self.renderer, self.grad_sync = renderer, grad_sync
        self.fused = fused            # None: the autograd-free step of train_fused.py whenever it applies; False: autograd
        self.jitter = None if jitter is None else tuple(t.detach().to(dev, torch.float32).contiguous().clone() for t in jitter)
        self.sched_args = (warm_up_end, end_iter, lr_alpha)
        self.base_lr = lr
        self.lr_t = torch.tensor(lr, dtype=torch.float32, device=dev)
        self.ray_generator, self.base_ray_lr = ray_generator, ray_lr
        self.ray_lr_t = torch.tensor(ray_lr, dtype=torch.float32, device=dev)
        groups = [{"params": list(renderer.parameters()), "lr": self.lr_t}]
        if ray_generator is not None:
            groups.append({"params": list(ray_generator.parameters()), "lr": self.ray_lr_t})
        n = batch_rays
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        from .containers import RawPixelBundle, RayBundle
        self.rays = RayBundle(origins=z(n, 3), directions=z(n, 3), pl_positions=z(n, 3), nears=z(n, 1), fars=z(n, 1))
        self.pixels = None
        if ray_generator is not None:
            self.pixels = RawPixelBundle(img_indices=torch.zeros(n, 1, dtype=torch.int64, device=dev), h_indices=z(n, 1),
                                         w_indices=z(n, 1), poses=z(n, 4, 4), pls=z(n, 3), rgb_gt=None)
        self.gt = z(n, 3)
        # Adam as one launch (adam.HipAdam: the arithmetic of torch's default Adam with device-side learning rates).  The fused
        # step's gradients live at fixed addresses; the autograd path's are staged through the optimiser's own buffers
        from . import train_fused
        from .adam import HipAdam
        # (pose / light refinement no longer leaves the fused step: train_fused computes the ray adjoints, nrh_ray_adjoint)
        self._use_fused = (train_fused.supported(renderer, self.rays) is None) if fused is None else bool(fused)
        self.optimizer = HipAdam(groups)
        self.bg = background_rgb.detach().to(dev, torch.float32).reshape(1, 3).clone()
        renderer.dyn_scalars = torch.zeros(2, dtype=torch.float32, device=dev)
        self._capture_step = global_step
        self._set_host_scalars(global_step)
        # sane static inputs for the warm-up / capture passes: a ring of rays looking at the origin
        ang = torch.linspace(0.0, 6.2832, n, device=dev)
        o = torch.stack([3.0 * torch.cos(ang), 3.0 * torch.sin(ang), torch.full_like(ang, 0.5)], dim=1)
        d = torch.nn.functional.normalize(-o, dim=1)
        self.rays.origins.copy_(o); self.rays.directions.copy_(d); self.rays.pl_positions.copy_(o * 1.3)
        mid = -(o * d).sum(1, keepdim=True)
        self.rays.nears.copy_(mid - 1.0); self.rays.fars.copy_(mid + 1.0)
        self.gt.fill_(0.5)
        if self.pixels is not None:
            # the same ring as pixels: the principal point of cameras that look at the origin (camera looks along its -z axis)
            cam = ray_generator.camera
            up0 = torch.tensor([0.0, 0.0, 1.0], device=dev).expand_as(d)
            right = torch.nn.functional.normalize(torch.linalg.cross(d, up0), dim=1)
            up = torch.linalg.cross(right, d)
            self.pixels.poses[:, :3, 0], self.pixels.poses[:, :3, 1], self.pixels.poses[:, :3, 2] = right, up, -d
            self.pixels.poses[:, :3, 3] = o
            self.pixels.poses[:, 3, 3] = 1.0
            self.pixels.h_indices.fill_(float(cam.cy) - 0.5); self.pixels.w_indices.fill_(float(cam.cx) - 0.5)
            self.pixels.pls.copy_(o * 1.3)
        self._keys: List[str] = []
        # the workspace pointer is baked into the graph: make it large enough for any later evaluation render as well, so
        # that the renderer never replaces (frees) it while the graph is alive
        renderer._workspace(dev, max(n, int(renderer.max_eval_rays_while_graphed)))
        # Warm-up passes build every cache (pack plans, constants, Adam state tensors) eagerly.  They run real optimiser
        # steps on synthetic rays, so the parameters are put back and the Adam state is zeroed afterwards: capturing a
        # step must not change the model or what a resumed optimiser remembers.
        params = list(renderer.parameters()) + (list(ray_generator.parameters()) if ray_generator is not None else [])
        keep = [p.detach().clone() for p in params]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup_steps)):
                self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for p, k in zip(params, keep):
                p.copy_(k)
            for st in self.optimizer.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
        if optimizer_state is not None:                 # resume: moments and step counts of a checkpointed Adam
            self.load_optimizer_state(optimizer_state)
        renderer._generation = getattr(renderer, "_generation", 0) + 1
        # the fused step's buffer set of this batch size is part of the graph from here on: keep train_fused from evicting it
        self._pin_key = (str(dev), n, bool(getattr(renderer, "_hints", False)))
        if self._use_fused:
            renderer.__dict__.setdefault("_fused_pinned", set()).add(self._pin_key)
        self.graph, self.graph_tail = torch.cuda.CUDAGraph(), None
        self.optimizer.zero_grad(set_to_none=True)
        # With a process group alive its watchdog thread polls events while we capture: legal only in thread-local capture mode
        # (in the default global mode hipEventQuery from ANY thread invalidates the capture / kills the watchdog - measured).
        mode = dict(capture_error_mode="thread_local") if (dist.is_available() and dist.is_initialized()) else {}
        torch.cuda.synchronize(dev)
        # a weight-range verdict the warm-up packs enqueued is read (and raised) NOW: no event query may happen inside the capture
        renderer._range_guard_poll(wait=True)
        self.collective_in_graph = bool(collective_in_graph and self._sync_active())
        if self._sync_active() and not self.collective_in_graph:
            # The collective stays OUTSIDE the graphs: graph 1 = forward + loss + backward + gradient flattening, one eager
            # all-reduce on the same stream, graph 2 = unflatten + Adam - three launches per step instead of one.  (A captured
            # single-rank RCCL all-reduce does replay on this stack, profiles/r02/rccl_graph_probe.log, but a collective inside
            # a graph ties the replay to the communicator's lifetime and to RCCL's capture support per version; eager is the
            # form that is correct by construction for the 8-GPU runs that cannot be rehearsed here.)
            with torch.cuda.graph(self.graph, **mode):
                self._loss_vec = self._body(upto="pack")
            self.grad_sync.reduce()
            self.graph_tail = torch.cuda.CUDAGraph()
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(self.graph_tail, pool=self.graph.pool(), **mode):
                self._body(upto="tail")
        else:
            with torch.cuda.graph(self.graph, **mode):
                self._loss_vec = self._body()

    def _set_host_scalars(self, global_step: int) -> None:
        cfg = self.renderer.config
        cos = min(1.0, global_step / cfg.anneal_end) if cfg.anneal_end > 0 else 1.0
        # one launch for the three (nrh_step_scalars; three fill kernels before)
        from . import _lib
        with torch.cuda.device(self.lr_t.device):
            _lib.step_scalars([(self.renderer.dyn_scalars[1:2], cos), (self.lr_t, self.base_lr * lr_factor(global_step, *self.sched_args)),
                               (self.ray_lr_t, self.base_ray_lr * lr_factor(global_step, *self.sched_args))])

    def _sync_active(self) -> bool:
        return self.grad_sync is not None and self.grad_sync._active()

    def _body(self, upto: str = "all") -> Optional[torch.Tensor]:
        """One step; ``upto``: "all" (eager warm-up / single graph), "pack" (graph 1 of the split form: through gradient
        flattening), "tail" (graph 2: unflatten + Adam)."""
        sync = self._sync_active()
        vec = None
        from . import train_fused
        rays, rg_live = self.rays, []
        if upto != "tail" and self.ray_generator is not None:
            # ray generation is part of the step: on aliases of the deltas for the same reason as the renderer's parameters below
            rg_named = list(self.ray_generator.named_parameters())
            rg_alias = {n: p.detach().requires_grad_(p.requires_grad) for n, p in rg_named}
            rays = torch.func.functional_call(self.ray_generator, rg_alias, args=(self.pixels,))
            rg_live = [(p, rg_alias[n]) for n, p in rg_named if p.requires_grad]
        use_fused = self._use_fused
        if upto != "tail" and use_fused:
            rgr = {} if rg_live else None
            loss8 = train_fused.train_step_backward(
                self.renderer, rays, self.gt, self.bg, self._capture_step,
                t_rand_primary=None if self.jitter is None else self.jitter[0], t_rand_shadow=None if self.jitter is None else self.jitter[1],
                ray_grads=rgr)
            if rg_live:
                # the ray adjoints through the ray generator's backward (nrh_generate_rays_indexed_backward + the exponential map)
                outs = [(t_, rgr[k]) for k, t_ in (("origins", rays.origins), ("directions", rays.directions),
                                                   ("pl_positions", rays.pl_positions)) if t_.requires_grad]
                gs = torch.autograd.grad([t_ for t_, _ in outs], [a for _, a in rg_live], grad_outputs=[g for _, g in outs], allow_unused=True)
                for (p, _), g in zip(rg_live, gs):
                    p.grad = g
            self._keys = list(train_fused.LOSS_KEYS)
            vec = loss8[:5]
            if sync:
                self.grad_sync.pack()
            if upto == "pack":
                return vec
        elif upto != "tail":
            jit = {} if self.jitter is None else dict(_t_rand_primary=self.jitter[0], _t_rand_shadow=self.jitter[1])
            # The forward runs on fresh leaf ALIASES of the parameters (same storage) and the gradients come from
            # torch.autograd.grad.  A parameter's AccumulateGrad node is cached together with the stream it was first used on;
            # a model that trained eagerly on the default stream before (and whose old autograd graph is still referenced
            # somewhere) would make the engine synchronise the capture stream with the default stream - measured: a segfault
            # at capture end when an RCCL process group is alive (profiles/r02/rccl_graph_probe_before_fix.log).
            named = [(n, p) for n, p in self.renderer.named_parameters()]
            alias = {n: p.detach().requires_grad_(p.requires_grad) for n, p in named}
            out = torch.func.functional_call(self.renderer, alias, args=(rays,),
                                             kwargs=dict(is_training=True, background_rgb=self.bg, global_step=self._capture_step, **jit))
            losses = train_loss_dict(out, self.gt, self.renderer.config.igr_weight)
            live = [(p, alias[n]) for n, p in named if p.requires_grad] + rg_live
            for (p, _), g in zip(live, torch.autograd.grad(losses["loss"], [a for _, a in live], allow_unused=True)):
                p.grad = g
            self._keys = list(losses)
            vec = torch.stack([losses[k].detach().float().reshape(()) for k in self._keys])
            if sync:
                self.grad_sync.pack()
            if upto == "pack":
                return vec
        if sync:
            if upto == "all":
                self.grad_sync.reduce()
            self.grad_sync.unpack()
        self.optimizer.step()
        return vec

    def __call__(self, ray_bundle, rgb_gt: torch.Tensor, global_step: int) -> Dict[str, float]:
        """``ray_bundle``: a RayBundle, or - for a step built with ``ray_generator`` - the RawPixelBundle of the batch."""
        n = self.gt.shape[0]
        cfg = self.renderer.config
        if (global_step < cfg.geometry_warmup_end) != (self._capture_step < cfg.geometry_warmup_end):
            raise RuntimeError("geometry warm-up state changed since capture: create a new GraphedTrainStep")
        if self.pixels is not None:
            pb = ray_bundle
            if not hasattr(pb, "h_indices"):
                raise TypeError("this step was built with a ray generator: pass the batch's RawPixelBundle")
            if pb.h_indices.shape[0] != n:
                raise ValueError(f"this graph was captured for {n} rays per step, got {pb.h_indices.shape[0]}")
            if pb.img_indices is None:
                raise ValueError("training pixels carry their view index (img_indices)")
            pairs = ((self.pixels.img_indices, pb.img_indices), (self.pixels.h_indices, pb.h_indices),
                     (self.pixels.w_indices, pb.w_indices), (self.pixels.poses, pb.poses), (self.pixels.pls, pb.pls),
                     (self.gt, rgb_gt))
        else:
            if ray_bundle.origins.shape[0] != n:
                raise ValueError(f"this graph was captured for {n} rays per step, got {ray_bundle.origins.shape[0]}")
            pairs = ((self.rays.origins, ray_bundle.origins), (self.rays.directions, ray_bundle.directions),
                     (self.rays.pl_positions, ray_bundle.pl_positions), (self.rays.nears, ray_bundle.nears),
                     (self.rays.fars, ray_bundle.fars), (self.gt, rgb_gt))
        srcs = [src.reshape(dst.shape) for dst, src in pairs]
        if all(s_.is_cuda and s_.device == d_.device and s_.dtype == d_.dtype == torch.float32 for (d_, _), s_ in zip(pairs, srcs)):
            torch._foreach_copy_([d_ for d_, _ in pairs], srcs)          # one launch for the batch's six tensors
        else:
            for (dst, _), src in zip(pairs, srcs):
                dst.copy_(src, non_blocking=True)
        self._set_host_scalars(global_step)
        self.graph.replay()
        if self.graph_tail is not None:
            self.grad_sync.reduce()
            self.graph_tail.replay()
        # the replay updated the parameters in place without touching their version counters: tell the renderer, so that
        # an evaluation render between replays re-packs instead of reusing a stale pack
        self.renderer._generation = getattr(self.renderer, "_generation", 0) + 1
        out = dict(zip(self._keys, self._loss_vec.tolist()))
        # the re-pack runs inside the graph, where nothing can raise: the f16x3 weight-range guard of the eager pack
        # (renderer.packed_params) is applied to the replayed pack every `range_check_every` steps (the read-back above already
        # synchronised, so this costs one small reduction + copy)
        self._replays = getattr(self, "_replays", 0) + 1
        if self._replays % max(1, int(self.renderer.range_check_every)) == 0:
            self.renderer.check_weight_range()
        return out

    def load_optimizer_state(self, state: Dict) -> None:
        """Copy the per-parameter Adam state of ``state`` (an ``optimizer.state_dict()`` of the same parameter order) into
        the captured optimiser's tensors IN PLACE (their addresses are part of the graph)."""
        mine = self.optimizer.state_dict()["state"]
        with torch.no_grad():
            for k, st in state["state"].items():
                for name, v in st.items():
                    if torch.is_tensor(v) and k in mine and name in mine[k]:
                        mine[k][name].copy_(v.to(mine[k][name].device, mine[k][name].dtype).reshape(mine[k][name].shape))

    def release(self) -> None:
        """Back to eager operation (drops the graph and the device-side scalars)."""
        self.graph = self.graph_tail = None
        self.renderer.__dict__.setdefault("_fused_pinned", set()).discard(getattr(self, "_pin_key", None))
        self.renderer.dyn_scalars = None
        # the cached pack was made while 1/s lived on the device (its host copy is NaN): packs of the eager mode must not
        # reuse it - a render after release() would otherwise run with inv_s = NaN and no device scalars
        self.renderer._packed_key = None
        self.renderer._generation = getattr(self.renderer, "_generation", 0) + 1


# ---- checkpoints in the reference's layout (trainer/trainer.py:149-158, 173-236) ---------------------------------------
def register_view(renderer, ray_generator, img_pixel_bundle, device, steps: int = 500, batch_size: int = 512,
                  white_background: bool = True, lr: Optional[float] = None, generator: Optional[torch.Generator] = None,
                  log=None, fused: Optional[bool] = None) -> List[float]:
    """Fit the ray generator's pose / light deltas of one view to its pixels with the renderer frozen in evaluation mode
    (pipelines/base_pipeline.py:71-91): ``steps`` Adam steps over random ``batch_size``-pixel batches of the [H,W]
    ``img_pixel_bundle`` with the summed L1 loss / (N + 1e-5).  The gradient reaches the deltas through the renderer's ray
    gradients (origins / directions / pl_positions).  Returns the loss trajectory.
    ``fused`` (None = whenever it applies): every step is the autograd-free sequence of train_fused.train_step_backward with the
    renderer frozen - evaluation-mode forward, loss, the sweeps, nrh_ray_adjoint, the ray generator's backward, HipAdam on its
    deltas - without the weight-gradient launches the reference computes and discards here, and with ONE read-back of the loss
    trajectory at the end (per step only when ``log`` is given, as upstream's print does)."""
    lr = ray_generator.config.opt_lr if lr is None else lr
    H, W = img_pixel_bundle.shape[0], img_pixel_bundle.shape[1]
    bg = torch.full((1, 3), 1.0 if white_background else 0.0, device=device)
    losses: List[float] = []
    from . import train_fused
    rg_params = [p for p in ray_generator.parameters() if p.requires_grad]
    if fused is not False and rg_params and torch.device(device).type == "cuda" and batch_size <= renderer.max_fused_train_rays:
        from types import SimpleNamespace
        z3 = torch.zeros(batch_size, 3, device=device, requires_grad=True)
        probe = SimpleNamespace(origins=z3, directions=z3, pl_positions=z3)
        frozen = [(p, p.requires_grad) for p in renderer.parameters()]
        for p, _ in frozen:
            p.requires_grad_(False)        # the reference computes and discards the renderer's gradients here; only the deltas step
        try:
            if train_fused.supported(renderer, probe) is None:
                from .adam import HipAdam
                optimizer = HipAdam(rg_params, lr=lr)
                dev_losses = []
                # the view's pixels move to the device ONCE (27 floats per pixel); a batch is then a gather on the GPU.  The pixel
                # indices keep upstream's host-side draws (same generator, same order: h then w)
                img_dev = img_pixel_bundle.to(device)
                with torch.enable_grad():
                    for i in range(steps):
                        hi = torch.randint(0, H, (batch_size,), device="cpu", generator=generator)
                        wi = torch.randint(0, W, (batch_size,), device="cpu", generator=generator)
                        pb = img_dev[hi.to(device, non_blocking=True), wi.to(device, non_blocking=True)]
                        optimizer.zero_grad(set_to_none=True)
                        # evaluation-mode forward + L1 loss / (N + 1e-5) + the ray adjoints as one fixed sequence of HIP launches
                        # (no weight gradients, no autograd inside the renderer); its backward continues into the ray generator
                        loss8 = train_fused.train_step_backward(renderer, ray_generator(pb), pb.rgb_gt, bg, 0, igr_weight=0.0, is_training=False)
                        optimizer.step()
                        dev_losses.append(loss8[1:2].clone())
                        if log is not None:
                            log(f"register step: {i} loss: {float(dev_losses[-1])}")
                # one read-back for the whole trajectory (upstream prints loss.item() per step: pass `log` to get that behaviour)
                return torch.cat(dev_losses).tolist() if dev_losses else []
        finally:
            for p, rq in frozen:
                p.requires_grad_(rq)
    optimizer = torch.optim.Adam(ray_generator.parameters(), lr=lr)
    with torch.enable_grad():
        for i in range(steps):
            hi = torch.randint(0, H, (batch_size,), device="cpu", generator=generator)
            wi = torch.randint(0, W, (batch_size,), device="cpu", generator=generator)
            pb = img_pixel_bundle[hi, wi].to(device)
            out = renderer(ray_generator(pb), background_rgb=bg, is_training=False)
            loss = torch.nn.functional.l1_loss(out.rgb, pb.rgb_gt, reduction="sum") / (out.rgb.size(0) + 1e-5)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            losses.append(loss.item())
            if log is not None:
                log(f"register step: {i} loss: {losses[-1]}")
    return losses


def checkpoint_state(renderer: nn.Module, optimizer, scheduler, global_step: int, world_size: int = 1,
                     extra_pipeline_state: Optional[Dict[str, torch.Tensor]] = None) -> Dict:
    """``model_states`` dict as the reference pickles it: the renderer's tensors live under the ``renderer.`` prefix of
    the pipeline state dict (``ray_generator.*`` entries can be passed through ``extra_pipeline_state``)."""
    pipeline = {"renderer." + k: v for k, v in renderer.state_dict().items()}
    if extra_pipeline_state:
        pipeline.update(extra_pipeline_state)
    return {"world_size": world_size, "global_step": global_step, "pipeline": pipeline,
            "optimizer": optimizer.state_dict() if optimizer is not None else None,
            "scheduler": scheduler.state_dict() if scheduler is not None else None}


def save_checkpoint(path: str, renderer, optimizer, scheduler, global_step: int, world_size: int = 1, **kw) -> None:
    torch.save(checkpoint_state(renderer, optimizer, scheduler, global_step, world_size, **kw), path)


def load_checkpoint(path_or_state, renderer, optimizer=None, scheduler=None, map_location="cpu") -> int:
    """Load a checkpoint written by ``save_checkpoint`` OR by the reference trainer (released ``*_step_1000000.ckpt``):
    picks the ``renderer.*`` entries of ``model_states['pipeline']``.  Returns the stored global step."""
    st = torch.load(path_or_state, map_location=map_location, weights_only=False) if isinstance(path_or_state, str) \
        else path_or_state
    pipe = st["pipeline"]
    sd = {k[len("renderer."):]: v for k, v in pipe.items() if k.startswith("renderer.")}
    renderer.load_state_dict(sd)
    if optimizer is not None and st.get("optimizer") is not None:
        optimizer.load_state_dict(st["optimizer"])
    if scheduler is not None and st.get("scheduler") is not None:
        scheduler.load_state_dict(st["scheduler"])
    return int(st.get("global_step", 0))
