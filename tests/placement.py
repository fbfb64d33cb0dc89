"""Test helper: the float64 oracle's training step ON A GIVEN SAMPLE PLACEMENT (VERDICT r5 "next round" item 1).

The reference keeps three products of its forward outside the autograd graph: the hierarchical sample positions
(models/neus_hint_model.py:696-713), the visibility hint (:379) and the specular cue (:589).  They are decided by float32
arithmetic at fp32 noise - one sample that falls the other side of a section boundary moves single gradients by 1e-4..1e-3 of
their scale - so an end-to-end gradient comparison mixes "where did the samplers put their samples" with "is the backward's
arithmetic right".  This helper takes the HIP path's own placement (z, visibility, cue as its no-grad kernels produced them) and
differentiates the oracle in float64 at exactly that placement: what is left in the comparison is the arithmetic of the
differentiable part alone, and the HIP gradients are held to the reference's UNWIDENED per-step bound against it.

Memory: the float64 double-backward graph of 131 072 points is ~30 GB; the step is therefore accumulated over chunks of rays (the
loss is a sum over rays once the two normalisers - the ray count and the number of samples inside the unit sphere, both
constants of the graph - are fixed up front; pipelines/base_pipeline.py:57-62)."""
import numpy as np
import torch

from oracle import neus_oracle as orc

T = torch.from_numpy


def oracle_step_at_placement(state, rays, rgb_gt, global_step, z, vis, cue, t_rand_primary, t_rand_shadow, igr_weight=0.1,
                             chunk=128, ray_grads=True, net_values=None, sections=None, device=None, **oracle_kw):
    """state: numpy state dict (reference key names); rays: dict o, d, pl, near, far (numpy float32); z [N,128] float64 tensor
    (section START positions: mid - dist / 2 of the HIP forward), vis [N,1], cue [N,4] float64 tensors.
    ``sections`` (mid [N,128], dists [N,128]; float64 tensors): the HIP forward's section mid-points and lengths as they are
    (render_forward sections_override) - lengths re-derived from z would carry the float32 rounding of positions ~3 into sections
    of 1e-5.  ``net_values`` dict(sdf [N*128,1], grad [N*128,3], feat [N*128,256]; float64 CPU tensors, any subset): the HIP forward's own
    network outputs at the composite samples; the oracle then differentiates AT those values (render_forward net_override).
    ``device``: where the restatement's float64 tensor program runs.  None = the CPU (what pins it: tests/test_oracle_golden.py).  The
    1 024-ray GPU tests pass "cuda": the same program in float64 on the device - 1 TFLOP of double-backward per step, 50 s on the
    box's host cores and 3 s there - after tests/test_gpu_train1024.py::test_oracle_on_device_equals_oracle_on_host has checked that
    the two agree to float64 round-off.
    Returns (losses dict, {parameter name: float64 gradient (numpy)}, {origins / directions / pl_positions: gradient}, rgb)."""
    n = rays["o"].shape[0]
    dev = torch.device("cpu" if device is None else device)
    leaves = {k: T(np.asarray(v)).double().to(dev).clone().requires_grad_(True) for k, v in state.items()}
    r64 = {k: T(np.asarray(rays[k])).double().to(dev) for k in ("o", "d", "pl", "near", "far")}
    ray_leaves = {k: r64[k].clone().requires_grad_(ray_grads) for k in ("o", "d", "pl")}
    gt = T(np.asarray(rgb_gt)).double().to(dev)
    z, vis, cue = z.to(dev), vis.to(dev), cue.to(dev)
    net_values = None if net_values is None else {k: v.to(dev) for k, v in net_values.items()}
    sections = None if sections is None else tuple(t.to(dev) for t in sections)
    tp64, ts64 = T(np.asarray(t_rand_primary)).double().to(dev), T(np.asarray(t_rand_shadow)).double().to(dev)
    # the eikonal normaliser: samples inside the unit sphere (models/neus_hint_model.py:512-514), a constant of the graph
    with torch.no_grad():
        sample_dist = 2.0 / 64
        dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), sample_dist, dtype=torch.float64, device=dev)], dim=-1)
        mid = z + dists * 0.5
        if sections is not None:
            mid = sections[0]
        pts = r64["o"][:, None, :] + r64["d"][:, None, :] * mid[..., None]
        m_total = float((torch.linalg.norm(pts, dim=-1) < 1.0).double().sum())
    rgb_sum = eik_sum = 0.0
    rgbs = []
    for i in range(0, n, chunk):
        sl = slice(i, min(n, i + chunk))
        params = orc.params_from_state(leaves, torch.float64)      # (the weight-norm fold is part of each chunk's graph)
        out = orc.render_forward(params, ray_leaves["o"][sl], ray_leaves["d"][sl], ray_leaves["pl"][sl], r64["near"][sl], r64["far"][sl],
                                 background_rgb=torch.ones(1, 3, dtype=torch.float64, device=dev), is_training=True, global_step=global_step,
                                 t_rand_primary=tp64[sl], t_rand_shadow=ts64[sl],
                                 mode="as_written", differentiable=True, z_override=z[sl], vis_override=vis[sl], cue_override=cue[sl],
                                 net_override=None if net_values is None else {k: v[sl.start * 128: sl.stop * 128] for k, v in net_values.items()},
                                 sections_override=None if sections is None else (sections[0][sl], sections[1][sl]),
                                 **oracle_kw)
        rgb_part = (out["rgb"] - gt[sl]).abs().sum() / (n + 1e-5)
        ge = (torch.linalg.norm(out["analytic_normals"], dim=-1) - 1.0) ** 2
        eik_part = (out["relax_inside_sphere"] * ge).sum() / (m_total + 1e-5)
        (rgb_part + igr_weight * eik_part).backward()
        rgb_sum += float(rgb_part.detach())
        eik_sum += float(eik_part.detach())
        rgbs.append(out["rgb"].detach())
    losses = dict(loss=rgb_sum + igr_weight * eik_sum, rgb_loss=rgb_sum, eikonal_loss=eik_sum)
    pgrads = {k: v.grad.cpu().numpy() for k, v in leaves.items() if v.grad is not None}
    rgrads = {}
    if ray_grads:
        rgrads = dict(origins=ray_leaves["o"].grad.cpu().numpy(), directions=ray_leaves["d"].grad.cpu().numpy(),
                      pl_positions=ray_leaves["pl"].grad.cpu().numpy())
    return losses, pgrads, rgrads, torch.cat(rgbs).cpu().numpy()


def hip_placement(forward_out):
    """The non-differentiable products of a fused training step's forward, from the dict ``train_fused.train_step_backward(...,
    forward_out=...)`` filled: (z [N,128], vis [N,1], cue [N,4]) as float64 CPU tensors, plus the SDF network's outputs at those
    samples (dict sdf / grad / feat: the forward VALUES the HIP backward linearises at) and the sections (mid, dists) as the kernels
    hold them.  Taken from the step ITSELF: a separate forward call may place samples differently (see train_step_backward)."""
    f = forward_out
    mid, dist = f["mid_z"].double().cpu(), f["dists"].double().cpu()
    net = dict(sdf=f["sdf"].double().cpu(), grad=f["normals"].reshape(-1, 3).double().cpu(), feat=f["feat"].double().cpu())
    return mid - 0.5 * dist, f["visibilities"].double().cpu().reshape(-1, 1), f["cue"][:, 0, :].double().cpu(), net, (mid, dist)
