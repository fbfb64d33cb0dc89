"""Host-side training logic on CPU: loss formula vs the recorded reference loss, LR schedule, and the flat-gradient
all-reduce over gloo with world size 2 (the renderer itself needs a GPU; a small stand-in module carries the grads)."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

import nrhints_amd as na
from nrhints_amd.training import FlatGradAllReduce, lr_factor, make_optimizer, train_loss_dict
from tests.conftest import load_npz


def test_loss_matches_reference_record():
    g = load_npz("train_b.npz")
    T = torch.from_numpy
    n = g["rgb"].shape[0]
    out = na.RenderOutput(rgb=T(g["rgb"]), depth=T(g["depth"]), weights=T(g["weights"]), s_val=torch.ones(n, 128),
                          inside_sphere=T(g["inside_sphere"]), relax_inside_sphere=T(g["inside_sphere"]),
                          analytic_normals=T(g["analytic_normals"]), normalized_analytic_normals=T(g["analytic_normals"]))
    d = train_loss_dict(out, T(g["rgb_gt"]))
    np.testing.assert_allclose(d["loss"].item(), g["loss"], rtol=1e-6)
    np.testing.assert_allclose(d["rgb_loss"].item(), g["rgb_loss"], rtol=1e-6)
    np.testing.assert_allclose(d["eikonal_loss"].item(), g["eikonal_loss"], rtol=1e-6)


def test_hip_adam_loads_a_default_adam_checkpoint_layout():
    """a reference checkpoint holds the state of torch's DEFAULT Adam - ``step`` a CPU tensor, groups with
    ``capturable: False`` (trainer/trainer.py:155, 222).  HipAdam.load_state_dict must turn that into the layout its kernel
    reads (float32 0-dim ``step`` on the parameter's device, capturable groups) - checked here on CPU tensors (no step)."""
    from nrhints_amd.adam import HipAdam
    torch.manual_seed(0)
    pa = [nn.Parameter(torch.randn(4, 3)), nn.Parameter(torch.randn(7))]
    ref = torch.optim.Adam([{"params": pa[:1], "lr": 5e-4}, {"params": pa[1:], "lr": 1e-4}])
    for _ in range(3):
        for p in pa:
            p.grad = torch.randn_like(p)
        ref.step()
    sd = ref.state_dict()
    assert sd["param_groups"][0]["capturable"] is False
    pb = [nn.Parameter(p.detach().clone()) for p in pa]
    hip = HipAdam([{"params": pb[:1], "lr": 5e-4}, {"params": pb[1:], "lr": 1e-4}])
    hip.load_state_dict(sd)
    for g in hip.param_groups:
        assert g["capturable"] is True and not g["foreach"]
    for p, q in zip(pb, pa):
        st = hip.state[p]
        assert st["step"].dtype == torch.float32 and st["step"].dim() == 0 and st["step"].device == p.device
        assert float(st["step"]) == 3.0
        np.testing.assert_array_equal(st["exp_avg"].numpy(), ref.state[q]["exp_avg"].numpy())
        np.testing.assert_array_equal(st["exp_avg_sq"].numpy(), ref.state[q]["exp_avg_sq"].numpy())
    # and the round trip back into torch's Adam (resume in the other direction)
    back = torch.optim.Adam([{"params": pa[:1], "lr": 5e-4}, {"params": pa[1:], "lr": 1e-4}])
    back.load_state_dict(hip.state_dict())


def test_lr_schedule():
    assert lr_factor(0) == 0.0 and abs(lr_factor(2500) - 0.5) < 1e-12 and abs(lr_factor(5000) - 1.0) < 1e-12
    assert abs(lr_factor(1_000_000) - 0.05) < 1e-12
    mid = (5000 + 1_000_000) // 2
    assert abs(lr_factor(mid) - (0.5 * 0.95 + 0.05)) < 1e-5
    m = na.NeuSHintRenderer()
    extra = [nn.Parameter(torch.zeros(10, 6))]
    opt, sched = make_optimizer(m, extra)
    assert len(opt.param_groups) == 2 and len(opt.param_groups[0]["params"]) == 46
    assert opt.param_groups[0]["initial_lr"] == 5e-4 and opt.param_groups[1]["initial_lr"] == 1e-4


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)           # ranks start with DIFFERENT weights ...
        net = nn.Sequential(nn.Linear(5, 7), nn.Softplus(beta=100), nn.Linear(7, 3))
        sync = FlatGradAllReduce(net.parameters())
        sync.broadcast_parameters(0)            # ... and are made identical, as DDP does at wrap time
        torch.manual_seed(7 + rank)             # per-rank batch
        x, y = torch.randn(16, 5), torch.randn(16, 3)
        loss = ((net(x) - y) ** 2).mean()
        loss.backward()
        sync()
        flat = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
        w = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        np.save(os.path.join(out_dir, f"g{rank}.npy"), flat.numpy())
        np.save(os.path.join(out_dir, f"w{rank}.npy"), w.numpy())
        np.save(os.path.join(out_dir, f"x{rank}.npy"), torch.cat([x.reshape(-1), y.reshape(-1)]).numpy())
    finally:
        dist.destroy_process_group()


def test_flat_grad_allreduce_two_ranks(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    w0, w1 = np.load(tmp_path / "w0.npy"), np.load(tmp_path / "w1.npy")
    np.testing.assert_array_equal(g0, g1)       # every rank holds the mean gradient
    np.testing.assert_array_equal(w0, w1)       # parameters were broadcast
    # the mean gradient equals the gradient of the mean loss over both batches, computed in one process
    torch.manual_seed(100)
    net = nn.Sequential(nn.Linear(5, 7), nn.Softplus(beta=100), nn.Linear(7, 3))
    total = 0.0
    for r in range(2):
        xy = torch.from_numpy(np.load(tmp_path / f"x{r}.npy"))
        x, y = xy[:80].reshape(16, 5), xy[80:].reshape(16, 3)
        total = total + ((net(x) - y) ** 2).mean() / 2
    total.backward()
    ref = torch.cat([p.grad.reshape(-1) for p in net.parameters()]).numpy()
    np.testing.assert_allclose(g0, ref, rtol=1e-5, atol=1e-7)


def _worker_zero_copy(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # gradients laid out as the fused training step leaves them: views of ONE flat buffer in parameter order, tagged
        params = [nn.Parameter(torch.zeros(4, 3)), nn.Parameter(torch.zeros(5)), nn.Parameter(torch.zeros(()))]
        flat = torch.arange(18, dtype=torch.float32) * (rank + 1)
        off = 0
        for p in params:
            v = flat[off: off + p.numel()].view(p.shape)
            v._nrh_flat = (flat, off)
            p.grad = v
            off += p.numel()
        ptrs = [p.grad.data_ptr() for p in params]
        sync = FlatGradAllReduce(params)
        sync()
        assert sync._in_place and sync._flat is flat                       # no staging buffer, no copies
        assert [p.grad.data_ptr() for p in params] == ptrs
        np.save(os.path.join(out_dir, f"z{rank}.npy"), flat.numpy())
        # a gradient that is NOT part of the buffer falls back to the copying form
        params[1].grad = params[1].grad.clone()
        sync2 = FlatGradAllReduce(params)
        sync2()
        assert not sync2._in_place
    finally:
        dist.destroy_process_group()


def test_flat_grad_allreduce_zero_copy_two_ranks(tmp_path):
    """when every .grad is a view of one flat buffer (train_fused._Buffers) the exchange is the all-reduce of
    that buffer in place - pack() and unpack() copy nothing."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker_zero_copy, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    z0, z1 = np.load(tmp_path / "z0.npy"), np.load(tmp_path / "z1.npy")
    np.testing.assert_array_equal(z0, z1)
    np.testing.assert_allclose(z0, np.arange(18, dtype=np.float32) * 1.5)      # mean of x1 and x2


def test_checkpoint_roundtrip_reference_layout(tmp_path, scene_states):
    from nrhints_amd.training import load_checkpoint, save_checkpoint
    m = na.NeuSHintRenderer()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in scene_states["b"].items()})
    opt, sched = make_optimizer(m)
    p = str(tmp_path / "step_0000123.ckpt")
    save_checkpoint(p, m, opt, sched, global_step=123, world_size=2,
                    extra_pipeline_state={"ray_generator.cam_pose_adjustment": torch.zeros(10, 6)})
    raw = torch.load(p, weights_only=False)
    assert set(raw) == {"world_size", "global_step", "pipeline", "optimizer", "scheduler"}
    assert "renderer.sdf_network.lin0.weight_g" in raw["pipeline"] and len(raw["pipeline"]) == 47  # SURVEY §5 key contract
    m2 = na.NeuSHintRenderer()
    opt2, sched2 = make_optimizer(m2)
    assert load_checkpoint(p, m2, opt2, sched2) == 123
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k


def test_sdf_function_manual_backward_matches_second_order_autograd(scene_states):
    """tests/torch_backends.SdfValueFeatGrad (the hand-derived tangent/adjoint sweeps the HIP kernels implement) against the oracle's create_graph autograd
    (the reference's formulation, fields/sdf_field.py:136-148) in fp64: gradients w.r.t. the points and all 40 raw
    SDF-network parameters through value, feature, d sdf/dx and an eikonal term."""
    import numpy as np
    from nrhints_amd import packing
    from tests.torch_backends import sdf_value_feat_grad_manual as sdf_value_feat_grad
    from oracle import neus_oracle as O
    torch.manual_seed(0)
    state = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in scene_states["b"].items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in state.items() if k.startswith("sdf_network")}
    pts = (torch.rand(40, 3, dtype=torch.float64) * 1.2 - 0.6)
    cs, cf, cg = (torch.randn(40, k, dtype=torch.float64) for k in (1, 256, 3))

    def loss_of(sdf, feat, g):
        return (sdf * cs).sum() + (feat * cf).sum() + (g * cg).sum() + ((g.norm(dim=-1) - 1) ** 2).sum()

    p1 = pts.clone().requires_grad_(True)
    l1 = loss_of(*sdf_value_feat_grad(packing.dense_params({**state, **leaves}), p1))
    g1 = torch.autograd.grad(l1, [p1] + list(leaves.values()))
    P = O.params_from_state({**state, **leaves}, dtype=torch.float64)
    p2 = pts.clone().requires_grad_(True)
    sdf2, feat2 = O.sdf_forward(P, p2)
    l2 = loss_of(sdf2, feat2, O.sdf_gradient_autograd(P, p2, create_graph=True))
    g2 = torch.autograd.grad(l2, [p2] + list(leaves.values()))
    assert abs(l1.item() - l2.item()) < 1e-10 * max(1.0, abs(l2.item()))
    for name, a, b in zip(["pts"] + list(leaves), g1, g2):
        assert (a - b).abs().max().item() <= 1e-10 * (b.abs().max().item() + 1e-30), name


@pytest.mark.parametrize("hints", [True, False])
def test_render_core_matches_oracle_formulation(scene_states, hints):
    """tests/torch_backends.render_core_torch (hand-derived SDF backward + the column-block form of the reflectance net's first layer:
    the torch statement of what autograd_core.render_core computes with HIP kernels)
    against the reference formulation restated in the oracle - second-order autograd, 361-wide concatenated input
    (models/neus_hint_model.py:504-510, :521-525, :621-637) - in fp64 on the CPU: rgb, weights, normals and the
    gradients w.r.t. all 46 raw parameters and the rays."""
    import numpy as np
    from nrhints_amd import packing
    from nrhints_amd.synthetic import naive_state
    from tests.torch_backends import render_core_torch
    from oracle import neus_oracle as O
    torch.manual_seed(1)
    f64 = torch.float64
    st = scene_states["b"] if hints else naive_state(scene_states["b"])
    n, T_ = 4, 128
    o = (torch.randn(n, 3, dtype=f64) * 0.2)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, dtype=f64), dim=-1)
    pl = torch.randn(n, 3, dtype=f64) * 3.0
    mid = torch.rand(n, T_, dtype=f64).sort(-1).values * 0.9
    dists = torch.rand(n, T_, dtype=f64) * 0.02
    vis = torch.rand(n, 1, dtype=f64) if hints else None
    cue = torch.rand(n, 4, dtype=f64) if hints else None
    bg = torch.ones(1, 3, dtype=f64)
    gt = torch.rand(n, 3, dtype=f64)

    def run(which):
        leaves = {k: torch.tensor(np.asarray(v), dtype=f64).requires_grad_(True) for k, v in st.items()}
        rays = [t.clone().requires_grad_(True) for t in (o, dirs, pl)]
        if which == "core":
            out = render_core_torch(packing.dense_params(leaves), leaves["deviation_network.variance"], *rays, mid, dists,
                                            vis, cue, 0.6, bg, sdf_impl="manual")
            rgb, w, g = out["rgb"], out["weights"], out["analytic_normals"]
        else:
            P = O.params_from_state(leaves, dtype=f64)
            ro, rd, rpl = rays
            pts = (ro[:, None, :] + rd[:, None, :] * mid[..., None]).reshape(-1, 3)
            sdf, feat = O.sdf_forward(P, pts)
            grad = O.sdf_gradient_autograd(P, pts, create_graph=True)
            view = rd[:, None, :].expand(n, T_, 3).reshape(-1, 3)
            inv_s = torch.exp(P.variance * 10.0).clip(1e-6, 1e6)
            alpha = O.alpha_from(sdf, grad, view, dists.reshape(-1, 1), inv_s, 0.6).reshape(n, T_)
            w = alpha * O.excl_cumprod_one_minus(alpha)
            rep = lambda x: x[:, None, :].expand(n, T_, x.shape[-1]).reshape(n * T_, -1)
            col = O.color_forward(P, pts, torch.nn.functional.normalize(grad, dim=-1), view, feat, rep(rpl),
                                  rep(vis) if hints else None, rep(cue) if hints else None).reshape(n, T_, 3)
            rgb = (col * w[..., None]).sum(1) + bg * (1.0 - w.sum(-1, keepdim=True))
            g = grad.reshape(n, T_, 3)
        loss = (rgb - gt).abs().sum() / n + 0.1 * ((g.norm(dim=-1) - 1.0) ** 2).mean() + 0.01 * (w ** 2).sum()
        grads = torch.autograd.grad(loss, list(leaves.values()) + rays)
        return rgb.detach(), w.detach(), grads, list(leaves) + ["o", "d", "pl"]

    rgb1, w1, g1, names = run("core")
    rgb2, w2, g2, _ = run("oracle")
    assert (rgb1 - rgb2).abs().max() < 1e-12 and (w1 - w2).abs().max() < 1e-12
    for name, a, b in zip(names, g1, g2):
        assert (a - b).abs().max().item() <= 1e-9 * (b.abs().max().item() + 1e-12), name
