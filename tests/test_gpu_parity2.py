"""GPU parity tests, second file: the branches and kernels the first file reaches only through whole renders - geometry
warm-up, the ray-generation kernels against the reference's fixture, the reflectance kernel and the alpha formula against the
reference's unit vectors, hipGraph replay against the eager step, the f16x3 split under large magnitudes, and the RCCL code
paths on one GPU (world size 1).  Everything calls through the C ABI (ctypes)."""
import os
import socket

import numpy as np
import pytest
import torch

import nrhints_amd as na
from nrhints_amd import ops, packing as pk
from nrhints_amd.synthetic import make_rays
from oracle import neus_oracle as orc
from tests.conftest import grad_bound, load_npz

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def cu(a):
    return (T(a) if isinstance(a, np.ndarray) else a).float().contiguous().cuda()


def _bundle(o, d, pl, near, far):
    return na.RayBundle(origins=cu(o), directions=cu(d), pl_positions=cu(pl), nears=cu(near), fars=cu(far))


def _model(state, prec="f16x3", cfg=None, train=False):
    m = na.NeuSHintRenderer(cfg or na.NeuSModelConfig(), precision=prec)
    m.load_state_dict({k: T(np.asarray(v)) for k, v in state.items()})
    m = m.cuda()
    return m if train else m.eval()


# ---- geometry warm-up ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_geometry_warmup_vs_reference(scene_states, prec):
    """Training below geometry_warmup_end (models/neus_hint_model.py:668, :577-579, :617-619; the Fish scene's recipe):
    zero hints into the reflectance net, no shadow march.  Outputs, loss and parameter gradients against what the imported
    reference produced (tests/golden/make_golden_warmup.py -> warmup_b.npz)."""
    from nrhints_amd.training import train_loss_dict
    g = load_npz("warmup_b.npz")
    model = _model(scene_states["b"], prec, na.NeuSModelConfig(geometry_warmup_end=int(g["geometry_warmup_end"])), train=True)
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    out = model(rb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=int(g["global_step"]),
                _t_rand_primary=cu(g["t_rand_primary"]))
    assert float(out.visibilities.abs().max()) == 0.0 and float(out.specular_cue.abs().max()) == 0.0
    np.testing.assert_allclose(out.rgb.detach().cpu().numpy(), g["rgb"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(out.depth.detach().cpu().numpy(), g["depth"], rtol=0, atol=2e-4)
    dw = np.abs(out.weights.detach().cpu().numpy() - g["weights"])
    assert dw.mean() < 3e-5 and dw.max() < 3e-2, (dw.mean(), dw.max())
    loss = train_loss_dict(out, cu(g["rgb_gt"]))["loss"]
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=2e-4)
    loss.backward()
    grads = dict(model.named_parameters())
    from tests.conftest import grad_bound
    for k in (k for k in g if k.startswith("grad.")):
        # bound derived from the fixture: 3 x the reference's own fp32-vs-fp64 distance on this tensor (d loss / d variance is a
        # 5e-7 scalar at this step - cos-anneal ratio 0.002, heavy cancellation: the reference's fp32 value is 53 % off its fp64 one)
        tol, scale = grad_bound(g[k], g["grad64." + k[5:]])
        err = float(np.abs(grads[k[5:]].grad.detach().cpu().numpy().astype(np.float64) - g["grad64." + k[5:]]).max())
        assert err < tol, (k, err / scale, tol / scale)
    # evaluation never takes the warm-up branch (:668 is_training and ...)
    with torch.no_grad():
        ev = model(rb, is_training=False, background_rgb=torch.ones(1, 3).cuda(), global_step=int(g["global_step"]))
    assert float(ev.visibilities.max()) > 0.0


# ---- ray generation kernels --------------------------------------------------------------------------------------------
def test_raygen_kernels_vs_reference_fixture():
    """nrh_generate_rays_indexed (+ its adjoint) and nrh_generate_rays against the imported reference's RayGenerator
    (camera/ray_generator.py:75-150 -> tests/golden/raygen.npz): off / SO3xR3 + light / SE3 / video (no view index) /
    noise buffers / z-plane near-far; rays and the gradients of a fixed scalar w.r.t. the per-view adjustments."""
    from nrhints_amd.containers import RawPixelBundle
    from nrhints_amd.pipeline import CameraModel, generate_rays
    from nrhints_amd.ray_generator import RayGenerator, RayGeneratorConfig
    g = load_npz("raygen.npz")
    H, W, cx, cy, fx, fy, zn, zf = g["camera"]
    cam = CameraModel(H=int(H), W=int(W), cx=float(cx), cy=float(cy), fx=float(fx), fy=float(fy))
    c3 = cu(g["probe"])
    runs = {"off": (RayGeneratorConfig(), None, True),
            "so3": (RayGeneratorConfig(cam_opt_mode="SO3xR3", pl_opt=True), None, True),
            "se3": (RayGeneratorConfig(cam_opt_mode="SE3"), None, True),
            "video": (RayGeneratorConfig(cam_opt_mode="SO3xR3", pl_opt=True), None, False),
            "noise": (RayGeneratorConfig(cam_opt_mode="SO3xR3", cam_position_noise_std=0.02, cam_orientation_noise_std=0.03,
                                         pl_position_noise_std=0.05), 11, True),
            "zplanes": (RayGeneratorConfig(override_near_far_from_sphere=False), None, True)}
    for tag, (cfg, seed, with_idx) in runs.items():
        if seed is not None:
            torch.manual_seed(seed)
        rg = RayGenerator(cam, 5, cfg, zn=float(zn), zf=float(zf))     # noise buffers drawn on the CPU like the reference
        if hasattr(rg, "cam_pose_adjustment"):
            rg.cam_pose_adjustment.data.copy_(T(g["adj"]))
        if hasattr(rg, "pl_adjustment"):
            rg.pl_adjustment.data.copy_(T(g["pladj"]))
        rg = rg.cuda()
        pb = RawPixelBundle(img_indices=T(g["img_indices"]).cuda() if with_idx else None, h_indices=cu(g["h_indices"]),
                            w_indices=cu(g["w_indices"]), poses=cu(g["poses"]), pls=cu(g["pls"]))
        rb = rg(pb)
        for k in ("origins", "directions", "pl_positions", "nears", "fars"):
            np.testing.assert_allclose(getattr(rb, k).detach().cpu().numpy(), g[f"{tag}.{k}"], rtol=0, atol=3e-6, err_msg=f"{tag}.{k}")
        names = [n for n, _ in rg.named_parameters()]
        if names and with_idx:
            loss = (rb.origins * c3).sum() + (rb.directions * c3.flip(0)).sum() * 2.0 + (rb.pl_positions * c3).sum() * 0.5 + \
                   (rb.nears * rb.fars).sum() * 0.1
            for n, gr in zip(names, torch.autograd.grad(loss, list(rg.parameters()))):
                want = g[f"{tag}.grad.{n}"]
                np.testing.assert_allclose(gr.cpu().numpy(), want, rtol=0, atol=2e-5 * max(1.0, np.abs(want).max()),
                                           err_msg=f"{tag}.grad.{n}")
    # the whole-view raster kernel against the fixture's scattered pixels of each pose ...
    for i in range(0, 40, 7):
        h, w = int(g["h_indices"][i, 0]), int(g["w_indices"][i, 0])
        row = generate_rays(cam, T(g["poses"][i]), T(g["pls"][i]), "cuda", row0=h, row1=h + 1)
        for k in ("origins", "directions", "pl_positions", "nears", "fars"):
            np.testing.assert_allclose(getattr(row, k)[w].cpu().numpy(), g[f"off.{k}"][i], rtol=0, atol=3e-6)
    # ... and against the indexed kernel over a full image (same formulas: bit-identical)
    hh, ww = torch.meshgrid(torch.arange(cam.H, dtype=torch.float32), torch.arange(cam.W, dtype=torch.float32), indexing="ij")
    n = cam.H * cam.W
    pb = RawPixelBundle(img_indices=None, h_indices=hh.reshape(n, 1).cuda(), w_indices=ww.reshape(n, 1).cuda(),
                        poses=cu(g["poses"][3])[None].expand(n, 4, 4).contiguous(), pls=cu(g["pls"][3])[None].expand(n, 3).contiguous())
    a, b = RayGenerator(cam, 5).cuda()(pb), generate_rays(cam, T(g["poses"][3]), T(g["pls"][3]), "cuda")
    for k in ("origins", "directions", "pl_positions", "nears", "fars"):
        assert torch.equal(getattr(a, k), getattr(b, k)), k


def test_raygen_indexed_argument_errors():
    from nrhints_amd import _lib
    lib = _lib.load()
    one = torch.zeros(16, device="cuda")
    P = _lib.ptr
    assert lib.nrh_generate_rays_indexed(None, P(one), P(one), P(one), 12, P(one), 1, P(one), None, 0, 0., 0., 1., 1., 1, .1, 10.,
                                         P(one), P(one), P(one), P(one), P(one), None) == -1
    assert "img_indices" in lib.nrh_last_error_string().decode()
    assert lib.nrh_generate_rays_indexed(None, P(one), P(one), P(one), 8, P(one), 1, None, None, 0, 0., 0., 1., 1., 1, .1, 10.,
                                         P(one), P(one), P(one), P(one), P(one), None) == -1
    assert lib.nrh_generate_rays_indexed(None, None, None, None, 12, None, 0, None, None, 0, 0., 0., 1., 1., 1, .1, 10.,
                                         None, None, None, None, None, None) == 0


# ---- reflectance kernel and alpha formula against the reference's unit vectors --------------------------------------------
@pytest.mark.parametrize("tag,prec", [("a", "f32"), ("b", "f32"), ("a", "f16x3"), ("b", "f16x3")])
def test_color_kernel_vs_reference_unit_vectors(scene_states, tag, prec):
    """color_kernel against the imported reference's ReflectanceNetwork outputs (fields/reflectance_network.py:68-96 ->
    unit_*.npz:col_out): each of the 160 fixture points becomes one ray whose 128 samples all sit on that point
    (origin = p - v, direction = v, t = 1)."""
    u = load_npz(f"unit_{tag}.npz")
    model = _model(scene_states[tag], prec)
    packed = model.packed_params(torch.device("cuda", torch.cuda.current_device()))
    N = u["col_pts"].shape[0]
    v, p = T(u["col_v"]), T(u["col_pts"])
    rep = lambda x: x[:, None, :].expand(N, 128, x.shape[-1]).reshape(N * 128, -1).contiguous()
    raymisc = torch.zeros(N, pk.RAYMISC_STRIDE)
    raymisc[:, 0:27] = orc.nerf_encode(v, 4)
    raymisc[:, 27:54] = orc.nerf_encode(T(u["col_pl"]), 4)
    raymisc[:, 54:63] = orc.nerf_encode(T(u["col_vis"]), 4)
    raymisc[:, 63:99] = orc.nerf_encode(T(u["col_cue"]), 4)
    col = ops.color_eval(packed["col_w"], packed["col_b"], pk.rows_to_feat_tiles(rep(T(u["col_feat"]))).cuda(), cu(p - v), cu(v),
                         torch.ones(N, 128).cuda(), cu(rep(T(u["col_n"]))), raymisc.cuda())
    got = col.reshape(N, 128, 3).cpu().numpy()
    assert np.abs(got - got[:, :1]).max() == 0.0            # the same point 128 times
    np.testing.assert_allclose(got[:, 0], u["col_out"], rtol=0, atol=5e-6)     # sigmoid outputs in [0,1]; p - v + v rounds once


@pytest.mark.parametrize("tag,prec", [("a", "f32"), ("b", "f32"), ("a", "f16x3"), ("b", "f16x3")])
def test_alpha_vs_reference_unit_vectors(scene_states, tag, prec):
    """get_alpha (models/neus_hint_model.py:339-356) against the imported reference's values (unit_*.npz:alpha_r1.0 /
    alpha_r0.37): SDF + gradient from the HIP MLP kernel at the fixture points, alpha from the alpha kernel - sample 0 of a
    ray has transmittance 1, so its weight IS alpha."""
    from nrhints_amd.autograd_core import AlphaWeightsNormalsHip
    u = load_npz(f"unit_{tag}.npz")
    model = _model(scene_states[tag], prec)
    packed = model.packed_params(torch.device("cuda", torch.cuda.current_device()))
    N = u["alpha_pts"].shape[0]
    sdf, grad = ops.sdf_at_points(1, packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], cu(u["alpha_pts"]))[:2]
    rep = lambda x: x.reshape(N, 1, -1).expand(N, 128, x.shape[-1]).reshape(N * 128, -1).contiguous()
    var = model.deviation_network.variance.detach()
    inv_s = float(torch.exp(var * 10.0).clip(1e-6, 1e6))
    for ratio in (1.0, 0.37):
        w, _ = AlphaWeightsNormalsHip.apply(rep(sdf.reshape(N, 1)), rep(grad.reshape(N, 3)), cu(u["alpha_dirs"]),
                                            cu(u["alpha_dists"]).expand(N, 128).contiguous(), var, inv_s, ratio)
        want = u[f"alpha_r{ratio}"]
        # alpha = 1 - sigmoid ratio at inv_s up to e^7: an SDF error of 1e-6 moves it by ~1e-3 * alpha near the surface
        np.testing.assert_allclose(w[:, :1].cpu().numpy(), want, rtol=0, atol=2e-4 if prec == "f32" else 4e-4)


# ---- hipGraph replay == eager ---------------------------------------------------------------------------------------------
def test_graph_replay_equals_eager_step(scene_states):
    """GraphedTrainStep against the eager step (training.train_step's sequence) on the same batches with the same jitter:
    losses of three consecutive steps, the gradients and the parameters after every step; an evaluation render BETWEEN
    replays sees the replayed parameters (pack cache).  The eager side uses the same optimiser as the graph (adam.HipAdam,
    tensor lr): with it the replay reproduces the eager step to the last bit, while two Adam implementations that differ by
    1 ulp after one step diverge visibly - the renderer amplifies it (a sample crossing the surface changes sensitive gradient
    entries by ~1 %; profiles/r02/graph_vs_eager_probe.log)."""
    from nrhints_amd.training import GraphedTrainStep, lr_factor, train_loss_dict
    n, lr, gs = 128, 5e-4, 30000
    bg = torch.ones(1, 3).cuda()
    rs = np.random.RandomState(5)
    batches = [(_bundle(*make_rays(n, seed=40 + i, spread=0.1)), cu(rs.rand(n, 3).astype(np.float32))) for i in range(3)]
    jit = [(cu(rs.rand(n, 1).astype(np.float32)), cu(rs.rand(n, 64).astype(np.float32))) for _ in range(3)]
    rb_eval = _bundle(*make_rays(200, seed=77, spread=0.1))
    eager = _model(scene_states["b"], train=True)
    lr_t = torch.tensor(lr, device="cuda")
    from nrhints_amd.adam import HipAdam
    opt = HipAdam([{"params": list(eager.parameters()), "lr": lr_t}])
    graphed = _model(scene_states["b"], train=True)
    before = {k: v.detach().clone() for k, v in graphed.state_dict().items()}
    step = GraphedTrainStep(graphed, n, bg, lr=lr, warm_up_end=20, global_step=gs,
                            jitter=(torch.zeros(n, 1), torch.zeros(n, 64)), fused=False)   # autograd path on both sides (fused: test_gpu_train_fused.py)
    for k, v in graphed.state_dict().items():       # capture (3 warm-up steps + graph build) leaves the model untouched
        assert torch.equal(v, before[k]), k
    for i, ((rb, gt), (tp, ts)) in enumerate(zip(batches, jit)):
        lr_t.fill_(lr * lr_factor(gs + i, 20, 1_000_000, 0.05))
        out = eager(rb, is_training=True, background_rgb=bg, global_step=gs + i, _t_rand_primary=tp, _t_rand_shadow=ts)
        ld = train_loss_dict(out, gt, eager.config.igr_weight)
        opt.zero_grad(set_to_none=True)
        ld["loss"].backward()
        grads = {k: p.grad.detach().clone() for k, p in eager.named_parameters()}
        opt.step()
        step.jitter[0].copy_(tp); step.jitter[1].copy_(ts)
        loss = step(rb, gt, global_step=gs + i)["loss"]
        assert abs(loss - float(ld["loss"].detach())) <= 1e-6 * abs(loss), (i, loss, float(ld["loss"].detach()))
        for k, p in graphed.named_parameters():
            scale = float(grads[k].abs().max()) + 1e-30
            assert float((p.grad - grads[k]).abs().max()) <= 1e-6 * scale, (i, k)
            assert float((p.detach() - dict(eager.named_parameters())[k].detach()).abs().max()) <= 2e-7, (i, k)
        with torch.no_grad():
            ev, ev_e = graphed(rb_eval, background_rgb=bg).rgb, eager(rb_eval, background_rgb=bg).rgb
        assert float((ev - ev_e).abs().max()) < 2e-6, i            # a stale pack would render the previous step's weights
    moved = max(float((p.detach() - before[k]).abs().max()) for k, p in graphed.named_parameters())
    assert moved > 2 * lr
    step.release()


# ---- f16x3 under stress ---------------------------------------------------------------------------------------------
def test_f16x3_large_magnitude_weights_match_or_raise(scene_states):
    """The fp16 3-term split covers |x| < 65504 per operand with ~2^-22 relative error; fine-tuned checkpoints with large
    weights must either still match fp32 or be refused - never silently wrong.  Scale one hidden layer's weight-norm gain
    (activations x8 through a softplus stack) and compare the two precisions; push it past the fp16 range and expect the
    packer to raise."""
    st = {k: np.asarray(v).copy() for k, v in scene_states["b"].items()}
    st["sdf_network.lin3.weight_g"] = st["sdf_network.lin3.weight_g"] * 8.0
    st["sdf_network.lin4.weight_g"] = st["sdf_network.lin4.weight_g"] / 8.0
    pts = cu(np.random.RandomState(3).uniform(-1, 1, size=(4096, 3)).astype(np.float32))
    res = {}
    for prec in ("f32", "f16x3"):
        m = _model(st, prec)
        packed = m.packed_params(torch.device("cuda", torch.cuda.current_device()))
        res[prec] = ops.sdf_at_points(2, packed["sdf_w"], packed["sdf_b"], packed["sdf_head"], pts)
    ref_sdf, ref_grad = orc.sdf_forward(orc.params_from_state(st), pts.cpu(), want_feat=False)[0], None
    np.testing.assert_allclose(res["f32"][0].cpu().numpy().reshape(-1), ref_sdf.numpy().reshape(-1), rtol=0, atol=5e-6)
    np.testing.assert_allclose(res["f16x3"][0].cpu().numpy(), res["f32"][0].cpu().numpy(), rtol=0, atol=5e-6)
    g32, g16 = res["f32"][1].cpu().numpy(), res["f16x3"][1].cpu().numpy()
    assert np.abs(g16 - g32).max() < 2e-4 * max(1.0, np.abs(g32).max())
    assert torch.isfinite(res["f16x3"][0]).all() and torch.isfinite(res["f16x3"][1]).all()
    # whole render with the scaled layer: the two precisions agree to the headline tolerance
    rb = _bundle(*make_rays(256, seed=5, spread=0.12))
    with torch.no_grad():
        a = _model(st, "f32")(rb, background_rgb=torch.ones(1, 3).cuda())
        b = _model(st, "f16x3")(rb, background_rgb=torch.ones(1, 3).cuda())
    assert float((a.rgb - b.rgb).abs().max()) < 1e-4
    # out of range for the split: refuse
    st2 = {k: np.asarray(v).copy() for k, v in scene_states["b"].items()}
    st2["sdf_network.lin3.weight_g"] = st2["sdf_network.lin3.weight_g"] * 1e6
    m = _model(st2, "f16x3")
    with pytest.raises((ValueError, RuntimeError), match="f16x3|fp16|range"):
        with torch.no_grad():
            m(rb, background_rgb=torch.ones(1, 3).cuda())


# ---- RCCL paths on one GPU ---------------------------------------------------------------------------------------------
@pytest.fixture()
def nccl_world1():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_world1_render_sharded_grad_allreduce_and_graph(scene_states, nccl_world1):
    """backend 'nccl' (= RCCL) with one rank: render_sharded's all-gather returns exactly the local render,
    FlatGradAllReduce(always=True) runs the flat all-reduce and leaves the mean-over-1 gradients unchanged, and a
    GraphedTrainStep with the all-reduce captured INSIDE the hipGraph replays and trains."""
    from nrhints_amd.parallel import render_sharded
    from nrhints_amd.training import FlatGradAllReduce, GraphedTrainStep, train_loss_dict
    bg = torch.ones(1, 3).cuda()
    model = _model(scene_states["b"], train=True)
    rb = _bundle(*make_rays(1001, seed=8, spread=0.12))
    with torch.no_grad():
        local = model(rb, background_rgb=bg)
        res = render_sharded(lambda r: model(r, background_rgb=bg), rb, fields=("rgb", "depth", "visibilities"))
    for f in ("rgb", "depth", "visibilities"):
        assert torch.equal(res[f], getattr(local, f)), f
    n = 128
    rb = _bundle(*make_rays(n, seed=9, spread=0.1))
    rs = np.random.RandomState(1)
    gt, tp, ts = (cu(rs.rand(n, k).astype(np.float32)) for k in (3, 1, 64))
    out = model(rb, is_training=True, background_rgb=bg, global_step=30000, _t_rand_primary=tp, _t_rand_shadow=ts)
    train_loss_dict(out, gt)["loss"].backward()
    want = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    sync = FlatGradAllReduce(model.parameters(), always=True)
    sync.broadcast_parameters()
    sync()
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, want[k]), k
    model.zero_grad(set_to_none=True)
    step = GraphedTrainStep(model, n, bg, lr=5e-4, warm_up_end=20, global_step=30000, grad_sync=sync, jitter=(tp, ts))
    plain = _model(scene_states["b"], train=True)
    step2 = GraphedTrainStep(plain, n, bg, lr=5e-4, warm_up_end=20, global_step=30000, jitter=(tp, ts))
    # the same step with the all-reduce captured INSIDE its one hipGraph (collective_in_graph): bit-equal to the two-graph form
    single = _model(scene_states["b"], train=True)
    sync1 = FlatGradAllReduce(single.parameters(), always=True)
    step3 = GraphedTrainStep(single, n, bg, lr=5e-4, warm_up_end=20, global_step=30000, grad_sync=sync1, jitter=(tp, ts), collective_in_graph=True)
    assert step.graph_tail is not None and step3.graph_tail is None and step3.collective_in_graph
    for i in range(3):
        a, b = step(rb, gt, global_step=30000 + i)["loss"], step2(rb, gt, global_step=30000 + i)["loss"]
        c = step3(rb, gt, global_step=30000 + i)["loss"]
        assert np.isfinite(a) and abs(a - b) < 2e-5 * max(1.0, abs(b)), (i, a, b)
        assert c == a, (i, c, a)
    for (k, pa), (_, pc) in zip(model.named_parameters(), single.named_parameters()):
        assert torch.equal(pa.detach(), pc.detach()), k
    step.release(); step2.release(); step3.release()
    # what bench.py's multi-rank training leg runs per rank: eager fused steps without a read-back, the flat all-reduce between
    # backward and the one-launch Adam - against the same steps without the exchange
    from nrhints_amd.training import make_optimizer, release_device_scalars, train_step
    ma, mb = _model(scene_states["b"], train=True), _model(scene_states["b"], train=True)
    (oa, sa), (ob, sb) = make_optimizer(ma, warm_up_end=2), make_optimizer(mb, warm_up_end=2)
    sync_a = FlatGradAllReduce(list(ma.parameters()), always=True)
    for i in range(3):
        torch.manual_seed(100 + i)
        la = train_step(ma, rb, gt, bg, 30000 + i, oa, sa, grad_sync=sync_a, sync=False)["loss"]
        torch.manual_seed(100 + i)
        lb = train_step(mb, rb, gt, bg, 30000 + i, ob, sb, sync=False)["loss"]
        assert torch.is_tensor(la) and abs(float(la) - float(lb)) < 2e-5 * max(1.0, abs(float(lb))), (i, float(la), float(lb))
    for (k, a), (_, b) in zip(ma.named_parameters(), mb.named_parameters()):
        assert float((a.detach() - b.detach()).abs().max()) < 1e-5, k
    release_device_scalars(ma); release_device_scalars(mb)


# ---- unit entries of the evaluation render's per-ray stages against the reference's recorded intermediates ----------------
@pytest.mark.parametrize("tag", ["a", "b"])
def test_alpha_composite_entry_vs_reference(tag):
    """nrh_alpha_composite (core_alpha_kernel, the kernel nrh_render_forward launches) on the reference's own sdf / gradients /
    section lengths (tests/golden/core_*.npz): alpha-composite weights, inside mask, unit normals, depth, hit point, hit normal,
    Cook-Torrance cue and the shadow ray it sets up (models/neus_hint_model.py:339-356, :512-533, :583-616, :380-386)."""
    g = load_npz(f"core_{tag}.npz")
    out = ops.alpha_composite(cu(g["o"]), cu(g["d"]), cu(g["pl"]), cu(g["sdf"]), cu(g["grad"]), cu(g["dists"]), cu(g["mid_z"]),
                              float(g["inv_s"]), 1.0)
    c = lambda k: out[k].cpu().numpy()
    np.testing.assert_allclose(c("weights"), g["weights"], rtol=0, atol=2e-6)      # 128-long product scan in another order
    np.testing.assert_array_equal(c("inside"), g["inside_sphere"])
    np.testing.assert_allclose(c("nhat"), g["nhat"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(c("depth"), g["depth"][:, 0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(c("wsum"), g["weights"].sum(-1), rtol=0, atol=1e-5)
    np.testing.assert_allclose(c("hit"), g["hit_points"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c("hit_normal"), g["hit_normal"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(c("cue"), g["cue"], rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(g["cue"]).max())))
    np.testing.assert_allclose(c("shadow_dirs"), g["s_dirs"], rtol=0, atol=2e-6)
    L = np.linalg.norm(g["hit_points"].astype(np.float64) - g["pl"], axis=-1)
    np.testing.assert_allclose(c("shadow_last_dist"), L / 64.0, rtol=2e-6)
    lin = torch.linspace(0.0, 1.0, 64).numpy()
    np.testing.assert_allclose(c("shadow_z")[:, :64], lin[None, :] * L[:, None] * (1.0 - 1e-2), rtol=3e-6, atol=1e-7)
    # MaximalWeightPoint depth (:534-538) through the same entry
    mw = ops.alpha_composite(cu(g["o"]), cu(g["d"]), cu(g["pl"]), cu(g["sdf"]), cu(g["grad"]), cu(g["dists"]), cu(g["mid_z"]),
                             float(g["inv_s"]), 1.0, depth_type=1)
    idx = g["weights"].argmax(-1)
    want = g["mid_z"][np.arange(idx.size), idx]
    assert np.mean(mw["depth"].cpu().numpy() != want) < 0.05      # an argmax between two near-equal weights may flip


@pytest.mark.parametrize("tag", ["a", "b"])
def test_visibility_entry_vs_reference(tag):
    """nrh_visibility (shadow_finish_kernel) on the reference's shadow-ray sdf / gradients / section lengths: the visibility
    (:429-432) and the reflectance net's per-ray encodings (fields/reflectance_network.py:70-84)."""
    g = load_npz(f"core_{tag}.npz")
    vis, raymisc = ops.visibility(cu(g["d"]), cu(g["pl"]), cu(g["s_dirs"]), cu(g["s_sdf"]), cu(g["s_grad"]), cu(g["s_dists"]), cu(g["cue"]),
                                  float(g["inv_s"]), 1.0)
    np.testing.assert_allclose(vis.cpu().numpy(), g["vis"][:, 0], rtol=0, atol=2e-6)
    want = torch.cat([orc.nerf_encode(T(g["d"]), 4), orc.nerf_encode(T(g["pl"]), 4), orc.nerf_encode(vis.cpu()[:, None], 4),
                      orc.nerf_encode(T(g["cue"]), 4)], dim=-1).numpy()
    # sin of arguments up to 8 * |pl| ~ 40 and 8 * cue: the kernel's Cody-Waite sine is within 2 ulp of libm's
    np.testing.assert_allclose(raymisc.cpu().numpy()[:, :99], want, rtol=0, atol=5e-6)
    v0, r0 = ops.visibility(cu(g["d"]), cu(g["pl"]), None, None, None, None, None, float(g["inv_s"]), 1.0, zero_hints=True)
    assert float(v0.abs().max()) == 0.0 and float(r0[:, 54:55].abs().max()) == 0.0          # warm-up: hints are zero (:577-579)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_color_composite_entry_vs_reference(tag):
    """nrh_color_composite (composite_kernel) on the reference's sampled colours and weights: rgb with white / black / no
    background (:635-637) and the weighted normal maps of the evaluation loop (pipelines/base_pipeline.py:125-131)."""
    g = load_npz(f"core_{tag}.npz")
    w = cu(g["weights"])
    wsum = w.sum(-1).contiguous()
    col = cu(g["sampled_color"].reshape(-1, 3))
    rgb1, nm, nnm = ops.color_composite(col, w, wsum, torch.ones(3).cuda(), cu(g["inside_sphere"]), cu(g["analytic_normals"].reshape(-1, 3)),
                                        cu(g["nhat"]), maps=True)
    np.testing.assert_allclose(rgb1.cpu().numpy(), g["rgb"], rtol=0, atol=2e-6)
    rgb0, _, _ = ops.color_composite(col, w, wsum, torch.zeros(3).cuda())
    np.testing.assert_allclose(rgb0.cpu().numpy(), g["rgb_bg0"], rtol=0, atol=2e-6)
    rgbn, _, _ = ops.color_composite(col, w, wsum, None)
    assert torch.equal(rgbn, rgb0)
    for got, field in ((nm, g["analytic_normals"]), (nnm, g["nhat"].reshape(-1, 128, 3))):
        want = np.einsum("nij,ni,ni->nj", field.astype(np.float64), g["weights"].astype(np.float64), g["inside_sphere"].astype(np.float64))
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0, atol=5e-6 * max(1.0, float(np.abs(want).max())))


def test_get_eval_dicts_vs_reference_fixture(scene_states):
    """The evaluation loop's products against the dictionaries the reference's own ``get_eval_dicts`` returned for one 24 x 32
    view of scene b (tests/golden/evaldict_b.npz, recorded by importing pipelines/base_pipeline.py:93-160): same keys, shapes and
    dtypes; rgb / depth / shadow map / the two camera-space normal maps / the specular hint within the end-to-end render
    tolerances of test_gpu_parity.py (the sampler places samples at fp32 noise); PSNR against the same ground truth; and the
    uint8 images its trainer would write (trainer/trainer.py:343-352) equal up to one code value at pixels whose float value
    sits on a code boundary."""
    from nrhints_amd.pipeline import CameraModel, get_eval_dicts, to_uint8_images
    fx = load_npz("evaldict_b.npz")
    H, W, cx, cy, fxx, fyy = fx["camera"]
    cam = CameraModel(H=int(H), W=int(W), cx=float(cx), cy=float(cy), fx=float(fxx), fy=float(fyy))
    model = _model(scene_states["b"])
    img, metrics, tensors = get_eval_dicts(model, cam, T(fx["pose"]), T(fx["pl"]), rgb_gt=T(fx["rgb_gt"]).cuda())
    want_img = {k[4:]: v for k, v in fx.items() if k.startswith("img.")}
    want_t = {k[7:]: v for k, v in fx.items() if k.startswith("tensor.")}
    assert set(img) == set(want_img) and set(tensors) == set(want_t) and set(metrics) == {"psnr"}
    for k, v in want_img.items():
        assert img[k].shape == v.shape and img[k].dtype == v.dtype, k
    for k, v in want_t.items():
        assert tensors[k].shape == v.shape and tensors[k].dtype == v.dtype, k
    tol = {"rgb": 1e-4, "shadow_map": 2e-3, "analytic_normals": 2e-3, "normalized_analytic_normals": 2e-3, "rgb_gt": 0.0}
    for k, v in want_img.items():
        err = float(np.abs(img[k].astype(np.float64) - v).max())
        assert err <= tol[k], (k, err)
    assert float(np.abs(tensors["depth"] - want_t["depth"]).max()) < 2e-3
    assert float(np.abs(tensors["specular_hint"] - want_t["specular_hint"]).max()) < 2e-4
    assert abs(metrics["psnr"] - float(fx["psnr"])) < 1e-4
    u8 = to_uint8_images(img)
    for k in want_img:
        want = fx["u8." + k]
        assert u8[k].shape == want.shape and u8[k].dtype == np.uint8, k
        diff = np.abs(u8[k].astype(np.int32) - want.astype(np.int32))
        assert int(diff.max()) <= 1 and float((diff > 0).mean()) < 0.02, (k, int(diff.max()), float((diff > 0).mean()))


def _orbit_pixels(n, ncam, seed, H=48, W=64):
    """A training batch as the data loader hands it over (data/data_loader.py:183-192): random pixels of ``ncam`` orbit cameras."""
    from nrhints_amd import RawPixelBundle
    rs = np.random.RandomState(seed)
    az = np.linspace(0.2, 5.0, ncam)
    poses = np.zeros((ncam, 4, 4), dtype=np.float32)
    for i, a in enumerate(az):
        pos = 3.6 * np.array([np.cos(0.4) * np.cos(a), np.cos(0.4) * np.sin(a), np.sin(0.4)])
        fwd = -pos / np.linalg.norm(pos)
        right = np.cross(fwd, [0.0, 0.0, 1.0]); right /= np.linalg.norm(right)
        poses[i, :3, :3] = np.stack([right, np.cross(right, fwd), -fwd], axis=1)
        poses[i, :3, 3], poses[i, 3, 3] = pos, 1.0
    pls = (poses[:, :3, 3] * 1.2 + np.array([0.3, -0.2, 0.5])).astype(np.float32)
    img = rs.randint(0, ncam, size=n)
    return RawPixelBundle(img_indices=T(img[:, None]).long().cuda(), h_indices=cu(rs.randint(8, H - 8, size=(n, 1)).astype(np.float32)),
                          w_indices=cu(rs.randint(8, W - 8, size=(n, 1)).astype(np.float32)), poses=cu(poses[img]), pls=cu(pls[img]),
                          rgb_gt=cu(rs.rand(n, 3).astype(np.float32)))


@pytest.mark.parametrize("refine", [True, False])
def test_graph_with_ray_generator_group(scene_states, refine):
    """GraphedTrainStep(ray_generator=...): the step starts at the RawPixelBundle, and Adam carries the reference's second parameter
    group (trainer/trainer.py:99-102).  With pose + light refinement on (cam_opt_mode SO3xR3, pl_opt) the replay must equal
    the eager sequence ray generator -> renderer -> loss -> backward (autograd path) -> two-group Adam on the same batches and
    jitter: losses, the deltas' gradients and values, the renderer's parameters - to fp32 round-off on the first step (the graph
    holds the fused autograd-free step incl. nrh_ray_adjoint and the ray generator's adjoint kernel); the ray generator's adjoint
    scatters into the per-view deltas with atomics, so from the second step on the two runs also differ by summation order in the
    deltas' last bit, which the renderer amplifies (see test_graph_replay_equals_eager_step).  With refinement off the group is
    empty and the optimiser state still has the two-group layout."""
    from nrhints_amd import RayGenerator, RayGeneratorConfig
    from nrhints_amd.pipeline import CameraModel
    from nrhints_amd.training import GraphedTrainStep, lr_factor, train_loss_dict
    n, ncam, lr, rlr, gs = 128, 3, 5e-4, 1e-3, 30000
    cam = CameraModel(H=48, W=64, cx=32.0, cy=24.0, fx=150.0, fy=150.0)
    cfg = RayGeneratorConfig(cam_opt_mode="SO3xR3", pl_opt=True) if refine else RayGeneratorConfig()
    bg = torch.ones(1, 3).cuda()
    rs = np.random.RandomState(9)
    batches = [_orbit_pixels(n, ncam, 60 + i) for i in range(3)]
    jit = [(cu(rs.rand(n, 1).astype(np.float32)), cu(rs.rand(n, 64).astype(np.float32))) for _ in range(3)]
    eager, graphed = _model(scene_states["b"], train=True), _model(scene_states["b"], train=True)
    rg_e, rg_g = RayGenerator(cam, ncam, cfg).cuda(), RayGenerator(cam, ncam, cfg).cuda()
    if refine:
        with torch.no_grad():
            for rg in (rg_e, rg_g):
                rg.cam_pose_adjustment.copy_(T(np.random.RandomState(1).randn(ncam, 6).astype(np.float32)) * 0.01)
                rg.pl_adjustment.copy_(T(np.random.RandomState(2).randn(ncam, 3).astype(np.float32)) * 0.02)
    lr_t, rlr_t = torch.tensor(lr, device="cuda"), torch.tensor(rlr, device="cuda")
    from nrhints_amd.adam import HipAdam
    opt = HipAdam([{"params": list(eager.parameters()), "lr": lr_t}, {"params": list(rg_e.parameters()), "lr": rlr_t}])
    step = GraphedTrainStep(graphed, n, bg, lr=lr, warm_up_end=20, global_step=gs, jitter=(torch.zeros(n, 1), torch.zeros(n, 64)),
                            ray_generator=rg_g, ray_lr=rlr)
    groups = step.optimizer.state_dict()["param_groups"]
    assert len(groups) == 2 and len(groups[1]["params"]) == (2 if refine else 0)
    for (k, a), (_, b) in zip(rg_g.named_parameters(), rg_e.named_parameters()):
        assert torch.equal(a, b), k                                  # capture left the deltas untouched
    with pytest.raises(TypeError):
        step(_bundle(*make_rays(n, seed=1)), batches[0].rgb_gt, global_step=gs)
    for i, (pb, (tp, ts)) in enumerate(zip(batches, jit)):
        f = lr_factor(gs + i, 20, 1_000_000, 0.05)
        lr_t.fill_(lr * f); rlr_t.fill_(rlr * f)
        out = eager(rg_e(pb), is_training=True, background_rgb=bg, global_step=gs + i, _t_rand_primary=tp, _t_rand_shadow=ts)
        ld = train_loss_dict(out, pb.rgb_gt, eager.config.igr_weight)
        opt.zero_grad(set_to_none=True)
        ld["loss"].backward()
        opt.step()
        step.jitter[0].copy_(tp); step.jitter[1].copy_(ts)
        loss = step(pb, pb.rgb_gt, global_step=gs + i)["loss"]
        want = float(ld["loss"].detach())
        # the graph replays the FUSED step in both cases (since round 4 also under refinement: train_fused + nrh_ray_adjoint), the
        # eager side is the autograd path: equal up to fp32 round-off of the few expressions that differ between the two
        assert abs(loss - want) <= (2e-5 if i == 0 else 1e-4) * abs(want), (i, loss, want)     # (after a step the parameters differ at round-off)
        assert abs(float(step.ray_lr_t) - rlr * f) < 1e-10
        for (k, a), (_, b) in zip(rg_g.named_parameters(), rg_e.named_parameters()):
            scale = float(b.grad.abs().max()) + 1e-30
            # (from the second step on the two runs' parameters differ by Adam's amplification of round-off-level gradient entries,
            # see below, and the small light-position gradients follow: 1.4 % of their scale at step 2)
            assert float((a.grad - b.grad).abs().max()) <= (1e-4 if i == 0 else 5e-2) * scale, (i, k)
            assert float((a.detach() - b.detach()).abs().max()) <= (5e-6 if i == 0 else 2e-4), (i, k)      # (a step is rlr f = 1e-3)
        if refine:
            # Adam normalises every entry's step to ~lr whatever its gradient's size, so where the gradient is at round-off level
            # (|g| ~ eps) the fused and the autograd run may step differently by up to 2 lr; everywhere else they agree
            for (k, a), (_, b) in zip(graphed.named_parameters(), eager.named_parameters()):
                dpar = (a.detach() - b.detach()).abs()
                assert float(dpar.max()) <= 2.0 * lr * f + 1e-7, (i, k)
                # (the share of such entries grows with every step the two runs take apart - the sample placement reacts to
                # parameter differences of 1e-6: at most 5 % after the first step, 10 % of the 256 out_feat biases after the third)
                assert float((dpar > 5e-6).float().mean()) <= (5e-2, 1e-1, 2e-1)[i], (i, k, float((dpar > 5e-6).float().mean()))
    if refine:
        assert float(rg_g.cam_pose_adjustment.grad.abs().max()) > 0 and float(rg_g.pl_adjustment.grad.abs().max()) > 0
        moved = float((rg_g.pl_adjustment.detach() - T(np.random.RandomState(2).randn(ncam, 3).astype(np.float32)).cuda() * 0.02).abs().max())
        assert moved > rlr
    step.release()


# ---- further off-default branches ---------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_sphere_tracing_depth(scene_states, prec):
    """DepthComputationType.SphereTracing (models/neus_hint_model.py:359-372, :527-528) against the reference's recorded run: the
    tracer on its own (nrh_sphere_trace through NeuSHintRenderer.sphere_trace) and the evaluation render whose hit points,
    shadow rays and specular cue come from it.  Rays that hit agree to the convergence threshold's order (a ray may stop one
    iteration apart when |sdf| sits at 1e-4); rays that miss run on to depth > 100 with ~50-long last steps, compared relatively."""
    g = load_npz("render_branches_b.npz")
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    model = _model(scene_states["b"], prec, cfg=na.NeuSModelConfig(renderer=na.NeuSRendererConfig(depth_type=na.DepthComputationType.SphereTracing)))
    pts, dep = model.sphere_trace(rb.origins, rb.directions, 2000, 1e-4, 100.0)
    want_d, want_p = g["st.trace_depths"], g["st.trace_pts"]
    hit = want_d[:, 0] < 100.0
    np.testing.assert_allclose(dep.cpu().numpy()[hit], want_d[hit], rtol=0, atol=3e-4)
    np.testing.assert_allclose(pts.cpu().numpy()[hit], want_p[hit], rtol=0, atol=3e-4)
    np.testing.assert_allclose(dep.cpu().numpy()[~hit], want_d[~hit], rtol=2e-3, atol=0)
    assert bool((dep.cpu().numpy()[~hit] > 100.0).all())
    # zero iterations: the origins, depth 0
    p0, d0 = model.sphere_trace(rb.origins, rb.directions, 0, 1e-4, 100.0)
    assert torch.equal(p0, rb.origins) and float(d0.abs().max()) == 0.0
    with torch.no_grad():
        out = model(rb, background_rgb=torch.ones(1, 3).cuda())
    got_d = out.depth.cpu().numpy()
    np.testing.assert_allclose(got_d[hit], g["st.depth"][hit], rtol=0, atol=3e-4)
    np.testing.assert_allclose(got_d[~hit], g["st.depth"][~hit], rtol=2e-3, atol=0)
    # the hints are driven by the traced hit point: compare where the ray hit (a miss puts the "hit point" > 100 away, where the
    # shadow ray's geometry amplifies the last-step difference)
    np.testing.assert_allclose(out.visibilities.cpu().numpy()[hit], g["st.visibilities"][hit], rtol=0, atol=3e-3)
    np.testing.assert_allclose(out.specular_cue.cpu().numpy()[hit], g["st.specular_cue"][hit], rtol=0, atol=3e-4)
    np.testing.assert_allclose(out.rgb.cpu().numpy()[hit], g["st.rgb"][hit], rtol=0, atol=1e-4)
    assert np.abs(out.rgb.cpu().numpy() - g["st.rgb"]).max() < 2e-3
    # training runs through the same C path (depth is not differentiated, :359); a captured graph refuses loudly
    model.train()
    o = model(rb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=1000)
    o.rgb.sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_one_hint_models_and_force_flags(scene_states, prec):
    """shadow_hint without specular_hint and the reverse (the kernels of the two-hint model with the missing hint's first-layer
    columns zero), and force_shadow_map / force_specular_cue: evaluation against the reference's recorded outputs, one training
    step's loss and gradients against its float64 run with bounds derived from its own float32 run (conftest.grad_bound)."""
    from nrhints_amd.synthetic import one_hint_state
    from nrhints_amd.training import train_loss_dict
    g = load_npz("render_branches_b.npz")
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    R, sb = na.NeuSRendererConfig, scene_states["b"]
    cases = {"sho": (R(shadow_hint=True, specular_hint=False), one_hint_state(sb, True)),
             "spo": (R(shadow_hint=False, specular_hint=True), one_hint_state(sb, False)),
             "frc": (R(force_shadow_map=True, force_specular_cue=True), sb)}
    bg = torch.ones(1, 3).cuda()
    for vt, (rcfg, st) in cases.items():
        model = _model(st, prec, cfg=na.NeuSModelConfig(renderer=rcfg))
        with torch.no_grad():
            out = model(rb, background_rgb=bg)
        np.testing.assert_allclose(out.rgb.cpu().numpy(), g[f"{vt}.rgb"], rtol=0, atol=1e-4, err_msg=vt)
        np.testing.assert_allclose(out.depth.cpu().numpy(), g[f"{vt}.depth"], rtol=0, atol=3e-4, err_msg=vt)
        if vt == "spo":
            assert out.visibilities is None
        else:
            np.testing.assert_allclose(out.visibilities.cpu().numpy(), g[f"{vt}.visibilities"], rtol=0, atol=3e-3, err_msg=vt)
        if vt == "sho":
            assert out.specular_cue is None
        else:
            np.testing.assert_allclose(out.specular_cue.cpu().numpy(), g[f"{vt}.specular_cue"], rtol=0, atol=3e-4, err_msg=vt)
    for bad in (R(shadow_hint=False, specular_hint=False, force_shadow_map=True), R(shadow_hint=False, specular_hint=False, force_specular_cue=True)):
        with pytest.raises(ValueError, match="fails in the reference itself"):     # recorded: RuntimeError on the lin0 shape
            na.NeuSHintRenderer(na.NeuSModelConfig(renderer=bad))
    assert str(g["force_shadow_only.outcome"]).startswith("RuntimeError")
    # one training step per one-hint model
    tb = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    for vt, shadow in (("sho", True), ("spo", False)):
        rcfg, st = cases[vt]
        model = _model(st, prec, cfg=na.NeuSModelConfig(renderer=rcfg), train=True)
        out = model(tb, is_training=True, background_rgb=bg, global_step=int(g["t.global_step"]),
                    _t_rand_primary=cu(g[f"{vt}.t_rand_primary"]), _t_rand_shadow=cu(g[f"{vt}.t_rand_shadow"]) if shadow else None)
        np.testing.assert_allclose(out.rgb.detach().cpu().numpy(), g[f"{vt}.t.rgb"], rtol=0, atol=1e-4)
        ld = train_loss_dict(out, cu(g["t.rgb_gt"]), 0.1)
        np.testing.assert_allclose(float(ld["loss"]), float(g[f"{vt}.loss"]), rtol=2e-4)
        ld["loss"].backward()
        named = dict(model.named_parameters())
        keys = [k for k in g if k.startswith(f"{vt}.grad.") and ".rays." not in k]
        assert len(keys) == 11
        for k in keys:
            name = k[len(vt) + 6:]
            want64 = g[k.replace(".grad.", ".grad64.")]
            bound, scale = grad_bound(g[k], want64, factor=4.0, floor=5e-3)      # 32 rays: one coarse draw of the reference's own noise
            got = named[name].grad.detach().cpu().numpy().astype(np.float64)
            assert got.shape == want64.shape, (vt, name)
            err = float(np.abs(got - want64).max())
            assert err <= bound, (vt, name, err, bound, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_free_scalars_vs_reference(scene_states, prec):
    """renderer.specular_roughness / shadow_ray_offset off their defaults (models/neus_hint_model.py:161, :163): kernel constants
    carried by NrhNet (custom_consts), not compiled shapes - evaluation render and one training step (autograd path and the fused
    step) against the reference's recorded run (tests/golden/render_consts_b.npz)."""
    from nrhints_amd import train_fused
    from nrhints_amd.training import train_loss_dict
    g = load_npz("render_consts_b.npz")
    rcfg = na.NeuSRendererConfig(specular_roughness=[float(x) for x in g["specular_roughness"]], shadow_ray_offset=float(g["shadow_ray_offset"]))
    cfg = na.NeuSModelConfig(renderer=rcfg)
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    bg = torch.ones(1, 3).cuda()
    model = _model(scene_states["b"], prec, cfg=cfg)
    with torch.no_grad():
        out = model(rb, background_rgb=bg)
    np.testing.assert_allclose(out.rgb.cpu().numpy(), g["rc.rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.depth.cpu().numpy(), g["rc.depth"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(out.visibilities.cpu().numpy(), g["rc.visibilities"], rtol=0, atol=3e-3)
    # the cue reaches 1.7 at these roughness values and the reference's own float32 run is 4.6e-4 away from its float64 run on it
    # (the hit normal is a weighted sum over samples placed at fp32 noise): compare with the float64 record, 3x that distance
    cue_tol = max(2e-4, 3.0 * float(np.abs(g["rc.specular_cue"] - g["rc64.specular_cue"]).max()))
    np.testing.assert_allclose(out.specular_cue.cpu().numpy(), g["rc64.specular_cue"], rtol=0, atol=cue_tol)
    # ... and the default model on the same rays gives different hints (the fixture discriminates)
    with torch.no_grad():
        dflt = _model(scene_states["b"], prec)(rb, background_rgb=bg)
    assert np.abs(dflt.specular_cue.cpu().numpy() - g["rc.specular_cue"]).max() > 1e-2
    tb = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    keys = [k for k in g if k.startswith("rc.grad.") and ".rays." not in k]
    assert len(keys) == 8
    for path in ("autograd", "fused"):
        model = _model(scene_states["b"], prec, cfg=cfg, train=True)
        jit = dict(t_rand_primary=cu(g["rc.t_rand_primary"]), t_rand_shadow=cu(g["rc.t_rand_shadow"]))
        if path == "autograd":
            out = model(tb, is_training=True, background_rgb=bg, global_step=int(g["t.global_step"]), _t_rand_primary=jit["t_rand_primary"],
                        _t_rand_shadow=jit["t_rand_shadow"])
            np.testing.assert_allclose(out.rgb.detach().cpu().numpy(), g["rc.t.rgb"], rtol=0, atol=1e-4)
            ld = train_loss_dict(out, cu(g["t.rgb_gt"]), 0.1)
            loss = float(ld["loss"])
            ld["loss"].backward()
        else:
            loss = float(train_fused.train_step_backward(model, tb, cu(g["t.rgb_gt"]), bg, int(g["t.global_step"]), **jit)[0])
        np.testing.assert_allclose(loss, float(g["rc.loss"]), rtol=2e-4)
        named = dict(model.named_parameters())
        for k in keys:
            name = k[len("rc.grad."):]
            want64 = g[k.replace(".grad.", ".grad64.")]
            bound, scale = grad_bound(g[k], want64, factor=4.0, floor=5e-3)      # 32 rays: one coarse draw of the reference's own noise
            err = float(np.abs(named[name].grad.detach().cpu().numpy().astype(np.float64) - want64).max())
            assert err <= bound, (path, name, err, bound, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("vt", ["shg", "spg", "bhg"])
def test_hint_gradients_vs_reference(scene_states, vt):
    """renderer.shadow_hint_gradient / specular_hint_gradient / both (models/neus_hint_model.py:379, :589): the hints stay inside
    the graph - the visibility through the SDF network at the shadow ray's 128 sections (same HIP forward / backward sweeps as
    the primary samples), the cue through the hit normal.  One training step: forward values, loss and the recorded gradient
    tensors against the reference's float64 run, bounds from its own float32 run; and the gradients differ from the
    hint-constant model's by what the fixture says they should."""
    from nrhints_amd.training import train_loss_dict
    g = load_npz("render_branches_b.npz")
    R = na.NeuSRendererConfig
    rcfg = R(shadow_hint_gradient=vt in ("shg", "bhg"), specular_hint_gradient=vt in ("spg", "bhg"))
    tb = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    bg = torch.ones(1, 3).cuda()

    def step(cfg):
        model = _model(scene_states["b"], "f16x3", cfg=na.NeuSModelConfig(renderer=cfg), train=True)
        out = model(tb, is_training=True, background_rgb=bg, global_step=int(g["t.global_step"]),
                    _t_rand_primary=cu(g[f"{vt}.t_rand_primary"]), _t_rand_shadow=cu(g[f"{vt}.t_rand_shadow"]))
        ld = train_loss_dict(out, cu(g["t.rgb_gt"]), 0.1)
        ld["loss"].backward()
        return out, ld, dict(model.named_parameters())

    out, ld, named = step(rcfg)
    np.testing.assert_allclose(out.rgb.detach().cpu().numpy(), g[f"{vt}.t.rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(float(ld["loss"]), float(g[f"{vt}.loss"]), rtol=2e-4)
    keys = [k for k in g if k.startswith(f"{vt}.grad.") and ".rays." not in k]
    assert len(keys) == 11
    for k in keys:
        name = k[len(vt) + 6:]
        want64 = g[k.replace(".grad.", ".grad64.")]
        bound, scale = grad_bound(g[k], want64, factor=4.0, floor=5e-3)
        err = float(np.abs(named[name].grad.detach().cpu().numpy().astype(np.float64) - want64).max())
        assert err <= bound, (vt, name, err, bound, scale)
    # the same step with the hints as constants gives visibly different gradients: the cue's gradient moves the first SDF layer's
    # by ~10 %, the visibility's is what makes d loss / d variance 1e-4 instead of 1e-6 here (the fixture's own numbers)
    _, _, plain = step(R())
    if vt == "shg":
        a, b = named["deviation_network.variance"].grad, plain["deviation_network.variance"].grad
        assert abs(float(a - b)) > 0.5 * abs(float(a))
    else:
        a, b = named["sdf_network.lin0.weight_v"].grad, plain["sdf_network.lin0.weight_v"].grad
        assert float((a - b).abs().max()) > 1e-2 * float(b.abs().max())
    if vt == "shg":
        with pytest.raises(NotImplementedError):
            model = _model(scene_states["b"], "f16x3", cfg=na.NeuSModelConfig(renderer=rcfg), train=True)
            rays = na.RayBundle(origins=tb.origins, directions=tb.directions, pl_positions=tb.pl_positions.clone().requires_grad_(True),
                                nears=tb.nears, fars=tb.fars)
            model(rays, is_training=True, background_rgb=bg, global_step=100)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_partial_visibility_hint(scene_states, prec):
    """renderer.n_shadow_importance_clip = 8 (models/neus_hint_model.py:553-575): a shadow ray per group of 16 samples, aimed at the
    group's first sample position; the reflectance net reads the group's visibility, the shadow map is the group value at the
    maximal-weight sample.  Evaluation against the reference's recorded run (which differs from the hit-point mode's by up to 1.0
    in the shadow map and 8e-4 in rgb on these rays), chunked evaluation == unchunked, and one training step's loss and
    gradients (shadow jitter [N * 8, 64] in the reference's draw order)."""
    from nrhints_amd.training import train_loss_dict
    g = load_npz("render_branches_b.npz")
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    cfg = na.NeuSModelConfig(renderer=na.NeuSRendererConfig(n_shadow_importance_clip=8))
    bg = torch.ones(1, 3).cuda()
    model = _model(scene_states["b"], prec, cfg=cfg)
    with torch.no_grad():
        out = model(rb, background_rgb=bg)
        prod = model.render_products(rb, bg)
    np.testing.assert_allclose(out.rgb.cpu().numpy(), g["psh.rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(prod["rgb"].cpu().numpy(), g["psh.rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.depth.cpu().numpy(), g["psh.depth"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(out.specular_cue.cpu().numpy(), g["psh.specular_cue"], rtol=0, atol=3e-4)
    dv = np.abs(out.visibilities.cpu().numpy() - g["psh.visibilities"])
    # the shadow map picks the group of the arg-max weight: two near-equal weights in different groups may swap (as MaximalWeightPoint)
    assert np.mean(dv > 3e-3) <= 0.05, np.sort(dv.ravel())[-5:]
    assert np.abs(g["psh.visibilities"] - g["frc.visibilities"]).max() > 0.5      # the fixture tells the two modes apart
    model.max_chunk_rays = 24
    with torch.no_grad():
        out2 = model(rb, background_rgb=bg)
    assert torch.equal(out2.rgb, out.rgb) and torch.equal(out2.visibilities, out.visibilities)
    # one training step
    tb = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    model = _model(scene_states["b"], prec, cfg=cfg, train=True)
    o = model(tb, is_training=True, background_rgb=bg, global_step=int(g["t.global_step"]),
              _t_rand_primary=cu(g["psh.t_rand_primary"]), _t_rand_shadow=cu(g["psh.t_rand_shadow"]))
    np.testing.assert_allclose(o.rgb.detach().cpu().numpy(), g["psh.t.rgb"], rtol=0, atol=1e-4)
    ld = train_loss_dict(o, cu(g["t.rgb_gt"]), 0.1)
    np.testing.assert_allclose(float(ld["loss"]), float(g["psh.loss"]), rtol=2e-4)
    ld["loss"].backward()
    named = dict(model.named_parameters())
    keys = [k for k in g if k.startswith("psh.grad.") and ".rays." not in k]
    assert len(keys) == 11
    for k in keys:
        name = k[len("psh") + 6:]
        want64 = g[k.replace(".grad.", ".grad64.")]
        # Group targets sit ON the primary ray's samples, many of them at the surface, where the transmittance of the shadow ray
        # swings between 0 and 1 within |delta sdf| ~ 1 / inv_s = 1e-3: a 1e-6 difference in the SDF arithmetic (HIP kernels vs
        # torch on the CPU) moves single group visibilities by 1e-3 (the evaluation check above allows that) and, through the
        # reflectance net, the parameter gradients by up to a few per cent of their scale - far more than the reference's own
        # float32-vs-float64 distance happens to be on this draw (measured: up to 3.4 % of the tensor's scale in f32 mode, 9.8 % in
        # f16x3 mode on the first reflectance layer).  Hence a direction + magnitude check against the REFERENCE's record here; the
        # arithmetic itself is held to a float32 bound right below, against the oracle evaluated on this path's own sample placement.
        got = named[name].grad.detach().cpu().numpy().astype(np.float64)
        scale = max(float(np.abs(want64).max()), 1e-30)
        err = float(np.abs(got - want64).max())
        assert err <= 0.15 * scale, (name, err, scale)
        if got.size > 1:
            cos = float((got * want64).sum() / (np.linalg.norm(got) * np.linalg.norm(want64) + 1e-300))
            assert cos > 0.995, (name, cos)
    # ... and the same gradients with PLACEMENT taken out of the comparison: the oracle in float64 on the HIP
    # path's own non-differentiable products - sample positions, group visibilities, cue (the reference keeps all three outside
    # its graph, :697, :553-575, :589) - differentiates exactly what the HIP backward differentiated, so the bound goes back to a
    # float32-arithmetic one: 2e-3 of the tensor's scale for the weight matrices, 1e-2 for the small tensors whose entries are single
    # sums over all 4 096 samples that cancel (biases, weight_g, the two scalars: measured up to 5e-3 on d loss / d out_sdf.bias, in
    # BOTH precision modes - float32 round-off of the summands, not the f16x3 split).  Both are below the reference's OWN
    # float32-vs-float64 distance on these layers (1-3 %, conftest.grad_bound) and 15-75 x tighter than the check above
    f32 = lambda t: t.detach().float().contiguous()
    res = model._render_train(f32(tb.origins), f32(tb.directions), f32(tb.pl_positions), f32(tb.nears).reshape(-1), f32(tb.fars).reshape(-1),
                              min(1.0, int(g["t.global_step"]) / model.config.anneal_end), cu(g["psh.t_rand_primary"]).reshape(-1),
                              cu(g["psh.t_rand_shadow"]), 0)
    n = tb.origins.shape[0]
    mid, dist = res["mid_z"].double().cpu(), res["dists"].double().cpu()
    z_hip = mid - 0.5 * dist
    leaves = {k: T(np.asarray(v)).double().clone().requires_grad_(True) for k, v in scene_states["b"].items()}
    o64 = orc.render_forward(orc.params_from_state(leaves, torch.float64), *(T(g["t." + k]).double() for k in ("o", "d", "pl", "near", "far")),
                             background_rgb=torch.ones(1, 3, dtype=torch.float64), is_training=True, global_step=int(g["t.global_step"]),
                             t_rand_primary=T(g["psh.t_rand_primary"]).double(), t_rand_shadow=T(g["psh.t_rand_shadow"]).double(),
                             mode="as_written", differentiable=True, n_shadow_importance_clip=8, z_override=z_hip,
                             vis_groups_override=res["vis_groups"].double().cpu().reshape(n, 8), cue_override=res["cue"][:, 0, :].double().cpu())
    np.testing.assert_allclose(o.rgb.detach().cpu().numpy(), o64["rgb"].detach().numpy(), rtol=0, atol=2e-5)
    loss64, _, _ = orc.train_loss(o64, T(g["t.rgb_gt"]).double())
    loss64.backward()
    report = []
    for name, prm in named.items():
        want = leaves[name].grad.numpy()
        scale = max(float(np.abs(want).max()), 1e-30)
        err = float(np.abs(prm.grad.detach().cpu().numpy().astype(np.float64) - want).max())
        report.append((err / scale, name, want.size))
    report.sort(reverse=True)
    print("placement-free gradient errors (err / scale, tensor, entries):", report[:8])
    # matrices (>= 256 entries) and everything else (biases, the two scalars: single sums that cancel)
    assert all(r[0] <= 2e-3 for r in report if r[2] >= 1024), report[:8]
    assert all(r[0] <= 1e-2 for r in report), report[:8]
    with pytest.raises(ValueError):
        na.NeuSHintRenderer(na.NeuSModelConfig(renderer=na.NeuSRendererConfig(n_shadow_importance_clip=3)))


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_no_importance_samples_plumbing_variant(scene_states, prec):
    """renderer.n_importance_samples = 0 with both hints off - BASELINE configs[0]'s plumbing variant (4096 rays x 64
    samples; models/neus_hint_model.py:696 skips the hierarchical sampling).  The kernels keep 128 slots per ray and give the upper
    64 weight exactly 0: every output has the reference's 64-sample shape and values, at the fixture's 64 rays and at the config's
    4096 (properties), and one training step's loss and gradients match the reference's."""
    from nrhints_amd.synthetic import naive_state
    from nrhints_amd.training import train_loss_dict
    g = load_npz("render_branches_b.npz")
    cfg = na.NeuSModelConfig(renderer=na.NeuSRendererConfig(n_importance_samples=0, shadow_hint=False, specular_hint=False))
    st = naive_state(scene_states["b"])
    bg = torch.ones(1, 3).cuda()
    model = _model(st, prec, cfg=cfg)
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    with torch.no_grad():
        out = model(rb, background_rgb=bg)
    assert out.weights.shape == (64, 64) and out.inside_sphere.shape == (64, 64) and out.s_val.shape == (64, 64)
    assert out.analytic_normals.shape == (64, 64, 3) and out.visibilities is None and out.specular_cue is None
    np.testing.assert_allclose(out.rgb.cpu().numpy(), g["i0.rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.depth.cpu().numpy(), g["i0.depth"], rtol=0, atol=3e-4)
    np.testing.assert_allclose(out.weights.cpu().numpy(), g["i0.weights"], rtol=0, atol=3e-4)
    assert np.array_equal(out.inside_sphere.cpu().numpy(), g["i0.inside_sphere"])
    dn = np.abs(out.normalized_analytic_normals.cpu().numpy() - g["i0.normalized_analytic_normals"])
    assert dn.mean() < 2e-5 and dn.max() < 5e-3
    np.testing.assert_allclose(out.s_val.cpu().numpy(), g["i0.s_val"], rtol=1e-5)
    # the config's size: 4096 rays x 64 samples
    big = _bundle(*make_rays(4096, seed=4, spread=0.12))
    with torch.no_grad():
        ob = model(big, background_rgb=bg)
        prod = model.render_products(big, bg)
    assert ob.weights.shape == (4096, 64) and bool(torch.isfinite(ob.rgb).all())
    assert bool((ob.weights >= 0).all()) and bool((ob.weights.sum(-1) <= 1.0 + 1e-4).all())
    assert torch.equal(prod["rgb"], ob.rgb)
    # one training step
    tb = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    model = _model(st, prec, cfg=cfg, train=True)
    o = model(tb, is_training=True, background_rgb=bg, global_step=int(g["t.global_step"]), _t_rand_primary=cu(g["i0.t_rand_primary"]))
    assert o.weights.shape == (32, 64) and o.analytic_normals.shape == (32, 64, 3)
    np.testing.assert_allclose(o.rgb.detach().cpu().numpy(), g["i0.t.rgb"], rtol=0, atol=1e-4)
    ld = train_loss_dict(o, cu(g["t.rgb_gt"]), 0.1)
    np.testing.assert_allclose(float(ld["loss"]), float(g["i0.loss"]), rtol=2e-4)
    ld["loss"].backward()
    named = dict(model.named_parameters())
    keys = [k for k in g if k.startswith("i0.grad.") and ".rays." not in k]
    assert len(keys) == 11
    for k in keys:
        name = k[len("i0") + 6:]
        want64 = g[k.replace(".grad.", ".grad64.")]
        bound, scale = grad_bound(g[k], want64, factor=4.0, floor=5e-3)
        err = float(np.abs(named[name].grad.detach().cpu().numpy().astype(np.float64) - want64).max())
        assert err <= bound, (name, err, bound, scale)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_outside_network_kernels_vs_reference_unit(prec):
    """The background network on its own (csrc/nrh_outside.hip through outside.OutsideNetHip): values against the unit I/O the
    reference's NeRF module recorded (tests/golden/outside_b.npz: unit.*), and the whole backward - the adjoint sweep, the 13
    weight-gradient jobs, the chain rule through both encodings - against float64 autograd of the oracle's restatement
    (oracle.neus_oracle.nerf_forward, itself pinned on the same unit vectors) for all 24 parameter tensors and the three inputs."""
    from nrhints_amd.outside import OutsideNeRF
    g = load_npz("outside_b.npz")
    ref = {k[5:]: T(v) for k, v in g.items() if k.startswith("nerf.")}
    nerf = OutsideNeRF()
    nerf.load_state_dict(ref)
    nerf = nerf.cuda()
    nerf.precision = prec
    pts4, views, pls = cu(g["unit.pts4"]), cu(g["unit.views"]), cu(g["unit.pls"])
    with torch.no_grad():
        dens, col = nerf(pts4, views, pls)
    # 10 ReLU layers of K <= 340 in fp32-equivalent arithmetic (the reference's own record is float32)
    np.testing.assert_allclose(dens.cpu().numpy(), g["unit.density"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(col.cpu().numpy(), g["unit.rgb"], rtol=0, atol=1e-5)
    # backward: 32 rays x 5 points (160 points, a multiple of 16), per-ray view / light rows
    rs = np.random.RandomState(5)
    P, ppr = 160, 5
    x = rs.randn(P, 3).astype(np.float32)
    r = np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), 1.0) * (1.0 + rs.rand(P, 1).astype(np.float32))
    p4 = np.concatenate([x / np.linalg.norm(x, axis=-1, keepdims=True), 1.0 / r], axis=-1).astype(np.float32)
    vw = rs.randn(P // ppr, 3).astype(np.float32); vw /= np.linalg.norm(vw, axis=-1, keepdims=True)
    lg = (4.0 * rs.randn(P // ppr, 3)).astype(np.float32)
    wd, wc = rs.randn(P, 1).astype(np.float32), rs.randn(P, 3).astype(np.float32)
    a, b, c = (cu(v).requires_grad_(True) for v in (p4, vw, lg))
    dens, col = nerf(a, b, c, pts_per_ray=ppr)
    ((dens * cu(wd)).sum() + (col * cu(wc)).sum()).backward()
    ref64 = {k: v.double().clone().requires_grad_(True) for k, v in ref.items()}
    a64, b64, c64 = (T(v).double().requires_grad_(True) for v in (p4, vw, lg))
    d64, c_64 = orc.nerf_forward(ref64, a64, b64.repeat_interleave(ppr, 0), c64.repeat_interleave(ppr, 0))
    ((d64 * T(wd).double()).sum() + (c_64 * T(wc).double()).sum()).backward()
    np.testing.assert_allclose(dens.detach().cpu().numpy(), d64.detach().numpy(), rtol=0, atol=2e-5)
    for name, p in nerf.named_parameters():
        want = ref64[name].grad.numpy()
        scale = max(float(np.abs(want).max()), 1e-12)
        err = float(np.abs(p.grad.cpu().numpy() - want).max())
        assert err <= 2e-5 * scale + 1e-7, (name, err, scale)      # bf16x3 split-K products: 4e-6 of an entry's magnitude (nrh_dw.hip)
    for got, want in ((a.grad, a64.grad), (b.grad, b64.grad), (c.grad, c64.grad)):
        scale = max(float(want.abs().max()), 1e-12)
        assert float((got.cpu().double() - want).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_outside_nerf_background(scene_states, prec):
    """renderer.use_outside_nerf (models/neus_hint_model.py:434-473, :516-519, :630-633, :677-724) against the reference's recorded
    run: evaluation (rgb, depth, the 160 weights per ray - 128 blended + 32 beyond the sphere -, visibility; the per-pixel products
    path) and one training step (loss; gradients of the renderer AND of the background network, float64 reference, bounds from its
    float32 run).  NeuS side in the HIP kernels with the alpha blend inside the alpha stage, the NeRF MLP as its own kernel pair
    (csrc/nrh_outside.hip) with its weight gradients through nrh_dw_gemm."""
    from nrhints_amd.training import train_loss_dict
    g = load_npz("outside_b.npz")
    cfg = na.NeuSModelConfig(renderer=na.NeuSRendererConfig(use_outside_nerf=True))
    state = {k: T(np.asarray(v)) for k, v in scene_states["b"].items()}
    state.update({"outside_nerf." + k[5:]: T(v) for k, v in g.items() if k.startswith("nerf.")})
    bg = torch.ones(1, 3).cuda()

    def build(train=False):
        m = na.NeuSHintRenderer(cfg, precision=prec)
        m.load_state_dict(state)
        m = m.cuda()
        return m if train else m.eval()

    model = build()
    rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
    with torch.no_grad():
        out = model(rb, background_rgb=bg)
        prod = model.render_products(rb, bg)
    assert out.weights.shape == (64, 160) and out.inside_sphere.shape == (64, 128)
    np.testing.assert_allclose(out.rgb.cpu().numpy(), g["eval.rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(prod["rgb"].cpu().numpy(), g["eval.rgb"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(out.depth.cpu().numpy(), g["eval.depth"], rtol=0, atol=3e-4)
    dw = np.abs(out.weights.cpu().numpy() - g["eval.weights"])
    assert dw.mean() < 3e-5 and dw.max() < 5e-3, (dw.mean(), dw.max())
    assert float(out.weights[:, 128:].sum(-1).mean()) > 0.05                   # the background is visible on these rays
    np.testing.assert_allclose(out.visibilities.cpu().numpy(), g["eval.visibilities"], rtol=0, atol=3e-3)
    assert np.array_equal(out.inside_sphere.cpu().numpy(), g["eval.inside_sphere"])
    # chunked == unchunked
    model.max_outside_rays = 24
    with torch.no_grad():
        out2 = model(rb, background_rgb=bg)
    assert float((out2.rgb - out.rgb).abs().max()) < 2e-6 and out2.weights.shape == (64, 160)
    # one training step
    model = build(train=True)
    tb = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    o = model(tb, is_training=True, background_rgb=bg, global_step=int(g["t.global_step"]), _t_rand_primary=cu(g["t.t_rand_primary"]),
              _t_rand_shadow=cu(g["t.t_rand_shadow"]), _t_rand_outside=cu(g["t.t_rand_outside"]))
    np.testing.assert_allclose(o.rgb.detach().cpu().numpy(), g["t.rgb"], rtol=0, atol=1e-4)
    dwt = np.abs(o.weights.detach().cpu().numpy() - g["t.weights"])
    assert dwt.mean() < 3e-5 and dwt.max() < 5e-3
    ld = train_loss_dict(o, cu(g["t.rgb_gt"]), 0.1)
    np.testing.assert_allclose(float(ld["loss"]), float(g["t.loss"]), rtol=2e-4)
    ld["loss"].backward()
    named = dict(model.named_parameters())
    keys = [k for k in g if k.startswith("t.grad.") and ".rays." not in k]
    assert len(keys) == 14 and sum("outside_nerf" in k for k in keys) == 7
    for k in keys:
        name = k[len("t.grad."):]
        want64 = g[k.replace("t.grad.", "t.grad64.")]
        # (the variance gradient is ONE number, a sum over 64 x 160 samples that cancels to 3.5e-6: the reference's own float32 error
        # on it - 0.15 % here - is a single draw of the placement noise, not a yardstick with a maximum over many entries behind
        # it; the two precision modes and the two sampler kernels land between 0.2 % and 0.8 %.  Floor 1 % for one-entry tensors)
        bound, scale = grad_bound(g[k], want64, factor=4.0, floor=1e-2 if np.size(want64) == 1 else 5e-3)
        got = named[name].grad.detach().cpu().numpy().astype(np.float64)
        err = float(np.abs(got - want64).max())
        assert err <= bound, (name, err, bound, scale)
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in model.parameters())


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_fused_step_outside_nerf(scene_states, prec):
    """renderer.use_outside_nerf on the autograd-free step (VERDICT r5 missing #3; train_fused._outside_forward / _outside_loss /
    _outside_backward): same batch, same three jitter draws as the autograd path of test_outside_nerf_background (itself held to the
    reference's float64 gradients) - loss and all 70 gradient tensors, the background network's 24 included, to float32 round-off;
    the gradients are views of the flat buffer; and the step is captured and replayed by GraphedTrainStep."""
    from nrhints_amd import train_fused
    from nrhints_amd.training import GraphedTrainStep, train_loss_dict
    g = load_npz("outside_b.npz")
    cfg = na.NeuSModelConfig(renderer=na.NeuSRendererConfig(use_outside_nerf=True))
    state = {k: T(np.asarray(v)) for k, v in scene_states["b"].items()}
    state.update({"outside_nerf." + k[5:]: T(v) for k, v in g.items() if k.startswith("nerf.")})
    bg = torch.ones(1, 3).cuda()

    def build():
        m = na.NeuSHintRenderer(cfg, precision=prec)
        m.load_state_dict(state)
        return m.cuda().train()

    tb = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    gs, gt = int(g["t.global_step"]), cu(g["t.rgb_gt"])
    tp, ts, to = cu(g["t.t_rand_primary"]), cu(g["t.t_rand_shadow"]), cu(g["t.t_rand_outside"])
    ref = build()
    o = ref(tb, is_training=True, background_rgb=bg, global_step=gs, _t_rand_primary=tp, _t_rand_shadow=ts, _t_rand_outside=to)
    ld = train_loss_dict(o, gt, 0.1)
    ld["loss"].backward()
    fused = build()
    assert train_fused.supported(fused, tb) is None
    l8 = train_fused.train_step_backward(fused, tb, gt, bg, gs, t_rand_primary=tp, t_rand_shadow=ts, t_rand_outside=to)
    for i, k in enumerate(("loss", "rgb_loss", "eikonal_loss", "s_val", "psnr")):
        np.testing.assert_allclose(float(l8[i]), float(ld[k].detach()), rtol=2e-5, err_msg=k)
    np.testing.assert_allclose(float(l8[0]), float(g["t.loss"]), rtol=2e-4)
    n_out = 0
    for (name, pa), (_, pf) in zip(ref.named_parameters(), fused.named_parameters()):
        assert pf.grad is not None and pf.grad.shape == pa.shape and hasattr(pf.grad, "_nrh_flat"), name
        scale = float(pa.grad.abs().max()) + 1e-30
        assert float((pa.grad - pf.grad).abs().max()) < 1e-4 * scale + 5e-6, (name, float((pa.grad - pf.grad).abs().max()), scale)
        n_out += name.startswith("outside_nerf.")
    assert n_out == 24
    # a second call overwrites (zero_grad + backward semantics), it does not accumulate
    l8b = train_fused.train_step_backward(fused, tb, gt, bg, gs, t_rand_primary=tp, t_rand_shadow=ts, t_rand_outside=to)
    for (name, pa), (_, pf) in zip(ref.named_parameters(), fused.named_parameters()):
        scale = float(pa.grad.abs().max()) + 1e-30
        assert float((pa.grad - pf.grad).abs().max()) < 1e-4 * scale + 5e-6, name
    # pose / light refinement with the background on the fused step: the background network's own dependence on the rays (sample
    # points, view and light inputs) joins nrh_ray_adjoint's; against the reference's recorded float64 ray gradients
    rb2 = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    for t_ in (rb2.origins, rb2.directions, rb2.pl_positions):
        t_.requires_grad_(True)
    assert train_fused.supported(fused, rb2) is None
    rg = {}
    train_fused.train_step_backward(fused, rb2, gt, bg, gs, t_rand_primary=tp, t_rand_shadow=ts, t_rand_outside=to, ray_grads=rg)
    for nm in ("origins", "directions", "pl_positions"):
        want64 = g[f"t.grad64.rays.{nm}"]
        bound, scale = grad_bound(g[f"t.grad.rays.{nm}"], want64, factor=4.0, floor=5e-3)
        err = float(np.abs(rg[nm].detach().cpu().numpy().astype(np.float64) - want64).max())
        assert err <= bound, (nm, err, bound, scale)
    # ... and pushed into the graph behind the bundle when no dict is given (here: leaves), the far bound included
    rb3 = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    for t_ in (rb3.origins, rb3.directions, rb3.pl_positions, rb3.fars):
        t_.requires_grad_(True)
    train_fused.train_step_backward(fused, rb3, gt, bg, gs, t_rand_primary=tp, t_rand_shadow=ts, t_rand_outside=to)
    assert float((rb3.origins.grad - rg["origins"]).abs().max()) < 1e-6 * float(rg["origins"].abs().max()) + 1e-9
    assert rb3.fars.grad is not None and rb3.fars.grad.shape == rb3.fars.shape and float(rb3.fars.grad.abs().max()) > 0.0
    # d loss / d far: against the autograd path, where the far bound keeps its graph through outside_z (:689-693)
    rb4 = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    for t_ in (rb4.origins, rb4.directions, rb4.pl_positions, rb4.fars):
        t_.requires_grad_(True)
    ref2 = build()
    o4 = ref2(rb4, is_training=True, background_rgb=bg, global_step=gs, _t_rand_primary=tp, _t_rand_shadow=ts, _t_rand_outside=to)
    train_loss_dict(o4, gt, 0.1)["loss"].backward()
    for a_, b_ in ((rb3.fars.grad, rb4.fars.grad), (rb3.origins.grad, rb4.origins.grad), (rb3.directions.grad, rb4.directions.grad),
                   (rb3.pl_positions.grad, rb4.pl_positions.grad)):
        assert float((a_ - b_).abs().max()) < 2e-4 * float(b_.abs().max()) + 1e-8, (float((a_ - b_).abs().max()), float(b_.abs().max()))
    # captured
    cap = build()
    n = tb.origins.shape[0]
    step = GraphedTrainStep(cap, n, bg, lr=5e-4, warm_up_end=20, global_step=gs)
    assert step._use_fused
    before = {k: v.detach().clone() for k, v in cap.named_parameters()}
    losses = [step(tb, gt, global_step=gs + i)["loss"] for i in range(4)]
    assert all(np.isfinite(losses)) and abs(losses[0] - float(ld["loss"].detach())) < 0.05 * abs(float(ld["loss"].detach()))
    moved = {k: float((p.detach() - before[k]).abs().max()) for k, p in cap.named_parameters()}
    assert min(moved[k] for k in moved if k.startswith("outside_nerf.") and k.endswith("weight")) > 0.0 and moved["sdf_network.lin3.weight_v"] > 0.0
    step.release()


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_fused_step_outside_nerf_with_hint_gradients(scene_states, prec):
    """use_outside_nerf + shadow_hint_gradient + specular_hint_gradient (VERDICT r5 missing #4; models/neus_hint_model.py:379,
    :516-519, :586-589): on the fused step the background island and the hint island meet in the weights' adjoint.  Against the
    reference's recorded float64 step (tests/golden/make_golden_outside_hintgrad.py: the variance gradient is 5x what it is without
    the hints, the SDF net's up to 9 % different - the bounds below would catch a dropped term); forward() + backward() refuses
    the pair with a pointer to the fused step; training.train_step takes the fused step by itself."""
    from nrhints_amd import train_fused
    g, src = load_npz("outside_hintgrad_b.npz"), load_npz("outside_b.npz")
    cfg = na.NeuSModelConfig(renderer=na.NeuSRendererConfig(use_outside_nerf=True, shadow_hint_gradient=True, specular_hint_gradient=True))
    assert na.unsupported_reason(cfg) is None
    state = {k: T(np.asarray(v)) for k, v in scene_states["b"].items()}
    state.update({"outside_nerf." + k[5:]: T(v) for k, v in src.items() if k.startswith("nerf.")})
    bg = torch.ones(1, 3).cuda()
    model = na.NeuSHintRenderer(cfg, precision=prec)
    model.load_state_dict(state)
    model = model.cuda().train()
    tb = _bundle(*(src["t." + k] for k in ("o", "d", "pl", "near", "far")))
    gs, gt = int(src["t.global_step"]), cu(src["t.rgb_gt"])
    jit = dict(t_rand_primary=cu(g["t.t_rand_primary"]), t_rand_shadow=cu(g["t.t_rand_shadow"]), t_rand_outside=cu(g["t.t_rand_outside"]))
    assert train_fused.supported(model, tb) is None
    l8 = train_fused.train_step_backward(model, tb, gt, bg, gs, **jit)
    np.testing.assert_allclose(float(l8[0]), float(g["t.loss64"]), rtol=3e-4)
    named = dict(model.named_parameters())
    keys = [k for k in g if k.startswith("t.grad.")]
    assert len(keys) == 14
    for k in keys:
        name = k[len("t.grad."):]
        want64 = g[k.replace("t.grad.", "t.grad64.")]
        bound, scale = grad_bound(g[k], want64, factor=4.0, floor=1e-2 if np.size(want64) == 1 else 5e-3)      # as test_outside_nerf_background
        err = float(np.abs(named[name].grad.detach().cpu().numpy().astype(np.float64) - want64).max())
        assert err <= bound, (name, err, bound, scale)
    # without the hint terms the variance gradient would be off by its own size: make sure the fixture discriminates
    v_with, v_without = float(g["t.grad64.deviation_network.variance"]), float(src["t.grad64.deviation_network.variance"])
    assert abs(float(named["deviation_network.variance"].grad) - v_with) < 0.1 * abs(v_with - v_without)
    with pytest.raises(NotImplementedError):
        model(tb, is_training=True, background_rgb=bg, global_step=gs)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_outside_nerf_ray_gradients_autograd_path(scene_states, prec):
    """Pose / light refinement with the background (forward() + backward(), the path train_fused.supported sends this pair to): the
    gradients of origins, directions and light positions against the reference's recorded float64 step (outside_b.npz: t.grad64.rays.*;
    the far bound of a ray is a function of its origin and direction in the reference's ray generator and positions the 32 samples
    beyond the sphere, :689-693 - here it is a constant of the bundle, as the fixture's rays are leaves)."""
    from nrhints_amd.training import train_loss_dict
    g = load_npz("outside_b.npz")
    cfg = na.NeuSModelConfig(renderer=na.NeuSRendererConfig(use_outside_nerf=True))
    state = {k: T(np.asarray(v)) for k, v in scene_states["b"].items()}
    state.update({"outside_nerf." + k[5:]: T(v) for k, v in g.items() if k.startswith("nerf.")})
    model = na.NeuSHintRenderer(cfg, precision=prec)
    model.load_state_dict(state)
    model = model.cuda().train()
    tb = _bundle(*(g["t." + k] for k in ("o", "d", "pl", "near", "far")))
    for t_ in (tb.origins, tb.directions, tb.pl_positions):
        t_.requires_grad_(True)
    o = model(tb, is_training=True, background_rgb=torch.ones(1, 3).cuda(), global_step=int(g["t.global_step"]),
              _t_rand_primary=cu(g["t.t_rand_primary"]), _t_rand_shadow=cu(g["t.t_rand_shadow"]), _t_rand_outside=cu(g["t.t_rand_outside"]))
    train_loss_dict(o, cu(g["t.rgb_gt"]), 0.1)["loss"].backward()
    for nm in ("origins", "directions", "pl_positions"):
        want64 = g[f"t.grad64.rays.{nm}"]
        bound, scale = grad_bound(g[f"t.grad.rays.{nm}"], want64, factor=4.0, floor=5e-3)
        err = float(np.abs(getattr(tb, nm).grad.detach().cpu().numpy().astype(np.float64) - want64).max())
        assert err <= bound, (nm, err, bound, scale)


def test_fused_register_view_step_with_outside_nerf(scene_states):
    """The step register_view takes (pipelines/base_pipeline.py:80-91: evaluation-mode forward, L1 / (N + 1e-5), renderer frozen, only
    the rays' gradients) on a model with the outside-NeRF background: fused against forward() + backward() on the same rays, the
    far bound's gradient included; no parameter of the renderer or of the background network receives a gradient."""
    from nrhints_amd import train_fused
    g = load_npz("outside_b.npz")
    cfg = na.NeuSModelConfig(renderer=na.NeuSRendererConfig(use_outside_nerf=True))
    state = {k: T(np.asarray(v)) for k, v in scene_states["b"].items()}
    state.update({"outside_nerf." + k[5:]: T(v) for k, v in g.items() if k.startswith("nerf.")})
    model = na.NeuSHintRenderer(cfg)
    model.load_state_dict(state)
    model = model.cuda().eval()
    bg = torch.ones(1, 3).cuda()
    n = 64
    gt = cu(np.random.RandomState(9).rand(n, 3).astype(np.float32))

    def rays():
        rb = _bundle(*(g[k] for k in ("o", "d", "pl", "near", "far")))
        for t_ in (rb.origins, rb.directions, rb.pl_positions, rb.fars):
            t_.requires_grad_(True)
        return rb

    rb = rays()
    out = model(rb, background_rgb=bg, is_training=False)
    loss = torch.nn.functional.l1_loss(out.rgb, gt, reduction="sum") / (n + 1e-5)
    loss.backward()
    want = dict(origins=rb.origins.grad.clone(), directions=rb.directions.grad.clone(), pl_positions=rb.pl_positions.grad.clone(), fars=rb.fars.grad.clone())
    model.zero_grad(set_to_none=True)
    for p in model.parameters():
        p.requires_grad_(False)
    rb2, grads = rays(), {}
    assert train_fused.supported(model, rb2) is None
    loss8 = train_fused.train_step_backward(model, rb2, gt, bg, 0, igr_weight=0.0, is_training=False, ray_grads=grads)
    np.testing.assert_allclose(float(loss8[0]), float(loss.detach()), rtol=1e-5)
    for nm, w in want.items():
        scale = float(w.abs().max()) + 1e-30
        assert float((grads[nm].reshape(w.shape) - w).abs().max()) < 2e-4 * scale + 1e-7, (nm, float((grads[nm].reshape(w.shape) - w).abs().max()), scale)
    assert all(p.grad is None for p in model.parameters())
