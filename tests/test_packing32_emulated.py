"""CPU proof for the wide f16x3 SDF kernel (csrc/nrh_sdf32.hip): the packed streams of nrhints_amd/packing32.py driven through
a numpy emulation of the kernel's lane / register / K-slot arithmetic (tests/mfma32_emulator.py) against the fp64 oracle."""
import numpy as np
import pytest
import torch

from nrhints_amd import packing as pk
from nrhints_amd import packing32 as pk32
from oracle import neus_oracle as orc
from tests import mfma32_emulator as emu


@pytest.fixture(scope="module", params=["a", "b"])
def packed32(request, scene_states):
    st = {k: torch.from_numpy(np.asarray(v)) for k, v in scene_states[request.param].items()}
    d = pk.dense_params(st)
    pk.check_default_shapes(d)
    streams, tables = pk32.pack_sdf32(d)
    return orc.params_from_state(scene_states[request.param], torch.float64), streams.numpy(), tables.numpy()


def _stream(streams, mode):
    o = pk32.stream_offset_bytes(mode) // 2
    return streams[o:o + pk32.stream_bytes(mode) // 2]


def test_stream_sizes():
    assert pk32.stream_bytes(0) == 49152 + 59 * 32768
    assert pk32.stream_bytes(1) == pk32.stream_bytes(0) + 60 * 32 * 1024
    assert pk32.stream_bytes(2) == pk32.stream_bytes(1) + 8 * 32 * 1024


def test_sdf32_chain_emulated(packed32):
    p64, streams, tables = packed32
    assert streams.dtype == np.float16 and streams.size * 2 == sum(pk32.stream_bytes(m) for m in range(3))
    rs = np.random.RandomState(5)
    pts = (rs.rand(32, 3) * 2 - 1) * 0.8
    o_sdf, o_feat, o_grad = orc.sdf_forward_grad_analytic(p64, torch.from_numpy(pts))
    sdf, grad, feat = emu.sdf32_tile(_stream(streams, 2), tables, pts, 2)
    # fp32-class: the weights are fp16 hi/lo pairs (22 bits) of the float32 parameters, the products drop lo*lo
    np.testing.assert_allclose(sdf, o_sdf.numpy()[:, 0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(feat, o_feat.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(grad, o_grad.numpy(), rtol=0, atol=2e-4)    # unorm16 sigma' (7.6e-6 per layer) dominates
    sdf1, grad1, _ = emu.sdf32_tile(_stream(streams, 1), tables, pts, 1)
    np.testing.assert_array_equal(sdf1, sdf)
    np.testing.assert_array_equal(grad1, grad)
    sdf0, _, _ = emu.sdf32_tile(_stream(streams, 0), tables, pts, 0)
    np.testing.assert_array_equal(sdf0, sdf)


def test_color32_chain_emulated(scene_states):
    """The reflectance net's block stream (packing32.pack_color32 + fuse_feature_head) through the numpy emulation of
    csrc/nrh_color32.hip against the fp64 oracle (fields/reflectance_network.py:68-96): column permutation of layer 0, the
    fused feature block, packed fp16 biases, the 3-row output chunk."""
    st = {k: torch.from_numpy(np.asarray(v)) for k, v in scene_states["b"].items()}
    d = pk.dense_params(st)
    stream, tables = pk32.pack_color32(d)
    assert stream.numel() * 2 == pk32.color32_stream_bytes() and tables.shape == (5, 256)
    p64 = orc.params_from_state(scene_states["b"], torch.float64)
    g = torch.Generator().manual_seed(4)
    pts = torch.rand(32, 3, generator=g, dtype=torch.float64) * 2 - 1
    nrm = torch.nn.functional.normalize(torch.randn(32, 3, generator=g, dtype=torch.float64), dim=-1)
    feat = torch.randn(32, 256, generator=g, dtype=torch.float64) * 0.3
    view = torch.nn.functional.normalize(torch.randn(1, 3, generator=g, dtype=torch.float64), dim=-1)
    pl = torch.randn(1, 3, generator=g, dtype=torch.float64) * 3
    vis, cue = torch.rand(1, 1, generator=g, dtype=torch.float64), torch.rand(1, 4, generator=g, dtype=torch.float64) * 2
    rep = lambda x: x.expand(32, x.shape[-1])
    ref = orc.color_forward(p64, pts, nrm, rep(view), feat, rep(pl), rep(vis), rep(cue)).numpy()
    raymisc = torch.cat([orc.nerf_encode(view, 4), orc.nerf_encode(pl, 4), orc.nerf_encode(vis, 4), orc.nerf_encode(cue, 4)], dim=-1)[0]
    assert raymisc.shape[0] == 99
    fd = pk32.fuse_feature_head(d)                      # part = W0feat (Wf h + bf): apply the fused head to the "feature" input
    part = feat @ d["col_w0"].double()[:, 60:316].t()   # (feat here plays Wf h + bf: the fusion is linear in it)
    got = emu.color32_tile(stream.numpy(), tables.numpy(), part.numpy(), pts.numpy(), nrm.numpy(), raymisc.numpy())
    np.testing.assert_allclose(got, ref, rtol=0, atol=3e-6)
    assert fd["feat_w"].shape == (256, 256)
    # the experiment build with the UNSCALED activation residual (gen_mlp32.py NRH32_COL_UNSCALED): same plan, same accuracy class
    got_u = emu.color32_tile(stream.numpy(), tables.numpy(), part.numpy(), pts.numpy(), nrm.numpy(), raymisc.numpy(), scaled=False)
    np.testing.assert_allclose(got_u, ref, rtol=0, atol=3e-6)


def test_sdf32_jvp_mode_emulated(packed32):
    """MODE 3 (value + derivative along a direction in forward mode, 16 points + 16 tangents per tile) on the forward-only
    stream against the fp64 oracle: sdf and <direction, gradient>."""
    p64, streams, tables = packed32
    rs = np.random.RandomState(9)
    pts = (rs.rand(16, 3) * 2 - 1) * 0.8
    dirs = rs.randn(16, 3)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    o_sdf, _, o_grad = orc.sdf_forward_grad_analytic(p64, torch.from_numpy(pts))
    sdf, dd = emu.sdf32_tile_jvp(_stream(streams, 0), tables, pts, dirs)
    np.testing.assert_allclose(sdf, o_sdf.numpy()[:, 0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(dd, (o_grad.numpy() * dirs).sum(-1), rtol=0, atol=2e-5)


@pytest.mark.parametrize("level", ["residuals_subnormal", "all_subnormal"])
def test_sdf32_subnormal_stress_emulated(scene_states, level):
    """The stress states of tests/stress_states.py through the emulation: with fp16 subnormals honoured (numpy float16 does) the
    wide kernel's plan stays at fp32 round-off against the fp64 oracle; with a split that flushes them it does NOT - so the GPU
    test on the same states (tests/test_gpu_wide.py::test_wide_sdf_subnormal_activations) can tell the two apart."""
    from tests.stress_states import subnormal_stress_state
    st_np = subnormal_stress_state(scene_states["b"], level)
    st = {k: torch.from_numpy(np.asarray(v)) for k, v in st_np.items()}
    d = pk.dense_params(st)
    streams, tables = pk32.pack_sdf32(d)
    assert bool(pk32.tables_in_f16_range(d)) and bool(torch.isfinite(streams).all())      # the gain stays inside the fp16 split
    p64 = orc.params_from_state(st_np, torch.float64)
    rs = np.random.RandomState(5)
    pts = (rs.rand(32, 3) * 2 - 1) * 0.8
    o_sdf, o_feat, o_grad = orc.sdf_forward_grad_analytic(p64, torch.from_numpy(pts))
    trace = {}
    sdf, grad, feat = emu.sdf32_tile(_stream(streams.numpy(), 2), tables.numpy(), pts, 2, trace=trace)
    u5 = np.abs(np.stack(trace["u"][5]))
    assert u5.max() < (2.0 ** -3 if level == "residuals_subnormal" else 2.0 ** -12)        # the regime the test is named after
    np.testing.assert_allclose(sdf, o_sdf.numpy()[:, 0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(feat, o_feat.numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(grad, o_grad.numpy(), rtol=0, atol=5e-4)
    # the same plan with fp16 subnormals flushed to zero: visibly wrong
    honest = emu.split16

    def flushing(x):
        hi, lo = honest(x)
        return np.where(np.abs(hi) < 2.0 ** -14, 0.0, hi), np.where(np.abs(lo) < 2.0 ** -14, 0.0, lo)

    emu.split16 = flushing
    try:
        sdf_f, _, feat_f = emu.sdf32_tile(_stream(streams.numpy(), 2), tables.numpy(), pts, 2)
    finally:
        emu.split16 = honest
    assert np.abs(sdf_f - o_sdf.numpy()[:, 0]).max() > 2e-5 and np.abs(feat_f - o_feat.numpy()).max() > 1e-4
