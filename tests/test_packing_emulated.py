"""CPU proof that the packed buffers + the kernels' index arithmetic compute the reference networks:
a numpy emulation of the MFMA register chain (tests/mfma_emulator.py) against the oracle in float64."""
import numpy as np
import pytest
import torch

from nrhints_amd import packing as pk
from oracle import neus_oracle as orc
from tests import mfma_emulator as emu


@pytest.fixture(scope="module", params=["a", "b"])
def packed(request, scene_states):
    st = {k: torch.from_numpy(np.asarray(v)).double() for k, v in scene_states[request.param].items()}
    d = pk.dense_params(st)
    pk.check_default_shapes(d)
    return (orc.params_from_state(scene_states[request.param], torch.float64), pk.pack_sdf(d), pk.pack_color(d))


def test_sdf_chain_emulated(packed):
    p64, (w, b, head), _ = packed
    rs = np.random.RandomState(0)
    pts = (rs.rand(16, 3) * 2 - 1) * 0.8
    sdf, grad, feat = emu.sdf_tile(w.numpy(), b.numpy(), head.numpy(), pts, mode=2)
    o_sdf, o_feat, o_grad = orc.sdf_forward_grad_analytic(p64, torch.from_numpy(pts))
    np.testing.assert_allclose(sdf, o_sdf.numpy()[:, 0], rtol=0, atol=1e-7)   # fp32 pi/2 phase only
    np.testing.assert_allclose(feat, o_feat.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(grad, o_grad.numpy(), rtol=0, atol=1e-5)
    sdf0, _, _ = emu.sdf_tile(w.numpy(), b.numpy(), head.numpy(), pts, mode=0)
    np.testing.assert_array_equal(sdf0, sdf)


def test_color_chain_emulated(packed):
    p64, _, (cw, cb) = packed
    rs = np.random.RandomState(1)
    P = 16
    T = torch.from_numpy
    pts, nrm, view = rs.randn(P, 3), rs.randn(P, 3), rs.randn(P, 3)
    feat, pl, vis, cue = rs.randn(P, 256) * 0.3, rs.randn(P, 3) * 3, rs.rand(P, 1), rs.rand(P, 4)
    ref = orc.color_forward(p64, T(pts), T(nrm), T(view), T(feat), T(pl), T(vis), T(cue)).numpy()
    misc = np.concatenate([pts, nrm, orc.nerf_encode(T(view), 4).numpy(), orc.nerf_encode(T(pl), 4).numpy(),
                           orc.nerf_encode(T(vis), 4).numpy(), orc.nerf_encode(T(cue), 4).numpy()], axis=1)
    assert misc.shape[1] == 105
    rgb = emu.color_tile(cw.numpy(), cb.numpy(), feat, misc)
    np.testing.assert_allclose(rgb, ref, rtol=0, atol=1e-10)


def test_f16x3_chain_emulated(scene_states):
    """The f16x3 packing + split arithmetic (hi, lo*2^11; three products, two accumulators) through the whole SDF
    chain and the colour net: fp32-class accuracy against the fp64 oracle."""
    st = {k: torch.from_numpy(np.asarray(v)) for k, v in scene_states["b"].items()}
    d = pk.dense_params(st)
    p64 = orc.params_from_state(scene_states["b"], torch.float64)
    w, b, head = pk.pack_sdf(d, precision=1)
    assert w.dtype == torch.float16 and w.numel() == 2 * pk.SDF_PACKED_FLOATS
    rs = np.random.RandomState(3)
    pts = (rs.rand(16, 3) * 2 - 1) * 0.8
    sdf, grad, feat = emu.sdf_tile(w.numpy(), b.numpy(), head.numpy(), pts, mode=2)
    o_sdf, o_feat, o_grad = orc.sdf_forward_grad_analytic(p64, torch.from_numpy(pts))
    np.testing.assert_allclose(sdf, o_sdf.numpy()[:, 0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(feat, o_feat.numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(grad, o_grad.numpy(), rtol=0, atol=1e-4)
    cw, cb = pk.pack_color(d, precision=1)
    P = 16
    T = torch.from_numpy
    ptsc, nrm, view = rs.randn(P, 3), rs.randn(P, 3), rs.randn(P, 3)
    featc, pl, vis, cue = rs.randn(P, 256) * 0.3, rs.randn(P, 3) * 3, rs.rand(P, 1), rs.rand(P, 4)
    ref = orc.color_forward(p64, T(ptsc), T(nrm), T(view), T(featc), T(pl), T(vis), T(cue)).numpy()
    misc = np.concatenate([ptsc, nrm, orc.nerf_encode(T(view), 4).numpy(), orc.nerf_encode(T(pl), 4).numpy(),
                           orc.nerf_encode(T(vis), 4).numpy(), orc.nerf_encode(T(cue), 4).numpy()], axis=1)
    rgb = emu.color_tile(cw.numpy(), cb.numpy(), featc, misc)
    np.testing.assert_allclose(rgb, ref, rtol=0, atol=2e-6)


def test_feat_tile_roundtrip():
    rows = torch.randn(37, 256)
    tiles = pk.rows_to_feat_tiles(rows)
    assert tiles.numel() == 3 * 4096
    assert torch.equal(pk.feat_tiles_to_rows(tiles, 37), rows)
    # explicit element check of the D-layout: tile t, block b, lane (q*16+j), r <-> point 16t+j, feature 16b+4q+r
    t4 = tiles.reshape(3, 16, 64, 4)
    assert t4[1, 5, 2 * 16 + 7, 3] == rows[16 + 7, 16 * 5 + 4 * 2 + 3]


def test_pack_stage_layout():
    w = torch.arange(32 * 48, dtype=torch.float32).reshape(32, 48)
    p = pk.pack_stage(w, 32, 48).reshape(1, 2, 3, 64, 4)
    lane = 3 * 16 + 5  # q = 3, i = 5
    assert p[0, 1, 2, lane, 1] == w[16 + 5, 2 * 16 + 4 * 3 + 1]


def test_shape_guard(scene_states):
    st = {k: torch.from_numpy(np.asarray(v)) for k, v in scene_states["a"].items()}
    d = pk.dense_params(st)
    d["sdf_w1"] = torch.zeros(64, 256)
    with pytest.raises(ValueError):
        pk.check_default_shapes(d)
