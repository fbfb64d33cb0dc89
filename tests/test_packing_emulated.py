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


def _to_dlayout(rows):
    """[16 points, F] row-major -> D-layout registers [F/4 regs, 64 lanes] (reg b*4+r of lane q*16+j = rows[j, 16b+4q+r])."""
    F = rows.shape[1]
    out = np.zeros((F // 4, 64))
    for b in range(F // 16):
        for r in range(4):
            out[b * 4 + r] = rows[emu.J, 16 * b + 4 * emu.Q + r]
    return out


def _from_dlayout(regs):
    nb = regs.shape[0] // 4
    rows = np.zeros((16, nb * 16))
    for b in range(nb):
        for r in range(4):
            rows[emu.J, 16 * b + 4 * emu.Q + r] = regs[b * 4 + r]
    return rows


@pytest.mark.parametrize("precision", [0, 1])
def test_training_stage_packings_emulated(scene_states, precision):
    """The packings only the training sweeps use - the transposed feature head (first stage of the SDF value sweep),
    the transposed reflectance stages (colour adjoint sweep) and the batched packer - executed by the MFMA emulation
    with the kernels' index arithmetic: every stage equals the dense transposed matrix product, and the whole colour
    adjoint sweep equals the chain rule through the 5 linears (csrc/nrh_sdf_train.hip, csrc/nrh_color.hip)."""
    st = {k: torch.from_numpy(np.asarray(v)).double() for k, v in scene_states["b"].items()}
    d = pk.dense_params(st)
    tol = 1e-12 if precision == 0 else 2e-6
    rs = np.random.RandomState(5)
    # batched packer == per-stage packer
    ws = [d["sdf_w1"], d["sdf_w3"], d["col_w2"].t()]
    ps = pk.pack_stage if precision == 0 else pk.pack_stage_h3
    assert torch.equal(pk.pack_stages(ws, 256, 256, precision), torch.cat([ps(w, 256, 256) for w in ws]))
    # transposed feature head: hbar = Wf^T fbar
    fbar = rs.randn(16, 256)
    out = emu.run_stage(pk.pack_feat_transposed(d, precision).numpy(), 16, 8, _to_dlayout(fbar))
    np.testing.assert_allclose(_from_dlayout(out), fbar @ d["feat_w"].numpy(), rtol=0, atol=tol * 30)
    # colour adjoint sweep
    for hints in (True, False):
        dd = dict(d)
        if not hints:
            dd["col_w0"] = d["col_w0"][:, :316]
        wt = pk.pack_color_transposed(dd, precision, hints).numpy()
        mkb = 8 if hints else 4
        per = 1 if precision == 0 else 2          # elements per packed float slot
        c4 = 8 * 2 * 2 * 256 * per
        reg = pk.SDF_REG_FLOATS * per
        assert wt.size == c4 + 4 * reg + (mkb // 2) * 2 * 16 * 256 * per
        hs = [np.maximum(rs.randn(16, 256), 0.0) for _ in range(4)]
        zbar4 = rs.randn(16, 3)
        z4 = np.zeros((16, 32)); z4[:, :3] = zbar4
        z = _from_dlayout(emu.run_stage(wt[:c4], 2, 8, _to_dlayout(z4))) * (hs[3] > 0)
        z_ref = (zbar4 @ dd["col_w4"].numpy()) * (hs[3] > 0)
        np.testing.assert_allclose(z, z_ref, rtol=0, atol=tol * 30)
        for i, l in enumerate((3, 2, 1)):
            z = _from_dlayout(emu.run_stage(wt[c4 + i * reg: c4 + (i + 1) * reg], 16, 8, _to_dlayout(z))) * (hs[l - 1] > 0)
            z_ref = (z_ref @ dd[f"col_w{l}"].numpy()) * (hs[l - 1] > 0)
            np.testing.assert_allclose(z, z_ref, rtol=0, atol=tol * 100)
        fi, mi = pk.color_input_permutation(hints)
        fb = _from_dlayout(emu.run_stage(wt[c4 + 3 * reg: c4 + 4 * reg], 16, 8, _to_dlayout(z)))
        mb = _from_dlayout(emu.run_stage(wt[c4 + 4 * reg:], 16, mkb // 2, _to_dlayout(z)))
        w0 = dd["col_w0"].numpy()
        np.testing.assert_allclose(fb, z_ref @ w0[:, fi.numpy()], rtol=0, atol=tol * 100)
        nm = 105 if hints else 60
        np.testing.assert_allclose(mb[:, :nm], z_ref @ w0[:, mi.numpy()], rtol=0, atol=tol * 100)
        assert np.abs(mb[:, nm:]).max() == 0.0
