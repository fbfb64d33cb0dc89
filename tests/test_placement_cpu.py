"""CPU checks of tests/placement.py (the same-placement gradient oracle of tests/test_gpu_train1024.py): accumulating the step over
chunks of rays gives the unchunked step, and the overrides are inert when fed the oracle's own placement."""
import numpy as np
import torch

from nrhints_amd.synthetic import make_rays
from oracle import neus_oracle as orc
from tests.placement import oracle_step_at_placement

T = torch.from_numpy


def test_chunked_step_equals_unchunked_and_overrides_are_inert(scene_states):
    st = scene_states["b"]
    n, gs = 12, 25000
    o, d, pl, near, far = make_rays(n, seed=3, spread=0.1)
    rays = dict(o=o, d=d, pl=pl, near=near, far=far)
    gen = torch.Generator().manual_seed(7)
    tp, ts = torch.rand(n, 1, generator=gen), torch.rand(n, 64, generator=gen)
    gt = np.random.RandomState(0).rand(n, 3).astype(np.float32)
    # the oracle's own float64 placement
    leaves = {k: T(np.asarray(v)).double().clone().requires_grad_(True) for k, v in st.items()}
    r64 = [T(a).double() for a in (o, d, pl, near, far)]
    for t in r64[:3]:
        t.requires_grad_(True)
    full = orc.render_forward(orc.params_from_state(leaves, torch.float64), *r64, background_rgb=torch.ones(1, 3, dtype=torch.float64),
                              is_training=True, global_step=gs, t_rand_primary=tp.double(), t_rand_shadow=ts.double(), mode="as_written",
                              differentiable=True, keep_intermediates=True)
    loss, rgb_l, eik = orc.train_loss(full, T(gt).double())
    loss.backward()
    z, vis, cue = full["z_vals"].detach(), full["visibilities"].detach(), full["specular_cue"].detach()[:, 0, :]
    for chunk in (5, n):
        l, pg, rg, rgb = oracle_step_at_placement(st, rays, gt, gs, z, vis, cue, tp.numpy(), ts.numpy(), chunk=chunk)
        assert abs(l["loss"] - float(loss)) < 1e-12 and abs(l["eikonal_loss"] - float(eik)) < 1e-12
        np.testing.assert_allclose(rgb, full["rgb"].detach().numpy(), rtol=0, atol=1e-13)
        assert len(pg) == 46
        for k, g in pg.items():
            want = leaves[k].grad.numpy()
            # (deviation_network.variance passes through the reference's float32 inv_s, models/neus_hint_model.py:104-110: 1e-7)
            assert np.abs(g - want).max() <= 2e-7 * np.abs(want).max() + 1e-300, (chunk, k)
        for k, t in zip(("origins", "directions", "pl_positions"), r64[:3]):
            assert np.abs(rg[k] - t.grad.numpy()).max() <= 1e-10 * np.abs(t.grad.numpy()).max() + 1e-300, (chunk, k)
