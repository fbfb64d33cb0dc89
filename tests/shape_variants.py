"""Networks NARROWER than the compiled shape (fields/sdf_field.py:11-36, fields/reflectance_network.py:9-22): the variants of
tests/golden/make_golden_shapes.py, shared by the CPU and GPU tests.  The fixture holds no weights: the state is rebuilt from the
package's constructor under the reference's seed - checked against the recorded sums of the reference's own initialisation - and
nrhints_amd.synthetic.perturb_state."""
import numpy as np
import torch

import nrhints_amd as na
from nrhints_amd.synthetic import perturb_state

VARIANTS = {
    "n128": (dict(d_hidden=128, multi_res=4, d_out_feat=128), dict(d_hidden=128, multi_res=2), dict()),
    "n192": (dict(d_hidden=192, multi_res=6, d_out_feat=64), dict(), dict()),
    "n160s": (dict(d_hidden=160, multi_res=5), dict(d_hidden=96, multi_res=3), dict(specular_hint=False)),
}


def config(vt):
    s, c, r = VARIANTS[vt]
    return na.NeuSModelConfig(sdf_network=na.SDFNetConfig(**s), reflectance_network=na.ReflectanceNetConfig(**c),
                              renderer=na.NeuSRendererConfig(**r))


def state(vt, g):
    """The variant's scene: constructor under torch.manual_seed(0) (== the reference's init: every tensor's float64 sum is compared
    with the fixture's record of the reference constructor), then perturb_state."""
    torch.manual_seed(0)
    sd = {k: v.detach().numpy().copy() for k, v in na.NeuSHintRenderer(config(vt)).state_dict().items()}
    keys = [k[len(vt) + 10:] for k in g if k.startswith(vt + ".init_sum.")]
    assert sorted(keys) == sorted(sd)
    for k in keys:
        assert float(sd[k].astype(np.float64).sum()) == float(g[f"{vt}.init_sum.{k}"]), (vt, k)
    return perturb_state(sd, pe_cols=6 * VARIANTS[vt][0].get("multi_res", 6))
