#!/usr/bin/env python3
"""Golden fixture for constructor variants that only change the INITIALISATION (no kernel shape): recorded by IMPORTING the reference.
Build container only; writes tests/golden/init_variants.npz (plain data).

    python tests/golden/make_golden_init_variants.py

``SDFNetConfig.inside_outside`` (fields/sdf_field.py:35, used at :95-100) flips the sign of the output layers' geometric initialisation
(mean of ``out_sdf`` / ``out_feat`` weights, sign of their bias); ``init_bias`` (:29, :60) sets the radius of the initial sphere.  Variant
"io": inside_outside=True, init_bias=0.05 (the value scripts/train_synthetic.sh passes for Complex_Ball) under torch.manual_seed(0):
the four output-layer tensors, plus sums of every other tensor (the rest of the state is the default constructor's, checked by
scene_a_state.npz already).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
    class _Sub:
        def __getitem__(self, item):
            return object

    jt = types.ModuleType("jaxtyping")
    for name in ("Float", "Int", "Shaped", "Bool"):
        setattr(jt, name, _Sub())
    sys.modules["jaxtyping"] = jt
    sys.modules["mcubes"] = types.ModuleType("mcubes")


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    import torch
    from fields.sdf_field import SDFNetConfig
    from models.neus_hint_model import NeuSHintRenderer, NeuSModelConfig

    torch.manual_seed(0)
    m = NeuSHintRenderer(NeuSModelConfig(sdf_network=SDFNetConfig(inside_outside=True, init_bias=0.05)))
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    out = {}
    for k, v in sd.items():
        if k.startswith("sdf_network.out_"):
            out["io." + k] = v
        out["io.sum." + k] = np.float64(v.astype(np.float64).sum())
    np.savez_compressed(os.path.join(HERE, "init_variants.npz"), **out)
    print("wrote init_variants.npz:", len(out), "entries;", {k: v.shape for k, v in out.items() if not k.startswith("io.sum.")})


if __name__ == "__main__":
    main()
