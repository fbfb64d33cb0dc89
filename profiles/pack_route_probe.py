#!/usr/bin/env python3
"""Which buffer differs between the evaluation pack (packed_params(dense=None)) and the training pack (packed_params(dense=...))
of the SAME parameters?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd import packing
from nrhints_amd.synthetic import perturb_state
T = torch.from_numpy
st = perturb_state(dict(np.load(os.path.join(ROOT, "tests", "golden", "scene_a_state.npz"))))
for prec in ("f16x3", "f32"):
    def model():
        m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision=prec)
        m.load_state_dict({k: T(np.asarray(v)) for k, v in st.items()})
        return m.cuda()
    dev = torch.device("cuda", 0)
    a, b = model(), model()
    pa = a.packed_params(dev)
    named = dict(b.named_parameters())
    gs = [named[k + ".weight_g"].detach() for k in packing._FOLD_LAYERS]
    vs = [named[k + ".weight_v"].detach() for k in packing._FOLD_LAYERS]
    ws = [torch.empty_like(v) for v in vs]
    packing.WeightNormFoldHip._call("nrh_weight_norm_fold", vs, gs, ws)
    dense = {}
    for (wk, bk), k, w in zip(packing._FOLD_KEYS, packing._FOLD_LAYERS, ws):
        dense[wk], dense[bk] = w, named[k + ".bias"].detach()
    pb = b.packed_params(dev, dense=dense)
    for k in sorted(set(pa) | set(pb)):
        x, y = pa.get(k), pb.get(k)
        if torch.is_tensor(x) and torch.is_tensor(y):
            same = x.shape == y.shape and x.dtype == y.dtype and torch.equal(x.view(torch.uint8) if x.dtype != torch.float32 else x, y.view(torch.uint8) if y.dtype != torch.float32 else y)
            nd = int((x.view(torch.int32) != y.view(torch.int32)).sum()) if x.shape == y.shape and x.dtype == y.dtype and x.element_size() % 4 == 0 else -1
            print(prec, k, "same" if same else f"DIFFERENT ({nd} words)", tuple(x.shape), x.dtype)
        else:
            print(prec, k, "a:", type(x).__name__ if not isinstance(x, (int, float, bool)) else x, "b:", type(y).__name__ if not isinstance(y, (int, float, bool)) else y)
    # the folds themselves
    state = {k: v.detach().to(dev) for k, v in a.state_dict().items()}
    d1 = packing.dense_params_hip(state)
    for k in dense:
        if not torch.equal(d1[k], dense[k]):
            print(prec, "fold differs:", k, float((d1[k] - dense[k]).abs().max()))
