import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import nrhints_amd as na
from nrhints_amd.synthetic import make_rays, perturb_state
st = dict(np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/scene_a_state.npz")))
if len(sys.argv) > 1 and sys.argv[1] == "b":
    st = perturb_state(st)
m = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in st.items()}); m = m.cuda().eval()
o, d, pl, near, far = (torch.from_numpy(a).cuda() for a in make_rays(64, seed=21, spread=0.1))
rb = na.RayBundle(origins=o, directions=d, pl_positions=pl, nears=near, fars=far)
bg = torch.ones(1, 3).cuda()
res = {}
with torch.no_grad():
    for name, wc, fu in (("wide", True, True), ("fused16", False, True), ("plain", False, False)):
        m.wide_color, m.fuse_feature_head = wc, fu
        out = m(rb, background_rgb=bg)
        res[name] = out.rgb.clone()
        cue, vis, w = out.specular_cue[:, 0, :].clone(), out.visibilities.clone(), out.weights.sum(-1)
for x, y in (("wide", "fused16"), ("fused16", "plain"), ("wide", "plain")):
    dlt = (res[x] - res[y]).abs()
    print(x, y, "max", float(dlt.max()), "mean", float(dlt.mean()))
dlt = (res["wide"] - res["fused16"])
print("signed mean per channel", dlt.mean(0).tolist(), "first rows", dlt[:4].tolist())

worst = (res["wide"] - res["fused16"]).abs().max(-1).values
idx = torch.argsort(worst, descending=True)[:6]
for i in idx.tolist():
    print(i, "diff", float(worst[i]), "cue", [round(v, 3) for v in cue[i].tolist()], "vis", round(float(vis[i]), 4), "wsum", round(float(w[i]), 4))
print("rays with diff > 1e-5:", int((worst > 1e-5).sum()), "of", worst.numel(), " median diff", float(worst.median()))
