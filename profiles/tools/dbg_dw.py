import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from nrhints_amd import dw
P=int(sys.argv[1])
g=lambda *s: torch.randn(*s, device='cuda')
A1,A2,B1,B2=g(P,256),g(P,256),g(P,256),g(P,256)
E,GE,misc=g(P,64),g(P,64),g(P,128)
sb=g(P); M3=g(P,3)
new=lambda *s: torch.full(s, float('nan'), device='cuda')
out=dict(full=new(256,256), bfull=new(256), l0=new(256,39), bl0=new(256), rows=new(217,256), brows=new(217), ws=new(1,256), bs=new(1), w0=new(256,361), b0=new(256), w4=new(3,256), b4=new(3))
fi, mi = dw.color_col_maps(torch.device("cuda"), True)
jobs = [dw.Job([A1, A2], [B1, B2], 256, 256, out["full"], colsum_a=out["bfull"]),
        dw.Job([A1, A2], [E, GE], 256, 39, out["l0"], colsum_a=out["bl0"]),
        dw.Job([A2], [B1], 256, 256, out["rows"], rows=217, scale=2.0 ** -0.5, colsum_a=out["brows"]),
        dw.Job([B1, A2], [sb.reshape(P, 1), dw.ones(P, "cuda").reshape(P, 1)], 256, 1, out["ws"], transpose=True, scale=1.0 / 3.0,
               colsum_b=out["bs"], scale_b=1.0 / 3.0),
        dw.Job([A1], [B1], 256, 256, out["w0"], col_map=fi, colsum_a=out["b0"]),
        dw.Job([A1], [misc], 256, 105, out["w0"], col_map=mi),
        dw.Job([B1], [M3], 256, 3, out["w4"], transpose=True, colsum_b=out["b4"])]
dw.run(jobs,P); torch.cuda.synchronize()
for k,v in out.items(): print(k, 'nan count', int(torch.isnan(v).sum()), 'of', v.numel())
print('bl0 err', float((out['bl0']-A1.sum(0)).abs().max()), 'brows err', float((out['brows']-A2.sum(0)[:217]).abs().max()))
