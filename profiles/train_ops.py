#!/usr/bin/env python3
"""Which torch ops make up a training step (count and device time), via torch.profiler.  MI355X only."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na
from nrhints_amd.synthetic import make_rays
from nrhints_amd.training import make_optimizer, train_step
from torch.profiler import profile, ProfilerActivity, record_function

torch.manual_seed(0)
m = na.NeuSHintRenderer().cuda()
opt, sched = make_optimizer(m, warm_up_end=10)
bg = torch.ones(1, 3, device="cuda")
o, d, pl, near, far = (torch.from_numpy(a).cuda() for a in make_rays(1024, seed=1, spread=0.08))
rb = na.RayBundle(origins=o, directions=d, pl_positions=pl, nears=near, fars=far)
gt = torch.rand(1024, 3, device="cuda")
for i in range(3):
    train_step(m, rb, gt, bg, global_step=20000 + i, optimizer=opt, scheduler=sched)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=False) as prof:
    train_step(m, rb, gt, bg, global_step=20003, optimizer=opt, scheduler=sched)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
print(prof.key_averages().table(sort_by="count", row_limit=30, max_name_column_width=60))
