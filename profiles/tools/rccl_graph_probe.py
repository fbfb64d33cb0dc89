#!/usr/bin/env python3
"""Probe: which combination of (RCCL process group alive, collective inside / outside the capture, capture_error_mode)
survives hipGraph capture of the training step on this ROCm/RCCL/PyTorch stack.  Each case runs in its own process.
    python profiles/tools/rccl_graph_probe.py            # runs all cases, prints one line each
"""
import os
import subprocess
import sys

CASES = ["pg_plain", "pg_split", "pg_inside_tl", "nopg_plain", "pg_seq_c", "nopg_seq_c"]


def run_case(case):
    import socket
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import nrhints_amd as na
    from nrhints_amd import training
    from nrhints_amd.synthetic import make_rays, perturb_state
    from nrhints_amd.training import FlatGradAllReduce, GraphedTrainStep
    if case.startswith("pg"):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if case.endswith("_tl"):
        real = torch.cuda.graph
        class tl(real):
            def __init__(self, *a, **k):
                k.setdefault("capture_error_mode", "thread_local")
                super().__init__(*a, **k)
        torch.cuda.graph = tl
    if case == "pg_inside_tl":      # collective inside one graph
        GraphedTrainStep._sync_active = lambda self: False
        body = GraphedTrainStep._body
        def inside(self, upto="all"):
            jit = {}
            out = self.renderer(self.rays, is_training=True, background_rgb=self.bg, global_step=self._capture_step)
            losses = training.train_loss_dict(out, self.gt, self.renderer.config.igr_weight)
            self.optimizer.zero_grad(set_to_none=True)
            params = list(self.renderer.parameters())
            for p, g in zip(params, torch.autograd.grad(losses["loss"], params, allow_unused=True)):
                p.grad = g
            self.grad_sync()
            self.optimizer.step()
            self._keys = list(losses)
            return torch.stack([losses[k].detach().float().reshape(()) for k in self._keys])
        GraphedTrainStep._body = inside
    torch.manual_seed(0)
    m = na.NeuSHintRenderer(na.NeuSModelConfig()).cuda()
    bg = torch.ones(1, 3).cuda()
    n = 128
    o, d, pl, near, far = (torch.from_numpy(a).cuda() for a in make_rays(n, seed=9, spread=0.1))
    rb = na.RayBundle(origins=o, directions=d, pl_positions=pl, nears=near, fars=far)
    gt = torch.rand(n, 3).cuda()
    sync = FlatGradAllReduce(m.parameters(), always=True) if case in ("pg_split", "pg_split_tl", "pg_inside_tl") or "pg_seq" in case else None
    jitter = None
    if "seq" in case:      # the order of tests/test_gpu_parity2.py::test_rccl_world1...
        from nrhints_amd.parallel import render_sharded
        from nrhints_amd.training import train_loss_dict
        if case >= "pg_seq_b":
            o2, d2, pl2, near2, far2 = (torch.from_numpy(a).cuda() for a in make_rays(1001, seed=8, spread=0.12))
            rb2 = na.RayBundle(origins=o2, directions=d2, pl_positions=pl2, nears=near2, fars=far2)
            with torch.no_grad():
                render_sharded(lambda r: m(r, background_rgb=bg), rb2, fields=("rgb", "depth", "visibilities"))
        if case >= "pg_seq_c":
            jitter = (torch.rand(n, 1).cuda(), torch.rand(n, 64).cuda())
            out = m(rb, is_training=True, background_rgb=bg, global_step=30000, _t_rand_primary=jitter[0], _t_rand_shadow=jitter[1])
            train_loss_dict(out, gt)["loss"].backward()
        if sync is not None:
            sync.broadcast_parameters()
            sync()
        m.zero_grad(set_to_none=True)
    step = GraphedTrainStep(m, n, bg, grad_sync=sync, global_step=30000, jitter=jitter)
    ls = [step(rb, gt, global_step=30000 + i)["loss"] for i in range(5)]
    print("CASE", case, "ok", ls[0], ls[-1], flush=True)
    if case.startswith("pg"):
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=300)
            tail = [l for l in r.stdout.splitlines() if l.startswith("CASE")]
            print(c, "rc", r.returncode, tail[-1] if tail else "\n".join(l[:300] for l in r.stderr.strip().splitlines() if "File" not in l and "pluggy" not in l)[-3000:], flush=True)
