"""Worker of tests/test_gpu_fullsize.py::test_two_rank_training_step: run under torch.distributed.run with N ranks (2 or 8), one GPU
each (backend nccl = RCCL) - or, with NRH_WORKER_SHARE_GPU=1, all ranks on cuda:0 with gloo collectives (RCCL refuses two ranks on
one device): the rehearsal of the same code on a one-GPU box.  Not collected by pytest (no test_ prefix)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nrhints_amd as na  # noqa: E402
from nrhints_amd.synthetic import make_rays, perturb_state  # noqa: E402
from nrhints_amd.training import FlatGradAllReduce, GraphedTrainStep, train_loss_dict  # noqa: E402


def main():
    rank, local, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    assert world in (2, 8)
    share = os.environ.get("NRH_WORKER_SHARE_GPU") == "1"
    local = 0 if share else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend="gloo" if share else "nccl")
    a = dict(np.load(os.path.join(ROOT, "tests", "golden", "scene_a_state.npz")))
    model = na.NeuSHintRenderer(na.NeuSModelConfig(), precision="f16x3")
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in perturb_state(a).items()})
    model = model.to(dev)
    n = 128                # configs[2]'s 1 024-ray global batch over 8 ranks (trainer/trainer.py:116-123)
    cu = lambda x: torch.from_numpy(x).float().contiguous().to(dev)
    rb = na.RayBundle(**{k: cu(v) for k, v in zip(("origins", "directions", "pl_positions", "nears", "fars"), make_rays(n, seed=100 + rank, spread=0.1))})
    rs = np.random.RandomState(7 + rank)
    gt, tp, ts = (cu(rs.rand(n, k).astype(np.float32)) for k in (3, 1, 64))
    bg = torch.ones(1, 3, device=dev)
    sync = FlatGradAllReduce(model.parameters())
    sync.broadcast_parameters()
    # 1. eager: the flat all-reduce gives the mean of the two ranks' local gradients
    out = model(rb, is_training=True, background_rgb=bg, global_step=30000, _t_rand_primary=tp, _t_rand_shadow=ts)
    train_loss_dict(out, gt)["loss"].backward()
    local_flat = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    gathered = [torch.empty_like(local_flat) for _ in range(world)]
    dist.all_gather(gathered, local_flat)
    want = torch.stack(gathered).sum(0) / world
    sync()
    got = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    assert float((got - want).abs().max()) <= 1e-7 * max(1.0, float(want.abs().max())), float((got - want).abs().max())
    model.zero_grad(set_to_none=True)
    # 2. graphed: two graphs around one eager all-reduce; parameters stay identical across ranks
    step = GraphedTrainStep(model, n, bg, lr=5e-4, warm_up_end=20, global_step=30000, grad_sync=sync, jitter=(tp, ts))
    assert step.graph_tail is not None
    losses = [step(rb, gt, global_step=30000 + i)["loss"] for i in range(3)]
    assert all(np.isfinite(losses))
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    assert all(torch.equal(both[0], b) for b in both[1:])
    step.release()
    dist.barrier()
    print(f"MULTI_GPU_WORKER_OK rank {rank} losses {losses}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
