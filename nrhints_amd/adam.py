"""``torch.optim.Adam`` whose ``step()`` is ONE HIP launch over all parameter tensors (csrc/nrh_adam.hip, nrh_adam_step).

The reference trains with ``torch.optim.Adam`` over two parameter groups (trainer/trainer.py:99-102, step at :281).  torch's
capturable implementation spends ~100 small kernels per step on the renderer's 46 tensors (0.5 ms of a 7 ms step on MI355X,
profiles/r03/train_graph_step_v1.txt); this subclass keeps the optimiser object - parameter groups, ``state_dict()`` layout
(``step`` float32 scalar on the device, ``exp_avg``, ``exp_avg_sq``), ``load_state_dict``, LR schedulers - and replaces the
arithmetic by one kernel in the operation order of torch's DEFAULT Adam, the one the reference runs (bias corrections as
double scalars; the capturable variant's division by the learning rate turns lr = 0 - step 0 of the warm-up - into NaN for
entries without gradient history).  The per-tensor descriptor table lives on the device and is rebuilt only
when a pointer changes (never in steady state: the fused training step writes gradients into persistent buffers); when the
gradients do move every step (the autograd path allocates them) they are copied into buffers of the optimiser's own first
(one multi-tensor launch).
"""
from __future__ import annotations

import ctypes
from typing import List

import torch

from . import _lib

CHUNK = 2048


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, capturable=True, foreach=False)
        if len(self.param_groups) > 4:
            raise ValueError("HipAdam: at most 4 parameter groups")
        self._key = None
        self._table = self._chunks = None
        self._keep: List[torch.Tensor] = []
        self._counts = (0, 0)
        self._rebuilt_last = False
        self._stage = None           # gradients that move every step (autograd allocates them): staged through fixed buffers

    def load_state_dict(self, state_dict):
        """``torch.optim.Adam.load_state_dict`` + the state layout this class steps on.  A checkpoint written by the reference's
        (non-capturable) Adam stores ``step`` as a CPU tensor and ``capturable: False`` in its groups (trainer/trainer.py:155,222);
        torch leaves both as they are, and the kernel would then read a HOST pointer.  After loading, every ``step`` is a float32
        0-dim tensor on its parameter's device and every group is capturable again; the descriptor table is rebuilt at the next
        step."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            group["capturable"], group["foreach"], group["fused"] = True, False, None
            for p in group["params"]:
                st = self.state.get(p)
                if not st:
                    continue
                step = st.get("step", 0.0)
                step = step.detach().to(device=p.device, dtype=torch.float32).reshape(()) if torch.is_tensor(step) \
                    else torch.tensor(float(step), dtype=torch.float32, device=p.device)
                st["step"] = step.clone()
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st:
                        st[k] = st[k].detach().to(device=p.device, dtype=p.dtype).contiguous()
        self._key = None
        self._rebuilt_last = False

    def _entries(self):
        ents, grads = [], []
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("weight_decay", 0) or group.get("maximize"):
                raise ValueError("HipAdam implements plain Adam (no amsgrad / weight decay / maximize)")
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise ValueError("HipAdam: parameters must be contiguous float32 CUDA tensors")
                if not (g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == p.numel()):
                    raise ValueError("HipAdam: gradients must be contiguous float32 CUDA tensors of the parameter's size")
                st = self.state[p]
                if len(st) == 0:       # torch.optim.Adam._init_group, capturable layout
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if not (torch.is_tensor(st["step"]) and st["step"].is_cuda and st["step"].dtype == torch.float32):
                    raise ValueError("HipAdam: optimizer state 'step' must be a float32 CUDA scalar (load the state through "
                                     "HipAdam.load_state_dict, which converts a default-Adam checkpoint)")
                ents.append((p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(),
                             p.numel(), gi))
                grads.append(g)
        return ents, grads

    def _upload(self, ents, dev) -> None:
        n = len(ents)
        tab = (_lib.NrhAdamTensor * max(n, 1))()
        pairs = []
        for i, (p, g, m, v, s, numel, gi) in enumerate(ents):
            tab[i] = _lib.NrhAdamTensor(p, g, m, v, s, numel, gi, 0)
            pairs += [(i, c) for c in range((numel + CHUNK - 1) // CHUNK)]
        host_t = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).clone().pin_memory()
        host_c = torch.tensor(pairs if pairs else [(0, 0)], dtype=torch.int32).reshape(-1).pin_memory()
        # (pinned staging buffers stay alive with the device copies: a copy captured into a hipGraph re-reads them at every replay)
        self._keep = [host_t, host_c]
        self._table = host_t.to(dev, non_blocking=True)
        self._chunks = host_c.to(dev, non_blocking=True)
        self._counts = (n, len(pairs))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ents, grads = self._entries()
        if not ents:
            return loss
        dev = grads[0].device
        if self._stage is not None and len(self._stage) == len(grads) and all(s.shape == g.shape for s, g in zip(self._stage, grads)):
            torch._foreach_copy_(self._stage, grads)
            ents = [(e[0], s.data_ptr()) + e[2:] for e, s in zip(ents, self._stage)]
        key = tuple(ents)
        if key != self._key:
            if self._rebuilt_last and self._key is not None and self._stage is None and not torch.cuda.is_current_stream_capturing():
                # second consecutive step with new gradient addresses: from now on the gradients are copied (one multi-tensor
                # launch) into buffers of our own, so that the descriptor table is uploaded once
                self._stage = [torch.empty_like(g) for g in grads]
                torch._foreach_copy_(self._stage, grads)
                ents = [(e[0], s.data_ptr()) + e[2:] for e, s in zip(ents, self._stage)]
                key = tuple(ents)
            self._upload(ents, dev)
            self._key, self._rebuilt_last = key, True
        else:
            self._rebuilt_last = False
        ng = len(self.param_groups)
        D, VP = ctypes.c_double * ng, ctypes.c_void_p * ng
        lr_val, lr_ptr = [], []
        for group in self.param_groups:
            lr = group["lr"]
            if torch.is_tensor(lr):
                if not (lr.is_cuda and lr.dtype == torch.float32):
                    raise ValueError("HipAdam: a tensor learning rate must be a float32 CUDA scalar")
                lr_val.append(0.0); lr_ptr.append(lr.data_ptr())
            else:
                lr_val.append(float(lr)); lr_ptr.append(None)
        lib = _lib.load()
        with torch.cuda.device(dev):
            rc = lib.nrh_adam_step(ctypes.c_void_p(self._table.data_ptr()), self._counts[0], ctypes.c_void_p(self._chunks.data_ptr()),
                                   self._counts[1], ng, D(*lr_val), VP(*lr_ptr), D(*[float(g["betas"][0]) for g in self.param_groups]),
                                   D(*[float(g["betas"][1]) for g in self.param_groups]), D(*[float(g["eps"]) for g in self.param_groups]),
                                   _lib.stream_handle())
        _lib.check(rc, "nrh_adam_step")
        # the kernel wrote the parameters behind autograd's back: advance their version counters, as an in-place torch op would
        # (the renderer's pack cache and autograd's saved-tensor checks key on them)
        torch.autograd.graph.increment_version([p for g in self.param_groups for p in g["params"] if p.grad is not None])
        return loss
