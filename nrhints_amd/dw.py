"""Weight gradients of the training step through ``nrh_dw_gemm`` (csrc/nrh_dw.hip): every ``dW = X^T Y`` that
``loss.backward()`` computes for the nn.Linear layers of the two networks (fields/sdf_field.py:81-101,
fields/reflectance_network.py:52-66) as split-K bf16x3 MFMA GEMMs in one launch, bias gradients as by-products.

This module only builds the job table (which saved array meets which, where the product goes) and sizes the K split; there is
no other implementation behind it - without the HIP library it raises like everything else in the package.
"""
from __future__ import annotations

import ctypes
import os
import math
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib


class NrhDwJob(ctypes.Structure):
    """include/nrhints_hip.h: NrhDwJob"""
    _fields_ = [("a", ctypes.c_void_p * 2), ("b", ctypes.c_void_p * 2), ("lda", ctypes.c_int * 2), ("ldb", ctypes.c_int * 2),
                ("npairs", ctypes.c_int), ("m", ctypes.c_int), ("n", ctypes.c_int), ("slabs", ctypes.c_int),
                ("out", ctypes.c_void_p), ("col_map", ctypes.c_void_p), ("ldo", ctypes.c_int), ("transpose", ctypes.c_int),
                ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("scale", ctypes.c_float),
                ("colsum_a", ctypes.c_void_p), ("scale_a", ctypes.c_float), ("colsum_b", ctypes.c_void_p), ("scale_b", ctypes.c_float),
                ("tiled_a", ctypes.c_int * 2), ("tiled_b", ctypes.c_int * 2), ("half_ops", ctypes.c_int), ("dyn_scale", ctypes.c_void_p)]


class Job:
    """out[i, j] = scale * sum_k sum_p A_k[p, i] B_k[p, j]  (A_k [P, lda], B_k [P, ldb] float32 row-major views)."""

    def __init__(self, a: Sequence[torch.Tensor], b: Sequence[torch.Tensor], m: int, n: int, out: Optional[torch.Tensor] = None,
                 rows: Optional[int] = None, cols: Optional[int] = None, transpose: bool = False, scale: float = 1.0,
                 col_map: Optional[torch.Tensor] = None, colsum_a: Optional[torch.Tensor] = None, scale_a: float = 1.0,
                 colsum_b: Optional[torch.Tensor] = None, scale_b: float = 1.0, tiled_a: Sequence[bool] = (), tiled_b: Sequence[bool] = (),
                 half: bool = False, dyn_scale: Optional[torch.Tensor] = None):
        """``tiled_a`` / ``tiled_b``: per pair, the operand is in the tiled layout of the training arrays (csrc/nrh_mlp.h: what the
        sweep kernels write h, t, abar, zbar in when ``arrays_tiled()``) instead of row-major; 256 channels only.
        ``half``: every operand is a float16 [P, 256] array in the HALF-TILED layout (``to_half_tiled``; what the f16x3 sweeps write
        when asked for 16-bit hand-offs) - full 256 x 256 products only, one fp16 MFMA pass.  ``dyn_scale``: device float32 [2]
        {S, 1 / S} (nrh_adjoint_range): the product and colsum_a are multiplied by its second entry."""
        assert len(a) == len(b) and 1 <= len(a) <= 2
        self.a, self.b, self.m, self.n, self.out = list(a), list(b), m, n, out
        self.rows, self.cols = (m if rows is None else rows), (n if cols is None else cols)
        self.transpose, self.scale, self.col_map = transpose, scale, col_map
        self.colsum_a, self.scale_a, self.colsum_b, self.scale_b = colsum_a, scale_a, colsum_b, scale_b
        self.tiled_a = [bool(t) for t in tiled_a] + [False] * (len(self.a) - len(tiled_a))
        self.tiled_b = [bool(t) for t in tiled_b] + [False] * (len(self.b) - len(tiled_b))
        self.half, self.dyn_scale = bool(half), dyn_scale
        if self.half:
            assert m == 256 and n == 256 and colsum_b is None and all(x.dtype == torch.float16 for x in self.a + self.b)

    def cost(self) -> float:
        """relative time of one K step: 6 MFMAs per 32 output columns per wave + the load / split / LDS overhead of both operands"""
        nf = 1 if self.n <= 32 else 2 if self.n <= 64 else 4 if self.n <= 128 else 8
        # measured cycles per K step (profiles/r03/dw_phase_cycles.log): 3 300 on the fused fast path (full 256 x 256 operands),
        # 5 000 on the general path whatever the column count (the A operand's conversion dominates)
        # 5 000 on the general path whatever the column count (the A operand's conversion dominates); products with <= 4 columns
        # stream A once through plain FMAs (thin path)
        if self.n <= 4:
            return len(self.a) * 1.0       # (latency-bound per workgroup: it needs as many items as an MFMA job to fill the chip)
        full = self.m == 256 and self.n == 256
        if self.half:                      # half the bytes, a third of the MFMAs, no conversion
            return len(self.a) * 0.45
        return len(self.a) * (1.0 if full else 1.5)


def _rows(t: torch.Tensor, half: bool = False):
    """(pointer, leading dimension) of a [P, C] float32 view whose rows are contiguous (``half``: a contiguous float16 [P, 256])"""
    if half:
        if not (t.is_cuda and t.dtype == torch.float16 and t.dim() == 2 and t.shape[1] == 256 and t.is_contiguous()):
            raise ValueError("half dw operands must be contiguous float16 [P, 256] GPU arrays")
        return t.data_ptr(), 256
    if t.dim() == 1:
        t = t[:, None]
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
        raise ValueError(f"dw operand must be a float32 GPU matrix with contiguous rows, got {tuple(t.shape)} {t.dtype} strides {t.stride()}")
    return t.data_ptr(), int(t.stride(0)) if t.shape[0] > 1 else int(t.shape[1])


# Split-K workspace, one per device.  A captured training step (training.GraphedTrainStep) bakes its ADDRESS into the hipGraph,
# and the size a call needs depends on its job table (768 slabs for the fused table at 256 CUs, 769 for the autograd SDF table,
# far fewer for a small batch), so the buffer is sized ONCE per device to an upper bound that no job table of `run` can exceed -
# (items + MAX_JOBS) slots: every job's share is rounded, at most one extra slab each - and is never replaced afterwards.  Should
# a caller ask for more items than the default, a larger buffer is allocated and the old one stays referenced (a live graph may
# still write to it).
MAX_JOBS = 24                     # csrc/nrh_dw.hip
SLOT_FLOATS = 256 * 256 + 512     # one item's partial product + its column sums
_WS: Dict[str, torch.Tensor] = {}
_WS_RETIRED: List[torch.Tensor] = []


def _default_items(dev, npts: Optional[int] = None) -> int:
    # three work items per CU: the jobs' per-step costs are only modelled roughly and short items level the tail, while
    # every item costs 256 KiB of partial sums to write and reduce (measured on the 1024-ray job table: 1.70 / 2.05 / 1.56 /
    # 1.58 ms with 1 / 2 / 3 / 4 items per CU; profiles/r03/dw_bench_items.log).  Up to 512 rays (65 536 points) an item is a
    # few K steps long and its fixed cost dominates: ONE item per CU there (profiles/r04/dw_items.log: 0.131 against 0.195 ms at
    # 64 rays, 0.222 / 0.272 at 128, 0.407 / 0.436 at 256, 0.792 / 0.801 at 512).  ``npts`` None: the upper bound (workspace size).
    cus = max(1, torch.cuda.get_device_properties(dev).multi_processor_count)
    # A/B runs only (1 .. 3: the workspace is sized for 3); honoured only together with NRH_PROFILING=1, like the library's overrides
    per_cu = os.environ.get("NRH_DW_ITEMS_PER_CU") if os.environ.get("NRH_PROFILING") == "1" else None
    if per_cu and npts is not None:
        return cus * max(1, min(3, int(per_cu)))
    return cus if (npts is not None and npts <= 65536) else 3 * cus


def workspace(dev, need_floats: int = 0) -> torch.Tensor:
    """The device's split-K workspace (see above): at least the fixed bound, grown (never freed) only for oversize requests."""
    key = str(dev)
    bound = max(int(need_floats), (_default_items(dev) + MAX_JOBS) * SLOT_FLOATS)
    ws = _WS.get(key)
    if ws is None or ws.numel() < bound:
        if ws is not None:
            _WS_RETIRED.append(ws)
        _WS[key] = ws = torch.empty(bound, dtype=torch.float32, device=dev)
    return ws


def run(jobs: List[Job], npts: int, total_items: Optional[int] = None) -> None:
    """One nrh_dw_gemm call for all jobs (two launches: split-K products, deterministic reduction).  The K split gives every job
    a share of ``total_items`` workgroups (default: the device's CU count) in proportion to its cost."""
    lib = _lib.load()
    dev = jobs[0].a[0].device
    with torch.cuda.device(dev):
        if total_items is None:
            total_items = _default_items(dev, npts)
        nsteps = npts // 32
        costs = [j.cost() for j in jobs]
        tot = sum(costs)
        arr = (NrhDwJob * len(jobs))()
        keep = []
        for q, j, c in zip(arr, jobs, costs):
            for k, (a, b) in enumerate(zip(j.a, j.b)):
                if a.shape[0] != npts or b.shape[0] != npts:
                    raise ValueError("dw operands must have one row per point")
                q.a[k], q.lda[k] = _rows(a, j.half)
                q.b[k], q.ldb[k] = _rows(b, j.half)
                q.tiled_a[k], q.tiled_b[k] = int(j.tiled_a[k]), int(j.tiled_b[k])
            q.npairs, q.m, q.n = len(j.a), j.m, j.n
            q.half_ops = int(j.half)
            if j.dyn_scale is not None:
                if not (j.dyn_scale.is_cuda and j.dyn_scale.dtype == torch.float32 and j.dyn_scale.numel() >= 2):
                    raise ValueError("dyn_scale must be a float32 GPU tensor {S, 1 / S}")
                q.dyn_scale = j.dyn_scale.data_ptr()
            q.slabs = max(1, min(nsteps, int(round(total_items * c / tot))))
            if j.out is not None:
                if not (j.out.is_cuda and j.out.dtype == torch.float32 and j.out.is_contiguous()):
                    raise ValueError("dw output must be a contiguous float32 GPU tensor")
                q.out, q.ldo = j.out.data_ptr(), int(j.out.shape[-1])
            q.transpose, q.rows, q.cols, q.scale = int(j.transpose), j.rows, j.cols, float(j.scale)
            if j.col_map is not None:
                if j.col_map.dtype != torch.int32 or not j.col_map.is_cuda:
                    raise ValueError("col_map must be an int32 GPU tensor")
                q.col_map = j.col_map.data_ptr()
            if j.colsum_a is not None:
                q.colsum_a, q.scale_a = j.colsum_a.data_ptr(), float(j.scale_a)
            if j.colsum_b is not None:
                q.colsum_b, q.scale_b = j.colsum_b.data_ptr(), float(j.scale_b)
            keep.append(j)
        need = int(lib.nrh_dw_workspace_floats(arr, len(jobs)))
        if need < 0:
            raise ValueError("nrh_dw_workspace_floats: bad job table")
        ws = workspace(dev, need)
        rc = lib.nrh_dw_gemm(arr, len(jobs), npts, ctypes.c_void_p(ws.data_ptr()), ws.numel(), _lib.stream_handle())
        _lib.check(rc, "nrh_dw_gemm")


_ONES: Dict[tuple, torch.Tensor] = {}


def ones(n: int, device) -> torch.Tensor:
    key = (str(device), n)
    if key not in _ONES:
        _ONES[key] = torch.ones(n, dtype=torch.float32, device=device)
    return _ONES[key]


_MAPS: Dict[tuple, tuple] = {}


def color_col_maps(device, hints: bool):
    """packing.color_input_permutation as int32 device tensors: destination columns of the feature block / the other block"""
    key = (str(device), hints)
    if key not in _MAPS:
        from . import packing
        fi, mi = packing.color_input_permutation(hints)
        _MAPS[key] = (fi.to(torch.int32).to(device), mi.to(torch.int32).to(device))
    return _MAPS[key]


def arrays_tiled() -> bool:
    """nrh_train_arrays_tiled(): whether the SDF sweeps' h / t / abar / zbar are in the tiled layout (a build constant of the library)"""
    return bool(_lib.load().nrh_train_arrays_tiled())


def to_tiled(x: torch.Tensor) -> torch.Tensor:
    """Row-major [..., P, 256] -> the tiled layout of the training arrays, same shape and storage size: per tile of 16 points,
    [block 16][point 16][16 channels] (tests; the kernels write this layout themselves)."""
    P = x.shape[-2]
    assert P % 16 == 0 and x.shape[-1] == 256
    lead = x.shape[:-2]
    return x.reshape(*lead, P // 16, 16, 16, 16).transpose(-3, -2).contiguous().reshape(*lead, P, 256)


def to_half_tiled(x: torch.Tensor) -> torch.Tensor:
    """Row-major [..., P, 256] -> float16 in the half-tiled layout of the 16-bit hand-offs: per tile of 16 points,
    [block pair 8][point 16][quarter 4][block of the pair 2][4 channels] (8 KiB; tests - the kernels write it themselves)."""
    P = x.shape[-2]
    assert P % 16 == 0 and x.shape[-1] == 256
    lead = x.shape[:-2]
    v = x.to(torch.float16).reshape(*lead, P // 16, 16, 8, 2, 4, 4)          # tile, point, pair, e, quarter, r
    nd = len(lead)
    v = v.permute(*range(nd), nd, nd + 2, nd + 1, nd + 4, nd + 3, nd + 5)    # tile, pair, point, quarter, e, r
    return v.contiguous().reshape(*lead, P, 256)


def from_half_tiled(x: torch.Tensor) -> torch.Tensor:
    """inverse of ``to_half_tiled`` (float16 row-major)"""
    P = x.shape[-2]
    lead = x.shape[:-2]
    nd = len(lead)
    v = x.reshape(*lead, P // 16, 8, 16, 4, 2, 4).permute(*range(nd), nd, nd + 2, nd + 1, nd + 4, nd + 3, nd + 5)
    return v.contiguous().reshape(*lead, P, 256)


def from_tiled(x: torch.Tensor) -> torch.Tensor:
    """inverse of ``to_tiled`` (the permutation is an involution on the [point, block] axes)"""
    return to_tiled(x)


def sdf_jobs(shapes, h, t, zbar, abar, gebar, emb, sbar, fbar, out, half=None) -> List[Job]:
    """Jobs for the SDF network's 8 layers and 2 heads (the maths: nrhints_amd/sdf_function.py).  h, t, zbar, abar [8,P,256];
    gebar, emb [P,64]; sbar [P]; fbar [P,256]; ``out``: dict of preallocated gradient tensors dW0..7, db0..7, ws, bs, Wf, bf;
    ``shapes``: the dense weights' shapes (layer 3 has 217 rows).
    ``half``: dict(h16, t16, zbar16, abar16 float16 [8,P,256] half-tiled, dyn) - the 16-bit hand-offs of the f16x3 step: the seven
    256 x 256 two-pair products (layers 1..7) read those instead; of the float32 arrays only h[7], t[0], zbar[0], abar[7] hold data."""
    P = h.shape[1]
    T_ = arrays_tiled()       # h, t, zbar, abar as the sweep kernels wrote them; emb, gebar, fbar, sbar are row-major
    jobs = [Job([zbar[0], t[0]], [emb, gebar], 256, 39, out["dW0"], rows=shapes[0][0], colsum_a=out["db0"], tiled_a=(T_, T_))]
    for l in range(1, 8):
        if half is not None:
            jobs.append(Job([half["zbar16"][l], half["t16"][l]], [half["h16"][l - 1], half["abar16"][l - 1]], 256, 256, out[f"dW{l}"],
                            rows=shapes[l][0], scale=(1.0 / math.sqrt(2.0) if l == 4 else 1.0), colsum_a=out[f"db{l}"], half=True,
                            dyn_scale=half["dyn"]))
            continue
        jobs.append(Job([zbar[l], t[l]], [h[l - 1], abar[l - 1]], 256, 256, out[f"dW{l}"], rows=shapes[l][0],
                        scale=(1.0 / math.sqrt(2.0) if l == 4 else 1.0), colsum_a=out[f"db{l}"], tiled_a=(T_, T_), tiled_b=(T_, T_)))
    jobs.append(Job([fbar], [h[7]], 256, 256, out["Wf"], colsum_a=out["bf"], tiled_b=(T_,)))
    # d w_s = (h_7^T sbar + sum_p abar_7) / 3,  d b_s = sum(sbar) / 3   (sdf = (w_s . h_7 + b_s) / 3)
    jobs.append(Job([h[7], abar[7]], [sbar.reshape(P, 1), ones(P, h.device).reshape(P, 1)], 256, 1, out["ws"], transpose=True,
                    scale=1.0 / 3.0, colsum_b=out["bs"], scale_b=1.0 / 3.0, tiled_a=(T_, T_)))
    return jobs


def color_jobs(hints: bool, zbar, zbar4, save_h, feat, save_misc, out, half=None) -> List[Job]:
    """Jobs for the reflectance network's 5 layers.  zbar, save_h [4,P,256]; zbar4 [P,3]; feat [P,256]; save_misc [P,128|64];
    ``out``: w0 [256,361|316], w1..w3 [256,256], w4 [3,256], b0..b3 [256], b4 [3].
    ``half``: dict(zbar16, h16 float16 [4,P,256] half-tiled, inv_scale) - the 16-bit hand-offs: layers 1..3 read zbar16[l] (stored as
    1 / inv_scale x the value) and h16[l - 1]; of the float32 arrays only zbar[0] and save_h[3] hold data."""
    fi, mi = color_col_maps(zbar.device, hints)
    nm = 105 if hints else 60
    jobs = [Job([zbar[0]], [feat], 256, 256, out["w0"], col_map=fi, colsum_a=out["b0"]),
            Job([zbar[0]], [save_misc], 256, nm, out["w0"], col_map=mi)]
    for l in (1, 2, 3):
        if half is not None:
            jobs.append(Job([half["zbar16"][l]], [half["h16"][l - 1]], 256, 256, out[f"w{l}"], colsum_a=out[f"b{l}"], half=True,
                            scale=half["inv_scale"], scale_a=half["inv_scale"]))
            continue
        jobs.append(Job([zbar[l]], [save_h[l - 1]], 256, 256, out[f"w{l}"], colsum_a=out[f"b{l}"]))
    jobs.append(Job([save_h[3]], [zbar4], 256, 3, out["w4"], transpose=True, colsum_b=out["b4"]))
    return jobs
