"""Host-side packing for the wide (32-point tile, one wave per SIMD) f16x3 kernels: csrc/nrh_mlp32.h, nrh_sdf32.hip.

Every GEMM stage is cut into chunks of 32 output rows; a chunk's LDS image is ``[K step s][hi | lo][lane 64]`` x 8 fp16:

    image[s][part][lane = 32 hf + row][i] = part(W[32 chunk + row][col32(s, hf, i)]),   col32 = 16 s + (i & 3) + 8 (i >> 2) + 4 hf

i.e. the A operand of ``v_mfma_f32_32x32x16_f16`` for K step s, with the K slots enumerated the way the C/D fragment of the
previous layer lands in registers (no shuffle between layers).  hi = fp16(w), lo = fp16((w - hi) * 2^11).

The kernels read ONE stream of chunks per evaluation mode, in execution order (csrc/nrh_sdf32.hip):

    E4 (48 KiB, resident in LDS) | L0 x2 blocks (four 8 KiB chunks each) | L1..L7 x56 | [FEAT x8] | HEAD x1 |
    [R7..R4 x32 | R4e x2 | R3..R1 x24 | R0 x2]                                  (blocks of 32 KiB)

E4 = W4[:, 217:] / sqrt2 is the skip connection's part of layer 4 (applied to the 39 embedding entries, in the embedding's K
order, 3 K steps per chunk); W4's main part has those columns zeroed.  The softplus layers work in the scaled domain t = z * 100/ln2,
u = h * 100/ln2 (see nrh_mlp32.h): L0 and E4 carry the factor 100/ln2, the head and the feature layer its inverse, biases
are scaled, L1..L7 are unchanged.
Torch ops only; not differentiable (evaluation kernels).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

from .packing import split_f16

IK = 100.0 / math.log(2.0)
KK = math.log(2.0) / 100.0
NTAB = 11
SMALL_KS, BIG_KS = 4, 16            # K steps stored per chunk (E4/L0 use 3 of their 4)
SCRATCH_WORDS_PER_WAVE = 8 * 8 * 2 * 64 * 4
WAVES, TILE = 4, 32
GROUP = WAVES * TILE


RESIDENT_BYTES = 49152


def stream_blocks(mode: int) -> int:
    return 2 + 56 + 1 + (8 if mode == 2 else 0) + (60 if mode >= 1 else 0)


def stream_bytes(mode: int) -> int:
    return RESIDENT_BYTES + stream_blocks(mode) * 32768


def _cols(ks: int) -> torch.Tensor:
    """col32(s, hf, i) as an index tensor [ks, 2, 8]."""
    s = torch.arange(ks).view(ks, 1, 1)
    hf = torch.arange(2).view(1, 2, 1)
    i = torch.arange(8).view(1, 1, 8)
    return 16 * s + (i & 3) + 8 * (i >> 2) + 4 * hf


def _layout32(wp: torch.Tensor, rows: int, ks: int) -> torch.Tensor:
    """Element order of ONE part (hi or lo) of a stage: [chunk][s][lane = 32 hf + row][i]; wp is [rows, 16 ks]."""
    g = wp[:, _cols(ks).to(wp.device)]                    # [rows, ks, 2, 8]
    return g.reshape(rows // 32, 32, ks, 2, 8).permute(0, 2, 3, 1, 4).reshape(-1)


def _pad(w: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    assert w.shape[0] <= rows and w.shape[1] <= cols
    return torch.nn.functional.pad(w, (0, cols - w.shape[1], 0, rows - w.shape[0]))


def pack_stage32(w: torch.Tensor, rows: int, ks: int) -> torch.Tensor:
    """Dense [out, in] (zero-padded to [rows, 16 ks]) -> fp16 [rows/32 chunks][ks][hi | lo][64][8] flattened."""
    assert rows % 32 == 0
    lay = _layout32(_pad(w.detach().float(), rows, 16 * ks), rows, ks)
    hi, lo = split_f16(lay)
    return torch.stack([hi.reshape(-1, 512), lo.reshape(-1, 512)], dim=1).reshape(-1)


def sdf32_tables(d: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[NTAB, 256] float32: 0..7 the bias rows of the eight layers as ONE PACKED fp16 PAIR per output row,
    bits (b_hi | b_lo * 2^11 << 16) of b_l * IK (rows >= 217 of layer 3 zero) - the kernel adds them through one extra MFMA
    per window (csrc/gen_mlp32.py, bias_mfma); 8 b_feat in the same packed form, 9 {b_s / 3, 0, ...}, 10 w_s / 3."""
    f = lambda t: t.detach().float()
    rows = []
    for l in range(8):
        b = torch.nn.functional.pad(f(d[f"sdf_b{l}"]) * IK, (0, 256 - d[f"sdf_b{l}"].shape[0]))
        hi, lo = split_f16(b)       # b = hi + lo / 2^11
        b = (hi.view(torch.int16).to(torch.int32) & 0xffff | (lo.view(torch.int16).to(torch.int32) << 16)).view(torch.float32)
        rows.append(b)
    hi, lo = split_f16(f(d["feat_b"]))             # the feature head takes its bias the same way (row 8)
    rows.append((hi.view(torch.int16).to(torch.int32) & 0xffff | (lo.view(torch.int16).to(torch.int32) << 16)).view(torch.float32))
    rows.append(torch.nn.functional.pad(f(d["sdf_head_b"]).reshape(1) / 3.0, (0, 255)))
    rows.append(f(d["sdf_head_w"]).reshape(256) / 3.0)
    return torch.stack(rows).contiguous()


def tables_in_f16_range(d: Dict[str, torch.Tensor]) -> torch.Tensor:
    """0-dim bool tensor: every packed bias fits fp16 (|b_l * IK| < 65504)."""
    return torch.stack([(d[f"sdf_b{l}"].detach().float().abs().max() * IK) < 65504.0 for l in range(0, 8)]).all()


_SDF32_KEYS = [f"sdf_w{l}" for l in range(8)] + ["feat_w", "sdf_head_w"]


def _stage_matrices(d: Dict[str, torch.Tensor], scaled: bool):
    """name -> (dense [rows, 16 ks] matrix, rows, ks) of every stage, from the folded weights d.  scaled = False leaves out
    every multiplication (used to lay out element indices); the scale factors are recovered by feeding all-ones."""
    sc = (lambda t, k: t * k) if scaled else (lambda t, k: t)
    w = [d[f"sdf_w{l}"] for l in range(8)]
    r2 = 1.0 / math.sqrt(2.0)
    # (one multiplication per element everywhere, so that the index plan below reproduces the direct packer bit for bit)
    w4m = _pad(sc(w[4][:, :217], r2), 256, 256)                     # main part of layer 4: columns 217.. are zero
    w4e_raw = w[4][:, 217:256]                                      # skip part: applied to the 39 embedding entries
    w4e = sc(w4e_raw, r2)
    fwd = [w[0], w[1], w[2], w[3], w4m, w[5], w[6], w[7]]
    m = {"E4": (sc(w4e_raw, r2 * IK), 256, 3), "L0": (sc(w[0], IK), 256, SMALL_KS)}
    for l in range(1, 8):
        m[f"L{l}"] = (fwd[l], 256, BIG_KS)
    m["FEAT"] = (sc(d["feat_w"], KK), 256, BIG_KS)
    m["HEAD"] = (sc(d["sdf_head_w"].reshape(1, 256), KK / 3.0), 32, BIG_KS)
    for l in range(1, 8):
        m[f"R{l}"] = (fwd[l].t(), 256, BIG_KS)
    m["R4e"] = (w4e.t(), 64, BIG_KS)
    m["R0"] = (w[0].t(), 64, BIG_KS)
    return m


def sdf32_pieces(d: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    f = {k: d[k].detach().float() for k in _SDF32_KEYS}
    return {k: pack_stage32(w, rows, ks) for k, (w, rows, ks) in _stage_matrices(f, True).items()}


def stream_order(mode: int):
    order = ["E4", "L0"] + [f"L{l}" for l in range(1, 8)]
    if mode == 2:
        order.append("FEAT")
    order.append("HEAD")
    if mode >= 1:
        order += ["R7", "R6", "R5", "R4", "R4e", "R3", "R2", "R1", "R0"]
    return order


_PLANS = {}


def pack_sdf32(d: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (streams: fp16 tensor holding the three mode streams back to back [mode 0 | mode 1 | mode 2], tables [11, 256]).
    GPU tensors go through the index plan and its HIP launches (PackPlan32.pack), so that every stream set of a renderer - plain
    and fused feature head - is packed by the same arithmetic: torch's float32 -> float16 conversion on the GPU does not round
    the residuals' fp16 subnormals like the CPU's (and like nrh_pack_gather) - measured: renders with the two stream sets
    stopped being bit-equal when only one of them came from torch ops."""
    if d["sdf_w0"].is_cuda:
        key = str(d["sdf_w0"].device)
        plan = _PLANS.get(key)
        if plan is None or not plan.matches(d):
            plan = _PLANS[key] = PackPlan32(d)
        return plan.pack(d)
    p = sdf32_pieces(d)
    streams = []
    for mode in range(3):
        s = torch.cat([p[k] for k in stream_order(mode)])
        assert s.numel() * 2 == stream_bytes(mode), (mode, s.numel() * 2, stream_bytes(mode))
        streams.append(s)
    return torch.cat(streams).contiguous(), sdf32_tables(d)


def fuse_feature_head(d: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The feature head of the SDF net and the feature block of the reflectance net's first layer are two linear maps in a
    row (fields/sdf_field.py:119-123 -> fields/reflectance_network.py:77-84, columns 60:316 of the 316 / 361-wide input):
    replace (W_feat, b_feat) by (W0[:, 60:316] W_feat, W0[:, 60:316] b_feat), evaluated in fp64.  The mode-2 kernel then
    returns that block's contribution to the first hidden layer and the reflectance kernel skips it (NrhNet.feat_fused)."""
    out = dict(d)
    w0 = d["col_w0"].detach()
    if w0.is_cuda:
        # on the device: one small HIP launch (csrc/nrh_fold.hip fuse_head_kernel) instead of two float64 library GEMM calls
        from . import _lib
        lib = _lib.load()
        f32c = lambda t: t.detach().to(torch.float32).contiguous()
        w0c, fw, fb = f32c(w0), f32c(d["feat_w"]), f32c(d["feat_b"])
        ow, ob = torch.empty(256, 256, dtype=torch.float32, device=w0.device), torch.empty(256, dtype=torch.float32, device=w0.device)
        with torch.cuda.device(w0.device):
            rc = lib.nrh_fuse_feature_head(_lib.ptr(w0c), int(w0c.shape[1]), _lib.ptr(fw), _lib.ptr(fb), _lib.ptr(ow), _lib.ptr(ob),
                                           _lib.stream_handle())
        _lib.check(rc, "nrh_fuse_feature_head")
        out["feat_w"], out["feat_b"] = ow, ob
        return out
    w0f = w0.double()[:, 60:316]
    out["feat_w"] = (w0f @ d["feat_w"].detach().double()).float()
    out["feat_b"] = (w0f @ d["feat_b"].detach().double()).float()
    return out


def pack_sdf32_fused(d: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """pack_sdf32 of the network with the fused feature head (evaluation renders only; modes 0 and 1 are unchanged)."""
    return pack_sdf32(fuse_feature_head(d))


# ---- the reflectance net on the wide machinery (csrc/nrh_color32.hip) -------------------------------------------------------
COLOR32_BLOCKS = 33
COLOR32_NTAB = 5


def color32_stream_bytes() -> int:
    return COLOR32_BLOCKS * 32768


def _raymisc_columns() -> torch.Tensor:
    """Column of the reference's 361-wide reflectance input (fields/reflectance_network.py:77-82) for each of the 99 per-ray
    values in the kernels' raymisc order [enc(view) 27 | enc(pl) 27 | enc(vis) 9 | enc(cue) 36]."""
    return torch.cat([torch.arange(3, 30), torch.arange(33, 60), torch.arange(316, 325), torch.arange(325, 361)])


def color32_layer0(d: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[256, 128]: layer 0 without its feature block in the kernel's column order - 0..2 point, 4..6 normal, 16 + m raymisc[m]
    (csrc/nrh_color32.hip header); the feature block (columns 60:316) comes in through packing32.fuse_feature_head."""
    w0 = d["col_w0"].detach().float()
    if w0.shape[1] != 361:
        raise ValueError("pack_color32: the wide reflectance kernel is built for the hinted model (361 inputs)")
    m0 = torch.zeros(256, 128, dtype=torch.float32, device=w0.device)
    m0[:, 0:3] = w0[:, 0:3]
    m0[:, 4:7] = w0[:, 30:33]
    m0[:, 16:16 + 99] = w0[:, _raymisc_columns().to(w0.device)]
    return m0


def pack_color32(d: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (stream: fp16 tensor, 33 blocks of 32 KiB: C0 (8; K steps 8..15 zero), C1, C2, C3 (8 each), C4 (1; rows 3..31 zero);
    tables [5, 256] float32: rows 0..3 the biases of C0..C3 as packed fp16 pairs (see sdf32_tables), row 4 b4)."""
    f = lambda t: t.detach().float()
    blocks = [pack_stage32(color32_layer0(d), 256, BIG_KS)]
    blocks += [pack_stage32(f(d[f"col_w{l}"]), 256, BIG_KS) for l in (1, 2, 3)]
    blocks.append(pack_stage32(f(d["col_w4"]), 32, BIG_KS))
    stream = torch.cat(blocks).contiguous()
    assert stream.numel() * 2 == color32_stream_bytes(), stream.numel()
    rows = []
    for l in range(4):
        hi, lo = split_f16(f(d[f"col_b{l}"]))
        rows.append((hi.view(torch.int16).to(torch.int32) & 0xffff | (lo.view(torch.int16).to(torch.int32) << 16)).view(torch.float32))
    rows.append(torch.nn.functional.pad(f(d["col_b4"]), (0, 256 - d["col_b4"].shape[0])))
    return stream, torch.stack(rows).contiguous()


def stream_offset_bytes(mode: int) -> int:
    return sum(stream_bytes(m) for m in range(mode))


class PackPlan32:
    """Index form of pack_sdf32: the packers are index permutations plus constant factors, so they run ONCE on element
    indices and on all-ones; afterwards a re-pack is one gather, one multiply and the fp16 split (a training step
    re-packs after every optimiser step, also inside a captured hipGraph: no host data, no new index tensors)."""

    def __init__(self, d: Dict[str, torch.Tensor]):
        dev = d["sdf_w0"].device
        self.shapes = {k: tuple(d[k].shape) for k in _SDF32_KEYS}
        idx, ones, off = {}, {}, 1
        for k in _SDF32_KEYS:
            n = int(torch.tensor(self.shapes[k]).prod())
            idx[k] = (torch.arange(n, dtype=torch.float64) + off).reshape(self.shapes[k])
            ones[k] = torch.ones(self.shapes[k], dtype=torch.float64)
            off += n
        mi, ms = _stage_matrices(idx, False), _stage_matrices(ones, True)
        lay = lambda m, k: _layout32(_pad(m[k][0], m[k][1], 16 * m[k][2]), m[k][1], m[k][2])
        order = [k for mode in range(3) for k in stream_order(mode)]
        self.index = torch.cat([lay(mi, k) for k in order]).to(torch.int64).to(dev)
        self.scale = torch.cat([lay(ms, k) for k in order]).to(torch.float32).to(dev)
        assert self.index.numel() * 4 == sum(stream_bytes(m) for m in range(3))

    def matches(self, d) -> bool:
        return all(tuple(d[k].shape) == self.shapes[k] for k in _SDF32_KEYS) and d["sdf_w0"].device == self.index.device

    def pack(self, d: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
        dev = self.index.device
        from .packing import zero1
        flat = torch.cat([zero1(dev)] + [d[k].detach().to(torch.float32).reshape(-1) for k in _SDF32_KEYS])
        if dev.type == "cuda":
            # one launch for the streams, one for the tables (csrc/nrh_fold.hip); below: the same plan in torch ops (host side)
            import ctypes
            from . import _lib
            lib = _lib.load()
            if getattr(self, "_index32", None) is None:
                self._index32 = self.index.to(torch.int32)
            n = self.index.numel()
            streams = torch.empty(2 * n, dtype=torch.float16, device=dev)
            tables = torch.empty(NTAB, 256, dtype=torch.float32, device=dev)
            P = _lib.ptr
            f32c = lambda t: t.detach().to(torch.float32).contiguous()
            bias = [f32c(d[f"sdf_b{l}"]) for l in range(8)]
            fb, hb, hw = f32c(d["feat_b"]), f32c(d["sdf_head_b"]).reshape(-1), f32c(d["sdf_head_w"]).reshape(-1)
            with torch.cuda.device(dev):
                _lib.check(lib.nrh_pack_gather(P(flat), P(self._index32, torch.int32), P(self.scale), n, 2, P(streams, torch.float16),
                                               _lib.stream_handle()), "nrh_pack_gather")
                _lib.check(lib.nrh_sdf32_tables((ctypes.c_void_p * 8)(*[b.data_ptr() for b in bias]), (ctypes.c_int * 8)(*[b.numel() for b in bias]),
                                                P(fb), P(hb), P(hw), P(tables), _lib.stream_handle()), "nrh_sdf32_tables")
            return streams, tables
        hi, lo = split_f16(flat[self.index] * self.scale)
        streams = torch.stack([hi.reshape(-1, 512), lo.reshape(-1, 512)], dim=1).reshape(-1)
        return streams, sdf32_tables(d)
