"""Host-side packing for the wide (32-point tile, one wave per SIMD) f16x3 kernels: csrc/nrh_mlp32.h, nrh_sdf32.hip.

Every GEMM stage is cut into chunks of 32 output rows; a chunk's LDS image is ``[K step s][hi | lo][lane 64]`` x 8 fp16:

    image[s][part][lane = 32 hf + row][i] = part(W[32 chunk + row][col32(s, hf, i)]),   col32 = 16 s + (i & 3) + 8 (i >> 2) + 4 hf

i.e. the A operand of ``v_mfma_f32_32x32x16_f16`` for K step s, with the K slots enumerated the way the C/D fragment of the
previous layer lands in registers (no shuffle between layers).  hi = fp16(w), lo = fp16((w - hi) * 2^11).

The kernels read ONE stream of chunks per evaluation mode, in execution order (csrc/nrh_sdf32.hip):

    E4 (48 KiB, resident in LDS) | L0 x2 blocks (four 8 KiB chunks each) | L1..L7 x56 | [FEAT x8] | HEAD x1 |
    [R7 R6 R5 x24 | R4e x2 | R4..R1 x32 | R0 x2]                                  (blocks of 32 KiB)

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


def pack_stage32(w: torch.Tensor, rows: int, ks: int) -> torch.Tensor:
    """Dense [out, in] (zero-padded to [rows, 16 ks]) -> fp16 [rows/32 chunks][ks][2][64][8] flattened."""
    assert rows % 32 == 0 and w.shape[0] <= rows and w.shape[1] <= 16 * ks
    wp = torch.nn.functional.pad(w.detach().float(), (0, 16 * ks - w.shape[1], 0, rows - w.shape[0]))
    g = wp[:, _cols(ks).to(wp.device)]                    # [rows, ks, 2, 8]
    g = g.reshape(rows // 32, 32, ks, 2, 8).permute(0, 2, 3, 1, 4)      # [chunk, s, hf, row, i]; lane = 32 hf + row
    g = g.reshape(rows // 32, ks, 64, 8)
    hi, lo = split_f16(g)
    return torch.stack([hi, lo], dim=2).reshape(-1)       # [chunk, s, part, lane, i]


def sdf32_tables(d: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[NTAB, 256] float32: 0..7 b_l * IK (rows >= 217 of layer 3 zero), 8 b_feat, 9 {b_s / 3, 0, ...}, 10 w_s / 3."""
    f = lambda t: t.detach().float()
    rows = []
    for l in range(8):
        b = f(d[f"sdf_b{l}"]) * IK
        rows.append(torch.nn.functional.pad(b, (0, 256 - b.shape[0])))
    rows.append(f(d["feat_b"]))
    rows.append(torch.nn.functional.pad(f(d["sdf_head_b"]).reshape(1) / 3.0, (0, 255)))
    rows.append(f(d["sdf_head_w"]).reshape(256) / 3.0)
    return torch.stack(rows).contiguous()


def sdf32_pieces(d: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    f = lambda t: t.detach().float()
    w = [f(d[f"sdf_w{l}"]) for l in range(8)]
    r2 = math.sqrt(2.0)
    w4m = torch.nn.functional.pad(w[4][:, :217] / r2, (0, 39))      # main part of layer 4: columns 217.. are zero
    w4e = w[4][:, 217:256] / r2                                    # skip part: applied to the 39 embedding entries
    fwd = [w[0], w[1], w[2], w[3], w4m, w[5], w[6], w[7]]
    p = {}
    p["E4"] = pack_stage32(w4e * IK, 256, 3)                     # 6 KiB chunks, resident
    p["L0"] = pack_stage32(w[0] * IK, 256, SMALL_KS)            # 8 KiB chunks (4 K steps stored, 3 used): 2 blocks
    for l in range(1, 8):
        p[f"L{l}"] = pack_stage32(fwd[l], 256, BIG_KS)
    p["FEAT"] = pack_stage32(f(d["feat_w"]) * KK, 256, BIG_KS)
    p["HEAD"] = pack_stage32(f(d["sdf_head_w"]).reshape(1, 256) * (KK / 3.0), 32, BIG_KS)
    for l in range(1, 8):
        p[f"R{l}"] = pack_stage32(fwd[l].t(), 256, BIG_KS)
    p["R4e"] = pack_stage32(w4e.t(), 64, BIG_KS)
    p["R0"] = pack_stage32(w[0].t(), 64, BIG_KS)
    return p


def stream_order(mode: int):
    order = ["E4", "L0"] + [f"L{l}" for l in range(1, 8)]
    if mode == 2:
        order.append("FEAT")
    order.append("HEAD")
    if mode >= 1:
        order += ["R7", "R6", "R5", "R4e", "R4", "R3", "R2", "R1", "R0"]
    return order


def pack_sdf32(d: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (streams: fp16 tensor holding the three mode streams back to back [mode 0 | mode 1 | mode 2], tables [11, 256])."""
    p = sdf32_pieces(d)
    streams = []
    for mode in range(3):
        s = torch.cat([p[k] for k in stream_order(mode)])
        assert s.numel() * 2 == stream_bytes(mode), (mode, s.numel() * 2, stream_bytes(mode))
        streams.append(s)
    return torch.cat(streams).contiguous(), sdf32_tables(d)


def stream_offset_bytes(mode: int) -> int:
    return sum(stream_bytes(m) for m in range(mode))
