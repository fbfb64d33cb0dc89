"""Numpy emulation of the gfx950 register-chain MLP kernels (csrc/nrh_sdf.hip, nrh_color.hip).

It executes the SAME packed buffers with the SAME lane/register index arithmetic as the HIP code
(v_mfma_f32_16x16x4_f32 fragment maps from /opt/skills/guides/cdna_hip_programming.md §3), in float64, so the CPU
test-suite can prove packing order, the skip-connection substitution, the reverse chain and the encoding
derivative before anything runs on a GPU.  Test infrastructure only.
"""
import math

import numpy as np

from nrhints_amd import packing as pk

LANES = np.arange(64)
J = LANES & 15
Q = LANES >> 4
HALF_PI32 = float(np.float32(math.pi / 2))


def mfma_16x16x4(a, b, c):
    """a, b: [64] per-lane operands; c: [4,64] accumulator (reg, lane). D = A.B + C with
    A[i][k] = a[lane = 16k + i], B[k][j] = b[lane = 16k + j], D[row = 4*(lane>>4) + r][col = lane & 15]."""
    A = a.reshape(4, 16).T          # [i, k]
    B = b.reshape(4, 16)            # [k, j]
    D = A @ B                       # [16 rows, 16 cols]
    out = c.copy()
    for r in range(4):
        out[r] += D[4 * Q + r, J]
    return out


def mfma_16x16x32(a, b, c):
    """a, b: [64, 8] per-lane fp16 operands (as float64 values); K slot (q, e) of lane q*16 + i/j.
    Any consistent slot -> k map gives the same result, which is all the kernels rely on."""
    A = a.reshape(4, 16, 8).transpose(1, 0, 2).reshape(16, 32)     # [i, (q,e)]
    B = b.reshape(4, 16, 8).transpose(0, 2, 1).reshape(32, 16)     # [(q,e), j]
    D = A @ B
    out = c.copy()
    for r in range(4):
        out[r] += D[4 * Q + r, J]
    return out


def split16(x):
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float64)) * 2048.0).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def run_stage_h3(packed16, KB, NCH, inp, init=None):
    """f16x3 stage: packed16 = fp16 buffer from packing.pack_stage_h3; inp [KB*4, 64] float values (split here
    exactly as Act<1>::set_chunk does)."""
    KS = KB // 2
    w = np.asarray(packed16, dtype=np.float64).reshape(NCH, 2, KS, 2, 64, 8)
    out = np.zeros((NCH * 8, 64))
    for ch in range(NCH):
        for obi in range(2):
            acc = np.zeros((4, 64)) if init is None else init[ch * 8 + obi * 4: ch * 8 + obi * 4 + 4].copy()
            cross = np.zeros((4, 64))
            for s in range(KS):
                vals = inp[s * 8: s * 8 + 8].T                      # [lane, 8]: blocks 2s (regs 0..3), 2s+1 (0..3)
                bh, bl = split16(vals)
                ah, al = w[ch, obi, s, 0], w[ch, obi, s, 1]
                acc = mfma_16x16x32(ah, bh, acc)
                cross = mfma_16x16x32(ah, bl, cross)
                cross = mfma_16x16x32(al, bh, cross)
            out[ch * 8 + obi * 4: ch * 8 + obi * 4 + 4] = acc + cross / 2048.0
    return out


def run_stage(packed, KB, NCH, inp, init=None):
    """packed: flat stage buffer; inp: [KB*4, 64]; returns out [NCH*8, 64] (D-layout).
    A float16 buffer selects the f16x3 emulation."""
    if packed.dtype == np.float16:
        return run_stage_h3(packed, KB, NCH, inp, init)
    w = packed.reshape(NCH, 2, KB, 64, 4)
    out = np.zeros((NCH * 8, 64))
    for ch in range(NCH):
        for obi in range(2):
            acc = np.zeros((4, 64)) if init is None else init[ch * 8 + obi * 4: ch * 8 + obi * 4 + 4].copy()
            for kb in range(KB):
                a4 = w[ch, obi, kb]          # [lane, c]
                for c in range(4):
                    acc = mfma_16x16x4(a4[:, c], inp[kb * 4 + c], acc)
            out[ch * 8 + obi * 4: ch * 8 + obi * 4 + 4] = acc
    return out


def bias_regs(bias, layer, nblocks=16):
    """bias value for reg [b*4+r] at each lane: bias[layer*256 + 16b + 4q + r]."""
    out = np.zeros((nblocks * 4, 64))
    for b in range(nblocks):
        for r in range(4):
            out[b * 4 + r] = bias[layer * 256 + 16 * b + 4 * Q + r]
    return out


def enc_all(x3):
    """[npts,3] -> [npts,39] with the float32-rounded pi/2 phase, as the kernel does."""
    out = [x3[:, 0], x3[:, 1], x3[:, 2]]
    s_part, c_part = [], []
    for d in range(3):
        for k in range(6):
            s = x3[:, d] * (1 << k)
            s_part.append(np.sin(s))
            c_part.append(np.sin(s + HALF_PI32))
    return np.stack(out + s_part + c_part, axis=1)


def enc_dall(x3):
    out = [np.ones_like(x3[:, 0])] * 3
    s_part, c_part = [], []
    for d in range(3):
        for k in range(6):
            fr = float(1 << k)
            s = x3[:, d] * fr
            s_part.append(np.cos(s) * fr)
            c_part.append(np.cos(s + HALF_PI32) * fr)
    return np.stack(out + s_part + c_part, axis=1)


def enc_dim(e):
    return e if e < 3 else ((e - 3) % 18) // 6


def softplus100(z):
    t = z * 100.0
    e = np.exp(np.minimum(t, 50.0))
    h = np.where(t > 20.0, z, np.log1p(e) / 100.0)
    d = np.where(t > 20.0, 1.0, e / (e + 1.0))
    return h, d


def sdf_tile(packed, bias, head, pts16, mode):
    """One wave's work: pts16 [16,3] -> (sdf [16], grad [16,3] or None, feat [16,256] or None)."""
    packed = np.asarray(packed)
    sc = 2 if packed.dtype == np.float16 else 1      # fp16 pairs: two elements per fp32-equivalent slot
    if sc == 1:
        packed = packed.astype(np.float64)

    packed_ = packed

    class _Buf:
        def __getitem__(self, sl):
            return packed_[sl.start * sc: sl.stop * sc]
    packed = _Buf()
    bias = np.asarray(bias, dtype=np.float64)
    head = np.asarray(head, dtype=np.float64)
    x3 = pts16[J] * 3.0                     # per lane [64,3]
    allv = enc_all(x3)                      # [64,39]
    emb = np.zeros((16, 64))
    for b in range(4):
        for r in range(4):
            e = 16 * b + 4 * Q + r
            emb[b * 4 + r] = np.where(e < 39, allv[LANES, np.minimum(e, 38)], 0.0)
    sig = {}
    off = 0
    out = run_stage(packed[off: off + pk.SDF_L0_FLOATS], 4, 8, emb)
    off += pk.SDF_L0_FLOATS
    h, sig[0] = softplus100(out + bias_regs(bias, 0))
    head_part = None
    for s in range(1, 8):
        out = run_stage(packed[off: off + pk.SDF_REG_FLOATS], 16, 8, h)
        off += pk.SDF_REG_FLOATS
        ho, d = softplus100(out + bias_regs(bias, s))
        if s == 3:
            for b in (13, 14, 15):
                for r in range(4):
                    e = 16 * b + 4 * Q + r - 217
                    m = e >= 0
                    ho[b * 4 + r] = np.where(m, allv[LANES, np.clip(e, 0, 38)], ho[b * 4 + r])
                    d[b * 4 + r] = np.where(m, 0.0, d[b * 4 + r])
        if s == 7:
            wsr = bias_regs(head[:256], 0)
            head_part = (wsr * ho).sum(0)
            d = d * (wsr / 3.0)
        sig[s] = d
        h = ho
    part = head_part.reshape(4, 16).sum(0)           # reduce over q
    sdf = (part + head[256]) / 3.0
    feat = None
    if mode == 2:
        out = run_stage(packed[off: off + pk.SDF_REG_FLOATS], 16, 8, h) + bias_regs(bias, 8)
        feat = np.zeros((16, 256))
        for b in range(16):
            for r in range(4):
                feat[J, 16 * b + 4 * Q + r] = out[b * 4 + r]
    off += pk.SDF_REG_FLOATS
    if mode == 0:
        return sdf, None, None
    g = sig[7].copy()
    skip = np.zeros((12, 64))
    for l in range(7, 0, -1):
        out = run_stage(packed[off: off + pk.SDF_REG_FLOATS], 16, 8, g)
        off += pk.SDF_REG_FLOATS
        if l == 4:
            skip = out[52:64].copy()
        g = out * sig[l - 1]
    ge = run_stage(packed[off: off + pk.SDF_R0_FLOATS], 16, 2, g)
    dc = enc_dall(x3)                       # [64,39]
    dx = np.zeros((3, 64))
    for b in range(3):
        for r in range(4):
            for qq in range(4):
                e = 16 * b + 4 * qq + r
                if e < 39:
                    dx[enc_dim(e)] += np.where(Q == qq, ge[b * 4 + r] * dc[:, e], 0.0)
                es = 16 * (b + 13) + 4 * qq + r - 217
                if 0 <= es < 39:
                    dx[enc_dim(es)] += np.where(Q == qq, skip[b * 4 + r] * dc[:, es], 0.0)
    grad = dx.reshape(3, 4, 16).sum(1).T * 3.0       # [16,3]
    return sdf, grad, feat


def color_tile(packed, bias, feat16, misc16):
    """feat16 [16,256], misc16 [16,105] (kernel order) -> rgb [16,3]."""
    packed = np.asarray(packed)
    sc = 2 if packed.dtype == np.float16 else 1
    if sc == 1:
        packed = packed.astype(np.float64)
    packed_ = packed

    class _Buf:
        def __getitem__(self, sl):
            return packed_[sl.start * sc: sl.stop * sc]
    packed = _Buf()
    bias = np.asarray(bias, dtype=np.float64)
    h = np.zeros((64, 64))
    for b in range(16):
        for r in range(4):
            h[b * 4 + r] = feat16[J, 16 * b + 4 * Q + r]
    off = 0
    part = run_stage(packed[off: off + pk.SDF_REG_FLOATS], 16, 8, h)
    off += pk.SDF_REG_FLOATS
    misc = np.zeros((32, 64))
    for b in range(8):
        for r in range(4):
            m = 16 * b + 4 * Q + r
            misc[b * 4 + r] = np.where(m < 105, misc16[J, np.minimum(m, 104)], 0.0)
    out = run_stage(packed[off: off + pk.COL_C0B_FLOATS], 8, 8, misc, init=part)
    off += pk.COL_C0B_FLOATS
    h = np.maximum(out + bias_regs(bias, 0), 0.0)
    for l in (1, 2, 3):
        out = run_stage(packed[off: off + pk.SDF_REG_FLOATS], 16, 8, h)
        off += pk.SDF_REG_FLOATS
        h = np.maximum(out + bias_regs(bias, l), 0.0)
    out = run_stage(packed[off: off + 2 * 16 * 256], 16, 1, h)
    b4 = bias[4 * 256: 4 * 256 + 16]
    rgb = np.zeros((16, 3))
    for r in range(3):
        v = out[r] + b4[4 * Q + r]
        rgb[J[Q == 0], r] = 1.0 / (1.0 + np.exp(-v[Q == 0]))
    return rgb
