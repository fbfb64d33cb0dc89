"""Numpy emulation of the wide f16x3 SDF kernel (csrc/nrh_sdf32.hip on csrc/nrh_mlp32.h): the SAME packed stream, lane /
register / K-slot arithmetic, scaled softplus domain, hi/lo splits and unorm16 sigma' hand-off, in float64.  It lets the
CPU suite prove packing order and the stage plan (E4 start values, zeroed skip columns, R4e, the encoding derivative)
before anything runs on a GPU.  Test infrastructure only."""
import math

import numpy as np

LANES = np.arange(64)
J = LANES & 31
HF = LANES >> 5
HALF_PI32 = float(np.float32(math.pi / 2))
IK = 100.0 / math.log(2.0)


def frow(r, hf):
    return (r & 3) + 8 * (r >> 2) + 4 * hf


def col32(s, hf, i):
    return 16 * s + (i & 3) + 8 * (i >> 2) + 4 * hf


def mfma_32x32x16(a, b, c):
    """a, b: [64, 8] per-lane operands; c: [16, 64] (reg, lane).  D[row][col] = sum over K slots (hf, i) of
    A[lane = 32 hf + row][i] * B[lane = 32 hf + col][i];  D register r of lane (hf, col) is row frow(r, hf)."""
    A = a.reshape(2, 32, 8).transpose(1, 0, 2).reshape(32, 16)
    B = b.reshape(2, 32, 8).transpose(0, 2, 1).reshape(16, 32)
    D = A @ B
    out = c.copy()
    for r in range(16):
        out[r] += D[frow(r, HF), J]
    return out


def split16(x):
    with np.errstate(over="ignore"):
        hi = x.astype(np.float16)
        lo = (x - hi.astype(np.float64)).astype(np.float16)      # activations: unscaled residual (fp16 subnormals count)
    return hi.astype(np.float64), lo.astype(np.float64)


def enc_entry(x3, e):
    """entry e of enc_6(x3) per point, x3 [32,3]; the cosine half as sin(x + float32(pi/2)) like the reference."""
    if e < 0 or e >= 39:
        return np.zeros(x3.shape[0])
    if e < 3:
        return x3[:, e]
    idx = e - 3
    ph = HALF_PI32 if idx >= 18 else 0.0
    idx %= 18
    return np.sin(x3[:, idx // 6] * (1 << (idx % 6)) + ph)


def enc_dentry(x3, e):
    if e < 0 or e >= 39:
        return np.zeros(x3.shape[0])
    if e < 3:
        return np.ones(x3.shape[0])
    idx = e - 3
    ph = HALF_PI32 if idx >= 18 else 0.0
    idx %= 18
    fr = float(1 << (idx % 6))
    return np.cos(x3[:, idx // 6] * fr + ph) * fr


def emb_dim(e):
    return e if e < 3 else ((e - 3) % 18) // 6


class Stream:
    def __init__(self, buf16):
        self.buf = np.asarray(buf16, dtype=np.float64)
        self.pos = 0

    def chunk(self, ks_stored):
        n = ks_stored * 2 * 64 * 8
        c = self.buf[self.pos:self.pos + n].reshape(ks_stored, 2, 64, 8)
        self.pos += n
        return c


def kloop(chunk, ks, bh, bl, hh):
    """hh: [16,64] start values.  Returns hh + cc / 2048 pieces (hh, cc) after ks K steps; bh, bl: [ks][64,8]."""
    cc = np.zeros((16, 64))
    hh = hh.copy()
    for s in range(ks):
        ah, al = chunk[s, 0], chunk[s, 1]
        hh = mfma_32x32x16(ah, bh[s], hh)
        hh = mfma_32x32x16(ah, bl[s], hh)          # B_lo is unscaled
        cc = mfma_32x32x16(al, bh[s], cc)          # A_lo is scaled by 2^11
    return hh, cc


def tab_init(tables, table, c):
    """[16,64]: register r of lane (hf, j) <- tables[table][32 c + frow(r, hf)]"""
    out = np.zeros((16, 64))
    for r in range(16):
        out[r] = tables[table][32 * c + frow(r, HF)]
    return out


def act_to_b(u):
    """u: [8 chunks][16 regs, 64 lanes] -> per K step (16 of them) hi/lo [64,8]: step 2c+t <- registers 8t..8t+7."""
    bh, bl = [], []
    for c in range(8):
        for t in range(2):
            v = u[c][8 * t: 8 * t + 8].T            # [64, 8]
            h, l = split16(v)
            bh.append(h)
            bl.append(l)
    return bh, bl


def unorm16(q):
    return np.rint(np.clip(q, 0.0, 1.0) * 65535.0)


def sdf32_tile(stream16, tables, pts, mode, trace=None):
    """One 32-point tile through the MODE `mode` stream.  pts [32,3] float64.  -> sdf [32], grad [32,3] | None, feat | None."""
    raw = np.ascontiguousarray(np.asarray(tables, dtype=np.float32))
    tables = raw.astype(np.float64)
    # rows 0..8: one packed fp16 pair per output row, (b_hi | b_lo * 2^11 << 16); the kernel adds it through one extra MFMA
    # with the B column [1, 2^-11, 0, ...]
    bits = raw.view(np.uint32)
    bias_hi = (bits & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)
    bias_lo = (bits >> 16).astype(np.uint16).view(np.float16).astype(np.float64)

    def bias_mfma(l, c):
        out = np.zeros((16, 64))
        for r in range(16):
            row = 32 * c + frow(r, HF)
            out[r] = bias_hi[l][row] + bias_lo[l][row] / 2048.0
        return out

    st = Stream(stream16)
    x3 = pts * 3.0
    x3l = x3[J]                                    # per lane
    # embedding B operands: K step s, element i <-> entry col32(s, hf, i)
    ebh, ebl = [], []
    for s in range(3):
        v = np.zeros((64, 8))
        for i in range(8):
            for hf in range(2):
                e = col32(s, hf, i)
                m = HF == hf
                v[m, i] = enc_entry(x3l[m], e)
        h, l = split16(v)
        ebh.append(h)
        ebl.append(l)
    e4 = [st.chunk(3) for _ in range(8)]           # resident preamble
    if trace is not None:
        trace['eb'] = (ebh, ebl)
    want_d = mode >= 1
    qs = {}

    def epi_fwd(layer, c, hh, cc):
        t = hh + cc / 2048.0
        e = np.exp2(np.minimum(t, 64.0))
        p = 1.0 + e
        if want_d:
            qs[(layer, c)] = unorm16(1.0 / p)
        return np.maximum(np.log2(p), t)

    u = []
    for c in range(8):
        hh, cc = kloop(st.chunk(4), 3, ebh, ebl, np.zeros((16, 64)))
        u.append(epi_fwd(0, c, hh + bias_mfma(0, c), cc))
    if trace is not None:
        trace['u'] = [u]
    for l in range(1, 8):
        bh, bl = act_to_b(u)
        nu = []
        for c in range(8):
            main = st.chunk(16)
            hh, cc = kloop(main, 16, bh, bl, np.zeros((16, 64)))
            hh = hh + bias_mfma(l, c)
            if l == 4:       # the skip part: resident E4 * emb on top of the window's sums
                h2, c2 = kloop(e4[c], 3, ebh, ebl, np.zeros((16, 64)))
                hh, cc = hh + h2, cc + c2
            nu.append(epi_fwd(l, c, hh, cc))
        u = nu
        if trace is not None:
            trace['u'].append(u)
    bh, bl = act_to_b(u)
    feat = None
    if mode == 2:
        feat = np.zeros((32, 256))
        for c in range(8):
            hh, cc = kloop(st.chunk(16), 16, bh, bl, np.zeros((16, 64)))
            v = hh + bias_mfma(8, c) + cc / 2048.0
            for r in range(16):
                feat[J, 32 * c + frow(r, HF)] = v[r]
    hh, cc = kloop(st.chunk(16), 16, bh, bl, tab_init(tables, 9, 0))
    sdf = (hh + cc / 2048.0)[0][HF == 0]
    if not want_d:
        assert st.pos == len(st.buf)
        return sdf, None, None
    # T7
    tcur = []
    for c in range(8):
        a8 = tab_init(tables, 10, c)
        tcur.append(a8 + (a8 * (-1.0 / 65535.0)) * qs[(7, c)])
    dx = np.zeros((3, 64))

    def emb_stage(bh, bl):
        for c in range(2):
            hh, cc = kloop(st.chunk(16), 16, bh, bl, np.zeros((16, 64)))
            g = hh + cc / 2048.0
            for r in range(16):
                for hf in range(2):
                    e = 32 * c + frow(r, hf)
                    if e >= 39:
                        continue
                    m = HF == hf
                    dx[emb_dim(e)][m] += g[r][m] * enc_dentry(x3l[m], e)

    for l in range(7, 0, -1):
        bh, bl = act_to_b(tcur)
        nxt = []
        for c in range(8):
            hh, cc = kloop(st.chunk(16), 16, bh, bl, np.zeros((16, 64)))
            g = hh + cc / 2048.0
            nxt.append(g + (g * (-1.0 / 65535.0)) * qs[(l - 1, c)])
        if l == 4:           # R4e follows R4 in the stream: both read t_4
            emb_stage(bh, bl)
        tcur = nxt
    bh, bl = act_to_b(tcur)
    emb_stage(bh, bl)
    assert st.pos == len(st.buf)
    grad = np.stack([(dx[d][HF == 0] + dx[d][HF == 1]) * 3.0 for d in range(3)], axis=1)
    return sdf, grad, feat


def split16_scaled(x):
    """hi + lo / 2^11 with the residual scaled (the reflectance net's convention: its activations are small)"""
    with np.errstate(over="ignore"):
        hi = x.astype(np.float16)
        lo = ((x - hi.astype(np.float64)) * 2048.0).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def kloop_scaled(chunk, ks, bh, bl, hh):
    """K loop with BOTH residuals scaled by 2^11: hh += A_hi B_hi, cc += A_hi B_lo + A_lo B_hi"""
    cc = np.zeros((16, 64))
    hh = hh.copy()
    for s in range(ks):
        ah, al = chunk[s, 0], chunk[s, 1]
        hh = mfma_32x32x16(ah, bh[s], hh)
        cc = mfma_32x32x16(al, bh[s], cc)
        cc = mfma_32x32x16(ah, bl[s], cc)
    return hh, cc


def act_to_b_scaled(u):
    bh, bl = [], []
    for c in range(8):
        for t in range(2):
            h, l = split16_scaled(u[c][8 * t: 8 * t + 8].T)
            bh.append(h)
            bl.append(l)
    return bh, bl


def color32_tile(stream16, tables, part, pts, nrm, raymisc, scaled=True):
    """One 32-sample tile of the reflectance net through the colour block stream (csrc/nrh_color32.hip).
    part [32,256]: the feature block's share of layer 0 (W0feat * feature); pts, nrm [32,3]; raymisc [>= 99].  -> rgb [32,3]."""
    raw = np.ascontiguousarray(np.asarray(tables, dtype=np.float32))
    bits = raw.view(np.uint32)
    bias_hi = (bits & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)
    bias_lo = (bits >> 16).astype(np.uint16).view(np.float16).astype(np.float64)

    def bias(l, c):
        out = np.zeros((16, 64))
        for r in range(16):
            row = 32 * c + frow(r, HF)
            out[r] = bias_hi[l][row] + bias_lo[l][row] / 2048.0
        return out

    # scaled=False: the experiment build with the unscaled residual (NRH32_COL_UNSCALED)
    sp, kl, a2b = (split16_scaled, kloop_scaled, act_to_b_scaled) if scaled else (split16, kloop, act_to_b)
    st = Stream(stream16)
    # B operands of C0: K step 0 = point (hf 0) / normal (hf 1) in slots 0..2; K steps 1..7 = raymisc[16 (s-1) + 8 (i>>2) + 4 hf + (i&3)]
    bh, bl = [], []
    for s in range(16):
        v = np.zeros((64, 8))
        if s == 0:
            for k in range(3):
                v[:, k] = np.where(HF == 0, pts[J, k], nrm[J, k])
        elif s < 8:
            for i in range(8):
                idx = 16 * (s - 1) + 8 * (i >> 2) + 4 * HF + (i & 3)
                ok = idx < 99
                v[:, i] = np.where(ok, np.asarray(raymisc, dtype=np.float64)[np.minimum(idx, 98)], 0.0)
        h, l = sp(v)
        bh.append(h)
        bl.append(l)
    u = []
    for c in range(8):
        hh, cc = kl(st.chunk(16), 16, bh, bl, np.zeros((16, 64)))
        t = hh + bias(0, c) + cc / 2048.0
        for r in range(16):
            t[r] += part[J, 32 * c + frow(r, HF)]
        u.append(np.maximum(t, 0.0))
    for l in range(1, 4):
        bh, bl = a2b(u)
        nu = []
        for c in range(8):
            hh, cc = kl(st.chunk(16), 16, bh, bl, np.zeros((16, 64)))
            nu.append(np.maximum(hh + bias(l, c) + cc / 2048.0, 0.0))
        u = nu
    bh, bl = a2b(u)
    hh, cc = kl(st.chunk(16), 16, bh, bl, tab_init(raw.astype(np.float64), 4, 0))
    assert st.pos == len(st.buf)
    v = hh + cc / 2048.0
    rgb = np.zeros((32, 3))
    for r in range(3):
        rgb[:, r] = 1.0 / (1.0 + np.exp(-v[r][HF == 0]))
    return rgb


def sdf32_tile_jvp(stream16, tables, pts, dirs):
    """MODE 3 of csrc/nrh_sdf32.hip: 16 points (columns 0..15) and their tangents along `dirs` (columns 16..31) through the
    forward-only stream.  pts, dirs [16,3] float64.  -> sdf [16], d sdf / dt [16]."""
    raw = np.ascontiguousarray(np.asarray(tables, dtype=np.float32))
    tab = raw.astype(np.float64)
    bits = raw.view(np.uint32)
    bias_hi = (bits & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)
    bias_lo = (bits >> 16).astype(np.uint16).view(np.float16).astype(np.float64)
    is_pt = J < 16
    pj = J & 15

    def bias(l, c):
        out = np.zeros((16, 64))
        for r in range(16):
            row = 32 * c + frow(r, HF)
            out[r] = np.where(is_pt, bias_hi[l][row] + bias_lo[l][row] / 2048.0, 0.0)     # tangent columns take no bias
        return out

    st = Stream(stream16)
    x3l = (pts * 3.0)[pj]
    xdl = (dirs * 3.0)[pj]
    ebh, ebl = [], []
    for s in range(3):
        v = np.zeros((64, 8))
        for i in range(8):
            for hf in range(2):
                e = col32(s, hf, i)
                m = HF == hf
                val = enc_entry(x3l[m], e)
                if e < 39:
                    tan = enc_dentry(x3l[m], e) * xdl[m][:, emb_dim(e)]
                    val = np.where(is_pt[m], val, tan)
                v[m, i] = val
        h, l = split16(v)
        ebh.append(h)
        ebl.append(l)
    e4 = [st.chunk(3) for _ in range(8)]

    def epi(hh, cc):
        t = hh + cc / 2048.0
        p = 1.0 + np.exp2(np.minimum(t, 64.0))
        u = np.maximum(np.log2(p), t)
        q = 1.0 / p
        qs = np.empty_like(q)
        qs[:, :] = q[:, (LANES - 16) % 64]            # what a tangent lane reads: its point lane, 16 below
        return np.where(is_pt, u, t - t * qs)

    u = []
    for c in range(8):
        hh, cc = kloop(st.chunk(4), 3, ebh, ebl, np.zeros((16, 64)))
        u.append(epi(hh + bias(0, c), cc))
    for l in range(1, 8):
        bh, bl = act_to_b(u)
        nu = []
        for c in range(8):
            hh, cc = kloop(st.chunk(16), 16, bh, bl, np.zeros((16, 64)))
            hh = hh + bias(l, c)
            if l == 4:
                h2, c2 = kloop(e4[c], 3, ebh, ebl, np.zeros((16, 64)))
                hh, cc = hh + h2, cc + c2
            nu.append(epi(hh, cc))
        u = nu
    bh, bl = act_to_b(u)
    init = tab_init(tab, 9, 0)
    init = np.where(is_pt, init, 0.0)
    hh, cc = kloop(st.chunk(16), 16, bh, bl, init)
    assert st.pos == len(st.buf)
    v = (hh + cc / 2048.0)[0]
    return v[(HF == 0) & is_pt], v[(HF == 0) & ~is_pt]
