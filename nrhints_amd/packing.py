"""Host-side parameter packing for the gfx950 MLP kernels (torch ops only, differentiable).

The kernels (csrc/nrh_mlp.h) consume every GEMM stage as chunks of two 16-row output blocks; inside a chunk the
A operands of ``v_mfma_f32_16x16x4_f32`` for (output block ob, K block kb) are one 1 KiB line, lane-linear:

    packed[chunk][obi][kb][lane][c] = W[(2*chunk + obi)*16 + (lane & 15)][kb*16 + 4*(lane >> 4) + c]

so that one ``ds_read_b128`` per lane fetches the four A values of four consecutive MFMAs and the copy
global -> LDS is a straight ``global_load_lds_dwordx4`` stream.

Weight-norm is folded here (W = g * v / ||v||_row, reference fields/sdf_field.py:81-82 /
fields/reflectance_network.py:61-62), the 1/sqrt(2) of the skip connection (fields/sdf_field.py:114) is folded
into W4, and the reflectance net's first-layer columns are permuted to the order the colour kernel produces
its inputs in (csrc/nrh_color.hip).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

# geometry shared with csrc/nrh_mlp.h
SDF_L0_FLOATS = 8 * 2 * 4 * 256
SDF_REG_FLOATS = 8 * 2 * 16 * 256
SDF_R0_FLOATS = 2 * 2 * 16 * 256
SDF_PACKED_FLOATS = SDF_L0_FLOATS + 15 * SDF_REG_FLOATS + SDF_R0_FLOATS
SDF_BIAS_FLOATS = 9 * 256
SDF_HEAD_FLOATS = 257
COL_C0B_FLOATS = 8 * 2 * 8 * 256
COL_PACKED_FLOATS = SDF_REG_FLOATS + COL_C0B_FLOATS + 3 * SDF_REG_FLOATS + 2 * 16 * 256


def col_packed_floats(hints: bool = True) -> int:
    return SDF_REG_FLOATS + 8 * 2 * (8 if hints else 4) * 256 + 3 * SDF_REG_FLOATS + 2 * 16 * 256
COL_BIAS_FLOATS = 4 * 256 + 16
RAYMISC_STRIDE = 100
SDF_SCRATCH_FLOATS_PER_WAVE = 8 * 16 * 256


def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """Dense weight of an old-style ``nn.utils.weight_norm`` Linear (dim=0): W = v * (g / ||v||_row)."""
    return v * (g / v.norm(dim=1, keepdim=True))


def pack_stage(w: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """Pack a dense [out,in] matrix, zero-padded to [rows, cols] (rows % 32 == 0, cols % 16 == 0)."""
    assert rows % 32 == 0 and cols % 16 == 0 and w.shape[0] <= rows and w.shape[1] <= cols
    wp = torch.nn.functional.pad(w, (0, cols - w.shape[1], 0, rows - w.shape[0]))
    nch, kb = rows // 32, cols // 16
    # [chunk, obi, i, kb, q, c] -> [chunk, obi, kb, q, i, c];  lane = q*16 + i
    return wp.reshape(nch, 2, 16, kb, 4, 4).permute(0, 1, 3, 4, 2, 5).reshape(-1)


def pack_stages(ws, rows: int, cols: int, precision: int) -> torch.Tensor:
    """Several same-shape stages in one pass (a training step re-packs every step: ~10 small kernels per call, so the
    15 regular 256x256 SDF stages go through as one [15,256,256] batch).  Equals the concatenation of the per-stage
    packings, in order."""
    src = ws if precision == 0 else [w.detach().float() for w in ws]      # the fp32 packing stays differentiable
    w3 = torch.stack([torch.nn.functional.pad(w, (0, cols - w.shape[1], 0, rows - w.shape[0])) for w in src])
    B = w3.shape[0]
    if precision == 0:
        nch, kb = rows // 32, cols // 16
        return w3.reshape(B, nch, 2, 16, kb, 4, 4).permute(0, 1, 2, 4, 5, 3, 6).reshape(-1)
    nch, ks = rows // 32, cols // 32
    hi, lo = split_f16(w3)
    both = torch.stack([hi, lo], 1)                                 # [B, part, rows, cols]
    t = both.reshape(B, 2, nch, 2, 16, ks, 2, 4, 4).permute(0, 2, 3, 5, 1, 7, 4, 6, 8)
    return t.reshape(-1).contiguous()


LO_SCALE = 2048.0


def split_f16(x: torch.Tensor):
    """fp32 -> (hi, lo) fp16 with x ~= hi + lo / 2^11  (csrc/nrh_mlp.h, precision mode "f16x3")."""
    hi = x.to(torch.float16)
    lo = ((x - hi.to(x.dtype)) * LO_SCALE).to(torch.float16)
    return hi, lo


def pack_stage_h3(w: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """f16x3 packing of a dense [out,in] matrix (rows % 32 == 0, cols % 32 == 0) -> fp16 tensor of 2*rows*cols
    elements (same byte count as the fp32 packing):

        packed[chunk][obi][s][part][lane][e]   part 0 = hi, 1 = lo;   e < 4: W[row][32s + 4q + e]
                                                                      e >= 4: W[row][32s + 16 + 4q + (e-4)]
    with row = (2*chunk + obi)*16 + (lane & 15), q = lane >> 4 - the 8 fp16 A elements of one
    v_mfma_f32_16x16x32_f16, K slot (q, e) matching the B operand built from two D-layout blocks."""
    assert rows % 32 == 0 and cols % 32 == 0 and w.shape[0] <= rows and w.shape[1] <= cols
    wp = torch.nn.functional.pad(w.detach().float(), (0, cols - w.shape[1], 0, rows - w.shape[0]))
    nch, ks = rows // 32, cols // 32
    hi, lo = split_f16(wp)
    both = torch.stack([hi, lo], 0)                                 # [part, rows, cols]
    # [part, chunk, obi, i, s, half(2 blocks), q, e4] -> [chunk, obi, s, part, q, i, half, e4]; lane = q*16+i, e = half*4+e4
    t = both.reshape(2, nch, 2, 16, ks, 2, 4, 4).permute(1, 2, 4, 0, 6, 3, 5, 7)
    return t.reshape(-1).contiguous()


def _pad_vec(b: torch.Tensor, n: int) -> torch.Tensor:
    return b if b.shape[0] == n else torch.cat([b, b.new_zeros(n - b.shape[0])])


def dense_params(state: Dict[str, torch.Tensor], prefix: str = "") -> Dict[str, torch.Tensor]:
    """Weight-norm-folded dense matrices from a reference-layout state dict / parameter dict."""
    def lin(name):
        return (fold_weight_norm(state[f"{prefix}{name}.weight_g"], state[f"{prefix}{name}.weight_v"]),
                state[f"{prefix}{name}.bias"])
    out = {}
    for i in range(8):
        out[f"sdf_w{i}"], out[f"sdf_b{i}"] = lin(f"sdf_network.lin{i}")
    out["sdf_head_w"], out["sdf_head_b"] = lin("sdf_network.out_sdf")
    out["feat_w"], out["feat_b"] = lin("sdf_network.out_feat")
    for i in range(5):
        out[f"col_w{i}"], out[f"col_b{i}"] = lin(f"color_network.lin{i}")
    return out


_FOLD_LAYERS = [f"sdf_network.lin{i}" for i in range(8)] + ["sdf_network.out_sdf", "sdf_network.out_feat"] + \
               [f"color_network.lin{i}" for i in range(5)]
_FOLD_KEYS = [(f"sdf_w{i}", f"sdf_b{i}") for i in range(8)] + [("sdf_head_w", "sdf_head_b"), ("feat_w", "feat_b")] + \
             [(f"col_w{i}", f"col_b{i}") for i in range(5)]


class WeightNormFoldHip(torch.autograd.Function):
    """fold_weight_norm for all linears in ONE HIP launch, adjoint in one launch (csrc/nrh_fold.hip).
    forward(g_0..g_{L-1}, v_0..v_{L-1}) -> (W_0..W_{L-1}); float32 CUDA tensors."""

    @staticmethod
    def _call(fn_name, vs, gs, *ptr_lists):
        import ctypes
        from . import _lib
        lib = _lib.load()
        L = len(vs)
        IntArr, PtrArr = ctypes.c_int * L, ctypes.c_void_p * L
        rows, cols = IntArr(*[v.shape[0] for v in vs]), IntArr(*[v.shape[1] for v in vs])
        arr = lambda ts: PtrArr(*[(None if t is None else t.data_ptr()) for t in ts])
        rc = getattr(lib, fn_name)(L, rows, cols, arr(vs), arr(gs), *[arr(p) for p in ptr_lists], _lib.stream_handle())
        _lib.check(rc, fn_name)

    @staticmethod
    def forward(ctx, *gv):
        L = len(gv) // 2
        gs = [t.detach().contiguous() for t in gv[:L]]
        vs = [t.detach().contiguous() for t in gv[L:]]
        if not all(t.is_cuda and t.dtype == torch.float32 for t in gs + vs):
            raise RuntimeError("WeightNormFoldHip needs float32 CUDA parameters")
        ws = [torch.empty_like(v) for v in vs]
        WeightNormFoldHip._call("nrh_weight_norm_fold", vs, gs, ws)
        ctx.save_for_backward(*gs, *vs)
        return tuple(ws)

    @staticmethod
    def backward(ctx, *wbars):
        saved = ctx.saved_tensors
        L = len(saved) // 2
        gs, vs = list(saved[:L]), list(saved[L:])
        wb = [None if t is None else t.to(torch.float32).contiguous() for t in wbars]
        vbars = [torch.empty_like(v) for v in vs]
        gbars = [torch.empty_like(g) for g in gs]
        WeightNormFoldHip._call("nrh_weight_norm_fold_backward", vs, gs, wb, vbars, gbars)
        return (*gbars, *vbars)


def dense_params_hip(params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``dense_params`` for live float32 CUDA parameters with the 15 folds (and their backward) in one launch each."""
    gs = [params[n + ".weight_g"] for n in _FOLD_LAYERS]
    vs = [params[n + ".weight_v"] for n in _FOLD_LAYERS]
    ws = WeightNormFoldHip.apply(*gs, *vs)
    out = {}
    for (wk, bk), n, w in zip(_FOLD_KEYS, _FOLD_LAYERS, ws):
        out[wk], out[bk] = w, params[n + ".bias"]
    return out


def dense_params_device(state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The fold the PRODUCT uses for tensors where they live: nrh_weight_norm_fold for float32 GPU tensors (every path on the GPU
    folds with this one kernel, so the same parameters give the same packed bits - and the same sample placement - in an evaluation
    render, a training forward and a file written by examples/dump_scene.py), the torch expression on the CPU."""
    on_gpu = all(t.is_cuda and t.dtype == torch.float32 for k, t in state.items() if k.endswith(("weight_g", "weight_v")))
    return dense_params_hip(state) if on_gpu else dense_params(state)


def enc_columns(dim: int, n_freq: int, n_freq_compiled: int) -> torch.Tensor:
    """Where the columns of a NeRF encoding with ``n_freq`` frequencies sit in the encoding with ``n_freq_compiled`` of them
    (fields/encodings.py:165-174: [x | sin(x_d 2^k), d-major | sin(x_d 2^k + pi/2), d-major], frequencies 2^0 .. 2^(n-1): the
    shorter encoding is a subset of the longer one's columns)."""
    assert 0 <= n_freq <= n_freq_compiled
    d, k = torch.arange(dim)[:, None], torch.arange(n_freq)[None, :]
    sin = (dim + d * n_freq_compiled + k).reshape(-1)
    return torch.cat([torch.arange(dim), sin, sin + dim * n_freq_compiled])


_PLACE_IDX: Dict[tuple, tuple] = {}


def _place(w: torch.Tensor, rows: int, cols: int, col_idx: torch.Tensor = None) -> torch.Tensor:
    """``w`` [r, c] inside a zero [rows, cols] matrix: rows 0..r-1, columns ``col_idx`` (default 0..c-1).  Differentiable; the
    identity (same tensor, no launch) when nothing moves.  The index tensors are cached per device (a captured training step
    re-pads every replay: no host-to-device copy may sit inside the capture)."""
    r, c = w.shape
    if col_idx is None:
        if (r, c) == (rows, cols):
            return w
        return torch.nn.functional.pad(w, (0, cols - c, 0, rows - r))
    key = (r, tuple(col_idx.tolist()), str(w.device))
    if key not in _PLACE_IDX:
        _PLACE_IDX[key] = (torch.arange(r, device=w.device)[:, None], col_idx.to(w.device)[None, :])
    return w.new_zeros(rows, cols).index_put(_PLACE_IDX[key], w)


def pad_to_compiled(d: Dict[str, torch.Tensor], shadow_hint: bool, specular_hint: bool, mv: int) -> Dict[str, torch.Tensor]:
    """Dense matrices of a network NARROWER than the compiled one (sdf d_hidden <= 256 with multi_res <= 6 and d_out_feat <= 256,
    reflectance d_hidden <= 256 with multi_res <= 4, one hint instead of two; fields/sdf_field.py:11-36,
    fields/reflectance_network.py:9-22) as the matrices of the compiled shape - 8 x 256 / 39-column embedding / 217-row skip
    layer, 4 x 256 / 361 (316 without hints) input columns - with zeros everywhere else.  Exact, not an approximation: a padded
    hidden channel computes softplus(0) or relu(0) and every weight that reads it is zero; a padded encoding column is computed
    by the kernels and multiplied by zero; value, gradient and every adjoint of the real entries are those of the narrow network.
    Differentiable (pad / index_put), so the autograd path's parameter gradients come out in the parameters' own shapes.
    ``mv``: the reflectance net's multi_res.  Widths are read from the matrices."""
    h = d["sdf_w1"].shape[0]
    e = d["sdf_w0"].shape[1]
    m = (e - 3) // 6
    f = d["feat_w"].shape[0]
    ch = d["col_w1"].shape[0]
    assert e == 3 + 6 * m and m <= 6 and h <= 256 and f <= 256 and ch <= 256 and d["sdf_w3"].shape[0] == h - e <= 217 and mv <= 4
    out = dict(d)
    emap = enc_columns(3, m, 6)
    out["sdf_w0"] = _place(d["sdf_w0"], 256, 39, emap)
    for l in (1, 2, 5, 6, 7):
        out[f"sdf_w{l}"] = _place(d[f"sdf_w{l}"], 256, 256)
    out["sdf_w3"] = _place(d["sdf_w3"], 217, 256)
    # layer 4 reads cat([h3 (h - e), embedding (e)]) (fields/sdf_field.py:113-114): the compiled layer's columns 0..216 | 217..255
    out["sdf_w4"] = _place(d["sdf_w4"], 256, 256, torch.cat([torch.arange(h - e), 217 + emap]))
    out["sdf_head_w"] = _place(d["sdf_head_w"], 1, 256)
    out["feat_w"] = _place(d["feat_w"], 256, 256)
    for l in range(8):
        out[f"sdf_b{l}"] = _pad_vec(d[f"sdf_b{l}"], 217 if l == 3 else 256)
    out["feat_b"] = _pad_vec(d["feat_b"], 256)
    # reflectance input (fields/reflectance_network.py:77-86): pts 3 | enc(view) | normal 3 | enc(pl) | feature f | enc(vis) | enc(cue)
    ev, hints = 3 + 6 * mv, shadow_hint or specular_hint
    cols = [torch.arange(3), 3 + enc_columns(3, mv, 4), 30 + torch.arange(3), 33 + enc_columns(3, mv, 4), 60 + torch.arange(f)]
    if shadow_hint:
        cols.append(316 + enc_columns(1, mv, 4))
    if specular_hint:
        cols.append(325 + enc_columns(4, mv, 4))
    cmap = torch.cat(cols)
    assert d["col_w0"].shape[1] == cmap.numel() == 6 + 2 * ev + f + (1 + 2 * mv) * (int(shadow_hint) + 4 * int(specular_hint))
    width = 361 if hints else 316
    identity = cmap.numel() == width and bool((cmap == torch.arange(width)).all())
    out["col_w0"] = _place(d["col_w0"], 256, width, None if identity else cmap)
    for l in (1, 2, 3):
        out[f"col_w{l}"] = _place(d[f"col_w{l}"], 256, 256)
    out["col_w4"] = _place(d["col_w4"], 3, 256)
    for l in range(4):
        out[f"col_b{l}"] = _pad_vec(d[f"col_b{l}"], 256)
    return out


def check_default_shapes(d: Dict[str, torch.Tensor], hints: bool = True) -> None:
    """The kernels are compiled for the default nr-hints network shape."""
    want = {"sdf_w0": (256, 39), "sdf_w1": (256, 256), "sdf_w2": (256, 256), "sdf_w3": (217, 256),
            "sdf_w4": (256, 256), "sdf_w5": (256, 256), "sdf_w6": (256, 256), "sdf_w7": (256, 256),
            "sdf_head_w": (1, 256), "feat_w": (256, 256), "col_w0": (256, 361 if hints else 316), "col_w1": (256, 256),
            "col_w2": (256, 256), "col_w3": (256, 256), "col_w4": (3, 256)}
    for k, shp in want.items():
        if tuple(d[k].shape) != shp:
            raise ValueError(f"unsupported network shape: {k} is {tuple(d[k].shape)}, the HIP kernels are built for {shp} "
                             "(default nr-hints config: d_hidden=256, 8/4 layers, skip_in=[4], multires 6/4, "
                             "shadow + 4-roughness specular hints)")


def pack_sdf(d: Dict[str, torch.Tensor], precision: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (packed stages, biases [9*256], head [257]) in kernel execution order
    L0 | L1..L7 | FEAT | R7..R1 | R0  (R_l = W_l^T for the analytic reverse chain).
    precision 0: float32 [SDF_PACKED_FLOATS]; precision 1 (f16x3): float16 [2 * SDF_PACKED_FLOATS]."""
    ps = pack_stage if precision == 0 else pack_stage_h3
    w = [d[f"sdf_w{i}"] for i in range(8)]
    # cat([h, embed]) / sqrt(2) folded into the weights.  A TRUE float32 division by float32(sqrt 2), element by element, as
    # PackPlan and nrh_pack_gather do it: ``tensor / python_float`` is a multiplication by the reciprocal on the GPU (and a
    # division on the CPU), which made this direct packer and the plan differ in the last bit of 42 % of W4's entries - enough to
    # move importance samples where the pdf sits at its floor, so an evaluation render (this packer) and a training forward (the
    # plan) of the same parameters placed their samples differently (round 6, profiles/pack_route_probe.py).
    w4 = w[4] / torch.full_like(w[4], math.sqrt(2.0))
    fwd = [w[0], w[1], w[2], w[3], w4, w[5], w[6], w[7]]
    regular = [fwd[l] for l in range(1, 8)] + [d["feat_w"]] + [fwd[l].t() for l in range(7, 0, -1)]
    packed = torch.cat([ps(fwd[0], 256, 64), pack_stages(regular, 256, 256, precision), ps(fwd[0].t(), 64, 256)])
    assert packed.numel() == SDF_PACKED_FLOATS * (1 if precision == 0 else 2)
    bias = torch.cat([_pad_vec(d[f"sdf_b{i}"], 256) for i in range(8)] + [d["feat_b"]])
    head = torch.cat([d["sdf_head_w"].reshape(-1), d["sdf_head_b"].reshape(-1)])
    return packed.contiguous(), bias.contiguous(), head.contiguous()


def pack_feat_transposed(d: Dict[str, torch.Tensor], precision: int = 0) -> torch.Tensor:
    """Wf^T as one regular 256x256 stage: first stage of the training value-adjoint sweep (hbar_7 = Wf^T fbar + ...)."""
    ps = pack_stage if precision == 0 else pack_stage_h3
    return ps(d["feat_w"].t(), 256, 256).contiguous()


def color_input_permutation(hints: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """Column indices of the reference's 361-wide reflectance input
    [pts 0:3 | enc4(view) 3:30 | normal 30:33 | enc4(pl) 33:60 | feat 60:316 | enc4(vis) 316:325 | enc4(cue) 325:361]
    (fields/reflectance_network.py:77-82) in kernel order: (feature part [256], other part [105] =
    [pts, normal, enc4(view), enc4(pl), enc4(vis), enc4(cue)])."""
    feat = torch.arange(60, 316)
    parts = [torch.arange(0, 3), torch.arange(30, 33), torch.arange(3, 30), torch.arange(33, 60)]
    if hints:  # without hints (pl-naive) the input simply ends after feat (fields/reflectance_network.py:77-82)
        parts += [torch.arange(316, 325), torch.arange(325, 361)]
    return feat, torch.cat(parts)


def pack_color(d: Dict[str, torch.Tensor], precision: int = 0, hints: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> (packed stages, biases [4*256+16]) in order C0a | C0b | C1 | C2 | C3 | C4 (fp32 or fp16 pairs, see pack_sdf)."""
    ps = pack_stage if precision == 0 else pack_stage_h3
    fi, mi = color_input_permutation(hints)
    w0 = d["col_w0"]
    parts = [ps(w0[:, fi.to(w0.device)], 256, 256), ps(w0[:, mi.to(w0.device)], 256, 128 if hints else 64),
             pack_stages([d[f"col_w{l}"] for l in (1, 2, 3)], 256, 256, precision), ps(d["col_w4"], 32, 256)]
    packed = torch.cat(parts)
    assert packed.numel() == col_packed_floats(hints) * (1 if precision == 0 else 2)
    bias = torch.cat([d["col_b0"], d["col_b1"], d["col_b2"], d["col_b3"], _pad_vec(d["col_b4"], 16)])
    return packed.contiguous(), bias.contiguous()


def pack_color_transposed(d: Dict[str, torch.Tensor], precision: int = 0, hints: bool = True) -> torch.Tensor:
    """Stages of the reflectance net's adjoint sweep (csrc/nrh_color.hip color_adjoint_kernel), in execution order
    W4^T (256 x 32, 3 columns used) | W3^T | W2^T | W1^T | W0[:, feat]^T | W0[:, other]^T (128 | 64 rows)."""
    ps = pack_stage if precision == 0 else pack_stage_h3
    fi, mi = color_input_permutation(hints)
    w0 = d["col_w0"]
    regular = [d["col_w3"].t(), d["col_w2"].t(), d["col_w1"].t(), w0[:, fi.to(w0.device)].t()]
    parts = [ps(d["col_w4"].t(), 256, 32), pack_stages(regular, 256, 256, precision),
             ps(w0[:, mi.to(w0.device)].t(), 128 if hints else 64, 256)]
    return torch.cat(parts).contiguous()


# ---- one-gather packing plan --------------------------------------------------------------------------------------------
# A training step re-packs every buffer after each optimiser step.  All packers above are pure index permutations (plus the
# 1/sqrt(2) of W4 and the fp16 split), so they are run ONCE on tensors of element indices; after that a re-pack is one
# gather + one divide + the split, for all buffers together (~10 launches instead of ~70).
_W_KEYS = [w for w, _ in _FOLD_KEYS]
_B_KEYS = [b for _, b in _FOLD_KEYS]


def _h3_layout(wp: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    """Element order of ONE part (hi or lo) of pack_stage_h3: [chunk][obi][s][lane = q*16+i][e = half*4+e4]."""
    nch, ks = rows // 32, cols // 32
    return wp.reshape(nch, 2, 16, ks, 2, 4, 4).permute(0, 1, 3, 5, 2, 4, 6).reshape(-1)


_ZERO1: Dict[str, torch.Tensor] = {}


def zero1(dev) -> torch.Tensor:
    """The constant element 0 that heads a pack plan's flat source (index 0 = "no source"): one tensor per device instead of a fill
    launch per re-pack (a training step re-packs twice)."""
    key = str(dev)
    if key not in _ZERO1:
        _ZERO1[key] = torch.zeros(1, dtype=torch.float32, device=dev)
    return _ZERO1[key]


class PackPlan:
    """Index form of pack_sdf / pack_color / pack_feat_transposed / pack_color_transposed for one (precision, hints)."""

    def __init__(self, dense: Dict[str, torch.Tensor], precision: int, hints: bool):
        dev = dense["sdf_w0"].device
        self.precision, self.hints = precision, hints
        self.shapes = {k: tuple(dense[k].shape) for k in _W_KEYS + _B_KEYS}
        # element indices (+1, so that the zero padding of the packers lands on slot 0 = the constant 0.0)
        idx, off = {}, 1
        for k in _W_KEYS + _B_KEYS:
            n = int(torch.tensor(self.shapes[k]).prod())
            idx[k] = (torch.arange(n, dtype=torch.float64) + off).reshape(self.shapes[k])
            off += n
        self.total = off
        div = {k: torch.ones(self.shapes[k], dtype=torch.float64) for k in _W_KEYS + _B_KEYS}
        div["sdf_w4"] = torch.full(self.shapes["sdf_w4"], math.sqrt(2.0), dtype=torch.float64)

        def layouts(dd):
            """The four weight buffers (fp32 layout, or the per-part fp16 layout) and the small fp32 vectors."""
            if precision == 0:
                ps = pack_stage
                many = lambda ws, r, c: torch.cat([pack_stage(w, r, c) for w in ws])
            else:
                ps = lambda w, r, c: _h3_layout(torch.nn.functional.pad(w, (0, c - w.shape[1], 0, r - w.shape[0])), r, c)
                many = lambda ws, r, c: torch.cat([ps(w, r, c) for w in ws])
            w = [dd[f"sdf_w{i}"] for i in range(8)]
            regular = [w[l] for l in range(1, 8)] + [dd["feat_w"]] + [w[l].t() for l in range(7, 0, -1)]
            sdf = torch.cat([ps(w[0], 256, 64), many(regular, 256, 256), ps(w[0].t(), 64, 256)])
            fi, mi = color_input_permutation(hints)
            w0 = dd["col_w0"]
            col = torch.cat([ps(w0[:, fi], 256, 256), ps(w0[:, mi], 256, 128 if hints else 64),
                             many([dd[f"col_w{l}"] for l in (1, 2, 3)], 256, 256), ps(dd["col_w4"], 32, 256)])
            wtf = ps(dd["feat_w"].t(), 256, 256)
            colt = torch.cat([ps(dd["col_w4"].t(), 256, 32),
                              many([dd["col_w3"].t(), dd["col_w2"].t(), dd["col_w1"].t(), w0[:, fi].t()], 256, 256),
                              ps(w0[:, mi].t(), 128 if hints else 64, 256)])
            sdf_b = torch.cat([_pad_vec(dd[f"sdf_b{i}"], 256) for i in range(8)] + [dd["feat_b"]])
            col_b = torch.cat([dd["col_b0"], dd["col_b1"], dd["col_b2"], dd["col_b3"], _pad_vec(dd["col_b4"], 16)])
            head = torch.cat([dd["sdf_head_w"].reshape(-1), dd["sdf_head_b"].reshape(-1)])
            return [sdf, col, wtf, colt], [sdf_b, col_b, head]

        wi, vi = layouts(idx)
        wd, _ = layouts(div)
        self.w_sizes = [t.numel() for t in wi]
        self.v_sizes = [t.numel() for t in vi]
        self.w_index = torch.cat(wi).to(torch.int64).to(dev)
        wdiv = torch.cat(wd)
        wdiv[wdiv == 0] = 1.0                                  # padding slots
        self.w_div = wdiv.to(torch.float32).to(dev)
        self.v_index = torch.cat(vi).to(torch.int64).to(dev)

    def matches(self, dense, precision, hints) -> bool:
        return precision == self.precision and hints == self.hints and \
            all(tuple(dense[k].shape) == self.shapes[k] for k in self.shapes) and dense["sdf_w0"].device == self.w_index.device

    def pack(self, dense: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """-> dict(sdf_w, col_w, sdf_wt_feat, col_wt, sdf_b, col_b, sdf_head), identical (bit for bit) to the direct packers.
        On the GPU: one concatenation + two launches of nrh_pack_gather (csrc/nrh_fold.hip); the torch expressions below are
        the host-side form of the same plan (CPU tests, and the definition the kernel is tested against)."""
        dev = self.w_index.device
        flat = torch.cat([zero1(dev)] + [dense[k].detach().to(torch.float32).reshape(-1) for k in _W_KEYS + _B_KEYS])
        if dev.type == "cuda":
            from . import _lib
            lib = _lib.load()
            if getattr(self, "_w_index32", None) is None:
                self._w_index32, self._v_index32 = self.w_index.to(torch.int32), self.v_index.to(torch.int32)
            nw, nv = self.w_index.numel(), self.v_index.numel()
            P = _lib.ptr
            with torch.cuda.device(dev):
                if self.precision == 0:
                    packed, scale = flat[self.w_index] / self.w_div, 1      # exact-fp32 mode: plain float32 stages
                else:
                    packed = torch.empty(2 * nw, dtype=torch.float16, device=dev)
                    _lib.check(lib.nrh_pack_gather(P(flat), P(self._w_index32, torch.int32), P(self.w_div), nw, 1, P(packed, torch.float16),
                                                   _lib.stream_handle()), "nrh_pack_gather")
                    scale = 2
                vflat = torch.empty(nv, dtype=torch.float32, device=dev)
                _lib.check(lib.nrh_pack_gather(P(flat), P(self._v_index32, torch.int32), None, nv, 0, P(vflat), _lib.stream_handle()),
                           "nrh_pack_gather")
            ws = torch.split(packed, [n * scale for n in self.w_sizes])
            vs = torch.split(vflat, self.v_sizes)
            return dict(sdf_w=ws[0], col_w=ws[1], sdf_wt_feat=ws[2], col_wt=ws[3], sdf_b=vs[0], col_b=vs[1], sdf_head=vs[2])
        x = flat[self.w_index] / self.w_div
        if self.precision == 0:
            packed, scale = x, 1
        else:
            hi, lo = split_f16(x)
            packed = torch.stack([hi.reshape(-1, 512), lo.reshape(-1, 512)], dim=1).reshape(-1)
            scale = 2
        ws = torch.split(packed, [n * scale for n in self.w_sizes])
        vs = torch.split(flat[self.v_index], self.v_sizes)
        return dict(sdf_w=ws[0], col_w=ws[1], sdf_wt_feat=ws[2], col_wt=ws[3], sdf_b=vs[0], col_b=vs[1], sdf_head=vs[2])


def feat_tiles_to_rows(tiles: torch.Tensor, npts: int) -> torch.Tensor:
    """D-layout feature tiles [ntiles,16(block),64(lane),4(r)] -> [npts,256]
    (lane = q*16 + j; feature = 16*block + 4*q + r; point = 16*tile + j)."""
    nt = tiles.numel() // 4096
    t = tiles.reshape(nt, 16, 4, 16, 4)            # tile, block, q, j, r
    return t.permute(0, 3, 1, 2, 4).reshape(nt * 16, 256)[:npts]


def rows_to_feat_tiles(rows: torch.Tensor) -> torch.Tensor:
    """[npts,256] -> D-layout tiles (npts padded up to a multiple of 16 with zeros)."""
    npts = rows.shape[0]
    nt = (npts + 15) // 16
    if nt * 16 != npts:
        rows = torch.cat([rows, rows.new_zeros(nt * 16 - npts, 256)], 0)
    return rows.reshape(nt, 16, 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)
