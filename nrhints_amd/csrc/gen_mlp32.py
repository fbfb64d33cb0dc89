#!/usr/bin/env python3
"""Instruction-schedule generator for the wide MLP machinery (csrc/nrh_mlp32.h, gfx950).

hipcc cannot be talked into the schedule this kernel needs (one wave per SIMD: every VALU instruction of the epilogue has to
sit in the 32-cycle shadow of an MFMA of the NEXT chunk, B operands have to stay in AGPRs across a loop), so the hot
"windows" are emitted here as straight-line C++ in which

  * every MFMA, every LDS read of a weight fragment and every AGPR write is an ``asm volatile`` with literal register
    numbers (volatile asm statements keep their order), the LDS counter waits are computed by this script;
  * the epilogue arithmetic stays ordinary C++ (hipcc allocates VGPRs and handles VALU hazards), cut into micro-operations
    that a small list scheduler distributes over the MFMA slots; every value is pinned by an empty ``asm volatile`` at the
    end of its slot and ``sched_barrier(0)`` closes the slot, so neither instruction selection nor the machine scheduler
    moves work across slots.

Round 6 (see Window.emit, DMA_SLOTS16, split_ops): inside a stage the block barrier of window n + 1 sits behind K step 14 of window n
and K step 15 already requests the next window's first weight fragments (no barrier, no cold LDS reads at a window's top); a window's
eight LDS-DMA pieces go out behind the first MFMA of K steps 6..13; the hi / lo residual is one v_fma_mixlo_f16 / v_fma_mixhi_f16
pair.  check_gen32.py replays every generated file against the LDS counter model; the Makefile runs it.

AGPR map (per wave): a[0:63] in.hi, a[64:127] in.lo, a[128:191] out.hi, a[192:255] out.lo; K step s reads a[4s:4s+3] and
a[64+4s:64+4s+3].  hipcc never touches AGPRs in these kernels (-mllvm -amdgpu-mfma-vgpr-form, no spills; the build checks the
disassembly for foreign v_accvgpr instructions).

Usage: gen_mlp32.py <outdir>   ->  <outdir>/*.inc, included by nrh_sdf32.hip.

Environment knobs (build-time; the Makefile sets none but NRH32_ONE_TERM for gen32_1t; experiment libraries:
profiles/tools/build_gen_variant.sh NAME "KNOB=.. KNOB=.."; what each was measured to do: profiles/README.md):
  NRH32_ONE_TERM=1      the single-pass schedules of precision "f16" (no cross-term MFMAs, no low halves)
  NRH32_XWIN=0          no cross-window prefetch: every window opens with its block barrier and cold fragment reads (rounds 2-5)
  NRH32_MIX_SPLIT=0     the hi / lo residual spelled out in C++ instead of v_fma_mixlo_f16 / v_fma_mixhi_f16
  NRH32_DMA_J / _FIRST / _STRIDE   where a 16-step window issues its eight LDS-DMA pieces: slot of the K step (default 0), first K step (6), stride (1)
  NRH32_DMA_PENALTY     VALU micro-operations fewer in a slot that issues a piece (2)
  NRH32_NV / _NV_D1 / _NV_REV      micro-operations per MFMA slot: forward (4), forward with sigma' encode (NV + 1), reverse (NV)
  NRH32_PF              weight-fragment reads in flight ahead of the MFMAs, in K steps (2; 3 measured null)
  NRH32_MID             which of a K step's three MFMAs takes A_lo (1: the two hh updates stay apart)
  NRH32_TRANS_COST      price of a transcendental in the slot scheduler, in plain operations (1)
  NRH32_HEAD_SLOTS / _HEAD_OPS     epilogue operations placed ahead of a window's first MFMA (0: measured negative)
  NRH32_SYNCK=1         window-opening waits that keep the previous window's stores in flight (measured null)
  NRH32_COL_UNSCALED=1  the reflectance net's activations with the unscaled residual (needs -DNRH32_COL_UNSCALED=1 as well)
  NRH32_NOEPI / _NOQSTORE / _NOHINIT / _ABL_TRANS / _ABL_APUT     timing ablations for profiles/ubench: WRONG RESULTS by construction
"""
import os
import sys

LU = "nrh32::LO_UNSCALE"
MFMA = "v_mfma_f32_32x32x16_f16"


TRANS_COST = float(os.environ.get("NRH32_TRANS_COST", "1"))
PF = int(os.environ.get("NRH32_PF", "2"))        # K steps of weight fragments requested ahead of the MFMAs that consume them (Window.emit)
# hi / lo split of the SDF kernels' activations through v_fma_mixlo_f16 / v_fma_mixhi_f16 (see split_ops); NRH32_MIX_SPLIT=0: the
# spelled-out C++ form of rounds 2-5 (A/B builds)
MIX_SPLIT = os.environ.get("NRH32_MIX_SPLIT", "1") != "0"
# NRH32_ONE_TERM=1: the SINGLE-PASS variant of every schedule (precision "f16": one v_mfma per K step - A_hi * B_hi, weights and
# activations at fp16's 11 bits, fp32 accumulation): the two cross-term MFMAs of a K step are left out (their slots keep the
# epilogue work and the DMA pieces), the low fragments are not read from LDS, the accumulator of the scaled cross terms (cc) does not
# exist for the epilogues (t = hh), and the activations' low halves are neither computed nor written.  Generated into another
# directory (Makefile: gen32_1t) and compiled as its own translation unit (nrh_wide1.hip, namespace nrh32t).
ONE_TERM = bool(os.environ.get("NRH32_ONE_TERM"))


def joined(cp, hp, r):
    """the value of accumulator entry r: hh + cc 2^-11, or hh alone in the one-term variant"""
    return f"{hp}[{r}]" if ONE_TERM else f"__builtin_fmaf({cp}[{r}], {LU}, {hp}[{r}])"


def pin(*names):
    """`asm volatile("" : "+v"(a), "+v"(b))` over the accumulator pair - the cross-term accumulator is not pinned in the one-term variant"""
    names = [n for n in names if not (ONE_TERM and (n == "cp" or n.startswith("cc") or n.startswith("pc")))]
    return 'asm volatile("" : ' + ", ".join(f'"+v"({n})' for n in names) + ");"



class Op:
    """One epilogue micro-operation: C++ statement(s), the values it defines (pinned at slot end) and its inputs."""

    def __init__(self, code, defs=(), uses=(), cost=1, kind="valu"):
        self.code, self.defs, self.uses, self.cost, self.kind = code, tuple(defs), tuple(uses), cost, kind
        self.slot = None


# timing ablations for profiles/ubench (WRONG RESULTS): the epilogues without their transcendentals / without their AGPR writes
ABL_TRANS = bool(os.environ.get("NRH32_ABL_TRANS"))
ABL_APUT = bool(os.environ.get("NRH32_ABL_APUT"))


def aput(idx, val):
    if ABL_APUT:
        return Op(f'asm volatile("" ::"v"({val}));', uses=(val,), kind="aput")
    return Op(f'asm volatile("v_accvgpr_write_b32 a{idx}, %0" ::"v"({val}) : "a{idx}");', uses=(val,), kind="aput")


def split_ops(i, v0, v1, out_hi, out_lo, pfx="", scaled=False):
    """hi/lo split of the pair (v0, v1) -> AGPRs out_hi / out_lo (nrh32::split2 / split2_scaled, spelled out per instruction).
    scaled: the residual is multiplied by 2^11 before it is rounded to fp16 (the reflectance net's activations are O(0.01..1):
    their unscaled residuals would all be fp16 subnormals - which the MFMA and v_cvt_pkrtz do honour (profiles/r02/denorm32.log) -
    two more VALU per pair keep them in the normal range and their full 11 bits)."""
    n = f"{pfx}{i}"
    if scaled:
        mid = [
            Op(f"float H{n}a = {v0} * nrh32::LO_SCALE;", defs=(f"H{n}a",), uses=(v0,)),
            Op(f"float H{n}b = {v1} * nrh32::LO_SCALE;", defs=(f"H{n}b",), uses=(v1,)),
            Op(f"float R{n}a = __builtin_fmaf((float)hi{n}.x, -nrh32::LO_SCALE, H{n}a);", defs=(f"R{n}a",), uses=(f"hi{n}", f"H{n}a")),
            Op(f"float R{n}b = __builtin_fmaf((float)hi{n}.y, -nrh32::LO_SCALE, H{n}b);", defs=(f"R{n}b",), uses=(f"hi{n}", f"H{n}b")),
        ]
    else:
        # the residual goes out UNSCALED (v_fma_mix_f32 on the packed fp16): an SDF-net activation below 2^-3 loses nothing that
        # matters in absolute terms (|error| <= 2^-25 if subnormals survive, <= 2^-12 |x| if they do not), see nrh_mlp32.h
        mid = [
            Op(f"float R{n}a = __builtin_fmaf((float)hi{n}.x, -1.0f, {v0});", defs=(f"R{n}a",), uses=(f"hi{n}", v0)),
            Op(f"float R{n}b = __builtin_fmaf((float)hi{n}.y, -1.0f, {v1});", defs=(f"R{n}b",), uses=(f"hi{n}", v1)),
        ]
    if ONE_TERM:
        # (round to nearest: the high half is all there is - nrh_mlp32.h cvt_rn2)
        return [Op(f"nrh32::h16x2 hi{n} = nrh32::cvt_rn2({v0}, {v1});", defs=(f"hi{n}",), uses=(v0, v1)), aput(out_hi, f"hi{n}")]
    if MIX_SPLIT and not scaled:
        # The residual pair straight into one packed register: v_fma_mixlo_f16 / v_fma_mixhi_f16 evaluate fma(hi (fp16, read in
        # place), -1, v) in float32 - exact - and round the result to fp16 (nearest even) into the low / high half of the
        # destination: 2 instructions per pair where hipcc made 5 of the spelled-out form (v_cvt_f32_f16 x 2, v_sub_f32 x 2 -
        # an fma by -1 is folded into a subtraction, so v_fma_mix_f32 was never selected - and v_cvt_pkrtz).  A partial-register
        # write must not be read by the NEXT VALU instruction (gfx940+ dst_sel forwarding hazard; hipcc cannot see into the
        # statement): the high half's AGPR write sits between the two, and the low half's consumer is another micro-operation,
        # scheduled at least one slot (one MFMA) later; check_wide_isa.py verifies both distances in the ISA.
        return [
            Op(f"nrh32::h16x2 hi{n} = __builtin_amdgcn_cvt_pkrtz({v0}, {v1});", defs=(f"hi{n}",), uses=(v0, v1)),
            Op(f'nrh32::h16x2 lo{n}; asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\\n\\t{"s_nop 0" if ABL_APUT else f"v_accvgpr_write_b32 a{out_hi}, %1"}\\n\\t'
               f'v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(lo{n}) : "v"(hi{n}), "v"({v0}), "v"({v1}) : "a{out_hi}");',
               defs=(f"lo{n}",), uses=(f"hi{n}", v0, v1), cost=3),
            aput(out_lo, f"lo{n}"),
        ]
    ops = [Op(f"nrh32::h16x2 hi{n} = __builtin_amdgcn_cvt_pkrtz({v0}, {v1});", defs=(f"hi{n}",), uses=(v0, v1))] + mid + [
        Op(f"nrh32::h16x2 lo{n} = __builtin_amdgcn_cvt_pkrtz(R{n}a, R{n}b);", defs=(f"lo{n}",), uses=(f"R{n}a", f"R{n}b")),
        aput(out_hi, f"hi{n}"),
        aput(out_lo, f"lo{n}"),
    ]
    return ops


def epi_fwd(c, hp, cp, want_d, out_base=128, qstore="W32_QSTORE", jvp=False):
    """Forward epilogue of chunk c: t (bias is in the accumulator) -> u = log2(1 + 2^t) -> hi/lo -> out; q = 1/(1+2^t).
    jvp: the tile holds 16 points (columns 0..15) and their 16 tangents (columns 16..31, the directional derivative along the
    ray): a tangent lane takes q from its point lane (16 lanes below: W32_SWAP = v_permlane16_swap) and writes (1 - q) t
    instead of the softplus; W32_ISPT is the per-lane predicate "column < 16"."""
    ops = []
    for i in range(8):
        for r in (2 * i, 2 * i + 1):
            ops += [
                Op(f"float t{r} = {joined(cp, hp, r)};", defs=(f"t{r}",)),
                Op(f"float m{r} = __builtin_amdgcn_fmed3f(t{r}, 64.0f, -3.0e38f);", defs=(f"m{r}",), uses=(f"t{r}",)),
                Op(f"float e{r} = {'0.5f * ' if ABL_TRANS else '__builtin_amdgcn_exp2f'}(m{r});", defs=(f"e{r}",), uses=(f"m{r}",), kind="trans"),
                Op(f"float p{r} = 1.0f + e{r};", defs=(f"p{r}",), uses=(f"e{r}",)),
                Op(f"float g{r} = {'0.5f * ' if ABL_TRANS else '__builtin_amdgcn_logf'}(p{r});", defs=(f"g{r}",), uses=(f"p{r}",), kind="trans"),
                Op(f"float {'s' if jvp else 'u'}{r} = __builtin_amdgcn_fmed3f(g{r}, t{r}, 3.0e38f);", defs=(f"{'s' if jvp else 'u'}{r}",), uses=(f"g{r}", f"t{r}")),
            ]
            if want_d or jvp:
                ops.append(Op(f"float q{r} = {'0.25f * ' if ABL_TRANS else '__builtin_amdgcn_rcpf'}(p{r});", defs=(f"q{r}",), uses=(f"p{r}",), kind="trans"))
            if jvp:
                ops += [
                    Op(f"float x{r} = W32_SWAP(q{r});", defs=(f"x{r}",), uses=(f"q{r}",)),
                    Op(f"float w{r} = __builtin_fmaf(-t{r}, x{r}, t{r});", defs=(f"w{r}",), uses=(f"t{r}", f"x{r}")),
                    Op(f"float u{r} = W32_ISPT ? s{r} : w{r};", defs=(f"u{r}",), uses=(f"s{r}", f"w{r}")),
                ]
        ops += split_ops(i, f"u{2 * i}", f"u{2 * i + 1}", out_base + 8 * c + i, out_base + 64 + 8 * c + i)
        if want_d:
            ops.append(Op(f"uint32_t qq{i} = nrh32::unorm16x2(q{2 * i}, q{2 * i + 1});", defs=(f"qq{i}",),
                          uses=(f"q{2 * i}", f"q{2 * i + 1}")))
            if i % 4 == 3 and not os.environ.get("NRH32_NOQSTORE"):      # (timing ablation: WRONG RESULTS)
                w = [f"qq{i - 3 + k}" for k in range(4)]
                ops.append(Op(f"{qstore}({c}, {i // 4}, (nrh32::u32x4{{{', '.join(w)}}}));", uses=w, kind="vmem"))
    return ops


def epi_rev(c, hp, cp, out_base=128):
    """Reverse epilogue of chunk c: g = W^T t (accumulator) -> t' = g - g q, q = unorm16 from scratch -> hi/lo -> out."""
    ops = []
    for i in range(8):
        for r in (2 * i, 2 * i + 1):
            w = f"qw{r // 8}[{(r % 8) // 2}]"
            ext = f"({w} >> 16)" if r & 1 else f"({w} & 0xffffu)"
            ops += [
                Op(f"float g{r} = {joined(cp, hp, r)};", defs=(f"g{r}",)),
                Op(f"float f{r} = (float){ext};", defs=(f"f{r}",)),          # one instruction: v_cvt_f32_u32_sdwa src0_sel:WORD_n
                Op(f"float n{r} = g{r} * (-1.0f / 65535.0f);", defs=(f"n{r}",), uses=(f"g{r}",)),
                Op(f"float u{r} = __builtin_fmaf(n{r}, f{r}, g{r});", defs=(f"u{r}",), uses=(f"n{r}", f"f{r}", f"g{r}")),
            ]
        ops += split_ops(i, f"u{2 * i}", f"u{2 * i + 1}", out_base + 8 * c + i, out_base + 64 + 8 * c + i)
    return ops


def epi_relu(c, hp, cp, out_base=128, part=False):
    """ReLU epilogue of chunk c (the reflectance net's hidden layers): t = hh + cc 2^-11 [+ the feature block's share pw0..pw3,
    16 floats loaded a window earlier] -> max(t, 0) -> hi/lo -> out."""
    ops = []
    for i in range(8):
        for r in (2 * i, 2 * i + 1):
            ops.append(Op(f"float t{r} = {joined(cp, hp, r)};", defs=(f"t{r}",)))
            src = f"t{r}"
            if part:
                ops.append(Op(f"float s{r} = t{r} + pw{r // 4}[{r % 4}];", defs=(f"s{r}",), uses=(f"t{r}",)))
                src = f"s{r}"
            ops.append(Op(f"float u{r} = __builtin_amdgcn_fmed3f({src}, 0.0f, 3.0e38f);", defs=(f"u{r}",), uses=(src,)))
        ops += split_ops(i, f"u{2 * i}", f"u{2 * i + 1}", out_base + 8 * c + i, out_base + 64 + 8 * c + i, scaled=not COL_UNSCALED)
    return ops


def epi_feat(c, hp, cp, store="W32_FSTORE"):
    """Feature-head epilogue of chunk c: v = hh + cc 2^-11, stored as four float4 (rows 32 c + 8 g + 4 hf + 0..3 of this lane's
    point): W32_FSTORE(c, g, value).  No AGPR output - nothing reads the feature as a B operand."""
    ops = []
    for g in range(4):
        for k in range(4):
            r = 4 * g + k
            ops.append(Op(f"float v{r} = {joined(cp, hp, r)};", defs=(f"v{r}",)))
        w = [f"v{4 * g + k}" for k in range(4)]
        ops.append(Op(f"{store}({c}, {g}, (f32x4{{{', '.join(w)}}}));", uses=w, kind="vmem"))
    return ops


def schedule(ops, nslots, per_slot, trans_cost=1.0):
    """Greedy list schedule: an op may run in slot k if everything it uses was defined in a slot < k (or outside).
    Returns (slots, tail): ops per slot, and what did not fit."""
    defined_in = {}
    for o in ops:
        for d in o.defs:
            defined_in[d] = o
    slots = [[] for _ in range(nslots)]
    todo = list(ops)
    for k in range(nslots):
        budget = per_slot(k) if callable(per_slot) else per_slot
        rest = []
        for o in todo:
            ready = all((u not in defined_in) or (defined_in[u].slot is not None and defined_in[u].slot < k) for u in o.uses)
            cost = trans_cost if o.kind == "trans" else o.cost     # (experiment knob NRH32_TRANS_COST: a transcendental holds the
            if ready and budget >= cost:                            # VALU issue port longer than a plain op)
                o.slot = k
                slots[k].append(o)
                budget -= cost
            else:
                rest.append(o)
        todo = rest
    return slots, todo


# Between two groups of micro-operations that no MFMA separates (slots behind a K loop, the finish blocks): one wait state, so that
# the consumer of a half-register write (split_ops: v_fma_mixhi_f16) is never the very next VALU instruction
PARTIAL_WRITE_GAP = 'asm volatile("s_nop 0");'


def emit_tail(out, ops, ind="  "):
    """Left-over operations in list order: dependent ones can be neighbours here, so every consumer of a half-register write gets
    its wait state explicitly."""
    for o in ops:
        if o.kind == "aput":
            out.append(ind + PARTIAL_WRITE_GAP)
        emit_ops(out, [o], ind)


def emit_ops(out, ops, ind="  "):
    pins = []
    for o in ops:
        out.append(ind + o.code)
        pins += [d for d in o.defs]
    for d in pins:
        out.append(ind + f'asm volatile("" ::"v"({d}));')


class Window:
    """K loop of one chunk (KS steps of 3 MFMAs) with filler ops in the MFMA shadows.

    b_src 'agpr': B operands are a[4s..] / a[64+4s..];  'vgpr': u32x4 expressions (bh(s), bl(s)) given by name pattern.
    hh_init: name of an f32x16 holding the start values (compiler-visible LDS loads), or None for zero."""

    def __init__(self, ks, hh, cc, b_src="agpr", bvar=("ebh", "ebl"), hh_zero=False, pf=PF, wa="wa", cd=None, use_ds=True, in_base=0,
                 acc_all=False, bias=None, b_lo_scaled=False):
        self.b_lo_scaled = b_lo_scaled   # B_lo carries a factor 2^11 as well (reflectance net): A_hi B_lo goes to cc, not hh
        self.ks, self.hh, self.cc, self.b_src, self.bvar, self.hh_zero, self.pf, self.wa = ks, hh, cc, b_src, bvar, hh_zero, pf, wa
        self.acc_all = acc_all  # every MFMA accumulates (hh and cc already hold partial sums)
        self.bias = bias        # (a_word, b_quad): one extra MFMA hh += {a_word,0,0,0} * b_quad ahead of the last K step (the bias row, see gen_stage)
        self.in_base = in_base  # AGPR set holding the B operands: 0 (a[0:127]) or 128 (a[128:255])
        self.cd = cd            # third accumulator (A_lo * B_hi products) - no back-to-back dependent MFMAs; None: they go to cc
        self.use_ds = use_ds    # False: micro-benchmarks without LDS traffic (fragments stay whatever they are)

    def frag(self, s, part):
        return f"fa{((s + self.rot) % (self.pf + 1)) * 2 + part}"

    @staticmethod
    def decl(pf=None):
        return "nrh32::u32x4 " + ", ".join(f"fa{i}" for i in range(((PF if pf is None else pf) + 1) * 2)) + ";"

    def emit(self, out, slots_ops, ind="  ", dma=None, head=None, declare=True, preloaded=False, prefetch_next=False, rot=0):
        """dma: {slot: [piece, ...]} - W32_DMA(piece) calls (LDS-DMA of a later block) issued in that slot.
        head: list of op lists emitted between the first LDS reads and the first MFMA (VALU work in the LDS latency: a SIMD
        does not overlap one wave's VALU with its MFMAs anyway, so what sits here is free up to that latency).
        Cross-window prefetch (gen_stage, XWIN): with prefetch_next the block barrier of the NEXT window (W32_SYNC_MID: this wave's
        pieces of the next block have landed, every fragment read of this block has returned, then s_barrier) sits behind the MFMAs
        of K step ks - 2, and the last K step issues the fragment reads of the next window's first pf steps (address W32_WADDR_NEXT)
        into the buffers the rotation would give steps ks, ks + 1, ... - the next window is emitted with preloaded=True and
        rot = (rot + ks) % (pf + 1): no barrier and no cold LDS reads at its top.  The fragment variables then live at stage scope
        (declare=False)."""
        ks, pf = self.ks, self.pf
        self.rot = rot
        dma = dma or {}
        assert all(k < 3 * ks for k in dma), "an LDS-DMA piece scheduled behind the window's last MFMA slot would never be issued"
        assert not prefetch_next or all(k < 3 * (ks - 1) for k in dma), "LDS-DMA pieces behind the block barrier (W32_SYNC_MID)"
        if declare:
            out.append(ind + self.decl(pf))
        issued = []  # LDS reads in issue order: (s, part)

        def ds(s, part, base=None, soff=None):
            off = (2 * (s if soff is None else soff) + part) * 1024
            issued.append((s, part))
            if not self.use_ds:
                return
            # "memory": compiler-visible LDS loads (the bias word) stay where they are written - one that hipcc sinks between
            # these reads would become one of the "N youngest" the K loop's lgkmcnt(N) waits deliberately leave in flight
            clob = ' : "memory"' if (self.bias or base is not None) else ""
            out.append(ind + f'asm volatile("ds_read_b128 %0, %1 offset:{off}" : "=v"({self.frag(s, part)}) : "v"({base or self.wa}){clob});')

        def wait_for(s, part):
            idx = issued.index((s, part))
            return len(issued) - 1 - idx

        for s in range(min(pf, ks)):
            if preloaded:
                # issued by the previous window's last K step, in this order (a compiler-visible LDS load between them and this
                # window's own reads - the bias word - is not counted: the waits then cover one read more than needed, never fewer)
                issued.append((s, 0))
                if not ONE_TERM:
                    issued.append((s, 1))
                continue
            ds(s, 0)
            if not ONE_TERM:
                ds(s, 1)
        for ops in (head or []):
            if ops:
                emit_ops(out, ops, ind)
                out.append(ind + "__builtin_amdgcn_sched_barrier(0);")
        slot = 0
        for s in range(ks):
            if self.bias and s == ks - 1:
                # The bias word is a compiler-visible LDS load from the top of the window; hipcc waits for it with lgkmcnt(0)
                # (it cannot see the asm reads), which is free HERE: the last K step needs every outstanding fragment anyway.
                # Not after the K loop: an MFMA result must not be the window's last write to hh (the next window's VALU
                # reads follow within the 11 wait states when no barrier separates the windows - measured on layer 0).
                out.append(ind + f"{{ const nrh32::u32x4 ba_ = {{{self.bias[0]}, 0u, 0u, 0u}};")
                out.append(ind + f'  asm volatile("{MFMA} %0, %1, %2, %0" : "+v"({self.hh}) : "v"(ba_), "v"({self.bias[1]})); }}')
                out.append(ind + "__builtin_amdgcn_sched_barrier(0);")
            for j in range(3):
                # weight fragments of K step s + pf go out in the first two slots of step s (their buffer was last read by
                # the MFMAs of step s - 1, all issued by now)
                if s + pf < ks and j < (1 if ONE_TERM else 2):
                    ds(s + pf, j)
                elif prefetch_next and s == ks - 1:
                    # the next window's K steps 0 .. pf - 1, as steps ks .. ks + pf - 1 of this one (pf = 2: one read ahead of each of
                    # the first two MFMAs, the second step's pair ahead of the third)
                    nxt = [(t, q) for t in range(pf) for q in ((0,) if ONE_TERM else (0, 1))]
                    per = (len(nxt) + 2) // 3
                    for t, q in (nxt[j * per:(j + 1) * per] if j < 2 else nxt[2 * per:]):
                        ds(ks + t, q, base="wa_next", soff=t)
                # j = 0: A_hi * B_hi -> hh;  j = MID: A_lo * B_hi (A_lo is scaled by 2^11) -> cc;  the other: A_hi * B_lo (B_lo is
                # unscaled) -> hh.  MID = 1 keeps the two hh updates of a K step apart (no back-to-back dependent MFMAs).
                lo_a = (j == MID)
                part = 1 if lo_a else 0
                to_cc = lo_a or (self.b_lo_scaled and j > 0)          # which accumulator: everything with one scaled factor -> cc
                acc = (self.cd if self.cd else self.cc) if to_cc else self.hh
                first_cc = 1 if self.b_lo_scaled else MID              # the first MFMA of the window that writes cc
                first = (s == 0 and ((to_cc and j == first_cc) or (j == 0 and self.hh_zero))) and not self.acc_all
                w = wait_for(s, 0 if ONE_TERM else 1)           # one wait per K step: both fragments (hi was issued first) before the first MFMA
                bpart = 1 if (j > 0 and not lo_a) else 0
                pre = f"s_waitcnt lgkmcnt({w})\\n\\t" if (j == 0 and self.use_ds) else ""
                if self.b_src == "agpr":
                    base = self.in_base + 4 * s + (64 if bpart else 0)
                    bop, bcons = f"a[{base}:{base + 3}]", ""
                else:
                    bop, bcons = "%2", f', "v"({self.bvar[bpart]}{s})'
                if ONE_TERM and j > 0:
                    pass                      # the cross terms do not exist; the slot keeps its epilogue work and DMA piece
                elif first:
                    out.append(ind + f'asm volatile("{pre}{MFMA} %0, %1, {bop}, 0" : "=&v"({acc}) : "v"({self.frag(s, part)}){bcons});')
                else:
                    out.append(ind + f'asm volatile("{pre}{MFMA} %0, %1, {bop}, %0" : "+v"({acc}) : "v"({self.frag(s, part)}){bcons});')
                for piece in dma.get(slot, []):
                    out.append(ind + f"W32_DMA({piece});")
                if slots_ops is not None and slot < len(slots_ops) and slots_ops[slot]:
                    emit_ops(out, slots_ops[slot], ind)
                out.append(ind + "__builtin_amdgcn_sched_barrier(0);")
                slot += 1
            if prefetch_next and s == ks - 2:
                out.append(ind + "W32_SYNC_MID();")
                out.append(ind + "const uint32_t wa_next = W32_WADDR_NEXT();")
                out.append(ind + "__builtin_amdgcn_sched_barrier(0);")
        # slots beyond the K loop (short K loops: the epilogue is longer than the MFMA stream)
        while slots_ops is not None and slot < len(slots_ops):
            if slots_ops[slot]:
                out.append(ind + PARTIAL_WRITE_GAP)
                emit_ops(out, slots_ops[slot], ind)
                out.append(ind + "__builtin_amdgcn_sched_barrier(0);")
            slot += 1


# regular window: one piece of block n + 2 behind the FIRST MFMA of K steps 6..13 - late in the window, where the epilogue (scheduled as
# early as its dependencies allow) has thinned out, and ahead of the block barrier behind K step 14.  Rounds 2-5 issued them behind the
# last MFMA of K steps 0..7 (NRH32_DMA_J=2 NRH32_DMA_FIRST=0): 521.1 -> 523.6 k rays/s interleaved (profiles/r06/eval_dma_slots_ab.log);
# knobs: NRH32_DMA_J = which of the K step's three slots, NRH32_DMA_FIRST = first K step, NRH32_DMA_STRIDE = every how many K steps
DMA_SLOTS16 = {int(os.environ.get("NRH32_DMA_J", "0")) + 3 * (int(os.environ.get("NRH32_DMA_FIRST", "6")) + int(os.environ.get("NRH32_DMA_STRIDE", "1")) * k): [k]
               for k in range(8)}
assert max(DMA_SLOTS16) < 42, "the LDS-DMA pieces of a window have to be issued ahead of its block barrier (behind K step 14)"
# a full block consumed in fewer K steps (the reflectance net's first stage, ks = 8): one piece behind the last MFMA of every K step
DMA_SLOTS_SHORT = {2 + 3 * k: [k] for k in range(8)}


# experiment: the reflectance net's activations with the UNSCALED residual as well - build with
# NRH32_COL_UNSCALED=1 in the generator's environment AND -DNRH32_COL_UNSCALED=1 for nrh_color32.hip (profiles/tools/build_colu_variant.sh)
COL_UNSCALED = bool(os.environ.get("NRH32_COL_UNSCALED"))
MID = int(os.environ.get("NRH32_MID", "1"))      # which of a K step's three MFMAs is the A_lo one (see Window.emit)
# VALU ops placed ahead of a window's first MFMA, in HEAD_SLOTS dependency levels of HEAD_OPS each (see Window.emit)
HEAD_SLOTS = int(os.environ.get("NRH32_HEAD_SLOTS", "0"))
HEAD_OPS = int(os.environ.get("NRH32_HEAD_OPS", "10"))
# cross-window prefetch of the weight fragments inside a stage (Window.emit); NRH32_XWIN=0: every window opens with its barrier and
# cold LDS reads (rounds 2-5)
XWIN = os.environ.get("NRH32_XWIN", "1") != "0"
# VALU micro-operations fewer in a slot that also issues an LDS-DMA piece (s_mov m0 + s_nop + global_load_lds)
DMA_PENALTY = int(os.environ.get("NRH32_DMA_PENALTY", "2"))


def acc_names(c):
    """accumulators of chunk c: chunk 7's are the stage's pending pair (hp, cp), consumed by the NEXT stage's window 0"""
    return ("hp", "cp") if c == 7 else (f"hh{c}", f"cc{c}")


# asm loads a stage issues per window for the epilogue that runs one window later: (register stems, macro)
STAGE_LOADS = {"rev": (("qa", "qb"), "W32_QLOAD_ASM"),
               "relu_part": (("pa", "pb", "pc", "pd"), "W32_PLOAD_ASM")}
LOAD_TYPE = {"rev": "nrh32::u32x4", "relu_part": "f32x4"}


def stage_epilogue(kind, c, ph, pc, want_d, out_base, qstore):
    if kind in ("fwd", "fwd_jvp"):
        return epi_fwd(c, ph, pc, want_d, out_base=out_base, qstore=qstore, jvp=(kind == "fwd_jvp"))
    if kind == "rev":
        return epi_rev(c, ph, pc, out_base=out_base)
    if kind in ("relu", "relu_part"):
        return epi_relu(c, ph, pc, out_base=out_base, part=(kind == "relu_part"))
    if kind == "feat":
        return epi_feat(c, ph, pc)
    raise ValueError(kind)


def gen_stage(kind, want_d, ks, b_src, nv, hh_zero, in_base=0, out_base=128, pend_in=True, bias_mfma=False, skip=False,
              pend_kind=None, full_block=False):
    """A pipelined 8-chunk stage, ping-pong form: B operands from AGPR set `in_base`, results into set `out_base`; no copy.
    Window c runs the K loop of chunk c and the epilogue of chunk c - 1; window 0 runs the epilogue of the PREVIOUS stage's
    chunk 7 (pending in hp, cp [, qpa, qpb]), whose outputs are this stage's K steps 14 and 15 - they are only read by the
    last six MFMAs of each window.  Chunk 7 of this stage stays pending in (hp, cp) for whoever comes next.
    ks = 16: one 32 KiB block per chunk.  ks = 3 (layer 0): four chunks per block (8 KiB each), the block changes at c = 4.
    Hooks (macros of the including kernel): W32_SYNC(), W32_FETCH_SETUP(), W32_DMA(piece), W32_NEXT(), W32_WADDR(),
    W32_HINIT(c) (f32x16 start values, unless hh_zero), W32_QSTORE(c, half, v) (this stage's layer), W32_QSTORE_P (the
    previous stage's layer), W32_QLOAD_ASM(dst, c, half)."""
    small = ks != 16 and not full_block      # full_block: fewer K steps, but still one 32 KiB block (and one sync) per chunk
    pend_kind = kind if pend_kind is None else pend_kind
    loads, load_macro = STAGE_LOADS.get(kind, ((), None))            # what this stage's windows request
    ploads = STAGE_LOADS.get(pend_kind, ((), None))[0]               # what arrives pending from the previous stage
    wname = {"rev": "qw", "relu_part": "pw"}
    out = []
    out.append(f"// generated by gen_mlp32.py: stage kind={kind} pend={pend_kind} want_d={want_d} ks={ks} b={b_src} valu/slot={nv} in=a{in_base} out=a{out_base}")
    out.append("{")
    # NRH32_SYNCK=1 (experiment): vmcnt counts the VMEM stores of the previous window's epilogue like the LDS-DMA pieces, so the
    # window's opening wait keeps that many more operations in flight instead of draining them with `s_waitcnt vmcnt(8)`.
    sync_k = bool(os.environ.get("NRH32_SYNCK"))
    prev_stores = 0
    # (16-step windows only: the block barrier behind K step ks - 2 has to come after ALL of the window's LDS-DMA pieces - its
    # `vmcnt(8)` counts them as the 8 youngest - and a shorter window issues one per K step up to its last)
    xwin = XWIN and ks == 16
    if xwin:
        out.append("  " + Window.decl())
    rot = 0
    for c in range(7):
        out.append(f"  nrh32::f32x16 hh{c}, cc{c};")
    for c in range(7):
        if loads:
            out.append(f"  {LOAD_TYPE[kind]} " + ", ".join(f"{st}{c}" for st in loads) + ";")
    ln = lambda c, stems: [(f"{st[0]}p{st[1:]}" if c == 7 else f"{st}{c}") for st in stems]   # chunk 7: qpa, qpb / ppa..ppd
    for c in range(8):
        out.append(f"  {{  // window {c}")
        hh, cc = acc_names(c)
        prev = (c - 1) % 8
        ph, pc = acc_names(prev)
        has_epi = (c > 0 or pend_in) and not os.environ.get("NRH32_NOEPI")
        ekind = kind if c > 0 else pend_kind
        if xwin and c > 0:
            out.append("    W32_FETCH_SETUP();")         # (the block barrier sat in the previous window: W32_SYNC_MID)
        elif not small or c % 4 == 0:
            if sync_k and not small and c > 0 and prev_stores:
                out.append(f"    W32_SYNC_K({8 + prev_stores});")
            elif sync_k and not small and c == 0 and pend_in:
                # what ran before this window is another stage's last window: the kernel knows whether it was a stage of the same
                # kind (then its stores count like ours: 8 + own_stores) or something else (then 8)
                own = sum(1 for o in stage_epilogue(pend_kind, 0, "hp", "cp", want_d, out_base, "W32_QSTORE") if o.kind == "vmem")
                out.append(f"    W32_SYNC_FIRST({8 + own});" if own else "    W32_SYNC();")
            else:
                out.append("    W32_SYNC();")
            out.append("    W32_FETCH_SETUP();")
        pnames = []
        if c == 0 and has_epi:
            # window 0 consumes the pending pair (and loaded words) of the previous stage; window 7 redefines those names
            # (the empty asm orders any copy hipcc makes of these registers behind W32_SYNC: the words were still in flight)
            out.append("    " + pin("hp", "cp"))
            out.append("    nrh32::f32x16 ph0 = hp, pc0 = cp;" if not ONE_TERM else "    nrh32::f32x16 ph0 = hp;")
            ph, pc = "ph0", "pc0"
            for nm in ln(7, ploads):
                out.append(f'    asm volatile("" : "+v"({nm}));')
                out.append(f"    {LOAD_TYPE[pend_kind]} c_{nm} = {nm};")
                pnames.append(f"c_{nm}")
        epi, head = None, None
        if has_epi:
            ob = out_base if c > 0 else in_base          # the previous stage's output set is this stage's input set
            epi = stage_epilogue(ekind, prev, ph, pc, want_d, ob, "W32_QSTORE" if c > 0 else "W32_QSTORE_P")
        if loads:
            # asm loads: hipcc does not see them, so it never waits for them with vmcnt(0) (which would also drain the
            # LDS-DMA pieces in flight); they are older than this window's 8 pieces: the next W32_SYNC covers them
            out.append("    " + " ".join(f"{load_macro}({nm}, {c}, {k});" for k, nm in enumerate(ln(c, loads))))
        if bias_mfma:
            out.append(f"    const uint32_t bw = W32_BIAS({c});")
        elif not hh_zero:
            out.append(f"    {hh} = W32_HINIT({c});")
        out.append("    const uint32_t wa = W32_WADDR()" + (f" + {(c % 4) * 8192};" if small else ";"))
        if epi is not None:
            # the previous chunk's accumulators are read by VALU only from here on: >= 11 wait states after its last MFMA
            out.append("    " + pin(ph, pc))
            estems = STAGE_LOADS.get(ekind, ((), None))[0]
            if estems:
                names = pnames if c == 0 else ln(prev, estems)
                for k, nm in enumerate(names):
                    out.append(f'    asm volatile("" : "+v"({nm}));')
                    out.append(f"    const {LOAD_TYPE[ekind]} {wname[ekind]}{k} = {nm};")
        win = Window(ks, hh, cc, b_src=b_src, hh_zero=(hh_zero or bias_mfma), in_base=in_base,
                     bias=(("bw", "W32_BCONST") if bias_mfma else None), b_lo_scaled=kind.startswith("relu") and not COL_UNSCALED)
        if small:
            dma = {2: [2 * (c % 4)], 5: [2 * (c % 4) + 1]}
        else:
            dma = DMA_SLOTS16 if ks == 16 else DMA_SLOTS_SHORT
        if epi is not None:
            w_nv = nv + 2 if (c == 0 and not small) else nv      # window 0: everything has to sit before K step 14 (slot 42)
            budget = (lambda k: max(1, w_nv - DMA_PENALTY) if k in dma else w_nv) if not small else w_nv
            nslots = max(3 * ks, (len(epi) + w_nv - 1) // w_nv + 8)
            if c == 0 and not small:
                budget0 = budget
                budget = lambda k: (budget0(k) if k < 41 else 0)
            if HEAD_SLOTS and not small:
                b1 = budget
                slots, tail = schedule(epi, HEAD_SLOTS + nslots, lambda k: HEAD_OPS if k < HEAD_SLOTS else b1(k - HEAD_SLOTS), TRANS_COST)
                head, slots = slots[:HEAD_SLOTS], slots[HEAD_SLOTS:]
            else:
                slots, tail = schedule(epi, nslots, budget, TRANS_COST)
            assert not (c == 0 and not small and tail), "pending epilogue does not fit ahead of K step 14"
        else:
            slots, tail = None, []
        prev_stores = sum(1 for o in (epi or []) if o.kind == "vmem")
        if xwin:
            win.emit(out, slots, "    ", dma=dma, head=head, declare=False, preloaded=(c > 0), prefetch_next=(c < 7), rot=rot)
            rot = (rot + ks) % (win.pf + 1)
        else:
            win.emit(out, slots, "    ", dma=dma, head=head)
        if tail:
            out.append("    // epilogue work that did not fit the MFMA shadows")
            emit_tail(out, tail, "    ")
        if skip:
            out.append(f"    W32_SKIP({c}, {hh}, {cc});")
        if not small or c % 4 == 3:
            out.append("    W32_NEXT();")
        out.append("  }")
    # whoever comes next may copy the pending registers: only after the last MFMAs have landed
    out.append('  asm volatile("s_nop 7\\n\\ts_nop 7" : "+v"(hp)' + ('' if ONE_TERM else ', "+v"(cp)') + ');')
    if loads:
        # same for the pending loaded words (asm loads of window 7, in flight, invisible to hipcc): they have landed before
        # anything may touch their registers (hipcc shuffles loop-carried registers at the loop edges - measured: 1-3 % of
        # the tiles got stale words without this).  They are older than window 7's eight LDS-DMA pieces.
        regs = ", ".join(f'"+v"({nm})' for nm in ln(7, loads))
        out.append(f'  asm volatile("s_waitcnt vmcnt({8 + (prev_stores if sync_k else 0)})" : {regs});')
    out.append("}")
    return "\n".join(out) + "\n"


def gen_finish(kind, want_d, out_base):
    """The pending chunk 7 of the last stage of a chain, on its own (no K loop to hide under): outputs into set `out_base`."""
    out = [f"// generated by gen_mlp32.py: finish kind={kind} want_d={want_d} out=a{out_base}", "{"]
    if kind in ("fwd", "fwd_jvp"):
        epi = epi_fwd(7, "hp", "cp", want_d, out_base=out_base, qstore="W32_QSTORE_P", jvp=(kind == "fwd_jvp"))
    elif kind == "relu":
        epi = epi_relu(7, "hp", "cp", out_base=out_base)
    elif kind == "feat":
        epi = epi_feat(7, "hp", "cp")
    else:
        out.append('  asm volatile("" : "+v"(qpa), "+v"(qpb));   // landed: the stage body ends with a wait for them')
        out.append("  const nrh32::u32x4 qw0 = qpa, qw1 = qpb;")
        epi = epi_rev(7, "hp", "cp", out_base=out_base)
    slots, tail = schedule(epi, (len(epi) + 7) // 8 + 8, 8)
    for ops in slots:
        if ops:
            out.append("  " + PARTIAL_WRITE_GAP)
            emit_ops(out, ops, "  ")
            out.append("  __builtin_amdgcn_sched_barrier(0);")
    assert not tail
    out.append("}")
    return "\n".join(out) + "\n"


def gen_kloop(ks, b_src, hh_zero, in_base=128, acc_all=False, b_lo_scaled=False):
    """K loop only (no fillers): for the light stages whose epilogue is written by hand after it.  Needs hh, cc, wa; the
    16-step form consumes a streamed block and therefore also issues the 8 LDS-DMA pieces of block n + 2 (W32_DMA)."""
    out = [f"// generated by gen_mlp32.py: bare K loop ks={ks} b={b_src}", "{"]
    Window(ks, "hh", "cc", b_src=b_src, hh_zero=hh_zero, in_base=in_base, acc_all=acc_all, b_lo_scaled=b_lo_scaled).emit(out, None, "  ", dma=(DMA_SLOTS16 if ks == 16 else None))
    if not acc_all:
        out.append('  asm volatile("s_nop 7\\n\\ts_nop 7" : "+v"(hh)' + ('' if ONE_TERM else ', "+v"(cc)') + ');   // MFMA results -> VALU reads: 11 wait states')
    out.append("}")
    return "\n".join(out) + "\n"


def gen_swap():
    out = ["// generated by gen_mlp32.py: out -> in (128 AGPR moves)"]
    for i in range(128):
        out.append(f'asm volatile("v_accvgpr_mov_b32 a{i}, a{128 + i}" ::: "a{i}");')
    return "\n".join(out) + "\n"


def gen_t7_loads():
    """The q_7 words of chunks 0..6 for the T7 pass, issued from inside the HEAD window (W32_QLOAD7_ASM(dst, c, half)) so that
    their L2 latency passes under HEAD's K loop: 56 VGPRs that nothing else needs there.  Declared by the kernel (T7Q_DECL)."""
    out = ["// generated by gen_mlp32.py: T7 loads"]
    for c in range(7):
        out.append(f"W32_QLOAD7_ASM(q{c}a, {c}, 0); W32_QLOAD7_ASM(q{c}b, {c}, 1);")
    return "\n".join(out) + "\n"


def gen_t7():
    """t_7 = (1 - q_7) * a8 written straight into AGPR set 0 (R7's input): W32_A8(c) -> f32x16 (w_s / 3 in D32 layout).  The q
    words were requested in the HEAD window (gen_t7_loads); they are older than HEAD's eight LDS-DMA pieces."""
    out = ["// generated by gen_mlp32.py: T7 pass train=False", "{"]
    regs = ", ".join(f'"+v"(q{c}{h})' for c in range(7) for h in "ab")
    out.append(f'  asm volatile("s_waitcnt vmcnt(8)" : {regs}, "+v"(qpa), "+v"(qpb) :: "memory");   // asm loads: the wait is ours')
    out.append("  __builtin_amdgcn_sched_barrier(0);")
    for c in range(7):   # chunk 7 becomes the pending pair of the reverse chain (hp = a8, cp = 0, q words): R7's window 0 finishes it
        out.append(f"  {{  // chunk {c}")
        out.append(f"    const nrh32::f32x16 a8 = W32_A8({c});")
        out.append(f"    const nrh32::u32x4 qw0 = q{c}a, qw1 = q{c}b;")
        ops = []
        for r in range(16):
            w = f"qw{r // 8}[{(r % 8) // 2}]"
            ext = f"({w} >> 16)" if r & 1 else f"({w} & 0xffffu)"
            ops += [Op(f"float f{r} = (float){ext};", defs=(f"f{r}",)),
                    Op(f"float n{r} = a8[{r}] * (-1.0f / 65535.0f);", defs=(f"n{r}",)),
                    Op(f"float u{r} = __builtin_fmaf(n{r}, f{r}, a8[{r}]);", defs=(f"u{r}",))]
        sp = [split_ops(i, f"u{2 * i}", f"u{2 * i + 1}", 8 * c + i, 64 + 8 * c + i) for i in range(8)]
        # the eight pairs stage by stage (all high halves, all residuals, all AGPR writes): no operation directly behind the one that
        # produces its input (the residual pair is a half-register write, see split_ops)
        for k in range(max(len(x) for x in sp)):
            ops += [x[k] for x in sp if k < len(x)]
        for o in ops:
            out.append("    " + o.code)
        out.append("    __builtin_amdgcn_sched_barrier(0);")
        out.append("  }")
    out.append("}")
    return "\n".join(out) + "\n"


def gen_dump():
    """debug: `in` AGPRs -> W32_DUMP(i, value) for i in 0..127"""
    out = ["// generated by gen_mlp32.py: debug dump of a[0:127]"]
    for i in range(128):
        out.append(f'{{ uint32_t x; asm volatile("v_accvgpr_read_b32 %0, a{i}" : "=v"(x)); W32_DUMP({i}, x); __builtin_amdgcn_sched_barrier(0); }}')
    return "\n".join(out) + "\n"


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "gen32")
    os.makedirs(outdir, exist_ok=True)
    nv = int(os.environ.get("NRH32_NV", "4"))
    nv_rev = int(os.environ.get("NRH32_NV_REV", str(nv)))       # the reverse stages' budget (their epilogue is the shortest)
    nv_d1 = int(os.environ.get("NRH32_NV_D1", str(nv + 1)))     # forward stages that also encode and store sigma'
    bm = True      # bias through one extra MFMA per window (see gen_stage); the older start-value form is gone from the kernel
    nohinit = bool(os.environ.get("NRH32_NOHINIT"))      # timing ablation (WRONG RESULTS): forward windows start from zero
    files = {
        "l0_d0.inc": gen_stage("fwd", False, 3, "vgpr", 10, False, in_base=0, out_base=0, pend_in=False, bias_mfma=bm),
        "l0_d1.inc": gen_stage("fwd", True, 3, "vgpr", 10, False, in_base=0, out_base=0, pend_in=False, bias_mfma=bm),
        "fwd_d0_p0.inc": gen_stage("fwd", False, 16, "agpr", nv, nohinit, in_base=0, out_base=128, bias_mfma=bm),
        "fwd_d0_p1.inc": gen_stage("fwd", False, 16, "agpr", nv, nohinit, in_base=128, out_base=0, bias_mfma=bm, skip=bm),
        "fwd_d1_p0.inc": gen_stage("fwd", True, 16, "agpr", nv_d1, nohinit, in_base=0, out_base=128, bias_mfma=bm),
        "fwd_d1_p1.inc": gen_stage("fwd", True, 16, "agpr", nv_d1, nohinit, in_base=128, out_base=0, bias_mfma=bm, skip=bm),
        "fwd_fin_d0.inc": gen_finish("fwd", False, 128),
        # forward-mode variant for rays that only need the derivative along the ray (shadow march): 16 points + 16 tangents per tile
        "l0_j.inc": gen_stage("fwd_jvp", False, 3, "vgpr", 12, False, in_base=0, out_base=0, pend_in=False, bias_mfma=bm),
        "fwd_j_p0.inc": gen_stage("fwd_jvp", False, 16, "agpr", nv + 2, nohinit, in_base=0, out_base=128, bias_mfma=bm),
        "fwd_j_p1.inc": gen_stage("fwd_jvp", False, 16, "agpr", nv + 2, nohinit, in_base=128, out_base=0, bias_mfma=bm, skip=bm),
        "fwd_fin_j.inc": gen_finish("fwd_jvp", False, 128),
        "fwd_fin_d1.inc": gen_finish("fwd", True, 128),
        "rev_p0.inc": gen_stage("rev", True, 16, "agpr", nv_rev, True, in_base=0, out_base=128),
        "rev_p1.inc": gen_stage("rev", True, 16, "agpr", nv_rev, True, in_base=128, out_base=0),
        "rev_fin.inc": gen_finish("rev", True, 128),
        "kloop16.inc": gen_kloop(16, "agpr", False),
        "kloop16z.inc": gen_kloop(16, "agpr", True),
        "kloop3v.inc": gen_kloop(3, "vgpr", False),
        "kloop3v_acc.inc": gen_kloop(3, "vgpr", False, acc_all=True),
        "feat.inc": gen_stage("feat", False, 16, "agpr", nv, True, in_base=128, out_base=128, pend_in=False, bias_mfma=True),
        "feat_fin.inc": gen_finish("feat", False, 128),
        "t7.inc": gen_t7(),
        "t7_loads.inc": gen_t7_loads(),
        # reflectance net on the same machinery (csrc/nrh_color32.hip): C0 (misc inputs, + the feature block's share loaded per
        # window) -> C1 -> C2 -> C3 (ReLU), then the 3-row output chunk as a bare K loop
        "col_c0.inc": gen_stage("relu_part", False, 8, "agpr", 6, True, in_base=0, out_base=128, pend_in=False, bias_mfma=True, full_block=True),
        "col_c1.inc": gen_stage("relu", False, 16, "agpr", nv, True, in_base=128, out_base=0, bias_mfma=True, pend_kind="relu_part"),
        "col_c2.inc": gen_stage("relu", False, 16, "agpr", nv, True, in_base=0, out_base=128, bias_mfma=True),
        "col_c3.inc": gen_stage("relu", False, 16, "agpr", nv, True, in_base=128, out_base=0, bias_mfma=True),
        "col_fin.inc": gen_finish("relu", False, 0),
        "kloop16_a0.inc": gen_kloop(16, "agpr", False, in_base=0, b_lo_scaled=not COL_UNSCALED),
    }
    for stale in ("fwd_d0.inc", "fwd_d1.inc", "rev.inc", "swap.inc", "dump_in.inc", "l0_t.inc", "fwd_t_p0.inc", "fwd_t_p1.inc", "fwd_fin_t.inc",
                  "rev_t_p0.inc", "rev_t_p1.inc", "rev_t_fin.inc", "feat_t.inc", "feat_t_fin.inc", "t7_t.inc"):
        if os.path.exists(os.path.join(outdir, stale)):
            os.remove(os.path.join(outdir, stale))
    for name, text in files.items():
        path = os.path.join(outdir, name)
        old = open(path).read() if os.path.exists(path) else None
        if old != text:
            open(path, "w").write(text)
    print(f"gen_mlp32: {len(files)} files in {outdir}")


if __name__ == "__main__":
    main()
