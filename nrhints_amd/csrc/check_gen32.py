#!/usr/bin/env python3
"""Static check of the generated wide-kernel schedules (gen32*/*.inc, written by gen_mlp32.py): the LDS counter arithmetic.

The generated windows wait for weight fragments with `s_waitcnt lgkmcnt(N)` computed by the generator ("all but the N youngest
LDS reads have returned"); a wrong N is a RACE - an MFMA that reads a fragment register before its ds_read_b128 has landed - which
the numeric tests can miss (the data usually arrives in time).  This walks every file in statement order and models the in-order
LDS queue exactly as the hardware counts it:

  * `ds_read_b128 ... "=v"(faK)`          the read joins the queue; faK is "in flight";
  * `s_waitcnt lgkmcnt(N)` (a statement of its own or the prefix of an MFMA statement): everything but the N youngest has landed;
  * `W32_SYNC()` / `W32_SYNC_MID()`       lgkmcnt(0) (csrc/nrh_mlp32.h chunk_sync);
  * an MFMA statement whose A operand is faK: faK must have landed - else the file is reported - and must hold the fragment this
    MFMA is due (window = definition of `wa`, K step and half from the MFMA's position in the window; reads through `wa_next` belong to
    the next window): a dropped or misdirected read is reported even where stale data would have been "ready".

Also checked (check_dma): every block set up by W32_FETCH_SETUP() gets exactly its eight LDS-DMA pieces, all of them ahead of a block
barrier placed inside the window.

The walk is sequential over a file, so the cross-window prefetch (a window that opens without reads of its own finds its first
fragments requested by the window before it, through `wa_next`) falls under the same two rules.

LDS operations the walk does not see (the compiler-visible bias word, a skip block's own reads) only ever add to the hardware's
outstanding count, which makes a given lgkmcnt(N) wait for MORE, never less: ignoring them is the worst case.

    python3 check_gen32.py <dir> [<dir> ...]      exit status 1 if any file has a problem
"""
import glob
import os
import re
import sys

RE_DS = re.compile(r'asm volatile\("ds_read_b128 %0, %1 offset:(\d+)" : "=v"\((fa\d+)\) : "v"\((\w+)\)')
RE_WAIT = re.compile(r's_waitcnt lgkmcnt\((\d+)\)')
RE_MFMA = re.compile(r'v_mfma_f32_32x32x16_f16 %0, %1, [^"]*" : "[^"]*"\(\w+\) : "v"\((\w+)\)')
MID = int(os.environ.get("NRH32_MID", "1"))          # which of a K step's three MFMAs takes the low fragment (gen_mlp32.py MID)


def check_text(text):
    """-> (number of MFMA operand reads checked, list of problems)"""
    n_ds = len(RE_DS.findall(text))
    n_mf = sum(1 for m in RE_MFMA.finditer(text) if m.group(1).startswith("fa"))
    one_term = n_ds > 0 and n_mf / n_ds < 1.25          # one MFMA per fragment read, against three per two
    queue = []          # outstanding reads, oldest first: register names
    inflight = set()
    holds = {}          # register -> (window id, byte offset) of the fragment last requested into it
    window, nth = 0, 0  # window id (one per definition of `wa`), fragment-reading MFMAs seen in it
    problems, checked = [], 0
    for ln, line in enumerate(text.split("\n"), 1):
        if "W32_SYNC" in line and "define" not in line:
            queue, inflight = [], set()
            continue
        if re.search(r"const uint32_t wa = ", line):
            window, nth = window + 1, 0
            continue
        m = RE_DS.search(line)
        if m:
            off, reg, base = int(m.group(1)), m.group(2), m.group(3)
            queue.append(reg)
            inflight.add(reg)
            holds[reg] = (window + 1 if base == "wa_next" else window, off)
            continue
        w = RE_WAIT.search(line)
        if w:
            n = int(w.group(1))
            if n < len(queue):
                queue = queue[len(queue) - n:] if n else []
            inflight = set(queue)
        m = RE_MFMA.search(line)
        if m and m.group(1).startswith("fa"):
            reg = m.group(1)
            checked += 1
            if reg in inflight:
                problems.append(f"line {ln}: MFMA reads {reg} while its ds_read_b128 may still be in flight "
                                f"({len(queue)} reads outstanding)")
            s, j = (nth, 0) if one_term else divmod(nth, 3)
            want = (window, (2 * s + (1 if (j == MID and not one_term) else 0)) * 1024)
            if holds.get(reg) != want:
                problems.append(f"line {ln}: MFMA {nth} of window {window} reads {reg} = fragment {holds.get(reg)}, expected {want}")
            nth += 1
    problems += check_dma(text)
    return checked, problems


def check_dma(text):
    """Every W32_FETCH_SETUP() (one 32 KiB block of the weight stream) is followed by exactly the pieces 0..7 of W32_DMA before the next
    one, and a block barrier inside a window (W32_SYNC_MID: `s_waitcnt vmcnt(8)` = "everything but this window's 8 pieces has landed")
    comes after all 8."""
    problems, pieces, start = [], None, 0
    def close(ln):
        if pieces is not None and sorted(pieces) != list(range(8)):
            problems.append(f"line {start}: block set up here issues LDS-DMA pieces {sorted(pieces)} before line {ln}, expected 0..7 once each")
    for ln, line in enumerate(text.split("\n"), 1):
        if "W32_FETCH_SETUP()" in line and "define" not in line:
            close(ln)
            pieces, start = [], ln
        elif pieces is not None and "W32_SYNC_MID()" in line and "define" not in line:
            if len(pieces) != 8:
                problems.append(f"line {ln}: block barrier behind {len(pieces)} of the window's 8 LDS-DMA pieces (its vmcnt(8) would let a piece of the NEXT block stay in flight)")
        else:
            for m in re.finditer(r"W32_DMA\((\d+)\)", line):
                if pieces is not None:
                    pieces.append(int(m.group(1)))
    close(len(text.split("\n")))
    return problems


def check_dir(d):
    bad = 0
    total = 0
    for path in sorted(glob.glob(os.path.join(d, "*.inc"))):
        n, problems = check_text(open(path).read())
        total += n
        for p in problems[:5]:
            print(f"{path}: {p}")
        bad += len(problems)
    print(f"{d}: {total} MFMA fragment reads checked against the in-order lgkmcnt model, {bad} problem(s)")
    return bad


if __name__ == "__main__":
    sys.exit(1 if sum(check_dir(d) for d in sys.argv[1:]) else 0)
