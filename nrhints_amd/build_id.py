"""Identity of the library's sources: ONE hash over every file libnrhints_hip.so is compiled or generated from.

The Makefile embeds it in the binary (``python3 ../build_id.py`` -> ``-DNRH_SOURCE_HASH``, returned by ``nrh_source_hash()``);
``_lib.load()`` recomputes it from the tree and REFUSES a library built from other sources - the built ``.so`` is git-ignored and
travels to the GPU box as a file, so without this check nothing ties the binary that runs to the tree that is judged
(tests/test_gpu_parity.py::test_native_library_loaded, bench.py's ``library_source_hash``).  Stand-alone on purpose: no package
import, usable from make."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "nrhints_hip.h")


def source_files():
    """Hand-written sources only (the generated gen32*/ schedules are a function of gen_mlp32.py, the objects of these files)."""
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")))
    files += [os.path.join(CSRC, n) for n in ("gen_mlp32.py", "check_wide_isa.py", "check_gen32.py", "Makefile")]
    return files + [HEADER]


def source_hash() -> str:
    h = hashlib.sha256()
    for path in source_files():
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
