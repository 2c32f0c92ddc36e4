"""Build libfcvsr_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache: the .so travels with the snapshot)."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfcvsr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-value", "-ffp-contract=off"] + os.environ.get("FCVSR_EXTRA_FLAGS", "").split()
# Packed-FP32 VALU ops (v_pk_add/mul/fma_f32, formed by the SLP vectoriser) that consume freshly returned LDS data were
# measured to produce wrong lanes when a workgroup of another HW queue keeps the CU's LDS busy (DESIGN.md section 6,
# "Multi-stream replays"; scripts/lds_pk_hazard.hip is the torch-free reproducer).  No file keeps the vectoriser by default:
# the fused IAC kernel (iac.hip) has the same instruction pattern and used to be exempt for 1.5 % of end-to-end throughput; it
# never failed, but "never observed" is not a guarantee.  tests/test_host_logic.py disassembles the built library and fails
# if any code object contains a packed-FP32 instruction.  FCVSR_SLP_FILES=a.hip,b.hip re-enables it per file (experiments).
SLP_FILES = set(f for f in os.environ.get("FCVSR_SLP_FILES", "").split(",") if f)


def flags_for(src: str):
    return FLAGS if os.path.basename(src) in SLP_FILES else FLAGS + ["-fno-slp-vectorize"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)) + [os.path.join("..", "..", "include", "fcvsr_hip.h")]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    h.update((" ".join(FLAGS) + "|" + ",".join(sorted(SLP_FILES))).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    stamp_file = LIB + ".stamp"
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(LIBDIR, os.path.basename(src) + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([HIPCC, *flags_for(src), "-c", src, "-o", obj], stdout=subprocess.PIPE,
                                            stderr=subprocess.STDOUT)))
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs])
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
