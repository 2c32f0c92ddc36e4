#!/usr/bin/env python3
"""Static audit of how a kernel's global loads are grouped: compiles a .hip file to gfx950 assembly and prints, per kernel, the
run-length sequence of global loads / vmcnt waits / barriers / stores (first N items) plus the VGPR count, scratch and
occupancy.  A kernel whose loads come in many small groups each followed by `vmcnt(0)` pays one memory round trip per group
(run-time dtype switches, predicated loads with arithmetic between them and per-element null tests all caused this here).
usage: scripts/isa_load_audit.py fcvsr_amd/csrc/fft.hip [name-substring] [max-items]"""
import os, re, subprocess, sys, tempfile

def main(src, filt="", nmax=40):
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = os.path.join(tempfile.gettempdir(), os.path.basename(src) + ".s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-ffp-contract=off",
           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "fcvsr_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
    for k in re.findall(r"^(_Z\w+):", s, re.M):
        i = s.index("\n" + k + ":")
        blk = s[i:]
        if "s_endpgm" not in blk:
            continue
        body = blk[:blk.index("s_endpgm")]
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        if filt and filt not in name:
            continue
        g = lambda n: (re.search(r"; " + n + r": (\d+)", blk) or [None, "?"])[1]
        seq = [m.group(1) for m in re.finditer(r"^\s+(global_load_\w+|buffer_load_\w+|s_waitcnt vmcnt\(\d+\)|s_barrier|global_store_\w+|scratch_\w+|s_cbranch_\w+)", body, re.M)]
        items, last, cnt = [], None, 0
        for x in seq:
            x = {"s_barrier": "BAR"}.get(x, x).replace("global_load_", "L:").replace("global_store_", "S:").replace("s_waitcnt vmcnt", "W").replace("s_cbranch_", "br:")
            if x == last:
                cnt += 1
            else:
                if last: items.append(f"{last}x{cnt}" if cnt > 1 else last)
                last, cnt = x, 1
        if last: items.append(f"{last}x{cnt}" if cnt > 1 else last)
        full_waits = sum(1 for a, b in zip(items, items[1:]) if a.startswith("L:") and b == "W(0)")
        print(f"{name[:110]}\n   vgpr {g('NumVgprs')} scratch {g('ScratchSize')} occ {g('Occupancy')}  load-groups followed by a full wait: {full_waits}")
        print("   " + " ".join(items[:nmax]))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", int(sys.argv[3]) if len(sys.argv) > 3 else 40)
