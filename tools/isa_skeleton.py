#!/usr/bin/env python3
"""Print the memory skeleton of the compiled kernels: per kernel, the order of
global loads / stores, LDS operations, s_waitcnt vmcnt(...) and branches, with
runs collapsed.  This is the view that shows a load stuck behind a branch
(`s_cbranch... | global_load | s_waitcnt vmcnt(0)` repeated per step) or a
wait in front of the loads it should overlap with.

    python tools/isa_skeleton.py [--file codec_kernels|stream_kernels|query_kernels] [--match k_unpack_fast] [--define SC_X=1]

Needs hipcc only (cross-compiles for gfx950, no GPU).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "simdcomp_amd", "csrc")
KEEP = ("global_load", "global_store", "flat_load", "flat_store", "buffer_", "ds_", "s_waitcnt", "s_cbranch", "s_barrier",
        "v_readlane", "s_endpgm")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        lines = out.stdout.splitlines()
        return lines if len(lines) == len(names) else names
    except OSError:
        return names


def skeleton(asm_path, match):
    name, seq, result = None, [], []
    valu = 0
    for line in open(asm_path):
        m = re.match(r"^(_Z[\w$.]*):", line)
        if m:
            name, seq, valu = m.group(1), [], 0
            continue
        if name and line.startswith(".Lfunc_end"):
            result.append((name, seq, valu))
            name = None
            continue
        t = line.split()
        if not name or not t:
            continue
        op = t[0]
        if op.startswith("v_"):
            valu += 1
        if op.startswith(KEEP):
            if op == "s_waitcnt":
                if "vmcnt" not in line:
                    continue
                op += " " + " ".join(x for x in t[1:] if x.startswith("vmcnt"))
            seq.append(op.replace("global_", "g_"))
    names = demangle([r[0] for r in result])
    for (raw, seq, valu), nice in zip(result, names):
        if match and match not in nice and match not in raw:
            continue
        runs = []
        for op in seq:
            if runs and runs[-1][0] == op:
                runs[-1][1] += 1
            else:
                runs.append([op, 1])
        print(f"{nice.split('(')[0]}   [{valu} vector ALU]")
        print("   " + " | ".join(f"{a} x{n}" if n > 1 else a for a, n in runs))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--file", default="codec_kernels")
    ap.add_argument("--match", default="")
    ap.add_argument("--define", action="append", default=[])
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-DSIMDCOMP_HIP_BUILDING", "--offload-arch=gfx950", "-S",
               "--cuda-device-only", os.path.join(CSRC, a.file + ".hip"), "-o", asm] + ["-D" + d for d in a.define]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stderr)
        skeleton(asm, a.match)


if __name__ == "__main__":
    main()
