"""Structural checks on the compiled block-codec kernels (hipcc cross-compiles
for gfx950 without a GPU): the fast decoders and encoders must not spill to
scratch memory, must address global memory with global (not flat)
instructions, and the uniform-width decoders must have every load in flight
before their first full wait -- the properties their throughput rests on."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "simdcomp_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "codec_kernels.s")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "-DSIMDCOMP_HIP_BUILDING", "--offload-arch=gfx950", "-S", "--cuda-device-only",
           os.path.join(CSRC, "codec_kernels.hip"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels, name, body = {}, None, []
    for line in open(out):
        m = re.match(r"^(_ZN2sc\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name and line.startswith(".Lfunc_end"):
            kernels[name] = body
            name = None
            continue
        if name:
            t = line.split()
            if t and not t[0].startswith((";", ".")):
                body.append(line.strip())
    meta = open(out).read()
    return kernels, meta


def fast(kernels):
    return {k: v for k, v in kernels.items() if "k_unpack_fast" in k or "k_pack_fast" in k}


def test_fast_kernels_exist_for_every_width(asm):
    kernels, _ = asm
    names = fast(kernels)
    for b in range(33):
        assert any(re.search(rf"k_unpack_fastILi{b}ELi1E", n) for n in names), b  # d1 decoder at width b
        assert any(re.search(rf"k_pack_fastILi{b}ELi1E", n) for n in names), b
    assert any("k_unpack_fastILin1E" in n for n in names)  # per-block widths


def test_no_flat_memory_instructions_in_fast_kernels(asm):
    kernels, _ = asm
    bad = [n for n, body in fast(kernels).items() if any(i.startswith(("flat_load", "flat_store", "scratch_")) for i in body)]
    assert not bad, bad[:5]


def test_no_scratch_in_fast_kernels(asm):
    _, meta = asm
    # .private_segment_fixed_size per kernel in the metadata block
    for m in re.finditer(r"\.name:\s+(_ZN2sc\w+)\n(?:.*\n){0,12}?\s+\.private_segment_fixed_size:\s+(\d+)", meta):
        if "k_unpack_fast" in m.group(1) or "k_pack_fast" in m.group(1):
            assert m.group(2) == "0", m.group(1)


def test_uniform_decoders_issue_all_loads_before_a_full_wait(asm):
    """In the decode path (after the prefetch workgroups' early return) no `s_waitcnt vmcnt(0)` may stand between two
    global loads: that is what a load stuck behind a branch, or a wait in front of the payload load, looks like."""
    kernels, _ = asm
    checked = 0
    for n, body in kernels.items():
        m = re.search(r"k_unpack_fastILi(\d+)ELi([012])ELi\dELi\d+ELi0ELi0E", n)  # no prefetch groups, 4-lane layout
        if not m or m.group(1) in ("0",):
            continue
        ops = [i.split()[0] + (" " + i.split()[1] if i.startswith("s_waitcnt") else "") for i in body]
        loads = [j for j, o in enumerate(ops) if o.startswith("global_load")]
        assert loads, n
        between = ops[loads[0]:loads[-1]]
        assert not any(o.startswith("s_waitcnt") and "vmcnt(0)" in o for o in between), (n, between)
        checked += 1
    assert checked >= 50
