#!/usr/bin/env python3
"""Is the slow / fast state of a process a power-management state?  Decodes b = 10 and b = 1 in a loop and reads the
clock tables and sensors sysfs offers while the GPU is busy."""
import glob
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from simdcomp_amd import codec  # noqa: E402

n = 1 << 30
nblk = n // 128
dev = torch.device("cuda:0")
out = torch.empty(n, dtype=torch.int32, device=dev)
corp = {}
for b in (10, 1):
    gaps = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev) & ((1 << b) - 1)
    ids = torch.cumsum(gaps, 0, dtype=torch.int32)
    del gaps
    offs = torch.empty(nblk, dtype=torch.int32, device=dev)
    corp[b] = (codec.packd1(ids, b, initvalue=0, masked=False, initvalues_out=offs), offs)
    del ids
torch.cuda.synchronize()


def star(text):
    return " ".join(l.strip() for l in text.splitlines() if "*" in l) or text.strip().replace("\n", " | ")[:80]


def sensors():
    res = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "pp_dpm_dcefclk", "gpu_busy_percent",
                     "mem_busy_percent"):
            try:
                res[name] = star(open(os.path.join(d, name)).read())
            except OSError:
                pass
        for h in glob.glob(os.path.join(d, "hwmon/hwmon*")):
            for name in ("power1_average", "power1_input", "temp1_input", "temp2_input", "temp3_input", "freq1_input", "freq2_input"):
                try:
                    res[name] = open(os.path.join(h, name)).read().strip()
                except OSError:
                    pass
    return res


for b in (10, 1, 10):
    pay, offs = corp[b]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 600
    e0.record()
    for _ in range(reps):
        codec.unpackd1(offs, pay, nblk, b, out=out)
    e1.record()
    time.sleep(0.25)  # the queue is some 0.5 s deep: the GPU is busy now
    s = sensors()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("b = %2d: %.1f %% of HBM peak  %s" % (b, nblk * (516 + 16 * b) / (ms * 1e-3) / 8e12 * 100, s), flush=True)
