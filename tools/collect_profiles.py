#!/usr/bin/env python3
"""Copy what tools/exp/final_profiles.sh left under gpurun_out/final/ into profiles/ (round given as argument):
bench line, bench line under rocprofv3, kernel stats, PMC traffic, byte-stream microbenchmark rows and kernel stats."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "final")
    out = os.path.join(ROOT, "profiles")
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(out, f"r{rnd}_bench.json"))
    shutil.copy(os.path.join(src, "bench_rocprof.json"), os.path.join(out, f"r{rnd}_bench_under_rocprof.json"))
    one = lambda pat: sorted(glob.glob(os.path.join(src, pat)))[-1]
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_profiles.py"), rnd, one("stats/vm/*_kernel_stats.csv"),
                           one("fetch/vm/*counter_collection.csv"), one("write/vm/*counter_collection.csv")], stdout=subprocess.DEVNULL)
    for name in ("entrypoints", "lanes", "lengths"):
        if os.path.exists(os.path.join(src, name + ".log")):
            shutil.copy(os.path.join(src, name + ".log"), os.path.join(out, f"r{rnd}_microbench_{name}.log"))
    if os.path.isdir(os.path.join(src, "pmc_stream")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(out, f"r{rnd}_pmc_stream.json")]
                              + sorted(glob.glob(os.path.join(src, "pmc_stream", "*"))), stdout=subprocess.DEVNULL)
    for line in open(os.path.join(src, "microbench.log")):
        for tag in ("stream", "mixed", "query"):
            if line.startswith(tag + " "):
                json.dump(json.loads(line.split(" ", 1)[1]), open(os.path.join(out, f"r{rnd}_microbench_{tag}.json"), "w"), indent=1)
    rows = list(csv.DictReader(open(one("stream_stats/vm/*_kernel_stats.csv"))))
    with open(os.path.join(out, f"r{rnd}_stream_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(r for r in rows if "sc::k_" in r["Name"])
    d = json.load(open(os.path.join(out, f"r{rnd}_bench.json")))
    r = d["roofline"]
    print("decode", d["value"], r["frac"], r["min_frac"], r.get("min_frac_fastest_launch"), r["traffic_over_algorithmic"])
    print("encode", d["encode"]["value"], d["encode"]["frac"], d["encode"]["min_frac"])
    print("callers", {k: v["ms"] for k, v in d["callers"].items() if isinstance(v, dict) and "ms" in v}, d["box"].get("card0/unique_id"))
    s = json.load(open(os.path.join(out, f"r{rnd}_microbench_stream.json")))
    print({k: v["ms"] for k, v in s.items() if isinstance(v, dict) and "ms" in v})


if __name__ == "__main__":
    main()
