"""`python bench.py --gpus N` with no torchrun around it must start N ranks itself.  --selftest runs that launcher end
to end on CPU: N child processes, a gloo process group, block-range shards from the library's partitioner, and the
verification all-reduce, without any kernel."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from simdcomp_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M64 = (1 << 64) - 1


def run_selftest(gpus, env_extra=None, args=()):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--selftest", *args],
                          capture_output=True, text=True, timeout=300, env=env)


def whole_checksum(total_blocks):
    words = synth.rand_u32(5, total_blocks * 128)
    idx = np.arange(total_blocks * 128, dtype=np.uint64) << np.uint64(32)
    return int(synth.splitmix64(idx ^ words.astype(np.uint64)).sum(dtype=np.uint64)) & M64


@pytest.mark.parametrize("gpus", [1, 2, 3])
def test_launcher_starts_n_ranks(gpus):
    out = run_selftest(gpus)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # only rank 0 prints on stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == gpus and doc["backend"] == "gloo"
    assert doc["blocks"] == gpus * 1001
    # shards are contiguous and equal: firsts are 0, 1001, 2002, ...
    assert doc["sum_of_first"] == 1001 * gpus * (gpus - 1) // 2
    assert doc["checksum"] == whole_checksum(gpus * 1001)
    # shards by bytes (simdcomp_hip_partition_by_bytes over the mixed-width corpus description): contiguous, cover
    # the corpus once, and no shard's cost is further from the mean than one block's
    bb = doc["by_bytes"]
    assert bb["contiguous"] and bb["blocks"] == gpus * 1001
    assert bb["cost_bytes"] == bb["whole_cost_bytes"] and bb["sum_of_widths"] == bb["whole_sum_of_widths"]
    mean = bb["whole_cost_bytes"] / gpus
    assert all(abs(c - mean) <= 512 + 16 * 20 for c in bb["cost_per_rank"]), bb
    if gpus > 1:  # the corpus gets wider along the way: equal bytes are not equal block counts
        assert bb["blocks_per_rank"][0] > bb["blocks_per_rank"][-1], bb


def test_world_size_must_match_gpus():
    """Under a launcher that set WORLD_SIZE (torch.distributed.run), --gpus has to agree with it."""
    out = run_selftest(2, env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_PORT": "29533"})
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_failing_rank_fails_the_launch():
    """Ranks that die take the launch down with a non-zero exit code; and a missing HIP library is fatal (there is
    no other implementation to fall back to)."""
    out = run_selftest(2, env_extra={"SIMDCOMP_HIP_LIB": "/nonexistent/libsimdcomp_hip.so"})
    assert out.returncode != 0
    assert "not found" in out.stderr


@pytest.mark.gpu
def test_two_ranks_share_one_gpu():
    """The N > 1 path of the benchmark itself (shards, per-rank legs, gathers, verification all-reduce) with both
    ranks on GPU 0 and the process group over gloo: a rehearsal for machines with one GPU, not a measurement."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["SIMDCOMP_BENCH_SHARE_GPU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--log2n", "22", "--steps", "2",
                          "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and "REHEARSAL" in doc["config"]["launch"]
    assert doc["config"]["ints_per_gpu"] == 1 << 22
    assert doc["verification"]["decode_mismatches"] == 0
    assert len(doc["roofline"]["per_rank_frac"]) == 2
    assert len(doc["encode"]["per_rank_value"]) == 2 and doc["encode"]["widths_with_wrong_bytes"] == 0
    # every leg an 8-GPU line carries has run with two ranks: frame of reference, the mixed-width d1 corpus cut by
    # simdcomp_hip_partition_by_bytes, host to host, and the CPU baseline on rank 0
    assert len(doc["for"]["per_rank"]) == 2 and doc["for"]["round_trip_mismatches"] == 0
    md = doc["mixed_d1"]
    assert len(md["per_rank"]) == 2 and md["decode_mismatches"] == 0
    assert md["shards_contiguous"] and md["widths_match_description"] and md["blocks_covered"] == md["total_blocks"]
    assert md["per_rank"][0]["blocks"] > md["per_rank"][1]["blocks"]  # equal bytes, not equal block counts
    assert md["imbalance"]["cost_bytes_max_over_mean"] < 1.001
    mixed = doc["host_to_host"]["per_width"]["mixed"]
    assert doc["host_to_host"]["ranks"] == 2 and mixed["round_trip_ok"] and mixed["staged_round_trip_ok"]
    assert doc["cpu_baseline"]["decode_vs_gpu_mismatches"] == 0 and doc["cpu_baseline"]["value"] > 0
