#!/usr/bin/env python3
"""Headline benchmark: device-resident simdunpackd1 over sorted posting lists.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): per GPU, a sorted list of 2^30 docIDs cut
into 128-integer blocks with stored per-block offsets, encoded with
simdpackwithoutmaskd1 at one uniform width b, for every b = 1..32.  One STEP
decodes the whole per-GPU corpus once at each of the 32 widths (32 launches of
simdcomp_hip_unpackd1, 2^30 integers each); inputs (payloads, offsets) and
outputs are resident in HBM.  value = decoded integers per second over all
GPUs.  Blocks are independent, so with N GPUs the global corpus of N*2^30
docIDs is split by block range (simdcomp_hip_partition) and no collective is on
the data path; the only communication is the final verification all-reduce
(mismatch count and checksum).

Extra objects on the JSON line:
  roofline      HBM roofline of the decode kernel: algorithmic bytes
                (512 + 16 b + 4 per block) / launch time from HIP events
                recorded on the launch stream inside the timed region,
                aggregated over the 32 widths, with the per-width table;
  cpu_baseline  the reference's own simdunpackd1 and simdpackwithoutmaskd1
                (compiled from its sources into oracle/_ref, SSE4.1) timed on
                this host over the first 2^26 docIDs of the SAME corpus the GPU
                decodes (payload and offsets copied back), 1 thread and every
                core this job may use; its output is compared with the GPU's;
  encode        simdpackwithoutmaskd1 over the same corpus (outside the timed
                steps), output compared byte for byte with the payload;
  callers       example.c's compress() and decode loop over the [width byte][payload]
                stream as one device call each, and a batched simdsearchwithlengthd1,
                on a mixed-width corpus; round trip and answers checked in place;
  for           BASELINE configs[3]: clustered integers through the frame-of-reference
                path (maxmin -> payload offsets -> pack_mixed / unpack_mixed), 2^30 per GPU;
  mixed_d1      BASELINE configs[4] device-resident: one mixed-width d1 corpus over all
                ranks, block-range shards from simdcomp_hip_partition_by_bytes, per-rank
                bytes and times, verification pair summed over the ranks;
  host_to_host  the same decode starting and ending in pinned host memory (H2D payload,
                kernel, D2H integers on three streams; and in place on the pinned buffers),
                at four uniform widths and on a mixed-width (1..20 bits) corpus.
roofline.traffic is measured in the same run at N = 1: after its own work rank 0
starts this script twice more as a child process under `rocprofv3 --pmc`
(FETCH_SIZE, WRITE_SIZE; one step each) and averages the counters of the 32 decode
launches (--no-traffic skips it; the committed summary of an earlier run,
profiles/pmc_traffic_latest.json, is the fallback when the profiler is missing).
`encode`, `for`, `mixed_d1` and `host_to_host` run on every rank at the same time
(each GPU has its own PCIe link; host memory is shared) and are reported per rank
and summed; cpu_baseline runs on rank 0 at every N (the other ranks wait without
spinning); callers and the counter passes run at N = 1 only.

Launching: under torch.distributed.run (RANK set) this process is one rank.
Started plainly with --gpus N > 1 it is only a launcher: it touches no GPU, starts
N child ranks of itself with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays
rank 0's JSON line and exits non-zero if any rank does.  --selftest runs the
launch, partition and verification all-reduce on CPU (gloo), without kernels.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # bytes/s, MI355X HBM3E nominal
WIDTHS = list(range(1, 33))

# set by load_package() in the ranks; the launcher never imports them
np = torch = dist = binding = codec = None


def load_package():
    """numpy, torch and the ctypes front end of libsimdcomp_hip.so (fails loudly if the library is missing)."""
    global np, torch, dist, binding, codec
    import numpy
    import torch as _torch
    import torch.distributed as _dist

    from simdcomp_amd import binding as b, codec as c

    np, torch, dist, binding, codec = numpy, _torch, _dist, b, c


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torchrun
# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(a, argv):
    """Start a.gpus child ranks of this script (one per GPU) and wait for them.  This process initialises nothing
    on the GPU.  Rank 0 prints the JSON line on our stdout; the other ranks' stdout goes to stderr."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), LOCAL_WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in live:  # a rank died: the others would wait in a barrier for ever
                    q.terminate()
        time.sleep(0.05)
    return rc


_REAL_STDOUT = None


def keep_stdout_for_the_line():
    """Libraries loaded below write to file descriptor 1 as well (RCCL prints a version banner when the process group
    comes up): from here on everything written to stdout goes to stderr, and only print_line() reaches the real
    stdout, so that rank 0 prints exactly one JSON line there."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def print_line(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def selftest(a, rank, world):
    """The multi-rank plumbing without a GPU: gloo process group, block-range shards from the library's partitioner,
    a host checksum per shard, and the verification all-reduce.  Rank 0 prints one JSON line."""
    from simdcomp_amd import synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    total_blocks = world * 1001
    first, nblocks = codec.partition(total_blocks, world, rank)
    words = synth.rand_u32(5, total_blocks * 128)[first * 128:(first + nblocks) * 128]
    idx = (np.arange(first * 128, (first + nblocks) * 128, dtype=np.uint64)) << np.uint64(32)
    chk = int(synth.splitmix64(idx ^ words.astype(np.uint64)).sum(dtype=np.uint64))
    # the mixed-width corpus description and its shards by bytes (what mixed_d1_leg runs on the GPUs)
    widths, prefix = corpus_description(total_blocks)
    bfirst, bcount = codec.partition_by_bytes(prefix, world, rank)
    cost = int(prefix[bfirst + bcount] - prefix[bfirst]) + 512 * bcount
    acc = torch.tensor([chk >> 32, chk & 0xFFFFFFFF, first, nblocks, bcount, cost, int(widths[bfirst:bfirst + bcount].sum())],
                       dtype=torch.int64)
    ends = torch.zeros(world * 2, dtype=torch.int64)
    ends[2 * rank], ends[2 * rank + 1] = bfirst, bfirst + bcount
    costs = torch.zeros(world, dtype=torch.int64)
    costs[rank] = cost
    dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    dist.all_reduce(ends, op=dist.ReduceOp.SUM)
    dist.all_reduce(costs, op=dist.ReduceOp.SUM)
    if rank == 0:
        got = acc.tolist()
        e = ends.tolist()
        print_line({"selftest": True, "backend": "gloo", "n_gpus": dist.get_world_size(), "blocks": got[3],
                    "sum_of_first": got[2], "checksum": ((got[0] << 32) + got[1]) & (2 ** 64 - 1),
                    "by_bytes": {"blocks": got[4], "cost_bytes": got[5], "sum_of_widths": got[6],
                                 "whole_cost_bytes": int(prefix[-1]) + 512 * total_blocks, "whole_sum_of_widths": int(widths.sum()),
                                 "contiguous": e[0] == 0 and e[-1] == total_blocks and all(e[2 * i + 1] == e[2 * i + 2] for i in range(world - 1)),
                                 "cost_per_rank": costs.tolist(), "blocks_per_rank": [e[2 * i + 1] - e[2 * i] for i in range(world)]}})
    dist.barrier()
    dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# corpus
# ---------------------------------------------------------------------------------------------------------------
def algorithmic_bytes(nblocks, b):
    """Bytes simdunpackd1 must move for nblocks blocks: payload read + output
    written + one 4-byte offset per block."""
    return nblocks * (512 + 16 * b + 4)


def make_ids(nblocks, b, seed, dev):
    """Sorted docIDs with random gaps below 2^b (the distribution of the
    reference's d1 benchmark generator, benchmarks/bitpackingbenchmark.c:44-53)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n = nblocks * 128
    gaps = torch.randint(0, 2 ** 31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
    if b < 31:
        gaps &= (1 << b) - 1
    elif b == 32:
        gaps = gaps * 2 + (gaps & 1)  # use all 32 bits
    return torch.cumsum(gaps, 0, dtype=torch.int32)  # wraps mod 2^32 like the reference's uint32 sums


def build_corpus(nblocks, b, seed, dev, payload_out=None, offsets_out=None):
    """Encode the docIDs of make_ids on the device.  Returns (payload, offsets,
    [checksum of the decoded array, number of decode mismatches])."""
    ids = make_ids(nblocks, b, seed, dev)
    offsets = torch.empty(nblocks, dtype=torch.int32, device=dev) if offsets_out is None else offsets_out
    payload = codec.packd1(ids, b, initvalue=0, masked=False, initvalues_out=offsets, out=payload_out)
    back = codec.unpackd1(offsets, payload, nblocks, b)
    acc = codec.checksum(back, other=ids)
    torch.cuda.synchronize()
    return payload, offsets, acc


def gather_floats(x, distributed, dev):
    """The value of every rank, in rank order."""
    if not distributed:
        return [float(x)]
    t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_rows(row, distributed, dev):
    """A short list of numbers from every rank, in rank order (floats are exact for integers below 2^53)."""
    if not distributed:
        return [[float(v) for v in row]]
    t = torch.tensor([float(v) for v in row], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(v) for v in o.tolist()] for o in out]


def timed_launches(fn, reps, warm=1):
    """Launch times of fn() from HIP events on the current stream: (average, best) in seconds over reps launches."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return sum(ts) / len(ts), min(ts)


def for_leg(dev, barrier, distributed, log2n, rank, reps=5):
    """BASELINE configs[3]: frame-of-reference coding of clustered integers, 2^log2n per GPU, on every rank at the
    same time.  The caller's sequence of tests/unit.c:659-666 per block -- frame = simdmin, width = bits(max - min) --
    as batched calls: simdcomp_hip_maxmin -> payload_offsets -> pack_mixed(FOR) -> unpack_mixed(FOR).  Round trip
    compared on the device.  Rates are per rank (HIP events around each launch) and summed."""
    n = 1 << log2n
    nblk = n // 128
    g = torch.Generator(device=dev)
    g.manual_seed(900 + rank)
    wid = torch.minimum(torch.randint(0, 25, (nblk,), device=dev, generator=g), torch.randint(0, 25, (nblk,), device=dev, generator=g))
    x = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev, generator=g)
    x &= ((1 << wid) - 1).to(torch.int32).repeat_interleave(128)
    base = torch.randint(-(1 << 31), (1 << 31) - 1, (nblk,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    x += base.repeat_interleave(128)  # cluster bases anywhere in 32 bits; the sums wrap like uint32
    del wid, base
    barrier()
    res = {}
    box = {}

    def frames():
        box["lo"], box["hi"], box["bits"] = codec.maxmin(x)

    t_mm, _ = timed_launches(frames, reps)
    lo, bits = box["lo"], box["bits"]
    del box["hi"]
    t_off, _ = timed_launches(lambda: box.__setitem__("off", codec.payload_offsets(bits)), reps)
    off, total = box["off"]
    nbytes = int(total.item())
    arena = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
    t_enc, _ = timed_launches(lambda: codec.pack_mixed(binding.FOR, x, bits, off, arena, initvalues=lo), reps)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    barrier()
    t_dec, t_dec_best = timed_launches(lambda: codec.unpack_mixed(binding.FOR, arena, bits, off, nblk, initvalues=lo, out=out), reps)
    acc = codec.checksum(out, other=x)
    torch.cuda.synchronize()
    bad = int(acc[1].item())
    moved = nbytes + nblk * (512 + 4 + 1 + 8)  # payload + integers + frame, width and offset of every block
    rows = gather_rows([n / t_dec, moved / t_dec / HBM_PEAK, n / t_enc, moved / t_enc / HBM_PEAK, n / t_mm, n * 4 / t_mm / HBM_PEAK,
                        bad, nbytes * 8 / n, t_dec * 1e3, t_enc * 1e3, t_mm * 1e3, t_off * 1e3, moved / t_dec_best / HBM_PEAK], distributed, dev)
    return {"what": "BASELINE configs[3]: clustered integers, 2^%d per GPU, per-block frame = simdmin and width = bits(max - min) "
                    "(tests/unit.c:659-666): simdcomp_hip_maxmin -> payload_offsets -> pack_mixed(FOR) -> unpack_mixed(FOR)" % log2n,
            "bytes_per_block": "512 + 16*b[k] + 4 (frame) + 1 (width) + 8 (offset)",
            "timing": f"average of {reps} launches per call, HIP events, every rank at the same time",
            "decode": {"aggregate_int32/s": sum(r[0] for r in rows), "min_frac": min(r[1] for r in rows)},
            "encode": {"aggregate_int32/s": sum(r[2] for r in rows), "min_frac": min(r[3] for r in rows)},
            "frames": {"aggregate_int32/s": sum(r[4] for r in rows), "min_frac": min(r[5] for r in rows)},
            "round_trip_mismatches": int(sum(r[6] for r in rows)),
            "per_rank": [{"decode_int32/s": r[0], "decode_frac": round(r[1], 4), "decode_frac_fastest_launch": round(r[12], 4),
                          "encode_int32/s": r[2], "encode_frac": round(r[3], 4),
                          "maxmin_int32/s": r[4], "maxmin_frac": round(r[5], 4), "round_trip_mismatches": int(r[6]),
                          "bits_per_int": round(r[7], 3), "decode_ms": round(r[8], 4), "encode_ms": round(r[9], 4),
                          "maxmin_ms": round(r[10], 4), "offsets_scan_ms": round(r[11], 4)} for r in rows]}


MIXED_SEED = 41


def corpus_description(total_blocks):
    """The global mixed-width corpus of BASELINE configs[4] as every rank can compute it by itself: the width of
    every block (counter-based, growing along the corpus: 1..8 bits at its start, 1..20 at its end) and the
    exclusive prefix of the payload bytes, total_blocks + 1 entries in host memory -- the array
    simdcomp_hip_partition_by_bytes cuts on."""
    from simdcomp_amd import synth

    widths = np.empty(total_blocks, dtype=np.uint8)
    step = 1 << 22  # in pieces: the generator's temporaries are 8-byte words, and eight ranks do this side by side
    for at in range(0, total_blocks, step):
        cnt = min(step, total_blocks - at)
        widths[at:at + cnt] = synth.ramped_widths(MIXED_SEED, cnt, 1, 8, 20, start=at, total=total_blocks)
    prefix = np.zeros(total_blocks + 1, dtype=np.uint64)
    np.cumsum(widths, dtype=np.uint64, out=prefix[1:])
    prefix *= np.uint64(16)
    return widths, prefix


def mixed_d1_leg(dev, barrier, distributed, log2n, rank, world, reps=5):
    """BASELINE configs[4], device-resident: ONE mixed-width d1 corpus of world * 2^log2n docIDs, sharded by block range.
    Every rank computes the corpus description (widths, payload prefix), takes its range from
    simdcomp_hip_partition_by_bytes (equal payload + output bytes, not equal block counts), builds the docIDs of its
    range on its GPU, encodes (simdmaxbitsd1 chained -> payload_offsets -> pack_mixed) and decodes (unpack_mixed) them.
    The widths the encoder finds must be the description's; the verification pair (decode mismatches, checksum with
    global indices) is summed over the ranks by the caller."""
    per = (1 << log2n) // 128
    total_blocks = world * per
    widths, prefix = corpus_description(total_blocks)
    first, count = codec.partition_by_bytes(prefix, world, rank)
    assert count > 0
    n = count * 128
    w = torch.from_numpy(widths[first:first + count].copy()).to(dev)
    del widths
    want_bytes = int(prefix[first + count] - prefix[first])
    g = torch.Generator(device=dev)
    g.manual_seed(7000 + first % 1000003)
    gaps = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev, generator=g)
    wi = w.to(torch.int32)
    gaps &= ((1 << wi) - 1).repeat_interleave(128)
    gaps.view(count, 128)[:, 0] |= 1 << (wi - 1)  # every block needs exactly its width
    start = (first * 2654435761) & 0x7FFFFFFF  # the docID in front of this range: any value, d1 coding is translation invariant
    ids = torch.cumsum(gaps, 0, dtype=torch.int32)
    ids += start
    del gaps, wi
    seeds = torch.empty(count, dtype=torch.int32, device=dev)
    barrier()
    box = {}
    t_bits, _ = timed_launches(lambda: box.__setitem__("bits", codec.maxbitsd1(ids, initvalue=start, initvalues_out=seeds)), reps)
    bits = box["bits"]
    widths_ok = bool(torch.equal(bits, w))
    del w
    off, total = codec.payload_offsets(bits)
    nbytes = int(total.item())
    arena = torch.empty(nbytes + 16, dtype=torch.uint8, device=dev)
    t_enc, _ = timed_launches(lambda: codec.pack_mixed(binding.D1, ids, bits, off, arena, initvalues=seeds, masked=False), reps)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    barrier()
    t0 = time.perf_counter()
    t_dec, t_dec_best = timed_launches(lambda: codec.unpack_mixed(binding.D1, arena, bits, off, count, initvalues=seeds, out=out), reps)
    barrier()
    wall = time.perf_counter() - t0
    acc = codec.checksum(out, other=ids, index0=first * 128)
    torch.cuda.synchronize()
    moved = nbytes + count * (512 + 4 + 1 + 8)
    rows = gather_rows([first, count, nbytes, t_dec * 1e3, moved / t_dec / HBM_PEAK, t_enc * 1e3, moved / t_enc / HBM_PEAK,
                        t_bits * 1e3, int(widths_ok and nbytes == want_bytes), moved / t_dec_best / HBM_PEAK], distributed, dev)
    if distributed:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM)
    ints = sum(r[1] for r in rows) * 128
    cost = [r[2] + r[1] * 512 for r in rows]
    return {"what": "BASELINE configs[4], device-resident: one d1 corpus of %d x 2^%d docIDs with per-block widths (1..8 bits at its "
                    "start, 1..20 at its end), block-range shards from simdcomp_hip_partition_by_bytes; per rank "
                    "simdcomp_hip_maxbitsd1 (chained) -> payload_offsets -> pack_mixed(D1) -> unpack_mixed(D1)" % (world, log2n),
            "bytes_per_block": "512 + 16*b[k] + 4 (offset) + 1 (width) + 8 (payload offset)",
            "timing": f"average of {reps} launches per call, HIP events, every rank at the same time",
            "total_blocks": total_blocks, "blocks_covered": int(sum(r[1] for r in rows)),
            "shards_contiguous": all(int(rows[i][0] + rows[i][1]) == int(rows[i + 1][0]) for i in range(len(rows) - 1)) and int(rows[0][0]) == 0,
            "widths_match_description": all(int(r[8]) == 1 for r in rows),
            "decode_mismatches": int(acc[1].item()), "checksum": int(acc[0].item()) & (2 ** 64 - 1),
            "decode": {"aggregate_int32/s": ints / max(r[3] for r in rows) * 1e3, "min_frac": min(r[4] for r in rows)},
            "encode": {"aggregate_int32/s": ints / max(r[5] for r in rows) * 1e3, "min_frac": min(r[6] for r in rows)},
            "imbalance": {"cost_bytes_max_over_mean": round(max(cost) / (sum(cost) / len(cost)), 5),
                          "blocks_max_over_mean": round(max(r[1] for r in rows) / (sum(r[1] for r in rows) / len(rows)), 5),
                          "decode_ms_max_over_mean": round(max(r[3] for r in rows) / (sum(r[3] for r in rows) / len(rows)), 4)},
            "wall_s_decode_launches_rank0": round(wall, 4),
            "per_rank": [{"first_block": int(r[0]), "blocks": int(r[1]), "payload_bytes": int(r[2]), "cost_bytes": int(r[2] + 512 * r[1]),
                          "bits_per_int": round(r[2] * 8 / (r[1] * 128), 3), "decode_ms": round(r[3], 4), "decode_frac": round(r[4], 4),
                          "decode_frac_fastest_launch": round(r[9], 4), "encode_ms": round(r[5], 4), "encode_frac": round(r[6], 4),
                          "maxbitsd1_ms": round(r[7], 4)} for r in rows]}


def encode_leg(corpora, nblocks, seed0, dev, reps=3):
    """simdpackwithoutmaskd1 over the same corpus at every width (outside the
    timed steps): regenerates the docIDs, times the encoder with HIP events and
    checks its output byte for byte against the payload the decoder was fed."""
    tot_bytes = tot_time = 0.0
    table, wrong = {}, 0
    for b in WIDTHS:
        payload, offsets = corpora[b]
        ids = make_ids(nblocks, b, seed0 + b, dev)
        out = torch.empty_like(payload)
        best = 1e9
        for _ in range(reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            codec.packd1(ids, b, initvalues=offsets, masked=False, out=out)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
        wrong += 0 if torch.equal(out, payload) else 1
        nbytes = algorithmic_bytes(nblocks, b)
        tot_bytes += nbytes
        tot_time += best
        table[str(b)] = {"ms": round(best * 1e3, 4), "frac": round(nbytes / best / HBM_PEAK, 4)}
        del ids, out
    return {"kernel": "sc::k_pack_fast<b, d1, unmasked>", "value": nblocks * 128 * len(WIDTHS) / tot_time, "unit": "int32/s per GPU",
            "achieved_GB/s": round(tot_bytes / tot_time / 1e9, 1), "frac": round(tot_bytes / tot_time / HBM_PEAK, 4),
            "min_frac": min(v["frac"] for v in table.values()), "widths_with_wrong_bytes": wrong,
            "timing": f"best of {reps} launches per width, HIP events", "per_width": table}


def ceilings(out):
    """Plain device-memory rates on this GPU for scale (not the roofline's peak): torch's device-to-device copy
    and fill over the 4 GiB output buffer, best of 5, HIP events."""
    half = out.numel() // 2
    res = {}
    for name, fn, nbytes in (("copy_d2d", lambda: out[:half].copy_(out[half:2 * half]), 8 * half),
                             ("fill", lambda: out.fill_(7), 4 * out.numel())):
        best = 1e9
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e-3)
        res[name + "_GB/s"] = round(nbytes / best / 1e9, 1)
    return res


def measured_traffic(nblocks):
    """Average HBM bytes per launch from the committed PMC summary (separate
    rocprofv3 --pmc passes of this same command, tools/summarize_profiles.py);
    None when the summary is absent or was taken at another corpus size."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")
    try:
        doc = json.load(open(path))
        rows = [doc["per_width"][str(b)] for b in WIDTHS]
        if any(abs(r["WRITE_SIZE_KB"] * 1024 - nblocks * 512) > nblocks for r in rows):
            return None
        return sum(r["hbm_bytes"] for r in rows) / len(rows)
    except Exception:
        return None


def live_traffic(log2n, nblocks, with_encode):
    """HBM bytes per launch of the decode kernel measured NOW: this same script is run twice more as a child
    process under `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE: the two do not fit one pass on gfx950), one step
    each, and the counters of the 32 decode launches are averaged.  Read bytes = 2 x FETCH_SIZE x 1024 (gfx950 tallies
    128-byte read requests as 64 bytes), written bytes = WRITE_SIZE x 1024.  Returns (bytes, per-width table) or
    (None, reason)."""
    import csv
    import glob
    import re
    import shutil
    import tempfile

    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return None, "rocprofv3 not found"
    pat = re.compile(r"k_unpack_fast<(\d+), 1, ")
    pat_enc = re.compile(r"k_pack_fast<(\d+), 1, false")
    per, enc = {}, {}
    tmp = tempfile.mkdtemp(prefix="simdcomp_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [tool, "--pmc", ctr, "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "1", "--warmup", "0", "--log2n", str(log2n), "--no-cpu", "--no-h2h", "--no-callers", "--no-traffic", "--no-for", "--no-mixed"]
            if not with_encode:
                cmd.append("--no-encode")
            child = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                     start_new_session=True)
            try:
                rc = child.wait(timeout=150)
            except subprocess.TimeoutExpired:
                import signal

                os.killpg(child.pid, signal.SIGKILL)  # the profiler and the run under it: the group started here
                child.wait()
                return None, f"rocprofv3 --pmc {ctr} timed out"
            if rc != 0:
                return None, f"rocprofv3 --pmc {ctr} exited with {rc}"
            files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {ctr} wrote no counter file"
            sums, esums = {}, {}
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] != ctr:
                    continue
                m = pat.search(row["Kernel_Name"])
                if m:
                    key = (int(m.group(1)), row["Dispatch_Id"])
                    sums[key] = sums.get(key, 0.0) + float(row["Counter_Value"])
                m = pat_enc.search(row["Kernel_Name"])
                if m:
                    key = (int(m.group(1)), row["Dispatch_Id"])
                    esums[key] = esums.get(key, 0.0) + float(row["Counter_Value"])
            for (b, _), v in sums.items():  # the last dispatch of a width is the timed step's
                per.setdefault(b, {})[ctr] = v
            for (b, _), v in esums.items():  # encoder launches of a width all move the same bytes
                enc.setdefault(b, {})[ctr] = v
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    table = {}
    for b in WIDTHS:
        if b not in per or len(per[b]) != 2:
            return None, f"no counters for width {b}"
        if abs(per[b]["WRITE_SIZE"] * 1024 - nblocks * 512) > nblocks * 8:
            return None, "written bytes do not match the corpus size"
        table[str(b)] = {"FETCH_SIZE_KB": per[b]["FETCH_SIZE"], "WRITE_SIZE_KB": per[b]["WRITE_SIZE"],
                         "hbm_bytes": int(2 * per[b]["FETCH_SIZE"] * 1024 + per[b]["WRITE_SIZE"] * 1024)}
    enc_table = {str(b): int(2 * enc[b]["FETCH_SIZE"] * 1024 + enc[b]["WRITE_SIZE"] * 1024) for b in WIDTHS
                 if b in enc and len(enc[b]) == 2}
    return sum(v["hbm_bytes"] for v in table.values()) / len(table), {"decode": table, "encode": enc_table}


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference itself on this host, over the same corpus
# ---------------------------------------------------------------------------------------------------------------
def usable_cores():
    """(cores this job may keep busy, how that was found).  os.cpu_count() is the whole machine; the scheduler
    affinity and the cgroup CPU quota say what this job gets."""
    nproc = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = nproc
    quota = None
    try:
        f = open("/sys/fs/cgroup/cpu.max").read().split()
        if f[0] != "max":
            quota = max(1, int(int(f[0]) / int(f[1]) + 0.5))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                quota = max(1, int(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5))
        except Exception:
            pass
    n = min(aff, quota) if quota else aff
    env = os.environ.get("OMP_NUM_THREADS")
    if quota is None and env and env.isdigit() and int(env) < n:
        n = int(env)  # no quota readable: the job's own thread budget
    return max(1, n), {"nproc": nproc, "affinity": aff, "cgroup_quota": quota, "OMP_NUM_THREADS": env}


def cpu_baseline(corpora, dev, log2n=26, reps=3):
    """The reference's simdunpackd1 / simdpackwithoutmaskd1 on this host's cores over the first 2^log2n docIDs of
    the corpus the GPU decodes (payload and per-block offsets copied back to host memory).  The checker side of the
    same pass: the reference's decode of that sample is compared with the GPU's, and the reference's encode of the
    decoded integers with the payload the GPU encoder produced."""
    try:
        import oracle
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "int32/s", "cores": 0, "kind": "none", "sample": f"oracle unavailable: {e}"}
    orc = oracle.Oracle()
    if oracle.have_ref():
        lib, kind = oracle.Reference(), "reference"
    else:
        lib, kind = orc, "port"
    n = 1 << log2n
    nblk = n // 128
    ncores, how = usable_cores()
    out = np.empty(n, dtype=np.uint32)
    t1 = tn = e1 = en = 0.0
    dec_bad = enc_bad = 0
    per_width = {}
    nb = nblk
    for b in WIDTHS:
        payload_d, offsets_d = corpora[b]
        nb = min(nblk, offsets_d.numel())
        payload = payload_d[: nb * 16 * b].cpu().numpy()
        seeds = offsets_d[:nb].cpu().numpy().view(np.uint32)
        d1 = orc.time_blocks(0, lib._unpackd1, seeds, payload, out, b, nb, 1, reps)
        dn = orc.time_blocks(0, lib._unpackd1, seeds, payload, out, b, nb, ncores, reps)
        gpu = codec.unpackd1(offsets_d[:nb], payload_d, nb, b).cpu().numpy().view(np.uint32)
        dec_bad += int((gpu != out[: nb * 128]).sum())
        repacked = np.zeros_like(payload)
        p1 = orc.time_blocks(2, lib._packwmd1, seeds, repacked, out, b, nb, 1, reps)
        pn = orc.time_blocks(2, lib._packwmd1, seeds, repacked, out, b, nb, ncores, reps)
        enc_bad += int((repacked != payload).sum())
        t1, tn, e1, en = t1 + d1, tn + dn, e1 + p1, en + pn
        per_width[str(b)] = {"decode_1": round(nb * 128 / d1 / 1e9, 2), "decode_all": round(nb * 128 / dn / 1e9, 2),
                             "encode_1": round(nb * 128 / p1 / 1e9, 2), "encode_all": round(nb * 128 / pn / 1e9, 2)}
        del gpu, payload, repacked
    total = nb * 128 * len(WIDTHS)
    avx2 = None
    if kind == "reference" and oracle.have_ref_avx():
        # the reference's fastest CPU path for scale: avxunpack (8-lane format, no differential coding)
        ra = oracle.ReferenceAvx()
        rng = np.random.default_rng(1)
        ta = 0.0
        for b in WIDTHS:
            payload = rng.integers(0, 256, nblk * 16 * b, dtype=np.uint8)
            ta += orc.time_blocks(4 + 1, ra.f[8][3], None, payload, out, b, nblk // 2, 1, reps)
        avx2 = {"value": n * len(WIDTHS) / ta, "unit": "int32/s", "cores": 1, "what": "avxunpack (plain, 256-integer blocks), random payload"}
    callers = cpu_callers(orc, lib)
    built = ("oracle/_ref built -O3 -msse4.1 (the reference's Makefile says -march=native; SSE4.1 so that one build "
             "runs on any host)") if kind == "reference" else "scalar restatement"
    return {"value": total / t1, "unit": "int32/s", "cores": 1, "kind": kind, "callers": callers,
            "stock_benchmark": stock_benchmark(), "cpu_model": cpu_model(),
            "sample": f"simdunpackd1 over the first 2^{log2n} docIDs of the GPU's own corpus (payload + offsets copied to the "
                      f"host, out of cache) at each b=1..32, best of {reps}, {built}",
            "all_cores": {"value": total / tn, "cores": ncores, "how_counted": how},
            "encode": {"what": "simdpackwithoutmaskd1 over the integers just decoded", "value": total / e1, "cores": 1,
                       "all_cores": {"value": total / en, "cores": ncores}},
            "decode_vs_gpu_mismatches": dec_bad, "encode_vs_gpu_payload_mismatches": enc_bad,
            "blocks_compared_per_width": nb, "per_width_Gint/s": per_width, "avx2_comparator": avx2}


def box_facts():
    """What sysfs says about the GPUs of this box (serial number, firmware, partition modes, clock tables): per-width
    rates come out in two populations across boxes, and this is what tells the boxes apart afterwards."""
    import glob

    facts = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("unique_id", "vbios_version", "current_compute_partition", "current_memory_partition", "pp_dpm_mclk",
                     "pp_dpm_sclk", "power_dpm_force_performance_level"):
            try:
                facts[os.path.basename(os.path.dirname(d)) + "/" + name] = open(os.path.join(d, name)).read().strip().replace("\n", " | ")
            except OSError:
                pass
    return facts


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def stock_benchmark():
    """The reference's own benchmarks/bitpackingbenchmark.c (one 128-integer block in cache, best of 500 calls, cycles
    per integer including its rdtsc/cpuid bracket), linked with the reference's own routines: oracle/_ref/
    bitpackingbenchmark_ref.  Mean over b=1..32 of each column; None when the binary was not built."""
    exe = os.path.join(ROOT, "oracle", "_ref", "bitpackingbenchmark_ref")
    if not os.path.exists(exe):
        return None
    try:
        text = subprocess.run([exe], capture_output=True, text=True, timeout=60).stdout
    except Exception as e:  # pragma: no cover
        return {"error": str(e)}
    res, name = {}, None
    for line in text.splitlines():
        if line.startswith("# --- "):
            name = line[6:].strip()
            res[name] = []
        elif name and line[:1].isdigit():
            f = line.split()
            if len(f) >= 3:
                res[name].append((float(f[1]), float(f[2])))
    return {k: {"pack_cycles_per_int": round(sum(p for p, _ in v) / len(v), 3),
                "unpack_cycles_per_int": round(sum(u for _, u in v) / len(v), 3), "widths": len(v)}
            for k, v in res.items() if v}


def cpu_callers(orc, block_codec, log2n=24):
    """The callers either side of the block codec on one host core, same bounded-sample idea: the decode loop of
    example.c:101-110 over a [width byte][payload] stream, and simdsearchwithlengthd1 once per (random block, key)."""
    from simdcomp_amd import synth

    nblk = (1 << log2n) // 128
    widths = synth.skewed_widths(11, nblk, 1, 20)
    gaps = synth.rand_u32(12, nblk * 128).reshape(nblk, 128)
    gaps &= ((np.uint64(1) << widths.astype(np.uint64)) - np.uint64(1)).astype(np.uint32)[:, None]
    ids = (np.cumsum(gaps.reshape(-1).astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    stream = orc.stream_compress(ids, d1=True, codec=block_codec)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        dec = orc.stream_decode(stream, nblk, d1=True, codec=block_codec)
        best = min(best, time.perf_counter() - t0)
    ok = bool(np.array_equal(dec, ids))
    n, bits, off, status = orc.stream_walk(stream, nblk)
    seeds = np.concatenate([[0], ids[127::128][:-1]]).astype(np.uint32)
    nq = 1 << 20
    qb = (synth.rand_u32(13, nq) % np.uint32(nblk)).astype(np.uint64)
    keys = (ids.reshape(nblk, 128)[qb.astype(np.int64), synth.rand_u32(14, nq) % np.uint32(128)] + np.uint32(1)).astype(np.uint32)
    pay = np.concatenate([stream, np.zeros(512, np.uint8)])
    ts, pos, val = orc.time_search(seeds, pay, off + np.arange(nblk, dtype=np.uint64) + np.uint64(1), bits, qb, keys, codec=block_codec)
    return {"stream_decode_int32/s": nblk * 128 / best, "stream_decode_ok": ok and status == 0 and n == nblk,
            "search_queries/s": nq / ts, "cores": 1,
            "sample": f"2^{log2n} docIDs, widths 1..20; 2^20 searches with a random block each"}


# ---------------------------------------------------------------------------------------------------------------
# callers on the device
# ---------------------------------------------------------------------------------------------------------------
def mixed_ids(n, dev, seed, lo=0, hi=20):
    """Sorted docIDs whose blocks need lo..hi bits for their gaps, skewed small (minimum of two uniform draws)."""
    nblk = n // 128
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    w = torch.minimum(torch.randint(lo, hi + 1, (nblk,), device=dev, generator=g), torch.randint(lo, hi + 1, (nblk,), device=dev, generator=g))
    gaps = torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev, generator=g)
    gaps &= ((1 << w) - 1).to(torch.int32).repeat_interleave(128)
    return torch.cumsum(gaps, 0, dtype=torch.int32), g


def callers_leg(dev, log2n):
    """The callers either side of the block codec, device-resident, on a mixed-width d1 corpus (widths 0..20 skewed
    small, mean ~7.5): compress() of example.c:56-75 and its decode loop :101-110 as one call each over the
    [width byte][payload] stream, and simdsearchwithlengthd1 as one batched launch.  The round trip and the search
    answers are checked here; the CPU counterparts are in cpu_baseline.callers."""
    L, P = binding.lib(), binding.ptr
    st = torch.cuda.current_stream().cuda_stream
    n = 1 << log2n
    nblk = n // 128
    ids, g = mixed_ids(n, dev, 77)

    def best(f, reps=5):
        f()
        torch.cuda.synchronize()
        t = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            e1.synchronize()
            t = min(t, e0.elapsed_time(e1) * 1e-3)
        return t

    buf = torch.empty(nblk * 513 + 16, dtype=torch.uint8, device=dev)
    nbytes = torch.zeros(1, dtype=torch.int64, device=dev)
    ws_c = torch.empty(max(16, L.simdcomp_hip_stream_compress_workspace(nblk)), dtype=torch.uint8, device=dev)
    t_c = best(lambda: binding.check(L.simdcomp_hip_stream_compress(1, 0, P(ids), nblk, P(buf), P(nbytes), P(ws_c), st), "compress"))
    T = int(nbytes.item())
    out = torch.empty(n, dtype=torch.int32, device=dev)
    res = torch.zeros(3, dtype=torch.int64, device=dev)
    ws_d = torch.empty(max(16, L.simdcomp_hip_stream_decode_workspace(T, nblk)), dtype=torch.uint8, device=dev)
    t_d = best(lambda: binding.check(L.simdcomp_hip_stream_decode(1, 0, P(buf), T, nblk, P(out), P(res), P(ws_d), st), "decode"))
    ok = bool(torch.equal(out, ids)) and res.tolist() == [nblk, T - nblk, 0]
    del out, ws_c, ws_d

    bits, off, _ = codec.stream_index(buf[:T], nblk)
    seeds = codec.chain_offsets(buf, bits, off, stream_layout=True)
    nq = 1 << 24
    qb = torch.randint(0, nblk, (nq,), dtype=torch.int64, device=dev, generator=g)
    slot = torch.randint(0, 128, (nq,), dtype=torch.int64, device=dev, generator=g)
    keys = ids[qb * 128 + slot]  # a key that is present: the answer is the first position holding it
    pos = val = None

    def run_search():
        nonlocal pos, val
        pos, val = codec.search(1, buf, keys, bits=bits, offsets=off, initvalues=seeds, query_blocks=qb, stream_layout=True)

    t_s = best(run_search, reps=3)
    nv = 1 << 20  # answers checked on the first 2^20 queries
    qb, keys, pos, val = qb[:nv], keys[:nv], pos[:nv], val[:nv]
    blockvals = ids.view(nblk, 128)[qb]
    # the first position whose value is >= key (unsigned): for sorted blocks that is the first occurrence of the key
    u = lambda x: x.to(torch.int64) & 0xFFFFFFFF
    first = (u(blockvals) >= u(keys)[:, None]).to(torch.int8).argmax(dim=1)
    sorted_block = (u(blockvals[:, 1:]) >= u(blockvals[:, :-1])).all(dim=1)
    search_ok = bool(torch.equal(pos.to(torch.int64)[sorted_block], first[sorted_block])) and \
        bool(torch.equal(val[sorted_block], keys[sorted_block])) and int(sorted_block.sum()) > nv // 2
    return {"corpus": f"2^{log2n} docIDs, widths 0..20, {T / nblk / 16:.2f} bits per integer, {T >> 20} MiB of stream",
            "stream_compress": {"ms": round(t_c * 1e3, 4), "int32/s": n / t_c},
            "stream_decode": {"ms": round(t_d * 1e3, 4), "int32/s": n / t_d},
            "round_trip_ok": ok,
            "search": {"queries": nq, "ms": round(t_s * 1e3, 4), "queries/s": nq / t_s, "answers_ok": search_ok}}


# ---------------------------------------------------------------------------------------------------------------
# host to host
# ---------------------------------------------------------------------------------------------------------------
class Pipeline:
    """Host-to-host pieces on three streams -- uploads, kernels, downloads -- chained by events, `depth` sets of device
    buffers.  With one stream per piece (upload, kernel, download in a row on it, two pieces alternating) each
    direction of the link idles half of the time: 7.0 against 10.9 G int32/s at 32 bits
    (profiles/r03_duplex_probe.json)."""

    def __init__(self, depth=3):
        self.depth = depth
        self.up, self.kernel, self.down = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()

    def run(self, pieces, upload, launch, download):
        kernel_done = [None] * self.depth
        down_done = [None] * self.depth
        for i, piece in enumerate(pieces):
            q = i % self.depth
            with torch.cuda.stream(self.up):
                if kernel_done[q] is not None:
                    self.up.wait_event(kernel_done[q])  # the kernel that read this slot's input
                upload(q, piece)
                up_done = self.up.record_event()
            with torch.cuda.stream(self.kernel):
                self.kernel.wait_event(up_done)
                if down_done[q] is not None:
                    self.kernel.wait_event(down_done[q])  # the download that read this slot's output
                launch(q, piece)
                kernel_done[q] = self.kernel.record_event()
            with torch.cuda.stream(self.down):
                self.down.wait_event(kernel_done[q])
                download(q, piece)
                down_done[q] = self.down.record_event()
        torch.cuda.synchronize()


def host_to_host(dev, barrier, distributed, log2n=28, widths=(1, 8, 16, 32)):
    """Decode that starts and ends in pinned host memory, two ways:
      staged    H2D payload (+ offsets, widths), kernel, D2H integers in 2^24-integer pieces through the three-stream
                Pipeline above;
      in_place  one launch whose kernel reads the payload from the pinned host buffer and writes the integers to the
                pinned host buffer itself (no copies, no device buffers).
    Every rank runs it on its own GPU at the same time (a barrier in front of every timed pass): `per_rank` lists
    each rank's rate, the headline figures their sum.  The `mixed` case is the per-block-width d1 corpus of BASELINE
    configs[4] (widths 1..20): widths, payload offsets and seeds travel with the payload, and
    simdcomp_hip_unpack_mixed decodes a piece."""
    n = 1 << log2n
    nblk = n // 128
    piece = 1 << 17  # blocks per piece (2^24 integers)
    depth = 3
    pipe = Pipeline(depth)
    h_out = torch.empty(n, dtype=torch.int32, pin_memory=True)
    L, P = binding.lib(), binding.ptr

    def timed(run):
        run()
        best = 1e9
        for _ in range(3):
            barrier()
            t0 = time.perf_counter()
            run()
            best = min(best, time.perf_counter() - t0)
        return best

    def row(t_staged, t_place, pcie, extra=None):
        """Gint/s is the in-place route (the one to use with pinned buffers: no copy engines, no device buffers, and
        no dependence on how the runtime maps streams to hardware queues, which moves the staged route by up to 40 %
        from process to process); staged_* is the three-stream copy pipeline."""
        rs = gather_floats(n / t_staged, distributed, dev)
        rp = gather_floats(n / t_place, distributed, dev)
        r = {"Gint/s": round(sum(rp) / 1e9, 2), "GB/s_over_pcie": round(sum(rp) / n * pcie / 1e9, 1),
             "per_rank_Gint/s": [round(v / 1e9, 2) for v in rp],
             "staged_Gint/s": round(sum(rs) / 1e9, 2), "staged_GB/s_over_pcie": round(sum(rs) / n * pcie / 1e9, 1),
             "staged_per_rank_Gint/s": [round(v / 1e9, 2) for v in rs]}
        r.update(extra or {})
        return r

    res = {}
    h_off = torch.empty(nblk, dtype=torch.int32, pin_memory=True).fill_(12345)
    d_out = [torch.empty(piece * 128, dtype=torch.int32, device=dev) for _ in range(depth)]
    firsts = list(range(0, nblk, piece))
    for b in widths:
        h_pay = torch.empty(nblk * 16 * b, dtype=torch.uint8, pin_memory=True).fill_(0x5B)
        d_pay = [torch.empty(piece * 16 * b, dtype=torch.uint8, device=dev) for _ in range(depth)]
        d_off = [torch.empty(piece, dtype=torch.int32, device=dev) for _ in range(depth)]

        def upload(q, first):
            cnt = min(piece, nblk - first)
            d_pay[q][: cnt * 16 * b].copy_(h_pay[first * 16 * b:(first + cnt) * 16 * b], non_blocking=True)
            d_off[q][:cnt].copy_(h_off[first:first + cnt], non_blocking=True)

        def launch(q, first):
            codec.unpackd1(d_off[q], d_pay[q], min(piece, nblk - first), b, out=d_out[q])

        def download(q, first):
            cnt = min(piece, nblk - first)
            h_out[first * 128:(first + cnt) * 128].copy_(d_out[q][: cnt * 128], non_blocking=True)

        t_staged = timed(lambda: pipe.run(firsts, upload, launch, download))
        staged_sum = int(h_out.sum(dtype=torch.int64).item())

        def in_place():
            binding.check(L.simdcomp_hip_unpackd1(P(h_off), P(h_pay), P(h_out), nblk, b, torch.cuda.current_stream().cuda_stream), "unpackd1")
            torch.cuda.synchronize()

        h_out.zero_()
        t_place = timed(in_place)
        same = int(h_out.sum(dtype=torch.int64).item()) == staged_sum
        res[str(b)] = row(t_staged, t_place, n * 4 + nblk * (16 * b + 4), {"staged_equals_in_place": same})
        del h_pay, d_pay, d_off

    # mixed widths: build the corpus on the device, keep it in pinned host memory the way an index file would be
    ids, _ = mixed_ids(n, dev, 79, 1, 20)
    seeds_d = torch.empty(nblk, dtype=torch.int32, device=dev)
    bits_d = codec.maxbitsd1(ids, initvalue=0, initvalues_out=seeds_d)
    off_d, total_d = codec.payload_offsets(bits_d)
    total = int(total_d.item())
    arena = torch.empty(total + 16, dtype=torch.uint8, device=dev)
    codec.pack_mixed(1, ids, bits_d, off_d, arena, initvalues=seeds_d, masked=False)
    h_ids = ids.cpu()
    h_pay = torch.empty(total + 16, dtype=torch.uint8, pin_memory=True).copy_(arena)
    h_bits = torch.empty(nblk, dtype=torch.uint8, pin_memory=True).copy_(bits_d)
    h_offs = torch.empty(nblk, dtype=torch.int64, pin_memory=True).copy_(off_d)
    h_seeds = torch.empty(nblk, dtype=torch.int32, pin_memory=True).copy_(seeds_d)
    del ids, arena, bits_d, off_d, seeds_d
    lo = {f: int(h_offs[f]) for f in firsts}
    hi = {f: (int(h_offs[f + piece]) if f + piece < nblk else total) for f in firsts}
    cap = max(hi[f] - lo[f] for f in firsts)
    d_pay = [torch.empty(cap + 16, dtype=torch.uint8, device=dev) for _ in range(depth)]
    d_bits = [torch.empty(piece, dtype=torch.uint8, device=dev) for _ in range(depth)]
    d_offs = [torch.empty(piece, dtype=torch.int64, device=dev) for _ in range(depth)]
    d_seeds = [torch.empty(piece, dtype=torch.int32, device=dev) for _ in range(depth)]

    def upload_mixed(q, first):
        cnt = min(piece, nblk - first)
        d_pay[q][: hi[first] - lo[first]].copy_(h_pay[lo[first]:hi[first]], non_blocking=True)
        d_bits[q][:cnt].copy_(h_bits[first:first + cnt], non_blocking=True)
        d_offs[q][:cnt].copy_(h_offs[first:first + cnt], non_blocking=True)
        d_seeds[q][:cnt].copy_(h_seeds[first:first + cnt], non_blocking=True)

    def launch_mixed(q, first):
        # the offsets are those of the whole arena: the piece's payload pointer is moved back by the offset of its
        # first block
        binding.check(L.simdcomp_hip_unpack_mixed(1, P(d_seeds[q]), d_pay[q].data_ptr() - lo[first], P(d_offs[q]), P(d_bits[q]),
                                                  P(d_out[q]), min(piece, nblk - first), torch.cuda.current_stream().cuda_stream), "unpack_mixed")

    def download_mixed(q, first):
        cnt = min(piece, nblk - first)
        h_out[first * 128:(first + cnt) * 128].copy_(d_out[q][: cnt * 128], non_blocking=True)

    h_out.zero_()
    t_staged = timed(lambda: pipe.run(firsts, upload_mixed, launch_mixed, download_mixed))
    ok = bool(torch.equal(h_out, h_ids))

    def in_place_mixed():
        binding.check(L.simdcomp_hip_unpack_mixed(1, P(h_seeds), P(h_pay), P(h_offs), P(h_bits), P(h_out), nblk,
                                                  torch.cuda.current_stream().cuda_stream), "unpack_mixed")
        torch.cuda.synchronize()

    h_out.zero_()
    t_place = timed(in_place_mixed)
    ok_place = bool(torch.equal(h_out, h_ids))
    res["mixed"] = row(t_staged, t_place, n * 4 + total + nblk * 13,
                       {"bits_per_int": round(total * 8 / n, 2), "round_trip_ok": ok_place, "staged_round_trip_ok": ok,
                        "what": "d1, per-block widths 1..20 (BASELINE configs[4] shape), simdcomp_hip_unpack_mixed"})
    return {"ints_per_rank": n, "pinned": True, "ranks": len(res["mixed"]["per_rank_Gint/s"]),
            "staged": "three streams (uploads, kernels, downloads) chained by events, 3 sets of device buffers, 2^24-integer pieces",
            "in_place": "one launch on the pinned host buffers themselves (the kernel's loads and stores cross the host link)",
            "per_width": res}


# ---------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=30, help="log2 of docIDs per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-h2h", action="store_true", help="skip the host-to-host leg")
    ap.add_argument("--no-encode", action="store_true", help="skip the encoder leg")
    ap.add_argument("--no-callers", action="store_true", help="skip the stream / search leg")
    ap.add_argument("--no-for", action="store_true", help="skip the frame-of-reference leg (configs[3])")
    ap.add_argument("--no-mixed", action="store_true", help="skip the mixed-width d1 leg (configs[4], device-resident)")
    ap.add_argument("--no-traffic", action="store_true", help="do not re-run under rocprofv3 --pmc for roofline.traffic")
    ap.add_argument("--selftest", action="store_true", help="CPU-only check of the multi-rank plumbing (gloo)")
    a = ap.parse_args()

    if "RANK" not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher (nothing below runs in this process)
        sys.exit(launch_ranks(a, sys.argv[1:]))

    keep_stdout_for_the_line()
    load_package()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE is {world}")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if a.selftest:
        return selftest(a, rank, world)

    # SIMDCOMP_BENCH_SHARE_GPU=1: every rank uses GPU 0 and the process group runs over gloo (RCCL refuses two ranks
    # on one device).  Only for rehearsing the N > 1 code path on a one-GPU machine; the line says so.
    share_gpu = os.environ.get("SIMDCOMP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the process group is always created, also for one rank, so that the
    # same code path runs at every N
    distributed = world > 1 or "RANK" in os.environ
    if distributed:
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
        wait_group = dist.new_group(backend="gloo")  # for waits that must not spin (see cpu_baseline below)

    # block-range shard of the global corpus (weak scaling: 2^log2n docIDs per GPU)
    total_blocks = world * ((1 << a.log2n) // 128)
    first, nblocks = codec.partition(total_blocks, world, rank)

    corpora = {}
    verify = torch.zeros(2, dtype=torch.int64, device=dev)
    seed0 = 1000 * (first % 100003)
    # The working set of the decode (payloads, offsets, output) lives in ONE allocation, cut into 2 MiB aligned slices.
    # With an allocation per array the widths 8..15 and 20..29 (and b = 1) ran 3-5 points slower in most processes --
    # which ones changed from process to process -- although the kernels fetch the same bytes: 79.3-80.0 % aggregate
    # in six runs on four boxes against 80.5-81.2 % with one allocation (profiles/r02_arena_ab.log).
    # SIMDCOMP_BENCH_ARENA=0 brings the separate allocations back, 1 leaves only the output outside, 3 puts the slices
    # off the 2 MiB grid (for comparison runs).
    arena = None
    arena_mode = int(os.environ.get("SIMDCOMP_BENCH_ARENA", "2"))
    if arena_mode:
        two_mib = 2 << 20
        # (SIMDCOMP_BENCH_ALIGN=a, comparison runs: every slice at an odd multiple of a instead of on the 2 MiB grid)
        odd = int(os.environ.get("SIMDCOMP_BENCH_ALIGN", "0"))
        skew = 4608 if arena_mode == 3 else 0
        up = lambda x: (x + two_mib - 1) // two_mib * two_mib + skew
        total = sum(up(nblocks * 16 * b) + up(nblocks * 4) for b in WIDTHS) + up(nblocks * 512) + 2 * two_mib
        arena = torch.empty(total, dtype=torch.uint8, device=dev)
        at = (-arena.data_ptr()) % two_mib + skew + odd
    for b in WIDTHS:
        if arena is not None:
            pay = arena[at:at + nblocks * 16 * b]
            at += up(nblocks * 16 * b)
            offs = arena[at:at + nblocks * 4].view(torch.int32)
            at += up(nblocks * 4)
            payload, offsets, acc = build_corpus(nblocks, b, seed=seed0 + b, dev=dev, payload_out=pay, offsets_out=offs)
        else:
            payload, offsets, acc = build_corpus(nblocks, b, seed=seed0 + b, dev=dev)
        corpora[b] = (payload, offsets)
        verify += acc
    if arena_mode == 2:
        out = arena[at:at + nblocks * 512].view(torch.int32)
    else:
        out = torch.empty(nblocks * 128, dtype=torch.int32, device=dev)

    def step(events=None):
        for b in WIDTHS:
            payload, offsets = corpora[b]
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            codec.unpackd1(offsets, payload, nblocks, b, out=out)
            if events is not None:
                e1.record()
                events.append((b, e0, e1))

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    events = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(events)
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(verify, op=dist.ReduceOp.SUM)  # the verification reduction (RCCL)
    elapsed = float(t.item())

    # per-width kernel time from the events recorded in the timed region
    per_b = {b: [] for b in WIDTHS}
    for b, e0, e1 in events:
        per_b[b].append(e0.elapsed_time(e1) * 1e-3)
    table = {}
    tot_bytes = tot_time = 0.0
    for b in WIDTHS:
        avg = sum(per_b[b]) / len(per_b[b])
        nbytes = algorithmic_bytes(nblocks, b)
        tot_bytes += nbytes
        tot_time += avg
        table[str(b)] = {"ms": round(avg * 1e3, 4), "GB/s": round(nbytes / avg / 1e9, 1),
                         "frac": round(nbytes / avg / HBM_PEAK, 4), "Gint/s": round(nblocks * 128 / avg / 1e9, 1),
                         "frac_fastest_launch": round(nbytes / min(per_b[b]) / HBM_PEAK, 4)}
    rank_fracs = gather_floats(tot_bytes / tot_time / HBM_PEAK, distributed, dev)

    line = None
    if rank == 0:
        ints_per_step = world * nblocks * 128 * len(WIDTHS)
        achieved = tot_bytes / tot_time
        if "TORCHELASTIC_RUN_ID" in os.environ:
            how = "torch.distributed.run"
        else:
            how = "bench.py launcher, one child process per GPU" if distributed else "single process"
        if share_gpu:
            how += "; REHEARSAL: all ranks share GPU 0, process group over gloo (not a scaling measurement)"
        line = {
            "metric": "device-resident decoded int32/s (simdunpackd1, b=1..32)",
            "value": ints_per_step * a.steps / elapsed,
            "unit": "int32/s",
            "n_gpus": dist.get_world_size() if distributed else 1,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: sorted posting lists, 2^%d docIDs per GPU, random gaps < 2^b, "
                                   "simdunpackd1 with stored per-block offsets, one launch per width b=1..32 per step"
                                   % a.log2n,
                       "ints_per_gpu": nblocks * 128, "widths": "1..32", "sharding": "block range, no data-path collective",
                       "allocation": {0: "one allocation per array", 1: "payloads and offsets in one allocation",
                                      2: "payloads, offsets and output in one allocation (2 MiB aligned slices)",
                                      3: "one allocation, slices off the 2 MiB grid"}.get(arena_mode, str(arena_mode)),
                       "launch": how},
            "verification": {"decode_mismatches": int(verify[1].item()), "checksum": int(verify[0].item()) & (2 ** 64 - 1)},
            "roofline": {"bound": "hbm", "achieved": round(achieved / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK, 4), "traffic": measured_traffic(nblocks),
                         "algorithmic_bytes_per_launch": tot_bytes / len(WIDTHS),
                         "traffic_source": "profiles/pmc_traffic_latest.json: an earlier rocprofv3 --pmc run of this same "
                                           "command (separate FETCH_SIZE / WRITE_SIZE passes), not measured in this run",
                         "traffic_note": "L2<->fabric bytes (reads = 2 x FETCH_SIZE on gfx950); widths 1..7 fetch their "
                                         "payload twice at this level (prefetch workgroup + decode workgroup, second copy "
                                         "from the Infinity Cache), widths >= 8 are within 4 % of algorithmic",
                         "kernel": "sc::k_unpack_fast<b, d1> (one instantiation per width), aggregate over b=1..32",
                         "bytes_per_block": "512 + 16*b + 4",
                         "min_frac": min(v["frac"] for v in table.values()),
                         "min_frac_fastest_launch": min(v["frac_fastest_launch"] for v in table.values()),
                         "spread_note": "frac is over the average of the timed launches of a width; single widths move by "
                                        "+-3 points between runs on one box (three runs in a row: b = 8 at 82 / 77 / 83 %), "
                                        "the aggregate by under one",
                         "per_rank_frac": [round(f, 4) for f in rank_fracs],
                         "per_width": table},
        }
        line["roofline"]["measured_ceilings"] = ceilings(out)
        line["box"] = box_facts()
    del out
    if not a.no_encode:
        barrier()
        enc = encode_leg(corpora, nblocks, seed0, dev)
        vals = gather_floats(enc["value"], distributed, dev)
        wrong = gather_floats(enc["widths_with_wrong_bytes"], distributed, dev)
        if rank == 0:
            enc["per_rank_value"] = vals
            enc["aggregate_value"] = sum(vals)
            enc["aggregate_unit"] = "int32/s over all GPUs"
            enc["widths_with_wrong_bytes"] = int(sum(wrong))
            line["encode"] = enc
    # cpu_baseline: rank 0 only, outside every timed region.  At N > 1 the other ranks wait for it in a barrier of a
    # gloo group (a blocked socket read: they keep no host core busy while the reference is being timed)
    if not a.no_cpu:
        if rank == 0:
            cb = cpu_baseline(corpora, dev)
            line["cpu_baseline"] = cb
            line["verification"]["sampled_blocks_vs_oracle_mismatches"] = cb.get("decode_vs_gpu_mismatches")
            line["verification"]["sampled_blocks_per_width"] = cb.get("blocks_compared_per_width")
        if distributed:
            dist.barrier(group=wait_group)
    del corpora
    arena = None
    torch.cuda.empty_cache()
    if not a.no_for:
        fl = for_leg(dev, barrier, distributed, a.log2n, rank)
        if rank == 0:
            line["for"] = fl
        torch.cuda.empty_cache()
    if not a.no_mixed:
        ml = mixed_d1_leg(dev, barrier, distributed, a.log2n, rank, world)
        if rank == 0:
            line["mixed_d1"] = ml
        torch.cuda.empty_cache()
    if rank == 0 and not a.no_callers and world == 1:
        line["callers"] = callers_leg(dev, a.log2n)
        torch.cuda.empty_cache()
    if rank == 0 and not a.no_traffic and world == 1:
        # HBM-side traffic of the decode kernel from the hardware counters, measured in this run (two child runs of
        # this script under rocprofv3 --pmc); the committed summary of an earlier run stays as the fallback
        live, detail = live_traffic(a.log2n, nblocks, not a.no_encode)
        if live is not None:
            line["roofline"]["traffic"] = live
            line["roofline"]["traffic_source"] = ("measured in this run: this script re-run twice as a child under rocprofv3 --pmc "
                                                  "(FETCH_SIZE, WRITE_SIZE; one step each), average over the 32 decode launches")
            line["roofline"]["traffic_over_algorithmic"] = round(live / line["roofline"]["algorithmic_bytes_per_launch"], 4)
            line["roofline"]["traffic_per_width"] = {b: v["hbm_bytes"] for b, v in detail["decode"].items()}
            if "encode" in line and len(detail["encode"]) == len(WIDTHS):
                et = detail["encode"]
                line["encode"]["traffic"] = sum(et.values()) / len(et)
                line["encode"]["traffic_over_algorithmic"] = round(sum(et.values()) / sum(algorithmic_bytes(nblocks, b) for b in WIDTHS), 4)
                line["encode"]["traffic_per_width"] = et
        else:
            line["roofline"]["traffic_live_failed"] = detail
    if not a.no_h2h:
        h2h = host_to_host(dev, barrier, distributed, log2n=min(28, a.log2n))
        if rank == 0:
            line["host_to_host"] = h2h
    if rank == 0:
        print_line(line)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
