#!/usr/bin/env python3
"""What the host link of this GPU can carry, and which way of driving it gets there.

  copies   pinned 1 GiB host -> device alone, device -> host alone, both at once on two streams, and each
           direction split over two streams (does a second copy engine add anything?);
  kernels  simdunpackd1 reading its payload straight from pinned host memory (no copy in front), writing its
           integers straight to pinned host memory (no copy behind), and both;
  mixes    the shapes a host-to-host decode can take: copy up + kernel + copy down on two and on four streams,
           kernel reads host + copy down, copy up + kernel writes host, kernel does both.

Prints one JSON object.  Rates are GB/s of the bytes named, best of 3."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simdcomp_amd import binding, codec  # noqa: E402

GIB = 1 << 30


def best(run, reps=3):
    run()
    torch.cuda.synchronize()
    t = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        t = min(t, time.perf_counter() - t0)
    return t


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    L, P = binding.lib(), binding.ptr
    res = {"copies": {}, "kernels": {}, "mixes": {}}
    h_a = torch.empty(GIB, dtype=torch.uint8, pin_memory=True).fill_(0x5B)
    h_b = torch.empty(GIB, dtype=torch.uint8, pin_memory=True).fill_(1)
    d_a = torch.empty(GIB, dtype=torch.uint8, device=dev)
    d_b = torch.empty(GIB, dtype=torch.uint8, device=dev).fill_(2)
    s = [torch.cuda.Stream() for _ in range(4)]

    def on(i, fn):
        with torch.cuda.stream(s[i]):
            fn()

    half = GIB // 2
    c = res["copies"]
    c["h2d_alone_GB/s"] = round(GIB / best(lambda: on(0, lambda: d_a.copy_(h_a, non_blocking=True))) / 1e9, 2)
    c["d2h_alone_GB/s"] = round(GIB / best(lambda: on(1, lambda: h_b.copy_(d_b, non_blocking=True))) / 1e9, 2)

    def both():
        on(0, lambda: d_a.copy_(h_a, non_blocking=True))
        on(1, lambda: h_b.copy_(d_b, non_blocking=True))

    t = best(both)
    c["both_at_once_sum_GB/s"] = round(2 * GIB / t / 1e9, 2)

    def h2d_two():
        on(0, lambda: d_a[:half].copy_(h_a[:half], non_blocking=True))
        on(1, lambda: d_a[half:].copy_(h_a[half:], non_blocking=True))

    c["h2d_two_streams_GB/s"] = round(GIB / best(h2d_two) / 1e9, 2)

    def d2h_two():
        on(0, lambda: h_b[:half].copy_(d_b[:half], non_blocking=True))
        on(1, lambda: h_b[half:].copy_(d_b[half:], non_blocking=True))

    c["d2h_two_streams_GB/s"] = round(GIB / best(d2h_two) / 1e9, 2)

    def four():
        on(0, lambda: d_a[:half].copy_(h_a[:half], non_blocking=True))
        on(1, lambda: d_a[half:].copy_(h_a[half:], non_blocking=True))
        on(2, lambda: h_b[:half].copy_(d_b[:half], non_blocking=True))
        on(3, lambda: h_b[half:].copy_(d_b[half:], non_blocking=True))

    c["both_two_streams_each_sum_GB/s"] = round(2 * GIB / best(four) / 1e9, 2)

    # pieces of 64 MiB alternating directions on two streams, the shape of the host API's pipeline
    piece = 64 << 20

    def pieces(nstreams):
        def run():
            for i, at in enumerate(range(0, GIB, piece)):
                k = i % nstreams

                def f():
                    d_a[at:at + piece].copy_(h_a[at:at + piece], non_blocking=True)
                    h_b[at:at + piece].copy_(d_b[at:at + piece], non_blocking=True)

                on(k, f)
        return run

    for ns in (2, 4):
        c[f"pieces_64MiB_up_then_down_{ns}_streams_sum_GB/s"] = round(2 * GIB / best(pieces(ns)) / 1e9, 2)

    # dedicated streams: all uploads on stream 0, all downloads on stream 1 (no event chaining: raw link only)
    def dedicated():
        for at in range(0, GIB, piece):
            on(0, lambda: d_a[at:at + piece].copy_(h_a[at:at + piece], non_blocking=True))
            on(1, lambda: h_b[at:at + piece].copy_(d_b[at:at + piece], non_blocking=True))

    c["pieces_64MiB_dedicated_up_and_down_streams_sum_GB/s"] = round(2 * GIB / best(dedicated) / 1e9, 2)

    # ---- kernels on host memory -------------------------------------------------------------------------------
    k = res["kernels"]
    n = 1 << 27  # integers (512 MiB out)
    nblk = n // 128
    h_out = torch.empty(n, dtype=torch.int32, pin_memory=True)
    d_out = torch.empty(n, dtype=torch.int32, device=dev)
    h_seed = torch.empty(nblk, dtype=torch.int32, pin_memory=True).fill_(7)
    d_seed = h_seed.to(dev)
    for b in (8, 32):
        nb = nblk * 16 * b
        h_pay = h_a[:nb]
        d_pay = d_a[:nb]
        d_pay.copy_(h_pay)

        def unpack(pay, seed, out, st=None):
            binding.check(L.simdcomp_hip_unpackd1(P(seed), P(pay), P(out), nblk, b,
                                                  st if st is not None else torch.cuda.current_stream().cuda_stream), "unpackd1")

        bytes_up, bytes_down = nb + nblk * 4, n * 4
        row = {}
        row["device_to_device_ms"] = round(best(lambda: unpack(d_pay, d_seed, d_out)) * 1e3, 3)
        t = best(lambda: unpack(h_pay, h_seed, d_out))
        row["payload_read_from_host_GB/s"] = round(bytes_up / t / 1e9, 2)
        t = best(lambda: unpack(d_pay, d_seed, h_out))
        row["integers_written_to_host_GB/s"] = round(bytes_down / t / 1e9, 2)
        t = best(lambda: unpack(h_pay, h_seed, h_out))
        row["both_through_the_kernel_Gint/s"] = round(n / t / 1e9, 2)
        row["both_through_the_kernel_sum_GB/s"] = round((bytes_up + bytes_down) / t / 1e9, 2)
        assert torch.equal(h_out, d_out.cpu())
        k[str(b)] = row

        # ---- the shapes of a host-to-host decode, 2^24-integer pieces ------------------------------------------
        pb = 1 << 17  # blocks per piece
        m = {}

        def classic(ns):
            dp = [torch.empty(pb * 16 * b, dtype=torch.uint8, device=dev) for _ in range(ns)]
            ds = [torch.empty(pb, dtype=torch.int32, device=dev) for _ in range(ns)]
            do = [torch.empty(pb * 128, dtype=torch.int32, device=dev) for _ in range(ns)]

            def run():
                for i, first in enumerate(range(0, nblk, pb)):
                    q = i % ns
                    with torch.cuda.stream(s[q]):
                        dp[q].copy_(h_pay[first * 16 * b:(first + pb) * 16 * b], non_blocking=True)
                        ds[q].copy_(h_seed[first:first + pb], non_blocking=True)
                        binding.check(L.simdcomp_hip_unpackd1(P(ds[q]), P(dp[q]), P(do[q]), pb, b, s[q].cuda_stream), "u")
                        h_out[first * 128:(first + pb) * 128].copy_(do[q], non_blocking=True)
            return run

        for ns in (2, 3, 4):
            m[f"copy_up_kernel_copy_down_{ns}_streams_Gint/s"] = round(n / best(classic(ns)) / 1e9, 2)

        def dedicated_pipeline(depth):
            """uploads on stream 0, kernels on stream 1, downloads on stream 2, chained by events; `depth` buffers"""
            dp = [torch.empty(pb * 16 * b, dtype=torch.uint8, device=dev) for _ in range(depth)]
            ds = [torch.empty(pb, dtype=torch.int32, device=dev) for _ in range(depth)]
            do = [torch.empty(pb * 128, dtype=torch.int32, device=dev) for _ in range(depth)]

            def run():
                up_done = [None] * depth
                k_done = [None] * depth
                down_done = [None] * depth
                for i, first in enumerate(range(0, nblk, pb)):
                    q = i % depth
                    with torch.cuda.stream(s[0]):
                        if k_done[q] is not None:
                            s[0].wait_event(k_done[q])  # the kernel that read this input buffer
                        dp[q].copy_(h_pay[first * 16 * b:(first + pb) * 16 * b], non_blocking=True)
                        ds[q].copy_(h_seed[first:first + pb], non_blocking=True)
                        up_done[q] = s[0].record_event()
                    with torch.cuda.stream(s[1]):
                        s[1].wait_event(up_done[q])
                        if down_done[q] is not None:
                            s[1].wait_event(down_done[q])  # the download that read this output buffer
                        binding.check(L.simdcomp_hip_unpackd1(P(ds[q]), P(dp[q]), P(do[q]), pb, b, s[1].cuda_stream), "u")
                        k_done[q] = s[1].record_event()
                    with torch.cuda.stream(s[2]):
                        s[2].wait_event(k_done[q])
                        h_out[first * 128:(first + pb) * 128].copy_(do[q], non_blocking=True)
                        down_done[q] = s[2].record_event()
            return run

        for depth in (2, 3, 4):
            m[f"dedicated_up_kernel_down_streams_{depth}_buffers_Gint/s"] = round(n / best(dedicated_pipeline(depth)) / 1e9, 2)

        def read_host_copy_down(ns):
            do = [torch.empty(pb * 128, dtype=torch.int32, device=dev) for _ in range(ns)]

            def run():
                for i, first in enumerate(range(0, nblk, pb)):
                    q = i % ns
                    with torch.cuda.stream(s[q]):
                        binding.check(L.simdcomp_hip_unpackd1(h_seed.data_ptr() + 4 * first, h_pay.data_ptr() + first * 16 * b,
                                                              P(do[q]), pb, b, s[q].cuda_stream), "u")
                        h_out[first * 128:(first + pb) * 128].copy_(do[q], non_blocking=True)
            return run

        for ns in (2, 3):
            m[f"kernel_reads_host_copy_down_{ns}_streams_Gint/s"] = round(n / best(read_host_copy_down(ns)) / 1e9, 2)

        def copy_up_write_host(ns):
            dp = [torch.empty(pb * 16 * b, dtype=torch.uint8, device=dev) for _ in range(ns)]
            ds = [torch.empty(pb, dtype=torch.int32, device=dev) for _ in range(ns)]

            def run():
                for i, first in enumerate(range(0, nblk, pb)):
                    q = i % ns
                    with torch.cuda.stream(s[q]):
                        dp[q].copy_(h_pay[first * 16 * b:(first + pb) * 16 * b], non_blocking=True)
                        ds[q].copy_(h_seed[first:first + pb], non_blocking=True)
                        binding.check(L.simdcomp_hip_unpackd1(P(ds[q]), P(dp[q]), h_out.data_ptr() + 512 * first, pb, b,
                                                              s[q].cuda_stream), "u")
            return run

        for ns in (2, 3):
            m[f"copy_up_kernel_writes_host_{ns}_streams_Gint/s"] = round(n / best(copy_up_write_host(ns)) / 1e9, 2)
        res["mixes"][str(b)] = m
        h_out.zero_()
        classic(2)()
        torch.cuda.synchronize()
        assert torch.equal(h_out, d_out.cpu())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
