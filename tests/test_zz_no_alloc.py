"""simdcomp_hip_payload_offsets_noalloc (the offsets scan for callers that cannot allocate: its tile sums live in a
device array of the library) against simdcomp_hip_payload_offsets with a caller workspace, through the C ABI: two
calls in a row, the second on a smaller input, plus the size limit.  Collected last on purpose."""
import numpy as np
import pytest
import torch

from simdcomp_amd import binding as B, codec, synth

pytestmark = pytest.mark.gpu


def offsets_both_ways(bits):
    L = B.lib()
    n = bits.numel()
    st = torch.cuda.current_stream().cuda_stream
    want_off, want_total = codec.payload_offsets(bits)
    off = torch.full((n + 8,), -1, dtype=torch.int64, device=bits.device)  # 8 canaries behind the result
    total = torch.full((2,), -1, dtype=torch.int64, device=bits.device)
    B.check(L.simdcomp_hip_payload_offsets_noalloc(B.ptr(bits), n, B.ptr(off), B.ptr(total), st), "payload_offsets_noalloc")
    torch.cuda.synchronize()
    assert torch.equal(off[:n], want_off)
    assert int(total[0]) == int(want_total[0]) and int(total[1]) == -1
    assert bool((off[n:] == -1).all())
    return off[:n], int(total[0])


def test_noalloc_offsets_match_the_workspace_form_twice():
    dev = torch.device("cuda", 0)
    # first call: several thousand tiles, ragged end; second call: smaller input on the same scratch array
    for seed, nblk in ((1, 3 * 4096 * 1000 + 77), (2, 4096 * 3 + 5), (3, 1), (4, 4096)):
        w = synth.skewed_widths(seed, nblk, 0, 32)
        bits = torch.from_numpy(w).to(dev)
        off, total = offsets_both_ways(bits)
        # and against plain arithmetic on the host
        want = np.concatenate([[0], np.cumsum(w.astype(np.uint64) * 16)])
        assert total == int(want[-1])
        sample = np.unique(np.concatenate([np.arange(min(nblk, 5000)), synth.rand_u32(seed, 2000) % np.uint32(nblk), [nblk - 1]])).astype(np.int64)
        assert np.array_equal(off.cpu().numpy()[sample].astype(np.uint64), want[:-1][sample])


def test_noalloc_rejects_sizes_beyond_its_scratch():
    L = B.lib()
    cap = L.simdcomp_hip_payload_offsets_noalloc_max_blocks()
    assert cap >= 2 ** 27
    dummy = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
    out = torch.zeros(16, dtype=torch.int64, device="cuda:0")
    rc = L.simdcomp_hip_payload_offsets_noalloc(B.ptr(dummy), cap + 1, B.ptr(out), None, None)
    assert rc == -5 and b"max_blocks" in L.simdcomp_hip_last_error()
    assert L.simdcomp_hip_payload_offsets_noalloc(None, 0, None, None, None) == 0  # nothing to do, nothing touched
    torch.cuda.synchronize()


def test_noalloc_call_captured_in_a_graph_and_replayed():
    """What the entry point is for: captured in a HIP graph (after one plain call on the device, which resolves the
    address of the library's scratch array -- that look-up is not capturable), then replayed on new widths in the same
    buffers.  Capture fails on any allocation, so passing also shows that the call makes none."""
    L = B.lib()
    dev = torch.device("cuda", 0)
    nblk = 4096 * 7 + 123
    bits = torch.from_numpy(synth.skewed_widths(11, nblk, 0, 32)).to(dev)
    off = torch.zeros(nblk, dtype=torch.int64, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        B.check(L.simdcomp_hip_payload_offsets_noalloc(B.ptr(bits), nblk, B.ptr(off), B.ptr(total), side.cuda_stream), "warm-up")
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        B.check(L.simdcomp_hip_payload_offsets_noalloc(B.ptr(bits), nblk, B.ptr(off), B.ptr(total), st), "captured")
    for seed in (12, 13):
        w = synth.skewed_widths(seed, nblk, 0, 32)
        bits.copy_(torch.from_numpy(w).to(dev))
        off.fill_(-1)
        total.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        want = np.concatenate([[0], np.cumsum(w.astype(np.uint64) * 16)])
        assert int(total[0]) == int(want[-1])
        assert np.array_equal(off.cpu().numpy().astype(np.uint64), want[:-1])
