"""Thin torch-tensor front end over the batched C ABI (include/simdcomp_hip.h).

torch is used for device memory and streams only.  Integer arrays are int32
tensors holding the uint32 bit patterns (torch has no first-class uint32),
payloads and widths are uint8 tensors, payload offsets int64 tensors holding
uint64 byte offsets.  Function names follow the reference's API; each call
enqueues one kernel on the current torch stream.
"""
import torch

from . import binding as B


def _i32(x):
    assert x.dtype == torch.int32 and x.is_cuda and x.is_contiguous()
    return x


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _s32(v):
    """uint32 value -> python int accepted by c_uint32."""
    return int(v) & 0xFFFFFFFF


def compressedbytes(length, bit):
    return int(B.lib().simdcomp_hip_compressedbytes(length, bit))


def pack(x, bit, masked=False, out=None):
    """nblocks x simdpack / simdpackwithoutmask at one width."""
    x = _i32(x)
    nblocks = x.numel() // 128
    if out is None:
        out = torch.empty(nblocks * 16 * bit, dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_pack(B.ptr(x), B.ptr(out), nblocks, bit, int(masked), _stream()), "pack")
    return out


def unpack(payload, nblocks, bit, out=None):
    if out is None:
        out = torch.empty(nblocks * 128, dtype=torch.int32, device=payload.device)
    B.check(B.lib().simdcomp_hip_unpack(B.ptr(payload), B.ptr(out), nblocks, bit, _stream()), "unpack")
    return out


def packd1(x, bit, initvalues=None, initvalue=0, masked=False, out=None, initvalues_out=None):
    x = _i32(x)
    nblocks = x.numel() // 128
    if out is None:
        out = torch.empty(nblocks * 16 * bit, dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_packd1(_s32(initvalue), B.ptr(initvalues), B.ptr(x), B.ptr(out), nblocks, bit,
                                        int(masked), B.ptr(initvalues_out), _stream()), "packd1")
    return out


def unpackd1(initvalues, payload, nblocks, bit, out=None):
    if out is None:
        out = torch.empty(nblocks * 128, dtype=torch.int32, device=payload.device)
    B.check(B.lib().simdcomp_hip_unpackd1(B.ptr(initvalues), B.ptr(payload), B.ptr(out), nblocks, bit, _stream()),
            "unpackd1")
    return out


def packFOR(base, x, bit, out=None):
    x = _i32(x)
    nblocks = x.numel() // 128
    if out is None:
        out = torch.empty(nblocks * 16 * bit, dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_packFOR(B.ptr(base), B.ptr(x), B.ptr(out), nblocks, bit, _stream()), "packFOR")
    return out


def unpackFOR(base, payload, nblocks, bit, out=None):
    if out is None:
        out = torch.empty(nblocks * 128, dtype=torch.int32, device=payload.device)
    B.check(B.lib().simdcomp_hip_unpackFOR(B.ptr(base), B.ptr(payload), B.ptr(out), nblocks, bit, _stream()),
            "unpackFOR")
    return out


def maxbits(x):
    x = _i32(x)
    nblocks = x.numel() // 128
    out = torch.empty(nblocks, dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_maxbits(B.ptr(x), nblocks, B.ptr(out), _stream()), "maxbits")
    return out


def maxbitsd1(x, initvalues=None, initvalue=0, initvalues_out=None):
    x = _i32(x)
    nblocks = x.numel() // 128
    out = torch.empty(nblocks, dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_maxbitsd1(_s32(initvalue), B.ptr(initvalues), B.ptr(x), nblocks, B.ptr(out),
                                           B.ptr(initvalues_out), _stream()), "maxbitsd1")
    return out


def maxmin(x, want_bits=True):
    x = _i32(x)
    nblocks = x.numel() // 128
    lo = torch.empty(nblocks, dtype=torch.int32, device=x.device)
    hi = torch.empty(nblocks, dtype=torch.int32, device=x.device)
    bits = torch.empty(nblocks, dtype=torch.uint8, device=x.device) if want_bits else None
    B.check(B.lib().simdcomp_hip_maxmin(B.ptr(x), nblocks, B.ptr(lo), B.ptr(hi), B.ptr(bits), _stream()), "maxmin")
    return lo, hi, bits


def reduce_length(x, length=None, initvalue=0):
    """-> int32[4] tensor: OR of values, OR of differences, min, max."""
    n = x.numel() if length is None else length
    out = torch.empty(4, dtype=torch.int32, device=x.device)
    B.check(B.lib().simdcomp_hip_reduce_length(B.ptr(x), n, _s32(initvalue), B.ptr(out), _stream()), "reduce_length")
    return out


def payload_offsets(bits):
    """Exclusive scan of 16*bits -> (int64 offsets tensor, total bytes as a 1-element int64 tensor)."""
    nblocks = bits.numel()
    off = torch.empty(nblocks, dtype=torch.int64, device=bits.device)
    total = torch.zeros(1, dtype=torch.int64, device=bits.device)
    ws = torch.empty(max(8, B.lib().simdcomp_hip_payload_offsets_workspace(nblocks)), dtype=torch.uint8,
                     device=bits.device)
    B.check(B.lib().simdcomp_hip_payload_offsets(B.ptr(bits), nblocks, B.ptr(off), B.ptr(total), B.ptr(ws),
                                                 _stream()), "payload_offsets")
    return off, total


def pack_mixed(mode, x, bits, offsets, payload, initvalues=None, initvalue=0, masked=False, initvalues_out=None):
    x = _i32(x)
    nblocks = x.numel() // 128
    B.check(B.lib().simdcomp_hip_pack_mixed(mode, int(masked), _s32(initvalue), B.ptr(initvalues), B.ptr(x),
                                            B.ptr(payload), B.ptr(offsets), B.ptr(bits), nblocks,
                                            B.ptr(initvalues_out), _stream()), "pack_mixed")
    return payload


def unpack_mixed(mode, payload, bits, offsets, nblocks, initvalues=None, out=None):
    if out is None:
        out = torch.empty(nblocks * 128, dtype=torch.int32, device=payload.device)
    B.check(B.lib().simdcomp_hip_unpack_mixed(mode, B.ptr(initvalues), B.ptr(payload), B.ptr(offsets), B.ptr(bits),
                                              B.ptr(out), nblocks, _stream()), "unpack_mixed")
    return out


def pack_stream(mode, x, bits, offsets, stream, initvalues=None, initvalue=0, masked=False, initvalues_out=None):
    """pack_mixed writing the [b][payload] byte stream of example.c:56-75 in place."""
    x = _i32(x)
    nblocks = x.numel() // 128
    B.check(B.lib().simdcomp_hip_pack_stream(mode, int(masked), _s32(initvalue), B.ptr(initvalues), B.ptr(x),
                                             B.ptr(stream), B.ptr(offsets), B.ptr(bits), nblocks,
                                             B.ptr(initvalues_out), _stream()), "pack_stream")
    return stream


def unpack_stream(mode, stream, bits, offsets, nblocks, initvalues=None, out=None):
    """unpack_mixed reading the [b][payload] byte stream in place."""
    if out is None:
        out = torch.empty(nblocks * 128, dtype=torch.int32, device=stream.device)
    B.check(B.lib().simdcomp_hip_unpack_stream(mode, B.ptr(initvalues), B.ptr(stream), B.ptr(offsets), B.ptr(bits),
                                               B.ptr(out), nblocks, _stream()), "unpack_stream")
    return out


def chain_offsets(payload, bits, offsets, initvalue=0, out=None, stream_layout=False):
    """Offsets simdunpackd1 needs per block when none were stored (example.c:101-110)."""
    nblocks = bits.numel()
    if out is None:
        out = torch.empty(nblocks, dtype=torch.int32, device=bits.device)
    ws = torch.empty(max(16, B.lib().simdcomp_hip_chain_workspace(nblocks)), dtype=torch.uint8, device=bits.device)
    B.check(B.lib().simdcomp_hip_chain_offsets(_s32(initvalue), B.ptr(payload), B.ptr(offsets), B.ptr(bits), nblocks,
                                               int(stream_layout), B.ptr(out), B.ptr(ws), _stream()), "chain_offsets")
    return out


def stream_build(payload, bits, offsets, total_bytes, out=None):
    """Aligned arena -> the [b][payload] byte stream of example.c:56-75."""
    nblocks = bits.numel()
    if out is None:
        out = torch.empty(nblocks + total_bytes, dtype=torch.uint8, device=bits.device)
    B.check(B.lib().simdcomp_hip_stream_build(B.ptr(payload), B.ptr(offsets), B.ptr(bits), nblocks, B.ptr(out),
                                              _stream()), "stream_build")
    return out


def stream_unbuild(stream, bits, offsets, total_bytes, out=None):
    """Payloads of a [b][payload] stream -> 16-byte aligned arena."""
    nblocks = bits.numel()
    if out is None:
        out = torch.empty(total_bytes + 16, dtype=torch.uint8, device=bits.device)
    B.check(B.lib().simdcomp_hip_stream_unbuild(B.ptr(stream), B.ptr(offsets), B.ptr(bits), nblocks, B.ptr(out),
                                                _stream()), "stream_unbuild")
    return out


def stream_index(stream, max_blocks, nbytes=None):
    """-> (bits uint8[max_blocks], offsets int64[max_blocks], result int64[3] = records, payload bytes, status)."""
    nbytes = stream.numel() if nbytes is None else nbytes
    dev = stream.device
    bits = torch.empty(max_blocks, dtype=torch.uint8, device=dev)
    off = torch.empty(max_blocks, dtype=torch.int64, device=dev)
    res = torch.empty(3, dtype=torch.int64, device=dev)
    ws = torch.empty(max(16, B.lib().simdcomp_hip_stream_index_workspace(nbytes)), dtype=torch.uint8, device=dev)
    B.check(B.lib().simdcomp_hip_stream_index(B.ptr(stream), nbytes, max_blocks, B.ptr(bits), B.ptr(off), B.ptr(res),
                                              B.ptr(ws), _stream()), "stream_index")
    return bits, off, res


def stream_compress(mode, x, initvalue=0):
    """example.c's compress() (D1) / varying_bit_width_compress() (PLAIN) -> (uint8 stream buffer, length as 1-element int64 tensor)."""
    x = _i32(x)
    nblocks = x.numel() // 128
    buf = torch.empty(nblocks * 513 + 16, dtype=torch.uint8, device=x.device)
    nbytes = torch.zeros(1, dtype=torch.int64, device=x.device)
    ws = torch.empty(max(16, B.lib().simdcomp_hip_stream_compress_workspace(nblocks)), dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_stream_compress(mode, _s32(initvalue), B.ptr(x), nblocks, B.ptr(buf), B.ptr(nbytes),
                                                 B.ptr(ws), _stream()), "stream_compress")
    return buf, nbytes


def stream_decode(mode, stream, nblocks, initvalue=0, nbytes=None):
    """example.c's decode loops -> (int32[nblocks*128], result int64[3] = records, payload bytes, status)."""
    nbytes = stream.numel() if nbytes is None else nbytes
    out = torch.empty(nblocks * 128, dtype=torch.int32, device=stream.device)
    res = torch.zeros(3, dtype=torch.int64, device=stream.device)
    ws = torch.empty(max(16, B.lib().simdcomp_hip_stream_decode_workspace(nbytes, nblocks)), dtype=torch.uint8,
                     device=stream.device)
    B.check(B.lib().simdcomp_hip_stream_decode(mode, _s32(initvalue), B.ptr(stream), nbytes, nblocks, B.ptr(out), B.ptr(res),
                                               B.ptr(ws), _stream()), "stream_decode")
    return out, res


def select(mode, payload, slots, bit=0, bits=None, offsets=None, initvalues=None, initvalue=0, query_blocks=None,
           stream_layout=False):
    """Batched simdselectd1 / simdselectFOR: slots int32[nq] -> values int32[nq]."""
    nq = slots.numel()
    out = torch.empty(nq, dtype=torch.int32, device=slots.device)
    B.check(B.lib().simdcomp_hip_select(mode, _s32(initvalue), B.ptr(initvalues), B.ptr(payload), B.ptr(offsets),
                                        B.ptr(bits), bit, int(stream_layout), B.ptr(query_blocks), B.ptr(slots), nq,
                                        B.ptr(out), _stream()), "select")
    return out


def block_directory(bits, offsets, initvalues=None, initvalue=0, stream_layout=False):
    """One 16-byte entry per block (payload position, width, initvalue) for select / search with directory=."""
    nblk = bits.numel()
    d = torch.empty(max(16, B.lib().simdcomp_hip_block_directory_bytes(nblk)), dtype=torch.uint8, device=bits.device)
    B.check(B.lib().simdcomp_hip_block_directory(_s32(initvalue), B.ptr(initvalues), B.ptr(offsets), B.ptr(bits), nblk,
                                                 int(stream_layout), B.ptr(d), _stream()), "block_directory")
    return d


def select_directory(mode, payload, directory, slots, query_blocks=None):
    """simdcomp_hip_select with the blocks' metadata taken from a block_directory."""
    nq = slots.numel()
    out = torch.empty(nq, dtype=torch.int32, device=slots.device)
    B.check(B.lib().simdcomp_hip_select_directory(mode, B.ptr(payload), B.ptr(directory), B.ptr(query_blocks), B.ptr(slots),
                                                  nq, B.ptr(out), _stream()), "select_directory")
    return out


def search_directory(mode, payload, directory, keys, query_blocks=None, lengths=None, want_last4=False):
    """simdcomp_hip_search with the blocks' metadata taken from a block_directory."""
    nq = keys.numel()
    pos = torch.empty(nq, dtype=torch.int32, device=keys.device)
    val = torch.empty(nq, dtype=torch.int32, device=keys.device)
    last4 = torch.empty(nq * 4, dtype=torch.int32, device=keys.device) if want_last4 else None
    B.check(B.lib().simdcomp_hip_search_directory(mode, B.ptr(payload), B.ptr(directory), B.ptr(query_blocks), B.ptr(keys),
                                                  B.ptr(lengths), nq, B.ptr(pos), B.ptr(val), B.ptr(last4), _stream()),
            "search_directory")
    return (pos, val, last4) if want_last4 else (pos, val)


def search(mode, payload, keys, bit=0, bits=None, offsets=None, initvalues=None, initvalue=0, query_blocks=None,
           lengths=None, stream_layout=False, want_last4=False):
    """Batched simdsearchwithlengthd1 / simdsearchwithlengthFOR -> (positions int32[nq], values int32[nq][, last4])."""
    nq = keys.numel()
    pos = torch.empty(nq, dtype=torch.int32, device=keys.device)
    val = torch.empty(nq, dtype=torch.int32, device=keys.device)
    last4 = torch.empty(nq * 4, dtype=torch.int32, device=keys.device) if want_last4 else None
    B.check(B.lib().simdcomp_hip_search(mode, _s32(initvalue), B.ptr(initvalues), B.ptr(payload), B.ptr(offsets),
                                        B.ptr(bits), bit, int(stream_layout), B.ptr(query_blocks), B.ptr(keys),
                                        B.ptr(lengths), nq, B.ptr(pos), B.ptr(val), B.ptr(last4), _stream()), "search")
    return (pos, val, last4) if want_last4 else (pos, val)


def scand1(payload, nblocks, bit=0, bits=None, offsets=None, initvalues=None, initvalue=0, stream_layout=False):
    """Batched simdscand1: the last four values of every block, int32[nblocks*4]."""
    out = torch.empty(nblocks * 4, dtype=torch.int32, device=payload.device)
    B.check(B.lib().simdcomp_hip_scand1(_s32(initvalue), B.ptr(initvalues), B.ptr(payload), B.ptr(offsets), B.ptr(bits),
                                        bit, int(stream_layout), nblocks, B.ptr(out), _stream()), "scand1")
    return out


def pack_lanes(lanes, x, bit, masked=False, out=None):
    """nblocks x avxpack / avxpackwithoutmask (lanes=8) or avx512pack* (lanes=16)."""
    x = _i32(x)
    nblocks = x.numel() // (32 * lanes)
    if out is None:
        out = torch.empty(nblocks * 4 * lanes * bit, dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_pack_lanes(lanes, B.ptr(x), B.ptr(out), nblocks, bit, int(masked), _stream()), "pack_lanes")
    return out


def unpack_lanes(lanes, payload, nblocks, bit, out=None):
    if out is None:
        out = torch.empty(nblocks * 32 * lanes, dtype=torch.int32, device=payload.device)
    B.check(B.lib().simdcomp_hip_unpack_lanes(lanes, B.ptr(payload), B.ptr(out), nblocks, bit, _stream()), "unpack_lanes")
    return out


def maxbits_lanes(lanes, x):
    x = _i32(x)
    nblocks = x.numel() // (32 * lanes)
    out = torch.empty(nblocks, dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_maxbits_lanes(lanes, B.ptr(x), nblocks, B.ptr(out), _stream()), "maxbits_lanes")
    return out


def pack_length(x, bit, out=None):
    x = _i32(x)
    n = x.numel()
    if out is None:
        out = torch.empty(compressedbytes(n, bit), dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_pack_length(B.ptr(x), n, B.ptr(out), bit, _stream()), "pack_length")
    return out


def unpack_length(payload, n, bit, out=None):
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=payload.device)
    B.check(B.lib().simdcomp_hip_unpack_length(B.ptr(payload), n, B.ptr(out), bit, _stream()), "unpack_length")
    return out


def packFOR_length(base, x, bit, out=None):
    x = _i32(x)
    n = x.numel()
    if out is None:
        out = torch.empty(compressedbytes(n, bit), dtype=torch.uint8, device=x.device)
    B.check(B.lib().simdcomp_hip_packFOR_length(_s32(base), B.ptr(x), n, B.ptr(out), bit, 0, _stream()),
            "packFOR_length")
    return out


def unpackFOR_length(base, payload, n, bit, out=None):
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=payload.device)
    B.check(B.lib().simdcomp_hip_unpackFOR_length(_s32(base), B.ptr(payload), n, B.ptr(out), bit, _stream()),
            "unpackFOR_length")
    return out


def checksum(data, other=None, index0=0, acc=None):
    """acc[0] += checksum(data), acc[1] += #words differing from other; returns acc (int64[2])."""
    if acc is None:
        acc = torch.zeros(2, dtype=torch.int64, device=data.device)
    nwords = data.numel() * data.element_size() // 4
    B.check(B.lib().simdcomp_hip_checksum(B.ptr(data), B.ptr(other), nwords, index0, B.ptr(acc), _stream()),
            "checksum")
    return acc


def partition(nblocks, nparts, part):
    import ctypes as C

    first, count = C.c_size_t(), C.c_size_t()
    B.check(B.lib().simdcomp_hip_partition(nblocks, nparts, part, C.byref(first), C.byref(count)), "partition")
    return first.value, count.value


def partition_by_bytes(h_offsets, nparts, part):
    """h_offsets: numpy uint64 array of nblocks+1 byte offsets (host)."""
    import ctypes as C

    first, count = C.c_size_t(), C.c_size_t()
    B.check(B.lib().simdcomp_hip_partition_by_bytes(B.ptr(h_offsets), len(h_offsets) - 1, nparts, part,
                                                    C.byref(first), C.byref(count)), "partition_by_bytes")
    return first.value, count.value
