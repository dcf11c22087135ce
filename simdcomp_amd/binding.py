"""ctypes binding of libsimdcomp_hip.so (the C ABI in include/simdcomp_hip.h and
include/simdcomp.h).  The library is the product; this module only passes
pointers.  Importing it fails loudly if the shared library is missing: there is
no Python or CPU implementation to fall back to.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SIMDCOMP_HIP_LIB names another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("SIMDCOMP_HIP_LIB") or os.path.join(_HERE, "libsimdcomp_hip.so")

u32, u64, vp, sz = C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t

PLAIN, D1, FOR = 0, 1, 2

# name -> (restype, argtypes): every symbol the two headers declare
DEVICE_API = {
    "simdcomp_hip_version": (C.c_int, []),
    "simdcomp_hip_last_error": (C.c_char_p, []),
    "simdcomp_hip_set_error_handler": (None, [vp, vp]),
    "simdcomp_hip_pack": (C.c_int, [vp, vp, sz, u32, C.c_int, vp]),
    "simdcomp_hip_unpack": (C.c_int, [vp, vp, sz, u32, vp]),
    "simdcomp_hip_packd1": (C.c_int, [u32, vp, vp, vp, sz, u32, C.c_int, vp, vp]),
    "simdcomp_hip_unpackd1": (C.c_int, [vp, vp, vp, sz, u32, vp]),
    "simdcomp_hip_packFOR": (C.c_int, [vp, vp, vp, sz, u32, vp]),
    "simdcomp_hip_unpackFOR": (C.c_int, [vp, vp, vp, sz, u32, vp]),
    "simdcomp_hip_pack_mixed": (C.c_int, [C.c_int, C.c_int, u32, vp, vp, vp, vp, vp, sz, vp, vp]),
    "simdcomp_hip_unpack_mixed": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, sz, vp]),
    "simdcomp_hip_maxbits": (C.c_int, [vp, sz, vp, vp]),
    "simdcomp_hip_maxbitsd1": (C.c_int, [u32, vp, vp, sz, vp, vp, vp]),
    "simdcomp_hip_maxmin": (C.c_int, [vp, sz, vp, vp, vp, vp]),
    "simdcomp_hip_reduce_length": (C.c_int, [vp, sz, u32, vp, vp]),
    "simdcomp_hip_payload_offsets_workspace": (sz, [sz]),
    "simdcomp_hip_payload_offsets": (C.c_int, [vp, sz, vp, vp, vp, vp]),
    "simdcomp_hip_payload_offsets_noalloc_max_blocks": (sz, []),
    "simdcomp_hip_payload_offsets_noalloc": (C.c_int, [vp, sz, vp, vp, vp]),
    "simdcomp_hip_chain_workspace": (sz, [sz]),
    "simdcomp_hip_chain_offsets": (C.c_int, [u32, vp, vp, vp, sz, C.c_int, vp, vp, vp]),
    "simdcomp_hip_pack_stream": (C.c_int, [C.c_int, C.c_int, u32, vp, vp, vp, vp, vp, sz, vp, vp]),
    "simdcomp_hip_unpack_stream": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, sz, vp]),
    "simdcomp_hip_stream_build": (C.c_int, [vp, vp, vp, sz, vp, vp]),
    "simdcomp_hip_stream_unbuild": (C.c_int, [vp, vp, vp, sz, vp, vp]),
    "simdcomp_hip_stream_index_workspace": (sz, [sz]),
    "simdcomp_hip_stream_index": (C.c_int, [vp, sz, sz, vp, vp, vp, vp, vp]),
    "simdcomp_hip_stream_compress_workspace": (sz, [sz]),
    "simdcomp_hip_stream_compress": (C.c_int, [C.c_int, u32, vp, sz, vp, vp, vp, vp]),
    "simdcomp_hip_stream_decode_workspace": (sz, [sz, sz]),
    "simdcomp_hip_stream_decode": (C.c_int, [C.c_int, u32, vp, sz, sz, vp, vp, vp, vp]),
    "simdcomp_hip_select": (C.c_int, [C.c_int, u32, vp, vp, vp, vp, u32, C.c_int, vp, vp, sz, vp, vp]),
    "simdcomp_hip_search": (C.c_int, [C.c_int, u32, vp, vp, vp, vp, u32, C.c_int, vp, vp, vp, sz, vp, vp, vp, vp]),
    "simdcomp_hip_block_directory_bytes": (sz, [sz]),
    "simdcomp_hip_block_directory": (C.c_int, [u32, vp, vp, vp, sz, C.c_int, vp, vp]),
    "simdcomp_hip_select_directory": (C.c_int, [C.c_int, vp, vp, vp, vp, sz, vp, vp]),
    "simdcomp_hip_search_directory": (C.c_int, [C.c_int, vp, vp, vp, vp, vp, sz, vp, vp, vp, vp]),
    "simdcomp_hip_scand1": (C.c_int, [u32, vp, vp, vp, vp, u32, C.c_int, sz, vp, vp]),
    "simdcomp_hip_pack_lanes": (C.c_int, [C.c_int, vp, vp, sz, u32, C.c_int, vp]),
    "simdcomp_hip_unpack_lanes": (C.c_int, [C.c_int, vp, vp, sz, u32, vp]),
    "simdcomp_hip_maxbits_lanes": (C.c_int, [C.c_int, vp, sz, vp, vp]),
    "simdcomp_hip_compressedbytes": (sz, [sz, u32]),
    "simdcomp_hip_pack_length": (C.c_int, [vp, sz, vp, u32, vp]),
    "simdcomp_hip_unpack_length": (C.c_int, [vp, sz, vp, u32, vp]),
    "simdcomp_hip_packFOR_length": (C.c_int, [u32, vp, sz, vp, u32, C.c_int, vp]),
    "simdcomp_hip_unpackFOR_length": (C.c_int, [u32, vp, sz, vp, u32, vp]),
    "simdcomp_hip_partition": (C.c_int, [sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]),
    "simdcomp_hip_partition_by_bytes": (C.c_int, [vp, sz, C.c_int, C.c_int, C.POINTER(sz), C.POINTER(sz)]),
    "simdcomp_hip_checksum": (C.c_int, [vp, vp, sz, u64, vp, vp]),
}

HOST_API = {
    "bits": (u32, [u32]),
    "maxbits": (u32, [vp]),
    "maxbits_length": (u32, [vp, u32]),
    "simdmin": (u32, [vp]),
    "simdmin_length": (u32, [vp, u32]),
    "simdmaxmin_length": (None, [vp, u32, vp, vp]),
    "simdmaxmin": (None, [vp, vp, vp]),
    "simdmaxbitsd1": (u32, [u32, vp]),
    "simdmaxbitsd1_length": (u32, [u32, vp, u32]),
    "simdpack": (None, [vp, vp, u32]),
    "simdpackwithoutmask": (None, [vp, vp, u32]),
    "simdunpack": (None, [vp, vp, u32]),
    "simdpack_compressedbytes": (C.c_int, [C.c_int, u32]),
    "simdpack_length": (vp, [vp, sz, vp, u32]),
    "simdunpack_length": (vp, [vp, sz, vp, u32]),
    "simdpack_shortlength": (vp, [vp, C.c_int, vp, u32]),
    "simdunpack_shortlength": (vp, [vp, C.c_int, vp, u32]),
    "simdfastset": (None, [vp, u32, u32, sz]),
    "simdpackd1": (None, [u32, vp, vp, u32]),
    "simdpackwithoutmaskd1": (None, [u32, vp, vp, u32]),
    "simdunpackd1": (None, [u32, vp, vp, u32]),
    "simdpackFOR": (None, [u32, vp, vp, u32]),
    "simdunpackFOR": (None, [u32, vp, vp, u32]),
    "simdpackFOR_compressedbytes": (C.c_int, [C.c_int, u32]),
    "simdpackFOR_length": (vp, [u32, vp, C.c_int, vp, u32]),
    "simdunpackFOR_length": (vp, [u32, vp, C.c_int, vp, u32]),
    "avxmaxbits": (u32, [vp]),
    "avxpack": (None, [vp, vp, u32]),
    "avxpackwithoutmask": (None, [vp, vp, u32]),
    "avxunpack": (None, [vp, vp, u32]),
    "avx512maxbits": (u32, [vp]),
    "avx512pack": (None, [vp, vp, u32]),
    "avx512packwithoutmask": (None, [vp, vp, u32]),
    "avx512unpack": (None, [vp, vp, u32]),
    "simdselectd1": (u32, [u32, vp, u32, C.c_int]),
    "simdsearchwithlengthd1": (C.c_int, [u32, vp, u32, C.c_int, u32, vp]),
    "simdsearchd1": (C.c_int, [vp, vp, u32, u32, vp]),
    "simdscand1": (None, [vp, vp, u32]),
    "simdfastsetd1": (None, [u32, vp, u32, u32, sz]),
    "simdfastsetd1fromprevious": (None, [vp, u32, u32, u32, sz]),
    "simdselectFOR": (u32, [u32, vp, u32, C.c_int]),
    "simdfastsetFOR": (None, [u32, vp, u32, u32, sz]),
    "simdsearchwithlengthFOR": (C.c_int, [u32, vp, u32, C.c_int, u32, vp]),
}


class SimdcompHipError(RuntimeError):
    pass


def load():
    """dlopen the library and attach prototypes to every declared symbol."""
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C simdcomp_amd/csrc` "
            "(or __graft_entry__.build()); there is no fallback implementation")
    # PyTorch ships its own copy of the HIP runtime under the same soname (libamdhip64.so.7) as the system one this
    # library is linked to, and the dynamic loader keeps whichever comes first for the whole process.  torch cannot
    # start on the system copy ("No HIP GPUs are available"), this library runs on either: so in a process that has
    # torch installed, let torch load its runtime first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for table in (DEVICE_API, HOST_API):
        for name, (res, args) in table.items():
            f = getattr(lib, name)  # AttributeError if the library does not export it
            f.restype = res
            f.argtypes = args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().simdcomp_hip_last_error().decode()
        raise SimdcompHipError(f"{what} failed with code {rc}: {msg}")


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def current_stream():
    import torch

    return torch.cuda.current_stream().cuda_stream
