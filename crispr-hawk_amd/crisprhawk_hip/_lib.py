"""ctypes binding of libhawk_hip.so (C ABI: include/hawk.h).

The product path has no CPU fallback: if the HIP library is missing or no GPU is visible,
every entry point raises HawkLibraryError / HawkDeviceError instead of computing on the host.
"""

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CRISPRHAWK_HIP_LIB", os.path.join(_HERE, "libhawk_hip.so"))

HAWK_OK = 0
HAWK_E_INVALID = -1
HAWK_E_HIP = -2
HAWK_E_CAPACITY = -3
HAWK_E_IUPAC = -4
HAWK_E_CFD = -5
HAWK_E_NODEVICE = -6
HAWK_E_UNSUPPORTED = -7
HAWK_E_COMM = -8
HAWK_E_OVERLAP = -9
HAWK_E_CLAMP = -10

EXPORTS = [
    "hawk_device_count", "hawk_init", "hawk_destroy", "hawk_strerror", "hawk_last_hip_error", "hawk_stream",
    "hawk_sync", "hawk_hapset_create", "hawk_hapset_destroy", "hawk_hapset_pack_ascii", "hawk_hapset_set_meta",
    "hawk_hapset_stride", "hawk_hapset_download_plane", "hawk_hapset_upload_planes", "hawk_pam_scan", "hawk_pam_scan_time",
    "hawk_search", "hawk_table_destroy", "hawk_table_counts", "hawk_table_download", "hawk_table_device_columns", "hawk_table_layout", "hawk_table_download_rows", "hawk_table_device_rows", "hawk_cfd",
    "hawk_genome_finalize", "hawk_offtarget_scan", "hawk_deepcpf1", "hawk_azimuth", "hawk_tm_nn", "hawk_hapset_expand",
    "hawk_table_collapse", "hawk_table_collapse_download", "hawk_gt_parse", "hawk_gt_destroy", "hawk_gt_codes", "hawk_gt_lists",
    "hawk_gt_lists_download", "hawk_gt_lists_indels", "hawk_host_build_segments", "hawk_host_posmap_rev", "hawk_release_cached_memory", "hawk_xplan_create", "hawk_xplan_set_meta", "hawk_xplan_run", "hawk_xplan_view", "hawk_xplan_cluster_stats", "hawk_xplan_cluster_rebuild", "hawk_host_gather_plan", "hawk_xplan_create_gt", "hawk_xplan_rows",
    "hawk_xplan_finish_meta", "hawk_xplan_segments", "hawk_xplan_install_meta", "hawk_host_alloc", "hawk_host_free", "hawk_hapset_rows_equal",
    "hawk_xplan_destroy", "hawk_hapset_set_ref_partner_range", "hawk_xplan_set_ref_partner_range", "hawk_table_collapse_ex", "hawk_table_collapse_export", "hawk_comm_unique_id", "hawk_comm_init",
    "hawk_comm_destroy", "hawk_comm_last_error", "hawk_comm_allgather_u64", "hawk_comm_gatherv", "hawk_table_gather", "hawk_host_ragged_join", "hawk_host_tsv_write", "hawk_host_vcf_index", "hawk_host_polish_rows", "hawk_host_variant_window", "hawk_host_polish_windows", "hawk_host_group_join", "hawk_host_group_samples", "hawk_gbt_predict", "hawk_gt_from_codes",
]


class HawkLibraryError(RuntimeError):
    """libhawk_hip.so is missing or does not export the C ABI — build it (python -c
    'import __graft_entry__ as g; g.build()' or make -C crispr-hawk_amd/csrc)."""


class HawkDeviceError(RuntimeError):
    """No MI355X visible / HIP failure.  There is deliberately no CPU path."""


class HawkCommError(RuntimeError):
    """RCCL failure (or librccl.so missing) in the multi-GPU exchange."""


class HawkStatusError(RuntimeError):
    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        super().__init__(f"{where}: {strerror(status)} (status {status}){': ' + detail if detail else ''}")


class SearchParams(C.Structure):
    _fields_ = [
        ("pam_fwd", C.c_uint64), ("pam_rev", C.c_uint64), ("pamlen", C.c_uint32), ("guidelen", C.c_uint32),
        ("right", C.c_uint32), ("score_cfdon", C.c_uint32), ("cfd_mm", C.c_void_p), ("cfd_pam", C.c_void_p),
    ]


class Timing(C.Structure):
    _fields_ = [
        ("count_ms", C.c_float), ("offsets_ms", C.c_float), ("emit_ms", C.c_float), ("total_ms", C.c_float),
        ("scanned_positions", C.c_uint64), ("emit_list_ms", C.c_float), ("v_count_ms", C.c_float), ("v_emit_ms", C.c_float),
        ("v_templates_ms", C.c_float), ("v_path", C.c_uint32), ("v_emit_rows_ms", C.c_float), ("reserved", C.c_float),
    ]


class GbtModel(C.Structure):
    _fields_ = [("n_trees", C.c_uint32), ("n_nodes", C.c_uint32), ("tree_off", C.c_void_p), ("feature", C.c_void_p),
                ("left", C.c_void_p), ("right", C.c_void_p), ("threshold", C.c_void_p), ("value", C.c_void_p),
                ("init", C.c_double), ("learning_rate", C.c_double)]


class OtParams(C.Structure):
    _fields_ = [("pam_fwd", C.c_uint64), ("pam_rev", C.c_uint64), ("pamlen", C.c_uint32), ("guidelen", C.c_uint32),
                ("right", C.c_uint32), ("max_mm", C.c_uint32)]


class OtTiming(C.Structure):
    _fields_ = [("scan_ms", C.c_float), ("sites_ms", C.c_float), ("match_ms", C.c_float), ("total_ms", C.c_float),
                ("n_sites", C.c_uint64), ("scanned_positions", C.c_uint64)]


_lib: Optional[C.CDLL] = None
_ctx = {}


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HawkLibraryError(f"{LIB_PATH} not found: the HIP extension has not been built")
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # e.g. libamdhip64 missing
            raise HawkLibraryError(f"cannot load {LIB_PATH}: {e}") from e
        missing = [s for s in EXPORTS if not hasattr(L, s)]
        if missing:
            raise HawkLibraryError(f"{LIB_PATH} lacks symbols {missing}")
        L.hawk_strerror.restype = C.c_char_p
        L.hawk_last_hip_error.restype = C.c_char_p
        L.hawk_stream.restype = C.c_void_p
        L.hawk_destroy.restype = None
        L.hawk_hapset_destroy.restype = None
        L.hawk_table_destroy.restype = None
        L.hawk_gt_destroy.restype = None
        L.hawk_xplan_destroy.restype = None
        L.hawk_comm_destroy.restype = None
        L.hawk_host_free.restype = None
        L.hawk_comm_last_error.restype = C.c_char_p
        for name in EXPORTS:
            fn = getattr(L, name)
            if fn.restype is C.c_int:
                fn.restype = C.c_int
        _lib = L
    return _lib


def strerror(status: int) -> str:
    return lib().hawk_strerror(status).decode()


def check(status: int, where: str) -> None:
    if status != HAWK_OK:
        detail = lib().hawk_last_hip_error().decode() if status == HAWK_E_HIP else ""
        if status == HAWK_E_COMM:
            raise HawkCommError(f"{where}: {lib().hawk_comm_last_error().decode()}")
        if status in (HAWK_E_HIP, HAWK_E_NODEVICE):
            raise HawkDeviceError(f"{where}: {strerror(status)} {detail}")
        raise HawkStatusError(status, where, detail)


def device_count() -> int:
    n = C.c_int(0)
    lib().hawk_device_count(C.byref(n))
    return n.value


def context(device: Optional[int] = None):
    """One hawk_ctx per process per device (include/hawk.h).  Device defaults to LOCAL_RANK."""
    if device is None:
        device = int(os.environ.get("HAWK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device not in _ctx:
        h = C.c_void_p()
        check(lib().hawk_init(device, C.byref(h)), "hawk_init")
        _ctx[device] = h
    return _ctx[device]


# ---------------------------------------------------------------------------- page-locked result buffers
# Blocks given back by arrays that died are kept for the next request of their size class (hipHostMalloc costs milliseconds per
# 100 MB) - in power-of-two classes, so that the differently sized exports of a per-tile loop share blocks, and only up to
# HAWK_PINNED_CACHE_MAX bytes (default 1 GiB): beyond that a returned block goes back to the driver (hawk_host_free), as does
# everything cached when the interpreter exits.  Page-locked memory is a scarce, unswappable resource.
_PINNED_FREE = {}  # size class (bytes) -> [address]
_PINNED_CACHED = [0]  # bytes sitting in _PINNED_FREE
_PINNED_MIN = int(os.environ.get("HAWK_PINNED_MIN", 1 << 20))  # bytes from which a host array is page-locked
_PINNED_CACHE_MAX = int(os.environ.get("HAWK_PINNED_CACHE_MAX", 1 << 30))


def _pinned_release(size: int, addr: int) -> None:
    if _lib is None:
        return
    if _PINNED_CACHED[0] + size <= _PINNED_CACHE_MAX:
        _PINNED_FREE.setdefault(size, []).append(addr)
        _PINNED_CACHED[0] += size
    else:
        _lib.hawk_host_free(C.c_void_p(addr))


def pinned_trim() -> None:
    """hand every cached page-locked block back to the driver"""
    for size, pool in _PINNED_FREE.items():
        while pool:
            addr = pool.pop()
            if _lib is not None:
                _lib.hawk_host_free(C.c_void_p(addr))
    _PINNED_CACHED[0] = 0


def pinned_stats() -> dict:
    return {"cached_bytes": _PINNED_CACHED[0], "blocks": sum(len(v) for v in _PINNED_FREE.values()), "cache_max": _PINNED_CACHE_MAX}


def _size_class(nbytes: int) -> int:
    """smallest of {2^k, 3 * 2^(k-2)} (k >= 20) holding nbytes: at most a third of a block is padding"""
    k = max(20, (nbytes - 1).bit_length())
    mid = 3 << (k - 2)
    return mid if (k > 20 and mid >= nbytes) else 1 << k


def pinned_empty(n: int, dtype, device: Optional[int] = None):
    """np.empty(n, dtype) in page-locked host memory when the array is large (a device-to-host copy into it runs at link
    speed); small arrays - and any array the driver refuses to page-lock - are plain numpy.  `device` names the device whose
    context allocates (the table's own; default: the process default).  The block returns to the cache when the array is collected."""
    import numpy as np
    import weakref
    dt = np.dtype(dtype)
    nbytes = int(n) * dt.itemsize
    if nbytes < _PINNED_MIN:
        return np.empty(n, dtype=dt)
    size = _size_class(nbytes)
    pool = _PINNED_FREE.get(size)
    if pool:
        addr = pool.pop()
        _PINNED_CACHED[0] -= size
    else:
        p = C.c_void_p()
        rc = lib().hawk_host_alloc(context(device), C.c_uint64(size), C.byref(p))
        if rc != HAWK_OK or not p.value:
            if _PINNED_CACHED[0]:  # make room once, then try again
                pinned_trim()
                rc = lib().hawk_host_alloc(context(device), C.c_uint64(size), C.byref(p))
            if rc != HAWK_OK or not p.value:
                return np.empty(n, dtype=dt)  # pageable: slower copies, same result
        addr = p.value
    buf = (C.c_uint8 * size).from_address(addr)
    arr = np.frombuffer(buf, dtype=dt, count=int(n))
    weakref.finalize(buf, _pinned_release, size, addr)
    return arr


import atexit  # noqa: E402

atexit.register(pinned_trim)
