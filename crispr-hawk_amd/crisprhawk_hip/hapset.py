"""Device-resident haplotype set and guide table (thin wrappers over the C ABI).

``DeviceHapSet`` owns the bit-sliced planes of a list of haplotypes in HBM together with the
per-haplotype metadata the reference's search reads from ``Haplotype`` objects (samples ==
"REF", scan bounds, position map).  ``GuideTable`` is the columnar result of the fused search:
the reference materialises one Python ``Guide`` per row (search_guides.py:488-502); here rows
stay as numpy columns and ``Guide`` objects are created on demand.
"""

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

GUIDESEQPAD = 10  # reference guide.py:21
_CODE2CHAR = np.frombuffer(b"?ACMGRSVTWYHKDBN?acmgrsvtwyhkdbn", dtype=np.uint8)


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _as_u8(seq) -> np.ndarray:
    if isinstance(seq, np.ndarray):
        return seq if seq.dtype == np.uint8 else seq.astype(np.uint8)
    if isinstance(seq, str):
        seq = seq.encode("ascii")
    return np.frombuffer(seq, dtype=np.uint8)


def segments_from_posmap(posmap: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Position map (haplotype.py:90-159) -> unit-slope segments (rel_start, genomic_start).
    Every index where posmap[i] != posmap[i-1] + 1 opens a segment (an inserted base repeats
    its anchor's position and therefore opens a segment of its own)."""
    pm = np.asarray(posmap, dtype=np.int64)
    brk = np.flatnonzero(np.diff(pm) != 1) + 1
    rel = np.concatenate(([0], brk)).astype(np.uint32)
    return rel, pm[rel].copy()


class PosSegments:
    """Host view of one haplotype's position map; mirrors the two dicts of the reference
    (``Haplotype.posmap`` / ``posmap_rev``) without materialising them."""

    __slots__ = ("rel", "gen", "length")

    def __init__(self, rel: np.ndarray, gen: np.ndarray, length: int):
        self.rel = np.asarray(rel, dtype=np.uint32)
        self.gen = np.asarray(gen, dtype=np.int64)
        self.length = int(length)

    @classmethod
    def identity(cls, start: int, length: int) -> "PosSegments":
        return cls(np.zeros(1, np.uint32), np.array([start], np.int64), length)

    def lookup(self, rel) -> np.ndarray:
        r = np.asarray(rel, dtype=np.int64)
        k = np.searchsorted(self.rel, r, side="right") - 1
        return self.gen[k] + (r - self.rel[k].astype(np.int64))

    def full(self) -> np.ndarray:
        return self.lookup(np.arange(self.length))

    def seg_last_gen(self) -> np.ndarray:
        ends = np.concatenate((self.rel[1:].astype(np.int64), [self.length]))
        return self.gen + (ends - self.rel.astype(np.int64)) - 1

    def max_gen(self) -> int:
        return int(self.seg_last_gen().max())

    def rev(self, g: int) -> int:
        """posmap_rev[g]: the LAST relative position mapping to g (the reference rebuilds the
        reverse dict by overwrite, haplotype.py:159), or -1 when g is absent (deleted)."""
        last = self.seg_last_gen()
        k = np.flatnonzero((self.gen <= g) & (g <= last))
        if len(k) == 0:
            return -1
        k = int(k[-1])
        return int(self.rel[k]) + int(g - self.gen[k])


@dataclass
class HostHaplotype:
    """What the device search needs to know about one haplotype."""
    seq: object  # cased ASCII (str, bytes or uint8 array): lower case = base introduced by a variant
    seg: PosSegments
    is_ref: bool
    scan: Tuple[int, int]


class DeviceHapSet:
    @classmethod
    def from_handle(cls, handle, hap_len: np.ndarray, device: Optional[int] = None) -> "DeviceHapSet":
        """Wrap planes that already exist in HBM (hawk_hapset_expand); set_meta() must follow."""
        self = cls.__new__(cls)
        self._L = _lib.lib()
        self._ctx = _lib.context(device)
        self.device = device
        self._h = handle
        self.hap_len = np.asarray(hap_len, dtype=np.uint32)
        self.n_hap = len(self.hap_len)
        s = C.c_uint32()
        _lib.check(self._L.hawk_hapset_stride(self._h, C.byref(s)), "hawk_hapset_stride")
        self.stride = s.value
        return self

    def __init__(self, haps: Sequence[HostHaplotype], device: Optional[int] = None):
        L = _lib.lib()
        self._ctx = _lib.context(device)
        self.device = device
        self._L = L
        self.n_hap = len(haps)
        if self.n_hap == 0:
            raise ValueError("empty haplotype set")
        self.hap_len = np.array([len(h.seq) for h in haps], dtype=np.uint32)
        self._h = C.c_void_p()
        _lib.check(L.hawk_hapset_create(self._ctx, self.n_hap, _p(self.hap_len), C.byref(self._h)), "hawk_hapset_create")
        s = C.c_uint32()
        _lib.check(L.hawk_hapset_stride(self._h, C.byref(s)), "hawk_hapset_stride")
        self.stride = s.value
        # K1: ASCII -> planes
        off = np.zeros(self.n_hap + 1, dtype=np.uint64)
        off[1:] = np.cumsum(self.hap_len.astype(np.uint64))
        blob = np.concatenate([_as_u8(h.seq) for h in haps]) if self.n_hap > 1 else np.ascontiguousarray(_as_u8(haps[0].seq))
        bad = C.c_uint64(0)
        rc = L.hawk_hapset_pack_ascii(self._h, _p(blob), _p(off), C.byref(bad))
        if rc == _lib.HAWK_E_IUPAC:
            hidx = int(np.searchsorted(off, bad.value, side="right") - 1)
            self.close()
            raise _lib.HawkStatusError(rc, "hawk_hapset_pack_ascii",
                                       f"haplotype {hidx} position {bad.value - int(off[hidx])} "
                                       f"({bytes(blob[bad.value:bad.value + 1])!r})")
        _lib.check(rc, "hawk_hapset_pack_ascii")
        self.set_meta(haps)

    def set_meta(self, haps: Sequence[HostHaplotype]) -> None:
        is_ref, ss, se, seg_off, seg_rel, seg_gen, self.ref_index = haps.meta_arrays() if hasattr(haps, "meta_arrays") else _meta_arrays(haps)
        self.is_ref = is_ref
        _lib.check(self._L.hawk_hapset_set_meta(self._h, _p(is_ref), _p(ss), _p(se), _p(seg_off), _p(seg_rel), _p(seg_gen),
                                                self.ref_index), "hawk_hapset_set_meta")

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.hawk_hapset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def planes(self) -> np.ndarray:
        """[5, n_hap, stride] uint32 copy of the planes (tests / host-side decoding)."""
        out = np.empty((5, self.n_hap, self.stride), dtype=np.uint32)
        for p in range(5):
            _lib.check(self._L.hawk_hapset_download_plane(self._h, p, _p(out[p])), "hawk_hapset_download_plane")
        return out

    def nibbles(self, h: int) -> np.ndarray:
        """The reference's encode() output for haplotype h (encoder.py:48-57), from the planes."""
        pl = self.planes()[:, h, :]
        n = int(self.hap_len[h])
        bits = ((pl[:4, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(4, -1)[:, :n].astype(np.uint8)
        return bits[0] | (bits[1] << 1) | (bits[2] << 2) | (bits[3] << 3)

    def pam_scan(self, pam_bits: int, pam_bitsrc: int, pamlen: int) -> List[Tuple[np.ndarray, np.ndarray]]:
        """pam_search() (search_guides.py:102-131): per haplotype (fwd hits, rev hits)."""
        L = self._L
        off_f = np.zeros(self.n_hap + 1, dtype=np.uint64)
        off_r = np.zeros(self.n_hap + 1, dtype=np.uint64)
        rc = L.hawk_pam_scan(self._h, C.c_uint64(pam_bits), C.c_uint64(pam_bitsrc), pamlen, None, None, C.c_uint64(0),
                             C.c_uint64(0), _p(off_f), _p(off_r))
        if rc not in (_lib.HAWK_OK, _lib.HAWK_E_CAPACITY):
            _lib.check(rc, "hawk_pam_scan")
        nf, nr = int(off_f[-1]), int(off_r[-1])
        hf = np.empty(max(nf, 1), dtype=np.uint32)
        hr = np.empty(max(nr, 1), dtype=np.uint32)
        if rc == _lib.HAWK_E_CAPACITY:
            _lib.check(L.hawk_pam_scan(self._h, C.c_uint64(pam_bits), C.c_uint64(pam_bitsrc), pamlen, _p(hf), _p(hr),
                                       C.c_uint64(len(hf)), C.c_uint64(len(hr)), _p(off_f), _p(off_r)), "hawk_pam_scan")
        return [(hf[int(off_f[h]):int(off_f[h + 1])].copy(), hr[int(off_r[h]):int(off_r[h + 1])].copy())
                for h in range(self.n_hap)]

    def search(self, pam_bits: int, pam_bitsrc: int, pamlen: int, guidelen: int, right: bool,
               cfd_mm: Optional[np.ndarray] = None, cfd_pam: Optional[np.ndarray] = None,
               download: bool = True, collapse: bool = False, cfd_na_on_ambiguous: bool = False) -> "GuideTable":
        """One fused search.  `collapse` additionally groups the rows the report merges (GuideTable.collapse)
        while the table is still in HBM.  `cfd_na_on_ambiguous`: a guide with a non-ACGT base under a CFD lookup
        scores NA instead of raising as the reference does (cfdscore.py:93-94) - for regions with N runs."""
        L = self._L
        sp = _lib.SearchParams()
        sp.pam_fwd, sp.pam_rev, sp.pamlen, sp.guidelen, sp.right = pam_bits, pam_bitsrc, pamlen, guidelen, int(bool(right))
        keep = []
        if cfd_mm is not None:
            mm = np.ascontiguousarray(cfd_mm, dtype=np.float64).reshape(20, 4, 4)
            pt = np.ascontiguousarray(cfd_pam, dtype=np.float64).reshape(16)
            keep = [mm, pt]
            sp.score_cfdon, sp.cfd_mm, sp.cfd_pam = (2 if cfd_na_on_ambiguous else 1), mm.ctypes.data, pt.ctypes.data
        t = C.c_void_p()
        tm = _lib.Timing()
        _lib.check(L.hawk_search(self._h, C.byref(sp), C.byref(t), C.byref(tm)), "hawk_search")
        del keep
        self.last_timing = tm
        tab = GuideTable(self, t, guidelen, pamlen, bool(right), tm)
        if collapse:
            tab.collapse()
        if download:
            tab.download()
        return tab


class ExpansionPlan:
    """hawk_xplan: the inputs of a device haplotype expansion kept in HBM; `run()` writes a fresh DeviceHapSet from them
    with device work only (the per-tile step of a whole-contig search)."""

    def __init__(self, handle, hap_len: np.ndarray, device: Optional[int] = None):
        self._L = _lib.lib()
        self._x = handle
        self.hap_len = np.asarray(hap_len, dtype=np.uint32)
        self.n_hap = len(self.hap_len)
        self.device = device
        self.ref_index = -1
        self.is_ref = None

    def set_meta(self, haps: Sequence["HostHaplotype"]) -> None:
        is_ref, ss, se, seg_off, seg_rel, seg_gen, ref_index = haps.meta_arrays() if hasattr(haps, "meta_arrays") else _meta_arrays(haps)
        _lib.check(self._L.hawk_xplan_set_meta(self._x, _p(is_ref), _p(ss), _p(se), _p(seg_off), _p(seg_rel), _p(seg_gen), ref_index),
                   "hawk_xplan_set_meta")
        self.ref_index, self.is_ref = ref_index, is_ref

    def run(self, want_hash: bool = False, timed: bool = False):
        """-> (DeviceHapSet, hashes[n_hap, 2] or None, kernel ms or None)"""
        handle = C.c_void_p()
        hashes = np.zeros((self.n_hap, 2), dtype=np.uint64) if want_hash else None
        ms = C.c_float(0)
        _lib.check(self._L.hawk_xplan_run(self._x, C.byref(handle), _p(hashes), C.byref(ms) if (timed or want_hash) else None),
                   "hawk_xplan_run")
        ds = DeviceHapSet.from_handle(handle, self.hap_len, self.device)
        ds.ref_index, ds.is_ref = self.ref_index, self.is_ref
        return ds, hashes, (ms.value if (timed or want_hash) else None)

    def view(self) -> "DeviceHapSet":
        """hawk_xplan_view: the plan's rows as a set WITHOUT planes.  `search()` on it runs encode + search in one step from
        REF + the rows' variant records (hawk_vsearch.hip) and returns the same table as on `run()`'s set; whatever reads
        planes (pam_scan, planes(), the off-target scan) is refused.  Cached: one view per plan; it keeps the plan alive."""
        v = getattr(self, "_view", None)
        if v is None or v._h is None:
            handle = C.c_void_p()
            _lib.check(self._L.hawk_xplan_view(self._x, C.byref(handle)), "hawk_xplan_view")
            v = DeviceHapSet.from_handle(handle, self.hap_len, self.device)
            v.ref_index, v.is_ref = self.ref_index, self.is_ref
            v._plan_ref = self
            for k in ("alias", "host_meta"):
                if hasattr(self, k):
                    setattr(v, k, getattr(self, k))
            self._view = v
        return v

    def rebuild_dictionary(self) -> None:
        """hawk_xplan_cluster_rebuild: the cluster dictionary built again (bench.py's timed step; the result is the same)"""
        _lib.check(self._L.hawk_xplan_cluster_rebuild(self._x), "hawk_xplan_cluster_rebuild")

    def cluster_stats(self) -> dict:
        """hawk_xplan_cluster_stats: the cluster dictionary the first view() built - how many cluster instances the rows hold,
        how many distinct clusters those are, and whether searches of the view run per distinct cluster."""
        u, ni, nd, st = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        slots, ms = C.c_uint64(), C.c_float()
        _lib.check(self._L.hawk_xplan_cluster_stats(self._x, C.byref(u), C.byref(ni), C.byref(nd), C.byref(slots), C.byref(ms), C.byref(st)),
                   "hawk_xplan_cluster_stats")
        return {"usable": bool(u.value), "instances": ni.value, "distinct": nd.value, "template_slots": slots.value, "build_ms": ms.value,
                "status": st.value}

    def close(self) -> None:
        v = getattr(self, "_view", None)
        if v is not None:
            v._plan_ref = None
            v.close()  # the view reads the plan's buffers: it goes first
            self._view = None
        if getattr(self, "_x", None):
            self._L.hawk_xplan_destroy(self._x)
            self._x = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _meta_arrays(haps: Sequence["HostHaplotype"]):
    n = len(haps)
    is_ref = np.array([1 if h.is_ref else 0 for h in haps], dtype=np.uint8)
    ss = np.array([h.scan[0] for h in haps], dtype=np.int32)
    se = np.array([h.scan[1] for h in haps], dtype=np.int32)
    seg_off = np.zeros(n + 1, dtype=np.uint32)
    seg_off[1:] = np.cumsum([len(h.seg.rel) for h in haps])
    seg_rel = np.concatenate([h.seg.rel for h in haps]).astype(np.uint32)
    seg_gen = np.concatenate([h.seg.gen for h in haps]).astype(np.int64)
    refs = np.flatnonzero(is_ref)
    return is_ref, ss, se, seg_off, seg_rel, seg_gen, (int(refs[0]) if len(refs) else -1)


class GroupTable:
    """Report groups of one collapsed guide table: representative rows as columns + CSR member haplotypes."""

    def __init__(self, guidelen: int, pamlen: int, right: bool, n_groups: int, n_rows: int, device: Optional[int] = None):
        self.guidelen, self.pamlen, self.right, self.n_groups, self.n_rows = guidelen, pamlen, right, n_groups, n_rows
        ng = n_groups

        def pe(n, dt):  # page-locked once an array reaches 1 MB (by the table's own device context): the export's copies then run at link speed, unstaged
            return _lib.pinned_empty(n, dt, device)
        self.rep_row = pe(ng, np.uint32); self.pos = pe(ng, np.uint32); self.strand = np.empty(ng, np.uint8)
        self.start = pe(ng, np.int64); self.stop = pe(ng, np.int64); self.flags = np.empty(ng, np.uint8)
        self.cfdon = pe(ng, np.float64); self.win = pe(5 * ng, np.uint64).reshape(5, ng)
        self.member_hap = pe(n_rows, np.uint32)  # C3: 112 MB, the bulk of the export
        self.member_off = np.zeros(ng + 1, np.int64)
        self.gc_num = np.zeros(ng, np.uint8); self.gc_den = np.zeros(ng, np.uint8)

    @property
    def window_len(self) -> int:
        return self.guidelen + self.pamlen + 2 * GUIDESEQPAD

    def windows(self, rows: Optional[np.ndarray] = None) -> List[str]:
        return decode_windows(self.win if rows is None else self.win[:, rows], self.window_len)


def decode_windows(win: np.ndarray, W: int) -> List[str]:
    """[5, n] window slices -> cased strings (bit 0 = leftmost base)."""
    n = win.shape[1]
    if n == 0:
        return []
    sh = np.arange(W, dtype=np.uint64)
    code = np.zeros((n, W), dtype=np.uint8)
    for p in range(5):
        code |= (((win[p][:, None] >> sh) & np.uint64(1)).astype(np.uint8) << p)
    raw = _CODE2CHAR[code].tobytes()
    return [raw[i * W:(i + 1) * W].decode("ascii") for i in range(n)]


class GuideTable:
    """Columnar guide table of one fused search (rows in (haplotype, tile, strand, position) order)."""

    def __init__(self, hs: DeviceHapSet, handle, guidelen: int, pamlen: int, right: bool, timing):
        self._hs, self._t = hs, handle
        self.guidelen, self.pamlen, self.right = guidelen, pamlen, right
        self._timing_struct = timing
        self.timing = {k: getattr(timing, k) for k, _ in timing._fields_}
        n, c, h = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _lib.check(hs._L.hawk_table_counts(handle, C.byref(n), C.byref(c), C.byref(h)), "hawk_table_counts")
        self.n_rows, self.n_candidates, self.n_hits = n.value, c.value, h.value
        self._downloaded = False
        lay, sp0 = C.c_uint32(), C.c_int64()
        _lib.check(hs._L.hawk_table_layout(handle, C.byref(lay), C.byref(sp0)), "hawk_table_layout")
        self._layout = "rows" if lay.value == 1 else "columns"

    def layout(self) -> str:
        """'columns' (plane kernels, per-word search of a view) or 'rows' (the cluster search's packed 64-byte rows, include/hawk.h);
        download() hands out columns either way."""
        return self._layout

    def download_rows(self) -> np.ndarray:
        """the packed rows as they lie in HBM: [n_rows, 16] uint32 (hawk_table_download_rows; layout 'rows' only)"""
        out = np.empty((self.n_rows, 16), np.uint32)
        _lib.check(self._hs._L.hawk_table_download_rows(self._t, _p(out)), "hawk_table_download_rows")
        return out

    def close(self) -> None:
        if self._t:
            self._hs._L.hawk_table_destroy(self._t)
            self._t = None

    def export_to(self, hap: int, pos: int, strand: int, start: int, stop: int, flags: int, cfdon: int, win: int) -> None:
        """Copy the columns into caller-owned buffers given as raw addresses (host or device, e.g.
        ``tensor.data_ptr()`` of the send buffers of a collective); 0 skips a column."""
        ptr = lambda a: C.c_void_p(a) if a else None
        _lib.check(self._hs._L.hawk_table_download(self._t, ptr(hap), ptr(pos), ptr(strand), ptr(start), ptr(stop),
                                                   ptr(flags), ptr(cfdon), ptr(win)), "hawk_table_download")

    def collapse(self, flank: Tuple[int, int] = (0, 0), download_perm: bool = True, download: bool = True) -> "GuideTable":
        """Group the rows the guide report merges (reports.py:958-1008) on the device, before download():
        `group_perm` (row indices ordered by (start, strand, group)), `group_off` (CSR into it),
        `gc_num / gc_den` per group (gc_content of the spacer, annotation.py:513-541), `collapse_ms`.
        `flank` = (4, 3) widens the compared sequence to the model scorers' k-mer (scoring.py:50-67): needed when
        Azimuth / DeepCpf1 columns are on, because the reference groups by the score columns too."""
        if self._t is None:
            raise RuntimeError("collapse() needs the device-resident table: call it before download()")
        ng, ms = C.c_uint64(), C.c_float()
        _lib.check(self._hs._L.hawk_table_collapse_ex(self._t, int(flank[0]), int(flank[1]), C.byref(ng), C.byref(ms)),
                   "hawk_table_collapse_ex")
        self.n_groups, self.collapse_ms, self.collapse_flank = ng.value, ms.value, (int(flank[0]), int(flank[1]))
        self.group_perm = self.group_off = self.gc_num = self.gc_den = None
        if download:  # (`download=False`: the results stay in HBM until export_groups() / collapse_results() asks for them)
            self.collapse_results(download_perm)
        return self

    def collapse_results(self, download_perm: bool = False) -> "GuideTable":
        """hawk_table_collapse_download: the CSR offsets and G/C counts of the groups (and the row permutation) to the host."""
        self.group_perm = np.empty(self.n_rows, np.uint32) if download_perm else None
        self.group_off = _lib.pinned_empty(self.n_groups + 1, np.uint64, getattr(self._hs, 'device', None))
        self.group_off[-1:] = 0
        self.gc_num = np.empty(self.n_groups, np.uint8)
        self.gc_den = np.empty(self.n_groups, np.uint8)
        _lib.check(self._hs._L.hawk_table_collapse_download(self._t, _p(self.group_perm), _p(self.group_off), _p(self.gc_num),
                                                            _p(self.gc_den)), "hawk_table_collapse_download")
        return self

    def export_groups(self) -> "GroupTable":
        """After collapse(): the group-level view the report is assembled from - one representative row per group (its
        first member in table order) and every row's haplotype in group order - without downloading the table."""
        if self._t is None or not hasattr(self, "n_groups"):
            raise RuntimeError("export_groups() needs collapse() on the device-resident table")
        ng, n = self.n_groups, self.n_rows
        g = GroupTable(self.guidelen, self.pamlen, self.right, ng, n, getattr(self._hs, 'device', None))
        ms = C.c_float()
        _lib.check(self._hs._L.hawk_table_collapse_export(self._t, _p(g.rep_row), _p(g.pos), _p(g.strand), _p(g.start), _p(g.stop),
                                                          _p(g.flags), _p(g.cfdon), _p(g.win), _p(g.member_hap), C.byref(ms)),
                   "hawk_table_collapse_export")
        if self.group_off is None:
            self.collapse_results(False)
        g.member_off = self.group_off.astype(np.int64)
        g.gc_num, g.gc_den = self.gc_num, self.gc_den
        g.export_ms = ms.value
        g.n_candidates, g.n_hits = self.n_candidates, self.n_hits
        return g

    def download(self) -> "GuideTable":
        if self._downloaded:
            return self
        n = self.n_rows
        self.hap = np.empty(n, np.uint32); self.pos = np.empty(n, np.uint32); self.strand = np.empty(n, np.uint8)
        self.start = np.empty(n, np.int64); self.stop = np.empty(n, np.int64); self.flags = np.empty(n, np.uint8)
        self.cfdon = np.empty(n, np.float64); self.win = np.empty((5, n), np.uint64)
        _lib.check(self._hs._L.hawk_table_download(self._t, _p(self.hap), _p(self.pos), _p(self.strand), _p(self.start),
                                                   _p(self.stop), _p(self.flags), _p(self.cfdon), _p(self.win)),
                   "hawk_table_download")
        self._hs._L.hawk_table_destroy(self._t)
        self._t = None
        self._downloaded = True
        return self

    @property
    def window_len(self) -> int:
        return self.guidelen + self.pamlen + 2 * GUIDESEQPAD

    def right_as_stored(self) -> np.ndarray:
        """`right` of each row as search() stores it: flipped for strand 1 (search_guides.py:538)."""
        return (self.strand.astype(bool)) ^ self.right

    def windows(self, rows: Optional[np.ndarray] = None) -> List[str]:
        """Guide.sequence of each row: the guidelen+pamlen+20-nt window on the + strand, case
        preserved (search_guides.py:134-160)."""
        return decode_windows(self.win if rows is None else self.win[:, rows], self.window_len)

    def emission_order(self) -> np.ndarray:
        """Row permutation giving the reference's pre-dedup emission order: haplotype, then strand
        (0 before 1), then ascending PAM position (search_guides.py:530-547).  The device writes rows
        per 32 768-position tile, so the strands of one haplotype interleave tile by tile."""
        return np.lexsort((self.pos, self.strand, self.hap))

    def reference_order(self) -> np.ndarray:
        """Row permutation reproducing the list order of the reference's search():
        remove_redundant_guides (search_guides.py:340-369) walks a dict keyed by
        f"{start}_{strand}" in first-seen order, each group in emission order."""
        n = self.n_rows
        if n == 0:
            return np.zeros(0, dtype=np.int64)
        emit = self.emission_order()
        key = self.start[emit].astype(np.int64) * 2 + self.strand[emit]
        _, inv = np.unique(key, return_inverse=True)
        first = np.full(inv.max() + 1, n, dtype=np.int64)
        np.minimum.at(first, inv, np.arange(n))
        return emit[np.lexsort((np.arange(n), first[inv]))]
