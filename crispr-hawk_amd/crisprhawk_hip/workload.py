"""Haplotype-set construction for a synthetic (or VCF-derived) phased panel.

Mirrors the reference's reconstruct_haplotypes() phased branch (haplotypes.py:132-159,
297-368, 232-294): REF first, then per sample chromosome copy 0 / copy 1 (one entry with
``1|1`` when both copies give the same sequence), identical cased sequences collapsed keeping
the first member's position map.  Used by bench.py and the tests to build the resident input of
the search kernels; the search itself is hawk_search().
"""

import hashlib
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .expand import expand_haplotype, scan_bounds, scan_start, scan_stop
from .hapset import HostHaplotype, PosSegments
from .synth import SynthRegion


class HapInfo:
    """Host-side labels of one device haplotype (what Guide rows inherit from it).  `variant_idx` may be given as a callable:
    the carried-variant lists of a device-built plan stay in HBM until a label is asked for (CarriedLists)."""

    __slots__ = ("samples", "_v")

    def __init__(self, samples: List[str], variant_idx):
        self.samples = samples
        self._v = variant_idx

    @property
    def variant_idx(self):
        if callable(self._v):
            self._v = self._v()
        return self._v


class CarriedLists:
    """The carried-variant lists of a genotype inversion (hawk_gt_lists), left on the device: `idx` downloads the variant
    indices once, when something on the host (a report label) first asks, and releases the device object."""

    def __init__(self, g, n_entries: int):
        self._g, self._n, self._idx = g, int(n_entries), None

    @property
    def idx(self) -> np.ndarray:
        if self._idx is None:
            import ctypes as C
            from . import _lib
            from .hapset import _p
            out = np.zeros(max(self._n, 1), dtype=np.uint32)
            _lib.check(_lib.lib().hawk_gt_lists_download(self._g, _p(out), None), "hawk_gt_lists_download")
            self._idx = out[:self._n]
            self.close()
        return self._idx

    def rows(self, a: int, b: int):
        return lambda: self.idx[a:b]

    def close(self) -> None:
        if self._g is not None:
            from . import _lib
            _lib.lib().hawk_gt_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _digest(arr: np.ndarray) -> bytes:
    return hashlib.blake2b(arr.tobytes(), digest_size=16).digest()


def build_phased_haplotypes(reg: SynthRegion, pamlen: int, max_haplotypes: Optional[int] = None,
                            sample_slice: Optional[slice] = None) -> Tuple[List[HostHaplotype], List[HapInfo]]:
    ref = np.frombuffer(reg.sequence.encode("ascii"), dtype=np.uint8)
    startp, stopp = reg.startp, reg.stopp
    haps: List[HostHaplotype] = [HostHaplotype(ref, PosSegments.identity(startp, len(ref)), True,
                                               scan_bounds(PosSegments.identity(startp, len(ref)), startp, stopp, pamlen))]
    info: List[HapInfo] = [HapInfo(["REF"], ())]
    if not reg.variants:
        return haps, info
    site = [(v.pos, v.ref.encode(), v.alt.encode()) for v in reg.variants]
    G = np.stack([v.gt.reshape(-1) for v in reg.variants])  # [site, 2*sample]
    index: Dict[bytes, int] = {_digest(ref): 0}
    samples = range(len(reg.samples))
    if sample_slice is not None:
        samples = samples[sample_slice]
    for si in samples:
        name = reg.samples[si]
        built = []
        for c in (0, 1):
            idx = np.flatnonzero(G[:, 2 * si + c])
            if len(idx) == 0:
                built.append((ref, PosSegments.identity(startp, len(ref)), ()))
            else:
                arr, seg = expand_haplotype(ref, startp, [site[k] for k in idx])
                built.append((arr, seg, tuple(int(k) for k in idx)))
        if not built[0][2] and not built[1][2]:
            continue  # sample carries no variant in this region (haplotypes.py:157)
        homo = len(built[0][0]) == len(built[1][0]) and np.array_equal(built[0][0], built[1][0])
        entries = [(built[0], f"{name}:1|1")] if homo else [(built[0], f"{name}:1|0"), (built[1], f"{name}:0|1")]
        for (arr, seg, vidx), label in entries:
            d = _digest(arr)
            j = index.get(d)
            if j is not None:
                if j != 0:  # collapsing onto REF keeps samples == "REF" (haplotypes.py:255-258)
                    info[j].samples.append(label)
                continue
            index[d] = len(haps)
            haps.append(HostHaplotype(arr, seg, False, scan_bounds(seg, startp, stopp, pamlen)))
            info.append(HapInfo([label], vidx))
            if max_haplotypes is not None and len(haps) >= max_haplotypes:
                return haps, info
    return haps, info


# ---------------------------------------------------------------------------------------------
# the same haplotype set, expanded on the device
# ---------------------------------------------------------------------------------------------
_NIB = np.zeros(256, dtype=np.uint8)
for _c, _v in {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3, "B": 14, "D": 13,
               "H": 11, "V": 7}.items():
    _NIB[ord(_c)] = _v
    _NIB[ord(_c.lower())] = _v


def _variant_table(pos: np.ndarray, refs: List[str], alts: List[str], seq: str, startp: int):
    """Per-variant arrays the expansion kernels index with (validated against the reference string)."""
    from .expand import HaplotypeBuildError
    n_ref = len(seq)
    nv = len(refs)
    r0 = np.asarray(pos, dtype=np.int64) - startp
    reflen = np.array([len(x) for x in refs], dtype=np.int64)
    altlen = np.array([len(x) for x in alts], dtype=np.int64)
    if np.any(np.diff(r0) < 0):
        raise HaplotypeBuildError("variants must be sorted by position")
    # replaced span as the reference computes it (haplotype.py:197-201): |chain|+1 for deletions, else 1
    chain = altlen - reflen
    span = np.where(chain < 0, -chain + 1, 1)
    if np.any((chain < 0) & (altlen != 1)) or np.any((chain == 0) & (reflen != 1)):
        raise HaplotypeBuildError("device expansion handles SNVs, deletions (alt of one base) and insertions")
    if np.any(r0 < 0) or np.any(r0 + span > n_ref):
        raise HaplotypeBuildError("variant outside the region")
    # REF allele must match (haplotype.py:203-208): its first base for all records at once, the rest of a deletion's
    # span (the few records with span > 1) one by one
    if nv:
        seq_u8 = np.frombuffer(seq.encode("ascii"), dtype=np.uint8) if isinstance(seq, str) else np.asarray(seq, dtype=np.uint8)
        first = np.frombuffer("".join(x[:1] or "\0" for x in refs).encode("ascii"), dtype=np.uint8)
        bad = np.flatnonzero(((seq_u8[r0] ^ first) & 0xDF) != 0)  # case-insensitive (letters only)
        if len(bad):
            raise HaplotypeBuildError(f"Mismatching reference alleles at position {int(pos[int(bad[0])])}")
        for i in np.flatnonzero(span > 1).tolist():
            a, sp = int(r0[i]), int(span[i])
            if len(refs[i]) < sp or seq[a:a + sp].upper() != refs[i][:sp].upper():
                raise HaplotypeBuildError(f"Mismatching reference alleles at position {int(pos[i])}")
    alt_blob = "".join(alts).encode("ascii")
    alt_codes = _NIB[np.frombuffer(alt_blob, dtype=np.uint8)] if alt_blob else np.zeros(0, np.uint8)
    alt_off = np.zeros(nv, dtype=np.int64)
    if nv:
        alt_off[1:] = np.cumsum(altlen)[:-1]
    return r0, span, chain, altlen, alt_off, alt_codes


def _variant_table_columns(cols, seq, startp: int):
    """_variant_table from records held as columns (SynthRegion.variant_columns): the same arrays and the same checks,
    without a Python step per record."""
    from .expand import HaplotypeBuildError
    n_ref = len(seq)
    pos = np.asarray(cols["pos"], dtype=np.int64)
    nv = len(pos)
    ref_off, alt_off_all = np.asarray(cols["ref_off"], dtype=np.int64), np.asarray(cols["alt_off"], dtype=np.int64)
    reflen, altlen = np.diff(ref_off), np.diff(alt_off_all)
    r0 = pos - startp
    if np.any(np.diff(r0) < 0):
        raise HaplotypeBuildError("variants must be sorted by position")
    chain = altlen - reflen
    span = np.where(chain < 0, -chain + 1, 1)
    if np.any((chain < 0) & (altlen != 1)) or np.any((chain == 0) & (reflen != 1)):
        raise HaplotypeBuildError("device expansion handles SNVs, deletions (alt of one base) and insertions")
    if np.any(r0 < 0) or np.any(r0 + span > n_ref):
        raise HaplotypeBuildError("variant outside the region")
    if nv:
        seq_u8 = np.frombuffer(seq.encode("ascii"), dtype=np.uint8) if isinstance(seq, str) else np.asarray(seq, dtype=np.uint8)
        ref_blob = np.asarray(cols["ref_blob"], dtype=np.uint8)
        if np.any(reflen < span):
            i = int(np.flatnonzero(reflen < span)[0])
            raise HaplotypeBuildError(f"Mismatching reference alleles at position {int(pos[i])}")
        # REF allele must match over the replaced span (haplotype.py:203-208), letters compared case-insensitively
        rec = np.repeat(np.arange(nv), span)
        k = np.arange(len(rec)) - np.repeat(np.cumsum(span) - span, span)
        bad = np.flatnonzero(((seq_u8[r0[rec] + k] ^ ref_blob[ref_off[rec] + k]) & 0xDF) != 0)
        if len(bad):
            raise HaplotypeBuildError(f"Mismatching reference alleles at position {int(pos[int(rec[int(bad[0])])])}")
    alt_blob = np.asarray(cols["alt_blob"], dtype=np.uint8)
    alt_codes = _NIB[alt_blob] if len(alt_blob) else np.zeros(0, np.uint8)
    return r0, span, chain, altlen, alt_off_all[:-1].copy(), alt_codes


class ScanOwnership:
    """Which PAM positions of a haplotype row this set scans, for a row string that is one tile of a larger region
    (tiling.py).  `own_lo` / `own_hi`: genomic positions of the seams (None = the region's own rule at that end,
    compute_scan_start_stop, search_guides.py:49-84); a seam maps to the LAST relative position with that genomic
    position (the reference's posmap_rev rule), walking forward when the position is deleted - the same rule in both
    neighbouring tiles, so every haplotype position is scanned by exactly one of them.  `partner` = the region's scan
    range in the REF row's relative positions (hawk_hapset_set_ref_partner_range).  `guard`: haplotype bases that must
    exist on either side of an interior seam so that is_pamhit_in_range (search_guides.py:395-420) cannot fire there."""

    def __init__(self, own_lo: Optional[int], own_hi: Optional[int], partner: Optional[Tuple[int, int]], guard: int):
        self.own_lo, self.own_hi, self.partner, self.guard = own_lo, own_hi, partner, guard


def _seam_rel(seg: PosSegments, g: int) -> int:
    r = seg.rev(g)
    if r >= 0:
        return r
    upper = seg.max_gen()
    for p in range(g + 1, upper + 1):
        r = seg.rev(p)
        if r >= 0:
            return r
    return seg.length


def _scan_for(seg: PosSegments, startp: int, stopp: int, pamlen: int, own: Optional[ScanOwnership]) -> Tuple[int, int]:
    if own is None:
        return scan_bounds(seg, startp, stopp, pamlen)
    if own.own_lo is None:
        lo = scan_start(seg, startp)
    if own.own_hi is None:
        hi = scan_stop(seg, stopp, pamlen)
    if own.own_lo is not None:
        lo = _seam_rel(seg, own.own_lo)
        if lo < own.guard:
            raise ValueError("tile flank too small: a guide window at the seam would leave the tile (left)")
    if own.own_hi is not None:
        hi = _seam_rel(seg, own.own_hi)
        if hi + own.guard > seg.length:
            raise ValueError("tile flank too small: a guide window at the seam would leave the tile (right)")
    return lo, max(lo, hi)


class RowMeta:
    """Position maps and scan ranges of the rows of an expanded set, as flat arrays (CSR segments) - what
    hawk_hapset_set_meta takes - with list-like access to per-row `HostHaplotype` views for the few callers that want one
    (labels keep `host_meta[r].seg`).  Replaces 5009 Python objects and as many numpy calls per expansion."""

    def __init__(self, seg_start, seg_rel, seg_gen, hap_len: np.ndarray, alias: np.ndarray, startp: int, fetch=None, rev=None):
        """`fetch`: a callable returning (seg_start, seg_rel, seg_gen) - the segments of a device-built plan are downloaded
        only when something on the host reads them; `rev`: {genomic position: posmap_rev of every row}, computed on the
        device beside the segments."""
        self._segs = None if fetch is not None else (seg_start, seg_rel, seg_gen)
        self._fetch, self._rev = fetch, dict(rev or {})
        self.hap_len, self.alias, self.startp = np.asarray(hap_len, dtype=np.int64), alias, startp
        self.n = len(hap_len)
        self.scan_lo = np.zeros(self.n, dtype=np.int64)
        self.scan_hi = np.zeros(self.n, dtype=np.int64)

    def _need(self):
        if self._segs is None:
            self._segs = self._fetch()
        return self._segs

    seg_start = property(lambda self: self._need()[0])
    seg_rel = property(lambda self: self._need()[1])
    seg_gen = property(lambda self: self._need()[2])

    def __len__(self):
        return self.n

    def seg(self, r: int) -> PosSegments:
        a, b = int(self.seg_start[r]), int(self.seg_start[r + 1])
        return PosSegments(self.seg_rel[a:b], self.seg_gen[a:b], int(self.hap_len[r]))

    def __getitem__(self, r: int) -> HostHaplotype:
        return HostHaplotype(b"", self.seg(r), r == 0, (int(self.scan_lo[r]), int(self.scan_hi[r])))

    def __iter__(self):
        return (self[r] for r in range(self.n))

    def _rev_all(self, g: int) -> np.ndarray:
        """posmap_rev[g] of every row at once: the last relative position whose genomic position is g, -1 where g is
        deleted (the reference rebuilds the reverse dict by overwrite, haplotype.py:159).  One pass over the segments in
        the library's host helper; `_rev_all_numpy` is the same in numpy (cross-check of tests/test_host_logic.py)."""
        import ctypes as C
        from . import _lib
        from .hapset import _p
        from .reports import _host_lib
        if int(g) in self._rev:
            return self._rev[int(g)]
        ss = np.ascontiguousarray(self.seg_start, dtype=np.uint64)
        sr = np.ascontiguousarray(self.seg_rel, dtype=np.uint32)
        sg = np.ascontiguousarray(self.seg_gen, dtype=np.int64)
        hl = np.ascontiguousarray(self.hap_len, dtype=np.uint32)
        out = np.empty(self.n, dtype=np.int64)
        _lib.check(_host_lib().hawk_host_posmap_rev(_p(ss), _p(sr), _p(sg), _p(hl), C.c_uint32(self.n), C.c_int64(int(g)), _p(out)),
                   "hawk_host_posmap_rev")
        return out

    def _rev_all_numpy(self, g: int) -> np.ndarray:
        ends = np.empty(len(self.seg_rel), dtype=np.int64)
        ends[:-1] = self.seg_rel[1:]
        ends[self.seg_start[1:] - 1] = self.hap_len          # a row's last segment runs to the row's end
        last_gen = self.seg_gen + (ends - self.seg_rel.astype(np.int64)) - 1
        hit = (self.seg_gen <= g) & (g <= last_gen)
        k = np.where(hit, np.arange(len(hit)), -1)
        lastk = np.maximum.reduceat(k, self.seg_start[:-1])
        rel = self.seg_rel[np.maximum(lastk, 0)].astype(np.int64) + (g - self.seg_gen[np.maximum(lastk, 0)])
        return np.where(lastk >= 0, rel, -1)

    def compute_scans(self, startp: int, stopp: int, pamlen: int, own: Optional["ScanOwnership"]) -> None:
        """compute_scan_start_stop (search_guides.py:49-84), or a tile's ownership range, for every live row; rows whose
        boundary position is deleted (rare) take the per-row walk of `_scan_for`."""
        live = self.alias == np.arange(self.n)
        lo_g = startp + 100 if (own is None or own.own_lo is None) else own.own_lo
        hi_g = stopp - 100 if (own is None or own.own_hi is None) else own.own_hi
        lo, hi = self._rev_all(lo_g), self._rev_all(hi_g)
        if own is None or own.own_hi is None:
            hi = np.where(hi >= 0, hi - pamlen + 1, hi)
        slow = live & ((lo < 0) | (hi < 0))
        if own is not None and own.own_lo is not None and np.any(live & (lo >= 0) & (lo < own.guard)):
            raise ValueError("tile flank too small: a guide window at the seam would leave the tile (left)")
        if own is not None and own.own_hi is not None and np.any(live & (hi >= 0) & (hi + own.guard > self.hap_len)):
            raise ValueError("tile flank too small: a guide window at the seam would leave the tile (right)")
        self.scan_lo = np.where(live, lo, 0)
        self.scan_hi = np.where(live, hi if own is None else np.maximum(lo, hi), 0)
        for r in np.flatnonzero(slow).tolist():
            a, b = _scan_for(self.seg(r), startp, stopp, pamlen, own)
            self.scan_lo[r], self.scan_hi[r] = a, b

    def meta_arrays(self):
        is_ref = np.zeros(self.n, dtype=np.uint8)
        is_ref[0] = 1
        return (is_ref, self.scan_lo.astype(np.int32), self.scan_hi.astype(np.int32), self.seg_start.astype(np.uint32),
                np.ascontiguousarray(self.seg_rel, dtype=np.uint32), np.ascontiguousarray(self.seg_gen, dtype=np.int64), 0)


def build_segments(ind, hv_idx, hv_o, hv_off, r0, chain, startp: int, hap_len, alias):
    """Position-map segments of all rows of an expansion as CSR arrays (seg_start[n + 1], seg_rel u32, seg_gen i64) from the
    carried indels `ind` (entry indices into hv_idx / hv_o, ascending): the library's host helper
    (hawk_host_build_segments; one pass over the ~10 % of list entries that are indels).  `build_segments_numpy` is the
    same in numpy - what ran before, kept as the cross-check of tests/test_host_logic.py."""
    import ctypes as C
    from . import _lib
    from .hapset import _p
    from .reports import _host_lib
    L = _host_lib()  # libhawk_hip.so, or the sanitizer build of the host helpers (HAWK_HOSTUTIL_LIB)
    n = len(hap_len)
    ind = np.ascontiguousarray(ind, dtype=np.uint32)
    hv_idx = np.ascontiguousarray(hv_idx, dtype=np.uint32)
    hv_o = np.ascontiguousarray(hv_o, dtype=np.int32)
    hv_off = np.ascontiguousarray(hv_off, dtype=np.uint64)
    r0_, ch_ = np.ascontiguousarray(r0, dtype=np.int64), np.ascontiguousarray(chain, dtype=np.int64)
    hl, al = np.ascontiguousarray(hap_len, dtype=np.uint32), np.ascontiguousarray(alias, dtype=np.int64)
    seg_start = np.zeros(n + 1, dtype=np.uint64)
    args = (_p(ind), C.c_uint64(len(ind)), _p(hv_idx), _p(hv_o), _p(hv_off), C.c_uint32(n), _p(r0_), _p(ch_), C.c_int64(int(startp)), _p(hl), _p(al),
            _p(seg_start))
    _lib.check(L.hawk_host_build_segments(*args, None, None, C.c_uint64(0)), "hawk_host_build_segments")
    tot = int(seg_start[-1])
    seg_rel = np.zeros(tot, dtype=np.uint32)
    seg_gen = np.zeros(tot, dtype=np.int64)
    _lib.check(L.hawk_host_build_segments(*args, _p(seg_rel), _p(seg_gen), C.c_uint64(tot)), "hawk_host_build_segments")
    return seg_start.astype(np.int64), seg_rel, seg_gen


def build_segments_numpy(ind, hv_idx, hv_o, hv_off, r0, chain, startp: int, hap_len, alias):
    """build_segments in numpy (all rows at once): every carried deletion opens one segment behind it, every carried
    insertion of n bases opens n + 1 (the inserted bases all map to the anchor position, haplotype.py:106-159)."""
    n_hap = len(hap_len)
    ind = np.asarray(ind, dtype=np.int64)
    o_i = hv_o[ind].astype(np.int64)
    pos_i = r0[hv_idx[ind]] + startp
    ch_i = chain[hv_idx[ind]]
    row_i = np.searchsorted(np.asarray(hv_off[1:], dtype=np.int64), ind, side="right")  # row of the entry (row 0 is REF)
    nseg_i = np.where(ch_i < 0, 1, ch_i + 1)
    first_of = np.cumsum(nseg_i) - nseg_i
    k_in = np.arange(int(nseg_i.sum())) - np.repeat(first_of, nseg_i)  # 0..n within an insertion's run, 0 for a deletion
    base_i = np.where(ch_i < 0, pos_i + 1 - ch_i, pos_i)
    bump_at = np.where(ch_i < 0, np.iinfo(np.int64).max, ch_i)
    seg_rel_all = np.repeat(o_i + 1, nseg_i) + k_in
    seg_gen_all = np.repeat(base_i, nseg_i) + (k_in >= np.repeat(bump_at, nseg_i))
    seg_row_all = np.repeat(row_i, nseg_i)
    keep = seg_rel_all < np.asarray(hap_len)[seg_row_all].astype(np.int64)
    seg_rel_all, seg_gen_all, seg_row_all = seg_rel_all[keep], seg_gen_all[keep], seg_row_all[keep]
    # rows collapsed onto another keep the identity map only; every row starts with the identity segment (rel 0 ->
    # startp).  The rows' own segments come out in (row, rel) order already - list entries are ordered by row, then by
    # variant, and output positions grow with the variant - so the identity segments are slotted in, not sorted in.
    live_seg = np.asarray(alias)[seg_row_all] == seg_row_all
    seg_rel_all, seg_gen_all, seg_row_all = seg_rel_all[live_seg], seg_gen_all[live_seg], seg_row_all[live_seg]
    own_cnt = np.bincount(seg_row_all, minlength=n_hap)
    seg_start = np.concatenate(([0], np.cumsum(own_cnt + 1)))
    dst = np.arange(len(seg_rel_all)) + seg_row_all + 1    # own segment i of row r lands behind r + 1 identity segments
    rel_m = np.zeros(int(seg_start[-1]), dtype=np.int64)
    gen_m = np.full(int(seg_start[-1]), startp, dtype=np.int64)
    rel_m[dst] = seg_rel_all
    gen_m[dst] = seg_gen_all
    return seg_start, rel_m.astype(np.uint32), gen_m


class _RowInfos:
    """The HapInfo of every row of an expansion (None for rows collapsed onto another), built when first read: the sample
    labels of 5009 rows are Python strings nobody needs before a report is written."""

    def __init__(self, n_hap, samples, e_s, e_g, e_r, e_owner, rows_of):
        self._args = (n_hap, samples, e_s, e_g, e_r, e_owner, rows_of)
        self._info = None

    def _build(self):
        if self._info is None:
            n_hap, samples, e_s, e_g, e_r, e_owner, rows_of = self._args
            gts = ("1|1", "1|0", "0|1")
            labels_e = [f"{samples[si]}:{gts[g]}" for si, g in zip(e_s.tolist(), e_g.tolist())]
            info: List[Optional[HapInfo]] = [HapInfo(["REF"], ())] + [None] * (n_hap - 1)
            for lab, r, o in zip(labels_e, e_r.tolist(), e_owner.tolist()):
                if o == 0:
                    continue  # collapses onto REF, which keeps samples == "REF" (haplotypes.py:255-258)
                if info[o] is None:
                    info[o] = HapInfo([lab], rows_of(o))
                else:
                    info[o].samples.append(lab)
            self._info = info
        return self._info

    def __getitem__(self, i):
        return self._build()[i]

    def __len__(self):
        return self._args[0]

    def __iter__(self):
        return iter(self._build())


class _KeptInfos:
    """info of the kept rows, in row order (what the expansion entry points return), as lazily as _RowInfos"""

    def __init__(self, infos: _RowInfos, kept: List[int]):
        self._infos, self._kept = infos, kept

    def __getitem__(self, j):
        if isinstance(j, slice):
            return [self._infos[i] for i in self._kept[j]]
        return self._infos[self._kept[j]]

    def __len__(self):
        return len(self._kept)

    def __iter__(self):
        return (self._infos[i] for i in self._kept)


def _exact_keys(ds, hashes: np.ndarray) -> np.ndarray:
    """Rows grouped by content, exactly: rows sharing a 128-bit content hash are compared base for base on the device
    (hawk_hapset_rows_equal) against the first row of their hash group; a row that differs (a hash collision - never seen)
    opens a group of its own with the rows that equal IT.  Returns one key per row."""
    import ctypes as C
    from . import _lib
    from .hapset import _p
    h64 = hashes[:, 0] * np.uint64(0x9e3779b97f4a7c15) + (hashes[:, 1] ^ (hashes[:, 1] >> np.uint64(29)))
    _, first_idx, key_id = np.unique(h64, return_index=True, return_inverse=True)
    key_id = key_id.reshape(-1).astype(np.int64)
    n = len(key_id)
    head = first_idx[key_id]
    todo = np.flatnonzero(head != np.arange(n))  # rows that would be merged with an earlier row
    next_key = int(key_id.max()) + 1 if n else 0
    while len(todo):
        a = np.ascontiguousarray(todo, dtype=np.uint32)
        b = np.ascontiguousarray(head[todo], dtype=np.uint32)
        eq = np.zeros(len(todo), dtype=np.uint8)
        _lib.check(_lib.lib().hawk_hapset_rows_equal(ds._h, len(todo), _p(a), _p(b), _p(eq)), "hawk_hapset_rows_equal")
        bad = todo[eq == 0]
        if len(bad) == 0:
            break
        # per old group: its first mismatching row heads a new group, the other mismatching rows are tried against it
        for k in np.unique(key_id[bad]).tolist():
            rows = bad[key_id[bad] == k]
            key_id[rows] = next_key
            head[rows] = rows[0]
            next_key += 1
        todo = bad[head[bad] != bad]
    return key_id


def _collapse_rows(hashes: np.ndarray, samples: List[str], live: np.ndarray, n_hap: int, rows_of, ds=None):
    """Labels, homozygous merge and collapse by content of the rows of an expansion (haplotypes.py:232-368), all rows at
    once on their 16-byte content hashes.  `rows_of(r)` -> the carried-variant indices of row r (or a callable yielding
    them).  Returns (alias[n_hap], info[n_hap] with None for rows collapsed onto another)."""
    # entries in the reference's order (haplotypes.py:297-368): samples in panel order, copy 0 then copy 1; a sample
    # whose two copies hold the same sequence contributes one "1|1" entry; a copy without variants is the REF sequence
    # rows that hold the same bases: grouped on the 128-bit content hash and then compared on the device, base for base
    # (the reference compares the strings, haplotypes.py:274-294); without a set to compare on, the hash alone
    if ds is not None:
        key_id = _exact_keys(ds, hashes)
    else:
        _, key_id = np.unique(hashes, axis=0, return_inverse=True)
        key_id = key_id.reshape(-1)
    ns = len(samples)
    row_of_col = np.zeros(2 * ns, dtype=np.int64)          # 0 = "no row" (the copy carries nothing: it IS REF)
    row_of_col[np.asarray(live, dtype=np.int64)] = np.arange(1, n_hap)
    r_a, r_b = row_of_col[0::2], row_of_col[1::2]           # rows of copy 0 / copy 1 per sample (0: none)
    k_a, k_b = key_id[r_a], key_id[r_b]                     # row 0 is REF, so "none" reads REF's key
    present = (r_a > 0) | (r_b > 0)
    homo = present & (k_a == k_b)
    alias = np.arange(n_hap)
    alias[(key_id == key_id[0]) & (alias > 0)] = 0          # a copy whose variants reproduce the REF sequence
    both = homo & (r_a > 0) & (r_b > 0)
    # entry list: (sample, row, genotype code 0 = 1|1, 1 = 1|0, 2 = 0|1), rows that do not exist dropped
    e_s = np.concatenate((np.flatnonzero(homo), np.flatnonzero(present & ~homo), np.flatnonzero(present & ~homo)))
    e_r = np.concatenate((r_a[homo], r_a[present & ~homo], r_b[present & ~homo]))
    e_g = np.concatenate((np.zeros(int(homo.sum()), np.int64), np.ones(int((present & ~homo).sum()), np.int64),
                          np.full(int((present & ~homo).sum()), 2, np.int64)))
    ok = e_r > 0
    e_s, e_r, e_g = e_s[ok], e_r[ok], e_g[ok]
    order = np.lexsort((e_g, e_s))                          # sample order, then 1|1 / 1|0 before 0|1
    e_s, e_r, e_g = e_s[order], e_r[order], e_g[order]
    # the first row seen with a key owns it (REF owns its own key); later rows with that key alias onto the owner
    e_k = key_id[e_r]
    owner_of_key = np.full(int(key_id.max()) + 1, -1, dtype=np.int64)
    uk, first_pos = np.unique(e_k, return_index=True)
    owner_of_key[uk] = e_r[first_pos]
    owner_of_key[key_id[0]] = 0
    e_owner = owner_of_key[e_k]
    alias[e_r] = e_owner
    alias[r_b[both]] = alias[r_a[both]]                     # the second copy of a homozygous sample follows the first
    return alias, _RowInfos(n_hap, samples, e_s, e_g, e_r, e_owner, rows_of)


def _expand_rows(ref_set, seq: str, startp: int, stopp: int, pamlen: int, samples: List[str], tab, live: np.ndarray,
                 counts_live: np.ndarray, hv_idx: np.ndarray, hv_o: np.ndarray, tot_live: np.ndarray, device,
                 own: Optional[ScanOwnership] = None, keep_plan: bool = False, indel_entries: Optional[np.ndarray] = None):
    """Common tail of the device expansions: rows = REF + every chromosome copy (column) with a non-empty carried
    list, in column order.  Builds the expansion plan (hawk_xplan_create), runs it, then labels / homozygous merge /
    collapse on the 16-byte content hashes (haplotypes.py:232-368) and the position-map segments + scan bounds of every
    kept row.  With `keep_plan` the plan (inputs + metadata resident in HBM) stays attached as `ds.plan`, so the set can
    be re-expanded with device work only."""
    import ctypes as C
    from . import _lib
    from .expand import HaplotypeBuildError
    from .hapset import DeviceHapSet, ExpansionPlan, _p
    r0, span, chain, altlen, alt_off, alt_codes = tab
    n_ref, nv = len(seq), len(r0)
    n_hap = 1 + len(live)
    hv_off = np.zeros(n_hap + 1, dtype=np.uint64)
    hv_off[2:] = np.cumsum(counts_live)
    hv_idx = np.ascontiguousarray(hv_idx, dtype=np.uint32)
    hv_o = np.ascontiguousarray(hv_o, dtype=np.int32)
    hap_len = np.concatenate(([n_ref], n_ref + np.asarray(tot_live, dtype=np.int64))).astype(np.uint32)
    # the carried indels: listed by the device inversion (hawk_gt_lists_indels), else one byte-table gather over the lists
    ind = indel_entries if indel_entries is not None else np.flatnonzero((chain != 0).astype(np.uint8)[hv_idx])
    if len(ind) and own is None:
        # the reference's clamp (haplotype.py:199-201) would fire.  It compares with the REGION's original length, which a
        # tile of a larger region cannot know (the shift accumulated before the tile): tiled searches do not reproduce
        # that end-of-region error (divergences).  Only INDELS can trip it: the reference applies a copy's SNVs
        # first (haplotype.py:494-512), while the position map is still the identity.
        if np.any(hv_o[ind].astype(np.int64) + span[hv_idx[ind]] > n_ref):
            raise HaplotypeBuildError("variant beyond the original region length (haplotype.py:199-201 clamp)")
    L = _lib.lib()
    xh = C.c_void_p()
    u32 = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    arrs = [u32(r0), u32(span), u32(alt_off), u32(altlen), np.ascontiguousarray(alt_codes)]
    rc = L.hawk_xplan_create(ref_set._h, nv, _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]), len(alt_codes),
                             n_hap, _p(hv_off), _p(hv_idx), _p(hv_o), _p(hap_len), C.byref(xh))
    if rc == _lib.HAWK_E_INVALID and len(hv_idx) > 1:
        # the library validates every list it is given (ascending, non-overlapping, prefix sums consistent); say which
        # rule of the reference a refused input broke (haplotype.py:214-252 raises on overlapping variants of one copy)
        row_of = np.repeat(np.arange(len(live)), counts_live)
        same_row = row_of[1:] == row_of[:-1]
        if np.any(same_row & (r0[hv_idx[1:]] < r0[hv_idx[:-1]] + span[hv_idx[:-1]])):
            raise HaplotypeBuildError("a chromosome copy carries overlapping variants")
    _lib.check(rc, "hawk_xplan_create")
    plan = ExpansionPlan(xh, hap_len, device)
    plan.n_records = int(len(hv_idx))
    ds, hashes, ms_val = plan.run(want_hash=True)
    ms = C.c_float(ms_val)
    hv_off_i = hv_off.astype(np.int64)
    alias, info = _collapse_rows(hashes, samples, live, n_hap, lambda o: hv_idx[hv_off_i[o]:hv_off_i[o + 1]], ds)
    # ---- position-map segments + scan bounds per row ------------------------------------------
    seg_start, seg_rel_all, seg_gen_all = build_segments(ind, hv_idx, hv_o, hv_off, r0, chain, startp, hap_len, alias)
    haps = RowMeta(seg_start, seg_rel_all.astype(np.uint32), seg_gen_all, hap_len, alias, startp)
    haps.compute_scans(startp, stopp, pamlen, own)
    ds.set_meta(haps)
    if own is not None and own.partner is not None:
        _lib.check(L.hawk_hapset_set_ref_partner_range(ds._h, int(own.partner[0]), int(own.partner[1])), "hawk_hapset_set_ref_partner_range")
    ds.alias = alias
    ds.host_meta = haps
    if keep_plan:
        plan.set_meta(haps)
        if own is not None and own.partner is not None:
            _lib.check(L.hawk_xplan_set_ref_partner_range(plan._x, int(own.partner[0]), int(own.partner[1])), "hawk_xplan_set_ref_partner_range")
        plan.alias, plan.host_meta = alias, haps
        ds.plan = plan
    else:
        plan.close()
    kept = np.flatnonzero(alias == np.arange(n_hap)).tolist()
    return ds, _KeptInfos(info, kept), float(ms.value), kept


def _expand_rows_gt(ref_set, seq, startp: int, stopp: int, pamlen: int, samples: List[str], tab, g, col_off: np.ndarray, device,
                    keep_plan: bool = False, own: Optional[ScanOwnership] = None):
    """_expand_rows for lists that are still on the device (`g`: a hawk_gt after hawk_gt_lists): the plan is created from
    them in place (hawk_xplan_create_gt) - checks, position-map segments and the scan bounds' reverse look-ups are kernels
    over the lists; the host sees a few words per ROW (lengths, content hashes, two look-ups) and decides which rows
    collapse.  Labels read the lists / segments lazily (CarriedLists, RowMeta's fetch).  Takes ownership of `g`."""
    import ctypes as C
    from . import _lib
    from .expand import HaplotypeBuildError
    from .hapset import ExpansionPlan, _p
    r0, span, chain, altlen, alt_off, alt_codes = tab
    L = _lib.lib()
    counts = np.diff(col_off.astype(np.int64))
    live = np.flatnonzero(counts)
    lists = CarriedLists(g, int(col_off[-1]))
    u32 = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    arrs = [u32(r0), u32(span), u32(alt_off), u32(altlen), np.ascontiguousarray(chain, dtype=np.int32), np.ascontiguousarray(alt_codes)]
    xh, n_hap_c = C.c_void_p(), C.c_uint32(0)
    # where the scan bounds start from: the region's own rule (search_guides.py:49-84) or, for a tile of a larger region, its
    # seams; the end-of-region clamp is the region's and cannot be reproduced by a tile (divergences)
    lo_g = startp + 100 if (own is None or own.own_lo is None) else own.own_lo
    hi_g = stopp - 100 if (own is None or own.own_hi is None) else own.own_hi
    rc = L.hawk_xplan_create_gt(ref_set._h, g, len(r0), _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]), _p(arrs[5]),
                                len(alt_codes), C.c_int64(startp), 1 if own is None else 0, C.c_int64(lo_g), C.c_int64(hi_g),
                                C.byref(n_hap_c), C.byref(xh))
    if rc == _lib.HAWK_E_OVERLAP:
        raise HaplotypeBuildError("a chromosome copy carries overlapping variants")
    if rc == _lib.HAWK_E_CLAMP:
        raise HaplotypeBuildError("variant beyond the original region length (haplotype.py:199-201 clamp)")
    _lib.check(rc, "hawk_xplan_create_gt")
    n_hap = n_hap_c.value
    hap_len = np.zeros(n_hap, dtype=np.uint32)
    rev0, rev1 = np.zeros(n_hap, dtype=np.int64), np.zeros(n_hap, dtype=np.int64)
    _lib.check(L.hawk_xplan_rows(xh, _p(hap_len), _p(rev0), _p(rev1)), "hawk_xplan_rows")
    plan = ExpansionPlan(xh, hap_len, device)
    plan.n_records = int(col_off[-1])
    ds, hashes, ms_val = plan.run(want_hash=True)
    row_end = np.concatenate(([0, 0], col_off.astype(np.int64)[live + 1]))  # row r's list: [row_end[r], row_end[r + 1])
    alias, info = _collapse_rows(hashes, samples, live, n_hap, lambda o: lists.rows(int(row_end[o]), int(row_end[o + 1])), ds)

    def fetch():
        nseg = C.c_uint64(0)
        _lib.check(L.hawk_xplan_segments(plan._x, None, None, None, C.c_uint64(0), C.byref(nseg)), "hawk_xplan_segments")
        so = np.zeros(n_hap + 1, dtype=np.uint32)
        sr, sg = np.zeros(nseg.value, dtype=np.uint32), np.zeros(nseg.value, dtype=np.int64)
        _lib.check(L.hawk_xplan_segments(plan._x, _p(so), _p(sr), _p(sg), C.c_uint64(nseg.value), None), "hawk_xplan_segments")
        return so.astype(np.int64), sr, sg
    haps = RowMeta(None, None, None, hap_len, alias, startp, fetch=fetch, rev={lo_g: rev0, hi_g: rev1})
    haps.compute_scans(startp, stopp, pamlen, own)
    _lib.check(L.hawk_xplan_finish_meta(plan._x, _p(haps.scan_lo.astype(np.int32)), _p(haps.scan_hi.astype(np.int32))), "hawk_xplan_finish_meta")
    if own is not None and own.partner is not None:
        _lib.check(L.hawk_xplan_set_ref_partner_range(plan._x, int(own.partner[0]), int(own.partner[1])), "hawk_xplan_set_ref_partner_range")
    is_ref = np.zeros(n_hap, dtype=np.uint8)
    is_ref[0] = 1
    plan.ref_index, plan.is_ref = 0, is_ref
    _lib.check(L.hawk_xplan_install_meta(plan._x, ds._h), "hawk_xplan_install_meta")
    ds.ref_index, ds.is_ref = 0, is_ref
    ds.alias = alias
    ds.host_meta = haps
    plan.alias, plan.host_meta = alias, haps
    plan._lists = lists  # the labels' lists die with the plan at the latest
    if keep_plan:
        ds.plan = plan
    else:
        ds._plan_keepalive = plan  # the lazy segment download reads the plan
    kept = np.flatnonzero(alias == np.arange(n_hap)).tolist()
    return ds, _KeptInfos(info, kept), float(ms_val), kept


def _ref_only_set(seq, startp: int, stopp: int, pamlen: int, device, own: Optional[ScanOwnership] = None):
    from . import _lib
    from .hapset import DeviceHapSet
    ref_u8 = np.frombuffer(seq.encode("ascii"), dtype=np.uint8) if isinstance(seq, str) else np.asarray(seq, dtype=np.uint8)
    ref_seg = PosSegments.identity(startp, len(ref_u8))
    meta = HostHaplotype(ref_u8, ref_seg, True, _scan_for(ref_seg, startp, stopp, pamlen, own))
    ds = DeviceHapSet([meta], device)
    if own is not None and own.partner is not None:
        _lib.check(_lib.lib().hawk_hapset_set_ref_partner_range(ds._h, int(own.partner[0]), int(own.partner[1])),
                   "hawk_hapset_set_ref_partner_range")
    ds.host_meta = [meta]
    return ds


def invert_on_device(ctx, G: np.ndarray, r0: np.ndarray, chain: np.ndarray):
    """A 0/1 genotype matrix G[variant, chromosome copy] -> (hawk_gt handle whose carried-variant lists are in HBM,
    col_off[n_cols + 1]).  The caller owns the handle (hawk_gt_destroy, or hand it to _expand_rows_gt)."""
    import ctypes as C
    from . import _lib
    from .hapset import _p
    G = np.ascontiguousarray(G, dtype=np.uint8)
    nv, n_cols = G.shape
    if n_cols % 2:
        raise ValueError("genotype matrix needs two columns per sample")
    L = _lib.lib()
    g = C.c_void_p()
    _lib.check(L.hawk_gt_from_codes(ctx, _p(G), C.c_uint64(nv), n_cols // 2, C.byref(g)), "hawk_gt_from_codes")
    try:
        col_off = np.zeros(n_cols + 1, dtype=np.uint64)
        _lib.check(L.hawk_gt_lists(g, _p(np.arange(nv, dtype=np.uint32)), _p(np.ones(nv, dtype=np.uint8)),
                                   _p(np.ascontiguousarray(r0, dtype=np.int32)), _p(np.ascontiguousarray(chain, dtype=np.int32)), nv,
                                   _p(col_off), None, None), "hawk_gt_lists")
    except Exception:
        L.hawk_gt_destroy(g)
        raise
    return g, col_off


def carried_lists_on_device(ctx, G: np.ndarray, r0: np.ndarray, chain: np.ndarray, want_indels: bool = False):
    """A 0/1 genotype matrix G[variant, chromosome copy] -> per-copy carried-variant lists, on the device: the matrix goes
    up as allele codes (hawk_gt_from_codes) and is inverted by the kernels of the VCF path (hawk_gt_lists).
    -> (col_off[n_cols + 1], col_delta[n_cols], hv_idx, hv_o[, entry indices of the carried indels])"""
    import ctypes as C
    from . import _lib
    from .hapset import _p
    G = np.ascontiguousarray(G, dtype=np.uint8)
    nv, n_cols = G.shape
    if n_cols % 2:
        raise ValueError("genotype matrix needs two columns per sample")
    L = _lib.lib()
    g = C.c_void_p()
    _lib.check(L.hawk_gt_from_codes(ctx, _p(G), C.c_uint64(nv), n_cols // 2, C.byref(g)), "hawk_gt_from_codes")
    try:
        col_off = np.zeros(n_cols + 1, dtype=np.uint64)
        col_delta = np.zeros(n_cols, dtype=np.int64)
        ms_l = C.c_float(0)
        _lib.check(L.hawk_gt_lists(g, _p(np.arange(nv, dtype=np.uint32)), _p(np.ones(nv, dtype=np.uint8)),
                                   _p(np.ascontiguousarray(r0, dtype=np.int32)), _p(np.ascontiguousarray(chain, dtype=np.int32)), nv,
                                   _p(col_off), _p(col_delta), C.byref(ms_l)), "hawk_gt_lists")
        ne = int(col_off[-1])
        hv_idx = np.zeros(max(ne, 1), dtype=np.uint32)
        hv_o = np.zeros(max(ne, 1), dtype=np.int32)
        _lib.check(L.hawk_gt_lists_download(g, _p(hv_idx), _p(hv_o)), "hawk_gt_lists_download")
        indel = None
        if want_indels:
            ni = C.c_uint64(0)
            _lib.check(L.hawk_gt_lists_indels(g, None, C.c_uint64(0), C.byref(ni)), "hawk_gt_lists_indels")
            indel = np.zeros(max(ni.value, 1), dtype=np.uint32)
            _lib.check(L.hawk_gt_lists_indels(g, _p(indel), C.c_uint64(ni.value), C.byref(ni)), "hawk_gt_lists_indels")
            indel = indel[:ni.value]
    finally:
        L.hawk_gt_destroy(g)
    if want_indels:
        return col_off, col_delta, hv_idx[:ne], hv_o[:ne], indel
    return col_off, col_delta, hv_idx[:ne], hv_o[:ne]


def expand_on_device(reg: SynthRegion, pamlen: int, device: Optional[int] = None, sample_range: Optional[Tuple[int, int]] = None,
                     keep_plan: bool = False):
    """build_phased_haplotypes() with the sequence work done by hawk_hapset_expand: the host only
    prepares index arrays (which variants each chromosome copy carries, prefix sums of their length
    changes), labels and position-map segments; no haplotype string is ever formed.
    Returns (DeviceHapSet, [HapInfo] of the kept rows, kernel ms, kept row indices).  Rows that collapse onto an earlier row
    (haplotypes.py:274-294; homozygous copies, 326-333) stay in HBM with an empty scan range."""
    seq, startp, stopp = reg.sequence, reg.startp, reg.stopp
    ref_set = _ref_only_set(seq, startp, stopp, pamlen, device)
    slo, shi = sample_range if sample_range is not None else (0, len(reg.samples))
    if not reg.variants or shi <= slo:
        ref_set.alias = np.zeros(1, dtype=np.int64)
        return ref_set, [HapInfo(["REF"], ())], 0.0, [0]
    if hasattr(reg, "variant_columns"):
        tab = _variant_table_columns(reg.variant_columns(), seq, startp)
    else:
        tab = _variant_table(np.array([v.pos for v in reg.variants]), [v.ref for v in reg.variants], [v.alt for v in reg.variants], seq, startp)
    r0, span, chain = tab[0], tab[1], tab[2]
    # which variants each chromosome copy carries: the in-memory genotype matrix goes to the device as allele codes and
    # is inverted there (hawk_gt_lists, the kernels of the VCF path) - a host-side nonzero scan of the 155 MB matrix
    # took 0.4 s of C3's expansion (`sample_range`: this rank's block of the panel - haplotypes shard across GPUs, REF
    # is on every rank)
    gm = getattr(reg, "gt_matrix", None)
    if gm is not None and gm.shape == (len(reg.variants), 2 * len(reg.samples)) and \
            all(np.array_equal(gm[i], reg.variants[i].gt.reshape(-1)) for i in (0, len(reg.variants) // 2, -1)):
        G = gm if (slo, shi) == (0, len(reg.samples)) else gm[:, 2 * slo:2 * shi]  # the panel already is one matrix
    else:
        G = np.stack([v.gt[slo:shi].reshape(-1) for v in reg.variants])  # [site, 2*sample]
    g, col_off = invert_on_device(ref_set._ctx, G, r0, chain)
    if int(col_off[-1]) == 0:
        from . import _lib
        _lib.lib().hawk_gt_destroy(g)
        ref_set.alias = np.zeros(1, dtype=np.int64)
        return ref_set, [HapInfo(["REF"], ())], 0.0, [0]
    return _expand_rows_gt(ref_set, seq, startp, stopp, pamlen, reg.samples[slo:shi], tab, g, col_off, device, keep_plan=keep_plan)


class VcfVariants:
    """The variant table of a VCF block after VariantRecord.split() (variant.py:313-331): one entry per ALT allele,
    alleles and position adjusted like adjust_multiallelic, ids like _compute_id, AF from INFO."""

    def __init__(self, block, contig: Optional[str] = None):
        from .variant import adjust_multiallelic
        pos, ref, alt, vid, af, line, allele = [], [], [], [], [], [], []
        nan = float("nan")
        for i, f in enumerate(block.fixed):
            chrom, p, r, alt_s, info = f[0], int(f[1]), f[3], f[4], f[7]
            k = info.find("AF=")
            if "," not in alt_s:  # one ALT allele: the usual record, no per-allele loop
                a1 = nan
                if k != -1:  # variant.py:188-206
                    e = info.find(";", k + 3)
                    a_s = info[k + 3: len(info) if e == -1 else e]
                    if "," in a_s:
                        raise ValueError(f"AF number does not match the alleles number ({a_s.count(',') + 1} - 1)")
                    a1 = float(a_s)
                if len(r) == 1 and len(alt_s) == 1:
                    r_, a_, p_ = r, alt_s, p
                else:
                    r_, a_, p_ = adjust_multiallelic(r, alt_s, p)
                pos.append(p_); ref.append(r_); alt.append(a_); af.append(a1)
                vid.append(f"{chrom}-{p}-{r}/{alt_s}")
                line.append(i); allele.append(1)
                continue
            alts = alt_s.split(",")
            afs = [nan] * len(alts)
            if k != -1:
                e = info.find(";", k + 3)
                afs = [float(x) for x in info[k + 3: len(info) if e == -1 else e].split(",")]
                if len(afs) != len(alts):
                    raise ValueError(f"AF number does not match the alleles number ({len(afs)} - {len(alts)})")
            for a, al in enumerate(alts):
                r_, a_, p_ = adjust_multiallelic(r, al, p)
                pos.append(p_); ref.append(r_); alt.append(a_); af.append(afs[a])
                vid.append(f"{chrom}-{p}-{r}/{al}")
                line.append(i); allele.append(a + 1)
        order = np.argsort(np.asarray(pos, dtype=np.int64), kind="stable")
        pick = lambda xs: [xs[int(j)] for j in order]
        self.pos = np.asarray(pos, dtype=np.int64)[order]
        self.ref, self.alt, self.id, self.af = pick(ref), pick(alt), pick(vid), pick(af)
        self.line = np.asarray(line, dtype=np.uint32)[order]
        self.allele = np.asarray(allele, dtype=np.uint8)[order]

    def __len__(self) -> int:
        return len(self.pos)


def expand_from_vcf(region_seq: str, startp: int, stopp: int, block, samples: List[str], pamlen: int, phased: bool = True,
                    device: Optional[int] = None, keep_plan: bool = False):
    """f3 -> f1: the records of a VCF block (readers.VCF.fetch_block) straight to device haplotypes.  The sample
    columns are parsed on the device (hawk_gt_parse) and inverted into carried-variant lists there
    (hawk_gt_lists); the host handles the fixed columns of the records only.
    Returns (DeviceHapSet, [HapInfo], kernel ms {parse, lists, expand}, kept rows, VcfVariants); with `keep_plan` the
    expansion plan stays attached as `ds.plan` (its view searches without planes: hawk_xplan_view)."""
    import ctypes as C
    from . import _lib
    from .hapset import _p
    if not phased:
        raise ValueError("expand_from_vcf builds phased haplotypes; unphased VCFs go through haplotypes.py")
    ref_set = _ref_only_set(region_seq, startp, stopp, pamlen, device)
    vt = VcfVariants(block)
    if len(vt) == 0:
        ref_set.alias = np.zeros(1, dtype=np.int64)
        return ref_set, [HapInfo(["REF"], ())], {"parse": 0.0, "lists": 0.0, "expand": 0.0}, [0], vt
    tab = _variant_table(vt.pos, vt.ref, vt.alt, region_seq, startp)
    r0, chain = tab[0], tab[2]
    L = _lib.lib()
    g = C.c_void_p()
    ms_parse, ms_lists = C.c_float(0), C.c_float(0)
    text = np.ascontiguousarray(block.text)
    _lib.check(L.hawk_gt_parse(ref_set._ctx, _p(text), C.c_uint64(len(text)), _p(block.line_off), _p(block.gt_off),
                               C.c_uint64(len(block)), len(samples), C.byref(g), C.byref(ms_parse)), "hawk_gt_parse")
    try:
        flags = np.zeros(len(block), dtype=np.uint8)
        _lib.check(L.hawk_gt_codes(g, None, _p(flags)), "hawk_gt_codes")
        if np.any(flags & 2):
            raise ValueError(f"VCF record {int(np.flatnonzero(flags & 2)[0])} does not have one genotype per sample")
        if np.any(flags & 1):  # variant.py:489-506: a phased VCF must hold a|b everywhere
            raise ValueError("Phased genotypes cannot have more than one allele on each copy")
        if np.any(flags & 4):
            raise ValueError(f"Malformed genotype in VCF record {int(np.flatnonzero(flags & 4)[0])}")
        n_cols = 2 * len(samples)
        col_off = np.zeros(n_cols + 1, dtype=np.uint64)
        col_delta = np.zeros(n_cols, dtype=np.int64)
        _lib.check(L.hawk_gt_lists(g, _p(vt.line), _p(vt.allele), _p(np.ascontiguousarray(r0, dtype=np.int32)),
                                   _p(np.ascontiguousarray(chain, dtype=np.int32)), len(vt), _p(col_off), _p(col_delta),
                                   C.byref(ms_lists)), "hawk_gt_lists")
        handed = False
        if int(col_off[-1]) == 0:  # no chromosome copy carries anything: REF alone
            ref_set.alias = np.zeros(1, dtype=np.int64)
            return ref_set, [HapInfo(["REF"], ())], {"parse": float(ms_parse.value), "lists": float(ms_lists.value), "expand": 0.0}, [0], vt
        handed = True
        ds, info, ms_expand, kept = _expand_rows_gt(ref_set, region_seq, startp, stopp, pamlen, samples, tab, g, col_off, device,
                                                    keep_plan=keep_plan)
    finally:
        if not locals().get("handed", False):
            L.hawk_gt_destroy(g)
    return ds, info, {"parse": float(ms_parse.value), "lists": float(ms_lists.value), "expand": ms_expand}, kept, vt


class RowLabel:
    """What the report needs to know about one device haplotype row (the Haplotype fields of guide.py:64-118)."""

    __slots__ = ("samples", "variants", "afs", "id", "segments")

    def __init__(self, samples: str, variants: str, afs, hid: str, segments):
        self.samples, self.variants, self.afs, self.id, self.segments = samples, variants, afs, hid, segments


def row_labels(reg: SynthRegion, ds, info: List[HapInfo], kept: List[int]) -> List[Optional[RowLabel]]:
    """Labels per device row of an expand_on_device() set (None for rows collapsed onto another row): samples
    joined as collapse_haplotypes does, variant ids `chr-pos-ref/alt`, allele frequencies by id."""
    vid = [f"{reg.contig}-{v.pos}-{v.ref}/{v.alt}" for v in reg.variants]
    af = {vid[i]: float(v.af) for i, v in enumerate(reg.variants)}
    out: List[Optional[RowLabel]] = [None] * ds.n_hap
    is_indel = [len(v.ref) != len(v.alt) for v in reg.variants]
    for r, inf in zip(kept, info):
        idx = [int(i) for i in inf.variant_idx]
        ids = [vid[i] for i in idx if not is_indel[i]] + [vid[i] for i in idx if is_indel[i]]  # haplotype.py:234-242
        out[r] = RowLabel(",".join(inf.samples), ",".join(ids) if ids else "NA", {k: af[k] for k in ids}, f"hap_{r:08d}",
                          ds.host_meta[r].seg)
    return out


class _LazySegments:
    """list-like: the PosSegments of a kept row on demand (host_meta[r].seg), None for rows that collapsed onto another"""

    def __init__(self, host_meta, kept, n: int):
        self._hm, self._kept, self._n = host_meta, set(int(r) for r in kept), n

    def __len__(self):
        return self._n

    def __getitem__(self, r):
        r = int(r)
        return self._hm[r].seg if r in self._kept else None


def hap_labels(contig: str, variants, ds, info: List[HapInfo], kept: List[int]):
    """reports.HapLabels of an expanded set, straight from the carried-variant index lists (no per-row id strings):
    `variants` = the variant records in table order with .pos .ref .alt .af (synth.VariantSite) or a VcfVariants."""
    from .reports import HapLabels
    if hasattr(variants, "id"):
        vid, af = list(variants.id), np.asarray(variants.af, dtype=np.float64)
    else:
        vid = [f"{contig}-{v.pos}-{v.ref}/{v.alt}" for v in variants]
        af = np.array([float(v.af) for v in variants], dtype=np.float64)
    n = ds.n_hap
    samples, ids = [""] * n, [""] * n
    hm = ds.host_meta
    segs = _LazySegments(hm, kept, n)  # a row's PosSegments only when the report's Python fallback asks for it
    is_ref = np.zeros(n, dtype=bool)
    cnt = np.zeros(n, dtype=np.int64)
    parts = []
    for k, (r, inf) in enumerate(zip(kept, info)):
        samples[r] = ",".join(inf.samples)
        ids[r] = f"hap_{k:08d}"
        is_ref[r] = r == 0 or samples[r] == "REF"
        idx = np.asarray(inf.variant_idx, dtype=np.int64)
        cnt[r] = len(idx)
    order = np.argsort(np.asarray(kept))
    for j in order.tolist():
        parts.append(np.asarray(info[j].variant_idx, dtype=np.int64))
    var_idx = np.concatenate(parts) if parts else np.zeros(0, np.int64)
    lab = HapLabels(samples, ids, is_ref, np.concatenate(([0], np.cumsum(cnt))), var_idx, vid, af, segs)
    if hasattr(hm, "seg_start"):  # the rows' position maps as flat CSR arrays: what the library's polish helper walks
        lab.seg_csr = (np.ascontiguousarray(hm.seg_start, dtype=np.uint64), np.ascontiguousarray(hm.seg_rel, dtype=np.uint32),
                       np.ascontiguousarray(hm.seg_gen, dtype=np.int64))
    return lab
