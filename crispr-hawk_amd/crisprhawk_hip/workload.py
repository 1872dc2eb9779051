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

from .expand import expand_haplotype, scan_bounds
from .hapset import HostHaplotype, PosSegments
from .synth import SynthRegion


class HapInfo:
    """Host-side labels of one device haplotype (what Guide rows inherit from it)."""

    __slots__ = ("samples", "variant_idx")

    def __init__(self, samples: List[str], variant_idx):
        self.samples = samples
        self.variant_idx = variant_idx


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


def expand_on_device(reg: SynthRegion, pamlen: int, device: Optional[int] = None):
    """build_phased_haplotypes() with the sequence work done by hawk_hapset_expand: the host only
    prepares index arrays (which variants each chromosome copy carries, prefix sums of their length
    changes), labels and position-map segments; no haplotype string is ever formed.
    Returns (DeviceHapSet, [HapInfo] of the kept rows, kernel ms, kept row indices).  Rows that collapse onto an earlier row
    (haplotypes.py:274-294; homozygous copies, 326-333) stay in HBM with an empty scan range."""
    import ctypes as C
    from . import _lib
    from .expand import HaplotypeBuildError
    from .hapset import DeviceHapSet, _p

    seq = reg.sequence
    startp, stopp, n_ref = reg.startp, reg.stopp, len(reg.sequence)
    ref_u8 = np.frombuffer(seq.encode("ascii"), dtype=np.uint8)
    ref_seg = PosSegments.identity(startp, n_ref)
    ref_set = DeviceHapSet([HostHaplotype(ref_u8, ref_seg, True, scan_bounds(ref_seg, startp, stopp, pamlen))], device)
    nv = len(reg.variants)
    if nv == 0:
        ref_set.alias = np.zeros(1, dtype=np.int64)
        return ref_set, [HapInfo(["REF"], ())], 0.0, [0]
    r0 = np.array([v.pos - startp for v in reg.variants], dtype=np.int64)
    reflen = np.array([len(v.ref) for v in reg.variants], dtype=np.int64)
    altlen = np.array([len(v.alt) for v in reg.variants], dtype=np.int64)
    order = np.argsort(r0, kind="stable")
    if np.any(order != np.arange(nv)):
        raise HaplotypeBuildError("variants must be sorted by position")
    # replaced span as the reference computes it (haplotype.py:197-201): |chain|+1 for deletions, else 1
    chain = altlen - reflen
    span = np.where(chain < 0, -chain + 1, 1)
    if np.any((chain < 0) & (altlen != 1)) or np.any((chain == 0) & (reflen != 1)):
        raise HaplotypeBuildError("device expansion handles SNVs, deletions (alt of one base) and insertions")
    if np.any(r0[1:] < r0[:-1] + span[:-1]) or np.any(r0 < 0) or np.any(r0 + span > n_ref):
        raise HaplotypeBuildError("overlapping variants / variant outside the region")
    for v, a, sp in zip(reg.variants, r0, span):  # REF allele must match (haplotype.py:203-208)
        if seq[a:a + sp] != v.ref[:sp] if len(v.ref) >= sp else True:
            raise HaplotypeBuildError(f"Mismatching reference alleles at position {v.pos}")
    alt_blob = "".join(v.alt for v in reg.variants).encode("ascii")
    alt_codes = _NIB[np.frombuffer(alt_blob, dtype=np.uint8)]
    alt_off = np.zeros(nv, dtype=np.int64)
    alt_off[1:] = np.cumsum(altlen)[:-1]
    # rows: REF, then every chromosome copy that carries at least one variant, in (sample, copy) order
    G = np.stack([v.gt.reshape(-1) for v in reg.variants])  # [site, 2*sample]
    cols, sites = np.nonzero(G.T)                             # sorted by column, then site
    counts = np.bincount(cols, minlength=G.shape[1])
    live = np.flatnonzero(counts)
    n_hap = 1 + len(live)
    hv_off = np.zeros(n_hap + 1, dtype=np.uint64)
    hv_off[2:] = np.cumsum(counts[live])
    hv_idx = sites.astype(np.uint32)
    c = chain[hv_idx]
    excl = np.cumsum(c) - c
    row_of = np.repeat(np.arange(len(live)), counts[live])
    starts = hv_off[1:-1].astype(np.int64)
    excl -= excl[starts][row_of] if len(starts) else 0
    hv_o = (r0[hv_idx] + excl).astype(np.int32)
    tot = np.zeros(n_hap, dtype=np.int64)
    np.add.at(tot, row_of + 1, c)
    hap_len = (n_ref + tot).astype(np.uint32)
    if np.any(hv_o.astype(np.int64) + span[hv_idx] > n_ref):  # the reference's clamp (haplotype.py:199-201) would fire
        raise HaplotypeBuildError("variant beyond the original region length (haplotype.py:199-201 clamp)")
    L = _lib.lib()
    handle = C.c_void_p()
    hashes = np.zeros((n_hap, 2), dtype=np.uint64)
    ms = C.c_float(0)
    u32 = lambda a: np.ascontiguousarray(a, dtype=np.uint32)
    arrs = [u32(r0), u32(span), u32(alt_off), u32(altlen), np.ascontiguousarray(alt_codes), hv_off, hv_idx, hv_o, hap_len]
    _lib.check(L.hawk_hapset_expand(ref_set._h, nv, _p(arrs[0]), _p(arrs[1]), _p(arrs[2]), _p(arrs[3]), _p(arrs[4]), len(alt_codes),
                                    n_hap, _p(hv_off), _p(hv_idx), _p(hv_o), _p(hap_len), C.byref(handle), _p(hashes), C.byref(ms)),
               "hawk_hapset_expand")
    ds = DeviceHapSet.from_handle(handle, hap_len, device)
    # ---- labels, homozygous merge, collapse by content (all on 16-byte hashes) ----------------
    key = [bytes(hashes[i]) for i in range(n_hap)]
    col_of_row = np.concatenate(([-1], live))
    first: Dict[bytes, int] = {key[0]: 0}
    info: List[Optional[HapInfo]] = [HapInfo(["REF"], ())] + [None] * (n_hap - 1)
    alias = np.arange(n_hap)
    row_of_col = {int(cc): i + 1 for i, cc in enumerate(live)}
    for si in range(len(reg.samples)):
        rows = [row_of_col.get(2 * si), row_of_col.get(2 * si + 1)]
        if rows[0] is None and rows[1] is None:
            continue
        keys = [key[r] if r is not None else key[0] for r in rows]  # a copy without variants is the REF sequence
        name = reg.samples[si]
        if keys[0] == keys[1]:
            entries = [(rows[0], f"{name}:1|1")]
            if rows[1] is not None and rows[0] is not None:
                alias[rows[1]] = rows[0]
        else:
            entries = [(rows[0], f"{name}:1|0"), (rows[1], f"{name}:0|1")]
        for r, label in entries:
            if r is None:
                continue  # collapses onto REF, which keeps samples == "REF" (haplotypes.py:255-258)
            j = first.get(key[r])
            if j is None:
                first[key[r]] = r
                info[r] = HapInfo([label], hv_idx[int(hv_off[r]):int(hv_off[r + 1])])
            else:
                alias[r] = j
                if j != 0:
                    info[j].samples.append(label)
    # ---- position-map segments + scan bounds per row ------------------------------------------
    # all rows at once: every carried deletion opens one segment behind it, every carried insertion of n bases
    # opens n + 1 (the inserted bases all map to the anchor position, haplotype.py:106-159)
    ch_all = chain[hv_idx]
    ind = np.flatnonzero(ch_all != 0)
    row_all = row_of + 1  # device row of every list entry (row 0 is REF)
    o_i = hv_o[ind].astype(np.int64)
    pos_i = r0[hv_idx[ind]] + startp
    ch_i = ch_all[ind]
    row_i = row_all[ind]
    nseg_i = np.where(ch_i < 0, 1, ch_i + 1)
    rep = np.repeat(np.arange(len(ind)), nseg_i)
    first_of = np.cumsum(nseg_i) - nseg_i
    k_in = np.arange(len(rep)) - first_of[rep]           # 0..n within an insertion's run, 0 for a deletion
    is_del = ch_i[rep] < 0
    seg_rel_all = o_i[rep] + 1 + k_in
    seg_gen_all = np.where(is_del, pos_i[rep] + 1 - ch_i[rep], np.where(k_in < ch_i[rep], pos_i[rep], pos_i[rep] + 1))
    seg_row_all = row_i[rep]
    keep = seg_rel_all < hap_len[seg_row_all].astype(np.int64)
    seg_rel_all, seg_gen_all, seg_row_all = seg_rel_all[keep], seg_gen_all[keep], seg_row_all[keep]
    srt = np.lexsort((seg_rel_all, seg_row_all))        # rows are already ascending; order each row by rel
    seg_rel_all, seg_gen_all, seg_row_all = seg_rel_all[srt], seg_gen_all[srt], seg_row_all[srt]
    seg_cnt = np.bincount(seg_row_all, minlength=n_hap)
    seg_start = np.concatenate(([0], np.cumsum(seg_cnt)))
    haps = []
    zero_rel, first_gen = np.zeros(1, np.uint32), np.array([startp], np.int64)
    for r in range(n_hap):
        if alias[r] != r:
            haps.append(HostHaplotype(b"", PosSegments.identity(startp, int(hap_len[r])), False, (0, 0)))
            continue
        a, b = int(seg_start[r]), int(seg_start[r + 1])
        if a == b:
            seg = PosSegments.identity(startp, int(hap_len[r]))
        else:
            seg = PosSegments(np.concatenate((zero_rel, seg_rel_all[a:b].astype(np.uint32))),
                              np.concatenate((first_gen, seg_gen_all[a:b])), int(hap_len[r]))
        haps.append(HostHaplotype(b"", seg, r == 0, scan_bounds(seg, startp, stopp, pamlen)))
    ds.set_meta(haps)
    ds.alias = alias
    ds.host_meta = haps
    kept = [i for i in range(n_hap) if alias[i] == i]
    return ds, [info[i] for i in kept], float(ms.value), kept


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
    for r, inf in zip(kept, info):
        ids = [vid[i] for i in inf.variant_idx]
        out[r] = RowLabel(",".join(inf.samples), ",".join(ids) if ids else "NA", {k: af[k] for k in ids}, f"hap_{r:08d}",
                          ds.host_meta[r].seg)
    return out
