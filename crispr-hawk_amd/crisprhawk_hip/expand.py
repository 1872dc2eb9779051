"""Haplotype construction on byte arrays + position-map segments.

Restates ``Haplotype.add_variants_phased`` (reference haplotype.py:214-252 with
_sort_variants 494-512, _insert_variant_phased 185-212, _update_sequence 106-121 and
_update_posmap 138-159) without the reference's per-base dicts: the position map is kept as
unit-slope segments, so building one 1 Mb haplotype costs O(variants) small slices plus one
concatenate instead of an O(L) dict rebuild per variant (~14 ms per variant at
200 kb in the reference).

This is host logic feeding the kernels ("next"); the search itself never
runs here.
"""

from typing import List, Sequence, Tuple

import numpy as np

from .hapset import PosSegments


class HaplotypeBuildError(ValueError):
    pass


def _lower(b: bytes) -> bytes:
    return b.lower()


def expand_haplotype(ref: np.ndarray, startp: int, variants: Sequence[Tuple[int, bytes, bytes]]):
    """Apply one chromosome copy's variants [(pos, ref_allele, alt_allele)] to the region.

    Returns (uint8 haplotype array, PosSegments).  Raises HaplotypeBuildError where the
    reference raises (mismatching REF allele, variant on a deleted position) and also where the
    reference's ``posrel_stop > self._size`` clamp (haplotype.py:199-201) would fire.
    """
    n = len(ref)
    snv = [(p, r, a) for p, r, a in variants if len(r) == len(a)]
    indel = sorted([(p, r, a) for p, r, a in variants if len(r) != len(a)], key=lambda v: v[0])
    arr = ref
    if snv:
        arr = ref.copy()
        for p, r, a in snv:  # SNVs first (haplotype.py:506-512); posmap is still the identity
            i = p - startp
            if not 0 <= i < n:
                raise HaplotypeBuildError(f"variant position {p} outside region")
            cur = arr[i:i + len(r)].tobytes()
            if cur != r and cur.isupper():
                raise HaplotypeBuildError(f"Mismatching reference alleles at position {p} ({cur.decode()} - {r.decode()})")
            # _update_sequence replaces [posrel, posrel+1) with alt.lower(); len(alt)==len(ref)
            if len(a) == 1:
                arr[i] = a.lower()[0]
            else:  # MNV: the reference replaces ONE base by the whole alt (haplotype.py:197-211)
                indel.append((p, r, a))
        indel.sort(key=lambda v: v[0])
    if not indel:
        return arr, PosSegments.identity(startp, n)
    pieces: List[np.ndarray] = []
    rel: List[int] = [0]
    gen: List[int] = [startp]
    prev_end = 0  # in region coordinates
    off = 0       # sum of chains applied so far
    for p, r, a in indel:
        chain = len(a) - len(r)
        i = p - startp
        if i < prev_end or i >= n:
            raise HaplotypeBuildError(f"variant position {p} deleted by a previous variant or outside region")
        posrel = i + off
        span = (-chain + 1) if chain < 0 else 1
        if posrel + span > n:
            raise HaplotypeBuildError(f"variant at {p} beyond the original region length (haplotype.py:199-201 clamp)")
        cur = arr[i:i + span].tobytes()
        if cur != r and cur.isupper():
            raise HaplotypeBuildError(f"Mismatching reference alleles at position {p} ({cur.decode()} - {r.decode()})")
        pieces.append(arr[prev_end:i])
        pieces.append(np.frombuffer(a.lower(), dtype=np.uint8))
        prev_end = i + span
        if chain < 0:  # deletion: everything after the anchor jumps |chain| genomic positions
            rel.append(posrel + 1)
            gen.append(p + 1 - chain)
        elif chain > 0:  # insertion: inserted bases repeat the anchor's position
            for k in range(1, chain + 1):
                rel.append(posrel + k)
                gen.append(p)
            rel.append(posrel + chain + 1)
            gen.append(p + 1)
        off += chain
    pieces.append(arr[prev_end:])
    out = np.concatenate(pieces)
    seg_rel = np.array(rel, dtype=np.uint32)
    seg_gen = np.array(gen, dtype=np.int64)
    keep = seg_rel < len(out)  # a segment opening exactly at the end has no base
    return out, PosSegments(seg_rel[keep], seg_gen[keep], len(out))


def scan_start(seg: PosSegments, region_start: int, padding: int = 100) -> int:
    """Start half of compute_scan_start_stop (search_guides.py:68-70): posmap_rev[region.start + 100]."""
    start_p = region_start + padding
    r = seg.rev(start_p)
    if r < 0:
        raise KeyError(start_p)
    return r


def scan_stop(seg: PosSegments, region_stop: int, pamlen: int, padding: int = 100) -> int:
    """Stop half (search_guides.py:71-84): a deleted stop position walks forward to the next mapped one."""
    stop_p = region_stop - padding
    if seg.rev(stop_p) < 0:
        upper = seg.max_gen()
        for p in range(stop_p, upper + 1):
            if seg.rev(p) >= 0:
                stop_p = p
                break
    r = seg.rev(stop_p)
    if r < 0:
        raise KeyError(stop_p)
    return r - pamlen + 1


def scan_bounds(seg: PosSegments, region_start: int, region_stop: int, pamlen: int, padding: int = 100) -> Tuple[int, int]:
    """compute_scan_start_stop (search_guides.py:49-84); region_start/stop are the padded
    coordinates.  KeyError where the reference's dict lookup fails."""
    stop = scan_stop(seg, region_stop, pamlen, padding)
    return scan_start(seg, region_start, padding), stop
