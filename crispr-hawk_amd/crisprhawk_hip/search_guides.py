"""search_guides — same entry points as the reference module (search_guides.py), with the
scan / filter / coordinate / redundancy work done by the HIP kernels through the C ABI.

``search()`` packs the haplotypes into bit-planes (K1), runs the fused device search
(hawk_search) and materialises ``Guide`` objects in the reference's list order.  The
``haplotypes_bits`` argument is accepted for signature compatibility; the planes are the
encoded form and are built once per call for all haplotypes together.
With an unphased VCF the candidates come from the same device pass (the PAM match is a set
intersection, so IUPAC-encoded haplotypes need no special casing) and are then expanded on the
host through ``resolve_guide`` exactly as search_guides.py:163-257, 473-480 does: that step is a
Cartesian product over a handful of ambiguous positions per window, string work by nature."""
import os
from collections import defaultdict
from typing import DefaultDict, Dict, List, Optional, Tuple, Union

import numpy as np

from itertools import product

from .crisprhawk_error import CrisprHawkCfdScoreError, CrisprHawkIupacTableError
from .exception_handlers import exception_handler
from .guide import GUIDESEQPAD, Guide
from .hapset import DeviceHapSet, GuideTable, HostHaplotype, PosSegments
from .haplotype import Haplotype
from .pam import PAM
from .region import Region
from .utils import IUPACTABLE, VERBOSITYLVL, print_verbosity

PADDING = 100  # region_constructor.py:21


def compute_scan_start_stop(hap: Haplotype, region_start: int, region_stop: int, pamlen: int) -> Tuple[int, int]:
    """search_guides.py:49-84"""
    rev = hap.posmap_rev
    stop_p = min(region_stop - PADDING, hap.stop)
    if stop_p == region_stop - PADDING and stop_p not in rev:
        upper = hap.segments.max_gen()
        for p in range(stop_p, upper + 1):
            if p in rev:
                stop_p = p
                break
    scan_stop = rev[stop_p] - pamlen + 1
    start_p = max(region_start + PADDING, hap.start)
    return rev[start_p], scan_stop


def _device_set(region: Region, haplotypes: List[Haplotype], pamlen: int) -> DeviceHapSet:
    haps = []
    for h in haplotypes:
        scan = compute_scan_start_stop(h, region.start, region.stop, pamlen)
        haps.append(HostHaplotype(h.array, h.segments, h.samples == "REF", scan))
    return DeviceHapSet(haps)


def scan_haplotype(pam: PAM, haplotype, start: int, stop: int, debug: bool) -> Tuple[List[int], List[int]]:
    """search_guides.py:87-99.  ``haplotype`` is an EncodedSequence (or a Haplotype)."""
    from .encoder import EncodedSequence
    if isinstance(haplotype, EncodedSequence):
        ds = haplotype._ds
        ds.set_meta([HostHaplotype(b"", PosSegments.identity(0, len(haplotype)), True, (start, stop))])
    else:
        ds = DeviceHapSet([HostHaplotype(haplotype.array, haplotype.segments, True, (start, stop))])
    (f, r), = ds.pam_scan(pam.bits, pam.bitsrc, len(pam))
    return f.tolist(), r.tolist()


def pam_search(pam: PAM, region: Region, haplotypes: List[Haplotype], haplotypes_bits, verbosity: int,
               debug: bool) -> List[Tuple[List[int], List[int]]]:
    """search_guides.py:102-131"""
    ds = _device_set(region, haplotypes, len(pam))
    hits = ds.pam_scan(pam.bits, pam.bitsrc, len(pam))
    for h, (f, r) in zip(haplotypes, hits):
        print_verbosity(f"Found {len(f) + len(r)} PAM occurrences ({len(f)} on 5'-3'; {len(r)} on 3'-5')", verbosity,
                        VERBOSITYLVL[3])
    return [(f.tolist(), r.tolist()) for f, r in hits]


def is_pamhit_valid(pamhit_pos: int, haplen: int, guidelen: int, pamlen: int, right: bool) -> bool:
    if right:
        return pamhit_pos + guidelen + pamlen + GUIDESEQPAD < haplen
    return pamhit_pos - guidelen - GUIDESEQPAD >= 0


def is_pamhit_in_range(poshit: int, guidelen: int, pamlen: int, haplen: int, right: bool) -> bool:
    lbound = poshit - GUIDESEQPAD if right else poshit - guidelen - GUIDESEQPAD
    rbound = poshit + guidelen + pamlen + GUIDESEQPAD if right else poshit + pamlen + GUIDESEQPAD
    return lbound >= 0 and rbound <= haplen


def extract_guide_sequence(haplotype: Haplotype, position: int, pamlen: int, guidelen: int, right: bool) -> str:
    if right:
        return "".join(haplotype[position - GUIDESEQPAD: position + guidelen + pamlen + GUIDESEQPAD])
    return "".join(haplotype[position - guidelen - GUIDESEQPAD: position + pamlen + GUIDESEQPAD])


def _valid_guide(pamguide: str, pam: PAM, direction: int, right: bool, debug: bool) -> bool:
    """search_guides.py:163-169: does the resolved PAM slice still match the PAM pattern?"""
    p = PAM(pamguide, right, debug)
    p.encode(0)
    pat = pam.bits if direction == 0 else pam.bitsrc
    n = len(pam)
    for i, nib in enumerate(p.bits_list):  # match() on nibbles (search_guides.py:32-46)
        pn = (pat >> (4 * (n - 1 - i))) & 0xF
        if pn and not (pn & nib):
            return False
    return True


def _decode_iupac(nt: str, pos: int, h: Haplotype, debug: bool) -> str:
    """search_guides.py:175-213"""
    try:
        ntiupac = IUPACTABLE[nt.upper()]
    except KeyError as e:
        exception_handler(CrisprHawkIupacTableError, f"Invalid IUPAC character ({nt})", os.EX_DATAERR, debug, e)
    if len(ntiupac) == 1:
        return ntiupac.lower() if nt.islower() else ntiupac
    return "".join(n if n == alleles[0] else n.lower() for n in list(ntiupac) for alleles in h.variant_alleles[pos])


def resolve_guide(guideseq: str, pam: PAM, direction: int, right: bool, pos: int, guidelen: int, h: Haplotype,
                  debug: bool) -> List[str]:
    """search_guides.py:216-257: all allele combinations of an IUPAC-encoded window whose PAM still matches."""
    p = pos - GUIDESEQPAD if right else pos - guidelen - GUIDESEQPAD
    alts = ["".join(g) for g in product(*[list(_decode_iupac(nt, p + i, h, debug)) for i, nt in enumerate(guideseq)])]
    idx = GUIDESEQPAD if right else (len(guideseq) - GUIDESEQPAD - len(pam))
    return [g for g in alts if _valid_guide(g[idx: idx + len(pam)], pam, direction, right, debug)]


def adjust_guide_position(posmap, posrel: int, guidelen: int, pamlen: int, right: bool) -> Tuple[int, int]:
    start = posmap[posrel] if right else posmap[posrel - guidelen]
    stop = posmap[posrel + guidelen + pamlen] if right else posmap[posrel + pamlen]
    return start, stop


def retrieve_guide_posmap(posmap, posrel: int, guidelen: int, pamlen: int, right: bool) -> Dict[int, int]:
    pivot = posrel if right else posrel - guidelen
    return {i: posmap[pivot + i] for i in range(guidelen + pamlen)}


def group_guides_position(guides: List[Guide], debug: bool):
    """search_guides.py:306-337"""
    pos_guide: DefaultDict[str, Dict[int, Union[Optional[Guide], List[Guide]]]] = defaultdict(lambda: {0: None, 1: []})
    for guide in guides:
        key = f"{guide.start}_{guide.strand}"
        if guide.samples == "REF":
            if pos_guide[key][0] is not None:
                exception_handler(CrisprHawkCfdScoreError,
                                  f"Duplicate REF guide at position {guide.start}? CFDon calculation failed",
                                  os.EX_DATAERR, debug)
            pos_guide[key][0] = guide
        pos_guide[key][1].append(guide)
    return pos_guide


def remove_redundant_guides(guides: List[Guide], debug: bool) -> List[Guide]:
    """search_guides.py:340-369 on Guide lists (search() applies the same rule on the device)."""
    out = []
    for _, grp in group_guides_position(guides, debug).items():
        ref, alts = grp[0], grp[1]
        if ref is None:
            out.extend(alts)
            continue
        rs = ref.sequence[GUIDESEQPAD:-GUIDESEQPAD].upper()
        for g in alts:
            gs = g.sequence[GUIDESEQPAD:-GUIDESEQPAD].upper()
            if (g.samples != "REF" and gs != rs) or (g.samples == "REF" and gs == rs):
                out.append(g)
    return out


def guides_from_table(tab: GuideTable, haplotypes: List[Haplotype], debug: bool, order: Optional[np.ndarray] = None) -> List[Guide]:
    """Guide(...) with the constructor arguments of search_guides.py:488-502 for each table row."""
    tab.download()
    order = tab.reference_order() if order is None else order
    wins = tab.windows()
    ras = tab.right_as_stored()
    L = tab.guidelen + tab.pamlen
    guides = []
    for i in order:
        h = haplotypes[int(tab.hap[i])]
        r = bool(ras[i])
        pivot = int(tab.pos[i]) if r else int(tab.pos[i]) - tab.guidelen

        def _pm(seg=h.segments, pivot=pivot):
            g = seg.lookup(np.arange(pivot, pivot + L))
            return {k: int(g[k]) for k in range(L)}

        g = Guide(int(tab.start[i]), int(tab.stop[i]), wins[i], tab.guidelen, tab.pamlen, int(tab.strand[i]), h.samples,
                  h.variants, h.afs, _pm, debug, r, h.id)
        g._hip_cfdon = float(tab.cfdon[i])
        g._hip_pos, g._hip_hap = int(tab.pos[i]), int(tab.hap[i])
        guides.append(g)
    return guides


def search(pam: PAM, region: Region, haplotypes: List[Haplotype], haplotypes_bits, guidelen: int, right: bool,
           variants_present: bool, phased: bool, verbosity: int, debug: bool, cfd_tables=None) -> List[Guide]:
    """search_guides.py:510-548.  ``cfd_tables=(mm, pam)`` additionally scores CFDon in the same
    device pass (kept on each Guide for scoring.cfdon_score)."""
    print_verbosity(f"Searching guide candidates in {region.coordinates}", verbosity, VERBOSITYLVL[3])
    ds = _device_set(region, haplotypes, len(pam))
    mm, pt = cfd_tables if cfd_tables is not None else (None, None)
    if variants_present and not phased:
        tab = ds.search(pam.bits, pam.bitsrc, len(pam), guidelen, right)
        return _resolve_unphased(tab, haplotypes, pam, guidelen, right, debug)
    tab = ds.search(pam.bits, pam.bitsrc, len(pam), guidelen, right, mm, pt)
    return guides_from_table(tab, haplotypes, debug)


def _resolve_unphased(tab: GuideTable, haplotypes: List[Haplotype], pam: PAM, guidelen: int, right: bool, debug: bool) -> List[Guide]:
    """retrieve_guides' unphased branch (search_guides.py:473-480) over the device rows, in the
    reference's emission order, then remove_redundant_guides on the resolved sequences."""
    tab.download()
    wins = tab.windows()
    ras = tab.right_as_stored()
    L = guidelen + len(pam)
    guides: List[Guide] = []
    for i in tab.emission_order():
        h = haplotypes[int(tab.hap[i])]
        r, pos, strand = bool(ras[i]), int(tab.pos[i]), int(tab.strand[i])
        if not is_pamhit_valid(pos, len(h), guidelen, len(pam), r):
            continue
        pivot = pos if r else pos - guidelen
        for seq in resolve_guide(wins[i], pam, strand, r, pos, guidelen, h, debug):
            def _pm(seg=h.segments, pivot=pivot):
                g = seg.lookup(np.arange(pivot, pivot + L))
                return {k: int(g[k]) for k in range(L)}
            g = Guide(int(tab.start[i]), int(tab.stop[i]), seq, guidelen, len(pam), strand, h.samples, h.variants,
                      h.afs, _pm, debug, r, h.id)
            g._hip_pos, g._hip_hap = pos, int(tab.hap[i])  # where the row came from (reports.report_from_guides)
            guides.append(g)
    return remove_redundant_guides(guides, debug)
