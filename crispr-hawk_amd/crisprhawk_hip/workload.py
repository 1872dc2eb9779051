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
