"""Haplotype — same constructor, properties and ``add_variants_phased`` as the reference's
haplotype.Haplotype (haplotype.py:23-491), on a byte array + position-map segments instead of a
character list and two per-base dicts (≈190 MB per Mb-haplotype in the reference, SURVEY.md
fact 5).  ``posmap`` / ``posmap_rev`` are read-only mapping views over the segments."""
import os
from collections.abc import Mapping
from typing import Dict, List, Tuple, Union

import numpy as np

from .coordinate import Coordinate
from .crisprhawk_error import CrisprHawkHaplotypeError, CrisprHawkIupacTableError
from .exception_handlers import exception_handler
from .expand import HaplotypeBuildError, expand_haplotype
from .hapset import PosSegments
from .region import Region
from .sequence import Sequence
from .utils import IUPAC_ENCODER, IUPACTABLE, match_iupac
from .variant import VTYPES, VariantRecord


class _PosMap(Mapping):
    """relative position -> genomic position (haplotype.py:90-104)."""

    def __init__(self, seg: PosSegments):
        self._seg = seg

    def __getitem__(self, rel: int) -> int:
        if not 0 <= rel < self._seg.length:
            raise KeyError(rel)
        return int(self._seg.lookup(rel))

    def __iter__(self):
        return iter(range(self._seg.length))

    def __len__(self) -> int:
        return self._seg.length


class _PosMapRev(Mapping):
    """genomic position -> LAST relative position carrying it (dict-overwrite rule, haplotype.py:159)."""

    def __init__(self, seg: PosSegments):
        self._seg = seg

    def __getitem__(self, g: int) -> int:
        r = self._seg.rev(g)
        if r < 0:
            raise KeyError(g)
        return r

    def __contains__(self, g) -> bool:
        return self._seg.rev(g) >= 0

    def __iter__(self):
        return iter(sorted(set(self._seg.full().tolist())))

    def __len__(self) -> int:
        return len(set(self._seg.full().tolist()))

    def keys(self):
        return list(iter(self))


class Haplotype(Region):
    def __init__(self, sequence: Sequence, coord: Coordinate, phased: bool, chromcopy: int, debug: bool) -> None:
        self._debug = debug
        super().__init__(sequence, coord)
        self._size = len(sequence)
        self._variant_alleles: Dict[int, List[Tuple[str, str, int]]] = {}
        self._variants = "NA"
        self._afs: Dict[str, float] = {}
        self._samples = "REF"
        self._phased = phased
        self._chromcopy = chromcopy
        self._arr = np.frombuffer(sequence.sequence.encode("ascii"), dtype=np.uint8)
        self._seg = PosSegments.identity(self.coordinates.start, self._size)

    def __str__(self) -> str:
        return f"{self._samples}: {self.sequence.sequence}"

    def substring(self, start: int, stop: int) -> str:
        return self.sequence.sequence[start:stop]

    def add_variants_phased(self, variants: List[VariantRecord], sample: str) -> None:
        """haplotype.py:214-252"""
        if not self._phased:
            exception_handler(ValueError, "Unphased haplotype, unable to add phased variants", os.EX_DATAERR, True)
        variants = _sort_variants(variants)
        try:
            arr, seg = expand_haplotype(self._arr, self.coordinates.start,
                                        [(v.position, v.ref.encode(), v.alt[0].encode()) for v in variants])
        except HaplotypeBuildError as e:
            raise ValueError(str(e)) from e
        self._arr, self._seg = arr, seg
        suffix = "1|0" if self._chromcopy == 0 else "0|1"
        self.sequence = Sequence(arr.tobytes().decode("ascii"), self._debug, allow_lower_case=True)
        self._samples = f"{sample}:{suffix}" if self._phased else sample
        self._variants = ",".join(v.id[0] for v in variants)
        self._afs = {v.id[0]: v.afs[0] for v in variants}

    def add_variants_unphased(self, variants: List[VariantRecord], sample: str) -> None:
        """haplotype.py:254-316: heterozygous-agnostic encoding of an unphased sample - every SNV becomes the
        lower-case IUPAC letter of {reference base(s), alt}, indels are applied as in the phased case, and
        `variant_alleles` remembers (ref, alt, position) per final relative position for resolve_guide."""
        variants = _sort_variants(variants)
        start = self.coordinates.start
        cur: Dict[int, str] = {}        # SNV letter written so far, by genomic position
        sites = []
        va: Dict[int, List[Tuple[str, str, int]]] = {}
        off = 0                          # length change of the indels already applied (they come last, by position)
        deleted: List[Tuple[int, int]] = []
        # _initialize_posmap (haplotype.py:90-104) sizes the position map for the FINAL length: a variant whose relative
        # position is not among its keys yet - a SNV in the last N bases of a stretch that loses N bases in total - raises
        # KeyError in the look-up, which the unphased insert swallows (274-277): the variant is listed but never applied
        map_keys = len(self.sequence.sequence) + sum(len(v.alt[0]) - len(v.ref) for v in variants)
        for v in variants:
            pos, ref, alt = v.position, v.ref, v.alt[0]
            if any(a < pos <= b for a, b in deleted):
                continue                 # position removed by a previous deletion (haplotype.py:262-265)
            chain = len(alt) - len(ref)
            posrel = pos - start + off
            if posrel >= map_keys:
                continue                 # not in the position map: skipped, like the deleted positions
            stop = posrel + abs(chain) + 1 if chain < 0 else posrel + 1
            ref_seq = self.sequence.sequence
            refnt = "".join(cur.get(pos + i, ref_seq[pos - start + i: pos - start + i + 1]) for i in range(stop - posrel))
            if not match_iupac(ref, refnt):
                raise ValueError(f"Mismatching reference alleles in VCF and reference sequence at position {pos} ({refnt} - {ref})")
            if posrel in va:
                if pos == va[posrel][0][2]:
                    va[posrel].append((ref, alt, pos))
            else:
                va[posrel] = [(ref, alt, pos)]
            if v.vtype[0] == VTYPES[0]:
                try:
                    letter = IUPAC_ENCODER["".join({IUPACTABLE[refnt.upper()], alt})]
                except KeyError as e:
                    exception_handler(CrisprHawkIupacTableError, f"An error occurred while encoding {refnt}>{alt} at position {pos} as IUPAC character",
                                      os.EX_DATAERR, self._debug, e)
                cur[pos] = letter.lower()
                sites = [x for x in sites if x[0] != pos] + [(pos, ref.encode(), letter.encode())]
            else:
                va = {p_: a_ for p_, a_ in va.items() if p_ <= posrel or p_ >= stop}
                va = {(p_ + chain if p_ > posrel else p_): a_ for p_, a_ in va.items()}
                sites.append((pos, ref.encode(), alt.encode()))
                if chain < 0:
                    deleted.append((pos, pos - chain))
                off += chain
        try:
            arr, seg = expand_haplotype(self._arr, start, sites)
        except HaplotypeBuildError as e:
            raise ValueError(str(e)) from e
        self._arr, self._seg = arr, seg
        self.sequence = Sequence(arr.tobytes().decode("ascii"), self._debug, allow_lower_case=True)
        self._variant_alleles = va
        self._samples = sample
        self._variants = ",".join(v.id[0] for v in variants)
        self._afs = {v.id[0]: v.afs[0] for v in variants}

    def homozygous_samples(self) -> None:
        if self._samples == "REF":
            exception_handler(CrisprHawkHaplotypeError, "REF haplotype cannot be homozygous", os.EX_DATAERR, self._debug)
        self._samples = ",".join(f"{s.split(':')[0]}:1|1" for s in self._samples.split(","))

    def set_afs(self, afs: Dict[str, float]) -> None:
        self._afs = afs

    def set_posmap(self, posmap, posmap_rev=None) -> None:
        if isinstance(posmap, _PosMap):
            self._seg = posmap._seg
        elif isinstance(posmap, PosSegments):
            self._seg = posmap
        else:  # a plain dict as the reference uses
            from .hapset import segments_from_posmap
            pm = np.array([posmap[i] for i in range(len(posmap))], dtype=np.int64)
            rel, gen = segments_from_posmap(pm)
            self._seg = PosSegments(rel, gen, len(pm))

    def set_variant_alleles(self, variant_alleles) -> None:
        self._variant_alleles = variant_alleles

    # the byte view the device packer consumes
    @property
    def array(self) -> np.ndarray:
        if len(self._arr) != len(self.sequence):
            self._arr = np.frombuffer(self.sequence.sequence.encode("ascii"), dtype=np.uint8)
        return self._arr

    @property
    def segments(self) -> PosSegments:
        return self._seg

    @property
    def samples(self) -> str:
        return self._samples

    @samples.setter
    def samples(self, value: str) -> None:
        if not isinstance(value, str):
            exception_handler(CrisprHawkHaplotypeError, f"Samples must be a string, got {type(value).__name__} instead",
                              os.EX_DATAERR, True)
        self._samples = value

    @property
    def variants(self) -> str:
        return self._variants

    @variants.setter
    def variants(self, value: str) -> None:
        if not isinstance(value, str):
            exception_handler(CrisprHawkHaplotypeError, f"Variants must be a string, got {type(value).__name__} instead",
                              os.EX_DATAERR, True)
        self._variants = value

    @property
    def afs(self) -> Dict[str, float]:
        return self._afs

    @property
    def phased(self) -> bool:
        return self._phased

    @property
    def posmap(self) -> Mapping:
        return _PosMap(self._seg)

    @property
    def posmap_rev(self) -> Mapping:
        return _PosMapRev(self._seg)

    @property
    def id(self) -> str:
        return self._id

    @id.setter
    def id(self, value: str) -> None:
        if not isinstance(value, str):
            exception_handler(CrisprHawkHaplotypeError, f"Haplotype id must be a string, got {type(value).__name__} instead",
                              os.EX_DATAERR, True)
        self._id = value

    @property
    def variant_alleles(self):
        return self._variant_alleles


def _sort_variants(variants: List[VariantRecord]) -> List[VariantRecord]:
    """haplotype.py:494-512: SNPs (sorted) before indels (sorted)."""
    snps = [v for v in variants if v.vtype[0] == VTYPES[0]]
    indels = [v for v in variants if v.vtype[0] != VTYPES[0]]
    return sorted(snps) + sorted(indels)


def _compute_chains(variants: List[VariantRecord]) -> List[int]:
    return [len(v.alt[0]) - len(v.ref) for v in variants]
