"""Haplotype — same constructor, properties and ``add_variants_phased`` as the reference's
haplotype.Haplotype (haplotype.py:23-491), on a byte array + position-map segments instead of a
character list and two per-base dicts (≈190 MB per Mb-haplotype in the reference, SURVEY.md
fact 5).  ``posmap`` / ``posmap_rev`` are read-only mapping views over the segments."""
import os
from collections.abc import Mapping
from typing import Dict, List, Tuple, Union

import numpy as np

from .coordinate import Coordinate
from .crisprhawk_error import CrisprHawkHaplotypeError
from .exception_handlers import exception_handler
from .expand import HaplotypeBuildError, expand_haplotype
from .hapset import PosSegments
from .region import Region
from .sequence import Sequence
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
        self._seg = PosSegments.identity(self._coordinates.start, self._size)

    def __str__(self) -> str:
        return f"{self._samples}: {self._sequence.sequence}"

    def substring(self, start: int, stop: int) -> str:
        return self._sequence.sequence[start:stop]

    def add_variants_phased(self, variants: List[VariantRecord], sample: str) -> None:
        """haplotype.py:214-252"""
        if not self._phased:
            exception_handler(ValueError, "Unphased haplotype, unable to add phased variants", os.EX_DATAERR, True)
        variants = _sort_variants(variants)
        try:
            arr, seg = expand_haplotype(self._arr, self._coordinates.start,
                                        [(v.position, v.ref.encode(), v.alt[0].encode()) for v in variants])
        except HaplotypeBuildError as e:
            raise ValueError(str(e)) from e
        self._arr, self._seg = arr, seg
        suffix = "1|0" if self._chromcopy == 0 else "0|1"
        self._sequence = Sequence(arr.tobytes().decode("ascii"), self._debug, allow_lower_case=True)
        self._samples = f"{sample}:{suffix}" if self._phased else sample
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
        if len(self._arr) != len(self._sequence):
            self._arr = np.frombuffer(self._sequence.sequence.encode("ascii"), dtype=np.uint8)
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
