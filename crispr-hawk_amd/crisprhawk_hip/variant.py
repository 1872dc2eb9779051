"""VariantRecord — one VCF row / allele (reference variant.py:25-620).  The tabix-backed VCF
reader of the reference (622-830) is pysam I/O and out of scope; records are built from the
tab-split fields of a VCF line exactly as VariantRecord.read_vcf_line (286-311) does."""
import os
from typing import List, Optional, Set, Tuple

import numpy as np

from .exception_handlers import exception_handler

VTYPES = ["snp", "indel"]


def _assign_vtype(ref: str, alt: str) -> str:
    return VTYPES[1] if len(ref) != len(alt) else VTYPES[0]


def _compute_id(chrom: str, pos: int, ref: str, alt: str) -> str:
    return f"{chrom}-{pos}-{ref}/{alt}"


def adjust_multiallelic(ref: str, alt: str, pos: int) -> Tuple[str, str, int]:
    """variant.py:456-486"""
    if len(ref) == len(alt):
        return ref[0], alt[0], pos
    if len(ref) > len(alt):
        return ref[len(alt) - 1:], alt[-1], pos + len(alt) - 1
    return ref[-1], alt[len(ref) - 1:], pos + len(ref) - 1


def _genotypes_to_samples(genotypes: List[str], samples: List[str], allelesnum: int, phased: bool,
                          debug: bool) -> List[Tuple[Set[str], Set[str]]]:
    """variant.py:489-619"""
    hap = [(set(), set()) for _ in range(allelesnum)]
    sep = "|" if phased else "/"
    for i, gt in enumerate(genotypes):
        alleles = gt.split(":")[0].split(sep)
        if phased:
            if len(alleles) != 2:
                exception_handler(ValueError, "Phased genotypes cannot have more than one allele on each copy",
                                  os.EX_DATAERR, debug)
            g1, g2 = alleles
            if g1 not in ("0", "."):
                hap[int(g1) - 1][0].add(samples[i])
            if g2 not in ("0", "."):
                hap[int(g2) - 1][1].add(samples[i])
        elif len(alleles) != 2:
            for g in alleles:
                if g not in ("0", "."):
                    hap[int(g) - 1][0].add(samples[i])
        else:
            g1, g2 = alleles
            if g1 not in ("0", ".") and g1 == g2:
                hap[int(g1) - 1][0].add(samples[i])
                hap[int(g2) - 1][1].add(samples[i])
            else:
                if g1 not in ("0", "."):
                    hap[int(g1) - 1][0].add(samples[i])
                if g2 not in ("0", "."):
                    hap[int(g2) - 1][0].add(samples[i])
    return hap


class VariantRecord:
    def __init__(self, debug: bool) -> None:
        self._debug = debug

    def __repr__(self) -> str:
        return f'<{self.__class__.__name__} object; variant="{self._chrom} {self._position} {self._ref} {",".join(self._alt)}">'

    def __str__(self) -> str:
        return f"{self._chrom}\t{self._position}\t{self._ref}\t{','.join(self._alt)}"

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, VariantRecord):
            return NotImplemented
        return (self._chrom == other.contig and self._position == other.position and self._ref == other.ref
                and self._alt == other.alt)

    def __lt__(self, other: "VariantRecord") -> bool:
        return self._position < other.position

    def __gt__(self, other: "VariantRecord") -> bool:
        return self._position > other.position

    def __hash__(self) -> int:
        return hash((self._chrom, self._position, self._ref, tuple(self._alt)))

    def _retrieve_af(self, info: str) -> List[float]:
        i = info.find("AF=")
        if i == -1:
            return [np.nan] * self._allelesnum
        i += 3
        j = info.find(";", i)
        j = len(info) if j == -1 else j
        afs = list(map(float, info[i:j].split(",")))
        if len(afs) != self._allelesnum:
            exception_handler(ValueError, f"AF number does not match the alleles number ({len(afs)} - {self._allelesnum})",
                              os.EX_DATAERR, self._debug)
        return afs

    def read_vcf_line(self, variant: List[str], samples: List[str], phased: bool) -> None:
        self._chrom = variant[0]
        self._position = int(variant[1])
        self._ref = variant[3]
        self._alt = variant[4].split(",")
        self._allelesnum = len(self._alt)
        self._vtype = [_assign_vtype(self._ref, a) for a in self._alt]
        self._filter = variant[6]
        self._afs = self._retrieve_af(variant[7])
        self._vid = [_compute_id(self._chrom, self._position, self._ref, a) for a in self._alt]
        self._samples = _genotypes_to_samples(variant[9:], samples, self._allelesnum, phased, self._debug)

    def _copy(self, i: int) -> "VariantRecord":
        v = VariantRecord(self._debug)
        ref, alt, position = adjust_multiallelic(self._ref, self._alt[i], self._position)
        v._chrom, v._position, v._ref, v._alt, v._allelesnum = self._chrom, position, ref, [alt], 1
        v._vtype, v._filter, v._afs, v._vid, v._samples = ([self._vtype[i]], self._filter, [self._afs[i]],
                                                           [self._vid[i]], [self._samples[i]])
        return v

    def split(self, vtype: Optional[str] = None) -> List["VariantRecord"]:
        keep = VTYPES if vtype is None else [vtype]
        return [self._copy(i) for i in range(len(self._vtype)) if self._vtype[i] in keep]

    @property
    def filter(self) -> str:
        return self._filter

    @property
    def contig(self) -> str:
        return self._chrom

    @property
    def position(self) -> int:
        return self._position

    @property
    def ref(self) -> str:
        return self._ref

    @property
    def alt(self) -> List[str]:
        return self._alt

    @property
    def vtype(self) -> List[str]:
        return self._vtype

    @property
    def afs(self) -> List[float]:
        return self._afs

    @property
    def samples(self) -> List[Tuple[Set[str], Set[str]]]:
        return self._samples

    @property
    def id(self) -> List[str]:
        return self._vid

    @property
    def allelesnum(self) -> int:
        return self._allelesnum
