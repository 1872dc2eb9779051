"""Phased haplotype reconstruction from in-memory variant records (reference
haplotypes.py:106-368, 793-815).  VCF file access (pysam) is out of scope: callers hand in the
``VariantRecord`` list that ``VCF.fetch`` would have returned for the region."""
import random
import string
from collections import defaultdict
from typing import Dict, List, Tuple

from .coordinate import Coordinate
from .haplotype import Haplotype
from .region import Region
from .sequence import Sequence
from .utils import flatten_list
from .variant import VariantRecord


def initialize_haplotypes(regions, debug: bool) -> Dict[Region, List[Haplotype]]:
    return {r: [Haplotype(Sequence(r.sequence.sequence, debug), r.coordinates, False, 0, debug)] for r in regions}


def compute_haplotypes_phased(variants: List[VariantRecord], samples: List[str]):
    """haplotypes.py:132-159"""
    sv = {s: ([], []) for s in samples}
    for v in variants:
        assert len(v.samples) == 1
        for c in (0, 1):
            for s in v.samples[0][c]:
                sv[s][c].append(v)
    return {s: v for s, v in sv.items() if v[0] or v[1]}


def ishomozygous(haplotypes: List[Haplotype]) -> bool:
    return len({h.sequence.sequence for h in haplotypes}) == 1


def _collapse_haplotypes(sequence: str, haplotypes: List[Haplotype], debug: bool) -> Haplotype:
    """haplotypes.py:232-271 (sample / variant unions are sorted here; the reference joins
    Python sets, whose order is not reproducible run to run)."""
    hap = Haplotype(Sequence(sequence, debug, allow_lower_case=True), haplotypes[0].coordinates, haplotypes[0].phased, 0, debug)
    hap.samples = "REF" if sequence.isupper() else ",".join(sorted({h.samples for h in haplotypes}))
    hap.variants = "NA" if sequence.isupper() else ",".join(sorted({h.variants for h in haplotypes}))
    hap.set_afs(haplotypes[0].afs)
    hap.set_posmap(haplotypes[0].segments)
    hap.set_variant_alleles(haplotypes[0].variant_alleles)
    return hap


def collapse_haplotypes(haplotypes: List[Haplotype], debug: bool) -> List[Haplotype]:
    groups = defaultdict(list)
    for h in haplotypes:
        groups[h.sequence.sequence].append(h)
    return [_collapse_haplotypes(seq, hl, debug) for seq, hl in groups.items()]


def _solve_haplotypes_phased(sequence: str, coordinates: Coordinate, phased: bool, variants, sample: str, debug: bool):
    h0 = Haplotype(Sequence(sequence, debug), coordinates, phased, 0, debug)
    h0.add_variants_phased(variants[0], sample)
    h1 = Haplotype(Sequence(sequence, debug), coordinates, phased, 1, debug)
    h1.add_variants_phased(variants[1], sample)
    if ishomozygous([h0, h1]):
        h0.homozygous_samples()
        return [h0]
    return [h0, h1]


def solve_haplotypes_phased(sample_variants, hapseqs: List[Haplotype], refseq: str, coordinates: Coordinate,
                            phased: bool, debug: bool) -> List[Haplotype]:
    for sample, variants in sample_variants.items():
        hapseqs += _solve_haplotypes_phased(refseq, coordinates, phased, variants, sample, debug)
    return collapse_haplotypes(hapseqs, debug)


def add_variants_phased(haplotypes: List[Haplotype], region: Region, samples: List[str],
                        variants: List[VariantRecord], phased: bool, debug: bool) -> List[Haplotype]:
    """haplotypes.py:714-745 with the VCF object replaced by its sample list."""
    variants = flatten_list([v.split() for v in variants])
    sv = compute_haplotypes_phased(variants, samples)
    return solve_haplotypes_phased(sv, haplotypes, region.sequence.sequence, region.coordinates, phased, debug)


def generate_haplotype_ids(haplotypes: Dict[Region, List[Haplotype]]) -> Dict[Region, List[Haplotype]]:
    chars = string.ascii_letters + string.digits
    for _, haps in haplotypes.items():
        ids = set()
        while len(ids) < len(haps):
            ids.add("hap_" + "".join(random.choices(chars, k=8)))
        for h, i in zip(haps, list(ids)):
            h.id = i
    return haplotypes
