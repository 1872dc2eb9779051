"""Haplotype reconstruction from in-memory variant records, phased (reference haplotypes.py:106-368) and
unphased (370-712).  Callers hand in the ``VariantRecord`` list that ``VCF.fetch`` returns for the region
(readers.VCF.fetch)."""
import random
import string
from collections import defaultdict
from typing import Dict, List, Set, Tuple

from .coordinate import Coordinate
from .haplotype import Haplotype
from .region import Region
from .sequence import Sequence
from .utils import flatten_list
from .variant import VTYPES, VariantRecord


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


# ---------------------------------------------------------------------------- unphased VCFs (haplotypes.py:370-712)
def compute_haplotypes_unphased(variants: List[VariantRecord], samples: List[str]) -> Dict[str, List[VariantRecord]]:
    """haplotypes.py:161-185: SNVs only, first-copy sample sets."""
    variants = [v for v in variants if v.vtype[0] == VTYPES[0]]
    sv: Dict[str, List[VariantRecord]] = {s: [] for s in samples}
    for v in variants:
        assert len(v.samples) == 1
        for s in v.samples[0][0]:
            sv[s].append(v)
    return {s: v for s, v in sv.items() if v}


def compute_indel_haplotypes_unphased(variants: List[VariantRecord], samples: List[str]) -> Dict[str, List[VariantRecord]]:
    """haplotypes.py:188-212: the indel and its overlapping SNVs, for the indel's carriers only."""
    sv: Dict[str, List[VariantRecord]] = {s: [] for s in samples}
    for v in variants:
        assert len(v.samples) == 1
        for s in v.samples[0][0]:
            if s in sv:
                sv[s].append(v)
    return {s: v for s, v in sv.items() if v}


def _solve_haplotypes_unphased(sequence: str, coordinates: Coordinate, phased: bool, variants: List[VariantRecord], sample: str,
                               debug: bool) -> List[Haplotype]:
    """haplotypes.py:384-405.  generate_variants_combinations keys its groups by variant id, so there is exactly
    one combination: all of the sample's variants."""
    h = Haplotype(Sequence(sequence, debug), coordinates, phased, 0, debug)
    h.add_variants_unphased([v for v in variants if v is not None], sample)
    return [h]


def solve_haplotypes_unphased(sample_variants: Dict[str, List[VariantRecord]], hapseqs: List[Haplotype], refseq: str,
                              coordinates: Coordinate, phased: bool, debug: bool) -> List[Haplotype]:
    for sample, variants in sample_variants.items():
        hapseqs += _solve_haplotypes_unphased(refseq, coordinates, phased, variants, sample, debug)
    return collapse_haplotypes(hapseqs, debug)


def classify_variants(variants: List[VariantRecord]) -> Tuple[List[VariantRecord], List[VariantRecord]]:
    snvs = [v for v in variants if v.vtype[0] == VTYPES[0]]
    indels = [v for v in variants if v.vtype[0] != VTYPES[0]]
    return snvs, indels


def compute_snvs_haplotype_unphased(snvs: List[VariantRecord], samples: List[str], refseq: str, coordinates: Coordinate,
                                    phased: bool, debug: bool) -> List[Haplotype]:
    return solve_haplotypes_unphased(compute_haplotypes_unphased(snvs, samples), [], refseq, coordinates, phased, debug)


def create_indel_window(indel: VariantRecord, region: Region):
    """haplotypes.py:465-484: the region cut to 100 nt either side of the indel."""
    indel_length = len(indel.alt[0]) - len(indel.ref) if indel.alt else 0
    window_start = max(region.start, indel.position - 100)
    window_stop = min(region.stop, indel.position - indel_length + 100)
    coords = Coordinate(region.contig, window_start, window_stop, 0)
    startrel, stoprel = window_start - region.start, window_stop - region.start + 1
    return region.sequence.sequence[startrel:stoprel], coords, window_start, window_stop, indel_length


def find_overlapping_snvs(indel_start: int, indel_stop: int, snvs: List[VariantRecord]) -> List[VariantRecord]:
    return [s for s in snvs if indel_start <= s.position <= indel_stop]


def retrieve_indel_samples(indel: VariantRecord) -> Set[str]:
    return set(indel.samples[0][0]) if indel and len(indel.samples) > 0 else set()


def set_haplotypes_samples(indel_haplotypes: List[Haplotype], indel_samples: Set[str]) -> List[Haplotype]:
    for hap in indel_haplotypes:
        if hap.samples != "REF":
            final = (set(hap.samples.split(",")) if hap.samples else set()).intersection(indel_samples)
            if final:
                hap.samples = ",".join(sorted(final))
    return indel_haplotypes


def create_indels_haplotype_unphased(indel: VariantRecord, snvs: List[VariantRecord], region: Region, phased: bool,
                                     debug: bool) -> List[Haplotype]:
    """haplotypes.py:512-547: one window haplotype set per indel, built for its carriers."""
    seq, coords, w_start, w_stop, _ = create_indel_window(indel, region)
    overlapping = find_overlapping_snvs(w_start, w_stop, snvs)
    carriers = retrieve_indel_samples(indel)
    haps: List[Haplotype] = []
    if carriers:
        sv = compute_indel_haplotypes_unphased([indel] + overlapping, sorted(carriers))  # the reference iterates a set: hash order
        haps = solve_haplotypes_unphased(sv, [], seq, coords, phased, debug)
        haps = set_haplotypes_samples(haps, carriers)
    return haps


def add_variants_unphased(haplotypes: List[Haplotype], region: Region, samples: List[str], variants: List[VariantRecord],
                          phased: bool, debug: bool) -> List[Haplotype]:
    """haplotypes.py:672-712 with the VCF object replaced by its sample list: SNV-only IUPAC haplotypes per sample
    over the whole region, plus a window haplotype set per indel."""
    variants = flatten_list([v.split() for v in variants])
    snvs, indels = classify_variants(variants)
    if snvs:
        haplotypes.extend(compute_snvs_haplotype_unphased(snvs, samples, region.sequence.sequence, region.coordinates, phased, debug))
    for indel in indels:
        if region.coordinates.startp <= indel.position < region.coordinates.stopp:
            haplotypes.extend(create_indels_haplotype_unphased(indel, snvs, region, phased, debug))
    return haplotypes


def generate_haplotype_ids(haplotypes: Dict[Region, List[Haplotype]]) -> Dict[Region, List[Haplotype]]:
    chars = string.ascii_letters + string.digits
    for _, haps in haplotypes.items():
        ids = set()
        while len(ids) < len(haps):
            ids.add("hap_" + "".join(random.choices(chars, k=8)))
        for h, i in zip(haps, list(ids)):
            h.id = i
    return haplotypes
