"""Haplotype reconstruction from in-memory variant records (what `reconstruct_haplotypes` does between reading the VCF
and handing haplotypes to the search, reference haplotypes.py:106-368 phased, 370-712 unphased).

Only the public behaviour is the reference's: which haplotypes exist for a region, their cased sequences, labels,
position maps and (unphased) per-position allele tables.  The construction is organised around one idea instead of the
reference's chain of helpers: every haplotype is "a region string + one list of variant records applied by
`Haplotype.add_variants_*`"; what differs between the modes is only how records are dealt to carriers (`_deal`) and
which stretch of the region a haplotype covers (`_Stretch`).  Identical cased sequences are then merged (`_merge_equal`,
collapse_haplotypes in the reference, 274-294).
"""
import random
import string
from typing import Callable, Dict, Iterable, List, Optional, Sequence as Seq, Tuple

from .coordinate import Coordinate
from .haplotype import Haplotype
from .region import Region
from .sequence import Sequence
from .variant import VTYPES, VariantRecord

INDEL_WINDOW = 100  # bases kept either side of an unphased indel (haplotypes.py:465-484)


class _Stretch:
    """The part of a region a haplotype is built over: its bases and its coordinates."""

    __slots__ = ("bases", "coords")

    def __init__(self, bases: str, coords: Coordinate):
        self.bases, self.coords = bases, coords

    @classmethod
    def whole(cls, region: Region) -> "_Stretch":
        return cls(region.sequence.sequence, region.coordinates)

    @classmethod
    def around(cls, region: Region, record: VariantRecord) -> Tuple["_Stretch", int, int]:
        """The region cut to INDEL_WINDOW bases either side of `record` (clamped to the region)."""
        grow = len(record.alt[0]) - len(record.ref) if record.alt else 0
        lo = max(region.start, record.position - INDEL_WINDOW)
        hi = min(region.stop, record.position - grow + INDEL_WINDOW)
        a = lo - region.start
        return cls(region.sequence.sequence[a:hi - region.start + 1], Coordinate(region.contig, lo, hi, 0)), lo, hi

    def fresh(self, phased: bool, copy: int, debug: bool) -> Haplotype:
        return Haplotype(Sequence(self.bases, debug), self.coords, phased, copy, debug)


def _split_alleles(records: Iterable[VariantRecord]) -> List[VariantRecord]:
    out: List[VariantRecord] = []
    for r in records:
        out.extend(r.split())
    return out


def _deal(records: Seq[VariantRecord], carriers: Seq[str], copies: Tuple[int, ...]) -> Dict[str, Tuple[List[VariantRecord], ...]]:
    """Records dealt to their carriers, per chromosome copy in `copies`, in record order; carriers without any record
    are dropped, carriers not listed are ignored.  (A record here is biallelic: `samples[0]` holds its carrier sets.)"""
    hands = {name: tuple([] for _ in copies) for name in carriers}
    for rec in records:
        sets = rec.samples[0]
        for slot, c in enumerate(copies):
            for name in sets[c]:
                hand = hands.get(name)
                if hand is not None:
                    hand[slot].append(rec)
    return {name: hand for name, hand in hands.items() if any(hand)}


def _merge_equal(haps: Seq[Haplotype], debug: bool) -> List[Haplotype]:
    """One haplotype per distinct cased sequence (first-seen order).  The merged haplotype takes the first member's
    coordinates, phase, position map, allele frequencies and allele table; labels are unions - sorted here, where the
    reference joins Python sets in hash order; an all-upper-case sequence is REF whatever its members were called."""
    order: List[str] = []
    members: Dict[str, List[Haplotype]] = {}
    for h in haps:
        key = h.sequence.sequence
        if key not in members:
            members[key] = []
            order.append(key)
        members[key].append(h)
    merged = []
    for key in order:
        group = members[key]
        head = group[0]
        m = Haplotype(Sequence(key, debug, allow_lower_case=True), head.coordinates, head.phased, 0, debug)
        is_ref = key.isupper()
        m.samples = "REF" if is_ref else ",".join(sorted({g.samples for g in group}))
        m.variants = "NA" if is_ref else ",".join(sorted({g.variants for g in group}))
        m.set_afs(head.afs)
        m.set_posmap(head.segments)
        m.set_variant_alleles(head.variant_alleles)
        merged.append(m)
    return merged


def collapse_haplotypes(haplotypes: List[Haplotype], debug: bool) -> List[Haplotype]:
    return _merge_equal(haplotypes, debug)


def initialize_haplotypes(regions, debug: bool) -> Dict[Region, List[Haplotype]]:
    """The REF haplotype of every region (haplotypes.py:106-129)."""
    return {r: [_Stretch.whole(r).fresh(False, 0, debug)] for r in regions}


# ---------------------------------------------------------------------------------------------- phased
def add_variants_phased(haplotypes: List[Haplotype], region: Region, samples: List[str], variants: List[VariantRecord], phased: bool,
                        debug: bool) -> List[Haplotype]:
    """Two chromosome copies per carrier sample (one when both come out identical, labelled 1|1), appended to
    `haplotypes`, then merged by sequence (haplotypes.py:297-368, 714-745; the VCF object replaced by its sample list)."""
    stretch = _Stretch.whole(region)
    for sample, (first, second) in _deal(_split_alleles(variants), samples, (0, 1)).items():
        pair = []
        for copy, records in ((0, first), (1, second)):
            h = stretch.fresh(phased, copy, debug)
            h.add_variants_phased(records, sample)
            pair.append(h)
        if pair[0].sequence.sequence == pair[1].sequence.sequence:
            pair[0].homozygous_samples()
            pair.pop()
        haplotypes += pair
    return _merge_equal(haplotypes, debug)


# ---------------------------------------------------------------------------------------------- unphased
def _unphased_set(stretch: _Stretch, hands: Dict[str, Tuple[List[VariantRecord], ...]], phased: bool, debug: bool) -> List[Haplotype]:
    """One IUPAC-encoded haplotype per carrier (all of its records at once: the reference's combination generator
    keys by variant id and therefore yields exactly this one combination, haplotypes.py:384-405), merged by sequence."""
    built = []
    for sample, (records,) in hands.items():
        h = stretch.fresh(phased, 0, debug)
        h.add_variants_unphased(records, sample)
        built.append(h)
    return _merge_equal(built, debug)


def add_variants_unphased(haplotypes: List[Haplotype], region: Region, samples: List[str], variants: List[VariantRecord],
                          phased: bool, debug: bool) -> List[Haplotype]:
    """Unphased VCFs (haplotypes.py:370-712): SNVs become IUPAC letters on one whole-region haplotype per carrier
    (carriers read from the first genotype slot); every indel inside the region gets its own set of short haplotypes
    - the region cut to 100 nt around it - built for the indel's carriers from the indel and the SNVs inside that
    window, and labelled with the carriers only."""
    alleles = _split_alleles(variants)
    snvs = [v for v in alleles if v.vtype[0] == VTYPES[0]]
    if snvs:
        haplotypes.extend(_unphased_set(_Stretch.whole(region), _deal(snvs, samples, (0,)), phased, debug))
    c = region.coordinates
    for indel in (v for v in alleles if v.vtype[0] != VTYPES[0]):
        if not c.startp <= indel.position < c.stopp:
            continue
        carriers = set(indel.samples[0][0]) if len(indel.samples) else set()
        if not carriers:
            continue
        stretch, lo, hi = _Stretch.around(region, indel)
        nearby = [indel] + [s for s in snvs if lo <= s.position <= hi]
        # the reference walks the carrier set in hash order; sorted here so that runs are reproducible
        for h in _unphased_set(stretch, _deal(nearby, sorted(carriers), (0,)), phased, debug):
            if h.samples != "REF":
                kept = set(h.samples.split(",")) & carriers if h.samples else set()
                if kept:
                    h.samples = ",".join(sorted(kept))
            haplotypes.append(h)
    return haplotypes


def generate_haplotype_ids(haplotypes: Dict[Region, List[Haplotype]]) -> Dict[Region, List[Haplotype]]:
    """`hap_` + 8 random alphanumerics, unique within a region (haplotypes.py:807-814; unseeded in the reference too)."""
    alphabet = string.ascii_letters + string.digits
    for haps in haplotypes.values():
        seen: Dict[str, None] = {}
        while len(seen) < len(haps):
            seen.setdefault("hap_" + "".join(random.choices(alphabet, k=8)))
        for h, hid in zip(haps, seen):
            h.id = hid
    return haplotypes
