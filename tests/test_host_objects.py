"""CPU tests of the host-side object mirrors, following the reference's own unit tests
(/root/reference/tests/test_offtarget.py, test_offtargets.py, test_guide.py, test_utils.py,
test_coordinate.py, test_haplotype.py, test_variant.py): same calls, same expectations."""
import numpy as np
import pytest

from crisprhawk_hip.coordinate import Coordinate
from crisprhawk_hip.guide import GUIDESEQPAD, Guide
from crisprhawk_hip.offtarget import Offtarget, _format_sequence, _retrieve_pam
from crisprhawk_hip.offtargets import _calculate_global_cfd, _calculate_offtargets_map, _filter_guides, crispritz_report_line
from crisprhawk_hip.genome import OffTargetHit, encode_guides, decode_window
from crisprhawk_hip.utils import calculate_chunks, dna2rna, match_iupac, reverse_complement, round_score
from crisprhawk_hip.variant import VariantRecord, adjust_multiallelic


def test_retrieve_pam_and_format_sequence():  # test_offtarget.py:25-50
    seq = "AGGTCGATCGG"
    assert _retrieve_pam(seq, 3, True) == seq[:3] and _retrieve_pam(seq, 3, False) == seq[-3:]
    spacer, formatted = _format_sequence(seq, "AGG", True)
    assert formatted.startswith("AGG") and spacer == seq[3:]
    spacer, formatted = _format_sequence(seq, "CGG", False)
    assert formatted.endswith("CGG") and spacer == seq[:-3]


def test_offtarget_parses_crispritz_row_and_reports():
    hit = OffTargetHit(0, "chr1", 12345, "+", 2, "ACGTACGTACGTACGTAGGTTGG")
    line = crispritz_report_line(hit, "ACGTACGTACGTACGTACGT", 3, False)
    ot = Offtarget(line, "NGG", False, True)
    assert (ot.chrom, ot.position, ot.strand, ot.mm) == ("chr1", 12345, "+", 2)
    assert ot.grna_ == "ACGTACGTACGTACGTACGT" and ot.grna == "ACGTACGTACGTACGTACGTNGG"
    assert ot.spacer == "ACGTACGTACGTACGTAgGTTGG"  # mismatches lower-case, observed PAM kept
    assert ot.cfd_inputs() == ("ACGTACGTACGTACGTACGT", "ACGTACGTACGTACGTAGGT", "GG")
    assert "position=12345" in repr(ot) and "strand=+" in repr(ot)
    cols = ot.report_line().split("\t")
    assert cols[:3] == ["chr1", "12345", "+"] and cols[5] == "NGG" and cols[9:] == ["NA", "NA"]
    ot.elevation = 0.75
    assert ot.elevation == "0.75"
    ot.elevation = float("nan")
    assert ot.elevation == "NA"
    with pytest.raises(TypeError):
        ot.elevation = "not_a_float"


def test_global_cfd_and_offtargets_map():  # test_offtargets.py: aggregates with dummy rows
    class G:
        def __init__(self, s): self.guide = s

    class O:
        def __init__(self, g, c): self.grna_, self.cfd = g, c

    guides = [G("ACGT"), G("acgt"), G("TTTT")]
    assert _filter_guides(guides) == {"ACGT", "TTTT"}
    ots = [O("ACGT", "0.5"), O("AC-GT", "NA"), O("TTTT", "1.0")]
    m = _calculate_offtargets_map(ots, guides)
    assert len(m["ACGT"]) == 2 and len(m["TTTT"]) == 1
    assert _calculate_global_cfd(m["ACGT"]) == 100 / 100.5
    assert _calculate_global_cfd([]) == 1.0


def test_guide_codec_roundtrip():
    g = encode_guides(["ACGTTGCA", "TTTTAAAA"])
    assert decode_window(int(g[0]), 0, 8) == "ACGTTGCA" and decode_window(int(g[1]), 0b101, 8) == "NTNTAAAA"
    with pytest.raises(ValueError):
        encode_guides(["ACGN"])


def _guide(seq, right=False, strand=0):
    return Guide(100, 123, seq, 20, 3, strand, "REF", "NA", {}, {}, True, right, "hap_x")


def test_guide_fields_and_reverse_complement():  # guide.py:184-255
    seq = "A" * 10 + "ACGTACGTACGTACGTACGT" + "TGG" + "C" * 10
    g = _guide(seq)
    assert (g.guide, g.pam, g.guidepam, len(g)) == ("ACGTACGTACGTACGTACGT", "TGG", "ACGTACGTACGTACGTACGTTGG", 43)
    assert g.guide_id == "100_123_0_hap_x_ACGTACGTACGTACGTACGT"
    g.reverse_complement()
    assert g.right and g.pam == "CCA" and g.sequence == reverse_complement(seq, True)
    assert g.azimuth_score == "NA"
    g.azimuth_score = 0.47249
    assert g.azimuth_score == "0.4725"
    g.cfdon_score = float("nan")
    assert g.cfdon_score == "NA"
    with pytest.raises(Exception):
        g.cfdon_score = "x"
    g.offtargets = 7
    assert g.offtargets == "7"


def test_utils_match_reference_tests():  # test_utils.py:10-57
    assert reverse_complement("ACGT", True) == "ACGT" and reverse_complement("AAGC", True) == "GCTT"
    assert reverse_complement("acgtn", True) == "nacgt"
    assert round_score(0.123456) == 0.1235
    assert dna2rna("ACGTt") == "ACGUu"
    assert match_iupac("A", "R") and not match_iupac("C", "R") and not match_iupac("AC", "R")
    assert calculate_chunks(list(range(10)), 3) == [(0, [0, 1, 2]), (3, [3, 4, 5]), (6, [6, 7, 8]), (9, [9])]


def test_coordinate_padding():  # test_coordinate.py
    c = Coordinate("chr1", 1000, 2000, 100)
    assert (c.start, c.stop, c.startp, c.stopp, str(c)) == (900, 2100, 1000, 2000, "chr1:1000-2000")
    assert Coordinate("chr1", 50, 60, 100).start == 0
    with pytest.raises(ValueError):
        Coordinate("chr1", 10, 5, 0)


def test_variant_record():  # test_variant.py:6-99
    assert adjust_multiallelic("A", "G", 10) == ("A", "G", 10)
    assert adjust_multiallelic("ACG", "A", 10) == ("ACG", "A", 10)
    assert adjust_multiallelic("A", "ACG", 10) == ("A", "ACG", 10)
    assert adjust_multiallelic("AC", "ACGT", 10) == ("C", "CGT", 11)
    v = VariantRecord(True)
    v.read_vcf_line(["chr1", "100", ".", "A", "G,AT", ".", "PASS", "AF=0.1,0.2", "GT", "1|0", "0|2", "2|1"], ["s1", "s2", "s3"], True)
    assert v.vtype == ["snp", "indel"] and v.id == ["chr1-100-A/G", "chr1-100-A/AT"] and v.afs == [0.1, 0.2]
    assert v.samples[0] == ({"s1"}, {"s3"}) and v.samples[1] == ({"s3"}, {"s2"})
    a, b = v.split()
    assert (a.alt, b.alt, b.position) == (["G"], ["AT"], 100) and b.samples == [({"s3"}, {"s2"})]
    assert v.split("snp")[0].id == ["chr1-100-A/G"]


G4_FIXTURES = ["g4_unphased", "g4_unphased_cpf1", "g4_unphased_dense"]


def _unphased_inputs(fx=None):
    """The inputs tests/golden/make_golden.py:g4_unphased fed the reference: rebuilt from the generator parameters the
    fixture stores."""
    from crisprhawk_hip import synth
    from crisprhawk_hip.coordinate import Coordinate
    from crisprhawk_hip.region import Region
    from crisprhawk_hip.sequence import Sequence
    from crisprhawk_hip.variant import VariantRecord
    sp = (fx or {}).get("synth") or dict(region=[4001, "chrU", 5000, 1000, 4000], variants=[4002, 40, 3],
                                         kw=dict(frac_snv=0.8, frac_del=0.1, max_indel=3, af_min=0.2, af_max=0.6))
    reg = synth.make_region(*sp["region"])
    synth.add_phased_variants(reg, *sp["variants"], **sp["kw"])
    region = Region(Sequence(reg.sequence, True), Coordinate(reg.contig, reg.bed_start, reg.bed_stop, synth.PADDING))
    recs = []
    for v in reg.variants:
        f = reg.vcf_fields(v)
        f[9:] = [g.replace("|", "/") for g in f[9:]]
        r = VariantRecord(True)
        r.read_vcf_line(f, reg.samples, False)
        recs.append(r)
    return reg, region, recs


@pytest.mark.parametrize("fixture", G4_FIXTURES)
def test_unphased_haplotype_construction_matches_reference_fixture(fixture):
    """haplotypes.add_variants_unphased (IUPAC-encoded SNV haplotypes per sample + one window haplotype set per
    indel, haplotypes.py:370-712) against the haplotypes the reference built for the same VCF records
    (g4_unphased): sequences, coordinates, samples, variants, position maps and variant_alleles.  The reference
    walks the indel carriers in set (hash) order; the mirror sorts them, so haplotypes are matched by content."""
    from crisprhawk_hip import haplotypes as H
    from crisprhawk_hip.haplotype import Haplotype
    from crisprhawk_hip.hapset import segments_from_posmap
    from crisprhawk_hip.sequence import Sequence
    from util import load_golden
    fx = load_golden(f"{fixture}.json.gz")
    reg, region, recs = _unphased_inputs(fx)
    assert reg.sequence == fx["region_seq"]
    haps = [Haplotype(Sequence(region.sequence.sequence, True), region.coordinates, False, 0, True)]
    haps = H.add_variants_unphased(haps, region, reg.samples, recs, False, True)

    def sig(seq, coord, samples, variants, breaks, n, va):
        return (seq, tuple(coord), tuple(sorted(samples.split(","))), tuple(sorted(variants.split(","))),
                tuple(map(tuple, breaks)), n, tuple(sorted((int(k), tuple(map(tuple, v))) for k, v in va.items())))

    got = []
    for h in haps:
        pm = h.segments.full()
        c = h.coordinates
        got.append(sig(h.sequence.sequence, [c.startp, c.stopp, c.start, c.stop], h.samples, h.variants,
                       [[int(a), int(b)] for a, b in zip(*segments_from_posmap(pm))], len(pm), h.variant_alleles))
    want = [sig(w["seq"], w["coord"], w["samples"], w["variants"], w["posmap_breaks"], w["posmap_len"], w["variant_alleles"])
            for w in fx["haplotypes"]]
    # REF and the SNV-only haplotypes of the whole region come first, in sample order; then one set of window haplotypes
    # per indel, in record order - within a set the reference's order is that of a Python set of carriers
    n_whole = sum(1 for w in want if w[1] == want[0][1])
    assert len(got) == len(want) and got[:n_whole] == want[:n_whole]
    assert [g[1] for g in got] == [w[1] for w in want] and sorted(got) == sorted(want)
