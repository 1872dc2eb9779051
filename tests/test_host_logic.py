"""Host-side logic of the product package against reference-generated vectors (no GPU):
PAM class, haplotype construction on segments, scan bounds."""
import numpy as np
import pytest

from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import build_phased_haplotypes
from crisprhawk_hip.expand import expand_haplotype, HaplotypeBuildError
from util import G3_CASES, load_golden, posmap_from_breaks, synth_region_from_fixture


def test_pam_class_matches_reference():
    g1 = load_golden("g1_tables.json.gz")
    for p in g1["pams"]:
        pm = PAM(p["pam"], p["right"], True)
        pm.encode(0)
        assert (pm.pam, pm.pamrc, pm.bits, pm.bitsrc, pm.bits_list, pm.cas_system, len(pm)) == (
            p["seq"], p["rc"], p["bits"], p["bitsrc"], p["bits_list"], p["cas_system"], p["length"])
    with pytest.raises(ValueError):
        PAM("NGX", False, True)
    assert str(PAM("ngg", False, True)) == "NGG"


@pytest.mark.parametrize("case", G3_CASES)
def test_haplotype_construction_on_segments(case):
    fx = load_golden(f"g3_search_{case}.json.gz")
    reg = synth_region_from_fixture(fx)
    haps, info = build_phased_haplotypes(reg, len(fx["pam"]))
    assert len(haps) == len(fx["haplotypes"])
    for h, i, gold, sc in zip(haps, info, fx["haplotypes"], fx["scan"]):
        assert bytes(h.seq).decode() == gold["seq"]
        assert sorted(i.samples) == gold["samples"]
        assert np.array_equal(h.seg.full(), posmap_from_breaks(gold["posmap_breaks"], gold["posmap_len"]))
        for g, rel in gold["posmap_rev_probe"]:
            assert h.seg.rev(g) == rel
        assert list(h.scan) == sc
        assert h.is_ref == (gold["samples"] == ["REF"])


def test_expand_errors():
    ref = np.frombuffer(b"ACGTACGTACGTACGTACGT", dtype=np.uint8)
    with pytest.raises(HaplotypeBuildError):  # reference ValueError: mismatching REF allele
        expand_haplotype(ref, 100, [(103, b"A", b"G")])
    with pytest.raises(HaplotypeBuildError):  # second variant sits on a deleted position
        expand_haplotype(ref, 100, [(103, b"TACG", b"T"), (105, b"C", b"CA")])
    out, seg = expand_haplotype(ref, 100, [(105, b"CGT", b"C")])  # SURVEY.md §7 probe
    assert out.tobytes() == b"ACGTAcACGTACGTACGT" and seg.full().tolist()[4:8] == [104, 105, 108, 109]
    out, seg = expand_haplotype(ref, 100, [(105, b"C", b"CAA")])
    assert out.tobytes() == b"ACGTAcaaGTACGTACGTACGT" and seg.full().tolist()[4:9] == [104, 105, 105, 105, 106]
    assert seg.rev(105) == 7
