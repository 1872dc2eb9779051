"""The object-level drop-in (crisprhawk_hip.search_guides.search & co.) against the reference's
own outputs, written the way the reference's tests call its API."""
import math

import numpy as np
import pytest

from crisprhawk_hip import scoring, synth
from crisprhawk_hip.annotation import reverse_guides
from crisprhawk_hip.coordinate import Coordinate
from crisprhawk_hip.crisprhawk_error import CrisprHawkCfdScoreError, CrisprHawkIupacTableError
from crisprhawk_hip.encoder import encode
from crisprhawk_hip.haplotype import Haplotype
from crisprhawk_hip.haplotypes import add_variants_phased
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.region import Region
from crisprhawk_hip.search_guides import compute_scan_start_stop, pam_search, scan_haplotype, search
from crisprhawk_hip.sequence import Sequence
from crisprhawk_hip.variant import VariantRecord
from util import G3_CASES, load_golden, synth_region_from_fixture

pytestmark = pytest.mark.gpu


def test_encode_matches_reference_tests():
    # reference tests/test_encoder.py:6-55
    assert encode("ACGTN", 0, True) == [1, 2, 4, 8, 15]
    assert encode("acgtn", 0, True) == [1, 2, 4, 8, 15]
    assert encode("", 0, True) == []
    assert list(encode("RYSWKMBDHV", 0, True)) == [5, 10, 6, 9, 12, 3, 14, 13, 11, 7]
    with pytest.raises(CrisprHawkIupacTableError):
        encode("ACGTXA", 0, True)


def _build(fx):
    reg = synth_region_from_fixture(fx)
    region = Region(Sequence(fx["region_seq"], True), Coordinate(fx["contig"], fx["bed_start"], fx["bed_stop"], 100))
    haps = [Haplotype(Sequence(region.sequence.sequence, True), region.coordinates, False, 0, True)]
    if reg.variants:
        recs = []
        for v in reg.variants:
            vr = VariantRecord(True)
            vr.read_vcf_line(reg.vcf_fields(v), reg.samples, True)
            recs.append(vr)
        haps = add_variants_phased(haps, region, reg.samples, recs, True, True)
    for i, h in enumerate(haps):
        h.id = f"hap_{i:08d}"
    return region, haps


@pytest.mark.parametrize("case", G3_CASES)
def test_search_returns_the_reference_guide_list(case):
    fx = load_golden(f"g3_search_{case}.json.gz")
    region, haps = _build(fx)
    assert [h.sequence.sequence for h in haps] == [h["seq"] for h in fx["haplotypes"]]
    assert [sorted(h.samples.split(",")) for h in haps] == [h["samples"] for h in fx["haplotypes"]]
    assert [h.variants for h in haps] == [h["variants"] for h in fx["haplotypes"]]
    pam = PAM(fx["pam"], fx["right"], True)
    pam.encode(0)
    assert [list(compute_scan_start_stop(h, region.start, region.stop, len(pam))) for h in haps] == fx["scan"]
    bits = [encode(h.sequence.sequence, 0, True) for h in haps[:2]]
    f, r = scan_haplotype(pam, bits[0], fx["scan"][0][0], fx["scan"][0][1], True)
    assert [f, r] == fx["hits"][0]
    assert [[f, r] for f, r in pam_search(pam, region, haps, None, 0, True)] == fx["hits"]
    guides = search(pam, region, haps, None, fx["guidelen"], fx["right"], fx["variants_present"], fx["phased"], 0, True)
    hidx = {h.id: i for i, h in enumerate(haps)}
    got = [[g.start, g.stop, g.strand, g.sequence, hidx[g.hapid], g.right] for g in guides]
    assert got == fx["guides"]
    for i, pm in fx["guide_posmaps"]:
        assert [guides[i].posmap[k] for k in range(fx["guidelen"] + len(pam))] == pm
    guides = reverse_guides(guides, 0)
    assert [[g.sequence, g.guide, g.pam, g.right] for g in guides] == fx["reversed"]
    assert scoring._extract_guide_sequences(guides) == fx["kmers"]
    if "cfdon" in fx:
        scoring.set_cfd_tables(*synth.cfd_tables())
        ids = {id(g): i for i, g in enumerate(guides)}
        scored = scoring.cfdon_score(guides, 0, True)
        assert [ids[id(g)] for g in scored] == fx["cfdon_order"]
        for g, want in zip(scored, fx["cfdon"]):
            assert g.cfdon_score == ("NA" if want is None else str(round(want, 4)))


def test_cfd_batch_bit_exact_and_errors():
    g5 = load_golden("g5_cfd.json.gz")
    scoring.set_cfd_tables(*synth.cfd_tables(g5["seed"]))
    by_len = {}
    for wt, sg, pam, want in g5["cases"]:
        by_len.setdefault(len(wt), []).append((wt, sg, pam, want))
    for ln, cases in by_len.items():
        got = scoring.compute_cfd_batch([c[0] for c in cases], [c[1] for c in cases], [c[2] for c in cases], True)
        assert got.tolist() == [c[3] for c in cases]
    with pytest.raises(CrisprHawkCfdScoreError):
        scoring.compute_cfd_batch(["ACGTN"], ["ACGTA"], ["GG"], True)
    with pytest.raises(CrisprHawkCfdScoreError):
        scoring.compute_cfd_batch(["ACGT"], ["ACGT"], ["NG"], True)


def test_deepcpf1_kernel_against_reference_forward():
    g6 = load_golden("g6_deepcpf1.json.gz")  # SeqDeepCpf1 (torch, fp32) run by the reference, seeded weights
    scoring.set_deepcpf1_weights(synth.deepcpf1_weights(g6["seed"]))
    got = np.array(scoring.deepcpf1(g6["seqs"]))
    assert np.max(np.abs(got - np.array(g6["scores"]))) < 1e-6  # fp32 tolerance stated by north_star
    assert scoring.deepcpf1([s.lower() for s in g6["seqs"][:3]]) == pytest.approx(g6["scores"][:3], abs=1e-6)
    from crisprhawk_hip.crisprhawk_error import CrisprHawkDeepCpf1ScoreError
    with pytest.raises(CrisprHawkDeepCpf1ScoreError):
        scoring.deepcpf1(["ACGT" * 8 + "NN"])


def _random_gbt(rng, n_trees=100, nfeat=627):
    """depth-3 regression trees in the flattened layout (complete trees, 15 nodes each)"""
    off, feat, left, right, thr, val = [0], [], [], [], [], []
    for _ in range(n_trees):
        for k in range(15):
            if k < 7:
                f = int(rng.integers(0, nfeat))
                feat.append(f); left.append(2 * k + 1); right.append(2 * k + 2)
                thr.append(float(rng.uniform(-40, 70)) if f >= 623 else (float(rng.integers(0, 20)) + 0.5 if f in (606,) or 120 <= f < 124 or 588 <= f < 604 else 0.5))
            else:
                feat.append(-1); left.append(0); right.append(0); thr.append(-2.0)
            val.append(float(rng.normal()))
        off.append(len(feat))
    return dict(tree_off=np.array(off, np.int32), feature=np.array(feat, np.int32), left=np.array(left, np.int32),
                right=np.array(right, np.int32), threshold=np.array(thr), value=np.array(val), init=0.37, learning_rate=0.1)


def test_azimuth_features_and_trees_against_oracle():
    from oracle import oracle as ora
    rng = np.random.default_rng(31)
    seqs = [synth.random_sequence(rng, 30) for _ in range(700)] + ["CCCCAGCTTAGCTAGCTAGCTAGCTAGCCC"]  # reference tests/test_scoring.py:55
    model = _random_gbt(rng)
    scoring.set_azimuth_model(model)
    got, feats = scoring.azimuth(seqs, return_features=True)
    want_f = ora.azimuth_features(seqs)
    assert np.array_equal(feats[:, :623], want_f[:, :623])                 # one-hots, counts, GC, NGGX: exact
    assert np.max(np.abs(feats[:, 623:] - want_f[:, 623:])) < 1e-9         # Tm: fp64, device log vs libm
    assert np.max(np.abs(np.array(got) - ora.gbt_predict(want_f, model))) < 1e-9
    from crisprhawk_hip.crisprhawk_error import CrisprHawkAzimuthScoreError
    with pytest.raises(CrisprHawkAzimuthScoreError):
        scoring.azimuth(["ACGTN" * 6])


def test_azimuth_against_sklearn_gbr():
    sk = pytest.importorskip("sklearn.ensemble")
    from oracle import oracle as ora
    rng = np.random.default_rng(32)
    train = [synth.random_sequence(rng, 30) for _ in range(400)]
    X = ora.azimuth_features(train)
    y = X[:, 606] * 0.03 + X[:, 623] * 0.01 + X[:, 4] - X[:, 130] + rng.normal(0, 0.1, len(train))
    gbr = sk.GradientBoostingRegressor(n_estimators=100, max_depth=3, learning_rate=0.1, random_state=1).fit(X, y)  # models/ensembles.py:28-30
    scoring.set_azimuth_model(gbr)
    test = [synth.random_sequence(rng, 30) for _ in range(300)]
    got = np.array(scoring.azimuth(test))
    assert np.max(np.abs(got - gbr.predict(ora.azimuth_features(test)))) < 1e-9


G4_FIXTURES = ["g4_unphased", "g4_unphased_cpf1", "g4_unphased_dense"]


@pytest.mark.parametrize("fixture", G4_FIXTURES)
def test_unphased_search_resolves_iupac_like_the_reference(fixture):
    """haplotypes built by the reference from an unphased VCF (lower-case IUPAC letters at
    heterozygous SNVs, one window haplotype per indel) -> search() with resolve_guide expansion."""
    from util import posmap_from_breaks
    fx = load_golden(f"{fixture}.json.gz")
    region = Region(Sequence(fx["region_seq"], True), Coordinate(fx["contig"], fx["bed_start"], fx["bed_stop"], 100))
    haps = []
    for i, gh in enumerate(fx["haplotypes"]):
        sp, ep, s0, e0 = gh["coord"]  # window haplotypes carry their own coordinates (padding 0)
        coord = Coordinate(gh["contig"], sp, ep, sp - s0)
        assert (coord.start, coord.stop) == (s0, e0)
        h = Haplotype(Sequence(gh["seq"], True, allow_lower_case=True), coord, False, 0, True)
        h.samples, h.variants = gh["samples"], gh["variants"]
        h.set_posmap({k: int(v) for k, v in enumerate(posmap_from_breaks(gh["posmap_breaks"], gh["posmap_len"]))})
        h.set_variant_alleles({int(k): [tuple(t) for t in v] for k, v in gh["variant_alleles"].items()})
        h.id = f"hap_{i:08d}"
        haps.append(h)
    pam = PAM(fx["pam"], fx["right"], True)
    pam.encode(0)
    assert [list(compute_scan_start_stop(h, region.start, region.stop, len(pam))) for h in haps] == fx["scan"]
    assert [[f, r] for f, r in pam_search(pam, region, haps, None, 0, True)] == fx["hits"]
    guides = search(pam, region, haps, None, fx["guidelen"], fx["right"], True, False, 0, True)
    hidx = {h.id: i for i, h in enumerate(haps)}
    got = [[g.start, g.stop, g.strand, g.sequence, hidx[g.hapid], g.right, g.samples] for g in guides]
    assert got == fx["guides"]


@pytest.mark.parametrize("fixture", G4_FIXTURES)
def test_unphased_vcf_records_to_guides_end_to_end(fixture):
    """Unphased VCF records -> haplotypes.add_variants_unphased (mirror) -> device search -> resolve_guide: the
    guide list the reference produced from its own haplotypes for the same records (g4_unphased)."""
    from test_host_objects import _unphased_inputs
    from crisprhawk_hip import haplotypes as H
    fx = load_golden(f"{fixture}.json.gz")
    reg, region, recs = _unphased_inputs(fx)
    haps = [Haplotype(Sequence(region.sequence.sequence, True), region.coordinates, False, 0, True)]
    haps = H.add_variants_unphased(haps, region, reg.samples, recs, False, True)
    for i, h in enumerate(haps):
        h.id = f"hap_{i:08d}"
    pam = PAM(fx["pam"], fx["right"], True)
    pam.encode(0)
    guides = search(pam, region, haps, None, fx["guidelen"], fx["right"], True, False, 0, True)
    got = sorted([g.start, g.stop, g.strand, g.sequence, ",".join(sorted(g.samples.split(","))), g.right] for g in guides)
    want = sorted([g[0], g[1], g[2], g[3], ",".join(sorted(g[6].split(","))), g[5]] for g in fx["guides"])
    assert got == want
