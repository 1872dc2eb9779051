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


@pytest.mark.parametrize("n_rows", [60, 5000], ids=["one-thread", "threaded"])
def test_segment_builder_helper_matches_numpy(n_rows):
    """hawk_host_build_segments (the library's host pass over the carried indels) against the numpy formulation it
    replaced, on random carried lists: deletions, insertions, rows aliased onto others, segments cut by the row's end."""
    from crisprhawk_hip.workload import build_segments, build_segments_numpy
    rng = np.random.default_rng(4242)
    nv, startp = 400, 1000
    r0 = np.sort(rng.choice(np.arange(10, 50_000), nv, replace=False)).astype(np.int64)
    chain = rng.choice([0, 0, 0, -1, -3, 1, 2, 5], nv).astype(np.int64)
    counts = np.concatenate(([0], rng.integers(0, 40, n_rows - 1)))
    hv_off = np.concatenate(([0], np.cumsum(counts))).astype(np.uint64)
    hv_idx = np.concatenate([np.sort(rng.choice(nv, c, replace=False)) for c in counts]).astype(np.uint32)
    hv_o = np.empty(len(hv_idx), dtype=np.int32)
    hap_len = np.empty(n_rows, dtype=np.uint32)
    for r in range(n_rows):
        a, b = int(hv_off[r]), int(hv_off[r + 1])
        ch = chain[hv_idx[a:b]]
        hv_o[a:b] = r0[hv_idx[a:b]] + np.concatenate(([0], np.cumsum(ch)[:-1])) if b > a else []
        hap_len[r] = 50_100 + int(ch.sum())
    hap_len[7] = int(hv_o[int(hv_off[7]) + 3]) + 2 if counts[7] > 3 else hap_len[7]  # a row ending inside its own list
    alias = np.arange(n_rows, dtype=np.int64)
    alias[[5, 11, 12]] = [2, 0, 11]
    ind = np.flatnonzero(chain[hv_idx] != 0)
    got = build_segments(ind, hv_idx, hv_o, hv_off, r0, chain, startp, hap_len, alias)
    want = build_segments_numpy(ind, hv_idx, hv_o, hv_off, r0, chain, startp, hap_len, alias)
    for g, w in zip(got, want):
        assert np.array_equal(np.asarray(g, dtype=np.int64), np.asarray(w, dtype=np.int64))
    assert got[0][1] == 1 and got[1][0] == 0 and got[2][0] == startp  # REF: the identity segment alone


def test_variant_table_checks_ref_alleles():
    """workload._variant_table: the REF allele of every record must match the region (haplotype.py:203-208), first base
    and the whole span of a deletion, case-insensitively."""
    from crisprhawk_hip.workload import _variant_table
    seq = "ACGTacgtNNACGTACGT"
    ok = _variant_table(np.array([101, 105, 113]), ["A", "acg", "G"], ["C", "a", "GTT"], seq, 101)
    assert ok[0].tolist() == [0, 4, 12] and ok[1].tolist() == [1, 3, 1] and ok[2].tolist() == [0, -2, 2]
    with pytest.raises(HaplotypeBuildError):
        _variant_table(np.array([102]), ["A"], ["C"], seq, 101)            # region has C there
    with pytest.raises(HaplotypeBuildError):
        _variant_table(np.array([105]), ["ACT"], ["A"], seq, 101)          # deletion: third base differs
    with pytest.raises(HaplotypeBuildError):
        _variant_table(np.array([105]), [""], ["A"], seq, 101)                 # no REF allele at all


def test_posmap_rev_helper_matches_numpy():
    from crisprhawk_hip.workload import RowMeta, build_segments_numpy
    rng = np.random.default_rng(77)
    nv, n_rows, startp = 300, 40, 5000
    r0 = np.sort(rng.choice(np.arange(10, 20_000), nv, replace=False)).astype(np.int64)
    chain = rng.choice([0, 0, -1, -4, 1, 3], nv).astype(np.int64)
    counts = np.concatenate(([0], rng.integers(0, 30, n_rows - 1)))
    hv_off = np.concatenate(([0], np.cumsum(counts))).astype(np.uint64)
    hv_idx = np.concatenate([np.sort(rng.choice(nv, c, replace=False)) for c in counts]).astype(np.uint32)
    hv_o = np.empty(len(hv_idx), dtype=np.int32)
    hap_len = np.empty(n_rows, dtype=np.uint32)
    for r in range(n_rows):
        a, b = int(hv_off[r]), int(hv_off[r + 1])
        ch = chain[hv_idx[a:b]]
        if b > a:
            hv_o[a:b] = r0[hv_idx[a:b]] + np.concatenate(([0], np.cumsum(ch)[:-1]))
        hap_len[r] = 20_100 + int(ch.sum())
    alias = np.arange(n_rows, dtype=np.int64)
    ind = np.flatnonzero(chain[hv_idx] != 0)
    ss, sr, sg = build_segments_numpy(ind, hv_idx, hv_o, hv_off, r0, chain, startp, hap_len, alias)
    m = RowMeta(ss, sr, sg, hap_len, alias, startp)
    v_del = int(hv_idx[np.flatnonzero(chain[hv_idx] < -1)[0]])      # a deletion some row carries
    deleted = startp + int(r0[v_del]) + 1                            # its first deleted base: gone from the rows that carry it
    for g in (startp, startp + 100, startp + 9_999, deleted, startp + 25_000, startp - 1):
        assert np.array_equal(m._rev_all(g), m._rev_all_numpy(g)), g
    assert (m._rev_all(deleted) == -1).any()
