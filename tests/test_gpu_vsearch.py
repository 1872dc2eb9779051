"""hawk_xplan_view: the search straight from an expansion plan - per dirty word of every row (hawk_vsearch.hip) and per
distinct variant cluster (hawk_csearch.hip) - against the search on the planes the
same plan materialises (hawk_xplan_run + hawk_search, itself held to the reference's fixtures and the oracle by
test_gpu_parity.py): every column, row for row, and the job totals, on inputs that reach each path of the kernel -
tiles with no variant, the usual few, more records than LDS stages, more dirty words than one pass holds, long
insertions and deletions across word / tile edges, rows that end in a tile, PAMs of every shape."""
import os

import numpy as np
import pytest

from crisprhawk_hip import _lib, synth
from crisprhawk_hip.workload import expand_on_device
from oracle import oracle as ora
from util import load_golden, oracle_haplotypes, synth_region_from_fixture

pytestmark = pytest.mark.gpu

COLS = ("hap", "pos", "strand", "start", "stop", "flags")


@pytest.fixture(autouse=True)
def _small_panels_take_the_cluster_path_too(monkeypatch):
    # the dictionary is only used when clusters are shared (>= 3 instances per distinct cluster): these panels have 3-6 samples
    monkeypatch.setenv("HAWK_CLUSTER_MIN_SHARE", "0")
    monkeypatch.delenv("HAWK_VIEW_SEARCH", raising=False)


def _canonical(t):
    """rows in the reference's emission order (haplotype, strand, position): the cluster path orders a haplotype's rows by
    cluster, the plane path by tile"""
    o = np.lexsort((t.pos, t.strand, t.hap))
    return {c: getattr(t, c)[o] for c in COLS} | {"win": t.win[:, o], "cfdon": t.cfdon[o]}


def _oracle_rows(reg, pam_s, guidelen, right, mm, pt):
    """the ORACLE's search of the region (haplotypes built by the oracle from the raw variant calls, util.oracle_haplotypes):
    a multiset of (carrier labels, start, stop, strand, position, window, CFDon) - rows named by WHO carries the haplotype, so the
    comparison does not depend on how either side numbers its haplotype rows"""
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, len(pam_s)) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    want = ora.search(hs, pam_s, guidelen, right)
    cfd = None
    if mm is not None:
        _, _, _, cfd, _ = ora.reverse_and_cfdon(want, hs.is_ref, guidelen, len(pam_s), mm, pt)
    g, wins = want.guides, want.windows
    lab = [tuple(sorted(h["samples"])) for h in haps]
    rows = sorted((lab[int(g["hap"][i])], int(g["start"][i]), int(g["stop"][i]), int(g["strand"][i]), int(g["pos"][i]), wins[i],
                   None if cfd is None or np.isnan(cfd[i]) else float(cfd[i])) for i in range(len(g)))
    return rows, want.n_candidates, want.n_hits


def _table_rows(t, info, kept, with_cfd):
    lab = {int(r): tuple(sorted(inf.samples)) for r, inf in zip(kept, info)}
    wins = t.windows()
    return sorted((lab[int(t.hap[i])], int(t.start[i]), int(t.stop[i]), int(t.strand[i]), int(t.pos[i]), wins[i],
                   None if not with_cfd or np.isnan(t.cfdon[i]) else float(t.cfdon[i])) for i in range(t.n_rows))


def _same_table(reg, pam_s, guidelen, right, cfd=True, na_on_ambiguous=False, oracle=False):
    bits, bitsrc, _, _ = ora.pam_encode(pam_s)
    mm, pt = synth.cfd_tables() if cfd else (None, None)
    ds, _info, _ms, _kept = expand_on_device(reg, len(pam_s), keep_plan=True)
    a = ds.search(bits, bitsrc, len(pam_s), guidelen, right, mm, pt, cfd_na_on_ambiguous=na_on_ambiguous)
    if not reg.variants:  # nothing to view
        return a
    assert ds.plan is not None
    view = ds.plan.view()
    os.environ["HAWK_VIEW_SEARCH"] = "words"
    try:
        b = view.search(bits, bitsrc, len(pam_s), guidelen, right, mm, pt, cfd_na_on_ambiguous=na_on_ambiguous)
    finally:
        del os.environ["HAWK_VIEW_SEARCH"]
    assert b.timing["v_path"] == 1
    assert (b.n_rows, b.n_candidates, b.n_hits) == (a.n_rows, a.n_candidates, a.n_hits)
    for c in COLS:
        assert np.array_equal(getattr(a, c), getattr(b, c)), c
    assert np.array_equal(a.win, b.win)
    assert np.array_equal(a.cfdon, b.cfdon, equal_nan=True)
    # per distinct cluster: the same rows, a haplotype's in cluster order
    st = ds.plan.cluster_stats()
    assert st["usable"], st
    c = view.search(bits, bitsrc, len(pam_s), guidelen, right, mm, pt, cfd_na_on_ambiguous=na_on_ambiguous)
    assert c.timing["v_path"] == 2 and c.layout() == "rows" and a.layout() == "columns"
    assert (c.n_rows, c.n_candidates, c.n_hits) == (a.n_rows, a.n_candidates, a.n_hits)
    if oracle:  # the cluster search's table against the oracle directly (not only through the plane path)
        want, n_cand, n_hits = _oracle_rows(reg, pam_s, guidelen, right, mm, pt)
        assert (c.n_candidates, c.n_hits) == (n_cand, n_hits)
        assert _table_rows(c, _info, _kept, mm is not None) == want
    ca, cc = _canonical(a), _canonical(c)
    for k in COLS:
        assert np.array_equal(ca[k], cc[k]), k
    assert np.array_equal(ca["win"], cc["win"])
    assert np.array_equal(ca["cfdon"], cc["cfdon"], equal_nan=True)
    # a second search on the view (cached REF bitmaps, reserved columns) and a different geometry after it
    b2 = view.search(bits, bitsrc, len(pam_s), guidelen, right, mm, pt, cfd_na_on_ambiguous=na_on_ambiguous)
    assert b2.n_rows == a.n_rows and np.array_equal(b2.start, c.start) and np.array_equal(b2.win, c.win)
    with pytest.raises(_lib.HawkStatusError):
        view.pam_scan(bits, bitsrc, len(pam_s))  # a view holds no planes
    ds.plan.close()
    assert b.timing["count_ms"] > 0
    return a


@pytest.mark.parametrize("case", ["phased4", "phased16", "cpf1", "indel_dense", "tiny"])
def test_view_search_on_reference_fixture_inputs(case):
    fx = load_golden(f"g3_search_{case}.json.gz")
    reg = synth_region_from_fixture(fx)
    a = _same_table(reg, fx["pam"], fx["guidelen"], fx["right"], cfd="cfdon" in fx, oracle=True)
    assert a.n_rows == len(fx["guides"])


@pytest.mark.parametrize("pam_s,guidelen,right", [("NGG", 20, False), ("TTTV", 23, True), ("NNGRRT", 21, False), ("NGN", 20, False),
                                                  ("TTTV", 23, False), ("NNNRRT", 24, True), ("NGG", 40, False), ("NG", 5, False)])
def test_view_search_pam_and_guide_shapes(pam_s, guidelen, right):
    reg = synth.make_region(8101, "chrV", 150_000, 3_000, 140_000)
    synth.add_phased_variants(reg, 8102, 2500, 6, frac_snv=0.8, frac_del=0.1, max_indel=8, af_min=0.05, af_max=0.6)
    _same_table(reg, pam_s, guidelen, right, cfd=(pam_s == "NGG" and guidelen == 20), oracle=pam_s in ("NGG", "TTTV", "NNGRRT"))


@pytest.mark.parametrize("region_len", [32_568, 32_569, 32_600, 65_336, 65_400, 70_000])
def test_view_search_rows_ending_around_tile_edges(region_len):
    reg = synth.make_region(8200 + region_len % 89, "chrB", region_len + 3000, 1200, 1200 + region_len)
    synth.add_phased_variants(reg, 8201, 90, 3, frac_snv=0.5, frac_del=0.25, af_min=0.3, af_max=0.8)
    _same_table(reg, "NGG", 20, False)
    _same_table(reg, "TTTV", 23, True, cfd=False)


def test_view_search_dense_tiles():
    # a variant every ~12 nt carried by most haplotypes: more records per tile than LDS stages (the builder reads the rest
    # from global memory) and more dirty words than one pass holds (strand 0 over all chunks, then strand 1)
    reg = synth.make_region(8301, "chrX", 75_000, 2_000, 72_000)
    synth.add_phased_variants(reg, 8302, 6000, 3, frac_snv=0.8, frac_del=0.1, max_indel=3, af_min=0.6, af_max=0.95)
    _same_table(reg, "NGG", 20, False)
    _same_table(reg, "TTTV", 23, True, cfd=False)


def test_view_search_crowded_tiles_long_deletions_long_insertion():
    reg = synth.make_region(8351, "chrK", 170_000, 5_000, 165_000)
    rng = np.random.default_rng(8352)
    seq = reg.contig_seq
    n_samples = 3
    reg.samples = [f"S{i:04d}" for i in range(n_samples)]
    sites = []

    def gt(p):
        g = (rng.random((n_samples, 2)) < p).astype(np.uint8)
        g[0, 0] = 1
        return g
    for pos in range(20_000, 60_000, 25):
        refb = seq[pos - 1]
        sites.append(synth.VariantSite(pos, refb, "ACGT"[("ACGT".index(refb) + 1) % 4], 0.7, gt(0.7)))
    for pos in (70_000, 76_000, 82_000, 88_000):
        sites.append(synth.VariantSite(pos, seq[pos - 1:pos + 1500], seq[pos - 1], 0.5, gt(0.5)))
    for pos, k in ((120_000, 400), (120_900, 31), (121_500, 32), (122_000, 33), (37_777, 70), (69_990, 45)):
        sites.append(synth.VariantSite(pos, seq[pos - 1], seq[pos - 1] + "".join("ACGT"[b] for b in rng.integers(0, 4, k)), 0.5, gt(0.5)))
    for pos in range(130_000, 160_000, 700):
        refb = seq[pos - 1]
        sites.append(synth.VariantSite(pos, refb, "ACGT"[("ACGT".index(refb) + 2) % 4], 0.3, gt(0.3)))
    sites.sort(key=lambda v: v.pos)
    reg.variants = sites
    _same_table(reg, "NGG", 20, False)
    _same_table(reg, "TTTV", 23, True, cfd=False)
    _same_table(reg, "NNGRRT", 21, False, cfd=False)


def test_view_search_variants_at_the_very_ends():
    # variants within the first / last bases of the region string: the first word's string has nothing in front of it
    reg = synth.make_region(8401, "chrE", 3000, 700, 1500)
    synth.add_phased_variants(reg, 8402, 80, 3, frac_snv=0.2, frac_del=0.4, max_indel=6, af_min=0.2, af_max=0.7, edge_margin=1)
    _same_table(reg, "NGG", 20, False)
    _same_table(reg, "NGN", 20, False)


def test_view_search_iupac_reference():
    # ambiguity codes in the REFERENCE sequence (N / R / Y ...): matched as sets by the PAM, NA under a CFD lookup
    reg = synth.make_region(8501, "chrN", 60_000, 500, 58_000, iupac_frac=0.01)
    synth.add_phased_variants(reg, 8502, 900, 4, af_min=0.1, af_max=0.6)
    _same_table(reg, "NNGRRT", 21, False, cfd=False)
    _same_table(reg, "NGG", 20, False, cfd=True, na_on_ambiguous=True)


def test_view_search_random_campaign():
    rng = np.random.default_rng(8601)
    shapes = [("NGG", 20, False), ("TTTV", 23, True), ("NNGRRT", 21, False), ("NRG", 20, False), ("TTCN", 20, True), ("NGK", 18, False)]
    for it in range(24):
        n = int(rng.integers(2_000, 120_000))
        reg = synth.make_region(8700 + it, "chrR", n + 2_500, 1_000, 1_000 + n)
        max_indel = int(rng.choice([2, 8, 40]))
        sites = min(int(n / rng.choice([15, 40, 120, 400, 2000])), n // (max_indel + 2) - 8)
        synth.add_phased_variants(reg, 8800 + it, max(sites, 2), int(rng.integers(1, 9)), frac_snv=float(rng.choice([0.3, 0.7, 0.9])),
                                  frac_del=float(rng.choice([0.05, 0.3])), max_indel=max_indel,
                                  af_min=0.05, af_max=float(rng.choice([0.3, 0.9])))
        pam_s, guidelen, right = shapes[it % len(shapes)]
        _same_table(reg, pam_s, guidelen, right, cfd=(pam_s in ("NGG", "NRG", "NGK") and not right), oracle=(it % 3 == 0 and n < 60_000))


def test_cluster_search_reruns_when_the_template_rows_outgrow_their_reservation(monkeypatch):
    # the first reservation is a guess (16 rows per distinct cluster); a dense PAM on a variant-rich panel needs more, the
    # search then produces no table, raises the reservation to the plan's bound and runs again
    reg = synth.make_region(8601, "chrT", 40_000, 1_000, 38_000)
    synth.add_phased_variants(reg, 8602, 900, 4, af_min=0.2, af_max=0.7)
    monkeypatch.setenv("HAWK_CLUSTER_ROWS0", "64")
    a = _same_table(reg, "NGN", 20, False)
    assert a.n_rows > 64


def test_cluster_dictionary_against_a_host_count():
    """the dictionary's cluster instances, counted independently on the host: a row's carried variants sorted by position, a
    new cluster wherever more than 64 reference bases separate one variant's end from the next one's start (equal to the gap
    between their alleles in the row), + one closing instance per searched row; distinct clusters = distinct variant tuples
    (clusters near a row's ends stay the row's own, so the device may count a few more)."""
    reg = synth.make_region(8701, "chrD", 400_000, 2_000, 395_000)
    synth.add_phased_variants(reg, 8702, 9000, 150, frac_snv=0.85, frac_del=0.08, max_indel=6, af_min=0.005, af_max=0.5)
    ds, info, _ms, kept = expand_on_device(reg, 3, keep_plan=True)
    ds.plan.view()
    st = ds.plan.cluster_stats()
    pos = np.array([v.pos for v in reg.variants], dtype=np.int64)
    end = pos + np.array([len(v.ref) for v in reg.variants], dtype=np.int64)
    inst, distinct = 0, set()
    for r, inf in zip(kept, info):
        idx = np.sort(np.asarray(inf.variant_idx, dtype=np.int64))
        if len(idx) == 0:
            continue  # REF
        brk = np.flatnonzero(pos[idx[1:]] - end[idx[:-1]] > 64) + 1
        for part in np.split(idx, brk):
            distinct.add(tuple(part.tolist()))
        inst += len(brk) + 2  # clusters + the closing instance
    assert st["instances"] == inst
    assert len(distinct) <= st["distinct"] <= len(distinct) + 4 * len(kept)
    assert st["usable"] and st["status"] == 0
    ds.plan.close()
    ds.close()


def test_cluster_dictionary_with_a_shared_alt_pool(monkeypatch):
    """The C ABI takes alt alleles as offsets into one code array and does not require them distinct: a caller may keep ONE copy
    of "A" for every x>A SNV.  A cluster's identity therefore has to be its locus + alleles, not the offsets (with
    the offset as identity two SNVs to the same base at different loci were merged and every instance copied the wrong rows)."""
    from crisprhawk_hip import workload
    orig = workload._variant_table

    def pooled(pos, refs, alts, seq, startp):
        r0, span, chain, altlen, alt_off, alt_codes = orig(pos, refs, alts, seq, startp)
        pool, where, codes = {}, np.zeros(len(alts), dtype=np.int64), []
        at = 0
        for i, a in enumerate(alts):
            if a not in pool:
                pool[a] = at
                codes.append(alt_codes[int(alt_off[i]):int(alt_off[i]) + int(altlen[i])])
                at += int(altlen[i])
            where[i] = pool[a]
        assert len(pool) < len(alts) // 4  # the pool really is shared
        return r0, span, chain, altlen, where, np.concatenate(codes)
    monkeypatch.setattr(workload, "_variant_table", pooled)
    reg = synth.make_region(8901, "chrP", 90_000, 2_000, 86_000)
    synth.add_phased_variants(reg, 8902, 1500, 24, frac_snv=0.9, frac_del=0.05, max_indel=4, af_min=0.05, af_max=0.6)
    _same_table(reg, "NGG", 20, False, oracle=True)
    _same_table(reg, "TTTV", 23, True, cfd=False)


def test_cluster_dictionary_numbers_with_holes():
    """A one-record shareable cluster is numbered by its variant's index; a variant that only ever travels with a neighbour within
    64 positions is nowhere a cluster of its own - its number stays a hole the search skips - and the pair is one cluster of the
    table.  Linked pairs (every carrier of the first SNV carries the second, 20 nt away), isolated SNVs and a few pairs that are
    split in some samples: same table as the oracle, and the distinct count the host works out."""
    rng = np.random.default_rng(8801)
    reg = synth.make_region(8802, "chrH", 60_000, 2_000, 56_000)
    n_samples = 12
    reg.samples = [f"S{i:04d}" for i in range(n_samples)]
    seq = reg.contig_seq.upper()
    sites, pos = [], 3_000
    for k in range(60):
        kind = k % 3  # 0: isolated SNV, 1: a linked pair, 2: a pair split in a few copies
        gt = (rng.random((n_samples, 2)) < 0.35).astype(np.uint8)
        if not gt.any():
            gt[0, 0] = 1

        def snv(p):
            ref = seq[p - 1]
            return ref, "ACGT"[("ACGT".index(ref) + 1) % 4]
        r, a = snv(pos)
        sites.append(synth.VariantSite(pos, r, a, float(gt.mean()), gt.copy()))
        if kind:
            g2 = gt.copy()
            if kind == 2:
                g2[rng.integers(0, n_samples), rng.integers(0, 2)] ^= 1
                if not g2.any():
                    g2 = gt.copy()
            r2, a2 = snv(pos + 20)
            sites.append(synth.VariantSite(pos + 20, r2, a2, float(g2.mean()), g2))
        pos += 800
    reg.variants = sites
    ds, info, _ms, kept = expand_on_device(reg, 3, keep_plan=True)
    ds.plan.view()
    st = ds.plan.cluster_stats()
    vpos = np.array([v.pos for v in reg.variants], dtype=np.int64)
    distinct = set()
    for inf in info:
        idx = np.sort(np.asarray(inf.variant_idx, dtype=np.int64))
        if len(idx):
            for part in np.split(idx, np.flatnonzero(np.diff(vpos[idx]) > 64) + 1):
                distinct.add(tuple(part.tolist()))
    assert st["usable"] and st["status"] == 0 and st["distinct"] == len(distinct)
    assert any(len(d) == 2 for d in distinct) and any(len(d) == 1 for d in distinct)
    alone = {d[0] for d in distinct if len(d) == 1}
    assert len(alone) < len(reg.variants)  # some variants are never a cluster of their own: holes
    ds.plan.close()
    ds.close()
    _same_table(reg, "NGG", 20, False, oracle=True)
    _same_table(reg, "TTTV", 23, True, cfd=False)


def test_cluster_dictionary_outgrows_its_first_hash_table():
    """The dictionary's first hash table is sized for the distinct clusters EXPECTED (an eighth of the instances); a panel of
    private variants - every cluster its own - fills it, the insert gives up and is repeated with two slots per instance."""
    reg = synth.make_region(9001, "chrU", 2_050_000, 20_000, 2_020_000)
    synth.add_phased_variants(reg, 9002, 400_000, 8, frac_snv=0.9, frac_del=0.05, max_indel=2, af_min=1 / 16, af_max=1 / 16)
    ds, _info, _ms, _kept = expand_on_device(reg, 3, keep_plan=True)
    ds.plan.view()
    st = ds.plan.cluster_stats()
    assert st["usable"] and st["status"] == 0, st
    first_table = 1 << max(16, (st["instances"] // 2 - 1).bit_length())
    assert st["distinct"] > first_table, (st, first_table)  # more distinct clusters than the first table has slots
    ds.plan.close()
    ds.close()
    _same_table(reg, "NGG", 20, False)
