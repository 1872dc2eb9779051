"""C4's mechanism at test size: a region searched tile by tile (crisprhawk_hip.tiling) must give exactly the report
groups of the same region searched in one piece - the untiled CPU oracle on one side (search + CFDon + the report's
grouping), the per-tile device pipeline (hawk_xplan_run -> hawk_search -> hawk_table_collapse_ex ->
hawk_table_collapse_export, seam merge on the host) on the other.  Seams are dense (every 2-3 kb) and the variant
panel is indel-rich, so seams cut through guides, through indels and between a haplotype guide and its REF partner."""
import numpy as np
import pytest

from crisprhawk_hip import reports, synth
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.tiling import TiledRegionSearch, VariantPanel, plan_tiles
from crisprhawk_hip.workload import expand_on_device, row_labels
from oracle import oracle as ora
from util import oracle_haplotypes

pytestmark = pytest.mark.gpu


def _oracle_groups(reg, pam_s, guidelen, right, mm=None, pt=None):
    """Untiled oracle: {(start, stop, strand, origin, core): (collapsed samples string, cfdon of the first member)}"""
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, len(pam_s)) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    res = ora.search(hs, pam_s, guidelen, right)
    g = res.guides
    cfd = None
    if mm is not None:
        _, _, _, cfd_o, order = ora.reverse_and_cfdon(res, hs.is_ref, guidelen, len(pam_s), mm, pt)
        cfd = np.full(len(g), np.nan)
        cfd[:] = cfd_o  # scores are returned per guide in the input order
    isref_row = np.asarray(hs.is_ref)[g["hap"]]
    groups, _gc = ora.collapse_rows(g["start"], g["stop"], g["strand"], isref_row, res.windows, guidelen, len(pam_s), right)
    out = {}
    for key, rows in groups.items():
        labels = [",".join(haps[int(g["hap"][r])]["samples"]) for r in rows]
        out[key] = (reports.collapse_samples(labels), None if cfd is None else cfd[rows[0]])
    return out, res


def _tiled_groups(mg):
    c = mg.cols
    L = mg.guidelen + mg.pamlen
    wins = [w[10:10 + L] for w in __import__("crisprhawk_hip.hapset", fromlist=["decode_windows"]).decode_windows(
        np.ascontiguousarray(c["win"].T), L + 20)]
    out = {}
    for gi in range(mg.n_groups):
        mem = mg.members[mg.member_off[gi]:mg.member_off[gi + 1]]
        labels = [mg.labels[int(h)].samples for h in mem]
        key = (int(c["start"][gi]), int(c["stop"][gi]), int(c["strand"][gi]), bool(c["origin"][gi]), wins[gi])
        assert key not in out, key
        out[key] = (reports.collapse_samples(labels), c["cfdon"][gi])
    return out


def _same(a, b):
    return (a != a and b != b) or a == b


@pytest.mark.parametrize("pam_s,guidelen,right,tile_nt,seed", [("NGG", 20, False, 3000, 1), ("NGG", 20, False, 2048, 2),
                                                              ("TTTV", 23, True, 2500, 3), ("NNGRRT", 21, False, 4000, 4)])
def test_tiled_region_equals_untiled_oracle(pam_s, guidelen, right, tile_nt, seed):
    reg = synth.make_region(4400 + seed, "chrT", 40_000, 1_000, 37_000)
    synth.add_phased_variants(reg, 4500 + seed, 1100, 8, frac_snv=0.5, frac_del=0.25, af_min=0.15, af_max=0.6)
    score = pam_s == "NGG"
    mm, pt = synth.cfd_tables() if score else (None, None)
    want, res = _oracle_groups(reg, pam_s, guidelen, right, mm, pt)
    pam = PAM(pam_s, right, True)
    pam.encode(0)
    panel = VariantPanel.from_region(reg)
    trs = TiledRegionSearch(lambda lo, hi: reg.contig_seq[lo - 1:hi], reg.contig, reg.startp, reg.stopp, panel, pam, guidelen, right,
                            tile_nt=tile_nt, flank=300)
    assert len(trs.tiles) >= 8
    # seams must really cut through variants' neighbourhoods
    seams = [t.own_lo for t in trs.tiles[1:]]
    near = sum(1 for s in seams if np.any(np.abs(panel.pos - s) < guidelen + len(pam_s)))
    assert near >= len(seams) // 2
    mg = trs.run(cfd=(mm, pt) if score else None)
    got = _tiled_groups(mg)
    assert len(got) == len(want)
    for key, (samples, cfd) in want.items():
        assert key in got, key
        assert got[key][0] == samples, key
        if score:
            assert _same(got[key][1], cfd), key
    # every oracle guide row sits in exactly one tile's table: per-tile row totals, after the report's own merge of
    # identical rows, cannot be compared row by row (haplotype identity is per tile) - the groups above are the contract
    assert sum(st["groups"] for st in mg.stats) >= len(want)


def test_tiled_report_equals_untiled_device_report():
    """The TSV the report assembler writes from tile-merged groups equals the one from the one-piece device search
    (itself pinned byte for byte to the reference by the g7 fixtures), haplotype ids aside."""
    reg = synth.make_region(4471, "chrU", 30_000, 1_000, 27_000)
    synth.add_phased_variants(reg, 4472, 500, 6, frac_snv=0.6, frac_del=0.2, af_min=0.2, af_max=0.6)
    pam = PAM("NGG", False, True)
    pam.encode(0)
    mm, pt = synth.cfd_tables()
    ds, info, _ms, kept = expand_on_device(reg, 3)
    tab = ds.search(pam.bits, pam.bitsrc, 3, 20, False, mm, pt, download=False, collapse=True)
    target = f"{reg.contig}:{reg.bed_start}-{reg.bed_stop}"
    df1 = reports.report_frame(reports.ReportInput.from_table(tab), row_labels(reg, ds, info, kept), pam, reg.contig, target)
    trs = TiledRegionSearch(lambda lo, hi: reg.contig_seq[lo - 1:hi], reg.contig, reg.startp, reg.stopp, VariantPanel.from_region(reg),
                            pam, 20, False, tile_nt=2600, flank=300)
    mg = trs.run(cfd=(mm, pt))
    df2 = reports.report_frame(mg.report_input(), mg.labels, pam, reg.contig, target)
    df3 = reports.report_from_groups(mg.groups(), mg.labels, pam, reg.contig, target)
    cols = [c for c in df1.columns if c != "haplotype_id"]
    assert len(df1) == len(df2) and len(df1) > 1000
    assert df1[cols].to_csv(sep="\t", index=False) == df2[cols].to_csv(sep="\t", index=False)
    assert df2.to_csv(sep="\t", index=False) == df3.to_csv(sep="\t", index=False)


def test_tiles_with_n_run_and_no_variants():
    """A region that starts with an N run (C4's leading N block in miniature): every N position is a PAM hit on both
    strands of REF (nibble 15 matches any PAM base, encoder.py:18-34), N-bearing guides score NA under the lenient
    CFD mode, and variant-free tiles run as REF-only sets."""
    rng = np.random.default_rng(4481)
    seq = "N" * 6000 + synth.random_sequence(rng, 14_000)
    reg = synth.SynthRegion("chrN", seq, 500, 19_000)
    pam = PAM("NGG", False, True)
    pam.encode(0)
    mm, pt = synth.cfd_tables()
    hs = ora.HapSet([reg.sequence], [np.arange(reg.startp, reg.startp + len(reg.sequence), dtype=np.int64)], [True],
                    [ora.scan_bounds(np.arange(reg.startp, reg.startp + len(reg.sequence), dtype=np.int64), reg.startp, reg.stopp, 3)])
    want = ora.search(hs, "NGG", 20, False)
    trs = TiledRegionSearch(lambda lo, hi: reg.contig_seq[lo - 1:hi], reg.contig, reg.startp, reg.stopp, None, pam, 20, False,
                            tile_nt=4096, flank=256)
    mg = trs.run(cfd=(mm, pt), cfd_na_on_ambiguous=True)
    assert mg.n_groups == len(want.guides) and sum(st["candidates"] for st in mg.stats) == want.n_candidates
    key = lambda s, e, d: (int(s), int(e), int(d))
    got = {key(s, e, d) for s, e, d in zip(mg.cols["start"], mg.cols["stop"], mg.cols["strand"])}
    assert got == {key(s, e, d) for s, e, d in zip(want.guides["start"], want.guides["stop"], want.guides["strand"])}
    # guides inside the N run have no defined CFD: NA, not an error
    in_n = mg.cols["stop"] < reg.startp + 5000
    assert in_n.any() and np.isnan(mg.cols["cfdon"][in_n]).all()
    assert (~np.isnan(mg.cols["cfdon"][mg.cols["start"] > 7000])).all()


def test_plan_tiles_covers_the_interval_once():
    for startp, stopp, tile, flank in [(901, 37_100, 3000, 300), (1, 1_000_201, 250_000, 1024), (10, 5000, 10_000, 100)]:
        tiles = plan_tiles(startp, stopp, tile, flank)
        assert tiles[0].seq_lo == startp and tiles[-1].seq_hi == stopp
        assert tiles[0].own_lo is None and tiles[-1].own_hi is None
        for a, b in zip(tiles, tiles[1:]):
            assert a.own_hi == b.own_lo and a.seq_hi >= a.own_hi + min(flank, stopp - a.own_hi) and b.seq_lo <= b.own_lo - min(flank, b.own_lo - startp)


def test_dense_carried_variants_overflow_the_staged_range():
    """More than HX_MAXV = 192 carried variants inside one expansion workgroup's 8192 output positions:
    the variants beyond the LDS-staged range are read from global memory; planes must still equal the host build."""
    from crisprhawk_hip.hapset import DeviceHapSet
    from crisprhawk_hip.workload import build_phased_haplotypes
    reg = synth.make_region(4491, "chrD", 30_000, 1_000, 28_000)
    synth.add_phased_variants(reg, 4492, 2400, 2, frac_snv=0.7, frac_del=0.15, af_min=0.9, af_max=0.99)
    per_8k = 8192 * len(reg.variants) * 0.9 / (reg.stopp - reg.startp)
    assert per_8k > 400
    haps, _ = build_phased_haplotypes(reg, 3)
    ds, info, _ms, kept = expand_on_device(reg, 3)
    host = DeviceHapSet(haps)
    pd_, ph = ds.planes(), host.planes()
    assert len(kept) == len(haps)
    for i, r in enumerate(kept):
        n = (int(host.hap_len[i]) + 31) // 32
        assert ds.hap_len[r] == host.hap_len[i]
        assert np.array_equal(pd_[:, r, :n], ph[:, i, :n]), (i, r)


def test_tiled_equals_one_piece_at_megabase_scale():
    """C4's mechanism at a size the oracle does not reach (1.5 Mb x 300 samples, 46 k sites, four tiles, block-generated
    genotypes): the tiled pipeline's report must equal the one-piece device search's report row for row - every column but
    the haplotype ids, which name per-tile rows - and the per-tile candidate counts must add up to the one-piece count
    over REF (the one haplotype every tile shares unchanged)."""
    from crisprhawk_hip.workload import hap_labels
    seq, panel = synth.contig_panel(4601, "chrW", 1_500_000, 0, 300, sites_per_mb=31_000)
    contig = seq.tobytes().decode()
    startp, stopp = 1, 1_500_000
    pam = PAM("NGG", False, True)
    pam.encode(0)
    mm, pt = synth.cfd_tables()
    fetch = lambda lo, hi: seq[lo - 1:hi]
    target = "chrW:101-1499900"
    trs = TiledRegionSearch(fetch, "chrW", startp, stopp, panel, pam, 20, False, tile_nt=400_000)
    assert len(trs.tiles) == 4
    mg = trs.run(cfd=(mm, pt))
    df_t = reports.report_from_groups(mg.groups(), mg.labels, pam, "chrW", target)
    # one piece: the same panel as a single tile spanning the region
    one = TiledRegionSearch(fetch, "chrW", startp, stopp, panel, pam, 20, False, tile_nt=10_000_000)
    assert len(one.tiles) == 1
    mg1 = one.run(cfd=(mm, pt))
    df_1 = reports.report_from_groups(mg1.groups(), mg1.labels, pam, "chrW", target)
    assert len(df_1) == len(df_t) > 300_000
    cols = [c for c in df_1.columns if c != "haplotype_id"]
    for c in cols:
        assert (df_1[c].values == df_t[c].values).all(), c
    ref_rows_t = int((df_t["origin"] == "ref").sum())
    assert ref_rows_t == int((df_1["origin"] == "ref").sum())
    assert sum(st["rows"] for st in mg.stats) >= sum(st["rows"] for st in mg1.stats) * 0.99


def test_soft_masked_reference_is_upper_cased_like_the_reference_does():
    """A genome FASTA with lower-case (soft-masked) stretches: the reference upper-cases every region it reads
    (sequence.py:49), so a tiled search over a fetch() that returns the masked text must give the groups of the unmasked one."""
    reg = synth.make_region(4491, "chrM", 16_000, 1_000, 14_000)
    synth.add_phased_variants(reg, 4492, 200, 4, frac_snv=0.7, frac_del=0.15, af_min=0.2, af_max=0.6)
    pam = PAM("NGG", False, True)
    pam.encode(0)
    masked = reg.contig_seq[:3000] + reg.contig_seq[3000:9000].lower() + reg.contig_seq[9000:]
    out = []
    for text in (reg.contig_seq, masked):
        trs = TiledRegionSearch(lambda lo, hi, t=text: t[lo - 1:hi], reg.contig, reg.startp, reg.stopp, VariantPanel.from_region(reg), pam, 20, False,
                                tile_nt=4000, flank=400)
        mg = trs.run(cfd=synth.cfd_tables())
        out.append(_tiled_groups(mg))
    assert len(out[0]) > 500 and out[0].keys() == out[1].keys()
    assert all(out[0][k][0] == out[1][k][0] and _same(out[0][k][1], out[1][k][1]) for k in out[0])


def test_record_longer_than_the_flank_at_a_seam_is_refused():
    """a record that only partly lies in a tile string is left to the neighbour that holds it whole - fine while
    it stays clear of what the tile scans.  A deletion longer than the flank that straddles the edge of a tile STRING and
    reaches into the tile's own range would map the seam differently on its two sides: prepare_tile refuses it."""
    reg = synth.make_region(9901, "chrL", 24_000, 1_000, 22_000)
    seq = reg.contig_seq
    reg.samples = ["S0000", "S0001"]
    gt = np.array([[1, 0], [0, 1]], dtype=np.uint8)
    pam = PAM("NGG", False, True)
    pam.encode(0)
    trs0 = TiledRegionSearch(lambda lo, hi: seq[lo - 1:hi], reg.contig, reg.startp, reg.stopp, VariantPanel.from_region(reg), pam, 20, False,
                             tile_nt=4000, flank=200)
    t1 = trs0.tiles[1]
    # a 300-nt deletion starting before tile 1's string and ending inside its flank, 40 nt short of its own range: refused
    pos = t1.seq_lo - 100
    span = t1.own_lo - 40 - pos
    assert span > trs0.flank
    reg.variants = [synth.VariantSite(pos, seq[pos - 1:pos - 1 + span + 1], seq[pos - 1], 0.5, gt)]
    trs = TiledRegionSearch(lambda lo, hi: seq[lo - 1:hi], reg.contig, reg.startp, reg.stopp, VariantPanel.from_region(reg), pam, 20, False,
                            tile_nt=4000, flank=200)
    with pytest.raises(ValueError, match="flank is too small"):
        trs.prepare_tile(1)
    # the same deletion with a flank that holds it: every tile prepares, and the tiled groups are the one-piece groups
    trs = TiledRegionSearch(lambda lo, hi: seq[lo - 1:hi], reg.contig, reg.startp, reg.stopp, VariantPanel.from_region(reg), pam, 20, False,
                            tile_nt=4000, flank=1024)
    mg = trs.run()
    want, _res = _oracle_groups(reg, "NGG", 20, False)
    got = _tiled_groups(mg)
    assert set(got) == set(want)
    # a short record cut by the far edge of a tile string (inside the flank, far from the own range) is simply left out
    pos = trs0.tiles[0].seq_hi - 2
    reg.variants = [synth.VariantSite(pos, seq[pos - 1:pos + 5], seq[pos - 1], 0.5, gt)]
    trs = TiledRegionSearch(lambda lo, hi: seq[lo - 1:hi], reg.contig, reg.startp, reg.stopp, VariantPanel.from_region(reg), pam, 20, False,
                            tile_nt=4000, flank=200)
    mg = trs.run()
    want, _res = _oracle_groups(reg, "NGG", 20, False)
    assert set(_tiled_groups(mg)) == set(want)
