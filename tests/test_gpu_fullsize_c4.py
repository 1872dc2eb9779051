"""One FULL-SIZE tile of BASELINE.json's C4 (whole chr22 x 2504 phased samples, region-tiled): 4 Mb x 5009 haplotype rows,
an all-N stretch at its start, 1000G variant density behind it - 2 x 10^10 scanned positions, ~10^8 guide rows.  The step
the bench runs per tile (search straight from the plan -> collapse) is held to size-independent properties and to the
oracle on haplotypes cut out of the full table; the materialised path (planes written, then searched) must give the same
totals and the same report groups.  ~40 s on an MI355X, most of it synthesising the panel on the host."""
import ctypes as C

import numpy as np
import pytest

from crisprhawk_hip import _lib, synth
from crisprhawk_hip.expand import expand_haplotype, scan_bounds
from crisprhawk_hip.hapset import _p
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.tiling import TiledRegionSearch
from oracle import oracle as ora

pytestmark = pytest.mark.gpu

PAM_S, GUIDELEN = "NGG", 20
TILE_NT, N_BLOCK, N_SAMPLES = 4_000_000, 300_000, 2504


@pytest.fixture(scope="module")
def c4():
    contig_len = TILE_NT + 200  # one tile: BED 100 .. contig_len - 100
    seq, panel = synth.contig_panel(1004, "chr22", contig_len, N_BLOCK, N_SAMPLES)
    pam = PAM(PAM_S, False, True)
    pam.encode(0)
    trs = TiledRegionSearch(lambda lo, hi: seq[lo - 1:hi], "chr22", 1, contig_len, panel, pam, GUIDELEN, False, tile_nt=2 * TILE_NT)
    assert len(trs.tiles) == 1
    pt = trs.prepare_tile(0, keep_plan=True)
    assert pt.n_hap == 2 * N_SAMPLES + 1  # no two chromosome copies agree at this density: REF + 5008 rows
    mm, ptab = synth.cfd_tables()
    view = pt.plan.view()
    tab = view.search(pam.bits, pam.bitsrc, 3, GUIDELEN, False, mm, ptab, download=False, collapse=True, cfd_na_on_ambiguous=True)
    assert tab.timing["v_path"] == 2 and tab.layout() == "rows"  # the bench's per-tile step: the cluster search, not a fall-back
    groups = tab.export_groups()
    tab.download()
    return dict(seq=seq, panel=panel, pam=pam, trs=trs, pt=pt, view=view, tab=tab, groups=groups, mm=mm, ptab=ptab)


def test_c4_tile_totals_rows_and_order(c4):
    tab, pt = c4["tab"], c4["pt"]
    n_ref_len = len(c4["seq"])
    assert tab.n_rows == len(tab.hap) > 50_000_000
    assert (np.diff(tab.hap.astype(np.int64)) >= 0).all()  # haplotype-major
    key = (tab.hap.astype(np.uint64) << np.uint64(33)) | (tab.strand.astype(np.uint64) << np.uint64(32)) | tab.pos.astype(np.uint64)
    assert (np.diff(key.view(np.int64)) != 0).all() and len(np.unique(key[:: 7])) == len(key[:: 7])
    # the N stretch: every position of REF is a PAM hit there (N matches everything, search_guides.py:32-46) and a REF row
    isref = tab.hap == 0
    ref_pos = tab.pos[isref & (tab.strand == 0)]
    in_n = ref_pos < N_BLOCK - 40
    assert in_n.sum() >= N_BLOCK - 200 and np.isnan(tab.cfdon[isref & (tab.strand == 0)][in_n]).all()  # scorers give NA on N
    # non-REF rows never lie in the N stretch: it carries no variant, and a window without a variant base is dropped
    assert tab.pos[~isref].min() > N_BLOCK
    assert tab.n_candidates <= tab.n_hits and tab.n_hits > 5008 * (n_ref_len - N_BLOCK) // 9


def test_c4_tile_ref_partners_and_cfdon(c4):
    tab = c4["tab"]
    isref = tab.hap == 0
    k = tab.start.astype(np.int64) * 2 + tab.strand
    ref_keys = np.unique(k[isref])
    assert isref.sum() == len(ref_keys)
    has = np.isin(k, ref_keys)
    assert np.array_equal(tab.flags & 1, has.astype(np.uint8))
    nan = np.isnan(tab.cfdon)
    assert not nan[has & (tab.start > N_BLOCK + 100)].any() and nan[~has].all()
    ok = ~nan
    assert (tab.cfdon[ok] >= 0).all() and (tab.cfdon[ok] <= 1).all()


def test_c4_tile_materialised_path_gives_the_same_totals_and_groups(c4):
    """hawk_xplan_run + hawk_search over 12.5 GB of planes vs the search from the plan: totals, rows, report groups."""
    pt, pam, tab, g = c4["pt"], c4["pam"], c4["tab"], c4["groups"]
    ds, _, _ = pt.plan.run()
    try:
        t2 = ds.search(pam.bits, pam.bitsrc, 3, GUIDELEN, False, c4["mm"], c4["ptab"], download=False, collapse=True,
                       cfd_na_on_ambiguous=True)
        assert (t2.n_rows, t2.n_candidates, t2.n_hits, t2.n_groups) == (tab.n_rows, tab.n_candidates, tab.n_hits, tab.n_groups)
        g2 = t2.export_groups()
        # (rep_row is a row index: the two tables order a haplotype's rows differently - by cluster, by tile; the representative
        # is the group's first member either way)
        for col in ("pos", "strand", "start", "stop", "flags", "member_hap", "member_off", "gc_num", "gc_den"):
            assert np.array_equal(getattr(g, col), getattr(g2, col)), col
        assert np.array_equal(g.win, g2.win) and np.array_equal(g.cfdon, g2.cfdon, equal_nan=True)
        # the independent scan kernel on the materialised planes counts the same PAM hits
        off_f = np.zeros(ds.n_hap + 1, dtype=np.uint64)
        off_r = np.zeros(ds.n_hap + 1, dtype=np.uint64)
        rc = ds._L.hawk_pam_scan(ds._h, C.c_uint64(pam.bits), C.c_uint64(pam.bitsrc), 3, None, None, C.c_uint64(0), C.c_uint64(0),
                                 _p(off_f), _p(off_r))
        assert rc in (_lib.HAWK_OK, _lib.HAWK_E_CAPACITY)
        assert int(off_f[-1]) + int(off_r[-1]) == tab.n_hits
        t2.close()
    finally:
        ds.close()


@pytest.mark.parametrize("sample", [3, 1777])
def test_c4_tile_sampled_haplotypes_match_the_oracle(c4, sample):
    """Both chromosome copies of one sample, cut out of the 10^8-row table, against the oracle on REF + those copies built on
    the host from the same calls - restricted to the 300 kb behind the N stretch so that the oracle finishes in seconds
    (rows are compared where the oracle's region and the tile agree: PAM positions inside that window's BED interval)."""
    seq, panel, tab = c4["seq"], c4["panel"], c4["tab"]
    lo, hi = N_BLOCK + 2_000, N_BLOCK + 302_000  # BED interval of the oracle's region (1-based genomic)
    startp, stopp = lo - 100, hi + 100
    sub = bytes(seq[startp - 1:stopp]).decode()
    v_lo, v_hi = int(np.searchsorted(panel.pos, startp)), int(np.searchsorted(panel.pos, stopp - 12))
    G = panel.genotypes.dense(v_lo, v_hi, 2 * sample, 2 * sample + 2)
    allcols = np.zeros(2 * N_SAMPLES, dtype=bool)  # live columns of the TILE: rows are numbered over them
    for c0 in range(0, 2 * N_SAMPLES, 626):
        allcols[c0:c0 + 626] = panel.genotypes.dense(0, len(panel.pos), c0, min(c0 + 626, 2 * N_SAMPLES)).any(axis=0)
    row_of_col = np.cumsum(allcols)
    ref_u8 = np.frombuffer(sub.encode(), dtype=np.uint8)
    from crisprhawk_hip.hapset import PosSegments
    for copy in (0, 1):
        idx = np.flatnonzero(G[:, copy])
        assert len(idx)
        sites = [(int(panel.pos[v_lo + k]), panel.ref[v_lo + k].encode(), panel.alt[v_lo + k].encode()) for k in idx]
        arr, seg = expand_haplotype(ref_u8, startp, sites)
        ident = PosSegments.identity(startp, len(ref_u8))
        hs = ora.HapSet([sub, bytes(arr).decode()], [ident.full(), seg.full()], [True, False],
                        [scan_bounds(ident, startp, stopp, 3), scan_bounds(seg, startp, stopp, 3)])
        want = ora.search(hs, PAM_S, GUIDELEN, False)
        _, _, _, cfd, _ = ora.reverse_and_cfdon(want, hs.is_ref, GUIDELEN, 3, c4["mm"], c4["ptab"], decode=False)
        w = np.flatnonzero(want.guides["hap"] == 1)
        w = w[np.lexsort((want.guides["pos"][w], want.guides["strand"][w]))]
        row = int(c4["pt"].plan.alias[int(row_of_col[2 * sample + copy])])
        a, b = np.searchsorted(tab.hap, [row, row + 1])
        sel = np.arange(a, b)
        # the tile's rows of this haplotype whose PAM lies in the oracle's BED interval
        gstart = tab.start[sel]
        inside = (gstart >= lo - 30) & (gstart < hi + 30)
        sel = sel[inside]
        sel = sel[np.lexsort((tab.pos[sel], tab.strand[sel]))]
        from collections import Counter
        wins_t = tab.windows(sel)

        def rows(start, stop, strand, wins, cf):
            return [(int(a_), int(b_), int(c_), w_, None if d_ != d_ else float(d_))
                    for a_, b_, c_, w_, d_ in zip(start.tolist(), stop.tolist(), strand.tolist(), wins, cf.tolist())]
        got = rows(tab.start[sel], tab.stop[sel], tab.strand[sel], wins_t, tab.cfdon[sel])
        exp = rows(want.guides["start"][w], want.guides["stop"][w], want.guides["strand"][w], [want.windows[j] for j in w.tolist()], cfd[w])
        assert len(exp) > 1000
        # every oracle row of the copy is in the tile's table (several rows may share a start: inserted bases repeat their
        # anchor's position, so rows are compared whole - coordinates, window, score) ...
        have = Counter(got)
        assert not (Counter(exp) - have)
        # ... and the tile holds no other row of it strictly inside the interval
        exp_set = set(exp)
        assert all(r in exp_set for r in got if lo + 30 <= r[0] < hi - 60)
