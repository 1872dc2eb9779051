"""Parity at BASELINE.json's full size (C3: 1 Mb x 2504 phased samples, 5009 haplotype rows, 5.0e9 scanned positions,
2.8e7 guide rows) through size-independent properties, plus exact comparison with the oracle on haplotypes sampled
out of the full table.  The table under test is the one bench.py's timed step produces: the search of the expansion plan's
VIEW, once per distinct variant cluster (hawk_csearch.hip, hawk_timing.v_path == 2, packed 64-byte rows); the search over
materialised planes must give the same rows.  One module-scoped workload; ~30 s on an MI355X."""
import ctypes as C

import numpy as np
import pytest

from crisprhawk_hip import _lib, synth
from crisprhawk_hip.hapset import _p
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.workload import build_phased_haplotypes, expand_on_device
from oracle import oracle as ora

pytestmark = pytest.mark.gpu

PAM_S, GUIDELEN = "NGG", 20


@pytest.fixture(scope="module")
def c3():
    reg = synth.config_c3()
    ds, info, _, kept = expand_on_device(reg, len(PAM_S), keep_plan=True)
    pam = PAM(PAM_S, False, True)
    pam.encode(0)
    mm, pt = synth.cfd_tables()
    view = ds.plan.view()
    st = ds.plan.cluster_stats()
    assert st["usable"] and st["instances"] >= 3 * st["distinct"]   # the dictionary is used by default on this panel
    tab = view.search(pam.bits, pam.bitsrc, 3, GUIDELEN, False, mm, pt, download=False, collapse=True)
    assert tab.timing["v_path"] == 2                                 # the headline step's kernels, not a fall-back
    assert tab.layout() == "rows"
    tab.download()
    return dict(reg=reg, ds=ds, view=view, info=info, kept=kept, pam=pam, tab=tab, mm=mm, pt=pt)


def test_plane_search_gives_the_same_rows_in_canonical_order(c3):
    # hawk_xplan_run's planes through k_search_count / k_emit_list: the same rows, a haplotype's by tile instead of by cluster
    ds, pam, tab = c3["ds"], c3["pam"], c3["tab"]
    t2 = ds.search(pam.bits, pam.bitsrc, 3, GUIDELEN, False, c3["mm"], c3["pt"])
    assert t2.timing["v_path"] == 0 and t2.layout() == "columns"
    assert (t2.n_rows, t2.n_candidates, t2.n_hits) == (tab.n_rows, tab.n_candidates, tab.n_hits)
    oa, ob = np.lexsort((tab.pos, tab.strand, tab.hap)), np.lexsort((t2.pos, t2.strand, t2.hap))
    for col in ("hap", "pos", "strand", "start", "stop", "flags"):
        assert np.array_equal(getattr(tab, col)[oa], getattr(t2, col)[ob]), col
    assert np.array_equal(tab.cfdon[oa], t2.cfdon[ob], equal_nan=True)
    for p in range(5):
        assert np.array_equal(tab.win[p][oa], t2.win[p][ob]), p
    t2.close()


def test_counts_agree_with_the_independent_scan_kernel(c3):
    # n_hits of the fused search = total PAM hits of hawk_pam_scan (k_scan_raw + its own offset scan), per strand sums
    ds, pam, tab = c3["ds"], c3["pam"], c3["tab"]
    off_f = np.zeros(ds.n_hap + 1, dtype=np.uint64)
    off_r = np.zeros(ds.n_hap + 1, dtype=np.uint64)
    rc = ds._L.hawk_pam_scan(ds._h, C.c_uint64(pam.bits), C.c_uint64(pam.bitsrc), 3, None, None, C.c_uint64(0), C.c_uint64(0),
                             _p(off_f), _p(off_r))
    assert rc in (_lib.HAWK_OK, _lib.HAWK_E_CAPACITY)
    assert int(off_f[-1]) + int(off_r[-1]) == tab.n_hits
    assert tab.n_candidates <= tab.n_hits and tab.n_rows == len(tab.hap) == 28_099_525


def test_rows_are_ordered_and_unique(c3):
    tab = c3["tab"]
    assert (np.diff(tab.hap.astype(np.int64)) >= 0).all()                       # haplotype-major
    key = (tab.hap.astype(np.uint64) << np.uint64(33)) | (tab.strand.astype(np.uint64) << np.uint64(32)) | tab.pos.astype(np.uint64)
    assert len(np.unique(key)) == tab.n_rows                                     # one row per (haplotype, strand, position)
    # start follows pos inside every (haplotype, strand): the position map is monotone
    order = np.lexsort((tab.pos, tab.strand, tab.hap))
    same = (tab.hap[order][1:] == tab.hap[order][:-1]) & (tab.strand[order][1:] == tab.strand[order][:-1])
    assert (np.diff(tab.start[order])[same] >= 0).all()
    d = tab.stop - tab.start
    assert d.min() >= 1 and np.median(d) == GUIDELEN + 3


def test_ref_partner_flags_and_cfdon(c3):
    tab, ds, pt = c3["tab"], c3["ds"], c3["pt"]
    isref = tab.hap == 0
    k = tab.start.astype(np.int64) * 2 + tab.strand
    ref_keys = np.unique(k[isref])
    assert isref.sum() == len(ref_keys)                                          # REF: one guide per (start, strand)
    has = np.isin(k, ref_keys)
    assert np.array_equal(tab.flags & 1, has.astype(np.uint8))                   # flag bit 0 <=> a REF guide shares the key
    assert np.array_equal(np.isnan(tab.cfdon), ~has)                             # no REF partner -> "NA"
    # a REF guide scored against itself has no mismatch: the score is its PAM's table entry
    assert np.isin(tab.cfdon[isref], np.asarray(pt).reshape(-1)).all()
    assert (tab.cfdon[has] >= 0).all() and (tab.cfdon[has] <= 1).all()
    # alt rows never repeat REF's spacer+PAM at the same key (remove_redundant_guides): compare the core slices
    core = (tab.win >> np.uint64(10)) & np.uint64((1 << (GUIDELEN + 3)) - 1)
    ref_sig = {}
    r = np.flatnonzero(isref)
    for kk, a, c, g, t in zip(k[r].tolist(), *(core[p][r].tolist() for p in range(4))):
        ref_sig[kk] = (a, c, g, t)
    alt = np.flatnonzero(has & ~isref)[:: max(1, int((has & ~isref).sum()) // 200_000)]  # a 200k-row sample
    for i in alt.tolist():
        assert ref_sig[int(k[i])] != tuple(int(core[p][i]) for p in range(4))


def test_search_is_deterministic_and_additive(c3):
    # same set, second search: identical table (bit for bit); a set without the second half of the rows' scan ranges
    # is exercised by the sampled-haplotype test below, additivity here is on the group structure
    view, pam, tab = c3["view"], c3["pam"], c3["tab"]
    t2 = view.search(pam.bits, pam.bitsrc, 3, GUIDELEN, False, c3["mm"], c3["pt"])
    assert t2.timing["v_path"] == 2
    for col in ("hap", "pos", "strand", "start", "stop", "flags", "win"):
        assert np.array_equal(getattr(tab, col), getattr(t2, col)), col
    assert np.array_equal(np.nan_to_num(tab.cfdon, nan=-1.0), np.nan_to_num(t2.cfdon, nan=-1.0))
    # collapse: a partition of the rows, ordered by start, members in table order
    perm, off = tab.group_perm.astype(np.int64), tab.group_off.astype(np.int64)
    assert off[0] == 0 and off[-1] == tab.n_rows and (np.diff(off) > 0).all()
    assert np.array_equal(np.sort(perm), np.arange(tab.n_rows))
    assert (np.diff(tab.start[perm]) >= 0).all()
    first = perm[off[:-1]]
    gid = np.repeat(np.arange(len(off) - 1), np.diff(off))
    for col in (tab.start, tab.stop, tab.strand):                                # every member agrees with its group's head
        assert np.array_equal(col[perm], col[first][gid])
    inner = np.ones(tab.n_rows, dtype=bool)
    inner[off[:-1]] = False
    assert (np.diff(perm)[inner[1:]] > 0).all()


def test_collapse_hash_table_path_equals_sort_path(c3, monkeypatch):
    # 2.8 x 10^7 rows in 2.2 x 10^5 groups: what hawk_table_collapse does through its hash table (groups numbered in key
    # order, then a sort of (group, row)) must be the sort path's output array for array
    ds, pam = c3["ds"], c3["pam"]
    t = ds.search(pam.bits, pam.bitsrc, 3, GUIDELEN, False, c3["mm"], c3["pt"], download=False)
    out = {}
    for mode in ("hash", "sort"):
        monkeypatch.setenv("HAWK_COLLAPSE_MODE", mode)
        t.collapse()
        out[mode] = (t.n_groups, t.group_perm.copy(), t.group_off.copy(), t.gc_num.copy(), t.gc_den.copy(), t.collapse_ms)
    assert out["hash"][0] == out["sort"][0] == c3["tab"].n_groups
    for a, b in zip(out["hash"][1:5], out["sort"][1:5]):
        assert np.array_equal(a, b)
    t.close()


@pytest.mark.parametrize("sample", [0, 1251, 2503])
def test_sampled_haplotypes_match_the_oracle_exactly(c3, sample):
    """Rows of one sample's chromosome copies, cut out of the full table, against the oracle run on REF + that sample
    built on the host from the same variant calls."""
    reg, ds, tab = c3["reg"], c3["ds"], c3["tab"]
    haps, hinfo = build_phased_haplotypes(reg, 3, sample_slice=slice(sample, sample + 1))
    hs = ora.HapSet([bytes(h.seq).decode("ascii") for h in haps], [h.seg.full() for h in haps], [h.is_ref for h in haps],
                    [h.scan for h in haps])
    want = ora.search(hs, PAM_S, GUIDELEN, False)
    _, _, _, cfd, _ = ora.reverse_and_cfdon(want, hs.is_ref, GUIDELEN, 3, c3["mm"], c3["pt"], decode=False)
    want_windows = want.windows
    # device rows of this sample's copies: row = 1 + rank of the column among the columns that carry a variant
    G = np.stack([v.gt[sample] for v in reg.variants])                           # [site, 2]
    allgt = np.stack([v.gt.reshape(-1) for v in reg.variants]).any(axis=0)       # which columns are live
    row_of_col = np.cumsum(allgt)                                                # 1-based row of a live column
    for local, inf in enumerate(hinfo):
        if local == 0:
            continue                                                             # REF is covered by every sample
        label = inf.samples[0]                                                   # "S....:1|0" / "0|1" / "1|1"
        copy = 0 if label.endswith("1|0") or label.endswith("1|1") else 1
        col = 2 * sample + copy
        assert G[:, copy].any()
        row = int(ds.alias[int(row_of_col[col])])                                # the row this copy's content lives in
        sel = np.flatnonzero(tab.hap == row)
        wsel = np.flatnonzero(want.guides["hap"] == local)
        o = sel[np.lexsort((tab.pos[sel], tab.strand[sel]))]
        w = wsel[np.lexsort((want.guides["pos"][wsel], want.guides["strand"][wsel]))]
        assert len(o) == len(w) > 0
        for colname in ("start", "stop", "pos", "strand"):
            assert np.array_equal(getattr(tab, colname)[o], want.guides[colname][w]), (label, colname)
        wins = tab.windows(o)
        assert wins == [want_windows[i] for i in w]
        assert np.array_equal(np.nan_to_num(tab.cfdon[o], nan=-1.0), np.nan_to_num(cfd[w], nan=-1.0))
