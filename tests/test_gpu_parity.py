"""Parity of the HIP path (through the C ABI) against the reference-generated golden vectors
and against the CPU oracle on seeded inputs.  Integer/byte/index work is compared bit-exact;
CFDon is fp64 with the reference's multiplication order, also compared bit-exact."""
import math

import numpy as np
import pytest

from crisprhawk_hip import _lib, synth
from crisprhawk_hip.hapset import DeviceHapSet, HostHaplotype, PosSegments, segments_from_posmap
from oracle import oracle as ora
from util import G3_CASES, hapset_from_golden, load_golden, oracle_haplotypes

pytestmark = pytest.mark.gpu


def device_set(hs: ora.HapSet) -> DeviceHapSet:
    haps = []
    for seq, pm, r, sc in zip(hs.seqs, hs.posmaps, hs.is_ref, hs.scan):
        rel, gen = segments_from_posmap(pm)
        haps.append(HostHaplotype(seq, PosSegments(rel, gen, len(seq)), r, sc))
    return DeviceHapSet(haps)


def test_pack_matches_encoder_table():
    g1 = load_golden("g1_tables.json.gz")
    letters = "".join(g1["table"].keys()) + "".join(g1["lower"].keys())
    seq = letters * 5 + "ACGT" * 700  # > one 2048-base wave chunk
    ds = DeviceHapSet([HostHaplotype(seq, PosSegments.identity(1, len(seq)), True, (0, len(seq) - 3))])
    assert np.array_equal(ds.nibbles(0), ora.encode(seq))
    vplane = ds.planes()[4, 0]
    vbits = ((vplane[:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1)[: len(seq)]
    assert np.array_equal(vbits.astype(bool), np.array([c.islower() for c in seq]))
    with pytest.raises(_lib.HawkStatusError) as e:
        DeviceHapSet([HostHaplotype("ACGTXACGT" * 10, PosSegments.identity(1, 90), True, (0, 80))])
    assert e.value.status == _lib.HAWK_E_IUPAC and "position 4" in str(e.value)


def test_g2_scan_hit_lists():
    g2 = load_golden("g2_scan.json.gz")
    for c in g2["cases"]:
        bits, bitsrc, _, _ = ora.pam_encode(c["pam"])
        seq = c["seq"]
        ds = DeviceHapSet([HostHaplotype(seq, PosSegments.identity(1, len(seq)), True, (c["start"], c["stop"]))])
        (fwd, rev), = ds.pam_scan(bits, bitsrc, len(c["pam"]))
        assert fwd.tolist() == c["fwd"] and rev.tolist() == c["rev"], (c["pam"], c["start"], c["stop"])


@pytest.mark.parametrize("case", G3_CASES)
def test_g3_search_against_reference_vectors(case):
    fx = load_golden(f"g3_search_{case}.json.gz")
    hs = hapset_from_golden(fx)
    pam, guidelen, right = fx["pam"], fx["guidelen"], fx["right"]
    bits, bitsrc, _, _ = ora.pam_encode(pam)
    ds = device_set(hs)
    hits = ds.pam_scan(bits, bitsrc, len(pam))
    for h, (f, r) in enumerate(hits):
        assert f.tolist() == fx["hits"][h][0] and r.tolist() == fx["hits"][h][1]
    mm = pt = None
    if "cfdon" in fx:
        mm, pt = synth.cfd_tables()
    tab = ds.search(bits, bitsrc, len(pam), guidelen, right, mm, pt)
    assert tab.n_hits == sum(len(f) + len(r) for f, r in fx["hits"])
    order = tab.reference_order()
    wins = tab.windows()
    got = [[int(tab.start[i]), int(tab.stop[i]), int(tab.strand[i]), wins[i], int(tab.hap[i]),
            bool(tab.right_as_stored()[i])] for i in order]
    assert got == fx["guides"]
    if "cfdon" in fx:
        ref_scores = {gi: sc for gi, sc in zip(fx["cfdon_order"], fx["cfdon"])}
        for k, i in enumerate(order):
            want = ref_scores[k]
            if want is None:
                assert math.isnan(tab.cfdon[i])
            else:
                assert tab.cfdon[i] == want, (k, tab.cfdon[i], want)


@pytest.mark.parametrize("pam,guidelen,right", [("NGG", 20, False), ("TTTV", 23, True), ("NNGRRT", 21, False)])
def test_search_against_oracle_200kb(pam, guidelen, right):
    reg = synth.make_region(7001, "chrR", 260_000, 30_000, 230_000)
    synth.add_phased_variants(reg, 7002, 4000, 6, af_min=0.05, af_max=0.5)
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, len(pam)) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    bits, bitsrc, _, _ = ora.pam_encode(pam)
    want = ora.search(hs, pam, guidelen, right)
    mm = pt = None
    if not right and pam == "NGG":
        mm, pt = synth.cfd_tables()
    ds = device_set(hs)
    tab = ds.search(bits, bitsrc, len(pam), guidelen, right, mm, pt)
    assert (tab.n_rows, tab.n_candidates, tab.n_hits) == (len(want.guides), want.n_candidates, want.n_hits)
    order = tab.reference_order()
    g = want.guides
    assert np.array_equal(tab.start[order], g["start"]) and np.array_equal(tab.stop[order], g["stop"])
    assert np.array_equal(tab.hap[order], g["hap"]) and np.array_equal(tab.pos[order], g["pos"])
    assert np.array_equal(tab.strand[order], g["strand"])
    wins = tab.windows()
    assert [wins[i] for i in order] == want.windows
    if mm is not None:
        _, _, _, cfd, _ = ora.reverse_and_cfdon(want, hs.is_ref, guidelen, len(pam), mm, pt)
        mine = tab.cfdon[order]
        assert np.array_equal(np.isnan(mine), np.isnan(cfd))
        assert np.array_equal(mine[~np.isnan(cfd)], cfd[~np.isnan(cfd)])


def test_cfd_error_on_iupac_spacer():
    # a REF/alt pair differing where the alt carries an IUPAC code must surface the
    # reference's KeyError -> CrisprHawkCfdScoreError path (cfdscore.py:93)
    ref = "ACGTACGTAC" + "ACGTTGCATGCATGCATGCA" + "TGG" + "ACGTACGTAC" + "ACGT" * 20
    alt = ref[:15] + "r" + ref[16:]
    n = len(ref)
    haps = [HostHaplotype(ref, PosSegments.identity(1, n), True, (10, n - 13)),
            HostHaplotype(alt, PosSegments.identity(1, n), False, (10, n - 13))]
    ds = DeviceHapSet(haps)
    bits, bitsrc, _, _ = ora.pam_encode("NGG")
    mm, pt = synth.cfd_tables()
    with pytest.raises(_lib.HawkStatusError) as e:
        ds.search(bits, bitsrc, 3, 20, False, mm, pt)
    assert e.value.status == _lib.HAWK_E_CFD


def _oracle_vs_device(reg, pam, guidelen, right, score):
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, len(pam)) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    bits, bitsrc, _, _ = ora.pam_encode(pam)
    want = ora.search(hs, pam, guidelen, right)
    mm, pt = synth.cfd_tables() if score else (None, None)
    tab = device_set(hs).search(bits, bitsrc, len(pam), guidelen, right, mm, pt)
    assert (tab.n_rows, tab.n_candidates, tab.n_hits) == (len(want.guides), want.n_candidates, want.n_hits)
    order = tab.reference_order()
    g = want.guides
    for col in ("start", "stop", "hap", "pos", "strand"):
        assert np.array_equal(getattr(tab, col)[order], g[col]), col
    wins = tab.windows()
    assert [wins[i] for i in order] == want.windows
    if score:
        _, _, _, cfd, _ = ora.reverse_and_cfdon(want, hs.is_ref, guidelen, len(pam), mm, pt)
        mine = tab.cfdon[order]
        assert np.array_equal(np.isnan(mine), np.isnan(cfd)) and np.array_equal(mine[~np.isnan(cfd)], cfd[~np.isnan(cfd)])
    return hs, tab


def test_more_than_64_segments_per_tile():
    # > NSEG position-map segments inside one 32 768-position tile: the kernel's global binary-search fallback
    reg = synth.make_region(7101, "chrD", 9000, 1500, 7500)
    synth.add_phased_variants(reg, 7102, 500, 3, frac_snv=0.1, frac_del=0.45, max_indel=5, af_min=0.5, af_max=0.9)
    hs, _ = _oracle_vs_device(reg, "NGG", 20, False, True)
    assert max(len(segments_from_posmap(pm)[0]) for pm in hs.posmaps) > 64


@pytest.mark.parametrize("region_len", [32768 - 201, 32768 - 200, 65536 - 200 + 1, 4 * 32768 - 200 - 33])
def test_tile_boundary_lengths(region_len):
    # haplotype lengths that end exactly on / just past plane-word and tile boundaries; indels move them around
    reg = synth.make_region(7200 + region_len % 97, "chrB", region_len + 3000, 1200, 1200 + region_len)
    synth.add_phased_variants(reg, 7201, 60, 2, frac_snv=0.5, frac_del=0.25, af_min=0.3, af_max=0.8)
    _oracle_vs_device(reg, "NGG", 20, False, True)
    _oracle_vs_device(reg, "TTTV", 23, True, False)


@pytest.mark.parametrize("case", ["phased4", "phased16", "cpf1", "indel_dense", "tiny"])
def test_device_haplotype_expansion_vs_reference_vectors(case):
    """planes written by hawk_hapset_expand == planes packed from the haplotype
    strings the reference built; labels, position maps and scan bounds too; then the search."""
    from crisprhawk_hip.workload import expand_on_device
    from util import synth_region_from_fixture, posmap_from_breaks
    fx = load_golden(f"g3_search_{case}.json.gz")
    reg = synth_region_from_fixture(fx)
    ds, info, _ms, kept = expand_on_device(reg, len(fx["pam"]))
    assert len(kept) == len(fx["haplotypes"])
    want = device_set(hapset_from_golden(fx))
    pw, pg = want.planes(), ds.planes()
    for j, r in enumerate(kept):
        n = (len(fx["haplotypes"][j]["seq"]) + 31) // 32
        assert np.array_equal(pg[:, r, :n], pw[:, j, :n]), (case, j)
        assert not pg[:, r, n:].any()
        assert sorted(info[j].samples) == fx["haplotypes"][j]["samples"]
        seg = ds.host_meta[r].seg
        assert np.array_equal(seg.full(), posmap_from_breaks(fx["haplotypes"][j]["posmap_breaks"], fx["haplotypes"][j]["posmap_len"]))
        assert list(ds.host_meta[r].scan) == fx["scan"][j]
    bits, bitsrc, _, _ = ora.pam_encode(fx["pam"])
    tab = ds.search(bits, bitsrc, len(fx["pam"]), fx["guidelen"], fx["right"])
    rowmap = {r: j for j, r in enumerate(kept)}
    order = tab.reference_order()
    wins = tab.windows()
    got = [[int(tab.start[i]), int(tab.stop[i]), int(tab.strand[i]), wins[i], rowmap[int(tab.hap[i])], bool(tab.right_as_stored()[i])]
           for i in order]
    assert got == fx["guides"]


def test_device_expansion_matches_host_expansion_200kb():
    from crisprhawk_hip.workload import build_phased_haplotypes, expand_on_device
    reg = synth.make_region(7301, "chrH", 260_000, 30_000, 230_000)
    synth.add_phased_variants(reg, 7302, 6000, 12, af_min=0.02, af_max=0.5)
    haps, info_h = build_phased_haplotypes(reg, 3)
    ds, info_d, ms, kept = expand_on_device(reg, 3)
    assert len(kept) == len(haps) and [sorted(i.samples) for i in info_d] == [sorted(i.samples) for i in info_h]
    want = DeviceHapSet(haps).planes()
    got = ds.planes()
    for j, r in enumerate(kept):
        n = (len(haps[j].seq) + 31) // 32
        assert np.array_equal(got[:, r, :n], want[:, j, :n])
        assert np.array_equal(ds.host_meta[r].seg.rel, haps[j].seg.rel) and np.array_equal(ds.host_meta[r].seg.gen, haps[j].seg.gen)
        assert ds.host_meta[r].scan == haps[j].scan


def test_device_expansion_crowded_tiles_and_long_deletions():
    # what the expansion kernel stages per tile of 32768 output positions has a capacity: 256 carried variants and the
    # tile's image in REF plus 2048 net deleted bases.  Here one tile carries a SNV every 25 nt (the variants beyond the
    # staged ones come from global memory), another loses 4 x 1500 nt to deletions (its REF image does not fit: copies
    # read global memory), and a 400-nt insertion crosses a word-quad boundary.
    from crisprhawk_hip.workload import build_phased_haplotypes, expand_on_device
    reg = synth.make_region(7351, "chrK", 170_000, 5_000, 165_000)
    rng = np.random.default_rng(7352)
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
        if refb not in "ACGT":
            continue
        sites.append(synth.VariantSite(pos, refb, "ACGT"[("ACGT".index(refb) + 1) % 4], 0.7, gt(0.7)))
    for pos in (70_000, 76_000, 82_000, 88_000):
        sites.append(synth.VariantSite(pos, seq[pos - 1:pos + 1500], seq[pos - 1], 0.5, gt(0.5)))
    pos = 120_000
    sites.append(synth.VariantSite(pos, seq[pos - 1], seq[pos - 1] + "".join("ACGT"[b] for b in rng.integers(0, 4, 400)), 0.5, gt(0.5)))
    for pos in range(130_000, 160_000, 700):
        refb = seq[pos - 1]
        if refb not in "ACGT":
            continue
        sites.append(synth.VariantSite(pos, refb, "ACGT"[("ACGT".index(refb) + 2) % 4], 0.3, gt(0.3)))
    reg.variants = sites
    haps, info_h = build_phased_haplotypes(reg, 3)
    ds, info_d, ms, kept = expand_on_device(reg, 3)
    assert len(kept) == len(haps) and [sorted(i.samples) for i in info_d] == [sorted(i.samples) for i in info_h]
    want = DeviceHapSet(haps).planes()
    got = ds.planes()
    for j, r in enumerate(kept):
        n = (len(haps[j].seq) + 31) // 32
        assert np.array_equal(got[:, r, :n], want[:, j, :n])
        assert not got[:, r, n:].any()
        assert np.array_equal(ds.host_meta[r].seg.rel, haps[j].seg.rel) and np.array_equal(ds.host_meta[r].seg.gen, haps[j].seg.gen)


def test_clamp_error_is_for_indels_only():
    # haplotype.py:199-201 compares a variant's span with the region's ORIGINAL length after upstream length changes; the
    # reference applies a copy's SNVs first, on the untouched sequence (494-512), so a SNV pushed past that length by an
    # upstream insertion is fine and only an indel there raises (found by tools/stress_parity.py: the device path refused
    # the SNV case too, the host builder did not)
    from crisprhawk_hip.expand import HaplotypeBuildError
    from crisprhawk_hip.workload import build_phased_haplotypes, expand_on_device
    reg = synth.make_region(7361, "chrQ", 30_000, 2_000, 28_000)
    seq, n = reg.contig_seq, len(reg.sequence)
    reg.samples = ["S0000", "S0001"]
    gt = np.array([[1, 0], [1, 1]], dtype=np.uint8)
    p_ins, p_snv = reg.startp + 500, reg.startp + n - 40
    ins = synth.VariantSite(p_ins, seq[p_ins - 1], seq[p_ins - 1] + "ACGT" * 20, 0.5, gt)
    snv = synth.VariantSite(p_snv, seq[p_snv - 1], "ACGT"[("ACGT".index(seq[p_snv - 1]) + 1) % 4], 0.5, gt)
    reg.variants = [ins, snv]
    haps, _ = build_phased_haplotypes(reg, 3)
    ds, _info, _ms, kept = expand_on_device(reg, 3)
    want, got = DeviceHapSet(haps).planes(), ds.planes()
    for j, r in enumerate(kept):
        w = (len(haps[j].seq) + 31) // 32
        assert np.array_equal(got[:, r, :w], want[:, j, :w])
    p_del = reg.startp + n - 40
    reg.variants = [ins, synth.VariantSite(p_del, seq[p_del - 1:p_del + 2], seq[p_del - 1], 0.5, gt)]
    with pytest.raises(HaplotypeBuildError):
        build_phased_haplotypes(reg, 3)
    with pytest.raises(HaplotypeBuildError):
        expand_on_device(reg, 3)


def _dense_region(seed, sites, samples):
    reg = synth.make_region(seed, "chrX", 75_000, 2_000, 72_000)
    synth.add_phased_variants(reg, seed + 1, sites, samples, frac_snv=0.8, frac_del=0.1, max_indel=3, af_min=0.6, af_max=0.95)
    return reg


def test_tiles_with_more_rows_than_the_handover_list():
    # a variant every ~12 nt carried by most haplotypes: alt tiles keep thousands of rows, far more than the
    # 512-entry hand-over list of the count pass, so they take the recompute emit pass next to list-driven tiles
    reg = _dense_region(7301, 6000, 3)
    hs, tab = _oracle_vs_device(reg, "NGG", 20, False, True)
    per_tile = np.bincount((tab.hap.astype(np.int64) * 8 + tab.pos // 32768)[~np.asarray(hs.is_ref)[tab.hap]])
    assert per_tile.max() > 512 and (per_tile[per_tile > 0] <= 512).any()
    _oracle_vs_device(reg, "TTTV", 23, True, False)


def test_search_without_a_ref_haplotype():
    # no REF row in the set: nothing is redundant, no CFDon partner exists (every score is NA)
    reg = synth.make_region(7401, "chrN", 50_000, 1_000, 48_000)
    synth.add_phased_variants(reg, 7402, 400, 4, af_min=0.2, af_max=0.7)
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = [h for h in oracle_haplotypes(fx) if h["samples"] != ["REF"]]
    assert haps
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, 3) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [False] * len(haps), scan)
    bits, bitsrc, _, _ = ora.pam_encode("NGG")
    want = ora.search(hs, "NGG", 20, False)
    mm, pt = synth.cfd_tables()
    tab = device_set(hs).search(bits, bitsrc, 3, 20, False, mm, pt)
    assert (tab.n_rows, tab.n_candidates, tab.n_hits) == (len(want.guides), want.n_candidates, want.n_hits)
    order = tab.reference_order()
    for col in ("start", "stop", "hap", "pos", "strand"):
        assert np.array_equal(getattr(tab, col)[order], want.guides[col]), col
    wins = tab.windows()
    assert [wins[i] for i in order] == want.windows
    assert np.isnan(tab.cfdon).all() and not tab.flags.any()


def test_recompute_emit_pass_matches_list_driven_emit():
    # HAWK_LIST_EMIT=0 switches the hand-over lists off (read once per process, hence the child): both emit
    # paths must produce the same table, compared here through a digest of every column
    import hashlib
    import os
    import subprocess
    import sys
    code = r'''
import hashlib, sys
import numpy as np
sys.path[:0] = [%r, %r, %r]
from crisprhawk_hip import synth
from oracle import oracle as ora
import test_gpu_parity as T
reg = T._dense_region(7301, 1500, 3)
hs, tab = T._oracle_vs_device(reg, "NGG", 20, False, True)
h = hashlib.sha256()
for col in ("hap", "pos", "strand", "start", "stop", "flags"):
    h.update(np.ascontiguousarray(getattr(tab, col)).tobytes())
h.update(np.nan_to_num(tab.cfdon, nan=-1.0).tobytes())
h.update(np.ascontiguousarray(tab.win).tobytes())
print("DIGEST", tab.n_rows, h.hexdigest())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (os.path.join(root, "tests"), os.path.join(root, "crispr-hawk_amd"), root)
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, HAWK_LIST_EMIT=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][0])
    assert outs[0] == outs[1]


def _check_collapse(hs, tab, guidelen, pamlen, right):
    """GuideTable.collapse() (device sort + group) against the oracle's dictionary grouping."""
    wins = tab.windows()
    isref_row = np.asarray(hs.is_ref)[tab.hap]
    want, want_gc = ora.collapse_rows(tab.start, tab.stop, tab.strand, isref_row, wins, guidelen, pamlen, right)
    assert tab.n_groups == len(want)
    perm, off = tab.group_perm.astype(np.int64), tab.group_off.astype(np.int64)
    assert sorted(perm.tolist()) == list(range(tab.n_rows)) and off[0] == 0 and off[-1] == tab.n_rows
    assert (np.diff(tab.start[perm]) >= 0).all()  # ordered by start: the report's primary sort key
    seen = set()
    for g in range(tab.n_groups):
        rows = perm[off[g]:off[g + 1]]
        assert len(rows) and (np.diff(rows) > 0).all()  # members keep table order
        r = int(rows[0])
        key = (int(tab.start[r]), int(tab.stop[r]), int(tab.strand[r]), bool(isref_row[r]), wins[r][10:-10])
        assert key not in seen and want[key] == rows.tolist()
        seen.add(key)
        assert (int(tab.gc_num[g]), int(tab.gc_den[g])) == want_gc[key]


@pytest.mark.parametrize("exact,mode", [("0", "sort"), ("0", "hash"), ("1", "sort")], ids=["hash-identity", "hash-table", "full-key"])
@pytest.mark.parametrize("pam,guidelen,right", [("NGG", 20, False), ("TTTV", 23, True)])
def test_collapse_groups_against_oracle(pam, guidelen, right, exact, mode, monkeypatch):
    # 12 haplotypes over common variants: most alt rows are shared by several haplotypes.  Rows of one (start, strand)
    # are told apart by 63 hash bits by default, by their full keys with HAWK_COLLAPSE_EXACT=1; grouping runs through a
    # sort of all rows or (large tables by default, here forced) through a hash table (both read per call)
    monkeypatch.setenv("HAWK_COLLAPSE_EXACT", exact)
    monkeypatch.setenv("HAWK_COLLAPSE_MODE", mode)
    reg = synth.make_region(7501, "chrC", 40_000, 1_000, 38_000)
    synth.add_phased_variants(reg, 7502, 300, 6, af_min=0.3, af_max=0.8)
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, len(pam)) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    bits, bitsrc, _, _ = ora.pam_encode(pam)
    tab = device_set(hs).search(bits, bitsrc, len(pam), guidelen, right, collapse=True)
    assert tab.n_groups < tab.n_rows
    _check_collapse(hs, tab, guidelen, len(pam), right)


def test_collapse_verify_pass_catches_hash_collisions():
    """The grouping's verification pass against deliberately colliding hashes - in the HOOKS build of the library
    (libhawk_hip_hooks.so, -DHAWK_TEST_HOOKS): the product library can neither weaken its hash nor skip the verification, so the
    check runs in a process of its own that loads the other library (tests/hooks_collapse_check.py).  The product library is
    asked too: the same environment must leave its grouping untouched."""
    import os
    import subprocess
    import sys
    from crisprhawk_hip import _lib
    here = os.path.dirname(os.path.abspath(__file__))
    hooks = os.path.join(os.path.dirname(_lib.LIB_PATH), "libhawk_hip_hooks.so")
    assert os.path.exists(hooks), "make -C crispr-hawk_amd/csrc builds libhawk_hip_hooks.so beside the product library"
    env = dict(os.environ, CRISPRHAWK_HIP_LIB=hooks)
    r = subprocess.run([sys.executable, os.path.join(here, "hooks_collapse_check.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "hooks ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    # the product library ignores both switches
    reg = synth.make_region(7501, "chrC", 40_000, 1_000, 38_000)
    synth.add_phased_variants(reg, 7502, 300, 6, af_min=0.3, af_max=0.8)
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, 3) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    bits, bitsrc, _, _ = ora.pam_encode("NGG")
    ds = device_set(hs)
    good = ds.search(bits, bitsrc, 3, 20, False, collapse=True)
    os.environ["HAWK_COLLAPSE_WEAK_HASH"], os.environ["HAWK_COLLAPSE_VERIFY"] = "1", "0"
    try:
        same = ds.search(bits, bitsrc, 3, 20, False, collapse=True)
    finally:
        del os.environ["HAWK_COLLAPSE_WEAK_HASH"], os.environ["HAWK_COLLAPSE_VERIFY"]
    assert same.n_groups == good.n_groups and np.array_equal(same.group_perm, good.group_perm)


def test_haplotype_collapse_compares_rows_not_only_hashes():
    """Rows are aliased onto another row of equal CONTENT: the 128-bit hashes only propose the pairs, the device compares
    them base for base (hawk_hapset_rows_equal).  Forged hashes - every row the same - must give the grouping of the
    true contents, and rows_equal itself must tell a one-base difference."""
    import ctypes as C
    from crisprhawk_hip import _lib
    from crisprhawk_hip.hapset import _p
    from crisprhawk_hip.workload import _exact_keys, expand_on_device
    reg = synth.make_region(7521, "chrD", 30_000, 1_000, 28_000)
    synth.add_phased_variants(reg, 7522, 5, 30, af_min=0.3, af_max=0.7)  # few sites, many samples: many equal copies
    ds, info, _ms, kept = expand_on_device(reg, 3)
    assert len(kept) < ds.n_hap  # some rows did collapse
    planes = ds.planes()
    content = {}
    want = np.array([content.setdefault(planes[:, r, :].tobytes(), len(content)) for r in range(ds.n_hap)])
    forged = np.zeros((ds.n_hap, 2), dtype=np.uint64)
    got = _exact_keys(ds, forged)
    # same partition of the rows
    assert len(set(zip(want.tolist(), got.tolist()))) == len(set(want.tolist())) == len(set(got.tolist()))
    # and the production keys (real hashes) give it too
    assert np.array_equal(ds.alias == np.arange(ds.n_hap), np.array([want[r] not in want[:r] for r in range(ds.n_hap)]))
    a = np.array([1, 1, 2], dtype=np.uint32)
    twin = int(np.flatnonzero(want == want[1])[-1])
    other = int(np.flatnonzero(want != want[1])[0])
    b = np.array([twin, other, 2], dtype=np.uint32)
    eq = np.zeros(3, dtype=np.uint8)
    _lib.check(_lib.lib().hawk_hapset_rows_equal(ds._h, 3, _p(a), _p(b), _p(eq)), "hawk_hapset_rows_equal")
    assert eq.tolist() == [1, 0, 1]


def test_collapse_with_scorer_flanks_splits_groups_and_both_paths_agree(monkeypatch):
    # with the model scorers' 4 + 3 flanking bases in the key (hawk_table_collapse_ex; reports.py:978-1003) rows whose
    # spacer+PAM agree but whose flanks differ stay apart: never fewer groups than without, and the same arrays from the
    # sort and the hash-table path
    reg = synth.make_region(7511, "chrC", 40_000, 1_000, 38_000)
    synth.add_phased_variants(reg, 7512, 600, 6, af_min=0.3, af_max=0.8)
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, 3) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    bits, bitsrc, _, _ = ora.pam_encode("NGG")
    tab = device_set(hs).search(bits, bitsrc, 3, 20, False, download=False)
    monkeypatch.setenv("HAWK_COLLAPSE_MODE", "sort")
    tab.collapse()
    plain = tab.n_groups
    out = {}
    for mode in ("sort", "hash"):
        monkeypatch.setenv("HAWK_COLLAPSE_MODE", mode)
        tab.collapse(flank=(4, 3))
        out[mode] = (tab.n_groups, tab.group_perm.copy(), tab.group_off.copy(), tab.gc_num.copy(), tab.gc_den.copy())
    assert out["sort"][0] == out["hash"][0] > plain
    for a, b in zip(out["sort"][1:], out["hash"][1:]):
        assert np.array_equal(a, b)
    # every group of the flank-aware collapse agrees on the widened window
    tab.download()
    wins = tab.windows()
    perm, off = out["hash"][1].astype(np.int64), out["hash"][2].astype(np.int64)
    for g in range(0, out["hash"][0], 7):
        rows = perm[off[g]:off[g + 1]]
        keys = {(wins[r][10 - 4:-10 + 3] if tab.strand[r] == 0 else wins[r][10 - 3:-10 + 4]) for r in rows.tolist()}
        assert len(keys) == 1


def test_collapse_hash_table_too_small_falls_back_to_the_sort(monkeypatch):
    # an all-N stretch: every position is a PAM hit on both strands and every row its own group, 16 x what the hash-table
    # path sizes its first table for - rows find no slot, the call must come back through the sort path with the same groups
    seq = "ACGT" * 50 + "N" * 12_000 + "ACGT" * 50
    ds = DeviceHapSet([HostHaplotype(seq, PosSegments.identity(1, len(seq)), True, (100, len(seq) - 100))])
    bits, bitsrc, _, _ = ora.pam_encode("NGG")
    monkeypatch.setenv("HAWK_COLLAPSE_MODE", "hash")
    tab = ds.search(bits, bitsrc, 3, 20, False, collapse=True)
    assert tab.n_rows > 20_000 and tab.n_groups == tab.n_rows
    hs = ora.HapSet([seq], [np.arange(1, len(seq) + 1, dtype=np.int64)], [True], [(100, len(seq) - 100)])
    _check_collapse(hs, tab, 20, 3, False)


def test_collapse_empty_and_single():
    seq = "ACGT" * 100
    ds = DeviceHapSet([HostHaplotype(seq, PosSegments.identity(1, len(seq)), True, (100, 300))])
    bits, bitsrc, _, _ = ora.pam_encode("NGG")
    tab = ds.search(bits, bitsrc, 3, 20, False, collapse=True)
    assert tab.n_rows == 0 and tab.n_groups == 0 and tab.group_off.tolist() == [0]


_FUZZ = [  # (seed, pam, guidelen, right, sites, samples, frac_snv, frac_del, max_indel)
    (1, "NGG", 17, False, 120, 3, 0.7, 0.15, 4), (2, "NGG", 24, False, 200, 2, 0.3, 0.35, 8), (3, "NAG", 20, True, 80, 4, 0.9, 0.05, 2),
    (4, "TTTV", 20, True, 150, 3, 0.5, 0.25, 6), (5, "TTN", 25, True, 60, 2, 0.6, 0.2, 3), (6, "NNGRRT", 22, False, 100, 3, 0.5, 0.25, 5),
    (7, "NNNRRT", 21, True, 90, 2, 0.8, 0.1, 3), (8, "TTCN", 20, True, 110, 3, 0.6, 0.2, 4), (9, "NGK", 20, False, 140, 5, 0.4, 0.3, 7),
    (10, "NNG", 18, False, 70, 2, 0.7, 0.15, 2), (11, "YTTV", 23, True, 130, 4, 0.5, 0.25, 5), (12, "NGNNNNNNNNNNNNGN", 28, False, 60, 2, 0.6, 0.2, 3),
    (13, "N", 20, False, 40, 2, 0.7, 0.15, 2), (14, "NRG", 41, False, 50, 2, 0.6, 0.2, 3),
]


@pytest.mark.parametrize("cfg", _FUZZ, ids=[f"{c[1]}-{c[2]}-{'R' if c[3] else 'L'}" for c in _FUZZ])
def test_search_parameter_sweep_against_oracle(cfg):
    # PAM shapes (single base, IUPAC sets, all-N, 16 long), guide lengths up to the 44-nt core limit, both guide sides,
    # SNV- to indel-heavy variant mixes; a region that spans two tiles with an IUPAC-bearing reference
    seed, pam, guidelen, right, sites, samples, fs, fd, mi = cfg
    reg = synth.make_region(7600 + seed, "chrZ", 41_000, 1_500, 39_500, iupac_frac=0.002 if seed % 3 == 0 else 0.0)
    synth.add_phased_variants(reg, 7700 + seed, sites, samples, frac_snv=fs, frac_del=fd, max_indel=mi, af_min=0.2, af_max=0.8)
    score = (not right) and reg.sequence.upper().count("N") == 0 and seed % 3 != 0 and len(pam) >= 2
    _oracle_vs_device(reg, pam, guidelen, right, score)


def test_repeated_searches_on_one_set_grow_and_shrink_the_table():
    # the second and later searches launch their emit pass before the row count is known (columns already reserved):
    # a bigger table than ever before must fall back to reserve-and-relaunch, a smaller one must not see stale rows
    reg = synth.make_region(7801, "chrG", 45_000, 1_200, 43_000)
    synth.add_phased_variants(reg, 7802, 250, 4, af_min=0.2, af_max=0.7)
    fx = dict(region_seq=reg.sequence, startp=reg.startp, samples=reg.samples,
              variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants])
    haps = oracle_haplotypes(fx)
    sizes = []
    ds = None
    for pam in ("NGG", "NNG", "TTTV", "N", "NGG"):
        scan = [ora.scan_bounds(h["posmap"], reg.startp, reg.stopp, len(pam)) for h in haps]
        hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
        if ds is None:
            ds = device_set(hs)
        else:  # same planes, new scan bounds for this PAM length
            ds.set_meta([HostHaplotype(seq, PosSegments(*segments_from_posmap(pm), len(seq)), r, sc)
                         for seq, pm, r, sc in zip(hs.seqs, hs.posmaps, hs.is_ref, hs.scan)])
        bits, bitsrc, _, _ = ora.pam_encode(pam)
        want = ora.search(hs, pam, 20, False)
        tab = ds.search(bits, bitsrc, len(pam), 20, False)
        assert (tab.n_rows, tab.n_candidates) == (len(want.guides["start"]), want.n_candidates)
        order = tab.reference_order()
        for col in ("start", "stop", "hap", "pos", "strand"):
            assert np.array_equal(getattr(tab, col)[order], want.guides[col]), (pam, col)
        wins = tab.windows()
        assert [wins[i] for i in order] == want.windows
        sizes.append(tab.n_rows)
    assert sizes[1] > sizes[0] and sizes[3] > sizes[1] and sizes[4] == sizes[0]
