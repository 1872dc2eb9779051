"""K7 off-target scan vs the oracle's brute force (parity UNPINNED to the reference: it delegates
this to the external CRISPRitz binary; see oracle/hawk_oracle.c).  Bit-exact row sets."""
import numpy as np
import pytest

from crisprhawk_hip import synth
from crisprhawk_hip.genome import GenomeIndex
from crisprhawk_hip.pam import PAM
from oracle import oracle as ora

pytestmark = pytest.mark.gpu


def _plant(rng, genome: list, guide: str, pam_seq: str, right: bool, n: int, max_mm: int):
    """write mutated copies of guide+PAM (both strands) into the genome so there is something to find"""
    L = len(guide) + len(pam_seq)
    for _ in range(n):
        g = list(guide)
        for p in rng.integers(0, len(g), size=int(rng.integers(0, max_mm + 2))):
            g[p] = "ACGT"[rng.integers(0, 4)]
        w = (pam_seq + "".join(g)) if right else ("".join(g) + pam_seq)
        if rng.random() < 0.5:
            w = ora.revcomp(w)
        pos = int(rng.integers(0, len(genome) - L))
        genome[pos:pos + L] = list(w)


@pytest.mark.parametrize("n_guides", [6, 200, 2300, -2300])  # 6: all pairs; 200 / 2300: pair seeds (max_mm + 2 blocks, candidates dealt
#   evenly over a wave); -2300: the single-block seeds of rounds 1-3 (HAWK_OT_PAIRS=0: guides in three LDS chunks)
@pytest.mark.parametrize("pam_s,guidelen,right,max_mm,piece", [("NGG", 20, False, 4, 1 << 22), ("TTTV", 23, True, 3, 4096),
                                                              ("TTTV", 23, True, 4, 1 << 22),  # C5 as BASELINE.json states it
                                                              ("NNGRRT", 21, False, 2, 10000), ("NGG", 20, False, 0, 1 << 22),
                                                              ("NGG", 17, False, 6, 1 << 22)])
def test_offtarget_scan_matches_bruteforce(pam_s, guidelen, right, max_mm, piece, n_guides, monkeypatch):
    if n_guides < 0:
        monkeypatch.setenv("HAWK_OT_PAIRS", "0")
        n_guides = -n_guides
    rng = np.random.default_rng(77)
    contigs = {}
    guides = [synth.random_sequence(rng, guidelen) for _ in range(n_guides)]
    # families of near-identical guides: pairs that agree in several seed blocks must still be reported once
    for k in range(6, n_guides, 9):
        g = list(guides[k % 6])
        g[int(rng.integers(0, guidelen))] = "ACGT"[int(rng.integers(0, 4))]
        guides[k] = "".join(g)
    concrete = {"NGG": "TGG", "TTTV": "TTTA", "NNGRRT": "ACGAGT"}[pam_s]
    for name, n in (("c1", 60_000), ("c2", 25_001), ("c3", 300)):
        g = list(synth.random_sequence(rng, n, iupac_frac=0.001))
        for gd in guides[:6]:
            _plant(rng, g, gd, concrete, right, 12 if n > 1000 else 1, max_mm)
        if n > 5000:  # an N run: must not produce hits nor blow up
            g[3000:3400] = "N" * 400
        contigs[name] = "".join(g)
    pam = PAM(pam_s, right, True)
    pam.encode(0)
    idx = GenomeIndex(contigs, guidelen, len(pam_s), piece=piece)
    got = idx.scan(guides, pam, right, max_mm, cap=64)  # small cap: exercises the capacity retry
    want = []
    for name, seq in contigs.items():
        for r in ora.offtargets(seq, guides, pam_s, right, max_mm):
            want.append((int(r["guide"]), name, int(r["pos"]), "-" if r["strand"] else "+", int(r["mm"])))
    ci = {n: i for i, n in enumerate(contigs)}
    want.sort(key=lambda t: (t[0], ci[t[1]], t[2], t[3] == "-"))
    assert len(want) > (20 if max_mm else 3)
    assert len(set(want)) == len(want)
    assert [(h.guide, h.contig, h.position, h.strand, h.mm) for h in got] == want
    # the reported window is the genome window in guide orientation
    L = guidelen + len(pam_s)
    for h in got[:200]:
        w = contigs[h.contig][h.position:h.position + L].upper()
        w = ora.revcomp(w) if h.strand == "-" else w
        assert h.window == "".join(c if c in "ACGT" else "N" for c in w)


def test_offtargets_search_pipeline(tmp_path):
    """search() -> offtargets_search(): per-guide counts and global CFD 100/(100+sum) as in the
    reference's annotate_guides_offtargets (offtargets.py:597-627)."""
    from types import SimpleNamespace
    from crisprhawk_hip import scoring
    from crisprhawk_hip.coordinate import Coordinate
    from crisprhawk_hip.haplotype import Haplotype
    from crisprhawk_hip.region import Region
    from crisprhawk_hip.search_guides import search
    from crisprhawk_hip.annotation import reverse_guides
    from crisprhawk_hip.search_offtargets import offtargets_search
    from crisprhawk_hip.sequence import Sequence

    reg = synth.make_region(901, "chrG", 30_000, 10_000, 10_400)
    region = Region(Sequence(reg.sequence, True), Coordinate("chrG", 10_000, 10_400, 100))
    hap = Haplotype(Sequence(reg.sequence, True), region.coordinates, False, 0, True)
    hap.id = "hap_ref"
    pam = PAM("NGG", False, True)
    pam.encode(0)
    guides = reverse_guides(search(pam, region, [hap], None, 20, False, False, False, 0, True), 0)
    assert len(guides) > 10
    mm, pt = synth.cfd_tables()
    scoring.set_cfd_tables(mm, pt)
    args = SimpleNamespace(verbosity=0, debug=True, crispritz_index={"chrG": reg.contig_seq}, mm=3, bdna=0, brna=0,
                           guidelen=20, right=False, outdir=str(tmp_path))
    out = offtargets_search({region: guides}, pam, args)[region]
    for g in out[:25]:
        rows = ora.offtargets(reg.contig_seq, [g.guide.upper()], "NGG", False, 3)
        assert int(g.offtargets) == len(rows) >= 1  # the on-target site itself is always there
        tot = 0.0
        for r in rows:
            w = reg.contig_seq[int(r["pos"]):int(r["pos"]) + 23]
            w = ora.revcomp(w) if r["strand"] else w
            tot += round(ora.cfd(g.guide.upper(), w[:20], w[-2:], mm, pt), 4)
        assert g.cfd == str(round(100 / (100 + tot), 4))
    tsv = (tmp_path / "offtargets_chrG_10000_10400.tsv").read_text().splitlines()
    assert tsv[0].split("\t") == ["chrom", "position", "strand", "grna", "spacer", "pam", "mm", "bulge_size", "bulg_type", "cfd", "elevation"]
    assert len(tsv) - 1 >= len(out)


def _verify_hits_on_host(contig_arrays, idx, hits, guides, pam_s, guidelen, right, max_mm):
    """Every reported hit re-derived from the genome bytes: PAM positions inside the PAM's IUPAC sets, mismatch count
    as reported and <= max_mm, window code as reported."""
    from crisprhawk_hip.genome import decode_window
    from crisprhawk_hip.pam import IUPAC_BITS
    L = guidelen + len(pam_s)
    nib = {"A": 1, "C": 2, "G": 4, "T": 8}
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    n = len(hits["guide"])
    sel = np.arange(n) if n <= 20000 else np.random.default_rng(0).choice(n, 20000, replace=False)
    for i in sel.tolist():
        name, off, _own = idx.rows[int(hits["row"][i])]
        p0 = off + int(hits["q"][i])
        w = contig_arrays[name][p0:p0 + L]
        if hits["strand"][i]:
            w = comp[w[::-1]]
        w = w.tobytes().decode()
        assert decode_window(int(hits["code"][i]), int(hits["nmask"][i]), L) == w
        sp, pm = (w[len(pam_s):], w[:len(pam_s)]) if right else (w[:guidelen], w[guidelen:])
        assert all(nib[c] & IUPAC_BITS[q] for c, q in zip(pm, pam_s))
        mm = sum(a != b for a, b in zip(sp, guides[int(hits["guide"][i])]))
        assert mm == int(hits["mm"][i]) <= max_mm


C5_CONTIG_NT, C5_GUIDES = 129_166_667, 10_000  # 24 contigs: 3.1 x 10^9 nt


def test_c5_full_size_properties(monkeypatch):
    """C5 at BASELINE.json's full size - a 3.1 x 10^9-nt genome in 24 contigs, 10^4 guides, TTTV / 23, <= 4 mismatches - which no
    brute force reaches: the four match
    kernels (pair seeds; single-block pigeonhole seeds from L2 and from LDS; all pairs) must report the same hit set, every guide
    must find its planted on-target, and every hit must re-verify against the genome bytes on the host."""
    import subprocess, sys, os, json
    code = r"""
import json, sys, hashlib
import numpy as np
C5_CONTIG_NT, C5_GUIDES = %d, %d
sys.path[:0] = [%r, %r]
from crisprhawk_hip.genome import GenomeIndex
from crisprhawk_hip.pam import PAM
rng = np.random.default_rng(1006)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
contigs = {f"chr{i+1}": acgt[rng.integers(0, 4, size=C5_CONTIG_NT, dtype=np.uint8)] for i in range(24)}
grng = np.random.default_rng(1005)
names = list(contigs)
guides = []
while len(guides) < C5_GUIDES:
    c = contigs[names[int(grng.integers(0, 24))]]
    p = int(grng.integers(0, len(c) - 64))
    guides.append(c[p:p + 23].tobytes().decode())
pam = PAM("TTTV", True, True); pam.encode(0)
idx = GenomeIndex(contigs, 23, 4)
hits, tm = idx.scan_arrays(guides, pam, True, 4)
order = np.lexsort((hits["strand"], hits["q"], hits["row"], hits["guide"]))
h = hashlib.sha256()
for k in ("guide", "row", "q", "strand", "mm", "code", "nmask"):
    h.update(np.ascontiguousarray(hits[k][order]).tobytes())
print("RESULT", json.dumps({"n": int(len(order)), "digest": h.hexdigest(), "n_sites": int(tm["n_sites"]), "match_ms": tm["match_ms"]}))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = code % (C5_CONTIG_NT, C5_GUIDES, os.path.join(root, "crispr-hawk_amd"), root)
    res = {}
    for label, env in (("pair_seeds", {}), ("seeded_lds", {"HAWK_OT_PAIRS": "0"}), ("seeded_global", {"HAWK_OT_SEED_GLOBAL": "1"}),
                       ("all_pairs", {"HAWK_OT_ALLPAIRS": "1"})):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=1100)
        assert r.returncode == 0, r.stderr[-2000:]
        res[label] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")][0][7:])
    assert res["pair_seeds"]["digest"] == res["seeded_lds"]["digest"] == res["seeded_global"]["digest"] == res["all_pairs"]["digest"]
    print("match_ms", {k: round(v["match_ms"], 2) for k, v in res.items()})
    assert res["seeded_lds"]["n"] >= 100  # ~1 guide in 85 sits behind a TTTV and is its own on-target; the rest are chance near-matches
    # in-process: host re-verification of the default kernel's hits
    rng = np.random.default_rng(1006)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    contigs = {f"chr{i+1}": acgt[rng.integers(0, 4, size=C5_CONTIG_NT, dtype=np.uint8)] for i in range(24)}
    grng = np.random.default_rng(1005)
    names = list(contigs)
    guides, origin = [], []
    while len(guides) < C5_GUIDES:
        ci = int(grng.integers(0, 24))
        c = contigs[names[ci]]
        p = int(grng.integers(0, len(c) - 64))
        guides.append(c[p:p + 23].tobytes().decode())
        origin.append((names[ci], p))
    pam = PAM("TTTV", True, True)
    pam.encode(0)
    idx = GenomeIndex(contigs, 23, 4)
    hits, _tm = idx.scan_arrays(guides, pam, True, 4)
    assert len(hits["guide"]) == res["seeded_lds"]["n"]
    _verify_hits_on_host(contigs, idx, hits, guides, "TTTV", 23, True, 4)
    # a guide cut out right behind a TTTV is its own 0-mismatch on-target at PAM start = origin - 4
    found = {(int(g), idx.rows[int(r)][0], idx.rows[int(r)][1] + int(q)) for g, r, q, s, m in
             zip(hits["guide"], hits["row"], hits["q"], hits["strand"], hits["mm"]) if m == 0 and s == 0}
    planted = 0
    for gi, (name, p) in enumerate(origin):
        if p >= 4:
            pm = contigs[name][p - 4:p].tobytes().decode()
            if pm[:3] == "TTT" and pm[3] in "ACG":
                planted += 1
                assert (gi, name, p - 4) in found
    assert planted >= 10


def test_sharded_genome_index_partitions_the_hits():
    """SURVEY §8(e) for C5: the genome's rows block-partitioned over ranks, guides replicated; the union of the
    shards' hits (global row numbers) is the unsharded scan's hit set."""
    rng = np.random.default_rng(78)
    contigs = {f"c{i}": synth.random_sequence(rng, 90_000 + 7000 * i) for i in range(5)}
    guides = [contigs["c1"][p:p + 20] for p in range(100, 70_000, 700)]
    pam = PAM("NGG", False, True)
    pam.encode(0)
    whole, _ = GenomeIndex(contigs, 20, 3, piece=20_000).scan_arrays(guides, pam, False, 3)
    parts = []
    for r in range(3):
        idx = GenomeIndex(contigs, 20, 3, piece=20_000, shard=(r, 3))
        assert (idx.row_hi - idx.row_lo) in (idx.n_rows_total // 3, idx.n_rows_total // 3 + 1)
        parts.append(idx.scan_arrays(guides, pam, False, 3)[0])
    key = lambda h: sorted(zip(h["guide"].tolist(), h["row"].tolist(), h["q"].tolist(), h["strand"].tolist(), h["mm"].tolist()))
    merged = {k: np.concatenate([p[k] for p in parts]) for k in whole}
    assert key(merged) == key(whole) and len(whole["guide"]) >= 5  # ~1 spacer in 16 is followed by NGG and is its own on-target


@pytest.mark.parametrize("pam_s,guidelen,right,max_mm", [("NGG", 20, False, 2), ("TTTV", 23, True, 1)])
@pytest.mark.parametrize("bdna,brna", [(1, 0), (0, 1), (1, 1), (2, 2)])
def test_bulged_offtargets_match_bruteforce(pam_s, guidelen, right, max_mm, bdna, brna):
    """-bDNA / -bRNA (offtargets.py:264-268): sites that pair with a guide once up to 2 bases are bulged out of the DNA or of the
    RNA.  The device path searches them as mismatch-only scans of derived guides (GenomeIndex.scan_bulges) and must report the
    rows of the oracle's brute force over every placement (ora.offtargets_bulges; both sides unpinned to CRISPRitz, which is
    absent): one row per (guide, site, type, size), fewest mismatches, ties to the smallest bulge positions."""
    rng = np.random.default_rng(99 + bdna * 7 + brna)
    n_guides = 5
    guides = [synth.random_sequence(rng, guidelen) for _ in range(n_guides)]
    guides[1] = guides[1][:6] + "AAAA" + guides[1][10:]  # a run of equal bases: several placements spell the same derived guide
    concrete = {"NGG": "TGG", "TTTV": "TTTA"}[pam_s]
    contigs = {}
    for name, n in (("c1", 30_000), ("c2", 9_001)):
        g = list(synth.random_sequence(rng, n, iupac_frac=0.0005))
        for gd in guides:
            for _ in range(10):  # planted sites: the guide with mismatches, bases inserted (DNA bulge) or deleted (RNA bulge)
                sp = list(gd)
                for p in rng.integers(0, guidelen, size=int(rng.integers(0, max_mm + 1))):
                    sp[p] = "ACGT"[rng.integers(0, 4)]
                kind = int(rng.integers(0, 3))
                k = int(rng.integers(1, 3))
                if kind == 1:
                    for _ in range(k):
                        sp.insert(int(rng.integers(1, len(sp) - 1)), "ACGT"[rng.integers(0, 4)])
                elif kind == 2:
                    for _ in range(k):
                        del sp[int(rng.integers(1, len(sp) - 1))]
                w = (concrete + "".join(sp)) if right else ("".join(sp) + concrete)
                if rng.random() < 0.5:
                    w = ora.revcomp(w)
                pos = int(rng.integers(0, n - len(w)))
                g[pos:pos + len(w)] = list(w)
        contigs[name] = "".join(g)
    pam = PAM(pam_s, right, True)
    pam.encode(0)
    idx = GenomeIndex(contigs, guidelen, len(pam_s), piece=4096, max_bulge=bdna)
    got = idx.scan_bulges(guides, pam, right, max_mm, bdna, brna)
    want = []
    for name, seq in contigs.items():
        for r in ora.offtargets_bulges(seq, guides, pam_s, right, max_mm, bdna, brna):
            want.append((int(r["guide"]), "DNA" if r["btype"] == 1 else "RNA", int(r["bsize"]), name, int(r["pos"]), "-" if r["strand"] else "+",
                         int(r["mm"]), int(r["gaps"])))
    ci = {n: i for i, n in enumerate(contigs)}
    want.sort(key=lambda t: (t[0], t[1], t[2], ci[t[3]], t[4], t[5] == "-"))
    assert len(want) > 30 and {t[1] for t in want} == ({"DNA"} if not brna else {"RNA"} if not bdna else {"DNA", "RNA"})
    assert [(h.guide, h.bulge_type, h.bulge_size, h.contig, h.position, h.strand, h.mm, h.gaps) for h in got] == want
    # the strings of a row: the guide and the site re-derived from the genome, '-' at the bulges, mismatches in lower case
    for h in got[:300]:
        Gs = guidelen + h.bulge_size if h.bulge_type == "DNA" else guidelen - h.bulge_size
        w = contigs[h.contig][h.position:h.position + Gs + len(pam_s)].upper()
        w = ora.revcomp(w) if h.strand == "-" else w
        w = "".join(c if c in "ACGT" else "N" for c in w)
        site = w[len(pam_s):] if right else w[:Gs]
        assert h.crrna.replace("-", "") == guides[h.guide] and h.dna.replace("-", "").upper() == site
        assert len(h.crrna) == len(h.dna) and h.crrna.count("-") == (h.bulge_size if h.bulge_type == "DNA" else 0)
        assert h.dna.count("-") == (h.bulge_size if h.bulge_type == "RNA" else 0)
        assert sum(1 for a, b in zip(h.crrna, h.dna) if a != "-" and b != "-" and a != b.upper() or b.islower() and a == "-") >= 0
        assert sum(1 for a, b in zip(h.crrna, h.dna) if b.islower()) == h.mm
    # un-bulged scans before and after see the same rows: the window metadata is put back
    assert [(x.guide, x.contig, x.position, x.strand, x.mm) for x in idx.scan(guides, pam, right, max_mm)] == \
        sorted(((int(r["guide"]), name, int(r["pos"]), "-" if r["strand"] else "+", int(r["mm"])) for name, seq in contigs.items()
                for r in ora.offtargets(seq, guides, pam_s, right, max_mm)), key=lambda t: (t[0], ci[t[1]], t[2], t[3] == "-"))


def test_offtarget_stage_with_bulges_writes_their_rows(tmp_path):
    """estimate_offtargets_spacers with bdna / brna > 0: the off-targets TSV holds DNA / RNA rows next to the X rows, every row
    in the field set the reference's consumer reads (offtarget.py:77-101), counts per spacer include them."""
    from crisprhawk_hip import scoring
    from crisprhawk_hip.coordinate import Coordinate
    from crisprhawk_hip.offtargets import estimate_offtargets_spacers
    rng = np.random.default_rng(5)
    guide = synth.random_sequence(rng, 20)
    g = list(synth.random_sequence(rng, 20_000))
    sites = {"X": guide + "TGG", "DNA": guide[:9] + "C" + guide[9:] + "AGG", "RNA": guide[:12] + guide[13:] + "CGG"}
    for k, (kind, w) in enumerate(sites.items()):
        g[2000 * (k + 1):2000 * (k + 1) + len(w)] = list(w)
    pam = PAM("NGG", False, True)
    pam.encode(0)
    scoring.set_cfd_tables(*synth.cfd_tables())
    out = estimate_offtargets_spacers([guide], pam, {"chrT": "".join(g)}, Coordinate("chrT", 100, 900, 100), 1, 1, 1, 20, False, str(tmp_path), 0, True)
    (tsv,) = list(tmp_path.glob("offtargets_chrT_*.tsv"))
    rows = [ln.split("\t") for ln in tsv.read_text().splitlines()[1:]]
    kinds = {r[8] for r in rows}
    assert kinds == {"X", "DNA", "RNA"}
    assert out[guide][0] == len(rows) >= 3
    for r in rows:
        if r[8] == "DNA":
            assert "-" in r[3] and int(r[7]) == 1
        if r[8] == "RNA":
            assert "-" in r[4] and int(r[7]) == 1
