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


@pytest.mark.parametrize("n_guides", [6, 200, 2300])  # 6: all pairs; 200: seeded, one LDS chunk; 2300: three chunks
@pytest.mark.parametrize("pam_s,guidelen,right,max_mm,piece", [("NGG", 20, False, 4, 1 << 22), ("TTTV", 23, True, 3, 4096),
                                                              ("NNGRRT", 21, False, 2, 10000), ("NGG", 20, False, 0, 1 << 22),
                                                              ("NGG", 17, False, 6, 1 << 22)])
def test_offtarget_scan_matches_bruteforce(pam_s, guidelen, right, max_mm, piece, n_guides):
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
