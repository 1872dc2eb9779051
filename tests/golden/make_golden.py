#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run in the build container only (``/root/reference`` is not present on the GPU box):

    python tests/golden/make_golden.py

The reference's search path is pure Python; the only imports it cannot satisfy here are
``colorama`` (terminal colours) and ``pysam`` (FASTA/VCF file readers), neither of which is
on the hot path.  Both are replaced by empty in-process modules below so that the leaf
modules (encoder, pam, search_guides, haplotype(s), variant, guide, scores/cfdscore,
scores/deepCpf1) import; file I/O is never exercised - regions, variants and haplotypes are
built from in-memory synthetic inputs (crisprhawk_hip/synth.py) through the reference's own
classes.  Nothing from the reference is copied: the fixtures hold inputs and the outputs the
reference computed for them.

Fixtures:
  g1_tables.json.gz     nibble table, PAM bits / bitsrc / cas_system for every listed PAM
  g2_scan.json.gz       scan_haplotype hit lists (IUPAC-bearing sequences, 5 PAMs, edge ranges)
  g4_unphased.json.gz   unphased VCF: IUPAC-encoded haplotypes + indel windows built by the reference and
                        its search() output (resolve_guide expansion)
  g3_search_*.json.gz   haplotype construction + pam_search + search() + reverse_guides +
                        scorer input k-mers + CFDon (synthetic tables) for several regions
  g5_cfd.json.gz        compute_cfd on random (wt, sg, pam) triples, synthetic tables
  g6_deepcpf1.json.gz   SeqDeepCpf1 forward on random 34-mers, seeded synthetic weights
  g8_vcf_lines.json.gz  VariantRecord.read_vcf_line / split() on multi-allelic, missing-allele and
                        extra-FORMAT records
  g9_azimuth.json.gz    scores/azimuth: features/featurization.featurize_data + util.concatenate_feature_sets with the
                        learn_options save_final_model_V3(include_position=False) pickles (model_comparison.py:474-497) on
                        random 30-mers -> the 627-column matrix; model_comparison.predict driven with a locally fitted
                        GradientBoostingRegressor (100 x depth 3, lr 0.1: models/ensembles.py:28-30).  Biopython is absent:
                        Tm_NN is a restatement (columns 623-626 and whatever the trees read of them stay unpinned)
  g10_offtargets.json.gz  the off-target host stage (offtargets.py:486-627, offtarget.py:77-129): report_offtargets on a
                        CRISPRitz-format targets.txt, annotate_guides_offtargets, and the guide report with the
                        offtargets / cfd columns (reports.py:292-333, 384-404, 612-660, 877-906)
  g7_report_*.json.gz   the guide report: search -> _annotate_variants -> annotate_variants_afs ->
                        reverse_guides -> gc -> CFDon -> reports._process_data -> _collapse_report_entries ->
                        _format_report, stored as the TSV text the reference would write.  gc_content comes from
                        a restatement of Biopython's gc_fraction (Biopython is absent: that column is unpinned)
"""

import gzip
import importlib.util
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/src"


def _install_standins():
    col = types.ModuleType("colorama")

    class _Blank:
        def __getattr__(self, k):
            return ""

    col.Fore = _Blank()
    col.Style = _Blank()
    col.Back = _Blank()
    col.init = lambda *a, **k: None
    sys.modules["colorama"] = col
    ps = types.ModuleType("pysam")
    psu = types.ModuleType("pysam.utils")

    class SamtoolsError(Exception):
        pass

    psu.SamtoolsError = SamtoolsError
    ps.utils = psu
    ps.FastaFile = object
    ps.TabixFile = object
    ps.faidx = lambda *a, **k: None
    ps.tabix_index = lambda *a, **k: None
    sys.modules["pysam"] = ps
    sys.modules["pysam.utils"] = psu
    bio = types.ModuleType("Bio")  # annotation.py imports gc_fraction at module level; never called through the stand-in
    bsu = types.ModuleType("Bio.SeqUtils")

    def _absent(*a, **k):
        raise NotImplementedError("Biopython is not installed here")

    bsu.gc_fraction = _absent
    bio.SeqUtils = bsu
    sys.modules["Bio"] = bio
    sys.modules["Bio.SeqUtils"] = bsu
    h5 = types.ModuleType("h5py")  # only load_deepcpf1_weights touches it; never called
    h5.File = object
    sys.modules["h5py"] = h5


_install_standins()
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "crispr-hawk_amd"))

import numpy as np  # noqa: E402

from crisprhawk import encoder as R_encoder  # noqa: E402
from crisprhawk import pam as R_pam  # noqa: E402
from crisprhawk import search_guides as R_search  # noqa: E402
from crisprhawk.coordinate import Coordinate  # noqa: E402
from crisprhawk.sequence import Sequence  # noqa: E402
from crisprhawk.region import Region  # noqa: E402
from crisprhawk.haplotype import Haplotype  # noqa: E402
from crisprhawk import haplotypes as R_haps  # noqa: E402
from crisprhawk.variant import VariantRecord  # noqa: E402
from crisprhawk.utils import flatten_list  # noqa: E402

from crisprhawk_hip import synth  # noqa: E402


def _load_by_path(name, relpath):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "crisprhawk", relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


R_cfd = _load_by_path("ref_cfdscore", "scores/cfdscore/cfdscore.py")


def dump(name, obj):
    path = os.path.join(HERE, name)
    raw = json.dumps(obj, separators=(",", ":")).encode()
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(raw)
    print(f"{name}: {len(raw)} B json -> {os.path.getsize(path)} B gz")


# ---------------------------------------------------------------------------- G1
def g1_tables():
    table = {c: R_encoder.encode(c, 0, True)[0] for c in "ACGTNRYSWKMBDHV"}
    lower = {c: R_encoder.encode(c, 0, True)[0] for c in "acgtnryswkmbdhv"}
    pams = []
    allp = R_pam.CASXPAM + R_pam.CPF1PAM + R_pam.SACAS9PAM + R_pam.SPCAS9PAM + R_pam.XCAS9PAM + ["NAG", "NNNNGATT", "ngg"]
    for p in allp:
        for right in (False, True):
            pm = R_pam.PAM(p, right, True)
            pm.encode(0)
            pams.append(
                dict(pam=p, right=right, seq=pm.pam, rc=pm.pamrc, bits=pm.bits, bitsrc=pm.bitsrc,
                     bits_list=pm.bits_list, cas_system=pm.cas_system, length=len(pm))
            )
    dump("g1_tables.json.gz", dict(table=table, lower=lower, pams=pams))


# ---------------------------------------------------------------------------- G2
def g2_scan():
    rng = np.random.default_rng(4242)
    cases = []
    for pam_s in ["NGG", "NAG", "NNGRRT", "TTTV", "TTCN", "NRG", "NNNRRT"]:
        for n, frac in [(64, 0.0), (257, 0.15), (2000, 0.05), (33, 0.5)]:
            seq = synth.random_sequence(rng, n, frac)
            pm = R_pam.PAM(pam_s, False, True)
            pm.encode(0)
            bits = R_encoder.encode(seq, 0, True)
            L = len(pm)
            ranges = [(0, n - L + 1), (min(5, n - L), max(min(5, n - L), n - L - 3))]
            a = int(rng.integers(0, max(1, n - L)))
            b = int(rng.integers(a, n - L + 2))
            ranges.append((a, b))
            for start, stop in ranges:
                fwd, rev = R_search.scan_haplotype(pm, bits, start, stop, True)
                cases.append(dict(pam=pam_s, seq=seq, start=start, stop=stop, fwd=fwd, rev=rev))
    dump("g2_scan.json.gz", dict(cases=cases))


# ---------------------------------------------------------------------------- G3
def _ref_region(reg: synth.SynthRegion) -> Region:
    return Region(Sequence(reg.sequence, True), Coordinate(reg.contig, reg.bed_start, reg.bed_stop, synth.PADDING))


def _ref_haplotypes(reg: synth.SynthRegion, region: Region):
    """initialize_haplotypes + add_variants_phased without the pysam-backed VCF reader
    (reference haplotypes.py:106-131, 714-745)."""
    haps = [Haplotype(Sequence(region.sequence.sequence, True), region.coordinates, False, 0, True)]
    if not reg.variants:
        return haps, False, False
    records = []
    for v in reg.variants:
        vr = VariantRecord(True)
        vr.read_vcf_line(reg.vcf_fields(v), reg.samples, True)
        records.append(vr)
    variants = flatten_list([r.split() for r in records])  # haplotypes.py:88-92
    sample_variants = R_haps.compute_haplotypes_phased(variants, reg.samples)
    haps = R_haps.solve_haplotypes_phased(
        sample_variants, haps, region.sequence.sequence, region.coordinates, True, True
    )
    return haps, True, True


def _posmap_breaks(posmap):
    out, prev = [], None
    for i in range(len(posmap)):
        g = posmap[i]
        if prev is None or g != prev + 1:
            out.append([i, g])
        prev = g
    return out


def _posmap_rev_probe(hap, rng, n=64):
    """A sample of posmap_rev lookups (incl. a miss -> -1) to pin the overwrite rule
    (haplotype.py:159)."""
    keys = sorted(hap.posmap_rev.keys())
    lo, hi = keys[0], keys[-1]
    probe = set(int(x) for x in rng.integers(lo, hi + 1, size=n))
    # always probe around every posmap break
    pm = hap.posmap
    for i in range(1, len(pm)):
        if pm[i] != pm[i - 1] + 1:
            for g in (pm[i - 1], pm[i - 1] + 1, pm[i], pm[i] - 1):
                if lo <= g <= hi:
                    probe.add(int(g))
    return [[g, int(hap.posmap_rev.get(g, -1))] for g in sorted(probe)]


def g3_search(name, reg, pam_s, guidelen, right, cfd=True, kmers=True):
    rng = np.random.default_rng(99)
    region = _ref_region(reg)
    haps, variants_present, phased = _ref_haplotypes(reg, region)
    for i, h in enumerate(haps):
        h.id = f"hap_{i:08d}"  # reference ids are unseeded random strings (haplotypes.py:807-814)
    pam = R_pam.PAM(pam_s, right, True)
    pam.encode(0)
    bits = [R_encoder.encode(h.sequence.sequence, 0, True) for h in haps]
    scan = [list(R_search.compute_scan_start_stop(h, region.start, region.stop, len(pam))) for h in haps]
    hits = R_search.pam_search(pam, region, haps, bits, 0, True)
    guides = R_search.search(pam, region, haps, bits, guidelen, right, variants_present, phased, 0, True)
    hapidx = {h.id: i for i, h in enumerate(haps)}
    out_haps = []
    for h in haps:
        out_haps.append(
            dict(
                seq=h.sequence.sequence,
                samples=sorted(h.samples.split(",")),
                variants=h.variants,
                afs={k: (None if v != v else v) for k, v in h.afs.items()},
                posmap_breaks=_posmap_breaks(h.posmap),
                posmap_len=len(h.posmap),
                posmap_rev_probe=_posmap_rev_probe(h, rng),
            )
        )
    g_search = [
        [g.start, g.stop, g.strand, g.sequence, hapidx[g.hapid], bool(g.right)] for g in guides
    ]
    # a few full per-guide posmaps (search_guides.py:283-303)
    g_posmaps = [[i, [guides[i].posmap[k] for k in range(guidelen + len(pam))]]
                 for i in range(0, len(guides), max(1, len(guides) // 25))]
    obj = dict(
        contig=reg.contig, bed_start=reg.bed_start, bed_stop=reg.bed_stop,
        startp=region.start, stopp=region.stop, region_seq=reg.sequence,
        samples=reg.samples,
        variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants],
        pam=pam_s, guidelen=guidelen, right=right, cas_system=pam.cas_system,
        variants_present=variants_present, phased=phased,
        haplotypes=out_haps, scan=scan,
        hits=[[list(f), list(r)] for f, r in hits],
        guides=g_search, guide_posmaps=g_posmaps,
    )
    # stage 6 slice: annotation.reverse_guides == Guide.reverse_complement on strand 1
    # (annotation.py:27-51 cannot be imported: it pulls Bio; the loop is two lines)
    for g in guides:
        if g.strand == 1:
            g.reverse_complement()
    obj["reversed"] = [[g.sequence, g.guide, g.pam, bool(g.right)] for g in guides]
    if kmers:  # scoring.py:50-67
        obj["kmers"] = [g.sequence[(10 - 4):(-10 + 3)].upper() for g in guides]
    if cfd:
        mm, pt = synth.cfd_tables()
        mmd, pamd = synth.cfd_tables_as_dicts(mm, pt)
        groups = R_search.group_guides_position(guides, True)  # same keying as scoring.py:303-349
        order, scores = [], []
        gid = {id(g): i for i, g in enumerate(guides)}
        for _, grp in groups.items():
            gref, members = grp[0], grp[1]
            for sg in members:
                order.append(gid[id(sg)])
                if gref is None:
                    scores.append(None)  # scores/crisprhawk_scores.py:81-82 -> NaN -> "NA"
                else:
                    scores.append(R_cfd.compute_cfd(gref.guide, sg.guide, sg.pam[-2:], mmd, pamd, True))
        obj["cfdon_order"] = order
        obj["cfdon"] = scores
    dump(f"g3_search_{name}.json.gz", obj)
    print(f"   {name}: {len(haps)} haplotypes, {sum(len(f)+len(r) for f, r in hits)} hits, {len(guides)} guides")


def g3_all():
    # C1: BASELINE configs[0]/[1]
    g3_search("c1", synth.config_c1(), "NGG", 20, False)
    # the survey probe shape: 3 kb + 4-sample phased VCF with SNV/del/ins
    reg = synth.make_region(3001, "chrP", 6000, 1500, 4500)
    synth.add_phased_variants(reg, 3002, 60, 4, frac_snv=0.6, frac_del=0.2, af_min=0.15, af_max=0.6)
    g3_search("phased4", reg, "NGG", 20, False)
    # 30 kb x 8 samples (<= 17 haplotypes)
    reg = synth.make_region(3011, "chrQ", 40000, 5000, 35000)
    synth.add_phased_variants(reg, 3012, 400, 8, af_min=0.05, af_max=0.5)
    g3_search("phased16", reg, "NGG", 20, False)
    # Cpf1: TTTV, 23 nt, guide on the right of the PAM
    reg = synth.make_region(3021, "chrC", 12000, 2000, 10000)
    synth.add_phased_variants(reg, 3022, 120, 4, frac_snv=0.7, frac_del=0.15, af_min=0.1, af_max=0.6)
    g3_search("cpf1", reg, "TTTV", 23, True, cfd=False)
    # SaCas9 NNGRRT 21 nt on an N/IUPAC-bearing reference, no VCF (search-level N semantics)
    reg = synth.make_region(3031, "chrN", 5000, 500, 4500, iupac_frac=0.02)
    g3_search("iupac", reg, "NNGRRT", 21, False, cfd=False)
    # dense indels incl. sites hugging the BED edges / region ends
    reg = synth.make_region(3041, "chrE", 3000, 700, 1500)
    synth.add_phased_variants(reg, 3042, 80, 3, frac_snv=0.2, frac_del=0.4, max_indel=6, af_min=0.2, af_max=0.7, edge_margin=1)
    g3_search("indel_dense", reg, "NGG", 20, False)
    # tiny region: padding-dominated, in-range filter bites
    reg = synth.make_region(3051, "chrT", 600, 200, 230)
    synth.add_phased_variants(reg, 3052, 6, 2, frac_snv=0.5, frac_del=0.25, max_indel=3, af_min=0.3, af_max=0.8)
    g3_search("tiny", reg, "NGG", 20, False)
    # XCas9 NGN (every G) on the C1 region: hit density stress
    g3_search("ngn", synth.make_region(3061, "chrX", 4000, 300, 3300), "NGN", 20, False)


# ---------------------------------------------------------------------------- G7 (guide report)
def _gc_fraction_restated(seq: str) -> float:
    """Biopython 1.83 SeqUtils.gc_fraction(seq) with its default ambiguous="remove": (C+G+S) / (A+C+G+T+S+W+U),
    0 for an empty denominator.  Restated because Biopython is not installed: gc_content is unpinned."""
    gc = sum(seq.count(c) for c in "CGScgs")
    n = gc + sum(seq.count(c) for c in "ATWUatwu")
    return gc / n if n else 0.0


def _ref_haplotypes_unphased(reg: synth.SynthRegion, region: Region):
    """The reference's add_variants_unphased body (haplotypes.py:672-712) on records whose genotypes read a/b."""
    haps = [Haplotype(Sequence(region.sequence.sequence, True), region.coordinates, False, 0, True)]
    records = []
    for v in reg.variants:
        vr = VariantRecord(True)
        fields = reg.vcf_fields(v)
        fields[9:] = [g.replace("|", "/") for g in fields[9:]]
        vr.read_vcf_line(fields, reg.samples, False)
        records.append(vr)
    variants = flatten_list([r.split() for r in records])
    snvs, indels = R_haps.classify_variants(variants)
    if snvs:
        haps.extend(R_haps.compute_snvs_haplotype_unphased(snvs, reg.samples, region.sequence.sequence, region.coordinates, False, True))
    for indel in indels:
        if region.coordinates.startp <= indel.position < region.coordinates.stopp:
            haps.extend(R_haps.create_indels_haplotype_unphased(indel, snvs, region, False, True))
    return haps, True, False


def g7_report(name, reg, pam_s, guidelen, right, cfd=True, unphased=False):
    from crisprhawk import annotation as R_ann
    from crisprhawk import reports as R_rep
    import math
    region = _ref_region(reg)
    haps, variants_present, phased = _ref_haplotypes_unphased(reg, region) if unphased else _ref_haplotypes(reg, region)
    for i, h in enumerate(haps):
        h.id = f"hap_{i:08d}"
    pam = R_pam.PAM(pam_s, right, True)
    pam.encode(0)
    bits = [R_encoder.encode(h.sequence.sequence, 0, True) for h in haps]
    guides = R_search.search(pam, region, haps, bits, guidelen, right, variants_present, phased, 0, True)
    # annotation.annotate_guides (annotation.py:545-600) without the BED annotations
    guides = R_ann._annotate_variants(guides, 0, True)
    guides = R_ann.annotate_variants_afs(guides, 0)
    guides = R_ann.reverse_guides(guides, 0)
    for g in guides:
        g.gc = _gc_fraction_restated(g.guide)  # annotation.gc_content with the restated gc_fraction
    if cfd:  # scoring.cfdon_score (scoring.py:352-387) with the synthetic tables; guides leave in group order
        mm, pt = synth.cfd_tables()
        mmd, pamd = synth.cfd_tables_as_dicts(mm, pt)
        groups = R_search.group_guides_position(guides, True)
        out = []
        for _, grp in groups.items():
            gref, members = grp[0], grp[1]
            for sg in members:
                sg.cfdon_score = float("nan") if gref is None else float(
                    R_cfd.compute_cfd(gref.guide, sg.guide, sg.pam[-2:], mmd, pamd, True))
                out.append(sg)
        guides = out
    df = R_rep._process_data(region, guides, pam, [], [], [], [], False, False)
    n_rows = len(df)
    df = R_rep._collapse_report_entries(df, pam, [], [], False)
    df = R_rep._format_report(df, pam, right, [], [], False)
    tsv = df.to_csv(sep="\t", index=False)
    obj = dict(
        contig=reg.contig, bed_start=reg.bed_start, bed_stop=reg.bed_stop, startp=region.start, stopp=region.stop,
        region_seq=reg.sequence, samples=reg.samples,
        variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants],
        pam=pam_s, guidelen=guidelen, right=right, cfd=cfd, target=str(region.coordinates),
        haplotypes=[dict(id=h.id, samples=h.samples, variants=h.variants, seq=(h.sequence.sequence if unphased else None),
                         start=h.coordinates.start, afs={k: (None if v != v else v) for k, v in h.afs.items()}) for h in haps],
        unphased=unphased,
        rows_before_collapse=n_rows, report_tsv=tsv,
    )
    if name is None:  # tools/campaign_report_fixtures.py collects the dicts
        return obj
    dump(f"g7_report_{name}.json.gz", obj)
    print(f"   {name}: {len(haps)} haplotypes, {n_rows} guide rows -> {len(df)} report rows")


def g7_all():
    reg = synth.make_region(3001, "chrP", 6000, 1500, 4500)
    synth.add_phased_variants(reg, 3002, 60, 4, frac_snv=0.6, frac_del=0.2, af_min=0.15, af_max=0.6)
    g7_report("phased4", reg, "NGG", 20, False)
    reg = synth.make_region(3011, "chrQ", 40000, 5000, 35000)
    synth.add_phased_variants(reg, 3012, 400, 8, af_min=0.05, af_max=0.5)
    g7_report("phased16", reg, "NGG", 20, False)
    reg = synth.make_region(3021, "chrC", 12000, 2000, 10000)
    synth.add_phased_variants(reg, 3022, 120, 4, frac_snv=0.7, frac_del=0.15, af_min=0.1, af_max=0.6)
    g7_report("cpf1", reg, "TTTV", 23, True, cfd=False)
    reg = synth.make_region(3041, "chrE", 3000, 700, 1500)
    synth.add_phased_variants(reg, 3042, 80, 3, frac_snv=0.2, frac_del=0.4, max_indel=6, af_min=0.2, af_max=0.7, edge_margin=1)
    g7_report("indel_dense", reg, "NGG", 20, False)
    g7_report("c1", synth.config_c1(), "NGG", 20, False)
    # SaCas9: no efficiency score columns at all
    reg = synth.make_region(3071, "chrS", 8000, 1000, 7000)
    synth.add_phased_variants(reg, 3072, 90, 3, af_min=0.2, af_max=0.6)
    g7_report("sacas9", reg, "NNGRRT", 21, False, cfd=False)
    # unphased VCF: IUPAC haplotypes + indel windows -> resolve_guide -> the same annotation / report chain
    reg = synth.make_region(3081, "chrV", 5000, 1000, 4000)
    synth.add_phased_variants(reg, 3082, 45, 3, frac_snv=0.8, frac_del=0.1, max_indel=3, af_min=0.2, af_max=0.6)
    g7_report("unphased", reg, "NGG", 20, False, unphased=True)


# ---------------------------------------------------------------------------- G4 (unphased)
G4_CASES = {  # fixture name -> (synth parameters, PAM, guide length, right); tests rebuild the inputs from the stored parameters
    "g4_unphased": (dict(region=[4001, "chrU", 5000, 1000, 4000], variants=[4002, 40, 3],
                         kw=dict(frac_snv=0.8, frac_del=0.1, max_indel=3, af_min=0.2, af_max=0.6)), "NGG", 20, False),
    "g4_unphased_cpf1": (dict(region=[4011, "chrV", 7000, 1500, 5500], variants=[4012, 70, 4],
                              kw=dict(frac_snv=0.6, frac_del=0.2, max_indel=5, af_min=0.2, af_max=0.7)), "TTTV", 23, True),
    "g4_unphased_dense": (dict(region=[4021, "chrW", 4000, 800, 3200], variants=[4022, 110, 6],
                               kw=dict(frac_snv=0.7, frac_del=0.15, max_indel=8, af_min=0.15, af_max=0.8)), "NGG", 20, False),
}


def g4_unphased(name="g4_unphased"):
    """Unphased VCF: the reference encodes heterozygous SNVs as lower-case IUPAC letters, builds one
    200-bp window haplotype per indel, and search() expands every candidate through resolve_guide
    (search_guides.py:163-257, 473-480).  The haplotypes are built by the reference's own
    add_variants_unphased body (haplotypes.py:672-712, VCF object replaced by its sample list)."""
    obj = g4_unphased_case(*G4_CASES[name])
    dump(f"{name}.json.gz", obj)
    print(f"   unphased: {len(obj['haplotypes'])} haplotypes, {sum(len(f)+len(r) for f, r in obj['hits'])} hits, {len(obj['guides'])} guides")


def g4_unphased_case(sp, pam_s, guidelen, right):
    """One unphased case as a fixture dict (also what tools/campaign_unphased_fixtures.py collects)."""
    reg = synth.make_region(*sp["region"])
    synth.add_phased_variants(reg, *sp["variants"], **sp["kw"])
    region = _ref_region(reg)
    haps = [Haplotype(Sequence(region.sequence.sequence, True), region.coordinates, False, 0, True)]
    records = []
    for v in reg.variants:
        vr = VariantRecord(True)
        fields = reg.vcf_fields(v)
        fields[9:] = [g.replace("|", "/") for g in fields[9:]]
        vr.read_vcf_line(fields, reg.samples, False)
        records.append(vr)
    variants = flatten_list([r.split() for r in records])
    snvs, indels = R_haps.classify_variants(variants)
    if snvs:
        haps.extend(R_haps.compute_snvs_haplotype_unphased(snvs, reg.samples, region.sequence.sequence, region.coordinates, False, True))
    for indel in indels:
        if region.coordinates.startp <= indel.position < region.coordinates.stopp:
            haps.extend(R_haps.create_indels_haplotype_unphased(indel, snvs, region, False, True))
    for i, h in enumerate(haps):
        h.id = f"hap_{i:08d}"
    pam = R_pam.PAM(pam_s, right, True)
    pam.encode(0)
    bits = [R_encoder.encode(h.sequence.sequence, 0, True) for h in haps]
    scan = [list(R_search.compute_scan_start_stop(h, region.start, region.stop, len(pam))) for h in haps]
    hits = R_search.pam_search(pam, region, haps, bits, 0, True)
    guides = R_search.search(pam, region, haps, bits, guidelen, right, True, False, 0, True)
    hapidx = {h.id: i for i, h in enumerate(haps)}
    out_haps = [dict(seq=h.sequence.sequence, samples=h.samples, variants=h.variants, contig=h.contig,
                     coord=[h.coordinates.startp, h.coordinates.stopp, h.coordinates.start, h.coordinates.stop],
                     posmap_breaks=_posmap_breaks(h.posmap), posmap_len=len(h.posmap),
                     variant_alleles={str(k): [list(t) for t in v] for k, v in h.variant_alleles.items()}) for h in haps]
    return dict(
        synth=sp,
        contig=reg.contig, bed_start=reg.bed_start, bed_stop=reg.bed_stop, startp=region.start, stopp=region.stop,
        region_seq=reg.sequence, pam=pam_s, guidelen=guidelen, right=right, haplotypes=out_haps, scan=scan,
        hits=[[list(f), list(r)] for f, r in hits],
        guides=[[g.start, g.stop, g.strand, g.sequence, hapidx[g.hapid], bool(g.right), g.samples] for g in guides])


# ---------------------------------------------------------------------------- G5
def g5_cfd():
    rng = np.random.default_rng(5005)
    mm, pt = synth.cfd_tables()
    mmd, pamd = synth.cfd_tables_as_dicts(mm, pt)
    cases = []
    for k in range(12000):
        n = 20 if k % 7 else int(rng.integers(17, 24))
        wt = synth.random_sequence(rng, n)
        sg = list(wt)
        nmm = int(rng.integers(0, 6))
        for p in rng.integers(0, n, size=nmm):
            sg[p] = "ACGT"[rng.integers(0, 4)]
        sg = "".join(sg)
        if k % 11 == 0:  # case noise: compute_cfd upper-cases
            sg = sg.lower()
        pam = synth.random_sequence(rng, 2)
        cases.append([wt, sg, pam, R_cfd.compute_cfd(wt, sg, pam, mmd, pamd, True)])
    dump("g5_cfd.json.gz", dict(seed=2001, cases=cases))


# ---------------------------------------------------------------------------- G6
def g6_deepcpf1():
    import torch

    R_dc = _load_by_path("ref_seqdeepcpf1", "scores/deepCpf1/seqdeepcpf1.py")
    w = synth.deepcpf1_weights()
    model = R_dc.SeqDeepCpf1()
    with torch.no_grad():
        model.conv.weight.copy_(torch.from_numpy(w["conv_w"]))
        model.conv.bias.copy_(torch.from_numpy(w["conv_b"]))
        for i, nm in enumerate(["fc1", "fc2", "fc3", "output"]):
            getattr(model, nm).weight.copy_(torch.from_numpy(w[f"w{i + 1}"]))
            getattr(model, nm).bias.copy_(torch.from_numpy(w[f"b{i + 1}"]))
    model.eval()
    rng = np.random.default_rng(6006)
    seqs = [synth.random_sequence(rng, 34) for _ in range(1000)]
    scores = R_dc.compute_deepcpf1(model, R_dc.preprocess(seqs))
    dump("g6_deepcpf1.json.gz", dict(seed=2002, seqs=seqs, scores=scores))


# ---------------------------------------------------------------------------- G8 (VCF records)
def g8_vcf_lines():
    """VariantRecord.read_vcf_line / split() (variant.py:286-331) on hand-made and random records: multi-allelic
    sites, multi-digit allele indices, missing alleles, extra FORMAT fields, no AF."""
    rng = np.random.default_rng(8008)
    samples = [f"S{i:03d}" for i in range(37)]
    recs = []

    def gts(n_alt, p_missing=0.05, extra=False):
        out = []
        for _ in samples:
            a = [("." if rng.random() < p_missing else str(int(rng.integers(0, n_alt + 1)))) for _ in range(2)]
            g = "|".join(a)
            if extra:
                g += f":{int(rng.integers(1, 99))}:{rng.random():.2f}"
            out.append(g)
        return out

    pos = 1000
    for i in range(60):
        pos += int(rng.integers(1, 40))
        kind = i % 6
        if kind == 0:
            ref, alts, info = "A", ["G"], "AF=0.25"
        elif kind == 1:
            ref, alts, info = "ACG", ["A"], "AC=3;AF=0.0125;AN=10"
        elif kind == 2:
            ref, alts, info = "T", ["TGA", "C"], "AF=0.1,0.2"
        elif kind == 3:
            ref, alts, info = "C", ["A", "G", "T"], "DP=10"
        elif kind == 4:
            ref, alts, info = "GT", ["G", "GTT", "AT"], "AF=0.01,0.02,0.5;DB"
        else:
            ref, alts = "A", [c * (1 + j % 3) for j, c in enumerate("CGTCGTCGTCGT")]  # 12 ALT alleles: two-digit indices
            info = "AF=" + ",".join(f"{0.001 * (j + 1):.3f}" for j in range(12))
        fmt = "GT:DP:GQ" if i % 4 == 1 else "GT"
        recs.append(["chrV", str(pos), ".", ref, ",".join(alts), "50", "PASS" if i % 5 else "q10", info, fmt]
                    + gts(len(alts), extra=(fmt != "GT")))
    out = []
    for fields in recs:
        vr = VariantRecord(True)
        vr.read_vcf_line(fields, samples, True)
        out.append(dict(
            fields=fields, alt=vr.alt, vtype=vr.vtype, afs=[None if a != a else a for a in vr.afs], ids=vr.id, filter=vr.filter,
            samples=[[sorted(s0), sorted(s1)] for s0, s1 in vr.samples],
            split=[[v.position, v.ref, v.alt[0], v.id[0], v.vtype[0]] for v in vr.split()],
        ))
    dump("g8_vcf_lines.json.gz", dict(samples=samples, records=out))


# ---------------------------------------------------------------------------- G9 (Azimuth / Rule Set 2)
def _tm_nn_restated(seq, **kw):
    """Biopython 1.83 Bio.SeqUtils.MeltingTemp.Tm_NN(seq) with its defaults (nn_table DNA_NN3 = Allawi & SantaLucia
    1997, dnac1 = dnac2 = 25 nM, Na = 50 mM, saltcorr = 5, no mismatches / dangling ends), restated from the published
    algorithm because Biopython is not installed: whatever passes through it is UNPINNED."""
    import math
    seq = str(seq).upper()
    nn = {"AA": (-7.9, -22.2), "AT": (-7.2, -20.4), "TA": (-7.2, -21.3), "CA": (-8.5, -22.7), "GT": (-8.4, -22.4),
          "CT": (-7.8, -21.0), "GA": (-8.2, -22.2), "CG": (-10.6, -27.2), "GC": (-9.8, -24.4), "GG": (-8.0, -19.9)}
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    dh = ds = 0.0
    for end in (seq[0], seq[-1]):  # init_A/T (2.3, 4.1), init_G/C (0.1, -2.8); the other init terms of DNA_NN3 are zero
        if end in "AT":
            dh += 2.3
            ds += 4.1
        else:
            dh += 0.1
            ds += -2.8
    for i in range(len(seq) - 1):
        pair = seq[i:i + 2]
        if pair not in nn:  # the table lists one strand of every complementary pair
            pair = comp[pair[1]] + comp[pair[0]]
        dh += nn[pair][0]
        ds += nn[pair][1]
    k = (25.0 - 25.0 / 2.0) * 1e-9
    ds += 0.368 * (len(seq) - 1) * math.log(50.0 * 1e-3)
    return (1000.0 * dh) / (ds + 1.987 * math.log(k)) - 273.15


def _azimuth_modules():
    """features/featurization.py, util.py and model_comparison.py imported from where they lie, as members of their
    package (their relative imports resolve to the reference's own files) but without running the package __init__s,
    which pull rs3 / elevation / h5py.  matplotlib, pylab and Biopython are absent: plotting is never reached, and the
    one Biopython function on the path (Tm_NN) is the restatement above."""
    import importlib

    class _Any:
        def __getattr__(self, k):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    def shell(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        m.__getattr__ = lambda k: _Any()
        sys.modules[name] = m
        return m

    for n in ("matplotlib", "matplotlib.pylab", "matplotlib.pyplot", "pylab"):
        shell(n)
    bio = shell("Bio", __path__=[])
    for n in ("SeqUtils", "Seq", "Entrez", "SeqIO"):
        setattr(bio, n, shell(f"Bio.{n}"))
    bio.SeqUtils.__path__ = []
    bio.SeqUtils.MeltingTemp = shell("Bio.SeqUtils.MeltingTemp", Tm_NN=_tm_nn_restated)
    for pk in ("crisprhawk.scores", "crisprhawk.scores.azimuth", "crisprhawk.scores.azimuth.features", "crisprhawk.scores.azimuth.models"):
        m = types.ModuleType(pk)
        m.__path__ = [os.path.join(REF, *pk.split("."))]
        m.__package__ = pk
        sys.modules[pk] = m
    feat = importlib.import_module("crisprhawk.scores.azimuth.features.featurization")
    util = importlib.import_module("crisprhawk.scores.azimuth.util")
    mc = importlib.import_module("crisprhawk.scores.azimuth.model_comparison")
    return feat, util, mc


def g9_azimuth():
    import pandas
    from sklearn.ensemble import GradientBoostingRegressor
    from crisprhawk_hip.scoring import azimuth_model_from_sklearn
    feat, util, mc = _azimuth_modules()
    # what save_final_model_V3(include_position=False) pickles beside the model (model_comparison.py:474-497), plus the
    # keys run_models / setup add that featurize_data reads (order 2: model_comparison.py:499; one process)
    learn_options = {
        "V": 3, "testing_non_binary_target_name": "ranks", "include_pi_nuc_feat": True, "gc_features": True,
        "nuc_features": True, "include_gene_position": False, "include_NGGX_interaction": True, "include_Tm": True,
        "include_strand": False, "include_gene_feature": False, "include_gene_guide_feature": 0, "extra pairs": False,
        "weighted": None, "training_metric": "spearmanr", "NDGC_k": 10, "cv": "gene", "include_gene_effect": False,
        "include_drug": False, "include_sgRNAscore": False, "adaboost_loss": "ls", "adaboost_alpha": 0.5,
        "normalize_features": False, "adaboost_CV": False,
        "order": 2, "num_proc": 1, "include_known_pairs": False, "include_microhomology": False,
    }
    rng = np.random.default_rng(9009)
    seqs = [synth.random_sequence(rng, 30) for _ in range(1200)]
    seqs[0] = "A" * 30
    seqs[1] = "G" * 30
    seqs[2] = "ACGT" * 7 + "AC"
    seqs[3] = seqs[3][:4] + "GC" * 10 + seqs[3][24:]  # GC count 20
    seqs[4] = seqs[4][:4] + "AT" * 10 + seqs[4][24:]  # GC count 0
    seqs[5] = seqs[5][:4] + "G" * 10 + "A" * 10 + seqs[5][24:]  # GC count exactly 10: neither flag
    arr = np.array(seqs)
    Xdf = pandas.DataFrame(columns=["30mer", "Strand"], data=list(zip(arr, ["NA"] * len(arr))))
    gene_position = pandas.DataFrame(columns=["Percent Peptide", "Amino Acid Cut position"],
                                     data=list(zip(np.ones(len(arr)) * -1, np.ones(len(arr)) * -1)))
    fs = feat.featurize_data(Xdf, dict(learn_options), pandas.DataFrame(), gene_position, pam_audit=False, length_audit=False)
    inputs, dim, dimsum, names = util.concatenate_feature_sets(fs)
    assert inputs.shape == (len(seqs), 627), inputs.shape
    head = inputs[:, :623]
    assert np.array_equal(head, np.rint(head)) and head.min() >= 0 and head.max() <= 30
    # the predict() plumbing (model_comparison.py:507-585) with a model of the shipped shape fitted on random targets
    y = rng.normal(0.5, 0.2, size=len(seqs)) + 0.05 * inputs[:, 606] - 0.02 * inputs[:, 623] / 10.0
    gbr = GradientBoostingRegressor(n_estimators=100, max_depth=3, learning_rate=0.1, random_state=1)
    gbr.fit(inputs[:800], y[:800])
    preds = mc.predict(arr, None, None, model=(gbr, dict(learn_options)), pam_audit=False, length_audit=False)
    flat = azimuth_model_from_sklearn(gbr)
    uses_tm = sorted(set(int(f) for f in flat["feature"] if f >= 623))
    dump("g9_azimuth.json.gz", dict(
        seqs=seqs, dim={k: int(v) for k, v in dim.items()}, names=[str(n) for n in names],
        features_int=head.astype(int).ravel().tolist(), tm=inputs[:, 623:].tolist(), tm_source="restated: Biopython absent, unpinned",
        model={k: (np.asarray(v).tolist() if not np.isscalar(v) else float(v)) for k, v in flat.items()},
        model_uses_tm_features=uses_tm, predictions=[float(p) for p in preds]))
    print(f"   azimuth: {len(seqs)} 30-mers, {dimsum} features in {len(dim)} sets; trees read Tm columns {uses_tm}")


# ---------------------------------------------------------------------------- G10 (off-target host stage)
def _brute_force_targets(genome, guides_seqs, pam_s, right, mm_max):
    """CRISPRitz-format rows (offtarget.py:89-101 reads fields 0,1,2,3,4,6,7,8) for every genome window, both strands,
    whose PAM positions IUPAC-match and whose spacer differs from a guide in <= mm_max positions: the INPUT of the stage
    under test, enumerated here independently of the package (CRISPRitz itself is absent)."""
    iupac = {"A": "A", "C": "C", "G": "G", "T": "T", "N": "ACGT", "R": "AG", "Y": "CT", "S": "CG", "W": "AT", "K": "GT",
             "M": "AC", "B": "CGT", "D": "AGT", "H": "ACT", "V": "ACG"}
    comp = str.maketrans("ACGTN", "TGCAN")
    pl, rows = len(pam_s), []
    for contig, seq in genome.items():
        for strand in "+-":
            for g in guides_seqs:
                gl = len(g)
                L = gl + pl
                for p in range(len(seq) - L + 1):
                    w = seq[p:p + L]
                    if strand == "-":
                        w = w[::-1].translate(comp)
                    sp, pm = (w[pl:], w[:pl]) if right else (w[:gl], w[gl:])
                    if any(b not in iupac[q] for b, q in zip(pm, pam_s)) or "N" in sp:
                        continue
                    nmm = sum(a != b for a, b in zip(sp, g))
                    if nmm > mm_max:
                        continue
                    dsp = "".join(t if t == q else t.lower() for t, q in zip(sp, g))
                    cr = ("N" * pl + g) if right else (g + "N" * pl)
                    dna = (pm + dsp) if right else (dsp + pm)
                    rows.append(f"X\t{cr}\t{dna}\t{contig}\t{p}\t{p}\t{strand}\t{nmm}\t0\t{nmm}")
    return rows


def g10_offtargets():
    import tempfile
    import importlib
    from crisprhawk import annotation as R_ann
    from crisprhawk import reports as R_rep
    # crisprhawk.scores' __init__ imports rs3 / elevation / ...: register the package without running it, and give
    # offtargets.py the one name it imports from crisprhawk_scores (elevation: never called, compute_elevation=False)
    if "crisprhawk.scores" not in sys.modules:
        m = types.ModuleType("crisprhawk.scores")
        m.__path__ = [os.path.join(REF, "crisprhawk", "scores")]
        sys.modules["crisprhawk.scores"] = m
    cs = types.ModuleType("crisprhawk.scores.crisprhawk_scores")
    cs.elevation = None
    sys.modules["crisprhawk.scores.crisprhawk_scores"] = cs
    R_off = importlib.import_module("crisprhawk.offtargets")
    mm, pt = synth.cfd_tables()
    mmd, pamd = synth.cfd_tables_as_dicts(mm, pt)
    R_off.load_mismatch_pam_scores = lambda debug: (mmd, pamd)  # the tables are input data (the pickles are not downloadable)

    out = {}
    for name, pam_s, guidelen, right, seedbase in (("ngg", "NGG", 20, False, 10010), ("cpf1", "TTTV", 23, True, 10020)):
        rng = np.random.default_rng(seedbase)
        reg = synth.make_region(seedbase + 1, "chrO", 2400, 600, 1500)
        synth.add_phased_variants(reg, seedbase + 2, 25, 3, frac_snv=0.7, frac_del=0.15, af_min=0.2, af_max=0.6)
        region = _ref_region(reg)
        haps, variants_present, phased = _ref_haplotypes(reg, region)
        for i, h in enumerate(haps):
            h.id = f"hap_{i:08d}"
        pam = R_pam.PAM(pam_s, right, True)
        pam.encode(0)
        bits = [R_encoder.encode(h.sequence.sequence, 0, True) for h in haps]
        guides = R_search.search(pam, region, haps, bits, guidelen, right, variants_present, phased, 0, True)
        guides = R_ann._annotate_variants(guides, 0, True)
        guides = R_ann.annotate_variants_afs(guides, 0)
        guides = R_ann.reverse_guides(guides, 0)
        for g in guides:
            g.gc = _gc_fraction_restated(g.guide)
        cfdon = pam_s == "NGG"
        if cfdon:
            groups = R_search.group_guides_position(guides, True)
            outg = []
            for _, grp in groups.items():
                gref, members = grp[0], grp[1]
                for sg in members:
                    sg.cfdon_score = float("nan") if gref is None else float(
                        R_cfd.compute_cfd(gref.guide, sg.guide, sg.pam[-2:], mmd, pamd, True))
                    outg.append(sg)
            guides = outg
        # genome: the region's own contig (on-target rows) + two decoys with near-copies of some guide sites planted
        uniq = sorted(R_off._filter_guides(guides))
        decoys = {}
        for cn, n in (("chrD1", 5000), ("chrD2", 4000)):
            s = list(synth.random_sequence(rng, n))
            for _ in range(40):
                g = guides[int(rng.integers(0, len(guides)))]
                site = list((g.pam + g.guide if right else g.guide + g.pam).upper())
                sp0 = len(pam_s) if right else 0
                for q in rng.integers(0, guidelen, size=int(rng.integers(0, 6))):
                    site[sp0 + int(q)] = "ACGT"[int(rng.integers(0, 4))]
                if rng.random() < 0.5:
                    site = list("".join(site)[::-1].translate(str.maketrans("ACGT", "TGCA")))
                at = int(rng.integers(0, n - len(site)))
                s[at:at + len(site)] = site
            decoys[cn] = "".join(s)
        genome = {reg.contig: reg.contig_seq, **decoys}
        mm_max = 4
        rows = _brute_force_targets(genome, uniq, pam_s, right, mm_max)
        rng.shuffle(rows)  # CRISPRitz writes in thread order: report_offtargets sorts
        # hand-made bulge rows on real guides: a DNA bulge has '-' in the crRNA, an RNA bulge '-' in the DNA
        pl = len(pam_s)
        for k, g in enumerate(uniq[:6]):
            cut = 5 + k
            site = synth.random_sequence(rng, 1)
            if k % 2 == 0:
                crsp, dsp, bt = g[:cut] + "-" + g[cut:], g[:cut] + site.lower() + g[cut:], "DNA"
            else:
                crsp, dsp, bt = g, g[:cut] + "-" + g[cut + 1:], "RNA"
            obs = "".join("ACGT"[int(rng.integers(0, 4))] if c == "N" else c for c in pam_s.replace("V", "A").replace("R", "A"))
            cr = ("N" * pl + crsp) if right else (crsp + "N" * pl)
            dna = (obs + dsp) if right else (dsp + obs)
            rows.insert(int(rng.integers(0, len(rows))), f"{bt}\t{cr}\t{dna}\tchrD1\t{100 + 37 * k}\t{100 + 37 * k}\t{'+-'[k % 2]}\t{k % 3}\t1\t{k % 3 + 1}")
        header = "#Bulge_type\tcrRNA\tDNA\tChromosome\tPosition\tCluster Position\tDirection\tMismatches\tBulge_Size\tTotal"
        targets_txt = header + "\n" + "\n".join(rows) + "\n"
        with tempfile.TemporaryDirectory() as td:
            tf = os.path.join(td, "x.targets.txt")
            with open(tf, "w") as f:
                f.write(targets_txt)
            ots = R_off.report_offtargets(tf, region, pam, guidelen, [], [], False, right, td, 0, True)
            rep = os.path.join(td, f"offtargets_{region.contig}_{region.start + 100}_{region.stop - 100}.tsv")
            with open(rep) as f:
                ot_tsv = f.read()
        guides = R_off.annotate_guides_offtargets(ots, guides, 0)
        per_guide = [[g.guide, int(g.offtargets), g.cfd] for g in guides]
        case = dict(
            contig=reg.contig, bed_start=reg.bed_start, bed_stop=reg.bed_stop, startp=region.start, stopp=region.stop,
            region_seq=reg.sequence, samples=reg.samples,
            variants=[[v.pos, v.ref, v.alt, v.af, ["".join(str(int(x)) for x in row) for row in v.gt]] for v in reg.variants],
            pam=pam_s, guidelen=guidelen, right=right, cfdon=cfdon, mm=mm_max, target=str(region.coordinates),
            haplotypes=[dict(id=h.id, samples=h.samples, variants=h.variants, start=h.coordinates.start,
                             afs={k: (None if v != v else v) for k, v in h.afs.items()}) for h in haps],
            genome=genome, unique_spacers=uniq, targets_txt=targets_txt, offtargets_tsv=ot_tsv, per_guide=per_guide,
            offtarget_objects=[[o.grna_, o.grna, o.spacer, o.cfd, o.elevation] for o in ots[:200]])
        def guide_report():
            try:  # the guide report with the offtargets / cfd columns
                df = R_rep._process_data(region, guides, pam, [], [], [], [], True, False)
                df = R_rep._collapse_report_entries(df, pam, [], [], True)
                df = R_rep._format_report(df, pam, right, [], [], True)
                return df.to_csv(sep="\t", index=False), None
            except Exception as e:  # recorded as the reference's behaviour for this PAM class
                return None, f"{type(e).__name__}: {e}"
        case["report_tsv"], case["report_error"] = guide_report()
        # the same stage on the bulge-free rows alone: what a search with -bDNA 0 -bRNA 0 (the reference's default) returns
        plain = header + "\n" + "\n".join(r for r in rows if r.startswith("X\t")) + "\n"
        with tempfile.TemporaryDirectory() as td:
            tf = os.path.join(td, "x.targets.txt")
            with open(tf, "w") as f:
                f.write(plain)
            ots = R_off.report_offtargets(tf, region, pam, guidelen, [], [], False, right, td, 0, True)
            with open(os.path.join(td, f"offtargets_{region.contig}_{region.start + 100}_{region.stop - 100}.tsv")) as f:
                case["nobulge_offtargets_tsv"] = f.read()
        guides = R_off.annotate_guides_offtargets(ots, guides, 0)
        case["nobulge_per_guide"] = [[g.guide, int(g.offtargets), g.cfd] for g in guides]
        case["nobulge_report_tsv"], _ = guide_report()
        out[name] = case
        print(f"   offtargets {name}: {len(uniq)} spacers, {len(rows)} target rows, {len(guides)} guides, report: "
              f"{'ok' if case['report_tsv'] else case['report_error']}")
    dump("g10_offtargets.json.gz", out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10"]
    if "g1" in which:
        g1_tables()
    if "g2" in which:
        g2_scan()
    if "g3" in which:
        g3_all()
    if "g4" in which:
        for _name in G4_CASES:
            g4_unphased(_name)
    if "g5" in which:
        g5_cfd()
    if "g6" in which:
        g6_deepcpf1()
    if "g7" in which:
        g7_all()
    if "g8" in which:
        g8_vcf_lines()
    if "g10" in which:  # before g9: g9 re-registers crisprhawk.scores' sub-packages
        g10_offtargets()
    if "g9" in which:
        g9_azimuth()
