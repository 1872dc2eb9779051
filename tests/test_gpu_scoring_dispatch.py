"""Rows a19 / a22 / a23 against reference-generated fixtures: the Azimuth kernel vs the reference's own featurisation
and predict() plumbing (g9), the scoring_guides dispatcher (scoring.py:749-867), and the off-target stage - device scan,
device CFD, report, per-guide aggregates - vs the reference's report_offtargets / annotate_guides_offtargets (g10)."""
import os
import types

import numpy as np
import pytest

from crisprhawk_hip import _lib, scoring, synth
from crisprhawk_hip.annotation import reverse_guides
from crisprhawk_hip.crisprhawk_error import (CrisprHawkAzimuthScoreError, CrisprHawkDeepCpf1ScoreError, CrisprHawkOffTargetsError,
                                             CrisprHawkRs3ScoreError)
from crisprhawk_hip.genome import GenomeIndex
from crisprhawk_hip.offtargets import (_filter_guides, annotate_guides_offtargets, estimate_offtargets, report_offtargets, search as ot_search)
from crisprhawk_hip.pam import PAM
from crisprhawk_hip.search_guides import search
from crisprhawk_hip.search_offtargets import offtargets_search
from test_gpu_api import _build
from oracle import oracle as ora
from util import load_golden

pytestmark = pytest.mark.gpu


def test_azimuth_kernel_against_reference_featurization_and_predict():
    g9 = load_golden("g9_azimuth.json.gz")
    n = len(g9["seqs"])
    model = {k: (np.array(v) if isinstance(v, list) else v) for k, v in g9["model"].items()}
    scoring.set_azimuth_model(model)
    got, feats = scoring.azimuth(g9["seqs"], return_features=True)
    want = np.array(g9["features_int"], dtype=np.float64).reshape(n, 623)
    assert np.array_equal(feats[:, :623], want)  # 623 columns of the reference's featurize_data, exactly
    assert np.max(np.abs(feats[:, 623:] - np.array(g9["tm"]))) < 1e-9  # Tm_NN: restated on both sides (Biopython absent)
    assert np.max(np.abs(np.array(got) - np.array(g9["predictions"]))) < 1e-6  # model_comparison.predict, north_star's tolerance
    # np.ndarray[str] input, as scores/crisprhawk_scores.py:31-44 passes it
    assert scoring.azimuth(np.array(g9["seqs"][:5])) == pytest.approx(g9["predictions"][:5], abs=1e-6)


def _args(**kw):
    base = dict(threads=4, verbosity=0, debug=True, guidelen=20, right=False, compute_elevation=False, mm=4, bdna=0, brna=0,
                offtargets_annotations=[], offtargets_annotation_colnames=[], outdir="", crispritz_config=None, crispritz_index=None)
    base.update(kw)
    return types.SimpleNamespace(**base)


def _guides_for(case):
    fx = load_golden(f"g3_search_{case}.json.gz")
    region, haps = _build(fx)
    pam = PAM(fx["pam"], fx["right"], True)
    pam.encode(0)
    guides = search(pam, region, haps, None, fx["guidelen"], fx["right"], fx["variants_present"], fx["phased"], 0, True)
    return fx, region, pam, reverse_guides(guides, 0)


SLOTS = ("azimuth_score", "rs3_score", "plmcrispr_score", "cfdon_score", "crispron_score", "sgdesigner_score", "deepcpf1_score",
         "elevationon_score")


def _filled(guides):
    return {s for s in SLOTS if any(getattr(g, s) != "NA" for g in guides)}


def test_device_tm_nn_against_biopythons_documented_value():
    """hawk_tm_nn runs the device function behind k_azimuth's Tm columns (az_tm) on the 28-mer whose Tm_NN Biopython documents
    (60.32 with the defaults the reference calls it with, featurization.py:358-397), and agrees with the oracle on random k-mers."""
    from crisprhawk_hip import scoring
    from test_oracle_golden import TM_NN_KNOWN
    seq, want = TM_NN_KNOWN
    got = scoring.tm_nn([seq])[0]
    assert f"{got:0.2f}" == f"{want:0.2f}" and abs(got - ora.tm_nn(seq)) < 1e-9
    rng = np.random.default_rng(77)
    for ln in (5, 8, 30, 32):
        seqs = ["".join("ACGT"[b] for b in rng.integers(0, 4, ln)) for _ in range(300)]
        assert np.allclose(scoring.tm_nn(seqs), [ora.tm_nn(x) for x in seqs], rtol=0, atol=1e-9)
    with pytest.raises(_lib.HawkStatusError):
        scoring.tm_nn(["ACGTN"])


def test_scoring_guides_dispatch_by_cas_system():
    """scoring.py:749-867: SpCas9-class PAMs get azimuth, rs3, cfdon (in that order, the list leaving in CFDon's group
    order); Cpf1 (TTTV --right) gets deepcpf1; any other PAM (SaCas9 NNGRRT) gets nothing."""
    g9 = load_golden("g9_azimuth.json.gz")
    scoring.set_azimuth_model({k: (np.array(v) if isinstance(v, list) else v) for k, v in g9["model"].items()})
    scoring.set_cfd_tables(*synth.cfd_tables())
    scoring.set_deepcpf1_weights(synth.deepcpf1_weights(2002))
    calls = []
    rng = np.random.default_rng(3)
    rs3_model = dict(tree_off=np.array([0, 3], np.int32), feature=np.array([0, -1, -1], np.int32), left=np.array([1, 0, 0], np.int32),
                     right=np.array([2, 0, 0], np.int32), threshold=np.array([0.5, 0, 0]), value=np.array([0.0, -1.0, 1.0]), init=0.0,
                     learning_rate=1.0, n_features=1)

    def feat(kmers):
        calls.append(len(kmers))
        return np.array([[1.0 if k[0] in "AC" else 0.0] for k in kmers])
    scoring.set_rs3_model(rs3_model, feat)
    scoring.skip_scorers()

    # SpCas9
    fx, region, pam, guides = _guides_for("phased4")
    ids = {id(g): i for i, g in enumerate(guides)}
    kmers = scoring._extract_guide_sequences(guides)
    want_az = scoring.azimuth(kmers)
    out = scoring.scoring_guides({region: guides}, pam, None, _args())
    scored = out[region]
    assert [ids[id(g)] for g in scored] == fx["cfdon_order"]  # cfdon_score regroups (scoring.py:383), the others keep order
    assert _filled(scored) == {"azimuth_score", "rs3_score", "cfdon_score"}
    for g in scored:
        i = ids[id(g)]
        assert g.azimuth_score == str(round(float(want_az[i]), 4))
        assert g.rs3_score == ("1.0" if kmers[i][0] in "AC" else "-1.0")  # feature 1 > 0.5 -> right leaf
    for g, want in zip(scored, fx["cfdon"]):
        assert g.cfdon_score == ("NA" if want is None else str(round(want, 4)))
    assert calls == [len(guides)]

    # Cpf1
    fx, region, pam, guides = _guides_for("cpf1")
    assert pam.cas_system == 1
    want_dc = scoring.deepcpf1(scoring._extract_guide_sequences(guides))
    scored = scoring.scoring_guides({region: guides}, pam, None, _args(guidelen=23, right=True))[region]
    assert scored is guides and _filled(scored) == {"deepcpf1_score"}
    assert [g.deepcpf1_score for g in scored] == [str(round(float(x), 4)) for x in want_dc]

    # SaCas9: no efficiency score at all
    fx, region, pam, guides = _guides_for("iupac")
    scored = scoring.scoring_guides({region: guides}, pam, None, _args(guidelen=21))[region]
    assert scored is guides and _filled(scored) == set()

    # TTTV without --right is not a Cpf1 system (pam.py:114-125): nothing is scored
    pam_l = PAM("TTTV", False, True)
    pam_l.encode(0)
    assert pam_l.cas_system not in (0, 1, 4)


def test_scoring_guides_without_models_raises_like_the_reference():
    fx, region, pam, guides = _guides_for("c1")
    scoring.skip_scorers()
    saved = scoring._AZIMUTH_MODEL, scoring._RS3, scoring._DEEPCPF1_W
    try:
        scoring._AZIMUTH_MODEL = None
        with pytest.raises(CrisprHawkAzimuthScoreError):
            scoring.scoring_guides({region: guides}, pam, None, _args())
        g9 = load_golden("g9_azimuth.json.gz")
        scoring.set_azimuth_model({k: (np.array(v) if isinstance(v, list) else v) for k, v in g9["model"].items()})
        scoring._RS3 = None
        with pytest.raises(CrisprHawkRs3ScoreError):
            scoring.scoring_guides({region: guides}, pam, None, _args())
        # an explicit opt-out is the only way to leave a scorer of the system out
        scoring.skip_scorers("rs3")
        scoring.set_cfd_tables(*synth.cfd_tables())
        scored = scoring.scoring_guides({region: guides}, pam, None, _args())[region]
        assert _filled(scored) == {"azimuth_score", "cfdon_score"}
        scoring.skip_scorers()
        fx2, region2, pam2, guides2 = _guides_for("cpf1")
        scoring._DEEPCPF1_W = None
        with pytest.raises(CrisprHawkDeepCpf1ScoreError):
            scoring.scoring_guides({region2: guides2}, pam2, None, _args(guidelen=23, right=True))
        from crisprhawk_hip.crisprhawk_error import CrisprHawkElevationScoreError
        scoring.skip_scorers("azimuth", "rs3", "cfdon")
        with pytest.raises(CrisprHawkElevationScoreError):
            scoring.scoring_guides({region: guides}, pam, None, _args(compute_elevation=True))
    finally:
        scoring._AZIMUTH_MODEL, scoring._RS3, scoring._DEEPCPF1_W = saved
        scoring.skip_scorers()


def _fields(line):
    f = line.split()
    return (f[0], f[1], f[2], f[3], int(f[4]), f[6], int(f[7]), int(f[8]))  # what offtarget.py:89-101 reads


@pytest.mark.parametrize("name", ["ngg", "cpf1"])
def test_offtarget_stage_against_reference(name, tmp_path):
    fx = load_golden("g10_offtargets.json.gz")[name]
    scoring.set_cfd_tables(*synth.cfd_tables())
    region, haps = _build(fx)
    pam = PAM(fx["pam"], fx["right"], True)
    pam.encode(0)
    guides = search(pam, region, haps, None, fx["guidelen"], fx["right"], True, True, 0, True)
    guides = reverse_guides(guides, 0)
    if fx["cfdon"]:
        guides = scoring.cfdon_score(guides, 0, True)
    assert [g.guide for g in guides] == [x[0] for x in fx["per_guide"]]
    assert sorted(_filter_guides(guides)) == fx["unique_spacers"]

    # (1) the reference's stage on the reference's input file: report text, Offtarget fields, per-guide aggregates
    tf = tmp_path / "x.targets.txt"
    tf.write_text(fx["targets_txt"])
    ots = report_offtargets(str(tf), region, pam, fx["guidelen"], [], [], False, fx["right"], str(tmp_path), 0, True)
    rep = tmp_path / f"offtargets_{fx['contig']}_{fx['bed_start']}_{fx['bed_stop']}.tsv"
    assert rep.read_text() == fx["offtargets_tsv"]
    assert [[o.grna_, o.grna, o.spacer, o.cfd, o.elevation] for o in ots[:200]] == fx["offtarget_objects"]
    annotate_guides_offtargets(ots, guides, 0)
    assert [[g.guide, int(g.offtargets), g.cfd] for g in guides] == fx["per_guide"]

    # (2) the device scan in CRISPRitz's place: the same rows as the independent enumeration the fixture's file holds
    #     (its hand-made bulge rows aside: the fixture's search was made without bulges)
    genome = GenomeIndex(fx["genome"], fx["guidelen"], len(pam))
    lines = ot_search(genome, fx["unique_spacers"], pam, fx["right"], fx["mm"], 0, True)
    want = sorted(_fields(ln) for ln in fx["targets_txt"].splitlines()[1:] if ln.startswith("X"))
    assert sorted(_fields(ln) for ln in lines) == want

    # (3) estimate_offtargets with the reference's seventeen arguments, through offtargets_search (search_offtargets.py:44-62)
    out2 = tmp_path / "est"
    out2.mkdir()
    res = offtargets_search({region: guides}, pam, _args(guidelen=fx["guidelen"], right=fx["right"], mm=fx["mm"], outdir=str(out2),
                                                        crispritz_index=fx["genome"]))
    bulged = {}
    for ln in fx["targets_txt"].splitlines()[1:]:
        if not ln.startswith("X"):
            f = ln.split()
            sp = (f[1][len(pam):] if fx["right"] else f[1][:-len(pam)]).replace("-", "")
            bulged[sp] = bulged.get(sp, 0) + 1
    for g, (_, n_ot, _cfd) in zip(res[region], fx["per_guide"]):
        assert int(g.offtargets) == n_ot - bulged.get(g.guide.upper(), 0)
    got_rows = (out2 / rep.name).read_text().splitlines()
    want_rows = [r for r in fx["offtargets_tsv"].splitlines() if r.split("\t")[8] in ("bulg_type", "X")]
    assert sorted(got_rows) == sorted(want_rows)  # same rows; among equal (chrom, position) the file order is the scan's
    with pytest.raises(CrisprHawkOffTargetsError):
        estimate_offtargets(guides, pam, genome, region, None, 4, 1, 0, [], [], fx["guidelen"], False, fx["right"], 1, "", 0, True)


@pytest.mark.parametrize("name", ["ngg", "cpf1"])
def test_search_files_with_estimate_offtargets(name, tmp_path):
    """FASTA + BED + VCF -> pipeline.search_files(estimate_offtargets=genome): the guide report with the `offtargets` / `cfd`
    columns where the reference puts them (reports.py:612-660) and offtargets_{contig}_{start}_{stop}.tsv
    (offtargets.py:530-544), against what the reference's stage made of the bulge-free rows of the same search."""
    from crisprhawk_hip import pipeline, readers
    fx = load_golden("g10_offtargets.json.gz")[name]
    fa, bed, vcf = str(tmp_path / "g.fa"), str(tmp_path / "r.bed"), str(tmp_path / "v.vcf")
    readers.write_fasta(fa, fx["contig"], fx["genome"][fx["contig"]], 60)
    with open(bed, "w") as f:
        f.write(f"{fx['contig']}\t{fx['bed_start']}\t{fx['bed_stop']}\n")
    rows = [[fx["contig"], str(p), ".", r, a, ".", "PASS", f"AF={af:.6g}", "GT"] + [f"{g[0]}|{g[1]}" for g in gts]
            for p, r, a, af, gts in fx["variants"]]
    readers.write_vcf(vcf, fx["contig"], fx["samples"], rows, False)
    out = tmp_path / "out"
    (path,) = pipeline.search_files(fa, bed, [vcf], fx["pam"], fx["guidelen"], fx["right"], str(out), cfd_tables=synth.cfd_tables(),
                                    estimate_offtargets=fx["genome"], mm=fx["mm"]).values()
    got = open(path).read()
    ot_path = out / f"offtargets_{fx['contig']}_{fx['bed_start']}_{fx['bed_stop']}.tsv"
    assert sorted(ot_path.read_text().splitlines()) == sorted(fx["nobulge_offtargets_tsv"].splitlines())
    keys = [tuple(r.split("\t")[:2]) for r in ot_path.read_text().splitlines()[1:]]
    assert keys == sorted(keys, key=lambda k: (k[0], int(k[1])))  # sorted by (chrom, position)
    if fx["nobulge_report_tsv"] is not None:
        assert got == fx["nobulge_report_tsv"]
    else:
        # the reference's own report step ends in KeyError: 'cfd' for a Cpf1 PAM with --estimate-offtargets (it groups on a
        # column it never made, reports.py:1000-1003); here the report is written: `offtargets` without `cfd`
        assert fx["report_error"] == "KeyError: 'cfd'"
        head = got.splitlines()[0].split("\t")
        assert head[-3:] == ["offtargets", "target", "haplotype_id"] and "cfd" not in head
        per = {sp.upper(): n for sp, n, _ in fx["nobulge_per_guide"]}
        col_sp, col_n = head.index("sgRNA_sequence"), head.index("offtargets")
        for line in got.splitlines()[1:]:
            f = line.split("\t")
            assert int(f[col_n]) == per[f[col_sp].upper()]
    # bulges of up to 2 bases are enumerated since round 4 (tests/test_gpu_offtargets.py holds them to the brute force); beyond that
    # the stage refuses with the reference's error class
    with pytest.raises(CrisprHawkOffTargetsError):
        pipeline.search_files(fa, bed, [vcf], fx["pam"], fx["guidelen"], fx["right"], str(out), cfd_tables=synth.cfd_tables(),
                              estimate_offtargets=fx["genome"], mm=fx["mm"], brna=3)
