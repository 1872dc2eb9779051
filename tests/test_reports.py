"""the guide report (annotate -> score -> collapse -> sort) against the TSV text the reference
produced for the same inputs (tests/golden/g7_report_*.json.gz, built by tests/golden/make_golden.py through
the reference's own annotation.py / reports.py).  The CPU test feeds the report assembler from the oracle's
rows and groups; the GPU test feeds it from the device table and hawk_table_collapse."""
import os

import numpy as np
import pytest

from crisprhawk_hip import reports, synth
from crisprhawk_hip.hapset import PosSegments, segments_from_posmap
from crisprhawk_hip.pam import PAM
from oracle import oracle as ora
from util import load_golden, oracle_haplotypes

CASES = ["phased4", "phased16", "cpf1", "indel_dense", "c1", "sacas9"]


class _Hap:
    def __init__(self, label, posmap, n):
        self.samples, self.variants, self.id = label["samples"], label["variants"], label["id"]
        self.afs = {k: (float("nan") if v is None else v) for k, v in label["afs"].items()}
        rel, gen = segments_from_posmap(posmap)
        self.segments = PosSegments(rel, gen, n)


def _oracle_side(fx):
    pam_s, guidelen, right = fx["pam"], fx["guidelen"], fx["right"]
    haps = oracle_haplotypes(fx)
    assert len(haps) == len(fx["haplotypes"])
    scan = [ora.scan_bounds(h["posmap"], fx["startp"], fx["stopp"], len(pam_s)) for h in haps]
    hs = ora.HapSet([h["seq"] for h in haps], [h["posmap"] for h in haps], [h["samples"] == ["REF"] for h in haps], scan)
    labels = [_Hap(lb, h["posmap"], len(h["seq"])) for lb, h in zip(fx["haplotypes"], haps)]
    return hs, labels


def _pam(fx):
    pam = PAM(fx["pam"], fx["right"], True)
    pam.encode(0)
    return pam


@pytest.mark.parametrize("case", CASES)
def test_report_from_oracle_rows_matches_reference_tsv(case):
    fx = load_golden(f"g7_report_{case}.json.gz")
    hs, labels = _oracle_side(fx)
    pam_s, guidelen, right = fx["pam"], fx["guidelen"], fx["right"]
    res = ora.search(hs, pam_s, guidelen, right)
    g = res.guides
    assert len(g["start"]) == fx["rows_before_collapse"]
    cfd = None
    if fx["cfd"]:
        mm, pt = synth.cfd_tables()
        _, _, _, cfd, _ = ora.reverse_and_cfdon(res, hs.is_ref, guidelen, len(pam_s), mm, pt)
    isref_row = np.asarray(hs.is_ref)[g["hap"]]
    groups, gc = ora.collapse_rows(g["start"], g["stop"], g["strand"], isref_row, res.windows, guidelen, len(pam_s), right)
    perm, off, num, den = [], [0], [], []
    for key, rows in groups.items():
        perm += rows
        off.append(len(perm))
        num.append(gc[key][0]); den.append(gc[key][1])
    inp = reports.ReportInput(g["start"], g["stop"], g["strand"], g["hap"], g["pos"], res.windows, cfd, np.array(perm), np.array(off),
                              np.array(num), np.array(den), guidelen, len(pam_s), right)
    df = reports.report_frame(inp, labels, _pam(fx), fx["contig"], fx["target"], with_cfdon=fx["cfd"])
    assert reports.to_tsv(df) == fx["report_tsv"]
    # the columnar assembler (group-level inputs, ragged joins in the library's host helper) writes the same text
    G = reports.ReportGroups.from_report_input(inp)
    df2 = reports.report_from_groups(G, labels, _pam(fx), fx["contig"], fx["target"], with_cfdon=fx["cfd"])
    assert reports.to_tsv(df2) == fx["report_tsv"]
    # variant_id / af three ways: candidates + walk inside the library for every alt row (hawk_host_variant_window + _polish_windows),
    # the numpy route with the library walking only the rows with an indel among their candidates (hawk_host_polish_rows), and the
    # Python mirror of annotation.polish_guide_variants - the same variants named
    lab = reports.HapLabels.from_objects(labels)
    assert lab.seg_csr is not None
    calls = {"all": 0, "indel": []}
    orig_all, orig_rows = reports._variant_columns_native, reports._polish_rows_native

    def counted_all(*a, **k):
        r = orig_all(*a, **k)
        calls["all"] += r is not None
        return r
    reports._variant_columns_native = counted_all
    reports._polish_rows_native = lambda *a, **k: (calls["indel"].append(len(a[4])), orig_rows(*a, **k))[1]
    try:
        native = reports.group_columns(G, lab, _pam(fx), fx["contig"], fx["target"], with_cfdon=fx["cfd"])
        assert calls["all"] == (1 if len(lab.var_idx) else 0) and not calls["indel"]
        reports._variant_columns_native = lambda *a, **k: None
        native_rows = reports.group_columns(G, lab, _pam(fx), fx["contig"], fx["target"], with_cfdon=fx["cfd"])
    finally:
        reports._variant_columns_native, reports._polish_rows_native = orig_all, orig_rows
    lab.seg_csr = None
    mirror = reports.group_columns(G, lab, _pam(fx), fx["contig"], fx["target"], with_cfdon=fx["cfd"])
    for c in ("variant_id", "af"):
        assert native[0][c].strings() == mirror[0][c].strings() == native_rows[0][c].strings(), c
    if case == "indel_dense":
        assert calls["indel"] and calls["indel"][0] > 50  # the helper really had rows to walk
    # ... and the columns written by the library's TSV writer are the reference's text
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "r.tsv")
        reports.write_report_tsv(path, *native)
        assert open(path).read() == fx["report_tsv"]


def test_samples_column_general_cases():
    """The vectorised `samples` column against the per-group reference logic (reports.py:767-810) on member sets the
    fixtures do not hold: collapsed haplotypes with several labels, both copies of a sample in one group, members in
    arbitrary order, sample names of different widths, multi-allelic genotypes, an unphased panel, a REF group."""
    rng = np.random.default_rng(5)
    names = ["S1", "S10", "NA12878", "HG00096", "x"]

    class H:
        def __init__(self, samples, hid):
            self.samples, self.id = samples, hid
    haps = [H("REF", "hap_ref")]
    for n in names:
        haps += [H(f"{n}:1|0", f"h{len(haps)}"), H(f"{n}:0|1", f"h{len(haps) + 1}"), H(f"{n}:2|0", f"h{len(haps) + 2}")]
    haps += [H("S1:1|1,x:0|1", "hA,hB"), H("HG00096:1|1", "hC")]
    member_off, member_hap = [0], []
    want_s, want_h = [], []
    for g in range(300):
        k = int(rng.integers(1, 8))
        mem = rng.choice(np.arange(1, len(haps)), size=k, replace=False) if g % 17 else np.array([0])
        member_hap += mem.tolist()
        member_off.append(len(member_hap))
        want_s.append(reports.collapse_samples([haps[int(x)].samples for x in mem]))
        want_h.append(reports.collapse_haplotype_ids([haps[int(x)].id for x in mem]))
    off, mh = np.array(member_off), np.array(member_hap)
    assert reports._samples_column(off, mh, [h.samples for h in haps]) == want_s
    assert reports._hapids_column(off, mh, [h.id for h in haps]) == want_h
    unph = [H("REF", "r")] + [H(f"{n}:0/1", f"u{i}") for i, n in enumerate(names)]
    off2, mh2 = np.array([0, 2, 3, 5]), np.array([1, 3, 0, 2, 5])
    want = [reports.collapse_samples([unph[int(x)].samples for x in mh2[a:b]]) for a, b in zip(off2[:-1], off2[1:])]
    assert reports._samples_column(off2, mh2, [h.samples for h in unph]) == want


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_report_from_device_table_matches_reference_tsv(case):
    from crisprhawk_hip.hapset import DeviceHapSet, HostHaplotype
    fx = load_golden(f"g7_report_{case}.json.gz")
    hs, labels = _oracle_side(fx)
    pam_s, guidelen, right = fx["pam"], fx["guidelen"], fx["right"]
    ds = DeviceHapSet([HostHaplotype(seq, lb.segments, r, sc) for seq, lb, r, sc in zip(hs.seqs, labels, hs.is_ref, hs.scan)])
    bits, bitsrc, _, _ = ora.pam_encode(pam_s)
    mm, pt = synth.cfd_tables() if fx["cfd"] else (None, None)
    tab = ds.search(bits, bitsrc, len(pam_s), guidelen, right, mm, pt, download=False, collapse=True)
    assert tab.n_rows == fx["rows_before_collapse"]
    groups = tab.export_groups()  # device group export -> columnar assembler (what pipeline.search_files runs)
    df2 = reports.report_from_groups(groups, labels, _pam(fx), fx["contig"], fx["target"], with_cfdon=fx["cfd"],
                                     is_ref_hap=np.asarray(ds.is_ref, dtype=bool))
    assert reports.to_tsv(df2) == fx["report_tsv"]
    df = reports.report_frame(reports.ReportInput.from_table(tab), labels, _pam(fx), fx["contig"], fx["target"], with_cfdon=fx["cfd"])
    assert reports.to_tsv(df) == fx["report_tsv"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_files_to_report_matches_reference_tsv(tmp_path, case):
    """FASTA + BED + VCF files -> pipeline.search_files -> the TSV the reference wrote for the same inputs."""
    from crisprhawk_hip import pipeline, readers
    fx = load_golden(f"g7_report_{case}.json.gz")
    contig_seq = "N" * (fx["startp"] - 1) + fx["region_seq"] + "ACGT" * 10
    fa, bed, vcf = str(tmp_path / "g.fa"), str(tmp_path / "r.bed"), str(tmp_path / "v.vcf.gz")
    readers.write_fasta(fa, fx["contig"], contig_seq, 80)
    with open(bed, "w") as f:
        f.write(f"{fx['contig']}\t{fx['bed_start']}\t{fx['bed_stop']}\n")
    vcfs = []
    if fx["variants"]:
        rows = [[fx["contig"], str(p), ".", r, a, ".", "PASS", f"AF={af:.6g}", "GT"] + [f"{g[0]}|{g[1]}" for g in gts]
                for p, r, a, af, gts in fx["variants"]]
        readers.write_vcf(vcf, fx["contig"], fx["samples"], rows, True)
        vcfs = [vcf]
    paths = pipeline.search_files(fa, bed, vcfs, fx["pam"], fx["guidelen"], fx["right"], str(tmp_path / "out"),
                                  cfd_tables=synth.cfd_tables() if fx["cfd"] else None)
    (path,) = paths.values()
    assert os.path.basename(path) == f"crisprhawk_guides__{fx['contig']}_{fx['bed_start']}_{fx['bed_stop']}_{fx['pam']}_{fx['guidelen']}.tsv"
    assert open(path).read() == fx["report_tsv"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["phased4", "indel_dense"])
def test_host_built_fallback_writes_the_same_report(tmp_path, monkeypatch, case):
    """Phased records the device expansion declines (overlapping records on one chromosome copy) fall back to
    the host haplotype builder; forced here on inputs the device does take, the fallback must write the reference's TSV."""
    from crisprhawk_hip import pipeline, readers
    from crisprhawk_hip.expand import HaplotypeBuildError
    fx = load_golden(f"g7_report_{case}.json.gz")
    contig_seq = "N" * (fx["startp"] - 1) + fx["region_seq"] + "ACGT" * 10
    fa, bed, vcf = str(tmp_path / "g.fa"), str(tmp_path / "r.bed"), str(tmp_path / "v.vcf")
    readers.write_fasta(fa, fx["contig"], contig_seq, 80)
    with open(bed, "w") as f:
        f.write(f"{fx['contig']}\t{fx['bed_start']}\t{fx['bed_stop']}\n")
    rows = [[fx["contig"], str(p), ".", r, a, ".", "PASS", f"AF={af:.6g}", "GT"] + [f"{g[0]}|{g[1]}" for g in gts]
            for p, r, a, af, gts in fx["variants"]]
    readers.write_vcf(vcf, fx["contig"], fx["samples"], rows, False)

    def refuse(*a, **k):
        raise HaplotypeBuildError("a chromosome copy carries overlapping variants")
    monkeypatch.setattr(pipeline, "expand_from_vcf", refuse)
    (path,) = pipeline.search_files(fa, bed, [vcf], fx["pam"], fx["guidelen"], fx["right"], str(tmp_path / "out"),
                                    cfd_tables=synth.cfd_tables() if fx["cfd"] else None).values()
    assert open(path).read() == fx["report_tsv"]


@pytest.mark.gpu
def test_unphased_files_to_report_matches_reference_tsv(tmp_path):
    """An unphased VCF through pipeline.search_files: host haplotype construction (haplotypes.add_variants_unphased) ->
    device search -> resolve_guide -> annotate / CFDon / report, against the TSV the reference wrote for the same
    records (g7_report_unphased).  The reference walks indel carriers in set order, so haplotype ids are matched by
    content before the text is compared."""
    import io
    import pandas as pd
    from crisprhawk_hip import pipeline, readers
    fx = load_golden("g7_report_unphased.json.gz")
    contig_seq = "N" * (fx["startp"] - 1) + fx["region_seq"] + "ACGT" * 10
    fa, bed, vcf = str(tmp_path / "g.fa"), str(tmp_path / "r.bed"), str(tmp_path / "v.vcf")
    readers.write_fasta(fa, fx["contig"], contig_seq, 80)
    with open(bed, "w") as f:
        f.write(f"{fx['contig']}\t{fx['bed_start']}\t{fx['bed_stop']}\n")
    rows = [[fx["contig"], str(p), ".", r, a, ".", "PASS", f"AF={af:.6g}", "GT"] + [f"{g[0]}/{g[1]}" for g in gts]
            for p, r, a, af, gts in fx["variants"]]
    readers.write_vcf(vcf, fx["contig"], fx["samples"], rows, False)
    (path,) = pipeline.search_files(fa, bed, [vcf], fx["pam"], fx["guidelen"], fx["right"], str(tmp_path / "out"),
                                    cfd_tables=synth.cfd_tables()).values()
    got = pd.read_csv(path, sep="\t", dtype=str, keep_default_na=False)
    want = pd.read_csv(io.StringIO(fx["report_tsv"]), sep="\t", dtype=str, keep_default_na=False)
    assert list(got.columns) == list(want.columns) and len(got) == len(want)
    for c in got.columns:
        if c != "haplotype_id":
            assert (got[c] == want[c]).all(), (c, got[c][got[c] != want[c]].head(), want[c][got[c] != want[c]].head())
    assert (got["haplotype_id"].str.count(",") == want["haplotype_id"].str.count(",")).all()


@pytest.mark.gpu
def test_pipeline_model_scorers_fill_their_columns(tmp_path):
    """With DeepCpf1 weights set, the Cpf1 report's score column holds str(round(score, 4)) of the device scorer
    on each row's 34-mer (guide.py:456-462); every other column stays as in the reference TSV."""
    import io
    import pandas as pd
    from crisprhawk_hip import pipeline, readers, scoring
    fx = load_golden("g7_report_cpf1.json.gz")
    contig_seq = "N" * (fx["startp"] - 1) + fx["region_seq"] + "ACGT" * 10
    fa, bed, vcf = str(tmp_path / "g.fa"), str(tmp_path / "r.bed"), str(tmp_path / "v.vcf")
    readers.write_fasta(fa, fx["contig"], contig_seq, 80)
    with open(bed, "w") as f:
        f.write(f"{fx['contig']}\t{fx['bed_start']}\t{fx['bed_stop']}\n")
    rows = [[fx["contig"], str(p), ".", r, a, ".", "PASS", f"AF={af:.6g}", "GT"] + [f"{g[0]}|{g[1]}" for g in gts]
            for p, r, a, af, gts in fx["variants"]]
    readers.write_vcf(vcf, fx["contig"], fx["samples"], rows, False)
    (path,) = pipeline.search_files(fa, bed, [vcf], fx["pam"], fx["guidelen"], fx["right"], str(tmp_path / "out"),
                                    deepcpf1_weights=synth.deepcpf1_weights()).values()
    got = pd.read_csv(path, sep="\t", dtype=str, keep_default_na=False)
    want = pd.read_csv(io.StringIO(fx["report_tsv"]), sep="\t", dtype=str, keep_default_na=False)
    assert list(got.columns) == list(want.columns) and len(got) == len(want)
    for c in got.columns:
        if c != "score_deepcpf1":
            assert (got[c] == want[c]).all(), c
    # the report groups on the score column, so equal order also means the scores did not split or reorder groups
    assert (got["score_deepcpf1"] != "NA").all() and np.isfinite(got["score_deepcpf1"].astype(float)).all()


def test_native_tsv_writer_equals_the_dataframe_route(tmp_path):
    """reports.write_report_tsv (hawk_host_tsv_write: the columns as they are kept -> the file, multi-threaded) against
    DataFrame -> to_tsv on every column kind, a permuted row order, empty fields, negative and large integers; a field csv
    quoting would touch sends the report through pandas; an empty report is the header alone."""
    import pandas as pd
    rng = np.random.default_rng(5)
    for n in (1, 7, 5000):
        cols = {"chr": reports.ConstCol("chr1", n), "start": reports.IntCol(rng.integers(-5, 10**15, n)),
                "sgRNA_sequence": reports.FixedCol(rng.integers(65, 70, (n, 20)).astype(np.uint8)),
                "strand": reports.VocabCol(rng.integers(0, 2, n), ["+", "-"], False),
                "score": reports.VocabCol(rng.integers(0, 4, n), ["NA", "0.5", "1e-05", ""]),
                "samples": reports.Ragged.from_strings([",".join(f"S{j}:1|0" for j in range(int(k))) for k in rng.integers(0, 40, n)]),
                "blank": reports.ConstCol("", n)}
        order = rng.permutation(n)
        want = reports.to_tsv(pd.DataFrame({c: reports._col_array(col)[order] for c, col in cols.items()}))
        path = str(tmp_path / f"r{n}.tsv")
        assert reports.write_report_tsv(path, cols, order) == len(want)
        assert open(path).read() == want
    odd = dict(cols, samples=reports.Ragged.from_strings(['a"b'] * n))
    want = reports.to_tsv(pd.DataFrame({c: reports._col_array(col)[order] for c, col in odd.items()}))
    reports.write_report_tsv(path, odd, order, plain=False)
    assert open(path).read() == want and '"a""b"' in want
    empty = {c: reports.Ragged(np.zeros(0, np.uint8), np.zeros(1, np.uint64)) for c in ("chr", "start")}
    reports.write_report_tsv(path, empty, np.zeros(0, np.int64))
    assert open(path).read() == "chr\tstart\n"


def test_report_order_resolves_ties_like_the_full_lexsort():
    """_report_order sorts on (start, stop) and forms the string keys only for rows that tie on both: same permutation as one
    lexsort over every group column"""
    rng = np.random.default_rng(6)
    n = 4000
    start = rng.integers(0, 300, n)
    stop = start + rng.integers(20, 23, n)
    cols = {"chr": reports.ConstCol("c", n), "start": reports.IntCol(start), "stop": reports.IntCol(stop),
            "sgRNA_sequence": reports.FixedCol(rng.integers(65, 68, (n, 3)).astype(np.uint8)), "pam": reports.FixedCol(rng.integers(65, 67, (n, 2)).astype(np.uint8)),
            "strand": reports.VocabCol(rng.integers(0, 2, n), ["+", "-"], False),
            "score_cfdon": reports.VocabCol(rng.integers(0, 3, n), ["NA", "0.25", "1.0"]),
            "gc_content": reports.VocabCol(rng.integers(0, 3, n), ["0.5", "0.55", "1.0"]),
            "origin": reports.VocabCol(rng.integers(0, 2, n), ["ref", "alt"], False)}
    gcols = ["chr", "start", "stop", "sgRNA_sequence", "pam", "strand", "score_cfdon", "gc_content", "origin"]
    got = reports._report_order(cols, gcols, n)
    keys = []
    for c in reversed(gcols):
        if c == "chr":
            continue
        k = reports._col_array(cols[c])
        keys.append(k if k.dtype.kind in "iuU" else k.astype("U"))
    keys += [stop, start]
    want = np.lexsort(keys)
    # rows that agree in every key may come in either order: compare the keys along the two permutations
    for k in keys:
        assert np.array_equal(k[got], k[want])


def test_variant_window_equals_searchsorted():
    """hawk_host_variant_window (every alt row's window of its haplotype's position-ordered variant list) against numpy's
    searchsorted over (haplotype, position) keys: random lists with empty rows, repeated positions and windows that miss;
    a list out of position order is refused (the caller keeps its numpy route)."""
    import ctypes as C
    from crisprhawk_hip import _lib
    L = reports._host_lib()
    rng = np.random.default_rng(11)
    n_var, n_hap, n = 400, 37, 3000
    t_pos = np.sort(rng.integers(1000, 9000, n_var)).astype(np.int64)
    lists = [np.sort(rng.choice(n_var, size=int(k), replace=False)) if k else np.zeros(0, np.int64) for k in rng.integers(0, 60, n_hap)]
    var_off = np.concatenate(([0], np.cumsum([len(x) for x in lists]))).astype(np.uint64)
    var_idx = np.ascontiguousarray(np.concatenate(lists), dtype=np.int64)
    hap = rng.integers(0, n_hap, n).astype(np.uint32)
    lo = rng.integers(900, 9100, n).astype(np.int64)
    hi = lo + rng.integers(0, 40, n)
    first, count = np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.hawk_host_variant_window(C.c_uint64(n), p(hap), p(lo), p(hi), p(var_off), p(var_idx), p(t_pos), C.c_uint64(n_hap), C.c_uint32(n_var),
                                    p(first), p(count))
    assert rc == _lib.HAWK_OK
    big = np.int64(1) << np.int64(40)
    key = np.repeat(np.arange(n_hap, dtype=np.int64), np.diff(var_off.astype(np.int64))) * big + t_pos[var_idx]
    a = np.searchsorted(key, hap.astype(np.int64) * big + lo, side="left")
    b = np.searchsorted(key, hap.astype(np.int64) * big + hi, side="right")
    assert np.array_equal(first.astype(np.int64)[b > a], a[b > a]) and np.array_equal(count.astype(np.int64), b - a)
    assert (count > 0).sum() > 100 and (count == 0).sum() > 100
    bad = var_idx.copy()
    h = int(np.argmax(np.diff(var_off.astype(np.int64)) > 3))
    s0 = int(var_off[h])
    bad[s0], bad[s0 + 2] = bad[s0 + 2], bad[s0]
    if t_pos[bad[s0]] != t_pos[bad[s0 + 2]]:
        rc = L.hawk_host_variant_window(C.c_uint64(n), p(hap), p(lo), p(hi), p(var_off), p(bad), p(t_pos), C.c_uint64(n_hap), C.c_uint32(n_var),
                                        p(first), p(count))
        assert rc == _lib.HAWK_E_UNSUPPORTED
    bad = var_idx.copy()
    bad[5] = n_var
    rc = L.hawk_host_variant_window(C.c_uint64(n), p(hap), p(lo), p(hi), p(var_off), p(bad), p(t_pos), C.c_uint64(n_hap), C.c_uint32(n_var),
                                    p(first), p(count))
    assert rc == _lib.HAWK_E_INVALID
