"""pysam-free FASTA / BED / VCF readers and the device genotype parser.
CPU tests: file round trips and the record parser against the reference's VariantRecord (g8 fixture).
GPU tests: hawk_gt_parse / hawk_gt_lists against the oracle, and files -> device haplotypes -> guide table equal
to the in-memory path."""
import ctypes as C
import os

import numpy as np
import pytest

from crisprhawk_hip import readers, synth
from crisprhawk_hip.coordinate import Coordinate
from crisprhawk_hip.variant import VariantRecord
from oracle import oracle as ora
from util import load_golden


def _files(tmp_path, reg, width=60, compress=True):
    fa, bed, vcf = str(tmp_path / "r.fa"), str(tmp_path / "r.bed"), str(tmp_path / ("v.vcf.gz" if compress else "v.vcf"))
    readers.write_fasta(fa, reg.contig, reg.contig_seq, width)
    with open(bed, "w") as f:
        f.write(f"# regions\n{reg.contig}\t{reg.bed_start}\t{reg.bed_stop}\tname\t0\t+\n\n")
    readers.write_vcf(vcf, reg.contig, reg.samples, [reg.vcf_fields(v) for v in reg.variants], compress)
    return fa, bed, vcf


@pytest.mark.parametrize("width", [60, 61, 1000])
def test_fasta_bed_roundtrip(tmp_path, width):
    reg = synth.make_region(9101, "chrF", 9000, 2000, 7000)
    fa, bed, _ = _files(tmp_path, synth.add_phased_variants(reg, 9102, 20, 3), width)
    f = readers.Fasta(fa)
    assert os.path.isfile(fa + ".fai") and f.contig == "chrF"
    b = readers.Bed(bed, synth.PADDING)
    assert len(b) == 1 and (b[0].contig, b[0].start, b[0].stop) == ("chrF", reg.startp, reg.stopp)
    assert f.fetch(b[0]).sequence == reg.sequence  # FASTA fetch is [start - 1, stop), sequence.py:340-343
    assert readers.Fasta(fa).fetch_str("chrF", 8990, 9100) == reg.contig_seq[8990:]  # second open reads the .fai; clipped
    regions = b.extract_regions({"chrF": f})
    assert len(regions) == 1 and regions[0].sequence.sequence == reg.sequence
    with pytest.raises(ValueError):
        f.fetch(Coordinate("chrZ", 10, 20, 0))


def test_bed_errors(tmp_path):
    p = tmp_path / "bad.bed"
    p.write_text("chr1\t10\n")
    with pytest.raises(ValueError):
        readers.Bed(str(p), 100)
    p.write_text("chr1\tx\t20\n")
    with pytest.raises(TypeError):
        readers.Bed(str(p), 100)
    p.write_text("chr1\t30\t20\n")
    with pytest.raises(ValueError):
        readers.Bed(str(p), 100)


@pytest.mark.parametrize("compress", [False, True, "bgzf"])
def test_vcf_fetch_matches_in_memory_records(tmp_path, compress):
    reg = synth.make_region(9111, "chrF", 9000, 2000, 7000)
    synth.add_phased_variants(reg, 9112, 80, 6)
    _, _, vcf = _files(tmp_path, reg, compress=compress)
    v = readers.VCF(vcf)
    assert (v.contig, v.phased, v.samples) == ("chrF", True, reg.samples)
    coord = Coordinate("chrF", reg.bed_start, reg.bed_stop, synth.PADDING)
    got = v.fetch(coord)
    want = []
    for s in reg.variants:
        r = VariantRecord(True)
        r.read_vcf_line(reg.vcf_fields(s), reg.samples, True)
        want.append(r)
    assert len(got) == len(want) and all(a == b and a.samples == b.samples and a.afs == b.afs for a, b in zip(got, want))
    # tabix semantics of the range: start < POS <= stop
    inner = Coordinate("chrF", want[10].position, want[20].position, 0)
    assert [r.position for r in v.fetch(inner)] == [r.position for r in want[11:21]]
    blk = v.fetch_block(coord)
    assert len(blk) == len(want)
    for i in (0, len(blk) - 1):
        line = bytes(blk.text[int(blk.line_off[i]):int(blk.line_off[i + 1])]).decode()
        assert line.endswith("\n") and line.rstrip("\n").split("\t") == reg.vcf_fields(reg.variants[i])
        assert bytes(blk.text[int(blk.gt_off[i]):int(blk.line_off[i + 1])]).decode().rstrip("\n").split("\t") == reg.vcf_fields(reg.variants[i])[9:]


def test_record_parser_matches_reference_fixture():
    fx = load_golden("g8_vcf_lines.json.gz")
    samples = fx["samples"]
    codes, flags = ora.vcf_genotype_codes([r["fields"] for r in fx["records"]], len(samples))
    assert not flags.any()
    for i, rec in enumerate(fx["records"]):
        v = VariantRecord(True)
        v.read_vcf_line(rec["fields"], samples, True)
        assert v.alt == rec["alt"] and v.vtype == rec["vtype"] and v.id == rec["ids"] and v.filter == rec["filter"]
        assert [None if a != a else a for a in v.afs] == rec["afs"]
        assert [[sorted(a), sorted(b)] for a, b in v.samples] == rec["samples"]
        assert [[s.position, s.ref, s.alt[0], s.id[0], s.vtype[0]] for s in v.split()] == rec["split"]
        # the allele-code matrix carries the same information as the per-allele sample sets
        for k in range(len(rec["alt"])):
            for c in range(2):
                assert sorted(samples[s] for s in np.flatnonzero(codes[i, c::2] == k + 1)) == rec["samples"][k][c]


# ------------------------------------------------------------------------------------------------ GPU
def _gt_parse(lines, n_samples):
    from crisprhawk_hip import _lib
    from crisprhawk_hip.hapset import _p
    text = np.frombuffer("".join(lines).encode(), dtype=np.uint8)
    line_off = np.zeros(len(lines) + 1, dtype=np.uint64)
    line_off[1:] = np.cumsum([len(x.encode()) for x in lines])
    gt_off = np.zeros(len(lines), dtype=np.uint64)
    for i, ln in enumerate(lines):
        pos = -1
        for _ in range(9):
            pos = ln.index("\t", pos + 1)
        gt_off[i] = int(line_off[i]) + pos + 1
    L, ctx = _lib.lib(), _lib.context(None)
    g, ms = C.c_void_p(), C.c_float()
    _lib.check(L.hawk_gt_parse(ctx, _p(text), C.c_uint64(len(text)), _p(line_off), _p(gt_off), C.c_uint64(len(lines)), n_samples,
                               C.byref(g), C.byref(ms)), "hawk_gt_parse")
    codes = np.zeros((len(lines), 2 * n_samples), dtype=np.uint8)
    flags = np.zeros(len(lines), dtype=np.uint8)
    _lib.check(L.hawk_gt_codes(g, _p(codes), _p(flags)), "hawk_gt_codes")
    return g, codes, flags


@pytest.mark.gpu
def test_device_genotype_codes_against_oracle_and_reference_fixture():
    from crisprhawk_hip import _lib
    fx = load_golden("g8_vcf_lines.json.gz")
    ns = len(fx["samples"])
    recs = [r["fields"] for r in fx["records"]]
    fixed = "chrV\t77\t.\tA\tC,G\t.\tPASS\tAF=0.1,0.2\tGT\t"
    odd = [fixed + "\t".join(["0|1"] * ns),                       # plain
           fixed + "\t".join(["0/1"] * ns),                       # unphased separator -> flag 1
           fixed + "\t".join(["1"] * ns),                         # haploid -> flag 1
           fixed + "\t".join(["0|1"] * (ns - 3)),                 # short record -> flag 2, absent samples stay 255
           fixed + "\t".join(["0|1"] * (ns + 2)),                 # long record -> flag 2, extra fields ignored
           fixed + "\t".join(["0|x"] + ["2|2"] * (ns - 1)),       # garbage -> flag 4
           fixed + "\t".join(["0|1|2"] + ["0|0"] * (ns - 1)),     # triploid -> flag 1
           fixed + "\t".join([".|.", ".", "12|3:9:.", "0|0:1"] + ["254|255"] * (ns - 4))]
    lines = ["\t".join(r) + "\n" for r in recs] + [x + ("\r\n" if i % 2 else "\n") for i, x in enumerate(odd)]
    want, wflags = ora.vcf_genotype_codes([ln.rstrip("\r\n").split("\t") for ln in lines], ns)
    g, codes, flags = _gt_parse(lines, ns)
    try:
        n_ref = len(recs)
        assert np.array_equal(codes[:n_ref], want[:n_ref]) and not flags[:n_ref].any()
        assert flags[n_ref:].tolist() == [0, 1, 1, 2, 2, 4, 1, 1]
        assert (flags[n_ref:] == wflags[n_ref:]).all()
        ok = [0, 1, 3, 4, 7]  # rows whose codes are well defined in both
        assert np.array_equal(codes[n_ref:][ok], want[n_ref:][ok])
        assert codes[n_ref + 2, 0] == 1 and codes[n_ref + 2, 1] == 255       # haploid: second copy missing
        # the reference fixture again, now from the device codes
        for i, rec in enumerate(fx["records"]):
            for k in range(len(rec["alt"])):
                for c in range(2):
                    assert sorted(fx["samples"][s] for s in np.flatnonzero(codes[i, c::2] == k + 1)) == rec["samples"][k][c]
    finally:
        _lib.lib().hawk_gt_destroy(g)


@pytest.mark.gpu
def test_device_carried_lists_against_oracle():
    from crisprhawk_hip import _lib
    from crisprhawk_hip.hapset import _p
    rng = np.random.default_rng(9201)
    ns, nrec = 70, 300
    lines, var_line, var_allele = [], [], []
    for i in range(nrec):
        nalt = 1 + int(rng.integers(0, 3))
        gts = ["|".join(str(int(x)) for x in rng.choice(nalt + 1, 2, p=[0.7] + [0.3 / nalt] * nalt)) for _ in range(ns)]
        lines.append("chrL\t%d\t.\tA\t%s\t.\tPASS\t.\tGT\t" % (100 + 7 * i, ",".join("CGT"[:nalt])) + "\t".join(gts) + "\n")
        for k in range(nalt):
            var_line.append(i); var_allele.append(k + 1)
    var_line, var_allele = np.array(var_line, np.uint32), np.array(var_allele, np.uint8)
    var_r0 = (np.array(var_line, np.int32) * 7 + 10).astype(np.int32)
    var_chain = rng.integers(-3, 4, len(var_line)).astype(np.int32)
    g, codes, flags = _gt_parse(lines, ns)
    try:
        assert not flags.any()
        col_off = np.zeros(2 * ns + 1, dtype=np.uint64)
        delta = np.zeros(2 * ns, dtype=np.int64)
        ms = C.c_float()
        L = _lib.lib()
        _lib.check(L.hawk_gt_lists(g, _p(var_line), _p(var_allele), _p(var_r0), _p(var_chain), len(var_line), _p(col_off), _p(delta),
                                   C.byref(ms)), "hawk_gt_lists")
        idx = np.zeros(int(col_off[-1]), np.uint32); off = np.zeros(int(col_off[-1]), np.int32)
        _lib.check(L.hawk_gt_lists_download(g, _p(idx), _p(off)), "hawk_gt_lists_download")
        w_off, w_idx, w_o, w_delta = ora.carried_lists(codes, var_line, var_allele, var_r0, var_chain)
        assert np.array_equal(col_off, w_off) and np.array_equal(idx, w_idx) and np.array_equal(off, w_o) and np.array_equal(delta, w_delta)
        # the carried indels, as entry indices of those lists (hawk_gt_lists_indels): asked for in two steps
        ni = C.c_uint64(0)
        _lib.check(L.hawk_gt_lists_indels(g, None, C.c_uint64(0), C.byref(ni)), "hawk_gt_lists_indels")
        want_ind = np.flatnonzero(var_chain[w_idx] != 0)
        assert ni.value == len(want_ind) > 0
        ind = np.zeros(ni.value, np.uint32)
        _lib.check(L.hawk_gt_lists_indels(g, _p(ind), C.c_uint64(ni.value), C.byref(ni)), "hawk_gt_lists_indels")
        assert np.array_equal(ind, want_ind)
    finally:
        _lib.lib().hawk_gt_destroy(g)


@pytest.mark.gpu
@pytest.mark.parametrize("compress", [False, True])
def test_files_to_device_haplotypes_equal_in_memory_path(tmp_path, compress):
    from crisprhawk_hip.workload import expand_from_vcf, expand_on_device
    reg = synth.make_region(9301, "chrF", 60_000, 5_000, 55_000)
    synth.add_phased_variants(reg, 9302, 500, 12, af_min=0.05, af_max=0.6)
    fa, bed, vcf = _files(tmp_path, reg, compress=compress)
    coord = readers.Bed(bed, synth.PADDING)[0]
    seq = readers.Fasta(fa).fetch(coord).sequence
    v = readers.VCF(vcf)
    blk = v.fetch_block(coord)
    ds1, info1, ms, kept1, vt = expand_from_vcf(seq, coord.start, coord.stop, blk, v.samples, 3, v.phased)
    ds0, info0, _, kept0 = expand_on_device(reg, 3)
    assert kept0 == kept1 and [i.samples for i in info0] == [i.samples for i in info1]
    assert all(np.array_equal(a.variant_idx, b.variant_idx) for a, b in zip(info0, info1))
    assert vt.id == [f"{reg.contig}-{s.pos}-{s.ref}/{s.alt}" for s in reg.variants]
    assert np.array_equal(ds0.planes(), ds1.planes())
    bits, bitsrc, _, _ = ora.pam_encode("NGG")
    mm, pt = synth.cfd_tables()
    t0 = ds0.search(bits, bitsrc, 3, 20, False, mm, pt)
    t1 = ds1.search(bits, bitsrc, 3, 20, False, mm, pt)
    assert t0.n_rows == t1.n_rows > 0
    for col in ("hap", "pos", "strand", "start", "stop", "flags", "win"):
        assert np.array_equal(getattr(t0, col), getattr(t1, col)), col
    assert np.array_equal(np.nan_to_num(t0.cfdon, nan=-1), np.nan_to_num(t1.cfdon, nan=-1))


@pytest.mark.gpu
def test_multiallelic_sites_expand_like_the_host_builder(tmp_path):
    """Records with several ALT alleles: every chromosome copy carries at most one allele of a site, so the split
    variants (VariantRecord.split, variant.py:313-331) never overlap within a row although they share a position in
    the table.  Device expansion from the record text against the host mirror of solve_haplotypes_phased."""
    from crisprhawk_hip import haplotypes as H
    from crisprhawk_hip.haplotype import Haplotype
    from crisprhawk_hip.region import Region
    from crisprhawk_hip.sequence import Sequence
    from crisprhawk_hip.workload import expand_from_vcf
    rng = np.random.default_rng(9401)
    reg = synth.make_region(9402, "chrM", 9_000, 1_000, 8_000)
    seq, startp = reg.sequence, reg.startp
    samples = [f"S{i}" for i in range(6)]
    rows, pos = [], startp + 150
    others = lambda b: [x for x in "ACGT" if x != b]
    while pos < reg.stopp - 200:
        ref = seq[pos - startp]
        kind = int(rng.integers(0, 4))
        if kind == 0:
            alts = others(ref)[:2]                                   # two SNV alleles
        elif kind == 1:
            alts = [others(ref)[0], ref + "GA"]                      # SNV + insertion at one site
        elif kind == 2:
            alts = [seq[pos - startp: pos - startp + 1], ref + "T"]  # placeholder, fixed below: deletion + insertion
            alts[0] = ref
        else:
            alts = others(ref)                                       # three SNV alleles
        vref = ref
        if kind == 2:  # REF = 3 bases, alleles: deletion to 1 base, and insertion behind the first base
            vref = seq[pos - startp: pos - startp + 3]
            alts = [vref[0], vref[0] + "C" + vref[1:]]
        gts = ["|".join(str(int(rng.integers(0, len(alts) + 1))) for _ in range(2)) for _ in samples]
        rows.append(["chrM", str(pos), ".", vref, ",".join(alts), ".", "PASS", "AF=" + ",".join(["0.1"] * len(alts)), "GT"] + gts)
        pos += int(rng.integers(40, 160))
    vcf = str(tmp_path / "m.vcf")
    readers.write_vcf(vcf, "chrM", samples, rows)
    v = readers.VCF(vcf)
    coord = Coordinate("chrM", reg.bed_start, reg.bed_stop, synth.PADDING)
    blk = v.fetch_block(coord)
    ds, info, _, kept, vt = expand_from_vcf(seq, coord.start, coord.stop, blk, v.samples, 3, v.phased)
    # host mirror: records -> split -> per-sample copies -> collapse
    region = Region(Sequence(seq, True), coord)
    want = H.add_variants_phased([Haplotype(Sequence(seq, True), coord, False, 0, True)], region, v.samples, v.fetch(coord), True, True)
    got_seqs = []
    nib2chr = {}
    from crisprhawk_hip.pam import IUPAC_BITS
    for ch, b in IUPAC_BITS.items():
        nib2chr[b] = ch
    nibbles = {r: ds.nibbles(r) for r in kept}
    vplane = ds.planes()[4]
    for r in kept:
        n = int(ds.hap_len[r])
        bits = ((vplane[r][:, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(-1)[:n].astype(bool)
        s = "".join(nib2chr[int(x)].lower() if low else nib2chr[int(x)] for x, low in zip(nibbles[r][:n], bits))
        got_seqs.append(s)
    assert sorted(got_seqs) == sorted(h.sequence.sequence for h in want)
    by_seq = {h.sequence.sequence: h for h in want}
    for s, inf, r in zip(got_seqs, info, kept):
        h = by_seq[s]
        assert sorted(inf.samples) == sorted(h.samples.split(",")), s[:40]
        assert np.array_equal(ds.host_meta[r].seg.full(), h.segments.full())


def test_vcf_index_streams_and_reads_only_what_it_returns(tmp_path, monkeypatch):
    """The reader keeps an index, not the text: a file spanning many index chunks and many BGZF blocks, with
    an unterminated last line; every fetch must equal the eager parse, and a bgzip fetch must inflate only the blocks
    its records lie in."""
    reg = synth.make_region(9401, "chrB", 260_000, 5_000, 255_000)
    synth.add_phased_variants(reg, 9402, 3000, 40, af_min=0.05, af_max=0.5)
    rows = [reg.vcf_fields(v) for v in reg.variants]
    plain, bg = str(tmp_path / "p.vcf"), str(tmp_path / "b.vcf.gz")
    readers.write_vcf(plain, reg.contig, reg.samples, rows, False)
    data = open(plain, "rb").read()[:-1]  # drop the final newline
    open(plain, "wb").write(data)
    readers.write_bgzf(bg, data, block=4096)
    monkeypatch.setattr(readers._TextSource, "CHUNK", 50_000)  # many chunks: lines straddle chunk borders
    vp, vb = readers.VCF(plain), readers.VCF(bg)
    assert vb._src.kind == "bgzf" and len(vb._src._c_off) > 100 and vp._src.kind == "plain"
    assert np.array_equal(vp._pos, [int(r[1]) for r in rows]) and np.array_equal(vb._pos, vp._pos)
    assert vp.samples == reg.samples and vb.phased and vp.contig == "chrB"
    inflated = []
    orig = readers._TextSource._inflate
    monkeypatch.setattr(readers._TextSource, "_inflate", lambda self, f, k: (inflated.append(k), orig(self, f, k))[1])
    from crisprhawk_hip.coordinate import Coordinate
    for lo, hi in ((5_000, 9_000), (100_000, 101_000), (250_000, 255_000), (1, 260_000)):
        c = Coordinate("chrB", lo, hi, 0)
        want = [r for r in rows if lo < int(r[1]) <= hi]
        inflated.clear()
        for v in (vp, vb):
            got = v.fetch(c)
            assert [(g.position, g.ref, g.alt) for g in got] == [(int(r[1]), r[3], r[4].split(",")) for r in want]
            blk = v.fetch_block(c)
            assert len(blk) == len(want) and (len(want) == 0 or bytes(blk.text[-1:]) == b"\n")
        if hi - lo < 10_000:
            assert 0 < len(set(inflated)) < 0.2 * (len(vb._src._c_off) - 1)


def test_vcf_index_rejects_truncated_and_one_tab_lines(tmp_path):
    """CHROM / POS come from the first 64 bytes of a line - a short body line must not borrow the tabs of the
    line behind it (it is malformed, variant.py would fail on it too), and a header line longer than an index chunk is
    carried across chunks."""
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n"
    good = "chrZ\t100\t.\tA\tG\t.\tPASS\tAF=0.5\tGT\t0|1\n"
    for bad_line in ("chrZ\t200\n", "chrZ\n", "chrZ\t\tx\tA\tG\t.\tPASS\t.\tGT\t0|1\n"):
        f = tmp_path / "bad.vcf"
        f.write_text(head + good + bad_line + good.replace("100", "300"))
        with pytest.raises(ValueError, match="Malformed VCF record"):
            readers.VCF(str(f), 0, True)
    f = tmp_path / "trunc.vcf"
    f.write_text(head + good + "chrZ\t200")  # truncated final record, no newline
    with pytest.raises(ValueError, match="Malformed VCF record"):
        readers.VCF(str(f), 0, True)
    # a #CHROM line longer than the index chunk (thousands of samples): read through the carry path
    names = [f"S{i:05d}" for i in range(3000)]
    long_head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n"
    f = tmp_path / "wide.vcf"
    f.write_text(long_head + "chrZ\t100\t.\tA\tG\t.\tPASS\tAF=0.5\tGT\t" + "\t".join(["0|1"] * 3000) + "\n")
    old = readers._TextSource.CHUNK
    readers._TextSource.CHUNK = 4096
    try:
        v = readers.VCF(str(f), 0, True)
    finally:
        readers._TextSource.CHUNK = old
    assert v.samples == names and v.phased and list(v._pos) == [100]


def test_mapped_vcf_index_equals_the_streamed_one(tmp_path, monkeypatch):
    """A plain-text VCF is indexed by the library's host helper over a mapping of the file (hawk_host_vcf_index: all cores, one
    pass); the numpy stream stays for gzip / bgzip / unterminated files.  Both must build the same index and hand out the same
    record blocks (text, line offsets, sample-column offsets, fixed fields)."""
    reg = synth.make_region(9501, "chrC", 180_000, 4_000, 176_000)
    synth.add_phased_variants(reg, 9502, 2500, 30, af_min=0.05, af_max=0.5)
    rows = [reg.vcf_fields(v) for v in reg.variants]
    plain = str(tmp_path / "p.vcf")
    readers.write_vcf(plain, reg.contig, reg.samples, rows, False)
    fast = readers.VCF(plain)
    assert fast._mm is not None and fast._gt_abs is not None
    monkeypatch.setattr(readers.VCF, "_index_mapped", lambda self: False)
    slow = readers.VCF(plain)
    assert slow._mm is None
    for k in ("_starts", "_ends", "_pos"):
        assert np.array_equal(getattr(fast, k), getattr(slow, k)), k
    assert (fast.samples, fast.contig, fast.phased, fast._total) == (slow.samples, slow.contig, slow.phased, slow._total)
    for lo, hi in ((4_000, 9_000), (100_000, 101_000), (1, 180_000), (50, 60)):
        c = Coordinate("chrC", lo, hi, 0)
        a, b = fast.fetch_block(c), slow.fetch_block(c)
        assert len(a) == len(b) and bytes(a.text) == bytes(b.text)
        assert np.array_equal(a.line_off, b.line_off) and np.array_equal(a.gt_off, b.gt_off) and a.fixed == b.fixed
        assert [(g.position, g.ref, g.alt) for g in fast.fetch(c)] == [(g.position, g.ref, g.alt) for g in slow.fetch(c)]
    # a record without sample columns is refused when its block is asked for, as by the stream
    head = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n"
    f = tmp_path / "nosamples.vcf"
    f.write_text(head + "chrZ\t100\t.\tA\tG\t.\tPASS\tAF=0.5\n")
    monkeypatch.undo()
    v = readers.VCF(str(f), 0, True)
    assert v._mm is not None
    with pytest.raises(ValueError, match="no sample columns"):
        v.fetch_block(Coordinate("chrZ", 1, 1000, 0))
