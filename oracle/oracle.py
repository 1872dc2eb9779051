"""ctypes front-end of the CPU parity oracle (oracle/hawk_oracle.c).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; the product package (crisprhawk_hip) must never import this module.
Parity status: pinned against reference-generated vectors G1-G10 (tests/test_oracle_golden.py),
except the off-target enumeration (CRISPRitz, absent: its targets file in G10 is a brute force of
the call's semantics) and Biopython's Tm_NN behind four Azimuth features ("parity unpinned", see
hawk_oracle.c).
"""

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ERRORS = {
    -1: "IUPAC table error",
    -2: "mismatching reference alleles",
    -3: "position-map key error",
    -4: "CFD table key error",
    -5: "capacity",
    -6: "variant beyond original region length (haplotype.py:199-201 clamp)",
    -7: "duplicate REF guide",
}


class OracleError(RuntimeError):
    def __init__(self, code):
        super().__init__(f"oracle error {code}: {ERRORS.get(code, '?')}")
        self.code = code


def build(force: bool = False) -> str:
    if os.environ.get("HAWK_ORACLE_LIB"):  # e.g. the ASan build (make -C oracle asan-run)
        return os.environ["HAWK_ORACLE_LIB"]
    so = os.path.join(_HERE, "libhawk_oracle.so")
    src = os.path.join(_HERE, "hawk_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libhawk_oracle.so"])
    return so


class _Guide(C.Structure):
    _fields_ = [("start", C.c_int64), ("stop", C.c_int64), ("hap", C.c_int32), ("pos", C.c_int32),
                ("strand", C.c_int32), ("right", C.c_int32), ("emit", C.c_int64)]


class _OT(C.Structure):
    _fields_ = [("guide", C.c_int32), ("strand", C.c_int32), ("pos", C.c_int64), ("mm", C.c_int32), ("pad", C.c_int32)]


GUIDE_DTYPE = np.dtype([("start", "<i8"), ("stop", "<i8"), ("hap", "<i4"), ("pos", "<i4"),
                        ("strand", "<i4"), ("right", "<i4"), ("emit", "<i8")])
OT_DTYPE = np.dtype([("guide", "<i4"), ("strand", "<i4"), ("pos", "<i8"), ("mm", "<i4"), ("pad", "<i4")])


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ora_posmap_rev.restype = C.c_int64
        _LIB.ora_search.restype = C.c_int64
        _LIB.ora_offtargets.restype = C.c_int64
    return _LIB


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def encode(seq: str) -> np.ndarray:
    b = seq.encode("ascii")
    out = np.empty(len(b), dtype=np.uint8)
    bad = C.c_int64(-1)
    rc = lib().ora_encode(b, C.c_int64(len(b)), _p(out), C.byref(bad))
    if rc:
        raise OracleError(rc)
    return out


def revcomp(seq: str) -> str:
    b = seq.encode("ascii")
    out = C.create_string_buffer(len(b))
    rc = lib().ora_revcomp(b, C.c_int64(len(b)), out)
    if rc:
        raise OracleError(rc)
    return out.raw.decode("ascii")


def pam_encode(pam: str) -> Tuple[int, int, str, str]:
    b = pam.encode("ascii")
    bits, bitsrc = C.c_uint64(), C.c_uint64()
    up, rc_ = C.create_string_buffer(len(b)), C.create_string_buffer(len(b))
    rc = lib().ora_pam_encode(b, len(b), C.byref(bits), C.byref(bitsrc), up, rc_)
    if rc:
        raise OracleError(rc)
    return bits.value, bitsrc.value, up.raw.decode(), rc_.raw.decode()


def cas_system(pam_upper: str, right: bool) -> int:
    b = pam_upper.encode("ascii")
    return lib().ora_cas_system(b, len(b), int(right))


def scan(nibbles: np.ndarray, start: int, stop: int, bits: int, bitsrc: int, pamlen: int):
    n = max(0, stop - start)
    fwd = np.empty(n + 1, dtype=np.int32)
    rev = np.empty(n + 1, dtype=np.int32)
    nf, nr = C.c_int64(), C.c_int64()
    lib().ora_scan(_p(nibbles), C.c_int64(start), C.c_int64(stop), C.c_uint64(bits), C.c_uint64(bitsrc),
                   pamlen, _p(fwd), C.byref(nf), _p(rev), C.byref(nr))
    return fwd[: nf.value].copy(), rev[: nr.value].copy()


def hap_build(region_seq: str, startp: int, variants: Sequence[Tuple[int, str, str]]):
    """variants: [(pos, ref, alt)] carried by this chromosome copy. -> (cased sequence, posmap)"""
    n = len(region_seq)
    grow = sum(max(0, len(a) - len(r)) for _, r, a in variants)
    cap = n + grow + 8
    buf = C.create_string_buffer(region_seq.encode("ascii"), cap)
    nv = len(variants)
    vpos = np.array([v[0] for v in variants], dtype=np.int64)
    blob, ro, rl, ao, al = bytearray(), [], [], [], []
    for _, r, a in variants:
        ro.append(len(blob)); rl.append(len(r)); blob += r.encode()
        ao.append(len(blob)); al.append(len(a)); blob += a.encode()
    ro, rl, ao, al = (np.array(x, dtype=np.int32) for x in (ro, rl, ao, al))
    posmap = np.empty(cap, dtype=np.int64)
    out_len = C.c_int64()
    rc = lib().ora_hap_build(buf, C.c_int64(n), C.c_int64(cap), C.c_int64(startp), nv, _p(vpos), _p(ro), _p(rl),
                             _p(ao), _p(al), bytes(blob), C.byref(out_len), _p(posmap))
    if rc:
        raise OracleError(rc)
    L = out_len.value
    return buf.raw[:L].decode("ascii"), posmap[:L].copy()


def posmap_rev(posmap: np.ndarray, g: int) -> int:
    return int(lib().ora_posmap_rev(_p(posmap), C.c_int64(len(posmap)), C.c_int64(g)))


def scan_bounds(posmap: np.ndarray, region_start: int, region_stop: int, pamlen: int,
                hap_start: Optional[int] = None, hap_stop: Optional[int] = None) -> Tuple[int, int]:
    hs = region_start if hap_start is None else hap_start
    he = region_stop if hap_stop is None else hap_stop
    a, b = C.c_int64(), C.c_int64()
    rc = lib().ora_scan_bounds(_p(posmap), C.c_int64(len(posmap)), C.c_int64(region_start), C.c_int64(region_stop),
                               C.c_int64(hs), C.c_int64(he), pamlen, C.byref(a), C.byref(b))
    if rc:
        raise OracleError(rc)
    return a.value, b.value


@dataclass
class HapSet:
    """Haplotypes in the oracle's (= the reference's) representation."""
    seqs: List[str]
    posmaps: List[np.ndarray]
    is_ref: List[bool]
    scan: List[Tuple[int, int]]

    def packed(self):
        off = np.zeros(len(self.seqs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in self.seqs])
        blob = "".join(self.seqs).encode("ascii")
        pm = np.concatenate(self.posmaps) if self.posmaps else np.zeros(0, dtype=np.int64)
        return blob, off, np.ascontiguousarray(pm, dtype=np.int64)


@dataclass
class SearchResult:
    guides: np.ndarray  # GUIDE_DTYPE, reference list order
    win_raw: np.ndarray  # uint8 [n, W] cased ASCII windows
    n_candidates: int
    n_hits: int

    @property
    def windows(self) -> List[str]:
        n, W = self.win_raw.shape
        wb = self.win_raw.tobytes()
        return [wb[i * W:(i + 1) * W].decode("ascii") for i in range(n)]


def search(hs: HapSet, pam: str, guidelen: int, right: bool, cap: Optional[int] = None) -> SearchResult:
    blob, off, pm = hs.packed()
    n_hap = len(hs.seqs)
    if cap is None:
        cap = 2 * int(off[-1]) + 16
    isref = np.array(hs.is_ref, dtype=np.uint8)
    ss = np.array([s[0] for s in hs.scan], dtype=np.int64)
    se = np.array([s[1] for s in hs.scan], dtype=np.int64)
    out = np.zeros(cap, dtype=GUIDE_DTYPE)
    W = guidelen + len(pam) + 20
    win = np.zeros(cap * W, dtype=np.uint8)
    ncand, nhits = C.c_int64(), C.c_int64()
    m = lib().ora_search(blob, _p(off), _p(pm), _p(isref), _p(ss), _p(se), n_hap, pam.encode(), len(pam), guidelen,
                         int(right), _p(out), _p(win), C.c_int64(cap), C.byref(ncand), C.byref(nhits))
    if m < 0:
        raise OracleError(int(m))
    return SearchResult(out[:m].copy(), win[: m * W].reshape(m, W).copy(), ncand.value, nhits.value)


def reverse_and_cfdon(res: SearchResult, is_ref: Sequence[bool], guidelen: int, pamlen: int,
                      mm: Optional[np.ndarray] = None, pamtab: Optional[np.ndarray] = None, decode: bool = True):
    """-> (guides with flipped `right`, reversed windows, kmers, cfdon scores, cfdon order)"""
    n = len(res.guides)
    W = guidelen + pamlen + 20
    g = res.guides.copy()
    win = res.win_raw.reshape(-1).copy()
    kmer = np.zeros(n * (W - 13), dtype=np.uint8)
    cfd = np.full(n, np.nan)
    order = np.zeros(n, dtype=np.int64)
    isref = np.array(is_ref, dtype=np.uint8)
    do_cfd = mm is not None
    mmc = np.ascontiguousarray(mm, dtype=np.float64) if do_cfd else np.zeros(1)
    ptc = np.ascontiguousarray(pamtab, dtype=np.float64) if do_cfd else np.zeros(1)
    rc = lib().ora_reverse_and_cfdon(_p(g), _p(win), C.c_int64(n), guidelen, pamlen, _p(isref), _p(mmc), _p(ptc),
                                     int(do_cfd), _p(cfd), _p(order), _p(kmer))
    if rc:
        raise OracleError(rc)
    if not decode:
        return g, win.reshape(n, W), kmer.reshape(n, W - 13), (cfd if do_cfd else None), (order if do_cfd else None)
    wb, kb = win.tobytes(), kmer.tobytes()
    K = W - 13
    windows = [wb[i * W:(i + 1) * W].decode() for i in range(n)]
    kmers = [kb[i * K:(i + 1) * K].decode() for i in range(n)]
    return g, windows, kmers, (cfd if do_cfd else None), (order if do_cfd else None)


def cfd(wt: str, sg: str, pam: str, mm: np.ndarray, pamtab: np.ndarray) -> float:
    out = C.c_double()
    mmc = np.ascontiguousarray(mm, dtype=np.float64)
    ptc = np.ascontiguousarray(pamtab, dtype=np.float64)
    rc = lib().ora_cfd(wt.encode(), len(wt), sg.encode(), len(sg), pam.encode(), len(pam), _p(mmc), _p(ptc), C.byref(out))
    if rc:
        raise OracleError(rc)
    return out.value


def deepcpf1(seqs: Sequence[str], w: dict) -> np.ndarray:
    n = len(seqs)
    blob = "".join(seqs).encode("ascii")
    assert len(blob) == 34 * n
    out = np.zeros(n, dtype=np.float32)
    arrs = [np.ascontiguousarray(w[k], dtype=np.float32) for k in
            ("conv_w", "conv_b", "w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4")]
    rc = lib().ora_deepcpf1(blob, C.c_int64(n), *[_p(a) for a in arrs], _p(out))
    if rc:
        raise OracleError(rc)
    return out


def offtargets(genome: str, guides: Sequence[str], pam: str, right: bool, max_mm: int, cap: int = 1 << 22) -> np.ndarray:
    """PARITY UNPINNED (no reference implementation; see hawk_oracle.c)."""
    guidelen = len(guides[0])
    out = np.zeros(cap, dtype=OT_DTYPE)
    gb = "".join(guides).encode("ascii")
    n = lib().ora_offtargets(genome.encode("ascii"), C.c_int64(len(genome)), gb, len(guides), guidelen, pam.encode(),
                             len(pam), int(right), max_mm, _p(out), C.c_int64(cap))
    if n < 0:
        raise OracleError(int(n))
    return out[:n].copy()


OTB_DTYPE = np.dtype([("guide", np.int32), ("strand", np.int32), ("pos", np.int64), ("mm", np.int32), ("btype", np.int32), ("bsize", np.int32),
                      ("pad", np.int32), ("gaps", np.uint64)])


def offtargets_bulges(genome: str, guides: Sequence[str], pam: str, right: bool, max_mm: int, bdna: int, brna: int, cap: int = 1 << 22) -> np.ndarray:
    """Bulged off-target sites by brute force over every placement (hawk_oracle.c: ora_offtargets_bulges; PARITY UNPINNED -
    CRISPRitz is absent): rows (guide, strand, pos, mm, btype 1 DNA / 2 RNA, bsize, gaps bitmask in guide orientation)."""
    guidelen = len(guides[0])
    out = np.zeros(cap, dtype=OTB_DTYPE)
    n = lib().ora_offtargets_bulges(genome.encode("ascii"), C.c_int64(len(genome)), "".join(guides).encode("ascii"), len(guides), guidelen,
                                    pam.encode(), len(pam), int(right), max_mm, bdna, brna, _p(out), C.c_int64(cap))
    if n < 0:
        raise OracleError(int(n))
    return out[:n].copy()


def tm_nn(seq: str) -> float:
    """Biopython's Tm_NN with its defaults, restated (hawk_oracle.c: tm_nn)"""
    out = C.c_double()
    rc = lib().ora_tm_nn(seq.encode("ascii"), len(seq), C.byref(out))
    if rc:
        raise OracleError(rc)
    return out.value


def azimuth_features(seqs: Sequence[str]) -> np.ndarray:
    """[n, 627] feature matrix (Tm columns 623..626: parity unpinned, see hawk_oracle.c)."""
    out = np.zeros((len(seqs), 627), dtype=np.float64)
    for i, s in enumerate(seqs):
        assert len(s) == 30
        rc = lib().ora_azimuth_features(s.encode("ascii"), _p(out[i]))
        if rc:
            raise OracleError(rc)
    return out


def gbt_predict(feats: np.ndarray, model: dict) -> np.ndarray:
    n, nf = feats.shape
    out = np.zeros(n, dtype=np.float64)
    f = np.ascontiguousarray(feats, dtype=np.float64)
    a = {k: np.ascontiguousarray(model[k]) for k in ("tree_off", "feature", "left", "right", "threshold", "value")}
    lib().ora_gbt_predict(_p(f), C.c_int64(n), nf, len(a["tree_off"]) - 1, _p(a["tree_off"].astype(np.int32)),
                          _p(a["feature"].astype(np.int32)), _p(a["left"].astype(np.int32)), _p(a["right"].astype(np.int32)),
                          _p(a["threshold"].astype(np.float64)), _p(a["value"].astype(np.float64)), C.c_double(model["init"]),
                          C.c_double(model["learning_rate"]), _p(out))
    return out


def collapse_rows(start, stop, strand, is_ref_row, windows: Sequence[str], guidelen: int, pamlen: int, right: bool):
    """Groups of guide rows the report merges (reports.py:958-1008 `_collapse_report_entries`: groupby over
    chr, start, stop, sgRNA_sequence, pam, strand, scores, gc_content, origin — scores and GC are functions of
    the rest).  Rows are given as search() leaves them (windows on the + strand, 10-nt pads, case preserved);
    equality of the + strand spacer+PAM is equality of the reverse-complemented guide the report prints
    (annotation.py:27-51).  Returns ({key: [row indices ascending]}, {key: (gc_num, gc_den)}) with
    gc_content = gc_num / gc_den of the spacer (annotation.py:513-541 -> Biopython `gc_fraction`, default
    ambiguous="remove": C, G, S over A, C, G, T, S, W, U; parity unpinned - Biopython is absent here)."""
    groups, gc = {}, {}
    for i, w in enumerate(windows):
        core = w[10:-10]
        key = (int(start[i]), int(stop[i]), int(strand[i]), bool(is_ref_row[i]), core)
        groups.setdefault(key, []).append(i)
        if key not in gc:
            pamfirst = bool(right) != bool(strand[i])
            spacer = core[pamlen:] if pamfirst else core[:guidelen]
            num = sum(spacer.count(c) for c in "CGScgs")
            gc[key] = (num, num + sum(spacer.count(c) for c in "ATWUatwu"))
    return groups, gc


def vcf_genotype_codes(records: Sequence[Sequence[str]], n_samples: int):
    """Allele codes of the sample columns of tab-split VCF records, the way _genotypes_to_samples reads them
    (variant.py:558-619): the genotype is the part before the first ':', alleles are separated by '|' (phased) or
    '/'; '0' = REF, '.' = missing, k = k-th ALT.  Returns codes[n, 2*n_samples] (255 = missing / absent) and flags
    (1: a genotype without '|', 2: field count != n_samples, 4: malformed)."""
    n = len(records)
    codes = np.full((n, 2 * n_samples), 255, dtype=np.uint8)
    flags = np.zeros(n, dtype=np.uint8)
    for i, rec in enumerate(records):
        gts = list(rec[9:])
        if len(gts) != n_samples:
            flags[i] |= 2
        for s, gt in enumerate(gts[:n_samples]):
            g = gt.split(":")[0]
            sep = "|" if "|" in g else ("/" if "/" in g else None)
            parts = g.split(sep) if sep else [g]
            if sep != "|" or len(parts) != 2:
                flags[i] |= 1
            for c, a in enumerate(parts[:2]):
                if a == ".":
                    codes[i, 2 * s + c] = 255
                elif a.isdigit():
                    codes[i, 2 * s + c] = min(int(a), 254)
                else:
                    flags[i] |= 4
    return codes, flags


def carried_lists(codes: np.ndarray, var_line, var_allele, var_r0, var_chain):
    """Per chromosome copy (column) the ascending list of carried variants, and per entry r0 + the running sum of
    the length changes of the variants before it (haplotypes.py:132-159 inverted; haplotype.py:185-252 offsets)."""
    n_cols = codes.shape[1]
    col_off = np.zeros(n_cols + 1, dtype=np.uint64)
    idx, off, delta = [], [], np.zeros(n_cols, dtype=np.int64)
    for c in range(n_cols):
        run = 0
        for j in range(len(var_line)):
            if codes[var_line[j], c] == var_allele[j]:
                idx.append(j)
                off.append(int(var_r0[j]) + run)
                run += int(var_chain[j])
        col_off[c + 1] = len(idx)
        delta[c] = run
    return col_off, np.array(idx, dtype=np.uint32), np.array(off, dtype=np.int32), delta
