"""Guide report assembly as a columnar pass over the device guide table.

The reference builds one `Guide` object per (haplotype, hit), annotates each, fills a dict of
lists, and lets pandas group identical rows (`annotation.py:246-370, 513-600`,
`reports.py:80-251, 612-703, 793-1008`).  Here the rows the report merges are grouped on the
device first (`GuideTable.collapse`, hawk_collapse.hip); everything string-valued is then
computed once per *group* from the representative row and the member haplotype ids:

    variant_id   polish_guide_variants on the + strand spacer+PAM and its genomic positions
    af           _format_af of the haplotype's allele frequencies for those variants
    sgRNA/pam    reverse_guides: strand-1 rows are reverse-complemented, case preserved
    gc_content   str(gc_num / gc_den) from the device counts (Biopython gc_fraction, default mode)
    samples      _collapse_samples over the member haplotypes, haplotype_id likewise
    order        pandas groupby order (tuple of the group columns) then a stable sort on (start, stop)

Scores other than CFDon need model files the reference downloads at run time: their columns hold
"NA" unless the caller passes arrays.  The result is a pandas DataFrame with the reference's
column order; `to_tsv` writes what `_store_report` writes.
"""
import os
from collections import defaultdict
from typing import Dict, List, Optional, Sequence

import numpy as np

from .pam import CPF1, PAM, SPCAS9, XCAS9
from .utils import IUPAC, IUPACTABLE, _RC_TRANS, round_score

GUIDESEQPAD = 10
REPORTCOLS = ["chr", "start", "stop", "sgRNA_sequence", "pam", "pam_class", "strand", "score_azimuth", "score_rs3",
              "score_plmcrispr", "score_crispron", "score_sgdesigner", "score_deepcpf1", "score_cfdon", "score_elevationon",
              "gc_content", "origin", "samples", "variant_id", "af", "target", "haplotype_id", "offtargets", "cfd"]


def compute_pam_class(pam: PAM) -> str:  # reports.py:64-77
    return "".join(nt if nt in IUPAC[:4] else f"[{IUPACTABLE[nt]}]" for nt in pam.pam)


def adjust_multiallelic(ref: str, alt: str, pos: int):  # variant.py:456-486
    if len(ref) == len(alt):
        return ref[0], alt[0], pos
    if len(ref) > len(alt):
        return ref[len(alt) - 1:], alt[-1], pos + len(alt) - 1
    return ref[-1], alt[len(ref) - 1:], pos + len(ref) - 1


_PARSED: Dict[str, tuple] = {}  # variant id -> (adjusted position, ref, alt): ids repeat across thousands of haplotypes


def _parse_variant(variant_id: str):  # annotation.py:53-62
    hit = _PARSED.get(variant_id)
    if hit is None:
        hit = _PARSED[variant_id] = _parse_variant_uncached(variant_id)
    return hit


def _parse_variant_uncached(variant_id: str):
    parts = variant_id.split("-")
    ref, alt = parts[2].split("/")
    ref_, alt_, pos_ = adjust_multiallelic(ref, alt, int(parts[1]))
    return pos_, ref_, alt_


def _find_insertion_stop(seg: str) -> int:  # annotation.py:185-189
    assert not seg[0].isupper() and not all(nt.isupper() for nt in seg)
    return next((i for i, nt in enumerate(seg) if nt.isupper()), 0)


def _check_insertion(seg: str, alt: str, posrel: int, pos: int, stop: int, is_snv: bool) -> bool:  # annotation.py:192-226
    if is_snv:
        return False
    if posrel == 0 and alt.endswith(seg.upper()[: _find_insertion_stop(seg)]):
        return True
    return bool(pos == stop and alt.startswith(seg.upper()))


def polish_guide_variants(guidepam: str, posmap: Sequence[int], stop: int, parsed: Dict[int, List]) -> str:
    """annotation.py:246-284 on the + strand spacer+PAM.  `parsed`: adjusted position -> [(id, ref, alt)] of
    the haplotype's variants."""
    polished = set()
    for i in range(len(guidepam)):
        offset = 0
        hit = parsed.get(int(posmap[i]))
        if hit:
            for vid, ref, alt in hit:
                is_snv = len(ref) == len(alt)
                if not is_snv:
                    offset = abs(len(ref) - len(alt)) if len(ref) < len(alt) else 0
                seg = guidepam[i: i + offset + 1]
                if _check_insertion(seg, alt, i, int(posmap[i]), stop, is_snv) or (seg.islower() and seg.upper() == alt):
                    polished.add(vid)
    return ",".join(sorted(polished))


def format_af(af: float) -> str:  # annotation.py:316-330
    s = f"{af:.10f}".rstrip("0").rstrip(".")
    decimal_digits = len(s.split(".")) if "." in s else 0
    return f"{af:.6e}" if decimal_digits > 3 else str(round(af, 6))


def _polish_samples_phased(samples: str) -> str:  # reports.py:767-790
    if "|" not in samples:
        return samples
    smap = defaultdict(lambda: [0, 0])
    for e in samples.split(","):
        sample, gt = e.split(":")
        a1, a2 = map(int, gt.split("|"))
        smap[sample][0] = max(smap[sample][0], a1)
        smap[sample][1] = max(smap[sample][1], a2)
    return ",".join(f"{s}:{a1}|{a2}" for s, (a1, a2) in smap.items())


def collapse_samples(samples: Sequence[str]) -> str:  # reports.py:793-810
    return "" if not len(samples) else _polish_samples_phased(",".join(sorted(set(",".join(samples).split(",")))))


def collapse_haplotype_ids(hapids: Sequence[str]) -> str:  # reports.py:845-857
    return "" if not len(hapids) else ",".join(sorted(set(",".join(hapids).split(","))))


def select_reportcols(pam: PAM, right: bool, estimate_offtargets: bool = False) -> List[str]:
    """Final column order (reports.py:612-660) without the optional annotation / Elevation columns.  With
    --estimate-offtargets: `offtargets` (+ `cfd` for SpCas9 / xCas9 PAMs, reports.py:384-404) between `af` and `target`."""
    cols = (REPORTCOLS[:3] + REPORTCOLS[4:5] + REPORTCOLS[3:4] + REPORTCOLS[5:7]) if right else REPORTCOLS[:7]
    if pam.cas_system in (SPCAS9, XCAS9):
        cols = cols + REPORTCOLS[7:12] + REPORTCOLS[13:14]
    elif pam.cas_system == CPF1:
        cols = cols + REPORTCOLS[12:13]
    cols = cols + REPORTCOLS[15:20]
    if estimate_offtargets:
        cols = cols + REPORTCOLS[22:23] + (REPORTCOLS[23:24] if pam.cas_system in (SPCAS9, XCAS9) else [])
    return cols + REPORTCOLS[20:22]


def _offtarget_columns(spacers: Sequence[str], offtargets: Dict[str, tuple], with_cfd: bool):
    """Per report row: Guide.offtargets / Guide.cfd of its spacer (offtargets.py:597-627 sets both per guide from the
    upper-cased spacer alone, so every row with that spacer carries the same pair; reports.py:292-333)."""
    pairs = [offtargets[sp.upper()] for sp in spacers]
    out = {"offtargets": np.array([str(n) for n, _ in pairs], dtype=object)}
    if with_cfd:
        out["cfd"] = np.array([c for _, c in pairs], dtype=object)
    return out


class _SampleIndex:
    """collapse_samples over many member sets without re-splitting the label strings every time: every distinct
    `sample:genotype` entry gets its rank in string order once; a member set is then a union of small rank arrays, and
    because entries of one sample are neighbours in that order, _polish_samples_phased's per-sample maxima are a
    reduceat."""

    def __init__(self, haplotypes):
        entries = sorted({e for h in haplotypes if h is not None for e in h.samples.split(",")})
        rank = {e: i for i, e in enumerate(entries)}
        self.phased = any("|" in e for e in entries)
        self.entries = entries
        names = [e.split(":")[0] for e in entries]
        uniq: Dict[str, int] = {}
        self.sample_id = np.array([uniq.setdefault(n, len(uniq)) for n in names], dtype=np.int64)
        self.names = list(uniq)
        if self.phased:
            gts = [(e.split(":")[1].split("|") if ":" in e and "|" in e else ["0", "0"]) for e in entries]
            self.a1 = np.array([int(g[0]) for g in gts], dtype=np.int64)
            self.a2 = np.array([int(g[1]) for g in gts], dtype=np.int64)
            self.odd = np.array([":" not in e or "|" not in e for e in entries])  # "REF" / unphased labels among phased ones
        self.ranks = [None if h is None else np.array(sorted({rank[e] for e in h.samples.split(",")}), dtype=np.int64)
                      for h in haplotypes]
        self._fmt: Dict[tuple, str] = {}

    def collapse(self, hap_ids: np.ndarray, haplotypes) -> str:
        if not self.phased:
            return collapse_samples([haplotypes[int(x)].samples for x in hap_ids])
        r = np.unique(np.concatenate([self.ranks[int(x)] for x in hap_ids]))
        if self.odd[r].any():  # e.g. the REF haplotype's group: the plain path
            return collapse_samples([haplotypes[int(x)].samples for x in hap_ids])
        sid = self.sample_id[r]
        starts = np.flatnonzero(np.concatenate(([True], sid[1:] != sid[:-1])))
        m1 = np.maximum.reduceat(self.a1[r], starts)
        m2 = np.maximum.reduceat(self.a2[r], starts)
        names = self.names
        return ",".join([f"{names[i]}:{a}|{b}" for i, a, b in zip(sid[starts].tolist(), m1.tolist(), m2.tolist())])


class ReportInput:
    """The columns the report needs, from a GuideTable (device rows + device groups) or from any other source
    of the same arrays (tests build it from the CPU oracle)."""

    def __init__(self, start, stop, strand, hap, pos, windows: List[str], cfdon, group_perm, group_off, gc_num, gc_den,
                 guidelen: int, pamlen: int, right: bool):
        self.start, self.stop, self.strand, self.hap, self.pos = start, stop, strand, hap, pos
        self.windows, self.cfdon = windows, cfdon
        self.group_perm, self.group_off, self.gc_num, self.gc_den = group_perm, group_off, gc_num, gc_den
        self.guidelen, self.pamlen, self.right = guidelen, pamlen, bool(right)

    @classmethod
    def from_table(cls, tab) -> "ReportInput":
        """tab: a GuideTable on which collapse() ran before download()."""
        tab.download()
        reps = tab.group_perm[tab.group_off[:-1].astype(np.int64)]
        wins = [None] * tab.n_rows  # only representatives are ever decoded
        for r, w in zip(reps.tolist(), tab.windows(reps)):
            wins[r] = w
        return cls(tab.start, tab.stop, tab.strand, tab.hap, tab.pos, wins, tab.cfdon, tab.group_perm, tab.group_off, tab.gc_num,
                   tab.gc_den, tab.guidelen, tab.pamlen, tab.right)


def scorer_kmers(inp: ReportInput):
    """Scorer inputs of the group representatives: `guide.sequence[6:-7].upper()` after reverse_guides
    (scoring.py:50-67) - 30-mers for 20+3 Cas9 guides, 34-mers for 23+4 Cpf1 guides.  Returns (rows, kmers)."""
    perm = np.asarray(inp.group_perm, dtype=np.int64)
    off = np.asarray(inp.group_off, dtype=np.int64)
    rows = perm[off[:-1]]
    out = []
    for r in rows.tolist():
        w = inp.windows[r]
        if int(inp.strand[r]) == 1:
            w = w[::-1].translate(_RC_TRANS)
        out.append(w[6:-7].upper())
    return rows, out


def report_frame(inp: ReportInput, haplotypes, pam: PAM, contig: str, target: str, scores: Optional[Dict[str, np.ndarray]] = None,
                 with_cfdon: bool = True, offtargets: Optional[Dict[str, tuple]] = None):
    """The collapsed, sorted report of one region as a pandas DataFrame (what reports.report_guides hands to
    to_csv).  `haplotypes[i]` needs .samples .variants .afs .id and .segments (PosSegments); `scores` may map a
    score column name to a per-row float array (NaN -> "NA")."""
    import pandas as pd
    cols = select_reportcols(pam, inp.right, offtargets is not None)
    L = inp.guidelen + inp.pamlen
    pamclass = compute_pam_class(pam)
    parsed_cache: Dict[int, Dict[int, List]] = {}
    member_cache: Dict[bytes, tuple] = {}
    sample_index = _SampleIndex(haplotypes)
    recs = []
    perm = np.asarray(inp.group_perm, dtype=np.int64)
    off = np.asarray(inp.group_off, dtype=np.int64)
    score_cols = [c for c in cols if c.startswith("score_")]
    for g in range(len(off) - 1):
        rows = perm[off[g]:off[g + 1]]
        r = int(rows[0])
        h = haplotypes[int(inp.hap[r])]
        s = int(inp.strand[r])
        stored_right = inp.right != bool(s)  # search_guides.py:538
        core = inp.windows[r][GUIDESEQPAD:-GUIDESEQPAD]
        # variants (annotation.py:287-312), before reverse_guides
        if h.samples == "REF" or h.variants in ("NA", ""):
            variant_id, af = "NA", "NA"
        else:
            hi = int(inp.hap[r])
            parsed = parsed_cache.get(hi)
            if parsed is None:
                parsed = defaultdict(list)
                for vid in set(h.variants.split(",")):
                    p_, ref_, alt_ = _parse_variant(vid)
                    parsed[p_].append((vid, ref_, alt_))
                parsed_cache[hi] = parsed
            pivot = int(inp.pos[r]) if stored_right else int(inp.pos[r]) - inp.guidelen
            gen = h.segments.lookup(np.arange(pivot, pivot + L))
            variant_id = polish_guide_variants(core, gen, int(inp.stop[r]), parsed)
            afs = ([format_af(h.afs[v]) if str(h.afs[v]) != "nan" else "NA" for v in variant_id.split(",")]
                   if variant_id != "NA" else ["NA"])
            af = "NA" if not afs or (len(set(afs)) == 1 and afs[0] == "NA") else ",".join(afs)  # guide.py:311-328
        # reverse_guides (annotation.py:27-51)
        if s == 1:
            core = core[::-1].translate(_RC_TRANS)
        if inp.right:
            pamseq, guideseq = core[:inp.pamlen], core[inp.pamlen:]
        else:
            guideseq, pamseq = core[:inp.guidelen], core[inp.guidelen:]
        rec = {"chr": contig, "start": int(inp.start[r]), "stop": int(inp.stop[r]), "sgRNA_sequence": guideseq, "pam": pamseq,
               "pam_class": pamclass, "strand": "+" if s == 0 else "-"}
        for c in score_cols:
            v = float("nan")
            if c == "score_cfdon" and with_cfdon and inp.cfdon is not None:
                v = float(inp.cfdon[r])
            elif scores and c in scores:
                v = float(scores[c][r])
            rec[c] = "NA" if v != v else str(round_score(v))
        den = int(inp.gc_den[g])
        rec["gc_content"] = str(int(inp.gc_num[g]) / den if den else 0.0)
        rec["origin"] = "ref" if h.samples == "REF" else "alt"
        mkey = np.ascontiguousarray(inp.hap[rows]).tobytes()  # guides around one variant share their carriers
        agg = member_cache.get(mkey)
        if agg is None:
            members = [haplotypes[int(x)] for x in inp.hap[rows]]
            agg = (sample_index.collapse(inp.hap[rows], haplotypes), collapse_haplotype_ids([m.id for m in members]))
            member_cache[mkey] = agg
        rec["samples"] = agg[0]
        rec["variant_id"] = ",".join(sorted(set(variant_id.split(",")))) if variant_id else ""  # _check_variant_ids
        rec["af"] = af
        rec["target"] = target
        rec["haplotype_id"] = agg[1]
        if offtargets is not None:
            rec["offtargets"], rec["cfd"] = str(offtargets[guideseq.upper()][0]), offtargets[guideseq.upper()][1]
        recs.append(rec)
    if not recs:
        return pd.DataFrame({c: [] for c in cols})
    # pandas groupby(sort=True) order over the group columns (reports.py:978-1003), then _format_report's
    # sort on (start, stop), which is stable
    gcols = REPORTCOLS[:5]
    if pam.cas_system in (SPCAS9, XCAS9):
        gcols = gcols + REPORTCOLS[6:12] + REPORTCOLS[13:14] + REPORTCOLS[15:17]
    elif pam.cas_system == CPF1:
        gcols = gcols + REPORTCOLS[6:7] + REPORTCOLS[12:13] + REPORTCOLS[15:17]
    else:
        gcols = gcols + REPORTCOLS[6:7] + REPORTCOLS[15:17]
    recs.sort(key=lambda d: tuple(d[c] for c in gcols))
    recs.sort(key=lambda d: (d["start"], d["stop"]))
    return pd.DataFrame({c: [d[c] for d in recs] for c in cols})


def to_tsv(df) -> str:
    """What _store_report writes (reports.py:739: DataFrame.to_csv(sep="\\t", index=False)).  The text is joined column-wise
    here (a C3 report is ~0.7 GB, which pandas' writer takes many seconds over); fields that csv's minimal quoting would
    touch - a quote, a tab, a line break - send the frame through pandas itself."""
    cols = list(df.columns)
    if len(df) == 0:
        return df.to_csv(sep="\t", index=False)
    lists = [df[c].tolist() if df[c].dtype == object else df[c].astype(str).tolist() for c in cols]
    body = "\n".join(["\t".join(r) for r in zip(*lists)])
    if '"' in body or "\r" in body or body.count("\n") != len(df) - 1 or body.count("\t") != (len(cols) - 1) * len(df):
        return df.to_csv(sep="\t", index=False)
    return "\t".join(cols) + "\n" + body + "\n"


def report_filename(contig: str, bed_start: int, bed_stop: int, pam: PAM, guidelen: int) -> str:
    """reports.py:1042-1047: crisprhawk_guides__{contig}_{BEDstart}_{BEDstop}_{PAM}_{guidelen}.tsv"""
    return f"crisprhawk_guides__{contig}_{bed_start}_{bed_stop}_{pam}_{guidelen}.tsv"


def report_table(tab, haplotypes, pam: PAM, contig: str, bed_start: int, bed_stop: int, outdir: str, scores=None,
                 with_cfdon: bool = True) -> str:
    """report_guides for one region from a collapsed GuideTable: assemble, write the TSV, return its path."""
    import os
    df = report_frame(ReportInput.from_table(tab), haplotypes, pam, contig, f"{contig}:{bed_start}-{bed_stop}", scores, with_cfdon)
    path = os.path.join(outdir, report_filename(contig, bed_start, bed_stop, pam, tab.guidelen))
    df.to_csv(path, sep="\t", index=False)
    return path


# ---------------------------------------------------------------------------------------------------
# Columnar assembly from report groups (the fast path: nothing is computed per group in Python
# except for rows whose variants include indels)
# ---------------------------------------------------------------------------------------------------
_RC_LUT = np.arange(256, dtype=np.uint8)
for _a, _b in _RC_TRANS.items():
    _RC_LUT[_a] = _b
_CODE2CHAR = np.frombuffer(b"?ACMGRSVTWYHKDBN?acmgrsvtwyhkdbn", dtype=np.uint8)


class ReportGroups:
    """What report_from_groups needs: one representative row per group as columns + CSR member haplotypes.
    hapset.GroupTable (device export) and tiling.MergedGroups provide it; `from_report_input` builds it from
    row-level inputs (tests: oracle rows)."""

    def __init__(self, guidelen, pamlen, right, pos, strand, start, stop, cfdon, win, gc_num, gc_den, member_off, member_hap):
        self.guidelen, self.pamlen, self.right = guidelen, pamlen, bool(right)
        self.pos, self.strand, self.start, self.stop, self.cfdon = pos, strand, start, stop, cfdon
        self.win, self.gc_num, self.gc_den = win, gc_num, gc_den  # win: [5, n_groups] uint64 window slices
        self.member_off = np.asarray(member_off, dtype=np.int64)
        self.member_hap = np.asarray(member_hap, dtype=np.int64)
        self.n_groups = len(self.member_off) - 1

    @classmethod
    def from_report_input(cls, inp: "ReportInput") -> "ReportGroups":
        perm = np.asarray(inp.group_perm, dtype=np.int64)
        off = np.asarray(inp.group_off, dtype=np.int64)
        reps = perm[off[:-1]]
        W = inp.guidelen + inp.pamlen + 2 * GUIDESEQPAD
        win = np.zeros((5, len(reps)), dtype=np.uint64)
        lut = np.zeros(256, dtype=np.uint8)
        for code, ch in enumerate(_CODE2CHAR.tobytes()):
            if ch != ord("?"):
                lut[ch] = code
        if len(reps):
            mat = np.frombuffer("".join(inp.windows[int(r)] for r in reps).encode("ascii"), dtype=np.uint8).reshape(len(reps), W)
            codes = lut[mat].astype(np.uint64)
            sh = np.arange(W, dtype=np.uint64)
            for p in range(5):
                win[p] = (((codes >> np.uint64(p)) & np.uint64(1)) << sh).sum(axis=1, dtype=np.uint64)
        cfd = None if inp.cfdon is None else np.asarray(inp.cfdon)[reps]
        return cls(inp.guidelen, inp.pamlen, inp.right, np.asarray(inp.pos)[reps], np.asarray(inp.strand)[reps], np.asarray(inp.start)[reps],
                   np.asarray(inp.stop)[reps], cfd, win, np.asarray(inp.gc_num), np.asarray(inp.gc_den), off, np.asarray(inp.hap)[perm])


class Ragged:
    """A column of byte strings as ONE blob + offsets (row g = blob[off[g]:off[g + 1]]) - how the report's text columns are kept
    until they are written (write_report_tsv hands blob and offsets to the library's TSV writer; `strings()` makes the Python
    strings the DataFrame route wants)."""

    __slots__ = ("blob", "off")

    def __init__(self, blob: np.ndarray, off: np.ndarray):
        self.blob, self.off = np.ascontiguousarray(blob, dtype=np.uint8), np.ascontiguousarray(off, dtype=np.uint64)

    def __len__(self):
        return len(self.off) - 1

    @classmethod
    def from_strings(cls, strings: Sequence[str]) -> "Ragged":
        return cls(*_pool(strings))

    def strings(self) -> List[str]:
        return _decode_groups(self.blob, self.off)

    def take(self, rows) -> List[str]:
        buf, o = memoryview(self.blob), self.off
        return [str(buf[int(o[g]):int(o[g + 1])], "ascii") for g in np.asarray(rows).tolist()]

    def select(self, labels: np.ndarray) -> "Ragged":
        """row g of the result = row labels[g] of this column (a gather of byte strings: hawk_host_ragged_join, one item per row)"""
        labels = np.ascontiguousarray(labels, dtype=np.uint32)
        n = len(labels)
        return _ragged_join_raw(labels, np.arange(n + 1, dtype=np.uint64), self.blob, self.off)

    def extended(self, more: Sequence[str]) -> "Ragged":
        """this column followed by the rows `more`"""
        if not more:
            return self
        b2, o2 = _pool(more)
        return Ragged(np.concatenate([self.blob, b2]), np.concatenate([self.off, o2[1:] + self.off[-1]]))


def _big_bytes(n: int) -> np.ndarray:
    """n writable bytes.  The carriers' columns of a big report are hundreds of MB that are written once, by many threads: from a
    fresh allocation in 4 KB pages that is one page fault per 4 KB (C3: 0.05 s of a 0.1 s join) - so large blobs come from an anonymous
    mapping advised for transparent huge pages (where the kernel has them; a plain mapping otherwise)."""
    n = max(int(n), 1)
    if n < (32 << 20):
        return np.empty(n, dtype=np.uint8)
    import mmap
    mm = mmap.mmap(-1, (n + (2 << 20) - 1) & ~((2 << 20) - 1), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    try:
        mm.madvise(mmap.MADV_HUGEPAGE)
    except (AttributeError, OSError, ValueError):
        pass
    return np.frombuffer(mm, dtype=np.uint8, count=n)


def _ragged_join_raw(item_label: np.ndarray, group_off: np.ndarray, pool: np.ndarray, pool_off: np.ndarray) -> Ragged:
    """Per group: the pool strings its items name, joined by commas (hawk_host_ragged_join), as a Ragged column."""
    import ctypes as C
    from . import _lib
    ng = len(group_off) - 1
    if ng <= 0:
        return Ragged(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    pool_off = np.ascontiguousarray(pool_off, dtype=np.uint64)
    item = np.ascontiguousarray(item_label, dtype=np.uint32)
    goff = np.ascontiguousarray(group_off, dtype=np.uint64)
    out_off = np.zeros(ng + 1, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L = _host_lib()
    n_lab = C.c_uint64(len(pool_off) - 1)
    _lib.check(L.hawk_host_ragged_join(p(item), p(goff), C.c_uint64(ng), p(pool), p(pool_off), n_lab, C.c_uint8(44), None,
                                       C.c_uint64(0), p(out_off)), "hawk_host_ragged_join")
    out = _big_bytes(int(out_off[-1]))
    _lib.check(L.hawk_host_ragged_join(p(item), p(goff), C.c_uint64(ng), p(pool), p(pool_off), n_lab, C.c_uint8(44), p(out),
                                       C.c_uint64(len(out)), p(out_off)), "hawk_host_ragged_join")
    return Ragged(out, out_off)


def _ragged_join(item_label: np.ndarray, group_off: np.ndarray, strings: Sequence[str]) -> List[str]:
    """Per group: ",".join(strings[l] for its items) through the library's host helper (hawk_host_ragged_join)."""
    if len(group_off) - 1 <= 0:
        return []
    return _ragged_join_raw(item_label, group_off, *_pool(strings)).strings()


def _host_lib():
    """The library holding the host-side join helpers: libhawk_hip.so, or - for the sanitizer run of csrc/Makefile's
    asan-host target - a host-only build of hawk_hostutil named by HAWK_HOSTUTIL_LIB."""
    import ctypes as C
    alt = os.environ.get("HAWK_HOSTUTIL_LIB")
    if alt:
        return C.CDLL(alt)
    from . import _lib
    return _lib.lib()


def _pool(strings: Sequence[str]):
    enc = [x.encode("ascii") for x in strings]
    pool = np.frombuffer(b"".join(enc), dtype=np.uint8) if enc else np.zeros(0, np.uint8)
    off = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        off[1:] = np.cumsum([len(e) for e in enc])
    return np.ascontiguousarray(pool), off


def _decode_groups(out: np.ndarray, out_off: np.ndarray) -> List[str]:
    buf = memoryview(out)
    o = out_off.tolist()
    return [str(buf[o[g]:o[g + 1]], "ascii") for g in range(len(o) - 1)]


def _hap_items(per_hap: List[List[str]]):
    """Per-haplotype item lists -> (vocabulary in sorted order, CSR offsets, item ranks ascending per haplotype)."""
    vocab = sorted({x for items in per_hap for x in items})
    rank = {x: i for i, x in enumerate(vocab)}
    off = np.zeros(len(per_hap) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(x) for x in per_hap])
    flat = np.array([r for items in per_hap for r in sorted(rank[x] for x in items)], dtype=np.uint32)
    return vocab, off, flat


def _group_join_raw(member_off, member_hap, per_hap: List[List[str]]) -> Ragged:
    """Per group ",".join(sorted(set(items of its member haplotypes))) - hawk_host_group_join - as a Ragged column."""
    import ctypes as C
    from . import _lib
    ng = len(member_off) - 1
    if ng <= 0:
        return Ragged(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    vocab, hoff, flat = _hap_items(per_hap)
    pool, poff = _pool(vocab)
    moff = np.ascontiguousarray(member_off, dtype=np.uint64)
    mh = np.ascontiguousarray(member_hap, dtype=np.uint32)
    out_off = np.zeros(ng + 1, dtype=np.uint64)
    p = lambda a_: a_.ctypes.data_as(C.c_void_p)
    L = _host_lib()
    args = (p(moff), p(mh), C.c_uint64(ng), p(hoff), p(flat), C.c_uint64(len(per_hap)), p(pool), p(poff), C.c_uint64(len(vocab)), C.c_uint8(44))
    _lib.check(L.hawk_host_group_join(*args, None, C.c_uint64(0), p(out_off)), "hawk_host_group_join")
    out = _big_bytes(int(out_off[-1]))
    _lib.check(L.hawk_host_group_join(*args, p(out), C.c_uint64(len(out)), p(out_off)), "hawk_host_group_join")
    return Ragged(out, out_off)


def _group_join(member_off, member_hap, per_hap: List[List[str]]) -> List[str]:
    return _group_join_raw(member_off, member_hap, per_hap).strings() if len(member_off) > 1 else []


def _samples_raw(member_off, member_hap, hap_samples: List[str]) -> Ragged:
    """reports.py:767-810 for every group: sorted unique `sample:genotype` entries, phased genotypes OR-ed per sample
    (hawk_host_group_samples).  `hap_samples[h]`: the samples label of haplotype row h ("" for rows that hold none)."""
    import ctypes as C
    from . import _lib
    per_hap = [x.split(",") if x else [] for x in hap_samples]
    ng = len(member_off) - 1
    if ng <= 0:
        return Ragged(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    vocab, hoff, flat = _hap_items(per_hap)
    if not any("|" in e for e in vocab):
        return _group_join_raw(member_off, member_hap, per_hap)
    n_e = len(vocab)
    ok = np.zeros(n_e, dtype=np.uint8)
    a1, a2 = np.zeros(n_e, np.uint16), np.zeros(n_e, np.uint16)
    names = []
    for i, e in enumerate(vocab):
        nm = e
        if e.count(":") == 1 and "|" in e:
            sname, gt = e.split(":")
            xy = gt.split("|")
            if len(xy) == 2 and xy[0].isdigit() and xy[1].isdigit() and int(xy[0]) < 65536 and int(xy[1]) < 65536:
                nm, a1[i], a2[i], ok[i] = sname, int(xy[0]), int(xy[1]), 1
        names.append(nm)
    uniq: Dict[str, int] = {}  # samples in the order of their first entry in the sorted entry list
    sid = np.array([uniq.setdefault(n, len(uniq)) for n in names], dtype=np.uint32)
    npool, noff = _pool(list(uniq))
    epool, eoff = _pool(vocab)
    moff = np.ascontiguousarray(member_off, dtype=np.uint64)
    mh = np.ascontiguousarray(member_hap, dtype=np.uint32)
    out_off = np.zeros(ng + 1, dtype=np.uint64)
    flags = np.zeros(ng, dtype=np.uint8)
    p = lambda a_: a_.ctypes.data_as(C.c_void_p)
    L = _host_lib()
    args = (p(moff), p(mh), C.c_uint64(ng), p(hoff), p(flat), C.c_uint64(len(per_hap)), p(sid), p(a1), p(a2), p(ok), C.c_uint64(n_e),
            p(npool), p(noff), C.c_uint64(len(uniq)), p(epool), p(eoff))
    _lib.check(L.hawk_host_group_samples(*args, None, C.c_uint64(0), p(out_off), p(flags)), "hawk_host_group_samples")
    out = _big_bytes(int(out_off[-1]))
    _lib.check(L.hawk_host_group_samples(*args, p(out), C.c_uint64(len(out)), p(out_off), p(flags)), "hawk_host_group_samples")
    col = Ragged(out, out_off)
    odd = np.flatnonzero(flags == 3)  # a group mixing phased entries with others: the reference's own per-group walk
    if len(odd):
        fixed = [collapse_samples([hap_samples[int(x)] for x in member_hap[member_off[g]:member_off[g + 1]]]) for g in odd.tolist()]
        labels = np.arange(ng, dtype=np.uint32)
        labels[odd] = ng + np.arange(len(odd), dtype=np.uint32)
        col = col.extended(fixed).select(labels)
    return col


def _samples_column(member_off, member_hap, hap_samples: List[str]) -> List[str]:
    return _samples_raw(member_off, member_hap, hap_samples).strings() if len(member_off) > 1 else []




def _hapids_raw(member_off, member_hap, hap_ids: List[str]) -> Ragged:
    return _group_join_raw(member_off, member_hap, [x.split(",") if x else [] for x in hap_ids])


def _hapids_column(member_off, member_hap, hap_ids: List[str]) -> List[str]:
    return _hapids_raw(member_off, member_hap, hap_ids).strings() if len(member_off) > 1 else []


class HapLabels:
    """What the report needs to know about the haplotype rows, as columns: per row its `samples` label, its id, whether
    it is REF, its position map, and the variants it carries as indices into ONE variant table (ids + allele
    frequencies) - instead of one object per row holding a 1000-id string and a 1000-entry AF dict."""

    def __init__(self, samples: List[str], ids: List[str], is_ref: np.ndarray, var_off: np.ndarray, var_idx: np.ndarray,
                 vid: List[str], af: np.ndarray, segments: list):
        self.samples, self.ids, self.is_ref = samples, ids, np.asarray(is_ref, dtype=bool)
        self.var_off, self.var_idx = np.asarray(var_off, dtype=np.int64), np.asarray(var_idx, dtype=np.int64)
        self.vid, self.af, self.segments = vid, np.asarray(af, dtype=np.float64), segments
        self._vt = None
        self.seg_csr = None  # optional (seg_start[H + 1], seg_rel, seg_gen): every row's position map as flat arrays

    def __len__(self):
        return len(self.samples)

    @classmethod
    def from_objects(cls, haplotypes) -> "HapLabels":
        """From per-row label objects (.samples .variants .afs .id .segments; None for rows collapsed onto another)."""
        vocab: Dict[str, int] = {}
        vid: List[str] = []
        af: List[float] = []
        off, idx, samples, ids, is_ref, segs = [0], [], [], [], [], []
        for h in haplotypes:
            if h is None:
                samples.append(""); ids.append(""); is_ref.append(False); segs.append(None)
                off.append(len(idx))
                continue
            samples.append(h.samples); ids.append(h.id); is_ref.append(h.samples == "REF"); segs.append(h.segments)
            if h.samples != "REF" and h.variants not in ("NA", ""):
                # (in position order, as the device path lists them: the library's candidate search is a binary search per row)
                for v in sorted(set(h.variants.split(",")), key=lambda x: (_parse_variant(x)[0], x)):
                    k = vocab.get(v)
                    if k is None:
                        k = vocab[v] = len(vid)
                        vid.append(v)
                        af.append(float(h.afs[v]))
                    idx.append(k)
            off.append(len(idx))
        lab = cls(samples, ids, np.array(is_ref, dtype=bool), np.array(off), np.array(idx, dtype=np.int64), vid, np.array(af), segs)
        if all(x is None or hasattr(x, "rel") for x in segs):  # the rows' position maps as flat arrays (rows without one: a single identity segment)
            rel = [np.zeros(1, np.uint32) if x is None else np.asarray(x.rel, dtype=np.uint32) for x in segs]
            gen = [np.zeros(1, np.int64) if x is None else np.asarray(x.gen, dtype=np.int64) for x in segs]
            start = np.concatenate(([0], np.cumsum([len(r) for r in rel]))).astype(np.uint64)
            lab.seg_csr = (start, np.ascontiguousarray(np.concatenate(rel)) if rel else np.zeros(0, np.uint32),
                           np.ascontiguousarray(np.concatenate(gen)) if gen else np.zeros(0, np.int64))
        return lab

    def variant_table(self):
        """(adjusted position, is SNV, SNV alt base, ref, alt, formatted AF) per variant of `vid` (annotation.py:53-62)."""
        if self._vt is None:
            parsed = [_parse_variant(v) for v in self.vid]
            pos = np.array([p[0] for p in parsed], dtype=np.int64) if parsed else np.zeros(0, np.int64)
            snv = np.array([len(p[1]) == len(p[2]) for p in parsed], dtype=bool) if parsed else np.zeros(0, bool)
            alt0 = np.array([ord(p[2][0]) if (len(p[1]) == len(p[2]) == 1) else 0 for p in parsed], dtype=np.uint8) if parsed else np.zeros(0, np.uint8)
            uniq, inv = np.unique(self.af, return_inverse=True)  # (one NaN entry at most: numpy sorts NaN last and merges them)
            fmt = np.array([format_af(x) if x == x else "NA" for x in uniq.tolist()], dtype=object)
            afs = fmt[inv.reshape(-1)].tolist() if len(self.af) else []
            self._vt = (pos, snv, alt0, [p[1] for p in parsed], [p[2] for p in parsed], afs)
        return self._vt


def _variant_columns_raw(G, rep_hap: np.ndarray, cores: np.ndarray, lab: HapLabels):
    """variant_id / af of every group from its representative (annotation.py:246-370), as two Ragged columns.  Rows whose
    haplotype carries only SNVs around the guide are resolved in bulk; rows with an indel among the candidates go through
    polish_guide_variants exactly as report_frame does."""
    ng, L = G.n_groups, G.guidelen + G.pamlen
    na = Ragged.from_strings(["NA"])
    all_na = lambda: na.select(np.zeros(ng, dtype=np.uint32))
    if len(lab.var_idx) == 0:
        return all_na(), all_na()
    t_pos, t_snv, t_alt0, t_ref, t_alt, t_af = lab.variant_table()
    H = len(lab)
    cnt_h = np.diff(lab.var_off)
    if lab.seg_csr is not None:  # candidates + the reference's walk for every alt row inside the library
        done = _variant_columns_native(G, rep_hap, cores, lab)
        if done is not None:
            return done
    v_hap = np.repeat(np.arange(H, dtype=np.int64), cnt_h)
    v_var = lab.var_idx
    BIG = np.int64(1) << np.int64(40)
    vkey = v_hap * BIG + t_pos[v_var]
    if len(vkey) > 1 and not (vkey[1:] >= vkey[:-1]).all():  # rows list their variants in position order as a rule
        o = np.argsort(vkey, kind="stable")
        vkey, v_var = vkey[o], v_var[o]
    alt_rows = np.flatnonzero(~lab.is_ref[rep_hap] & (cnt_h[rep_hap] > 0))
    if len(alt_rows) == 0:
        return all_na(), all_na()
    hh = rep_hap[alt_rows]
    start, stop = np.asarray(G.start, np.int64)[alt_rows], np.asarray(G.stop, np.int64)[alt_rows]
    a = np.searchsorted(vkey, hh * BIG + start, side="left")
    b = np.searchsorted(vkey, hh * BIG + np.maximum(stop, start + L), side="right")
    cnt = b - a
    pair_row = np.repeat(np.arange(len(alt_rows)), cnt)
    pair_var = v_var[np.repeat(a, cnt) + (np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt))]
    has_indel = np.zeros(len(alt_rows), dtype=bool)
    np.logical_or.at(has_indel, pair_row, ~t_snv[pair_var])
    fast = ~has_indel & ((stop - start) == L)
    # ---- SNV-only rows: base i = pos - start of the + strand core must be the alt allele in lower case
    fp = fast[pair_row]
    pr, pv = pair_row[fp], pair_var[fp]
    i = t_pos[pv] - start[pr]
    inside = (i >= 0) & (i < L)
    pr, pv, i = pr[inside], pv[inside], i[inside]
    ch = cores[alt_rows[pr], i]
    ok = (ch == (t_alt0[pv] | 0x20)) & (t_alt0[pv] != 0)
    pr, pv = pr[ok], pv[ok]
    # ids of one row in string order (annotation.py:284: sorted), duplicates dropped
    name_rank = np.argsort(np.argsort(np.array(lab.vid))) if lab.vid else np.zeros(0, np.int64)
    order = np.lexsort((name_rank[pv], pr))
    pr, pv = pr[order], pv[order]
    keep = np.ones(len(pr), dtype=bool)
    keep[1:] = (pr[1:] != pr[:-1]) | (pv[1:] != pv[:-1])
    pr, pv = pr[keep], pv[keep]
    roff = np.concatenate(([0], np.cumsum(np.bincount(pr, minlength=len(alt_rows)))))
    n_alt = len(alt_rows)
    ids = _ragged_join_raw(pv, roff, *_pool(lab.vid))
    af_names = sorted(set(t_af))
    af_rank = {s_: k for k, s_ in enumerate(af_names)}
    af_of_var = np.array([af_rank[x] for x in t_af], dtype=np.int64)
    afs = _ragged_join_raw(af_of_var[pv], roff, *_pool(af_names))  # allele frequencies follow the ids' order (guide.py:311-328)
    na_id = af_rank.get("NA", -1)
    some_af = np.zeros(n_alt, dtype=bool)
    np.logical_or.at(some_af, pr, af_of_var[pv] != na_id)
    # per group: which row of (ids | afs, then "NA", then the slow rows' strings) it shows
    NA = n_alt
    vid_lab = np.full(ng, NA, dtype=np.uint32)
    af_lab = np.full(ng, NA, dtype=np.uint32)
    fk = np.flatnonzero(fast)
    vid_lab[alt_rows[fk]] = fk
    has_id = np.diff(ids.off.astype(np.int64))[fk] > 0
    af_ok = some_af[fk] & has_id
    af_lab[alt_rows[fk[af_ok]]] = fk[af_ok]
    slow_vid: List[str] = []
    slow_af: List[str] = []
    # ---- rows with an indel among the candidates (or a non-linear position map): the reference's own walk over the
    # candidates (variants elsewhere on the haplotype cannot match a position of this guide) - by the library's helper for all
    # such rows at once when the rows' position maps are at hand as flat arrays, row by row in Python otherwise
    slow = np.flatnonzero(~fast)
    polished = None
    if len(slow) and lab.seg_csr is not None:
        polished = _polish_rows_native(G, lab, alt_rows, hh, slow, a, cnt, v_var, cores, t_pos, t_ref, t_alt, name_rank)
    if polished is not None:
        p_off, p_var, need_py = polished
        # ids in string order; allele frequencies in the ids' order, "NA" when the row shows nothing or every frequency is missing
        s_ids = _ragged_join_raw(p_var, p_off, *_pool(lab.vid))
        s_afs = _ragged_join_raw(af_of_var[p_var], p_off, *_pool(af_names))
        n_shown = np.diff(p_off.astype(np.int64))
        has_af = np.zeros(len(slow), dtype=bool)
        if len(p_var):
            row_of = np.repeat(np.arange(len(slow)), n_shown)
            np.logical_or.at(has_af, row_of, af_of_var[p_var] != na_id)
        base_v = NA + 1  # rows of the extended pools: ids / afs of the fast rows, "NA", then the slow rows' strings
        gsl = alt_rows[slow]
        vid_lab[gsl] = base_v + np.arange(len(slow))
        af_lab[gsl] = np.where(has_af, base_v + np.arange(len(slow)), NA)
        slow_vid, slow_af = s_ids.strings(), s_afs.strings()
        slow = slow[need_py != 0]  # (rows on which the reference's own assertion fires: its Python mirror raises below)
    if len(slow):
        s_off = np.concatenate(([0], np.cumsum(cnt)))
        pvl = pair_var.tolist()
        for k in slow.tolist():
            g = int(alt_rows[k])
            v_s, a_s = _polish_row_python(G, lab, g, int(hh[k]), set(pvl[s_off[k]:s_off[k + 1]]), cores)
            vid_lab[g] = af_lab[g] = NA + 1 + len(slow_vid)
            slow_vid.append(v_s)
            slow_af.append(a_s)
    return ids.extended(["NA"] + slow_vid).select(vid_lab), afs.extended(["NA"] + slow_af).select(af_lab)


def _polish_row_python(G, lab: "HapLabels", g: int, h: int, cand, cores):
    """(variant_id, af) of group g on haplotype row h by the Python mirror of annotation.polish_guide_variants (it raises where the
    reference's own assertion fires)"""
    t_pos, _, _, t_ref, t_alt, t_af = lab.variant_table()
    L = G.guidelen + G.pamlen
    parsed = defaultdict(list)
    for v in cand:
        parsed[int(t_pos[v])].append((lab.vid[v], t_ref[v], t_alt[v]))
    s = int(G.strand[g])
    stored_right = G.right != bool(s)
    pivot = int(G.pos[g]) if stored_right else int(G.pos[g]) - G.guidelen
    gen = lab.segments[h].lookup(np.arange(pivot, pivot + L))
    core = cores[g].tobytes().decode("ascii")
    variant_id = polish_guide_variants(core, gen, int(G.stop[g]), parsed)
    vids = variant_id.split(",")
    afl = [t_af[_vid_index(lab, v)] for v in vids] if variant_id else []
    return (",".join(sorted(set(vids))) if variant_id else "",
            "NA" if not afl or (len(set(afl)) == 1 and afl[0] == "NA") else ",".join(afl))


def _variant_columns_native(G, rep_hap: np.ndarray, cores: np.ndarray, lab: "HapLabels"):
    """_variant_columns_raw with the candidate search (annotation.py:246-284) and the walk over the candidates done by the library for
    every alt row (hawk_host_variant_window + hawk_host_polish_windows): nothing per candidate exists on the Python side.  None when the
    library cannot take the job (an older build; rows that do not list their variants in position order)."""
    import ctypes as C
    from . import _lib
    try:
        f_win, f_pol = _host_lib().hawk_host_variant_window, _host_lib().hawk_host_polish_windows
    except AttributeError:
        return None
    ng, L = G.n_groups, G.guidelen + G.pamlen
    t_pos, _, _, t_ref, t_alt, t_af = lab.variant_table()
    cnt_h = np.diff(lab.var_off)
    na = Ragged.from_strings(["NA"])
    alt_rows = np.flatnonzero(~lab.is_ref[rep_hap] & (cnt_h[rep_hap] > 0))
    n = len(alt_rows)
    if n == 0:
        z = np.zeros(ng, dtype=np.uint32)
        return na.select(z), na.select(z)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    hp = np.ascontiguousarray(rep_hap[alt_rows], dtype=np.uint32)
    start = np.ascontiguousarray(np.asarray(G.start, np.int64)[alt_rows])
    stop = np.ascontiguousarray(np.asarray(G.stop, np.int64)[alt_rows])
    p_hi = np.maximum(stop, start + L)
    var_off = np.ascontiguousarray(lab.var_off, dtype=np.uint64)
    var_idx = np.ascontiguousarray(lab.var_idx, dtype=np.int64)
    tp = np.ascontiguousarray(t_pos, dtype=np.int64)
    first = np.zeros(n, dtype=np.uint64)
    count = np.zeros(n, dtype=np.uint32)
    rc = f_win(C.c_uint64(n), p(hp), p(start), p(p_hi), p(var_off), p(var_idx), p(tp), C.c_uint64(len(lab)), C.c_uint32(len(tp)), p(first), p(count))
    if rc == _lib.HAWK_E_UNSUPPORTED:
        return None
    _lib.check(rc, "hawk_host_variant_window")
    c_off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(count, out=c_off[1:])
    strand = np.asarray(G.strand)[alt_rows].astype(bool)
    pos = np.asarray(G.pos)[alt_rows].astype(np.int64)
    pivot = np.ascontiguousarray(np.where(np.logical_xor(bool(G.right), strand), pos, pos - G.guidelen), dtype=np.int64)
    cr = np.ascontiguousarray(cores[alt_rows], dtype=np.uint8)
    seg_start, seg_rel, seg_gen = lab.seg_csr
    rpool, roff = _pool(t_ref)
    apool, aoff = _pool(t_alt)
    rpool = rpool if len(rpool) else np.zeros(1, np.uint8)
    apool = apool if len(apool) else np.zeros(1, np.uint8)
    name_rank = np.argsort(np.argsort(np.array(lab.vid))).astype(np.uint32) if lab.vid else np.zeros(0, np.uint32)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    out_var = np.zeros(max(int(c_off[-1]), 1), dtype=np.uint32)
    need = np.zeros(n, dtype=np.uint8)
    _lib.check(f_pol(C.c_uint64(n), C.c_uint32(L), p(cr), p(hp), p(pivot), p(stop), p(first), p(c_off), p(var_idx), C.c_uint64(len(var_idx)),
                     p(seg_start), p(seg_rel), p(seg_gen), C.c_uint64(len(seg_start) - 1), p(tp), p(rpool), p(roff), p(apool), p(aoff),
                     C.c_uint32(len(tp)), p(name_rank), p(out_off), p(out_var), p(need)), "hawk_host_polish_windows")
    shown = out_var[:int(out_off[-1])]
    # ids in string order; allele frequencies in the ids' order (guide.py:311-328), "NA" when the row shows nothing or every frequency is missing
    af_names = sorted(set(t_af))
    af_rank = {s_: k for k, s_ in enumerate(af_names)}
    af_of_var = np.array([af_rank[x] for x in t_af], dtype=np.int64)
    ids = _ragged_join_raw(shown, out_off, *_pool(lab.vid))
    afs = _ragged_join_raw(af_of_var[shown], out_off, *_pool(af_names))
    n_shown = np.diff(out_off.astype(np.int64))
    has_af = np.zeros(n, dtype=bool)
    if len(shown):
        known = af_of_var[shown] != af_rank.get("NA", -1)
        has_af = np.bincount(np.repeat(np.arange(n), n_shown), weights=known, minlength=n) > 0
    NA = n  # rows of the extended pools: the alt rows' own strings, "NA", then the strings of the rows the Python mirror walked
    vid_lab = np.full(ng, NA, dtype=np.uint32)
    af_lab = np.full(ng, NA, dtype=np.uint32)
    vid_lab[alt_rows] = np.arange(n)
    af_lab[alt_rows] = np.where(has_af, np.arange(n), NA)
    py_vid: List[str] = []
    py_af: List[str] = []
    for k in np.flatnonzero(need).tolist():  # (rows on which the reference's own assertion fires: its Python mirror raises)
        g = int(alt_rows[k])
        v_s, a_s = _polish_row_python(G, lab, g, int(hp[k]), set(var_idx[int(first[k]):int(first[k]) + int(count[k])].tolist()), cores)
        vid_lab[g] = af_lab[g] = NA + 1 + len(py_vid)
        py_vid.append(v_s)
        py_af.append(a_s)
    return ids.extended(["NA"] + py_vid).select(vid_lab), afs.extended(["NA"] + py_af).select(af_lab)


def _polish_rows_native(G, lab: "HapLabels", alt_rows, hh, slow, a, cnt, v_var, cores, t_pos, t_ref, t_alt, name_rank):
    """annotation.polish_guide_variants for the rows `slow` (indices into alt_rows) by hawk_host_polish_rows.  Returns (CSR offsets,
    the shown variants in id order, need_python flags) or None when the helper is not available."""
    import ctypes as C
    from . import _lib
    try:
        fn = _host_lib().hawk_host_polish_rows
    except AttributeError:
        return None
    L = G.guidelen + G.pamlen
    n = len(slow)
    # candidates of every slow row, ascending and without duplicates
    c_cnt = cnt[slow]
    c_off = np.concatenate(([0], np.cumsum(c_cnt))).astype(np.uint64)
    tot = int(c_off[-1])
    src = np.repeat(a[slow], c_cnt) + (np.arange(tot) - np.repeat(c_off[:-1].astype(np.int64), c_cnt))
    cand = v_var[src].astype(np.uint32)
    row = np.repeat(np.arange(n), c_cnt)
    o = np.lexsort((cand, row))
    cand, row = cand[o], row[o]
    keep = np.ones(tot, dtype=bool)
    keep[1:] = (row[1:] != row[:-1]) | (cand[1:] != cand[:-1])
    cand, row = np.ascontiguousarray(cand[keep]), row[keep]
    c_off = np.concatenate(([0], np.cumsum(np.bincount(row, minlength=n)))).astype(np.uint64)
    g_rows = alt_rows[slow]
    strand = np.asarray(G.strand)[g_rows].astype(bool)
    stored_right = np.logical_xor(bool(G.right), strand)
    pos = np.asarray(G.pos)[g_rows].astype(np.int64)
    pivot = np.ascontiguousarray(np.where(stored_right, pos, pos - G.guidelen), dtype=np.int64)
    stop = np.ascontiguousarray(np.asarray(G.stop, dtype=np.int64)[g_rows])
    hp = np.ascontiguousarray(hh[slow], dtype=np.uint32)
    cr = np.ascontiguousarray(cores[g_rows], dtype=np.uint8)
    seg_start, seg_rel, seg_gen = lab.seg_csr
    rpool, roff = _pool(t_ref)
    apool, aoff = _pool(t_alt)
    if len(rpool) == 0:
        rpool = np.zeros(1, np.uint8)
    if len(apool) == 0:
        apool = np.zeros(1, np.uint8)
    tp = np.ascontiguousarray(t_pos, dtype=np.int64)
    nr = np.ascontiguousarray(name_rank, dtype=np.uint32)
    out_off = np.zeros(n + 1, dtype=np.uint64)
    out_var = np.zeros(max(len(cand), 1), dtype=np.uint32)
    need = np.zeros(n, dtype=np.uint8)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    _lib.check(fn(C.c_uint64(n), C.c_uint32(L), p(cr), p(hp), p(pivot), p(stop), p(c_off), p(cand), p(seg_start), p(seg_rel), p(seg_gen),
                  C.c_uint64(len(seg_start) - 1), p(tp), p(rpool), p(roff), p(apool), p(aoff), C.c_uint32(len(tp)), p(nr), p(out_off), p(out_var), p(need)),
               "hawk_host_polish_rows")
    return out_off, out_var[:int(out_off[-1])], need


def _variant_columns(G, rep_hap: np.ndarray, cores: np.ndarray, lab: HapLabels):
    vid, af = _variant_columns_raw(G, rep_hap, cores, lab)
    return vid.strings(), af.strings()


def _vid_index(lab: HapLabels, v: str) -> int:
    m = getattr(lab, "_vid_map", None)
    if m is None:
        m = lab._vid_map = {x: i for i, x in enumerate(lab.vid)}
    return m[v]


def group_kmers(G) -> List[str]:
    """Scorer inputs of the group representatives: `guide.sequence[6:-7].upper()` after reverse_guides
    (scoring.py:50-67) - 30-mers for 20+3 Cas9 guides, 34-mers for 23+4 Cpf1 guides."""
    from .hapset import decode_windows
    W = G.guidelen + G.pamlen + 2 * GUIDESEQPAD
    out = []
    for w, s in zip(decode_windows(np.asarray(G.win), W), np.asarray(G.strand).tolist()):
        if s == 1:
            w = w[::-1].translate(_RC_TRANS)
        out.append(w[6:-7].upper())
    return out


# ---- the report's columns as they are kept until they are written -------------------------------------------------------
class ConstCol:
    """the same text in every row"""

    def __init__(self, text: str, n: int):
        self.text, self.n = text, n

    def array(self):
        return np.full(self.n, self.text, dtype=object)

    def take(self, rows):
        return np.full(len(rows), self.text, dtype="U")


class FixedCol:
    """equally long ASCII strings as a byte matrix [rows, width]"""

    def __init__(self, m: np.ndarray):
        self.m = np.ascontiguousarray(m, dtype=np.uint8)

    def array(self):
        w = self.m.shape[1]
        return self.m.view(f"S{w}").ravel().astype(f"U{w}")

    def take(self, rows):
        w = self.m.shape[1]
        return np.ascontiguousarray(self.m[rows]).view(f"S{w}").ravel().astype(f"U{w}")


class IntCol:
    def __init__(self, v: np.ndarray):
        self.v = np.ascontiguousarray(v, dtype=np.int64)

    def array(self):
        return self.v

    def take(self, rows):
        return self.v[rows]


class VocabCol:
    """a few distinct strings, one index per row; `as_object`: what dtype the DataFrame route had for the column"""

    def __init__(self, idx: np.ndarray, vocab: Sequence[str], as_object: bool = True):
        self.idx, self.vocab, self.as_object = np.ascontiguousarray(idx, dtype=np.uint32), list(vocab), as_object

    def array(self):
        v = np.array(self.vocab, dtype=object) if self.as_object else np.array(self.vocab)
        return v[self.idx]

    def take(self, rows):
        return np.array(self.vocab, dtype="U")[self.idx[rows]]


def _ragged_array(col: Ragged):
    return np.array(col.strings(), dtype=object)


def _col_array(col):
    return _ragged_array(col) if isinstance(col, Ragged) else col.array()


def _col_take(col, rows):
    return np.array(col.take(rows), dtype="U") if isinstance(col, Ragged) else col.take(rows)


_UNPLAIN = ('"', "\t", "\n", "\r")


def _plain(strings) -> bool:
    """no string holds a character csv's minimal quoting reacts to (DataFrame.to_csv would quote the field)"""
    return not any(ch in x for x in strings for ch in _UNPLAIN)


def _report_order(cols: Dict[str, object], gcols: List[str], n: int) -> np.ndarray:
    """pandas groupby(sort=True) order over the group columns (reports.py:978-1003), then _format_report's stable sort on
    (start, stop) = ONE lexicographic order with (start, stop) as the leading keys and the group columns, compared as strings,
    behind them.  (start, stop) alone decides nearly every row: the string keys are only formed for the rows of a tie."""
    start, stop = cols["start"].v, cols["stop"].v
    order = np.lexsort((stop, start))
    s_, e_ = start[order], stop[order]
    same = np.zeros(n, dtype=bool)
    same[1:] = (s_[1:] == s_[:-1]) & (e_[1:] == e_[:-1])
    in_tie = same.copy()
    in_tie[:-1] |= same[1:]
    tpos = np.flatnonzero(in_tie)  # positions (in the (start, stop) order) of rows that share both with a neighbour
    if len(tpos) == 0:
        return order
    rows = order[tpos]
    # One group column after another, in the groupby's priority, and only over the rows the keys so far have not told apart (the
    # spacer decides nearly all of them; `samples` - kilobytes per row - is hardly ever looked at): every stage contributes an
    # integer rank, the final order is a lexsort of integers.
    cur = np.cumsum(~same[tpos]) - 1  # tie runs, numbered in output order
    ranks = [cur]
    live = np.arange(len(rows))
    for c in gcols:
        if c in ("chr", "start", "stop"):
            continue
        col = cols[c]
        if isinstance(col, ConstCol):
            continue
        _, inv = np.unique(_col_take(col, rows[live]), return_inverse=True)
        sub = np.zeros(len(rows), dtype=np.int64)
        sub[live] = inv.reshape(-1)
        ranks.append(sub)
        comp = cur[live] * (int(inv.max()) + 1) + inv.reshape(-1)
        _, cinv, ccnt = np.unique(comp, return_inverse=True, return_counts=True)
        cur = cur.copy()
        cur[live] = cinv.reshape(-1)           # (only compared among rows that were tied before: a dense id within `live` does)
        live = live[ccnt[cinv.reshape(-1)] > 1]
        if len(live) == 0:
            break
    order[tpos] = rows[np.lexsort(ranks[::-1])]
    return order


def group_columns(G, haplotypes, pam: PAM, contig: str, target: str, scores: Optional[Dict[str, np.ndarray]] = None,
                  with_cfdon: bool = True, is_ref_hap: Optional[np.ndarray] = None, offtargets=None):
    """The guide report of group-level inputs (hapset.GroupTable, tiling.MergedGroups.groups(), ReportGroups) as COLUMNS -
    {name: ConstCol / FixedCol / IntCol / VocabCol / Ragged} in the report's column order - plus the row order and whether every
    field is plain text (no character csv quoting reacts to).  Nothing here is a Python string per row: the two columns that list
    every carrier of every row (C3: 0.64 GB) stay the byte blobs the library's helpers wrote.  `scores[c]` is per GROUP."""
    names = select_reportcols(pam, G.right, offtargets is not None)
    ng = G.n_groups
    if ng == 0:
        return {c: Ragged(np.zeros(0, np.uint8), np.zeros(1, np.uint64)) for c in names}, np.zeros(0, np.int64), True
    L = G.guidelen + G.pamlen
    member_off = np.asarray(G.member_off, dtype=np.int64)
    member_hap = np.asarray(G.member_hap, dtype=np.int64)
    rep_hap = member_hap[member_off[:-1]]
    lab = haplotypes if isinstance(haplotypes, HapLabels) else HapLabels.from_objects(haplotypes)
    if is_ref_hap is None:
        is_ref_hap = lab.is_ref
    # The two columns that list every carrier of every row are byte gathers inside the library (no GIL held), and the row order needs
    # none of the text columns: three more threads while this one decodes the spacers and works out the variant columns
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=3)
    try:
        c_samples = pool.submit(_samples_raw, member_off, member_hap, lab.samples)
        c_hapids = pool.submit(_hapids_raw, member_off, member_hap, lab.ids)
        return _group_columns(G, lab, pam, contig, target, scores, with_cfdon, is_ref_hap, offtargets, names, member_off, member_hap, rep_hap,
                              pool, c_samples, c_hapids)
    finally:
        pool.shutdown(wait=True)


def _group_columns(G, lab, pam, contig, target, scores, with_cfdon, is_ref_hap, offtargets, names, member_off, member_hap, rep_hap, pool, c_samples,
                   c_hapids):
    ng, L = G.n_groups, G.guidelen + G.pamlen
    # cased + strand cores of the representatives, from the window slices
    sh = np.arange(GUIDESEQPAD, GUIDESEQPAD + L, dtype=np.uint64)
    code = np.zeros((ng, L), dtype=np.uint8)
    for p in range(5):
        code |= (((np.asarray(G.win[p])[:, None] >> sh) & np.uint64(1)).astype(np.uint8) << p)
    cores = _CODE2CHAR[code]
    # reverse_guides (annotation.py:27-51): strand-1 rows read as their reverse complement, case preserved
    strand = np.asarray(G.strand).astype(np.int64)
    guide = cores.copy()
    rev = strand == 1
    guide[rev] = _RC_LUT[cores[rev][:, ::-1]]
    if G.right:
        pam_b, sg_b = guide[:, :G.pamlen], guide[:, G.pamlen:]
    else:
        sg_b, pam_b = guide[:, :G.guidelen], guide[:, G.guidelen:]
    data: Dict[str, object] = {"chr": ConstCol(contig, ng), "start": IntCol(G.start), "stop": IntCol(G.stop), "sgRNA_sequence": FixedCol(sg_b),
                               "pam": FixedCol(pam_b), "pam_class": ConstCol(compute_pam_class(pam), ng),
                               "strand": VocabCol(rev.astype(np.uint32), ["+", "-"], as_object=False)}
    for c in [c for c in names if c.startswith("score_")]:
        vals = None
        if c == "score_cfdon" and with_cfdon and G.cfdon is not None:
            vals = np.asarray(G.cfdon, dtype=np.float64)
        elif scores and c in scores:
            vals = np.asarray(scores[c], dtype=np.float64)
        if vals is None:
            data[c] = ConstCol("NA", ng)
        else:  # str(round(score, 4)) once per distinct value ("NA" for a missing score)
            uniq, inv = np.unique(vals, return_inverse=True)
            data[c] = VocabCol(inv.reshape(-1), ["NA" if v != v else str(round_score(v)) for v in uniq.tolist()])
    gkey = np.asarray(G.gc_num).astype(np.int64) * 65536 + np.asarray(G.gc_den).astype(np.int64)
    uniq, inv = np.unique(gkey, return_inverse=True)
    data["gc_content"] = VocabCol(inv.reshape(-1), [str((int(k) >> 16) / (int(k) & 0xffff) if (int(k) & 0xffff) else 0.0) for k in uniq.tolist()])
    data["origin"] = VocabCol((~is_ref_hap[rep_hap]).astype(np.uint32), ["ref", "alt"], as_object=False)
    gcols = REPORTCOLS[:5]
    if pam.cas_system in (SPCAS9, XCAS9):
        gcols = gcols + REPORTCOLS[6:12] + REPORTCOLS[13:14] + REPORTCOLS[15:17]
    elif pam.cas_system == CPF1:
        gcols = gcols + REPORTCOLS[6:7] + REPORTCOLS[12:13] + REPORTCOLS[15:17]
    else:
        gcols = gcols + REPORTCOLS[6:7] + REPORTCOLS[15:17]
    c_order = pool.submit(_report_order, dict(data), gcols, ng)
    vid_col, af_col = _variant_columns_raw(G, rep_hap, cores, lab)
    samples_col, hapids_col, order = c_samples.result(), c_hapids.result(), c_order.result()
    data["samples"] = samples_col
    data["variant_id"] = vid_col
    data["af"] = af_col
    data["target"] = ConstCol(target, ng)
    data["haplotype_id"] = hapids_col
    plain = _plain([contig, target]) and _plain(lab.samples) and _plain(lab.ids) and _plain(lab.vid)
    if offtargets is not None:  # a dict {SPACER: (count, cfd)} or a callable that builds it from the rows' spacers
        spacers = data["sgRNA_sequence"].array()
        if callable(offtargets):
            offtargets = offtargets(sorted(set(np.char.upper(spacers).tolist())))
        for c, arr in _offtarget_columns(spacers.tolist(), offtargets, pam.cas_system in (SPCAS9, XCAS9)).items():
            data[c] = Ragged.from_strings([str(x) for x in arr.tolist()])
            plain = plain and _plain(arr.tolist())
    return {c: data[c] for c in names}, order, plain


def report_from_groups(G, haplotypes, pam: PAM, contig: str, target: str, scores: Optional[Dict[str, np.ndarray]] = None,
                       with_cfdon: bool = True, is_ref_hap: Optional[np.ndarray] = None, offtargets=None):
    """report_frame's result from group-level inputs, assembled column by column (group_columns): same DataFrame, same order,
    same strings."""
    import pandas as pd
    cols, order, _ = group_columns(G, haplotypes, pam, contig, target, scores, with_cfdon, is_ref_hap, offtargets)
    if G.n_groups == 0:
        return pd.DataFrame({c: [] for c in cols})
    return pd.DataFrame({c: _col_array(col)[order] for c, col in cols.items()})


def write_report_tsv(path: str, cols: Dict[str, object], order: np.ndarray, plain: bool = True) -> int:
    """What _store_report writes (reports.py:739: DataFrame.to_csv(sep="\\t", index=False)) for the columns of group_columns,
    by the library's TSV writer (hawk_host_tsv_write: multi-threaded, straight into a mapping of the file) - no row string, no
    DataFrame.  Fields csv quoting would touch (`plain` False) go through pandas as before.  Returns the bytes written."""
    import ctypes as C
    from . import _lib
    names = list(cols)
    n = len(order)
    if n == 0 or not plain:
        import pandas as pd
        df = pd.DataFrame({c: (_col_array(col)[order] if n else []) for c, col in cols.items()})
        txt = to_tsv(df)
        with open(path, "w") as f:
            f.write(txt)
        return len(txt)

    class TsvCol(C.Structure):
        _fields_ = [("kind", C.c_uint32), ("width", C.c_uint32), ("data", C.c_void_p), ("off", C.c_void_p), ("pool", C.c_void_p), ("n_vocab", C.c_uint64)]
    keep = []  # arrays the descriptors point into
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    arr = (TsvCol * len(names))()
    for k, c in enumerate(names):
        col = cols[c]
        if isinstance(col, ConstCol):
            b = np.frombuffer(col.text.encode("ascii"), dtype=np.uint8).copy() if col.text else np.zeros(1, np.uint8)
            keep.append(b)
            arr[k] = TsvCol(0, len(col.text), p(b).value, None, None, 0)
        elif isinstance(col, FixedCol):
            arr[k] = TsvCol(1, col.m.shape[1], p(col.m).value, None, None, 0)
        elif isinstance(col, Ragged):
            arr[k] = TsvCol(2, 0, p(col.blob).value, p(col.off).value, None, 0)
        elif isinstance(col, IntCol):
            arr[k] = TsvCol(3, 0, p(col.v).value, None, None, 0)
        else:
            pool, poff = _pool(col.vocab)
            if len(pool) == 0:
                pool = np.zeros(1, np.uint8)
            keep += [pool, poff]
            arr[k] = TsvCol(4, 0, p(col.idx).value, p(poff).value, p(pool).value, len(col.vocab))
    header = ("\t".join(names) + "\n").encode("ascii")
    od = np.ascontiguousarray(order, dtype=np.uint64)
    nbytes = C.c_uint64(0)
    _lib.check(_host_lib().hawk_host_tsv_write(path.encode(), header, C.c_uint64(len(header)), C.c_uint64(n), p(od), C.c_uint32(len(names)), arr,
                                               C.byref(nbytes)), "hawk_host_tsv_write")
    return int(nbytes.value)


def report_from_guides(guides, haplotypes, pam: PAM, contig: str, target: str, cfdon: Optional[Sequence[float]] = None,
                       offtargets: Optional[Dict[str, tuple]] = None):
    """The report of a Guide list as search() returns it (windows still on the + strand, i.e. BEFORE
    annotation.reverse_guides) - the route of unphased inputs, whose guides are resolved on the host
    (search_guides.resolve_guide) and therefore are not rows of the device table.  Rows the report merges are grouped
    here as _collapse_report_entries groups them; `cfdon[i]` is guide i's CFDon score (NaN = "NA")."""
    if not guides:
        return report_frame(ReportInput(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros(0, np.int64),
                                        np.zeros(0, np.int64), [], None, np.zeros(0, np.int64), np.zeros(1, np.int64), np.zeros(0, np.uint8),
                                        np.zeros(0, np.uint8), 0, len(pam), False), haplotypes, pam, contig, target)
    g0 = guides[0]
    guidelen, pamlen = g0.guidelen, g0.pamlen
    right = bool(g0.right) != bool(g0.strand)  # Guide.right is stored flipped for strand 1 (search_guides.py:538)
    n = len(guides)
    start = np.array([g.start for g in guides], dtype=np.int64)
    stop = np.array([g.stop for g in guides], dtype=np.int64)
    strand = np.array([g.strand for g in guides], dtype=np.uint8)
    hap = np.array([g._hip_hap for g in guides], dtype=np.int64)
    pos = np.array([g._hip_pos for g in guides], dtype=np.int64)
    wins = [g.sequence for g in guides]
    cfd = None if cfdon is None else np.asarray(cfdon, dtype=np.float64)
    groups: Dict[tuple, List[int]] = {}
    for i in range(n):
        sc = None if cfd is None else (None if cfd[i] != cfd[i] else round_score(float(cfd[i])))
        key = (int(start[i]), int(stop[i]), int(strand[i]), haplotypes[int(hap[i])].samples == "REF", wins[i][GUIDESEQPAD:-GUIDESEQPAD], sc)
        groups.setdefault(key, []).append(i)
    perm, off, num, den = [], [0], [], []
    for key, rows in groups.items():
        perm += rows
        off.append(len(perm))
        core = key[4]
        pamfirst = right != bool(key[2])
        spacer = core[pamlen:] if pamfirst else core[:guidelen]
        gc = sum(spacer.count(c) for c in "CGScgs")
        num.append(gc)
        den.append(gc + sum(spacer.count(c) for c in "ATWUatwu"))
    inp = ReportInput(start, stop, strand, hap, pos, wins, cfd, np.array(perm, dtype=np.int64), np.array(off, dtype=np.int64),
                      np.array(num), np.array(den), guidelen, pamlen, right)
    return report_frame(inp, haplotypes, pam, contig, target, with_cfdon=cfd is not None, offtargets=offtargets)
