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


def select_reportcols(pam: PAM, right: bool) -> List[str]:
    """Final column order (reports.py:612-660) without the optional annotation / off-target / Elevation columns."""
    cols = (REPORTCOLS[:3] + REPORTCOLS[4:5] + REPORTCOLS[3:4] + REPORTCOLS[5:7]) if right else REPORTCOLS[:7]
    if pam.cas_system in (SPCAS9, XCAS9):
        cols = cols + REPORTCOLS[7:12] + REPORTCOLS[13:14]
    elif pam.cas_system == CPF1:
        cols = cols + REPORTCOLS[12:13]
    return cols + REPORTCOLS[15:20] + REPORTCOLS[20:22]


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
                 with_cfdon: bool = True):
    """The collapsed, sorted report of one region as a pandas DataFrame (what reports.report_guides hands to
    to_csv).  `haplotypes[i]` needs .samples .variants .afs .id and .segments (PosSegments); `scores` may map a
    score column name to a per-row float array (NaN -> "NA")."""
    import pandas as pd
    cols = select_reportcols(pam, inp.right)
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
    """What _store_report writes (reports.py:739)."""
    return df.to_csv(sep="\t", index=False)


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
