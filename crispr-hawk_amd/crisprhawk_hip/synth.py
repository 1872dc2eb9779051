"""Deterministic synthetic inputs for the guide-search hot path.

The recipes follow SURVEY.md §8(d) / BASELINE.md §3: iid uniform ACGT contigs, a BED
interval padded by 100 nt on each side, and a phased variant panel (SNV / deletion /
insertion sites with log-uniform allele frequencies, every haplotype column carrying the
alt allele independently).  Everything is driven by ``numpy.random.default_rng(seed)`` so
the golden-fixture generator (tests/golden/make_golden.py, which feeds the *reference*),
the parity tests and bench.py all see the same bytes.

Coordinates follow the reference: a BED line ``chrom start stop`` gives a region whose
string index 0 is the 1-based genomic position ``max(0, start-100)`` and whose sequence is
``contig[startp-1 : stop+100]`` (reference: coordinate.py:21-43, sequence.py:340-343,
region_constructor.py:21).
"""

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

PADDING = 100  # reference: region_constructor.py:21

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_IUPAC_EXTRA = np.frombuffer(b"NRYSWKMBDHV", dtype=np.uint8)


def random_sequence(rng: np.random.Generator, n: int, iupac_frac: float = 0.0) -> str:
    """iid uniform ACGT string; with ``iupac_frac`` > 0 sprinkle N/IUPAC codes."""
    seq = _ACGT[rng.integers(0, 4, size=n)]
    if iupac_frac > 0.0 and n > 0:
        mask = rng.random(n) < iupac_frac
        seq = seq.copy()
        seq[mask] = _IUPAC_EXTRA[rng.integers(0, len(_IUPAC_EXTRA), size=int(mask.sum()))]
    return seq.tobytes().decode("ascii")


@dataclass
class VariantSite:
    """One biallelic VCF row (left-anchored indels, VCF conventions)."""

    pos: int  # 1-based genomic position of the first REF base
    ref: str
    alt: str
    af: float
    gt: np.ndarray  # uint8 [n_samples, 2]; 1 = alt on that chromosome copy

    @property
    def chain(self) -> int:
        return len(self.alt) - len(self.ref)

    @property
    def is_snv(self) -> bool:
        return len(self.ref) == len(self.alt)


@dataclass
class SynthRegion:
    contig: str
    contig_seq: str
    bed_start: int
    bed_stop: int
    samples: List[str] = field(default_factory=list)
    variants: List[VariantSite] = field(default_factory=list)
    gt_matrix: Optional[np.ndarray] = None  # uint8 [len(variants), 2 * n_samples]: the records' genotypes as one matrix

    def variant_columns(self):
        """The records as columns - positions, REF / ALT alleles as one byte blob each with offsets - next to the genotype
        matrix: the in-memory form the device path ingests (built once per variant list, like `gt_matrix`)."""
        key = (id(self.variants), len(self.variants))
        c = getattr(self, "_vcols", None)
        if c is None or c[0] != key:
            pos = np.array([v.pos for v in self.variants], dtype=np.int64)
            reflen = np.array([len(v.ref) for v in self.variants], dtype=np.int64)
            altlen = np.array([len(v.alt) for v in self.variants], dtype=np.int64)
            ref_blob = np.frombuffer("".join(v.ref for v in self.variants).encode("ascii"), dtype=np.uint8)
            alt_blob = np.frombuffer("".join(v.alt for v in self.variants).encode("ascii"), dtype=np.uint8)
            off = lambda ln: np.concatenate(([0], np.cumsum(ln)))
            c = (key, dict(pos=pos, ref_blob=ref_blob, ref_off=off(reflen), alt_blob=alt_blob, alt_off=off(altlen)))
            self._vcols = c
        return c[1]

    @property
    def startp(self) -> int:  # padded start (reference Coordinate.start)
        return max(0, self.bed_start - PADDING)

    @property
    def stopp(self) -> int:  # padded stop (reference Coordinate.stop)
        return self.bed_stop + PADDING

    @property
    def sequence(self) -> str:
        return self.contig_seq[self.startp - 1 : self.stopp]

    def vcf_fields(self, v: VariantSite) -> List[str]:
        """The tab-split VCF row the reference's VariantRecord.read_vcf_line consumes."""
        gts = [f"{int(a)}|{int(b)}" for a, b in v.gt]
        return [self.contig, str(v.pos), ".", v.ref, v.alt, ".", "PASS", f"AF={v.af:.6g}", "GT"] + gts


def make_region(
    seed: int,
    contig: str,
    contig_len: int,
    bed_start: int,
    bed_stop: int,
    iupac_frac: float = 0.0,
) -> SynthRegion:
    rng = np.random.default_rng(seed)
    return SynthRegion(contig, random_sequence(rng, contig_len, iupac_frac), bed_start, bed_stop)


def add_phased_variants(
    reg: SynthRegion,
    seed: int,
    n_sites: int,
    n_samples: int,
    frac_snv: float = 0.90,
    frac_del: float = 0.05,
    max_indel: int = 8,
    af_min: Optional[float] = None,
    af_max: float = 0.5,
    edge_margin: int = 1,
    panel: str = "independent",
    n_founders: int = 128,
    block_sites: int = 3000,
) -> SynthRegion:
    """Place ``n_sites`` non-overlapping variant sites inside the padded region.

    SNV / deletion / insertion mix 90/5/5 %, indel length 1+Geometric(0.5) capped at
    ``max_indel``; AF log-uniform on [af_min, af_max]; every one of the 2*n_samples
    haplotype columns carries the alt independently with probability AF
    (``panel="independent"``: no linkage, the worst case for haplotype uniqueness - SURVEY 8(d)'s C3).
    ``panel="linked"``: the columns are mosaics of ``n_founders`` founder haplotypes (drawn as above), switching founder after
    Geometric(1 / block_sites) sites (~100 kb at C3's density) - linkage blocks as in a real phased panel: neighbouring
    variants travel together, so variant clusters are shared by more chromosome copies.
    """
    rng = np.random.default_rng(seed)
    n_hap = 2 * n_samples
    if af_min is None:
        af_min = 1.0 / n_hap
    lo = reg.startp + edge_margin  # first allowed 1-based position
    hi = reg.stopp - edge_margin - max_indel - 1  # last allowed start so REF allele fits
    span = max_indel + 2  # footprint reserved per site: keeps sites non-overlapping
    n_slots = (hi - lo + 1) // span
    if n_sites > n_slots:
        raise ValueError(f"too many variant sites ({n_sites}) for region ({n_slots} slots)")
    slots = np.sort(rng.choice(n_slots, size=n_sites, replace=False))
    jitter = rng.integers(0, 2, size=n_sites)  # 0/1 nt jitter inside the slot
    kinds = rng.random(n_sites)
    lens = np.minimum(rng.geometric(0.5, size=n_sites), max_indel)
    afs = np.exp(rng.uniform(np.log(af_min), np.log(af_max), size=n_sites))
    alt_pick = rng.integers(1, 4, size=n_sites)
    ins_bases = rng.integers(0, 4, size=(n_sites, max_indel))
    seq = reg.contig_seq
    reg.samples = [f"S{i:04d}" for i in range(n_samples)]
    out: List[VariantSite] = []
    ks: List[int] = []  # row of gt_all behind every record of `out`
    gt_all = np.empty((n_sites, n_hap), dtype=np.uint8)
    if panel == "linked":
        founders = (rng.random((n_sites, n_founders)) < afs[:, None]).astype(np.uint8)
        lrng = np.random.default_rng(seed ^ 0x5EED1)
        for c in range(n_hap):
            k = 0
            while k < n_sites:
                k1 = min(n_sites, k + int(lrng.geometric(1.0 / block_sites)))
                gt_all[k:k1, c] = founders[k:k1, int(lrng.integers(0, n_founders))]
                k = k1
    elif panel == "independent":
        for k0 in range(0, n_sites, 4096):  # chunked: 31k x 5008 float64 would be 1.2 GB at once
            k1 = min(n_sites, k0 + 4096)
            gt_all[k0:k1] = rng.random((k1 - k0, n_hap)) < afs[k0:k1, None]
    else:
        raise ValueError(f"unknown panel kind {panel!r}")
    empty = np.flatnonzero(~gt_all.any(axis=1))  # keep every site carried by at least one haplotype
    gt_all[empty, rng.integers(0, n_hap, size=len(empty))] = 1
    for k in range(n_sites):
        pos = int(lo + slots[k] * span + jitter[k])
        refb = seq[pos - 1]
        if refb not in "ACGT":
            continue  # no variant on an ambiguous reference base
        gt = gt_all[k].reshape(n_samples, 2)
        ks.append(k)
        if kinds[k] < frac_snv:
            altb = "ACGT"[("ACGT".index(refb) + int(alt_pick[k])) % 4]
            out.append(VariantSite(pos, refb, altb, float(afs[k]), gt))
        elif kinds[k] < frac_snv + frac_del:
            ln = int(lens[k])
            out.append(VariantSite(pos, seq[pos - 1 : pos + ln], refb, float(afs[k]), gt))
        else:
            ln = int(lens[k])
            ins = "".join("ACGT"[b] for b in ins_bases[k, :ln])
            out.append(VariantSite(pos, refb, refb + ins, float(afs[k]), gt))
    # The reference clamps a variant's REF span with the *original* region length
    # (haplotype.py:199-201: ``if posrel_stop > self._size``), so a site whose haplotype
    # offset has been pushed past the region end by upstream insertions makes it raise
    # "Mismatching reference alleles".  Keep sites out of that tail: the guard is the
    # largest total inserted length any one haplotype carries.
    if out:
        ins_len = np.array([max(0, v.chain) for v in out], dtype=np.int64)
        ins_rows = np.flatnonzero(ins_len > 0)
        ins_gt = gt_all[np.asarray(ks, dtype=np.int64)[ins_rows]]  # [kept insertion, hap] (sites on ambiguous REF bases were skipped)
        guard = int((ins_len[ins_rows, None] * ins_gt.astype(np.int64)).sum(axis=0).max()) if len(ins_rows) else 0
        keep = [i for i, v in enumerate(out) if v.pos + len(v.ref) + guard <= reg.stopp]
        out, ks = [out[i] for i in keep], [ks[i] for i in keep]
    reg.variants = out
    reg.gt_matrix = gt_all[np.asarray(ks, dtype=np.int64)] if ks else None
    reg.variant_columns()
    return reg


# --- named configurations (BASELINE.json configs / SURVEY.md §8d) -----------------

def config_c1() -> SynthRegion:
    """C1/C2: 12 kb contig, BED chrS 1000 11000 -> 10 201-nt region, no VCF."""
    return make_region(1001, "chrS", 12_000, 1_000, 11_000)


def config_c3(n_samples: int = 2504, n_sites: int = 31_000, region_len: int = 1_000_000) -> SynthRegion:
    """C3: 1.2 Mb synthetic chr22, BED 100000..1100000, 2504 phased samples, 31k sites."""
    reg = make_region(1003, "chr22", region_len + 200_000, 100_000, 100_000 + region_len)
    return add_phased_variants(reg, 1003_1, n_sites, n_samples)


def cfd_tables(seed: int = 2001) -> Tuple[np.ndarray, np.ndarray]:
    """Seeded synthetic CFD tables (the real Doench-2016 pickles are fetched from Zenodo by
    the reference at run time and are not available offline).

    Returns ``mm[20, 4, 4]`` indexed [position, wildtype-RNA base (A,C,G,U), sgRNA base
    (A,C,G,T)] and ``pam[16]`` indexed by the dinucleotide 4*b0+b1 over (A,C,G,T).
    """
    rng = np.random.default_rng(seed)
    return rng.uniform(0.0, 1.0, size=(20, 4, 4)), rng.uniform(0.0, 1.0, size=16)


def cfd_tables_as_dicts(mm: np.ndarray, pam: np.ndarray):
    """The two dicts in the key format reference compute_cfd expects
    (scores/cfdscore/cfdscore.py:88-94): ``r{wt}:d{RC(sg)},{i+1}`` and ``{pam[-2:]}``."""
    rna, dna = "ACGU", "ACGT"
    rc = {"A": "T", "C": "G", "G": "C", "T": "A"}
    mmd = {}
    for i in range(20):
        for a in range(4):
            for b in range(4):
                mmd[f"r{rna[a]}:d{rc[dna[b]]},{i + 1}"] = float(mm[i, a, b])
    pamd = {dna[a] + dna[b]: float(pam[4 * a + b]) for a in range(4) for b in range(4)}
    return mmd, pamd


def deepcpf1_weights(seed: int = 2002) -> dict:
    """Seeded N(0, 0.1) fp32 parameters in the torch layout of the reference's SeqDeepCpf1
    (scores/deepCpf1/seqdeepcpf1.py:43-56): conv (80,4,5), fc 1200-80-40-40-1."""
    rng = np.random.default_rng(seed)

    def n(*shape):
        return rng.normal(0.0, 0.1, size=shape).astype(np.float32)

    return dict(
        conv_w=n(80, 4, 5), conv_b=n(80),
        w1=n(80, 1200), b1=n(80), w2=n(40, 80), b2=n(40),
        w3=n(40, 40), b3=n(40), w4=n(1, 40), b4=n(1),
    )


# --- C4: a whole contig with a phased panel too large to hold as a dense matrix ------------------
class BlockGenotypes:
    """The SURVEY §8(d) genotype recipe (every chromosome copy carries the alt independently with probability AF) as a
    function of (variant, column) that is generated on demand: the matrix is cut into fixed blocks of `vblock` variants
    x `cblock` columns, each block drawn from its own `default_rng([seed, variant block, column block])`, so any
    rectangle - a tile's variants x a rank's columns - is reproducible whatever the tiling or the number of ranks.
    Interface of tiling.DenseGenotypes: carried(var_lo, var_hi, col_lo, col_hi)."""

    def __init__(self, seed: int, af: np.ndarray, n_cols: int, vblock: int = 4096, cblock: int = 626):
        self.seed, self.af, self.n_cols, self.vblock, self.cblock = seed, np.asarray(af, dtype=np.float32), n_cols, vblock, cblock

    def block(self, vb: int, cb: int) -> np.ndarray:
        v0, v1 = vb * self.vblock, min(len(self.af), (vb + 1) * self.vblock)
        c0, c1 = cb * self.cblock, min(self.n_cols, (cb + 1) * self.cblock)
        rng = np.random.default_rng([self.seed, vb, cb])
        return rng.random((v1 - v0, c1 - c0), dtype=np.float32) < self.af[v0:v1, None]

    def dense(self, var_lo: int, var_hi: int, col_lo: int, col_hi: int) -> np.ndarray:
        out = np.zeros((var_hi - var_lo, col_hi - col_lo), dtype=bool)
        for vb in range(var_lo // self.vblock, (max(var_hi, var_lo + 1) - 1) // self.vblock + 1):
            v0 = vb * self.vblock
            a, b = max(var_lo, v0), min(var_hi, v0 + self.vblock)
            if b <= a:
                continue
            for cb in range(col_lo // self.cblock, (max(col_hi, col_lo + 1) - 1) // self.cblock + 1):
                c0 = cb * self.cblock
                c, d = max(col_lo, c0), min(col_hi, c0 + self.cblock)
                if d <= c:
                    continue
                out[a - var_lo:b - var_lo, c - col_lo:d - col_lo] = self.block(vb, cb)[a - v0:b - v0, c - c0:d - c0]
        return out

    def carried(self, var_lo: int, var_hi: int, col_lo: int, col_hi: int):
        counts = np.zeros(col_hi - col_lo, dtype=np.int64)
        parts = []
        step = self.cblock
        for c in range(col_lo, col_hi, step):  # column slabs: the transpose scan stays cache-sized
            d = min(col_hi, c + step)
            sub = self.dense(var_lo, var_hi, c, d)
            sites, cols = np.nonzero(sub)  # variant-major scan, then a stable (radix) sort by column: (column, variant) order
            order = np.argsort(cols.astype(np.uint16), kind="stable")
            counts[c - col_lo:d - col_lo] = np.bincount(cols, minlength=d - c)
            parts.append(sites[order].astype(np.uint32))
        return counts, (np.concatenate(parts) if parts else np.zeros(0, np.uint32))


def contig_panel(seed: int, contig: str, contig_len: int, n_block: int, n_samples: int, sites_per_mb: float = 31_000.0,
                 frac_snv: float = 0.90, frac_del: float = 0.05, max_indel: int = 8):
    """C4: an iid ACGT contig with one leading N block and a phased panel at 1000G density over the
    non-N part, fully vectorised (1.25 M sites for chr22).  Returns (contig bases as a uint8 array, tiling.VariantPanel
    with BlockGenotypes).  Sites are non-overlapping (one per slot of max_indel + 2 bases) and never touch N."""
    from .tiling import VariantPanel
    rng = np.random.default_rng(seed)
    seq = np.empty(contig_len, dtype=np.uint8)
    seq[:n_block] = ord("N")
    seq[n_block:] = _ACGT[rng.integers(0, 4, size=contig_len - n_block)]
    span = max_indel + 2
    lo = n_block + 201            # 1-based first allowed position: clear of the N block and of the region pad
    hi = contig_len - 200 - span  # keep the tail free: inserted bases push later variants towards the region end
    n_slots = max(0, (hi - lo) // span)
    n_sites = min(n_slots, int(round(sites_per_mb * (contig_len - n_block) / 1e6)))
    slots = np.sort(rng.choice(n_slots, size=n_sites, replace=False))
    pos = lo + slots * span + rng.integers(0, 2, size=n_sites)
    kinds = rng.random(n_sites)
    lens = np.minimum(rng.geometric(0.5, size=n_sites), max_indel)
    n_cols = 2 * n_samples
    af = np.exp(rng.uniform(np.log(1.0 / n_cols), np.log(0.5), size=n_sites))
    alt_pick = rng.integers(1, 4, size=n_sites)
    ins = rng.integers(0, 4, size=(n_sites, max_indel))
    code = np.zeros(256, dtype=np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    refb = seq[pos - 1]
    is_snv = kinds < frac_snv
    is_del = ~is_snv & (kinds < frac_snv + frac_del)
    ref: List[str] = [None] * n_sites
    alt: List[str] = [None] * n_sites
    refc = [chr(c) for c in refb.tolist()]
    snv_alt = _ACGT[(code[refb] + alt_pick) % 4]
    sa = [chr(c) for c in snv_alt.tolist()]
    blob = seq.tobytes()
    for k in np.flatnonzero(is_snv).tolist():
        ref[k], alt[k] = refc[k], sa[k]
    for k in np.flatnonzero(is_del).tolist():
        p = int(pos[k])
        ref[k], alt[k] = blob[p - 1:p + int(lens[k])].decode(), refc[k]
    for k in np.flatnonzero(~is_snv & ~is_del).tolist():
        ref[k], alt[k] = refc[k], refc[k] + "".join("ACGT"[b] for b in ins[k, :int(lens[k])])
    vid = [f"{contig}-{p}-{r}/{a}" for p, r, a in zip(pos.tolist(), ref, alt)]
    samples = [f"S{i:04d}" for i in range(n_samples)]
    return seq, VariantPanel(pos.astype(np.int64), ref, alt, vid, af.astype(np.float64), samples,
                             BlockGenotypes(seed + 1, af, n_cols))


def write_region_files(reg: SynthRegion, directory: str, stem: str = "region"):
    """The region as the FILES the reference's `crisprhawk search` takes - FASTA (60 columns), BED, phased VCF text (one `a|b`
    per sample and record, `AF=` in INFO) - for the files -> TSV measurements and tests.  Returns (fasta, bed, vcf) paths."""
    import os
    fa, bed, vcf = (os.path.join(directory, f"{stem}.{ext}") for ext in ("fa", "bed", "vcf"))
    with open(fa, "w") as f:
        f.write(f">{reg.contig}\n")
        s = reg.contig_seq
        f.write("\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n")
    with open(bed, "w") as f:
        f.write(f"{reg.contig}\t{reg.bed_start}\t{reg.bed_stop}\n")
    ns = len(reg.samples)
    with open(vcf, "wb") as f:
        f.write(b"##fileformat=VCFv4.2\n##contig=<ID=" + reg.contig.encode() + b">\n")
        f.write(("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(reg.samples) + "\n").encode())
        for v in reg.variants:
            head = f"{reg.contig}\t{v.pos}\t.\t{v.ref}\t{v.alt}\t.\tPASS\tAF={v.af:.6g}\tGT\t".encode()
            g = np.empty((ns, 4), np.uint8)
            g[:, 0] = v.gt[:, 0] + 48
            g[:, 1] = ord("|")
            g[:, 2] = v.gt[:, 1] + 48
            g[:, 3] = 9
            f.write(head + g.reshape(-1).tobytes()[:-1] + b"\n")
    return fa, bed, vcf

