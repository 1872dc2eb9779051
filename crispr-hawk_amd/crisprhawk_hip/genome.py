"""Genome index for the off-target scan (K7): contigs cut into equal plane rows in HBM.

The reference hands CRISPRitz a pre-built genome index directory (offtargets.py:264-268); here
the "index" is the genome itself as one-hot bit-planes.  Contigs are split into pieces of
``piece`` bases, each extended by ``overlap`` bases of its successor so every window is seen
whole by exactly one piece (the piece that owns its start)."""
import ctypes as C
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .hapset import DeviceHapSet, HostHaplotype, PosSegments, _p

_CODE2BASE = np.frombuffer(b"ACGT", dtype=np.uint8)


def read_fasta(path: str) -> Dict[str, str]:
    """Minimal multi-FASTA reader (plain text).  The reference reads FASTA through pysam
    (sequence.py:183-360), which is out of scope; this is only a convenience feeder."""
    out, name, buf = {}, None, []
    with open(path) as f:
        for line in f:
            if line.startswith(">"):
                if name is not None:
                    out[name] = "".join(buf)
                name, buf = line[1:].split()[0], []
            else:
                buf.append(line.strip())
    if name is not None:
        out[name] = "".join(buf)
    return out


def encode_guides(guides: Sequence[str]) -> np.ndarray:
    """Spacers (5'->3', ACGT only) -> one uint64 each, base i at bits 2i,2i+1 (A0 C1 G2 T3)."""
    n = len(guides)
    out = np.zeros(n, dtype=np.uint64)
    if n == 0:
        return out
    lens = {len(g) for g in guides}
    if len(lens) != 1 or not 0 < next(iter(lens)) <= 32:
        raise ValueError("guides must share one length of 1..32 bases")
    L = next(iter(lens))
    m = np.frombuffer("".join(guides).upper().encode("ascii"), dtype=np.uint8).reshape(n, L)
    lut = np.full(256, 255, dtype=np.uint8)
    lut[[65, 67, 71, 84]] = (0, 1, 2, 3)
    codes = lut[m]
    bad = np.flatnonzero((codes == 255).any(axis=1))
    if len(bad):
        raise ValueError(f"guide {guides[int(bad[0])]!r} holds a non-ACGT base")
    return (codes.astype(np.uint64) << (2 * np.arange(L, dtype=np.uint64))).sum(axis=1, dtype=np.uint64)


def decode_window(code: int, nmask: int, length: int) -> str:
    return "".join("N" if (nmask >> i) & 1 else "ACGT"[(code >> (2 * i)) & 3] for i in range(length))


@dataclass
class OffTargetHit:
    guide: int      # index into the guide list
    contig: str
    position: int   # 0-based start of the window on the + strand
    strand: str     # "+" / "-"
    mm: int
    window: str     # guidelen+pamlen bases in guide orientation (5'->3'), N for ambiguous bases


@dataclass
class BulgeHit:
    guide: int        # index into the guide list
    contig: str
    position: int     # 0-based start of the window (site spacer + PAM) on the + strand
    strand: str
    mm: int           # mismatches among the paired bases
    bulge_type: str   # "DNA" / "RNA"
    bulge_size: int
    crrna: str        # the guide spacer, '-' where the DNA has a base without partner
    dna: str          # the site spacer, mismatches lower-case, '-' where the guide has a base without partner
    pam: str          # the site's own PAM
    gaps: int         # bit i: position i (of the site spacer for DNA bulges, of the guide for RNA bulges) is bulged out


def _derived_guides(guides: Sequence[str], G: int, b: int, dna: bool):
    """The mismatch-only guides a bulge of b bases turns every guide into - RNA: b interior bases deleted; DNA: b bases (each of
    A, C, G, T) inserted between two bases - with the guide each came from and its bulge positions as a bitmask (of the guide
    for RNA, of the derived guide = the site spacer for DNA).  RNA: identical derived guides of one guide are kept once, under
    their lexicographically smallest positions (the paired bases are the same string either way).  DNA: every (placement,
    inserted bases) is its own derived guide - two placements may spell the same string and still pair the site's bases with
    different guide bases, so the host must see each."""
    from itertools import combinations, combinations_with_replacement, product
    derived, owner, gaps = [], [], []
    for gi, g in enumerate(guides):
        seen = {}
        if dna:
            for slots in combinations_with_replacement(range(1, G), b):          # insert before guide base s, ascending
                pos = tuple(s + k for k, s in enumerate(slots))                    # positions of the inserted bases in the derived guide
                for bases in product("ACGT", repeat=b):
                    chars, ins = list(g), 0
                    for s, x in zip(slots, bases):
                        chars.insert(s + ins, x)
                        ins += 1
                    derived.append("".join(chars))
                    owner.append(gi)
                    gaps.append(sum(1 << q for q in pos))
        else:
            for dele in combinations(range(1, G - 1), b):
                d = "".join(c for i, c in enumerate(g) if i not in dele)
                if d not in seen or dele < seen[d]:
                    seen[d] = dele
        for d, pos in seen.items():
            derived.append(d)
            owner.append(gi)
            gaps.append(sum(1 << p for p in pos))
    return derived, owner, gaps


class GenomeIndex:
    """`shard = (rank, world)`: the genome's rows (pieces) are block-partitioned over the ranks of a multi-GPU job
    ("shard the genome by contig/offset across ranks, guides replicated, rows gathered"); every rank keeps the
    descriptors of ALL rows, so a hit's global row index names its contig and offset anywhere."""

    def __init__(self, contigs: Dict[str, object], guidelen: int, pamlen: int, piece: int = 1 << 22, device: Optional[int] = None,
                 shard: Optional[Tuple[int, int]] = None, max_bulge: int = 0):
        """`max_bulge`: the largest DNA bulge the index will be asked about - neighbouring rows then overlap by that many bases
        more, so that the longer windows of bulged sites (guidelen + bulge + pamlen) still lie inside one row."""
        self.L = guidelen + pamlen
        self.guidelen, self.pamlen, self.max_bulge = guidelen, pamlen, int(max_bulge)
        overlap = self.L + self.max_bulge - 1
        self.rows: List[Tuple[str, int, int]] = []  # (contig, offset, owned window starts), all ranks' rows
        spans = []
        self.total = 0
        for name, seq in contigs.items():
            n = len(seq)
            self.total += n
            for off in range(0, max(n, 1), piece):
                end = min(n, off + piece + overlap)
                if end - off < self.L:
                    continue
                own = min(piece, n - off)  # window starts [0, own) belong to this row ...
                own = min(own, end - off - self.L + 1)  # ... as far as the window fits
                spans.append((name, off, end))
                self.rows.append((name, off, own))
        if not self.rows:
            raise ValueError("genome shorter than one guide+PAM window")
        self.n_rows_total = len(self.rows)
        self.row_lo, self.row_hi = 0, self.n_rows_total
        if shard is not None and shard[1] > 1:
            from .parallel import shard_range
            self.row_lo, self.row_hi = shard_range(self.n_rows_total, shard[0], shard[1])
        haps: List[HostHaplotype] = []
        for (name, off, end), (_n, _o, own) in zip(spans[self.row_lo:self.row_hi], self.rows[self.row_lo:self.row_hi]):
            seq = contigs[name]
            arr = np.frombuffer(seq.encode("ascii"), dtype=np.uint8) if isinstance(seq, str) else np.frombuffer(seq, dtype=np.uint8)
            haps.append(HostHaplotype(arr[off:end], PosSegments.identity(off, end - off), True, (0, own)))
        self.ds = None
        self._spans = spans[self.row_lo:self.row_hi]  # (contig, offset, end) of this rank's rows
        self._piece, self._lens = piece, {name: len(seq) for name, seq in contigs.items()}
        self._meta_guidelen = guidelen
        if haps:  # a rank may own no row of a tiny genome
            self.ds = DeviceHapSet(haps, device)
            _lib.check(self.ds._L.hawk_genome_finalize(self.ds._h), "hawk_genome_finalize")
        self.last_timing = None

    def _set_window(self, guidelen: int) -> None:
        """Which window starts every row owns depends on the window length (a window must fit its row): scans with another
        spacer length - the sites of bulged alignments - get their own scan ranges (hawk_hapset_set_meta; planes untouched)."""
        if guidelen == self._meta_guidelen or self.ds is None:
            return
        if not (1 <= guidelen <= self.guidelen + self.max_bulge):
            raise ValueError(f"window of {guidelen} + {self.pamlen} bases: the index was built for spacers of up to {self.guidelen + self.max_bulge}")
        L = guidelen + self.pamlen
        haps = []
        for name, off, end in self._spans:
            own = max(0, min(self._piece, self._lens[name] - off, end - off - L + 1))
            haps.append(HostHaplotype(b"", PosSegments.identity(off, end - off), True, (0, own)))
        self.ds.set_meta(haps)
        self._meta_guidelen = guidelen

    def scan_arrays(self, guides: Sequence[str], pam, right: bool, max_mm: int, cap: int = 1 << 20, guidelen: Optional[int] = None):
        """One hawk_offtarget_scan over this rank's rows: ({guide, row (global), q, strand, mm, code, nmask} arrays, timing).
        `guidelen` (default: the index's): the spacer length of this scan's guides - bulged alignments are searched as
        mismatch-only scans of derived guides that are shorter or longer than the guides themselves (scan_bulges)."""
        guidelen = self.guidelen if guidelen is None else int(guidelen)
        self._set_window(guidelen)
        empty = dict(guide=np.zeros(0, np.uint32), row=np.zeros(0, np.uint32), q=np.zeros(0, np.uint32), strand=np.zeros(0, np.uint8),
                     mm=np.zeros(0, np.uint8), code=np.zeros(0, np.uint64), nmask=np.zeros(0, np.uint32))
        if self.ds is None:
            return empty, dict(scan_ms=0.0, sites_ms=0.0, match_ms=0.0, total_ms=0.0, n_sites=0, scanned_positions=0)
        L = self.ds._L
        g2 = encode_guides(guides)
        par = _lib.OtParams(pam.bits, pam.bitsrc, len(pam), guidelen, int(bool(right)), max_mm)
        while True:
            og = np.empty(cap, np.uint32); orow = np.empty(cap, np.uint32); oq = np.empty(cap, np.uint32)
            ost = np.empty(cap, np.uint8); omm = np.empty(cap, np.uint8); oc = np.empty(cap, np.uint64); onm = np.empty(cap, np.uint32)
            n = C.c_uint64(0)
            tm = _lib.OtTiming()
            rc = L.hawk_offtarget_scan(self.ds._h, C.byref(par), _p(g2), len(g2), _p(og), _p(orow), _p(oq), _p(ost), _p(omm),
                                       _p(oc), _p(onm), C.c_uint64(cap), C.byref(n), C.byref(tm))
            if rc == _lib.HAWK_E_CAPACITY:
                cap = int(n.value) + 1024
                continue
            _lib.check(rc, "hawk_offtarget_scan")
            break
        self.last_timing = {k: getattr(tm, k) for k, _ in tm._fields_}
        k = int(n.value)
        hits = dict(guide=og[:k].copy(), row=(orow[:k] + np.uint32(self.row_lo)), q=oq[:k].copy(), strand=ost[:k].copy(), mm=omm[:k].copy(),
                    code=oc[:k].copy(), nmask=onm[:k].copy())
        return hits, self.last_timing

    # ---- bulged sites (the -bDNA / -bRNA arguments of the reference's CRISPRitz call, offtargets.py:264-268) -------------
    def scan_bulges(self, guides: Sequence[str], pam, right: bool, max_mm: int, bdna: int, brna: int, cap: int = 1 << 20) -> List["BulgeHit"]:
        """Sites that pair with a guide once `b` bases are bulged out - of the DNA (the site's spacer is b bases longer, b <= bdna)
        or of the RNA (b bases shorter, b <= brna) - with at most `max_mm` mismatches among the paired bases.  A bulged alignment
        is a mismatch-only alignment of a DERIVED guide: the guide with b interior bases deleted (RNA bulge), or with b bases
        inserted between its bases, every base tried (DNA bulge); each family of derived guides goes through hawk_offtarget_scan
        with its own spacer length, and the host keeps, per (guide, site, type, size), the placement with the fewest mismatches
        (ties: the lexicographically smallest bulge positions) - the definitions of oracle/hawk_oracle.c: ora_offtargets_bulges.
        Bulges of up to 2 bases are enumerated (CRISPRitz's own limit)."""
        if not (0 <= bdna <= 2 and 0 <= brna <= 2):
            raise ValueError("bulges of 0..2 bases are enumerated")
        if bdna > self.max_bulge:
            raise ValueError(f"the index was built for DNA bulges of up to {self.max_bulge} bases (GenomeIndex(max_bulge=...))")
        G = self.guidelen
        guides = [g.upper() for g in guides]
        out: List[BulgeHit] = []
        for dna, bmax in ((True, bdna), (False, brna)):
            for b in range(1, bmax + 1):
                derived, owner, gaps = _derived_guides(guides, G, b, dna)
                if not derived:
                    continue
                Gs = G + b if dna else G - b
                h, _tm = self.scan_arrays(derived, pam, right, max_mm, cap, guidelen=Gs)
                out += self._bulge_rows(h, guides, np.asarray(owner), np.asarray(gaps, dtype=np.uint64), Gs, b, dna, right, max_mm)
        self._set_window(self.guidelen)
        out.sort(key=lambda r: (r.guide, r.bulge_type, r.bulge_size, self._contig_rank(r.contig), r.position, r.strand == "-"))
        return out

    def _contig_rank(self, name: str) -> int:
        m = getattr(self, "_crank", None)
        if m is None:
            m = self._crank = {n: i for i, n in enumerate(dict.fromkeys(r[0] for r in self.rows))}
        return m[name]

    def _bulge_rows(self, h, guides, owner, gaps, Gs: int, b: int, dna: bool, right: bool, max_mm: int) -> List["BulgeHit"]:
        n = len(h["guide"])
        if n == 0:
            return []
        G, P = self.guidelen, self.pamlen
        L = Gs + P
        # the windows as bytes [n, L] (guide orientation, N for ambiguous bases) and their spacers
        sh = (2 * np.arange(L, dtype=np.uint64))[None, :]
        codes = ((h["code"][:, None] >> sh) & np.uint64(3)).astype(np.uint8)
        amb = ((h["nmask"][:, None].astype(np.uint64) >> np.arange(L, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)
        win = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
        win[amb] = ord("N")
        site = win[:, P:] if right else win[:, :Gs]
        g_of = owner[h["guide"]]
        gp = gaps[h["guide"]]
        gmat = np.frombuffer("".join(guides).encode("ascii"), dtype=np.uint8).reshape(len(guides), G)[g_of]  # [n, G]
        # paired positions: site position i <-> guide position j, skipping the bulged ones
        span = Gs if dna else G
        gapbits = ((gp[:, None] >> np.arange(span, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(bool)   # [n, span]
        keep = ~gapbits
        idx = np.argsort(~keep, axis=1, kind="stable")[:, : span - b]  # the span - b unbulged positions, ascending
        if dna:
            s_al, g_al = np.take_along_axis(site, idx, axis=1), gmat
        else:
            s_al, g_al = site, np.take_along_axis(gmat, idx, axis=1)
        mism = (s_al != g_al) | (s_al == ord("N"))
        mm = mism.sum(axis=1)
        ok = mm <= max_mm
        if dna:  # a bulged base is a definite base
            ok &= ~((site == ord("N")) & gapbits).any(axis=1)
        # placement order: the tuple of bulge positions, ascending
        pos_sorted = np.sort(np.where(gapbits, np.arange(span)[None, :], 1 << 20), axis=1)[:, :b]
        rank = np.zeros(n, dtype=np.int64)
        for k in range(b):
            rank = rank * 64 + pos_sorted[:, k]
        sel = np.flatnonzero(ok)
        if len(sel) == 0:
            return []
        order = sel[np.lexsort((rank[sel], mm[sel], h["strand"][sel], h["q"][sel], h["row"][sel], g_of[sel]))]
        key = np.stack([g_of[order].astype(np.int64), h["row"][order].astype(np.int64), h["q"][order].astype(np.int64), h["strand"][order].astype(np.int64)], axis=1)
        first = np.ones(len(order), dtype=bool)
        first[1:] = (key[1:] != key[:-1]).any(axis=1)
        rows = []
        for i in order[first].tolist():
            name, off, _ = self.rows[int(h["row"][i])]
            g = int(g_of[i])
            sp_site = site[i].tobytes().decode("ascii")
            pam_site = (win[i, :P] if right else win[i, Gs:]).tobytes().decode("ascii")
            gbits = int(gp[i])
            cr, dn = [], []
            si = gi = 0
            while si < Gs or gi < G:
                if dna and si < Gs and (gbits >> si) & 1:
                    cr.append("-"); dn.append(sp_site[si]); si += 1
                elif (not dna) and gi < G and (gbits >> gi) & 1:
                    cr.append(guides[g][gi]); dn.append("-"); gi += 1
                else:
                    t, q = sp_site[si], guides[g][gi]
                    cr.append(q); dn.append(t if t == q else t.lower())
                    si += 1; gi += 1
            rows.append(BulgeHit(g, name, off + int(h["q"][i]), "-" if h["strand"][i] else "+", int(mm[i]), "DNA" if dna else "RNA", b,
                                 "".join(cr), "".join(dn), pam_site, gbits))
        return rows

    def hits_from_arrays(self, h) -> List["OffTargetHit"]:
        """Arrays of scan_arrays (of this rank, or gathered from every rank) -> sorted OffTargetHit list."""
        order = np.lexsort((h["strand"], h["q"], h["row"], h["guide"]))
        out = []
        for i in order:
            name, off, _ = self.rows[int(h["row"][i])]
            out.append(OffTargetHit(int(h["guide"][i]), name, off + int(h["q"][i]), "-" if h["strand"][i] else "+", int(h["mm"][i]),
                                    decode_window(int(h["code"][i]), int(h["nmask"][i]), self.L)))
        return out

    def scan(self, guides: Sequence[str], pam, right: bool, max_mm: int, cap: int = 1 << 20, comm=None) -> List[OffTargetHit]:
        """All windows within ``max_mm`` mismatches of any guide, both strands, sorted by
        (guide, contig order, position, strand).  With a communicator (parallel.RcclComm / TcpComm) the hits of every
        rank's genome shard are gathered to rank 0, which returns the whole list (other ranks return [])."""
        hits, _tm = self.scan_arrays(guides, pam, right, max_mm, cap)
        if comm is not None and comm.world > 1:
            parts = {k: comm.gatherv_bytes(v, 0) for k, v in hits.items()}
            if comm.rank != 0:
                return []
            hits = {k: np.concatenate(v) for k, v in parts.items()}
        return self.hits_from_arrays(hits)
