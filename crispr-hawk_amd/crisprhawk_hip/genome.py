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


class GenomeIndex:
    """`shard = (rank, world)`: the genome's rows (pieces) are block-partitioned over the ranks of a multi-GPU job
    ("shard the genome by contig/offset across ranks, guides replicated, rows gathered"); every rank keeps the
    descriptors of ALL rows, so a hit's global row index names its contig and offset anywhere."""

    def __init__(self, contigs: Dict[str, object], guidelen: int, pamlen: int, piece: int = 1 << 22, device: Optional[int] = None,
                 shard: Optional[Tuple[int, int]] = None):
        self.L = guidelen + pamlen
        self.guidelen, self.pamlen = guidelen, pamlen
        overlap = self.L - 1
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
        if haps:  # a rank may own no row of a tiny genome
            self.ds = DeviceHapSet(haps, device)
            _lib.check(self.ds._L.hawk_genome_finalize(self.ds._h), "hawk_genome_finalize")
        self.last_timing = None

    def scan_arrays(self, guides: Sequence[str], pam, right: bool, max_mm: int, cap: int = 1 << 20):
        """One hawk_offtarget_scan over this rank's rows: ({guide, row (global), q, strand, mm, code, nmask} arrays, timing)."""
        empty = dict(guide=np.zeros(0, np.uint32), row=np.zeros(0, np.uint32), q=np.zeros(0, np.uint32), strand=np.zeros(0, np.uint8),
                     mm=np.zeros(0, np.uint8), code=np.zeros(0, np.uint64), nmask=np.zeros(0, np.uint32))
        if self.ds is None:
            return empty, dict(scan_ms=0.0, sites_ms=0.0, match_ms=0.0, total_ms=0.0, n_sites=0, scanned_positions=0)
        L = self.ds._L
        g2 = encode_guides(guides)
        par = _lib.OtParams(pam.bits, pam.bitsrc, len(pam), self.guidelen, int(bool(right)), max_mm)
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
