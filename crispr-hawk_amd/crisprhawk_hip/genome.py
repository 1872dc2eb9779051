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
    out = np.zeros(len(guides), dtype=np.uint64)
    lut = {"A": 0, "C": 1, "G": 2, "T": 3}
    for k, g in enumerate(guides):
        v = 0
        for i, c in enumerate(g.upper()):
            if c not in lut:
                raise ValueError(f"guide {g!r} holds a non-ACGT base")
            v |= lut[c] << (2 * i)
        out[k] = v
    return out


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
    def __init__(self, contigs: Dict[str, str], guidelen: int, pamlen: int, piece: int = 1 << 22, device: Optional[int] = None):
        self.L = guidelen + pamlen
        self.guidelen, self.pamlen = guidelen, pamlen
        overlap = self.L - 1
        self.rows: List[Tuple[str, int, int]] = []  # (contig, offset, owned window starts)
        haps: List[HostHaplotype] = []
        self.total = 0
        for name, seq in contigs.items():
            n = len(seq)
            self.total += n
            arr = np.frombuffer(seq.encode("ascii") if isinstance(seq, str) else seq, dtype=np.uint8)
            for off in range(0, max(n, 1), piece):
                end = min(n, off + piece + overlap)
                if end - off < self.L:
                    continue
                own = min(piece, n - off)  # window starts [0, own) belong to this row ...
                own = min(own, end - off - self.L + 1)  # ... as far as the window fits
                haps.append(HostHaplotype(arr[off:end], PosSegments.identity(off, end - off), True, (0, own)))
                self.rows.append((name, off, own))
        if not haps:
            raise ValueError("genome shorter than one guide+PAM window")
        self.ds = DeviceHapSet(haps, device)
        _lib.check(self.ds._L.hawk_genome_finalize(self.ds._h), "hawk_genome_finalize")
        self.last_timing = None

    def scan(self, guides: Sequence[str], pam, right: bool, max_mm: int, cap: int = 1 << 20) -> List[OffTargetHit]:
        """All windows within ``max_mm`` mismatches of any guide, both strands, sorted by
        (guide, contig order, position, strand)."""
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
        order = np.lexsort((ost[:k], oq[:k], orow[:k], og[:k]))
        hits = []
        for i in order:
            name, off, _ = self.rows[int(orow[i])]
            hits.append(OffTargetHit(int(og[i]), name, off + int(oq[i]), "-" if ost[i] else "+", int(omm[i]),
                                     decode_window(int(oc[i]), int(onm[i]), self.L)))
        return hits
