"""pysam-free FASTA / BED / VCF readers with the interface the reference's pipeline uses.

Reference classes: `Fasta` (`sequence.py:183-360`, pysam.FastaFile + samtools faidx), `Bed`
(`bedfile.py:27-187`), `VCF` (`variant.py:622-830`, pysam.TabixFile).  Here:

* `Fasta` reads / writes the 5-column `.fai` index itself and fetches by seek arithmetic;
* `Bed` parses `chrom start stop [...]` lines into padded `Coordinate`s;
* `VCF` handles plain, gzip and bgzip text (bgzip is a sequence of gzip members, which `gzip` reads).  Instead
  of a tabix index it keeps one in-memory scan of the body: line offsets and POS per record, so a region fetch is
  a binary search.  `fetch()` returns `VariantRecord`s like the reference; `fetch_block()` returns the raw text of
  the records in range plus per-line offsets - the input of the device genotype parser (`hawk_vcf_genotypes`),
  which turns the sample columns into an allele-code matrix without any per-sample Python work.

Error behaviour follows the reference: `exception_handler(exc, message, code, debug)`.
"""
import gzip
import os
from typing import Dict, List, Optional, Tuple

import numpy as np

from .coordinate import Coordinate
from .exception_handlers import exception_handler
from .region import Region, RegionList
from .sequence import Sequence
from .variant import VariantRecord

FAI = "fai"


# ------------------------------------------------------------------------------------------- FASTA
class Fasta:
    """One-contig-per-file FASTA as the reference uses it (`sequence.py:183-360`); multi-contig files work too
    (`contig` is then the first one, `fetch` takes any)."""

    def __init__(self, fname: str, verbosity: int = 0, debug: bool = True, faidx: Optional[str] = "") -> None:
        self._debug, self._verbosity = debug, verbosity
        if not os.path.isfile(fname):
            exception_handler(FileNotFoundError, f"Cannot find input FASTA {fname}", os.EX_DATAERR, debug)
        self._fname = fname
        self._faidx = self._search_index(faidx)
        if not self._faidx:
            self.index_fasta()
        self._index = self._read_index()
        self._contig = next(iter(self._index))

    def _search_index(self, faidx: Optional[str]) -> str:
        if not faidx:
            cand = f"{os.path.abspath(self._fname)}.{FAI}"
            return cand if os.path.isfile(cand) and os.stat(cand).st_size > 0 else ""
        if not (os.path.isfile(faidx) and os.stat(faidx).st_size > 0):
            exception_handler(FileNotFoundError, f"Not existing or empty FASTA index {faidx}", os.EX_DATAERR, self._debug)
        return faidx

    def index_fasta(self) -> None:
        """Write `<fasta>.fai` (NAME LENGTH OFFSET LINEBASES LINEWIDTH, the samtools faidx layout)."""
        rows, name, length, offset, lb, lw, pos = [], None, 0, 0, 0, 0, 0
        with open(self._fname, "rb") as f:
            for line in f:
                if line.startswith(b">"):
                    if name is not None:
                        rows.append((name, length, offset, lb, lw))
                    name, length, lb, lw = line[1:].split()[0].decode(), 0, 0, 0
                    offset = pos + len(line)
                else:
                    body = line.rstrip(b"\r\n")
                    if lb == 0 and body:
                        lb, lw = len(body), len(line)
                    length += len(body)
                pos += len(line)
        if name is not None:
            rows.append((name, length, offset, lb, lw))
        if not rows:
            exception_handler(RuntimeError, f"An error occurred while indexing {self._fname}", os.EX_SOFTWARE, self._debug)
        self._faidx = f"{self._fname}.{FAI}"
        with open(self._faidx, "w") as out:
            for r in rows:
                out.write("\t".join(str(x) for x in r) + "\n")

    def _read_index(self) -> Dict[str, Tuple[int, int, int, int]]:
        idx = {}
        with open(self._faidx) as f:
            for line in f:
                c = line.rstrip("\n").split("\t")
                if len(c) >= 5:
                    idx[c[0]] = (int(c[1]), int(c[2]), int(c[3]), int(c[4]))
        return idx

    def fetch_str(self, contig: str, start0: int, stop: int) -> str:
        """Bases [start0, stop) of `contig`, 0-based half-open, clipped to the contig like pysam's fetch."""
        length, offset, lb, lw = self._index[contig]
        start0, stop = max(0, start0), min(stop, length)
        if stop <= start0:
            return ""
        first = offset + (start0 // lb) * lw + start0 % lb
        last = offset + ((stop - 1) // lb) * lw + (stop - 1) % lb + 1
        with open(self._fname, "rb") as f:
            f.seek(first)
            raw = f.read(last - first)
        return raw.replace(b"\n", b"").replace(b"\r", b"").decode("ascii")

    def fetch(self, coord: Coordinate) -> Sequence:  # sequence.py:318-343: FASTA fetch is [start - 1, stop)
        if coord.contig not in self._index:
            exception_handler(ValueError, f"Input contig ({coord.contig}) not available in {self._fname}", os.EX_DATAERR, self._debug)
        return Sequence(self.fetch_str(coord.contig, coord.start - 1, coord.stop).strip(), self._debug)

    fname = property(lambda self: self._fname)
    contig = property(lambda self: self._contig)


def write_fasta(path: str, contig: str, sequence: str, width: int = 60) -> None:
    with open(path, "w") as f:
        f.write(f">{contig}\n")
        for i in range(0, len(sequence), width):
            f.write(sequence[i:i + width] + "\n")


# --------------------------------------------------------------------------------------------- BED
def _parse_bed_line(bedline: str, linenum: int, padding: int, debug: bool) -> Coordinate:  # bedfile.py:190-230
    columns = bedline.strip().split()
    if len(columns) < 3:
        exception_handler(ValueError, f"Less than three columns at line {linenum}", os.EX_DATAERR, debug)
    try:
        chrom, start, stop = columns[0], int(columns[1]), int(columns[2])
    except ValueError as e:
        exception_handler(TypeError, f"Start/stop values at line {linenum} are not int", os.EX_DATAERR, debug, e)
    if stop < start:
        exception_handler(ValueError, f"Stop < start coordinate ({stop} < {start}) at line {linenum}", os.EX_DATAERR, debug)
    return Coordinate(chrom, start, stop, padding)


class Bed:
    def __init__(self, bedfile: str, padding: int, debug: bool = True) -> None:
        self._debug = debug
        if not os.path.isfile(bedfile):
            exception_handler(FileNotFoundError, f"Cannot find input BED file {bedfile}", os.EX_DATAERR, debug)
        self._fname = bedfile
        with open(bedfile) as f:
            self._coordinates = [_parse_bed_line(line, i + 1, padding, debug) for i, line in enumerate(f)
                                 if not line.startswith("#") and line.strip()]

    def __len__(self) -> int:
        return len(self._coordinates)

    def __iter__(self):
        return iter(self._coordinates)

    def __getitem__(self, idx):
        return self._coordinates[idx]

    def extract_regions(self, fastas: Dict[str, Fasta]) -> RegionList:  # bedfile.py:163-169
        return RegionList([Region(fastas[c.contig].fetch(c), c) for c in self._coordinates])


# --------------------------------------------------------------------------------------------- VCF
class VcfBlock:
    """Raw text of consecutive VCF records plus where things are in it (input of the device genotype parser)."""

    def __init__(self, text: np.ndarray, line_off: np.ndarray, gt_off: np.ndarray, fixed: List[List[str]]):
        self.text = text          # uint8, the records' bytes, '\n'-terminated
        self.line_off = line_off  # uint64 [n + 1], record i = text[line_off[i]:line_off[i + 1]]
        self.gt_off = gt_off      # uint64 [n], offset of record i's first sample column
        self.fixed = fixed        # the 9 fixed columns of every record, split

    def __len__(self) -> int:
        return len(self.gt_off)


class _TextSource:
    """Random access to the (decompressed) bytes of a VCF without holding them: a plain text file is read by seek; a
    BGZF file (bgzip: independent <= 64 KB gzip members with their size in the header's BC field) by inflating only the
    blocks a byte range touches; a plain gzip stream has no random access and is held decompressed (the one case that
    costs memory).  `chunks()` streams the whole text once for the index pass."""

    CHUNK = 32 << 20

    def __init__(self, fname: str):
        self.fname = fname
        with open(fname, "rb") as f:
            head = f.read(18)
        self.kind = "plain"
        self._whole = None
        if head[:2] == b"\x1f\x8b":
            is_bgzf = len(head) >= 18 and head[3] & 4 and head[12:14] == b"BC"
            self.kind = "bgzf" if is_bgzf else "gzip"
        if self.kind == "bgzf":
            self._index_blocks()
        elif self.kind == "gzip":
            with gzip.open(fname, "rb") as f:
                self._whole = np.frombuffer(f.read(), dtype=np.uint8)

    def _index_blocks(self) -> None:
        c_off, u_off, u = [], [], 0
        with open(self.fname, "rb") as f:
            pos = 0
            while True:
                hdr = f.read(18)
                if len(hdr) < 18:
                    break
                bsize = int.from_bytes(hdr[16:18], "little") + 1
                f.seek(pos + bsize - 4)
                isize = int.from_bytes(f.read(4), "little")
                c_off.append(pos); u_off.append(u)
                u += isize
                pos += bsize
                f.seek(pos)
        self._c_off = np.array(c_off + [pos], dtype=np.int64)
        self._u_off = np.array(u_off + [u], dtype=np.int64)

    def _inflate(self, f, k: int) -> bytes:
        import zlib
        f.seek(int(self._c_off[k]))
        raw = f.read(int(self._c_off[k + 1] - self._c_off[k]))
        xlen = int.from_bytes(raw[10:12], "little")
        return zlib.decompress(raw[12 + xlen:-8], -15)

    def chunks(self):
        """(offset of the chunk in the text, uint8 array) over the whole text."""
        if self.kind == "gzip":
            yield 0, self._whole
        elif self.kind == "plain":
            with open(self.fname, "rb") as f:
                off = 0
                while True:
                    raw = f.read(self.CHUNK)
                    if not raw:
                        break
                    yield off, np.frombuffer(raw, dtype=np.uint8)
                    off += len(raw)
        else:
            with open(self.fname, "rb") as f:
                k, nb = 0, len(self._c_off) - 1
                while k < nb:
                    parts, off, size = [], int(self._u_off[k]), 0
                    while k < nb and size < self.CHUNK:
                        parts.append(self._inflate(f, k))
                        size += len(parts[-1])
                        k += 1
                    yield off, np.frombuffer(b"".join(parts), dtype=np.uint8)

    def read(self, lo: int, hi: int) -> np.ndarray:
        """Text bytes [lo, hi)."""
        if hi <= lo:
            return np.zeros(0, np.uint8)
        if self.kind == "gzip":
            return self._whole[lo:hi]
        if self.kind == "plain":
            with open(self.fname, "rb") as f:
                f.seek(lo)
                return np.frombuffer(f.read(hi - lo), dtype=np.uint8)
        k0 = int(np.searchsorted(self._u_off, lo, side="right") - 1)
        k1 = int(np.searchsorted(self._u_off, hi, side="left"))
        with open(self.fname, "rb") as f:
            data = b"".join(self._inflate(f, k) for k in range(k0, k1))
        base = int(self._u_off[k0])
        return np.frombuffer(data, dtype=np.uint8)[lo - base:hi - base]


class VCF:
    """One-contig VCF (plain, gzip or bgzip) with position-indexed access, standing in for the pysam/tabix-backed class
    of variant.py:622-830.  Construction streams the file once and keeps only an index - 24 bytes per record (line start,
    line end, POS) - so a 2504-sample chromosome VCF (tens of GB of text) is never resident; a fetch reads just the bytes
    of the records it returns."""

    def __init__(self, fname: str, verbosity: int = 0, debug: bool = True, vcfidx: Optional[str] = "") -> None:
        self._debug, self._verbosity = debug, verbosity
        if not os.path.isfile(fname):
            exception_handler(FileNotFoundError, f"Cannot find input VCF {fname}", os.EX_DATAERR, debug)
        self._fname = fname
        self._src = _TextSource(fname)
        self._mm = self._gt_abs = None
        if self._src.kind == "plain" and self._index_mapped():
            return
        starts_l, ends_l, pos_l = [], [], []
        contigs = set()
        header_last = None
        carry = np.zeros(0, np.uint8)
        carry_off = 0
        total = 0
        for off, chunk in self._src.chunks():
            total = off + len(chunk)
            buf = np.concatenate((carry, chunk)) if len(carry) else chunk
            base = off - len(carry)
            nl = np.flatnonzero(buf == 10)
            if len(nl) == 0:
                carry, carry_off = buf, base
                continue
            starts = np.concatenate(([0], nl[:-1] + 1)).astype(np.int64)
            self._index_lines(buf, base, starts, nl, starts_l, ends_l, pos_l, contigs)
            hdr = starts[buf[starts] == ord("#")]
            if len(hdr):
                k = int(np.searchsorted(starts, hdr[-1]))
                header_last = bytes(buf[starts[k]:nl[k]]).decode()
            carry, carry_off = buf[nl[-1] + 1:].copy(), base + int(nl[-1]) + 1
        if len(carry):  # unterminated last line
            buf = np.concatenate((carry, np.array([10], np.uint8)))
            self._index_lines(buf, carry_off, np.zeros(1, np.int64), np.array([len(buf) - 1]), starts_l, ends_l, pos_l, contigs)
            if buf[0] == ord("#"):
                header_last = bytes(buf[:-1]).decode()
        if header_last is None or not header_last.startswith("#CHROM"):
            exception_handler(ValueError, f"Input VCF {fname} has no #CHROM header line", os.EX_DATAERR, debug)
        self._samples = header_last.strip().split()[9:]  # variant.py:657
        self._total = total
        self._starts = np.concatenate(starts_l) if starts_l else np.zeros(0, np.int64)
        self._ends = np.concatenate(ends_l) if ends_l else np.zeros(0, np.int64)
        self._pos = np.concatenate(pos_l) if pos_l else np.zeros(0, np.int64)
        if len(contigs) > 1:  # variant.py:650-656: one contig per VCF
            exception_handler(ValueError, f"Input VCF {fname} store variants belonging to multiple contigs", os.EX_DATAERR, debug)
        self._contig = next(iter(contigs)) if contigs else ""
        if len(self._pos) > 1 and np.any(np.diff(self._pos) < 0):
            exception_handler(ValueError, f"Input VCF {fname} is not sorted by position", os.EX_DATAERR, debug)
        self._phased = False  # variant.py:700-708: decided by the first record's first genotype
        if len(self._starts):
            first = bytes(self._read(int(self._starts[0]), int(self._ends[0]))).decode().strip().split()
            self._phased = len(first) > 9 and "|" in first[9]

    def _index_mapped(self) -> bool:
        """The index pass of a plain-text VCF by the library's host helper (hawk_host_vcf_index: the file mapped, every byte
        walked once by all cores) instead of the numpy stream below - same index, plus where every record's sample columns begin,
        so that fetch_block hands out views of the mapping.  False (nothing set) when the helper does not apply: an empty file,
        no trailing newline, a library without the helper."""
        import ctypes as C
        size = os.path.getsize(self._fname)
        if size == 0:
            return False
        try:
            from . import _lib
            L = _lib.lib()
            fn = L.hawk_host_vcf_index
        except Exception:
            return False
        mm = np.memmap(self._fname, dtype=np.uint8, mode="r")
        if mm[-1] != 10:
            return False
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        n, multi = C.c_uint64(0), C.c_uint32(0)
        rc = fn(p(mm), C.c_uint64(size), C.c_uint64(0), None, None, None, None, C.byref(n), C.byref(multi))
        if rc not in (_lib.HAWK_OK, _lib.HAWK_E_CAPACITY):
            return False
        nl = int(n.value)
        ls = np.zeros(nl + 1, dtype=np.uint64)
        pos = np.zeros(nl, dtype=np.int64)
        gt = np.zeros(nl, dtype=np.uint64)
        cl = np.zeros(nl, dtype=np.uint32)
        _lib.check(fn(p(mm), C.c_uint64(size), C.c_uint64(nl), p(ls), p(pos), p(gt), p(cl), C.byref(n), C.byref(multi)), "hawk_host_vcf_index")
        bad = np.flatnonzero(pos == -2)
        if len(bad):
            exception_handler(ValueError, f"Malformed VCF record near byte {int(ls[bad[0]])} of {self._fname}", os.EX_DATAERR, self._debug)
        hdr = np.flatnonzero((pos == -1) & (ls[1:] - ls[:-1] > 1))
        header_last = bytes(mm[int(ls[hdr[-1]]):int(ls[hdr[-1] + 1]) - 1]).decode() if len(hdr) else None
        if header_last is None or not header_last.startswith("#CHROM"):
            exception_handler(ValueError, f"Input VCF {self._fname} has no #CHROM header line", os.EX_DATAERR, self._debug)
        self._samples = header_last.strip().split()[9:]  # variant.py:657
        body = np.flatnonzero(pos >= 0)
        self._total = size
        self._starts, self._ends, self._pos = ls[body].astype(np.int64), ls[body + 1].astype(np.int64), pos[body]
        self._gt_abs = gt[body]
        if multi.value:  # variant.py:650-656: one contig per VCF
            exception_handler(ValueError, f"Input VCF {self._fname} store variants belonging to multiple contigs", os.EX_DATAERR, self._debug)
        self._contig = bytes(mm[int(ls[body[0]]):int(ls[body[0]]) + int(cl[body[0]])]).decode() if len(body) else ""
        if len(self._pos) > 1 and np.any(np.diff(self._pos) < 0):
            exception_handler(ValueError, f"Input VCF {self._fname} is not sorted by position", os.EX_DATAERR, self._debug)
        self._phased = False  # variant.py:700-708: decided by the first record's first genotype
        if len(self._starts):
            first = bytes(mm[int(self._starts[0]):int(self._ends[0])]).decode().strip().split()
            self._phased = len(first) > 9 and "|" in first[9]
        self._mm = mm
        return True

    def _index_lines(self, buf, base, starts, nl, starts_l, ends_l, pos_l, contigs) -> None:
        """CHROM and POS of every body line of a chunk, without a Python loop: the first 64 bytes of each line as a matrix,
        the first two tabs by argmax, POS from its digit columns."""
        body = (buf[starts] != ord("#")) & (nl - starts > 0)
        st, en = starts[body], nl[body] + 1
        if len(st) == 0:
            return
        W = 64
        idx = np.minimum(st[:, None] + np.arange(W)[None, :], len(buf) - 1)
        head = buf[idx]
        # only the line's own bytes count: a short line must not borrow the tabs of the line behind it
        istab = (head == 9) & (np.arange(W)[None, :] < (en - st)[:, None])
        t1 = istab.argmax(axis=1)
        rest = istab.copy()
        rest[np.arange(len(st)), t1] = False
        t2 = rest.argmax(axis=1)
        ok = istab.any(axis=1) & rest.any(axis=1) & (t2 > t1 + 1)
        if not ok.all():
            exception_handler(ValueError, f"Malformed VCF record near byte {base + int(st[np.flatnonzero(~ok)[0]])} of {self._fname}",
                              os.EX_DATAERR, self._debug)
        col = np.arange(W)[None, :]
        digit = (col > t1[:, None]) & (col < t2[:, None])
        vals = np.where(digit, head.astype(np.int64) - 48, 0)
        if np.any(digit & ((vals < 0) | (vals > 9))):
            exception_handler(ValueError, f"Non-numeric POS in {self._fname}", os.EX_DATAERR, self._debug)
        power = np.where(digit, 10 ** np.clip(t2[:, None] - 1 - col, 0, 18), 0)
        pos_l.append((vals * power).sum(axis=1))
        starts_l.append(st + base)
        ends_l.append(en + base)
        # contig names: distinct first fields (one per VCF as a rule: compare against the first line's)
        first = bytes(head[0, :t1[0]])
        same = (t1 == t1[0]) & (head[:, :t1[0]] == head[0, :t1[0]]).all(axis=1)
        contigs.add(first.decode())
        for i in np.flatnonzero(~same).tolist():
            contigs.add(bytes(head[i, :t1[i]]).decode())

    def _read(self, lo: int, hi: int) -> np.ndarray:
        data = self._src.read(lo, min(hi, self._total))
        if hi > self._total:  # the file's last line had no newline
            data = np.concatenate((data, np.full(hi - self._total, 10, np.uint8)))
        return data

    def _range(self, coordinate: Coordinate) -> Tuple[int, int]:
        if self._contig != coordinate.contig and self._contig.replace("chr", "") != coordinate.contig.replace("chr", ""):
            exception_handler(ValueError, f"Mismatching VCF and coordinate contigs ({self._contig} - {coordinate.contig})",
                              os.EX_DATAERR, self._debug)
        # tabix fetch(contig, start, stop) is 0-based half-open: records with start < POS <= stop
        a = int(np.searchsorted(self._pos, coordinate.start, side="right"))
        b = int(np.searchsorted(self._pos, coordinate.stop, side="right"))
        return a, b

    def fetch(self, coordinate: Coordinate) -> List[VariantRecord]:  # variant.py:710-760
        a, b = self._range(coordinate)
        out = []
        if a == b:
            return out
        s0 = int(self._starts[a])
        text = self._read(s0, int(self._ends[b - 1]))
        for s, e in zip(self._starts[a:b], self._ends[a:b]):
            v = VariantRecord(self._debug)
            v.read_vcf_line(bytes(text[int(s) - s0:int(e) - s0]).decode().strip().split(), self._samples, self._phased)
            out.append(v)
        return out

    def fetch_block(self, coordinate: Coordinate) -> VcfBlock:
        a, b = self._range(coordinate)
        if a == b:
            return VcfBlock(np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(0, np.uint64), [])
        s0, e1 = int(self._starts[a]), int(self._ends[b - 1])
        line_off = np.concatenate((self._starts[a:b] - s0, [e1 - s0])).astype(np.uint64)
        if self._mm is not None:  # a view of the mapping; the sample columns' offsets are in the index already
            text = self._mm[s0:e1]
            gabs = self._gt_abs[a:b]
            short = np.flatnonzero(gabs == 0)
            if len(short):
                exception_handler(ValueError, f"VCF record at byte {int(self._starts[a + short[0]])} has no sample columns", os.EX_DATAERR, self._debug)
            gt_off = (gabs - np.uint64(s0)).astype(np.uint64)
            lo_l, g_l = line_off[:-1].tolist(), gt_off.tolist()
            fixed = [bytes(text[lo:g - 1]).decode().split("\t") for lo, g in zip(lo_l, g_l)]
            return VcfBlock(text, line_off, gt_off, fixed)
        text = self._read(s0, e1)
        gt_off = np.zeros(b - a, dtype=np.uint64)
        fixed = []
        for i in range(b - a):
            lo, hi = int(line_off[i]), int(line_off[i + 1])
            # the 9th tab of the record ends the FORMAT column (records are short up to there)
            tabs = np.flatnonzero(text[lo:min(hi, lo + 4096)] == 9)
            if len(tabs) < 9:
                tabs = np.flatnonzero(text[lo:hi] == 9)
            if len(tabs) < 9:
                exception_handler(ValueError, f"VCF record at byte {s0 + lo} has no sample columns", os.EX_DATAERR, self._debug)
            gt_off[i] = lo + int(tabs[8]) + 1
            fixed.append(bytes(text[lo:lo + int(tabs[8])]).decode().split("\t"))
        return VcfBlock(text, line_off, gt_off, fixed)

    @property
    def contig(self) -> str:  # variant.py:762-764
        return self._contig if self._contig.startswith("chr") else f"chr{self._contig}"

    phased = property(lambda self: self._phased)
    samples = property(lambda self: self._samples)


def write_vcf(path: str, contig: str, samples: List[str], rows: List[List[str]], compress: bool = False) -> None:
    """rows: tab-split VCF records (9 fixed columns + one genotype per sample)."""
    head = ["##fileformat=VCFv4.2", f"##contig=<ID={contig}>", '##INFO=<ID=AF,Number=A,Type=Float,Description="Allele frequency">',
            '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
            "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples)]
    data = ("\n".join(head + ["\t".join(r) for r in rows]) + "\n").encode()
    if compress == "bgzf":
        write_bgzf(path, data)
    elif compress:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def write_bgzf(path: str, data: bytes, block: int = 0xff00) -> None:
    """bgzip's container: independent deflate members of <= 64 KB of text each, the member size in a BC extra field,
    an empty member as the end marker (SAM spec §4.1)."""
    import struct
    import zlib

    def member(chunk: bytes) -> bytes:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        return (b"\x1f\x8b\x08\x04" + b"\x00" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) + body +
                struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    with open(path, "wb") as f:
        for i in range(0, len(data), block):
            f.write(member(data[i:i + block]))
        f.write(member(b""))
