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


class VCF:
    def __init__(self, fname: str, verbosity: int = 0, debug: bool = True, vcfidx: Optional[str] = "") -> None:
        self._debug, self._verbosity = debug, verbosity
        if not os.path.isfile(fname):
            exception_handler(FileNotFoundError, f"Cannot find input VCF {fname}", os.EX_DATAERR, debug)
        self._fname = fname
        with open(fname, "rb") as f:
            magic = f.read(2)
        opener = gzip.open if magic == b"\x1f\x8b" else open
        with opener(fname, "rb") as f:
            raw = f.read()
        buf = np.frombuffer(raw, dtype=np.uint8)
        nl = np.flatnonzero(buf == 10)
        if len(buf) and (len(nl) == 0 or nl[-1] != len(buf) - 1):  # unterminated last line
            buf = np.concatenate((buf, np.array([10], np.uint8)))
            nl = np.flatnonzero(buf == 10)
        starts = np.concatenate(([0], nl[:-1] + 1)).astype(np.int64) if len(nl) else np.zeros(0, np.int64)
        is_hdr = buf[starts] == ord("#") if len(starts) else np.zeros(0, bool)
        header = [bytes(buf[s:e]).decode() for s, e in zip(starts[is_hdr], nl[is_hdr])]
        if not header or not header[-1].startswith("#CHROM"):
            exception_handler(ValueError, f"Input VCF {fname} has no #CHROM header line", os.EX_DATAERR, debug)
        self._samples = header[-1].strip().split()[9:]  # variant.py:657
        self._buf = buf
        body = ~is_hdr & (nl - starts > 0)
        self._starts, self._ends = starts[body], nl[body] + 1
        # CHROM and POS of every record: the first two tab-separated fields
        self._pos = np.zeros(len(self._starts), dtype=np.int64)
        contigs = set()
        for i, s in enumerate(self._starts):
            head = bytes(buf[s:s + 64]).split(b"\t", 2)
            contigs.add(head[0].decode())
            self._pos[i] = int(head[1])
        if len(contigs) > 1:  # variant.py:650-656: one contig per VCF
            exception_handler(ValueError, f"Input VCF {fname} store variants belonging to multiple contigs", os.EX_DATAERR, debug)
        self._contig = next(iter(contigs)) if contigs else ""
        if len(self._pos) > 1 and np.any(np.diff(self._pos) < 0):
            exception_handler(ValueError, f"Input VCF {fname} is not sorted by position", os.EX_DATAERR, debug)
        self._phased = False  # variant.py:700-708: decided by the first record's first genotype
        if len(self._starts):
            first = bytes(buf[self._starts[0]:self._ends[0]]).decode().strip().split()
            self._phased = len(first) > 9 and "|" in first[9]

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
        for s, e in zip(self._starts[a:b], self._ends[a:b]):
            v = VariantRecord(self._debug)
            v.read_vcf_line(bytes(self._buf[s:e]).decode().strip().split(), self._samples, self._phased)
            out.append(v)
        return out

    def fetch_block(self, coordinate: Coordinate) -> VcfBlock:
        a, b = self._range(coordinate)
        if a == b:
            return VcfBlock(np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(0, np.uint64), [])
        s0, e1 = int(self._starts[a]), int(self._ends[b - 1])
        text = self._buf[s0:e1]
        line_off = np.concatenate((self._starts[a:b] - s0, [e1 - s0])).astype(np.uint64)
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
    if compress:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)
