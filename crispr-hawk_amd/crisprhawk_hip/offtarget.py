"""Offtarget — one off-target site of one guide (reference offtarget.py:22-239).

The reference builds it from a CRISPRitz `targets.txt` row (fields [0] bulge type, [1] crRNA with the PAM as N's,
[2] DNA with the observed PAM and mismatches in lower case, [3] chrom, [4] position, [6] strand, [7] mismatches,
[8] bulge size).  The GPU scan (hawk_offtarget_scan) produces those fields directly, so here an Offtarget is a record
of them (`from_hit`); the report-line constructor is kept because it is the reference's signature.  Everything the
report needs is derived on demand from the two PAM-bearing strings:

    grna_ / spacer_   the spacer alone (guide / genomic site)
    grna              spacer re-joined with the NOMINAL PAM          (offtarget.py:93-95)
    spacer            site spacer re-joined with its OBSERVED PAM    (offtarget.py:96-98)
"""
import os
from typing import NamedTuple, Tuple

import numpy as np

from .exception_handlers import exception_handler
from .utils import round_score


def split_pam(sequence: str, pamlen: int, right: bool) -> Tuple[str, str]:
    """(spacer, PAM) of a spacer+PAM string: the PAM leads when the guide sits to its right (Cpf1-like)."""
    cut = pamlen if right else len(sequence) - pamlen
    head, tail = sequence[:cut], sequence[cut:]
    return (tail, head) if right else (head, tail)


def join_pam(spacer: str, pam: str, right: bool) -> str:
    return pam + spacer if right else spacer + pam


# the two helper names the reference's unit tests exercise (tests/test_offtarget.py:25-50)
def _retrieve_pam(sequence: str, length: int, right: bool) -> str:
    return split_pam(sequence, length, right)[1]


def _format_sequence(sequence: str, pam: str, right: bool) -> Tuple[str, str]:
    spacer = split_pam(sequence, len(pam), right)[0]
    return spacer, join_pam(spacer, pam, right)


class _Site(NamedTuple):
    bulge_type: str
    crrna: str      # guide spacer + PAM placeholder, as searched
    dna: str        # genomic site + observed PAM, mismatches lower-case
    chrom: str
    position: int
    strand: str
    mm: int
    bulge_size: int


class Offtarget:
    __slots__ = ("_site", "_pam", "_right", "_debug", "_cfd_score", "_elevation_score")

    def __init__(self, reportline: str, pam: str, right: bool, debug: bool) -> None:
        f = reportline.split()
        self._init(_Site(f[0], f[1], f[2], f[3], int(f[4]), f[6], int(f[7]), int(f[8])), pam, right, debug)

    @classmethod
    def from_hit(cls, hit, guide: str, pam: str, right: bool, debug: bool = False) -> "Offtarget":
        """From a genome.OffTargetHit of the GPU scan and the guide spacer it belongs to."""
        site_spacer, site_pam = split_pam(hit.window, len(pam), right)
        marked = "".join(t if t == g else t.lower() for t, g in zip(site_spacer, guide))
        self = cls.__new__(cls)
        self._init(_Site("X", join_pam(guide, "N" * len(pam), right), join_pam(marked, site_pam, right), hit.contig, hit.position,
                         hit.strand, hit.mm, 0), pam, right, debug)
        return self

    def _init(self, site: _Site, pam: str, right: bool, debug: bool) -> None:
        self._site, self._pam, self._right, self._debug = site, pam, right, debug
        self._cfd_score = self._elevation_score = "NA"

    def __repr__(self) -> str:
        return f"<{type(self).__name__} object; position={self.position} spacer={self.spacer} strand={self.strand}>"

    # -- fields of the site
    chrom = property(lambda self: self._site.chrom)
    position = property(lambda self: self._site.position)
    strand = property(lambda self: self._site.strand)
    mm = property(lambda self: self._site.mm)
    bulge_type = property(lambda self: self._site.bulge_type)
    bulge_size = property(lambda self: self._site.bulge_size)

    # -- the two sequences, with and without PAM
    @property
    def grna_(self) -> str:
        return split_pam(self._site.crrna, len(self._pam), self._right)[0]

    @property
    def grna(self) -> str:
        return join_pam(self.grna_, self._pam, self._right)

    @property
    def spacer_(self) -> str:
        return split_pam(self._site.dna, len(self._pam), self._right)[0]

    @property
    def spacer(self) -> str:
        return self._site.dna  # already the site spacer with its own PAM

    # -- scores
    def cfd_inputs(self) -> Tuple[str, str, str]:
        """(wildtype, sgRNA, PAM[-2:]) exactly as compute_cfd receives them (offtarget.py:103-129)."""
        return self.grna_.upper(), self.spacer_.upper(), self.spacer[-2:]

    def set_cfd(self, value: float) -> None:
        self._cfd_score = str(round_score(value))

    cfd = property(lambda self: self._cfd_score)

    @property
    def elevation(self) -> str:
        return self._elevation_score

    @elevation.setter
    def elevation(self, value: float) -> None:
        if not isinstance(value, float):
            exception_handler(TypeError, f"Elevation must be a float, got {type(value).__name__} instead", os.EX_DATAERR, self._debug)
        self._elevation_score = "NA" if np.isnan(value) else str(round_score(value))

    def report_line(self) -> str:
        """One row of offtargets_*.tsv (offtargets.py:41-53, 530-544)."""
        s = self._site
        return "\t".join(map(str, (s.chrom, s.position, s.strand, self.grna, self.spacer, self._pam, s.mm, s.bulge_size, s.bulge_type,
                                   self._cfd_score, self._elevation_score)))
