"""Offtarget — one off-target site parsed from a CRISPRitz-format report line (reference
offtarget.py:22-239): fields [0] bulge type, [1] crRNA (with PAM), [2] DNA (with PAM),
[3] chrom, [4] position, [6] strand, [7] mismatches, [8] bulge size."""
import os
from typing import Tuple

import numpy as np

from .exception_handlers import exception_handler
from .utils import round_score


def _retrieve_pam(sequence: str, length: int, right: bool) -> str:
    return sequence[:length] if right else sequence[-length:]


def _format_sequence(sequence: str, pam: str, right: bool) -> Tuple[str, str]:
    s_ = sequence[len(pam):] if right else sequence[: -len(pam)]
    s = f"{pam}{s_}" if right else f"{s_}{pam}"
    return s_, s


class Offtarget:
    def __init__(self, reportline: str, pam: str, right: bool, debug: bool) -> None:
        self._debug = debug
        self._parse_reportline(reportline, pam, right)
        self._pam = pam
        self._cfd_score = "NA"
        self._elevation_score = "NA"

    def __repr__(self) -> str:
        return f"<{self.__class__.__name__} object; position={self._pos} spacer={self._spacer} strand={self._strand}>"

    def _parse_reportline(self, line: str, pam: str, right: bool) -> None:
        fields = line.strip().split()
        self._chrom = fields[3]
        self._pos = int(fields[4])
        self._strand = fields[6]
        self._grna_, self._grna = _format_sequence(fields[1], pam, right)
        self._spacer_, self._spacer = _format_sequence(fields[2], _retrieve_pam(fields[2], len(pam), right), right)
        self._mm = int(fields[7])
        self._bulge_type = fields[0]
        self._bulge_size = int(fields[8])

    def cfd_inputs(self) -> Tuple[str, str, str]:
        """(wildtype, sgRNA, PAM[-2:]) exactly as compute_cfd receives them (offtarget.py:103-129)."""
        return self._grna_.upper(), self._spacer_.upper(), self._spacer[-2:]

    def set_cfd(self, value: float) -> None:
        self._cfd_score = str(round_score(value))

    def report_line(self) -> str:
        return "\t".join(map(str, [self._chrom, self._pos, self._strand, self._grna, self._spacer, self._pam, self._mm,
                                   self._bulge_size, self._bulge_type, self._cfd_score, self._elevation_score]))

    grna = property(lambda self: self._grna)
    grna_ = property(lambda self: self._grna_)
    spacer = property(lambda self: self._spacer)
    cfd = property(lambda self: self._cfd_score)
    chrom = property(lambda self: self._chrom)
    position = property(lambda self: self._pos)
    strand = property(lambda self: self._strand)
    mm = property(lambda self: self._mm)

    @property
    def elevation(self) -> str:
        return self._elevation_score

    @elevation.setter
    def elevation(self, value: float) -> None:
        if not isinstance(value, float):
            exception_handler(TypeError, f"Elevation must be a float, got {type(value).__name__} instead", os.EX_DATAERR, self._debug)
        self._elevation_score = "NA" if np.isnan(value) else str(round_score(value))
