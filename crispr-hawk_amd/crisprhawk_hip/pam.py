"""PAM — same interface as the reference's pam.PAM (pam.py:46-173): validation, upper-casing,
reverse complement, Cas-system assessment, nibble packing of the forward and reverse-complement
patterns (first PAM base in the most significant nibble; NGG -> 0xF44, CCN -> 0x22F).
Pure host bookkeeping of <= 16 characters; the packed ints are what the scan kernel takes."""
import os
from typing import List

from .crisprhawk_error import CrisprHawkPamError
from .exception_handlers import exception_handler
from .utils import IUPAC, reverse_complement

CASXPAM = ["TTCN"]
CPF1PAM = ["TTN", "TTTN", "TYCV", "TATV", "TTTV", "TTTR", "ATTN", "TTTA", "TCTA", "TCCA", "CCCA", "YTTV", "TTYN"]
SACAS9PAM = ["NNGRRT", "NNNRRT"]
SPCAS9PAM = ["NGG", "NGA", "NRG", "NGC"]
XCAS9PAM = ["NGK", "NGN", "NNG"]
CASX, CPF1, SACAS9, SPCAS9, XCAS9 = 0, 1, 2, 3, 4

IUPAC_BITS = {"A": 1, "C": 2, "G": 4, "T": 8, "N": 15, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3, "B": 14,
              "D": 13, "H": 11, "V": 7}  # encoder.py:18-34


class PAM:
    def __init__(self, pamseq: str, right: bool, debug: bool):
        self._debug = debug
        if any(nt.upper() not in IUPAC for nt in pamseq):
            exception_handler(ValueError, f"Invalid PAM sequence {pamseq}", os.EX_DATAERR, self._debug)
        self._sequence = pamseq.upper()
        self._sequence_rc = reverse_complement(pamseq, debug)
        self._assess_cas_system(right)

    def __len__(self) -> int:
        return len(self._sequence)

    def __eq__(self, pam: object) -> bool:
        return self._sequence == pam.pam if isinstance(pam, PAM) else NotImplemented

    def __repr__(self) -> str:
        return f"<{self.__class__.__name__} object; sequence={self._sequence}>"

    def __str__(self) -> str:
        return f"{self._sequence}"

    def _assess_cas_system(self, right: bool) -> None:
        self._cas_system = -1
        if self._sequence in CASXPAM:
            self._cas_system = CASX
        elif self._sequence in CPF1PAM and right:
            self._cas_system = CPF1
        elif self._sequence in SACAS9PAM:
            self._cas_system = SACAS9
        elif self._sequence in SPCAS9PAM and not right:
            self._cas_system = SPCAS9
        elif self._sequence in XCAS9PAM and not right:
            self._cas_system = XCAS9

    def encode(self, verbosity: int) -> None:
        try:
            self._sequence_bits = [IUPAC_BITS[c] for c in self._sequence.upper()]
            self._sequence_rc_bits = [IUPAC_BITS[c] for c in self._sequence_rc.upper()]
            self._packed_bits = _pack_bits(self._sequence_bits)
            self._packed_bitsrc = _pack_bits(self._sequence_rc_bits)
        except (ValueError, KeyError) as e:
            exception_handler(CrisprHawkPamError, "PAM bit encoding failed", os.EX_DATAERR, self._debug, e)

    @property
    def pam(self) -> str:
        return self._sequence

    @property
    def pamrc(self) -> str:
        return self._sequence_rc

    @property
    def bits(self) -> int:
        return self._packed_bits

    @property
    def bitsrc(self) -> int:
        return self._packed_bitsrc

    @property
    def bits_list(self) -> List[int]:
        return self._sequence_bits

    @property
    def cas_system(self) -> int:
        return self._cas_system


def _pack_bits(bits: List[int]) -> int:
    packed = 0
    for b in bits:
        packed = (packed << 4) | b
    return packed
