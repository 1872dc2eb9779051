"""PAM — same interface as the reference's pam.PAM (pam.py:46-173): validation, upper-casing,
reverse complement, Cas-system assessment, nibble packing of the forward and reverse-complement
patterns (first PAM base in the most significant nibble; NGG -> 0xF44, CCN -> 0x22F).
Pure host bookkeeping of <= 16 characters; the packed ints are what the scan kernel takes."""
import os
from functools import reduce
from typing import List, Optional

from .crisprhawk_error import CrisprHawkPamError
from .exception_handlers import exception_handler
from .utils import IUPAC, reverse_complement

CASX, CPF1, SACAS9, SPCAS9, XCAS9 = range(5)

# Cas system -> (its PAMs, the guide side it needs: None = either, True = guide right of the PAM, False = left);
# the order is the order the reference tests them in (pam.py:114-125)
_SYSTEMS = (
    (CASX, ("TTCN",), None),
    (CPF1, ("TTN", "TTTN", "TYCV", "TATV", "TTTV", "TTTR", "ATTN", "TTTA", "TCTA", "TCCA", "CCCA", "YTTV", "TTYN"), True),
    (SACAS9, ("NNGRRT", "NNNRRT"), None),
    (SPCAS9, ("NGG", "NGA", "NRG", "NGC"), False),
    (XCAS9, ("NGK", "NGN", "NNG"), False),
)
CASXPAM, CPF1PAM, SACAS9PAM, SPCAS9PAM, XCAS9PAM = (list(pams) for _, pams, _ in _SYSTEMS)

# IUPAC letter -> 4-bit base set, bit 0 A, bit 1 C, bit 2 G, bit 3 T (encoder.py:18-34)
IUPAC_BITS = {letter: sum(1 << "ACGT".index(b) for b in bases) for letter, bases in
              dict(A="A", C="C", G="G", T="T", N="ACGT", R="AG", Y="CT", S="CG", W="AT", K="GT", M="AC", B="CGT", D="AGT",
                   H="ACT", V="ACG").items()}


def _cas_system_of(pam_upper: str, right: bool) -> int:
    for system, pams, side in _SYSTEMS:
        if pam_upper in pams and (side is None or side == bool(right)):
            return system
    return -1


def _pack_bits(bits: List[int]) -> int:
    return reduce(lambda acc, b: (acc << 4) | b, bits, 0)


class PAM:
    def __init__(self, pamseq: str, right: bool, debug: bool):
        self._debug = debug
        bad = [nt for nt in pamseq if nt.upper() not in IUPAC]
        if bad:
            exception_handler(ValueError, f"Invalid PAM sequence {pamseq}", os.EX_DATAERR, debug)
        self._fwd, self._rev = pamseq.upper(), reverse_complement(pamseq, debug)
        self._cas_system = _cas_system_of(self._fwd, right)
        self._codes: Optional[List[List[int]]] = None
        self._packed: Optional[List[int]] = None

    def encode(self, verbosity: int) -> None:
        try:
            self._codes = [[IUPAC_BITS[c] for c in s.upper()] for s in (self._fwd, self._rev)]
            self._packed = [_pack_bits(c) for c in self._codes]
        except (ValueError, KeyError) as e:
            exception_handler(CrisprHawkPamError, "PAM bit encoding failed", os.EX_DATAERR, self._debug, e)

    def __len__(self) -> int:
        return len(self._fwd)

    def __eq__(self, other: object) -> bool:
        return self._fwd == other.pam if isinstance(other, PAM) else NotImplemented

    def __hash__(self) -> int:
        return hash(self._fwd)

    def __repr__(self) -> str:
        return f"<{type(self).__name__} object; sequence={self._fwd}>"

    def __str__(self) -> str:
        return self._fwd

    pam = property(lambda self: self._fwd)
    pamrc = property(lambda self: self._rev)
    bits = property(lambda self: self._packed[0])
    bitsrc = property(lambda self: self._packed[1])
    bits_list = property(lambda self: self._codes[0])
    cas_system = property(lambda self: self._cas_system)
