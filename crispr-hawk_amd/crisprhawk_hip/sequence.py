"""Sequence — validated DNA string (reference sequence.py:22-130).  The FASTA reader half of
the reference module is pysam I/O and out of scope."""
import os
from typing import Union

from .exception_handlers import exception_handler
from .utils import IUPAC

_VALID = set(IUPAC) | {c.lower() for c in IUPAC}


class Sequence:
    def __init__(self, sequence: str, debug: bool, allow_lower_case: bool = False) -> None:
        self._debug = debug
        sequence = sequence if allow_lower_case else sequence.upper()
        if not set(sequence) <= _VALID:
            exception_handler(ValueError, "The input string is not a DNA string", os.EX_DATAERR, self._debug)
        self._sequence = sequence

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, Sequence):
            return NotImplemented
        return self._sequence == other.sequence

    def __len__(self) -> int:
        return len(self._sequence)

    def __str__(self) -> str:
        return self._sequence

    def __iter__(self):
        return iter(self._sequence)

    def __getitem__(self, idx: Union[int, slice]):
        # the reference indexes a list of characters: an int gives one char, a slice a list
        try:
            return list(self._sequence[idx]) if isinstance(idx, slice) else self._sequence[idx]
        except IndexError as e:
            raise IndexError(f"Index {idx} out of range") from e

    @property
    def sequence(self) -> str:
        return self._sequence

    @property
    def _sequence_raw(self):
        return list(self._sequence)
