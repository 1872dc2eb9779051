"""encode() — the reference's encoder.encode (encoder.py:48-57) with the work done by K1 on
the GPU.  Returns an ``EncodedSequence``: list-like view of the per-base IUPAC nibbles
(what the reference returns as List[int]) backed by the bit-planes in HBM."""
import os
from typing import List

import numpy as np

from . import _lib
from .crisprhawk_error import CrisprHawkIupacTableError
from .exception_handlers import exception_handler
from .hapset import DeviceHapSet, HostHaplotype, PosSegments
from .utils import VERBOSITYLVL, print_verbosity


class EncodedSequence:
    def __init__(self, ds: DeviceHapSet, n: int):
        self._ds, self._n, self._nib = ds, n, None

    def _nibbles(self) -> np.ndarray:
        if self._nib is None:
            self._nib = self._ds.nibbles(0)
        return self._nib

    def __len__(self) -> int:
        return self._n

    def __getitem__(self, i):
        v = self._nibbles()[i]
        return v.tolist() if isinstance(i, slice) else int(v)

    def __iter__(self):
        return iter(self._nibbles().tolist())

    def __eq__(self, other) -> bool:
        return list(self) == list(other)

    def tolist(self) -> List[int]:
        return self._nibbles().tolist()


def encode(sequence: str, verbosity: int, debug: bool) -> EncodedSequence:
    print_verbosity(f"Encoding sequence {sequence} in bits", verbosity, VERBOSITYLVL[3])
    if len(sequence) == 0:
        return _Empty()
    try:
        ds = DeviceHapSet([HostHaplotype(sequence, PosSegments.identity(0, len(sequence)), True, (0, 0))])
    except _lib.HawkStatusError as e:
        if e.status != _lib.HAWK_E_IUPAC:
            raise
        pos = int(str(e).split("position ")[1].split(" ")[0])
        exception_handler(CrisprHawkIupacTableError,
                          f"The nucleotide {sequence[pos].upper()} at position {pos} is not a IUPAC character",
                          os.EX_DATAERR, debug)
    return EncodedSequence(ds, len(sequence))


class _Empty(EncodedSequence):
    def __init__(self):
        self._n, self._nib = 0, np.zeros(0, np.uint8)

    def _nibbles(self):
        return self._nib
