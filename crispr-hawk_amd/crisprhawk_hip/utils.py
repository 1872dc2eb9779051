"""Constants and small helpers shared by the hot path (reference utils.py:37-105, 123-145,
148-182, 203-262, 318-338)."""
import sys
from itertools import permutations
from typing import Any, List, Tuple

from .exception_handlers import exception_handler

VERBOSITYLVL = [0, 1, 2, 3]
DNA = ["A", "C", "G", "T", "N"]
IUPAC = DNA + ["R", "Y", "S", "W", "K", "M", "B", "D", "H", "V"]
_RC_UP = {"A": "T", "C": "G", "G": "C", "T": "A", "U": "A", "R": "Y", "Y": "R", "M": "K", "K": "M", "H": "D",
          "D": "H", "B": "V", "V": "B", "N": "N", "S": "S", "W": "W"}
RC = dict(_RC_UP)
RC.update({k.lower(): v.lower() for k, v in _RC_UP.items()})
IUPACTABLE = {"A": "A", "C": "C", "G": "G", "T": "T", "R": "AG", "Y": "CT", "M": "AC", "K": "GT", "S": "CG",
              "W": "AT", "H": "ACT", "B": "CGT", "V": "ACG", "D": "AGT", "N": "ACGT"}
IUPAC_ENCODER = {perm: k for k, v in IUPACTABLE.items() for perm in {"".join(p) for p in permutations(v)}}
STRAND = [0, 1]
_RC_TRANS = str.maketrans("".join(RC.keys()), "".join(RC.values()))


def reverse_complement(sequence: str, debug: bool) -> str:
    if any(c not in RC for c in sequence):
        exception_handler(ValueError, f"Failed reverse complement on {sequence}", 65, debug)
    return sequence[::-1].translate(_RC_TRANS)


def print_verbosity(message: str, verbosity: int, verbosity_threshold: int) -> None:
    if verbosity >= verbosity_threshold:
        sys.stdout.write(f"{message}\n")


def warning(message: str, verbosity: int) -> None:
    if verbosity >= VERBOSITYLVL[1]:
        sys.stderr.write(f"WARNING: {message}\n")


def round_score(score: float) -> float:
    return round(score, 4)


def flatten_list(lst: List[List[Any]]) -> List[Any]:
    return [e for sub in lst for e in sub]


def dna2rna(sequence: str) -> str:
    return sequence.replace("T", "U").replace("t", "u")


def match_iupac(seq: str, pattern: str) -> bool:
    if len(seq) != len(pattern):
        return False
    return all(s in IUPACTABLE[p] for s, p in zip(seq.upper(), pattern.upper()))


def calculate_chunks(lst: List[Any], threads: int) -> List[Tuple[int, List[Any]]]:
    size = len(lst)
    chunk = max(1, size // threads)
    return [(i, lst[i:min(i + chunk, size)]) for i in range(0, size, chunk)]
