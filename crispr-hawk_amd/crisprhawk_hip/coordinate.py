"""Coordinate — genomic interval with padding (reference coordinate.py:8-136).

`start` / `stop` are the PADDED bounds (what the region string covers, coordinate.py:118-128), `startp` / `stopp`
the interval as given (the BED bounds)."""
from typing import NamedTuple


class _Bounds(NamedTuple):
    contig: str
    given_start: int
    given_stop: int
    padding: int


class Coordinate:
    __slots__ = ("_b",)

    def __init__(self, contig: str, start: int, stop: int, padding: int) -> None:
        if stop < start:
            raise ValueError("Stop < start coordinate")
        self._b = _Bounds(contig, start, stop, padding)

    # the padded interval; a start closer to the contig's beginning than the padding is clipped at 0
    start = property(lambda self: max(0, self._b.given_start - self._b.padding))
    stop = property(lambda self: self._b.given_stop + self._b.padding)
    # the interval as given
    startp = property(lambda self: self._b.given_start)
    stopp = property(lambda self: self._b.given_stop)
    contig = property(lambda self: self._b.contig)

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, Coordinate):
            return NotImplemented
        return (self.contig, self.start, self.stop) == (other.contig, other.start, other.stop)

    def __hash__(self) -> int:
        return hash((self.contig, self.start, self.stop))

    def __str__(self) -> str:
        return f"{self.contig}:{self.startp}-{self.stopp}"

    def __repr__(self) -> str:
        b = self._b
        return (f"<{type(self).__name__} object; coordinate={b.contig}:{b.given_start + b.padding}-"
                f"{b.given_stop - b.padding}; padding={b.padding}>")

    def contains(self, query: object) -> bool:
        if not isinstance(query, Coordinate):
            return NotImplemented
        return self.contig == query.contig and self.startp <= query.start and self.stopp >= query.stop
