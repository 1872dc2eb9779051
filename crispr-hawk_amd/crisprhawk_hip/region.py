"""Region: a stretch of a contig - its bases (`Sequence`) and where they sit (`Coordinate`) - plus RegionList, the
container the region constructor hands to the search (reference region.py:15-330).  Both are thin: the search reads
`region.contig/.start/.stop`, slices `region[a:b]` and hashes regions as dictionary keys (`guides[region]`)."""
from dataclasses import dataclass
from typing import Iterable, Union

from .coordinate import Coordinate
from .sequence import Sequence


@dataclass(eq=False, repr=False)
class Region:
    sequence: Sequence
    coordinates: Coordinate

    # -- identity: two regions are the same when bases and interval are
    def _key(self):
        c = self.coordinates
        return self.sequence.sequence, c.contig, c.start, c.stop

    def __eq__(self, other: object) -> bool:
        return self._key() == other._key() if isinstance(other, Region) else NotImplemented

    def __hash__(self) -> int:
        return hash(self._key())

    # -- the bases
    def __len__(self) -> int:
        return len(self.sequence)

    def __getitem__(self, idx: Union[int, slice]):
        return self.sequence[idx]

    def __str__(self) -> str:
        return f">{self.coordinates}\n{self.sequence}"

    def __repr__(self) -> str:
        return f"<{type(self).__name__} object; region={self.coordinates}>"

    # -- the interval
    contig = property(lambda self: self.coordinates.contig)
    start = property(lambda self: self.coordinates.start)
    stop = property(lambda self: self.coordinates.stop)

    def _same_contig(self, other: "Region", what: str) -> bool:
        if not isinstance(other, type(self)):
            raise TypeError(f"{what} on input region can only be done on {type(self).__name__}")
        return self.contig == other.contig

    def contains(self, other: "Region") -> bool:
        return self._same_contig(other, "Full overlap check") and self.start <= other.start and other.stop <= self.stop

    def overlap(self, other: "Region") -> bool:
        return self._same_contig(other, "Overlap check") and self.start <= other.stop and other.start <= self.stop


class RegionList(list):
    """A list of Region objects (the reference wraps a list; here it is one)."""

    def __init__(self, regions: Iterable[Region] = (), debug: bool = False):
        super().__init__(regions)
        self._debug = debug
