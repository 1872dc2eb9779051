"""Region / RegionList (reference region.py:15-330)."""
from typing import List, Union

from .coordinate import Coordinate
from .sequence import Sequence


class Region:
    def __init__(self, sequence: Sequence, coord: Coordinate):
        self._sequence = sequence
        self._coordinates = coord

    def __len__(self) -> int:
        return len(self._sequence)

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, Region):
            return NotImplemented
        return self._sequence == other.sequence and self._coordinates == other._coordinates

    def __hash__(self) -> int:
        return hash((self._sequence.sequence, self._coordinates.contig, self._coordinates.start, self._coordinates.stop))

    def __str__(self) -> str:
        return f">{str(self._coordinates)}\n{str(self._sequence)}"

    def __repr__(self) -> str:
        return f"<{self.__class__.__name__} object; region={str(self._coordinates)}>"

    def __getitem__(self, idx: Union[int, slice]):
        return self._sequence[idx]

    def contains(self, other: "Region") -> bool:
        if not isinstance(other, self.__class__):
            raise TypeError(f"Full overlap check on input region can only be done on {self.__class__.__name__}")
        return self.contig == other.contig and self.start <= other.start and self.stop >= other.stop

    def overlap(self, other: "Region") -> bool:
        if not isinstance(other, self.__class__):
            raise TypeError(f"Overlap check on input region can only be done on {self.__class__.__name__}")
        return self.contig == other.contig and self.start <= other.stop and self.stop >= other.start

    @property
    def contig(self) -> str:
        return self._coordinates.contig

    @property
    def start(self) -> int:
        return self._coordinates.start

    @property
    def stop(self) -> int:
        return self._coordinates.stop

    @property
    def sequence(self) -> Sequence:
        return self._sequence

    @property
    def coordinates(self) -> Coordinate:
        return self._coordinates


class RegionList:
    def __init__(self, regions: List[Region], debug: bool = False):
        self._regions = list(regions)
        self._debug = debug

    def __len__(self) -> int:
        return len(self._regions)

    def __iter__(self):
        return iter(self._regions)

    def __getitem__(self, idx):
        return self._regions[idx]

    def extend(self, regions: "RegionList") -> None:
        self._regions.extend(regions._regions)

    def append(self, region: Region) -> None:
        self._regions.append(region)
