"""Coordinate — genomic interval with padding (reference coordinate.py:8-136)."""


class Coordinate:
    def __init__(self, contig: str, start: int, stop: int, padding: int) -> None:
        if stop < start:
            raise ValueError("Stop < start coordinate")
        self._contig = contig
        self._start = start
        self._startp = max(0, start - padding)
        self._stop = stop
        self._stopp = stop + padding
        self._padding = padding

    def __eq__(self, other: object) -> bool:
        if not isinstance(other, Coordinate):
            return NotImplemented
        return self._contig == other.contig and self._startp == other.start and self._stopp == other.stop

    def __hash__(self) -> int:
        return hash((self.contig, self.start, self.stop))

    def __repr__(self) -> str:
        return (f"<{self.__class__.__name__} object; coordinate={self._contig}:{self._start + self._padding}-"
                f"{self._stop - self._padding}; padding={self._padding}>")

    def __str__(self) -> str:
        return f"{self._contig}:{self._start}-{self._stop}"

    def contains(self, query: object) -> bool:
        if not isinstance(query, Coordinate):
            return NotImplemented
        return self._contig == query.contig and self._start <= query.start and self._stop >= query.stop

    @property
    def contig(self) -> str:
        return self._contig

    @property
    def start(self) -> int:  # padded start (coordinate.py:122-124)
        return self._startp

    @property
    def stop(self) -> int:  # padded stop
        return self._stopp

    @property
    def startp(self) -> int:  # BED start
        return self._start

    @property
    def stopp(self) -> int:  # BED stop
        return self._stop
