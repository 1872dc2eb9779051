"""Guide — the reference's per-candidate record (guide.py:24-744): 10-nt padded window,
genomic start/stop, strand, sample/variant labels, ten score slots stored as strings
("NA" or str(round(x, 4))).  Pure host object; rows come from the device GuideTable."""
import os
from typing import Dict, List, Union

import numpy as np

from .crisprhawk_error import CrisprHawkGuideError
from .exception_handlers import exception_handler
from .utils import RC, round_score

GUIDESEQPAD = 10  # guide.py:21
_RC_TRANS = str.maketrans("".join(RC.keys()), "".join(RC.values()))


def _score_property(name: str, label: str):
    attr = f"_{name}"

    def getter(self) -> str:
        return getattr(self, attr)

    def setter(self, value: float) -> None:
        if not isinstance(value, float):
            exception_handler(CrisprHawkGuideError, f"{label} score must be a float, got {type(value).__name__} instead",
                              os.EX_DATAERR, True)
        setattr(self, attr, "NA" if np.isnan(value) else str(round_score(value)))

    return property(getter, setter)


class Guide:
    def __init__(self, position_start: int, position_stop: int, sequence: str, guidelen: int, pamlen: int,
                 direction: int, samples: str, variants: str, afs: Dict[str, float], posmap: Dict[int, int],
                 debug: bool, right: bool, hapid: str) -> None:
        self._debug = debug
        self._guidelen = guidelen
        self._pamlen = pamlen
        self._start = position_start
        self._stop = position_stop
        self._sequence = sequence
        self._right = right
        self._compute_pamguide_sequences()
        self._direction = direction
        self._samples = samples
        self._variants = variants
        self._afs = afs
        self._posmap = posmap
        self._hapid = hapid
        self._compute_guide_id()
        self._initialize_scores()
        self._initialize_annotations()

    def __repr__(self) -> str:
        return (f"<{self.__class__.__name__} object; start={self._start} stop={self._stop} "
                f"sequence={self._sequence} direction={self._direction}>")

    def __len__(self) -> int:
        return len(self._sequence)

    def __getitem__(self, idx: Union[int, slice]) -> str:
        try:
            return "".join(self._sequence[idx])
        except IndexError as e:
            exception_handler(CrisprHawkGuideError, f"Index {idx} out of range", os.EX_DATAERR, self._debug, e)

    def __iter__(self):
        return iter(self._sequence)

    def _compute_pamguide_sequences(self) -> None:  # guide.py:184-197
        core = self._sequence[GUIDESEQPAD:-GUIDESEQPAD]
        if self._right:
            self._pamseq, self._guideseq = core[: self._pamlen], core[self._pamlen:]
        else:
            self._pamseq, self._guideseq = core[-self._pamlen:], core[: -self._pamlen]

    def _compute_guide_id(self) -> None:
        self._guide_id = f"{self._start}_{self._stop}_{self._direction}_{self._hapid}_{self._guideseq}"

    def _initialize_scores(self) -> None:
        for s in ("azimuth_score", "rs3_score", "cfdon_score", "elevationon_score", "deepcpf1_score", "ooframe_score",
                  "cfd", "plmcrispr_score", "crispron_score", "sgdesigner_score"):
            setattr(self, f"_{s}", "NA")

    def _initialize_annotations(self) -> None:
        self._gc = "NA"
        self._offtargets_num = "NA"
        self._funcann: List[str] = []
        self._geneann: List[str] = []
        self._afs_ = "NA"

    def reverse_complement(self) -> None:  # guide.py:245-255
        self._sequence = self._sequence[::-1].translate(_RC_TRANS)
        self._right = not self._right
        self._compute_pamguide_sequences()

    start = property(lambda self: self._start)
    stop = property(lambda self: self._stop)
    strand = property(lambda self: self._direction)
    sequence = property(lambda self: self._sequence)
    samples = property(lambda self: self._samples)
    afs = property(lambda self: self._afs)
    pam = property(lambda self: self._pamseq)
    pamlen = property(lambda self: self._pamlen)
    guide = property(lambda self: self._guideseq)
    guidelen = property(lambda self: self._guidelen)
    right = property(lambda self: self._right)
    guide_id = property(lambda self: self._guide_id)
    hapid = property(lambda self: self._hapid)

    @property
    def guidepam(self) -> str:  # guide.py:369-372
        return self._pamseq + self._guideseq if self._right else self._guideseq + self._pamseq

    @property
    def variants(self) -> str:
        return self._variants

    @variants.setter
    def variants(self, value: str) -> None:
        if not isinstance(value, str):
            exception_handler(CrisprHawkGuideError, f"Variants must be a string, got {type(value).__name__} instead",
                              os.EX_DATAERR, True)
        self._variants = value

    @property
    def afs_str(self) -> str:
        return self._afs_

    @afs_str.setter
    def afs_str(self, value: List[str]) -> None:
        if not isinstance(value, list):
            exception_handler(CrisprHawkGuideError, f"AFs must be a list, got {type(value).__name__} instead",
                              os.EX_DATAERR, True)
        self._afs_ = ",".join(value) if value else "NA"

    @property
    def posmap(self) -> Dict[int, int]:
        if callable(self._posmap):  # lazily materialised from the haplotype's segments
            self._posmap = self._posmap()
        return self._posmap

    @posmap.setter
    def posmap(self, value: Dict[int, int]) -> None:
        if not isinstance(value, dict):
            exception_handler(CrisprHawkGuideError, f"Posmap must be a dict, got {type(value).__name__} instead",
                              os.EX_DATAERR, True)
        self._posmap = value

    azimuth_score = _score_property("azimuth_score", "Azimuth")
    rs3_score = _score_property("rs3_score", "RS3")
    deepcpf1_score = _score_property("deepcpf1_score", "DeepCpf1")
    cfdon_score = _score_property("cfdon_score", "CFD-on")
    elevationon_score = _score_property("elevationon_score", "Elevation-on")
    plmcrispr_score = _score_property("plmcrispr_score", "PLM-CRISPR")
    crispron_score = _score_property("crispron_score", "CRISPRon")
    sgdesigner_score = _score_property("sgdesigner_score", "sgDesigner")
    cfd = _score_property("cfd", "CFD")
    gc = _score_property("gc", "GC content")

    @property
    def funcann(self) -> List[str]:
        return self._funcann

    @funcann.setter
    def funcann(self, value: str) -> None:
        self._funcann.append(value)

    @property
    def geneann(self) -> List[str]:
        return self._geneann

    @geneann.setter
    def geneann(self, value) -> None:
        self._geneann.append(value)

    @property
    def offtargets(self) -> str:
        return self._offtargets_num

    @offtargets.setter
    def offtargets(self, value: int) -> None:
        if not isinstance(value, int) or isinstance(value, bool):
            exception_handler(CrisprHawkGuideError, f"Off-targets number must be an int, got {type(value).__name__} instead",
                              os.EX_DATAERR, True)
        self._offtargets_num = str(value)
