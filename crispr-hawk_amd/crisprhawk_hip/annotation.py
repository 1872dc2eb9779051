"""reverse_guides — the stage-6 slice the scorers depend on (reference annotation.py:27-51).
The rest of annotation.py (variant polishing, GC, BED annotation) is out of scope."""
from typing import List

from .guide import Guide
from .utils import VERBOSITYLVL, print_verbosity


def reverse_guides(guides: List[Guide], verbosity: int) -> List[Guide]:
    print_verbosity("Reversing guides occurring on reverse strand", verbosity, VERBOSITYLVL[3])
    for guide in guides:
        if guide.strand == 1:
            guide.reverse_complement()
    return guides
