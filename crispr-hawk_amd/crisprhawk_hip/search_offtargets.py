"""offtargets_search — reference search_offtargets.py:20-67.  ``args.crispritz_index`` may be a
GenomeIndex, a {contig: sequence} dict or a FASTA path; it plays the role of the CRISPRitz
genome index directory of the reference."""
from typing import Dict, List

from .genome import GenomeIndex, read_fasta
from .guide import Guide
from .offtargets import estimate_offtargets
from .pam import PAM
from .region import Region
from .utils import VERBOSITYLVL, print_verbosity


def offtargets_search(guides: Dict[Region, List[Guide]], pam: PAM, args) -> Dict[Region, List[Guide]]:
    print_verbosity("Searching off-targets", args.verbosity, VERBOSITYLVL[1])
    genome = args.crispritz_index
    if isinstance(genome, str):
        genome = read_fasta(genome)
    if isinstance(genome, dict):
        genome = GenomeIndex(genome, args.guidelen, len(pam))  # once for all regions
    for region, guides_list in guides.items():
        guides[region] = estimate_offtargets(  # search_offtargets.py:44-62: the reference's call, argument for argument
            guides_list, pam, genome, region, getattr(args, "crispritz_config", None), args.mm, args.bdna, args.brna,
            getattr(args, "offtargets_annotations", []), getattr(args, "offtargets_annotation_colnames", []), args.guidelen,
            getattr(args, "compute_elevation", False), args.right, getattr(args, "threads", 1), getattr(args, "outdir", ""),
            args.verbosity, args.debug)
    return guides
