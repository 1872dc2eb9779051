"""Off-target estimation — the reference's offtargets.py with the external CRISPRitz search
(offtargets.py:222-293) replaced by the K7 GPU scan over a GenomeIndex, and the per-site CFD
(offtargets.py:328-363) computed in one GPU batch.  The scan's hits are rendered as
CRISPRitz-format report lines so that ``Offtarget`` parsing, the per-guide counts and the global
CFD ``100 / (100 + sum(cfd))`` (offtargets.py:561-627) keep the reference's semantics.
BED annotation of the off-target table and Elevation are out of scope."""
import os
from typing import Dict, List, Set

from .crisprhawk_error import CrisprHawkOffTargetsError
from .exception_handlers import exception_handler
from .genome import GenomeIndex, OffTargetHit
from .guide import Guide
from .offtarget import Offtarget
from .pam import PAM, SPCAS9, XCAS9
from .region import Region
from .utils import VERBOSITYLVL, print_verbosity

PADDING = 100
OTREPCNAMES = ["chrom", "position", "strand", "grna", "spacer", "pam", "mm", "bulge_size", "bulg_type", "cfd", "elevation"]


def _filter_guides(guides: List[Guide]) -> Set[str]:
    return {g.guide.upper() for g in guides}


def crispritz_report_line(hit: OffTargetHit, guide: str, pamlen: int, right: bool) -> str:
    """One `targets.txt` row: bulge type, crRNA, DNA (mismatches lower-case), chrom, position,
    cluster position, strand, mismatches, bulge size, total (offtarget.py:89-101)."""
    glen = len(guide)
    sp = hit.window[pamlen:] if right else hit.window[:glen]
    pm = hit.window[:pamlen] if right else hit.window[glen:]
    dna_sp = "".join(t if t == g else t.lower() for t, g in zip(sp, guide))
    crrna = ("N" * pamlen + guide) if right else (guide + "N" * pamlen)
    dna = (pm + dna_sp) if right else (dna_sp + pm)
    return f"X\t{crrna}\t{dna}\t{hit.contig}\t{hit.position}\t{hit.position}\t{hit.strand}\t{hit.mm}\t0\t{hit.mm}"


def search(genome: GenomeIndex, guides_seqs: List[str], pam: PAM, right: bool, mm: int, verbosity: int, debug: bool) -> List[str]:
    """The CRISPRitz call's replacement: report lines for every hit of every unique spacer."""
    try:
        hits = genome.scan(guides_seqs, pam, right, mm)
    except ValueError as e:
        exception_handler(CrisprHawkOffTargetsError, f"Off-targets search failed: {e}", os.EX_DATAERR, debug, e)
    return [crispritz_report_line(h, guides_seqs[h.guide], len(pam), right) for h in hits]


def _compute_cfd_score(offtargets: List[Offtarget], verbosity: int, debug: bool) -> List[Offtarget]:
    from .scoring import compute_cfd_batch
    print_verbosity(f"Computing CFD score for {len(offtargets)} off-targets", verbosity, VERBOSITYLVL[3])
    if offtargets:
        wt, sg, pm = zip(*(ot.cfd_inputs() for ot in offtargets))
        for ot, s in zip(offtargets, compute_cfd_batch(list(wt), list(sg), list(pm), debug).tolist()):
            ot.set_cfd(float(s))
    return offtargets


def report_offtargets(lines: List[str], region: Region, pam: PAM, guidelen: int, right: bool, outdir: str, verbosity: int,
                      debug: bool) -> List[Offtarget]:
    offtargets = [Offtarget(line, pam.pam, right, debug) for line in lines]
    if pam.cas_system in (SPCAS9, XCAS9):
        offtargets = _compute_cfd_score(offtargets, verbosity, debug)
    if outdir:
        fname = os.path.join(outdir, f"offtargets_{region.contig}_{region.start + PADDING}_{region.stop - PADDING}.tsv")
        try:
            rows = sorted(offtargets, key=lambda o: (o.chrom, o.position))  # offtargets.py:541
            with open(fname, "w") as f:
                f.write("\t".join(OTREPCNAMES) + "\n")
                f.write("\n".join(o.report_line() for o in rows) + ("\n" if rows else ""))
        except OSError as e:
            exception_handler(CrisprHawkOffTargetsError, f"Failed writing off-targets report for region {region}", os.EX_IOERR, debug, e)
    return offtargets


def _calculate_offtargets_map(offtargets: List[Offtarget], guides: List[Guide]) -> Dict[str, List[Offtarget]]:
    otmap: Dict[str, List[Offtarget]] = {g.guide.upper(): [] for g in guides}
    for ot in offtargets:
        otmap[ot.grna_.upper().replace("-", "")].append(ot)
    return otmap


def _calculate_global_cfd(offtargets: List[Offtarget]) -> float:
    cfds = [0 if ot.cfd == "NA" else float(ot.cfd) for ot in offtargets]
    return 100 / (100 + sum(cfds))


def annotate_guides_offtargets(offtargets: List[Offtarget], guides: List[Guide], verbosity: int) -> List[Guide]:
    otmap = _calculate_offtargets_map(offtargets, guides)
    for guide in guides:
        guide.offtargets = len(otmap[guide.guide.upper()])
        guide.cfd = _calculate_global_cfd(otmap[guide.guide.upper()])
    return guides


def estimate_offtargets(guides: List[Guide], pam: PAM, genome: GenomeIndex, region: Region, mm: int, bdna: int, brna: int,
                        guidelen: int, right: bool, outdir: str, verbosity: int, debug: bool) -> List[Guide]:
    """offtargets.py:630-722"""
    if bdna or brna:
        exception_handler(CrisprHawkOffTargetsError, "DNA/RNA bulges are not supported by the GPU off-target scan",
                          os.EX_DATAERR, debug)
    guides_seqs = sorted(_filter_guides(guides))
    print_verbosity("Estimating off-targets for found guides", verbosity, VERBOSITYLVL[3])
    lines = search(genome, guides_seqs, pam, right, mm, verbosity, debug)
    offtargets = report_offtargets(lines, region, pam, guidelen, right, outdir, verbosity, debug)
    return annotate_guides_offtargets(offtargets, guides, verbosity)
