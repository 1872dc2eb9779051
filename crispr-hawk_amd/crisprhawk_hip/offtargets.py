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


def crispritz_bulge_line(hit, pamlen: int, right: bool) -> str:
    """The `targets.txt` row of a bulged site (genome.BulgeHit): type DNA / RNA, crRNA and DNA with '-' at the bulge positions
    (the field set offtarget.py:77-101 reads; `total` = mismatches + bulge size)."""
    crrna = ("N" * pamlen + hit.crrna) if right else (hit.crrna + "N" * pamlen)
    dna = (hit.pam + hit.dna) if right else (hit.dna + hit.pam)
    return (f"{hit.bulge_type}\t{crrna}\t{dna}\t{hit.contig}\t{hit.position}\t{hit.position}\t{hit.strand}\t{hit.mm}\t{hit.bulge_size}\t"
            f"{hit.mm + hit.bulge_size}")


def search(genome: GenomeIndex, guides_seqs: List[str], pam: PAM, right: bool, mm: int, verbosity: int, debug: bool,
           bdna: int = 0, brna: int = 0) -> List[str]:
    """The CRISPRitz call's replacement (`crispritz.py search ... -mm M -bDNA B -bRNA R`, offtargets.py:264-268): report lines
    for every hit of every unique spacer - the un-bulged sites, then the DNA- / RNA-bulged ones (genome.GenomeIndex.scan_bulges:
    bulges of up to 2 bases, one row per (guide, site, type, size); CRISPRitz itself is absent, so its output beyond the field
    set the reference parses is unpinned)."""
    if bdna < 0 or brna < 0 or bdna > 2 or brna > 2:
        exception_handler(CrisprHawkOffTargetsError, f"DNA / RNA bulges of 0..2 bases are enumerated (got {bdna} / {brna})", os.EX_DATAERR, debug)
    try:
        hits = genome.scan(guides_seqs, pam, right, mm)
        bulged = genome.scan_bulges(guides_seqs, pam, right, mm, bdna, brna) if (bdna or brna) else []
    except ValueError as e:
        exception_handler(CrisprHawkOffTargetsError, f"Off-targets search failed: {e}", os.EX_DATAERR, debug, e)
    return [crispritz_report_line(h, guides_seqs[h.guide], len(pam), right) for h in hits] + \
        [crispritz_bulge_line(h, len(pam), right) for h in bulged]


def _compute_cfd_score(offtargets: List[Offtarget], verbosity: int, debug: bool) -> List[Offtarget]:
    from .scoring import compute_cfd_batch
    print_verbosity(f"Computing CFD score for {len(offtargets)} off-targets", verbosity, VERBOSITYLVL[3])
    if offtargets:
        wt, sg, pm = zip(*(ot.cfd_inputs() for ot in offtargets))
        for ot, s in zip(offtargets, compute_cfd_batch(list(wt), list(sg), list(pm), debug).tolist()):
            ot.set_cfd(float(s))
    return offtargets


def _read_offtargets(crispritz_targets, pam: PAM, right: bool, debug: bool) -> List[Offtarget]:
    """offtargets.py:296-325.  `crispritz_targets`: the path of a CRISPRitz `*.targets.txt` (first line = header,
    skipped) or the scan's report lines themselves."""
    try:
        if isinstance(crispritz_targets, (str, os.PathLike)):
            with open(crispritz_targets) as infile:
                infile.readline()
                return [Offtarget(line, pam.pam, right, debug) for line in infile]
        return [Offtarget(line, pam.pam, right, debug) for line in crispritz_targets]
    except Exception as e:
        exception_handler(CrisprHawkOffTargetsError, f"Failed retrieving CRISPRitz off-targets in {crispritz_targets}", os.EX_DATAERR, debug, e)


def _tsv_float(x: str) -> str:
    """A score as pandas writes it after the reference's read_csv / to_csv round trip (offtargets.py:540-548): the text
    is parsed to float64 and written with repr(), NaN as NA."""
    return "NA" if x == "NA" else repr(float(x))


def offtargets_table(offtargets: List[Offtarget]) -> str:
    """The text of offtargets_*.tsv: header, rows sorted by (chrom, position) - pandas' two-key sort_values is a stable
    lexsort, ties keep file order -, a trailing newline.  A column without a single number (elevation; cfd for PAMs
    outside SpCas9 / xCas9) comes back from read_csv as all-NaN and is written NA throughout."""
    rows = sorted(offtargets, key=lambda o: (o.chrom, o.position))
    out = ["\t".join(OTREPCNAMES)]
    for o in rows:
        f = o.report_line().split("\t")
        f[9], f[10] = _tsv_float(f[9]), _tsv_float(f[10])
        out.append("\t".join(f))
    return "\n".join(out) + "\n"


def report_offtargets(crispritz_targets_file, region: Region, pam: PAM, guidelen: int, annotations: List[str], anncolnames: List[str],
                      compute_elevation: bool, right: bool, outdir: str, verbosity: int, debug: bool) -> List[Offtarget]:
    """offtargets.py:486-558, same arguments.  CFD for SpCas9 / xCas9 PAMs in one device batch; BED annotation of the
    table and Elevation are outside the path and refused, not skipped."""
    if annotations:
        from .crisprhawk_error import CrisprHawkAnnotationError
        exception_handler(CrisprHawkAnnotationError, "BED annotation of the off-targets table is not part of the GPU path", os.EX_DATAERR, debug)
    if compute_elevation and guidelen + len(pam) == 23 and not right:
        from .crisprhawk_error import CrisprHawkElevationScoreError
        exception_handler(CrisprHawkElevationScoreError, "Elevation is not part of the GPU scoring path", os.EX_DATAERR, debug)
    offtargets = _read_offtargets(crispritz_targets_file, pam, right, debug)
    if pam.cas_system in (SPCAS9, XCAS9):
        offtargets = _compute_cfd_score(offtargets, verbosity, debug)
    print_verbosity("Writing off-targets report", verbosity, VERBOSITYLVL[1])
    if outdir:
        fname = os.path.join(outdir, f"offtargets_{region.contig}_{region.start + PADDING}_{region.stop - PADDING}.tsv")
        try:
            with open(fname, "w") as f:
                f.write(offtargets_table(offtargets))
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


def _genome_index(crispritz_index, guidelen: int, pamlen: int, max_bulge: int = 0) -> GenomeIndex:
    """What stands where the reference passes a CRISPRitz index directory: a GenomeIndex (built with max_bulge >= the DNA bulges
    asked for), a {contig: sequence} dict or a FASTA path."""
    if isinstance(crispritz_index, GenomeIndex):
        return crispritz_index
    if isinstance(crispritz_index, (str, os.PathLike)):
        from .genome import read_fasta
        crispritz_index = read_fasta(str(crispritz_index))
    return GenomeIndex(crispritz_index, guidelen, pamlen, max_bulge=max_bulge)


def offtargets_by_spacer(offtargets: List[Offtarget], spacers) -> Dict[str, tuple]:
    """{SPACER: (number of rows, global CFD as the report prints it)} - what annotate_guides_offtargets leaves on every
    guide with that spacer (offtargets.py:597-627; Guide.cfd stores str(round(100 / (100 + sum), 4)), guide.py:723-744)."""
    from .utils import round_score
    rows: Dict[str, List[Offtarget]] = {sp.upper(): [] for sp in spacers}
    for ot in offtargets:
        rows[ot.grna_.upper().replace("-", "")].append(ot)
    return {sp: (len(r), str(round_score(_calculate_global_cfd(r)))) for sp, r in rows.items()}


def estimate_offtargets_spacers(spacers, pam: PAM, crispritz_index, region, mm: int, bdna: int, brna: int, guidelen: int, right: bool,
                                outdir: str, verbosity: int, debug: bool) -> Dict[str, tuple]:
    """estimate_offtargets for the columnar report (pipeline.search_files): the same stage - unique spacers -> device scan
    -> CFD -> offtargets_*.tsv -> per-spacer aggregates - without Guide objects."""
    uniq = sorted({sp.upper() for sp in spacers})
    lines = search(_genome_index(crispritz_index, guidelen, len(pam), bdna), uniq, pam, right, mm, verbosity, debug, bdna, brna) if uniq else []
    ots = report_offtargets(lines, region, pam, guidelen, [], [], False, right, outdir, verbosity, debug)
    return offtargets_by_spacer(ots, uniq)


def estimate_offtargets(guides: List[Guide], pam: PAM, crispritz_index, region: Region, crispritz_config, mm: int, bdna: int, brna: int,
                        annotations: List[str], anncolnames: List[str], guidelen: int, compute_elevation: bool, right: bool,
                        threads: int, outdir: str, verbosity: int, debug: bool) -> List[Guide]:
    """offtargets.py:630-722 with the reference's seventeen arguments in the reference's order.  `crispritz_index` is the
    genome (see _genome_index); `crispritz_config` (the conda environment of the external tool) and `threads` have no
    meaning on the device path and are ignored."""
    guides_seqs = sorted(_filter_guides(guides))
    print_verbosity("Estimating off-targets for found guides", verbosity, VERBOSITYLVL[3])
    genome = _genome_index(crispritz_index, guidelen, len(pam), bdna)
    lines = search(genome, guides_seqs, pam, right, mm, verbosity, debug, bdna, brna)
    offtargets = report_offtargets(lines, region, pam, guidelen, annotations, anncolnames, compute_elevation, right, outdir, verbosity, debug)
    return annotate_guides_offtargets(offtargets, guides, verbosity)
