"""Files in, guide reports out: the path of `crisprhawk_search` (`crisprhawk.py:121-138`) for phased or variant-free
inputs, with every stage on the device path of this package.

    FASTA + BED (+ VCF)  --readers-->  region string, VCF record text
                         --hawk_gt_parse / hawk_gt_lists / hawk_hapset_expand-->  haplotype planes in HBM
                         --hawk_search (+ CFDon)-->  guide table in HBM
                         --hawk_table_collapse-->  report groups
                         --reports.report_frame-->  crisprhawk_guides__*.tsv

Only what the reference's search sub-command does between reading its inputs and writing the guide report is covered;
BED/gene annotations, the non-CFDon scorers (model files) and the off-target stage have their own entry points
(`scoring.py`, `offtargets.py`).
"""
import os
from typing import Dict, List, Optional

import numpy as np

from . import reports
from . import scoring
from .pam import CPF1, PAM, SPCAS9, XCAS9
from .readers import VCF, Bed, Fasta
from .expand import HaplotypeBuildError
from .workload import HapInfo, RowLabel, expand_from_vcf, hap_labels

PADDING = 100  # region_constructor.py:21


def _labels(ds, info: List[HapInfo], kept: List[int], vt) -> List[Optional[RowLabel]]:
    """RowLabel per device row: samples joined as collapse_haplotypes does, variant ids in the reference's order
    (SNVs by position, then indels: haplotype.py:234-242), allele frequencies by id, ids hap_<k> in list order."""
    out: List[Optional[RowLabel]] = [None] * ds.n_hap
    for k, (r, inf) in enumerate(zip(kept, info)):
        idx = [int(i) for i in inf.variant_idx]
        snv = [i for i in idx if len(vt.ref[i]) == len(vt.alt[i])]
        indel = [i for i in idx if len(vt.ref[i]) != len(vt.alt[i])]
        ids = [vt.id[i] for i in snv + indel]
        out[r] = RowLabel(",".join(inf.samples), ",".join(ids) if ids else "NA", {vt.id[i]: float(vt.af[i]) for i in idx},
                          f"hap_{k:08d}", ds.host_meta[r].seg)
    return out


def _offtargets(spacers, pam: PAM, ot, coord, guidelen: int, right: bool, outdir: str, debug: bool):
    """--estimate-offtargets for one region: {SPACER: (count, global CFD)} + offtargets_{contig}_{start}_{stop}.tsv."""
    from .offtargets import estimate_offtargets_spacers
    return estimate_offtargets_spacers(spacers, pam, ot["genome"], coord, ot["mm"], ot["bdna"], ot["brna"], guidelen, right, outdir, 0, debug)


def _search_host_built(coord, seq: str, vcf, phased: bool, pam: PAM, guidelen: int, right: bool, outdir: str, mm, pt, debug: bool,
                       ot=None) -> str:
    """One BED interval with the haplotypes built on the host by the mirror of the reference's own construction
    (haplotypes.py:106-368 phased, 370-712 unphased) - the route of unphased VCFs (IUPAC haplotypes + indel windows,
    resolve_guide on the host, search_guides.py:163-257) and the fallback for phased records the device expansion
    declines (a chromosome copy carrying overlapping records, multi-base deletion alts).  The search itself still runs
    on the device; then the reference's order of business: annotate -> reverse_guides -> CFDon -> report."""
    from . import haplotypes as hap_mod
    from .annotation import reverse_guides
    from .haplotype import Haplotype
    from .region import Region
    from .search_guides import search
    from .sequence import Sequence
    region = Region(Sequence(seq, debug), coord)
    haps = [Haplotype(Sequence(seq, debug), region.coordinates, False, 0, debug)]
    records = vcf.fetch(coord)
    if records:
        build = hap_mod.add_variants_phased if phased else hap_mod.add_variants_unphased
        haps = build(haps, region, vcf.samples, records, phased, debug)
    for i, h in enumerate(haps):
        h.id = f"hap_{i:08d}"
    guides = search(pam, region, haps, None, guidelen, right, bool(records), phased, 0, debug)
    cfd = None
    if mm is not None:
        scoring.set_cfd_tables(mm, pt)
        windows = [g.sequence for g in guides]  # the report wants the + strand windows: keep them across the reversal
        rights = [g.right for g in guides]
        scored = scoring.cfdon_score(reverse_guides(list(guides), 0), 0, debug)
        val = {id(g): g.cfdon_score for g in scored}
        cfd = [float("nan") if val[id(g)] == "NA" else float(val[id(g)]) for g in guides]
        for g, w, r in zip(guides, windows, rights):  # undo the in-place reversal
            if g.strand == 1:
                g.reverse_complement()
            assert g.sequence == w and g.right == r
    bed_start, bed_stop = coord.start + PADDING, coord.stop - PADDING
    otmap = None
    if ot is not None:  # spacers as reverse_guides would leave them (annotation.py:27-51)
        from .utils import _RC_TRANS
        cores = [g.sequence[10:-10][::-1].translate(_RC_TRANS) if g.strand == 1 else g.sequence[10:-10] for g in guides]
        otmap = _offtargets([c[len(pam):] if right else c[:guidelen] for c in cores], pam, ot, coord, guidelen, right, outdir, debug)
    df = reports.report_from_guides(guides, haps, pam, coord.contig, f"{coord.contig}:{bed_start}-{bed_stop}", cfd, otmap)
    path = os.path.join(outdir, reports.report_filename(coord.contig, bed_start, bed_stop, pam, guidelen))
    with open(path, "w") as f:
        f.write(reports.to_tsv(df))
    return path


def search_files(fasta: str, bedfile: str, vcfs: List[str], pam_seq: str, guidelen: int, right: bool, outdir: str,
                 cfd_tables=None, azimuth_model=None, deepcpf1_weights=None, device: Optional[int] = None,
                 debug: bool = True, estimate_offtargets=None, mm: int = 4, bdna: int = 0, brna: int = 0,
                 timings: Optional[Dict[str, float]] = None) -> Dict[str, str]:
    """One report per BED interval; returns {str(coordinate): path}.  `cfd_tables = (mm[20,4,4], pam[16])` adds the
    CFDon column for SpCas9-class PAMs (scoring.py:352-387); `azimuth_model` (a fitted sklearn GBR or the flattened
    dict of scoring.azimuth_model_from_sklearn) and `deepcpf1_weights` (scoring.set_deepcpf1_weights layout) switch
    their score columns on - the reference reads those parameters from files it downloads.  `estimate_offtargets` (the
    reference's --estimate-offtargets with its --crispritz-index: a genome.GenomeIndex, a {contig: sequence} dict or a
    FASTA path) runs the off-target stage per region: the `offtargets` / `cfd` columns of the guide report
    (reports.py:292-333, 612-660) and offtargets_{contig}_{start}_{stop}.tsv next to it (offtargets.py:486-558); `mm`,
    `bdna`, `brna` as on the reference's command line (bulges of up to 2 bases).  The per-site CFD needs `cfd_tables`."""
    import time as _time
    _t = [_time.perf_counter()]

    def lap(stage: str) -> None:  # stage seconds into `timings` (bench.py's files_to_tsv line); no-op without it
        if timings is not None:
            now = _time.perf_counter()
            timings[stage] = timings.get(stage, 0.0) + now - _t[0]
            _t[0] = now
    ot = None
    if estimate_offtargets is not None:
        from .genome import GenomeIndex, read_fasta
        genome = estimate_offtargets
        if isinstance(genome, (str, os.PathLike)):
            genome = read_fasta(str(genome))
        if isinstance(genome, dict):
            genome = GenomeIndex(genome, guidelen, len(pam_seq), device=device, max_bulge=bdna)  # once for all regions
        ot = dict(genome=genome, mm=mm, bdna=bdna, brna=brna)
        if cfd_tables is not None:
            scoring.set_cfd_tables(*cfd_tables)
    if azimuth_model is not None:
        scoring.set_azimuth_model(azimuth_model)
    if deepcpf1_weights is not None:
        scoring.set_deepcpf1_weights(deepcpf1_weights)
    pam = PAM(pam_seq, right, debug)
    pam.encode(0)
    fa = Fasta(fasta, 0, debug)
    fastas = {fa.contig: fa}
    vcf_by_contig = {}
    for f in vcfs or []:
        v = VCF(f, 0, debug)
        vcf_by_contig[v.contig] = v
    score = cfd_tables is not None and pam.cas_system in (SPCAS9, XCAS9) and not right
    mmt, pt = cfd_tables if score else (None, None)
    os.makedirs(outdir, exist_ok=True)
    paths = {}
    lap("open inputs (FASTA index, VCF header + line index)")
    for coord in Bed(bedfile, PADDING, debug):
        seq = fastas[coord.contig].fetch(coord).sequence
        v = vcf_by_contig.get(coord.contig)
        if v is not None and not v.phased:
            paths[str(coord)] = _search_host_built(coord, seq, v, False, pam, guidelen, right, outdir, mmt if score else None,
                                                   pt if score else None, debug, ot)
            continue
        from .readers import VcfBlock
        blk = v.fetch_block(coord) if v is not None else VcfBlock(np.zeros(0, np.uint8), np.zeros(1, np.uint64), np.zeros(0, np.uint64), [])
        samples = v.samples if v is not None else []
        lap("fetch region + VCF record text")
        try:
            ds, info, _, kept, vt = expand_from_vcf(seq, coord.start, coord.stop, blk, samples, len(pam), True, device, keep_plan=True)
        except HaplotypeBuildError:
            # records the device expansion does not take (overlapping records on one chromosome copy, deletions with a
            # multi-base alt): the host builder mirrors the reference's own construction, the search stays on the device
            paths[str(coord)] = _search_host_built(coord, seq, v, True, pam, guidelen, right, outdir, mmt if score else None,
                                                   pt if score else None, debug, ot)
            continue
        # the search runs from the expansion plan (hawk_xplan_view: once per distinct cluster of neighbouring variants when the
        # panel shares them, per row otherwise; no planes read) - a region without variants has no plan and searches REF's planes
        lap("genotype parse + plan on the device")
        plan = getattr(ds, "plan", None)
        target = plan.view() if plan is not None else ds
        tab = target.search(pam.bits, pam.bitsrc, len(pam), guidelen, right, mmt, pt, download=False)
        labels = hap_labels(coord.contig, vt, ds, info, kept)
        bed_start, bed_stop = coord.start + PADDING, coord.stop - PADDING  # reports.py:1036-1041
        # With a model scorer on, the reference's groupby includes its score column (reports.py:978-1003): rows that
        # differ only in the 4 + 3 flanking bases the scorer reads stay separate, so the device groups on the k-mer.
        azimuth_on = pam.cas_system in (SPCAS9, XCAS9) and azimuth_model is not None
        deepcpf1_on = pam.cas_system == CPF1 and deepcpf1_weights is not None
        tab.collapse((4, 3) if (azimuth_on or deepcpf1_on) else (0, 0), download_perm=False)
        groups = tab.export_groups()
        tab.close()
        lap("dictionary + search + collapse + export of the groups")
        # model-based scorers run once per report row, on the group representatives (scoring.py:749-813), when the
        # caller has supplied their parameters
        scores = {}
        if groups.n_groups and (azimuth_on or deepcpf1_on):
            kmers = reports.group_kmers(groups)
            if azimuth_on:
                scores["score_azimuth"] = np.asarray(scoring.azimuth(kmers, debug), dtype=np.float64)
            if deepcpf1_on:
                scores["score_deepcpf1"] = np.asarray(scoring.deepcpf1(kmers, debug), dtype=np.float64)
        otcb = None if ot is None else (lambda spacers: _offtargets(spacers, pam, ot, coord, guidelen, right, outdir, debug))
        # the report as columns (no Python string per row: the carriers' columns of a 2504-sample panel are 0.6 GB of text), written
        # by the library's TSV writer
        cols, order, plain = reports.group_columns(groups, labels, pam, coord.contig, f"{coord.contig}:{bed_start}-{bed_stop}", scores, score,
                                                   is_ref_hap=np.asarray(ds.is_ref, dtype=bool), offtargets=otcb)
        if plan is not None:
            plan.close()
        ds.close()
        lap("report assembly")
        path = os.path.join(outdir, reports.report_filename(coord.contig, bed_start, bed_stop, pam, guidelen))
        reports.write_report_tsv(path, cols, order, plain)
        paths[str(coord)] = path
        lap("TSV text + write")
    return paths
