#!/usr/bin/env python3
"""C3 as FILES (FASTA + BED + phased VCF text, 2504 samples) -> pipeline.search_files -> the guide report TSV: wall clock and
stage seconds (the reference's `crisprhawk search` from its inputs to its report).  The files are written to a scratch directory first (untimed)."""
import json, os, shutil, sys, tempfile, time
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "crispr-hawk_amd")]
import numpy as np
from crisprhawk_hip import synth
from crisprhawk_hip.pipeline import search_files


def write_inputs(reg, d):
    return synth.write_region_files(reg, d, "c3")


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    reg = synth.config_c3()
    d = tempfile.mkdtemp(prefix="hawk_c3_files_", dir=os.environ.get("HAWK_SCRATCH", "/tmp"))
    try:
        t0 = time.perf_counter()
        fa, bed, vcf = write_inputs(reg, d)
        print(f"inputs written in {time.perf_counter() - t0:.1f} s: VCF {os.path.getsize(vcf) / 1e6:.0f} MB", file=sys.stderr, flush=True)
        mm, pt = synth.cfd_tables()
        out = []
        for r in range(reps):
            tm = {}
            t0 = time.perf_counter()
            paths = search_files(fa, bed, [vcf], "NGG", 20, False, os.path.join(d, f"out{r}"), cfd_tables=(mm, pt), timings=tm)
            wall = time.perf_counter() - t0
            p = list(paths.values())[0]
            out.append({"wall_s": wall, "stages_s": tm, "tsv_bytes": os.path.getsize(p)})
            print(json.dumps(out[-1]), file=sys.stderr, flush=True)
        print(json.dumps({"files_to_tsv": out}))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
