#!/usr/bin/env python3
"""K7 measurement on the scaled C5 configuration (BASELINE.json configs[4], SURVEY.md §8d: TTTV,
23-nt guides, --right, up to 4 mismatches; synthetic genome, seed 1006).  Not the driver's
bench (that is bench.py on C3): prints one JSON object with the kernel times of one
hawk_offtarget_scan over the whole genome.

    python tools/bench_offtargets.py [--genome-nt 100000000 --guides 1000 --mm 4 --pam TTTV --guidelen 23 --right]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "crispr-hawk_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-nt", type=int, default=100_000_000)
    ap.add_argument("--contigs", type=int, default=24)
    ap.add_argument("--guides", type=int, default=1000)
    ap.add_argument("--mm", type=int, default=4)
    ap.add_argument("--pam", default="TTTV")
    ap.add_argument("--guidelen", type=int, default=23)
    ap.add_argument("--right", action="store_true", default=True)
    ap.add_argument("--left", dest="right", action="store_false")
    args = ap.parse_args()
    from crisprhawk_hip.genome import GenomeIndex
    from crisprhawk_hip.pam import PAM

    rng = np.random.default_rng(1006)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    per = args.genome_nt // args.contigs
    t0 = time.time()
    contigs = {f"chr{i + 1}": acgt[rng.integers(0, 4, size=per)].tobytes() for i in range(args.contigs)}
    # guides: windows taken from the genome (so every guide has its on-target) 
    pam = PAM(args.pam, args.right, True)
    pam.encode(0)
    names = list(contigs)
    guides = []
    while len(guides) < args.guides:
        c = contigs[names[int(rng.integers(0, len(names)))]]
        p = int(rng.integers(0, per - 64))
        guides.append(c[p:p + args.guidelen].decode())
    t_syn = time.time() - t0
    t0 = time.time()
    idx = GenomeIndex(contigs, args.guidelen, len(pam))
    t_idx = time.time() - t0
    idx.scan(guides[:8], pam, args.right, args.mm)  # warm-up
    t0 = time.time()
    hits = idx.scan(guides, pam, args.right, args.mm)
    wall = time.time() - t0
    tm = idx.last_timing
    pairs = tm["n_sites"] * len(guides)
    print(json.dumps({
        "workload": f"scaled C5: {args.genome_nt} nt synthetic genome in {args.contigs} contigs (seed 1006), {len(guides)} guides, "
                    f"{args.pam} {args.guidelen} nt right={args.right}, mm<={args.mm}, bulges 0",
        "index_build_s": t_idx, "synth_s": t_syn, "hits": len(hits), "pam_sites": tm["n_sites"],
        "kernels_ms": {k: tm[k] for k in ("scan_ms", "sites_ms", "match_ms", "total_ms")},
        "scan_positions_per_s": tm["scanned_positions"] / (tm["scan_ms"] * 1e-3),
        "site_guide_compares_per_s": pairs / (tm["match_ms"] * 1e-3) if tm["match_ms"] else None,
        "wall_s_incl_download_and_python_rows": wall}))


if __name__ == "__main__":
    main()
