#!/usr/bin/env python3
"""K4 / K5 / K6 throughput on one MI355X: hawk_cfd, hawk_azimuth, hawk_deepcpf1 on random k-mers with the seeded
synthetic parameters.  Not the driver's bench; prints one JSON object.  The times include the H2D of
the k-mers and the D2H of the scores (these entry points take host buffers).

    python tools/bench_scorers.py [--n 1000000]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "crispr-hawk_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    args = ap.parse_args()
    from crisprhawk_hip import scoring, synth
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = {"n": args.n}

    def kmers(k):
        a = acgt[rng.integers(0, 4, size=(args.n, k))]
        return [bytes(r).decode() for r in a]

    # K6 DeepCpf1: conv 4->80 k=5 over 34 nt (30 x 80 x 20 MAC) + 1200x80 + 80x40 + 40x40 + 40 MACs
    scoring.set_deepcpf1_weights(synth.deepcpf1_weights())
    seqs = kmers(34)
    scoring.deepcpf1(seqs[:1000])
    t0 = time.perf_counter(); scoring.deepcpf1(seqs); dt = time.perf_counter() - t0
    macs = 30 * 80 * 20 + 1200 * 80 + 80 * 40 + 40 * 40 + 40
    out["deepcpf1"] = {"guides_per_s": args.n / dt, "s": dt, "GFLOPs": 2 * macs * args.n / dt / 1e9, "dtype": "f32"}

    # K5 Azimuth: 627 features on demand + 100 depth-3 trees
    from test_gpu_api import _random_gbt  # depth-3 complete trees in the flattened layout hawk_azimuth takes
    n_trees = 100
    model = _random_gbt(rng, n_trees)
    scoring.set_azimuth_model(model)
    seqs = kmers(30)
    scoring.azimuth(seqs[:1000])
    t0 = time.perf_counter(); scoring.azimuth(seqs); dt = time.perf_counter() - t0
    out["azimuth"] = {"guides_per_s": args.n / dt, "s": dt, "trees": n_trees, "dtype": "f64"}

    # K4 CFD on string triples
    mm, pt = synth.cfd_tables()
    scoring.set_cfd_tables(mm, pt)
    wt = kmers(20); sg = kmers(20); pam2 = [s[:2] for s in kmers(2)]
    scoring.compute_cfd_batch(wt[:1000], sg[:1000], pam2[:1000], True)
    t0 = time.perf_counter(); scoring.compute_cfd_batch(wt, sg, pam2, True); dt = time.perf_counter() - t0
    out["cfd"] = {"pairs_per_s": args.n / dt, "s": dt, "dtype": "f64"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
