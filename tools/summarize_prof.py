#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + separate --pmc FETCH_SIZE / WRITE_SIZE passes)
into the per-round summary committed under profiles/.

usage: summarize_prof.py <prof_dir> <out.md> [title]
  <prof_dir>/kt/**/**_kernel_stats.csv, pmc_fetch/**/**_counter_collection.csv, pmc_write/...
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts wide
coalesced streaming reads at half their bytes (MI355X_MICROARCH.md §HBM), so the corrected
column doubles it for the streaming kernels (k_scan, k_emit, k_compact_*, k_search_*).
"""
import collections
import csv
import glob
import os
import sys

STREAMING = ("k_scan", "k_emit", "k_compact", "k_pack", "k_search_")  # NOT k_vsearch: its reads are 32-byte records at a 32-byte lane stride, for which the raw FETCH_SIZE matches the bytes known to be read (calibrated: 316 MB of records + 11 look-back records per tile = ~0.40 GB expected, 0.418 GB counted)


def short(name):
    n = name.replace("void ", "").split("(")[0]
    return n  # template arguments stay: k_vsearch<0> / k_vsearch<1> are two kernels


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(out)
    lines = [f"# {title}", ""]
    ks = glob.glob(os.path.join(d, "kt", "**", "*_kernel_stats.csv"), recursive=True)
    if ks:
        lines += ["## rocprofv3 --kernel-trace --stats (kernel_stats.csv)", "",
                  "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(ks[0])):
            lines.append(f"| {short(r['Name'])} | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |")
        lines.append("")
    pmc = {}
    for key, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        fs = glob.glob(os.path.join(d, sub, "**", "*_counter_collection.csv"), recursive=True)
        if not fs:
            continue
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] == key:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        pmc[key] = {k: sum(v) / len(v) for k, v in acc.items()}
    if pmc:
        lines += ["## HBM traffic per launch (separate --pmc passes, KiB -> bytes)", "",
                  "| kernel | FETCH_SIZE raw (MB) | FETCH corrected x2 for streaming reads (MB) | WRITE_SIZE (MB) | total (MB) |",
                  "|---|---|---|---|---|"]
        names = sorted(set(pmc.get("FETCH_SIZE", {})) | set(pmc.get("WRITE_SIZE", {})))
        for k in names:
            if k.startswith("__amd"):
                continue
            f = pmc.get("FETCH_SIZE", {}).get(k, 0.0) * 1024 / 1e6
            w = pmc.get("WRITE_SIZE", {}).get(k, 0.0) * 1024 / 1e6
            fc = 2 * f if k.startswith(STREAMING) else f
            lines.append(f"| {k} | {f:.1f} | {fc:.1f} | {w:.1f} | {fc + w:.1f} |")
        lines.append("")
    if pmc:  # machine-readable: what bench.py's roofline.traffic reads (profiles/r02_traffic.json, key "c3")
        import json
        tr = {}
        for k in sorted(set(pmc.get("FETCH_SIZE", {})) | set(pmc.get("WRITE_SIZE", {}))):
            if k.startswith("__amd"):
                continue
            f = pmc.get("FETCH_SIZE", {}).get(k, 0.0) * 1024
            tr[k] = {"fetch_bytes": 2 * f if k.startswith(STREAMING) else f, "write_bytes": pmc.get("WRITE_SIZE", {}).get(k, 0.0) * 1024}
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `bench.py --steps 10 --warmup 2`; KiB -> bytes; "
                             "FETCH_SIZE doubled for the streaming kernels per MI355X_MICROARCH.md (HBM section, gfx950 note)",
                   "c3": tr}, open(os.path.splitext(out)[0] + "_traffic.json", "w"), indent=1)
    for j in sorted(glob.glob(os.path.join(d, "bench_kt.json"))):
        lines += ["## bench.py line of the profiled run", "", "```json", open(j).read().strip(), "```", ""]
    open(out, "w").write("\n".join(lines))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
