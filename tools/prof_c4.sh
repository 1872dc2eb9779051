#!/bin/bash
# Per-kernel times of the C4 per-tile loop on a shortened contig (3 tiles): run through gpurun from the repo root.
#   tools/prof_c4.sh <tag>  ->  gpurun_out/prof_c4_<tag>/kt/kt_kernel_stats.csv, bench.json
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_c4_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/bench.py --config c4 --contig-len 13000000 --n-block-frac 0.077 --steps 1 --warmup 1 --no-cpu-baseline > $out/bench.json 2> $out/kt.log
cd $root
find $out -name "*_kernel_trace.csv" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/kt/kt_kernel_stats.csv")))
for r in rows[:16]:
    print(f"{r['Name'][:60]:60s} {r['Calls']:>5s} {int(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e6:8.3f} ms")
PY
