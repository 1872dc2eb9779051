#!/bin/bash
# Per-kernel times of the default bench step (GPU box, through gpurun from the repo root): tools/kstats.sh <tag> [bench args]
#   -> gpurun_out/kstats_<tag>.txt  (kernels of the cluster dictionary and the cluster search: calls, average and minimum us)
tag=${1:-x}; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/kt_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $root/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-shares "$@" > /dev/null 2> $out/log.txt
cd $root
find $out -name "*kernel_trace.csv" -delete
python3 - $out > gpurun_out/kstats_$tag.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith(("k_cl", "k_cs", "k_mscan", "k_scan_u32", "k_rows", "__amd", "k_search", "k_emit")):
        print(f"{n:40s} {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}")
PY
cat gpurun_out/kstats_$tag.txt
