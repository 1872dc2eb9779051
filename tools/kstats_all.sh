#!/bin/bash
# Top kernels by total time of a bench.py command line (GPU box, through gpurun from the repo root): tools/kstats_all.sh <tag> [bench args]
tag=${1:-x}; shift
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/kta_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o kt -- python3 $root/bench.py "$@" > $out/bench.json 2> $out/log.txt
cd $root
find $out -name "*kernel_trace.csv" -delete
python3 - $out > gpurun_out/kstats_all_$tag.txt <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -int(r["TotalDurationNs"]))
for r in rows[:40]:
    n = r["Name"].split("(")[0].replace("void ", "")
    print(f"{n:44s} {r['Calls']:>5s} total {int(r['TotalDurationNs']) / 1e6:9.3f} ms  avg {float(r['AverageNs']) / 1e3:9.1f} us")
PY
cat gpurun_out/kstats_all_$tag.txt
