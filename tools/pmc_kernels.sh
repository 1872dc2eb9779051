#!/bin/bash
# Per-wave SQ counters of named kernels over a bench.py command line (run through gpurun from the repo root):
#   KERNELS="k_hx_build k_collapse" BENCH_ARGS="--config c4 ..." tools/pmc_kernels.sh <tag>
set -e
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmck_$tag
mkdir -p $out
args=${BENCH_ARGS:---config c4 --contig-len 9000000 --n-block-frac 0.1 --steps 1 --warmup 0 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/s$i -o p -- python3 $root/bench.py $args > /dev/null 2> $out/s$i.log
done
cd $root
KERNELS="${KERNELS:-k_hx_build}" OUT=$out python3 - <<'PY'
import csv, glob, collections, os
out = os.environ["OUT"]
names = tuple(os.environ["KERNELS"].split())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith(names):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(x) / len(x) for c, x in d.items()}
    w = m.get("SQ_WAVES", 1)
    print(k, "waves", w, {c: round(x / w, 1) for c, x in sorted(m.items()) if c != "SQ_WAVES"})
PY
find $out -name "*counter_collection.csv" -size +5M -delete
