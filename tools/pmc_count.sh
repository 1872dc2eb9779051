#!/bin/bash
# SQ instruction counters of the count pass (full build and the -DABL ablations under build_abl/), count-only runs
set -e
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_count
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export HAWK_COUNT_ONLY=1
for v in ${PMC_VARIANTS:-full abl1 abl2}; do
  if [ $v != full ]; then export CRISPRHAWK_HIP_LIB=$root/build_abl/libhawk_$v.so; else unset CRISPRHAWK_HIP_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $out/$v -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $out/$v.log
done
cd $root
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
for v in os.environ.get("PMC_VARIANTS", "full abl1 abl2").split():
    fs = glob.glob(f"{root}/gpurun_out/pmc_count/{v}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Kernel_Name"].startswith("k_search_count"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(x) / len(x) for k, x in acc.items()}
    w = m.get("SQ_WAVES", 1)
    print(v, {k: round(x / w, 1) for k, x in m.items() if k != "SQ_WAVES"}, "waves", w)
PY
find $out -name "*counter_collection.csv" -size +5M -delete
