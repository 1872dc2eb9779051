#!/bin/bash
# Where the waves of the two search kernels spend their cycles: lifetime, waitcnt stalls, issue stalls, active cycles per unit
set -e
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_wait
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $out/s$i -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $out/s$i.log
done
cd $root
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/gpurun_out/pmc_wait/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith(("k_search_count", "k_emit_list", "k_scan_raw")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(x) / len(x) for c, x in d.items()}
    w = m.get("SQ_WAVES", 1)
    print(k, "waves", w, {c: round(x / w, 1) for c, x in sorted(m.items()) if c != "SQ_WAVES"})
PY
find $out -name "*counter_collection.csv" -size +5M -delete
