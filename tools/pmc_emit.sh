#!/bin/bash
# SQ instruction counters of k_emit_list (per wave)
set -e
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_emit
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $out/a -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $out/a.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/b -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $out/b.log || true
cd $root
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
for v in ("a", "b"):
    fs = glob.glob(f"{root}/gpurun_out/pmc_emit/{v}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    for kn in ("k_emit_list", "k_search_count"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if r["Kernel_Name"].startswith(kn):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        m = {k: sum(x) / len(x) for k, x in acc.items()}
        w = m.get("SQ_WAVES", 0)
        if w: print(kn, {k: round(x / w, 1) for k, x in m.items() if k != "SQ_WAVES"}, "waves", w)
        else: print(kn, {k: f"{x:.4g}" for k, x in m.items()})
PY
find $out -name "*counter_collection.csv" -size +5M -delete
