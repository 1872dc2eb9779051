#!/bin/bash
# Per-wave instruction counters of the two search kernels, in-tree library vs build_abl/libhawk_${HAWK_AB_PREV:-prev}.so
set -e
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_ab
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in new prev; do
  if [ $v = prev ]; then export CRISPRHAWK_HIP_LIB=$root/build_abl/libhawk_${HAWK_AB_PREV:-prev}.so; else unset CRISPRHAWK_HIP_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $out/$v -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $out/$v.log
done
cd $root
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
for v in ("new", "prev"):
    fs = glob.glob(f"{root}/gpurun_out/pmc_ab/{v}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith(("k_search_count", "k_emit_list")):
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        m = {c: sum(x) / len(x) for c, x in d.items()}
        w = m.get("SQ_WAVES", 1)
        print(v, k, {c: round(x / w, 1) for c, x in m.items() if c != "SQ_WAVES"}, "waves", w)
PY
find $out -name "*counter_collection.csv" -size +5M -delete
