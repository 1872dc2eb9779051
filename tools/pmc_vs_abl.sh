#!/bin/bash
# per-wave instruction counters of k_vsearch<0> for the in-tree library and the -DVS_ABL builds
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/pmc_vs_abl
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in new ${VARIANTS:-abl2 abl3}; do
  if [ $v = new ]; then unset CRISPRHAWK_HIP_LIB; else export CRISPRHAWK_HIP_LIB=$root/build_abl/libhawk_$v.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $out/$v -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2> $out/$v.log
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES --output-format csv -d $out/${v}_w -o p -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end > /dev/null 2>> $out/$v.log
done
cd $root
python3 - <<'PY'
import csv, glob, collections, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
for d in sorted(glob.glob(f"{root}/gpurun_out/pmc_vs_abl/*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0]
        if "k_vsearch" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in acc.items():
        m = {c: sum(x) / len(x) for c, x in dd.items()}
        w = m.get("SQ_WAVES", 1)
        print(os.path.basename(d.rstrip("/")), k, "waves", w, {c: round(x / w, 1) for c, x in sorted(m.items()) if c != "SQ_WAVES"})
PY
find $out -name "*counter_collection.csv" -size +5M -delete
