#!/bin/bash
# does the emit pass's placement dependence show in the address-translation counters?  (tools/cols_pad_probe.py under --pmc)
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/tlb_probe
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d $out/p -o t -- python3 $root/tools/cols_pad_probe.py 0 4096 65536 100000 0 > $out/probe.txt 2> $out/log.txt
cd $root
cat $out/probe.txt
python3 - <<'PY'
import csv, glob, os, collections
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
rows = []
for f in glob.glob(f"{root}/gpurun_out/tlb_probe/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_cs_emit"):
            rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
d = collections.defaultdict(dict)
for i, n, v in rows:
    d[i][n] = v
for k, i in enumerate(sorted(d)):
    m, h = d[i].get("TCP_UTCL1_TRANSLATION_MISS_sum", 0), d[i].get("TCP_UTCL1_TRANSLATION_HIT_sum", 0)
    if k % 8 == 0:
        print("dispatch", k, "miss", int(m), "hit", int(h), "miss rate %.4f" % (m / max(m + h, 1)))
PY
find $out -name "*counter_collection.csv" -size +5M -delete
