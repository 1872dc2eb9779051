#!/bin/bash
# step / kernel times of hawk_csearch.hip's ablation builds (build_abl/libhawk_cs<n>.so, -DCS_ABL=n) next to the in-tree library
root=${GRAFT_REPO_ROOT:-$PWD}
for v in new ${VARIANTS}; do
  if [ $v = new ]; then unset CRISPRHAWK_HIP_LIB; else export CRISPRHAWK_HIP_LIB=$root/build_abl/libhawk_$v.so; fi
  python3 $root/bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',round(d['ms_per_step'],3),{k:round(x,3) for k,x in d['kernels_ms'].items() if k.startswith(('vsearch','view_t','offsets','device'))}, d['config']['guide_rows_per_step'])"
done
