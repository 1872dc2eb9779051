#!/bin/bash
# bench.py kernel times for several library variants on one box: in-tree ("new") and build_abl/libhawk_<v>.so for v in $VARIANTS
root=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2; do
  for v in new ${VARIANTS:-prev}; do
    if [ $v = new ]; then unset CRISPRHAWK_HIP_LIB; else export CRISPRHAWK_HIP_LIB=$root/build_abl/libhawk_$v.so; fi
    python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',round(d['ms_per_step'],3),{k:round(x,3) for k,x in d['kernels_ms'].items()}, d['config']['guide_rows_per_step'])"
  done
done
