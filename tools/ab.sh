#!/bin/bash
# A/B on one box: the in-tree library against build_abl/libhawk_${HAWK_AB_PREV:-prev}.so, count and emit times, alternating runs
root=${GRAFT_REPO_ROOT:-$PWD}
for i in 1 2; do
  for v in new prev; do
    if [ $v = prev ]; then export CRISPRHAWK_HIP_LIB=$root/build_abl/libhawk_${HAWK_AB_PREV:-prev}.so; else unset CRISPRHAWK_HIP_LIB; fi
    python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',round(d['ms_per_step'],3),{k:round(x,3) for k,x in d['kernels_ms'].items()})"
  done
done
