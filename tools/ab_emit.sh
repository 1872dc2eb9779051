#!/bin/bash
root=${GRAFT_REPO_ROOT:-$PWD}
for v in base eabl1 eabl2 base eabl1 eabl2; do
  if [ $v != base ]; then export CRISPRHAWK_HIP_LIB=$root/build_abl/libhawk_$v.so; else unset CRISPRHAWK_HIP_LIB; fi
  python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-collapse 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',round(d['ms_per_step'],3),{k:round(x,3) for k,x in d['kernels_ms'].items()})"
done
