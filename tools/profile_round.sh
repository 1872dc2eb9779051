#!/bin/bash
# Collect the per-round rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>     ->  gpurun_out/prof_<tag>/{kt,pmc_fetch,pmc_write}/..., bench_kt.json
# Kernel timing and the two PMC passes are separate runs (the counters do not fit one pass, and gpurun
# refuses --pmc together with the wider trace domains).
set -e
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
args="--steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end --no-shares"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/bench.py $args > $out/bench_kt.json 2> $out/kt.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o f -- python3 $root/bench.py $args > /dev/null 2> $out/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o w -- python3 $root/bench.py $args > /dev/null 2> $out/pmc_write.log
cd $root
python3 tools/summarize_prof.py $out $out/summary.md "profile $tag" > /dev/null
# keep the merge small: the raw traces are large
find $out -name "*_kernel_trace.csv" -delete
find $out -name "*_counter_collection.csv" -size +20M -delete
echo done
