#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters pass 1>" "<counters pass 2>" ...   (run on the GPU box)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_${tag}_$i
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --equil 60 --no-cpu-baseline > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/scripts/pmc_summary.py $out
done
