#!/bin/bash
# usage: ENVA="X=1" ENVB="X=0" scripts/ab_env.sh [bench args] -- same-box A/B of two environments, alternating, 3 reps
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then E="$ENVA"; else E="$ENVB"; fi
    env $E python bench.py --no-cpu-baseline "$@" > gpurun_out/abe_$v.json 2>/dev/null
    python -c "
import json
d=json.loads([l for l in open('gpurun_out/abe_$v.json') if l.startswith('{')][-1])
print('$v ($E) rep $rep: value %.4g ms/step %.4f kern_ms %.4f prunes %s rebuilds %s'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms'],d['config'].get('prunes_in_timed_region'),d['config'].get('rebuilds_in_timed_region')))"
  done
done
