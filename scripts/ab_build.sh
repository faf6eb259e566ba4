#!/bin/bash
# usage: scripts/ab_build.sh [bench args] -- same-box A/B of two library builds, reporting the list-build time too
cd $GRAFT_REPO_ROOT
D=moleculardynamics/jl_amd/csrc
for rep in 1 2 3; do
  for v in ${VARIANTS:-A B}; do
    cp $D/libmdhip_$v.so $D/libmdhip.so
    python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2>/dev/null
    python -c "
import json
d=json.loads([l for l in open('gpurun_out/ab_$v.json') if l.startswith('{')][-1])
b=d['step_breakdown_ms']
print('$v rep $rep: ms/step %.4f  ord %.4f prune %.4f build %.4f'%(d['ms_per_step'],b['ordinary_kernel'],b['prune_kernel'] or 0,b['list_build'] or 0))"
  done
done
