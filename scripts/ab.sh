#!/bin/bash
# usage: scripts/ab.sh [bench args]  -- same-box A/B of two library builds (csrc/libmdhip_A.so, libmdhip_B.so), alternating
cd $GRAFT_REPO_ROOT
D=moleculardynamics/jl_amd/csrc
for rep in 1 2 3; do
  for v in ${VARIANTS:-A B}; do
    cp $D/libmdhip_$v.so $D/libmdhip.so
    python bench.py --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2>/dev/null
    python -c "
import json
d=json.loads([l for l in open('gpurun_out/ab_$v.json') if l.startswith('{')][-1])
print('$v rep $rep: value %.4g ms/step %.4f kern_ms %.4f'%(d['value'],d['ms_per_step'],d['roofline']['kernel_ms']))"
  done
done
