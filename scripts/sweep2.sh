#!/bin/bash
# usage: scripts/sweep2.sh  -- skin x inner-skin sweep of the bench workload (run on the GPU box), 200 timed steps each
cd $GRAFT_REPO_ROOT
for sk in 0.5 0.6 0.7; do
  for isk in 0.12 0.16 0.20; do
    MDHIP_INNER_SKIN=$isk python bench.py --no-cpu-baseline --steps 200 --warmup 50 --skin $sk > gpurun_out/sw_${sk}_${isk}.json 2>/dev/null
    python -c "
import json
d=json.loads([l for l in open('gpurun_out/sw_${sk}_${isk}.json') if l.startswith('{')][-1])
b=d['step_breakdown_ms']
print('skin $sk inner $isk: ms/step %.4f  ord %.4f prune %.4f builds %d prunes %d build_ms %.3f'%(d['ms_per_step'],b['ordinary_kernel'],b['prune_kernel'] or 0,d['config']['rebuilds_in_timed_region'],d['config']['prunes_in_timed_region'],b['list_build'] or 0))"
  done
done
