#!/bin/bash
# usage: scripts/sweep_skin.sh  -- bench at several (skin, inner skin) settings; appends to gpurun_out/sweep.log
IFS=","; for cfg in ${SWEEP:-0.4 0.10}; do IFS=" "
set -- $cfg
echo "skin $1 inner $2" >> gpurun_out/sweep.log
MDHIP_INNER_SKIN=$2 timeout -k 10 100 python bench.py --steps 600 --warmup 100 --skin $1 --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['value']/1e9, d['ms_per_step'], d['roofline']['kernel_ms'], c['rebuilds_in_timed_region'], c['prunes_in_timed_region'], c['max_tile_halo'])" >> gpurun_out/sweep.log
done
