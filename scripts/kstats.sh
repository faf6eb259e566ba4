#!/bin/bash
# usage: scripts/kstats.sh <tag> [bench args...]   -- rocprofv3 kernel stats of bench.py (run on the GPU box)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ks_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/ks_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
grep "^{" gpurun_out/ks_$tag.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4g  ms/step %.4f  force_kernel_ms %.4f'%(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])); print(d['config'])"
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ks_$tag/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-56s calls=%5s avg_us=%9.1f tot_ms=%8.2f %5s%%"%(r["Name"].split("(")[0][-56:], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"]))
PY
