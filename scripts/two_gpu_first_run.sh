#!/bin/bash
# The first thing to run on a box with TWO (or more) GPUs -- DESIGN.md section 6, "First action on a 2-GPU lease".
# Nothing of the multi-rank path has run between distinct devices yet (one-GPU boxes only): this script takes the
# measurements that decide the defaults, in the order in which a failure is cheapest to understand.
#   usage: bash scripts/two_gpu_first_run.sh [nranks=2]        (writes gpurun_out/two_gpu/)
set -u
N=${1:-2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/two_gpu
mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
run() { echo "== $*" | tee -a $O/log.txt; "$@" >> $O/log.txt 2>&1; echo "   rc=$?" | tee -a $O/log.txt; }

# 1. one-sided stores and flags between two PROCESSES on two DEVICES (the probe takes device rank % ndev)
hipcc -O3 --offload-arch=gfx950 -o /tmp/p2p_probe scripts/probe/p2p_probe.hip
run timeout -k 10 120 /tmp/p2p_probe $N 500 1500000
run timeout -k 10 120 /tmp/p2p_probe $N 2000 32

# 2. the slab worker over real RCCL between the devices: direct exchange, collectives only, overlapped
for mode in "MDHIP_DOM_P2P=1" "MDHIP_DOM_P2P=0" "MDHIP_DOM_P2P=1 MDHIP_DOM_OVERLAP=1"; do
  run env $mode MDHIP_DEBUG=1 DOM_BACKEND=nccl DOM_ASYNC=native DOM_PRUNE=1 DOM_NVT=1 DOM_N=110592 DOM_STEPS=120 DOM_KT=2.0 \
      timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=$N --master-addr 127.0.0.1 --master-port 29711 tests/domain_gpu_worker.py
done

# 3. the bench: weak scaling and config 4, each over the direct exchange / the collectives / the overlapped window
for cfg in "" "--config 4"; do
  for mode in "MDHIP_DOM_P2P=1" "MDHIP_DOM_P2P=0" "MDHIP_DOM_P2P=1 MDHIP_DOM_OVERLAP=1" "MDHIP_DOM_P2P=0 MDHIP_DOM_OVERLAP=1"; do
    tag=$(echo "n${N}_${cfg}_${mode}" | tr ' =-' '___')
    echo "== bench $cfg $mode" | tee -a $O/log.txt
    env $mode timeout -k 10 900 python3 bench.py --gpus $N --steps 200 --warmup 50 --no-cpu-baseline $cfg > $O/bench_$tag.json 2> $O/bench_$tag.err
    grep "^{" $O/bench_$tag.json | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('   value %.4g ms/step %.4f  %s'%(d['value'], d['ms_per_step'], d['config']['step_loop'][:60]))" | tee -a $O/log.txt
  done
done
python3 bench.py --gpus 1 --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err
grep "^{" $O/bench_n1.json | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('== 1 GPU: value %.4g ms/step %.4f'%(d['value'], d['ms_per_step']))" | tee -a $O/log.txt

# (no rocprofv3 pass here: the N-rank bench starts its ranks as child processes, and a profiler-preloaded parent must not
# exec; per-kernel times of a multi-rank run come from MDHIP_DOM_TIMING=1 and the bench line's step_breakdown_ms)
echo "done: $O/log.txt"
