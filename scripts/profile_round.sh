#!/bin/bash
# usage (on the GPU box): bash scripts/profile_round.sh r02   -- the round's evidence set, written under gpurun_out/prof_<tag>/
set -e
tag=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
cd $R
python bench.py --steps 200 --warmup 50 > $O/bench_n1.json 2> $O/bench_n1.err
echo "bench 200 done"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1_driver_cmd.json 2> $O/bench_n1_driver_cmd.err
echo "bench 20 done"
python bench.py --gpus 1 --config 4 --steps 100 --warmup 20 --no-cpu-baseline > $O/bench_config4_n1.json 2> $O/bench_config4_n1.err
echo "bench config4 done"
# the slab machinery with one rank (its own neighbour): over the direct peer exchange (the default) and over RCCL
MDHIP_BENCH_DOMAIN=1 python bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_slab_path_world1_direct.json 2> $O/bench_slab.err || echo "slab path bench failed"
MDHIP_DOM_P2P=0 MDHIP_BENCH_DOMAIN=1 python bench.py --steps 200 --warmup 50 --no-cpu-baseline > $O/bench_slab_path_world1_rccl.json 2> $O/bench_slab_rccl.err || echo "slab path bench (rccl) failed"
echo "bench slab done"
# energies every step (the reference's frequency = 1): one md_run call per step, the last step of each with U and W
python bench.py --steps 200 --warmup 50 --frequency 1 --no-cpu-baseline > $O/bench_n1_frequency1.json 2> $O/bench_f1.err
echo "bench frequency 1 done"
# bench.py --gpus 2 launches its two ranks itself (both on this box's one GPU: gloo carries the exchanges) -- the
# launcher's plumbing, not a scaling number
MDHIP_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_gpus2_launcher_gloo_one_gpu.json 2> $O/bench_g2.err || echo "2-rank launcher run failed"
# ... and the NATIVE two-rank loop over the direct peer exchange, both ranks on this one GPU (list-build collectives through
# tests/shim/libncclshim.so): 2 x 2^20 particles, the mailboxes mapped between the two processes -- functional, not a scaling number
hipcc -O2 -fPIC -shared --offload-arch=gfx950 -o tests/shim/libncclshim.so tests/shim/nccl_shim.cpp -lrt
MDHIP_BENCH_BACKEND=gloo MDHIP_DOM_LOOP=native MDHIP_RCCL_PATH=$R/tests/shim/libncclshim.so python3 bench.py --gpus 2 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_gpus2_native_direct_shim_one_gpu.json 2> $O/bench_g2n.err || echo "2-rank native run failed"
echo "bench --gpus 2 done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 $R/bench.py --no-cpu-baseline --steps 400 --warmup 50 > $O/ks.log 2>&1
cp $(ls $O/ks/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
echo "kernel stats done"
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc_$i -- python3 $R/bench.py --steps 20 --warmup 5 --equil 60 --no-cpu-baseline > $O/pmc_$i.log 2>&1
  python3 $R/scripts/pmc_summary.py $O/pmc_$i >> $O/pmc_summary.txt
  echo "pmc pass $i done"
done
cd $R
python3 scripts/make_traffic_json.py $O/pmc_summary.txt $O/traffic_k_step_tile.json
rm -rf $O/ks $O/pmc_[0-9]
echo "profile set complete"
