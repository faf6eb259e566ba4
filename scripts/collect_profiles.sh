#!/bin/bash
# usage (in the repo, after gpurun merged gpurun_out/prof_<tag>/ back): bash scripts/collect_profiles.sh r02
# copies the round's evidence set (scripts/profile_round.sh) into profiles/<tag>_* and prints the headline numbers
tag=${1:-rXX}
P=gpurun_out/prof_$tag
for pair in "bench_n1.json bench_n1.json" "bench_n1_driver_cmd.json bench_n1_steps20_warmup5.json" "bench_config4_n1.json bench_config4_n1.json" "bench_slab_path_world1_rccl.json bench_slab_path_world1_rccl.json" "bench_slab_path_world1_direct.json bench_slab_path_world1_direct.json" "bench_n1_frequency1.json bench_n1_frequency1.json" "bench_gpus2_launcher_gloo_one_gpu.json bench_gpus2_launcher_gloo_one_gpu.json" "bench_gpus2_native_direct_shim_one_gpu.json bench_gpus2_native_direct_shim_one_gpu.json"; do
  set -- $pair
  grep "^{" $P/$1 | tail -1 > profiles/${tag}_$2
  python3 - profiles/${tag}_$2 <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], "n_gpus %d value %.4g ms/step %.4f step_frac %.3f kernel frac %.3f kern %.4f valu %.3f traffic %s"%(d["n_gpus"],d["value"],d["ms_per_step"],d["step_roofline"]["frac_of_8TBps"],d["roofline"]["frac"],d["roofline"]["kernel_ms"],(d.get("valu_roofline") or {}).get("frac_of_measured") or 0, d["roofline"].get("traffic")), d["config"].get("rebuilds_in_timed_region"), d["config"].get("prunes_in_timed_region"), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline_1thread") or {}).get("value"))
PY
done
cp $P/kernel_stats.csv profiles/${tag}_kernel_stats.csv
cp $P/pmc_summary.txt profiles/${tag}_pmc_summary.txt
cp $P/traffic_k_step_tile.json profiles/${tag}_traffic_k_step_tile.json
python3 - $tag <<'PY'
import csv,sys
tag=sys.argv[1]
rows=list(csv.DictReader(open(f'profiles/{tag}_kernel_stats.csv')))
out=["# rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline --steps 400 --warmup 50` (650 steps incl. equilibration), final sources of the round","# name | calls | avg us | total ms | %"]
for r in rows[:24]:
    out.append("%-62s %6s %10.1f %9.2f %6s"%(r["Name"].split("(")[0][-62:], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, r["Percentage"][:5]))
open(f'profiles/{tag}_kernel_stats_summary.txt','w').write("\n".join(out)+"\n")
import json
sys.path.insert(0,'.')
import bench
print("kernel hash now", bench.kernel_hash(), "profile", json.load(open(f'profiles/{tag}_traffic_k_step_tile.json'))['kernel_sources_sha256_16'])
PY
