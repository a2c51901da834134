#!/bin/bash
# everything profiles/ holds for one round, on the GPU box: rocprofv3 evidence per workload, phase cycles, bench lines
# usage: bash tools/collect_all.sh [tag]      (then, in the build container: python3 tools/copy_profiles.py [tag])
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
PART=${PART:-all}      # the whole collection exceeds one 20-minute GPU call: PART=1 (counters, phases), PART=2 (bench lines, Volume, parity report)
if [ "$PART" != "2" ]; then
for W in ${WORKLOADS:-config4 config3 config2}; do
  bash tools/collect_traffic.sh $W $TAG > gpurun_out/collect_${W}.log 2>&1
  echo "collected $W"
done
# FETCH_SIZE / WRITE_SIZE against a known byte count in this kernel's access pattern (8 B per lane, 30-double rows)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${TAG}_cal/fetch -- tools/ubench/traffic_cal > gpurun_out/traffic_cal.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_${TAG}_cal/write -- tools/ubench/traffic_cal >> gpurun_out/traffic_cal.txt 2>&1
echo "traffic calibration done"
( echo "== rollout window (5 steps after 10) =="; WARM=10 python3 tools/prof_phases.py config2 config3 config4 config4v config5; echo "== rocking regime (5 steps after 100) =="; WARM=100 python3 tools/prof_phases.py config4 config5 ) > gpurun_out/phase_cycles.txt 2>&1
echo "phases done"
fi
if [ "$PART" = "1" ]; then exit 0; fi
python3 bench.py --warmup 5 --steps 20 > gpurun_out/bench_driver_cmd.json 2> gpurun_out/bench_driver_cmd.log
python3 bench.py --horizon 0 --no-cpu-baseline > gpurun_out/bench_config4_h0.json 2> gpurun_out/bench_config4_h0.log
python3 bench.py --workload config5 --steps 100 --warmup 20 > gpurun_out/bench_config5.json 2> gpurun_out/bench_config5.log
python3 bench.py --workload config5 --steps 100 --warmup 20 --horizon 0 --no-cpu-baseline > gpurun_out/bench_config5_h0.json 2> gpurun_out/bench_config5_h0.log
python3 bench.py --workload config4v --steps 100 --fuse 25 > gpurun_out/bench_config4v.json 2> gpurun_out/bench_config4v.log
python3 bench.py --workload config5v --steps 50 --warmup 5 --batch 1024 --min-seconds 2 --no-cpu-baseline > gpurun_out/bench_config5v.json 2> gpurun_out/bench_config5v.log
python3 bench.py --workload config1b --no-cpu-baseline > gpurun_out/bench_config1b.json 2> gpurun_out/bench_config1b.log
python3 bench.py --workload config3_26 --no-cpu-baseline > gpurun_out/bench_config3_26.json 2> gpurun_out/bench_config3_26.log
python3 bench.py --workload config4_26 --no-cpu-baseline > gpurun_out/bench_config4_26.json 2> gpurun_out/bench_config4_26.log
# the Volume plugin (the one the reference's drivers select): bench lines, kernel trace of the humanoid workload, phase cycles
python3 bench.py --workload config1_volume > gpurun_out/bench_config1_volume.json 2> gpurun_out/bench_config1_volume.log
python3 bench.py --workload config4_volume > gpurun_out/bench_config4_volume.json 2> gpurun_out/bench_config4_volume.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_config4_volume/trace -- python3 bench.py --workload config4_volume --no-cpu-baseline > gpurun_out/prof_${TAG}_config4_volume.bench.json 2> gpurun_out/prof_${TAG}_config4_volume.trace.log || true
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_config5/trace -- python3 bench.py --workload config5 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/prof_${TAG}_config5.bench.json 2> gpurun_out/prof_${TAG}_config5.trace.log || true
WARM=10 python3 tools/prof_phases.py config1_volume config4_volume > gpurun_out/phase_cycles_volume.txt 2>&1
echo "benches done"
tools/ubench/pgs 11 > gpurun_out/ubench_pgs.txt 2>&1
python3 tools/parity_report.py 1000 16 > gpurun_out/parity_report.txt 2>&1
echo "parity report done"
