#!/bin/bash
# everything profiles/ holds for one round, on the GPU box: rocprofv3 evidence per workload, phase cycles, bench lines
set -e
TAG=${1:-r01}
for W in config4 config3 config2; do
  bash tools/collect_traffic.sh $W $TAG > gpurun_out/collect_${W}.log 2>&1
  echo "collected $W"
done
WARM=100 python3 tools/prof_phases.py config2 config3 config4 config4v config5 > gpurun_out/phase_cycles.txt 2>&1
echo "phases done"
python3 bench.py --workload config5 --steps 100 --warmup 20 > gpurun_out/bench_config5.json 2> gpurun_out/bench_config5.log
echo "bench config5 done"
python3 bench.py --workload config4v --fuse 40 > gpurun_out/bench_config4v.json 2> gpurun_out/bench_config4v.log
echo "bench config4v done"
python3 tools/parity_report.py 1000 16 > gpurun_out/parity_report.txt 2>&1
echo "parity report done"
