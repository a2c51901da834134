#!/bin/bash
# Collects the rocprofv3 evidence for one workload on the GPU box (run from the repo root):
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes, then SQ counters.
# usage: tools/collect_traffic.sh <workload> <tag>
set -e
W=${1:-config4}; TAG=${2:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_${W}
rm -rf $OUT && mkdir -p $OUT
# counter passes: the driver's command, but one timed block only (--min-seconds 0): PMC collection serialises the dispatches
ARGS="bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 0"
# kernel trace + stats of the SAME command whose JSON line is reported (default steps / warmup), so that
# the average kernel duration can be compared with bench.py's own HIP-event figure
# (--no-continuous: the continuous-trajectory record of the default line launches the same kernel on a lighter contact regime,
#  which would pull the trace's average below the rollouts' figure; the default line itself is bench_driver_cmd.json)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --workload $W --no-continuous > $OUT.bench.json 2> $OUT.trace.log
# the counter passes run the lane mapping the traced run chose, without measuring it again: the tuner's launches (both kernels)
# would otherwise outnumber the timed ones in the per-launch medians
IPW=$(python3 -c "import json; print(json.loads(open('$OUT.bench.json').read().strip().splitlines()[-1])['config']['instances_per_wavefront'])")
ARGS="$ARGS --ipw $IPW"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT.fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT.write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/sq1 -- python3 $ARGS > $OUT.sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F64 --output-format csv -d $OUT/sq2 -- python3 $ARGS > $OUT.sq2.log 2>&1 || true
python3 tools/parse_prof.py $OUT $W
