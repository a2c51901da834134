#!/bin/bash
# instruction-cache and wait counters of the step kernel (diagnostic)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_icache; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT.a.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT.b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for sub in ("a","b"):
    fs = glob.glob(f"gpurun_out/prof_icache/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs: print(sub, "no counters (see log)"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "rkfd_step_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = sorted(v); print(k, v[len(v)//2])
PY
