#!/bin/bash
# LDS bank-conflict ratio of the step kernel (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, median over the launches of the driver's
# command) and the in-kernel phase cycles, for the workloads given.  usage: bash tools/collect_lds_conflicts.sh <tag> [workload ...]
set -e
TAG=${1:-r03}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
for W in ${@:-config4}; do
  OUT=gpurun_out/$TAG/lds_$W
  rm -rf $OUT
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT -- python3 bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --min-seconds 0 > $OUT.log 2>&1
  python3 - "$OUT" "$W" <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.getcwd())
import rkfd_pkg
d, w = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rkfd_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in acc.items()}
med["workload"] = w; med["device_source_sha256"] = rkfd_pkg.device_source_hash()
if med.get("SQ_LDS_IDX_ACTIVE"):
    med["lds_bank_conflict_ratio"] = med["SQ_LDS_BANK_CONFLICT"] / med["SQ_LDS_IDX_ACTIVE"]
if med.get("SQ_WAVES"):
    med["lds_insts_per_instance_step"] = med["SQ_INSTS_LDS"] / med["SQ_WAVES"]; med["valu_insts_per_instance_step"] = med["SQ_INSTS_VALU"] / med["SQ_WAVES"]
    med["wait_any_over_wave_cycles"] = med["SQ_WAIT_ANY"] / med["SQ_WAVE_CYCLES"]
json.dump(med, open(d + "_summary.json", "w"), indent=1)
print(json.dumps(med))
PY
done
WARM=10 python3 tools/prof_phases.py ${@:-config4} > gpurun_out/$TAG/phase_cycles.txt 2>&1
cat gpurun_out/$TAG/phase_cycles.txt
