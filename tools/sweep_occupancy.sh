#!/bin/bash
# step time against instances per CU (single launch per step): how latency and throughput move with residency
W=${1:-config4}
for b in 256 512 1024 1536 2048 2304 2560 3072; do
  timeout -k 10 200 python3 bench.py --workload $W --batch $b --split 1 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/occ.json
  python3 - $b <<'PY'
import json, sys
d = json.load(open("/tmp/occ.json")); b = int(sys.argv[1])
print("batch %5d  %.1f per CU  %.4f ms/step  %.3f M steps/s" % (b, b/256, d["ms_per_step"], d["value"]/1e6))
PY
done
