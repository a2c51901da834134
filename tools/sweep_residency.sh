#!/bin/bash
# saturated throughput against instances resident per CU: pad the LDS request (diagnostic env of the library)
W=${1:-config4}
for pad in 0 1800 4200 7100 11000 16400 24600; do
  RKFD_LDS_PAD_BYTES=$pad timeout -k 10 200 python3 bench.py --workload $W --batch 16384 --split 3 --steps 60 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/res.json
  python3 - $pad <<'PY'
import json, sys
d = json.load(open("/tmp/res.json")); r = d["roofline"]
print("pad %6s  LDS %6d B  %2d per CU  %.4f ms/step  %.3f M steps/s" % (sys.argv[1], r["lds_bytes_per_instance"], r["resident_instances_per_cu"], d["ms_per_step"], d["value"]/1e6))
PY
done
