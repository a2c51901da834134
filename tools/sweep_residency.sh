#!/bin/bash
# saturated throughput against instances REALLY resident per CU: pad the LDS request (diagnostic env of the library)
# up to the next allocation piece counts.  The hardware hands LDS out in 1280-byte pieces, 128 per CU
# (tools/ubench/residency.hip), so n instances are resident when each needs at most floor(128/n) pieces.
W=${1:-config4}
B=${2:-16384}
LDS=$(python3 bench.py --workload $W --batch 256 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['roofline']['lds_bytes_per_instance'])")
for pieces in 10 11 12 14 16 18 21 25 32 42 64; do
  pad=$(( pieces*1280 - LDS ))
  if [ $pad -lt 0 ]; then continue; fi
  RKFD_LDS_PAD_BYTES=$pad timeout -k 10 200 python3 bench.py --workload $W --batch $B --split 3 --steps 60 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/res.json
  python3 - $pieces <<'PY'
import json, sys
d = json.load(open("/tmp/res.json")); r = d["roofline"]; p = int(sys.argv[1]); n = 128 // p
print("%2d pieces  LDS %6d B  %2d per CU  %.4f ms/step  %.3f M steps/s  %.0f k cycles per instance-step at 2.39 GHz" % (p, r["lds_bytes_per_instance"], n, d["ms_per_step"], d["value"]/1e6, n*256/d["value"]*2.39e9/1e3))
PY
done
