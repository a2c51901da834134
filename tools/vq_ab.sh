#!/bin/bash
# Vert QP work aid: bit-identity hashes + config4v throughput in one GPU call.  usage: bash tools/vq_ab.sh <tag>
set -e
tag=${1:-x}
mkdir -p gpurun_out/vq
python tools/state_hash.py --batch 512 --steps 60 vertbox vertbox4 config4v > gpurun_out/vq/hash_$tag.txt 2>&1
cat gpurun_out/vq/hash_$tag.txt
python bench.py --workload config4v --no-continuous --no-cpu-baseline --min-seconds 3 > gpurun_out/vq/bench_$tag.json 2> gpurun_out/vq/bench_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/vq/bench_$tag.json')); print('config4v', d['value'], d['roofline']['resident_instances_per_cu'])"
