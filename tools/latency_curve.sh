set -e
mkdir -p gpurun_out/r03b
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03b/gputest.log 2>&1 || true
tail -8 gpurun_out/r03b/gputest.log
for B in 256 512 1024 2048 2816 4096; do
  timeout -k 10 100 python bench.py --batch $B --split 1 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r03b/lat_$B.json 2> gpurun_out/r03b/lat_$B.err
  python3 -c "import json;r=json.load(open('gpurun_out/r03b/lat_$B.json'));print($B, r['value'], r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['resident_instances_per_cu'])"
done
