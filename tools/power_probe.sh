#!/bin/bash
# Samples GPU power, clocks and temperature while a bench.py workload runs (GPU box): tools/power_probe.sh c3 8
WL=${1:-c3}; STEPS=${2:-8}
mkdir -p gpurun_out
( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)|Performance" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/power_$WL.log 2>&1 &
SMI=$!
timeout -k 10 300 python -u bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu-baseline > gpurun_out/power_bench_$WL.log 2>&1
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
tail -1 gpurun_out/power_bench_$WL.log | cut -c1-200
