#!/bin/bash
# Collects the evidence kept under profiles/ on the GPU box: bench line, rocprofv3 kernel statistics of the same
# command, the kernels of one training step, FETCH_SIZE / WRITE_SIZE of the streaming kernels (one --pmc pass each).
# usage (repo root): bash scripts/collect_profiles.sh   -> files under gpurun_out/prof/
set -e -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o b -- python3 "$ROOT/bench.py" --steps 30 --warmup 10 > "$OUT/bench_prof.json" 2> "$OUT/bench_prof.err"
echo "stats done"
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o f -- python3 "$ROOT/scripts/k1_only.py" > "$OUT/fetch.log" 2>&1
echo "fetch done"
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o w -- python3 "$ROOT/scripts/k1_only.py" > "$OUT/write.log" 2>&1
echo "write done"
cd "$ROOT"
ST=$(find "$OUT/stats" -name "b_kernel_stats.csv" | head -1)
TR=$(find "$OUT/stats" -name "b_kernel_trace.csv" | head -1)
cp "$ST" "$OUT/kernel_stats.csv"
python3 scripts/trace_step.py "$TR" > "$OUT/step_kernels.txt"
python3 scripts/pmc_summary.py "$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)" > "$OUT/pmc_fetch.txt"
python3 scripts/pmc_summary.py "$(find "$OUT/write" -name "*counter_collection.csv" | head -1)" > "$OUT/pmc_write.txt"
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
echo "profiles collected"
