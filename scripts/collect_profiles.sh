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
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/r18" -o t -- python3 "$ROOT/bench.py" --workload inq_resnet18 --skip-extras --steps 10 --warmup 3 > "$OUT/r18_bench.json" 2> "$OUT/r18.err"
echo "resnet18 trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/ttq" -o t -- python3 "$ROOT/bench.py" --workload ttq_resnet20 --skip-extras --steps 30 --warmup 10 > "$OUT/ttq_bench.json" 2> "$OUT/ttq.err"
echo "ttq trace done"
CNT="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
timeout -s KILL 120 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/c18" -o p -- python3 "$ROOT/scripts/dev/conv_probe.py" 64 64 3 1 1 56 > "$OUT/c18.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/c20" -o p -- python3 "$ROOT/scripts/dev/conv_probe.py" 16 16 3 1 1 32 128 > "$OUT/c20.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d "$OUT/cst" -o p -- python3 "$ROOT/scripts/dev/conv_probe.py" 3 64 7 2 3 224 > "$OUT/cst.log" 2>&1
echo "conv counters done"
cd "$ROOT"
python3 scripts/trace_step.py "$(find "$OUT/r18" -name "t_kernel_trace.csv" | head -1)" inq_effective_kernel > "$OUT/resnet18_step_kernels.txt"
python3 scripts/trace_step.py "$(find "$OUT/ttq" -name "t_kernel_trace.csv" | head -1)" tern_absstat > "$OUT/ttq_step_kernels.txt"
for c in c18 c20 cst; do
    echo "== conv_probe $c ==" >> "$OUT/conv_counters.txt"
    python3 scripts/dev/pmc_rates.py "$(find "$OUT/$c" -name "*counter_collection.csv" | head -1)" | grep -v "^at::" >> "$OUT/conv_counters.txt"
done
rm -rf "$OUT/r18" "$OUT/ttq" "$OUT/c18" "$OUT/c20" "$OUT/cst"
ST=$(find "$OUT/stats" -name "b_kernel_stats.csv" | head -1)
TR=$(find "$OUT/stats" -name "b_kernel_trace.csv" | head -1)
cp "$ST" "$OUT/kernel_stats.csv"
python3 scripts/trace_step.py "$TR" > "$OUT/step_kernels.txt"
python3 scripts/pmc_summary.py "$(find "$OUT/fetch" -name "*counter_collection.csv" | head -1)" > "$OUT/pmc_fetch.txt"
python3 scripts/pmc_summary.py "$(find "$OUT/write" -name "*counter_collection.csv" | head -1)" > "$OUT/pmc_write.txt"
rm -rf "$OUT/stats" "$OUT/fetch" "$OUT/write"
echo "profiles collected"
