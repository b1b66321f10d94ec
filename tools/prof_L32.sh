set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/${1:-r03}_L32; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
HD_DUMP_OPS=$OUT/ops.txt timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 1 --warmup 0 --latent 32 --kind ddim --diffusion-steps 30 --no-cpu-baseline > "$OUT/trace.log" 2>&1
cd $ROOT
TRACE=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
python tools/kernel_table.py "$TRACE" "$OUT/ops.txt" 64 32 > "$OUT/kernel_table.txt" 2>&1 || true
python tools/prof_summary.py "$TRACE" "$OUT/ops.txt" > "$OUT/summary.txt" 2>&1 || true
rm -f "$TRACE"
tail -60 $OUT/kernel_table.txt
