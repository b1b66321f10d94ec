#!/bin/bash
# Per-launch GEMM mode sweep (HD_OP_MODE, hd_internal.hpp: add_gemm): one kernel trace per mode, the rows of the named launches side by side.
# usage (GPU box, repo root): bash tools/mode_sweep.sh <latent> <kind> "<key> <key> ..." "<mode> <mode> ..." ["<row regex>" [tag]]   -> gpurun_out/mode_sweep_L<latent><tag>.txt
# keys are substrings of launch names (HD_DUMP_OPS); the rows printed are those of tools/kernel_table.py matching the regex (default: rows starting with a key)
set -e -o pipefail
LAT=${1:-32}; KIND=${2:-ddim}; KEYS=${3:-"downs ups"}; MODES=${4:-"0 1 2 3 4 5 6"}; TAG=${6:-}
ROWS=${5:-"^($(echo $KEYS | tr ' ' '|'))"}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/mode_sweep_L$LAT; mkdir -p "$OUT"; export TMPDIR=/tmp
RES=$ROOT/gpurun_out/mode_sweep_L$LAT$TAG.txt; : > "$RES"
for M in default $MODES; do
  OV=""; if [ "$M" != "default" ]; then for K in $KEYS; do OV="$OV$K=$M,"; done; fi
  rm -rf "$OUT/trace"
  cd /tmp
  HD_EXPERIMENTS=1 HD_OP_MODE="$OV" HD_DUMP_OPS=$OUT/ops.txt timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- \
      python "$ROOT/bench.py" --steps 1 --warmup 0 --diffusion-steps 12 --latent $LAT --kind $KIND --no-cpu-baseline > "$OUT/trace.log" 2>&1
  cd "$ROOT"
  TRACE=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
  python tools/kernel_table.py "$TRACE" "$OUT/ops.txt" 64 $LAT > "$OUT/table_$M.txt"
  echo "== mode $M" >> "$RES"
  grep -E "$ROWS" "$OUT/table_$M.txt" >> "$RES" || true
  grep "^total" "$OUT/table_$M.txt" >> "$RES"
  echo "mode $M done"
done
rm -rf "$OUT/trace"
