#!/bin/bash
# Per-launch kernel names and times of one warm diffusion step (tools/op_kernels.py).  usage (GPU box, repo root): bash tools/op_times.sh [filter]
set -e
ROOT=$(pwd); OUT=$ROOT/gpurun_out/opk; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
HD_DUMP_OPS=$OUT/ops.txt timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python $ROOT/bench.py --steps 1 --warmup 0 --diffusion-steps 6 --no-cpu-baseline > $OUT/trace.log 2>&1
cd $ROOT
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/op_kernels.py $T $OUT/ops.txt "$1" > $OUT/op_kernels.txt
rm -rf $OUT/trace
