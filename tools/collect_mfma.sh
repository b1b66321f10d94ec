#!/bin/bash
# MFMA-utilisation evidence: one rocprofv3 --pmc pass of the bench command at latent 16 (1000-step DDPM
# configuration, 20 steps) and at latent 32 (250-step DDIM configuration, 20 steps).  Counters only with --kernel-trace.
#   gpurun --timeout 900 -- 'bash tools/collect_mfma.sh r03'        -> gpurun_out/<round>/mfma_*.txt / .json (copy to profiles/)
set -e -o pipefail
R=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INSTS_VALU_MFMA_MOPS_BF16\|SQ_BUSY_CYCLES\|GRBM_GUI_ACTIVE" | sort | uniq -c > "$OUT/mfma_counters_available.txt" || true
for CFG in "16 ddpm" "32 ddim"; do
  set -- $CFG
  L=$1; K=$2
  cd /tmp
  HD_DUMP_OPS=$OUT/ops_L$L.txt timeout -k 10 420 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace \
      --output-format csv -d "$OUT/pmc_mfma_L$L" -- python "$ROOT/bench.py" --steps 1 --warmup 0 --diffusion-steps 20 --latent $L --kind $K --no-cpu-baseline \
      > "$OUT/pmc_mfma_L$L.log" 2>&1
  cd "$ROOT"
  C=$(find "$OUT/pmc_mfma_L$L" -name "*counter_collection.csv" | head -1)
  python tools/pmc_mfma.py "$C" "$OUT/ops_L$L.txt" $L $K "$OUT/mfma_L$L.json" | tee "$OUT/mfma_L$L.txt"
  rm -rf "$OUT/pmc_mfma_L$L"
done
python - "$OUT" <<'PY'
import json, os, sys
src = open("bench.py").read()
ns = {}
exec(src[src.index("def kernel_source_hash"):src.index("def cpu_baseline")], {"os": os, "ROOT": "."}, ns)
out = {}
for L in (16, 32):
    j = json.load(open(os.path.join(sys.argv[1], "mfma_L%d.json" % L)))
    j["kernel_source_hash"] = ns["kernel_source_hash"]()
    out["L%d" % L] = j
json.dump(out, open(os.path.join(sys.argv[1], "mfma_latest.json"), "w"), indent=1)
PY
# copy mfma_latest.json to profiles/: bench.py reports roofline.mfma_busy_frac from it while the kernel sources hash to the recorded value
echo done
