#!/bin/bash
# Collects the per-round evidence bench.py's roofline block refers to.  Run on the GPU box from the repo root:
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh r01'
# Outputs land in gpurun_out/<round>/ ; copy the summaries into profiles/ afterwards (see the end of this file).
set -e -o pipefail
R=${1:-r02}
LAT=${2:-16}                  # latent side and sampler: `collect_profiles.sh r04 32 ddim` is BASELINE configs[3]
KIND=${3:-ddpm}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
if [ "$LAT" != "16" ]; then OUT=$ROOT/gpurun_out/${R}_L$LAT; fi
TRACE_STEPS=100; if [ "$LAT" != "16" ]; then TRACE_STEPS=30; fi
SUFFIX=""; if [ "$LAT" != "16" ]; then SUFFIX="_L$LAT"; fi
mkdir -p "$OUT"
export TMPDIR=/tmp

# 1. kernel trace + stats of the same command at 100 diffusion steps (1000 steps = 157k dispatches per pass)
cd /tmp
HD_DUMP_OPS=$OUT/ops.txt timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
    python "$ROOT/bench.py" --steps 1 --warmup 0 --diffusion-steps $TRACE_STEPS --latent $LAT --kind $KIND --no-cpu-baseline > "$OUT/trace.log" 2>&1
cd "$ROOT"
TRACE=$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)
STATS=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
python tools/prof_summary.py "$TRACE" "$OUT/ops.txt" > "$OUT/kernel_trace_summary.txt"
python tools/kernel_table.py "$TRACE" "$OUT/ops.txt" 64 $LAT > "$OUT/kernel_table.txt"
cp "$STATS" "$OUT/kernel_stats.csv"
rm -f "$TRACE"                                     # tens of MB; the summary and stats are what is kept
grep "steps x" "$OUT/kernel_trace_summary.txt"
if [ -n "$ONLY_TRACE" ]; then echo "trace only"; exit 0; fi      # ONLY_TRACE=1: the per-kernel table, nothing else

# 2. HBM traffic: one counter per pass, two lengths, difference isolates the replay loop
for C in FETCH_SIZE WRITE_SIZE; do
  for N in 10 30; do
    cd /tmp
    timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_${C}_$N" -- \
        python "$ROOT/bench.py" --steps 1 --warmup 0 --diffusion-steps $N --latent $LAT --kind $KIND --no-cpu-baseline > "$OUT/pmc_${C}_$N.log" 2>&1
    cd "$ROOT"
  done
  A=$(find "$OUT/pmc_${C}_10" -name "*counter_collection.csv" | head -1)
  B=$(find "$OUT/pmc_${C}_30" -name "*counter_collection.csv" | head -1)
  python tools/pmc_traffic.py "$A" 10 "$B" 30 $C "$OUT/traffic.json" | tee -a "$OUT/traffic.txt"
  rm -rf "$OUT/pmc_${C}_10" "$OUT/pmc_${C}_30"
done
python - "$LAT" "$KIND" "$OUT/traffic.json" "$ROOT/profiles/${R}_traffic$SUFFIX.json" "$ROOT/profiles/traffic_latest$SUFFIX.json" <<'PY'
import json, sys
sys.path.insert(0, ".")
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "bench.py"); b = importlib.util.module_from_spec(spec)
src = open("bench.py").read()
ns = {}
exec(src[src.index("def kernel_source_hash"):src.index("def cpu_baseline")], {"os": __import__("os"), "ROOT": "."}, ns)
j = json.load(open(sys.argv[3]))
j["kernel_source_hash"] = ns["kernel_source_hash"](); j["latent"] = int(sys.argv[1]); j["kind"] = sys.argv[2]
if j["latent"] != 16:
    j.pop("algorithmic_bytes_per_step", None); j["source"] = j["source"].replace("latent 16", "latent %d" % j["latent"])
for out in sys.argv[4:]:
    json.dump(j, open(out, "w"), indent=1)
PY
# bench.py reads profiles/traffic_latest.json and reports it only while the kernel sources still hash to the recorded value

# 3. the headline bench line (includes the cpu_baseline leg)
if [ "$LAT" = "16" ]; then
  timeout -k 10 600 python bench.py --steps 3 --warmup 1 > "$OUT/bench_1gpu.json.log" 2> "$OUT/bench_1gpu.stderr"
else
  N=1000; if [ "$KIND" = "ddim" ]; then N=250; fi
  timeout -k 10 600 python bench.py --steps 3 --warmup 1 --latent $LAT --kind $KIND --diffusion-steps $N --no-cpu-baseline > "$OUT/bench_1gpu.json.log" 2> "$OUT/bench_1gpu.stderr"
fi
tail -c 600 "$OUT/bench_1gpu.json.log"; echo

# 4. the persistent stages of the program that was just timed, block by block against the oracle on their own inputs (so that the
#    report under profiles/ cannot go stale against the kernels: VERDICT r04 weak #1)
if [ "$LAT" = "16" ]; then
  for NB in 2 64; do
    timeout -k 10 900 python tools/op_forced.py --stages --batch $NB --out "$OUT/op_forced_stages_B$NB.txt" | tail -3
  done
fi

cp "$ROOT/profiles/${R}_traffic$SUFFIX.json" "$ROOT/profiles/traffic_latest$SUFFIX.json" "$OUT/" 2>/dev/null || true   # profiles/ does not travel back: gpurun_out/ does
echo done
