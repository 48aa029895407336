#!/bin/bash
# Round profile collection (run on the GPU box from the repo root):
#   GT4_COMMIT=<sha> bash tools/collect_profiles.sh <tag> [workload ...]      (default: intersect c2 union8)
# Per workload: 1. plain bench line, 2. rocprofv3 --kernel-trace --stats of the same command,
# 3./4. FETCH_SIZE and WRITE_SIZE in their own --pmc passes (never mixed with other trace domains).
# Raw output goes to gpurun_out/prof_<tag>/<workload>/; tools/summarize_profiles.py turns it into
# the small files that are committed under profiles/.
set -u
TAG=${1:-r}
shift || true
WORKLOADS=${*:-intersect c2 union8}
ROOT=$(pwd)
export TMPDIR=/tmp
for W in $WORKLOADS; do
  OUT=$ROOT/gpurun_out/prof_$TAG/$W
  mkdir -p "$OUT"
  BENCH="python3 $ROOT/bench.py --workload $W"
  cd "$ROOT"
  timeout 900 $BENCH --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/stats.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $BENCH --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1
  cd "$ROOT"
  python3 tools/summarize_profiles.py "$OUT" "${TAG}_$W" "$W" > "$OUT/summary.log" 2>&1
  cat "$OUT/bench.json"; cat "$OUT/summary.log"
  # keep only the small files (gpurun merges <= 64 MiB back)
  find "$OUT" -name '*_kernel_trace.csv' -size +4M -delete
  find "$OUT" -name '*_counter_collection.csv' -size +4M -delete
done
