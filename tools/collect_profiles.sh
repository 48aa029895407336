#!/bin/bash
# Round profile collection (run on the GPU box from the repo root):
#   GT4_COMMIT=<sha> GT4_ROUND=round4 bash tools/collect_profiles.sh <tag> [workload[@dist] ...]   (default: intersect c2 union8)
# Per workload: 1. plain bench line, 2. rocprofv3 --kernel-trace --stats of the same command,
# 3./4. FETCH_SIZE and WRITE_SIZE in their own --pmc passes (never mixed with other trace domains).
# `intersect` is the driver's default command (its line embeds the union8 record; the kernel summary therefore holds
# both kernels); its PMC passes run with --no-union8 so that the per-launch bytes are the intersection kernel's.
# Raw output goes to gpurun_out/prof_<tag>/<workload>/; tools/summarize_profiles.py turns it into
# the small files that are committed under profiles/.
set -u
TAG=${1:-r}
shift || true
WORKLOADS=${*:-intersect c2 union8}
ROOT=$(pwd)
export TMPDIR=/tmp
for SPEC in $WORKLOADS; do
  W=${SPEC%@*}
  D=stride
  if [ "$SPEC" != "$W" ]; then D=${SPEC#*@}; fi
  NAME=$W
  if [ "$D" != "stride" ]; then NAME=${W}_$D; fi
  OUT=$ROOT/gpurun_out/prof_$TAG/$NAME
  mkdir -p "$OUT"
  if [ "$W" = "intersect" ] && [ "$D" = "stride" ]; then BENCH="python3 $ROOT/bench.py"; PMCX="--no-union8 --no-extras"; else BENCH="python3 $ROOT/bench.py --workload $W --dist $D"; PMCX=""; fi
  if [ "$W" = "intersect" ] && [ "$D" != "stride" ]; then BENCH="$BENCH --no-union8"; fi
  if [ "$W" = "table32" ]; then BENCH="python3 $ROOT/bench.py --workload table --nt-lists 32 --nt 20000000"; fi # (the count table of 32 lists: km32, one launch)
  cd "$ROOT"
  timeout 900 $BENCH --steps 10 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- $BENCH --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/stats.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $BENCH $PMCX --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $BENCH $PMCX --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/write.log" 2>&1
  cd "$ROOT"
  python3 tools/summarize_profiles.py "$OUT" "${TAG}_$NAME" "$NAME" > "$OUT/summary.log" 2>&1
  cat "$OUT/bench.json"; cat "$OUT/summary.log"
  # keep only the small files (gpurun merges <= 64 MiB back)
  find "$OUT" -name '*_kernel_trace.csv' -size +4M -delete
  find "$OUT" -name '*_counter_collection.csv' -size +4M -delete
done
