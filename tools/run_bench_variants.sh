#!/bin/bash
# usage: run_bench_variants.sh <outdir> "<bench args>" variant...   (one short bench line per library variant)
O=gpurun_out/$1; ARGS=$2; shift; shift; mkdir -p $O
for v in "$@"; do
  echo "== $v $ARGS" | tee -a $O/variants.log
  GT4HIP_LIB=$PWD/genometester4_amd/libgt4hip_$v.so timeout 300 python bench.py $ARGS --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print('ms/step %.3f kernel %.3f frac %.3f' % (b['ms_per_step'], b['roofline'].get('kernel_ms_avg') or 0, b['roofline']['frac']))" | tee -a $O/variants.log
done
