#!/bin/bash
# bash tools/pmc_run.sh <tag> <python script and args...>: FETCH_SIZE and WRITE_SIZE passes + per-kernel summary
TAG=$1; shift
export TMPDIR=/tmp; R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 "$@" > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 "$@" > $O/w.log 2>&1
cd $R; echo r > $O/bench.json; python3 tools/summarize_profiles.py $O $TAG > /dev/null; cat $O/summary/${TAG}_pmc*; grep "merge\|Error" $O/w.log
