#!/bin/bash
# usage: tools/prof_cmd.sh <tag> <command...>   -- rocprofv3 kernel-trace stats of an arbitrary command, output under gpurun_out/prof_<tag>
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof_$TAG
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run --output-format csv -- "$@" > $R/gpurun_out/prof_$TAG/stdout.txt 2> $R/gpurun_out/prof_$TAG/stderr.txt || true
cd $R
python - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_$TAG/run_kernel_stats.csv')))
for r in rows[:14]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f} min_us={float(r['MinNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
