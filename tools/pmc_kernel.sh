#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <kernel-substring> <counter> [<counter> ...] -- <bench args>
# one rocprofv3 --pmc pass per counter (kernel-trace only) over a short bench run; prints the per-launch average for the kernel
TAG=$1; KEY=$2; shift 2
CS=()
while [ "$1" != "--" ]; do CS+=("$1"); shift; done
shift
export TMPDIR=/tmp
R=$PWD
for C in "${CS[@]}"; do
  D=$R/gpurun_out/pmck_${TAG}_$C
  mkdir -p $D
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace -d $D -o pmc --output-format csv -- python $R/bench.py "$@" > $D/stdout.txt 2> $D/stderr.txt || true)
  python - <<PY
import csv, glob
fs = glob.glob("$D/*counter_collection.csv")
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r.get("Counter_Name") == "$C" and "$KEY" in r["Kernel_Name"]] if fs else []
print("%-24s n=%3d avg=%16.1f" % ("$C", len(v), sum(v) / len(v) if v else 0))
PY
done
