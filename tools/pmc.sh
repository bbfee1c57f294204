#!/bin/bash
# usage: tools/pmc.sh <tag>  -- two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over a short bench run
TAG=$1
export TMPDIR=/tmp
R=$PWD
for C in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $R/gpurun_out/pmc_${TAG}_$C
  cd /tmp
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_${TAG}_$C -o pmc --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_${TAG}_$C/stdout.txt 2> $R/gpurun_out/pmc_${TAG}_$C/stderr.txt || true
  cd $R
  ls gpurun_out/pmc_${TAG}_$C | head
done
python - <<PY
import csv, glob, collections
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/pmc_${TAG}_%s/*counter_collection.csv" % C)
    if not fs: print(C, "no counter file"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") == C:
            agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("%-11s %-60s n=%3d avg=%12.1f" % (C, k, len(v), sum(v) / len(v)))
PY
