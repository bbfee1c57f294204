#!/bin/bash
# usage: tools/pmc.sh <tag> [bench args...]
# Two separate PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as MI355X_MICROARCH.md prescribes) over a short bench run, then
# gpurun_out/pmc_<tag>_summary.json = {"traffic_bytes_per_launch": {kernel: bytes}} for `bench.py --pmc-summary`.
# Corrections (guide, HBM section): the counters are in KiB; FETCH_SIZE tallies 128-B requests of wide coalesced reads as 64 B on gfx950
# -> doubled; WRITE_SIZE is calibrated in the same run on gnms_profile_fill's known store stream (prof_fill_kernel writes exactly
# 4*B*N*N bytes per launch) and scaled by that factor.
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
for C in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $R/gpurun_out/pmc_${TAG}_$C
  cd /tmp
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_${TAG}_$C -o pmc --output-format csv -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-kind "$@" > $R/gpurun_out/pmc_${TAG}_$C/stdout.txt 2> $R/gpurun_out/pmc_${TAG}_$C/stderr.txt || true
  cd $R
done
python - "$TAG" "$@" <<'PY'
import csv, glob, collections, json, re, sys
tag = sys.argv[1]
args = sys.argv[2:]
def opt(name, default):
    return int(args[args.index(name) + 1]) if name in args else default
B, N = opt("--batch", 8), opt("--boxes", 4096)
raw = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/pmc_%s_%s/*counter_collection.csv" % (tag, C))
    agg = collections.defaultdict(list)
    if fs:
        for r in csv.DictReader(open(fs[0])):
            if r.get("Counter_Name") == C:
                name = re.sub(r"^void ", "", r["Kernel_Name"])
                name = re.sub(r"\(anonymous namespace\)::|gnms::", "", name)
                name = re.sub(r"[<(].*", "", name)
                agg[name].append(float(r["Counter_Value"]))
    raw[C] = {k: sum(v) / len(v) for k, v in agg.items()}
fill = raw["WRITE_SIZE"].get("prof_fill_kernel")
wcal = (4.0 * B * N * N / 1024.0) / fill if fill else 1.0          # WRITE_SIZE calibration on a known store stream
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --kernel-trace only; bench.py --steps 5 --warmup 2 " + " ".join(args),
       "correction": "KiB counters; FETCH_SIZE x2 (gfx950 tallies 128-B requests of wide reads as 64 B); WRITE_SIZE x %.4f (calibrated on prof_fill_kernel's 4*B*N*N bytes in the same run)" % wcal,
       "config": {"images_per_gpu": B, "boxes_per_image": N, "args": args}, "kernels": {}, "traffic_bytes_per_launch": {}}
for k in sorted(set(raw["FETCH_SIZE"]) | set(raw["WRITE_SIZE"])):
    if k.startswith(("at::", "prof_", "void at")) or "elementwise" in k:
        continue
    rd = raw["FETCH_SIZE"].get(k, 0.0) * 1024 * 2
    wr = raw["WRITE_SIZE"].get(k, 0.0) * 1024 * wcal
    out["kernels"][k] = {"FETCH_SIZE_KiB_raw": round(raw["FETCH_SIZE"].get(k, 0.0), 1), "WRITE_SIZE_KiB_raw": round(raw["WRITE_SIZE"].get(k, 0.0), 1),
                         "hbm_read_bytes": round(rd), "hbm_write_bytes": round(wr)}
    out["traffic_bytes_per_launch"][k] = round(rd + wr)
json.dump(out, open("gpurun_out/pmc_%s_summary.json" % tag, "w"), indent=1)
for k, v in sorted(out["traffic_bytes_per_launch"].items(), key=lambda kv: -kv[1])[:10]:
    print("%-40s %14d B/launch" % (k, v))
PY
