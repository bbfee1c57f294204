#!/bin/bash
# PMC passes over the fp32 MFMA GEMM (one counter group per pass, --kernel-trace only).  usage: tools/sgemm_pmc.sh [n | M N K]
# (SQ_VALU_MFMA_BUSY_CYCLES = MFMA instructions x 64 cycles, summed over the SIMDs -- exactly 2^31 at 4096^3 because every dimension is a
# power of two, not a saturated counter: 3072^3 gives 905 969 664.  GRBM_GUI_ACTIVE is summed over the 8 XCDs.  MFMA utilisation =
# SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).)
export TMPDIR=/tmp
R=$PWD
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "MemUnitStalled" "LDSBankConflict" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES"; do
  T=$(echo $C | tr " " "_")
  mkdir -p $R/gpurun_out/pmc_gemm_$T
  cd /tmp
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_gemm_$T -o pmc --output-format csv -- python $R/tools/sgemm_only.py ${@:-4096} > /dev/null 2>&1 || true
  cd $R
  python - "$T" <<'PY'
import csv, glob, sys, collections
for f in glob.glob("gpurun_out/pmc_gemm_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sgemm" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(k, "launches", len(v), "avg %.1f" % (sum(v) / len(v)))
PY
done
