#!/bin/bash
# A/B of the symmetric leader scan (GNMS_SCAN_V4=0: round 3, 1: round 4) on one box: two-calls and one-call steps, both generators
O=gpurun_out/ab_scan; mkdir -p $O
for v in 0 1; do for k in uniform clustered; do
  GNMS_SCAN_V4=$v python bench.py --two-calls --kind $k --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 > $O/tc_${k}_v$v.json
  GNMS_SCAN_V4=$v python bench.py --kind $k --steps 100 --warmup 10 --no-cpu-baseline --no-other-kind 2>/dev/null | tail -1 > $O/oc_${k}_v$v.json
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_scan/*.json')):
    try:
        d=json.load(open(f)); r=d.get('roofline') or {}
        print(f.split('/')[-1], d['ms_per_step'], r.get('kernel'), r.get('kernel_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
