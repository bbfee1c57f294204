#!/bin/bash
# builds with -DGNMS_TIMING on the box and prints the chain's phase ticks for uniform / clustered, push vs pull
GNMS_EXTRA_FLAGS="-DGNMS_TIMING" python -m groomed_nms_amd.build > /dev/null 2>&1
for k in uniform clustered; do for p in 100000 128; do echo "== $k GNMS_PULL_LEADERS=$p"; GNMS_PULL_LEADERS=$p python tools/phase_ticks.py --kind $k 2>&1 | grep slot; done; done
