"""Runs gnms_sgemm a few times at one size (for rocprofv3 --pmc passes).  python tools/sgemm_only.py [n]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groomed_nms_amd.groomed_nms import _sgemm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
a = torch.rand((n, n), device="cuda") * 2 - 1
b = torch.rand((n, n), device="cuda") * 2 - 1
for _ in range(4):
    _sgemm(a, b)
torch.cuda.synchronize()
