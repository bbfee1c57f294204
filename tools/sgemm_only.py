"""Runs gnms_sgemm a few times at one size (for rocprofv3 --pmc passes).  python tools/sgemm_only.py [n | M N K]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groomed_nms_amd.groomed_nms import _sgemm
a = [int(x) for x in sys.argv[1:]] or [4096]
M, N, K = (a[0], a[0], a[0]) if len(a) == 1 else a[:3]
x = torch.rand((M, K), device="cuda") * 2 - 1
y = torch.rand((K, N), device="cuda") * 2 - 1
for _ in range(4):
    _sgemm(x, y)
torch.cuda.synchronize()
