"""fp32 MFMA GEMM of the soft-sort path (gnms_sgemm) against torch.matmul (rocBLAS / hipBLASLt) on the same box.
python tools/sgemm_time.py [sizes ...]"""
import sys
import torch
sys.path.insert(0, ".")
from groomed_nms_amd.groomed_nms import _sgemm, soft_sort

sizes = [int(x) for x in sys.argv[1:]] or [1024, 4096, 8192]
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for n in sizes:
    a = torch.rand((n, n), device="cuda") * 2 - 1
    b = torch.rand((n, n), device="cuda") * 2 - 1
    for _ in range(3):
        _sgemm(a, b)
    torch.cuda.synchronize()
    s.record()
    for _ in range(10):
        d = _sgemm(a, b)
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / 10
    err = float((d.double() - (a.double() @ b.double())).abs().max()) if n <= 8192 else float("nan")
    for _ in range(3):
        torch.matmul(a, b)
    torch.cuda.synchronize()
    s.record()
    for _ in range(10):
        torch.matmul(a, b)
    e.record(); e.synchronize()
    mt = s.elapsed_time(e) / 10
    print(f"sgemm_mfma n={n}: {ms:.3f} ms  {2*n**3/ms/1e9:.1f} TFLOP/s = {2*n**3/ms/1e9/157.3:.3f} of the 157.3 TFLOP/s fp32 matrix peak  max|err|={err:.2e}"
          f"   | torch.matmul {mt:.3f} ms {2*n**3/mt/1e9:.1f} TFLOP/s")
sc = torch.sort(torch.rand(4096, device="cuda"), descending=True)[0]
m = torch.rand((4096, 4096), device="cuda")
for _ in range(3):
    soft_sort(sc, m, 1e-4)
torch.cuda.synchronize()
s.record()
for _ in range(10):
    soft_sort(sc, m, 1e-4)
e.record(); e.synchronize()
print(f"soft_sort n=4096 total {s.elapsed_time(e)/10:.3f} ms")
