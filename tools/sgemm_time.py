import sys, time, torch
sys.path.insert(0, ".")
from groomed_nms_amd.groomed_nms import _sgemm, soft_sort
for n in (1024, 4096, 8192):
    a = torch.rand((n, n), device="cuda") * 2 - 1
    b = torch.rand((n, n), device="cuda") * 2 - 1
    _sgemm(a, b); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): d = _sgemm(a, b)
    e.record(); e.synchronize()
    ms = s.elapsed_time(e) / 5
    ref = (a.double() @ b.double())
    err = float((d.double() - ref).abs().max())
    print(f"sgemm_mfma n={n}: {ms:.3f} ms  {2*n**3/ms/1e9:.1f} TFLOP/s  max|err|={err:.2e}")
    t0 = time.perf_counter(); torch.matmul(a, b); torch.cuda.synchronize()
    s.record()
    for _ in range(5): torch.matmul(a, b)
    e.record(); e.synchronize()
    print(f"   torch.matmul (rocBLAS/hipBLASLt) n={n}: {s.elapsed_time(e)/5:.3f} ms")
sc = torch.sort(torch.rand(4096, device="cuda"), descending=True)[0]
m = torch.rand((4096, 4096), device="cuda")
soft_sort(sc, m, 1e-4); torch.cuda.synchronize()
s.record(); soft_sort(sc, m, 1e-4); e.record(); e.synchronize()
print(f"soft_sort n=4096 total {s.elapsed_time(e):.3f} ms")
