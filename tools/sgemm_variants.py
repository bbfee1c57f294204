"""The fp32 GEMM behind the soft sort, path by path on one box (gnms_profile_sgemm): this library's MFMA kernels, the same with the large
kernel de-phased, rocBLAS through the library's own dispatch, and torch.matmul.   python tools/sgemm_variants.py [sizes ...]"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from groomed_nms_amd import _lib
from groomed_nms_amd._lib import ptr, check, stream_ptr
lib = _lib.load()
dev = torch.device("cuda")
sizes = [int(x) for x in sys.argv[1:]] or [512, 1024, 2048, 4096, 8192]
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps

for n in sizes:
    a = torch.rand((n, n), device=dev) * 2 - 1
    b = torch.rand((n, n), device=dev) * 2 - 1
    d = torch.empty((n, n), device=dev)
    ref = (a.double() @ b.double()) if n <= 4096 else None
    reps = 40 if n <= 2048 else 10
    out = []
    for name, variant in (("own MFMA kernels", 1), ("own, de-phased", 3), ("rocBLAS via the library", 2), ("hipBLASLt via the library", 4), ("the product path", 0)):
        ms = timed(lambda: check(lib.gnms_profile_sgemm(ptr(a), ptr(b), ptr(d), n, n, n, n, n, n, variant, stream_ptr(dev)), "sgemm"), reps)
        err = float((d.double() - ref).abs().max()) if ref is not None else float("nan")
        out.append("%s %.3f ms %.1f TF (max|err| %.1e)" % (name, ms, 2 * n ** 3 / ms / 1e9, err))
    mt = timed(lambda: torch.matmul(a, b, out=d), reps)
    out.append("torch.matmul %.3f ms %.1f TF" % (mt, 2 * n ** 3 / mt / 1e9))
    print("n=%d: " % n + " | ".join(out), flush=True)
