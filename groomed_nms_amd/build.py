"""Builds libgroomed_nms_hip.so (gfx950) in-tree with hipcc, and the C++ torch binding gnms_torch*.so (host compiler, links the
former).  `python -m groomed_nms_amd.build`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgroomed_nms_hip.so")
SOURCES = ["iou_kernels.hip", "nms_layer.hip", "soft_sort.hip", "classic_nms.hip", "nms_others.hip", "aploss.hip", "proposals.hip", "host_mailbox.hip"]
HEADERS = ["gnms_common.h", "gnms_prof.h", "iou_tile.h", "iou3d_pair.h", "iou3d_tile.h", "iou3d_sym.h", "nms_kernels.h", "nms_one_launch.h", "nms_backward_kernels.h", "nms_solve_kernels.h",
           os.path.join("..", "..", "include", "groomed_nms_hip.h")]
# -ffp-contract=off: products and sums round separately, like the reference's torch CPU kernels
# (lib/core.py:499-508) -- this is what makes the overlap matrices bit-identical.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"] + os.environ.get("GNMS_EXTRA_FLAGS", "").split()      # (developer experiments: -DGNMS_TIMING ...)


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    # objects compiled with other flags (GNMS_EXTRA_FLAGS experiments) are stale whatever their timestamps say
    stamp = os.path.join(CSRC, ".build_flags")
    flags = " ".join(FLAGS)
    if not os.path.exists(stamp) or open(stamp).read() != flags:
        force = True
    with open(stamp, "w") as f:
        f.write(flags)
    deps = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + deps):
            jobs.append([hipcc()] + FLAGS + ["-c", s, "-o", o])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
            if verbose and r.stderr.strip():
                print(r.stderr[-3000:])
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-6000:])
    return OUT


TORCH_SRC = os.path.join(CSRC, "torch_binding.cpp")


def torch_binding_path():
    import sysconfig
    return os.path.join(HERE, "gnms_torch" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_torch_binding(force=False, verbose=False):
    """The C++ autograd binding (csrc/torch_binding.cpp): one host-compiler invocation against torch's headers, in-tree so that it
    travels to the GPU box like the HIP library it links (rpath $ORIGIN).  No device code, no GPU needed."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    out = torch_binding_path()
    lib = build()
    deps = [TORCH_SRC, os.path.join(HERE, "..", "include", "groomed_nms_hip.h"), os.path.abspath(__file__), lib]
    if not force and not _stale(out, deps):
        return out
    cxx = os.environ.get("CXX", "g++")
    inc = []
    for d in ce.include_paths(device_type="cuda") + [sysconfig.get_paths()["include"], os.path.join(os.environ.get("ROCM_HOME", "/opt/rocm"), "include")]:
        if d not in inc:
            inc.append(d)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ([cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-deprecated-declarations", "-Wno-unknown-pragmas",
            "-DTORCH_EXTENSION_NAME=gnms_torch", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
            "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
           + ["-I" + d for d in inc] + [TORCH_SRC, "-o", out,
           "-L" + tlib, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
           "-L" + HERE, "-lgroomed_nms_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tlib])
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("torch binding build failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_torch_binding(force="--force" in sys.argv, verbose=True))
