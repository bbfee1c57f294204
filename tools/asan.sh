#!/bin/bash
# Sanitizer pass over the HOST side of the library: the C-ABI shim, the pinned-memory mailboxes and
# their pollers (host_mailbox.hip, classic_nms.hip), the claim ring and the side streams (nms_layer.hip), the torch binding.
#   tools/asan.sh build address|thread   (build container) every .hip's host half + torch_binding.cpp with -fsanitize=..., device code
#                                        untouched (-fno-gpu-sanitize), into build/san_<kind>/
#   tools/asan.sh run address|thread     (GPU box: the scratch copy of the tree) swaps those two .so files into groomed_nms_amd/ and runs the
#                                        host-protocol tests under the sanitizer's runtime; report -> gpurun_out/r06_<kind>.txt
set -e
cd "$(dirname "$0")/.."
MODE=$1; KIND=${2:-address}
OUT=build/san_$KIND
C=groomed_nms_amd/csrc
LLVM=/opt/rocm/lib/llvm
RT=$(ls $LLVM/lib/clang/*/lib/linux/libclang_rt.$([ $KIND = address ] && echo asan || echo tsan)-x86_64.so | head -1)
if [ "$KIND" = binding ]; then
    # ROCm's ASAN runtime wraps hsa_amd_memory_pool_allocate and aborts at the first device allocation in this image (profiles/r06_asan.txt), so the
    # address pass covers what a host compiler can instrument without it: torch_binding.cpp -- the host shim every Python call goes through --
    # with g++ -fsanitize=address against the production library, under gcc's libasan
    OUT=build/san_binding
    if [ "$MODE" = build ]; then
        mkdir -p $OUT
        EXT=$(python -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
        INC=$(python -c "
import sysconfig
from torch.utils import cpp_extension as ce
print(' '.join('-I' + d for d in ce.include_paths(device_type='cuda') + [sysconfig.get_paths()['include'], '/opt/rocm/include']))")
        TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
        ABI=$(python -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
        g++ -O1 -g -fno-omit-frame-pointer -std=c++17 -fPIC -shared -fsanitize=address -Wno-deprecated-declarations -Wno-unknown-pragmas \
            -DTORCH_EXTENSION_NAME=gnms_torch -DTORCH_API_INCLUDE_EXTENSION_H -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI=$ABI $INC \
            $C/torch_binding.cpp -o $OUT/gnms_torch$EXT -L$TL -lc10 -lc10_hip -ltorch_cpu -ltorch_hip -ltorch -ltorch_python -Lgroomed_nms_amd -lgroomed_nms_hip \
            -Wl,-rpath,'$ORIGIN' -Wl,-rpath,$TL
        ls -la $OUT/*.so
        exit 0
    fi
    mkdir -p gpurun_out
    cp $OUT/gnms_torch*.so groomed_nms_amd/
    # (under a preloaded sanitizer runtime dlopen() no longer honours the caller's RUNPATH: torch's lazy initialisation then misses its own
    # libcaffe2_nvrtc.so, throws, and gcc's __cxa_throw interceptor -- installed before libstdc++ was mapped -- aborts: torch/lib on the
    # library path, libstdc++ preloaded behind libasan)
    export LD_LIBRARY_PATH=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))"):$LD_LIBRARY_PATH
    export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)"
    export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:log_path=gpurun_out/r06_asan_binding_raw
    timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider \
        -k "mailbox or slot or lazy or graphed or training_tail or aploss or best_targets or select_topk or one_launch or random_vs_oracle or api_edges or probabilities_only" 2>&1 | tail -8 > gpurun_out/r06_asan_binding_pytest.txt || true
    cat gpurun_out/r06_asan_binding_pytest.txt
    ( echo "g++ -fsanitize=address on torch_binding.cpp, runtime $LD_PRELOAD"; echo "reports (SUMMARY lines): $(cat gpurun_out/r06_asan_binding_raw.* 2>/dev/null | grep -c '^SUMMARY')"; cat gpurun_out/r06_asan_binding_raw.* 2>/dev/null | head -120 ) > gpurun_out/r06_asan_binding.txt
    head -40 gpurun_out/r06_asan_binding.txt
    exit 0
fi
if [ "$MODE" = build ]; then
    mkdir -p $OUT
    FL="--offload-arch=gfx950 -O1 -g -fno-omit-frame-pointer -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -fsanitize=$KIND -fno-gpu-sanitize -shared-libsan"
    pids=""
    for f in iou_kernels nms_layer soft_sort classic_nms nms_others aploss proposals host_mailbox; do
        /opt/rocm/bin/hipcc $FL -c $C/$f.hip -o $OUT/$f.o 2> $OUT/$f.log &
        pids="$pids $!"
    done
    for p in $pids; do wait $p; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=$KIND -fno-gpu-sanitize -shared-libsan -o $OUT/libgroomed_nms_hip.so $OUT/*.o
    EXT=$(python -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
    INC=$(python - <<'PY'
import sysconfig
from torch.utils import cpp_extension as ce
print(" ".join("-I" + d for d in ce.include_paths(device_type="cuda") + [sysconfig.get_paths()["include"], "/opt/rocm/include"]))
PY
)
    TL=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
    ABI=$(python -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
    $LLVM/bin/clang++ -O1 -g -fno-omit-frame-pointer -std=c++17 -fPIC -shared -fsanitize=$KIND -shared-libsan -Wno-deprecated-declarations -Wno-unknown-pragmas \
        -DTORCH_EXTENSION_NAME=gnms_torch -DTORCH_API_INCLUDE_EXTENSION_H -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI=$ABI $INC \
        $C/torch_binding.cpp -o $OUT/gnms_torch$EXT -L$TL -lc10 -lc10_hip -ltorch_cpu -ltorch_hip -ltorch -ltorch_python -L$OUT -lgroomed_nms_hip \
        -Wl,-rpath,'$ORIGIN' -Wl,-rpath,$TL
    ls -la $OUT/*.so
    exit 0
fi
# run (on the GPU box)
mkdir -p gpurun_out
cp $OUT/libgroomed_nms_hip.so $OUT/gnms_torch*.so groomed_nms_amd/
export LD_LIBRARY_PATH=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))"):$LD_LIBRARY_PATH
export LD_PRELOAD=$RT
if [ $KIND = address ]; then
    export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:log_path=gpurun_out/r06_asan_raw
else
    # (the interpreter, torch and the HIP runtime are not instrumented: their own synchronisation is invisible to the tool -- reports whose stacks
    # never enter libgroomed_nms_hip.so / gnms_torch are theirs)
    export TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:history_size=4:log_path=gpurun_out/r06_tsan_raw:ignore_noninstrumented_modules=1
fi
timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider \
    -k "mailbox or pinned or threads or graph or launcher or one_launch or slot or classic_nms or side_stream" 2>&1 | tail -15 > gpurun_out/r06_${KIND}_pytest.txt || true
cat gpurun_out/r06_${KIND}_pytest.txt
R=gpurun_out/r06_$([ $KIND = address ] && echo asan || echo tsan)
cat ${R}_raw.* 2>/dev/null | grep -c "^SUMMARY" > ${R}_count.txt || true
( echo "sanitizer: $KIND; runtime: $RT"; echo "reports (SUMMARY lines): $(cat ${R}_count.txt)"; cat ${R}_raw.* 2>/dev/null | grep -A12 "^==.*ERROR\|^WARNING: ThreadSanitizer" | grep -B2 -A12 "groomed_nms\|gnms_" | head -200 ) > ${R}.txt
cat ${R}.txt | head -60
