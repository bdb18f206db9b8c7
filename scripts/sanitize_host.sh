#!/bin/bash
# Host-side sanitizer pass (CPU container only; GPU ASan is not available on the pool): builds the C oracle with gcc's ASan+UBSan and
# the HIP library with the HOST half instrumented (-fsanitize=address,undefined -fno-gpu-sanitize) under /tmp/igcn_san and points the
# loaders at those copies (IGCN_ORACLE_LIB_PATH / IGCN_LIB_PATH) for the CPU tests that call into them (plans, exports, oracle
# against the golden vectors, the gloo ranks).  The shipped libraries are never touched (round 4 copied the instrumented builds over
# them and relied on an EXIT trap to put them back; a SIGKILL would have left an instrumented library in place).
#   bash scripts/sanitize_host.sh
# Preload only the ASan runtime (clang's carries the UBSan handlers; asan + ubsan_standalone together spin at start-up), and only
# on python — not on `timeout`.
set -u
cd "$(dirname "$0")/.."
S=/tmp/igcn_san; mkdir -p $S/o
R=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
python -c "import __graft_entry__ as g; g.build()" > /dev/null || exit 1
gcc -O1 -g -mavx2 -mfma -fopenmp -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $S/liboracle_c.so oracle/oracle_c.c || exit 1
IGCN_ORACLE_LIB_PATH=$S/liboracle_c.so LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
  timeout -k 10 900 python -m pytest tests/test_host_cpu.py tests/test_oracle_golden.py -x -q -k "oracle" -p no:cacheprovider 2>&1 | tee $S/oracle.log | tail -2
for f in spmm bpr score_topk topk_order sampler csr_util; do
  extra=""; [ $f = score_topk ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -Iigcn_cf_amd/csrc -fsanitize=address,undefined -fno-gpu-sanitize \
    -fno-omit-frame-pointer $extra -c igcn_cf_amd/csrc/$f.hip -o $S/o/$f.o || exit 1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o $S/libigcn_hip.so $S/o/*.o || exit 1
# (the -O1 device code of the instrumented build spills: the no-scratch test reads the SHIPPED library's code object, deselected here)
timeout -k 10 1500 env IGCN_LIB_PATH=$S/libigcn_hip.so LD_PRELOAD=$R/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/test_host_cpu.py tests/test_dist_cpu.py -x -q -m "not gpu" -k "not private_segment" -p no:cacheprovider > $S/host.log 2>&1
echo "host tests rc=$? ; sanitizer reports: $(cat $S/oracle.log $S/host.log | grep -c 'runtime error\|AddressSanitizer')"; tail -2 $S/host.log
