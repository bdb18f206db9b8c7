#!/bin/bash
# Host-side sanitizer pass (CPU container only; GPU ASan is not available on the pool): builds the C oracle with gcc's ASan+UBSan and
# the HIP library with the HOST half instrumented (-fsanitize=address,undefined -fno-gpu-sanitize) under /tmp/igcn_san, puts them in
# place of the shipped ones for the run, runs the CPU tests that call into them (plans, exports, oracle against the golden vectors,
# the gloo ranks), and puts the shipped files back.   bash scripts/sanitize_host.sh
# Preload only the ASan runtime (clang's carries the UBSan handlers; asan + ubsan_standalone together spin at start-up), and only
# on python — not on `timeout`.
set -u
cd "$(dirname "$0")/.."
S=/tmp/igcn_san; mkdir -p $S/o
R=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux | head -1)
restore() { [ -f $S/liboracle_c.orig.so ] && cp $S/liboracle_c.orig.so oracle/liboracle_c.so; [ -f $S/libigcn_hip.orig.so ] && cp $S/libigcn_hip.orig.so igcn_cf_amd/libigcn_hip.so && touch igcn_cf_amd/libigcn_hip.so; }
trap restore EXIT
python -c "import __graft_entry__ as g; g.build()" > /dev/null || exit 1
cp oracle/liboracle_c.so $S/liboracle_c.orig.so; cp igcn_cf_amd/libigcn_hip.so $S/libigcn_hip.orig.so
gcc -O1 -g -mavx2 -mfma -fopenmp -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $S/liboracle_c.so oracle/oracle_c.c || exit 1
cp $S/liboracle_c.so oracle/liboracle_c.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
  timeout -k 10 900 python -m pytest tests/test_host_cpu.py tests/test_oracle_golden.py -x -q -k "oracle" -p no:cacheprovider 2>&1 | tee $S/oracle.log | tail -2
cp $S/liboracle_c.orig.so oracle/liboracle_c.so
for f in spmm bpr score_topk topk_order sampler csr_util; do
  extra=""; [ $f = score_topk ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -fPIC -Iinclude -Iigcn_cf_amd/csrc -fsanitize=address,undefined -fno-gpu-sanitize \
    -fno-omit-frame-pointer $extra -c igcn_cf_amd/csrc/$f.hip -o $S/o/$f.o || exit 1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o $S/libigcn_hip.so $S/o/*.o || exit 1
cp $S/libigcn_hip.so igcn_cf_amd/libigcn_hip.so
timeout -k 10 1500 env LD_PRELOAD=$R/libclang_rt.asan-x86_64.so ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest tests/test_host_cpu.py tests/test_dist_cpu.py -x -q -m "not gpu" -p no:cacheprovider > $S/host.log 2>&1
echo "host tests rc=$? ; sanitizer reports: $(cat $S/oracle.log $S/host.log | grep -c 'runtime error\|AddressSanitizer')"; tail -2 $S/host.log
