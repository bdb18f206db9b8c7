"""Builds libigcn_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so sits
next to the package so that it travels to the GPU box with the source tree."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libigcn_hip.so')
ROOF_LIB = os.path.join(PKG, 'libigcn_roof.so')
SOURCES = [f for f in ('spmm.hip', 'bpr.hip', 'score_topk.hip', 'topk_order.hip', 'sampler.hip', 'csr_util.hip')
           if os.path.exists(os.path.join(CSRC, f))]
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-Wall', '-Wno-unused-function',
         '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
# per source: the d = 128 candidate sweep runs one wave per SIMD with 396 registers; with MFMA results in VGPRs (this option)
# the selection reads the accumulators where they are, the item tiles and user planes spill over into AGPRs instead
# (left to itself the allocator puts the accumulators in AGPRs and copies 32 of them to VGPRs every tile step)
EXTRA_FLAGS = {'score_topk.hip': ['-mllvm', '-amdgpu-mfma-vgpr-form=1']}


def _newer(src_list, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, verbose=True):
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(ROOT, 'include', 'igcn_hip.h'))
    srcs = [os.path.join(CSRC, name) for name in SOURCES]
    # The shipped library is current (the GPU box gets the .so but not the objects: *.o is in .gpurunignore): nothing to do there
    main_current = not force and not _newer(srcs + headers, LIB)
    objs = []
    for name in ([] if main_current else SOURCES):
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            raise FileNotFoundError(src)
        obj = os.path.join(CSRC, name.replace('.hip', '.o'))
        if force or _newer([src] + headers, obj):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(name, []) + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if not main_current and (force or _newer(objs, LIB)):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    # measurement kernels of bench.py (the in-run gather roof): a library of their own, never loaded by the product
    roof_src = os.path.join(CSRC, 'roof_probe.hip')
    if os.path.exists(roof_src) and (force or _newer([roof_src], ROOF_LIB)):
        cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared', '-o', ROOF_LIB, roof_src]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
