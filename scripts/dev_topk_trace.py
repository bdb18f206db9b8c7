"""Developer tool: per-phase shader-clock breakdown of score_topk_kernel.

Builds a second copy of the library with -DIGCN_TOPK_TRACE (never shipped; the product
library has no trace code) and prints the average cycles a wave spends per 32-item tile in
load wait / MFMA chain / masking / selection."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, 'gpurun_out', 'libigcn_hip_trace.so')


def build():
    os.makedirs(os.path.dirname(TRACE_LIB), exist_ok=True)
    csrc = os.path.join(ROOT, 'igcn_cf_amd', 'csrc')
    srcs = [os.path.join(csrc, f) for f in ('spmm.hip', 'bpr.hip', 'score_topk.hip', 'sampler.hip', 'csr_util.hip')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared',
                           '-DIGCN_TOPK_TRACE', '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-o', TRACE_LIB] + srcs)


def main():
    build()
    import torch
    import igcn_cf_amd._lib as _lib
    _lib.LIB_PATH = TRACE_LIB
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.ops import score_topk
    from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr
    dbg = _lib.handle().igcn_debug_topk_trace
    dbg.restype, dbg.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_int]
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
    g = torch.Generator(device='cuda').manual_seed(0)
    U = torch.randn(ds.n_users, 64, device='cuda', generator=g) * 0.1
    I = torch.randn(ds.n_items, 64, device='cuda', generator=g) * 0.1
    excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
    rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
    users = torch.arange(ds.n_users, device='cuda')
    buf = (C.c_uint64 * 8)()
    for masks in (True, False):
        kw = dict(excl_rowptr=rp, excl_col=cl) if masks else {}
        score_topk(U, I, 20, user_ids=users, **kw)
        dbg(buf, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        score_topk(U, I, 20, user_ids=users, **kw)
        e1.record()
        torch.cuda.synchronize()
        dbg(buf, 1)
        t = list(buf)
        tiles = max(t[5], 1)
        print(json.dumps(dict(masks=masks, ms=round(e0.elapsed_time(e1), 2), waves=t[6], tiles_per_wave=t[5] / max(t[6], 1),
                              per_tile=dict(load_wait=t[0] / tiles, chain=t[1] / tiles, mask=t[2] / tiles, select=t[3] / tiles),
                              wave_total_per_tile=t[4] / tiles)), flush=True)


if __name__ == '__main__':
    main()
