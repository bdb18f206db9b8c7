"""Developer tool: per-phase shader-clock breakdown of score_topk_kernel.

Builds a second copy of the library with -DIGCN_TOPK_TRACE (never shipped; the product
library has no trace code) and prints the average cycles a wave spends per 32-item tile in
load wait / MFMA chain / masking / selection — for kernel versions whose tile loop still has the four
stamps (v3-v5; the shipped pipelined loop is one pinned block and only reports the per-wave timeline:
begin / end / HW_ID of every wave, i.e. residency and balance)."""
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
                           '-DIGCN_TOPK_TRACE'] + os.environ.get('IGCN_TRACE_FLAGS', '').split() + [ '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-o', TRACE_LIB] + srcs)


def main():
    build()
    tag = os.environ.get('IGCN_TRACE_FLAGS', '')
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
    occ = _lib.handle().igcn_debug_topk_occupancy
    occ.restype, occ.argtypes = C.c_int, [C.c_int]
    print(json.dumps({'occupancy_api_blocks_per_cu': {str(b): occ(b) for b in (0, 5120, 10240)}}))
    wt = _lib.handle().igcn_debug_topk_wave_times
    wt.restype, wt.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_int]
    for cap, waves_per_simd in ((int(os.environ.get('IGCN_TRACE_CAP', '5')), int(os.environ.get('IGCN_TRACE_WPS', '3'))),):
      os.environ['IGCN_TOPK_CAP'], os.environ['IGCN_TOPK_WAVES'] = str(cap), str(waves_per_simd)
      masks = True
      if True:
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
        nw = min(int(t[6]), 8192)
        wbuf = (C.c_uint64 * (4 * nw))()
        wt(wbuf, nw)
        import numpy as np
        w = np.frombuffer(wbuf, dtype=np.uint64).reshape(nw, 4).astype(np.int64)
        t0 = w[:, 0].min()
        begin_ms, end_ms = (w[:, 0] - t0) / 1e5, (w[:, 1] - t0) / 1e5
        print(json.dumps(dict(cap=cap, waves_per_simd=waves_per_simd, waves=nw, started_within_0p2ms=int((begin_ms < 0.2).sum()),
                              begin_ms_quantiles=[round(float(x), 2) for x in np.quantile(begin_ms, [0.5, 0.6, 0.7, 0.8, 0.9, 1.0])],
                              end_ms_quantiles=[round(float(x), 2) for x in np.quantile(end_ms, [0.0, 0.1, 0.5, 0.9, 1.0])])))
        dur = (w[:, 1] - w[:, 0]) / 1e5
        hw = w[:, 2]
        simd, cu, se, xcc = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 13) & 7, np.arange(nw) % 8
        cands, excl = w[:, 3] & 0xffffffff, w[:, 3] >> 32
        def by(key):
            return {int(v): round(float(dur[key == v].mean()), 2) for v in np.unique(key)}
        order = np.argsort(dur)
        print(json.dumps(dict(dur_by_simd=by(simd), dur_by_xcc=by(xcc), dur_by_block_decile=by(np.arange(nw) * 10 // nw),
                              corr_dur_cands=round(float(np.corrcoef(dur, cands)[0, 1]), 3), corr_dur_excl=round(float(np.corrcoef(dur, excl)[0, 1]), 3),
                              cands_mean=float(cands.mean()), excl_mean=float(excl.mean()),
                              fastest=[(int(i), round(float(dur[i]), 2), int(cands[i]), int(excl[i]), int(simd[i])) for i in order[:4]],
                              slowest=[(int(i), round(float(dur[i]), 2), int(cands[i]), int(excl[i]), int(simd[i])) for i in order[-4:]])))
        tiles = max(t[5], 1)
        print(json.dumps(dict(flags=tag, masks=masks, ms=round(e0.elapsed_time(e1), 2), waves=t[6], tiles_per_wave=t[5] / max(t[6], 1),
                              per_tile=dict(load_wait=t[0] / tiles, chain=t[1] / tiles, mask=t[2] / tiles, select=t[3] / tiles),
                              wave_total_per_tile=t[4] / tiles,
                              wave_life_ms=t[7] / max(t[6], 1) / 1e5, shader_clock_GHz=t[4] / max(t[7], 1) * 0.1)), flush=True)


if __name__ == '__main__':
    main()
