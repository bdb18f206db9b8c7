"""Developer ablation of the top-k kernel: `build` (here, cross-compiling) makes variants of the library with
-DIGCN_X_* flags under igcn_cf_amd/_variants/; `run` (on the GPU box) times each on the full Amazon-like evaluation
in one process.   python scripts/dev_topk_variants.py build | run"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, 'igcn_cf_amd', '_variants')
VARIANTS = {'base': [], 'nohits': ['-DIGCN_X_NOHITS'], 'noselect': ['-DIGCN_X_NOSELECT', '-DIGCN_X_NOHITS'],
            'noloada': ['-DIGCN_X_NOLOADA'], 'bare': ['-DIGCN_X_NOSELECT', '-DIGCN_X_NOHITS', '-DIGCN_X_NOLOADA'],
            'stats': ['-DIGCN_TOPK_STATS'], 'sametile': ['-DIGCN_X_SAMETILE'],
            'turn6': ['-DIGCN_X_TURN6'], 'bare_sametile': ['-DIGCN_X_NOSELECT', '-DIGCN_X_NOHITS', '-DIGCN_X_SAMETILE'], 'todoonly': ['-DIGCN_X_TODOONLY'], 'noflush': ['-DIGCN_X_NOFLUSH']}


def build(only=None):
    from igcn_cf_amd import _build
    _build.build()
    os.makedirs(VDIR, exist_ok=True)
    objs = [os.path.join(_build.CSRC, f.replace('.hip', '.o')) for f in _build.SOURCES if f != 'score_topk.hip']
    for name, flags in VARIANTS.items():
        if only and name not in only:
            continue
        obj = os.path.join(VDIR, name + '.o')
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + _build.FLAGS + _build.EXTRA_FLAGS.get('score_topk.hip', []) + flags + ['-c', os.path.join(_build.CSRC, 'score_topk.hip'), '-o', obj])
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o',
                               os.path.join(VDIR, 'lib_%s.so' % name), obj] + objs)
        os.remove(obj)
        print('built', name, flush=True)


def run():
    import torch
    import igcn_cf_amd._lib as _lib
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.ops import score_topk
    from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr
    from scripts.dev_spmm_bench import time_ms
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
    g = torch.Generator(device='cuda').manual_seed(0)
    D = int(os.environ.get('D', 64))
    if os.environ.get('WIDE'):
        _lib.set_tuning('topk_fast_wide', int(os.environ['WIDE']))
    U = torch.randn(ds.n_users, D, device='cuda', generator=g) * 0.1
    I = torch.randn(ds.n_items, D, device='cuda', generator=g) * 0.1
    excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
    rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
    users = torch.arange(ds.n_users, device='cuda')
    names = sys.argv[2:] or list(VARIANTS)
    for name in names:
        _lib._handle, _lib._bound = C.CDLL(os.path.join(VDIR, 'lib_%s.so' % name)), {}
        if name == 'stats':
            buf = (C.c_ulonglong * 16)()
            for k in (1, 20):
                score_topk(U, I, k, user_ids=users, excl_rowptr=rp, excl_col=cl)
                _lib._handle.igcn_debug_topk_stats(buf, 1)
                ms = time_ms(lambda: score_topk(U, I, k, user_ids=users, excl_rowptr=rp, excl_col=cl), reps=1, warm=0)
                _lib._handle.igcn_debug_topk_stats(buf, 1)
                t = list(buf)
                import numpy as np
                wt = (C.c_ulonglong * (3 * 2048))()
                _lib._handle.igcn_debug_topk_wave_times(wt, 2048)
                w = np.array(list(wt), dtype=np.uint64).reshape(2048, 3)
                b0 = w[:, 0].min()
                beg, end = (w[:, 0] - b0).astype(np.float64) / 100, (w[:, 1] - b0).astype(np.float64) / 100   # us
                hw = w[:, 2]
                os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                np.save(os.path.join(ROOT, 'gpurun_out', 'census_k%d.npy' % k), np.stack([beg, end, hw.astype(np.float64)], axis=1))
                print(json.dumps(dict(census=dict(started_in_first_100us=int((beg < 100).sum()), begin_us_percentiles=[round(float(np.percentile(beg, q)), 1) for q in (0, 25, 50, 75, 100)],
                                                  life_us_percentiles=[round(float(np.percentile(end - beg, q)), 1) for q in (0, 25, 50, 75, 100)],
                                                  end_us_max=round(float(end.max()), 1),
                                                  simd_of_first_16=[int((x >> 4) & 3) for x in hw[:16]], cu_of_first_16=[int((x >> 8) & 15) for x in hw[:16]]))), flush=True)
                print(json.dumps(dict(raw=t)), flush=True)
                print(json.dumps(dict(variant=name, k=k, tiles=t[0], tiles_with_hits=t[1], hit_quads=t[2], flushes=t[3],
                                      flush_iterations=t[4], staged_drained=t[5], waves=t[6],
                                      ms=round(ms, 2), wave_us=t[11] / t[6] / 100, clock_GHz=round(t[9] / max(t[11], 1) / 10, 3),
                                      cycles_per_wave=dict(flush=t[7] // t[6], stage_hits=t[8] // t[6], wave=t[9] // t[6], build_masks=t[10] // t[6], first_63_tiles=t[14] // t[6], first_255_tiles=t[15] // t[6]))), flush=True)
            continue
        for tune in [dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in t.split(',') if kv) for t in os.environ.get('TUNES', '').split(';')]:   # e.g. TUNES=';topk_cap=8': defaults, then one knob
            for key, v in tune.items():
                _lib.set_tuning(key, v)
            rec = dict(variant=name, tune=tune)
            for k in ((20,) if os.environ.get('ONLY_K20') else (1, 20)):
                for masks in (False, True):
                    kw = dict(excl_rowptr=rp, excl_col=cl) if masks else {}
                    mode = os.environ.get('TOPK_MODE', 'exact')      # 'fast': the two-stage path (ablated builds: garbage lists, timing only)
                    rec['k%d_masks%d_ms' % (k, masks)] = round(time_ms(lambda: score_topk(U, I, k, user_ids=users, mode=mode, **kw), reps=3, warm=1), 2)
                    if mode == 'fast':
                        rec['k%d_masks%d_flagged' % (k, masks)] = score_topk.last_flagged
            for key in tune:
                _lib.set_tuning(key, None)
            print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:])
    else:
        run()
