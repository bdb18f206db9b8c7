"""Developer A/B of SpMM kernel variants in one process.
  python scripts/dev_spmm_variants.py build base= early=-DIGCN_X_EARLYADDS=1     (here: variant libraries under igcn_cf_amd/_variants/)
  python scripts/dev_spmm_variants.py run base early                              (GPU box: one launch and the 3-layer pass, ms)"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VDIR = os.path.join(ROOT, 'igcn_cf_amd', '_variants')


def build(specs):
    from igcn_cf_amd import _build
    _build.build()
    os.makedirs(VDIR, exist_ok=True)
    objs = [os.path.join(_build.CSRC, f.replace('.hip', '.o')) for f in _build.SOURCES if f != 'spmm.hip']
    for spec in specs:
        name, _, flags = spec.partition('=')
        obj = os.path.join(VDIR, 'spmm_' + name + '.o')
        subprocess.check_call(['/opt/rocm/bin/hipcc'] + _build.FLAGS + [f for f in flags.split(',') if f] +
                              ['-c', os.path.join(_build.CSRC, 'spmm.hip'), '-o', obj])
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o',
                               os.path.join(VDIR, 'libspmm_%s.so' % name), obj] + objs)
        os.remove(obj)
        print('built', name, flush=True)


def run(names):
    import torch
    import igcn_cf_amd._lib as _lib
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
    from igcn_cf_amd import ops
    from scripts.dev_spmm_bench import time_ms
    res = {}
    for preset in ('amazon', 'gowalla'):
        ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset})
        nu, ni = ds.n_users, ds.n_items
        rowptr, col, val = normalized_adjacency_host(ds.train_array, nu, ni)
        csr = CsrMatrix(rowptr, col, val, (nu + ni, nu + ni), 'cuda', order_blocks=[0, nu, nu + ni])
        x = torch.randn(nu + ni, 64, device='cuda') * 0.1
        y = torch.empty_like(x)
        t = {n: [] for n in names}
        tp = {n: [] for n in names}
        ref = None
        for rnd in range(5):
            for n in names:
                _lib._handle, _lib._bound = C.CDLL(os.path.join(VDIR, 'libspmm_%s.so' % n)), {}
                out = ops.propagate_mean(csr, x, 3)
                if ref is None:
                    ref = out.clone()
                err = float((out - ref).abs().max() / ref.abs().max())
                assert err < 1e-6, (n, err)
                t[n].append(time_ms(lambda: ops.spmm(csr, x, out=y), reps=100))
                tp[n].append(time_ms(lambda: ops.propagate_mean(csr, x, 3), reps=40))
        res[preset] = {n: {'spmm_ms': round(sorted(t[n])[2], 4), 'pass_ms': round(sorted(tp[n])[2], 4)} for n in names}
    print(json.dumps(res))


if __name__ == '__main__':
    {'build': build, 'run': run}[sys.argv[1]](sys.argv[2:])
