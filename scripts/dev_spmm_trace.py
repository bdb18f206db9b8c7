"""Developer tool: when does each wave of spmm_csr_rows_kernel start and finish?

Builds a second copy of the library with -DIGCN_SPMM_TRACE (never shipped) and prints the spread of
the waves' end times for one layer on the Amazon-like graph: a long tail = static row dealing is unbalanced."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRACE_LIB = os.path.join(ROOT, 'gpurun_out', 'libigcn_hip_spmm_trace.so')


def build():
    os.makedirs(os.path.dirname(TRACE_LIB), exist_ok=True)
    csrc = os.path.join(ROOT, 'igcn_cf_amd', 'csrc')
    srcs = [os.path.join(csrc, f) for f in ('spmm.hip', 'bpr.hip', 'score_topk.hip', 'sampler.hip', 'csr_util.hip')]
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared',
                           '-DIGCN_SPMM_TRACE'] + os.environ.get('IGCN_TRACE_FLAGS', '').split() +
                          ['-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-o', TRACE_LIB] + srcs)


def main():
    build()
    import torch
    import igcn_cf_amd._lib as _lib
    _lib.LIB_PATH = TRACE_LIB
    from igcn_cf_amd import ops
    from igcn_cf_amd.dataset import SyntheticDataset
    from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_host
    wt = _lib.handle().igcn_debug_spmm_wave_times
    wt.restype, wt.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_int]
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': os.environ.get('IGCN_TRACE_PRESET', 'amazon')})
    n = ds.n_users + ds.n_items
    rowptr, col, val = normalized_adjacency_host(ds.train_array, ds.n_users, ds.n_items)
    csr = CsrMatrix(rowptr, col, val, (n, n), 'cuda')
    x = torch.randn(n, 64, device='cuda') * 0.1
    for _ in range(3):
        y = ops.spmm(csr, x)
    torch.cuda.synchronize()
    nw = 131072
    buf = (C.c_uint64 * (6 * nw))()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = ops.spmm(csr, x)
    e1.record()
    torch.cuda.synchronize()
    wt(buf, nw)
    w = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 6).astype(np.int64)
    w = w[w[:, 1] > 0]
    t0 = w[:, 0].min()
    b, e = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0           # microseconds
    q = [0.0, 0.1, 0.5, 0.9, 0.99, 1.0]
    hw = w[:, 2]
    early = b < 10.0
    # HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    cu_key = (hw >> 8) & 0xff
    xcc = np.arange(len(w)) // 4 % 8          # workgroup id modulo 8 = XCD (round-robin dispatch)
    per_cu = {}
    for k2, x2, e2 in zip(cu_key, xcc, early):
        per_cu.setdefault((int(x2), int(k2)), [0, 0])
        per_cu[(int(x2), int(k2))][0 if e2 else 1] += 1
    counts = np.array(list(per_cu.values()))
    print(json.dumps(dict(early_fraction=round(float(early.mean()), 3), distinct_cu_keys=len(per_cu),
                          early_per_cu_quantiles=[int(v) for v in np.quantile(counts[:, 0], [0, 0.1, 0.5, 0.9, 1.0])],
                          late_per_cu_quantiles=[int(v) for v in np.quantile(counts[:, 1], [0, 0.1, 0.5, 0.9, 1.0])],
                          simd_of_late=np.bincount((hw[~early] >> 4) & 3, minlength=4).tolist(),
                          wave_slot_of_late=np.bincount(hw[~early] & 15, minlength=16).tolist())))
    busy = e - b
    A = np.stack([w[:, 3], w[:, 5], w[:, 4], np.ones(len(w))], axis=1).astype(np.float64)
    coef, *_ = np.linalg.lstsq(A, busy, rcond=None)
    pred = A @ coef
    print(json.dumps(dict(fit_busy_us='a*rows + b*chunks + c*nnz + e', a=round(float(coef[0]), 3), b=round(float(coef[1]), 3),
                          c=round(float(coef[2]), 4), e=round(float(coef[3]), 2),
                          r2=round(float(1 - ((busy - pred) ** 2).sum() / ((busy - busy.mean()) ** 2).sum()), 3),
                          rows_per_wave=[int(v) for v in np.quantile(w[:, 3], [0, 0.5, 1])],
                          nnz_per_wave=[int(v) for v in np.quantile(w[:, 4], [0, 0.1, 0.5, 0.9, 1])],
                          corr_busy_nnz=round(float(np.corrcoef(busy, w[:, 4])[0, 1]), 3))))
    print(json.dumps(dict(kernel_us=round(e0.elapsed_time(e1) * 1e3, 1), waves=int(len(w)),
                          begin_us=[round(float(v), 1) for v in np.quantile(b, q)],
                          end_us=[round(float(v), 1) for v in np.quantile(e, q)],
                          busy_us_mean=round(float((e - b).mean()), 1))))


if __name__ == '__main__':
    main()
