"""Developer A/B (round 5): the scoring sweeps of one library build, timed through the raw C ABI — fp32 sweep at d = 64 / 128 / 256,
two-stage call at d = 64 / 128, Amazon-like sizes, random tables, train + val lists masked.  The library is the one IGCN_LIB_PATH
names (default: the shipped one); IGCN_EXPECT_ABI lets a build of an earlier round load.  One JSON line per measurement."""
import json
import os
import sys

import torch

sys.path.insert(0, '.')
from igcn_cf_amd import _lib                                             # noqa: E402
if os.environ.get('IGCN_EXPECT_ABI'):
    _lib.EXPECTED_ABI = int(os.environ['IGCN_EXPECT_ABI'])
from igcn_cf_amd.dataset import SyntheticDataset                         # noqa: E402
from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr        # noqa: E402


def time_ms(fn, reps, warm):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    tag = os.environ.get('IGCN_LIB_PATH', 'shipped')
    L = _lib.lib()
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
    excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
    rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
    k = 20
    for d, n_users in ((64, ds.n_users), (128, ds.n_users), (256, 32768)):
        g = torch.Generator(device='cuda').manual_seed(d)
        U = torch.randn(n_users, d, device='cuda', generator=g) * 0.1
        I = torch.randn(ds.n_items, d, device='cuda', generator=g) * 0.1
        out_idx = torch.empty((n_users, k), dtype=torch.int64, device='cuda')
        out_val = torch.empty((n_users, k), dtype=torch.float32, device='cuda')
        ws = torch.empty(max(L.igcn_score_topk_workspace_bytes(n_users, ds.n_items, d, k), 8), dtype=torch.uint8, device='cuda')
        flops = 2.0 * n_users * ds.n_items * d
        for masks in (() if os.environ.get('IGCN_SWEEPS_FAST_ONLY') else (False, True)):
            a = (rp.data_ptr(), cl.data_ptr()) if masks else (None, None)

            def exact():
                _lib.check(L.igcn_score_topk_f32(U.data_ptr(), U.stride(0), None, n_users, I.data_ptr(), I.stride(0), ds.n_items, d, a[0], a[1],
                                                 None, k, out_idx.data_ptr(), out_val.data_ptr(), ws.data_ptr(), _lib.current_stream()), 'exact')
            ms = min(time_ms(exact, 3, 1) for _ in range(3))
            print(json.dumps(dict(lib=tag, call='igcn_score_topk_f32', d=d, users=n_users, masks=masks, ms=round(ms, 3),
                                  tflops=round(flops / ms / 1e9, 1), frac_of_157=round(flops / ms / 1e9 / 157.3, 3))), flush=True)
        if d in (64, 128):
            wsf = torch.empty(L.igcn_score_topk_fast_workspace_bytes(n_users, ds.n_items, d, k, n_users, cl.numel()) + 256, dtype=torch.uint8, device='cuda')
            wp = (wsf.data_ptr() + 255) // 256 * 256
            flagged = torch.empty(n_users + 1, dtype=torch.int32, device='cuda')
            bounds = torch.empty(n_users, dtype=torch.float32, device='cuda')

            for masks in (True, False):
                ex = (rp.data_ptr(), cl.data_ptr(), n_users, cl.numel()) if masks else (None, None, 0, 0)

                def fast():
                    _lib.check(L.igcn_score_topk_fast_f32(U.data_ptr(), U.stride(0), None, n_users, I.data_ptr(), I.stride(0), ds.n_items, d,
                                                          ex[0], ex[1], ex[2], ex[3], None, k, out_idx.data_ptr(), out_val.data_ptr(),
                                                          flagged.data_ptr(), bounds.data_ptr(), wp, _lib.current_stream()), 'fast')
                samples = sorted(time_ms(fast, 5, 2) for _ in range(5))
                print(json.dumps(dict(lib=tag, call='igcn_score_topk_fast_f32', d=d, users=n_users, masks=masks, ms_min=round(samples[0], 3),
                                      ms_median=round(samples[2], 3), flagged=int(flagged[0]))), flush=True)


if __name__ == '__main__':
    main()
