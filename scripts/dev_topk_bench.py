"""Developer micro-benchmark of the fused score+mask+top-k kernel."""
import json
import sys

import torch

sys.path.insert(0, '.')
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import _csr_to_device, _merge_sorted_csr
from scripts.dev_spmm_bench import time_ms


def main():
    ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon'})
    d = 64
    g = torch.Generator(device='cuda').manual_seed(0)
    U = torch.randn(ds.n_users, d, device='cuda', generator=g) * 0.1
    I = torch.randn(ds.n_items, d, device='cuda', generator=g) * 0.1
    excl = _merge_sorted_csr(ds.csr('train'), ds.csr('val'))
    rp, cl = _csr_to_device(excl[0], excl[1], 'cuda')
    users = torch.arange(ds.n_users, device='cuda')
    flops = 2.0 * ds.n_users * ds.n_items * d
    for k in (1, 5, 20, 50):
        for masks in (False, True):
            kw = dict(excl_rowptr=rp, excl_col=cl) if masks else {}
            ms = min(time_ms(lambda: score_topk(U, I, k, user_ids=users, **kw), reps=3, warm=1) for _ in range(2))
            print(json.dumps(dict(k=k, masks=masks, ms=round(ms, 2), users_per_s=round(ds.n_users / ms * 1e3),
                                  tflops=round(flops / ms / 1e9, 1))), flush=True)
    # batch-size sensitivity (item-range splits kick in for small batches)
    for B in (512, 4096, 32768):
        ms = min(time_ms(lambda: score_topk(U, I, 20, user_ids=users[:B].contiguous(), excl_rowptr=rp, excl_col=cl),
                         reps=5, warm=1) for _ in range(2))
        print(json.dumps(dict(B=B, k=20, ms=round(ms, 3), users_per_s=round(B / ms * 1e3))), flush=True)


if __name__ == '__main__':
    main()
