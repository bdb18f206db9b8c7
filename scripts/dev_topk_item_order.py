"""Developer experiment: does the ORDER in which the candidate sweep meets the items matter?  The running thresholds
tighten sooner when likely winners (large-norm rows) come first, so fewer candidates are staged later.  Times the
two-stage and the fp32 sweeps (no masks) on the Amazon-like shapes with the item rows in the given order, sorted by
descending norm, and ascending (the worst case), for random-init and for a 'trained-like' table (row scales spread)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd.ops import score_topk

nu, ni, d, k = 109730, 96421, 64, 20
g = torch.Generator(device='cuda').manual_seed(0)
U = torch.randn(nu, d, device='cuda', generator=g) * 0.1
I0 = torch.randn(ni, d, device='cuda', generator=g) * 0.1
users = torch.arange(nu, device='cuda')


def timed(I, mode):
    score_topk(U, I, k, user_ids=users, mode=mode)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        score_topk(U, I, k, user_ids=users, mode=mode)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / 3, 3)


from igcn_cf_amd import _lib
for tag, I in (('random_init', I0), ('trained_like_row_scales_lognormal_0.5', I0 * torch.exp(0.5 * torch.randn(ni, 1, device='cuda', generator=g))),
               ('row_scales_lognormal_1.0', I0 * torch.exp(1.0 * torch.randn(ni, 1, device='cuda', generator=g)))):
    n2 = (I * I).sum(1)
    out = {'table': tag}
    for name, order in (('given', None), ('norm_descending', torch.argsort(n2, descending=True)), ('norm_ascending', torch.argsort(n2))):
        T = I if order is None else I[order].contiguous()
        out[name] = {'two_stage_ms': timed(T, 'fast'), 'flagged': score_topk.last_flagged, 'fp32_sweep_ms': timed(T, 'exact')}
    _lib.set_tuning('topk_fast_exit', 0)
    out['given_without_early_exit'] = {'two_stage_ms': timed(I, 'fast')}
    _lib.set_tuning('topk_fast_exit', None)
    print(json.dumps(out), flush=True)
