"""Developer: the two-stage scoring when the users do not fill the chip (the candidate sweeps are then cut into pieces): waves
per CU the plan may use (8 = two per SIMD, the default; 4 = one per SIMD; 2), random-init tables, k = 20, no masks."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from igcn_cf_amd import _lib
from igcn_cf_amd.ops import score_topk
g = torch.Generator(device='cuda').manual_seed(0)
CASES = (('gowalla', 29858, 40988), ('yelp', 75173, 42706), ('amazon users / 8', 13716, 96421), ('amazon users / 2', 54865, 96421),
                    ('amazon', 109730, 96421), ('one test batch', 512, 96421))
if len(sys.argv) > 1 and sys.argv[1] == 'crossover':
    CASES = tuple(('%d users x %d items' % (b, n), b, n) for n in (96421, 40988, 10000) for b in (1024, 2048, 4096, 8192, 16384))
for tag, nu, ni in CASES:
    U = torch.randn(nu, 64, device='cuda', generator=g) * 0.1
    I = torch.randn(ni, 64, device='cuda', generator=g) * 0.1
    rec = dict(case=tag, users=nu, items=ni)
    ref = score_topk(U, I, 20, mode='exact')
    for tag, narrow, share in (('default', None, None), ('pieces_do_not_share_thresholds', None, 0), ('64_user_groups', 0, None)):
        _lib.set_tuning('topk_fast_narrow', narrow)
        _lib.set_tuning('topk_fast_share', share)
        a = score_topk(U, I, 20, mode='fast')
        assert torch.equal(a[0], ref[0]) and torch.equal(a[1], ref[1]), tag
        rec['ms_' + tag] = round(min(bench.time_ms(lambda: score_topk(U, I, 20, mode='fast'), 5, 2) for _ in range(2)), 3)
        rec['flagged_' + tag] = score_topk.last_flagged
    _lib.set_tuning('topk_fast_narrow', None)
    _lib.set_tuning('topk_fast_share', None)
    rec['ms_fp32_sweep'] = round(bench.time_ms(lambda: score_topk(U, I, 20, mode='exact'), 3, 1), 3)
    print(json.dumps(rec), flush=True)
