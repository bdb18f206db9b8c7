"""Developer: event counts of the candidate sweep (stats build of the library) on TRAINED LightGCN tables, epoch by epoch:
tiles swept per wave (early exit), hit quads, staged candidates, flushes."""
import ctypes as C, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import igcn_cf_amd._lib as _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.model import get_model
from igcn_cf_amd.ops import score_topk
from igcn_cf_amd.trainer import get_trainer
dev = torch.device('cuda')
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': 'amazon', 'seed': 2021, 'device': dev})
torch.manual_seed(2021)
model = get_model({'name': 'LightGCN', 'embedding_size': 64, 'n_layers': 3, 'device': dev}, ds)
trainer = get_trainer({'name': 'BPRTrainer', 'optimizer': 'Adam', 'lr': 1e-3, 'l2_reg': 1e-5, 'device': dev, 'n_epochs': 1,
                       'batch_size': 2048, 'dataloader_num_workers': 0, 'test_batch_size': 512, 'topks': [20]}, ds, model)
users = torch.arange(ds.n_users, device=dev)
main_handle, main_bound = _lib.lib(), None
stats = C.CDLL(os.path.join(ROOT, 'igcn_cf_amd', '_variants', 'lib_stats.so'))
buf = (C.c_ulonglong * 16)()
for epoch in range(4):
    model.eval()
    with torch.no_grad():
        rep = model.get_rep().clone()
    U, I = rep[:ds.n_users], rep[ds.n_users:]
    saved = (_lib._handle, _lib._bound)
    _lib._handle, _lib._bound = stats, {}
    # the candidate sweep + re-scoring alone (the library's own fall-back switched off, no host fall-back): the per-wave records
    # of the sweep must not be overwritten by the bounded sweep's waves
    _lib.set_tuning('topk_fast_fallback', 0)
    L = _lib.lib()
    B, n_items, d, k = ds.n_users, ds.n_items, 64, 20
    ws = torch.empty(L.igcn_score_topk_fast_workspace_bytes(B, n_items, d, k, 0, 0) + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    out_idx = torch.empty((B, k), dtype=torch.int64, device=dev); out_val = torch.empty((B, k), dtype=torch.float32, device=dev)
    flagged = torch.empty(B + 1, dtype=torch.int32, device=dev); bounds = torch.empty(B, dtype=torch.float32, device=dev)

    def sweep():
        _lib.check(L.igcn_score_topk_fast_f32(U.data_ptr(), U.stride(0), users.data_ptr(), B, I.data_ptr(), I.stride(0), n_items, d, None, None, 0, 0,
                                              None, k, out_idx.data_ptr(), out_val.data_ptr(), flagged.data_ptr(), bounds.data_ptr(), ws_ptr,
                                              _lib.current_stream()), 'fast')
        torch.cuda.synchronize()
    sweep()
    stats.igcn_debug_topk_stats(buf, 1)
    sweep()
    stats.igcn_debug_topk_stats(buf, 1)
    n_flagged = int(flagged[0])
    t = list(buf)
    import numpy as np
    wt = (C.c_ulonglong * (3 * 2048))()
    stats.igcn_debug_topk_wave_times(wt, 2048)
    w = np.array(list(wt), dtype=np.uint64).reshape(2048, 3)[:1715]
    tiles_w, flushes_w = ((w[:, 2] >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64), (w[:, 2] >> np.uint64(48)).astype(np.int64)
    life = (w[:, 1] - w[:, 0]).astype(np.float64) / 100          # us
    start = (w[:, 0] - w[:, 0].min()).astype(np.float64) / 100
    _lib._handle, _lib._bound = saved
    print(json.dumps(dict(epochs_trained=epoch, waves=t[6], tiles_per_wave=round(t[0] / max(t[6], 1), 1), tiles_with_hits_per_wave=round(t[1] / max(t[6], 1), 1),
                          hit_quads_per_wave=round(t[2] / max(t[6], 1), 1), flushes_per_wave=round(t[3] / max(t[6], 1), 1),
                          staged_per_wave=round(t[5] / max(t[6], 1), 1), wave_life_us_quantiles_10_50_75_90_99_max=[round(float(np.percentile(life, q)), 1) for q in (10, 50, 75, 90, 99, 100)],
                          wave_start_us_max=round(float(start.max()), 1), waves_alive_after_us={str(t_): int((life > t_).sum()) for t_ in (200, 300, 400, 500, 600, 700)},
                          flagged=n_flagged, longest_waves_us_tiles_flushes=[[round(float(life[i]), 1), int(tiles_w[i]), int(flushes_w[i])] for i in np.argsort(-life)[:6]],
                          cycles_per_wave=dict(flush=t[7] // max(t[6], 1), stage_hits_incl_flush=t[8] // max(t[6], 1), wave=t[9] // max(t[6], 1), build_masks=t[10] // max(t[6], 1), before_first_tile=t[12] // max(t[6], 1), after_last_tile=t[13] // max(t[6], 1)),
                          clock_GHz=round(t[9] / max(t[11], 1) / 10, 3), median_wave_us_tiles_flushes=[round(float(np.median(life)), 1), int(np.median(tiles_w)), int(np.median(flushes_w))])), flush=True)
    if n_flagged and epoch >= 2:
        # the bounded fp32 sweep of the handed-over users alone (host-side call: no threshold pooling), same stats build
        pos = flagged[1:1 + n_flagged].long()
        stats.igcn_debug_topk_stats(buf, 1)
        saved2 = (_lib._handle, _lib._bound)
        _lib._handle, _lib._bound = stats, {}
        score_topk(U, I, 20, user_ids=users[pos].contiguous(), mode='exact', lower_bound=bounds[:n_flagged].contiguous())
        torch.cuda.synchronize()
        stats.igcn_debug_topk_stats(buf, 1)
        _lib._handle, _lib._bound = saved2
        t = list(buf)
        wt = (C.c_ulonglong * (3 * 2048))()
        stats.igcn_debug_topk_wave_times(wt, 2048)
        nw = int(t[6])
        w = np.array(list(wt), dtype=np.uint64).reshape(2048, 3)[:nw]
        life = (w[:, 1] - w[:, 0]).astype(np.float64) / 100
        print(json.dumps(dict(bounded_sweep_of_handed_over_users=n_flagged, waves=nw, tiles_per_wave=round(t[0] / max(nw, 1), 1), hit_quads_per_wave=round(t[2] / max(nw, 1), 1),
                              flushes_per_wave=round(t[3] / max(nw, 1), 1), drained_per_wave=round(t[5] / max(nw, 1), 1),
                              cycles_per_wave=dict(flush=t[7] // max(nw, 1), stage_hits_incl_flush=t[8] // max(nw, 1), wave=t[9] // max(nw, 1), build_masks=t[10] // max(nw, 1),
                                                   before_first_tile=t[12] // max(nw, 1), after_last_tile=t[13] // max(nw, 1)),
                              wave_life_us_quantiles_10_50_90_max=[round(float(np.percentile(life, q)), 1) for q in (10, 50, 90, 100)])), flush=True)
    model.train()
    trainer.train_one_epoch()
