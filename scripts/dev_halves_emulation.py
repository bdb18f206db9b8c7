"""Developer A/B: the d = 64 SpMM as two column halves pinned to XCD parities (an XCD's L2 then only ever holds one
128-byte half of every 256-byte operand row), 4 row slices x 2 halves instead of 8 row slices.  Emulated with the
production kernel, no kernel change: M' = M (x) I_2 on the operand viewed as [2n, 32] computes the same product, and
list 2 i + s of its plan holds the rows 2 r + s of slice-list i of M's 4-list plan."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from igcn_cf_amd import _lib
from igcn_cf_amd.dataset import SyntheticDataset
from igcn_cf_amd.graph import CsrMatrix, normalized_adjacency_device, xcd_plan
from igcn_cf_amd.ops import spmm

dev = torch.device('cuda')
preset = sys.argv[1] if len(sys.argv) > 1 else 'amazon'
ds = SyntheticDataset({'name': 'SyntheticDataset', 'preset': preset})
nu, n = ds.n_users, ds.n_users + ds.n_items
base = normalized_adjacency_device(ds.train_array, ds.n_users, ds.n_items, dev)
x = torch.randn(n, 64, device=dev) * 0.1
y0 = torch.empty_like(x)
t0 = min(bench.time_ms(lambda: spmm(base, x, out=y0), 50, 5) for _ in range(3))
print(json.dumps(dict(case='production (8 row slices)', us=round(t0 * 1e3, 1))), flush=True)
rowptr, col, val = base.rowptr, base.col.long(), base.val
lens = rowptr[1:] - rowptr[:-1]
nnz = int(col.shape[0])
i64 = dict(dtype=torch.int64, device=dev)


def doubled(threshold, n_slices, assign, parity_lists=True):
    lr, sg, order, xoff, _ = xcd_plan(rowptr, col, [0, nu, n], threshold, min(threshold, 256), assign=assign, n_lists=n_slices)
    # M' = M (x) I_2: row 2r + s holds row r's nonzeros with columns 2c + s
    lens2 = torch.repeat_interleave(lens, 2)
    rp2 = torch.zeros(2 * n + 1, **i64)
    rp2[1:] = torch.cumsum(lens2, 0)
    row_e = torch.repeat_interleave(torch.arange(n, **i64), lens)
    j = torch.arange(nnz, **i64) - rowptr[row_e]
    col2 = torch.empty(2 * nnz, dtype=torch.int32, device=dev)
    val2 = torch.empty(2 * nnz, dtype=torch.float32, device=dev)
    for s in (0, 1):
        p = 2 * rowptr[row_e] + s * lens[row_e] + j
        col2[p] = (2 * col + s).int()
        val2[p] = val
    n_seg = sg.shape[0]
    seg_start = sg[:, 0:2].contiguous().view(torch.int64).reshape(-1)
    seg_len, seg_row = sg[:, 2].long(), sg[:, 4].long()
    first, count = lr[:, 1].long(), lr[:, 2].long()
    first_of_seg = torch.repeat_interleave(first, count)
    count_of_seg = torch.repeat_interleave(count, count)
    w = torch.arange(n_seg, **i64) - first_of_seg
    sg2 = torch.zeros((2 * n_seg, 6), dtype=torch.int32, device=dev)
    lr2 = torch.zeros((2 * lr.shape[0], 4), dtype=torch.int32, device=dev)
    q2 = []
    for s in (0, 1):
        q = 2 * first_of_seg + s * count_of_seg + w
        q2.append(q)
        start2 = 2 * rowptr[seg_row] + s * lens[seg_row] + (seg_start - rowptr[seg_row])
        sg2[q, 0:2] = start2.contiguous().view(torch.int32).reshape(-1, 2)
        sg2[q, 2], sg2[q, 3], sg2[q, 4] = seg_len.int(), q.int(), (2 * seg_row + s).int()
        lr2[s::2, 0], lr2[s::2, 1], lr2[s::2, 2] = (2 * lr[:, 0].long() + s).int(), (2 * first + s * count).int(), count.int()
    lists = []
    order = order.long()
    for i in range(n_slices):
        ent = order[int(xoff[i]):int(xoff[i + 1])]
        is_seg = ent >= n
        per_half = []
        for s in (0, 1):
            e2 = torch.where(is_seg, 2 * n + q2[s][torch.clamp(ent - n, min=0)], 2 * ent + s)
            per_half.append(e2)
        if parity_lists:
            lists += per_half                                      # XCD 2 i + s: half s of slice-list i
        else:
            lists.append(torch.stack(per_half, 1).reshape(-1))      # both halves of a row in the same list, adjacent
    if not parity_lists:                                            # n_slices == 8 then
        assert len(lists) == 8
    m = CsrMatrix.from_device(rp2, col2, val2, (2 * n, 2 * n))
    m.n_long, m.n_segments = int(lr2.shape[0]), int(sg2.shape[0])
    m.long_rows = lr2.contiguous().view(torch.uint8).reshape(-1)
    m.segments = sg2.contiguous().view(torch.uint8).reshape(-1)
    m.row_order = torch.cat(lists).int()
    off = torch.zeros(9, **i64)
    off[1:] = torch.cumsum(torch.tensor([a.shape[0] for a in lists], **i64), 0)
    m.xcd_off = off
    m.long_threshold, m.segment_len = threshold, min(threshold, 256)
    m._partial = {}
    return m


x2 = x.view(2 * n, 32)
for threshold, n_slices, assign, parity in ((112, 4, 'affinity', True), (112, 8, 'affinity', False), (56, 4, 'affinity', True),
                                            (224, 4, 'affinity', True), (112, 4, 'spread', True)):
    m = doubled(threshold, n_slices, assign, parity)
    y = torch.empty_like(x)
    spmm(m, x2, out=y.view(2 * n, 32))
    torch.cuda.synchronize()
    err = float((y - y0).abs().max())
    for bpc in (None, 16, 32, 64):
        _lib.set_tuning('spmm_blocks_per_cu', bpc)
        t = min(bench.time_ms(lambda: spmm(m, x2, out=y.view(2 * n, 32)), 50, 5) for _ in range(3))
        print(json.dumps(dict(case='halves emulation', threshold=threshold, row_slices=n_slices, assign=assign, halves_on_xcd_parity=parity,
                              blocks_per_cu=bpc, us=round(t * 1e3, 1), vs_production=round(t / t0, 3), max_abs_diff=err)), flush=True)
    _lib.set_tuning('spmm_blocks_per_cu', None)
