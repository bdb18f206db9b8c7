"""Child process of tests/test_score_bpr_gpu.py::test_two_stage_call_replays_from_a_captured_hip_graph: a FRESH process in which
nothing has run on the replaying stream, and PyTorch's documented capture recipe — warm-up on a side stream, capture, replay on the
current stream.  (Round 4: this sequence faulted the GPU on the first replay.  Round 5, under rocgdb: order_place_kernel wrote out of
bounds because its bins had not been zeroed — the call's hipMemsetAsync, captured as a memset NODE, is not ordered against the kernel
nodes behind it on ROCm 7.2; the library now zeroes with a kernel of its own.)  Prints 'ok <n>' per replay checked."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from igcn_cf_amd import _lib                                                  # noqa: E402
from igcn_cf_amd.ops import score_topk                                         # noqa: E402


def capture_standard_recipe(call):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                                             # warm-up on a SIDE stream (kernel attributes, lazy module load)
        call()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):                                             # (captures on a stream of its own)
        call()
    return graph


def main():
    rng = np.random.default_rng(71)
    L = _lib.lib()
    dev = 'cuda'
    checked = 0
    for d, n_users, n_items, k in ((64, 2500, 20000, 20), (128, 700, 6000, 10)):
        ex = [np.sort(rng.choice(n_items, size=int(rng.integers(0, 40)), replace=False)) for _ in range(n_users)]
        rowptr = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum([len(x) for x in ex], out=rowptr[1:])
        rp = torch.from_numpy(rowptr).to(dev)
        cl = torch.from_numpy(np.concatenate(ex).astype(np.int32)).to(dev)
        U = torch.empty((n_users, d), dtype=torch.float32, device=dev)
        I = torch.empty((n_items, d), dtype=torch.float32, device=dev)
        ws_bytes = L.igcn_score_topk_fast_workspace_bytes(n_users, n_items, d, k, n_users, cl.numel())
        ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
        ws_ptr = (ws.data_ptr() + 255) // 256 * 256
        out_idx = torch.empty((n_users, k), dtype=torch.int64, device=dev)
        out_val = torch.empty((n_users, k), dtype=torch.float32, device=dev)
        flagged = torch.empty(n_users + 1, dtype=torch.int32, device=dev)
        bounds = torch.empty(n_users, dtype=torch.float32, device=dev)
        ws2 = torch.empty(max(L.igcn_score_topk_workspace_bytes(n_users, n_items, d, k), 8), dtype=torch.uint8, device=dev)
        idx2 = torch.empty((n_users, k), dtype=torch.int64, device=dev)
        val2 = torch.empty((n_users, k), dtype=torch.float32, device=dev)

        def fast():
            _lib.check(L.igcn_score_topk_fast_f32(U.data_ptr(), U.stride(0), None, n_users, I.data_ptr(), I.stride(0), n_items, d,
                                                  rp.data_ptr(), cl.data_ptr(), n_users, cl.numel(), None, k, out_idx.data_ptr(),
                                                  out_val.data_ptr(), flagged.data_ptr(), bounds.data_ptr(), ws_ptr, _lib.current_stream()),
                       'igcn_score_topk_fast_f32')

        def exact():                                                           # the fp32 sweep, captured as well
            _lib.check(L.igcn_score_topk_f32(U.data_ptr(), U.stride(0), None, n_users, I.data_ptr(), I.stride(0), n_items, d,
                                             rp.data_ptr(), cl.data_ptr(), None, k, idx2.data_ptr(), val2.data_ptr(), ws2.data_ptr(),
                                             _lib.current_stream()), 'igcn_score_topk_f32')

        def fill(seed, spread):
            g = torch.Generator(device=dev).manual_seed(seed)
            U.copy_(torch.randn(U.shape, device=dev, generator=g) * 0.1)
            I.copy_(torch.randn(I.shape, device=dev, generator=g) * 0.1 * torch.exp(spread * torch.randn((n_items, 1), device=dev, generator=g)))
        fill(0, 0.0)
        g_fast = capture_standard_recipe(fast)
        g_exact = capture_standard_recipe(exact)
        done = int(L.igcn_score_topk_fast_finished_max(n_users, 1))
        for seed, spread in ((1, 0.0), (2, 1.0), (3, 0.3)):                   # flat norms (warm-up pass), spread norms (early exits), in between
            fill(seed, spread)
            out_idx.fill_(-7)
            idx2.fill_(-7)
            g_fast.replay()                                                    # on the CURRENT stream: it has run nothing of the library eagerly
            g_exact.replay()
            torch.cuda.synchronize()
            ref = score_topk(U, I, k, excl_rowptr=rp, excl_col=cl, mode='exact')
            assert int(flagged[0]) <= done, (int(flagged[0]), done)           # (beyond that the caller would have to re-do users itself)
            assert torch.equal(out_idx, ref[0]) and torch.equal(out_val, ref[1]), (d, seed, spread)
            assert torch.equal(idx2, ref[0]) and torch.equal(val2, ref[1]), (d, seed, spread)
            checked += 1
            print('ok %d' % checked, flush=True)
    # the one call that refuses a capturing stream (rocPRIM's sort carries scratch): an error code, not a fault at replay time
    rowptr = torch.tensor([0, 2, 3], dtype=torch.int64, device=dev)
    col = torch.tensor([0, 1, 1], dtype=torch.int32, device=dev)
    t_rp = torch.empty(3, dtype=torch.int64, device=dev)
    t_col = torch.empty(3, dtype=torch.int32, device=dev)
    eid = torch.empty(3, dtype=torch.int32, device=dev)
    tws = torch.empty(L.igcn_csr_transpose_workspace_bytes(3) + 256, dtype=torch.uint8, device=dev)
    tws_ptr = (tws.data_ptr() + 255) // 256 * 256
    args = (rowptr.data_ptr(), col.data_ptr(), 2, 2, 3, t_rp.data_ptr(), t_col.data_ptr(), eid.data_ptr(), tws_ptr)
    assert L.igcn_csr_transpose(*args, _lib.current_stream()) == 0
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        rc = L.igcn_csr_transpose(*args, _lib.current_stream())
        torch.zeros(1, device=dev)                                             # (an empty capture is a warning)
    assert rc == -6, rc                                                        # IGCN_E_CAPTURE
    print('ok refused')


if __name__ == '__main__':
    main()
