"""torch-facing wrappers of the C-ABI kernels (include/igcn_hip.h).

PyTorch is plumbing here: it owns device memory, the current HIP stream and the
autograd graph.  All arithmetic of the hot path runs in libigcn_hip.so; there is
no eager/PyTorch fallback — a missing library or a non-GPU tensor raises.
"""
import ctypes as C

import torch

from . import _lib
from .graph import CsrMatrix


def _require_gpu_f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise _lib.IgcnError('%s must be a float32 tensor on the GPU (no CPU path)' % name)
    if t.dim() != 2 or t.stride(1) != 1:
        raise _lib.IgcnError('%s must be 2-D with unit inner stride' % name)


def spmm(csr: CsrMatrix, x, out=None, adds=(), out_scale=1.0, add_scale=1.0, row_scale=None, col_scale=None,
         keep_prob=1.0, seed=0, row_mask=None, masked_rows_zero=False, col_mask=None, order_bits=None, tune=None):
    """out = (out_scale * csr @ x + add_scale * sum(adds)) * row_scale  (see igcn_spmm_csr_f32 / igcn_spmm_csr_f32_args).
    tune: None, or {'blocks_per_cu' | 'multirow' | 'fold': int} — launch knobs of THIS call (never change the result).
    col_mask: bit mask over the columns (int32 words, pack_mask_bits): rows of x whose bit is clear are all zero and are not read.
    order_bits: row_mask once more in THIS matrix's dealing order (mark_rows(...).order_bits): the launch visits the wanted entries only.
    seed: an int, or a one-element int64 tensor on the GPU (read by the kernel: HIP-graph replays see its current value)."""
    _require_gpu_f32(x, 'x')
    n_rows, n_cols = csr.shape
    if x.shape[0] < n_cols:
        raise _lib.IgcnError('x has %d rows, matrix has %d columns' % (x.shape[0], n_cols))
    d = x.shape[1]
    if out is None:
        out = torch.empty((n_rows, d), dtype=torch.float32, device=x.device)
    _require_gpu_f32(out, 'out')
    if out.shape[0] < n_rows or out.shape[1] != d:
        raise _lib.IgcnError('bad output shape')
    if len(adds) > _lib.MAX_ADDS:
        raise _lib.IgcnError('at most %d epilogue addends' % _lib.MAX_ADDS)
    for a in adds:
        _require_gpu_f32(a, 'add')
        if a.shape[0] < n_rows or a.shape[1] != d or a.stride(0) != out.stride(0):
            raise _lib.IgcnError('epilogue addends must match the output layout')
    if row_scale is not None and (row_scale.dtype != torch.float32 or row_scale.numel() < n_rows or not row_scale.is_cuda):
        raise _lib.IgcnError('row_scale must be float32 [n_rows] on the GPU')
    if row_mask is not None and (row_mask.dtype != torch.uint8 or row_mask.numel() < n_rows or not row_mask.is_cuda):
        raise _lib.IgcnError('row_mask must be uint8 [n_rows] on the GPU')
    if col_mask is not None and (col_mask.dtype != torch.int32 or col_mask.numel() * 32 < n_cols or not col_mask.is_cuda):
        raise _lib.IgcnError('col_mask must be a bit mask (int32 words, ops.pack_mask_bits) over the columns, on the GPU')
    if col_scale is not None and (col_scale.dtype != torch.float32 or col_scale.numel() < n_cols or not col_scale.is_cuda):
        raise _lib.IgcnError('col_scale must be float32 [n_cols] on the GPU')
    if order_bits is not None:
        need = 2 * ((n_rows + csr.n_segments + 63) // 64) + 2
        if row_mask is None or order_bits.dtype != torch.int32 or not order_bits.is_cuda or order_bits.numel() < need:
            raise _lib.IgcnError('order_bits: int32 [%d] on the GPU (mark_rows(...).order_bits of this matrix), together with row_mask' % need)
    partial = csr.partial(d)
    seed_dev = None
    if isinstance(seed, torch.Tensor):
        if seed.dtype != torch.int64 or seed.numel() != 1 or not seed.is_cuda:
            raise _lib.IgcnError('a device seed must be a one-element int64 tensor on the GPU')
        seed_dev, seed = seed, 0
    # the struct form of the launch (igcn_spmm_csr_f32_args, ABI v10): what is not set stays zero = off / library default
    a = _lib.SpmmArgs()
    a.struct_size = C.sizeof(_lib.SpmmArgs)
    a.flags = (1 if masked_rows_zero else 0) | (2 if getattr(csr, 'closing_segments', False) else 0)
    a.rowptr, a.col, a.val = csr.rowptr.data_ptr(), csr.col.data_ptr(), _lib.ptr(csr.val)
    a.n_rows, a.n_cols, a.x, a.y, a.d = n_rows, n_cols, x.data_ptr(), out.data_ptr(), d
    a.ldx, a.ldy, a.nnz = x.stride(0), out.stride(0), csr.nnz
    a.n_adds = len(adds)
    for i, t in enumerate(adds):
        a.adds[i] = t.data_ptr()
    if out_scale == 0.0 or add_scale == 0.0 or not keep_prob > 0.0:
        raise _lib.IgcnError('out_scale / add_scale must not be zero and keep_prob must be in (0, 1]')    # (the struct reads 0 as 1)
    a.out_scale, a.add_scale, a.keep_prob = float(out_scale), float(add_scale), float(keep_prob)
    a.row_scale, a.col_scale = _lib.ptr(row_scale), _lib.ptr(col_scale)
    a.long_rows, a.n_long_rows, a.segments, a.n_segments = _lib.ptr(csr.long_rows), csr.n_long, _lib.ptr(csr.segments), csr.n_segments
    a.partial, a.long_threshold = _lib.ptr(partial), csr.long_threshold
    a.edge_id, a.seed, a.seed_dev = _lib.ptr(csr.edge_id), int(seed) & 0xFFFFFFFFFFFFFFFF, _lib.ptr(seed_dev)
    a.row_mask, a.col_mask, a.order_bits = _lib.ptr(row_mask), _lib.ptr(col_mask), _lib.ptr(order_bits)
    a.row_order, a.xcd_off = _lib.ptr(csr.row_order), _lib.ptr(csr.xcd_off)
    if tune:
        for k_, v_ in tune.items():                          # per-call launch knobs (result-neutral): blocks_per_cu / multirow / fold
            setattr(a, 'tune_' + k_, int(v_) + 1)
    _lib.check(_lib.lib().igcn_spmm_csr_f32_args(C.byref(a), _lib.current_stream()), 'igcn_spmm_csr_f32_args')
    return out


def pack_mask_bits(masks):
    """uint8 masks [n] or [m, n] (contiguous rows) -> int32 bit words [ceil(n / 32)] or [m, ceil(n / 32)] (igcn_pack_mask_bits)."""
    if masks.dtype != torch.uint8 or not masks.is_cuda or masks.stride(-1) != 1:
        raise _lib.IgcnError('masks must be uint8 on the GPU with contiguous rows')
    two = masks.dim() == 2
    m, n = (masks.shape if two else (1, masks.shape[0]))
    bits = torch.empty((m, (n + 31) // 32), dtype=torch.int32, device=masks.device)
    _lib.check(_lib.lib().igcn_pack_mask_bits(masks.data_ptr(), n, masks.stride(0) if two else n, m, bits.data_ptr(),
                                              _lib.current_stream()), 'igcn_pack_mask_bits')
    return bits if two else bits[0]


class RowMarks(tuple):
    """(mask1, mask2, bits1, bits2) of mark_rows, plus `order_bits`: mask1 in the dealing order of the matrix `order_of` (what
    spmm's order_bits takes for THAT matrix only), or None."""
    order_bits = None
    order_of = None

    def order_bits_for(self, csr):
        return self.order_bits if self.order_of is csr else None


def mark_rows(csr: CsrMatrix, ids, with_neighbours=True):
    """(mask1, mask2, bits1, bits2): uint8 [n_rows] masks — mask1 = the listed rows, mask2 = those rows and their
    neighbourhood in `csr` (igcn_mark_rows) — and the same two as bit masks (what spmm's col_mask takes).  The tuple also
    carries mask1 in the dealing order of `csr` (RowMarks.order_bits; same launch as the bit masks)."""
    _require_i64(ids, 'ids')
    n_rows = csr.shape[0]
    masks = torch.zeros((2, n_rows), dtype=torch.uint8, device=ids.device)
    _lib.check(_lib.lib().igcn_mark_rows(ids.data_ptr(), ids.numel(), csr.rowptr.data_ptr(), csr.col.data_ptr(),
                                         masks[0].data_ptr(), masks[1].data_ptr() if with_neighbours else None, n_rows,
                                         _lib.current_stream()), 'igcn_mark_rows')
    bits = torch.empty((2, (n_rows + 31) // 32), dtype=torch.int32, device=ids.device)
    order_bits = torch.empty(2 * ((n_rows + csr.n_segments + 63) // 64) + 2, dtype=torch.int32, device=ids.device)
    _lib.check(_lib.lib().igcn_pack_mask_bits_ordered(masks.data_ptr(), n_rows, masks.stride(0), 2, bits.data_ptr(),
                                                      _lib.ptr(csr.row_order), csr.row_order.numel() if csr.row_order is not None else 0,
                                                      _lib.ptr(csr.segments), csr.n_segments,
                                                      order_bits.data_ptr(), _lib.current_stream()), 'igcn_pack_mask_bits_ordered')
    marks = RowMarks((masks[0], masks[1], bits[0], bits[1]))
    marks.order_bits, marks.order_of = order_bits, csr
    return marks


def mean_plan(n_layers, form='factored'):
    """The K launches that give (X_0 + A X_0 + ... + A^K X_0) with as few epilogue addends as the polynomial allows.

    Returns `adds`, one entry per launch: table l + 1 = A @ table l (+ table adds[l] if not None), table 0 = X_0; the last
    table is the sum.  form='stack': the reference's own association instead — every X_l kept, all of them added in the last
    launch ([None, ..., (0, 1, ..., K - 1)]; an entry may be a tuple of tables) — kept for A/Bs.
    The layer loop of model.py:101-105 keeps every X_l and averages the stack — K extra row reads however they are spread
    (Horner's rule: X_0 once per launch).  1 + x + ... + x^K factors instead:
    K odd: (1 + x) p_m(x^2), m = (K - 1) / 2; K even: 1 + x p_(K-1)(x) — K = 3, the depth every reference config uses, is
    (I + A)(I + A^2) X_0: U = X_0 + A (A X_0), then U + A U: two addend reads instead of three (-3.2 % of a pass at the
    headline size, -4.6 % Gowalla-like; float64 error of the result not larger; scripts/dev_r05_factorized_mean.py).
    The sums are associated differently from the reference's stack().mean(): fp32 rounding, ~1e-7 relative."""
    if form == 'stack':
        return [None] * (n_layers - 1) + [tuple(range(n_layers))] if n_layers else []
    if form != 'factored':
        raise ValueError("form must be 'factored' or 'stack'")
    adds = []

    def poly(k, stride):                                  # table index of p_k(A^stride) X_0
        if k == 0:
            return 0
        if k % 2:
            w = poly((k - 1) // 2, 2 * stride)           # p_k(y) = (1 + y) p_m(y^2)
            last_add = w
        else:
            w = poly(k - 1, stride)                      # p_k(y) = 1 + y p_(k-1)(y)
            last_add = 0
        assert w == len(adds)                            # every launch reads the table the one before it wrote
        adds.extend([None] * (stride - 1) + [last_add])
        return len(adds)

    poly(n_layers, 1)
    return adds


def plan_addends(entry, tables):
    """The addend tables of one mean_plan entry (None, an index or a tuple of indices)."""
    if entry is None:
        return ()
    return [tables[i] for i in (entry if isinstance(entry, tuple) else (entry,))]


def propagate_mean(csr: CsrMatrix, x0, n_layers, row_scale_last=None, masks=None, zero_masked=True, plan=None, tune=None, out=None):
    """mean(X_0..X_K), X_{l+1} = csr @ X_l — the layer loop + stack/mean of
    model.py:101-105, as the K launches of mean_plan(); the 1 / (K + 1) is the last launch's epilogue (no stack).

    masks = (rows, rows_and_neighbours) from mark_rows(): only the listed rows of the result are
    needed (a training step reads the propagated rows of its batch only, model.py:114-115).  The
    last launch is then computed for those rows alone (others are zero) and the one before it for
    their neighbourhood alone (other rows of that intermediate are never read) — same values on
    the needed rows, ~45 % fewer edges at Amazon scale with a 2048-triplet batch.
    tune: per-call launch knobs handed to every spmm (result-neutral); out: the [N, d] buffer the result goes to (a captured pass
    writes to the same tensor at every replay)."""
    if n_layers == 0:
        return x0.clone() if out is None else out.copy_(x0)
    tables = [x0]
    s = 1.0 / (n_layers + 1)
    for l, add in enumerate(plan if plan is not None else mean_plan(n_layers)):
        adds = plan_addends(add, tables)
        if l == n_layers - 1:
            # (rows outside the mask left untouched: the launch walks the wanted entries only — RowMarks.order_bits)
            skip = masks.order_bits_for(csr) if masks and not zero_masked and isinstance(masks, RowMarks) else None
            y = spmm(csr, tables[-1], out=out, adds=adds, out_scale=s, add_scale=s, row_scale=row_scale_last,
                     row_mask=masks[0] if masks else None, masked_rows_zero=zero_masked, order_bits=skip, tune=tune)
        elif l == n_layers - 2 and masks:
            y = spmm(csr, tables[-1], adds=adds, row_mask=masks[1], masked_rows_zero=False, tune=tune)
        else:
            y = spmm(csr, tables[-1], adds=adds, tune=tune)
        tables.append(y)
    return tables[-1]


def propagate_mean_backward(csr_t: CsrMatrix, grad, n_layers, row_scale=None, masks=None):
    """d/dX_0 of propagate_mean:  s * sum_l (M^T)^l g — the same polynomial in M^T, the same mean_plan() launches
    applied to g.  row_scale (optional) multiplies the final rows (used by the INMO path).
    With masks (g is zero outside masks[0]) the first hop is non-zero only on masks[1]: it is computed for those
    rows alone and gathers only the rows of g in masks[0]; the second hop gathers only the rows in masks[1]
    (about 70 % of the rows at B = 2048 on the Amazon-like graph: 124 -> 116 us)."""
    s = 1.0 / (n_layers + 1)
    if n_layers == 0:
        return grad.clone()
    tables = [grad.contiguous()]
    for l, add in enumerate(mean_plan(n_layers)):
        last = l == n_layers - 1
        adds = [tables[add]] if add is not None else ()
        scale = s if last else 1.0
        if l == 0:
            # rows outside masks[1] are zero: written as zeros only when this hop is the result, else never read
            y = spmm(csr_t, tables[-1], adds=adds, out_scale=scale, add_scale=scale, row_scale=row_scale if last else None,
                     row_mask=masks[1] if masks else None, masked_rows_zero=last, col_mask=masks[2] if masks else None)
        else:
            y = spmm(csr_t, tables[-1], adds=adds, out_scale=scale, add_scale=scale, row_scale=row_scale if last else None,
                     col_mask=masks[3] if masks and l == 1 else None)
        tables.append(y)
    return tables[-1]


class PropagateFn(torch.autograd.Function):
    """LightGCN.get_rep (model.py:96-106) as one autograd node.  A_hat is
    symmetric (model.py:85-94), so the backward uses the same CSR.
    needed_rows (int64 ids or None): the only rows of the output the caller will read AND the
    only rows that will receive a gradient (the batch rows of a BPR step)."""

    @staticmethod
    def forward(ctx, emb, csr, csr_t, n_layers, needed_rows=None):
        masks = mark_rows(csr, needed_rows) if needed_rows is not None and n_layers > 0 else None
        ctx.csr_t, ctx.n_layers, ctx.masks = csr_t, n_layers, masks
        return propagate_mean(csr, emb.detach(), n_layers, masks=masks)

    @staticmethod
    def backward(ctx, grad):
        return propagate_mean_backward(ctx.csr_t, grad, ctx.n_layers, masks=ctx.masks), None, None, None, None


class FeatureLayerFn(torch.autograd.Function):
    """IGCN.inductive_rep_layer on the (dropped-out) feature matrix:
    X0 = dropout(F) @ T with F's values = row_scale[row]  (model.py:374-377,
    :423-432, :263-275 via :435).  Backward: dT = dropout(F)^T @ (row_scale * dX0)
    on the transposed CSR (row_scale enters as col_scale), same edges dropped (edge ids)."""

    @staticmethod
    def forward(ctx, templ, feat, feat_t, row_scale, keep_prob, seed):
        ctx.feat_t, ctx.row_scale, ctx.keep_prob, ctx.seed = feat_t, row_scale, keep_prob, seed
        return spmm(feat, templ.detach(), row_scale=row_scale, keep_prob=keep_prob, seed=seed)

    @staticmethod
    def backward(ctx, grad):
        return spmm(ctx.feat_t, grad.contiguous(), col_scale=ctx.row_scale, keep_prob=ctx.keep_prob,
                    seed=ctx.seed), None, None, None, None, None


class BatchGradTable:
    """The dense [N, d] table a training step's row-sparse batch gradients are scattered into before the propagation
    backward gathers from it.  It is all zeros between steps: a step adds into <= 3 B rows and puts exactly those rows
    back to zero afterwards (igcn_rows_zero_f32), instead of allocating and zero-filling N x d floats per step."""

    def __init__(self):
        self._t = None

    def get(self, like):
        if self._t is None or self._t.shape != like.shape or self._t.device != like.device:
            self._t = torch.zeros_like(like)
        return self._t


class GraphBprFn(torch.autograd.Function):
    """The differentiable part of one BPR step of a graph model as ONE autograd node (model.py:108-116 / :293-299 +
    trainer.py:238-243): K-layer propagation pruned to what the batch reads, row gathers, dots, softplus, L2.

    forward(x0, csr, csr_t, n_layers, nodes [3 B] = users | n_users + positives | n_users + negatives, l2_on,
            table: BatchGradTable, prune, l2_reg)
      l2_reg None  -> tensor [2] = (mean softplus(neg - pos), mean l2_norm_sq)
      l2_reg float -> the scalar training loss bpr + l2_reg * l2 (trainer.py:242), computed inside the reduce kernel
    l2_on = 'raw': the L2 term reads the rows of x0 (LightGCN, model.py:110-113); 'rep': the propagated rows (IGCN).
    backward: batch gradients -> the persistent zero table (float atomics, 256-B row segments), K SpMMs (Horner), then
    ONE launch adds the L2 rows of 'raw' straight into the result and puts the table's rows back to zero."""

    @staticmethod
    def forward(ctx, x0, csr, csr_t, n_layers, nodes, l2_on, table, prune, l2_reg):
        _require_gpu_f32(x0, 'x0')
        _require_i64(nodes, 'nodes')
        if nodes.numel() % 3 or not x0.is_contiguous():
            raise _lib.IgcnError('nodes must hold 3 * B ids and x0 must be contiguous')
        B, d = nodes.numel() // 3, x0.shape[1]
        xd = x0.detach()
        masks = mark_rows(csr, nodes) if prune and n_layers > 0 else None
        # only the batch rows of the result are read (by the kernels below): the rest is left unwritten
        rep = propagate_mean(csr, xd, n_layers, masks=masks, zero_masked=False)
        l2t = xd if l2_on == 'raw' else rep
        users, pos, neg = nodes[:B], nodes[B:2 * B], nodes[2 * B:]
        out = torch.empty(3, dtype=torch.float32, device=x0.device)
        work = torch.empty(3 * B, dtype=torch.float32, device=x0.device)
        _lib.check(_lib.lib().igcn_bpr_loss_f32(
            rep.data_ptr(), rep.data_ptr(), rep.data_ptr(), rep.stride(0), l2t.data_ptr(), l2t.data_ptr(), l2t.data_ptr(),
            l2t.stride(0), users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B, d, None,
            0.0 if l2_reg is None else float(l2_reg), out.data_ptr(), work.data_ptr(), _lib.current_stream()), 'igcn_bpr_loss_f32')
        ctx.state = (xd, rep, csr_t, n_layers, nodes, l2_on, table, masks, work, l2_reg)
        return out[:2] if l2_reg is None else out[2]

    @staticmethod
    def backward(ctx, g_out):
        xd, rep, csr_t, n_layers, nodes, l2_on, table, masks, work, l2_reg = ctx.state
        B, d = nodes.numel() // 3, rep.shape[1]
        users, pos, neg = nodes[:B], nodes[B:2 * B], nodes[2 * B:]
        g = g_out.contiguous().float()
        gt = table.get(rep)
        on_rep = l2_on == 'rep'
        rp, gp = rep.data_ptr(), gt.data_ptr()
        l2p, g2p = (rp, gp) if on_rep else (None, None)
        common = (rp, rp, rp, rep.stride(0), l2p, l2p, l2p, rep.stride(0), users.data_ptr(), pos.data_ptr(), neg.data_ptr(),
                  B, d, None, work.data_ptr(), g.data_ptr())
        grads = (gp, gp, gp, g2p, g2p, g2p, None, _lib.current_stream())
        if l2_reg is None:
            _lib.check(_lib.lib().igcn_bpr_bwd_f32(*common, *grads), 'igcn_bpr_bwd_f32')
            l2_dev, l2_host = g[1:], 2.0 / B
        else:
            _lib.check(_lib.lib().igcn_bpr_loss_bwd_f32(*common, float(l2_reg), *grads), 'igcn_bpr_loss_bwd_f32')
            l2_dev, l2_host = g, 2.0 * float(l2_reg) / B
        grad = propagate_mean_backward(csr_t, gt, n_layers, masks=masks)
        raw = not on_rep                                 # d/dx0 of mean_b |x0[row]|^2, row by row, into the dense result
        _lib.check(_lib.lib().igcn_rows_finish_f32(grad.data_ptr() if raw else None, grad.stride(0), xd.data_ptr() if raw else None,
                                                   xd.stride(0), gp, gt.stride(0), nodes.data_ptr(), 3 * B, d,
                                                   l2_dev.data_ptr(), l2_host, _lib.current_stream()), 'igcn_rows_finish_f32')
        return grad, None, None, None, None, None, None, None, None


def graph_bpr_terms(x0, csr, csr_t, n_layers, nodes, l2_on, table, prune=True, l2_reg=None):
    return GraphBprFn.apply(x0, csr, csr_t, n_layers, nodes, l2_on, table, prune, l2_reg)


class InmoStepFn(torch.autograd.Function):
    """The differentiable part of one INMO training step (IGCN / IMF under IGCNTrainer, trainer.py:294-320) as ONE autograd
    node: template layer X0 = dropout(F) T (model.py:423-435), K-layer propagation pruned to the batch, the BPR loss with
    its L2 term on the propagated rows (model.py:293-299, trainer.py:300-303), AND the self-enhanced auxiliary loss on the
    raw template rows weighted by w (trainer.py:304-312):

        loss = bpr + l2_reg * mean l2_norm_sq + aux_reg * aux_bpr.

    Backward: batch-row gradients -> the persistent zero table, K SpMMs (Horner), the transposed template layer — which
    yields the DENSE gradient of the template table — and then the auxiliary loss adds its row gradients straight into
    that result (float atomics) and produces d / d w.  As separate autograd nodes the auxiliary loss cost a zero-filled
    30 MB table, a dense add of two table gradients and four elementwise launches per step.

    forward(templ [Tn, d], w [d], feat, feat_t, feat_scale, keep_prob, seed, csr, n_layers, nodes [3 B], aux [Ba, 3]
            (template-space users | positives | negatives), item_offset (= len(user_map): template rows of the items start
            there), table: BatchGradTable, prune, l2_reg, aux_reg) -> scalar loss"""

    @staticmethod
    def forward(ctx, templ, w, feat, feat_t, feat_scale, keep_prob, seed, csr, n_layers, nodes, aux, item_offset, table, prune,
                l2_reg, aux_reg):
        _require_gpu_f32(templ, 'templ')
        _require_i64(nodes, 'nodes')
        if nodes.numel() % 3 or not templ.is_contiguous() or aux.dim() != 2 or aux.shape[1] != 3 or aux.dtype != torch.int64:
            raise _lib.IgcnError('nodes must hold 3 * B ids, aux must be int64 [Ba, 3], templ contiguous')
        L = _lib.lib()
        td, wd = templ.detach(), w.detach().contiguous()
        B, d = nodes.numel() // 3, td.shape[1]
        x0 = spmm(feat, td, row_scale=feat_scale, keep_prob=keep_prob, seed=seed)
        masks = mark_rows(csr, nodes) if prune and n_layers > 0 else None
        rep = propagate_mean(csr, x0, n_layers, masks=masks, zero_masked=False) if n_layers > 0 else x0
        users, pos, neg = nodes[:B], nodes[B:2 * B], nodes[2 * B:]
        out = torch.empty(3, dtype=torch.float32, device=td.device)
        work = torch.empty(3 * B, dtype=torch.float32, device=td.device)
        rp, ld = rep.data_ptr(), rep.stride(0)
        _lib.check(L.igcn_bpr_loss_f32(rp, rp, rp, ld, rp, rp, rp, ld, users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B, d, None,
                                       float(l2_reg), out.data_ptr(), work.data_ptr(), _lib.current_stream()), 'igcn_bpr_loss_f32')
        a3 = aux.t().contiguous()                                  # [3, Ba]: users | positives | negatives
        Ba = a3.shape[1]
        out_a = torch.empty(3, dtype=torch.float32, device=td.device)
        work_a = torch.empty(3 * Ba, dtype=torch.float32, device=td.device)
        tp, tpi = td.data_ptr(), _row_view(td, item_offset)
        _lib.check(L.igcn_bpr_loss_f32(tp, tpi, tpi, td.stride(0), None, None, None, td.stride(0), a3[0].data_ptr(), a3[1].data_ptr(),
                                       a3[2].data_ptr(), Ba, d, wd.data_ptr(), 0.0, out_a.data_ptr(), work_a.data_ptr(),
                                       _lib.current_stream()), 'igcn_bpr_loss_f32')
        ctx.state = (td, wd, x0, rep, feat_t, feat_scale, keep_prob, seed, csr, n_layers, nodes, a3, item_offset, table, masks, work,
                     work_a, float(l2_reg), float(aux_reg))
        return torch.add(out[2], out_a[0], alpha=float(aux_reg))

    @staticmethod
    def backward(ctx, g_out):
        (td, wd, x0, rep, feat_t, feat_scale, keep_prob, seed, csr, n_layers, nodes, a3, item_offset, table, masks, work, work_a,
         l2_reg, aux_reg) = ctx.state
        L = _lib.lib()
        B, d = nodes.numel() // 3, rep.shape[1]
        users, pos, neg = nodes[:B], nodes[B:2 * B], nodes[2 * B:]
        g = g_out.contiguous().float().reshape(1)
        gt = table.get(rep)
        rp, gp, ld = rep.data_ptr(), gt.data_ptr(), rep.stride(0)
        _lib.check(L.igcn_bpr_loss_bwd_f32(rp, rp, rp, ld, rp, rp, rp, ld, users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B, d, None,
                                           work.data_ptr(), g.data_ptr(), l2_reg, gp, gp, gp, gp, gp, gp, None,
                                           _lib.current_stream()), 'igcn_bpr_loss_bwd_f32')
        grad_x0 = propagate_mean_backward(csr, gt, n_layers, masks=masks)          # A_hat is symmetric; n_layers == 0: a copy
        _lib.check(L.igcn_rows_finish_f32(None, grad_x0.stride(0), None, grad_x0.stride(0), gp, gt.stride(0), nodes.data_ptr(), 3 * B, d,
                                          None, 0.0, _lib.current_stream()), 'igcn_rows_finish_f32')
        g_templ = spmm(feat_t, grad_x0, col_scale=feat_scale, keep_prob=keep_prob, seed=seed)     # dense [Tn, d]
        g_w = torch.zeros_like(wd)
        Ba = a3.shape[1]
        tp, tpi = td.data_ptr(), _row_view(td, item_offset)
        gtp, gtpi = g_templ.data_ptr(), _row_view(g_templ, item_offset)
        _lib.check(L.igcn_bpr_loss_bwd_scaled_f32(tp, tpi, tpi, td.stride(0), None, None, None, td.stride(0), a3[0].data_ptr(),
                                                  a3[1].data_ptr(), a3[2].data_ptr(), Ba, d, wd.data_ptr(), work_a.data_ptr(),
                                                  g.data_ptr(), aux_reg, 0.0, gtp, gtpi, gtpi, None, None, None, g_w.data_ptr(),
                                                  _lib.current_stream()), 'igcn_bpr_loss_bwd_scaled_f32')
        return (g_templ, g_w) + (None,) * 14


def inmo_step_loss(templ, w, feat, feat_t, feat_scale, keep_prob, seed, csr, n_layers, nodes, aux, item_offset, table, prune,
                   l2_reg, aux_reg):
    return InmoStepFn.apply(templ, w, feat, feat_t, feat_scale, keep_prob, seed, csr, n_layers, nodes, aux, item_offset, table, prune,
                            l2_reg, aux_reg)


def bpr_sample_nodes(train_rowptr, train_col, nonempty_users, n_items, batch, seed, item_offset, out=None):
    """int64 [3 * batch] node ids of the draws of bpr_sample(seed): users | item_offset + positives | item_offset +
    negatives (igcn_bpr_sample_nodes).  out: write there (e.g. the input buffer of a captured step) instead of a new tensor."""
    _require_i64(train_rowptr, 'train_rowptr')
    if out is None:
        out = torch.empty(3 * batch, dtype=torch.int64, device=train_rowptr.device)
    elif out.dtype != torch.int64 or out.numel() != 3 * batch or not out.is_contiguous() or out.device != train_rowptr.device:
        raise _lib.IgcnError('out must be a contiguous int64 [3 * batch] tensor on the sampler\'s device')
    _lib.check(_lib.lib().igcn_bpr_sample_nodes(train_rowptr.data_ptr(), train_col.data_ptr(), nonempty_users.data_ptr(),
                                                nonempty_users.numel(), n_items, batch, int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                int(item_offset), out.data_ptr(), _lib.current_stream()), 'igcn_bpr_sample_nodes')
    return out


def _require_i64(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
        raise _lib.IgcnError('%s must be a contiguous int64 tensor on the GPU' % name)


def _row_view(t, offset):
    """Base pointer of table `t` shifted by `offset` rows (e.g. '+ n_users')."""
    return t.data_ptr() + offset * t.stride(0) * 4


class BprLossFn(torch.autograd.Function):
    """Fused BPR triplet scoring (igcn_bpr_fwd_f32 / igcn_bpr_bwd_f32).

    forward(u_tab, p_tab, l2u_tab | None, l2p_tab | None, w | None, users, pos, neg,
            item_offset, l2_item_offset) -> tensor [2] = (mean softplus(neg - pos), mean l2_norm_sq)

    u_tab / p_tab: tables the user / item rows of the scores are gathered from;
    the same tensor may be passed for both (stacked [users; items] layout of
    LightGCN / IGCN, model.py:110-115, :295-298) with item_offset = n_users
    (len(user_map) for the auxiliary loss, trainer.py:306-308).  l2 tables: where
    the squared-norm term gathers from (raw embeddings for LightGCN / MF,
    propagated rows for IGCN).  A tensor passed several times gets ONE dense
    gradient buffer.
    """

    @staticmethod
    def forward(ctx, u_tab, p_tab, l2u_tab, l2p_tab, w, users, pos, neg, item_offset, l2_item_offset, reduce_fn=None):
        has_l2 = l2u_tab is not None
        tabs = [u_tab, p_tab] + ([l2u_tab, l2p_tab] if has_l2 else [])
        for t in tabs:
            _require_gpu_f32(t, 'table')
            if not t.is_contiguous():
                raise _lib.IgcnError('BPR tables must be contiguous')
        for t, n in ((users, 'users'), (pos, 'pos_items'), (neg, 'neg_items')):
            _require_i64(t, n)
        d = u_tab.shape[1]
        if any(t.shape[1] != d for t in tabs) or (w is not None and w.numel() != d):
            raise _lib.IgcnError('BPR tables / w differ in width')
        B = users.numel()
        out = torch.empty(2, dtype=torch.float32, device=u_tab.device)
        work = torch.empty(3 * B, dtype=torch.float32, device=u_tab.device)
        p_ptr = _row_view(p_tab, item_offset)
        l2u_ptr = l2u_tab.data_ptr() if has_l2 else None
        l2p_ptr = _row_view(l2p_tab, l2_item_offset) if has_l2 else None
        if reduce_fn is None:
            _lib.check(_lib.lib().igcn_bpr_fwd_f32(
                u_tab.data_ptr(), p_ptr, p_ptr, d, l2u_ptr, l2p_ptr, l2p_ptr, d,
                users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B, d,
                _lib.ptr(w), out.data_ptr(), work.data_ptr(), _lib.current_stream()), 'igcn_bpr_fwd_f32')
        else:
            # the tables hold one embedding-column slice: partial dots, summed over the slices by
            # reduce_fn (an all-reduce), then softplus / sigmoid on the complete dots
            dots = torch.empty(3 * B, dtype=torch.float32, device=u_tab.device)
            _lib.check(_lib.lib().igcn_bpr_dots_f32(
                u_tab.data_ptr(), p_ptr, p_ptr, d, l2u_ptr, l2p_ptr, l2p_ptr, d,
                users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B, d,
                _lib.ptr(w), dots.data_ptr(), _lib.current_stream()), 'igcn_bpr_dots_f32')
            reduce_fn(dots)
            _lib.check(_lib.lib().igcn_bpr_finish_f32(dots.data_ptr(), B, out.data_ptr(), work.data_ptr(),
                                                      _lib.current_stream()), 'igcn_bpr_finish_f32')
        ctx.tabs = (u_tab.detach(), p_tab.detach(), l2u_tab.detach() if has_l2 else None,
                    l2p_tab.detach() if has_l2 else None, w.detach() if w is not None else None)
        ctx.idx = (users, pos, neg, work)
        ctx.item_offset, ctx.l2_item_offset = item_offset, l2_item_offset
        return out

    @staticmethod
    def backward(ctx, g_out):
        u_tab, p_tab, l2u_tab, l2p_tab, w = ctx.tabs
        users, pos, neg, work = ctx.idx
        has_l2, has_w = l2u_tab is not None, w is not None
        bufs = {}

        def buf_for(t):
            key = (t.data_ptr(), tuple(t.shape))
            if key not in bufs:
                bufs[key] = torch.zeros_like(t)
            return bufs[key]
        gu, gp = buf_for(u_tab), buf_for(p_tab)
        g2u = buf_for(l2u_tab) if has_l2 else None
        g2p = buf_for(l2p_tab) if has_l2 else None
        gw = torch.zeros_like(w) if has_w else None
        g = g_out.contiguous().float()
        B, d = users.numel(), u_tab.shape[1]
        p_ptr, gp_ptr = _row_view(p_tab, ctx.item_offset), _row_view(gp, ctx.item_offset)
        l2p_ptr = _row_view(l2p_tab, ctx.l2_item_offset) if has_l2 else None
        g2p_ptr = _row_view(g2p, ctx.l2_item_offset) if has_l2 else None
        _lib.check(_lib.lib().igcn_bpr_bwd_f32(
            u_tab.data_ptr(), p_ptr, p_ptr, d,
            l2u_tab.data_ptr() if has_l2 else None, l2p_ptr, l2p_ptr, d,
            users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B, d, _lib.ptr(w),
            work.data_ptr(), g.data_ptr(), gu.data_ptr(), gp_ptr, gp_ptr,
            g2u.data_ptr() if has_l2 else None, g2p_ptr, g2p_ptr,
            _lib.ptr(gw), _lib.current_stream()), 'igcn_bpr_bwd_f32')
        seen = set()

        def once(bf):
            if bf is None or id(bf) in seen:
                return None
            seen.add(id(bf))
            return bf
        return once(gu), once(gp), once(g2u), once(g2p), gw, None, None, None, None, None, None


class BprScalarLossFn(torch.autograd.Function):
    """trainer.py:238-243 for a model without propagation (MF, model.py:62-67) as ONE autograd node: the scalar
    bpr + l2_reg * mean l2_norm_sq straight from the reduce kernel (igcn_bpr_loss_f32) and its gradient back in as one float
    (igcn_bpr_loss_bwd_f32) — composed from the two-term vector with torch ops, the scaling, the add and their backward
    cost eight launch-bound elementwise launches per step.

    forward(u_tab [U, d], i_tab [I, d], users, pos, neg [B] int64, l2_reg) -> scalar; the L2 term reads the same rows."""

    @staticmethod
    def forward(ctx, u_tab, i_tab, users, pos, neg, l2_reg):
        for t in (u_tab, i_tab):
            _require_gpu_f32(t, 'table')
            if not t.is_contiguous():
                raise _lib.IgcnError('BPR tables must be contiguous')
        for t, n in ((users, 'users'), (pos, 'pos_items'), (neg, 'neg_items')):
            _require_i64(t, n)
        d, B = u_tab.shape[1], users.numel()
        if i_tab.shape[1] != d:
            raise _lib.IgcnError('BPR tables differ in width')
        ud, idt = u_tab.detach(), i_tab.detach()
        out = torch.empty(3, dtype=torch.float32, device=ud.device)
        work = torch.empty(3 * B, dtype=torch.float32, device=ud.device)
        up, ip = ud.data_ptr(), idt.data_ptr()
        _lib.check(_lib.lib().igcn_bpr_loss_f32(up, ip, ip, d, up, ip, ip, d, users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B, d, None,
                                                float(l2_reg), out.data_ptr(), work.data_ptr(), _lib.current_stream()), 'igcn_bpr_loss_f32')
        ctx.state = (ud, idt, users, pos, neg, work, float(l2_reg))
        return out[2]

    @staticmethod
    def backward(ctx, g_out):
        ud, idt, users, pos, neg, work, l2_reg = ctx.state
        d, B = ud.shape[1], users.numel()
        g = g_out.contiguous().float().reshape(1)
        gu, gi = torch.zeros_like(ud), torch.zeros_like(idt)
        up, ip, gup, gip = ud.data_ptr(), idt.data_ptr(), gu.data_ptr(), gi.data_ptr()
        _lib.check(_lib.lib().igcn_bpr_loss_bwd_f32(up, ip, ip, d, up, ip, ip, d, users.data_ptr(), pos.data_ptr(), neg.data_ptr(), B, d,
                                                    None, work.data_ptr(), g.data_ptr(), l2_reg, gup, gip, gip, gup, gip, gip, None,
                                                    _lib.current_stream()), 'igcn_bpr_loss_bwd_f32')
        return gu, gi, None, None, None, None


def bpr_scalar_loss(u_tab, i_tab, users, pos, neg, l2_reg):
    return BprScalarLossFn.apply(u_tab, i_tab, users, pos, neg, l2_reg)


def bpr_loss_terms(u_tab, p_tab, l2u_tab, l2p_tab, w, users, pos, neg, item_offset=0, l2_item_offset=0, reduce_fn=None):
    return BprLossFn.apply(u_tab, p_tab, l2u_tab, l2p_tab, w, users, pos, neg, item_offset, l2_item_offset, reduce_fn)


# users x items from which the two-stage path pays for its order / packing passes and its host check (random-init tables,
# MI355X, profiles/r03ai_*, with the pieces of a cut sweep sharing their thresholds: 2 048 users x 96 k items 0.49 against 0.57 ms,
# 1 024 users 0.43 against 0.47; 4 096 x 41 k 0.54 against 0.52, 2 048 x 41 k 0.46 against 0.42; 16 384 x 10 k 0.56 against 0.56)
FAST_TOPK_MIN_WORK = 1 << 27


def score_topk(user_rows, item_rows, k, user_ids=None, excl_rowptr=None, excl_col=None, banned=None, batch=None, mode='auto',
               lower_bound=None):
    """Top-k item ids (best first) and scores for each user row; masked items are
    never returned unless fewer than k unmasked items exist.

    mode 'exact': the fp32 sweep (igcn_score_topk_f32).  'fast' (d = 64 or 128, k <= 60): the fp16 candidate sweep + exact fp32
    re-scoring (igcn_score_topk_fast_f32) — the same ids, exact fp32 scores; users whose candidate set cannot be proven
    complete (near-ties at the k-th place; counted on the device, read back here) go through the fp32 sweep.  'auto':
    'fast' where it applies and the problem is large enough to pay for it.
    lower_bound (mode 'exact' only): float32 [B] on the GPU, per user a lower bound of its k-th best score
    (igcn_score_topk_bounded_f32): only items that reach it are looked at.  A bound that turns out too high costs time,
    not correctness: a list it left short is redone by the plain fp32 sweep."""
    if mode not in ('auto', 'exact', 'fast'):
        raise ValueError("mode must be 'auto', 'exact' or 'fast'")
    _require_gpu_f32(user_rows, 'user_rows')
    _require_gpu_f32(item_rows, 'item_rows')
    if user_ids is not None:
        _require_i64(user_ids, 'user_ids')
        B = user_ids.numel()
    else:
        B = user_rows.shape[0] if batch is None else batch
    n_items, d = item_rows.shape
    if user_rows.shape[1] != d:
        raise _lib.IgcnError('user and item rows differ in width')
    if d % 4 or user_rows.stride(0) % 4 or item_rows.stride(0) % 4 or user_rows.data_ptr() % 16 or item_rows.data_ptr() % 16:
        # the kernel reads rows as 16-byte pieces: zero-pad odd widths / realign views (zeros add nothing to a dot)
        pad = (-d) % 4
        user_rows = torch.nn.functional.pad(user_rows, (0, pad)).contiguous()
        item_rows = torch.nn.functional.pad(item_rows, (0, pad)).contiguous()
        d += pad
    if excl_rowptr is not None and (excl_col is None or excl_col.numel() == 0):
        excl_rowptr = excl_col = None                      # nothing is excluded
    if excl_rowptr is not None:
        _require_i64(excl_rowptr, 'excl_rowptr')
        if excl_col.dtype != torch.int32 or not excl_col.is_cuda:
            raise _lib.IgcnError('excl_col must be int32 on the GPU')
    if banned is not None and (banned.dtype != torch.uint8 or banned.numel() != n_items or not banned.is_cuda):
        raise _lib.IgcnError('banned must be uint8 [n_items] on the GPU')
    L = _lib.lib()
    # the re-scoring wave has 64 lanes, one per candidate: k + 6 candidates while k <= 58, 64 - k of them at k = 59, 60 (the library's
    # own rule, topk_fast_extra), nothing above — igcn_score_topk_fast_workspace_bytes refuses k > 60
    fast_ok = d in (64, 128) and k <= 60 and k <= n_items and B > 0
    if lower_bound is not None and mode != 'exact':
        raise _lib.IgcnError("lower_bound goes with mode='exact'")
    if mode == 'fast' and not fast_ok:
        raise _lib.IgcnError('the two-stage top-k path needs d == 64 or 128 and k <= 60 (got d=%d k=%d)' % (d, k))
    if mode == 'fast' or (mode == 'auto' and fast_ok and B * n_items >= FAST_TOPK_MIN_WORK):
        return _score_topk_fast(L, user_rows, item_rows, k, user_ids, excl_rowptr, excl_col, banned, B, n_items, d)
    ws_bytes = L.igcn_score_topk_workspace_bytes(B, n_items, d, k)
    if ws_bytes < 0:
        raise _lib.IgcnError('unsupported top-k shape: batch=%d n_items=%d d=%d k=%d (need d%%4==0, d<=256, '
                             'k<=%d, k<=n_items)' % (B, n_items, d, k, _lib.MAX_TOPK))
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=item_rows.device)
    out_idx = torch.empty((B, k), dtype=torch.int64, device=item_rows.device)
    out_val = torch.empty((B, k), dtype=torch.float32, device=item_rows.device)
    if lower_bound is not None:
        if lower_bound.dtype != torch.float32 or lower_bound.numel() != B or not lower_bound.is_cuda or not lower_bound.is_contiguous():
            raise _lib.IgcnError('lower_bound must be a contiguous float32 [batch] tensor on the GPU')
        _lib.check(L.igcn_score_topk_bounded_f32(
            user_rows.data_ptr(), user_rows.stride(0), _lib.ptr(user_ids), B,
            item_rows.data_ptr(), item_rows.stride(0), n_items, d,
            _lib.ptr(excl_rowptr), _lib.ptr(excl_col), _lib.ptr(banned), k, lower_bound.data_ptr(),
            out_idx.data_ptr(), out_val.data_ptr(), ws.data_ptr(), _lib.current_stream()), 'igcn_score_topk_bounded_f32')
        # A bound above a user's true k-th best score (an invalid one, or one that is an ulp high) leaves that user's list
        # short instead of wrong: such users — and the ones who really have fewer than k unmasked items — go through the
        # plain fp32 sweep again.  One small host read, on a path that is only taken for a handful of users.
        short = out_idx[:, k - 1] < 0
        if bool(short.any()):
            pos = torch.nonzero(short).flatten()
            ids = user_ids[pos] if user_ids is not None else pos
            idx_s, val_s = score_topk(user_rows, item_rows, k, user_ids=ids.contiguous(), excl_rowptr=excl_rowptr, excl_col=excl_col,
                                      banned=banned, mode='exact')
            out_idx[pos] = idx_s
            out_val[pos] = val_s
        return out_idx, out_val
    _lib.check(L.igcn_score_topk_f32(
        user_rows.data_ptr(), user_rows.stride(0), _lib.ptr(user_ids), B,
        item_rows.data_ptr(), item_rows.stride(0), n_items, d,
        _lib.ptr(excl_rowptr), _lib.ptr(excl_col), _lib.ptr(banned), k,
        out_idx.data_ptr(), out_val.data_ptr(), ws.data_ptr(), _lib.current_stream()), 'igcn_score_topk_f32')
    return out_idx, out_val


def set_fast_fallback(inside):
    """Developer / test switch: whether igcn_score_topk_fast_f32 finishes its first FAST_FALLBACK_MAX flagged users itself
    (False = the ABI v6 split: every flagged user re-done from here).  The knob lives in the library and only there:
    _score_topk_fast asks it how many users a call finishes (igcn_score_topk_fast_finished_max)."""
    _lib.set_tuning('topk_fast_fallback', None if inside else 0)


FAST_WORKSPACE_FILL = None        # test hook: a byte value the two-stage call's workspace is filled with before the call (None: left as allocated)


def _score_topk_fast(L, user_rows, item_rows, k, user_ids, excl_rowptr, excl_col, banned, B, n_items, d):
    dev = item_rows.device
    excl_rows = excl_rowptr.numel() - 1 if excl_rowptr is not None else 0
    excl_nnz = excl_col.numel() if excl_rowptr is not None else 0
    ws_bytes = L.igcn_score_topk_fast_workspace_bytes(B, n_items, d, k, excl_rows, excl_nnz)
    if ws_bytes < 0:
        raise _lib.IgcnError('unsupported two-stage top-k shape: batch=%d n_items=%d d=%d k=%d' % (B, n_items, d, k))
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev)
    if FAST_WORKSPACE_FILL is not None:                      # tests: the call must not rely on what the workspace held before
        ws.fill_(FAST_WORKSPACE_FILL)
    ws_ptr = (ws.data_ptr() + 255) // 256 * 256
    out_idx = torch.empty((B, k), dtype=torch.int64, device=dev)
    out_val = torch.empty((B, k), dtype=torch.float32, device=dev)
    flagged = torch.empty(B + 1, dtype=torch.int32, device=dev)
    bounds = torch.empty(B, dtype=torch.float32, device=dev)
    # how many flagged users the call will finish itself: asked of the library, BEFORE the call, under the knobs the call will see
    done = int(L.igcn_score_topk_fast_finished_max(B, 1))
    _lib.check(L.igcn_score_topk_fast_f32(
        user_rows.data_ptr(), user_rows.stride(0), _lib.ptr(user_ids), B,
        item_rows.data_ptr(), item_rows.stride(0), n_items, d,
        _lib.ptr(excl_rowptr), _lib.ptr(excl_col), excl_rows, excl_nnz, _lib.ptr(banned), k,
        out_idx.data_ptr(), out_val.data_ptr(), flagged.data_ptr(), bounds.data_ptr(), ws_ptr, _lib.current_stream()),
        'igcn_score_topk_fast_f32')
    # The call has already re-done its first FAST_FALLBACK_MAX flagged users with the bounded fp32 sweep, planned on the device
    # (ABI v7).  The one host read of the path comes AFTER everything is queued — the GPU is never idle waiting for it — and only
    # tells whether more users than that were flagged (exact-arithmetic tables with ties in droves; a few dozen is the norm).
    n_flagged = int(flagged[0].item())
    if n_flagged > done:
        # the rest: the fp32 sweep, started from the k-th exact score of their candidates (a valid lower bound) instead of
        # from an empty list
        pos = flagged[1 + done:1 + n_flagged].long()
        ids = user_ids[pos] if user_ids is not None else pos
        idx_f, val_f = score_topk(user_rows, item_rows, k, user_ids=ids.contiguous(), excl_rowptr=excl_rowptr, excl_col=excl_col,
                                  banned=banned, mode='exact', lower_bound=bounds[done:n_flagged].contiguous())
        out_idx[pos] = idx_f
        out_val[pos] = val_f
    score_topk.last_flagged = n_flagged                     # developer statistic
    return out_idx, out_val


def hit_matrix(rec, eval_rowptr, eval_col):
    """float32 [U, k] of 0/1: rec[u, j] in eval list of u (sorted CSR)."""
    _require_i64(rec, 'rec')
    _require_i64(eval_rowptr, 'eval_rowptr')
    hit = torch.empty(rec.shape, dtype=torch.float32, device=rec.device)
    # a split whose lists are all empty has no column array to point at: the kernel then reports no hit anywhere
    col_ptr = eval_col.data_ptr() if eval_col is not None and eval_col.numel() else None
    _lib.check(_lib.lib().igcn_hit_matrix(rec.data_ptr(), rec.shape[0], rec.shape[1], eval_rowptr.data_ptr(),
                                          col_ptr, hit.data_ptr(), _lib.current_stream()), 'igcn_hit_matrix')
    return hit


def eval_metric_sums(rec, eval_rowptr, eval_col, topks):
    """calculate_metrics (trainer.py:109-138) fused (igcn_eval_metrics): numpy float64 [len(topks), 3] of the sums over
    the users with a non-empty list of (hits / k, hits / |list|, DCG / IDCG), and the number of such users.  One launch
    pair and one small copy instead of the hit matrix + ~25 reductions per cut-off."""
    _require_i64(rec, 'rec')
    _require_i64(eval_rowptr, 'eval_rowptr')
    if len(topks) > _lib.MAX_METRIC_CUTS:
        raise _lib.IgcnError('at most %d cut-offs per call' % _lib.MAX_METRIC_CUTS)
    rec = rec.contiguous()
    L = _lib.lib()
    ws = torch.empty(max(int(L.igcn_eval_metrics_workspace_bytes(rec.shape[0])), 8), dtype=torch.uint8, device=rec.device)
    out = torch.empty(3 * _lib.MAX_METRIC_CUTS + 1, dtype=torch.float64, device=rec.device)
    cuts = (C.c_int32 * len(topks))(*[int(k) for k in topks])
    col_ptr = eval_col.data_ptr() if eval_col is not None and eval_col.numel() else None
    _lib.check(L.igcn_eval_metrics(rec.data_ptr(), rec.shape[0], rec.shape[1], eval_rowptr.data_ptr(), col_ptr,
                                   C.addressof(cuts), len(topks), out.data_ptr(), ws.data_ptr(), _lib.current_stream()), 'igcn_eval_metrics')
    host = out.cpu().numpy()
    return host[:3 * len(topks)].reshape(len(topks), 3), float(host[-1])


def bpr_sample(train_rowptr, train_col, nonempty_users, n_items, batch, seed, out=None):
    """int64 [batch, 3] (user, pos, neg) drawn on the device (igcn_bpr_sample).  out: as bpr_sample_nodes."""
    _require_i64(train_rowptr, 'train_rowptr')
    if out is None:
        out = torch.empty((batch, 3), dtype=torch.int64, device=train_rowptr.device)
    elif out.dtype != torch.int64 or tuple(out.shape) != (batch, 3) or not out.is_contiguous() or out.device != train_rowptr.device:
        raise _lib.IgcnError('out must be a contiguous int64 [batch, 3] tensor on the sampler\'s device')
    _lib.check(_lib.lib().igcn_bpr_sample(train_rowptr.data_ptr(), train_col.data_ptr(), nonempty_users.data_ptr(),
                                          nonempty_users.numel(), n_items, batch, int(seed) & 0xFFFFFFFFFFFFFFFF,
                                          out.data_ptr(), _lib.current_stream()), 'igcn_bpr_sample')
    return out
