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
         keep_prob=1.0, seed=0):
    """out = (out_scale * csr @ x + add_scale * sum(adds)) * row_scale  (see igcn_spmm_csr_f32)."""
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
    if col_scale is not None and (col_scale.dtype != torch.float32 or col_scale.numel() < n_cols or not col_scale.is_cuda):
        raise _lib.IgcnError('col_scale must be float32 [n_cols] on the GPU')
    add_ptrs = (C.c_void_p * max(1, len(adds)))(*[a.data_ptr() for a in adds])
    partial = csr.partial(d)
    _lib.check(_lib.lib().igcn_spmm_csr_f32(
        csr.rowptr.data_ptr(), csr.col.data_ptr(), _lib.ptr(csr.val),
        x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0),
        n_rows, n_cols, d, float(out_scale), add_ptrs, len(adds), float(add_scale), _lib.ptr(row_scale), _lib.ptr(col_scale),
        _lib.ptr(csr.long_rows), csr.n_long, _lib.ptr(csr.segments), csr.n_segments, _lib.ptr(partial),
        csr.long_threshold, _lib.ptr(csr.edge_id), int(seed) & 0xFFFFFFFFFFFFFFFF, float(keep_prob),
        _lib.current_stream()), 'igcn_spmm_csr_f32')
    return out


def propagate_mean(csr: CsrMatrix, x0, n_layers, row_scale_last=None):
    """mean(X_0..X_K), X_{l+1} = csr @ X_l — the layer loop + stack/mean of
    model.py:101-105.  The mean is the epilogue of the last SpMM (no stack)."""
    if n_layers == 0:
        return x0.clone()
    if n_layers > _lib.MAX_ADDS:
        raise _lib.IgcnError('n_layers > %d not supported' % _lib.MAX_ADDS)
    layers = [x0]
    s = 1.0 / (n_layers + 1)
    for l in range(n_layers):
        last = l == n_layers - 1
        if last:
            y = spmm(csr, layers[-1], adds=layers, out_scale=s, add_scale=s, row_scale=row_scale_last)
        else:
            y = spmm(csr, layers[-1])
        layers.append(y)
    return layers[-1]


def propagate_mean_backward(csr_t: CsrMatrix, grad, n_layers, row_scale=None):
    """d/dX_0 of propagate_mean:  s * sum_l (M^T)^l g  by Horner's rule,
    G <- s*g + M^T G, K times — one SpMM per layer with the add fused.
    row_scale (optional) multiplies the final rows (used by the INMO path)."""
    s = 1.0 / (n_layers + 1)
    if n_layers == 0:
        return grad.clone()
    g = grad.contiguous()
    cur = None
    for l in range(n_layers):
        last = l == n_layers - 1
        rs = row_scale if last else None
        if cur is None:
            cur = spmm(csr_t, g, adds=[g], out_scale=s, add_scale=s, row_scale=rs)
        else:
            cur = spmm(csr_t, cur, adds=[g], out_scale=1.0, add_scale=s, row_scale=rs)
    return cur


class PropagateFn(torch.autograd.Function):
    """LightGCN.get_rep (model.py:96-106) as one autograd node.  A_hat is
    symmetric (model.py:85-94), so the backward uses the same CSR."""

    @staticmethod
    def forward(ctx, emb, csr, csr_t, n_layers):
        ctx.csr_t, ctx.n_layers = csr_t, n_layers
        return propagate_mean(csr, emb.detach(), n_layers)

    @staticmethod
    def backward(ctx, grad):
        return propagate_mean_backward(ctx.csr_t, grad, ctx.n_layers), None, None, None


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
