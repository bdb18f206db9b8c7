"""ctypes binding of libigcn_hip.so (C ABI declared in include/igcn_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
symbol is absent, importing an op raises immediately.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# IGCN_LIB_PATH: developer override (an instrumented host build, an A/B build of another round) — the shipped library is never
# overwritten for that (scripts/sanitize_host.sh used to copy its ASan build over it)
LIB_PATH = os.environ.get('IGCN_LIB_PATH') or os.path.join(_PKG, 'libigcn_hip.so')

EXPECTED_ABI = int(os.environ.get('IGCN_EXPECT_ABI') or 10)   # IGCN_ABI_VERSION of include/igcn_hip.h this binding was written against (env: developer A/Bs against a build of an earlier round)
MAX_ADDS = 8
MAX_TOPK = 256
MAX_METRIC_CUTS = 8
FAST_FALLBACK_MAX = 256          # IGCN_FAST_FALLBACK_MAX: flagged users igcn_score_topk_fast_f32 finishes by itself

c_i64_p = C.POINTER(C.c_int64)
vp = C.c_void_p

ROW_SEGMENT_DTYPE = np.dtype([('start', '<i8'), ('len', '<i4'), ('slot', '<i4'), ('row', '<i4'), ('long_index', '<i4')])
LONG_ROW_DTYPE = np.dtype([('row', '<i4'), ('first_slot', '<i4'), ('n_slots', '<i4'), ('reserved', '<i4')])



class SpmmArgs(C.Structure):
    """igcn_spmm_args (include/igcn_hip.h, ABI v10): zero-initialised by ctypes; zero / NULL = not used / library default."""
    _fields_ = [('struct_size', C.c_uint32), ('flags', C.c_uint32),
                ('rowptr', vp), ('col', vp), ('val', vp), ('n_rows', C.c_int64), ('n_cols', C.c_int64), ('x', vp), ('y', vp), ('d', C.c_int32),
                ('n_adds', C.c_int32), ('ldx', C.c_int64), ('ldy', C.c_int64), ('nnz', C.c_int64),
                ('out_scale', C.c_float), ('add_scale', C.c_float), ('keep_prob', C.c_float), ('long_threshold', C.c_int32),
                ('adds', vp * MAX_ADDS), ('row_scale', vp), ('col_scale', vp),
                ('long_rows', vp), ('n_long_rows', C.c_int64), ('segments', vp), ('n_segments', C.c_int64), ('partial', vp),
                ('edge_id', vp), ('seed', C.c_uint64), ('seed_dev', vp), ('row_mask', vp), ('col_mask', vp), ('order_bits', vp),
                ('row_order', vp), ('xcd_off', vp),
                ('tune_blocks_per_cu', C.c_int32), ('tune_multirow', C.c_int32), ('tune_fold', C.c_int32), ('reserved', C.c_int32)]


# name -> (restype, argtypes); every symbol of include/igcn_hip.h
SIGNATURES = {
    'igcn_abi_version': (C.c_int, []),
    'igcn_error_string': (C.c_char_p, [C.c_int]),
    'igcn_set_tuning': (C.c_int, [C.c_char_p, C.c_int32]),
    'igcn_spmm_plan_count_host': (C.c_int, [vp, C.c_int64, C.c_int32, C.c_int32, c_i64_p, c_i64_p]),
    'igcn_spmm_plan_fill_host': (C.c_int, [vp, C.c_int64, C.c_int32, C.c_int32, vp, C.c_int64, vp, C.c_int64]),
    'igcn_spmm_csr_f32': (C.c_int, [vp, vp, vp, vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                    C.c_float, C.POINTER(vp), C.c_int32, C.c_float, vp, vp,
                                    vp, C.c_int64, vp, C.c_int64, vp, C.c_int32,
                                    vp, C.c_uint64, C.c_float, vp, C.c_int32, C.c_int64, vp, vp, vp, vp, vp, vp]),
    'igcn_spmm_csr_f32_args': (C.c_int, [C.POINTER(SpmmArgs), vp]),
    'igcn_mark_rows': (C.c_int, [vp, C.c_int64, vp, vp, vp, vp, C.c_int64, vp]),
    'igcn_pack_mask_bits': (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32, vp, vp]),
    'igcn_pack_mask_bits_ordered': (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32, vp, vp, C.c_int64, vp, C.c_int64, vp, vp]),
    'igcn_csr_from_sorted_coo': (C.c_int, [vp, C.c_int64, C.c_int64, vp, vp]),
    'igcn_csr_transpose_workspace_bytes': (C.c_int64, [C.c_int64]),
    'igcn_csr_transpose': (C.c_int, [vp, vp, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp, vp]),
    'igcn_csr_row_pow_f32': (C.c_int, [vp, vp, C.c_float, vp, vp, C.c_int64, vp]),
    'igcn_bpr_fwd_f32': (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, vp,
                                   C.c_int64, C.c_int32, vp, vp, vp, vp]),
    'igcn_bpr_dots_f32': (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, vp,
                                    C.c_int64, C.c_int32, vp, vp, vp]),
    'igcn_bpr_finish_f32': (C.c_int, [vp, C.c_int64, vp, vp, vp]),
    'igcn_bpr_bwd_f32': (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, vp,
                                   C.c_int64, C.c_int32, vp, vp, vp,
                                   vp, vp, vp, vp, vp, vp, vp, vp]),
    'igcn_score_topk_workspace_bytes': (C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    'igcn_score_topk_f32': (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int32,
                                      vp, vp, vp, C.c_int32, vp, vp, vp, vp]),
    'igcn_score_topk_bounded_f32': (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int32,
                                              vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp]),
    'igcn_score_topk_fast_workspace_bytes': (C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64]),
    'igcn_score_topk_fast_finished_max': (C.c_int64, [C.c_int64, C.c_int32]),
    'igcn_score_topk_fast_f32': (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int64, C.c_int32,
                                           vp, vp, C.c_int64, C.c_int64, vp, C.c_int32, vp, vp, vp, vp, vp, vp]),
    'igcn_hit_matrix': (C.c_int, [vp, C.c_int64, C.c_int32, vp, vp, vp, vp]),
    'igcn_eval_metrics_workspace_bytes': (C.c_int64, [C.c_int64]),
    'igcn_eval_metrics': (C.c_int, [vp, C.c_int64, C.c_int32, vp, vp, vp, C.c_int32, vp, vp, vp]),
    'igcn_bpr_sample': (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, vp, vp]),
    'igcn_bpr_sample_nodes': (C.c_int, [vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_uint64, C.c_int64, vp, vp]),
    'igcn_rows_finish_f32': (C.c_int, [vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, vp, C.c_int64, C.c_int32, vp, C.c_float, vp]),
    'igcn_owned_rows_gather_f32': (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                             vp, C.c_int64, vp, C.c_int64, C.c_int32, vp, C.c_int64, vp]),
    'igcn_owned_rows_scatter_add_f32': (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                                  vp, C.c_int64, C.c_int32, vp, C.c_int64, vp, C.c_int64, vp]),
    'igcn_bpr_loss_f32': (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, vp,
                                    C.c_int64, C.c_int32, vp, C.c_float, vp, vp, vp]),
    'igcn_bpr_loss_bwd_f32': (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, vp,
                                        C.c_int64, C.c_int32, vp, vp, vp, C.c_float,
                                        vp, vp, vp, vp, vp, vp, vp, vp]),
    'igcn_bpr_loss_bwd_scaled_f32': (C.c_int, [vp, vp, vp, C.c_int64, vp, vp, vp, C.c_int64, vp, vp, vp,
                                               C.c_int64, C.c_int32, vp, vp, vp, C.c_float, C.c_float,
                                               vp, vp, vp, vp, vp, vp, vp, vp]),
}

_handle = None
_bound = {}


class IgcnError(RuntimeError):
    pass


def handle():
    """The loaded shared library; raises if it is missing (no fallback)."""
    global _handle
    if _handle is None:
        # torch first: its wheel bundles a HIP runtime of its own (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's, which
        # this library is linked against).  Whichever is loaded first serves the whole process; loaded the other way round
        # the process ends up with two runtimes and this library's first launch fails with "no ROCm-capable device"
        # (seen when __graft_entry__.build() and smoke() ran in one process on the GPU box).
        import torch                                            # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise IgcnError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                            '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
        h = C.CDLL(LIB_PATH)
        # a stale library (the build keeps a .so that is newer than its sources) must not load silently: the wrappers rely on the
        # current call contracts (who finishes which flagged users, buffer layouts)
        try:
            h.igcn_abi_version.restype = C.c_int
            got = int(h.igcn_abi_version())
        except AttributeError:
            got = -1
        if got != EXPECTED_ABI:
            raise IgcnError('%s has ABI version %d, this package expects %d: rebuild it (python -c "import __graft_entry__ as g; '
                            'g.build()", or igcn_cf_amd/_build.py --force)' % (LIB_PATH, got, EXPECTED_ABI))
        _handle = h
    return _handle


class _Lib:
    """Attribute access binds the symbol with its declared signature on first use
    and raises IgcnError when the library does not export it."""

    def __getattr__(self, name):
        fn = _bound.get(name)
        if fn is None:
            if name not in SIGNATURES:
                raise AttributeError(name)
            try:
                fn = getattr(handle(), name)
            except AttributeError:
                raise IgcnError('libigcn_hip.so does not export %s; rebuild it' % name)
            fn.restype, fn.argtypes = SIGNATURES[name]
            _bound[name] = fn
        return fn


_LIB = _Lib()


def lib():
    return _LIB


def check(code, what):
    if code != 0:
        msg = lib().igcn_error_string(int(code))
        raise IgcnError('%s failed: %s (code %d)' % (what, msg.decode() if msg else '?', code))


def set_tuning(name, value=None):
    """Developer / test knob of the launch heuristics (igcn_set_tuning); value None restores the default."""
    check(lib().igcn_set_tuning(name.encode(), -1 if value is None else int(value)), 'igcn_set_tuning(%s)' % name)


def ptr(t):
    """Raw device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
