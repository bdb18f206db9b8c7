"""ctypes loader of oracle/liboracle_c.so (C restatement; checker / CPU baseline only)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# IGCN_ORACLE_LIB_PATH: developer override (the ASan / UBSan build of scripts/sanitize_host.sh); the built file is never overwritten
LIB = os.environ.get('IGCN_ORACLE_LIB_PATH') or os.path.join(HERE, 'liboracle_c.so')
_lib = None


def build():
    subprocess.check_call(['make', '-s', '-C', HERE, 'liboracle_c.so'])
    return LIB


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads():
    return int(lib().oracle_num_threads())


def spmm_csr(rowptr, col, val, x, threads=0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    n_rows = rowptr.shape[0] - 1
    y = np.empty((n_rows, x.shape[1]), dtype=np.float32)
    lib().oracle_spmm_csr_f32(_p(rowptr), _p(col), _p(val), _p(x), _p(y), C.c_int64(n_rows), C.c_int32(x.shape[1]),
                              C.c_int(threads))
    return y


def propagate_mean(rowptr, col, val, x0, n_layers, threads=0):
    x0 = np.ascontiguousarray(x0, dtype=np.float32)
    n_rows, d = x0.shape
    out = np.empty_like(x0)
    work = np.empty((2, n_rows, d), dtype=np.float32)
    lib().oracle_propagate_mean_f32(_p(rowptr), _p(col), _p(val), _p(x0), _p(out), _p(work), C.c_int64(n_rows),
                                    C.c_int32(d), C.c_int32(n_layers), C.c_int(threads))
    return out


def score_topk(user_rows, item_rows, k, user_ids=None, excl_rowptr=None, excl_col=None, banned=None, threads=0):
    user_rows = np.ascontiguousarray(user_rows, dtype=np.float32)
    item_rows = np.ascontiguousarray(item_rows, dtype=np.float32)
    batch = user_rows.shape[0] if user_ids is None else user_ids.shape[0]
    out_idx = np.empty((batch, k), dtype=np.int64)
    out_val = np.empty((batch, k), dtype=np.float32)
    lib().oracle_score_topk_f32(_p(user_rows), _p(user_ids), C.c_int64(batch), _p(item_rows),
                                C.c_int64(item_rows.shape[0]), C.c_int32(item_rows.shape[1]), _p(excl_rowptr),
                                _p(excl_col), _p(banned), C.c_int32(k), _p(out_idx), _p(out_val), C.c_int(threads))
    return out_idx, out_val
