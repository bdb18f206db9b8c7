"""Property tests (hypothesis) of the host-side builders against scipy / direct definitions."""
import numpy as np
import scipy.sparse as sp
from hypothesis import given, settings, strategies as st

from igcn_cf_amd import graph
from igcn_cf_amd.dataset import CsrBackedDataset, dropit_dataset, lists_to_csr, resize_dataset


@st.composite
def interaction_lists(draw):
    n_users = draw(st.integers(1, 12))
    n_items = draw(st.integers(1, 15))
    lists = [draw(st.lists(st.integers(0, n_items - 1), max_size=10)) for _ in range(n_users)]
    return n_users, n_items, lists


@settings(max_examples=60, deadline=None)
@given(interaction_lists())
def test_adjacency_and_normalisation_match_scipy(data):
    n_users, n_items, lists = data
    ta = np.array([[u, i] for u, items in enumerate(lists) for i in items], dtype=np.int64).reshape(-1, 2)
    n = n_users + n_items
    rowptr, col, val = graph.adjacency_host(ta, n_users, n_items)
    row = np.concatenate([ta[:, 0], ta[:, 1] + n_users])
    colu = np.concatenate([ta[:, 1] + n_users, ta[:, 0]])
    ref = sp.coo_matrix((np.ones(len(row)), (row, colu)), shape=(n, n), dtype=np.float32).tocsr()
    ref.sort_indices()
    np.testing.assert_array_equal(rowptr, ref.indptr)
    np.testing.assert_array_equal(col, ref.indices)
    np.testing.assert_array_equal(val, ref.data)
    rowptr, col, val = graph.normalized_adjacency_host(ta, n_users, n_items)
    deg = np.maximum(1., np.array(ref.sum(axis=1)).squeeze())
    dm = sp.diags(np.power(np.atleast_1d(deg), -0.5), format='csr', dtype=np.float32)
    want = dm.dot(ref).dot(dm).tocsr()
    want.sort_indices()
    np.testing.assert_array_equal(col, want.indices)
    np.testing.assert_array_equal(val, want.data.astype(np.float32))


@settings(max_examples=60, deadline=None)
@given(interaction_lists())
def test_transpose_and_feature_matrix(data):
    n_users, n_items, lists = data
    ta = np.array([[u, i] for u, items in enumerate(lists) for i in items], dtype=np.int64).reshape(-1, 2)
    rowptr, col, row_sum, shape = graph.feature_matrix_host(ta, n_users, n_items)
    m = sp.csr_matrix((np.ones(len(col)), col, rowptr), shape=shape)
    assert shape == (n_users + n_items, n_users + n_items + 2)
    assert (np.diff(rowptr) >= 1).all()                      # every row has its global column
    # row_sum counts duplicates (the reference sums them before overwriting the values)
    dup = np.zeros(n_users + n_items)
    np.add.at(dup, ta[:, 0], 1); np.add.at(dup, ta[:, 1] + n_users, 1)
    np.testing.assert_array_equal(row_sum, dup + 1)
    t_rowptr, t_col, eid = graph.transpose_host(rowptr, col, shape[1])
    mt = sp.csr_matrix((np.ones(len(t_col)), t_col, t_rowptr), shape=(shape[1], shape[0]))
    assert abs(mt - m.T).max() == 0
    rows = np.repeat(np.arange(shape[0]), np.diff(rowptr))
    np.testing.assert_array_equal(rows[eid], t_col)


@settings(max_examples=40, deadline=None)
@given(interaction_lists(), st.floats(0.1, 1.0))
def test_dataset_derivations(data, ratio):
    n_users, n_items, lists = data
    csrs = {name: lists_to_csr(lists) for name in ('train', 'val', 'test')}
    ds = CsrBackedDataset({'name': 'x', 'device': 'cpu'}, n_users, n_items, csrs)
    a = dropit_dataset(ds, ratio)
    for u in range(n_users):
        assert a.train_data[u] == lists[u][:int(len(lists[u]) * ratio)] and a.val_data[u] == lists[u]
    b = resize_dataset(ds, ratio)
    nu, ni = int(n_users * ratio), int(n_items * ratio)
    assert (b.n_users, b.n_items) == (nu, ni)
    for u in range(nu):
        assert b.test_data[u] == [i for i in lists[u] if i < ni]
