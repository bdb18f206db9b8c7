// Device-side CSR index utilities for the graph-swap path (a trained INMO model gets a new
// interaction graph / template-feature matrix without retraining: model.py:402-421,
// run/dropui/igcn_dropui.py:26-35).  The reference builds every sparse structure on the host with
// scipy and re-coalesces COO tensors; here a CSR and its transposed view (needed for the backward
// of the rectangular feature layer) can be derived without leaving HBM.
//   igcn_csr_from_sorted_coo: rowptr[r] = lower_bound(sorted_row, r)  (one thread per row)
//   igcn_csr_transpose      : stable radix sort of the column ids (rocPRIM device_radix_sort — the
//                             one library primitive in this .so) with the entry positions as
//                             payload; positions = edge ids of the transposed view, rows by binary
//                             search in rowptr.  Order inside a transposed row = ascending source
//                             row, identical to the host builder (graph.transpose_host).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "common.h"

namespace igcn {

template <typename T>
__global__ void lower_bound_rows_kernel(const T *__restrict__ sorted, int64_t n, int64_t n_rows, int64_t *__restrict__ rowptr)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted[mid] < r) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = lo;
}

__global__ void iota_kernel(int32_t *__restrict__ v, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = (int32_t)i;
}

// t_col[q] = row that owns entry edge_id[q]  (upper_bound in rowptr)
__global__ void rows_of_entries_kernel(const int64_t *__restrict__ rowptr, int64_t n_rows, const int32_t *__restrict__ edge_id,
                                       int64_t nnz, int32_t *__restrict__ t_col)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz) return;
    const int64_t p = edge_id[q];
    int64_t lo = 0, hi = n_rows;                 // first row r with rowptr[r + 1] > p
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (rowptr[mid + 1] <= p) lo = mid + 1; else hi = mid;
    }
    t_col[q] = (int32_t)lo;
}

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static hipError_t sort_temp_bytes(int64_t nnz, size_t *bytes)
{
    *bytes = 0;
    return rocprim::radix_sort_pairs(nullptr, *bytes, (const int32_t *)nullptr, (int32_t *)nullptr, (const int32_t *)nullptr,
                                     (int32_t *)nullptr, (size_t)nnz, 0, 32, (hipStream_t)0);
}

}  // namespace igcn

using namespace igcn;

extern "C" int igcn_csr_from_sorted_coo(const int64_t *sorted_row, int64_t nnz, int64_t n_rows, int64_t *rowptr, void *stream)
{
    if (!rowptr || (nnz > 0 && !sorted_row)) return IGCN_E_NULL;
    if (nnz < 0 || n_rows < 0) return IGCN_E_SHAPE;
    hipLaunchKernelGGL((lower_bound_rows_kernel<int64_t>), dim3((unsigned)((n_rows + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), sorted_row, nnz, n_rows, rowptr);
    return launch_status();
}

extern "C" int64_t igcn_csr_transpose_workspace_bytes(int64_t nnz)
{
    if (nnz < 0 || nnz >= ((int64_t)1 << 31)) return -1;
    size_t tmp = 0;
    if (nnz > 0 && sort_temp_bytes(nnz, &tmp) != hipSuccess) return -1;
    return (int64_t)(2 * align256((size_t)nnz * 4) + align256(tmp) + 256);
}

extern "C" int igcn_csr_transpose(const int64_t *rowptr, const int32_t *col, int64_t n_rows, int64_t n_cols, int64_t nnz,
                                  int64_t *t_rowptr, int32_t *t_col, int32_t *edge_id, void *workspace, void *stream)
{
    if (!rowptr || !t_rowptr) return IGCN_E_NULL;
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || nnz >= ((int64_t)1 << 31)) return IGCN_E_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        // rocPRIM's one-sweep radix sort kernels carry a private segment (80 bytes a lane as built here): the one call of the library
        // that is refused on a capturing stream (see capture_guard in common.h).  A graph build is not something to replay anyway.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return IGCN_E_CAPTURE;
        (void)hipGetLastError();
    }
    if (nnz == 0) {
        return zero_async(t_rowptr, (size_t)(n_cols + 1) * 8, st);
    }
    if (!col || !t_col || !edge_id || !workspace) return IGCN_E_NULL;
    if (reinterpret_cast<uintptr_t>(workspace) % 256) return IGCN_E_ALIGN;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    int32_t *sorted_col = reinterpret_cast<int32_t *>(ws);
    int32_t *iota = reinterpret_cast<int32_t *>(ws + align256((size_t)nnz * 4));
    void *tmp = ws + 2 * align256((size_t)nnz * 4);
    size_t tmp_bytes = 0;
    hipError_t e = sort_temp_bytes(nnz, &tmp_bytes);
    if (e != hipSuccess) return (int)e;
    const unsigned blocks = (unsigned)((nnz + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(kBlock), 0, st, iota, nnz);
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, col, sorted_col, (const int32_t *)iota, edge_id, (size_t)nnz, 0, 32, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((lower_bound_rows_kernel<int32_t>), dim3((unsigned)((n_cols + 1 + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                       (const int32_t *)sorted_col, nnz, n_cols, t_rowptr);
    hipLaunchKernelGGL(rows_of_entries_kernel, dim3(blocks), dim3(kBlock), 0, st, rowptr, n_rows, (const int32_t *)edge_id, nnz, t_col);
    return launch_status();
}
