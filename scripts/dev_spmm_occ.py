import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['IGCN_SPMM_DEBUG'] = '1'
from igcn_cf_amd.graph import CsrMatrix
from igcn_cf_amd.ops import spmm
import numpy as np
rp = np.arange(0, 4001, 4, dtype=np.int64); col = np.random.default_rng(0).integers(0, 1000, 4000).astype(np.int32); val = np.ones(4000, np.float32)
csr = CsrMatrix(rp, col, val, (1000, 1000), 'cuda')
for d in (4, 8, 16, 32, 64, 128, 256):
    spmm(csr, torch.randn(1000, d, device='cuda'))
torch.cuda.synchronize()
