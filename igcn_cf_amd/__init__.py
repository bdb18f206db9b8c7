"""MI355X-native propagation + scoring path for INMO / LightGCN collaborative
filtering (hand-written gfx950 HIP kernels behind a C ABI, PyTorch-ROCm host)."""
__version__ = '0.1.0'
