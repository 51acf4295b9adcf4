"""One Gaussian Langevin chain launch (for rocprofv3 passes):  run_gauss_chain.py <dim> <log2 n> <k> [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchebm_amd as ta
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients
dim, n, k = int(sys.argv[1]), 1 << int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda")
g = torch.Generator().manual_seed(dim)
a = torch.randn(dim, dim, generator=g)
model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
x = torch.randn(n, dim, device=dev)
spec = model.fused_spec().to_c()
aa, sq, coef = em_coefficients(0.01, 1.0)
st = _lib.stream_handle(dev)
for _ in range(reps):
    _lib.call("ebm_langevin_chain_f32", spec, x.data_ptr(), n, dim, k, aa, sq, coef, None, 0, 0.0, 0.0, 1, None, None, None, 1, 0, st)
torch.cuda.synchronize()
