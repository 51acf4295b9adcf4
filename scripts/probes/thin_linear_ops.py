#!/usr/bin/env python3
"""Which torch op to use for the THIN layers of config 5's network (Linear(2, 128), Linear(128, 1)) on 65 536 rows, forward and
weight gradient: the library GEMMs autograd picks take 100 - 220 us each here (profiles/r05_c5_step_kernels.txt)."""
import time

import torch
import torch.nn.functional as F

dev = torch.device("cuda")
N = 65536
x = torch.randn(N, 2, device=dev)
W1 = torch.randn(128, 2, device=dev)
b1 = torch.randn(128, device=dev)
g = torch.randn(N, 128, device=dev)
h2 = torch.randn(N, 128, device=dev)
w3 = torch.randn(1, 128, device=dev)
b3 = torch.randn(1, device=dev)
ge = torch.randn(N, device=dev)


def t(name, fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - t0) / reps * 1e6:8.1f} us  {name}", flush=True)


t("fwd in : F.linear(x, W1, b1)", lambda: F.linear(x, W1, b1))
t("fwd in : broadcast  b1 + x0*W1[:,0] + x1*W1[:,1]", lambda: torch.addcmul(torch.addcmul(b1, x[:, 0:1], W1[:, 0]), x[:, 1:2], W1[:, 1]))
t("dW1    : g.t() @ x", lambda: g.t() @ x)
t("dW1    : (x.t() @ g).t()", lambda: (x.t() @ g).t())
t("dW1    : 2 x torch.mv(g.t(), x[:, c])", lambda: (torch.mv(g.t(), x[:, 0].contiguous()), torch.mv(g.t(), x[:, 1].contiguous())))
t("dW1    : 2 x (g * x[:, c:c+1]).sum(0)", lambda: ((g * x[:, 0:1]).sum(0), (g * x[:, 1:2]).sum(0)))
t("dW1    : einsum nc,nh->hc", lambda: torch.einsum("nc,nh->hc", x, g))
t("dx     : g @ W1", lambda: g @ W1)
t("dx     : 2 x torch.mv(g, W1[:, c])", lambda: torch.stack((torch.mv(g, W1[:, 0].contiguous()), torch.mv(g, W1[:, 1].contiguous())), 1))
t("db1    : g.sum(0)", lambda: g.sum(0))
t("fwd out: F.linear(h2, w3, b3)", lambda: F.linear(h2, w3, b3))
t("fwd out: torch.mv(h2, w3[0]) + b3", lambda: torch.mv(h2, w3[0]) + b3)
t("dw3    : ge[None] @ h2", lambda: ge[None] @ h2)
t("dw3    : torch.mv(h2.t(), ge)", lambda: torch.mv(h2.t(), ge))
t("dw3    : (h2 * ge[:, None]).sum(0)", lambda: (h2 * ge[:, None]).sum(0))
t("dh2    : ge[:, None] @ w3", lambda: ge[:, None] @ w3)
t("dh2    : ge[:, None] * w3", lambda: ge[:, None] * w3)
t("mid    : F.linear(h2, W2[128,128], b)", lambda: F.linear(h2, g[:128].contiguous(), b1))
# ---- the SQUARE layer's weight gradient dW2 = dz2^T h1: a 128 x 128 output over K = 65 536 (the library picks a 32 x 32 macro
# tile without split-K: 16 workgroups on a 256-CU chip)
dz2 = torch.randn(N, 128, device=dev)
h1 = torch.randn(N, 128, device=dev)
t("dW2    : dz2.t() @ h1", lambda: dz2.t() @ h1)
for S in (8, 32, 64, 128, 256):
    t(f"dW2    : bmm over {S} row blocks + sum", lambda S=S: torch.bmm(dz2.view(S, N // S, 128).transpose(1, 2), h1.view(S, N // S, 128)).sum(0))
ref = dz2.double().t() @ h1.double()
a = (dz2.t() @ h1).double()
b = torch.bmm(dz2.view(64, N // 64, 128).transpose(1, 2), h1.view(64, N // 64, 128)).sum(0).double()
print("max |err| vs fp64: gemm", (a - ref).abs().max().item(), " bmm64+sum", (b - ref).abs().max().item())
# ---- the same row-block trick for the thin gradients (n = 131 072: data and negatives in one call)
N2 = 2 * N
dzz = torch.randn(N2, 128, device=dev)
xx = torch.randn(N2, 2, device=dev)
gee = torch.randn(N2, device=dev)
S = 32
t("dw3 2n : (h * ge[:, None]).sum(0)", lambda: (dzz * gee[:, None]).sum(0))
t("dw3 2n : bmm blocks [S,1,n/S] x [S,n/S,128] + sum", lambda: torch.bmm(gee.view(S, 1, N2 // S), dzz.view(S, N2 // S, 128)).sum(0))
t("dW1 2n : 2 x (dz * x[:, c:c+1]).sum(0)", lambda: ((dzz * xx[:, 0:1]).sum(0), (dzz * xx[:, 1:2]).sum(0)))
t("dW1 2n : bmm blocks dz^T x + sum", lambda: torch.bmm(dzz.view(S, N2 // S, 128).transpose(1, 2), xx.view(S, N2 // S, 2)).sum(0))
x1 = torch.cat((xx, torch.ones(N2, 1, device=dev)), 1)
t("dW1+db1 2n : bmm blocks dz^T [x 1] + sum", lambda: torch.bmm(dzz.view(S, N2 // S, 128).transpose(1, 2), x1.view(S, N2 // S, 3)).sum(0))
t("db 2n  : dz.sum(0)", lambda: dzz.sum(0))
t("db 2n  : bmm blocks ones^T dz + sum", lambda: torch.bmm(torch.ones(S, 1, N2 // S, device=dev), dzz.view(S, N2 // S, 128)).sum(0))
# ---- wider inputs (the reference's benchmark network at dim 8 / 32 / 128): is the first layer's weight gradient slow there too?
for d in (8, 32, 128):
    xd = torch.randn(N, d, device=dev)
    x1d = torch.cat((xd, torch.ones(N, 1, device=dev)), 1)
    t(f"dW1 d={d}: g.t() @ x", lambda xd=xd: g.t() @ xd)
    t(f"dW1+db1 d={d}: bmm blocks g^T [x 1] + sum", lambda x1d=x1d: torch.bmm(g.view(32, N // 32, 128).transpose(1, 2), x1d.view(32, N // 32, x1d.shape[1])).sum(0))
# ---- the HIDDEN-major layout of ebm_mlp_backward_acts_f32 ([H, n] rows contiguous): products with thin right-hand sides
A = torch.randn(128, N2, device=dev)
B2 = torch.randn(128, N2, device=dev)
v1 = torch.randn(N2, device=dev)
x3 = torch.randn(N2, 3, device=dev)
t("[H,n] : A.sum(1)", lambda: A.sum(1))
t("[H,n] : torch.mv(A, v)", lambda: torch.mv(A, v1))
t("[H,n] : (A * v).sum(1)", lambda: (A * v1).sum(1))
t("[H,n] : A @ x3  (q = 3)", lambda: A @ x3)
t("[H,n] : bmm blocks A @ x3", lambda: torch.bmm(A.unflatten(1, (32, N2 // 32)).permute(1, 0, 2), x3.view(32, N2 // 32, 3)).sum(0))
t("[H,n] : 3 x torch.mv(A, x3[:, c])", lambda: [torch.mv(A, x3[:, c].contiguous()) for c in range(3)])
t("[H,n] : A @ B^T (dW2)", lambda: A @ B2.t())
t("[H,n] : bmm blocks A @ B^T", lambda: torch.bmm(A.unflatten(1, (32, N2 // 32)).permute(1, 0, 2), B2.t().view(32, N2 // 32, 128) if False else B2.unflatten(1, (32, N2 // 32)).permute(1, 2, 0)).sum(0))
