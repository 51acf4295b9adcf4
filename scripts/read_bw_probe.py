import torch, time
x = torch.randn(1<<20, 64, device='cuda')
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/reps
for name, fn in [("sum_all", lambda: x.sum()), ("sum_dim0", lambda: x.sum(dim=0)), ("mean_var_dim0", lambda: (x.mean(dim=0), x.var(dim=0, unbiased=False))), ("max", lambda: x.max())]:
    ms = t(fn); print(name, round(ms,4), "ms", round(x.numel()*4/ms/1e6), "GB/s")
