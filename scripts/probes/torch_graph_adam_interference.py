#!/usr/bin/env python3
"""Plain PyTorch, nothing of this package: a whole training step captured in a CUDA/HIP graph the way torch's own
documentation prescribes (side-stream warm-up, Adam(capturable=True)), replayed (a) alone and (b) with EAGER steps of a
second, unrelated Adam(capturable=True) in between -- all on one stream, no synchronisation.  Prints the first step at which
the replayed loss curve leaves the eager one.  (Round 5: tests/test_graphed_step_gpu.py first compared the two trainers
interleaved and saw the graphed one drift; this probe shows where that comes from.)"""
import torch

dev = torch.device("cuda")


def make(seed=0):
    torch.manual_seed(seed)
    m = torch.nn.Sequential(torch.nn.Linear(2, 128), torch.nn.SiLU(), torch.nn.Linear(128, 128), torch.nn.SiLU(), torch.nn.Linear(128, 1)).to(dev)
    return m, torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)


x = torch.randn(4096, 2, device=dev)


def step(m, o):
    o.zero_grad(set_to_none=True)
    loss = m(x).square().mean()
    loss.backward()
    o.step()
    return loss.detach()


def eager_run(steps):
    m, o = make()
    return torch.stack([step(m, o) for _ in range(steps)]).cpu()


def graph_run(steps, between):
    m, o = make()
    out = []
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            out.append(step(m, o).clone())
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    o.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        static_loss = m(x).square().mean()
        static_loss.backward()
        o.step()
    for i in range(steps - 3):
        between(i)
        g.replay()
        out.append(static_loss.detach().clone())
    return torch.stack(out).cpu()


R = eager_run(40)
m2, o2 = make(1)
a = graph_run(40, lambda i: None)
b = graph_run(40, lambda i: step(m2, o2))
m3 = make(2)[0]
o3 = torch.optim.SGD(m3.parameters(), lr=1e-3)
c = graph_run(40, lambda i: step(m3, o3))
for name, v in (("graph alone", a), ("graph + eager Adam(capturable=True) steps of another model in between", b), ("graph + eager SGD steps in between", c)):
    d = (v != R).nonzero().flatten().tolist()
    print(f"{name}: first differing step {d[:1]}, max |loss - eager| {(v - R).abs().max().item():.3e}")
