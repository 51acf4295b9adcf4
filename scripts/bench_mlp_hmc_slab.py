"""HMC on the benchmark MLP at H = 128, dims 96 / 128 (csrc/mlp_wide_hmc_slab.hip: MODE 3, W1's split image streamed through LDS)
beside the same call without the image (MODE 0: exact-f32 MFMA, fp32 weights in LDS): kernel ms per 10 transitions of 10 leapfrog steps."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchebm_amd as ta
from torchebm_amd import _lib
dev = torch.device("cuda")
n, T, L = 65536, 10, 10
for dim in (96, 128):
    torch.manual_seed(0)
    model = ta.MLPEnergy(dim, 128, device=dev)
    x0 = torch.randn(n, dim, device=dev)
    spec = model.fused_spec()
    row = {"case": f"HMC on MLP {dim}-128-128-1, n={n}, L={L}, {T} transitions"}
    for label in ("with_image", "aux_null"):
        d = spec.to_c()
        if label == "aux_null":
            d.aux = None
        mask = torch.empty(T, n, dtype=torch.uint8, device=dev)
        def fn():
            x = x0.clone()
            _lib.call("ebm_hmc_chain_f32", d, x.data_ptr(), n, dim, T, L, 0.05, None, 0, 0.0, None, 1, None, None, mask.data_ptr(), None,
                      None, None, 7, 0, _lib.stream_handle(dev))
            return x
        fn(); fn()
        _lib.timed_events["ebm_hmc_chain_f32"] = []
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in _lib.timed_events.pop("ebm_hmc_chain_f32"))
        row[label + "_ms"] = ts[1]
        row[label + "_accept"] = float(mask.float().mean())
        row[label + "_finite"] = bool(torch.isfinite(out).all())
    print(json.dumps(row), flush=True)
