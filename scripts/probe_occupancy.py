"""Plain-VALU issue rate of the chip against resident waves per SIMD (ebm_probe_valu_f32: eight independent FMA chains per lane)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchebm_amd import _lib
dev = torch.device("cuda"); st = _lib.stream_handle(dev)
iters = 20000
for per_cu in (1, 2, 3, 4, 6, 8):
    blocks = 256 * per_cu
    out = torch.empty(blocks * 256, device=dev)
    f = lambda: _lib.call("ebm_probe_valu_f32", out.data_ptr(), blocks, iters, st)
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    ms = sorted(ts)[2]
    instr = blocks * 4 * iters * 8  # wave-instructions
    print(json.dumps({"waves_per_simd": per_cu, "ms": ms, "wave_instr_per_s": instr / (ms * 1e-3)}))
