"""Issue cost of packed-f32 VALU instructions against plain v_fma_f32 on this box (ebm_probe_issue_f32)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torchebm_amd import _lib
dev = torch.device("cuda")
blocks, iters = 256 * 8, 4096
out = torch.empty(blocks * 256, device=dev)
st = _lib.stream_handle(dev)
def t(kind, reps=5):
    _lib.call("ebm_probe_issue_f32", out.data_ptr(), blocks, iters, kind, st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): _lib.call("ebm_probe_issue_f32", out.data_ptr(), blocks, iters, kind, st)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
names = {0: "v_fma_f32", 3: "v_pk_fma_f32 (SGPR-pair multiplicand)", 5: "v_pk_fma_f32 (VGPR operands)", 6: "v_pk_mul_f32"}
t0 = t(0)
for k, nm in names.items():
    ms = t(k)
    wave_instr = blocks * 4 * 8 * iters
    print(json.dumps({"kind": k, "instr": nm, "ms": ms, "wave_instr_per_s": wave_instr / ms * 1e3, "units_of_plain": ms / t0,
                      "fma_TFLOPs": wave_instr * 64 * 2 * (2 if k else 1) / ms * 1e3 / 1e12 if k != 6 else None}))
