#!/usr/bin/env python3
"""Generate scripts/experiments/mfma_valu_overlap2.hip: the in-wave / cross-wave MFMA-vs-VALU overlap measurement with the
instruction stream PINNED (one `asm volatile` block per loop trip; hipcc neither re-orders, clusters nor SLP-packs it).

VERDICT r5 item 1a.  Per loop trip: 8 x v_mfma_f32_32x32x16_bf16 on four rotating accumulators; behind every MFMA exactly R
filler instructions, each on its own register out of 16 independent chains (a chain is touched again >= 16/R gaps later, so no
filler ever waits for another).  Filler kinds: fma (v_fma_f32), pk (v_pk_fma_f32: R/2 of them = the same flops), mul
(v_mul_f32), exp (v_exp_f32), mad64 (v_mad_u64_u32, Philox's multiply), cvt (v_cvt_pk_bf16_f32), bitop (v_bitop3_b32).
In the D modes the filler waves run A/B times the trips (equal solo durations); D:mfma reads against A, D:valu against B.
--agpr: the MFMA accumulators in the accumulation file (a[..], as in the kernels whose state fills the vector file), and three more
filler kinds: accrd (v_accvgpr_read_b32), accwr (v_accvgpr_write_b32), ds128 (ds_read_b128, one s_waitcnt lgkmcnt(0) per trip).
Modes:  A  MFMAs only      B  the fillers only (same stream with the MFMAs removed)      C  both, one wave per SIMD
        D  two waves per SIMD: waves 0-3 run A, waves 4-7 run B    D1  D with `s_setprio 1` on the filler waves
        D2  D with `s_setprio 1` on the MFMA waves
Time: s_memtime (= shader cycles, MI355X_MICROARCH.md) around the loop of every wave, max over the workgroup's waves, median
over workgroups; reported as cycles per MFMA slot (32 = the matrix pipe's floor).
"""
import sys

R_LIST = [0, 1, 2, 3, 4, 5, 6, 8, 12]
KINDS = ["fma", "pk", "mul", "exp", "mad64", "cvt", "bitop"]
NCH = 16  # independent filler chains (VGPRs; pk uses 8 register pairs)


def filler(kind, j):
    """j-th filler of a trip -> one asm line.  Operands: %6.. = chains (or pairs), c = %4, d = %5."""
    if kind == "pk":
        r = 6 + (j % 8)
        return f"v_pk_fma_f32 %{r}, %{r}, %[c2], %[d2]"
    r = 6 + (j % NCH)
    if kind == "fma":
        return f"v_fma_f32 %{r}, %{r}, %[c], %[d]"
    if kind == "mul":
        return f"v_mul_f32 %{r}, %{r}, %[c]"
    if kind == "exp":
        return f"v_exp_f32 %{r}, %{r}"
    if kind == "mad64":
        # 64-bit result in a register pair: chains are pairs here (8 of them)
        r = 6 + (j % 8)
        return f"v_mad_u64_u32 %{r}, vcc, %[ci], %[di], %{r}"
    if kind == "cvt":
        return f"v_cvt_pk_bf16_f32 %{r}, %{r}, %[c]"
    if kind == "bitop":
        return f"v_bitop3_b32 %{r}, %{r}, %[ci], %[di] bitop3:0x96"
    if kind == "accrd":
        return f"v_accvgpr_read_b32 %{6 + j % 8}, %{14 + j % 8}"
    if kind == "accwr":
        return f"v_accvgpr_write_b32 %{14 + j % 8}, %{6 + j % 8}"
    if kind == "ds128":
        return f"ds_read_b128 %{6 + j % 8}, %[ci] offset:{(j % 8) * 4096}"
    raise ValueError(kind)


ACC_C = "v"  # "a" with --agpr
MFMA = "v_mfma_f32_32x32x16_bf16"  # or v_mfma_f32_16x16x32_bf16 (--m16: accumulators of 4 registers, same operands)


def trip(kind, R, mfma, valu, blocky=False):
    """One trip = 8 MFMA slots.  blocky: 4 trips' worth as 32 MFMAs back to back, then the 32 R fillers (the shape of a kernel
    whose vector phases carry no MFMAs)."""
    lines = []
    j = 0
    nf = (R // 2) if kind == "pk" else R
    if blocky:
        for g in range(32):
            lines.append(f"{MFMA} %{g & 3}, %[a], %[b], %{g & 3}")
        for _ in range(32 * nf):
            lines.append(filler(kind, j))
            j += 1
        return lines
    for g in range(8):
        if mfma:
            lines.append(f"{MFMA} %{g & 3}, %[a], %[b], %{g & 3}")
        if valu:
            for _ in range(nf):
                lines.append(filler(kind, j))
                j += 1
    if kind == "ds128" and valu:
        lines.append("s_waitcnt lgkmcnt(0)")
    return lines


def kernel(kind, R):
    pair = kind in ("pk", "mad64")
    nch = 8 if pair or kind in ("accrd", "accwr", "ds128") else NCH
    chain_t = "v2" if kind == "pk" else ("unsigned long long" if kind == "mad64" else ("unsigned" if kind == "bitop" else "float"))
    name = f"k_{kind}_{R}"
    out = []
    out.append(f"__global__ __launch_bounds__(512) void {name}(long long* ticks, float* sink, int iters, int mode, int iters_v) {{")
    out.append("  acc_t acc0 = (acc_t)(0.0f), acc1 = acc0, acc2 = acc0, acc3 = acc0;")
    out.append("  bf16x8 a, b;")
    out.append("  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }")
    if kind == "pk":
        out.append(f"  v2 ch[{nch}]; for (int i = 0; i < {nch}; ++i) ch[i] = v2{{threadIdx.x * 1e-3f + i, threadIdx.x * 2e-3f + i}};")
    elif kind == "mad64":
        out.append(f"  unsigned long long ch[{nch}]; for (int i = 0; i < {nch}; ++i) ch[i] = threadIdx.x * 77ull + i;")
    elif kind == "bitop":
        out.append(f"  unsigned ch[{nch}]; for (int i = 0; i < {nch}; ++i) ch[i] = threadIdx.x * 77u + i;")
    elif kind == "ds128":
        out.append("  __shared__ float lds_[16384]; if (iters < 0) lds_[threadIdx.x] = 1.0f;")
        out.append(f"  f32x4 ch[{nch}]; for (int i = 0; i < {nch}; ++i) ch[i] = (f32x4)(threadIdx.x * 1e-3f + i);")
    else:
        out.append(f"  float ch[{nch}]; for (int i = 0; i < {nch}; ++i) ch[i] = threadIdx.x * 1e-3f + i;")
    out.append("  const float c = 1.0001f, d = 1e-3f; const v2 c2 = {1.0001f, 1.0002f}, d2 = {1e-3f, 2e-3f};")
    if kind == "ds128":
        out.append("  const unsigned ci = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds_ + (threadIdx.x & 63) * 16u, di = 0x9E3779B9u;")
    else:
        out.append("  const unsigned ci = 0xD2511F53u + threadIdx.x, di = 0x9E3779B9u;")
    if kind in ("accrd", "accwr"):
        out.append("  float ag[8]; for (int i = 0; i < 8; ++i) ag[i] = threadIdx.x * 1e-3f - i;")
    out.append("  const int wave = threadIdx.x >> 6;")
    out.append("  // role: 0 = both (C), 1 = MFMAs only (A), 2 = fillers only (B)")
    out.append("  // modes 6 (E: every wave runs C's stream) and 7 (F: every wave runs the blocky stream) at one or two waves per SIMD")
    out.append("  int role = mode == 0 ? 1 : mode == 1 ? 2 : (mode == 2 || mode == 6) ? 0 : mode == 7 ? 3 : (wave < 4 ? 1 : 2);")
    out.append("  if (mode == 4 && role == 2) __builtin_amdgcn_s_setprio(1);")
    out.append("  if (mode == 5 && role == 1) __builtin_amdgcn_s_setprio(1);")
    out.append("  __syncthreads();")
    out.append("  const long long t0 = __builtin_readcyclecounter();")
    ops = ", ".join([f'"+{ACC_C}"(acc{i})' for i in range(4)])
    chain_ops = ", ".join([f'"+v"(ch[{i}])' for i in range(nch)])
    if kind in ("accrd", "accwr"):
        chain_ops += ", " + ", ".join([f'"+a"(ag[{i}])' for i in range(8)])
    ins = '[a] "v"(a), [b] "v"(b), [c] "v"(c), [d] "v"(d), [c2] "v"(c2), [d2] "v"(d2), [ci] "v"(ci), [di] "v"(di)'
    # operand numbering: %0-3 acc, %4,%5 placeholders so that chains start at %6
    for role, (mf, vl) in ((0, (True, True)), (1, (True, False)), (2, (False, True)), (3, (True, True))):
        body = trip(kind, R, mf, vl, blocky=(role == 3))
        if not body:
            body = ["s_nop 0"]
        asm = "\\n\\t".join(body)
        out.append(f"  if (role == {role}) {{")
        out.append("    for (int it = 0; it < (role == 2 && mode >= 3 ? iters_v : (role == 3 ? iters / 4 : iters)); ++it) {")
        out.append(f'      asm volatile("{asm}"')
        out.append(f"        : {ops}, \"+v\"(dummy0), \"+v\"(dummy1), {chain_ops}")
        out.append(f"        : {ins} : \"vcc\");")
        out.append("    }")
        out.append("  }")
    out.append("  const long long t1 = __builtin_readcyclecounter();")
    out.append("  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;")
    out.append("  float s = acc0[0] + acc1[1] + acc2[2] + acc3[3] + dummy0 + dummy1;")
    if kind in ("accrd", "accwr"):
        out.append("  for (int i = 0; i < 8; ++i) s += ag[i];")
    if kind == "pk":
        out.append(f"  for (int i = 0; i < {nch}; ++i) s += ch[i].x + ch[i].y;")
    elif kind == "ds128":
        out.append(f"  for (int i = 0; i < {nch}; ++i) s += ch[i].x + ch[i].w;")
    else:
        out.append(f"  for (int i = 0; i < {nch}; ++i) s += (float)ch[i];")
    out.append("  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;")
    out.append("}")
    src = "\n".join(out)
    # dummies are declared before use
    src = src.replace("  const float c = 1.0001f", "  float dummy0 = 0.0f, dummy1 = 0.0f;\n  const float c = 1.0001f", 1)
    return name, src


HEADER = r'''// GENERATED by scripts/experiments/gen_overlap2.py -- do not edit.  In-wave and cross-wave MFMA / VALU overlap on gfx950 with the
// instruction stream pinned in inline asm (VERDICT r5 item 1a).  Build + run:
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap2.hip -o /tmp/ovl2 && /tmp/ovl2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef ACC_T acc_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v2 __attribute__((ext_vector_type(2)));
typedef void (*kern_t)(long long*, float*, int, int, int);
struct Row { const char* kind; int R; kern_t k; };
'''

MAIN = r'''
// cycles per MFMA slot (a trip has 8): per workgroup the slowest wave of the role measured, then the median over workgroups
static double run(kern_t k, int mode, int threads, int iters, int role_lo, int role_hi, int iters_v = 0) {
  const int nb = 256;
  long long* ticks; float* sink;
  hipMalloc(&ticks, nb * 8 * sizeof(long long)); hipMalloc(&sink, nb * 512 * sizeof(float));
  hipMemset(ticks, 0, nb * 8 * sizeof(long long));
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k, dim3(nb), dim3(threads), 0, 0, ticks, sink, iters, mode, iters_v);
  hipDeviceSynchronize();
  std::vector<long long> h(nb * 8);
  hipMemcpy(h.data(), ticks, nb * 8 * sizeof(long long), hipMemcpyDeviceToHost);
  std::vector<double> per;
  for (int b = 0; b < nb; ++b) {
    long long m = 0;
    for (int w = role_lo; w < role_hi; ++w) m = std::max(m, h[b * 8 + w]);
    per.push_back((double)m / ((double)(iters_v && role_lo >= 4 ? iters_v : iters) * 8.0));
  }
  std::sort(per.begin(), per.end());
  hipFree(ticks); hipFree(sink);
  return per[per.size() / 2];
}

int main() {
  const int iters = 4000;
  printf("# cycles per MFMA slot (s_memtime ticks / (trips x 8)); 32 = matrix-pipe floor.  hidden = fillers per gap whose issue cost vanished = R - (C - A) / (B / R)\n");
  printf("# E1/E2: every wave runs C's stream, one / two waves per SIMD; F1/F2: every wave runs the BLOCKY stream (32 MFMAs, then their 32 R fillers).  E2, F2: cycles per MFMA slot of ONE wave -- per SIMD the pair does two slots in that time\n");
  printf("%-6s %3s | %7s %7s %7s | %7s | %9s %9s | %9s %9s | %9s %9s | %7s %7s | %7s %7s\n", "kind", "R", "A", "B", "C", "hiddenR", "D:mfma", "D:valu", "D1:mfma", "D1:valu", "D2:mfma", "D2:valu", "E1", "E2", "F1", "F2");
  for (const Row& r : rows) {
    const double A = run(r.k, 0, 256, iters, 0, 4), B = run(r.k, 1, 256, iters, 0, 4), C = run(r.k, 2, 256, iters, 0, 4);
    // the filler waves of D run A / B times the trips, so that both roles are busy for the same time when nothing interferes
    const int iv = r.R && B > 0 ? (int)(iters * A / B) : iters;
    const double Dm = run(r.k, 3, 512, iters, 0, 4, iv), Dv = run(r.k, 3, 512, iters, 4, 8, iv);
    const double D1m = run(r.k, 4, 512, iters, 0, 4, iv), D1v = run(r.k, 4, 512, iters, 4, 8, iv);
    const double D2m = run(r.k, 5, 512, iters, 0, 4, iv), D2v = run(r.k, 5, 512, iters, 4, 8, iv);
    const double per_filler = r.R ? B / r.R : 0.0;
    const double hidden = r.R && per_filler > 0 ? r.R - (C - A) / per_filler : 0.0;
    const double E1 = run(r.k, 6, 256, iters, 0, 4), E2 = run(r.k, 6, 512, iters, 0, 8);
    const double F1 = run(r.k, 7, 256, iters, 0, 4), F2 = run(r.k, 7, 512, iters, 0, 8);
    printf("%-6s %3d | %7.2f %7.2f %7.2f | %7.2f | %9.2f %9.2f | %9.2f %9.2f | %9.2f %9.2f | %7.2f %7.2f | %7.2f %7.2f\n", r.kind, r.R, A, B, C, hidden, Dm, Dv, D1m, D1v, D2m, D2v, E1, E2, F1, F2);
    fflush(stdout);
  }
  return 0;
}
'''


def main():
    global MFMA, ACC_C, KINDS
    if "--agpr" in sys.argv:
        sys.argv.remove("--agpr")
        ACC_C = "a"
        KINDS = ["fma", "accrd", "accwr", "ds128", "cvt"]
    m16 = "--m16" in sys.argv
    if m16:
        sys.argv.remove("--m16")
        MFMA = "v_mfma_f32_16x16x32_bf16"
    out = [HEADER.replace("ACC_T", "f32x4" if m16 else "f32x16")]
    rows = []
    for kind in KINDS:
        for R in R_LIST:
            if kind != "fma" and R in (0, 1, 3, 5, 12) and not (ACC_C == "a" and R in (1, 3)):
                continue
            name, src = kernel(kind, R)
            out.append(src)
            rows.append(f'  {{"{kind}", {R}, {name}}},')
    out.append("static const Row rows[] = {\n" + "\n".join(rows) + "\n};")
    out.append(MAIN)
    path = sys.argv[1] if len(sys.argv) > 1 else "scripts/experiments/mfma_valu_overlap2.hip"
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    main()
