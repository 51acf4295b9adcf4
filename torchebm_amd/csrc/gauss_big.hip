// k-fused Langevin chain for the dense Gaussian energy at dims 129 .. 512 (multiples of 4) on the bf16 matrix pipe.
//
//   reference: torchebm/samplers/langevin_dynamics.py:150-185 (the step loop), torchebm/core/base_model.py (GaussianModel:
//   E = 0.5 (x - mu)^T P (x - mu), gradient P (x - mu))
//
// Below 129 the chain state lives in registers for the whole call and the three bf16 splits of Ps are resident in LDS
// (gauss_mfma.hip).  Neither fits here: a row is 0.5 - 2 KB and the splits of Ps are 0.4 - 1.5 MB.  So a step is one pass
// of a tiled GEMM over the state,  g^T = Ps (x - mu)^T,  with the Euler-Maruyama update as its epilogue:
//
//   * a workgroup (4 waves, one per SIMD) owns 128 CTW chains for the whole call; wave w owns CTW chain tiles of 32.
//     The state goes through HBM / L2 once per step (read as the B operand, read again by the epilogue, written once):
//     8 dim bytes per chain-step = one step-equivalent, the same accounting as every other chain kernel.
//   * B operand (x - mu, K x chains): PRIVATE to the wave -- lane (m, h) loads the eight coordinates 16 kb + 8 h .. + 7
//     of chain m as two float4 (a 32 B run of the chain's row), subtracts mu and splits into three bf16x8 in registers.
//   * A operand (Ps, out-rows x K): SHARED by the four waves -- per stage of two K-blocks the workgroup loads the
//     [32 OT] x 32 slab of Ps once (lane-operand units of 8 consecutive fp32 of a row; Ps is symmetric), splits it and
//     writes the three operand-ready images to LDS (double-buffered: one barrier per stage); every wave reads each image
//     with one ds_read_b128 per (tile, K-block).  Ps therefore crosses L2 -> CU once per 128 CTW chains and step.
//   * six products per (out tile, chain tile, K-block) as in gauss_bf16x3.h, smallest first, two independent
//     accumulators alternating; fp32 accumulation.
//   * out-dims beyond 256 are done in two SLICES of <= 8 tiles; the updated first slice waits in registers until the
//     second slice has read the old state (in place, no second state buffer).
//   * epilogue in the C/D layout (lane = chain m, registers = 4 consecutive coordinates per quad): one Philox counter and
//     one float4 of old state per quad, the update in the reference's op order, float4 store.
//
// Registers: 16 OT CTW accumulators per slice (<= 256), one wave per SIMD.  LDS: 2 x 3 x OT x 2 KB of slabs + mu.
#include "ebm_common.h"
#include "gauss_bf16x3.h"
#include "mlp_b16.h"  // EBM_BLOCK_CUT

namespace ebm {
namespace {

using gauss3::bf16x8;
using gauss3::f32x16;
using gauss3::f32x8;
using gauss3::static_for;

constexpr int kBigBlock = 256;

struct BigArgs {
  float* x;
  int64_t n_chains;
  int32_t dim, k_steps;
  float eta, sqrt_eta, noise_coef;
  const float4* table;
  const float* noise;
  int32_t clamp_on;
  float cmin, cmax;
  int32_t thin;
  int64_t n_kept;
  float* traj;
  const float* mean;
  const float* prec;
  RngKey key;
  uint64_t step0;
};

struct Tri {
  bf16x8 h, m, l;
};
__device__ __forceinline__ Tri split8(const f32x8 d) {
  Tri t;
  t.h = __builtin_convertvector(d, bf16x8);
  const f32x8 r1 = d - __builtin_convertvector(t.h, f32x8);
  t.m = __builtin_convertvector(r1, bf16x8);
  const f32x8 r2 = r1 - __builtin_convertvector(t.m, f32x8);
  t.l = __builtin_convertvector(r2, bf16x8);
  return t;
}
// (native vectors throughout: a conditional on HIP's float4 STRUCT is compiled through a stack slot)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x8 join8(const f32x4 a, const f32x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
// *p when ok, zeros otherwise -- the load is unconditional from an always-valid address (the caller's fallback)
__device__ __forceinline__ f32x4 load4_or_zero(const float* p, const float* fallback, bool ok) {
  const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? p : fallback);
  const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
  return ok ? v : z;
}

template <int OT, int NS>
struct BigCfg {
  static constexpr int CTW = NS == 1 ? 2 : 1;            // chain tiles per wave
  static constexpr int CHAINS = 128 * CTW;               // per workgroup
  static constexpr int UNITS = OT * 128;                 // lane-operand units of a slab: [OT][2 K-blocks][64 lanes]
  static constexpr int UPT = (UNITS + kBigBlock - 1) / kBigBlock;
  static constexpr size_t SLAB = (size_t)3 * UNITS * 16;  // bytes of one buffer (three splits)
  static constexpr size_t SMEM = 2 * SLAB + 512 * sizeof(float);
};

template <int OT, int NS>
__global__ __launch_bounds__(kBigBlock) void gauss_big_langevin_kernel(BigArgs a) {
  using C = BigCfg<OT, NS>;
  constexpr int CTW = C::CTW, UNITS = C::UNITS, UPT = C::UPT;
  extern __shared__ __align__(16) unsigned char big_smem[];
  bf16x8* slab = reinterpret_cast<bf16x8*>(big_smem);                   // [2][3][UNITS]
  float* mus = reinterpret_cast<float*>(big_smem + 2 * C::SLAB);        // [d32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
  const int dim = a.dim, d32 = (dim + 31) & ~31, n_stage = d32 >> 5;
  for (int i = tid; i < d32; i += kBigBlock) mus[i] = i < dim ? a.mean[i] : 0.0f;

  int64_t chain[CTW];
  bool active[CTW];
  int64_t xoff[CTW];  // element offset of the chain's row (row 0 for lanes past the last chain: never stored).  An OFFSET, not a
                      // pointer: the per-step launder below would strip a pointer of its address space (flat loads count in lgkmcnt too)
#pragma unroll
  for (int c = 0; c < CTW; ++c) {
    chain[c] = (int64_t)blockIdx.x * C::CHAINS + (wave * CTW + c) * 32 + m;
    active[c] = chain[c] < a.n_chains;
    xoff[c] = (active[c] ? chain[c] : 0) * (int64_t)dim;
  }
  __syncthreads();

  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;
  int until_keep = a.thin;
  int64_t kept = 0;

  // one lane-operand unit of the slab of slice rows `row0`, stage s: eight fp32 of a row of Ps
  auto load_a = [&](int row0, int s, int j, f32x4& v0, f32x4& v1) {
    const int u = tid + kBigBlock * j;
    const int it = u >> 7, kb2 = (u >> 6) & 1, ul = u & 63;
    const int row = row0 + 32 * it + (ul & 31), kcol = 32 * s + 16 * kb2 + 8 * (ul >> 5);
    const bool ok = u < UNITS && row < dim;
    const float* p = a.prec + (int64_t)(ok ? row : 0) * dim + kcol;
    v0 = load4_or_zero(p, a.prec, ok && kcol < dim);
    v1 = load4_or_zero(p + 4, a.prec, ok && kcol + 4 < dim);
  };
  auto store_a = [&](int buf, int j, const f32x4& v0, const f32x4& v1) {
    const int u = tid + kBigBlock * j;
    if (u < UNITS) {
      const Tri t = split8(join8(v0, v1));
      bf16x8* dst = slab + (size_t)buf * 3 * UNITS + u;
      dst[0] = t.h; dst[UNITS] = t.m; dst[2 * UNITS] = t.l;
    }
  };
  auto load_b = [&](int c, int s, int kb2, f32x4& v0, f32x4& v1) {
    const int kcol = 32 * s + 16 * kb2 + 8 * h;
    const float* p = a.x + xoff[c] + kcol;
    v0 = load4_or_zero(p, a.x, active[c] && kcol < dim);
    v1 = load4_or_zero(p + 4, a.x, active[c] && kcol + 4 < dim);
  };

  for (int step = 0; step < a.k_steps; ++step) {
    if (a.table) {
      const float4 tb = a.table[step];
      eta = tb.x; sqrt_eta = tb.y; noise_coef = tb.z;
    }
    const bool keep_now = a.traj && until_keep == 1;
    // (hidden from LICM: left visible, every quad's address of every step is formed before the step loop and spilled)
    int h4 = 4 * h;
    asm volatile("" : "+v"(h4));
    uint64_t e_rows[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      asm volatile("" : "+v"(xoff[c]));
      e_rows[c] = (uint64_t)chain[c] * (uint64_t)dim;
      asm volatile("" : "+v"(e_rows[c]));
    }
    f32x16 res[NS][CTW][OT];  // accumulators of a slice, then its updated state

    static_for<NS>([&](auto slc) {
      constexpr int sl = decltype(slc)::value;
      constexpr int row0 = sl * 32 * OT;
      static_for<CTW * OT>([&](auto ic) {
        constexpr int c = decltype(ic)::value / OT, ot = decltype(ic)::value % OT;
#pragma unroll
        for (int r = 0; r < 16; ++r) res[sl][c][ot][r] = 0.0f;
      });

      // ---- stage 0 of the slice: its slab and this wave's B operands
      f32x4 ra[UPT][2], rb[CTW][2][2];
      static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; load_a(row0, 0, j, ra[j][0], ra[j][1]); });
      static_for<CTW * 2>([&](auto ic) {
        constexpr int c = decltype(ic)::value >> 1, kb2 = decltype(ic)::value & 1;
        load_b(c, 0, kb2, rb[c][kb2][0], rb[c][kb2][1]);
      });
      static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; store_a(0, j, ra[j][0], ra[j][1]); });
      __syncthreads();

      for (int s = 0; s < n_stage; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < n_stage;
        f32x4 rbn[CTW][2][2];
        if (more) {  // the next stage's operands: a whole stage of matrix work for them to land
          static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; load_a(row0, s + 1, j, ra[j][0], ra[j][1]); });
          static_for<CTW * 2>([&](auto ic) {
            constexpr int c = decltype(ic)::value >> 1, kb2 = decltype(ic)::value & 1;
            load_b(c, s + 1, kb2, rbn[c][kb2][0], rbn[c][kb2][1]);
          });
        }
        const bf16x8* sb = slab + (size_t)buf * 3 * UNITS + lane;
        static_for<2>([&](auto kc) {
          constexpr int kb2 = decltype(kc)::value;
          Tri b[CTW];
          const f32x4 m0 = *reinterpret_cast<const f32x4*>(mus + 32 * s + 16 * kb2 + 8 * h);
          const f32x4 m1 = *reinterpret_cast<const f32x4*>(mus + 32 * s + 16 * kb2 + 8 * h + 4);
          static_for<CTW>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            b[c] = split8(join8(rb[c][kb2][0], rb[c][kb2][1]) - join8(m0, m1));
          });
          // pairs of independent accumulators alternate: (two chain tiles, one A triple) or (one chain tile, two out tiles)
          constexpr int PAIRS = CTW == 2 ? OT : (OT + 1) / 2;
          // the A triples of pair p + 1 are requested before the twelve MFMAs of pair p (fenced: left alone, the scheduler
          // hoists every ds_read of the stage to its top -- 6 OT operand registers per K-block)
          auto read_a = [&](auto pc, bf16x8 (&a6)[6]) {
            constexpr int pi = decltype(pc)::value;
            constexpr int ot0 = CTW == 2 ? pi : 2 * pi, ot1 = CTW == 2 ? pi : (2 * pi + 1 < OT ? 2 * pi + 1 : 2 * pi);
            a6[0] = sb[2 * UNITS + ot0 * 128 + kb2 * 64]; a6[1] = sb[UNITS + ot0 * 128 + kb2 * 64]; a6[2] = sb[ot0 * 128 + kb2 * 64];
            if constexpr (CTW == 1 && ot1 != ot0) {
              a6[3] = sb[2 * UNITS + ot1 * 128 + kb2 * 64]; a6[4] = sb[UNITS + ot1 * 128 + kb2 * 64]; a6[5] = sb[ot1 * 128 + kb2 * 64];
            } else {
              a6[3] = a6[0]; a6[4] = a6[1]; a6[5] = a6[2];
            }
          };
          bf16x8 acur[6];
          read_a(std::integral_constant<int, 0>{}, acur);
          static_for<PAIRS>([&](auto pc) {
            constexpr int pi = decltype(pc)::value;
            constexpr int ot0 = CTW == 2 ? pi : 2 * pi, ot1 = CTW == 2 ? pi : (2 * pi + 1 < OT ? 2 * pi + 1 : 2 * pi);
            constexpr int c1 = CTW == 2 ? 1 : 0;
            constexpr bool two = CTW == 2 || 2 * pi + 1 < OT;
            bf16x8 anext[6];
            if constexpr (pi + 1 < PAIRS) read_a(std::integral_constant<int, pi + 1>{}, anext);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 g0 = res[sl][0][ot0], g1;
            if constexpr (two) g1 = res[sl][c1][ot1];
            const Tri& b0 = b[0];
            const Tri& b1 = b[c1];
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[0], b0.h, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[3], b1.h, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[1], b0.m, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[4], b1.m, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[1], b0.h, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[4], b1.h, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[2], b0.l, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[5], b1.l, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[2], b0.m, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[5], b1.m, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[2], b0.h, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[5], b1.h, g1, 0, 0, 0);
            res[sl][0][ot0] = g0;
            if constexpr (two) res[sl][c1][ot1] = g1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (pi + 1 < PAIRS) {
#pragma unroll
              for (int i = 0; i < 6; ++i) acur[i] = anext[i];
            }
          });
        });
        if (more) {
          static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; store_a(buf ^ 1, j, ra[j][0], ra[j][1]); });
          static_for<CTW * 2>([&](auto ic) {
            constexpr int c = decltype(ic)::value >> 1, kb2 = decltype(ic)::value & 1;
            rb[c][kb2][0] = rbn[c][kb2][0];
            rb[c][kb2][1] = rbn[c][kb2][1];
          });
        }
        __syncthreads();  // the next slab is written, this one is read by everyone
      }

      // ---- Euler-Maruyama update of the slice in the reference's op order (one Philox counter per register quad)
      static_for<CTW * OT>([&](auto ic) {
        constexpr int c = decltype(ic)::value / OT, ot = decltype(ic)::value % OT;
        const uint64_t e_row = e_rows[c];
        static_for<4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const int d0 = row0 + 32 * ot + 8 * q + h4;
          const bool ok = active[c] && d0 < dim;
          int off = ok ? d0 : 0;
          asm volatile("" : "+v"(off));  // pins the quad's loads here (volatile asm keeps its order: the block cuts, the next quad)
          const f32x4 xo = *reinterpret_cast<const f32x4*>(a.x + xoff[c] + off);  // (lanes that are not ok: some valid word, never stored)
          f32x4 eps;
          if (a.noise) {
            eps = *reinterpret_cast<const f32x4*>(a.noise + (int64_t)step * a.n_chains * dim + (active[c] ? (int64_t)e_row : 0) + off);
          } else {
            const F4 n4 = normal4_at(a.key, (e_row + (uint64_t)d0) >> 2, a.step0 + (uint64_t)step);
            eps = f32x4{n4.v[0], n4.v[1], n4.v[2], n4.v[3]};
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float x1 = xo[i] - eta * res[sl][c][ot][4 * q + i];
            const float dw = eps[i] * sqrt_eta;
            float nv = x1 + noise_coef * dw;
            if (a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
            res[sl][c][ot][4 * q + i] = nv;
          }
          if constexpr (sl == NS - 1) {  // nobody reads the old state after the last slice's K loop: store at once
            if (ok) {
              const f32x4 v = {res[sl][c][ot][4 * q], res[sl][c][ot][4 * q + 1], res[sl][c][ot][4 * q + 2], res[sl][c][ot][4 * q + 3]};
              *reinterpret_cast<f32x4*>(a.x + xoff[c] + d0) = v;
              if (keep_now) *reinterpret_cast<f32x4*>(a.traj + ((int64_t)e_row * a.n_kept + kept * (int64_t)dim) + d0) = v;
            }
          }
          __builtin_amdgcn_sched_barrier(0);  // one Philox call's temporaries at a time
        });
        EBM_BLOCK_CUT();  // one tile per basic block: the scheduler does not stretch 64 Philox calls over each other
      });
    });

    // ---- the held slices: every slice has read the old state by now
    if constexpr (NS > 1) {
      int h4s = 4 * h;
      asm volatile("" : "+v"(h4s));  // (fresh conditions: shared with the epilogue's, 2 SGPRs per quad stay live across a slice)
      static_for<(NS - 1) * CTW * OT>([&](auto ic) {
        constexpr int sl = decltype(ic)::value / (CTW * OT), c = (decltype(ic)::value / OT) % CTW, ot = decltype(ic)::value % OT;
        constexpr int row0 = sl * 32 * OT;
        static_for<4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const int d0 = row0 + 32 * ot + 8 * q + h4s;
          if (active[c] && d0 < dim) {
            const f32x4 v = {res[sl][c][ot][4 * q], res[sl][c][ot][4 * q + 1], res[sl][c][ot][4 * q + 2], res[sl][c][ot][4 * q + 3]};
            *reinterpret_cast<f32x4*>(a.x + xoff[c] + d0) = v;
            if (keep_now) *reinterpret_cast<f32x4*>(a.traj + ((int64_t)e_rows[c] * a.n_kept + kept * (int64_t)dim) + d0) = v;
          }
        });
      });
    }
    if (--until_keep == 0) {
      until_keep = a.thin;
      ++kept;
    }
    // (a wave reads only its own chains' rows, as B operands and in its epilogue: its stores above are complete before
    //  its loads of the next step -- no workgroup barrier is needed for x)
    __builtin_amdgcn_s_waitcnt(0);
  }
}

template <int OT, int NS>
int launch_big(const BigArgs& a, hipStream_t st) {
  using C = BigCfg<OT, NS>;
  static DeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_big_langevin_kernel<OT, NS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
  const int64_t blocks = ceil_div64(a.n_chains, C::CHAINS);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  hipLaunchKernelGGL((gauss_big_langevin_kernel<OT, NS>), dim3((unsigned)blocks), dim3(kBigBlock), C::SMEM, st, a);
  return check_launch("ebm_langevin_chain_f32");
}

}  // namespace

bool gauss_big_supported(int32_t dim) { return dim > 128 && dim <= 512 && (dim % 4) == 0; }

int launch_langevin_chain_gauss_big(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                                    float eta, float sqrt_eta, float noise_coef, const float* coef_table,
                                    int clamp_on, float cmin, float cmax, int32_t thin, float* traj,
                                    const float* noise, uint64_t seed, uint64_t offset, hipStream_t st) {
  if (!gauss_big_supported(dim)) return fail(EBM_EDIM, "ebm_langevin_chain_f32: the tiled Gaussian kernel takes dims 132 .. 512 in steps of 4, not %d", dim);
  BigArgs a;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.noise = noise; a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj;
  a.mean = e.dev0; a.prec = e.dev1;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  const int tiles = (dim + 31) / 32;  // 5 .. 16
  if (tiles <= 8) {
    switch (tiles) {
      case 5: return launch_big<5, 1>(a, st);
      case 6: return launch_big<6, 1>(a, st);
      case 7: return launch_big<7, 1>(a, st);
      default: return launch_big<8, 1>(a, st);
    }
  }
  switch ((tiles + 1) / 2) {  // two slices of ceil(tiles / 2) out tiles (the second one may end in a zero tile)
    case 5: return launch_big<5, 2>(a, st);
    case 6: return launch_big<6, 2>(a, st);
    case 7: return launch_big<7, 2>(a, st);
    default: return launch_big<8, 2>(a, st);
  }
}

}  // namespace ebm
