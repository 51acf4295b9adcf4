// Probe: does an LDS-direct load (global_load_lds_dwordx4, destination base in M0) reach LDS addresses beyond 64 KiB on gfx950?
//   hipcc --offload-arch=gfx950 -O2 -o lds_dma_high.exe lds_dma_high.hip && ./lds_dma_high.exe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
__global__ void k(const uint32_t* src, uint32_t* out, uint32_t lds_off) {
  const int lane = threadIdx.x & 63;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + lds_off;
  const uint32_t voff = lane * 16;
  uint32_t keep;
  const uint32_t b = __builtin_amdgcn_readfirstlane(base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
               : "=&s"(keep) : "v"(voff), "s"(src), "s"(b) : "memory");
  __syncthreads();
  const uint32_t* p = reinterpret_cast<const uint32_t*>(smem + lds_off);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = p[lane * 4 + i];
}
int main() {
  std::vector<uint32_t> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 0xabc00000u + i;
  uint32_t *d, *o;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (uint32_t off : {0u, 32768u, 65536u, 98304u, 131072u, 160u * 1024u - 1024u}) {
    hipMemset(o, 0, 1024);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, d, o, off);
    std::vector<uint32_t> r(256);
    hipError_t e = hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[i] != h[i];
    printf("lds offset %6u: %s (%d wrong words) err=%d\n", off, bad ? "MISMATCH" : "ok", bad, (int)e);
  }
  return 0;
}
