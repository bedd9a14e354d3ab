// Micro-benchmark: issue rate of the integer VALU instructions the solid-blend hot path is made of (gfx950).
// Each lane keeps 32 independent registers (the 16 pixels x 2 channel pairs of the raster loop) and applies
// one "blend" (a multiply-add + a byte extraction) to all of them per iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o tools/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(uint32_t* out, int iters, uint32_t K, uint32_t C) {
  uint32_t p[32];
#pragma unroll
  for (int i = 0; i < 32; i++) p[i] = threadIdx.x * 7 + i;
  const uint32_t sel = 0x0c030c01u;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      if (MODE == 0) asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n\tv_perm_b32 %0, 0, %0, %3" : "+v"(p[i]) : "s"(K), "v"(C), "v"(sel));
      if (MODE == 1) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(p[i]) : "s"(K), "v"(C));
      if (MODE == 2) asm volatile("v_perm_b32 %0, 0, %0, %1" : "+v"(p[i]) : "v"(sel));
      if (MODE == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n\tv_pk_lshrrev_b16 %0, 8, %0" : "+v"(p[i]) : "s"(K), "v"(C));
      if (MODE == 4) asm volatile("v_pk_mad_u16 %0, %0, %1, %2\n\tv_pk_lshrrev_b16 %0, 8, %0" : "+v"(p[i]) : "v"(K | (K << 16)), "v"(C));
      if (MODE == 5) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(p[i]) : "v"(K | (K << 16)), "v"(C));
      if (MODE == 6) asm volatile("v_pk_lshrrev_b16 %0, 8, %0" : "+v"(p[i]));
      if (MODE == 7) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(K), "v"(C));
      if (MODE == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n\tv_lshrrev_b32 %0, 8, %0\n\tv_and_b32 %0, 0xff00ff, %0" : "+v"(p[i]) : "s"(K), "v"(C));
      if (MODE == 9) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(p[i]) : "v"(K));
      if (MODE == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(p[i]) : "v"(K));
      if (MODE == 11) asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n\tv_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "+v"(p[i]) : "s"(K), "v"(C));
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 32; i++) s ^= p[i];
  if (s == 0x12345678u) out[threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int n_instr_per_reg, uint32_t* d) {
  const int iters = 2000;
  for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
    const int blocks = 256 * waves_per_simd;      // 256 CUs x (4 waves per block = one per SIMD) x waves_per_simd
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 200u, 0x00ff00ffu);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 200u, 0x00ff00ffu);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    const double wave_instr = double(blocks) * 4 * iters * 32 * n_instr_per_reg;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    const double cyc = (ms * 1e-3 * 2.4e9) * 1024.0 / wave_instr;
    printf("%-34s waves/SIMD %d  %8.3f ms  %7.2f G wave-instr/s  %5.2f cyc/instr/SIMD @2.4GHz\n", name, waves_per_simd, ms, wave_instr / ms / 1e6, cyc);
  }
  return 0;
}

int main() {
  uint32_t* d; CHECK(hipMalloc(&d, 4096));
  run<0>("mad_u32_u24 + perm_b32", 2, d);
  run<1>("mad_u32_u24", 1, d);
  run<2>("perm_b32", 1, d);
  run<3>("mad_u32_u24 + pk_lshrrev_b16", 2, d);
  run<4>("pk_mad_u16 + pk_lshrrev_b16", 2, d);
  run<5>("pk_mad_u16", 1, d);
  run<6>("pk_lshrrev_b16", 1, d);
  run<7>("fma_f32", 1, d);
  run<8>("mad_u24 + lshr + and", 3, d);
  run<9>("mul_u32_u24", 1, d);
  run<10>("add_u32", 1, d);
  run<11>("mad_u24 + mov_sdwa", 2, d);
  return 0;
}
