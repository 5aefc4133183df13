// micro-benchmark (round 6, review item 8a): does gfx950 skip a 16-lane quarter of a wave64 VALU instruction whose EXEC quarter is zero?
// A wave64 f64 VALU instruction passes through the 16-lane SIMD in four quarter passes; if the hardware elided passes whose EXEC bits are all zero, masking a
// converged environment's DPP row (= one quarter) off would make frozen slots of the packed PGS sweep free.  One wave per SIMD (as k_rollout_packed runs), a
// dependent chain and an independent stream of v_fma_f64 / v_fmac_f64_dpp / v_add_f32 under EXEC = all four quarters, three, two, one; shader cycles per instruction.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/execq.hip -o tools/ubench/execq && tools/ubench/execq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int REPS = 2048;

template <int KIND>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, unsigned long long mask, double seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  float f0 = (float)a0, f1 = f0 + 1;
  unsigned long long saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=s"(saved) : "s"(mask));
  long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
#pragma unroll 1
  for (int r = 0; r < REPS; r++) {
    if constexpr (KIND == 0) {        // dependent chain of 8 v_fma_f64
      asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\t"
                   "v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
    } else if constexpr (KIND == 1) { // 8 independent v_fma_f64
      asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                   "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    } else if constexpr (KIND == 2) { // the PGS row of wave.h: v_max ; v_fma ; s_nop 0 ; v_fmac_f64_dpp   (x 2)
      asm volatile("v_max_f64 %2, %3, %0\n\tv_fma_f64 %1, %4, %0, %1\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %2, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f64 %2, %3, %0\n\tv_fma_f64 %1, %4, %0, %1\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %2, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "=&v"(a2) : "v"(a3), "v"(c), "v"(c));
    } else if constexpr (KIND == 4) { // row variant: v_max ; v_fma(ts) ; v_mov_b64_dpp ; plain v_fma     (x 2)
      asm volatile("v_max_f64 %2, %3, %0\n\tv_fma_f64 %1, %4, %0, %1\n\ts_nop 0\n\tv_mov_b64_dpp %6, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %0, %6, %5, %0\n\t"
                   "v_max_f64 %2, %3, %0\n\tv_fma_f64 %1, %4, %0, %1\n\ts_nop 0\n\tv_mov_b64_dpp %6, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fma_f64 %0, %6, %5, %0"
                   : "+v"(a0), "+v"(a1), "=&v"(a2) : "v"(a3), "v"(c), "v"(c), "v"(a4));
    } else if constexpr (KIND == 6) { // current row without the tsave multiply-add (s_nop 1 in its place)
      asm volatile("v_max_f64 %2, %3, %0\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %2, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                   "v_max_f64 %2, %3, %0\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %2, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "=&v"(a2) : "v"(a3), "v"(c), "v"(c));
    } else if constexpr (KIND == 7) { // dependent pair without DPP: v_max ; v_fma (what two dependent f64 operations cost)
      asm volatile("v_max_f64 %2, %3, %0\n\tv_fma_f64 %0, %2, %5, %0\n\tv_max_f64 %2, %3, %0\n\tv_fma_f64 %0, %2, %5, %0"
                   : "+v"(a0), "+v"(a1), "=&v"(a2) : "v"(a3), "v"(c), "v"(c));
    } else if constexpr (KIND == 8) { // dependent chain of v_fmac_f64_dpp alone (acc += bcast(acc) * a: DPP source = the accumulator just written) with its 2 wait states
      asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\tv_fmac_f64_dpp %0, %0, %5 row_newbcast:7 row_mask:0xf bank_mask:0xf"
                   : "+v"(a0), "+v"(a1), "=&v"(a2) : "v"(a3), "v"(c), "v"(c));
    } else {                          // 8 independent v_add_f32
      asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\t"
                   "v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(f0), "+v"(f1) : "v"(1.0f));
    }
  }
  asm volatile("s_nop 8\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  asm volatile("s_mov_b64 exec, %0" ::"s"(saved));
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> double run(unsigned long long mask, int blocks) {
  double* out; long long* cyc;
  hipMalloc(&out, blocks * 64 * sizeof(double)); hipMalloc(&cyc, blocks * sizeof(long long));
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, out, cyc, mask, 1.0);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += (double)v;
  hipFree(out); hipFree(cyc);
  const int per = (KIND == 2 || KIND >= 4) ? 2 : 8;        // instructions (KIND 2: rows) per repetition
  return s / blocks / REPS / per;
}

int main() {
  const unsigned long long masks[4] = {~0ull, 0x0000ffffffffffffull, 0x00000000ffffffffull, 0x000000000000ffffull};
  const char* names[4] = {"dependent v_fma_f64", "independent v_fma_f64", "PGS row (max, fma, nop, fmac_dpp)", "independent v_add_f32"};
  // s_memtime ticks at a constant 100 MHz on this part; report ratios to the full-EXEC run as well
  printf("one wave per workgroup, 256 workgroups (one wave per CU); s_memtime ticks per instruction (per ROW for the PGS block), and ratio to EXEC = all 64 lanes\n");
  printf("| stream | 4 quarters | 3 quarters | 2 quarters | 1 quarter |\n|---|---|---|---|---|\n");
  for (int kd = 0; kd < 4; kd++) {
    double v[4];
    for (int q = 0; q < 4; q++) v[q] = kd == 0 ? run<0>(masks[q], 256) : kd == 1 ? run<1>(masks[q], 256) : kd == 2 ? run<2>(masks[q], 256) : run<3>(masks[q], 256);
    printf("| %s | %.4f | %.4f (%.2f) | %.4f (%.2f) | %.4f (%.2f) |\n", names[kd], v[0], v[1], v[1] / v[0], v[2], v[2] / v[0], v[3], v[3] / v[0]);
  }
  // round 6: what bounds a PGS row — alternatives for the (max -> broadcast -> multiply-add) chain, full EXEC, ticks per ROW
  printf("\nrow-chain alternatives (full EXEC), ticks per row:\n");
  printf("  current block (v_max, v_fma ts, s_nop 0, v_fmac_f64_dpp)          %.2f\n", run<2>(masks[0], 256));
  printf("  v_max, v_fma ts, s_nop 0, v_mov_b64_dpp, v_fma                    %.2f\n", run<4>(masks[0], 256));
  printf("  v_max, s_nop 1, v_fmac_f64_dpp (no tsave multiply-add)            %.2f\n", run<6>(masks[0], 256));
  printf("  v_max, v_fma without any DPP (two dependent f64 operations)       %.2f\n", run<7>(masks[0], 256));
  printf("  s_nop 1, v_fmac_f64_dpp on its own result (one op + hazard)       %.2f\n", run<8>(masks[0], 256));
  return 0;
}
