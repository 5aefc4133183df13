// micro-benchmark: instruction costs seen by a LONE wave (one wave per SIMD: nothing hides a latency) — what the four-envs-per-wave kernel pays
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lone.hip -o tools/ubench/lone && tools/ubench/lone
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ __launch_bounds__(64) void k(long long* out, double* sink) {
  __shared__ double lds[4096];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) lds[i] = i * 0.5;
  __syncthreads();
  double a = lane * 0.25 + 1.0, b = 1.0000001, c = 0.5, d0 = 1, d1 = 2, d2 = 3, d3 = 4;
  long long t0, t1; int r = 0;
  // 1. dependent v_fma_f64
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
  t1 = __builtin_readcyclecounter(); out[r++] = t1 - t0;
  // 2. independent v_fma_f64 (4 chains)
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c)); }
  t1 = __builtin_readcyclecounter(); out[r++] = t1 - t0;
  // 3. v_fmac_f64_dpp, independent accumulators (4), old source
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(b), "v"(c)); }
  t1 = __builtin_readcyclecounter(); out[r++] = t1 - t0;
  // 4. the PGS row chain: max -> fma -> nop -> fmac_dpp (dependent through t)
  double t = a, ts = 0, nf = -1.0, oh = 0.0;
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { REP16(asm volatile("v_max_f64 %3, %4, %0\n v_fma_f64 %1, %6, %0, %1\n s_nop 0\n v_fmac_f64_dpp %0, %3, %5 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(t), "+v"(ts), "+v"(d3), "=&v"(d2) : "v"(nf), "v"(c), "v"(oh));) }
  t1 = __builtin_readcyclecounter(); out[r++] = t1 - t0;
  // 5. ds_add_f64 (no return), distinct addresses, back to back
  double* p = &lds[lane * 8];
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { REP16(__hip_atomic_fetch_add(p + (i & 7), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);) }
  __builtin_amdgcn_s_waitcnt(0);
  t1 = __builtin_readcyclecounter(); out[r++] = t1 - t0;
  // 6. ds_write_b64 back to back
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { REP16(asm volatile("ds_write_b64 %0, %1" :: "v"((int)(lane * 64 + (i & 7) * 8)), "v"(b) : "memory");) }
  __builtin_amdgcn_s_waitcnt(0);
  t1 = __builtin_readcyclecounter(); out[r++] = t1 - t0;
  // 7. dependent LDS read chain (pointer chase): latency
  int idx = lane;
  for (int i = lane; i < 4096; i += 64) ((int*)lds)[i] = (i + 64) & 4095;
  __syncthreads();
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { REP16(idx = ((volatile int*)lds)[idx];) }
  t1 = __builtin_readcyclecounter(); out[r++] = t1 - t0;
  // 8. fence + wave barrier (dmw::sync) after an LDS write
  t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 64; i++) { REP16(lds[2048 + lane] = b; __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); b += lds[2048 + ((lane + 1) & 63)];) }
  t1 = __builtin_readcyclecounter(); out[r++] = t1 - t0;
  sink[lane] = a + d0 + d1 + d2 + d3 + t + ts + idx + b + lds[lane];
}
int main() {
  long long* d; double* s; hipMalloc(&d, 64 * 8); hipMalloc(&s, 64 * 8);
  for (int rep = 0; rep < 2; rep++) { k<<<1, 64>>>(d, s); hipDeviceSynchronize(); }
  long long h[16]; hipMemcpy(h, d, 16 * 8, hipMemcpyDeviceToHost);
  const char* nm[] = {"dependent v_fma_f64", "independent v_fma_f64 (4 chains)", "v_fmac_f64_dpp, 4 independent accumulators", "PGS row: max, fma, nop, fmac_dpp (chain through t)",
                      "ds_add_f64 no-return, distinct addresses", "ds_write_b64", "dependent LDS read (latency)", "LDS write -> wave fence -> read of a neighbour's word"};
  for (int i = 0; i < 8; i++) printf("%-60s %7.1f cycles each\n", nm[i], h[i] / 1024.0);
  return 0;
}
