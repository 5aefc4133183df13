// micro-benchmark: what a lane-predicated region costs (compare, exec save / restore, execz skip) next to the same code run
// unconditionally, one wave.  MI355X: 16 LDS stores 312 cycles unconditional / 380 predicated; 2 stores 56 / 100.
// build: hipcc --offload-arch=gfx950 -O3 -o predicate predicate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int reps, int thr) {
  const int lane = threadIdx.x;
  double a = out[lane], b = out[64 + lane], c = 1.0;
  __shared__ double sh[64 * 16];
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
    if (MODE == 0) {               // block under a lane predicate (>12 instrs -> execz skip branch emitted), predicate true for some lanes
      if (lane < thr) {
#pragma unroll
        for (int j = 0; j < 16; j++) sh[j * 64 + lane] = a + j;
      }
    } else if (MODE == 1) {        // same stores, unconditional
#pragma unroll
      for (int j = 0; j < 16; j++) sh[j * 64 + lane] = a + j;
    } else if (MODE == 2) {        // small predicated block (no skip branch expected)
      if (lane < thr) { sh[lane] = a; sh[64 + lane] = b; }
    } else {                       // small unconditional
      sh[lane] = a; sh[64 + lane] = b;
    }
    a = a * 1.0000001 + c;
    asm volatile("" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  out[lane] = a + sh[lane] + sh[64 * 15 + lane];
  if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
  hipMemset(out, 0, 4096);
  const int reps = 20000;
  long long h;
  for (int pass = 0; pass < 2; pass++) {
    k<0><<<1, 64>>>(out, cyc, reps, 48); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); if (pass) printf("big block predicated   : %.1f ticks/iter\n", (double)h / reps);
    k<1><<<1, 64>>>(out, cyc, reps, 48); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); if (pass) printf("big block unconditional: %.1f ticks/iter\n", (double)h / reps);
    k<2><<<1, 64>>>(out, cyc, reps, 48); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); if (pass) printf("small predicated       : %.1f ticks/iter\n", (double)h / reps);
    k<3><<<1, 64>>>(out, cyc, reps, 48); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); if (pass) printf("small unconditional    : %.1f ticks/iter\n", (double)h / reps);
  }
  return 0;
}
