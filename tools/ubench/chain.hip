// micro-benchmark: dependent-chain latencies of the f64 sequences the PGS row uses (one wave per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ inline double bcast(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src); hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, double dinvr, double a, int reps, int ne_in = 32) {
  const int ne = __builtin_amdgcn_readfirstlane(ne_in);
  double res = threadIdx.x * 1e-3, f0 = 0.5, acc = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int i = 0; i < 32; i++) {
      if (MODE == 0) { res = fma(-dinvr, res, f0); }                                   // 1 dependent fma
      if (MODE == 1) { res = fmax(fma(-dinvr, res, f0), 0.0); }                       // fma + max
      if (MODE == 2) { double d = fmax(fma(-dinvr, res, f0), 0.0) - f0; res = fma(a, d, res); }   // fma max add fma
      if (MODE == 3) { double d = fmax(fma(-dinvr, res, f0), 0.0) - f0; double di = bcast(d, i); res = fma(a, di, res); }  // + readlane
      if (MODE == 4) { double d = fmax(fma(-dinvr, res, f0), 0.0) - f0; double di = bcast(d, i); if (threadIdx.x == i) acc = res; res = fma(a, di, res); }
      if (MODE == 8) { if (i < ne) { double d = fmax(fma(-dinvr, res, f0), 0.0) - f0; double di = bcast(d, i); if (threadIdx.x == i) acc = res; res = fma(a, di, res); } }
      if (MODE == 5) { float x = (float)res; x = fmaf(x, 0.5f, 1.0f); res = x; }      // cvt chain (reference point)
      if (MODE == 6) { res = res * dinvr; }                                            // mul
      if (MODE == 7) { res = res + dinvr; }                                            // add
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = res + acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1 << 16);
  const int reps = 64;
  const char* names[] = {"fma", "fma+max", "fma,max,add,fma", "fma,max,add,readlane,fma", "same + select", "cvt+fmaf+cvt", "mul", "add"};
  for (int blocks : {1, 2048}) {
    for (int m = 0; m < 8; m++) {
      for (int rep = 0; rep < 2; rep++) {
        switch (m) {
          case 0: k<0><<<blocks, 64>>>(out, cyc, 0.3, 0.1, reps); break; case 1: k<1><<<blocks, 64>>>(out, cyc, 0.3, 0.1, reps); break;
          case 2: k<2><<<blocks, 64>>>(out, cyc, 0.3, 0.1, reps); break; case 3: k<3><<<blocks, 64>>>(out, cyc, 0.3, 0.1, reps); break;
          case 4: k<4><<<blocks, 64>>>(out, cyc, 0.3, 0.1, reps); break; case 5: k<5><<<blocks, 64>>>(out, cyc, 0.3, 0.1, reps); break;
          case 6: k<6><<<blocks, 64>>>(out, cyc, 0.3, 0.1, reps); break; case 7: k<7><<<blocks, 64>>>(out, cyc, 0.3, 0.1, reps); break;
        }
        hipDeviceSynchronize();
      }
      std::vector<long long> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : h) s += v;
      printf("blocks %5d  %-28s %8.1f cycles per row-iteration\n", blocks, names[m], s / blocks / (reps * 32.0));
    }
  }
  for (int ne : {32, 6}) {
    for (int rep = 0; rep < 2; rep++) { k<8><<<1, 64>>>(out, cyc, 0.3, 0.1, reps, ne); hipDeviceSynchronize(); }
    long long t; hipMemcpy(&t, cyc, 8, hipMemcpyDeviceToHost);
    printf("per-row scalar guard, ne=%d: %.1f cycles per executed row (%.1f per sweep of 32 slots)\n", ne, (double)t / (reps * (double)ne), (double)t / reps);
  }
  {  // calibrate the s_memtime tick against wall time
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<0><<<1, 64>>>(out, cyc, 0.3, 0.1, 20000); hipDeviceSynchronize();
    hipEventRecord(e0); k<0><<<1, 64>>>(out, cyc, 0.3, 0.1, 20000); hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long t; hipMemcpy(&t, cyc, 8, hipMemcpyDeviceToHost);
    printf("calibration: %lld ticks in %.3f ms -> %.3f ticks/ns (fma chain: %.2f ns per dependent fma)\n", t, ms, t / (ms * 1e6), ms * 1e6 / (20000.0 * 32));
  }
  return 0;
}
