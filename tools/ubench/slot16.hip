// micro-benchmark for the next kernel design: the PGS sweep of the constraint stage with
//   MODE 0  one env per wave, rows on lanes 0..15, the step broadcast by two v_readlane            (today's form, <= 16 rows)
//   MODE 1  FOUR envs per wave, one 16-lane DPP row each, the step broadcast inside every row by ONE v_mov_b64_dpp row_newbcast
// Same arithmetic per env in both (scaled-residual form: delta = max(-f, t); t += A_s[:, i] * delta_i; f += delta at the own row), checked
// against a host loop; prints shader cycles per sweep per ENV at 1 workgroup and at 2 048 (two waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/slot16.hip -o tools/ubench/slot16 && tools/ubench/slot16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

constexpr int NR = 16;   // rows per env

__device__ inline double bcast_wave(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src); hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}
template <int I> __device__ inline double bcast_row(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(I));
  return r;
}
template <int MODE, int I> __device__ inline void row_step(const double* AR, double& t, double& tsave, double nf0, int ln) {
  double delta;
  asm("v_max_f64 %0, %1, %2" : "=v"(delta) : "v"(nf0), "v"(t));
  const double di = MODE == 0 ? bcast_wave(delta, I) : bcast_row<I>(delta);
  if (ln == I) tsave = t;
  t += AR[I] * di;
}
template <int MODE, int I> struct Rows {
  static __device__ inline void run(const double* AR, double& t, double& tsave, double nf0, int ln) {
    row_step<MODE, I>(AR, t, tsave, nf0, ln);
    Rows<MODE, I + 1>::run(AR, t, tsave, nf0, ln);
  }
};
template <int MODE> struct Rows<MODE, NR> { static __device__ inline void run(const double*, double&, double&, double, int) {} };

// A_s: [env][row i][col j] = A[i][j] * (-1 / A[i][i]); t0 = -b_i / A_ii (f = 0 start)
template <int MODE>
__global__ __launch_bounds__(64) void k_pgs(const double* __restrict__ As, const double* __restrict__ t0, double* __restrict__ fout, long long* cyc, int sweeps) {
  const int lane = threadIdx.x, ln = lane & 15;
  const int envs_per_wave = MODE == 0 ? 1 : 4;
  const int env = blockIdx.x * envs_per_wave + (MODE == 0 ? 0 : lane >> 4);
  const bool rowlane = MODE == 0 ? lane < NR : true;
  double AR[NR];
#pragma unroll
  for (int j = 0; j < NR; j++) AR[j] = rowlane ? As[((size_t)env * NR + ln) * NR + j] : 0.0;
  double t = rowlane ? t0[(size_t)env * NR + ln] : 0.0, f = 0.0;
  int lnv = ln; asm volatile("" : "+v"(lnv));
  const long long c0 = __builtin_readcyclecounter();
  for (int s = 0; s < sweeps; s++) {
    const double nf0 = -f;
    double tsave = t;
    Rows<MODE, 0>::run(AR, t, tsave, nf0, MODE == 0 ? (lane < NR ? lnv : -1) : lnv);
    double delta;
    asm("v_max_f64 %0, %1, %2" : "=v"(delta) : "v"(nf0), "v"(tsave));
    f += delta;
  }
  const long long c1 = __builtin_readcyclecounter();
  if (rowlane) fout[(size_t)env * NR + ln] = f;
  if (lane == 0) cyc[blockIdx.x] = c1 - c0;
}

int main() {
  const int nenv = 8192, sweeps = 64;
  std::vector<double> As((size_t)nenv * NR * NR), t0((size_t)nenv * NR), fref((size_t)nenv * NR);
  unsigned long long st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  for (int e = 0; e < nenv; e++) {          // A = G G^T + I (SPD), b random
    double G[NR][NR], A[NR][NR], b[NR];
    for (int i = 0; i < NR; i++) { b[i] = 2.0 * rnd(); for (int j = 0; j < NR; j++) G[i][j] = rnd(); }
    for (int i = 0; i < NR; i++) for (int j = 0; j < NR; j++) { double a = i == j ? 1.0 : 0.0; for (int k = 0; k < NR; k++) a += G[i][k] * G[j][k]; A[i][j] = a; }
    for (int i = 0; i < NR; i++) { for (int j = 0; j < NR; j++) As[((size_t)e * NR + i) * NR + j] = A[i][j] * (-1.0 / A[i][i]); t0[(size_t)e * NR + i] = -b[i] / A[i][i]; }
    // host reference in the same scaled form: t_j = -(A f + b)_j / A_jj
    double f[NR] = {0}, t[NR];
    for (int i = 0; i < NR; i++) t[i] = t0[(size_t)e * NR + i];
    for (int s = 0; s < sweeps; s++)
      for (int i = 0; i < NR; i++) {
        const double delta = std::fmax(-f[i], t[i]);
        f[i] += delta;
        for (int j = 0; j < NR; j++) t[j] = std::fma(As[((size_t)e * NR + j) * NR + i], delta, t[j]);
      }
    for (int i = 0; i < NR; i++) fref[(size_t)e * NR + i] = f[i];
  }
  double *dA, *dt, *df; long long* dc;
  hipMalloc(&dA, As.size() * 8); hipMalloc(&dt, t0.size() * 8); hipMalloc(&df, t0.size() * 8); hipMalloc(&dc, nenv * 8);
  hipMemcpy(dA, As.data(), As.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dt, t0.data(), t0.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> fo(t0.size()); std::vector<long long> cy(nenv);
  for (int mode = 0; mode < 2; mode++) {
    const int epw = mode == 0 ? 1 : 4;
    for (int blocks : {1, 2048}) {
      for (int rep = 0; rep < 2; rep++) {
        if (mode == 0) k_pgs<0><<<blocks, 64>>>(dA, dt, df, dc, sweeps); else k_pgs<1><<<blocks, 64>>>(dA, dt, df, dc, sweeps);
        hipDeviceSynchronize();
      }
      hipMemcpy(fo.data(), df, fo.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(cy.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
      double worst = 0, sum = 0;
      for (size_t i = 0; i < (size_t)blocks * epw * NR; i++) worst = std::fmax(worst, std::fabs(fo[i] - fref[i]) / (1.0 + std::fabs(fref[i])));
      for (int b = 0; b < blocks; b++) sum += (double)cy[b];
      printf("%-34s blocks %5d: %8.1f cycles per sweep per wave, %8.1f per sweep per ENV; max rel error vs host %.1e\n",
             mode == 0 ? "one env / wave, v_readlane x2" : "four envs / wave, row_newbcast", blocks, sum / blocks / sweeps, sum / blocks / sweeps / epw, worst);
    }
  }
  return 0;
}
