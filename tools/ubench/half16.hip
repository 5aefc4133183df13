// micro-benchmark for the next kernel design: the rows' half solve  y <- L^-T y  (tree-sparse unit factor of
// the 34-dof humanoid, 310 entries; one constraint row per lane, y[34] in registers) with the factor
//   MODE 0  in LDS, one env per wave, broadcast reads at wave-uniform addresses                      (today's form)
//   MODE 1  in LDS, FOUR envs per wave (16-lane rows), one base address per row
//   MODE 2  in REGISTERS, four envs per wave: entry e lives in lane e % 16 of its env's row (register e / 16) and reaches the row through
//           one v_mov_b64_dpp row_newbcast — no LDS at all (what frees 3 KB of LDS per env)
// Checked against a host loop; prints shader cycles per solve per env.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ideepmimic_mujoco_amd/csrc tools/ubench/half16.hip -o half16 && ./half16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "topology.h"
using namespace dmt;
constexpr Topo T = make_topo();
constexpr int NE = 310;          // stored entries (row i: (i,i), (i, parent), ...)
constexpr int NREG = (NE + 15) / 16;

template <int I> __device__ inline double bcast_row(double v) {
  double r;
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(I));
  return r;
}
// x[anc_a(I)] -= L(I, anc_a) * x[I] for rows I = NV-1 .. 1
template <int MODE, int I, int A> struct Anc {
  static __device__ inline void run(double* x, const double* Lsrc, const double* Lreg) {
    if constexpr (A < 14) {
      if constexpr (T.dof_anc[I][A] >= 0) {
        constexpr int e = T.madr[I] + A, j = T.dof_anc[I][A];
        double w;
        if constexpr (MODE == 2) w = bcast_row<e % 16>(Lreg[e / 16]); else w = Lsrc[e];
        x[j] -= w * x[I];
        Anc<MODE, I, A + 1>::run(x, Lsrc, Lreg);
      }
    }
  }
};
template <int MODE, int I> struct Row {
  static __device__ inline void run(double* x, const double* Lsrc, const double* Lreg) {
    Anc<MODE, I, 1>::run(x, Lsrc, Lreg);
    if constexpr (I > 1) Row<MODE, I - 1>::run(x, Lsrc, Lreg);
  }
};

template <int MODE>
__global__ __launch_bounds__(64) void k_half(const double* __restrict__ L, const double* __restrict__ y0, double* __restrict__ yout, long long* cyc, int reps) {
  __shared__ double Ls[4][312];
  const int lane = threadIdx.x, ln = lane & 15, row = lane >> 4;
  const int epw = MODE == 0 ? 1 : 4;
  const int env = blockIdx.x * epw + (MODE == 0 ? 0 : row);
  for (int e = lane; e < NE * epw; e += 64) Ls[e / NE][e % NE] = L[(size_t)(blockIdx.x * epw + e / NE) * NE + e % NE];
  double Lreg[NREG];
#pragma unroll
  for (int r = 0; r < NREG; r++) { const int e = r * 16 + ln; Lreg[r] = (MODE == 2 && e < NE) ? L[(size_t)env * NE + e] : 0.0; }
  __syncthreads();
  double x[NV];
  const int slot = MODE == 0 ? lane : ln;                       // row id inside the env (MODE 0: 64 rows of one env)
#pragma unroll
  for (int d = 0; d < NV; d++) x[d] = y0[((size_t)env * 64 + slot) * NV + d];
  const double* Lsrc = MODE == 0 ? &Ls[0][0] : &Ls[row][0];
  const long long c0 = __builtin_readcyclecounter();
  for (int r = 0; r < reps; r++) {
    Row<MODE, NV - 1>::run(x, Lsrc, Lreg);
#pragma unroll
    for (int d = 0; d < NV; d++) asm volatile("" : "+v"(x[d]));
  }
  const long long c1 = __builtin_readcyclecounter();
#pragma unroll
  for (int d = 0; d < NV; d++) yout[((size_t)env * 64 + slot) * NV + d] = x[d];
  if (lane == 0) cyc[blockIdx.x] = c1 - c0;
}

int main() {
  const int nenv = 8192, reps = 4;
  std::vector<double> L((size_t)nenv * NE), y0((size_t)nenv * 64 * NV), ref(y0.size());
  unsigned long long st = 1234567ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0 - 0.5; };
  for (auto& v : L) v = 0.2 * rnd();
  for (auto& v : y0) v = rnd();
  for (int e = 0; e < nenv; e++)
    for (int s = 0; s < 64; s++) {
      double x[NV];
      for (int d = 0; d < NV; d++) x[d] = y0[((size_t)e * 64 + s) * NV + d];
      for (int r = 0; r < reps; r++)
        for (int i = NV - 1; i >= 1; i--)
          for (int a = 1; a < 14 && T.dof_anc[i][a] >= 0; a++) x[T.dof_anc[i][a]] = std::fma(-L[(size_t)e * NE + T.madr[i] + a], x[i], x[T.dof_anc[i][a]]);
      for (int d = 0; d < NV; d++) ref[((size_t)e * 64 + s) * NV + d] = x[d];
    }
  double *dL, *dy, *dout; long long* dc;
  hipMalloc(&dL, L.size() * 8); hipMalloc(&dy, y0.size() * 8); hipMalloc(&dout, y0.size() * 8); hipMalloc(&dc, nenv * 8);
  hipMemcpy(dL, L.data(), L.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y0.data(), y0.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> out(y0.size()); std::vector<long long> cy(nenv);
  const char* names[] = {"LDS, one env / wave", "LDS, four envs / wave", "registers + row_newbcast, four envs"};
  for (int mode = 0; mode < 3; mode++) {
    const int epw = mode == 0 ? 1 : 4, rows = mode == 0 ? 64 : 16;
    for (int blocks : {1, 2048}) {
      for (int rep = 0; rep < 2; rep++) {
        if (mode == 0) k_half<0><<<blocks, 64>>>(dL, dy, dout, dc, reps); else if (mode == 1) k_half<1><<<blocks, 64>>>(dL, dy, dout, dc, reps); else k_half<2><<<blocks, 64>>>(dL, dy, dout, dc, reps);
        hipDeviceSynchronize();
      }
      hipMemcpy(out.data(), dout, out.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(cy.data(), dc, blocks * 8, hipMemcpyDeviceToHost);
      double worst = 0, sum = 0;
      for (int e = 0; e < blocks * epw; e++) for (int s = 0; s < rows; s++) for (int d = 0; d < NV; d++) {
        const size_t i = ((size_t)e * 64 + s) * NV + d; worst = std::fmax(worst, std::fabs(out[i] - ref[i]) / (1.0 + std::fabs(ref[i])));
      }
      for (int b = 0; b < blocks; b++) sum += (double)cy[b];
      printf("%-38s blocks %5d: %8.1f cycles per solve per wave, %8.1f per ENV; max rel error vs host %.1e\n", names[mode], blocks, sum / blocks / reps, sum / blocks / reps / epw, worst);
    }
  }
  return 0;
}
