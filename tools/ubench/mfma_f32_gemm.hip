// layout check of the two fp32 MFMAs the learner kernels use (pg_kernel.h, vf_kernel.h): a wave computes D = A B with the operand / result lane maps
// written out below and the host compares with a plain triple loop.
//   v_mfma_f32_32x32x2_f32: A 32x2, B 2x32:  lane l supplies A[l % 32][l / 32], B[l / 32][l % 32];  D[8 (r / 4) + 4 (l / 32) + r % 4][l % 32] in register r of lane l
//   v_mfma_f32_16x16x4_f32: A 16x4, B 4x16:  lane l supplies A[l % 16][l / 16], B[l / 16][l % 16];  D[4 (l / 16) + r][l % 16] in register r of lane l
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_gemm.hip -o build_ab/mfma_f32_gemm && build_ab/mfma_f32_gemm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int K = 24;
__global__ void k32(const float* A, const float* B, float* D) {          // A [32][K], B [K][32], D [32][32]
  const int l = threadIdx.x, li = l % 32, hf = l / 32;
  v16f acc = {0};
  for (int t = 0; t < K / 2; t++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * K + 2 * t + hf], B[(2 * t + hf) * 32 + li], acc, 0, 0, 0);
  for (int r = 0; r < 16; r++) D[(8 * (r / 4) + 4 * hf + r % 4) * 32 + li] = acc[r];
}
__global__ void k16(const float* A, const float* B, float* D) {          // A [16][K], B [K][16], D [16][16]
  const int l = threadIdx.x, li = l % 16, q = l / 16;
  v4f acc = {0};
  for (int t = 0; t < K / 4; t++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[li * K + 4 * t + q], B[(4 * t + q) * 16 + li], acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[(4 * q + r) * 16 + li] = acc[r];
}
static int check(int n, void (*kern)(const float*, const float*, float*)) {
  float hA[32 * K], hB[K * 32], hD[32 * 32], *dA, *dB, *dD;
  srand(n);
  for (int i = 0; i < n * K; i++) { hA[i] = rand() / (float)RAND_MAX - 0.5f; hB[i] = rand() / (float)RAND_MAX - 0.5f; }
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dD, sizeof hD);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  (void)hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
    double s = 0; for (int k = 0; k < K; k++) s += (double)hA[i * K + k] * hB[k * n + j];
    worst = fmax(worst, fabs(s - hD[i * n + j]));
  }
  printf("mfma %dx%d: max |D - A B| = %.3g  %s\n", n, n, worst, worst < 1e-5 ? "OK" : "LAYOUT MISMATCH");
  return worst < 1e-5 ? 0 : 1;
}
int main() { return check(32, k32) | check(16, k16); }
