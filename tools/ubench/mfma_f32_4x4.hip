// layout probe for v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4x4x1 blocks per wave): which lane supplies A(block, i), B(block, j) and which lane /
// register holds D(block, i, j) — for a policy step of FOUR environments (i) x 64 output units (16 blocks x j) per instruction (policy_kernel.h).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_4x4.hip -o build_ab/mfma_f32_4x4 && build_ab/mfma_f32_4x4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int v = 0; v < 4; v++) d[l * 4 + v] = acc[v];
}
int main() {
  float ha[64], hb[64], hd[256], *da, *db, *dd;
  hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
  // A[lane] = lane + 1 everywhere, B = one-hot at lane `src`: D nonzero entries tell (i) which D (lane, reg) a B lane feeds and (ii) which A lane pairs with it
  for (int src = 0; src < 64; src += 1) {
    for (int l = 0; l < 64; l++) { ha[l] = l + 1.0f; hb[l] = l == src ? 1.0f : 0.0f; }
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd); hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    if (src < 8 || src % 16 == 0 || src == 63) {
      printf("B lane %2d = 1 ->", src);
      for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) if (hd[l * 4 + v] != 0.f) printf(" D[lane %d][reg %d]=A[lane %d]", l, v, (int)hd[l * 4 + v] - 1);
      printf("\n");
    }
  }
  return 0;
}
