// layout probe for v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks per wave: one per 16-lane DPP row?) — for the 16-lane env slots
// of DESIGN.md section 10, where each env's A = Y Y^T could be tiled from 4x4 blocks of its own row.  Prints, per lane, the D value obtained
// from A = "row index i of the lane" and B = "1 for one (k, j)" probes, from which the lane -> (block, i, k) / (block, k, j) / D maps follow.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma4x4.hip -o mfma4x4 && ./mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const double* a, const double* b, double* d) {
  const int l = threadIdx.x;
  double acc = 0.0;
  acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], acc, 0, 0, 0);
  d[l] = acc;
}
int main() {
  double ha[64], hb[64], hd[64], *da, *db, *dd;
  hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 512);
  // probe 1: A[lane] = lane + 1, B = 1 everywhere: D[lane] = sum over k of A(block, i(lane), k)  -> which A lanes feed which D lane
  for (int l = 0; l < 64; l++) { ha[l] = 1.0; hb[l] = 0.0; }
  for (int src = 0; src < 64; src++) {
    for (int l = 0; l < 64; l++) hb[l] = l == src ? 1.0 : 0.0;
    for (int l = 0; l < 64; l++) ha[l] = l + 1.0;
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    k<<<1, 64>>>(da, db, dd); hipMemcpy(hd, dd, 512, hipMemcpyDeviceToHost);
    printf("B lane %2d = 1 -> D nonzero at:", src);
    for (int l = 0; l < 64; l++) if (hd[l] != 0.0) printf(" D[%d]=A[%d]", l, (int)hd[l] - 1);
    printf("\n");
  }
  return 0;
}
