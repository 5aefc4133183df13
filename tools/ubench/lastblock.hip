// micro-benchmark (round 6, review item 7): what the "one launch per value-fit minibatch" form would pay for its reduction.
// Today a minibatch is k_vf_grad (128 blocks x 32 samples, each block leaves a partial gradient of the value net's 15 901 parameters = 64 KB) + k_vf_adam (many blocks
// sum the 128 partials in block order and apply Adam: 4.9 us).  In the one-launch form the LAST block to finish (atomic ticket) sums the 8 MB of partials alone, in block
// order, theta and the Adam moments in its LDS.  This measures exactly that read: ONE workgroup of 256 / 1024 threads sums P = 128 partial vectors of N = 15 901 floats in
// order (float4 loads, 8 in flight per thread), against the same sum spread over 64 workgroups (the two-launch form's reduction).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/ubench/lastblock.hip -o tools/ubench/lastblock && tools/ubench/lastblock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N = 15904, P = 128;     // (15 901 padded to a float4 multiple)

template <int THREADS>
__global__ void __launch_bounds__(THREADS) k_sum(const float4* __restrict__ part, float4* __restrict__ out, int n4, int slice) {
  // block b owns float4 columns [b * slice, (b + 1) * slice); every column summed over the P partials IN ORDER (the reproducible order of k_vf_adam)
  const int lo = blockIdx.x * slice, hi = min(n4, lo + slice);
  for (int c = lo + threadIdx.x; c < hi; c += THREADS) {
    float4 acc = {0, 0, 0, 0};
#pragma unroll 8
    for (int p = 0; p < P; p++) { const float4 v = part[(size_t)p * n4 + c]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    out[c] = acc;
  }
}

template <int THREADS> float run(int blocks, const float4* part, float4* out) {
  const int n4 = N / 4, slice = (n4 + blocks - 1) / blocks;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k_sum<THREADS>, dim3(blocks), dim3(THREADS), 0, 0, part, out, n4, slice);
  hipEventRecord(e0);
  const int reps = 50;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_sum<THREADS>, dim3(blocks), dim3(THREADS), 0, 0, part, out, n4, slice);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return 1e3f * ms / reps;
}

int main() {
  float4 *part, *out;
  hipMalloc(&part, (size_t)P * N * sizeof(float)); hipMalloc(&out, N * sizeof(float));
  std::vector<float> h((size_t)P * N, 1.0f);
  hipMemcpy(part, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
  printf("sum of %d partial gradients of %d floats (%.1f MB), in block order; us per reduction (back-to-back launches, so launch gaps are included in every line):\n", P, N, P * N * 4 / 1e6);
  printf("  ONE workgroup of 256 threads  (the last-block form)        %8.1f us\n", run<256>(1, part, out));
  printf("  ONE workgroup of 1024 threads (the last-block form)        %8.1f us\n", run<1024>(1, part, out));
  printf("  4 workgroups of 1024 threads                               %8.1f us\n", run<1024>(4, part, out));
  printf("  64 workgroups of 256 threads  (the two-launch form's shape) %8.1f us\n", run<256>(64, part, out));
  printf("  256 workgroups of 64 threads                               %8.1f us\n", run<64>(256, part, out));
  return 0;
}
