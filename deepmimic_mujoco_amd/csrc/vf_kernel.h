// vf_kernel.h — the value fit of the TRPO learner (src/trpo.py:288-296) as three kernels per minibatch instead of ~60 library
// launches.  One minibatch step of the reference is
//     pi.ob_rms.update(mbob);  g = grad of mean((vpred(mbob) - mbret)^2) w.r.t. the value net;  vfadam.update(g, vf_stepsize)
// with the 56-100-100-1 tanh value net of src/mlp_policy_trpo.py:43-48, the obs filter of src/utils/misc_util.py:32-70 and the
// MpiAdam rule of src/mpi_adam.py:21-35.  Here:
//   k_vf_rms   column sums / sums of squares of the minibatch in float64 (fixed reduction order), the LAST block to finish adds
//              them to the filter's state and refreshes its float32 mean / std,
//   k_vf_grad  a block takes 16 samples: normalise + clip, forward, backward, all in LDS (weights staged once per block), and
//              writes its partial gradient of the 15 901 parameters,
//   k_vf_adam  partial gradients summed in a fixed order (four quarters of the blocks, each in block order), Adam moments, step.
// fp32 like the reference's TF graph (sums of the filter in float64 like its numpy arrays).  A whole epoch of minibatches is
// enqueued by one C call (dm_vf_fit_epoch); nothing comes back to the host.  There the filter sums of all minibatches are taken up front
// (k_vf_rms_part / k_vf_rms_scan below): two launches per minibatch remain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmv {

constexpr int OB = 56, H = 100, SB = 16;            // SB: samples per block of k_vf_grad (16: a 4 096-sample minibatch is 256 blocks, one per CU of an MI355X)
constexpr int O_W1 = 0, O_B1 = O_W1 + OB * H, O_W2 = O_B1 + H, O_B2 = O_W2 + H * H, O_W3 = O_B2 + H, O_B3 = O_W3 + H, NP = O_B3 + 1;
constexpr int NPAD = (NP + 63) / 64 * 64;
constexpr int RMS_BLOCKS = 64;

// ---- obs filter ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vf_rms(const float* __restrict__ ob, int bs, double* __restrict__ part /*[RMS_BLOCKS][2*OB]*/,
                                                unsigned* __restrict__ ticket, double* __restrict__ sum, double* __restrict__ sumsq,
                                                double* __restrict__ count, float* __restrict__ mean, float* __restrict__ stdv) {
  __shared__ double red[4][2 * OB];
  __shared__ bool last;
  const int tid = threadIdx.x, col = tid % OB, rg = tid / OB;            // 224 working threads: 4 row groups x 56 columns
  const int rows = (bs + RMS_BLOCKS - 1) / RMS_BLOCKS, r0 = blockIdx.x * rows, r1 = min(bs, r0 + rows);
  if (rg < 4) {
    double s = 0.0, q = 0.0;
    int r = r0 + rg;
    for (; r + 12 < r1; r += 16) {                   // four rows of this row group in flight
      const float x0 = ob[(size_t)r * OB + col], x1 = ob[(size_t)(r + 4) * OB + col], x2 = ob[(size_t)(r + 8) * OB + col], x3 = ob[(size_t)(r + 12) * OB + col];
      s += (double)x0; q += (double)x0 * (double)x0; s += (double)x1; q += (double)x1 * (double)x1;
      s += (double)x2; q += (double)x2 * (double)x2; s += (double)x3; q += (double)x3 * (double)x3;
    }
    for (; r < r1; r += 4) { const double x = (double)ob[(size_t)r * OB + col]; s += x; q += x * x; }
    red[rg][col] = s; red[rg][OB + col] = q;
  }
  __syncthreads();
  if (tid < 2 * OB) part[blockIdx.x * 2 * OB + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  __threadfence();
  __syncthreads();
  if (tid == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (tid < 2 * OB) {
    double a = 0.0;
    for (int b = 0; b < (int)gridDim.x; b += 8) {                               // fixed order: results do not depend on block timing
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; u++) x[u] = b + u < (int)gridDim.x ? part[(b + u) * 2 * OB + tid] : 0.0;
#pragma unroll
      for (int u = 0; u < 8; u++) a += x[u];
    }
    if (tid < OB) sum[tid] += a; else sumsq[tid - OB] += a;
  }
  __syncthreads();
  if (tid == 0) { *count += (double)bs; *ticket = 0u; }
  __syncthreads();
  if (tid < OB) {                                                               // RunningMeanStd._refresh (policy.py)
    const double c = *count;
    const float m = (float)(sum[tid] / c);
    const float var = (float)(sumsq[tid] / c) - m * m;
    mean[tid] = m; stdv[tid] = sqrtf(fmaxf(var, 1e-2f));
  }
}

// ---- forward + backward of 32 samples ------------------------------------------------------------------------------------------
// Activations are kept TRANSPOSED in LDS ([unit][sample]): a thread of the dense layers owns a 2 units x 4 samples register tile (round 3;
// 4 x 4 on 32-sample blocks before: half the blocks, half the CUs, twice the serial work per thread — 36 us per 4 096-sample minibatch) and
// feeds 8 FMAs from a 16-byte and an 8-byte LDS read (four samples of one input, two weights of that input); the weight-gradient products run
// over the sample axis with 16-byte reads as well (4 x 10 tiles).
struct alignas(16) VfShared {
  float W1[OB * H], W2[H * H];
  float z[OB][SB], h1[H][SB], h2[H][SB], d1[H][SB], d2[H][SB];
  float w3[H], b1[H], b2[H], dv[SB];
};
__device__ inline float4 f4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void fma4(float4& a, float w, const float4& x) { a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w; }
__device__ inline float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// out[i0 + a][jj + b] = sum_s A[i0 + a][s] B[jj + b][s]  for a 4 x 10 tile (both operands [.][SB] in LDS), row stride H in `out`
__device__ inline void tile_4x10(const float (*A)[SB], const float (*B)[SB], int i0, int jj, float* out) {
  float t[4][10];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) t[a][b] = 0.0f;
#pragma unroll 2
  for (int s4 = 0; s4 < SB; s4 += 4) {
    float4 av[4], bv[10];
#pragma unroll
    for (int a = 0; a < 4; a++) av[a] = f4(&A[i0 + a][s4]);
#pragma unroll
    for (int b = 0; b < 10; b++) bv[b] = f4(&B[jj + b][s4]);
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 10; b++) t[a][b] += dot4(av[a], bv[b]);
  }
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) out[(i0 + a) * H + jj + b] = t[a][b];
}
__device__ inline float row_sum(const float* r) {            // sum over the SB samples of one unit
  float a = 0.0f;
#pragma unroll
  for (int s4 = 0; s4 < SB; s4 += 4) { const float4 x = f4(r + s4); a += (x.x + x.y) + (x.z + x.w); }
  return a;
}

// the value net's weights into LDS (once per block and minibatch)
__device__ inline void vf_stage_weights(VfShared& S, const float* __restrict__ theta, int tid) {
  {   // weights: 16-byte loads, several in flight (both blocks start 16-byte aligned in the packed layout)
    const float4* g1 = reinterpret_cast<const float4*>(theta + O_W1); float4* l1 = reinterpret_cast<float4*>(S.W1);
    const float4* g2 = reinterpret_cast<const float4*>(theta + O_W2); float4* l2 = reinterpret_cast<float4*>(S.W2);
#pragma unroll 6
    for (int i = tid; i < OB * H / 4; i += 256) l1[i] = g1[i];
#pragma unroll 10
    for (int i = tid; i < H * H / 4; i += 256) l2[i] = g2[i];
  }
  if (tid < H) { S.w3[tid] = theta[O_W3 + tid]; S.b1[tid] = theta[O_B1 + tid]; S.b2[tid] = theta[O_B2 + tid]; }
}
// forward + backward of samples s0 .. s0 + SB - 1 of the minibatch; the tile's partial gradient goes to `out` (NPAD floats)
__device__ inline void vf_grad_tile(VfShared& S, const float* __restrict__ ob, const float* __restrict__ ret, int bs, const float* __restrict__ theta,
                                    const float* __restrict__ mean, const float* __restrict__ stdv, float* __restrict__ out, int s0, int tid) {
#pragma unroll 4
  for (int i = tid; i < SB * OB; i += 256) {                  // coalesced read of [sample][input], transposed store
    const int sm = i / OB, k = i % OB, r = s0 + sm;
    float v = 0.0f;
    if (r < bs) v = fminf(fmaxf((ob[(size_t)r * OB + k] - mean[k]) / stdv[k], -5.0f), 5.0f);
    S.z[k][sm] = v;
  }
  __syncthreads();
  static_assert(SB == 16, "thread tiles below: 4 sample quads x 50 unit pairs");
  const int sq = (tid % 4) * 4, uq = (tid / 4) * 2;           // this thread's 4 samples x 2 units (threads 0..199)
  const bool dense = tid < 200;
  // layer 1, layer 2
  if (dense) {
    float4 acc[2];
#pragma unroll
    for (int u = 0; u < 2; u++) { const float b = S.b1[uq + u]; acc[u] = make_float4(b, b, b, b); }
#pragma unroll 8
    for (int k = 0; k < OB; k++) {
      const float4 x = f4(&S.z[k][sq]); const float2 w = *reinterpret_cast<const float2*>(&S.W1[k * H + uq]);
      fma4(acc[0], w.x, x); fma4(acc[1], w.y, x);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) *reinterpret_cast<float4*>(&S.h1[uq + u][sq]) = make_float4(tanhf(acc[u].x), tanhf(acc[u].y), tanhf(acc[u].z), tanhf(acc[u].w));
  }
  __syncthreads();
  if (dense) {
    float4 acc[2];
#pragma unroll
    for (int u = 0; u < 2; u++) { const float b = S.b2[uq + u]; acc[u] = make_float4(b, b, b, b); }
#pragma unroll 10
    for (int k = 0; k < H; k++) {
      const float4 x = f4(&S.h1[k][sq]); const float2 w = *reinterpret_cast<const float2*>(&S.W2[k * H + uq]);
      fma4(acc[0], w.x, x); fma4(acc[1], w.y, x);
    }
#pragma unroll
    for (int u = 0; u < 2; u++) *reinterpret_cast<float4*>(&S.h2[uq + u][sq]) = make_float4(tanhf(acc[u].x), tanhf(acc[u].y), tanhf(acc[u].z), tanhf(acc[u].w));
  }
  __syncthreads();
  // output, error, d loss / d vpred  (loss = mean over the minibatch of (vpred - ret)^2)
  if (tid < SB) {
    float v = theta[O_B3];
    for (int j = 0; j < H; j++) v += S.h2[j][tid] * S.w3[j];
    S.dv[tid] = (s0 + tid < bs) ? 2.0f * (v - ret[s0 + tid]) / (float)bs : 0.0f;
  }
  __syncthreads();
  // d a2 = dv w3 (1 - h2^2);  dw3, db3
  for (int i = tid; i < SB * H; i += 256) { const int j = i / SB, sm = i % SB; const float h = S.h2[j][sm]; S.d2[j][sm] = S.dv[sm] * S.w3[j] * (1.0f - h * h); }
  if (tid < H) { float a = 0.0f; for (int sm = 0; sm < SB; sm++) a += S.h2[tid][sm] * S.dv[sm]; out[O_W3 + tid] = a; }
  if (tid == H) { float a = 0.0f; for (int sm = 0; sm < SB; sm++) a += S.dv[sm]; out[O_B3] = a; }
  __syncthreads();
  // dW2 = h1^T d2 (250 tiles), db2
  if (tid < 250) tile_4x10(S.h1, S.d2, (tid / 10) * 4, (tid % 10) * 10, out + O_W2);
  if (tid < H) out[O_B2 + tid] = row_sum(S.d2[tid]);
  // d h1 = d2 W2^T, d a1 = d h1 (1 - h1^2): 4 samples x 2 units per thread, four j at a time
  if (dense) {
    float4 acc[2] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
#pragma unroll 5
    for (int j = 0; j < H; j += 4) {
      const float4 d0 = f4(&S.d2[j][sq]), d1 = f4(&S.d2[j + 1][sq]), d2 = f4(&S.d2[j + 2][sq]), d3 = f4(&S.d2[j + 3][sq]);
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const float4 w = f4(&S.W2[(uq + u) * H + j]);
        fma4(acc[u], w.x, d0); fma4(acc[u], w.y, d1); fma4(acc[u], w.z, d2); fma4(acc[u], w.w, d3);
      }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const float4 h = f4(&S.h1[uq + u][sq]);
      *reinterpret_cast<float4*>(&S.d1[uq + u][sq]) = make_float4(acc[u].x * (1.0f - h.x * h.x), acc[u].y * (1.0f - h.y * h.y), acc[u].z * (1.0f - h.z * h.z), acc[u].w * (1.0f - h.w * h.w));
    }
  }
  __syncthreads();
  // dW1 = z^T d1 (140 tiles), db1
  if (tid < 140) tile_4x10(S.z, S.d1, (tid / 10) * 4, (tid % 10) * 10, out + O_W1);
  else if (tid >= 156) out[O_B1 + tid - 156] = row_sum(S.d1[tid - 156]);
}
__global__ __launch_bounds__(256) void k_vf_grad(const float* __restrict__ ob, const float* __restrict__ ret, int bs, const float* __restrict__ theta,
                                                 const float* __restrict__ mean, const float* __restrict__ stdv, float* __restrict__ partial) {
  __shared__ VfShared S;                                      // 93 KB: one block per CU (a gfx950 workgroup may hold up to 160 KB)
  const int tid = threadIdx.x;
  vf_stage_weights(S, theta, tid);
  vf_grad_tile(S, ob, ret, bs, theta, mean, stdv, partial + (size_t)blockIdx.x * NPAD, blockIdx.x * SB, tid);
}

// ---- gradient reduction + MpiAdam step (src/mpi_adam.py:21-35) -----------------------------------------------------------------
// A block takes 64 parameters; its four waves each sum a quarter of the blocks' partials (in block order, sixteen loads in flight), the
// quarters are added in order: a fixed summation tree — results do not depend on timing.
constexpr int ADAM_PARAMS = 64;
__global__ __launch_bounds__(256) void k_vf_adam(const float* __restrict__ partial, int nblk, float* __restrict__ theta, float* __restrict__ m,
                                                 float* __restrict__ v, float a, float beta1, float beta2, float eps) {
  __shared__ float quarter[4][ADAM_PARAMS];
  const int w = threadIdx.x / ADAM_PARAMS, p = blockIdx.x * ADAM_PARAMS + threadIdx.x % ADAM_PARAMS;
  const int per = (nblk + 3) / 4, b0 = w * per, b1 = min(nblk, b0 + per);
  float g = 0.0f;
  if (p < NP) {
    int b = b0;
    for (; b + 16 <= b1; b += 16) {
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; u++) x[u] = partial[(size_t)(b + u) * NPAD + p];
#pragma unroll
      for (int u = 0; u < 16; u++) g += x[u];
    }
    for (; b < b1; b++) g += partial[(size_t)b * NPAD + p];
  }
  quarter[w][threadIdx.x % ADAM_PARAMS] = g;
  __syncthreads();
  if (w != 0 || p >= NP) return;
  g = ((quarter[0][threadIdx.x] + quarter[1][threadIdx.x]) + quarter[2][threadIdx.x]) + quarter[3][threadIdx.x];
  const float mm = beta1 * m[p] + (1.0f - beta1) * g;
  const float vv = beta2 * v[p] + (1.0f - beta2) * g * g;
  m[p] = mm; v[p] = vv;
  theta[p] += (-a) * mm / (sqrtf(vv) + eps);
}

// ---- the obs filter's statistics for a whole epoch up front -------------------------------------------------------------------
// k_vf_rms is a third of a minibatch's time (rocprofv3: 12.9 of 57 us at 4 096 samples) and does not depend on the parameters:
//   k_vf_rms_part   the column sums / sums of squares of EVERY minibatch of the epoch at once (grid RMS_BLOCKS x nb; the same partial sums,
//                   in the same order, as k_vf_rms computes for one),
//   k_vf_rms_scan   one block: minibatch after minibatch it adds the partials (block order) to the filter's state and records the
//                   float32 mean / std the filter holds AFTER that minibatch — what that minibatch's gradient step normalises with.
// Same arithmetic in the same order as a k_vf_rms per minibatch: bit-identical filter state and parameters (tests/test_trpo.py).
// (Measured dead end, round 3: the whole epoch as ONE launch — resident blocks walking the minibatches with a grid barrier between the
//  gradient and the Adam step, exchanged data through device-scope accesses — 67 us per minibatch against 44 us for the two launches: a
//  barrier across 128 CUs on eight XCDs costs more than a kernel boundary here.)
__global__ __launch_bounds__(256) void k_vf_rms_part(const float* __restrict__ ob_all, int bs, double* __restrict__ part_all /*[nb][RMS_BLOCKS][2*OB]*/) {
  __shared__ double red[4][2 * OB];
  const float* ob = ob_all + (size_t)blockIdx.y * bs * OB;
  double* part = part_all + (size_t)blockIdx.y * RMS_BLOCKS * 2 * OB;
  const int tid = threadIdx.x, col = tid % OB, rg = tid / OB;
  const int rows = (bs + RMS_BLOCKS - 1) / RMS_BLOCKS, r0 = blockIdx.x * rows, r1 = min(bs, r0 + rows);
  if (rg < 4) {
    double s = 0.0, q = 0.0;
    int r = r0 + rg;
    for (; r + 12 < r1; r += 16) {
      const float x0 = ob[(size_t)r * OB + col], x1 = ob[(size_t)(r + 4) * OB + col], x2 = ob[(size_t)(r + 8) * OB + col], x3 = ob[(size_t)(r + 12) * OB + col];
      s += (double)x0; q += (double)x0 * (double)x0; s += (double)x1; q += (double)x1 * (double)x1;
      s += (double)x2; q += (double)x2 * (double)x2; s += (double)x3; q += (double)x3 * (double)x3;
    }
    for (; r < r1; r += 4) { const double x = (double)ob[(size_t)r * OB + col]; s += x; q += x * x; }
    red[rg][col] = s; red[rg][OB + col] = q;
  }
  __syncthreads();
  if (tid < 2 * OB) part[blockIdx.x * 2 * OB + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}
__global__ __launch_bounds__(128) void k_vf_rms_scan(const double* __restrict__ part_all, int nb, int bs, double* __restrict__ sum, double* __restrict__ sumsq,
                                                     double* __restrict__ count, float* __restrict__ mean, float* __restrict__ stdv,
                                                     float* __restrict__ means /*[nb][OB]*/, float* __restrict__ stds /*[nb][OB]*/) {
  __shared__ double st[2 * OB];
  const int tid = threadIdx.x;
  if (tid < 2 * OB) st[tid] = tid < OB ? sum[tid] : sumsq[tid - OB];
  double c = *count;
  float m = 0.0f, sd = 1.0f;
  __syncthreads();
  for (int i = 0; i < nb; i++) {
    const double* part = part_all + (size_t)i * RMS_BLOCKS * 2 * OB;
    if (tid < 2 * OB) {
      double a = 0.0;
      for (int b = 0; b < RMS_BLOCKS; b += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = part[(b + u) * 2 * OB + tid];
#pragma unroll
        for (int u = 0; u < 8; u++) a += x[u];
      }
      st[tid] += a;
    }
    c += (double)bs;
    __syncthreads();
    if (tid < OB) {                                                             // RunningMeanStd._refresh (policy.py)
      m = (float)(st[tid] / c);
      const float var = (float)(st[OB + tid] / c) - m * m;
      sd = sqrtf(fmaxf(var, 1e-2f));
      means[(size_t)i * OB + tid] = m; stds[(size_t)i * OB + tid] = sd;
    }
    __syncthreads();
  }
  if (tid < OB) { sum[tid] = st[tid]; mean[tid] = m; stdv[tid] = sd; }
  else if (tid < 2 * OB) sumsq[tid - OB] = st[tid];
  if (tid == 0) *count = c;
}
}  // namespace dmv
