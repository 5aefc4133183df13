// vf_kernel.h — the value fit of the TRPO learner (src/trpo.py:288-296) as three kernels per minibatch instead of ~60 library
// launches.  One minibatch step of the reference is
//     pi.ob_rms.update(mbob);  g = grad of mean((vpred(mbob) - mbret)^2) w.r.t. the value net;  vfadam.update(g, vf_stepsize)
// with the 56-100-100-1 tanh value net of src/mlp_policy_trpo.py:43-48, the obs filter of src/utils/misc_util.py:32-70 and the
// MpiAdam rule of src/mpi_adam.py:21-35.  Here:
//   k_vf_rms   column sums / sums of squares of the minibatch in float64 (fixed reduction order), the LAST block to finish adds
//              them to the filter's state and refreshes its float32 mean / std,
//   k_vf_grad  a block takes 32 samples: normalise + clip, forward, backward as fp32 MFMA tiles out of LDS (theta staged once per block), and
//              writes its partial gradient of the 15 901 parameters,
//   k_vf_adam  partial gradients summed in a fixed order (four quarters of the blocks, each in block order), Adam moments, step.
// fp32 like the reference's TF graph (sums of the filter in float64 like its numpy arrays).  A whole epoch of minibatches is
// enqueued by one C call (dm_vf_fit_epoch); nothing comes back to the host.  There the filter sums of all minibatches are taken up front
// (k_vf_rms_part / k_vf_rms_scan below): two launches per minibatch remain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmv {

constexpr int OB = 56, H = 100, SB = 32;            // SB: samples per block of k_vf_grad (a 4 096-sample minibatch is 128 blocks: the policy step can have the other CUs)
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
  const int rows = (bs + (int)gridDim.x - 1) / (int)gridDim.x, r0 = blockIdx.x * rows, r1 = min(bs, r0 + rows);
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
// Every product runs on the matrix cores in fp32 (v_mfma_f32_32x32x2_f32), laid out like the policy kernels (pg_kernel.h): activations
// transposed in LDS ([unit][sample], row stride 33 floats: conflict-free as the B operand of a layer and as an operand of the weight-gradient
// products, which sum over the samples), ONE copy of theta in LDS whose order (W1, b1, W2, b2, w3, b3) makes each bias the row after its
// matrix — a constant row of ones under z / h1 makes the biases part of the products, forward and backward, and the bias gradients rows 56 /
// 100 of the weight-gradient tiles.  Wave w owns hidden units 32 w .. 32 w + 31 (100 padded to 128: rows past 99 read finite junk and are
// never stored).  (Rounds 2-3: 4 x 4 / 2 x 4 register tiles of FMAs on 16-sample blocks, 28 us per 4 096-sample minibatch.)
constexpr int SBP = SB + 1, ZR = OB + 2, HR = H + 4;
constexpr int NWT = (NP + 3) / 4 * 4;
typedef float v16f __attribute__((ext_vector_type(16)));
struct alignas(16) VfShared {
  float Wt[NWT];                                      // theta: W1 [56][100], b1, W2 [100][100], b2, w3 [100], b3
  float z[ZR][SBP];                                   // row 56 = 1, row 57 = 0
  float h1[HR][SBP], h2[HR][SBP];                     // h1: row 100 = 1, row 101 = 0
  float d2[HR][SBP], d1[HR][SBP];                     // (as operands of the weight-gradient products their 128-row tiles read on into what follows)
  float vpart[8][SB], dv[SB];
  float tail[24 * SBP];                               // ... zeros
};
static_assert(SB * OB % 256 == 0, "the observation tile is read in whole rounds of the block");
static_assert(sizeof(VfShared) <= 160 * 1024, "VfShared must fit a CU's LDS");
static_assert(O_W2 + 127 * H + H <= NWT + (ZR + HR) * SBP, "padded W2 rows (A operand of the backward product) read into z / h1");
__device__ inline v16f mfma32(float a, float b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ inline int row32(int r, int hf) { return 8 * (r / 4) + 4 * hf + (r % 4); }      // row of a 32x32 result tile in register r of the lanes of half hf
// tanh(x) = sign(x) (1 - t) / (1 + t), t = exp(-2 |x|), on the hardware exponential / reciprocal (absolute error ~1e-7; pg_kernel.h)
__device__ inline float fast_tanh(float x) {
  const float t = __expf(-2.0f * fabsf(x));
  return copysignf((1.0f - t) * __frcp_rn(1.0f + t), x);
}

// forward + backward of samples s0 .. s0 + SB - 1 of the minibatch; the tile's partial gradient goes to `out` (NPAD floats, theta order)
__global__ __launch_bounds__(256) void k_vf_grad(const float* __restrict__ ob, const float* __restrict__ ret, int bs, const float* __restrict__ theta,
                                                 const float* __restrict__ mean, const float* __restrict__ stdv, float* __restrict__ partial) {
  __shared__ VfShared S;                                      // 126 KB: one block per CU
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, li = l & 31, hf = l >> 5, u0 = 32 * w;
  const int s0 = blockIdx.x * SB;
  float* out = partial + (size_t)blockIdx.x * NPAD;
  {
    // One round trip for everything the block reads: the tile's observations (with the filter's mean / std) and theta are requested together,
    // the pads are zeroed while they are in flight (operands that reach into pad rows / columns must be finite).
    constexpr int NZ = SB * OB / 256, NT = (NP / 4 + 255) / 256;
    float x[NZ], mu[NZ], sd[NZ];
#pragma unroll
    for (int j = 0; j < NZ; j++) {
      const int i = tid + 256 * j, sm = i / OB, k = i % OB, r = s0 + sm;
      x[j] = r < bs ? ob[(size_t)r * OB + k] : 0.0f; mu[j] = mean[k]; sd[j] = stdv[k];
    }
    const float4* g = reinterpret_cast<const float4*>(theta);
    float4 th[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) { const int i = tid + 256 * j; th[j] = i < NP / 4 ? g[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
    const float last = tid < NWT - NP / 4 * 4 && NP / 4 * 4 + tid < NP ? theta[NP / 4 * 4 + tid] : 0.0f;
    float4* act = reinterpret_cast<float4*>(&S.z[0][0]);
    for (int i = tid; i < (int)((sizeof(VfShared) - sizeof(S.Wt)) / 16); i += 256) act[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    float4* d = reinterpret_cast<float4*>(S.Wt);
#pragma unroll
    for (int j = 0; j < NT; j++) { const int i = tid + 256 * j; if (i < NP / 4) d[i] = th[j]; }
    if (tid < NWT - NP / 4 * 4) S.Wt[NP / 4 * 4 + tid] = last;
    if (tid < SB) { S.z[OB][tid] = 1.0f; S.h1[H][tid] = 1.0f; }
#pragma unroll
    for (int j = 0; j < NZ; j++) {                            // coalesced read of [sample][input], transposed store
      const int i = tid + 256 * j, sm = i / OB, k = i % OB;
      S.z[k][sm] = (s0 + sm < bs) ? fminf(fmaxf((x[j] - mu[j]) / sd[j], -5.0f), 5.0f) : 0.0f;
    }
  }
  __syncthreads();
  {   // layer 1: h1 = tanh(W1ext^T zext)
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < (OB + 2) / 2; t++) { const int k = 2 * t + hf; acc = mfma32(S.Wt[O_W1 + k * H + u0 + li], S.z[k][li], acc); }
#pragma unroll
    for (int r = 0; r < 16; r++) { const int u = u0 + row32(r, hf); if (u < H) S.h1[u][li] = fast_tanh(acc[r]); }
  }
  __syncthreads();
  {   // layer 2
    v16f acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
    for (int t = 0; t < (H + 2) / 2; t++) { const int k = 2 * t + hf; acc = mfma32(S.Wt[O_W2 + k * H + u0 + li], S.h1[k][li], acc); }
#pragma unroll
    for (int r = 0; r < 16; r++) { const int u = u0 + row32(r, hf); if (u < H) S.h2[u][li] = fast_tanh(acc[r]); }
  }
  __syncthreads();
  {   // output: vpred = w3 . h2 + b3, eight partial sums per sample
    const int sm = tid % SB, part = tid / SB;
    float v = 0.0f;
    for (int j = part; j < H; j += 8) v += S.h2[j][sm] * S.Wt[O_W3 + j];
    S.vpart[part][sm] = v;
  }
  __syncthreads();
  if (tid < SB) {                                             // error, d loss / d vpred  (loss = mean over the minibatch of (vpred - ret)^2)
    float v = S.Wt[O_B3];
#pragma unroll
    for (int p = 0; p < 8; p++) v += S.vpart[p][tid];
    S.dv[tid] = (s0 + tid < bs) ? 2.0f * (v - ret[s0 + tid]) / (float)bs : 0.0f;
  }
  __syncthreads();
  // delta2 = dv w3 (1 - h2^2);  dw3, db3
  for (int i = tid; i < SB * H; i += 256) { const int j = i / SB, sm = i % SB; const float h = S.h2[j][sm]; S.d2[j][sm] = S.dv[sm] * S.Wt[O_W3 + j] * (1.0f - h * h); }
  if (tid < H) { float a = 0.0f; for (int sm = 0; sm < SB; sm++) a += S.h2[tid][sm] * S.dv[sm]; out[O_W3 + tid] = a; }
  if (tid == H) { float a = 0.0f; for (int sm = 0; sm < SB; sm++) a += S.dv[sm]; out[O_B3] = a; }
  __syncthreads();
  const int col = u0 + li;
  {
    v16f g2[4], acc;
#pragma unroll
    for (int r = 0; r < 16; r++) { g2[0][r] = 0.0f; g2[1][r] = 0.0f; g2[2][r] = 0.0f; g2[3][r] = 0.0f; acc[r] = 0.0f; }
#pragma unroll
    for (int t = 0; t < SB / 2; t++) {                        // dW2ext = h1ext delta2^T  (row 100: db2)
      const float b = S.d2[u0 + li][2 * t + hf];
#pragma unroll
      for (int mt = 0; mt < 4; mt++) g2[mt] = mfma32(S.h1[32 * mt + li][2 * t + hf], b, g2[mt]);
    }
#pragma unroll
    for (int t = 0; t < H / 2; t++) acc = mfma32(S.Wt[O_W2 + (u0 + li) * H + 2 * t + hf], S.d2[2 * t + hf][li], acc);    // W2 delta2
#pragma unroll
    for (int r = 0; r < 16; r++) {                            // delta1 = (W2 delta2) (1 - h1^2)
      const int u = u0 + row32(r, hf);
      if (u < H) { const float h = S.h1[u][li]; S.d1[u][li] = acc[r] * (1.0f - h * h); }
    }
    if (col < H) {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int mt = 0; mt < 4; mt++) { const int i = 32 * mt + row32(r, hf); if (i <= H) out[O_W2 + i * H + col] = g2[mt][r]; }
    }
  }
  __syncthreads();
  {
    v16f g1[2];
#pragma unroll
    for (int r = 0; r < 16; r++) { g1[0][r] = 0.0f; g1[1][r] = 0.0f; }
#pragma unroll
    for (int t = 0; t < SB / 2; t++) {                        // dW1ext = zext delta1^T  (row 56: db1)
      const float b = S.d1[u0 + li][2 * t + hf];
#pragma unroll
      for (int mt = 0; mt < 2; mt++) g1[mt] = mfma32(S.z[32 * mt + li][2 * t + hf], b, g1[mt]);
    }
    if (col < H) {
#pragma unroll
      for (int r = 0; r < 16; r++)
#pragma unroll
        for (int mt = 0; mt < 2; mt++) { const int i = 32 * mt + row32(r, hf); if (i <= OB) out[O_W1 + i * H + col] = g1[mt][r]; }
    }
  }
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
// Two launches: k_vf_rms_fold adds every (minibatch, column) pair's 64 partials up in block order, one block per minibatch (the sums replace the
// first partial in place); k_vf_rms_scan is one wave of 56 threads that walks the minibatches, each carrying its column's sum AND sum of squares:
// no exchange between threads, eight minibatches' sums in flight.  (Round 3's form added each minibatch's partials inside the walk, two block
// barriers per minibatch: 193 us per epoch of 128 minibatches on the fit's critical path; folding inside the scan's own block — one CU pulling
// 7 MB through its L2 port — still 123 us.)
__global__ __launch_bounds__(128) void k_vf_rms_fold(double* __restrict__ part_all) {
  const int c = threadIdx.x;
  if (c >= 2 * OB) return;
  double* part = part_all + (size_t)blockIdx.x * RMS_BLOCKS * 2 * OB;
  double a = 0.0;
  for (int b = 0; b < RMS_BLOCKS; b += 8) {
    double x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = part[(b + u) * 2 * OB + c];
#pragma unroll
    for (int u = 0; u < 8; u++) a += x[u];
  }
  part[c] = a;                                                                  // (this thread is the only reader of column c of this minibatch)
}
__global__ __launch_bounds__(64) void k_vf_rms_scan(const double* __restrict__ part_all, int nb, int bs, double* __restrict__ sum, double* __restrict__ sumsq,
                                                    double* __restrict__ count, float* __restrict__ mean, float* __restrict__ stdv,
                                                    float* __restrict__ means /*[nb][OB]*/, float* __restrict__ stds /*[nb][OB]*/) {
  const int tid = threadIdx.x;
  if (tid >= OB) return;
  double s = sum[tid], q = sumsq[tid], c = *count;
  float m = 0.0f, sd = 1.0f;
  for (int i0 = 0; i0 < nb; i0 += 8) {
    double as[8], aq[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = i0 + u < nb ? i0 + u : nb - 1;
      as[u] = part_all[(size_t)i * RMS_BLOCKS * 2 * OB + tid]; aq[u] = part_all[(size_t)i * RMS_BLOCKS * 2 * OB + OB + tid];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = i0 + u;
      if (i >= nb) break;
      s += as[u]; q += aq[u]; c += (double)bs;
      m = (float)(s / c);                                                       // RunningMeanStd._refresh (policy.py)
      const float var = (float)(q / c) - m * m;
      sd = sqrtf(fmaxf(var, 1e-2f));
      means[(size_t)i * OB + tid] = m; stds[(size_t)i * OB + tid] = sd;
    }
  }
  sum[tid] = s; sumsq[tid] = q; mean[tid] = m; stdv[tid] = sd;
  if (tid == 0) *count = c;
}
}  // namespace dmv
