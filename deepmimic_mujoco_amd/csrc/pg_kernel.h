// pg_kernel.h — the policy half of the TRPO update (src/trpo.py:228-230, 249-283; src/cg.py:2-34) as hand-written kernels: the
// surrogate losses and their flat gradient, and the Fisher-vector product of conjugate gradients, each ONE launch (+ a reduction)
// instead of the dozens of launch-bound autograd micro-kernels a 56-100-100-28 tanh MLP turns into.
//
//   policy (src/mlp_policy_trpo.py:50-60):  z = clip((ob - mean) / std, +-5),  h1 = tanh(z W1 + b1),  h2 = tanh(h1 W2 + b2),
//                                           m = h2 W3 + b3,  pd = N(m, exp(logstd)) with a state-independent logstd
//   losses (src/trpo.py:118-134):           surrgain = mean(exp(logp_new - logp_old) atarg),  meankl = mean KL(old || new),
//                                           optimgain = surrgain + entcoeff * mean entropy
//   F v (src/trpo.py:228-230):              gradient of (grad meankl . v) at new == old.  For a Gaussian with state-independent logstd that
//                                           Hessian is EXACTLY  J^T diag(1 / sigma^2) J / N  on the mean parameters (J = d m / d theta) and
//                                           2 I on logstd (the residual m_old - m is zero, so the network's second derivatives drop out):
//                                           one forward-mode pass (J v) and one reverse pass (J^T u) per sample, no double back-propagation.
//
// A block owns a fixed set of 32-sample tiles (tile t -> block t mod gridDim.x) and keeps its weight-gradient tiles in REGISTERS across all of
// them; partial gradients are written once per block and summed in block order by k_pg_reduce: results do not depend on timing.
//
// Every product runs on the matrix cores in fp32 (v_mfma_f32_32x32x2_f32; the 28-wide output layer as 16x16x4 tiles), operands straight from
// LDS with no re-layout between layers:
//   * activations sit transposed, [unit][sample] with a row stride of 33 floats: a row pair [k, k+1][32 samples] is a B operand (forward,
//     "units x samples" results), a column pair [32 units][s, s+1] is an A or B operand of the weight-gradient products (sum over samples);
//     both reads are bank-conflict free;
//   * the parameters are ONE copy of theta in LDS.  theta's order (W1, b1, W2, b2, W3, b3) makes each bias the row after its matrix, so with a
//     constant row of ones appended to z / h1 / h2 the biases are part of the products, forward AND backward: the bias gradients are row
//     56 / 100 / 100 of the weight-gradient tiles, which land in theta order by themselves;
//   * wave w owns output units 32 w .. 32 w + 31 of the hidden layers (100 padded to 128: rows past 99 read finite junk and are never stored);
//     a Fisher product keeps ITS slices of the direction v in registers for the whole launch (106 per lane) instead of streaming v per tile.
// fp32 like the reference's TF graph; loss sums leave the block in float64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmg {

constexpr int OB = 56, H = 100, AC = 28, SB = 32, SBP = SB + 1;
constexpr int O_W1 = 0, O_B1 = O_W1 + OB * H, O_W2 = O_B1 + H, O_B2 = O_W2 + H * H, O_W3 = O_B2 + H, O_B3 = O_W3 + H * AC, O_LS = O_B3 + AC, NP = O_LS + AC;
constexpr int NPAD = (NP + 63) / 64 * 64;
constexpr int NWT = (O_LS + 3) / 4 * 4 + 4;           // theta up to logstd, as float4s (the pad holds the first logstd entries: never used as a weight)
constexpr int MAX_BLOCKS = 256;                       // one block per CU (LDS-limited)
constexpr int ZR = OB + 2, HR = H + 4, MR = 32;       // rows: z + {ones, zeros};  h + {ones, 3 x zeros};  action rows padded to a tile
enum { MODE_LOSS = 0, MODE_GRAD = 1, MODE_FVP = 2 };
static_assert(O_LS % 4 == 0 && NP >= NWT, "theta is copied to LDS as float4s");

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct alignas(16) PgShared {
  float Wt[NWT];                                      // theta: W1 [56][100], b1, W2 [100][100], b2, W3 [100][28], b3
  float z[ZR][SBP];                                   // row 56 = 1, row 57 = 0
  float h1[HR][SBP], h2[HR][SBP];                     // row 100 = 1, rows 101..103 = 0
  float a1[HR][SBP], a2[HR][SBP];                     // FVP: tangents d h1, d h2;  backward: deltas of layer 1, 2
  float mo[MR][SBP];                                  // action mean -> output gradient G
  float act[AC][SBP], om[AC][SBP];                    // the tile's actions and old means, transposed like mo
  float ls[AC], ols[AC], at[SB];
  float redp[8][SB][2];                               // partial sums of the per-sample log-ratio / KL over eight action groups
  float red[SB][2];
};
static_assert(sizeof(PgShared) <= 160 * 1024, "PgShared must fit a CU's LDS");
// operand reads past a buffer's rows (padded unit tiles) must stay inside the struct: the furthest is W3's row 127 as an A operand
static_assert(O_W3 + 127 * AC + AC <= NWT + ZR * SBP, "padded W3 rows read into z");

__device__ inline v16f mfma32(float a, float b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ inline v4f mfma16(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// tanh(x) = sign(x) (1 - t) / (1 + t), t = exp(-2 |x|): the hardware exponential and reciprocal, absolute error ~1e-7 — an ulp of the
// activation's range, like the float32 graph of the reference.  (The library tanhf is ~60 instructions with two divergent branches: with the
// products on the matrix cores it was a quarter of a tile's time.)
__device__ inline float tanh_fast(float x) {
  const float t = __expf(-2.0f * fabsf(x));
  return copysignf((1.0f - t) * __frcp_rn(1.0f + t), x);
}
// row of a 32x32 result tile held in register r by the lanes of half `hf`
__device__ inline int row32(int r, int hf) { return 8 * (r / 4) + 4 * hf + (r % 4); }

// ob: [.., 56] f32, sample i at row i * stride.  theta: packed policy parameters (NP).  MODE_LOSS / MODE_GRAD: ac [n, 28], atarg [n],
// old_logstd [28], old_mean [n, 28] — with write_old != 0 the kernel treats old == new and WRITES old_mean (src/trpo.py:247 assign_old_eq_new).
// MODE_FVP: v = the direction (NP, in global memory).  partial: [gridDim.x][NPAD] f32 gradients; lpart: [gridDim.x][2] f64 (sum of ratio * atarg, sum of KL).
template <int MODE>
__global__ __launch_bounds__(256) void k_pg(const float* __restrict__ ob, int stride, int n, const float* __restrict__ ac, const float* __restrict__ atarg,
                                            float* __restrict__ old_mean, const float* __restrict__ old_logstd, int write_old,
                                            const float* __restrict__ theta, const float* __restrict__ v, const float* __restrict__ mean, const float* __restrict__ stdv,
                                            float inv_n, float* __restrict__ partial, double* __restrict__ lpart) {
  __shared__ PgShared S;                                      // 141 KB: one block per CU
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, li = l & 31, hf = l >> 5, l16 = l & 15, q = l >> 4;
  const int u0 = 32 * w;                                      // this wave's hidden-unit tile
  const int a0 = 16 * (w >> 1), s0 = 16 * (w & 1);            // ... and its 16 actions x 16 samples of the output layer
  {
    // every pad column / row starts as zero: operands that reach into them must be finite
    float4* all = reinterpret_cast<float4*>(&S);
    for (int i = tid; i < (int)(sizeof(PgShared) / 16); i += 256) all[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    __syncthreads();
    const float4* g = reinterpret_cast<const float4*>(theta); float4* d = reinterpret_cast<float4*>(S.Wt);
#pragma unroll 6
    for (int i = tid; i < NWT / 4; i += 256) d[i] = g[i];
    if (tid < SB) { S.z[OB][tid] = 1.0f; S.h1[H][tid] = 1.0f; S.h2[H][tid] = 1.0f; }
    if (tid < AC) { S.ls[tid] = theta[O_LS + tid]; S.ols[tid] = (MODE != MODE_FVP && !write_old) ? old_logstd[tid] : theta[O_LS + tid]; }
  }
  // a Fisher product's direction, as this wave's A operands (v is laid out like theta: its bias entries are the rows the ones multiply)
  float v1[(OB + 2) / 2], v2[(H + 2) / 2], v3[(H + 4) / 4];
  if (MODE == MODE_FVP) {
#pragma unroll
    for (int t = 0; t < (OB + 2) / 2; t++) { const int k = 2 * t + hf; v1[t] = (k <= OB && u0 + li < H) ? v[O_W1 + k * H + u0 + li] : 0.0f; }
#pragma unroll
    for (int t = 0; t < (H + 2) / 2; t++) { const int k = 2 * t + hf; v2[t] = (k <= H && u0 + li < H) ? v[O_W2 + k * H + u0 + li] : 0.0f; }
#pragma unroll
    for (int t = 0; t < (H + 4) / 4; t++) { const int k = 4 * t + q; v3[t] = (k <= H && a0 + l16 < AC) ? v[O_W3 + k * AC + a0 + l16] : 0.0f; }
  }
  // register-resident partial gradients of this block, in theta order: rows k (inputs, + the bias row) x this wave's 32 output columns
  v16f gW1[2], gW2[4], gW3;
#pragma unroll
  for (int r = 0; r < 16; r++) { gW1[0][r] = 0.0f; gW1[1][r] = 0.0f; gW2[0][r] = 0.0f; gW2[1][r] = 0.0f; gW2[2][r] = 0.0f; gW2[3][r] = 0.0f; gW3[r] = 0.0f; }
  double lsum0 = 0.0, lsum1 = 0.0;
  const int ntiles = (n + SB - 1) / SB;
  // A tile's inputs are contiguous in memory (32 rows of ob / ac / old_mean): every thread fetches its share of the NEXT tile while the block
  // works on this one, and hands it to LDS (transposed) at the top of the loop.
  constexpr int NZ = SB * OB / 256, NA = (SB * AC + 255) / 256;
  static_assert(SB * OB % 256 == 0, "the observation tile is read in whole rounds of the block");
  float mu[NZ], sd[NZ], obx[NZ], acx[NA], omx[NA], atx = 0.0f;
#pragma unroll
  for (int j = 0; j < NZ; j++) { const int k = (tid + 256 * j) % OB; mu[j] = mean[k]; sd[j] = stdv[k]; }
  auto fetch = [&](int tile) {
    const int s0n = tile * SB;
#pragma unroll
    for (int j = 0; j < NZ; j++) { const int i = tid + 256 * j, r = s0n + i / OB; obx[j] = r < n ? ob[(size_t)r * stride * OB + i % OB] : 0.0f; }
    if (MODE != MODE_FVP) {
#pragma unroll
      for (int j = 0; j < NA; j++) {
        const int i = tid + 256 * j; const bool ok = i < SB * AC && s0n + i / AC < n;
        acx[j] = ok ? ac[(size_t)s0n * AC + i] : 0.0f;
        omx[j] = (ok && !write_old) ? old_mean[(size_t)s0n * AC + i] : 0.0f;
      }
      if (tid < SB) atx = s0n + tid < n ? atarg[s0n + tid] : 0.0f;
    }
  };
  fetch(blockIdx.x);
  float gls[NA];                                              // d / d logstd_a: lanes 0 and 32 of wave w hold actions 2 w + (lane / 32) + 8 j
#pragma unroll
  for (int j = 0; j < NA; j++) gls[j] = 0.0f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int s0g = tile * SB;
    __syncthreads();                                          // the previous tile's readers are done (first pass: the weights are in place)
#pragma unroll
    for (int j = 0; j < NZ; j++) {
      const int i = tid + 256 * j, sm = i / OB, k = i % OB;
      S.z[k][sm] = (s0g + sm < n) ? fminf(fmaxf((obx[j] - mu[j]) / sd[j], -5.0f), 5.0f) : 0.0f;
    }
    if (MODE != MODE_FVP) {
#pragma unroll
      for (int j = 0; j < NA; j++) { const int i = tid + 256 * j; if (i < SB * AC) { S.act[i % AC][i / AC] = acx[j]; S.om[i % AC][i / AC] = omx[j]; } }
      if (tid < SB) S.at[tid] = atx;
    }
    fetch(tile + gridDim.x);
    __syncthreads();
    // ---- layer 1 (+ tangent): h1 = tanh(W1ext^T zext) ----
    {
      v16f acc, dac;
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[r] = 0.0f; dac[r] = 0.0f; }
#pragma unroll
      for (int t = 0; t < (OB + 2) / 2; t++) {
        const int k = 2 * t + hf;
        const float a = S.Wt[O_W1 + k * H + u0 + li], b = S.z[k][li];
        acc = mfma32(a, b, acc);
        if (MODE == MODE_FVP) dac = mfma32(v1[t], b, dac);
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int u = u0 + row32(r, hf);
        if (u < H) {
          const float h = tanh_fast(acc[r]);
          S.h1[u][li] = h;
          if (MODE == MODE_FVP) S.a1[u][li] = dac[r] * (1.0f - h * h);
        }
      }
    }
    __syncthreads();
    // ---- layer 2 (+ tangent: W2^T d h1 + V2ext^T h1ext) ----
    {
      v16f acc, dac;
#pragma unroll
      for (int r = 0; r < 16; r++) { acc[r] = 0.0f; dac[r] = 0.0f; }
#pragma unroll
      for (int t = 0; t < (H + 2) / 2; t++) {
        const int k = 2 * t + hf;
        const float a = S.Wt[O_W2 + k * H + u0 + li], b = S.h1[k][li];
        acc = mfma32(a, b, acc);
        if (MODE == MODE_FVP) {
          dac = mfma32(v2[t], b, dac);
          if (t < H / 2) dac = mfma32(a, S.a1[k][li], dac);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int u = u0 + row32(r, hf);
        if (u < H) {
          const float h = tanh_fast(acc[r]);
          S.h2[u][li] = h;
          if (MODE == MODE_FVP) S.a2[u][li] = dac[r] * (1.0f - h * h);
        }
      }
    }
    __syncthreads();
    // ---- output layer: the action mean (FVP: its tangent J v), 16 actions x 16 samples per wave ----
    {
      v4f acc = {0.0f, 0.0f, 0.0f, 0.0f};
      const int ac_col = a0 + l16 < AC ? a0 + l16 : AC - 1;   // (rows 28..31 of the tile: a copy of row 27, not stored)
#pragma unroll
      for (int t = 0; t < (H + 4) / 4; t++) {
        const int k = 4 * t + q, kw = k < H ? k : H;          // rows 101..103 of h2 are zero: any finite weight does
        const float a = S.Wt[O_W3 + kw * AC + ac_col], b = S.h2[k][s0 + l16];
        if (MODE == MODE_FVP) {
          acc = mfma16(v3[t], b, acc);
          if (t < H / 4) acc = mfma16(a, S.a2[k][s0 + l16], acc);
        } else acc = mfma16(a, b, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) { const int a = a0 + 4 * q + r; if (a < AC) S.mo[a][s0 + l16] = acc[r]; }
    }
    __syncthreads();
    if (MODE == MODE_FVP) {
      // ---- output gradient of the Fisher product: u = (J v) / sigma^2 / N, in place ----
      for (int i = tid; i < AC * SB; i += 256) {
        const int a = i / SB, sm = i % SB;
        S.mo[a][sm] = (s0g + sm < n) ? S.mo[a][sm] * __expf(-2.0f * S.ls[a]) * inv_n : 0.0f;
      }
    } else {
      // ---- per-sample likelihood ratio and KL (src/distributions.py:235-243): thread = (sample, one of eight action groups) ----
      const int sm = tid % SB, grp = tid / SB;
      {
        float d = 0.0f, kl = 0.0f;                            // logp_new - logp_old = neglogp_old - neglogp_new
#pragma unroll
        for (int j = 0; j < NA; j++) {
          const int a = grp + 8 * j;
          if (a < AC) {
            const float m = S.mo[a][sm], x = S.act[a][sm], mold = write_old ? m : S.om[a][sm];
            const float ls = S.ls[a], lo = S.ols[a];
            const float en = (x - m) * __expf(-ls), eo = (x - mold) * __expf(-lo);
            d += 0.5f * (eo * eo - en * en) + (lo - ls);
            kl += ls - lo + (__expf(2.0f * lo) + (mold - m) * (mold - m)) * 0.5f * __expf(-2.0f * ls) - 0.5f;
          }
        }
        S.redp[grp][sm][0] = d; S.redp[grp][sm][1] = kl;
      }
      if (write_old) {                                        // oldpi <- pi: the tile's means, rows contiguous in old_mean
#pragma unroll
        for (int j = 0; j < NA; j++) { const int i = tid + 256 * j; if (i < SB * AC && s0g + i / AC < n) old_mean[(size_t)s0g * AC + i] = S.mo[i % AC][i / AC]; }
      }
      __syncthreads();
      if (tid < SB) {
        float d = 0.0f, kl = 0.0f;
#pragma unroll
        for (int g8 = 0; g8 < 8; g8++) { d += S.redp[g8][tid][0]; kl += S.redp[g8][tid][1]; }
        const bool ok = s0g + tid < n;
        const float ra = ok ? __expf(d) * S.at[tid] : 0.0f;
        if (!ok) kl = 0.0f;
        S.red[tid][0] = ra; S.red[tid][1] = kl;
        double b0 = (double)ra, b1 = (double)kl;              // the tile's loss sums: a butterfly over the 32 lanes, in float64
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) { b0 += __shfl_xor(b0, o, 32); b1 += __shfl_xor(b1, o, 32); }
        lsum0 += b0; lsum1 += b1;
      }
      if (MODE == MODE_LOSS) continue;
      __syncthreads();
      // G = d optimgain / d mean = ratio atarg (x - m) / sigma^2 / N, in place of the mean;  on the way
      // d optimgain / d logstd_a = sum_s ratio atarg (((x - m) / sigma)^2 - 1) / N   (the entropy bonus's constant is added by the reduction)
#pragma unroll
      for (int j = 0; j < NA; j++) {
        const int a = grp + 8 * j;
        float e2 = 0.0f;
        if (a < AC) {
          const float ra = S.red[sm][0], dx = S.act[a][sm] - S.mo[a][sm], is = __expf(-S.ls[a]);
          e2 = ra * (dx * is * dx * is - 1.0f);
          S.mo[a][sm] = ra * inv_n * dx * is * is;
        }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) e2 += __shfl_xor(e2, o, 32);
        gls[j] += e2 * inv_n;
      }
    }
    __syncthreads();
    // ---- reverse pass with the output gradient G = S.mo.  Sums over the 32 samples: operands are column pairs [.][s, s + 1] ----
    {
      v16f acc;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
      for (int t = 0; t < SB / 2; t++) gW3 = mfma32(S.h2[u0 + li][2 * t + hf], S.mo[li][2 * t + hf], gW3);      // dW3ext = h2ext G^T  (row 100: db3)
#pragma unroll
      for (int t = 0; t < AC / 2; t++) acc = mfma32(S.Wt[O_W3 + (u0 + li) * AC + 2 * t + hf], S.mo[2 * t + hf][li], acc);   // W3 G
#pragma unroll
      for (int r = 0; r < 16; r++) {                          // delta2 = (W3 G) (1 - h2^2)   (a2's tangents have been consumed)
        const int u = u0 + row32(r, hf);
        if (u < H) { const float h = S.h2[u][li]; S.a2[u][li] = acc[r] * (1.0f - h * h); }
      }
    }
    __syncthreads();
    {
      v16f acc;
#pragma unroll
      for (int r = 0; r < 16; r++) acc[r] = 0.0f;
#pragma unroll
      for (int t = 0; t < SB / 2; t++) {                      // dW2ext = h1ext delta2^T  (row 100: db2)
        const float b = S.a2[u0 + li][2 * t + hf];
#pragma unroll
        for (int mt = 0; mt < 4; mt++) gW2[mt] = mfma32(S.h1[32 * mt + li][2 * t + hf], b, gW2[mt]);
      }
#pragma unroll
      for (int t = 0; t < H / 2; t++) acc = mfma32(S.Wt[O_W2 + (u0 + li) * H + 2 * t + hf], S.a2[2 * t + hf][li], acc);    // W2 delta2
#pragma unroll
      for (int r = 0; r < 16; r++) {                          // delta1 = (W2 delta2) (1 - h1^2)
        const int u = u0 + row32(r, hf);
        if (u < H) { const float h = S.h1[u][li]; S.a1[u][li] = acc[r] * (1.0f - h * h); }
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < SB / 2; t++) {                        // dW1ext = zext delta1^T  (row 56: db1)
      const float b = S.a1[u0 + li][2 * t + hf];
#pragma unroll
      for (int mt = 0; mt < 2; mt++) gW1[mt] = mfma32(S.z[32 * mt + li][2 * t + hf], b, gW1[mt]);
    }
  }
  if (tid == 0) { lpart[2 * blockIdx.x] = lsum0; lpart[2 * blockIdx.x + 1] = lsum1; }
  if (MODE == MODE_LOSS) return;
  float* out = partial + (size_t)blockIdx.x * NPAD;
  const int col = u0 + li;
#pragma unroll
  for (int r = 0; r < 16; r++) {
    const int i = row32(r, hf);
    if (col < H) {
#pragma unroll
      for (int mt = 0; mt < 2; mt++) if (32 * mt + i <= OB) out[O_W1 + (32 * mt + i) * H + col] = gW1[mt][r];       // row 56 is b1's place in theta
#pragma unroll
      for (int mt = 0; mt < 4; mt++) if (32 * mt + i <= H) out[O_W2 + (32 * mt + i) * H + col] = gW2[mt][r];        // row 100: b2
    }
    if (li < AC && u0 + i <= H) out[O_W3 + (u0 + i) * AC + li] = gW3[r];                                            // row 100: b3
  }
  if (tid % SB == 0) {                                                      // (Fisher product: zero here; the reduction writes 2 v)
#pragma unroll
    for (int j = 0; j < NA; j++) if (tid / SB + 8 * j < AC) out[O_LS + tid / SB + 8 * j] = gls[j];
  }
}

// partial gradients summed in block order (eight loads in flight; the additions stay in order), plus the parts that do not come from the samples:
// gradient: + entcoeff on logstd (d (entcoeff * mean entropy) / d logstd_a = entcoeff);  Fisher product: 2 v on logstd.
// Thread 0 of block 0 also finishes the losses: out_losses = {surrgain, meankl} = sums / n.
__global__ __launch_bounds__(256) void k_pg_reduce(const float* __restrict__ partial, const double* __restrict__ lpart, int nblk, int mode, float entcoeff,
                                                   const float* __restrict__ v, double inv_n, float* __restrict__ out, double* __restrict__ out_losses) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p == 0 && mode != MODE_FVP) {
    double a0 = 0.0, a1 = 0.0;
    for (int b = 0; b < nblk; b++) { a0 += lpart[2 * b]; a1 += lpart[2 * b + 1]; }
    out_losses[0] = a0 * inv_n; out_losses[1] = a1 * inv_n;
  }
  if (p >= NP || mode == MODE_LOSS) return;
  if (mode == MODE_FVP && p >= O_LS) { out[p] = 2.0f * v[p]; return; }
  float g = 0.0f;
  int b = 0;
  for (; b + 8 <= nblk; b += 8) {
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) x[u] = partial[(size_t)(b + u) * NPAD + p];
#pragma unroll
    for (int u = 0; u < 8; u++) g += x[u];
  }
  for (; b < nblk; b++) g += partial[(size_t)b * NPAD + p];
  if (mode == MODE_GRAD && p >= O_LS) g += entcoeff;
  out[p] = g;
}

}  // namespace dmg
