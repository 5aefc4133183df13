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
// Weights sit in LDS (73.6 KB), activations transposed ([unit][sample]) beside them; the direction v of a Fisher product is streamed from L2.
// fp32 like the reference's TF graph; loss sums leave the block in float64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmg {

constexpr int OB = 56, H = 100, AC = 28, SB = 32;
constexpr int O_W1 = 0, O_B1 = O_W1 + OB * H, O_W2 = O_B1 + H, O_B2 = O_W2 + H * H, O_W3 = O_B2 + H, O_B3 = O_W3 + H * AC, O_LS = O_B3 + AC, NP = O_LS + AC;
constexpr int NPAD = (NP + 63) / 64 * 64;
constexpr int MAX_BLOCKS = 256;                       // one block per CU (LDS-limited)
enum { MODE_LOSS = 0, MODE_GRAD = 1, MODE_FVP = 2 };

struct alignas(16) PgShared {
  float W1[OB * H], W2[H * H], W3[H * AC];
  float z[OB][SB], h1[H][SB], h2[H][SB];
  float a1[H][SB], a2[H][SB];                         // FVP: tangents d h1, d h2;  backward: deltas of layer 1, 2
  float mo[AC][SB];                                   // action mean -> output gradient G
  float b1[H], b2[H], b3[AC], ls[AC], ols[AC];
  float red[SB][2];
};
static_assert(sizeof(PgShared) <= 160 * 1024, "PgShared must fit a CU's LDS");

__device__ inline float4 f4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ inline void fma4(float4& a, float w, const float4& x) { a.x += w * x.x; a.y += w * x.y; a.z += w * x.z; a.w += w * x.w; }
__device__ inline float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ inline float4 tanh4(const float4& a) { return make_float4(tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w)); }
__device__ inline float4 dtanh4(const float4& d, const float4& h) { return make_float4(d.x * (1.0f - h.x * h.x), d.y * (1.0f - h.y * h.y), d.z * (1.0f - h.z * h.z), d.w * (1.0f - h.w * h.w)); }

// acc[a][b] += sum_s A[i0 + a][s] B[j0 + b][s]   (both operands [.][SB] in LDS)
template <int NA, int NB>
__device__ inline void tile_acc(const float (*A)[SB], const float (*B)[SB], int i0, int j0, float (&acc)[NA][NB]) {
#pragma unroll 2
  for (int s4 = 0; s4 < SB; s4 += 4) {
    float4 av[NA], bv[NB];
#pragma unroll
    for (int a = 0; a < NA; a++) av[a] = f4(&A[i0 + a][s4]);
#pragma unroll
    for (int b = 0; b < NB; b++) bv[b] = f4(&B[j0 + b][s4]);
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
      for (int b = 0; b < NB; b++) acc[a][b] += dot4(av[a], bv[b]);
  }
}
__device__ inline float row_sum(const float* r) {
  float a = 0.0f;
#pragma unroll
  for (int s4 = 0; s4 < SB; s4 += 4) { const float4 x = f4(r + s4); a += (x.x + x.y) + (x.z + x.w); }
  return a;
}

// ob: [.., 56] f32, sample i at row i * stride.  theta: packed policy parameters (NP).  MODE_LOSS / MODE_GRAD: ac [n, 28], atarg [n],
// old_logstd [28], old_mean [n, 28] — with write_old != 0 the kernel treats old == new and WRITES old_mean (src/trpo.py:247 assign_old_eq_new).
// MODE_FVP: v = the direction (NP, in global memory).  partial: [gridDim.x][NPAD] f32 gradients; lpart: [gridDim.x][2] f64 (sum of ratio * atarg, sum of KL).
template <int MODE>
__global__ __launch_bounds__(256) void k_pg(const float* __restrict__ ob, int stride, int n, const float* __restrict__ ac, const float* __restrict__ atarg,
                                            float* __restrict__ old_mean, const float* __restrict__ old_logstd, int write_old,
                                            const float* __restrict__ theta, const float* __restrict__ v, const float* __restrict__ mean, const float* __restrict__ stdv,
                                            float inv_n, float* __restrict__ partial, double* __restrict__ lpart) {
  __shared__ PgShared S;                                      // 137 KB: one block per CU
  const int tid = threadIdx.x;
  {
    const float4* g1 = reinterpret_cast<const float4*>(theta + O_W1); float4* l1 = reinterpret_cast<float4*>(S.W1);
    const float4* g2 = reinterpret_cast<const float4*>(theta + O_W2); float4* l2 = reinterpret_cast<float4*>(S.W2);
    const float4* g3 = reinterpret_cast<const float4*>(theta + O_W3); float4* l3 = reinterpret_cast<float4*>(S.W3);
#pragma unroll 6
    for (int i = tid; i < OB * H / 4; i += 256) l1[i] = g1[i];
#pragma unroll 10
    for (int i = tid; i < H * H / 4; i += 256) l2[i] = g2[i];
#pragma unroll 3
    for (int i = tid; i < H * AC / 4; i += 256) l3[i] = g3[i];
  }
  if (tid < H) { S.b1[tid] = theta[O_B1 + tid]; S.b2[tid] = theta[O_B2 + tid]; }
  if (tid < AC) { S.b3[tid] = theta[O_B3 + tid]; S.ls[tid] = theta[O_LS + tid]; S.ols[tid] = (MODE != MODE_FVP && !write_old) ? old_logstd[tid] : theta[O_LS + tid]; }
  // register-resident partial gradients of this block.  Owners: W1 tiles threads 0..139, W2 tiles 0..249, W3 tiles 0..99, b2 0..99,
  // b3 100..127, logstd 128..155, b1 156..255
  float gW1[4][10], gW2[4][10], gW3[4][7], gb2 = 0.0f, gbx = 0.0f;          // gbx: b3 / logstd / b1 by thread range
#pragma unroll
  for (int a = 0; a < 4; a++) {
#pragma unroll
    for (int b = 0; b < 10; b++) { gW1[a][b] = 0.0f; gW2[a][b] = 0.0f; }
#pragma unroll
    for (int b = 0; b < 7; b++) gW3[a][b] = 0.0f;
  }
  double lsum0 = 0.0, lsum1 = 0.0;
  const int sq = (tid % 8) * 4, uq = (tid / 8) * 4;           // this thread's 4 samples x 4 units
  const bool dense = tid < 200, outl = tid < 56;              // output layer: 7 unit groups x 8 sample groups
  const int ntiles = (n + SB - 1) / SB;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int s0 = tile * SB;
    __syncthreads();                                          // the previous tile's readers are done (first pass: the weights are in place)
#pragma unroll 7
    for (int i = tid; i < SB * OB; i += 256) {
      const int sm = i / OB, k = i % OB, r = s0 + sm;
      float x = 0.0f;
      if (r < n) x = fminf(fmaxf((ob[(size_t)r * stride * OB + k] - mean[k]) / stdv[k], -5.0f), 5.0f);
      S.z[k][sm] = x;
    }
    __syncthreads();
    // ---- layer 1 (+ tangent) ----
    if (dense) {
      float4 acc[4], dac[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float b = S.b1[uq + u]; acc[u] = make_float4(b, b, b, b);
        if (MODE == MODE_FVP) { const float c = v[O_B1 + uq + u]; dac[u] = make_float4(c, c, c, c); }
      }
#pragma unroll 4
      for (int k = 0; k < OB; k++) {
        const float4 x = f4(&S.z[k][sq]), w = f4(&S.W1[k * H + uq]);
        fma4(acc[0], w.x, x); fma4(acc[1], w.y, x); fma4(acc[2], w.z, x); fma4(acc[3], w.w, x);
        if (MODE == MODE_FVP) { const float4 q = f4(&v[O_W1 + k * H + uq]); fma4(dac[0], q.x, x); fma4(dac[1], q.y, x); fma4(dac[2], q.z, x); fma4(dac[3], q.w, x); }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float4 h = tanh4(acc[u]);
        *reinterpret_cast<float4*>(&S.h1[uq + u][sq]) = h;
        if (MODE == MODE_FVP) *reinterpret_cast<float4*>(&S.a1[uq + u][sq]) = dtanh4(dac[u], h);
      }
    }
    __syncthreads();
    // ---- layer 2 (+ tangent) ----
    if (dense) {
      float4 acc[4], dac[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float b = S.b2[uq + u]; acc[u] = make_float4(b, b, b, b);
        if (MODE == MODE_FVP) { const float c = v[O_B2 + uq + u]; dac[u] = make_float4(c, c, c, c); }
      }
#pragma unroll 4
      for (int k = 0; k < H; k++) {
        const float4 x = f4(&S.h1[k][sq]), w = f4(&S.W2[k * H + uq]);
        fma4(acc[0], w.x, x); fma4(acc[1], w.y, x); fma4(acc[2], w.z, x); fma4(acc[3], w.w, x);
        if (MODE == MODE_FVP) {
          const float4 dx = f4(&S.a1[k][sq]), q = f4(&v[O_W2 + k * H + uq]);
          fma4(dac[0], w.x, dx); fma4(dac[1], w.y, dx); fma4(dac[2], w.z, dx); fma4(dac[3], w.w, dx);
          fma4(dac[0], q.x, x); fma4(dac[1], q.y, x); fma4(dac[2], q.z, x); fma4(dac[3], q.w, x);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const float4 h = tanh4(acc[u]);
        *reinterpret_cast<float4*>(&S.h2[uq + u][sq]) = h;
        if (MODE == MODE_FVP) *reinterpret_cast<float4*>(&S.a2[uq + u][sq]) = dtanh4(dac[u], h);
      }
    }
    __syncthreads();
    // ---- output layer: the action mean (FVP: its tangent J v) ----
    if (outl) {
      float4 acc[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const float b = MODE == MODE_FVP ? v[O_B3 + uq + u] : S.b3[uq + u]; acc[u] = make_float4(b, b, b, b); }
#pragma unroll 4
      for (int k = 0; k < H; k++) {
        const float4 w = f4(&S.W3[k * AC + uq]), x = f4(&S.h2[k][sq]);
        if (MODE == MODE_FVP) {
          const float4 dx = f4(&S.a2[k][sq]), q = f4(&v[O_W3 + k * AC + uq]);
          fma4(acc[0], w.x, dx); fma4(acc[1], w.y, dx); fma4(acc[2], w.z, dx); fma4(acc[3], w.w, dx);
          fma4(acc[0], q.x, x); fma4(acc[1], q.y, x); fma4(acc[2], q.z, x); fma4(acc[3], q.w, x);
        } else { fma4(acc[0], w.x, x); fma4(acc[1], w.y, x); fma4(acc[2], w.z, x); fma4(acc[3], w.w, x); }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) *reinterpret_cast<float4*>(&S.mo[uq + u][sq]) = acc[u];
    }
    __syncthreads();
    if (MODE == MODE_FVP) {
      // ---- output gradient of the Fisher product: u = (J v) / sigma^2 / N, in place ----
      for (int i = tid; i < AC * SB; i += 256) {
        const int a = i / SB, sm = i % SB;
        S.mo[a][sm] = (s0 + sm < n) ? S.mo[a][sm] * __expf(-2.0f * S.ls[a]) * inv_n : 0.0f;
      }
    } else {
      // ---- per-sample likelihood ratio and KL (src/distributions.py:235-243) ----
      if (tid < SB) {
        const int r = s0 + tid;
        float ra = 0.0f, kl = 0.0f;
        if (r < n) {
          float d = 0.0f;                                     // logp_new - logp_old = neglogp_old - neglogp_new
          for (int a = 0; a < AC; a++) {
            const float m = S.mo[a][tid], x = ac[(size_t)r * AC + a];
            float mold = m;
            if (write_old) old_mean[(size_t)r * AC + a] = m; else mold = old_mean[(size_t)r * AC + a];
            const float ls = S.ls[a], lo = S.ols[a];
            const float en = (x - m) * __expf(-ls), eo = (x - mold) * __expf(-lo);
            d += 0.5f * (eo * eo - en * en) + (lo - ls);
            kl += ls - lo + (__expf(2.0f * lo) + (mold - m) * (mold - m)) * 0.5f * __expf(-2.0f * ls) - 0.5f;
          }
          ra = __expf(d) * atarg[r];
        }
        S.red[tid][0] = ra; S.red[tid][1] = kl;
      }
      __syncthreads();
      if (tid == 0) { double a0 = 0.0, a1 = 0.0; for (int sm = 0; sm < SB; sm++) { a0 += (double)S.red[sm][0]; a1 += (double)S.red[sm][1]; } lsum0 += a0; lsum1 += a1; }
      if (MODE == MODE_LOSS) continue;
      // d optimgain / d logstd_a = sum_s ratio atarg (((x - m) / sigma)^2 - 1) / N   (the entropy bonus's constant is added by the reduction)
      if (tid >= 128 && tid < 128 + AC) {
        const int a = tid - 128;
        const float is = __expf(-S.ls[a]);
        float acc = 0.0f;
        for (int sm = 0; sm < SB; sm++) { const int r = s0 + sm; if (r < n) { const float e = (ac[(size_t)r * AC + a] - S.mo[a][sm]) * is; acc += S.red[sm][0] * (e * e - 1.0f); } }
        gbx += acc * inv_n;
      }
      __syncthreads();
      for (int i = tid; i < AC * SB; i += 256) {              // G = d optimgain / d mean = ratio atarg (x - m) / sigma^2 / N, in place of the mean
        const int a = i / SB, sm = i % SB, r = s0 + sm;
        S.mo[a][sm] = (r < n) ? S.red[sm][0] * inv_n * (ac[(size_t)r * AC + a] - S.mo[a][sm]) * __expf(-2.0f * S.ls[a]) : 0.0f;
      }
    }
    __syncthreads();
    // ---- reverse pass with the output gradient G = S.mo ----
    if (tid < 100) tile_acc<4, 7>(S.h2, S.mo, (tid / 4) * 4, (tid % 4) * 7, gW3);          // dW3 = h2^T G
    else if (tid < 100 + AC) gbx += row_sum(S.mo[tid - 100]);                                // db3
    if (dense) {                                                                            // delta2 = (W3 G) (1 - h2^2)   (a2's tangents have been consumed)
      float4 acc[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
#pragma unroll 7
      for (int a = 0; a < AC; a += 4) {
        const float4 g0 = f4(&S.mo[a][sq]), g1 = f4(&S.mo[a + 1][sq]), g2 = f4(&S.mo[a + 2][sq]), g3 = f4(&S.mo[a + 3][sq]);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float4 w = f4(&S.W3[(uq + u) * AC + a]);
          fma4(acc[u], w.x, g0); fma4(acc[u], w.y, g1); fma4(acc[u], w.z, g2); fma4(acc[u], w.w, g3);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) *reinterpret_cast<float4*>(&S.a2[uq + u][sq]) = dtanh4(acc[u], f4(&S.h2[uq + u][sq]));
    }
    __syncthreads();
    if (tid < 250) tile_acc<4, 10>(S.h1, S.a2, (tid / 10) * 4, (tid % 10) * 10, gW2);       // dW2 = h1^T delta2
    if (tid < H) gb2 += row_sum(S.a2[tid]);                                                  // db2
    if (dense) {                                                                            // delta1 = (W2 delta2) (1 - h1^2)
      float4 acc[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
#pragma unroll 2
      for (int j = 0; j < H; j += 4) {
        const float4 d0 = f4(&S.a2[j][sq]), d1 = f4(&S.a2[j + 1][sq]), d2 = f4(&S.a2[j + 2][sq]), d3 = f4(&S.a2[j + 3][sq]);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float4 w = f4(&S.W2[(uq + u) * H + j]);
          fma4(acc[u], w.x, d0); fma4(acc[u], w.y, d1); fma4(acc[u], w.z, d2); fma4(acc[u], w.w, d3);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) *reinterpret_cast<float4*>(&S.a1[uq + u][sq]) = dtanh4(acc[u], f4(&S.h1[uq + u][sq]));
    }
    __syncthreads();
    if (tid < 140) tile_acc<4, 10>(S.z, S.a1, (tid / 10) * 4, (tid % 10) * 10, gW1);        // dW1 = z^T delta1
    else if (tid >= 156) gbx += row_sum(S.a1[tid - 156]);                                    // db1
  }
  if (tid == 0) { lpart[2 * blockIdx.x] = lsum0; lpart[2 * blockIdx.x + 1] = lsum1; }
  if (MODE == MODE_LOSS) return;
  float* out = partial + (size_t)blockIdx.x * NPAD;
  if (tid < 140) {
    const int i0 = (tid / 10) * 4, j0 = (tid % 10) * 10;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 10; b++) out[O_W1 + (i0 + a) * H + j0 + b] = gW1[a][b];
  }
  if (tid < 250) {
    const int i0 = (tid / 10) * 4, j0 = (tid % 10) * 10;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 10; b++) out[O_W2 + (i0 + a) * H + j0 + b] = gW2[a][b];
  }
  if (tid < 100) {
    const int i0 = (tid / 4) * 4, j0 = (tid % 4) * 7;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 7; b++) out[O_W3 + (i0 + a) * AC + j0 + b] = gW3[a][b];
    out[O_B2 + tid] = gb2;
  } else if (tid < 100 + AC) out[O_B3 + tid - 100] = gbx;
  else if (tid >= 128 && tid < 128 + AC) out[O_LS + tid - 128] = gbx;       // (Fisher product: zero here; the reduction writes 2 v)
  else if (tid >= 156) out[O_B1 + tid - 156] = gbx;
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
