// policy_kernel.h — the learner's policy step for a batch of environments as ONE kernel:
//   obz = clip((ob - mean) / std, -5, 5);  two 2x100 tanh MLPs (policy mean [28], value [1]);  ac = mean + exp(logstd) * N(0,1)
// (src/mlp_policy_trpo.py:35-58, src/distributions.py:220-245).  It replaces ~15 launch-bound library calls per rollout step.
// The matrices are tiny (56x100, 100x100, 100x28): a workgroup takes 16 environments (256 workgroups at 4096 envs: one per CU; 32 per workgroup left half the chip idle: 39 -> 22 us), keeps their activations in LDS and
// streams the weights once per workgroup from L2 (coalesced across the 200 hidden-unit threads); every LDS read feeds four
// FMAs.  fp32 like the reference's TF graph.  No MFMA: 37 k MACs per env would not amortise a fragment layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dmp {

constexpr int OB = 56, HID = 100, AC = 28, EB = 16, HS = 204;   // HS: padded row stride of the hidden activations (bank spread)
// packed weight layout (floats)
constexpr int O_MEAN = 0, O_STD = O_MEAN + OB;
constexpr int O_PW1 = O_STD + OB, O_PB1 = O_PW1 + OB * HID, O_PW2 = O_PB1 + HID, O_PB2 = O_PW2 + HID * HID;
constexpr int O_PW3 = O_PB2 + HID, O_PB3 = O_PW3 + HID * AC, O_LOGSTD = O_PB3 + AC;
constexpr int O_VW1 = O_LOGSTD + AC, O_VB1 = O_VW1 + OB * HID, O_VW2 = O_VB1 + HID, O_VB2 = O_VW2 + HID * HID;
constexpr int O_VW3 = O_VB2 + HID, O_VB3 = O_VW3 + HID, N_WEIGHTS = O_VB3 + 1;

__device__ inline unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// standard normal from a counter: two 24-bit uniforms, Box-Muller
__device__ inline float normal_from(unsigned long long seed, unsigned long long counter, unsigned idx) {
  const unsigned long long h = mix64(mix64(seed ^ (counter * 0xD1342543DE82EF95ull)) + idx);
  const float u1 = ((float)((h >> 40) & 0xFFFFFF) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
  const float u2 = (float)((h >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// one dense layer for EB environments: thread u (< 2 * HID) owns hidden unit (net = u / HID, j = u % HID)
template <int K>
__device__ inline void dense32(const float* __restrict__ W, const float* __restrict__ bias, int j, const float* in, int in_stride,
                               int in_off, float* acc) {
  const float b = bias[j];
#pragma unroll
  for (int e = 0; e < EB; e++) acc[e] = b;
  for (int k = 0; k < K; k += 4) {
    const float w0 = W[(k + 0) * HID + j], w1 = W[(k + 1) * HID + j], w2 = W[(k + 2) * HID + j], w3 = W[(k + 3) * HID + j];
#pragma unroll
    for (int e = 0; e < EB; e++) {
      const float4 x = *reinterpret_cast<const float4*>(in + e * in_stride + in_off + k);   // same address in every lane: broadcast
      acc[e] += x.x * w0 + x.y * w1 + x.z * w2 + x.w * w3;
    }
  }
}

#ifndef DM_NO_LAUNCH_KERNELS        // (a translation unit that only uses the in-wave policy steps defines it: kernels_packed.hip)
__global__ __launch_bounds__(256) void k_policy_act(const float* __restrict__ P, const double* __restrict__ obs, double* __restrict__ action,
                                                    float* __restrict__ vpred, int n, int stochastic, unsigned long long seed,
                                                    unsigned long long counter) {
  __shared__ __attribute__((aligned(16))) float z[EB * OB];
  __shared__ __attribute__((aligned(16))) float h1[EB * HS];
  __shared__ __attribute__((aligned(16))) float h2[EB * HS];
  const int tid = threadIdx.x, e0 = blockIdx.x * EB;
  for (int i = tid; i < EB * OB; i += 256) {
    const int e = i / OB, k = i % OB, env = e0 + e;
    float v = 0.0f;
    if (env < n) { v = ((float)obs[(size_t)env * OB + k] - P[O_MEAN + k]) / P[O_STD + k]; v = fminf(fmaxf(v, -5.0f), 5.0f); }
    z[i] = v;
  }
  __syncthreads();
  float acc[EB];
  const int net = tid / HID, j = tid % HID;                    // net 0: policy, 1: value (threads >= 200 idle in the hidden layers)
  if (tid < 2 * HID) {
    dense32<OB>(P + (net ? O_VW1 : O_PW1), P + (net ? O_VB1 : O_PB1), j, z, OB, 0, acc);
#pragma unroll
    for (int e = 0; e < EB; e++) h1[e * HS + tid] = tanhf(acc[e]);
  }
  __syncthreads();
  if (tid < 2 * HID) {
    dense32<HID>(P + (net ? O_VW2 : O_PW2), P + (net ? O_VB2 : O_PB2), j, h1, HS, net * HID, acc);
#pragma unroll
    for (int e = 0; e < EB; e++) h2[e * HS + tid] = tanhf(acc[e]);
  }
  __syncthreads();
  // output layer: (env, output) pairs over all 256 threads; outputs 0..27 = action means, 28 = value
  const int e = tid % EB, env = e0 + e;
  for (int o = tid / EB; o < AC + 1; o += 256 / EB) {
    const bool isv = o == AC;
    const float* W = P + (isv ? O_VW3 : O_PW3);
    const int ws = isv ? 1 : AC, wo = isv ? 0 : o;
    const float* h = h2 + e * HS + (isv ? HID : 0);
    float s = isv ? P[O_VB3] : P[O_PB3 + o];
    for (int k = 0; k < HID; k += 4) {
      const float4 x = *reinterpret_cast<const float4*>(h + k);
      s += x.x * W[(k + 0) * ws + wo] + x.y * W[(k + 1) * ws + wo] + x.z * W[(k + 2) * ws + wo] + x.w * W[(k + 3) * ws + wo];
    }
    if (env < n) {
      if (isv) vpred[env] = s;
      else {
        if (stochastic) s += expf(P[O_LOGSTD + o]) * normal_from(seed, counter, (unsigned)(env * AC + o));
        action[(size_t)env * AC + o] = (double)s;
      }
    }
  }
}
#endif

// The same policy step for ONE environment by ONE wave, as the epilogue of the env step kernel (dm_batch_step_act): the wave that
// has just produced an observation turns it into the next action and its value estimate before it exits, so a rollout step is one
// launch and consecutive steps of a pipelined batch overlap exactly like open-loop stepping (the separate launch sat on every
// step's critical path: policy 20..40 us + launch gaps against an env step of ~310 us).  Lane l < 50 owns four consecutive hidden
// units (lanes 0..24 the policy net, 25..49 the value net): one 16-byte weight load per input and lane, 14 / 20 of them in flight —
// every wave streams the 87 KB of weights from L2 itself, so the epilogue is bound by those loads' latency, not by its ~1 k FMAs;
// activations go through `scr` (LDS, >= 464 floats).
struct PolicyArgs {
  const float* P;            // packed weights (layout above); nullptr: no policy step
  double* action;            // [N, 28] out: action for the observation just produced
  float* vpred;              // [N] out
  int stochastic;
  unsigned long long seed, counter;
};
template <int K, int UNROLL>
__device__ inline float4 dense4_wave(const float* __restrict__ W, const float* __restrict__ bias, const float* in) {
  // four consecutive hidden units of one net: one 16-byte weight load per input (coalesced across lanes), UNROLL of them in flight
  float4 acc = *reinterpret_cast<const float4*>(bias);
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += UNROLL) {
    float4 w[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (k0 + u < K) w[u] = *reinterpret_cast<const float4*>(W + (size_t)(k0 + u) * HID);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (k0 + u < K) {
      const float x = in[k0 + u];
      acc.x += x * w[u].x; acc.y += x * w[u].y; acc.z += x * w[u].z; acc.w += x * w[u].w;
    }
  }
  return acc;
}
template <class T>
__device__ inline void policy_wave(const PolicyArgs& pa, int env, int lane, const T* ob_q /* qpos + 7: 28 */, const T* ob_v /* qvel + 6: 28 */, float* scr) {
  const float* __restrict__ P = pa.P;
  float* z = scr; float* h1 = scr + 64; float* h2 = scr + 264;
  if (lane < OB) {
    const float o = lane < 28 ? (float)ob_q[lane] : (float)ob_v[lane - 28];
    z[lane] = fminf(fmaxf((o - P[O_MEAN + lane]) / P[O_STD + lane], -5.0f), 5.0f);
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
  // lanes 0..24: units 4 l .. 4 l + 3 of the policy net; lanes 25..49: of the value net
  const bool hid = lane < 2 * (HID / 4);
  const int net = lane >= HID / 4 ? 1 : 0, j4 = 4 * (lane - net * (HID / 4));
  const int jj = hid ? j4 : 0;
  {
    const float4 a = dense4_wave<OB, 14>(P + (net ? O_VW1 : O_PW1) + jj, P + (net ? O_VB1 : O_PB1) + jj, z);
    if (hid) *reinterpret_cast<float4*>(h1 + net * HID + j4) = make_float4(tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w));
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
  {
    const float4 a = dense4_wave<HID, 20>(P + (net ? O_VW2 : O_PW2) + jj, P + (net ? O_VB2 : O_PB2) + jj, h1 + net * HID);
    if (hid) *reinterpret_cast<float4*>(h2 + net * HID + j4) = make_float4(tanhf(a.x), tanhf(a.y), tanhf(a.z), tanhf(a.w));
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
  // output layer: lanes 0..27 the action means, lane 28 the value
  {
    const bool isv = lane == AC, on = lane <= AC;
    const float* W = P + (isv ? O_VW3 : O_PW3);
    const int ws = isv ? 1 : AC, wo = (isv || !on) ? 0 : lane;
    const float* h = h2 + (isv ? HID : 0);
    float sacc = isv ? P[O_VB3] : P[O_PB3 + wo];
#pragma unroll 1
    for (int k0 = 0; k0 < HID; k0 += 20) {
      float w[20];
#pragma unroll
      for (int u = 0; u < 20; u++) w[u] = W[(k0 + u) * ws + wo];
#pragma unroll
      for (int u = 0; u < 20; u++) sacc += h[k0 + u] * w[u];
    }
    if (isv) pa.vpred[env] = sacc;
    else if (on) {
      if (pa.stochastic) sacc += expf(P[O_LOGSTD + lane]) * normal_from(pa.seed, pa.counter, (unsigned)(env * AC + lane));
      pa.action[(size_t)env * AC + lane] = (double)sacc;
    }
  }
}

// (Measured, round 4: the same step on the MATRIX CORE — v_mfma_f32_4x4x1, sixteen 4x4x1 blocks = four environments x 64 output units per
//  instruction, lane = unit, register = environment, weights as [chunk of four inputs][lane][4] panels (tools/ubench/mfma_f32_4x4.hip has the lane
//  map) — 824 matrix instructions + 206 loads instead of 2 900 multiply-adds + 1 000 LDS reads + 256 loads, all 33 rollout tests green: exactly as
//  fast (rollout 14.1 M, standing 12.9–13.0 M either way).  The step is bound by each wave streaming the weights through its CU's L1 — 87 KB here,
//  211 KB as 64-unit panels — not by the arithmetic; dropped.  profiles/r04_ab_kernel_variants.md)
// ... and for the FOUR environments of a packed wave (slot_kernel.h) at once: the same lane -> hidden-unit map, every 16-byte weight load
// now feeds 16 multiply-adds (four environments x four units), so the weight stream that dominated the one-env epilogue is paid once
// per four environments.  in0..in3 / scratch: one block of >= 464 floats per environment (z[64] | h1[200] | h2[200]).
template <int K, int UNROLL>
__device__ inline void dense4_wave4(const float* __restrict__ W, const float* __restrict__ bias, const float* (&in)[4], float4 (&acc)[4]) {
  const float4 b = *reinterpret_cast<const float4*>(bias);
#pragma unroll
  for (int e = 0; e < 4; e++) acc[e] = b;
#pragma unroll 1
  for (int k0 = 0; k0 < K; k0 += UNROLL) {
    float4 w[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (k0 + u < K) w[u] = *reinterpret_cast<const float4*>(W + (size_t)(k0 + u) * HID);
#pragma unroll
    for (int u = 0; u < UNROLL; u++) if (k0 + u < K) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float x = in[e][k0 + u];
        acc[e].x += x * w[u].x; acc[e].y += x * w[u].y; acc[e].z += x * w[u].z; acc[e].w += x * w[u].w;
      }
    }
  }
}
// env[e] / write[e]: the environment of slot e and whether its outputs are stored.  The four slots' blocks lie `stride` bytes apart from
// `slot0` (LDS); inside a block: observation halves at off_q (qpos + 7: 28 T) and off_v (qvel + 6: 28 T), scratch at off_scr.
template <class T>
__device__ inline void policy_wave4(const PolicyArgs& pa, const int (&env)[4], const bool (&write)[4], int lane, char* slot0, unsigned stride, unsigned off_q, unsigned off_v, unsigned off_scr) {
  const float* __restrict__ P = pa.P;
  const T* ob_q[4]; const T* ob_v[4]; float* scr[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    ob_q[e] = reinterpret_cast<const T*>(slot0 + e * stride + off_q); ob_v[e] = reinterpret_cast<const T*>(slot0 + e * stride + off_v);
    scr[e] = reinterpret_cast<float*>(slot0 + e * stride + off_scr);
  }
  {   // every slot normalises its own observation: slot lane sl takes inputs sl, sl + 16, ...
    const int slot = lane >> 4, sl = lane & 15;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int i = sl + 16 * c;
      if (i < OB) {
        const float o = i < 28 ? (float)ob_q[slot][i] : (float)ob_v[slot][i - 28];
        scr[slot][i] = fminf(fmaxf((o - P[O_MEAN + i]) / P[O_STD + i], -5.0f), 5.0f);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
  const bool hid = lane < 2 * (HID / 4);
  const int net = lane >= HID / 4 ? 1 : 0, j4 = 4 * (lane - net * (HID / 4));
  const int jj = hid ? j4 : 0;
  {
    const float* in[4] = {scr[0], scr[1], scr[2], scr[3]};
    float4 a[4];
    dense4_wave4<OB, 14>(P + (net ? O_VW1 : O_PW1) + jj, P + (net ? O_VB1 : O_PB1) + jj, in, a);
    if (hid) {
#pragma unroll
      for (int e = 0; e < 4; e++) *reinterpret_cast<float4*>(scr[e] + 64 + net * HID + j4) = make_float4(tanhf(a[e].x), tanhf(a[e].y), tanhf(a[e].z), tanhf(a[e].w));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
  {
    const float* in[4] = {scr[0] + 64 + net * HID, scr[1] + 64 + net * HID, scr[2] + 64 + net * HID, scr[3] + 64 + net * HID};
    float4 a[4];
    dense4_wave4<HID, 20>(P + (net ? O_VW2 : O_PW2) + jj, P + (net ? O_VB2 : O_PB2) + jj, in, a);
    if (hid) {
#pragma unroll
      for (int e = 0; e < 4; e++) *reinterpret_cast<float4*>(scr[e] + 264 + net * HID + j4) = make_float4(tanhf(a[e].x), tanhf(a[e].y), tanhf(a[e].z), tanhf(a[e].w));
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
  {
    const bool isv = lane == AC, on = lane <= AC;
    const float* W = P + (isv ? O_VW3 : O_PW3);
    const int ws = isv ? 1 : AC, wo = (isv || !on) ? 0 : lane;
    const int ho = 264 + (isv ? HID : 0);
    const float b = isv ? P[O_VB3] : P[O_PB3 + wo];
    float sacc[4] = {b, b, b, b};
#pragma unroll 1
    for (int k0 = 0; k0 < HID; k0 += 20) {
      float w[20];
#pragma unroll
      for (int u = 0; u < 20; u++) w[u] = W[(k0 + u) * ws + wo];
#pragma unroll
      for (int u = 0; u < 20; u++) {
#pragma unroll
        for (int e = 0; e < 4; e++) sacc[e] += scr[e][ho + k0 + u] * w[u];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (!write[e]) continue;
      if (isv) pa.vpred[env[e]] = sacc[e];
      else if (on) {
        float a = sacc[e];
        if (pa.stochastic) a += expf(P[O_LOGSTD + lane]) * normal_from(pa.seed, pa.counter, (unsigned)(env[e] * AC + lane));
        pa.action[(size_t)env[e] * AC + lane] = (double)a;
      }
    }
  }
}

#ifndef DM_NO_LAUNCH_KERNELS
// GAE(lambda) of src/trpo.py:83-94 for N environments: thread = env, a backward loop over the T rows of the [T, N] segment
// (coalesced across envs), instead of T small launches.
__global__ __launch_bounds__(256) void k_gae(const float* __restrict__ rew, const float* __restrict__ vpred, const int* __restrict__ isnew,
                                             const float* __restrict__ nextvpred, float* __restrict__ adv, float* __restrict__ tdlamret,
                                             int T, int n, float gamma, float lam) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  float last = 0.0f, vnext = nextvpred[e];
  float nonterminal = 1.0f;                                   // new[T] := 0
  for (int t = T - 1; t >= 0; t--) {
    const size_t i = (size_t)t * n + e;
    const float v = vpred[i];
    const float delta = rew[i] + gamma * vnext * nonterminal - v;
    last = delta + gamma * lam * nonterminal * last;
    adv[i] = last; tdlamret[i] = last + v;
    vnext = v; nonterminal = 1.0f - (float)isnew[i];         // for row t - 1: new[t]
  }
}
// Episode bookkeeping of the generator (src/trpo.py:68-79: `cur_ep_ret += rew; cur_ep_len += 1; if new: ep_rets.append(...)`) for N environments
// over a [T, N] segment: thread = env walks its column, adds rewards in float64 in step order (the sums a per-env host loop would form) and
// appends a record {t << 32 | env, return (bits), length} of every episode that ends to a list — in arrival order; the host sorts the few
// records by their first word (t, then env).  The open episode's return / length are carried in cur_ret / cur_len to the next segment.
__global__ __launch_bounds__(256) void k_episodes(const double* __restrict__ rew, const uint8_t* __restrict__ done, int T, int n, double* __restrict__ cur_ret,
                                                  long long* __restrict__ cur_len, int* __restrict__ count, int cap, long long* __restrict__ rec) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  double r = cur_ret[e];
  long long len = cur_len[e];
  for (int t0 = 0; t0 < T; t0 += 8) {
    double x[8]; uint8_t d[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { const bool in = t0 + u < T; const size_t i = (size_t)(in ? t0 + u : 0) * n + e; x[u] = in ? rew[i] : 0.0; d[u] = in ? done[i] : 0; }   // eight rows in flight
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (t0 + u >= T) break;
      r += x[u]; len += 1;
      if (d[u]) {
        const int slot = atomicAdd(count, 1);
        if (slot < cap) { rec[3 * (size_t)slot] = ((long long)(t0 + u) << 32) | (long long)e; rec[3 * (size_t)slot + 1] = __double_as_longlong(r); rec[3 * (size_t)slot + 2] = len; }
        r = 0.0; len = 0;
      }
    }
  }
  cur_ret[e] = r; cur_len[e] = len;
}
#endif

}  // namespace dmp
