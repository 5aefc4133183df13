// dmenv.hip — libdmenv.so: HIP kernels (gfx950) + the C ABI of include/dmenv.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I../../include dmenv.hip -o libdmenv.so
// There is no CPU execution path in this library: every entry point that computes runs a HIP kernel.
#include <cstddef>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "kernels.h"
#include "vf_kernel.h"
#include "pg_kernel.h"
#include "model_host.h"

using namespace dm;
static_assert((DM_PACKED_MAXROWS == SLOT_MAXROWS || DM_SLOT_MAXROWS != 40 /* an experiment build */) && DM_PACKED_MAXROWS_PER_STEP == 2 * SW && DM_PACKED_MAXLIMROWS == SLOT_MAXLIMROWS && DM_PACKED_MAXCON == SLOT_MAXCON && DM_PACKED_MAXFRAME == SLOT_MAXFRAME &&
              DM_PACKED_MAXCAND == SLOT_MAXCAND, "include/dmenv.h documents the packed path's capacities: keep it in step with slot_kernel.h");
// ============================================ kernels ======================================================
// one 64-lane workgroup (= one wavefront) per environment.
// k_step_narrow (the timed path) keeps NARROW_ROWS columns of the constraint matrix A per env in registers, which fits
// 256 VGPRs -> 2 waves per SIMD; evaluations with more rows (up to MAXEFC = 64; < 1% in practice) keep the remaining
// columns in a per-env global-memory strip.  k_step holds all 64 columns in registers (512 VGPRs, 1 wave per SIMD) and is
// the single-tier fallback (DM option 102 = 0) and the profiling / debug instantiation.  Both perform identical
// arithmetic on the rows that exist.
static_assert(AOVF_COLS >= MAXEFC, "memory strip too small");
__global__ __launch_bounds__(64, DM_STEP_WAVES) void k_step_narrow(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                                    Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                                    int n_substeps, int first, int count) {
  __shared__ Shared<Real> s;
  __shared__ StepScratch<Real> x;
#ifdef DM_LDS_PAD      // experiment (profiles/r02_slot16_ubench.md): extra LDS per workgroup lowers the residency (6 880 -> 6 envs per CU,
  __shared__ volatile char lds_pad[DM_LDS_PAD];   // 20 500 -> 4) without touching the code path; build with DM_BUILD_DEFINES=-DDM_LDS_PAD=...
  if (n_substeps < 0) lds_pad[threadIdx.x] = 1;
#endif
  // position blockIdx.x in the dispatch order of the part [first, first + count) (a pipelined sub-batch, or the whole batch)
  if ((int)blockIdx.x >= count) return;
  int env;
  dispatch_env<1>(B, first, count, (int)blockIdx.x, dmw::lane(), blockIdx.x == 0, &env);
  env_step<Real, NARROW_ROWS>(*Mp, B, s, x, env, dmw::lane(), action, obs, reward, done, n_substeps);
}
// the same step followed, in the same wave, by the policy's step on the observation it produced (dm_batch_step_act)
__global__ __launch_bounds__(64, DM_STEP_WAVES) void k_step_act(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                                 Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                                 int n_substeps, int first, int count, dmp::PolicyArgs pa) {
  __shared__ Shared<Real> s;
  __shared__ StepScratch<Real> x;
  if ((int)blockIdx.x >= count) return;
  int env;
  dispatch_env<1>(B, first, count, (int)blockIdx.x, dmw::lane(), blockIdx.x == 0, &env);
  env_step<Real, NARROW_ROWS>(*Mp, B, s, x, env, dmw::lane(), action, obs, reward, done, n_substeps);
  // s.qpos / s.qvel hold the state the observation was written from (the fresh episode's after an auto-reset); the row-descriptor
  // region is free
  static_assert(sizeof(s.u) >= 464 * sizeof(float), "policy scratch");
  dmw::sync();
  dmp::policy_wave(pa, env, dmw::lane(), &s.qpos[7], &s.qvel[6], reinterpret_cast<float*>(&s.u));
}
__global__ __launch_bounds__(64) void k_step(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                             Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                             int n_substeps) {
  __shared__ Shared<Real> s;
  __shared__ StepScratch<Real> x;
  const int env = blockIdx.x;
  if (env >= B.n_envs) return;
  env_step<Real, MAXEFC>(*Mp, B, s, x, env, dmw::lane(), action, obs, reward, done, n_substeps);
}

// the batch descriptor into device memory, stream-ordered before the horizon launch that reads it there
__global__ void k_put_batch(Batch<Real> B, Batch<Real>* __restrict__ dst) { if (threadIdx.x == 0 && blockIdx.x == 0) *dst = B; }
// the horizon's table of per-step buffers (slot_step.h StepRow), stream-ordered before the launch that reads it: rows of the caller's
// [T, N, .] tensors (dm_batch_rollout) ...
__global__ void k_fill_rows(StepRow* __restrict__ dst, const Ext* action, Ext* obs, Ext* reward, unsigned char* done, int T, size_t n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < T) dst[t] = StepRow{action + (size_t)t * n * NU, obs + (size_t)t * n * NOBS, reward + (size_t)t * n, done + (size_t)t * n};
}
// ... or the buffers of queued dm_batch_step calls (DM_OPT_STEP_QUEUE), a chunk of them per launch in the kernel-argument segment
__global__ void k_put_rows(StepRowChunk c, StepRow* __restrict__ dst, int count) { const int t = threadIdx.x; if (t < count) dst[t] = c.r[t]; }
// ... and stepped here, from their unchanged state, by the one-env code (a handful of persistent single-wave workgroups walk the list)
__global__ __launch_bounds__(64, DM_STEP_WAVES) void k_step_redo(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                                  Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                                  int n_substeps, int first, const int* __restrict__ redo_count, int* __restrict__ redo_count_next, dmp::PolicyArgs pa) {
  __shared__ Shared<Real> s;
  __shared__ StepScratch<Real> x;
  const int n = dmw::uniform(*redo_count);
  // the two counters of a sub-batch alternate from step to step: this launch clears the one the NEXT packed launch will count into (its last
  // reader, the redo launch of the step before, finished before this step's packed launch began: stream order) — no memset between launches
  if (blockIdx.x == 0 && threadIdx.x == 0) *redo_count_next = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0 && n > 0) atomicAdd(B.redo_why, n);       // running total (dm_batch_redo_total)
  for (int i = (int)blockIdx.x; i < n; i += (int)gridDim.x) {
    const int env = B.redo_list[first + i];
    env_step<Real, NARROW_ROWS>(*Mp, B, s, x, env, dmw::lane(), action, obs, reward, done, n_substeps);
    if (pa.P) { dmw::sync(); dmp::policy_wave(pa, env, dmw::lane(), &s.qpos[7], &s.qvel[6], reinterpret_cast<float*>(&s.u)); }
    dmw::sync_mem();
  }
}

// Dispatch order of a HORIZON launch (k_rollout_packed): environments with similar cost keys (env_step.h DM_ORDER_KEY: constraint rows + a quarter
// of the PGS sweeps of the last evaluation) share a wave — a wave pays for its largest row count and its slowest PGS.  Counting sort, descending;
// the order inside a bucket is arbitrary — results never depend on the dispatch order.  One launch per horizon: off the per-step path, where the
// step kernels order themselves (env_step.h dispatch_env / order_ticket; rounds 1-4 ran this kernel — or its one-wave form — between every two steps:
// 9 % of the GPU time of the one-launch-per-call form).
// (measured, round 4: DEALING the sorted list over the waves of a horizon launch — rank r to wave r % W, slot r / W, one environment of each
//  quartile per wave instead of the four heaviest together — shortens the slowest wave of a 64-step launch by 1.3 % and lengthens the
//  driver's 20-step window by 2 %: profiles/r04_ab_kernel_variants.md; the grouped order stays)
#ifndef DM_GROUP_KEY
#define DM_GROUP_KEY(nefc, iter) DM_ORDER_KEY(nefc, iter)
#endif
#ifndef DM_GROUP_BUCKETS
#define DM_GROUP_BUCKETS 64
#endif
DM_DEV int group_bucket(int nefc, int iter) { const int k = DM_GROUP_KEY(nefc, iter); return k < 0 ? 0 : (k > DM_GROUP_BUCKETS - 1 ? DM_GROUP_BUCKETS - 1 : k); }
__global__ __launch_bounds__(1024) void k_order(Batch<Real> B, int* __restrict__ order, int first, int count) {
  constexpr int NBK = DM_GROUP_BUCKETS;
  static_assert(NBK <= 1024, "one thread per bucket");
  __shared__ int hist[NBK], start[NBK];
  const int tid = threadIdx.x, n = count;                // envs first .. first + count - 1 are sorted into order[first ..]
  if (tid < NBK) hist[tid] = 0;
  __syncthreads();
  constexpr int PER = 8;                 // keys of up to 8192 envs stay in registers between the two passes
  int key[PER];
#pragma unroll
  for (int j = 0; j < PER; j++) {
    const int e = tid + j * 1024;
    key[j] = -1;
    if (e < n) { key[j] = group_bucket(B.nefc[first + e], B.solver_iter[first + e]); atomicAdd(&hist[key[j]], 1); }
  }
  for (int e = tid + PER * 1024; e < n; e += 1024) atomicAdd(&hist[group_bucket(B.nefc[first + e], B.solver_iter[first + e])], 1);
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int k = NBK - 1; k >= 0; k--) { start[k] = acc; acc += hist[k]; } }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; j++) if (key[j] >= 0) order[first + atomicAdd(&start[key[j]], 1)] = first + tid + j * 1024;
  for (int e = tid + PER * 1024; e < n; e += 1024) order[first + atomicAdd(&start[group_bucket(B.nefc[first + e], B.solver_iter[first + e])], 1)] = first + e;
}

// same step with per-stage shader-clock accounting (DM_OPT 101); not used on the timed path
__global__ __launch_bounds__(64) void k_step_prof(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                                  Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                                  int n_substeps, long long* prof) {
  __shared__ Shared<Real> s;
  __shared__ StepScratch<Real> x;
  const int env = blockIdx.x;
  if (env >= B.n_envs) return;
  env_step<Real, MAXEFC, true>(*Mp, B, s, x, env, dmw::lane(), action, obs, reward, done, n_substeps, prof);
}

__global__ __launch_bounds__(64) void k_set_state(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ qpos,
                                                  const Ext* __restrict__ qvel, const int* __restrict__ frame_idx,
                                                  const unsigned char* __restrict__ mask) {
  __shared__ Shared<Real> s;
  const int env = blockIdx.x, lane = dmw::lane();
  if (env >= B.n_envs) return;
  if (mask && !mask[env]) return;
  load_env(*Mp, B, s, env, lane, (const Ext*)0);
  if (lane < NQ) s.qpos[lane] = (Real)qpos[(size_t)env * NQ + lane];
  if (lane < NV) s.qvel[lane] = (Real)qvel[(size_t)env * NV + lane];
  if (frame_idx && lane == 0) set_frame(B, env, frame_idx[env]);
  dmw::sync();
  store_state(B, s, env, lane);
  { const LaneTopo lt = lane_topo(lane); stage_tables(s, lane); dmw::sync(); forward(*Mp, s, lane, lt, (const DebugOut*)0); }   // sim.forward()
  store_derived(B, *Mp, s, env, lane);
}

__global__ __launch_bounds__(64) void k_reset(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, int mode, int hard,
                                              const unsigned char* __restrict__ mask) {
  __shared__ Shared<Real> s;
  const int env = blockIdx.x, lane = dmw::lane();
  if (env >= B.n_envs) return;
  if (mask && !mask[env]) return;
  load_env(*Mp, B, s, env, lane, (const Ext*)0);
  reset_env(*Mp, B, s, env, lane, mode, hard);
  store_state(B, s, env, lane);
  { const LaneTopo lt = lane_topo(lane); stage_tables(s, lane); dmw::sync(); forward(*Mp, s, lane, lt, (const DebugOut*)0); }
  store_derived(B, *Mp, s, env, lane);
}

__global__ void k_get_obs(Batch<Real> B, Ext* __restrict__ obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B.n_envs * NOBS) return;
  const int env = i / NOBS, k = i % NOBS;
  obs[i] = k < 28 ? B.qpos[(size_t)env * NQ + 7 + k] : B.qvel[(size_t)env * NV + 6 + (k - 28)];
}

// field reads / writes of the float32 build: device state is `Real`, the C ABI speaks float64
__global__ void k_to_ext(const Real* __restrict__ src, Ext* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (Ext)src[i];
}
__global__ void k_from_ext(const Ext* __restrict__ src, Real* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (Real)src[i];
}

__global__ __launch_bounds__(64) void k_debug_forward(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, int env, double* out) {
  __shared__ Shared<Real> s;
  const int lane = dmw::lane();
  load_env(*Mp, B, s, env, lane, (const Ext*)0);
  // actuator forces from the stored ctrl
  if (lane < NU) { const int d = lane + 6; s.act[d] = Mp->gear[d] * clampr(B.ctrl[(size_t)env * NU + lane], Mp->ctrl_lo[d], Mp->ctrl_hi[d]); }
  dmw::sync();
  DebugOut dbg{out};
  { const LaneTopo lt = lane_topo(lane); stage_tables(s, lane); dmw::sync(); forward(*Mp, s, lane, lt, &dbg); }
  store_derived(B, *Mp, s, env, lane);
}

// ============================================ host side ====================================================
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(DM_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

struct dm_model { DevModel<Real> h; };
struct dm_mocap { std::vector<double> cfg, vel, imit_table, imit_params; int n_frames; double dt; };
struct dm_batch {
  int n = 0, device = 0;
  hipStream_t stream = nullptr; bool own_stream = false;
  DevModel<Real>* d_model = nullptr;
  Batch<Real> B{};
  Batch<Real>* d_B = nullptr;   // a copy of B in device memory for the horizon launch (refreshed before each: its code is reached through calls, which take a pointer)
  Real *d_cfg = nullptr, *d_vel = nullptr, *d_imit = nullptr; int* d_order = nullptr;
  // self-ordering per-step launches (env_step.h dispatch_env): per pipelined part three phases of 64 bucket counters, and three phases of bucket
  // lists ([phase][bucket][n], a part's entries at its first env); ord_phase = the phase the part's last launch counted into, valid once one did
  int* d_ord_cnt = nullptr; int* d_ord_list = nullptr; int ord_phase[DM_MAX_PIPELINE] = {}; bool ord_valid[DM_MAX_PIPELINE] = {};
  // staging for DM_PTR_HOST callers
  Ext *d_action = nullptr, *d_obs = nullptr, *d_reward = nullptr; unsigned char *d_done = nullptr, *d_mask = nullptr;
  // host-pointer steps: obs | reward | done are ONE device block (d_obs points at its start) mirrored in pinned host memory, so a
  // step costs one H2D (action, from the pinned mirror) and one D2H instead of one pageable copy per array
  unsigned char* h_out = nullptr; Ext* h_action = nullptr; size_t out_bytes = 0;
  Ext* d_cvt = nullptr;   // float32 build: float64 staging for field reads / writes through host pointers
  Ext *d_qpos_in = nullptr, *d_qvel_in = nullptr; int* d_fidx_in = nullptr;
  double* d_debug = nullptr;
  long long* d_prof = nullptr; bool prof = false;
  int redo_phase = 0;    // which of a sub-batch's two redo counters the next packed launch counts into
  int redo_mode = -1;    // 1 / 0: the last packed step was / was not pipelined (the counter pairs are re-zeroed when that changes)
  // DM_OPT_STEP_QUEUE: dm_batch_step calls with device pointers are queued (nothing is launched) and executed together — one horizon launch,
  // every wave at its own pace — when the queue is full or any other entry point of the batch is called (dm_batch_join, ...)
  int queue_cap = 0; std::vector<StepRow> q; int q_nsub = 1; StepRow* d_rows = nullptr; int rows_cap = 0; long long queue_flushes = 0, queue_steps = 0;
  int horizon_mode = -1; // option 106: dm_batch_rollout on the packed path as ONE launch per horizon (1), as step launches (0), by batch size (-1, default)
  bool packed = false;   // option 105: four environments per wavefront (k_step_packed) where that kernel covers the configuration
  bool packed_ext = false;   // DM_OPT_PACKED = 2: per-step packed launches with the three-set code (k_step_packed_ext: 40 rows per env, ~8 % slower otherwise)
  bool two_tier = true, reorder = true, has_rows = true; int resident_waves = 2048;   // CUs x 8 single-wave workgroups (LDS-limited)
  bool timing = false; hipEvent_t ev0 = nullptr, ev1 = nullptr; float last_ms = 0.f; bool ev_pending = false;
  // pipelined sub-batches (DM_OPT_PIPELINE): the env range is cut into `pipe` contiguous parts, each stepped on its own stream
  int pipe = 1; hipStream_t ps[DM_MAX_PIPELINE] = {}; hipEvent_t ev_in = nullptr, ev_done[DM_MAX_PIPELINE] = {}; bool pipe_pending = false;
};
// make the batch's stream wait for every sub-batch launch still in flight (no host wait)
static int pipe_join(dm_batch* b) {
  if (!b->pipe_pending) return DM_OK;
  for (int h = 0; h < b->pipe; h++) if (hipStreamWaitEvent(b->stream, b->ev_done[h], 0) != hipSuccess) return DM_EHIP;
  b->pipe_pending = false;
  return DM_OK;
}

static int flush_queue(dm_batch* b);
// what every entry point other than a queued dm_batch_step does first: run the queued steps, then make the batch's stream wait for the sub-batch launches
static int settle(dm_batch* b) {
  if (!b->q.empty()) { const int rc = flush_queue(b); if (rc != DM_OK) return rc; }
  return pipe_join(b);
}

extern "C" const char* dm_last_error(void) { return g_err.c_str(); }
extern "C" int dm_abi_version(void) { return DM_ABI_VERSION; }
extern "C" int dm_real_bits(void) { return (int)(8 * sizeof(Real)); }   /* 64: libdmenv.so, 32: libdmenv32.so */
extern "C" int dm_device_count(void) { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

extern "C" int dm_model_create(const dm_model_desc* d, dm_model** out) {
  if (!d || !out) return fail(DM_EINVAL, "dm_model_create: null argument");
  dm_model* m = new (std::nothrow) dm_model();
  if (!m) return fail(DM_ENOMEM, "dm_model_create: out of memory");
  std::string err;
  const int rc = dm::build_dev_model(d, &m->h, &err);
  if (rc != DM_OK) { delete m; return fail(rc, err); }
  *out = m;
  return DM_OK;
}
extern "C" void dm_model_destroy(dm_model* m) { delete m; }

extern "C" int dm_mocap_create(const double* cfg, const double* vel, int32_t F, double dt, dm_mocap** out) {
  if (!cfg || !vel || F <= 0 || !out) return fail(DM_EINVAL, "dm_mocap_create: bad argument");
  dm_mocap* mc = new (std::nothrow) dm_mocap();
  if (!mc) return fail(DM_ENOMEM, "dm_mocap_create: out of memory");
  mc->cfg.assign(cfg, cfg + (size_t)F * NQ); mc->vel.assign(vel, vel + (size_t)F * NV); mc->n_frames = F; mc->dt = dt;
  *out = mc;
  return DM_OK;
}
extern "C" int dm_mocap_set_imitation(dm_mocap* mc, const double* table, int32_t n_cols, const double* params) {
  if (!mc || !table || !params) return fail(DM_EINVAL, "dm_mocap_set_imitation: null argument");
  if (n_cols != IMIT_FEAT) return fail(DM_EINVAL, "dm_mocap_set_imitation: a feature row has 112 columns");
  for (int e = 0; e < 4; e++) { const int b = (int)params[16 + e]; if (b < 1 || b >= NB) return fail(DM_EINVAL, "dm_mocap_set_imitation: end-effector body id out of range"); }
  mc->imit_table.assign(table, table + (size_t)mc->n_frames * IMIT_FEAT);
  mc->imit_params.assign(params, params + 32);
  return DM_OK;
}
extern "C" void dm_mocap_destroy(dm_mocap* mc) { delete mc; }

template <class T> static hipError_t dalloc(T** p, size_t n) { hipError_t e = hipMalloc((void**)p, n * sizeof(T)); if (e == hipSuccess) e = hipMemset(*p, 0, n * sizeof(T)); return e; }

extern "C" void dm_batch_destroy(dm_batch* b) {
  if (!b) return;
  hipSetDevice(b->device);
  b->q.clear();          // queued steps are dropped, not run: their buffers belong to the caller and may be gone
  pipe_join(b);
  if (b->stream) hipStreamSynchronize(b->stream);
  for (int h = 0; h < DM_MAX_PIPELINE; h++) { if (b->ps[h]) hipStreamDestroy(b->ps[h]); if (b->ev_done[h]) hipEventDestroy(b->ev_done[h]); }
  if (b->ev_in) hipEventDestroy(b->ev_in);
  void* ptrs[] = {b->d_model, b->B.qpos, b->B.qvel, b->B.qws, b->B.time, b->B.ctrl, b->B.xipos, b->B.comz, b->B.frame_idx, b->B.frame_init,
                  b->B.ncon, b->B.nefc, b->B.cong, b->B.status, b->B.solver_iter, b->B.episode, b->d_cfg, b->d_vel, b->d_action, b->d_obs,
                  b->d_mask, b->d_cvt, b->d_qpos_in, b->d_qvel_in, b->d_fidx_in, b->d_debug, b->d_prof, b->B.aovf, b->B.cycle, b->d_imit, b->d_order, b->B.kin, b->B.kin_ok, b->B.redo_list, b->B.redo_count, b->B.redo_why, b->d_B, b->d_rows, b->d_ord_cnt, b->d_ord_list};
  for (void* p : ptrs) if (p) hipFree(p);
  if (b->h_out) hipHostFree(b->h_out);
  if (b->h_action) hipHostFree(b->h_action);
  if (b->ev0) hipEventDestroy(b->ev0);
  if (b->ev1) hipEventDestroy(b->ev1);
  if (b->own_stream && b->stream) hipStreamDestroy(b->stream);
  delete b;
}

extern "C" int dm_batch_create(const dm_model* m, const dm_mocap* mc, int32_t n, int32_t device, uint32_t flags, dm_batch** out) {
  if (!m || !mc || n <= 0 || !out) return fail(DM_EINVAL, "dm_batch_create: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DM_ENODEVICE, "dm_batch_create: no HIP device visible (libdmenv has no CPU path)");
  if (device < 0 || device >= ndev) return fail(DM_EINVAL, "dm_batch_create: bad device id");
  HIPCHK(hipSetDevice(device));
  dm_batch* b = new (std::nothrow) dm_batch();
  if (!b) return fail(DM_ENOMEM, "dm_batch_create: out of memory");
  b->n = n; b->device = device;
  hipError_t e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete b; return fail(DM_EHIP, "hipStreamCreate failed"); }
  b->own_stream = true;
  DevModel<Real> hm = m->h;
  hm.enable_contact = (flags & DM_FLAG_NO_CONTACT) ? 0 : 1;
  hm.enable_limit = (flags & DM_FLAG_NO_LIMIT) ? 0 : 1;
  b->has_rows = hm.enable_contact || hm.enable_limit;     // without rows every env costs the same: no reordering
  bool ok = true;
#define A(p, cnt) ok = ok && (dalloc(&(p), (size_t)(cnt)) == hipSuccess)
  A(b->d_model, 1); A(b->d_B, 1);
  A(b->B.qpos, (size_t)n * NQ); A(b->B.qvel, (size_t)n * NV); A(b->B.qws, (size_t)n * NV); A(b->B.time, n); A(b->B.ctrl, (size_t)n * NU);
  A(b->B.xipos, (size_t)n * NB * 3); A(b->B.comz, n); A(b->B.frame_idx, n); A(b->B.frame_init, n); A(b->B.ncon, n); A(b->B.nefc, n);
  A(b->B.cong, (size_t)n * MAXEFC * 2); A(b->B.status, n); A(b->B.solver_iter, n); A(b->B.episode, n); A(b->B.cycle, n); A(b->d_order, n);
  A(b->d_cfg, (size_t)mc->n_frames * NQ); A(b->d_vel, (size_t)mc->n_frames * NV);
  if (!mc->imit_table.empty()) A(b->d_imit, mc->imit_table.size() + 32);   // [32 parameters][F x 112 feature rows]
  A(b->d_action, (size_t)n * NU); A(b->d_mask, n);
  b->out_bytes = (size_t)n * (NOBS + 1) * sizeof(Ext) + (size_t)n;
  { unsigned char* blk = nullptr; ok = ok && (dalloc(&blk, b->out_bytes) == hipSuccess);
    b->d_obs = (Ext*)blk; b->d_reward = b->d_obs + (size_t)n * NOBS; b->d_done = (unsigned char*)(b->d_reward + n); }
  ok = ok && hipHostMalloc((void**)&b->h_out, b->out_bytes, hipHostMallocDefault) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&b->h_action, (size_t)n * NU * sizeof(Ext), hipHostMallocDefault) == hipSuccess;
  A(b->B.aovf, (size_t)n * AOVF_COLS * 64);
  A(b->B.kin, (size_t)n * KIN_DOUBLES); A(b->B.kin_ok, n);
  A(b->B.redo_list, n); A(b->B.redo_count, 2 * DM_MAX_PIPELINE); A(b->B.redo_why, 8);
  A(b->d_ord_cnt, DM_MAX_PIPELINE * 3 * ORD_BUCKETS);
  A(b->d_ord_list, (size_t)3 * ORD_BUCKETS * n);                      // (768 B per env of address space, 8 B of it touched per step; allocated here so that no step call can run out of memory half-way through its parts)
  A(b->d_qpos_in, (size_t)n * NQ); A(b->d_qvel_in, (size_t)n * NV); A(b->d_fidx_in, n); A(b->d_debug, DM_DEBUG_DOUBLES);
  if (sizeof(Real) != sizeof(Ext)) A(b->d_cvt, (size_t)n * NB * 3);   // largest Real field per env: xipos (42)
#undef A
  if (!ok) { dm_batch_destroy(b); return fail(DM_ENOMEM, "dm_batch_create: hipMalloc failed"); }
  ok = hipMemcpy(b->d_model, &hm, sizeof hm, hipMemcpyHostToDevice) == hipSuccess;
  auto upload = [](Real* dst, const double* src, size_t cnt) {      // host float64 table -> device `Real` array
    std::vector<Real> tmp(src, src + cnt);
    return hipMemcpy(dst, tmp.data(), cnt * sizeof(Real), hipMemcpyHostToDevice) == hipSuccess;
  };
  ok = ok && upload(b->d_cfg, mc->cfg.data(), mc->cfg.size());
  ok = ok && upload(b->d_vel, mc->vel.data(), mc->vel.size());
  if (b->d_imit) {
    ok = ok && upload(b->d_imit, mc->imit_params.data(), 32);
    ok = ok && upload(b->d_imit + 32, mc->imit_table.data(), mc->imit_table.size());
    for (int k = 0; k < 32; k++) b->B.imit_params[k] = mc->imit_params[k];
  }
  b->B.imit_pdev = b->d_imit;
  b->B.imit_table = b->d_imit ? b->d_imit + 32 : nullptr;
  // initial state = MjSim(model): qpos0, zero velocity
  std::vector<Real> q0((size_t)n * NQ);
  for (int e2 = 0; e2 < n; e2++) for (int k = 0; k < NQ; k++) q0[(size_t)e2 * NQ + k] = hm.qpos0[k];
  ok = ok && hipMemcpy(b->B.qpos, q0.data(), q0.size() * sizeof(Real), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { dm_batch_destroy(b); return fail(DM_EHIP, "dm_batch_create: upload failed"); }
  b->B.mocap_cfg = b->d_cfg; b->B.mocap_vel = b->d_vel; b->B.mocap_dt = mc->dt; b->B.n_frames = mc->n_frames; b->B.n_envs = n; b->B.env_offset = 0;
  b->B.reward_mode = 0; b->B.autoreset = 0; b->B.action_mode = 0; b->B.seed = 0; b->B.diag = 1;
  hipEventCreate(&b->ev0); hipEventCreate(&b->ev1);
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) == hipSuccess && cus > 0) b->resident_waves = cus * 4 * DM_STEP_WAVES; }
  *out = b;
  return DM_OK;
}

extern "C" int dm_batch_set_stream(dm_batch* b, void* s) {
  if (!b) return fail(DM_EINVAL, "null batch");
  if (settle(b)) return fail(DM_EHIP, "dm_batch_set_stream: flushing the step queue failed");
  hipStreamSynchronize(b->stream);   /* keep ordering with work already queued on the previous stream */
  if (b->own_stream && b->stream) hipStreamDestroy(b->stream);
  b->stream = (hipStream_t)s; b->own_stream = false;
  return DM_OK;
}

extern "C" int dm_batch_set_option(dm_batch* b, int32_t opt, int64_t v) {
  if (!b) return fail(DM_EINVAL, "null batch");
  if (!b->q.empty()) { HIPCHK(hipSetDevice(b->device)); const int rc = flush_queue(b); if (rc != DM_OK) return rc; }   /* queued steps ran under the old options */
  switch (opt) {
    case DM_OPT_STEP_QUEUE:
      if (v < 0 || v > DM_MAX_STEP_QUEUE) return fail(DM_EINVAL, "step queue depth must be 0..DM_MAX_STEP_QUEUE");
      b->queue_cap = (int)v; break;
    case DM_OPT_REWARD_MODE:
      if (v < 0 || v > 4) return fail(DM_EINVAL, "reward mode must be 0..4");
      if (v >= 3 && !b->d_imit) return fail(DM_EINVAL, "reward modes 3 and 4 need dm_mocap_set_imitation() before dm_batch_create()");
      b->B.reward_mode = (int)v; break;
    case DM_OPT_AUTORESET: if (v < 0 || v > 2) return fail(DM_EINVAL, "autoreset must be 0..2"); b->B.autoreset = (int)v; break;
    case DM_OPT_ACTION_MODE: if (v < 0 || v > 2) return fail(DM_EINVAL, "action mode must be 0..2"); b->B.action_mode = (int)v; break;
    case DM_OPT_SEED: b->B.seed = (unsigned long long)v; break;
    case DM_OPT_DIAGNOSTICS: b->B.diag = v != 0; break;
    case DM_OPT_PIPELINE: {
      if (v < 1 || v > DM_MAX_PIPELINE) return fail(DM_EINVAL, "pipeline depth must be 1..DM_MAX_PIPELINE");
      HIPCHK(hipSetDevice(b->device));
      if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
      for (int h = 0; h < (int)v && v > 1; h++) {
        if (!b->ps[h]) HIPCHK(hipStreamCreateWithFlags(&b->ps[h], hipStreamNonBlocking));
        if (!b->ev_done[h]) HIPCHK(hipEventCreateWithFlags(&b->ev_done[h], hipEventDisableTiming));
      }
      if (v > 1 && !b->ev_in) HIPCHK(hipEventCreateWithFlags(&b->ev_in, hipEventDisableTiming));
      b->pipe = (int)v;
      b->B.order = nullptr;   /* the stored dispatch order belongs to the previous partition: identity for the next launch */
      for (int h = 0; h < DM_MAX_PIPELINE; h++) b->ord_valid[h] = false;   /* ... and so do the tickets of the per-step launches */
      b->redo_mode = -1;      /* sub-batches beyond the new depth keep whatever their redo counters last held: the next packed step re-zeroes all
                                 2 * DM_MAX_PIPELINE counters and restarts the phase (everything in flight was joined above) */
      break;
    }
    case 106: b->horizon_mode = v < 0 ? -1 : (v != 0 ? 1 : 0); break;   /* dm_batch_rollout on the packed path: 1 one launch per horizon, 0 step launches, -1 (default) by batch size */
    case DM_OPT_PACKED: case 105:                                   /* 1: four environments per wavefront (k_step_packed) where it covers the configuration; 2: the same, per-step launches with the three-set code */
      if (v < 0 || v > 2) return fail(DM_EINVAL, "DM_OPT_PACKED must be 0, 1 or 2");
      b->packed = v != 0; b->packed_ext = v == 2; break;
    case 104: b->reorder = v != 0; if (!b->reorder) b->B.order = nullptr; break;   /* 1 (default): longest-first dispatch order (k_order) */
    case 100: b->B.env_offset = (int)v; break;  /* global id of env 0 (multi-GPU sharding) */
    case 102: b->two_tier = v != 0; break;       /* 1 (default): register tier of NARROW_ROWS columns + overflow strip; 0: all 64 columns in registers */
    case 103: {                                 /* test hook: 1 = every PGS sweep takes the guarded-replay path (results must not change) */
      const Real lvl = v ? Real(-1e300) : Real(1e-10);
      HIPCHK(hipMemcpyAsync((char*)b->d_model + offsetof(DevModel<Real>, pgs_detect), &lvl, sizeof lvl, hipMemcpyHostToDevice, b->stream));
      HIPCHK(hipStreamSynchronize(b->stream));
      break;
    }
    case 101:                                   /* per-stage cycle profile on/off (diagnostic) */
      b->prof = v != 0;
      if (b->prof && !b->d_prof) { if (hipMalloc((void**)&b->d_prof, (size_t)b->n * dm::PROF_SLOTS * sizeof(long long)) != hipSuccess) return fail(DM_ENOMEM, "prof alloc"); }
      break;
    default: return fail(DM_EINVAL, "unknown option");
  }
  return DM_OK;
}

static int stage_in(dm_batch* b, void* dst, const void* src, size_t bytes, int kind, const void** use) {
  if (!src) { *use = nullptr; return DM_OK; }
  if (kind == DM_PTR_DEVICE) { *use = src; return DM_OK; }
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, b->stream));
  *use = dst;
  return DM_OK;
}

extern "C" int dm_batch_set_state(dm_batch* b, const double* qpos, const double* qvel, const int32_t* fidx, const uint8_t* mask, int32_t kind) {
  if (!b || !qpos || !qvel) return fail(DM_EINVAL, "dm_batch_set_state: null argument");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  const void *q, *v, *f, *mk; int rc;
  if ((rc = stage_in(b, b->d_qpos_in, qpos, (size_t)b->n * NQ * 8, kind, &q))) return rc;
  if ((rc = stage_in(b, b->d_qvel_in, qvel, (size_t)b->n * NV * 8, kind, &v))) return rc;
  if ((rc = stage_in(b, b->d_fidx_in, fidx, (size_t)b->n * 4, kind, &f))) return rc;
  if ((rc = stage_in(b, b->d_mask, mask, (size_t)b->n, kind, &mk))) return rc;
  HIPCHK(hipMemsetAsync(b->B.kin_ok, 0, (size_t)b->n, b->stream));          // parked kinematics belong to the old states
  hipLaunchKernelGGL(k_set_state, dim3(b->n), dim3(64), 0, b->stream, b->d_model, b->B, (const Ext*)q, (const Ext*)v, (const int*)f, (const unsigned char*)mk);
  HIPCHK(hipGetLastError());
  if (kind == DM_PTR_HOST) HIPCHK(hipStreamSynchronize(b->stream));
  return DM_OK;
}

extern "C" int dm_batch_reset(dm_batch* b, int32_t mode, int32_t hard, const uint8_t* mask, int32_t kind) {
  if (!b || mode < 0 || mode > 2) return fail(DM_EINVAL, "dm_batch_reset: bad argument");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  const void* mk; int rc;
  if ((rc = stage_in(b, b->d_mask, mask, (size_t)b->n, kind, &mk))) return rc;
  HIPCHK(hipMemsetAsync(b->B.kin_ok, 0, (size_t)b->n, b->stream));
  hipLaunchKernelGGL(k_reset, dim3(b->n), dim3(64), 0, b->stream, b->d_model, b->B, (int)mode, (int)hard, (const unsigned char*)mk);
  HIPCHK(hipGetLastError());
  if (kind == DM_PTR_HOST) HIPCHK(hipStreamSynchronize(b->stream));
  return DM_OK;
}

// the packed path covers this batch's configuration (reward modes alive / v3-config / v2-pose / imitation, the two-tier kernel family)
static bool packed_covers(const dm_batch* b) { return b->packed && b->B.reward_mode <= 4 && b->two_tier; }
// a dm_batch_step call may be queued: device pointers, no fused policy step, no per-stage profiling, and a configuration for which ONE launch
// per horizon is the faster form (rollout_as_one_launch)
static bool rollout_as_one_launch(const dm_batch* b) {
  // ONE launch for the horizon where that wins (measured, profiles/r03_bench_*): batches with constraint rows of up to two packed waves per
  // SIMD (8 192 envs on an MI355X; at 4 096 envs 17.3 M env-steps/s against 12.2 M for the one-env steps and 11.4 M for the packed ones).
  // Larger batches run several rounds of waves per step, which balances the slow waves by itself, while a horizon launch has its own
  // end-of-horizon tail (16 384 envs: 19.5 M per step, ~17 M per horizon); without rows there is no slow wave to wait for.
  const int simds = b->resident_waves / DM_STEP_WAVES;
  return packed_covers(b) && (b->horizon_mode == 1 || (b->horizon_mode < 0 && b->has_rows && b->n <= 2 * SLOTS * simds));
}
static bool can_queue(const dm_batch* b, int kind, const dmp::PolicyArgs* pol) {
  return kind == DM_PTR_DEVICE && !pol && !b->prof && rollout_as_one_launch(b);      // (with timing on, the events bracket the horizon launch)
}
static int ensure_rows(dm_batch* b, int T) {
  if (T <= b->rows_cap) return DM_OK;
  HIPCHK(hipStreamSynchronize(b->stream));            // (a launch still reading the old table)
  if (b->d_rows) HIPCHK(hipFree(b->d_rows));
  b->d_rows = nullptr; b->rows_cap = 0;
  const int cap = (T + 255) / 256 * 256;
  HIPCHK(hipMalloc((void**)&b->d_rows, (size_t)cap * sizeof(StepRow)));
  b->rows_cap = cap;
  return DM_OK;
}
// T steps of every environment in ONE launch (k_rollout_packed) on the batch's stream; b->d_rows[0 .. T) has been written on that stream
static int launch_horizon(dm_batch* b, int T, int nsub, const dmp::PolicyArgs& pa) {
  hipLaunchKernelGGL(k_put_batch, dim3(1), dim3(64), 0, b->stream, b->B, b->d_B);
  if (b->timing) HIPCHK(hipEventRecord(b->ev0, b->stream));
  hipLaunchKernelGGL(k_rollout_packed, dim3((b->n + SLOTS - 1) / SLOTS), dim3(64), 0, b->stream, b->d_model, (const Batch<Real>*)b->d_B, (const StepRow*)b->d_rows, (int)nsub, 0, b->n, (int)T, pa, b->prof ? b->d_prof : (long long*)nullptr);
  if (b->timing) { HIPCHK(hipEventRecord(b->ev1, b->stream)); b->ev_pending = true; }
  // dispatch order for the next launch: environments with similar row counts share a wave (and, for per-step launches, longest first)
  if (b->reorder && b->has_rows) {
    const int parts = b->pipe > 1 ? b->pipe : 1;
    for (int h = 0; h < parts; h++) {
      const int lo = (int)((long long)b->n * h / parts), hi = (int)((long long)b->n * (h + 1) / parts);
      if (hi > lo) hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, b->stream, b->B, b->d_order, lo, hi - lo);
    }
    b->B.order = b->d_order;
  }
  HIPCHK(hipGetLastError());
  return DM_OK;
}
static int flush_queue(dm_batch* b) {
  const int T = (int)b->q.size();
  if (T == 0) return DM_OK;
  HIPCHK(hipSetDevice(b->device));
  if (pipe_join(b)) return fail(DM_EHIP, "pipeline join failed");
  int rc = ensure_rows(b, T);
  if (rc != DM_OK) return rc;
  for (int t0 = 0; t0 < T; t0 += ROW_CHUNK) {
    StepRowChunk c;
    const int cnt = T - t0 < ROW_CHUNK ? T - t0 : ROW_CHUNK;
    for (int k = 0; k < cnt; k++) c.r[k] = b->q[t0 + k];
    for (int k = cnt; k < ROW_CHUNK; k++) c.r[k] = StepRow{nullptr, nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_put_rows, dim3(1), dim3(ROW_CHUNK), 0, b->stream, c, b->d_rows + t0, cnt);
  }
  const int nsub = b->q_nsub;
  b->q.clear();                                        // (before the launch: an error below must not leave the steps queued for a second run)
  b->queue_flushes += 1; b->queue_steps += T;
  const dmp::PolicyArgs nopol{nullptr, nullptr, nullptr, 0, 0ull, 0ull};
  return launch_horizon(b, T, nsub, nopol);
}

// Self-ordering per-step launches (env_step.h dispatch_env / order_ticket): the descriptor a step launch of part h = [lo, ...) gets — it reads the
// tickets the part's previous launch left (phase p), its envs take theirs from phase p + 1, its first workgroup clears phase p + 2 for the launch
// after it.  `st` = the stream the launch goes to.  A part's tickets stay a permutation of the part whatever runs in between (horizon launches,
// resets: only their keys grow stale); a new partition (DM_OPT_PIPELINE) starts afresh.
// The part's phase and `valid` flag move on only once its launch is known to have gone out (ord_commit after hipGetLastError): a failed launch leaves the
// descriptor where the last successful one put it, so the next launch reads tickets somebody really counted.
static int ord_bind(dm_batch* b, Batch<Real>& Bh, int h, int lo, hipStream_t st) {
  int* cnt = b->d_ord_cnt + (size_t)h * 3 * ORD_BUCKETS;
  if (!b->ord_valid[h]) { HIPCHK(hipMemsetAsync(cnt, 0, 3 * ORD_BUCKETS * sizeof(int), st)); b->ord_phase[h] = 0; }
  const int p = b->ord_phase[h], pn = (p + 1) % 3, pz = (p + 2) % 3;
  const size_t n = (size_t)b->n;
  Bh.ord_in = b->ord_valid[h] ? cnt + p * ORD_BUCKETS : nullptr;
  Bh.ordl_in = b->d_ord_list + (size_t)p * ORD_BUCKETS * n + lo;
  Bh.ord_out = cnt + pn * ORD_BUCKETS;
  Bh.ordl_out = b->d_ord_list + (size_t)pn * ORD_BUCKETS * n + lo;
  Bh.ord_zero = cnt + pz * ORD_BUCKETS;
  Bh.ord_stride = b->n;
  return DM_OK;
}
static void ord_commit(dm_batch* b, int h) { b->ord_phase[h] = (b->ord_phase[h] + 1) % 3; b->ord_valid[h] = true; }

static int step_impl(dm_batch* b, const double* action, double* obs, double* reward, uint8_t* done, int32_t nsub, int32_t kind, const dmp::PolicyArgs* pol) {
  if (!b || !action || !obs || !reward || !done || nsub < 1) return fail(DM_EINVAL, "dm_batch_step: bad argument");
  if (pol && (kind != DM_PTR_DEVICE || b->prof || !b->two_tier)) return fail(DM_EINVAL, "dm_batch_step_act: device pointers, the two-tier kernel and no profiling");
  HIPCHK(hipSetDevice(b->device));
  if (b->queue_cap > 0 && can_queue(b, kind, pol)) {
    // DM_OPT_STEP_QUEUE: remember the call; the queued steps run as ONE horizon launch (flush_queue).  A call that names a buffer an
    // earlier queued call names (the same action / output tensors step after step) runs that earlier call first: a caller that reuses
    // buffers has, by the pipelined contract, joined in between and sees plain step-by-step behaviour.
    // "names a buffer" = its bytes overlap the bytes of ANY buffer of a queued call, in whatever role: the same tensors step after step, a view that
    // starts elsewhere in one of them, a queued call's observations handed in as this call's action.
    bool reuse = nsub != b->q_nsub && !b->q.empty();
    {
      const size_t n = (size_t)b->n;
      const char* p[4] = {(const char*)action, (const char*)obs, (const char*)reward, (const char*)done};
      const size_t len[4] = {n * NU * sizeof(Ext), n * NOBS * sizeof(Ext), n * sizeof(Ext), n};
      for (const StepRow& r : b->q) {
        const char* qp[4] = {(const char*)r.action, (const char*)r.obs, (const char*)r.reward, (const char*)r.done};
        for (int i = 0; i < 4 && !reuse; i++) for (int j = 0; j < 4; j++) {
          if (p[i] < qp[j] + len[j] && qp[j] < p[i] + len[i]) { reuse = true; break; }
        }
        if (reuse) break;
      }
    }
    if (reuse || (int)b->q.size() >= b->queue_cap) { const int rc = flush_queue(b); if (rc != DM_OK) return rc; }
    b->q.push_back(StepRow{action, obs, reward, done}); b->q_nsub = nsub;
    return DM_OK;
  }
  if (!b->q.empty()) { const int rc = flush_queue(b); if (rc != DM_OK) return rc; }
  const void* a = action;
  if (kind == DM_PTR_HOST) {
    memcpy(b->h_action, action, (size_t)b->n * NU * sizeof(Ext));
    HIPCHK(hipMemcpyAsync(b->d_action, b->h_action, (size_t)b->n * NU * sizeof(Ext), hipMemcpyHostToDevice, b->stream));
    a = b->d_action;
  }
  Ext* o = kind == DM_PTR_DEVICE ? obs : b->d_obs;
  Ext* r = kind == DM_PTR_DEVICE ? reward : b->d_reward;
  unsigned char* dn = kind == DM_PTR_DEVICE ? done : b->d_done;
  const bool piped = b->pipe > 1 && kind == DM_PTR_DEVICE && !b->prof && b->two_tier;
  if (!piped && pipe_join(b)) return fail(DM_EHIP, "pipeline join failed");
  if (b->timing) { if (b->ev_pending) { hipEventSynchronize(b->ev1); hipEventElapsedTime(&b->last_ms, b->ev0, b->ev1); } if (!piped) HIPCHK(hipEventRecord(b->ev0, b->stream)); }
  const bool reorder = b->reorder && b->has_rows && b->n > b->resident_waves;   // more envs than resident waves: later rounds exist, their tail matters
  // the packed kernel covers: models without constraint rows, reward modes alive / v3-config / v2-pose, no fused policy step
  const bool packed_step = b->packed && b->B.reward_mode <= 4 && b->two_tier;      // this call runs on the packed kernels (profiled or not)
  const bool use_packed = packed_step && !b->prof;
  const dmp::PolicyArgs nopol{nullptr, nullptr, nullptr, 0, 0ull, 0ull};
  if (packed_step) {
    const int mode = piped ? 1 : 0;
    if (mode != b->redo_mode) {      // (rare: the first packed step, or a host-pointer step between pipelined ones; every earlier launch is ordered before this stream here)
      if (piped && pipe_join(b)) return fail(DM_EHIP, "pipeline join failed");
      HIPCHK(hipMemsetAsync(b->B.redo_count, 0, 2 * DM_MAX_PIPELINE * sizeof(int), b->stream));
      b->redo_mode = mode; b->redo_phase = 0;
    }
  }
  unsigned ord_bound = 0;               // parts whose launch of this call took a ticket descriptor (committed below, after the launches went out)
  constexpr int REDO_BLOCKS = 1024;     // (round 5: 64 persistent one-wave workgroups took 5 ms to walk the 1 100 overflows a standing population of 8 192 envs produces per step on the lean kernel; an empty launch of 1 024 costs the same few microseconds)
  if (b->prof && packed_step) {
    HIPCHK(hipMemsetAsync(b->d_prof, 0, (size_t)b->n * dm::PROF_SLOTS * sizeof(long long), b->stream));
    int* rc = b->B.redo_count + b->redo_phase; int* rn = b->B.redo_count + (1 - b->redo_phase);
    hipLaunchKernelGGL(k_step_packed_prof, dim3((b->n + SLOTS - 1) / SLOTS), dim3(64), 0, b->stream, b->d_model, b->B, (const Ext*)a, o, r, dn, (int)nsub, 0, b->n, rc, b->d_prof);
    if (b->has_rows) hipLaunchKernelGGL(k_step_redo, dim3(REDO_BLOCKS), dim3(64), 0, b->stream, b->d_model, b->B, (const Ext*)a, o, r, dn, (int)nsub, 0, (const int*)rc, rn, nopol);
  } else if (b->prof) hipLaunchKernelGGL(k_step_prof, dim3(b->n), dim3(64), 0, b->stream, b->d_model, b->B, (const Ext*)a, o, r, dn, (int)nsub, b->d_prof);
  else if (piped) {
    // Sub-batch h's launch of THIS call depends on its own launch of the previous call (stream order on ps[h]) and on the
    // caller's inputs (ev_in), not on the other sub-batches: while the last, cheap workgroups of one sub-batch drain, the
    // next sub-batch's fill the freed wave slots, across calls.  Nothing is inserted into the caller's stream here — it joins
    // (dm_batch_join, or any other entry point) when it needs the outputs.
    HIPCHK(hipEventRecord(b->ev_in, b->stream));
    for (int h = 0; h < b->pipe; h++) {
      const int lo = (int)((long long)b->n * h / b->pipe), hi = (int)((long long)b->n * (h + 1) / b->pipe);
      if (hi <= lo) continue;
      HIPCHK(hipStreamWaitEvent(b->ps[h], b->ev_in, 0));
      Batch<Real> Bh = b->B;                                                   // this part's launch orders itself from the tickets its previous launch left
      if (reorder) { const int rc2 = ord_bind(b, Bh, h, lo, b->ps[h]); if (rc2 != DM_OK) return rc2; ord_bound |= 1u << h; }
      if (b->timing && h == 0) HIPCHK(hipEventRecord(b->ev0, b->ps[0]));     // timing: sub-batch 0's kernel on ITS stream
      if (use_packed) {
        int* rc = b->B.redo_count + 2 * h + b->redo_phase; int* rn = b->B.redo_count + 2 * h + (1 - b->redo_phase);
        if (pol) hipLaunchKernelGGL(b->packed_ext ? k_step_packed_act_ext : k_step_packed_act, dim3((hi - lo + SLOTS - 1) / SLOTS), dim3(64), 0, b->ps[h], b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, lo, hi - lo, rc, *pol);
        else hipLaunchKernelGGL(b->packed_ext ? k_step_packed_ext : k_step_packed, dim3((hi - lo + SLOTS - 1) / SLOTS), dim3(64), 0, b->ps[h], b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, lo, hi - lo, rc);
        if (b->has_rows) hipLaunchKernelGGL(k_step_redo, dim3(REDO_BLOCKS), dim3(64), 0, b->ps[h], b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, lo, (const int*)rc, rn, pol ? *pol : nopol);
      }
      else if (pol) hipLaunchKernelGGL(k_step_act, dim3(hi - lo), dim3(64), 0, b->ps[h], b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, lo, hi - lo, *pol);
      else hipLaunchKernelGGL(k_step_narrow, dim3(hi - lo), dim3(64), 0, b->ps[h], b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, lo, hi - lo);
      if (b->timing && h == 0) { HIPCHK(hipEventRecord(b->ev1, b->ps[0])); b->ev_pending = true; }
      HIPCHK(hipEventRecord(b->ev_done[h], b->ps[h]));
    }
    b->pipe_pending = true;
  } else if (b->two_tier) {
    // (one launch over the whole batch.  With a pipeline depth configured the tickets are kept per part — a pipelined launch must find exactly its
    //  part's envs in them, otherwise two streams could step one env at once — so this launch takes none and falls back to the stored order: DM_OPT 104
    //  has no effect on host-pointer steps and other unpipelined launches of a batch whose DM_OPT_PIPELINE is above 1 — a performance matter only,
    //  results do not depend on the dispatch order; include/dmenv.h says so.)
    Batch<Real> Bh = b->B;
    if (reorder && b->pipe <= 1) { const int rc2 = ord_bind(b, Bh, 0, 0, b->stream); if (rc2 != DM_OK) return rc2; ord_bound |= 1u; }
    if (use_packed) {
      // (a step that is not pipelined has joined every sub-batch stream: all of them are idle, so ONE pair of counters is clean — pair 0's
      //  two are cleared here once if a pipelined step used them before)
      int* rc = b->B.redo_count + b->redo_phase; int* rn = b->B.redo_count + (1 - b->redo_phase);
      if (pol) hipLaunchKernelGGL(b->packed_ext ? k_step_packed_act_ext : k_step_packed_act, dim3((b->n + SLOTS - 1) / SLOTS), dim3(64), 0, b->stream, b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, 0, b->n, rc, *pol);
      else hipLaunchKernelGGL(b->packed_ext ? k_step_packed_ext : k_step_packed, dim3((b->n + SLOTS - 1) / SLOTS), dim3(64), 0, b->stream, b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, 0, b->n, rc);
      if (b->has_rows) hipLaunchKernelGGL(k_step_redo, dim3(REDO_BLOCKS), dim3(64), 0, b->stream, b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, 0, (const int*)rc, rn, pol ? *pol : nopol);
    }
    else if (pol) hipLaunchKernelGGL(k_step_act, dim3(b->n), dim3(64), 0, b->stream, b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, 0, b->n, *pol);
    else hipLaunchKernelGGL(k_step_narrow, dim3(b->n), dim3(64), 0, b->stream, b->d_model, Bh, (const Ext*)a, o, r, dn, (int)nsub, 0, b->n);
  } else hipLaunchKernelGGL(k_step, dim3(b->n), dim3(64), 0, b->stream, b->d_model, b->B, (const Ext*)a, o, r, dn, (int)nsub);
  HIPCHK(hipGetLastError());
  for (int h = 0; h < DM_MAX_PIPELINE; h++) if ((ord_bound >> h) & 1u) ord_commit(b, h);
  if (packed_step) b->redo_phase ^= 1;
  if (b->timing && !piped) { HIPCHK(hipEventRecord(b->ev1, b->stream)); b->ev_pending = true; }
  if (kind == DM_PTR_HOST) {
    HIPCHK(hipMemcpyAsync(b->h_out, b->d_obs, b->out_bytes, hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    const size_t ob = (size_t)b->n * NOBS * sizeof(Ext), rb = (size_t)b->n * sizeof(Ext);
    memcpy(obs, b->h_out, ob); memcpy(reward, b->h_out + ob, rb); memcpy(done, b->h_out + ob + rb, (size_t)b->n);
  }
  return DM_OK;
}

extern "C" int dm_batch_step(dm_batch* b, const double* action, double* obs, double* reward, uint8_t* done, int32_t nsub, int32_t kind) {
  return step_impl(b, action, obs, reward, done, nsub, kind, nullptr);
}
extern "C" int dm_batch_step_act(dm_batch* b, const double* action, double* obs, double* reward, uint8_t* done, int32_t nsub,
                                 const float* weights, double* next_action, float* next_vpred, int32_t stochastic, uint64_t seed, uint64_t counter) {
  if (!weights || !next_action || !next_vpred) return fail(DM_EINVAL, "dm_batch_step_act: null policy argument");
  dmp::PolicyArgs pa{weights, next_action, next_vpred, (int)stochastic, (unsigned long long)seed, (unsigned long long)counter};
  return step_impl(b, action, obs, reward, done, nsub, DM_PTR_DEVICE, &pa);
}

/* T steps per call (device pointers).  On the packed path ONE launch runs the whole horizon (k_rollout_packed); elsewhere T step launches. */
extern "C" int dm_batch_rollout(dm_batch* b, double* action, double* obs, double* reward, uint8_t* done, int32_t T, int32_t nsub,
                                const float* weights, float* vpred, int32_t stochastic, uint64_t seed, uint64_t counter) {
  if (!b || !action || !obs || !reward || !done || T < 1 || nsub < 1) return fail(DM_EINVAL, "dm_batch_rollout: bad argument");
  if (weights && !vpred) return fail(DM_EINVAL, "dm_batch_rollout: a policy needs the value rows");
  const size_t n = (size_t)b->n;
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  if (!rollout_as_one_launch(b)) {        // (with option 101 the horizon launch records every wave's cycles)
    for (int t = 0; t < T; t++) {
      dmp::PolicyArgs pa{weights, action + (size_t)(t + 1) * n * NU, weights ? vpred + (size_t)t * n : nullptr, (int)stochastic, (unsigned long long)seed, (unsigned long long)counter + t};
      const int rc = step_impl(b, action + (size_t)t * n * NU, obs + (size_t)t * n * NOBS, reward + (size_t)t * n, done + (size_t)t * n, nsub, DM_PTR_DEVICE, weights ? &pa : nullptr);
      if (rc != DM_OK) return rc;
    }
    return DM_OK;
  }
  const int rc = ensure_rows(b, (int)T);
  if (rc != DM_OK) return rc;
  hipLaunchKernelGGL(k_fill_rows, dim3(((int)T + 63) / 64), dim3(64), 0, b->stream, b->d_rows, (const Ext*)action, obs, reward, done, (int)T, n);
  dmp::PolicyArgs pa{weights, action, vpred, (int)stochastic, (unsigned long long)seed, (unsigned long long)counter};
  return launch_horizon(b, (int)T, (int)nsub, pa);
}

extern "C" int dm_batch_get_obs(dm_batch* b, double* obs, int32_t kind) {
  if (!b || !obs) return fail(DM_EINVAL, "dm_batch_get_obs: null argument");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  Ext* o = kind == DM_PTR_DEVICE ? obs : b->d_obs;
  const int tot = b->n * NOBS;
  hipLaunchKernelGGL(k_get_obs, dim3((tot + 255) / 256), dim3(256), 0, b->stream, b->B, o);
  HIPCHK(hipGetLastError());
  if (kind == DM_PTR_HOST) { HIPCHK(hipMemcpyAsync(obs, b->d_obs, (size_t)tot * 8, hipMemcpyDeviceToHost, b->stream)); HIPCHK(hipStreamSynchronize(b->stream)); }
  return DM_OK;
}

// field -> device array, element count, and whether its elements are `Real` (float64 at the ABI) or int32
static int field_ptr(dm_batch* b, int field, void** p, size_t* count, bool* is_real) {
  const size_t n = b->n;
  *is_real = false;
  switch (field) {
    case DM_F_QPOS: *p = b->B.qpos; *count = n * NQ; *is_real = true; break;
    case DM_F_QVEL: *p = b->B.qvel; *count = n * NV; *is_real = true; break;
    case DM_F_QACC_WARMSTART: *p = b->B.qws; *count = n * NV; *is_real = true; break;
    case DM_F_TIME: *p = b->B.time; *count = n; *is_real = true; break;
    case DM_F_FRAME_IDX: *p = b->B.frame_idx; *count = n; break;
    case DM_F_FRAME_INIT: *p = b->B.frame_init; *count = n; break;
    case DM_F_XIPOS: *p = b->B.xipos; *count = n * NB * 3; *is_real = true; break;
    case DM_F_COM_Z: *p = b->B.comz; *count = n; *is_real = true; break;
    case DM_F_NCON: *p = b->B.ncon; *count = n; break;
    case DM_F_NEFC: *p = b->B.nefc; *count = n; break;
    case DM_F_CONTACT_GEOMS: *p = b->B.cong; *count = n * MAXEFC * 2; break;
    case DM_F_STATUS: *p = b->B.status; *count = n; break;
    case DM_F_SOLVER_ITER: *p = b->B.solver_iter; *count = n; break;
    case DM_F_CTRL: *p = b->B.ctrl; *count = n * NU; *is_real = true; break;
    case DM_F_EPISODE: *p = b->B.episode; *count = n; break;
    case DM_F_CYCLE: *p = b->B.cycle; *count = n; break;
    default: return fail(DM_EINVAL, "unknown field");
  }
  return DM_OK;
}
extern "C" int dm_batch_get(dm_batch* b, int32_t field, void* out, size_t bytes, int32_t kind) {
  if (!b || !out) return fail(DM_EINVAL, "dm_batch_get: null argument");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  void* p; size_t cnt; bool real; int rc;
  if ((rc = field_ptr(b, field, &p, &cnt, &real))) return rc;
  const size_t need = cnt * (real ? sizeof(Ext) : 4);
  if (bytes != need) return fail(DM_EINVAL, "dm_batch_get: buffer size does not match the field");
  if (real && sizeof(Real) != sizeof(Ext)) {       // float32 build: widen on the device, then copy
    Ext* dst = kind == DM_PTR_DEVICE ? (Ext*)out : b->d_cvt;
    hipLaunchKernelGGL(k_to_ext, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, b->stream, (const Real*)p, dst, cnt);
    HIPCHK(hipGetLastError());
    if (kind == DM_PTR_HOST) HIPCHK(hipMemcpyAsync(out, dst, need, hipMemcpyDeviceToHost, b->stream));
  } else HIPCHK(hipMemcpyAsync(out, p, need, kind == DM_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, b->stream));
  if (kind == DM_PTR_HOST) HIPCHK(hipStreamSynchronize(b->stream));
  return DM_OK;
}
extern "C" int dm_batch_set(dm_batch* b, int32_t field, const void* in, size_t bytes, int32_t kind) {
  if (!b || !in) return fail(DM_EINVAL, "dm_batch_set: null argument");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  void* p; size_t cnt; bool real; int rc;
  if ((rc = field_ptr(b, field, &p, &cnt, &real))) return rc;
  const size_t need = cnt * (real ? sizeof(Ext) : 4);
  if (bytes != need) return fail(DM_EINVAL, "dm_batch_set: buffer size does not match the field");
  if (field == DM_F_XIPOS || field == DM_F_COM_Z || field == DM_F_NCON || field == DM_F_NEFC || field == DM_F_CONTACT_GEOMS || field == DM_F_SOLVER_ITER)
    return fail(DM_EINVAL, "dm_batch_set: derived field is read-only");
  if (field == DM_F_QPOS) HIPCHK(hipMemsetAsync(b->B.kin_ok, 0, (size_t)b->n, b->stream));   // parked kinematics belong to the old positions
  if (real && sizeof(Real) != sizeof(Ext)) {       // float32 build: copy, then narrow on the device
    const Ext* src = (const Ext*)in;
    if (kind == DM_PTR_HOST) { HIPCHK(hipMemcpyAsync(b->d_cvt, in, need, hipMemcpyHostToDevice, b->stream)); src = b->d_cvt; }
    hipLaunchKernelGGL(k_from_ext, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, b->stream, src, (Real*)p, cnt);
    HIPCHK(hipGetLastError());
  } else HIPCHK(hipMemcpyAsync(p, in, need, kind == DM_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, b->stream));
  if (kind == DM_PTR_HOST) HIPCHK(hipStreamSynchronize(b->stream));
  return DM_OK;
}

extern "C" int dm_batch_debug_forward(dm_batch* b, int32_t env, double* out_host) {
  if (!b || !out_host || env < 0 || env >= b->n) return fail(DM_EINVAL, "dm_batch_debug_forward: bad argument");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  HIPCHK(hipMemsetAsync(b->d_debug, 0, DM_DEBUG_DOUBLES * 8, b->stream));
  hipLaunchKernelGGL(k_debug_forward, dim3(1), dim3(64), 0, b->stream, b->d_model, b->B, (int)env, b->d_debug);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_host, b->d_debug, DM_DEBUG_DOUBLES * 8, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return DM_OK;
}

extern "C" int dm_batch_read_profile(dm_batch* b, long long* out_host) {   /* [N,8]: kin, mass, bias, rows, constraint, total, nefc, iters */
  if (!b || !out_host || !b->d_prof) return fail(DM_EINVAL, "profile not enabled");
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out_host, b->d_prof, (size_t)b->n * dm::PROF_SLOTS * sizeof(long long), hipMemcpyDeviceToHost));
  return DM_OK;
}
extern "C" int dm_batch_enable_timing(dm_batch* b, int32_t on) { if (!b) return fail(DM_EINVAL, "null batch"); b->timing = on != 0; b->ev_pending = false; return DM_OK; }
extern "C" int dm_batch_last_step_ms(dm_batch* b, float* ms) {
  if (!b || !ms) return fail(DM_EINVAL, "null argument");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "dm_batch_last_step_ms: running the queued steps failed");      // (queued steps are timed by the launch that runs them)
  if (!b->ev_pending) return fail(DM_EINVAL, "no timed step recorded");
  HIPCHK(hipEventSynchronize(b->ev1));
  HIPCHK(hipEventElapsedTime(&b->last_ms, b->ev0, b->ev1));
  *ms = b->last_ms;
  return DM_OK;
}
extern "C" int dm_policy_weight_count(void) { return dmp::N_WEIGHTS; }
extern "C" int dm_policy_act(const float* weights, const double* obs, double* action, float* vpred, int32_t n, int32_t stochastic,
                             uint64_t seed, uint64_t counter, void* hip_stream) {
  if (!weights || !obs || !action || !vpred || n <= 0) return fail(DM_EINVAL, "dm_policy_act: bad argument");
  hipLaunchKernelGGL(dmp::k_policy_act, dim3((n + dmp::EB - 1) / dmp::EB), dim3(256), 0, (hipStream_t)hip_stream, weights, obs, action, vpred,
                     (int)n, (int)stochastic, (unsigned long long)seed, (unsigned long long)counter);
  HIPCHK(hipGetLastError());
  return DM_OK;
}
extern "C" int dm_gae(const float* rew, const float* vpred, const int32_t* isnew, const float* nextvpred, float* adv, float* tdlamret,
                      int32_t T, int32_t n, double gamma, double lam, void* hip_stream) {
  if (!rew || !vpred || !isnew || !nextvpred || !adv || !tdlamret || T <= 0 || n <= 0) return fail(DM_EINVAL, "dm_gae: bad argument");
  hipLaunchKernelGGL(dmp::k_gae, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, rew, vpred, (const int*)isnew, nextvpred, adv,
                     tdlamret, (int)T, (int)n, (float)gamma, (float)lam);
  HIPCHK(hipGetLastError());
  return DM_OK;
}
static int pg_set_device(const void* p);
extern "C" int dm_episode_scan(const double* reward, const uint8_t* done, int32_t T, int32_t n, double* cur_ret, int64_t* cur_len, int32_t* count,
                               int32_t cap, int64_t* records, void* hip_stream) {
  if (!reward || !done || !cur_ret || !cur_len || !count || !records || T <= 0 || n <= 0 || cap < 0) return fail(DM_EINVAL, "dm_episode_scan: bad argument");
  if (pg_set_device(reward)) return fail(DM_EHIP, "dm_episode_scan: hipSetDevice failed");      // (the launches go to the device that owns the arrays, whatever the thread's current one)
  hipStream_t st = (hipStream_t)hip_stream;
  HIPCHK(hipMemsetAsync(count, 0, sizeof(int32_t), st));
  hipLaunchKernelGGL(dmp::k_episodes, dim3((n + 255) / 256), dim3(256), 0, st, reward, done, (int)T, (int)n, cur_ret, (long long*)cur_len, (int*)count, (int)cap,
                     (long long*)records);
  HIPCHK(hipGetLastError());
  return DM_OK;
}
extern "C" int dm_vf_param_count(void) { return dmv::NP; }
static size_t up256(size_t x) { return (x + 255) / 256 * 256; }
struct VfScratch { size_t partial, rpart, part_all, means, stds, total; };
static VfScratch vf_scratch_layout(int nb, int bs) {
  const size_t ntile = (size_t)((bs + dmv::SB - 1) / dmv::SB);
  VfScratch L;
  size_t o = 0;
  L.partial = o; o += up256(ntile * dmv::NPAD * sizeof(float));
  L.rpart = o; o += up256((size_t)dmv::RMS_BLOCKS * 2 * dmv::OB * sizeof(double) + 64);          // + the ticket of the three-launch form
  L.part_all = o; o += up256((size_t)nb * dmv::RMS_BLOCKS * 2 * dmv::OB * sizeof(double));
  L.means = o; o += up256((size_t)nb * dmv::OB * sizeof(float));
  L.stds = o; o += up256((size_t)nb * dmv::OB * sizeof(float));
  L.total = o;
  return L;
}
extern "C" size_t dm_vf_scratch_bytes(int32_t nb, int32_t bs) { return vf_scratch_layout(nb < 1 ? 1 : nb, bs < 1 ? 1 : bs).total; }
extern "C" int dm_vf_fit_epoch(const float* ob, const float* ret, int32_t nb, int32_t bs, float* theta, float* adam_m, float* adam_v,
                               const float* step_scale_host, double beta1, double beta2, double eps, double* rms_sum, double* rms_sumsq,
                               double* rms_count, float* rms_mean, float* rms_std, void* scratch, void* hip_stream, int32_t epoch_filter) {
  if (!ob || !ret || !theta || !adam_m || !adam_v || !step_scale_host || !rms_sum || !rms_sumsq || !rms_count || !rms_mean || !rms_std || !scratch ||
      nb < 1 || bs < 1)
    return fail(DM_EINVAL, "dm_vf_fit_epoch: bad argument");
  hipStream_t st = (hipStream_t)hip_stream;
  { // launch on the device that owns the parameters (the caller's stream belongs to it), whatever the thread's current device is
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, theta) == hipSuccess) HIPCHK(hipSetDevice(at.device));
    else (void)hipGetLastError();
  }
  const VfScratch L = vf_scratch_layout(nb, bs);
  const int nblk = (bs + dmv::SB - 1) / dmv::SB;
  char* base = (char*)scratch;
  float* partial = (float*)(base + L.partial);
  if (epoch_filter) {
    // the obs filter's sums of every minibatch up front and their scan (csrc/vf_kernel.h); per minibatch: gradient partials, reduction + Adam
    double* part_all = (double*)(base + L.part_all);
    float* means = (float*)(base + L.means); float* stds = (float*)(base + L.stds);
    hipLaunchKernelGGL(dmv::k_vf_rms_part, dim3(dmv::RMS_BLOCKS, nb), dim3(256), 0, st, ob, (int)bs, part_all);
    hipLaunchKernelGGL(dmv::k_vf_rms_fold, dim3(nb), dim3(128), 0, st, part_all);
    hipLaunchKernelGGL(dmv::k_vf_rms_scan, dim3(1), dim3(64), 0, st, (const double*)part_all, (int)nb, (int)bs, rms_sum, rms_sumsq, rms_count, rms_mean, rms_std, means, stds);
    for (int i = 0; i < nb; i++) {
      hipLaunchKernelGGL(dmv::k_vf_grad, dim3(nblk), dim3(256), 0, st, ob + (size_t)i * bs * dmv::OB, ret + (size_t)i * bs, (int)bs, (const float*)theta,
                         (const float*)(means + (size_t)i * dmv::OB), (const float*)(stds + (size_t)i * dmv::OB), partial);
      hipLaunchKernelGGL(dmv::k_vf_adam, dim3((dmv::NP + dmv::ADAM_PARAMS - 1) / dmv::ADAM_PARAMS), dim3(256), 0, st, (const float*)partial, nblk, theta, adam_m, adam_v,
                         step_scale_host[i], (float)beta1, (float)beta2, (float)eps);
    }
    HIPCHK(hipGetLastError());
    return DM_OK;
  }
  double* rpart = (double*)(base + L.rpart);
  unsigned* ticket = (unsigned*)(rpart + dmv::RMS_BLOCKS * 2 * dmv::OB);
  HIPCHK(hipMemsetAsync(ticket, 0, sizeof(unsigned), st));
  for (int i = 0; i < nb; i++) {
    const float* mbob = ob + (size_t)i * bs * dmv::OB;
    const float* mbret = ret + (size_t)i * bs;
    hipLaunchKernelGGL(dmv::k_vf_rms, dim3(dmv::RMS_BLOCKS), dim3(256), 0, st, mbob, (int)bs, rpart, ticket, rms_sum, rms_sumsq, rms_count, rms_mean, rms_std);
    hipLaunchKernelGGL(dmv::k_vf_grad, dim3(nblk), dim3(256), 0, st, mbob, mbret, (int)bs, (const float*)theta, (const float*)rms_mean,
                       (const float*)rms_std, partial);
    hipLaunchKernelGGL(dmv::k_vf_adam, dim3((dmv::NP + dmv::ADAM_PARAMS - 1) / dmv::ADAM_PARAMS), dim3(256), 0, st, (const float*)partial, nblk, theta, adam_m, adam_v,
                       step_scale_host[i], (float)beta1, (float)beta2, (float)eps);
    HIPCHK(hipGetLastError());
  }
  return DM_OK;
}
// the obs filter's update with a whole batch (src/trpo.py:242 `pi.ob_rms.update(ob)`): k_vf_rms on a grid sized to the batch — one launch
constexpr int RMS_UPDATE_BLOCKS = 256;              // (the last block adds the partials up column by column: 32 rounds of 8 loads)
extern "C" size_t dm_rms_scratch_bytes(void) { return (size_t)RMS_UPDATE_BLOCKS * 2 * dmv::OB * sizeof(double) + 64; }
extern "C" int dm_rms_update(const float* ob, int32_t n, double* rms_sum, double* rms_sumsq, double* rms_count, float* rms_mean, float* rms_std,
                             void* scratch, void* hip_stream) {
  if (!ob || n < 1 || !rms_sum || !rms_sumsq || !rms_count || !rms_mean || !rms_std || !scratch) return fail(DM_EINVAL, "dm_rms_update: bad argument");
  if (pg_set_device(ob)) return fail(DM_EHIP, "dm_rms_update: hipSetDevice failed");
  hipStream_t st = (hipStream_t)hip_stream;
  int blocks = (n + 255) / 256;                               // >= 64 rows per row group of a block
  if (blocks > RMS_UPDATE_BLOCKS) blocks = RMS_UPDATE_BLOCKS;
  double* part = (double*)scratch;
  unsigned* ticket = (unsigned*)(part + (size_t)RMS_UPDATE_BLOCKS * 2 * dmv::OB);
  HIPCHK(hipMemsetAsync(ticket, 0, sizeof(unsigned), st));
  hipLaunchKernelGGL(dmv::k_vf_rms, dim3(blocks), dim3(256), 0, st, ob, (int)n, part, ticket, rms_sum, rms_sumsq, rms_count, rms_mean, rms_std);
  HIPCHK(hipGetLastError());
  return DM_OK;
}
// ---- policy half of the TRPO update (csrc/pg_kernel.h) -------------------------------------------------------------------------
static int pg_set_device(const void* p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) == hipSuccess) { if (hipSetDevice(at.device) != hipSuccess) return DM_EHIP; }
  else (void)hipGetLastError();
  return DM_OK;
}
// one block per CU; `max_blocks` (0: all of them) leaves CUs to a kernel of another stream (the value fit beside the policy step)
static int pg_blocks(int ntiles, int max_blocks) {
  const int cap = (max_blocks > 0 && max_blocks < dmg::MAX_BLOCKS) ? max_blocks : dmg::MAX_BLOCKS;
  return ntiles < cap ? ntiles : cap;
}
extern "C" int dm_pg_param_count(void) { return dmg::NP; }
extern "C" size_t dm_pg_scratch_bytes(void) { return (size_t)dmg::MAX_BLOCKS * dmg::NPAD * sizeof(float) + (size_t)dmg::MAX_BLOCKS * 2 * sizeof(double) + 256; }
extern "C" int dm_pg_losses(const float* ob, int32_t n, const float* ac, const float* atarg, float* old_mean, const float* old_logstd, int32_t write_old,
                            const float* theta, const float* rms_mean, const float* rms_std, double entcoeff, int32_t with_grad,
                            float* out_grad, double* out_losses, void* scratch, void* hip_stream, int32_t max_blocks) {
  if (!ob || !ac || !atarg || !old_mean || !old_logstd || !theta || !rms_mean || !rms_std || !out_losses || !scratch || n < 1 || (with_grad && !out_grad) || max_blocks < 0)
    return fail(DM_EINVAL, "dm_pg_losses: bad argument");
  if (pg_set_device(theta)) return fail(DM_EHIP, "dm_pg_losses: hipSetDevice failed");
  hipStream_t st = (hipStream_t)hip_stream;
  const int ntiles = (n + dmg::SB - 1) / dmg::SB, nblk = pg_blocks(ntiles, max_blocks);
  float* partial = (float*)scratch;
  double* lpart = (double*)((char*)scratch + (((size_t)dmg::MAX_BLOCKS * dmg::NPAD * sizeof(float) + 255) / 256) * 256);
  if (with_grad)
    hipLaunchKernelGGL(dmg::k_pg<dmg::MODE_GRAD>, dim3(nblk), dim3(256), 0, st, ob, 1, (int)n, ac, atarg, old_mean, old_logstd, (int)write_old, theta,
                       (const float*)nullptr, rms_mean, rms_std, 1.0f / (float)n, partial, lpart);
  else
    hipLaunchKernelGGL(dmg::k_pg<dmg::MODE_LOSS>, dim3(nblk), dim3(256), 0, st, ob, 1, (int)n, ac, atarg, old_mean, old_logstd, (int)write_old, theta,
                       (const float*)nullptr, rms_mean, rms_std, 1.0f / (float)n, partial, lpart);
  hipLaunchKernelGGL(dmg::k_pg_reduce, dim3((dmg::NP + 255) / 256), dim3(256), 0, st, (const float*)partial, (const double*)lpart, nblk,
                     with_grad ? (int)dmg::MODE_GRAD : (int)dmg::MODE_LOSS, (float)entcoeff, (const float*)nullptr, 1.0 / (double)n, out_grad, out_losses);
  HIPCHK(hipGetLastError());
  return DM_OK;
}
extern "C" int dm_pg_fvp(const float* ob, int32_t stride, int32_t n, const float* theta, const float* v, const float* rms_mean, const float* rms_std,
                         float* out_fv, void* scratch, void* hip_stream, int32_t max_blocks) {
  if (!ob || !theta || !v || !rms_mean || !rms_std || !out_fv || !scratch || n < 1 || stride < 1 || max_blocks < 0) return fail(DM_EINVAL, "dm_pg_fvp: bad argument");
  if (pg_set_device(theta)) return fail(DM_EHIP, "dm_pg_fvp: hipSetDevice failed");
  hipStream_t st = (hipStream_t)hip_stream;
  const int ntiles = (n + dmg::SB - 1) / dmg::SB, nblk = pg_blocks(ntiles, max_blocks);
  float* partial = (float*)scratch;
  double* lpart = (double*)((char*)scratch + (((size_t)dmg::MAX_BLOCKS * dmg::NPAD * sizeof(float) + 255) / 256) * 256);
  hipLaunchKernelGGL(dmg::k_pg<dmg::MODE_FVP>, dim3(nblk), dim3(256), 0, st, ob, (int)stride, (int)n, (const float*)nullptr, (const float*)nullptr,
                     (float*)nullptr, (const float*)nullptr, 0, theta, v, rms_mean, rms_std, 1.0f / (float)n, partial, lpart);
  hipLaunchKernelGGL(dmg::k_pg_reduce, dim3((dmg::NP + 255) / 256), dim3(256), 0, st, (const float*)partial, (const double*)lpart, nblk, (int)dmg::MODE_FVP,
                     0.0f, v, 1.0 / (double)n, out_fv, (double*)nullptr);
  HIPCHK(hipGetLastError());
  return DM_OK;
}
extern "C" int dm_batch_redo_total(dm_batch* b, int64_t* out) {
  if (!b || !out) return fail(DM_EINVAL, "dm_batch_redo_total: null argument");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  int v[8] = {0};
  HIPCHK(hipMemcpyAsync(v, b->B.redo_why, sizeof v, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  for (int k = 0; k < 8; k++) out[k] = v[k];      /* [0] total; [1..5] by reason: candidates, box slots, contacts, rows, PGS cost test */
  return DM_OK;
}
extern "C" int dm_batch_queue_stats(dm_batch* b, int64_t* out) {
  if (!b || !out) return fail(DM_EINVAL, "dm_batch_queue_stats: null argument");
  out[0] = b->queue_flushes; out[1] = b->queue_steps; out[2] = (int64_t)b->q.size();
  return DM_OK;
}
extern "C" int dm_batch_join(dm_batch* b) {
  if (!b) return fail(DM_EINVAL, "null batch");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  return DM_OK;
}
extern "C" int dm_batch_sync(dm_batch* b) {
  if (!b) return fail(DM_EINVAL, "null batch");
  HIPCHK(hipSetDevice(b->device));
  if (settle(b)) return fail(DM_EHIP, "pipeline join failed");
  HIPCHK(hipStreamSynchronize(b->stream));
  return DM_OK;
}
