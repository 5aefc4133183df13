// env_step.h — per-environment drivers built on env_kernel.h: the RK4 step (mj_step), the DPEnv epilogue
// (obs / reward / done, src/dp_env_v3.py:115-132), set_state + forward, and the reset variants
// (src/dp_env_v3.py:148-164).  One wavefront per environment; `lane` is the lane id, `env` the env index.
#pragma once

#include "env_kernel.h"

namespace dm {

// device pointers to the batch's state (row-major [N, width] arrays) and its configuration
template <class R>
struct Batch {
  R* qpos;      // [N,35]
  R* qvel;      // [N,34]
  R* qws;       // [N,34] qacc_warmstart
  R* time;      // [N]
  R* ctrl;      // [N,28] last raw ctrl
  R* xipos;     // [N,14,3]
  R* comz;      // [N]
  int* frame_idx;   // [N] DPEnv.idx_curr
  int* frame_init;  // [N] DPEnv.idx_init
  int* ncon;        // [N]
  int* nefc;        // [N]
  int* cong;        // [N,MAXEFC,2]
  R* aovf;          // [N, AOVF_COLS, 64] overflow columns of A for the register-tier kernel (nullptr if unused)
  int* status;      // [N]
  int* solver_iter; // [N]
  int* episode;     // [N]
  int* order;       // [N] dispatch order: workgroup w steps env order[w] (nullptr: identity) — costly envs first shortens the tail
  // self-ordering per-step launches (dispatch_env / order_ticket below): every env stepped by launch t takes a ticket in the bucket of its cost
  // key; launch t + 1 turns (bucket counts, bucket lists) into its dispatch order on the fly — no ordering kernel between two steps.
  // All null: off.  The pointers are those of ONE launch (one pipelined part, one phase): the host rotates three phases per part.
  const int* ord_in;       // [64] bucket counts the previous launch of this part left (nullptr: no order yet -> `order` / identity)
  const int* ordl_in;      // its bucket lists: env of (bucket k, ticket p) at ordl_in[k * ord_stride + p]
  int* ord_out;            // [64] bucket counters this launch's envs take their tickets from (zero when the launch starts)
  int* ordl_out;           // this launch's bucket lists
  int* ord_zero;           // [64] the counters the NEXT launch will count into: cleared by this launch's first workgroup
  int ord_stride;
  int* cycle;       // [N] completed motion cycles since the episode started (imitation reward: root advance of the reference)
  R* kin;           // [N, KIN_DOUBLES] kinematics of the state an env was left in (see save_kin), valid where kin_ok[env] != 0
  unsigned char* kin_ok;   // [N]
  int* redo_list;          // [N] envs the four-envs-per-wave kernel could not step (a capacity of slot_kernel.h exceeded): re-stepped by the one-env kernel
  int* redo_count;         // [DM_MAX_PIPELINE] one counter per pipelined sub-batch (its list starts at redo_list[first env of the sub-batch])
  int* redo_why;           // [8] diagnostic tallies of the reasons (see slot_step.h)
  const R* mocap_cfg;  // [F,35]
  const R* mocap_vel;  // [F,34]
  const R* imit_table; // [F,112] reference feature rows of the 5-term imitation reward (nullptr: not provided)
  const R* imit_pdev;  // the same 32 parameters in device memory (the reward indexes them per lane)
  R imit_params[32];   // joint weights [12], root weight, cycle shift x y, loop flag, end-effector bodies [4], offsets [4][3]
  R mocap_dt;          // duration of a mocap frame (MocapDM.dt): dp_env_v1's update interval (reward mode 4)
  int n_frames;
  int n_envs;
  int env_offset;      // global id of env 0 of this shard (multi-GPU: RNG streams do not depend on the sharding)
  int reward_mode, autoreset, action_mode;
  int diag;            // 1: k_step also stores the per-step diagnostics (xipos, contact geom list); 0: state, obs and the row counts only
  unsigned long long seed;
};

// the same batch for code behind a real call (slot_step.h): there the struct arrives through a generic pointer and so would its members
template <class R>
DM_DEV Batch<R> global_members(Batch<R> b) {
  using dmw::in_global;          // (the members are wave-uniform: loaded through a uniform pointer)
  b.qpos = in_global(b.qpos); b.qvel = in_global(b.qvel); b.qws = in_global(b.qws); b.time = in_global(b.time); b.ctrl = in_global(b.ctrl);
  b.xipos = in_global(b.xipos); b.comz = in_global(b.comz); b.frame_idx = in_global(b.frame_idx); b.frame_init = in_global(b.frame_init);
  b.ncon = in_global(b.ncon); b.nefc = in_global(b.nefc); b.cong = in_global(b.cong); b.aovf = in_global(b.aovf); b.status = in_global(b.status);
  b.solver_iter = in_global(b.solver_iter); b.episode = in_global(b.episode); b.order = in_global(b.order); b.cycle = in_global(b.cycle);
  b.kin = in_global(b.kin); b.kin_ok = in_global(b.kin_ok); b.redo_list = in_global(b.redo_list); b.redo_count = in_global(b.redo_count);
  b.redo_why = in_global(b.redo_why); b.mocap_cfg = in_global(b.mocap_cfg); b.mocap_vel = in_global(b.mocap_vel);
  b.imit_table = in_global(b.imit_table); b.imit_pdev = in_global(b.imit_pdev);
  // (the ord_* members stay as they are: code behind a call runs inside horizon launches, which take no tickets — the members are null there)
  return b;
}

// ---- self-ordering per-step launches ---------------------------------------------------------------------------------------------
// Dispatch order of a per-step launch: environments with more constraint rows (a good proxy for their step time: 0.32 .. 0.72 M shader
// ticks from 0 to 32 rows) first — 4 096 envs are two rounds of resident one-env waves, so a launch ends when the last wave of round two
// does, and longest-first list scheduling trims that tail (measured: -9 % kernel time).  Rounds 1-4 computed the order with a counting-sort
// kernel between two step launches (k_order / k_order_wave: 9 % of the GPU time of the one-launch-per-call form, all of it on the stream's
// critical path).  Now the step kernels do it themselves: an env's last lane-0 act in launch t is to take a ticket in the bucket of its cost
// key (order_ticket: one atomic + one store); a workgroup of launch t + 1 reads the 64 bucket counts (one coalesced load), finds the bucket
// its dispatch position falls into by a wave scan, and reads its env from that bucket's list (dispatch_env).  The order inside a bucket is
// the order of arrival — arbitrary, as before: results never depend on the dispatch order.
// cost key of an environment: constraint rows + a quarter of the PGS sweeps of its last evaluation, 64 buckets
#ifndef DM_ORDER_KEY
#define DM_ORDER_KEY(nefc, iter) ((nefc) + ((iter) >> 2))
#endif
constexpr int ORD_BUCKETS = 64;
DM_DEV int order_bucket(int nefc, int iter) { const int k = DM_ORDER_KEY(nefc, iter); return k < 0 ? 0 : (k > ORD_BUCKETS - 1 ? ORD_BUCKETS - 1 : k); }
// one lane per stepped env, once per launch
template <class R>
DM_DEV void order_ticket(const Batch<R>& B, int env, int nefc, int iter) {
  if (!B.ord_out) return;
  const int k = order_bucket(nefc, iter);
  const int p = dmw::global_counter_next(B.ord_out + k);
  B.ordl_out[(size_t)k * B.ord_stride + p] = env;
}
// The envs at the dispatch positions first + r0 .. first + r0 + NPOS - 1 of a part of `count` envs starting at `first` (wave-collective;
// NPOS = 1: one env per wave, 4: one per slot).  Buckets in DESCENDING key order.  Falls back to B.order / the identity when the previous
// launch left no tickets — or not exactly `count` of them (cannot happen while the host's bookkeeping is right; a permutation is never
// assumed on faith: stepping an env twice would corrupt it).
template <int NPOS, class R>
DM_DEV void dispatch_env(const Batch<R>& B, int first, int count, int r0, int lane, bool first_group, int* env_out) {
  if (B.ord_zero && first_group) B.ord_zero[lane] = 0;
  bool have = false;
  int c = 0, start = 0;
  if (B.ord_in) {
    c = B.ord_in[lane];                               // bucket `lane`
    int incl = c;                                     // sum over the buckets >= lane
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = dmw::shfl_i(incl, lane + d < 64 ? lane + d : lane); if (lane + d < 64) incl += o; }
    start = incl - c;                                 // envs in buckets with a larger key
    have = dmw::bcast_i(incl, 0) == count;
  }
#pragma unroll
  for (int j = 0; j < NPOS; j++) {
    int r = r0 + j;
    if (r > count - 1) r = count - 1;                 // (spare slots of the last wave repeat the last env)
    int e = first + r;
    if (have) {
      const unsigned long long m = dmw::ballot(c > 0 && start <= r && r < start + c);
      const int k = __builtin_ctzll(m | (1ull << 63));
      e = B.ordl_in[(size_t)k * B.ord_stride + (r - dmw::bcast_i(start, k))];
    } else if (B.order) e = B.order[e];
    env_out[j] = e;
  }
}

// counter-based RNG: splitmix64 finaliser over (seed, global env, episode, k) -> U[0,1)
DM_DEV unsigned long long mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
DM_DEV double rng_uniform(unsigned long long seed, int genv, int episode, int k) {
  unsigned long long h = mix64(seed ^ mix64((unsigned long long)(unsigned)genv * 0x100000001B3ull + 0x1234567ull));
  h = mix64(h ^ ((unsigned long long)(unsigned)episode << 32 | (unsigned)k));
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

// [MJ mj_integratePos] s.qpos <- x0q (+) h * dv, with dv read from s.ua.f.tau (scratch).  Caller syncs before and after.
template <class R>
DM_DEV void integrate_pos(Shared<R>& s, const R* x0q, int lane, R h) {
  if (lane < 3) s.qpos[lane] = x0q[lane] + h * s.ua.f.tau[lane];
  else if (lane == 3) {
    R ax[3] = {s.ua.f.tau[3], s.ua.f.tau[4], s.ua.f.tau[5]}, q[4] = {x0q[3], x0q[4], x0q[5], x0q[6]}, qr[4];
    const R angle = h * normalize3(ax);
    if (angle == R(0)) { qr[0] = 1; qr[1] = qr[2] = qr[3] = 0; } else axisangle2quat(qr, ax, angle);
    normalize4(q);
    quat_mul(q, q, qr);
    s.qpos[3] = q[0]; s.qpos[4] = q[1]; s.qpos[5] = q[2]; s.qpos[6] = q[3];
  } else if (lane >= 7 && lane < NQ) s.qpos[lane] = x0q[lane] + h * s.ua.f.tau[lane - 1];
}

// x0q/x0v: LDS copies of the state at the start of the step (kept outside `Shared`'s scratch regions)
template <class R>
struct StepScratch {
  R x0q[36], x0v[NV];
  R vprev[NV], aprev[NV], sumv[NV], suma[NV];   // RK accumulators parked in LDS while forward() needs the registers
};

// [MJ mj_step, integrator RK4] on the state in s.qpos / s.qvel / s.qws / s.act.  Returns with the new state in
// s.qpos / s.qvel / s.qws and the derived quantities of the 4th stage evaluation in `s` (as sim.data after sim.step()).
template <class R, int ROWS = MAXEFC, bool PROF = false>
DM_DEV bool rk4_step(const DevModel<R>& M, Shared<R>& s, StepScratch<R>& x, int lane, const LaneTopo& lt, long long* prof = 0, bool kin0 = false) {
  const R h = M.timestep;
  const R A[3] = {R(0.5), R(0.5), R(1)};
  const R Bw[4] = {R(1) / 6, R(1) / 3, R(1) / 3, R(1) / 6};
  if (lane < NQ) x.x0q[lane] = s.qpos[lane];
  if (lane < NV) { const R v0 = s.qvel[lane]; x.x0v[lane] = v0; x.vprev[lane] = v0; x.sumv[lane] = 0; x.suma[lane] = 0; }
  dmw::sync();
  for (int i = 0; i < 4; i++) {   // single call site of forward(): the four evaluations share one copy of the code
    if (i > 0) {
      const R c = A[i - 1];
      if (lane < NV) s.ua.f.tau[lane] = c * x.vprev[lane];   // dX (position part)
      dmw::sync();
      integrate_pos(s, x.x0q, lane, h);
      if (lane < NV) { const R vi = x.x0v[lane] + h * (c * x.aprev[lane]); x.vprev[lane] = vi; s.qvel[lane] = vi; }
      dmw::sync();
    }
    forward<R, ROWS, PROF>(M, s, lane, lt, (const DebugOut*)0, prof, i == 0 && kin0);
    if (lane < NV) {
      const R a = s.ua.f.qacc[lane];
      x.aprev[lane] = a; x.sumv[lane] += Bw[i] * x.vprev[lane]; x.suma[lane] += Bw[i] * a;
    }
  }
  if (lane < NV) s.ua.f.tau[lane] = x.sumv[lane];
  dmw::sync();
  integrate_pos(s, x.x0q, lane, h);
  if (lane < NV) { s.qvel[lane] = x.x0v[lane] + h * x.suma[lane]; s.qws[lane] = x.aprev[lane]; }
  dmw::sync();
  return true;
}

// load one env's row from HBM (coalesced: lane k reads element k) and turn the action into actuator forces
template <class R>
DM_DEV void load_env(const DevModel<R>& M, const Batch<R>& B, Shared<R>& s, int env, int lane, const double* action) {
  if (lane < NQ) s.qpos[lane] = B.qpos[(size_t)env * NQ + lane];
  if (lane < NV) { s.qvel[lane] = B.qvel[(size_t)env * NV + lane]; s.qws[lane] = B.qws[(size_t)env * NV + lane]; s.act[lane] = 0; }
  if (lane == 0) { s.status = 0; s.nefc = 0; s.ncon = 0; s.solver_iter = 0; s.aovf = B.aovf ? B.aovf + (size_t)env * AOVF_COLS * 64 : (R*)0; }
  dmw::sync();
  if (action && lane < NU) {
    R a = (R)action[(size_t)env * NU + lane];   // (the C ABI's buffers are float64 whatever the arithmetic type R)
    if (B.action_mode == 1) {  // P-control towards the current mocap frame (src/env_torque_test.py:14-20)
      const int idx = B.frame_idx[env];
      a += R(0.8) * (B.mocap_cfg[(size_t)idx * NQ + 7 + lane] - s.qpos[7 + lane]);
    } else if (B.action_mode == 2) {  // PD torque kp*dq + kd*dv written to ctrl (src/mujoco/setting_states.py:207-226)
      const int idx = B.frame_idx[env];
      a += M.kp[lane + 6] * (B.mocap_cfg[(size_t)idx * NQ + 7 + lane] - s.qpos[7 + lane]) +
           M.kd[lane + 6] * (B.mocap_vel[(size_t)idx * NV + 6 + lane] - s.qvel[6 + lane]);
    }
    B.ctrl[(size_t)env * NU + lane] = a;  // data.ctrl keeps the unclamped value
    const int d = lane + 6;
    s.act[d] = M.gear[d] * clampr(a, M.ctrl_lo[d], M.ctrl_hi[d]);
  }
  dmw::sync_mem();
}

template <class R>
DM_DEV R com_z(const DevModel<R>& M, const Shared<R>& s) {  // src/dp_env_v3.py:134-139
  R sz = 0, sm = 0;
  for (int b = 0; b < NB; b++) { sz += M.body_mass[b] * s.xipos[b][2]; sm += M.body_mass[b]; }
  return sz / sm;
}

// `full`: also the diagnostic arrays (sim.data.xipos, the contact geom list): 848 B per env that nothing on the training path
// reads back; set_state / reset / debug always store them, the step only when DM_OPT_DIAGNOSTICS is on.
template <class R>
DM_DEV void store_derived(const Batch<R>& B, const DevModel<R>& M, Shared<R>& s, int env, int lane, bool full = true) {
  if (full) {
    if (lane < NB * 3) B.xipos[(size_t)env * NB * 3 + lane] = s.xipos[lane / 3][lane % 3];
    for (int k = lane; k < MAXEFC * 2; k += 64) {
      const int c = k >> 1;
      B.cong[(size_t)env * MAXEFC * 2 + k] = (c < s.ncon && c < MAXEFC) ? s.cong[c][k & 1] : -1;
    }
  }
  if (lane == 0) {
    B.comz[env] = com_z(M, s);
    B.ncon[env] = s.ncon; B.nefc[env] = s.nefc; B.status[env] = s.status; B.solver_iter[env] = s.solver_iter;
  }
}

template <class R>
DM_DEV void store_state(const Batch<R>& B, Shared<R>& s, int env, int lane) {
  if (lane < NQ) B.qpos[(size_t)env * NQ + lane] = s.qpos[lane];
  if (lane < NV) { B.qvel[(size_t)env * NV + lane] = s.qvel[lane]; B.qws[(size_t)env * NV + lane] = s.qws[lane]; }
}

// DPEnv.reference_state_init for one env: `idx` is the drawn mocap frame.  dp_env_v3 (src/dp_env_v3.py:67-71) starts its frame
// cursor AT the drawn frame; dp_env_v2 (src/dp_env_v2.py:68-70) keeps the draw in idx_init and counts steps from 0 in idx_curr
// (its target frame is (idx_curr + idx_init) % F, :128-129); dp_env_v1 does the same (src/dp_env_v1.py:58-60,94-96).
template <class R>
DM_DEV void set_frame(const Batch<R>& B, int env, int idx) {
  B.frame_init[env] = idx;
  B.frame_idx[env] = (B.reward_mode == REW_V2_POSE || B.reward_mode == REW_V1_QUAT) ? 0 : idx;
  B.cycle[env] = 0;
}

// reset variants (src/dp_env_v3.py:67-71,148-164).  mode 0: RSI, 1: noisy init pose, 2: qpos0 / zero velocity.
// `hard` = sim.reset() semantics: time = 0, qacc_warmstart = 0.  Writes s.qpos / s.qvel (/ s.qws) and frame indices.
// Modes 0 and 1 with `hard` are the two episode starts of the reference's trainer: env.reset() = sim.reset() + reset_model()
// (RSI: draws the frame, copies the mocap state), optionally followed by reset_model_init() which overrides the STATE only
// (src/trpo.py:78-79) — so mode 1 draws the frame as well; the frame-indexed rewards then start from it.
template <class R>
DM_DEV void reset_env(const DevModel<R>& M, const Batch<R>& B, Shared<R>& s, int env, int lane, int mode, int hard) {
  const int ep = B.episode[env];
  const int genv = B.env_offset + env;
  int idx = (int)(rng_uniform(B.seed, genv, ep, 0) * (double)B.n_frames);
  if (idx >= B.n_frames) idx = B.n_frames - 1;
  if (mode == 0) {
    if (lane < NQ) s.qpos[lane] = B.mocap_cfg[(size_t)idx * NQ + lane];
    if (lane < NV) s.qvel[lane] = B.mocap_vel[(size_t)idx * NV + lane];
  } else if (mode == 1) {
    if (lane < NQ) s.qpos[lane] = M.qpos0[lane] + (R)((rng_uniform(B.seed, genv, ep, 1 + lane) * 2.0 - 1.0) * 0.01);
    if (lane < NV) s.qvel[lane] = (R)((rng_uniform(B.seed, genv, ep, 64 + lane) * 2.0 - 1.0) * 0.01);
  } else {
    if (lane < NQ) s.qpos[lane] = M.qpos0[lane];
    if (lane < NV) s.qvel[lane] = 0;
  }
  if (hard) {
    if (lane < NV) s.qws[lane] = 0;
    if (lane == 0) B.time[env] = 0;
  }
  dmw::sync_mem();
  if (lane == 0) {
    B.episode[env] = ep + 1;
    if (mode == 0 || (mode == 1 && hard)) set_frame(B, env, idx); else B.cycle[env] = 0;
  }
}

// ---- 5-term imitation reward (code.md:1017-1143; feature layout: deepmimic_mujoco_amd/imitation.py) -----------------------
// [upstream] cMathUtil::QuatDiffTheta: rotation angle of q1 * conj(q0), normalised to [-pi, pi]
template <class R>
DM_DEV R quat_diff_theta(const R* q0, const R* q1) {
  const R c[4] = {q0[0], -q0[1], -q0[2], -q0[3]};
  R dq[4];
  quat_mul(dq, q1, c);
  const R w = dq[0] > R(1) ? R(1) : (dq[0] < R(-1) ? R(-1) : dq[0]);
  if (sqrt(fmax(R(0), 1 - w * w)) <= R(1e-6)) return 0;
  const R th = 2 * acos_once(w);
  return th > R(M_PI) ? th - R(2 * M_PI) : th;
}
// The simulated state's features are formed from the FK of the INTEGRATED state (one extra kinematics pass; the 4th-stage
// quantities `is_done` reads have been stored by then) and compared with row `ref` of the reference table.
// Lane = body - 1 as in the kinematics stage, whose by-products ARE the joint features (child-in-parent quaternion, hinge
// axes in the parent frame): lane 0 root, lanes 1..12 joint groups; lanes 13..16 end effectors; lanes 0..33 again one dof
// each for the linear momentum  p = sum_d qvel_d (m_sub(d) lin_d + ang_d x S_sub(d))  with the subtree mass / first moment the
// composite-inertia pass has just left in LDS.  One acos, one exp per lane; no other transcendental.
template <class R>
DM_DEV R imitation_reward(const DevModel<R>& M, const Batch<R>& B, Shared<R>& s, int lane, const LaneTopo& lt, const R* ref, R shx, R shy) {
  R qloc[4], aloc[3][3];
  stage_kinematics(M, s, lane, lt, qloc, aloc);          // ends with a sync
  const R* P = B.imit_pdev;                               // device copy of the parameter block (per-lane indexing)
  R rq[4] = {s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]};
  normalize4(rq);
  R pose = 0, vel = 0, eff = 0, root = 0, mx = 0, my = 0, mz = 0;
  if (lane < 13) {
    const int g = lane - 1;                               // joint group (lane 0: the root, weight slot 12)
    const int da = dmw::launder(lt.dofadr), nd = dmw::launder(lt.dofnum);
    const bool isroot = lane == 0, ball = !isroot && nd == 3;
    const R* rquat = ref + (isroot ? 3 : 13 + 4 * g);
    const R ident[4] = {1, 0, 0, 0};
    R q0[4], q1[4];
    for (int k = 0; k < 4; k++) { q0[k] = isroot ? rq[k] : (ball ? qloc[k] : ident[k]); q1[k] = (isroot || ball) ? rquat[k] : ident[k]; }
    const R th = quat_diff_theta(q0, q1);
    R pe = th * th, ve = 0;
    if (isroot) {
      R wv[3];
      const R wloc[3] = {s.qvel[3], s.qvel[4], s.qvel[5]};
      quat_rot(wv, rq, wloc);                             // free-joint angular velocity is body-local
      R dv2 = 0, dp2 = 0;
      for (int k = 0; k < 3; k++) { const R a = ref[10 + k] - wv[k]; ve += a * a; const R c = ref[7 + k] - s.qvel[k]; dv2 += c * c; }
      const R p1[3] = {ref[0] + shx, ref[1] + shy, ref[2]};
      for (int k = 0; k < 3; k++) { const R a = s.qpos[k] - p1[k]; dp2 += a * a; }
      root = dp2 + R(0.1) * pe + R(0.01) * dv2 + R(0.001) * ve;
    } else if (ball) {
      R wl[3] = {0, 0, 0};                                // child = R1 R2 R3;  w = sum_k (R1..R(k-1) a_k) rate_k
      for (int k = 0; k < 3; k++) { const R rate = s.qvel[da + k]; wl[0] += aloc[k][0] * rate; wl[1] += aloc[k][1] * rate; wl[2] += aloc[k][2] * rate; }
      for (int k = 0; k < 3; k++) { const R w = ref[61 + 3 * g + k] - wl[k]; ve += w * w; }
    } else {
      const R a = ref[13 + 4 * g] - s.qpos[da + 1], w = ref[61 + 3 * g] - s.qvel[da];
      pe = a * a; ve = w * w;
    }
    const R wj = P[isroot ? 12 : g];
    pose = wj * pe; vel = wj * ve;
  } else if (lane < 17) {
    const int e = lane - 13, b = (int)P[16 + e];
    const R ex[3] = {1, 0, 0};
    R fwd[3], p[3], rel[3];
    quat_rot(fwd, rq, ex);
    const R hn = sqrt(fwd[0] * fwd[0] + fwd[1] * fwd[1]);   // heading about the vertical: (cos, sin) of atan2(fwd_y, fwd_x)
    const R c = hn > R(0) ? fwd[0] / hn : R(1), sn = hn > R(0) ? fwd[1] / hn : R(0);
    mat_vec(p, s.xmat[b], P + 20 + 3 * e);
    for (int k = 0; k < 3; k++) { p[k] += s.xpos[b][k]; rel[k] = p[k] - s.qpos[k]; }
    rel[2] = p[2];                                        // height above the ground plane
    const R f0[3] = {c * rel[0] + sn * rel[1], -sn * rel[0] + c * rel[1], rel[2]};
    for (int k = 0; k < 3; k++) { const R a = ref[97 + 3 * e + k] - f0[k]; eff += a * a; }
  }
  if (lane < NV) {                                        // momentum carried by dof `lane`: its whole subtree moves with it
    const int b = TOPO.dof_body[lane];
    const R* cb = s.ub.i.crb[b];
    const R* cd = s.cdof[lane];
    const R qd = s.qvel[lane], ms = cb[9];
    R axs[3];
    cross3(axs, cd, cb + 6);
    mx = qd * (ms * cd[3] + axs[0]); my = qd * (ms * cd[4] + axs[1]); mz = qd * (ms * cd[5] + axs[2]);
  }
  pose = dmw::wave_sum(pose); vel = dmw::wave_sum(vel); eff = dmw::wave_sum(eff) / 4; root = dmw::bcast(root, 0);
  mx = dmw::wave_sum(mx) / M.total_mass; my = dmw::wave_sum(my) / M.total_mass; mz = dmw::wave_sum(mz) / M.total_mass;
  const R dc[3] = {ref[109] - mx, ref[110] - my, ref[111] - mz};
  const R com = R(0.1) * dot3(dc, dc);
  // the five terms, one lane each:  0.5 e^(-2 pose) + 0.05 e^(-0.1 vel) + 0.15 e^(-40 eff) + 0.2 e^(-5 root) + 0.1 e^(-10 com)
  const R arg = lane == 0 ? R(-2) * pose : lane == 1 ? R(-0.1) * vel : lane == 2 ? R(-40) * eff : lane == 3 ? R(-5) * root : R(-10) * com;
  const R wgt = lane == 0 ? R(0.5) : lane == 1 ? R(0.05) : lane == 2 ? R(0.15) : lane == 3 ? R(0.2) : R(0.1);
  R term = 0;
  if (lane < 5) term = wgt * exp_once(arg);
  return dmw::wave_sum(term);
}

// ---- dp_env_v1's reward (src/dp_env_v1.py:82-141) on the hinge-triple model, from the same features as the 5-term reward ------------
//   err_pose = sum_j JOINT_WEIGHT_j |rotation angle of q_sim^* q_ref|   (MujocoInterface.calc_config_errs, mujoco_interface.py:169-190;
//              |angle difference| for the 1-hinge joints; un-normalised weights = P[g] / P[12])
//   err_vel  = L1 distance of the angular rates (root, joints) from the clip's rates frame k -> k + 1 (row `refv`)       (:205-210)
//   err_root = L1 distance of the root positions                                                                          (:192-199)
//   r = 0.5 e^(-2 err_pose) + 0.05 e^(-0.1 err_vel) + 0.2 e^(-5 err_root)
// Lane = body - 1 as in imitation_reward (lane 0 root, lanes 1..12 joint groups); one acos per lane, three exps on three lanes.
template <class R>
DM_DEV R v1_reward(const DevModel<R>& M, const Batch<R>& B, Shared<R>& s, int lane, const LaneTopo& lt, const R* ref, const R* refv) {
  R qloc[4], aloc[3][3];
  stage_kinematics(M, s, lane, lt, qloc, aloc);          // ends with a sync
  const R* P = B.imit_pdev;
  R pose = 0, vel = 0, root = 0;
  if (lane < 13) {
    const int g = lane - 1;
    const int da = dmw::launder(lt.dofadr), nd = dmw::launder(lt.dofnum);
    const bool isroot = lane == 0, ball = !isroot && nd == 3;
    R rq[4] = {s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]};
    normalize4(rq);
    const R* rquat = ref + (isroot ? 3 : 13 + 4 * g);
    const R ident[4] = {1, 0, 0, 0};
    R q0[4], q1[4];
    for (int k = 0; k < 4; k++) { q0[k] = isroot ? rq[k] : (ball ? qloc[k] : ident[k]); q1[k] = (isroot || ball) ? rquat[k] : ident[k]; }
    R pe = fabs(quat_diff_theta(q0, q1)), ve = 0;
    if (isroot) {
      R wv[3];
      const R wloc[3] = {s.qvel[3], s.qvel[4], s.qvel[5]};
      quat_rot(wv, rq, wloc);
      for (int k = 0; k < 3; k++) { ve += fabs(refv[10 + k] - wv[k]); root += fabs(s.qpos[k] - ref[k]); }
    } else if (ball) {
      R wl[3] = {0, 0, 0};
      for (int k = 0; k < 3; k++) { const R rate = s.qvel[da + k]; wl[0] += aloc[k][0] * rate; wl[1] += aloc[k][1] * rate; wl[2] += aloc[k][2] * rate; }
      for (int k = 0; k < 3; k++) ve += fabs(refv[61 + 3 * g + k] - wl[k]);
    } else {
      pe = fabs(ref[13 + 4 * g] - s.qpos[da + 1]);
      ve = fabs(refv[61 + 3 * g] - s.qvel[da]) + fabs(refv[61 + 3 * g + 1]) + fabs(refv[61 + 3 * g + 2]);   // (the two unused slots of the row are 0)
    }
    pose = (isroot ? R(1) : P[g] / P[12]) * pe; vel = ve;
  }
  pose = dmw::wave_sum(pose); vel = dmw::wave_sum(vel); root = dmw::bcast(root, 0);
  const R arg = lane == 0 ? R(-2) * pose : lane == 1 ? R(-0.1) * vel : R(-5) * root;
  const R wgt = lane == 0 ? R(0.5) : lane == 1 ? R(0.05) : R(0.2);
  R term = 0;
  if (lane < 3) term = wgt * exp_once(arg);
  return dmw::wave_sum(term);
}

// The imitation rewards end the step with a kinematics pass on the state the env is left in — exactly the pass the NEXT step's first
// evaluation starts with.  Its results (body frames, motion axes, inertias: 694 numbers) are parked in the env's HBM strip and read
// back instead of being recomputed: ~25 memory instructions against ~1 000, bit-identical values.  Whoever changes an env's state
// from outside (reset, set_state, field writes) clears its flag.
constexpr int KIN_PER_LANE = (KIN_DOUBLES + 63) / 64;
template <class R>
DM_DEV void save_kin(const Shared<R>& s, R* dst, int lane) {
  static_assert(sizeof(s.xpos) + sizeof(s.xmat) + sizeof(s.xipos) + sizeof(s.cdof) == KIN_A * sizeof(R) && sizeof(s.ub.i) == KIN_B * sizeof(R), "kinematics block layout");
  const R* a = &s.xpos[0][0];
  const R* b = &s.ub.i.sin[0][0];
#pragma unroll
  for (int c = 0; c < KIN_PER_LANE; c++) { const int i = lane + 64 * c; if (i < KIN_DOUBLES) dst[i] = i < KIN_A ? a[i] : b[i - KIN_A]; }
}
template <class R>
DM_DEV void fetch_kin(R* reg, const R* src, int lane) {          // issued at the top of the step, consumed after the state has been loaded
#pragma unroll
  for (int c = 0; c < KIN_PER_LANE; c++) { const int i = lane + 64 * c; reg[c] = i < KIN_DOUBLES ? src[i] : R(0); }
}
template <class R>
DM_DEV void place_kin(Shared<R>& s, const R* reg, int lane) {
  R* a = &s.xpos[0][0];
  R* b = &s.ub.i.sin[0][0];
#pragma unroll
  for (int c = 0; c < KIN_PER_LANE; c++) { const int i = lane + 64 * c; if (i < KIN_A) a[i] = reg[c]; else if (i < KIN_DOUBLES) b[i - KIN_A] = reg[c]; }
}

// DPEnv.step for one environment
// ROWS = columns of A = J M^-1 J^T + R that this instantiation keeps in registers; an evaluation with more constraint rows
// (up to MAXEFC) keeps the remaining columns in the env's global-memory strip s.aovf (see stage_constraint).
template <class R, int ROWS = MAXEFC, bool PROF = false>
DM_DEV bool env_step(const DevModel<R>& M, const Batch<R>& B, Shared<R>& s, StepScratch<R>& x, int env, int lane,
                     const double* action, double* obs, double* reward, unsigned char* done, int n_substeps, long long* prof_out = 0) {
  long long prof[PROF_SLOTS];
  for (int k = 0; k < PROF_SLOTS; k++) prof[k] = 0;
  long long tstart = 0;
  if (PROF) tstart = dmw::clk();
  const LaneTopo lt = lane_topo(lane);
  stage_tables(s, lane);
  const bool kin0 = B.kin && dmw::uniform((int)B.kin_ok[env]) != 0;
  R kreg[KIN_PER_LANE];
  if (kin0) fetch_kin(kreg, B.kin + (size_t)env * KIN_DOUBLES, lane);
  load_env(M, B, s, env, lane, action);
  if (kin0) { place_kin(s, kreg, lane); dmw::sync(); }
  for (int k = 0; k < n_substeps; k++)   // do_simulation(action, n)
    if (!rk4_step<R, ROWS, PROF>(M, s, x, lane, lt, prof, k == 0 && kin0)) { if (lane == 0) order_ticket(B, env, s.nefc, s.solver_iter); return false; }
  bool kin_saved = false;
  const R z = com_z(M, s);
  bool dn = (z < R(0.7)) || (z > R(2.0));
  store_derived(B, M, s, env, lane, B.diag != 0);   // sim.data.* as they stand after sim.step(): 4th-stage quantities
  // reward
  R rew = 1;
  if (B.reward_mode == REW_V3_CONFIG) {          // src/dp_env_v3.py:89-104
    const int idx = B.frame_idx[env];
    R err = 0;
    for (int i = 7; i < NQ; i++) err += fabs(s.qpos[i] - B.mocap_cfg[(size_t)idx * NQ + i]);
    rew = exp_once(-err);
    dmw::sync_mem();
    if (lane == 0) B.frame_idx[env] = (idx + 1) % B.n_frames;
  } else if (B.reward_mode == REW_V2_POSE) {     // src/dp_env_v2.py:116-183
    const int idx = B.frame_idx[env] + 1;
    const int im = (idx + B.frame_init[env]) % B.n_frames;
    R err = 0, acs = 0;
    for (int i = 3; i < NQ; i++) err += fabs(s.qpos[i] - B.mocap_cfg[(size_t)im * NQ + i]);
    for (int u = 0; u < NU; u++) { const R c = B.ctrl[(size_t)env * NU + u]; acs += c * c; }
    rew = exp_once(R(-2) * err) - R(0.1) * acs;
    dmw::sync_mem();
    if (lane == 0) B.frame_idx[env] = idx;
  } else if (B.reward_mode == REW_IMITATION) {   // code.md:1017-1143: the state after the step against frame idx + 1
    int k = dmw::uniform(B.frame_idx[env]) + 1, cyc = dmw::uniform(B.cycle[env]);
    bool ended = false;
    if (k >= B.n_frames) { if (B.imit_params[15] != R(0)) { k = 0; cyc += 1; } else { k = B.n_frames - 1; ended = true; } }
    rew = imitation_reward(M, B, s, lane, lt, B.imit_table + (size_t)k * IMIT_FEAT, cyc * B.imit_params[13], cyc * B.imit_params[14]);
    dn = dn || ended;                            // a "Loop: none" clip holds its last frame and ends the episode there
    dmw::sync_mem();
    if (lane == 0) { B.frame_idx[env] = k; B.cycle[env] = cyc; }
    if (B.kin && !(dn && B.autoreset)) { save_kin(s, B.kin + (size_t)env * KIN_DOUBLES, lane); kin_saved = true; }   // (after the fence: the stores drain behind the rest of the epilogue)
  }
  else if (B.reward_mode == REW_V1_QUAT) {     // src/dp_env_v1.py:82-158: cursor counts steps, reward every `upd` steps, minus the control cost
    const int idx = dmw::uniform(B.frame_idx[env]) + 1;
    int upd = (int)floor(B.mocap_dt / (M.timestep * n_substeps));
    if (upd < 1) upd = 1;
    R robs = 0;
    if (idx % upd == 0) {
      const int k = (idx / upd + dmw::uniform(B.frame_init[env])) % B.n_frames, kv = k + 1 < B.n_frames ? k + 1 : B.n_frames - 1;
      robs = v1_reward(M, B, s, lane, lt, B.imit_table + (size_t)k * IMIT_FEAT, B.imit_table + (size_t)kv * IMIT_FEAT);
      if (B.kin) { save_kin(s, B.kin + (size_t)env * KIN_DOUBLES, lane); kin_saved = true; }
    }
    R acs = 0;
    for (int u = 0; u < NU; u++) { const R c = B.ctrl[(size_t)env * NU + u]; acs += c * c; }
    rew = robs - R(0.1) * acs;
    dmw::sync_mem();
    if (lane == 0) B.frame_idx[env] = idx;
  }
  if (lane == 0) { B.time[env] += M.timestep * n_substeps; reward[env] = rew; done[env] = dn ? 1 : 0; if (B.kin) B.kin_ok[env] = (kin_saved && !(dn && B.autoreset)) ? 1 : 0; }
  if (dn && B.autoreset) {                        // DummyVecEnv convention: obs of the fresh episode is returned
    dmw::sync_mem();
    reset_env(M, B, s, env, lane, B.autoreset == 1 ? 0 : 1, 1);
  }
  // obs = qpos[7:] (+) qvel[6:]   (src/dp_env_v3.py:62-65), one coalesced 56-wide store
  if (lane < 28) obs[(size_t)env * NOBS + lane] = s.qpos[7 + lane];
  else if (lane < NOBS) obs[(size_t)env * NOBS + lane] = s.qvel[6 + (lane - 28)];
  store_state(B, s, env, lane);
  if (lane == 0) order_ticket(B, env, s.nefc, s.solver_iter);
  if (PROF && lane == 0) {
    prof[5] = dmw::clk() - tstart;
    prof[6] = s.nefc; prof[7] = s.solver_iter;
    for (int k = 0; k < PROF_SLOTS; k++) prof_out[(size_t)env * PROF_SLOTS + k] = prof[k];
  }
  return true;
}

}  // namespace dm
