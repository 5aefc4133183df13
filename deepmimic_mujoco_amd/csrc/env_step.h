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
  const R* mocap_cfg;  // [F,35]
  const R* mocap_vel;  // [F,34]
  int n_frames;
  int n_envs;
  int env_offset;      // global id of env 0 of this shard (multi-GPU: RNG streams do not depend on the sharding)
  int reward_mode, autoreset, action_mode;
  unsigned long long seed;
};

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
DM_DEV bool rk4_step(const DevModel<R>& M, Shared<R>& s, StepScratch<R>& x, int lane, const LaneTopo& lt, long long* prof = 0) {
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
    forward<R, ROWS, PROF>(M, s, lane, lt, (const DebugOut*)0, prof);
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
DM_DEV void load_env(const DevModel<R>& M, const Batch<R>& B, Shared<R>& s, int env, int lane, const R* action) {
  if (lane < NQ) s.qpos[lane] = B.qpos[(size_t)env * NQ + lane];
  if (lane < NV) { s.qvel[lane] = B.qvel[(size_t)env * NV + lane]; s.qws[lane] = B.qws[(size_t)env * NV + lane]; s.act[lane] = 0; }
  if (lane == 0) { s.status = 0; s.nefc = 0; s.ncon = 0; s.solver_iter = 0; s.aovf = B.aovf ? B.aovf + (size_t)env * AOVF_COLS * 64 : (R*)0; }
  dmw::sync();
  if (action && lane < NU) {
    R a = action[(size_t)env * NU + lane];
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
  dmw::sync();
}

template <class R>
DM_DEV R com_z(const DevModel<R>& M, const Shared<R>& s) {  // src/dp_env_v3.py:134-139
  R sz = 0, sm = 0;
  for (int b = 0; b < NB; b++) { sz += M.body_mass[b] * s.xipos[b][2]; sm += M.body_mass[b]; }
  return sz / sm;
}

template <class R>
DM_DEV void store_derived(const Batch<R>& B, const DevModel<R>& M, Shared<R>& s, int env, int lane) {
  if (lane < NB * 3) B.xipos[(size_t)env * NB * 3 + lane] = s.xipos[lane / 3][lane % 3];
  for (int k = lane; k < MAXEFC * 2; k += 64) {
    const int c = k >> 1;
    B.cong[(size_t)env * MAXEFC * 2 + k] = (c < s.ncon && c < MAXEFC) ? s.cong[c][k & 1] : -1;
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

// reset variants (src/dp_env_v3.py:67-71,148-164).  mode 0: RSI, 1: noisy init pose, 2: qpos0 / zero velocity.
// `hard` = sim.reset() semantics: time = 0, qacc_warmstart = 0.  Writes s.qpos / s.qvel (/ s.qws) and frame indices.
template <class R>
DM_DEV void reset_env(const DevModel<R>& M, const Batch<R>& B, Shared<R>& s, int env, int lane, int mode, int hard) {
  const int ep = B.episode[env];
  const int genv = B.env_offset + env;
  if (mode == 0) {
    int idx = (int)(rng_uniform(B.seed, genv, ep, 0) * (double)B.n_frames);
    if (idx >= B.n_frames) idx = B.n_frames - 1;
    if (lane < NQ) s.qpos[lane] = B.mocap_cfg[(size_t)idx * NQ + lane];
    if (lane < NV) s.qvel[lane] = B.mocap_vel[(size_t)idx * NV + lane];
    if (lane == 0) { B.frame_idx[env] = idx; B.frame_init[env] = idx; }
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
  dmw::sync();
  if (lane == 0) B.episode[env] = ep + 1;
}

// DPEnv.step for one environment
// ROWS = columns of A = J M^-1 J^T + R that this instantiation keeps in registers; an evaluation with more constraint rows
// (up to MAXEFC) keeps the remaining columns in the env's global-memory strip s.aovf (see stage_constraint).
template <class R, int ROWS = MAXEFC, bool PROF = false>
DM_DEV bool env_step(const DevModel<R>& M, const Batch<R>& B, Shared<R>& s, StepScratch<R>& x, int env, int lane,
                     const R* action, R* obs, R* reward, unsigned char* done, int n_substeps, long long* prof_out = 0) {
  long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tstart = 0;
  if (PROF) tstart = dmw::clk();
  const LaneTopo lt = lane_topo(lane);
  stage_tables(s, lane);
  load_env(M, B, s, env, lane, action);
  for (int k = 0; k < n_substeps; k++)   // do_simulation(action, n)
    if (!rk4_step<R, ROWS, PROF>(M, s, x, lane, lt, prof)) return false;
  const R z = com_z(M, s);
  const bool dn = (z < R(0.7)) || (z > R(2.0));
  // reward
  R rew = 1;
  if (B.reward_mode == REW_V3_CONFIG) {          // src/dp_env_v3.py:89-104
    const int idx = B.frame_idx[env];
    R err = 0;
    for (int i = 7; i < NQ; i++) err += fabs(s.qpos[i] - B.mocap_cfg[(size_t)idx * NQ + i]);
    rew = exp(-err);
    dmw::sync();
    if (lane == 0) B.frame_idx[env] = (idx + 1) % B.n_frames;
  } else if (B.reward_mode == REW_V2_POSE) {     // src/dp_env_v2.py:116-183
    const int idx = B.frame_idx[env] + 1;
    const int im = (idx + B.frame_init[env]) % B.n_frames;
    R err = 0, acs = 0;
    for (int i = 3; i < NQ; i++) err += fabs(s.qpos[i] - B.mocap_cfg[(size_t)im * NQ + i]);
    for (int u = 0; u < NU; u++) { const R c = B.ctrl[(size_t)env * NU + u]; acs += c * c; }
    rew = exp(R(-2) * err) - R(0.1) * acs;
    dmw::sync();
    if (lane == 0) B.frame_idx[env] = idx;
  }
  store_derived(B, M, s, env, lane);
  if (lane == 0) { B.time[env] += M.timestep * n_substeps; reward[env] = rew; done[env] = dn ? 1 : 0; }
  if (dn && B.autoreset) {                        // DummyVecEnv convention: obs of the fresh episode is returned
    dmw::sync();
    reset_env(M, B, s, env, lane, B.autoreset == 1 ? 0 : 1, 1);
  }
  // obs = qpos[7:] (+) qvel[6:]   (src/dp_env_v3.py:62-65), one coalesced 56-wide store
  if (lane < 28) obs[(size_t)env * NOBS + lane] = s.qpos[7 + lane];
  else if (lane < NOBS) obs[(size_t)env * NOBS + lane] = s.qvel[6 + (lane - 28)];
  store_state(B, s, env, lane);
  if (PROF && lane == 0) {
    prof[5] = dmw::clk() - tstart;
    prof[6] = s.nefc; prof[7] = s.solver_iter;
    for (int k = 0; k < 16; k++) prof_out[(size_t)env * 16 + k] = prof[k];
  }
  return true;
}

}  // namespace dm
