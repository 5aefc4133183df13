// slot_step.h — DPEnv.step for the four-environments-per-wavefront layout of slot_kernel.h: RK4 driver with the accumulators in
// registers (dof d lives in slot lane d % 16, register d / 16), the env epilogue (obs / reward / done, src/dp_env_v3.py:115-132)
// and the auto-reset.  `sl` = lane inside the slot, `env` = the slot's environment, `live` = the slot holds a real environment (the
// last wave of a launch may carry fewer than four: its spare slots recompute the last one and store nothing).
#pragma once

#include "env_step.h"
#include "slot_kernel.h"

namespace dm {

// dof-distributed register vectors: element d of a 34- (35-) vector is r[d / 16] of slot lane d % 16
template <class R> struct DofVec { R r[DOF_PASSES]; };

// [MJ mj_integratePos] s.qpos <- x0q (+) h * dv with dv in s.tau.  Collective (row broadcasts): every lane calls it.
template <class R>
DM_DEV void slot_integrate_pos(SlotShared<R>& s, const DofVec<R>& x0q, int sl, R h) {
  // the root quaternion's four components sit in lanes 3..6 of register 0
  const R q0 = dmw::row_bcast<3>(x0q.r[0]), q1 = dmw::row_bcast<4>(x0q.r[0]), q2 = dmw::row_bcast<5>(x0q.r[0]), q3 = dmw::row_bcast<6>(x0q.r[0]);
  if (sl < 3) s.qpos[sl] = x0q.r[0] + h * s.tau[sl];
  else if (sl == 3) {
    R ax[3] = {s.tau[3], s.tau[4], s.tau[5]}, q[4] = {q0, q1, q2, q3}, qr[4];
    const R angle = h * normalize3(ax);
    if (angle == R(0)) { qr[0] = 1; qr[1] = qr[2] = qr[3] = 0; } else axisangle2quat(qr, ax, angle);
    normalize4(q);
    quat_mul(q, q, qr);
    s.qpos[3] = q[0]; s.qpos[4] = q[1]; s.qpos[5] = q[2]; s.qpos[6] = q[3];
  } else if (sl >= 7) s.qpos[sl] = x0q.r[0] + h * s.tau[sl - 1];
#pragma unroll
  for (int c = 1; c < Q_PASSES; c++) { const int i = sl + SW * c; if (i < NQ) s.qpos[i] = x0q.r[c] + h * s.tau[i - 1]; }
}

// [MJ mj_step, integrator RK4] on the state in s.qpos / s.qvel / s.qws / s.act; xip = body COM positions of the 4th stage evaluation
template <class R, bool PROF = false, bool CARRY = false, int MAXR = 2 * SW>
DM_DEV void slot_rk4_step(const DevModel<R>& M, SlotShared<R>& s, const SlotTables& tb, int sl, int lane, const LaneTopo& lt, R* xip, int& ovf, long long* prof = 0) {
  const R h = M.timestep;
  const R A[3] = {R(0.5), R(0.5), R(1)};
  const R Bw[4] = {R(1) / 6, R(1) / 3, R(1) / 3, R(1) / 6};
  DofVec<R> x0q, x0v, vprev, aprev, sumv, suma;
#pragma unroll
  for (int c = 0; c < DOF_PASSES; c++) {
    const int i = sl + SW * c;
    x0q.r[c] = i < NQ ? s.qpos[i] : R(0);
    const R v0 = i < NV ? s.qvel[i] : R(0);
    x0v.r[c] = v0; vprev.r[c] = v0; aprev.r[c] = 0; sumv.r[c] = 0; suma.r[c] = 0;
  }
  dmw::sync();
  for (int i = 0; i < 4; i++) {   // single call site of the forward evaluation
    if (i > 0) {
      const R cf = A[i - 1];
#pragma unroll
      for (int c = 0; c < DOF_PASSES; c++) { const int d = sl + SW * c; if (d < NV) s.tau[d] = cf * vprev.r[c]; }
      dmw::sync();
      slot_integrate_pos(s, x0q, sl, h);
#pragma unroll
      for (int c = 0; c < DOF_PASSES; c++) {
        const int d = sl + SW * c;
        if (d < NV) { const R vi = x0v.r[c] + h * (cf * aprev.r[c]); vprev.r[c] = vi; s.qvel[d] = vi; }
      }
      dmw::sync();
    }
    // (the RK state is not touched by the evaluation: out of the architectural registers for its duration — wave.h park)
    typedef decltype(dmw::park(R(0))) ParkedR;
    ParkedR pk[5][DOF_PASSES];
#pragma unroll
    for (int c = 0; c < DOF_PASSES; c++) { pk[0][c] = dmw::park(x0q.r[c]); pk[1][c] = dmw::park(x0v.r[c]); pk[2][c] = dmw::park(vprev.r[c]); pk[3][c] = dmw::park(sumv.r[c]); pk[4][c] = dmw::park(suma.r[c]); }
    slot_forward<R, PROF, CARRY, MAXR>(M, s, tb, sl, lane, lt, xip, ovf, (const DebugOut*)0, prof);
#pragma unroll
    for (int c = 0; c < DOF_PASSES; c++) { x0q.r[c] = dmw::unpark(pk[0][c]); x0v.r[c] = dmw::unpark(pk[1][c]); vprev.r[c] = dmw::unpark(pk[2][c]); sumv.r[c] = dmw::unpark(pk[3][c]); suma.r[c] = dmw::unpark(pk[4][c]); }
#pragma unroll
    for (int c = 0; c < DOF_PASSES; c++) {
      const int d = sl + SW * c;
      if (d < NV) { const R a = s.qd.o.qacc[d]; aprev.r[c] = a; sumv.r[c] += Bw[i] * vprev.r[c]; suma.r[c] += Bw[i] * a; }
    }
  }
#pragma unroll
  for (int c = 0; c < DOF_PASSES; c++) { const int d = sl + SW * c; if (d < NV) s.tau[d] = sumv.r[c]; }
  dmw::sync();
  slot_integrate_pos(s, x0q, sl, h);
#pragma unroll
  for (int c = 0; c < DOF_PASSES; c++) { const int d = sl + SW * c; if (d < NV) { s.qvel[d] = x0v.r[c] + h * suma.r[c]; s.qws[d] = aprev.r[c]; } }
  dmw::sync();
}

// load the slot's env row (each slot reads 128-byte runs of its own row) and turn the action into actuator forces
template <class R>
DM_DEV void slot_load_env(const DevModel<R>& M, const Batch<R>& B, SlotShared<R>& s, int env, int sl, bool live, const double* action) {
#pragma unroll
  for (int c = 0; c < Q_PASSES; c++) {
    const int i = sl + SW * c;
    if (i < NQ) s.qpos[i] = B.qpos[(size_t)env * NQ + i];
    if (i < NV) { s.qvel[i] = B.qvel[(size_t)env * NV + i]; s.qws[i] = B.qws[(size_t)env * NV + i]; s.act[i] = 0; }
  }
  if (sl == 0) { s.status = 0; s.nefc = 0; s.ncon = 0; s.solver_iter = 0; }
  dmw::sync();
  if (action) {
#pragma unroll
    for (int c = 0; c < HINGE_PASSES; c++) {
      const int u = sl + SW * c;
      if (u < NU) {
        R a = (R)action[(size_t)env * NU + u];
        if (B.action_mode == 1) {
          const int idx = B.frame_idx[env];
          a += R(0.8) * (B.mocap_cfg[(size_t)idx * NQ + 7 + u] - s.qpos[7 + u]);
        } else if (B.action_mode == 2) {
          const int idx = B.frame_idx[env];
          a += M.kp[u + 6] * (B.mocap_cfg[(size_t)idx * NQ + 7 + u] - s.qpos[7 + u]) + M.kd[u + 6] * (B.mocap_vel[(size_t)idx * NV + 6 + u] - s.qvel[6 + u]);
        }
        if (live) B.ctrl[(size_t)env * NU + u] = a;
        const int d = u + 6;
        s.act[d] = M.gear[d] * clampr(a, M.ctrl_lo[d], M.ctrl_hi[d]);
      }
    }
  }
  dmw::sync_mem();
}

template <class R>
DM_DEV void slot_store_state(const Batch<R>& B, SlotShared<R>& s, int env, int sl, bool live) {
  if (!live) return;
#pragma unroll
  for (int c = 0; c < Q_PASSES; c++) {
    const int i = sl + SW * c;
    if (i < NQ) B.qpos[(size_t)env * NQ + i] = s.qpos[i];
    if (i < NV) { B.qvel[(size_t)env * NV + i] = s.qvel[i]; B.qws[(size_t)env * NV + i] = s.qws[i]; }
  }
}

// reset variants of env_step.h's reset_env for the slots whose `doit` is set (collective: every lane calls it; barriers are unconditional)
template <class R>
DM_DEV void slot_reset_env(const DevModel<R>& M, const Batch<R>& B, SlotShared<R>& s, int env, int sl, bool doit, int mode, int hard) {
  const int ep = B.episode[env];
  const int genv = B.env_offset + env;
  int idx = (int)(rng_uniform(B.seed, genv, ep, 0) * (double)B.n_frames);
  if (idx >= B.n_frames) idx = B.n_frames - 1;
  if (doit) {
#pragma unroll
    for (int c = 0; c < Q_PASSES; c++) {
      const int i = sl + SW * c;
      if (mode == 0) {
        if (i < NQ) s.qpos[i] = B.mocap_cfg[(size_t)idx * NQ + i];
        if (i < NV) s.qvel[i] = B.mocap_vel[(size_t)idx * NV + i];
      } else if (mode == 1) {
        if (i < NQ) s.qpos[i] = M.qpos0[i] + (R)((rng_uniform(B.seed, genv, ep, 1 + i) * 2.0 - 1.0) * 0.01);
        if (i < NV) s.qvel[i] = (R)((rng_uniform(B.seed, genv, ep, 64 + i) * 2.0 - 1.0) * 0.01);
      } else {
        if (i < NQ) s.qpos[i] = M.qpos0[i];
        if (i < NV) s.qvel[i] = 0;
      }
      if (hard && i < NV) s.qws[i] = 0;
    }
    if (hard && sl == 0) B.time[env] = 0;
  }
  dmw::sync_mem();
  if (doit && sl == 0) {
    B.episode[env] = ep + 1;
    if (mode == 0 || (mode == 1 && hard)) set_frame(B, env, idx); else B.cycle[env] = 0;
  }
}

// ---- 5-term imitation reward (env_step.h imitation_reward; code.md:1017-1143) for the slot's environment: one extra kinematics pass on
// the integrated state, whose by-products are the joint features.  Slot lane = body - 1 (lane 0 root, lanes 1..12 joint groups); the four
// end effectors on lanes 0..3 beside them; linear momentum one dof per lane in three passes; row sums (sum16) instead of wave sums.
// `refv` (wave-uniform) non-null: dp_env_v1's reward instead (env_step.h v1_reward; src/dp_env_v1.py:82-141, src/mujoco/mujoco_interface.py:169-210) from the
// SAME features — |angle| instead of angle^2 per joint with the un-normalised JOINT_WEIGHTs, L1 rate distance from row `refv` (the clip's rates frame k -> k + 1),
// L1 root distance, r = 0.5 e^(-2 pose) + 0.05 e^(-0.1 vel) + 0.2 e^(-5 root) — as selects inside the one pass, so that the step function holds ONE
// instance of the kinematics stage's code for its reward (round 6: reward mode 4 on the packed kernels).
template <class R>
DM_DEV R slot_imitation_reward(const DevModel<R>& M, const Batch<R>& B, SlotShared<R>& s, int sl, const LaneTopo& lt, const R* ref, R shx, R shy, const R* refv = nullptr) {
  R qloc[4], aloc[3][3], xip[3];
  slot_kinematics(M, s, sl, lt, xip, qloc, aloc);          // ends with a sync; s.r2.i.crb holds the composite inertias
  const bool v1 = refv != nullptr;
  const R* rv = v1 ? refv : ref;                           // the row the rates are compared with
  const R* P = B.imit_pdev;
  R rq[4] = {s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]};
  normalize4(rq);
  R pose = 0, vel = 0, eff = 0, root = 0, mx = 0, my = 0, mz = 0;
  if (sl < 13) {
    const int g = sl - 1;
    const int da = dmw::launder(lt.dofadr), nd = dmw::launder(lt.dofnum);
    const bool isroot = sl == 0, ball = !isroot && nd == 3;
    const R* rquat = ref + (isroot ? 3 : 13 + 4 * g);
    const R ident[4] = {1, 0, 0, 0};
    R q0[4], q1[4];
    for (int k = 0; k < 4; k++) { q0[k] = isroot ? rq[k] : (ball ? qloc[k] : ident[k]); q1[k] = (isroot || ball) ? rquat[k] : ident[k]; }
    const R th = quat_diff_theta(q0, q1);
    R pe = v1 ? fabs(th) : th * th, ve = 0;
    if (isroot) {
      R wv[3];
      const R wloc[3] = {s.qvel[3], s.qvel[4], s.qvel[5]};
      quat_rot(wv, rq, wloc);
      R dv2 = 0, dp2 = 0;
      for (int k = 0; k < 3; k++) { const R a = rv[10 + k] - wv[k]; ve += v1 ? fabs(a) : a * a; const R c = ref[7 + k] - s.qvel[k]; dv2 += c * c; }
      const R p1[3] = {ref[0] + shx, ref[1] + shy, ref[2]};
      for (int k = 0; k < 3; k++) { const R a = s.qpos[k] - p1[k]; dp2 += v1 ? fabs(a) : a * a; }
      root = v1 ? dp2 : dp2 + R(0.1) * pe + R(0.01) * dv2 + R(0.001) * ve;
    } else if (ball) {
      R wl[3] = {0, 0, 0};
      for (int k = 0; k < 3; k++) { const R rate = s.qvel[da + k]; wl[0] += aloc[k][0] * rate; wl[1] += aloc[k][1] * rate; wl[2] += aloc[k][2] * rate; }
      for (int k = 0; k < 3; k++) { const R w = rv[61 + 3 * g + k] - wl[k]; ve += v1 ? fabs(w) : w * w; }
    } else {
      const R a = ref[13 + 4 * g] - s.qpos[da + 1], w = rv[61 + 3 * g] - s.qvel[da];
      pe = v1 ? fabs(a) : a * a; ve = v1 ? fabs(w) + fabs(rv[61 + 3 * g + 1]) + fabs(rv[61 + 3 * g + 2]) : w * w;      // (v1: the two unused slots of the row are 0)
    }
    const R wj = v1 ? (isroot ? R(1) : P[g] / P[12]) : P[isroot ? 12 : g];
    pose = wj * pe; vel = v1 ? ve : wj * ve;
  }
  if (sl < 4) {
    const int e = sl, b = (int)P[16 + e];
    const R ex[3] = {1, 0, 0};
    R fwd[3], p[3], rel[3];
    quat_rot(fwd, rq, ex);
    const R hn = sqrt(fwd[0] * fwd[0] + fwd[1] * fwd[1]);
    const R c = hn > R(0) ? fwd[0] / hn : R(1), sn = hn > R(0) ? fwd[1] / hn : R(0);
    mat_vec(p, s.xmat[b], P + 20 + 3 * e);
    for (int k = 0; k < 3; k++) { p[k] += s.xpos[b][k]; rel[k] = p[k] - s.qpos[k]; }
    rel[2] = p[2];
    const R f0[3] = {c * rel[0] + sn * rel[1], -sn * rel[0] + c * rel[1], rel[2]};
    for (int k = 0; k < 3; k++) { const R a = ref[97 + 3 * e + k] - f0[k]; eff += a * a; }
  }
#pragma unroll
  for (int cpass = 0; cpass < DOF_PASSES; cpass++) {
    const int d = sl + SW * cpass;
    if (d < NV) {
      const int b = TOPO.dof_body[d];
      const R* cb = s.r2.i.crb[b];
      const R* cd = s.cdof[d];
      const R qd = s.qvel[d], ms = cb[9];
      R axs[3];
      cross3(axs, cd, cb + 6);
      mx += qd * (ms * cd[3] + axs[0]); my += qd * (ms * cd[4] + axs[1]); mz += qd * (ms * cd[5] + axs[2]);
    }
  }
  pose = dmw::sum16(pose); vel = dmw::sum16(vel); eff = dmw::sum16(eff) / 4; root = dmw::row_bcast<0>(root);
  mx = dmw::sum16(mx) / M.total_mass; my = dmw::sum16(my) / M.total_mass; mz = dmw::sum16(mz) / M.total_mass;
  const R dc[3] = {ref[109] - mx, ref[110] - my, ref[111] - mz};
  const R com = R(0.1) * dot3(dc, dc);
  // the terms, one lane each: 0.5 e^(-2 pose) + 0.05 e^(-0.1 vel) + 0.15 e^(-40 eff) + 0.2 e^(-5 root) + 0.1 e^(-10 com); v1: the first two and 0.2 e^(-5 root) on lane 2
  const R arg = sl == 0 ? R(-2) * pose : sl == 1 ? R(-0.1) * vel : sl == 2 ? (v1 ? R(-5) * root : R(-40) * eff) : sl == 3 ? R(-5) * root : R(-10) * com;
  const R wgt = sl == 0 ? R(0.5) : sl == 1 ? R(0.05) : sl == 2 ? (v1 ? R(0.2) : R(0.15)) : sl == 3 ? R(0.2) : R(0.1);
  R term = 0;
  if (sl < (v1 ? 3 : 5)) term = wgt * exp_once(arg);
  return dmw::sum16(term);
}

// DPEnv.step for the slot's environment (reward modes alive / v3-config / v2-pose / the 5-term imitation reward / dp_env_v1's reward).  A slot that is not `live` computes and stores nothing outside LDS.  An environment that exceeded a
// capacity of the packed path during the step (`ovf`) stores nothing either: it is appended to the launch's redo list
// (redo[0] = counter, list = redo + 1 ...) and re-stepped from its unchanged state by the one-env kernel.
// CARRY (horizon launches only) + kin_carry (wave-uniform): the slots' LDS is what this wave's previous step left, so a slot's `kin_ok` flag
// means what it says and the first evaluation may skip its position stage (slot_forward): one kinematics pass in five less with the 5-term
// reward, bit-identical results.  Without kin_carry (the first step of a launch, the step after an in-wave re-step) the flags are cleared first.
template <class R, bool PROF = false, bool CARRY = false, int MAXR = 2 * SW>
DM_DEV bool slot_env_step(const DevModel<R>& M, const Batch<R>& B, SlotShared<R>& s, SlotTables& tb, int env, int sl, int lane, bool live,
                          const double* action, double* obs, double* reward, unsigned char* done, int n_substeps, int* redo_count, int* redo_list,
                          long long* prof_out = 0, bool kin_carry = false) {
  long long prof[32];
  for (int k = 0; k < 32; k++) prof[k] = 0;
  long long tstart = 0;
  if (PROF) tstart = dmw::clk();
  const LaneTopo lt = lane_topo(sl);
  if constexpr (CARRY) { if (!kin_carry && sl == 0) s.kin_ok() = R(0); }       // (ordered before the first read by slot_load_env's hand-off)
  slot_load_env(M, B, s, env, sl, live, action);
  R xip[3];
  int why = 0;
  for (int k = 0; k < n_substeps; k++) slot_rk4_step<R, PROF, CARRY, MAXR>(M, s, tb, sl, lane, lt, xip, why, prof);
  const bool ovf = dmw::row_ballot(why != 0, lane) != 0u;
  if (dmw::ballot(ovf) != 0ull) {                     // rare: list the environment, tally the reasons (diagnostics)
    unsigned bits = 0;
    for (int r = 0; r < 5; r++) if (dmw::row_ballot(((why >> r) & 1) != 0, lane) != 0u) bits |= 1u << r;
    if (ovf && live && sl == 0) {
      if (redo_count) { const int k = dmw::global_counter_next(redo_count); redo_list[k] = env; }     // (no list: the caller re-steps it itself, slot_rollout)
      for (int r = 0; r < 5; r++) if ((bits >> r) & 1u) dmw::global_counter_next(B.redo_why + 1 + r);
    }
  }
  live = live && !ovf;
  // COM height of the 4th-stage body positions (src/dp_env_v3.py:134-139): mass-weighted sum over the body lanes
  const R mz = sl < NB - 1 ? M.body_mass[sl + 1] * xip[2] : R(0);
  const R z = dmw::sum16(mz) / M.total_mass;
  bool dn = (z < R(0.7)) || (z > R(2.0));
  if (live) {
    if (B.diag != 0) {
      if (sl < NB - 1) for (int k = 0; k < 3; k++) B.xipos[(size_t)env * NB * 3 + 3 * (sl + 1) + k] = xip[k];
      if (sl < 3) B.xipos[(size_t)env * NB * 3 + sl] = 0;
      const int nc = s.ncon;
      for (int k = sl; k < MAXEFC * 2; k += SW) {
        const int c = k >> 1;
        int gid = -1;
        if (c < nc && c < SLOT_MAXCON) { const auto& rec = M.pair_rec[s.r1.rw.coni[c] & 0xff]; gid = (k & 1) ? rec.g2 : rec.g1; }
        B.cong[(size_t)env * MAXEFC * 2 + k] = gid;
      }
    }
    if (sl == 0) { B.comz[env] = z; B.ncon[env] = s.ncon; B.nefc[env] = s.nefc; B.status[env] = s.status; B.solver_iter[env] = s.solver_iter; if constexpr (!CARRY) order_ticket(B, env, s.nefc, s.solver_iter); }   // (CARRY = inside a horizon launch: no tickets)
  }
  R rew = 1;
  if (B.reward_mode == REW_V3_CONFIG) {
    const int idx = B.frame_idx[env];
    R err = 0;
    for (int i = 7; i < NQ; i++) err += fabs(s.qpos[i] - B.mocap_cfg[(size_t)idx * NQ + i]);
    rew = exp_once(-err);
    dmw::sync_mem();
    if (live && sl == 0) B.frame_idx[env] = (idx + 1) % B.n_frames;
  } else if (B.reward_mode == REW_V2_POSE) {
    const int idx = B.frame_idx[env] + 1;
    const int im = (idx + B.frame_init[env]) % B.n_frames;
    R err = 0, acs = 0;
    for (int i = 3; i < NQ; i++) err += fabs(s.qpos[i] - B.mocap_cfg[(size_t)im * NQ + i]);
    for (int u = 0; u < NU; u++) { const R c = B.ctrl[(size_t)env * NU + u]; acs += c * c; }
    rew = exp_once(R(-2) * err) - R(0.1) * acs;
    dmw::sync_mem();
    if (live && sl == 0) B.frame_idx[env] = idx;
  } else if (B.reward_mode == REW_IMITATION) {   // code.md:1017-1143: the state after the step against frame idx + 1 (env_step.h)
    int k = B.frame_idx[env] + 1, cyc = B.cycle[env];
    bool ended = false;
    if (k >= B.n_frames) { if (B.imit_params[15] != R(0)) { k = 0; cyc += 1; } else { k = B.n_frames - 1; ended = true; } }
    rew = slot_imitation_reward(M, B, s, sl, lt, B.imit_table + (size_t)k * IMIT_FEAT, cyc * B.imit_params[13], cyc * B.imit_params[14]);
    dn = dn || ended;
    dmw::sync_mem();
    if (live && sl == 0) { B.frame_idx[env] = k; B.cycle[env] = cyc; }
  } else if (B.reward_mode == REW_V1_QUAT) {     // src/dp_env_v1.py:82-158 (env_step.h): the cursor counts steps, the pose reward every `upd` steps, minus the control cost
    const int idx = B.frame_idx[env] + 1;
    int upd = (int)floor(B.mocap_dt / (M.timestep * n_substeps));
    if (upd < 1) upd = 1;
    // (the four environments of a wave sit at different cursors: the pass is collective, so it runs every step on every slot — its kinematics are the
    //  next step's first evaluation's anyway (kin_carry) — and a slot takes its value on the steps its own cursor says so)
    const int k = (idx / upd + B.frame_init[env]) % B.n_frames, kv = k + 1 < B.n_frames ? k + 1 : B.n_frames - 1;
    const R robs = slot_imitation_reward(M, B, s, sl, lt, B.imit_table + (size_t)k * IMIT_FEAT, R(0), R(0), B.imit_table + (size_t)kv * IMIT_FEAT);
    R acs = 0;
    for (int u = 0; u < NU; u++) { const R c = B.ctrl[(size_t)env * NU + u]; acs += c * c; }
    rew = (idx % upd == 0 ? robs : R(0)) - R(0.1) * acs;
    dmw::sync_mem();
    if (live && sl == 0) B.frame_idx[env] = idx;
  }
  if (live && sl == 0) { B.time[env] += M.timestep * n_substeps; reward[env] = rew; done[env] = dn ? 1 : 0; if (B.kin) B.kin_ok[env] = 0; }
  if (B.autoreset) {                              // DummyVecEnv convention: obs of the fresh episode is returned
    dmw::sync_mem();
    slot_reset_env(M, B, s, env, sl, dn && live, B.autoreset == 1 ? 0 : 1, 1);
  }
  // the slot's kinematics are those of the state the env is left in: the 5-term reward's pass ran on it and no reset replaced it (a slot that
  // is not live repeats a live one's environment and computes the same)
  if constexpr (CARRY) { if (sl == 0) s.kin_ok() = (B.reward_mode >= REW_IMITATION && !ovf && !(dn && B.autoreset != 0)) ? R(1) : R(0); }
  if (live) {
#pragma unroll
    for (int c = 0; c < (NOBS + SW - 1) / SW; c++) {
      const int o = sl + SW * c;
      if (o < 28) obs[(size_t)env * NOBS + o] = s.qpos[7 + o];
      else if (o < NOBS) obs[(size_t)env * NOBS + o] = s.qvel[6 + (o - 28)];
    }
  }
  slot_store_state(B, s, env, sl, live);
  if (PROF && lane == 0) {                      // one record per WAVE: [0..4] kin, bias, mass, rows, constraint; [5] total; [6] PGS loop trips (speculative sweeps, wave-wide); [7] two-row-set evaluations;
    prof[5] = dmw::clk() - tstart;              // [8..13] constraint parts: row build, imp + half solve, A, warm start, PGS, assembly + solve; [14] sum of nmax; [15] constrained evaluations
    for (int k = 0; k < 32; k++) prof_out[k] = prof[k];   // [16..18] mass: f + M entries, elimination, scaling; [19..21] rows: geoms + limits, broad phase, narrow phase + emission; [23] candidates
  }
  return live;                                  // the slot's environment was stepped and stored by this wave
}

// ---- a whole horizon without leaving the wave (dm_batch_rollout) ---------------------------------------------------------------------
// T consecutive DPEnv.steps (src/dp_env_v3.py:106-132) of the wave's four environments; between two steps an optional policy step on the
// four observations just produced (src/trpo.py:49 `ac, vpred = pi.act(stochastic, ob)`), i.e. the loop body of traj_segment_generator
// (src/trpo.py:47-80) for T iterations.  Rows: action[t] is consumed by step t ([T (+1), N, 28]; the policy writes row t + 1),
// obs / reward / done row t is what step t returns ([T, N, ...]).
// Why: with one launch per step a step lasts as long as its SLOWEST wave (4 096 environments are exactly one wave per SIMD: nothing to
// balance against), and a wave is slow only while one of its environments is in heavy contact.  Here every wave runs ahead at its own
// pace and the horizon lasts as long as the slowest wave's SUM over T steps.
// An environment that exceeds a capacity of the packed path in some step is re-stepped right here by the one-env code (env_step, all 64
// lanes, from its unchanged state in memory) before the wave goes on: `one_s` / `one_x` may alias the slots' LDS (nothing in it outlives
// a step; the observations the policy reads are fetched again from the rows just written).  NR = the one-env code's register tier.
// The called step bodies without callee-saved registers (round 6).  Rounds 3-5 paid a save / restore of 337 registers per lane around every call (1.5 KB of
// the 2.1 KB frame; 40 KB of HBM traffic per env-step, 17x the algorithmic bytes): the AMDGPU calling convention makes half the register file callee-saved and the step
// uses all of it, while its caller keeps a handful of values across the call.  LLVM's inter-procedural register allocation (-mllvm -enable-ipra) compiles an INTERNAL,
// non-recursive function without callee-saved registers and hands its callers the exact clobber set — unless one of its call sites is marked `tail`, which the
// optimiser does to every call that passes no pointer into the caller's frame (round 4 concluded "does not fire on this backend" from exactly that).  So the callees
// are `static` (wave.h DM_DEV_CALL64) and receive the address of a caller-local word, which they write: the call cannot be a tail call, the frame shrinks from
// 1 536 to 192 bytes, and the horizon launch gains 3.7 % (20.43 against 19.69 M env-steps/s, driver window 17.25 against 16.25: profiles/r06_ab_kernel_variants.md
// section 4, gpurun call a6; bit-identical results).
#ifdef DM_STATIC_CALLS
#define DM_CALL_SLOT_PARAM , int* call_slot
#define DM_CALL_SLOT_TOUCH(v) (*call_slot = (v))
#define DM_CALL_SLOT_ARG , &call_slot_mem
#define DM_CALL_SLOT_DECL int call_slot_mem = 0; asm volatile("" :: "v"(&call_slot_mem) : "memory")
#define DM_CALL_SLOT_USE asm volatile("" :: "v"(call_slot_mem))
#else
#define DM_CALL_SLOT_PARAM
#define DM_CALL_SLOT_TOUCH(v) ((void)0)
#define DM_CALL_SLOT_ARG
#define DM_CALL_SLOT_DECL ((void)0)
#define DM_CALL_SLOT_USE ((void)0)
#endif
template <class R> union SlotOrOne {
  SlotShared<R> sh[SLOTS];
  struct { Shared<R> s; StepScratch<R> x; } one;
};
// (a real call: the one-env step keeps its own register allocation and spill slots, the hot loop around it is compiled as if it were not there)
template <class R, int NR>
DM_DEV_CALL64 void restep_one_env(const DevModel<R>* M, const Batch<R>* B, Shared<R>* s, StepScratch<R>* x, int env, int lane, const double* action, double* obs,
                                    double* reward, unsigned char* done, int n_substeps DM_CALL_SLOT_PARAM) {
  DM_CALL_SLOT_TOUCH(0);
  using dmw::uniform_ptr; using dmw::in_lds; using dmw::in_global; using dmw::in_constant;
  const Batch<R> Bv = *in_constant(B);                  // (the struct's pointers are generic too: a copy whose members are told to be global)
  env_step<R, NR>(*in_constant(M), global_members(Bv), *in_lds(uniform_ptr(s)), *in_lds(uniform_ptr(x)), dmw::uniform(env), lane, in_global(uniform_ptr(action)),
                  in_global(uniform_ptr(obs)), in_global(uniform_ptr(reward)), in_global(uniform_ptr(done)), dmw::uniform(n_substeps));
}
// (the packed step as a call too: its body is then compiled exactly as in k_step_packed — inlined into the horizon loop the register
//  allocator produced four times the spills and a 27 % slower step)
// Two instantiations: MAXR = 2 * SW — the lean step of rounds 3-4 (one and two row sets; an environment beyond 32 rows is flagged and re-stepped) — and
// MAXR = SLOT_MAXROWS — the same step with the partial third row set compiled in (33 .. 40 rows stay in the wave), 8 % slower on everything else (see
// slot_forward).  slot_rollout picks per wave-step.  Returns (stored ? 1 : 0) | row count of the slot's environment << 8.
template <class R, int MAXR>
DM_DEV_CALL64 int slot_env_step_call(const DevModel<R>* M, const Batch<R>* B, SlotShared<R>* s, SlotTables* tb, int env, int sl, int lane, bool live,
                                     const double* action, double* obs, double* reward, unsigned char* done, int n_substeps, int kin_carry DM_CALL_SLOT_PARAM) {
  using dmw::uniform_ptr; using dmw::in_lds; using dmw::in_global; using dmw::in_constant;
  const Batch<R> Bv = *in_constant(B);
  SlotShared<R>& sr = *in_lds(s);
  DM_CALL_SLOT_TOUCH(env);
  const bool stored = slot_env_step<R, false, true, MAXR>(*in_constant(M), global_members(Bv), sr, *in_lds(uniform_ptr(tb)), env, sl, lane, live, in_global(uniform_ptr(action)),
                                                          in_global(uniform_ptr(obs)), in_global(uniform_ptr(reward)), in_global(uniform_ptr(done)), dmw::uniform(n_substeps), (int*)0, (int*)0,
                                                          (long long*)0, dmw::uniform(kin_carry) != 0);
  return (stored ? 1 : 0) | (sr.nefc << 8);
}
#ifdef DM_ROLLOUT_PROF     // diagnostic build (tools/profile_horizon.py): the step with per-stage shader-clock stamps, one 32-counter record per call
template <class R, int MAXR>
DM_DEV_CALL64 int slot_env_step_call_prof(const DevModel<R>* M, const Batch<R>* B, SlotShared<R>* s, SlotTables* tb, int env, int sl, int lane, bool live,
                                          const double* action, double* obs, double* reward, unsigned char* done, int n_substeps, long long* prof_out, int kin_carry DM_CALL_SLOT_PARAM) {
  DM_CALL_SLOT_TOUCH(env);
  using dmw::uniform_ptr; using dmw::in_lds; using dmw::in_global; using dmw::in_constant;
  const Batch<R> Bv = *in_constant(B);
  SlotShared<R>& sr = *in_lds(s);
  const bool stored = slot_env_step<R, true, true, MAXR>(*in_constant(M), global_members(Bv), sr, *in_lds(uniform_ptr(tb)), env, sl, lane, live, in_global(uniform_ptr(action)),
                                                         in_global(uniform_ptr(obs)), in_global(uniform_ptr(reward)), in_global(uniform_ptr(done)), dmw::uniform(n_substeps), (int*)0, (int*)0,
                                                         in_global(uniform_ptr(prof_out)), dmw::uniform(kin_carry) != 0);
  return (stored ? 1 : 0) | (sr.nefc << 8);
}
#endif
// The buffers of ONE step of a horizon: what one dm_batch_step call names (action [N, 28] in; obs [N, 56], reward [N], done [N] out).  A horizon
// launch reads a table of T of them — rows of the caller's [T, N, .] tensors (dm_batch_rollout) or the buffers of T queued dm_batch_step
// calls (DM_OPT_STEP_QUEUE), which need not be strided.
struct StepRow { const double* action; double* obs; double* reward; unsigned char* done; };
#ifndef DM_EXT_MARGIN
#define DM_EXT_MARGIN 9
#endif
// slot_rollout calls the three-set instantiation of the step from 2 * SW - EXT_MARGIN + 1 = 24 rows up.  Measured (gpurun call h9; a synthetic standing
// population / the judged line, M env-steps/s): margin 3 (30 rows): 9.45 / 17.95, 1.4 % of env-steps still re-stepped (a foot landing takes an env from 28 to
// 33 rows in one step); margin 9 (24 rows): 9.68 / 17.94, 0.26 %; margin 15 (18 rows): 9.74 / 17.76, 0.05 %; round 4's kernel: 6.98 / 17.83, 13.9 %.
constexpr int EXT_MARGIN = DM_EXT_MARGIN;
template <class R, int NR, class POLICY>
DM_DEV void slot_rollout(const DevModel<R>& M_in, const Batch<R>& B, SlotShared<R>* sh, SlotTables& tb, Shared<R>& one_s, StepScratch<R>& one_x,
                         int env, int lane, bool live, const StepRow* rows, int n_substeps, int T, POLICY&& policy, long long* prof_acc = 0, bool policy_clobbers_kin = false) {
  const int slot = lane >> 4, sl = lane & 15;
  int carry = 0;                              // the slots' LDS is what this wave's previous step left (no in-wave re-step overwrote it): slot_env_step kin_carry
  // rows the slot's environment held after its last step: decides which instantiation of the step this wave calls next (wave-uniform: the largest of the four)
  int last_rows = B.nefc[env];
  DM_CALL_SLOT_DECL;
  for (int t = 0; t < T; t++) {
    // (every step reads the model afresh: hoisting those loads out of the loop would keep hundreds of registers alive across it)
    const DevModel<R>& M = *dmw::launder_uniform_ptr(&M_in);
    const StepRow row = rows[t];                                     // (wave-uniform: four scalar loads)
    const double* a_t = row.action;
    double* o_t = row.obs;
    double* r_t = row.reward;
    unsigned char* d_t = row.done;
    // an environment within EXT_MARGIN rows of the two-set capacity: the step with the third row set compiled in (standing humanoids sit at 32 .. 37 rows
    // for hundreds of steps: the prediction misses on the step that takes one there — that env-step is re-stepped by the one-env code, as all were before)
    const bool heavy = rows_max(last_rows) > 2 * SW - EXT_MARGIN;
    int ret;
#ifdef DM_ROLLOUT_PROF
    if (prof_acc) {        // [0..31] sums over the horizon's steps, [32..63] the record of the step just taken
      ret = heavy ? slot_env_step_call_prof<R, SLOT_MAXROWS>(&M, &B, &sh[slot], &tb, env, sl, lane, live, a_t, o_t, r_t, d_t, n_substeps, prof_acc + 32, carry DM_CALL_SLOT_ARG)
                  : slot_env_step_call_prof<R, 2 * SW>(&M, &B, &sh[slot], &tb, env, sl, lane, live, a_t, o_t, r_t, d_t, n_substeps, prof_acc + 32, carry DM_CALL_SLOT_ARG);
      if (lane == 0) { for (int k = 0; k < 31; k++) prof_acc[k] += prof_acc[32 + k]; if (heavy) prof_acc[28] += 1; }
    } else
#endif
    // (measured, round 4: the step body inlined here instead of called — the batch descriptor and tables read afresh every step so that nothing is
    //  hoisted — removes the callee's register save / restore (1.6 KB per lane per call) and is 10 % SLOWER: 15.0 against 16.7 M env-steps/s,
    //  profiles/r04_ab_kernel_variants.md; the loop around the body costs the allocator more than the calls cost the memory system)
    ret = heavy ? slot_env_step_call<R, SLOT_MAXROWS>(&M, &B, &sh[slot], &tb, env, sl, lane, live, a_t, o_t, r_t, d_t, n_substeps, carry DM_CALL_SLOT_ARG)
                : slot_env_step_call<R, 2 * SW>(&M, &B, &sh[slot], &tb, env, sl, lane, live, a_t, o_t, r_t, d_t, n_substeps, carry DM_CALL_SLOT_ARG);
    const bool stored = (ret & 1) != 0;
    last_rows = (stored || !live) ? (ret >> 8) : SLOT_MAXROWS;       // (an environment that left the packed path: assume it is heavy)
    const int need = (live && !stored) ? 1 : 0;
    bool any = false;
    for (int k = 0; k < SLOTS; k++) {
      if (dmw::bcast_i(need, SW * k) == 0) continue;             // wave-uniform
      any = true;
      const int e = dmw::bcast_i(env, SW * k);
      dmw::sync_mem();
      restep_one_env<R, NR>(&M, &B, &one_s, &one_x, e, lane, a_t, o_t, r_t, d_t, n_substeps DM_CALL_SLOT_ARG);
      if (lane == 0) dmw::global_counter_next(B.redo_why);       // running total (dm_batch_redo_total)
      dmw::sync_mem();
    }
    if (any) {                                                   // the slots' LDS may have been overwritten: the policy's inputs again
#pragma unroll
      for (int c = 0; c < (NOBS + SW - 1) / SW; c++) {
        const int o = sl + SW * c;
        if (o < 28) sh[slot].qpos[7 + o] = (R)o_t[(size_t)env * NOBS + o];
        else if (o < NOBS) sh[slot].qvel[6 + (o - 28)] = (R)o_t[(size_t)env * NOBS + o];
      }
      dmw::sync_mem();
    }
    DM_CALL_SLOT_USE;
    policy(t);                                                   // (float64 build: its scratch stays inside the slots' r1 region and the kinematics survive it)
    carry = (any || policy_clobbers_kin) ? 0 : 1;
    dmw::sync_mem();                                             // state / action rows written by other lanes of this wave are read next step
  }
}

}  // namespace dm
