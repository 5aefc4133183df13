// slot_kernel.h — FOUR environments per wavefront: every 16-lane DPP row of the wave is one environment's "slot".
//
// Why: one environment offers 13 bodies, 34 dofs and a handful of constraint rows — most stages of env_kernel.h (one environment
// per 64-lane wave) issue their instructions for 64 lanes and use 6..13 of them; the instruction count per environment, not any
// unit's throughput, bounds that kernel.  Here the same instruction advances four environments: lane
// l = 16 * slot + sl works on item sl (body, geom, constraint row, ...) of environment `slot`; items that outnumber 16 lanes (28
// hinges, 34 dofs, 310 matrix entries) take several passes.  Cross-lane traffic stays inside a row: sums are the DPP reductions of
// wave.h (sum16), broadcasts are `row_bcast<I>` / `row_fmac<I>` (v_mov_b64_dpp / v_fmac_f64_dpp row_newbcast) — one instruction for
// all four environments.  Each slot owns a `SlotShared` block of LDS (<= 10 KB, so that four waves — one per SIMD — fit a CU);
// LDS addresses are per-lane (base of the own slot + constant), wave-level fences order the hand-offs exactly as in the one-env
// kernel.
// The arithmetic per environment is the one of env_kernel.h, stage by stage, in the same order wherever the lane mapping allows it;
// results do not depend on which other environments share the wave (loops run to the wave's maximum with finished slots frozen).
// Stage order differs in one place: bias forces are formed BEFORE the mass matrix (they are independent), so that the factor can
// take over the LDS region of the inertias.
#pragma once

#include "env_kernel.h"

namespace dm {

constexpr int SLOTS = 4, SW = 16;            // environments per wavefront, lanes per environment
constexpr int DOF_PASSES = (NV + SW - 1) / SW, HINGE_PASSES = (NU + SW - 1) / SW, Q_PASSES = (NQ + SW - 1) / SW, ENT_PASSES = (310 + SW - 1) / SW;

// per-slot LDS working set.  r1 / r2 are reused along one forward evaluation:
//   r1: kinematics scratch (body quaternions, hinge half-angle sines, frame offsets) | RNE scratch | M-build scratch (fdof)
//   r2: spatial / composite inertias (kinematics .. M build)                        | the L^T D L factor (M build .. end of the evaluation)
template <class R>
struct SlotShared {
  R qpos[36], qvel[NV];
  R xpos[NB][3], xmat[NB][9];
  R tau[NV], qacc[NV];
  R cdof[NV][6];
  R dinv[NV], dsq[NV], act[NV], qws[NV];
  union {
    struct { R xquat[NB][4], sc[NU][2], off[NB][3]; } k;
    struct { R cvel[NB][6], cacc[NB][6], cfrc[NB][6]; } v;      // (the subtree sums of cfrc overwrite cvel)
    R fdof[NV][6];
  } r1;
  union {
    struct { R sin[NB][10], crb[NB][10]; } i;
    R qLD[312];
  } r2;
  int nefc, ncon, status, solver_iter;
};
// index tables shared by the four slots of a workgroup (compile-time topology; see LaneTables)
struct SlotTables {
  unsigned short tab_dst[NV][14];
  unsigned short tab_ent[312];
};
DM_DEV void stage_slot_tables(SlotTables& t, int lane) {
#pragma unroll
  for (int c = 0; c < (312 + 63) / 64; c++) { const int e = lane + 64 * c; if (e < 312) t.tab_ent[e] = LTAB.tab_ent[e]; }
#pragma unroll
  for (int c = 0; c < (NV * 14 + 63) / 64; c++) { const int i = lane + 64 * c; if (i < NV * 14) (&t.tab_dst[0][0])[i] = LTAB.tab_dst[i]; }
}

// ---- subtree sums: out[b][k] = sum over the bodies c of b's subtree of in[c][k], component k = slot lane, one body per pass,
// members added in ascending body order (the order of env_kernel.h's SubtreeAcc) ---------------------------------------------------
template <int NC, int B, int C, class R>
struct SlotSubtreeAcc {
  static DM_DEV void run(R& acc, const R (*in)[NC], int k) {
    if constexpr (C < NB) {
      if constexpr ((TOPO.subtree[B] >> C) & 1u) acc += in[C][k];
      SlotSubtreeAcc<NC, B, C + 1, R>::run(acc, in, k);
    }
  }
};
template <int NC, int B, class R>
DM_DEV void slot_subtree_sums(const R (*in)[NC], R (*out)[NC], int sl) {
  if constexpr (B < NB) {
    if (sl < NC) {
      R acc = 0;
      SlotSubtreeAcc<NC, B, 1, R>::run(acc, in, sl);
      out[B][sl] = acc;
    }
    slot_subtree_sums<NC, B + 1, R>(in, out, sl);
  }
}

// ---- position stage (env_kernel.h stage_kinematics, lane -> slot lane).  xip: this body lane's inertial-frame position (the body
// COM of `sim.data.xipos`), kept in registers instead of LDS. ----------------------------------------------------------------------
template <class R>
DM_DEV void slot_kinematics(const DevModel<R>& M, SlotShared<R>& s, int sl_in, const LaneTopo& lt, R* xip, R* qloc_out = nullptr, R (*aloc_out)[3] = nullptr) {
  const int sl = dmw::launder(sl_in);
  const int b = sl + 1;
  const bool isbody = sl < NB - 1;
  const int depth = dmw::launder(lt.depth), da = dmw::launder(lt.dofadr), nd = dmw::launder(lt.dofnum), panc = dmw::launder(lt.parent), p = panc & 15;
  R qloc[4] = {1, 0, 0, 0}, aloc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  if (sl == 0) {
    s.xpos[0][0] = s.xpos[0][1] = s.xpos[0][2] = 0; s.r1.k.off[0][0] = s.r1.k.off[0][1] = s.r1.k.off[0][2] = 0;
    s.r1.k.xquat[0][0] = 1; s.r1.k.xquat[0][1] = s.r1.k.xquat[0][2] = s.r1.k.xquat[0][3] = 0;
    for (int k = 0; k < 9; k++) s.xmat[0][k] = (k % 4 == 0) ? R(1) : R(0);
  }
  // 1a. half-angle sine / cosine of every hinge
#pragma unroll
  for (int c = 0; c < HINGE_PASSES; c++) {
    const int h = sl + SW * c;
    if (h < NU) {
      const R half = (s.qpos[h + 7] - M.qpos0[h + 7]) * R(0.5);
      const SinCos<R> sc = sincos_once(half);
      s.r1.k.sc[h][0] = sc.c; s.r1.k.sc[h][1] = sc.s;
    }
  }
  dmw::sync();
  // 1b. local hinge chain (bodies 2..13)
  if (isbody && b > 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) if (k < nd) {
      const int d = da + k, j = d - 5;
      const R axl[3] = {M.jnt_axis[j][0], M.jnt_axis[j][1], M.jnt_axis[j][2]};
      R qm[9];
      quat2mat(qm, qloc);
      mat_vec(aloc[k], qm, axl);
      const R c = s.r1.k.sc[d - 6][0], sn = s.r1.k.sc[d - 6][1];
      const R ql[4] = {c, axl[0] * sn, axl[1] * sn, axl[2] * sn};
      quat_mul(qloc, qloc, ql);
    }
  }
  if (qloc_out) {
    for (int k = 0; k < 4; k++) qloc_out[k] = qloc[k];
    for (int k = 0; k < 3; k++) for (int r = 0; r < 3; r++) aloc_out[k][r] = aloc[k][r];
  }
  // 2. compose down the tree (unnormalised quaternion products ride the serial chain)
  R q[4] = {1, 0, 0, 0};
#pragma unroll
  for (int L = 1; L <= MAXDEPTH_BODY; L++) {
    if (isbody && depth == L) {
      if (b == 1) { q[0] = s.qpos[3]; q[1] = s.qpos[4]; q[2] = s.qpos[5]; q[3] = s.qpos[6]; }
      else quat_mul(q, s.r1.k.xquat[p], qloc);
      for (int k = 0; k < 4; k++) s.r1.k.xquat[b][k] = q[k];
    }
    dmw::sync();
  }
  if (isbody) {
    R mat[9];
    normalize4(q);
    quat2mat(mat, q);
    for (int k = 0; k < 9; k++) s.xmat[b][k] = mat[k];
  }
  dmw::sync();
  if (isbody) {
    R v[3];
    if (b == 1) { v[0] = s.qpos[0]; v[1] = s.qpos[1]; v[2] = s.qpos[2]; }
    else mat_vec(v, s.xmat[p], M.body_pos[b]);
    for (int k = 0; k < 3; k++) s.r1.k.off[b][k] = v[k];
  }
  dmw::sync();
  R xp[3] = {0, 0, 0};
  if (isbody) {
    const int p2 = (panc >> 4) & 15, p3 = (panc >> 8) & 15;
    for (int k = 0; k < 3; k++) xp[k] = ((s.r1.k.off[p3][k] + s.r1.k.off[p2][k]) + s.r1.k.off[p][k]) + s.r1.k.off[b][k];
  }
  if (isbody) for (int k = 0; k < 3; k++) s.xpos[b][k] = xp[k];
  // 3. motion axes, inertial frame position, own spatial inertia about the origin
  xip[0] = xip[1] = xip[2] = 0;
  if (isbody) {
    const R* mat = s.xmat[b];
    if (b == 1) {
      for (int k = 0; k < 3; k++) {
        for (int r = 0; r < 6; r++) s.cdof[k][r] = (r == 3 + k) ? R(1) : R(0);
        const R ax[3] = {mat[k], mat[3 + k], mat[6 + k]};
        s.cdof[3 + k][0] = ax[0]; s.cdof[3 + k][1] = ax[1]; s.cdof[3 + k][2] = ax[2];
        cross3(&s.cdof[3 + k][3], xp, ax);
      }
    } else {
#pragma unroll
      for (int k = 0; k < 3; k++) if (k < nd) {
        R axw[3];
        mat_vec(axw, s.xmat[p], aloc[k]);
        s.cdof[da + k][0] = axw[0]; s.cdof[da + k][1] = axw[1]; s.cdof[da + k][2] = axw[2];
        cross3(&s.cdof[da + k][3], xp, axw);
      }
    }
    const R ip[3] = {M.body_ipos[b][0], M.body_ipos[b][1], M.body_ipos[b][2]};
    R c[3];
    mat_vec(c, mat, ip);
    c[0] += xp[0]; c[1] += xp[1]; c[2] += xp[2];
    xip[0] = c[0]; xip[1] = c[1]; xip[2] = c[2];
    const R* Ib = M.body_inertia[b];
    const R A[9] = {Ib[0], Ib[3], Ib[4], Ib[3], Ib[1], Ib[5], Ib[4], Ib[5], Ib[2]};
    R T[9], Iw[9];
    for (int i = 0; i < 3; i++) for (int jx = 0; jx < 3; jx++) T[3 * i + jx] = mat[3 * i] * A[jx] + mat[3 * i + 1] * A[3 + jx] + mat[3 * i + 2] * A[6 + jx];
    for (int i = 0; i < 3; i++) for (int jx = 0; jx < 3; jx++) Iw[3 * i + jx] = T[3 * i] * mat[3 * jx] + T[3 * i + 1] * mat[3 * jx + 1] + T[3 * i + 2] * mat[3 * jx + 2];
    const R m = M.body_mass[b];
    const R cc = dot3(c, c);
    R* S = s.r2.i.sin[b];
    S[0] = Iw[0] + m * (cc - c[0] * c[0]); S[1] = Iw[4] + m * (cc - c[1] * c[1]); S[2] = Iw[8] + m * (cc - c[2] * c[2]);
    S[3] = Iw[1] - m * c[0] * c[1]; S[4] = Iw[2] - m * c[0] * c[2]; S[5] = Iw[5] - m * c[1] * c[2];
    S[6] = m * c[0]; S[7] = m * c[1]; S[8] = m * c[2]; S[9] = m;
  }
  dmw::sync();
  // 4. composite inertias   [MJ mj_crb backward pass]
  slot_subtree_sums<10, 1, R>(s.r2.i.sin, s.r2.i.crb, sl);
  dmw::sync();
}

// ---- velocity stage (env_kernel.h stage_bias): bias forces incl. gravity, smooth generalized force --------------------------------
template <class R>
DM_DEV void slot_bias(const DevModel<R>& M, SlotShared<R>& s, int sl_in, const LaneTopo& lt) {
  const int sl = dmw::launder(sl_in);
  const int b = sl + 1;
  const bool isbody = sl < NB - 1;
  const int depth = dmw::launder(lt.depth), p = dmw::launder(lt.parent) & 15, da = dmw::launder(lt.dofadr), nd = dmw::launder(lt.dofnum);
  if (sl == 0) {
    for (int r = 0; r < 6; r++) { s.r1.v.cvel[0][r] = 0; s.r1.v.cacc[0][r] = 0; }
    s.r1.v.cacc[0][3] = -M.gravity[0]; s.r1.v.cacc[0][4] = -M.gravity[1]; s.r1.v.cacc[0][5] = -M.gravity[2];
  }
  R S[6] = {0, 0, 0, 0, 0, 0}, C[6] = {0, 0, 0, 0, 0, 0};
  if (isbody && b > 1) {
#pragma unroll
    for (int k = 0; k < 3; k++) if (k < nd) {
      const R qd = s.qvel[da + k];
      R T[6], cd[6];
      for (int r = 0; r < 6; r++) T[r] = s.cdof[da + k][r] * qd;
      if (k > 0) { cross_motion(cd, S, T); for (int r = 0; r < 6; r++) C[r] += cd[r]; }
      for (int r = 0; r < 6; r++) S[r] += T[r];
    }
  }
  dmw::sync();
#pragma unroll
  for (int L = 1; L <= MAXDEPTH_BODY; L++) {
    if (isbody && depth == L) {
      R v[6], a[6];
      for (int r = 0; r < 6; r++) { v[r] = s.r1.v.cvel[p][r]; a[r] = s.r1.v.cacc[p][r]; }
      if (b == 1) {
        for (int k = 0; k < 3; k++) { const R qd = s.qvel[k]; for (int r = 0; r < 6; r++) v[r] += s.cdof[k][r] * qd; }
        R vb[6];
        for (int r = 0; r < 6; r++) vb[r] = v[r];
        for (int k = 3; k < 6; k++) {
          R cd[6]; cross_motion(cd, vb, s.cdof[k]);
          const R qd = s.qvel[k];
          for (int r = 0; r < 6; r++) { a[r] += cd[r] * qd; v[r] += s.cdof[k][r] * qd; }
        }
      } else {
        R cd[6];
        cross_motion(cd, v, S);
        for (int r = 0; r < 6; r++) { a[r] += cd[r] + C[r]; v[r] += S[r]; }
      }
      for (int r = 0; r < 6; r++) { s.r1.v.cvel[b][r] = v[r]; s.r1.v.cacc[b][r] = a[r]; }
    }
    dmw::sync();
  }
  if (isbody) {
    R v[6], a[6];
    for (int r = 0; r < 6; r++) { v[r] = s.r1.v.cvel[b][r]; a[r] = s.r1.v.cacc[b][r]; }
    R Ia[6], Iv[6], x[6];
    sinert_mul(Ia, s.r2.i.sin[b], a); sinert_mul(Iv, s.r2.i.sin[b], v); cross_force(x, v, Iv);
    for (int r = 0; r < 6; r++) s.r1.v.cfrc[b][r] = Ia[r] + x[r];
  }
  dmw::sync();
  slot_subtree_sums<6, 1, R>(s.r1.v.cfrc, s.r1.v.cvel, sl);       // csub -> the velocity region (dead by now)
  dmw::sync();
#pragma unroll
  for (int c = 0; c < DOF_PASSES; c++) {
    const int d = sl + SW * c;
    if (d < NV) {
      const R bias = dot6(s.cdof[d], s.r1.v.cvel[TOPO.dof_body[d]]);
      s.tau[d] = -M.dof_damping[d] * s.qvel[d] - bias + s.act[d];
    }
  }
  dmw::sync();
}

// ---- mass matrix and its L^T D L factor (env_kernel.h stage_mass_matrix).  The elimination steps are those of ELIM_STEPS; the
// columns of a step are taken one after the other by the slot's 16 lanes, in the order (pass, column, pair) in which the one-env
// kernel's lane groups apply them, so that entries which several limbs update receive their contributions in the same order. ---------
template <int S, int P, int CI, class R>
DM_DEV void slot_eliminate_chunk(SlotShared<R>& s, const SlotTables& tb, int sl, const LaneTopo& lt) {
  // chunk = 16 consecutive pair numbers [16 * q, 16 * q + 16) of column K = ELIM_STEPS[S].K[CI]; P is the one-env kernel's pass number
  constexpr ElimStep st = ELIM_STEPS[S];
  constexpr int K = st.K[CI], gs = elim_group_size(st.ncol, CI), np = elim_npairs(K), base = TOPO.madr[K];
  constexpr int sub = gs / SW;                                  // 16-lane chunks per pass of the one-env lane group
  const R inv = dmw::rcp_fast(s.r2.qLD[base]);
#pragma unroll
  for (int u = 0; u < sub; u++) {
    constexpr int dummy = 0; (void)dummy;
    const int t0 = gs * P + SW * u;
    if (t0 < np) {
      const int t = t0 + sl;
      const bool on = t < np;
      const int code = on ? (int)(lt.tri >> (8 * (t >> 4))) & 0xff : 0, e = code >> 4, a = code & 15;
      const int dst = tb.tab_dst[K][a] + (e - a);
      dmw::lds_sub(on, &s.r2.qLD[dst], s.r2.qLD[base + e] * (s.r2.qLD[base + a] * inv));
    }
  }
}
template <int S, int P, int CI, class R>
struct SlotElimCols {
  static DM_DEV void run(SlotShared<R>& s, const SlotTables& tb, int sl, const LaneTopo& lt) {
    if constexpr (CI < ELIM_STEPS[S].ncol) {
      constexpr int gs = elim_group_size(ELIM_STEPS[S].ncol, CI), np = elim_npairs(ELIM_STEPS[S].K[CI]);
      if constexpr (gs * P < np) slot_eliminate_chunk<S, P, CI, R>(s, tb, sl, lt);
      SlotElimCols<S, P, CI + 1, R>::run(s, tb, sl, lt);
    }
  }
};
template <int S, int P, class R>
struct SlotElimPasses {
  static DM_DEV void run(SlotShared<R>& s, const SlotTables& tb, int sl, const LaneTopo& lt) {
    if constexpr (P < elim_passes(S)) {
      SlotElimCols<S, P, 0, R>::run(s, tb, sl, lt);
      SlotElimPasses<S, P + 1, R>::run(s, tb, sl, lt);
    }
  }
};
template <int S, class R>
struct SlotEliminateFrom {
  static DM_DEV void run(SlotShared<R>& s, const SlotTables& tb, int sl, const LaneTopo& lt) {
    if constexpr (S < N_ELIM_STEPS) {
      SlotElimPasses<S, 0, R>::run(s, tb, dmw::launder(sl), lt);
      dmw::sync();
      SlotEliminateFrom<S + 1, R>::run(s, tb, sl, lt);
    }
  }
};
static_assert(elim_group_size(4, 0) % SW == 0 && elim_group_size(3, 2) % SW == 0 && elim_group_size(1, 0) % SW == 0, "lane groups are whole slots");
// (pair codes of a lane cover t = (lane & 15) + 16 j, j < 6, i.e. t < 96: elim_codes_ok)

template <class R>
DM_DEV void slot_mass_matrix(const DevModel<R>& M, SlotShared<R>& s, const SlotTables& tb, int sl_in, const LaneTopo& lt, const DebugOut* dbg) {
  const int sl = dmw::launder(sl_in);
  // f_d = (composite inertia of the dof's body) cdof_d: the inertias are read before the region they share with the factor is written
  R f[DOF_PASSES][6];
#pragma unroll
  for (int c = 0; c < DOF_PASSES; c++) {
    const int d = sl + SW * c;
    if (d < NV) sinert_mul(f[c], s.r2.i.crb[TOPO.dof_body[d]], s.cdof[d]);
  }
#pragma unroll
  for (int c = 0; c < DOF_PASSES; c++) {
    const int d = sl + SW * c;
    if (d < NV) {
      for (int r = 0; r < 6; r++) s.r1.fdof[d][r] = f[c][r];
      s.dinv[d] = M.dof_armature[d];
    }
  }
  dmw::sync();
#pragma unroll
  for (int c = 0; c < ENT_PASSES; c++) {
    const int e = sl + SW * c;
    if ((c + 1) * SW <= TOPO.nM || e < TOPO.nM) {
      const int ij = tb.tab_ent[e], i = ij >> 8, j = ij & 0xff;
      R v = dot6(s.cdof[j], s.r1.fdof[i]);
      if (i == j) v += s.dinv[i];
      s.r2.qLD[e] = v;
      if (dbg) { dbg->out[i * NV + j] = (double)v; dbg->out[j * NV + i] = (double)v; }
    }
  }
  dmw::sync();
  SlotEliminateFrom<0, R>::run(s, tb, sl, lt);
#pragma unroll
  for (int c = 0; c < DOF_PASSES; c++) {
    const int d = sl + SW * c;
    if (d < NV) { const R inv = R(1) / s.r2.qLD[TOPO.madr[d]]; s.dinv[d] = inv; s.dsq[d] = sqrt(inv); }
  }
  dmw::sync();
#pragma unroll
  for (int c = 0; c < ENT_PASSES; c++) {
    const int e = sl + SW * c;
    if ((c + 1) * SW <= TOPO.nM || e < TOPO.nM) {
      const int ij = tb.tab_ent[e], i = ij >> 8, j = ij & 0xff;
      const R sc = i != j ? s.dinv[i] : R(1);
      s.r2.qLD[e] *= sc;
    }
  }
  dmw::sync();
}

// ---- unconstrained solve: qacc = qacc_smooth = L^-1 D^-1 L^-T tau.  Every lane of the slot carries the whole vector (factor entries
// are slot-uniform LDS reads); lane 0 publishes the result. --------------------------------------------------------------------------
template <class R>
DM_DEV void slot_smooth_solve(SlotShared<R>& s, int sl, const DebugOut* dbg) {
  R x[NV];
#pragma unroll
  for (int d = 0; d < NV; d++) x[d] = s.tau[d];
  solve_LT(x, s.r2.qLD);
#pragma unroll
  for (int d = 0; d < NV; d++) x[d] *= s.dinv[d];
  solve_L(x, s.r2.qLD);
  if (sl == 0) {
#pragma unroll
    for (int d = 0; d < NV; d++) { s.qacc[d] = x[d]; if (dbg) dbg->out[34 * 34 + 34 + d] = (double)x[d]; }
  }
  dmw::sync();
}

// one forward-dynamics evaluation of the slot's environment: s.qpos, s.qvel, s.act, s.qws -> s.qacc; xip = body COM positions (body lanes)
template <class R>
DM_DEV void slot_forward(const DevModel<R>& M, SlotShared<R>& s, const SlotTables& tb, int sl, const LaneTopo& lt, R* xip, const DebugOut* dbg) {
  DM_MARK("slot_kinematics");
  slot_kinematics(M, s, sl, lt, xip);
  if (dbg) { for (int e = sl; e < NV * NV; e += SW) dbg->out[e] = 0; dmw::sync(); }
  DM_MARK("slot_bias");
  slot_bias(M, s, sl, lt);
  if (dbg) {
    for (int c = 0; c < DOF_PASSES; c++) { const int d = sl + SW * c; if (d < NV) dbg->out[34 * 34 + d] = (double)(-M.dof_damping[d] * s.qvel[d] + s.act[d] - s.tau[d]); }
  }
  DM_MARK("slot_mass_factor");
  slot_mass_matrix(M, s, tb, sl, lt, dbg);
  DM_MARK("slot_constraint");
  if (sl == 0) { s.nefc = 0; s.ncon = 0; s.solver_iter = 0; }     // (models with contacts / limits: rows + constraint stages, below)
  slot_smooth_solve(s, sl, dbg);
  DM_MARK("slot_forward_end");
  if (dbg) {
    for (int c = 0; c < DOF_PASSES; c++) { const int d = sl + SW * c; if (d < NV) dbg->out[34 * 34 + 68 + d] = (double)s.qacc[d]; }
    if (sl < NB - 1) for (int k = 0; k < 3; k++) dbg->out[34 * 34 + 102 + 3 * (sl + 1) + k] = (double)xip[k];
    if (sl == 0) { dbg->out[34 * 34 + 144] = s.nefc; dbg->out[34 * 34 + 145] = s.ncon; dbg->out[34 * 34 + 146] = s.solver_iter; }
  }
}

}  // namespace dm
