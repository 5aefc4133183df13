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

#include <cstddef>

#include "env_kernel.h"

namespace dm {

constexpr int SLOTS = 4, SW = 16;            // environments per wavefront, lanes per environment
#ifndef DM_SLOT_MAXROWS
#define DM_SLOT_MAXROWS 40          // (32: the two-set capacity of rounds 3-4 — the three-set code is then not compiled; experiments only)
#endif
constexpr int SLOT_MAXROWS = DM_SLOT_MAXROWS, SLOT_MAXLIMROWS = 16, SLOT_MAXCON = 13, SLOT_MAXFRAME = 8, SLOT_MAXCAND = 32, SLOT_BOXSLOTS = 3;   // per-environment capacities of the packed path; beyond: fix-up by the one-env kernel
constexpr int SLOT_EXTROWS = SLOT_MAXROWS - 2 * SW;
static_assert(SLOT_EXTROWS == 8 || SLOT_EXTROWS == 0, "the partial third row set holds eight rows (or is absent)");     // rows 32 .. 39: the partial third row set of slot_constraint<3> (owned by the even lanes)
constexpr int PAIR_PASSES = MAXPAIR / SW;
static_assert(MAXPAIR <= 128 && NG <= 16, "a candidate word holds the pair number in 7 bits and both geom numbers in 4 bits each");
constexpr int DOF_PASSES = (NV + SW - 1) / SW, HINGE_PASSES = (NU + SW - 1) / SW, Q_PASSES = (NQ + SW - 1) / SW, ENT_PASSES = (310 + SW - 1) / SW;

// per-slot LDS working set.  r1 / r2 are reused along one forward evaluation:
//   r1: kinematics scratch (body quaternions, hinge half-angle sines, frame offsets) | RNE scratch | M-build scratch (fdof)
//   r2: spatial / composite inertias (kinematics .. M build)                        | the L^T D L factor (M build .. end of the evaluation)
// (alignas(16): the slots' stride must stay a multiple of 16 bytes — a lane's LDS address is slot base + constant, and only then can the compiler prove
//  the 16-byte alignment its ds_read_b128 / ds_write_b128 need.  A stride of 9 720 B — the first round-5 layout, 24 B more than round 4's 9 696 — turned
//  all 1 350 of them into pairs of 8-byte accesses and cost every workload 8 %: gpurun calls h2 / h3.)
// The slot lane number reaches every stage through dmw::launder (the optimiser must not fold it: address arithmetic hoisted out of the RK loop ends up in
// scratch memory).  Its RANGE may be known, though (dmw::launder_slot_lane): `d = sl + 16 c < NV` is then true at compile time for the first two of the
// three passes over the 34 dofs, and those passes need no lane predicate — no exec-masked block each, the loads of all passes in one scheduling region.
// DM_SLOT_ASSUME: bit k = stage k (kinematics, bias, mass matrix, rows, constraint) knows the range.  Measured per mask with the hoisted pair codes of the
// elimination (profiles/r05_ab_kernel_variants.md section 6): all five +1.3 % on the horizon launch, none -2.8 %.
#ifndef DM_SLOT_ASSUME
#define DM_SLOT_ASSUME 31
#endif
#define DM_SLOT_LANE_AT(bit, x) (((DM_SLOT_ASSUME) >> (bit)) & 1 ? dmw::launder_slot_lane(x) : dmw::launder(x))

template <class R>
struct alignas(16) SlotShared {
  R qpos[36], qvel[NV];
  R xpos[NB][3], xmat[NB][9];
  R tau[NV];
  union {
    struct { R qacc[NV], dinv[NV]; } o;                         // result of the evaluation | 1 / D of the factorisation (M build only)
    struct { R axes[2][3][3], poly[2][8][3]; } bb;              // box-box scratch of the narrow phase (both are dead then)
  } qd;
  R cdof[NV][6];
  R dsq[NV], act[NV], qws[NV];
  union {
    struct { R xquat[NB][4], sc[NU][2], off[NB][3]; } k;
    struct { R cvel[NB][6], cacc[NB][6], cfrc[NB][6]; } v;      // (the subtree sums of cfrc overwrite cvel)
    R fdof[NV][6];
    struct {                                                    // collision .. constraint stage
      R gpos[NG][3];                                            // geom world positions (orientations are re-formed per candidate pair)
      R boxc[SLOT_BOXSLOTS][4][4];                              // contacts (dist, pos) of plane-box (slots 0, 1) and box-box (slot 2)
      R con[SLOT_MAXCON][4];                                    // staged contacts: pos[3], dist (emission .. row build)
      R frm[SLOT_MAXFRAME][6];                                  // contact frames, one per pair with contacts: normal[3], tangent 1 [3]
      R rowv[SLOT_MAXLIMROWS];                                  // limit rows (they come first): distance
      int rowi[SLOT_MAXROWS];                                   // row codes (see slot_rows)
      unsigned short cand[SLOT_MAXCAND];                        // candidate pair numbers past the broad phase, in pair-list order (16 bits each: the eight more
                                                                //  row codes of round 5 must not move anything behind them — see the layout note below)
      int coni[SLOT_MAXCON + 1];                                // pair number | frame << 8 of a staged contact
#if DM_SLOT_MAXROWS == 40
      char keep_r4_layout_[32];
#endif
    } rw;
  } r1;
  union {
    struct { R sin[NB][10], crb[NB][10]; } i;
    R qLD[312];
  } r2;
  int nefc, ncon, status, solver_iter;
  // horizon launches: qpos[35] — the spare element behind the 35 coordinates — is the slot's `kin_ok` flag: nonzero = the kinematics in this block
  // are those of the env's current state (slot_step.h kin_carry).  (A field of its own changes the slots' stride: measured 1.5 % slower.)
  DM_DEV R& kin_ok() { return qpos[NQ]; }
};
static_assert(NQ == 35, "qpos[36] has one spare element");
static_assert(sizeof(SlotShared<double>) % 16 == 0 && sizeof(SlotShared<float>) % 16 == 0, "slot stride: a multiple of 16 bytes");
// Layout note (round 5, gpurun calls h2-h5): the float64 slot is kept at round 4's 9 696 bytes with `r2` at its old offset.  Growing `rowi` from 32 to 40
// codes in place (r2 and the tail 32 B further back, stride 9 720 or 9 728) made EVERY workload 8 % slower — 16.5 against 18.2 M env-steps/s on the judged
// line, with or without the three-set code compiled in — while the same code on the old offsets is 1.7 % faster than round 4.
#if DM_SLOT_MAXROWS == 40
static_assert(sizeof(SlotShared<double>) == 9696, "the float64 slot layout is a measured quantity: see the layout note");
#endif
static_assert(offsetof(SlotShared<double>, cdof) == offsetof(SlotShared<double>, qd) + sizeof(((SlotShared<double>*)0)->qd) &&
              offsetof(SlotShared<float>, cdof) == offsetof(SlotShared<float>, qd) + sizeof(((SlotShared<float>*)0)->qd) &&
              sizeof(((SlotShared<double>*)0)->qd) + sizeof(((SlotShared<double>*)0)->cdof) >= sizeof(double) * NV * (SLOT_MAXROWS - 2 * SW),
              "slot_constraint<3> parks the surplus rows' half-solved vectors in the adjacent qd + cdof regions");
// index tables shared by the four slots of a workgroup (compile-time topology; see LaneTables)
// TOPO.dof_body / TOPO.madr of a lane's dofs come from the wave's LDS tables (104 B) instead of constant memory — in the bias, mass-matrix and
// D stages they were the first link of a dependent chain, an L2 round trip each on a lone wave (the L1 is flushed by the callee-saved registers' scratch traffic):
// +0.8 % on the horizon launch (gpurun call g6).  The same two tables as bit fields of a lane register: -3 % (profiles/r05_ab_kernel_variants.md section 6).
#ifndef DM_ENTRY_GROUP
#define DM_ENTRY_GROUP 5
#endif
constexpr bool root_alone_at_depth_one() { for (int b = 2; b < NB; b++) if (TOPO.body_depth[b] <= 1) return false; return TOPO.body_depth[1] == 1; }
static_assert(root_alone_at_depth_one(), "slot_bias does the root body (the only one at depth 1) ahead of the level loop");
struct SlotTables {
  unsigned short tab_dst[NV][14];
  unsigned short tab_ent[312];
  unsigned short madr[NV];
  unsigned char dof_body[NV + 2];
};
static_assert(sizeof(SlotShared<double>) * SLOTS + sizeof(SlotTables) <= 40 * 1024, "four waves (four environments and the tables each) must fit a CU's 160 KB of LDS");
#define DM_DOF_BODY(tb, d) ((int)(tb).dof_body[d])
#define DM_DOF_MADR(tb, d) ((int)(tb).madr[d])
DM_DEV void stage_slot_tables(SlotTables& t, int lane) {
#pragma unroll
  for (int c = 0; c < (312 + 63) / 64; c++) { const int e = lane + 64 * c; if (e < 312) t.tab_ent[e] = LTAB.tab_ent[e]; }
#pragma unroll
  for (int c = 0; c < (NV * 14 + 63) / 64; c++) { const int i = lane + 64 * c; if (i < NV * 14) (&t.tab_dst[0][0])[i] = LTAB.tab_dst[i]; }
  if (lane < NV) { t.madr[lane] = (unsigned short)TOPO.madr[lane]; t.dof_body[lane] = (unsigned char)TOPO.dof_body[lane]; }
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
  const int sl = DM_SLOT_LANE_AT(0, sl_in);
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
DM_DEV void slot_bias(const DevModel<R>& M, SlotShared<R>& s, const SlotTables& tb, int sl_in, const LaneTopo& lt) {
  const int sl = DM_SLOT_LANE_AT(1, sl_in);
  const int b = sl + 1;
  const bool isbody = sl < NB - 1;
  const int depth = dmw::launder(lt.depth), p = dmw::launder(lt.parent) & 15, da = dmw::launder(lt.dofadr), nd = dmw::launder(lt.dofnum);
  if (sl == 0) {
    for (int r = 0; r < 6; r++) { s.r1.v.cvel[0][r] = 0; s.r1.v.cacc[0][r] = 0; }
    s.r1.v.cacc[0][3] = -M.gravity[0]; s.r1.v.cacc[0][4] = -M.gravity[1]; s.r1.v.cacc[0][5] = -M.gravity[2];
  }
  R S[6] = {0, 0, 0, 0, 0, 0}, C[6] = {0, 0, 0, 0, 0, 0};
  // A lone wave waits out every LDS round trip it takes: operands are requested together, a scheduling fence keeps the arithmetic behind them.  The root body
  // (the only one at depth 1, a static property of the tree) needs nothing but the world's constants: it is done here, beside the other bodies' joint
  // velocities, and the level loop starts at depth 2 — one level and one hand-off less, same arithmetic.
  if (isbody && b > 1) {
    R qd3[3], cd3[3][6];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int dk = k < nd ? da + k : da;
      qd3[k] = s.qvel[dk];
#pragma unroll
      for (int r = 0; r < 6; r++) cd3[k][r] = s.cdof[dk][r];
    }
    dmw::sched_fence();
#pragma unroll
    for (int k = 0; k < 3; k++) if (k < nd) {
      const R qd = qd3[k];
      R T[6], cd[6];
      for (int r = 0; r < 6; r++) T[r] = cd3[k][r] * qd;
      if (k > 0) { cross_motion(cd, S, T); for (int r = 0; r < 6; r++) C[r] += cd[r]; }
      for (int r = 0; r < 6; r++) S[r] += T[r];
    }
  } else if (b == 1) {
    R v[6], a[6], qd6[6], cd6[6][6];
#pragma unroll
    for (int r = 0; r < 6; r++) { v[r] = s.r1.v.cvel[0][r]; a[r] = s.r1.v.cacc[0][r]; }      // (written by this very lane above: the LDS keeps a wave's accesses in order)
#pragma unroll
    for (int k = 0; k < 6; k++) {
      qd6[k] = s.qvel[k];
#pragma unroll
      for (int r = 0; r < 6; r++) cd6[k][r] = s.cdof[k][r];
    }
    dmw::sched_fence();
    for (int k = 0; k < 3; k++) { const R qd = qd6[k]; for (int r = 0; r < 6; r++) v[r] += cd6[k][r] * qd; }
    R vb[6];
    for (int r = 0; r < 6; r++) vb[r] = v[r];
    for (int k = 3; k < 6; k++) {
      R cd[6]; cross_motion(cd, vb, cd6[k]);
      const R qd = qd6[k];
      for (int r = 0; r < 6; r++) { a[r] += cd[r] * qd; v[r] += cd6[k][r] * qd; }
    }
    for (int r = 0; r < 6; r++) { s.r1.v.cvel[1][r] = v[r]; s.r1.v.cacc[1][r] = a[r]; }
  }
  dmw::sync();
#pragma unroll
  for (int L = 2; L <= MAXDEPTH_BODY; L++) {
    if (isbody && depth == L) {
      R v[6], a[6];
      for (int r = 0; r < 6; r++) { v[r] = s.r1.v.cvel[p][r]; a[r] = s.r1.v.cacc[p][r]; }
      R cd[6];
      cross_motion(cd, v, S);
      for (int r = 0; r < 6; r++) { a[r] += cd[r] + C[r]; v[r] += S[r]; }
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
  {
    R cdd[DOF_PASSES][6], cvb[DOF_PASSES][6], qv[DOF_PASSES], ac[DOF_PASSES], dmp[DOF_PASSES];
#pragma unroll
    for (int c = 0; c < DOF_PASSES; c++) {
      const int d = sl + SW * c, dd = d < NV ? d : 0, bd = DM_DOF_BODY(tb, dd);
#pragma unroll
      for (int r = 0; r < 6; r++) { cdd[c][r] = s.cdof[dd][r]; cvb[c][r] = s.r1.v.cvel[bd][r]; }
      qv[c] = s.qvel[dd]; ac[c] = s.act[dd]; dmp[c] = M.dof_damping[dd];
    }
    dmw::sched_fence();
#pragma unroll
    for (int c = 0; c < DOF_PASSES; c++) {
      const int d = sl + SW * c;
      const R bias = dot6(cdd[c], cvb[c]);
      const R t = -dmp[c] * qv[c] - bias + ac[c];
      if (d < NV) s.tau[d] = t;
    }
  }
  dmw::sync();
}

// ---- mass matrix and its L^T D L factor (env_kernel.h stage_mass_matrix).  The elimination steps are those of ELIM_STEPS; the
// columns of a step are taken one after the other by the slot's 16 lanes, in the order (pass, column, pair) in which the one-env
// kernel's lane groups apply them, so that entries which several limbs update receive their contributions in the same order. ---------
// The chunks (16 consecutive pair numbers of one column) of a step, in the one-env kernel's order of application (pass, column, sub-chunk)
// A lane's pair codes are decoded once per evaluation and used unmasked by all 15 elimination steps (slot_eliminate_step) — round 4 decoded and masked them per chunk,
// the lane number laundered per step: 1 304 -> 775 integer instructions in the stage, elimination 54.7 k -> 35.8 k cycles per wave-step (round 5).
struct SlotElimChunks { int n; int K[24]; int t0[24]; };
constexpr bool elim_unmasked_reads_fit() { for (int K = 0; K < NV; K++) if (TOPO.madr[K] + 14 >= 312) return false; return true; }
static_assert(elim_unmasked_reads_fit(), "an unmasked pair code (e, a <= 14) must address inside qLD[312] from every column's base");
constexpr SlotElimChunks make_slot_elim_chunks(int S) {
  SlotElimChunks c{};
  const ElimStep st = ELIM_STEPS[S];
  for (int P = 0; P < elim_passes(S); P++)
    for (int ci = 0; ci < st.ncol; ci++) {
      const int gs = elim_group_size(st.ncol, ci), np = elim_npairs(st.K[ci]);
      for (int u = 0; u < gs / SW; u++) {
        const int t0 = gs * P + SW * u;
        if (t0 < np) { c.K[c.n] = st.K[ci]; c.t0[c.n] = t0; c.n++; }
      }
    }
  return c;
}
constexpr bool slot_elim_chunks_fit() { for (int S = 0; S < N_ELIM_STEPS; S++) if (make_slot_elim_chunks(S).n > 24) return false; return true; }
static_assert(slot_elim_chunks_fit(), "a step has at most 24 chunks");
// One elimination step.  A column only READS its own row and only WRITES rows of its proper ancestors, and the columns of a step lie in
// different branches: no update of the step touches anything the step reads.  So all operands are fetched first (one LDS round trip for
// the whole step instead of one per chunk: a lone wave has nothing to hide them behind), the products formed, and the updates then
// issued back to back as fire-and-forget LDS atomics in the one-env kernel's order.
template <int S, class R>
DM_DEV void slot_eliminate_step(SlotShared<R>& s, const SlotTables& tb, int sl, const LaneTopo& lt) {
  constexpr SlotElimChunks CH = make_slot_elim_chunks(S);
  constexpr ElimStep st = ELIM_STEPS[S];
  R inv[4];
#pragma unroll
  for (int ci = 0; ci < st.ncol; ci++) inv[ci] = s.r2.qLD[TOPO.madr[st.K[ci]]];
  R xe[CH.n], xa[CH.n];
  int dst[CH.n];
  bool on[CH.n];
#pragma unroll
  for (int c = 0; c < CH.n; c++) {
    const int K = CH.K[c], np = elim_npairs(K), base = TOPO.madr[K];
    const int t = CH.t0[c] + sl;
    on[c] = t < np;
    // the pair code of a lane depends on (lane, t0 / 16) only: unmasked (a lane past the column's pairs reads inside qLD — madr[K] + 14 <= 311 — and
    // keeps its update to itself), the codes and the addresses formed from them are the same in all 15 steps and the compiler keeps them
    const int j = CH.t0[c] >> 4;                     // (a constant after unrolling)
    const int code = (int)(lt.tri >> (8 * j)) & 0xff, e = code >> 4, a = code & 15;
    dst[c] = tb.tab_dst[K][a] + (e - a);
    xe[c] = s.r2.qLD[base + e]; xa[c] = s.r2.qLD[base + a];
  }
#pragma unroll
  for (int ci = 0; ci < st.ncol; ci++) inv[ci] = dmw::rcp_fast(inv[ci]);
#pragma unroll
  for (int c = 0; c < CH.n; c++) {
    int ci = 0;
#pragma unroll
    for (int q = 0; q < st.ncol; q++) if (st.K[q] == CH.K[c]) ci = q;
    xe[c] = xe[c] * (xa[c] * inv[ci]);
  }
  dmw::reload_fence();
#pragma unroll
  for (int c = 0; c < CH.n; c++) dmw::lds_sub(on[c], &s.r2.qLD[dst[c]], xe[c]);
  dmw::sync();
}
template <int S, class R>
struct SlotEliminateFrom {
  static DM_DEV void run(SlotShared<R>& s, const SlotTables& tb, int sl, const LaneTopo& lt) {
    if constexpr (S < N_ELIM_STEPS) {
      slot_eliminate_step<S, R>(s, tb, sl, lt);
      SlotEliminateFrom<S + 1, R>::run(s, tb, sl, lt);
    }
  }
};
static_assert(elim_group_size(4, 0) % SW == 0 && elim_group_size(3, 2) % SW == 0 && elim_group_size(1, 0) % SW == 0, "lane groups are whole slots");
// (pair codes of a lane cover t = (lane & 15) + 16 j, j < 6, i.e. t < 96: elim_codes_ok)

template <class R, bool PROF = false>
DM_DEV void slot_mass_matrix(const DevModel<R>& M, SlotShared<R>& s, const SlotTables& tb, int sl_in, const LaneTopo& lt, const DebugOut* dbg, long long* prof = 0) {
  const int sl = DM_SLOT_LANE_AT(2, sl_in);
  long long pt0 = 0, pt1 = 0;
  if (PROF) pt0 = dmw::clk();
#define SLOT_MSTAMP(k) if (PROF) { pt1 = dmw::clk(); prof[k] += pt1 - pt0; pt0 = pt1; }
  // f_d = (composite inertia of the dof's body) cdof_d: the inertias are read before the region they share with the factor is written
  R f[DOF_PASSES][6];
#pragma unroll
  for (int c = 0; c < DOF_PASSES; c++) {
    const int d = sl + SW * c;
    if (d < NV) sinert_mul(f[c], s.r2.i.crb[DM_DOF_BODY(tb, d)], s.cdof[d]);
  }
#pragma unroll
  for (int c = 0; c < DOF_PASSES; c++) {
    const int d = sl + SW * c;
    if (d < NV) {
      for (int r = 0; r < 6; r++) s.r1.fdof[d][r] = f[c][r];
      s.qd.o.dinv[d] = M.dof_armature[d];
    }
  }
  int ijc[ENT_PASSES];                      // (i << 8) | j of this lane's entries, fetched ahead of the hand-off: the loop below then has no dependent look-up
#pragma unroll
  for (int c = 0; c < ENT_PASSES; c++) { const int e = sl + SW * c; ijc[c] = tb.tab_ent[e < 312 ? e : 0]; }
  dmw::sync();
  // All twenty passes' entries are formed in registers first and stored afterwards, and the armature of a diagonal entry is added by SELECT (its operand
  // fetched by every lane).  With a store and a branch per pass the passes were twenty scheduling regions, each waiting out its own four LDS round trips:
  // 10 k of the stage's 12 k cycles per evaluation on a lone wave.  Same values, same order of operations per entry.
  R ent[ENT_PASSES];
  constexpr int EG = DM_ENTRY_GROUP;         // passes whose operands are in flight together (52 LDS reads per group of four)
  static_assert(ENT_PASSES % EG == 0, "whole groups of passes");
#pragma unroll
  for (int g = 0; g < ENT_PASSES / EG; g++) {
    R ca[EG][6], fb[EG][6], dg[EG];
#pragma unroll
    for (int u = 0; u < EG; u++) {
      const int ij = ijc[EG * g + u], i = ij >> 8, j = ij & 0xff;
#pragma unroll
      for (int r = 0; r < 6; r++) { ca[u][r] = s.cdof[j][r]; fb[u][r] = s.r1.fdof[i][r]; }
      dg[u] = s.qd.o.dinv[i];
    }
    dmw::sched_fence();                      // the group's loads above, its arithmetic below: one LDS round trip per group instead of four per pass
#pragma unroll
    for (int u = 0; u < EG; u++) {
      const int ij = ijc[EG * g + u], i = ij >> 8, j = ij & 0xff;
      const R v = dot6(ca[u], fb[u]);
      const R vd = v + dg[u];
      ent[EG * g + u] = i == j ? vd : v;
    }
  }
#pragma unroll
  for (int c = 0; c < ENT_PASSES; c++) {
    const int e = sl + SW * c;
    if ((c + 1) * SW <= TOPO.nM || e < TOPO.nM) {
      s.r2.qLD[e] = ent[c];
      if (dbg) { const int ij = ijc[c], i = ij >> 8, j = ij & 0xff; dbg->out[i * NV + j] = (double)ent[c]; dbg->out[j * NV + i] = (double)ent[c]; }
    }
  }
  dmw::sync();
  SLOT_MSTAMP(16)
  LaneTopo le = lt;
  le.tri = dmw::launder(lt.tri);            // what the steps derive from the pair codes stays inside this evaluation (not hoisted out of the RK loop and spilled)
  SlotEliminateFrom<0, R>::run(s, tb, sl, le);
  SLOT_MSTAMP(17)
  {
    R dgl[DOF_PASSES];                       // (likewise: the three diagonals first, then the three division + square-root chains side by side)
#pragma unroll
    for (int c = 0; c < DOF_PASSES; c++) { const int d = sl + SW * c, dd = d < NV ? d : 0; dgl[c] = s.r2.qLD[DM_DOF_MADR(tb, dd)]; }
    dmw::sched_fence();
#pragma unroll
    for (int c = 0; c < DOF_PASSES; c++) {
      const int d = sl + SW * c;
      const R inv = R(1) / dgl[c], sq = sqrt(inv);
      if (d < NV) { s.qd.o.dinv[d] = inv; s.dsq[d] = sq; }
    }
  }
  dmw::sync();
  {
    R sc[ENT_PASSES], di[ENT_PASSES];        // (likewise: all forty operands, a fence, the products, the stores)
#pragma unroll
    for (int c = 0; c < ENT_PASSES; c++) {
      const int e = sl + SW * c, ee = ((c + 1) * SW <= TOPO.nM || e < TOPO.nM) ? e : 0;
      const int ij = ijc[c], i = ij >> 8;
      di[c] = s.qd.o.dinv[i];
      sc[c] = s.r2.qLD[ee];
    }
    dmw::sched_fence();
#pragma unroll
    for (int c = 0; c < ENT_PASSES; c++) {
      const int ij = ijc[c], i = ij >> 8, j = ij & 0xff;
      sc[c] = sc[c] * (i != j ? di[c] : R(1));
    }
#pragma unroll
    for (int c = 0; c < ENT_PASSES; c++) {
      const int e = sl + SW * c;
      if ((c + 1) * SW <= TOPO.nM || e < TOPO.nM) s.r2.qLD[e] = sc[c];
    }
  }
  dmw::sync();
  SLOT_MSTAMP(18)
#undef SLOT_MSTAMP
}

// ---- small collectives over the four rows ------------------------------------------------------------------------------------------
// maximum over the wave of a value that is uniform inside each row -> wave-uniform (SGPR)
DM_DEV int rows_max(int v) {
  const int a = dmw::bcast_i(v, 0), b = dmw::bcast_i(v, 16), c = dmw::bcast_i(v, 32), d = dmw::bcast_i(v, 48);
  const int ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
// exclusive prefix sum over the lanes of the own row of a small non-negative int (< 32): one ballot + popcount per bit
DM_DEV int row_exclusive_scan(int v, int sl, int lane, int* total) {
  const unsigned below = (1u << sl) - 1u;
  int pre = 0, tot = 0;
#pragma unroll
  for (int b = 0; b < 5; b++) {
    const unsigned m = dmw::row_ballot(((v >> b) & 1) != 0, lane);
    pre += __builtin_popcount(m & below) << b;
    tot += __builtin_popcount(m) << b;
  }
  *total = tot;
  return pre;
}

// ---- constraint rows of the slot's environment: joint limits first (joint order), then contacts (pair-list order)
//      [MJ mj_collision, mj_makeConstraint]; env_kernel.h stage_rows re-mapped:
//   * lane g forms the world POSITION of geom g; orientations are only formed for the pairs that pass the bounding spheres;
//   * the 104 candidate pairs take 7 broad-phase passes of 16 lanes; survivors are compacted (in pair-list order) onto the lanes, so
//     the divergent narrow phase and the row emission run ONCE for all four environments of the wave;
//   * what a row's lane needs later is staged per CONTACT (position, normal, first tangent, distance) plus one code word per row;
//     the row's Jacobian wrench is formed in registers by the constraint stage.
// Row code: limit  ROW_LIMIT   | dof << 8 | (sign > 0) << 16;   contact  ROW_CONTACT | contact << 8 | pyramid edge q << 16 | condim << 20.
// Capacities (SLOT_MAXCAND candidates, SLOT_MAXCON contacts, SLOT_MAXROWS rows): an environment that exceeds one is flagged (`ovf`: bit 0
// candidates, 1 box slots, 2 contacts, 3 rows, 4 a PGS step the cost test would reject, 6 any) and re-stepped by the one-env kernel;
// nothing of it is stored by this wave.  Returns the slot's row count.
// MAXR = the row capacity of the instantiation: 2 * SW (the two-set code only) or SLOT_MAXROWS (with the partial third set, slot_constraint<3>).
template <class R, bool PROF = false, int MAXR = 2 * SW>
DM_DEV int slot_rows(const DevModel<R>& M, SlotShared<R>& s, int sl_in, int lane_in, int& ovf, long long* prof = 0) {
  static_assert(MAXR == 2 * SW || MAXR == SLOT_MAXROWS, "row capacity: two sets, or two and the partial third");
  const int sl = DM_SLOT_LANE_AT(3, sl_in), lane = dmw::launder(lane_in);
  long long pt0 = 0, pt1 = 0;
  if (PROF) pt0 = dmw::clk();
#define SLOT_RSTAMP(k) if (PROF) { pt1 = dmw::clk(); prof[k] += pt1 - pt0; pt0 = pt1; }
  const unsigned below = (1u << sl) - 1u;
  const int npair = dmw::uniform(M.npair);
  int nrow = 0;
  auto& W = s.r1.rw;
  // broad-phase operands of this lane's pairs sl + 16 p, requested up front
  int pk[PAIR_PASSES]; R pbound[PAIR_PASSES];
#pragma unroll
  for (int p = 0; p < PAIR_PASSES; p++) {
    const int pidx = sl + SW * p, pr = pidx < npair ? pidx : 0;
    pk[p] = M.pair_rec[pr].g1 | (M.pair_rec[pr].g2 << 8) | ((M.pair_rec[pr].t1t2 & 0xff) << 16);
    pbound[p] = M.pair_rec[pr].bound;
  }
  // (round 6) every model constant of this section — the lane's geom (body, local position) and its two hinges' limit flags and bounds — is requested HERE, beside the
  // pair records, unconditionally: the section used to fetch them one after the other behind lane predicates (flag -> lower bound -> upper bound, per hinge pass:
  // eight exposed L2 round trips per evaluation on a wave that has nothing else to run).  Same arithmetic on the same operands: bit-identical rows.
  int gb0 = 0; R gp0[3] = {0, 0, 0};
  if (M.enable_contact) { gb0 = M.geom_body[sl]; gp0[0] = M.geom_pos[sl][0]; gp0[1] = M.geom_pos[sl][1]; gp0[2] = M.geom_pos[sl][2]; }
  int jl[HINGE_PASSES]; R jlo[HINGE_PASSES], jhi[HINGE_PASSES];
#pragma unroll
  for (int c = 0; c < HINGE_PASSES; c++) {
    const int h = sl + SW * c, hh = h < NU ? h + 1 : 1;
    jl[c] = M.enable_limit ? (int)M.jnt_limited[hh] : 0; jlo[c] = M.enable_limit ? M.jnt_lo[hh] : R(0); jhi[c] = M.enable_limit ? M.jnt_hi[hh] : R(0);
  }
  dmw::sched_fence();
  // (measured and dropped: the plane pairs' body + local normal for all seven broad-phase passes requested here as well — 28 more loads, 49 registers: -1.1 %,
  //  profiles/r06_ab_kernel_variants.md section 7)
  if (M.enable_contact) {
    const int g = sl;                                        // NG == SW: one geom per lane
    R v[3];
    mat_vec(v, s.xmat[gb0], gp0);
    W.gpos[g][0] = s.xpos[gb0][0] + v[0]; W.gpos[g][1] = s.xpos[gb0][1] + v[1]; W.gpos[g][2] = s.xpos[gb0][2] + v[2];
  }
  // ---- joint limits: hinge h = sl + 16 c, dof h + 6
  if (M.enable_limit) {
#pragma unroll
    for (int c = 0; c < HINGE_PASSES; c++) {
      const int h = sl + SW * c;
      const bool lim = h < NU && jl[c] != 0;
      const R q = s.qpos[(h < NU ? h : 0) + 7];
      const R dlo = q - jlo[c], dhi = jhi[c] - q;
      const bool vlo = lim && dlo < 0, vhi = lim && !vlo && dhi < 0;
      const bool viol = vlo || vhi;
      const R dist = vlo ? dlo : (vhi ? dhi : R(0));
      const int pos_sign = vlo ? 1 : 0;
      const unsigned mask = dmw::row_ballot(viol, lane);
      if (viol) {
        const int r = nrow + __builtin_popcount(mask & below);
        if (r < SLOT_MAXLIMROWS) { W.rowi[r] = ROW_LIMIT | ((h + 6) << 8) | (pos_sign << 16); W.rowv[r] = dist; }
      }
      nrow += __builtin_popcount(mask);
    }
    if (nrow > SLOT_MAXLIMROWS) ovf |= 8;
  }
  dmw::sync();
  SLOT_RSTAMP(19)
  int ncon = 0, nfr = 0;
  if (M.enable_contact) {
    // ---- broad phase: bounding spheres (plane pairs: signed distance of the centre)
    int ncand = 0;
#pragma unroll
    for (int p = 0; p < PAIR_PASSES; p++) {
      if (p * SW < npair) {
        const int pidx = sl + SW * p;
        const int g1 = pk[p] & 0xff, g2 = (pk[p] >> 8) & 0xff, t1 = (pk[p] >> 16) & 0xff;
        bool cand = false;
        if (pidx < npair) {
          const R d[3] = {W.gpos[g2][0] - W.gpos[g1][0], W.gpos[g2][1] - W.gpos[g1][1], W.gpos[g2][2] - W.gpos[g1][2]};
          if (t1 == GEOM_PLANE) {
            // plane normal = third column of its world orientation  xmat[body] * geom_mat
            const int gb = M.geom_body[g1];
            const R* a = s.xmat[gb]; const R* bm = M.geom_mat[g1];
            const R b2 = bm[2], b5 = bm[5], b8 = bm[8];
            const R nx = a[0] * b2 + a[1] * b5 + a[2] * b8, ny = a[3] * b2 + a[4] * b5 + a[5] * b8, nz = a[6] * b2 + a[7] * b5 + a[8] * b8;
            cand = d[0] * nx + d[1] * ny + d[2] * nz <= pbound[p];
          } else cand = dot3(d, d) <= pbound[p] * pbound[p];
        }
        const unsigned mask = dmw::row_ballot(cand, lane);
        // candidate word: pair number | geom 1 << 7 | geom 2 << 11 (round 6: the narrow phase asks for the pair record AND both geoms' local frames in one go)
        if (cand) { const int k = ncand + __builtin_popcount(mask & below); if (k < SLOT_MAXCAND) W.cand[k] = (unsigned short)(pidx | (g1 << 7) | (g2 << 11)); }
        ncand += __builtin_popcount(mask);
      }
    }
    if (ncand > SLOT_MAXCAND) { ovf |= 1; ncand = SLOT_MAXCAND; }
    const int maxc = rows_max(ncand);
    if (maxc > 0) dmw::sync();
    SLOT_RSTAMP(20)
    if (PROF) prof[23] += maxc;
    for (int trip = 0; trip * SW < maxc; trip++) {            // (one trip unless an environment of the wave has more than 16 candidates)
      // ---- narrow phase: candidate k of the environment on lane k % 16
      const int kc = trip * SW + sl;
      const bool has = kc < ncand;
      const int cw = has ? (int)W.cand[kc] : 0;
      const int pidx = cw & 127;
      const auto& rec = M.pair_rec[pidx];
      const int g1 = (cw >> 7) & 15, g2 = (cw >> 11) & 15;          // (= rec.g1, rec.g2: known before the record arrives)
      const int tt = rec.t1t2, meta = rec.meta;
      R bm1[9], bm2[9];
#pragma unroll
      for (int k = 0; k < 9; k++) { bm1[k] = M.geom_mat[g1][k]; bm2[k] = M.geom_mat[g2][k]; }
      const R margin = rec.margin;
      const R z1[3] = {rec.s1[0], rec.s1[1], rec.s1[2]}, z2[3] = {rec.s2[0], rec.s2[1], rec.s2[2]};
      const int t1 = tt & 0xff, t2 = (tt >> 8) & 0xff, dim = (tt >> 16) & 0xff;
      PairContacts<R> pc;
      pc.n = 0; pc.boxslot = -1;
      if (has) {
        R p1[3], p2[3], m1[9], m2[9];
        {
          const R* a = s.xmat[meta & 0xff]; const R* bm = bm1;
          for (int i = 0; i < 3; i++) for (int jx = 0; jx < 3; jx++) m1[3 * i + jx] = a[3 * i] * bm[jx] + a[3 * i + 1] * bm[3 + jx] + a[3 * i + 2] * bm[6 + jx];
          const R* c = s.xmat[(meta >> 8) & 0xff]; const R* dm2 = bm2;
          for (int i = 0; i < 3; i++) for (int jx = 0; jx < 3; jx++) m2[3 * i + jx] = c[3 * i] * dm2[jx] + c[3 * i + 1] * dm2[3 + jx] + c[3 * i + 2] * dm2[6 + jx];
          for (int k = 0; k < 3; k++) { p1[k] = W.gpos[g1][k]; p2[k] = W.gpos[g2][k]; }
        }
        const int raw = ((meta >> 16) & 0xff) - 1;             // staging slot of the one-env kernel: plane-box 0..3, box-box 4..5
        const int slotb = raw < 0 ? -1 : (raw < 4 ? raw : raw - 2);
        if (slotb >= SLOT_BOXSLOTS || (raw >= 2 && raw < 4)) ovf |= 2;      // (a model with more boxes than the humanoid's two feet)
        else {
          const BoxScratch<R> bx{s.qd.bb.axes, s.qd.bb.poly, W.boxc};
          narrowphase_at(bx, p1, m1, p2, m2, t1, t2, z1, z2, M.pair_rec[pidx].s1, M.pair_rec[pidx].s2, slotb, margin, pc);
        }
      }
      if (rows_max((int)dmw::row_ballot(pc.n > 0, lane)) != 0) {
        // ---- emission: contacts staged in list order, one code word per row
        const int rows_per = dim == 1 ? 1 : 2 * (dim - 1);
        int tot_rows, tot_con;
        const int r0 = nrow + row_exclusive_scan(pc.n * rows_per, sl, lane, &tot_rows);
        const int c0 = ncon + row_exclusive_scan(pc.n, sl, lane, &tot_con);
        const unsigned fmask = dmw::row_ballot(pc.n > 0, lane);
        const int fi = nfr + __builtin_popcount(fmask & below);          // this pair's frame
        nfr += __builtin_popcount(fmask);
        if (pc.n > 0) {
          R fr[9];
          make_frame(fr, pc.nrm, pc.hint);
          if (fi < SLOT_MAXFRAME) { R* o = W.frm[fi]; for (int t = 0; t < 6; t++) o[t] = fr[t]; }
          for (int k = 0; k < pc.n; k++) {
            const int ci = c0 + k, rk = r0 + k * rows_per;
            if (ci >= SLOT_MAXCON || fi >= SLOT_MAXFRAME || rk + rows_per > MAXR) { ovf |= (ci >= SLOT_MAXCON || fi >= SLOT_MAXFRAME) ? 4 : 8; continue; }
            R cdist, cpos[3];
            if (pc.boxslot >= 0) { const R* o = W.boxc[pc.boxslot][k]; cdist = o[0]; cpos[0] = o[1]; cpos[1] = o[2]; cpos[2] = o[3]; }
            else if (k == 0) { cdist = pc.d0; cpos[0] = pc.p0[0]; cpos[1] = pc.p0[1]; cpos[2] = pc.p0[2]; }
            else { cdist = pc.d1; cpos[0] = pc.p1[0]; cpos[1] = pc.p1[1]; cpos[2] = pc.p1[2]; }
            R* o = W.con[ci];
            o[0] = cpos[0]; o[1] = cpos[1]; o[2] = cpos[2]; o[3] = cdist;
            W.coni[ci] = pidx | (fi << 8);
            for (int q = 0; q < rows_per; q++) W.rowi[rk + q] = ROW_CONTACT | (ci << 8) | (q << 16) | (dim << 20);
          }
        }
        nrow += tot_rows; ncon += tot_con;
      }
    }
  }
  // an overflow anywhere in the row -> the whole environment is flagged (ovf is per lane so far)
  if (nrow > MAXR) ovf |= 8;
  if (dmw::row_ballot(ovf != 0, lane) != 0u) { ovf |= 64; nrow = 0; ncon = 0; }    // (bit 6: some lane of the row holds a reason)                             // nothing of this evaluation is used: no row may be read (some were never staged)
  if (sl == 0) { s.nefc = nrow; s.ncon = ncon; }
  dmw::sync();
  SLOT_RSTAMP(21)
#undef SLOT_RSTAMP
  return nrow;
}

// ---- half solves of several register vectors at once (the factor entries are loaded once per row): x_v <- L^-T x_v ---------------
template <int I, int NVEC, class R>
struct SlotSolveLTStep {
  static DM_DEV void run(R (*x)[NV], const R* qLD, const R* cur) {
    R nxt[14];
    dmw::reload_fence();
    load_factor_row<I - 1>(nxt, qLD);
    dmw::sched_fence();
#pragma unroll
    for (int v = 0; v < NVEC; v++) {
#pragma unroll
      for (int a = 1; a < 14; a++) { const int j = TOPO.dof_anc[I][a]; if (j >= 0) x[v][j] -= cur[a] * x[v][I]; }
      dmw::pin_value(x[v][TOPO.dof_anc[I][1]]);
    }
    SlotSolveLTStep<I - 1, NVEC, R>::run(x, qLD, nxt);
  }
};
template <int NVEC, class R> struct SlotSolveLTStep<0, NVEC, R> { static DM_DEV void run(R (*)[NV], const R*, const R*) {} };
template <int NVEC, class R> DM_DEV void slot_solve_LT(R (*x)[NV], const R* qLD) {
  // (measured, round 6: the factor rows requested TWO rows ahead here and in the forward solve — nothing, 20.85 against 20.85 M: profiles/r06_ab_kernel_variants.md section 10)
  R cur[14];
  dmw::reload_fence();
  load_factor_row<NV - 1>(cur, qLD);
  SlotSolveLTStep<NV - 1, NVEC, R>::run(x, qLD, cur);
}

// Jacobian rows of NS row sets at once (env_kernel.h RowStep; the dof's operands are loaded once for all sets)
template <int D, class R>
DM_DEV void slot_load_dof_operands(R* dst, const SlotShared<R>& s) {
  if constexpr (D < NV) {
#pragma unroll
    for (int r = 0; r < 6; r++) dst[r] = s.cdof[D][r];
    dst[6] = s.qvel[D]; dst[7] = s.qws[D];
  }
}
template <int D, int NS, class R>
struct SlotRowStep {
  static DM_DEV void run(R (*y)[NV], RowAcc<R>* ra, const SlotShared<R>& s, const R* cur) {
    R nxt[8];
    dmw::reload_fence();
    slot_load_dof_operands<D + 1>(nxt, s);
    dmw::sched_fence();
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const int pb = (int)(D < 32 ? (ra[k].plus_lo >> D) & 1u : (ra[k].plus_hi >> (D - 32)) & 1u);
      const int mb = (int)(D < 32 ? (ra[k].minus_lo >> D) & 1u : (ra[k].minus_hi >> (D - 32)) & 1u);
      R j = ra[k].w[0] * cur[0] + ra[k].w[1] * cur[1] + ra[k].w[2] * cur[2] + ra[k].w[3] * cur[3] + ra[k].w[4] * cur[4] + ra[k].w[5] * cur[5];
      j = (j + ra[k].w7) * (R)(pb - mb);
      ra[k].vel += j * cur[6]; ra[k].jws += j * cur[7];
      y[k][D] = j;
      dmw::pin_value(ra[k].vel); dmw::pin_value(ra[k].jws);
    }
    SlotRowStep<D + 1, NS, R>::run(y, ra, s, nxt);
  }
};
template <int NS, class R> struct SlotRowStep<NV, NS, R> { static DM_DEV void run(R (*)[NV], RowAcc<R>*, const SlotShared<R>&, const R*) {} };

// ---- nested unrolling over the wave's largest row count (wave-uniform `nmax`): column / row group c is only reached through c - 4 ---
// column CC of A = Y Y^T: AR[k][CC] = Y_(row of the lane, set k) . Y_CC, the partner row broadcast inside the DPP row
template <int CC, int NS, class R>
DM_DEV void slot_acol(R (*AR)[16 * NS], const R (*y)[NV]) {
#pragma unroll
  for (int k = 0; k < NS; k++) {
    R acc0 = 0, acc1 = 0;                 // two partial sums (even / odd dofs): halves the dependent chain of the 34 fused multiply-adds
#pragma unroll
    for (int d = 0; d + 8 <= NV; d += 8) dmw::row_fmac8<CC % 16>(acc0, acc1, &y[CC / 16][d], &y[k][d]);
#pragma unroll
    for (int d = NV / 8 * 8; d < NV; d += 2) { dmw::row_fmac_old<CC % 16>(acc0, y[CC / 16][d], y[k][d]); dmw::row_fmac_old<CC % 16>(acc1, y[CC / 16][d + 1], y[k][d + 1]); }
    AR[k][CC] = acc0 + acc1;
  }
}
template <int C, int NS, class R>
struct SlotACols {
  static DM_DEV void run(R (*AR)[16 * NS], const R (*y)[NV], int nmax) {
    if constexpr (C < 16 * NS) {
      slot_acol<C, NS, R>(AR, y); slot_acol<C + 1, NS, R>(AR, y); slot_acol<C + 2, NS, R>(AR, y); slot_acol<C + 3, NS, R>(AR, y);
      if (C + 4 < nmax) SlotACols<C + 4, NS, R>::run(AR, y, nmax);
    }
  }
};
// warm start: AR[k][c] scaled in place (row by -1 / A_rr);  t_k += A_scaled[k][c] f_c
template <int CC, int NS, class R>
DM_DEV void slot_warm_col(R (*AR)[16 * NS], R* t, const R* f, const R* ndinv) {
#pragma unroll
  for (int k = 0; k < NS; k++) { AR[k][CC] *= ndinv[k]; dmw::row_fmac_old<CC % 16>(t[k], f[CC / 16], AR[k][CC]); }
}
template <int C, int NS, class R>
struct SlotWarm {
  static DM_DEV void run(R (*AR)[16 * NS], R* t, const R* f, const R* ndinv, int nmax) {
    if constexpr (C < 16 * NS) {
      slot_warm_col<C, NS, R>(AR, t, f, ndinv); slot_warm_col<C + 1, NS, R>(AR, t, f, ndinv); slot_warm_col<C + 2, NS, R>(AR, t, f, ndinv); slot_warm_col<C + 3, NS, R>(AR, t, f, ndinv);
      if (C + 4 < nmax) SlotWarm<C + 4, NS, R>::run(AR, t, f, ndinv, nmax);
    }
  }
};
// one PGS row (scaled-residual form of env_kernel.h): delta = max(-f, t) of the row's own lane, broadcast, t += A_s[:, row] delta; the own
// lane's residual at its row is kept through a one-hot multiply-add (oh[i] = 1 in slot lane i)
// four consecutive rows CC .. CC + 3 (CC a multiple of 4: all in one row set) as ONE assembly block (wave.h pgs_rows4 / pgs_rows4_2)
template <int CC, int NS, class R>
DM_DEV void slot_sweep_rows4(const R (*AR)[16 * NS], R* t, R* tsave, const R* nf0, const R* oh) {
  static_assert(CC % 4 == 0, "row groups of four");
  if constexpr (NS == 1) dmw::pgs_rows4<CC % 16>(t[0], tsave[0], nf0[0], &AR[0][CC], &oh[CC % 16]);
  else dmw::pgs_rows4_2<CC % 16>(t[CC / 16], tsave[CC / 16], t[1 - CC / 16], nf0[CC / 16], &AR[CC / 16][CC], &AR[1 - CC / 16][CC], &oh[CC % 16]);
}
template <int C, int NS, class R>
struct SlotSweep {
  static DM_DEV void run(const R (*AR)[16 * NS], R* t, R* tsave, const R* nf0, const R* oh, int nmax) {
    if constexpr (C < 16 * NS) {
      slot_sweep_rows4<C, NS, R>(AR, t, tsave, nf0, oh);
      if (C + 4 < nmax) SlotSweep<C + 4, NS, R>::run(AR, t, tsave, nf0, oh, nmax);
    }
  }
};
// sum over rows of f_r Y_r, row by row in order (the order of the one-env kernel's assembly): ws[d] += (f Y)_(row CC)[d]
template <int CC, class R>
DM_DEV void slot_assemble_row(R* ws, const R (*fy)[NV], R one) {
#pragma unroll
  for (int d = 0; d + 8 <= NV; d += 8) dmw::row_add8<CC % 16>(&ws[d], &fy[CC / 16][d], one);
#pragma unroll
  for (int d = NV / 8 * 8; d < NV; d++) dmw::row_fmac_old<CC % 16>(ws[d], fy[CC / 16][d], one);
}
template <int C, int NS, class R>
struct SlotAssemble {
  static DM_DEV void run(R* ws, const R (*fy)[NV], R one, int nmax) {
    if constexpr (C < 16 * NS) {
      slot_assemble_row<C, R>(ws, fy, one); slot_assemble_row<C + 1, R>(ws, fy, one); slot_assemble_row<C + 2, R>(ws, fy, one); slot_assemble_row<C + 3, R>(ws, fy, one);
      if (C + 4 < nmax) SlotAssemble<C + 4, NS, R>::run(ws, fy, one, nmax);
    }
  }
};

// ---- the partial third row set (rows 32 .. 39; slot_constraint<3>) ------------------------------------------------------------------
// A standing humanoid holds 8 foot corners x 4 pyramid edges = 32 contact rows plus one to five joint limits: 33 .. 37 rows, ~5 % of a learning
// run's env-steps (profiles/r04_ab_kernel_variants.md), which rounds 3-4 re-stepped one env per wave at ~785 k cycles each.  A third full row set
// does not fit a lane's registers (3 x 34 doubles of Y and 3 x 48 of A); what fits is the SYMMETRIC part of it: row 32 + j lives on lane 2 j of the
// slot (j < 8) and only the two blocks of A that touch the surplus rows are kept —
//   U[k][j] = A[row (sl, k)][32 + j]   (every lane, both full sets: the column block; scaled by -1 / A_rr like the rest of the lane's rows)
//   B[j]    = A[32 + sl / 2][32 + j]   (the 8 x 8 corner, rows on the even lanes)
// — the row block A[32 + j][c < 32] is the column block transposed, so a surplus row's residual is FORMED once per sweep from the forces
// (sum over the 16 lanes of U[k][j] f_k: a reduce-scatter whose natural landing lanes are 2 j, 2 j + 1 — hence the ownership) instead of being carried
// through the 32 rows before it.  The 32 rows themselves run the two-set code unchanged: an environment with <= 32 rows that shares its wave
// with a heavier one gets bit-identical results (its surplus terms are exact zeros).
// sum over the 16 lanes of the own DPP row of eight values at once; p[j]'s total lands on lanes 2 j and 2 j + 1 (returned)
template <class R>
DM_DEV R row_reduce8(const R* p, int sl) {
  const bool b3 = (sl & 8) != 0, b2 = (sl & 4) != 0, b1 = (sl & 2) != 0;
  R q[4], r2[2];
#pragma unroll
  for (int i = 0; i < 4; i++) { const R keep = b3 ? p[4 + i] : p[i], send = b3 ? p[i] : p[4 + i]; q[i] = keep + dmw::perm_row_mirror(send); }       // partner 15 - sl keeps the other half
#pragma unroll
  for (int i = 0; i < 2; i++) { const R keep = b2 ? q[2 + i] : q[i], send = b2 ? q[i] : q[2 + i]; r2[i] = keep + dmw::perm_half_mirror(send); }      // partner (sl ^ 7): same bit 3, other bit 2
  const R keep = b1 ? r2[1] : r2[0], send = b1 ? r2[0] : r2[1];
  R v = keep + dmw::perm_xor2(send);                                                                                                                // partner sl ^ 2
  v += dmw::perm_xor1(v);                                                                                                                           // sl ^ 1 holds the same index
  return v;
}
// column block / corner entries of A for surplus row J: U[k][J] = Y_(own row, set k) . Y_(32 + J), B[J] = Y_(own surplus row) . Y_(32 + J)
template <int J, class R>
DM_DEV void slot_ext_col(R (*U)[SLOT_EXTROWS], R* B, const R (*y)[NV], const R* y3) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const R* mine = k < 2 ? y[k] : y3;
    R acc0 = 0, acc1 = 0;
#pragma unroll
    for (int d = 0; d + 8 <= NV; d += 8) dmw::row_fmac8<2 * J>(acc0, acc1, &y3[d], &mine[d]);
#pragma unroll
    for (int d = NV / 8 * 8; d < NV; d += 2) { dmw::row_fmac_old<2 * J>(acc0, y3[d], mine[d]); dmw::row_fmac_old<2 * J>(acc1, y3[d + 1], mine[d + 1]); }
    if (k < 2) U[k][J] = acc0 + acc1; else B[J] = acc0 + acc1;
  }
}
template <int J, class R>
struct SlotExtCols {
  static DM_DEV void run(R (*U)[SLOT_EXTROWS], R* B, const R (*y)[NV], const R* y3, int next) {
    if constexpr (J < SLOT_EXTROWS) {
      slot_ext_col<J, R>(U, B, y, y3); slot_ext_col<J + 1, R>(U, B, y, y3); slot_ext_col<J + 2, R>(U, B, y, y3); slot_ext_col<J + 3, R>(U, B, y, y3);
      if (J + 4 < next) SlotExtCols<J + 4, R>::run(U, B, y, y3, next);
    }
  }
};
// warm start: the surplus columns scaled in place (rows by -1 / A_rr), t_k += A_s[k][32 + J] f3_J
template <int J, class R>
struct SlotExtWarm {
  static DM_DEV void run(R (*U)[SLOT_EXTROWS], R* B, R* t, R f3, const R* ndinv, R ndinv3, int next) {
    if constexpr (J < SLOT_EXTROWS) {
#pragma unroll
      for (int j = J; j < J + 4; j++) { U[0][j] *= ndinv[0]; U[1][j] *= ndinv[1]; B[j] *= ndinv3; }
      dmw::row_fmac_old<2 * J>(t[0], f3, U[0][J]); dmw::row_fmac_old<2 * J>(t[1], f3, U[1][J]);
      dmw::row_fmac_old<2 * J + 2>(t[0], f3, U[0][J + 1]); dmw::row_fmac_old<2 * J + 2>(t[1], f3, U[1][J + 1]);
      dmw::row_fmac_old<2 * J + 4>(t[0], f3, U[0][J + 2]); dmw::row_fmac_old<2 * J + 4>(t[1], f3, U[1][J + 2]);
      dmw::row_fmac_old<2 * J + 6>(t[0], f3, U[0][J + 3]); dmw::row_fmac_old<2 * J + 6>(t[1], f3, U[1][J + 3]);
      if (J + 4 < next) SlotExtWarm<J + 4, R>::run(U, B, t, f3, ndinv, ndinv3, next);
    }
  }
};
// scaled residual of the own surplus row from the forces:  tb3 + (-1 / A_rr) sum_(c < 32) A[32 + j][c] f_c + sum_i B_s[i] f3_i
// (U holds A_s[c][32 + j] = -A[c][32 + j] / A_cc: times g_c = -A_cc f_c it is the entry of A times the force again)
template <class R>
DM_DEV R slot_ext_residual(const R (*U)[SLOT_EXTROWS], const R* B, const R* f, const R* diag, R f3, R tb3, R ndinv3, int sl) {
  const R g0 = -(diag[0] * f[0]), g1 = -(diag[1] * f[1]);
  R p[SLOT_EXTROWS > 0 ? SLOT_EXTROWS : 1];
#pragma unroll
  for (int j = 0; j < SLOT_EXTROWS; j++) p[j] = U[0][j] * g0 + U[1][j] * g1;
  R t3 = tb3 + ndinv3 * row_reduce8(p, sl);
  dmw::dpp_settle();
  dmw::row_fmac_old<0>(t3, f3, B[0]); dmw::row_fmac_old<2>(t3, f3, B[1]); dmw::row_fmac_old<4>(t3, f3, B[2]); dmw::row_fmac_old<6>(t3, f3, B[3]);
  dmw::row_fmac_old<8>(t3, f3, B[4]); dmw::row_fmac_old<10>(t3, f3, B[5]); dmw::row_fmac_old<12>(t3, f3, B[6]); dmw::row_fmac_old<14>(t3, f3, B[7]);
  return t3;
}
// the surplus rows of one Gauss-Seidel sweep, in order, after the 32 rows before them: their steps also move the residuals of both full sets
template <int J, class R>
struct SlotExtSweep {
  static DM_DEV void run(const R (*U)[SLOT_EXTROWS], const R* B, R& t3, R& tsave3, R* t, R nf03, const R* oh, int next) {
    if constexpr (J < SLOT_EXTROWS) {
      dmw::pgs_row3<2 * J>(t3, tsave3, t[0], t[1], nf03, B[J], U[0][J], U[1][J], oh[2 * J]);
      dmw::pgs_row3<2 * J + 2>(t3, tsave3, t[0], t[1], nf03, B[J + 1], U[0][J + 1], U[1][J + 1], oh[2 * J + 2]);
      dmw::pgs_row3<2 * J + 4>(t3, tsave3, t[0], t[1], nf03, B[J + 2], U[0][J + 2], U[1][J + 2], oh[2 * J + 4]);
      dmw::pgs_row3<2 * J + 6>(t3, tsave3, t[0], t[1], nf03, B[J + 3], U[0][J + 3], U[1][J + 3], oh[2 * J + 6]);
      if (J + 4 < next) SlotExtSweep<J + 4, R>::run(U, B, t3, tsave3, t, nf03, oh, next);
    }
  }
};
template <int I, class R>
DM_DEV void slot_assemble_lane(R* ws, const R* fy, R one) {          // ws[d] += (lane I of the row's fy[d])
#pragma unroll
  for (int d = 0; d + 8 <= NV; d += 8) dmw::row_add8<I>(&ws[d], &fy[d], one);
#pragma unroll
  for (int d = NV / 8 * 8; d < NV; d++) dmw::row_fmac_old<I>(ws[d], fy[d], one);
}
template <int J, class R>
struct SlotExtAssemble {
  static DM_DEV void run(R* ws, const R* fy3, R one, int next) {
    if constexpr (J < SLOT_EXTROWS) {
      slot_assemble_lane<2 * J, R>(ws, fy3, one); slot_assemble_lane<2 * J + 2, R>(ws, fy3, one); slot_assemble_lane<2 * J + 4, R>(ws, fy3, one); slot_assemble_lane<2 * J + 6, R>(ws, fy3, one);
      if (J + 4 < next) SlotExtAssemble<J + 4, R>::run(ws, fy3, one, next);
    }
  }
};

// Jacobian wrench, distance and regularisation constants of constraint row r of the slot (what the row's lane needs before the dof loop)
template <class R>
DM_DEV void slot_row_setup(const DevModel<R>& M, const SlotShared<R>& s, int r, bool active, RowAcc<R>& ra, R& pos, R& margin, R& dA, R& rscale) {
  const auto& W = s.r1.rw;
  const int code = active ? W.rowi[r] : 0;
  const int type = code & 0xff;
  R w[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long mplus = 0, mminus = 0;
  pos = 0; margin = 0; dA = 0; rscale = 1;
  R w7 = 0;
  if (type == ROW_LIMIT) {
    const int d = (code >> 8) & 0xff;
    w7 = ((code >> 16) & 1) ? R(1) : R(-1);
    pos = W.rowv[r]; dA = M.dof_invw[d]; mplus = 1ull << d;
  } else if (type == ROW_CONTACT) {
    const int ci = (code >> 8) & 0xff, q = (code >> 16) & 0xf, cdim = (code >> 20) & 0xf;
    const R* c = W.con[ci];
    const int cw = W.coni[ci];
    const R* fm = W.frm[(cw >> 8) & 0xff];
    const auto& rec = M.pair_rec[cw & 0xff];
    const R cmu = rec.mu, ctran = rec.tran;
    const int meta = rec.meta;
    R dir[3] = {fm[0], fm[1], fm[2]};
    if (cdim != 1) {
      const R n[3] = {fm[0], fm[1], fm[2]}, t1[3] = {fm[3], fm[4], fm[5]};
      R t2[3];
      cross3(t2, n, t1);
      const R sg = (q & 1) ? -cmu : cmu;
      const R* tg = (q >> 1) ? t2 : t1;
      dir[0] += sg * tg[0]; dir[1] += sg * tg[1]; dir[2] += sg * tg[2];
    }
    cross3(w, c, dir);
    w[3] = dir[0]; w[4] = dir[1]; w[5] = dir[2];
    pos = c[3]; margin = rec.margin;
    dA = cdim == 1 ? ctran : ctran + cmu * cmu * ctran;
    rscale = cdim == 1 ? R(1) : 2 * cmu * cmu;
    mminus = TOPO.chain[meta & 0xff]; mplus = TOPO.chain[(meta >> 8) & 0xff];
  }
  for (int r6 = 0; r6 < 6; r6++) ra.w[r6] = w[r6];
  ra.w7 = w7; ra.w8 = 0; ra.vel = 0; ra.jws = 0;
  ra.plus_lo = (unsigned)mplus; ra.plus_hi = (unsigned)(mplus >> 32); ra.minus_lo = (unsigned)mminus; ra.minus_hi = (unsigned)(mminus >> 32);
}

// ---- constraint solve of the slot's environment; lane sl owns rows sl + 16 k, k < NS (limits first, then contacts in list order)
//      [MJ mj_fwdAcceleration, mj_projectConstraint, mj_fwdConstraint (warmstart, mj_solPGS)] -----------------------------------------
// Same mathematics as env_kernel.h stage_constraint (half-solved vectors Y = D^-1/2 L^-T J^T, A = Y Y^T + R, scaled-residual PGS,
// qacc = L^-1 D^-1/2 (z + sum f Y)); what differs is the plumbing: rows of Y are exchanged by DPP row broadcasts instead of LDS, the
// smooth force's half solve z is carried by EVERY lane beside its rows (same factor loads), and the final L^-1 pass runs on a
// register vector that all 16 lanes of the slot hold.  `nefc` is the slot's row count, `nmax` the wave's largest (row loops run to it;
// a slot's absent rows are exact zeros).  The sweeps of a converged environment are frozen, so its result does not depend on its partners.
// NS = 3: the two full sets plus the partial third one (rows 32 .. 39 on the even lanes; see "the partial third row set" above).
template <class R, int NS, bool PROF = false>
DM_DEV void slot_constraint(const DevModel<R>& M, SlotShared<R>& s, int sl_in, int lane_in, int nefc, int nmax, int& ovf, const DebugOut* dbg, long long* prof = 0) {
  const int sl = DM_SLOT_LANE_AT(4, sl_in), lane = dmw::launder(lane_in);
  long long pt0 = 0, pt1 = 0, pt_enter = 0;
  if (PROF) { pt0 = dmw::clk(); pt_enter = pt0; }
#define SLOT_STAMP(k) if constexpr (NS == 1) { DM_MARK("slot_constraint_ns1_" #k); } else if constexpr (NS == 2) { DM_MARK("slot_constraint_ns2_" #k); } else { DM_MARK("slot_constraint_ns3_" #k); } if (PROF) { pt1 = dmw::clk(); prof[k] += pt1 - pt0; pt0 = pt1; }
  constexpr bool EXT = NS == 3;              // the partial third row set is there
  constexpr int NF = EXT ? 2 : NS;           // full row sets
  constexpr int NC = 16 * NF;
  constexpr int NX = EXT ? SLOT_EXTROWS : 1;
  if constexpr (NS == 1) { DM_MARK("slot_constraint_ns1_rows"); } else if constexpr (NS == 2) { DM_MARK("slot_constraint_ns2_rows"); } else { DM_MARK("slot_constraint_ns3_rows"); }
  R y[NF + 1][NV];                     // rows' Jacobians -> Y;  y[NF] = tau -> z (identical in every lane of the slot)
  RowAcc<R> ra[NF];
  bool active[NF];
  R pos[NF], margin[NF], dA[NF], rscale[NF];
#pragma unroll
  for (int k = 0; k < NF; k++) {
    const int r = sl + 16 * k;
    active[k] = r < nefc;
    slot_row_setup(M, s, r, active[k], ra[k], pos[k], margin[k], dA[k], rscale[k]);
  }
  {
    R cur[8];
    dmw::reload_fence();
    slot_load_dof_operands<0>(cur, s);
    SlotRowStep<0, NF, R>::run(y, ra, s, cur);
  }
#pragma unroll
  for (int d = 0; d < NV; d++) y[NF][d] = s.tau[d];
  if (dbg) {
#pragma unroll
    for (int k = 0; k < NF; k++) if (active[k]) {
      double* o = dbg->out + (34 * 34 + 34 * 3 + 42 + 3) + (sl + 16 * k) * (34 + 6);
#pragma unroll
      for (int d = 0; d < NV; d++) o[d] = (double)y[k][d];
    }
  }
  SLOT_STAMP(8)
  R Rr[NF], aref[NF], f[NF];
#pragma unroll
  for (int k = 0; k < NF; k++) {
    const R imp = impedance(M, pos[k] - margin[k]);
    Rr[k] = fmax(R(DM_MINVAL), (1 - imp) * dA[k] / imp);
    if (rscale[k] != R(1)) Rr[k] = fmax(R(DM_MINVAL), rscale[k] * Rr[k]);
    aref[k] = -M.B * ra[k].vel - M.K * imp * (pos[k] - margin[k]);
    const R jar = ra[k].jws - aref[k];
    f[k] = (active[k] && jar < 0) ? -jar / Rr[k] : R(0);
  }
  // half solves: rows J^T -> Y, tau -> z
  slot_solve_LT<NF + 1>(y, s.r2.qLD);
#pragma unroll
  for (int d = 0; d < NV; d++) {
    const R sc = s.dsq[d];
#pragma unroll
    for (int k = 0; k <= NF; k++) { y[k][d] *= sc; dmw::pin_value(y[k][d]); }
  }
  if (dbg) {      // debug dump only: qacc_smooth = L^-1 D^-1/2 z next to a constrained solve
    R x[NV];
#pragma unroll
    for (int d = 0; d < NV; d++) x[d] = y[NF][d] * s.dsq[d];
    solve_L(x, s.r2.qLD);
    if (sl == 0) {
#pragma unroll
      for (int d = 0; d < NV; d++) dbg->out[34 * 34 + 34 + d] = (double)x[d];
    }
  }
  R bb[NF];
#pragma unroll
  for (int k = 0; k < NF; k++) {
    R acc0 = 0, acc1 = 0;
#pragma unroll
    for (int d = 0; d < NV; d++) { if (d & 1) acc1 += y[k][d] * y[NF][d]; else acc0 += y[k][d] * y[NF][d]; }
    bb[k] = active[k] ? (acc0 + acc1) - aref[k] : R(0);
  }
  // ---- the surplus rows (32 + sl / 2 on the even lanes): their own row build and half solve, once z is out of the registers ---------------
  R y3[EXT ? NV : 1];
  bool active3 = false;
  R Rr3 = 0, f3 = 0, bb3 = 0;
  if constexpr (EXT) {
    dmw::sync();                                                  // (every lane has read tau)
    if (sl == 0) {
#pragma unroll
      for (int d = 0; d < NV; d++) s.tau[d] = y[NF][d];           // z: identical in every lane of the slot; read back at the force assembly
    }
    const int r3 = 2 * SW + (sl >> 1);
    active3 = (sl & 1) == 0 && r3 < nefc;
    RowAcc<R> ra3[1];
    R pos3, margin3, dA3, rscale3;
    slot_row_setup(M, s, r3, active3, ra3[0], pos3, margin3, dA3, rscale3);
    R yy[1][NV];
    {
      R cur[8];
      dmw::reload_fence();
      slot_load_dof_operands<0>(cur, s);
      SlotRowStep<0, 1, R>::run(yy, ra3, s, cur);
    }
    const R imp = impedance(M, pos3 - margin3);
    Rr3 = fmax(R(DM_MINVAL), (1 - imp) * dA3 / imp);
    if (rscale3 != R(1)) Rr3 = fmax(R(DM_MINVAL), rscale3 * Rr3);
    const R aref3 = -M.B * ra3[0].vel - M.K * imp * (pos3 - margin3);
    const R jar = ra3[0].jws - aref3;
    f3 = (active3 && jar < 0) ? -jar / Rr3 : R(0);
    slot_solve_LT<1>(yy, s.r2.qLD);
    dmw::sync();                                                  // (z went through LDS)
    R acc0 = 0, acc1 = 0;
#pragma unroll
    for (int d = 0; d < NV; d++) {
      y3[d] = yy[0][d] * s.dsq[d]; dmw::pin_value(y3[d]);
      if (d & 1) acc1 += y3[d] * s.tau[d]; else acc0 += y3[d] * s.tau[d];
    }
    bb3 = active3 ? (acc0 + acc1) - aref3 : R(0);
  }
  // (the surplus blocks are built BEFORE the 32 x 32 part: Y of all three sets is in registers now and none of A yet; afterwards the surplus rows' Y
  //  goes to LDS and the 64 entries of A per lane take its place)
  R U[EXT ? 2 : 1][NX], B3[NX];      // surplus columns of the lane's two rows; the corner row of the lane's surplus row
  R diag3 = 1;
  if constexpr (EXT) {
#pragma unroll
    for (int j = 0; j < NX; j++) { U[0][j] = 0; U[1][j] = 0; B3[j] = 0; dmw::pin_value(U[0][j]); dmw::pin_value(U[1][j]); dmw::pin_value(B3[j]); }
    dmw::dpp_settle();
    SlotExtCols<0, R>::run(U, B3, y, y3, nmax - 2 * SW);
    R dg = 1;
#pragma unroll
    for (int j = 0; j < NX; j++) { if (sl == 2 * j) { B3[j] += active3 ? Rr3 : R(0); dg = B3[j]; } }
    diag3 = active3 ? dg : R(1);
    R* y3lds = &s.qd.o.qacc[0];
    if ((sl & 1) == 0) {
#pragma unroll
      for (int d = 0; d < NV; d++) y3lds[d * SLOT_EXTROWS + (sl >> 1)] = y3[d];
    }
  }
  SLOT_STAMP(9)
  // ---- A = Y Y^T + diag(R) --------------------------------------------------------------------------------------------------------
  R AR[NF][NC];
#pragma unroll
  for (int k = 0; k < NF; k++)
#pragma unroll
    for (int c = 0; c < NC; c++) { AR[k][c] = 0; dmw::pin_value(AR[k][c]); }
  dmw::dpp_settle();
  SlotACols<0, NF, R>::run(AR, y, nmax);
  R diag[NF];
#pragma unroll
  for (int k = 0; k < NF; k++) {
    R dg = 1;
#pragma unroll
    for (int c = 0; c < 16; c++) { if (sl == c) { AR[k][16 * k + c] += active[k] ? Rr[k] : R(0); dg = AR[k][16 * k + c]; } }
    diag[k] = active[k] ? dg : R(1);
  }
  SLOT_STAMP(10)
  // Two row sets: from here to the end of the sweeps the hot data are the 64 entries of A each lane holds (128 registers, read by every
  // row of every sweep); the three half-solved vectors (204 registers) are not touched again before the force assembly.  Left to the
  // register allocator, a third of A sits in accumulation registers and is copied out operand by operand inside the sweep loop (74 of a
  // trip's 277 instructions).  So the vectors are moved out of the architectural file explicitly, once: z — identical in every lane of the
  // slot — into the slot's `tau` (dead since it was read above), the rows' Y into accumulation registers; both come back after the sweeps.
  // (Three: z went to `tau` before the surplus rows were built; their Y goes to the slot's qd + cdof regions in LDS — dead since the row builds.)
  typedef decltype(dmw::park(R(0))) ParkedR;
  ParkedR ypark[NF == 2 ? NF : 1][NF == 2 ? NV : 1];
  if constexpr (NF == 2) {
    if constexpr (!EXT) {
      if (sl == 0) {
#pragma unroll
        for (int d = 0; d < NV; d++) s.tau[d] = y[NF][d];
      }
    }
#pragma unroll
    for (int k = 0; k < NF; k++)
#pragma unroll
      for (int d = 0; d < NV; d++) ypark[k][d] = dmw::park(y[k][d]);
    // (a fresh definition of every entry of A now that registers are free: the values the build left in accumulation registers would
    //  otherwise stay there, reloaded at each use)
#pragma unroll
    for (int k = 0; k < NF; k++)
#pragma unroll
      for (int c = 0; c < NC; c++) dmw::pin_value(AR[k][c]);
    if constexpr (EXT) {
#pragma unroll
      for (int j = 0; j < NX; j++) { dmw::pin_value(U[0][j]); dmw::pin_value(U[1][j]); dmw::pin_value(B3[j]); }
    }
  }
  // ---- warm start: keep f(qacc_warmstart) only if its dual cost beats f = 0 ---------------------------------------------------------
  R ndinv[NF], tb[NF], t[NF];
#pragma unroll
  for (int k = 0; k < NF; k++) { ndinv[k] = active[k] ? R(-1) / diag[k] : R(0); dmw::pin_value(ndinv[k]); tb[k] = bb[k] * ndinv[k]; t[k] = tb[k]; }
  SlotWarm<0, NF, R>::run(AR, t, f, ndinv, nmax);
  R ndinv3 = 0, tb3 = 0;
  if constexpr (EXT) {
    ndinv3 = active3 ? R(-1) / diag3 : R(0); dmw::pin_value(ndinv3);
    tb3 = bb3 * ndinv3;
    dmw::dpp_settle();
    SlotExtWarm<0, R>::run(U, B3, t, f3, ndinv, ndinv3, nmax - 2 * SW);
  }
  {
    R c = 0;
#pragma unroll
    for (int k = 0; k < NF; k++) { const R res = -t[k] * diag[k]; c += active[k] ? f[k] * (R(0.5) * (res - bb[k]) + bb[k]) : R(0); }
    if constexpr (EXT) {
      const R t3 = slot_ext_residual(U, B3, f, diag, f3, tb3, ndinv3, sl);
      const R res = -t3 * diag3;
      c += active3 ? f3 * (R(0.5) * (res - bb3) + bb3) : R(0);
    }
    const R cost = dmw::sum16(c);
    if (cost > 0) {
#pragma unroll
      for (int k = 0; k < NF; k++) { f[k] = 0; t[k] = tb[k]; }
      f3 = 0;
    }
  }
  SLOT_STAMP(11)
  // ---- projected Gauss-Seidel: rows in order, every environment of the wave in step; a converged environment is frozen ----------------
  const int maxiter = dmw::uniform(M.iterations);
  R pgs_scale = M.pgs_scale, pgs_tol = M.tolerance, pgs_detect = M.pgs_detect;
  dmw::pin_value(pgs_scale); dmw::pin_value(pgs_tol); dmw::pin_value(pgs_detect);
  int iter = 0;
  bool frozen = nefc == 0 || maxiter <= 0, anybad = false;
  if (frozen) {
#pragma unroll
    for (int k = 0; k < NF; k++) t[k] = 0;
  }
  R oh[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { oh[i] = sl == i ? R(1) : R(0); dmw::pin_value(oh[i]); }
  // The loop's state is the NEGATED force nf = -f (what a row's clamp compares against: no negation per sweep) and a lane's share of the cost CHANGE of a
  // sweep (negative = improvement; the termination test compares its sum against -tolerance: no negation either).  Sign flips and the order of the
  // summands are exact in floating point, so every force, every residual and every decision is bit-identical to the form with f and an "improvement".
  R nf[NF], nf3 = -f3;
#pragma unroll
  for (int k = 0; k < NF; k++) nf[k] = -f[k];
  R pgs_ntol = -pgs_tol;
  dmw::pin_value(pgs_ntol);
  // one sweep: negated forces nf, scaled residuals t; returns this lane's share of the cost change and whether [MJ costChange] would object
  auto sweep = [&](R& chg, bool& bad) {
    const int nm = dmw::launder_uniform(nmax);
    R tsave[NF];
#pragma unroll
    for (int k = 0; k < NF; k++) tsave[k] = 0;
    SlotSweep<0, NF, R>::run(AR, t, tsave, nf, oh, nm);
    bad = false;
#pragma unroll
    for (int k = 0; k < NF; k++) {
      const R delta = dmw::max_raw(nf[k], tsave[k]);
      const R change = (delta * diag[k]) * (R(0.5) * delta - tsave[k]);
      chg = k == 0 ? change : chg + change;      // (0 + c == c bit for bit except for the sign of a zero, which no comparison downstream sees)
      nf[k] -= delta; bad = bad || (change > pgs_detect);
    }
    if constexpr (EXT) {
      // the surplus rows: residual formed from the forces as they stand after the 32 rows before them (a frozen environment: no step)
      R fnow[NF];
#pragma unroll
      for (int k = 0; k < NF; k++) fnow[k] = -nf[k];
      const R f3now = -nf3;
      R t3 = slot_ext_residual(U, B3, fnow, diag, f3now, tb3, ndinv3, sl);
      t3 = frozen ? R(0) : t3;
      R tsave3 = 0;
      SlotExtSweep<0, R>::run(U, B3, t3, tsave3, t, nf3, oh, nm - 2 * SW);
      const R delta = dmw::max_raw(nf3, tsave3);
      const R change = (delta * diag3) * (R(0.5) * delta - tsave3);
      nf3 -= delta; chg += change; bad = bad || (change > pgs_detect);
    }
  };
  // The termination test of sweep k (a DPP reduction) is independent of the rows of sweep k + 1: sweep k + 1 is issued speculatively
  // beside it and dropped — forces restored, rows frozen — for the environments that turn out to have converged at sweep k.
  // Freezing: the residuals are set to zero; a row's step is max(-f, t) = max(-f, 0) = 0 whatever force it holds (f >= 0), so a finished
  // environment's forces no longer move while its partners go on.
  if (dmw::ballot(!frozen) != 0ull) {
    R chg; bool bad;
    sweep(chg, bad);
    if (!frozen) { iter = 1; anybad = bad; }
    bool more = true;
    while (more) {
      R nfprev[NF], nfprev3 = nf3;
#pragma unroll
      for (int k = 0; k < NF; k++) nfprev[k] = nf[k];
      const R schg = dmw::sum16(chg) * pgs_scale;                  // of the last accepted sweep: -improvement
      R chg2; bool bad2;
      sweep(chg2, bad2);                                          // speculative
      // (selects, not branches: `frozen` differs from slot to slot, and a divergent region costs more than the handful of moves it guards)
      const bool act = !frozen;
      const bool conv = (int)act & ((int)(schg > pgs_ntol) | (int)(iter >= maxiter));     // (bit operations: no short-circuit region)
#pragma unroll
      for (int k = 0; k < NF; k++) {
        nf[k] = conv ? nfprev[k] : nf[k];
        t[k] = conv ? R(0) : t[k];
      }
      if constexpr (EXT) nf3 = conv ? nfprev3 : nf3;
      const bool go = (int)act & (int)!conv;
      iter += go ? 1 : 0;
      anybad = (int)anybad | ((int)go & (int)bad2);
      frozen = (int)frozen | (int)conv;
      chg = chg2;
      more = dmw::ballot(!frozen) != 0ull;
      if (PROF) prof[6] += 1;
    }
  }
#pragma unroll
  for (int k = 0; k < NF; k++) f[k] = -nf[k];
  f3 = -nf3;
  if (PROF && NS >= 2) { prof[22] += dmw::clk() - pt0; }
  if (PROF && NS == 3) { prof[27] += dmw::clk() - pt0; }
  SLOT_STAMP(12)
  if (PROF) { prof[14] += nmax; prof[15] += 1; }
  if (dmw::row_ballot(anybad, lane) != 0u) ovf |= 16 | 64;        // [MJ costChange] would have rejected a step: the one-env kernel's guarded replay decides
  if (dbg) {
#pragma unroll
    for (int k = 0; k < NF; k++) if (active[k]) {
      double* o = dbg->out + (34 * 34 + 34 * 3 + 42 + 3) + (sl + 16 * k) * (34 + 6) + 34;
      o[0] = (double)pos[k]; o[1] = (double)margin[k]; o[2] = (double)Rr[k]; o[3] = (double)aref[k]; o[4] = (double)bb[k]; o[5] = (double)f[k];
    }
  }
  // ---- qacc = L^-1 D^-1/2 (z + sum_r f_r Y_r) -----------------------------------------------------------------------------------------
  R fy3[EXT ? NV : 1];
  if constexpr (EXT) {            // the surplus rows' Y back from LDS (own lane's words), times their forces
    const R* y3lds = &s.qd.o.qacc[0];
    const R f3a = active3 ? f3 : R(0);
#pragma unroll
    for (int d = 0; d < NV; d++) { fy3[d] = ((sl & 1) == 0 ? y3lds[d * SLOT_EXTROWS + (sl >> 1)] : R(0)) * f3a; dmw::pin_value(fy3[d]); }
  }
  if constexpr (NF == 2) {
    dmw::sync();                                   // (z went through LDS)
#pragma unroll
    for (int d = 0; d < NV; d++) y[NF][d] = s.tau[d];
  }
#pragma unroll
  for (int k = 0; k < NF; k++) {
    const R fk = active[k] ? f[k] : R(0);
#pragma unroll
    for (int d = 0; d < NV; d++) { if constexpr (NF == 2) y[k][d] = dmw::unpark(ypark[k][d]); y[k][d] *= fk; dmw::pin_value(y[k][d]); }
  }
  R ws[NV];
#pragma unroll
  for (int d = 0; d < NV; d++) { ws[d] = 0; dmw::pin_value(ws[d]); }
  R one = 1;
  dmw::pin_value(one);
  dmw::dpp_settle();
  SlotAssemble<0, NF, R>::run(ws, y, one, nmax);
  if constexpr (EXT) SlotExtAssemble<0, R>::run(ws, fy3, one, nmax - 2 * SW);
#pragma unroll
  for (int d = 0; d < NV; d++) ws[d] = (ws[d] + y[NF][d]) * s.dsq[d];
  solve_L(ws, s.r2.qLD);
  if (sl == 0) {
#pragma unroll
    for (int d = 0; d < NV; d++) s.qd.o.qacc[d] = ws[d];
    s.solver_iter = iter;
  }
  dmw::sync();
  SLOT_STAMP(13)
  if (PROF && NS >= 2) prof[24] += dmw::clk() - pt_enter;
  if (PROF && NS == 3) prof[26] += dmw::clk() - pt_enter;
#undef SLOT_STAMP
}

// ---- the collision stage as an internal function (round 6) --------------------------------------------------------------------------------------------------
// With inter-procedural register allocation a call costs no register save / restore (slot_step.h DM_CALL_SLOT), so a stage compiled as a function of its own gets a
// register assignment that the 30 000 other instructions of the step cannot perturb — and ONE copy of its code serves every kernel of the unit.  Measured per stage
// (profiles/r06_ab_kernel_variants.md section 7): the collision stage +0.6 % alone, +1.7 % with its constants requested up front; the constraint stage -1.3 %;
// mass matrix, bias, kinematics nothing.  The pointer to the caller's `ovf` is the frame pointer that keeps the call from being a tail call.
#if !defined(DM_WAVE_TESTBENCH)
template <class R, int MAXR>
static __device__ __noinline__ int slot_rows_call(const DevModel<R>* M, SlotShared<R>* s, int sl, int lane, int* ovf) {
  int o = *ovf;
  const int n = slot_rows<R, false, MAXR>(*dmw::in_constant(M), *dmw::in_lds(s), sl, lane, o, (long long*)0);
  *ovf = o;
  return n;
}
#endif

// ---- no rows anywhere in the wave: qacc = L^-1 D^-1/2 (D^-1/2 L^-T tau), the constrained formula with an empty sum (so that an
// environment's result does not depend on whether a partner has rows).  Every lane of the slot carries the whole vector. -----------
template <class R>
DM_DEV void slot_smooth_solve(SlotShared<R>& s, int sl, const DebugOut* dbg) {
  R x[1][NV];
#pragma unroll
  for (int d = 0; d < NV; d++) x[0][d] = s.tau[d];
  slot_solve_LT<1>(x, s.r2.qLD);
#pragma unroll
  for (int d = 0; d < NV; d++) { x[0][d] *= s.dsq[d]; dmw::pin_value(x[0][d]); }
#pragma unroll
  for (int d = 0; d < NV; d++) x[0][d] = (R(0) + x[0][d]) * s.dsq[d];
  solve_L(x[0], s.r2.qLD);
  if (sl == 0) {
#pragma unroll
    for (int d = 0; d < NV; d++) { s.qd.o.qacc[d] = x[0][d]; if (dbg) dbg->out[34 * 34 + 34 + d] = (double)x[0][d]; }
    s.solver_iter = 0;
  }
  dmw::sync();
}

// one forward-dynamics evaluation of the slot's environment: s.qpos, s.qvel, s.act, s.qws -> s.qacc; xip = body COM positions (body
// lanes); ovf: the environment exceeded a capacity of the packed path (sticky)
// CARRY (horizon launches): where every slot's `kin_ok` flag is set, the position stage's results are already in the slots — the 5-term
// reward ended the previous step with the kinematics pass of exactly the state this evaluation starts from — and the stage is skipped (xip is
// then left alone: only the 4th evaluation's is used).  The flag is read and cleared HERE, from LDS, so that no register carries the decision
// across the stages (a flag handed down through the RK loop cost the collision stage 49 more spill instructions: measured 1.5 % slower).
template <class R, bool PROF = false, bool CARRY = false, int MAXR = 2 * SW>
DM_DEV void slot_forward(const DevModel<R>& M, SlotShared<R>& s, const SlotTables& tb, int sl, int lane, const LaneTopo& lt, R* xip, int& ovf, const DebugOut* dbg, long long* prof = 0) {
  long long t0 = 0, t1 = 0;
  if (PROF) t0 = dmw::clk();
#define SLOT_FSTAMP(k) if (PROF) { t1 = dmw::clk(); prof[k] += t1 - t0; t0 = t1; }
  DM_MARK("slot_kinematics");
  bool skip_kin = false;
  if constexpr (CARRY) {
    skip_kin = dmw::ballot(s.kin_ok() != R(0)) == ~0ull;
    dmw::sync();
    if (sl == 0) s.kin_ok() = R(0);              // one use: the next evaluation starts from another state
  }
  if (!skip_kin) slot_kinematics(M, s, sl, lt, xip);
  SLOT_FSTAMP(0)
  if (dbg) { for (int e = sl; e < NV * NV; e += SW) dbg->out[e] = 0; dmw::sync(); }
  DM_MARK("slot_bias");
  slot_bias(M, s, tb, sl, lt);
  SLOT_FSTAMP(1)
  if (dbg) {
    for (int c = 0; c < DOF_PASSES; c++) { const int d = sl + SW * c; if (d < NV) dbg->out[34 * 34 + d] = (double)(-M.dof_damping[d] * s.qvel[d] + s.act[d] - s.tau[d]); }
  }
  DM_MARK("slot_mass_factor");
  slot_mass_matrix<R, PROF>(M, s, tb, sl, lt, dbg, prof);
  SLOT_FSTAMP(2)
  DM_MARK("slot_rows");
  int nefc = 0;
#if !defined(DM_WAVE_TESTBENCH)
  if (!PROF && (M.enable_contact || M.enable_limit)) nefc = slot_rows_call<R, MAXR>(&M, &s, sl, lane, &ovf);      // (the diagnostic build keeps the stage inline for its stamps)
  else
#endif
  if (M.enable_contact || M.enable_limit) nefc = slot_rows<R, PROF, MAXR>(M, s, sl, lane, ovf, prof);
  else { if (sl == 0) { s.nefc = 0; s.ncon = 0; } dmw::sync(); }
  SLOT_FSTAMP(3)
  DM_MARK("slot_constraint");
  const int nmax = rows_max(nefc);
  if (nmax == 0) slot_smooth_solve(s, sl, dbg);
#ifdef DM_FORCE_EXT      // test hook (testbench builds only): every constrained evaluation through the three-set code — results must not change
  else if (nmax > 0 && MAXR > 2 * SW) slot_constraint<R, MAXR > 2 * SW ? 3 : 2, PROF>(M, s, sl, lane, nefc, nmax > 2 * SW ? nmax : 2 * SW + 1, ovf, dbg, prof);
#endif
  else if (nmax <= 16) slot_constraint<R, 1, PROF>(M, s, sl, lane, nefc, nmax, ovf, dbg, prof);
  else if (nmax <= 2 * SW) { slot_constraint<R, 2, PROF>(M, s, sl, lane, nefc, nmax, ovf, dbg, prof); if (PROF) prof[7] += 1; }
  // 33 .. 40 rows somewhere in the wave: the three-set code — in the instantiations that carry it (MAXR = SLOT_MAXROWS; elsewhere slot_rows has flagged such
  // an environment and dropped its rows).  Two lessons of round 5 (profiles/r05_ab_kernel_variants.md): (1) behind a REAL call it kept its register appetite
  // to itself, but a call inside the step function stops the allocator from spilling into accumulation registers anywhere in it (slot_rows 19 -> 125 scratch
  // instructions, the two-set assembly 1 -> 239): -8 %; (2) INLINED, the one- and two-set paths keep their instruction sequences (99 % identical opcode
  // streams) and are 8 % slower all the same — the larger function perturbs the register ASSIGNMENT of the hot constraint stage (one-set evaluations 48.7 ->
  // 55.7 k cycles, two-set 108 -> 143 k).  So the three-set code lives in its own instantiation of the step function, which a wave calls only while one
  // of its environments is near the two-set capacity (slot_step.h slot_rollout); every other wave-step runs the lean instantiation.
  else if constexpr (MAXR > 2 * SW) { slot_constraint<R, 3, PROF>(M, s, sl, lane, nefc, nmax, ovf, dbg, prof); if (PROF) { prof[7] += 1; prof[25] += 1; } }
  SLOT_FSTAMP(4)
  DM_MARK("slot_forward_end");
#undef SLOT_FSTAMP
  if (dbg) {
    for (int c = 0; c < DOF_PASSES; c++) { const int d = sl + SW * c; if (d < NV) dbg->out[34 * 34 + 68 + d] = (double)s.qd.o.qacc[d]; }
    if (sl < NB - 1) for (int k = 0; k < 3; k++) dbg->out[34 * 34 + 102 + 3 * (sl + 1) + k] = (double)xip[k];
    if (sl == 0) { dbg->out[34 * 34 + 144] = s.nefc; dbg->out[34 * 34 + 145] = s.ncon; dbg->out[34 * 34 + 146] = s.solver_iter; }
  }
}

}  // namespace dm
