// model_host.h — host-side construction of the device constant block from the C-ABI model tables
// (dm_model_desc, include/dmenv.h).  Pure C++ (no HIP calls): shared by libdmenv.so and the wave testbench.
#pragma once

#include <cmath>
#include <cstring>
#include <string>

#include "dmenv.h"
#include "env_kernel.h"

namespace dm {

inline int mfail(std::string* err, int code, const char* msg) { if (err) *err = msg; return code; }

template <class R>
inline int build_dev_model(const dm_model_desc* d, DevModel<R>* hp, std::string* err) {
  if (!d || !hp) return mfail(err, DM_EINVAL, "dm_model_create: null argument");
  if (d->abi_version != DM_ABI_VERSION) return mfail(err, DM_EINVAL, "dm_model_create: ABI version mismatch");
  if (d->nbody != NB || d->njnt != NJ || d->nq != NQ || d->nv != NV || d->nu != NU || d->ngeom != NG)
    return mfail(err, DM_EUNSUPPORTED, "dm_model_create: the kernels are specialised to the DeepMimic humanoid tree (14 bodies, 29 joints, 34 dofs, 16 geoms)");
  if (d->npair < 0 || d->npair > MAXPAIR) return mfail(err, DM_EUNSUPPORTED, "dm_model_create: too many candidate contact pairs");
  const Topo T = make_topo();
  for (int b = 0; b < NB; b++)
    if (d->body_parentid[b] != T.body_parent[b] || d->body_dofnum[b] != T.body_dofnum[b])
      return mfail(err, DM_EUNSUPPORTED, "dm_model_create: body tree differs from the compiled-in humanoid topology");
  if (d->jnt_type[0] != 0 || d->jnt_bodyid[0] != 1) return mfail(err, DM_EUNSUPPORTED, "dm_model_create: joint 0 must be the root free joint");
  for (int j = 1; j < NJ; j++)
    if (d->jnt_type[j] != 3 || d->jnt_bodyid[j] != T.dof_body[j + 5]) return mfail(err, DM_EUNSUPPORTED, "dm_model_create: joint layout differs from the humanoid");
  for (int u = 0; u < NU; u++) if (d->actuator_dofid[u] != u + 6) return mfail(err, DM_EUNSUPPORTED, "dm_model_create: actuator u must drive hinge dof u+6");
  DevModel<R>& h = *hp;
  memset(&h, 0, sizeof h);
  for (int b = 0; b < NB; b++) {
    for (int k = 0; k < 3; k++) { h.body_pos[b][k] = d->body_pos[3 * b + k]; h.body_ipos[b][k] = d->body_ipos[3 * b + k]; }
    h.body_mass[b] = d->body_mass[b];
    const double* I = d->body_inertia + 9 * b;
    h.body_inertia[b][0] = I[0]; h.body_inertia[b][1] = I[4]; h.body_inertia[b][2] = I[8];
    h.body_inertia[b][3] = I[1]; h.body_inertia[b][4] = I[2]; h.body_inertia[b][5] = I[5];
    h.body_invw[b] = d->body_invweight0[2 * b];
  }
  for (int j = 0; j < NJ; j++) {
    for (int k = 0; k < 3; k++) h.jnt_axis[j][k] = d->jnt_axis[3 * j + k];
    h.jnt_lo[j] = d->jnt_range[2 * j]; h.jnt_hi[j] = d->jnt_range[2 * j + 1]; h.jnt_limited[j] = d->jnt_limited[j];
  }
  for (int i = 0; i < NV; i++) { h.dof_armature[i] = d->dof_armature[i]; h.dof_damping[i] = d->dof_damping[i]; h.dof_invw[i] = d->dof_invweight0[i]; }
  for (int u = 0; u < NU; u++) { h.gear[u + 6] = d->actuator_gear[u]; h.ctrl_lo[u + 6] = d->actuator_ctrlrange[2 * u]; h.ctrl_hi[u + 6] = d->actuator_ctrlrange[2 * u + 1]; }
  for (int g = 0; g < NG; g++) {
    h.geom_type[g] = d->geom_type[g]; h.geom_body[g] = d->geom_bodyid[g]; h.geom_condim[g] = d->geom_condim[g];
    if (h.geom_type[g] != GEOM_PLANE && h.geom_type[g] != GEOM_SPHERE && h.geom_type[g] != GEOM_CAPSULE && h.geom_type[g] != GEOM_BOX) { return mfail(err, DM_EUNSUPPORTED, "dm_model_create: unsupported geom type"); }
    if (h.geom_condim[g] != 1 && h.geom_condim[g] != 3) { return mfail(err, DM_EUNSUPPORTED, "dm_model_create: condim must be 1 or 3"); }
    for (int k = 0; k < 3; k++) { h.geom_pos[g][k] = d->geom_pos[3 * g + k]; h.geom_size[g][k] = d->geom_size[3 * g + k]; }
    for (int k = 0; k < 9; k++) h.geom_mat[g][k] = d->geom_mat[9 * g + k];
    h.geom_margin[g] = d->geom_margin[g]; h.geom_mu[g] = d->geom_friction[3 * g];
    const double* sz = d->geom_size + 3 * g;
    double rb = 0;
    if (h.geom_type[g] == GEOM_SPHERE) rb = sz[0];
    else if (h.geom_type[g] == GEOM_CAPSULE) rb = sz[0] + sz[1];
    else if (h.geom_type[g] == GEOM_BOX) rb = std::sqrt(sz[0] * sz[0] + sz[1] * sz[1] + sz[2] * sz[2]);
    h.geom_rbound[g] = rb;
  }
  { int nbox = 0; for (int g = 0; g < NG; g++) h.geom_boxslot[g] = (h.geom_type[g] == GEOM_BOX) ? nbox++ : -1;
    if (nbox > 4) return mfail(err, DM_EUNSUPPORTED, "dm_model_create: at most 4 box geoms are supported"); }
  { // Kp / Kd per hinge dof, joint order chest, neck, r_shoulder, r_elbow, l_shoulder, l_elbow, r_hip, r_knee, r_ankle, l_hip, l_knee, l_ankle
    // (PARAMS_KP_KD, src/mujoco/mocap_util.py:22-24; assembled like MujocoInterface.__init__, src/mujoco/mujoco_interface.py:66-72)
    static const double kp_j[12] = {1000, 100, 400, 300, 400, 300, 500, 500, 400, 500, 500, 400};
    static const double kd_j[12] = {100, 10, 40, 30, 40, 30, 50, 50, 40, 50, 50, 40};
    for (int dd = 6; dd < NV; dd++) { const int jb = dm::make_topo().dof_body[dd] - 2; h.kp[dd] = kp_j[jb]; h.kd[dd] = kd_j[jb]; }
  }
  h.npair = d->npair;
  { int nbb = 0;
    for (int p = 0; p < d->npair; p++) {
      const int a = d->pair_geom[2 * p], b = d->pair_geom[2 * p + 1];
      h.pair_g1[p] = (short)a; h.pair_g2[p] = (short)b; h.pair_stage[p] = -1;
      if (h.geom_type[a] == GEOM_PLANE && h.geom_type[b] == GEOM_BOX) h.pair_stage[p] = (short)h.geom_boxslot[b];
      else if (h.geom_type[a] == GEOM_BOX && h.geom_type[b] == GEOM_BOX) {
        if (nbb >= 2) return mfail(err, DM_EUNSUPPORTED, "dm_model_create: at most 2 box-box candidate pairs are supported");
        h.pair_stage[p] = (short)(4 + nbb++);
      }
    }
  }
  for (int p = 0; p < d->npair; p++) {
    const int a = h.pair_g1[p], b = h.pair_g2[p];
    auto& r = h.pair_rec[p];
    r.g1 = a; r.g2 = b;
    const int dim = h.geom_condim[a] > h.geom_condim[b] ? h.geom_condim[a] : h.geom_condim[b];
    r.t1t2 = h.geom_type[a] | (h.geom_type[b] << 8) | (dim << 16);
    r.meta = h.geom_body[a] | (h.geom_body[b] << 8) | ((h.pair_stage[p] + 1) << 16);
    r.margin = h.geom_margin[a] > h.geom_margin[b] ? h.geom_margin[a] : h.geom_margin[b];
    r.mu = h.geom_mu[a] > h.geom_mu[b] ? h.geom_mu[a] : h.geom_mu[b];
    r.bound = (h.geom_type[a] == GEOM_PLANE ? R(0) : h.geom_rbound[a]) + h.geom_rbound[b] + r.margin;
    r.tran = h.body_invw[h.geom_body[a]] + h.body_invw[h.geom_body[b]];
    for (int k = 0; k < 3; k++) { r.s1[k] = h.geom_size[a][k]; r.s2[k] = h.geom_size[b][k]; }
  }
  h.qpos0[0] = d->body_pos[3]; h.qpos0[1] = d->body_pos[4]; h.qpos0[2] = d->body_pos[5]; h.qpos0[3] = 1;
  h.timestep = d->timestep; h.tolerance = d->tolerance; h.meaninertia = d->meaninertia; h.iterations = d->iterations;
  for (int k = 0; k < 3; k++) h.gravity[k] = d->gravity[k];
  for (int k = 0; k < 2; k++) h.solref[k] = d->solref[k];
  for (int k = 0; k < 5; k++) h.solimp[k] = d->solimp[k];
  double tm = 0; for (int b = 0; b < NB; b++) tm += d->body_mass[b];
  h.total_mass = tm;
  // [MJ mj_makeImpedance] K, B from solref with the refsafe clamp timeconst >= 2*timestep
  const double tc = std::fmax(d->solref[0], 2 * d->timestep), dr = d->solref[1], dmax = d->solimp[1];
  h.K = 1 / std::fmax(DM_MINVAL, dmax * dmax * tc * tc * dr * dr);
  h.B = 2 / std::fmax(DM_MINVAL, dmax * tc);
  h.pgs_scale = 1 / (d->meaninertia * (NV > 1 ? NV : 1));
  h.pgs_detect = 1e-10;
  h.imp_rlo = 1.0 / std::pow(d->solimp[3], d->solimp[4] - 1); h.imp_rhi = 1.0 / std::pow(1 - d->solimp[3], d->solimp[4] - 1);
  h.enable_contact = 1; h.enable_limit = 1;
  return DM_OK;
}

}  // namespace dm
