/* dm_oracle.c — CPU ORACLE (test infrastructure; see dm_oracle.h for the parity statement).
 *
 * Restates, in straight-line float64 C for ONE environment:
 *   - what `MujocoEnv.do_simulation(action, n)` does at src/dp_env_v3.py:112, i.e. MuJoCo 2.0's
 *     mj_step on dp_env_v3.xml (EXTERNAL, closed source: restated from the published pipeline);
 *   - the env layer of src/dp_env_v3.py:62-164 (obs, done, rewards, set_state, resets).
 * Every "[MJ ...]" tag names the MuJoCo routine whose published behaviour the block follows.
 * Spatial quantities are expressed about the WORLD ORIGIN (MuJoCo uses the subtree COM of the root;
 * the two differ only by rounding).  Dense linear algebra throughout: clarity over speed.
 */
#include "dm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15 /* mjMINVAL */

/* =============================== small vector / quaternion helpers ============================== */
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void copy3(double* r, const double* a) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void zero3(double* r) { r[0] = r[1] = r[2] = 0; }
static inline void add3(double* r, const double* a, const double* b) { r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; }
static inline void sub3(double* r, const double* a, const double* b) { r[0] = a[0] - b[0]; r[1] = a[1] - b[1]; r[2] = a[2] - b[2]; }
static inline void addscl3(double* r, const double* a, const double* b, double s) { r[0] = a[0] + s * b[0]; r[1] = a[1] + s * b[1]; r[2] = a[2] + s * b[2]; }
static inline double norm3(const double* a) { return sqrt(dot3(a, a)); }
/* [MJ mju_normalize3] returns the norm; a (near-)zero vector becomes (1,0,0) */
static double normalize3(double* v) {
  double n = norm3(v);
  if (n < MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; }
  else { double s = 1.0 / n; v[0] *= s; v[1] *= s; v[2] *= s; }
  return n;
}
/* [MJ mju_normalize4] */
static void normalize4(double* q) {
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else if (fabs(n - 1) > MINVAL) { double s = 1.0 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
static void quat_mul(double* r, const double* a, const double* b) {
  double w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  double x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  double y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  double z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat2mat(double* m, const double* q) {
  double q00 = q[0] * q[0], q11 = q[1] * q[1], q22 = q[2] * q[2], q33 = q[3] * q[3];
  double q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3], q12 = q[1] * q[2], q13 = q[1] * q[3], q23 = q[2] * q[3];
  m[0] = q00 + q11 - q22 - q33; m[1] = 2 * (q12 - q03);       m[2] = 2 * (q13 + q02);
  m[3] = 2 * (q12 + q03);       m[4] = q00 - q11 + q22 - q33; m[5] = 2 * (q23 - q01);
  m[6] = 2 * (q13 - q02);       m[7] = 2 * (q23 + q01);       m[8] = q00 - q11 - q22 + q33;
}
static inline void mat_vec(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[1] * v[1] + m[2] * v[2], y = m[3] * v[0] + m[4] * v[1] + m[5] * v[2], z = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void matT_vec(double* r, const double* m, const double* v) {
  double x = m[0] * v[0] + m[3] * v[1] + m[6] * v[2], y = m[1] * v[0] + m[4] * v[1] + m[7] * v[2], z = m[2] * v[0] + m[5] * v[1] + m[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mat_mul3(double* r, const double* a, const double* b) {
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  memcpy(r, t, sizeof t);
}
static void quat_rot(double* r, const double* q, const double* v) { double m[9]; quat2mat(m, q); mat_vec(r, m, v); }
static void axisangle2quat(double* q, const double* axis, double angle) {
  if (angle == 0) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
/* [MJ mju_quatIntegrate] quat <- normalize(quat) * exp(scale * vel), vel in the local frame */
static void quat_integrate(double* quat, const double* vel, double scale) {
  double ax[3] = {vel[0], vel[1], vel[2]}, qr[4];
  double angle = scale * normalize3(ax);
  axisangle2quat(qr, ax, angle);
  normalize4(quat);
  quat_mul(quat, quat, qr);
}
static inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* spatial algebra, vectors are [ang(3); lin(3)] about the world origin */
static void cross_motion(double* r, const double* v, const double* s) {
  double a[3], b[3], c[3];
  cross3(a, v, s); cross3(b, v, s + 3); cross3(c, v + 3, s);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double* r, const double* v, const double* f) {
  double a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2]; r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
/* spatial inertia (about the origin) of a body with mass m, COM c, inertia Ic (world axes, about c),
 * stored as 10 numbers: Ic sym 3x3 about ORIGIN (6: xx,yy,zz,xy,xz,yz), m*c (3), m */
typedef struct { double I[9]; double mc[3]; double m; } sinert;
static void sinert_make(sinert* s, double m, const double* c, const double* Ic) {
  double cc = dot3(c, c);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) s->I[3 * i + j] = Ic[3 * i + j] + m * ((i == j ? cc : 0) - c[i] * c[j]);
  s->mc[0] = m * c[0]; s->mc[1] = m * c[1]; s->mc[2] = m * c[2]; s->m = m;
}
static void sinert_add(sinert* a, const sinert* b) {
  for (int i = 0; i < 9; i++) a->I[i] += b->I[i];
  for (int i = 0; i < 3; i++) a->mc[i] += b->mc[i];
  a->m += b->m;
}
/* f = I * v :  torque about origin = I_O w + mc x v ;  force = m v + w x mc */
static void sinert_mul(double* f, const sinert* s, const double* v) {
  double t[3], u[3];
  mat_vec(t, s->I, v); cross3(u, s->mc, v + 3);
  f[0] = t[0] + u[0]; f[1] = t[1] + u[1]; f[2] = t[2] + u[2];
  cross3(u, v, s->mc);
  f[3] = s->m * v[3] + u[0]; f[4] = s->m * v[4] + u[1]; f[5] = s->m * v[5] + u[2];
}
static inline double dot6(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }

/* ======================================= model: the humanoid ================================== */
/* dp_env_v3.xml restated (src/mujoco/humanoid_deepmimic/envs/asset/dp_env_v3.xml; line numbers cited) */
void dmo_humanoid_spec(dmo_spec* s) {
  memset(s, 0, sizeof *s);
  /* <option integrator="RK4" iterations="50" solver="PGS" timestep="0.0166"> :9 ; MuJoCo 2.0 defaults otherwise */
  s->timestep = 0.0166; s->iterations = 50; s->tolerance = 1e-8;
  s->gravity[0] = 0; s->gravity[1] = 0; s->gravity[2] = -9.81;
  s->solref[0] = 0.02; s->solref[1] = 1;
  s->solimp[0] = 0.9; s->solimp[1] = 0.95; s->solimp[2] = 0.001; s->solimp[3] = 0.5; s->solimp[4] = 2;
  /* bodies: parent, pos (:21,28,33,41,47,55,61,69,75,79,88,94,98) */
  static const int parent[14] = {0, 0, 1, 2, 2, 4, 2, 6, 1, 8, 9, 1, 11, 12};
  static const double bpos[14][3] = {
      {0, 0, 0}, {0, 0, 0.9}, {0, 0, 0.236151}, {0, 0, 0.223894},
      {-0.02405, -0.18311, 0.2435}, {0, 0, -0.274788}, {-0.02405, 0.18311, 0.2435}, {0, 0, -0.274788},
      {0, -0.084887, 0}, {0, 0, -0.421546}, {0, 0, -0.40987},
      {0, 0.084887, 0}, {0, 0, -0.421546}, {0, 0, -0.40987}};
  s->nbody = 14;
  for (int b = 0; b < 14; b++) { s->body_parent[b] = parent[b]; copy3(s->body_pos[b], bpos[b]); }
  /* joints: free root (:25, armature 0 damping 0 unlimited) then hinges with default armature=1 damping=1 limited (:4) */
  int j = 0;
  s->jnt_type[j] = DMO_JNT_FREE; s->jnt_body[j] = 1; s->jnt_limited[j] = 0; s->jnt_axis[j][2] = 1; j++;
#define HINGE(body, ax, ay, az, lo, hi) do { s->jnt_type[j] = DMO_JNT_HINGE; s->jnt_body[j] = body; s->jnt_limited[j] = 1; \
    s->jnt_axis[j][0] = ax; s->jnt_axis[j][1] = ay; s->jnt_axis[j][2] = az; s->jnt_range[j][0] = lo; s->jnt_range[j][1] = hi; \
    s->jnt_armature[j] = 1; s->jnt_damping[j] = 1; j++; } while (0)
  HINGE(2, 1, 0, 0, -1.2, 1.2); HINGE(2, 0, 1, 0, -1.2, 1.2); HINGE(2, 0, 0, 1, -1.2, 1.2);       /* chest :30-32 */
  HINGE(3, 1, 0, 0, -1.0, 1.0); HINGE(3, 0, 1, 0, -1.0, 1.0); HINGE(3, 0, 0, 1, -1.0, 1.0);       /* neck :35-37 */
  HINGE(4, 1, 0, 0, -3.14, 0.5); HINGE(4, 0, 1, 0, -3.14, 0.7); HINGE(4, 0, 0, 1, -1.5, 1.5);     /* right_shoulder :44-46 */
  HINGE(5, 0, -1, 0, 0, 2.8);                                                                     /* right_elbow :49 */
  HINGE(6, 1, 0, 0, -0.5, 3.14); HINGE(6, 0, 1, 0, -3.14, 0.7); HINGE(6, 0, 0, 1, -1.5, 1.5);     /* left_shoulder :57-59 */
  HINGE(7, 0, -1, 0, 0, 2.8);                                                                     /* left_elbow :63 */
  HINGE(8, 1, 0, 0, -1.2, 1.2); HINGE(8, 0, 1, 0, -2.57, 1.57); HINGE(8, 0, 0, 1, -1.0, 1.0);     /* right_hip :72-74 */
  HINGE(9, 0, -1, 0, -2.7, 0.0);                                                                  /* right_knee :78 */
  HINGE(10, 1, 0, 0, -1.0, 1.0); HINGE(10, 0, 1, 0, -1.0, 1.57); HINGE(10, 0, 0, 1, -1.0, 1.0);   /* right_ankle :80-82 */
  HINGE(11, 1, 0, 0, -1.2, 1.2); HINGE(11, 0, 1, 0, -2.57, 1.57); HINGE(11, 0, 0, 1, -1.0, 1.0);  /* left_hip :91-93 */
  HINGE(12, 0, -1, 0, -2.7, 0.0);                                                                 /* left_knee :97 */
  HINGE(13, 1, 0, 0, -1.0, 1.0); HINGE(13, 0, 1, 0, -1.0, 1.57); HINGE(13, 0, 0, 1, -1.0, 1.0);   /* left_ankle :99-101 */
#undef HINGE
  s->njnt = j;
  /* geoms in XML order; defaults contype=conaffinity=1 condim=1 margin=0.001 (:5), friction default (1,.005,.0001) */
  int g = 0;
#define GEOM_COMMON(body, type, mass_) do { s->geom_type[g] = type; s->geom_body[g] = body; s->geom_condim[g] = 1; \
    s->geom_contype[g] = 1; s->geom_conaffinity[g] = 1; s->geom_mass[g] = mass_; s->geom_margin[g] = 0.001; \
    s->geom_friction[g][0] = 1; s->geom_friction[g][1] = 0.005; s->geom_friction[g][2] = 0.0001; } while (0)
#define SPHERE(body, r, px, py, pz, mass_) do { GEOM_COMMON(body, DMO_GEOM_SPHERE, mass_); s->geom_size[g][0] = r; \
    s->geom_pos[g][0] = px; s->geom_pos[g][1] = py; s->geom_pos[g][2] = pz; g++; } while (0)
#define CAPSULE(body, r, z0, z1, mass_) do { GEOM_COMMON(body, DMO_GEOM_CAPSULE, mass_); s->geom_size[g][0] = r; \
    s->geom_has_fromto[g] = 1; s->geom_fromto[g][2] = z0; s->geom_fromto[g][5] = z1; g++; } while (0)
#define BOX(body, sx, sy, sz, px, py, pz, mass_) do { GEOM_COMMON(body, DMO_GEOM_BOX, mass_); s->geom_size[g][0] = sx; \
    s->geom_size[g][1] = sy; s->geom_size[g][2] = sz; s->geom_pos[g][0] = px; s->geom_pos[g][1] = py; s->geom_pos[g][2] = pz; g++; } while (0)
  /* floor :19 — plane, condim 3, friction (1,.1,.1) */
  GEOM_COMMON(0, DMO_GEOM_PLANE, 0.0); s->geom_condim[g] = 3; s->geom_friction[g][1] = 0.1; s->geom_friction[g][2] = 0.1;
  s->geom_size[g][0] = 50; s->geom_size[g][1] = 50; s->geom_size[g][2] = 0.2; g++;
  SPHERE(1, 0.09, 0, 0, 0.07, 6.0);           /* root :22 */
  SPHERE(2, 0.11, 0, 0, 0.12, 14.0);          /* chest :29 */
  SPHERE(3, 0.1025, 0, 0, 0.175, 2.0);        /* neck :34 */
  CAPSULE(4, 0.045, -0.05, -0.23, 1.5);       /* right_shoulder :42 */
  CAPSULE(5, 0.04, -0.0525, -0.1875, 1.0);    /* right_elbow :48 */
  SPHERE(5, 0.04, 0, 0, -0.258947, 0.5);      /* right_wrist :51 (on body right_elbow) */
  CAPSULE(6, 0.045, -0.05, -0.23, 1.5);       /* left_shoulder :56 */
  CAPSULE(7, 0.04, -0.0525, -0.1875, 1.0);    /* left_elbow :62 */
  SPHERE(7, 0.04, 0, 0, -0.258947, 0.5);      /* left_wrist :65 */
  CAPSULE(8, 0.055, -0.06, -0.36, 4.5);       /* right_hip :70 */
  CAPSULE(9, 0.05, -0.045, -0.355, 3.0);      /* right_knee :76 */
  BOX(10, 0.0885, 0.045, 0.0275, 0.045, 0, -0.0225, 1.0); /* right_ankle :84 */
  CAPSULE(11, 0.055, -0.06, -0.36, 4.5);      /* left_hip :89 */
  CAPSULE(12, 0.05, -0.045, -0.355, 3.0);     /* left_knee :95 */
  BOX(13, 0.0885, 0.045, 0.0275, 0.045, 0, -0.0225, 1.0); /* left_ankle :103 */
#undef GEOM_COMMON
#undef SPHERE
#undef CAPSULE
#undef BOX
  s->ngeom = g;
  /* <contact><exclude> :110-117 (all parent/child pairs, redundant with the default parent filter) */
  static const int excl[8][2] = {{8, 1}, {11, 1}, {8, 9}, {11, 12}, {9, 10}, {12, 13}, {5, 4}, {7, 6}};
  s->nexclude = 8;
  for (int i = 0; i < 8; i++) { s->exclude[i][0] = excl[i][0]; s->exclude[i][1] = excl[i][1]; }
  /* motors :121-155, one per hinge in hinge order; ctrlrange +-0.5 (:7) */
  static const double gear[28] = {200, 200, 200, 50, 50, 50, 100, 100, 100, 60, 100, 100, 100, 60,
                                  200, 200, 200, 150, 90, 90, 90, 200, 200, 200, 150, 90, 90, 90};
  s->nu = 28;
  for (int u = 0; u < 28; u++) { s->act_jnt[u] = u + 1; s->act_gear[u] = gear[u]; s->act_ctrlrange[u][0] = -0.5; s->act_ctrlrange[u][1] = 0.5; }
}

/* ==================================== position-stage kinematics =============================== */
/* [MJ mj_kinematics] (engine_core_smooth) + geom poses + cdof about the origin */
static void kinematics(const dmo_model* m, dmo_data* d) {
  const dmo_spec* s = &m->s;
  zero3(d->xpos[0]); d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  quat2mat(d->xmat[0], d->xquat[0]); zero3(d->xipos[0]);
  for (int b = 1; b < s->nbody; b++) {
    double xpos[3], xquat[4];
    int j0 = m->body_jntadr[b], nj = m->body_jntnum[b], p = s->body_parent[b];
    if (nj == 1 && s->jnt_type[j0] == DMO_JNT_FREE) {
      int qa = m->jnt_qposadr[j0];
      copy3(xpos, d->qpos + qa);
      memcpy(xquat, d->qpos + qa + 3, 4 * sizeof(double));
      normalize4(xquat);
      copy3(d->xanchor[j0], xpos);
      copy3(d->xaxis[j0], s->jnt_axis[j0]);
    } else {
      double v[3];
      mat_vec(v, d->xmat[p], s->body_pos[b]);
      add3(xpos, d->xpos[p], v);
      memcpy(xquat, d->xquat[p], 4 * sizeof(double)); /* body_quat = identity for every body of this model */
      for (int k = 0; k < nj; k++) {
        int jid = j0 + k, qa = m->jnt_qposadr[jid];
        double ql[4];
        quat_rot(d->xaxis[jid], xquat, s->jnt_axis[jid]);
        copy3(d->xanchor[jid], xpos); /* jnt_pos = 0 for every joint of this model */
        axisangle2quat(ql, s->jnt_axis[jid], d->qpos[qa] - m->qpos0[qa]);
        quat_mul(xquat, xquat, ql);
      }
    }
    normalize4(xquat);
    copy3(d->xpos[b], xpos); memcpy(d->xquat[b], xquat, sizeof xquat);
    quat2mat(d->xmat[b], xquat);
    double v[3];
    mat_vec(v, d->xmat[b], m->body_ipos[b]);
    add3(d->xipos[b], xpos, v);
  }
  for (int g = 0; g < s->ngeom; g++) {
    int b = s->geom_body[g];
    double v[3], gm[9];
    mat_vec(v, d->xmat[b], m->geom_lpos[g]);
    add3(d->geom_xpos[g], d->xpos[b], v);
    quat2mat(gm, m->geom_quat[g]);
    mat_mul3(d->geom_xmat[g], d->xmat[b], gm);
  }
  /* motion axes [MJ mj_comPos cdof], reference point = origin: [axis; anchor x axis]; translations [0; e] */
  for (int j = 0; j < s->njnt; j++) {
    int da = m->jnt_dofadr[j];
    if (s->jnt_type[j] == DMO_JNT_FREE) {
      int b = s->jnt_body[j];
      for (int k = 0; k < 3; k++) { memset(d->cdof[da + k], 0, 6 * sizeof(double)); d->cdof[da + k][3 + k] = 1; }
      for (int k = 0; k < 3; k++) {
        double ax[3] = {d->xmat[b][k], d->xmat[b][3 + k], d->xmat[b][6 + k]};
        copy3(d->cdof[da + 3 + k], ax);
        cross3(d->cdof[da + 3 + k] + 3, d->xpos[b], ax);
      }
    } else {
      copy3(d->cdof[da], d->xaxis[j]);
      cross3(d->cdof[da] + 3, d->xanchor[j], d->xaxis[j]);
    }
  }
}

static void body_sinert(const dmo_model* m, const dmo_data* d, int b, sinert* out) {
  double t[9], Iw[9], Rt[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = d->xmat[b][3 * j + i];
  mat_mul3(t, d->xmat[b], m->body_inertia[b]);
  mat_mul3(Iw, t, Rt);
  sinert_make(out, m->body_mass[b], d->xipos[b], Iw);
}

/* [MJ mj_crb + mj_factorM] composite-rigid-body mass matrix (dense, + armature) and its Cholesky factor */
static void mass_matrix(const dmo_model* m, dmo_data* d) {
  const dmo_spec* s = &m->s;
  int nv = m->nv;
  sinert crb[DMO_MAXBODY];
  for (int b = 0; b < s->nbody; b++) body_sinert(m, d, b, &crb[b]);
  for (int b = s->nbody - 1; b > 0; b--) if (s->body_parent[b] > 0) sinert_add(&crb[s->body_parent[b]], &crb[b]);
  for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) d->M[i][j] = 0;
  for (int i = 0; i < nv; i++) {
    double f[6];
    sinert_mul(f, &crb[m->dof_body[i]], d->cdof[i]);
    for (int j = i; j >= 0; j = m->dof_parent[j]) { d->M[i][j] = dot6(d->cdof[j], f); d->M[j][i] = d->M[i][j]; }
    d->M[i][i] += m->dof_armature[i];
  }
  /* Cholesky M = L L^T */
  for (int i = 0; i < nv; i++) {
    for (int j = 0; j <= i; j++) {
      double sum = d->M[i][j];
      for (int k = 0; k < j; k++) sum -= d->L[i][k] * d->L[j][k];
      d->L[i][j] = (i == j) ? sqrt(sum) : sum / d->L[j][j];
    }
    for (int j = i + 1; j < nv; j++) d->L[i][j] = 0;
  }
}
/* x <- M^-1 x */
static void solve_M(const dmo_model* m, const dmo_data* d, double* x) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) { double sum = x[i]; for (int k = 0; k < i; k++) sum -= d->L[i][k] * x[k]; x[i] = sum / d->L[i][i]; }
  for (int i = nv - 1; i >= 0; i--) { double sum = x[i]; for (int k = i + 1; k < nv; k++) sum -= d->L[k][i] * x[k]; x[i] = sum / d->L[i][i]; }
}

/* translational / rotational Jacobian of a point p fixed to body b [MJ mj_jac] */
static void jac_point(const dmo_model* m, const dmo_data* d, int b, const double* p, double (*jacp)[DMO_MAXV], double (*jacr)[DMO_MAXV]) {
  int nv = m->nv;
  for (int r = 0; r < 3; r++) for (int c = 0; c < nv; c++) { if (jacp) jacp[r][c] = 0; if (jacr) jacr[r][c] = 0; }
  if (b <= 0 || m->body_dofnum[b] == 0) {
    /* walk up to the first ancestor that has dofs */
    while (b > 0 && m->body_dofnum[b] == 0) b = m->s.body_parent[b];
    if (b <= 0) return;
  }
  for (int i = m->body_dofadr[b] + m->body_dofnum[b] - 1; i >= 0; i = m->dof_parent[i]) {
    double v[3];
    cross3(v, d->cdof[i], p);
    for (int r = 0; r < 3; r++) { if (jacp) jacp[r][i] = v[r] + d->cdof[i][3 + r]; if (jacr) jacr[r][i] = d->cdof[i][r]; }
  }
}

/* ======================================== model compiler ====================================== */
static void z2quat(double* q, const double* vec) { /* [MJ mjuu_z2quat] rotation taking +z to vec */
  double z[3] = {0, 0, 1}, ax[3];
  cross3(ax, z, vec);
  double sn = norm3(ax);
  if (sn < 1e-10) { ax[0] = 1; ax[1] = 0; ax[2] = 0; } else { ax[0] /= sn; ax[1] /= sn; ax[2] /= sn; }
  double ang = atan2(sn, vec[2]);
  q[0] = cos(ang / 2); q[1] = ax[0] * sin(ang / 2); q[2] = ax[1] * sin(ang / 2); q[3] = ax[2] * sin(ang / 2);
}

int dmo_compile(const dmo_spec* sp, dmo_model* m) {
  memset(m, 0, sizeof *m);
  m->s = *sp;
  const dmo_spec* s = &m->s;
  m->enable_contact = 1; m->enable_limit = 1;
  m->pyramid_diag_mu2 = 1; m->pyramid_r_rescale = 1; m->max_efc = DMO_MAXEFC;
  if (s->nbody > DMO_MAXBODY || s->njnt > DMO_MAXJNT || s->ngeom > DMO_MAXGEOM || s->nu > DMO_MAXU) return -1;
  /* address tables */
  int nq = 0, nv = 0;
  for (int b = 0; b < s->nbody; b++) { m->body_jntadr[b] = -1; m->body_dofadr[b] = -1; }
  for (int j = 0; j < s->njnt; j++) {
    int b = s->jnt_body[j];
    if (m->body_jntnum[b] == 0) { m->body_jntadr[b] = j; m->body_dofadr[b] = nv; }
    m->body_jntnum[b]++;
    m->jnt_qposadr[j] = nq; m->jnt_dofadr[j] = nv;
    int dq = s->jnt_type[j] == DMO_JNT_FREE ? 7 : 1, dv = s->jnt_type[j] == DMO_JNT_FREE ? 6 : 1;
    for (int k = 0; k < dv; k++) {
      m->dof_body[nv + k] = b; m->dof_jnt[nv + k] = j;
      m->dof_armature[nv + k] = s->jnt_armature[j]; m->dof_damping[nv + k] = s->jnt_damping[j];
    }
    m->body_dofnum[b] += dv;
    nq += dq; nv += dv;
  }
  if (nq > DMO_MAXQ || nv > DMO_MAXV) return -2;
  m->nq = nq; m->nv = nv;
  for (int i = 0; i < nv; i++) {
    int b = m->dof_body[i];
    if (i > m->body_dofadr[b]) { m->dof_parent[i] = i - 1; continue; }
    int p = s->body_parent[b];
    while (p > 0 && m->body_dofnum[p] == 0) p = s->body_parent[p];
    m->dof_parent[i] = (p > 0) ? m->body_dofadr[p] + m->body_dofnum[p] - 1 : -1;
  }
  /* qpos0: free joint = body pos + identity quat (body at its XML pose), hinges 0 */
  for (int j = 0; j < s->njnt; j++) if (s->jnt_type[j] == DMO_JNT_FREE) {
    int qa = m->jnt_qposadr[j], b = s->jnt_body[j];
    copy3(m->qpos0 + qa, s->body_pos[b]); m->qpos0[qa + 3] = 1;
  }
  /* geom local frames (fromto -> pos/quat/half-length) [MJ mjCGeom::SetFromTo / mjuu_z2quat] */
  for (int g = 0; g < s->ngeom; g++) {
    m->geom_quat[g][0] = 1;
    copy3(m->geom_lpos[g], s->geom_pos[g]); copy3(m->geom_lsize[g], s->geom_size[g]);
    if (s->geom_has_fromto[g]) {
      double vec[3];
      sub3(vec, s->geom_fromto[g] + 3, s->geom_fromto[g]);
      double len = norm3(vec);
      for (int k = 0; k < 3; k++) m->geom_lpos[g][k] = 0.5 * (s->geom_fromto[g][k] + s->geom_fromto[g][3 + k]);
      m->geom_lsize[g][1] = 0.5 * len;
      vec[0] /= len; vec[1] /= len; vec[2] /= len;
      z2quat(m->geom_quat[g], vec);
    }
  }
  /* inertiafromgeom (:2): body mass / COM / inertia from its geoms [MJ mjCGeom::SetInertia, mjCBody::GeomFrame] */
  for (int b = 0; b < s->nbody; b++) {
    double mass = 0, com[3] = {0, 0, 0};
    for (int g = 0; g < s->ngeom; g++) if (s->geom_body[g] == b) { mass += s->geom_mass[g]; for (int k = 0; k < 3; k++) com[k] += s->geom_mass[g] * m->geom_lpos[g][k]; }
    m->body_mass[b] = mass;
    if (mass <= 0) continue;
    for (int k = 0; k < 3; k++) com[k] /= mass;
    copy3(m->body_ipos[b], com);
    for (int g = 0; g < s->ngeom; g++) if (s->geom_body[g] == b) {
      double mg = s->geom_mass[g], I[3] = {0, 0, 0};
      const double* sz = m->geom_lsize[g];
      if (s->geom_type[g] == DMO_GEOM_SPHERE) { I[0] = I[1] = I[2] = 0.4 * mg * sz[0] * sz[0]; }
      else if (s->geom_type[g] == DMO_GEOM_BOX) {
        I[0] = mg / 3 * (sz[1] * sz[1] + sz[2] * sz[2]); I[1] = mg / 3 * (sz[0] * sz[0] + sz[2] * sz[2]); I[2] = mg / 3 * (sz[0] * sz[0] + sz[1] * sz[1]);
      } else if (s->geom_type[g] == DMO_GEOM_CAPSULE) {
        double r = sz[0], h = sz[1];
        double vs = 4.0 / 3.0 * r, vc = 2 * h; /* volumes / (pi r^2) */
        double ms = mg * vs / (vs + vc), mc = mg - ms;
        I[2] = mc * r * r / 2 + 0.4 * ms * r * r;
        I[0] = I[1] = mc * (3 * r * r + 4 * h * h) / 12 + ms * (0.4 * r * r + h * h + 0.75 * r * h);
      }
      double R[9], D[9] = {I[0], 0, 0, 0, I[1], 0, 0, 0, I[2]}, Rt[9], t[9], Ig[9], rr[3];
      quat2mat(R, m->geom_quat[g]);
      for (int i = 0; i < 3; i++) for (int jx = 0; jx < 3; jx++) Rt[3 * i + jx] = R[3 * jx + i];
      mat_mul3(t, R, D); mat_mul3(Ig, t, Rt);
      sub3(rr, m->geom_lpos[g], com);
      double r2 = dot3(rr, rr);
      for (int i = 0; i < 3; i++) for (int jx = 0; jx < 3; jx++) m->body_inertia[b][3 * i + jx] += Ig[3 * i + jx] + mg * ((i == jx ? r2 : 0) - rr[i] * rr[jx]);
    }
    m->total_mass += mass;
  }
  /* candidate geom pairs, in the order MuJoCo lists contacts: body pairs ascending (b1 < b2), then geoms.
   * [MJ mj_collision / filterBodyPair]: same body and parent-child pairs are skipped unless the parent is the world;
   * <exclude> pairs skipped; contype/conaffinity must match; geom1 is the lower geom TYPE. */
  m->npair = 0;
  for (int b1 = 0; b1 < s->nbody; b1++) for (int b2 = b1 + 1; b2 < s->nbody; b2++) {
    if (b1 != 0 && (s->body_parent[b2] == b1 || s->body_parent[b1] == b2)) continue;
    int ex = 0;
    for (int e = 0; e < s->nexclude; e++) if ((s->exclude[e][0] == b1 && s->exclude[e][1] == b2) || (s->exclude[e][0] == b2 && s->exclude[e][1] == b1)) ex = 1;
    if (ex) continue;
    for (int g1 = 0; g1 < s->ngeom; g1++) if (s->geom_body[g1] == b1) for (int g2 = 0; g2 < s->ngeom; g2++) if (s->geom_body[g2] == b2) {
      if (!((s->geom_contype[g1] & s->geom_conaffinity[g2]) || (s->geom_contype[g2] & s->geom_conaffinity[g1]))) continue;
      if (b1 == 0 && s->geom_type[g1] == DMO_GEOM_PLANE && s->geom_type[g2] == DMO_GEOM_PLANE) continue;
      if (m->npair >= DMO_MAXPAIR) return -3;
      int a = g1, c = g2;
      if (s->geom_type[a] > s->geom_type[c]) { int t = a; a = c; c = t; }
      m->pair_g1[m->npair] = a; m->pair_g2[m->npair] = c; m->npair++;
    }
  }
  /* [MJ mj_setConst / set0]: at qpos0 compute M, dof_invweight0, body_invweight0, meaninertia */
  dmo_data* d = dmo_data_create(m);
  kinematics(m, d); mass_matrix(m, d);
  double Minv[DMO_MAXV][DMO_MAXV];
  for (int i = 0; i < nv; i++) {
    double e[DMO_MAXV]; for (int k = 0; k < nv; k++) e[k] = 0; e[i] = 1;
    solve_M(m, d, e);
    for (int k = 0; k < nv; k++) Minv[k][i] = e[k];
  }
  m->meaninertia = 0;
  for (int i = 0; i < nv; i++) m->meaninertia += d->M[i][i];
  m->meaninertia /= nv > 0 ? nv : 1;
  for (int j = 0; j < s->njnt; j++) {
    int da = m->jnt_dofadr[j];
    if (s->jnt_type[j] == DMO_JNT_FREE) {
      double t = (Minv[da][da] + Minv[da + 1][da + 1] + Minv[da + 2][da + 2]) / 3, r = (Minv[da + 3][da + 3] + Minv[da + 4][da + 4] + Minv[da + 5][da + 5]) / 3;
      for (int k = 0; k < 3; k++) { m->dof_invweight0[da + k] = t; m->dof_invweight0[da + 3 + k] = r; }
    } else m->dof_invweight0[da] = Minv[da][da];
  }
  for (int b = 1; b < s->nbody; b++) {
    double jp[3][DMO_MAXV], jr[3][DMO_MAXV], acc[2] = {0, 0};
    jac_point(m, d, b, d->xipos[b], jp, jr);
    for (int r = 0; r < 3; r++) {
      double tp[DMO_MAXV], tr[DMO_MAXV];
      for (int k = 0; k < nv; k++) { tp[k] = 0; tr[k] = 0; for (int l = 0; l < nv; l++) { tp[k] += Minv[k][l] * jp[r][l]; tr[k] += Minv[k][l] * jr[r][l]; } }
      for (int k = 0; k < nv; k++) { acc[0] += jp[r][k] * tp[k]; acc[1] += jr[r][k] * tr[k]; }
    }
    m->body_invweight0[b][0] = acc[0] / 3; m->body_invweight0[b][1] = acc[1] / 3;
  }
  dmo_data_destroy(d);
  return 0;
}

dmo_data* dmo_data_create(const dmo_model* m) {
  dmo_data* d = (dmo_data*)calloc(1, sizeof(dmo_data));
  d->efc_AR = (double (*)[DMO_MAXEFC])calloc((size_t)DMO_MAXEFC * DMO_MAXEFC, sizeof(double));
  dmo_reset_data(m, d);
  return d;
}
void dmo_data_destroy(dmo_data* d) { if (d) { free(d->efc_AR); free(d); } }
/* [MJ mj_resetData] what gym's MujocoEnv.reset() -> sim.reset() does before reset_model (src/dp_env_v3.py:148) */
void dmo_reset_data(const dmo_model* m, dmo_data* d) {
  memcpy(d->qpos, m->qpos0, sizeof(double) * DMO_MAXQ);
  memset(d->qvel, 0, sizeof d->qvel); memset(d->ctrl, 0, sizeof d->ctrl);
  memset(d->qacc_warmstart, 0, sizeof d->qacc_warmstart);
  d->time = 0;
}

/* ========================================== collision ========================================= */
typedef struct { double dist, pos[3], frame[6]; } rawcon;

/* [MJ mjc_PlaneSphere-style core] plane through p0 with unit normal n vs sphere (c, r) */
static int plane_sphere(rawcon* con, const double* p0, const double* n, const double* c, double r, double margin) {
  double t[3]; sub3(t, c, p0);
  double cdist = dot3(t, n);
  if (cdist > margin + r) return 0;
  con->dist = cdist - r;
  copy3(con->frame, n); zero3(con->frame + 3);
  addscl3(con->pos, c, n, -con->dist / 2 - r);
  return 1;
}
/* [MJ _SphereSphere] */
static int sphere_sphere(rawcon* con, const double* c1, double r1, const double* c2, double r2, double margin) {
  double dif[3]; sub3(dif, c2, c1);
  double bound = margin + r1 + r2;
  if (dot3(dif, dif) > bound * bound) return 0;
  copy3(con->frame, dif);
  con->dist = normalize3(con->frame) - r1 - r2;
  zero3(con->frame + 3);
  addscl3(con->pos, c1, con->frame, r1 + 0.5 * con->dist);
  return 1;
}
/* ---- diagnostics (round 6; no effect on any result): which CASE of the two own narrow-phase routines a call took --------------------------------------
 * The contact LIST — (geom1, geom2) per contact, the quantity the parity bar wants bit-exact — of these two routines can differ from MuJoCo's
 * mjc_BoxBox / mjc_CapsuleBox by construction; the tallies bound how often (tests/test_oracle_physics.py):
 *   box-box      [0] separated (no contact: a separating axis beyond the margin — the same decision in any exact SAT)
 *                [1] edge-edge: ONE contact here; MuJoCo's routine also reports one point for an edge-edge configuration            -> same count
 *                [2] face axis, no clipped vertex within the margin: no contact                                                      -> same count
 *                [3] face axis, 1 .. 4 clipped vertices within the margin, all kept (polygon order)                                  -> same count; the ORDER
 *                    of the points inside the pair may differ (MuJoCo walks the incident face's vertices, then the reference face's), which does not change
 *                    the (geom1, geom2) list
 *                [4] face axis, 5 .. 8 clipped vertices within the margin, PRUNED to the 4 deepest here; MuJoCo's routine returns up to 8   -> COUNT DIFFERS
 *   capsule-box  [5] rejected (slab test / closest point beyond radius + margin): no contact                                          -> same count
 *                [6] one contact, NEITHER end of the axis segment within radius + margin of the box (the capsule touches with its side)
 *                [7] one contact, ONE end within reach (an end pokes the box)
 *                [8] one contact, BOTH ends within reach (the capsule lies along the box)
 *                [9] one contact, the axis passes through the box's interior (zero-distance plateau; any of the above end counts)
 *                    MuJoCo's routine adds a SECOND sphere-box contact at another point of the segment unless the capsule points at or away from the
 *                    closest corner, whenever that second sphere is within the margin: it cannot in [6] with a short reach interval, it can in [7],
 *                    it does in [8]                                                                                                   -> COUNT MAY DIFFER (1 here, 2 there)
 * Enabled by dmo_narrow_cases(out, 1) (reset + on); off by default so that the timed cpu_baseline leg does not pay for the end tests. */
static long long g_np_case[10];
static int g_np_diag = 0;
static void np_tally(int k) {
  if (!g_np_diag) return;
#pragma omp atomic
  g_np_case[k]++;
}
/* out[10] <- the tallies; mode 1: reset and switch on, 0: switch off, -1: just read */
void dmo_narrow_cases(long long* out, int mode) {
  if (out) for (int k = 0; k < 10; k++) out[k] = g_np_case[k];
  if (mode == 1) { for (int k = 0; k < 10; k++) g_np_case[k] = 0; g_np_diag = 1; }
  if (mode == 0) g_np_diag = 0;
}
/* OWN ALGORITHM for box-box (MuJoCo's mjc_BoxBox, ~700 lines derived from ODE, is not restated): separating-axis test
 * over the 6 face normals and 9 edge cross products; the axis of least penetration decides between
 *   - face contact: the incident face of the other box is clipped (Sutherland-Hodgman) against the side planes of the
 *     reference face; clipped vertices within `margin` of the reference face become contacts (the 4 deepest are kept);
 *   - edge contact: one contact at the midpoint of the closest points of the two supporting edges.
 * All contacts share the normal (from geom1 to geom2).  The HIP kernel runs the identical procedure. */
static int box_box(rawcon* con, const double* p1, const double* m1, const double* s1, const double* p2, const double* m2,
                   const double* s2, double margin) {
  double A[3][3], B[3][3], d[3], Rm[3][3], aR[3][3];
  for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { A[i][k] = m1[3 * k + i]; B[i][k] = m2[3 * k + i]; }
  sub3(d, p2, p1);
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rm[i][j] = dot3(A[i], B[j]); aR[i][j] = fabs(Rm[i][j]); }
  double best = -1e300; int code = -1;
  for (int i = 0; i < 3; i++) {
    double sep = fabs(dot3(d, A[i])) - (s1[i] + s2[0] * aR[i][0] + s2[1] * aR[i][1] + s2[2] * aR[i][2]);
    if (sep > margin) { np_tally(0); return 0; }
    if (sep > best) { best = sep; code = i; }
  }
  for (int j = 0; j < 3; j++) {
    double sep = fabs(dot3(d, B[j])) - (s2[j] + s1[0] * aR[0][j] + s1[1] * aR[1][j] + s1[2] * aR[2][j]);
    if (sep > margin) { np_tally(0); return 0; }
    if (sep > best) { best = sep; code = 3 + j; }
  }
  double ebest = -1e300, en[3] = {0, 0, 0}; int ei = -1, ej = -1;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double L[3]; cross3(L, A[i], B[j]);
    double len = norm3(L);
    if (len < 1e-6) continue;
    L[0] /= len; L[1] /= len; L[2] /= len;
    double rA = s1[0] * fabs(dot3(A[0], L)) + s1[1] * fabs(dot3(A[1], L)) + s1[2] * fabs(dot3(A[2], L));
    double rB = s2[0] * fabs(dot3(B[0], L)) + s2[1] * fabs(dot3(B[1], L)) + s2[2] * fabs(dot3(B[2], L));
    double dl = dot3(d, L), sep = fabs(dl) - rA - rB;
    if (sep > margin) { np_tally(0); return 0; }
    if (sep > ebest) { ebest = sep; ei = i; ej = j; double sg = dl < 0 ? -1 : 1; en[0] = sg * L[0]; en[1] = sg * L[1]; en[2] = sg * L[2]; }
  }
  if (ei >= 0 && ebest > best + 1e-6) {
    /* edge-edge: supporting edges, closest points of the two lines */
    double ca[3], cb[3];
    copy3(ca, p1); copy3(cb, p2);
    for (int k = 0; k < 3; k++) if (k != ei) { double sg = dot3(en, A[k]) > 0 ? 1 : -1; addscl3(ca, ca, A[k], sg * s1[k]); }
    for (int k = 0; k < 3; k++) if (k != ej) { double sg = dot3(en, B[k]) > 0 ? -1 : 1; addscl3(cb, cb, B[k], sg * s2[k]); }
    double w[3]; sub3(w, ca, cb);
    double ab = dot3(A[ei], B[ej]), aw = dot3(A[ei], w), bw = dot3(B[ej], w), den = 1 - ab * ab;
    double ta = den > 1e-12 ? (ab * bw - aw) / den : 0, tb = den > 1e-12 ? (bw - ab * aw) / den : 0;
    ta = clampd(ta, -s1[ei], s1[ei]); tb = clampd(tb, -s2[ej], s2[ej]);
    double qa[3], qb[3];
    addscl3(qa, ca, A[ei], ta); addscl3(qb, cb, B[ej], tb);
    con->dist = ebest;
    copy3(con->frame, en); zero3(con->frame + 3);
    for (int k = 0; k < 3; k++) con->pos[k] = 0.5 * (qa[k] + qb[k]);
    np_tally(1);
    return 1;
  }
  /* face contact: reference box owns the axis */
  const int refB = code >= 3, ax = refB ? code - 3 : code;
  const double *pr = refB ? p2 : p1, *pi = refB ? p1 : p2, *sr = refB ? s2 : s1, *si = refB ? s1 : s2;
  double (*Rr)[3] = refB ? B : A, (*Ri)[3] = refB ? A : B;
  double dr[3]; sub3(dr, pi, pr);
  double sgn = dot3(dr, Rr[ax]) < 0 ? -1 : 1, n[3] = {sgn * Rr[ax][0], sgn * Rr[ax][1], sgn * Rr[ax][2]}; /* ref -> inc */
  int k = 0; double bestdot = -1;
  for (int j = 0; j < 3; j++) { double dd = fabs(dot3(n, Ri[j])); if (dd > bestdot) { bestdot = dd; k = j; } }
  double fs = dot3(n, Ri[k]) > 0 ? -1 : 1; /* incident face = the one facing the reference box */
  int k1 = (k + 1) % 3, k2 = (k + 2) % 3, u = (ax + 1) % 3, v = (ax + 2) % 3;
  double poly[2][8][3]; int np = 4, cur = 0;
  for (int c = 0; c < 4; c++) {
    double vert[3], rel[3];
    double a1 = (c == 0 || c == 3) ? 1 : -1, a2 = (c < 2) ? 1 : -1;
    for (int t = 0; t < 3; t++) vert[t] = pi[t] + fs * si[k] * Ri[k][t] + a1 * si[k1] * Ri[k1][t] + a2 * si[k2] * Ri[k2][t];
    sub3(rel, vert, pr);
    poly[0][c][0] = dot3(rel, Rr[u]); poly[0][c][1] = dot3(rel, Rr[v]); poly[0][c][2] = sgn * dot3(rel, Rr[ax]);
  }
  for (int e = 0; e < 4 && np > 0; e++) { /* clip against  +u, -u, +v, -v  */
    const int cdim = e / 2; const double sg = (e % 2) ? -1 : 1, lim = cdim == 0 ? sr[u] : sr[v];
    int nn = 0;
    for (int a = 0; a < np; a++) {
      const double* P = poly[cur][a]; const double* Q = poly[cur][(a + 1) % np];
      double dp = lim - sg * P[cdim], dq = lim - sg * Q[cdim]; /* >= 0 inside */
      if (dp >= 0 && nn < 8) { copy3(poly[1 - cur][nn], P); nn++; }
      if ((dp >= 0) != (dq >= 0) && nn < 8) { double tt = dp / (dp - dq); for (int t = 0; t < 3; t++) poly[1 - cur][nn][t] = P[t] + tt * (Q[t] - P[t]); nn++; }
    }
    np = nn; cur = 1 - cur;
  }
  /* candidates within the margin of the reference face; keep the (up to) 4 deepest, in polygon order */
  double dist[8]; int keep[8], nk = 0;
  for (int a = 0; a < np; a++) { dist[a] = poly[cur][a][2] - sr[ax]; keep[a] = dist[a] < margin; nk += keep[a]; }
  np_tally(nk == 0 ? 2 : nk <= 4 ? 3 : 4);
  while (nk > 4) { int worst = -1; for (int a = 0; a < np; a++) if (keep[a] && (worst < 0 || dist[a] > dist[worst])) worst = a; keep[worst] = 0; nk--; }
  int cnt = 0;
  for (int a = 0; a < np; a++) if (keep[a]) {
    double wv[3];
    for (int t = 0; t < 3; t++) wv[t] = pr[t] + poly[cur][a][0] * Rr[u][t] + poly[cur][a][1] * Rr[v][t] + sgn * poly[cur][a][2] * Rr[ax][t];
    con[cnt].dist = dist[a];
    for (int t = 0; t < 3; t++) { con[cnt].frame[t] = refB ? -n[t] : n[t]; con[cnt].frame[3 + t] = 0; con[cnt].pos[t] = wv[t] - n[t] * dist[a] / 2; }
    cnt++;
  }
  return cnt;
}

static int narrowphase(const dmo_model* m, const dmo_data* d, int g1, int g2, double margin, rawcon* con) {
  const dmo_spec* s = &m->s;
  int t1 = s->geom_type[g1], t2 = s->geom_type[g2];
  const double *p1 = d->geom_xpos[g1], *p2 = d->geom_xpos[g2], *m1 = d->geom_xmat[g1], *m2 = d->geom_xmat[g2];
  const double *s1 = m->geom_lsize[g1], *s2 = m->geom_lsize[g2];
  if (t1 == DMO_GEOM_PLANE) {
    double n[3] = {m1[2], m1[5], m1[8]};
    if (t2 == DMO_GEOM_SPHERE) return plane_sphere(con, p1, n, p2, s2[0], margin);
    if (t2 == DMO_GEOM_CAPSULE) { /* [MJ mjc_PlaneCapsule]: the two end spheres, +axis end first; tangent aligned with the axis */
      double ax[3] = {m2[2], m2[5], m2[8]}, c[3];
      int n0 = 0;
      addscl3(c, p2, ax, s2[1]);
      n0 += plane_sphere(con + n0, p1, n, c, s2[0], margin);
      addscl3(c, p2, ax, -s2[1]);
      n0 += plane_sphere(con + n0, p1, n, c, s2[0], margin);
      for (int i = 0; i < n0; i++) copy3(con[i].frame + 3, ax);
      return n0;
    }
    if (t2 == DMO_GEOM_BOX) { /* [MJ mjc_PlaneBox]: corners below margin, at most 4 */
      double dif[3]; sub3(dif, p2, p1);
      double dist = dot3(dif, n);
      int cnt = 0;
      for (int i = 0; i < 8; i++) {
        double vec[3] = {(i & 1 ? s2[0] : -s2[0]), (i & 2 ? s2[1] : -s2[1]), (i & 4 ? s2[2] : -s2[2])}, corner[3];
        mat_vec(corner, m2, vec);
        double ldist = dot3(n, corner);
        if (dist + ldist > margin || ldist > 0) continue;
        con[cnt].dist = dist + ldist;
        copy3(con[cnt].frame, n); zero3(con[cnt].frame + 3);
        add3(corner, corner, p2);
        addscl3(con[cnt].pos, corner, n, -con[cnt].dist / 2);
        if (++cnt >= 4) return 4;
      }
      return cnt;
    }
    return 0;
  }
  if (t1 == DMO_GEOM_SPHERE && t2 == DMO_GEOM_SPHERE) return sphere_sphere(con, p1, s1[0], p2, s2[0], margin);
  if (t1 == DMO_GEOM_SPHERE && t2 == DMO_GEOM_CAPSULE) { /* [MJ mjc_SphereCapsule] nearest point on the segment */
    double ax[3] = {m2[2], m2[5], m2[8]}, v[3], c[3];
    sub3(v, p1, p2);
    double x = clampd(dot3(ax, v), -s2[1], s2[1]);
    addscl3(c, p2, ax, x);
    return sphere_sphere(con, p1, s1[0], c, s2[0], margin);
  }
  if (t1 == DMO_GEOM_CAPSULE && t2 == DMO_GEOM_CAPSULE) { /* [MJ mjc_CapsuleCapsule] closest points of two segments */
    double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]}, dif[3];
    sub3(dif, p1, p2);
    double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2), u = -dot3(a1, dif), v = dot3(a2, dif);
    double det = ma * mc - mb * mb, x1, x2, c1[3], c2[3];
    if (fabs(det) >= MINVAL) {
      x1 = (mc * u - mb * v) / det; x2 = (ma * v - mb * u) / det;
      if (x1 > s1[1]) { x1 = s1[1]; x2 = (v - mb * s1[1]) / mc; } else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = (v + mb * s1[1]) / mc; }
      if (x2 > s2[1]) { x2 = s2[1]; x1 = (u - mb * s2[1]) / ma; } else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = (u + mb * s2[1]) / ma; }
      if (x1 > s1[1]) x1 = s1[1]; else if (x1 < -s1[1]) x1 = -s1[1];
      addscl3(c1, p1, a1, x1); addscl3(c2, p2, a2, x2);
      return sphere_sphere(con, c1, s1[0], c2, s2[0], margin);
    }
    /* parallel axes: test both ends of capsule 1 against segment 2, then both ends of 2 against 1 (max 2 contacts) */
    int n0 = 0;
    for (int e = 0; e < 2 && n0 < 2; e++) {
      double t[3];
      addscl3(c1, p1, a1, e == 0 ? s1[1] : -s1[1]);
      sub3(t, c1, p2); x2 = clampd(dot3(t, a2), -s2[1], s2[1]);
      addscl3(c2, p2, a2, x2);
      n0 += sphere_sphere(con + n0, c1, s1[0], c2, s2[0], margin);
    }
    for (int e = 0; e < 2 && n0 < 2; e++) {
      double t[3];
      addscl3(c2, p2, a2, e == 0 ? s2[1] : -s2[1]);
      sub3(t, c2, p1); x1 = clampd(dot3(t, a1), -s1[1], s1[1]);
      addscl3(c1, p1, a1, x1);
      n0 += sphere_sphere(con + n0, c1, s1[0], c2, s2[0], margin);
    }
    return n0;
  }
  if (t1 == DMO_GEOM_SPHERE && t2 == DMO_GEOM_BOX) { /* [MJ mjc_SphereBox] */
    double t[3], center[3], clamped[3], nrm[3], pl[3];
    sub3(t, p1, p2); matT_vec(center, m2, t);
    for (int i = 0; i < 3; i++) clamped[i] = clampd(center[i], -s2[i], s2[i]);
    sub3(t, center, clamped);
    double dist = norm3(t);
    if (dist - s1[0] > margin) return 0;
    if (dist <= MINVAL) { /* centre inside the box: push out through the closest face */
      double closest = 2 * fmax(s2[0], fmax(s2[1], s2[2])); int k = 0;
      for (int i = 0; i < 6; i++) { double fd = fabs((i % 2 ? 1 : -1) * s2[i / 2] - center[i / 2]); if (closest > fd) { closest = fd; k = i; } }
      zero3(nrm); nrm[k / 2] = (k % 2 ? -1 : 1);
      addscl3(pl, center, nrm, (s1[0] - closest) / 2);
      con->dist = -closest - s1[0];
    } else {
      for (int i = 0; i < 3; i++) nrm[i] = -t[i] / dist; /* from sphere centre towards the box */
      addscl3(pl, center, nrm, s1[0] + 0.5 * (dist - s1[0]));
      con->dist = dist - s1[0];
    }
    mat_vec(con->frame, m2, nrm); zero3(con->frame + 3);
    mat_vec(con->pos, m2, pl); add3(con->pos, con->pos, p2);
    return 1;
  }
  if (t1 == DMO_GEOM_CAPSULE && t2 == DMO_GEOM_BOX) {
    /* OWN ALGORITHM (MuJoCo's mjc_CapsuleBox is a long case analysis that is not restated): the point of the capsule
     * segment closest to the box, then one sphere-box contact [MJ mjc_SphereBox] there.  In the box frame the squared
     * distance of c0 + t u to the box is convex and piecewise quadratic in t; its half-derivative
     *     g(t) = sum_k u_k (p_k - clamp(p_k, -s_k, s_k))
     * is piecewise linear and non-decreasing, with breakpoints where a coordinate crosses a face plane (at most 6).  The
     * zero of g is bracketed between consecutive breakpoints inside [-L, L] and found by linear interpolation — exact up to
     * rounding, no iteration.  The HIP kernel runs the identical sequence of operations. */

    double ax[3] = {m1[2], m1[5], m1[8]}, t[3], c0[3], u[3];
    sub3(t, p1, p2); matT_vec(c0, m2, t); matT_vec(u, m2, ax);
#define SEGBOX_G(tt, out) do { double g_ = 0; for (int k_ = 0; k_ < 3; k_++) { double pk_ = c0[k_] + (tt) * u[k_]; \
      g_ += u[k_] * (pk_ - clampd(pk_, -s2[k_], s2[k_])); } (out) = g_; } while (0)
    /* Where the segment runs through the INSIDE of the box, g is zero on a whole interval and its computed value at the interval's
     * ends (a face crossing) is +-1 ulp with a sign that depends on rounding: a bracket driven by `g <= 0` would then pick one end or
     * the other at random (and with it the face the contact is pushed out through).  So values within eps of zero are a set of
     * their own: if any sample point (the two ends, the face crossings inside the segment) lies in it, the answer is the MIDDLE of
     * that set's extent — the root itself when it falls on a sample point, the middle of the zero plateau otherwise; only when no
     * sample is numerically zero is the root bracketed between the neighbouring samples and interpolated (the generic case). */
    const double L = s1[1], eps = 1e-12;
    double ta = -L, tb = L, ga, gb, ts, center[3], clamped[3], nrm[3], pl[3];
    double z0 = 1e300, z1 = -1e300;
    SEGBOX_G(ta, ga); SEGBOX_G(tb, gb);
    const double g_lo = ga, g_hi = gb;
    if (fabs(ga) <= eps) { z0 = ta; z1 = ta; }
    if (fabs(gb) <= eps) { if (tb < z0) z0 = tb; if (tb > z1) z1 = tb; }
    for (int k = 0; k < 3; k++) for (int sg = 0; sg < 2; sg++) {
      if (fabs(u[k]) <= 1e-12) continue;
      double tc = ((sg ? s2[k] : -s2[k]) - c0[k]) / u[k], gc;
      if (!(tc > -L && tc < L)) continue;
      SEGBOX_G(tc, gc);
      if (fabs(gc) <= eps) { if (tc < z0) z0 = tc; if (tc > z1) z1 = tc; }
      else if (gc < 0) { if (tc > ta) { ta = tc; ga = gc; } }
      else { if (tc < tb) { tb = tc; gb = gc; } }
    }
    if (z0 <= z1) ts = z0 > 0 ? z0 : (z1 < 0 ? z1 : 0.0);   /* the point of the zero set nearest the capsule's centre */
    else if (g_lo > 0) ts = -L;
    else if (g_hi < 0) ts = L;
    else ts = (gb - ga > 1e-300) ? ta - ga * (tb - ta) / (gb - ga) : 0.5 * (ta + tb);
#undef SEGBOX_G
    for (int k = 0; k < 3; k++) { center[k] = c0[k] + ts * u[k]; clamped[k] = clampd(center[k], -s2[k], s2[k]); t[k] = center[k] - clamped[k]; }
    double dist = norm3(t);
    if (dist - s1[0] > margin) { np_tally(5); return 0; }
    if (g_np_diag) {   /* (diagnostics only) how many ends of the axis segment lie within radius + margin of the box */
      int ends_in = 0;
      for (int sg = -1; sg <= 1; sg += 2) {
        double e2 = 0;
        for (int k = 0; k < 3; k++) { double pk = c0[k] + sg * L * u[k], dk = pk - clampd(pk, -s2[k], s2[k]); e2 += dk * dk; }
        if (sqrt(e2) - s1[0] <= margin) ends_in++;
      }
      np_tally(dist <= MINVAL ? 9 : 6 + ends_in);
    }
    if (dist <= MINVAL) {
      double closest = 2 * fmax(s2[0], fmax(s2[1], s2[2])); int kk = 0;
      for (int i = 0; i < 6; i++) { double fd = fabs((i % 2 ? 1 : -1) * s2[i / 2] - center[i / 2]); if (closest > fd) { closest = fd; kk = i; } }
      zero3(nrm); nrm[kk / 2] = (kk % 2 ? -1 : 1);
      addscl3(pl, center, nrm, (s1[0] - closest) / 2);
      con->dist = -closest - s1[0];
    } else {
      for (int i = 0; i < 3; i++) nrm[i] = -t[i] / dist;
      addscl3(pl, center, nrm, s1[0] + 0.5 * (dist - s1[0]));
      con->dist = dist - s1[0];
    }
    mat_vec(con->frame, m2, nrm); zero3(con->frame + 3);
    mat_vec(con->pos, m2, pl); add3(con->pos, con->pos, p2);
    return 1;
  }
  if (t1 == DMO_GEOM_BOX && t2 == DMO_GEOM_BOX) return box_box(con, p1, m1, s1, p2, m2, s2, margin);
  return 0;
}
/* [MJ mju_makeFrame] complete (normal, tangent hint) into a right-handed orthonormal frame, rows = axes */
static void make_frame(double* f) {
  double t[3];
  normalize3(f);
  if (norm3(f + 3) < 0.5) { zero3(f + 3); if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1; }
  double dp = dot3(f, f + 3);
  t[0] = f[0] * dp; t[1] = f[1] * dp; t[2] = f[2] * dp;
  sub3(f + 3, f + 3, t);
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}
/* [MJ mj_collision]: walk the precompiled pair list; contact parameters by max / mix [MJ mj_contactParam] */
static void collision(const dmo_model* m, dmo_data* d) {
  const dmo_spec* s = &m->s;
  d->ncon = 0;
  if (!m->enable_contact) return;
  for (int k = 0; k < m->npair; k++) {
    int g1 = m->pair_g1[k], g2 = m->pair_g2[k];
    double margin = fmax(s->geom_margin[g1], s->geom_margin[g2]);
    rawcon rc[8];
    int n = narrowphase(m, d, g1, g2, margin, rc);
    for (int i = 0; i < n && d->ncon < DMO_MAXCON; i++) {
      dmo_contact* c = &d->contact[d->ncon++];
      c->geom1 = g1; c->geom2 = g2;
      c->dist = rc[i].dist; copy3(c->pos, rc[i].pos);
      memcpy(c->frame, rc[i].frame, 6 * sizeof(double));
      make_frame(c->frame);
      c->includemargin = margin; /* gap = 0 */
      c->dim = s->geom_condim[g1] > s->geom_condim[g2] ? s->geom_condim[g1] : s->geom_condim[g2];
      double f0 = fmax(s->geom_friction[g1][0], s->geom_friction[g2][0]), f1 = fmax(s->geom_friction[g1][1], s->geom_friction[g2][1]), f2 = fmax(s->geom_friction[g1][2], s->geom_friction[g2][2]);
      c->friction[0] = f0; c->friction[1] = f0; c->friction[2] = f1; c->friction[3] = f2; c->friction[4] = f2;
    }
  }
}

/* ======================================= constraints ========================================== */
/* [MJ getimpedance], 5-parameter solimp */
static double impedance(const double* si, double x) {
  if (si[0] == si[1] || si[2] <= MINVAL) return 0.5 * (si[0] + si[1]);
  x = fabs(x / si[2]);
  if (x >= 1) return si[1];
  if (x <= 0) return si[0];
  double y;
  if (si[4] == 1) y = x;
  else if (x <= si[3]) y = pow(x, si[4]) / pow(si[3], si[4] - 1);
  else y = 1 - pow(1 - x, si[4]) / pow(1 - si[3], si[4] - 1);
  return si[0] + y * (si[1] - si[0]);
}
/* [MJ mj_makeConstraint + mj_diagApprox + mj_makeImpedance + mj_referenceConstraint + mj_projectConstraint] */
static void make_constraint(const dmo_model* m, dmo_data* d) {
  const dmo_spec* s = &m->s;
  int nv = m->nv, n = 0;
  /* joint limits, joint order, lower side first */
  if (m->enable_limit) for (int j = 0; j < s->njnt; j++) if (s->jnt_limited[j] && s->jnt_type[j] == DMO_JNT_HINGE) {
    double value = d->qpos[m->jnt_qposadr[j]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (s->jnt_range[j][(side + 1) / 2] - value);
      if (dist < 0 && n < m->max_efc) { /* jnt_margin = 0 */
        for (int k = 0; k < nv; k++) d->efc_J[n][k] = 0;
        d->efc_J[n][m->jnt_dofadr[j]] = -(double)side;
        d->efc_pos[n] = dist; d->efc_margin[n] = 0;
        d->efc_diagApprox[n] = m->dof_invweight0[m->jnt_dofadr[j]];
        n++;
      }
    }
  }
  d->nlimit = n;
  /* contacts: condim 1 -> one normal row; condim 3 pyramidal -> (Jn +- mu Jt1), (Jn +- mu Jt2) */
  for (int ci = 0; ci < d->ncon; ci++) {
    const dmo_contact* c = &d->contact[ci];
    int b1 = s->geom_body[c->geom1], b2 = s->geom_body[c->geom2];
    double j1[3][DMO_MAXV], j2[3][DMO_MAXV], jc[3][DMO_MAXV];
    jac_point(m, d, b1, c->pos, j1, NULL); jac_point(m, d, b2, c->pos, j2, NULL);
    for (int r = 0; r < 3; r++) for (int k = 0; k < nv; k++) {
      jc[r][k] = 0;
      for (int a = 0; a < 3; a++) jc[r][k] += c->frame[3 * r + a] * (j2[a][k] - j1[a][k]);
    }
    double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
    if (c->dim == 1) {
      if (n >= m->max_efc) break;
      for (int k = 0; k < nv; k++) d->efc_J[n][k] = jc[0][k];
      d->efc_pos[n] = c->dist; d->efc_margin[n] = c->includemargin; d->efc_diagApprox[n] = tran; n++;
    } else {
      if (n + 2 * (c->dim - 1) > m->max_efc) break;
      for (int t = 1; t < c->dim; t++) for (int sg = 0; sg < 2; sg++) {
        double mu = c->friction[t - 1];
        for (int k = 0; k < nv; k++) d->efc_J[n][k] = jc[0][k] + (sg == 0 ? mu : -mu) * jc[t][k];
        d->efc_pos[n] = c->dist; d->efc_margin[n] = c->includemargin;
        d->efc_diagApprox[n] = m->pyramid_diag_mu2 ? tran + mu * mu * tran : tran;
        n++;
      }
    }
  }
  d->nefc = n;
  /* impedance, regulariser, reference acceleration */
  double tc = fmax(s->solref[0], 2 * s->timestep) /* refsafe */, dr = s->solref[1], dmax = s->solimp[1];
  double K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr), B = 2 / fmax(MINVAL, dmax * tc);
  for (int i = 0; i < n; i++) {
    double imp = impedance(s->solimp, d->efc_pos[i] - d->efc_margin[i]);
    d->efc_R[i] = fmax(MINVAL, (1 - imp) * d->efc_diagApprox[i] / imp);
    d->efc_KBI[i][0] = K; d->efc_KBI[i][1] = B; d->efc_KBI[i][2] = imp;
  }
  if (m->pyramid_r_rescale) { /* pyramidal edges: R <- 2 mu^2 R(first edge) for all 2(dim-1) rows of the contact */
    int i = d->nlimit;
    for (int ci = 0; ci < d->ncon && i < n; ci++) {
      const dmo_contact* c = &d->contact[ci];
      if (c->dim == 1) { i++; continue; }
      double Rpy = fmax(MINVAL, 2 * c->friction[0] * c->friction[0] * d->efc_R[i]);
      for (int k = 0; k < 2 * (c->dim - 1); k++) d->efc_R[i + k] = Rpy;
      i += 2 * (c->dim - 1);
    }
  }
  for (int i = 0; i < n; i++) {
    double v = 0;
    for (int k = 0; k < nv; k++) v += d->efc_J[i][k] * d->qvel[k];
    d->efc_vel[i] = v;
    d->efc_aref[i] = -d->efc_KBI[i][1] * v - d->efc_KBI[i][0] * d->efc_KBI[i][2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
  /* AR = J M^-1 J^T + diag(R) */
  static __thread double X[DMO_MAXEFC][DMO_MAXV];
  for (int i = 0; i < n; i++) { for (int k = 0; k < nv; k++) X[i][k] = d->efc_J[i][k]; solve_M(m, d, X[i]); }
  for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) {
    double a = 0;
    for (int k = 0; k < nv; k++) a += d->efc_J[i][k] * X[j][k];
    d->efc_AR[i][j] = a; d->efc_AR[j][i] = a;
  }
  for (int i = 0; i < n; i++) d->efc_AR[i][i] += d->efc_R[i];
}

/* [MJ mj_fwdConstraint: warmstart + mj_solPGS + dual finish] */
static void solve_constraint(const dmo_model* m, dmo_data* d) {
  const dmo_spec* s = &m->s;
  int nv = m->nv, n = d->nefc;
  d->solver_iter = 0; d->solver_improvement = 0;
  if (n == 0) { for (int k = 0; k < nv; k++) { d->qacc[k] = d->qacc_smooth[k]; d->qfrc_constraint[k] = 0; } return; }
  for (int i = 0; i < n; i++) {
    double b = 0, jar = 0;
    for (int k = 0; k < nv; k++) { b += d->efc_J[i][k] * d->qacc_smooth[k]; jar += d->efc_J[i][k] * d->qacc_warmstart[k]; }
    d->efc_b[i] = b - d->efc_aref[i];
    jar -= d->efc_aref[i];
    d->efc_force[i] = jar < 0 ? -jar / d->efc_R[i] : 0; /* [MJ mj_constraintUpdate], all rows are unilateral */
  }
  /* dual cost of the warm start; fall back to zero forces if it is not an improvement over f = 0 */
  double cost = 0;
  for (int i = 0; i < n; i++) {
    double r = 0;
    for (int j = 0; j < n; j++) r += d->efc_AR[i][j] * d->efc_force[j];
    cost += d->efc_force[i] * (0.5 * r + d->efc_b[i]);
  }
  if (cost > 0) for (int i = 0; i < n; i++) d->efc_force[i] = 0;
  /* projected Gauss-Seidel sweeps in row order */
  double scale = 1 / (m->meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  while (iter < s->iterations) {
    double improvement = 0;
    for (int i = 0; i < n; i++) {
      double res = d->efc_b[i];
      for (int j = 0; j < n; j++) res += d->efc_AR[i][j] * d->efc_force[j];
      double old = d->efc_force[i];
      double f = old - res * (1.0 / d->efc_AR[i][i]); /* MuJoCo multiplies by the precomputed inverse diagonal */
      if (f < 0) f = 0;
      double delta = f - old, change = 0.5 * delta * delta * d->efc_AR[i][i] + delta * res;
      if (change > 1e-10) { f = old; change = 0; } /* [MJ costChange] never accept an increase */
      d->efc_force[i] = f;
      improvement -= change;
    }
    improvement *= scale;
    iter++;
    d->solver_improvement = improvement;
    if (improvement < s->tolerance) break;
  }
  d->solver_iter = iter;
  for (int k = 0; k < nv; k++) { double q = 0; for (int i = 0; i < n; i++) q += d->efc_J[i][k] * d->efc_force[i]; d->qfrc_constraint[k] = q; d->qacc[k] = q; }
  solve_M(m, d, d->qacc);
  for (int k = 0; k < nv; k++) d->qacc[k] += d->qacc_smooth[k];
}

/* ================================ velocity / force stages ===================================== */
/* [MJ mj_comVel + mj_rne(flg_acc=0)] bias forces C(q,v) incl. gravity */
static void rne_bias(const dmo_model* m, dmo_data* d) {
  const dmo_spec* s = &m->s;
  double cvel[DMO_MAXBODY][6], cacc[DMO_MAXBODY][6], cfrc[DMO_MAXBODY][6];
  memset(cvel[0], 0, sizeof cvel[0]); memset(cacc[0], 0, sizeof cacc[0]); memset(cfrc[0], 0, sizeof cfrc[0]);
  cacc[0][3] = -s->gravity[0]; cacc[0][4] = -s->gravity[1]; cacc[0][5] = -s->gravity[2];
  for (int b = 1; b < s->nbody; b++) {
    int p = s->body_parent[b];
    double v[6], a[6];
    memcpy(v, cvel[p], sizeof v); memcpy(a, cacc[p], sizeof a);
    int da = m->body_dofadr[b], nd = m->body_dofnum[b], j0 = m->body_jntadr[b];
    if (nd == 6 && s->jnt_type[j0] == DMO_JNT_FREE) {
      /* translations: constant axes; rotations: all three cdof_dot use the velocity BEFORE the rotation is added */
      for (int k = 0; k < 3; k++) for (int r = 0; r < 6; r++) v[r] += d->cdof[da + k][r] * d->qvel[da + k];
      double vb[6]; memcpy(vb, v, sizeof vb);
      for (int k = 3; k < 6; k++) {
        double cd[6]; cross_motion(cd, vb, d->cdof[da + k]);
        for (int r = 0; r < 6; r++) { a[r] += cd[r] * d->qvel[da + k]; v[r] += d->cdof[da + k][r] * d->qvel[da + k]; }
      }
    } else {
      for (int k = 0; k < nd; k++) {
        double cd[6]; cross_motion(cd, v, d->cdof[da + k]);
        for (int r = 0; r < 6; r++) { a[r] += cd[r] * d->qvel[da + k]; v[r] += d->cdof[da + k][r] * d->qvel[da + k]; }
      }
    }
    memcpy(cvel[b], v, sizeof v); memcpy(cacc[b], a, sizeof a);
    sinert si; body_sinert(m, d, b, &si);
    double Ia[6], Iv[6], vxIv[6];
    sinert_mul(Ia, &si, a); sinert_mul(Iv, &si, v); cross_force(vxIv, v, Iv);
    for (int r = 0; r < 6; r++) cfrc[b][r] = Ia[r] + vxIv[r];
  }
  for (int b = s->nbody - 1; b > 0; b--) { int p = s->body_parent[b]; if (p > 0) for (int r = 0; r < 6; r++) cfrc[p][r] += cfrc[b][r]; }
  for (int i = 0; i < m->nv; i++) d->qfrc_bias[i] = dot6(d->cdof[i], cfrc[m->dof_body[i]]);
}

/* [MJ mj_forward] = fwdPosition, fwdVelocity, fwdActuation, fwdAcceleration, fwdConstraint */
void dmo_forward(const dmo_model* m, dmo_data* d) {
  const dmo_spec* s = &m->s;
  int nv = m->nv;
  kinematics(m, d);
  mass_matrix(m, d);
  collision(m, d);
  make_constraint(m, d);
  rne_bias(m, d);
  for (int i = 0; i < nv; i++) { d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i]; d->qfrc_actuator[i] = 0; }
  for (int u = 0; u < s->nu; u++) { /* motor: force = gear * clamp(ctrl); data.ctrl itself stays unclamped */
    double c = clampd(d->ctrl[u], s->act_ctrlrange[u][0], s->act_ctrlrange[u][1]);
    d->qfrc_actuator[m->jnt_dofadr[s->act_jnt[u]]] += s->act_gear[u] * c;
  }
  for (int i = 0; i < nv; i++) d->qacc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  solve_M(m, d, d->qacc_smooth);
  solve_constraint(m, d);
}

/* [MJ mj_integratePos] */
static void integrate_pos(const dmo_model* m, double* qpos, const double* qvel, double h) {
  const dmo_spec* s = &m->s;
  for (int j = 0; j < s->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (s->jnt_type[j] == DMO_JNT_FREE) {
      for (int k = 0; k < 3; k++) qpos[qa + k] += h * qvel[da + k];
      quat_integrate(qpos + qa + 3, qvel + da + 3, h);
    } else qpos[qa] += h * qvel[da];
  }
}

/* [MJ mj_step with integrator RK4]: mj_forward, then mj_RungeKutta(4); qacc_warmstart <- qacc of the last stage.
 * The derived quantities left in `d` (xipos, contacts, ...) are those of the 4th stage evaluation, exactly what
 * sim.data holds after sim.step() — is_done() reads that xipos (src/dp_env_v3.py:134-139). */
void dmo_step(const dmo_model* m, dmo_data* d) {
  static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1}}, Bw[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
  int nq = m->nq, nv = m->nv;
  double h = m->s.timestep, t0 = d->time;
  double X[4][DMO_MAXQ + DMO_MAXV], F[4][DMO_MAXV], dX[2 * DMO_MAXV];
  dmo_forward(m, d);
  memcpy(X[0], d->qpos, nq * sizeof(double)); memcpy(X[0] + nq, d->qvel, nv * sizeof(double));
  memcpy(F[0], d->qacc, nv * sizeof(double));
  for (int i = 1; i < 4; i++) {
    double C = 0;
    for (int j = 0; j < i; j++) C += A[i - 1][j];
    memset(dX, 0, sizeof dX);
    for (int j = 0; j < i; j++) for (int k = 0; k < nv; k++) { dX[k] += A[i - 1][j] * X[j][nq + k]; dX[nv + k] += A[i - 1][j] * F[j][k]; }
    memcpy(X[i], X[0], (nq + nv) * sizeof(double));
    integrate_pos(m, X[i], dX, h);
    for (int k = 0; k < nv; k++) X[i][nq + k] += h * dX[nv + k];
    memcpy(d->qpos, X[i], nq * sizeof(double)); memcpy(d->qvel, X[i] + nq, nv * sizeof(double));
    d->time = t0 + C * h;
    dmo_forward(m, d);
    memcpy(F[i], d->qacc, nv * sizeof(double));
  }
  memset(dX, 0, sizeof dX);
  for (int j = 0; j < 4; j++) for (int k = 0; k < nv; k++) { dX[k] += Bw[j] * X[j][nq + k]; dX[nv + k] += Bw[j] * F[j][k]; }
  memcpy(d->qpos, X[0], nq * sizeof(double));
  for (int k = 0; k < nv; k++) d->qvel[k] = X[0][nq + k] + h * dX[nv + k];
  integrate_pos(m, d->qpos, dX, h);
  d->time = t0 + h;
  memcpy(d->qacc_warmstart, d->qacc, nv * sizeof(double));
}

/* ========================================= env layer ========================================== */
/* src/dp_env_v3.py:62-65: obs = qpos[7:] (+) qvel[6:] */
void dmo_get_obs(const dmo_model* m, const dmo_data* d, double* obs) {
  int k = 0;
  for (int i = 7; i < m->nq; i++) obs[k++] = d->qpos[i];
  for (int i = 6; i < m->nv; i++) obs[k++] = d->qvel[i];
}
/* src/dp_env_v3.py:134-139: z of sum(mass * xipos) / sum(mass) over ALL bodies (world has mass 0) */
double dmo_com_z(const dmo_model* m, const dmo_data* d) {
  double sz = 0, sm = 0;
  for (int b = 0; b < m->s.nbody; b++) { sz += m->body_mass[b] * d->xipos[b][2]; sm += m->body_mass[b]; }
  return sz / sm;
}
int dmo_is_done(const dmo_model* m, const dmo_data* d) { double z = dmo_com_z(m, d); return (z < 0.7) || (z > 2.0); }
/* gym MujocoEnv.set_state (called at src/dp_env_v3.py:153,160): qpos, qvel replaced; time / warmstart kept; sim.forward() */
void dmo_set_state(const dmo_model* m, dmo_data* d, const double* qpos, const double* qvel) {
  memcpy(d->qpos, qpos, m->nq * sizeof(double)); memcpy(d->qvel, qvel, m->nv * sizeof(double));
  dmo_forward(m, d);
}
/* src/dp_env_v3.py:85-104: exp(-sum|qpos[7:] - data_config[idx_curr][7:]|); idx_curr <- (idx_curr+1) % F */
double dmo_config_reward(const dmo_model* m, const dmo_data* d, const double* cfg, int F, int* idx_curr) {
  const double* tgt = cfg + (size_t)(*idx_curr) * m->nq;
  double err = 0;
  for (int i = 7; i < m->nq; i++) err += fabs(d->qpos[i] - tgt[i]);
  *idx_curr = (*idx_curr + 1) % F;
  return exp(-err);
}
void dmo_env_step(const dmo_model* m, dmo_data* d, const double* action, int n_substeps, int reward_mode,
                  const double* cfg, int F, int* idx_curr, int idx_init, double* obs, double* reward, int* done) {
  for (int u = 0; u < m->s.nu; u++) d->ctrl[u] = action[u];                 /* do_simulation: data.ctrl[:] = ctrl */
  for (int k = 0; k < n_substeps; k++) dmo_step(m, d);                      /* src/dp_env_v3.py:108-112 (1 substep) */
  dmo_get_obs(m, d, obs);
  if (reward_mode == DMO_REW_ALIVE) *reward = 1.0;                          /* src/dp_env_v3.py:117,128 */
  else if (reward_mode == DMO_REW_V3_CONFIG) *reward = dmo_config_reward(m, d, cfg, F, idx_curr); /* :127 (disabled upstream) */
  else { /* src/dp_env_v2.py:116-183: idx_curr += 1; exp(-2 * sum|qpos[3:] - cfg[(idx_curr+idx_init)%F][3:]|) - 0.1 sum ctrl^2 */
    *idx_curr += 1;
    int im = (*idx_curr + idx_init) % F;
    const double* tgt = cfg + (size_t)im * m->nq;
    double err = 0, acs = 0;
    for (int i = 3; i < m->nq; i++) err += fabs(d->qpos[i] - tgt[i]);
    for (int u = 0; u < m->s.nu; u++) acs += d->ctrl[u] * d->ctrl[u];
    *reward = exp(-1.0 * 2.0 * err) - 0.1 * acs;
  }
  *done = dmo_is_done(m, d);
}


/* ============================ 5-term imitation reward (code.md:1017-1143) ============================
 * cSceneImitate::CalcRewardImitate as quoted in the reference's porting notes; the reference itself never implemented it
 * (src/dp_env_v3.py:117-128 returns 1.0).  A feature row (112 doubles, layout in deepmimic_mujoco_amd/imitation.py) of the
 * simulated state is compared with the row of the mocap frame.  params[32]: joint weights [12] (model body order 2..13),
 * root weight, cycle shift x y, loop flag, end-effector body ids [4], offsets [4][3].  Items marked [upstream] come from
 * upstream DeepMimic (KinTree / MathUtil), not from the reference's files. */
#define DMO_FEAT 112
static double quat_diff_theta(const double* q0, const double* q1) {    /* [upstream] cMathUtil::QuatDiffTheta */
  const double c[4] = {q0[0], -q0[1], -q0[2], -q0[3]};
  double dq[4];
  quat_mul(dq, q1, c);
  double w = dq[0] > 1 ? 1 : (dq[0] < -1 ? -1 : dq[0]);
  if (sqrt(fmax(0.0, 1 - w * w)) <= 1e-6) return 0;
  double th = 2 * acos(w);
  return th > M_PI ? th - 2 * M_PI : th;
}
void dmo_imitation_features(const dmo_model* m, const double* qpos, const double* qvel, const double* params, double* f) {
  const dmo_spec* sp = &m->s;
  dmo_data* d = dmo_data_create(m);          /* scratch: kinematics of THIS state, the caller's derived quantities stay untouched */
  for (int i = 0; i < m->nq; i++) d->qpos[i] = qpos[i];
  for (int i = 0; i < m->nv; i++) d->qvel[i] = qvel[i];
  kinematics(m, d);
  for (int i = 0; i < DMO_FEAT; i++) f[i] = 0;
  double rq[4] = {qpos[3], qpos[4], qpos[5], qpos[6]};
  normalize4(rq);
  for (int k = 0; k < 3; k++) { f[k] = qpos[k]; f[7 + k] = qvel[k]; }
  for (int k = 0; k < 4; k++) f[3 + k] = rq[k];
  quat_rot(f + 10, rq, qvel + 3);                                      /* free-joint angular velocity is body-local */
  for (int g = 0; g < 12; g++) {
    int b = g + 2, j0 = m->body_jntadr[b], nj = m->body_jntnum[b];
    if (nj == 1) { f[13 + 4 * g] = qpos[m->jnt_qposadr[j0]]; f[61 + 3 * g] = qvel[m->jnt_dofadr[j0]]; continue; }
    double ql[4] = {1, 0, 0, 0}, wl[3] = {0, 0, 0};
    for (int j = j0; j < j0 + nj; j++) {                               /* child = R1 R2 R3;  w = sum_k R1..R(k-1) a_k rate_k */
      double a[3], qa[4], t[4];
      quat_rot(a, ql, sp->jnt_axis[j]);
      for (int k = 0; k < 3; k++) wl[k] += a[k] * qvel[m->jnt_dofadr[j]];
      axisangle2quat(qa, sp->jnt_axis[j], qpos[m->jnt_qposadr[j]]);
      quat_mul(t, ql, qa);
      for (int k = 0; k < 4; k++) ql[k] = t[k];
    }
    for (int k = 0; k < 4; k++) f[13 + 4 * g + k] = ql[k];
    for (int k = 0; k < 3; k++) f[61 + 3 * g + k] = wl[k];
  }
  const double ex[3] = {1, 0, 0};
  double fwd[3];
  quat_rot(fwd, rq, ex);
  const double hd = atan2(fwd[1], fwd[0]), c = cos(hd), sn = sin(hd);  /* heading about the vertical */
  for (int e = 0; e < 4; e++) {
    int b = (int)params[16 + e];
    double p[3], rel[3];
    mat_vec(p, d->xmat[b], params + 20 + 3 * e);
    for (int k = 0; k < 3; k++) { p[k] += d->xpos[b][k]; rel[k] = p[k] - qpos[k]; }
    rel[2] = p[2];                                                     /* height above the ground plane */
    f[97 + 3 * e] = c * rel[0] + sn * rel[1]; f[98 + 3 * e] = -sn * rel[0] + c * rel[1]; f[99 + 3 * e] = rel[2];
  }
  double mom[3] = {0, 0, 0};                                           /* linear momentum: sum_b m_b (v_origin + w x xipos) */
  for (int b = 1; b < sp->nbody; b++) {
    double cv[6] = {0, 0, 0, 0, 0, 0};
    for (int bb = b; bb > 0; bb = sp->body_parent[bb])
      for (int dd = m->body_dofadr[bb]; dd < m->body_dofadr[bb] + m->body_dofnum[bb]; dd++)
        for (int k = 0; k < 6; k++) cv[k] += d->cdof[dd][k] * qvel[dd];
    double wxr[3];
    cross3(wxr, cv, d->xipos[b]);
    for (int k = 0; k < 3; k++) mom[k] += m->body_mass[b] * (cv[3 + k] + wxr[k]);
  }
  for (int k = 0; k < 3; k++) f[109 + k] = mom[k] / m->total_mass;
  dmo_data_destroy(d);
}
double dmo_imitation_reward(const dmo_model* m, const double* f0, const double* f1, const double* params, double shift_x,
                            double shift_y, double* terms) {
  const double th = quat_diff_theta(f0 + 3, f1 + 3);
  double dw2 = 0, dv2 = 0, dp2 = 0, dc2 = 0, de2 = 0;
  for (int k = 0; k < 3; k++) { double a = f1[10 + k] - f0[10 + k]; dw2 += a * a; }
  double pose = params[12] * th * th, vel = params[12] * dw2;
  for (int g = 0; g < 12; g++) {
    double pe;
    if (m->body_jntnum[g + 2] == 1) { double a = f1[13 + 4 * g] - f0[13 + 4 * g]; pe = a * a; }
    else { double t = quat_diff_theta(f0 + 13 + 4 * g, f1 + 13 + 4 * g); pe = t * t; }
    double ve = 0;
    for (int k = 0; k < 3; k++) { double a = f1[61 + 3 * g + k] - f0[61 + 3 * g + k]; ve += a * a; }
    pose += params[g] * pe; vel += params[g] * ve;
  }
  for (int k = 0; k < 12; k++) { double a = f1[97 + k] - f0[97 + k]; de2 += a * a; }
  const double p1[3] = {f1[0] + shift_x, f1[1] + shift_y, f1[2]};
  for (int k = 0; k < 3; k++) { double a = f0[k] - p1[k]; dp2 += a * a; a = f1[7 + k] - f0[7 + k]; dv2 += a * a; a = f1[109 + k] - f0[109 + k]; dc2 += a * a; }
  const double e[5] = {pose, vel, de2 / 4, dp2 + 0.1 * th * th + 0.01 * dv2 + 0.001 * dw2, 0.1 * dc2};
  static const double w[5] = {0.5, 0.05, 0.15, 0.2, 0.1}, sc[5] = {2, 0.1, 40, 5, 10};   /* code.md:1019-1037 (weights sum to 1) */
  double r = 0;
  for (int k = 0; k < 5; k++) { if (terms) terms[k] = e[k]; r += w[k] * exp(-sc[k] * e[k]); }
  return r;
}
/* dp_env_v1's reward (src/dp_env_v1.py:82-141) on the feature rows: weighted |quaternion-difference angle| pose error (JOINT_WEIGHT
 * un-normalised = params[g] / params[12]), L1 angular-rate error against the rates of row f1v, L1 root position error. */
double dmo_v1_reward(const dmo_model* m, const double* f0, const double* f1, const double* f1v, const double* params, double* terms) {
  double pose = fabs(quat_diff_theta(f0 + 3, f1 + 3)), vel = 0, root = 0;
  for (int k = 0; k < 3; k++) { vel += fabs(f1v[10 + k] - f0[10 + k]); root += fabs(f0[k] - f1[k]); }
  for (int g = 0; g < 12; g++) {
    double pe = m->body_jntnum[g + 2] == 1 ? fabs(f1[13 + 4 * g] - f0[13 + 4 * g]) : fabs(quat_diff_theta(f0 + 13 + 4 * g, f1 + 13 + 4 * g));
    pose += params[g] / params[12] * pe;
    for (int k = 0; k < 3; k++) vel += fabs(f1v[61 + 3 * g + k] - f0[61 + 3 * g + k]);
  }
  if (terms) { terms[0] = pose; terms[1] = vel; terms[2] = root; }
  return 0.5 * exp(-2.0 * pose) + 0.05 * exp(-0.1 * vel) + 0.2 * exp(-5.0 * root);
}
/* one env step with dp_env_v1's reward and cursor (src/dp_env_v1.py:143-158,88-96): idx_curr counts steps; the reward is
 * evaluated when idx_curr is a multiple of update_interval = int(mocap_dt // dt) (else 0), against frame (idx_curr // interval +
 * idx_init) % F, rates from the following frame; minus 0.1 sum ctrl^2. */
void dmo_env_step_v1(const dmo_model* m, dmo_data* d, const double* action, int n_substeps, const double* table, int F,
                     const double* params, double mocap_dt, int* idx_curr, int idx_init, double* obs, double* reward, int* done) {
  double acs = 0;
  for (int u = 0; u < m->s.nu; u++) { d->ctrl[u] = action[u]; acs += action[u] * action[u]; }
  for (int k = 0; k < n_substeps; k++) dmo_step(m, d);
  dmo_get_obs(m, d, obs);
  *idx_curr += 1;
  int upd = (int)floor(mocap_dt / (m->s.timestep * n_substeps));
  if (upd < 1) upd = 1;
  double robs = 0;
  if (*idx_curr % upd == 0) {
    const int k = (*idx_curr / upd + idx_init) % F, kv = k + 1 < F ? k + 1 : F - 1;
    double f0[DMO_FEAT];
    dmo_imitation_features(m, d->qpos, d->qvel, params, f0);
    robs = dmo_v1_reward(m, f0, table + (size_t)k * DMO_FEAT, table + (size_t)kv * DMO_FEAT, params, 0);
  }
  *reward = robs - 0.1 * acs;
  *done = dmo_is_done(m, d);
}
/* one env step in imitation mode: the state after the step is compared with frame idx_curr + 1 (wrapping clips add the
 * cycle shift per completed cycle; "Loop: none" clips hold the last frame and end the episode there). */
void dmo_env_step_imitation(const dmo_model* m, dmo_data* d, const double* action, int n_substeps, const double* table, int F,
                            const double* params, int* idx_curr, int* cycle, double* obs, double* reward, int* done) {
  for (int u = 0; u < m->s.nu; u++) d->ctrl[u] = action[u];
  for (int k = 0; k < n_substeps; k++) dmo_step(m, d);
  dmo_get_obs(m, d, obs);
  int k = *idx_curr + 1, ended = 0;
  if (k >= F) { if (params[15] != 0) { k = 0; *cycle += 1; } else { k = F - 1; ended = 1; } }
  *idx_curr = k;
  double f0[DMO_FEAT];
  dmo_imitation_features(m, d->qpos, d->qvel, params, f0);
  *reward = dmo_imitation_reward(m, f0, table + (size_t)k * DMO_FEAT, params, *cycle * params[13], *cycle * params[14], 0);
  *done = dmo_is_done(m, d) || ended;
}

void dmo_batch_step(const dmo_model* m, dmo_data** ds, int n, const double* actions, int n_substeps,
                    double* obs, double* reward, unsigned char* done, int nthreads) {
  int nu = m->s.nu;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int e = 0; e < n; e++) {
    int dn, idx = 0;
    dmo_env_step(m, ds[e], actions + (size_t)e * nu, n_substeps, DMO_REW_ALIVE, 0, 1, &idx, 0, obs + (size_t)e * 56, reward + e, &dn);
    done[e] = (unsigned char)dn;
  }
}

/* the same loop with the 5-term imitation reward (bench.py's cpu_baseline for the default workload) */
void dmo_batch_step_imitation(const dmo_model* m, dmo_data** ds, int n, const double* actions, int n_substeps, const double* table, int F,
                              const double* params, int* idx_curr, int* cycle, double* obs, double* reward, unsigned char* done, int nthreads) {
  int nu = m->s.nu;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int e = 0; e < n; e++) {
    int dn;
    dmo_env_step_imitation(m, ds[e], actions + (size_t)e * nu, n_substeps, table, F, params, idx_curr + e, cycle + e,
                           obs + (size_t)e * 56, reward + e, &dn);
    done[e] = (unsigned char)dn;
  }
}

/* bench.py's cpu_baseline, entirely in C: every env runs `steps` env-steps of the benchmark's workload — actions ~ N(0, sigma^2)
 * from a per-env xorshift64* stream (Box-Muller), the 5-term imitation reward when `table` is given (else the alive reward), RSI
 * reset to a uniformly drawn mocap frame on done.  Envs are independent, so OpenMP runs whole trajectories per thread (no
 * per-step barrier, nothing serial): the figure is the host's, not the harness'.  Returns the number of env-steps done;
 * *n_done = episodes ended, *reward_sum = sum of rewards (keeps the loop observable). */
static inline unsigned long long xs64(unsigned long long* s) {
  unsigned long long x = *s; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; *s = x; return x * 0x2545F4914F6CDD1DULL;
}
static inline double xs_uniform(unsigned long long* s) { return (double)(xs64(s) >> 11) * (1.0 / 9007199254740992.0); }
long dmo_bench_rollout(const dmo_model* m, dmo_data** ds, int n, int steps, const double* cfg, const double* vel, int F, const double* table,
                       const double* params, double sigma, unsigned long long seed, int nthreads, long* n_done, double* reward_sum) {
  const int nu = m->s.nu, nq = m->nq, nv = m->nv;
  long total = 0, dones = 0;
  double rsum = 0;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1) reduction(+ : total, dones, rsum)
#endif
  for (int e = 0; e < n; e++) {
    unsigned long long st = seed * 0x9E3779B97F4A7C15ULL + (unsigned long long)(e + 1) * 0xD1B54A32D192ED03ULL;
    if (!st) st = 1;
    dmo_data* d = ds[e];
    int idx = (int)(xs_uniform(&st) * F), cyc = 0;
    if (idx >= F) idx = F - 1;
    dmo_reset_data(m, d);
    dmo_set_state(m, d, cfg + (size_t)idx * nq, vel + (size_t)idx * nv);
    double act[DMO_MAXU], obs[64], rew;
    for (int t = 0; t < steps; t++) {
      for (int u = 0; u < nu; u += 2) {
        const double u1 = 1.0 - xs_uniform(&st), u2 = xs_uniform(&st);
        const double r = sqrt(-2.0 * log(u1)), a = 6.283185307179586 * u2;
        act[u] = sigma * r * cos(a);
        if (u + 1 < nu) act[u + 1] = sigma * r * sin(a);
      }
      int dn;
      if (table) dmo_env_step_imitation(m, d, act, 1, table, F, params, &idx, &cyc, obs, &rew, &dn);
      else { int ic = idx; dmo_env_step(m, d, act, 1, DMO_REW_ALIVE, 0, 1, &ic, 0, obs, &rew, &dn); }
      rsum += rew; total++;
      if (dn) {
        dones++;
        idx = (int)(xs_uniform(&st) * F); if (idx >= F) idx = F - 1; cyc = 0;
        dmo_reset_data(m, d);
        dmo_set_state(m, d, cfg + (size_t)idx * nq, vel + (size_t)idx * nv);
      }
    }
  }
  if (n_done) *n_done = dones;
  if (reward_sum) *reward_sum = rsum;
  return total;
}

int dmo_sizeof_model(void) { return (int)sizeof(dmo_model); }
int dmo_sizeof_data(void) { return (int)sizeof(dmo_data); }

/* ============================ string-keyed accessors for the ctypes test harness ================= */
dmo_model* dmo_model_new(const dmo_spec* s) {
  dmo_model* m = (dmo_model*)malloc(sizeof(dmo_model));
  dmo_spec hs;
  if (!s) { dmo_humanoid_spec(&hs); s = &hs; }
  if (dmo_compile(s, m) != 0) { free(m); return 0; }
  return m;
}
void dmo_model_free(dmo_model* m) { free(m); }

#define FIELD(name, ptr, count) if (!strcmp(field, name)) { int c_ = (count); if (c_ > max) c_ = max; for (int i_ = 0; i_ < c_; i_++) out[i_] = (double)((ptr)[i_]); return (count); }
int dmo_model_get(const dmo_model* m, const char* field, double* out, int max) {
  const dmo_spec* s = &m->s;
  FIELD("nq", &m->nq, 1) FIELD("nv", &m->nv, 1) FIELD("nbody", &s->nbody, 1) FIELD("njnt", &s->njnt, 1)
  FIELD("ngeom", &s->ngeom, 1) FIELD("nu", &s->nu, 1) FIELD("npair", &m->npair, 1)
  FIELD("body_mass", m->body_mass, s->nbody) FIELD("body_ipos", &m->body_ipos[0][0], 3 * s->nbody)
  FIELD("body_inertia", &m->body_inertia[0][0], 9 * s->nbody) FIELD("body_invweight0", &m->body_invweight0[0][0], 2 * s->nbody)
  FIELD("dof_invweight0", m->dof_invweight0, m->nv) FIELD("dof_armature", m->dof_armature, m->nv)
  FIELD("dof_damping", m->dof_damping, m->nv) FIELD("dof_parent", m->dof_parent, m->nv) FIELD("dof_body", m->dof_body, m->nv)
  FIELD("qpos0", m->qpos0, m->nq) FIELD("meaninertia", &m->meaninertia, 1) FIELD("total_mass", &m->total_mass, 1)
  FIELD("pair_g1", m->pair_g1, m->npair) FIELD("pair_g2", m->pair_g2, m->npair)
  FIELD("geom_quat", &m->geom_quat[0][0], 4 * s->ngeom) FIELD("geom_lpos", &m->geom_lpos[0][0], 3 * s->ngeom)
  FIELD("geom_lsize", &m->geom_lsize[0][0], 3 * s->ngeom) FIELD("timestep", &s->timestep, 1)
  FIELD("jnt_range", &s->jnt_range[0][0], 2 * s->njnt) FIELD("act_gear", s->act_gear, s->nu)
  return -1;
}
int dmo_model_set(dmo_model* m, const char* field, double v) {
  if (!strcmp(field, "enable_contact")) { m->enable_contact = (int)v; return 0; }
  if (!strcmp(field, "enable_limit")) { m->enable_limit = (int)v; return 0; }
  if (!strcmp(field, "pyramid_diag_mu2")) { m->pyramid_diag_mu2 = (int)v; return 0; }
  if (!strcmp(field, "pyramid_r_rescale")) { m->pyramid_r_rescale = (int)v; return 0; }
  if (!strcmp(field, "iterations")) { m->s.iterations = (int)v; return 0; }
  if (!strcmp(field, "max_efc")) { m->max_efc = (int)v > DMO_MAXEFC ? DMO_MAXEFC : (int)v; return 0; }
  if (!strcmp(field, "timestep")) { m->s.timestep = v; return 0; }
  if (!strcmp(field, "tolerance")) { m->s.tolerance = v; return 0; }
  if (!strcmp(field, "gravity_z")) { m->s.gravity[2] = v; return 0; }
  return -1;
}
int dmo_data_get(const dmo_model* m, const dmo_data* d, const char* field, double* out, int max) {
  const dmo_spec* s = &m->s;
  int nv = m->nv, n = d->nefc;
  FIELD("qpos", d->qpos, m->nq) FIELD("qvel", d->qvel, nv) FIELD("ctrl", d->ctrl, s->nu)
  FIELD("qacc_warmstart", d->qacc_warmstart, nv) FIELD("time", &d->time, 1)
  FIELD("xpos", &d->xpos[0][0], 3 * s->nbody) FIELD("xquat", &d->xquat[0][0], 4 * s->nbody)
  FIELD("xmat", &d->xmat[0][0], 9 * s->nbody) FIELD("xipos", &d->xipos[0][0], 3 * s->nbody)
  FIELD("geom_xpos", &d->geom_xpos[0][0], 3 * s->ngeom) FIELD("geom_xmat", &d->geom_xmat[0][0], 9 * s->ngeom)
  FIELD("qfrc_bias", d->qfrc_bias, nv) FIELD("qfrc_passive", d->qfrc_passive, nv) FIELD("qfrc_actuator", d->qfrc_actuator, nv)
  FIELD("qacc_smooth", d->qacc_smooth, nv) FIELD("qfrc_constraint", d->qfrc_constraint, nv) FIELD("qacc", d->qacc, nv)
  FIELD("ncon", &d->ncon, 1) FIELD("nefc", &d->nefc, 1) FIELD("nlimit", &d->nlimit, 1)
  FIELD("solver_iter", &d->solver_iter, 1) FIELD("solver_improvement", &d->solver_improvement, 1)
  FIELD("efc_pos", d->efc_pos, n) FIELD("efc_margin", d->efc_margin, n) FIELD("efc_R", d->efc_R, n)
  FIELD("efc_diagApprox", d->efc_diagApprox, n) FIELD("efc_vel", d->efc_vel, n) FIELD("efc_aref", d->efc_aref, n)
  FIELD("efc_b", d->efc_b, n) FIELD("efc_force", d->efc_force, n)
  if (!strcmp(field, "M")) { int c = 0; for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) if (c < max) out[c++] = d->M[i][j]; return nv * nv; }
  if (!strcmp(field, "efc_J")) { int c = 0; for (int i = 0; i < n; i++) for (int j = 0; j < nv; j++) if (c < max) out[c++] = d->efc_J[i][j]; return n * nv; }
  if (!strcmp(field, "efc_AR")) { int c = 0; for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) if (c < max) out[c++] = d->efc_AR[i][j]; return n * n; }
  if (!strcmp(field, "contact_geom")) { int c = 0; for (int i = 0; i < d->ncon; i++) { if (c + 1 < max) { out[c] = d->contact[i].geom1; out[c + 1] = d->contact[i].geom2; } c += 2; } return 2 * d->ncon; }
  if (!strcmp(field, "contact_dist")) { for (int i = 0; i < d->ncon && i < max; i++) out[i] = d->contact[i].dist; return d->ncon; }
  if (!strcmp(field, "contact_pos")) { int c = 0; for (int i = 0; i < d->ncon; i++) for (int k = 0; k < 3; k++) if (c < max) out[c++] = d->contact[i].pos[k]; return 3 * d->ncon; }
  if (!strcmp(field, "contact_frame")) { int c = 0; for (int i = 0; i < d->ncon; i++) for (int k = 0; k < 9; k++) if (c < max) out[c++] = d->contact[i].frame[k]; return 9 * d->ncon; }
  if (!strcmp(field, "contact_dim")) { for (int i = 0; i < d->ncon && i < max; i++) out[i] = d->contact[i].dim; return d->ncon; }
  return -1;
}
int dmo_data_set(const dmo_model* m, dmo_data* d, const char* field, const double* in, int n) {
  if (!strcmp(field, "qpos")) { memcpy(d->qpos, in, sizeof(double) * (n < m->nq ? n : m->nq)); return 0; }
  if (!strcmp(field, "qvel")) { memcpy(d->qvel, in, sizeof(double) * (n < m->nv ? n : m->nv)); return 0; }
  if (!strcmp(field, "ctrl")) { memcpy(d->ctrl, in, sizeof(double) * (n < m->s.nu ? n : m->s.nu)); return 0; }
  if (!strcmp(field, "qacc_warmstart")) { memcpy(d->qacc_warmstart, in, sizeof(double) * (n < m->nv ? n : m->nv)); return 0; }
  if (!strcmp(field, "time")) { d->time = in[0]; return 0; }
  if (!strcmp(field, "xipos")) { memcpy(&d->xipos[0][0], in, sizeof(double) * (n < 3 * m->s.nbody ? n : 3 * m->s.nbody)); return 0; }
  return -1;
}
