/* dm_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C float64 restatement of the hot path of mingfeisun/DeepMimic_mujoco:
 *   DPEnv.step / reset_model / reset_model_init / _get_obs / is_done / calc_config_reward
 *   (src/dp_env_v3.py:62-164), which bottoms out in MuJoCo 2.0's mj_step / mj_forward on the model
 *   src/mujoco/humanoid_deepmimic/envs/asset/dp_env_v3.xml (RK4, PGS 50 iterations, h = 0.0166).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The product (libdmenv.so, HIP) never links or calls it.
 *
 * PARITY STATUS
 *   - env logic (obs, done, rewards, frame index): pinned against tests/golden/env_logic_golden.npz,
 *     produced by executing the reference's own Python.
 *   - physics: PARITY UNPINNED.  MuJoCo 2.0 (`mujoco200`, closed source, behind mujoco-py; unpinned
 *     version per README.md:24-27) is neither in /root/reference nor installable here, and the
 *     reference holds no test or golden vector for it.  The step below restates MuJoCo's published
 *     computation pipeline ("Computation" chapter; engine_forward / engine_core_smooth /
 *     engine_core_constraint / engine_collision_* / engine_solver of the later Apache-2.0 release)
 *     from memory and is anchored by physical known-answer tests (tests/test_oracle_physics.py).
 */
#ifndef DM_ORACLE_H
#define DM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define DMO_MAXBODY 16
#define DMO_MAXJNT 32
#define DMO_MAXV 36
#define DMO_MAXQ 40
#define DMO_MAXU 32
#define DMO_MAXGEOM 20
#define DMO_MAXPAIR 160
#define DMO_MAXCON 96
#define DMO_MAXEFC 256

enum { DMO_JNT_FREE = 0, DMO_JNT_HINGE = 3 };                       /* mjtJoint values */
enum { DMO_GEOM_PLANE = 0, DMO_GEOM_SPHERE = 2, DMO_GEOM_CAPSULE = 3, DMO_GEOM_BOX = 6 }; /* mjtGeom */

/* ---- primitive (MJCF-level) description ------------------------------------------------------ */
typedef struct {
  int nbody, njnt, ngeom, nu;
  /* bodies (0 = world) */
  int body_parent[DMO_MAXBODY];
  double body_pos[DMO_MAXBODY][3];
  /* joints, in XML order; a body's joints are contiguous */
  int jnt_type[DMO_MAXJNT], jnt_body[DMO_MAXJNT], jnt_limited[DMO_MAXJNT];
  double jnt_axis[DMO_MAXJNT][3], jnt_range[DMO_MAXJNT][2];
  double jnt_armature[DMO_MAXJNT], jnt_damping[DMO_MAXJNT];
  /* geoms */
  int geom_type[DMO_MAXGEOM], geom_body[DMO_MAXGEOM], geom_condim[DMO_MAXGEOM];
  int geom_contype[DMO_MAXGEOM], geom_conaffinity[DMO_MAXGEOM], geom_has_fromto[DMO_MAXGEOM];
  double geom_size[DMO_MAXGEOM][3], geom_pos[DMO_MAXGEOM][3], geom_fromto[DMO_MAXGEOM][6];
  double geom_mass[DMO_MAXGEOM], geom_friction[DMO_MAXGEOM][3], geom_margin[DMO_MAXGEOM];
  /* motors */
  int act_jnt[DMO_MAXU];
  double act_gear[DMO_MAXU], act_ctrlrange[DMO_MAXU][2];
  /* <contact><exclude> body pairs */
  int nexclude, exclude[16][2];
  /* <option> */
  double timestep, gravity[3], tolerance;
  int iterations;
  double solref[2], solimp[5]; /* global defaults used by every joint limit and geom */
} dmo_spec;

/* ---- compiled model ------------------------------------------------------------------------- */
typedef struct {
  dmo_spec s;
  int nq, nv;
  int jnt_qposadr[DMO_MAXJNT], jnt_dofadr[DMO_MAXJNT];
  int body_jntadr[DMO_MAXBODY], body_jntnum[DMO_MAXBODY], body_dofadr[DMO_MAXBODY], body_dofnum[DMO_MAXBODY];
  int dof_body[DMO_MAXV], dof_parent[DMO_MAXV], dof_jnt[DMO_MAXV];
  double dof_armature[DMO_MAXV], dof_damping[DMO_MAXV], dof_invweight0[DMO_MAXV];
  double body_mass[DMO_MAXBODY], body_ipos[DMO_MAXBODY][3], body_inertia[DMO_MAXBODY][9];
  double body_invweight0[DMO_MAXBODY][2];
  double geom_quat[DMO_MAXGEOM][4], geom_lpos[DMO_MAXGEOM][3]; /* in body frame (fromto resolved) */
  double geom_lsize[DMO_MAXGEOM][3];
  double qpos0[DMO_MAXQ];
  double meaninertia, total_mass;
  int npair, pair_g1[DMO_MAXPAIR], pair_g2[DMO_MAXPAIR]; /* candidate geom pairs in contact-list order */
  /* switches (configs 2/3 of BASELINE.json): */
  int enable_contact, enable_limit;
  /* [L]-confidence details of the pyramidal regulariser kept switchable (see dm_oracle.c) */
  int pyramid_diag_mu2, pyramid_r_rescale;
  /* rows kept per evaluation (default DMO_MAXEFC); set to 63 to mirror the HIP path's on-chip capacity: contacts
   * whose rows do not fit are dropped in list order (MuJoCo itself stops at njmax with a warning) */
  int max_efc;
} dmo_model;

typedef struct {
  int geom1, geom2, dim;
  double dist, pos[3], frame[9], includemargin, friction[5];
} dmo_contact;

/* ---- per-environment state + the intermediates of the LAST forward evaluation ------------------ */
typedef struct {
  /* integration state */
  double qpos[DMO_MAXQ], qvel[DMO_MAXV], ctrl[DMO_MAXU], qacc_warmstart[DMO_MAXV], time;
  /* position stage */
  double xpos[DMO_MAXBODY][3], xquat[DMO_MAXBODY][4], xmat[DMO_MAXBODY][9], xipos[DMO_MAXBODY][3];
  double xanchor[DMO_MAXJNT][3], xaxis[DMO_MAXJNT][3];
  double geom_xpos[DMO_MAXGEOM][3], geom_xmat[DMO_MAXGEOM][9];
  double cdof[DMO_MAXV][6];                /* [ang; lin], reference point = world origin */
  double M[DMO_MAXV][DMO_MAXV], L[DMO_MAXV][DMO_MAXV]; /* mass matrix and its Cholesky factor */
  int ncon;
  dmo_contact contact[DMO_MAXCON];
  int nefc, nlimit;
  double efc_J[DMO_MAXEFC][DMO_MAXV], efc_pos[DMO_MAXEFC], efc_margin[DMO_MAXEFC];
  double efc_diagApprox[DMO_MAXEFC], efc_R[DMO_MAXEFC], efc_KBI[DMO_MAXEFC][3];
  double efc_vel[DMO_MAXEFC], efc_aref[DMO_MAXEFC], efc_b[DMO_MAXEFC], efc_force[DMO_MAXEFC];
  double (*efc_AR)[DMO_MAXEFC];            /* heap: [DMO_MAXEFC][DMO_MAXEFC] */
  /* velocity / force stage */
  double qfrc_bias[DMO_MAXV], qfrc_passive[DMO_MAXV], qfrc_actuator[DMO_MAXV];
  double qacc_smooth[DMO_MAXV], qfrc_constraint[DMO_MAXV], qacc[DMO_MAXV];
  int solver_iter;
  double solver_improvement;
} dmo_data;

/* model */
void dmo_humanoid_spec(dmo_spec* s);                 /* dp_env_v3.xml restated as a table */
int  dmo_compile(const dmo_spec* s, dmo_model* m);   /* 0 = ok */
dmo_data* dmo_data_create(const dmo_model* m);       /* = MjSim(model): state at qpos0, zeros */
void dmo_data_destroy(dmo_data* d);
void dmo_reset_data(const dmo_model* m, dmo_data* d);/* mj_resetData (sim.reset()) */

/* physics */
void dmo_forward(const dmo_model* m, dmo_data* d);   /* mj_forward (sim.forward()) */
void dmo_step(const dmo_model* m, dmo_data* d);      /* mj_step with RK4 (sim.step()) */

/* env layer (src/dp_env_v3.py) */
void   dmo_get_obs(const dmo_model* m, const dmo_data* d, double* obs56);
double dmo_com_z(const dmo_model* m, const dmo_data* d);
int    dmo_is_done(const dmo_model* m, const dmo_data* d);
void   dmo_set_state(const dmo_model* m, dmo_data* d, const double* qpos, const double* qvel);
double dmo_config_reward(const dmo_model* m, const dmo_data* d, const double* data_config, int n_frames,
                         int* idx_curr);
/* one DPEnv.step: ctrl <- action; n_substeps x mj_step; obs; reward (mode); done */
enum { DMO_REW_ALIVE = 0, DMO_REW_V3_CONFIG = 1, DMO_REW_V2_POSE = 2 };
void dmo_env_step(const dmo_model* m, dmo_data* d, const double* action, int n_substeps, int reward_mode,
                  const double* data_config, int n_frames, int* idx_curr, int idx_init,
                  double* obs56, double* reward, int* done);

/* batched helper for the CPU baseline (OpenMP over envs when compiled with -fopenmp) */
/* 5-term imitation reward (code.md:1017-1143); feature row layout: deepmimic_mujoco_amd/imitation.py */
void dmo_imitation_features(const dmo_model* m, const double* qpos, const double* qvel, const double* params, double* feat112);
double dmo_imitation_reward(const dmo_model* m, const double* f0, const double* f1, const double* params, double shift_x,
                            double shift_y, double* terms5);
double dmo_v1_reward(const dmo_model* m, const double* f0, const double* f1, const double* f1v, const double* params, double* terms);
void dmo_env_step_v1(const dmo_model* m, dmo_data* d, const double* action, int n_substeps, const double* table, int F,
                     const double* params, double mocap_dt, int* idx_curr, int idx_init, double* obs, double* reward, int* done);
void dmo_env_step_imitation(const dmo_model* m, dmo_data* d, const double* action, int n_substeps, const double* table, int F,
                            const double* params, int* idx_curr, int* cycle, double* obs, double* reward, int* done);
void dmo_batch_step(const dmo_model* m, dmo_data** ds, int n, const double* actions, int n_substeps,
                    double* obs, double* reward, unsigned char* done, int nthreads);
void dmo_batch_step_imitation(const dmo_model* m, dmo_data** ds, int n, const double* actions, int n_substeps, const double* table, int F,
                              const double* params, int* idx_curr, int* cycle, double* obs, double* reward, unsigned char* done, int nthreads);
/* bench.py's cpu_baseline loop in C: per-env action streams, RSI reset on done, OpenMP over whole trajectories. */
long dmo_bench_rollout(const dmo_model* m, dmo_data** ds, int n, int steps, const double* cfg, const double* vel, int F, const double* table,
                       const double* params, double sigma, unsigned long long seed, int nthreads, long* n_done, double* reward_sum);

/* heap model + string-keyed accessors for the ctypes test harness (oracle/oracle.py) */
dmo_model* dmo_model_new(const dmo_spec* s);         /* s == NULL -> the dp_env_v3 humanoid */
void dmo_model_free(dmo_model* m);
int dmo_model_get(const dmo_model* m, const char* field, double* out, int max);
int dmo_model_set(dmo_model* m, const char* field, double v);
int dmo_data_get(const dmo_model* m, const dmo_data* d, const char* field, double* out, int max);
int dmo_data_set(const dmo_model* m, dmo_data* d, const char* field, const double* in, int n);

/* diagnostics: tallies of the narrow-phase cases of the two own routines (box-box [0..4], capsule-box [5..9]; see dm_oracle.c);
 * mode 1: reset + switch on, 0: switch off, -1: read only.  No effect on any result. */
void dmo_narrow_cases(long long* out, int mode);

int dmo_sizeof_model(void);
int dmo_sizeof_data(void);

#ifdef __cplusplus
}
#endif
#endif
