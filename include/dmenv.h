/* dmenv.h — C ABI of libdmenv.so: the MI355X-native batched DeepMimic humanoid environment.
 *
 * This is the drop-in boundary for the hot path of mingfeisun/DeepMimic_mujoco.  In the reference the
 * path sits behind mujoco-py's Cython API (third-party, EXTERNAL) as used by gym's MujocoEnv and by
 * src/dp_env_v3.py; each entry point below names the reference interface it replaces.  Plain C types
 * only (no torch / numpy types); every function returns 0 on success or a negative DM_E* code, and
 * dm_last_error() returns a thread-local message.  A dm_batch is confined to one host thread and one
 * HIP stream; different batches (e.g. one per GPU) may be driven from different threads/processes.
 *
 * All per-environment state (qpos, qvel, time, qacc_warmstart, mocap frame indices) lives in device
 * memory owned by the library, as [N, 35] / [N, 34] row-major float64 arrays: one wavefront per
 * environment loads its row with one coalesced access.  Caller buffers (actions in, obs / reward /
 * done out) may be host or device memory (DM_PTR_HOST / DM_PTR_DEVICE).
 */
#ifndef DMENV_H
#define DMENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM_ABI_VERSION 4

/* fixed sizes of the DeepMimic humanoid (dp_env_v3.xml:21-156): the kernels are specialised to this tree */
#define DM_NBODY 14
#define DM_NJNT 29
#define DM_NQ 35
#define DM_NV 34
#define DM_NU 28
#define DM_NGEOM 16
#define DM_NOBS 56
#define DM_MAXPAIR 128
#define DM_MAXEFC 64   /* lanes of the per-env wavefront = stride of the per-row arrays */
#define DM_MAXROWS 63  /* constraint rows per environment held on chip (one lane each; the 64th lane carries the smooth force);
                          overflow is reported (DM_F_STATUS bit 0), not silent */

enum { DM_OK = 0, DM_EINVAL = -1, DM_EHIP = -2, DM_ENOMEM = -3, DM_EUNSUPPORTED = -4, DM_ENODEVICE = -5 };
enum { DM_PTR_HOST = 0, DM_PTR_DEVICE = 1 };

typedef struct dm_model dm_model;
typedef struct dm_mocap dm_mocap;
typedef struct dm_batch dm_batch;

/* Compiled model tables (host pointers, float64 / int32, read once by dm_model_create).
 * Replaces: mujoco_py.load_model_from_path(xml) -> PyMjModel, called from gym MujocoEnv.__init__
 * (reference call site src/dp_env_v3.py:59).  Filled by deepmimic_mujoco_amd/model.py. */
typedef struct {
  int32_t abi_version;
  int32_t nbody, njnt, nq, nv, nu, ngeom, npair, iterations;
  const int32_t* body_parentid;   /* [nbody] */
  const int32_t* body_dofnum;     /* [nbody] */
  const double* body_pos;         /* [nbody,3] */
  const double* body_ipos;        /* [nbody,3] */
  const double* body_mass;        /* [nbody] */
  const double* body_inertia;     /* [nbody,9] about COM, body axes */
  const double* body_invweight0;  /* [nbody,2] */
  const int32_t* jnt_type;        /* [njnt] 0 free, 3 hinge */
  const int32_t* jnt_bodyid;      /* [njnt] */
  const int32_t* jnt_limited;     /* [njnt] */
  const double* jnt_axis;         /* [njnt,3] */
  const double* jnt_range;        /* [njnt,2] */
  const double* dof_armature;     /* [nv] */
  const double* dof_damping;      /* [nv] */
  const double* dof_invweight0;   /* [nv] */
  const int32_t* geom_type;       /* [ngeom] mjtGeom */
  const int32_t* geom_bodyid;     /* [ngeom] */
  const int32_t* geom_condim;     /* [ngeom] */
  const double* geom_pos;         /* [ngeom,3] in body frame */
  const double* geom_mat;         /* [ngeom,9] in body frame */
  const double* geom_size;        /* [ngeom,3] */
  const double* geom_margin;      /* [ngeom] */
  const double* geom_friction;    /* [ngeom,3] */
  const int32_t* pair_geom;       /* [npair,2] candidate pairs in contact-list order, geom1 = lower type */
  const int32_t* actuator_dofid;  /* [nu] */
  const double* actuator_gear;    /* [nu] */
  const double* actuator_ctrlrange; /* [nu,2] */
  double timestep, gravity[3], tolerance, solref[2], solimp[5], meaninertia;
} dm_model_desc;

int dm_model_create(const dm_model_desc* desc, dm_model** out);
void dm_model_destroy(dm_model* m);

/* Mocap reference tables (MocapDM.data_config [F,35], .data_vel [F,34]; src/mujoco/mocap_v2.py:78-149).
 * Replaces: the Python-list lookups `self.mocap.data_config[idx]`, `.data_vel[idx]` at
 * src/dp_env_v3.py:93,150-151 — the tables become device-resident. */
int dm_mocap_create(const double* data_config, const double* data_vel, int32_t n_frames, double dt,
                    dm_mocap** out);
/* Reference feature rows of the full 5-term imitation reward (code.md:1017-1143 — the reward the reference's notes
 * specify but dp_env_v3.py:117-128 never computes): table [n_frames, 112] and params [32], both built on the host by
 * deepmimic_mujoco_amd/imitation.py (row layout documented there).  Call before dm_batch_create; enables reward mode 3. */
int dm_mocap_set_imitation(dm_mocap* mc, const double* table, int32_t n_cols, const double* params);
void dm_mocap_destroy(dm_mocap* mc);

/* flags for dm_batch_create */
#define DM_FLAG_NO_CONTACT (1u << 0) /* BASELINE.json config 2: collision off   */
#define DM_FLAG_NO_LIMIT   (1u << 1) /* BASELINE.json config 2: joint limits off */

/* Replaces: mujoco_py.MjSim(model) (gym MujocoEnv.__init__), once per environment; here once per batch.
 * device_id: HIP device ordinal.  There is no CPU backend: no device -> DM_ENODEVICE. */
int dm_batch_create(const dm_model* m, const dm_mocap* mc, int32_t n_envs, int32_t device_id, uint32_t flags,
                    dm_batch** out);
void dm_batch_destroy(dm_batch* b);
int dm_batch_set_stream(dm_batch* b, void* hip_stream); /* default: a stream owned by the batch */

/* options */
enum {
  DM_OPT_REWARD_MODE = 1, /* 0 alive=1.0 (dp_env_v3.py:117-128, default), 1 v3-config (:89-104), 2 v2-pose (dp_env_v2.py:116-183),
                             3 imitation: pose/velocity/end-effector/root/COM terms of code.md:1017-1143 against frame idx+1,
                             4 v1-quat: dp_env_v1's reward (dp_env_v1.py:82-158: JOINT_WEIGHT-ed |quaternion difference angle| pose error,
                               L1 angular-rate and root errors, every int(mocap_dt // dt) steps, minus 0.1 sum ctrl^2) on this model's
                               hinge triples; modes 3 and 4 need dm_mocap_set_imitation() */
  DM_OPT_AUTORESET = 2,   /* 0 off (default), 1 RSI on done, 2 noisy-init on done (DummyVecEnv convention) */
  DM_OPT_ACTION_MODE = 3, /* 0 raw ctrl (dp_env_v3.py:112, default), 1 P-control 0.8*(mocap_cfg - q) + action (env_torque_test.py:20),
                             2 PD kp*(mocap_cfg - q) + kd*(mocap_vel - v) + action (setting_states.py:207-226, gains mocap_util.py:22-24) */
  DM_OPT_SEED = 4,
  DM_OPT_DIAGNOSTICS = 5, /* 1 (default): every step also stores sim.data.xipos and the contact (geom1, geom2) list (DM_F_XIPOS,
                             DM_F_CONTACT_GEOMS: 848 B per env-step); 0: state, obs, reward, done and the row / contact counts only —
                             those two fields then keep the values of the last set_state / reset (DPVecEnv's default) */
  DM_OPT_PIPELINE = 6,    /* 1 (default): a step is one launch on the batch's stream.  P = 2..DM_MAX_PIPELINE: the env range is cut
                             into P contiguous sub-batches, each stepped on its own internal stream.  With DEVICE pointers a
                             sub-batch's launch of call k+1 waits only for its own launch of call k and for the caller's stream
                             at the time of the call (the inputs), so consecutive calls overlap: the next sub-batch's workgroups
                             take the wave slots freed while the previous one drains.  The outputs of such calls are complete
                             once the caller's stream has joined: dm_batch_join() (no host wait), dm_batch_sync(), or any other
                             entry point of the batch.  Results are identical for every P. */
  DM_OPT_PACKED = 7,      /* 0 (default): one environment per wavefront (k_step_narrow).  1 / 2: FOUR environments per wavefront, one 16-lane
                             DPP row each (k_step_packed, csrc/slot_kernel.h) wherever that kernel covers the call: every reward mode (0..4; mode 4,
                             v1-quat, included), with or without the fused policy step.  Per-environment
                             capacities of that path (csrc/slot_kernel.h SLOT_*): DM_PACKED_MAXROWS constraint rows inside a horizon launch
                             (dm_batch_rollout, DM_OPT_STEP_QUEUE: two full 16-row sets and a partial third — a humanoid standing on both
                             feet holds 32 contact rows plus joint limits), DM_PACKED_MAXROWS_PER_STEP in a per-step launch — unless the option
                             is 2: per-step launches then run k_step_packed_ext, the same step with the three-set code compiled in
                             (DM_PACKED_MAXROWS rows; ~8 % slower for environments that never get there: meant for populations that stand
                             on both feet) — (of them at most DM_PACKED_MAXLIMROWS joint limits), DM_PACKED_MAXCON contacts from at most DM_PACKED_MAXFRAME geom pairs,
                             DM_PACKED_MAXCAND pairs past the bounding spheres; an environment that exceeds one in some step is re-stepped
                             by the one-env code in the same call (dm_batch_redo_total counts them).  The throughput kernel from 4096 envs up
                             on one MI355X (round 6, closed loop with two pipelined sub-batches: 13.3 against 12.4 M env-steps/s at 4096 envs,
                             18.7 against 12.4 M at 6144; below — 10.3 against 10.6 M at 3072 — a per-step launch is less than one round of
                             lone waves and the one-env kernel stays ahead), while horizon launches (dm_batch_rollout, DM_OPT_STEP_QUEUE)
                             use it at any size.  Results agree with the oracle to the same 1e-9 bar and do not depend
                             on which environments share a wave; they differ from the one-env kernel's in the last bits (other
                             summation orders). */
  DM_OPT_STEP_QUEUE = 8   /* 0 (default): every dm_batch_step call launches.  Q = 1..DM_MAX_STEP_QUEUE: dm_batch_step calls with DEVICE
                             pointers are QUEUED — nothing is launched — and run together, in call order, as one horizon launch
                             (k_rollout_packed: every wavefront steps its four environments through all queued steps at its own pace
                             instead of waiting for the slowest wave of every step) when Q calls are queued or when any other entry
                             point of the batch is called: dm_batch_join() (no host wait), dm_batch_sync(), dm_batch_get(), ...  The
                             contract is DM_OPT_PIPELINE's, extended to the inputs: the outputs of a queued call are complete once the
                             caller's stream has joined, and its action buffer must stay untouched until then.  A call one of whose four
                             buffers OVERLAPS (byte ranges, in any role) a buffer of a queued call — the same tensors step after step,
                             a view into them, a queued call's observations handed in as actions: a closed loop — first runs what is
                             queued, i.e. degenerates to one launch per call.  Results are bit-identical to unqueued DM_OPT_PACKED
                             steps.  Queuing applies where dm_batch_rollout uses one launch per horizon (DM_OPT_PACKED on, reward modes
                             0..3, constraint rows, at most two packed waves per SIMD); elsewhere calls launch at once as without it.
                             dm_batch_destroy() DROPS what is still queued (the buffers belong to the caller and may be gone). */
};
/* per-environment capacities of the DM_OPT_PACKED path (= csrc/slot_kernel.h SLOT_MAXROWS, SLOT_MAXLIMROWS, SLOT_MAXCON, SLOT_MAXFRAME, SLOT_MAXCAND) */
#define DM_PACKED_MAXROWS 40
#define DM_PACKED_MAXROWS_PER_STEP 32
#define DM_PACKED_MAXLIMROWS 16
#define DM_PACKED_MAXCON 13
#define DM_PACKED_MAXFRAME 8
#define DM_PACKED_MAXCAND 32
#define DM_MAX_STEP_QUEUE 256
#define DM_MAX_PIPELINE 8
/* further option ids (diagnostics / tests; defaults are what the timed path uses):
 *   100 global id of env 0 of this batch (multi-GPU sharding: RNG streams are keyed by the global env id)
 *   101 per-stage shader-clock profile (k_step_prof + dm_batch_read_profile)
 *   102 1: register tier of 32 columns of A + memory strip (default); 0: all 64 columns in registers (k_step)
 *   103 1: force the guarded PGS re-solve path (results must not change)
 *   104 1: longest-first dispatch order from the previous step's row counts (default: per-step launches order themselves through tickets their envs take at the end of a step, horizon launches are grouped by a counting sort once per launch); 0: identity.
 *       Tickets are kept per pipelined part (DM_OPT_PIPELINE): a launch over the WHOLE batch of a batch configured with a pipeline depth above 1 (host-pointer
 *       steps, profiled steps) takes none and runs in the stored order — results never depend on the dispatch order, only the duration of such a launch does */
int dm_batch_set_option(dm_batch* b, int32_t opt, int64_t value);

/* Replaces: MujocoEnv.set_state(qpos, qvel) = sim.set_state(...) + sim.forward() (src/dp_env_v3.py:153,160):
 * qpos/qvel replaced for the masked envs (mask NULL = all), time and qacc_warmstart kept, derived
 * quantities (xipos ...) recomputed.  frame_idx may be NULL (unchanged). */
int dm_batch_set_state(dm_batch* b, const double* qpos, const double* qvel, const int32_t* frame_idx,
                       const uint8_t* mask, int32_t ptr_kind);

/* Replaces: MujocoEnv.reset() = sim.reset() + DPEnv.reset_model() (src/dp_env_v3.py:148-156), or
 * DPEnv.reset_model_init() (:158-164).  mode 0 = RSI (frame ~ U{0..F-1}), 1 = noisy init pose
 * (init_qpos/init_qvel + U(-0.01, 0.01)), both from a counter-based per-env RNG keyed by (seed, env, episode);
 * mode 2 = sim.reset() only (qpos0, zero velocity).  time and qacc_warmstart are zeroed for modes 0/1 only when
 * `hard` is nonzero (sim.reset() semantics); reset_model_init() called on its own keeps them. */
int dm_batch_reset(dm_batch* b, int32_t mode, int32_t hard, const uint8_t* mask, int32_t ptr_kind);

/* Replaces: DPEnv.step(action) (src/dp_env_v3.py:106-132) = do_simulation(action, n) [ctrl <- action; n x mj_step]
 * + _get_obs() + reward + is_done(), for every environment of the batch in one launch.
 * action [N,28], obs [N,56], reward [N] float64; done [N] uint8. */
int dm_batch_step(dm_batch* b, const double* action, double* obs, double* reward, uint8_t* done,
                  int32_t n_substeps, int32_t ptr_kind);
/* Replaces: DPEnv._get_obs() (src/dp_env_v3.py:62-65) without stepping. */
int dm_batch_get_obs(dm_batch* b, double* obs, int32_t ptr_kind);

/* state / diagnostics access.  Replaces reads and writes of sim.data.<field> (numpy views in mujoco-py). */
enum {
  DM_F_QPOS = 1,        /* double [N,35] */
  DM_F_QVEL = 2,        /* double [N,34] */
  DM_F_QACC_WARMSTART = 3, /* double [N,34] */
  DM_F_TIME = 4,        /* double [N] */
  DM_F_FRAME_IDX = 5,   /* int32 [N]  (DPEnv.idx_curr) */
  DM_F_FRAME_INIT = 6,  /* int32 [N]  (DPEnv.idx_init) */
  DM_F_XIPOS = 7,       /* double [N,14,3] body COM positions of the last forward evaluation */
  DM_F_COM_Z = 8,       /* double [N] */
  DM_F_NCON = 9,        /* int32 [N] contacts of the last forward evaluation */
  DM_F_NEFC = 10,       /* int32 [N] constraint rows */
  DM_F_CONTACT_GEOMS = 11, /* int32 [N,DM_MAXEFC,2] (geom1, geom2) per contact, -1 padded */
  DM_F_STATUS = 12,     /* int32 [N] bit0: constraint rows overflowed DM_MAXROWS, bit1: non-finite state */
  DM_F_SOLVER_ITER = 13,/* int32 [N] PGS sweeps of the last forward evaluation */
  DM_F_CTRL = 14,       /* double [N,28] last (unclamped) ctrl */
  DM_F_EPISODE = 15,    /* int32 [N] episode counter used by the reset RNG */
  DM_F_CYCLE = 16       /* int32 [N] completed motion cycles since the episode started (reward mode 3) */
};
int dm_batch_get(dm_batch* b, int32_t field, void* out, size_t bytes, int32_t ptr_kind);
int dm_batch_set(dm_batch* b, int32_t field, const void* in, size_t bytes, int32_t ptr_kind);

/* One forward evaluation (mj_forward = sim.forward()) of environment `env` with every intermediate dumped:
 * used by the parity tests to compare stage by stage with the oracle.  Layout of `out` (float64):
 *   M[34*34] | qfrc_bias[34] | qacc_smooth[34] | qacc[34] | xipos[14*3] | nefc | ncon | solver_iter |
 *   per row r < DM_MAXEFC: J[34] , then pos, margin, R, aref, b, force   (DM_DEBUG_DOUBLES in total) */
#define DM_DEBUG_DOUBLES (34 * 34 + 34 * 3 + 42 + 3 + DM_MAXEFC * (34 + 6))
int dm_batch_debug_forward(dm_batch* b, int32_t env, double* out_host);

/* kernel timing of the last dm_batch_step launch, measured with HIP events on the batch's stream (ms) */
int dm_batch_last_step_ms(dm_batch* b, float* ms);
int dm_batch_enable_timing(dm_batch* b, int32_t on);

/* diagnostic: per-env shader-clock cycles of the last step by stage (enable with dm_batch_set_option(b, 101, 1)):
 * out [N,32] int64 = kinematics, mass matrix+factor, bias, rows(collision), constraint, whole step, nefc, PGS sweeps;
 * [8..13] the constraint stage's parts (smooth solve, J rows, impedance + half solve, A, warm start + PGS, assembly);
 * [16] geom poses + limit rows, [17,18,19] / [20,21,22] broad phase, narrow phase, row emission of pair pass 0 / 1, [23] tail,
 * [24,25] evaluations in which a pair of pass 0 / 1 passed the bounding spheres, [26,27] such pairs (summed over the step) */
int dm_batch_read_profile(dm_batch* b, long long* out_host);

/* Replaces: MlpPolicy.act(stochastic, ob) (src/mlp_policy_trpo.py:63-65; network :35-58, DiagGaussianPd src/distributions.py:220-245)
 * for a whole batch in one launch: obz = clip((ob - mean) / std, +-5), policy and value 2x100 tanh MLPs, Gaussian sample.
 * All pointers are DEVICE pointers on the current HIP device; `weights` is the packed float32 parameter block
 * (dm_policy_weight_count() floats; layout in csrc/policy_kernel.h, filled by deepmimic_mujoco_amd/policy.py MlpPolicy.pack).
 * Noise is a counter-based stream keyed by (seed, counter, env, action): pass a fresh `counter` per call. */
int dm_policy_weight_count(void);
int dm_policy_act(const float* weights, const double* obs, double* action, float* vpred, int32_t n, int32_t stochastic,
                  uint64_t seed, uint64_t counter, void* hip_stream);

/* dm_batch_step followed, INSIDE the step kernel, by dm_policy_act on the observations it produced: the wave that steps env e writes
 * obs / reward / done as dm_batch_step does and then next_action[e] (28) and next_vpred[e] for that observation (the fresh episode's
 * after an auto-reset).  One launch per rollout step (src/trpo.py:47-66: `ac, vpred = pi.act(ob)`; `ob, rew, new, _ = env.step(ac)`),
 * so consecutive steps of a pipelined batch (DM_OPT_PIPELINE) overlap although every step's action depends on the last one's
 * observation: that dependency stays inside a sub-batch's stream.  Device pointers only; same noise stream as dm_policy_act. */
int dm_batch_step_act(dm_batch* b, const double* action, double* obs, double* reward, uint8_t* done, int32_t n_substeps,
                      const float* weights, double* next_action, float* next_vpred, int32_t stochastic, uint64_t seed, uint64_t counter);

/* T steps per call — the loop body of traj_segment_generator (src/trpo.py:47-80: `ac, vpred = pi.act(stochastic, ob)`;
 * `ob, rew, new, _ = env.step(ac)`; on `new` the reset of :77-79 through DM_OPT_AUTORESET) for a whole horizon.  Device pointers only.
 *   action [T + 1, N, 28] f64: row t is consumed by step t; with `weights` rows 1..T are WRITTEN (the policy's action for the observation
 *                              of step t - 1: row 0 is the caller's, row T belongs to the next horizon); without, rows 0..T-1 are read only
 *   obs [T, N, 56] f64, reward [T, N] f64, done [T, N] u8: row t = what dm_batch_step returns for step t
 *   vpred [T, N] f32 (with `weights`): row t = the value of obs row t;  counter: the draw counter of step 0 (step t uses counter + t)
 * Results are those of T dm_batch_step / dm_batch_step_act calls.  With DM_OPT_PACKED (a reward mode that kernel covers, a model with
 * constraint rows, at most two packed waves per SIMD = 8 192 environments on an MI355X; DM option 106 = 1 / 0 forces / forbids it) the
 * horizon is ONE launch: every wavefront steps its four environments T times at its own pace, so the horizon lasts as long as the slowest
 * wave's sum over T steps instead of the sum of every step's slowest wave (4 096 envs: 17.3 M env-steps/s against 12.2 M through
 * dm_batch_step); an environment that exceeds the packed path's capacities in some step is re-stepped inside its wave by the one-env code.
 * Otherwise T step launches are issued. */
int dm_batch_rollout(dm_batch* b, double* action, double* obs, double* reward, uint8_t* done, int32_t T, int32_t n_substeps,
                     const float* weights, float* vpred, int32_t stochastic, uint64_t seed, uint64_t counter);

/* Replaces: add_vtarg_and_adv (src/trpo.py:83-94) for N environments at once: rew, vpred, adv, tdlamret [T, N] float32,
 * isnew [T, N] int32 (isnew[t] = the observation of step t starts an episode), nextvpred [N]; device pointers. */
int dm_gae(const float* rew, const float* vpred, const int32_t* isnew, const float* nextvpred, float* adv, float* tdlamret,
           int32_t T, int32_t n, double gamma, double lam, void* hip_stream);

/* Replaces: the episode bookkeeping of the generator's loop (src/trpo.py:68-79) over a [T, N] segment that dm_batch_rollout / T dm_batch_step
 * calls wrote (reward f64, done u8).  cur_ret [N] f64 / cur_len [N] i64: return and length of every environment's open episode, read and
 * updated.  Every episode that ends inside the segment appends a record of three int64 words to `records` [cap][3] — {t << 32 | env, the
 * float64 return's bits, length} — in arrival order (*count = number of episodes, which may exceed cap: the rest is dropped); sorted by the
 * first word they are in the order a one-env loop appends them (t, then env).  Device pointers; returns are float64 sums in step order. */
int dm_episode_scan(const double* reward, const uint8_t* done, int32_t T, int32_t n, double* cur_ret, int64_t* cur_len, int32_t* count,
                    int32_t cap, int64_t* records, void* hip_stream);

/* Replaces: one epoch of the value fit of src/trpo.py:288-296 — for each of `nb` minibatches of `bs` samples (already shuffled:
 * ob [nb*bs, 56] float32, ret [nb*bs] float32): `pi.ob_rms.update(mbob)` (src/utils/misc_util.py:53-70: rms_sum / rms_sumsq [56] and
 * rms_count float64, rms_mean / rms_std [56] float32 refreshed), the gradient of mean((vpred - ret)^2) w.r.t. the 56-100-100-1 tanh
 * value net (theta: dm_vf_param_count() floats = vffc1/w, vffc1/b, vffc2/w, vffc2/b, vffinal/w, vffinal/b, weights row-major
 * [in][out]) and the MpiAdam step (src/mpi_adam.py:21-35) with the caller's per-step scale a_i = stepsize sqrt(1 - b2^t) / (1 - b1^t)
 * (step_scale_host: HOST array [nb]).  All of it enqueued by this one call; device pointers otherwise; `scratch`:
 * dm_vf_scratch_bytes(nb, bs) bytes on the device.  epoch_filter = 1: the filter's sums of all nb minibatches are taken up front (two
 * launches) and two launches per minibatch remain (gradient partials; reduction + Adam); 0: three launches per minibatch.  Same arithmetic
 * in the same order either way: bit-identical results. */
int dm_vf_param_count(void);
size_t dm_vf_scratch_bytes(int32_t nb, int32_t bs);
int dm_vf_fit_epoch(const float* ob, const float* ret, int32_t nb, int32_t bs, float* theta, float* adam_m, float* adam_v,
                    const float* step_scale_host, double beta1, double beta2, double eps, double* rms_sum, double* rms_sumsq,
                    double* rms_count, float* rms_mean, float* rms_std, void* scratch, void* hip_stream, int32_t epoch_filter);

/* Replaces: `pi.ob_rms.update(ob)` of src/trpo.py:242 with the whole batch (RunningMeanStd.update, src/utils/misc_util.py:47-70) in one launch:
 * column sums / sums of squares of ob [n, 56] float32 in float64 (fixed order), added to the filter's state; mean / std (float32, the
 * variance floored at 1e-2) refreshed.  Device pointers; `scratch`: dm_rms_scratch_bytes() bytes on the device. */
size_t dm_rms_scratch_bytes(void);
int dm_rms_update(const float* ob, int32_t n, double* rms_sum, double* rms_sumsq, double* rms_count, float* rms_mean, float* rms_std,
                  void* scratch, void* hip_stream);

/* Replaces: the policy half of one TRPO update (src/trpo.py:228-230, 245-283) for the 56-100-100-28 tanh Gaussian policy of
 * src/mlp_policy_trpo.py:50-60.  theta: dm_pg_param_count() floats = polfc1/w, polfc1/b, polfc2/w, polfc2/b, polfinal/w, polfinal/b, logstd
 * (weights row-major [in][out]: the learner's flat `var_list` order, :139); ob [n, 56] float32 raw observations, normalised inside with
 * rms_mean / rms_std [56] and clipped to +-5; everything is a device pointer; `scratch`: dm_pg_scratch_bytes() bytes on the device.
 *   dm_pg_losses  `compute_losses` / `compute_lossandgrad` (:224-226): out_losses[2] (float64) = {surrgain = mean(pnew / pold * atarg),
 *                 meankl = mean KL(old || new)}; with_grad: out_grad = flat gradient of optimgain = surrgain + entcoeff * mean entropy.
 *                 write_old != 0: old == new, and old_mean [n, 28] is WRITTEN (`assign_old_eq_new`, :247); else it is read, with old_logstd [28].
 *   dm_pg_fvp     `compute_fvp` (:228-230) on the samples ob[i * stride], i < n (`fvpargs = arr[::5]`, :245): out_fv = Hessian of the mean KL
 *                 at new == old times v — exactly J^T diag(1 / sigma^2) J v / n on the mean parameters and 2 v on logstd (pg_kernel.h); the
 *                 caller adds cg_damping * v (:229).  One forward-mode and one reverse pass per sample, no second-order graph.
 * Both run one workgroup per CU (their LDS fills it); max_blocks > 0 caps the grid, which leaves the other CUs to a kernel of another stream
 * (the learner runs the value fit beside the policy step that way).  Sums are taken in block order: the result is a function of the grid
 * size, not of timing.  max_blocks = 0: every CU. */
int dm_pg_param_count(void);
size_t dm_pg_scratch_bytes(void);
int dm_pg_losses(const float* ob, int32_t n, const float* ac, const float* atarg, float* old_mean, const float* old_logstd, int32_t write_old,
                 const float* theta, const float* rms_mean, const float* rms_std, double entcoeff, int32_t with_grad,
                 float* out_grad, double* out_losses, void* scratch, void* hip_stream, int32_t max_blocks);
int dm_pg_fvp(const float* ob, int32_t stride, int32_t n, const float* theta, const float* v, const float* rms_mean, const float* rms_std,
              float* out_fv, void* scratch, void* hip_stream, int32_t max_blocks);

/* Diagnostics of DM_OPT_PACKED (four environments per wavefront, csrc/slot_kernel.h): env-steps so far that exceeded a capacity of that
 * path (DM_PACKED_*: rows, contacts / contact pairs, pairs past the bounding spheres, box staging slots; or a PGS step the cost test would
 * reject) and were re-stepped by the one-env code.  out[0] total; out[1..5] by reason: candidates, box slots, contacts, rows, PGS cost test. */
int dm_batch_redo_total(dm_batch* b, int64_t* out /* [8] */);
/* Diagnostics of DM_OPT_STEP_QUEUE: out[0] = horizon launches issued for queued steps so far, out[1] = steps they carried, out[2] = steps queued now. */
int dm_batch_queue_stats(dm_batch* b, int64_t* out /* [3] */);
int dm_batch_sync(dm_batch* b);
/* Run what DM_OPT_STEP_QUEUE has queued and make the batch's stream wait (device-side, no host wait) for every pipelined sub-batch launch in
 * flight (DM_OPT_PIPELINE): afterwards work enqueued on the batch's stream sees the outputs of every earlier dm_batch_step call. */
int dm_batch_join(dm_batch* b);
const char* dm_last_error(void);
int dm_abi_version(void);
/* Arithmetic / device-state type of the loaded library: 64 (libdmenv.so) or 32 (libdmenv32.so, the same source built with
 * -DDM_REAL_FLOAT: SURVEY.md section 8b's `dtype 32`).  Every buffer of this ABI is float64 in both. */
int dm_real_bits(void);
int dm_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* DMENV_H */
