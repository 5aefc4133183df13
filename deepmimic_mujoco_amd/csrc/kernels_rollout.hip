// kernels_rollout.hip — the horizon launch of libdmenv.so: k_rollout_packed and the step bodies it calls (slot_step.h slot_rollout).  A translation unit of its
// own since round 6 because it wants one backend option more than the per-step packed kernels (csrc/build.py ROLLOUT_FLAGS): with the IR-level load-store
// vectoriser off the horizon launch is 0.8-1.4 % faster (four same-call comparisons), the per-step kernel 1 % slower (profiles/r06_ab_kernel_variants.md section 15).
// Memory-operation merging does not touch the arithmetic: the launch forms stay bit-identical (tests/test_gpu_queue.py).
#define DM_NO_LAUNCH_KERNELS
#include "kernels.h"

using namespace dm;

// A whole horizon of T steps in ONE launch (dm_batch_rollout; slot_step.h slot_rollout): every wave steps its four environments T times
// without waiting for any other wave — optionally with the policy's step in between (pa.P; pa.action = the [T + 1, N, 28] action rows,
// pa.vpred = the [T, N] value rows, pa.counter = the first step's draw counter) — and re-steps an environment that exceeds a capacity of
// the packed path itself, with the one-env code, in the LDS the slots leave free between two steps.
__global__ __launch_bounds__(64) void k_rollout_packed(const DevModel<Real>* __restrict__ Mp, const Batch<Real>* __restrict__ Bp, const StepRow* __restrict__ rows,
                                                       int n_substeps, int first, int count, int T, dmp::PolicyArgs pa, long long* __restrict__ wave_clk) {
  __shared__ SlotOrOne<Real> u;
  __shared__ SlotTables tb;
  static_assert(sizeof(SlotOrOne<Real>) == sizeof(SlotShared<Real>) * SLOTS, "the one-env code's LDS fits into the four slots'");
  // the policy step's scratch (464 floats per env at the slot's r1): inside r1 in the float64 build, so a slot's kinematics (xpos, xmat, cdof, r2) survive
  // it; the float32 build's r1 is smaller and the scratch runs into r2: no kinematics carried over a policy step there (slot_env_step kin_carry)
  const bool policy_clobbers_kin = pa.P != nullptr && sizeof(((SlotShared<Real>*)0)->r1) < 464 * sizeof(float);
  const Batch<Real>& B = *Bp;       // (in device memory, not a by-value argument: the called step functions are handed its address)
  const int lane = dmw::lane(), slot = lane >> 4;
  stage_slot_tables(tb, lane);
  const bool live = SLOTS * (int)blockIdx.x + slot < count;
  int envs4[SLOTS];
  dispatch_env<SLOTS>(B, first, count, SLOTS * (int)blockIdx.x, lane, blockIdx.x == 0, envs4);
  const int env = slot == 0 ? envs4[0] : slot == 1 ? envs4[1] : slot == 2 ? envs4[2] : envs4[3];
  const int envs[4] = {dmw::bcast_i(env, 0), dmw::bcast_i(env, 16), dmw::bcast_i(env, 32), dmw::bcast_i(env, 48)};
  const int lv = live ? 1 : 0;
  const bool wr[4] = {dmw::bcast_i(lv, 0) != 0, dmw::bcast_i(lv, 16) != 0, dmw::bcast_i(lv, 32) != 0, dmw::bcast_i(lv, 48) != 0};
  const size_t n = (size_t)B.n_envs;
  const long long t_enter = wave_clk ? dmw::clk() : 0;
#ifdef DM_ROLLOUT_PROF
  long long* prof_acc = wave_clk ? wave_clk + (size_t)blockIdx.x * 4 * dm::PROF_SLOTS : (long long*)nullptr;   // four envs' records per wave: [0..31] sums, [32..63] scratch
  if (prof_acc && lane < 64) prof_acc[lane] = 0;
  dmw::sync_mem();
#else
  long long* prof_acc = nullptr;
#endif
  slot_rollout<Real, RESTEP_ROWS>(*Mp, B, u.sh, tb, u.one.s, u.one.x, env, lane, live, rows, n_substeps, T, [&](int t) {
    if (!pa.P) return;
    dmp::PolicyArgs p = pa;
    p.action = pa.action + (size_t)(t + 1) * n * NU; p.vpred = pa.vpred + (size_t)t * n; p.counter = pa.counter + (unsigned long long)t;
    dmw::sync();
    dmp::policy_wave4<Real>(p, envs, wr, lane, reinterpret_cast<char*>(&u.sh[0]), (unsigned)sizeof(SlotShared<Real>), (unsigned)(offsetof(SlotShared<Real>, qpos) + 7 * sizeof(Real)),
                            (unsigned)(offsetof(SlotShared<Real>, qvel) + 6 * sizeof(Real)), (unsigned)offsetof(SlotShared<Real>, r1));
  }, prof_acc, policy_clobbers_kin);
  // diagnostic (DM option 101): shader-clock cycles this wave spent on its horizon, slot 5 ("total") of workgroup w's profile record
#ifdef DM_ROLLOUT_PROF
  if (prof_acc && lane == 0) prof_acc[31] = dmw::clk() - t_enter;        // (this build: per-stage sums in [0..30] of the wave's first record, the horizon's total in [31])
#else
  if (wave_clk && lane == 0) wave_clk[(size_t)blockIdx.x * dm::PROF_SLOTS + 5] = dmw::clk() - t_enter;
#endif
}
