// kernels_packed.hip — the four-environments-per-wavefront step kernels of libdmenv.so (see kernels.h; built by csrc/build.py with PACKED_FLAGS)
#define DM_NO_LAUNCH_KERNELS
#include "kernels.h"

using namespace dm;

// FOUR environments per wavefront (slot_kernel.h / slot_step.h): workgroup w steps the envs at dispatch positions first + 4 w .. + 3.
// Environments that exceed a capacity of that path are appended to the sub-batch's redo list instead of being stored ...
template <int MAXR>
DM_DEV void step_packed_body(const DevModel<Real>* __restrict__ Mp, const Batch<Real>& B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                             unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count) {
  __shared__ SlotShared<Real> sh[SLOTS];
  __shared__ SlotTables tb;
  const int lane = dmw::lane(), slot = lane >> 4, sl = lane & 15;
  stage_slot_tables(tb, lane);
  const bool live = SLOTS * (int)blockIdx.x + slot < count;
  int envs4[SLOTS];
  dispatch_env<SLOTS>(B, first, count, SLOTS * (int)blockIdx.x, lane, blockIdx.x == 0, envs4);
  const int env = slot == 0 ? envs4[0] : slot == 1 ? envs4[1] : slot == 2 ? envs4[2] : envs4[3];
  slot_env_step<Real, false, false, MAXR>(*Mp, B, sh[slot], tb, env, sl, lane, live, action, obs, reward, done, n_substeps, redo_count, B.redo_list + first);
}
__global__ __launch_bounds__(64) void k_step_packed(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                                    Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                                    int n_substeps, int first, int count, int* __restrict__ redo_count) {
  step_packed_body<2 * SW>(Mp, B, action, obs, reward, done, n_substeps, first, count, redo_count);
}
// The same launch with the three-set code compiled in (DM_OPT_PACKED = 2): 33 .. 40 rows stay in their wave, everything else runs ~8 % slower than in
// k_step_packed (slot_kernel.h slot_forward) — for populations that stand on both feet.  Horizon launches pick per wave-step instead (slot_step.h slot_rollout).
__global__ __launch_bounds__(64) void k_step_packed_ext(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                                        Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                                        int n_substeps, int first, int count, int* __restrict__ redo_count) {
  step_packed_body<SLOT_MAXROWS>(Mp, B, action, obs, reward, done, n_substeps, first, count, redo_count);
}
// ... followed, in the same wave, by the policy's step on the four observations it produced (dm_batch_step_act on the packed path): one
// weight stream per wave serves four environments
template <int MAXR>
DM_DEV void step_packed_act_body(const DevModel<Real>* __restrict__ Mp, const Batch<Real>& B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                                 unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count, const dmp::PolicyArgs& pa) {
  __shared__ SlotShared<Real> sh[SLOTS];
  __shared__ SlotTables tb;
  const int lane = dmw::lane(), slot = lane >> 4, sl = lane & 15;
  stage_slot_tables(tb, lane);
  const bool live = SLOTS * (int)blockIdx.x + slot < count;
  int envs4[SLOTS];
  dispatch_env<SLOTS>(B, first, count, SLOTS * (int)blockIdx.x, lane, blockIdx.x == 0, envs4);
  const int env = slot == 0 ? envs4[0] : slot == 1 ? envs4[1] : slot == 2 ? envs4[2] : envs4[3];
  const bool stored = slot_env_step<Real, false, false, MAXR>(*Mp, B, sh[slot], tb, env, sl, lane, live, action, obs, reward, done, n_substeps, redo_count, B.redo_list + first);
  // s.qpos / s.qvel of every slot hold the state its observation was written from (the fresh episode's after an auto-reset); r1 is free
  // the r1 + r2 regions (adjacent) are free
  static_assert(offsetof(SlotShared<Real>, r2) == offsetof(SlotShared<Real>, r1) + sizeof(sh[0].r1) && sizeof(sh[0].r1) + sizeof(sh[0].r2) >= 464 * sizeof(float), "policy scratch");
  dmw::sync();
  const int envs[4] = {dmw::bcast_i(env, 0), dmw::bcast_i(env, 16), dmw::bcast_i(env, 32), dmw::bcast_i(env, 48)};
  const int st = stored ? 1 : 0;
  const bool wr[4] = {dmw::bcast_i(st, 0) != 0, dmw::bcast_i(st, 16) != 0, dmw::bcast_i(st, 32) != 0, dmw::bcast_i(st, 48) != 0};
  dmp::policy_wave4<Real>(pa, envs, wr, lane, reinterpret_cast<char*>(&sh[0]), (unsigned)sizeof(SlotShared<Real>), (unsigned)(offsetof(SlotShared<Real>, qpos) + 7 * sizeof(Real)),
                          (unsigned)(offsetof(SlotShared<Real>, qvel) + 6 * sizeof(Real)), (unsigned)offsetof(SlotShared<Real>, r1));
}
__global__ __launch_bounds__(64) void k_step_packed_act(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                                        Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                                        int n_substeps, int first, int count, int* __restrict__ redo_count, dmp::PolicyArgs pa) {
  step_packed_act_body<2 * SW>(Mp, B, action, obs, reward, done, n_substeps, first, count, redo_count, pa);
}
__global__ __launch_bounds__(64) void k_step_packed_act_ext(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                                            Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                                            int n_substeps, int first, int count, int* __restrict__ redo_count, dmp::PolicyArgs pa) {
  step_packed_act_body<SLOT_MAXROWS>(Mp, B, action, obs, reward, done, n_substeps, first, count, redo_count, pa);
}
// the same with shader-clock stamps per stage, one record of 16 per wave (DM option 101 with option 105; diagnostic)
__global__ __launch_bounds__(64) void k_step_packed_prof(const DevModel<Real>* __restrict__ Mp, Batch<Real> B, const Ext* __restrict__ action,
                                                         Ext* __restrict__ obs, Ext* __restrict__ reward, unsigned char* __restrict__ done,
                                                         int n_substeps, int first, int count, int* __restrict__ redo_count, long long* __restrict__ prof) {
  __shared__ SlotShared<Real> sh[SLOTS];
  __shared__ SlotTables tb;
  const int lane = dmw::lane(), slot = lane >> 4, sl = lane & 15;
  stage_slot_tables(tb, lane);
  const bool live = SLOTS * (int)blockIdx.x + slot < count;
  int envs4[SLOTS];
  dispatch_env<SLOTS>(B, first, count, SLOTS * (int)blockIdx.x, lane, blockIdx.x == 0, envs4);
  const int env = slot == 0 ? envs4[0] : slot == 1 ? envs4[1] : slot == 2 ? envs4[2] : envs4[3];
  slot_env_step<Real, true>(*Mp, B, sh[slot], tb, env, sl, lane, live, action, obs, reward, done, n_substeps, redo_count, B.redo_list + first, prof + (size_t)blockIdx.x * 32);
}
