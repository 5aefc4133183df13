// emu_main.cpp — TEST INFRASTRUCTURE: C entry points that run the HIP kernel source on the fibre wave testbench.
// Mirrors the subset of include/dmenv.h the parity tests use, one environment after another.
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <algorithm>
#include <vector>

#include "wave_testbench.h"
// kernel source, unmodified
#include "env_step.h"
#include "slot_step.h"
#include "model_host.h"

namespace dmw {
static WaveBench g_bench;
WaveBench& bench() { return g_bench; }
}  // namespace dmw

namespace {
constexpr size_t STACK = 1 << 20;
ucontext_t g_main, g_fib[64];
bool g_done[64];
std::function<void(int)>* g_body;
char* g_stacks;

void yield_to_main() { swapcontext(&g_fib[dmw::g_bench.cur_lane], &g_main); }
void fibre_entry() {
  const int l = dmw::g_bench.cur_lane;
  (*g_body)(l);
  g_done[l] = true;
  swapcontext(&g_fib[l], &g_main);
}
// run body(lane) on 64 fibres to completion
void run_wave(std::function<void(int)> body) {
  if (!g_stacks) g_stacks = (char*)malloc(STACK * 64);
  g_body = &body;
  dmw::g_bench.arrived = 0; dmw::g_bench.gen = 0; dmw::g_bench.yield_fn = yield_to_main;
  for (int l = 0; l < 64; l++) {
    g_done[l] = false;
    getcontext(&g_fib[l]);
    g_fib[l].uc_stack.ss_sp = g_stacks + STACK * l; g_fib[l].uc_stack.ss_size = STACK; g_fib[l].uc_link = &g_main;
    makecontext(&g_fib[l], fibre_entry, 0);
  }
  for (;;) {
    bool all = true;
    for (int l = 0; l < 64; l++) if (!g_done[l]) { all = false; dmw::g_bench.cur_lane = l; swapcontext(&g_main, &g_fib[l]); }
    if (all) break;
  }
}

using namespace dm;
struct EmuBatch {
  DevModel<double> M;
  Batch<double> B;
  std::vector<double> qpos, qvel, qws, time, ctrl, xipos, comz, cfg, vel, aovf, imit, kin;
  std::vector<unsigned char> kin_ok;
  bool two_tier = true, packed = false, packed_ext = false;   // packed_ext: DM_OPT_PACKED = 2 (per-step launches with the three-set code: k_step_packed_ext)
  SlotShared<double> slots[SLOTS];
  SlotOrOne<double> roll;          // emu_rollout: the one-env code's LDS ALIASES the four slots', as in k_rollout_packed
  SlotTables slot_tabs;
  std::vector<int> fidx, finit, ncon, nefc, cong, status, siter, episode, cycle, redo;
  long redo_total = 0;
  Shared<double> sh;
  StepScratch<double> xs;
};
}  // namespace

extern "C" {
void* emu_create(const dm_model_desc* d, const double* cfg, const double* vel, int F, int n, unsigned flags, double mocap_dt) {
  EmuBatch* e = new EmuBatch();
  std::string err;
  if (build_dev_model(d, &e->M, &err) != 0) { fprintf(stderr, "emu_create: %s\n", err.c_str()); delete e; return nullptr; }
  e->M.enable_contact = (flags & DM_FLAG_NO_CONTACT) ? 0 : 1;
  e->M.enable_limit = (flags & DM_FLAG_NO_LIMIT) ? 0 : 1;
  e->qpos.assign((size_t)n * NQ, 0); e->qvel.assign((size_t)n * NV, 0); e->qws.assign((size_t)n * NV, 0);
  e->time.assign(n, 0); e->ctrl.assign((size_t)n * NU, 0); e->xipos.assign((size_t)n * NB * 3, 0); e->comz.assign(n, 0);
  e->cfg.assign(cfg, cfg + (size_t)F * NQ); e->vel.assign(vel, vel + (size_t)F * NV);
  e->fidx.assign(n, 0); e->finit.assign(n, 0); e->ncon.assign(n, 0); e->nefc.assign(n, 0); e->cong.assign((size_t)n * MAXEFC * 2, -1);
  e->status.assign(n, 0); e->siter.assign(n, 0); e->episode.assign(n, 0); e->cycle.assign(n, 0); e->aovf.assign((size_t)n * AOVF_COLS * 64, 0);
  for (int i = 0; i < n; i++) for (int k = 0; k < NQ; k++) e->qpos[(size_t)i * NQ + k] = e->M.qpos0[k];
  Batch<double>& B = e->B;
  B.qpos = e->qpos.data(); B.qvel = e->qvel.data(); B.qws = e->qws.data(); B.time = e->time.data(); B.ctrl = e->ctrl.data();
  B.xipos = e->xipos.data(); B.comz = e->comz.data(); B.frame_idx = e->fidx.data(); B.frame_init = e->finit.data();
  B.ncon = e->ncon.data(); B.nefc = e->nefc.data(); B.cong = e->cong.data(); B.status = e->status.data();
  B.aovf = e->aovf.data(); B.cycle = e->cycle.data(); B.imit_table = nullptr; B.imit_pdev = nullptr; B.order = nullptr;
  B.solver_iter = e->siter.data(); B.episode = e->episode.data(); B.mocap_cfg = e->cfg.data(); B.mocap_vel = e->vel.data();
  e->kin.assign((size_t)n * KIN_DOUBLES, 0); e->kin_ok.assign(n, 0); B.kin = e->kin.data(); B.kin_ok = e->kin_ok.data();
  e->redo.assign((size_t)n + 16, 0); B.redo_list = e->redo.data() + 16; B.redo_count = e->redo.data(); B.redo_why = e->redo.data() + 8;
  B.mocap_dt = mocap_dt; B.n_frames = F; B.n_envs = n; B.env_offset = 0; B.reward_mode = 0; B.autoreset = 0; B.action_mode = 0; B.seed = 0; B.diag = 1;
  return e;
}
void emu_set_imitation(void* h, const double* table, const double* params) {
  EmuBatch* e = (EmuBatch*)h;
  e->imit.assign(table, table + (size_t)e->B.n_frames * IMIT_FEAT);
  e->B.imit_table = e->imit.data();
  for (int k = 0; k < 32; k++) e->B.imit_params[k] = params[k];
  e->B.imit_pdev = e->B.imit_params;
}
void emu_destroy(void* h) { delete (EmuBatch*)h; }
void emu_invalidate_kin(void* h) { EmuBatch* e = (EmuBatch*)h; std::fill(e->kin_ok.begin(), e->kin_ok.end(), 0); }   // as the device entry points do
void emu_set_option(void* h, int opt, long long v) {
  EmuBatch* e = (EmuBatch*)h;
  if (opt == DM_OPT_REWARD_MODE) e->B.reward_mode = (int)v;
  else if (opt == DM_OPT_AUTORESET) e->B.autoreset = (int)v;
  else if (opt == DM_OPT_ACTION_MODE) e->B.action_mode = (int)v;
  else if (opt == DM_OPT_SEED) e->B.seed = (unsigned long long)v;
  else if (opt == DM_OPT_DIAGNOSTICS) e->B.diag = v != 0;
  else if (opt == 100) e->B.env_offset = (int)v;
  else if (opt == 102) e->two_tier = v != 0;
  else if (opt == 105 || opt == DM_OPT_PACKED) { e->packed = v != 0; e->packed_ext = v == 2; }
  else if (opt == 103) e->M.pgs_detect = v ? -1e300 : 1e-10;
}
void* emu_field(void* h, int field) {
  EmuBatch* e = (EmuBatch*)h;
  switch (field) {
    case DM_F_QPOS: return e->qpos.data(); case DM_F_QVEL: return e->qvel.data(); case DM_F_QACC_WARMSTART: return e->qws.data();
    case DM_F_TIME: return e->time.data(); case DM_F_FRAME_IDX: return e->fidx.data(); case DM_F_FRAME_INIT: return e->finit.data();
    case DM_F_XIPOS: return e->xipos.data(); case DM_F_COM_Z: return e->comz.data(); case DM_F_NCON: return e->ncon.data();
    case DM_F_NEFC: return e->nefc.data(); case DM_F_CONTACT_GEOMS: return e->cong.data(); case DM_F_STATUS: return e->status.data();
    case DM_F_SOLVER_ITER: return e->siter.data(); case DM_F_CTRL: return e->ctrl.data(); case DM_F_EPISODE: return e->episode.data();
    case DM_F_CYCLE: return e->cycle.data();
  }
  return nullptr;
}
void emu_step(void* h, const double* action, double* obs, double* reward, unsigned char* done, int nsub) {
  EmuBatch* e = (EmuBatch*)h;
  if (e->packed && e->B.reward_mode <= 4) {          // the device's routing (dmenv.hip step_impl): four envs per wave, overflowing envs re-stepped one per wave
    const int n = e->B.n_envs;
    e->B.redo_count[0] = 0;
    for (int first = 0; first < n; first += SLOTS)
      run_wave([&](int lane) {
        const int slot = lane >> 4, sl = lane & 15;
        stage_slot_tables(e->slot_tabs, lane);
        int pos = first + slot;
        const bool live = pos < n;
        if (!live) pos = n - 1;
        if (e->packed_ext) slot_env_step<double, false, false, SLOT_MAXROWS>(e->M, e->B, e->slots[slot], e->slot_tabs, pos, sl, lane, live, action, obs, reward, done, nsub, e->B.redo_count, e->B.redo_list);
        else slot_env_step<double>(e->M, e->B, e->slots[slot], e->slot_tabs, pos, sl, lane, live, action, obs, reward, done, nsub, e->B.redo_count, e->B.redo_list);
      });
    e->redo_total += e->B.redo_count[0];
    for (int i = 0; i < e->B.redo_count[0]; i++) {
      const int env = e->B.redo_list[i];
      run_wave([&](int lane) { env_step<double, 32>(e->M, e->B, e->sh, e->xs, env, lane, action, obs, reward, done, nsub); });
    }
    return;
  }
  for (int env = 0; env < e->B.n_envs; env++)
  {
    // same scheme as the device: register tier (32 columns of A here, so that the overflow strip is exercised by every
    // evaluation with more than 32 rows) or, with option 102 = 0, all 64 columns in registers
    if (e->two_tier) run_wave([&](int lane) { env_step<double, 32>(e->M, e->B, e->sh, e->xs, env, lane, action, obs, reward, done, nsub); });
    else run_wave([&](int lane) { env_step<double, MAXEFC>(e->M, e->B, e->sh, e->xs, env, lane, action, obs, reward, done, nsub); });
  }
}
// dm_batch_rollout on the packed path, open loop (k_rollout_packed without a policy): action [T, n, 28], obs [T, n, 56], reward / done [T, n]
void emu_rollout(void* h, const double* action, double* obs, double* reward, unsigned char* done, int nsub, int T) {
  EmuBatch* e = (EmuBatch*)h;
  const int n = e->B.n_envs;
  const int before = e->B.redo_why[0];
  std::vector<StepRow> rows((size_t)T);               // the horizon's table of per-step buffers (k_fill_rows)
  for (int t = 0; t < T; t++) rows[t] = StepRow{action + (size_t)t * n * NU, obs + (size_t)t * n * NOBS, reward + (size_t)t * n, done + (size_t)t * n};
  for (int first = 0; first < n; first += SLOTS)
    run_wave([&](int lane) {
      const int slot = lane >> 4;
      stage_slot_tables(e->slot_tabs, lane);
      int pos = first + slot;
      const bool live = pos < n;
      if (!live) pos = n - 1;
      slot_rollout<double, 48 /* = csrc/kernels.h RESTEP_ROWS: the in-wave re-step's register tier */>(e->M, e->B, e->roll.sh, e->slot_tabs, e->roll.one.s, e->roll.one.x, pos, lane, live, rows.data(), nsub, T, [](int) {});
    });
  e->redo_total += e->B.redo_why[0] - before;
}
long emu_redo_total(void* h) { return ((EmuBatch*)h)->redo_total; }
// Self-ordering per-step launches (env_step.h order_ticket / dispatch_env) without the physics: `count` envs first .. first + count - 1 of a batch of n take
// their tickets in the order `arrival` names them (n_arrival of them; keys from nefc / iter), then the waves of the next launch look their envs up — one env per wave
// (out1[count]) and four per wave (out4[4 * ceil(count / 4)], spare slots repeat the last position).  with_tickets = 0: no tickets -> `order` / identity.
void emu_dispatch(int n, int first, int count, const int* nefc, const int* iter, const int* arrival, int n_arrival, const int* order, int with_tickets, int* out1, int* out4) {
  std::vector<int> cnt(3 * ORD_BUCKETS, 0), list((size_t)3 * ORD_BUCKETS * n, -1);
  Batch<double> B{};
  B.order = const_cast<int*>(order);
  B.ord_stride = n;
  B.ord_out = cnt.data() + ORD_BUCKETS; B.ordl_out = list.data() + (size_t)ORD_BUCKETS * n + first;
  if (with_tickets) for (int i = 0; i < n_arrival; i++) { const int e = arrival[i]; order_ticket(B, e, nefc[e], iter[e]); }
  B.ord_in = with_tickets ? cnt.data() + ORD_BUCKETS : nullptr; B.ordl_in = list.data() + (size_t)ORD_BUCKETS * n + first;
  B.ord_out = cnt.data() + 2 * ORD_BUCKETS; B.ordl_out = list.data() + (size_t)2 * ORD_BUCKETS * n + first;
  B.ord_zero = cnt.data();
  for (int k = 0; k < ORD_BUCKETS; k++) cnt[k] = 12345;              // the phase the launch's first workgroup clears
  for (int w = 0; w < count; w++)
    run_wave([&](int lane) { int e; dispatch_env<1>(B, first, count, w, lane, w == 0, &e); if (lane == 0) out1[w] = e; });
  for (int k = 0; k < ORD_BUCKETS; k++) if (cnt[k] != 0) out1[0] = -1000 - k;   // (reported as a bad env)
  for (int w = 0; w * SLOTS < count; w++)
    run_wave([&](int lane) { int e[SLOTS]; dispatch_env<SLOTS>(B, first, count, SLOTS * w, lane, false, e); if (lane == 0) for (int j = 0; j < SLOTS; j++) out4[SLOTS * w + j] = e[j]; });
}
void emu_set_state(void* h, const double* qpos, const double* qvel, const int* fidx, const unsigned char* mask) {
  EmuBatch* e = (EmuBatch*)h;
  std::fill(e->kin_ok.begin(), e->kin_ok.end(), 0);
  for (int env = 0; env < e->B.n_envs; env++) {
    if (mask && !mask[env]) continue;
    run_wave([&](int lane) {
      load_env(e->M, e->B, e->sh, env, lane, (const double*)0);
      if (lane < NQ) e->sh.qpos[lane] = qpos[(size_t)env * NQ + lane];
      if (lane < NV) e->sh.qvel[lane] = qvel[(size_t)env * NV + lane];
      if (fidx && lane == 0) set_frame(e->B, env, fidx[env]);
      dmw::sync();
      store_state(e->B, e->sh, env, lane);
      { const LaneTopo lt = lane_topo(lane); stage_tables(e->sh, lane); dmw::sync(); forward(e->M, e->sh, lane, lt, (const DebugOut*)0); }
      store_derived(e->B, e->M, e->sh, env, lane);
    });
  }
}
void emu_reset(void* h, int mode, int hard, const unsigned char* mask) {
  EmuBatch* e = (EmuBatch*)h;
  std::fill(e->kin_ok.begin(), e->kin_ok.end(), 0);
  for (int env = 0; env < e->B.n_envs; env++) {
    if (mask && !mask[env]) continue;
    run_wave([&](int lane) {
      load_env(e->M, e->B, e->sh, env, lane, (const double*)0);
      reset_env(e->M, e->B, e->sh, env, lane, mode, hard);
      store_state(e->B, e->sh, env, lane);
      { const LaneTopo lt = lane_topo(lane); stage_tables(e->sh, lane); dmw::sync(); forward(e->M, e->sh, lane, lt, (const DebugOut*)0); }
      store_derived(e->B, e->M, e->sh, env, lane);
    });
  }
}
void emu_debug_forward(void* h, int env, double* out) {
  EmuBatch* e = (EmuBatch*)h;
  for (int i = 0; i < DM_DEBUG_DOUBLES; i++) out[i] = 0;
  run_wave([&](int lane) {
    load_env(e->M, e->B, e->sh, env, lane, (const double*)0);
    if (lane < NU) { const int d = lane + 6; e->sh.act[d] = e->M.gear[d] * clampr(e->B.ctrl[(size_t)env * NU + lane], e->M.ctrl_lo[d], e->M.ctrl_hi[d]); }
    dmw::sync();
    DebugOut dbg{out};
    { const LaneTopo lt = lane_topo(lane); stage_tables(e->sh, lane); dmw::sync(); forward(e->M, e->sh, lane, lt, &dbg); }
    store_derived(e->B, e->M, e->sh, env, lane);
  });
}
}
