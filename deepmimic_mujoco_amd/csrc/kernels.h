// kernels.h — what the three translation units of libdmenv.so share: the build's arithmetic type and the prototypes of the packed step kernels.
//   dmenv.hip           one-env step kernels, reset / state / ordering / learner kernels, the C ABI (host side)
//   kernels_rollout.hip the horizon launch (k_rollout_packed + the step bodies it calls): the packed options and the load-store vectoriser off
//   kernels_packed.hip  the four-environments-per-wavefront per-step kernels (slot_kernel.h / slot_step.h): compiled with their own backend options
//                       (csrc/build.py PACKED_FLAGS — one wave per SIMD with the whole register file wants a scheduler that goes for
//                       instruction-level parallelism, the two-waves-per-SIMD one-env kernels do not: profiles/r04_ab_kernel_variants.md)
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>

#include "dmenv.h"
#include "policy_kernel.h"
#include "env_step.h"
#include "slot_step.h"

// Arithmetic / device-state type of this build: float64 (libdmenv.so, the parity build) or float32 (libdmenv32.so, -DDM_REAL_FLOAT:
// the `dtype 32` batch of SURVEY.md section 8b — same kernels, half the registers and LDS per env).  Everything that crosses the
// C ABI (actions, observations, rewards, field reads / writes, mocap tables) stays float64 (`Ext`) in both builds.
#ifdef DM_REAL_FLOAT
typedef float Real;
#define DM_STEP_WAVES 3
#else
typedef double Real;
#define DM_STEP_WAVES 2
#endif
typedef double Ext;

#ifndef DM_NARROW_ROWS
#define DM_NARROW_ROWS 32
#endif
constexpr int NARROW_ROWS = DM_NARROW_ROWS;
// The one-env step a packed horizon launch falls back to for an environment beyond the packed path's 32 rows (restep_one_env, slot_step.h): those
// environments hold 33 .. 37 rows (a standing humanoid: 8 foot corners x 4 pyramid edges + joint limits; profiles/r04_ab_kernel_variants.md).  With the
// 32-column register tier every PGS sweep fetches their last columns of A from the memory strip; the wave is alone on its SIMD here, so the tier is 48
// columns: no strip below 49 rows.  Worth 4 % of a re-step (734 k -> 705 k cycles: the one-env step of such an environment is 680 k cycles by itself,
// 300 k of it the 36-row sweeps), 1.4 % of a standing population's horizon.
#ifndef DM_RESTEP_ROWS
#define DM_RESTEP_ROWS 48
#endif
constexpr int RESTEP_ROWS = DM_RESTEP_ROWS;

// the table of per-step buffers of a horizon launch arrives a chunk per launch in the kernel-argument segment (k_put_rows)
constexpr int ROW_CHUNK = 64;
struct StepRowChunk { dm::StepRow r[ROW_CHUNK]; };

// ---- kernels_packed.hip ------------------------------------------------------------------------------------------------------------
__global__ void k_step_packed(const dm::DevModel<Real>* __restrict__ Mp, dm::Batch<Real> B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                              unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count);
__global__ void k_step_packed_act(const dm::DevModel<Real>* __restrict__ Mp, dm::Batch<Real> B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                                  unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count, dmp::PolicyArgs pa);
__global__ void k_step_packed_ext(const dm::DevModel<Real>* __restrict__ Mp, dm::Batch<Real> B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                                  unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count);
__global__ void k_step_packed_act_ext(const dm::DevModel<Real>* __restrict__ Mp, dm::Batch<Real> B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                                      unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count, dmp::PolicyArgs pa);
__global__ void k_rollout_packed(const dm::DevModel<Real>* __restrict__ Mp, const dm::Batch<Real>* __restrict__ Bp, const dm::StepRow* __restrict__ rows, int n_substeps, int first,
                                 int count, int T, dmp::PolicyArgs pa, long long* __restrict__ wave_clk);
__global__ void k_step_packed_prof(const dm::DevModel<Real>* __restrict__ Mp, dm::Batch<Real> B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                                   unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count, long long* __restrict__ prof);
