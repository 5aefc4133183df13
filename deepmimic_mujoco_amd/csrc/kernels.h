// kernels.h — what the two translation units of libdmenv.so share: the build's arithmetic type and the prototypes of the packed step kernels.
//   dmenv.hip           one-env step kernels, reset / state / ordering / learner kernels, the C ABI (host side)
//   kernels_packed.hip  the four-environments-per-wavefront kernels (slot_kernel.h / slot_step.h): compiled with their own backend options
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

// the table of per-step buffers of a horizon launch arrives a chunk per launch in the kernel-argument segment (k_put_rows)
constexpr int ROW_CHUNK = 64;
struct StepRowChunk { dm::StepRow r[ROW_CHUNK]; };

// ---- kernels_packed.hip ------------------------------------------------------------------------------------------------------------
__global__ void k_step_packed(const dm::DevModel<Real>* __restrict__ Mp, dm::Batch<Real> B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                              unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count);
__global__ void k_step_packed_act(const dm::DevModel<Real>* __restrict__ Mp, dm::Batch<Real> B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                                  unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count, dmp::PolicyArgs pa);
__global__ void k_rollout_packed(const dm::DevModel<Real>* __restrict__ Mp, const dm::Batch<Real>* __restrict__ Bp, const dm::StepRow* __restrict__ rows, int n_substeps, int first,
                                 int count, int T, dmp::PolicyArgs pa, long long* __restrict__ wave_clk);
__global__ void k_step_packed_prof(const dm::DevModel<Real>* __restrict__ Mp, dm::Batch<Real> B, const Ext* __restrict__ action, Ext* __restrict__ obs, Ext* __restrict__ reward,
                                   unsigned char* __restrict__ done, int n_substeps, int first, int count, int* __restrict__ redo_count, long long* __restrict__ prof);
