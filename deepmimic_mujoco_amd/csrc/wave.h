// wave.h — the wave64 primitives the env kernels are written against.
//
// Product build (hipcc, gfx950): thin inline wrappers over the CDNA4 cross-lane builtins.  One 64-lane
// wavefront == one workgroup == one environment, so a workgroup barrier is a single-wave s_barrier.
//
// Test build (-DDM_WAVE_TESTBENCH, g++): tests/emu/ provides the same five primitives on top of 64
// cooperative fibres so that the *identical kernel source* can be checked lane-for-lane against the CPU
// oracle in the GPU-less build container.  That testbench is test infrastructure only; libdmenv.so is
// never built with it and has no CPU execution path.
#pragma once

#if defined(DM_WAVE_TESTBENCH)
#include "wave_testbench.h"  // tests/emu/
#else
#include <hip/hip_runtime.h>

#define DM_DEV __device__ __forceinline__
#define DM_DEV_NOINLINE __device__ __noinline__
#define DM_CONSTANT __device__ constexpr

namespace dmw {
DM_DEV int lane() { return (int)(threadIdx.x & 63u); }
// LDS hand-off between lanes of the (single-wave) workgroup
DM_DEV void sync() { __syncthreads(); }
DM_DEV unsigned long long ballot(bool p) { return __ballot(p); }
DM_DEV int shfl_i(int v, int src) { return __shfl(v, src, 64); }
DM_DEV double shfl(double v, int src) { return __shfl(v, src, 64); }
DM_DEV float shfl(float v, int src) { return __shfl(v, src, 64); }
DM_DEV double shfl_xor(double v, int m) { return __shfl_xor(v, m, 64); }
DM_DEV float shfl_xor(float v, int m) { return __shfl_xor(v, m, 64); }
DM_DEV int shfl_xor_i(int v, int m) { return __shfl_xor(v, m, 64); }
DM_DEV int shfl_up_i(int v, int d) { return __shfl_up(v, d, 64); }
// broadcast from a wave-uniform lane index (v_readlane)
DM_DEV double bcast(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}
DM_DEV float bcast(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
DM_DEV int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
// assert wave-uniformity to the compiler: the value moves to an SGPR, so loops / branches on it are scalar
// (s_cmp + s_cbranch) instead of exec-masked.  Only for values that ARE identical in all lanes.
DM_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
DM_DEV bool uniform(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
// shader clock (s_memtime), for the optional per-stage profile
DM_DEV long long clk() { return (long long)__builtin_readcyclecounter(); }
// compiler scheduling fence: nothing moves across it (no instruction is emitted)
DM_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// compiler-only memory fence: values loaded from LDS before it are not kept live / reused after it.
// (Without it the optimiser CSEs the 310 factor entries across the three triangular solves: 620 live VGPRs.)
// order pins.  pin_zero() is an opaque 0 defined by a volatile asm: adding it to an LDS index makes those loads wait,
// in program order, for every earlier volatile asm; pin_value() is such an asm on a computed value.  Together they
// bound how far the compiler may hoist a block's LDS loads above the arithmetic of the previous block — without them
// the fully unrolled row code below is scheduled "all loads first" and needs >500 VGPRs.
DM_DEV int pin_zero() { int z = 0; asm volatile("" : "+s"(z)); return z; }
DM_DEV void pin_value(double& v) { asm volatile("" : "+v"(v)); }
DM_DEV void pin_value(float& v) { asm volatile("" : "+v"(v)); }
DM_DEV void reload_fence() { asm volatile("" ::: "memory"); }
}  // namespace dmw
#endif

namespace dmw {
// sum over the 64 lanes, result in every lane (butterfly: 6 exchange steps)
template <class R>
DM_DEV R wave_sum(R v) {
  v += shfl_xor(v, 32); v += shfl_xor(v, 16); v += shfl_xor(v, 8);
  v += shfl_xor(v, 4);  v += shfl_xor(v, 2);  v += shfl_xor(v, 1);
  return v;
}
// exclusive prefix sum of small non-negative ints over lanes (Hillis-Steele on shfl_up)
DM_DEV int wave_exclusive_scan(int v, int lane_id, int* total) {
  int x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int y = shfl_up_i(x, d);
    if (lane_id >= d) x += y;
  }
  *total = shfl_i(x, 63);
  return x - v;
}
}  // namespace dmw
