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
// a real call from a one-wave workgroup (the callee inherits its callers' register budget: the backend propagates the kernel's work-group size)
// Internal linkage (round 6): with -mllvm -enable-ipra (csrc/build.py PACKED_FLAGS) an internal, non-recursive function none of whose call sites is a `tail` call is
// compiled WITHOUT callee-saved registers — see DM_CALL_SLOT in slot_step.h.  -DDM_NO_STATIC_CALLS: rounds 3-5's external functions (A/B).
#ifndef DM_NO_STATIC_CALLS
#define DM_STATIC_CALLS 1
#define DM_DEV_CALL64 static __device__ __noinline__
#else
#define DM_DEV_CALL64 __device__ __noinline__
#endif
#define DM_CONSTANT __device__ constexpr

namespace dmw {
DM_DEV int lane() { return (int)(threadIdx.x & 63u); }
// LDS hand-off between lanes of the (single-wave) workgroup: a fence at WAVEFRONT scope.  A wave's memory instructions of
// one kind execute in issue order, so a ds_read issued after a ds_write sees its data whichever lane wrote it; the AMDGPU
// memory model therefore implements wavefront-scope synchronisation without any s_waitcnt (the compiler still waits where a
// loaded REGISTER is first used) — it only keeps the compiler from moving memory accesses across the fence.
// (__syncthreads() is a workgroup-scope fence: `s_waitcnt vmcnt(0) lgkmcnt(0)` at every hand-off, i.e. each one drained
// all outstanding LDS traffic and every global load / scratch store in flight.)
DM_DEV void sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// the same with a full workgroup fence: for hand-offs through GLOBAL memory (prologue / epilogue of the step)
DM_DEV void sync_mem() { __syncthreads(); }
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
// DPP lane permutations on doubles (two 32-bit DPP moves; ~8 cycles instead of a ~100-cycle ds_bpermute round trip)
template <int CTRL>
DM_DEV double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
DM_DEV float dpp_f32(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true)); }
DM_DEV float perm_xor1(float v) { return dpp_f32<0xB1>(v); }
DM_DEV float perm_xor2(float v) { return dpp_f32<0x4E>(v); }
DM_DEV float perm_half_mirror(float v) { return dpp_f32<0x141>(v); }
DM_DEV float perm_row_mirror(float v) { return dpp_f32<0x140>(v); }
DM_DEV double perm_xor1(double v) { return dpp_f64<0xB1>(v); }         // quad_perm [1,0,3,2]
DM_DEV double perm_xor2(double v) { return dpp_f64<0x4E>(v); }         // quad_perm [2,3,0,1]
DM_DEV double perm_half_mirror(double v) { return dpp_f64<0x141>(v); } // lane i <-> 7-i within each 8
DM_DEV double perm_row_mirror(double v) { return dpp_f64<0x140>(v); }  // lane i <-> 15-i within each 16
// assert wave-uniformity to the compiler: the value moves to an SGPR, so loops / branches on it are scalar
// (s_cmp + s_cbranch) instead of exec-masked.  Only for values that ARE identical in all lanes.
DM_DEV int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
DM_DEV bool uniform(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
// 1 / x from the hardware estimate and two Newton steps: 5 dependent instructions (a correctly rounded division is 11), within
// an ulp of the quotient.  For the reciprocal pivots of the factorisation, which sit on its serial chain.
DM_DEV double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0); r = __builtin_fma(e, r, r);
  e = __builtin_fma(-x, r, 1.0); r = __builtin_fma(e, r, r);
  return r;
}
DM_DEV float rcp_fast(float x) { return 1.0f / x; }
// D += A B on the matrix core, one 16x16x4 block: lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15]; it holds, in d[v],
// D[mfma_row(l, v)][l & 15].  (v_mfma_f64_16x16x4_f64: row = (l >> 4) + 4 v; the f32 form: row = 4 (l >> 4) + v.)
DM_DEV void mfma_16x16x4(double a, double b, double (&d)[4]) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  v4d c = {d[0], d[1], d[2], d[3]};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  d[0] = c[0]; d[1] = c[1]; d[2] = c[2]; d[3] = c[3];
}
DM_DEV void mfma_16x16x4(float a, float b, float (&d)[4]) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f c = {d[0], d[1], d[2], d[3]};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  d[0] = c[0]; d[1] = c[1]; d[2] = c[2]; d[3] = c[3];
}
DM_DEV int mfma_row(int lane_id, int v, double) { return (lane_id >> 4) + 4 * v; }
DM_DEV int mfma_row(int lane_id, int v, float) { return 4 * (lane_id >> 4) + v; }
// max(a, b) as the bare instruction.  fmax() first quiets each operand (`v_max x, x`) for signalling NaNs: one more
// instruction on the PGS row-to-row chain, where no NaN can be signalling (operands come straight from arithmetic).
DM_DEV double max_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
DM_DEV float max_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
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
// opaque copy of a per-lane integer: index arithmetic derived from it cannot be hoisted out of the enclosing loop or
// shared across stages (LICM/GVN otherwise precompute every stage's lane->index maps at kernel entry and keep them
// live across the whole step: hundreds of VGPRs)
DM_DEV int launder(int v) { asm volatile("" : "+v"(v)); return v; }
// launder() for the slot lane number of the packed kernels (0 .. 15): the optimiser must not fold the value, but may know its range — `d = sl + 16 c < NV`
// is then true at compile time for the first two of three passes over the 34 dofs, and those passes need no lane predicate (no exec-masked block each,
// one scheduling region instead of three: a lone wave's loads of all passes go out together)
DM_DEV int launder_slot_lane(int v) { asm volatile("" : "+v"(v)); __builtin_assume(v >= 0 && v < 16); return v; }
DM_DEV unsigned long long launder(unsigned long long v) { asm volatile("" : "+v"(v)); return v; }
DM_DEV int launder_uniform(int v) { asm volatile("" : "+s"(v)); return v; }   // same for a wave-uniform (SGPR) value
// a pointer that is the same in every lane, told to the compiler (arguments of a called function arrive in vector registers and count as
// divergent: loads through them would be vector loads and their addresses would occupy vector registers)
template <class T> DM_DEV T* uniform_ptr(T* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T*)(((unsigned long long)hi << 32) | lo);
}
// ... and whose address space is told to it as well (a called function sees generic pointers: every access, LDS included, would be a flat one)
template <class T> DM_DEV T* in_lds(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_assume(__builtin_amdgcn_is_shared((const void*)p));
#endif
  return p;
}
template <class T> DM_DEV T* in_global(T* p) {      // wave-uniform pointers into global memory (through an opaque copy: a plain cast pair would be folded away before the address spaces are inferred)
  __attribute__((address_space(1))) T* g = (__attribute__((address_space(1))) T*)uniform_ptr(p);
  asm volatile("" : "+s"(g));
  return (T*)g;
}
// ... into global memory that NOTHING writes while the kernel runs (model, batch descriptor): the constant address space makes every
// wave-uniform load through it a scalar load again (inside a called function a plain global pointer may alias the function's own stores, so
// the compiler turns them into vector loads: 58 of them in the packed step, each a ~600-cycle stall of a lone wave)
template <class T> DM_DEV const T* in_constant(const T* p) {
  __attribute__((address_space(4))) const T* g = (__attribute__((address_space(4))) const T*)uniform_ptr(p);
  asm volatile("" : "+s"(g));
  return (const T*)g;
}
template <class T> DM_DEV T* launder_uniform_ptr(T* p) { asm volatile("" : "+s"(p)); return p; }   // loads through the result cannot be hoisted above this point
DM_DEV void reload_fence() { asm volatile("" ::: "memory"); }
// if (pred) *p -= v in LDS as one fire-and-forget ds_add_f64: lanes of one instruction may hit the same address (the
// LDS applies them one after the other), and nothing comes back, so the issuing lane does not wait for the old value.
// Called by all lanes (the testbench implements it as a collective).
DM_DEV void lds_sub(bool pred, double* p, double v) { if (pred) __hip_atomic_fetch_add(p, -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DM_DEV void lds_sub(bool pred, float* p, float v) { if (pred) __hip_atomic_fetch_add(p, -v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// ---- 16-lane env slots (slot_kernel.h: four environments per wavefront, one DPP row each) -------------------------------------
// row broadcast: every lane gets the value of lane I of ITS OWN 16-lane row — one v_mov_b64_dpp row_newbcast (gfx90a+), for all
// four rows (= four environments) at once.  I is a compile-time lane number.
template <int I> DM_DEV double row_bcast(double v) {
  long long x = __double_as_longlong(v);
  x = __builtin_amdgcn_update_dpp(0ll, x, 0x150 + I, 0xf, 0xf, true);
  return __longlong_as_double(x);
}
template <int I> DM_DEV float row_bcast(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + I, 0xf, 0xf, true)); }
template <int I> DM_DEV int row_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + I, 0xf, 0xf, true); }
// acc += (lane I of the row's x) * y as ONE instruction: v_fmac_f64 with a DPP source (the only f64 arithmetic with a DPP form on
// gfx950).  The leading s_nop covers the VALU-write -> DPP-read hazard, which the compiler cannot see through inline assembly.
template <int I> DM_DEV void row_fmac(double& acc, double x, double y) {
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(I));
}
template <int I> DM_DEV void row_fmac(float& acc, float x, float y) { acc += row_bcast<I>(x) * y; }
// the same without the hazard slot, for a DPP source x that was written long before: only between a dpp_settle() and the next write
// of any such x.  volatile: these keep their order among themselves and after dpp_settle().
template <int I> DM_DEV void row_fmac_old(double& acc, double x, double y) {
  // (the hazard slot stays: a register that is "old" in the source may still have been refilled from an accumulation register or from
  //  scratch by the instruction before — the compiler's hazard recogniser does not look inside inline assembly)
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y), "n"(I));
}
template <int I> DM_DEV void row_fmac_old(float& acc, float x, float y) { acc += row_bcast<I>(x) * y; }
// eight of them behind ONE hazard slot: the operands are in their registers before the block starts (whatever refills the compiler
// inserted come first), so the single s_nop covers all eight.   acc0 += sum of the even k, acc1 += sum of the odd k of bcast_I(x[k]) * y[k]
#define DM_FMAC_DPP(acc, x, y) "v_fmac_f64_dpp " acc ", " x ", " y " row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
template <int I> DM_DEV void row_fmac8(double& acc0, double& acc1, const double* x, const double* y) {
  asm volatile("s_nop 1\n\t" DM_FMAC_DPP("%0", "%2", "%10") DM_FMAC_DPP("%1", "%3", "%11") DM_FMAC_DPP("%0", "%4", "%12") DM_FMAC_DPP("%1", "%5", "%13")
               DM_FMAC_DPP("%0", "%6", "%14") DM_FMAC_DPP("%1", "%7", "%15") DM_FMAC_DPP("%0", "%8", "%16") DM_FMAC_DPP("%1", "%9", "%17")
               : "+v"(acc0), "+v"(acc1)
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
                 "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]), "v"(y[4]), "v"(y[5]), "v"(y[6]), "v"(y[7]), "n"(I));
}
#undef DM_FMAC_DPP
// a[k] += bcast_I(x[k]) * one for eight accumulators, one hazard slot
#define DM_FMAC_DPP(acc, x) "v_fmac_f64_dpp " acc ", " x ", %16 row_newbcast:%17 row_mask:0xf bank_mask:0xf\n\t"
template <int I> DM_DEV void row_add8(double* a, const double* x, double one) {
  asm volatile("s_nop 1\n\t" DM_FMAC_DPP("%0", "%8") DM_FMAC_DPP("%1", "%9") DM_FMAC_DPP("%2", "%10") DM_FMAC_DPP("%3", "%11")
               DM_FMAC_DPP("%4", "%12") DM_FMAC_DPP("%5", "%13") DM_FMAC_DPP("%6", "%14") DM_FMAC_DPP("%7", "%15")
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(one), "n"(I));
}
#undef DM_FMAC_DPP
template <int I> DM_DEV void row_fmac8(float& acc0, float& acc1, const float* x, const float* y) { for (int k = 0; k < 8; k++) { if (k & 1) acc1 += row_bcast<I>(x[k]) * y[k]; else acc0 += row_bcast<I>(x[k]) * y[k]; } }
template <int I> DM_DEV void row_add8(float* a, const float* x, float one) { for (int k = 0; k < 8; k++) a[k] += row_bcast<I>(x[k]) * one; }
// One row of the packed Gauss-Seidel sweep as a single block (slot_kernel.h):  delta = max(nf0, t);  tsave += onehot * t  (lane I of the
// row keeps the residual it saw at its own row: onehot is 1 there and 0 elsewhere);  t += bcast_I(delta) * a.  The fused multiply-add and a
// one-cycle s_nop between the write of delta and its DPP read are exactly the two wait states that hazard needs.
template <int I> DM_DEV void pgs_row(double& t, double& tsave, double nf0, double a, double onehot) {
  double delta;
  asm volatile("v_max_f64 %0, %3, %1\n\tv_fma_f64 %2, %5, %1, %2\n\ts_nop 0\n\tv_fmac_f64_dpp %1, %0, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf"
               : "=&v"(delta), "+v"(t), "+v"(tsave) : "v"(nf0), "v"(a), "v"(onehot), "n"(I));
}
// two row sets: the row belongs to set `own` (t_own, tsave_own); both sets' residuals are updated
template <int I> DM_DEV void pgs_row2(double& t_own, double& tsave_own, double& t_other, double nf0, double a_own, double a_other, double onehot) {
  double delta;
  asm volatile("v_max_f64 %0, %4, %1\n\tv_fma_f64 %2, %7, %1, %2\n\ts_nop 0\n\tv_fmac_f64_dpp %1, %0, %5 row_newbcast:%8 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f64_dpp %3, %0, %6 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
               : "=&v"(delta), "+v"(t_own), "+v"(tsave_own), "+v"(t_other) : "v"(nf0), "v"(a_own), "v"(a_other), "v"(onehot), "n"(I));
}
// a surplus row (slot_kernel.h slot_constraint<3>): its step also moves the residuals of both full row sets
template <int I> DM_DEV void pgs_row3(double& t3, double& tsave3, double& t0, double& t1, double nf0, double b, double u0, double u1, double onehot) {
  double delta;
  asm volatile("v_max_f64 %0, %5, %1\n\tv_fma_f64 %2, %9, %1, %2\n\ts_nop 0\n\tv_fmac_f64_dpp %1, %0, %6 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
               "v_fmac_f64_dpp %3, %0, %7 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %4, %0, %8 row_newbcast:%10 row_mask:0xf bank_mask:0xf"
               : "=&v"(delta), "+v"(t3), "+v"(tsave3), "+v"(t0), "+v"(t1) : "v"(nf0), "v"(b), "v"(u0), "v"(u1), "v"(onehot), "n"(I));
}
template <int I> DM_DEV void pgs_row3(float& t3, float& tsave3, float& t0, float& t1, float nf0, float b, float u0, float u1, float onehot) {
  const float d = max_raw(nf0, t3); tsave3 += onehot * t3; const float bc = row_bcast<I>(d); t3 += bc * b; t0 += bc * u0; t1 += bc * u1;
}
// FOUR consecutive rows I0 .. I0 + 3 as one block (round 5).  The compiler cannot see inside inline assembly, so it fenced every one-row block with a hazard
// slot of its own (one s_nop per row on top of the block's own); and in the two-set form the second multiply-add of a row — the OTHER set's residuals, which
// nothing reads before the next set boundary — can sit in the NEXT row's hazard slot instead of an s_nop.  Same operations on the same operands in the same
// order per accumulator: bit-identical results; 16 / 17 instructions per four rows instead of 20 / 24.
//   one set:   per row  v_max d, nf0, t ; v_fma ts += oh t ; s_nop 0 ; v_fmac_dpp t += bcast(d) a
#define DM_PGS1(D, A, OH, I) "v_max_f64 " D ", %3, %0\n\tv_fma_f64 %1, " OH ", %0, %1\n\ts_nop 0\n\tv_fmac_f64_dpp %0, " D ", " A " row_newbcast:" I " row_mask:0xf bank_mask:0xf\n\t"
template <int I0> DM_DEV void pgs_rows4(double& t, double& tsave, double nf0, const double* a, const double* oh) {
  double d;
  asm volatile(DM_PGS1("%2", "%4", "%8", "%12") DM_PGS1("%2", "%5", "%9", "%13") DM_PGS1("%2", "%6", "%10", "%14") DM_PGS1("%2", "%7", "%11", "%15")
               : "+v"(t), "+v"(tsave), "=&v"(d)
               : "v"(nf0), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(oh[0]), "v"(oh[1]), "v"(oh[2]), "v"(oh[3]), "n"(I0), "n"(I0 + 1), "n"(I0 + 2), "n"(I0 + 3));
}
#undef DM_PGS1
//   two sets:  row i   v_max d_i, nf0, t ; v_fma ts += oh t ; [v_fmac_dpp t_other += bcast(d_(i-1)) a_other(i-1)  |  s_nop 0 for the first] ; v_fmac_dpp t += bcast(d_i) a_own(i)
//              then the last row's other-set multiply-add.  d alternates between two registers.
template <int I0> DM_DEV void pgs_rows4_2(double& t_own, double& tsave_own, double& t_other, double nf0, const double* a_own, const double* a_other, const double* oh) {
  double d0, d1;
  asm volatile(
      "v_max_f64 %3, %5, %0\n\tv_fma_f64 %1, %14, %0, %1\n\ts_nop 0\n\tv_fmac_f64_dpp %0, %3, %6 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f64 %4, %5, %0\n\tv_fma_f64 %1, %15, %0, %1\n\tv_fmac_f64_dpp %2, %3, %10 row_newbcast:%18 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %4, %7 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f64 %3, %5, %0\n\tv_fma_f64 %1, %16, %0, %1\n\tv_fmac_f64_dpp %2, %4, %11 row_newbcast:%19 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %3, %8 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\t"
      "v_max_f64 %4, %5, %0\n\tv_fma_f64 %1, %17, %0, %1\n\tv_fmac_f64_dpp %2, %3, %12 row_newbcast:%20 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %0, %4, %9 row_newbcast:%21 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\tv_fmac_f64_dpp %2, %4, %13 row_newbcast:%21 row_mask:0xf bank_mask:0xf"
      : "+v"(t_own), "+v"(tsave_own), "+v"(t_other), "=&v"(d0), "=&v"(d1)
      : "v"(nf0), "v"(a_own[0]), "v"(a_own[1]), "v"(a_own[2]), "v"(a_own[3]), "v"(a_other[0]), "v"(a_other[1]), "v"(a_other[2]), "v"(a_other[3]),
        "v"(oh[0]), "v"(oh[1]), "v"(oh[2]), "v"(oh[3]), "n"(I0), "n"(I0 + 1), "n"(I0 + 2), "n"(I0 + 3));
}
template <int I> DM_DEV void pgs_row(float& t, float& tsave, float nf0, float a, float onehot) { const float d = max_raw(nf0, t); tsave += onehot * t; t += row_bcast<I>(d) * a; }
template <int I> DM_DEV void pgs_row2(float& t_own, float& tsave_own, float& t_other, float nf0, float a_own, float a_other, float onehot) {
  const float d = max_raw(nf0, t_own); tsave_own += onehot * t_own; const float b = row_bcast<I>(d); t_own += b * a_own; t_other += b * a_other;
}
template <int I0> DM_DEV void pgs_rows4(float& t, float& tsave, float nf0, const float* a, const float* oh) {
  pgs_row<I0>(t, tsave, nf0, a[0], oh[0]); pgs_row<I0 + 1>(t, tsave, nf0, a[1], oh[1]); pgs_row<I0 + 2>(t, tsave, nf0, a[2], oh[2]); pgs_row<I0 + 3>(t, tsave, nf0, a[3], oh[3]);
}
template <int I0> DM_DEV void pgs_rows4_2(float& t_own, float& tsave_own, float& t_other, float nf0, const float* a_own, const float* a_other, const float* oh) {
  pgs_row2<I0>(t_own, tsave_own, t_other, nf0, a_own[0], a_other[0], oh[0]); pgs_row2<I0 + 1>(t_own, tsave_own, t_other, nf0, a_own[1], a_other[1], oh[1]);
  pgs_row2<I0 + 2>(t_own, tsave_own, t_other, nf0, a_own[2], a_other[2], oh[2]); pgs_row2<I0 + 3>(t_own, tsave_own, t_other, nf0, a_own[3], a_other[3], oh[3]);
}
DM_DEV void dpp_settle() { asm volatile("s_nop 4"); }
// A double parked in two ACCUMULATION registers across a region in which the architectural registers are needed for something hotter (the
// register allocator does this on its own, but per USE: inside a loop that means one v_accvgpr_read per operand and trip).  park() / unpark()
// move it once each way; between the two the value occupies no architectural register.
struct Parked { int lo, hi; };
DM_DEV Parked park(double v) {
  Parked p;
  asm volatile("v_accvgpr_write_b32 %0, %2\n\tv_accvgpr_write_b32 %1, %3" : "=a"(p.lo), "=a"(p.hi) : "v"(__double2loint(v)), "v"(__double2hiint(v)));
  return p;
}
DM_DEV double unpark(const Parked& p) {
  int lo, hi;
  asm volatile("v_accvgpr_read_b32 %0, %2\n\tv_accvgpr_read_b32 %1, %3" : "=v"(lo), "=v"(hi) : "a"(p.lo), "a"(p.hi));
  return __hiloint2double(hi, lo);
}
struct ParkedF { int x; };
DM_DEV ParkedF park(float v) { ParkedF p; asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(p.x) : "v"(__float_as_int(v))); return p; }
DM_DEV float unpark(const ParkedF& p) { int x; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(p.x)); return __int_as_float(x); }
// counter in global memory shared by all waves of a launch: returns the value before the increment
DM_DEV int global_counter_next(int* p) { return atomicAdd(p, 1); }
// this lane's 16 bits of a wave ballot (bit i = lane i of the own row)
DM_DEV unsigned row_ballot(bool p, int lane_id) { return (unsigned)((__ballot(p) >> (lane_id & 48)) & 0xffffull); }
}  // namespace dmw
#endif

namespace dmw {
// sums over lane groups, result in every lane of the group (no LDS traffic)
DM_DEV double sum8(double v) { v += perm_xor1(v); v += perm_xor2(v); v += perm_half_mirror(v); return v; }
DM_DEV double sum16(double v) { v = sum8(v); v += perm_row_mirror(v); return v; }
DM_DEV double wave_sum(double v) {
  v = sum16(v);
  return ((bcast(v, 0) + bcast(v, 16)) + bcast(v, 32)) + bcast(v, 48);
}
#if !defined(DM_WAVE_TESTBENCH)
DM_DEV float sum8(float v) { v += perm_xor1(v); v += perm_xor2(v); v += perm_half_mirror(v); return v; }
DM_DEV float sum16(float v) { v = sum8(v); v += perm_row_mirror(v); return v; }
DM_DEV float wave_sum(float v) {
  v = sum16(v);
  return ((bcast(v, 0) + bcast(v, 16)) + bcast(v, 32)) + bcast(v, 48);
}
#endif
// exclusive prefix sum over lanes of a small non-negative int (< 32): one ballot + popcount per bit instead of a
// 6-step shuffle scan (each shuffle is an LDS-crossbar round trip)
DM_DEV int wave_exclusive_scan(int v, int lane_id, int* total) {
  const unsigned long long below = (1ull << lane_id) - 1ull;
  int pre = 0, tot = 0;
#pragma unroll
  for (int b = 0; b < 5; b++) {
    const unsigned long long m = ballot(((v >> b) & 1) != 0);
    pre += __builtin_popcountll(m & below) << b;
    tot += __builtin_popcountll(m) << b;
  }
  *total = tot;
  return pre;
}
}  // namespace dmw
