// wave_testbench.h — TEST INFRASTRUCTURE.  Runs the unmodified kernel source (deepmimic_mujoco_amd/csrc/env_*.h)
// on the CPU of the GPU-less build container by giving each of the 64 lanes of a wavefront its own cooperative
// fibre (ucontext).  Cross-lane primitives (sync / shfl / ballot / bcast) are rendezvous points; between them a
// fibre runs alone, so any missing barrier or divergent collective in the kernel shows up as a wrong result or a
// deadlock here, before GPU time is spent.  Never linked into libdmenv.so; the product has no CPU path.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#define DM_DEV inline
#define DM_DEV_NOINLINE inline
#define DM_DEV_CALL64 inline
#define DM_CONSTANT constexpr

namespace dmw {

struct WaveBench {
  int cur_lane;
  int arrived;
  unsigned long long gen;
  unsigned long long slot[2][64];
  void* aptr[64]; double aval[64];   // operands of a wave-wide LDS atomic (lds_sub)
  void (*yield_fn)(void);
};
WaveBench& bench();

inline int lane() { return bench().cur_lane; }

inline void rendezvous() {
  WaveBench& b = bench();
  const unsigned long long g = b.gen;
  if (++b.arrived == 64) { b.arrived = 0; b.gen++; }
  else while (b.gen == g) b.yield_fn();
}
inline void sync() { rendezvous(); }
inline void sync_mem() { rendezvous(); }

template <class T> inline unsigned long long to_bits(T v) { unsigned long long u = 0; std::memcpy(&u, &v, sizeof(T)); return u; }
template <class T> inline T from_bits(unsigned long long u) { T v; std::memcpy(&v, &u, sizeof(T)); return v; }

template <class T> inline T exchange(T v, int src) {
  WaveBench& b = bench();
  const int p = (int)(b.gen & 1ull);
  b.slot[p][b.cur_lane] = to_bits(v);
  rendezvous();
  return from_bits<T>(bench().slot[p][src & 63]);
}
inline unsigned long long ballot(bool pr) {
  WaveBench& b = bench();
  const int p = (int)(b.gen & 1ull);
  b.slot[p][b.cur_lane] = pr ? 1ull : 0ull;
  rendezvous();
  unsigned long long m = 0;
  for (int l = 0; l < 64; l++) m |= (bench().slot[p][l] & 1ull) << l;
  return m;
}
inline int shfl_i(int v, int src) { return exchange(v, src); }
inline double shfl(double v, int src) { return exchange(v, src); }
inline float shfl(float v, int src) { return exchange(v, src); }
inline double shfl_xor(double v, int m) { return exchange(v, lane() ^ m); }
inline float shfl_xor(float v, int m) { return exchange(v, lane() ^ m); }
inline int shfl_xor_i(int v, int m) { return exchange(v, lane() ^ m); }
inline int shfl_up_i(int v, int d) { const int l = lane(); return exchange(v, l >= d ? l - d : l); }
inline double bcast(double v, int src) { return exchange(v, src); }
inline float bcast(float v, int src) { return exchange(v, src); }
inline int bcast_i(int v, int src) { return exchange(v, src); }
// one 16x16x4 matrix-core block (see wave.h): a collective; every fibre gathers the operands of all lanes, then forms its own four
// results with the k index innermost
template <class T> inline void mfma_16x16x4(T a, T b, T (&d)[4]) {
  T A[64], B[64];
  {
    WaveBench& w = bench();
    const int p = (int)(w.gen & 1ull);
    w.slot[p][w.cur_lane] = to_bits(a);
    rendezvous();
    for (int l = 0; l < 64; l++) A[l] = from_bits<T>(bench().slot[p][l]);
  }
  {
    WaveBench& w = bench();
    const int p = (int)(w.gen & 1ull);
    w.slot[p][w.cur_lane] = to_bits(b);
    rendezvous();
    for (int l = 0; l < 64; l++) B[l] = from_bits<T>(bench().slot[p][l]);
  }
  const int l = lane();
  for (int v = 0; v < 4; v++) {
    const int row = (l >> 4) + 4 * v, col = l & 15;
    T acc = d[v];
    for (int k = 0; k < 4; k++) acc = std::fma(A[row + 16 * k], B[col + 16 * k], acc);
    d[v] = acc;
  }
}
template <class T> inline int mfma_row(int lane_id, int v, T) { return (lane_id >> 4) + 4 * v; }
inline long long clk() { return 0; }
template <class T> inline T max_raw(T a, T b) { return std::fmax(a, b); }
template <class T> inline T rcp_fast(T x) { return T(1) / x; }
inline int uniform(int v) { return v; }
inline bool uniform(bool v) { return v; }
inline double perm_xor1(double v) { return exchange(v, lane() ^ 1); }
inline double perm_xor2(double v) { return exchange(v, lane() ^ 2); }
inline double perm_half_mirror(double v) { const int l = lane(); return exchange(v, (l & ~7) | (7 - (l & 7))); }
inline double perm_row_mirror(double v) { const int l = lane(); return exchange(v, (l & ~15) | (15 - (l & 15))); }
inline void sched_fence() {}
inline void reload_fence() {}
// wave-wide LDS atomic: a collective here (all 64 fibres call it, `pred` says who takes part) so that updates of one
// address are applied in lane order whatever order the fibres happen to run in; the device adds the ROUNDED operand
template <class T> inline void lds_sub(bool pred, T* p, T v) {
  WaveBench& b = bench();
  b.aptr[b.cur_lane] = pred ? (void*)p : nullptr; b.aval[b.cur_lane] = (double)v;
  rendezvous();
  if (lane() == 0) for (int l = 0; l < 64; l++) if (bench().aptr[l]) *(T*)bench().aptr[l] -= (T)bench().aval[l];
  rendezvous();
}
// 16-lane env slots (wave.h): row-relative broadcast / fused multiply-add / ballot
template <int I, class T> inline T row_bcast(T v) { return exchange(v, (lane() & 48) | I); }
template <int I> inline int row_bcast_i(int v) { return exchange(v, (lane() & 48) | I); }
// (unfused like every other multiply-add of the testbench build, -ffp-contract=off: paths that must agree bit for bit then do so here as on the device)
template <int I, class T> inline void row_fmac(T& acc, T x, T y) { acc = acc + row_bcast<I>(x) * y; }
template <int I, class T> inline void row_fmac_old(T& acc, T x, T y) { acc = acc + row_bcast<I>(x) * y; }
template <int I, class T> inline void row_fmac8(T& acc0, T& acc1, const T* x, const T* y) {
  for (int k = 0; k < 8; k++) { if (k & 1) acc1 = acc1 + row_bcast<I>(x[k]) * y[k]; else acc0 = acc0 + row_bcast<I>(x[k]) * y[k]; }
}
template <int I, class T> inline void row_add8(T* a, const T* x, T one) { for (int k = 0; k < 8; k++) a[k] = a[k] + row_bcast<I>(x[k]) * one; }
template <int I, class T> inline void pgs_row(T& t, T& tsave, T nf0, T a, T onehot) {
  const T d = std::fmax(nf0, t); tsave = tsave + onehot * t; t = t + row_bcast<I>(d) * a;
}
template <int I, class T> inline void pgs_row2(T& t_own, T& tsave_own, T& t_other, T nf0, T a_own, T a_other, T onehot) {
  const T d = std::fmax(nf0, t_own); tsave_own = tsave_own + onehot * t_own; const T b = row_bcast<I>(d);
  t_own = t_own + b * a_own; t_other = t_other + b * a_other;
}
template <int I0, class T> inline void pgs_rows4(T& t, T& tsave, T nf0, const T* a, const T* oh) {
  pgs_row<I0>(t, tsave, nf0, a[0], oh[0]); pgs_row<I0 + 1>(t, tsave, nf0, a[1], oh[1]); pgs_row<I0 + 2>(t, tsave, nf0, a[2], oh[2]); pgs_row<I0 + 3>(t, tsave, nf0, a[3], oh[3]);
}
template <int I0, class T> inline void pgs_rows4_2(T& t_own, T& tsave_own, T& t_other, T nf0, const T* a_own, const T* a_other, const T* oh) {
  pgs_row2<I0>(t_own, tsave_own, t_other, nf0, a_own[0], a_other[0], oh[0]); pgs_row2<I0 + 1>(t_own, tsave_own, t_other, nf0, a_own[1], a_other[1], oh[1]);
  pgs_row2<I0 + 2>(t_own, tsave_own, t_other, nf0, a_own[2], a_other[2], oh[2]); pgs_row2<I0 + 3>(t_own, tsave_own, t_other, nf0, a_own[3], a_other[3], oh[3]);
}
template <int I, class T> inline void pgs_row3(T& t3, T& tsave3, T& t0, T& t1, T nf0, T b, T u0, T u1, T onehot) {
  const T d = std::fmax(nf0, t3); tsave3 = tsave3 + onehot * t3; const T bc = row_bcast<I>(d);
  t3 = t3 + bc * b; t0 = t0 + bc * u0; t1 = t1 + bc * u1;
}
inline void dpp_settle() {}
// register parking (wave.h): a value copy here
template <class T> struct ParkedT { T v; };
typedef ParkedT<double> Parked;
typedef ParkedT<float> ParkedF;
inline Parked park(double v) { return Parked{v}; }
inline ParkedF park(float v) { return ParkedF{v}; }
template <class T> inline T unpark(const ParkedT<T>& p) { return p.v; }
inline int global_counter_next(int* p) { return (*p)++; }
inline unsigned row_ballot(bool p, int lane_id) { return (unsigned)((ballot(p) >> (lane_id & 48)) & 0xffffull); }
inline int pin_zero() { return 0; }
inline int launder(int v) { return v; }
inline int launder_slot_lane(int v) { return v; }
inline unsigned long long launder(unsigned long long v) { return v; }
inline int launder_uniform(int v) { return v; }
template <class T> inline T* launder_uniform_ptr(T* p) { return p; }
template <class T> inline T* uniform_ptr(T* p) { return p; }
template <class T> inline T* in_lds(T* p) { return p; }
template <class T> inline T* in_global(T* p) { return p; }
template <class T> inline const T* in_constant(const T* p) { return p; }
inline void pin_value(double&) {}
inline void pin_value(float&) {}

}  // namespace dmw

using std::fabs; using std::fmax; using std::sqrt; using std::sin; using std::cos; using std::pow; using std::exp;
