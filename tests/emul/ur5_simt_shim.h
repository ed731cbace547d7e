// ur5_simt_shim.h -- TEST-ONLY. Lets a host compiler build the DEVICE path of csrc/ur5_engine.h (everything under `#ifndef UR5_EMUL`: DPP
// reductions, v_readlane pivots, lane shuffles, ballots, the 8-lanes-per-pair MPR, register-resident factorisations) and run it with the
// semantics of a workgroup of 64-lane wavefronts (one for the small-scene kernel, four for the many-object kernel): every lane is a fibre (ucontext) with its own stack and "registers", LDS is one shared array, and each
// cross-lane operation is a rendezvous of the lanes it involves. The plain lane emulation (ur5sim_emul.cpp, -DUR5_EMUL) runs lanes one after
// another and therefore needs LDS stand-ins for everything that lives in registers on the GPU; this build has no stand-ins.
//
// Rendezvous scopes follow what the hardware instruction can reach: quad_perm / row_half_mirror and shuffles by 1, 2, 4 stay inside an aligned
// group of 8 lanes, row_mirror and shuffles by 8 inside a row of 16, shuffles by 16 inside 32, row broadcasts, v_readlane, general shuffles,
// ballots and barriers span the wave. The engine calls an operation of scope N only where control flow is uniform over the aligned N-lane
// group (the cooperative MPR runs different pairs in different 8-lane sub-groups), which is exactly what the rendezvous requires; a lane
// that waits for a partner which never arrives is reported as a deadlock instead of hanging.
// LDS atomics: executed immediately, then the lane yields, so that lanes running the same loop take turns in lane order as the hardware does
// instruction by instruction (slot claiming with an LDS counter depends on that order).
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __constant__
#define __shared__
#define __restrict__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(n)
#define __HIP_MEMORY_SCOPE_WORKGROUP 0

namespace simt {
constexpr int NL = 256, RING = 16, NCLS = 5;     // up to 256 lanes per workgroup; scope classes: 8, 16, 32, 64 lanes, the whole workgroup
struct Tid { unsigned x, y, z; };
extern Tid tid, bid;
struct Wave {
  ucontext_t sched, ctx[NL];
  char* stack[NL];
  bool done[NL];
  int cur, nl;                                   // running lane, lanes of this workgroup (64 or 256)
  uint64_t slot[NCLS][RING][NL];
  uint32_t seq[NCLS][NL];
  long idle;                                     // consecutive blocked yields (deadlock detector)
  long nbar[NL];                                 // wave barriers each lane has passed in this workgroup run
  long ops[8];                                   // lane 0's operation counts: 0 dpp, 1 readlane, 2 shuffle, 3 ballot, 4 wave barrier, 5 __syncthreads, 6 LDS atomic
  const char* waiting[NL];
};
extern Wave W;
extern int g_order, g_skip_barrier;
void yield_blocked(const char* what);
void yield_runnable();
void run_workgroup(int nlanes, void (*body)(void*), void* arg);

inline int cls_of(int lanes) { return lanes <= 8 ? 0 : lanes <= 16 ? 1 : lanes <= 32 ? 2 : 3; }
// publish `v` for this lane's next operation of scope class `cls` and wait until every live lane of the aligned group has published its own
inline void count(int what) { if (W.cur == 0) W.ops[what]++; }
inline uint32_t publish(int cls, uint64_t v, const char* what) {
  const int me = W.cur, sz = cls == 4 ? W.nl : 8 << cls, base = cls == 4 ? 0 : me & ~(sz - 1);
  const uint32_t k = ++W.seq[cls][me];
  W.slot[cls][k % RING][me] = v;
  for (;;) {
    bool ok = true;
    for (int l = base; l < base + sz; l++) if (!W.done[l] && W.seq[cls][l] < k) { ok = false; break; }
    if (ok) break;
    yield_blocked(what);
  }
  W.idle = 0;
  return k;
}
inline uint64_t peek(int cls, uint32_t k, int lane) { return W.slot[cls][k % RING][lane]; }
// (g_skip_barrier = k: every lane ignores its k-th wave barrier -- a planted race for the detector's self-test)
inline void wave_barrier() { count(4); if (++W.nbar[W.cur] == g_skip_barrier) return; publish(3, 0, "wave barrier"); }
inline void block_barrier() { count(5); publish(4, 0, "__syncthreads"); }
}  // namespace simt

#define threadIdx (simt::tid)
#define blockIdx (simt::bid)

// ---- the builtins / HIP intrinsics the engine's device path uses
inline void __builtin_amdgcn_fence(int, const char*) {}
inline void __builtin_amdgcn_wave_barrier() { simt::wave_barrier(); }
inline void __syncthreads() { simt::block_barrier(); }
inline int __double2loint(double d) { uint64_t u; memcpy(&u, &d, 8); return (int)(uint32_t)u; }
inline int __double2hiint(double d) { uint64_t u; memcpy(&u, &d, 8); return (int)(uint32_t)(u >> 32); }
inline double __hiloint2double(int hi, int lo) { uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double d; memcpy(&d, &u, 8); return d; }
inline int __popcll(unsigned long long m) { return __builtin_popcountll(m); }
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }

// v_mov_b32 dpp: ctrl < 0x100 quad_perm, 0x140 row_mirror, 0x141 row_half_mirror, 0x142 row_bcast:15, 0x143 row_bcast:31
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool /*bound_ctrl*/) {
  const int me = simt::W.cur, wl = me & 63, wb = me & ~63;      // lane inside its wavefront, first lane of the wavefront
  int scope, from = -1;
  if (ctrl < 0x100) { scope = 8; from = (me & ~3) | ((ctrl >> (2 * (me & 3))) & 3); }
  else if (ctrl == 0x141) { scope = 8; from = (me & ~7) | (7 - (me & 7)); }
  else if (ctrl == 0x140) { scope = 16; from = (me & ~15) | (15 - (me & 15)); }
  else if (ctrl == 0x142) { scope = 64; from = wl >= 16 ? (me & ~15) - 1 : -1; }
  else if (ctrl == 0x143) { scope = 64; from = wl >= 32 ? wb + 31 : -1; }
  else { fprintf(stderr, "simt: dpp control 0x%x is not modelled\n", ctrl); abort(); }
  const int cls = simt::cls_of(scope);
  simt::count(0);
  const uint32_t k = simt::publish(cls, (uint64_t)(uint32_t)src, "dpp");
  const bool enabled = (row_mask >> (wl >> 4) & 1) && (bank_mask >> ((wl >> 2) & 3) & 1);
  return enabled && from >= 0 ? (int)(uint32_t)simt::peek(cls, k, from) : old;
}
inline int __builtin_amdgcn_readlane(int v, int src) {
  simt::count(1);
  const uint32_t k = simt::publish(3, (uint64_t)(uint32_t)v, "readlane");
  return (int)(uint32_t)simt::peek(3, k, (simt::W.cur & ~63) | (src & 63));
}
inline int* __builtin_amdgcn_permlane16_swap(int, int, bool, bool) { fprintf(stderr, "simt: permlane16_swap (GS = 32) is not modelled\n"); abort(); }
template <class T> inline T simt_bits_get(uint64_t u) { T t; memcpy(&t, &u, sizeof(T)); return t; }
template <class T> inline uint64_t simt_bits_put(T t) { uint64_t u = 0; memcpy(&u, &t, sizeof(T)); return u; }
template <class T> inline T __shfl(T v, int src, int /*width*/ = 64) {
  simt::count(2);
  const uint32_t k = simt::publish(3, simt_bits_put(v), "shfl");
  return simt_bits_get<T>(simt::peek(3, k, (simt::W.cur & ~63) | (src & 63)));
}
template <class T> inline T __shfl_xor(T v, int o, int /*width*/ = 64) {
  const int cls = simt::cls_of(2 * o);
  simt::count(2);
  const uint32_t k = simt::publish(cls, simt_bits_put(v), "shfl_xor");
  return simt_bits_get<T>(simt::peek(cls, k, simt::W.cur ^ o));
}
inline unsigned long long __ballot(bool p) {
  simt::count(3);
  const uint32_t k = simt::publish(3, p ? 1 : 0, "ballot");
  unsigned long long m = 0;
  const int wb = simt::W.cur & ~63;
  for (int l = 0; l < 64; l++) if (!simt::W.done[wb + l] && simt::peek(3, k, wb + l)) m |= 1ull << l;
  return m;
}
template <class T, class U> inline T __hip_atomic_fetch_add(T* p, U v, int, int) { T o = *p; *p = o + (T)v; simt::count(6); simt::yield_runnable(); return o; }
template <class T, class U> inline T __hip_atomic_fetch_max(T* p, U v, int, int) { T o = *p; if ((T)v > o) *p = (T)v; simt::yield_runnable(); return o; }
template <class T, class U> inline T __hip_atomic_fetch_min(T* p, U v, int, int) { T o = *p; if ((T)v < o) *p = (T)v; simt::yield_runnable(); return o; }
template <class T, class U> inline T __hip_atomic_fetch_or(T* p, U v, int, int) { T o = *p; *p = o | (T)v; simt::yield_runnable(); return o; }
