// ur5_engine.h -- UR5 grasp-scene engine, device code shared by both kernels of libur5sim.so (ur5sim.hip: one 64-lane wavefront
// per scene for up to 6 objects; ur5sim_many.hip: -DUR5_MANY, one 256-thread workgroup per scene for 40-object piles).
//
// Replaces, for a whole batch of scenes, what the reference does one scene at a time through mujoco_py [3P]:
//   sim.step()                      gym_grasper/controller/MujocoController.py:379  -> Engine::step()
//   PID.__call__ / ctrl writes      MujocoController.py:325-327                     -> Engine::pid_and_deltas()
//   move_group_to_joint_target      MujocoController.py:269-393                     -> the move loop in Engine::run()
//   stay / open / close / grasp     MujocoController.py:408-444, 621-636            -> script opcodes in Engine::run()
//   move_ee / ik                    MujocoController.py:446-517                     -> Engine::ik() + the move loop
//   move_and_grasp                  gym_grasper/envs/GraspingEnv.py:205-386         -> the grasp script in Engine::run()
//
// Execution model: ONE workgroup (a single wavefront in the small-scene build) owns ONE scene. All per-scene state lives in LDS (struct Lds) for the whole
// launch -- thousands of 2 ms physics steps -- and is read/written from HBM once. The code is a sequence of *phases*:
// PAR(i, n) distributes n independent items over the 64 lanes, SYNC() separates phases, WAVE_SUM/WAVE_MAX combine lane
// partials. Statements outside PAR are wave-uniform (every lane computes the same value from LDS).
//
// Contacts act on bodies through wrenches and bodies map twists to dofs ("twist space"): a contact never sees more than
// two 6-vectors, the robot's 8x8 block and each object's 6x6 block of the Newton Hessian are assembled from per-body
// 6x6 accumulators, and only contacts between two movable bodies add coupling blocks. Solver = MuJoCo's default
// Newton on the primal with exact line search [3P], same as oracle/ur5_oracle.cpp solve_newton().
//
// -DUR5_EMUL compiles the same source for the host with lanes run sequentially; tests use that build to check the kernel
// logic without a GPU. It is never loaded by the package (see DESIGN.md "Lane emulation").
#pragma once
#include <math.h>
#include <stddef.h>
#include "ur5_devmodel.h"

#ifdef UR5_EMUL
#define UR5_FN inline
#define UR5_BIG inline
#define UR5_CALL inline
#define UR5_ATOMIC_ADD(p, v) (*(p) += (v))
#define UR5_ATOMIC_MAX(p, v) (*(p) = *(p) > (v) ? *(p) : (v))
#define UR5_MPR_ATTR inline
#define UR5_BOXBOX_ATTR inline
#define UR5_PHASE_A inline
#define UR5_PHASE_B inline
#define UR5_PHASE_C inline
#define UR5_PHASE_D inline
#define UR5_PHASE_E inline
#define UR5_PHASE_F inline
#define UR5_PHASE_G inline
#define UR5_PHASE_H inline
static void* ur5_emul_lds = nullptr;
static const Ur5DevModel* ur5_emul_model = nullptr;
#define UR5_LDS_PTR(T) (static_cast<T*>(ur5_emul_lds))
#define UR5_MODEL (*ur5_emul_model)
#define PAR(i, n) for (int i = 0; i < (n); ++i)
#define SYNC() ((void)0)
#define WAVE_SUM(v) (v)
#define WAVE_MAX(v) (v)
#define UR5_LANE 0
#define UR5_GBASE 0
#else
#ifdef UR5_SIMT   // test-only: a host build of THIS (device) code path, intrinsics from tests/emul/ur5_simt_shim.h (one fibre per lane)
#include "ur5_simt_shim.h"
#else
#include <hip/hip_runtime.h>
#endif
#define UR5_FN __device__ __forceinline__
#define UR5_BIG __device__ __forceinline__  // phase routines: the interpreter in run() calls each of them from one place
#define UR5_CALL __device__ __noinline__    // small helpers with many call sites: kept as real functions
#define UR5_ATOMIC_ADD(p, v) __hip_atomic_fetch_add((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define UR5_ATOMIC_MAX(p, v) __hip_atomic_fetch_max((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#ifndef UR5_MPR_ATTR
#define UR5_MPR_ATTR UR5_BIG
#endif
// Phase routines with one call site each. In the wavefront-per-scene kernel (256-register cap, 2 waves per SIMD) the register-
// hungry ones are real functions: each then gets its own register allocation instead of sharing one 50 k-instruction
// function body with every other phase, which cuts the spill traffic (+30 % env-steps/s measured, same-box A/B of the
// inline/noinline combinations); the rest stay inlined. The many-object kernel has 512 registers per lane and inlines all.
// (Engine::FLAT decides per instantiation: the bodies below are always inlinable, each has a real-function wrapper next to it.) The kernels
// that own all 512 registers of a lane -- two scenes per wavefront, and the many-object kernel -- inline everything.
#define UR5_PHASE_A UR5_BIG   // collision
#define UR5_PHASE_B UR5_BIG   // make_constraints
#define UR5_PHASE_C UR5_BIG   // solve_newton
#define UR5_PHASE_D UR5_BIG   // kinematics
#define UR5_PHASE_H UR5_BIG   // newton_direction
#ifndef UR5_PHASE_E
#define UR5_PHASE_E UR5_BIG    // crb_and_factor, velocity_stage, integrate: they share the register-resident robot factors
#endif
#ifndef UR5_PHASE_F
#define UR5_PHASE_F UR5_BIG
#endif
#ifndef UR5_PHASE_G
#define UR5_PHASE_G UR5_BIG
#endif
#ifndef UR5_PHASE_C
#define UR5_PHASE_C UR5_BIG
#endif
#ifndef UR5_BOXBOX_ATTR
#define UR5_BOXBOX_ATTR UR5_BIG
#endif
// The scene lives in dynamic LDS and the model in constant memory, both reached through these file-scope symbols so that
// every (non-inlined) phase routine addresses them with ds_* / s_load instead of flat instructions.
extern __shared__ __attribute__((aligned(16))) double ur5_smem[];
__constant__ Ur5DevModel ur5_cmodel;
// A scene is owned by a GROUP of GS lanes (Engine<real, NV, GS>): GS = 64 is one wavefront per scene, GS = 32 packs two scenes into a
// wavefront (64 / GS scenes per 64-thread workgroup, each with its own Lds image), GS = UR5_NT = 256 is the many-object variant.
// UR5_LANE is the lane's index inside its group, UR5_GBASE the workgroup thread index of the group's first lane. Everything that is
// "wave-uniform" in the one-scene-per-wave layout is group-uniform here; the compiler masks the lanes of a scene whose control flow
// differs from its wave neighbour's (scenes never exchange data, so the only cost of divergence is the idle half-wave).
#define UR5_LDS_PTR(T) (reinterpret_cast<T*>(ur5_smem) + (GS == UR5_NT ? 0 : (int)threadIdx.x / GS))
#define UR5_MODEL ur5_cmodel
#define PAR(i, n) for (int i = UR5_LANE; i < (n); i += GS)
// SYNC orders the LDS traffic of the lanes that share a scene. With one wavefront per workgroup (UR5_NT == 64) the hardware already
// executes a wave's LDS instructions in issue order, so all that is needed is that the COMPILER keeps the accesses on their side of
// the line: wavefront-scope fences and a scheduling barrier, no instruction. __syncthreads() in a 64-thread workgroup costs an
// `s_waitcnt lgkmcnt(0)` -- a full drain of the LDS queue and of every scalar load in flight -- at each of the several hundred SYNCs of a step.
#if UR5_NT == 64 && !defined(UR5_BLOCK_SYNC)
#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#else
#define SYNC() __syncthreads()
#endif
#if defined(UR5_OPAQUE_LANE) && !defined(UR5_SIMT)
// build option (+1 % without step_fn in round 2, to be re-measured): the lane index through an empty volatile asm, so that the optimiser can neither hoist
// lane-derived LDS addresses and predicates out of loops (where they end up in scratch) nor share them between phases
__device__ __forceinline__ int ur5_lane_opaque() { int l = (int)threadIdx.x; asm volatile("" : "+v"(l)); return l; }
#define UR5_LANE (ur5_lane_opaque() & (GS - 1))
#else
#define UR5_LANE ((int)threadIdx.x & (GS - 1))
#endif
#define UR5_GBASE ((int)threadIdx.x & ~(GS - 1))
// Wave-wide sum / max with DPP (data-parallel primitives: the adder reads a neighbour lane's register directly) instead of
// __shfl_xor, which goes through the LDS crossbar (ds_bpermute) six times per value: quad swaps, row mirrors, then the two
// row broadcasts of GFX9 leave the total in lane 63, which v_readlane hands to every lane.
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double ur5_dpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double ur5_lane63(double v) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
template <class T> __device__ __forceinline__ T ur5_wave_sum(T v0) {
  double v = (double)v0;
  v += ur5_dpp<0xb1, 0xf>(v);    // quad_perm [1,0,3,2]
  v += ur5_dpp<0x4e, 0xf>(v);    // quad_perm [2,3,0,1]
  v += ur5_dpp<0x141, 0xf>(v);   // row_half_mirror
  v += ur5_dpp<0x140, 0xf>(v);   // row_mirror: every lane of a 16-lane row holds the row sum
  v += ur5_dpp<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3 (other rows add the 0 of `old`)
  v += ur5_dpp<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3: lane 63 = total
  return (T)ur5_lane63(v);
}
// the same for a 32-lane group (two scenes per wavefront): the four in-row steps, then v_permlane16_swap (gfx950) exchanges row 1 of one
// copy with row 0 of the other (rows 3 / 2 likewise), so copy + copy = the sum of the group's two rows in each of its lanes
template <class T> __device__ __forceinline__ T ur5_half_sum(T v0) {
  double v = (double)v0;
  v += ur5_dpp<0xb1, 0xf>(v);
  v += ur5_dpp<0x4e, 0xf>(v);
  v += ur5_dpp<0x141, 0xf>(v);
  v += ur5_dpp<0x140, 0xf>(v);
  const int lo = __double2loint(v), hi = __double2hiint(v);
  auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return (T)(__hiloint2double(rh[0], rl[0]) + __hiloint2double(rh[1], rl[1]));
}
template <int W, class T> __device__ __forceinline__ T ur5_wave_max(T v) {   // max over aligned groups of W lanes
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) { T w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return v;
}
// sums / maxima over the lanes of a scene: one wavefront (GS = 64), half of one (GS = 32) or several (Engine::block_sum / block_max
// combine the per-wave results through LDS in a fixed order)
#define WAVE_SUM(v) group_sum(v)
#define WAVE_MAX(v) group_max(v)
#endif

// optional per-phase cycle accounting (-DUR5_PROFILE builds libur5sim_prof.so; never defined for the product library)
#if defined(UR5_PROFILE) && !defined(UR5_EMUL)
#define PROF_T0() unsigned long long prof_t_ = __builtin_readcyclecounter()
#define PROF_RE() prof_t_ = __builtin_readcyclecounter()
#define PROF(id) do { unsigned long long n_ = __builtin_readcyclecounter(); if (UR5_LANE == 0) S.prof[id] += (double)(n_ - prof_t_); prof_t_ = n_; } while (0)
#else
#define PROF_T0() ((void)0)
#define PROF_RE() ((void)0)
#define PROF(id) ((void)0)
#endif
// -DUR5_PROFILE_LEVELS (many-object profile builds): the sub-interval slots x0..x5 book the level loops of the envelope factorisation / solves instead of the
// sub-intervals of `rows` and of the narrow phase: x1 uncoupled + unreached blocks at once, x2 panel rows (A1, with the forward substitution) + barrier,
// x3 write-back + trailing update + barrier, x4 terminal blocks, x5 forward sweep of the solves that reuse a factor, x0 backward sweep
#if defined(UR5_PROFILE_LEVELS) && defined(UR5_MANY)
#define PROFR(id) ((void)0)
#define PROFL_T0() PROF_T0()
#define PROFL(id) PROF(id)
#else
#define PROFR(id) PROF(id)
#define PROFL_T0() ((void)0)
#define PROFL(id) ((void)0)
#endif
enum { PF_KIN = 0, PF_CRB, PF_VEL, PF_BROAD, PF_NARROW, PF_ROWS, PF_NEWTON_INIT, PF_IMAGES, PF_LINESEARCH, PF_GRADG, PF_HASM, PF_CHOL, PF_SOLVE,
       PF_INTEGRATE, PF_PID, PF_IK, PF_CORECLK, PF_REALCLK, PF_X0, PF_X1, PF_X2, PF_X3, PF_X4, PF_X5, PF_X6, PF_X7, PF_COUNT };   // PF_X7: MPR pairs (count, not cycles)   // the last two: start / end of the scene's wave in 100 MHz ticks (s_memrealtime)

#ifndef UR5_FORCE_GLOBAL_ENV
#define UR5_FORCE_GLOBAL_ENV 0   // experiment: 1 = the envelope always lives in the scene's global-memory scratch (what a smaller LDS image would cost)
#endif
#ifndef UR5_STG_LDS
#define UR5_STG_LDS 1            // 0 = the contact sides' wrench / Hessian terms are staged in the global scratch even when they fit the LDS pool (A/B)
#endif
#ifndef UR5_DCACHE_LDS
#define UR5_DCACHE_LDS 1         // 0 = the factored diagonal blocks stay in the global scratch even when the envelope is in LDS (A/B)
#endif
#ifndef UR5_SUP_K
#define UR5_SUP_K 4   // hull vertices per lane and trip of the cooperative support scan
#endif
#ifndef UR5_SUP_DELTA
#define UR5_SUP_DELTA 0.02   // travel (m) of a moving geom after which the broad phase's pair superset is rebuilt
#endif
#ifndef UR5_MPR_W
#define UR5_MPR_W 8   // lanes that share one hull pair in the cooperative MPR pass (8: eight pairs per wavefront in flight; 16: four pairs, half the trips per scan)
#endif
#ifndef UR5_INL_POW
#define UR5_INL_POW 1
#endif
#ifndef UR5_INL_IMAGES
#define UR5_INL_IMAGES 1
#endif
#ifndef UR5_INL_MATVEC
#define UR5_INL_MATVEC 1
#endif
#ifndef UR5_INL_COST
#define UR5_INL_COST 1
#endif
#ifndef UR5_INL_DUMP
#define UR5_INL_DUMP 0   // the introspection dump (FORWARD op only) stays a real function: inlined into the flat kernel it made the kernel fault (gfx950, ROCm 7.2)
#endif
#ifndef UR5_EMUL
namespace {   // internal linkage for every engine function: lets -enable-ipra drop the callee-saved register spills of the phase functions
#endif
namespace ur5 {

enum { RES_NONE = -1, RES_SUCCESS = 0, RES_MAX_STEPS = 1, RES_IK_FAIL = 2 };
constexpr int NB = UR5_NB;  // base directions per contact: normal, 2 tangents, torsion (+ 2 rolling directions for condim 6)

// ---------------------------------------------------------------------------------------------- small maths
// Many-object kernel: geometry is computed WITHOUT fused multiply-adds -- the vector helpers below and (UR5_STRICT, first statement of their bodies) the
// kinematics and collision routines. The oracle is built with -ffp-contract=off; Minkowski portal refinement turns a last-bit difference of its inputs into
// another portal face now and then (a 5e-3 jump of a contact normal), and piles of cylinders are full of the degenerate configurations where that happens.
// With identical arithmetic the kernel reproduces the oracle's contacts from the same state instead of its own variant of them. The wavefront-per-scene
// kernel keeps contraction: its scenes have few such pairs, and kinematics + collision are a third of its step.
#if (defined(UR5_MANY) || defined(UR5_STRICT_SMALL)) && !defined(UR5_EMUL) && !defined(UR5_SIMT)
#pragma clang fp contract(off)
#define UR5_STRICT _Pragma("clang fp contract(off)")
#else
#define UR5_STRICT
#endif
template <class T> struct V3 {
  T x, y, z;
  UR5_FN V3() : x(0), y(0), z(0) {}
  UR5_FN V3(T a, T b, T c) : x(a), y(b), z(c) {}
  template <class U> UR5_FN explicit V3(const U* p) : x((T)p[0]), y((T)p[1]), z((T)p[2]) {}
  // values first, then the selects: `i == 0 ? x : y` on members is a select between two ADDRESSES followed by one load (and what clang does not
  // write that way, InstCombine turns into it: phi(load p, load q) -> load(phi(p, q))), which keeps the object in scratch memory behind a
  // run-time offset: collision_fn and newton_direction_fn stored whole rotation matrices to scratch to read three of their entries back
  UR5_FN T operator[](int i) const { const T a = x, b = y, c = z; return i == 0 ? a : (i == 1 ? b : c); }
  UR5_FN void set(int i, T v) { if (i == 0) x = v; else if (i == 1) y = v; else z = v; }
  template <class U> UR5_FN void store(U* p) const { p[0] = x; p[1] = y; p[2] = z; }
};
template <class T> UR5_FN V3<T> operator+(V3<T> a, V3<T> b) { return V3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> UR5_FN V3<T> operator-(V3<T> a, V3<T> b) { return V3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> UR5_FN V3<T> operator-(V3<T> a) { return V3<T>(-a.x, -a.y, -a.z); }
template <class T> UR5_FN V3<T> operator*(V3<T> a, T s) { return V3<T>(a.x * s, a.y * s, a.z * s); }
template <class T> UR5_FN T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> UR5_FN V3<T> cross(V3<T> a, V3<T> b) { return V3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
template <class T> UR5_FN T norm(V3<T> a) { return sqrt(dot(a, a)); }
template <class T> UR5_FN V3<T> normalized(V3<T> a) {
  T n = norm(a);
  return n > (T)1e-300 ? a * ((T)1 / n) : V3<T>(1, 0, 0);
}
template <class T> struct Q4 { T w, x, y, z; };
template <class T> UR5_FN Q4<T> qmul(Q4<T> a, Q4<T> b) {
  return Q4<T>{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
               a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
template <class T> UR5_FN Q4<T> qnormalize(Q4<T> q) {
  T n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < (T)1e-300) return Q4<T>{1, 0, 0, 0};
  T s = (T)1 / n;
  return Q4<T>{q.w * s, q.x * s, q.y * s, q.z * s};
}
template <class T> struct M3 {  // row-major
  T m[9];
  // selects instead of m[j]: a run-time index into a register array would push the matrix into scratch memory
  UR5_FN V3<T> col(int j) const {   // per-component scalar selects (a select between whole structs is lowered through scratch)
    const T a0 = m[0], a1 = m[1], a2 = m[2], a3 = m[3], a4 = m[4], a5 = m[5], a6 = m[6], a7 = m[7], a8 = m[8];   // see V3::operator[]
    return V3<T>(j == 0 ? a0 : (j == 1 ? a1 : a2), j == 0 ? a3 : (j == 1 ? a4 : a5), j == 0 ? a6 : (j == 1 ? a7 : a8));
  }
  UR5_FN V3<T> row(int i) const {
    const T a0 = m[0], a1 = m[1], a2 = m[2], a3 = m[3], a4 = m[4], a5 = m[5], a6 = m[6], a7 = m[7], a8 = m[8];
    return V3<T>(i == 0 ? a0 : (i == 1 ? a3 : a6), i == 0 ? a1 : (i == 1 ? a4 : a7), i == 0 ? a2 : (i == 1 ? a5 : a8));
  }
  // explicit element lists: in this very large kernel a counted loop over m[] is not always unrolled, and a run-time index
  // would move the whole matrix to scratch memory
  template <class U> UR5_FN void load(const U* p) {
    m[0] = (T)p[0]; m[1] = (T)p[1]; m[2] = (T)p[2]; m[3] = (T)p[3]; m[4] = (T)p[4]; m[5] = (T)p[5]; m[6] = (T)p[6]; m[7] = (T)p[7]; m[8] = (T)p[8];
  }
  template <class U> UR5_FN void store(U* p) const {
    p[0] = m[0]; p[1] = m[1]; p[2] = m[2]; p[3] = m[3]; p[4] = m[4]; p[5] = m[5]; p[6] = m[6]; p[7] = m[7]; p[8] = m[8];
  }
};
template <class T> UR5_FN M3<T> qmat(Q4<T> q) {
  T w = q.w, x = q.x, y = q.y, z = q.z;
  M3<T> r;
  r.m[0] = w * w + x * x - y * y - z * z; r.m[1] = 2 * (x * y - w * z); r.m[2] = 2 * (x * z + w * y);
  r.m[3] = 2 * (x * y + w * z); r.m[4] = w * w - x * x + y * y - z * z; r.m[5] = 2 * (y * z - w * x);
  r.m[6] = 2 * (x * z - w * y); r.m[7] = 2 * (y * z + w * x); r.m[8] = w * w - x * x - y * y + z * z;
  return r;
}
template <class T> UR5_FN V3<T> mul(const M3<T>& a, V3<T> v) { return V3<T>(dot(a.row(0), v), dot(a.row(1), v), dot(a.row(2), v)); }
template <class T> UR5_FN V3<T> mulT(const M3<T>& a, V3<T> v) { return V3<T>(dot(a.col(0), v), dot(a.col(1), v), dot(a.col(2), v)); }
template <class T> UR5_FN M3<T> matmul(const M3<T>& a, const M3<T>& b) {
  M3<T> r;
#define UR5_MM(i, j) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j]
  UR5_MM(0, 0); UR5_MM(0, 1); UR5_MM(0, 2); UR5_MM(1, 0); UR5_MM(1, 1); UR5_MM(1, 2); UR5_MM(2, 0); UR5_MM(2, 1); UR5_MM(2, 2);
#undef UR5_MM
  return r;
}
template <class T> UR5_FN T clampv(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
template <class T> UR5_FN T maxv(T a, T b) { return a > b ? a : b; }
template <class T> UR5_FN T minv(T a, T b) { return a < b ? a : b; }

// index of (i, j), i >= j, in a packed symmetric 6x6 (21 entries, row-major lower)
UR5_FN int sym6(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

#if (defined(UR5_MANY) || defined(UR5_STRICT_SMALL)) && !defined(UR5_EMUL) && !defined(UR5_SIMT)
#pragma clang fp contract(fast)
#endif
// ---------------------------------------------------------------------------------------------- LDS image of one scene
template <class real, int NV_> struct Lds {
  static constexpr int NV = NV_;
  static constexpr int NBODY = UR5_MAXRD + (NV_ - UR5_MAXRD) / 6;   // cbodies of this instantiation
  static constexpr int NSLOT = UR5_MAXRG + (NV_ - UR5_MAXRD) / 6;   // bodies that can carry contacts: robot weld groups with collision geoms + objects
  static constexpr int LD = NV_ + 1;                 // padded leading dimension of H (odd multiple of the bank width)
  real rec[UR5_REC_STRIDE];                          // persistent state, same layout as the HBM record
  // kinematics that the Newton phase still needs
  real bpos[NBODY][3], bmat[NBODY][9], cdof[UR5_MAXRD][6];
  real Mr[UR5_MAXRD][UR5_MAXRD + 1];
#ifdef UR5_EMUL
  real Lr[UR5_MAXRD][UR5_MAXRD + 1], Ld[UR5_MAXRD][UR5_MAXRD + 1];   // GPU build: the two factors live in registers (Fact)
#endif
  real Mobj[6 * UR5_MAXOBJ];
  static constexpr int HSIZE = NV_ * (NV_ + 1) / 2;  // packed lower triangle (GPU and lane emulation alike)
  // Six-object image (NV = 44): the packed Hessian (7 920 B) is 3.5 KB longer than the kinematic temporaries it shares its LDS with. That tail holds
  // what is dead while a Hessian is being assembled / factored / solved with: the body twists (images()), the search direction's images (cde, sr_jv:
  // and M search: written after the solve, dead once the step along it is taken) and the aref offsets (consumed by the warm start) -- 25 496 -> 22 744 B =
  // SEVEN scenes per CU instead of six (LDS is handed out in 1 280 B granules: 18 of the CU's 128, profiles/r04_z_lds_residency.log). The four-box image is bound by the temporaries, not by H, and keeps its layout.
#ifndef UR5_MANY
  static constexpr bool TAIL = NV_ > 32;
#else
  static constexpr bool TAIL = false;
#endif
  static constexpr int KIN_DOUBLES = UR5_MAXRD * (3 + 3 + 6 + 10 + 6 + 6) + NBODY * 6 + UR5_MAXDG * 12;   // the temporaries of the union below
  static constexpr int TAIL_TW = KIN_DOUBLES, TAIL_CDE = TAIL_TW + 6 * NSLOT, TAIL_CEOFF = TAIL_CDE + NB * UR5_MAXCON, TAIL_SRJV = TAIL_CEOFF + NB * UR5_MAXCON,
                       TAIL_MV = TAIL_SRJV + UR5_MAXSR;
  static_assert(!TAIL || TAIL_MV + NV_ <= HSIZE, "the aliased arrays fit behind the temporaries");
#define UR5_HIDX(i, j) ((i) * ((i) + 1) / 2 + (j))
  // The Hessian staging area shares its LDS with everything that is dead once the constraint rows exist: per-step
  // kinematic temporaries, body velocities and the moving geoms' poses are all recomputed by the next step.
  union {
    struct {
      real anchor[UR5_MAXRD][3], axis[UR5_MAXRD][3], cdd[UR5_MAXRD][6];
      real cinert[UR5_MAXRD][10], buf[UR5_MAXRD][6], cfrc[UR5_MAXRD][6];
      real cvel[NBODY][6];                           // body twist velocity [rot; lin] about the body's reference point
      real dgpos[UR5_MAXDG][3], dgmat[UR5_MAXDG][9];
    };
#ifndef UR5_MANY
    real H[HSIZE];
#else
    real tw[NSLOT][6];                               // body twists of a dof vector: built and consumed inside images() / the Newton warm start
    real stw[2 * UR5_MAXCON][6];                     // staged wrench terms of the contact sides: built and consumed inside contact_gather(); the factorisation's panel rows
    // Round 5: the envelope of the Newton Hessian / its factor is back in LDS -- in a POOL of arrays that are all dead between the gradient of a Newton
    // iteration and the images of its search direction: the temporaries / twists / staged terms above, then the aref offsets (ceoff: consumed by the warm start
    // before the first iteration) and the search direction's images (cde: written after the solve, dead once the step along the direction is taken).
    // 2 384 doubles = 19 KB, no byte added to the image (two scenes per CU stay); every envelope of the sampled settle / grasp trajectories fits (max 1 776,
    // tools/ + DESIGN.md), larger ones take the global-scratch path as before. The factor does not survive into the next iteration (images() and the gather
    // write here), so an iteration whose active set did not change refactors instead of reusing it: same Hessian bits, same factor bits.
    struct { real pool_kin_[KIN_DOUBLES]; real ceoff[UR5_MAXCON][NB], cde[UR5_MAXCON][NB]; };
    double henv[KIN_DOUBLES + 2 * UR5_MAXCON * NB];
#endif
  };
#ifdef UR5_MANY
  static constexpr int HENV_DOUBLES = KIN_DOUBLES + 2 * UR5_MAXCON * NB;
  static_assert(sizeof(real) == sizeof(double) && KIN_DOUBLES >= 2 * UR5_MAXCON * 6 && KIN_DOUBLES >= NSLOT * 6, "the pool's head holds the union's other members");
#endif
#ifdef UR5_MANY
  // envelope (skyline) storage of the Newton Hessian in global memory, dofs permuted: objects sorted along x, robot last
  double* hess;
  unsigned char env_first[NV_];                      // first stored column of a row (a multiple of 6 below 6 * 41),
  unsigned short env_ptr[NV_ + 1];                   // offset of the row (the full lower triangle is 31 k doubles)
  short obj_rank[UR5_MAXOBJ], obj_at[UR5_MAXOBJ];   // sorted position of an object and back
  int island[UR5_MAXOBJ + 1];                        // island label of object k / of the robot (index nobj)
  int blk_first[UR5_MAXOBJ + 1], blk_ptr[UR5_MAXOBJ + 1];      // per block (sorted position; robot = block nobj): first coupled block (atomic min), envelope offset of its first row
  short blk_last[UR5_MAXOBJ + 1], lv[UR5_MAXOBJ + 1], reach_cnt[UR5_MAXOBJ + 1];   // last block reaching it, level of its panel (-1: none), blocks reaching it
  int env_inlds, dc_inlds, nseq, act_changed, nskip; // env_inlds: this step's envelope fits the LDS pool (henv in the union above); dc_inlds: and so does the block cache
  static constexpr int REACH_CAP = (UR5_MAXOBJ + 1) * UR5_MAXOBJ / 4;                // half the worst case: 410 off-diagonal blocks = an envelope of > 15 k doubles
  short reach_ptr[UR5_MAXOBJ + 2], reach_list[REACH_CAP];                            // (settled piles: ~45 blocks, 2.5 k doubles); beyond it the scene is flagged
                                                                                     // per panel: the blocks below it whose rows reach it
  unsigned short cact[UR5_MAXCON];                   // active-row signature of every contact at the previous Newton iteration
  int sr_act[UR5_MAXSR];
  short seq[UR5_MAXOBJ + 1];                         // blocks that take part in the sequential factorisation (the others are uncoupled)
  short lvl_ptr[UR5_MAXOBJ + 3], lvl_list[UR5_MAXOBJ + 1];   // the same panels grouped by level: panels of one level belong to different
  int nlvl;                                          // envelope groups (islands) and are processed together, one wavefront each
  real red[3 * (UR5_NT / 64)];                       // cross-wave reductions
  int redi[UR5_NT / 64];
  // Fixed-order accumulation (round 4): four wavefronts share a scene, so LDS float atomics would land in an order that changes from run to
  // run. Instead every body slot owns the list of its contact sides (2 c + side, in contact order): the contact lanes stage their terms, the slot's
  // lanes sum them along the list; every Hessian coupling block is owned by one wavefront, which adds its contacts' terms in contact order. A scene's
  // results then do not depend on how its wavefronts are scheduled (MujocoController.py:379 is one deterministic thread).
  short csl[UR5_MAXCON][2];                          // accumulator slot of a contact's two bodies (-1: static side); flat index = side id 2 c + side
  short side_list[2 * UR5_MAXCON], slot_ptr[NSLOT + 1];
  short side_pos[2 * UR5_MAXCON];                    // a side's position in side_list (sides of static bodies have none): where its staged terms go
  unsigned char slot_cnt[UR5_NT / 64][NSLOT + 1];    // sides of a slot held by the lanes of each wavefront (list construction)
  unsigned long long wrec[UR5_MAXCON];               // the coupled contacts grouped by the wavefront that owns their block pair, contact order inside a group:
  short wptr[UR5_NT / 64 + 1];                       // one packed record each (contact | body A << 8 | body B << 16 | block pair << 24 | A-owns-the-row-block << 40)

#endif
  // dynamics vectors (dof space)
  real fs[NV_], as[NV_], x[NV_], Ma[NV_], search[NV_], Mv[TAIL ? 1 : NV_], tmpv[NV_ + 4];
#ifndef UR5_MANY
  real grad[NV_];                                    // (the many-object kernel keeps the gradient in `search` until the solve overwrites it)
#endif
  // contacts
  int ncon, nsr, ncand, ncouple;
  unsigned cplmask;              // bit k: object k takes part in a contact between two movable bodies; bit 31: one of them has a robot side
  unsigned long long bodymask;   // cbodies that carry at least one contact
#ifndef UR5_MANY
  int cA[UR5_MAXCON], cB[UR5_MAXCON], cdim[UR5_MAXCON];
#else
  int cA[UR5_MAXCON];                                // (also the sort key of sort_contacts() while the contacts are collected)
  signed char cB[UR5_MAXCON];                        // narrow types: every byte counts towards two scenes per CU
  unsigned char cdim[UR5_MAXCON];
#endif
  short cg1[UR5_MAXCON], cg2[UR5_MAXCON];
  short cand[UR5_MAXCAND];
#ifndef UR5_MANY
  // broad-phase cache (wavefront-per-scene engine): `sup` is a SUPERSET of the pairs that can pass cull() while no moving geom has travelled more than
  // UR5_SUP_DELTA since the list was built; `moved` bounds each moving geom's travel since then (sum of |v| h over the steps). nsup < 0: no valid list.
  short sup[UR5_MAXCAND];
  float moved[UR5_MAXDG];
  int nsup;
#endif
#ifndef UR5_MANY
  int couple[UR5_MAXCON];
#else
  short couple[UR5_MAXCON];                          // contacts between two movable bodies (and, during the narrow phase, the queue of hull pairs)
#endif
  real cpos[UR5_MAXCON][3], cframe[UR5_MAXCON][6], cdist[UR5_MAXCON], cfri[UR5_MAXCON][NB > 4 ? 3 : 2];   // cframe: normal, tangent 1 (tangent 2 = n x t1)
  real cD[UR5_MAXCON];
#ifndef UR5_MANY
  real ceoff[TAIL ? 1 : UR5_MAXCON][NB], ce[UR5_MAXCON][NB], cde[TAIL ? 1 : UR5_MAXCON][NB];   // ceoff: -aref in base space
#else
  real ce[UR5_MAXCON][NB];                           // (ceoff and cde live in the Hessian pool of the union above)
#endif
#ifdef UR5_EMUL
  real cfn[UR5_MAXCON];                              // normal force, test introspection only
#endif
  // special rows (joint equality, joint limits): jar = c1 x[d1] + c2 x[d2] - aref
  int sr_d1[UR5_MAXSR], sr_d2[UR5_MAXSR], sr_uni[UR5_MAXSR];
  real sr_c1[UR5_MAXSR], sr_c2[UR5_MAXSR], sr_D[UR5_MAXSR], sr_aref[UR5_MAXSR], sr_jar[UR5_MAXSR], sr_jv[TAIL ? 1 : UR5_MAXSR];
  // body accumulators (twist space)
#ifndef UR5_MANY
  real tw[TAIL ? 1 : NSLOT][6];                      // (four boxes: could share the staging union as in the many-object image: -384 B, no residency step gained by it alone)
#endif
  real WB[NSLOT][6], G[NSLOT][21];   // indexed by slot_of(body)
#if defined(UR5_PROFILE) && !defined(UR5_EMUL)
  double prof[PF_COUNT];
#endif
#ifdef UR5_LDS_PAD
  char lds_pad[UR5_LDS_PAD];                         // experiment: a larger image = fewer scenes per CU (how much throughput does one resident scene buy?)
#endif
  int status, solver_iters, ncon_max, badstate;
  real pid_dt;
  int contacts_enabled, last_steps, total_steps;
  // the arrays that live in the Hessian's tail in the six-object image (TAIL), at their own address otherwise
#ifndef UR5_MANY
  UR5_FN real (*tw_())[6] { if constexpr (TAIL) return reinterpret_cast<real (*)[6]>(H + TAIL_TW); else return tw; }
  UR5_FN real (*cde_())[NB] { if constexpr (TAIL) return reinterpret_cast<real (*)[NB]>(H + TAIL_CDE); else return cde; }
  UR5_FN real (*ceoff_())[NB] { if constexpr (TAIL) return reinterpret_cast<real (*)[NB]>(H + TAIL_CEOFF); else return ceoff; }
  UR5_FN real* sr_jv_() { if constexpr (TAIL) return H + TAIL_SRJV; else return sr_jv; }
  UR5_FN real* Mv_() { if constexpr (TAIL) return H + TAIL_MV; else return Mv; }
#else
  UR5_FN real (*tw_())[6] { return tw; }
  UR5_FN real (*cde_())[NB] { return cde; }
  UR5_FN real (*ceoff_())[NB] { return ceoff; }
  UR5_FN real* sr_jv_() { return sr_jv; }
  UR5_FN real* Mv_() { return Mv; }
#endif
};

// ---------------------------------------------------------------------------------------------- the engine
template <class real, int NV_, int GS_ = UR5_NT> struct Engine {
  static constexpr int GS = GS_;   // lanes that own one scene
  // The register-hungry phases (kinematics, collision, constraint rows, Newton solve / direction) are real functions in the one-scene-per-
  // wavefront kernel (256-register cap, 2 waves per SIMD): each then gets its own register allocation instead of sharing one 50 k-instruction
  // body (+30 % measured in round 1). FLAT kernels have the whole register file and inline them.
#ifdef UR5_EMUL
  static constexpr bool FLAT = true;
#else
#ifdef UR5_MANY_SPLIT_PHASES      // experiment: the many-object kernel with its phases as real functions (own register allocation each), like the wavefront-per-scene kernel
  static constexpr bool FLAT = false;
#else
  static constexpr bool FLAT = GS_ != 64;
#endif
#endif
  typedef Lds<real, NV_> L;
  typedef V3<real> v3;
  typedef M3<real> m3;
  typedef Q4<real> q4;
#define S (*UR5_LDS_PTR(L))
#define M UR5_MODEL

  // views into the persistent record
  UR5_FN real* qpos() { return S.rec + UR5_REC_QPOS; }
  UR5_FN real* qvel() { return S.rec + UR5_REC_QVEL; }
  UR5_FN real* warm() { return S.rec + UR5_REC_WARM; }
  UR5_FN real* ctrl() { return S.rec + UR5_REC_CTRL; }
  UR5_FN real* target() { return S.rec + UR5_REC_TARGET; }
  UR5_FN real* pid_in() { return S.rec + UR5_REC_PIDIN; }
  UR5_FN real* pid_out() { return S.rec + UR5_REC_PIDOUT; }
  UR5_FN real* kp() { return S.rec + UR5_REC_KP; }
  UR5_FN int nb() const { return M.nrd + M.nobj; }
  UR5_FN int nslot() const { return M.nrg + M.nobj; }
  UR5_FN int slot_of(int b) const { return b < M.nrd ? M.rd_gslot[b] : M.nrg + (b - M.nrd); }
  UR5_FN int body_of_slot(int sl) const { return sl < M.nrg ? M.rg_body[sl] : M.nrd + (sl - M.nrg); }
#if !defined(UR5_EMUL)
  UR5_FN real group_sum(real v) {
    if constexpr (GS == 64) return ur5_wave_sum(v); else if constexpr (GS == 32) return ur5_half_sum(v); else return block_sum(v);
  }
  UR5_FN real group_max(real v) {
    if constexpr (GS == 64) return ur5_wave_max<64>(v); else if constexpr (GS == 32) return ur5_wave_max<32>(v); else return block_max(v);
  }
  // sums / maxima over all wavefronts of the scene: wave shuffle, then the per-wave results through LDS in wave order
  UR5_FN real block_sum(real v) {
    v = ur5_wave_sum(v);
    SYNC();
    if ((UR5_LANE & 63) == 0) S.red[UR5_LANE >> 6] = v;
    SYNC();
    real t = 0;
#pragma unroll
    for (int w = 0; w < UR5_NT / 64; w++) t += S.red[w];
    return t;
  }
  UR5_FN void block_sum3(real& a, real& b, real& c) {
    a = ur5_wave_sum(a); b = ur5_wave_sum(b); c = ur5_wave_sum(c);
    SYNC();
    if ((UR5_LANE & 63) == 0) { const int w = UR5_LANE >> 6; S.red[3 * w] = a; S.red[3 * w + 1] = b; S.red[3 * w + 2] = c; }
    SYNC();
    real ta = 0, tb = 0, tc = 0;
#pragma unroll
    for (int w = 0; w < UR5_NT / 64; w++) { ta += S.red[3 * w]; tb += S.red[3 * w + 1]; tc += S.red[3 * w + 2]; }
    a = ta; b = tb; c = tc;
  }
  UR5_FN real block_max(real v) {
    v = ur5_wave_max<64>(v);
    SYNC();
    if ((UR5_LANE & 63) == 0) S.red[UR5_LANE >> 6] = v;
    SYNC();
    real t = S.red[0];
#pragma unroll
    for (int w = 1; w < UR5_NT / 64; w++) t = S.red[w] > t ? S.red[w] : t;
    return t;
  }
#endif

  UR5_FN void load(const double* rec, real dt, int con) {
    PAR(i, UR5_REC_STRIDE) S.rec[i] = (real)rec[i];
    if (UR5_LANE == 0) { S.pid_dt = dt; S.contacts_enabled = con; S.last_steps = 0; S.total_steps = 0; }
    if (UR5_LANE == 0) { S.status = 0; S.solver_iters = 0; S.ncon_max = 0; S.ncon = 0; S.nsr = 0; S.badstate = 0; }
#ifndef UR5_MANY
    if (UR5_LANE == 0) S.nsup = -1;
#endif
#ifdef UR5_MANY
    if (UR5_LANE == 0) { S.nskip = 0; S.act_changed = 1; }
#endif
#if defined(UR5_PROFILE) && !defined(UR5_EMUL)
    if (UR5_LANE == 0) { for (int i = 0; i < PF_COUNT; i++) S.prof[i] = 0; S.prof[PF_CORECLK] = (double)wall_clock64(); }
#endif
    SYNC();
    S.status = (int)S.rec[UR5_REC_MISC + 3];
  }
#ifdef UR5_MANY
  UR5_FN void set_hess(double* h) { if (UR5_LANE == 0) S.hess = h; SYNC(); }
#endif
  UR5_FN void save(double* rec) {
    SYNC();
    {   // a non-finite state is flagged, never silently written back as if it were a result
      bool bad = false;
      PAR(i, M.nq + M.nv) { real v = S.rec[i < M.nq ? UR5_REC_QPOS + i : UR5_REC_QVEL + (i - M.nq)]; if (!(v - v == 0)) bad = true; }
      if (bad) S.status |= UR5_ST_NAN;   // benign race: every writer ORs the same bit into a word nobody else changes here
    }
    SYNC();
    if (UR5_LANE == 0) {
      S.rec[UR5_REC_MISC + 0] += (real)S.total_steps;
      S.rec[UR5_REC_MISC + 1] = (real)S.last_steps;
      S.rec[UR5_REC_MISC + 3] = (real)S.status;
      S.rec[UR5_REC_MISC + 4] += (real)S.solver_iters;
      S.rec[UR5_REC_MISC + 5] = maxv(S.rec[UR5_REC_MISC + 5], (real)S.ncon_max);
#ifdef UR5_MANY
      S.rec[UR5_REC_MISC + 6] += (real)S.nskip;   // Newton iterations that reused the previous factor
#endif
    }
    SYNC();
    PAR(i, UR5_REC_STRIDE) rec[i] = (double)S.rec[i];
  }

  // ------------------------------------------------------------------ kinematics (mj_kinematics + mj_comPos [3P])
  UR5_CALL void kinematics_fn() { kinematics_body(); }
  UR5_FN void kinematics() { if constexpr (FLAT) kinematics_body(); else kinematics_fn(); }
  UR5_PHASE_D void kinematics_body() { UR5_STRICT;
    // ping-pong buffers of the pointer-jumping pass: ce / cde (contact images, dead until this step's constraint rows are
    // built) hold the frames, cand (broad-phase list, rebuilt later) the ancestor links
    static_assert(UR5_MAXCON * NB >= 12 * UR5_MAXRD && UR5_MAXCAND * sizeof(short) >= 2 * UR5_MAXRD * sizeof(int), "scratch aliasing");
    real* const kbuf[2] = {&S.ce[0][0], &S.cde_()[0][0]};
    int* const kanc = reinterpret_cast<int*>(S.cand);
#define UR5_KR(bf, d) (kbuf[bf] + 9 * (d))
#define UR5_KP(bf, d) (kbuf[bf] + 9 * UR5_MAXRD + 3 * (d))
#define UR5_KA(bf, d) kanc[(bf) * UR5_MAXRD + (d)]
    // robot tree: local transform of every weld group (parent frame -> own frame, joint rotation included) in parallel, then
    // three rounds of pointer jumping compose them to world frames (tree depth <= 8) instead of walking the chain serially
    PAR(d, M.nrd) {
      real a = qpos()[d] - (real)M.rd_qpos0[d];
      real sn = sin(a), cs = cos(a), oc = (real)1 - cs;
      v3 u(M.rd_jaxis[d]);
      m3 Rq;   // Rodrigues
      Rq.m[0] = cs + oc * u.x * u.x; Rq.m[1] = oc * u.x * u.y - sn * u.z; Rq.m[2] = oc * u.x * u.z + sn * u.y;
      Rq.m[3] = oc * u.y * u.x + sn * u.z; Rq.m[4] = cs + oc * u.y * u.y; Rq.m[5] = oc * u.y * u.z - sn * u.x;
      Rq.m[6] = oc * u.z * u.x - sn * u.y; Rq.m[7] = oc * u.z * u.y + sn * u.x; Rq.m[8] = cs + oc * u.z * u.z;
      m3 R0; R0.load(M.rd_mat[d]);
      v3 jp(M.rd_jpos[d]);
      matmul(R0, Rq).store(UR5_KR(0, d));
      (v3(M.rd_pos[d]) + mul(R0, jp - mul(Rq, jp))).store(UR5_KP(0, d));
      UR5_KA(0, d) = M.rd_parent[d];
    }
    PAR(k, M.nobj) {
      int b = M.nrd + k, qa = M.nrd + 7 * k;
      v3 p(qpos()[qa], qpos()[qa + 1], qpos()[qa + 2]);
      if (M.obj_kind[k] == 0) p = p + v3(M.obj_pos0[k]);
      q4 q = qnormalize(q4{qpos()[qa + 3], qpos()[qa + 4], qpos()[qa + 5], qpos()[qa + 6]});
      p.store(S.bpos[b]);
      qmat(q).store(S.bmat[b]);
    }
    SYNC();
    for (int round = 0; round < 3; round++) {
      const int src = round & 1, dst = src ^ 1;
      PAR(d, M.nrd) {
        int a = UR5_KA(src, d);
        m3 R; R.load(UR5_KR(src, d));
        v3 p(UR5_KP(src, d));
        if (a >= 0) {
          m3 Ra; Ra.load(UR5_KR(src, a));
          p = v3(UR5_KP(src, a)) + mul(Ra, p);
          R = matmul(Ra, R);
          a = UR5_KA(src, a);
        }
        R.store(UR5_KR(dst, d)); p.store(UR5_KP(dst, d)); UR5_KA(dst, d) = a;
      }
      SYNC();
    }
    PAR(d, M.nrd) {   // after three rounds buffer 1 holds the world frames
      m3 R; R.load(UR5_KR(1, d));
      v3 p(UR5_KP(1, d));
      R.store(S.bmat[d]); p.store(S.bpos[d]);
      (p + mul(R, v3(M.rd_jpos[d]))).store(S.anchor[d]);
      mul(R, v3(M.rd_jaxis[d])).store(S.axis[d]);
    }
    SYNC();
#undef UR5_KR
#undef UR5_KP
#undef UR5_KA
    v3 o(M.ref_point);
    PAR(d, M.nrd) {
      v3 ax(S.axis[d]);
      v3 lin = cross(ax, o - v3(S.anchor[d]));
      ax.store(S.cdof[d]); lin.store(S.cdof[d] + 3);
      // spatial inertia of the weld group about o, world axes: I(6) h(3) m
      m3 R; R.load(S.bmat[d]);
      const double* bi = M.rd_inertia[d];
      m3 Ib;
      Ib.m[0] = (real)bi[0]; Ib.m[4] = (real)bi[1]; Ib.m[8] = (real)bi[2];
      Ib.m[1] = Ib.m[3] = (real)bi[3]; Ib.m[2] = Ib.m[6] = (real)bi[4]; Ib.m[5] = Ib.m[7] = (real)bi[5];
      m3 Rt;
      Rt.m[0] = R.m[0]; Rt.m[1] = R.m[3]; Rt.m[2] = R.m[6]; Rt.m[3] = R.m[1]; Rt.m[4] = R.m[4]; Rt.m[5] = R.m[7]; Rt.m[6] = R.m[2]; Rt.m[7] = R.m[5]; Rt.m[8] = R.m[8];
      m3 Iw = matmul(matmul(R, Ib), Rt);
      real m = (real)M.rd_mass[d];
      v3 c = v3(S.bpos[d]) + mul(R, v3(M.rd_ipos[d])) - o;
      real* ci = S.cinert[d];
      ci[0] = Iw.m[0] + m * (c.y * c.y + c.z * c.z); ci[1] = Iw.m[4] + m * (c.x * c.x + c.z * c.z); ci[2] = Iw.m[8] + m * (c.x * c.x + c.y * c.y);
      ci[3] = Iw.m[1] - m * c.x * c.y; ci[4] = Iw.m[2] - m * c.x * c.z; ci[5] = Iw.m[5] - m * c.y * c.z;
      ci[6] = m * c.x; ci[7] = m * c.y; ci[8] = m * c.z; ci[9] = m;
    }
    PAR(i, M.ndg) {
      int g = M.dg_geom[i], ow = M.g_owner[g];
      if (M.g_kind[g] == UR5_KIND_ROBOT) {
        m3 R; R.load(S.bmat[ow]);
        (v3(S.bpos[ow]) + mul(R, v3(M.g_pos[g]))).store(S.dgpos[i]);
        m3 G; G.load(M.g_mat[g]);
        matmul(R, G).store(S.dgmat[i]);
      } else {
        int b = M.nrd + ow;
        for (int k = 0; k < 3; k++) S.dgpos[i][k] = S.bpos[b][k];
        for (int k = 0; k < 9; k++) S.dgmat[i][k] = S.bmat[b][k];
      }
    }
    SYNC();
  }

  // inertia (10 numbers: I6 h3 m) times spatial vector [rot; lin] -> [rot; lin]
  UR5_FN static void mul_inert(const real* ci, const real* v, real* out) {
    v3 w(v), l(v + 3), h(ci + 6);
    v3 r(ci[0] * w.x + ci[3] * w.y + ci[4] * w.z, ci[3] * w.x + ci[1] * w.y + ci[5] * w.z, ci[4] * w.x + ci[5] * w.y + ci[2] * w.z);
    r = r + cross(h, l);
    v3 f = l * ci[9] - cross(h, w);
    r.store(out); f.store(out + 3);
  }

  // ------------------------------------------------------------------ CRBA (robot block) + object diagonals + factors
  // factors of Mr (lanes 0-7) and Mr + h B (lanes 8-15) kept in registers for the whole step (GPU build); the lane-emulation
  // build keeps them in LDS (S.Lr / S.Ld) and leaves this empty
  struct Fact {
#ifndef UR5_EMUL
    real r[UR5_MAXRD];
    real inv;
    int base, loc, size;
#endif
  };
  UR5_PHASE_E void crb_and_factor(Fact& fr) {
    PAR(d, M.nrd) {
      real crb[10];
      for (int i = 0; i < 10; i++) crb[i] = 0;
      for (int b = 0; b < M.nrd; b++) if (M.rd_desc[d] >> b & 1u) for (int i = 0; i < 10; i++) crb[i] += S.cinert[b][i];
      mul_inert(crb, S.cdof[d], S.buf[d]);
    }
    PAR(i, 6 * M.nobj) {
      int k = i / 6, j = i % 6;
      S.Mobj[i] = j < 3 ? (real)(M.obj_mass[k] + M.obj_arm[k][0]) : (real)(M.obj_inertia[k][j - 3] + M.obj_arm[k][1]);
    }
    SYNC();
    PAR(idx, M.nrd * M.nrd) {
      int d = idx / M.nrd, e = idx % M.nrd;
      if (e <= d) {
        real v = 0;
        if (M.rd_anc[d] >> e & 1u) for (int i = 0; i < 6; i++) v += S.cdof[e][i] * S.buf[d][i];
        if (e == d) v += (real)M.rd_armature[d];
        S.Mr[d][e] = v; S.Mr[e][d] = v;
      }
    }
    SYNC();
    real h = (real)M.timestep;
#ifdef UR5_EMUL
    PAR(idx, M.nrd * M.nrd) {
      int d = idx / M.nrd, e = idx % M.nrd;
      S.Lr[d][e] = S.Mr[d][e];
      S.Ld[d][e] = S.Mr[d][e] + (d == e ? h * (real)M.rd_damping[d] : (real)0);
    }
    SYNC();
    cholesky(&S.Lr[0][0], M.nrd, UR5_MAXRD + 1);
    cholesky(&S.Ld[0][0], M.nrd, UR5_MAXRD + 1);
#else
    {   // lanes 0-7: rows of Mr, lanes 8-15: rows of Mr + h B; both factored at once (block-parallel register Cholesky)
      const int lane = UR5_LANE, nrd = M.nrd;
      Blk b;
      b.size = lane < 2 * UR5_MAXRD ? nrd : 0;
      b.base = lane < UR5_MAXRD ? 0 : UR5_MAXRD;
      b.loc = lane - b.base;
      if (b.loc >= nrd) b.size = 0;
      if (b.size == 0) { b.base = lane; b.loc = 0; }
      real r[UR5_MAXRD];
      const real hb = (lane >= UR5_MAXRD && b.size > 0) ? h * (real)M.rd_damping[b.loc] : (real)0;
#pragma unroll
      for (int j = 0; j < UR5_MAXRD; j++) r[j] = (b.size > 0 && j <= b.loc) ? S.Mr[b.loc][j] + (j == b.loc ? hb : (real)0) : (real)0;
      fr.inv = blk_cholesky(r, b);
#pragma unroll
      for (int j = 0; j < UR5_MAXRD; j++) fr.r[j] = r[j];
      fr.base = b.base; fr.loc = b.loc; fr.size = b.size;
    }
#endif
  }

#ifndef UR5_EMUL
  // ---- block-parallel register linear algebra: every lane owns one row of one diagonal block (<= 8 columns, held in
  // registers r[0..7] by block-local column). All blocks are factored / solved simultaneously; the pivot row of each block
  // is fetched with a lane shuffle (ds_bpermute), so 8 column steps serve the robot block and every object block at once.
  struct Blk { int base, loc, size; };   // first lane of my block (index inside the scene's lane group), my row inside it, its order (0 = lane idle)
  static __device__ __forceinline__ real shfl_d(real v, int src) { return (real)__shfl((double)v, src, 64); }
  // in-place Cholesky (lower, row-wise); returns 1 / (own diagonal entry)
  static __device__ __forceinline__ real blk_cholesky(real (&r)[UR5_MAXRD], const Blk& b) {
    real myinv = 1;
    const int gb = UR5_GBASE;
#pragma unroll
    for (int j = 0; j < UR5_MAXRD; j++) {
      const bool act = j < b.size;
      const int src = act ? gb + b.base + j : (int)threadIdx.x;
      real sacc = r[j];
#pragma unroll
      for (int k = 0; k < j; k++) sacc -= r[k] * shfl_d(r[k], src);
      real djj = shfl_d(sacc, src);
      djj = djj < (real)1e-15 ? (real)1e-15 : djj;
      real inv = rsqrt(djj);
      inv = inv * ((real)1.5 - (real)0.5 * djj * inv * inv);   // one Newton step: rsqrt -> full fp64 accuracy
      if (act) {
        r[j] = b.loc == j ? djj * inv : (b.loc > j ? sacc * inv : (real)0);
        if (b.loc == j) myinv = inv;
      }
    }
    return myinv;
  }
  // x <- (L L^T)^-1 x for every block; lt = LDS scratch of 64 x 8 reals used to transpose the factors
  static __device__ __forceinline__ real blk_solve(const real (&r)[UR5_MAXRD], real myinv, const Blk& b, real x, real* lt, int nl = GS) {
    const int lane = UR5_LANE, gb = UR5_GBASE, self = (int)threadIdx.x;
#pragma unroll
    for (int j = 0; j < UR5_MAXRD; j++) {
      const bool act = j < b.size;
      real yj = shfl_d(x * myinv, act ? gb + b.base + j : self);
      if (act) x = b.loc == j ? yj : (b.loc > j ? x - r[j] * yj : x);
    }
    if (lane < nl) {
#pragma unroll
      for (int j = 0; j < UR5_MAXRD; j++) lt[lane * UR5_MAXRD + j] = r[j];
    }
    SYNC();
    real t[UR5_MAXRD];
#pragma unroll
    for (int k = 0; k < UR5_MAXRD; k++) t[k] = (k < b.size && k > b.loc) ? lt[(b.base + k) * UR5_MAXRD + b.loc] : (real)0;
    SYNC();
#pragma unroll
    for (int k = UR5_MAXRD - 1; k >= 0; k--) {
      const bool act = k < b.size;
      real xk = shfl_d(x * myinv, act ? gb + b.base + k : self);
      if (act) x = b.loc == k ? xk : (b.loc < k ? x - t[k] * xk : x);
    }
    return x;
  }
#endif

  // in-place lower Cholesky of the n x n matrix A (leading dimension ld) -- left-looking, one column per step
  UR5_CALL void cholesky(real* A, int n, int ld) {
    for (int j = 0; j < n; j++) {
      PAR(ii, n - j) {
        int i = j + ii;
        real s = A[i * ld + j];
        for (int k = 0; k < j; k++) s -= A[i * ld + k] * A[j * ld + k];
        S.tmpv[i] = s;
      }
      SYNC();
      real d = S.tmpv[j];
      d = sqrt(d < (real)1e-15 ? (real)1e-15 : d);
      real inv = (real)1 / d;
      PAR(ii, n - j) {
        int i = j + ii;
        A[i * ld + j] = (i == j) ? d : S.tmpv[i] * inv;
      }
      SYNC();
    }
  }
  // b <- (L L^T)^-1 b
  UR5_CALL void chol_solve(const real* A, int n, int ld, real* b) {
    for (int k = 0; k < n; k++) {
      real yk = b[k] / A[k * ld + k];
      SYNC();
      PAR(ii, n - k) {
        int i = k + ii;
        if (i == k) b[k] = yk; else b[i] -= A[i * ld + k] * yk;
      }
      SYNC();
    }
    for (int k = n - 1; k >= 0; k--) {
      real xk = b[k] / A[k * ld + k];
      SYNC();
      PAR(i, k + 1) {
        if (i == k) b[k] = xk; else b[i] -= A[k * ld + i] * xk;
      }
      SYNC();
    }
  }

  // ------------------------------------------------------------------ velocity stage: body twists, bias, passive (mj_comVel + mj_rne)
  UR5_PHASE_F void velocity_stage(const Fact& fr) {
    PAR(b, M.nrd) {
      real v[6] = {0, 0, 0, 0, 0, 0};
      for (int e = 0; e < M.nrd; e++) if (M.rd_anc[b] >> e & 1u) { real q = qvel()[e]; for (int i = 0; i < 6; i++) v[i] += S.cdof[e][i] * q; }
      for (int i = 0; i < 6; i++) S.cvel[b][i] = v[i];
      // cdof_dot = crossMotion(cvel, cdof): own joint's contribution to cvel is parallel to cdof and drops out
      v3 w(v), l(v + 3), cr(S.cdof[b]), cl(S.cdof[b] + 3);
      cross(w, cr).store(S.cdd[b]);
      (cross(w, cl) + cross(l, cr)).store(S.cdd[b] + 3);
    }
    PAR(k, M.nobj) {
      int b = M.nrd + k, va = M.nrd + 6 * k;
      m3 R; R.load(S.bmat[b]);
      mul(R, v3(qvel()[va + 3], qvel()[va + 4], qvel()[va + 5])).store(S.cvel[b]);
      v3(qvel()[va], qvel()[va + 1], qvel()[va + 2]).store(S.cvel[b] + 3);
    }
    SYNC();
    PAR(b, M.nrd) {
      real a[6] = {0, 0, 0, -(real)M.gravity[0], -(real)M.gravity[1], -(real)M.gravity[2]};
      for (int e = 0; e < M.nrd; e++) if (M.rd_anc[b] >> e & 1u) { real q = qvel()[e]; for (int i = 0; i < 6; i++) a[i] += S.cdd[e][i] * q; }
      real f[6], mv[6];
      mul_inert(S.cinert[b], a, f);
      mul_inert(S.cinert[b], S.cvel[b], mv);
      v3 w(S.cvel[b]), l(S.cvel[b] + 3), mr(mv), ml(mv + 3);
      v3 fr = v3(f) + cross(w, mr) + cross(l, ml), fl = v3(f + 3) + cross(w, ml);
      fr.store(S.cfrc[b]); fl.store(S.cfrc[b] + 3);
    }
    SYNC();
    PAR(d, M.nrd) {
      real bias = 0;
      for (int b = 0; b < M.nrd; b++) if (M.rd_desc[d] >> b & 1u) for (int i = 0; i < 6; i++) bias += S.cdof[d][i] * S.cfrc[b][i];
      S.fs[d] = -(real)M.rd_damping[d] * qvel()[d] - bias;
    }
    PAR(k, M.nobj) {
      int va = M.nrd + 6 * k;
      real m = (real)M.obj_mass[k];
      for (int j = 0; j < 3; j++) S.fs[va + j] = -(real)M.obj_damp[k][0] * qvel()[va + j] + m * (real)M.gravity[j];
      v3 w(qvel()[va + 3], qvel()[va + 4], qvel()[va + 5]);
      v3 Iw(w.x * (real)M.obj_inertia[k][0], w.y * (real)M.obj_inertia[k][1], w.z * (real)M.obj_inertia[k][2]);
      v3 gy = cross(w, Iw);
      for (int j = 0; j < 3; j++) S.fs[va + 3 + j] = -(real)M.obj_damp[k][1] * w[j] - gy[j];
    }
    SYNC();
    PAR(a, M.nu) {  // mj_fwdActuation: clamp to ctrlrange, gear
      real c = clampv(ctrl()[a], (real)M.act_lo[a], (real)M.act_hi[a]);
      S.fs[M.act_dof[a]] += (real)M.act_gear[a] * c;
    }
    SYNC();
    PAR(i, M.nv) S.as[i] = i < M.nrd ? S.fs[i] : S.fs[i] / S.Mobj[i - M.nrd];
    SYNC();
#ifdef UR5_EMUL
    chol_solve(&S.Lr[0][0], M.nrd, UR5_MAXRD + 1, S.as);
#else
    {   // qacc_smooth of the robot: lanes 0-7 hold the rows of chol(Mr); S.x..S.Mv_() are free until the Newton solve (scratch)
      Blk b; b.base = fr.base; b.loc = fr.loc; b.size = fr.size;
      real rhs = UR5_LANE < M.nrd ? S.as[UR5_LANE] : (real)0;
      real xs = blk_solve(fr.r, fr.inv, b, rhs, S.x, 2 * UR5_MAXRD);
      if (UR5_LANE < M.nrd) S.as[UR5_LANE] = xs;
      SYNC();
    }
#endif
  }

  // ------------------------------------------------------------------ collision
  struct GeomPose { v3 pos; m3 mat; };
  UR5_FN GeomPose geom_pose(int g) const { UR5_STRICT;
    GeomPose r;
    int dg = M.g_dg[g];
    if (dg < 0) { r.pos = v3(M.g_pos[g]); r.mat.load(M.g_mat[g]); }
    else { r.pos = v3(S.dgpos[dg]); r.mat.load(S.dgmat[dg]); }
    return r;
  }
  UR5_FN static real dist_point_box(v3 p, const GeomPose& B, v3 s) { UR5_STRICT;
    v3 l = mulT(B.mat, p - B.pos);
    v3 d(maxv(fabs(l.x) - s.x, (real)0), maxv(fabs(l.y) - s.y, (real)0), maxv(fabs(l.z) - s.z, (real)0));
    return norm(d);
  }
  struct Shape { int type, vadr, vnum; v3 pos, size, center; m3 mat; real margin; };
  UR5_FN Shape make_shape(int g, real margin) const { UR5_STRICT;
    Shape s;
    GeomPose P = geom_pose(g);
    s.type = M.g_type[g]; s.pos = P.pos; s.mat = P.mat; s.size = v3(M.g_size[g]);
    s.vadr = M.g_vadr[g]; s.vnum = M.g_vnum[g];
    s.center = P.pos + mul(P.mat, v3(M.g_center[g]));
    s.margin = margin;
    return s;
  }
  // W = 1: the calling lane scans the hull's vertices itself. W = 8 (GPU narrow phase): the 8 lanes of an aligned sub-group work on the same
  // pair with identical arguments; lane `sl` of the sub-group takes vertices sl, sl + 8, ... and a 3-step lane exchange picks the winner --
  // larger dot product, smaller index on ties, which is exactly the vertex the serial scan (strict >) returns.
  template <int W = 1> UR5_BIG v3 support(const Shape& s, v3 dir, int sl = 0) const { UR5_STRICT;
    v3 d = mulT(s.mat, dir), l;
    if (s.type == UR5_GEOM_SPHERE) l = d * s.size.x;
    else if (s.type == UR5_GEOM_BOX) l = v3(d.x >= 0 ? s.size.x : -s.size.x, d.y >= 0 ? s.size.y : -s.size.y, d.z >= 0 ? s.size.z : -s.size.z);
    else if (s.type == UR5_GEOM_CAPSULE) l = d * s.size.x + v3(0, 0, d.z >= 0 ? s.size.y : -s.size.y);
    else if (s.type == UR5_GEOM_CYLINDER) {
      real n = sqrt(d.x * d.x + d.y * d.y);
      l = n > (real)1e-12 ? v3(d.x / n * s.size.x, d.y / n * s.size.x, 0) : v3();
      l.z = d.z >= 0 ? s.size.y : -s.size.y;
    } else if (s.type == UR5_GEOM_MESH) {
      real best = -1e300;
      int bi = 0;
#if defined(UR5_MPR_DPP_COORDS) && !defined(UR5_EMUL)
      // build option (not measured yet): the lane keeps the coordinates of its best vertex and the DPP exchange carries them along, so the
      // winner does not have to be fetched again with a second, dependent load after the exchange
      v3 bl;
#endif
      if constexpr (W == 1) {
        for (int i = 0; i < s.vnum; i++) {
          const double* p = M.hullvert[s.vadr + i];
          real v = (real)p[0] * d.x + (real)p[1] * d.y + (real)p[2] * d.z;
          if (v > best) { best = v; bi = i; }
        }
      } else {
        // UR5_SUP_K vertices per lane and trip, all their loads issued before the first use (the hulls live in constant memory: one L1 / L2 round
        // trip per trip instead of one per vertex)
        for (int base = 0; base < s.vnum; base += UR5_SUP_K * W) {
          real px[UR5_SUP_K], py[UR5_SUP_K], pz[UR5_SUP_K];
#pragma unroll
          for (int k = 0; k < UR5_SUP_K; k++) {
            const int i = base + sl + W * k;
            const double* p = M.hullvert[s.vadr + (i < s.vnum ? i : s.vnum - 1)];
            px[k] = (real)p[0]; py[k] = (real)p[1]; pz[k] = (real)p[2];
          }
#pragma unroll
          for (int k = 0; k < UR5_SUP_K; k++) {
            const int i = base + sl + W * k;
            const real v = px[k] * d.x + py[k] * d.y + pz[k] * d.z;
#if defined(UR5_MPR_DPP_COORDS) && !defined(UR5_EMUL)
            if (i < s.vnum && v > best) { best = v; bi = i; bl = v3(px[k], py[k], pz[k]); }
#else
            if (i < s.vnum && v > best) { best = v; bi = i; }
#endif
          }
        }
      }
#if defined(UR5_MPR_DPP_COORDS) && !defined(UR5_EMUL)
      if constexpr (W >= 8) {
        support_exchange<0xb1>(best, bi, bl);
        support_exchange<0x4e>(best, bi, bl);
        support_exchange<0x141>(best, bi, bl);
        if constexpr (W == 16) support_exchange<0x140>(best, bi, bl);
        l = bl;
      } else l = v3(M.hullvert[s.vadr + bi]);
    }
#else
#ifndef UR5_EMUL
      if constexpr (W >= 8) {
        support_exchange<0xb1>(best, bi);    // quad_perm [1,0,3,2]: lane ^ 1
        support_exchange<0x4e>(best, bi);    // quad_perm [2,3,0,1]: lane ^ 2
        support_exchange<0x141>(best, bi);   // row_half_mirror: lane -> 7 - lane, the other quad of the sub-group
        if constexpr (W == 16) support_exchange<0x140>(best, bi);   // row_mirror: lane -> 15 - lane, the other half of a 16-lane sub-group (one DPP row)
      }
#endif
      l = v3(M.hullvert[s.vadr + bi]);
    }
#endif
    return s.pos + mul(s.mat, l) + dir * ((real)0.5 * s.margin);
  }
#ifndef UR5_EMUL
  template <int CTRL> static __device__ __forceinline__ void support_exchange(real& best, int& bi) {
    const real ob = (real)ur5_dpp<CTRL, 0xf>((double)best);
    const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, false);
    const bool take = ob > best || (ob == best && oi < bi);
    best = take ? ob : best; bi = take ? oi : bi;
  }
  template <int CTRL> static __device__ __forceinline__ void support_exchange(real& best, int& bi, v3& bl) {
    const real ob = (real)ur5_dpp<CTRL, 0xf>((double)best);
    const int oi = __builtin_amdgcn_update_dpp(0, bi, CTRL, 0xf, 0xf, false);
    const real ox = (real)ur5_dpp<CTRL, 0xf>((double)bl.x), oy = (real)ur5_dpp<CTRL, 0xf>((double)bl.y), oz = (real)ur5_dpp<CTRL, 0xf>((double)bl.z);
    const bool take = ob > best || (ob == best && oi < bi);
    best = take ? ob : best; bi = take ? oi : bi;
    bl.x = take ? ox : bl.x; bl.y = take ? oy : bl.y; bl.z = take ? oz : bl.z;
  }
#endif
  struct MV { v3 v, a, b; };
  template <int W = 1> UR5_FN MV msupport(const Shape& A, const Shape& B, v3 dir, int sl = 0) const { UR5_STRICT;
    MV r;
    r.a = support<W>(A, dir, sl); r.b = support<W>(B, -dir, sl); r.v = r.a - r.b;
    return r;
  }
  // Minkowski portal refinement; same scheme, tolerances and result definition as oracle mpr_penetration()
  template <int W = 1> UR5_MPR_ATTR bool mpr(const Shape& A, const Shape& B, real* depth, v3* dir_out, v3* pos_out, int sl = 0) const { UR5_STRICT;
    const real tol = (real)1e-6;
    const int maxit = 50;
    MV v0, v1, v2, v3_, v4;
    v0.a = A.center; v0.b = B.center; v0.v = v0.a - v0.b;
    if (norm(v0.v) < (real)1e-12) v0.v = v3((real)1e-5, 0, 0);
    v3 dir = normalized(-v0.v);
    v1 = msupport<W>(A, B, dir, sl);
    if (dot(v1.v, dir) <= 0) return false;
    dir = cross(v0.v, v1.v);
    if (norm(dir) < (real)1e-12 * maxv((real)1, norm(v0.v) * norm(v1.v))) {
      v3 d = normalized(-v0.v);
      *depth = dot(v1.v, d); *dir_out = d; *pos_out = (v1.a + v1.b) * (real)0.5;
      return true;
    }
    dir = normalized(dir);
    v2 = msupport<W>(A, B, dir, sl);
    if (dot(v2.v, dir) <= 0) return false;
    dir = normalized(cross(v1.v - v0.v, v2.v - v0.v));
    if (dot(dir, v0.v) > 0) { MV t = v1; v1 = v2; v2 = t; dir = -dir; }
    for (int it = 0;; it++) {
      if (it > maxit) return false;
      v3_ = msupport<W>(A, B, dir, sl);
      if (dot(v3_.v, dir) <= 0) return false;
      bool cont = false;
      if (dot(cross(v1.v, v3_.v), v0.v) < 0) { v2 = v3_; cont = true; }
      else if (dot(cross(v3_.v, v2.v), v0.v) < 0) { v1 = v3_; cont = true; }
      if (!cont) break;
      dir = normalized(cross(v1.v - v0.v, v2.v - v0.v));
    }
    bool hit = false;
    for (int it = 0;; it++) {
      dir = normalized(cross(v2.v - v1.v, v3_.v - v1.v));
      if (dot(dir, v0.v) > 0) dir = -dir;
      real d1 = dot(v1.v, dir);
      if (d1 >= 0) hit = true;
      v4 = msupport<W>(A, B, dir, sl);
      real d4 = dot(v4.v, dir);
      if (!hit && d4 < 0) return false;
      if (d4 - d1 <= tol || it >= maxit) {
        if (!hit) return false;
        v3 p = dir * d1;
        v3 e1 = v2.v - v1.v, e2 = v3_.v - v1.v, ep = p - v1.v;
        real a11 = dot(e1, e1), a12 = dot(e1, e2), a22 = dot(e2, e2), b1 = dot(ep, e1), b2 = dot(ep, e2);
        real det = a11 * a22 - a12 * a12;
        real w2 = (real)1 / 3, w3 = (real)1 / 3;
        if (fabs(det) > (real)1e-30) { w2 = (a22 * b1 - a12 * b2) / det; w3 = (a11 * b2 - a12 * b1) / det; }
        real w1 = (real)1 - w2 - w3;
        w1 = maxv(w1, (real)0); w2 = maxv(w2, (real)0); w3 = maxv(w3, (real)0);
        real ws = w1 + w2 + w3;
        if (ws < (real)1e-30) { w1 = w2 = w3 = (real)1 / 3; ws = 1; }
        w1 /= ws; w2 /= ws; w3 /= ws;
        v3 pa = v1.a * w1 + v2.a * w2 + v3_.a * w3, pb = v1.b * w1 + v2.b * w2 + v3_.b * w3;
        *depth = d1; *dir_out = dir; *pos_out = (pa + pb) * (real)0.5;
        return true;
      }
      v3 c = cross(v4.v, v0.v);
      if (dot(v1.v, c) > 0) { if (dot(v2.v, c) > 0) v1 = v4; else v3_ = v4; }
      else { if (dot(v3_.v, c) > 0) v2 = v4; else v1 = v4; }
    }
  }

  // ---- narrow phase. Contacts go to a Sink: mode 0 only counts them, mode 1 writes them to their LDS slots. Every candidate
  // pair is evaluated twice (count -> wave prefix sum -> write) so that no per-lane contact array (= scratch memory) is needed;
  // the one expensive routine, MPR, produces a single contact that is kept in registers between the two passes.
  struct Sink { int mode, slot, n, g1, g2, pair; };
  struct Single { bool hit; v3 pos, normal; real dist; int sat_code; bool sat_flip; real sat_best; };
  UR5_FN void emit(Sink& k, v3 pos, v3 normal, real dist) const {
    // one pass: a slot is claimed with an LDS atomic counter. Only this wavefront touches the counter, so the order is
    // reproducible (it differs from the oracle's pair order, which only permutes floating-point sums downstream).
#ifdef UR5_EMUL
    int c = S.ncon++;
#else
    int c = __hip_atomic_fetch_add(&S.ncon, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    if (c < UR5_MAXCON) {
      pos.store(S.cpos[c]);
      normal.store(S.cframe[c]);          // tangents, friction, condim, bodies: make_constraints(), one lane per contact
      S.cdist[c] = dist;
      S.cg1[c] = k.g1; S.cg2[c] = k.g2;
#if defined(UR5_MANY) && !defined(UR5_EMUL)
      S.cA[c] = k.pair * 8 + k.n;   // sort key of sort_contacts() (a pair emits at most 8 contacts); cA proper is written by make_constraints
#endif
    }
    k.n++;
  }
  // box-box: separating-axis test, then the vertices of (incident face) n (reference face) enumerated directly -- incident
  // corners inside the reference rectangle, reference corners inside the incident rectangle, edge/edge crossings -- which is
  // the vertex set Sutherland-Hodgman clipping (oracle collide_box_box) produces, without its run-time-indexed polygon arrays.
  // separating-axis test of two oriented boxes (15 axes, standard |R| formulation). Returns false when an axis separates them
  // by more than margin; otherwise the axis of least penetration: code 0-2 face of A, 3-5 face of B, 6+3i+j edge i x edge j
  // (an edge axis must beat the best face axis by 5 % to be chosen), its signed overlap `best` and whether it points B->A.
  struct Sat { int code; bool flip; real best; };
  UR5_FN bool box_sat(const GeomPose& A, v3 a, const GeomPose& B, v3 b, real margin, Sat& o) const { UR5_STRICT;
    v3 t = B.pos - A.pos;
    real R[3][3], Q[3][3], tA[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      tA[i] = dot(t, A.mat.col(i));
#pragma unroll
      for (int j = 0; j < 3; j++) { R[i][j] = dot(A.mat.col(i), B.mat.col(j)); Q[i][j] = fabs(R[i][j]); }
    }
    real best = -1e300; int code = -1; bool flip = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      real s = fabs(tA[i]) - (a[i] + b.x * Q[i][0] + b.y * Q[i][1] + b.z * Q[i][2]);
      if (s > margin) return false;
      if (s > best) { best = s; code = i; flip = tA[i] < 0; }
    }
#pragma unroll
    for (int j = 0; j < 3; j++) {
      real tb = tA[0] * R[0][j] + tA[1] * R[1][j] + tA[2] * R[2][j];
      real s = fabs(tb) - (b[j] + a.x * Q[0][j] + a.y * Q[1][j] + a.z * Q[2][j]);
      if (s > margin) return false;
      if (s > best) { best = s; code = 3 + j; flip = tb < 0; }
    }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        real l2 = (real)1 - R[i][j] * R[i][j];
        if (l2 < (real)1e-12) continue;
        real il = (real)1 / sqrt(l2);
        real tl = (tA[i2] * R[i1][j] - tA[i1] * R[i2][j]) * il;
        real ra = (a[i1] * Q[i2][j] + a[i2] * Q[i1][j]) * il, rb = (b[j1] * Q[i][j2] + b[j2] * Q[i][j1]) * il;
        real s = fabs(tl) - (ra + rb);
        if (s > margin) return false;
        if (s > best + (real)1e-6 + (real)0.05 * fabs(best)) { best = s; code = 6 + 3 * i + j; flip = tl < 0; }
      }
    o.code = code; o.flip = flip; o.best = best;
    return true;
  }
  // box-box: SAT (cached in `sat` between the count and the write pass), then the vertices of (incident face) n (reference
  // face) enumerated directly -- incident corners inside the reference rectangle, reference corners inside the incident
  // rectangle, edge/edge crossings -- the vertex set Sutherland-Hodgman clipping (oracle collide_box_box) produces, without
  // its run-time-indexed polygon arrays.
  UR5_BOXBOX_ATTR void box_box(const GeomPose& A, v3 a, const GeomPose& B, v3 b, real margin, Sink& out, Sat& sat) const { UR5_STRICT;
    if (out.mode == 0) { if (!box_sat(A, a, B, b, margin, sat)) { sat.code = -1; return; } }
    if (sat.code < 0) return;
    const int code = sat.code;
    const bool flip = sat.flip;
    const real best = sat.best;
    v3 bestn;
    if (code < 3) bestn = A.mat.col(code);
    else if (code < 6) bestn = B.mat.col(code - 3);
    else bestn = normalized(cross(A.mat.col((code - 6) / 3), B.mat.col((code - 6) % 3)));
    v3 n = flip ? -bestn : bestn;
    if (code >= 6) {
      int i = (code - 6) / 3, j = (code - 6) % 3;
      v3 ea = A.pos, eb = B.pos;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (k != i) ea = ea + A.mat.col(k) * ((dot(n, A.mat.col(k)) > 0 ? (real)1 : (real)-1) * a[k]);
        if (k != j) eb = eb - B.mat.col(k) * ((dot(n, B.mat.col(k)) > 0 ? (real)1 : (real)-1) * b[k]);
      }
      v3 ua = A.mat.col(i), ub = B.mat.col(j), w = ea - eb;
      real uaub = dot(ua, ub), q1 = dot(ua, w), q2 = dot(ub, w), den = (real)1 - uaub * uaub;
      real sa = 0, sb = 0;
      if (den > (real)1e-12) { sa = (uaub * q2 - q1) / den; sb = (q2 - uaub * q1) / den; }
      sa = clampv(sa, -a[i], a[i]); sb = clampv(sb, -b[j], b[j]);
      emit(out, ((ea + ua * sa) + (eb + ub * sb)) * (real)0.5, n, best);
      return;
    }
    bool refA = code < 3;
    int ax = refA ? code : code - 3;
    const GeomPose &Rr = refA ? A : B, &Ri = refA ? B : A;
    v3 r = refA ? a : b, in = refA ? b : a;
    v3 nref = refA ? n : -n;
    int iax = 0; real bd = -1;
#pragma unroll
    for (int k = 0; k < 3; k++) { real d = fabs(dot(Ri.mat.col(k), nref)); if (d > bd) { bd = d; iax = k; } }
    real isgn = dot(Ri.mat.col(iax), nref) > 0 ? (real)-1 : (real)1;
    v3 ninc = Ri.mat.col(iax) * isgn;
    v3 ic = Ri.pos + ninc * in[iax];
    int iu = (iax + 1) % 3, iv = (iax + 2) % 3, ru = (ax + 1) % 3, rv = (ax + 2) % 3;
    v3 Iu = Ri.mat.col(iu), Iv = Ri.mat.col(iv), Ru = Rr.mat.col(ru), Rv = Rr.mat.col(rv);
    real inu = in[iu], inv_ = in[iv], hu = r[ru], hv = r[rv];
    real rsgn = dot(Rr.mat.col(ax), nref) > 0 ? (real)1 : (real)-1;
    v3 rc = Rr.pos + Rr.mat.col(ax) * (rsgn * r[ax]);
    // incident corners (same cyclic order as the oracle) and their reference-face coordinates
    v3 q0 = ic + Iu * inu + Iv * inv_, q1 = ic - Iu * inu + Iv * inv_, q2 = ic - Iu * inu - Iv * inv_, q3 = ic + Iu * inu - Iv * inv_;
    real u0 = dot(q0 - rc, Ru), u1 = dot(q1 - rc, Ru), u2 = dot(q2 - rc, Ru), u3 = dot(q3 - rc, Ru);
    real w0 = dot(q0 - rc, Rv), w1 = dot(q1 - rc, Rv), w2 = dot(q2 - rc, Rv), w3 = dot(q3 - rc, Rv);
    // Ties (a corner ON a reference edge line: equal boxes stacked flush, as in the model's own qpos0) are decided once, here: a
    // coordinate within `tie` of +-h IS +-h. Then every vertex of the closed intersection polygon has exactly one owner below --
    // incident corners in the closed rectangle; reference corners in the closed incident face that are not also incident corners;
    // crossings strictly inside both edges -- which is the vertex set the oracle's clipping yields whichever way the ties round.
    const real tie = (real)(sizeof(real) == 8 ? 1e-9 : 1e-5);
#define UR5_SNAP(x, h) x = fabs(fabs(x) - h) <= tie ? (x < 0 ? -h : h) : x;
    UR5_SNAP(u0, hu) UR5_SNAP(u1, hu) UR5_SNAP(u2, hu) UR5_SNAP(u3, hu) UR5_SNAP(w0, hv) UR5_SNAP(w1, hv) UR5_SNAP(w2, hv) UR5_SNAP(w3, hv)
#undef UR5_SNAP
#define UR5_INC(q, u, w) if (fabs(u) <= hu && fabs(w) <= hv) { real d = dot(q - rc, nref); if (d < margin) emit(out, q - nref * ((real)0.5 * d), n, d); }
    UR5_INC(q0, u0, w0) UR5_INC(q1, u1, w1) UR5_INC(q2, u2, w2) UR5_INC(q3, u3, w3)
#undef UR5_INC
    // the incident face lies entirely inside the reference face (a box resting on a larger one): the four corners are the
    // whole intersection polygon -- no reference corner can be inside it and no edges cross
    if (fabs(u0) <= hu && fabs(w0) <= hv && fabs(u1) <= hu && fabs(w1) <= hv && fabs(u2) <= hu && fabs(w2) <= hv && fabs(u3) <= hu && fabs(w3) <= hv) return;
    real den = dot(nref, ninc);
    if (fabs(den) > (real)1e-9) {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        v3 c0 = rc + Ru * ((k & 1) ? hu : -hu) + Rv * ((k & 2) ? hv : -hv);
        real d = dot(ic - c0, ninc) / den;
        v3 pc = c0 + nref * d;
        real cu = fabs(dot(pc - ic, Iu)), cv = fabs(dot(pc - ic, Iv));
        bool on_corner = fabs(cu - inu) <= tie && fabs(cv - inv_) <= tie;      // coincides with an incident corner: emitted above
        if (cu <= inu + tie && cv <= inv_ + tie && !on_corner && d < margin) emit(out, pc - nref * ((real)0.5 * d), n, d);
      }
    }
    // incident edge (qa -> qb) against the four reference edge lines
#define UR5_EDGE(qa, ua, wa, qb, ub, wb)                                                                          \
    {                                                                                                               \
      _Pragma("unroll") for (int sd = 0; sd < 4; sd++) {                                                            \
        real La = sd < 2 ? ua : wa, Lb = sd < 2 ? ub : wb, lim = (sd < 2 ? hu : hv) * ((sd & 1) ? (real)-1 : (real)1); \
        real Oa = sd < 2 ? wa : ua, Ob = sd < 2 ? wb : ub, olim = sd < 2 ? hv : hu;                                 \
        if ((La - lim) * (Lb - lim) < 0) {                                                                          \
          real tt = (lim - La) / (Lb - La);                                                                         \
          if (fabs(Oa + tt * (Ob - Oa)) < olim - tie) {                                                             \
            v3 pe = qa + (qb - qa) * tt;                                                                            \
            real d = dot(pe - rc, nref);                                                                            \
            if (d < margin) emit(out, pe - nref * ((real)0.5 * d), n, d);                                           \
          }                                                                                                         \
        }                                                                                                           \
      }                                                                                                             \
    }
    UR5_EDGE(q0, u0, w0, q1, u1, w1) UR5_EDGE(q1, u1, w1, q2, u2, w2) UR5_EDGE(q2, u2, w2, q3, u3, w3) UR5_EDGE(q3, u3, w3, q0, u0, w0)
#undef UR5_EDGE
  }
  // ---- capsule helpers (same restatements as oracle collide_plane_capsule / sphere_capsule / capsule_capsule / capsule_box)
  UR5_FN void sphere_sphere_at(Sink& out, v3 p1, real r1, v3 p2, real r2, real margin) const { UR5_STRICT;
    v3 d = p2 - p1;
    real len = norm(d), dist = len - r1 - r2;
    if (dist >= margin) return;
    v3 n = len > (real)1e-12 ? d * ((real)1 / len) : v3(1, 0, 0);
    emit(out, p1 + n * (r1 + (real)0.5 * dist), n, dist);
  }
  // sphere (centre c, radius r) against box (B, s): signed distance; emits the contact when asked to and closer than margin
  UR5_FN real sphere_box_at(Sink& out, v3 c, real r, const GeomPose& B, v3 s, real margin, bool do_emit) const { UR5_STRICT;
    v3 cl = mulT(B.mat, c - B.pos);
    v3 p(clampv(cl.x, -s.x, s.x), clampv(cl.y, -s.y, s.y), clampv(cl.z, -s.z, s.z));
    v3 d = p - cl;
    real len = norm(d);
    if (len > (real)1e-12) {
      real dist = len - r;
      if (do_emit && dist < margin) { v3 n = mul(B.mat, d * ((real)1 / len)); emit(out, c + n * (r + (real)0.5 * dist), n, dist); }
      return dist;
    }
    int ax = 0; real best = 1e300;
    for (int i = 0; i < 3; i++) { real g = s[i] - fabs(cl[i]); if (g < best) { best = g; ax = i; } }
    v3 el; el.set(ax, cl[ax] >= 0 ? (real)1 : (real)-1);
    v3 e = mul(B.mat, el);
    if (do_emit) emit(out, c + e * ((real)0.5 * (best - r)), -e, -best - r);
    return -best - r;
  }
  UR5_BIG void narrow(int g1, int g2, real margin, Sink& out, Single& keep, int pair = -1) { UR5_STRICT;
    int t1 = M.g_type[g1], t2 = M.g_type[g2];
    GeomPose A = geom_pose(g1), B = geom_pose(g2);
    if (t1 == UR5_GEOM_PLANE) {
      v3 n = A.mat.col(2);
      if (t2 == UR5_GEOM_SPHERE) {
        real r = (real)M.g_size[g2][0];
        real d = dot(B.pos - A.pos, n) - r;
        if (d < margin) emit(out, B.pos - n * (r + (real)0.5 * d), n, d);
      } else if (t2 == UR5_GEOM_BOX) {
        v3 s(M.g_size[g2]);
        int cnt = 0;
        for (int k = 0; k < 8 && cnt < 4; k++) {
          v3 l((k & 1) ? s.x : -s.x, (k & 2) ? s.y : -s.y, (k & 4) ? s.z : -s.z);
          v3 v = B.pos + mul(B.mat, l);
          real d = dot(v - A.pos, n);
          if (d < margin) { emit(out, v - n * ((real)0.5 * d), n, d); cnt++; }
        }
      } else if (t2 == UR5_GEOM_CAPSULE) {
        const real r = (real)M.g_size[g2][0], h = (real)M.g_size[g2][1];
        const v3 ax = B.mat.col(2);
        for (int e = 0; e < 2; e++) {
          v3 c = B.pos + ax * (e == 0 ? h : -h);
          real d = dot(c - A.pos, n) - r;
          if (d < margin) emit(out, c - n * (r + (real)0.5 * d), n, d);
        }
      } else {
        Shape s = make_shape(g2, 0);
        v3 v = support(s, -n);
        real d = dot(v - A.pos, n);
        if (d < margin) emit(out, v - n * ((real)0.5 * d), n, d);
      }
    } else if (t1 == UR5_GEOM_SPHERE && t2 == UR5_GEOM_CAPSULE) {
      const v3 ax = B.mat.col(2);
      const real h = (real)M.g_size[g2][1];
      const real t = clampv(dot(A.pos - B.pos, ax), -h, h);
      sphere_sphere_at(out, A.pos, (real)M.g_size[g1][0], B.pos + ax * t, (real)M.g_size[g2][0], margin);
    } else if (t1 == UR5_GEOM_CAPSULE && t2 == UR5_GEOM_CAPSULE) {
      const v3 a1 = A.mat.col(2), a2 = B.mat.col(2), w = A.pos - B.pos;
      const real h1 = (real)M.g_size[g1][1], h2 = (real)M.g_size[g2][1], r1 = (real)M.g_size[g1][0], r2 = (real)M.g_size[g2][0];
      const real b = dot(a1, a2), d = dot(a1, w), e = dot(a2, w), den = (real)1 - b * b;
      if (den < (real)1e-6) {   // parallel axes: the overlap of the two segments, one contact at each of its ends
        const real sgn = b >= 0 ? (real)1 : (real)-1, c2 = -d;
        real t_lo = maxv(-h1, c2 - h2), t_hi = minv(h1, c2 + h2);
        if (t_lo > t_hi) { real tm = clampv(c2, -h1, h1); t_lo = t_hi = tm; }
        const int cnt = t_hi - t_lo > (real)1e-9 ? 2 : 1;
        for (int k = 0; k < cnt; k++) {
          real t1p = k == 0 ? t_lo : t_hi;
          real t2p = clampv(sgn * (t1p - c2), -h2, h2);
          sphere_sphere_at(out, A.pos + a1 * t1p, r1, B.pos + a2 * t2p, r2, margin);
        }
      } else {
        real t1p = clampv((b * e - d) / den, -h1, h1);
        real t2p = clampv(e + b * t1p, -h2, h2);
        t1p = clampv(b * t2p - d, -h1, h1);
        sphere_sphere_at(out, A.pos + a1 * t1p, r1, B.pos + a2 * t2p, r2, margin);
      }
    } else if (t1 == UR5_GEOM_CAPSULE && t2 == UR5_GEOM_BOX) {
      const v3 ax = A.mat.col(2), s(M.g_size[g2]);
      const real r = (real)M.g_size[g1][0], h = (real)M.g_size[g1][1];
      const real d_hi = sphere_box_at(out, A.pos + ax * h, r, B, s, margin, false), d_lo = sphere_box_at(out, A.pos - ax * h, r, B, s, margin, false);
      if (d_hi < margin && d_lo < margin) {   // lying against a face: the two end spheres
        sphere_box_at(out, A.pos + ax * h, r, B, s, margin, true);
        sphere_box_at(out, A.pos - ax * h, r, B, s, margin, true);
      } else {   // the point of the segment nearest to the box (the distance is convex along the segment: golden-section search)
        const real gr = (real)0.6180339887498949;
        real lo = -h, hi = h, x1 = hi - gr * (hi - lo), x2 = lo + gr * (hi - lo);
        real f1 = sphere_box_at(out, A.pos + ax * x1, r, B, s, margin, false), f2 = sphere_box_at(out, A.pos + ax * x2, r, B, s, margin, false);
        for (int it = 0; it < 40; it++) {
          if (f1 < f2) { hi = x2; x2 = x1; f2 = f1; x1 = hi - gr * (hi - lo); f1 = sphere_box_at(out, A.pos + ax * x1, r, B, s, margin, false); }
          else { lo = x1; x1 = x2; f1 = f2; x2 = lo + gr * (hi - lo); f2 = sphere_box_at(out, A.pos + ax * x2, r, B, s, margin, false); }
        }
        real t = (real)0.5 * (lo + hi);
        const real fm = minv(f1, f2);
        if (d_hi <= fm) t = h; else if (d_lo <= fm) t = -h;
        sphere_box_at(out, A.pos + ax * t, r, B, s, margin, true);
      }
    } else if (t1 == UR5_GEOM_SPHERE && t2 == UR5_GEOM_SPHERE) {
      v3 d = B.pos - A.pos;
      real len = norm(d), r1 = (real)M.g_size[g1][0], r2 = (real)M.g_size[g2][0];
      real dist = len - r1 - r2;
      if (dist < margin) {
        v3 n = len > (real)1e-12 ? d * ((real)1 / len) : v3(1, 0, 0);
        emit(out, A.pos + n * (r1 + (real)0.5 * dist), n, dist);
      }
    } else if (t1 == UR5_GEOM_SPHERE && t2 == UR5_GEOM_BOX) {
      real r = (real)M.g_size[g1][0];
      v3 s(M.g_size[g2]);
      v3 cl = mulT(B.mat, A.pos - B.pos);
      v3 p(clampv(cl.x, -s.x, s.x), clampv(cl.y, -s.y, s.y), clampv(cl.z, -s.z, s.z));
      v3 d = p - cl;
      real len = norm(d);
      if (len > (real)1e-12) {
        real dist = len - r;
        if (dist < margin) { v3 n = mul(B.mat, d * ((real)1 / len)); emit(out, A.pos + n * (r + (real)0.5 * dist), n, dist); }
      } else {
        int ax = 0; real best = 1e300;
        for (int i = 0; i < 3; i++) { real g = s[i] - fabs(cl[i]); if (g < best) { best = g; ax = i; } }
        v3 el; el.set(ax, cl[ax] >= 0 ? (real)1 : (real)-1);
        v3 e = mul(B.mat, el);
        emit(out, A.pos + e * ((real)0.5 * (best - r)), -e, -best - r);
      }
    } else if (t1 == UR5_GEOM_BOX && t2 == UR5_GEOM_BOX) {
      Sat sat;
      sat.code = keep.sat_code; sat.flip = keep.sat_flip; sat.best = keep.sat_best;
      box_box(A, v3(M.g_size[g1]), B, v3(M.g_size[g2]), margin, out, sat);
      keep.sat_code = sat.code; keep.sat_flip = sat.flip; keep.sat_best = sat.best;
    } else {
#ifndef UR5_EMUL
      // general convex pair. Pairs with a mesh hull (a support call scans 70 / 120 / 400 vertices) are queued for the cooperative MPR pass of
      // collision_body, 8 lanes per pair. Pairs of analytic shapes (the piles' cylinders) have O(1) support functions: sharing them between lanes
      // would only reduce the pairs in flight, so the many-object kernel runs those here, one pair per lane.
#ifdef UR5_MANY
      const bool coop = t1 == UR5_GEOM_MESH || t2 == UR5_GEOM_MESH;
#else
      const bool coop = true;
#endif
      if (coop) {
        const int q = __hip_atomic_fetch_add(&S.ncouple, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (q < UR5_MAXCON) S.couple[q] = pair; else S.status |= UR5_ST_CAND_OVERFLOW;   // more such pairs than list slots: flagged, never silently dropped
        (void)keep;
      } else
#endif
      {
        if (out.mode == 0) {
          Shape sa = make_shape(g1, margin), sb = make_shape(g2, margin);
          real depth;
          keep.hit = mpr(sa, sb, &depth, &keep.normal, &keep.pos);
          keep.dist = margin - depth;
          if (keep.hit && !(keep.dist < margin)) keep.hit = false;
        }
        if (keep.hit) emit(out, keep.pos, keep.normal, keep.dist);
      }
    }
  }
  UR5_FN v3 geom_position(int g) const { const int dg = M.g_dg[g]; return dg < 0 ? v3(M.g_pos[g]) : v3(S.dgpos[dg]); }
  UR5_BIG bool cull(int g1, int g2, real margin) const {  // true = cannot touch
    UR5_STRICT;
    int t1 = M.g_type[g1], t2 = M.g_type[g2];
    real r1 = (real)M.g_rbound[g1], r2 = (real)M.g_rbound[g2];
    if (t1 != UR5_GEOM_PLANE) {   // bounding spheres first, from the two positions alone: most pairs end here, before any rotation matrix is fetched
      v3 d0 = geom_position(g2) - geom_position(g1);
      real rr0 = r1 + r2 + margin;
      if (dot(d0, d0) > rr0 * rr0) return true;
    }
    GeomPose A = geom_pose(g1), B = geom_pose(g2);
    if (t1 == UR5_GEOM_PLANE) return dot(B.pos - A.pos, A.mat.col(2)) > r2 + margin;
    v3 d = B.pos - A.pos;
    real rr = r1 + r2 + margin;
    if (dot(d, d) > rr * rr) return true;
    // conservative OBB refinement (keeps the finger/plate pair out of the narrow phase while the gripper is high above it)
    if (t2 == UR5_GEOM_BOX && dist_point_box(A.pos, B, v3(M.g_size[g2])) > r1 + margin) return true;
    if (t1 == UR5_GEOM_BOX && dist_point_box(B.pos, A, v3(M.g_size[g1])) > r2 + margin) return true;
    // hull pairs go to MPR (expensive): first separate their oriented bounding boxes. g_size of a mesh = half extents of its hull's TIGHT box, whose
    // centre g_boxc is off the geom origin (the gripper base's mesh origin is at its back end: an origin-centred box would be twice as long and reach
    // the grasped object in every grasp). A conservative test: any separating axis it finds also makes MPR report "no contact".
    if (t2 == UR5_GEOM_MESH && (t1 == UR5_GEOM_BOX || t1 == UR5_GEOM_MESH)) {
      Sat tmp;
      if (t1 == UR5_GEOM_MESH) A.pos = A.pos + mul(A.mat, v3(M.g_boxc[g1]));
      B.pos = B.pos + mul(B.mat, v3(M.g_boxc[g2]));
      if (!box_sat(A, v3(M.g_size[g1]), B, v3(M.g_size[g2]), margin, tmp)) return true;
    }
    return false;
  }
  // The rotation-invariant part of cull() with `slack` added to every bound: true = the pair cannot pass cull() as long as each of its moving geoms stays
  // within slack / 2 of where it is now, whatever its orientation becomes. (Bounding spheres about the geom origins; the point-to-box distance only against
  // STATIC boxes -- the plate and the bins' walls, whose bounding spheres contain the whole scene.)
  UR5_BIG bool cull_loose(int g1, int g2, real margin, real slack) const {
    const int t1 = M.g_type[g1], t2 = M.g_type[g2];
    const real r1 = (real)M.g_rbound[g1], r2 = (real)M.g_rbound[g2];
    const v3 p1 = geom_position(g1), p2 = geom_position(g2);
    if (t1 == UR5_GEOM_PLANE) { m3 A; A.load(M.g_mat[g1]); return dot(p2 - p1, A.col(2)) > r2 + margin + slack; }
    const v3 d = p2 - p1;
    const real rr = r1 + r2 + margin + slack;
    if (dot(d, d) > rr * rr) return true;
    if (t2 == UR5_GEOM_BOX && M.g_dg[g2] < 0) { GeomPose B = geom_pose(g2); if (dist_point_box(p1, B, v3(M.g_size[g2])) > r1 + margin + slack) return true; }
    if (t1 == UR5_GEOM_BOX && M.g_dg[g1] < 0) { GeomPose A = geom_pose(g1); if (dist_point_box(p2, A, v3(M.g_size[g1])) > r2 + margin + slack) return true; }
    return false;
  }
  UR5_FN void invalidate_pair_cache() {
#ifndef UR5_MANY
    S.nsup = -1;
#endif
  }
  UR5_FN static void make_frame(v3 n, real* fr) {
    v3 y = fabs(n.y) < (real)0.5 ? v3(0, 1, 0) : v3(0, 0, 1);
    y = normalized(y - n * dot(n, y));   // |y - n (n.y)| >= 0.86: never degenerate
    n.store(fr); y.store(fr + 3);   // the third axis is cross(n, y), recomputed where needed
  }
  UR5_FN int body_of_geom(int g) const {  // cbody index (robot dof or nrd + object) or -1 for static
    int k = M.g_kind[g];
    return k == UR5_KIND_STATIC ? -1 : (k == UR5_KIND_ROBOT ? M.g_owner[g] : M.nrd + M.g_owner[g]);
  }

#if !defined(UR5_EMUL)
  // General convex pairs (hull against hull / box / ...): Minkowski portal refinement with 8 lanes per pair. The lanes of a sub-group run the
  // same MPR on the same pair (sub-group-uniform control flow) and share the hull scans of its support calls; GS / 8 pairs are in flight at
  // once. S.couple / S.ncouple (filled by narrow()) are free until make_constraints rebuilds them. Its own function in the 256-register
  // kernel: the portal (five Minkowski points with their witnesses) and two shapes are ~170 registers by themselves.
  UR5_CALL void mpr_pass_fn() { mpr_pass_body(); }
  UR5_BIG void mpr_pass_body() { UR5_STRICT;
    {
      const int nm = S.ncouple < UR5_MAXCON ? S.ncouple : UR5_MAXCON;
      constexpr int W = UR5_MPR_W;
      static_assert(W == 8 || W == 16, "sub-groups are aligned groups of 8 or 16 lanes (DPP quad / half-row / row exchanges)");
      const int sub = UR5_LANE / W, sl = UR5_LANE & (W - 1);
      for (int base = 0; base < nm; base += GS / W) {
        const int idx = base + sub;
        if (idx < nm) {
          const int p = S.couple[idx];
          Sink sink;
          sink.mode = 0; sink.slot = 0; sink.n = 0; sink.pair = p;
          sink.g1 = M.pair_g1[p]; sink.g2 = M.pair_g2[p];
          const real margin = maxv((real)M.g_margin[sink.g1], (real)M.g_margin[sink.g2]);
          Shape sa = make_shape(sink.g1, margin), sb = make_shape(sink.g2, margin);
          real depth;
          v3 nrm, pos;
          const bool hit = mpr<W>(sa, sb, &depth, &nrm, &pos, sl);
#if defined(UR5_PROFILE) && !(defined(UR5_PROFILE_LEVELS) && defined(UR5_MANY))
          if (sl == 0) UR5_ATOMIC_ADD(&S.prof[PF_X7], 1.0 + 1e-9 * (double)((sa.type == UR5_GEOM_MESH ? sa.vnum : 0) + (sb.type == UR5_GEOM_MESH ? sb.vnum : 0)));   // pairs + 1e-9 x hull vertices per support call
#endif
          const real dist = margin - depth;
          if (hit && dist < margin && sl == 0) emit(sink, pos, nrm, dist);
        }
      }
      SYNC();
    }
  }
#endif
  UR5_CALL void collision_fn() { collision_body(); }
  UR5_FN void collision() { if constexpr (FLAT) collision_body(); else collision_fn(); }
  UR5_PHASE_A void collision_body() {
    if (UR5_LANE == 0) { S.ncon = 0; S.ncand = 0; S.ncouple = 0; }
    SYNC();
    if (!S.contacts_enabled) return;
    PROF_T0();
    // broad phase: ordered compaction of the surviving pairs
    int ncand = 0;
#ifndef UR5_MANY
    // Pair cache: most of the M.npair pairs are far apart for many steps in a row (a box on the pick plate and the walls of the drop bin), yet the scan
    // below costs a step ~21 k cycles -- four trips of dependent model reads -- whatever the scene does. `sup` lists, in pair order, every pair that passes
    // the rotation-invariant tests of cull() with 2 * UR5_SUP_DELTA of slack; while no moving geom has travelled more than UR5_SUP_DELTA since the list was
    // built (S.moved: the sum of |velocity of the geom origin| * h over the steps, an upper bound of its displacement), any pair outside the list still
    // fails cull(), so running cull() over the list alone gives the SAME candidate list, in the same order -- results are bit-identical.
    bool use_cache = false;
    {
      const real h = (real)M.timestep;
      bool over = false;
      PAR(i, M.ndg) {
        const int g = M.dg_geom[i];
        const int b = body_of_geom(g);
        v3 om(S.cvel[b]), vl(S.cvel[b] + 3);
        const v3 v = vl + cross(om, v3(S.dgpos[i]) - body_ref(b));
        // |v| h is the travel to first order; along a step the velocity of a point on the articulated chain turns by up to |omega| h (second-order term
        // <= |omega|^2 r h^2 / 2, ~0.3 % of |v| h at 3 rad/s): 1 % of inflation and a rebuild at 90 % of the slack keep the superset property strict
        const float mv = S.moved[i] + (float)(norm(v) * h * (real)1.01 + (real)1e-9);
        S.moved[i] = mv;
        if (!(mv <= (float)(0.9 * UR5_SUP_DELTA))) over = true;
      }
      if (over) S.nsup = -1;   // benign race: every writer stores the same value
      SYNC();
      use_cache = S.nsup >= 0;
    }
    if (use_cache) {
      const int nsup = S.nsup;
      if (UR5_LANE == 0) S.rec[UR5_REC_MISC + 6] += 1;   // counters[5]: steps whose broad phase ran from the list
#ifdef UR5_EMUL
      for (int i = 0; i < nsup; i++) {
        const int p = S.sup[i];
        const int g1 = M.pair_g1[p], g2 = M.pair_g2[p];
        const real margin = maxv((real)M.g_margin[g1], (real)M.g_margin[g2]);
        if (!cull(g1, g2, margin)) { if (ncand < UR5_MAXCAND) S.cand[ncand] = (short)p; ncand++; }
      }
#else
      for (int i0 = 0; i0 < nsup; i0 += GS) {
        const int i = i0 + UR5_LANE;
        bool keep = false;
        int p = 0;
        if (i < nsup) {
          p = S.sup[i];
          const int g1 = M.pair_g1[p], g2 = M.pair_g2[p];
          const real margin = maxv((real)M.g_margin[g1], (real)M.g_margin[g2]);
          keep = !cull(g1, g2, margin);
        }
        unsigned long long mask = __ballot(keep);
        if constexpr (GS < 64) mask = (mask >> UR5_GBASE) & ((1ull << (GS & 63)) - 1ull);
        const int slot = ncand + __popcll(mask & ((1ull << UR5_LANE) - 1ull));
        if (keep && slot < UR5_MAXCAND) S.cand[slot] = (short)p;
        ncand += __popcll(mask);
      }
#endif
#if defined(UR5_EMUL) || defined(UR5_SIMT)
      {   // test builds: the cached list must equal the full scan's, pair for pair
        SYNC();
        int nfull = 0;
        bool same = true;
        for (int p = 0; p < M.npair; p++) {
          const int g1 = M.pair_g1[p], g2 = M.pair_g2[p];
          if (!cull(g1, g2, maxv((real)M.g_margin[g1], (real)M.g_margin[g2]))) { if (nfull < UR5_MAXCAND && (nfull >= ncand || S.cand[nfull] != (short)p)) same = false; nfull++; }
        }
        if (!same || nfull != ncand) S.status |= UR5_ST_CACHE_MISMATCH;
      }
#endif
    } else {
    int nsup = 0;
#endif
    for (int p0 = 0; p0 < M.npair; p0 += GS) {
#ifdef UR5_EMUL
      for (int p = p0; p < p0 + GS && p < M.npair; p++) {
        int g1 = M.pair_g1[p], g2 = M.pair_g2[p];
        real margin = maxv((real)M.g_margin[g1], (real)M.g_margin[g2]);
        if (!cull(g1, g2, margin)) { if (ncand < UR5_MAXCAND) S.cand[ncand] = (short)p; ncand++; }
#ifndef UR5_MANY
        if (!cull_loose(g1, g2, margin, (real)(2 * UR5_SUP_DELTA))) { if (nsup < UR5_MAXCAND) S.sup[nsup] = (short)p; nsup++; }
#endif
      }
#else
      int p = p0 + UR5_LANE;
      bool keep = false;
#ifndef UR5_MANY
      bool loose = false;
#endif
      if (p < M.npair) {
        int g1 = M.pair_g1[p], g2 = M.pair_g2[p];
        real margin = maxv((real)M.g_margin[g1], (real)M.g_margin[g2]);
        keep = !cull(g1, g2, margin);
#ifndef UR5_MANY
        loose = !cull_loose(g1, g2, margin, (real)(2 * UR5_SUP_DELTA));
#endif
      }
      unsigned long long mask = __ballot(keep);
#if UR5_NT == 64
      if constexpr (GS < 64) mask = (mask >> UR5_GBASE) & ((1ull << (GS & 63)) - 1ull);   // the votes of this scene's lanes
      int slot = ncand + __popcll(mask & ((1ull << UR5_LANE) - 1ull));
      if (keep && slot < UR5_MAXCAND) S.cand[slot] = (short)p;
      ncand += __popcll(mask);
#ifndef UR5_MANY
      {
        unsigned long long lm = __ballot(loose);
        if constexpr (GS < 64) lm = (lm >> UR5_GBASE) & ((1ull << (GS & 63)) - 1ull);
        const int ls = nsup + __popcll(lm & ((1ull << UR5_LANE) - 1ull));
        if (loose && ls < UR5_MAXCAND) S.sup[ls] = (short)p;
        nsup += __popcll(lm);
      }
#endif
#else   // ordered compaction across the wavefronts of the scene
      if ((UR5_LANE & 63) == 0) S.redi[UR5_LANE >> 6] = __popcll(mask);
      SYNC();
      int base = ncand, total = 0;
#pragma unroll
      for (int w = 0; w < UR5_NT / 64; w++) { if (w < (UR5_LANE >> 6)) base += S.redi[w]; total += S.redi[w]; }
      int slot = base + __popcll(mask & ((1ull << (UR5_LANE & 63)) - 1ull));
      if (keep && slot < UR5_MAXCAND) S.cand[slot] = (short)p;
      ncand += total;
      SYNC();
#endif
#endif
    }
#ifndef UR5_MANY
      // the list is valid from this step on (a list that does not fit stays invalid: every step then scans all pairs, as before)
      PAR(i, M.ndg) S.moved[i] = 0;
      if (UR5_LANE == 0) S.nsup = nsup <= UR5_MAXCAND ? nsup : -1;
    }
#endif
    if (UR5_LANE == 0) { S.ncand = ncand < UR5_MAXCAND ? ncand : UR5_MAXCAND; if (ncand > UR5_MAXCAND) S.status |= UR5_ST_CAND_OVERFLOW; }
    if (ncand > UR5_MAXCAND) ncand = UR5_MAXCAND;
    SYNC();
    PROF(PF_BROAD);
    // narrow phase: one candidate per lane, single pass
#ifdef UR5_EMUL
    for (int ci = 0; ci < ncand; ci++) {
#else
    for (int ci = UR5_LANE; ci < ncand; ci += GS) {
#endif
      if (ci < ncand) {
        Sink sink;
        Single keep;
        keep.hit = false; keep.sat_code = -1; keep.sat_flip = false; keep.sat_best = 0;
        sink.mode = 0; sink.slot = 0; sink.n = 0;
        int p = S.cand[ci];
        sink.pair = p;
        sink.g1 = M.pair_g1[p]; sink.g2 = M.pair_g2[p];
        real margin = maxv((real)M.g_margin[sink.g1], (real)M.g_margin[sink.g2]);
        narrow(sink.g1, sink.g2, margin, sink, keep, p);
      }
    }
    SYNC();
    PROFR(PF_X0);   // profile builds: the one-candidate-per-lane pass (analytic pairs, box-box); PF_NARROW then is the cooperative MPR pass
#if !defined(UR5_EMUL)
    if constexpr (FLAT) mpr_pass_body(); else mpr_pass_fn();
#endif
    if (UR5_LANE == 0) {
      int n = S.ncon;
      if (n > UR5_MAXCON) { n = UR5_MAXCON; S.status |= UR5_ST_CONTACT_OVERFLOW; }
      S.ncon = n;
      if (n > S.ncon_max) S.ncon_max = n;
    }
    SYNC();
#if defined(UR5_MANY) && !defined(UR5_EMUL)
    sort_contacts();
#endif
    PROF(PF_NARROW);
  }
#if defined(UR5_MANY) && !defined(UR5_EMUL)
  // Contact slots are claimed with an atomic counter by lanes of four wavefronts (and by the cooperative MPR pass after them), so the slot ORDER of a step
  // depends on how the wavefronts were scheduled -- and every sum over contacts downstream with it. Sorting the contacts by (geom pair, number within the
  // pair) gives the one order the lane emulation and the oracle produce anyway: pair order. One contact per lane: rank by counting, move through registers.
  UR5_FN void sort_contacts() {
    static_assert(UR5_MAXCON <= UR5_NT, "one contact per lane");
    const int n = S.ncon, c = UR5_LANE;
    int rank = 0, g1 = 0, g2 = 0;
    real px = 0, py = 0, pz = 0, nx = 0, ny = 0, nz = 0, dist = 0;
    if (c < n) {
      const int key = S.cA[c];
      for (int j = 0; j < n; j++) rank += S.cA[j] < key ? 1 : 0;
      px = S.cpos[c][0]; py = S.cpos[c][1]; pz = S.cpos[c][2];
      nx = S.cframe[c][0]; ny = S.cframe[c][1]; nz = S.cframe[c][2];
      dist = S.cdist[c]; g1 = S.cg1[c]; g2 = S.cg2[c];
    }
    SYNC();
    if (c < n) {
      S.cpos[rank][0] = px; S.cpos[rank][1] = py; S.cpos[rank][2] = pz;
      S.cframe[rank][0] = nx; S.cframe[rank][1] = ny; S.cframe[rank][2] = nz;
      S.cdist[rank] = dist; S.cg1[rank] = (short)g1; S.cg2[rank] = (short)g2;
    }
    SYNC();
  }
#endif

  // ------------------------------------------------------------------ constraint rows (mj_makeConstraint + mj_makeImpedance [3P])
  // x^p of the impedance sigmoid: p = 2 (MuJoCo's default solimp) is a product; the general pow() is large, so it is kept
  // out of line instead of being expanded at each of the eight places an impedance is evaluated
  UR5_CALL static real pow_any(real x, real p) { return pow(x, p); }
  UR5_FN static real powr(real x, real p) { if constexpr (FLAT && UR5_INL_POW) return p == (real)2 ? x * x : pow(x, p); else return p == (real)2 ? x * x : pow_any(x, p); }
  UR5_FN static real impedance(const double* solimp, real x_abs) {
    real dmin = clampv((real)solimp[0], (real)0.0001, (real)0.9999), dmax = clampv((real)solimp[1], (real)0.0001, (real)0.9999);
    real width = (real)solimp[2], mid = (real)solimp[3], power = (real)solimp[4];
    if (dmin == dmax || width <= (real)1e-15) return (real)0.5 * (dmin + dmax);
    real x = x_abs / width;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    real y;
    if (power == 1) y = x;
    else if (x <= mid) y = powr(x / mid, power) * mid;
    else y = 1 - powr((1 - x) / (1 - mid), power) * (1 - mid);
    return dmin + y * (dmax - dmin);
  }
  UR5_FN void kbi(const double* solref, const double* solimp, real imp, real* K, real* B) const {
    real tc = maxv((real)solref[0], 2 * (real)M.timestep), dr = (real)solref[1];
    real dmax = clampv((real)solimp[1], (real)0.0001, (real)0.9999);
    *K = (real)1 / (dmax * dmax * tc * tc * dr * dr);
    *B = (real)2 / (dmax * tc);
  }
  UR5_FN v3 body_ref(int b) const { return b < M.nrd ? v3(M.ref_point) : v3(S.bpos[b]); }
  // relative twist-space image of a contact: e[k] for the NB base directions given the two body twists (B minus A)
  UR5_FN void contact_image(int c, const real twA[6], const real twB[6], bool hasA, bool hasB, real* e) const {
    v3 p(S.cpos[c]);
    v3 u, w;
    if (hasB) { v3 r = p - body_ref(S.cB[c]); v3 om(twB), vl(twB + 3); u = vl + cross(om, r); w = om; }
    if (hasA) { v3 r = p - body_ref(S.cA[c]); v3 om(twA), vl(twA + 3); u = u - (vl + cross(om, r)); w = w - om; }
    v3 n(S.cframe[c]), t1(S.cframe[c] + 3);
    v3 t2 = cross(n, t1);
    e[0] = dot(n, u); e[1] = dot(t1, u); e[2] = dot(t2, u); e[3] = dot(n, w);
    if constexpr (NB > 4) { e[NB - 2] = dot(t1, w); e[NB - 1] = dot(t2, w); }
  }
#if defined(UR5_MANY) && !defined(UR5_EMUL)
  // side lists: for every accumulator slot the contact sides (side id 2 c + side) that act on its body, in contact order. Lane = side id, so a counting sort
  // by slot that keeps lane order is a stable sort: one ballot per slot gives every side its rank among the same-slot sides of its wavefront, the wavefronts'
  // counts per slot (LDS) give the offsets. Schedule-free: ballots and integer sums only.
  UR5_FN void build_side_lists() {
    static_assert(2 * UR5_MAXCON <= 2 * UR5_NT, "at most two rounds of one side per lane");
    const short* key = &S.csl[0][0];
    const int ns2 = 2 * S.ncon, nsl = nslot(), wv = UR5_LANE >> 6, wl = UR5_LANE & 63;
    int carry_round = 0;                                                     // sides of my slot placed by earlier rounds (only for > UR5_NT sides)
    for (int r0 = 0; r0 == 0 || r0 < ns2; r0 += UR5_NT) {
      const int sd = r0 + UR5_LANE;
      const int my = sd < ns2 ? (int)key[sd] : -1;
      int rank = 0;
      for (int sl = 0; sl < nsl; sl++) {
        const unsigned long long m = __ballot(my == sl);
        if (my == sl) rank = __popcll(m & ((1ull << wl) - 1ull));
        if (wl == 0) S.slot_cnt[wv][sl] = (unsigned char)__popcll(m);
      }
      SYNC();
      if (r0 == 0) {
        PAR(sl, nsl + 1) {                                                   // first round: the slots' offsets need the totals of ALL rounds -> count the sides beyond this round directly
          int cnt = 0;
          for (int q = 0; q < sl; q++) for (int w = 0; w < UR5_NT / 64; w++) cnt += S.slot_cnt[w][q];
          for (int j = UR5_NT; j < ns2; j++) cnt += (key[j] >= 0 && key[j] < sl) ? 1 : 0;
          S.slot_ptr[sl] = (short)cnt;
        }
        SYNC();
      }
      if (my >= 0) {
        int o = S.slot_ptr[my] + carry_round + rank;
        for (int w = 0; w < wv; w++) o += S.slot_cnt[w][my];
        S.side_list[o] = (short)sd;
        S.side_pos[sd] = (short)o;
      }
      if (r0 + UR5_NT >= ns2) break;
      SYNC();
      // (more than UR5_NT sides: never seen on the reference's piles, 80 contacts at most) the next round's sides of a slot go behind this round's
      { int t = 0; const int mine = (r0 + UR5_NT + UR5_LANE) < ns2 ? (int)key[r0 + UR5_NT + UR5_LANE] : -1; if (mine >= 0) for (int w = 0; w < UR5_NT / 64; w++) t += S.slot_cnt[w][mine]; carry_round += t; }
      SYNC();
    }
  }
#endif
  UR5_CALL void make_constraints_fn() { make_constraints_body(); }
  UR5_FN void make_constraints() { if constexpr (FLAT) make_constraints_body(); else make_constraints_fn(); }
  UR5_PHASE_B void make_constraints_body() {
#ifdef UR5_MANY
    PROF_T0();   // profile builds of the many-object kernel: sub-intervals x1..x5 of `rows`
#endif
#if !defined(UR5_EMUL) && UR5_NT == 64
    // special rows (joint equality, violated joint / slide limits): one candidate per lane -- [0, neq) equalities, then 2 sides of
    // every robot joint, then 2 sides of every object slide -- compacted in candidate order (= the serial order below) with a ballot
    // (candidates are taken GS at a time when there are more of them than lanes in the scene's group)
    const int ncand_rows = M.neq + 2 * M.nrd + 6 * M.nobj;
    int nrows = 0;
    for (int t0 = 0; t0 < ncand_rows; t0 += GS) {
      const int t = t0 + UR5_LANE;
      bool on = false;
      int d1 = 0, d2 = -1, uni = 1;
      real c1 = 0, c2 = 0, Dr = 0, aref = 0;
      if (t < M.neq) {
        const int e = t;
        d1 = M.eq_d1[e]; d2 = M.eq_d2[e];
        const double* pc = M.eq_poly[e];
        real xq = qpos()[d2] - (real)M.rd_qpos0[d2];
        real poly = (real)pc[0] + xq * ((real)pc[1] + xq * ((real)pc[2] + xq * ((real)pc[3] + xq * (real)pc[4])));
        real dpoly = (real)pc[1] + xq * (2 * (real)pc[2] + xq * (3 * (real)pc[3] + xq * 4 * (real)pc[4]));
        real pos = (qpos()[d1] - (real)M.rd_qpos0[d1]) - poly;
        real imp = impedance(M.eq_solimp[e], fabs(pos)), K, B;
        kbi(M.eq_solref[e], M.eq_solimp[e], imp, &K, &B);
        real dA = (real)M.rd_invweight[d1] + (real)M.rd_invweight[d2];
        real R = maxv((real)1e-15, (1 - imp) * dA / imp);
        real vel = qvel()[d1] - dpoly * qvel()[d2];
        on = true; c1 = 1; c2 = -dpoly; uni = 0; Dr = (real)1 / R; aref = -B * vel - K * imp * pos;
      } else if (t < ncand_rows) {
        const int u = t - M.neq;
        const bool robot = u < 2 * M.nrd;
        const int v = robot ? u : u - 2 * M.nrd;
        const int side = v & 1, jj = v >> 1;                    // robot: jj = dof; object: jj = 3 * k + slide
        const int k = robot ? 0 : jj / 3, j = robot ? 0 : jj % 3;
        const bool limited = robot ? M.rd_limited[jj] != 0 : (M.obj_kind[k] == 0 && M.obj_limited[k][j] != 0);
        if (limited) {
          const int qa = robot ? jj : M.nrd + 7 * k + j;
          d1 = robot ? jj : M.nrd + 6 * k + j;
          const real lo = robot ? (real)M.rd_lo[jj] : (real)M.obj_lo[k][j], hi = robot ? (real)M.rd_hi[jj] : (real)M.obj_hi[k][j];
          const real dist = side == 0 ? qpos()[qa] - lo : hi - qpos()[qa];
          if (dist < 0) {
            const real sg = side == 0 ? (real)1 : (real)-1;
            real imp = impedance(M.jnt_solimp, fabs(dist)), K, B;
            kbi(M.jnt_solref, M.jnt_solimp, imp, &K, &B);
            const real iw = robot ? (real)M.rd_invweight[jj] : (real)M.obj_invweight[k][0];
            real R = maxv((real)1e-15, (1 - imp) * iw / imp);
            on = true; c1 = sg; Dr = (real)1 / R; aref = -B * sg * qvel()[d1] - K * imp * dist;
          }
        }
      }
      unsigned long long mask = __ballot(on);
      if constexpr (GS < 64) mask = (mask >> UR5_GBASE) & ((1ull << (GS & 63)) - 1ull);   // the votes of this scene's lanes
      const int slot = nrows + __popcll(mask & ((1ull << UR5_LANE) - 1ull));
      if (on && slot < UR5_MAXSR) {
        S.sr_d1[slot] = d1; S.sr_d2[slot] = d2; S.sr_c1[slot] = c1; S.sr_c2[slot] = c2; S.sr_uni[slot] = uni; S.sr_D[slot] = Dr; S.sr_aref[slot] = aref;
      }
      nrows += __popcll(mask);
    }
    if (UR5_LANE == 0) { S.nsr = nrows < UR5_MAXSR ? nrows : UR5_MAXSR; if (nrows > UR5_MAXSR) S.status |= UR5_ST_ROW_OVERFLOW; }
#else
    // ... or lane 0 builds them one after the other (lane emulation, the many-object variant)
    if (UR5_LANE == 0) {
      int ns = 0;
      for (int e = 0; e < M.neq; e++) {
        int d1 = M.eq_d1[e], d2 = M.eq_d2[e];
        const double* pc = M.eq_poly[e];
        real xq = qpos()[d2] - (real)M.rd_qpos0[d2];
        real poly = (real)pc[0] + xq * ((real)pc[1] + xq * ((real)pc[2] + xq * ((real)pc[3] + xq * (real)pc[4])));
        real dpoly = (real)pc[1] + xq * (2 * (real)pc[2] + xq * (3 * (real)pc[3] + xq * 4 * (real)pc[4]));
        real pos = (qpos()[d1] - (real)M.rd_qpos0[d1]) - poly;
        real imp = impedance(M.eq_solimp[e], fabs(pos)), K, B;
        kbi(M.eq_solref[e], M.eq_solimp[e], imp, &K, &B);
        real dA = (real)M.rd_invweight[d1] + (real)M.rd_invweight[d2];
        real R = maxv((real)1e-15, (1 - imp) * dA / imp);
        real vel = qvel()[d1] - dpoly * qvel()[d2];
        S.sr_d1[ns] = d1; S.sr_d2[ns] = d2; S.sr_c1[ns] = 1; S.sr_c2[ns] = -dpoly; S.sr_uni[ns] = 0;
        S.sr_D[ns] = (real)1 / R; S.sr_aref[ns] = -B * vel - K * imp * pos;
        ns++;
      }
      for (int d = 0; d < M.nrd; d++) {
        if (!M.rd_limited[d]) continue;
        for (int side = 0; side < 2; side++) {
          real dist = side == 0 ? qpos()[d] - (real)M.rd_lo[d] : (real)M.rd_hi[d] - qpos()[d];
          if (dist >= 0) continue;
          if (ns >= UR5_MAXSR) { S.status |= UR5_ST_ROW_OVERFLOW; continue; }
          real sg = side == 0 ? (real)1 : (real)-1;
          real imp = impedance(M.jnt_solimp, fabs(dist)), K, B;
          kbi(M.jnt_solref, M.jnt_solimp, imp, &K, &B);
          real R = maxv((real)1e-15, (1 - imp) * (real)M.rd_invweight[d] / imp);
          S.sr_d1[ns] = d; S.sr_d2[ns] = -1; S.sr_c1[ns] = sg; S.sr_c2[ns] = 0; S.sr_uni[ns] = 1;
          S.sr_D[ns] = (real)1 / R; S.sr_aref[ns] = -B * sg * qvel()[d] - K * imp * dist;
          ns++;
        }
      }
      for (int k = 0; k < M.nobj; k++) {
        if (M.obj_kind[k] != 0) continue;
        for (int j = 0; j < 3; j++) {
          if (!M.obj_limited[k][j]) continue;
          int qa = M.nrd + 7 * k + j, d = M.nrd + 6 * k + j;
          for (int side = 0; side < 2; side++) {
            real dist = side == 0 ? qpos()[qa] - (real)M.obj_lo[k][j] : (real)M.obj_hi[k][j] - qpos()[qa];
            if (dist >= 0) continue;
          if (ns >= UR5_MAXSR) { S.status |= UR5_ST_ROW_OVERFLOW; continue; }
            real sg = side == 0 ? (real)1 : (real)-1;
            real imp = impedance(M.jnt_solimp, fabs(dist)), K, B;
            kbi(M.jnt_solref, M.jnt_solimp, imp, &K, &B);
            real R = maxv((real)1e-15, (1 - imp) * (real)M.obj_invweight[k][0] / imp);
            S.sr_d1[ns] = d; S.sr_d2[ns] = -1; S.sr_c1[ns] = sg; S.sr_c2[ns] = 0; S.sr_uni[ns] = 1;
            S.sr_D[ns] = (real)1 / R; S.sr_aref[ns] = -B * sg * qvel()[d] - K * imp * dist;
            ns++;
          }
        }
      }
      S.nsr = ns;
    }
#endif
#ifdef UR5_MANY
    PROFR(PF_X1);   // special rows (lane 0)
#endif
    PAR(c, S.ncon) {
      int g1 = S.cg1[c], g2 = S.cg2[c];
      make_frame(v3(S.cframe[c]), S.cframe[c]);
      S.cA[c] = body_of_geom(g1); S.cB[c] = body_of_geom(g2);
#if defined(UR5_MANY) && !defined(UR5_EMUL)
      S.csl[c][0] = (short)(S.cA[c] >= 0 ? slot_of(S.cA[c]) : -1); S.csl[c][1] = (short)(S.cB[c] >= 0 ? slot_of(S.cB[c]) : -1);
#endif
      S.cdim[c] = M.g_condim[g1] > M.g_condim[g2] ? M.g_condim[g1] : M.g_condim[g2];
      for (int j = 0; j < (NB > 4 ? 3 : 2); j++) S.cfri[c][j] = maxv((real)M.g_friction[g1][j], (real)M.g_friction[g2][j]);
      real margin = maxv((real)M.g_margin[g1], (real)M.g_margin[g2]);
      real pos = S.cdist[c];
      double solref[2], solimp[5];
      for (int i = 0; i < 2; i++) solref[i] = 0.5 * (M.g_solref[g1][i] + M.g_solref[g2][i]);
      for (int i = 0; i < 5; i++) solimp[i] = 0.5 * (M.g_solimp[g1][i] + M.g_solimp[g2][i]);
      real imp = impedance(solimp, fabs(pos - margin)), K, B;
      kbi(solref, solimp, imp, &K, &B);
      real tran = (real)(M.g_invw[g1][0] + M.g_invw[g2][0]);
      real fri0 = S.cfri[c][0];
      real R0 = maxv((real)1e-15, (1 - imp) * (tran + fri0 * fri0 * tran) / imp);
      real R;
      if (S.cdim[c] == 1) R = maxv((real)1e-15, (1 - imp) * tran / imp);
      else { real mu = fri0 * sqrt((real)1 / maxv((real)1e-15, (real)M.impratio)); R = 2 * mu * mu * R0; }
      S.cD[c] = (real)1 / R;
      real ckr = K * imp * (pos - margin);
      bool hasA = S.cA[c] >= 0, hasB = S.cB[c] >= 0;
      real vb[NB];
      contact_image(c, hasA ? S.cvel[S.cA[c]] : S.cvel[0], hasB ? S.cvel[S.cB[c]] : S.cvel[0], hasA, hasB, vb);
      for (int k = 0; k < NB; k++) S.ceoff_()[c][k] = B * vb[k];
      S.ceoff_()[c][0] += ckr;
    }
    SYNC();
#ifdef UR5_MANY
    PROFR(PF_X2);   // contact rows
#endif
#if !defined(UR5_EMUL) && !defined(UR5_MANY)
    {   // contacts between two movable bodies, compacted in contact order with a ballot; body / coupling masks with LDS atomics
      static_assert(UR5_MAXCON <= 32, "one contact per lane of the smallest lane group");
      if (UR5_LANE == 0) { S.bodymask = 0; S.cplmask = 0; }
      SYNC();
      const int c = UR5_LANE;
      bool cp = false;
      if (c < S.ncon) {
        const int A = S.cA[c], B = S.cB[c];
        cp = A >= 0 && B >= 0;
        unsigned long long bm = 0;
        if (A >= 0) bm |= 1ull << A;
        if (B >= 0) bm |= 1ull << B;
        if (bm) __hip_atomic_fetch_or(&S.bodymask, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cp) {
          unsigned cm = 0;
          if (A >= M.nrd) cm |= 1u << (A - M.nrd);
          if (B >= M.nrd) cm |= 1u << (B - M.nrd);
          if ((A < M.nrd) != (B < M.nrd)) cm |= 1u << 31;
          if (cm) __hip_atomic_fetch_or(&S.cplmask, cm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
      unsigned long long mask = __ballot(cp);
      if constexpr (GS < 64) mask = (mask >> UR5_GBASE) & ((1ull << (GS & 63)) - 1ull);
      if (cp) S.couple[__popcll(mask & ((1ull << UR5_LANE) - 1ull))] = c;
      if (UR5_LANE == 0) S.ncouple = __popcll(mask);
    }
#elif defined(UR5_EMUL) || !defined(UR5_MANY)
    if (UR5_LANE == 0) {
      int nc = 0;
      unsigned long long bm = 0;
      for (int c = 0; c < S.ncon; c++) {
        if (S.cA[c] >= 0 && S.cB[c] >= 0) S.couple[nc++] = c;
        if (S.cA[c] >= 0) bm |= 1ull << S.cA[c];
        if (S.cB[c] >= 0) bm |= 1ull << S.cB[c];
      }
      S.ncouple = nc;
      S.bodymask = bm;
    }
#else
    {   // many-object kernel: contacts between two movable bodies, compacted in contact order across the scene's wavefronts (one contact per lane)
      static_assert(UR5_MAXCON <= UR5_NT, "one contact per lane");
      if (UR5_LANE == 0) S.bodymask = 0;
      SYNC();
      const int c = UR5_LANE;
      bool cp = false;
      if (c < S.ncon) {
        const int A = S.cA[c], B = S.cB[c];
        cp = A >= 0 && B >= 0;
        unsigned long long bm = 0;
        if (A >= 0) bm |= 1ull << A;
        if (B >= 0) bm |= 1ull << B;
        if (bm) __hip_atomic_fetch_or(&S.bodymask, bm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      const unsigned long long mask = __ballot(cp);
      if ((UR5_LANE & 63) == 0) S.redi[UR5_LANE >> 6] = __popcll(mask);
      SYNC();
      int base = 0, total = 0;
#pragma unroll
      for (int w = 0; w < UR5_NT / 64; w++) { if (w < (UR5_LANE >> 6)) base += S.redi[w]; total += S.redi[w]; }
      if (cp) S.couple[base + __popcll(mask & ((1ull << (UR5_LANE & 63)) - 1ull))] = c;
      if (UR5_LANE == 0) S.ncouple = total;
      build_side_lists();
    }
#endif
    SYNC();
#ifdef UR5_MANY
    PROFR(PF_X3);   // couple list
    // (Round 3 tried to keep the envelope structure across steps -- it is a function of the set of coupled body pairs and of the objects' x-order. Measured on
    // piles: the pair set changes in > 95 % of the steps even after 2 s of settling, resting contacts sit within 1e-5 m of the margin at which they are detected
    // and come and go every step, so the structure is rebuilt every step: 130 k of the 154 k cycles of this phase.)
    envelope_structure();
    PROFR(PF_X5);   // envelope structure
#endif
  }

  // ------------------------------------------------------------------ Newton solver pieces
  UR5_FN real row_mu(int c, int k) const { return k <= 2 ? S.cfri[c][0] : (k == 3 ? S.cfri[c][1] : S.cfri[c][NB > 4 ? 2 : 1]); }
  // twists of every body for a dof-space vector, then base images (with or without the aref offsets)
  UR5_CALL void images_fn(const real* vec, bool offset, real (*out)[NB], real* srout) { images_body(vec, offset, out, srout); }
  UR5_FN void images(const real* vec, bool offset, real (*out)[NB], real* srout) { if constexpr (FLAT && UR5_INL_IMAGES) images_body(vec, offset, out, srout); else images_fn(vec, offset, out, srout); }
  UR5_FN void images_body(const real* vec, bool offset, real (*out)[NB], real* srout) {
    PAR(sl, nslot()) {
      const int b = body_of_slot(sl);
      if (b < M.nrd) {
        real v[6] = {0, 0, 0, 0, 0, 0};
        for (int e = 0; e < M.nrd; e++) if (M.rd_anc[b] >> e & 1u) { real q = vec[e]; for (int i = 0; i < 6; i++) v[i] += S.cdof[e][i] * q; }
        for (int i = 0; i < 6; i++) S.tw_()[sl][i] = v[i];
      } else {
        int va = M.nrd + 6 * (b - M.nrd);
        m3 R; R.load(S.bmat[b]);
        mul(R, v3(vec[va + 3], vec[va + 4], vec[va + 5])).store(S.tw_()[sl]);
        v3(vec[va], vec[va + 1], vec[va + 2]).store(S.tw_()[sl] + 3);
      }
    }
    SYNC();
    PAR(c, S.ncon) {
      bool hasA = S.cA[c] >= 0, hasB = S.cB[c] >= 0;
      real e[NB];
      contact_image(c, hasA ? S.tw_()[slot_of(S.cA[c])] : S.tw_()[0], hasB ? S.tw_()[slot_of(S.cB[c])] : S.tw_()[0], hasA, hasB, e);
      if (offset) for (int k = 0; k < NB; k++) e[k] += S.ceoff_()[c][k];
      for (int k = 0; k < NB; k++) out[c][k] = e[k];
    }
    PAR(s, S.nsr) {
      real v = S.sr_c1[s] * vec[S.sr_d1[s]];
      if (S.sr_d2[s] >= 0) v += S.sr_c2[s] * vec[S.sr_d2[s]];
      srout[s] = offset ? v - S.sr_aref[s] : v;
    }
    SYNC();
  }
  UR5_CALL void mat_vec_M_fn(const real* vec, real* out) { mat_vec_M_body(vec, out); }
  UR5_FN void mat_vec_M(const real* vec, real* out) { if constexpr (FLAT && UR5_INL_MATVEC) mat_vec_M_body(vec, out); else mat_vec_M_fn(vec, out); }
  UR5_FN void mat_vec_M_body(const real* vec, real* out) {
    PAR(i, M.nv) {
      if (i < M.nrd) { real s = 0; for (int e = 0; e < M.nrd; e++) s += S.Mr[i][e] * vec[e]; out[i] = s; }
      else out[i] = S.Mobj[i - M.nrd] * vec[i];
    }
    SYNC();
  }
  // constraint cost of the current images (ce, sr_jar) shifted by alpha along (cde, sr_jv); also first/second derivative
  struct Cost3 { real c, d1, d2; };
  UR5_CALL Cost3 constraint_cost_fn(real alpha) { return constraint_cost_body(alpha); }
  UR5_FN Cost3 constraint_cost(real alpha) { if constexpr (FLAT && UR5_INL_COST) return constraint_cost_body(alpha); else return constraint_cost_fn(alpha); }
  UR5_FN Cost3 constraint_cost_body(real alpha) {
    real c0 = 0, g1 = 0, g2 = 0;
    PAR(c, S.ncon) {
      real D = S.cD[c];
      real e0 = S.ce[c][0] + alpha * S.cde_()[c][0], j0 = S.cde_()[c][0];
      if (S.cdim[c] == 1) {
        if (e0 < 0) { c0 += (real)0.5 * D * e0 * e0; g1 += D * e0 * j0; g2 += D * j0 * j0; }
      } else {
#pragma unroll
        for (int k = 1; k < NB; k++) {
          if (k >= S.cdim[c]) continue;
          real mu = row_mu(c, k);
          real ek = mu * (S.ce[c][k] + alpha * S.cde_()[c][k]), jk = mu * S.cde_()[c][k];
          real rp = e0 + ek, rm = e0 - ek;
          if (rp < 0) { c0 += (real)0.5 * D * rp * rp; g1 += D * rp * (j0 + jk); g2 += D * (j0 + jk) * (j0 + jk); }
          if (rm < 0) { c0 += (real)0.5 * D * rm * rm; g1 += D * rm * (j0 - jk); g2 += D * (j0 - jk) * (j0 - jk); }
        }
      }
    }
    PAR(s, S.nsr) {
      real r = S.sr_jar[s] + alpha * S.sr_jv_()[s], j = S.sr_jv_()[s];
      if (!S.sr_uni[s] || r < 0) { c0 += (real)0.5 * S.sr_D[s] * r * r; g1 += S.sr_D[s] * r * j; g2 += S.sr_D[s] * j * j; }
    }
    Cost3 r;
#if !defined(UR5_EMUL) && UR5_NT > 64
    block_sum3(c0, g1, g2);   // one pair of barriers for the three sums
    r.c = c0; r.d1 = g1; r.d2 = g2;
#else
    r.c = WAVE_SUM(c0); r.d1 = WAVE_SUM(g1); r.d2 = WAVE_SUM(g2);
#endif
    return r;
  }
  UR5_FN real gauss_cost(const real* xv, const real* Ma) {
    real g = 0;
    PAR(i, M.nv) g += (real)0.5 * (Ma[i] - S.fs[i]) * (xv[i] - S.as[i]);
    return WAVE_SUM(g);
  }
  // base-space forces fb and the "arrow" weight matrix W (w[0] = W_00, w[k] = W_0k, w[NB-1+k] = W_kk) of contact c at S.ce
  UR5_FN void contact_weights(int c, real* fb, real* w) const {
    real D = S.cD[c], e0 = S.ce[c][0];
#pragma unroll
    for (int k = 0; k < NB; k++) fb[k] = 0;
#pragma unroll
    for (int k = 0; k < 2 * NB - 1; k++) w[k] = 0;
    if (S.cdim[c] == 1) {
      if (e0 < 0) { fb[0] = -D * e0; w[0] = D; }
    } else {
      const int cdim = S.cdim[c];
#pragma unroll
      for (int k = 1; k < NB; k++) {   // compile-time trip count: fb / w stay in registers (a run-time bound made them scratch-memory arrays)
        if (k >= cdim) continue;
        real mu = row_mu(c, k), ek = mu * S.ce[c][k];
        real rp = e0 + ek, rm = e0 - ek;
        real ap = rp < 0 ? (real)1 : (real)0, am = rm < 0 ? (real)1 : (real)0;
        real fp = -D * rp * ap, fm = -D * rm * am;
        fb[0] += fp + fm; fb[k] += mu * (fp - fm);
        w[0] += D * (ap + am); w[k] = D * mu * (ap - am); w[NB - 1 + k] = D * mu * mu * (ap + am);
      }
    }
  }
  // unit twist [rot; lin] of dof-local index i of cbody b (zero when the dof does not move the body)
  UR5_FN bool unit_twist(int b, int i, real* tw) const {
    if (b < M.nrd) {
      if (!(M.rd_anc[b] >> i & 1u)) { tw[0] = tw[1] = tw[2] = tw[3] = tw[4] = tw[5] = 0; return false; }
      tw[0] = S.cdof[i][0]; tw[1] = S.cdof[i][1]; tw[2] = S.cdof[i][2]; tw[3] = S.cdof[i][3]; tw[4] = S.cdof[i][4]; tw[5] = S.cdof[i][5];
    } else {
      m3 R; R.load(S.bmat[b]);
      v3 cc = R.col(i < 3 ? 0 : i - 3);
      tw[0] = i < 3 ? (real)0 : cc.x; tw[1] = i < 3 ? (real)0 : cc.y; tw[2] = i < 3 ? (real)0 : cc.z;
      tw[3] = i == 0 ? (real)1 : (real)0; tw[4] = i == 1 ? (real)1 : (real)0; tw[5] = i == 2 ? (real)1 : (real)0;
    }
    return true;
  }
  // (J e_ia)^T W (J e_ib) for side-A dof ia and side-B dof ib of contact c
  UR5_BIG real couple_term(int c, int A, int ia, int B, int ib) const {
    real twA[6], twB[6], zero[6] = {0, 0, 0, 0, 0, 0}, ea[NB], eb[NB];
    if (!unit_twist(A, ia, twA) || !unit_twist(B, ib, twB)) return 0;
    contact_image(c, twA, zero, true, false, ea);  // carries the minus sign of side A
    contact_image(c, zero, twB, false, true, eb);
    real fb[NB], w[2 * NB - 1];
    contact_weights(c, fb, w);
    real v = w[0] * ea[0] * eb[0];
    for (int k = 1; k < NB; k++) v += w[k] * (ea[0] * eb[k] + ea[k] * eb[0]) + w[NB - 1 + k] * ea[k] * eb[k];
    return v;
  }
#if defined(UR5_MANY) && !defined(UR5_EMUL)
  // The many-object kernel STAGES and GATHERS instead of scattering with atomics: the contact lanes compute exactly the terms of the scatter below and store them
  // per side -- wrench terms [side][6] in the panel area (only the factorisation uses it), Hessian terms [side][21] in the envelope area (G is only built when the
  // envelope is about to be re-assembled) --, then lane (slot, entry) adds its slot's sides in list order = contact order. No float atomic, no dependence on the
  // wavefronts' timing; every entry of every slot is written, so nothing is zeroed first. Contacts are staged UR5_GCHUNK at a time (a settled pile has 40-80).
  // Round 5: the staging area is LDS whenever it fits. A side's 6 wrench + 21 Hessian terms are filed at the side's POSITION in the slot lists (sides of static
  // bodies are not staged at all), wrench terms first: 27 doubles x (contacts + contacts between two movable bodies) <= the 2 384 doubles of the Hessian pool, which
  // is dead at this point of an iteration (the envelope is rebuilt after the gather). The second phase then reads a slot's terms from consecutive LDS words instead
  // of from the scene's global scratch -- five trips of (slot, entry) lanes, each of which used to wait for a global-memory round trip. A scene with more sides than
  // fit (never seen: 88) stages in the global scratch as before. (Round 4's chunked LDS staging, 21 and then 44 contacts per round, lost 7.5 % / 5 %: every extra
  // round repeats the five trips. This one never needs a second round.) Same terms, same order of every sum: same bits.
  template <bool STL> UR5_FN void contact_gather_in(const bool doW, const bool doG) {
    const int nsides = S.slot_ptr[nslot()];
    real* const stW = STL ? (real*)S.henv : &S.stw[0][0];
    real* const stG = STL ? (real*)S.henv + 6 * nsides : S.hess + UR5_SCR_STG;
    for (int sd = UR5_LANE; sd < 2 * S.ncon; sd += GS) {          // one lane per SIDE: the two sides of a contact recompute its weights, and finish in half the time
      const int c = sd >> 1, side = sd & 1;
      const int b = side == 0 ? S.cA[c] : S.cB[c];
      if (b < 0) continue;
      const int at = STL ? (int)S.side_pos[sd] : sd;               // where the side's terms are filed
      v3 ax[3] = {v3(S.cframe[c]), v3(S.cframe[c] + 3), cross(v3(S.cframe[c]), v3(S.cframe[c] + 3))};
      real fb[NB], w[2 * NB - 1];
      contact_weights(c, fb, w);
      const real sg = side == 0 ? (real)-1 : (real)1;
      const v3 r = v3(S.cpos[c]) - body_ref(b);
      if (doW) {
        v3 F = ax[0] * fb[0] + ax[1] * fb[1] + ax[2] * fb[2];
        v3 T = ax[0] * fb[3];
        if constexpr (NB > 4) T = T + ax[1] * fb[NB - 2] + ax[2] * fb[NB - 1];
        const v3 Mo = cross(r, F) + T;
        real* o = stW + 6 * at;
        o[0] = sg * Mo.x; o[1] = sg * Mo.y; o[2] = sg * Mo.z; o[3] = sg * F.x; o[4] = sg * F.y; o[5] = sg * F.z;
      }
      if (!doG) continue;
      // F_k (6-vector [rot; lin]): k<3 -> [r x a_k ; a_k], k>=3 -> [a_{k-3} ; 0];  G += sum_kl W_kl F_k F_l^T (arrow-shaped W)
      real Fk[NB][6];
      for (int k = 0; k < 3; k++) {
        v3 ra = cross(r, ax[k]);
        Fk[k][0] = ra.x; Fk[k][1] = ra.y; Fk[k][2] = ra.z; Fk[k][3] = ax[k].x; Fk[k][4] = ax[k].y; Fk[k][5] = ax[k].z;
        if (3 + k < NB) { Fk[3 + k][0] = ax[k].x; Fk[3 + k][1] = ax[k].y; Fk[3 + k][2] = ax[k].z; Fk[3 + k][3] = 0; Fk[3 + k][4] = 0; Fk[3 + k][5] = 0; }
      }
      real* o = stG + 21 * at;
      int ent = 0;
      for (int gi = 0; gi < 6; gi++)
        for (int gj = 0; gj <= gi; gj++, ent++) {
          real v = w[0] * Fk[0][gi] * Fk[0][gj];
          for (int k = 1; k < NB; k++) v += w[k] * (Fk[0][gi] * Fk[k][gj] + Fk[k][gi] * Fk[0][gj]) + w[NB - 1 + k] * Fk[k][gi] * Fk[k][gj];
          o[ent] = v;
        }
    }
    SYNC();
    PAR(idx, nslot() * 27) {
      const int sl = idx / 27, ent = idx - 27 * sl;
      if (ent < 6 ? !doW : !doG) continue;
      real acc = 0;
      const int o1 = S.slot_ptr[sl + 1];
      const real* const st = ent < 6 ? stW + ent : stG + (ent - 6);
      const int stride = ent < 6 ? 6 : 21;
      // a slot's list is in contact order; four sides per trip, their loads issued together (the sum keeps list order: a side past the end contributes an exact zero)
      for (int o = S.slot_ptr[sl]; o < o1; o += 4) {
        real v[4];
        if constexpr (STL) {
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = o + k < o1 ? st[stride * (o + k)] : (real)0;
        } else {
          int sd[4];
#pragma unroll
          for (int k = 0; k < 4; k++) sd[k] = S.side_list[o + k < o1 ? o + k : o1 - 1];
#pragma unroll
          for (int k = 0; k < 4; k++) v[k] = o + k < o1 ? st[stride * sd[k]] : (real)0;
        }
        acc = ((acc + v[0]) + v[1]) + v[2];
        acc += v[3];
      }
      if (ent < 6) S.WB[sl][ent] = acc; else S.G[sl][ent - 6] = acc;
    }
  }
  UR5_FN void contact_gather(const bool doW, const bool doG) {
    if (27 * (int)S.slot_ptr[nslot()] <= L::HENV_DOUBLES && UR5_STG_LDS) contact_gather_in<true>(doW, doG); else contact_gather_in<false>(doW, doG);
  }
#endif
  // every contact lane scatters its two sides: doW -> body wrenches WB (the gradient), doG -> twist-space Hessians G
  UR5_FN void contact_scatter(const bool doW, const bool doG) {
#if defined(UR5_MANY) && !defined(UR5_EMUL)
    contact_gather(doW, doG);
    return;
#endif
    PAR(c, S.ncon) {
      v3 ax[3] = {v3(S.cframe[c]), v3(S.cframe[c] + 3), cross(v3(S.cframe[c]), v3(S.cframe[c] + 3))};
      real fb[NB], w[2 * NB - 1];
      contact_weights(c, fb, w);
#ifdef UR5_EMUL
      S.cfn[c] = fb[0];
#endif
      v3 F = ax[0] * fb[0] + ax[1] * fb[1] + ax[2] * fb[2];
      v3 T = ax[0] * fb[3];
      if constexpr (NB > 4) T = T + ax[1] * fb[NB - 2] + ax[2] * fb[NB - 1];
      for (int side = 0; side < 2; side++) {
        int b = side == 0 ? S.cA[c] : S.cB[c];
        if (b < 0) continue;
        real sg = side == 0 ? (real)-1 : (real)1;
        v3 r = v3(S.cpos[c]) - body_ref(b);
        v3 Mo = cross(r, F) + T;
        const int sl = slot_of(b);
        if (doW) {
          UR5_ATOMIC_ADD(&S.WB[sl][0], sg * Mo.x); UR5_ATOMIC_ADD(&S.WB[sl][1], sg * Mo.y); UR5_ATOMIC_ADD(&S.WB[sl][2], sg * Mo.z);
          UR5_ATOMIC_ADD(&S.WB[sl][3], sg * F.x); UR5_ATOMIC_ADD(&S.WB[sl][4], sg * F.y); UR5_ATOMIC_ADD(&S.WB[sl][5], sg * F.z);
        }
        if (!doG) continue;
        // F_k (6-vector [rot; lin]): k<3 -> [r x a_k ; a_k], k>=3 -> [a_{k-3} ; 0];  G += sum_kl W_kl F_k F_l^T (arrow-shaped W)
        real Fk[NB][6];
        for (int k = 0; k < 3; k++) {
          v3 ra = cross(r, ax[k]);
          Fk[k][0] = ra.x; Fk[k][1] = ra.y; Fk[k][2] = ra.z; Fk[k][3] = ax[k].x; Fk[k][4] = ax[k].y; Fk[k][5] = ax[k].z;
          if (3 + k < NB) { Fk[3 + k][0] = ax[k].x; Fk[3 + k][1] = ax[k].y; Fk[3 + k][2] = ax[k].z; Fk[3 + k][3] = 0; Fk[3 + k][4] = 0; Fk[3 + k][5] = 0; }
        }
        int ent = 0;
        for (int gi = 0; gi < 6; gi++)
          for (int gj = 0; gj <= gi; gj++, ent++) {
            real v = w[0] * Fk[0][gi] * Fk[0][gj];
            for (int k = 1; k < NB; k++) v += w[k] * (Fk[0][gi] * Fk[k][gj] + Fk[k][gi] * Fk[0][gj]) + w[NB - 1 + k] * Fk[k][gi] * Fk[k][gj];
            if (v != 0) UR5_ATOMIC_ADD(&S.G[sl][ent], v);
          }
      }
    }
  }
  // gradient at S.x (images in S.ce / S.sr_jar must be current) and, unless `check` finds it below the tolerance (returns true:
  // converged, nothing else computed), the Newton direction S.search = -H^-1 grad
  UR5_CALL bool newton_direction_fn(const bool check, const real scale, const real tolerance) { return newton_direction_body(check, scale, tolerance); }
  UR5_FN bool newton_direction(const bool check, const real scale, const real tolerance) {
    if constexpr (FLAT) return newton_direction_body(check, scale, tolerance); else return newton_direction_fn(check, scale, tolerance);
  }
  UR5_PHASE_H bool newton_direction_body(const bool check, const real scale, const real tolerance) {
    PROF_T0();
    const int nbod = nb();
    // per-body wrench (gradient) and 6x6 twist-space Hessian accumulators: every contact lane scatters its two sides with
    // LDS float atomics (ds_add_f64). Only this wavefront touches these words, so the sums are reproducible run to run.
#if !defined(UR5_MANY) || defined(UR5_EMUL)
    PAR(idx, nslot() * 27) { int b = idx / 27, ent = idx % 27; if (ent < 6) S.WB[b][ent] = 0; else S.G[b][ent - 6] = 0; }
#endif
#ifdef UR5_MANY
    // The Newton Hessian depends on the iterate only through the SET of active rows (D is fixed within a step), so it is
    // piecewise constant: when no row changed state since the previous iteration, the factor in LDS is still the factor
    // of H and assembly + factorisation are skipped (MuJoCo's Newton updates its factor incrementally for the same reason).
    PAR(c, S.ncon) {
      const real e0 = S.ce[c][0];
      unsigned mask = 0;
      if (S.cdim[c] == 1) mask = e0 < 0 ? 1u : 0u;
      else for (int k = 1; k < S.cdim[c]; k++) { real ek = row_mu(c, k) * S.ce[c][k]; mask |= (e0 + ek < 0 ? 1u : 0u) << (2 * k) | (e0 - ek < 0 ? 2u : 0u) << (2 * k); }
      if (mask != S.cact[c]) { S.cact[c] = (unsigned short)mask; S.act_changed = 1; }
    }
    PAR(s2, S.nsr) {
      const int on = !(S.sr_uni[s2] && S.sr_jar[s2] >= 0);
      if (on != S.sr_act[s2]) { S.sr_act[s2] = on; S.act_changed = 1; }
    }
#endif
    SYNC();
#ifdef UR5_MANY
    // (an envelope in the LDS pool does not outlive its iteration: images() and the staged gather write there. Such a step refactors in every iteration; with an
    // unchanged active set that reproduces the same Hessian and factor bit for bit)
    const bool refactor = S.act_changed != 0 || S.env_inlds;
#else
    const bool refactor = true;
#endif
#if defined(UR5_MANY) && !defined(UR5_EMUL)
    // the staged gather costs two barriers and a walk over the side lists whether it sums 6 or 27 entries per slot: wrench and Hessian terms go through it together
    // whenever the factor has to be rebuilt (the one iteration per step that turns out to be converged builds its G for nothing; the other ~10 save a second pass)
    contact_scatter(true, refactor);
#else
    contact_scatter(true, !check && refactor);   // first iteration: one pass does both
#endif
    SYNC();
    // gradient = Ma - fs - J^T f
    PAR(i, M.nv) {
      real jf = 0;
      if (i < M.nrd) {
        for (int rg = 0; rg < M.nrg; rg++) if (M.rd_desc[i] >> M.rg_body[rg] & 1u) for (int k = 0; k < 6; k++) jf += S.cdof[i][k] * S.WB[rg][k];
      } else {
        int k = (i - M.nrd) / 6, j = (i - M.nrd) % 6, b = M.nrd + k;
        const int sl = M.nrg + k;
        if (j < 3) jf = S.WB[sl][3 + j];
        else { m3 R; R.load(S.bmat[b]); jf = dot(R.col(j - 3), v3(S.WB[sl])); }
      }
      for (int s = 0; s < S.nsr; s++) {
        real r = S.sr_jar[s];
        if (S.sr_uni[s] && r >= 0) continue;
        real f = -S.sr_D[s] * r;
        if (S.sr_d1[s] == i) jf += S.sr_c1[s] * f;
        if (S.sr_d2[s] == i) jf += S.sr_c2[s] * f;
      }
      const real gi = S.Ma[i] - S.fs[i] - jf;
#ifndef UR5_MANY
      S.grad[i] = gi;
#endif
      S.search[i] = gi;
    }
    if (check) {   // iterations after the first: the Hessian is only worth building when the gradient says "not converged"
      real gn = 0;
      PAR(i, M.nv) gn += S.search[i] * S.search[i];
      gn = WAVE_SUM(gn);
      if (scale * sqrt(gn) < tolerance) return true;
#if !defined(UR5_MANY) || defined(UR5_EMUL)
      if (refactor) { contact_scatter(false, true); SYNC(); }
#endif
    }
    PROF(PF_GRADG);
#ifdef UR5_MANY
    if (UR5_LANE == 0) { S.act_changed = 0; if (!refactor) S.nskip++; }   // every lane read the flag before the barrier above
    PAR(i, M.nv) S.Mv_()[pdof(i)] = S.search[i];   // right-hand side in permuted order (search still holds the gradient)
    if (S.env_inlds) {
      if (refactor) { envelope_assemble<true>(); PROF(PF_HASM); envelope_factor<true>(); PROF(PF_CHOL); }
      envelope_solve<true>(refactor); PROF(PF_SOLVE);
    } else {
      if (refactor) { envelope_assemble<false>(); PROF(PF_HASM); envelope_factor<false>(); PROF(PF_CHOL); }
      envelope_solve<false>(refactor); PROF(PF_SOLVE);
    }
    return false;
#else
#ifndef UR5_EMUL
    if (S.ncouple == 0 && M.nrd == UR5_MAXRD) { newton_blockdiag(); PROF(PF_SOLVE); return false; }
#endif
    // Hessian, lower triangle
    const int nv = M.nv, LD = L::LD;
    PAR(idx, nv * nv) { int i = idx / nv, j = idx % nv; if (j <= i) S.H[UR5_HIDX(i, j)] = 0; }
    SYNC();
    PROF(PF_X1);   // zeroing H
    PAR(idx, M.nrd * M.nrd) {
      int d = idx / M.nrd, e = idx % M.nrd;
      if (e > d) continue;
      real v = S.Mr[d][e];
      unsigned common = M.rd_desc[d] & M.rd_desc[e];
      for (int rg = 0; rg < M.nrg; rg++) {
        const int b = M.rg_body[rg];
        if (!(common >> b & 1u) || !(S.bodymask >> b & 1u)) continue;
        for (int i = 0; i < 6; i++) {
          real t = 0;
          for (int j = 0; j < 6; j++) t += S.G[rg][sym6(i, j)] * S.cdof[e][j];
          v += S.cdof[d][i] * t;
        }
      }
      for (int s = 0; s < S.nsr; s++) {
        if (S.sr_uni[s] && S.sr_jar[s] >= 0) continue;
        real cd = (S.sr_d1[s] == d ? S.sr_c1[s] : (real)0) + (S.sr_d2[s] == d ? S.sr_c2[s] : (real)0);
        real ce = (S.sr_d1[s] == e ? S.sr_c1[s] : (real)0) + (S.sr_d2[s] == e ? S.sr_c2[s] : (real)0);
        v += S.sr_D[s] * cd * ce;
      }
      S.H[UR5_HIDX(d, e)] = v;
    }
    PAR(idx, M.nobj * 21) {
      int k = idx / 21, ent = idx % 21, b = M.nrd + k;
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= ent) i++;
      int j = ent - i * (i + 1) / 2;  // dof-local indices, i >= j; 0-2 lin, 3-5 rot
      m3 R; R.load(S.bmat[b]);
      // twist per unit dof: lin j -> [0; e_j], rot j -> [R col_j; 0]
      real ti[6], tj[6];
      {
        v3 cI = R.col(i < 3 ? 0 : i - 3), cJ = R.col(j < 3 ? 0 : j - 3);
        ti[0] = i < 3 ? (real)0 : cI.x; ti[1] = i < 3 ? (real)0 : cI.y; ti[2] = i < 3 ? (real)0 : cI.z;
        ti[3] = i == 0 ? (real)1 : (real)0; ti[4] = i == 1 ? (real)1 : (real)0; ti[5] = i == 2 ? (real)1 : (real)0;
        tj[0] = j < 3 ? (real)0 : cJ.x; tj[1] = j < 3 ? (real)0 : cJ.y; tj[2] = j < 3 ? (real)0 : cJ.z;
        tj[3] = j == 0 ? (real)1 : (real)0; tj[4] = j == 1 ? (real)1 : (real)0; tj[5] = j == 2 ? (real)1 : (real)0;
      }
      real v = 0;
      for (int a = 0; a < 6; a++) { real t = 0; for (int bb = 0; bb < 6; bb++) t += S.G[M.nrg + k][sym6(a, bb)] * tj[bb]; v += ti[a] * t; }
      int di = M.nrd + 6 * k + i, dj = M.nrd + 6 * k + j;
      if (i == j) {
        v += S.Mobj[6 * k + i];
        for (int s = 0; s < S.nsr; s++) if (S.sr_d1[s] == di && !(S.sr_uni[s] && S.sr_jar[s] >= 0)) v += S.sr_D[s] * S.sr_c1[s] * S.sr_c1[s];
      }
      S.H[UR5_HIDX(di, dj)] = v;
    }
    SYNC();
    PROF(PF_X2);   // diagonal blocks; PF_HASM then is the coupling loop
    // coupling blocks: contacts between two movable bodies, one contact at a time (entries may collide across contacts)
    for (int q = 0; q < S.ncouple; q++) {
      int c = S.couple[q], A = S.cA[c], B = S.cB[c];
      int nA = A < M.nrd ? M.nrd : 6, nBd = B < M.nrd ? M.nrd : 6;
      bool both_robot = A < M.nrd && B < M.nrd;
      PAR(idx, nA * nBd) {
        int ia = idx / nBd, ib = idx % nBd;
        if (both_robot && ia < ib) continue;  // (ia, ib) and (ib, ia) land on one entry: the ia > ib lane adds both
        int da = A < M.nrd ? ia : M.nrd + 6 * (A - M.nrd) + ia;
        int db = B < M.nrd ? ib : M.nrd + 6 * (B - M.nrd) + ib;
        real v = couple_term(c, A, ia, B, ib);
        if (both_robot) v = ia == ib ? 2 * v : v + couple_term(c, A, ib, B, ia);
        if (v != 0) {
          if (da >= db) S.H[UR5_HIDX(da, db)] += v; else S.H[UR5_HIDX(db, da)] += v;
        }
      }
      SYNC();
    }
    PROF(PF_HASM);
#ifdef UR5_EMUL   // lane emulation: the same packed H, factored and solved in place (the GPU keeps row i in the registers of lane i)
    for (int j = 0; j < nv; j++) {
      real d = S.H[UR5_HIDX(j, j)];
      for (int k = 0; k < j; k++) d -= S.H[UR5_HIDX(j, k)] * S.H[UR5_HIDX(j, k)];
      d = sqrt(d < (real)1e-15 ? (real)1e-15 : d);
      S.H[UR5_HIDX(j, j)] = d;
      for (int i = j + 1; i < nv; i++) {
        real v = S.H[UR5_HIDX(i, j)];
        for (int k = 0; k < j; k++) v -= S.H[UR5_HIDX(i, k)] * S.H[UR5_HIDX(j, k)];
        S.H[UR5_HIDX(i, j)] = v / d;
      }
    }
    for (int i = 0; i < nv; i++) { real v = S.search[i]; for (int k = 0; k < i; k++) v -= S.H[UR5_HIDX(i, k)] * S.search[k]; S.search[i] = v / S.H[UR5_HIDX(i, i)]; }
    for (int i = nv - 1; i >= 0; i--) { real v = S.search[i]; for (int k = i + 1; k < nv; k++) v -= S.H[UR5_HIDX(k, i)] * S.search[k]; S.search[i] = v / S.H[UR5_HIDX(i, i)]; }
    for (int i = 0; i < nv; i++) S.search[i] = -S.search[i];
#else
    (void)LD;
    if constexpr (FLAT) factor_solve_rows_body<false>(); else factor_solve_rows<false>();
    PROF_RE();   // profile builds: the routine books its own sub-intervals (x3..x6)
#endif
#endif   // UR5_MANY
    return false;
  }

#if !defined(UR5_EMUL) && !defined(UR5_MANY)
  // No contact couples two movable bodies: H = diag(robot 8x8, object 6x6, ...). Each lane builds ITS row of ITS block
  // straight into registers (robot row d: Mr + sum_b cdof_d^T G_b cdof_e + equality/limit rows; object row: M + T^T G T)
  // and the block-parallel Cholesky / solves above produce S.search = -H^-1 grad without any Hessian in LDS.
  __device__ __forceinline__ void newton_blockdiag() {
    PROF_T0();
    const int lane = UR5_LANE, nv = M.nv;
    Blk b;
    real r[UR5_MAXRD];
#pragma unroll
    for (int j = 0; j < UR5_MAXRD; j++) r[j] = 0;
    if (lane < UR5_MAXRD) {
      b.base = 0; b.loc = lane; b.size = UR5_MAXRD;
      const int d = lane;
#pragma unroll
      for (int e = 0; e < UR5_MAXRD; e++) if (e <= d) r[e] = S.Mr[d][e];
      for (int rg = 0; rg < M.nrg; rg++) {
        const int bb = M.rg_body[rg];
        if (!(M.rd_desc[d] >> bb & 1u) || !(S.bodymask >> bb & 1u)) continue;
        real t[6];
#pragma unroll
        for (int i = 0; i < 6; i++) { real a = 0; for (int j = 0; j < 6; j++) a += S.G[rg][sym6(i, j)] * S.cdof[d][j]; t[i] = a; }
#pragma unroll
        for (int e = 0; e < UR5_MAXRD; e++)
          if (e <= d && (M.rd_desc[e] >> bb & 1u)) { real a = 0; for (int i = 0; i < 6; i++) a += S.cdof[e][i] * t[i]; r[e] += a; }
      }
      for (int s = 0; s < S.nsr; s++) {
        if (S.sr_uni[s] && S.sr_jar[s] >= 0) continue;
        real cd = (S.sr_d1[s] == d ? S.sr_c1[s] : (real)0) + (S.sr_d2[s] == d ? S.sr_c2[s] : (real)0);
        if (cd == 0) continue;
#pragma unroll
        for (int e = 0; e < UR5_MAXRD; e++) {
          real ce = (S.sr_d1[s] == e ? S.sr_c1[s] : (real)0) + (S.sr_d2[s] == e ? S.sr_c2[s] : (real)0);
          if (e <= d) r[e] += S.sr_D[s] * cd * ce;
        }
      }
    } else if (lane < nv) {
      const int k = (lane - UR5_MAXRD) / 6, loc = (lane - UR5_MAXRD) % 6, body = UR5_MAXRD + k;
      b.base = UR5_MAXRD + 6 * k; b.loc = loc; b.size = 6;
      m3 R; R.load(S.bmat[body]);
      real ti[6];
      {
        v3 c = R.col(loc < 3 ? 0 : loc - 3);
        ti[0] = loc < 3 ? (real)0 : c.x; ti[1] = loc < 3 ? (real)0 : c.y; ti[2] = loc < 3 ? (real)0 : c.z;
        ti[3] = loc == 0 ? (real)1 : (real)0; ti[4] = loc == 1 ? (real)1 : (real)0; ti[5] = loc == 2 ? (real)1 : (real)0;
      }
      real t[6];
#pragma unroll
      for (int i = 0; i < 6; i++) { real a = 0; for (int j = 0; j < 6; j++) a += S.G[M.nrg + k][sym6(i, j)] * ti[j]; t[i] = a; }
#pragma unroll
      for (int j = 0; j < 6; j++) {
        if (j > loc) continue;
        real v = j < 3 ? t[3 + j] : dot(R.col(j - 3), v3(t[0], t[1], t[2]));
        if (j == loc) {
          v += S.Mobj[6 * k + loc];
          for (int s = 0; s < S.nsr; s++) if (S.sr_d1[s] == lane && !(S.sr_uni[s] && S.sr_jar[s] >= 0)) v += S.sr_D[s] * S.sr_c1[s] * S.sr_c1[s];
        }
        r[j] = v;
      }
    } else { b.base = lane; b.loc = 0; b.size = 0; }
    real g = lane < nv ? S.grad[lane] : (real)0;
    SYNC();     // every lane has read G / grad; H (aliased scratch) may be overwritten now
    PROF(PF_CHOL);     // profile builds: row assembly is booked under "chol", factor + solves under "solve"
    real myinv = blk_cholesky(r, b);
    real x = blk_solve(r, myinv, b, g, S.H);
    if (lane < nv) S.search[lane] = -x;
    SYNC();
  }
#endif

#if !defined(UR5_EMUL) && !defined(UR5_MANY)
  // broadcast lane `src` (group-uniform index inside the scene's lane group) of a double: two v_readlane when the group is the whole
  // wavefront, a lane shuffle (ds_bpermute) when two scenes share it -- both scenes are at the same column, sources src and 32 + src
  static __device__ __forceinline__ real bcast(real v, int src) {
    if constexpr (GS != 64) return shfl_d(v, UR5_GBASE + src);
    double d = (double)v;
    int lo = __double2loint(d), hi = __double2hiint(d);
    lo = __builtin_amdgcn_readlane(lo, src);
    hi = __builtin_amdgcn_readlane(hi, src);
    return (real)__hiloint2double(hi, lo);
  }
  // H (lower triangle in LDS) -> S.search = -H^-1 grad. Row i of the factor lives in the registers of lane i; the pivot row
  // is broadcast with v_readlane, so the whole factorisation runs without touching LDS. BLOCKDIAG: no contact couples two
  // movable bodies, H = diag(robot 8x8, object 6x6 ...) and every column only looks back to the start of its own block.
  template <bool BLOCKDIAG> __device__ __noinline__ void factor_solve_rows() { factor_solve_rows_body<BLOCKDIAG>(); }
  template <bool BLOCKDIAG> __device__ __forceinline__ void factor_solve_rows_body() {
    constexpr int N = NV_;
    static_assert(NV_ <= GS, "one Hessian row per lane of the scene's group");
    const int lane = UR5_LANE, nv = M.nv;
    PROF_T0();
    real Lrow[N];
#pragma unroll
    for (int j = 0; j < N; j++) Lrow[j] = (lane < nv && j <= lane) ? S.H[UR5_HIDX(lane, j)] : (j == lane ? (real)1 : (real)0);
    real myinv = 1;
    PROF(PF_X3);   // rows of H into registers
    // Structure of H (dof order: robot 0-7, then 6 per object): an object that shares no contact with another movable body only has its
    // own diagonal block; the others ("coupled": S.cplmask) may reach the robot columns (if any contact joins the robot and an object)
    // and the blocks of earlier coupled objects (direct coupling or fill-in). Every skipped product has an exactly zero factor, so the
    // result is the dense factorisation's, bit for bit; a box in the gripper needs ~140 of the 496 column products.
    const unsigned cpl = S.cplmask;
#pragma unroll
    for (int j = 0; j < N; j++) {
      const int bj = j < UR5_MAXRD ? -1 : (j - UR5_MAXRD) / 6;                      // compile-time after unrolling
      const int kb = j < UR5_MAXRD ? 0 : UR5_MAXRD + 6 * bj;
      real sacc = Lrow[j];
      if (!BLOCKDIAG && bj >= 0 && (cpl >> bj & 1u)) {
        if (cpl >> 31) {
#pragma unroll
          for (int k = 0; k < UR5_MAXRD; k++) sacc -= Lrow[k] * bcast(Lrow[k], j);
        }
#pragma unroll
        for (int c = 0; c < (N - UR5_MAXRD) / 6; c++) {
          if (c < bj && (cpl >> c & 1u)) {
#pragma unroll
            for (int k = UR5_MAXRD + 6 * c; k < UR5_MAXRD + 6 * c + 6; k++) sacc -= Lrow[k] * bcast(Lrow[k], j);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < j; k++) if (k >= kb) sacc -= Lrow[k] * bcast(Lrow[k], j);
      real djj = bcast(sacc, j);
      djj = djj < (real)1e-15 ? (real)1e-15 : djj;
      real inv = rsqrt(djj);
      inv = inv * ((real)1.5 - (real)0.5 * djj * inv * inv);
      Lrow[j] = lane == j ? djj * inv : (lane > j ? sacc * inv : (real)0);
      if (lane == j) myinv = inv;
    }
    PROF(PF_X4);   // factorisation
    real b = lane < nv ? S.search[lane] : (real)0;
#pragma unroll
    for (int j = 0; j < N; j++) {
      real yj = bcast(b * myinv, j);
      b = lane == j ? yj : (lane > j ? b - Lrow[j] * yj : b);
    }
    PROF(PF_X5);   // forward substitution
    // transpose the factor through LDS (H is free now): lane j then holds column j, i.e. row j of L^T
    SYNC();
    if (lane < nv) {
#pragma unroll
      for (int j = 0; j < N; j++) if (j <= lane) S.H[UR5_HIDX(lane, j)] = Lrow[j];
    }
    SYNC();
#pragma unroll
    for (int k = 0; k < N; k++) Lrow[k] = (k >= lane && k < nv && lane < nv) ? S.H[UR5_HIDX(k, lane)] : (real)0;
#pragma unroll
    for (int k = N - 1; k >= 0; k--) {
      real xk = bcast(b * myinv, k);
      b = lane == k ? xk : (lane < k ? b - Lrow[k] * xk : b);
    }
    if (lane < nv) S.search[lane] = -b;
    SYNC();
    PROF(PF_X6);   // transposition through LDS + backward substitution
  }
#endif

#ifdef UR5_MANY
  // ------------------------------------------------------------------ many-object scenes: envelope (skyline) Newton Hessian
  // H = blockdiag(robot 8x8, object 6x6 ...) + one coupling block per pair of movable bodies in contact. The blocks are
  // ordered objects-sorted-along-x, robot last, so that touching bodies are close in the ordering; every row stores the
  // columns from the first block it is coupled with up to the diagonal, and the Cholesky factor fills exactly that
  // envelope. A settled 40-object pile needs 1.5-2.5 k doubles, so the envelope lives in LDS (S.henv); only when it does
  // not fit (UR5_HENV_CAP) the same code runs on the scene's global-memory scratch (S.hess, INLDS = false).
  // Blocks that are coupled to nothing are factored / solved all at once; the others go through a right-looking
  // factorisation by block columns (one body = one panel, 2 barriers each) and panel-wise triangular solves.
  UR5_FN int pdof(int i) const {   // engine dof -> permuted row
    if (i < M.nrd) return 6 * M.nobj + i;
    const int k = (i - M.nrd) / 6, j = (i - M.nrd) % 6;
    return 6 * S.obj_rank[k] + j;
  }
  UR5_FN int edof(int I) const {   // permuted row -> engine dof
    if (I >= 6 * M.nobj) return I - 6 * M.nobj;
    return M.nrd + 6 * S.obj_at[I / 6] + I % 6;
  }
  UR5_FN int blk_of_body(int b) const { return b < M.nrd ? M.nobj : S.obj_rank[b - M.nrd]; }
  UR5_FN int blk_width(int p) const { return p < M.nobj ? 6 : M.nrd; }
  UR5_FN real* panel_row(int i) { return S.hess + UR5_SCR_PANEL + UR5_MAXRD * i; }   // current block column of the factorisation, by global row (global scratch)
  // ... or, when the panel is an object's (6 columns) and at most LPANEL_ROWS rows reach it -- nearly always --, in the wavefront's quarter of the LDS area that the
  // staged wrench terms / body twists / kinematic temporaries use at other times (nothing else touches it during a factorisation), indexed by the row's position
  // among the reaching rows: the write -> barrier -> read of every level stays out of global memory
  // UR5_PANEL_SLOT: lanes that work on one panel of a pass. Settled piles have few levels (3.7) but 6-9 NARROW panels in the first (6 own + ~6 reaching rows, 21 row
  // pairs: tools/pile_structure_stats.py, profiles/r04_pile_structure_stats_12piles.log), so with a wavefront per panel (64, rounds 3-4) a pass of four panels used 48 of
  // its 256 lanes and a factorisation took 4.7 passes; with 32-lane slots a pass holds 8 panels (passes ~ levels), a wider panel's slot makes more trips. Same bits.
  // Same-box A/B at 2048 piles with the envelope in LDS (profiles/r05_a_ab_many.log): 64 -> 559.9 k, 32 -> 576.0 k, 16 -> 574.5 k env-steps/s.
#ifndef UR5_PANEL_SLOT
#define UR5_PANEL_SLOT 32
#endif
  static constexpr int PSLOT = UR5_PANEL_SLOT;       // lanes that work on one panel of a pass
  static_assert(PSLOT == 64 || PSLOT == 32 || PSLOT == 16, "a panel slot is a whole wavefront or an aligned part of one");
  static constexpr int LPANEL_ROWS = 2 * UR5_MAXCON * 6 / (UR5_NT / PSLOT) / 6;
  UR5_FN static bool panel_in_lds(int w, int nr) { return w == 6 && nr <= LPANEL_ROWS; }
  template <bool INLDS> UR5_FN double* hptr(int I, int J) { return (INLDS ? S.henv : S.hess) + S.env_ptr[I] + (J - S.env_first[I]); }
  // ... and with the envelope itself in LDS (INLDS, round 5) there is no panel at all: a reaching row's finished entries of block column c0 go straight to their place
  // H(i, c0..) -- during A1 only that row's own lane reads or writes them (the block's rows, which every lane reads, are not written) -- and the trailing update reads
  // them there. (The staging area is part of the envelope's pool in that mode.)
#ifdef UR5_EMUL
  template <bool INLDS> UR5_FN real* panel_at(int i, int c, int c0, bool inl, int) { if constexpr (INLDS) return (real*)hptr<true>(i, c0); else return inl ? &S.stw[0][0] + 6 * c : panel_row(i); }
#else
  template <bool INLDS> UR5_FN real* panel_at(int i, int c, int c0, bool inl, int owner) {   // owner: the wavefront whose quarter holds the panel
    if constexpr (INLDS) return (real*)hptr<true>(i, c0); else return inl ? &S.stw[0][0] + owner * (LPANEL_ROWS * 6) + 6 * c : panel_row(i);
  }
#endif
  UR5_BIG void envelope_structure() {
    static_assert(UR5_NT >= UR5_MAXNV, "one thread per Hessian row");
    const int nobj = M.nobj, nblk = nobj + 1, nv = M.nv;
    // islands of the coupling graph by label propagation (label = largest member; the robot counts as member nobj, so its
    // island sorts last and the robot block stays the last block). A fixed number of rounds: if an island is not fully
    // labelled the ordering is merely less compact -- the envelope below is computed from whatever order results.
    PAR(p2, nblk) S.island[p2] = p2;
    SYNC();
#ifdef UR5_EMUL
    for (int round = 0; round < 6; round++) {
      PAR(q, S.ncouple) {
        const int c = S.couple[q];
        const int ia = S.cA[c] < M.nrd ? nobj : S.cA[c] - M.nrd, ib = S.cB[c] < M.nrd ? nobj : S.cB[c] - M.nrd;
        const int la = S.island[ia], lb = S.island[ib];
        if (la < lb) UR5_ATOMIC_MAX(&S.island[ia], lb); else if (lb < la) UR5_ATOMIC_MAX(&S.island[ib], la);
      }
      SYNC();
    }
#else
    // Every round reads the labels of the previous round only (the new ones collect in blk_first, which is initialised further down): the maximum is exact
    // and order-free, so the labels after each round -- converged or not -- are the same whatever the wavefronts' timing. A label is itself a block of the
    // island, so label[label[.]] (pointer jumping) is a member too and doubles the reach of a round.
    for (int round = 0; round < 6; round++) {
      PAR(p2, nblk) S.blk_first[p2] = S.island[p2];
      SYNC();
      PAR(q, S.ncouple) {
        const int c = S.couple[q];
        const int ia = S.cA[c] < M.nrd ? nobj : S.cA[c] - M.nrd, ib = S.cB[c] < M.nrd ? nobj : S.cB[c] - M.nrd;
        const int la = S.island[ia], lb = S.island[ib];
        if (la < lb) UR5_ATOMIC_MAX(&S.blk_first[ia], lb); else if (lb < la) UR5_ATOMIC_MAX(&S.blk_first[ib], la);
      }
      SYNC();
      PAR(p2, nblk) S.island[p2] = S.blk_first[S.blk_first[p2]];
      SYNC();
    }
#endif
    PAR(k, nobj) {   // order: island, then x
      const real key = S.bpos[M.nrd + k][0];
      const int lab = S.island[k];
      int r = 0;
      for (int j = 0; j < nobj; j++) {
        const real kj = S.bpos[M.nrd + j][0];
        const int lj = S.island[j];
        if (lj < lab || (lj == lab && (kj < key || (kj == key && j < k)))) r++;
      }
      S.obj_rank[k] = (short)r; S.obj_at[r] = (short)k;
    }
    // Everything below used to be four loops on lane 0 (over the coupled contacts, the 248 rows, the blocks, and levels x blocks with a scratch-memory
    // array): 130 k cycles per step with the other 255 lanes waiting. Same lists, built by all lanes (round 3).
    PAR(p2, nblk) S.blk_first[p2] = p2;
    SYNC();
    PAR(q, S.ncouple) {   // first block every block is coupled with: a minimum over the coupled contacts
      const int c = S.couple[q];
      int pa = blk_of_body(S.cA[c]), pb = blk_of_body(S.cB[c]);
      if (pa > pb) { int t = pa; pa = pb; pb = t; }
#ifdef UR5_EMUL
      if (S.blk_first[pb] > pa) S.blk_first[pb] = pa;
#else
      __hip_atomic_fetch_min(&S.blk_first[pb], pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
    }
    SYNC();
    PAR(p2, nblk) {   // blocks below p2 whose rows reach it (their envelope starts at or before it)
      int last = p2, cnt = 0;
      for (int q = p2 + 1; q < nblk; q++) if (S.blk_first[q] <= p2) { last = q; cnt++; }
      S.blk_last[p2] = (short)last; S.reach_cnt[p2] = (short)cnt;
    }
    PAR(i, nv) S.env_first[i] = (unsigned char)(6 * S.blk_first[i < 6 * nobj ? i / 6 : nobj]);
    SYNC();
    // envelope pointers: row i of block p starts at (stored entries of the blocks before p) + (stored entries of the block's rows before i);
    // a row of block p at local index j stores 6 p + j - 6 blk_first[p] + 1 entries
    PAR(p2, nblk) {
      int o = 0;
      for (int q = 0; q < p2; q++) { const int w = 6, f = 6 * (q - S.blk_first[q]); o += w * f + w * (w + 1) / 2; }   // sum_{j<w} (f + j + 1); only objects precede a block
      S.blk_ptr[p2] = o;
    }
    // level of a panel = its position in its envelope group's chain (groups: maximal block ranges that no row crosses; they share no Hessian entry, so
    // their panels are independent): a scan over the blocks, 41 steps on one lane, results in LDS
    if (UR5_LANE == 0) {
      int grp_end = -1, pos = 0, nl = 0, ns = 0;
      for (int p2 = 0; p2 < nblk; p2++) {
        const int bl = S.blk_last[p2];
        if (p2 > grp_end) pos = 0;
        if (bl > grp_end) grp_end = bl;
        if (bl != p2) { S.lv[p2] = (short)pos; pos++; if (pos > nl) nl = pos; S.seq[ns++] = (short)p2; } else S.lv[p2] = -1;   // seq: panels of the sequential sweep
      }
      S.nlvl = nl; S.nseq = ns;
    }
    SYNC();
    PAR(i, nv) {
      const int p2 = i < 6 * nobj ? i / 6 : nobj, j = i - 6 * p2, f = 6 * (p2 - S.blk_first[p2]);
      S.env_ptr[i] = (unsigned short)(S.blk_ptr[p2] + j * f + j * (j + 1) / 2);
      if (i == nv - 1) {
        const int tot = S.env_ptr[i] + f + j + 1;
        S.env_ptr[nv] = (unsigned short)tot;
        S.env_inlds = tot <= UR5_HENV_CAP && tot <= L::HENV_DOUBLES && !UR5_FORCE_GLOBAL_ENV;
        S.dc_inlds = S.env_inlds && tot + 44 * DC_POOL <= L::HENV_DOUBLES && UR5_DCACHE_LDS;
      }
    }
    PAR(l, S.nlvl + 1) {   // panels with a lower level come first: lvl_ptr[l] = their number
      int o = 0;
      for (int p2 = 0; p2 < nblk; p2++) { const int v = S.lv[p2]; if (v >= 0 && v < l) o++; }
      S.lvl_ptr[l] = (short)o;
    }
    PAR(p2, nblk) {       // exclusive prefix of the reach counts, clamped to the list's capacity: when the lists overflow (flagged below) every later READ
      int o = 0;          // reach_list[reach_ptr[p] + k] of the factorisation / the solves stays inside the array -- the flagged scene's numbers are wrong, its
      for (int q = 0; q < p2; q++) o += S.reach_cnt[q];   // accesses are not (round-4 advice: the unclamped reads ran into the neighbouring LDS arrays)
      const int e = o + S.reach_cnt[p2];
      if (p2 == nblk - 1 && e > L::REACH_CAP) S.status |= UR5_ST_ROW_OVERFLOW;
      S.reach_ptr[p2] = (short)(o < L::REACH_CAP ? o : L::REACH_CAP);
      if (p2 == nblk - 1) S.reach_ptr[nblk] = (short)(e < L::REACH_CAP ? e : L::REACH_CAP);
    }
    SYNC();
    PAR(p2, nblk) {       // a panel's slot inside its level: panels of the same level in block order
      const int v = S.lv[p2];
      if (v >= 0) {
        int o = S.lvl_ptr[v];
        for (int q = 0; q < p2; q++) if (S.lv[q] == v) o++;
        S.lvl_list[o] = (short)p2;
      }
    }
    // (an overflow was flagged above: the lists below are cut off at the capacity, the pointers with them -- flagged, never silent, never out of bounds)
    PAR(p2, nblk) { int o = S.reach_ptr[p2]; for (int q = p2 + 1; q <= S.blk_last[p2]; q++) if (S.blk_first[q] <= p2) { if (o < L::REACH_CAP) S.reach_list[o] = (short)q; o++; } }
#ifndef UR5_EMUL
    // the pair of Hessian blocks every coupled contact adds to (envelope_assemble: a block pair is owned by ONE wavefront)
    {
      // ... and the coupled contacts grouped by owner wavefront (owner = a hash of the block pair), contact order inside a group: one coupled contact per lane,
      // a ballot per owner ranks the lanes, the wavefronts' counts give the offsets (as for the side lists)
      static_assert(UR5_MAXCON <= UR5_NT, "one coupled contact per lane");
      constexpr int NW = UR5_NT / 64;
      const int q = UR5_LANE, wv = UR5_LANE >> 6, wl = UR5_LANE & 63;
      int own = -1, ky = -1;
      unsigned long long rec = 0;
      if (q < S.ncouple) {
        const int c = S.couple[q], A = S.cA[c], B = S.cB[c];
        const int pa = blk_of_body(A), pb = blk_of_body(B);
        ky = pa < pb ? pa * 64 + pb : pb * 64 + pa;
        own = (pa + pb + (pa < pb ? pa : pb)) & (NW - 1);
        rec = (unsigned long long)c | (unsigned long long)A << 8 | (unsigned long long)B << 16 | (unsigned long long)ky << 24 | (unsigned long long)(pa >= pb ? 1 : 0) << 40;
      }
#ifndef UR5_HASHED_PAIR_OWNERS
      // Round 5: block pairs are dealt to the wavefronts by LOAD, not by a hash. A wavefront walks its contacts one after the other in every Hessian assembly of the
      // step (~10), so the busiest wavefront sets the time of the coupling loop; with ~13 coupled contacts hashed into four lists that was 5-6 against a mean of 3.
      // Greedy, in contact order: the first contact of a pair (its leader) takes the least loaded wavefront for the whole pair (ties: the lowest), weighted with the
      // pair's contact count. A function of the contact list alone -- no timing enters --, and an entry of H still gets its terms from ONE wavefront in contact order.
      {
        static_assert(4 * UR5_MAXCON <= UR5_MAXCAND, "scratch in the (dead) broad-phase candidate list");
        short* const keys = S.cand, * const lead = S.cand + UR5_MAXCON, * const cntl = S.cand + 2 * UR5_MAXCON, * const ownl = S.cand + 3 * UR5_MAXCON;
        if (q < S.ncouple) keys[q] = (short)ky;
        SYNC();
        if (q < S.ncouple) {
          int ld = q, cnt = 0;
          for (int o = S.ncouple - 1; o >= 0; o--) if (keys[o] == ky) { ld = o; cnt++; }
          lead[q] = (short)ld; cntl[q] = (short)(ld == q ? cnt : 0);
        }
        SYNC();
        if (UR5_LANE == 0) {
          int load[NW];
          for (int w = 0; w < NW; w++) load[w] = 0;
          for (int o = 0; o < S.ncouple; o++) {
            const int cnt = cntl[o];
            if (cnt == 0) continue;
            int best = 0;
            for (int w = 1; w < NW; w++) if (load[w] < load[best]) best = w;
            ownl[o] = (short)best; load[best] += cnt;
          }
        }
        SYNC();
        if (q < S.ncouple) own = ownl[lead[q]];
      }
#endif
      int rank = 0;
#pragma unroll
      for (int w = 0; w < NW; w++) {
        const unsigned long long m = __ballot(own == w);
        if (own == w) rank = __popcll(m & ((1ull << wl) - 1ull));
        if (wl == 0) S.slot_cnt[wv][w] = (unsigned char)__popcll(m);
      }
      SYNC();
      if (own >= 0) {
        int o = rank;
        for (int w = 0; w < NW; w++) for (int v = 0; v < NW; v++) if (w < own || (w == own && v < wv)) o += S.slot_cnt[v][w];
        S.wrec[o] = rec;
      }
      if (UR5_LANE <= NW) { int o = 0; for (int w = 0; w < UR5_LANE; w++) for (int v = 0; v < NW; v++) o += S.slot_cnt[v][w]; S.wptr[UR5_LANE] = (short)o; }
    }
#endif
    SYNC();
  }
  // row number c (0 .. nr-1) among the rows below panel p2 that reach it; nrb = number of reaching blocks (only the last can be the 8-wide robot)
  UR5_FN int reach_row(int p2, int nrb, int c) const {
    int bl = c / 6;
    bl = bl < nrb - 1 ? bl : nrb - 1;
    return 6 * S.reach_list[S.reach_ptr[p2] + bl] + (c - 6 * bl);
  }
  UR5_FN int reach_rows(int p2, int nrb) const { return nrb == 0 ? 0 : 6 * (nrb - 1) + blk_width(S.reach_list[S.reach_ptr[p2] + nrb - 1]); }
  UR5_FN bool blk_single(int p2) const { return S.blk_first[p2] == p2 && S.blk_last[p2] == p2; }
  // reaches earlier blocks, reached by none: factored / solved after (forward) resp. before (backward) the sequential sweep,
  // all such blocks at once -- their column ranges cannot overlap
  UR5_FN bool blk_terminal(int p2) const { return S.blk_first[p2] != p2 && S.blk_last[p2] == p2; }
  template <bool INLDS> UR5_BIG void envelope_assemble() {
    const int tot = S.env_ptr[M.nv];
    PROF_T0();
    double* const hb = INLDS ? S.henv : S.hess;
    // Round 5: only the part of a row LEFT of its diagonal block is zeroed (that is where coupling blocks and fill land); the diagonal blocks are written, never added
    // to, by the lanes below -- disjoint entries, so the barrier that used to separate the zeroing from them is gone. When the robot carries contacts its block is
    // ~3 k cycles of work on wavefront 0: the object blocks then go to the other three wavefronts instead of queueing behind it.
    (void)tot;
    PAR(i, M.nv) {
      const int p2 = i < 6 * M.nobj ? i / 6 : M.nobj;
      const int n = 6 * p2 - S.env_first[i];
      double* r = hb + S.env_ptr[i];
      for (int k = 0; k < n; k++) r[k] = 0;
    }
    PAR(idx, M.nrd * M.nrd) {   // robot block: Mr + sum_b cdof^T G_b cdof + equality / limit rows
      int d = idx / M.nrd, e = idx % M.nrd;
      if (e > d) continue;
      real v = S.Mr[d][e];
      unsigned common = M.rd_desc[d] & M.rd_desc[e];
      for (int rg = 0; rg < M.nrg; rg++) {
        const int b = M.rg_body[rg];
        if (!(common >> b & 1u) || !(S.bodymask >> b & 1u)) continue;
        for (int i = 0; i < 6; i++) {
          real t = 0;
          for (int j = 0; j < 6; j++) t += S.G[rg][sym6(i, j)] * S.cdof[e][j];
          v += S.cdof[d][i] * t;
        }
      }
      for (int s2 = 0; s2 < S.nsr; s2++) {
        if (S.sr_uni[s2] && S.sr_jar[s2] >= 0) continue;
        real cd = (S.sr_d1[s2] == d ? S.sr_c1[s2] : (real)0) + (S.sr_d2[s2] == d ? S.sr_c2[s2] : (real)0);
        real ce = (S.sr_d1[s2] == e ? S.sr_c1[s2] : (real)0) + (S.sr_d2[s2] == e ? S.sr_c2[s2] : (real)0);
        v += S.sr_D[s2] * cd * ce;
      }
      *hptr<INLDS>(pdof(d), pdof(e)) = (double)v;
    }
#if !defined(UR5_EMUL) && UR5_NT > 64
    const bool robot_busy = (S.bodymask & ((1ull << M.nrd) - 1ull)) != 0;    // scene-uniform
    const int obj_l0 = robot_busy ? 64 : 0;
    for (int idx = UR5_LANE - obj_l0; idx >= 0 && idx < M.nobj * 21; idx += GS - obj_l0) {     // object blocks: M + T^T G T
#else
    PAR(idx, M.nobj * 21) {     // object blocks: M + T^T G T
#endif
      int k = idx / 21, ent = idx % 21, b = M.nrd + k;
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= ent) i++;
      int j = ent - i * (i + 1) / 2;
      int di = M.nrd + 6 * k + i, dj = M.nrd + 6 * k + j;
      real v = 0;
      if (S.bodymask >> b & 1ull) {
        // T^T G T with the object's unit twists written out (unit_twist: dof i < 3 = [0; e_i], dof i >= 3 = [R col(i-3); 0]): the linear block is a copy of G's,
        // the mixed block one contraction with R, the angular block two -- the terms (and their order) that the generic 6 x 6 x 6 product leaves non-zero
        const real* Gs = S.G[M.nrg + k];
        if (i < 3) v = Gs[sym6(3 + i, 3 + j)];
        else {
          m3 R; R.load(S.bmat[b]);
          const v3 ci = R.col(i - 3);
          real t[3];
          if (j < 3) { for (int a2 = 0; a2 < 3; a2++) t[a2] = Gs[sym6(a2, 3 + j)]; }
          else { const v3 cj = R.col(j - 3); for (int a2 = 0; a2 < 3; a2++) t[a2] = Gs[sym6(a2, 0)] * cj.x + Gs[sym6(a2, 1)] * cj.y + Gs[sym6(a2, 2)] * cj.z; }
          v = ci.x * t[0] + ci.y * t[1] + ci.z * t[2];
        }
      }
      if (i == j) {
        v += S.Mobj[6 * k + i];
        for (int s2 = 0; s2 < S.nsr; s2++) if (S.sr_d1[s2] == di && !(S.sr_uni[s2] && S.sr_jar[s2] >= 0)) v += S.sr_D[s2] * S.sr_c1[s2] * S.sr_c1[s2];
      }
      *hptr<INLDS>(pdof(di), pdof(dj)) = (double)v;
    }
    SYNC();
    PROF(PF_X6);   // zeroing + diagonal blocks (the rest of H_asm is the coupling loop)
#ifndef UR5_EMUL
    // coupling blocks. Several contacts add to the same block (a box resting on a box: four), and float atomics from four wavefronts would land in a different
    // order from run to run. Instead a block pair belongs to ONE wavefront: it walks ITS coupled contacts (S.wlist, contact order), and its 64 lanes add the contact's (up to 8 x 8) entries with plain read-modify-writes -- lane `ent` always owns the same entry of a block,
    // so a sum's order is the contact order whatever the other wavefronts do.
    {
      static_assert(UR5_MAXRD * UR5_MAXRD <= 64 && UR5_MAXOBJ + 1 <= 64, "one entry of a coupling block per lane of a wavefront");
      const int wv = UR5_LANE >> 6, ent = UR5_LANE & 63;
      for (int k = S.wptr[wv]; k < S.wptr[wv + 1]; k++) {
        const unsigned long long rec = S.wrec[k];   // everything the entry needs to find its data in ONE load (the chain list -> contact -> bodies -> blocks was four)
        const int c = (int)(rec & 255u), A = (int)(rec >> 8 & 255u), B = (int)(rec >> 16 & 255u), ky = (int)(rec >> 24 & 0xffffu);
        const int colblk = ky >> 6, rowblk = ky & 63;
        const int nr = blk_width(rowblk), ncw = blk_width(colblk);
        if (ent >= nr * ncw) continue;
        const int li = ent / ncw, lj = ent - li * ncw;
        const bool both_robot = rowblk == colblk;   // two robot bodies (finger against finger): the robot's own diagonal block, symmetrised
        if (both_robot && li < lj) continue;
        real v;
        if (both_robot) { v = couple_term(c, A, li, B, lj); v = li == lj ? 2 * v : v + couple_term(c, A, lj, B, li); }
        else if (rec >> 40 & 1u) v = couple_term(c, A, li, B, lj);
        else v = couple_term(c, A, lj, B, li);
        if (v != 0) *hptr<INLDS>(6 * rowblk + li, 6 * colblk + lj) += (double)v;
      }
    }
    SYNC();
    return;
#endif
    // lane emulation: every (contact, entry) pair, one after the other
    PAR(idx, S.ncouple * 64) {
      const int c = S.couple[idx >> 6], ent = idx & 63;
      const int A = S.cA[c], B = S.cB[c];
      const int nA = A < M.nrd ? M.nrd : 6, nBd = B < M.nrd ? M.nrd : 6;
      if (ent >= nA * nBd) continue;
      const int ia = ent / nBd, ib = ent % nBd;
      const bool both_robot = A < M.nrd && B < M.nrd;
      if (both_robot && ia < ib) continue;
      real v = couple_term(c, A, ia, B, ib);
      if (both_robot) v = ia == ib ? 2 * v : v + couple_term(c, A, ib, B, ia);
      if (v == 0) continue;
      int I = pdof(A < M.nrd ? ia : M.nrd + 6 * (A - M.nrd) + ia), J = pdof(B < M.nrd ? ib : M.nrd + 6 * (B - M.nrd) + ib);
      if (I < J) { int t = I; I = J; J = t; }
      UR5_ATOMIC_ADD(hptr<INLDS>(I, J), (double)v);
    }
    SYNC();
  }
  // lower Cholesky factor of a diagonal block (order W = 6 for an object, 8 for the robot) in registers
  template <int W> struct Diag { real l[W][W]; real inv[W]; };
  // unfactored block at row/column c0 from H -> factor in registers
  template <bool INLDS, int W> UR5_FN void diag_factor(int c0, Diag<W>& d) {
    double* const hb = INLDS ? S.henv : S.hess;
    const int off = c0 - S.env_first[c0];
#pragma unroll
    for (int a = 0; a < W; a++) {
      const double* row = hb + S.env_ptr[c0 + a] + off;
#pragma unroll
      for (int bb = 0; bb < W; bb++) d.l[a][bb] = bb <= a ? (real)row[bb] : (real)0;
    }
#pragma unroll
    for (int j = 0; j < W; j++) {
      real dj = d.l[j][j];
#pragma unroll
      for (int k = 0; k < j; k++) dj -= d.l[j][k] * d.l[j][k];
      dj = dj < (real)1e-15 ? (real)1e-15 : dj;
#ifdef UR5_EMUL
      const real sq = sqrt(dj), inv = (real)1 / sq;
#else
      real inv = rsqrt(dj);                                          // this chain is the critical path of a panel:
      inv = inv * ((real)1.5 - (real)0.5 * dj * inv * inv);          // rsqrt + one Newton step instead of sqrt and a division
      const real sq = dj * inv;
#endif
      d.l[j][j] = sq; d.inv[j] = inv;
#pragma unroll
      for (int a = j + 1; a < W; a++) {
        real sacc = d.l[a][j];
#pragma unroll
        for (int k = 0; k < j; k++) sacc -= d.l[a][k] * d.l[j][k];
        d.l[a][j] = sacc * inv;
      }
    }
  }
  // factored blocks are kept packed (lower triangle, then 1 / diagonal) in S.dcache for the triangular solves
  // Round 5: with the envelope in the LDS pool (env_inlds) the block cache is in LDS too whenever it fits (S.dc_inlds; it does for every sampled envelope below
  // 1 856 doubles): the first DC_WBG blocks in the body accumulators WB | G -- dead from the end of the Hessian assembly to the next gather, which rewrites every
  // entry --, the others behind the envelope in the pool. A level of the factorisation / the sweeps then touches no global memory at all. (Round 4 measured an LDS
  // COPY of a global cache as slower: that was one more round trip per solve; this is the cache itself.)
  static constexpr int DC_WBG = (int)((sizeof(L::WB) + sizeof(L::G)) / sizeof(real)) / 44;
  static_assert(offsetof(L, G) == offsetof(L, WB) + sizeof(L::WB), "WB | G are one stretch of LDS");
  static constexpr int DC_POOL = UR5_MAXOBJ + 1 > DC_WBG ? UR5_MAXOBJ + 1 - DC_WBG : 0;   // blocks cached behind the envelope
  UR5_FN real* dc_lds(int p2) { return p2 < DC_WBG ? &S.WB[0][0] + 44 * p2 : (real*)S.henv + S.env_ptr[M.nv] + 44 * (p2 - DC_WBG); }
  template <int W> UR5_FN static void diag_put(real* c, const Diag<W>& d) {
#pragma unroll
    for (int a = 0; a < W; a++) {
#pragma unroll
      for (int bb = 0; bb <= a; bb++) c[a * (a + 1) / 2 + bb] = d.l[a][bb];
      c[36 + a] = d.inv[a];
    }
  }
  template <int W> UR5_FN static void diag_get(const real* c, Diag<W>& d) {
#pragma unroll
    for (int a = 0; a < W; a++) {
#pragma unroll
      for (int bb = 0; bb < W; bb++) d.l[a][bb] = bb <= a ? c[a * (a + 1) / 2 + bb] : (real)0;
      d.inv[a] = c[36 + a];
    }
  }
  template <int W> UR5_FN void diag_store(int p2, const Diag<W>& d) {
    if (S.dc_inlds) diag_put<W>(dc_lds(p2), d); else diag_put<W>(S.hess + UR5_SCR_DCACHE + 44 * p2, d);   // (two code paths: an LDS and a global pointer must not meet in one select)
  }
  template <int W> UR5_FN void diag_cached(int p2, Diag<W>& d) {
    if (S.dc_inlds) diag_get<W>(dc_lds(p2), d); else diag_get<W>(S.hess + UR5_SCR_DCACHE + 44 * p2, d);
  }
  // (Round 4 also tried an LDS copy of the factored diagonal blocks for the duration of every triangular solve -- 9 KB packed, one cooperative copy per solve into the
  // staging area: the solves got 8 % SLOWER (one more memory round trip and two barriers per solve; the per-level fetches hit the vector L1 anyway),
  // profiles/r04_i_many_phase_cycles_512piles.log. The level loops are bound by their dependent arithmetic and barriers, not by where the blocks live.)
  template <int W> UR5_FN static real pick(const real (&v)[W], int k) {   // v[k] without a run-time register index
    real o = v[0];
#pragma unroll
    for (int a = 1; a < W; a++) o = a == k ? v[a] : o;
    return o;
  }
  // x <- L^-1 x and x <- L^-T x for one block
  template <int W> UR5_FN static void fwd_blk(const Diag<W>& d, real (&x)[W]) {
#pragma unroll
    for (int k = 0; k < W; k++) {
      real sacc = x[k];
#pragma unroll
      for (int m = 0; m < k; m++) sacc -= d.l[k][m] * x[m];
      x[k] = sacc * d.inv[k];
    }
  }
  template <int W> UR5_FN static void bwd_blk(const Diag<W>& d, real (&x)[W]) {
#pragma unroll
    for (int k = W - 1; k >= 0; k--) {
      real sacc = x[k];
#pragma unroll
      for (int m = k + 1; m < W; m++) sacc -= d.l[m][k] * x[m];
      x[k] = sacc * d.inv[k];
    }
  }
  // uncoupled blocks: every row factors its own diagonal block (redundantly per row); row 0 of the block files it in dcache
  // kind 0: the head of a chain (its factor goes to the block cache for the panel rows below it); kind 1: an uncoupled block; kind 2: a terminal block.
  // Round 5: kinds 1 and 2 finish their part of the solve that follows every factorisation right here -- the lane has the block's factor in registers and the block's
  // right-hand side is final (uncoupled: always; terminal: after the last level), so x_blk = L^-T L^-1 b_blk costs it two 6 x 6 substitutions instead of a store to the
  // block cache, a barrier and a reload in envelope_solve. Same operands, same order as solve_single_row / terminal_fwd_bwd: same bits. (An envelope in global memory
  // may be reused by a later iteration, so that path still files the factor.)
  template <bool INLDS, int W> UR5_FN void factor_single_row(int i, int p2, int kind, const real* b, real* y) {
    const int c0 = 6 * p2, r = i - c0;
    Diag<W> d;
    diag_factor<INLDS, W>(c0, d);
    if (r == 0 && (kind == 0 || !INLDS)) diag_store<W>(p2, d);
    if (kind == 0) return;
    real t[W];
#pragma unroll
    for (int k = 0; k < W; k++) t[k] = b[c0 + k];
    fwd_blk<W>(d, t);
    bwd_blk<W>(d, t);
    const real xi = pick<W>(t, r);
    S.search[edof(i)] = -xi;
    if (kind == 2) y[i] = xi;   // x of the block, for the update of the columns to its left
  }
  // The forward substitution of the solve that follows every factorisation rides along (b, y as in envelope_solve): the lane that has just computed row i's entries
  // of block column p2 has the block's factor in registers, so y_blk = L_pp^-1 b_blk and b_i -= L_i,blk y_blk cost it a handful of multiply-adds instead of
  // a second sweep over the levels (one barrier + the reload of every diagonal block and row per level). Same operands, same order as fwd_panel_row: same bits.
  template <bool INLDS, int W> UR5_FN void factor_panel_row(int i, int ii, int p2, int c0, bool inl, int owner, real* b, real* y) {
    Diag<W> d;
    const bool prefactored = S.blk_first[p2] == p2;
    if (prefactored) diag_cached<W>(p2, d); else diag_factor<INLDS, W>(c0, d);
    real yb[W];
#pragma unroll
    for (int k = 0; k < W; k++) yb[k] = b[c0 + k];
    fwd_blk<W>(d, yb);
    real out[W];
    if (ii < W) {
#pragma unroll
      for (int k = 0; k < W; k++) {
        real v = 0;
#pragma unroll
        for (int a = 0; a < W; a++) if (a == ii && k <= a) v = d.l[a][k];
        out[k] = v;
      }
      if (ii == 0 && !prefactored) diag_store<W>(p2, d);
      y[i] = pick<W>(yb, ii);
    } else {
      const double* row = hptr<INLDS>(i, c0);
#pragma unroll
      for (int k = 0; k < W; k++) {
        real sacc = (real)row[k];
#pragma unroll
        for (int m = 0; m < k; m++) sacc -= out[m] * d.l[k][m];
        out[k] = sacc * d.inv[k];
      }
      real sacc = 0;
#pragma unroll
      for (int k = 0; k < W; k++) sacc += (real)(double)out[k] * yb[k];
      b[i] -= sacc;
    }
    if (ii >= W) {   // (the block's own rows live on in dcache; only the rows below it are read back by the trailing update)
      real* o = panel_at<INLDS>(i, ii - W, c0, inl, owner);
#pragma unroll
      for (int k = 0; k < W; k++) o[k] = out[k];
    }
  }
  // Work split inside a level: on the GPU wavefront w of the workgroup owns panel base + w of the pass and its 64 lanes stride over
  // that panel's rows; the lane-emulation build walks the panels of a pass one after the other.
#ifdef UR5_EMUL
#define UR5_PANELS_PER_PASS 1
#define UR5_FOR_MY_PANELS(j, base, np) for (int j = (base); j < (base) + 1 && j < (np); j++)
#define UR5_PLANE(t, n) for (int t = 0; t < (n); t++)
#define UR5_PLANE_SHARED(t, n, sh) for (int t = 0; t < (n); t++)
#else
#define UR5_PANELS_PER_PASS (UR5_NT / PSLOT)
#define UR5_FOR_MY_PANELS(j, base, np) for (int j = (base) + UR5_LANE / PSLOT, once_ = 1; once_ && j < (np); once_ = 0)
#define UR5_PLANE(t, n) for (int t = UR5_LANE % PSLOT; t < (n); t += PSLOT)
#define UR5_PLANE_SHARED(t, n, sh) for (int t = UR5_LANE % PSLOT + PSLOT * (sh).part; t < (n); t += PSLOT * (sh).parts)
#endif
  // A pass of the factorisation holds cnt <= 4 panels. The profile (profiles/r04_ae_*) says a pile's levels are few (about four passes per factorisation) and WIDE (a
  // panel's rows and row pairs take several trips of one wavefront): with fewer panels than wavefronts the spare wavefronts take a share of a panel's rows / pairs.
  // Wavefront w works on panel q = w mod cnt of the pass, as part w / cnt of `parts`; the panel's rows sit in the LDS quarter of wavefront q. Every row / entry is
  // still computed by exactly one lane from the same operands: same bits.
  struct Share { int q, part, parts; };
  UR5_FN static int own_wave() {
#ifdef UR5_EMUL
    return 0;
#else
    return UR5_LANE / PSLOT;
#endif
  }
  UR5_FN static Share pass_share(int cnt) {
    Share sh;
#if defined(UR5_EMUL) || defined(UR5_NO_PANEL_SHARING)   // (build option: every wavefront keeps to its own panel, as before the sharing)
    sh.q = own_wave(); sh.part = 0; sh.parts = 1; (void)cnt;
#else
    const int w = own_wave();
    sh.q = w % cnt; sh.part = w / cnt; sh.parts = (UR5_PANELS_PER_PASS - 1 - sh.q) / cnt + 1;
#endif
    return sh;
  }
  // the panel (index j in the level's list) this wavefront works on in the pass starting at `base`, and its share of it
#if defined(UR5_EMUL) || defined(UR5_NO_PANEL_SHARING)
#define UR5_FOR_MY_SHARE(sh, j, base, np) const Share sh = pass_share(1); UR5_FOR_MY_PANELS(j, base, np)
#else
#define UR5_FOR_MY_SHARE(sh, j, base, np) const Share sh = pass_share((np) - (base) < UR5_PANELS_PER_PASS ? (np) - (base) : UR5_PANELS_PER_PASS); for (int j = (base) + sh.q, once_ = 1; once_; once_ = 0)
#endif
  // A1: the panel's own rows and every reaching row compute their entries of the block column (LDS panel); H is only read
  template <bool INLDS> UR5_FN void panel_factor_rows(int p2, const Share& sh) {
    const int c0 = 6 * p2, w = blk_width(p2);
    const int nrb = S.reach_ptr[p2 + 1] - S.reach_ptr[p2], nr = reach_rows(p2, nrb);
    const bool inl = panel_in_lds(w, nr);
    UR5_PLANE_SHARED(t, w + nr, sh) {
      const int i = t < w ? c0 + t : reach_row(p2, nrb, t - w);
      if (p2 < M.nobj) factor_panel_row<INLDS, 6>(i, t, p2, c0, inl, sh.q, S.Mv_(), S.tmpv); else factor_panel_row<INLDS, UR5_MAXRD>(i, t, p2, c0, inl, sh.q, S.Mv_(), S.tmpv);
    }
  }
  // A2: the finished column entries of the rows below go back to H (the block itself lives on in dcache);
  // B: trailing update of every pair of reaching rows below the block
  template <bool INLDS> UR5_FN void panel_trailing_update(int p2, const Share& sh) {
    const int c0 = 6 * p2, w = blk_width(p2);
    const int nrb = S.reach_ptr[p2 + 1] - S.reach_ptr[p2], nr = reach_rows(p2, nrb);
    const bool inl = panel_in_lds(w, nr);
    if constexpr (!INLDS)
    UR5_PLANE_SHARED(c, nr, sh) {
      const int i = reach_row(p2, nrb, c);
      double* row = hptr<INLDS>(i, c0);
      const real* pr = panel_at<INLDS>(i, c, c0, inl, sh.q);
      for (int k = 0; k < w; k++) row[k] = (double)pr[k];
    }
    // the pairs (ii >= jj) of the lower triangle, folded into a rectangle so that no lane draws an empty (jj > ii) slot: row r of the rectangle holds row r of the
    // triangle followed by row n - 1 - r (n = nr rounded up to even; the padding row is skipped) -- half the trips of an nr x nr sweep
    const int nre = nr + (nr & 1), wid = nre + 1;
    UR5_PLANE_SHARED(idx, (nre >> 1) * wid, sh) {
      const int r = idx / wid, c = idx - r * wid;
      const int ii = c <= r ? r : nre - 1 - r, jj = c <= r ? c : c - r - 1;
      if (ii >= nr) continue;
      const int i = reach_row(p2, nrb, ii), j = reach_row(p2, nrb, jj);
      real sacc = 0;
      { const real* pi = panel_at<INLDS>(i, ii, c0, inl, sh.q); const real* pj = panel_at<INLDS>(j, jj, c0, inl, sh.q); for (int k = 0; k < w; k++) sacc += pi[k] * pj[k]; }
      *hptr<INLDS>(i, j) -= (double)sacc;
    }
  }
  template <bool INLDS> UR5_BIG void envelope_factor() {
    PROFL_T0();
    // a block that reaches no earlier block gets no trailing update: its diagonal block is final after assembly, so all
    // of these (the uncoupled blocks among them) are factored at once, ahead of the sequential sweep
    PAR(i, M.nv) {
      const int p2 = i < 6 * M.nobj ? i / 6 : M.nobj;
      if (S.blk_first[p2] != p2) continue;
      const int kind = S.blk_last[p2] == p2 ? 1 : 0;
      if (p2 < M.nobj) factor_single_row<INLDS, 6>(i, p2, kind, S.Mv_(), S.tmpv); else factor_single_row<INLDS, UR5_MAXRD>(i, p2, kind, S.Mv_(), S.tmpv);
    }
    SYNC();
    PROFL(PF_X1);
    // level by level; inside a level every wavefront takes one panel (lanes = rows / row pairs of that panel)
    for (int l = 0; l < S.nlvl; l++) {
      const int lp0 = S.lvl_ptr[l], np = S.lvl_ptr[l + 1] - lp0;
      for (int base = 0; base < np; base += UR5_PANELS_PER_PASS) {
#if defined(UR5_EMUL) || defined(UR5_NO_PANEL_SHARING)
        const Share sh = pass_share(1);
        UR5_FOR_MY_PANELS(j, base, np) panel_factor_rows<INLDS>(S.lvl_list[lp0 + j], sh);
        SYNC();
        PROFL(PF_X2);
        UR5_FOR_MY_PANELS(j, base, np) panel_trailing_update<INLDS>(S.lvl_list[lp0 + j], sh);
        SYNC();
#else
        const Share sh = pass_share(np - base < UR5_PANELS_PER_PASS ? np - base : UR5_PANELS_PER_PASS);
        const int p2s = S.lvl_list[lp0 + base + sh.q];
        panel_factor_rows<INLDS>(p2s, sh);
        SYNC();
        PROFL(PF_X2);
        panel_trailing_update<INLDS>(p2s, sh);
        SYNC();
#endif
        PROFL(PF_X3);
#if defined(UR5_PROFILE_LEVELS) && defined(UR5_PROFILE) && !defined(UR5_EMUL)
        if (UR5_LANE == 0) S.prof[PF_X7] += 1.0;   // level passes (count)
#endif
      }
    }
    PAR(i, M.nv) {   // terminal blocks: every trailing update has landed, and so has every forward substitution into their right-hand side
      const int p2 = i < 6 * M.nobj ? i / 6 : M.nobj;
      if (!blk_terminal(p2)) continue;
      if (p2 < M.nobj) factor_single_row<INLDS, 6>(i, p2, 2, S.Mv_(), S.tmpv); else factor_single_row<INLDS, UR5_MAXRD>(i, p2, 2, S.Mv_(), S.tmpv);
    }
    PROFL(PF_X4);
  }
  // S.search = -H^-1 grad with the factor in place: b = S.Mv_() (permuted right-hand side, consumed), y = S.tmpv
  template <int W> UR5_FN void solve_single_row(int i, int p2, const real* b) {
    const int c0 = 6 * p2;
    Diag<W> d;
    diag_cached<W>(p2, d);
    real t[W];
#pragma unroll
    for (int k = 0; k < W; k++) t[k] = b[c0 + k];
    fwd_blk<W>(d, t);
    bwd_blk<W>(d, t);
    S.search[edof(i)] = -pick<W>(t, i - c0);
  }
  template <int W> UR5_FN void terminal_fwd_bwd(int i, int p2, const real* b, real* y) {
    const int c0 = 6 * p2;
    Diag<W> d;
    diag_cached<W>(p2, d);
    real t[W];
#pragma unroll
    for (int k = 0; k < W; k++) t[k] = b[c0 + k];
    fwd_blk<W>(d, t);
    bwd_blk<W>(d, t);
    const real xi = pick<W>(t, i - c0);
    S.search[edof(i)] = -xi;
    y[i] = xi;   // x of the block, for the update of the columns to its left
  }
  template <bool INLDS, int W> UR5_FN void fwd_panel_row(int i, int ii, int p2, int c0, real* b, real* y) {
    Diag<W> d;
    diag_cached<W>(p2, d);
    real yb[W];
#pragma unroll
    for (int k = 0; k < W; k++) yb[k] = b[c0 + k];
    fwd_blk<W>(d, yb);
    if (ii < W) y[i] = pick<W>(yb, ii);
    else {
      const double* row = hptr<INLDS>(i, c0);
      real sacc = 0;
#pragma unroll
      for (int k = 0; k < W; k++) sacc += (real)row[k] * yb[k];
      b[i] -= sacc;
    }
  }
  template <bool INLDS, int W> UR5_FN void bwd_panel_col(int j, int p2, int c0, real* y) {
    Diag<W> d;
    diag_cached<W>(p2, d);
    real xb[W];
#pragma unroll
    for (int k = 0; k < W; k++) xb[k] = y[c0 + k];
    bwd_blk<W>(d, xb);
    if (j >= c0) S.search[edof(j)] = -pick<W>(xb, j - c0);
    else {
      real sacc = 0;
#pragma unroll
      for (int k = 0; k < W; k++) sacc += (real)*hptr<INLDS>(c0 + k, j) * xb[k];
      y[j] -= sacc;   // the lanes of this panel only read y inside the block, never left of it
    }
  }
  template <bool INLDS> UR5_BIG void envelope_solve(bool forward_done) {   // forward_done: the factorisation that has just run did the forward sweep over the levels
    static_assert(sizeof(S.tmpv) / sizeof(real) >= (size_t)NV_, "tmpv holds a dof vector");
    real* b = S.Mv_();
    real* y = S.tmpv;
    PROFL_T0();
    SYNC();   // the terminal blocks' x (left in y by the factorisation); for a reused factor: nothing in flight
    if (!forward_done)
    PAR(i, M.nv) {   // uncoupled blocks: the whole solve at once (a fresh factorisation has done it, factor_single_row)
      const int p2 = i < 6 * M.nobj ? i / 6 : M.nobj;
      if (!blk_single(p2)) continue;
      if (p2 < M.nobj) solve_single_row<6>(i, p2, b); else solve_single_row<UR5_MAXRD>(i, p2, b);
    }
    if (!forward_done)
    for (int l = 0; l < S.nlvl; l++) {   // forward, column-oriented: y_blk = L_pp^-1 b_blk, then b_i -= L_i,blk y_blk for the rows below
      const int lp0 = S.lvl_ptr[l], np = S.lvl_ptr[l + 1] - lp0;
      for (int base = 0; base < np; base += UR5_PANELS_PER_PASS) {
        UR5_FOR_MY_SHARE(sh, j, base, np) {   // (spare wavefronts take a share of a panel's rows, as in the factorisation)
          const int p2 = S.lvl_list[lp0 + j];
          const int c0 = 6 * p2, w = blk_width(p2);
          const int nrb = S.reach_ptr[p2 + 1] - S.reach_ptr[p2], nr = reach_rows(p2, nrb);
          UR5_PLANE_SHARED(t, w + nr, sh) {
            const int i = t < w ? c0 + t : reach_row(p2, nrb, t - w);
            if (p2 < M.nobj) fwd_panel_row<INLDS, 6>(i, t, p2, c0, b, y); else fwd_panel_row<INLDS, UR5_MAXRD>(i, t, p2, c0, b, y);
          }
        }
        SYNC();
      }
    }
    PROFL(PF_X5);
    if (!forward_done) {
      PAR(t, M.nv) {   // terminal blocks: forward and backward substitution inside the block, then their share of y to the left
        const int p2 = t < 6 * M.nobj ? t / 6 : M.nobj;
        if (!blk_terminal(p2)) continue;
        if (p2 < M.nobj) terminal_fwd_bwd<6>(t, p2, b, y); else terminal_fwd_bwd<UR5_MAXRD>(t, p2, b, y);
      }
      SYNC();
    }
    PAR(j, M.nv) {   // y_j -= L_blk,j^T x_blk for the columns j left of a terminal block (x_blk was left in y)
      const int pj = j < 6 * M.nobj ? j / 6 : M.nobj;
      for (int o = S.reach_ptr[pj]; o < S.reach_ptr[pj + 1]; o++) {   // the blocks whose rows reach column block pj
        const int p2 = S.reach_list[o];
        if (!blk_terminal(p2)) continue;
        const int c0 = 6 * p2, w = blk_width(p2);
        real sacc = 0;
        for (int k = 0; k < w; k++) sacc += (real)*hptr<INLDS>(c0 + k, j) * y[c0 + k];
        y[j] -= sacc;
      }
    }
    SYNC();
    PROFL(PF_X4);
    for (int l = S.nlvl - 1; l >= 0; l--) {   // backward, row-oriented: x_blk = L_pp^-T y_blk, then y_j -= L_blk,j^T x_blk for the columns left of it
      const int lp0 = S.lvl_ptr[l], np = S.lvl_ptr[l + 1] - lp0;
      for (int base = 0; base < np; base += UR5_PANELS_PER_PASS) {
        UR5_FOR_MY_SHARE(sh, j, base, np) {
          const int p2 = S.lvl_list[lp0 + j];
          const int c0 = 6 * p2, w = blk_width(p2);
          const int f0 = S.env_first[c0];
          UR5_PLANE_SHARED(jj, c0 + w - f0, sh) {
            const int col = f0 + jj;
            if (p2 < M.nobj) bwd_panel_col<INLDS, 6>(col, p2, c0, y); else bwd_panel_col<INLDS, UR5_MAXRD>(col, p2, c0, y);
          }
        }
        SYNC();
      }
    }
    SYNC();   // S.search of the uncoupled blocks (there may be no sequential block at all)
    PROFL(PF_X0);
  }
#endif

  UR5_CALL void solve_newton_fn() { solve_newton_body(); }
  UR5_FN void solve_newton() { if constexpr (FLAT) solve_newton_body(); else solve_newton_fn(); }
  UR5_PHASE_C void solve_newton_body() {
    const int nv = M.nv;
    if (S.ncon == 0 && S.nsr == 0) {
      PAR(i, nv) S.x[i] = S.as[i];
      SYNC();
      return;
    }
    PROF_T0();
#ifdef UR5_MANY
    if (UR5_LANE == 0) S.act_changed = 1;   // new contacts, new Hessian: the first iteration of a step always factors
#endif
    // warm start: cheaper of qacc_warmstart and qacc_smooth. Both candidates go through the same three passes together: M v and
    // body twists (tw <- warm start, WB <- qacc_smooth; WB is free until the gradient is built), contact / row images, costs.
    PAR(i, nv) {
      S.x[i] = warm()[i];
      if (i < M.nrd) {
        real sw = 0, ss = 0;
        for (int e = 0; e < M.nrd; e++) { sw += S.Mr[i][e] * warm()[e]; ss += S.Mr[i][e] * S.as[e]; }
        S.Ma[i] = sw; S.Mv_()[i] = ss;
      } else { S.Ma[i] = S.Mobj[i - M.nrd] * warm()[i]; S.Mv_()[i] = S.Mobj[i - M.nrd] * S.as[i]; }
    }
    PAR(sl, nslot()) {
      const int b = body_of_slot(sl);
      if (b < M.nrd) {
        real vw[6] = {0, 0, 0, 0, 0, 0}, vs[6] = {0, 0, 0, 0, 0, 0};
        for (int e = 0; e < M.nrd; e++) if (M.rd_anc[b] >> e & 1u) {
          const real qw = warm()[e], qs = S.as[e];
          for (int i = 0; i < 6; i++) { vw[i] += S.cdof[e][i] * qw; vs[i] += S.cdof[e][i] * qs; }
        }
        for (int i = 0; i < 6; i++) { S.tw_()[sl][i] = vw[i]; S.WB[sl][i] = vs[i]; }
      } else {
        const int va = M.nrd + 6 * (b - M.nrd);
        m3 R; R.load(S.bmat[b]);
        mul(R, v3(warm()[va + 3], warm()[va + 4], warm()[va + 5])).store(S.tw_()[sl]);
        v3(warm()[va], warm()[va + 1], warm()[va + 2]).store(S.tw_()[sl] + 3);
        mul(R, v3(S.as[va + 3], S.as[va + 4], S.as[va + 5])).store(S.WB[sl]);
        v3(S.as[va], S.as[va + 1], S.as[va + 2]).store(S.WB[sl] + 3);
      }
    }
    SYNC();
    real cw, cs;
    {
      real c_w = 0, c_s = 0;
      PAR(c, S.ncon) {
        const bool hasA = S.cA[c] >= 0, hasB = S.cB[c] >= 0;
        const int slA = hasA ? slot_of(S.cA[c]) : 0, slB = hasB ? slot_of(S.cB[c]) : 0;
        real ew[NB], es[NB];
        contact_image(c, S.tw_()[slA], S.tw_()[slB], hasA, hasB, ew);
        contact_image(c, S.WB[slA], S.WB[slB], hasA, hasB, es);
#pragma unroll
        for (int k = 0; k < NB; k++) { ew[k] += S.ceoff_()[c][k]; es[k] += S.ceoff_()[c][k]; S.ce[c][k] = ew[k]; S.cde_()[c][k] = es[k]; }
        const real D = S.cD[c];
        if (S.cdim[c] == 1) {
          if (ew[0] < 0) c_w += (real)0.5 * D * ew[0] * ew[0];
          if (es[0] < 0) c_s += (real)0.5 * D * es[0] * es[0];
        } else
#pragma unroll
        for (int k = 1; k < NB; k++) {   // compile-time trip count keeps ew / es in registers
          if (k >= S.cdim[c]) continue;
          const real mu = row_mu(c, k);
          real rp = ew[0] + mu * ew[k], rm = ew[0] - mu * ew[k];
          if (rp < 0) c_w += (real)0.5 * D * rp * rp;
          if (rm < 0) c_w += (real)0.5 * D * rm * rm;
          rp = es[0] + mu * es[k]; rm = es[0] - mu * es[k];
          if (rp < 0) c_s += (real)0.5 * D * rp * rp;
          if (rm < 0) c_s += (real)0.5 * D * rm * rm;
        }
      }
      PAR(s2, S.nsr) {
        real vw = S.sr_c1[s2] * warm()[S.sr_d1[s2]], vs = S.sr_c1[s2] * S.as[S.sr_d1[s2]];
        if (S.sr_d2[s2] >= 0) { vw += S.sr_c2[s2] * warm()[S.sr_d2[s2]]; vs += S.sr_c2[s2] * S.as[S.sr_d2[s2]]; }
        vw -= S.sr_aref[s2]; vs -= S.sr_aref[s2];
        S.sr_jar[s2] = vw; S.tmpv[s2] = vs;
        if (!S.sr_uni[s2] || vw < 0) c_w += (real)0.5 * S.sr_D[s2] * vw * vw;
        if (!S.sr_uni[s2] || vs < 0) c_s += (real)0.5 * S.sr_D[s2] * vs * vs;
      }
      PAR(i, nv) c_w += (real)0.5 * (S.Ma[i] - S.fs[i]) * (S.x[i] - S.as[i]);   // Gauss term; it vanishes at qacc_smooth
      cw = WAVE_SUM(c_w); cs = WAVE_SUM(c_s);
    }
    SYNC();
    real cost;
    if (cw < cs) cost = cw;
    else {
      cost = cs;
      PAR(i, nv) { S.x[i] = S.as[i]; S.Ma[i] = S.Mv_()[i]; }
      PAR(c, S.ncon) for (int k = 0; k < NB; k++) S.ce[c][k] = S.cde_()[c][k];
      PAR(s, S.nsr) S.sr_jar[s] = S.tmpv[s];
      SYNC();
    }
    const real scale = (real)1 / ((real)M.meaninertia * (real)(nv > 1 ? nv : 1));
    const real tolerance = (real)M.tolerance;
    PROF(PF_NEWTON_INIT);
    int iters = 0;
    real improvement = 1;
    for (int it = 0;; it++) {
      // MuJoCo evaluates gradient + Hessian, then tests (improvement < tol || |grad| < tol) and the iteration cap; the tests that do
      // not need the new gradient come first here, the gradient test sits inside newton_direction before the Hessian is built
      if (it > 0 && improvement < tolerance) break;
      if (newton_direction(it > 0, scale, tolerance)) break;   // single call site
      if (it >= M.iterations) break;
      iters = it + 1;
      PROF_T0();
      mat_vec_M(S.search, S.Mv_());
      images(S.search, false, S.cde_(), S.sr_jv_());
      PROF(PF_IMAGES);
      real q1 = 0, q2 = 0, sn = 0;
      PAR(i, nv) { q1 += S.search[i] * (S.Ma[i] - S.fs[i]); q2 += S.search[i] * S.Mv_()[i]; sn += S.search[i] * S.search[i]; }
#if !defined(UR5_EMUL) && UR5_NT > 64
      block_sum3(q1, q2, sn); sn = sqrt(sn);
#else
      q1 = WAVE_SUM(q1); q2 = WAVE_SUM(q2); sn = sqrt(WAVE_SUM(sn));
#endif
      if (sn < (real)1e-15) break;
      real gtol = tolerance * (real)0.01 * sn / scale;
      real lo = 0, hi = -1, a = 0, d1, d2;
#if !defined(UR5_EMUL) && ((defined(UR5_MANY) && UR5_NT > 64 && !defined(UR5_LS_BLOCK)) || (!defined(UR5_MANY) && !defined(UR5_SMALL_LS_LDS)))
      // The exact line search is a scalar iteration over sums of <= 160 contacts + 16 rows: with the contacts spread over the workgroup every evaluation paid two
      // workgroup barriers and an LDS round trip for ~100 instructions of work. Wavefront 0 alone takes all of it: lane l keeps contacts l, l + 64, l + 128 (their
      // images along the iterate and along the search direction, friction factors folded in) and special row l in registers for the whole search, an evaluation is
      // pure arithmetic + three DPP wave sums, and the other wavefronts wait at ONE barrier for the step length and the constraint cost at it.
      // The wavefront-per-scene kernel searches the same way (its scene has one wavefront: nothing to broadcast, contact l on lane l): +3.2 % env-steps/s on the headline
      // workload, same-box A/B (profiles/r04_n_ab_small_line_search_in_registers.log); -DUR5_SMALL_LS_LDS is the old search that re-reads the images from LDS per evaluation
      real ccost_a = 0;
#if UR5_NT > 64
#define UR5_LS_SUM(v) ur5_wave_sum(v)
      if (UR5_LANE < 64) {
#else
#define UR5_LS_SUM(v) group_sum(v)
      {
#endif
        constexpr int CPL = (UR5_MAXCON + 63) / 64;
        real e0[CPL], j0[CPL], Dc[CPL], ek[CPL][NB - 1], jk[CPL][NB - 1];
        int nk[CPL];                        // friction directions of the slot's contact (0: a frictionless contact, one row); -1: no contact
#pragma unroll
        for (int t = 0; t < CPL; t++) {
          const int c = UR5_LANE + 64 * t;
          nk[t] = -1; e0[t] = j0[t] = Dc[t] = 0;
#pragma unroll
          for (int k = 0; k < NB - 1; k++) ek[t][k] = jk[t][k] = 0;
          if (c < S.ncon) {
            const int cdim = S.cdim[c];
            nk[t] = cdim == 1 ? 0 : cdim - 1;
            e0[t] = S.ce[c][0]; j0[t] = S.cde_()[c][0]; Dc[t] = S.cD[c];
#pragma unroll
            for (int k = 1; k < NB; k++) if (k < cdim) { const real mu = row_mu(c, k); ek[t][k - 1] = mu * S.ce[c][k]; jk[t][k - 1] = mu * S.cde_()[c][k]; }
          }
        }
        const bool has_row = UR5_LANE < S.nsr;
        const real r0 = has_row ? S.sr_jar[UR5_LANE] : (real)0, rv = has_row ? S.sr_jv_()[UR5_LANE] : (real)0, rD = has_row ? S.sr_D[UR5_LANE] : (real)0;
        const bool r_uni = has_row && S.sr_uni[UR5_LANE] != 0;
        const int ncon = S.ncon;
        auto eval = [&](const real alpha, real& cc, real& g1, real& g2) {
          cc = 0; g1 = 0; g2 = 0;
#pragma unroll
          for (int t = 0; t < CPL; t++) {
            if (64 * t >= ncon) continue;                                     // wave-uniform: most steps have fewer than 64 contacts
            if (nk[t] < 0) continue;
            const real D = Dc[t], t0 = e0[t] + alpha * j0[t];
            if (nk[t] == 0) { if (t0 < 0) { cc += (real)0.5 * D * t0 * t0; g1 += D * t0 * j0[t]; g2 += D * j0[t] * j0[t]; } }
            else {
#pragma unroll
              for (int k = 0; k < NB - 1; k++) {
                if (k >= nk[t]) continue;
                const real tk = ek[t][k] + alpha * jk[t][k];
                const real rp = t0 + tk, rm = t0 - tk, bp = j0[t] + jk[t][k], bm = j0[t] - jk[t][k];
                if (rp < 0) { cc += (real)0.5 * D * rp * rp; g1 += D * rp * bp; g2 += D * bp * bp; }
                if (rm < 0) { cc += (real)0.5 * D * rm * rm; g1 += D * rm * bm; g2 += D * bm * bm; }
              }
            }
          }
          if (has_row) { const real r = r0 + alpha * rv; if (!r_uni || r < 0) { cc += (real)0.5 * rD * r * r; g1 += rD * r * rv; g2 += rD * rv * rv; } }
          cc = UR5_LS_SUM(cc); g1 = UR5_LS_SUM(g1); g2 = UR5_LS_SUM(g2);
        };
        real kc, k1, k2;
        eval(0, kc, k1, k2);
        d1 = k1 + q1; d2 = k2 + q2;
        if (d1 < 0) {
          for (int ls = 0; ls < 50; ls++) {
            real an = a - d1 / d2;
            if (hi > 0 && (an <= lo || an >= hi)) an = (real)0.5 * (lo + hi);
            if (hi < 0 && an <= lo) an = 2 * lo + (real)1e-12;
            a = an;
            eval(a, kc, k1, k2);
            d1 = k1 + q1 + a * q2; d2 = k2 + q2;
            if (fabs(d1) <= gtol) break;
            if (d1 < 0) lo = a; else hi = a;
          }
        }
#if UR5_NT > 64
        if (UR5_LANE == 0) { S.red[0] = a; S.red[1] = kc; }                  // kc: the constraint cost at the last point evaluated = the accepted step
      }
      SYNC();
      a = S.red[0]; ccost_a = S.red[1];
#else
        ccost_a = kc;
      }
#endif
#undef UR5_LS_SUM
      if (a <= 0) break;
      SYNC();
      PAR(i, nv) { S.x[i] += a * S.search[i]; S.Ma[i] += a * S.Mv_()[i]; }
      PAR(c, S.ncon) for (int k = 0; k < NB; k++) { S.ce[c][k] += a * S.cde_()[c][k]; S.cde_()[c][k] = 0; }
      PAR(s, S.nsr) { S.sr_jar[s] += a * S.sr_jv_()[s]; S.sr_jv_()[s] = 0; }
      SYNC();
      real newcost = gauss_cost(S.x, S.Ma) + ccost_a;
#else
      { Cost3 k0 = constraint_cost(0); d1 = k0.d1 + q1; d2 = k0.d2 + q2; }
      if (d1 < 0) {
        for (int ls = 0; ls < 50; ls++) {
          real an = a - d1 / d2;
          if (hi > 0 && (an <= lo || an >= hi)) an = (real)0.5 * (lo + hi);
          if (hi < 0 && an <= lo) an = 2 * lo + (real)1e-12;
          a = an;
          Cost3 ka = constraint_cost(a);
          d1 = ka.d1 + q1 + a * q2; d2 = ka.d2 + q2;
          if (fabs(d1) <= gtol) break;
          if (d1 < 0) lo = a; else hi = a;
        }
      }
      if (a <= 0) break;
      SYNC();
      PAR(i, nv) { S.x[i] += a * S.search[i]; S.Ma[i] += a * S.Mv_()[i]; }
      PAR(c, S.ncon) for (int k = 0; k < NB; k++) { S.ce[c][k] += a * S.cde_()[c][k]; S.cde_()[c][k] = 0; }
      PAR(s, S.nsr) { S.sr_jar[s] += a * S.sr_jv_()[s]; S.sr_jv_()[s] = 0; }
      SYNC();
      real newcost = gauss_cost(S.x, S.Ma) + constraint_cost(0).c;
#endif
      improvement = scale * (cost - newcost);
      cost = newcost;
      SYNC();
      PROF(PF_LINESEARCH);
    }
    if (UR5_LANE == 0) S.solver_iters += iters;
    SYNC();
  }

  // ------------------------------------------------------------------ mj_Euler with implicit joint damping, then the clock [3P, C.5]
  UR5_PHASE_G void integrate(const Fact& fr) {
    const real h = (real)M.timestep;
    PAR(i, M.nv) {
      warm()[i] = S.x[i];
      if (i < M.nrd) { real s = 0; for (int e = 0; e < M.nrd; e++) s += S.Mr[i][e] * S.x[e]; S.tmpv[i] = s; }
    }
    SYNC();
#ifdef UR5_EMUL
    chol_solve(&S.Ld[0][0], M.nrd, UR5_MAXRD + 1, S.tmpv);
#else
    {   // (Mr + h B) qacc' = Mr qacc: lanes 8-15 hold the rows of chol(Mr + h B); S.Ma..S.Mv_() are dead after the solve (scratch)
      Blk b; b.base = fr.base; b.loc = fr.loc; b.size = fr.size;
      const int l8 = UR5_LANE - UR5_MAXRD;
      real rhs = (l8 >= 0 && l8 < M.nrd) ? S.tmpv[l8] : (real)0;
      real xs = blk_solve(fr.r, fr.inv, b, rhs, S.Ma, 2 * UR5_MAXRD);
      SYNC();
      if (l8 >= 0 && l8 < M.nrd) S.tmpv[l8] = xs;
      SYNC();
    }
#endif
    PAR(i, M.nv) {
      if (i < M.nrd) { qvel()[i] += h * S.tmpv[i]; qpos()[i] += h * qvel()[i]; }
      else {
        int k = (i - M.nrd) / 6, j = (i - M.nrd) % 6;
        real md = S.Mobj[i - M.nrd], damp = (real)M.obj_damp[k][j < 3 ? 0 : 1];
        real acc = damp > 0 ? md * S.x[i] / (md + h * damp) : S.x[i];
        qvel()[i] += h * acc;
        if (j < 3) qpos()[M.nrd + 7 * k + j] += h * qvel()[i];
      }
    }
    SYNC();
    PAR(k, M.nobj) {
      int va = M.nrd + 6 * k + 3, qa = M.nrd + 7 * k + 3;
      v3 w(qvel()[va], qvel()[va + 1], qvel()[va + 2]);
      real ang = norm(w) * h;
      q4 q{qpos()[qa], qpos()[qa + 1], qpos()[qa + 2], qpos()[qa + 3]};
      if (ang > 0) {
        v3 ax = normalized(w);
        real s = sin((real)0.5 * ang);
        q = qmul(q, q4{cos((real)0.5 * ang), ax.x * s, ax.y * s, ax.z * s});
      }
      q = qnormalize(q);
      qpos()[qa] = q.w; qpos()[qa + 1] = q.x; qpos()[qa + 2] = q.y; qpos()[qa + 3] = q.z;
    }
    if (UR5_LANE == 0) S.rec[UR5_REC_MISC + 2] += h;
    SYNC();
  }

  UR5_CALL void integrate_fn(const Fact& fr) { integrate(fr); }
  UR5_FN void forward(Fact& fr) {
    PROF_T0();
    kinematics(); PROF(PF_KIN);
    crb_and_factor(fr); PROF(PF_CRB);
    velocity_stage(fr); PROF(PF_VEL);
    collision();
    PROF_RE();
    make_constraints(); PROF(PF_ROWS);
    solve_newton();
  }
  UR5_BIG void step_body() {  // sim.step(), MujocoController.py:379
    Fact fr;
    forward(fr);
    PROF_T0();
    // the six-object instantiation (NV = 44) sits at the 256-register cap inside the step: with the integration inlined its block Cholesky of the robot
    // factors reloaded ~170 spilled values per step; as a real function (the factors travel in registers: argument promotion of the internal function)
    // the step spills 15 / reloads 11. Same-box A/B (profiles/r03_g_ab_phase_functions.log): it4 rounds +7 %; the NV = 32 kernel loses 0.7 % that way and
    // keeps it inlined.
    if constexpr (!FLAT && NV_ > 32) integrate_fn(fr); else integrate(fr);
    PROF(PF_INTEGRATE);
    if (UR5_LANE == 0) S.total_steps++;
    guard_state();
  }
  // mj_step's state guard [3P]: mj_checkPos / mj_checkVel / mj_checkAcc -> mjWARN_BAD* + mj_resetData when an entry is NaN or beyond mjMAXVAL = 1e10:
  // the scene returns to qpos0 with zero velocity / warm start / controls / time (the PID state is the controller's and persists) and stays flagged
  // (UR5_ST_NAN, sticky until the next reset). Same rule, same place as oracle Sim::step().
  UR5_FN void guard_state() {
    bool bad = false;
    PAR(i, M.nq + M.nv) { const real v = S.rec[i < M.nq ? UR5_REC_QPOS + i : UR5_REC_QVEL + (i - M.nq)]; if (!(fabs(v) <= (real)1e10)) bad = true; }
    if (bad) S.badstate = 1;   // benign race: every writer stores the same value
    SYNC();
    if (S.badstate) {
      SYNC();
      PAR(i, M.nq) qpos()[i] = i < M.nrd ? (real)M.rd_qpos0[i] : (real)M.obj_qpos0[(i - M.nrd) / 7][(i - M.nrd) % 7];
      PAR(i, M.nv) { qvel()[i] = 0; warm()[i] = 0; }
      PAR(a, M.nu) ctrl()[a] = 0;
      if (UR5_LANE == 0) { S.rec[UR5_REC_MISC + 2] = 0; S.status |= UR5_ST_NAN; S.badstate = 0; invalidate_pair_cache(); }
      SYNC();
    }
  }
  // In the wavefront-per-scene kernel the step is a real function: the script interpreter, the IK and the PID around it then have their own
  // register allocation, and nothing lane-derived (LDS addresses, lane predicates) that the step uses can be hoisted out of the script's
  // loops and spilled there -- the kernel reloaded ~60 such values per step. With -enable-ipra the call itself saves no registers.
  UR5_CALL void step_fn() { step_body(); }
  UR5_FN void step() { if constexpr (FLAT) step_body(); else step_fn(); }

  // ------------------------------------------------------------------ controller layer (MujocoController.py)
  // :325-329 -- all 7 PIDs are evaluated every iteration; returns max |target - q| over the group
  UR5_BIG real pid_and_deltas(unsigned mask) {
    PROF_T0();
    real md = 0;
    PAR(a, M.nu) {
      real q = qpos()[M.act_dof[a]];
      real err = target()[a] - q;
      real dterm = -(real)M.pid_kd[a] * (q - pid_in()[a]) / S.pid_dt;
      real out = clampv(kp()[a] * err + dterm, (real)M.pid_lo[a], (real)M.pid_hi[a]);
      pid_in()[a] = q; pid_out()[a] = out; ctrl()[a] = out;
      if (mask >> a & 1u) md = maxv(md, fabs(err));
    }
    md = WAVE_MAX(md);
    SYNC();
    PROF(PF_PID);
    return md;
  }
  UR5_FN unsigned mask_all() const { return (1u << M.nu) - 1u; }

  // ee_link pose for the 6 arm angles; every lane computes it (wave-uniform)
  // (loops over the 6 arm joints / the 5 x 6 normal equations have compile-time bounds and are fully unrolled: q, the axes, J and A then
  // live in registers -- with run-time indices they were scratch-memory arrays and one IK call cost as much as 6 physics steps)
  UR5_BIG void arm_fk(const real (&q6)[6], v3* p, m3* Rout, v3 (&axes)[6], v3 (&anchors)[6]) const {
    v3 pos;
    q4 quat{1, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < 6; d++) {
      if (d > M.ee_cbody) continue;
      pos = pos + mul(qmat(quat), v3(M.rd_pos[d]));
      quat = qmul(quat, q4{(real)M.rd_quat[d][0], (real)M.rd_quat[d][1], (real)M.rd_quat[d][2], (real)M.rd_quat[d][3]});
      m3 Rb = qmat(quat);
      v3 anchor = pos + mul(Rb, v3(M.rd_jpos[d]));
      axes[d] = mul(Rb, v3(M.rd_jaxis[d]));
      anchors[d] = anchor;
      real a = (real)0.5 * (q6[d] - (real)M.rd_qpos0[d]);
      real s = sin(a), c = cos(a);
      quat = qmul(quat, q4{c, (real)M.rd_jaxis[d][0] * s, (real)M.rd_jaxis[d][1] * s, (real)M.rd_jaxis[d][2] * s});
      pos = anchor - mul(qmat(quat), v3(M.rd_jpos[d]));
    }
    m3 R = qmat(qnormalize(quat));
    *p = pos + mul(R, v3(M.ee_pos));
    m3 E; E.load(M.ee_mat);
    *Rout = matmul(R, E);
  }
  // :467-517 -- fixed-iteration Levenberg-Marquardt from the home pose, identical to oracle Sim::ik()
  UR5_BIG bool ik(v3 ee_position, real* out5) const {
    v3 tgt = ee_position + v3(0, (real)-0.005, (real)0.16);
    real q[6] = {0, (real)-1.57, (real)1.57, (real)-1.57, (real)-1.57, 0};
    const real lambda = (real)1e-4;
    for (int it = 0; it < 60; it++) {
      v3 p; m3 R; v3 ax[6], an[6];
      arm_fk(q, &p, &R, ax, an);
      v3 xe = R.col(0);
      real r[6] = {p.x - tgt.x, p.y - tgt.y, p.z - tgt.z, xe.x, xe.y, xe.z + 1};
      real J[6][5];
#pragma unroll
      for (int j = 0; j < 5; j++) {
        v3 dp = cross(ax[j], p - an[j]), dx = cross(ax[j], xe);
        J[0][j] = dp.x; J[1][j] = dp.y; J[2][j] = dp.z; J[3][j] = dx.x; J[4][j] = dx.y; J[5][j] = dx.z;
      }
      real A[5][6];
#pragma unroll
      for (int i = 0; i < 5; i++) {
#pragma unroll
        for (int j = 0; j < 5; j++) {
          real s = 0;
#pragma unroll
          for (int k = 0; k < 6; k++) s += J[k][i] * J[k][j];
          A[i][j] = s;
        }
        A[i][i] += lambda;
        real s = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) s += J[k][i] * r[k];
        A[i][5] = -s;
      }
#pragma unroll
      for (int c = 0; c < 5; c++) {
        int piv = c;
        real best = fabs(A[c][c]);
#pragma unroll
        for (int i = c + 1; i < 5; i++) { const real v = fabs(A[i][c]); if (v > best) { best = v; piv = i; } }   // = "if |A[i][c]| > |A[piv][c]|"
#pragma unroll
        for (int k = 0; k < 6; k++) {   // swap rows c and piv without a run-time row index
          const real rc = A[c][k];
          real rp = rc;
#pragma unroll
          for (int i = c + 1; i < 5; i++) rp = piv == i ? A[i][k] : rp;
#pragma unroll
          for (int i = c + 1; i < 5; i++) A[i][k] = piv == i ? rc : A[i][k];
          A[c][k] = rp;
        }
#pragma unroll
        for (int i = c + 1; i < 5; i++) {
          const real f = A[i][c] / A[c][c];
#pragma unroll
          for (int k = c; k < 6; k++) A[i][k] -= f * A[c][k];
        }
      }
      real dq[5];
#pragma unroll
      for (int i = 4; i >= 0; i--) {
        real s = A[i][5];
#pragma unroll
        for (int k = i + 1; k < 5; k++) s -= A[i][k] * dq[k];
        dq[i] = s / A[i][i];
      }
#pragma unroll
      for (int j = 0; j < 5; j++) q[j] = clampv(q[j] + clampv(dq[j], (real)-0.5, (real)0.5), (real)M.rd_lo[j], (real)M.rd_hi[j]);
    }
    v3 p; m3 R; v3 ax[6], an[6];
    arm_fk(q, &p, &R, ax, an);
    for (int j = 0; j < 5; j++) out5[j] = q[j];
    return norm(p - tgt) <= (real)0.02;
  }
  // ------------------------------------------------------------------ one launch = one script per scene
  // Every operation of the C ABI is a short script over ONE blocking primitive, "move the group until converged or out of
  // steps" (MujocoController.py:269-393). The interpreter below keeps a wave-uniform program counter and has exactly one
  // call site for ik(), pid_and_deltas() and step(), so the whole physics step is inlined once into the kernel.
  struct Prim {
    bool done, need_ik;
    int repeat;           // stay(): number of 10-step chunks; otherwise 1
    unsigned mask;
    real tol;
    int max_steps;
    v3 xyz;               // gripper-centre target for need_ik
  };
  UR5_FN void write_target(int a, real v) { SYNC(); if (UR5_LANE == 0) target()[a] = v; SYNC(); }
  UR5_FN void write_kp0(real v) { SYNC(); if (UR5_LANE == 0) kp()[0] = v; SYNC(); }
  UR5_FN int stay_chunks_for(real ms) const { return (int)ceil(ms / (real)1000 / (real)M.timestep / (real)10 - (real)1e-9); }  // :621-636, H2

  // The scripted aiming rule of a multi-round launch (Ur5Launch::rule_*; bench.py It1Rounds.actions, rule "aimed"): in round r scene g tries the boxes
  // (g + (g + r) % ep + i) % nobj, i = 0.., and aims at the first one that still lies on the pick plate, z = the fixed grasp height, wrist rotation (g / ep + r) % 6;
  // an empty plate gets an attempt at the fallback point. Reads the scene's own record only. Also the seed of the episode the scene starts after the round (0: none).
  struct Aim { v3 xyz; int rotation; bool found; unsigned long long seed; };
  UR5_FN Aim aim_rule(const Ur5Launch& P, int env, int round) const {
    Aim a;
    const long long g = P.rule_gid0 + env;
    const int r = P.rule_r0 + round, ep = P.rule_ep;
    const int j = (int)((g + r) % ep);
    real x = (real)P.rule_plate[6], y = (real)P.rule_plate[7];
    a.found = false;
    for (int i = 0; i < M.nobj && !a.found; i++) {
      const int k = (int)((g + j + i) % M.nobj);
      const real* q = S.rec + UR5_REC_QPOS + M.nrd + 7 * k;
      const real px = M.obj_kind[k] == 0 ? q[0] + (real)M.obj_pos0[k][0] : q[0], py = M.obj_kind[k] == 0 ? q[1] + (real)M.obj_pos0[k][1] : q[1],
                 pz = M.obj_kind[k] == 0 ? q[2] + (real)M.obj_pos0[k][2] : q[2];
      if (fabs(px) <= (real)P.rule_plate[0] && fabs(py - (real)P.rule_plate[1]) <= (real)P.rule_plate[2] && pz >= (real)P.rule_plate[3] && pz <= (real)P.rule_plate[4]) { x = px; y = py; a.found = true; }
    }
    a.xyz = v3(x, y, (real)P.rule_plate[5]);
    a.rotation = (int)((g / ep + r) % 6);
    const long long kk = g + r + 1;
    a.seed = kk % ep == 0 ? P.rule_base_seed + (unsigned long long)g + (unsigned long long)P.rule_ntotal * (unsigned long long)(kk / ep) : 0ull;
    return a;
  }
  // ------------------------------------------------------------------ the scripts (one text for both interpreters below)
  // State of a scene's script: program counter, the last primitive's outcome, and the grasp script's registers (GraspingEnv.py:205-386; mirrors oracle
  // Sim::grasp_attempt). Wave-uniform: in the wavefront-per-scene kernel it lives in SGPRs.
  struct Script {
    int pc = 0, result = RES_NONE, last_res = RES_NONE, last_n = 0;
    v3 coord;
    int rotation = 0, result1 = RES_NONE, result_final = RES_NONE;
    bool result_grasp = false, grasped = false;
    // several rounds of the scene in one launch (Ur5Launch::rounds): the round in flight, whether it skips, the seed of the episode that starts after it
    int round = 0;
    bool skip_round = false;
    unsigned long long round_seed = 0;
  };
  UR5_FN static bool ruled(const Ur5Launch& P) { return P.op == UR5_OP_GRASP && P.rule_kind != 0; }
  UR5_FN void record(const Ur5Launch& P, int env, int slot, int res, int n) {
    if (UR5_LANE == 0) { if (P.phase_steps) P.phase_steps[12 * env + slot] = n; if (P.phase_result) P.phase_result[12 * env + slot] = res; }
  }
  // start of a grasp round: what save() + load() + the caller's action record do between two launches of the lock-step shape
  UR5_FN void begin_round(const Ur5Launch& P, int env, Script& sc) {
    SYNC();
    if (ruled(P)) {
      const Aim a = aim_rule(P, env, sc.round);
      sc.coord = a.xyz; sc.rotation = a.rotation; sc.skip_round = false; sc.round_seed = a.seed;
      if (UR5_LANE == 0 && P.action_out) {
        double* o = P.action_out + ((size_t)sc.round * P.n_env + env) * 8;
        o[0] = (double)a.xyz.x; o[1] = (double)a.xyz.y; o[2] = (double)a.xyz.z; o[3] = (double)a.rotation; o[4] = 0; o[5] = a.found ? 1.0 : 0.0; o[6] = 0; o[7] = 0;
      }
    } else {
      sc.coord = v3((real)P.target[8 * env], (real)P.target[8 * env + 1], (real)P.target[8 * env + 2]);
      sc.rotation = (int)P.target[8 * env + 3]; sc.skip_round = P.target[8 * env + 4] != 0;
      sc.round_seed = P.reset_seeds ? P.reset_seeds[env] : 0ull;
    }
    sc.result1 = RES_NONE; sc.result_final = RES_NONE; sc.result_grasp = false; sc.grasped = false;
    if (UR5_LANE == 0) for (int i = 0; i < 12; i++) { if (P.phase_steps) P.phase_steps[12 * env + i] = 0; if (P.phase_result) P.phase_result[12 * env + i] = -1; }
    if (sc.round > 0) { if (UR5_LANE == 0) { S.last_steps = 0; S.ncon = 0; S.nsr = 0; invalidate_pair_cache(); } SYNC(); }   // load()'s fresh per-launch fields
  }
  UR5_FN void begin_script(const Ur5Launch& P, int env, Script& sc) {
    if (P.op == UR5_OP_MOVE_EE || P.op == UR5_OP_IK) sc.coord = v3((real)P.target[8 * env], (real)P.target[8 * env + 1], (real)P.target[8 * env + 2]);
    if (P.op == UR5_OP_GRASP) begin_round(P, env, sc);
  }
  // script logic: consume the previous primitive's result (sc.last_res / sc.last_n), choose the next primitive (pr; pr.done: the script is over, sc.result is its
  // result) and the phase slot it reports into. Every operation of the C ABI is a short script over ONE blocking primitive.
  UR5_FN void choose(const Ur5Launch& P, int env, Script& sc, Prim& pr, int& slot) {
    const int op = P.op;
    int& pc = sc.pc; int& result = sc.result; const int last_res = sc.last_res, last_n = sc.last_n;
    v3& coord = sc.coord; int& rotation = sc.rotation; int& result1 = sc.result1; int& result_final = sc.result_final;
    bool& result_grasp = sc.result_grasp; bool& grasped = sc.grasped;
    const int nrounds = ruled(P) && P.rounds > 1 ? P.rounds : 1;
    (void)rotation; (void)last_n;
    if (op == UR5_OP_MOVE) {
      if (pc == 0) {
        pr.mask = P.group_mask[env];
        SYNC();
        if (UR5_LANE == 0 && P.target) {
          int k = 0;
          for (int a = 0; a < M.nu; a++) if (pr.mask >> a & 1u) { double t = P.target[8 * env + k++]; if (t == t) target()[a] = (real)t; }
        }
        SYNC();
        pr.tol = (real)P.tol[env]; pr.max_steps = P.max_steps[env];
      } else { result = last_res; pr.done = true; }
    } else if (op == UR5_OP_STAY) {
      if (pc == 0) { pr.mask = mask_all(); pr.tol = (real)1e-7; pr.max_steps = 10; pr.repeat = P.max_steps[env]; }
      else { result = RES_SUCCESS; pr.done = true; }
    } else if (op == UR5_OP_MOVE_EE) {
      if (pc == 0) { pr.need_ik = true; pr.xyz = coord; pr.mask = 0x1fu; pr.tol = (real)P.tol[env]; pr.max_steps = P.max_steps[env]; }
      else { result = last_res; pr.done = true; }
    } else if (op == UR5_OP_GRASP) {
      const real table_height = (real)P.table_height;
      bool chosen = false;
      while (!chosen) {
        chosen = true;
        switch (pc) {
          case 0:   // GraspEnv.step's skip rule (GraspingEnv.py:124-131): the caller flags targets it must not act on
            if (sc.skip_round) { result = 0; pr.done = true; break; }
            // :212 move above the target
            pr.need_ik = true; pr.xyz = v3(coord.x, coord.y, (real)1.1); pr.mask = 0x1fu; pr.tol = (real)0.05; pr.max_steps = 1000; slot = 0; break;
          case 1:   // :227-239 centre fallback when the IK failed
            result1 = last_res;
            if (result1 == RES_IK_FAIL) { pr.need_ik = true; pr.xyz = v3(0, (real)-0.6, (real)1.1); pr.mask = 0x1fu; pr.tol = (real)0.05; pr.max_steps = 1000; slot = 0; }
            else { pc = 3; chosen = false; }
            break;
          case 2: result1 = last_res; pc = 3; chosen = false; break;
          case 3:   // :242 stuck -> skip the grasp; else :252 rotate the wrist
            if (result1 == RES_MAX_STEPS) { pc = 9; chosen = false; break; }
            {
              const real rot_deg[6] = {0, 30, 60, 90, -30, -60};
              real deg = rotation == 0 ? rot_deg[0] : rotation == 1 ? rot_deg[1] : rotation == 2 ? rot_deg[2] : rotation == 3 ? rot_deg[3] : rotation == 4 ? rot_deg[4] : rot_deg[5];
              write_target(5, deg * (real)3.14159265358979323846 / (real)180);
            }
            pr.mask = mask_all(); pr.tol = (real)0.05; pr.max_steps = 500; slot = 1; break;
          case 4:   // :255 open_gripper(half=True)
            write_target(6, (real)0); pr.mask = 1u << 6; pr.tol = (real)0.05; pr.max_steps = 1000; slot = 2; break;
          case 5:   // :258-269 descend
            pr.need_ik = true; pr.xyz = v3(coord.x, coord.y, maxv(table_height, coord.z - (real)0.01)); pr.mask = 0x1fu; pr.tol = (real)0.01; pr.max_steps = 300; slot = 3; break;
          case 6:   // :272-277 could not reach -> no grasp; else stay(100)
            if (last_res == RES_MAX_STEPS) { pc = 9; chosen = false; break; }
            pr.mask = mask_all(); pr.tol = (real)1e-7; pr.max_steps = 10; pr.repeat = stay_chunks_for(100); break;
          case 7:   // :278 grasp() = close_gripper(max_steps=300)
            write_target(6, (real)-0.4); pr.mask = 1u << 6; pr.tol = (real)0.01; pr.max_steps = 300; break;
          case 8:
            result_grasp = last_res != RES_SUCCESS;
            record(P, env, 5, result_grasp ? RES_MAX_STEPS : RES_SUCCESS, last_n);
            pc = 9; chosen = false; break;
          case 9:   // :282
            write_kp0(10);
            if (P.check_mode == 1) { pr.need_ik = true; pr.xyz = v3(coord.x, coord.y, (real)1.1); pr.mask = 0x1fu; pr.tol = (real)0.05; pr.max_steps = 1000; slot = 6; }
            else { pc = 12; chosen = false; }
            break;
          case 10:  // IT1 (README.md:20): 500-step closing check right after lifting
            if (result_grasp) { write_target(6, (real)-0.4); pr.mask = 1u << 6; pr.tol = (real)0.01; pr.max_steps = 500; slot = 9; }
            else { pc = 12; chosen = false; }
            break;
          case 11: result_final = last_res; pc = 12; chosen = false; break;
          case 12:  // :285 back above the table centre
            pr.need_ik = true; pr.xyz = v3(0, (real)-0.6, (real)1.1); pr.mask = 0x1fu; pr.tol = (real)0.05; pr.max_steps = 1000; slot = 7; break;
          case 13:  // :297 to the drop position
            pr.need_ik = true; pr.xyz = v3((real)0.6, 0, (real)1.15); pr.mask = 0x1fu; pr.tol = (real)0.01; pr.max_steps = 1200; slot = 8; break;
          case 14:  // :312-321 closing check at the drop position
            if (P.check_mode != 1 && result_grasp) { write_target(6, (real)-0.4); pr.mask = 1u << 6; pr.tol = (real)0.01; pr.max_steps = P.check_mode == 2 ? 100 : 1000; slot = 9; }   // check_mode 2 = demo_mode (:318-321)
            else { pc = 16; chosen = false; }
            break;
          case 15: result_final = last_res; pc = 16; chosen = false; break;
          case 16:  // :327, :338 open the gripper
            grasped = (result_final == RES_MAX_STEPS) && result_grasp;
            write_target(6, (real)0.4); pr.mask = 1u << 6; pr.tol = (real)0.05; pr.max_steps = 1000; slot = 10; break;
          case 17:  // :341-342
            if (grasped) { pr.mask = mask_all(); pr.tol = (real)1e-7; pr.max_steps = 10; pr.repeat = stay_chunks_for(200); }
            else { pc = 18; chosen = false; }
            break;
          case 18:  // :345 rotate back
            write_target(5, (real)0); pr.mask = mask_all(); pr.tol = (real)0.05; pr.max_steps = 500; slot = 11; break;
          case 19:  // :347; then, for a scene whose episode ends here, GraspEnv.reset_model (GraspingEnv.py:409-477) inside the same launch
            write_kp0(20);
            result = grasped ? 1 : 0;
            if (sc.round_seed != 0) {
              SYNC();
              if (UR5_LANE == 0) { const real ended = (real)((int)S.rec[UR5_REC_MISC + 7] | S.status); ur5_reset_record(M, P.qpos0, S.rec, sc.round_seed); S.rec[UR5_REC_MISC + 7] = ended; S.status = 0; invalidate_pair_cache(); }   // the attempt's status bits stay readable (counters: bits 8-15) after the episode reset that follows it in this launch
              SYNC();
              pr.mask = mask_all(); pr.tol = (real)1e-7; pr.max_steps = 10; pr.repeat = P.reset_chunks;   // :473 stay(1000)
              if (pr.repeat <= 0) { pc = 20; chosen = false; }
            } else { pc = 20; chosen = false; }
            break;
          case 20:  // the round is over: its reward; the scene's next round of this launch starts at once, whatever the other scenes are doing
            if (UR5_LANE == 0 && P.result) P.result[(size_t)sc.round * P.n_env + env] = result;
            sc.round++;
            if (sc.round < nrounds) { begin_round(P, env, sc); pc = 0; chosen = false; } else pr.done = true;
            break;
          default: pr.done = true; break;
        }
      }
    } else if (op == UR5_OP_STEP) {
      if (pc == 0) { pr.mask = 0; pr.tol = (real)-1; pr.max_steps = P.max_steps[env]; pr.repeat = -1; }   // repeat < 0: raw sim.step() x max_steps
      else { result = RES_SUCCESS; pr.done = true; }
    } else if (op == UR5_OP_IK) {
      if (pc == 0) { pr.need_ik = true; pr.xyz = coord; pr.repeat = 0; }
      else { result = last_res; pr.done = true; }
    } else {  // UR5_OP_FORWARD
      if (pc == 0) pr.repeat = -2; else { result = RES_SUCCESS; pr.done = true; }
    }
  }
  // One scene per wavefront (GS = 64): the script as nested loops -- its registers are wave-uniform (SGPRs) and are not live across step().
  UR5_FN void run_nested(const Ur5Launch& P, int env) {
    const int op = P.op;
    Script sc;
    begin_script(P, env, sc);
    for (;;) {
      Prim pr;
      pr.done = false; pr.need_ik = false; pr.repeat = 1; pr.mask = 0; pr.tol = 0; pr.max_steps = 0;
      int slot = -1;       // phase slot the primitive reports into (grasp script)
      choose(P, env, sc, pr, slot);
      if (pr.done) break;
      sc.pc++;
      // ---------------- the primitive
      int res = RES_NONE, steps = 0;
      bool ikfail = false;
      if (pr.need_ik) {   // :446-465 move_ee = ik + move_group("Arm")
        PROF_T0();
        real q5[5];
        bool ok = ik(pr.xyz, q5);
        SYNC();
        if (ok && op != UR5_OP_IK) { if (UR5_LANE == 0) for (int j = 0; j < 5; j++) target()[j] = q5[j]; }
        if (op == UR5_OP_IK && UR5_LANE == 0 && P.out) for (int j = 0; j < 5; j++) P.out[8 * env + j] = (double)q5[j];
        SYNC();
        ikfail = !ok;
        res = ok ? RES_SUCCESS : RES_IK_FAIL;
        PROF(PF_IK);
      }
      if (pr.repeat == -2) {
        Fact fr0;
        forward(fr0);
        if (P.debug) dump(P.debug + (size_t)UR5_DEBUG_STRIDE * env);
      } else if (!ikfail && pr.repeat != 0) {
        const bool raw = pr.repeat < 0;
        const int reps = raw ? 1 : pr.repeat;
        for (int rep = 0; rep < reps; rep++) {
          steps = 1; res = RES_NONE;
          bool reached = false;
          while (!reached) {   // MujocoController.py:318-382
            if (!raw) {
              real md = pid_and_deltas(pr.mask);
              if (md < pr.tol) { res = RES_SUCCESS; reached = true; }   // no break: one more sim.step() follows (:351-363)
            }
            if (steps > pr.max_steps) { res = RES_MAX_STEPS; break; }
            step();
            steps++;
          }
        }
        if (raw) { steps = pr.max_steps; res = RES_SUCCESS; }
      }
      SYNC();
      if (UR5_LANE == 0) S.last_steps = steps;
      SYNC();
      sc.last_res = res; sc.last_n = steps;
      if (slot >= 0) record(P, env, slot, res, steps);
    }
    if (UR5_LANE == 0) {
      if (P.result && !ruled(P)) P.result[env] = sc.result;   // (a ruled launch has filed every round's reward in its [rounds][n] array)
      if (P.steps) P.steps[env] = S.last_steps;
    }
  }

  // One loop for the whole launch. Each trip (a) advances the scene's script until it needs a physics step -- consuming the result of the
  // finished primitive, choosing the next one, running its IK, evaluating the PIDs and the termination tests of the move loop
  // (MujocoController.py:318-382) -- and (b) takes that one step. Scenes that share a wavefront (GS < 64) are at different points of their
  // scripts; with this shape they still execute every physics step together, and only the short control code of (a) diverges.
  UR5_FN void run(const Ur5Launch& P, int env, bool live = true) {
    if constexpr (FLAT) run_flat(P, env, live); else { if (live) run_nested(P, env); }
#if defined(UR5_PROFILE) && !defined(UR5_EMUL)
    if (live && UR5_LANE == 0 && P.debug && P.op != UR5_OP_FORWARD) {   // [PF_CORECLK] = start, [PF_REALCLK] = end of the scene's wave, 100 MHz ticks
      S.prof[PF_REALCLK] = (double)wall_clock64();
      for (int i = 0; i < PF_COUNT; i++) P.debug[(size_t)UR5_DEBUG_STRIDE * env + i] = S.prof[i];
    }
#endif
  }
  UR5_FN void run_flat(const Ur5Launch& P, int env, bool live) {
    const int op = P.op;
    Script sc;
    if (live) begin_script(P, env, sc);
    // the primitive in flight
    Prim pr;
    pr.done = false; pr.need_ik = false; pr.repeat = 1; pr.mask = 0; pr.tol = 0; pr.max_steps = 0;
    int slot = -1;       // phase slot the primitive reports into (grasp script)
    int res = RES_NONE, steps = 0, rep = 0, reps = 0;
    bool reached = false, raw = false, need_new = true, done = !live;
    // the primitive is over: publish its step count, hand its result to the script
    auto finish = [&]() {
      SYNC();
      if (UR5_LANE == 0) S.last_steps = steps;
      SYNC();
      sc.last_res = res; sc.last_n = steps;
      if (slot >= 0) record(P, env, slot, res, steps);
      need_new = true;
    };
    // one repetition of the move loop ended (converged, or out of steps): start the next one (stay: chunks of 10 steps) or finish
    auto end_rep = [&]() {
      rep++;
      if (rep < reps) { steps = 1; res = RES_NONE; reached = false; }
      else { if (raw) { steps = pr.max_steps; res = RES_SUCCESS; } finish(); }
    };
    while (!done) {
      bool want_step = false;
      while (!want_step && !done) {
        if (need_new) {
          pr.done = false; pr.need_ik = false; pr.repeat = 1; pr.mask = 0; pr.tol = 0; pr.max_steps = 0;
          slot = -1;
          choose(P, env, sc, pr, slot);
          if (pr.done) { done = true; break; }
          sc.pc++;
          // ---------------- the primitive
          res = RES_NONE; steps = 0;
          bool ikfail = false;
          if (pr.need_ik) {   // :446-465 move_ee = ik + move_group("Arm")
            PROF_T0();
            real q5[5];
            bool ok = ik(pr.xyz, q5);
            SYNC();
            if (ok && op != UR5_OP_IK) { if (UR5_LANE == 0) for (int j = 0; j < 5; j++) target()[j] = q5[j]; }
            if (op == UR5_OP_IK && UR5_LANE == 0 && P.out) for (int j = 0; j < 5; j++) P.out[8 * env + j] = (double)q5[j];
            SYNC();
            ikfail = !ok;
            res = ok ? RES_SUCCESS : RES_IK_FAIL;
            PROF(PF_IK);
          }
          if (pr.repeat == -2) {
            Fact fr0;
            forward(fr0);
            if (P.debug) dump(P.debug + (size_t)UR5_DEBUG_STRIDE * env);
            finish();
          } else if (!ikfail && pr.repeat != 0) {   // enter the move loop
            raw = pr.repeat < 0;
            reps = raw ? 1 : pr.repeat;
            rep = 0; steps = 1; res = RES_NONE; reached = false;
            need_new = false;
          } else finish();
        } else {   // one trip of MujocoController.py:318-382 up to (not including) its sim.step()
          if (!raw) {
            real md = pid_and_deltas(pr.mask);
            if (md < pr.tol) { res = RES_SUCCESS; reached = true; }   // no break: one more sim.step() follows (:351-363)
          }
          if (steps > pr.max_steps) { res = RES_MAX_STEPS; end_rep(); }
          else want_step = true;
        }
      }
      if (want_step) {
        step();
        steps++;
        if (reached) end_rep();
      }
    }
    if (live && UR5_LANE == 0) {
      if (P.result && !ruled(P)) P.result[env] = sc.result;
      if (P.steps) P.steps[env] = S.last_steps;
    }
  }

  // introspection for the parity tests: [0] ncon, [1] nsr, [2..] fixed sections (see tests/test_parity_forward.py)
  UR5_CALL void dump_fn(double* out) { dump_body(out); }
  UR5_FN void dump(double* out) { if constexpr (FLAT && UR5_INL_DUMP) dump_body(out); else dump_fn(out); }
  UR5_FN void dump_body(double* out) {
    SYNC();
    if (UR5_LANE != 0) return;
    int o = 0;
    out[o++] = S.ncon; out[o++] = S.nsr; out[o++] = S.solver_iters; out[o++] = S.status;
#ifdef UR5_MANY
    out[o++] = S.env_ptr[M.nv]; out[o++] = S.ncouple;   // envelope size (doubles), contacts between two movable bodies
    { int ns = 0; for (int p2 = 0; p2 <= M.nobj; p2++) if (S.blk_first[p2] != p2 || S.blk_last[p2] != p2) ns++; out[o++] = ns; }   // coupled blocks
    out[UR5_DEBUG_STRIDE - UR5_MAXCAND - 2] = S.env_inlds; out[UR5_DEBUG_STRIDE - UR5_MAXCAND - 1] = S.dc_inlds;   // where this step's envelope / block cache lived
#endif
    out[7] = S.ncand;
    for (int i = 0; i < S.ncand && i < UR5_MAXCAND; i++) out[UR5_DEBUG_STRIDE - UR5_MAXCAND + i] = S.cand[i];   // broad-phase survivors (pair indices)
    o = 8;
    for (int b = 0; b < UR5_MAXB; b++) for (int k = 0; k < 3; k++) out[o++] = b < nb() ? (double)S.bpos[b][k] : 0;      // 8   .. 50
    for (int d = 0; d < UR5_MAXRD; d++) for (int e = 0; e < UR5_MAXRD; e++) out[o++] = (double)S.Mr[d][e];               // 50  .. 114
    for (int i = 0; i < UR5_MAXNV; i++) out[o++] = i < M.nv ? (double)S.fs[i] : 0;                                      // 114 .. 158
    for (int i = 0; i < UR5_MAXNV; i++) out[o++] = i < M.nv ? (double)S.as[i] : 0;                                      // 158 .. 202
    for (int i = 0; i < UR5_MAXNV; i++) out[o++] = i < M.nv ? (double)S.x[i] : 0;                                       // 202 .. 246
    for (int c = 0; c < UR5_MAXCON; c++) {                                                                             // 246 .. 246+32*10
      bool ok = c < S.ncon;
      out[o++] = ok ? (double)S.cdist[c] : 0;
      for (int k = 0; k < 3; k++) out[o++] = ok ? (double)S.cpos[c][k] : 0;
      for (int k = 0; k < 3; k++) out[o++] = ok ? (double)S.cframe[c][k] : 0;
      out[o++] = ok ? S.cg1[c] : -1; out[o++] = ok ? S.cg2[c] : -1;
#ifdef UR5_EMUL
      out[o++] = ok ? (double)S.cfn[c] : 0;
#else
      out[o++] = 0;
#endif
    }
  }
};

#undef S
#undef M

}  // namespace ur5
#ifndef UR5_EMUL
}  // anonymous namespace
#endif
