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
#include "ur5_raster.h"   // the ray caster of the observation (stand-alone kernels of ur5sim.hip and Engine::observe)

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
inline const Ur5DevModel* ur5_uniform_model(const Ur5DevModel* p) { return p; }
#define UR5_LDS_PTR(T) (static_cast<T*>(ur5_emul_lds))
#define PAR(i, n) for (int i = 0; i < (n); ++i)
#define SYNC() ((void)0)
#define SYNC1() ((void)0)
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
// The scene image is STATIC LDS, one variable per Engine instantiation (round 5; rounds 1-4: dynamic LDS behind one `extern __shared__` symbol). A phase routine that is a
// real function does not know where dynamic LDS starts: the compiler hands every function the launching kernel's id and has it read the base from a table in constant
// memory (llvm.amdgcn.dynlds.offset.table) -- s_getpc / s_add / s_addc / s_lshl / s_load_dword / s_waitcnt in front of the function's LDS accesses and, with the machine
// LICM off, again inside loops (552 of the small-scene unit's 1 246 scalar loads). A static variable that only ONE kernel reaches gets an absolute address: no table, no load,
// immediate ds_* offsets. Same-box A/B: headline +2 %, six-object scenes +3 %, piles +1.3 %, bit-identical (profiles/r05_s_*; the detour over a literal address:
// profiles/r05_v_lds_base_lookup_finding.txt).
#ifdef UR5_SIMT
extern __shared__ __attribute__((aligned(16))) double ur5_smem[];   // the host fibres' "LDS" (tests/emul/ur5sim_simt.cpp)
#endif
// The model is the HANDLE's (ur5_sim::dm, uploaded once by ur5_create), a kernel argument. struct Engine's only data member is its address: the kernel holds it in a
// scalar register pair, every phase routine that is a real function gets it BY VALUE (two VGPRs at the call, v_readfirstlane in the callee -- no memory access, where
// re-reading it from the kernel arguments costs every called function a dependent scalar load: -1.5 % on the headline kernel, profiles/r05_n_*) and builds its own
// Engine around it. Constant address space + a uniform address: every uniform model read is an s_load through the scalar cache, as it was from the one
// __constant__ symbol per unit and device of rounds 1-4 (which handles with different models re-uploaded whenever they took turns).
#if defined(UR5_SIMT)
inline const Ur5DevModel* ur5_uniform_model(const Ur5DevModel* p) { return p; }
#else
__device__ __forceinline__ const Ur5DevModel* ur5_uniform_model(const Ur5DevModel* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  typedef const Ur5DevModel __attribute__((address_space(4)))* cptr;
  return (const Ur5DevModel*)(cptr)(((unsigned long long)hi << 32) | lo);
}
#endif
// A scene is owned by the GS = UR5_NT lanes of ONE workgroup (Engine<real, NV, GS>): a wavefront (64) in the small-scene unit, four wavefronts (256) in the pile unit.
// UR5_LANE is the lane's index inside it. (Rounds 1-4 also carried GS = 32, two scenes per wavefront: -30 % on the bench, profiles/r02_*, removed in round 5.)
#ifdef UR5_SIMT
#define UR5_LDS_PTR(T) (reinterpret_cast<T*>(ur5_smem))
#else
template <class T> __device__ __forceinline__ T* ur5_lds_image() { __shared__ __attribute__((aligned(16))) T image; return &image; }
#define UR5_LDS_PTR(T) (ur5_lds_image<T>())
#endif
#define PAR(i, n) for (int i = UR5_LANE; i < (n); i += GS)
// SYNC orders the LDS traffic of the lanes that share a scene. With one wavefront per workgroup (UR5_NT == 64) the hardware already
// executes a wave's LDS instructions in issue order, so all that is needed is that the COMPILER keeps the accesses on their side of
// the line: wavefront-scope fences and a scheduling barrier, no instruction. __syncthreads() in a 64-thread workgroup costs an
// `s_waitcnt lgkmcnt(0)` -- a full drain of the LDS queue and of every scalar load in flight -- at each of the several hundred SYNCs of a step.
// SYNC1: a phase boundary that only the one-wavefront unit keeps (there it is a compiler fence and costs nothing; its kernel stays as measured). In the pile unit a
// nearby workgroup barrier already orders the accesses in question -- each site says which -- and a further s_barrier would only add a rendezvous of four wavefronts.
#if UR5_NT == 64
#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#define SYNC1() SYNC()
#else
#define SYNC() __syncthreads()
#define SYNC1() ((void)0)
#endif
#define UR5_LANE ((int)threadIdx.x & (GS - 1))
#define UR5_GBASE 0
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
template <int W, class T> __device__ __forceinline__ T ur5_wave_max(T v) {   // max over aligned groups of W lanes
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) { T w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
  return v;
}
// sums / maxima over the lanes of a scene: one wavefront (GS = 64) or several (Engine::block_sum / block_max
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

#ifndef UR5_SUP_K
#define UR5_SUP_K 4   // hull vertices per lane and trip of the cooperative support scan
#endif
#ifndef UR5_SUP_DELTA
#define UR5_SUP_DELTA 0.02   // travel (m) of a moving geom after which the broad phase's pair superset is rebuilt
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
// (A lane-emulation build made with clang and -mfma -ffp-contract=fast -- tools/qacc_error_budget.py -- gets the same split: the HIP pile unit's arithmetic on the host.)
#if defined(UR5_MANY) && !defined(UR5_SIMT) && (!defined(UR5_EMUL) || defined(__clang__))
#define UR5_STRICT_GEOMETRY 1
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
// The oracle's text of the same normalisation (oracle/ur5_oracle.cpp qnormalize: four divisions). The pile unit uses it for the objects' orientations: q / n and
// q * (1 / n) differ in the last bit of some components, and its contacts are to be bit-equal to the oracle's from the same state (tools/contact_bits.py).
template <class T> UR5_FN Q4<T> qnormalize_div(Q4<T> q) {
  T n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < (T)1e-300) return Q4<T>{1, 0, 0, 0};
  return Q4<T>{q.w / n, q.x / n, q.y / n, q.z / n};
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

#ifdef UR5_STRICT_GEOMETRY
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
  int redi[UR5_NT / 64], redi2[UR5_NT / 64];
  int nheavy;                                        // broad-phase survivors refined lane by lane: they sit at the END of `cand`, last slot downwards (collision_body)
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
  int status, solver_iters, ncon_max, badstate;
  real pid_dt;
  int contacts_enabled, last_steps, total_steps, step_cap;
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
#ifdef UR5_LDS_PAD   // residency probe (make variant EXTRA=-DUR5_LDS_PAD=2400): dead bytes at the end of the image -- the four-box kernel at SEVEN scenes per CU prices the eighth
  char lds_pad_[UR5_LDS_PAD];
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
  static constexpr bool FLAT = false;
#endif
  static_assert(GS_ == UR5_NT, "one workgroup = one scene (a wavefront in the small-scene unit, four in the pile unit)");
  typedef Lds<real, NV_> L;
  typedef V3<real> v3;
  typedef M3<real> m3;
  typedef Q4<real> q4;
#define S (*UR5_LDS_PTR(L))
#define M (*mp_)
  const Ur5DevModel* mp_;   // the handle's model: the ONLY data member (the scene lives in LDS), so that an Engine is rebuilt for free inside every phase routine that is a real function
  UR5_FN explicit Engine(const Ur5DevModel* mp) : mp_(ur5_uniform_model(mp)) {}

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
    if constexpr (GS == 64) return ur5_wave_sum(v); else return block_sum(v);
  }
  UR5_FN real group_max(real v) {
    if constexpr (GS == 64) return ur5_wave_max<64>(v); else return block_max(v);
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

  UR5_FN void load(const double* rec, real dt, int con, int step_cap = 0x7fffffff) {
    PAR(i, UR5_REC_STRIDE) S.rec[i] = (real)rec[i];
    if (UR5_LANE == 0) { S.pid_dt = dt; S.contacts_enabled = con; S.last_steps = 0; S.total_steps = 0; S.step_cap = step_cap; }
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

#include "ur5_engine_dynamics.inc"   // kinematics, CRBA + factors, velocity stage (mj_kinematics / mj_crb / mj_rne [3P])
#include "ur5_engine_collision.inc"   // broad phase (pair cache), analytic pairs, box-box, Minkowski portal refinement (per lane and cooperative), contact sort
#include "ur5_engine_rows.inc"   // constraint rows: impedance / reference acceleration, side lists, coupling list (mj_makeConstraint [3P])
#include "ur5_engine_newton.inc"   // Newton solver pieces: images, costs, staged gather of wrenches / twist-space Hessians, gradient + direction, small-scene register factorisation
#include "ur5_engine_envelope.inc"   // pile unit only: envelope (skyline) storage of the Newton Hessian -- structure, assembly, level-parallel factorisation, sweeps
#include "ur5_engine_solve.inc"   // the Newton iteration itself: warm start, exact line search in registers, step
#include "ur5_engine_integrate.inc"   // mj_Euler with implicit damping, forward(), step(), mj_step's state guard
#include "ur5_engine_observe.inc"   // the observation of a round, rendered by the lanes that own the scene between two rounds of a launch
#include "ur5_engine_script.inc"   // controller layer (PID, IK), the scripts of the C ABI operations, the interpreter, the introspection dump
};

#undef S
#undef M

}  // namespace ur5
#ifndef UR5_EMUL
}  // anonymous namespace
#endif
