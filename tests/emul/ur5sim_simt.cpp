// ur5sim_simt.cpp -- TEST-ONLY build of the engine's DEVICE code path for the host: csrc/ur5_engine.h compiled WITHOUT -DUR5_EMUL, its HIP
// intrinsics supplied by ur5_simt_shim.h, every scene executed as one 64-lane wavefront of fibres (see the shim's header). Covers on the CPU
// what the plain lane emulation cannot reach: DPP / readlane / shuffle code, ballot compactions, the cooperative MPR and the register-resident
// factorisations of the small-scene kernel. Slow (a rendezvous per cross-lane instruction): used for single steps and short moves only.
// Built into tests/emul/_build by tests/conftest.py; never loaded by the package. Second unit: ur5sim_simt_many.cpp (256 fibres = 4 wavefronts per scene).
#define UR5_SIMT 1
#include "ur5_simt_shim.h"
#include "../../mujoco_rl_ur5_amd/csrc/ur5_engine.h"
#include "../../mujoco_rl_ur5_amd/csrc/ur5sim_host.h"

alignas(16) double ur5_smem[20480];               // "LDS": 160 KB, one scene at a time

#ifndef UR5_SIMT_NO_RUNTIME   // (the many-object unit, ur5sim_simt_many.cpp, includes this file a second time and shares the runtime)
// Fibre switch. glibc's swapcontext makes a signal-mask system call per switch (~0.3 us, and a step has ~10^5 of them); on x86-64 the six
// callee-saved registers and the stack pointer are all that has to change hands.
#if defined(__x86_64__)
extern "C" void simt_switch(void** save_sp, void* load_sp);
asm(".text\n.globl simt_switch\n.type simt_switch,@function\nsimt_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n");
#define SIMT_ASM_SWITCH 1
#endif

namespace simt {
Tid tid{0, 0, 0}, bid{0, 0, 0};
Wave W;
static constexpr size_t STACK = 1 << 20;
#ifdef SIMT_ASM_SWITCH
static void* g_sched_sp;
static void* g_lane_sp[NL];
static inline void to_sched(int me) { simt_switch(&g_lane_sp[me], g_sched_sp); }
static inline void to_lane(int l) { simt_switch(&g_sched_sp, g_lane_sp[l]); }
#else
static inline void to_sched(int me) { swapcontext(&W.ctx[me], &W.sched); }
static inline void to_lane(int l) { swapcontext(&W.sched, &W.ctx[l]); }
#endif
void yield_blocked(const char* what) {
  const int me = W.cur;
  W.waiting[me] = what;
  if (++W.idle > 64L * NL) {
    fprintf(stderr, "simt: deadlock -- no lane can make progress. Lanes wait at:");
    for (int l = 0; l < W.nl; l++) fprintf(stderr, " %d:%s", l, W.done[l] ? "done" : (W.waiting[l] ? W.waiting[l] : "runnable"));
    fprintf(stderr, "\n");
    abort();
  }
  to_sched(me);
  W.waiting[me] = nullptr;
}
void yield_runnable() {
  const int me = W.cur;
  W.idle = 0;
  to_sched(me);
}
int g_skip_barrier = -1;
int g_order = -1;                                   // 0 ascending, 1 descending, 2 a fresh permutation per pass (ur5_simt_set_order / UR5_SIMT_ORDER)
struct Entry { void (*body)(void*); void* arg; };
static Entry g_entry;
static void lane_main() {
  g_entry.body(g_entry.arg);
  W.done[W.cur] = true;
  W.idle = 0;
  to_sched(W.cur);                                  // never resumed
}
void run_workgroup(int nlanes, void (*body)(void*), void* arg) {
  g_entry = Entry{body, arg};
  memset(W.seq, 0, sizeof W.seq);
  memset(W.nbar, 0, sizeof W.nbar);
  W.idle = 0;
  W.nl = nlanes;
  for (int l = 0; l < NL; l++) W.done[l] = true;
  for (int l = 0; l < nlanes; l++) {
    if (!W.stack[l]) W.stack[l] = (char*)malloc(STACK);
    W.done[l] = false; W.waiting[l] = nullptr;
#ifdef SIMT_ASM_SWITCH
    // initial frame: six register slots, the entry point as return address, one slot so that lane_main starts with the ABI's stack alignment
    uintptr_t top = ((uintptr_t)W.stack[l] + STACK) & ~(uintptr_t)15;
    void** sp = (void**)(top - 64);
    for (int k = 0; k < 6; k++) sp[k] = nullptr;
    sp[6] = (void*)lane_main;
    sp[7] = nullptr;
    g_lane_sp[l] = sp;
#else
    getcontext(&W.ctx[l]);
    W.ctx[l].uc_stack.ss_sp = W.stack[l];
    W.ctx[l].uc_stack.ss_size = STACK;
    W.ctx[l].uc_link = nullptr;
    makecontext(&W.ctx[l], lane_main, 0);
#endif
  }
  // Lane order of the scheduler. Results must not depend on it beyond the order in which LDS atomics land (rounding level): the engine may only
  // rely on what the rendezvous guarantee. ur5_simt_set_order(1 | 2) (or UR5_SIMT_ORDER=reverse / =shuffle) is the race detector: a missing SYNC between a lane's LDS write and
  // another lane's read shows up as a result that changes with the order (or as the NaN poison of the LDS image).
  if (g_order < 0) { const char* e = getenv("UR5_SIMT_ORDER"); g_order = !e ? 0 : (e[0] == 'r' ? 1 : (e[0] == 's' ? 2 : 0)); }
  const int mode = g_order;
  static uint64_t rng = 0x9E3779B97F4A7C15ull;
  for (;;) {
    bool any = false;
    int start = 0, stride = 1;
    if (mode == 2) { rng = rng * 6364136223846793005ull + 1442695040888963407ull; start = (int)((rng >> 33) % (uint64_t)W.nl); stride = (int)((rng >> 20) % 16) * 2 + 1; }
    for (int i = 0; i < W.nl; i++) {
      const int l = mode == 1 ? W.nl - 1 - i : (mode == 2 ? (start + i * stride) % W.nl : i);      // odd stride: a permutation of 64 / 256 lanes
      if (W.done[l]) continue;
      any = true;
      W.cur = l; tid.x = (unsigned)l;
      to_lane(l);
    }
    if (!any) break;
  }
}
}  // namespace simt
#endif   // UR5_SIMT_NO_RUNTIME

// lane 0's cross-lane operation counts since the library was loaded (dpp, readlane, shuffle, ballot, wave barrier, __syncthreads, LDS atomic, -)
#ifndef UR5_SIMT_NO_RUNTIME
extern "C" void ur5_simt_op_counts(long* out) { for (int k = 0; k < 8; k++) out[k] = simt::W.ops[k]; }
extern "C" void ur5_simt_set_order(int mode) { simt::g_order = mode; }
extern "C" void ur5_simt_skip_barrier(int k) { simt::g_skip_barrier = k; }
#endif
static int be_open(ur5_sim*, int) { return 0; }
static void be_close(ur5_sim*) {}
static void* be_alloc(ur5_sim*, size_t bytes) { void* p = malloc(bytes); if (p) memset(p, 0xFF, bytes); return p; }   // poisoned like the plain emulation
static void be_free(ur5_sim*, void* p) { free(p); }
static int be_h2d(ur5_sim*, void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
static int be_d2h(ur5_sim*, void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
static int be_d2d_async(ur5_sim*, void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); return 0; }
static int be_sync(ur5_sim*) { return 0; }
static int be_set_stream(ur5_sim*, void*, int) { return 0; }
static int be_reset_dev(ur5_sim* h, const uint64_t* seeds, const uint8_t* mask, int chunks, int* max_steps) {
  for (int e = 0; e < h->n; e++) {
    const bool on = !mask || mask[e];
    if (on) ur5host::reset_record(*h->dm, h->d_qpos0, h->d_rec + (size_t)e * UR5_REC_STRIDE, seeds[e]);
    max_steps[e] = on ? chunks : 0;
  }
  return 0;
}

// the body of ur5_run_kernel<NV, 64> (csrc/ur5sim.hip), executed by each of the 64 fibres of a wave
template <int NV> struct KernelArgs { double* rec; const Ur5Launch* P; const Ur5DevModel* model; };
template <int NV> static void kernel_body(void* a) {
  const KernelArgs<NV>& K = *(const KernelArgs<NV>*)a;
  const Ur5Launch& P = *K.P;
  const int slot = (int)blockIdx.x;
  const bool present = slot < P.n_env;
  const int env = (present && P.order) ? P.order[slot] : slot;
  const bool live = present && !(P.op == UR5_OP_STAY && P.max_steps[env] <= 0);
  ur5::Engine<double, NV, UR5_NT> eng(K.model);
  double* r = K.rec + (size_t)(live ? env : 0) * UR5_REC_STRIDE;
  if (live) eng.load(r, P.pid_dt, P.contacts_enabled, P.step_cap ? P.step_cap[env] : 0x7fffffff);
#ifdef UR5_MANY
  eng.set_hess(P.hess + (size_t)env * UR5_HESS_STRIDE);
#endif
  eng.run(P, env, live);
  if (live) eng.save(r);
}
template <int NV> static void run_all(ur5_sim* h, const Ur5Launch& P) {
  static_assert(sizeof(ur5::Lds<double, NV>) <= sizeof(ur5_smem), "LDS image fits");
  KernelArgs<NV> K{h->d_rec, &P, h->dm};
  for (int b = 0; b < h->n; b++) {
    memset(ur5_smem, 0xFF, sizeof(ur5::Lds<double, NV>));       // a workgroup starts with whatever the previous one left in the CU's LDS
    simt::bid.x = (unsigned)b;
    simt::run_workgroup(UR5_NT, kernel_body<NV>, &K);
  }
}
static int be_render(ur5_sim*, int, int, int, int, uint8_t*, float*) { return ur5host::fail(UR5_ERR_ARG, "the SIMT test build has no renderer"); }
static bool be_can_observe(ur5_sim* h) {
#ifdef UR5_MANY
  (void)h; return ur5::Engine<double, UR5_MAXNV, UR5_NT>::CAN_OBSERVE;
#else
  return h->nvt == 32 ? ur5::Engine<double, 32, UR5_NT>::CAN_OBSERVE : ur5::Engine<double, UR5_MAXNV, UR5_NT>::CAN_OBSERVE;
#endif
}
static int be_launch(ur5_sim* h, const Ur5Launch& P) {
#ifdef UR5_MANY
  run_all<UR5_MAXNV>(h, P);
#else
  if (h->nvt == 32) run_all<32>(h, P); else run_all<UR5_MAXNV>(h, P);
#endif
  return 0;
}
