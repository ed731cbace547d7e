// ur5sim_emul.cpp -- TEST-ONLY build of the engine: compiles csrc/ur5_engine.h with -DUR5_EMUL so that the 64 lanes of
// each wavefront run sequentially on the host. Lets `pytest -m "not gpu"` exercise the kernel source against the oracle
// without a GPU. Built into tests/emul/_build/ by tests/conftest.py; the package loader (mujoco_rl_ur5_amd/native.py)
// only ever opens csrc/libur5sim.so, so this can never stand in for the HIP path.
// Like the product library it is two translation units: this one (small scenes) and ur5sim_emul_many.cpp (-DUR5_MANY).
#define UR5_EMUL 1
#include <cstdlib>
#include "../../mujoco_rl_ur5_amd/csrc/ur5_engine.h"
#include "../../mujoco_rl_ur5_amd/csrc/ur5sim_host.h"

static int be_open(ur5_sim*, int) { return 0; }
static void be_close(ur5_sim*) {}
// Device memory and LDS are NOT zero on the GPU (a kernel starts with whatever the previous one left in the CU's LDS; hipMalloc returns
// recycled memory): poison both with 0xFF bytes (NaN doubles, -1 ints) so that a read-before-write in the engine shows up in the CPU suite.
static void* be_alloc(ur5_sim*, size_t bytes) { void* p = malloc(bytes); if (p) memset(p, 0xFF, bytes); return p; }
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

template <int NV> static void run_all(ur5_sim* h, const Ur5Launch& P) {
  typedef ur5::Lds<double, NV> L;
  L* lds = new L();
  for (int i = 0; i < h->n; i++) {
    const int e = P.order ? P.order[i] : i;
    if (P.op == UR5_OP_STAY && P.max_steps[e] <= 0) continue;   // as in ur5_run_kernel: a stay of zero chunks touches nothing
    memset((void*)lds, 0xFF, sizeof(L));
    ur5_emul_lds = lds;
    ur5::Engine<double, NV> eng(h->dm);
    eng.load(h->d_rec + (size_t)e * UR5_REC_STRIDE, P.pid_dt, P.contacts_enabled, P.step_cap ? P.step_cap[e] : 0x7fffffff);
#ifdef UR5_MANY
    eng.set_hess(P.hess + (size_t)e * UR5_HESS_STRIDE);
#endif
    eng.run(P, e);
    eng.save(h->d_rec + (size_t)e * UR5_REC_STRIDE);
  }
  delete lds;
}
static int be_render(ur5_sim* h, int cam, int W, int Hh, int mode, uint8_t* rgb, float* depth) {
  static float bp[UR5_MAXB][12], gp[UR5_R_MAXG][12];
  for (int e = 0; e < h->n; e++) {
    ur5r::body_poses(*h->dm, h->d_rec + (size_t)e * UR5_REC_STRIDE, bp);
    for (int g = 0; g < h->d_rm->ngeom; g++) ur5r::geom_pose(*h->d_rm, *h->dm, bp, g, gp[g]);
    for (int py = 0; py < Hh; py++) for (int px = 0; px < W; px++) {
      size_t o = ((size_t)e * Hh + py) * W + px;
      float z = ur5r::shade_pixel(*h->d_rm, gp, cam, W, Hh, px, py, rgb + 3 * o);
      depth[o] = mode == 0 ? z : ur5r::gl_depth(*h->d_rm, z);
    }
  }
  return 0;
}
static bool be_can_observe(ur5_sim* h) {
#ifdef UR5_MANY
  (void)h; return ur5::Engine<double, UR5_MAXNV>::CAN_OBSERVE;
#else
  return h->nvt == 32 ? ur5::Engine<double, 32>::CAN_OBSERVE : ur5::Engine<double, UR5_MAXNV>::CAN_OBSERVE;
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
