// ur5sim.hip -- HIP backend of libur5sim.so (gfx950 / MI355X): kernels + device plumbing behind include/ur5sim.h.
//
// One 64-thread workgroup == one wavefront == one scene (SURVEY.md section 2.1 launch shape). The whole scripted
// operation (a move, a stay, a 12-phase grasp attempt = thousands of 2 ms physics steps) runs inside ONE launch with the
// scene resident in LDS; HBM sees one coalesced record read and one record write per scene per launch.
#include <hip/hip_runtime.h>
#include "ur5_engine.h"
#include "ur5sim_host.h"

template <int NV>
__global__ void __launch_bounds__(64) ur5_run_kernel(double* __restrict__ rec, Ur5Launch P) {
  const int env = blockIdx.x;
  if (env >= P.n_env) return;
  ur5::Engine<double, NV> eng;
  double* r = rec + (size_t)env * UR5_REC_STRIDE;
  eng.load(r, P.pid_dt, P.contacts_enabled);
  eng.run(P, env);
  eng.save(r);
}

// the model sits in __constant__ memory (one copy per device); a handle re-uploads it only when another handle used the device last
static ur5_sim* g_model_owner[64] = {nullptr};

struct HipBackend {
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
};
#define HIPCHK(call)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) return ur5host::fail(UR5_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)

static int be_open(ur5_sim* h, int device_id) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return ur5host::fail(UR5_ERR_NOGPU, "no HIP device visible: libur5sim.so runs on an MI355X (gfx950) only, there is no CPU fallback");
  if (device_id < 0 || device_id >= count) return ur5host::fail(UR5_ERR_ARG, "device_id out of range");
  HIPCHK(hipSetDevice(device_id));
  HipBackend* b = new HipBackend();
  h->be = b;
  HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&b->ev0));
  HIPCHK(hipEventCreate(&b->ev1));
  return 0;
}
static void be_close(ur5_sim* h) {
  HipBackend* b = (HipBackend*)h->be;
  if (g_model_owner[h->device & 63] == h) g_model_owner[h->device & 63] = nullptr;
  if (!b) return;
  (void)hipSetDevice(h->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
  h->be = nullptr;
}
static void* be_alloc(ur5_sim* h, size_t bytes) {
  void* p = nullptr;
  (void)hipSetDevice(h->device);
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  (void)hipMemset(p, 0, bytes);
  return p;
}
static void be_free(ur5_sim* h, void* p) { (void)hipSetDevice(h->device); (void)hipFree(p); }
static int be_h2d(ur5_sim* h, void* dst, const void* src, size_t bytes) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}
static int be_d2h(ur5_sim* h, void* dst, const void* src, size_t bytes) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return 0;
}
static int be_launch(ur5_sim* h, const Ur5Launch& P) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  if (g_model_owner[h->device & 63] != h) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(ur5_cmodel), &h->hm, sizeof(Ur5DevModel), 0, hipMemcpyHostToDevice));
    g_model_owner[h->device & 63] = h;
  }
  HIPCHK(hipEventRecord(b->ev0, b->stream));
  dim3 grid(h->n), block(64);
  if (h->nvt == 32) hipLaunchKernelGGL(ur5_run_kernel<32>, grid, block, sizeof(ur5::Lds<double, 32>), b->stream, h->d_rec, P);
  else hipLaunchKernelGGL(ur5_run_kernel<UR5_MAXNV>, grid, block, sizeof(ur5::Lds<double, UR5_MAXNV>), b->stream, h->d_rec, P);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(b->ev1, b->stream));
  b->timed = true;
  return 0;
}
static int be_sync(ur5_sim* h) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (b->timed) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    h->last_ms = ms;
    b->timed = false;
  }
  return 0;
}
