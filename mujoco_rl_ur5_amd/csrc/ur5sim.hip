// ur5sim.hip -- HIP backend of libur5sim.so (gfx950 / MI355X): kernels + device plumbing behind include/ur5sim.h.
//
// One 64-thread workgroup == one wavefront == one scene (SURVEY.md section 2.1 launch shape). The whole scripted
// operation (a move, a stay, a 12-phase grasp attempt = thousands of 2 ms physics steps) runs inside ONE launch with the
// scene resident in LDS; HBM sees one coalesced record read and one record write per scene per launch.
#include <hip/hip_runtime.h>
#include "ur5_engine.h"
#include "ur5sim_host.h"

// Register budget. Small-scene unit: one scene per wavefront, two wavefronts per SIMD at 256 registers (8 scenes per CU, the LDS limit): the two resident waves hide each
// other's latencies almost perfectly (a wave alone takes 68 us per settled step, 8 per CU 90 us each). A 168-register cap for a third wave costs 26 %.
// Pile unit: two scenes per CU need <= 256 registers per lane and an LDS image <= 80 KB; the cap alone was +-0 % (profiles/r04_d_ab_many_register_cap.log).
#ifdef UR5_MANY
#define UR5_KERNEL_ATTR(GS) __launch_bounds__(UR5_NT) __attribute__((amdgpu_waves_per_eu(2, 2)))
static_assert(2 * sizeof(ur5::Lds<double, UR5_MAXNV>) <= 160 * 1024, "two 40-object scenes per CU: the pile's LDS image must stay below 80 KB");
#else
#define UR5_KERNEL_ATTR(GS) __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
#ifndef UR5_PROFILE   // (the per-phase cycle accounting build adds its counters to the image)
#ifndef UR5_LDS_PAD   // (the residency probe pads the image on purpose)
static_assert(8 * sizeof(ur5::Lds<double, 32>) <= 160 * 1024, "the IT1 scene image must leave room for 8 scenes per CU (2 waves per SIMD): LDS is what caps residency");
static_assert(7 * sizeof(ur5::Lds<double, 44>) <= 160 * 1024 && sizeof(ur5::Lds<double, 44>) <= 18 * 1280, "the six-object image: 7 scenes per CU = 18 of the 128 LDS granules of 1 280 B (profiles/r04_z_lds_residency.log)");
#endif
#endif
#endif
template <int NV, int GS>
__global__ void UR5_KERNEL_ATTR(GS) ur5_run_kernel(double* __restrict__ rec, Ur5Launch P, const Ur5DevModel* __restrict__ model) {
  const int slot = blockIdx.x * (UR5_NT / GS) + (int)threadIdx.x / GS;
  const bool present = slot < P.n_env;             // a half-filled last workgroup: the lanes of the missing scene idle
  // workgroups are dispatched in blockIdx order: a caller that knows which scenes have the most work ahead of them (an episode reset to
  // settle first, a box to carry) lists those first, so that the short ones fill the tail of the launch
  const int env = (present && P.order) ? P.order[slot] : slot;
  // a stay of zero chunks (the unflagged scenes of ur5_reset_dev) touches nothing: the record, last_movement_steps included, stays as it is
  const bool live = present && !(P.op == UR5_OP_STAY && P.max_steps[env] <= 0);
  ur5::Engine<double, NV, GS> eng(model);
  double* r = rec + (size_t)(live ? env : 0) * UR5_REC_STRIDE;
  if (live) eng.load(r, P.pid_dt, P.contacts_enabled, P.step_cap ? P.step_cap[env] : 0x7fffffff);
#ifdef UR5_MANY
  eng.set_hess(P.hess + (size_t)env * UR5_HESS_STRIDE);
#endif
  eng.run(P, env, live);
  if (live) eng.save(r);
}

// RGB-D observation in two launches. (1) ur5_render_pose_kernel, one 64-thread block per scene: the (serial, fp64) forward kinematics of
// the scene ONCE, then every render geom's world pose and its conservative screen box. (2) ur5_render_kernel, one block per 16x16 pixel tile
// of one scene: the tile stages the scene's geom poses in LDS, keeps (in geom order) only the geoms whose box meets the tile, and every
// thread casts the ray of its pixel against that short list.
struct Ur5GeomPose { float p[12]; short box[4]; };
__global__ void __launch_bounds__(64) ur5_render_pose_kernel(const Ur5DevModel* __restrict__ Mp, const Ur5RenderModel* __restrict__ R, const double* __restrict__ rec, int n, int cam, int W, int H,
                                                             Ur5GeomPose* __restrict__ gpose) {
  __shared__ float bp[UR5_MAXB][12];
  const int scene = blockIdx.x;
  if (scene >= n) return;
  const double* r = rec + (size_t)scene * UR5_REC_STRIDE;
  if (threadIdx.x == 0) ur5r::robot_poses(*Mp, r, bp);                                            // a serial chain: one lane
  else for (int k = (int)threadIdx.x - 1; k < Mp->nobj; k += (int)blockDim.x - 1) ur5r::object_pose(*Mp, r, k, bp);   // the objects: the other lanes
  __syncthreads();
  for (int g = threadIdx.x; g < R->ngeom; g += blockDim.x) {
    Ur5GeomPose* o = gpose + (size_t)scene * UR5_R_MAXG + g;
    float gp[12];
    ur5r::geom_pose(*R, *Mp, bp, g, gp);
    ur5r::geom_screen_box(*R, gp, g, cam, W, H, o->box);
#pragma unroll
    for (int k = 0; k < 12; k++) o->p[k] = gp[k];
  }
}
__global__ void __launch_bounds__(256) ur5_render_kernel(const Ur5RenderModel* __restrict__ R, const Ur5GeomPose* __restrict__ gpose, int cam, int W, int H,
                                                         int mode, uint8_t* __restrict__ rgb, float* __restrict__ depth) {
  __shared__ float gp[UR5_R_MAXG][12];
  __shared__ short list[UR5_R_MAXG];
  __shared__ unsigned long long votes[4];
  const int scene = blockIdx.y, tiles_x = (W + 15) / 16;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int x0 = tx * 16, y0 = ty * 16, x1 = x0 + 15, y1 = y0 + 15;
  const int ngeom = R->ngeom;
  static_assert(UR5_R_MAXG <= 256, "one geom per thread");
  const int g = threadIdx.x;
  bool keep = false;
  if (g < ngeom) {
    const Ur5GeomPose* s = gpose + (size_t)scene * UR5_R_MAXG + g;
#pragma unroll
    for (int k = 0; k < 12; k++) gp[g][k] = s->p[k];
    keep = s->box[0] <= s->box[1] && s->box[0] <= x1 && s->box[1] >= x0 && s->box[2] <= y1 && s->box[3] >= y0;
  }
  const unsigned long long vote = __ballot(keep);
  if ((threadIdx.x & 63) == 0) votes[threadIdx.x >> 6] = vote;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) { const int c = __popcll(votes[w]); if (w < (int)(threadIdx.x >> 6)) before += c; total += c; }
  if (keep) list[before + __popcll(vote & ((1ull << (threadIdx.x & 63)) - 1ull))] = (short)g;
  __syncthreads();
  const int px = x0 + (threadIdx.x & 15), py = y0 + (threadIdx.x >> 4);
  if (px >= W || py >= H) return;
  uint8_t c[3];
  float z = ur5r::shade_pixel(*R, gp, cam, W, H, px, py, c, list, total);
  const size_t o = ((size_t)scene * H + py) * W + px;
  rgb[3 * o] = c[0]; rgb[3 * o + 1] = c[1]; rgb[3 * o + 2] = c[2];
  depth[o] = mode == 0 ? z : ur5r::gl_depth(*R, z);
}

// GraspEnv.reset_model for the flagged scenes, one thread per scene (ur5host::reset_record: the same code the host path runs);
// max_steps[e] = number of 10-step settle chunks the following stay launch gives scene e (0: the scene sits it out).
__global__ void __launch_bounds__(64) ur5_reset_kernel(const Ur5DevModel* __restrict__ Mp, double* __restrict__ rec, const double* __restrict__ qpos0, const uint64_t* __restrict__ seeds,
                                                       const uint8_t* __restrict__ mask, int n, int chunks, int* __restrict__ max_steps) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const bool on = !mask || mask[e];
  if (on) ur5host::reset_record(*Mp, qpos0, rec + (size_t)e * UR5_REC_STRIDE, seeds[e]);
  max_steps[e] = on ? chunks : 0;
}

// Every launch is bracketed by its own pair of HIP events on the handle's stream; ur5_sync resolves the pairs recorded since
// the previous sync (several launches may be queued: reset + settle + render + grasp attempt of one round).
struct HipBackend {
  hipStream_t stream = nullptr;    // the stream launches go to: `own` or a caller-owned one (ur5_set_stream)
  hipStream_t own = nullptr;
  std::vector<hipEvent_t> ev;      // pool: pairs [2k, 2k+1]
  int pending = 0;                 // pairs recorded since the last sync
};
static void be_resolve(ur5_sim* h, HipBackend* b) {   // the stream must be idle
  for (int k = 0; k < b->pending; k++) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, b->ev[2 * k], b->ev[2 * k + 1]) == hipSuccess) { h->last_ms = ms; h->kernel_ms_total += ms; }
  }
  b->pending = 0;
}
static int be_event_pair(ur5_sim* h, HipBackend* b, hipEvent_t* e0, hipEvent_t* e1) {
  if (b->pending >= 64) { (void)hipStreamSynchronize(b->stream); be_resolve(h, b); }   // a caller that never syncs must not grow the pool
  while ((int)b->ev.size() < 2 * (b->pending + 1)) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return -1;
    b->ev.push_back(e);
  }
  *e0 = b->ev[2 * b->pending]; *e1 = b->ev[2 * b->pending + 1];
  return 0;   // the pair counts (be_event_commit) only once BOTH events have been recorded: an error in between must not leave a half-recorded pair pending
}
static void be_event_commit(HipBackend* b) { b->pending++; }
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
  HIPCHK(hipStreamCreateWithFlags(&b->own, hipStreamNonBlocking));
  b->stream = b->own;
  return 0;
}
static void be_close(ur5_sim* h) {
  HipBackend* b = (HipBackend*)h->be;
  if (!b) return;
  (void)hipSetDevice(h->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  for (hipEvent_t e : b->ev) (void)hipEventDestroy(e);
  if (b->own) (void)hipStreamDestroy(b->own);
  delete b;
  h->be = nullptr;
}
static int be_set_stream(ur5_sim* h, void* stream, int external) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(b->stream));   // nothing of ours may still be queued on the stream we leave
  be_resolve(h, b);
  b->stream = external ? (hipStream_t)stream : b->own;   // external NULL = the device's default stream (torch's default)
  return 0;
}
static void* be_alloc(ur5_sim* h, size_t bytes) {
  void* p = nullptr;
  (void)hipSetDevice(h->device);
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  // zero it ON THE HANDLE'S STREAM and wait: hipMemset runs on the null stream, which the handle's non-blocking stream does not synchronise
  // with -- an upload or a kernel queued right after the allocation could otherwise be overtaken by the fill
  HipBackend* b = (HipBackend*)h->be;
  hipStream_t s = b ? b->stream : nullptr;
  if (hipMemsetAsync(p, 0, bytes, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(p); return nullptr; }
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
static int be_d2d_async(ur5_sim* h, void* dst, const void* src, size_t bytes) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, b->stream));
  return 0;
}
static int be_launch(ur5_sim* h, const Ur5Launch& P) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  hipEvent_t ev0, ev1;
  if (be_event_pair(h, b, &ev0, &ev1)) return ur5host::fail(UR5_ERR_DEVICE, "hipEventCreate failed");
  HIPCHK(hipEventRecord(ev0, b->stream));
  dim3 block(UR5_NT);
#ifdef UR5_MANY
  dim3 grid(h->n);
  hipLaunchKernelGGL((ur5_run_kernel<UR5_MAXNV, UR5_NT>), grid, block, 0, b->stream, h->d_rec, P, (const Ur5DevModel*)h->dm);
#else
  if (h->nvt == 32) hipLaunchKernelGGL((ur5_run_kernel<32, 64>), dim3(h->n), block, 0, b->stream, h->d_rec, P, (const Ur5DevModel*)h->dm);
  else hipLaunchKernelGGL((ur5_run_kernel<UR5_MAXNV, 64>), dim3(h->n), block, 0, b->stream, h->d_rec, P, (const Ur5DevModel*)h->dm);
#endif
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ev1, b->stream));
  be_event_commit(b);
  return 0;
}
static bool be_can_observe(ur5_sim* h) {
#ifdef UR5_MANY
  (void)h; return ur5::Engine<double, UR5_MAXNV, UR5_NT>::CAN_OBSERVE;
#else
  return h->nvt == 32 ? ur5::Engine<double, 32, 64>::CAN_OBSERVE : ur5::Engine<double, UR5_MAXNV, 64>::CAN_OBSERVE;
#endif
}
static int be_reset_dev(ur5_sim* h, const uint64_t* seeds_dev, const uint8_t* mask_dev, int chunks, int* max_steps_dev) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  hipLaunchKernelGGL(ur5_reset_kernel, dim3((h->n + 63) / 64), dim3(64), 0, b->stream, (const Ur5DevModel*)h->dm, h->d_rec, h->d_qpos0, seeds_dev, mask_dev, h->n, chunks, max_steps_dev);
  HIPCHK(hipGetLastError());
  return 0;
}
static int be_render(ur5_sim* h, int cam, int W, int Hh, int mode, uint8_t* rgb_dev, float* depth_dev) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  if (!h->d_gpose) {
    h->d_gpose = be_alloc(h, (size_t)h->n * UR5_R_MAXG * sizeof(Ur5GeomPose));
    if (!h->d_gpose) return ur5host::fail(UR5_ERR_DEVICE, "device allocation failed (render geom poses)");
  }
  hipEvent_t ev0, ev1;
  if (be_event_pair(h, b, &ev0, &ev1)) return ur5host::fail(UR5_ERR_DEVICE, "hipEventCreate failed");
  HIPCHK(hipEventRecord(ev0, b->stream));
  hipLaunchKernelGGL(ur5_render_pose_kernel, dim3(h->n), dim3(64), 0, b->stream, (const Ur5DevModel*)h->dm, h->d_rm, h->d_rec, h->n, cam, W, Hh, (Ur5GeomPose*)h->d_gpose);
  dim3 grid(((W + 15) / 16) * ((Hh + 15) / 16), h->n), block(256);
  hipLaunchKernelGGL(ur5_render_kernel, grid, block, 0, b->stream, h->d_rm, (const Ur5GeomPose*)h->d_gpose, cam, W, Hh, mode, rgb_dev, depth_dev);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(ev1, b->stream));
  be_event_commit(b);
  return 0;
}
static int be_sync(ur5_sim* h) {
  HipBackend* b = (HipBackend*)h->be;
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  be_resolve(h, b);
  return 0;
}
