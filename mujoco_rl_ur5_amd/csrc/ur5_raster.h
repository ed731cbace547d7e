// ur5_raster.h -- RGB-D observation of a scene: replaces sim.render(width, height, camera_name, depth=True)
// + the two flips of MJ_Controller.get_image_data (gym_grasper/controller/MujocoController.py:708-727) and, in metric
// mode, depth_2_meters (:729-740).
//
// The camera is fixed and the scene is a handful of convex shapes (plane, boxes, spheres, cylinders, capsules, convex hulls of the meshes, as
// MuJoCo's collision geometry -- capped hulls), so the image is produced by casting one ray per pixel against
// every geom (bounding-sphere reject first) instead of rasterising ~190 k visual triangles: one thread = one pixel, one block =
// one 16x16 tile of one scene, geom poses of the scene staged in LDS by the block. Shading: flat albedo (geom rgba / material
// colour), Lambert term from one directional light + ambient; no textures, shadows or reflections.
// fp32: depth errors stay below 1e-5 m at these distances.
#pragma once
#include <math.h>
#include <stdint.h>
#include "ur5_devmodel.h"

#define UR5_R_MAXG 80
#define UR5_R_MAXPL 1024
#define UR5_R_MAXCAM 4

struct Ur5RenderModel {
  int ngeom, nplane, ncam, pad;
  int g_type[UR5_R_MAXG], g_kind[UR5_R_MAXG], g_owner[UR5_R_MAXG], g_padr[UR5_R_MAXG], g_pnum[UR5_R_MAXG];
  float g_size[UR5_R_MAXG][3], g_pos[UR5_R_MAXG][3], g_mat[UR5_R_MAXG][9], g_rgba[UR5_R_MAXG][4], g_rbound[UR5_R_MAXG];
  float plane[UR5_R_MAXPL][4];
  float cam_pos[UR5_R_MAXCAM][3], cam_mat[UR5_R_MAXCAM][9], cam_fovy[UR5_R_MAXCAM];
  float znear, zfar;          // already multiplied by stat.extent
  float light[3];             // unit vector towards the light
  float sky[3];
};

#ifdef UR5_EMUL
#define UR5_RFN inline
#else
#define UR5_RFN __device__ __forceinline__
#endif

namespace ur5r {

struct F3 { float x, y, z; };
UR5_RFN F3 f3(float a, float b, float c) { F3 r; r.x = a; r.y = b; r.z = c; return r; }
UR5_RFN F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
UR5_RFN F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
UR5_RFN F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
UR5_RFN float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
UR5_RFN F3 mulm(const float* m, F3 v) { return f3(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[3] * v.x + m[4] * v.y + m[5] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z); }
UR5_RFN F3 mulmT(const float* m, F3 v) { return f3(m[0] * v.x + m[3] * v.y + m[6] * v.z, m[1] * v.x + m[4] * v.y + m[7] * v.z, m[2] * v.x + m[5] * v.y + m[8] * v.z); }

// world poses of the engine's cbodies (robot weld groups, then objects) from a state record; out: [UR5_MAXB][12] = pos, mat
// robot weld groups: a serial chain (fp64, as the engine's kinematics); objects: independent of each other, one per caller (the pose kernel spreads them over lanes)
UR5_RFN void robot_poses(const Ur5DevModel& M, const double* rec, float (*out)[12]) {
  double pos[UR5_MAXRD][3], quat[UR5_MAXRD][4];
  for (int d = 0; d < M.nrd; d++) {
    int p = M.rd_parent[d];
    double pp[3] = {0, 0, 0}, pq[4] = {1, 0, 0, 0};
    if (p >= 0) { for (int k = 0; k < 3; k++) pp[k] = pos[p][k]; for (int k = 0; k < 4; k++) pq[k] = quat[p][k]; }
    auto qmat = [](const double* q, double* m) {
      double w = q[0], x = q[1], y = q[2], z = q[3];
      m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
      m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
      m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
    };
    auto qmul = [](const double* a, const double* b, double* r) {
      double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
      double t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
      r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
    };
    double m[9], q[4], ps[3];
    qmat(pq, m);
    for (int i = 0; i < 3; i++) ps[i] = pp[i] + m[3 * i] * M.rd_pos[d][0] + m[3 * i + 1] * M.rd_pos[d][1] + m[3 * i + 2] * M.rd_pos[d][2];
    qmul(pq, M.rd_quat[d], q);
    qmat(q, m);
    double anc[3];
    for (int i = 0; i < 3; i++) anc[i] = ps[i] + m[3 * i] * M.rd_jpos[d][0] + m[3 * i + 1] * M.rd_jpos[d][1] + m[3 * i + 2] * M.rd_jpos[d][2];
    double a = 0.5 * (rec[UR5_REC_QPOS + d] - M.rd_qpos0[d]), sn = sin(a), cs = cos(a);
    double jq[4] = {cs, M.rd_jaxis[d][0] * sn, M.rd_jaxis[d][1] * sn, M.rd_jaxis[d][2] * sn};
    qmul(q, jq, q);
    double nn = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int k = 0; k < 4; k++) q[k] *= nn;
    qmat(q, m);
    for (int i = 0; i < 3; i++) ps[i] = anc[i] - (m[3 * i] * M.rd_jpos[d][0] + m[3 * i + 1] * M.rd_jpos[d][1] + m[3 * i + 2] * M.rd_jpos[d][2]);
    for (int k = 0; k < 3; k++) { pos[d][k] = ps[k]; out[d][k] = (float)ps[k]; }
    for (int k = 0; k < 4; k++) quat[d][k] = q[k];
    for (int k = 0; k < 9; k++) out[d][3 + k] = (float)m[k];
  }
}
UR5_RFN void object_pose(const Ur5DevModel& M, const double* rec, int k, float (*out)[12]) {
  {
    const double* qp = rec + UR5_REC_QPOS + M.nrd + 7 * k;
    double w = qp[3], x = qp[4], y = qp[5], z = qp[6];
    double nn = 1.0 / sqrt(w * w + x * x + y * y + z * z);
    w *= nn; x *= nn; y *= nn; z *= nn;
    float* o = out[M.nrd + k];
    for (int a = 0; a < 3; a++) o[a] = (float)(qp[a] + (M.obj_kind[k] == 0 ? M.obj_pos0[k][a] : 0.0));
    o[3] = (float)(w * w + x * x - y * y - z * z); o[4] = (float)(2 * (x * y - w * z)); o[5] = (float)(2 * (x * z + w * y));
    o[6] = (float)(2 * (x * y + w * z)); o[7] = (float)(w * w - x * x + y * y - z * z); o[8] = (float)(2 * (y * z - w * x));
    o[9] = (float)(2 * (x * z - w * y)); o[10] = (float)(2 * (y * z + w * x)); o[11] = (float)(w * w - x * x - y * y + z * z);
  }
}

UR5_RFN void body_poses(const Ur5DevModel& M, const double* rec, float (*out)[12]) {
  robot_poses(M, rec, out);
  for (int k = 0; k < M.nobj; k++) object_pose(M, rec, k, out);
}

// world pose of render geom g given the body poses
UR5_RFN void geom_pose(const Ur5RenderModel& R, const Ur5DevModel& M, const float (*bp)[12], int g, float* gp /* [12] */) {
  int kind = R.g_kind[g];
  if (kind == UR5_KIND_STATIC) {
    for (int k = 0; k < 3; k++) gp[k] = R.g_pos[g][k];
    for (int k = 0; k < 9; k++) gp[3 + k] = R.g_mat[g][k];
    return;
  }
  const float* b = bp[kind == UR5_KIND_ROBOT ? R.g_owner[g] : M.nrd + R.g_owner[g]];
  F3 p = f3(b[0], b[1], b[2]) + mulm(b + 3, f3(R.g_pos[g][0], R.g_pos[g][1], R.g_pos[g][2]));
  gp[0] = p.x; gp[1] = p.y; gp[2] = p.z;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) gp[3 + 3 * i + j] = b[3 + 3 * i] * R.g_mat[g][j] + b[3 + 3 * i + 1] * R.g_mat[g][3 + j] + b[3 + 3 * i + 2] * R.g_mat[g][6 + j];
}

// One pixel of the FINAL image (after the reference's flipud + fliplr): x right-to-left, y as in pixel_2_world
// (MujocoController.py:783-806). Returns depth along the optical axis in metres (zfar when nothing is hit) and an RGB triple.
// `list` / `nlist` (optional): the geoms to test, in ascending order (a tile's survivors of geom_screen_box); NULL = every geom.
UR5_RFN float shade_pixel(const Ur5RenderModel& R, const float (*gposes)[12], int cam, int W, int H, int px, int py, uint8_t* rgb,
                          const short* list = nullptr, int nlist = 0) {
  const float f = 0.5f * (float)H / tanf(R.cam_fovy[cam] * 3.14159265358979f / 360.0f);
  // GL pixel (i, j): i = W-1-px, j = H-1-py (the two flips); camera looks along -Z, +Y up, +X right
  const float xc = ((float)(W - 1 - px) + 0.5f - 0.5f * (float)W) / f, yc = ((float)(H - 1 - py) + 0.5f - 0.5f * (float)H) / f;
  const F3 o = f3(R.cam_pos[cam][0], R.cam_pos[cam][1], R.cam_pos[cam][2]);
  const F3 d = mulm(R.cam_mat[cam], f3(xc, yc, -1.0f));   // not normalised: the ray parameter t IS the depth along the axis
  const float dd = dot(d, d);
  float tbest = R.zfar;
  F3 nbest = f3(0, 0, 1);
  int gbest = -1;
  const int ntest = list ? nlist : R.ngeom;
  for (int gi = 0; gi < ntest; gi++) {
    const int g = list ? (int)list[gi] : gi;
    const float* gp = gposes[g];
    const F3 c = f3(gp[0], gp[1], gp[2]);
    const int type = R.g_type[g];
    if (type != UR5_GEOM_PLANE) {   // bounding sphere
      F3 oc = c - o;
      float tc = dot(oc, d) / dd;
      F3 perp = oc - d * tc;
      float rb = R.g_rbound[g];
      if (dot(perp, perp) > rb * rb || tc - rb > tbest) continue;
    }
    const F3 ol = mulmT(gp + 3, o - c), dl = mulmT(gp + 3, d);
    float t = -1.0f;
    F3 nl = f3(0, 0, 1);
    if (type == UR5_GEOM_PLANE) {
      if (dl.z < -1e-9f) { t = -ol.z / dl.z; nl = f3(0, 0, 1); }
    } else if (type == UR5_GEOM_SPHERE) {
      float r = R.g_size[g][0], b = dot(ol, dl), cc = dot(ol, ol) - r * r, disc = b * b - dd * cc;
      if (disc >= 0) { t = (-b - sqrtf(disc)) / dd; F3 p = ol + dl * t; nl = p * (1.0f / r); }
    } else if (type == UR5_GEOM_BOX) {
      float t0 = -1e30f, t1 = 1e30f;
      int ax = 0; float sg = 1;
      const float s[3] = {R.g_size[g][0], R.g_size[g][1], R.g_size[g][2]};
      const float oo[3] = {ol.x, ol.y, ol.z}, dv[3] = {dl.x, dl.y, dl.z};
      bool miss = false;
      for (int k = 0; k < 3; k++) {
        if (fabsf(dv[k]) < 1e-12f) { if (fabsf(oo[k]) > s[k]) miss = true; continue; }
        float ta = (-s[k] - oo[k]) / dv[k], tb = (s[k] - oo[k]) / dv[k];
        float tn = ta < tb ? ta : tb, tf = ta < tb ? tb : ta;
        if (tn > t0) { t0 = tn; ax = k; sg = dv[k] > 0 ? -1.0f : 1.0f; }
        if (tf < t1) t1 = tf;
      }
      if (!miss && t0 <= t1 && t0 > 0) { t = t0; nl = f3(ax == 0 ? sg : 0, ax == 1 ? sg : 0, ax == 2 ? sg : 0); }
    } else if (type == UR5_GEOM_CYLINDER || type == UR5_GEOM_CAPSULE) {
      // axis = local z, radius size[0], half length size[1]. Cylinder: (infinite side) n (slab |z| <= h), entry = later of the
      // two entries. Capsule: side hit inside the slab, else the entering root of the end sphere the ray meets beyond it.
      const float r = R.g_size[g][0], hl = R.g_size[g][1];
      const float a = dl.x * dl.x + dl.y * dl.y, b = ol.x * dl.x + ol.y * dl.y, cc = ol.x * ol.x + ol.y * ol.y - r * r;
      float ts0 = -1e30f, ts1 = 1e30f;
      bool miss = false;
      if (a > 1e-12f) { float disc = b * b - a * cc; if (disc < 0) miss = true; else { float sq = sqrtf(disc); ts0 = (-b - sq) / a; ts1 = (-b + sq) / a; } }
      else if (cc > 0) miss = true;
      if (!miss && type == UR5_GEOM_CYLINDER) {
        float tz0 = -1e30f, tz1 = 1e30f;
        if (fabsf(dl.z) > 1e-12f) { float ta = (-hl - ol.z) / dl.z, tb = (hl - ol.z) / dl.z; tz0 = ta < tb ? ta : tb; tz1 = ta < tb ? tb : ta; }
        else if (fabsf(ol.z) > hl) miss = true;
        float te = ts0 > tz0 ? ts0 : tz0, tx = ts1 < tz1 ? ts1 : tz1;
        if (!miss && te <= tx && te > 0) {
          t = te;
          if (ts0 > tz0) { F3 p = ol + dl * t; nl = f3(p.x / r, p.y / r, 0); } else nl = f3(0, 0, dl.z > 0 ? -1.0f : 1.0f);
        }
      } else if (type == UR5_GEOM_CAPSULE) {
        float tb2 = 1e30f;
        if (!miss && a > 1e-12f) { float z = ol.z + dl.z * ts0; if (fabsf(z) <= hl && ts0 > 0) { tb2 = ts0; F3 p = ol + dl * ts0; nl = f3(p.x / r, p.y / r, 0); } }
        for (int e = 0; e < 2; e++) {
          const float zc = e == 0 ? hl : -hl;
          const F3 oc = f3(ol.x, ol.y, ol.z - zc);
          float bs = dot(oc, dl), cs = dot(oc, oc) - r * r, disc = bs * bs - dd * cs;
          if (disc < 0) continue;
          float ts = (-bs - sqrtf(disc)) / dd;
          F3 p = oc + dl * ts;
          if (ts > 0 && ts < tb2 && (e == 0 ? p.z >= 0 : p.z <= 0)) { tb2 = ts; nl = p * (1.0f / r); }
        }
        if (tb2 < 1e29f) t = tb2;
      }
    } else if (type == UR5_GEOM_MESH) {
      float t0 = -1e30f, t1 = 1e30f;
      int kb = -1;
      bool miss = false;
      // the hull's planes, four at a time: the loads of a chunk are issued together, then the planes are clipped in order with the same early exits -- the same
      // arithmetic in the same order as one plane per trip, but a quarter of the exposed memory latencies (a hull has 60 - 400 planes; inside a launch a scene's
      // few wavefronts have nothing to hide them behind: Engine::observe)
      const int np = R.g_pnum[g], pa = R.g_padr[g];
      for (int k0 = 0; k0 < np && !miss; k0 += 4) {
        float pl[4][4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float* src = R.plane[pa + (k0 + j < np ? k0 + j : np - 1)];
          pl[j][0] = src[0]; pl[j][1] = src[1]; pl[j][2] = src[2]; pl[j][3] = src[3];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int k = k0 + j;
          if (k >= np || miss) break;
          float nd = pl[j][0] * dl.x + pl[j][1] * dl.y + pl[j][2] * dl.z, no = pl[j][0] * ol.x + pl[j][1] * ol.y + pl[j][2] * ol.z + pl[j][3];
          if (fabsf(nd) < 1e-12f) { if (no > 0) miss = true; continue; }
          float tt = -no / nd;
          if (nd < 0) { if (tt > t0) { t0 = tt; kb = k; } } else if (tt < t1) t1 = tt;
          if (t0 > t1) miss = true;
        }
      }
      if (!miss && kb >= 0 && t0 > 0) { t = t0; const float* pl = R.plane[R.g_padr[g] + kb]; nl = f3(pl[0], pl[1], pl[2]); }
    }
    if (t > R.znear && t < tbest) { tbest = t; nbest = mulm(gp + 3, nl); gbest = g; }
  }
  if (gbest < 0) { rgb[0] = (uint8_t)(255.0f * R.sky[0]); rgb[1] = (uint8_t)(255.0f * R.sky[1]); rgb[2] = (uint8_t)(255.0f * R.sky[2]); return R.zfar; }
  float ln = nbest.x * R.light[0] + nbest.y * R.light[1] + nbest.z * R.light[2];
  float sh = 0.35f + 0.65f * (ln > 0 ? ln : 0);
  for (int k = 0; k < 3; k++) {
    float v = 255.0f * R.g_rgba[gbest][k] * sh + 0.5f;
    rgb[k] = (uint8_t)(v > 255.0f ? 255.0f : v);
  }
  return tbest;
}

// Conservative pixel box [x0, x1] x [y0, y1] (final image coordinates, inclusive, clipped) that contains every pixel whose ray can meet
// geom g's bounding sphere; box[0] > box[1] = off screen. Planes and spheres that reach the near plane cover the whole image.
UR5_RFN void geom_screen_box(const Ur5RenderModel& R, const float* gp, int g, int cam, int W, int H, short* box) {
  box[0] = 0; box[1] = (short)(W - 1); box[2] = 0; box[3] = (short)(H - 1);
  if (R.g_type[g] == UR5_GEOM_PLANE) return;
  const float f = 0.5f * (float)H / tanf(R.cam_fovy[cam] * 3.14159265358979f / 360.0f);
  const F3 pc = mulmT(R.cam_mat[cam], f3(gp[0] - R.cam_pos[cam][0], gp[1] - R.cam_pos[cam][1], gp[2] - R.cam_pos[cam][2]));
  const float r = R.g_rbound[g] * 1.0001f + 1e-6f, D = -pc.z;
  if (D - r <= R.znear) { if (D + r <= 0) { box[0] = 1; box[1] = 0; } return; }       // behind the camera entirely / straddles the near plane
  const float xlo = (pc.x - r) / (pc.x - r <= 0 ? D - r : D + r), xhi = (pc.x + r) / (pc.x + r >= 0 ? D - r : D + r);
  const float ylo = (pc.y - r) / (pc.y - r <= 0 ? D - r : D + r), yhi = (pc.y + r) / (pc.y + r >= 0 ? D - r : D + r);
  // GL pixel i = f xc + W/2 - 1/2 ; final pixel px = W - 1 - i (the two flips of get_image_data); one pixel of slack on each side
  const float ilo = f * xlo + 0.5f * (float)W - 0.5f, ihi = f * xhi + 0.5f * (float)W - 0.5f;
  const float jlo = f * ylo + 0.5f * (float)H - 0.5f, jhi = f * yhi + 0.5f * (float)H - 0.5f;
  float x0 = (float)(W - 1) - ihi - 1.0f, x1 = (float)(W - 1) - ilo + 1.0f, y0 = (float)(H - 1) - jhi - 1.0f, y1 = (float)(H - 1) - jlo + 1.0f;
  x0 = floorf(x0); y0 = floorf(y0); x1 = ceilf(x1); y1 = ceilf(y1);
  if (x1 < 0 || y1 < 0 || x0 > (float)(W - 1) || y0 > (float)(H - 1)) { box[0] = 1; box[1] = 0; return; }
  box[0] = (short)(x0 < 0 ? 0 : x0); box[1] = (short)(x1 > (float)(W - 1) ? (float)(W - 1) : x1);
  box[2] = (short)(y0 < 0 ? 0 : y0); box[3] = (short)(y1 > (float)(H - 1) ? (float)(H - 1) : y1);
}

// window-space depth in [0, 1] as sim.render returns it, so that depth_2_meters (MujocoController.py:737-740) inverts it
UR5_RFN float gl_depth(const Ur5RenderModel& R, float z) { return (1.0f - R.znear / z) / (1.0f - R.znear / R.zfar); }

}  // namespace ur5r
