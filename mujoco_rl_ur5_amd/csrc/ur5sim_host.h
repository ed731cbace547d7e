// ur5sim_host.h -- host half of libur5sim: model specialisation (blob -> Ur5DevModel), reset sampling and the C ABI
// of include/ur5sim.h. Backend-agnostic: the including translation unit supplies be_* hooks (HIP in ur5sim.hip; the
// test-only lane emulation in tests/emul/ur5sim_emul.cpp).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/ur5sim.h"
#include "../../include/ur5sim_test.h"
#include "ur5_devmodel.h"
#include "ur5_raster.h"

namespace ur5host {

static thread_local std::string g_err;
static thread_local bool g_err_many = false;   // small-scene unit: the last failing call was forwarded to the many-object unit
static int fail(int code, const std::string& msg) { g_err = msg; g_err_many = false; return code; }
// number of free-floating top-level bodies in a model blob (-1: unreadable; at least 7 when a collidable geom has condim 6) -- decides which engine variant serves it
static int count_objects(const void* data, size_t nbytes);

// ------------------------------------------------------------------ blob reader (format: mujoco_rl_ur5_amd/model.py)
struct Blob {
  const char* p;
  size_t n;
  bool ok() const { return n >= 16 && memcmp(p, "UR5MODL1", 8) == 0; }
  const void* find(const char* name, int code, int* count) const {
    uint64_t ns;
    memcpy(&ns, p + 8, 8);
    size_t off = 16;
    for (uint64_t i = 0; i < ns && off + 40 <= n; i++) {
      char nm[33];
      memcpy(nm, p + off, 32);
      nm[32] = 0;
      uint32_t c, cnt;
      memcpy(&c, p + off + 32, 4);
      memcpy(&cnt, p + off + 36, 4);
      off += 40;
      size_t bytes = c == 0 ? 8ull * cnt : (c == 1 ? 4ull * cnt : cnt);
      if (!strcmp(nm, name) && (int)c == code) { *count = (int)cnt; return p + off; }
      off += bytes + (8 - bytes % 8) % 8;
    }
    *count = -1;
    return nullptr;
  }
  const double* F(const char* nm, int* c = nullptr) const { int k; auto r = (const double*)find(nm, 0, &k); if (c) *c = k; return r; }
  const int* I(const char* nm, int* c = nullptr) const { int k; auto r = (const int*)find(nm, 1, &k); if (c) *c = k; return r; }
};

// ------------------------------------------------------------------ tiny rigid-transform helpers (double)
static int count_objects(const void* data, size_t nbytes) {
  Blob B{(const char*)data, nbytes};
  if (!B.ok()) return -1;
  int nbody = 0;
  const int* tree = B.I("body_treeid", &nbody);
  if (!tree) return -1;
  int n = 0;
  for (int b = 1; b < nbody; b++) if (tree[b] > 0) n++;
  // condim 6 (rolling friction, UR5gripper_2_finger_many_objects.xml:29) needs the six base directions of the many-object engine, however few objects
  // the scene has (single-object scenes of that file are how the per-shape grasp table of tools/ is made)
  int ng = 0;
  const int* condim = B.I("geom_condim", &ng);
  const int* collide = B.I("geom_collide");
  for (int g = 0; condim && g < ng; g++) if (condim[g] > 4 && (!collide || collide[g])) return n > 7 ? n : 7;
  return n;
}
struct Xf { double p[3]; double q[4]; };
static void qmul(const double* a, const double* b, double* r) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof t);
}
static void qmat(const double* q, double* m) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void mv(const double* m, const double* v, double* r) {
  double t[3] = {m[0] * v[0] + m[1] * v[1] + m[2] * v[2], m[3] * v[0] + m[4] * v[1] + m[5] * v[2], m[6] * v[0] + m[7] * v[1] + m[8] * v[2]};
  memcpy(r, t, sizeof t);
}
static Xf compose(const Xf& a, const Xf& b) {  // a * b
  Xf r;
  double m[9], t[3];
  qmat(a.q, m);
  mv(m, b.p, t);
  for (int i = 0; i < 3; i++) r.p[i] = a.p[i] + t[i];
  qmul(a.q, b.q, r.q);
  return r;
}
static Xf identity() { Xf r = {{0, 0, 0}, {1, 0, 0, 0}}; return r; }

// ------------------------------------------------------------------ blob -> Ur5DevModel
static int build_model(const void* data, size_t nbytes, int ee_body, Ur5DevModel* out, std::vector<int>* dev2model_geom, Ur5RenderModel* RM) {
  Blob B{(const char*)data, nbytes};
  if (!B.ok()) return fail(UR5_ERR_MODEL, "model blob: bad magic");
  Ur5DevModel& D = *out;
  memset(&D, 0, sizeof D);
  int nbody, njnt, nv, nq, ngeom, npair, neq, nu, ntree;
  const int* body_parent = B.I("body_parentid", &nbody);
  const double *body_pos = B.F("body_pos"), *body_quat = B.F("body_quat");
  const int *body_jntadr = B.I("body_jntadr"), *body_jntnum = B.I("body_jntnum"), *body_weld = B.I("body_weldid"), *body_tree = B.I("body_treeid");
  const double *body_mass = B.F("body_mass"), *body_ipos = B.F("body_ipos"), *body_inertia = B.F("body_inertia"), *body_invw = B.F("body_invweight0");
  const int *jnt_type = B.I("jnt_type", &njnt), *jnt_qposadr = B.I("jnt_qposadr"), *jnt_dofadr = B.I("jnt_dofadr"), *jnt_body = B.I("jnt_bodyid");
  const int* jnt_limited = B.I("jnt_limited");
  const double *jnt_pos = B.F("jnt_pos"), *jnt_axis = B.F("jnt_axis"), *jnt_range = B.F("jnt_range"), *qpos0 = B.F("qpos0", &nq);
  const double *dof_arm = B.F("dof_armature", &nv), *dof_damp = B.F("dof_damping"), *dof_invw = B.F("dof_invweight0");
  const int* tree_dofadr = B.I("tree_dofadr", &ntree);
  const int *geom_type = B.I("geom_type", &ngeom), *geom_body = B.I("geom_bodyid"), *geom_condim = B.I("geom_condim"), *geom_mesh = B.I("geom_meshid"),
            *geom_collide = B.I("geom_collide");
  const double *geom_size = B.F("geom_size"), *geom_pos = B.F("geom_pos"), *geom_quat = B.F("geom_quat"), *geom_friction = B.F("geom_friction"),
               *geom_margin = B.F("geom_margin"), *geom_solref = B.F("geom_solref"), *geom_solimp = B.F("geom_solimp"), *geom_rbound = B.F("geom_rbound");
  const int *mesh_adr = B.I("mesh_vertadr"), *mesh_num = B.I("mesh_vertnum");
  const double* mesh_vert = B.F("mesh_vert");
  const int *pair1 = B.I("pair_geom1", &npair), *pair2 = B.I("pair_geom2");
  const int *eq1 = B.I("eq_jnt1", &neq), *eq2 = B.I("eq_jnt2");
  const double *eq_poly = B.F("eq_polycoef"), *eq_solref = B.F("eq_solref"), *eq_solimp = B.F("eq_solimp");
  const int *act_jnt = B.I("act_jntid", &nu), *act_lim = B.I("act_ctrllimited");
  const double *act_gear = B.F("act_gear"), *act_range = B.F("act_ctrlrange"), *optf = B.F("opt_f");
  const int* opti = B.I("opt_i");
  if (!body_parent || !qpos0 || !optf || !opti || !pair1) return fail(UR5_ERR_MODEL, "model blob: missing sections");
  (void)tree_dofadr; (void)geom_quat;

  // world frame of every body at qpos0 composition of fixed transforms (joints contribute nothing to the *local* frames we store)
  auto local = [&](int b) { Xf x; memcpy(x.p, body_pos + 3 * b, 24); memcpy(x.q, body_quat + 4 * b, 32); return x; };
  // ---- robot tree = tree 0: weld roots with exactly one hinge, dofs 0..nrd-1 in order
  std::vector<int> cb_of_body(nbody, -1);
  int nrd = 0;
  for (int b = 1; b < nbody; b++) {
    if (body_tree[b] != 0 || body_jntnum[b] == 0) continue;
    int j = body_jntadr[b];
    if (body_jntnum[b] != 1 || jnt_type[j] != 3) return fail(UR5_ERR_MODEL, "robot tree: every jointed body must carry exactly one hinge");
    if (jnt_dofadr[j] != nrd || jnt_qposadr[j] != nrd) return fail(UR5_ERR_MODEL, "robot tree: its dofs must come first and in body order");
    if (nrd >= UR5_MAXRD) return fail(UR5_ERR_MODEL, "robot tree: more than 8 hinges");
    cb_of_body[b] = nrd++;
  }
  if (nrd == 0) return fail(UR5_ERR_MODEL, "no robot tree found");
  D.nrd = nrd;
  std::vector<int> body_of_cb(nrd);
  for (int b = 0; b < nbody; b++) if (cb_of_body[b] >= 0) body_of_cb[cb_of_body[b]] = b;
  // frame of body b relative to its weld root (robot) or to the world (static)
  auto rel_to = [&](int b, int stop) {  // transform of b's frame in the frame of ancestor `stop` (stop = 0 -> world)
    Xf x = identity();
    std::vector<int> chain;
    for (int k = b; k != stop; k = body_parent[k]) chain.push_back(k);
    for (int i = (int)chain.size() - 1; i >= 0; i--) x = compose(x, local(chain[i]));
    return x;
  };
  for (int d = 0; d < nrd; d++) {
    int b = body_of_cb[d];
    int pb = body_parent[b];
    while (pb > 0 && cb_of_body[pb] < 0) pb = body_parent[pb];
    D.rd_parent[d] = pb > 0 ? cb_of_body[pb] : -1;
    Xf x = rel_to(b, pb > 0 ? pb : 0);
    memcpy(D.rd_pos[d], x.p, 24);
    memcpy(D.rd_quat[d], x.q, 32);
    qmat(x.q, D.rd_mat[d]);
    int j = body_jntadr[b];
    memcpy(D.rd_jpos[d], jnt_pos + 3 * j, 24);
    memcpy(D.rd_jaxis[d], jnt_axis + 3 * j, 24);
    D.rd_armature[d] = dof_arm[d]; D.rd_damping[d] = dof_damp[d]; D.rd_invweight[d] = dof_invw[d]; D.rd_qpos0[d] = qpos0[d];
    D.rd_limited[d] = jnt_limited[j]; D.rd_lo[d] = jnt_range[2 * j]; D.rd_hi[d] = jnt_range[2 * j + 1];
  }
  for (int d = 0; d < nrd; d++) {
    unsigned m = 0;
    for (int e = d; e >= 0; e = D.rd_parent[e]) m |= 1u << e;
    D.rd_anc[d] = m;
  }
  for (int d = 0; d < nrd; d++) {
    unsigned m = 0;
    for (int b = 0; b < nrd; b++) if (D.rd_anc[b] >> d & 1u) m |= 1u << b;
    D.rd_desc[d] = m;
  }
  // fold welded children into their weld root: mass, com and inertia in the root frame
  for (int d = 0; d < nrd; d++) {
    int root = body_of_cb[d];
    double mass = 0, com[3] = {0, 0, 0};
    std::vector<int> members;
    for (int b = 1; b < nbody; b++) if (body_weld[b] == root) members.push_back(b);
    for (int b : members) {
      Xf x = rel_to(b, root);
      double m9[9], c[3];
      qmat(x.q, m9);
      mv(m9, body_ipos + 3 * b, c);
      for (int i = 0; i < 3; i++) com[i] += body_mass[b] * (x.p[i] + c[i]);
      mass += body_mass[b];
    }
    if (mass <= 0) return fail(UR5_ERR_MODEL, "robot weld group without mass");
    for (int i = 0; i < 3; i++) com[i] /= mass;
    double I[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int b : members) {
      Xf x = rel_to(b, root);
      double R[9], c[3];
      qmat(x.q, R);
      mv(R, body_ipos + 3 * b, c);
      const double* bi = body_inertia + 6 * b;
      double Ib[9] = {bi[0], bi[3], bi[4], bi[3], bi[1], bi[5], bi[4], bi[5], bi[2]}, T[9], Iw[9];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += R[3 * i + k] * Ib[3 * k + j]; T[3 * i + j] = s; }
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += T[3 * i + k] * R[3 * j + k]; Iw[3 * i + j] = s; }
      double r[3] = {x.p[0] + c[0] - com[0], x.p[1] + c[1] - com[1], x.p[2] + c[2] - com[2]};
      double rr = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) I[3 * i + j] += Iw[3 * i + j] + body_mass[b] * ((i == j ? rr : 0) - r[i] * r[j]);
    }
    D.rd_mass[d] = mass;
    memcpy(D.rd_ipos[d], com, 24);
    D.rd_inertia[d][0] = I[0]; D.rd_inertia[d][1] = I[4]; D.rd_inertia[d][2] = I[8]; D.rd_inertia[d][3] = I[1]; D.rd_inertia[d][4] = I[2]; D.rd_inertia[d][5] = I[5];
  }
  memcpy(D.ref_point, D.rd_pos[0], 24);
  // ee_link
  if (ee_body <= 0 || ee_body >= nbody || body_tree[ee_body] != 0) return fail(UR5_ERR_MODEL, "cfg.ee_body is not a robot body");
  D.ee_cbody = cb_of_body[body_weld[ee_body]];
  for (int d = 0; d <= D.ee_cbody; d++) if (D.rd_parent[d] != d - 1) return fail(UR5_ERR_MODEL, "arm chain up to ee_link must be serial");
  {
    Xf x = rel_to(ee_body, body_weld[ee_body]);
    memcpy(D.ee_pos, x.p, 24);
    qmat(x.q, D.ee_mat);
  }
  // ---- objects: trees 1.. , one body, one geom at the origin, diagonal inertia
  std::vector<int> obj_of_body(nbody, -1);
  int nobj = 0;
  for (int b = 1; b < nbody; b++) {
    if (body_tree[b] <= 0) continue;
    if (body_parent[b] != 0 || nobj >= UR5_MAXOBJ) return fail(UR5_ERR_MODEL, "objects must be top-level bodies, at most " + std::to_string(UR5_MAXOBJ) + " per scene");
    int j = body_jntadr[b], k = nobj;
    if (body_jntnum[b] == 1 && jnt_type[j] == 0) D.obj_kind[k] = 1;
    else if (body_jntnum[b] == 4 && jnt_type[j] == 2 && jnt_type[j + 1] == 2 && jnt_type[j + 2] == 2 && jnt_type[j + 3] == 1) D.obj_kind[k] = 0;
    else return fail(UR5_ERR_MODEL, "object joints must be 'free' or 3 slides + ball");
    if (jnt_dofadr[j] != nrd + 6 * k || jnt_qposadr[j] != nrd + 7 * k) return fail(UR5_ERR_MODEL, "object dofs must follow the robot dofs contiguously");
    const double* bi = body_inertia + 6 * b;
    if (fabs(body_ipos[3 * b]) + fabs(body_ipos[3 * b + 1]) + fabs(body_ipos[3 * b + 2]) > 1e-12 || fabs(bi[3]) + fabs(bi[4]) + fabs(bi[5]) > 1e-12)
      return fail(UR5_ERR_MODEL, "object inertia must be diagonal about the body origin");
    if (D.obj_kind[k] == 0) {
      for (int a = 0; a < 3; a++) {
        const double* ax = jnt_axis + 3 * (j + a);
        if (fabs(ax[a] - 1) > 1e-12) return fail(UR5_ERR_MODEL, "object slides must be x, y, z");
        D.obj_limited[k][a] = jnt_limited[j + a]; D.obj_lo[k][a] = jnt_range[2 * (j + a)]; D.obj_hi[k][a] = jnt_range[2 * (j + a) + 1];
      }
      memcpy(D.obj_pos0[k], body_pos + 3 * b, 24);
    }
    int d0 = nrd + 6 * k;
    memcpy(D.obj_qpos0[k], qpos0 + nrd + 7 * k, 56);
    D.obj_mass[k] = body_mass[b];
    for (int a = 0; a < 3; a++) D.obj_inertia[k][a] = bi[a];
    D.obj_arm[k][0] = dof_arm[d0]; D.obj_arm[k][1] = dof_arm[d0 + 3];
    D.obj_damp[k][0] = dof_damp[d0]; D.obj_damp[k][1] = dof_damp[d0 + 3];
    D.obj_invweight[k][0] = dof_invw[d0]; D.obj_invweight[k][1] = dof_invw[d0 + 3];
    obj_of_body[b] = nobj++;
  }
  D.nobj = nobj;
  D.nv = nrd + 6 * nobj; D.nq = nrd + 7 * nobj;
  if (D.nv != nv || D.nq != nq) return fail(UR5_ERR_MODEL, "model has dofs outside the robot tree and the objects");
  // ---- geoms (collidable only)
  std::vector<int> dev_of_geom(ngeom, -1);
  int ng = 0, nhv = 0, ndg = 0;
  std::vector<int> mesh_dev_adr;
  dev2model_geom->clear();
  for (int g = 0; g < ngeom; g++) {
    if (!geom_collide[g]) continue;
    bool used = false;
    for (int p = 0; p < npair; p++) if (pair1[p] == g || pair2[p] == g) used = true;
    if (!used) continue;
    if (ng >= UR5_MAXG) return fail(UR5_ERR_MODEL, "too many collidable geoms");
    int b = geom_body[g], k = ng++;
    dev_of_geom[g] = k;
    dev2model_geom->push_back(g);
    D.g_type[k] = geom_type[g]; D.g_condim[k] = geom_condim[g];
    if (geom_condim[g] > (UR5_NB > 4 ? 6 : 4)) return fail(UR5_ERR_MODEL, "condim 6 (rolling friction) needs the many-object variant (NB = 6)");
    memcpy(D.g_size[k], geom_size + 3 * g, 24);
    D.g_rbound[k] = geom_rbound[g]; D.g_margin[k] = geom_margin[g];
    memcpy(D.g_friction[k], geom_friction + 3 * g, 24);
    memcpy(D.g_solref[k], geom_solref + 2 * g, 16);
    memcpy(D.g_solimp[k], geom_solimp + 5 * g, 40);
    D.g_invw[k][0] = body_invw[2 * b]; D.g_invw[k][1] = body_invw[2 * b + 1];
    Xf gl; memcpy(gl.p, geom_pos + 3 * g, 24); memcpy(gl.q, B.F("geom_quat") + 4 * g, 32);
    D.g_dg[k] = -1;
    if (body_weld[b] == 0) {
      D.g_kind[k] = UR5_KIND_STATIC; D.g_owner[k] = -1;
      Xf x = compose(rel_to(b, 0), gl);
      memcpy(D.g_pos[k], x.p, 24); qmat(x.q, D.g_mat[k]);
    } else if (body_tree[b] == 0) {
      D.g_kind[k] = UR5_KIND_ROBOT; D.g_owner[k] = cb_of_body[body_weld[b]];
      Xf x = compose(rel_to(b, body_weld[b]), gl);
      memcpy(D.g_pos[k], x.p, 24); qmat(x.q, D.g_mat[k]);
    } else {
      D.g_kind[k] = UR5_KIND_OBJECT; D.g_owner[k] = obj_of_body[b];
      if (fabs(gl.p[0]) + fabs(gl.p[1]) + fabs(gl.p[2]) > 1e-12 || fabs(gl.q[0] - 1) > 1e-12) return fail(UR5_ERR_MODEL, "object geoms must sit at the body origin");
      double id[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      memcpy(D.g_mat[k], id, 72);
    }
    if (D.g_kind[k] != UR5_KIND_STATIC) {
      if (ndg >= UR5_MAXDG) return fail(UR5_ERR_MODEL, "too many moving collidable geoms");
      D.g_dg[k] = ndg; D.dg_geom[ndg++] = k;
    }
    if (geom_type[g] == UR5_GEOM_MESH) {
      // hull vertices are stored once per MESH (the two fingers / knuckles share theirs): the reference's full hulls, 400 / 70 / 120 vertices
      // (UR5gripper_2_finger.xml:54-71,188-212), 590 per scene
      int mid = geom_mesh[g], n = mesh_num[mid];
      if ((int)mesh_dev_adr.size() <= mid) mesh_dev_adr.resize(mid + 1, -1);
      if (mesh_dev_adr[mid] < 0) {
        if (nhv + n > UR5_MAXHV) return fail(UR5_ERR_MODEL, "collidable hulls exceed the vertex budget");
        mesh_dev_adr[mid] = nhv;
        for (int i = 0; i < n; i++) memcpy(D.hullvert[nhv + i], mesh_vert + 3 * (mesh_adr[mid] + i), 24);
        nhv += n;
      }
      D.g_vadr[k] = mesh_dev_adr[mid]; D.g_vnum[k] = n;
      double c[3] = {0, 0, 0}, lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
      for (int i = 0; i < n; i++) {
        const double* v = mesh_vert + 3 * (mesh_adr[mid] + i);
        for (int a = 0; a < 3; a++) { c[a] += v[a]; lo[a] = v[a] < lo[a] ? v[a] : lo[a]; hi[a] = v[a] > hi[a] ? v[a] : hi[a]; }
      }
      // g_center: an interior point (MPR's v0). g_boxc / g_size: the hull's tight bounding box in the geom frame (broad phase only; the blob's
      // geom_size of a mesh is the looser origin-centred max |v|)
      for (int a = 0; a < 3; a++) { D.g_center[k][a] = c[a] / (n > 0 ? n : 1); D.g_boxc[k][a] = 0.5 * (hi[a] + lo[a]); D.g_size[k][a] = 0.5 * (hi[a] - lo[a]); }
    }
  }
  D.ngeom = ng; D.ndg = ndg;
  D.nrg = 0;
  for (int d = 0; d < nrd; d++) D.rd_gslot[d] = -1;
  for (int k = 0; k < ng; k++) if (D.g_kind[k] == UR5_KIND_ROBOT && D.rd_gslot[D.g_owner[k]] < 0) {
    if (D.nrg >= UR5_MAXRG) return fail(UR5_ERR_MODEL, "too many robot weld groups carry collision geoms for this engine variant (4 in the wavefront-per-scene engine: gripper only; 8 in the many-object engine)");
    D.rd_gslot[D.g_owner[k]] = D.nrg; D.rg_body[D.nrg++] = D.g_owner[k];
  }
  int np = 0;
  for (int p = 0; p < npair; p++) {
    int a = dev_of_geom[pair1[p]], b = dev_of_geom[pair2[p]];
    if (a < 0 || b < 0) continue;
    if (np >= UR5_MAXPAIR) return fail(UR5_ERR_MODEL, "too many collision pairs");
    if (D.g_type[a] > D.g_type[b]) { int t = a; a = b; b = t; }
    D.pair_g1[np] = (ur5_pair_t)a; D.pair_g2[np] = (ur5_pair_t)b; np++;
  }
  D.npair = np;
  // ---- render model: every geom (collidable or not) with its pose relative to the engine body that carries it
  if (RM) {
    Ur5RenderModel& R = *RM;
    memset(&R, 0, sizeof R);
    int npl_tot = 0, ncam = 0;
    const int* vadr = B.I("vis_planeadr");
    const int* vnum = B.I("vis_planenum");
    const double* vpl = B.F("vis_plane", &npl_tot);
    const double* rgba = B.F("geom_rgba");
    const double *cpos = B.F("cam_pos"), *cmat = B.F("cam_mat"), *cfov = B.F("cam_fovy", &ncam);
    if (!vadr || !vnum || !rgba || !cfov) return fail(UR5_ERR_MODEL, "model blob lacks the render sections (re-run tools/compile_models.py)");
    if (ngeom > UR5_R_MAXG || npl_tot / 4 > UR5_R_MAXPL || ncam > UR5_R_MAXCAM) return fail(UR5_ERR_MODEL, "render model exceeds its budgets");
    R.ngeom = ngeom; R.nplane = npl_tot / 4; R.ncam = ncam;
    for (int i = 0; i < npl_tot; i++) R.plane[i / 4][i % 4] = (float)vpl[i];
    const double* gq = B.F("geom_quat");
    for (int g = 0; g < ngeom; g++) {
      int b = geom_body[g];
      Xf gl; memcpy(gl.p, geom_pos + 3 * g, 24); memcpy(gl.q, gq + 4 * g, 32);
      Xf x;
      if (body_weld[b] == 0) { R.g_kind[g] = UR5_KIND_STATIC; R.g_owner[g] = -1; x = compose(rel_to(b, 0), gl); }
      else if (body_tree[b] == 0) { R.g_kind[g] = UR5_KIND_ROBOT; R.g_owner[g] = cb_of_body[body_weld[b]]; x = compose(rel_to(b, body_weld[b]), gl); }
      else { R.g_kind[g] = UR5_KIND_OBJECT; R.g_owner[g] = obj_of_body[b]; x = gl; }
      double m9[9];
      qmat(x.q, m9);
      for (int k = 0; k < 3; k++) { R.g_pos[g][k] = (float)x.p[k]; R.g_size[g][k] = (float)geom_size[3 * g + k]; }
      for (int k = 0; k < 9; k++) R.g_mat[g][k] = (float)m9[k];
      for (int k = 0; k < 4; k++) R.g_rgba[g][k] = (float)rgba[4 * g + k];
      R.g_type[g] = geom_type[g]; R.g_rbound[g] = (float)geom_rbound[g];
      if (geom_type[g] == UR5_GEOM_MESH) { R.g_padr[g] = vadr[geom_mesh[g]]; R.g_pnum[g] = vnum[geom_mesh[g]]; }
    }
    for (int c = 0; c < ncam; c++) {
      for (int k = 0; k < 3; k++) R.cam_pos[c][k] = (float)cpos[3 * c + k];
      for (int k = 0; k < 9; k++) R.cam_mat[c][k] = (float)cmat[9 * c + k];
      R.cam_fovy[c] = (float)cfov[c];
    }
    R.znear = (float)(optf[15] * optf[14]); R.zfar = (float)(optf[16] * optf[14]);
    const double l[3] = {1.0, -1.0, 3.0 - 0.435};   // light3: directional, pos (1,-1,3) aimed at box_link (UR5gripper_2_finger.xml:108)
    double ln = sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
    for (int k = 0; k < 3; k++) R.light[k] = (float)(l[k] / ln);
    R.sky[0] = 0.65f; R.sky[1] = 0.65f; R.sky[2] = 0.9f;
  }
  // ---- equality, actuators, options
  if (neq > 2) return fail(UR5_ERR_MODEL, "at most two joint equalities");
  D.neq = neq;
  for (int e = 0; e < neq; e++) {
    D.eq_d1[e] = jnt_dofadr[eq1[e]]; D.eq_d2[e] = jnt_dofadr[eq2[e]];
    if (D.eq_d1[e] >= nrd || D.eq_d2[e] >= nrd) return fail(UR5_ERR_MODEL, "joint equality must couple robot joints");
    memcpy(D.eq_poly[e], eq_poly + 5 * e, 40); memcpy(D.eq_solref[e], eq_solref + 2 * e, 16); memcpy(D.eq_solimp[e], eq_solimp + 5 * e, 40);
  }
  if (nu > UR5_MAXNU || nu != 7) return fail(UR5_ERR_MODEL, "expected the 7 motors of UR5gripper_2_finger*.xml:347-357");
  D.nu = nu;
  // MujocoController.py:157-235: Kp = {7,10,5,7,5,5,2.5} x 3, Kd = {1.1,1.0,0.5,0.1,0.1,0.1,0} x 0.1, output limits = ctrlrange
  static const double kd[7] = {1.1 * 0.1, 1.0 * 0.1, 0.5 * 0.1, 0.1 * 0.1, 0.1 * 0.1, 0.1 * 0.1, 0.0};
  static const double lim[7] = {2, 2, 2, 1, 1, 1, 1};
  for (int a = 0; a < nu; a++) {
    D.act_dof[a] = jnt_dofadr[act_jnt[a]];
    if (D.act_dof[a] >= nrd) return fail(UR5_ERR_MODEL, "actuators must drive robot joints");
    D.act_gear[a] = act_gear[a];
    D.act_lo[a] = act_lim[a] ? act_range[2 * a] : -1e300; D.act_hi[a] = act_lim[a] ? act_range[2 * a + 1] : 1e300;
    D.pid_kd[a] = kd[a]; D.pid_lo[a] = -lim[a]; D.pid_hi[a] = lim[a];
  }
  D.timestep = optf[0]; D.tolerance = optf[1]; D.impratio = optf[2];
  for (int i = 0; i < 3; i++) D.gravity[i] = optf[3 + i];
  for (int i = 0; i < 2; i++) D.jnt_solref[i] = optf[6 + i];
  for (int i = 0; i < 5; i++) D.jnt_solimp[i] = optf[8 + i];
  D.meaninertia = optf[13];
  D.iterations = opti[0];
  return 0;
}

// MujocoController.py:157-247: controller construction state (each PID called once with input 0)
static const double PID_KP[7] = {7 * 3.0, 10 * 3.0, 5 * 3.0, 7 * 3.0, 5 * 3.0, 5 * 3.0, 2.5 * 3.0};
static const double PID_SP[7] = {0, -1.57, 1.57, -1.57, -1.57, 0, 0};

using ::Ur5SplitMix;
UR5_HD inline void reset_record(const Ur5DevModel& M, const double* qpos0, double* r, uint64_t seed) { ur5_reset_record(M, qpos0, r, seed); }

}  // namespace ur5host

// ======================================================================================================= handle + C ABI
struct ur5_sim {
  int variant = 0;   // 0: wavefront-per-scene engine (<= 6 objects), 1: many-object engine; first member in both translation units
  Ur5DevModel hm;
  Ur5DevModel* dm = nullptr;
  int n = 0, nvt = 0, device = 0, contacts_enabled = 1;
  double pid_dt = 0;
  std::vector<double> qpos0;
  std::vector<int> dev2model_geom;
  double* d_rec = nullptr;
  Ur5RenderModel hrm;
  Ur5RenderModel* d_rm = nullptr;
  uint8_t* d_rgb = nullptr;
  float* d_depth = nullptr;
  size_t img_cap = 0;
  unsigned* d_mask = nullptr;
  double *d_target = nullptr, *d_tol = nullptr, *d_debug = nullptr, *d_hess = nullptr, *d_qpos0 = nullptr;
  void* d_gpose = nullptr;        // render: per-scene geom poses + screen boxes (HIP backend)
  const int* d_step_cap = nullptr;   // test hook (ur5_set_step_cap_dev): [n] physics-step caps of the scripted launches (the handle's copy d_step_cap_buf), NULL = none
  int* d_step_cap_buf = nullptr;
  // observation inside ruled launches (ur5_set_observation_dev): caller-owned frame buffers
  uint8_t* obs_rgb = nullptr; float* obs_depth = nullptr; int obs_cam = 0, obs_w = 0, obs_h = 0, obs_mode = 0, obs_frames = 0;
  const int* d_order = nullptr;   // dispatch order of the scripted launches (ur5_set_order_dev), NULL = scene order; points at d_order_buf
  int* d_order_buf = nullptr;     // handle-owned copy, made by an ASYNCHRONOUS device-to-device copy on the handle's stream: the caller's buffer must stay valid (and unmodified) until the work
                                  // queued on that stream so far has run (include/ur5sim.h); same-stream callers (ur5_set_stream) have nothing to do
  double kernel_ms_total = 0;   // engine-kernel time of every launch since ur5_create (HIP events on the handle's stream)
  int *d_max = nullptr, *d_result = nullptr, *d_steps = nullptr, *d_ps = nullptr, *d_pr = nullptr;
  std::vector<double> h_rec;
  double last_ms = 0;
  void* be = nullptr;  // backend private
};

// backend hooks, defined by the including translation unit
static int be_open(ur5_sim* h, int device_id);
static void be_close(ur5_sim* h);
static void* be_alloc(ur5_sim* h, size_t bytes);
static void be_free(ur5_sim* h, void* p);
static int be_h2d(ur5_sim* h, void* dst, const void* src, size_t bytes);
static int be_d2h(ur5_sim* h, void* dst, const void* src, size_t bytes);
static int be_d2d_async(ur5_sim* h, void* dst, const void* src, size_t bytes);   // device-to-device on the handle's stream, stream-ordered, no host wait
static int be_launch(ur5_sim* h, const Ur5Launch& P);
static int be_sync(ur5_sim* h);
static int be_set_stream(ur5_sim* h, void* stream, int external);
static int be_render(ur5_sim* h, int cam, int W, int Hh, int mode, uint8_t* rgb_dev, float* depth_dev);
static bool be_can_observe(ur5_sim* h);   // the engine instantiation that runs this handle's scenes has room for Engine::observe's working set
static int be_reset_dev(ur5_sim* h, const uint64_t* seeds_dev, const uint8_t* mask_dev, int chunks, int* max_steps_dev);
static long g_model_uploads = 0;   // test hook (ur5_model_uploads): model copies this unit has sent to a device -- one per handle, by ur5_create, never at a launch

namespace ur5host {
static int pull(ur5_sim* h) { h->h_rec.resize((size_t)h->n * UR5_REC_STRIDE); return be_d2h(h, h->h_rec.data(), h->d_rec, h->h_rec.size() * 8); }
static int push(ur5_sim* h) { return be_h2d(h, h->d_rec, h->h_rec.data(), h->h_rec.size() * 8); }
static Ur5Launch base_launch(ur5_sim* h, int op) {
  Ur5Launch P;
  memset(&P, 0, sizeof P);
  P.op = op; P.n_env = h->n; P.contacts_enabled = h->contacts_enabled; P.pid_dt = h->pid_dt; P.table_height = 0.91;
  P.hess = h->d_hess;
  P.order = (op == UR5_OP_GRASP || op == UR5_OP_STAY) ? h->d_order : nullptr;
  P.step_cap = op == UR5_OP_GRASP ? h->d_step_cap : nullptr;
#ifdef UR5_PROFILE
  if (!h->d_debug) h->d_debug = (double*)be_alloc(h, (size_t)h->n * UR5_DEBUG_STRIDE * 8);
  P.debug = h->d_debug;
#endif
  return P;
}
template <class T> static int upload(ur5_sim* h, T* dst, const T* src, size_t count) { return be_h2d(h, dst, src, count * sizeof(T)); }
static int ensure_qpos0(ur5_sim* h) {
  if (h->d_qpos0) return 0;
  h->d_qpos0 = (double*)be_alloc(h, h->qpos0.size() * 8);
  if (!h->d_qpos0) return fail(UR5_ERR_DEVICE, "device allocation failed (qpos0)");
  return be_h2d(h, h->d_qpos0, h->qpos0.data(), h->qpos0.size() * 8);
}
}  // namespace ur5host

// One library, two engines: this header is compiled twice. The many-object translation unit (-DUR5_MANY, ur5sim_many.hip)
// renames its entry points ur5_* -> ur5m_* (ur5_many_names.h); the small-scene unit owns the public names and forwards any
// handle whose variant tag is 1 -- and ur5_create for models with more than 6 objects -- to them.
#ifdef UR5_MANY
#define UR5_FWD(name, args)
#define UR5_FWD_VOID(name, args)
#else
#define UR5_FWD(name, args) if (h && h->variant == 1) { ur5host::g_err_many = true; return ur5m_##name args; }
#define UR5_FWD_VOID(name, args) if (h && h->variant == 1) { ur5m_##name args; return; }
extern "C" {
int ur5m_create(const void* blob, size_t nbytes, int n_env, int device_id, const ur5_config* cfg, ur5_sim** out);
const char* ur5m_last_error(void);
void ur5m_destroy(ur5_sim* h);
int ur5m_num_envs(const ur5_sim* h); int ur5m_nq(const ur5_sim* h); int ur5m_nv(const ur5_sim* h); int ur5m_nu(const ur5_sim* h);
int ur5m_set_state(ur5_sim* h, const double* qpos, const double* qvel, const double* warm, const double* pid);
int ur5m_get_state(ur5_sim* h, double* qpos, double* qvel, double* warm, double* pid);
int ur5m_set_ctrl(ur5_sim* h, const double* ctrl); int ur5m_get_ctrl(ur5_sim* h, double* ctrl);
int ur5m_get_counters(ur5_sim* h, int64_t* c);
int ur5m_stay(ur5_sim* h, double ms); int ur5m_reset(ur5_sim* h, const uint64_t* seeds, int mode, double settle_ms);
int ur5m_reset_dev(ur5_sim* h, const uint64_t* seeds_dev, const uint8_t* mask_dev, double settle_ms); double ur5m_kernel_ms_total(ur5_sim* h); int ur5m_step(ur5_sim* h, int nsteps);
int ur5m_move_group(ur5_sim* h, const uint32_t* mask, const double* target, const double* tol, const int* max_steps, int* result, int* steps);
int ur5m_move_ee(ur5_sim* h, const double* xyz, const double* tol, const int* max_steps, int* result, int* steps);
int ur5m_grasp_attempt_dev(ur5_sim* h, const double* action_dev, int check_mode, double table_height, int* reward_dev);
int ur5m_grasp_attempt_reset_dev(ur5_sim* h, const double* action_dev, int check_mode, double table_height, int* reward_dev, const uint64_t* reset_seeds_dev, double settle_ms);
int ur5m_grasp_attempt(ur5_sim* h, const double* action, const uint8_t* skip, int check_mode, double table_height, int* reward, int* phase_steps, int* phase_result);
int ur5m_ik(ur5_sim* h, const double* xyz, double* q5, int* result);
int ur5m_render_dev(ur5_sim* h, int camera_id, int width, int height, int depth_mode, uint8_t* rgb_dev, float* depth_dev);
int ur5m_render(ur5_sim* h, int camera_id, int width, int height, int depth_mode, uint8_t* rgb, float* depth);
int ur5m_sync(ur5_sim* h); int ur5m_set_order_dev(ur5_sim* h, const int* order_dev); int ur5m_set_order_view_dev(ur5_sim* h, const int* order_dev); int ur5m_set_observation_dev(ur5_sim* h, int camera_id, int width, int height, int depth_mode, uint8_t* rgb_dev, float* depth_dev, int frames);
int ur5m_grasp_rounds_dev(ur5_sim* h, const ur5_aim_rule* rule, int round0, int rounds, int check_mode, double table_height, int* reward_dev, double* action_out_dev, double settle_ms); int ur5m_set_stream(ur5_sim* h, void* s, int external); double ur5m_last_launch_ms(ur5_sim* h); void* ur5m_state_device_ptr(ur5_sim* h);
int ur5m_forward_debug(ur5_sim* h, double* out); int ur5m_set_step_cap_dev(ur5_sim* h, const int* cap_dev); long ur5m_model_uploads(ur5_sim* h); int ur5m_body_xpos(ur5_sim* h, double* out); int ur5m_profile_read(ur5_sim* h, double* out);
}
#endif

extern "C" {
#ifdef UR5_MANY
const char* ur5_last_error(void) { return ur5host::g_err.c_str(); }
#else
// the two translation units keep separate thread-local messages: report whichever was set last
const char* ur5_last_error(void) { return ur5host::g_err_many ? ur5m_last_error() : ur5host::g_err.c_str(); }
#endif

int ur5_create(const void* blob, size_t nbytes, int n_env, int device_id, const ur5_config* cfg, ur5_sim** out) {
  using namespace ur5host;
  if (!blob || !out || !cfg || n_env <= 0) return fail(UR5_ERR_ARG, "ur5_create: bad arguments");
#ifndef UR5_MANY
  if (ur5host::count_objects(blob, nbytes) > UR5_MAXOBJ) { ur5host::g_err_many = true; return ur5m_create(blob, nbytes, n_env, device_id, cfg, out); }
#endif
  ur5_sim* h = new ur5_sim();
  int rc = build_model(blob, nbytes, cfg->ee_body, &h->hm, &h->dev2model_geom, &h->hrm);
  if (rc) { delete h; return rc; }
  h->n = n_env; h->device = device_id; h->contacts_enabled = cfg->contacts_enabled;
  h->pid_dt = cfg->pid_dt > 0 ? cfg->pid_dt : h->hm.timestep;
  h->nvt = h->hm.nv <= 32 ? 32 : UR5_MAXNV;
#ifdef UR5_MANY
  h->variant = 1; h->nvt = UR5_MAXNV;
#endif
  Blob B{(const char*)blob, nbytes};
  int nq;
  const double* q0 = B.F("qpos0", &nq);
  h->qpos0.assign(q0, q0 + nq);
  rc = be_open(h, device_id);
  if (rc) { delete h; return rc; }
  size_t n = (size_t)n_env;
  h->dm = (Ur5DevModel*)be_alloc(h, sizeof(Ur5DevModel));
  h->d_rm = (Ur5RenderModel*)be_alloc(h, sizeof(Ur5RenderModel));
  h->d_rec = (double*)be_alloc(h, n * UR5_REC_STRIDE * 8);
  h->d_mask = (unsigned*)be_alloc(h, n * 4); h->d_target = (double*)be_alloc(h, n * 8 * 8); h->d_tol = (double*)be_alloc(h, n * 8);
  h->d_max = (int*)be_alloc(h, n * 4); h->d_result = (int*)be_alloc(h, n * 4); h->d_steps = (int*)be_alloc(h, n * 4);
  h->d_ps = (int*)be_alloc(h, n * 12 * 4); h->d_pr = (int*)be_alloc(h, n * 12 * 4);
#ifdef UR5_MANY
  h->d_hess = (double*)be_alloc(h, n * (size_t)UR5_HESS_STRIDE * 8);
  if (!h->d_hess) { ur5_destroy(h); return fail(UR5_ERR_DEVICE, "device allocation failed (Hessian scratch)"); }
#endif
  if (!h->dm || !h->d_rec || !h->d_mask || !h->d_target || !h->d_tol || !h->d_max || !h->d_result || !h->d_steps || !h->d_ps || !h->d_pr) {
    ur5_destroy(h);
    return fail(UR5_ERR_DEVICE, "device allocation failed");
  }
  be_h2d(h, h->dm, &h->hm, sizeof(Ur5DevModel));   // the kernels read the model THROUGH this pointer (a kernel argument, struct Engine's only member): handles never share or evict a model
  g_model_uploads++;
  if (h->d_rm) be_h2d(h, h->d_rm, &h->hrm, sizeof(Ur5RenderModel));
  // initial records: qpos0, controller construction state
  h->h_rec.assign(n * UR5_REC_STRIDE, 0.0);
  for (size_t e = 0; e < n; e++) {
    double* r = h->h_rec.data() + e * UR5_REC_STRIDE;
    for (int i = 0; i < nq; i++) r[UR5_REC_QPOS + i] = q0[i];
    for (int a = 0; a < h->hm.nu; a++) {
      r[UR5_REC_TARGET + a] = PID_SP[a]; r[UR5_REC_KP + a] = PID_KP[a]; r[UR5_REC_PIDIN + a] = 0.0;
      double o = PID_KP[a] * PID_SP[a];
      r[UR5_REC_PIDOUT + a] = o < h->hm.pid_lo[a] ? h->hm.pid_lo[a] : (o > h->hm.pid_hi[a] ? h->hm.pid_hi[a] : o);
    }
  }
  rc = push(h);
  if (rc) { ur5_destroy(h); return rc; }
  *out = h;
  return 0;
}

void ur5_destroy(ur5_sim* h) {
  UR5_FWD_VOID(destroy, (h));
  if (!h) return;
  void* ptrs[] = {h->dm, h->d_rec, h->d_mask, h->d_target, h->d_tol, h->d_max, h->d_result, h->d_steps, h->d_ps, h->d_pr, h->d_debug, h->d_rm, h->d_rgb, h->d_depth, h->d_hess, h->d_qpos0, h->d_gpose, h->d_order_buf, h->d_step_cap_buf};
  for (void* p : ptrs) if (p) be_free(h, p);
  be_close(h);
  delete h;
}
int ur5_num_envs(const ur5_sim* h) {
  UR5_FWD(num_envs, (h)); return h->n; }
int ur5_nq(const ur5_sim* h) {
  UR5_FWD(nq, (h)); return h->hm.nq; }
int ur5_nv(const ur5_sim* h) {
  UR5_FWD(nv, (h)); return h->hm.nv; }
int ur5_nu(const ur5_sim* h) {
  UR5_FWD(nu, (h)); return h->hm.nu; }

int ur5_set_state(ur5_sim* h, const double* qpos, const double* qvel, const double* warm, const double* pid) {
  UR5_FWD(set_state, (h, qpos, qvel, warm, pid));
  using namespace ur5host;
  int rc = pull(h);
  if (rc) return rc;
  const Ur5DevModel& M = h->hm;
  for (int e = 0; e < h->n; e++) {
    double* r = h->h_rec.data() + (size_t)e * UR5_REC_STRIDE;
    if (qpos) for (int i = 0; i < M.nq; i++) r[UR5_REC_QPOS + i] = qpos[(size_t)e * M.nq + i];
    if (qvel) for (int i = 0; i < M.nv; i++) r[UR5_REC_QVEL + i] = qvel[(size_t)e * M.nv + i];
    if (warm) for (int i = 0; i < M.nv; i++) r[UR5_REC_WARM + i] = warm[(size_t)e * M.nv + i];
    if (pid) for (int a = 0; a < M.nu; a++) {
      const double* p = pid + ((size_t)e * M.nu + a) * 4;
      r[UR5_REC_TARGET + a] = p[0]; r[UR5_REC_PIDIN + a] = p[1]; r[UR5_REC_PIDOUT + a] = p[2]; r[UR5_REC_KP + a] = p[3];
    }
  }
  return push(h);
}
int ur5_get_state(ur5_sim* h, double* qpos, double* qvel, double* warm, double* pid) {
  UR5_FWD(get_state, (h, qpos, qvel, warm, pid));
  using namespace ur5host;
  int rc = pull(h);
  if (rc) return rc;
  const Ur5DevModel& M = h->hm;
  for (int e = 0; e < h->n; e++) {
    const double* r = h->h_rec.data() + (size_t)e * UR5_REC_STRIDE;
    if (qpos) for (int i = 0; i < M.nq; i++) qpos[(size_t)e * M.nq + i] = r[UR5_REC_QPOS + i];
    if (qvel) for (int i = 0; i < M.nv; i++) qvel[(size_t)e * M.nv + i] = r[UR5_REC_QVEL + i];
    if (warm) for (int i = 0; i < M.nv; i++) warm[(size_t)e * M.nv + i] = r[UR5_REC_WARM + i];
    if (pid) for (int a = 0; a < M.nu; a++) {
      double* p = pid + ((size_t)e * M.nu + a) * 4;
      p[0] = r[UR5_REC_TARGET + a]; p[1] = r[UR5_REC_PIDIN + a]; p[2] = r[UR5_REC_PIDOUT + a]; p[3] = r[UR5_REC_KP + a];
    }
  }
  return 0;
}
int ur5_set_ctrl(ur5_sim* h, const double* ctrl) {
  UR5_FWD(set_ctrl, (h, ctrl));
  using namespace ur5host;
  int rc = pull(h);
  if (rc) return rc;
  for (int e = 0; e < h->n; e++) for (int a = 0; a < h->hm.nu; a++) h->h_rec[(size_t)e * UR5_REC_STRIDE + UR5_REC_CTRL + a] = ctrl[(size_t)e * h->hm.nu + a];
  return push(h);
}
int ur5_get_ctrl(ur5_sim* h, double* ctrl) {
  UR5_FWD(get_ctrl, (h, ctrl));
  using namespace ur5host;
  int rc = pull(h);
  if (rc) return rc;
  for (int e = 0; e < h->n; e++) for (int a = 0; a < h->hm.nu; a++) ctrl[(size_t)e * h->hm.nu + a] = h->h_rec[(size_t)e * UR5_REC_STRIDE + UR5_REC_CTRL + a];
  return 0;
}
int ur5_get_counters(ur5_sim* h, int64_t* c) {
  UR5_FWD(get_counters, (h, c));
  using namespace ur5host;
  int rc = pull(h);
  if (rc) return rc;
  for (int e = 0; e < h->n; e++) {
    const double* r = h->h_rec.data() + (size_t)e * UR5_REC_STRIDE + UR5_REC_MISC;
    c[6 * e] = (int64_t)r[0]; c[6 * e + 1] = (int64_t)r[1]; c[6 * e + 2] = (int64_t)r[3] | ((int64_t)r[7] << 8); c[6 * e + 3] = (int64_t)r[4]; c[6 * e + 4] = (int64_t)r[5]; c[6 * e + 5] = (int64_t)r[6];
  }
  return 0;
}

int ur5_stay(ur5_sim* h, double ms) {
  UR5_FWD(stay, (h, ms));
  using namespace ur5host;
  int chunks = (int)std::ceil(ms / 1000.0 / h->hm.timestep / 10.0 - 1e-9);
  std::vector<int> mx(h->n, chunks);
  int rc = upload(h, h->d_max, mx.data(), mx.size());
  if (rc) return rc;
  Ur5Launch P = base_launch(h, UR5_OP_STAY);
  P.max_steps = h->d_max;
  rc = be_launch(h, P);
  return rc ? rc : be_sync(h);
}

int ur5_reset(ur5_sim* h, const uint64_t* seeds, int mode, double settle_ms) {
  UR5_FWD(reset, (h, seeds, mode, settle_ms));
  using namespace ur5host;
  (void)mode;
  if (!seeds) return fail(UR5_ERR_ARG, "ur5_reset: seeds is NULL");
  int rc = pull(h);
  if (rc) return rc;
  for (int e = 0; e < h->n; e++) reset_record(h->hm, h->qpos0.data(), h->h_rec.data() + (size_t)e * UR5_REC_STRIDE, seeds[e]);
  rc = push(h);
  if (rc) return rc;
  return settle_ms > 0 ? ur5_stay(h, settle_ms) : 0;  // GraspingEnv.py:473
}

int ur5_reset_dev(ur5_sim* h, const uint64_t* seeds_dev, const uint8_t* mask_dev, double settle_ms) {
  UR5_FWD(reset_dev, (h, seeds_dev, mask_dev, settle_ms));
  using namespace ur5host;
  if (!seeds_dev) return fail(UR5_ERR_ARG, "ur5_reset_dev: seeds_dev is NULL");
  { int rc0 = ensure_qpos0(h); if (rc0) return rc0; }
  const int chunks = settle_ms > 0 ? (int)std::ceil(settle_ms / 1000.0 / h->hm.timestep / 10.0 - 1e-9) : 0;
  int rc = be_reset_dev(h, seeds_dev, mask_dev, chunks, h->d_max);   // samples the flagged records, d_max[e] = flagged ? chunks : 0
  if (rc || chunks == 0) return rc;
  Ur5Launch P = base_launch(h, UR5_OP_STAY);
  P.max_steps = h->d_max;
  return be_launch(h, P);
}

int ur5_step(ur5_sim* h, int nsteps) {
  UR5_FWD(step, (h, nsteps));
  using namespace ur5host;
  std::vector<int> mx(h->n, nsteps);
  int rc = upload(h, h->d_max, mx.data(), mx.size());
  if (rc) return rc;
  Ur5Launch P = base_launch(h, UR5_OP_STEP);
  P.max_steps = h->d_max;
  rc = be_launch(h, P);
  return rc ? rc : be_sync(h);
}

int ur5_move_group(ur5_sim* h, const uint32_t* mask, const double* target, const double* tol, const int* max_steps, int* result, int* steps) {
  UR5_FWD(move_group, (h, mask, target, tol, max_steps, result, steps));
  using namespace ur5host;
  if (!mask || !tol || !max_steps) return fail(UR5_ERR_ARG, "ur5_move_group: mask/tol/max_steps are required");
  int rc = upload(h, h->d_mask, (const unsigned*)mask, (size_t)h->n);
  if (!rc && target) rc = upload(h, h->d_target, target, (size_t)h->n * 8);
  if (!rc) rc = upload(h, h->d_tol, tol, (size_t)h->n);
  if (!rc) rc = upload(h, h->d_max, max_steps, (size_t)h->n);
  if (rc) return rc;
  Ur5Launch P = base_launch(h, UR5_OP_MOVE);
  P.group_mask = h->d_mask; P.target = target ? h->d_target : nullptr; P.tol = h->d_tol; P.max_steps = h->d_max;
  P.result = h->d_result; P.steps = h->d_steps;
  rc = be_launch(h, P);
  if (!rc) rc = be_sync(h);
  if (!rc && result) rc = be_d2h(h, result, h->d_result, (size_t)h->n * 4);
  if (!rc && steps) rc = be_d2h(h, steps, h->d_steps, (size_t)h->n * 4);
  return rc;
}

int ur5_move_ee(ur5_sim* h, const double* xyz, const double* tol, const int* max_steps, int* result, int* steps) {
  UR5_FWD(move_ee, (h, xyz, tol, max_steps, result, steps));
  using namespace ur5host;
  if (!xyz || !tol || !max_steps) return fail(UR5_ERR_ARG, "ur5_move_ee: xyz/tol/max_steps are required");
  std::vector<double> t((size_t)h->n * 8, 0.0);
  for (int e = 0; e < h->n; e++) for (int k = 0; k < 3; k++) t[8 * e + k] = xyz[3 * e + k];
  int rc = upload(h, h->d_target, t.data(), t.size());
  if (!rc) rc = upload(h, h->d_tol, tol, (size_t)h->n);
  if (!rc) rc = upload(h, h->d_max, max_steps, (size_t)h->n);
  if (rc) return rc;
  Ur5Launch P = base_launch(h, UR5_OP_MOVE_EE);
  P.target = h->d_target; P.tol = h->d_tol; P.max_steps = h->d_max; P.result = h->d_result; P.steps = h->d_steps;
  rc = be_launch(h, P);
  if (!rc) rc = be_sync(h);
  if (!rc && result) rc = be_d2h(h, result, h->d_result, (size_t)h->n * 4);
  if (!rc && steps) rc = be_d2h(h, steps, h->d_steps, (size_t)h->n * 4);
  return rc;
}

int ur5_grasp_attempt_reset_dev(ur5_sim* h, const double* action_dev, int check_mode, double table_height, int* reward_dev,
                                const uint64_t* reset_seeds_dev, double settle_ms) {
  UR5_FWD(grasp_attempt_reset_dev, (h, action_dev, check_mode, table_height, reward_dev, reset_seeds_dev, settle_ms));
  using namespace ur5host;
  if (!action_dev || !reward_dev) return fail(UR5_ERR_ARG, "ur5_grasp_attempt_dev: NULL pointer");
  Ur5Launch P = base_launch(h, UR5_OP_GRASP);
  P.target = action_dev; P.check_mode = check_mode; P.table_height = table_height;
  P.result = reward_dev; P.steps = h->d_steps; P.phase_steps = h->d_ps; P.phase_result = h->d_pr;
  if (reset_seeds_dev) {
    int rc = ensure_qpos0(h);
    if (rc) return rc;
    P.reset_seeds = reset_seeds_dev; P.qpos0 = h->d_qpos0;
    P.reset_chunks = settle_ms > 0 ? (int)std::ceil(settle_ms / 1000.0 / h->hm.timestep / 10.0 - 1e-9) : 0;
  }
  return be_launch(h, P);
}
int ur5_grasp_rounds_dev(ur5_sim* h, const ur5_aim_rule* rule, int round0, int rounds, int check_mode, double table_height, int* reward_dev,
                         double* action_out_dev, double settle_ms) {
  UR5_FWD(grasp_rounds_dev, (h, rule, round0, rounds, check_mode, table_height, reward_dev, action_out_dev, settle_ms));
  using namespace ur5host;
  if (!h) return fail(UR5_ERR_ARG, "ur5_grasp_rounds_dev: NULL handle");
  if (!rule || !reward_dev || rounds < 1 || round0 < 0 || (rule->kind != 1 && rule->kind != 2) || rule->episode_rounds < 1 || rule->n_total < 1 || rule->first_scene_id < 0)   // (a negative round would index the rule's modular arithmetic out of range)
    return fail(UR5_ERR_ARG, "ur5_grasp_rounds_dev: rule (kind 1 or 2), reward_dev, round0 >= 0 and rounds >= 1 are required");
#ifdef UR5_MANY
  if (rule->kind != 2 || h->hm.nobj == 0 || h->hm.obj_kind[0] == 0) return fail(UR5_ERR_MODEL, "ur5_grasp_rounds_dev: 40-object piles are aimed by rule kind 2 (the box rule)");
#else
  if (rule->kind != 1) return fail(UR5_ERR_MODEL, "ur5_grasp_rounds_dev: rule kind 2 (the pile box rule) needs the 40-object scene");
#endif
  if (rule->z_from_depth && !h->obs_depth) return fail(UR5_ERR_ARG, "ur5_grasp_rounds_dev: z_from_depth needs the observation inside the launch (ur5_set_observation_dev)");
  Ur5Launch P = base_launch(h, UR5_OP_GRASP);
  P.check_mode = check_mode; P.table_height = table_height;
  P.result = reward_dev; P.steps = h->d_steps; P.phase_steps = h->d_ps; P.phase_result = h->d_pr;
  int rc = ensure_qpos0(h);
  if (rc) return rc;
  P.qpos0 = h->d_qpos0;
  P.reset_chunks = settle_ms > 0 ? (int)std::ceil(settle_ms / 1000.0 / h->hm.timestep / 10.0 - 1e-9) : 0;
  P.rounds = rounds; P.rule_kind = rule->kind; P.rule_r0 = round0; P.rule_ep = rule->episode_rounds;
  P.rule_gid0 = rule->first_scene_id; P.rule_ntotal = rule->n_total; P.rule_base_seed = rule->base_seed;
  const double plate[8] = {rule->plate_half_x, rule->plate_centre_y, rule->plate_half_y, rule->z_min, rule->z_max, rule->grasp_z, rule->fallback_x, rule->fallback_y};
  for (int k = 0; k < 8; k++) P.rule_plate[k] = plate[k];
  P.action_out = action_out_dev;
  if (h->obs_depth) {
    P.obs_rm = h->d_rm; P.obs_rgb = h->obs_rgb; P.obs_depth = h->obs_depth; P.obs_cam = h->obs_cam; P.obs_w = h->obs_w; P.obs_h = h->obs_h; P.obs_mode = h->obs_mode; P.obs_frames = h->obs_frames;
    P.rule_z_from_depth = rule->z_from_depth ? 1 : 0;
    const double cam[5] = {rule->cam_x0, rule->cam_y0, rule->cam_dx, rule->cam_dy, rule->cam_z};
    for (int k = 0; k < 5; k++) P.rule_cam[k] = cam[k];
    if (P.rule_z_from_depth && (cam[2] == 0 || cam[3] == 0)) return fail(UR5_ERR_ARG, "ur5_grasp_rounds_dev: z_from_depth needs the camera's pixel map (cam_dx, cam_dy != 0)");
  }
  return be_launch(h, P);
}
int ur5_set_observation_dev(ur5_sim* h, int camera_id, int width, int height, int depth_mode, uint8_t* rgb_dev, float* depth_dev, int frames) {
  UR5_FWD(set_observation_dev, (h, camera_id, width, height, depth_mode, rgb_dev, depth_dev, frames));
  using namespace ur5host;
  if (!h) return fail(UR5_ERR_ARG, "ur5_set_observation_dev: NULL handle");
  if (!rgb_dev) { h->obs_rgb = nullptr; h->obs_depth = nullptr; h->obs_frames = 0; return 0; }
  if (!depth_dev || width <= 0 || height <= 0 || frames < 1) return fail(UR5_ERR_ARG, "ur5_set_observation_dev: bad arguments");
  if (camera_id < 0 || camera_id >= h->hrm.ncam) return fail(UR5_ERR_ARG, "ur5_set_observation_dev: unknown camera id");
  if (!be_can_observe(h)) return fail(UR5_ERR_MODEL, "ur5_set_observation_dev: this scene's engine image has no room for the ray caster's working set (render with ur5_render_dev between launches)");
  h->obs_rgb = rgb_dev; h->obs_depth = depth_dev; h->obs_cam = camera_id; h->obs_w = width; h->obs_h = height; h->obs_mode = depth_mode; h->obs_frames = frames;
  return 0;
}
int ur5_grasp_attempt_dev(ur5_sim* h, const double* action_dev, int check_mode, double table_height, int* reward_dev) {
  return ur5_grasp_attempt_reset_dev(h, action_dev, check_mode, table_height, reward_dev, nullptr, 0.0);
}
int ur5_grasp_attempt(ur5_sim* h, const double* action, const uint8_t* skip, int check_mode, double table_height, int* reward, int* phase_steps, int* phase_result) {
  UR5_FWD(grasp_attempt, (h, action, skip, check_mode, table_height, reward, phase_steps, phase_result));
  using namespace ur5host;
  if (!action || !reward) return fail(UR5_ERR_ARG, "ur5_grasp_attempt: action/reward are required");
  std::vector<double> t((size_t)h->n * 8, 0.0);
  for (int e = 0; e < h->n; e++) { for (int k = 0; k < 4; k++) t[8 * e + k] = action[4 * e + k]; t[8 * e + 4] = (skip && skip[e]) ? 1.0 : 0.0; }
  int rc = upload(h, h->d_target, t.data(), t.size());
  if (rc) return rc;
  rc = ur5_grasp_attempt_dev(h, h->d_target, check_mode, table_height, h->d_result);
  if (!rc) rc = be_sync(h);
  if (!rc) rc = be_d2h(h, reward, h->d_result, (size_t)h->n * 4);
  if (!rc && phase_steps) rc = be_d2h(h, phase_steps, h->d_ps, (size_t)h->n * 48);
  if (!rc && phase_result) rc = be_d2h(h, phase_result, h->d_pr, (size_t)h->n * 48);
  return rc;
}
int ur5_ik(ur5_sim* h, const double* xyz, double* q5, int* result) {
  UR5_FWD(ik, (h, xyz, q5, result));
  using namespace ur5host;
  if (!xyz || !q5) return fail(UR5_ERR_ARG, "ur5_ik: xyz/q5 are required");
  std::vector<double> t((size_t)h->n * 8, 0.0);
  for (int e = 0; e < h->n; e++) for (int k = 0; k < 3; k++) t[8 * e + k] = xyz[3 * e + k];
  int rc = upload(h, h->d_target, t.data(), t.size());
  if (rc) return rc;
  Ur5Launch P = base_launch(h, UR5_OP_IK);
  P.target = h->d_target; P.out = h->d_target; P.result = h->d_result;
  rc = be_launch(h, P);
  if (!rc) rc = be_sync(h);
  if (!rc) rc = be_d2h(h, t.data(), h->d_target, t.size() * 8);
  if (!rc && result) rc = be_d2h(h, result, h->d_result, (size_t)h->n * 4);
  if (!rc) for (int e = 0; e < h->n; e++) for (int k = 0; k < 5; k++) q5[5 * e + k] = t[8 * e + k];
  return rc;
}
int ur5_render_dev(ur5_sim* h, int camera_id, int width, int height, int depth_mode, uint8_t* rgb_dev, float* depth_dev) {
  UR5_FWD(render_dev, (h, camera_id, width, height, depth_mode, rgb_dev, depth_dev));
  using namespace ur5host;
  if (!rgb_dev || !depth_dev || width <= 0 || height <= 0) return fail(UR5_ERR_ARG, "ur5_render_dev: bad arguments");
  if (camera_id < 0 || camera_id >= h->hrm.ncam) return fail(UR5_ERR_ARG, "ur5_render: unknown camera id");
  return be_render(h, camera_id, width, height, depth_mode, rgb_dev, depth_dev);
}
int ur5_render(ur5_sim* h, int camera_id, int width, int height, int depth_mode, uint8_t* rgb, float* depth) {
  UR5_FWD(render, (h, camera_id, width, height, depth_mode, rgb, depth));
  using namespace ur5host;
  if (!rgb || !depth || width <= 0 || height <= 0) return fail(UR5_ERR_ARG, "ur5_render: bad arguments");
  size_t px = (size_t)h->n * width * height;
  if (px > h->img_cap) {
    if (h->d_rgb) be_free(h, h->d_rgb);
    if (h->d_depth) be_free(h, h->d_depth);
    h->d_rgb = (uint8_t*)be_alloc(h, px * 3); h->d_depth = (float*)be_alloc(h, px * 4); h->img_cap = px;
    if (!h->d_rgb || !h->d_depth) { h->img_cap = 0; return fail(UR5_ERR_DEVICE, "image buffer allocation failed"); }
  }
  int rc = ur5_render_dev(h, camera_id, width, height, depth_mode, h->d_rgb, h->d_depth);
  if (!rc) rc = be_sync(h);
  if (!rc) rc = be_d2h(h, rgb, h->d_rgb, px * 3);
  if (!rc) rc = be_d2h(h, depth, h->d_depth, px * 4);
  return rc;
}
int ur5_set_order_dev(ur5_sim* h, const int* order_dev) {
  UR5_FWD(set_order_dev, (h, order_dev));
  if (!order_dev) { h->d_order = nullptr; return 0; }
  if (!h->d_order_buf) h->d_order_buf = (int*)be_alloc(h, (size_t)h->n * 4);
  if (!h->d_order_buf) return ur5host::fail(UR5_ERR_DEVICE, "device allocation failed (dispatch order)");
  // copied on the handle's stream: later launches read the copy, so the caller's buffer only has to stay valid until work queued so far has run
  int rc = be_d2d_async(h, h->d_order_buf, order_dev, (size_t)h->n * 4);
  if (rc) return rc;
  h->d_order = h->d_order_buf;
  return 0;
}
int ur5_set_order_view_dev(ur5_sim* h, const int* order_dev) {
  UR5_FWD(set_order_view_dev, (h, order_dev));
  h->d_order = order_dev;   // no copy: the caller's buffer is what the following launches read (include/ur5sim.h)
  return 0;
}
int ur5_sync(ur5_sim* h) {
  UR5_FWD(sync, (h)); return be_sync(h); }
int ur5_set_stream(ur5_sim* h, void* hip_stream, int external) {
  UR5_FWD(set_stream, (h, hip_stream, external)); return be_set_stream(h, hip_stream, external); }
double ur5_last_launch_ms(ur5_sim* h) {
  UR5_FWD(last_launch_ms, (h)); return h->last_ms; }
double ur5_kernel_ms_total(ur5_sim* h) {
  UR5_FWD(kernel_ms_total, (h)); return h->kernel_ms_total; }
void* ur5_state_device_ptr(ur5_sim* h) {
  UR5_FWD(state_device_ptr, (h)); return h->d_rec; }

long ur5_model_uploads(ur5_sim* h) {
  UR5_FWD(model_uploads, (h));
  return g_model_uploads;
}
int ur5_set_step_cap_dev(ur5_sim* h, const int* cap_dev) {
  UR5_FWD(set_step_cap_dev, (h, cap_dev));
  if (!h) return ur5host::fail(UR5_ERR_ARG, "ur5_set_step_cap_dev: NULL handle");
  if (!cap_dev) { h->d_step_cap = nullptr; return 0; }
  // copied on the handle's stream, like the dispatch order: a caller that frees its tensor without clearing the hook leaves no dangling pointer behind (round-5 advice)
  if (!h->d_step_cap_buf) h->d_step_cap_buf = (int*)be_alloc(h, (size_t)h->n * 4);
  if (!h->d_step_cap_buf) return ur5host::fail(UR5_ERR_DEVICE, "device allocation failed (step caps)");
  int rc = be_d2d_async(h, h->d_step_cap_buf, cap_dev, (size_t)h->n * 4);
  if (rc) return rc;
  h->d_step_cap = h->d_step_cap_buf;
  return 0;
}
int ur5_forward_debug(ur5_sim* h, double* out) {
  UR5_FWD(forward_debug, (h, out));
  using namespace ur5host;
  if (!h->d_debug) h->d_debug = (double*)be_alloc(h, (size_t)h->n * UR5_DEBUG_STRIDE * 8);
  if (!h->d_debug) return fail(UR5_ERR_DEVICE, "debug buffer allocation failed");
  Ur5Launch P = base_launch(h, UR5_OP_FORWARD);
  P.debug = h->d_debug;
  int rc = be_launch(h, P);
  if (!rc) rc = be_sync(h);
  if (!rc) rc = be_d2h(h, out, h->d_debug, (size_t)h->n * UR5_DEBUG_STRIDE * 8);
  return rc;
}
#ifdef UR5_PROFILE
// profile build only: per-env, per-phase cycle totals of the last launch ([n][18] host: 16 phases, shader cycles and 100 MHz ticks of the launch)
int ur5_profile_read(ur5_sim* h, double* out) {
  UR5_FWD(profile_read, (h, out));
  std::vector<double> dbg((size_t)h->n * UR5_DEBUG_STRIDE);
  int rc = be_d2h(h, dbg.data(), h->d_debug, dbg.size() * 8);
  if (rc) return rc;
  for (int e = 0; e < h->n; e++) memcpy(out + (size_t)e * 26, dbg.data() + (size_t)e * UR5_DEBUG_STRIDE, 26 * 8);
  return 0;
}
#endif
int ur5_body_xpos(ur5_sim* h, double* out) {
  UR5_FWD(body_xpos, (h, out));
  std::vector<double> dbg((size_t)h->n * UR5_DEBUG_STRIDE);
  int rc = ur5_forward_debug(h, dbg.data());
  if (rc) return rc;
  for (int e = 0; e < h->n; e++) memcpy(out + (size_t)e * UR5_MAXB * 3, dbg.data() + (size_t)e * UR5_DEBUG_STRIDE + 8, UR5_MAXB * 3 * 8);
  return 0;
}
}  // extern "C"
