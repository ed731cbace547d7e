// =====================================================================================================
// ur5_oracle.cpp -- CPU (fp64, single env, scalar) RESTATEMENT of the reference's hot path.
//
//   *** TEST INFRASTRUCTURE ONLY. ***  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
//   may load this library. The product (mujoco_rl_ur5_amd/csrc) never includes, links or calls it.
//
//   *** PARITY UNPINNED ***  The reference (PaulDanielML/MuJoCo_RL_UR5) carries no golden vectors, and all of
//   its arithmetic lives in un-vendored, unpinned third-party code that is absent here: MuJoCo 2.0 via
//   mujoco_py (sim.step), simple_pid (PID.__call__), ikpy (inverse_kinematics). This file restates their
//   PUBLISHED algorithms (MuJoCo documentation "Computation" chapter; simple_pid README) and anchors on the
//   reference's own call sites. It is pinned only by analytic known answers (tests/test_oracle_*.py) and by
//   the one recorded vector of media/console.png (pixel (136,80) -> world), see SURVEY.md section 8c.
//
// What follows which reference line (all under /root/reference):
//   sim.step()                        gym_grasper/controller/MujocoController.py:379   -> Sim::step()  [3P mj_step]
//   simple_pid.PID(...) / __call__    MujocoController.py:157-235, :326                -> Pid, Sim::pid_eval()
//   move_group_to_joint_target        MujocoController.py:269-393 (loop :318-382)      -> Sim::move_group()
//   open/close_gripper, grasp         MujocoController.py:408-444                      -> Sim::open_gripper() ...
//   move_ee / ik                      MujocoController.py:446-517                      -> Sim::move_ee(), Sim::ik()
//   stay                              MujocoController.py:621-636                      -> Sim::stay()   (H2: fixed steps)
//   move_and_grasp                    gym_grasper/envs/GraspingEnv.py:205-386          -> Sim::grasp_attempt()
//   reset_model                       GraspingEnv.py:409-477 (IT4 variant :435-463)    -> Sim::reset()
// Determinism knobs: pid_dt = timestep; stay(ms) = ceil(ms/1000/h/10) chunks of 10 steps.
// Documented deviations from MuJoCo: DESIGN.md section "Deviations".
// =====================================================================================================
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>

namespace {

// The control twin "fmadyn" of tools/pile_divergence_time.py (oracle/Makefile libur5_oracle_fmadyn.so: clang, -mfma -ffp-contract=fast-honor-pragmas) compiles the
// GEOMETRY of this text -- the helpers below, kinematics, every collide_* routine -- without fused multiply-adds and everything else with them: the split of the HIP
// pile unit (csrc/ur5_engine.h UR5_STRICT). In the oracle proper (g++ -ffp-contract=off) nothing is fused anywhere and the two macros are empty.
#if defined(__clang__) && defined(UR5O_FMA_DYNAMICS)
#pragma clang fp contract(off)
#define UR5O_STRICT _Pragma("clang fp contract(off)")
#else
#define UR5O_STRICT
#endif

// ------------------------------------------------------------------------------------------ small maths
struct V3 {
  double x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  double& operator[](int i) { return (&x)[i]; }
  double operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(double s, V3 a) { return a * s; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
inline V3 normalized(V3 a) {
  double n = norm(a);
  return n > 1e-300 ? a * (1.0 / n) : V3(1, 0, 0);
}

struct Q4 { double w, x, y, z; };
inline Q4 qmul(Q4 a, Q4 b) {
  return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
inline Q4 qnormalize(Q4 q) {
  double n = std::sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < 1e-300) return Q4{1, 0, 0, 0};
  return Q4{q.w / n, q.x / n, q.y / n, q.z / n};
}
inline Q4 qaxisangle(V3 axis, double ang) {
  double s = std::sin(0.5 * ang);
  return Q4{std::cos(0.5 * ang), axis.x * s, axis.y * s, axis.z * s};
}
struct M3 {  // row-major; columns are the frame axes
  double m[9];
  V3 col(int j) const { return V3(m[j], m[3 + j], m[6 + j]); }
  V3 row(int i) const { return V3(m[3 * i], m[3 * i + 1], m[3 * i + 2]); }
};
inline M3 qmat(Q4 q) {
  double w = q.w, x = q.x, y = q.y, z = q.z;
  M3 r;
  r.m[0] = w * w + x * x - y * y - z * z; r.m[1] = 2 * (x * y - w * z); r.m[2] = 2 * (x * z + w * y);
  r.m[3] = 2 * (x * y + w * z); r.m[4] = w * w - x * x + y * y - z * z; r.m[5] = 2 * (y * z - w * x);
  r.m[6] = 2 * (x * z - w * y); r.m[7] = 2 * (y * z + w * x); r.m[8] = w * w - x * x - y * y + z * z;
  return r;
}
inline V3 mul(const M3& a, V3 v) { return V3(dot(a.row(0), v), dot(a.row(1), v), dot(a.row(2), v)); }
inline V3 mulT(const M3& a, V3 v) { return V3(dot(a.col(0), v), dot(a.col(1), v), dot(a.col(2), v)); }
inline M3 matmul(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
  return r;
}

// spatial 6-vectors: [rot(3); lin(3)]  (MuJoCo convention, [3P])
struct S6 {
  V3 r, l;
};
inline S6 operator+(S6 a, S6 b) { return S6{a.r + b.r, a.l + b.l}; }
inline S6 operator*(S6 a, double s) { return S6{a.r * s, a.l * s}; }
inline double dot(S6 a, S6 b) { return dot(a.r, b.r) + dot(a.l, b.l); }
inline S6 crossMotion(S6 v, S6 m) { return S6{cross(v.r, m.r), cross(v.r, m.l) + cross(v.l, m.r)}; }
inline S6 crossForce(S6 v, S6 f) { return S6{cross(v.r, f.r) + cross(v.l, f.l), cross(v.r, f.l)}; }
// rigid-body inertia about a reference point o, world axes: I (sym 3x3, 6 numbers xx yy zz xy xz yz), h = m (c - o), m
struct Inert {
  double I[6];
  V3 h;
  double m;
};
inline Inert operator+(const Inert& a, const Inert& b) {
  Inert r;
  for (int i = 0; i < 6; i++) r.I[i] = a.I[i] + b.I[i];
  r.h = a.h + b.h;
  r.m = a.m + b.m;
  return r;
}
inline V3 symmul(const double* I, V3 v) {
  return V3(I[0] * v.x + I[3] * v.y + I[4] * v.z, I[3] * v.x + I[1] * v.y + I[5] * v.z, I[4] * v.x + I[5] * v.y + I[2] * v.z);
}
inline S6 mulInert(const Inert& a, S6 v) {  // spatial momentum / force
  return S6{symmul(a.I, v.r) + cross(a.h, v.l), v.l * a.m - cross(a.h, v.r)};
}

// ------------------------------------------------------------------------------------------ model blob
enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_CYLINDER = 5, GEOM_BOX = 6, GEOM_MESH = 7 };
enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
const double MINVAL = 1e-15;

struct Model {
  std::vector<char> blob;
  int nq, nv, nu, nbody, njnt, ngeom, npair, neq, ntree, ncam;
  const int *body_parentid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum, *body_weldid, *body_treeid;
  const double *body_pos, *body_quat, *body_mass, *body_ipos, *body_inertia, *body_invweight0;
  const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_range, *qpos0;
  const int *dof_bodyid, *dof_jntid, *dof_parentid, *dof_treeid, *tree_dofadr, *tree_dofnum;
  const double *dof_armature, *dof_damping, *dof_invweight0;
  const int *geom_type, *geom_bodyid, *geom_condim, *geom_meshid, *geom_collide;
  const double *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_margin, *geom_solref, *geom_solimp, *geom_rbound;
  const int *mesh_vertadr, *mesh_vertnum;
  const double* mesh_vert;
  const int *pair_geom1, *pair_geom2, *eq_jnt1, *eq_jnt2, *vis_planeadr, *vis_planenum;
  const double *vis_plane, *geom_rgba;
  double extent, znear, zfar;
  const double *eq_polycoef, *eq_solref, *eq_solimp;
  const int *act_jntid, *act_ctrllimited;
  const double *act_gear, *act_ctrlrange, *cam_pos, *cam_mat, *cam_fovy;
  double timestep, tolerance, impratio, gravity[3], jnt_solref[2], jnt_solimp[5], meaninertia;
  int iterations;
  std::vector<int> body_rootid;
  std::vector<V3> mesh_center;  // mean of hull vertices (geom frame)

  const void* find(const char* name, int code, int* count) const {
    const char* p = blob.data();
    uint64_t n;
    memcpy(&n, p + 8, 8);
    size_t off = 16;
    for (uint64_t i = 0; i < n; i++) {
      char nm[33];
      memcpy(nm, p + off, 32);
      nm[32] = 0;
      uint32_t c, cnt;
      memcpy(&c, p + off + 32, 4);
      memcpy(&cnt, p + off + 36, 4);
      off += 40;
      size_t bytes = (c == 0) ? 8ull * cnt : (c == 1 ? 4ull * cnt : cnt);
      if (strcmp(nm, name) == 0 && (int)c == code) {
        *count = (int)cnt;
        return p + off;
      }
      off += bytes + ((8 - bytes % 8) % 8);
    }
    fprintf(stderr, "ur5_oracle: section %s missing\n", name);
    abort();
  }
  const double* F(const char* n, int* c = nullptr) const {
    int k;
    const void* p = find(n, 0, &k);
    if (c) *c = k;
    return (const double*)p;
  }
  const int* I(const char* n, int* c = nullptr) const {
    int k;
    const void* p = find(n, 1, &k);
    if (c) *c = k;
    return (const int*)p;
  }
  bool load(const void* data, size_t nbytes) {
    blob.assign((const char*)data, (const char*)data + nbytes);
    if (nbytes < 16 || memcmp(blob.data(), "UR5MODL1", 8) != 0) return false;
    body_parentid = I("body_parentid", &nbody);
    body_pos = F("body_pos"); body_quat = F("body_quat");
    body_jntadr = I("body_jntadr"); body_jntnum = I("body_jntnum"); body_dofadr = I("body_dofadr"); body_dofnum = I("body_dofnum");
    body_weldid = I("body_weldid"); body_treeid = I("body_treeid");
    body_mass = F("body_mass"); body_ipos = F("body_ipos"); body_inertia = F("body_inertia"); body_invweight0 = F("body_invweight0");
    jnt_type = I("jnt_type", &njnt); jnt_qposadr = I("jnt_qposadr"); jnt_dofadr = I("jnt_dofadr"); jnt_bodyid = I("jnt_bodyid");
    jnt_pos = F("jnt_pos"); jnt_axis = F("jnt_axis"); jnt_limited = I("jnt_limited"); jnt_range = F("jnt_range");
    qpos0 = F("qpos0", &nq);
    dof_bodyid = I("dof_bodyid", &nv); dof_jntid = I("dof_jntid"); dof_parentid = I("dof_parentid");
    dof_armature = F("dof_armature"); dof_damping = F("dof_damping"); dof_treeid = I("dof_treeid"); dof_invweight0 = F("dof_invweight0");
    tree_dofadr = I("tree_dofadr", &ntree); tree_dofnum = I("tree_dofnum");
    geom_type = I("geom_type", &ngeom); geom_bodyid = I("geom_bodyid"); geom_size = F("geom_size"); geom_pos = F("geom_pos");
    geom_quat = F("geom_quat"); geom_friction = F("geom_friction"); geom_condim = I("geom_condim"); geom_margin = F("geom_margin");
    geom_solref = F("geom_solref"); geom_solimp = F("geom_solimp"); geom_meshid = I("geom_meshid"); geom_rbound = F("geom_rbound");
    geom_collide = I("geom_collide");
    int nmesh;
    mesh_vertadr = I("mesh_vertadr", &nmesh); mesh_vertnum = I("mesh_vertnum"); mesh_vert = F("mesh_vert");
    pair_geom1 = I("pair_geom1", &npair); pair_geom2 = I("pair_geom2");
    eq_jnt1 = I("eq_jnt1", &neq); eq_jnt2 = I("eq_jnt2"); eq_polycoef = F("eq_polycoef"); eq_solref = F("eq_solref"); eq_solimp = F("eq_solimp");
    act_jntid = I("act_jntid", &nu); act_gear = F("act_gear"); act_ctrlrange = F("act_ctrlrange"); act_ctrllimited = I("act_ctrllimited");
    cam_fovy = F("cam_fovy", &ncam); cam_pos = F("cam_pos"); cam_mat = F("cam_mat");
    const double* of = F("opt_f");
    timestep = of[0]; tolerance = of[1]; impratio = of[2];
    for (int i = 0; i < 3; i++) gravity[i] = of[3 + i];
    for (int i = 0; i < 2; i++) jnt_solref[i] = of[6 + i];
    for (int i = 0; i < 5; i++) jnt_solimp[i] = of[8 + i];
    meaninertia = of[13]; extent = of[14]; znear = of[15]; zfar = of[16];
    vis_planeadr = I("vis_planeadr"); vis_planenum = I("vis_planenum"); vis_plane = F("vis_plane"); geom_rgba = F("geom_rgba");
    iterations = I("opt_i")[0];
    body_rootid.assign(nbody, 0);
    for (int b = 1; b < nbody; b++) body_rootid[b] = body_parentid[b] == 0 ? b : body_rootid[body_parentid[b]];
    mesh_center.assign(nmesh, V3());
    for (int k = 0; k < nmesh; k++) {
      V3 c;
      for (int i = 0; i < mesh_vertnum[k]; i++) {
        const double* v = mesh_vert + 3 * (mesh_vertadr[k] + i);
        c = c + V3(v[0], v[1], v[2]);
      }
      mesh_center[k] = c * (1.0 / std::max(1, mesh_vertnum[k]));
    }
    return true;
  }
};
inline V3 v3(const double* p) { return V3(p[0], p[1], p[2]); }
inline Q4 q4(const double* p) { return Q4{p[0], p[1], p[2], p[3]}; }

// ------------------------------------------------------------------------------------------ contacts / rows
struct Contact {
  double dist;
  V3 pos;
  V3 frame[3];  // normal (geom1 -> geom2), tangent1, tangent2
  int geom1, geom2, dim;
  double includemargin;
  double friction[5];
  double solref[2], solimp[5];
  double mu;
  int efc_address;
  int color;
};

struct Row {
  std::vector<int> idx;     // dofs of every touched tree
  std::vector<double> J, B;  // B = M^-1 J^T
  double pos, margin, vel, aref, R, D, diagApprox, Adiag;
  double force;
  bool unilateral;
  int contact;  // owning contact or -1
};

// simple_pid.PID with Ki = 0 handled generally ([3P] semantics restated from SURVEY.md Appendix A)
struct Pid {
  double Kp, Ki, Kd, setpoint, lo, hi;
  double integral, last_input, last_output;
  bool has_last;
  double eval(double input, double dt) {
    double error = setpoint - input;
    double d_input = has_last ? input - last_input : 0.0;
    double p = Kp * error;
    integral += Ki * error * dt;
    integral = std::min(std::max(integral, lo), hi);
    double d = -Kd * d_input / dt;
    double out = std::min(std::max(p + integral + d, lo), hi);
    last_input = input;
    has_last = true;
    last_output = out;
    return out;
  }
};

enum Result { RES_SUCCESS = 0, RES_MAX_STEPS = 1, RES_IK_FAIL = 2 };

// splitmix64 -> uniform double in [0,1): per-env deterministic stream
struct SplitMix {
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double uniform(double lo, double hi) { return lo + (hi - lo) * uniform(); }
};

// ------------------------------------------------------------------------------------------ convex support shapes
struct Shape {
  int type;
  V3 pos;
  M3 mat;
  V3 size;
  const double* verts;
  int nvert;
  V3 center;  // world-space interior point
  double margin;
};

V3 support(const Shape& s, V3 dir) {  // dir is unit length
  V3 d = mulT(s.mat, dir);
  V3 l;
  switch (s.type) {
    case GEOM_SPHERE: l = d * s.size.x; break;
    case GEOM_BOX: l = V3(d.x >= 0 ? s.size.x : -s.size.x, d.y >= 0 ? s.size.y : -s.size.y, d.z >= 0 ? s.size.z : -s.size.z); break;
    case GEOM_CAPSULE: l = d * s.size.x + V3(0, 0, d.z >= 0 ? s.size.y : -s.size.y); break;
    case GEOM_CYLINDER: {
      double n = std::sqrt(d.x * d.x + d.y * d.y);
      l = n > 1e-12 ? V3(d.x / n * s.size.x, d.y / n * s.size.x, 0) : V3();
      l.z = d.z >= 0 ? s.size.y : -s.size.y;
    } break;
    case GEOM_MESH: {
      double best = -1e300;
      int bi = 0;
      for (int i = 0; i < s.nvert; i++) {
        double v = s.verts[3 * i] * d.x + s.verts[3 * i + 1] * d.y + s.verts[3 * i + 2] * d.z;
        if (v > best) { best = v; bi = i; }
      }
      l = V3(s.verts[3 * bi], s.verts[3 * bi + 1], s.verts[3 * bi + 2]);
    } break;
    default: l = V3();
  }
  return s.pos + mul(s.mat, l) + dir * (0.5 * s.margin);
}

struct MV {  // Minkowski-difference vertex (shape1 - shape2) with its witnesses
  V3 v, a, b;
};
inline MV msupport(const Shape& A, const Shape& B, V3 dir) {
  MV r;
  r.a = support(A, dir);
  r.b = support(B, -dir);
  r.v = r.a - r.b;
  return r;
}

// Minkowski Portal Refinement penetration query (own restatement of the XenoCollide/MPR scheme that MuJoCo 2.0
// reaches through libccd [3P]; tolerance 1e-6 and 50 iterations are MuJoCo's mpr_tolerance / mpr_iterations).
// Returns true when the margin-inflated shapes overlap; depth along dir (dir pushes B away from A).
bool mpr_penetration(const Shape& A, const Shape& B, double* depth, V3* dir_out, V3* pos_out) {
  const double tol = 1e-6;
  const int maxit = 50;
  MV v0, v1, v2, v3, v4;
  v0.a = A.center; v0.b = B.center; v0.v = v0.a - v0.b;
  if (norm(v0.v) < 1e-12) v0.v = V3(1e-5, 0, 0);
  V3 dir = normalized(-v0.v);
  v1 = msupport(A, B, dir);
  if (dot(v1.v, dir) <= 0) return false;
  dir = cross(v0.v, v1.v);
  if (norm(dir) < 1e-12 * std::max(1.0, norm(v0.v) * norm(v1.v))) {
    // origin lies on the v0-v1 ray: penetration along that ray
    V3 d = normalized(-v0.v);
    *depth = dot(v1.v, d);
    *dir_out = d;
    *pos_out = (v1.a + v1.b) * 0.5;
    return true;
  }
  dir = normalized(dir);
  v2 = msupport(A, B, dir);
  if (dot(v2.v, dir) <= 0) return false;
  dir = normalized(cross(v1.v - v0.v, v2.v - v0.v));
  if (dot(dir, v0.v) > 0) {
    std::swap(v1, v2);
    dir = -dir;
  }
  for (int it = 0;; it++) {
    if (it > maxit) return false;
    v3 = msupport(A, B, dir);
    if (dot(v3.v, dir) <= 0) return false;
    bool cont = false;
    if (dot(cross(v1.v, v3.v), v0.v) < 0) { v2 = v3; cont = true; }
    else if (dot(cross(v3.v, v2.v), v0.v) < 0) { v1 = v3; cont = true; }
    if (!cont) break;
    dir = normalized(cross(v1.v - v0.v, v2.v - v0.v));
  }
  bool hit = false;
  for (int it = 0;; it++) {
    dir = normalized(cross(v2.v - v1.v, v3.v - v1.v));
    if (dot(dir, v0.v) > 0) dir = -dir;  // keep pointing away from the interior point
    double d1 = dot(v1.v, dir);
    if (d1 >= 0) hit = true;
    v4 = msupport(A, B, dir);
    double d4 = dot(v4.v, dir);
    if (!hit && d4 < 0) return false;
    if (d4 - d1 <= tol || it >= maxit) {
      if (!hit) return false;
      // barycentric coordinates of the origin's projection on the portal plane
      V3 p = dir * d1;
      V3 e1 = v2.v - v1.v, e2 = v3.v - v1.v, ep = p - v1.v;
      double a11 = dot(e1, e1), a12 = dot(e1, e2), a22 = dot(e2, e2), b1 = dot(ep, e1), b2 = dot(ep, e2);
      double det = a11 * a22 - a12 * a12;
      double w2 = 1.0 / 3, w3 = 1.0 / 3;
      if (std::fabs(det) > 1e-30) { w2 = (a22 * b1 - a12 * b2) / det; w3 = (a11 * b2 - a12 * b1) / det; }
      double w1 = 1.0 - w2 - w3;
      w1 = std::max(w1, 0.0); w2 = std::max(w2, 0.0); w3 = std::max(w3, 0.0);
      double ws = w1 + w2 + w3;
      if (ws < 1e-30) { w1 = w2 = w3 = 1.0 / 3; ws = 1; }
      w1 /= ws; w2 /= ws; w3 /= ws;
      V3 pa = v1.a * w1 + v2.a * w2 + v3.a * w3, pb = v1.b * w1 + v2.b * w2 + v3.b * w3;
      *depth = d1;
      *dir_out = dir;
      *pos_out = (pa + pb) * 0.5;
      return true;
    }
    V3 c = cross(v4.v, v0.v);
    if (dot(v1.v, c) > 0) {
      if (dot(v2.v, c) > 0) v1 = v4; else v3 = v4;
    } else {
      if (dot(v3.v, c) > 0) v2 = v4; else v1 = v4;
    }
  }
}

#if defined(__clang__) && defined(UR5O_FMA_DYNAMICS)
#pragma clang fp contract(fast)
#endif
// ------------------------------------------------------------------------------------------ the simulator
struct Sim {
  Model M;
  int nq, nv, nu;
  // persistent state
  std::vector<double> qpos, qvel, qacc_warmstart, ctrl;
  double time;
  std::vector<Pid> pid;
  std::vector<double> target;  // current_target_joint_values (MujocoController.py:236-241)
  int last_steps;
  long total_steps;
  // config
  double pid_dt;
  int contacts_enabled;
  int solver_iter_last;
  long solver_iter_total;
  // derived
  std::vector<V3> xpos, xipos, subtree_com, xanchor, xaxis, gxpos;
  std::vector<Q4> xquat;
  std::vector<M3> xmat, gxmat;
  std::vector<Inert> cinert, crb;
  std::vector<S6> cdof, cdof_dot, cvel, cacc, cfrc;
  std::vector<double> Mm, L, Ld;  // dense mass matrix, its L^T D L factor (L unit lower in tree order), Ld second factor (M + hB)
  std::vector<double> qfrc_bias, qfrc_passive, qfrc_actuator, qfrc_smooth, qacc_smooth, qacc, qfrc_constraint;
  std::vector<Contact> contacts;
  std::vector<Row> rows;
  std::vector<int> order;  // PGS sweep order over rows
  int ikfail_count;

  bool init(const void* blob, size_t n) {
    if (!M.load(blob, n)) return false;
    nq = M.nq; nv = M.nv; nu = M.nu;
    qpos.assign(M.qpos0, M.qpos0 + nq);
    qvel.assign(nv, 0); qacc_warmstart.assign(nv, 0); ctrl.assign(nu, 0);
    time = 0; last_steps = 0; total_steps = 0;
    pid_dt = M.timestep;
    contacts_enabled = 1;
    solver_iter_last = 0; solver_iter_total = 0; ikfail_count = 0;
    xpos.resize(M.nbody); xipos.resize(M.nbody); subtree_com.resize(M.nbody); xquat.resize(M.nbody); xmat.resize(M.nbody);
    xanchor.resize(M.njnt); xaxis.resize(M.njnt); gxpos.resize(M.ngeom); gxmat.resize(M.ngeom);
    cinert.resize(M.nbody); crb.resize(M.nbody); cvel.resize(M.nbody); cacc.resize(M.nbody); cfrc.resize(M.nbody);
    cdof.resize(nv); cdof_dot.resize(nv);
    Mm.assign((size_t)nv * nv, 0); L.assign((size_t)nv * nv, 0); Ld.assign((size_t)nv * nv, 0);
    qfrc_bias.assign(nv, 0); qfrc_passive.assign(nv, 0); qfrc_actuator.assign(nv, 0); qfrc_smooth.assign(nv, 0);
    qacc_smooth.assign(nv, 0); qacc.assign(nv, 0); qfrc_constraint.assign(nv, 0);
    init_pids();
    forward_position();
    return true;
  }

  // MujocoController.py:157-247 (create_lists): gains, limits, setpoints; each PID is called once with input 0.
  void init_pids() {
    static const double kp[7] = {7 * 3.0, 10 * 3.0, 5 * 3.0, 7 * 3.0, 5 * 3.0, 5 * 3.0, 2.5 * 3.0};
    static const double kd[7] = {1.1 * 0.1, 1.0 * 0.1, 0.5 * 0.1, 0.1 * 0.1, 0.1 * 0.1, 0.1 * 0.1, 0.0};
    static const double sp[7] = {0, -1.57, 1.57, -1.57, -1.57, 0, 0};
    static const double lim[7] = {2, 2, 2, 1, 1, 1, 1};
    pid.resize(nu);
    target.assign(nu, 0);
    for (int j = 0; j < nu && j < 7; j++) {
      pid[j] = Pid{kp[j], 0.0, kd[j], sp[j], -lim[j], lim[j], 0.0, 0.0, 0.0, false};
      target[j] = sp[j];
      pid[j].eval(0.0, pid_dt);  // current_output = [controller(0) ...]  (MujocoController.py:247)
    }
  }

  // ------------------------------------------------------------------ mj_kinematics + mj_comPos  [3P, SURVEY C.1/C.2]
  void kinematics() { UR5O_STRICT;
    xpos[0] = V3(); xquat[0] = Q4{1, 0, 0, 0}; xmat[0] = qmat(xquat[0]); xipos[0] = V3();
    for (int b = 1; b < M.nbody; b++) {
      int p = M.body_parentid[b];
      V3 pos = xpos[p] + mul(xmat[p], v3(M.body_pos + 3 * b));
      Q4 quat = qmul(xquat[p], q4(M.body_quat + 4 * b));
      for (int j = M.body_jntadr[b]; j < M.body_jntadr[b] + M.body_jntnum[b]; j++) {
        int qa = M.jnt_qposadr[j], t = M.jnt_type[j];
        if (t == JNT_FREE) {
          pos = V3(qpos[qa], qpos[qa + 1], qpos[qa + 2]);
          quat = qnormalize(Q4{qpos[qa + 3], qpos[qa + 4], qpos[qa + 5], qpos[qa + 6]});
          xanchor[j] = pos; xaxis[j] = V3(0, 0, 1);
          continue;
        }
        M3 Rb = qmat(quat);
        xanchor[j] = pos + mul(Rb, v3(M.jnt_pos + 3 * j));
        xaxis[j] = mul(Rb, v3(M.jnt_axis + 3 * j));
        if (t == JNT_SLIDE) {
          pos = pos + xaxis[j] * (qpos[qa] - M.qpos0[qa]);
        } else if (t == JNT_HINGE) {
          quat = qmul(quat, qaxisangle(v3(M.jnt_axis + 3 * j), qpos[qa] - M.qpos0[qa]));
          pos = xanchor[j] - mul(qmat(quat), v3(M.jnt_pos + 3 * j));
        } else {  // ball
          quat = qmul(quat, qnormalize(Q4{qpos[qa], qpos[qa + 1], qpos[qa + 2], qpos[qa + 3]}));
          pos = xanchor[j] - mul(qmat(quat), v3(M.jnt_pos + 3 * j));
        }
      }
      xquat[b] = qnormalize(quat);
      xpos[b] = pos;
      xmat[b] = qmat(xquat[b]);
      xipos[b] = pos + mul(xmat[b], v3(M.body_ipos + 3 * b));
    }
    for (int g = 0; g < M.ngeom; g++) {
      int b = M.geom_bodyid[g];
      gxpos[g] = xpos[b] + mul(xmat[b], v3(M.geom_pos + 3 * g));
      gxmat[g] = matmul(xmat[b], qmat(q4(M.geom_quat + 4 * g)));
    }
  }

  void com_pos() {
    std::vector<double> mass(M.nbody, 0);
    for (int b = 0; b < M.nbody; b++) { subtree_com[b] = xipos[b] * M.body_mass[b]; mass[b] = M.body_mass[b]; }
    for (int b = M.nbody - 1; b > 0; b--) {
      int p = M.body_parentid[b];
      subtree_com[p] = subtree_com[p] + subtree_com[b];
      mass[p] += mass[b];
    }
    for (int b = 0; b < M.nbody; b++) subtree_com[b] = mass[b] > MINVAL ? subtree_com[b] * (1.0 / mass[b]) : xipos[b];
    for (int b = 1; b < M.nbody; b++) {
      V3 o = subtree_com[M.body_rootid[b]];
      // world-frame inertia about the body com, then shifted to o
      const double* bi = M.body_inertia + 6 * b;
      M3 Ib;
      Ib.m[0] = bi[0]; Ib.m[4] = bi[1]; Ib.m[8] = bi[2]; Ib.m[1] = Ib.m[3] = bi[3]; Ib.m[2] = Ib.m[6] = bi[4]; Ib.m[5] = Ib.m[7] = bi[5];
      M3 R = xmat[b], Rt;
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt.m[3 * i + j] = R.m[3 * j + i];
      M3 Iw = matmul(matmul(R, Ib), Rt);
      double m = M.body_mass[b];
      V3 c = xipos[b] - o;
      Inert in;
      in.I[0] = Iw.m[0] + m * (c.y * c.y + c.z * c.z);
      in.I[1] = Iw.m[4] + m * (c.x * c.x + c.z * c.z);
      in.I[2] = Iw.m[8] + m * (c.x * c.x + c.y * c.y);
      in.I[3] = Iw.m[1] - m * c.x * c.y;
      in.I[4] = Iw.m[2] - m * c.x * c.z;
      in.I[5] = Iw.m[5] - m * c.y * c.z;
      in.h = c * m;
      in.m = m;
      cinert[b] = in;
    }
    cinert[0] = Inert{{0, 0, 0, 0, 0, 0}, V3(), 0};
    for (int j = 0; j < M.njnt; j++) {
      int b = M.jnt_bodyid[j], d = M.jnt_dofadr[j], t = M.jnt_type[j];
      V3 off = subtree_com[M.body_rootid[b]] - xanchor[j];
      if (t == JNT_SLIDE) cdof[d] = S6{V3(), xaxis[j]};
      else if (t == JNT_HINGE) cdof[d] = S6{xaxis[j], cross(xaxis[j], off)};
      else {
        if (t == JNT_FREE) {
          cdof[d] = S6{V3(), V3(1, 0, 0)}; cdof[d + 1] = S6{V3(), V3(0, 1, 0)}; cdof[d + 2] = S6{V3(), V3(0, 0, 1)};
          d += 3;
        }
        for (int k = 0; k < 3; k++) { V3 ax = xmat[b].col(k); cdof[d + k] = S6{ax, cross(ax, off)}; }
      }
    }
  }

  // ------------------------------------------------------------------ mj_crb + mj_factorM (L^T D L in tree order)  [3P]
  void crb_and_factor() {
    for (int b = 0; b < M.nbody; b++) crb[b] = cinert[b];
    for (int b = M.nbody - 1; b > 0; b--) crb[M.body_parentid[b]] = crb[M.body_parentid[b]] + crb[b];
    std::fill(Mm.begin(), Mm.end(), 0.0);
    for (int i = 0; i < nv; i++) {
      S6 buf = mulInert(crb[M.dof_bodyid[i]], cdof[i]);
      for (int j = i; j >= 0; j = M.dof_parentid[j]) {
        double v = dot(cdof[j], buf);
        Mm[(size_t)i * nv + j] = v;
        Mm[(size_t)j * nv + i] = v;
      }
      Mm[(size_t)i * nv + i] += M.dof_armature[i];
    }
    factor(Mm, L, 0.0);
  }
  // out = L^T D L factor of (A + h*diag(damping)); out[k][k] = D_k, out[k][i] (i ancestor of k) = L_ki
  void factor(const std::vector<double>& A, std::vector<double>& out, double h) {
    out = A;
    if (h != 0.0) for (int i = 0; i < nv; i++) out[(size_t)i * nv + i] += h * M.dof_damping[i];
    for (int k = nv - 1; k >= 0; k--) {
      double dk = out[(size_t)k * nv + k];
      for (int i = M.dof_parentid[k]; i >= 0; i = M.dof_parentid[i]) {
        double a = out[(size_t)k * nv + i] / dk;
        for (int j = i; j >= 0; j = M.dof_parentid[j]) out[(size_t)i * nv + j] -= a * out[(size_t)k * nv + j];
        out[(size_t)k * nv + i] = a;
      }
    }
  }
  // x <- (L^T D L)^-1 x, restricted to dofs [lo, hi)
  void solve(const std::vector<double>& F, double* x, int lo, int hi) const {
    for (int i = hi - 1; i >= lo; i--)
      for (int j = M.dof_parentid[i]; j >= 0; j = M.dof_parentid[j]) x[j] -= F[(size_t)i * nv + j] * x[i];
    for (int i = lo; i < hi; i++) x[i] /= F[(size_t)i * nv + i];
    for (int i = lo; i < hi; i++)
      for (int j = M.dof_parentid[i]; j >= 0; j = M.dof_parentid[j]) x[i] -= F[(size_t)i * nv + j] * x[j];
  }

  // ------------------------------------------------------------------ mj_comVel + mj_rne (bias) + passive  [3P]
  void velocity_stage() {
    cvel[0] = S6{V3(), V3()};
    for (int b = 1; b < M.nbody; b++) {
      S6 v = cvel[M.body_parentid[b]];
      for (int j = M.body_jntadr[b]; j < M.body_jntadr[b] + M.body_jntnum[b]; j++) {
        int d = M.jnt_dofadr[j], t = M.jnt_type[j];
        if (t == JNT_SLIDE || t == JNT_HINGE) {
          cdof_dot[d] = crossMotion(v, cdof[d]);
          v = v + cdof[d] * qvel[d];
        } else {
          if (t == JNT_FREE) {
            for (int k = 0; k < 3; k++) { cdof_dot[d + k] = S6{V3(), V3()}; }
            for (int k = 0; k < 3; k++) v = v + cdof[d + k] * qvel[d + k];
            d += 3;
          }
          for (int k = 0; k < 3; k++) cdof_dot[d + k] = crossMotion(v, cdof[d + k]);
          for (int k = 0; k < 3; k++) v = v + cdof[d + k] * qvel[d + k];
        }
      }
      cvel[b] = v;
    }
    cacc[0] = S6{V3(), V3(-M.gravity[0], -M.gravity[1], -M.gravity[2])};
    for (int b = 1; b < M.nbody; b++) {
      S6 a = cacc[M.body_parentid[b]];
      for (int d = M.body_dofadr[b]; d >= 0 && d < M.body_dofadr[b] + M.body_dofnum[b]; d++) a = a + cdof_dot[d] * qvel[d];
      cacc[b] = a;
      cfrc[b] = mulInert(cinert[b], a) + crossForce(cvel[b], mulInert(cinert[b], cvel[b]));
    }
    cfrc[0] = S6{V3(), V3()};
    for (int b = M.nbody - 1; b > 0; b--) cfrc[M.body_parentid[b]] = cfrc[M.body_parentid[b]] + cfrc[b];
    for (int i = 0; i < nv; i++) {
      qfrc_bias[i] = dot(cdof[i], cfrc[M.dof_bodyid[i]]);
      qfrc_passive[i] = -M.dof_damping[i] * qvel[i];
    }
  }

  // ------------------------------------------------------------------ mj_fwdActuation (motors) [3P]; MujocoController.py:327
  void actuation() {
    std::fill(qfrc_actuator.begin(), qfrc_actuator.end(), 0.0);
    for (int a = 0; a < nu; a++) {
      double c = ctrl[a];
      if (M.act_ctrllimited[a]) c = std::min(std::max(c, M.act_ctrlrange[2 * a]), M.act_ctrlrange[2 * a + 1]);
      qfrc_actuator[M.jnt_dofadr[M.act_jntid[a]]] += M.act_gear[a] * c;
    }
  }

  // ------------------------------------------------------------------ collision  [3P, SURVEY C.3]
  static void make_frame(V3 n, V3* fr) { UR5O_STRICT;
    fr[0] = n;
    V3 y = std::fabs(n.y) < 0.5 ? V3(0, 1, 0) : V3(0, 0, 1);
    y = y - n * dot(n, y);
    y = normalized(y);
    fr[1] = y;
    fr[2] = cross(n, y);
  }
  Shape make_shape(int g, double margin) const { UR5O_STRICT;
    Shape s;
    s.type = M.geom_type[g];
    s.pos = gxpos[g];
    s.mat = gxmat[g];
    s.size = v3(M.geom_size + 3 * g);
    s.verts = nullptr; s.nvert = 0;
    s.center = s.pos;
    if (s.type == GEOM_MESH) {
      int k = M.geom_meshid[g];
      s.verts = M.mesh_vert + 3 * M.mesh_vertadr[k];
      s.nvert = M.mesh_vertnum[k];
      s.center = s.pos + mul(s.mat, M.mesh_center[k]);
    }
    s.margin = margin;
    return s;
  }
  void add_contact(int g1, int g2, double dist, V3 pos, V3 n, double margin) { UR5O_STRICT;
    Contact c;
    c.dist = dist; c.pos = pos;
    make_frame(n, c.frame);
    c.geom1 = g1; c.geom2 = g2;
    c.dim = std::max(M.geom_condim[g1], M.geom_condim[g2]);
    c.includemargin = margin;
    const double *f1 = M.geom_friction + 3 * g1, *f2 = M.geom_friction + 3 * g2;
    double f[3] = {std::max(f1[0], f2[0]), std::max(f1[1], f2[1]), std::max(f1[2], f2[2])};
    c.friction[0] = c.friction[1] = f[0]; c.friction[2] = f[1]; c.friction[3] = c.friction[4] = f[2];
    // solmix: equal weights; every geom of the supported scenes shares one solref/solimp, so this is the identity
    for (int i = 0; i < 2; i++) c.solref[i] = 0.5 * (M.geom_solref[2 * g1 + i] + M.geom_solref[2 * g2 + i]);
    for (int i = 0; i < 5; i++) c.solimp[i] = 0.5 * (M.geom_solimp[5 * g1 + i] + M.geom_solimp[5 * g2 + i]);
    c.mu = 0; c.efc_address = -1; c.color = 0;
    contacts.push_back(c);
  }

  void collide_plane_sphere(int g1, int g2, double margin) { UR5O_STRICT;
    V3 n = gxmat[g1].col(2);
    double r = M.geom_size[3 * g2];
    double d = dot(gxpos[g2] - gxpos[g1], n) - r;
    if (d < margin) add_contact(g1, g2, d, gxpos[g2] - n * (r + 0.5 * d), n, margin);
  }
  void collide_plane_box(int g1, int g2, double margin) { UR5O_STRICT;
    V3 n = gxmat[g1].col(2);
    V3 s = v3(M.geom_size + 3 * g2);
    int cnt = 0;
    for (int k = 0; k < 8 && cnt < 4; k++) {
      V3 l((k & 1) ? s.x : -s.x, (k & 2) ? s.y : -s.y, (k & 4) ? s.z : -s.z);
      V3 v = gxpos[g2] + mul(gxmat[g2], l);
      double d = dot(v - gxpos[g1], n);
      if (d < margin) { add_contact(g1, g2, d, v - n * (0.5 * d), n, margin); cnt++; }
    }
  }
  void collide_plane_convex(int g1, int g2, double margin) { UR5O_STRICT;
    V3 n = gxmat[g1].col(2);
    Shape s = make_shape(g2, 0.0);
    V3 v = support(s, -n);
    double d = dot(v - gxpos[g1], n);
    if (d < margin) add_contact(g1, g2, d, v - n * (0.5 * d), n, margin);
  }
  void collide_sphere_sphere(int g1, int g2, double margin) { UR5O_STRICT;
    V3 d = gxpos[g2] - gxpos[g1];
    double len = norm(d), r1 = M.geom_size[3 * g1], r2 = M.geom_size[3 * g2];
    double dist = len - r1 - r2;
    if (dist >= margin) return;
    V3 n = len > 1e-12 ? d * (1.0 / len) : V3(1, 0, 0);
    add_contact(g1, g2, dist, gxpos[g1] + n * (r1 + 0.5 * dist), n, margin);
  }
  void collide_sphere_box(int g1, int g2, double margin) { UR5O_STRICT;
    double r = M.geom_size[3 * g1];
    V3 s = v3(M.geom_size + 3 * g2);
    V3 cl = mulT(gxmat[g2], gxpos[g1] - gxpos[g2]);
    V3 p(std::min(std::max(cl.x, -s.x), s.x), std::min(std::max(cl.y, -s.y), s.y), std::min(std::max(cl.z, -s.z), s.z));
    V3 d = p - cl;
    double len = norm(d);
    if (len > 1e-12) {
      double dist = len - r;
      if (dist >= margin) return;
      V3 n = mul(gxmat[g2], d * (1.0 / len));
      add_contact(g1, g2, dist, gxpos[g1] + n * (r + 0.5 * dist), n, margin);
    } else {
      int ax = 0;
      double best = 1e300;
      for (int i = 0; i < 3; i++) { double g = s[i] - std::fabs(cl[i]); if (g < best) { best = g; ax = i; } }
      V3 el; el[ax] = cl[ax] >= 0 ? 1.0 : -1.0;
      V3 e = mul(gxmat[g2], el);
      double dist = -best - r;
      add_contact(g1, g2, dist, gxpos[g1] + e * (0.5 * (best - r)), -e, margin);
    }
  }

  // ---- capsules: a segment of half length size[1] along local z with radius size[0]. MuJoCo treats plane / sphere / capsule /
  // box partners analytically [3P: mjc_PlaneCapsule, mjc_SphereCapsule, mjc_CapsuleCapsule, mjc_CapsuleBox]; only their results'
  // shape is documented (two contacts for a capsule lying on a plane or a box face, one otherwise), so the routines below are own
  // restatements built from sphere tests at points of the segment. Other partners (cylinder, mesh) go through MPR as in MuJoCo.
  void sphere_at_vs_sphere_at(int g1, int g2, V3 p1, double r1, V3 p2, double r2, double margin) { UR5O_STRICT;
    V3 d = p2 - p1;
    double len = norm(d), dist = len - r1 - r2;
    if (dist >= margin) return;
    V3 n = len > 1e-12 ? d * (1.0 / len) : V3(1, 0, 0);
    add_contact(g1, g2, dist, p1 + n * (r1 + 0.5 * dist), n, margin);
  }
  // sphere (centre c, radius r, geom g1) against box g2; returns the signed distance (1e300 when the pair was not evaluated)
  double sphere_at_vs_box(int g1, int g2, V3 c, double r, double margin, bool emit) { UR5O_STRICT;
    V3 s = v3(M.geom_size + 3 * g2);
    V3 cl = mulT(gxmat[g2], c - gxpos[g2]);
    V3 p(std::min(std::max(cl.x, -s.x), s.x), std::min(std::max(cl.y, -s.y), s.y), std::min(std::max(cl.z, -s.z), s.z));
    V3 d = p - cl;
    double len = norm(d);
    if (len > 1e-12) {
      double dist = len - r;
      if (emit && dist < margin) { V3 n = mul(gxmat[g2], d * (1.0 / len)); add_contact(g1, g2, dist, c + n * (r + 0.5 * dist), n, margin); }
      return dist;
    }
    int ax = 0;
    double best = 1e300;
    for (int i = 0; i < 3; i++) { double g = s[i] - std::fabs(cl[i]); if (g < best) { best = g; ax = i; } }
    V3 el; el[ax] = cl[ax] >= 0 ? 1.0 : -1.0;
    V3 e = mul(gxmat[g2], el);
    if (emit) add_contact(g1, g2, -best - r, c + e * (0.5 * (best - r)), -e, margin);
    return -best - r;
  }
  void collide_plane_capsule(int g1, int g2, double margin) { UR5O_STRICT;   // both end spheres
    V3 n = gxmat[g1].col(2), ax = gxmat[g2].col(2);
    double r = M.geom_size[3 * g2], h = M.geom_size[3 * g2 + 1];
    for (int e = 0; e < 2; e++) {
      V3 c = gxpos[g2] + ax * (e == 0 ? h : -h);
      double d = dot(c - gxpos[g1], n) - r;
      if (d < margin) add_contact(g1, g2, d, c - n * (r + 0.5 * d), n, margin);
    }
  }
  void collide_sphere_capsule(int g1, int g2, double margin) { UR5O_STRICT;  // sphere against the nearest point of the segment
    V3 ax = gxmat[g2].col(2);
    double h = M.geom_size[3 * g2 + 1];
    double t = std::min(std::max(dot(gxpos[g1] - gxpos[g2], ax), -h), h);
    sphere_at_vs_sphere_at(g1, g2, gxpos[g1], M.geom_size[3 * g1], gxpos[g2] + ax * t, M.geom_size[3 * g2], margin);
  }
  void collide_capsule_capsule(int g1, int g2, double margin) { UR5O_STRICT;
    V3 a1 = gxmat[g1].col(2), a2 = gxmat[g2].col(2), w = gxpos[g1] - gxpos[g2];
    double h1 = M.geom_size[3 * g1 + 1], h2 = M.geom_size[3 * g2 + 1], r1 = M.geom_size[3 * g1], r2 = M.geom_size[3 * g2];
    double b = dot(a1, a2), d = dot(a1, w), e = dot(a2, w), den = 1.0 - b * b;
    if (den < 1e-6) {   // parallel axes: the overlap of the two segments, one contact at each of its ends (or one if it is a point)
      // parameters on segment 1 of the projections of segment 2's ends
      double sgn = b >= 0 ? 1.0 : -1.0;
      double c2 = -d;                          // projection of centre 2 on axis 1, relative to centre 1
      double t_lo = std::max(-h1, c2 - h2), t_hi = std::min(h1, c2 + h2);
      if (t_lo > t_hi) { double tm = std::min(std::max(c2, -h1), h1); t_lo = t_hi = tm; }
      for (int k = 0; k < (t_hi - t_lo > 1e-9 ? 2 : 1); k++) {
        double t1 = k == 0 ? t_lo : t_hi;
        double t2 = std::min(std::max(sgn * (t1 - c2), -h2), h2);
        sphere_at_vs_sphere_at(g1, g2, gxpos[g1] + a1 * t1, r1, gxpos[g2] + a2 * t2, r2, margin);
      }
      return;
    }
    double t1 = std::min(std::max((b * e - d) / den, -h1), h1);
    double t2 = std::min(std::max(e + b * t1, -h2), h2);
    t1 = std::min(std::max(b * t2 - d, -h1), h1);           // re-project after clamping
    sphere_at_vs_sphere_at(g1, g2, gxpos[g1] + a1 * t1, r1, gxpos[g2] + a2 * t2, r2, margin);
  }
  void collide_capsule_box(int g1, int g2, double margin) { UR5O_STRICT;
    V3 ax = gxmat[g1].col(2);
    double r = M.geom_size[3 * g1], h = M.geom_size[3 * g1 + 1];
    double d_hi = sphere_at_vs_box(g1, g2, gxpos[g1] + ax * h, r, margin, false), d_lo = sphere_at_vs_box(g1, g2, gxpos[g1] - ax * h, r, margin, false);
    if (d_hi < margin && d_lo < margin) {   // lying against a face: the two end spheres
      sphere_at_vs_box(g1, g2, gxpos[g1] + ax * h, r, margin, true);
      sphere_at_vs_box(g1, g2, gxpos[g1] - ax * h, r, margin, true);
      return;
    }
    // otherwise the point of the segment nearest to the box: the distance is convex along the segment (golden-section search)
    const double gr = 0.6180339887498949;
    double lo = -h, hi = h, x1 = hi - gr * (hi - lo), x2 = lo + gr * (hi - lo);
    double f1 = sphere_at_vs_box(g1, g2, gxpos[g1] + ax * x1, r, margin, false), f2 = sphere_at_vs_box(g1, g2, gxpos[g1] + ax * x2, r, margin, false);
    for (int it = 0; it < 40; it++) {
      if (f1 < f2) { hi = x2; x2 = x1; f2 = f1; x1 = hi - gr * (hi - lo); f1 = sphere_at_vs_box(g1, g2, gxpos[g1] + ax * x1, r, margin, false); }
      else { lo = x1; x1 = x2; f1 = f2; x2 = lo + gr * (hi - lo); f2 = sphere_at_vs_box(g1, g2, gxpos[g1] + ax * x2, r, margin, false); }
    }
    double t = 0.5 * (lo + hi);
    if (d_hi <= std::min(f1, f2)) t = h; else if (d_lo <= std::min(f1, f2)) t = -h;   // an end sphere is at least as close
    sphere_at_vs_box(g1, g2, gxpos[g1] + ax * t, r, margin, true);
  }

  // box-box: separating-axis test + reference-face clipping (own algorithm; MuJoCo's mjc_BoxBox [3P] likewise
  // returns up to 8 points for face contacts and 1 for edge-edge)
  void collide_box_box(int g1, int g2, double margin) { UR5O_STRICT;
    V3 pa = gxpos[g1], pb = gxpos[g2], a = v3(M.geom_size + 3 * g1), b = v3(M.geom_size + 3 * g2);
    const M3 &Ra = gxmat[g1], &Rb = gxmat[g2];
    V3 t = pb - pa;
    double R[3][3], Q[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = dot(Ra.col(i), Rb.col(j)); Q[i][j] = std::fabs(R[i][j]); }
    V3 ta(dot(t, Ra.col(0)), dot(t, Ra.col(1)), dot(t, Ra.col(2)));
    double best = -1e300; int code = -1; V3 bestn; bool flip = false;
    // face axes of A
    for (int i = 0; i < 3; i++) {
      double s = std::fabs(ta[i]) - (a[i] + b.x * Q[i][0] + b.y * Q[i][1] + b.z * Q[i][2]);
      if (s > margin) return;
      if (s > best) { best = s; code = i; bestn = Ra.col(i); flip = ta[i] < 0; }
    }
    // face axes of B
    for (int j = 0; j < 3; j++) {
      double tb = dot(t, Rb.col(j));
      double s = std::fabs(tb) - (b[j] + a.x * Q[0][j] + a.y * Q[1][j] + a.z * Q[2][j]);
      if (s > margin) return;
      if (s > best) { best = s; code = 3 + j; bestn = Rb.col(j); flip = tb < 0; }
    }
    // edge x edge axes; must beat the best face by 5 % (and 1e-6 absolute) to be chosen
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
      V3 Lx = cross(Ra.col(i), Rb.col(j));
      double l = norm(Lx);
      if (l < 1e-6) continue;
      Lx = Lx * (1.0 / l);
      double tl = dot(t, Lx);
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += a[k] * std::fabs(dot(Ra.col(k), Lx)); rb += b[k] * std::fabs(dot(Rb.col(k), Lx)); }
      double s = std::fabs(tl) - (ra + rb);
      if (s > margin) return;
      if (s > best + 1e-6 + 0.05 * std::fabs(best)) { best = s; code = 6 + 3 * i + j; bestn = Lx; flip = tl < 0; }
    }
    V3 n = flip ? -bestn : bestn;  // from A to B
    if (code >= 6) {
      int i = (code - 6) / 3, j = (code - 6) % 3;
      V3 ea = pa, eb = pb;
      for (int k = 0; k < 3; k++) if (k != i) ea = ea + Ra.col(k) * ((dot(n, Ra.col(k)) > 0 ? 1.0 : -1.0) * a[k]);
      for (int k = 0; k < 3; k++) if (k != j) eb = eb - Rb.col(k) * ((dot(n, Rb.col(k)) > 0 ? 1.0 : -1.0) * b[k]);
      V3 ua = Ra.col(i), ub = Rb.col(j), w = ea - eb;
      double uaub = dot(ua, ub), q1 = dot(ua, w), q2 = dot(ub, w), den = 1.0 - uaub * uaub;
      double sa = 0, sb = 0;
      if (den > 1e-12) { sa = (uaub * q2 - q1) / den; sb = (q2 - uaub * q1) / den; }
      sa = std::min(std::max(sa, -a[i]), a[i]);
      sb = std::min(std::max(sb, -b[j]), b[j]);
      V3 ca = ea + ua * sa, cb = eb + ub * sb;
      add_contact(g1, g2, best, (ca + cb) * 0.5, n, margin);
      return;
    }
    // face contact: reference box owns the axis
    bool refA = code < 3;
    int ax = refA ? code : code - 3;
    V3 pr = refA ? pa : pb, pi = refA ? pb : pa, r = refA ? a : b, in = refA ? b : a;
    const M3 &Rr = refA ? Ra : Rb, &Ri = refA ? Rb : Ra;
    V3 nref = refA ? n : -n;  // outward normal of the reference face, towards the incident box
    // incident face: most anti-parallel to nref
    int iax = 0; double bd = -1;
    for (int k = 0; k < 3; k++) { double d = std::fabs(dot(Ri.col(k), nref)); if (d > bd) { bd = d; iax = k; } }
    double isgn = dot(Ri.col(iax), nref) > 0 ? -1.0 : 1.0;
    V3 ic = pi + Ri.col(iax) * (isgn * in[iax]);
    int u = (iax + 1) % 3, v = (iax + 2) % 3;
    V3 poly[16], tmp[16];
    int np = 4;
    poly[0] = ic + Ri.col(u) * in[u] + Ri.col(v) * in[v];
    poly[1] = ic - Ri.col(u) * in[u] + Ri.col(v) * in[v];
    poly[2] = ic - Ri.col(u) * in[u] - Ri.col(v) * in[v];
    poly[3] = ic + Ri.col(u) * in[u] - Ri.col(v) * in[v];
    int ru = (ax + 1) % 3, rv = (ax + 2) % 3;
    double rsgn = dot(Rr.col(ax), nref) > 0 ? 1.0 : -1.0;
    V3 rc = pr + Rr.col(ax) * (rsgn * r[ax]);
    // clip against the four side planes of the reference face
    for (int side = 0; side < 4 && np > 0; side++) {
      V3 pn = (side < 2 ? Rr.col(ru) : Rr.col(rv)) * ((side & 1) ? -1.0 : 1.0);
      double lim = side < 2 ? r[ru] : r[rv];
      int nn = 0;
      for (int k = 0; k < np; k++) {
        V3 p0 = poly[k], p1 = poly[(k + 1) % np];
        double d0 = dot(p0 - rc, pn) - lim, d1 = dot(p1 - rc, pn) - lim;
        if (std::fabs(d0) <= 1e-9) d0 = 0;  // a vertex ON the clip line stays, and an edge ALONG it yields no crossing: without this a
        if (std::fabs(d1) <= 1e-9) d1 = 0;  // flush stack of equal boxes (the model's own qpos0) gets vertices where the rounding falls
        if (d0 <= 0) tmp[nn++] = p0;
        if ((d0 < 0 && d1 > 0) || (d0 > 0 && d1 < 0)) tmp[nn++] = p0 + (p1 - p0) * (d0 / (d0 - d1));
      }
      np = std::min(nn, 8);
      for (int k = 0; k < np; k++) poly[k] = tmp[k];
    }
    for (int k = 0; k < np; k++) {
      double d = dot(poly[k] - rc, nref);
      if (d < margin) add_contact(g1, g2, d, poly[k] - nref * (0.5 * d), n, margin);
    }
  }
  void collide_convex(int g1, int g2, double margin) { UR5O_STRICT;
    Shape A = make_shape(g1, margin), B = make_shape(g2, margin);
    double depth; V3 dir, pos;
    if (!mpr_penetration(A, B, &depth, &dir, &pos)) return;
    double dist = margin - depth;
    if (dist < margin) add_contact(g1, g2, dist, pos, dir, margin);
  }

  void collision() { UR5O_STRICT;
    contacts.clear();
    if (!contacts_enabled) return;
    for (int p = 0; p < M.npair; p++) {
      int g1 = M.pair_geom1[p], g2 = M.pair_geom2[p];
      if (M.geom_type[g1] > M.geom_type[g2]) std::swap(g1, g2);
      double margin = std::max(M.geom_margin[g1], M.geom_margin[g2]);
      int t1 = M.geom_type[g1], t2 = M.geom_type[g2];
      if (t1 != GEOM_PLANE) {  // bounding-sphere cull
        V3 c1 = gxpos[g1], c2 = gxpos[g2];
        double rr = M.geom_rbound[g1] + M.geom_rbound[g2] + margin;
        V3 d = c2 - c1;
        if (dot(d, d) > rr * rr) continue;
      } else {
        V3 n = gxmat[g1].col(2);
        if (dot(gxpos[g2] - gxpos[g1], n) > M.geom_rbound[g2] + margin) continue;
      }
      if (t1 == GEOM_PLANE) {
        if (t2 == GEOM_SPHERE) collide_plane_sphere(g1, g2, margin);
        else if (t2 == GEOM_BOX) collide_plane_box(g1, g2, margin);
        else if (t2 == GEOM_CAPSULE) collide_plane_capsule(g1, g2, margin);
        else collide_plane_convex(g1, g2, margin);
      } else if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE) collide_sphere_sphere(g1, g2, margin);
      else if (t1 == GEOM_SPHERE && t2 == GEOM_CAPSULE) collide_sphere_capsule(g1, g2, margin);
      else if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE) collide_capsule_capsule(g1, g2, margin);
      else if (t1 == GEOM_CAPSULE && t2 == GEOM_BOX) collide_capsule_box(g1, g2, margin);
      else if (t1 == GEOM_SPHERE && t2 == GEOM_BOX) collide_sphere_box(g1, g2, margin);
      else if (t1 == GEOM_BOX && t2 == GEOM_BOX) collide_box_box(g1, g2, margin);
      else collide_convex(g1, g2, margin);
    }
    // test hook (ur5o_set_contact_order): the same contact SET in reversed order. Mathematically the same step; every sum over contacts associates
    // differently, i.e. a last-bit perturbation -- what tools/pile_chaos_floor.py uses to measure how far a pile attempt amplifies rounding.
    if (contact_order == 1) std::reverse(contacts.begin(), contacts.end());
  }
  int contact_order = 0;

  // ------------------------------------------------------------------ constraint rows  [3P, SURVEY C.4]
  static double powr(double x, double p) { return p == 2.0 ? x * x : std::pow(x, p); }   // p = 2 is MuJoCo's default
  static double impedance(const double* solimp, double x_abs) {
    double dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
    dmin = std::min(std::max(dmin, 0.0001), 0.9999);
    dmax = std::min(std::max(dmax, 0.0001), 0.9999);
    if (dmin == dmax || width <= MINVAL) return 0.5 * (dmin + dmax);
    double x = x_abs / width;
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    double y;
    if (power == 1) y = x;
    else if (x <= mid) y = powr(x / mid, power) * mid;  // a x^p with a = 1/mid^(p-1)
    else y = 1 - powr((1 - x) / (1 - mid), power) * (1 - mid);
    return dmin + y * (dmax - dmin);
  }
  void jac_point(int body, V3 p, std::vector<double>& jp, std::vector<double>& jr) const {
    // translational / rotational Jacobian (3 x nv each) of point p attached to body
    std::fill(jp.begin(), jp.end(), 0.0);
    std::fill(jr.begin(), jr.end(), 0.0);
    if (body <= 0) return;
    V3 off = p - subtree_com[M.body_rootid[body]];
    int b = body;
    while (b > 0 && M.body_dofnum[b] == 0) b = M.body_parentid[b];
    if (b <= 0) return;
    for (int d = M.body_dofadr[b] + M.body_dofnum[b] - 1; d >= 0; d = M.dof_parentid[d]) {
      V3 lin = cdof[d].l + cross(cdof[d].r, off);
      for (int k = 0; k < 3; k++) { jp[(size_t)k * nv + d] = lin[k]; jr[(size_t)k * nv + d] = cdof[d].r[k]; }
    }
  }
  void finish_row(Row& r, const std::vector<double>& Jdense, const double* solref, const double* solimp) {
    // sparsify over touched trees, compute B = M^-1 J^T, Adiag, vel, impedance -> R, aref
    std::vector<char> touched(M.ntree, 0);
    for (int d = 0; d < nv; d++) if (Jdense[d] != 0.0) touched[M.dof_treeid[d]] = 1;
    std::vector<double> x(nv, 0.0);
    r.idx.clear(); r.J.clear(); r.B.clear();
    for (int t = 0; t < M.ntree; t++) if (touched[t]) {
      int lo = M.tree_dofadr[t], hi = lo + M.tree_dofnum[t];
      for (int d = lo; d < hi; d++) x[d] = Jdense[d];
      if (solver == 1) solve(L, x.data(), lo, hi);
      for (int d = lo; d < hi; d++) { r.idx.push_back(d); r.J.push_back(Jdense[d]); r.B.push_back(x[d]); }
    }
    r.Adiag = 0; r.vel = 0;
    for (size_t k = 0; k < r.idx.size(); k++) { r.Adiag += r.J[k] * r.B[k]; r.vel += r.J[k] * qvel[r.idx[k]]; }
    double imp = impedance(solimp, std::fabs(r.pos - r.margin));
    r.R = std::max(MINVAL, (1 - imp) * r.diagApprox / imp);
    double tc = std::max(solref[0], 2 * M.timestep), dr = solref[1], dmax = std::min(std::max(solimp[1], 0.0001), 0.9999);
    double k = 1.0 / (dmax * dmax * tc * tc * dr * dr), bb = 2.0 / (dmax * tc);
    r.aref = -bb * r.vel - k * imp * (r.pos - r.margin);
    r.force = 0;
  }
  void make_constraints() {
    rows.clear();
    std::vector<double> Jd(nv);
    // joint equalities (UR5gripper_2_finger.xml:333)
    for (int e = 0; e < M.neq; e++) {
      int j1 = M.eq_jnt1[e], j2 = M.eq_jnt2[e];
      int q1 = M.jnt_qposadr[j1], q2 = M.jnt_qposadr[j2], d1 = M.jnt_dofadr[j1], d2 = M.jnt_dofadr[j2];
      const double* pc = M.eq_polycoef + 5 * e;
      double x = qpos[q2] - M.qpos0[q2];
      double poly = pc[0] + x * (pc[1] + x * (pc[2] + x * (pc[3] + x * pc[4])));
      double dpoly = pc[1] + x * (2 * pc[2] + x * (3 * pc[3] + x * 4 * pc[4]));
      Row r;
      r.pos = (qpos[q1] - M.qpos0[q1]) - poly; r.margin = 0; r.unilateral = false; r.contact = -1;
      std::fill(Jd.begin(), Jd.end(), 0.0);
      Jd[d1] = 1.0; Jd[d2] = -dpoly;
      r.diagApprox = M.dof_invweight0[d1] + M.dof_invweight0[d2];
      finish_row(r, Jd, M.eq_solref + 2 * e, M.eq_solimp + 5 * e);
      rows.push_back(r);
    }
    // joint limits
    for (int j = 0; j < M.njnt; j++) {
      if (!M.jnt_limited[j]) continue;
      int qa = M.jnt_qposadr[j], d = M.jnt_dofadr[j];
      for (int side = 0; side < 2; side++) {
        double dist = side == 0 ? qpos[qa] - M.jnt_range[2 * j] : M.jnt_range[2 * j + 1] - qpos[qa];
        if (dist >= 0) continue;
        Row r;
        r.pos = dist; r.margin = 0; r.unilateral = true; r.contact = -1;
        std::fill(Jd.begin(), Jd.end(), 0.0);
        Jd[d] = side == 0 ? 1.0 : -1.0;
        r.diagApprox = M.dof_invweight0[d];
        finish_row(r, Jd, M.jnt_solref, M.jnt_solimp);
        rows.push_back(r);
      }
    }
    // contacts: pyramidal cones, 2 (dim-1) rows each
    std::vector<double> jp1(3 * nv), jr1(3 * nv), jp2(3 * nv), jr2(3 * nv);
    for (size_t ci = 0; ci < contacts.size(); ci++) {
      Contact& c = contacts[ci];
      int b1 = M.geom_bodyid[c.geom1], b2 = M.geom_bodyid[c.geom2];
      jac_point(b1, c.pos, jp1, jr1);
      jac_point(b2, c.pos, jp2, jr2);
      // base rows in the contact frame: 0 normal, 1-2 tangents (translation), 3 torsion, 4-5 rolling (rotation)
      std::vector<std::vector<double>> base(c.dim, std::vector<double>(nv, 0.0));
      for (int k = 0; k < c.dim; k++) {
        V3 ax = c.frame[k < 3 ? k : k - 3];
        const std::vector<double>&A1 = k < 3 ? jp1 : jr1, &A2 = k < 3 ? jp2 : jr2;
        for (int d = 0; d < nv; d++)
          base[k][d] = ax.x * (A2[d] - A1[d]) + ax.y * (A2[nv + d] - A1[nv + d]) + ax.z * (A2[2 * nv + d] - A1[2 * nv + d]);
      }
      double tran = M.body_invweight0[2 * b1] + M.body_invweight0[2 * b2];
      double rot = M.body_invweight0[2 * b1 + 1] + M.body_invweight0[2 * b2 + 1];
      c.efc_address = (int)rows.size();
      if (c.dim == 1) {
        Row r; r.pos = c.dist; r.margin = c.includemargin; r.unilateral = true; r.contact = (int)ci;
        r.diagApprox = tran;
        finish_row(r, base[0], c.solref, c.solimp);
        rows.push_back(r);
        continue;
      }
      for (int k = 0; k < c.dim - 1; k++) {
        double fri = c.friction[k];
        for (int s = 0; s < 2; s++) {
          Row r; r.pos = c.dist; r.margin = c.includemargin; r.unilateral = true; r.contact = (int)ci;
          for (int d = 0; d < nv; d++) Jd[d] = base[0][d] + (s == 0 ? fri : -fri) * base[k + 1][d];
          r.diagApprox = tran + fri * fri * (k < 2 ? tran : rot);
          finish_row(r, Jd, c.solref, c.solimp);
          rows.push_back(r);
        }
      }
      // pyramidal: all rows share Rpy = 2 mu^2 R[first], mu = friction[0] / sqrt(impratio)
      c.mu = c.friction[0] * std::sqrt(1.0 / std::max(MINVAL, M.impratio));
      double Rpy = 2 * c.mu * c.mu * rows[c.efc_address].R;
      for (int k = 0; k < 2 * (c.dim - 1); k++) rows[c.efc_address + k].R = Rpy;
    }
    for (auto& r : rows) r.D = 1.0 / r.R;
    // PGS sweep order: equality rows, limit rows, then contacts in greedy-colour order. Contacts of one colour
    // share no movable tree, so their updates commute; the HIP engine runs one colour's contacts in parallel lanes.
    order.clear();
    size_t nfirst = rows.size();
    for (size_t ci = 0; ci < contacts.size(); ci++) nfirst = std::min(nfirst, (size_t)contacts[ci].efc_address);
    for (size_t i = 0; i < nfirst; i++) order.push_back((int)i);
    std::vector<std::vector<int>> used;  // used[color] = trees
    int ncolor = 0;
    for (size_t ci = 0; ci < contacts.size(); ci++) {
      int t1 = M.body_treeid[M.geom_bodyid[contacts[ci].geom1]], t2 = M.body_treeid[M.geom_bodyid[contacts[ci].geom2]];
      int col = 0;
      for (;; col++) {
        if (col >= (int)used.size()) { used.push_back({}); }
        bool clash = false;
        for (int t : used[col]) if ((t1 >= 0 && t == t1) || (t2 >= 0 && t == t2)) clash = true;
        if (!clash) break;
      }
      if (t1 >= 0) used[col].push_back(t1);
      if (t2 >= 0) used[col].push_back(t2);
      contacts[ci].color = col;
      ncolor = std::max(ncolor, col + 1);
    }
    for (int col = 0; col < ncolor; col++)
      for (size_t ci = 0; ci < contacts.size(); ci++)
        if (contacts[ci].color == col) {
          int nr = contacts[ci].dim == 1 ? 1 : 2 * (contacts[ci].dim - 1);
          for (int k = 0; k < nr; k++) order.push_back(contacts[ci].efc_address + k);
        }
  }

  // ------------------------------------------------------------------ PGS on the dual, velocity-space form  [3P, SURVEY C.4]
  double row_jar(const Row& r, const std::vector<double>& a) const {
    double s = 0;
    for (size_t k = 0; k < r.idx.size(); k++) s += r.J[k] * a[r.idx[k]];
    return s - r.aref;
  }
  // ------------------------------------------------------------------ Newton on the primal (MuJoCo's default solver) [3P]
  // min_x 1/2 (x - a_s)^T M (x - a_s) + sum_i s_i(J_i x - aref_i),  s_i(r) = 1/2 D_i r^2 (equality) or 1/2 D_i min(r,0)^2
  // (limits, pyramidal contact rows). H = M + J_act^T D J_act is re-assembled and Cholesky-factored every iteration; the
  // line search is exact (safeguarded Newton on the piecewise-linear derivative). Warm start: better of qacc_warmstart / a_s.
  std::vector<double> nH, nMa, ngrad, nsearch, nMv, njar, njv;
  double primal_cost(const std::vector<double>& x, const std::vector<double>& Ma) const {
    double c = 0;
    for (int i = 0; i < nv; i++) c += 0.5 * (Ma[i] - qfrc_smooth[i]) * (x[i] - qacc_smooth[i]);
    for (const Row& r : rows) {
      double jar = row_jar(r, x);
      if (!r.unilateral || jar < 0) c += 0.5 * r.D * jar * jar;
    }
    return c;
  }
  void matvecM(const std::vector<double>& x, std::vector<double>& out) const {
    for (int i = 0; i < nv; i++) {
      double s2 = 0;
      for (int j = 0; j < nv; j++) s2 += Mm[(size_t)i * nv + j] * x[j];
      out[i] = s2;
    }
  }
  // test hook (ur5o_set_cholesky_order): 2 = pivots by reciprocal square root (below); 1 = eliminate the dofs in REVERSED order. Mathematically the same solve; the rounding differs the way it differs between two
  // implementations that factor in different orders (the HIP pile kernel eliminates by island and x position, this oracle by dof number) -- the "different text" twin of
  // tools/pile_divergence_time.py, next to the summation-order and 1-ulp twins.
  int cholesky_order = 0;
  bool dense_cholesky_solve(std::vector<double>& A, std::vector<double>& b) const {
    if (cholesky_order == 0 || cholesky_order == 2) return dense_cholesky_solve_natural(A, b);
    std::vector<double> Ap((size_t)nv * nv), bp(nv);
    for (int i = 0; i < nv; i++) { bp[i] = b[nv - 1 - i]; for (int j = 0; j < nv; j++) Ap[(size_t)i * nv + j] = A[(size_t)(nv - 1 - i) * nv + (nv - 1 - j)]; }
    const bool ok = dense_cholesky_solve_natural(Ap, bp);
    for (int i = 0; i < nv; i++) b[nv - 1 - i] = bp[i];
    return ok;
  }
  bool dense_cholesky_solve_natural(std::vector<double>& A, std::vector<double>& b) const {  // A = L L^T in place (lower), b <- A^-1 b
    for (int j = 0; j < nv; j++) {
      double d = A[(size_t)j * nv + j];
      for (int k = 0; k < j; k++) d -= A[(size_t)j * nv + k] * A[(size_t)j * nv + k];
      if (d < MINVAL) d = MINVAL;
      if (cholesky_order == 2) {   // test hook: pivots through 1 / sqrt and multiplications, the way the HIP kernels factor (rsqrt + one Newton step): a "second text" twin
        const double inv = 1.0 / std::sqrt(d);
        A[(size_t)j * nv + j] = d * inv;
        for (int i = j + 1; i < nv; i++) {
          double v = A[(size_t)i * nv + j];
          for (int k = 0; k < j; k++) v -= A[(size_t)i * nv + k] * A[(size_t)j * nv + k];
          A[(size_t)i * nv + j] = v * inv;
        }
        continue;
      }
      d = std::sqrt(d);
      A[(size_t)j * nv + j] = d;
      for (int i = j + 1; i < nv; i++) {
        double v = A[(size_t)i * nv + j];
        for (int k = 0; k < j; k++) v -= A[(size_t)i * nv + k] * A[(size_t)j * nv + k];
        A[(size_t)i * nv + j] = v / d;
      }
    }
    for (int i = 0; i < nv; i++) {
      double v = b[i];
      for (int k = 0; k < i; k++) v -= A[(size_t)i * nv + k] * b[k];
      b[i] = v / A[(size_t)i * nv + i];
    }
    for (int i = nv - 1; i >= 0; i--) {
      double v = b[i];
      for (int k = i + 1; k < nv; k++) v -= A[(size_t)k * nv + i] * b[k];
      b[i] = v / A[(size_t)i * nv + i];
    }
    return true;
  }
  // test hook (ur5o_newton_trace): the active flag of every constraint row at each Hessian evaluation of the last solve -- what tools/newton_iteration_analysis.py
  // reads to see how much of the Hessian really changes from one Newton iteration to the next
  bool trace_newton = false;
  std::vector<std::vector<unsigned char>> newton_trace;
  void newton_direction() {  // nsearch = -H^-1 grad at the current qacc (uses njar)
    if (trace_newton) {
      std::vector<unsigned char> a(rows.size());
      for (size_t ri = 0; ri < rows.size(); ri++) a[ri] = !(rows[ri].unilateral && njar[ri] >= 0);
      newton_trace.push_back(a);
    }
    nH = Mm;
    for (size_t ri = 0; ri < rows.size(); ri++) {
      const Row& r = rows[ri];
      if (r.unilateral && njar[ri] >= 0) continue;
      for (size_t a = 0; a < r.idx.size(); a++) {
        if (r.J[a] == 0.0) continue;
        double da = r.D * r.J[a];
        for (size_t b = 0; b < r.idx.size(); b++) nH[(size_t)r.idx[a] * nv + r.idx[b]] += da * r.J[b];
      }
    }
    for (int i = 0; i < nv; i++) ngrad[i] = nMa[i] - qfrc_smooth[i];
    for (size_t ri = 0; ri < rows.size(); ri++) {
      const Row& r = rows[ri];
      if (r.unilateral && njar[ri] >= 0) continue;
      double f = -r.D * njar[ri];
      for (size_t a = 0; a < r.idx.size(); a++) ngrad[r.idx[a]] -= r.J[a] * f;
    }
    nsearch = ngrad;
    dense_cholesky_solve(nH, nsearch);
    for (int i = 0; i < nv; i++) nsearch[i] = -nsearch[i];
  }
  void solve_newton() {
    size_t ne = rows.size();
    nH.resize((size_t)nv * nv); nMa.assign(nv, 0); ngrad.assign(nv, 0); nsearch.assign(nv, 0); nMv.assign(nv, 0);
    njar.assign(ne, 0); njv.assign(ne, 0);
    // warm start
    std::vector<double> Maw(nv), Mas(nv);
    matvecM(qacc_warmstart, Maw);
    matvecM(qacc_smooth, Mas);
    double cw = primal_cost(qacc_warmstart, Maw), cs = primal_cost(qacc_smooth, Mas);
    if (cw < cs) { qacc = qacc_warmstart; nMa = Maw; } else { qacc = qacc_smooth; nMa = Mas; }
    for (size_t i = 0; i < ne; i++) njar[i] = row_jar(rows[i], qacc);
    double cost = std::min(cw, cs);
    double scale = 1.0 / (M.meaninertia * std::max(1, nv));
    newton_trace.clear();
    newton_direction();
    for (int it = 0; it < solver_iterations(); it++) {
      // exact line search on phi(alpha) = cost(qacc + alpha * search)
      matvecM(nsearch, nMv);
      for (size_t i = 0; i < ne; i++) {
        const Row& r = rows[i];
        double s2 = 0;
        for (size_t k = 0; k < r.idx.size(); k++) s2 += r.J[k] * nsearch[r.idx[k]];
        njv[i] = s2;
      }
      double q1 = 0, q2 = 0;  // Gauss part: phi_g'(a) = q1 + a q2
      for (int i = 0; i < nv; i++) { q1 += nsearch[i] * (nMa[i] - qfrc_smooth[i]); q2 += nsearch[i] * nMv[i]; }
      auto deriv = [&](double a, double* d2) {
        double d1 = q1 + a * q2, dd = q2;
        for (size_t i = 0; i < ne; i++) {
          const Row& r = rows[i];
          double x = njar[i] + a * njv[i];
          if (!r.unilateral || x < 0) { d1 += r.D * x * njv[i]; dd += r.D * njv[i] * njv[i]; }
        }
        *d2 = dd;
        return d1;
      };
      double snorm = 0;
      for (int i = 0; i < nv; i++) snorm += nsearch[i] * nsearch[i];
      snorm = std::sqrt(snorm);
      solver_iter_last = it + 1;
      if (snorm < MINVAL) break;
      double gtol = M.tolerance * 0.01 * snorm / scale;  // MuJoCo: tolerance * ls_tolerance * |search| * meaninertia * nv
      double lo = 0, hi = -1, a = 0, d2;
      double d1 = deriv(0.0, &d2);
      if (d1 < 0) {
        for (int ls = 0; ls < 50; ls++) {
          double an = a - d1 / d2;
          if (hi > 0 && (an <= lo || an >= hi)) an = 0.5 * (lo + hi);
          if (hi < 0 && an <= lo) an = 2 * lo + 1e-12;
          a = an;
          d1 = deriv(a, &d2);
          if (std::fabs(d1) <= gtol) break;
          if (d1 < 0) lo = a; else hi = a;
        }
      }
      if (a <= 0) break;
      for (int i = 0; i < nv; i++) { qacc[i] += a * nsearch[i]; nMa[i] += a * nMv[i]; }
      for (size_t i = 0; i < ne; i++) njar[i] += a * njv[i];
      double newcost = 0;
      for (int i = 0; i < nv; i++) newcost += 0.5 * (nMa[i] - qfrc_smooth[i]) * (qacc[i] - qacc_smooth[i]);
      for (size_t i = 0; i < ne; i++) if (!rows[i].unilateral || njar[i] < 0) newcost += 0.5 * rows[i].D * njar[i] * njar[i];
      double improvement = scale * (cost - newcost);
      cost = newcost;
      newton_direction();
      double gn = 0;
      for (int i = 0; i < nv; i++) gn += ngrad[i] * ngrad[i];
      double gradient = scale * std::sqrt(gn);
      if (improvement < solver_tolerance() || gradient < solver_tolerance()) break;
    }
    for (size_t i = 0; i < ne; i++) rows[i].force = (!rows[i].unilateral || njar[i] < 0) ? -rows[i].D * njar[i] : 0.0;
    solver_iter_total += solver_iter_last;
  }

  int solver = 0;  // 0 = Newton (reference default [3P]), 1 = PGS (kept for comparison; north_star names it)
  // test hook (ur5o_set_solver_limits): run a solver past the model's iteration cap / tolerance, e.g. PGS to convergence for the solver cross-check
  int iter_override = 0;
  double tol_override = -1;
  int solver_iterations() const { return iter_override > 0 ? iter_override : M.iterations; }
  double solver_tolerance() const { return tol_override >= 0 ? tol_override : M.tolerance; }
  void solve_constraints() {
    qacc = qacc_smooth;
    solver_iter_last = 0;
    if (rows.empty()) return;
    if (solver == 0) { solve_newton(); return; }
    // warm start: forces from the previous step's qacc (primal -> dual map), kept only if it beats f = 0
    for (auto& r : rows) {
      double jar = row_jar(r, qacc_warmstart);
      double f = -r.D * jar;
      if (r.unilateral && f < 0) f = 0;
      r.force = f;
    }
    std::vector<double> dq(nv, 0.0);
    for (auto& r : rows) for (size_t k = 0; k < r.idx.size(); k++) dq[r.idx[k]] += r.B[k] * r.force;
    double cost = 0;
    for (auto& r : rows) {
      double jd = 0, b = -r.aref;
      for (size_t k = 0; k < r.idx.size(); k++) { jd += r.J[k] * dq[r.idx[k]]; b += r.J[k] * qacc_smooth[r.idx[k]]; }
      cost += r.force * (0.5 * (jd + r.R * r.force) + b);
    }
    if (cost > 0) { for (auto& r : rows) r.force = 0; }
    else { for (int d = 0; d < nv; d++) qacc[d] += dq[d]; }
    double scale = 1.0 / (M.meaninertia * std::max(1, nv));
    for (int it = 0; it < solver_iterations(); it++) {
      double improvement = 0;
      for (int ri : order) {
        Row& r = rows[ri];
        double res = row_jar(r, qacc) + r.R * r.force;
        double den = r.Adiag + r.R;
        double fnew = r.force - res / den;
        if (r.unilateral && fnew < 0) fnew = 0;
        double delta = fnew - r.force;
        if (delta != 0.0) {
          r.force = fnew;
          for (size_t k = 0; k < r.idx.size(); k++) qacc[r.idx[k]] += r.B[k] * delta;
          improvement -= delta * (0.5 * delta * den + res);
        }
      }
      solver_iter_last = it + 1;
      if (improvement * scale < solver_tolerance()) break;
    }
    solver_iter_total += solver_iter_last;
  }

  // ------------------------------------------------------------------ mj_forward / mj_Euler / mj_step  [3P, SURVEY C.2, C.5]
  void forward_position() {
    kinematics();
    com_pos();
    crb_and_factor();
    collision();
    make_constraints();
  }
  void forward() {
    forward_position();
    velocity_stage();
    // rows carry vel/aref computed in make_constraints() from the current qvel (position + velocity stages fused)
    actuation();
    for (int i = 0; i < nv; i++) { qfrc_smooth[i] = qfrc_passive[i] - qfrc_bias[i] + qfrc_actuator[i]; qacc_smooth[i] = qfrc_smooth[i]; }
    solve(L, qacc_smooth.data(), 0, nv);
    solve_constraints();
  }
  void integrate() {
    double h = M.timestep;
    // implicit joint damping: (M + h B) qacc' = qfrc_smooth + J^T f = M qacc
    std::vector<double> rhs(nv, 0.0);
    for (int i = 0; i < nv; i++) {
      double s = 0;
      for (int j = 0; j < nv; j++) s += Mm[(size_t)i * nv + j] * qacc[j];
      rhs[i] = s;
    }
    for (int i = 0; i < nv; i++) qfrc_constraint[i] = rhs[i] - qfrc_smooth[i];
    bool damp = false;
    for (int i = 0; i < nv; i++) if (M.dof_damping[i] > 0) damp = true;
    if (damp) {
      factor(Mm, Ld, h);
      solve(Ld, rhs.data(), 0, nv);
    } else rhs = qacc;
    for (int i = 0; i < nv; i++) qvel[i] += h * rhs[i];
    for (int j = 0; j < M.njnt; j++) {
      int qa = M.jnt_qposadr[j], d = M.jnt_dofadr[j], t = M.jnt_type[j];
      if (t == JNT_SLIDE || t == JNT_HINGE) { qpos[qa] += h * qvel[d]; continue; }
      if (t == JNT_FREE) {
        for (int k = 0; k < 3; k++) qpos[qa + k] += h * qvel[d + k];
        qa += 3; d += 3;
      }
      V3 w(qvel[d], qvel[d + 1], qvel[d + 2]);
      double ang = norm(w) * h;
      Q4 q{qpos[qa], qpos[qa + 1], qpos[qa + 2], qpos[qa + 3]};
      if (ang > 0) {
        Q4 dq = qaxisangle(normalized(w), ang);
        q = qmul(q, dq);
      }
      q = qnormalize(q);
      qpos[qa] = q.w; qpos[qa + 1] = q.x; qpos[qa + 2] = q.y; qpos[qa + 3] = q.z;
    }
    time += h;
  }
  // mj_step [3P] guards the state: mj_checkPos / mj_checkVel / mj_checkAcc raise mjWARN_BADQPOS / BADQVEL / BADQACC when an entry is NaN or larger than
  // mjMAXVAL = 1e10 and call mj_resetData -- the scene silently returns to qpos0 with zero velocity, warm start, controls and time (the controller's
  // PID objects live in Python and keep their state). Here the test runs once per step, on the state the step produced (an internal blow-up is
  // caught in the step it happens in; MuJoCo tests the incoming state and the acceleration), and the scene is flagged (bad_state_resets).
  long bad_state_resets = 0;
  bool state_is_bad() const {
    for (double v : qpos) if (!(std::fabs(v) <= 1e10)) return true;
    for (double v : qvel) if (!(std::fabs(v) <= 1e10)) return true;
    return false;
  }
  void reset_data() {
    qpos.assign(M.qpos0, M.qpos0 + nq);
    std::fill(qvel.begin(), qvel.end(), 0.0);
    std::fill(qacc_warmstart.begin(), qacc_warmstart.end(), 0.0);
    std::fill(ctrl.begin(), ctrl.end(), 0.0);
    time = 0;
  }
  // bench.py's CPU legs bound their sample by wall time: past the deadline a step does nothing, so the scripts of the scenes in flight run out quickly
  // (their loops end on max_steps) and the leg returns; the steps taken before the deadline are counted
  const std::atomic<bool>* stop_flag = nullptr;
  // test hook (ur5o_set_checkpoints): qpos after the given numbers of steps from the moment the hook was armed -- the trajectory samples of tools/pile_divergence_time.py
  std::vector<int> ckpt_steps;
  std::vector<std::vector<double>> ckpt_qpos;
  long ckpt_base = 0;
  void step() {  // sim.step(), MujocoController.py:379
    if (stop_flag && stop_flag->load(std::memory_order_relaxed)) return;
    forward();
    qacc_warmstart = qacc;
    integrate();
    total_steps++;
    if (state_is_bad()) { reset_data(); bad_state_resets++; }
    if (ckpt_qpos.size() < ckpt_steps.size() && total_steps - ckpt_base == ckpt_steps[ckpt_qpos.size()]) ckpt_qpos.push_back(qpos);
  }

  // ------------------------------------------------------------------ controller layer
  // MujocoController.py:269-393. group_mask bit j = actuator j belongs to the group; target may be null.
  int plot_every = 0, plot_cap = 0, plot_n = 0, plot_stride = 0;  // plot=True of move_group_to_joint_target (:275,303-304)
  int* plot_steps = nullptr;
  double* plot_q = nullptr;
  int move_group(unsigned group_mask, const double* tgt, double tolerance, int max_steps) {
    int k = 0;
    if (tgt) for (int j = 0; j < nu; j++) if (group_mask >> j & 1) target[j] = tgt[k++];
    for (int j = 0; j < nu; j++) pid[j].setpoint = target[j];
    int steps = 1, result = -1;
    bool reached = false;
    while (!reached) {
      double maxdelta = 0;
      for (int j = 0; j < nu; j++) {
        double q = qpos[M.jnt_qposadr[M.act_jntid[j]]];
        ctrl[j] = pid[j].eval(q, pid_dt);
        if (group_mask >> j & 1) maxdelta = std::max(maxdelta, std::fabs(target[j] - q));
      }
      if (plot_every && steps % plot_every == 0 && plot_n < plot_cap) {  // fill_plot_list, :338-339,639-652
        plot_steps[plot_n] = steps;
        int c = 0;
        for (int j = 0; j < nu; j++) if (group_mask >> j & 1) plot_q[plot_n * plot_stride + c++] = qpos[M.jnt_qposadr[M.act_jntid[j]]];
        plot_n++;
      }
      if (maxdelta < tolerance) { result = RES_SUCCESS; reached = true; }  // no break: one more sim.step() follows
      if (steps > max_steps) { result = RES_MAX_STEPS; break; }
      step();
      steps++;
    }
    last_steps = steps;
    return result;
  }
  unsigned mask_all() const { return (1u << nu) - 1; }
  // MujocoController.py:621-636, made deterministic (H2): ceil(ms / 1000 / h / 10) chunks of 10 steps
  void stay(double ms) {
    int chunks = (int)std::ceil(ms / 1000.0 / M.timestep / 10.0 - 1e-9);
    for (int c = 0; c < chunks; c++) move_group(mask_all(), nullptr, 1e-7, 10);
  }
  int open_gripper(bool half) { double t = half ? 0.0 : 0.4; return move_group(1u << 6, &t, 0.05, 1000); }  // :408-421
  int close_gripper(int max_steps) { double t = -0.4; return move_group(1u << 6, &t, 0.01, max_steps); }  // :423-434
  bool grasp() { return close_gripper(300) != RES_SUCCESS; }                                             // :436-444

  // Kinematic chain of the arm for IK: world pose of ee_link for given 6 arm angles (uses the model tree directly;
  // the reference builds the same chain from ur5_gripper.urdf:61-234 through ikpy [3P])
  void arm_fk(const double* q6, int ee_body, V3* p, M3* R, V3* axes, V3* anchors) const {
    // walk from the root down to ee_body
    std::vector<int> chain;
    for (int b = ee_body; b > 0; b = M.body_parentid[b]) chain.push_back(b);
    V3 pos; Q4 quat{1, 0, 0, 0};
    int k = 0;
    for (int ci = (int)chain.size() - 1; ci >= 0; ci--) {
      int b = chain[ci];
      pos = pos + mul(qmat(quat), v3(M.body_pos + 3 * b));
      quat = qmul(quat, q4(M.body_quat + 4 * b));
      for (int j = M.body_jntadr[b]; j < M.body_jntadr[b] + M.body_jntnum[b]; j++) {
        M3 Rb = qmat(quat);
        V3 anchor = pos + mul(Rb, v3(M.jnt_pos + 3 * j));
        V3 axis = mul(Rb, v3(M.jnt_axis + 3 * j));
        if (k < 6) { axes[k] = axis; anchors[k] = anchor; }
        quat = qmul(quat, qaxisangle(v3(M.jnt_axis + 3 * j), q6[k] - M.qpos0[M.jnt_qposadr[j]]));
        pos = anchor - mul(qmat(quat), v3(M.jnt_pos + 3 * j));
        k++;
      }
    }
    *p = pos;
    *R = qmat(qnormalize(quat));
  }
  // MujocoController.py:467-517: gripper-centre target -> 5 arm joint angles with the ee x-axis pointing down.
  // ikpy's scipy optimiser [3P] is replaced by a fixed-iteration Levenberg-Marquardt from the home pose (H8);
  // the reference's 2 cm FK acceptance test is kept.
  bool ik(const double* ee_position, double* out5) const {
    int ee = -1, base = -1;
    // bodies are looked up by their role: ee_link = body carrying the 0.005 box under wrist_3; resolved by name in the
    // blob's JSON on the Python side and passed through set_ik_bodies(); defaults are found structurally here.
    ee = ik_ee_body; base = ik_base_body;
    V3 tgt = V3(ee_position[0], ee_position[1], ee_position[2]) + V3(0, -0.005, 0.16);  // :493 (world frame: base offset cancels)
    (void)base;
    double q[6] = {0, -1.57, 1.57, -1.57, -1.57, 0};
    const V3 xdown(0, 0, -1);
    double lambda = 1e-4;
    for (int it = 0; it < 60; it++) {
      V3 p; M3 R; V3 ax[6], an[6];
      arm_fk(q, ee, &p, &R, ax, an);
      V3 xe = R.col(0);
      double r[6] = {p.x - tgt.x, p.y - tgt.y, p.z - tgt.z, xe.x - xdown.x, xe.y - xdown.y, xe.z - xdown.z};
      double J[6][5];
      for (int j = 0; j < 5; j++) {
        V3 dp = cross(ax[j], p - an[j]), dx = cross(ax[j], xe);
        J[0][j] = dp.x; J[1][j] = dp.y; J[2][j] = dp.z; J[3][j] = dx.x; J[4][j] = dx.y; J[5][j] = dx.z;
      }
      double A[5][6];
      for (int i = 0; i < 5; i++) {
        for (int j = 0; j < 5; j++) { double s = 0; for (int k = 0; k < 6; k++) s += J[k][i] * J[k][j]; A[i][j] = s; }
        A[i][i] += lambda;
        double s = 0; for (int k = 0; k < 6; k++) s += J[k][i] * r[k];
        A[i][5] = -s;
      }
      for (int c = 0; c < 5; c++) {  // Gaussian elimination with partial pivoting
        int piv = c;
        for (int i = c + 1; i < 5; i++) if (std::fabs(A[i][c]) > std::fabs(A[piv][c])) piv = i;
        for (int k = 0; k < 6; k++) std::swap(A[c][k], A[piv][k]);
        for (int i = c + 1; i < 5; i++) { double f = A[i][c] / A[c][c]; for (int k = c; k < 6; k++) A[i][k] -= f * A[c][k]; }
      }
      double dq[5];
      for (int i = 4; i >= 0; i--) { double s = A[i][5]; for (int k = i + 1; k < 5; k++) s -= A[i][k] * dq[k]; dq[i] = s / A[i][i]; }
      for (int j = 0; j < 5; j++) {
        double step = std::min(std::max(dq[j], -0.5), 0.5);
        int jid = M.act_jntid[j];
        q[j] = std::min(std::max(q[j] + step, M.jnt_range[2 * jid]), M.jnt_range[2 * jid + 1]);
      }
    }
    V3 p; M3 R; V3 ax[6], an[6];
    arm_fk(q, ee, &p, &R, ax, an);
    double err = norm(p - tgt);
    for (int j = 0; j < 5; j++) out5[j] = q[j];
    return err <= 0.02;  // :502-510
  }
  int ik_ee_body = -1, ik_base_body = -1;

  // sim.render(w, h, camera, depth=True) + fliplr(flipud(.)) + depth_2_meters (MujocoController.py:708-740), restated as a
  // per-pixel ray cast against the scene's convex shapes in fp64 (the engine's ur5_raster.h does the same in fp32).
  // mode 0: metres along the optical axis, 1: GL window depth. rgb: flat albedo x (0.35 + 0.65 max(0, n.l)).
  void render(int cam, int W, int H, int mode, unsigned char* rgb, float* depth) {
    kinematics();
    const double PI = 3.14159265358979323846;
    double f = 0.5 * H / std::tan(M.cam_fovy[cam] * PI / 360.0);
    V3 o = v3(M.cam_pos + 3 * cam);
    M3 Rc; for (int k = 0; k < 9; k++) Rc.m[k] = M.cam_mat[9 * cam + k];
    double near = M.znear * M.extent, far = M.zfar * M.extent;
    V3 light = normalized(V3(1.0, -1.0, 3.0 - 0.435));
    const double sky[3] = {0.65, 0.65, 0.9};
    for (int py = 0; py < H; py++) for (int px = 0; px < W; px++) {
      double xc = ((W - 1 - px) + 0.5 - 0.5 * W) / f, yc = ((H - 1 - py) + 0.5 - 0.5 * H) / f;
      V3 d = mul(Rc, V3(xc, yc, -1.0));
      double tbest = far; V3 nbest(0, 0, 1); int gbest = -1;
      for (int g = 0; g < M.ngeom; g++) {
        V3 ol = mulT(gxmat[g], o - gxpos[g]), dl = mulT(gxmat[g], d);
        V3 s = v3(M.geom_size + 3 * g);
        double t = -1; V3 nl(0, 0, 1);
        int type = M.geom_type[g];
        if (type == GEOM_PLANE) { if (dl.z < -1e-12) t = -ol.z / dl.z; }
        else if (type == GEOM_SPHERE) {
          double dd = dot(dl, dl), b = dot(ol, dl), c = dot(ol, ol) - s.x * s.x, disc = b * b - dd * c;
          if (disc >= 0) { t = (-b - std::sqrt(disc)) / dd; nl = (ol + dl * t) * (1.0 / s.x); }
        } else if (type == GEOM_BOX) {
          double t0 = -1e300, t1 = 1e300; int ax = 0; double sg = 1; bool miss = false;
          for (int k = 0; k < 3; k++) {
            if (std::fabs(dl[k]) < 1e-15) { if (std::fabs(ol[k]) > s[k]) miss = true; continue; }
            double ta = (-s[k] - ol[k]) / dl[k], tb = (s[k] - ol[k]) / dl[k];
            double tn = std::min(ta, tb), tf = std::max(ta, tb);
            if (tn > t0) { t0 = tn; ax = k; sg = dl[k] > 0 ? -1.0 : 1.0; }
            t1 = std::min(t1, tf);
          }
          if (!miss && t0 <= t1 && t0 > 0) { t = t0; nl = V3(); nl[ax] = sg; }
        } else if (type == GEOM_CYLINDER || type == GEOM_CAPSULE) {   // axis = local z, radius s.x, half length s.y
          double r = s.x, hl = s.y, dd = dot(dl, dl);
          double a = dl.x * dl.x + dl.y * dl.y, b = ol.x * dl.x + ol.y * dl.y, cc = ol.x * ol.x + ol.y * ol.y - r * r;
          double ts0 = -1e300, ts1 = 1e300; bool miss = false;
          if (a > 1e-15) { double disc = b * b - a * cc; if (disc < 0) miss = true; else { double sq = std::sqrt(disc); ts0 = (-b - sq) / a; ts1 = (-b + sq) / a; } }
          else if (cc > 0) miss = true;
          if (!miss && type == GEOM_CYLINDER) {
            double tz0 = -1e300, tz1 = 1e300;
            if (std::fabs(dl.z) > 1e-15) { double ta = (-hl - ol.z) / dl.z, tb = (hl - ol.z) / dl.z; tz0 = std::min(ta, tb); tz1 = std::max(ta, tb); }
            else if (std::fabs(ol.z) > hl) miss = true;
            double te = std::max(ts0, tz0), tx = std::min(ts1, tz1);
            if (!miss && te <= tx && te > 0) {
              t = te;
              if (ts0 > tz0) { V3 p = ol + dl * t; nl = V3(p.x / r, p.y / r, 0); } else nl = V3(0, 0, dl.z > 0 ? -1.0 : 1.0);
            }
          } else if (type == GEOM_CAPSULE) {
            double tb2 = 1e300;
            if (!miss && a > 1e-15) { double z = ol.z + dl.z * ts0; if (std::fabs(z) <= hl && ts0 > 0) { tb2 = ts0; V3 p = ol + dl * ts0; nl = V3(p.x / r, p.y / r, 0); } }
            for (int e = 0; e < 2; e++) {
              double zc = e == 0 ? hl : -hl;
              V3 oc(ol.x, ol.y, ol.z - zc);
              double bs = dot(oc, dl), cs = dot(oc, oc) - r * r, disc = bs * bs - dd * cs;
              if (disc < 0) continue;
              double ts = (-bs - std::sqrt(disc)) / dd;
              V3 p = oc + dl * ts;
              if (ts > 0 && ts < tb2 && (e == 0 ? p.z >= 0 : p.z <= 0)) { tb2 = ts; nl = p * (1.0 / r); }
            }
            if (tb2 < 1e299) t = tb2;
          }
        } else if (type == GEOM_MESH) {
          int k0 = M.vis_planeadr[M.geom_meshid[g]], kn = M.vis_planenum[M.geom_meshid[g]];
          double t0 = -1e300, t1 = 1e300; int kb = -1; bool miss = false;
          for (int k = 0; k < kn && !miss; k++) {
            const double* pl = M.vis_plane + 4 * (k0 + k);
            double nd = pl[0] * dl.x + pl[1] * dl.y + pl[2] * dl.z, no = pl[0] * ol.x + pl[1] * ol.y + pl[2] * ol.z + pl[3];
            if (std::fabs(nd) < 1e-15) { if (no > 0) miss = true; continue; }
            double tt = -no / nd;
            if (nd < 0) { if (tt > t0) { t0 = tt; kb = k; } } else t1 = std::min(t1, tt);
            if (t0 > t1) miss = true;
          }
          if (!miss && kb >= 0 && t0 > 0) { t = t0; const double* pl = M.vis_plane + 4 * (k0 + kb); nl = V3(pl[0], pl[1], pl[2]); }
        }
        if (t > near && t < tbest) { tbest = t; nbest = mul(gxmat[g], nl); gbest = g; }
      }
      size_t idx = (size_t)py * W + px;
      if (gbest < 0) for (int k = 0; k < 3; k++) rgb[3 * idx + k] = (unsigned char)(255.0 * sky[k]);
      else {
        double sh = 0.35 + 0.65 * std::max(0.0, dot(nbest, light));
        for (int k = 0; k < 3; k++) rgb[3 * idx + k] = (unsigned char)std::min(255.0, 255.0 * M.geom_rgba[4 * gbest + k] * sh + 0.5);
      }
      depth[idx] = (float)(mode == 0 ? tbest : (1.0 - near / tbest) / (1.0 - near / far));
    }
  }

  int move_ee(const double* xyz, double tolerance, int max_steps) {  // :446-465
    double q5[5];
    if (!ik(xyz, q5)) { last_steps = 0; ikfail_count++; return RES_IK_FAIL; }
    return move_group(0x1f, q5, tolerance, max_steps);
  }
  int rotate_wrist3(double degrees) {  // GraspingEnv.py:193-197
    target[5] = degrees * M_PI / 180.0;
    return move_group(mask_all(), nullptr, 0.05, 500);
  }

  // GraspingEnv.py:205-386. check_mode 0 = in-tree script (IT2+: check after moving to the drop position, 1000 steps);
  // check_mode 1 = IT1 (README.md:20): lift straight up, close_gripper(max_steps=500), then carry on;
  // check_mode 2 = the in-tree script with demo_mode=True (GraspingEnv.py:318-321): the check at the drop position lasts 100 steps.
  // phase_steps[12] receives last_movement_steps of each scripted movement (0 when skipped).
  int grasp_attempt(const double* coordinates, int rotation, int check_mode, double table_height, int* phase_steps, int* phase_result) {
    static const double rot_deg[6] = {0, 30, 60, 90, -30, -60};  // GraspingEnv.py:40
    for (int i = 0; i < 12; i++) { phase_steps[i] = 0; phase_result[i] = -1; }
    double c1[3] = {coordinates[0], coordinates[1], 1.1};
    int result1 = move_ee(c1, 0.05, 1000);                       // :212
    phase_steps[0] = last_steps; phase_result[0] = result1;
    if (result1 == RES_IK_FAIL) {                               // :227
      double cc[3] = {0.0, -0.6, 1.1};
      result1 = move_ee(cc, 0.05, 1000);
      phase_steps[0] = last_steps; phase_result[0] = result1;
    }
    bool result_grasp = false;
    int result2 = -1;
    if (result1 == RES_MAX_STEPS) {                             // :242
      result_grasp = false;
    } else {
      phase_result[1] = rotate_wrist3(rot_deg[rotation]);       // :252
      phase_steps[1] = last_steps;
      phase_result[2] = open_gripper(true);                     // :255
      phase_steps[2] = last_steps;
      double c2[3] = {coordinates[0], coordinates[1], std::max(table_height, coordinates[2] - 0.01)};  // :258-259
      result2 = move_ee(c2, 0.01, 300);                         // :260
      phase_steps[3] = last_steps; phase_result[3] = result2;
      if (result2 == RES_MAX_STEPS) {
        result_grasp = false;                                   // :272-274
      } else {
        stay(100);                                              // :277
        result_grasp = grasp();                                 // :278
        phase_steps[5] = last_steps; phase_result[5] = result_grasp ? RES_MAX_STEPS : RES_SUCCESS;
      }
    }
    pid[0].Kp = 10.0;                                           // :282
    int result_final = -1;
    if (check_mode == 1) {
      // IT1: straight up, then the 500-step closing check
      double cu[3] = {coordinates[0], coordinates[1], 1.1};
      phase_result[6] = move_ee(cu, 0.05, 1000);
      phase_steps[6] = last_steps;
      if (result_grasp) { result_final = close_gripper(500); phase_steps[9] = last_steps; phase_result[9] = result_final; }
    }
    double cc[3] = {0.0, -0.6, 1.1};
    phase_result[7] = move_ee(cc, 0.05, 1000);                  // :285
    phase_steps[7] = last_steps;
    double cd[3] = {0.6, 0.0, 1.15};
    phase_result[8] = move_ee(cd, 0.01, 1200);                  // :297
    phase_steps[8] = last_steps;
    if (check_mode != 1 && result_grasp) {                      // :312-321 (check_mode 2 = demo_mode: close_gripper(max_steps=100))
      result_final = close_gripper(check_mode == 2 ? 100 : 1000);
      phase_steps[9] = last_steps; phase_result[9] = result_final;
    }
    bool grasped = (result_final == RES_MAX_STEPS) && result_grasp;  // :327
    phase_result[10] = open_gripper(false);                     // :338
    phase_steps[10] = last_steps;
    if (grasped) stay(200);                                     // :341-342
    phase_result[11] = rotate_wrist3(0);                        // :345
    phase_steps[11] = last_steps;
    pid[0].Kp = 20.0;                                           // :347
    return grasped ? 1 : 0;
  }

  // GraspingEnv.py:409-477. mode 0 = IT5 (free objects: x U(-.25,.25), y U(-.77,-.43), z U(1,1.5), random unit quat);
  // mode 1 = IT4 (commented reset :435-463: slide x U(-.25,.25), slide y U(-.17,.17), z 0, identity quat).
  // Random numbers: splitmix64 keyed by the caller (20 + env id) instead of numpy's global RNG.
  void reset(uint64_t seed, int mode, int settle) {
    // MujocoEnv.reset() -> sim.reset() [3P]: state back to qpos0, zero velocities / warm start / time
    qpos.assign(M.qpos0, M.qpos0 + nq);
    std::fill(qvel.begin(), qvel.end(), 0.0);
    std::fill(qacc_warmstart.begin(), qacc_warmstart.end(), 0.0);
    std::fill(ctrl.begin(), ctrl.end(), 0.0);
    time = 0;
    static const double home[7] = {0, -1.57, 1.57, -1.57, -1.57, 0.0, 0.3};  // :418
    for (int a = 0; a < nu; a++) qpos[M.jnt_qposadr[M.act_jntid[a]]] = home[a];
    SplitMix rng{seed};
    for (int b = 1; b < M.nbody; b++) {
      if (M.body_parentid[b] != 0 || M.body_jntnum[b] == 0) continue;
      int j0 = M.body_jntadr[b];
      if (M.jnt_type[j0] == JNT_FREE) {
        int qa = M.jnt_qposadr[j0];
        qpos[qa] = rng.uniform(-0.25, 0.25);
        qpos[qa + 1] = rng.uniform(-0.77, -0.43);
        qpos[qa + 2] = rng.uniform(1.0, 1.5);
        double r1 = rng.uniform(), r2 = rng.uniform(), r3 = rng.uniform();  // pyquaternion Quaternion.random() [3P]
        double q1 = std::sqrt(1.0 - r1) * std::sin(2 * M_PI * r2), q2 = std::sqrt(1.0 - r1) * std::cos(2 * M_PI * r2);
        double q3 = std::sqrt(r1) * std::sin(2 * M_PI * r3), q4v = std::sqrt(r1) * std::cos(2 * M_PI * r3);
        qpos[qa + 3] = q1; qpos[qa + 4] = q2; qpos[qa + 5] = q3; qpos[qa + 6] = q4v;
      } else if (M.body_jntnum[b] == 4 && M.jnt_type[j0] == JNT_SLIDE) {
        qpos[M.jnt_qposadr[j0]] = rng.uniform(-0.25, 0.25);
        qpos[M.jnt_qposadr[j0 + 1]] = rng.uniform(-0.17, 0.17);
        qpos[M.jnt_qposadr[j0 + 2]] = 0.0;
        int qa = M.jnt_qposadr[j0 + 3];
        qpos[qa] = 1; qpos[qa + 1] = qpos[qa + 2] = qpos[qa + 3] = 0;
      }
    }
    (void)mode;
    for (int a = 0; a < nu; a++) target[a] = home[a];          // :468-470
    forward_position();
    if (settle) stay(1000);                                     // :473
  }
};

}  // namespace

// =====================================================================================================  C ABI (ctypes, tests only)
extern "C" {
void* ur5o_create(const void* blob, size_t nbytes, int ee_body, int base_body) {
  Sim* s = new Sim();
  if (!s->init(blob, nbytes)) { delete s; return nullptr; }
  s->ik_ee_body = ee_body; s->ik_base_body = base_body;
  return s;
}
void ur5o_destroy(void* h) { delete (Sim*)h; }
int ur5o_nq(void* h) { return ((Sim*)h)->nq; }
int ur5o_nv(void* h) { return ((Sim*)h)->nv; }
int ur5o_nu(void* h) { return ((Sim*)h)->nu; }
void ur5o_set_contact_order(void* h, int mode) { ((Sim*)h)->contact_order = mode; }
void ur5o_set_options(void* h, int contacts_enabled, double pid_dt, int solver) {
  Sim* s = (Sim*)h;
  s->contacts_enabled = contacts_enabled;
  s->solver = solver;
  if (pid_dt > 0) s->pid_dt = pid_dt;
}
// test hook: the primal objective of the constraint QP (what both solvers minimise) at an arbitrary acceleration, and the norm of its gradient there,
// for the rows of the last forward() -- the solver cross-check compares solutions by their cost, which is the convergence measure of a convex problem
void ur5o_primal_cost(void* h, const double* qacc, double* cost, double* gradnorm) {
  Sim* s = (Sim*)h;
  std::vector<double> x(qacc, qacc + s->nv), Ma(s->nv), g(s->nv);
  s->matvecM(x, Ma);
  *cost = s->primal_cost(x, Ma);
  for (int i = 0; i < s->nv; i++) g[i] = Ma[i] - s->qfrc_smooth[i];
  for (const Row& r : s->rows) {
    double jar = s->row_jar(r, x);
    if (!r.unilateral || jar < 0) for (size_t k = 0; k < r.idx.size(); k++) g[r.idx[k]] += r.D * jar * r.J[k];
  }
  double n = 0;
  for (double v : g) n += v * v;
  *gradnorm = std::sqrt(n);
}
void ur5o_set_solver_limits(void* h, int iterations, double tolerance) { ((Sim*)h)->iter_override = iterations; ((Sim*)h)->tol_override = tolerance; }
void ur5o_get_state(void* h, double* qpos, double* qvel, double* warm, double* pidstate) {
  Sim* s = (Sim*)h;
  if (qpos) memcpy(qpos, s->qpos.data(), 8 * s->nq);
  if (qvel) memcpy(qvel, s->qvel.data(), 8 * s->nv);
  if (warm) memcpy(warm, s->qacc_warmstart.data(), 8 * s->nv);
  if (pidstate) for (int j = 0; j < s->nu; j++) {  // [setpoint(target), last_input, last_output, Kp] per actuator
    pidstate[4 * j] = s->target[j]; pidstate[4 * j + 1] = s->pid[j].last_input;
    pidstate[4 * j + 2] = s->pid[j].last_output; pidstate[4 * j + 3] = s->pid[j].Kp;
  }
}
void ur5o_set_state(void* h, const double* qpos, const double* qvel, const double* warm, const double* pidstate) {
  Sim* s = (Sim*)h;
  if (qpos) memcpy(s->qpos.data(), qpos, 8 * s->nq);
  if (qvel) memcpy(s->qvel.data(), qvel, 8 * s->nv);
  if (warm) memcpy(s->qacc_warmstart.data(), warm, 8 * s->nv);
  if (pidstate) for (int j = 0; j < s->nu; j++) {
    s->target[j] = pidstate[4 * j]; s->pid[j].setpoint = pidstate[4 * j]; s->pid[j].last_input = pidstate[4 * j + 1];
    s->pid[j].last_output = pidstate[4 * j + 2]; s->pid[j].Kp = pidstate[4 * j + 3]; s->pid[j].has_last = true;
  }
  s->forward_position();
}
void ur5o_set_ctrl(void* h, const double* ctrl) { Sim* s = (Sim*)h; memcpy(s->ctrl.data(), ctrl, 8 * s->nu); }
void ur5o_get_ctrl(void* h, double* ctrl) { Sim* s = (Sim*)h; memcpy(ctrl, s->ctrl.data(), 8 * s->nu); }
void ur5o_forward(void* h) { ((Sim*)h)->forward(); }
void ur5o_step(void* h, int n) { for (int i = 0; i < n; i++) ((Sim*)h)->step(); }
void ur5o_reset(void* h, uint64_t seed, int mode, int settle) { ((Sim*)h)->reset(seed, mode, settle); }
int ur5o_move_group(void* h, unsigned mask, const double* target, double tol, int max_steps, int* steps) {
  Sim* s = (Sim*)h;
  int r = s->move_group(mask, target, tol, max_steps);
  if (steps) *steps = s->last_steps;
  return r;
}
// move_group with plot=True: the group's joint angles every `every` steps (the reference plots them, create_joint_angle_plot :654-705)
int ur5o_move_group_plot(void* h, unsigned mask, const double* target, double tol, int max_steps, int every, int cap, int* plot_steps,
                         double* plot_q, int* nplot, int* steps) {
  Sim* s = (Sim*)h;
  int stride = 0;
  for (int j = 0; j < s->nu; j++) stride += mask >> j & 1;
  s->plot_every = every; s->plot_cap = cap; s->plot_n = 0; s->plot_stride = stride; s->plot_steps = plot_steps; s->plot_q = plot_q;
  int r = s->move_group(mask, target, tol, max_steps);
  s->plot_every = 0;
  if (nplot) *nplot = s->plot_n;
  if (steps) *steps = s->last_steps;
  return r;
}
void ur5o_stay(void* h, double ms) { ((Sim*)h)->stay(ms); }
int ur5o_ik(void* h, const double* xyz, double* out5) { return ((Sim*)h)->ik(xyz, out5) ? 1 : 0; }
int ur5o_move_ee(void* h, const double* xyz, double tol, int max_steps, int* steps) {
  Sim* s = (Sim*)h;
  int r = s->move_ee(xyz, tol, max_steps);
  if (steps) *steps = s->last_steps;
  return r;
}
int ur5o_open_gripper(void* h, int half) { return ((Sim*)h)->open_gripper(half != 0); }
int ur5o_close_gripper(void* h, int max_steps) { return ((Sim*)h)->close_gripper(max_steps); }
// CPU baseline of bench.py: `nthreads` host threads, one scene each at a time, until `budget_s` of wall time has passed.
// mode 0: IT1 round of scene g = reset(seed 20 + g, settle) + one grasp attempt aimed at object g % 4 (z = 0.91, check_mode 1),
// i.e. exactly bench.py's aimed_actions(); mode 1: the first `nsteps` steps of the many-object drop of scene g. Returns the
// physics steps executed by all threads; *scenes_out = scenes completed; *wall_out = seconds.
// bench.py's "aimed" workload rule (bench.py aim_targets does the same on the device): scene g, round j of its episode aims at the first of
// boxes (g + j + i) % nobj, i = 0.., that still lies on the pick plate, z = 0.91; an empty plate gets an attempt at its centre.
static void bench_aim(const Sim& s, int g, int j, double* xyz) {
  int objs[16], nobj = 0;
  for (int b = 1; b < s.M.nbody && nobj < 16; b++)
    if (s.M.body_parentid[b] == 0 && s.M.body_jntnum[b] == 4 && s.M.jnt_type[s.M.body_jntadr[b]] == JNT_SLIDE) objs[nobj++] = b;
  xyz[0] = 0; xyz[1] = -0.6; xyz[2] = 0.91;
  for (int i = 0; i < nobj; i++) {
    const int b = objs[(g + j + i) % nobj], qa = s.M.jnt_qposadr[s.M.body_jntadr[b]];
    const double x = s.M.body_pos[3 * b] + s.qpos[qa], y = s.M.body_pos[3 * b + 1] + s.qpos[qa + 1], z = s.M.body_pos[3 * b + 2] + s.qpos[qa + 2];
    if (std::fabs(x) <= 0.27 && std::fabs(y + 0.6) <= 0.19 && z >= 0.905 && z <= 1.0) { xyz[0] = x; xyz[1] = y; return; }
  }
}
// bench.py's aiming rule for 40-object piles (It1Rounds.pile_box_actions = tools/pile_aim.py pick_box, evaluated on the GPU by the timed rounds): the box inside
// the bin with the most level top face whose sides are parallel to the fingers at wrist angle 0 / +30 / -30 deg (rotation index 0 / 1 / 4) and nothing lying on it;
// without such a box the highest object inside the bin (rotation cycling), an empty bin gets an attempt at its centre. Returns the rotation index.
static int bench_pile_aim(const Sim& s, int g, int r, double* xyz) {
  const double PI = 3.14159265358979323846;
  std::vector<int> objs;
  for (int b = 1; b < s.M.nbody; b++)
    if (s.M.body_parentid[b] == 0 && s.M.body_jntnum[b] == 1 && s.M.jnt_type[s.M.body_jntadr[b]] == JNT_FREE) objs.push_back(b);
  double best = 1e300; int best_rot = 0; bool found = false;
  for (size_t k = 0; k < objs.size(); k++) {
    const int b = objs[k];
    int gbox = -1;
    for (int gi = 0; gi < s.M.ngeom; gi++) if (s.M.geom_bodyid[gi] == b && s.M.geom_type[gi] == GEOM_BOX) gbox = gi;
    if (gbox < 0) continue;
    const double* q = &s.qpos[s.M.jnt_qposadr[s.M.body_jntadr[b]]];
    if (!(std::fabs(q[0]) < 0.17 && std::fabs(q[1] + 0.6) < 0.10 && q[2] > 0.89)) continue;
    const double w = q[3], x = q[4], y = q[5], z = q[6];
    const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)}, {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)},
                            {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)}};
    int a = 0;
    for (int i = 1; i < 3; i++) if (std::fabs(R[2][i]) > std::fabs(R[2][a])) a = i;
    const double tilt = std::acos(std::min(1.0, std::fabs(R[2][a]))) * 180.0 / PI;
    const int bb = (a + 1) % 3;
    const double want = -std::atan2(R[1][bb], R[0][bb]) * 180.0 / PI;
    const double ang[3] = {0.0, 30.0, -30.0};
    const int rots[3] = {0, 1, 4};
    double mis_min = 1e300; int which = 0;
    for (int i = 0; i < 3; i++) {
      double v = std::fmod(want - ang[i] + 45.0, 90.0);
      if (v < 0) v += 90.0;                                                    // python's / torch.remainder's sign convention
      const double mis = std::fabs(v - 45.0);
      if (mis < mis_min) { mis_min = mis; which = i; }
    }
    bool on_top = false;
    for (size_t o = 0; o < objs.size(); o++) {
      if (o == k) continue;
      const double* p = &s.qpos[s.M.jnt_qposadr[s.M.body_jntadr[objs[o]]]];
      if (std::hypot(p[0] - q[0], p[1] - q[1]) < 0.05 && p[2] - q[2] > 0.01) on_top = true;
    }
    const double score = tilt + mis_min + (on_top ? 100.0 : 0.0);
    if (score < best) { best = score; best_rot = rots[which]; xyz[0] = q[0]; xyz[1] = q[1]; found = true; }
  }
  if (found) return best_rot;
  xyz[0] = 0; xyz[1] = -0.6;
  double top = -1;
  for (int b : objs) {
    const double* q = &s.qpos[s.M.jnt_qposadr[s.M.body_jntadr[b]]];
    if (std::fabs(q[0]) < 0.2 && std::fabs(q[1] + 0.6) < 0.13 && q[2] > 0.85 && q[2] > top) { top = q[2]; xyz[0] = q[0]; xyz[1] = q[1]; }
  }
  return (g / 4 + r) % 6;
}
// the observation step of bench.py's rendered workloads on the CPU: render the 200x200 RGB-D image of camera `cam` (GraspEnv.get_observation), find the
// pixel over world (x, y) at table height (the inverse of the renderer's own pixel -> ray map) and return the world height the depth image shows there
static int g_batch_cam = 1;   // "top_down" of both scene files
void ur5o_batch_camera(int cam) { g_batch_cam = cam; }
static double bench_observed_height(Sim& s, double x, double y, std::vector<unsigned char>& rgb, std::vector<float>& depth) {
  const int W = 200, H = 200, cam = g_batch_cam;
  rgb.resize((size_t)W * H * 3); depth.resize((size_t)W * H);
  s.render(cam, W, H, 0, rgb.data(), depth.data());
  const double PI = 3.14159265358979323846;
  const double f = 0.5 * H / std::tan(s.M.cam_fovy[cam] * PI / 360.0);
  V3 o = v3(s.M.cam_pos + 3 * cam);
  M3 Rc; for (int k = 0; k < 9; k++) Rc.m[k] = s.M.cam_mat[9 * cam + k];
  V3 v = mulT(Rc, V3(x, y, 0.91) - o);
  int px = (int)std::lround(W - 1 - (v.x / -v.z * f + 0.5 * W - 0.5)), py = (int)std::lround(H - 1 - (v.y / -v.z * f + 0.5 * H - 0.5));
  px = px < 0 ? 0 : (px > W - 1 ? W - 1 : px); py = py < 0 ? 0 : (py > H - 1 ? H - 1 : py);
  return o.z - (double)depth[(size_t)py * W + px];
}
// mode 0: reset + settle + ONE aimed attempt per scene (round-1 sample, kept for comparison); mode 1: reset + nsteps raw steps (many-object drop);
// mode 2: bench.py's stationary IT1 workload -- whole episodes of reset + settle + `nsteps` aimed attempts (bench_aim) per scene;
// mode 3: the same episodes on the rendered workload (bench.py kind "it4"): every attempt renders the observation and takes its height from the depth
// image, in-tree script (check_mode 0); mode 4: 40-object piles (kind "many"): reset + 1000 ms settle + ONE rendered attempt aimed by the timed GPU
// rounds' box rule (bench_pile_aim) per scene (a whole episode of a pile is minutes of CPU time: the sample is bounded to one attempt).
long ur5o_batch(const void* blob, size_t nbytes, int ee_body, int base_body, int nthreads, double budget_s, int mode, int nsteps,
                long* scenes_out, double* wall_out, long* attempts_out, long* success_out) {
  std::atomic<long> steps{0}, scenes{0}, attempts{0}, success{0};
  std::atomic<int> next{0};
  std::atomic<bool> stop{false};
  auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  auto work = [&]() {
    Sim s;
    if (!s.init(blob, nbytes)) return;
    s.ik_ee_body = ee_body; s.ik_base_body = base_body;
    if (mode == 4) s.stop_flag = &stop;                            // piles: one scene outlasts the budget; cut it off
    std::vector<unsigned char> rgb;
    std::vector<float> depth;
    while (elapsed() < budget_s) {
      int g = next.fetch_add(1);
      long before = s.total_steps;
      if (mode == 0) {
        s.reset(20 + (uint64_t)g, 1, 1);
        int nobj = (s.nq - 8) / 7, k = g % (nobj < 4 ? nobj : 4);
        double xyz[3] = {s.qpos[8 + 7 * k], -0.6 + s.qpos[8 + 7 * k + 1], 0.91};
        int ps[12], pr[12];
        success += s.grasp_attempt(xyz, (g / 4) % 6, 1, 0.91, ps, pr);
        attempts++;
      } else if (mode == 2 || mode == 3) {
        s.reset(20 + (uint64_t)g, 1, 1);
        for (int j = 0; j < nsteps; j++) {
          double xyz[3];
          bench_aim(s, g, j, xyz);
          if (mode == 3) xyz[2] = bench_observed_height(s, xyz[0], xyz[1], rgb, depth);
          int ps[12], pr[12];
          success += s.grasp_attempt(xyz, (g / 4 + j) % 6, mode == 2 ? 1 : 0, 0.91, ps, pr);
          attempts++;
        }
      } else if (mode == 4) {
        s.reset(20 + (uint64_t)g, 1, 1);
        double xyz[3] = {0, -0.6, 0.91};
        const int rot = bench_pile_aim(s, g, 0, xyz);               // the timed GPU rounds' rule (bench.py It1Rounds.pile_box_actions)
        xyz[2] = bench_observed_height(s, xyz[0], xyz[1], rgb, depth);
        int ps[12], pr[12];
        int r = s.grasp_attempt(xyz, rot, 0, 0.91, ps, pr);
        if (!stop.load()) { success += r; attempts++; }
      } else {
        s.reset(20 + (uint64_t)g, 1, 0);
        for (int i = 0; i < nsteps; i++) s.step();
      }
      steps += s.total_steps - before;
      scenes++;
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; t++) th.emplace_back(work);
  std::thread timer([&]() { while (elapsed() < budget_s && scenes.load() < 0x7fffffff && !stop.load()) std::this_thread::sleep_for(std::chrono::milliseconds(20)); stop = true; });
  for (auto& t : th) t.join();
  stop = true;
  timer.join();
  if (scenes_out) *scenes_out = scenes.load();
  if (wall_out) *wall_out = elapsed();
  if (attempts_out) *attempts_out = attempts.load();
  if (success_out) *success_out = success.load();
  return steps.load();
}
// checkpoints (ascending step counts, counted from now): ur5o_get_checkpoints copies the qpos rows recorded so far, returns their number
void ur5o_set_cholesky_order(void* h, int mode) { ((Sim*)h)->cholesky_order = mode; }
void ur5o_set_checkpoints(void* h, const int* steps, int n) { Sim* s = (Sim*)h; s->ckpt_steps.assign(steps, steps + n); s->ckpt_qpos.clear(); s->ckpt_base = s->total_steps; }
int ur5o_get_checkpoints(void* h, double* out) {
  Sim* s = (Sim*)h;
  for (size_t k = 0; k < s->ckpt_qpos.size(); k++) if (out) memcpy(out + k * s->nq, s->ckpt_qpos[k].data(), 8ull * s->nq);
  return (int)s->ckpt_qpos.size();
}
int ur5o_bench_pile_aim(void* h, int g, int r, double* xyz) { return bench_pile_aim(*(Sim*)h, g, r, xyz); }
int ur5o_grasp_attempt(void* h, const double* xyz, int rot, int check_mode, double table_height, int* phase_steps, int* phase_result) {
  return ((Sim*)h)->grasp_attempt(xyz, rot, check_mode, table_height, phase_steps, phase_result);
}
long ur5o_total_steps(void* h) { return ((Sim*)h)->total_steps; }
long ur5o_bad_state_resets(void* h) { return ((Sim*)h)->bad_state_resets; }
long ur5o_solver_iters(void* h) { return ((Sim*)h)->solver_iter_total; }
int ur5o_last_steps(void* h) { return ((Sim*)h)->last_steps; }
// introspection
void ur5o_body_xpos(void* h, double* out) { Sim* s = (Sim*)h; for (int b = 0; b < s->M.nbody; b++) for (int k = 0; k < 3; k++) out[3 * b + k] = s->xpos[b][k]; }
void ur5o_body_xmat(void* h, double* out) { Sim* s = (Sim*)h; for (int b = 0; b < s->M.nbody; b++) memcpy(out + 9 * b, s->xmat[b].m, 72); }
void ur5o_geom_pose(void* h, double* out) { Sim* s = (Sim*)h; for (int g = 0; g < s->M.ngeom; g++) { for (int k = 0; k < 3; k++) out[12 * g + k] = s->gxpos[g][k]; memcpy(out + 12 * g + 3, s->gxmat[g].m, 72); } }
void ur5o_mass_matrix(void* h, double* out) { Sim* s = (Sim*)h; memcpy(out, s->Mm.data(), 8ull * s->nv * s->nv); }
void ur5o_get_vec(void* h, int which, double* out) {
  Sim* s = (Sim*)h;
  const std::vector<double>* v[] = {&s->qfrc_bias, &s->qfrc_passive, &s->qfrc_actuator, &s->qacc_smooth, &s->qacc, &s->qfrc_constraint};
  memcpy(out, v[which]->data(), 8 * s->nv);
}
void ur5o_render(void* h, int cam, int W, int H, int mode, unsigned char* rgb, float* depth) { ((Sim*)h)->render(cam, W, H, mode, rgb, depth); }
int ur5o_ncon(void* h) { return (int)((Sim*)h)->contacts.size(); }
int ur5o_nefc(void* h) { return (int)((Sim*)h)->rows.size(); }
int ur5o_solver_iter_last(void* h) { return ((Sim*)h)->solver_iter_last; }
// Newton trace of the last solve: ur5o_newton_trace(h, 1) switches recording on; ur5o_get_newton_trace copies [evaluations][rows] active flags (returns the evaluations)
void ur5o_newton_trace(void* h, int on) { ((Sim*)h)->trace_newton = on != 0; }
int ur5o_get_newton_trace(void* h, unsigned char* out, int cap_evals) {
  Sim* s = (Sim*)h;
  const int ne = (int)s->rows.size(), n = (int)s->newton_trace.size();
  for (int e = 0; e < n && e < cap_evals; e++) if (out && (int)s->newton_trace[e].size() == ne) memcpy(out + (size_t)e * ne, s->newton_trace[e].data(), ne);
  return n;
}
// per row: the two dofs' bodies are not stored; the contact a row belongs to is (contact index) for contact rows and -1 for equality / limit rows
void ur5o_get_row_contacts(void* h, int* out) {
  Sim* s = (Sim*)h;
  for (size_t i = 0; i < s->rows.size(); i++) out[i] = -1;
  for (size_t ci = 0; ci < s->contacts.size(); ci++) {
    const Contact& c = s->contacts[ci];
    const int nr = c.dim == 1 ? 1 : 2 * (c.dim - 1);
    if (c.efc_address >= 0) for (int k = 0; k < nr && c.efc_address + k < (int)s->rows.size(); k++) out[c.efc_address + k] = (int)ci;
  }
}
// per contact: dist, pos[3], normal[3], geom1, geom2, dim, normal force (sum of pyramid row forces), color  -> 12 doubles
void ur5o_get_contacts(void* h, double* out) {
  Sim* s = (Sim*)h;
  for (size_t i = 0; i < s->contacts.size(); i++) {
    const Contact& c = s->contacts[i];
    double* o = out + 12 * i;
    o[0] = c.dist; o[1] = c.pos.x; o[2] = c.pos.y; o[3] = c.pos.z; o[4] = c.frame[0].x; o[5] = c.frame[0].y; o[6] = c.frame[0].z;
    o[7] = c.geom1; o[8] = c.geom2; o[9] = c.dim;
    double f = 0;
    int nr = c.dim == 1 ? 1 : 2 * (c.dim - 1);
    if (c.efc_address >= 0 && c.efc_address + nr <= (int)s->rows.size()) for (int k = 0; k < nr; k++) f += s->rows[c.efc_address + k].force;
    o[10] = f; o[11] = c.color;
  }
}
// per row: pos, aref, R, Adiag, force, unilateral
void ur5o_get_rows(void* h, double* out) {
  Sim* s = (Sim*)h;
  for (size_t i = 0; i < s->rows.size(); i++) {
    const Row& r = s->rows[i];
    double* o = out + 6 * i;
    o[0] = r.pos; o[1] = r.aref; o[2] = r.R; o[3] = r.Adiag; o[4] = r.force; o[5] = r.unilateral;
  }
}
}
