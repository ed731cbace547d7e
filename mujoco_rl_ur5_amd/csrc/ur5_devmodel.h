// ur5_devmodel.h -- the scene description the HIP engine consumes (POD, lives in device global memory).
//
// It is the *specialised* view of a CompiledModel blob (mujoco_rl_ur5_amd/model.py): one articulated robot tree whose
// weld groups each carry exactly one hinge ("cbody" d <-> dof d), plus up to UR5_MAXOBJ free-floating single-geom objects
// (3 slides + ball, UR5gripper_2_finger.xml:233-279, or a free joint, objects.xml), plus static geoms. build_model()
// (ur5sim_host.h) derives it and rejects scenes that do not fit, loudly. The limits below exist twice: the header is compiled once
// per engine variant (UR5_MANY undefined / defined).
#pragma once

#define UR5_MAXRD 8                                // robot dofs == robot weld groups ("cbodies")
#define UR5_MAXNU 8
#ifndef UR5_MAXSR
#define UR5_MAXSR 16                               // equality + limit rows
#endif
#ifndef UR5_MANY
// ---- small scenes (UR5gripper_2_finger.xml, IT1): one 64-lane wavefront per scene, everything in LDS
#define UR5_MAXOBJ 6                               // free objects handled by one wavefront
#define UR5_MAXRG 4                                // robot weld groups that carry collision geoms (wrist_3 group, two knuckle groups)
#define UR5_MAXG 48
#define UR5_MAXDG 16                               // dynamic (robot / object) collidable geoms
#define UR5_MAXPAIR 384
#ifndef UR5_MAXCON
#define UR5_MAXCON 30
#endif
#define UR5_MAXCAND 64
#define UR5_MAXHV 1024                             // hull vertices of collidable meshes
#define UR5_NB 4                                   // base directions per contact: normal, 2 tangents, torsion (condim <= 4)
#define UR5_NT 64                                  // threads per scene
typedef int ur5_pair_t;
#else
// ---- many-object piles (UR5gripper_2_finger_many_objects.xml, IT5: 40 objects, condim 6): one multi-wave workgroup per
// scene, state in LDS (~110 KB: one scene per CU), Newton Hessian in envelope (skyline) storage in global memory
#define UR5_MAXOBJ 40
#define UR5_MAXRG 8                                // every robot weld group may carry collision geoms: scenes compiled with the seven arm-link hulls
                                                   // (mjcf.compile_mjcf(arm_collision=True), UR5gripper_2_finger_many_objects.xml:158-185) fit this variant
#define UR5_MAXG 80
#define UR5_MAXDG 56
#define UR5_MAXPAIR 2560
#define UR5_MAXCON 96                              // 3072 settled + grasped piles of the reference's scene peak at 84 contacts (profiles/r05_z_many_determinism_3072piles.json: 4 scenes above 80);
                                                   // more than 96 raises UR5_ST_CONTACT_OVERFLOW. 160 slots cost 19 KB of LDS that two scenes per CU cannot spare
#define UR5_MAXCAND 512
#define UR5_MAXHV 1024                             // 590 for the gripper's full hulls + 7 x 32 for the optional arm-link hulls
#define UR5_NB 6                                   // + 2 rolling directions (condim 6)
#ifndef UR5_NT
#define UR5_NT 256
#endif
typedef short ur5_pair_t;
#endif
#define UR5_MAXB (UR5_MAXRD + UR5_MAXOBJ)          // cbodies
#define UR5_MAXNV (UR5_MAXRD + 6 * UR5_MAXOBJ)     // 44 / 248
#define UR5_MAXNQ (UR5_MAXRD + 7 * UR5_MAXOBJ)     // 50 / 288

enum { UR5_GEOM_PLANE = 0, UR5_GEOM_SPHERE = 2, UR5_GEOM_CAPSULE = 3, UR5_GEOM_CYLINDER = 5, UR5_GEOM_BOX = 6, UR5_GEOM_MESH = 7 };
enum { UR5_KIND_STATIC = 0, UR5_KIND_ROBOT = 1, UR5_KIND_OBJECT = 2 };

// per-env record in HBM: doubles, [env][field]; one wavefront reads its record with coalesced 64-lane loads
#define UR5_REC_QPOS 0
#define UR5_REC_QVEL (UR5_REC_QPOS + UR5_MAXNQ)        // 50
#define UR5_REC_WARM (UR5_REC_QVEL + UR5_MAXNV)        // 94
#define UR5_REC_CTRL (UR5_REC_WARM + UR5_MAXNV)        // 138
#define UR5_REC_TARGET (UR5_REC_CTRL + UR5_MAXNU)      // 146
#define UR5_REC_PIDIN (UR5_REC_TARGET + UR5_MAXNU)     // 154
#define UR5_REC_PIDOUT (UR5_REC_PIDIN + UR5_MAXNU)     // 162
#define UR5_REC_KP (UR5_REC_PIDOUT + UR5_MAXNU)        // 170
#define UR5_REC_MISC (UR5_REC_KP + UR5_MAXNU)          // 178: total_steps, last_steps, time, status, solver_iters, ncon_max, reused factors / cached broad-phase steps, status bits of episodes ended in-launch
#define UR5_REC_STRIDE ((UR5_REC_MISC + 8 + 63) / 64 * 64)   // 192 / 832

// status bits (per env, sticky until reset)
#define UR5_ST_CONTACT_OVERFLOW 1
#define UR5_ST_NAN 2                                 // a step produced a non-finite (or > 1e10) state: the scene went back to qpos0 (mj_resetData [3P]) and is flagged
#define UR5_ST_ROW_OVERFLOW 4
#define UR5_ST_CAND_OVERFLOW 8                       // more broad-phase survivors than UR5_MAXCAND: pairs were dropped
#define UR5_ST_CACHE_MISMATCH 16                     // test builds only: the broad phase's pair cache disagreed with the full scan

#include <stdint.h>
#include <math.h>
struct Ur5DevModel {
  int nrd, nobj, nv, nq, nu, ngeom, npair, neq, ndg, iterations, ee_cbody, nrg;
  int rd_gslot[UR5_MAXRD], rg_body[UR5_MAXRG];   // contact-accumulator slot of a robot cbody (-1: it has no collision geom) and back
  // ---- robot weld groups (cbody d == dof d)
  int rd_parent[UR5_MAXRD];
  unsigned rd_anc[UR5_MAXRD];   // ancestors incl. self (bit e set: dof e moves cbody d)
  unsigned rd_desc[UR5_MAXRD];  // descendants incl. self
  int rd_limited[UR5_MAXRD];
  double rd_pos[UR5_MAXRD][3], rd_quat[UR5_MAXRD][4];  // weld-root frame in the parent cbody frame (world for the root)
  double rd_mat[UR5_MAXRD][9];                         // rotation matrix of rd_quat
  double rd_jpos[UR5_MAXRD][3], rd_jaxis[UR5_MAXRD][3];
  double rd_mass[UR5_MAXRD], rd_ipos[UR5_MAXRD][3], rd_inertia[UR5_MAXRD][6];  // welded children folded in
  double rd_armature[UR5_MAXRD], rd_damping[UR5_MAXRD], rd_lo[UR5_MAXRD], rd_hi[UR5_MAXRD], rd_invweight[UR5_MAXRD], rd_qpos0[UR5_MAXRD];
  double ref_point[3];
  double ee_pos[3], ee_mat[9];  // ee_link frame inside its cbody (for IK / move_ee)
  // ---- objects
  int obj_kind[UR5_MAXOBJ];     // 0 = 3 slides + ball, 1 = free joint
  int obj_limited[UR5_MAXOBJ][3];
  double obj_pos0[UR5_MAXOBJ][3];
  double obj_mass[UR5_MAXOBJ], obj_inertia[UR5_MAXOBJ][3];
  double obj_arm[UR5_MAXOBJ][2], obj_damp[UR5_MAXOBJ][2];     // [lin, rot]
  double obj_invweight[UR5_MAXOBJ][2];                         // dof_invweight0 lin / rot
  double obj_lo[UR5_MAXOBJ][3], obj_hi[UR5_MAXOBJ][3];
  double obj_qpos0[UR5_MAXOBJ][7];                             // the object's 7 qpos0 entries (mj_resetData of a blown-up scene)
  // ---- geoms
  int g_type[UR5_MAXG], g_kind[UR5_MAXG], g_owner[UR5_MAXG], g_condim[UR5_MAXG], g_vadr[UR5_MAXG], g_vnum[UR5_MAXG], g_dg[UR5_MAXG];
  double g_size[UR5_MAXG][3], g_pos[UR5_MAXG][3], g_mat[UR5_MAXG][9], g_rbound[UR5_MAXG], g_margin[UR5_MAXG];
  double g_friction[UR5_MAXG][3], g_solref[UR5_MAXG][2], g_solimp[UR5_MAXG][5], g_invw[UR5_MAXG][2], g_center[UR5_MAXG][3];
  double g_boxc[UR5_MAXG][3];   // mesh geoms: centre of the hull's bounding box in the geom frame (g_size = its half extents); zero otherwise
  int dg_geom[UR5_MAXDG];
  ur5_pair_t pair_g1[UR5_MAXPAIR], pair_g2[UR5_MAXPAIR];
  double hullvert[UR5_MAXHV][3];
  // ---- joint equality (robot dofs), actuators, options
  int eq_d1[2], eq_d2[2];
  double eq_poly[2][5], eq_solref[2][2], eq_solimp[2][5];
  int act_dof[UR5_MAXNU];
  double act_gear[UR5_MAXNU], act_lo[UR5_MAXNU], act_hi[UR5_MAXNU];
  double pid_kd[UR5_MAXNU], pid_lo[UR5_MAXNU], pid_hi[UR5_MAXNU];
  double timestep, tolerance, impratio, gravity[3], jnt_solref[2], jnt_solimp[5], meaninertia;
};

#if defined(__HIPCC__)
#define UR5_HD __host__ __device__
#else
#define UR5_HD
#endif
struct Ur5SplitMix {
  uint64_t s;
  UR5_HD uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  UR5_HD double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  UR5_HD double uniform(double lo, double hi) { return lo + (hi - lo) * uniform(); }
};

// GraspEnv.reset_model (GraspingEnv.py:409-477) for ONE scene record: MujocoEnv.reset() -> sim.reset() [3P] (qpos0, zero velocity /
// warm start / ctrl / time; the controller's PID state persists), arm teleported to the home pose (:418), objects re-sampled from
// the scene's own SplitMix64 stream (:420-430 free-joint piles; :435-463 the IT4 slide+ball objects). Shared by the host path
// (ur5_reset) and the device path (ur5_reset_dev) so that both produce the same record.
UR5_HD inline void ur5_reset_record(const Ur5DevModel& M, const double* qpos0, double* r, uint64_t seed) {
  const double home[7] = {0, -1.57, 1.57, -1.57, -1.57, 0.0, 0.3};   // GraspingEnv.py:418
  for (int i = 0; i < M.nq; i++) r[UR5_REC_QPOS + i] = qpos0[i];
  for (int i = 0; i < M.nv; i++) { r[UR5_REC_QVEL + i] = 0; r[UR5_REC_WARM + i] = 0; }
  for (int a = 0; a < M.nu; a++) { r[UR5_REC_CTRL + a] = 0; r[UR5_REC_QPOS + M.act_dof[a]] = home[a]; r[UR5_REC_TARGET + a] = home[a]; }
  r[UR5_REC_MISC + 2] = 0; r[UR5_REC_MISC + 3] = 0; r[UR5_REC_MISC + 7] = 0;   // time, status, status bits of the episodes ended inside a launch (the fused attempt + reset restores those)
  Ur5SplitMix rng{seed};
  const double two_pi = 6.283185307179586476925286766559;
  for (int k = 0; k < M.nobj; k++) {
    double* q = r + UR5_REC_QPOS + M.nrd + 7 * k;
    if (M.obj_kind[k] == 1) {  // GraspingEnv.py:420-430
      q[0] = rng.uniform(-0.25, 0.25); q[1] = rng.uniform(-0.77, -0.43); q[2] = rng.uniform(1.0, 1.5);
      double r1 = rng.uniform(), r2 = rng.uniform(), r3 = rng.uniform();
      q[3] = sqrt(1.0 - r1) * sin(two_pi * r2); q[4] = sqrt(1.0 - r1) * cos(two_pi * r2);
      q[5] = sqrt(r1) * sin(two_pi * r3); q[6] = sqrt(r1) * cos(two_pi * r3);
    } else {                   // GraspingEnv.py:435-463 (IT4)
      q[0] = rng.uniform(-0.25, 0.25); q[1] = rng.uniform(-0.17, 0.17); q[2] = 0.0;
      q[3] = 1; q[4] = q[5] = q[6] = 0;
    }
  }
}


// run-time parameters of one launch (wave-uniform unless per-env arrays are given)
enum { UR5_OP_MOVE = 0, UR5_OP_STAY = 1, UR5_OP_MOVE_EE = 2, UR5_OP_GRASP = 3, UR5_OP_STEP = 4, UR5_OP_FORWARD = 5, UR5_OP_IK = 6 };
struct Ur5Launch {
  int op, n_env, contacts_enabled, check_mode;
  double pid_dt, table_height;
  // per-env inputs (device pointers, may be null depending on op)
  const unsigned* group_mask;   // [n]
  const double* target;         // [n][8]  (MOVE: group targets in group order; NaN = keep)  / xyz for MOVE_EE, GRASP ([n][8]: x y z rot)
  const double* tol;            // [n]
  const int* max_steps;         // [n]   (STAY: number of 10-step chunks; STEP: number of steps)
  // per-env outputs
  int* result;                  // [n]
  int* steps;                   // [n]
  int* phase_steps;             // [n][12] (GRASP)
  int* phase_result;            // [n][12]
  double* out;                  // [n][8] (IK: 5 joint angles)
  double* debug;                // optional [n][UR5_DEBUG_STRIDE] introspection dump (FORWARD)
  double* hess;                 // many-object variant: [n][UR5_HESS_STRIDE] envelope storage of the Newton Hessian / its factor
  // GRASP: scenes whose reset_seeds entry is non-zero end the launch with GraspEnv.reset_model (ur5_reset_record + reset_chunks x 10 settle steps)
  const uint64_t* reset_seeds;  // optional [n]
  const double* qpos0;          // model reference pose [nq] (device), needed with reset_seeds
  int reset_chunks;
  const int* order;             // optional [n] permutation: workgroup slot i simulates scene order[i] (longest-first dispatch, ur5_set_order_dev)
  // GRASP, several consecutive rounds of a scene in ONE launch (ur5_grasp_rounds_dev; wavefront-per-scene engine): round k of the launch is round rule_r0 + k of the
  // job. The action of a round is NOT an input: the scene aims by itself, from its own record, with the scripted rule below -- a function of that record only, so a
  // scene never waits for the other scenes' round to end (example_agent.py:15-27 has no barrier between scenes either: it has one scene). result is [rounds][n].
  int rounds;                   // 0 / 1: one round with the caller's action records (every other entry point)
  int rule_kind;                // 0: none; 1: "first candidate object still on the pick plate" (bench.py It1Rounds, rule "aimed"); 2: the box rule for 40-object piles (tools/pile_aim.py)
  int rule_r0, rule_ep;         // first round of the launch, rounds per episode
  long long rule_gid0, rule_ntotal;   // global id of scene 0 of this handle, scenes of the whole job (episode seeds: base + gid + ntotal * episode)
  uint64_t rule_base_seed;
  double rule_plate[8];         // plate: half width in x, centre y, half width in y, lowest / highest z of an object that counts as "on the plate"; grasp z; fallback x, y
  const int* step_cap;          // test hook, optional [n]: the scene stops after this many physics steps of the launch (capped replay: the same attempt cut off at
                                // several step counts shows WHEN two implementations part; include/ur5sim_test.h ur5_set_step_cap_dev)
  double* action_out;           // optional [rounds][n][8]: the action records the rule produced (x y z rot skip box-found - -), for the caller's outcome records
  // Round 6, ruled launches: the OBSERVATION of a round is produced inside the launch, by the scene's own workgroup, at the start of the round (GraspEnv.step's
  // current_observation, GraspingEnv.py:87-88,152: get_observation -> sim.render(width, height, camera, depth=True), MujocoController.py:708-740) -- frame
  // (round of the launch) % obs_frames of obs_rgb [frames][n][h][w][3] / obs_depth [frames][n][h][w] (obs_mode 0: metres along the optical axis, 1: GL window depth).
  // A scene that renders for itself never waits for a free wave slot (a stand-alone render between two launches of a stream does, while another handle's launch holds
  // the chip), and with rule_z_from_depth its grasp height is what the depth image shows under the aimed pixel (GraspingEnv.py:100-104): the rendered workloads of
  // bench.py then run K rounds per launch like the headline.
  const struct Ur5RenderModel* obs_rm;
  unsigned char* obs_rgb;
  float* obs_depth;
  int obs_cam, obs_w, obs_h, obs_mode, obs_frames;
  int rule_z_from_depth;        // the rule's grasp height = rule_cam[4] - (metric depth under the aimed pixel) instead of rule_plate[5]
  double rule_cam[5];           // top-down camera at table height: world x of pixel column 0, world y of pixel row 0, dx per column, dy per row (pixel = rint((x - x0) / dx)); camera z
};
#ifndef UR5_MANY
#define UR5_DEBUG_STRIDE 2048
#else
#define UR5_DEBUG_STRIDE 4096
// Per-scene scratch in global memory (L2-resident, 4 MB per XCD): everything of the Newton factorisation that is touched once per panel instead of once per
// instruction lives here since round 4, so that the LDS image of a pile is < 80 KB and TWO scenes share a CU (two wavefronts per SIMD hide each other's
// latencies; the same-box A/B of the envelope in global memory alone: -6 %). Layout in doubles: envelope of H / its factor (worst case: the full lower
// triangle) | current block column of the factorisation | factored diagonal blocks | staged twist-space Hessian terms of the contact sides.
#define UR5_SCR_PANEL (UR5_MAXNV * (UR5_MAXNV + 1) / 2 + UR5_MAXNV)
#define UR5_SCR_DCACHE (UR5_SCR_PANEL + UR5_MAXNV * UR5_MAXRD)
#define UR5_SCR_STG (UR5_SCR_DCACHE + (UR5_MAXOBJ + 1) * 44)
#define UR5_HESS_STRIDE ((UR5_SCR_STG + 2 * UR5_MAXCON * 21 + 63) / 64 * 64)
#ifndef UR5_HENV_CAP
#define UR5_HENV_CAP (1 << 20)                     // envelopes up to min(this, Lds::HENV_DOUBLES = 2 384) doubles live in the LDS pool (round 5); -DUR5_HENV_CAP=8 forces the global-scratch path (tests)
#endif
#endif
