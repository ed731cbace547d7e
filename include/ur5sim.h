/* ur5sim.h -- C ABI of the MI355X batched UR5 grasp-scene engine (libur5sim.so).
 *
 * This is the drop-in boundary for the reference's hot path. Each entry point replaces, for a
 * whole batch of N independent scenes, one call the reference makes through mujoco_py / simple_pid / ikpy [3P]
 * (paths under /root/reference):
 *
 *   ur5_create            mujoco_py.load_model_from_path + MjSim + MJ_Controller.__init__/create_lists
 *                         gym_grasper/controller/MujocoController.py:29-51,136-254 ; gym_grasper/envs/GraspingEnv.py:47-51
 *   ur5_reset             GraspEnv.reset_model                      GraspingEnv.py:409-477 (IT4 variant :435-463)
 *   ur5_reset_dev         the same for the flagged scenes only, seeds / flags in device memory (episode boundaries of a batch)
 *   ur5_set_state/get     MujocoEnv.set_state, sim.data.qpos/qvel   GraspingEnv.py:412-466 ; MujocoController.py:319
 *   ur5_set_ctrl          MJ_Controller.actuate_joint_group         MujocoController.py:256-267
 *   ur5_step              sim.step()                                MujocoController.py:379,611
 *   ur5_move_group        MJ_Controller.move_group_to_joint_target  MujocoController.py:269-393
 *   ur5_stay              MJ_Controller.stay                        MujocoController.py:621-636 (deterministic: 10-step chunks)
 *   ur5_move_ee           MJ_Controller.move_ee + ik                MujocoController.py:446-517
 *   ur5_ik                MJ_Controller.ik                          MujocoController.py:467-517
 *   ur5_grasp_attempt     GraspEnv.move_and_grasp                   GraspingEnv.py:205-386
 *   ur5_body_xpos         sim.data.body_xpos[...]                   MujocoController.py:341,488
 *   ur5_render            sim.render(w, h, camera, depth=True) + flips (get_image_data)   MujocoController.py:708-727
 *
 * Conventions: every function returns 0 on success and a negative code on error (ur5_last_error() has the text). Per-env
 * soft outcomes use the result codes below, which the Python facade maps back to the reference's strings "success",
 * "max. steps reached: N", "No valid joint angles received, could not move EE to position.". All array arguments are
 * caller-owned; "host" pointers are ordinary memory, "dev" pointers are HIP device pointers (e.g. torch data_ptr()).
 * One handle = one GPU = one HIP stream; a handle is not thread-safe. No torch types cross this boundary.
 */
#ifndef UR5SIM_H
#define UR5SIM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ur5_sim ur5_sim;

enum { UR5_RES_NONE = -1, UR5_RES_SUCCESS = 0, UR5_RES_MAX_STEPS = 1, UR5_RES_IK_FAIL = 2 };
enum { UR5_ERR_ARG = -1, UR5_ERR_MODEL = -2, UR5_ERR_DEVICE = -3, UR5_ERR_NOGPU = -4 };

typedef struct ur5_config {
  int ee_body;           /* model body id of "ee_link" (MujocoController.py:341) */
  int contacts_enabled;  /* 1 = full physics; 0 = contact-free (tests) */
  double pid_dt;         /* <= 0: model timestep */
} ur5_config;

const char* ur5_last_error(void);
int ur5_create(const void* model_blob, size_t nbytes, int n_env, int device_id, const ur5_config* cfg, ur5_sim** out);
void ur5_destroy(ur5_sim* h);
int ur5_num_envs(const ur5_sim* h);
int ur5_nq(const ur5_sim* h);
int ur5_nv(const ur5_sim* h);
int ur5_nu(const ur5_sim* h);

/* seeds[n] (host). mode is informational (object joint type decides the distribution); settle_ms = 1000 in the reference. */
int ur5_reset(ur5_sim* h, const uint64_t* seeds, int mode, double settle_ms);
/* GraspEnv.reset_model (GraspingEnv.py:409-477) for the scenes whose flag is set, without a host round trip: seeds_dev[n] uint64 and
   mask_dev[n] uint8 (NULL = every scene) are HIP device pointers; a scene's new state depends only on its seed (SplitMix64 stream, same
   sampling code as ur5_reset). Unflagged scenes keep their state and sit the settle launch out. Asynchronous on the handle's stream. */
int ur5_reset_dev(ur5_sim* h, const uint64_t* seeds_dev, const uint8_t* mask_dev, double settle_ms);
/* host arrays, any may be NULL: qpos[n][nq] qvel[n][nv] warmstart[n][nv] pid[n][nu][4] = target, last_input, last_output, Kp */
int ur5_set_state(ur5_sim* h, const double* qpos, const double* qvel, const double* warmstart, const double* pid);
int ur5_get_state(ur5_sim* h, double* qpos, double* qvel, double* warmstart, double* pid);
int ur5_set_ctrl(ur5_sim* h, const double* ctrl /* [n][nu] host */);
int ur5_get_ctrl(ur5_sim* h, double* ctrl);
int ur5_step(ur5_sim* h, int nsteps);
/* mask[n] bit j = actuator j in the group; target [n][8] in group order (NaN keeps the old target) or NULL. host pointers. */
int ur5_move_group(ur5_sim* h, const uint32_t* mask, const double* target, const double* tol, const int* max_steps,
                   int* result, int* steps);
int ur5_stay(ur5_sim* h, double ms);
int ur5_move_ee(ur5_sim* h, const double* xyz /* [n][3] */, const double* tol, const int* max_steps, int* result, int* steps);
/* q5[n][5] arm joint angles; result[n] = UR5_RES_SUCCESS or UR5_RES_IK_FAIL (FK(IK) further than 2 cm from the target) */
int ur5_ik(ur5_sim* h, const double* xyz /* [n][3] */, double* q5, int* result);
/* action[n][4] = world x, y, z, rotation index 0..5 (GraspingEnv.py:40). check_mode 0 = in-tree script, 1 = IT1,
   2 = in-tree script of a demo_mode env (GraspingEnv.py:313-321: the closing check at the drop position lasts 100 steps instead of 1000).
   skip[n] (or NULL): non-zero = GraspEnv.step's rule for targets off the table (GraspingEnv.py:124-131): the scene does not move, reward 0 */
int ur5_grasp_attempt(ur5_sim* h, const double* action, const uint8_t* skip, int check_mode, double table_height, int* reward,
                      int* phase_steps /* [n][12] or NULL */, int* phase_result /* [n][12] or NULL */);
/* same with HIP device pointers: action_dev [n][8] doubles (x y z rot skip - - -), reward_dev [n] int32; asynchronous.
   skip != 0: the scene sits the launch out with reward 0 -- GraspEnv.step's rule for targets off the table (GraspingEnv.py:124-131) */
int ur5_grasp_attempt_dev(ur5_sim* h, const double* action_dev, int check_mode, double table_height, int* reward_dev);
/* GraspEnv.step followed, for the scenes whose episode ends with it, by GraspEnv.reset_model (GraspingEnv.py:409-477) in the SAME launch:
   reset_seeds_dev[n] uint64 (device; 0 = the scene does not reset, NULL = none does) seeds the re-sampling exactly as ur5_reset /
   ur5_reset_dev do, then the scene settles for settle_ms. The reward is the attempt's. Saves the separate, poorly filled settle launch. */
int ur5_grasp_attempt_reset_dev(ur5_sim* h, const double* action_dev, int check_mode, double table_height, int* reward_dev,
                                const uint64_t* reset_seeds_dev, double settle_ms);
/* K consecutive grasp rounds of every scene in ONE launch, with NO lock step between scenes (round 5). The reference's episode loop
   (example_agent.py:15-27: action = policy(observation); env.step(action); reset every few steps) has no barrier between scenes because it has
   one scene; a batch that waits for its slowest scene after every round idles most of the chip when the scenes are few (strong scaling: 512 per
   GPU). For a SCRIPTED policy that reads only the scene's own state the engine can evaluate the policy itself: round k of the launch (= round
   round0 + k of the job) aims with `rule`, runs GraspEnv.step (the script of ur5_grasp_attempt_dev) and, when the scene's episode ends with that
   round, GraspEnv.reset_model (seed = base_seed + global scene id + n_total * episode, then settle_ms) -- then goes straight on to its next round.
   rule kind 1 (bench.py It1Rounds, "aimed"): in round r scene g tries the boxes (g + (g + r) % episode_rounds + i) % nobj, i = 0.., and aims at
   the first whose centre lies on the pick plate (|x| <= plate_half_x, |y - plate_centre_y| <= plate_half_y, z_min <= z <= z_max) at height
   grasp_z with wrist rotation (g / episode_rounds + r) % 6; an empty plate gets an attempt at (fallback_x, fallback_y). Scene g's episode ends
   after the rounds r with (r + 1 + g) % episode_rounds == 0. Every per-scene result is bit-identical to `rounds` launches of
   ur5_grasp_attempt_reset_dev fed with the same rule's actions (tests/test_grasp_rounds.py). reward_dev [rounds][n] int32; action_out_dev
   [rounds][n][8] doubles or NULL: the records the rule produced (x y z rot 0 box-found - -), for the caller's outcome records. Asynchronous.
   Round 6 -- the rendered workloads: rule kind 2 (40-object piles only) is the box rule of bench.py It1Rounds.pile_box_actions / tools/pile_aim.py (the box in
   the bin with the most level top face whose sides are parallel to the fingers at wrist angle 0 / +30 / -30 degrees and nothing lying on it; else the highest
   object in the bin; else the bin's centre). z_from_depth = 1 (either kind; needs ur5_set_observation_dev): the grasp height is cam_z minus the metric depth
   the round's observation shows under the aimed pixel -- GraspEnv.step, GraspingEnv.py:100-104 -- where pixel column = rint((x - cam_x0) / cam_dx),
   row = rint((y - cam_y0) / cam_dy) (the top-down camera's affine map at table height, MujocoController.py:742-806). */
typedef struct {
  int kind, episode_rounds;
  int64_t first_scene_id, n_total;
  uint64_t base_seed;
  double plate_half_x, plate_centre_y, plate_half_y, z_min, z_max, grasp_z, fallback_x, fallback_y;
  int z_from_depth, pad;
  double cam_x0, cam_y0, cam_dx, cam_dy, cam_z;
} ur5_aim_rule;
int ur5_grasp_rounds_dev(ur5_sim* h, const ur5_aim_rule* rule, int round0, int rounds, int check_mode, double table_height, int* reward_dev,
                         double* action_out_dev, double settle_ms);
/* The observation of a round INSIDE the launch (round 6): every following ur5_grasp_rounds_dev launch has each scene render its own RGB-D observation --
   GraspEnv.get_observation, GraspingEnv.py:390-406 -> sim.render(width, height, camera, depth=True) + the two flips, MujocoController.py:708-727 -- at the start of
   each of its rounds, from the state it has then, into frame (round of the launch % frames) of rgb_dev [frames][n][height][width][3] uint8 and depth_dev
   [frames][n][height][width] float32 (depth_mode as ur5_render_dev). The pixels are those of ur5_render_dev on the same state (one ray caster, csrc/ur5_raster.h).
   A scene that renders for itself never waits for a free wave slot, which a stand-alone render between two launches of a stream does while another handle's launch
   holds the chip; and K rounds of a rendered workload fit one launch (frames >= K keeps every round's frame: a rollout with its observations, as
   generate_data.py stores them). rgb_dev = NULL switches it off. Engines whose scene image has no room for the ray caster's working set (the four-box IT1
   scene) refuse with UR5_ERR_MODEL. */
int ur5_set_observation_dev(ur5_sim* h, int camera_id, int width, int height, int depth_mode, uint8_t* rgb_dev, float* depth_dev, int frames);
/* Dispatch order of the following grasp-attempt / settle launches: order_dev[n] int32 (HIP device pointer) is a permutation of the scene
   ids; the engine starts scenes in that order. The handle COPIES the list (device to device, on its stream, ordered with its launches): the
   caller's buffer may be reused or freed once work queued on that stream so far has run; a caller that filled it on another stream
   synchronises first (with ur5_set_stream on the caller's stream there is nothing to do). Results do not depend on it -- only the
   makespan does: a launch ends with its slowest scene, so callers list the scenes with the most work first. NULL = scene order. */
int ur5_set_order_dev(ur5_sim* h, const int* order_dev);
/* The same WITHOUT the copy: the following launches read order_dev itself, so the caller keeps it valid and unmodified until they have run. For callers
   that queue several launches back to back with a precomputed order each (bench.py: the K-rounds-per-launch rollouts of example_agent.py:15-27's
   loop shape): the copy of ur5_set_order_dev is a small device kernel of its own, and between two launches of a stream it waits for a free wave slot
   while another handle's launch holds every register of the chip. NULL = scene order. */
int ur5_set_order_view_dev(ur5_sim* h, const int* order_dev);
int ur5_sync(ur5_sim* h);
/* Queue the handle's launches on a caller-owned HIP stream (hipStream_t, e.g. torch.cuda.current_stream().cuda_stream) so that the
   caller's own device work (action tensors, the CNN) is ordered with them without host synchronisation. external = 1: use hip_stream
   as given (NULL is the device's default stream -- which is what torch's default stream is); external = 0: back to the private stream. */
int ur5_set_stream(ur5_sim* h, void* hip_stream, int external);
/* duration of the last launch in ms, from HIP events recorded on the handle's stream around the kernel */
double ur5_last_launch_ms(ur5_sim* h);
/* engine-kernel time (ms) of every launch since ur5_create whose events a ur5_sync has resolved: callers difference it around a region */
double ur5_kernel_ms_total(ur5_sim* h);
/* counters[n][6] host: total physics steps, last_movement_steps, status bits, Newton iterations, max contacts seen in a step,
   and a variant-specific work counter: many-object engine -- Newton iterations that reused the previous Cholesky factor; wavefront-per-scene
   engine -- physics steps whose broad phase ran from the cached pair list instead of scanning every pair (same candidates either way).
   Status bits (sticky until ur5_reset): 1 = more contacts than slots (30 / 96), 2 = a step produced a non-finite (or > 1e10) state: the
   scene went back to qpos0 like mj_resetData [3P] and keeps running, 4 = more equality/limit rows than slots (16), 8 = more broad-phase
   survivors than slots (64 / 512). Any set bit means the scene's results are not trustworthy. The status column carries the bits of the running episode
   in bits 0-7 and, in bits 8-15, the bits of every episode that ur5_grasp_attempt_reset_dev has ended INSIDE a launch since the last ur5_reset /
   ur5_reset_dev: the reward of an episode-ending attempt is returned by the launch that also resets the scene, and its flag must stay readable. */
int ur5_get_counters(ur5_sim* h, int64_t* counters);
/* world positions of the engine's bodies [n][8 + max objects][3] (max objects: 6, or 40 for many-object models):
   8 robot weld groups (dof order) then the objects */
int ur5_body_xpos(ur5_sim* h, double* out);
/* RGB-D image of every scene from model camera `camera_id` (1 = "top_down" in the reference's files): rgb[n][h][w][3] uint8,
 * depth[n][h][w] float32, already in the orientation get_image_data returns (both flips applied). depth_mode 0 = metres along
 * the optical axis (what depth_2_meters produces), 1 = GL window depth in [0,1] (what sim.render returns). Host pointers. */
int ur5_render(ur5_sim* h, int camera_id, int width, int height, int depth_mode, uint8_t* rgb, float* depth);
/* same with HIP device pointers; asynchronous on the handle's stream */
int ur5_render_dev(ur5_sim* h, int camera_id, int width, int height, int depth_mode, uint8_t* rgb_dev, float* depth_dev);
/* raw device pointer of the [n][192] double state records (layout: csrc/ur5_devmodel.h) */
void* ur5_state_device_ptr(ur5_sim* h);

#ifdef __cplusplus
}
#endif
#endif
