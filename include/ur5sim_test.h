/* ur5sim_test.h -- introspection hooks of libur5sim.so used by tests/ and tools/ only. NOT part of the drop-in boundary (include/ur5sim.h):
 * nothing here replaces a call of the reference. */
#ifndef UR5SIM_TEST_H
#define UR5SIM_TEST_H
#include "ur5sim.h"
#ifdef __cplusplus
extern "C" {
#endif
/* runs forward dynamics once without integrating and dumps internals (contacts, mass matrix, accelerations): [n][2048] doubles host
   ([n][4096] for many-object models); layout: Engine::dump_body in csrc/ur5_engine.h, decoded by native.BatchSim.forward_debug */
int ur5_forward_debug(ur5_sim* h, double* out);
/* capped replay: cap_dev[n] int32 (HIP device pointer, caller-owned; NULL = off): every following grasp-attempt launch stops scene e after cap_dev[e] physics
   steps and saves its record as it stands. Copies of one scene with caps 10, 20, 40 ... show WHEN the engine and the oracle (or the oracle and its rounding-level
   twins) part on a chaotic pile -- tools/gpu_many_divergence.py, tools/pile_divergence_time.py. Results of scenes whose script ends before the cap are unchanged. */
int ur5_set_step_cap_dev(ur5_sim* h, const int* cap_dev);
/* how often the constant-memory model of the handle's engine unit has been (re-)written in this process. The model of a unit is ONE __constant__ copy per device:
   handles with equal models share it, the small-scene unit and the pile unit have one each (an IT4 handle and a pile handle never evict each other), two handles of the
   SAME unit with DIFFERENT models (IT1 and the six-object scene) re-write 43 KB behind a device synchronisation on every alternating launch. */
long ur5_model_uploads(ur5_sim* h);
#ifdef __cplusplus
}
#endif
#endif
