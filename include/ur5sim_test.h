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
/* capped replay: cap_dev[n] int32 (HIP device pointer; the handle copies it on its stream, like a dispatch order; NULL = off): every following grasp-attempt launch stops scene e after cap_dev[e] physics
   steps and saves its record as it stands. Copies of one scene with caps 10, 20, 40 ... show WHEN the engine and the oracle (or the oracle and its rounding-level
   twins) part on a chaotic pile -- tools/gpu_many_divergence.py, tools/pile_divergence_time.py. Results of scenes whose script ends before the cap are unchanged. */
int ur5_set_step_cap_dev(ur5_sim* h, const int* cap_dev);
/* how many model copies the handle's engine unit has sent to a device in this process: one per handle, by ur5_create. The kernels read the model through the handle's
   own device copy (a kernel argument, struct Engine's only member: csrc/ur5_engine.h), so no launch uploads anything and handles with different models never evict each other (rounds 1-4: one
   __constant__ copy per unit and device, re-written whenever handles with different models took turns). */
long ur5_model_uploads(ur5_sim* h);
#ifdef __cplusplus
}
#endif
#endif
