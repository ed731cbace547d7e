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
#ifdef __cplusplus
}
#endif
#endif
