// ur5sim_simt_many.cpp -- TEST-ONLY: the many-object variant (one 256-thread workgroup = 4 wavefronts per scene) of the SIMT test build,
// see ur5sim_simt.cpp. Shares that unit's fibre runtime.
#define UR5_MANY 1
#define UR5_SIMT_NO_RUNTIME 1
#include "../../mujoco_rl_ur5_amd/csrc/ur5_many_names.h"
#include "ur5sim_simt.cpp"
