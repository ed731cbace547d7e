// ur5sim_emul_many.cpp -- TEST-ONLY: the many-object variant of the lane-emulation build (see ur5sim_emul.cpp).
#define UR5_MANY 1
#include "../../mujoco_rl_ur5_amd/csrc/ur5_many_names.h"
#include "ur5sim_emul.cpp"
