// ur5sim_many.hip -- the many-object variant of the engine (UR5gripper_2_finger_many_objects.xml: 40 objects, condim 6):
// the same source as ur5sim.hip compiled with the limits, thread count and Hessian storage of ur5_devmodel.h's UR5_MANY
// section. One multi-wave workgroup owns one scene; linked into libur5sim.so next to the small-scene unit.
#define UR5_MANY 1
#include "ur5_many_names.h"
#include "ur5sim.hip"
