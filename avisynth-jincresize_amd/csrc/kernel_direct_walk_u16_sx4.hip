// kernel_direct_walk_u16_sx4.hip -- see kernel_direct_walk.inc
#define JINC_DIRECT_WALK_T uint16_t
#define JINC_DIRECT_WALK_SX 4
#define JINC_DIRECT_WALK_NAME launch_direct_walk_u16_sx4
#define JINC_DIRECT_RUNS_NAME launch_direct_runs_u16_sx4
#include "kernel_direct_walk.inc"
