// obca_kernel_f5_3_6.hip -- the fused closed-loop kernel of csrc/obca_kernel.hip instantiated for ONE family of problem shapes
// known at compile time: N_free = N_fix = 5, 3 static obstacles with 6 half-space rows, 0 / 1 / 2 sensed moving rectangles
// (csrc/obca_device.h: OBCA_FAMILIES): obca_rollout_fused_kernel_f5_3_6.
#define OBCA_TU_FAMILY(X) X(5, 3, 6)
#include "obca_kernel.hip"
