// obca_kernel_s5_3_6.hip -- the one-wavefront solver of csrc/obca_kernel.hip instantiated for ONE problem shape known at compile
// time (N = 5, 3 obstacles, 6 half-space rows; csrc/obca_device.h: OBCA_SHAPES): obca_ipm_kernel_s5_3_6.
#define OBCA_TU_SHAPE(X) X(5, 3, 6)
#include "obca_kernel.hip"
