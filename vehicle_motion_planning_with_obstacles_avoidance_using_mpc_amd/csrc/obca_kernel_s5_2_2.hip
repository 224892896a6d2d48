// obca_kernel_s5_2_2.hip -- the one-wavefront solver of csrc/obca_kernel.hip instantiated for ONE problem shape known at compile
// time (N = 5, 2 obstacles, 2 half-space rows; csrc/obca_device.h: OBCA_SHAPES): obca_ipm_kernel_s5_2_2.
#define OBCA_TU_SHAPE(X) X(5, 2, 2)
#include "obca_kernel.hip"
