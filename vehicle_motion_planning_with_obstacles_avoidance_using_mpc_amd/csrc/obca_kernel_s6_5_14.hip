// obca_kernel_s6_5_14.hip -- the one-wavefront solver of csrc/obca_kernel.hip instantiated for ONE problem shape known at compile
// time (N = 6, 5 obstacles, 14 half-space rows; csrc/obca_device.h: OBCA_SHAPES): obca_ipm_kernel_s6_5_14.
#define OBCA_TU_SHAPE(X) X(6, 5, 14)
#include "obca_kernel.hip"
