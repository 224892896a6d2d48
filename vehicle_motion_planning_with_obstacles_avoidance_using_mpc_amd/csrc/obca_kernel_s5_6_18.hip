// obca_kernel_s5_6_18.hip -- the one-wavefront solver of csrc/obca_kernel.hip instantiated for ONE problem shape known at compile
// time (N = 5, 6 obstacles, 18 half-space rows; csrc/obca_device.h: OBCA_SHAPES): obca_ipm_kernel_s5_6_18.
#define OBCA_TU_SHAPE(X) X(5, 6, 18)
#include "obca_kernel.hip"
