// obca_kernel_s5_4_10.hip -- the one-wavefront solver of csrc/obca_kernel.hip instantiated for ONE problem shape known at compile
// time (N = 5, 4 obstacles, 10 half-space rows; csrc/obca_device.h: OBCA_SHAPES): obca_ipm_kernel_s5_4_10.
#define OBCA_TU_SHAPE(X) X(5, 4, 10)
#include "obca_kernel.hip"
