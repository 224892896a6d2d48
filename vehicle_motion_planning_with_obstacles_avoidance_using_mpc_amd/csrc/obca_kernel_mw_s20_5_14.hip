// obca_kernel_mw_s20_5_14.hip -- the four-wavefront solver (csrc/obca_kernel_mw.hip) instantiated for ONE problem shape known at
// compile time (N = 20, 5 obstacles, 14 half-space rows; csrc/obca_device.h: OBCA_MW_SHAPES): obca_ipm_kernel_mw_s20_5_14.
#define OBCA_NT 256
#define OBCA_TU_SHAPE(X) X(20, 5, 14)
#include "obca_kernel.hip"
