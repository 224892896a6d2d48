// obca_kernel_mw_s20_3_6.hip -- the four-wavefront solver (csrc/obca_kernel_mw.hip) instantiated for ONE problem shape known at
// compile time (N = 20, 3 obstacles, 6 half-space rows; csrc/obca_device.h: OBCA_MW_SHAPES): obca_ipm_kernel_mw_s20_3_6.
#define OBCA_NT 256
#define OBCA_TU_SHAPE(X) X(20, 3, 6)
#include "obca_kernel.hip"
