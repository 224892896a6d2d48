// obca_kernel_mw.hip -- the solver of csrc/obca_kernel.hip compiled with FOUR wavefronts per instance (256 threads,
// one wavefront on each SIMD of a CU) for shapes whose rows do not fit one wavefront's registers and whose working set
// needs most of the CU's LDS: long horizons (N = 20 with three obstacles: 694 rows, 121 KB).
#define OBCA_NT 256
#include "obca_kernel.hip"
