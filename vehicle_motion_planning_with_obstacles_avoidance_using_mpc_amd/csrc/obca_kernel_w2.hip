// obca_kernel_w2.hip -- the solver of csrc/obca_kernel.hip compiled with TWO wavefronts per instance and a register
// budget of 256 per lane, so that two waves share each SIMD (four instances per CU as with one wavefront each, but twice
// the lanes per instance for the row algebra and a partner wave to issue while the other waits).
#define OBCA_NT 128
#include "obca_kernel.hip"
