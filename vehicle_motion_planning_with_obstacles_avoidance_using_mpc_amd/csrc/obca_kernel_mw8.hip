// obca_kernel_mw8.hip -- the solver of csrc/obca_kernel.hip compiled with EIGHT wavefronts per instance (512 threads, two
// wavefronts on each SIMD of a CU, 256 registers per lane): experiment for the long-horizon shapes, whose row phases are
// instruction-bound at one wavefront per SIMD (DESIGN.md 4a').
#define OBCA_NT 512
#include "obca_kernel.hip"
