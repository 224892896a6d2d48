// obca_lpi.hip -- lane-per-instance kernel: 64 instances per wavefront, working set in an HBM workspace laid
// out [array element][instance] so that every access of a wave is one coalesced 512-byte transaction.
// The solver itself is csrc/obca_lpi_core.h (same algorithm as the wave-per-instance kernel).
#include <hip/hip_runtime.h>
#include "obca_lpi_core.h"

extern "C" __global__ void __launch_bounds__(64)
obca_lpi_kernel(ObcaLaunch A, double* ws, unsigned long long stride, const int* offm, int ipw) {
    // ipw instances per wavefront (<= 64): the kernel is bound by memory latency, so small batches are spread over
    // more, thinner waves to put one on every SIMD of the chip
    if ((int)threadIdx.x >= ipw) return;
    const size_t inst = (size_t)blockIdx.x * ipw + threadIdx.x;
    if (inst >= (size_t)A.B) return;
    lpi::run_instance(A, ws, (size_t)stride, inst, offm, inst);
}
