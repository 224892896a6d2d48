// One bounded measurement (north_star: "MFMA only where rocprof shows it beating the scalar path"): the only GEMM-shaped work of
// the solver is the chain of small dense fp64 products of the stage-serial Riccati sweep (8 x 8 stage matrices; the level-1
// blocks are 10 x 10 LDL^T factorisations, which have no product to hand to a matrix core).  This probe times a DEPENDENT chain
// P <- M P of 8 x 8 fp64 products per wavefront -- one wavefront per SIMD, as the solver runs -- done two ways:
//   valu : the solver's way -- lane (i, j) owns entry (i, j), M's row in registers, P read from LDS, result written back to LDS
//          (one LDS round trip per product, 8 v_fma_f64 per lane);
//   mfma : v_mfma_f64_16x16x4_f64, the 8 x 8 operands padded to the 16 x 16 x 4 tile (two instructions for K = 8).  This chain is
//          the BEST case for the matrix core: the result's register layout (row = 4 reg + lane / 16, column = lane % 16) is
//          exactly the B-operand layout of the next product, so no shuffle is needed; a real Riccati stage (A' P A, transposes,
//          a 2 x 2 inverse in between) would add LDS round trips on top.
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/mfma_probe.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(64) void chain_valu(const double* M, const double* P0, double* out, int T) {
    __shared__ double P[64];
    const int l = threadIdx.x, i = l >> 3, j = l & 7;
    double m[8];
    for (int k = 0; k < 8; ++k) m[k] = M[i * 8 + k];
    P[l] = P0[l];
    __syncthreads();
    double c = 0.0;
    for (int t = 0; t < T; ++t) {
        c = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = fma(m[k], P[k * 8 + j], c);
        __syncthreads();
        P[l] = c;
        __syncthreads();
    }
    out[(size_t)blockIdx.x * 64 + l] = c;
}

__global__ __launch_bounds__(64) void chain_mfma(const double* M, const double* P0, double* out, int T) {
    const int l = threadIdx.x, r = l & 15, kq = l >> 4;
    // A operand of k-step s: lane holds A[row = l & 15][k = 4 s + (l >> 4)]; rows 8..15 are padding (zero)
    const double a0 = r < 8 ? M[r * 8 + kq] : 0.0, a1 = r < 8 ? M[r * 8 + 4 + kq] : 0.0;
    // the running P in the D layout: reg q of lane l = P[row = 4 q + (l >> 4)][col = l & 15]; rows / columns >= 8 padding
    d4 D;
    for (int q = 0; q < 4; ++q) { const int row = 4 * q + kq; D[q] = (row < 8 && r < 8) ? P0[row * 8 + r] : 0.0; }
    for (int t = 0; t < T; ++t) {
        d4 Z = {0.0, 0.0, 0.0, 0.0};
        Z = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, D[0], Z, 0, 0, 0);       // B operand of k-step 0 = rows 0..3 of P = D reg 0
        D = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, D[1], Z, 0, 0, 0);       // k-step 1 = rows 4..7 = D reg 1
    }
    // entry (i, j), i < 8: lane j + 16 (i & 3), reg i >> 2 -> same order as chain_valu's output
    for (int q = 0; q < 2; ++q) { const int row = 4 * q + kq; if (r < 8) out[(size_t)blockIdx.x * 64 + row * 8 + r] = D[q]; }
}

int main() {
    const int waves = 1024, T = 20000;
    double hM[64], hP[64];
    // M = a rotation in every coordinate pair scaled by 0.999..: the chain stays bounded
    for (int i = 0; i < 64; ++i) hM[i] = 0.0;
    for (int b = 0; b < 4; ++b) {
        const double th = 0.1 + 0.07 * b;
        hM[(2 * b) * 8 + 2 * b] = cos(th); hM[(2 * b) * 8 + 2 * b + 1] = -sin(th);
        hM[(2 * b + 1) * 8 + 2 * b] = sin(th); hM[(2 * b + 1) * 8 + 2 * b + 1] = cos(th);
    }
    for (int i = 0; i < 64; ++i) { hM[i] += 1e-3 * ((i * 37) % 11 - 5); hP[i] = 0.01 * ((i * 53) % 17 - 8); }
    double *M, *P, *o1, *o2;
    hipMalloc(&M, 512); hipMalloc(&P, 512); hipMalloc(&o1, waves * 512); hipMalloc(&o2, waves * 512);
    hipMemcpy(M, hM, 512, hipMemcpyHostToDevice); hipMemcpy(P, hP, 512, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2] = {0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(chain_valu, dim3(waves), dim3(64), 0, 0, M, P, o1, T); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms[0], e0, e1);
        hipEventRecord(e0); hipLaunchKernelGGL(chain_mfma, dim3(waves), dim3(64), 0, 0, M, P, o2, T); hipEventRecord(e1);
        hipEventSynchronize(e1); hipEventElapsedTime(&ms[1], e0, e1);
    }
    double h1[64], h2[64], err = 0, mag = 0;
    hipMemcpy(h1, o1, 512, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, 512, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) { err = fmax(err, fabs(h1[i] - h2[i])); mag = fmax(mag, fabs(h1[i])); }
    printf("dependent chain of %d products of 8 x 8 fp64 matrices per wavefront, %d wavefronts (one per SIMD)\n", T, waves);
    printf("valu (LDS round trip per product, 8 v_fma_f64 per lane): %.3f ms = %.1f ns per product\n", ms[0], ms[0] * 1e6 / T);
    printf("mfma (2 x v_mfma_f64_16x16x4_f64, no shuffle needed)   : %.3f ms = %.1f ns per product\n", ms[1], ms[1] * 1e6 / T);
    printf("largest difference of the two results %.3e (largest entry %.3e)\n", err, mag);
    return 0;
}
