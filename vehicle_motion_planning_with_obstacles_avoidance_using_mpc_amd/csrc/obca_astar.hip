// obca_astar.hip -- batched global planner (obca_astar_* of include/obca_mpc.h): one lane per rollout runs the
// serial A* of csrc/obca_astar_core.h on its own grid with a private HBM workspace; the output is the reference
// trajectory in the layout obca_rollouts_reset consumes, so Monte-Carlo worlds never leave the device.
#include <hip/hip_runtime.h>
#include "../../include/obca_mpc.h"
#include "obca_astar_core.h"

namespace {

struct AstarLaunch {
    const uint8_t* grid; const int32_t *start, *goal;
    int32_t B, rows, cols, path_max;
    double* path; int32_t* path_len;
    unsigned char* work; size_t work_stride;
    double yaw9[9];
};

__global__ void __launch_bounds__(64) astar_kernel(AstarLaunch A) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= A.B) return;
    const size_t cells = (size_t)A.rows * A.cols;
    double yaw9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) yaw9[i] = A.yaw9[i];
    A.path_len[b] = astar::plan(A.grid + (size_t)b * cells, A.rows, A.cols, A.start[2 * b], A.start[2 * b + 1], A.goal[2 * b],
                                A.goal[2 * b + 1], A.work + (size_t)b * A.work_stride, yaw9,
                                A.path + (size_t)b * 3 * A.path_max, A.path_max);
}

// Occupancy grid of B worlds (reference mapModel.shape2grid, src/model_map.py:21-56, with reOrderVertex :88-101 and
// world2gridmap :58-71 applied by the caller / here): obstacle k of world b is the axis-aligned bounding box
// (xmin, ymin, xmax, ymax) of its polygon; cells x .. x + int(xmax/res - xmin/res), y .. y + int(ymax/res - ymin/res) are
// set, x = int(xmin/res), y = int(ymin/res) -- the reference's truncations.  One thread per cell: byte work, coalesced.
__global__ void __launch_bounds__(256) rasterise_kernel(const double* __restrict__ boxes, int B, int K, double res, int rows,
                                                        int cols, uint8_t* __restrict__ grid) {
    const size_t cells = (size_t)rows * cols;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * cells) return;
    const int b = (int)(t / cells);
    const int cell = (int)(t - (size_t)b * cells), r = cell / cols, c = cell - r * cols;
    uint8_t occ = 0;
    for (int k = 0; k < K; ++k) {
        const double* q = boxes + ((size_t)b * K + k) * 4;
        const double xmin = q[0] / res, ymin = q[1] / res, xmax = q[2] / res, ymax = q[3] / res;
        if (!(xmin <= xmax) || !(ymin <= ymax)) continue;                   // padding entry (NaN or inverted box)
        const int x = (int)xmin, y = (int)ymin, xl = (int)(xmax - xmin) + 1, yl = (int)(ymax - ymin) + 1;
        if (c >= x && c < x + xl && r >= y && r < y + yl) occ = 1;
    }
    grid[t] = occ;
}

}  // namespace

extern "C" int obca_rasterise_batch(const double* boxes, int32_t B, int32_t K, double resolution, int32_t rows, int32_t cols,
                                    uint8_t* grid, void* hip_stream) {
    if (!boxes || !grid || B < 0 || K < 0 || rows < 1 || cols < 1 || !(resolution > 0.0)) return OBCA_E_INVAL;
    if (B == 0) return OBCA_OK;
    const size_t total = (size_t)B * rows * cols;
    hipLaunchKernelGGL(rasterise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, boxes, B, K,
                       resolution, rows, cols, grid);
    return hipGetLastError() == hipSuccess ? OBCA_OK : OBCA_E_HIP;
}

extern "C" int64_t obca_astar_workspace_bytes(int32_t B, int32_t rows, int32_t cols) {
    if (B < 0 || rows < 1 || cols < 1 || (int64_t)rows * cols > 65535) return -1;
    return (int64_t)astar::work_bytes(rows * cols) * (B > 0 ? B : 1);
}

extern "C" int obca_astar_batch(const uint8_t* grid, int32_t B, int32_t rows, int32_t cols, const int32_t* start,
                                const int32_t* goal, const double* yaw9, int32_t path_max, double* path,
                                int32_t* path_len, void* workspace, int64_t workspace_bytes, void* hip_stream) {
    if (!grid || !start || !goal || !yaw9 || !path || !path_len || !workspace || B < 0 || path_max < 1) return OBCA_E_INVAL;
    const int64_t need = obca_astar_workspace_bytes(B, rows, cols);
    if (need < 0 || workspace_bytes < need) return OBCA_E_INVAL;
    if (B == 0) return OBCA_OK;
    AstarLaunch A;
    A.grid = grid; A.start = start; A.goal = goal; A.B = B; A.rows = rows; A.cols = cols; A.path_max = path_max;
    A.path = path; A.path_len = path_len; A.work = static_cast<unsigned char*>(workspace);
    A.work_stride = astar::work_bytes(rows * cols);
    for (int i = 0; i < 9; ++i) A.yaw9[i] = yaw9[i];
    // the search is serial and divergent: thin blocks put the rollouts on as many SIMDs as possible
    int tpb = 64;
    while (tpb > 1 && (B + tpb - 1) / tpb < 1024) tpb >>= 1;
    hipLaunchKernelGGL(astar_kernel, dim3((B + tpb - 1) / tpb), dim3(tpb), 0, (hipStream_t)hip_stream, A);
    return hipGetLastError() == hipSuccess ? OBCA_OK : OBCA_E_HIP;
}
