// obca_astar_core.h -- the global planner of one rollout as serial code (one GPU lane per rollout).
//
// Restates the reference's grid A* (src/a_star.py:16-102) and the two helpers that turn its route into the
// reference trajectory (rebuild_path :137-147, create_reference_path :189-200) so that routes come out cell for
// cell as in the reference:
//   * open list ordered by (f, (row, col)) -- Python tuple order of heapq entries; (row, col) order = cell index;
//   * neighbours expanded in the order E, W, S, N, SE, SW, NE, NW of (row, col) offsets;
//   * no closed test on pop: stale duplicates in the open list are expanded again with the current g;
//   * a neighbour is (re)queued when g is strictly better than its recorded g (an unseen cell counts as g = 0,
//     `gscore.get(nb, 0)`) or when no open-list entry names it; closed cells are skipped unless strictly better;
//   * the returned chain runs goal -> start WITHOUT the start cell.
// f and g are sums of 1.0 and sqrt(2) in the reference's order (no contraction), the heuristic is the correctly
// rounded sqrt of an exact integer.  Compiles for the device (obca_astar.hip) and the host (tests/native).
#ifndef OBCA_ASTAR_CORE_H
#define OBCA_ASTAR_CORE_H

#include <math.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define AS_FN __host__ __device__ inline
#else
#define AS_FN inline
#endif
#if defined(__clang__)
#define AS_EXACT _Pragma("clang fp contract(off)")
#else
#define AS_EXACT
#endif

namespace astar {

enum { NO_ROUTE = -1, HEAP_OVERFLOW = -2, PATH_TOO_LONG = -3 };

AS_FN int heap_capacity(int cells) { return 4 * cells + 64; }
AS_FN size_t align8(size_t v) { return (v + 7) & ~(size_t)7; }
// bytes of workspace one instance needs (all sub-arrays 8-byte aligned)
AS_FN size_t work_bytes(int cells) {
    const size_t cap = (size_t)heap_capacity(cells);
    return align8(8 * (size_t)cells) + align8(4 * (size_t)cells) + align8((size_t)cells) + align8((size_t)cells) +
           align8(2 * (size_t)cells) + align8(8 * cap) + align8(4 * cap);
}

struct Work {
    double* g; int32_t* from; uint8_t* closed; uint8_t* seen; uint16_t* queued; double* hf; int32_t* hc;
    int cap, n;
};

AS_FN void bind(Work& W, unsigned char* base, int cells) {
    const size_t cap = (size_t)heap_capacity(cells);
    unsigned char* p = base;
    W.g = (double*)p; p += align8(8 * (size_t)cells);
    W.hf = (double*)p; p += align8(8 * cap);
    W.from = (int32_t*)p; p += align8(4 * (size_t)cells);
    W.hc = (int32_t*)p; p += align8(4 * cap);
    W.queued = (uint16_t*)p; p += align8(2 * (size_t)cells);
    W.closed = (uint8_t*)p; p += align8((size_t)cells);
    W.seen = (uint8_t*)p;
    W.cap = (int)cap; W.n = 0;
}

AS_FN bool before(double f1, int c1, double f2, int c2) { return f1 < f2 || (f1 == f2 && c1 < c2); }

AS_FN bool heap_push(Work& W, double f, int c) {
    if (W.n >= W.cap) return false;
    int i = W.n++;
    while (i > 0) {
        const int p = (i - 1) >> 1;
        if (!before(f, c, W.hf[p], W.hc[p])) break;
        W.hf[i] = W.hf[p]; W.hc[i] = W.hc[p];
        i = p;
    }
    W.hf[i] = f; W.hc[i] = c;
    return true;
}

AS_FN int heap_pop(Work& W) {
    const int top = W.hc[0];
    const int last = --W.n;
    if (last > 0) {
        const double f = W.hf[last];
        const int c = W.hc[last];
        int i = 0;
        for (;;) {
            int l = 2 * i + 1;
            if (l >= last) break;
            if (l + 1 < last && before(W.hf[l + 1], W.hc[l + 1], W.hf[l], W.hc[l])) ++l;
            if (!before(W.hf[l], W.hc[l], f, c)) break;
            W.hf[i] = W.hf[l]; W.hc[i] = W.hc[l];
            i = l;
        }
        W.hf[i] = f; W.hc[i] = c;
    }
    return top;
}

AS_FN double cell_distance(int r0, int c0, int r1, int c1) {
    AS_EXACT
    const double dr = (double)(r1 - r0), dc = (double)(c1 - c0);
    return sqrt(dr * dr + dc * dc);                                   // src/a_star.py:30-32
}

// grid [rows*cols] (1 = occupied); start/goal as (row, col).  Writes the reference trajectory x/y/yaw into
// path[0..2][*] (row stride path_max) and returns its length, or a negative code.  yaw9[(dy+1)*3 + (dx+1)] =
// arctan2(dy, dx) as the host evaluates it (lattice steps only take these nine values).
AS_FN int plan(const uint8_t* grid, int rows, int cols, int sr, int sc, int gr, int gc, unsigned char* work,
               const double* yaw9, double* path, int path_max) {
    AS_EXACT
    const int cells = rows * cols;
    Work W;
    bind(W, work, cells);
    for (int i = 0; i < cells; ++i) { W.from[i] = -1; W.closed[i] = 0; W.seen[i] = 0; W.queued[i] = 0; W.g[i] = 0.0; }
    const int start = sr * cols + sc, goal = gr * cols + gc;
    const int dR[8] = {0, 0, 1, -1, 1, 1, -1, -1}, dC[8] = {1, -1, 0, 0, 1, -1, 1, -1};
    W.g[start] = 0.0; W.seen[start] = 1;
    heap_push(W, cell_distance(sr, sc, gr, gc), start);
    W.queued[start] = 1;
    int found = 0;
    while (W.n > 0) {
        const int cur = heap_pop(W);
        W.queued[cur] -= 1;
        if (cur == goal) { found = 1; break; }
        W.closed[cur] = 1;
        const int r = cur / cols, c = cur - r * cols;
        for (int q = 0; q < 8; ++q) {
            const int nr = r + dR[q], nc = c + dC[q];
            const double g = W.g[cur] + cell_distance(r, c, nr, nc);
            if (nr < 0 || nr >= rows || nc < 0 || nc >= cols) continue;
            const int nb = nr * cols + nc;
            if (grid[nb] == 1) continue;
            const double g_nb = W.seen[nb] ? W.g[nb] : 0.0;                      // gscore.get(nb, 0)
            if (W.closed[nb] && g >= g_nb) continue;
            if (g < g_nb || W.queued[nb] == 0) {
                W.from[nb] = cur; W.g[nb] = g; W.seen[nb] = 1;
                if (!heap_push(W, g + cell_distance(nr, nc, gr, gc), nb)) return HEAP_OVERFLOW;
                W.queued[nb] += 1;
            }
        }
    }
    if (!found) return NO_ROUTE;
    int len = 0;
    for (int cur = goal; W.from[cur] >= 0; cur = W.from[cur]) {
        if (++len > cells) return NO_ROUTE;                                      // cannot happen; guards the walk
    }
    if (len > path_max) return PATH_TOO_LONG;
    int i = len - 1;
    for (int cur = goal; W.from[cur] >= 0; cur = W.from[cur], --i) {            // rebuild_path: [x, y] = [col, row]
        path[i] = (double)(cur % cols);
        path[path_max + i] = (double)(cur / cols);
    }
    for (int k = 0; k + 1 < len; ++k) {                                         // create_reference_path
        const int dx = (int)(path[k + 1] - path[k]), dy = (int)(path[path_max + k + 1] - path[path_max + k]);
        path[2 * path_max + k] = yaw9[(dy + 1) * 3 + (dx + 1)];
    }
    if (len >= 2) path[2 * path_max + len - 1] = path[2 * path_max + len - 2];
    else if (len == 1) path[2 * path_max] = 0.0;
    for (int k = len; k < path_max; ++k)                                       // padding repeats the last point
        for (int j = 0; j < 3; ++j) path[j * path_max + k] = len ? path[j * path_max + len - 1] : 0.0;
    return len;
}

}  // namespace astar
#endif
