// Marching cubes on the device (second half): decoded SDF grids -> triangle meshes, the
// step eval_3dfront.py performs right after the sampling path through model/diff_utils/util_3d.py:194-236
// (mcubes.marching_cubes(sdf_i, level), level 0.02, PyMCubes).  HBM-bound integer/compare work: no MFMA.
//
// Data layout: sdf [O][n][n][n] fp32, array axes (i, j, k) = vertex coordinates (x, y, z) as in PyMCubes.  Every grid point
// owns its three edges towards +i, +j, +k; an edge whose endpoints lie on different sides of the level (inside: value <
// level) carries exactly one vertex, shared by the up-to-four cubes around it -> meshes come out indexed and watertight.
//   pass 1 (k_mc_count): per grid point the 3 edge flags and, for cell origins, the triangle count of the cube's case;
//                        per 256-point block the two totals.
//   pass 2 (k_mc_scan):  exclusive scan of the block totals per object (one workgroup per object).
//   pass 3 (k_mc_emit):  block-local scans + block bases -> vertex ids; vertices are interpolated along their edge, faces look
//                        up the ids of the cube's 12 edges through their owner points (recomputed flags, no atomics:
//                        the output order is the grid order, deterministic).
// The 256 x 16 case table is a kernel argument (device pointer) generated on the host (echoscene_amd/mc_tables.py).
#include "es_common.h"

namespace {

constexpr int MC_BLOCK = 256;

struct McGrid {
    const float* sdf;
    int O, n;
    float level;
};

__device__ __forceinline__ int mc_flags(const McGrid& g, long base, int i, int j, int k, float v0) {
    // bit a: the edge from (i,j,k) along axis a crosses the level
    const int n = g.n;
    const bool in0 = v0 < g.level;
    int f = 0;
    if (i + 1 < n && ((g.sdf[base + (long)n * n] < g.level) != in0)) f |= 1;
    if (j + 1 < n && ((g.sdf[base + n] < g.level) != in0)) f |= 2;
    if (k + 1 < n && ((g.sdf[base + 1] < g.level) != in0)) f |= 4;
    return f;
}

__device__ __forceinline__ int mc_case(const McGrid& g, long base) {
    const int n = g.n;
    const long sx = (long)n * n, sy = n;
    // corner numbering: v0 (0,0,0) v1 (1,0,0) v2 (1,1,0) v3 (0,1,0) v4..v7 = the same with k + 1
    int c = 0;
    c |= (g.sdf[base] < g.level) ? 1 : 0;
    c |= (g.sdf[base + sx] < g.level) ? 2 : 0;
    c |= (g.sdf[base + sx + sy] < g.level) ? 4 : 0;
    c |= (g.sdf[base + sy] < g.level) ? 8 : 0;
    c |= (g.sdf[base + 1] < g.level) ? 16 : 0;
    c |= (g.sdf[base + sx + 1] < g.level) ? 32 : 0;
    c |= (g.sdf[base + sx + sy + 1] < g.level) ? 64 : 0;
    c |= (g.sdf[base + sy + 1] < g.level) ? 128 : 0;
    return c;
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* sh, int& total) {
    // 256 threads: wave scans by shuffles, 4 wave totals through LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(x, o);
        if (lane >= o) x += y;
    }
    if (lane == 63) sh[w] = x;
    __syncthreads();
    int off = 0;
    for (int q = 0; q < w; ++q) off += sh[q];
    total = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return off + x - v;
}

__global__ __launch_bounds__(MC_BLOCK) void k_mc_count(const McGrid g, const int8_t* tri, int* blk_v, int* blk_t) {
    __shared__ int sh[4];
    const long npt = (long)g.n * g.n * g.n;
    const long p = (long)blockIdx.x * MC_BLOCK + threadIdx.x;           // point index inside the object
    const int o = blockIdx.y;
    int nv = 0, nt = 0;
    if (p < npt) {
        const int n = g.n;
        const int k = (int)(p % n), j = (int)((p / n) % n), i = (int)(p / ((long)n * n));
        const long base = (long)o * npt + p;
        nv = __popc(mc_flags(g, base, i, j, k, g.sdf[base]));
        if (i + 1 < n && j + 1 < n && k + 1 < n) {
            const int8_t* row = tri + mc_case(g, base) * 16;
            while (nt < 5 && row[3 * nt] >= 0) ++nt;
        }
    }
    int tv, tt;
    block_exclusive_scan(nv, sh, tv);
    block_exclusive_scan(nt, sh, tt);
    if (threadIdx.x == 0) {
        blk_v[(long)o * gridDim.x + blockIdx.x] = tv;
        blk_t[(long)o * gridDim.x + blockIdx.x] = tt;
    }
}

// one workgroup per object: exclusive scan of the per-block totals in place, object totals to counts[o] = {verts, tris}
__global__ __launch_bounds__(MC_BLOCK) void k_mc_scan(int* blk_v, int* blk_t, int nblk, int* counts) {
    __shared__ int sh[4];
    const int o = blockIdx.x;
    int* bv = blk_v + (long)o * nblk;
    int* bt = blk_t + (long)o * nblk;
    int run_v = 0, run_t = 0;
    for (int b0 = 0; b0 < nblk; b0 += MC_BLOCK) {
        const int b = b0 + threadIdx.x;
        const int v = b < nblk ? bv[b] : 0, t = b < nblk ? bt[b] : 0;
        int tv, tt;
        const int ev = block_exclusive_scan(v, sh, tv), et = block_exclusive_scan(t, sh, tt);
        if (b < nblk) { bv[b] = run_v + ev; bt[b] = run_t + et; }
        run_v += tv;
        run_t += tt;
    }
    if (threadIdx.x == 0) { counts[2 * o] = run_v; counts[2 * o + 1] = run_t; }
}

// vbase[point] = (first vertex id of the point's edges << 3) | flags   (ids are per object)
__global__ __launch_bounds__(MC_BLOCK) void k_mc_vertices(const McGrid g, const int* blk_v, int* vbase, const long* vofs, float* verts) {
    __shared__ int sh[4];
    const long npt = (long)g.n * g.n * g.n;
    const long p = (long)blockIdx.x * MC_BLOCK + threadIdx.x;
    const int o = blockIdx.y;
    int f = 0, i = 0, j = 0, k = 0;
    long base = 0;
    float v0 = 0.f;
    if (p < npt) {
        const int n = g.n;
        k = (int)(p % n); j = (int)((p / n) % n); i = (int)(p / ((long)n * n));
        base = (long)o * npt + p;
        v0 = g.sdf[base];
        f = mc_flags(g, base, i, j, k, v0);
    }
    int tot;
    const int id0 = blk_v[(long)o * gridDim.x + blockIdx.x] + block_exclusive_scan(__popc(f), sh, tot);
    if (p >= npt) return;
    vbase[base] = (id0 << 3) | f;
    float* out = verts + (vofs[o] + id0) * 3;
    const int n = g.n;
    const long st[3] = {(long)n * n, n, 1};
    int id = 0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (f & (1 << a)) {
            const float v1 = g.sdf[base + st[a]];
            // PyMCubes: x1 + (x2 - x1) * (level - f1) / (f2 - f1); the endpoints differ in side, so f2 != f1
            const float t = (g.level - v0) / (v1 - v0);
            out[3 * id + 0] = (float)i + (a == 0 ? t : 0.f);
            out[3 * id + 1] = (float)j + (a == 1 ? t : 0.f);
            out[3 * id + 2] = (float)k + (a == 2 ? t : 0.f);
            ++id;
        }
    }
}

__global__ __launch_bounds__(MC_BLOCK) void k_mc_faces(const McGrid g, const int8_t* tri, const int* blk_t, const int* vbase, const long* tofs,
                                                      int* faces) {
    __shared__ int sh[4];
    const long npt = (long)g.n * g.n * g.n;
    const long p = (long)blockIdx.x * MC_BLOCK + threadIdx.x;
    const int o = blockIdx.y;
    const int n = g.n;
    int nt = 0;
    const int8_t* row = tri;
    long base = 0;
    if (p < npt) {
        const int k = (int)(p % n), j = (int)((p / n) % n), i = (int)(p / ((long)n * n));
        base = (long)o * npt + p;
        if (i + 1 < n && j + 1 < n && k + 1 < n) {
            row = tri + mc_case(g, base) * 16;
            while (nt < 5 && row[3 * nt] >= 0) ++nt;
        }
    }
    int tot;
    const int t0 = blk_t[(long)o * gridDim.x + blockIdx.x] + block_exclusive_scan(nt, sh, tot);
    if (nt == 0) return;
    const long sx = (long)n * n, sy = n;
    // cube edge -> (owner point offset, axis): e0 v0-v1 .. e3 v3-v0 (k), e4..e7 (k+1), e8..e11 verticals v0-v4, v1-v5, v2-v6, v3-v7
    const long own[12] = {0, sx, sy, 0, 1, sx + 1, sy + 1, 1, 0, sx, sx + sy, sy};
    const int axis[12] = {0, 1, 0, 1, 0, 1, 0, 1, 2, 2, 2, 2};
    int* out = faces + (tofs[o] + t0) * 3;
    for (int t = 0; t < 3 * nt; ++t) {
        const int e = row[t];
        const int vb = vbase[base + own[e]];
        const int fl = vb & 7;
        out[t] = (vb >> 3) + __popc(fl & ((1 << axis[e]) - 1));
    }
}

}  // namespace

extern "C" size_t es_marching_cubes_workspace(int O, int n) {
    const long npt = (long)n * n * n;
    const long nblk = (npt + MC_BLOCK - 1) / MC_BLOCK;
    return (size_t)(2 * (long)O * nblk + (long)O * npt) * sizeof(int32_t);      // block totals (verts, tris) + vbase
}

extern "C" int es_marching_cubes_count(const float* sdf, int O, int n, float level, const int8_t* tri_table, void* workspace,
                                       int32_t* counts, es_stream stream) {
    ES_REQUIRE(sdf && tri_table && workspace && counts && O > 0 && n >= 2 && n <= 1024, "es_marching_cubes_count: bad args (O=%d n=%d)", O, n);
    const long npt = (long)n * n * n;
    const int nblk = (int)((npt + MC_BLOCK - 1) / MC_BLOCK);
    int* blk_v = (int*)workspace;
    int* blk_t = blk_v + (long)O * nblk;
    McGrid g{sdf, O, n, level};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mc_count, dim3(nblk, O), dim3(MC_BLOCK), 0, s, g, tri_table, blk_v, blk_t);
    hipLaunchKernelGGL(k_mc_scan, dim3(O), dim3(MC_BLOCK), 0, s, blk_v, blk_t, nblk, counts);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_marching_cubes_emit(const float* sdf, int O, int n, float level, const int8_t* tri_table, void* workspace,
                                      const int64_t* vert_offset, const int64_t* tri_offset, float* verts, int32_t* faces,
                                      es_stream stream) {
    ES_REQUIRE(sdf && tri_table && workspace && vert_offset && tri_offset && verts && faces && O > 0 && n >= 2,
               "es_marching_cubes_emit: bad args");
    const long npt = (long)n * n * n;
    const int nblk = (int)((npt + MC_BLOCK - 1) / MC_BLOCK);
    int* blk_v = (int*)workspace;
    int* blk_t = blk_v + (long)O * nblk;
    int* vbase = blk_t + (long)O * nblk;
    McGrid g{sdf, O, n, level};
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_mc_vertices, dim3(nblk, O), dim3(MC_BLOCK), 0, s, g, blk_v, vbase, (const long*)vert_offset, verts);
    hipLaunchKernelGGL(k_mc_faces, dim3(nblk, O), dim3(MC_BLOCK), 0, s, g, tri_table, blk_t, vbase, (const long*)tri_offset, faces);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}
