// Chamfer nearest-neighbour distance: CDNA4 re-design of the reference's only native
// kernel pair (extension/old_chamfer/chamfer.cu:12-134 NmDistanceKernel, :155-174 NmDistanceGradKernel).
//
// The reference launches a fixed dim3(32,16) x 512 grid with a 512-point shared tile and a hand 4x unrolled scan.
// Here: one lane per query point, 256-thread workgroups (4 wave64), the other cloud streamed through LDS in tiles of
// 2048 points stored as float4 (x, y, z, 0) so that the inner loop is one conflict-free broadcast ds_read_b128 per
// candidate; grid = (ceil(n/256), batch) so any n fills the 256 CUs once n*batch >= 64k.  Same arithmetic as the
// reference (dx*dx + dy*dy + dz*dz, first minimum wins); the backward pass is the same atomicAdd scatter.
#include "es_common.h"

namespace {

constexpr int CH_TILE = 2048;

__global__ __launch_bounds__(256) void k_chamfer_nn(int n, const float* xyz, int m, const float* xyz2, float* result, int* result_i) {
    __shared__ f4 buf[CH_TILE];
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const float* q = xyz + ((long)b * n + (j < n ? j : 0)) * 3;
    const float x1 = q[0], y1 = q[1], z1 = q[2];
    float best = 0.f;
    int best_i = 0;
    for (int k2 = 0; k2 < m; k2 += CH_TILE) {
        const int cnt = min(CH_TILE, m - k2);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += 256) {
            const float* p = xyz2 + ((long)b * m + k2 + t) * 3;
            buf[t] = f4{p[0], p[1], p[2], 0.f};
        }
        __syncthreads();
        for (int k = 0; k < cnt; ++k) {
            const f4 p = buf[k];
            const float dx = p[0] - x1, dy = p[1] - y1, dz = p[2] - z1;
            const float d = dx * dx + dy * dy + dz * dz;
            if ((k2 + k) == 0 || d < best) { best = d; best_i = k2 + k; }
        }
    }
    if (j < n) {
        result[(long)b * n + j] = best;
        result_i[(long)b * n + j] = best_i;
    }
}

__global__ __launch_bounds__(256) void k_chamfer_grad(int n, const float* xyz1, int m, const float* xyz2, const float* grad_dist1,
                                                     const int* idx1, float* grad_xyz1, float* grad_xyz2) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float* p1 = xyz1 + ((long)b * n + j) * 3;
    const int j2 = idx1[(long)b * n + j];
    const float* p2 = xyz2 + ((long)b * m + j2) * 3;
    const float g = grad_dist1[(long)b * n + j] * 2;
    float* g1 = grad_xyz1 + ((long)b * n + j) * 3;
    float* g2 = grad_xyz2 + ((long)b * m + j2) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = g * (p1[c] - p2[c]);
        atomicAdd(&g1[c], v);
        atomicAdd(&g2[c], -v);
    }
}

}  // namespace

extern "C" int es_chamfer_forward(const float* xyz1, const float* xyz2, int batch, int n, int m, float* dist1, int32_t* idx1,
                                  float* dist2, int32_t* idx2, es_stream stream) {
    ES_REQUIRE(batch > 0 && n > 0 && m > 0, "es_chamfer_forward: empty clouds (batch=%d n=%d m=%d)", batch, n, m);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_chamfer_nn, dim3((n + 255) / 256, batch), dim3(256), 0, s, n, xyz1, m, xyz2, dist1, idx1);
    hipLaunchKernelGGL(k_chamfer_nn, dim3((m + 255) / 256, batch), dim3(256), 0, s, m, xyz2, n, xyz1, dist2, idx2);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_chamfer_backward(const float* xyz1, const float* xyz2, int batch, int n, int m, const float* graddist1,
                                   const float* graddist2, const int32_t* idx1, const int32_t* idx2, float* gradxyz1,
                                   float* gradxyz2, es_stream stream) {
    ES_REQUIRE(batch > 0 && n > 0 && m > 0, "es_chamfer_backward: empty clouds");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_chamfer_grad, dim3((n + 255) / 256, batch), dim3(256), 0, s, n, xyz1, m, xyz2, graddist1, idx1, gradxyz1, gradxyz2);
    hipLaunchKernelGGL(k_chamfer_grad, dim3((m + 255) / 256, batch), dim3(256), 0, s, m, xyz2, n, xyz1, graddist2, idx2, gradxyz2, gradxyz1);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}
