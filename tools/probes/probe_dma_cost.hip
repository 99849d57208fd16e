// Hardware probe: throughput of `buffer_load_dwordx4 ... lds` (1 KiB per wave-instruction) by access shape, data
// L2-resident: (a) 16 rows x 64 B (row stride 448 B)  (b) 8 rows x 128 B, 128-B aligned (stride 512)
// (c) 8 rows x 128 B at 64-B misalignment (stride 448)  (d) 1 KiB contiguous.   8 waves per CU, 256 CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
template <int MODE>
__global__ __launch_bounds__(512) void k(const char* base, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, 0x7fffffff, 0x00020000);
    unsigned voff;
    if (MODE == 0) voff = (lane >> 2) * 448 + (lane & 3) * 16;              // 16 rows x 64 B
    else if (MODE == 1) voff = (lane >> 3) * 512 + (lane & 7) * 16;         // 8 rows x 128 B aligned
    else if (MODE == 2) voff = (lane >> 3) * 448 + (lane & 7) * 16;         // 8 rows x 128 B, odd rows straddle
    else voff = lane * 16;                                                  // contiguous
    voff += (blockIdx.x & 255) * 65536 + wave * 8192;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(smem + wave * 4096 + p * 1024), 16, voff, (i & 7) * 1024 + p * 7168, 0, 0);
        if ((i & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = ((float*)smem)[lane];
}
int main() {
    char* buf; float* sink;
    hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20); hipMalloc(&sink, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    const char* names[] = {"16 rows x 64 B (stride 448)", "8 rows x 128 B aligned", "8 rows x 128 B misaligned", "1 KiB contiguous"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 4; ++m) {
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 32768, 0, buf, iters, sink);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 32768, 0, buf, iters, sink);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 32768, 0, buf, iters, sink);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(512), 32768, 0, buf, iters, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double pieces = 256.0 * 8 * iters * 4;
            if (rep) printf("%-32s %7.3f ms  %.1f ns per piece per CU-wave  %.1f GB/s per CU  (%.1f TB/s chip)\n", names[m], ms,
                            ms * 1e6 / (iters * 4), 8 * iters * 4 * 1024.0 / (ms * 1e6), pieces * 1024 / (ms * 1e9));
        }
    return 0;
}
