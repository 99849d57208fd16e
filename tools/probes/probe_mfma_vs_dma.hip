// Hardware probe: does a wave that is issuing LDS-DMA (buffer_load ... lds) slow down the MFMA stream of ANOTHER wave
// on the same SIMD?  Block = 8 waves: waves 0-3 run a pure MFMA loop, waves 4-7 (their SIMD partners) either idle
// (mode 0), issue DMA pieces (mode 1: 16 rows x 64 B; mode 2: contiguous) or issue ds_read_b128 (mode 3).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(const char* base, int iters, int mode, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __syncthreads();
    if (wave < 4) {
        f4 acc[28];
        for (int j = 0; j < 28; ++j) acc[j] = f4{0, 0, 0, 0};
        h8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 28; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int j = 0; j < 28; ++j) s += acc[j][0];
        if (lane == 0) { out[blockIdx.x * 4 + wave] = t1 - t0; sink[blockIdx.x * 8 + wave] = s; }
    } else if (mode == 1 || mode == 2) {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, 0x7fffffff, 0x00020000);
        unsigned voff = mode == 1 ? (lane >> 2) * 448 + (lane & 3) * 16 : lane * 16;
        voff += (blockIdx.x & 255) * 65536 + wave * 8192;
        for (int i = 0; i < iters; ++i) {            // 4 pieces per 28 MFMAs of the partner: the conv kernel's ratio
#pragma unroll
            for (int p = 0; p < 4; ++p)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(smem + wave * 4096 + p * 1024), 16, voff, (i & 7) * 1024 + p * 7168, 0, 0);
            if ((i & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (mode == 3) {
        float s = 0;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int p = 0; p < 11; ++p) s += ((const f4*)(smem + ((lane * 16 + p * 1024 + i * 64) & 32767)))[0][0];
        }
        if (lane == 0) sink[blockIdx.x * 8 + wave] = s;
    }
}
int main() {
    char* buf; float* sink; unsigned long long* out; unsigned long long h[1024];
    hipMalloc(&buf, 64 << 20); hipMemset(buf, 1, 64 << 20); hipMalloc(&sink, 65536); hipMalloc(&out, 8192);
    const int iters = 2000;
    const char* names[] = {"partners idle", "partners issue DMA 16x64B", "partners issue DMA contiguous", "partners issue ds_read_b128 x11"};
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 4; ++m) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 32768, 0, buf, iters, m, out, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, out, 8192, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 1024; ++i) s += (double)h[i];
            if (rep) printf("%-36s MFMA wave: %.1f ticks per MFMA; kernel %.3f ms (MFMA-only ideal at 16 cyc @2.4GHz: %.3f ms)\n", names[m],
                            s / 1024 / iters / 28, ms, iters * 28 * 16 / 2.4e6);
        }
    return 0;
}
