// Hardware probe: 8 MFMA waves + barrier per 28 MFMAs, with 0 or 4 extra waves that only join the barrier.
// Prints MFMA rate and the SIMD each wave landed on (HW_ID bits [5:4]).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(NT) void k(int iters, unsigned long long* out, int* simd, float* sink) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if (lane == 0 && blockIdx.x == 0) simd[wave] = (hwid >> 4) & 3;
    if (wave < 8) {
        f4 acc[28];
        for (int j = 0; j < 28; ++j) acc[j] = f4{0, 0, 0, 0};
        h8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 1, 1, 1, 1, 1, 1, 1};
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int j = 0; j < 28; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0;
        for (int j = 0; j < 28; ++j) s += acc[j][0];
        if (lane == 0) { out[blockIdx.x * 8 + wave] = t1 - t0; sink[blockIdx.x * 16 + wave] = s; }
    } else {
        for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();
    }
}
int main() {
    float* sink; unsigned long long* out; int* simd; unsigned long long h[2048]; int hs[16];
    hipMalloc(&sink, 65536); hipMalloc(&out, 16384); hipMalloc(&simd, 64);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep)
        for (int m = 0; m < 2; ++m) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipMemset(simd, 0xff, 64);
            hipEventRecord(e0);
            if (m == 0) hipLaunchKernelGGL(k<512>, dim3(256), dim3(512), 0, 0, iters, out, simd, sink);
            else hipLaunchKernelGGL(k<768>, dim3(256), dim3(768), 0, 0, iters, out, simd, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h, out, 16384, hipMemcpyDeviceToHost); hipMemcpy(hs, simd, 64, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < 2048; ++i) s += (double)h[i];
            if (rep) {
                printf("%d waves per block: %.1f ticks per MFMA per wave (2 MFMA waves per SIMD -> ideal 32); kernel %.3f ms; SIMD of waves:", m ? 12 : 8, s / 2048 / iters / 28, ms);
                for (int w = 0; w < (m ? 12 : 8); ++w) printf(" %d", hs[w]);
                printf("\n");
            }
        }
    return 0;
}
