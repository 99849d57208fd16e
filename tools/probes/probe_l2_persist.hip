// Hardware probe (round 5): does a line that kernel A pulled into an XCD's L2 survive the kernel boundary for kernel B?
// The layout step is a chain of ~110 dependent launches whose weight loads (1-8 MB per launch, read once per step) are HBM misses on the
// critical path of every launch; if L2 contents survive a boundary, launch i can fetch launch i+1's weight slices into the L2 of the XCD
// that will read them (workgroup b runs on XCD b % 8 in both launches).
// Measured per workgroup (wave 0, s_memrealtime = 100 MHz ticks and s_memtime): issue -> data of one 1 KiB wave-level load, and the
// first kernarg scalar load, for: cold (after a 1 GB flush), same-XCD touch by the previous kernel, other-XCD touch (Infinity Cache
// only), same-XCD touch with a writing kernel in between.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

struct BigArgs { const f4* w; long chunk_f4; unsigned long long* out; int pad[300]; };   // ~1.2 KB of kernarg like the rows kernels

__global__ __launch_bounds__(512) void k_flush(const f4* buf, long n, float* sink) {
    f4 a = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a += buf[i];
    if (a[0] + a[1] + a[2] + a[3] == 12345.f) sink[0] = a[0];
}

// workgroup b reads the chunk of workgroup (b + shift) % gridDim.x: shift 0 = the XCD that reads it next, shift 1 = a neighbour XCD
__global__ __launch_bounds__(512) void k_touch(const f4* w, long chunk_f4, int shift, int per, float* sink) {
    const int c = ((int)blockIdx.x + shift) % (int)gridDim.x;
    const f4* p = w + (long)c * chunk_f4;
    f4 a = {0, 0, 0, 0};
    for (int j = 0; j < per; ++j) a += p[j * 512 + threadIdx.x];
    if (a[0] + a[1] + a[2] + a[3] == 12345.f) sink[0] = a[0];
}

__global__ void k_write(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] = 1.0f; }

__global__ __launch_bounds__(512) void k_read(const BigArgs A) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const f4* w = A.w;                                           // first kernarg use: a scalar load
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long chunk = A.chunk_f4;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    const f4* p = w + (long)blockIdx.x * chunk + threadIdx.x;
    f4 v = *p;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory");
    const unsigned long long c2 = __builtin_amdgcn_s_memtime();
    f4 v2 = p[512];                                              // the next 8 KiB of the chunk: a second, dependent round trip
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v2) :: "memory");
    const unsigned long long c3 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        unsigned long long* o = A.out + (long)blockIdx.x * 8;
        o[0] = c1 - c0; o[1] = c2 - c1; o[2] = c3 - c2; o[3] = r1 - r0; o[4] = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        o[5] = (unsigned long long)(v[0] + v2[0] == 12345.f);
    }
}

__global__ void k_calib(unsigned long long* o) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < 100000) r1 = __builtin_amdgcn_s_memrealtime();            // 1 ms
    o[0] = r1 - r0; o[1] = __builtin_amdgcn_s_memtime() - c0;
}

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int NWG = 256;
    const long chunk_f4 = 1024;                 // 16 KiB per workgroup: 2 x (512 lanes x 16 B)
    const long region_f4 = NWG * chunk_f4;      // 4 MiB per repetition
    const int REPS = 12;
    f4 *w, *big; float* sink; unsigned long long* out;
    const long big_n = (1l << 30) / 16;
    hipMalloc(&w, REPS * region_f4 * 16 * 8); hipMemset(w, 0, REPS * region_f4 * 16 * 8);
    hipMalloc(&big, big_n * 16); hipMemset(big, 0, big_n * 16);
    hipMalloc(&sink, 1 << 20); hipMalloc(&out, NWG * 8 * 8);
    std::vector<unsigned long long> h(NWG * 8);
    const char* names[] = {"cold (after a 1 GiB flush)", "touched by the SAME XCD's workgroup in the previous kernel",
                           "touched by ANOTHER XCD in the previous kernel (Infinity Cache)", "same XCD, a small writing kernel in between",
                           "same XCD, touched TWO kernels earlier (a 256-WG touch of other data in between)"};
    hipLaunchKernelGGL(k_calib, dim3(1), dim3(1), 0, 0, out);
    hipMemcpy(h.data(), out, 16, hipMemcpyDeviceToHost);
    printf("s_memtime: %.3f ticks per 10 ns (s_memrealtime tick)\n", (double)h[1] / (double)h[0]);
    // ticks of s_memtime per 100 MHz tick are reported as raw numbers; 100 MHz tick = 10 ns
    for (int sc = 0; sc < 5; ++sc) {
        std::vector<double> karg, l1, l2, tot;
        int xcd_ok = 0;
        for (int rep = 0; rep < REPS; ++rep) {
            const f4* reg = w + (long)(sc * REPS + rep) * region_f4 % (REPS * region_f4 * 8);
            hipLaunchKernelGGL(k_flush, dim3(2048), dim3(512), 0, 0, big, big_n, sink);
            if (sc == 1 || sc == 3 || sc == 4) hipLaunchKernelGGL(k_touch, dim3(NWG), dim3(512), 0, 0, reg, chunk_f4, 0, 2, sink);
            if (sc == 2) hipLaunchKernelGGL(k_touch, dim3(NWG), dim3(512), 0, 0, reg, chunk_f4, 1, 2, sink);
            if (sc == 3) hipLaunchKernelGGL(k_write, dim3(64), dim3(256), 0, 0, sink + 4096);
            if (sc == 4) hipLaunchKernelGGL(k_touch, dim3(NWG), dim3(512), 0, 0, (const f4*)big, chunk_f4, 0, 2, sink);
            BigArgs A; A.w = reg; A.chunk_f4 = chunk_f4; A.out = out;
            hipLaunchKernelGGL(k_read, dim3(NWG), dim3(512), 0, 0, A);
            hipMemcpy(h.data(), out, NWG * 64, hipMemcpyDeviceToHost);
            for (int b = 0; b < NWG; ++b) {
                karg.push_back((double)h[b * 8 + 0]); l1.push_back((double)h[b * 8 + 1]); l2.push_back((double)h[b * 8 + 2]);
                tot.push_back((double)h[b * 8 + 3]);
                xcd_ok += (int)h[b * 8 + 4] == b % 8;
            }
        }
        printf("%-82s kernarg %6.0f  load#1 %6.0f  load#2 %6.0f  (s_memtime ticks, median over %d WGs x %d)  whole %.2f us  [block b on XCD b%%8: %d/%d]\n",
               names[sc], med(karg), med(l1), med(l2), NWG, REPS, med(tot) / 100.0, xcd_ok, NWG * REPS);
    }
    return 0;
}
