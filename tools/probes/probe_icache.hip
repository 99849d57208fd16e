// Hardware probe (round 5): what does a COLD instruction cache cost a short kernel?  Every launch of the layout step's ~110 dependent
// rows kernels starts with an invalidated I-cache, and all 256 CUs fetch the same code lines from L2 at the same moment.
// A workgroup runs a straight-line block of N 4-byte VALU instructions three times (loop): pass 0 fetches it (cold), passes 1-2 hit the
// I-cache.  Reported per grid size / workgroup size: s_memtime ticks of each pass (wave 0, median over workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define STR2(x) #x
#define STR(x) STR2(x)
template <int N>
__global__ __launch_bounds__(512) void k_code(unsigned long long* out) {
    unsigned long long t[4];
    int v = threadIdx.x;
    for (int it = 0; it < 3; ++it) {
        t[it] = __builtin_amdgcn_s_memtime();
        if (N == 256) asm volatile(".rept 256\n v_add_u32 %0, %0, 1\n .endr" : "+v"(v));
        if (N == 1024) asm volatile(".rept 1024\n v_add_u32 %0, %0, 1\n .endr" : "+v"(v));
        if (N == 4096) asm volatile(".rept 4096\n v_add_u32 %0, %0, 1\n .endr" : "+v"(v));
    }
    t[3] = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { for (int k = 0; k < 3; ++k) out[blockIdx.x * 4 + k] = t[k + 1] - t[k]; out[blockIdx.x * 4 + 3] = (unsigned long long)v; }
}
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
template <int N> void run(int nwg, int nthr, unsigned long long* out) {
    std::vector<unsigned long long> h(nwg * 4);
    std::vector<double> a, b, c;
    for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(k_code<N>, dim3(nwg), dim3(nthr), 0, 0, out);
        hipMemcpy(h.data(), out, nwg * 32, hipMemcpyDeviceToHost);
        if (!rep) continue;
        for (int i = 0; i < nwg; ++i) { a.push_back((double)h[i * 4]); b.push_back((double)h[i * 4 + 1]); c.push_back((double)h[i * 4 + 2]); }
    }
    printf("%5d instr (%5.1f KiB)  grid %4d x %3d thr:  cold %7.0f   warm %7.0f %7.0f ticks   -> cold - warm = %6.0f ticks = %.1f per 64-B line\n", N, N * 4 / 1024.0, nwg, nthr,
           med(a), med(b), med(c), med(a) - med(b), (med(a) - med(b)) / (N * 4 / 64.0));
}
int main() {
    unsigned long long* out; hipMalloc(&out, 4096 * 32);
    for (int nthr : {64, 512})
        for (int nwg : {1, 32, 256, 512}) { run<256>(nwg, nthr, out); run<1024>(nwg, nthr, out); run<4096>(nwg, nthr, out); }
    return 0;
}
