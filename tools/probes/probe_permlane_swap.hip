// semantics of v_permlane16_swap / v_permlane32_swap as used by es_pair16 / es_pair32 (csrc/es_common.h)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
    const unsigned x = threadIdx.x;
    const u2 a = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    const u2 b = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    out[threadIdx.x * 4 + 0] = a[0]; out[threadIdx.x * 4 + 1] = a[1];
    out[threadIdx.x * 4 + 2] = b[0]; out[threadIdx.x * 4 + 3] = b[1];
}
int main() {
    unsigned* d; unsigned h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) printf("lane %2d: swap16 -> (%2u, %2u)   swap32 -> (%2u, %2u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const unsigned e16 = (l & ~16u), o16 = (l | 16u), e32 = (l & ~32u), o32 = (l | 32u);
        if (h[l * 4] != e16 || h[l * 4 + 1] != o16 || h[l * 4 + 2] != e32 || h[l * 4 + 3] != o32) ++bad;
    }
    printf("lanes that do not hold (even-row value, odd-row value) of their pair: %d\n", bad);
    return bad != 0;
}
