// What does ds_read_b64_tr_b16 return?  Every lane supplies the LDS address of 4 contiguous 16-bit elements; the probe fills LDS with
// element index values, gives lane l the address (l * 4 elements) and prints what each lane receives.
// build: hipcc --offload-arch=gfx950 -O2 probe_ds_read_tr.hip -o probe_ds_read_tr
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    typedef __attribute__((address_space(3))) s4* lp;
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(lds + l * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    return 0;
}
