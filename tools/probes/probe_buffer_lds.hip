// Hardware probe: does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` write ZEROS to LDS (or skip the
// write)?  How does soffset enter the address / the range check?  (Design input for the conv kernel's halo handling.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* a, float* out, unsigned nrec, int soff, unsigned badoff) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s = (float*)smem;
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = 7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)a, (short)0, nrec, 0x00020000);
    unsigned voff = threadIdx.x * 16;
    if (threadIdx.x & 1) voff = badoff;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = s[i];
}
int main() {
    float *a, *o, h[4096], ho[256];
    for (int i = 0; i < 4096; ++i) h[i] = 100.f + i;
    hipMalloc(&a, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
    struct { unsigned nrec; int soff; unsigned bad; const char* what; } cases[] = {
        {0x80000000u, 0, 0x80000000u, "nrec 2^31, soff 0, bad=2^31"},
        {0x80000000u, 64, 0x80000000u, "nrec 2^31, soff 64, bad=2^31"},
        {4096u, 0, 4096u, "nrec 4096, soff 0, bad=4096"},
        {4096u, 2048, 3000u, "nrec 4096, soff 2048, voff 3000 (voff<nrec, voff+soff>nrec)"},
        {0x80000000u, 64, 0xFFFFFFF0u, "nrec 2^31, soff 64, bad=0xFFFFFFF0"},
    };
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, a, o, c.nrec, c.soff, c.bad);
        hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
        printf("%s:\n  lane0: %g %g %g %g | lane1: %g %g %g %g | lane2: %g %g | lane3: %g\n", c.what, ho[0], ho[1], ho[2], ho[3],
               ho[4], ho[5], ho[6], ho[7], ho[8], ho[9], ho[12]);
    }
    return 0;
}
