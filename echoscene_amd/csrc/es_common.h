// Shared helpers for the echoscene HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/echoscene_hip.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

void es_set_error(const char* fmt, ...);
// plan executor -> rows launcher: the single-problem rows product the plan runs NEXT (NULL: unknown / not one); consumed by the next
// rows launch of this thread, whose extra wave prefetches that product's weights (es_rows_x.h)
struct es_linear_args;
void es_rows_hint_next(const struct es_linear_args* next);

// Host-side parallel loop for the one-off weight re-layouts (es_pack_*): 430 M shape-UNet weights through a scalar loop were 9 of the
// 9.2 s of a process's first scene call.  f(i) for i in [0, n), strided over up to 32 threads; serial when threads cannot be had.
template <class F>
inline void es_parallel_for(long n, F f) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 32) nt = 32;
    if ((long)nt > n) nt = (unsigned)n;
    if (nt <= 1) { for (long i = 0; i < n; ++i) f(i); return; }
    std::vector<std::thread> th;
    unsigned started = 0;
    try {
        for (; started < nt; ++started) th.emplace_back([=] { for (long i = started; i < n; i += nt) f(i); });
    } catch (...) {                              // (could not start every thread: the missing residues run here)
        for (unsigned t = started; t < nt; ++t) for (long i = t; i < n; i += nt) f(i);
    }
    for (auto& t : th) t.join();
}

#define ES_CHECK_HIP(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            es_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)

#define ES_REQUIRE(cond, ...)                                                                \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            es_set_error(__VA_ARGS__);                                                       \
            return 2;                                                                        \
        }                                                                                    \
    } while (0)

// x combined with the value of lane ^ 16 / lane ^ 32 without the LDS crossbar: v_permlane16_swap / v_permlane32_swap (gfx950) exchange
// the odd 16- / 32-lane rows of one operand with the even rows of the other, so both results hold, in EVERY lane, the even-row and the
// odd-row value of its pair (two VALU instructions instead of a ds_bpermute round trip in a dependent chain).
typedef unsigned es_u2 __attribute__((ext_vector_type(2)));
// (both results pass through an empty asm: hipcc 7.2 folds `f(r[0], r[1])` of this intrinsic to `f(r[0], r[0])` at -O3 -- the IR keeps
//  only extractvalue 0 and the reduction silently loses the odd rows; tools/probes/probe_permlane_swap.hip holds the semantics)
__device__ __forceinline__ void es_pair16(float x, float& even, float& odd) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const es_u2 r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    unsigned e = r[0], o = r[1];
    asm volatile("" : "+v"(e), "+v"(o));
    even = __builtin_bit_cast(float, e); odd = __builtin_bit_cast(float, o);
}
__device__ __forceinline__ void es_pair32(float x, float& even, float& odd) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const es_u2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    unsigned e = r[0], o = r[1];
    asm volatile("" : "+v"(e), "+v"(o));
    even = __builtin_bit_cast(float, e); odd = __builtin_bit_cast(float, o);
}

// x * sigmoid(x) with the hardware exp2 / rcp (relative error ~1e-6, far inside the 1e-4 parity budget)
// (v_rcp_f32, 1 ulp: `__frcp_rn` and `/` are the correctly rounded quotient -- v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup,
//  ten instructions per evaluation in the staging prologue of every GroupNorm + SiLU launch of the latency-bound layout chain)
__device__ __forceinline__ float es_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float es_silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); } // volume path (fp16 operands follow)
// exact (erf) GELU, as torch.nn.functional.gelu default
__device__ __forceinline__ float es_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU for the volume path (the result is rounded to fp16 right after): erfc(|x|/sqrt 2) by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7), evaluated on the side where it does not cancel: gelu(x) = max(x, 0) - (|x| / 2) E(|x| / sqrt 2).  Max abs error
// 2.2e-7 over [-12, 12] (tests/test_hip_vol.py); ~12 VALU ops instead of erff's ~40 with divergent branches -- the exact
// version cost ~9 us of the ~27 us a 256 x 224 FeedForward tile took (56 evaluations per lane).
__device__ __forceinline__ float es_gelu_fast(float x) {
    // gelu(x) = max(x, 0) - (|x| / 2) E(|x| / sqrt 2): one expression for both signs (E = erfc of the A&S form, never cancelling)
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f, z, 1.0f));     // v_rcp_f32 (1 ulp), see es_gelu_fast2
    float pl = __fmaf_rn(t, 1.061405429f, -1.453152027f);
    pl = __fmaf_rn(t, pl, 1.421413741f);
    pl = __fmaf_rn(t, pl, -0.284496736f);
    pl = __fmaf_rn(t, pl, 0.254829592f);
    const float E = pl * t * __builtin_amdgcn_exp2f(ax * ax * -0.72134752044448170368f);    // exp(-z^2) = 2^(-x^2 log2(e) / 2)
    return __fmaf_rn(-0.5f * ax, E, fmaxf(x, 0.f));
}

// Two evaluations of es_gelu_fast at once on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32: one issue slot for two lanes' worth of
// the polynomial; the reciprocal and the exponential stay scalar-width).  The same operations in the same order as es_gelu_fast (equal up
// to where the compiler contracts a multiply-add); every route of the fused GEGLU epilogue goes through THIS function, so results do
// not depend on the route.  Accuracy bound as es_gelu_fast (tests/test_hip_vol.py::test_gelu_of_the_volume_path_over_its_whole_range).
typedef float es_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ es_f2 es_gelu_fast2(es_f2 x) {
    const es_f2 ax = {fabsf(x[0]), fabsf(x[1])};
    const es_f2 z = ax * 0.70710678118654752440f;
    const es_f2 one = {1.0f, 1.0f};
    const es_f2 d = __builtin_elementwise_fma(es_f2{0.3275911f, 0.3275911f}, z, one);
    // v_rcp_f32 (1 ulp; d is in [1, 3.8]): `__frcp_rn` is the correctly rounded quotient -- v_div_scale x2, v_rcp, five fma, v_div_fmas,
    // v_div_fixup, ten instructions per evaluation and 560 of the 1790 VALU instructions of a GEGLU tile epilogue (round 4, from the ISA)
    const es_f2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    es_f2 pl = __builtin_elementwise_fma(t, es_f2{1.061405429f, 1.061405429f}, es_f2{-1.453152027f, -1.453152027f});
    pl = __builtin_elementwise_fma(t, pl, es_f2{1.421413741f, 1.421413741f});
    pl = __builtin_elementwise_fma(t, pl, es_f2{-0.284496736f, -0.284496736f});
    pl = __builtin_elementwise_fma(t, pl, es_f2{0.254829592f, 0.254829592f});
    const es_f2 w = ax * ax * -0.72134752044448170368f;                 // exp(-z^2) = 2^(-x^2 log2(e) / 2): no separate log2(e) multiply
    const es_f2 ex = {__builtin_amdgcn_exp2f(w[0]), __builtin_amdgcn_exp2f(w[1])};
    const es_f2 E = pl * t * ex;
    // max(x, 0) - (|x| / 2) E: one expression for both signs instead of compare / subtract / select
    return __builtin_elementwise_fma(ax * -0.5f, E, es_f2{fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)});
}

// Kernel arguments are read with scalar loads as the code reaches each field: several s_load -> s_waitcnt -> branch steps in a row
// at kernel entry (a dozen for the 1.4 KB block of the rows kernels), each a scalar-cache miss on a fresh launch.  Touch every 64-byte line of the kernarg segment at once
// (one round trip), so that the loads the compiler emits later hit the scalar cache.
template <int BYTES>
__device__ __forceinline__ void kernarg_warm() {
    const unsigned long ka = (unsigned long)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int NL = (BYTES + 63) / 64;
    static_assert(NL <= 28, "kernarg_warm: argument block larger than expected");
    // ONE asm statement, loads AND wait: the loads are asynchronous, and as separate statements (round 3-4) the compiler was free to put a
    // load of its own between them into the very SGPR the statements used as their dead destination -- two loads in flight to one
    // register return in any order (round 5: a kernel read `ny` of its argument block as whatever line 0x180 held; found in the ISA).
    unsigned d;
    asm volatile(".set es_kw_off, 0\n\t.rept %c2\n\ts_load_dword %0, %1, es_kw_off\n\t.set es_kw_off, es_kw_off + 64\n\t.endr\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(d) : "s"(ka), "i"(NL) : "memory");
}

