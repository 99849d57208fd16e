// Shared helpers for the echoscene HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include "../../include/echoscene_hip.h"

typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

void es_set_error(const char* fmt, ...);

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

// x * sigmoid(x) with the hardware exp2 / rcp (relative error ~1e-6, far inside the 1e-4 parity budget)
__device__ __forceinline__ float es_silu(float x) { return x * __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float es_silu_fast(float x) { return x / (1.0f + __expf(-x)); } // volume path (fp16 operands follow)
// exact (erf) GELU, as torch.nn.functional.gelu default
__device__ __forceinline__ float es_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
