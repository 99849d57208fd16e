// "volume" path: the 3-D latent-SDF UNet (reference openai_model_3d.py:816-863).
//
// Roofline: 557.8 GFLOP per object per DDIM step, 87 % of it 3x3x3 Conv3d
// -> MFMA-bound.  Layout: channels-last [O][D][H][W][C] so that an implicit-GEMM conv reads
// K-contiguous rows: out[m][n] = sum_{tap} sum_c A[shift(m,tap)][c] * W[n][tap][c].
//   * residual stream fp32; every contraction takes fp16 operands and accumulates fp32 on
//     v_mfma_f32_16x16x32_f16;
//   * A/B tiles go HBM -> LDS with global_load_lds (16 B/lane, no VGPR round trip); zero padding
//     of the 3x3x3 halo is done by pointing out-of-volume rows at a zero page;
//   * LDS tiles are [rows][32 halfs] with a 16-B-chunk XOR swizzle applied on the (per-lane)
//     SOURCE address and on the fragment read (glds destinations are lane-linear);
//   * GroupNorm/SiLU, LayerNorm and GEGLU are bandwidth-trivial side kernels that emit the fp16
//     operand of the next contraction; bias / time-embedding / residual adds are fused into the
//     contraction epilogue.
#include "es_common.h"
#include <string>
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <type_traits>

namespace {

__device__ __forceinline__ int f_swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

// ---------------------------------------------------------------------------------------------
// GroupNorm over channels-last volumes.  Pass 1: per (object, 64-voxel tile) partial sums.
// ---------------------------------------------------------------------------------------------
constexpr int GN_VT = 64;

// 4 channels per thread (16-B loads), 64 voxels per workgroup, then channel -> group reduction in LDS.
__global__ __launch_bounds__(256) void k_gn_partial(const es_gn_args a, float* part, int vt) {
    // part: [O][ntiles][groups][2]
    __shared__ float ssum4[4][2048], ssq4[4][2048];       // per voxel-row lane partials (fixed-order combine: deterministic)
    float* ssum = ssum4[0];
    float* ssq = ssq4[0];
    const int o = blockIdx.y, tile = blockIdx.x, C = a.C1 + a.C2;
    const int v0 = tile * vt;
    const int nv = min(vt, a.V - v0);
    const int c4n = C >> 2;
    const int CX = 64;                                   // 64 lanes span 256 channels per pass
    const int cx = threadIdx.x & (CX - 1), vy = threadIdx.x >> 6;        // 4 voxel rows in flight
    for (int c4 = cx; c4 < c4n; c4 += CX) {
        const int c = c4 * 4;
        const float* src; int ld, cc;
        if (c < a.C1) { src = a.x1; ld = a.C1; cc = c; } else { src = a.x2; ld = a.C2; cc = c - a.C1; }
        const float* p = src + ((long)o * a.V + v0) * ld + cc;
        f4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
        if (nv == vt && (vt & 15) == 0) {        // full tile: 4 independent loads in flight per lane
            for (int v = vy; v < vt; v += 16) {
                const f4 x0 = *(const f4*)(p + (long)v * ld), x1 = *(const f4*)(p + (long)(v + 4) * ld),
                         x2 = *(const f4*)(p + (long)(v + 8) * ld), x3 = *(const f4*)(p + (long)(v + 12) * ld);
                s += x0; q += x0 * x0; s += x1; q += x1 * x1; s += x2; q += x2 * x2; s += x3; q += x3 * x3;
            }
        } else {
            for (int v = vy; v < nv; v += 4) { const f4 x = *(const f4*)(p + (long)v * ld); s += x; q += x * x; }
        }
        *(f4*)&ssum4[vy][c] = s;
        *(f4*)&ssq4[vy][c] = q;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        ssum[c] = (ssum4[0][c] + ssum4[1][c]) + (ssum4[2][c] + ssum4[3][c]);
        ssq[c] = (ssq4[0][c] + ssq4[1][c]) + (ssq4[2][c] + ssq4[3][c]);
    }
    __syncthreads();
    const int gs = C / a.groups;
    if (threadIdx.x < a.groups) {
        float s = 0.f, q = 0.f;
        for (int k = 0; k < gs; ++k) { s += ssum[threadIdx.x * gs + k]; q += ssq[threadIdx.x * gs + k]; }
        float* dst = part + (((long)o * gridDim.x + tile) * a.groups + threadIdx.x) * 2;
        dst[0] = s; dst[1] = q;
    }
}

// Statistics of one object from the per-tile partials: 256/groups slices of the tile list per group, combined in fixed order
// (deterministic), in double.  Called by every k_gn_apply block when the tile list is short (the UNet: <= 64 tiles per object), or
// once per object by k_gn_finalize when it is long (the VQ-VAE decoder at 32^3 / 64^3: 512 / 4096 tiles -- re-reducing 1 MB of
// partials in each of 65536 apply blocks made k_gn_apply 48 % of the decode, 1.28 ms per call).
__device__ __forceinline__ void gn_reduce_stats(const es_gn_args& a, const float* part, int ntiles, int o, float* smean, float* srstd) {
    __shared__ double ds[256], dq[256];
    const int C = a.C1 + a.C2, gs = C / a.groups;
    const int gi = threadIdx.x % a.groups, sl = threadIdx.x / a.groups, nsl = 256 / a.groups;
    double s = 0.0, q = 0.0;
    if (sl < nsl) {
        // sixteen partials in flight per thread, added in tile order (the bits of the plain loop: round 5 -- with one dependent L2 round
        // trip per tile the 128-tile lists of a few-objects launch made this prologue most of k_gn_apply's 10 us; four in flight: 8.3 us)
        typedef float f2g __attribute__((ext_vector_type(2)));
        const f2g* pp = (const f2g*)part + ((long)o * ntiles * a.groups + gi);
        const long st = (long)nsl * a.groups;
        int t = sl;
        for (; t + 15 * nsl < ntiles; t += 16 * nsl) {
            f2g p[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) p[k] = pp[(long)t * a.groups + k * st];
#pragma unroll
            for (int k = 0; k < 16; ++k) { s += p[k][0]; q += p[k][1]; }
        }
        for (; t + 3 * nsl < ntiles; t += 4 * nsl) {
            const f2g p0 = pp[(long)t * a.groups], p1 = pp[(long)t * a.groups + st], p2 = pp[(long)t * a.groups + 2 * st], p3 = pp[(long)t * a.groups + 3 * st];
            s += p0[0]; q += p0[1]; s += p1[0]; q += p1[1]; s += p2[0]; q += p2[1]; s += p3[0]; q += p3[1];
        }
        for (; t < ntiles; t += nsl) { const f2g p0 = pp[(long)t * a.groups]; s += p0[0]; q += p0[1]; }
    }
    ds[threadIdx.x] = s; dq[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < a.groups) {
        s = 0.0; q = 0.0;
        for (int k = 0; k < nsl; ++k) { s += ds[threadIdx.x + k * a.groups]; q += dq[threadIdx.x + k * a.groups]; }
        const double n = (double)gs * a.V;
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0.0) var = 0.0;
        smean[threadIdx.x] = (float)mean;
        srstd[threadIdx.x] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
}

// one block per (object, group): 256 threads stride the tile list, fixed-order tree in LDS (deterministic), double accumulation
__global__ __launch_bounds__(256) void k_gn_finalize(const es_gn_args a, const float* part, int ntiles, float* fin) {
    __shared__ double ds[256], dq[256];
    const int o = blockIdx.y, g = blockIdx.x;
    double s = 0.0, q = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += 256) {
        const float* p = part + (((long)o * ntiles + t) * a.groups + g) * 2;
        s += p[0]; q += p[1];
    }
    ds[threadIdx.x] = s; dq[threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) { ds[threadIdx.x] += ds[threadIdx.x + w]; dq[threadIdx.x] += dq[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int gs = (a.C1 + a.C2) / a.groups;
        const double n = (double)gs * a.V;
        const double mean = ds[0] / n;
        double var = dq[0] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        fin[((long)o * a.groups + g) * 2] = (float)mean;
        fin[((long)o * a.groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
}

// Pass 2: statistics (above, or read from `fin`), folded with the affine into per-channel scale/shift in LDS,
// then y = x*scale + shift (+SiLU/GELU) -> fp16.  Thread (tx, ty): 8 channels at 8*(tx + TX k), voxels ty + TY k, with TX = the
// number of 8-channel groups rounded up to a power of two (<= 32) -- 64- and 128-channel tensors (VQ-VAE decoder) used a quarter / a
// half of the lanes with TX fixed at 32.  No integer divisions in the streaming loop, 32-B reads / 16-B writes per thread.
__global__ __launch_bounds__(256) void k_gn_apply(const es_gn_args a, const float* part, int ntiles, int vox_per_block, const float* fin) {
    __shared__ float smean[64], srstd[64];
    __shared__ float ssc[2048], ssh[2048];
    const int o = blockIdx.y, C = a.C1 + a.C2, gs = C / a.groups;
    if (fin) {
        if (threadIdx.x < a.groups) {
            smean[threadIdx.x] = fin[((long)o * a.groups + threadIdx.x) * 2];
            srstd[threadIdx.x] = fin[((long)o * a.groups + threadIdx.x) * 2 + 1];
        }
    } else {
        gn_reduce_stats(a, part, ntiles, o, smean, srstd);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int g = c / gs;
        const float sc = srstd[g] * a.gamma[c];
        ssc[c] = sc;
        ssh[c] = a.beta[c] - smean[g] * sc;
    }
    __syncthreads();
    const int c8n = C >> 3;
    int lt = 5;                                          // log2(TX)
    while (lt > 0 && (1 << (lt - 1)) >= c8n) --lt;
    const int TX = 1 << lt, TY = 256 >> lt;
    const int tx = threadIdx.x & (TX - 1), ty = threadIdx.x >> lt;
    const int v0 = blockIdx.x * vox_per_block;
    for (int c8 = tx; c8 < c8n; c8 += TX) {
        const int c = c8 * 8;
        const float* src; int ld, cc;
        if (c < a.C1) { src = a.x1; ld = a.C1; cc = c; } else { src = a.x2; ld = a.C2; cc = c - a.C1; }
        const f4 sc0 = *(const f4*)&ssc[c], sc1 = *(const f4*)&ssc[c + 4], sh0 = *(const f4*)&ssh[c], sh1 = *(const f4*)&ssh[c + 4];
        for (int vl = ty; vl < vox_per_block; vl += TY) {
            const int v = v0 + vl;
            if (v >= a.V) break;
            f4 x0, x1;
            if (a.x1_is_f16) {                       // (wave-uniform) the source is the producing conv's f16-only output
                const h8 hx = *(const h8*)((const _Float16*)(const void*)src + ((long)o * a.V + v) * ld + cc);
                x0 = f4{(float)hx[0], (float)hx[1], (float)hx[2], (float)hx[3]};
                x1 = f4{(float)hx[4], (float)hx[5], (float)hx[6], (float)hx[7]};
            } else {
                const float* p = src + ((long)o * a.V + v) * ld + cc;
                x0 = *(const f4*)p; x1 = *(const f4*)(p + 4);
            }
            f4 y0 = x0 * sc0 + sh0, y1 = x1 * sc1 + sh1;
            h8 y, r;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t0 = y0[e], t1 = y1[e];
                if (a.silu == 1) { t0 = es_silu_fast(t0); t1 = es_silu_fast(t1); }
                else if (a.silu == 2) { t0 = es_gelu_fast(t0); t1 = es_gelu_fast(t1); }
                y[e] = (_Float16)t0; y[4 + e] = (_Float16)t1;
                r[e] = (_Float16)x0[e]; r[4 + e] = (_Float16)x1[e];
            }
            const long off = ((long)o * a.V + v) * C + c;
            if (a.y_is_f32) {                        // (wave-uniform) fp32-operand validation route: the operand stays fp32
                f4 z0, z1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t0 = y0[e], t1 = y1[e];
                    if (a.silu == 1) { t0 = es_silu(t0); t1 = es_silu(t1); }
                    else if (a.silu == 2) { t0 = es_gelu(t0); t1 = es_gelu(t1); }
                    z0[e] = t0; z1[e] = t1;
                }
                *(f4*)((float*)a.y_f16 + off) = z0; *(f4*)((float*)a.y_f16 + off + 4) = z1;
                if (a.raw_f16) { *(f4*)((float*)a.raw_f16 + off) = x0; *(f4*)((float*)a.raw_f16 + off + 4) = x1; }
                continue;
            }
            *(h8*)((_Float16*)a.y_f16 + off) = y;
            if (a.raw_f16) *(h8*)((_Float16*)a.raw_f16 + off) = r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over tokens: fp32 in, fp16 out.  A wave owns rows wave, wave + W, wave + 2 W, ... (W = waves of the grid): 16-byte
// loads (lane l: columns 4 (l + 64 i) ..+3), the NEXT row's loads are in flight while the current one is reduced and stored, the
// affine vectors sit in registers for the whole launch, 8-byte stores.  Round 5: the one-row-per-wave version (8192 workgroups of four
// short-lived waves, 4-byte loads, 2-byte stores) reached 4.1-4.5 TB/s on the two shapes of the UNet (32768 x 448, 8192 x 672).
// Statistics as before: mean, then the centred sum of squares, from the registers (one pass over memory).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_layernorm(const es_ln_args a) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int C = a.C, c4n = C >> 2;                 // (C % 4 == 0: host-checked; C <= 1024 -> at most 4 chunks per lane)
    f4 ga[4], be[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c4 = lane + 64 * i;
        ga[i] = c4 < c4n ? *(const f4*)(a.gamma + 4 * c4) : f4{0.f, 0.f, 0.f, 0.f};
        be[i] = c4 < c4n ? *(const f4*)(a.beta + 4 * c4) : f4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_c = 1.0f / (float)C;
    f4 nx[4];
    auto load = [&](int row) __attribute__((always_inline)) {
        const float* p = a.x + (long)row * C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = lane + 64 * i;
            nx[i] = (row < a.M && c4 < c4n) ? *(const f4*)(p + 4 * c4) : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (wid < a.M) load(wid);
    for (int row = wid; row < a.M; row += nw) {
        f4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = nx[i];
        load(row + nw);                              // (past the end: zeros, nothing is read)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * inv_c;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (lane + 64 * i < c4n) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; q += d * d; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q * inv_c + a.eps);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c4 = lane + 64 * i;
            if (c4 < c4n) {
                f4 t;
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = (v[i][e] - mean) * rstd * ga[i][e] + be[i][e];
                if (a.y_is_f32) *(f4*)((float*)a.y_f16 + (long)row * C + 4 * c4) = t;          // the fp32-operand validation route
                else *(h4*)((_Float16*)a.y_f16 + (long)row * C + 4 * c4) = h4{(_Float16)t[0], (_Float16)t[1], (_Float16)t[2], (_Float16)t[3]};
            }
        }
    }
}

// GEGLU: h fp32 [M, 2*C4] (value | gate) -> fp16 [M, C4]
__global__ __launch_bounds__(256) void k_geglu(const es_geglu_args a) {
    const long n4 = (long)a.M * (a.C4 >> 2);
    const int c4n = a.C4 >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long m = i / c4n;
        const int c = (int)(i - m * c4n) * 4;
        const float* p = a.h_f32 + m * 2 * a.C4 + c;
        const f4 x = *(const f4*)p, g = *(const f4*)(p + a.C4);
        if (a.out_is_f32) {                          // fp32-operand validation route: exact erf GELU, fp32 result
            f4 z;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = x[e] * es_gelu(g[e]);
            *(f4*)((float*)a.out_f16 + m * a.C4 + c) = z;
            continue;
        }
        const es_f2 g01 = es_gelu_fast2(es_f2{g[0], g[1]}), g23 = es_gelu_fast2(es_f2{g[2], g[3]});     // (the fused epilogue's GELU)
        const h4 y = {(_Float16)(x[0] * g01[0]), (_Float16)(x[1] * g01[1]), (_Float16)(x[2] * g23[0]), (_Float16)(x[3] * g23[1])};
        *(h4*)((_Float16*)a.out_f16 + m * a.C4 + c) = y;
    }
}

// NCDHW fp32 [O,C,V] -> channels-last f16 [O,V,Cpad] (zero padded channels)
__global__ __launch_bounds__(256) void k_to_cl(const float* x, int O, int C, int V, int Cpad, _Float16* out, int is_f32) {
    const long n = (long)O * V * Cpad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % Cpad);
        const long ov = i / Cpad;
        const long o = ov / V, v = ov - o * V;
        const float t = c < C ? x[(o * C + c) * V + v] : 0.f;
        if (is_f32) ((float*)(void*)out)[i] = t; else out[i] = (_Float16)t;
    }
}

// Split-operand image of an fp32 activation (round 6, precision 'fp32x'): x = hi + lo with hi = f16(x), lo = f16(x - hi) -- 22 bits of
// the mantissa in two f16 values.  out[m] = [hi(0..C) | lo(0..C) | hi(0..C)] (3C channels): against the weight image [w_hi | w_hi | w_lo]
// the ordinary f16 contraction accumulates hi w_hi + lo w_hi + hi w_lo in fp32 -- the product to ~2^-21 relative (the lo x lo term
// is below that), at 3x the K of the f16 route instead of the 1/16 matrix rate of the fp32 instruction.
__global__ __launch_bounds__(256) void k_split_f16x3(const float* __restrict__ x, long M, int C, _Float16* __restrict__ out) {
    const int c4n = C >> 2;
    const long n = M * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long m = i / c4n;
        const int c = (int)(i - m * c4n) << 2;
        const f4 v = *(const f4*)(x + m * C + c);
        h4 hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = (_Float16)v[e]; lo[e] = (_Float16)(v[e] - (float)hi[e]); }
        _Float16* o = out + m * 3 * C + c;
        *(h4*)o = hi;
        *(h4*)(o + C) = lo;
        *(h4*)(o + 2 * C) = hi;
    }
}

// ---------------------------------------------------------------------------------------------
// conv-pool stem of shape_messsage_passing (openai_model_3d.py:757-764), fp32, tiny.
//   stage 1: Conv3d(3|4->32,k3,p1) @16^3 then MaxPool3d(2,2)  -> [O,32,8,8,8]   (4 input channels: 'concat' family)
//   stage 2: Conv3d(32->64,k3,p1) @8^3 then MaxPool3d(k=2,s=4) -> [O,64,2,2,2] -> flatten(512)
// ---------------------------------------------------------------------------------------------
// One workgroup per (object, pooled depth pd, pooled row ph): the 4 x 4 input (depth slice, row) lines 2pd-1 .. 2pd+2 x 2ph-1 .. 2ph+2
// (zero halo) sit in LDS with a one-voxel border, the 32 x Cx x 27 weights too; thread (c = tid / 8, pw = tid & 7) produces ONE pooled
// output of channel c (8 conv voxels x Cx x 27 taps).  (History: every tap from global: 220 us per step at O = 32; one workgroup per
// (object, pd) with 8 pooled outputs per thread: ~120 us whatever O -- hidden on the side branch of the single-GPU step, but on the
// critical path in front of the all-gather of a sharded step, where 4 objects gave only 32 workgroups.)
__global__ __launch_bounds__(256) void k_stem1(const es_stem_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Cx = a.Cin ? a.Cin : 3;
    float* xs = (float*)smem;                    // [Cx][4 slices][4 rows][18]
    float* ws = xs + Cx * 4 * 4 * 18;            // [32][Cx*27]
    const int o = blockIdx.y, pd = blockIdx.x >> 3, ph = blockIdx.x & 7, tid = threadIdx.x;
    const float* x = a.x + (long)o * (a.x_ostride ? a.x_ostride : Cx * 4096);
    for (int i = tid; i < Cx * 288; i += 256) {
        const int ww = i % 18, rr = (i / 18) & 3, sl = (i / 72) & 3, ci = i / 288;
        const int d = 2 * pd - 1 + sl, h = 2 * ph - 1 + rr, w = ww - 1;
        xs[i] = (d >= 0 && d < 16 && h >= 0 && h < 16 && w >= 0 && w < 16) ? x[ci * 4096 + d * 256 + h * 16 + w] : 0.f;
    }
    for (int i = tid; i < 32 * Cx * 27; i += 256) ws[i] = a.w0[i];
    __syncthreads();
    const int c = tid >> 3, pw = tid & 7;
    const float* w = ws + c * Cx * 27;
    const float bias = a.b0[c];
    float best = -INFINITY;
    for (int dz = 0; dz < 2; ++dz) for (int dy = 0; dy < 2; ++dy) for (int dx = 0; dx < 2; ++dx) {
        // conv output voxel (2pd+dz, 2ph+dy, 2pw+dx); LDS coordinates: slice dz+kd, row dy+kh, col 2pw+dx+kw.  Same summation order as
        // before (bias first, then ci, kd, kh, kw), so the result is bit-identical to the previous kernel.
        float sacc = bias;
        for (int ci = 0; ci < Cx; ++ci)
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        sacc += xs[((ci * 4 + dz + kd) * 4 + dy + kh) * 18 + 2 * pw + dx + kw] * w[ci * 27 + kd * 9 + kh * 3 + kw];
        best = fmaxf(best, sacc);
    }
    a.scratch[(((long)o * 32 + c) * 8 + pd) * 64 + ph * 8 + pw] = best;
}

__global__ __launch_bounds__(256) void k_stem2(const es_stem_args a) {
    // 8 lanes per output [o][c=64][2][2][2] (each lane 4 of the 32 input channels, all 8 window positions);
    // pooling windows start at 0 and 4 (kernel 2, stride 4).  A workgroup = 32 consecutive outputs = 4 channels of ONE object: the
    // object's pooled stage-1 map [32][8][8][8] (64 KB) is staged in LDS once -- the first version issued its 864 loads per thread
    // against global memory (50 us per step whatever O).  Same arithmetic and summation order.
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = (float*)smem;                    // [32][513]: channel stride 513 -> the 8 lanes of an output (4 channels apart) hit 8 banks
    const long n = (long)a.O * 512;
    const long gi = ((long)blockIdx.x * 256 + threadIdx.x);
    const long i = gi >> 3;
    const int sl = (int)(gi & 7);
    const long o = ((long)blockIdx.x * 32) >> 9;          // object of this workgroup
    {
        const f4* src = (const f4*)(a.scratch + o * 32 * 512);
        f4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[threadIdx.x + 256 * u];       // 16 loads in flight per thread
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = (threadIdx.x + 256 * u) * 4, ci = e >> 9, off = e & 511;
#pragma unroll
            for (int q = 0; q < 4; ++q) xs[ci * 513 + off + q] = v[u][q];
        }
    }
    __syncthreads();
    if (i >= n) return;
    const int pw = i & 1, ph = (i >> 1) & 1, pd = (i >> 2) & 1, c = (i >> 3) & 63;
    const float* w = a.w1 + c * 32 * 27;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    for (int cj = 0; cj < 4; ++cj) {
        const int ci = sl * 4 + cj;
        for (int kd = 0; kd < 3; ++kd) for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) {
            const float wv = w[ci * 27 + kd * 9 + kh * 3 + kw];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int id = 4 * pd + (k >> 2) + kd - 1, ih = 4 * ph + ((k >> 1) & 1) + kh - 1, iw = 4 * pw + (k & 1) + kw - 1;
                if (id >= 0 && id < 8 && ih >= 0 && ih < 8 && iw >= 0 && iw < 8) s[k] += xs[ci * 513 + id * 64 + ih * 8 + iw] * wv;
            }
        }
    }
    float best = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float v = s[k];
        v += __shfl_xor(v, 1, 8); v += __shfl_xor(v, 2, 8); v += __shfl_xor(v, 4, 8);
        best = fmaxf(best, v + a.b1[c]);
    }
    if (sl == 0) a.out[i] = best;       // i = o*512 + c*8 + pd*4 + ph*2 + pw  == nn.Flatten order of [O,64,2,2,2]
}

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM convolution / linear on MFMA (fp16 in, fp32 accumulate).
//   workgroup tile 256 / 128 / 64 (voxels) x 224 (output channels), K unit = 32 channels x one tap; waves as (rows) x 2 (columns),
//   wave tile 64 (32) x 112 = 4 (2) x 7 MFMA 16x16x32 tiles (112 accumulator VGPRs).
//   N = 224 / 448 / 672 (and 3*C, 8*C) are all multiples of 224 at full width; ragged N / M, halo and out-of-volume rows are
//   out-of-range lanes of the LDS-DMA gather (hardware zero fill) and masked stores.
// ---------------------------------------------------------------------------------------------
constexpr int BN = 224, BK = 32, BNP = 256;     // BNP: LDS rows of a weight tile (224 padded to a whole number of 1 KiB pieces per wave;
                                                // the 32 pad rows are zero-filled by out-of-range pieces, never fetched)
// LDS ring: 3 slots (units ks+1, ks+2 in flight).  A 6-deep ring for lone workgroups measured neutral (round 1).

struct ConvGeom {
    int O, D, H, W;          // output grid
    int Hi, Wi;              // input grid (H,W may differ from output for DOWN/UP)
    int Di;                  // input depth (differs from D for DOWN_DHW / UP_DHW)
    int lw, lh, ld;          // log2 of W, H, D (output)
};


// Split-K ranges are cut in the SAME places by every conv kernel: in units of one (channel chunk, kd, kh) group = 3 K steps
// for 27-tap launches (1 step otherwise), the fused 1x1 skip phase in single steps.  Giving all kernels the same cuts makes the partial sums -- and therefore the fp32 result -- independent of which tile
// size / kernel the dispatcher picked for a launch (sharded == unsharded runs).
__device__ __forceinline__ void split_range(int nks0, int kch2, int taps, int bz, int S, int& ks_begin, int& ks_end) {
    const int unit = taps == 27 ? 3 : 1;
    const int U0 = nks0 / unit, UT = U0 + kch2;
    const int ub = (int)((long)UT * bz / S), ue = (int)((long)UT * (bz + 1) / S);
    ks_begin = ub <= U0 ? ub * unit : nks0 + (ub - U0);
    ks_end = ue <= U0 ? ue * unit : nks0 + (ue - U0);
}


// XCD-aware tile mapping, shared by the conv kernels.  The dispatcher places hardware workgroup id b on XCD b % 8 (observed,
// speed only): taken literally, the 8 row tiles that share one weight slab (same column tile / K split) would sit on 8
// different XCDs and every XCD's L2 would pull the whole weight matrix from HBM (measured: the 16x4x4 level ran at 300 GB/s of
// weight traffic per XCD-copy, 6x off the MFMA time).  Remap (bijective for any grid size) so that each XCD owns a contiguous
// range of logical ids L, then order the tiles along L by what the launch re-uses:
//   3x3x3 convs (<= 3 column tiles): x (row tiles) fastest -- neighbours share the B slab and the A halo (adjacent depth
//     slices) inside one L2;
//   1x1x1 / linear launches with several column tiles (qkv: 6, FeedForward: 16-24): COLUMN tiles fastest inside panels of
//     column tiles whose weight slabs fit an L2 together (~3 MiB), so the workgroups that run side by side on an XCD share ONE
//     activation row tile and a resident weight panel.  With x fastest the 29 MB activation of the 16x8x8 FeedForward
//     projection was re-fetched once per column tile: rocprofv3 FETCH_SIZE 364 MB per launch (x2-corrected), MFMA busy 0.20.
__device__ __forceinline__ void conv_tile_of(const es_conv_args& a, int& bx, int& by, int& bz) {
    const int gx = gridDim.x, gy = gridDim.y;
    const int nwg = gx * gy * (int)gridDim.z;
    const int orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    const int per_z = gx * gy;
    bz = L / per_z;
    const int t = L - bz * per_z;
    if (a.taps == 1 && gy >= 2) {
        const long slab = ((long)a.Cin + (a.a2 ? a.Cin2 : 0)) * BN * 2;                 // bytes of one column tile's weights
        int npanel = (int)(((long)gy * slab + (3L << 20) - 1) / (3L << 20));
        npanel = npanel < 1 ? 1 : (npanel > gy ? gy : npanel);
        const int Pw = (gy + npanel - 1) / npanel;                                      // column tiles per panel
        const int full = gx * Pw;
        const int p = t / full, rr = t - p * full;
        const int w = (gy - p * Pw) < Pw ? (gy - p * Pw) : Pw;                          // the last panel may be narrower
        bx = rr / w;
        by = p * Pw + (rr - bx * w);
    } else {
        bx = t % gx;
        by = t / gx;
    }
}

// a wave-uniform pointer the compiler can keep in SGPRs (buffer descriptors must be scalar; a descriptor it cannot prove uniform
// is applied through a waterfall loop around every buffer instruction)
__device__ __forceinline__ void* uniform_ptr(const void* p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return (void*)(((unsigned long)hi << 32) | lo);
}

// ---------------------------------------------------------------------------------------------
// Epilogue shared by the conv kernels.
// MFMA D layout: lane holds D[row = q*4 + r][col = i16].  A direct store is 112 four-byte stores (+112 residual
// loads) per wave -- measured 25 % of the kernel at the 16^3 level.  Each 16-row slab is therefore transposed
// through LDS (the ring is free now) so that a lane owns 4 consecutive columns: 16-byte loads and stores.
// ---------------------------------------------------------------------------------------------
// LOWREG: k_conv_ws runs 12 waves per CU (168 VGPRs): the 7 float4 items per lane are processed one at a time instead of
// all in flight (fully unrolled the epilogue spilled 100 registers there and cost more than the K loop gained).
// EPI_: -1 = a.epilogue decides at run time (general kernels); ES_EPI_NONE / ES_EPI_GEGLU = compiled for that epilogue only
// (k_conv_ws: with both paths in one function the register allocator spilled 150-250 dwords at the 168-register cap).
// STATS_ (LOWREG only): besides storing the tile, form the row-group sums of es_conv_args.gn_stats_out from the stored values.
template <int BM_, int NW_, bool ACTIVE = true, bool LOWREG = false, int EPI_ = -1, bool NOSYNC = false, bool STATS_ = false>      // ACTIVE = false: a producer wave of k_conv_ws, joins the barriers only
__device__ __forceinline__ void conv_epilogue(const es_conv_args& a, const ConvGeom& g, f4 (&acc)[BM_ / (NW_ / 2) / 16][7],
                                              char* smem, long M, long m0, int n0, int wave, int lane, int S, int bz,
                                              int ncdhw) {
    constexpr int WROWS = BM_ / (NW_ / 2), MI = WROWS / 16;
    const bool geglu_epi = EPI_ < 0 ? a.epilogue == ES_EPI_GEGLU : EPI_ == ES_EPI_GEGLU;
    if constexpr (!ACTIVE) {
        __syncthreads();
        return;
    }
    const int wm = wave >> 1, wn = wave & 1, i16 = lane & 15, q = lane >> 4;
    const int V = g.D * g.H * g.W;
    if constexpr (!NOSYNC) __syncthreads();                   // all waves done with the ring (NOSYNC: the slabs live behind the ring, k_linear_ws)
    float* slab = (float*)smem + wave * (16 * 116);           // per wave: 16 rows x 112 cols (+4 pad) fp32 = 7.25 KB
    const bool vec_ok = !ncdhw && (a.N % 4 == 0) && (a.out_ld % 4 == 0) && (!a.rowvec || a.rowvec_ld % 4 == 0);
    float* part = S > 1 ? (float*)a.workspace + (long)bz * M * a.N : nullptr;   // [S][M][N] partial sums
    if (geglu_epi) {
        // Weight packing (PackedConv(geglu=True)): inside every 16-column MFMA tile, columns 0..7 are VALUE columns and columns
        // 8..15 the GATE columns of the same 8 outputs.  In the accumulator layout (lane = column i16, rows q*4 + r) the value
        // and the gate of one output element therefore sit in lanes i16 and i16 ^ 8 of ONE wave: two DPP row rotations exchange
        // them (the low lanes hand over their rows 2, 3 and receive the gates of rows 0, 1; the high lanes the reverse), every
        // lane evaluates two outputs, and the wave transposes its 16 x 56 fp16 results through a private 2.3 KB LDS slab into
        // 16-byte stores.  The first version kept value and gate columns in the two waves of a pair and exchanged whole fp32
        // slabs through LDS: 9 workgroup barriers, 112 ds_write_b32 and two slab reads per tile and wave.
        const bool lo = i16 < 8;
        const int cw = i16 & 7;
#pragma unroll
        for (int j = 0; j < 7; ++j) {                         // bias first: a lane's column of tile j is fixed
            const float bj = a.bias[n0 + wn * 112 + j * 16 + i16];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] += bj;
        }
        constexpr int HLD = 72;                               // halfs per slab row (56 + pad; 144 B keeps the 16-byte reads aligned)
        _Float16* hslab = (_Float16*)smem + wave * (16 * HLD);
        typedef unsigned int u4g __attribute__((ext_vector_type(4)));
        const int rr = q * 4 + (lo ? 0 : 2);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                // row_ror:8 with a BANK mask: the rotation is written only into lanes 0-7 (banks 0, 1) resp. 8-15 (banks 2, 3) of a row, the
                // other half keeps `old` -- the exchange and the selection in one instruction per operand (the first version selected
                // with six v_cndmask per pair around two unmasked rotations).  Low lanes (value columns) take the gates of their rows
                // 0, 1 from the partner and keep their values; high lanes (gate columns) take the values of rows 2, 3 and keep their gates.
                auto dpp8 = [](float old, float src, int bank_mask_lo) __attribute__((always_inline)) {
                    return bank_mask_lo
                        ? __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x128, 0xf, 0x3, false))
                        : __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x128, 0xf, 0xc, false));
                };
                const float g0 = dpp8(acc[i][j][2], acc[i][j][0], 1), g1 = dpp8(acc[i][j][3], acc[i][j][1], 1);
                const float v0 = dpp8(acc[i][j][0], acc[i][j][2], 0), v1 = dpp8(acc[i][j][1], acc[i][j][3], 0);
                const es_f2 ge = es_gelu_fast2(es_f2{g0, g1});      // (packed fp32 polynomial: two evaluations per issue slot)
                hslab[rr * HLD + j * 8 + cw] = (_Float16)(v0 * ge[0]);
                hslab[(rr + 1) * HLD + j * 8 + cw] = (_Float16)(v1 * ge[1]);
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): own slab writes visible to own wave
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 2; ++t) {                     // 16 rows x 7 pieces of 16 bytes
                // (buffer stores with an out-of-range offset for idle lanes / rows past M instead of a branch: behind a branch the
                //  compiler waits for the previous store -- s_waitcnt vmcnt(0) -- before every store, 8 serialised round trips per tile)
                const int idx = lane + 64 * t, idl = idx < 112 ? idx : 0;
                const int row = idl / 7, c8 = idl - row * 7;
                const long mrow = m0 + wm * WROWS + i * 16;
                const bool ok = idx < 112 && mrow + row < M;
                // (descriptor at the slab's first row, soffset 0: an SGPR soffset lost pieces -- see the LOWREG path below)
                const __amdgpu_buffer_rsrc_t rOg = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((_Float16*)a.out_f16 + mrow * (long)a.out_ld), (short)0, (int)0x80000000u, 0x00020000);
                const unsigned vo = ok ? ((unsigned)row * (unsigned)a.out_ld + (unsigned)((n0 >> 1) + wn * 56 + c8 * 8)) * 2u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4g, *(const h8*)&hslab[row * HLD + c8 * 8]), rOg, (int)vo, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
    if constexpr (LOWREG) {
        // The host routes a launch to k_conv_ws only when ws_epilogue_ok() holds (vector-aligned N / leading dimensions,
        // channels-last output, 32-bit element offsets: a uniform 64-bit base + one VGPR keeps the address registers out
        // of the 168-register budget); there is no scalar fall-back in this instantiation.
        {
            // k_conv_ws (168-register cap).  The first version walked the 7 float4 items of a slab one at a time and paid
            // one dependent bias / per-object vector / RESIDUAL load latency per item: 28 serial round trips per tile,
            // ~25 us of the ~30 us fixed cost of a 256-row tile (fit over Cin, profiles/r01_notes.md).  Now a lane owns ONE
            // column quad for the whole tile (56 of 64 lanes: 2 rows x 28 quads per pass, 8 passes per 16-row slab), so
            // bias and the per-object vector are loaded once, and the 8 residual quads of a slab are fetched together,
            // those of slab i+1 while slab i is combined and stored (from the second slab on, when the accumulators
            // already released leave the registers for it).
            const int vsh = g.lw + g.lh + g.ld;
            const int c4 = lane % 28, rsub = lane / 28;
            const int n = n0 + wn * 112 + c4 * 4;
            const bool n_ok = lane < 56 && n < a.N;
            const long mw0 = m0 + wm * WROWS;
            const int rows_left = (int)((M - mw0) < (long)WROWS ? (M - mw0 > 0 ? M - mw0 : 0) : (long)WROWS);   // valid rows of this wave
            f4 bias4 = {0.f, 0.f, 0.f, 0.f}, rv4 = {0.f, 0.f, 0.f, 0.f};
            if (!part && a.bias && n_ok) bias4 = *(const f4*)&a.bias[n];
            if (!part && a.rowvec && n_ok && rows_left > 0) rv4 = *(const f4*)&a.rowvec[(mw0 >> vsh) * a.rowvec_ld + n];
            const bool use_res = a.res && !part;
            const unsigned ld = (unsigned)a.out_ld;
            auto write_slab = [&](int i) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 7; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[(q * 4 + r) * 116 + j * 16 + i16] = acc[i][j][r];
            };
            f4 gs = {0.f, 0.f, 0.f, 0.f}, gq = {0.f, 0.f, 0.f, 0.f};          // STATS_: this lane's rows (rsub, rsub + 2, ...) of the wave's 64-row group
            // STRAIGHT-LINE row passes.  The first version tested the operands inside the pass (`if (a.res) v += ...; if (a.out_f32)
            // ...`, a per-row rowvec load when a wave straddled two objects): with loads and stores on different paths the compiler
            // cannot keep the in-order vmcnt budget and put s_waitcnt vmcnt(0) in front of EVERY row pass -- 32 serialised store
            // round trips per tile (tools/linear_stamps.py: 12.9 us for a 114 KB fp16 tile, the same 11.5 us for a 229 KB fp32
            // one; rounds 1-3 read that as an HBM-bound burst).  Here absent operands are buffer descriptors with zero records
            // (loads return 0, stores are dropped) and invalid lanes / rows carry an out-of-range offset, so a row pass is LDS read
            // -> three adds -> two buffer stores without control flow, and no store is ever waited for.  The host routes a launch
            // here only when a wave's 64 rows lie in ONE object (voxels per object % 64 == 0, or no rowvec): rv4 is per wave.
            typedef unsigned int u4v __attribute__((ext_vector_type(4)));
            typedef unsigned int u2v __attribute__((ext_vector_type(2)));
            constexpr unsigned OOBV = 0x80000000u;
            // The descriptors start at the wave's first row and every access uses soffset 0.  (A first version kept one descriptor
            // per tensor and passed the slab's row offset as the SGPR soffset: the GEGLU stores written that way lost 16-byte
            // pieces now and then at 4 objects per GPU -- the only shape that sends them through k_conv_ws -- whenever the
            // compiler reused the offset SGPR right behind the store; with the offset folded into the VGPR the runs are clean.)
            const unsigned ldp = part ? (unsigned)a.N : ld;
            float* const o32 = part ? part + mw0 * (long)a.N : (a.out_f32 ? a.out_f32 + mw0 * (long)a.out_ld : nullptr);
            const __amdgpu_buffer_rsrc_t rO32 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(o32), (short)0, o32 ? (int)OOBV : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rO16 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((!part && a.out_f16) ? (_Float16*)a.out_f16 + mw0 * (long)a.out_ld : nullptr), (short)0, (!part && a.out_f16) ? (int)OOBV : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rRes = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(use_res ? a.res + mw0 * (long)a.out_ld : nullptr), (short)0, use_res ? (int)OOBV : 0, 0x00020000);
            const int rsub_l = lane < 56 ? rsub : 0;                              // (idle lanes read a valid LDS row)
            const unsigned vrow32 = n_ok ? ((unsigned)rsub * ldp + (unsigned)n) * 4u : OOBV;
            const unsigned vrow16 = n_ok ? ((unsigned)rsub * ld + (unsigned)n) * 2u : OOBV;
            auto load_res_s = [&](int i, f4 (&rr)[8]) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const unsigned vo = (i * 16 + rsub + 2 * t < rows_left) ? vrow32 + (unsigned)(i * 16 + 2 * t) * ld * 4u : OOBV;
                    rr[t] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rRes, (int)vo, 0, 0));
                }
            };
            auto combine_s = [&](int i, const f4 (&rr)[8]) __attribute__((always_inline)) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const bool ok = i * 16 + rsub + 2 * t < rows_left;
                    f4 v = *(const f4*)&slab[(rsub_l + 2 * t) * 116 + c4 * 4];
                    v += bias4;
                    v += rv4;
                    v += rr[t];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), rO32, (int)(ok ? vrow32 + (unsigned)(i * 16 + 2 * t) * ldp * 4u : OOBV), 0, 0);
                    const h4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, hv), rO16, (int)(ok ? vrow16 + (unsigned)(i * 16 + 2 * t) * ld * 2u : OOBV), 0, 0);
                    if constexpr (STATS_) {
                        gs += v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) gq[e] = fmaf(v[e], v[e], gq[e]);
                    }
                }
            };
            if (part) {                              // split K: the raw partial tile (wave-uniform branch; nothing to load or add)
                write_slab(0);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const bool ok = i * 16 + rsub + 2 * t < rows_left;
                        const f4 v = *(const f4*)&slab[(rsub_l + 2 * t) * 116 + c4 * 4];
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), rO32, (int)(ok ? vrow32 + (unsigned)(i * 16 + 2 * t) * ldp * 4u : OOBV), 0, 0);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (i + 1 < MI) write_slab(i + 1);
                }
                return;
            }
            f4 rbuf[2][8];
            write_slab(0);
            __builtin_amdgcn_sched_barrier(0);
            load_res_s(0, rbuf[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0): own slab writes visible to own wave
                __builtin_amdgcn_wave_barrier();
                if (i >= 1 && i + 1 < MI) load_res_s(i + 1, rbuf[(i + 1) & 1]);       // two slabs already released
                __builtin_amdgcn_sched_barrier(0);
                combine_s(i, rbuf[i & 1]);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_wave_barrier();
                if (i + 1 < MI) write_slab(i + 1);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 0 && MI > 1) load_res_s(1, rbuf[1]);                            // behind slab 1's LDS writes
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (STATS_) {
                // One (sum, sum of squares) pair per column for the wave's 64 rows: the even rows were summed top to bottom by lanes
                // 0..27, the odd rows by lanes 28..55; even + odd is the order k_rowgroup_stats repeats.
                static_assert(WROWS == 64, "row groups of gn_stats_out are 64 rows");
                f4 os, oq;
#pragma unroll
                for (int e = 0; e < 4; ++e) { os[e] = __shfl(gs[e], lane + 28); oq[e] = __shfl(gq[e], lane + 28); }
                if (lane < 28 && n_ok && rows_left > 0) {
                    const unsigned nrg = (unsigned)((M + 63) >> 6);
                    const unsigned so = (unsigned)(mw0 >> 6) * (unsigned)a.N + (unsigned)n;
                    *(f4*)&a.gn_stats_out[so] = gs + os;
                    *(f4*)&a.gn_stats_out[so + nrg * (unsigned)a.N] = gq + oq;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (vec_ok) {
#pragma unroll
            for (int j = 0; j < 7; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[(q * 4 + r) * 116 + j * 16 + i16] = acc[i][j][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): own writes visible to own wave
            __builtin_amdgcn_wave_barrier();
            // 16 rows x 28 float4 = 448 float4 per slab, 7 per lane.  Straight-line as in the LOWREG path below: absent operands
            // are descriptors with zero records, idle lanes carry an out-of-range offset.  (The first version loaded bias, rowvec
            // and residual behind `if`s inside the item loop: the compiler put s_waitcnt vmcnt(0) between all of them -- three
            // dependent load round trips and a store wait per item, 425 full waits in the kernel.)  The descriptors start at the
            // slab's first row, so the per-lane offsets stay small whatever the tensor size.
            {
                typedef unsigned int u4e __attribute__((ext_vector_type(4)));
                typedef unsigned int u2e __attribute__((ext_vector_type(2)));
                constexpr unsigned OOBE = 0x80000000u;
                const long rb = m0 + wm * WROWS + i * 16;
                const int vshe = g.lw + g.lh + g.ld;
                float* const o32 = part ? part + rb * a.N : (a.out_f32 ? a.out_f32 + rb * a.out_ld : nullptr);
                const unsigned ldp = part ? (unsigned)a.N : (unsigned)a.out_ld, lde = (unsigned)a.out_ld;
                const bool ep = !part;
                const __amdgpu_buffer_rsrc_t rO32 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(o32), (short)0, o32 ? (int)OOBE : 0, 0x00020000);
                const __amdgpu_buffer_rsrc_t rO16 = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ep && a.out_f16 ? (_Float16*)a.out_f16 + rb * a.out_ld : nullptr), (short)0, (ep && a.out_f16) ? (int)OOBE : 0, 0x00020000);
                const __amdgpu_buffer_rsrc_t rRes = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ep && a.res ? a.res + rb * a.out_ld : nullptr), (short)0, (ep && a.res) ? (int)OOBE : 0, 0x00020000);
                const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ep ? a.bias : nullptr), (short)0, (ep && a.bias) ? (int)OOBE : 0, 0x00020000);
                const __amdgpu_buffer_rsrc_t rRv = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ep ? a.rowvec : nullptr), (short)0, (ep && a.rowvec) ? (int)OOBE : 0, 0x00020000);
                if (part) {                          // split K: the raw partial tile, nothing to load (wave-uniform branch)
#pragma unroll 7
                    for (int t = 0; t < 7; ++t) {
                        const int idx = lane + 64 * t;
                        const int row = idx / 28, c4 = idx - row * 28;
                        const int n = n0 + wn * 112 + c4 * 4;
                        const bool ok = rb + row < M && n < a.N;
                        const f4 v = *(const f4*)&slab[row * 116 + c4 * 4];
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4e, v), rO32, (int)(ok ? ((unsigned)row * ldp + (unsigned)n) * 4u : OOBE), 0, 0);
                    }
                } else
#pragma unroll 7
                for (int t = 0; t < 7; ++t) {
                    const int idx = lane + 64 * t;
                    const int row = idx / 28, c4 = idx - row * 28;
                    const int n = n0 + wn * 112 + c4 * 4;
                    const bool ok = rb + row < M && n < a.N;
                    f4 v = *(const f4*)&slab[row * 116 + c4 * 4];
                    const f4 vb = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rBias, (int)(ok ? (unsigned)n * 4u : OOBE), 0, 0));
                    const f4 vr = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rRv, (int)(ok ? ((unsigned)((rb + row) >> vshe) * (unsigned)a.rowvec_ld + (unsigned)n) * 4u : OOBE), 0, 0));
                    const f4 vs = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rRes, (int)(ok ? ((unsigned)row * lde + (unsigned)n) * 4u : OOBE), 0, 0));
                    v += vb;
                    v += vr;
                    v += vs;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4e, v), rO32, (int)(ok ? ((unsigned)row * ldp + (unsigned)n) * 4u : OOBE), 0, 0);
                    const h4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2e, hv), rO16, (int)(ok ? ((unsigned)row * lde + (unsigned)n) * 2u : OOBE), 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long m = m0 + wm * WROWS + i * 16 + q * 4 + r;
                if (m >= M) continue;
                const long o = m / V;
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int n = n0 + wn * 112 + j * 16 + i16;
                    if (n >= a.N) continue;
                    float v = acc[i][j][r];
                    if (a.bias) v += a.bias[n];
                    if (a.rowvec) v += a.rowvec[o * a.rowvec_ld + n];
                    if (a.res) v += a.res[m * a.out_ld + n];
                    if (ncdhw) {
                        a.out_f32[(o * a.N + n) * V + (m - o * V)] = v;
                    } else {
                        if (a.out_f32) a.out_f32[m * a.out_ld + n] = v;
                        if (a.out_f16) ((_Float16*)a.out_f16)[m * a.out_ld + n] = (_Float16)v;
                    }
                }
            }
        }
    }
}

// 27-bit validity mask of the 3x3x3 taps around source voxel (cd, ch, cw) of a [nd, nh, nw] grid (bit t = kd*9 + kh*3 + kw set when
// the tap lies inside): the outer product of three 3-bit axis masks, ~25 VALU operations.  The first version tested the 27 taps one
// by one (27 x 6 compares per row, 4 rows per producer lane): 7.3 us of every tile's 8.7 us before its first K unit was published
// (tools/conv_stamps.py), with the 8 consumer waves parked at their first barrier.
__device__ __forceinline__ unsigned tap_mask27(int cd, int nd, int ch, int nh, int cw, int nw) {
    const unsigned vw = ((cw >= 1 && cw <= nw) ? 1u : 0u) | ((cw >= 0 && cw < nw) ? 2u : 0u) | ((cw >= -1 && cw + 1 < nw) ? 4u : 0u);
    const unsigned vh = ((ch >= 1 && ch <= nh) ? 1u : 0u) | ((ch >= 0 && ch < nh) ? 2u : 0u) | ((ch >= -1 && ch + 1 < nh) ? 4u : 0u);
    const unsigned vd = ((cd >= 1 && cd <= nd) ? 1u : 0u) | ((cd >= 0 && cd < nd) ? 2u : 0u) | ((cd >= -1 && cd + 1 < nd) ? 4u : 0u);
    const unsigned m9 = ((vh & 1u) ? vw : 0u) | ((vh & 2u) ? vw << 3 : 0u) | ((vh & 4u) ? vw << 6 : 0u);
    return ((vd & 1u) ? m9 : 0u) | ((vd & 2u) ? m9 << 9 : 0u) | ((vd & 4u) ? m9 << 18 : 0u);
}

// Workgroup = NW_ waves as (NW_/2)(M) x 2(N); tile BM_ x 224:
//   <256, 8> wave tile 64 x 112 (16^3 and 16x8x8 levels: A+B bytes per flop -32 % vs <128,4>)
//   <128, 4> wave tile 64 x 112
//   < 64, 4> wave tile 32 x 112 (16x4x4 level: enough workgroups to cover the 256 CUs)
template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// ---------------------------------------------------------------------------------------------
// k_conv_lean: implicit-GEMM conv / linear for every mode (SAME, strided, nearest-up fused, 1x1 / linear, fused 1x1 skip
// phase), with the K loop stripped to what the hardware needs.  Why: ablation of the first conv kernel of round 1 (64-bit
// pointer selects against a zero page; retired in round 2) showed the loop was
// INSTRUCTION-ISSUE bound, not memory or MFMA bound -- ~400 instructions per K step (64-bit pointer selects against a
// zero page, per-piece tap arithmetic, M0 through VALU + readfirstlane, the inlined up-sampling path) against 28 MFMAs:
// 0.64 us per K step with all loads removed, 0.19 us of it MFMA.  Here:
//   * A and B stream through `buffer_load_dwordx4 ... offen lds`: the tap shift and the channel chunk are ONE scalar
//     soffset per K step, the per-lane voffset is loop-invariant, and halo / ragged rows are lanes whose voffset is
//     out of range -- the hardware writes zeros for them (probe: tools/probes/probe_buffer_lds.hip), no zero page,
//     no 64-bit select;
//   * the per-tap offsets sit in one VGPR (lane t = tap t) and are fetched with v_readlane;
//   * the ring slot is a compile-time constant (loop unrolled x3): LDS fragment reads use immediate offsets.
// Tried on top of this and measured neutral (kept out): a register double buffer of the fragments (reads of step
// ks+1 under the MFMAs of step ks, 4 tiles in flight), a 6-deep ring for lone workgroups, issuing a wave's DMA pieces
// in one block staggered against its SIMD partner.
// ---------------------------------------------------------------------------------------------
// UP_: nearest-neighbour up-sampling fused into the gather (UP_HW / UP_DHW): the source of tap k along an up-sampled
// axis is (x + k) >> 1, i.e. the centre source shifted by -1 (k = -1, x even), +1 (k = +1, x odd) or 0 -- two per-lane
// byte shifts per axis, selected by the wave-uniform tap.
template <int BM_, int NW_, bool UP_ = false>
__global__ __launch_bounds__(64 * NW_, 2) void k_conv_lean(const es_conv_args a, const ConvGeom g, int ncdhw) {
    constexpr int NS = 3;
    constexpr int NT = 64 * NW_;
    constexpr int WROWS = BM_ / (NW_ / 2);
    constexpr int MI = WROWS / 16;
    constexpr int NA = BM_ * 4 / NT;
    constexpr int NB = BNP * 4 / NT;
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BNP * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NLOAD = NA + NB;
    constexpr unsigned OOB = 0x80000000u;        // >= num_records of both descriptors: the lane's 16 B arrive as zeros
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const long M = (long)g.O * g.D * g.H * g.W;
    int bx, by, bz;
    conv_tile_of(a, bx, by, bz);                 // XCD-aware, re-use-aware tile order
    const long m0 = (long)bx * BM_;
    const int n0 = by * BN;

    // ---- per-lane staging roles ----
    int a_lc[NA], a_o[NA], a_d[NA], a_h[NA], a_w[NA];
    bool a_ok[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int p = tid + NT * j;
        const int row = p >> 2;
        a_lc[j] = (p & 3) ^ f_swz(row);
        const long m = m0 + row;
        a_ok[j] = m < M;
        const long mm = a_ok[j] ? m : 0;
        a_w[j] = (int)(mm & (g.W - 1));
        a_h[j] = (int)((mm >> g.lw) & (g.H - 1));
        a_d[j] = (int)((mm >> (g.lw + g.lh)) & (g.D - 1));
        a_o[j] = (int)(mm >> (g.lw + g.lh + g.ld));
    }
    const int kch0 = a.Cin >> 5;
    const int nks0 = a.taps * kch0;
    const int S = gridDim.z;
    int ks_begin, ks_end;
    split_range(nks0, a.a2 ? (a.Cin2 >> 5) : 0, a.taps, bz, S, ks_begin, ks_end);
    const int nloc = ks_end - ks_begin;
    // K-step generator state (wave-uniform): phase, tap, channel chunk, byte offset of the B block
    int st_phase = ks_begin >= nks0 ? 1 : 0;
    int st_tap = st_phase ? 0 : ks_begin % a.taps;
    int st_c = st_phase ? (ks_begin - nks0) : (ks_begin / a.taps);               // chunk index (32 channels)
    int st_ntap = st_phase ? 1 : a.taps, st_kch = st_phase ? (a.Cin2 >> 5) : kch0;
    unsigned st_boff = (unsigned)(st_phase ? (ks_begin - nks0) : ks_begin) * (unsigned)B_BYTES;
    unsigned voff[NA], msk[NA];
    int upm[NA][3], upp[NA][3];                  // UP_: byte shift of tap -1 / +1 along (d, h, w) for this lane's row
    int dtab = 0;                                // lane t: byte shift of tap t (+ bias so that it is >= 0)
    __amdgpu_buffer_rsrc_t rA, rB;
    auto set_phase = [&]() __attribute__((always_inline)) {
        const _Float16* Wg = (const _Float16*)(st_phase ? a.w2 : a.w);
        const long nks_ph = st_phase ? (long)(a.Cin2 >> 5) : (long)nks0;
        rB = __builtin_amdgcn_make_buffer_rsrc((void*)(Wg + ((long)by * nks_ph) * (BNP * BK)), (short)0, (int)OOB, 0x00020000);
        const _Float16* Ag = (const _Float16*)(st_phase ? a.a2 : a.a);
        const int Cin = st_phase ? a.Cin2 : a.Cin;
        const bool down = !st_phase && (a.mode == ES_CONV_DOWN_HW || a.mode == ES_CONV_DOWN_DHW);
        const bool downd = !st_phase && a.mode == ES_CONV_DOWN_DHW;
        const int Dsrc = downd ? 2 * g.D : g.D;
        const int Hi = st_phase ? g.H : g.Hi, Wi = st_phase ? g.W : g.Wi;
        const int ntap = st_phase ? 1 : a.taps;
        const bool updhw = UP_ && a.mode == ES_CONV_UP_DHW;
        const int Di = updhw ? g.D / 2 : g.D;
        const int bias = ntap == 27 ? ((Hi + 1) * Wi + 1) * Cin * 2 : 0;          // bytes; largest negative tap shift
        rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Ag - bias), (short)0, (int)OOB, 0x00020000);
        {
            const int t = lane < 27 ? lane : 13;
            const int kd = t / 9 - 1, kh = (t / 3) % 3 - 1, kw = t % 3 - 1;
            dtab = ntap == 27 ? ((kd * Hi + kh) * Wi + kw) * Cin * 2 + bias : 0;
            if (UP_) dtab = ntap == 27 ? (updhw ? 0 : kd * Hi * Wi * Cin * 2) + bias : 0;     // h, w (and d) shifts are per lane
        }
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int ch = down ? 2 * a_h[j] : a_h[j];
            const int cw = down ? 2 * a_w[j] : a_w[j];
            const int cd = downd ? 2 * a_d[j] : a_d[j];
            voff[j] = (unsigned)(((((long)a_o[j] * Dsrc + cd) * Hi + ch) * Wi + cw) * Cin * 2 + a_lc[j] * 16);
            if (UP_) {
                const int sd = updhw ? a_d[j] >> 1 : a_d[j];
                voff[j] = (unsigned)(((((long)a_o[j] * Di + sd) * Hi + (a_h[j] >> 1)) * Wi + (a_w[j] >> 1)) * Cin * 2 + a_lc[j] * 16);
                const int SD = Hi * Wi * Cin * 2, SH = Wi * Cin * 2, SW = Cin * 2;
                upm[j][0] = (updhw && !(a_d[j] & 1)) ? -SD : 0; upp[j][0] = (updhw && (a_d[j] & 1)) ? SD : 0;
                upm[j][1] = !(a_h[j] & 1) ? -SH : 0;            upp[j][1] = (a_h[j] & 1) ? SH : 0;
                upm[j][2] = !(a_w[j] & 1) ? -SW : 0;            upp[j][2] = (a_w[j] & 1) ? SW : 0;
            }
            unsigned m = 0;
            if (ntap == 1) {
                m = a_ok[j] ? 1u : 0u;
            } else {
                m = a_ok[j] ? tap_mask27(cd, Dsrc, ch, UP_ ? g.H : Hi, cw, UP_ ? g.W : Wi) : 0u;
            }
            msk[j] = m;
        }
    };
    set_phase();
    const unsigned voffB = (unsigned)tid * 16u;

    unsigned sA = 0, sbit = 1;                   // per-K-step scalars of the tile being staged
    int ukd = 0, ukh = 0, ukw = 0;               // UP_: tap coordinates of the tile being staged
    auto stage_prep = [&]() __attribute__((always_inline)) {
        sA = (unsigned)__builtin_amdgcn_readlane(dtab, st_tap) + (unsigned)st_c * 64u;
        sbit = 1u << st_tap;
        if (UP_) { ukd = st_tap / 9 - 1; ukh = (st_tap / 3) % 3 - 1; ukw = st_tap % 3 - 1; }
    };
    auto stage_piece = [&](int slot, int pj) __attribute__((always_inline)) {
        char* dst = smem + slot * STAGE_BYTES + wave * 1024;
        if (pj < NA) {
            unsigned vs = voff[pj];
            if (UP_) {
                vs += (unsigned)(ukd < 0 ? upm[pj][0] : ukd > 0 ? upp[pj][0] : 0);
                vs += (unsigned)(ukh < 0 ? upm[pj][1] : ukh > 0 ? upp[pj][1] : 0);
                vs += (unsigned)(ukw < 0 ? upm[pj][2] : ukw > 0 ? upp[pj][2] : 0);
            }
            const unsigned vo = (msk[pj] & sbit) ? vs : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(dst + pj * (NT * 16)), 16, (int)vo, (int)sA, 0, 0);
        } else {
            const int j = pj - NA;
            // (bytes >= 224 rows x 64 B of the block are the zero rows that pad the weight tile to 256: zero fill, no L2 traffic)
            const unsigned vB = (unsigned)tid * 16u + (unsigned)j * (NT * 16) >= (unsigned)(BN * BK * 2) ? OOB : voffB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(dst + A_BYTES + j * (NT * 16)), 16, (int)vB,
                                                     (int)(st_boff + (unsigned)j * (NT * 16)), 0, 0);
        }
    };
    auto stage_advance = [&]() __attribute__((always_inline)) {
        st_boff += (unsigned)B_BYTES;
        if (++st_tap == st_ntap) {
            st_tap = 0;
            if (++st_c == st_kch && !st_phase && a.a2) {
                st_phase = 1; st_c = 0; st_ntap = 1; st_kch = a.Cin2 >> 5; st_boff = 0;
                set_phase();
            }
        }
    };
    auto stage_all = [&](int slot) __attribute__((always_inline)) {
        stage_prep();
#pragma unroll
        for (int pj = 0; pj < NLOAD; ++pj) stage_piece(slot, pj);
        stage_advance();
    };

    f4 acc[MI][7];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};

    const int i16 = lane & 15, q = lane >> 4;
    // fragment read addresses: row bits 2..3 (the swizzle key) come from i16 only, so one per-lane base serves every
    // 16-row MFMA tile with an immediate offset
    const int fragA = (wm * WROWS + i16) * 64 + ((q ^ f_swz(i16)) << 4);
    const int fragB = A_BYTES + (wn * 112 + i16) * 64 + ((q ^ f_swz(i16)) << 4);

    stage_all(0);
    if (nloc > 1) stage_all(1);
    int ks = 0;
    auto body = [&](auto slot_c) __attribute__((always_inline)) {
        constexpr int RS = decltype(slot_c)::value;          // ring slot read in this step
        constexpr int WS = (RS + 2) % NS;                    // ring slot refilled (tile ks + 2)
        if (ks + 1 < nloc) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();            // tile ks visible to all waves; all waves done reading slot WS
        const bool pf = ks + 2 < nloc;
        if (pf) stage_prep();
        h8 af[MI], bfr[7];
        const char* As = smem + RS * STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 7; ++j) bfr[j] = *(const h8*)(As + fragB + j * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const h8*)(As + fragA + i * 1024);
        constexpr int PPR = (NLOAD + MI - 1) / MI;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < 7; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (pf) {
#pragma unroll
                for (int pp = 0; pp < PPR; ++pp)
                    if (i * PPR + pp < NLOAD) stage_piece(WS, i * PPR + pp);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pf) stage_advance();
    };
    while (true) {
        body(std::integral_constant<int, 0>{}); if (++ks >= nloc) break;
        body(std::integral_constant<int, 1>{}); if (++ks >= nloc) break;
        body(std::integral_constant<int, 2>{}); if (++ks >= nloc) break;
    }
    conv_epilogue<BM_, NW_>(a, g, acc, smem, M, m0, n0, wave, lane, S, bz, ncdhw);
}

// ---------------------------------------------------------------------------------------------
// k_linear_deep: 1x1 / linear launches that are SMALL and K-SHORT (the transformer linears at few objects per GPU: 1024-4096 rows,
// 14-21 K units).  On the 3-slot ring of k_conv_lean<64> such a workgroup is a lone latency chain -- two units in flight, every further
// unit waits a full L2 -> LDS round trip: 22-30 us for < 1 GFLOP -- and round 2's answer (split K over 2-4 workgroups + a reduction
// kernel, "tiny split") is two launches of 7 + 8 us.  Here the ring is SEVEN slots deep (7 x 20 KB) and the whole ring is issued at
// kernel entry, so the K loop pays ONE round trip and then runs at LDS speed; no split, no reduction kernel, the full epilogue
// (bias / per-object vector / residual / f16 copy / GEGLU) in the launch.  Same tile (64 x 224), LDS image, K order and MFMA
// accumulation chain as k_conv_lean<64> with S = 1: bit-identical to the unsplit launch of any other conv kernel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_linear_deep(const es_conv_args a, const ConvGeom g) {
    constexpr int BM_ = 64, NW_ = 4, NS = 7, NT = 256;
    constexpr int WROWS = BM_ / (NW_ / 2), MI = WROWS / 16;        // 32 rows per wave, 2 MFMA row tiles
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BNP * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NA = BM_ * 4 / NT, NB = BNP * 4 / NT, NLOAD = NA + NB;           // 1 + 4 pieces per thread and unit
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const long M = (long)g.O * g.D * g.H * g.W;
    int bx, by, bz;
    conv_tile_of(a, bx, by, bz);
    const long m0 = (long)bx * BM_;
    const int n0 = by * BN;
    const int nloc = a.Cin >> 5;
    // staging: thread's A piece = row tid >> 2, 16-B chunk (tid & 3) ^ swizzle; B pieces = the block's 16 KiB, 4 per thread
    unsigned voffA;
    {
        const int row = tid >> 2;
        const long m = m0 + row;
        voffA = m < M ? (unsigned)(m * a.Cin * 2 + (((tid & 3) ^ f_swz(row)) * 16)) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.a), (short)0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr((const _Float16*)a.w + ((long)by * nloc) * (BNP * BK)), (short)0, (int)OOB, 0x00020000);
    auto stage = [&](int ks, int slot) __attribute__((always_inline)) {
        char* dst = smem + slot * STAGE_BYTES + wave * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)dst, 16, (int)voffA, (int)((unsigned)ks * 64u), 0, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const unsigned vB = (unsigned)tid * 16u + (unsigned)j * (NT * 16) >= (unsigned)(BN * BK * 2) ? OOB : (unsigned)tid * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(dst + A_BYTES + j * (NT * 16)), 16, (int)vB,
                                                     (int)((unsigned)ks * (unsigned)B_BYTES + (unsigned)j * (NT * 16)), 0, 0);
        }
    };
    f4 acc[MI][7];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, q = lane >> 4;
    const int fragA = (wm * WROWS + i16) * 64 + ((q ^ f_swz(i16)) << 4);
    const int fragB = A_BYTES + (wn * 112 + i16) * 64 + ((q ^ f_swz(i16)) << 4);
    int issued = 0;
    for (; issued < NS - 1 && issued < nloc; ++issued) stage(issued, issued);
    int slot = 0;
    for (int ks = 0; ks < nloc; ++ks) {
        // unit ks has landed when at most (units issued after it) x NLOAD of this thread's loads are outstanding
        const int after = issued - 1 - ks;
        if (after >= 5) wait_vmcnt<5 * NLOAD>(); else if (after == 4) wait_vmcnt<4 * NLOAD>(); else if (after == 3) wait_vmcnt<3 * NLOAD>();
        else if (after == 2) wait_vmcnt<2 * NLOAD>(); else if (after == 1) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();            // unit ks visible to all waves; the slot of unit ks - 1 is free
        if (issued < nloc) { const int ws_ = slot == 0 ? NS - 1 : slot - 1; stage(issued, ws_); ++issued; }
        const char* As = smem + slot * STAGE_BYTES;
        h8 af[MI], bfr[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) bfr[j] = *(const h8*)(As + fragB + j * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const h8*)(As + fragA + i * 1024);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        slot = slot == NS - 1 ? 0 : slot + 1;
    }
    conv_epilogue<BM_, NW_>(a, g, acc, smem, M, m0, n0, wave, lane, 1, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// k_conv_ws: warp-specialised version of k_conv_lean (same tile, LDS image, K order and epilogue).
// Why: in k_conv_lean every wave alternates MFMA rows with LDS-DMA issue, and a wave is blocked for ~100-190 cycles
// per DMA piece while the TA path accepts it -- the K step costs MFMA time PLUS DMA issue time (ablation: 0.46 us + 0.5
// us per step; tools/microbench_conv.py).  A probe (tools/probes/probe_mfma_vs_dma.hip) shows the block is per WAVE, not
// per SIMD: a wave that only issues DMA does not slow the MFMA stream of its SIMD partners at all (16.6 cycles per
// MFMA with or without it).  So the roles are split: NC_ consumer waves (2 per SIMD; LDS fragment reads + MFMA, never
// touch vmcnt) and NP_ producer waves (1 per SIMD; all LDS-DMA of the tile, counted vmcnt).  12 waves per CU need
// <= 168 VGPRs per lane; the consumers hold 112 accumulators + 44 fragment registers.
//
// One s_barrier per K unit (32-channel chunk x tap, 16 KiB A + 16 KiB B), shared by both roles, 3-slot ring:
//   producer:  wait(own pieces of unit ks) -> barrier -> issue unit ks+2 into the slot the consumers just released
//   consumer:  28 MFMAs of unit ks, with the barrier of unit ks+1 after the first MFMA row and the fragments of unit ks+1 read
//              into the registers the MFMA stream has released (see the consumer loop)
// Measured in round 2 and NOT kept (profiles/r02_notes.md): two K units per barrier with the second unit's fragments re-filled
// after last use (341 -> 351 us on the 16^3 224->224 launch; the 3-tiles-in-flight variant of round 1 was equal too), and the A
// tile shared by the three kw taps of a (chunk, kd, kh) group, i.e. A LDS-DMA / 3 (340 -> 360 us, commit 53fb667).  The
// launch runs ~690k cycles at 1.97 GHz with the matrix pipe busy 49 % (exactly 16 cycles per MFMA): neither barrier count nor
// DMA piece count is what the other half waits for.  Nor is it the phase relation of the two consumer waves of a SIMD (LDS-flag
// hand-offs with a per-SIMD turn token instead of the barrier, 4-slot ring, commit f3c4f56: 341 -> 379 us) or the issue priority of
// the producer waves (s_setprio 3: 340 -> 336 us, noise).  SQ buckets over all 12 waves: 45 % parked (waitcnt / barrier), 36 %
// issue-stalled, 19 % issuing.
// ---------------------------------------------------------------------------------------------
template <int MI>
__device__ __forceinline__ void ws_read_frags(const char* As, int fragA, int fragB, h8 (&af)[MI], h8 (&bfr)[7]) {
#pragma unroll
    for (int j = 0; j < 7; ++j) bfr[j] = *(const h8*)(As + fragB + j * 1024);
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = *(const h8*)(As + fragA + i * 1024);
}

template <int MI>
__device__ __forceinline__ void ws_mma(f4 (&acc)[MI][7], const h8 (&af)[MI], const h8 (&bfr)[7]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
}

#ifdef ES_STAMP
// phase stamps of k_conv_ws (tools/conv_stamps.py; build with ES_BUILD_FLAGS=-DES_STAMP): 8 x 100 MHz wall-clock ticks per wave
__device__ unsigned long long* g_stamp_buf = nullptr;
#define ES_STAMP_AT(k) do { if (stamp && lane == 0) stamp[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ES_STAMP_AT(k) do { } while (0)
#endif

// One (tile, K range) of k_conv_ws: producer / consumer roles, 3-slot ring, epilogue.  S > 1: the raw partial tile goes to slab bz of
// a.workspace (split K / stream-K); S == 1: the final epilogue.  Every wave of the workgroup calls it with the same arguments.
// counted wait of a producer wave that runs up to MAXA units of NL loads ahead of the unit it publishes
template <int NL, int MAXA>
__device__ __forceinline__ void wait_ahead(int ahead) {
    if constexpr (MAXA > 0) {
        if (ahead >= MAXA) { wait_vmcnt<MAXA * NL>(); return; }
        wait_ahead<NL, MAXA - 1>(ahead);
    } else {
        wait_vmcnt<0>();
    }
}

template <int BM_, int NC_, int NP_, bool UP_, int EPI_, bool STATS_, int NS_ = 3>
__device__ __forceinline__ void conv_ws_tile(const es_conv_args& a, const ConvGeom& g, int ncdhw, char* smem, const int wave, const int lane,
                                             const long M, const int bx, const int by, const int ks_begin, const int ks_end,
                                             const int S, const int bz, unsigned long long* stamp) {
    constexpr int NS = NS_, UPS_ = 1;                             // ring depth (3; deeper for the small tiles of round 6, whose K units are
                                                                  // bound by the landing latency of a unit / (NS - 1)); K units per barrier
    constexpr int WROWS = BM_ / (NC_ / 2);
    constexpr int MI = WROWS / 16;
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BNP * BK * 2, UNIT_BYTES = A_BYTES + B_BYTES;
    constexpr int STAGE_BYTES = UPS_ * UNIT_BYTES;
    constexpr int APIECES = BM_ / 16, BPIECES = BNP / 16;          // 1 KiB pieces per unit
    constexpr int NA = APIECES / NP_, NB = BPIECES / NP_, NLOAD = NA + NB;        // per producer wave per unit
    static_assert(APIECES % NP_ == 0 && BPIECES % NP_ == 0, "pieces must divide over the producer waves");
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    (void)stamp;
    const long m0 = (long)bx * BM_;
    const int n0 = by * BN;
    const int kch0 = a.Cin >> 5;
    const int nks0 = a.taps * kch0;
    const int nloc = ks_end - ks_begin;                           // K units of this workgroup
    const int nstage = (nloc + UPS_ - 1) / UPS_;
    if (wave >= NC_) {
        // =============================== producer ===============================
        const int pw = wave - NC_;
        int a_lc[NA], a_o[NA], a_d[NA], a_h[NA], a_w[NA];
        bool a_ok[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int p = (pw + NP_ * j) * 64 + lane;    // 16-B slot of the A tile: row = p >> 2, physical chunk = p & 3
            const int row = p >> 2;
            a_lc[j] = (p & 3) ^ f_swz(row);
            const long m = m0 + row;
            a_ok[j] = m < M;
            const long mm = a_ok[j] ? m : 0;
            a_w[j] = (int)(mm & (g.W - 1));
            a_h[j] = (int)((mm >> g.lw) & (g.H - 1));
            a_d[j] = (int)((mm >> (g.lw + g.lh)) & (g.D - 1));
            a_o[j] = (int)(mm >> (g.lw + g.lh + g.ld));
        }
        int st_phase = ks_begin >= nks0 ? 1 : 0;
        int st_tap = st_phase ? 0 : ks_begin % a.taps;
        int st_c = st_phase ? (ks_begin - nks0) : (ks_begin / a.taps);
        int st_ntap = st_phase ? 1 : a.taps, st_kch = st_phase ? (a.Cin2 >> 5) : kch0;
        unsigned st_boff = (unsigned)(st_phase ? (ks_begin - nks0) : ks_begin) * (unsigned)B_BYTES;
        unsigned voff[NA], msk[NA];
        int upm[NA][3], upp[NA][3];
        int dtab = 0;
        __amdgpu_buffer_rsrc_t rA, rB;
        auto set_phase = [&]() __attribute__((always_inline)) {
            const _Float16* Wg = (const _Float16*)(st_phase ? a.w2 : a.w);
            const long nks_ph = st_phase ? (long)(a.Cin2 >> 5) : (long)nks0;
            rB = __builtin_amdgcn_make_buffer_rsrc((void*)(Wg + ((long)by * nks_ph) * (BNP * BK)), (short)0, (int)OOB, 0x00020000);
            const _Float16* Ag = (const _Float16*)(st_phase ? a.a2 : a.a);
            const int Cin = st_phase ? a.Cin2 : a.Cin;
            const bool down = !st_phase && (a.mode == ES_CONV_DOWN_HW || a.mode == ES_CONV_DOWN_DHW);
            const bool downd = !st_phase && a.mode == ES_CONV_DOWN_DHW;
            const int Dsrc = downd ? 2 * g.D : g.D;
            const int Hi = st_phase ? g.H : g.Hi, Wi = st_phase ? g.W : g.Wi;
            const int ntap = st_phase ? 1 : a.taps;
            const bool updhw = UP_ && a.mode == ES_CONV_UP_DHW;
            const int Di = updhw ? g.D / 2 : g.D;
            const int bias = ntap == 27 ? ((Hi + 1) * Wi + 1) * Cin * 2 : 0;
            rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Ag - bias), (short)0, (int)OOB, 0x00020000);
            {
                const int t = lane < 27 ? lane : 13;
                const int kd = t / 9 - 1, kh = (t / 3) % 3 - 1, kw = t % 3 - 1;
                dtab = ntap == 27 ? ((kd * Hi + kh) * Wi + kw) * Cin * 2 + bias : 0;
                if (UP_) dtab = ntap == 27 ? (updhw ? 0 : kd * Hi * Wi * Cin * 2) + bias : 0;
            }
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                const int ch = down ? 2 * a_h[j] : a_h[j];
                const int cw = down ? 2 * a_w[j] : a_w[j];
                const int cd = downd ? 2 * a_d[j] : a_d[j];
                voff[j] = (unsigned)(((((long)a_o[j] * Dsrc + cd) * Hi + ch) * Wi + cw) * Cin * 2 + a_lc[j] * 16);
                if (UP_) {
                    const int sd = updhw ? a_d[j] >> 1 : a_d[j];
                    voff[j] = (unsigned)(((((long)a_o[j] * Di + sd) * Hi + (a_h[j] >> 1)) * Wi + (a_w[j] >> 1)) * Cin * 2 + a_lc[j] * 16);
                    const int SD = Hi * Wi * Cin * 2, SH = Wi * Cin * 2, SW = Cin * 2;
                    upm[j][0] = (updhw && !(a_d[j] & 1)) ? -SD : 0; upp[j][0] = (updhw && (a_d[j] & 1)) ? SD : 0;
                    upm[j][1] = !(a_h[j] & 1) ? -SH : 0;            upp[j][1] = (a_h[j] & 1) ? SH : 0;
                    upm[j][2] = !(a_w[j] & 1) ? -SW : 0;            upp[j][2] = (a_w[j] & 1) ? SW : 0;
                }
                unsigned m = 0;
                if (ntap == 1) {
                    m = a_ok[j] ? 1u : 0u;
                } else {
                    m = a_ok[j] ? tap_mask27(cd, Dsrc, ch, UP_ ? g.H : Hi, cw, UP_ ? g.W : Wi) : 0u;
                }
                msk[j] = m;
            }
        };
        set_phase();
        ES_STAMP_AT(1);
        const unsigned voffB = (unsigned)lane * 16u;
        auto issue_unit = [&](char* dst) __attribute__((always_inline)) {        // dst: LDS base of the unit (A tile, then B tile)
            const unsigned sA = (unsigned)__builtin_amdgcn_readlane(dtab, st_tap) + (unsigned)st_c * 64u;
            const unsigned sbit = 1u << st_tap;
            int ukd = 0, ukh = 0, ukw = 0;
            if (UP_) { ukd = st_tap / 9 - 1; ukh = (st_tap / 3) % 3 - 1; ukw = st_tap % 3 - 1; }
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                unsigned vs = voff[j];
                if (UP_) {
                    vs += (unsigned)(ukd < 0 ? upm[j][0] : ukd > 0 ? upp[j][0] : 0);
                    vs += (unsigned)(ukh < 0 ? upm[j][1] : ukh > 0 ? upp[j][1] : 0);
                    vs += (unsigned)(ukw < 0 ? upm[j][2] : ukw > 0 ? upp[j][2] : 0);
                }
                const unsigned vo = (msk[j] & sbit) ? vs : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(dst + (pw + NP_ * j) * 1024), 16, (int)vo, (int)sA, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int q = pw + NP_ * j;
                // pieces 14, 15 are the 32 zero rows that pad the 224-column weight tile to 256: never read by the consumers --
                // an out-of-range voffset makes them zero fills without L2 traffic (the load count per wave stays the same; A/B on one
                // box: 3x3x3 launches -0.5 ... -1 %, full step 47.9 -> 48.3 steps/s)
                const unsigned vB = q * 16 >= BN ? OOB : voffB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(dst + A_BYTES + q * 1024), 16, (int)vB,
                                                         (int)(st_boff + (unsigned)q * 1024u), 0, 0);
            }
            st_boff += (unsigned)B_BYTES;
            if (++st_tap == st_ntap) {
                st_tap = 0;
                if (++st_c == st_kch && !st_phase && a.a2) {
                    st_phase = 1; st_c = 0; st_ntap = 1; st_kch = a.Cin2 >> 5; st_boff = 0;
                    set_phase();
                }
            }
        };
        int issued = 0;                          // K units issued so far
        auto issue_stage = [&](int st) __attribute__((always_inline)) {
            char* base = smem + (st % NS) * STAGE_BYTES;
#pragma unroll
            for (int u = 0; u < UPS_; ++u)
                if (issued < nloc) { issue_unit(base + u * UNIT_BYTES); ++issued; }
        };
        {
            static_assert(NLOAD * (NS - 2) <= 63, "vmcnt is a 6-bit counter");
#pragma unroll
            for (int st = 0; st < NS - 1; ++st)
                if (st < nstage) issue_stage(st);
            for (int st = 0; st < nstage; ++st) {
                wait_ahead<NLOAD, NS - 2>(nstage - 1 - st);                      // own pieces of unit st have landed (up to NS - 2 later units in flight)
                __builtin_amdgcn_s_barrier();        // unit st visible to the consumers; the slot of unit st - 1 released by them
                if (st == 0) ES_STAMP_AT(2);
                if (st + NS - 1 < nstage) issue_stage(st + NS - 1);
            }
        }
        ES_STAMP_AT(3);
        f4 dummy[MI][7];
        conv_epilogue<BM_, NC_, false, true, EPI_, false, STATS_>(a, g, dummy, smem, M, m0, n0, wave, lane, S, bz, ncdhw);
        ES_STAMP_AT(4);
        return;
    }
    // =============================== consumer ===============================
    const int wm = wave >> 1, wn = wave & 1;
    f4 acc[MI][7];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, q = lane >> 4;
    const int fragA = (wm * WROWS + i16) * 64 + ((q ^ f_swz(i16)) << 4);
    const int fragB = A_BYTES + (wn * 112 + i16) * 64 + ((q ^ f_swz(i16)) << 4);
    h8 af[MI], bfr[7];
    {
        // Software pipeline WITHOUT a second fragment buffer (112 accumulators + 44 fragment registers is all the 168-register
        // cap of 12 waves per CU allows): the fragments of unit ks+1 are read INTO THE REGISTERS OF UNIT ks as the MFMA
        // stream releases them -- A row i-1 while row i multiplies, B column j right after its last use in the last row.
        // The barrier that publishes unit ks+1 therefore sits after the first MFMA row of unit ks (all reads of slot ks are
        // waited for there), and the LDS reads of a unit (11 x ds_read_b128 per wave, ~350 LDS-array cycles per CU and unit)
        // run under the matrix pipe instead of in front of it.  A/B on one box (ES_CONV_PIPE, since removed): 3x3x3 launches
        // -2.5 ... -3.2 % (319 -> 311 us at 16^3 224->224, 283 -> 275 and 409 -> 396 us at 16x8x8), 14-unit 1x1 launches +1 %.
        int slot = 0;
        ES_STAMP_AT(1);
        __builtin_amdgcn_s_barrier();                                           // unit 0 published
        ES_STAMP_AT(2);
        ws_read_frags<MI>(smem, fragA, fragB, af, bfr);
        for (int ks = 0; ks + 1 < nloc; ++ks) {
            slot = slot == NS - 1 ? 0 : slot + 1;
            const char* const An = smem + slot * STAGE_BYTES;                     // slot of unit ks + 1
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bfr[j], acc[0][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // the last A fragment of unit ks (requested after the B's) is in
            __builtin_amdgcn_s_barrier();                                       // unit ks + 1 published; slot of unit ks released
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 1; i < MI; ++i) {
                af[i - 1] = *(const h8*)(An + fragA + (i - 1) * 1024);
                __builtin_amdgcn_sched_barrier(0);
                if (i < MI - 1) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        bfr[j] = *(const h8*)(An + fragB + j * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            af[MI - 1] = *(const h8*)(An + fragA + (MI - 1) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        ws_mma<MI>(acc, af, bfr);                                               // last unit
    }
    ES_STAMP_AT(3);
    conv_epilogue<BM_, NC_, true, true, EPI_, false, STATS_>(a, g, acc, smem, M, m0, n0, wave, lane, S, bz, ncdhw);
    ES_STAMP_AT(4);
}

template <int BM_, int NC_, int NP_, bool UP_ = false, int EPI_ = ES_EPI_NONE, bool STATS_ = false, int NS_ = 3>
__global__ __launch_bounds__(64 * (NC_ + NP_), 3) void k_conv_ws(const es_conv_args a, const ConvGeom g, int ncdhw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned long long* stamp = nullptr;
#ifdef ES_STAMP
    stamp = g_stamp_buf ? g_stamp_buf + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * (NC_ + NP_) + wave) * 8 : nullptr;
#endif
    ES_STAMP_AT(0);
    const long M = (long)g.O * g.D * g.H * g.W;
    int bx, by, bz;
    conv_tile_of(a, bx, by, bz);                 // XCD-aware, re-use-aware tile order
    const int S = gridDim.z;
    int ks_begin, ks_end;
    split_range(a.taps * (a.Cin >> 5), a.a2 ? (a.Cin2 >> 5) : 0, a.taps, bz, S, ks_begin, ks_end);
    conv_ws_tile<BM_, NC_, NP_, UP_, EPI_, STATS_, NS_>(a, g, ncdhw, smem, wave, lane, M, bx, by, ks_begin, ks_end, S, bz, stamp);
}

// ---------------------------------------------------------------------------------------------
// k_conv_ws3 (round 6): k_conv_ws<256, 8, 4> for 3x3x3 SAME convs with the A tile of a (channel chunk, kd, kh) group staged ONCE.
// The three K units of such a group are the taps kw = -1, 0, +1 of the same 32 channels: their A rows are the SAME voxel line shifted
// by one voxel along W, and a 16-row MFMA tile is one W line (W = 16; two / four lines at W = 8 / 4) -- so the A fragments of the
// kw = -1 / +1 units are the centre fragments moved by ONE LANE inside a 16-lane row (v_mov_dpp row_shr:1 / row_shl:1, zero fill at
// the ends of a line: exactly the zero padding of the convolution).  The producers load ONE centre A tile per group (LDS-DMA bytes
// per group 90 -> 58 KB), the consumers read A fragments once per group (LDS reads per wave and group 33 -> 25 KB) and derive the
// shifted operands in registers right before the MFMA row that uses them.  Why: per K unit the LDS moves 118 KB against 128 B /
// clock = 920 clocks, next to 896 clocks of MFMA issue -- both pipes are co-critical (profiles/r06_notes.md section 4).  Same
// products in the same order: bit-identical to k_conv_ws.  (Round 2 had tried the shared A tile with shifted LDS reads of a
// halo'd tile: +6 %; here the shift costs 8 VALU moves per fragment pair and no LDS access.)
// Ring: 4 B slots (one per K unit) + 2 A slots (one per group) = 96 KB, as k_conv_ws; one barrier per K unit.
// ---------------------------------------------------------------------------------------------
typedef int i4v __attribute__((ext_vector_type(4)));
template <int DIR>      // -1: lane w takes lane w - 1 (tap kw = -1), +1: lane w takes lane w + 1 (tap kw = +1); zero past the row's ends
__device__ __forceinline__ h8 a3_shift(const h8 v, const bool zero_lane) {
    i4v x = __builtin_bit_cast(i4v, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int s = DIR < 0 ? __builtin_amdgcn_update_dpp(0, x[e], 0x111, 0xf, 0xf, true) : __builtin_amdgcn_update_dpp(0, x[e], 0x101, 0xf, 0xf, true);
        x[e] = zero_lane ? 0 : s;
    }
    return __builtin_bit_cast(h8, x);
}

// GB_: ONE barrier per (chunk, kd, kh) group instead of one per K unit -- the producers issue a whole group (its A tile and three B
// tiles) right after the barrier that publishes the previous one; 7 B slots (the tile still being read + one group published + one
// landing) + 2 A slots = 144 KB.
template <bool STATS_, bool GB_ = false>
__global__ __launch_bounds__(768, 3) void k_conv_ws3(const es_conv_args a, const ConvGeom g) {
    constexpr int BM_ = 256, NC_ = 8, NP_ = 4, NSB = GB_ ? 7 : 4, NSA = 2, MI = 4;
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BNP * BK * 2, A_RING = NSB * B_BYTES;      // A slots behind the B ring
    constexpr int NA = (BM_ / 16) / NP_, NB = (BNP / 16) / NP_;
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long M = (long)g.O * g.D * g.H * g.W;
    int bx, by, bz;
    conv_tile_of(a, bx, by, bz);
    const int S = gridDim.z;
    const int kch0 = a.Cin >> 5, nks0 = 27 * kch0;
    int ks_begin, ks_end;
    split_range(nks0, 0, 27, bz, S, ks_begin, ks_end);            // (cuts in whole (chunk, kd, kh) groups: multiples of 3 K units)
    const int nloc = ks_end - ks_begin, ngrp = nloc / 3;
    const long m0 = (long)bx * BM_;
    const int n0 = by * BN;
    if (wave >= NC_) {
        // =============================== producer ===============================
        const int pw = wave - NC_;
        unsigned voff[NA], msk[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int p = (pw + NP_ * j) * 64 + lane;
            const int row = p >> 2;
            const int lc = (p & 3) ^ f_swz(row);
            const long m = m0 + row;
            const bool ok = m < M;
            const long mm = ok ? m : 0;
            const int w = (int)(mm & (g.W - 1)), h = (int)((mm >> g.lw) & (g.H - 1)), d = (int)((mm >> (g.lw + g.lh)) & (g.D - 1));
            const int o = (int)(mm >> (g.lw + g.lh + g.ld));
            voff[j] = (unsigned)(((((long)o * g.D + d) * g.H + h) * g.W + w) * a.Cin * 2 + lc * 16);
            msk[j] = ok ? tap_mask27(d, g.D, h, g.H, w, g.W) : 0u;
        }
        const int bias = ((g.H + 1) * g.W + 1) * a.Cin * 2;
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.a - bias), (short)0, (int)OOB, 0x00020000);
        const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)((const _Float16*)a.w + ((long)by * nks0) * (BNP * BK)), (short)0, (int)OOB, 0x00020000);
        int dtab;
        {
            const int t = lane < 27 ? lane : 13;
            const int kd = t / 9 - 1, kh = (t / 3) % 3 - 1, kw = t % 3 - 1;
            dtab = ((kd * g.H + kh) * g.W + kw) * a.Cin * 2 + bias;
        }
        int a_tap = ks_begin % 27, a_c = ks_begin / 27;               // first tap (kw = -1) and channel chunk of the next group to load
        unsigned b_off = (unsigned)ks_begin * (unsigned)B_BYTES;
        const unsigned voffB = (unsigned)lane * 16u;
        auto issue_A = [&](int slot) __attribute__((always_inline)) {
            char* dst = smem + A_RING + slot * A_BYTES;
            const unsigned sA = (unsigned)__builtin_amdgcn_readlane(dtab, a_tap + 1) + (unsigned)a_c * 64u;      // the group's centre tap (kw = 0)
            const unsigned sbit = 1u << (a_tap + 1);
#pragma unroll
            for (int j = 0; j < NA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(dst + (pw + NP_ * j) * 1024), 16, (int)((msk[j] & sbit) ? voff[j] : OOB), (int)sA, 0, 0);
            a_tap += 3;
            if (a_tap == 27) { a_tap = 0; ++a_c; }
        };
        auto issue_B = [&](int slot) __attribute__((always_inline)) {
            char* dst = smem + slot * B_BYTES;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int q = pw + NP_ * j;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(dst + q * 1024), 16, (int)(q * 16 >= BN ? OOB : voffB), (int)(b_off + (unsigned)q * 1024u), 0, 0);
            }
            b_off += (unsigned)B_BYTES;
        };
        if constexpr (GB_) {
            int sb = 0;
            auto issue_group = [&](int gi) __attribute__((always_inline)) {
                issue_A(gi & 1);
#pragma unroll
                for (int t = 0; t < 3; ++t) { issue_B(sb); sb = sb == NSB - 1 ? 0 : sb + 1; }
            };
            issue_group(0);
            for (int gi = 0; gi < ngrp; ++gi) {
                wait_vmcnt<0>();                     // group gi landed (it is the only one in flight)
                __builtin_amdgcn_s_barrier();        // group gi visible; every consumer is past the first two units of group gi - 1
                if (gi + 1 < ngrp) issue_group(gi + 1);
            }
            __builtin_amdgcn_s_barrier();            // (the consumers' barrier behind the last group)
            f4 dummy[MI][7];
            conv_epilogue<BM_, NC_, false, true, ES_EPI_NONE, false, STATS_>(a, g, dummy, smem, M, m0, n0, wave, lane, S, bz, 0);
            return;
        }
        issue_A(0);
        issue_B(0);
        if (nloc > 1) issue_B(1);
        int kw3 = 0, grp = 0;
        for (int u = 0; u < nloc; ++u) {
            // loads issued AFTER the ones unit u needs: B(u + 1), and the next group's A tile when u is not a group's first unit
            const int newer = (u + 1 < nloc ? NB : 0) + ((kw3 != 0 && grp + 1 < ngrp) ? NA : 0);
            if (newer >= NA + NB) wait_vmcnt<NA + NB>(); else if (newer >= NB) wait_vmcnt<NB>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();            // unit u (and its group's A tile) visible; the slots behind released
            if (u + 2 < nloc) issue_B((u + 2) % NSB);
            if (kw3 == 0 && grp + 1 < ngrp) issue_A((grp + 1) % NSA);
            if (++kw3 == 3) { kw3 = 0; ++grp; }
        }
        __builtin_amdgcn_s_barrier();                // (the consumers' barrier behind the last unit: their loop is straight-line)
        f4 dummy[MI][7];
        conv_epilogue<BM_, NC_, false, true, ES_EPI_NONE, false, STATS_>(a, g, dummy, smem, M, m0, n0, wave, lane, S, bz, 0);
        return;
    }
    // =============================== consumer ===============================
    static_assert(NA == NB, "the producer's wait ladder assumes equal A and B piece counts per wave");
    const int wm = wave >> 1, wn = wave & 1;
    f4 acc[MI][7];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, q = lane >> 4;
    const int fragA = A_RING + (wm * 64 + i16) * 64 + ((q ^ f_swz(i16)) << 4);
    const int fragB = (wn * 112 + i16) * 64 + ((q ^ f_swz(i16)) << 4);
    const bool zl = (i16 & (g.W - 1)) == 0, zr = (i16 & (g.W - 1)) == ((g.W - 1) & 15);      // first / last voxel of a W line
    // MFMA order: B column outer, A row inner -- one B fragment is live at a time (3 in a prefetch ring) instead of all 7, which pays for
    // the 4 shifted A fragments of the kw = -1 / +1 units inside the 168-register budget (112 accumulators + 16 centre + 16 shifted +
    // 12 B).  The current unit's B slot is therefore read until the END of the unit, past the barrier that publishes the next one:
    // the B ring has 4 slots (the producers refill slot u % 4 after barrier u + 2, when every consumer has left unit u).
    h8 afc[MI], bq[3];
    __builtin_amdgcn_s_barrier();                                           // unit 0 and group 0 published
    bq[0] = *(const h8*)(smem + fragB);
    bq[1] = *(const h8*)(smem + fragB + 1024);
#pragma unroll
    for (int i = 0; i < MI; ++i) afc[i] = *(const h8*)(smem + fragA + i * 1024);
    int slotB = 0;
    for (int grp = 0; grp < ngrp; ++grp) {
        const char* const An = smem + ((grp + 1) & 1) * A_BYTES;              // A slot of the NEXT group
        auto unit = [&](auto kwc) __attribute__((always_inline)) {
            constexpr int KW = decltype(kwc)::value;                         // 0, 1, 2 = taps kw -1, 0, +1
            // (straight-line: behind the LAST unit the "next" reads fetch stale LDS that nobody uses, and its barrier is matched by
            //  one extra barrier of the producers -- with `if (next unit exists)` around them the register allocator spilled 50 dwords)
            const char* const Bc = smem + slotB * B_BYTES;                   // this unit's B tile
            slotB = slotB == NSB - 1 ? 0 : slotB + 1;
            const char* const Bn = smem + slotB * B_BYTES;                   // the next unit's
            h8 ash[MI];
            if constexpr (KW == 0) {
#pragma unroll
                for (int i = 0; i < MI; ++i) ash[i] = a3_shift<-1>(afc[i], zl);
            } else if constexpr (KW == 2) {
#pragma unroll
                for (int i = 0; i < MI; ++i) ash[i] = a3_shift<1>(afc[i], zr);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                constexpr int C0 = KW * 7;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(KW == 1 ? afc[i] : ash[i], bq[(C0 + j) % 3], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j == 0 && (!GB_ || KW == 2)) __builtin_amdgcn_s_barrier();   // the next unit (after kw = +1: the next group's A tile) published; GB_: the next GROUP
                // the fragment of column j + 3 into the buffer column j has just released: two columns of MFMAs (of both SIMD partners) cover
                // the read (with "column j + 2 into the buffer of column j - 1" the compiler's waitcnt sat one column behind every read)
                // (the fragment of column j + 2 into the buffer of column j - 1.  Three columns ahead -- into the buffer column j has just
                //  released -- measured the same: 54.91 vs 54.84 steps/s, profiles/r06_conv_launch_ab.txt)
                if (j + 2 < 7) bq[(C0 + j + 2) % 3] = *(const h8*)(Bc + fragB + (j + 2) * 1024);
                else bq[(C0 + j + 2) % 3] = *(const h8*)(Bn + fragB + (j + 2 - 7) * 1024);
                if (KW == 2 && j >= 1 && j <= MI) afc[j - 1] = *(const h8*)(An + fragA + (j - 1) * 1024);     // (the shifted copies multiply; the centre registers are free)
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        unit(std::integral_constant<int, 0>{});
        unit(std::integral_constant<int, 1>{});
        unit(std::integral_constant<int, 2>{});
    }
    conv_epilogue<BM_, NC_, true, true, ES_EPI_NONE, false, STATS_>(a, g, acc, smem, M, m0, n0, wave, lane, S, bz, 0);
}

// ---------------------------------------------------------------------------------------------
// k_conv_kw (round 6): K split INSIDE a workgroup -- the small problems of the few-objects regime (a shard of 4 .. 16 objects, the
// transformer linears of every level there, the 16x4x4 level at any object count).  Such a launch has fewer 256-row tiles than the
// chip has CUs; until round 5 it kept the big tiles and split K over workgroups: every partial tile went to HBM as an fp32 slab and a
// second launch (k_conv_splitk_reduce) read S slabs per output -- 2.9 of the 4.7 ms of a 4-object step were such round trips and
// the lone K chains of the 64-row kernels (profiles/r05b_shard_emulation.txt).  Here a workgroup owns a 64 x (112 NCH_) tile -- enough
// tiles to cover the CUs without a cross-workgroup split -- and its consumer waves are KS_ K STREAMS x NCH_ column halves:
// stream s multiplies the s-th of KS_ contiguous K ranges (the cuts of split_range, i.e. exactly the ranges an S = KS_ split over
// workgroups has) into its own 64 x 112 accumulator tile; the streams' tiles meet in LDS and are added in stream order, then the
// ordinary epilogue (bias / per-object vector / residual / f16 copy / GEGLU) runs on the sum.  No slab, no reduction launch, a K chain
// KS_ times shorter -- and the SAME BITS as a split of S = KS_ over workgroups followed by k_conv_splitk_reduce (same ranges, same
// MFMA chains, same left fold, same epilogue order), which is what lets a bit-exact shard use it where the whole problem splits.
// Roles: KS_ x NCH_ consumer waves (the 64 x 112 wave tile of k_conv_ws: 112 accumulators), 8 producer waves = 8 / KS_ per stream,
// each stream with its own K-step generator; one barrier per STAGE = one K unit of every stream; 3-stage ring of KS_ x (4 KiB A +
// NCH_ x 7 KiB B).  gridDim = (row tiles of 64, column tiles of 112 NCH_, S): a cross-workgroup split on top (S > 1) writes partial
// slabs like every other conv kernel (virtual split index bz KS_ + s of S KS_).
// ---------------------------------------------------------------------------------------------
template <int KS_, int NCH_, bool UP_ = false, int EPI_ = ES_EPI_NONE>
__global__ __launch_bounds__(64 * (KS_ * NCH_ + 8), 3) void k_conv_kw(const es_conv_args a, const ConvGeom g, int ncdhw) {
    constexpr int BM_ = 64, NC_ = KS_ * NCH_, NP_ = 8, NPW = NP_ / KS_;            // producer waves per stream
    static_assert(NC_ == 4 && (KS_ == 2 || KS_ == 4), "4 consumer waves: 4 streams x 1 column half or 2 streams x 2");
    constexpr int NS = 3;
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = NCH_ * 112 * BK * 2, UNIT_BYTES = A_BYTES + B_BYTES, STAGE_BYTES = KS_ * UNIT_BYTES;
    constexpr int APIECES = BM_ / 16, BPIECES = NCH_ * 7, NPIECE = APIECES + BPIECES;
    constexpr int JA = APIECES / NPW, JMAX = (NPIECE + NPW - 1) / NPW;              // pieces j < JA of a producer wave are A pieces
    static_assert(APIECES % NPW == 0, "A pieces must divide over a stream's producer waves");
    constexpr int PART_FLOATS = 112 * 64;                                         // one consumer wave's accumulators, [register][lane]
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long M = (long)g.O * g.D * g.H * g.W;
    // tile mapping: XCD-contiguous ranges of logical ids, row tiles fastest (neighbours share the weight slab and the halo)
    int bx, byh, bz;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int nwg = gx * gy * (int)gridDim.z;
        const int orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        const int per_z = gx * gy;
        bz = L / per_z;
        const int t = L - bz * per_z;
        bx = t % gx;
        byh = t / gx;
    }
    const int by = NCH_ == 2 ? byh : (byh >> 1), half0 = NCH_ == 2 ? 0 : (byh & 1);      // 224-column weight tile, first 112-column half
    const int S = gridDim.z;
    const long m0 = (long)bx * BM_;
    const int kch0 = a.Cin >> 5;
    const int nks0 = a.taps * kch0;
    // stream s of this workgroup = virtual split index bz KS_ + s of S KS_
    int nstage = 0;
#pragma unroll
    for (int s = 0; s < KS_; ++s) {
        int b0, e0;
        split_range(nks0, a.a2 ? (a.Cin2 >> 5) : 0, a.taps, bz * KS_ + s, S * KS_, b0, e0);
        nstage = (e0 - b0) > nstage ? (e0 - b0) : nstage;
    }
    if (wave >= NC_) {
        // =============================== producer ===============================
        const int pw = wave - NC_, strm = pw / NPW, sub = pw - strm * NPW;
        int ks_begin, ks_end;
        split_range(nks0, a.a2 ? (a.Cin2 >> 5) : 0, a.taps, bz * KS_ + strm, S * KS_, ks_begin, ks_end);
        const int nloc = ks_end - ks_begin;
        const int npc = (NPIECE - sub + NPW - 1) / NPW;          // pieces of this wave per unit (JMAX or JMAX - 1)
        int a_lc[JA], a_o[JA], a_d[JA], a_h[JA], a_w[JA];
        bool a_ok[JA];
#pragma unroll
        for (int j = 0; j < JA; ++j) {
            const int p = (sub + NPW * j) * 64 + lane;   // 16-B slot of the A tile: row = p >> 2, physical chunk = p & 3
            const int row = p >> 2;
            a_lc[j] = (p & 3) ^ f_swz(row);
            const long m = m0 + row;
            a_ok[j] = m < M;
            const long mm = a_ok[j] ? m : 0;
            a_w[j] = (int)(mm & (g.W - 1));
            a_h[j] = (int)((mm >> g.lw) & (g.H - 1));
            a_d[j] = (int)((mm >> (g.lw + g.lh)) & (g.D - 1));
            a_o[j] = (int)(mm >> (g.lw + g.lh + g.ld));
        }
        int st_phase = ks_begin >= nks0 ? 1 : 0;
        int st_tap = st_phase ? 0 : ks_begin % a.taps;
        int st_c = st_phase ? (ks_begin - nks0) : (ks_begin / a.taps);
        int st_ntap = st_phase ? 1 : a.taps, st_kch = st_phase ? (a.Cin2 >> 5) : kch0;
        unsigned st_boff = (unsigned)(st_phase ? (ks_begin - nks0) : ks_begin) * (unsigned)(BNP * BK * 2);      // weight images: 16 KiB per K step
        unsigned voff[JA], msk[JA];
        int upm[JA][3], upp[JA][3];
        int dtab = 0;
        __amdgpu_buffer_rsrc_t rA, rB;
        auto set_phase = [&]() __attribute__((always_inline)) {
            const _Float16* Wg = (const _Float16*)(st_phase ? a.w2 : a.w);
            const long nks_ph = st_phase ? (long)(a.Cin2 >> 5) : (long)nks0;
            rB = __builtin_amdgcn_make_buffer_rsrc((void*)(Wg + ((long)by * nks_ph) * (BNP * BK)), (short)0, (int)OOB, 0x00020000);
            const _Float16* Ag = (const _Float16*)(st_phase ? a.a2 : a.a);
            const int Cin = st_phase ? a.Cin2 : a.Cin;
            const bool down = !st_phase && (a.mode == ES_CONV_DOWN_HW || a.mode == ES_CONV_DOWN_DHW);
            const bool downd = !st_phase && a.mode == ES_CONV_DOWN_DHW;
            const int Dsrc = downd ? 2 * g.D : g.D;
            const int Hi = st_phase ? g.H : g.Hi, Wi = st_phase ? g.W : g.Wi;
            const int ntap = st_phase ? 1 : a.taps;
            const bool updhw = UP_ && a.mode == ES_CONV_UP_DHW;
            const int Di = updhw ? g.D / 2 : g.D;
            const int bias = ntap == 27 ? ((Hi + 1) * Wi + 1) * Cin * 2 : 0;
            rA = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)Ag - bias), (short)0, (int)OOB, 0x00020000);
            {
                const int t = lane < 27 ? lane : 13;
                const int kd = t / 9 - 1, kh = (t / 3) % 3 - 1, kw = t % 3 - 1;
                dtab = ntap == 27 ? ((kd * Hi + kh) * Wi + kw) * Cin * 2 + bias : 0;
                if (UP_) dtab = ntap == 27 ? (updhw ? 0 : kd * Hi * Wi * Cin * 2) + bias : 0;
            }
#pragma unroll
            for (int j = 0; j < JA; ++j) {
                const int ch = down ? 2 * a_h[j] : a_h[j];
                const int cw = down ? 2 * a_w[j] : a_w[j];
                const int cd = downd ? 2 * a_d[j] : a_d[j];
                voff[j] = (unsigned)(((((long)a_o[j] * Dsrc + cd) * Hi + ch) * Wi + cw) * Cin * 2 + a_lc[j] * 16);
                if (UP_) {
                    const int sd = updhw ? a_d[j] >> 1 : a_d[j];
                    voff[j] = (unsigned)(((((long)a_o[j] * Di + sd) * Hi + (a_h[j] >> 1)) * Wi + (a_w[j] >> 1)) * Cin * 2 + a_lc[j] * 16);
                    const int SD = Hi * Wi * Cin * 2, SH = Wi * Cin * 2, SW = Cin * 2;
                    upm[j][0] = (updhw && !(a_d[j] & 1)) ? -SD : 0; upp[j][0] = (updhw && (a_d[j] & 1)) ? SD : 0;
                    upm[j][1] = !(a_h[j] & 1) ? -SH : 0;            upp[j][1] = (a_h[j] & 1) ? SH : 0;
                    upm[j][2] = !(a_w[j] & 1) ? -SW : 0;            upp[j][2] = (a_w[j] & 1) ? SW : 0;
                }
                unsigned m = 0;
                if (ntap == 1) m = a_ok[j] ? 1u : 0u;
                else m = a_ok[j] ? tap_mask27(cd, Dsrc, ch, UP_ ? g.H : Hi, cw, UP_ ? g.W : Wi) : 0u;
                msk[j] = m;
            }
        };
        set_phase();
        const unsigned voffB = (unsigned)lane * 16u;
        auto issue_unit = [&](char* dst) __attribute__((always_inline)) {        // dst: LDS base of this stream's unit (A tile, then B tile)
            const unsigned sA = (unsigned)__builtin_amdgcn_readlane(dtab, st_tap) + (unsigned)st_c * 64u;
            const unsigned sbit = 1u << st_tap;
            int ukd = 0, ukh = 0, ukw = 0;
            if (UP_) { ukd = st_tap / 9 - 1; ukh = (st_tap / 3) % 3 - 1; ukw = st_tap % 3 - 1; }
#pragma unroll
            for (int j = 0; j < JA; ++j) {
                unsigned vs = voff[j];
                if (UP_) {
                    vs += (unsigned)(ukd < 0 ? upm[j][0] : ukd > 0 ? upp[j][0] : 0);
                    vs += (unsigned)(ukh < 0 ? upm[j][1] : ukh > 0 ? upp[j][1] : 0);
                    vs += (unsigned)(ukw < 0 ? upm[j][2] : ukw > 0 ? upp[j][2] : 0);
                }
                const unsigned vo = (msk[j] & sbit) ? vs : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(dst + (sub + NPW * j) * 1024), 16, (int)vo, (int)sA, 0, 0);
            }
#pragma unroll
            for (int j = JA; j < JMAX; ++j) {
                const int qb = sub + NPW * j - APIECES;                           // 1 KiB piece (16 weight rows) of this tile's NCH_ x 7
                if (j < JMAX - 1 || qb < BPIECES)                                 // (wave-uniform: the last piece exists for the first waves only)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(dst + A_BYTES + qb * 1024), 16, (int)voffB,
                                                             (int)(st_boff + (unsigned)(half0 * 7 + qb) * 1024u), 0, 0);
            }
            st_boff += (unsigned)(BNP * BK * 2);
            if (++st_tap == st_ntap) {
                st_tap = 0;
                if (++st_c == st_kch && !st_phase && a.a2) {
                    st_phase = 1; st_c = 0; st_ntap = 1; st_kch = a.Cin2 >> 5; st_boff = 0;
                    set_phase();
                }
            }
        };
        auto wait_own = [&](bool more) __attribute__((always_inline)) {           // the pieces of the oldest unit in flight have landed
            if (!more) wait_vmcnt<0>();
            else if (npc == JMAX) wait_vmcnt<JMAX>();
            else wait_vmcnt<JMAX - 1>();
        };
        char* const base = smem + strm * UNIT_BYTES;
        if (nloc > 0) issue_unit(base);
        if (nloc > 1) issue_unit(base + STAGE_BYTES);
        for (int st = 0; st < nstage; ++st) {
            if (st < nloc) wait_own(st + 1 < nloc);                               // own pieces of unit st (a stream may be one unit shorter)
            __builtin_amdgcn_s_barrier();        // stage st visible to the consumers; the slot of stage st - 1 released by them
            if (st + 2 < nloc) issue_unit(base + ((st + 2) % NS) * STAGE_BYTES);
        }
        __builtin_amdgcn_s_barrier();            // (A) ring free
        __builtin_amdgcn_s_barrier();            // (B) stream tiles in LDS
        f4 dummy[1][7];
        if constexpr (NCH_ == 1) conv_epilogue<64, 8, false, true, EPI_>(a, g, dummy, smem, M, m0, 0, wave, lane, S, bz, ncdhw);
        else { f4 dummy2[2][7]; conv_epilogue<64, 4, false, true, EPI_>(a, g, dummy2, smem, M, m0, 0, wave, lane, S, bz, ncdhw); }
        return;
    }
    // =============================== consumer ===============================
    const int strm = wave / NCH_, hh = wave - strm * NCH_;
    int ks_begin, ks_end;
    split_range(nks0, a.a2 ? (a.Cin2 >> 5) : 0, a.taps, bz * KS_ + strm, S * KS_, ks_begin, ks_end);
    const int nloc = ks_end - ks_begin;
    constexpr int MI = 4;
    f4 acc[MI][7];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, q = lane >> 4;
    const int fragA = strm * UNIT_BYTES + i16 * 64 + ((q ^ f_swz(i16)) << 4);
    const int fragB = strm * UNIT_BYTES + A_BYTES + hh * (112 * 64) + i16 * 64 + ((q ^ f_swz(i16)) << 4);
    {
        h8 af[MI], bfr[7];
        int slot = 0, done = 0;                                                 // barriers passed so far
        if (nloc > 0) {
            __builtin_amdgcn_s_barrier();                                       // stage 0 published
            done = 1;
            ws_read_frags<MI>(smem, fragA, fragB, af, bfr);
            for (int ks = 0; ks + 1 < nloc; ++ks) {                             // pipelined as in k_conv_ws
                slot = slot == NS - 1 ? 0 : slot + 1;
                const char* const An = smem + slot * STAGE_BYTES;
#pragma unroll
                for (int j = 0; j < 7; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bfr[j], acc[0][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                ++done;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 1; i < MI; ++i) {
                    af[i - 1] = *(const h8*)(An + fragA + (i - 1) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                    if (i < MI - 1) {
#pragma unroll
                        for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
#pragma unroll
                        for (int j = 0; j < 7; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            bfr[j] = *(const h8*)(An + fragB + j * 1024);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                af[MI - 1] = *(const h8*)(An + fragA + (MI - 1) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
            ws_mma<MI>(acc, af, bfr);                                           // last unit of this stream
        }
        for (; done < nstage; ++done) __builtin_amdgcn_s_barrier();             // a shorter stream keeps the others' stage barriers company
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // (A) every wave is done with the ring
    {
        f4* P = (f4*)smem + (long)wave * (PART_FLOATS / 4);                    // this wave's tile, [MFMA tile][lane][4]: lane-linear 16-byte writes
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) P[(i * 7 + j) * 64 + lane] = acc[i][j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                // (B) stream tiles in LDS
    // wave (strm, hh) finishes rows [16 MI2 strm, 16 MI2 (strm + 1)) of column half hh: the streams' tiles added in stream order
    constexpr int MI2 = MI / KS_;
    f4 fin[MI2][7];
#pragma unroll
    for (int i2 = 0; i2 < MI2; ++i2)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int i = strm * MI2 + i2;
            f4 v = ((const f4*)smem)[(long)(0 * NCH_ + hh) * (PART_FLOATS / 4) + (i * 7 + j) * 64 + lane];
#pragma unroll
            for (int s = 1; s < KS_; ++s) v += ((const f4*)smem)[(long)(s * NCH_ + hh) * (PART_FLOATS / 4) + (i * 7 + j) * 64 + lane];
            fin[i2][j] = v;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // the ordinary epilogue on the sum: (KS_ 4, one column half) wave = 16 rows x 112 columns -> the <64, 8> geometry with wn = 0;
    // (KS_ 2, two halves) wave = 32 rows x 112 columns -> the <64, 4> geometry as it is
    if constexpr (NCH_ == 1) conv_epilogue<64, 8, true, true, EPI_>(a, g, fin, smem, M, m0, by * BN + half0 * 112, 2 * wave, lane, S, bz, ncdhw);
    else conv_epilogue<64, 4, true, true, EPI_>(a, g, fin, smem, M, m0, by * BN, wave, lane, S, bz, ncdhw);
}

// ---------------------------------------------------------------------------------------------
// k_linear_ws: k_conv_ws for the 1x1 / linear launches with several column tiles (qkv: 6, FeedForward: 16-24), where a tile has only
// 14-84 K units against ~23 us of per-tile cost outside the K loop (tools/microbench_epilogue.py).  A workgroup owns ONE row tile and
// walks NCB consecutive column tiles: the K units of all of them form one stream through the ring -- the producers' per-row set-up is
// done once, they run two units ahead across the column-tile boundary (only the weight descriptor changes), and the consumers'
// epilogue of tile c overlaps the loads of tile c+1 (the epilogue slabs live BEHIND the ring: no workgroup barrier in it).
// Same tile, LDS image, K order and arithmetic as k_conv_ws (results are bit-identical).
// ---------------------------------------------------------------------------------------------
template <int EPI_, int NS_ = 3>
__global__ __launch_bounds__(768, 3) void k_linear_ws(const es_conv_args a, const ConvGeom g, int ncb) {
    // ring depth NS_: 3; 4 (ES_LIN_RING=4, GEGLU variant only: its fp16 epilogue slabs leave the room) measured neutral
    constexpr int BM_ = 256, NC_ = 8, NP_ = 4, NS = NS_;
    constexpr int WROWS = BM_ / (NC_ / 2), MI = WROWS / 16;
    constexpr int A_BYTES = BM_ * BK * 2, B_BYTES = BNP * BK * 2, STAGE_BYTES = A_BYTES + B_BYTES, RING_BYTES = NS * STAGE_BYTES;
    constexpr int NA = (BM_ / 16) / NP_, NB = (BNP / 16) / NP_, NLOAD = NA + NB;
    constexpr unsigned OOB = 0x80000000u;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long M = (long)g.O * g.D * g.H * g.W;
#ifdef ES_STAMP
    // (tools/linear_stamps.py) 0 entry, then per column tile: 1 + 2 cb = K loop done, 2 + 2 cb = epilogue done (first three tiles)
    unsigned long long* stamp = g_stamp_buf ? g_stamp_buf + ((size_t)(blockIdx.x + gridDim.x * blockIdx.y) * (NC_ + NP_) + wave) * 8 : nullptr;
#endif
    ES_STAMP_AT(0);
    // tile mapping as conv_tile_of (XCD-contiguous ranges, column GROUPS fastest inside ~3 MiB weight panels)
    int bx, byg;
    {
        const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy;
        const int orig = blockIdx.x + gx * blockIdx.y;
        const int xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
        const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
        const long slab = (long)a.Cin * BN * 2 * ncb;
        int npanel = (int)(((long)gy * slab + (3L << 20) - 1) / (3L << 20));
        npanel = npanel < 1 ? 1 : (npanel > gy ? gy : npanel);
        const int Pw = (gy + npanel - 1) / npanel, full = gx * Pw;
        const int p = L / full, rr = L - p * full;
        const int w = (gy - p * Pw) < Pw ? (gy - p * Pw) : Pw;
        bx = rr / w;
        byg = p * Pw + (rr - bx * w);
    }
    const long m0 = (long)bx * BM_;
    const int by0 = byg * ncb;
    const int kch = a.Cin >> 5;                  // K units per column tile
    const int total = kch * ncb;
    if (wave >= NC_) {
        // =============================== producer ===============================
        const int pw = wave - NC_;
        unsigned voff[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const int p = (pw + NP_ * j) * 64 + lane;    // 16-B slot of the A tile: row = p >> 2, physical chunk = p & 3
            const int row = p >> 2;
            const long m = m0 + row;
            voff[j] = m < M ? (unsigned)(m * a.Cin * 2 + (((p & 3) ^ f_swz(row)) * 16)) : OOB;
        }
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.a), (short)0, (int)OOB, 0x00020000);
        const _Float16* Wg = (const _Float16*)a.w;
        int cb = 0, c = 0, islot = 0;
        __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Wg + (long)by0 * kch * (BNP * BK)), (short)0, (int)OOB, 0x00020000);
        const unsigned voffB = (unsigned)lane * 16u;
        auto issue = [&]() __attribute__((always_inline)) {
            char* dst = smem + __builtin_amdgcn_readfirstlane(islot) * STAGE_BYTES;
            const unsigned sA = (unsigned)__builtin_amdgcn_readfirstlane(c) * 64u;
            const unsigned sB = (unsigned)__builtin_amdgcn_readfirstlane(c) * (unsigned)B_BYTES;
#pragma unroll
            for (int j = 0; j < NA; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(dst + (pw + NP_ * j) * 1024), 16, (int)voff[j], (int)sA, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int q = pw + NP_ * j;
                const unsigned vB = q * 16 >= BN ? OOB : voffB;          // the zero rows padding the weight tile: zero fill, no traffic
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(dst + A_BYTES + q * 1024), 16, (int)vB, (int)(sB + (unsigned)q * 1024u), 0, 0);
            }
            islot = islot == NS - 1 ? 0 : islot + 1;
            if (++c == kch) {                    // next column tile: same rows, next weight slab
                c = 0; ++cb;
                rB = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(Wg + (long)(by0 + cb) * kch * (BNP * BK)), (short)0, (int)OOB, 0x00020000);
            }
        };
        issue();
        if (total > 1) issue();
        if (NS == 4 && total > 2) issue();
        for (int u = 0; u < total; ++u) {
            // own pieces of unit u have landed (units u + 1 .. u + NS - 2 may still be in flight)
            const int ahead = total - 1 - u < NS - 2 ? total - 1 - u : NS - 2;
            if (ahead >= 2) wait_vmcnt<2 * NLOAD>(); else if (ahead == 1) wait_vmcnt<NLOAD>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();        // unit u visible to the consumers; the slot NS - 1 behind released by them
            if (u + NS - 1 < total) issue();
        }
        return;
    }
    // =============================== consumer ===============================
    const int wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, q = lane >> 4;
    const int fragA = (wm * WROWS + i16) * 64 + ((q ^ f_swz(i16)) << 4);
    const int fragB = A_BYTES + (wn * 112 + i16) * 64 + ((q ^ f_swz(i16)) << 4);
    int slot = 0;
    for (int cb = 0; cb < ncb; ++cb) {
        f4 acc[MI][7];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
        h8 af[MI], bfr[7];
        __builtin_amdgcn_s_barrier();                                           // first unit of this column tile published
        ws_read_frags<MI>(smem + slot * STAGE_BYTES, fragA, fragB, af, bfr);
        slot = slot == NS - 1 ? 0 : slot + 1;
        for (int ks = 0; ks + 1 < kch; ++ks) {                                  // pipelined as in k_conv_ws
            const char* const An = smem + slot * STAGE_BYTES;
            slot = slot == NS - 1 ? 0 : slot + 1;
#pragma unroll
            for (int j = 0; j < 7; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0], bfr[j], acc[0][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 1; i < MI; ++i) {
                af[i - 1] = *(const h8*)(An + fragA + (i - 1) * 1024);
                __builtin_amdgcn_sched_barrier(0);
                if (i < MI - 1) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        bfr[j] = *(const h8*)(An + fragB + j * 1024);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            af[MI - 1] = *(const h8*)(An + fragA + (MI - 1) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        ws_mma<MI>(acc, af, bfr);
#ifdef ES_STAMP
        if (cb < 3) ES_STAMP_AT(1 + 2 * cb);
#endif
        conv_epilogue<BM_, NC_, true, true, EPI_, true>(a, g, acc, smem + RING_BYTES, M, m0, (by0 + cb) * BN, wave, lane, 1, 0, 0);
#ifdef ES_STAMP
        if (cb < 3) ES_STAMP_AT(2 + 2 * cb);
#endif
    }
}

// ---------------------------------------------------------------------------------------------
// Self-attention, flash style, fp16 MFMA.  One workgroup = 64 query rows of one (batch, head);
// 4 waves x 16 rows.  K tile [64 keys][dp], V tile transposed [dp][64 keys] in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int AT_K = 64;

// DP: padded head dim 32, 64, 96 or 256 (VQ-VAE AttnBlock: one head of 256).  A wave owns NRT x 16 query rows, a workgroup
// NWV waves: 128 rows for long sequences (every K / V fragment read from LDS feeds NRT MFMAs; three workgroups per CU).
//
// Both products are computed TRANSPOSED so that the probabilities never leave the registers:
//   S^T = K Q^T   (A = K fragment, B = Q fragment): a lane holds S^T[key = t*16 + q*4 + r][row = i16] -- ONE query row per lane,
//                 16 of its 64 keys; the row max / sum are 16 in-lane operations + two cross-lane steps (xor 16, 32);
//   O^T = V^T P^T (A = V^T fragment, B = P^T): MFMA k-slot (q, e) of chunk kc is fed with key kc*32 + (e < 4 ? 0 : 16) + q*4 + (e & 3)
//                 on BOTH operands -- exactly the keys the lane already holds for its row (the contraction does not care about
//                 the order of the keys), so P^T is a register pack, and V^T comes as two transposing 8-byte LDS reads
//                 (ds_read_b64_tr_b16) of the row-major V tile per fragment.
//   The lane then holds O[row = i16][d = j*16 + q*4 + r]: the online-softmax rescale is a per-lane scalar and the output store
//   is 8 bytes per lane.
// Round 1 / early round 2 wrote P to LDS in A-fragment order (16 ds_write_b16 + 2 ds_read_b128 per wave and K tile -- as many
// LDS cycles as all K / V fragment reads together; rocprofv3: LDS active 0.57 of CU-cycles, MFMA busy 0.11).
template <int DP, int NWV, int NRT, int MINW = (DP <= 64 ? 3 : 2)>
__global__ __launch_bounds__(64 * NWV, MINW) void k_attention(const es_attn_args a) {   // DP <= 64: <= 168 registers, three 4-wave workgroups per CU
    constexpr int AT_Q = 16 * NRT * NWV;
    constexpr int KLD = DP + 8;              // halfs; +8 keeps 16-B alignment and skews banks
    __shared__ __attribute__((aligned(16))) _Float16 Ks[AT_K * KLD];
    __shared__ __attribute__((aligned(16))) _Float16 Vs[AT_K * KLD];      // V row-major like K; transposed by the read (below)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q4 = lane >> 4;
    // XCD-aware order (hardware workgroup id b runs on XCD b % 8; same bijective remap as conv_tile_of): each XCD owns a contiguous
    // range of (batch*head, row tile) pairs with the row tile fastest, so the workgroups that share one K / V (229 KB at 1024 tokens)
    // follow each other on ONE XCD and hit its L2.  With the row tiles spread over the 8 XCDs every L2 pulled its own copy:
    // rocprofv3 FETCH_SIZE 323 MB per call (x2-corrected) against 88 MB of qkv.
    const int nx = gridDim.x, nwg = nx * (int)gridDim.y;
    const int orig = blockIdx.x + nx * blockIdx.y;
    const int xcd = orig & 7, xq = nwg >> 3, xr = nwg & 7;
    const int L = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (orig >> 3);
    const int bh = L / nx, rtile = L - bh * nx;
    const int b = bh / a.heads, h = bh - b * a.heads;
    const int C = a.heads * a.dhead, ldq = 3 * C;
    const _Float16* base = (const _Float16*)a.qkv + (long)b * a.Ntok * ldq + h * a.dhead;
    const int q0 = rtile * AT_Q + wave * (16 * NRT);

    // Q fragments: lane (i16, q4) holds d = kc*32 + q4*8 .. +7 of query row q0 + rt*16 + i16 (the B operand of S^T)
    h8 qf[NRT][DP / 32];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
        for (int kc = 0; kc < DP / 32; ++kc) {
            const int d0 = kc * 32 + q4 * 8, row = q0 + rt * 16 + i16;
            const _Float16* p = base + (long)row * ldq + d0;
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[rt][kc][e] = (d0 + e < a.dhead && row < a.Ntok) ? p[e] : (_Float16)0.f;
        }
    f4 oacc[NRT][DP / 16];
    float mrow[NRT], lrow[NRT];
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        mrow[rt] = -INFINITY; lrow[rt] = 0.f;
#pragma unroll
        for (int j = 0; j < DP / 16; ++j) oacc[rt][j] = f4{0.f, 0.f, 0.f, 0.f};
    }

    // K / V tiles are fetched one tile AHEAD into registers and written to LDS after the current tile's readers are done: the
    // global-load round trip used to sit in front of every tile.  Items are 8 bytes with the CHANNEL chunk fastest across lanes:
    // a wave's load covers whole key rows (coalesced) and both tiles are stored row-major -- V is not transposed on the way in
    // (the first versions scattered it with four ds_write_b16 per item, which forced lanes along keys and 64 cache lines per load
    // instruction); the PV step reads it through ds_read_b64_tr_b16 instead.
    constexpr int NT = 64 * NWV, CPR = DP / 4, NITEM = AT_K * CPR, NIT = (NITEM + NT - 1) / NT;
    constexpr bool PREFETCH = DP <= 96;                      // (DP = 256, the VQ-VAE block: 16 items per thread would not fit; load in place)
    h4 kreg[NIT], vreg[NIT];
    // Buffer loads whose descriptor ends behind the last key row: an item past Ntok, past the head's channels or past the tile returns
    // zeros by the range check -- no compare / select per item and tile (they were ~75 of the ~330 instructions of a K tile).
    unsigned kvoff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NT;
        const int key = idx / CPR, d = (idx - key * CPR) * 4;
        kvoff[it] = (idx < NITEM && d < a.dhead) ? (unsigned)(key * ldq + d) * 2u : 0x80000000u;
    }
    const int nrec = (int)(((long)(a.Ntok - 1) * ldq + a.dhead) * 2);
    const __amdgpu_buffer_rsrc_t rK = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(base + C), (short)0, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(base + 2 * C), (short)0, nrec, 0x00020000);
    auto gload = [&](int k0) __attribute__((always_inline)) {
        const unsigned koff = (unsigned)k0 * (unsigned)ldq * 2u;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            kreg[it] = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(rK, (int)(kvoff[it] + koff), 0, 0));
            vreg[it] = __builtin_bit_cast(h4, __builtin_amdgcn_raw_buffer_load_b64(rV, (int)(kvoff[it] + koff), 0, 0));
        }
    };
    // (a second K / V buffer with ONE barrier per tile was measured in round 4: 115 us against 113 at 1024 tokens -- no gain, not kept)
    if (PREFETCH) gload(0);
    for (int k0 = 0; k0 < a.Ntok; k0 += AT_K) {
        __syncthreads();                                      // the previous tile's fragment reads are done
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * NT;
            if (idx < NITEM) {
                const int key = idx / CPR, d = (idx - key * CPR) * 4;
                h4 kv = kreg[PREFETCH ? it : 0], vv = vreg[PREFETCH ? it : 0];
                if (!PREFETCH) {                             // load in place, one item at a time
                    kv = h4{0, 0, 0, 0}; vv = h4{0, 0, 0, 0};
                    if (d < a.dhead && k0 + key < a.Ntok) {
                        const _Float16* p = base + (long)(k0 + key) * ldq + d;
                        kv = *(const h4*)(p + C);
                        vv = *(const h4*)(p + 2 * C);
                    }
                }
                *(h4*)&Ks[key * KLD + d] = kv;
                *(h4*)&Vs[key * KLD + d] = vv;
            }
        }
        __syncthreads();
        if (PREFETCH && k0 + AT_K < a.Ntok) gload(k0 + AT_K); // in flight under this tile's MFMAs
        // S^T = K Q^T (64 keys x 16 rows per row tile)
        f4 s[NRT][4];
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
            for (int t = 0; t < 4; ++t) s[rt][t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kc = 0; kc < DP / 32; ++kc) {
                const h8 kf = *(const h8*)&Ks[(t * 16 + i16) * KLD + kc * 32 + q4 * 8];
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) s[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[rt][kc], s[rt][t], 0, 0, 0);
            }
        // online softmax of the lane's row, in the exp2 domain (scale * log2 e folded into one multiply); lane holds keys
        // k0 + t*16 + q4*4 + r; only the last tile can hold keys >= Ntok
        // The kernel is VALU-bound here (~130 scalar-width operations per row tile and K tile against 16 MFMAs): the row maximum
        // is taken over the RAW scores (scale > 0), scale * log2 e and the subtraction of the maximum are one packed FMA per two
        // keys, sums and the rescale of O are packed too, and O is only rescaled when some row's maximum moved.
        h8 pf[NRT][2];
        typedef float f2 __attribute__((ext_vector_type(2)));
        const float c2 = a.scale * 1.44269504088896340736f;
        // (the mask of the keys past Ntok sits behind a real branch: written as `if (ragged && key >= Ntok)` inside the loops below it
        //  was if-converted into 16 compares + 16 selects per row tile in EVERY K tile -- a third of the instructions of this phase,
        //  for a case only the last tile of a ragged sequence can meet)
        if (k0 + AT_K > a.Ntok) {                            // wave-uniform
            asm volatile("" ::: "memory");
#pragma unroll
            for (int rt = 0; rt < NRT; ++rt)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = k0 + t * 16 + q4 * 4 + r >= a.Ntok ? -INFINITY : s[rt][t][r];
                        asm volatile("" : "+v"(v));           // (not speculatable: keeps the selects inside the branch)
                        s[rt][t][r] = v;
                    }
        }
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            float mx = -INFINITY;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[rt][t][r]);
            {                                                // over the four 16-lane rows (lane ^ 16, lane ^ 32), no LDS round trip
                float e, o;
                es_pair16(mx, e, o); mx = fmaxf(e, o);
                es_pair32(mx, e, o); mx = fmaxf(e, o);
            }
            const float mnew = fmaxf(mrow[rt], mx * c2);     // running maximum, exp2 domain
            const float alpha = __builtin_amdgcn_exp2f(mrow[rt] - mnew);
            const f2 c22 = {c2, c2}, nm2 = {-mnew, -mnew};
            f2 ps2 = {0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const f2 v = {s[rt][t][2 * hf], s[rt][t][2 * hf + 1]};
                    const f2 e = __builtin_elementwise_fma(v, c22, nm2);
                    const f2 pp = {__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])};
                    ps2 += pp;
                    pf[rt][t >> 1][(t & 1) * 4 + 2 * hf] = (_Float16)pp[0];
                    pf[rt][t >> 1][(t & 1) * 4 + 2 * hf + 1] = (_Float16)pp[1];
                }
            float ps = ps2[0] + ps2[1];
            {
                float e, o;
                es_pair16(ps, e, o); ps = e + o;
                es_pair32(ps, e, o); ps = e + o;
            }
            lrow[rt] = lrow[rt] * alpha + ps;
            mrow[rt] = mnew;
            if (__ballot(alpha != 1.0f) != 0) {              // some row's maximum moved: rescale O (wave-uniform branch)
                const f2 al2 = {alpha, alpha};
#pragma unroll
                for (int j = 0; j < DP / 16; ++j) {
                    f2 lo = {oacc[rt][j][0], oacc[rt][j][1]}, hi = {oacc[rt][j][2], oacc[rt][j][3]};
                    lo *= al2; hi *= al2;
                    oacc[rt][j] = f4{lo[0], lo[1], hi[0], hi[1]};
                }
            }
        }
        // O^T += V^T P^T
#pragma unroll
        for (int kc = 0; kc < AT_K / 32; ++kc)
#pragma unroll
            for (int j = 0; j < DP / 16; ++j) {
                // ds_read_b64_tr_b16 (tools/probes/probe_ds_read_tr.hip): within a 16-lane group lane i supplies the address of 4
                // contiguous halfs, lane c receives element c % 4 of the chunks addressed by lanes c / 4, 4 + c / 4, 8 + c / 4, 12 + c / 4.
                // Lane i = 4 jj + m points at V[key kb + jj][d = 16 j + 4 m ..]; lane c then holds V[kb + 0..3][16 j + c]: four keys
                // of ITS channel -- the V^T fragment, with no transposed copy of V anywhere.
                typedef short s4v __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(3))) s4v* lds_s4;
                const _Float16* vp = &Vs[(kc * 32 + q4 * 4 + (i16 >> 2)) * KLD + j * 16 + (i16 & 3) * 4];
                const h4 v0 = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)vp));
                const h4 v1 = __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(vp + 16 * KLD)));
                const h8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int rt = 0; rt < NRT; ++rt) oacc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[rt][kc], oacc[rt][j], 0, 0, 0);
            }
    }
    // write O / l : lane holds O[row = i16 of the row tile][d = j*16 + q4*4 + r]
    _Float16* out = (_Float16*)a.out_f16 + (long)b * a.Ntok * C + h * a.dhead;
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
        const int row = q0 + rt * 16 + i16;
        if (row >= a.Ntok) continue;
        const float inv = 1.0f / lrow[rt];
#pragma unroll
        for (int j = 0; j < DP / 16; ++j) {
            const int d = j * 16 + q4 * 4;
            if (d < a.dhead) {
                const f4 o = oacc[rt][j] * inv;
                const h4 hv = {(_Float16)o[0], (_Float16)o[1], (_Float16)o[2], (_Float16)o[3]};
                *(h4*)&out[(long)row * C + d] = hv;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Direct 3x3x3 conv for N <= 4 output channels (final eps conv 224->3, VQ-VAE conv_out 64->1):
// an MFMA tile would be >98 % padding.  One thread per voxel, weights [N][27][Cin] f16 in LDS.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_small_n(const es_conv_args a, const ConvGeom g, int ncdhw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* wl = (_Float16*)smem;                                  // [N][27][Cin]
    const int wn = a.N * 27 * a.Cin;
    for (int i = threadIdx.x * 8; i < wn; i += 256 * 8) *(h8*)(wl + i) = *(const h8*)((const _Float16*)a.w + i);
    __syncthreads();
    const long M = (long)g.O * g.D * g.H * g.W;
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int w_ = (int)(m & (g.W - 1)), h_ = (int)((m >> g.lw) & (g.H - 1)), d_ = (int)((m >> (g.lw + g.lh)) & (g.D - 1));
    const long o = m >> (g.lw + g.lh + g.ld);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const _Float16* A = (const _Float16*)a.a;
    for (int tap = 0; tap < 27; ++tap) {
        const int id = d_ + tap / 9 - 1, ih = h_ + (tap / 3) % 3 - 1, iw = w_ + tap % 3 - 1;
        if (id < 0 || id >= g.D || ih < 0 || ih >= g.H || iw < 0 || iw >= g.W) continue;
        const _Float16* src = A + (((o * g.D + id) * g.H + ih) * g.W + iw) * a.Cin;
        for (int c = 0; c < a.Cin; c += 8) {
            const h8 x = *(const h8*)(src + c);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (n < a.N) {
                    const h8 wv = *(const h8*)(wl + (n * 27 + tap) * a.Cin + c);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[n] += (float)x[e] * (float)wv[e];
                }
            }
        }
    }
    const long V = (long)g.D * g.H * g.W;
    for (int n = 0; n < a.N; ++n) {
        float v = acc[n] + (a.bias ? a.bias[n] : 0.f);
        if (ncdhw) a.out_f32[(o * a.N + n) * V + (m - o * V)] = v;
        else a.out_f32[m * a.out_ld + n] = v;
    }
}

// LDS-tiled version for volumes whose sides are multiples of 8 (the VQ-VAE's conv_out 64 -> 1 at 64^3): a workgroup owns an
// 8x8x8 block of output voxels, stages its 10x10x10 halo (out-of-volume voxels as zeros) ONCE -- voxel stride Cin*2 + 16 bytes, which
// spreads the 16 lanes of a ds_read_b128 group over all 64 banks -- and every thread accumulates 2 outputs with v_dot2_f32_f16
// (fp32 accumulate).  The direct kernel above re-reads every input voxel 27 times through L2: 5.5 ms per call at 8 x 64^3 x 64
// channels, 30 % of a scene's VQ-VAE decode.
__global__ __launch_bounds__(256) void k_conv_small_n_tiled(const es_conv_args a, const ConvGeom g, int ncdhw) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int Cin = a.Cin, c8n = Cin >> 3, vstride = Cin * 2 + 16;
    char* xt = smem;                                                   // [10][10][10] voxels x vstride bytes
    _Float16* wl = (_Float16*)(smem + 1000 * vstride);                 // [N][27][Cin]
    const int wn = a.N * 27 * Cin;
    for (int i = threadIdx.x * 8; i < wn; i += 256 * 8) *(h8*)(wl + i) = *(const h8*)((const _Float16*)a.w + i);
    const int tw = g.W >> 3, th = g.H >> 3, td = g.D >> 3;
    int t = blockIdx.x;
    const int bw = t % tw; t /= tw;
    const int bh = t % th; t /= th;
    const int bd = t % td;
    const long o = t / td;
    const _Float16* A = (const _Float16*)a.a + o * (long)g.D * g.H * g.W * Cin;
    for (int i = threadIdx.x; i < 1000 * c8n; i += 256) {
        const int c8 = i % c8n, v = i / c8n;
        const int lw = v % 10, lh = (v / 10) % 10, ld_ = v / 100;
        const int iw = bw * 8 + lw - 1, ih = bh * 8 + lh - 1, id = bd * 8 + ld_ - 1;
        h8 x = {0, 0, 0, 0, 0, 0, 0, 0};
        if (iw >= 0 && iw < g.W && ih >= 0 && ih < g.H && id >= 0 && id < g.D)
            x = *(const h8*)(A + (((long)id * g.H + ih) * g.W + iw) * Cin + c8 * 8);
        *(h8*)(xt + v * vstride + c8 * 16) = x;
    }
    __syncthreads();
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // thread -> output voxels (lw = tid & 7, lh = (tid >> 3) & 7, ld = tid >> 6 and ld + 4)
    const int lw = threadIdx.x & 7, lh = (threadIdx.x >> 3) & 7, ld0 = threadIdx.x >> 6;
    for (int tap = 0; tap < 27; ++tap) {
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
        const char* p0 = xt + (((ld0 + kd) * 10 + lh + kh) * 10 + lw + kw) * vstride;
        const char* p1 = p0 + 400 * vstride;                           // ld0 + 4
        for (int c8 = 0; c8 < c8n; ++c8) {
            const h8 x0 = *(const h8*)(p0 + c8 * 16), x1 = *(const h8*)(p1 + c8 * 16);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (n < a.N) {
                    const h8 wv = *(const h8*)(wl + (n * 27 + tap) * Cin + c8 * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const h2v w2 = {wv[2 * e], wv[2 * e + 1]};
                        acc[0][n] = __builtin_amdgcn_fdot2(h2v{x0[2 * e], x0[2 * e + 1]}, w2, acc[0][n], false);
                        acc[1][n] = __builtin_amdgcn_fdot2(h2v{x1[2 * e], x1[2 * e + 1]}, w2, acc[1][n], false);
                    }
                }
            }
        }
    }
    const long V = (long)g.D * g.H * g.W;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const long vox = (((long)(bd * 8 + ld0 + 4 * k) * g.H) + bh * 8 + lh) * g.W + bw * 8 + lw;
        for (int n = 0; n < a.N; ++n) {
            const float v = acc[k][n] + (a.bias ? a.bias[n] : 0.f);
            if (ncdhw) a.out_f32[(o * a.N + n) * V + vox] = v;
            else a.out_f32[(o * V + vox) * a.out_ld + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_conv_n16: 3x3x3 conv with N <= 16 output channels from a WIDE input (the UNet's output conv 224 -> 3, out.2 of
// openai_model_3d.py:735-739).  On the 224-column MFMA tile this launch is 98.7 % padding and 265 us (1.4 % of the step): a
// workgroup there streams one gathered 256 x 32 A tile per (chunk, tap) -- the same voxels 27 times -- for ONE useful MFMA column.
// Here a workgroup owns a 4 x 4 x 16 block of output voxels and stages its 6 x 6 x 18 halo (out-of-volume voxels = zero fill of the
// LDS-DMA) ONCE per 32-channel chunk: 2.5x the block instead of 27x; the 27 taps are 27 shifted fragment reads of that LDS image.
// B = the first sixteen rows (1 KiB, already swizzled) of the K step's block in the ordinary tiled weight image.  Wave w owns depth
// slice w: four 16-voxel MFMA row tiles (one per h row), K order = (chunk outer, tap inner) as in the tile kernels -- same bits.
// ---------------------------------------------------------------------------------------------
constexpr int N16_ROWS = 6 * 6 * 18;                       // halo'd voxels of a 4 x 4 x 16 block
constexpr int N16_A_BYTES = 44 * 1024, N16_B_BYTES = 28 * 1024;

__global__ __launch_bounds__(256, 2) void k_conv_n16(const es_conv_args a, const ConvGeom g) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + N16_A_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tw = g.W >> 4, th = g.H >> 2, td = g.D >> 2;
    int t = blockIdx.x;
    const int bw = t % tw; t /= tw;
    const int bh = t % th; t /= th;
    const int bd = t % td;
    const int o = t / td;
    const int Cin = a.Cin, kch = Cin >> 5;
    // ---- staging roles: piece p = wave + 4 j (j < 11) covers LDS bytes [p KiB, p KiB + 1 KiB): 16 halo'd voxels x 64 B
    constexpr int NPA = 11, NPB = 7;
    unsigned voff[NPA];
#pragma unroll
    for (int j = 0; j < NPA; ++j) {
        const int piece = wave + 4 * j;
        const int slot = piece * 64 + lane;                  // 16-B slot of the image: row = slot >> 2, physical chunk = slot & 3
        const int row = slot >> 2;
        const int lc = (slot & 3) ^ f_swz(row);
        const int hw = row % 18, hh = (row / 18) % 6, hd = row / 108;
        const int d = bd * 4 - 1 + hd, h = bh * 4 - 1 + hh, w = bw * 16 - 1 + hw;
        const bool ok = row < N16_ROWS && d >= 0 && d < g.D && h >= 0 && h < g.H && w >= 0 && w < g.W;
        voff[j] = ok ? (unsigned)((((o * g.D + d) * g.H + h) * g.W + w) * Cin * 2 + lc * 16) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.a), (short)0, (int)OOB, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(a.w), (short)0, (int)OOB, 0x00020000);
    const int i16 = lane & 15, q = lane >> 4;
    const int rowbase = (wave * 6) * 18 + i16;              // LDS row of (depth slice wave, h 0, w i16) at tap (0, 0, 0)
    f4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
    const int fragB = i16 * 64 + ((q ^ f_swz(i16)) << 4);
    for (int c = 0; c < kch; ++c) {
        if (c) __syncthreads();                              // every wave is done with the previous chunk's image
#pragma unroll
        for (int j = 0; j < NPA; ++j) {
            const int piece = wave + 4 * j;
            if (piece < 44)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr)(As + piece * 1024), 16, (int)voff[j], (int)((unsigned)c * 64u), 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NPB; ++j) {
            const int tap = wave + 4 * j;                    // 27 pieces: the first 16 rows of the K step's weight block
            if (tap < 27)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_ptr)(Bs + tap * 1024), 16, (int)((unsigned)lane * 16u),
                                                         (int)((unsigned)(c * 27 + tap) * (unsigned)(BNP * BK * 2)), 0, 0);
        }
        wait_vmcnt<0>();
        __syncthreads();
#pragma unroll 3
        for (int tap = 0; tap < 27; ++tap) {
            const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
            const h8 bf = *(const h8*)(Bs + tap * 1024 + fragB);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rowbase + ((kd * 6) + kh + i) * 18 + kw;
                const h8 af = *(const h8*)(As + row * 64 + ((q ^ f_swz(row)) << 4));
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc[i], 0, 0, 0);
            }
        }
    }
    // D layout: lane holds D[voxel = q*4 + r of the row tile][column = i16]; NCDHW fp32 output (+ bias)
    if (i16 < a.N) {
        const float bias = a.bias ? a.bias[i16] : 0.f;
        const long V = (long)g.D * g.H * g.W;
        float* outp = a.out_f32 + ((long)o * a.N + i16) * V;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long vox = ((long)(bd * 4 + wave) * g.H + bh * 4 + i) * g.W + bw * 16 + q * 4;
            *(f4*)(outp + vox) = f4{acc[i][0] + bias, acc[i][1] + bias, acc[i][2] + bias, acc[i][3] + bias};
        }
    }
}

// ---------------------------------------------------------------------------------------------
// VQ nearest-codebook lookup (quantizer.py:68-119): d_j = |z|^2 + |e_j|^2 - 2 z.e_j, argmin_j
// (first minimum), output = lut[argmin] written as channels-last f16 padded to Cpad.  lut is the
// codebook already passed through post_quant_conv (a 1x1x1 conv, folded on the host).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vq_lookup(const es_vq_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* E = (float*)smem;                     // [n_embed][4]: e0,e1,e2,|e|^2
    for (int j = threadIdx.x; j < a.n_embed; j += 256) {
        const float e0 = a.codebook[j * 3], e1 = a.codebook[j * 3 + 1], e2 = a.codebook[j * 3 + 2];
        E[j * 4] = e0; E[j * 4 + 1] = e1; E[j * 4 + 2] = e2;
        E[j * 4 + 3] = e0 * e0 + e1 * e1 + e2 * e2;          // torch.sum(E**2, dim=1)
    }
    __syncthreads();
    const long M = (long)a.O * a.V;
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const long o = m / a.V, v = m - o * a.V;
    const float z0 = a.z[(o * 3 + 0) * a.V + v], z1 = a.z[(o * 3 + 1) * a.V + v], z2 = a.z[(o * 3 + 2) * a.V + v];
    const float zz = z0 * z0 + z1 * z1 + z2 * z2;
    float best = INFINITY;
    int bi = 0;
    for (int j = 0; j < a.n_embed; ++j) {
        const f4 e = *(const f4*)&E[j * 4];
        const float dot = fmaf(z2, e[2], fmaf(z1, e[1], z0 * e[0]));
        const float d = (zz + e[3]) - 2.0f * dot;
        if (d < best) { best = d; bi = j; }
    }
    if (a.idx_out) a.idx_out[m] = bi;
    _Float16* out = (_Float16*)a.out_f16 + m * a.Cpad;
    for (int c = 0; c < a.Cpad; ++c) out[c] = c < 3 ? (_Float16)a.lut[bi * 3 + c] : (_Float16)0.f;
}


// split-K reduction + epilogue: out = sum_z part[z] (fixed order) + bias + rowvec + res
// The S slab loads of an element are independent: they are issued four at a time and added in slab order (with the slab count
// only known at run time the plain loop waited for one load after the other -- 8-16 dependent round trips at the 16x4x4 level).
__global__ __launch_bounds__(256) void k_conv_splitk_reduce(const es_conv_args a, long M, int V, int S) {
    const long n4 = M * (a.N >> 2);
    const int N4 = a.N >> 2;
    const long MN = M * a.N;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long m = i / N4;
        const int n = (int)(i - m * N4) * 4;
        const float* p = (const float*)a.workspace + m * a.N + n;
        f4 v = *(const f4*)p;
        int z = 1;
        for (; z + 3 < S; z += 4) {
            const f4 t0 = *(const f4*)(p + (long)z * MN), t1 = *(const f4*)(p + (long)(z + 1) * MN),
                     t2 = *(const f4*)(p + (long)(z + 2) * MN), t3 = *(const f4*)(p + (long)(z + 3) * MN);
            v += t0; v += t1; v += t2; v += t3;
        }
        if (z + 2 < S) {
            const f4 t0 = *(const f4*)(p + (long)z * MN), t1 = *(const f4*)(p + (long)(z + 1) * MN), t2 = *(const f4*)(p + (long)(z + 2) * MN);
            v += t0; v += t1; v += t2;
        } else if (z + 1 < S) {
            const f4 t0 = *(const f4*)(p + (long)z * MN), t1 = *(const f4*)(p + (long)(z + 1) * MN);
            v += t0; v += t1;
        } else if (z < S) {
            v += *(const f4*)(p + (long)z * MN);
        }
        if (a.bias) v += *(const f4*)&a.bias[n];
        if (a.rowvec) v += *(const f4*)&a.rowvec[(m / V) * a.rowvec_ld + n];
        if (a.res) v += *(const f4*)&a.res[m * a.out_ld + n];
        if (a.out_f32) *(f4*)&a.out_f32[m * a.out_ld + n] = v;
        if (a.out_f16) {
            h4 hv = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            *(h4*)((_Float16*)a.out_f16 + m * a.out_ld + n) = hv;
        }
    }
}

// The split-K reduction above fused with the statistics pass of the GroupNorm that reads the tensor next (es_conv_args.gn_part_out):
// the workgroup / thread layout and the summation order of k_gn_partial (one workgroup per (object, voxel tile); lane = 4 channels,
// 4 voxel rows in flight; per-lane sums over its rows top to bottom, the 4 row lanes combined pairwise, channels -> group left to
// right), with the element formed here instead of loaded -- slab sum in slab order + bias + rowvec + res, as k_conv_splitk_reduce --
// and stored on the way.  One launch instead of two, and the same bits as the two.
__global__ __launch_bounds__(256) void k_conv_splitk_reduce_gn(const es_conv_args a, long M, int V, int S, int vt) {
    __shared__ float ssum4[4][2048], ssq4[4][2048];
    float* ssum = ssum4[0];
    float* ssq = ssq4[0];
    const int o = blockIdx.y, tile = blockIdx.x, C = a.N;
    const int v0 = tile * vt;
    const int nv = min(vt, V - v0);
    const int c4n = C >> 2;
    const int cx = threadIdx.x & 63, vy = threadIdx.x >> 6;
    const long MN = M * a.N;
    auto elem = [&](int c, int v) -> f4 {
        const long m = (long)o * V + v0 + v;
        const float* p = (const float*)a.workspace + m * a.N + c;
        f4 x = *(const f4*)p;
        int z = 1;
        for (; z + 3 < S; z += 4) {
            const f4 t0 = *(const f4*)(p + (long)z * MN), t1 = *(const f4*)(p + (long)(z + 1) * MN),
                     t2 = *(const f4*)(p + (long)(z + 2) * MN), t3 = *(const f4*)(p + (long)(z + 3) * MN);
            x += t0; x += t1; x += t2; x += t3;
        }
        if (z + 2 < S) {
            const f4 t0 = *(const f4*)(p + (long)z * MN), t1 = *(const f4*)(p + (long)(z + 1) * MN), t2 = *(const f4*)(p + (long)(z + 2) * MN);
            x += t0; x += t1; x += t2;
        } else if (z + 1 < S) {
            const f4 t0 = *(const f4*)(p + (long)z * MN), t1 = *(const f4*)(p + (long)(z + 1) * MN);
            x += t0; x += t1;
        } else if (z < S) {
            x += *(const f4*)(p + (long)z * MN);
        }
        if (a.bias) x += *(const f4*)&a.bias[c];
        if (a.rowvec) x += *(const f4*)&a.rowvec[(long)o * a.rowvec_ld + c];
        if (a.res) x += *(const f4*)&a.res[m * a.out_ld + c];
        *(f4*)&a.out_f32[m * a.out_ld + c] = x;
        if (a.out_f16) {
            h4 hv = {(_Float16)x[0], (_Float16)x[1], (_Float16)x[2], (_Float16)x[3]};
            *(h4*)((_Float16*)a.out_f16 + m * a.out_ld + c) = hv;
        }
        return x;
    };
    for (int c4 = cx; c4 < c4n; c4 += 64) {
        const int c = c4 * 4;
        f4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
        if (nv == vt && (vt & 15) == 0) {
            for (int v = vy; v < vt; v += 16) {
                const f4 x0 = elem(c, v), x1 = elem(c, v + 4), x2 = elem(c, v + 8), x3 = elem(c, v + 12);
                s += x0; q += x0 * x0; s += x1; q += x1 * x1; s += x2; q += x2 * x2; s += x3; q += x3 * x3;
            }
        } else {
            for (int v = vy; v < nv; v += 4) { const f4 x = elem(c, v); s += x; q += x * x; }
        }
        *(f4*)&ssum4[vy][c] = s;
        *(f4*)&ssq4[vy][c] = q;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        ssum[c] = (ssum4[0][c] + ssum4[1][c]) + (ssum4[2][c] + ssum4[3][c]);
        ssq[c] = (ssq4[0][c] + ssq4[1][c]) + (ssq4[2][c] + ssq4[3][c]);
    }
    __syncthreads();
    const int gs = C / a.gn_part_groups;
    if (threadIdx.x < a.gn_part_groups) {
        float s = 0.f, q = 0.f;
        for (int k = 0; k < gs; ++k) { s += ssum[threadIdx.x * gs + k]; q += ssq[threadIdx.x * gs + k]; }
        float* dst = a.gn_part_out + (((long)o * gridDim.x + tile) * a.gn_part_groups + threadIdx.x) * 2;
        dst[0] = s; dst[1] = q;
    }
}

// voxel-tile size of the GroupNorm statistics pass by workgroup count: small problems (few objects per GPU when sharded) get smaller
// tiles; Oh = the object count of the WHOLE problem (O_hint) so that a shard tiles like the unsharded run
static inline int gn_voxel_tile(long Oh, int V) {
    int vt = GN_VT;
    while (vt > 8 && Oh * ((V + vt - 1) / vt) < 512) vt >>= 1;
    return vt;
}

// Row-group sums of es_conv_args.gn_stats_out for the routes whose epilogue does not form them (64- / 128-row tiles, k_linear_ws,
// split K): one thread per (64-row group, column quad), the summation order of conv_epilogue<STATS_> -- the even rows of the group
// top to bottom, the odd rows top to bottom, then the two halves -- so that every route leaves the same bits.
__global__ __launch_bounds__(256) void k_rowgroup_stats(const float* x, long M, int N, int ld, float* st) {
    const int N4 = N >> 2;
    const long nrg = (M + 63) >> 6;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nrg * N4; i += (long)gridDim.x * 256) {
        const long rg = i / N4;
        const int n = (int)(i - rg * N4) * 4;
        const float* p = x + rg * 64 * ld + n;
        const int nv = (int)((M - rg * 64) < 64 ? (M - rg * 64) : 64);
        f4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, q[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        for (int r0 = 0; r0 < 64; r0 += 16) {
            f4 v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = r0 + r < nv ? *(const f4*)(p + (long)(r0 + r) * ld) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (r0 + r < nv) {
                    s[r & 1] += v[r];
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[r & 1][e] = fmaf(v[r][e], v[r][e], q[r & 1][e]);
                }
        }
        *(f4*)&st[rg * N + n] = s[0] + s[1];
        *(f4*)&st[(nrg + rg) * N + n] = q[0] + q[1];
    }
}

// (object, group) statistics from the producers' row-group sums (es_gn_args.stats1 / stats2): V/64 row groups x channels-per-group
// values per block, double accumulation, fixed-order tree -- independent of how many objects the launch holds.
__global__ __launch_bounds__(256) void k_gn_finalize_rg(const es_gn_args a, float* fin) {
    __shared__ double ds[256], dq[256];
    const int o = blockIdx.y, g = blockIdx.x;
    const int C = a.C1 + a.C2, gs = C / a.groups, nrgo = a.V >> 6;
    const long nrg = (long)a.O * nrgo;
    double s = 0.0, q = 0.0;
    for (int e = threadIdx.x; e < nrgo * gs; e += 256) {
        const int rgi = e / gs, c = g * gs + (e - rgi * gs);
        const float* st; int Cs, cc;
        if (c < a.C1) { st = a.stats1; Cs = a.C1; cc = c; } else { st = a.stats2; Cs = a.C2; cc = c - a.C1; }
        const long idx = ((long)o * nrgo + rgi) * Cs + cc;
        s += st[idx];
        q += st[nrg * Cs + idx];
    }
    ds[threadIdx.x] = s; dq[threadIdx.x] = q;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) { ds[threadIdx.x] += ds[threadIdx.x + w]; dq[threadIdx.x] += dq[threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)gs * a.V;
        const double mean = ds[0] / n;
        double var = dq[0] / n - mean * mean;
        if (var < 0.0) var = 0.0;
        fin[((long)o * a.groups + g) * 2] = (float)mean;
        fin[((long)o * a.groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)a.eps));
    }
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace

#ifdef ES_STAMP
extern "C" int es_debug_set_stamp(void* p) {
    unsigned long long* q = (unsigned long long*)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &q, sizeof(q)) == hipSuccess ? 0 : 1;
}
#endif

// library-wide one-off initialisation hook (nothing to allocate since the zero-page gather kernel was retired)
int es_vol_init(void) { return 0; }

// Tiled weight image consumed by the conv kernels: [n-tile of 224][K step][1024 slots of 16 B] where slot p holds
// row = p >> 2 (output channel within the tile, rows 224..255 are zero padding), physical 16-B chunk p & 3 =
// logical chunk ^ swizzle(row); K step index = (channel chunk of 32) * taps + tap  (tap inner).
static inline int h_swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

extern "C" size_t es_pack_conv_f16_size(int N, int Cin, int taps) {
    return (size_t)((N + BN - 1) / BN) * (size_t)(Cin / 32) * taps * (BNP * BK);
}

static inline uint16_t f32_to_f16_bits(float f) {
    _Float16 h = (_Float16)f;     // round-to-nearest-even, as torch .half()
    uint16_t u;
    memcpy(&u, &h, 2);
    return u;
}

// h_w: [N][CinW][taps] (PyTorch conv weight flattened over kd,kh,kw); Cin = CinW rounded up to 32
extern "C" int es_pack_conv_f16(const float* h_w, int N, int CinW, int taps, uint16_t* h_out) {
    const int Cin = (CinW + 31) / 32 * 32;
    const int nt = (N + BN - 1) / BN, kch = Cin / 32;
    // one 16 KiB block per (n-tile, channel chunk, tap); the blocks are independent: spread over the host's threads
    es_parallel_for((long)nt * kch * taps, [=](long bi) {
        const int tap = (int)(bi % taps), kc = (int)((bi / taps) % kch), t = (int)(bi / ((long)taps * kch));
        uint16_t* blk = h_out + (size_t)bi * (BNP * BK);
        for (int p = 0; p < BNP * 4; ++p) {
            const int row = p >> 2, lc = (p & 3) ^ h_swz(row);
            const int n = t * BN + row;
            for (int e = 0; e < 8; ++e) {
                const int c = kc * 32 + lc * 8 + e;
                blk[p * 8 + e] = (row < BN && n < N && c < CinW)
                                     ? f32_to_f16_bits(h_w[((size_t)n * CinW + c) * taps + tap]) : 0;
            }
        }
    });
    return 0;
}

// The same image formed on the device from the fp32 weight already in HBM (one 16-byte slot per thread): a process's first scene
// call spent 6 of its 9 s in the host loop above (a strided gather with a software fp32 -> fp16 conversion, ~70 M elements/s on 8
// threads against 410 M weights); reading the uploaded tensor with the stride of the taps costs the GPU milliseconds.
__global__ __launch_bounds__(256) void k_pack_conv_f16(const float* __restrict__ w, int N, int CinW, int taps, int kch, h8* __restrict__ out,
                                                        long nslots) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= nslots) return;
    const int p = (int)(s & (BNP * 4 - 1));
    const long bi = s / (BNP * 4);
    const int tap = (int)(bi % taps), kc = (int)((bi / taps) % kch), t = (int)(bi / ((long)taps * kch));
    const int row = p >> 2, lc = (p & 3) ^ ((0x1320 >> (((row >> 2) & 3) * 4)) & 3);
    const int n = t * BN + row;
    h8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = kc * 32 + lc * 8 + e;
        v[e] = (row < BN && n < N && c < CinW) ? (_Float16)w[((size_t)n * CinW + c) * taps + tap] : (_Float16)0.f;   // round to nearest even
    }
    out[s] = v;
}

extern "C" int es_pack_conv_f16_dev(const float* d_w, int N, int CinW, int taps, uint16_t* d_out, es_stream stream) {
    ES_REQUIRE(d_w && d_out && N > 0 && CinW > 0 && (taps == 1 || taps == 27), "es_pack_conv_f16_dev: N=%d Cin=%d taps=%d", N, CinW, taps);
    const int Cin = (CinW + 31) / 32 * 32;
    const long nslots = (long)es_pack_conv_f16_size(N, Cin, taps) / 8;
    hipLaunchKernelGGL(k_pack_conv_f16, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_w, N, CinW, taps, Cin / 32,
                       (h8*)d_out, nslots);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

// Row-major image [N][taps][Cin] for k_conv_small_n (N <= 4)
extern "C" int es_pack_conv_rows_f16(const float* h_w, int N, int CinW, int taps, uint16_t* h_out) {
    const int Cin = (CinW + 31) / 32 * 32;
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < Cin; ++c)
                h_out[((size_t)n * taps + t) * Cin + c] = c < CinW ? f32_to_f16_bits(h_w[((size_t)n * CinW + c) * taps + t]) : 0;
    return 0;
}

// ---- Route options of the volume path (round 5: no environment access) -----------------------------------------------------------
// Everything below decides WHERE an fp32 sum is cut (split-K factors, which kernel forms a GroupNorm's sums) or which kernel family
// multiplies: it changes the bits of the results.  Until round 4 these were process-environment switches read inside the library: two
// ranks -- or a process that saves a model file and one that replays it -- with different environments silently disagreed (ADVICE r4,
// VERDICT r4 #6).  Now they are process-wide options with constant defaults that ONLY an explicit es_vol_set_option() call changes
// (tests and A/B tools); es_model_save records them in the file and es_model_load refuses a file written under other values.
#ifndef ES_CONV_A3_DEFAULT
#define ES_CONV_A3_DEFAULT true
#endif
struct VolOpt { const char* name; int value; };
static VolOpt g_vo[] = {
    {"conv_tile", 0},          // 128: force 128-row tiles
    {"conv_force256", 0},      // 1: 256-row producer/consumer tiles for any problem size (tests)
    {"conv_ws", 1},            // 0: k_conv_lean instead of the warp-specialised k_conv_ws
    {"conv_wssplit", 1},       // 0: small problems on 128- / 64-row tiles instead of 256-row tiles with split K
    {"conv_wss_target", 256},  // workgroup target of that split
    {"conv_deep", 1},          // 0: no k_linear_deep for small K-short linear launches
    {"conv_tinysplit", 1},     // 0: tiny K-short launches without split K
    {"gn_rg", 1},              // 0: GroupNorm statistics always from a pass over the tensor
    {"conv_few", 1},           // 0: the few-objects routes of round 6 off (round-5 routing for small problems)
    {"conv_st_bm", 0},         // tools only: 64 / 128 = every producer/consumer-eligible launch on k_conv_ws tiles of that many rows
    {"conv_st_np", 4},         // tools only: producer waves of that tile (4; 8 with 128-row tiles)
    {"conv_st_ns", 3},         // tools only: ring depth of that tile (3; 6 with 64-row tiles, 5 with 128-row tiles and 8 producers)
    {"conv_kw_ks", 0},         // tools only: 4 / 2 = every eligible launch on k_conv_kw with that many K streams per workgroup
};
static int vo(const char* name) {
    for (const VolOpt& o : g_vo) if (!strcmp(o.name, name)) return o.value;
    return 0;
}
extern "C" int es_vol_set_option(const char* name, int value) {
    ES_REQUIRE(name != nullptr, "es_vol_set_option: null name");
    for (VolOpt& o : g_vo) if (!strcmp(o.name, name)) { o.value = value; return 0; }
    ES_REQUIRE(false, "es_vol_set_option: unknown option '%s'", name);
}
// "name=value;name=value;..." of every route option, in a fixed order (what es_model_save records); returns the length needed
extern "C" int es_vol_options(char* out, int cap) {
    std::string s;
    for (const VolOpt& o : g_vo) { s += o.name; s += '='; s += std::to_string(o.value); s += ';'; }
    if (out && cap > 0) { strncpy(out, s.c_str(), (size_t)cap - 1); out[cap - 1] = 0; }
    return (int)s.size() + 1;
}

// k_conv_ws launcher for one (tile rows, consumer waves, producer waves, ring depth): the 256-row tile of rounds 1-5 (8 consumers as
// 4 x 2, 4 producers) and, round 6, the few-objects tiles of 64 / 128 rows (4 consumers as 2 x 2 of BM_/2 x 112, 4 or 8 producers,
// deeper rings).  The instantiation's dynamic-LDS limit is set once per process.
template <int BM_, int NC_, int NP_, int NS_>
static int launch_ws(const es_conv_args* a, const ConvGeom& g, int ncdhw, dim3 grid, hipStream_t st, bool upm, bool geglu, bool stats) {
    constexpr int LDS = NS_ * (BM_ * BK * 2 + BNP * BK * 2);
    constexpr bool CAN_STATS = BM_ / (NC_ / 2) == 64;             // row-group sums are per 64-row wave
    static_assert(LDS <= 160 * 1024, "ring exceeds the CU's LDS");
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        auto set = [](const void* f) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess && attr_err == hipSuccess) attr_err = e;
        };
        set((const void*)k_conv_ws<BM_, NC_, NP_, false, ES_EPI_NONE, false, NS_>);
        set((const void*)k_conv_ws<BM_, NC_, NP_, true, ES_EPI_NONE, false, NS_>);
        set((const void*)k_conv_ws<BM_, NC_, NP_, false, ES_EPI_GEGLU, false, NS_>);
        if constexpr (CAN_STATS) {
            set((const void*)k_conv_ws<BM_, NC_, NP_, false, ES_EPI_NONE, true, NS_>);
            set((const void*)k_conv_ws<BM_, NC_, NP_, true, ES_EPI_NONE, true, NS_>);
        }
    });
    ES_REQUIRE(attr_err == hipSuccess, "es_conv_mfma_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    const dim3 blk(64 * (NC_ + NP_));
    if (geglu) hipLaunchKernelGGL((k_conv_ws<BM_, NC_, NP_, false, ES_EPI_GEGLU, false, NS_>), grid, blk, LDS, st, *a, g, ncdhw);
    else if (stats) {
        if constexpr (CAN_STATS) {
            if (upm) hipLaunchKernelGGL((k_conv_ws<BM_, NC_, NP_, true, ES_EPI_NONE, true, NS_>), grid, blk, LDS, st, *a, g, ncdhw);
            else hipLaunchKernelGGL((k_conv_ws<BM_, NC_, NP_, false, ES_EPI_NONE, true, NS_>), grid, blk, LDS, st, *a, g, ncdhw);
        } else ES_REQUIRE(false, "launch_ws: row-group sums need 64-row waves");
    }
    else if (upm) hipLaunchKernelGGL((k_conv_ws<BM_, NC_, NP_, true, ES_EPI_NONE, false, NS_>), grid, blk, LDS, st, *a, g, ncdhw);
    else hipLaunchKernelGGL((k_conv_ws<BM_, NC_, NP_, false, ES_EPI_NONE, false, NS_>), grid, blk, LDS, st, *a, g, ncdhw);
    return 0;
}

// k_conv_kw: K split inside the workgroup (KS_ streams x NCH_ column halves of 112)
template <int KS_, int NCH_>
static int launch_kw(const es_conv_args* a, const ConvGeom& g, int ncdhw, long M, int ntn, int S, hipStream_t st, bool upm, bool geglu) {
    constexpr int RING = 3 * KS_ * (64 * BK * 2 + NCH_ * 112 * BK * 2), PART = 4 * 112 * 64 * 4;
    constexpr int LDS = RING > PART ? RING : PART;
    static std::once_flag once;
    static hipError_t attr_err = hipSuccess;
    std::call_once(once, [] {
        auto set = [](const void* f) {
            const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (e != hipSuccess && attr_err == hipSuccess) attr_err = e;
        };
        set((const void*)k_conv_kw<KS_, NCH_, false, ES_EPI_NONE>);
        set((const void*)k_conv_kw<KS_, NCH_, true, ES_EPI_NONE>);
        set((const void*)k_conv_kw<KS_, NCH_, false, ES_EPI_GEGLU>);
    });
    ES_REQUIRE(attr_err == hipSuccess, "es_conv_mfma_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    const dim3 grid((unsigned)((M + 63) / 64), (unsigned)(ntn * (2 / NCH_)), (unsigned)S), blk(768);
    if (geglu) hipLaunchKernelGGL((k_conv_kw<KS_, NCH_, false, ES_EPI_GEGLU>), grid, blk, LDS, st, *a, g, ncdhw);
    else if (upm) hipLaunchKernelGGL((k_conv_kw<KS_, NCH_, true, ES_EPI_NONE>), grid, blk, LDS, st, *a, g, ncdhw);
    else hipLaunchKernelGGL((k_conv_kw<KS_, NCH_, false, ES_EPI_NONE>), grid, blk, LDS, st, *a, g, ncdhw);
    return 0;
}

// emits != nullptr: dry run -- report whether this launch would form gn_stats_out in its epilogue, launch nothing
static int conv_dispatch(const es_conv_args* a, es_stream stream, int* emits) {
    ES_REQUIRE(a->Cin % 32 == 0 && a->Cin > 0, "es_conv_mfma_f16: Cin=%d must be a positive multiple of 32", a->Cin);
    ES_REQUIRE(a->taps == 27 || a->taps == 1, "es_conv_mfma_f16: taps=%d", a->taps);
    ES_REQUIRE(!a->a2 || (a->Cin2 % 32 == 0 && a->Cin2 > 0), "es_conv_mfma_f16: Cin2=%d", a->Cin2);
    ES_REQUIRE(a->out_f32 || a->out_f16, "es_conv_mfma_f16: no output");
    ES_REQUIRE(a->epilogue == ES_EPI_NONE || (a->epilogue == ES_EPI_GEGLU && a->N % 224 == 0 && a->out_f16 && !a->out_f32 && !a->res &&
                                              !a->rowvec && !a->a2 && a->bias && a->out_ld >= a->N / 2 && a->out_ld % 8 == 0 && a->splitk <= 1),
               "es_conv_mfma_f16: GEGLU epilogue needs N %% 224 == 0 (N=%d), bias, f16 output only, no split-K", a->N);
    ConvGeom g;
    g.O = a->O; g.D = a->D; g.H = a->H; g.W = a->W;
    g.Hi = a->H; g.Wi = a->W;
    g.Di = a->D;
    if (a->mode == ES_CONV_DOWN_HW) { g.Hi = 2 * a->H; g.Wi = 2 * a->W; }
    if (a->mode == ES_CONV_DOWN_DHW) { g.Hi = 2 * a->H; g.Wi = 2 * a->W; g.Di = 2 * a->D; }
    if (a->mode == ES_CONV_UP_DHW) g.Di = a->D / 2;
    if (a->mode == ES_CONV_UP_HW || a->mode == ES_CONV_UP_DHW) { g.Hi = a->H / 2; g.Wi = a->W / 2; }
    g.lw = ilog2_exact(a->W); g.lh = ilog2_exact(a->H); g.ld = ilog2_exact(a->D);
    ES_REQUIRE(g.lw >= 0 && g.lh >= 0 && g.ld >= 0, "es_conv_mfma_f16: D,H,W must be powers of two (%d,%d,%d)", a->D, a->H, a->W);
    int ncdhw = a->out_ld < 0 ? 1 : 0;            // out_ld < 0 selects NCDHW fp32 output [O,N,V]
    ES_REQUIRE(!ncdhw || (a->out_f32 && !a->res && !a->out_f16), "es_conv_mfma_f16: NCDHW output is fp32-only, no residual");
    const long M = (long)a->O * a->D * a->H * a->W;
    // split-K factors are derived from Mh: the row count of the WHOLE problem when this launch is one shard of it
    // O_hint < 0 (round 6): the CANONICAL arithmetic of a reference shard of -O_hint objects, whatever the launch's own object count --
    // every rank of every world size (1 included) cuts K where a shard of that size is cut best, so all of them leave the same bits
    // and the small shards are the fast ones (until round 5 the reference was the unsharded run and its shards paid 2x)
    const long Oh = a->O_hint < 0 ? -(long)a->O_hint : (long)(a->O_hint > a->O ? a->O_hint : a->O);
    const long Mh = Oh * a->D * a->H * a->W;
    // The kernels address their operands with 31-bit byte offsets from a buffer descriptor.  Larger tensors are processed in
    // object chunks (every object is independent; O_hint keeps the split-K choice of the whole problem).
    {
        const long per_obj_in = (long)g.Di * g.Hi * g.Wi * a->Cin * 2, per_obj_in2 = a->a2 ? (long)a->D * a->H * a->W * a->Cin2 * 2 : 0;
        const long halo = 4L * ((g.Hi + 1) * g.Wi + 1) * a->Cin;
        const long lim = (1L << 31) - halo - 1;
        long omax = a->O;
        if (per_obj_in > 0) omax = std::min(omax, lim / per_obj_in);
        if (per_obj_in2 > 0) omax = std::min(omax, lim / per_obj_in2);
        ES_REQUIRE(omax >= 1, "es_conv_mfma_f16: one object exceeds 2 GiB of input (%ld bytes)", per_obj_in);
        if (omax < a->O) {
            if (emits) {
                // a chunked launch forms no GroupNorm sums in its epilogues; its chunks split K as launches of omax objects do (with
                // O_hint: as the whole / the reference problem does) and write their slabs into the SAME workspace: report the slab
                // count in units of the whole launch's M x N (es_conv_split_of: the planner sizes the workspace with it)
                es_conv_args c = *a;
                c.O = (int32_t)omax;
                c.O_hint = a->O_hint < 0 ? a->O_hint : (a->O_hint > a->O ? a->O_hint : a->O);
                c.gn_stats_out = nullptr; c.gn_part_out = nullptr;
                int e2 = 0;
                if (int rc = conv_dispatch(&c, nullptr, &e2)) return rc;
                const long sc = e2 >> 8;
                *emits = (int)((sc * omax + a->O - 1) / a->O) << 8;
                return 0;
            }
            const long V = (long)a->D * a->H * a->W;
            for (long o0 = 0; o0 < a->O; o0 += omax) {
                es_conv_args c = *a;
                c.O = (int32_t)std::min(omax, (long)a->O - o0);
                c.O_hint = a->O_hint < 0 ? a->O_hint : (a->O_hint > a->O ? a->O_hint : a->O);
                c.a = (const char*)a->a + o0 * per_obj_in;
                if (a->a2) c.a2 = (const char*)a->a2 + o0 * per_obj_in2;
                if (a->rowvec) c.rowvec = a->rowvec + o0 * a->rowvec_ld;
                if (ncdhw) {
                    c.out_f32 = a->out_f32 + o0 * a->N * V;
                } else {
                    if (a->res) c.res = a->res + o0 * V * a->out_ld;
                    if (a->out_f32) c.out_f32 = a->out_f32 + o0 * V * a->out_ld;
                    if (a->out_f16) c.out_f16 = (char*)a->out_f16 + o0 * V * a->out_ld * 2;
                }
                c.gn_stats_out = nullptr;                  // (the planes are laid out for the whole tensor: one pass below)
                c.gn_part_out = nullptr;                   // (never requested: es_conv_emits_gn_part() == 0 for a chunked launch)
                if (int rc = conv_dispatch(&c, stream, nullptr)) return rc;
            }
            if (a->gn_stats_out) {
                ES_REQUIRE(a->out_f32 && !ncdhw && a->N % 4 == 0 && a->out_ld % 4 == 0 && V % 64 == 0, "es_conv_mfma_f16: gn_stats_out needs a channels-last fp32 output");
                const long Mt = (long)a->O * V, n4 = ((Mt + 63) / 64) * (a->N / 4);
                const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
                hipLaunchKernelGGL(k_rowgroup_stats, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)a->out_f32, Mt, a->N, a->out_ld, a->gn_stats_out);
                ES_CHECK_HIP(hipGetLastError());
            }
            return 0;
        }
    }
    // N <= 4 with a narrow input (VQ-VAE conv_out 64 -> 1 at 64^3): direct kernel, weights in the rows layout.  Wider
    // inputs (UNet eps conv 224 -> 3) go through the MFMA tile: 98 % column padding, but the direct kernel's 756
    // dependent 16-B loads per voxel cost 2.4x (O=32) to 10x (O=4) more than the padded tile.
    if (a->N <= 4 && a->taps == 27 && a->Cin <= 64) {
        if (emits) { *emits = 0; return 0; }
        ES_REQUIRE(a->mode == ES_CONV_SAME && !a->a2 && !a->res && !a->rowvec && !a->out_f16 && !a->gn_stats_out,
                   "es_conv_mfma_f16: the N<=4, Cin<=64 direct kernel takes SAME mode, fp32 output, no fusions");
        const size_t lds_t = (size_t)1000 * (a->Cin * 2 + 16) + (size_t)a->N * 27 * a->Cin * 2;
        if (a->D % 8 == 0 && a->H % 8 == 0 && a->W % 8 == 0 && a->Cin % 8 == 0 && lds_t <= 160 * 1024) {
            static std::once_flag once_t;
            static hipError_t attr_t = hipSuccess;
            std::call_once(once_t, [] { attr_t = hipFuncSetAttribute((const void*)k_conv_small_n_tiled, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
            ES_REQUIRE(attr_t == hipSuccess, "es_conv_mfma_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_t));
            const unsigned nblk = (unsigned)((long)a->O * (a->D / 8) * (a->H / 8) * (a->W / 8));
            hipLaunchKernelGGL(k_conv_small_n_tiled, dim3(nblk), dim3(256), lds_t, (hipStream_t)stream, *a, g, ncdhw);
        } else {
            hipLaunchKernelGGL(k_conv_small_n, dim3((unsigned)((M + 255) / 256)), dim3(256), (size_t)a->N * 27 * a->Cin * 2,
                               (hipStream_t)stream, *a, g, ncdhw);
        }
        ES_CHECK_HIP(hipGetLastError());
        return 0;
    }
    // N <= 16 from a wide input, NCDHW fp32 output, no fusions (the UNet's output conv): halo'd LDS image + one MFMA column
    static const char* n16_env = getenv("ES_CONV_N16");           // A/B switch (timing only: same K order, same bits): 0 = off
    if (a->N <= 16 && a->taps == 27 && a->Cin > 64 && ncdhw && a->mode == ES_CONV_SAME && !a->a2 && !a->res && !a->rowvec &&
        !a->out_f16 && a->D % 4 == 0 && a->H % 4 == 0 && a->W % 16 == 0 && !(n16_env && atoi(n16_env) == 0) &&
        (long)a->O * a->D * a->H * a->W * a->Cin * 2 < (1L << 31)) {
        if (emits) { *emits = 0; return 0; }
        static std::once_flag once_n;
        static hipError_t attr_n = hipSuccess;
        std::call_once(once_n, [] { attr_n = hipFuncSetAttribute((const void*)k_conv_n16, hipFuncAttributeMaxDynamicSharedMemorySize, N16_A_BYTES + N16_B_BYTES); });
        ES_REQUIRE(attr_n == hipSuccess, "es_conv_mfma_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_n));
        const unsigned nblk = (unsigned)((long)a->O * (a->D / 4) * (a->H / 4) * (a->W / 16));
        hipLaunchKernelGGL(k_conv_n16, dim3(nblk), dim3(256), N16_A_BYTES + N16_B_BYTES, (hipStream_t)stream, *a, g);
        ES_CHECK_HIP(hipGetLastError());
        return 0;
    }
    const int ntn = (a->N + BN - 1) / BN;
    // small-M layers (16x4x4 level): 64-row tiles double the number of workgroups (>= 1 per CU)
    // Tile / split choice by workgroup count (256 CUs x 2 resident workgroups):
    //   >= 256 workgroups of 256 rows -> 8-wave 256x224 tiles;
    //   otherwise 128x224 tiles, and when even those give < 512 workgroups (16x4x4 level: M = 8192) K is split
    //   over S workgroups per tile (fixed-order reduction kernel, deterministic).
    const long wg256 = ((M + 255) / 256) * ntn, wg128 = ((M + 127) / 128) * ntn;
    const long hg256 = ((Mh + 255) / 256) * ntn, hg128 = ((Mh + 127) / 128) * ntn;      // the same counts for the whole problem
    const int nks = a->taps * (a->Cin / 32) + (a->a2 ? a->Cin2 / 32 : 0);
    constexpr int LDSLIN_GEGLU = 4 * (256 * BK * 2 + BNP * BK * 2) + 8 * 16 * 72 * 2;      // k_linear_ws<GEGLU>: 4-slot ring + fp16 slabs
    constexpr int LDS256 = 3 * (256 * BK * 2 + BNP * BK * 2), LDS128 = 3 * (128 * BK * 2 + BNP * BK * 2),
                  LDS64 = 3 * (64 * BK * 2 + BNP * BK * 2);
    hipStream_t st = (hipStream_t)stream;
    const bool want_stats = a->gn_stats_out != nullptr;
    bool stats_done = false;
    ES_REQUIRE(!want_stats || ((a->out_f32 || a->out_f16) && !ncdhw && a->N % 4 == 0 && a->out_ld % 4 == 0 && (a->D * a->H * a->W) % 64 == 0 &&
                               a->epilogue == ES_EPI_NONE),
               "es_conv_mfma_f16: gn_stats_out needs a channels-last output, N %% 4 == 0 and voxels per object %% 64 == 0");
    int S = a->splitk;
    const bool can_split = a->workspace && !ncdhw && a->N % 4 == 0 && a->out_ld % 4 == 0 && (!a->rowvec || a->rowvec_ld % 4 == 0);
    const bool no256 = vo("conv_tile") == 128;               // (route options: es_vol_set_option, no environment access)
    const bool force256 = vo("conv_force256") == 1;
    const bool geglu = a->epilogue == ES_EPI_GEGLU;
    const bool upm = a->mode == ES_CONV_UP_HW || a->mode == ES_CONV_UP_DHW;
    // k_conv_ws's epilogue is compiled for vector-aligned channels-last outputs addressed with 32-bit element offsets
    const bool ws_epilogue_ok = !ncdhw && a->N % 4 == 0 && a->out_ld % 4 == 0 && (!a->rowvec || a->rowvec_ld % 4 == 0) &&
                                M * (long)a->out_ld < (1L << 30) && M * (long)a->N < (1L << 30) &&
                                (!a->rowvec || (a->D * a->H * a->W) % 64 == 0);      // a wave's 64 rows in one object: rowvec per wave
    const bool ws = vo("conv_ws") != 0 && ws_epilogue_ok;
    // ---- round 6: the few-objects routes (fewer than 256 tiles of 256 rows in the reference problem) --------------------------------
    // few_ref: what the REFERENCE problem (Mh rows) decides -- 1: a plain split of S over workgroups, 2: S = 4 K streams inside the
    // workgroup (k_conv_kw, no slabs; the same bits as a plain split of 4).  The launch's own row count only picks the tile that
    // realises that split: 64- / 128-row producer/consumer tiles, k_conv_kw, or the 256-row tiles when it has >= 256 of them.
    // Measured at 4 objects per GPU (profiles/r06_tiles_microbench.txt): 3x3x3 launches -5 ... -15 % on 128-row tiles with half the
    // slabs of the 256-row split; 1x1 launches with 14-105 K units -20 ... -25 % on k_conv_kw; wide 1x1 launches -15 ... -25 % on
    // 64-row producer/consumer tiles (no split).
    int few_ref = 0, few_bm = 0;
    bool few_kw = false;
    if (vo("conv_few") != 0 && ws && geglu && !upm && a->taps == 1 && a->splitk <= 0 && hg256 < 256 && !no256 && !force256 &&
        !vo("conv_st_bm") && !vo("conv_kw_ks") && ((M + 127) / 128) * ntn < 320) {
        // the GEGLU projection of a small problem (never split: its epilogue needs the whole sum): 64-row producer/consumer tiles
        // instead of the non-specialised 64-row kernel (672 -> 5376 at 16x4x4 with 4 objects)
        few_ref = 1; S = 1; few_bm = 64;
    }
    if (vo("conv_few") != 0 && ws && !geglu && a->splitk < 0 && can_split && hg256 < 256 && !no256 && !force256 &&
        !vo("conv_st_bm") && !vo("conv_kw_ks")) {
        const long h128 = ((Mh + 127) / 128) * ntn, h64h = ((Mh + 63) / 64) * ((a->N + 111) / 112);
        if (a->taps == 27 || nks >= 112) {
            int s2 = (int)((256 + h128 / 2) / h128);
            s2 = s2 < 1 ? 1 : (s2 > 8 ? 8 : s2);
            // (the split that fills the chip best must fit ONE round of workgroups: 96 tiles of 128 rows -- the 16x4x4 level at 16
            //  objects -- want S = 3 = 288 workgroups; with S = 2 the 256-row tiles' S = 5 is the better use of the chip, 91 against 96 us)
            const bool one_round = h128 * s2 <= 272;
            while (s2 > 1 && nks / s2 < 24) --s2;
            // (very long K on a handful of tiles -- 1344 -> 672 at 16x4x4 with 4 objects -- stays on the 256-row tiles with S = 16;
            //  an UNSPLIT launch that would leave a quarter of the CUs idle -- 192 tiles of 128 rows: the 16x4x4 level at 32 objects --
            //  stays on the 256-row tiles with S = 2: the same workgroup count on the tile that moves fewer bytes, 169 against 179 us)
            if (!(nks >= 800 && h128 * 8 < 256) && one_round && (s2 >= 2 || h128 >= 224)) { few_ref = 1; S = s2; }
        } else if (a->taps == 1) {
            if (h64h <= 272 && nks >= 12) { few_ref = 2; S = 4; }          // (one round of 64 x 112 tiles: with 384 of them the 64-row tiles win)
            else { few_ref = 1; S = 1; }
        }
        if (few_ref) {
            const long l128 = ((M + 127) / 128) * ntn, l64h = ((M + 63) / 64) * ((a->N + 111) / 112);
            if (wg256 >= 256) few_bm = 0;                                          // enough 256-row tiles: they take the split as it is
            else if (few_ref == 2 && l64h <= 768) few_kw = true;
            else if (a->taps == 1 && few_ref == 1 && S == 1) few_bm = l128 >= 320 ? 128 : 64;
            else few_bm = 128;
        }
    }
    if (S < 0 && !few_ref) {                               // auto
        S = 1;
        if (can_split && hg256 < 256 && hg128 < 512) {
            // ~3 workgroups per CU: enough for the dynamic scheduler to balance the tail, few enough that the epilogue
            // and the reduction (S slabs of M x N floats) stay small (measured, tools/microbench_small.py: M = 8192
            // rows best at S = 4, 16384 rows at S = 2-3, 1024-4096 rows at S = 8)
            S = (int)((768 + hg128 / 2) / hg128);
            if (S < 1) S = 1;
            if (S > 8) S = 8;
            while (S > 1 && nks / S < 24) --S;
        }
    }
    bool split256 = false;
    if (a->splitk < 0 && !few_ref && S == 1 && can_split && hg256 >= 256) {
        // tile quantisation: e.g. 384 workgroups of 256 rows on 256 CUs = 2 rounds, the second half empty.  Split K by
        // the smallest factor that fills the last round (>= 95 %) if the plain launch wastes more than 20 %.
        const long r1 = (hg256 + 255) / 256;
        if ((double)hg256 / (double)(r1 * 256) < 0.8)
            for (int s2 = 2; s2 <= 4; ++s2) {
                const long w = hg256 * s2;
                if ((double)w / (double)(((w + 255) / 256) * 256) >= 0.95 && nks / s2 >= 48) { S = s2; split256 = true; break; }
            }
    }
    if (S <= 1 || !can_split) S = 1;
    ES_REQUIRE(!upm || (!a->a2 && a->taps == 27), "es_conv_mfma_f16: nearest-up modes take 3x3x3 convs without a fused skip");
    if (force256 && a->splitk < 0 && !split256) S = 1;
    // Small problems (few objects per GPU: the strong-scaling regime, or the 16x4x4 level): fewer than 256 tiles of 256 rows.  The
    // 128- / 64-row kernels below stream the weight tile twice / four times per 256 rows and cost 0.9 us per K unit and workgroup
    // against 0.64 us for a 256-row producer/consumer tile, so keep the 256-row tiles and split K until about one workgroup per
    // CU runs (A/B ES_CONV_WSSPLIT: shape step 21.44 -> 20.76 ms at 32 objects, 13.69 -> 12.61 / 9.22 -> 8.32 / 6.68 -> 6.27 ms at 16 / 8 / 4).
    bool ws_split = false;
    if (ws && !geglu && a->splitk < 0 && !few_ref && can_split && hg256 < 256 && !no256 && !force256 && vo("conv_wssplit") != 0) {
        // (hg256: the tile count of the WHOLE problem -- a shard with O_hint makes the choice the unsharded run makes, so its
        //  partial sums are cut in the same places and the results stay bit-identical; it then simply runs fewer workgroups)
        const long wst = vo("conv_wss_target");              // workgroup target of the split (256 = one round)
        int s2 = (int)(wst / hg256);
        const int s2max = Mh * (long)a->N <= (1L << 22) ? 16 : 8;      // workspace contract (echoscene_hip.h): 16 slabs for small outputs
        if (s2 > s2max) s2 = s2max;
        while (s2 > 1 && nks / s2 < 24) --s2;
        if (s2 >= 2 && hg256 * s2 >= 160) { S = s2; ws_split = true; }
    }
    // Tiny K-short problems (the transformer linears at <= 8 objects per GPU: e.g. 1024 rows x 672 columns, 21 K units): even the
    // 64-row tiles give only a few dozen workgroups, each a lone, latency-bound chain of K units (22-30 us for < 1 GFLOP).
    // Split K over 64-row tiles until about one workgroup per CU runs (>= 5 units per split).
    // ... round 4: such launches go to k_linear_deep (7-slot ring issued at entry, no split, no reduction kernel) when they are plain
    // linears of at most ONE round of 64-row tiles (140 KB of LDS = one workgroup per CU: with more tiles than CUs the 3-slot kernel's two
    // workgroups per CU win; stand-alone at 4 objects: 448 -> 448 19.7 -> 12.5 us, 672 -> 2016 21.4 -> 13.8, but 448 -> 1344 with 384
    // tiles 16.7 -> 21.3); the split below remains for the shapes it does not take (fused skip phase, 27 taps, more tiles)
    bool deep = false;
    {
        const long hg64 = ((Mh + 63) / 64) * ntn;
        deep = !ws_split && !few_ref && a->splitk <= 0 && !force256 && !no256 && hg256 < 256 && hg64 <= 256 && a->taps == 1 && !a->a2 && a->mode == ES_CONV_SAME &&
               nks >= 4 && nks <= 28 && !ncdhw && M * (long)a->Cin * 2 < (1L << 31) && vo("conv_deep") != 0;
    }
    bool tiny_split = false;
    {
        const long hg64 = ((Mh + 63) / 64) * ntn;
        if (!deep && !ws_split && !few_ref && a->splitk < 0 && can_split && !force256 && hg256 < 256 && hg128 < 128 && hg64 < 256 && nks >= 10 &&
            vo("conv_tinysplit") != 0) {
            int s3 = (int)((256 + hg64 - 1) / hg64);
            const int s3max = Mh * (long)a->N <= (1L << 22) ? 16 : 8;
            if (s3 > s3max) s3 = s3max;
            if (s3 > nks / 5) s3 = nks / 5;
            if (s3 >= 2) { S = s3; tiny_split = true; }
        }
    }
    // tools: every eligible launch on the small producer/consumer tiles (split K as the caller says, 1 when it says nothing)
    const int kw_ks = ws && (!geglu || !upm) ? vo("conv_kw_ks") : 0;
    const int st_bm = kw_ks ? 64 : (ws && (!geglu || !upm) ? vo("conv_st_bm") : 0);
    if (st_bm) {
        deep = tiny_split = ws_split = split256 = false;
        S = a->splitk > 1 && can_split && !geglu ? a->splitk : 1;
    }
    const bool few_small = few_ref && (few_kw || few_bm);      // a few-objects launch on its own kernels (not the 256-row tiles)
    const bool route256 = !st_bm && !few_small && (wg256 >= 256 || force256 || ws_split) && (S == 1 || split256 || ws_split || wg256 >= 256) && !no256 && !tiny_split && !deep;
    // 1x1 / linear launches with several column tiles: one workgroup walks NCB column tiles of its row tile (k_linear_ws)
    static const char* lin_env = getenv("ES_CONV_LINWS");        // A/B switch: 0 = off
    int ncb = 1;
    if (route256 && ws && a->taps == 1 && !a->a2 && a->mode == ES_CONV_SAME && S == 1 && ntn >= 2 && !(lin_env && atoi(lin_env) == 0)) {
        static const int cand[5] = {8, 6, 4, 3, 2};
        static const char* ncb_env = getenv("ES_LIN_NCB_MAX");    // A/B switch (timing only: same K order, same bits): cap of the walk
        const int ncb_max = ncb_env ? atoi(ncb_env) : 8;
        for (int k = 0; k < 5; ++k)
            if (cand[k] <= ncb_max && ntn % cand[k] == 0 && ((M + 255) / 256) * (ntn / cand[k]) >= 256) { ncb = cand[k]; break; }
    }
    // the producer/consumer 256-row kernel forms the row-group sums of gn_stats_out in its epilogue; every other route runs
    // k_rowgroup_stats over the finished output
    const bool epi_stats = (route256 || st_bm == 128 || (few_small && few_bm == 128)) && ncb == 1 && ws && !geglu && S == 1 && (a->out_f32 || a->out_f16) && a->N % 4 == 0 && a->out_ld % 4 == 0 &&
                           (a->D * a->H * a->W) % 64 == 0;
    // (bit 1) a split launch can form the next GroupNorm's per-tile partial sums in its reduction kernel (gn_part_out)
    // (the few-objects routes: from the REFERENCE problem's decision -- K streams inside the workgroup have no reduction launch,
    //  whatever kernel realises the split in this launch)
    const bool part_ok = S > 1 && !deep && few_ref != 2 && a->out_f32 && !ncdhw && a->N % 4 == 0 && a->N <= 2048 && a->out_ld == a->N && a->gn_part_groups > 0 &&
                         a->gn_part_groups <= 64 && a->N % a->gn_part_groups == 0;
    if (emits) { *emits = (epi_stats ? 1 : 0) | (part_ok ? 2 : 0) | ((few_kw ? 1 : S) << 8); return 0; }
    {   // one-off per process, thread-safe: dynamic LDS limits of the conv kernels
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            auto set = [](const void* f, int bytes) {
                const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess && attr_err == hipSuccess) attr_err = e;
            };
            set((const void*)k_conv_lean<256, 8>, LDS256);
            set((const void*)k_conv_lean<128, 4>, LDS128);
            set((const void*)k_conv_lean<64, 4>, LDS64);
            set((const void*)k_conv_lean<256, 8, true>, LDS256);
            set((const void*)k_conv_lean<128, 4, true>, LDS128);
            set((const void*)k_conv_lean<64, 4, true>, LDS64);
            set((const void*)k_linear_deep, 7 * (64 * BK * 2 + BNP * BK * 2));
            set((const void*)k_linear_ws<ES_EPI_NONE>, LDS256 + 8 * 16 * 116 * 4);
            set((const void*)k_linear_ws<ES_EPI_GEGLU>, LDS256 + 8 * 16 * 116 * 4);
            set((const void*)k_linear_ws<ES_EPI_GEGLU, 4>, LDSLIN_GEGLU);
        });
        ES_REQUIRE(attr_err == hipSuccess, "es_conv_mfma_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    }
    if (few_kw) {
        if (int rc = launch_kw<4, 1>(a, g, ncdhw, M, ntn, 1, st, upm, geglu)) return rc;
        S = 1;                                                 // (the four K ranges met in LDS: no slabs, no reduction launch)
    } else if (few_small) {
        const dim3 grid((unsigned)((M + few_bm - 1) / few_bm), ntn, S);
        const bool stt = want_stats && epi_stats;
        int rc;
        if (few_bm == 64) rc = launch_ws<64, 4, 4, 3>(a, g, ncdhw, grid, st, upm, geglu, stt);
        else rc = launch_ws<128, 4, 8, 5>(a, g, ncdhw, grid, st, upm, geglu, stt);
        if (rc) return rc;
        stats_done = stt;
    } else if (kw_ks) {
        int rc = 1;
        if (kw_ks == 4) rc = launch_kw<4, 1>(a, g, ncdhw, M, ntn, S, st, upm, geglu);
        else if (kw_ks == 2) rc = launch_kw<2, 2>(a, g, ncdhw, M, ntn, S, st, upm, geglu);
        else ES_REQUIRE(false, "es_conv_mfma_f16: conv_kw_ks=%d (2 or 4)", kw_ks);
        if (rc) return rc;
    } else if (st_bm) {
        const int np = vo("conv_st_np");
        const dim3 grid((unsigned)((M + st_bm - 1) / st_bm), ntn, S);
        const bool stt = want_stats && epi_stats;
        int rc = 1;
        const int ns = vo("conv_st_ns");
        if (st_bm == 64 && np == 4 && ns == 3) rc = launch_ws<64, 4, 4, 3>(a, g, ncdhw, grid, st, upm, geglu, stt);
        else if (st_bm == 64 && np == 4 && ns == 6) rc = launch_ws<64, 4, 4, 6>(a, g, ncdhw, grid, st, upm, geglu, stt);
        else if (st_bm == 128 && np == 4 && ns == 3) rc = launch_ws<128, 4, 4, 3>(a, g, ncdhw, grid, st, upm, geglu, stt);
        else if (st_bm == 128 && np == 8 && ns == 3) rc = launch_ws<128, 4, 8, 3>(a, g, ncdhw, grid, st, upm, geglu, stt);
        else if (st_bm == 128 && np == 8 && ns == 5) rc = launch_ws<128, 4, 8, 5>(a, g, ncdhw, grid, st, upm, geglu, stt);
        else ES_REQUIRE(false, "es_conv_mfma_f16: conv_st_bm=%d conv_st_np=%d conv_st_ns=%d is not a built tile", st_bm, np, ns);
        if (rc) return rc;
        stats_done = stt;
    } else if (deep) {
        S = 1;
        hipLaunchKernelGGL(k_linear_deep, dim3((unsigned)((M + 63) / 64), ntn, 1), dim3(256), 7 * (64 * BK * 2 + BNP * BK * 2), st, *a, g);
    } else if (route256) {
        dim3 grid((unsigned)((M + 255) / 256), ntn, S);
        if (ncb > 1) {
            const dim3 lgrid(grid.x, (unsigned)(ntn / ncb), 1);
            constexpr int LDSLIN = LDS256 + 8 * 16 * 116 * 4;
            // (a FOURTH ring slot fits behind the GEGLU variant's small fp16 slabs -- the step DESIGN.md had listed as cheap since round 2;
            //  measured in round 4 on one box: shape step 18.41 ms with three slots, 18.44-18.47 with four: no gain, kept as a switch)
            static const char* ring_env = getenv("ES_LIN_RING");      // A/B switch (timing only): 4 = four-slot ring for the GEGLU variant
            if (geglu && ring_env && atoi(ring_env) == 4) hipLaunchKernelGGL((k_linear_ws<ES_EPI_GEGLU, 4>), lgrid, dim3(768), LDSLIN_GEGLU, st, *a, g, ncb);
            else if (geglu) hipLaunchKernelGGL((k_linear_ws<ES_EPI_GEGLU>), lgrid, dim3(768), LDSLIN, st, *a, g, ncb);
            else hipLaunchKernelGGL((k_linear_ws<ES_EPI_NONE>), lgrid, dim3(768), LDSLIN, st, *a, g, ncb);
        } else if (ws && (!geglu || !upm)) {
            // ring depth of the 256-row tile: 3 (rounds 1-5); ES_CONV_NS = 4 / 5 is a timing-only A/B switch (same K order, same bits)
            static const char* ns_env = getenv("ES_CONV_NS");
            const int ns = ns_env ? atoi(ns_env) : 3;
            const bool stt = want_stats && epi_stats;
            // 3x3x3 SAME convs on volumes with W <= 16: the A tile of a (chunk, kd, kh) group staged once, the kw = -1 / +1 operands
            // shifted in registers (k_conv_ws3; bit-identical to k_conv_ws).  ES_CONV_A3 = 0 / 1: timing-only A/B switch
            static const char* a3_env = getenv("ES_CONV_A3");
            const bool a3 = (a3_env ? atoi(a3_env) != 0 : ES_CONV_A3_DEFAULT) && a->taps == 27 && a->mode == ES_CONV_SAME && !a->a2 && a->W <= 16 && a->W >= 4;
            if (a3) {
                static std::once_flag once3;
                static hipError_t err3 = hipSuccess;
                std::call_once(once3, [] {
                    err3 = hipFuncSetAttribute((const void*)k_conv_ws3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 16384);
                    for (auto fl : {std::pair<const void*, int>{(const void*)k_conv_ws3<true>, 6 * 16384}, {(const void*)k_conv_ws3<false, true>, 9 * 16384},
                                    {(const void*)k_conv_ws3<true, true>, 9 * 16384}}) {
                        const hipError_t e2 = hipFuncSetAttribute(fl.first, hipFuncAttributeMaxDynamicSharedMemorySize, fl.second);
                        if (err3 == hipSuccess) err3 = e2;
                    }
                });
                static const char* gb_env = getenv("ES_CONV_GB");          // timing-only A/B switch: 0 = one barrier per K unit instead of one per (chunk, kd, kh) group
                ES_REQUIRE(err3 == hipSuccess, "es_conv_mfma_f16: hipFuncSetAttribute failed: %s", hipGetErrorString(err3));
                if (!gb_env || atoi(gb_env) != 0) {
                    if (stt) hipLaunchKernelGGL((k_conv_ws3<true, true>), grid, dim3(768), 9 * 16384, st, *a, g);
                    else hipLaunchKernelGGL((k_conv_ws3<false, true>), grid, dim3(768), 9 * 16384, st, *a, g);
                }
                else if (stt) hipLaunchKernelGGL((k_conv_ws3<true>), grid, dim3(768), 6 * 16384, st, *a, g);
                else hipLaunchKernelGGL((k_conv_ws3<false>), grid, dim3(768), 6 * 16384, st, *a, g);
                if (stt) stats_done = true;
            } else {
            int rc;
            if (ns == 4) rc = launch_ws<256, 8, 4, 4>(a, g, ncdhw, grid, st, upm, geglu, stt);
            else if (ns == 5) rc = launch_ws<256, 8, 4, 5>(a, g, ncdhw, grid, st, upm, geglu, stt);
            else rc = launch_ws<256, 8, 4, 3>(a, g, ncdhw, grid, st, upm, geglu, stt);
            if (rc) return rc;
            if (stt) stats_done = true;
            }
        }
        else if (upm) hipLaunchKernelGGL((k_conv_lean<256, 8, true>), grid, dim3(512), LDS256, st, *a, g, ncdhw);
        else hipLaunchKernelGGL((k_conv_lean<256, 8>), grid, dim3(512), LDS256, st, *a, g, ncdhw);
    } else if ((wg128 >= 512 || S > 1) && !tiny_split) {
        dim3 grid((unsigned)((M + 127) / 128), ntn, S);
        if (upm) hipLaunchKernelGGL((k_conv_lean<128, 4, true>), grid, dim3(256), LDS128, st, *a, g, ncdhw);
        else hipLaunchKernelGGL((k_conv_lean<128, 4>), grid, dim3(256), LDS128, st, *a, g, ncdhw);
    } else {
        dim3 grid((unsigned)((M + 63) / 64), ntn, tiny_split ? S : 1);
        if (upm) hipLaunchKernelGGL((k_conv_lean<64, 4, true>), grid, dim3(256), LDS64, st, *a, g, ncdhw);
        else hipLaunchKernelGGL((k_conv_lean<64, 4>), grid, dim3(256), LDS64, st, *a, g, ncdhw);
    }
    if (S > 1) {
        const long n4 = M * (a->N / 4);
        const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
        if (a->gn_part_out && part_ok) {
            const int V = a->D * a->H * a->W;
            const int vt = gn_voxel_tile(Oh, V);
            hipLaunchKernelGGL(k_conv_splitk_reduce_gn, dim3((V + vt - 1) / vt, a->O), dim3(256), 0, st, *a, M, V, S, vt);
        } else
            hipLaunchKernelGGL(k_conv_splitk_reduce, dim3(blocks), dim3(256), 0, st, *a, M, a->D * a->H * a->W, S);
    }
    if (want_stats && !stats_done) {
        ES_REQUIRE(a->out_f32, "es_conv_mfma_f16: this launch's route forms gn_stats_out by a pass over the fp32 output: out_f32 must be given "
                               "(es_conv_emits_gn_stats() == 0)");
        const long n4 = ((M + 63) / 64) * (a->N / 4);
        const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
        hipLaunchKernelGGL(k_rowgroup_stats, dim3(blocks), dim3(256), 0, st, (const float*)a->out_f32, M, a->N, a->out_ld, a->gn_stats_out);
    }
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_conv_mfma_f16(const es_conv_args* a, es_stream stream) { return conv_dispatch(a, stream, nullptr); }

// 1 when es_conv_mfma_f16(args) would form gn_stats_out inside its own epilogue (the planner only asks a conv for the sums then:
// behind any other route they would cost a pass over the output, i.e. what the GroupNorm's own statistics pass costs), 0 when
// not, -1 on invalid arguments.  Host-only: launches nothing.
extern "C" int es_conv_emits_gn_stats(const es_conv_args* a) {
    int e = 0;
    return conv_dispatch(a, nullptr, &e) == 0 ? (e & 1) : -1;
}

extern "C" int es_conv_emits_gn_part(const es_conv_args* a) {
    int e = 0;
    return conv_dispatch(a, nullptr, &e) == 0 ? ((e >> 1) & 1) : -1;
}

// The number of fp32 slabs [S][M][N] es_conv_mfma_f16(args) writes into args->workspace (1: none -- no split, or K split inside the
// workgroup), -1 on invalid arguments.  Host-only: the planner sizes the workspace with it.
extern "C" int es_conv_split_of(const es_conv_args* a) {
    int e = 0;
    return conv_dispatch(a, nullptr, &e) == 0 ? (e >> 8) : -1;
}

extern "C" int es_groupnorm_vol(const es_gn_args* a, es_stream stream) {
    const int C = a->C1 + a->C2;
    ES_REQUIRE(C % a->groups == 0 && C <= 2048 && a->groups <= 64, "es_groupnorm_vol: C=%d groups=%d", C, a->groups);
    ES_REQUIRE(a->C1 % 8 == 0 && a->C2 % 8 == 0, "es_groupnorm_vol: channel counts must be multiples of 8 (%d,%d)", a->C1, a->C2);
    ES_REQUIRE(a->stats != nullptr, "es_groupnorm_vol: stats scratch missing");
    // voxel-tile sizes by workgroup count: small problems (few objects per GPU when sharded) get smaller tiles
    // tile sizes from the whole problem when this launch is a shard (O_hint > O), from the reference shard when O_hint < 0
    const long Oh = a->O_hint < 0 ? -(long)a->O_hint : (long)(a->O_hint > a->O ? a->O_hint : a->O);
    const int vt = gn_voxel_tile(Oh, a->V);
    const int ntiles = (a->V + vt - 1) / vt;
    float* part = a->stats;      // caller-provided scratch of O*ceil(V/8)*groups*2 floats
    const bool from_rg = a->stats1 && (!a->x2 || a->stats2) && a->V % 64 == 0 && (a->x1_is_f16 || vo("gn_rg") != 0);
    ES_REQUIRE(!a->x1_is_f16 || (from_rg && !a->x2 && !a->raw_f16),
               "es_groupnorm_vol: an f16 source needs the producer's row-group sums (stats1), one source, no raw copy");
    ES_REQUIRE(!a->part_in || (!from_rg && !a->x2), "es_groupnorm_vol: part_in needs one source and no row-group sums");
    if (a->part_in) part = (float*)a->part_in;
    else if (!from_rg) hipLaunchKernelGGL(k_gn_partial, dim3(ntiles, a->O), dim3(256), 0, (hipStream_t)stream, *a, part, vt);
    int vpb = 32;
    while (vpb > 8 && (long)a->O * ((a->V + vpb - 1) / vpb) < 512) vpb >>= 1;
    while (vpb < 1024 && (long)a->O * ((a->V + vpb - 1) / vpb) > 16384) vpb <<= 1;     // (64^3 volumes: fewer, larger blocks)
    // long tile lists (VQ-VAE decoder at 32^3 / 64^3): the statistics are reduced ONCE per object; the final [O][groups][2] floats
    // sit behind the partials in the caller's scratch (es_gn_args.stats)
    const float* fin = nullptr;
    if (from_rg) {                 // statistics from the producers' row-group sums: no pass over x1 / x2
        hipLaunchKernelGGL(k_gn_finalize_rg, dim3(a->groups, a->O), dim3(256), 0, (hipStream_t)stream, *a, part);
        fin = part;
    } else if (ntiles > 128) {
        // (with part_in the partials live in the PRODUCER's buffer, which has no room behind them: the final statistics go to the
        //  front of the caller's own scratch then.  Until round 6 they were written behind part_in -- 256 bytes past the end of a
        //  buffer that ends on a page boundary was a GPU memory fault at one object per GPU, 16^3 level)
        float* f = a->part_in ? a->stats : part + (size_t)a->O * ntiles * a->groups * 2;
        hipLaunchKernelGGL(k_gn_finalize, dim3(a->groups, a->O), dim3(256), 0, (hipStream_t)stream, *a, part, ntiles, f);
        fin = f;
    }
    hipLaunchKernelGGL(k_gn_apply, dim3((a->V + vpb - 1) / vpb, a->O), dim3(256), 0, (hipStream_t)stream, *a, part, ntiles, vpb, fin);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_layernorm_tokens(const es_ln_args* a, es_stream stream) {
    ES_REQUIRE(a->C <= 1024 && a->C > 0 && a->C % 4 == 0, "es_layernorm_tokens: C=%d (a multiple of 4, max 1024)", a->C);
    // resident waves loop over the rows: 8 workgroups of 4 waves per CU at most
    long wgs = ((long)a->M + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    hipLaunchKernelGGL(k_layernorm, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, *a);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_attention_f16(const es_attn_args* a, es_stream stream) {
    ES_REQUIRE(a->dhead % 4 == 0 && a->dhead <= 256 && a->dhead > 0, "es_attention_f16: dhead=%d (multiple of 4, <= 256)", a->dhead);
    ES_REQUIRE(a->scale > 0.f, "es_attention_f16: scale=%g must be positive (the row maximum is taken over the raw scores)", (double)a->scale);
    const bool big = a->Ntok >= 512 && a->dhead <= 96;       // 128 query rows per workgroup (4 waves x 2 row tiles), else 64 (4 x 1)
    const int rows = big ? 128 : 64;
    dim3 grid((a->Ntok + rows - 1) / rows, a->B * a->heads);
    hipStream_t st = (hipStream_t)stream;
    if (a->dhead <= 32) { if (big) hipLaunchKernelGGL((k_attention<32, 4, 2>), grid, dim3(256), 0, st, *a); else hipLaunchKernelGGL((k_attention<32, 4, 1>), grid, dim3(256), 0, st, *a); }
    else if (a->dhead <= 64) { if (big) hipLaunchKernelGGL((k_attention<64, 4, 2>), grid, dim3(256), 0, st, *a); else hipLaunchKernelGGL((k_attention<64, 4, 1>), grid, dim3(256), 0, st, *a); }
    else if (a->dhead <= 96) { if (big) hipLaunchKernelGGL((k_attention<96, 4, 2>), grid, dim3(256), 0, st, *a); else hipLaunchKernelGGL((k_attention<96, 4, 1>), grid, dim3(256), 0, st, *a); }
    else hipLaunchKernelGGL((k_attention<256, 4, 1>), grid, dim3(256), 0, st, *a);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_geglu_f16(const es_geglu_args* a, es_stream stream) {
    ES_REQUIRE(a->C4 % 4 == 0, "es_geglu_f16: C4=%d", a->C4);
    const long n4 = (long)a->M * (a->C4 / 4);
    const int blocks = (int)((n4 + 255) / 256 < 8192 ? (n4 + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_geglu, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *a);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

static int latent_to_cl(const float* x, int O, int C, int V, int Cpad, void* out, int is_f32, es_stream stream) {
    const long n = (long)O * V * Cpad;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_to_cl, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, O, C, V, Cpad, (_Float16*)out, is_f32);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}
extern "C" int es_latent_to_cl_f16(const float* x, int O, int C, int V, int Cpad, void* out, es_stream stream) { return latent_to_cl(x, O, C, V, Cpad, out, 0, stream); }
extern "C" int es_latent_to_cl_f32(const float* x, int O, int C, int V, int Cpad, void* out, es_stream stream) { return latent_to_cl(x, O, C, V, Cpad, out, 1, stream); }

extern "C" int es_split_f16x3(const float* x, long M, int C, void* out, es_stream stream) {
    ES_REQUIRE(x && out && M > 0 && C > 0 && C % 4 == 0, "es_split_f16x3: M=%ld C=%d (C a multiple of 4)", M, C);
    const long n = M * (C / 4);
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_split_f16x3, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, M, C, (_Float16*)out);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_shape_stem(const es_stem_args* a, es_stream stream) {
    const long n1 = (long)a->O * 32 * 512, n2 = (long)a->O * 512;
    const int Cx = a->Cin ? a->Cin : 3;
    ES_REQUIRE(Cx >= 1 && Cx <= 4, "es_shape_stem: Cin=%d (3 or 4)", Cx);
    (void)n1;
    hipLaunchKernelGGL(k_stem1, dim3(64, a->O), dim3(256), (size_t)(Cx * 288 + 32 * Cx * 27) * 4, (hipStream_t)stream, *a);
    {
        static std::once_flag once_s;
        static hipError_t attr_s = hipSuccess;
        std::call_once(once_s, [] { attr_s = hipFuncSetAttribute((const void*)k_stem2, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 513 * 4); });
        ES_REQUIRE(attr_s == hipSuccess, "es_shape_stem: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_s));
    }
    hipLaunchKernelGGL(k_stem2, dim3((unsigned)((n2 * 8 + 255) / 256)), dim3(256), 32 * 513 * 4, (hipStream_t)stream, *a);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_vq_lookup(const es_vq_args* a, es_stream stream) {
    ES_REQUIRE(a->n_embed > 0 && a->n_embed * 16 <= 160 * 1024 - 1024, "es_vq_lookup: n_embed=%d too large for LDS", a->n_embed);
    ES_REQUIRE(a->Cpad >= 3, "es_vq_lookup: Cpad=%d", a->Cpad);
    {
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute((const void*)k_vq_lookup, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        });
        ES_REQUIRE(attr_err == hipSuccess, "es_vq_lookup: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    }
    const long M = (long)a->O * a->V;
    hipLaunchKernelGGL(k_vq_lookup, dim3((unsigned)((M + 255) / 256)), dim3(256), (size_t)a->n_embed * 16, (hipStream_t)stream, *a);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}
