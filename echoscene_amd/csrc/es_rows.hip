// "rows" path: fp32 fused linear over small-M node/triple matrices + the diffusion updates.
//
// Roofline note: one layout denoising step streams ~524 MB of fp32 weights
// for ~8.9 GFLOP (17 flop/B) -- HBM-bound, and in practice launch/latency-bound because it is a
// chain of ~140 dependent [32 x K] @ [K x N] products.  Design consequences:
//   * weights are pre-packed in MFMA-fragment order so that every wave-level load is one
//     contiguous 1 KiB global_load_dwordx4 that lands directly in B-operand registers
//     (no LDS round trip for the streamed operand -- it is used once per workgroup);
//   * the small activation tile (<= 32 rows x 1024 cols) is staged ONCE per workgroup in LDS,
//     where the norm/activation prologue (GroupNorm+SiLU / LayerNorm / GEGLU / gather /
//     CSR mean pooling) is applied, so no separate elementwise kernels (= no extra launches);
//   * exact fp32 on the matrix pipe: v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain);
//   * K is split over the 8 waves of a workgroup and reduced through LDS in a fixed order
//     (deterministic, no float atomics).
#include "es_common.h"
#include <cstdlib>

namespace {

constexpr int MT = 32;          // rows per workgroup (two 16-row MFMA tiles)
constexpr int KC = 1024;        // K chunk staged in LDS
constexpr int LDX = KC + 8;     // +8 floats: conflict-free ds_read_b128 of A fragments
constexpr int NWAVE = 8;
constexpr int NTHREAD = NWAVE * 64;
constexpr int MAXJ = KC / 16 / NWAVE;   // 16-wide k-blocks per wave per chunk

struct Smem {
    float x[MT][LDX];
    float gb[2][KC];          // gamma / beta of the norm prologue (staged once, read by every row)
};

__device__ __forceinline__ const float* seg_base(const es_seg& s) {
    const float* p = s.ptr;   // (batched launches add blockIdx.z * a_bstride to segment 0 at the call site)
    if (s.step) p += (long)(*s.step) * s.step_stride;
    return p;
}

// Staging of one K chunk of the (virtually concatenated) A operand into LDS.
// Thread mapping: row r = tid >> 4 (32 rows), column lane cl = tid & 15; a thread walks its row in
// steps of 16 float4 -> 16 consecutive lanes read 256 contiguous bytes, no integer division, and the
// segment loop / mode switch are wave-uniform (scalar branches only).  Loads are unconditional
// (row index clamped, result masked) so that 8 of them are in flight per thread before first use.
template <int PRO>
__device__ __forceinline__ f4 post_a4(f4 v, f4 g) {
    if (PRO == ES_PRO_GEGLU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * es_gelu(g[e]);
    } else if (PRO == ES_PRO_SILU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = es_silu(v[e]);
    }
    return v;
}

template <int PRO>
__device__ __forceinline__ void stage_chunk(const es_linear_args& a, Smem& sm, int m0, int kc0, int kc, int tid) {
    const int r = tid >> 4, cl = tid & 15;
    const int m = m0 + r;
    const bool row_ok = m < a.M;
    const int mc = row_ok ? m : a.M - 1;
    int koff = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const es_seg& sg = a.seg[s];
        const int lo = max(kc0, koff), hi = min(min(kc0 + kc, koff + sg.width), a.K);
        if (lo < hi) {
            const int w4 = (hi - lo) >> 2;
            const float* base = seg_base(sg) + (lo - koff) + (s == 0 ? (long)blockIdx.z * a.a_bstride : 0);
            float* dst = &sm.x[r][lo - kc0];
            if (sg.mode == ES_SEG_CSRMEAN) {
                const int e0 = sg.idx[mc], e1 = sg.idx[mc + 1];
                const float inv = 1.0f / (float)max(e1 - e0, 1);
                for (int c4 = cl; c4 < w4; c4 += 16) {
                    f4 v = {0.f, 0.f, 0.f, 0.f};
                    // stored order == scatter_add order of the reference.  Entries are fetched 8 at a time (index pairs,
                    // then rows) and added in order: the scene node has ~2*O incident edges, and one dependent
                    // index -> row load pair per entry made this op 49 us (12 % of a layout step).
                    for (int e = e0; e < e1; e += 8) {
                        long off[8];
                        f4 x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int ee = min(e + u, e1 - 1);
                            off[u] = (long)sg.ent_row[ee] * sg.ld + sg.ent_off[ee];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = *(const f4*)(base + off[u] + 4 * c4);
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (e + u < e1) v += x[u];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = row_ok ? v[e] / (float)max(e1 - e0, 1) : 0.f;
                    (void)inv;
                    *(f4*)(dst + 4 * c4) = v;
                }
            } else {
                const long rowoff = (long)(sg.mode == ES_SEG_GATHER ? sg.idx[mc] : mc) * sg.ld;
                const float* src = base + rowoff;
                for (int u0 = 0; u0 * 16 < w4; u0 += 8) {
                    f4 v[8], gt[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c4 = min(cl + 16 * (u0 + u), w4 - 1);
                        v[u] = *(const f4*)(src + 4 * c4);
                        if (PRO == ES_PRO_GEGLU) gt[u] = *(const f4*)(src + a.K + 4 * c4);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c4 = cl + 16 * (u0 + u);
                        if (c4 < w4) {
                            f4 y = post_a4<PRO>(v[u], gt[u]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) y[e] = row_ok ? y[e] : 0.f;
                            *(f4*)(dst + 4 * c4) = y;
                        }
                    }
                }
            }
        }
        koff += sg.width;
    }
    // zero the K padding (K not a multiple of 16)
    const int kend = min(kc0 + kc, a.K) - kc0;
    for (int c = kend + cl; c < kc; c += 16) sm.x[r][c] = 0.f;
}

// Norm prologues (GroupNorm32 [+SiLU] / LayerNorm) done while staging: the thread that stages columns
// 4*(cl + 16u) of row r keeps them in registers (u < K/64 <= 16), statistics are reduced with lane shuffles inside the
// 16-lane row group (GroupNorm group = gs/4 adjacent lanes; LayerNorm = all 16), the normalised tile is written to LDS
// once.  No dependent LDS walks, no per-element global loads of the affine (gamma/beta sit in LDS).
template <int PRO>
__device__ __forceinline__ void stage_norm_load(const es_linear_args& a, int m0, int tid, f4 (&v)[16]) {
    const int r = tid >> 4, cl = tid & 15;
    const int m = m0 + r;
    const int mc = m < a.M ? m : a.M - 1;
    const int nu = a.K >> 6;                             // K % 64 == 0 (host-checked)
    const int w0 = a.seg[0].width, w1 = a.nseg > 1 ? a.seg[1].width : 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        if (u < nu) {
            const int c0 = u << 6;                       // first column of this 64-wide slice: uniform segment choice
            const es_seg& sg = c0 < w0 ? a.seg[0] : (c0 < w0 + w1 ? a.seg[1] : a.seg[2]);
            const int cb = c0 < w0 ? 0 : (c0 < w0 + w1 ? w0 : w0 + w1);
            v[u] = *(const f4*)(seg_base(sg) + (long)mc * sg.ld + (c0 - cb) + 4 * cl);
        }
    }
}

template <int PRO>
__device__ __forceinline__ void stage_norm(const es_linear_args& a, Smem& sm, int m0, int tid, f4 (&v)[16]) {
    const int r = tid >> 4, cl = tid & 15;
    const int m = m0 + r;
    const bool row_ok = m < a.M;
    const int K = a.K, nu = K >> 6;
    if (PRO == ES_PRO_LN) {
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < nu) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
        const float mean = s / (float)K;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < nu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; q += d * d; }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o, 16);
        const float rstd = 1.0f / sqrtf(q / (float)K + a.eps);
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < nu) {
            const int c = (u << 6) + 4 * cl;
            const f4 ga = *(const f4*)&sm.gb[0][c], be = *(const f4*)&sm.gb[1][c];
            f4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = row_ok ? (v[u][e] - mean) * rstd * ga[e] + be[e] : 0.f;
            *(f4*)&sm.x[r][c] = y;
        }
    } else {
        const int gs = K >> 5;                           // channels per group: 4, 8, 16 or 32  (K = 128 .. 1024)
        const int lpg = gs >> 2;                         // lanes per group: 1, 2, 4, 8
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < nu) {
            float s = (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
            for (int o = 1; o < lpg; o <<= 1) s += __shfl_xor(s, o, 16);
            const float mean = s / (float)gs;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; q += d * d; }
            for (int o = 1; o < lpg; o <<= 1) q += __shfl_xor(q, o, 16);
            const float rstd = 1.0f / sqrtf(q / (float)gs + a.eps);
            const int c = (u << 6) + 4 * cl;
            const f4 ga = *(const f4*)&sm.gb[0][c], be = *(const f4*)&sm.gb[1][c];
            f4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = (v[u][e] - mean) * rstd * ga[e] + be[e];
                if (PRO == ES_PRO_GN_SILU) t = es_silu(t);
                y[e] = row_ok ? t : 0.f;
            }
            *(f4*)&sm.x[r][c] = y;
        }
    }
}

template <int PRO>
__global__ __launch_bounds__(NTHREAD) void k_linear_rows(const es_linear_args a) {
    __shared__ Smem sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt = blockIdx.x;
    const int m0 = blockIdx.y * MT;
    const int Kp = (a.K + 15) & ~15;
    const int nkb_total = Kp >> 4;
    const int bz = blockIdx.z;                           // batched launch: z-th problem of identical shape
    const f4* wp = (const f4*)a.wpack + ((size_t)bz * gridDim.x + nt) * nkb_total * 64;
    f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, q = lane >> 4;

    // (0) Epilogue operands (bias, residuals, the affine of the second GroupNorm output) depend only on the output
    // coordinates: issued FIRST so that their latency overlaps the weight stream, the staging and the product instead
    // of adding one more dependent L2/HBM round trip after the reduction (launches here are ~5 us of pure latency).
    const int ml = tid >> 4, nl = tid & 15;              // 32 x 16 outputs, one per thread
    const int m_e = m0 + ml, n_e = nt * 16 + nl;
    const bool geglu = a.act == ES_ACT_GEGLU;
    const float* bias = a.bias ? a.bias + (long)bz * a.N : nullptr;
    const bool ok_e = m_e < a.M && n_e < a.N;
    const int nres = geglu ? nt * 8 + nl : n_e;          // column of the residual / output
    const bool ok_res = geglu ? (ok_e && nl < 8) : ok_e;
    float e_bias = 0.f, e_res = 0.f, e_res2 = 0.f, e_g2 = 0.f, e_b2 = 0.f;
    if (bias && n_e < a.N) e_bias = bias[n_e];
    if (a.res && ok_res) e_res = a.res[(long)m_e * a.res_ld + nres];
    if (a.res2 && ok_res) e_res2 = a.res2[(long)m_e * a.res2_ld + nres];
    if (a.out2 && ok_e) { e_g2 = a.gn2_gamma[n_e]; e_b2 = a.gn2_beta[n_e]; }

    for (int kc0 = 0; kc0 < Kp; kc0 += KC) {
        const int kc = min(KC, Kp - kc0);
        const int nkb = kc >> 4;
        // (1) issue this wave's weight-fragment loads first: HBM latency overlaps the staging below
        f4 bf[MAXJ];
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int kb = min(wave + j * NWAVE, nkb - 1);
            bf[j] = __builtin_nontemporal_load(&wp[(size_t)((kc0 >> 4) + kb) * 64 + lane]);
        }
        // (2) stage the activation chunk with its prologue applied
        if (kc0 > 0) __syncthreads();
        if (PRO == ES_PRO_GN || PRO == ES_PRO_GN_SILU || PRO == ES_PRO_LN) {
            // affine (one float4 per thread: K <= KC, single chunk) and the activation rows are fetched in ONE round
            // trip; the affine goes through LDS because every row needs all of it
            const int c = tid * 4;
            f4 gv = {0.f, 0.f, 0.f, 0.f};
            const bool isg = c < a.K, isb = !isg && c - NTHREAD * 2 >= 0 && c - NTHREAD * 2 < a.K;
            if (isg) gv = *(const f4*)&a.gamma[c];
            else if (isb) gv = *(const f4*)&a.beta[c - NTHREAD * 2];
            f4 v[16];
            stage_norm_load<PRO>(a, m0, tid, v);
            if (isg) *(f4*)&sm.gb[0][c] = gv;
            else if (isb) *(f4*)&sm.gb[1][c - NTHREAD * 2] = gv;
            __syncthreads();
            stage_norm<PRO>(a, sm, m0, tid, v);
        } else {
            stage_chunk<PRO>(a, sm, m0, kc0, kc, tid);
        }
        __syncthreads();
        // (4) MFMA: D[m][n] += X[m][k] * W[n][k]; 4 k-steps per 16-wide block, 2 row tiles
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int kb = wave + j * NWAVE;
            if (kb < nkb) {
                const f4 a0 = *(const f4*)&sm.x[i16][kb * 16 + 4 * q];
                const f4 a1 = *(const f4*)&sm.x[16 + i16][kb * 16 + 4 * q];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bf[j][s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bf[j][s], acc1, 0, 0, 0);
                }
            }
        }
    }
    // (5) fixed-order cross-wave reduction through LDS, then the epilogue: one output per thread
    __syncthreads();
    float* red = &sm.x[0][0];                            // [NWAVE][2][256]
    *(f4*)&red[(wave * 2 + 0) * 256 + lane * 4] = acc0;
    *(f4*)&red[(wave * 2 + 1) * 256 + lane * 4] = acc1;
    __syncthreads();
    const int mt = ml >> 4, row = ml & 15;
    // D layout of mfma 16x16: lane = (row>>2)*16 + col holds D[row][col] in register row&3
    const int off = mt * 256 + ((row >> 2) * 16 + nl) * 4 + (row & 3);
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) s += red[w * 512 + off];
    const int m = m_e, n = n_e;
    float* out = a.out + (long)bz * a.out_bstride;
    if (geglu) {
        // tile rows: [8 value | 8 gate]; lane nl < 8 holds the value of output column 8*nt + nl, lane nl + 8 its gate
        float sb = s + e_bias;
        const float gate = __shfl_xor(sb, 8, 16);
        if (ok_res) {
            float v = sb * es_gelu(gate);
            if (a.res) v += e_res;
            out[(long)m * a.out_ld + nres] = v;
        }
        return;
    }
    const bool ok = ok_e;
    if (ok) {
        if (bias) s += e_bias;
        if (a.act == ES_ACT_RELU) s = fmaxf(s, 0.f);
        else if (a.act == ES_ACT_SILU) s = es_silu(s);
        if (a.res) s += e_res;
        if (a.res2) s += e_res2;
        out[(long)m * a.out_ld + n] = s;
    }
    if (a.out2) {
        // the 16 lanes of a row hold one GroupNorm32 group of the output (N = 512): two-pass statistics by shuffles
        float t = ok ? s : 0.f;
        float sum = t;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float mean = sum * (1.0f / 16.0f);
        const float d = t - mean;
        float sq = d * d;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 16);
        if (ok) {
            float y = d * rsqrtf(sq * (1.0f / 16.0f) + a.gn2_eps) * e_g2 + e_b2;
            if (a.gn2_silu) y = es_silu(y);
            a.out2[(long)m * a.out2_ld + n] = y;
        }
    }
}

__global__ void k_ddpm_update(const es_update_args a) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int st = *a.step;
    if (i < a.n) {
        const float* c = a.coef + (long)st * a.coef_stride;
        const float x = a.x[i], e = a.eps[i];
        const float nz = a.noise[(long)st * a.noise_stride + i];
        const float x0 = c[0] * x - c[1] * e;
        const float mean = c[2] * x0 + c[3] * x;
        a.x[i] = mean + c[4] * nz;
    }
}

__global__ void k_ddim_update(const es_update_args a) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int st = *a.step;
    if (i < a.n) {
        const float* c = a.coef + (long)st * a.coef_stride;
        const float x = a.x[i], e = a.eps[i];
        const float px0 = (x - c[0] * e) / c[1];
        a.x[i] = c[2] * px0 + c[3] * e;
    }
}

__global__ void k_step_inc(int32_t* step) { *step += 1; }

__global__ void k_row_select(const es_rowsel_args a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const float v = a.table[(long)(*a.step) * a.stride + i];
    for (int r = blockIdx.y; r < a.rows; r += gridDim.y) a.out[(long)r * a.out_ld + i] = v;
}

// Box de-normalisation after the layout loop (helpers/util.py:542-568): [-1,1] -> [min,max] for sizes and
// translations (in place, stats = {min_lhw[3], max_lhw[3], min_xyz[3], max_xyz[3], min_angle, max_angle}) and
// (sin, cos) -> arctan2 in degrees-or-radians (scale).
__global__ void k_box_postprocess(float* boxes, int ld, const float* sincos, float* angle_out, const float* stats,
                                  int O, float angle_scale) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= O) return;
    if (boxes) {
        for (int c = 0; c < 6; ++c) {
            const float lo = stats[c < 3 ? c : 3 + c], hi = stats[c < 3 ? 3 + c : 6 + c];
            float v = boxes[(long)i * ld + c];
            v = (v + 1.0f) / 2.0f;
            boxes[(long)i * ld + c] = v * (hi - lo) + lo;
        }
    }
    if (sincos && angle_out) angle_out[i] = atan2f(sincos[2 * i], sincos[2 * i + 1]) * angle_scale;
}

}  // namespace

extern "C" size_t es_pack_linear_f32_size(int N, int K) {
    return (size_t)((N + 15) / 16) * ((K + 15) / 16) * 256;
}

extern "C" int es_pack_linear_f32(const float* w, int N, int K, float* out) {
    const int NT = (N + 15) / 16, KB = (K + 15) / 16;
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 15, q = lane >> 4;
                const int n = nt * 16 + j;
                for (int e = 0; e < 4; ++e) {
                    const int k = kb * 16 + 4 * q + e;
                    out[(((size_t)nt * KB + kb) * 64 + lane) * 4 + e] = (n < N && k < K) ? w[(size_t)n * K + k] : 0.f;
                }
            }
    return 0;
}

// GEGLU variant: W[2*Nh, K] = [value rows | gate rows]  ->  rows interleaved per 16-row tile as 8 value + 8 gate,
// then packed as usual.  h_bias (2*Nh, may be NULL) is permuted into h_bias_out the same way.
extern "C" int es_pack_linear_geglu_f32(const float* w, const float* h_bias, int Nh, int K, float* out, float* h_bias_out) {
    if (Nh % 8) return 2;
    const int N = 2 * Nh;
    float* tmp = (float*)malloc((size_t)N * K * sizeof(float));
    if (!tmp) return 1;
    for (int t = 0; t < N / 16; ++t)
        for (int j = 0; j < 16; ++j) {
            const int src = j < 8 ? t * 8 + j : Nh + t * 8 + (j - 8);
            memcpy(tmp + (size_t)(t * 16 + j) * K, w + (size_t)src * K, (size_t)K * sizeof(float));
            if (h_bias && h_bias_out) h_bias_out[t * 16 + j] = h_bias[src];
        }
    const int rc = es_pack_linear_f32(tmp, N, K, out);
    free(tmp);
    return rc;
}

extern "C" int es_linear_rows_f32(const es_linear_args* a, es_stream stream) {
    ES_REQUIRE(a->nseg >= 1 && a->nseg <= 3, "es_linear_rows_f32: nseg=%d", a->nseg);
    int ksum = 0;
    for (int s = 0; s < a->nseg; ++s) {
        ES_REQUIRE(a->seg[s].width % 4 == 0 && a->seg[s].ld % 4 == 0,
                   "es_linear_rows_f32: segment %d width/ld must be multiples of 4 (width=%d ld=%d)", s,
                   a->seg[s].width, a->seg[s].ld);
        ksum += a->seg[s].width;
    }
    ES_REQUIRE(ksum == a->K, "es_linear_rows_f32: segment widths sum to %d, K=%d", ksum, a->K);
    ES_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, "es_linear_rows_f32: empty problem M=%d N=%d K=%d", a->M, a->N, a->K);
    const bool norm = a->prologue == ES_PRO_GN || a->prologue == ES_PRO_GN_SILU || a->prologue == ES_PRO_LN;
    if (norm) {
        ES_REQUIRE(a->K <= KC && a->K % 64 == 0, "es_linear_rows_f32: norm prologue needs K <= %d and K %% 64 == 0 (K=%d)", KC, a->K);
        ES_REQUIRE(a->gamma && a->beta, "es_linear_rows_f32: norm prologue without affine");
        for (int s = 0; s < a->nseg; ++s)
            ES_REQUIRE(a->seg[s].mode == ES_SEG_DIRECT && a->seg[s].width % 64 == 0,
                       "es_linear_rows_f32: norm prologue needs direct segments with width %% 64 == 0");
        if (a->prologue != ES_PRO_LN)
            ES_REQUIRE(a->K % 128 == 0, "es_linear_rows_f32: GroupNorm32 prologue needs K %% 128 == 0 (K=%d)", a->K);
    }
    if (a->prologue == ES_PRO_GEGLU)
        ES_REQUIRE(a->nseg == 1 && a->seg[0].mode == ES_SEG_DIRECT, "es_linear_rows_f32: GEGLU prologue needs one direct segment");
    const int nb = a->nbatch > 1 ? a->nbatch : 1;
    ES_REQUIRE(nb == 1 || (a->nseg == 1 && a->seg[0].mode == ES_SEG_DIRECT && !norm && !a->res && !a->res2),
               "es_linear_rows_f32: batched launch supports one direct segment, no norm prologue, no residuals");
    ES_REQUIRE(a->act != ES_ACT_GEGLU || (a->N % 16 == 0 && !a->res2), "es_linear_rows_f32: GEGLU epilogue needs N %% 16 == 0");
    ES_REQUIRE(!a->out2 || (a->N == 512 && a->act != ES_ACT_GEGLU && nb == 1 && a->gn2_gamma && a->gn2_beta),
               "es_linear_rows_f32: the GroupNorm32 second output needs N == 512 (N=%d), an affine, no GEGLU, no batching", a->N);
    dim3 grid((a->N + 15) / 16, (a->M + MT - 1) / MT, nb);
    hipStream_t st = (hipStream_t)stream;
    switch (a->prologue) {
        case ES_PRO_NONE: hipLaunchKernelGGL(k_linear_rows<ES_PRO_NONE>, grid, dim3(NTHREAD), 0, st, *a); break;
        case ES_PRO_SILU: hipLaunchKernelGGL(k_linear_rows<ES_PRO_SILU>, grid, dim3(NTHREAD), 0, st, *a); break;
        case ES_PRO_GN: hipLaunchKernelGGL(k_linear_rows<ES_PRO_GN>, grid, dim3(NTHREAD), 0, st, *a); break;
        case ES_PRO_GN_SILU: hipLaunchKernelGGL(k_linear_rows<ES_PRO_GN_SILU>, grid, dim3(NTHREAD), 0, st, *a); break;
        case ES_PRO_LN: hipLaunchKernelGGL(k_linear_rows<ES_PRO_LN>, grid, dim3(NTHREAD), 0, st, *a); break;
        case ES_PRO_GEGLU: hipLaunchKernelGGL(k_linear_rows<ES_PRO_GEGLU>, grid, dim3(NTHREAD), 0, st, *a); break;
        default: ES_REQUIRE(false, "es_linear_rows_f32: unknown prologue %d", a->prologue);
    }
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_row_select(const es_rowsel_args* a, es_stream stream) {
    ES_REQUIRE(a->table && a->step && a->out && a->n > 0 && a->rows > 0, "es_row_select: bad args");
    hipLaunchKernelGGL(k_row_select, dim3((a->n + 255) / 256, a->rows < 64 ? a->rows : 64), dim3(256), 0, (hipStream_t)stream, *a);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_ddpm_update(const es_update_args* a, es_stream stream) {
    ES_REQUIRE(a->n > 0 && a->step && a->noise, "es_ddpm_update: bad args");
    hipLaunchKernelGGL(k_ddpm_update, dim3((a->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    if (a->inc_step) hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, a->step);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_box_postprocess(float* boxes, int ld, const float* sincos, float* angle_out, const float* stats,
                                  int O, float angle_scale, es_stream stream) {
    ES_REQUIRE(O > 0 && (!boxes || (stats && ld >= 6)), "es_box_postprocess: bad args");
    hipLaunchKernelGGL(k_box_postprocess, dim3((O + 63) / 64), dim3(64), 0, (hipStream_t)stream, boxes, ld, sincos,
                       angle_out, stats, O, angle_scale);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_ddim_update(const es_update_args* a, es_stream stream) {
    ES_REQUIRE(a->n > 0 && a->step, "es_ddim_update: bad args");
    hipLaunchKernelGGL(k_ddim_update, dim3((a->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    if (a->inc_step) hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, a->step);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}
