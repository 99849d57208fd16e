// fp32-OPERAND route of the "volume" path (3-D latent-SDF UNet, openai_model_3d.py:816-863; VQ-VAE decoder, vqvae_modules.py:376-409).
//
// The product path (es_vol.hip) multiplies fp16 operands on v_mfma_f32_16x16x32_f16.  The reference is fp32 everywhere, so "the only
// difference is operand rounding" has to be MEASURABLE on the hardware, not argued: this file runs the same plans with fp32 activations
// and fp32 weights on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32 (1/16 of the fp16 rate: 157 TFLOP/s peak,
// MI355X_MICROARCH.md).  It is a VALIDATION route (ShapeDenoiser(precision='fp32'), bench.py --precision fp32): one straightforward
// LDS-tiled implicit-GEMM kernel for every conv / linear shape, a flash-style fp32 attention, and fp32-output variants of the
// normalisation kernels (es_vol.hip, `*_is_f32` flags).  Same op list, same fusion structure, same epilogue order as the fp16 route.
#include "es_common.h"
#include <mutex>
#include <algorithm>

namespace {

struct Geom32 {
    int O, D, H, W;          // output grid
    int Di, Hi, Wi;          // input grid
    int lw, lh, ld;          // log2 of the output grid
};

// source row (voxel index into the channels-last input) of output voxel (o, d, h, w) under tap (kd, kh, kw) in {0,1,2}, or -1 (zero pad)
__device__ __forceinline__ long src_row(const es_conv_args& a, const Geom32& g, int o, int d, int h, int w, int kd, int kh, int kw) {
    int sd, sh, sw;
    if (a.taps == 1) { sd = d; sh = h; sw = w; }
    else if (a.mode == ES_CONV_SAME) { sd = d + kd - 1; sh = h + kh - 1; sw = w + kw - 1; }
    else if (a.mode == ES_CONV_DOWN_HW) { sd = d + kd - 1; sh = 2 * h + kh - 1; sw = 2 * w + kw - 1; }
    else if (a.mode == ES_CONV_DOWN_DHW) { sd = 2 * d + kd - 1; sh = 2 * h + kh - 1; sw = 2 * w + kw - 1; }
    else {
        // nearest-neighbour up-sampling folded into the gather: the tap addresses the UP-SAMPLED grid (= the output grid)
        const int ud = d + kd - 1, uh = h + kh - 1, uw = w + kw - 1;
        if (ud < 0 || ud >= g.D || uh < 0 || uh >= g.H || uw < 0 || uw >= g.W) return -1;
        sd = a.mode == ES_CONV_UP_DHW ? ud >> 1 : ud; sh = uh >> 1; sw = uw >> 1;
        return (((long)o * g.Di + sd) * g.Hi + sh) * g.Wi + sw;
    }
    if (sd < 0 || sd >= g.Di || sh < 0 || sh >= g.Hi || sw < 0 || sw >= g.Wi) return -1;
    return (((long)o * g.Di + sd) * g.Hi + sh) * g.Wi + sw;
}

// ---------------------------------------------------------------------------------------------
// k_conv_f32: out[m][n] = sum_tap sum_c A[src(m, tap)][c] * W[n][tap][c]  (+ the fused 1x1 skip phase a2 / w2) + bias + rowvec + res.
// Workgroup tile 128 rows x 64 columns, 4 waves (32 x 64 each = 2 x 4 MFMA tiles of 16 x 16), K chunk 16 floats: A / B tiles are
// [rows][16 floats] in LDS, a lane's b128 read = its k slot of four consecutive v_mfma_f32_16x16x4_f32 (A and B use the same slot
// assignment: a contraction does not care about the order of its k).  Global -> register -> LDS double buffering.
// Weights: fp32 [N][taps][Cin16] row-major (Cin16 = Cin rounded up to 16, zero filled); activations fp32 channels-last [M][Cin16].
// ---------------------------------------------------------------------------------------------
constexpr int T_BM = 128, T_BN = 64, T_BK = 16;

__global__ __launch_bounds__(256) void k_conv_f32(const es_conv_args a, const Geom32 g, const int ncdhw) {
    __shared__ __attribute__((aligned(16))) float As[2][T_BM * T_BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][T_BN * T_BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long M = (long)g.O * g.D * g.H * g.W;
    const long m0 = (long)blockIdx.x * T_BM;
    const int n0 = blockIdx.y * T_BN;
    const int kch0 = a.Cin >> 4, kch2 = a.a2 ? (a.Cin2 >> 4) : 0;
    const int nk = a.taps * kch0 + kch2;                // K chunks: (tap outer, chunk inner), then the skip phase
    // staging roles: A slots s = tid, tid + 256 (row = s >> 2, float4 c4 = s & 3); B slot tid (row = tid >> 2, c4 = tid & 3)
    int ao[2], ad[2], ah[2], aw[2];
    bool aok[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const long m = m0 + ((tid + 256 * j) >> 2);
        aok[j] = m < M;
        const long mm = aok[j] ? m : 0;
        aw[j] = (int)(mm & (g.W - 1)); ah[j] = (int)((mm >> g.lw) & (g.H - 1));
        ad[j] = (int)((mm >> (g.lw + g.lh)) & (g.D - 1)); ao[j] = (int)(mm >> (g.lw + g.lh + g.ld));
    }
    const int bn = n0 + (tid >> 2);
    const bool bok = bn < a.N;
    f4 ra[2], rb;
    auto gload = [&](int kc) {
        const float* Ap; const float* Wp; int Cs, chunk, tap, ntap;
        if (kc < a.taps * kch0) { tap = kc / kch0; chunk = kc - tap * kch0; Ap = (const float*)a.a; Wp = (const float*)a.w; Cs = a.Cin; ntap = a.taps; }
        else { tap = 0; chunk = kc - a.taps * kch0; Ap = (const float*)a.a2; Wp = (const float*)a.w2; Cs = a.Cin2; ntap = 1; }
        const bool second = kc >= a.taps * kch0;
        const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            long r = -1;
            if (aok[j]) r = second ? ((((long)ao[j] * g.D + ad[j]) * g.H + ah[j]) * g.W + aw[j]) : src_row(a, g, ao[j], ad[j], ah[j], aw[j], kd, kh, kw);
            ra[j] = r >= 0 ? *(const f4*)(Ap + r * Cs + chunk * 16 + ((tid + 256 * j) & 3) * 4) : f4{0.f, 0.f, 0.f, 0.f};
        }
        rb = bok ? *(const f4*)(Wp + ((long)bn * ntap + tap) * Cs + chunk * 16 + (tid & 3) * 4) : f4{0.f, 0.f, 0.f, 0.f};
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) *(f4*)&As[buf][(tid + 256 * j) * 4] = ra[j];
        *(f4*)&Bs[buf][tid * 4] = rb;
    };
    f4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, q = lane >> 4;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < nk) gload(kc + 1);
        f4 af[2], bf[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = *(const f4*)&As[cur][(wave * 32 + i * 16 + i16) * T_BK + q * 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *(const f4*)&Bs[cur][(j * 16 + i16) * T_BK + q * 4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        if (kc + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
    }
    // epilogue: lane holds D[row = q*4 + r][col = i16] of every 16 x 16 tile
    const long V = (long)g.D * g.H * g.W;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = m0 + wave * 32 + i * 16 + q * 4 + r;
            if (m >= M) continue;
            const long o = m / V;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + j * 16 + i16;
                if (n >= a.N) continue;
                float v = acc[i][j][r];
                if (a.bias) v += a.bias[n];
                if (a.rowvec) v += a.rowvec[o * a.rowvec_ld + n];
                if (a.res) v += a.res[m * a.out_ld + n];
                if (ncdhw) a.out_f32[(o * a.N + n) * V + (m - o * V)] = v;
                else a.out_f32[m * a.out_ld + n] = v;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// fp32 self-attention (CrossAttention.forward with context = x, attention.py:172-219): one workgroup = 64 query rows of one
// (batch, head), one thread per query row; K / V tiles of 64 keys in LDS; online softmax, everything in fp32.
// qkv f32 [B*Ntok, 3C] (q | k | v, heads packed (h d)), out f32 [B*Ntok, C].  dhead <= 96 (the UNet's heads: 56 / 84; the VQ-VAE decoder's
// single 256-wide head stays on the fp16 route -- the validation target is the denoiser's latents).
// ---------------------------------------------------------------------------------------------
template <int DMAX>
__global__ __launch_bounds__(64) void k_attention_f32(const es_attn_args a) {
    __shared__ float Ks[64 * (DMAX + 1)], Vs[64 * (DMAX + 1)];
    const int C = a.heads * a.dhead, ldq = 3 * C;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const float* base = (const float*)a.qkv + (long)b * a.Ntok * ldq + h * a.dhead;
    const int row = blockIdx.x * 64 + threadIdx.x;
    const bool rok = row < a.Ntok;
    float qv[DMAX], ov[DMAX];
#pragma unroll
    for (int d = 0; d < DMAX; ++d) { qv[d] = (rok && d < a.dhead) ? base[(long)row * ldq + d] * a.scale : 0.f; ov[d] = 0.f; }
    float mx = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < a.Ntok; k0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * a.dhead; i += 64) {
            const int key = i / a.dhead, d = i - key * a.dhead;
            const bool ok = k0 + key < a.Ntok;
            Ks[key * (DMAX + 1) + d] = ok ? base[(long)(k0 + key) * ldq + C + d] : 0.f;
            Vs[key * (DMAX + 1) + d] = ok ? base[(long)(k0 + key) * ldq + 2 * C + d] : 0.f;
        }
        __syncthreads();
        const int nkey = min(64, a.Ntok - k0);
        for (int key = 0; key < nkey; ++key) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) if (d < a.dhead) s = fmaf(qv[d], Ks[key * (DMAX + 1) + d], s);
            const float mn = fmaxf(mx, s);
            const float al = __expf(mx - mn), p = __expf(s - mn);
            l = l * al + p;
#pragma unroll
            for (int d = 0; d < DMAX; ++d) if (d < a.dhead) ov[d] = fmaf(p, Vs[key * (DMAX + 1) + d], ov[d] * al);
            mx = mn;
        }
    }
    if (rok) {
        float* out = (float*)a.out_f16 + ((long)b * a.Ntok + row) * C + h * a.dhead;
        const float inv = 1.0f / l;
        for (int d = 0; d < a.dhead; ++d) out[d] = ov[d] * inv;
    }
}

// Round 6: the same attention on the exact-fp32 MATRIX instruction (flash style) -- the scalar kernel above was 155 of the 205 ms of a
// 32-object step on the split-operand route (precision 'fp32x'; 25 ms per launch at 1024 tokens x 8 heads x 32 objects).
// One workgroup = 64 query rows of one (batch, head), 4 waves x 16 rows; K / V tiles of 64 keys in LDS (fp32, head dimension padded to
// DP = 64 / 96); S = Q K^T as 4 x DP/4 v_mfma_f32_16x16x4_f32 (A = Q rows from registers, B = K rows from LDS), online softmax on the D
// layout (a lane holds rows 4 q + r of key column i16: row maxima / sums are 16-lane reductions), P through a private LDS slab into
// the A layout, O += P V as 16 x DP/16 MFMAs.  fp32 products, fp32 accumulation, fp32 softmax: the reference's arithmetic.
template <int DP>
__global__ __launch_bounds__(256) void k_attention_f32m(const es_attn_args a) {
    constexpr int KT = 64, LDK = DP + 4, LDP = KT + 4, NKK = DP / 4, NCT = DP / 16;
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float* Ks = smf;
    float* Vs = Ks + KT * LDK;
    float* Ps = Vs + KT * LDK;                    // [4 waves][16 rows][LDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, q = lane >> 4;
    const int C = a.heads * a.dhead, ldq = 3 * C;
    const int bh = blockIdx.y, b = bh / a.heads, h = bh - b * a.heads;
    const float* base = (const float*)a.qkv + (long)b * a.Ntok * ldq + h * a.dhead;
    const int row0 = blockIdx.x * 64 + wave * 16;
    float qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int d = 4 * kk + q;
        qf[kk] = (row0 + i16 < a.Ntok && d < a.dhead) ? base[(long)(row0 + i16) * ldq + d] * a.scale : 0.f;
    }
    f4 o[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) o[ct] = f4{0.f, 0.f, 0.f, 0.f};
    float mrow[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, lrow[4] = {0.f, 0.f, 0.f, 0.f};
    float* Pw = Ps + wave * 16 * LDP;
    for (int k0 = 0; k0 < a.Ntok; k0 += KT) {
        __syncthreads();                                                      // every wave is done with the previous K / V tile
        for (int i = tid; i < KT * NKK; i += 256) {
            const int key = i / NKK, d4 = (i - key * NKK) * 4;
            f4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (k0 + key < a.Ntok && d4 < a.dhead) {                          // (dhead % 4 == 0: a quad is inside or outside)
                const float* p = base + (long)(k0 + key) * ldq + d4;
                kv = *(const f4*)(p + C);
                vv = *(const f4*)(p + 2 * C);
            }
            *(f4*)&Ks[key * LDK + d4] = kv;
            *(f4*)&Vs[key * LDK + d4] = vv;
        }
        __syncthreads();
        f4 s[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            s[st] = f4{0.f, 0.f, 0.f, 0.f};
            const float* kr = Ks + (st * 16 + i16) * LDK + q;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) s[st] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kk], kr[4 * kk], s[st], 0, 0, 0);
            if (k0 + st * 16 + i16 >= a.Ntok) s[st] = f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // keys past the end
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = fmaxf(fmaxf(s[0][r], s[1][r]), fmaxf(s[2][r], s[3][r]));
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
            const float mn = fmaxf(mrow[r], mx);
            const float al = __expf(mrow[r] - mn);
            float ps = 0.f;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const float p = __expf(s[st][r] - mn);
                ps += p;
                Pw[(q * 4 + r) * LDP + st * 16 + i16] = p;
            }
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) ps += __shfl_xor(ps, sh);
            lrow[r] = lrow[r] * al + ps;
            mrow[r] = mn;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) o[ct][r] *= al;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // own P slab written
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int kk = 0; kk < KT / 4; ++kk) {
            const float pa = Pw[i16 * LDP + 4 * kk + q];
            const float* vr = Vs + (4 * kk + q) * LDK + i16;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) o[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, vr[ct * 16], o[ct], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
    float* out = (float*)a.out_f16 + (long)b * a.Ntok * C + h * a.dhead;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = row0 + q * 4 + r;
        const float inv = 1.0f / lrow[r];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int d = ct * 16 + i16;
            if (row < a.Ntok && d < a.dhead) out[(long)row * C + d] = o[ct][r] * inv;
        }
    }
}

int ilog2x(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace

// fp32 [N][taps][Cin16] image of a PyTorch conv / linear weight [N, CinW, kd, kh, kw] (or [N, CinW])
extern "C" size_t es_pack_conv_f32_size(int N, int CinW, int taps) { return (size_t)N * taps * ((CinW + 15) / 16 * 16); }
extern "C" int es_pack_conv_f32(const float* h_w, int N, int CinW, int taps, float* h_out) {
    const int Cin = (CinW + 15) / 16 * 16;
    for (int n = 0; n < N; ++n)
        for (int t = 0; t < taps; ++t)
            for (int c = 0; c < Cin; ++c)
                h_out[((size_t)n * taps + t) * Cin + c] = c < CinW ? h_w[((size_t)n * CinW + c) * taps + t] : 0.f;
    return 0;
}

extern "C" int es_conv_f32(const es_conv_args* a, es_stream stream) {
    ES_REQUIRE(a->Cin % 16 == 0 && a->Cin > 0, "es_conv_f32: Cin=%d must be a positive multiple of 16", a->Cin);
    ES_REQUIRE(a->taps == 27 || a->taps == 1, "es_conv_f32: taps=%d", a->taps);
    ES_REQUIRE(!a->a2 || (a->Cin2 % 16 == 0 && a->Cin2 > 0), "es_conv_f32: Cin2=%d", a->Cin2);
    ES_REQUIRE(a->out_f32 && !a->out_f16 && a->epilogue == ES_EPI_NONE && !a->gn_stats_out, "es_conv_f32: fp32 output only, no fused GEGLU / statistics");
    Geom32 g;
    g.O = a->O; g.D = a->D; g.H = a->H; g.W = a->W;
    g.Di = a->D; g.Hi = a->H; g.Wi = a->W;
    if (a->mode == ES_CONV_DOWN_HW) { g.Hi = 2 * a->H; g.Wi = 2 * a->W; }
    if (a->mode == ES_CONV_DOWN_DHW) { g.Di = 2 * a->D; g.Hi = 2 * a->H; g.Wi = 2 * a->W; }
    if (a->mode == ES_CONV_UP_HW) { g.Hi = a->H / 2; g.Wi = a->W / 2; }
    if (a->mode == ES_CONV_UP_DHW) { g.Di = a->D / 2; g.Hi = a->H / 2; g.Wi = a->W / 2; }
    g.lw = ilog2x(a->W); g.lh = ilog2x(a->H); g.ld = ilog2x(a->D);
    ES_REQUIRE(g.lw >= 0 && g.lh >= 0 && g.ld >= 0, "es_conv_f32: D,H,W must be powers of two (%d,%d,%d)", a->D, a->H, a->W);
    const int ncdhw = a->out_ld < 0 ? 1 : 0;
    ES_REQUIRE(!ncdhw || !a->res, "es_conv_f32: NCDHW output takes no residual");
    const long M = (long)a->O * a->D * a->H * a->W;
    dim3 grid((unsigned)((M + T_BM - 1) / T_BM), (unsigned)((a->N + T_BN - 1) / T_BN));
    hipLaunchKernelGGL(k_conv_f32, grid, dim3(256), 0, (hipStream_t)stream, *a, g, ncdhw);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_attention_f32(const es_attn_args* a, es_stream stream) {
    ES_REQUIRE(a->dhead > 0 && a->dhead <= 96, "es_attention_f32: dhead=%d (<= 96)", a->dhead);
    dim3 grid((a->Ntok + 63) / 64, a->B * a->heads);
    hipStream_t st = (hipStream_t)stream;
    if (a->dhead % 4 == 0) {
        // round 6: the matrix-instruction kernel (dhead a multiple of 4: the K / V tiles are staged in 16-byte quads)
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            attr_err = hipFuncSetAttribute((const void*)k_attention_f32m<96>, hipFuncAttributeMaxDynamicSharedMemorySize, (2 * 64 * 100 + 4 * 16 * 68) * 4);
        });
        ES_REQUIRE(attr_err == hipSuccess, "es_attention_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
        if (a->dhead <= 64) hipLaunchKernelGGL((k_attention_f32m<64>), grid, dim3(256), (2 * 64 * 68 + 4 * 16 * 68) * 4, st, *a);
        else hipLaunchKernelGGL((k_attention_f32m<96>), grid, dim3(256), (2 * 64 * 100 + 4 * 16 * 68) * 4, st, *a);
        ES_CHECK_HIP(hipGetLastError());
        return 0;
    }
    if (a->dhead <= 64) hipLaunchKernelGGL((k_attention_f32<64>), grid, dim3(64), 0, st, *a);
    else hipLaunchKernelGGL((k_attention_f32<96>), grid, dim3(64), 0, st, *a);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}
