// "rows" path: fp32 fused linear over small-M node/triple matrices + the diffusion updates.
//
// Roofline note: one layout denoising step streams ~335 MB of fp32 weights for ~6.5 GFLOP --
// HBM-bound on paper (42 us at 8 TB/s), in practice a chain of ~130 DEPENDENT [32 x K] @ [K x N] products whose cost is
// the kernel boundary (~1.5 us) plus each kernel's own latency chain.  Round 2 gave one workgroup per 16 output columns
// (N = 512 -> 32 workgroups on a 256-CU chip), each staging and normalising the WHOLE [32 x K] activation tile:
// 6-28 us per launch, 2.6 % of the HBM roofline.  Round 3 design:
//   * K is split over workgroups as well: grid = (column tiles x K slices, row tiles).  A slice's partial products go
//     to its own SLAB of the output, out[s][M][N] (slice 0 adds bias and residuals); there is no reduction kernel and
//     no in-launch hand-off -- the CONSUMER sums the slabs in fixed order while it stages its A operand (the
//     launch-boundary reduce of the MI355X guide: costs nothing extra because that staging exists anyway).  Every
//     operand that can be a slab tensor (A segments, the residual, the eps input of the DDPM update) carries
//     (nslab, slab_stride);
//   * prologues are PER SEGMENT and applied to the slice only: a workgroup normalises 32 x (K / S) elements instead of
//     32 x K (GroupNorm groups never straddle a slice; LayerNorm reads its whole rows for the statistics -- L2 hits --
//     and normalises the slice);
//   * the choice of S depends on (K, N) only, never on M: per-row arithmetic is independent of the batch
//     (collated scenes == single scenes, bit for bit);
//   * weights are pre-packed in MFMA-fragment order: a wave-level load is one contiguous 1 KiB global_load_dwordx4
//     that lands in B-operand registers (streamed once per workgroup, no LDS round trip), issued before the staging;
//   * exact fp32 on the matrix pipe: v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain); the k-blocks of a slice are
//     dealt to the 8 waves (NWAVE) and reduced through LDS in a fixed order (deterministic, no float atomics).
// Round 5: the product launches run on k_rows_x (es_rows_x.h; family 1, the default) -- the same decomposition, slabs, K cuts and
// fixed-order reductions, but no staging tile: a wave loads its A fragments straight into MFMA operand registers and applies the
// slab sum / ReLU / GroupNorm / LayerNorm prologue there (lane swaps instead of LDS), every load of a workgroup is issued
// unconditionally behind ONE 64-byte per-slice descriptor (zero-record buffer descriptors gate what does not exist), slices are cut at
// SEGMENT boundaries (a workgroup never multiplies two segments), and a ninth wave pulls the NEXT launch's weights into this XCD's
// L2.  Measured by wave stamps (profiles/r05_rows_stamps_*.txt): these kernels are bound by the number of dependent VALU
// instructions in front of the first MFMA (~5 clocks each at one wave per SIMD), not by bytes.  The kernel below (k_linear_rows,
// family 0) remains for the CSR pooling launch, the tables built once per schedule and A/B timing (es_rows_set_kernel_family).
#include "es_common.h"
#include <cstdlib>
#include <mutex>
#include <initializer_list>

namespace {

// Row tile = ONE 16-row MFMA tile.  The slab traffic of a launch is (column tiles) x S x M x K x 4 bytes whatever the row tile
// (every workgroup re-reads its [MT x K/S] operand from all S slabs of the producer): 16 MB per [32 x 512] x [512 x 512] launch
// at S = 8, and that L2 traffic -- not latency -- was the staging cost (1.3-2 us per 64 columns).  Halving the row tile gives the
// same 256 workgroups at S = 4, i.e. half the bytes.
constexpr int MT = 16;          // rows per workgroup
constexpr int KCH = 1024;       // K columns of a slice staged in LDS at a time
constexpr int NWAVE = 8;        // wave w owns the k-blocks w, w + 8, ... of the chunk
constexpr int NKG = 8;
constexpr int NTHREAD = NWAVE * 64;
constexpr int MAXJ = KCH / 16 / NKG;     // 16-wide k-blocks per wave per chunk
constexpr int LPR = NTHREAD / MT;        // lanes per row while staging (32: four 128-byte lines per row and pass)

__device__ __forceinline__ const float* seg_base(const es_seg& s) {
    const float* p = s.ptr;   // (batched launches add blockIdx.z * a_bstride to segment 0 at the call site)
    if (s.step) p += (long)(*s.step) * s.step_stride;
    return p;
}

__device__ __forceinline__ float load_slabs1(const float* p, int nslab, long slab_stride) {
    float v = *p;
    for (int j = 1; j < nslab; ++j) v += p[(long)j * slab_stride];
    return v;
}
// The same sum with the first four loads in flight at once and no control flow: slab j >= nslab re-reads the last slab (same cache
// line) and is not added.  load_slabs1's run-time loop made every slab of an epilogue operand a dependent round trip IN FRONT of the weight and
// staging loads of the slice-0 workgroups (a residual with 4 slabs: four serial L2 / Infinity-Cache round trips before anything else
// was issued).  Same additions in the same order: the same bits.
__device__ __forceinline__ void issue_slabs1(float (&r)[4], const float* p, int nslab, long slab_stride) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = p[(long)(j < nslab ? j : nslab - 1) * slab_stride];
}
__device__ __forceinline__ float sum_slabs1(const float (&r)[4], const float* p, int nslab, long slab_stride) {
    float v = r[0];
#pragma unroll
    for (int j = 1; j < 4; ++j) v = j < nslab ? v + r[j] : v;
    for (int j = 4; j < nslab; ++j) v += p[(long)j * slab_stride];      // (more than 4 slabs: not produced by the shipped planner)
    return v;
}

// NS float4 loads of one slab tensor element issued together (slabs >= nslab are wave-uniformly skipped), summed in the fixed
// order 0 .. nslab-1 by sum_slabs: the deferred split-K reduction of the producer.  These launches are pure latency chains: a
// runtime loop over the slabs made every slab a dependent L2 round trip (8.6 us per split launch instead of ~4).
template <int NS>
__device__ __forceinline__ void load_slabs(f4 (&t)[NS], const float* p, int nslab, long slab_stride) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
        t[j] = f4{0.f, 0.f, 0.f, 0.f};
        if (j == 0 || j < nslab) t[j] = *(const f4*)(p + (long)j * slab_stride);
    }
}
template <int NS>
__device__ __forceinline__ f4 sum_slabs(const f4 (&t)[NS], int nslab) {
    f4 v = t[0];
#pragma unroll
    for (int j = 1; j < NS; ++j) if (j < nslab) v += t[j];
    return v;
}

// Staging of the columns [c0, c0 + kc) of the (virtually concatenated) A operand into LDS tile x[MT][ldx], with each
// segment's slab sum, input activation and prologue applied.  Thread mapping (512 threads, 16-row tile): row r = tid >> 5, lane
// cl = tid & 31 (LPR = 32 lanes per row); a thread walks its row in steps of 32 float4 (8 lanes read one 128-byte line); 16 float4 loads (columns x slabs) are in flight
// per thread before the first is used.  Segment loop / mode switches are wave-uniform.  GroupNorm: a group (gs = 4 .. 32
// channels) is gs/4 adjacent lanes of one pass -> statistics by lane shuffles, two-pass (mean, then squared deviations).
// NS: compile-time bound of the slab counts of this launch's segments (1, 2, 4, 8).
// PROC: prologue class compiled in -- 0: none (raw / ReLU'd operands), 1: + GroupNorm(+SiLU) segments, 2: + SiLU / GEGLU
// (tables, tests).  CSR: the segmented-mean gather is compiled in.  A launch costs ~4 us of which ~1 us was instruction fetch
// when every path lived in one 31 KB kernel (each launch starts with a cold instruction cache): one lean kernel per class.
// U1: ONE pass per batch (UB = 1) -- the variant for launches of MORE than one round of workgroups (a rider on a GCN launch: 352
// workgroups): 87-120 VGPRs instead of 106-194, i.e. two workgroups per CU instead of one, at the price of one load batch per
// 128 columns of a slice (the riders' slices ARE 128 columns).  Same loads, same arithmetic, same order: the same bits.
template <int NS, int PROC, bool CSR, bool U1 = false>
__device__ __forceinline__ void stage_chunk(const es_linear_args& a, float* x, int ldx, int m0, int c0, int kc, int tid) {
    // passes (32 lanes x float4 = 128 columns of a row) per batch: up to 16 float4 loads in flight per thread.  A pass that lies
    // wholly past the region is skipped (wave-uniform): in the 4-wave version with 32-load batches the clamped duplicates of a 64-column
    // slice cost +1.9 us per launch -- at one wave per SIMD nothing hides the dependent VALU chain of the prologue.
    constexpr int UB = U1 ? 1 : NS >= 8 ? 2 : 4;
    const int r = tid >> 5, cl = tid & (LPR - 1);
    const int m = m0 + r;
    const bool row_ok = m < a.M;
    const int mc = row_ok ? m : a.M - 1;
    int koff = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const es_seg& sg = a.seg[s];
        const int lo = max(c0, koff), hi = min(min(c0 + kc, koff + sg.width), a.K);
        if (lo < hi) {
            const int w4 = (hi - lo) >> 2;
            const int scol = lo - koff;                                 // first column inside the segment
            const float* base = seg_base(sg) + scol + (s == 0 ? (long)blockIdx.z * a.a_bstride : 0);
            float* dst = x + r * ldx + (lo - c0);
            const int nslab = sg.nslab > 1 ? sg.nslab : 1;
            const long sstr = sg.slab_stride;
            const int pro = sg.pro;
            const bool relu_in = sg.pre_act == ES_ACT_RELU;
            if (CSR && (sg.mode == ES_SEG_CSRMEAN || sg.mode == ES_SEG_CSRSUM || sg.mode == ES_SEG_CSRWAVG)) {
                const int e0 = sg.idx[mc], e1 = sg.idx[mc + 1];
                const bool wavg = sg.mode == ES_SEG_CSRWAVG;
                float wden = 1.0f;
                if (wavg) {
                    // pooling='wAvg' (graph.py:178-184): the weights of the node's entries summed in the reference's scatter_add
                    // order -- the object slots first, then the subject slots, each in triple order -- plus 1e-4
                    float ws = 0.f;
                    for (int e = e0; e < e1; ++e) if (sg.ent_off[e] != 0) ws += sg.ent_wt[2 * sg.ent_row[e] + 1];
                    for (int e = e0; e < e1; ++e) if (sg.ent_off[e] == 0) ws += sg.ent_wt[2 * sg.ent_row[e]];
                    wden = ws + 0.0001f;
                }
                for (int c4 = cl; c4 < w4; c4 += LPR) {
                    f4 v = {0.f, 0.f, 0.f, 0.f};
                    if (!wavg && nslab == 1) {
                        // The sampling path's pooling ('avg' / 'sum', one slab).  Round 5: the scene node's row (~2*O entries) was a chain
                        // of 8 batches x 2 dependent round trips (index pairs, then rows) = 4 us of this launch's 7.8 (wave stamps).  Now
                        // 12 entries per batch (16 spill), and the index pairs of batch b + 1 are requested BEHIND the rows of batch b (loads
                        // retire in order: the additions of batch b wait for its rows only) -- one round trip per 12 entries.  Same
                        // entries, added in the same (stored == the reference's scatter_add) order: the same bits.
                        constexpr int EB = 12;
                        int off[EB];
#pragma unroll
                        for (int u = 0; u < EB; ++u) {
                            const int ee = min(e0 + u, e1 - 1);
                            off[u] = e0 < e1 ? sg.ent_row[ee] * sg.ld + sg.ent_off[ee] : 0;
                        }
                        for (int e = e0; e < e1; e += EB) {
                            f4 t[EB];
#pragma unroll
                            for (int u = 0; u < EB; ++u) t[u] = *(const f4*)(base + off[u] + 4 * c4);
                            if (e + EB < e1) {
#pragma unroll
                                for (int u = 0; u < EB; ++u) {
                                    const int ee = min(e + EB + u, e1 - 1);
                                    off[u] = sg.ent_row[ee] * sg.ld + sg.ent_off[ee];     // (the row loads above have been issued: their address registers are free)
                                }
                            }
#pragma unroll
                            for (int u = 0; u < EB; ++u) {
                                if (relu_in) {
#pragma unroll
                                    for (int q = 0; q < 4; ++q) t[u][q] = fmaxf(t[u][q], 0.f);
                                }
                                if (e + u < e1) v += t[u];
                            }
                        }
                    } else
                    // stored order == scatter_add order of the reference.  Entries are fetched 8 at a time (index pairs,
                    // then rows) and added in order (the scene node has ~2*O incident edges).
                    for (int e = e0; e < e1; e += 8) {
                        long off[8];
                        f4 t[8];
                        float wt[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int ee = min(e + u, e1 - 1);
                            off[u] = (long)sg.ent_row[ee] * sg.ld + sg.ent_off[ee];
                            wt[u] = wavg ? sg.ent_wt[2 * sg.ent_row[ee] + (sg.ent_off[ee] != 0 ? 1 : 0)] : 1.0f;
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) t[u] = *(const f4*)(base + off[u] + 4 * c4);
                        for (int j = 1; j < nslab; ++j) {               // (slab sources under a CSR mean: not on the sampling path)
#pragma unroll
                            for (int u = 0; u < 8; ++u) t[u] += *(const f4*)(base + off[u] + 4 * c4 + (long)j * sstr);
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            if (relu_in) {
#pragma unroll
                                for (int q = 0; q < 4; ++q) t[u][q] = fmaxf(t[u][q], 0.f);
                            }
                            if (wavg) {                                  // s_weights * new_s_vecs is a rounded product, THEN scatter_add (graph.py:169-177)
#pragma unroll
                                for (int q = 0; q < 4; ++q) t[u][q] = __fmul_rn(wt[u], t[u][q]);
                            }
                            if (e + u < e1) v += t[u];
                        }
                    }
                    const float den = sg.mode == ES_SEG_CSRMEAN ? (float)max(e1 - e0, 1) : wavg ? wden : 1.0f;   // 'avg' pooling divides by the clamped count
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = row_ok ? v[q] / den : 0.f;
                    *(f4*)(dst + 4 * c4) = v;
                }
            } else {
                const long rowoff = (long)(sg.mode == ES_SEG_GATHER ? sg.idx[mc] : mc) * sg.ld;
                const float* src = base + rowoff;
                const bool gn = PROC >= 1 && (pro == ES_PRO_GN || pro == ES_PRO_GN_SILU);
                const int lpg = gn ? (sg.gs >> 2) : 1;                  // lanes per GroupNorm group (1, 2, 4, 8)
                const float* ga = sg.gamma ? sg.gamma + scol : src;
                const float* be = sg.beta ? sg.beta + scol : src;
                for (int u0 = 0; u0 * LPR < w4; u0 += UB) {
                    f4 t[UB][NS], gt[PROC >= 2 ? UB : 1][NS], gav[PROC >= 1 ? UB : 1], bev[PROC >= 1 ? UB : 1];
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        if (LPR * (u0 + u) >= w4) continue;
                        const int c4 = min(cl + LPR * (u0 + u), w4 - 1);
                        load_slabs<NS>(t[u], src + 4 * c4, nslab, sstr);
                        if (PROC >= 2 && pro == ES_PRO_GEGLU) load_slabs<NS>(gt[PROC >= 2 ? u : 0], src + a.K + 4 * c4, nslab, sstr);
                        if (PROC >= 1 && gn) { gav[PROC >= 1 ? u : 0] = *(const f4*)(ga + 4 * c4); bev[PROC >= 1 ? u : 0] = *(const f4*)(be + 4 * c4); }
                    }
#pragma unroll
                    for (int u = 0; u < UB; ++u) {
                        if (LPR * (u0 + u) >= w4) continue;
                        const int c4 = cl + LPR * (u0 + u);
                        f4 y = sum_slabs<NS>(t[u], nslab);
                        if (relu_in) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) y[q] = fmaxf(y[q], 0.f);
                        }
                        if (PROC >= 2 && pro == ES_PRO_GEGLU) {
                            const f4 g = sum_slabs<NS>(gt[PROC >= 2 ? u : 0], nslab);
#pragma unroll
                            for (int q = 0; q < 4; ++q) y[q] = y[q] * es_gelu(g[q]);
                        } else if (PROC >= 2 && pro == ES_PRO_SILU) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) y[q] = es_silu(y[q]);
                        } else if (PROC >= 1 && gn) {
                            // (lanes past the region hold clamped duplicates of the last float4: their groups are whole
                            //  duplicates too, because w4 is a multiple of lpg -- no foreign value enters a real group)
                            float sm = (y[0] + y[1]) + (y[2] + y[3]);
                            for (int o = 1; o < lpg; o <<= 1) sm += __shfl_xor(sm, o, LPR);
                            const float inv_gs = __builtin_amdgcn_rcpf((float)sg.gs);   // gs is a power of two: v_rcp is exact there, the multiplies below are exact divisions
                            const float mean = sm * inv_gs;
                            float sq = 0.f;
#pragma unroll
                            for (int q = 0; q < 4; ++q) { const float d = y[q] - mean; sq += d * d; }
                            for (int o = 1; o < lpg; o <<= 1) sq += __shfl_xor(sq, o, LPR);
                            const float rstd = __builtin_amdgcn_rsqf(sq * inv_gs + sg.eps);      // v_rsq_f32 (1 ulp) instead of sqrt + quotient (~25 instructions)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                float z = (y[q] - mean) * rstd;
                                if (sg.gamma) z = z * gav[PROC >= 1 ? u : 0][q] + bev[PROC >= 1 ? u : 0][q];      // (NULL: affine folded into the weights)
                                if (pro == ES_PRO_GN_SILU) z = es_silu(z);
                                y[q] = z;
                            }
                        }
                        if (c4 < w4) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) y[q] = row_ok ? y[q] : 0.f;
                            *(f4*)(dst + 4 * c4) = y;
                        }
                    }
                }
            }
        }
        koff += sg.width;
    }
    // zero the K padding (K not a multiple of 16)
    const int kend = min(c0 + kc, a.K) - c0;
    for (int c = kend + cl; c < kc; c += LPR) x[r * ldx + c] = 0.f;
}

// LayerNorm prologue: the operand is ONE direct segment of width K <= 128 * LNU.  Thread (r, cl) fetches its LNU float4 of
// the WHOLE row (all slabs, all in flight), keeps them in registers for the statistics (two passes, 32-lane shuffles) and writes
// the normalised columns of the slice [c0, c0 + kc) to the LDS tile -- the statistics need the row, the product only the slice.
template <int NS, int LNU>
__device__ __forceinline__ void stage_ln(const es_linear_args& a, float* x, int ldx, int m0, int c0, int kc, int tid) {
    const int r = tid >> 5, cl = tid & (LPR - 1);
    const int m = m0 + r;
    const bool row_ok = m < a.M;
    const int mc = row_ok ? m : a.M - 1;
    const es_seg& sg = a.seg[0];
    const int K = sg.width, w4 = K >> 2;
    const float* src = seg_base(sg) + (long)blockIdx.z * a.a_bstride + (long)mc * sg.ld;
    const int nslab = sg.nslab > 1 ? sg.nslab : 1;
    f4 v[LNU];
    {
        f4 t[LNU][NS];
#pragma unroll
        for (int u = 0; u < LNU; ++u) load_slabs<NS>(t[u], src + 4 * min(cl + LPR * u, w4 - 1), nslab, sg.slab_stride);
#pragma unroll
        for (int u = 0; u < LNU; ++u) v[u] = sum_slabs<NS>(t[u], nslab);
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < LNU; ++u) if (cl + LPR * u < w4) s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
    const float mean = s / (float)K;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < LNU; ++u) {
        if (cl + LPR * u < w4) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[u][e] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, LPR);
    const float rstd = __builtin_amdgcn_rsqf(q / (float)K + sg.eps);
#pragma unroll
    for (int u = 0; u < LNU; ++u) {
        const int c = 4 * (cl + LPR * u);
        if (c >= c0 && c < c0 + kc && c < K) {
            f4 ga = {1.f, 1.f, 1.f, 1.f}, be = {0.f, 0.f, 0.f, 0.f};
            if (sg.gamma) { ga = *(const f4*)(sg.gamma + c); be = *(const f4*)(sg.beta + c); }     // (NULL: affine folded into the weights)
            f4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = row_ok ? (sg.gamma ? (v[u][e] - mean) * rstd * ga[e] + be[e] : (v[u][e] - mean) * rstd) : 0.f;
            *(f4*)&x[r * ldx + (c - c0)] = y;
        }
    }
    const int kend = min(c0 + kc, a.K) - c0;
    for (int c = kend + cl; c < kc; c += LPR) x[r * ldx + c] = 0.f;
}

// grid.x = column tiles x K slices (slice fastest: with the observed block -> XCD b % 8 placement the column tiles of one
// slice share an XCD, i.e. one L2 copy of that slice of A -- speed only), grid.y = row tiles, grid.z = batch.
// One launch = up to 3 INDEPENDENT problems (es_linear_rows_multi_f32): problem i owns the blockIdx.x range [wg0[i], wg0[i+1]) and
// the first ny[i] row tiles.  A problem with FEWER row tiles than the launch (a node-row product riding on a triple-row launch:
// 2 row tiles against 8) is FOLDED: its xw[i] x ny[i] tiles are laid row-major over fw[i] columns of ALL gridDim.y rows, so that
// the launch carries no workgroups that exit at entry (768 of 1024 for a 512-column product on an 8-row launch).
struct RowsLaunch {
    es_linear_args p[3];
    int S[3], kbps[3], wg0[3], ny[3];
    int xw[3], fw[3];            // tiles per row of the problem (column tiles x slices); folded width (0 = not folded)
    int n, ldx, dbg;
#ifdef ES_STAMP
    unsigned long long* stamp;   // tools/rows_stamps.py: 8 x 100 MHz wall-clock ticks per workgroup (wave 0), slot = launch_id * 1024 + workgroup
    int launch_id;
#endif
};

#ifdef ES_STAMP
#define ES_RSTAMP(k) do { if (L.stamp) st_[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ES_RSTAMP(k) do { } while (0)
#endif

// NS: bound of the slab counts of the A segments; LNU > 0: LayerNorm prologue over rows of up to 128 * LNU columns (PROC / CSR
// unused); GEGLU_EPI: the value * gelu(gate) epilogue is compiled in.
// NT = 2: a workgroup owns TWO adjacent 16-column tiles (column tiles 2t, 2t + 1) of its 16 rows: ONE staged / normalised A tile, two
// weight streams, two accumulators, threads 256..511 (idle in the one-tile epilogue) finalise the second tile.  For the GEGLU
// projection behind a LayerNorm (N = 4096: 512 workgroups each normalising its 16 x 512 rows from two slabs -> 256): half the
// staging traffic and LayerNorm work per output.  Per output the k-blocks, their order and the reduction are those of NT = 1.
template <int NS, int PROC, bool CSR, int LNU, bool GEGLU_EPI, bool U1 = false, int NT = 1>
__global__ __launch_bounds__(NTHREAD) void k_linear_rows(const RowsLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef ES_STAMP
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    st_[0] = __builtin_amdgcn_s_memrealtime();
#endif
    kernarg_warm<sizeof(RowsLaunch)>();
    ES_RSTAMP(1);
    const int ldx = L.ldx, dbg = L.dbg;
    float* x = smem;                                     // [MT][ldx]
    float* red = smem + MT * ldx;                        // [NT][NKG][256]
    if (dbg & 4) return;                                 // (calibration of the launch floor, tools/microbench_rows.py)
    const int pi = (L.n > 1 && (int)blockIdx.x >= L.wg0[1] ? 1 : 0) + (L.n > 2 && (int)blockIdx.x >= L.wg0[2] ? 1 : 0);
    const es_linear_args& a = L.p[pi];
    int bx = (int)blockIdx.x - L.wg0[pi], by = (int)blockIdx.y;
    if (L.fw[pi] > 0) {                                  // folded problem: tile v of xw x ny, row-major over fw columns x gridDim.y rows
        const int v = by * L.fw[pi] + bx, xw = L.xw[pi];
        if (v >= xw * L.ny[pi]) return;
        by = v / xw;
        bx = v - by * xw;
    } else if (by >= L.ny[pi]) return;
    const int S = L.S[pi], kbps = L.kbps[pi];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slice = bx % S, nt = (bx / S) * NT;        // (first) column tile of the workgroup
    const int m0 = by * MT;
    const int nkb_total = (a.K + 15) >> 4;
    const int kb0 = slice * kbps, kb1 = slice == S - 1 ? nkb_total : kb0 + kbps;
    const int bz = blockIdx.z;                           // batched launch: z-th problem of identical shape
    const int nct = (a.N + 15) >> 4;
    const f4* wp = (const f4*)a.wpack + ((size_t)bz * nct + nt) * nkb_total * 64;
    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    const int i16 = lane & 15, q = lane >> 4;

    // (0) Epilogue operands depend only on the output coordinates: issued FIRST so that their latency overlaps the weight
    // stream, the staging and the product.  Thread (ml, nl), tid < 256, owns output (ml, nl) of the 16 x 16 tile.
    const int ml = (tid >> 4) & 15, nl = tid & 15;
    const int te = NT > 1 ? (tid >> 8) : 0;              // the tile this thread finalises (NT = 1: threads 256.. take no part)
    const int nt_e = nt + te;
    const int n_e = nt_e * 16 + nl, m_e = m0 + ml;
    const bool geglu = GEGLU_EPI && a.act == ES_ACT_GEGLU;
    const float* bias = a.bias ? a.bias + (long)bz * a.N : nullptr;
    const int nres = geglu ? nt_e * 8 + nl : n_e;        // column of the residual / output
    const bool first = slice == 0;                       // slice 0 carries bias and residuals
    float e_bias = 0.f, e_res = 0.f, e_res2 = 0.f;
    const bool ok_e = tid < 256 * NT && m_e < a.M && n_e < a.N;
    const bool ok_res = geglu ? (ok_e && nl < 8) : ok_e;
    float r1[4] = {0.f, 0.f, 0.f, 0.f}, r2[4] = {0.f, 0.f, 0.f, 0.f};
    const int ns1 = a.res_nslab > 1 ? a.res_nslab : 1, ns2 = a.res2_nslab > 1 ? a.res2_nslab : 1;
    const bool has1 = first && a.res && ok_res && !(dbg & 8), has2 = first && a.res2 && ok_res && !(dbg & 8);
    const float* const pr1 = a.res + (long)m_e * a.res_ld + nres + (a.res_step ? (long)(*a.res_step) * a.res_step_stride : 0);
    const float* const pr2 = a.res2 + (long)m_e * a.res2_ld + nres;
    if (has1) issue_slabs1(r1, pr1, ns1, a.res_slab_stride);        // summed in the epilogue
    if (has2) issue_slabs1(r2, pr2, ns2, a.res2_slab_stride);
    if (first && bias && n_e < a.N && !(dbg & 8)) e_bias = bias[n_e];
    const int kg = wave;

    for (int c0 = kb0 * 16; c0 < kb1 * 16; c0 += KCH) {
        const int kc = min(KCH, kb1 * 16 - c0);
        const int nkb = kc >> 4;
        // (1) issue this wave's weight-fragment loads first: HBM latency overlaps the staging below
        f4 bf[NT][MAXJ];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int kb = kg + j * NKG;
                bf[t][j] = f4{1.f, 1.f, 1.f, 1.f};
                if (kb < nkb && !(dbg & 1))
                    bf[t][j] = __builtin_nontemporal_load(&wp[(size_t)(t * nkb_total + (c0 >> 4) + kb) * 64 + lane]);
            }
        }
        // (2) stage the activation slice with its prologue applied
        if (c0 > kb0 * 16) __syncthreads();
        if (dbg & 2) { }
        else if (LNU > 0) stage_ln<NS, (LNU > 0 ? LNU : 1)>(a, x, ldx, m0, c0, kc, tid);
        else stage_chunk<NS, PROC, CSR, U1>(a, x, ldx, m0, c0, kc, tid);
        if (c0 == kb0 * 16) ES_RSTAMP(2);
        __syncthreads();
        if (c0 == kb0 * 16) ES_RSTAMP(3);
        // (3) MFMA: D[m][n] += X[m][k] * W[n][k]; 4 k-steps per 16-wide block
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int kb = kg + j * NKG;
            if (kb < nkb) {
                const f4 a0 = *(const f4*)&x[i16 * ldx + kb * 16 + 4 * q];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bf[t][j][s], acc[t], 0, 0, 0);
                }
            }
        }
    }
    // (4) fixed-order reduction over the k-block groups through LDS, then the epilogue: one output per thread
#pragma unroll
    for (int t = 0; t < NT; ++t) *(f4*)&red[(t * NKG + wave) * 256 + lane * 4] = acc[t];
    ES_RSTAMP(4);
    __syncthreads();
    ES_RSTAMP(5);
    // D layout of mfma 16x16: lane = (row>>2)*16 + col holds D[row][col] in register row&3
    const int off = ((ml >> 2) * 16 + nl) * 4 + (ml & 3);
    float sres = 0.f;
#pragma unroll
    for (int w = 0; w < NKG; ++w) sres += red[(te * NKG + w) * 256 + off];
    if (has1) e_res = sum_slabs1(r1, pr1, ns1, a.res_slab_stride);
    if (has2) e_res2 = sum_slabs1(r2, pr2, ns2, a.res2_slab_stride);
    float* out = a.out + (long)bz * a.out_bstride + (long)slice * a.out_slab_stride;
    if (GEGLU_EPI && geglu) {
        // tile rows: [8 value | 8 gate]; lane nl < 8 holds the value of output column 8*nt_e + nl, lane nl + 8 its gate
        float sb = sres + e_bias;
        const float gate = __shfl_xor(sb, 8, 16);
        if (ok_res) {
            float v = sb * es_gelu(gate);
            if (a.res) v += e_res;
            out[(long)m_e * a.out_ld + nres] = v;
        }
#ifdef ES_STAMP
        if (L.stamp && tid == 0) {
            st_[6] = __builtin_amdgcn_s_memrealtime();
            st_[7] = __builtin_amdgcn_s_getreg((3 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
            unsigned long long* d = L.stamp + ((size_t)L.launch_id * 1024 + (blockIdx.y * gridDim.x + blockIdx.x)) * 8;
            for (int k = 0; k < 8; ++k) d[k] = st_[k];
        }
#endif
        return;
    }
    if (ok_e) {
        if (first) {
            if (bias) sres += e_bias;
            if (a.act == ES_ACT_RELU) sres = fmaxf(sres, 0.f);
            else if (a.act == ES_ACT_SILU) sres = es_silu(sres);
            else if (a.act == ES_ACT_SIGMOID) sres = 1.0f / (1.0f + expf(-sres));
            if (a.res) sres += e_res;
            if (a.res2) sres += e_res2;
        }
        out[(long)m_e * a.out_ld + n_e] = sres;
    }
#ifdef ES_STAMP
    if (L.stamp && tid == 0 && blockIdx.y * gridDim.x + blockIdx.x < 1024) {
        st_[6] = __builtin_amdgcn_s_memrealtime();
        // XCC_ID (hwreg 20, 4 bits) | HW_ID (hwreg 4, 32 bits) << 32
        st_[7] = __builtin_amdgcn_s_getreg((3 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
        unsigned long long* d = L.stamp + ((size_t)L.launch_id * 1024 + (blockIdx.y * gridDim.x + blockIdx.x)) * 8;
        for (int k = 0; k < 8; ++k) d[k] = st_[k];
    }
#endif
}

#include "es_rows_x.h"

// ONE_BLOCK: the whole state in one workgroup (n <= 4096: one scene) -- the step counter is advanced by the same kernel, after every
// thread has read it (a launch less per step of the latency-bound layout loop)
template <bool ONE_BLOCK>
__global__ void k_ddpm_update(const es_update_args a) {
#pragma clang fp contract(off)
    const int st = *a.step;
    const float* c = a.coef + (long)st * a.coef_stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += ONE_BLOCK ? (int)blockDim.x : a.n) {
        const float x = a.x[i], e = load_slabs1(a.eps + i, a.eps_nslab > 1 ? a.eps_nslab : 1, a.eps_slab_stride);
        const float nz = a.noise[(long)st * a.noise_stride + i];
        float x0 = c[0] * x - c[1] * e;
        if (a.clip_x0) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);          // clip_denoised=True: torch.clamp(x_recon, -1, 1), diffusion_ddpm.py:243-244
        const float mean = c[2] * x0 + c[3] * x;
        a.x[i] = mean + c[4] * nz;
    }
    if (ONE_BLOCK && a.inc_step) {
        __syncthreads();
        if (threadIdx.x == 0) *a.step = st + 1;
    }
}

__global__ void k_ddim_update(const es_update_args a) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int st = *a.step;
    if (i < a.n) {
        const float* c = a.coef + (long)st * a.coef_stride;
        const float x = a.x[i], e = load_slabs1(a.eps + i, a.eps_nslab > 1 ? a.eps_nslab : 1, a.eps_slab_stride);
        const float px0 = (x - c[0] * e) / c[1];
        float xn = c[2] * px0 + c[3] * e;
        // eta != 0 (samplers/ddim.py:256-260): + sigma_t * randn; c[3] then already is sqrt(1 - a_prev - sigma_t^2), c[4] = sigma_t
        if (a.noise) xn = xn + c[4] * a.noise[(long)st * a.noise_stride + i];
        a.x[i] = xn;
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
// ncol = 7: also the angle column (descale_box_params(angle=True), helpers/util.py:553-555: stats[12], stats[13]).
__global__ void k_box_postprocess(float* boxes, int ld, const float* sincos, float* angle_out, const float* stats,
                                  int O, float angle_scale, int ncol) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= O) return;
    if (boxes) {
        for (int c = 0; c < ncol; ++c) {
            const float lo = stats[c < 3 ? c : c < 6 ? 3 + c : 12], hi = stats[c < 3 ? 3 + c : c < 6 ? 6 + c : 13];
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
    es_parallel_for(NT, [=](long nt) {
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 15, q = lane >> 4;
                const int n = (int)nt * 16 + j;
                for (int e = 0; e < 4; ++e) {
                    const int k = kb * 16 + 4 * q + e;
                    out[(((size_t)nt * KB + kb) * 64 + lane) * 4 + e] = (n < N && k < K) ? w[(size_t)n * K + k] : 0.f;
                }
            }
    });
    return 0;
}

// the same image formed on the device from the fp32 weight already in HBM (one float4 per thread)
__global__ __launch_bounds__(256) void k_pack_linear_f32(const float* __restrict__ w, int N, int K, int KB, f4* __restrict__ out, long nvec) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    const int lane = (int)(i & 63);
    const long tb = i >> 6;
    const int kb = (int)(tb % KB);
    const long nt = tb / KB;
    const int j = lane & 15, q = lane >> 4;
    const long n = nt * 16 + j;
    f4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int k = kb * 16 + 4 * q + e;
        v[e] = (n < N && k < K) ? w[(size_t)n * K + k] : 0.f;
    }
    out[i] = v;
}

extern "C" int es_pack_linear_f32_dev(const float* d_w, int N, int K, float* d_out, es_stream stream) {
    ES_REQUIRE(d_w && d_out && N > 0 && K > 0, "es_pack_linear_f32_dev: N=%d K=%d", N, K);
    const int NT = (N + 15) / 16, KB = (K + 15) / 16;
    const long nvec = (long)NT * KB * 64;
    hipLaunchKernelGGL(k_pack_linear_f32, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_w, N, K, KB, (f4*)d_out, nvec);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

// fp64 product for the planners' weight folds (to_out . to_v, proj_out . ff2, W . beta ...) when the model's parameters already live
// on the GPU: C[N, M] = A[N, K] B[K, M], every element a left fold over k = 0..K-1 of fma(a, b, acc) -- one fixed order whatever the
// shape or the launch, so two processes (object shards) fold bit-identical weights; a BLAS call gives no such promise.
// 64x64 tile per workgroup, 4x4 outputs per thread, 16-deep LDS panels.
__global__ __launch_bounds__(256) void k_matmul_f64(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ Cm, int N, int K, int M) {
    __shared__ double sa[16][64 + 1], sb[16][64];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    double acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int r = e >> 4, kk = e & 15;                      // A panel: 64 rows x 16 k (k fastest in memory)
            sa[kk][r] = (i0 + r < N && k0 + kk < K) ? A[(long)(i0 + r) * K + k0 + kk] : 0.0;
            const int kb = e >> 6, c = e & 63;                      // B panel: 16 k x 64 columns
            sb[kb][c] = (k0 + kb < K && j0 + c < M) ? B[(long)(k0 + kb) * M + j0 + c] : 0.0;
        }
        __syncthreads();
        const int kn = K - k0 < 16 ? K - k0 : 16;                   // (the zero padding is never added: the fold stops at K)
        for (int kk = 0; kk < kn; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = sa[kk][ty * 4 + u]; b[u] = sb[kk][tx * 4 + u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[u][v] = fma(a[u], b[v], acc[u][v]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = i0 + ty * 4 + u, j = j0 + tx * 4 + v;
            if (i < N && j < M) Cm[(long)i * M + j] = acc[u][v];
        }
}

extern "C" int es_matmul_f64(const double* d_a, const double* d_b, double* d_c, int N, int K, int M, es_stream stream) {
    ES_REQUIRE(d_a && d_b && d_c && N > 0 && K > 0 && M > 0, "es_matmul_f64: N=%d K=%d M=%d", N, K, M);
    hipLaunchKernelGGL(k_matmul_f64, dim3((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64)), dim3(256), 0, (hipStream_t)stream, d_a, d_b, d_c, N, K, M);
    ES_CHECK_HIP(hipGetLastError());
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

// K-slice choice.  It depends on (K, N) only -- never on M -- so that a row's arithmetic is independent of the batch it is part of.
// kb_per_slice = 0: automatic when the op allows a split (no activation / GEGLU epilogue), else one slice.
extern "C" int es_linear_rows_slices(const es_linear_args* a, int* kb_per_slice) {
    const int nkb = (a->K + 15) / 16;
    int kalign = 1;                                       // slices are cut at multiples of the largest GroupNorm group
    for (int s = 0; s < a->nseg; ++s) {
        const int pro = a->seg[s].pro ? a->seg[s].pro : a->prologue;
        if (pro == ES_PRO_GN || pro == ES_PRO_GN_SILU) {
            const int gs = a->seg[s].gs ? a->seg[s].gs : a->K / 32;
            if (gs > 16 * kalign) kalign = (gs + 15) / 16;
        }
    }
    int kbps = a->kb_per_slice;
    if (kbps <= 0 || kbps >= nkb) { *kb_per_slice = nkb; return 1; }
    kbps = (kbps + kalign - 1) / kalign * kalign;
    if (kbps >= nkb) { *kb_per_slice = nkb; return 1; }
    *kb_per_slice = kbps;
    if (a->seg_slices) {
        // round 5: slices never straddle two segments -- every segment (a multiple of 16 columns wide) is cut into
        // ceil(width / (16 kbps)) slices of kbps k-blocks, the last one of a segment possibly shorter
        // bits 1..3 of seg_slices: segment 0..2 is cut with HALF the slice length (a GroupNorm segment next to plain ones: its
        // k-blocks cost a wave ~2.5x a plain block)
        int S = 0;
        for (int s = 0; s < a->nseg; ++s) {
            if (a->seg[s].width % 16) return -1;
            const bool half = (a->seg_slices >> (1 + s)) & 1;
            const int kb = half ? (kbps + 1) / 2 : kbps;
            if (half) {
                // a half-length slice of a GroupNorm segment must still hold whole groups (the kernel pairs k-blocks (j, j ^ 1) of a
                // 32-channel group): -2 = misaligned (the planner checks the same before it sets the bit)
                const int pro = a->seg[s].pro ? a->seg[s].pro : a->prologue;
                const int gs = a->seg[s].gs ? a->seg[s].gs : a->K / 32;
                if ((pro == ES_PRO_GN || pro == ES_PRO_GN_SILU) && kb % ((gs + 15) / 16) != 0) return -2;
            }
            S += (a->seg[s].width / 16 + kb - 1) / kb;
        }
        return S;
    }
    return (nkb + kbps - 1) / kbps;
}

extern "C" int es_linear_rows_auto_slices(int K, int N, int kalign_cols) {
    // ~256 workgroups for TWO 16-row tiles (M = 32, one scene), slices of at least 128 columns (one k-block per wave), <= 8 slabs
    // (a constant of the build, not an environment switch: the slice count decides where a K sum is cut, i.e. the fp32 bits.
    //  Round-3 A/B, layout steps/s: target 128 -> 949, 256 -> 1093, 512 -> 958)
    constexpr int target = 256;
    const int nkb = (K + 15) / 16, nct = (N + 15) / 16;
    int S = target / (2 * nct);
    if (S > 8) S = 8;
    if (S <= 1) return 0;
    int kbps = (nkb + S - 1) / S;
    if (kbps < 8) kbps = 8;
    const int kal = kalign_cols > 16 ? (kalign_cols + 15) / 16 : 1;
    kbps = (kbps + kal - 1) / kal * kal;
    return kbps >= nkb ? 0 : kbps;
}

#ifdef ES_STAMP
static unsigned long long* g_rows_stamp = nullptr;
static int g_rows_launch_id = 0;
static FILE* g_rows_log = nullptr;
// stamps of every rows launch enqueued (or captured) from now on: slot = launch order; `log` (may be NULL): one line per launch
extern "C" int es_debug_rows_stamp(void* p, const char* log) {
    g_rows_stamp = (unsigned long long*)p;
    g_rows_launch_id = 0;
    if (g_rows_log) { fclose(g_rows_log); g_rows_log = nullptr; }
    if (p && log) g_rows_log = fopen(log, "w");
    return 0;
}
#endif

static thread_local const es_linear_args* g_rows_next = nullptr;
void es_rows_hint_next(const es_linear_args* next) { g_rows_next = next; }

namespace {
struct RowsPrep { es_linear_args a; int S, kbps, nsmax, proc, nb; bool has_ln, csr, gepi, lnattn; };

// validation + normalisation of one problem (op-level prologue -> segments; slice choice)
int rows_prepare(const es_linear_args* a_in, RowsPrep* out) {
    es_linear_args a = *a_in;
    ES_REQUIRE(a.nseg >= 1 && a.nseg <= 3, "es_linear_rows_f32: nseg=%d", a.nseg);
    int ksum = 0;
    for (int s = 0; s < a.nseg; ++s) {
        ES_REQUIRE(a.seg[s].width % 4 == 0 && a.seg[s].ld % 4 == 0,
                   "es_linear_rows_f32: segment %d width/ld must be multiples of 4 (width=%d ld=%d)", s,
                   a.seg[s].width, a.seg[s].ld);
        ES_REQUIRE(a.seg[s].nslab <= 1 || a.seg[s].slab_stride % 4 == 0, "es_linear_rows_f32: segment %d slab stride %d", s, a.seg[s].slab_stride);
        ES_REQUIRE(a.seg[s].mode >= ES_SEG_DIRECT && a.seg[s].mode <= ES_SEG_CSRWAVG, "es_linear_rows_f32: segment %d mode %d", s, a.seg[s].mode);
        ES_REQUIRE(a.seg[s].mode != ES_SEG_CSRWAVG || (a.seg[s].ent_wt && a.seg[s].idx && a.seg[s].ent_row && a.seg[s].ent_off),
                   "es_linear_rows_f32: segment %d: weighted pooling needs row pointers, entries and the weight matrix", s);
        ksum += a.seg[s].width;
    }
    ES_REQUIRE(a.act >= ES_ACT_NONE && a.act <= ES_ACT_SIGMOID, "es_linear_rows_f32: epilogue activation %d", a.act);
    ES_REQUIRE(ksum == a.K, "es_linear_rows_f32: segment widths sum to %d, K=%d", ksum, a.K);
    ES_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "es_linear_rows_f32: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
    // the op-level prologue is shorthand for "every segment": GroupNorm32 / LayerNorm over the concatenation
    if (a.prologue != ES_PRO_NONE) {
        const bool norm = a.prologue == ES_PRO_GN || a.prologue == ES_PRO_GN_SILU || a.prologue == ES_PRO_LN;
        if (norm) ES_REQUIRE(a.gamma && a.beta, "es_linear_rows_f32: norm prologue without affine");
        if (a.prologue == ES_PRO_GN || a.prologue == ES_PRO_GN_SILU)
            ES_REQUIRE(a.K % 128 == 0, "es_linear_rows_f32: GroupNorm32 prologue needs K %% 128 == 0 (K=%d)", a.K);
        int koff = 0;
        for (int s = 0; s < a.nseg; ++s) {
            ES_REQUIRE(a.seg[s].pro == ES_PRO_NONE, "es_linear_rows_f32: op-level and segment-level prologues are exclusive");
            a.seg[s].pro = a.prologue;
            if (norm) {
                a.seg[s].gamma = a.gamma + koff; a.seg[s].beta = a.beta + koff; a.seg[s].eps = a.eps;
                a.seg[s].gs = a.prologue == ES_PRO_LN ? a.K : a.K / 32;
            }
            koff += a.seg[s].width;
        }
    }
    bool has_ln = false, lnattn = false;
    {
        int koff = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const es_seg& sg = a.seg[s];
            if (sg.pro == ES_PRO_GN || sg.pro == ES_PRO_GN_SILU) {
                ES_REQUIRE(sg.mode == ES_SEG_DIRECT && (sg.gamma != nullptr) == (sg.beta != nullptr), "es_linear_rows_f32: GroupNorm segments are direct, with gamma AND beta or neither (affine folded into the weights)");
                ES_REQUIRE(sg.gs >= 4 && sg.gs <= 32 && (sg.gs & (sg.gs - 1)) == 0 && sg.width % sg.gs == 0 && koff % sg.gs == 0,
                           "es_linear_rows_f32: GroupNorm group size %d (4..32, power of two, dividing the segment width %d and its offset %d)",
                           sg.gs, sg.width, koff);
            } else if (sg.pro == ES_PRO_LN) {
                ES_REQUIRE(a.nseg == 1 && sg.mode == ES_SEG_DIRECT && (sg.gamma != nullptr) == (sg.beta != nullptr) && sg.width <= 1024,
                           "es_linear_rows_f32: LayerNorm prologue needs ONE direct segment of width <= 1024 (gamma AND beta, or neither: affine folded into the weights)");
                has_ln = true;
            } else if (sg.pro == ES_PRO_LN_ATTN) {
                // LayerNorm over a row formed from [t0 | u], two vectors and the cross-attention vector (echoscene_hip.h, es_seg.pro)
                ES_REQUIRE(a.nseg == 1 && sg.mode == ES_SEG_DIRECT && !sg.gamma && !sg.beta && sg.width <= 512 && sg.width % 16 == 0 &&
                           sg.gs >= sg.width && sg.gs % 4 == 0 && a.res && a.res2 && a.res_nslab <= 1 && a.res2_nslab <= 1 &&
                           a.res_ld % 4 == 0 && a.res2_ld % 4 == 0 && !a.res_step,
                           "es_linear_rows_f32: ES_PRO_LN_ATTN needs ONE direct segment of width <= 512, no affine vectors, u at gs >= width columns, "
                           "and plain res (output) / res2 (cross-attention vector) matrices");
                has_ln = true; lnattn = true;
            } else if (sg.pro == ES_PRO_GEGLU) {
                ES_REQUIRE(a.nseg == 1 && sg.mode == ES_SEG_DIRECT, "es_linear_rows_f32: GEGLU prologue needs one direct segment");
            } else {
                ES_REQUIRE(sg.pro == ES_PRO_NONE || sg.pro == ES_PRO_SILU, "es_linear_rows_f32: unknown prologue %d", sg.pro);
            }
            ES_REQUIRE(sg.pre_act == ES_ACT_NONE || sg.pre_act == ES_ACT_RELU, "es_linear_rows_f32: segment input activation %d", sg.pre_act);
            koff += sg.width;
        }
    }
    const int nb = a.nbatch > 1 ? a.nbatch : 1;
    ES_REQUIRE(nb == 1 || (a.nseg == 1 && a.seg[0].mode == ES_SEG_DIRECT && a.seg[0].pro == ES_PRO_NONE && !a.res && !a.res2),
               "es_linear_rows_f32: batched launch supports one direct segment, no prologue, no residuals");
    ES_REQUIRE(a.act != ES_ACT_GEGLU || (a.N % 16 == 0 && (!a.res2 || lnattn)), "es_linear_rows_f32: GEGLU epilogue needs N %% 16 == 0");
    int kbps = 0;
    const int S = es_linear_rows_slices(&a, &kbps);
    ES_REQUIRE(S != -2, "es_linear_rows_f32: a half-length slice (seg_slices bits 1..3, kb_per_slice=%d) would cut a GroupNorm group", a.kb_per_slice);
    ES_REQUIRE(S >= 1, "es_linear_rows_f32: segment-aligned slices need segment widths that are multiples of 16");
    ES_REQUIRE(S == 1 || (a.act == ES_ACT_NONE && nb == 1 && a.out_slab_stride >= a.M * a.out_ld),
               "es_linear_rows_f32: a K split (%d slices) needs no activation epilogue, no batching and out_slab_stride >= M * out_ld", S);
    int nsmax = 1;
    for (int s = 0; s < a.nseg; ++s) if (a.seg[s].nslab > nsmax) nsmax = a.seg[s].nslab;
    ES_REQUIRE(nsmax <= 8, "es_linear_rows_f32: a segment with %d slabs (max 8)", nsmax);
    ES_REQUIRE(!has_ln || nsmax <= 2, "es_linear_rows_f32: the LayerNorm operand may have at most 2 slabs (%d)", nsmax);
    int proc = 0;
    bool csr = false;
    for (int s = 0; s < a.nseg; ++s) {
        const int pro = a.seg[s].pro;
        if (pro == ES_PRO_GN || pro == ES_PRO_GN_SILU) proc = proc < 1 ? 1 : proc;
        if (pro == ES_PRO_SILU || pro == ES_PRO_GEGLU) proc = 2;
        csr = csr || a.seg[s].mode == ES_SEG_CSRMEAN || a.seg[s].mode == ES_SEG_CSRSUM || a.seg[s].mode == ES_SEG_CSRWAVG;
    }
    out->a = a; out->S = S; out->kbps = kbps; out->nsmax = nsmax; out->proc = proc; out->nb = nb;
    out->has_ln = has_ln; out->csr = csr; out->gepi = a.act == ES_ACT_GEGLU; out->lnattn = lnattn;
    return 0;
}

// ---- k_rows_x dispatch (es_rows_x.h) ---------------------------------------------------------------------------------------------
int g_rows_family = 1;       // 1 = k_rows_x where it applies (default), 0 = k_linear_rows only (es_rows_set_kernel_family: A/B tools)

struct XPlan { int S, Jw, cut[XMAXS + 1], segof[XMAXS], jws[XMAXS]; bool uniform; };

// The K slices of a problem as k_rows_x wants them (every slice inside ONE segment); false = not a problem for k_rows_x
bool x_plan(const RowsPrep& p, XPlan* xp) {
    const es_linear_args& a = p.a;
    if (p.nb != 1 || p.csr || a.K % 16) return false;
    if (p.gepi && !p.has_ln) return false;
    if (a.res_nslab > XMAXS || a.res2_nslab > XMAXS) return false;
    int S = 0, koff = 0, maxn = 0;
    for (int s = 0; s < a.nseg; ++s) {
        const es_seg& sg = a.seg[s];
        if (sg.mode != ES_SEG_DIRECT && sg.mode != ES_SEG_GATHER) return false;
        if (sg.pro != ES_PRO_NONE && sg.pro != ES_PRO_GN && sg.pro != ES_PRO_GN_SILU && sg.pro != ES_PRO_LN && sg.pro != ES_PRO_LN_ATTN) return false;
        if (sg.step || sg.nslab > 6 || sg.width % 16) return false;
        if ((sg.pro == ES_PRO_GN || sg.pro == ES_PRO_GN_SILU) && sg.gs < 4) return false;
        koff += sg.width;
    }
    {
        // k_rows_x addresses its operands with 32-bit byte offsets inside a 2 GiB buffer window: an operand whose extent
        // reaches 2^31 bytes stays on k_linear_rows (64-bit pointers).  Gathered segments index node-level tables (a few thousand rows).
        const long lim = (1L << 31) - 4096;
        for (int s = 0; s < a.nseg; ++s) {
            const es_seg& sg = a.seg[s];
            const long rows = sg.mode == ES_SEG_DIRECT ? a.M : 0;
            const long ext = (rows * sg.ld + sg.width) * 4 + (long)(sg.nslab > 1 ? sg.nslab - 1 : 0) * sg.slab_stride * 4;
            if (ext >= lim) return false;
        }
        const long oext = ((long)a.M * a.out_ld + a.N) * 4;
        if (oext >= lim || (a.res && ((long)a.M * a.res_ld + a.N) * 4 >= lim) || (a.res2 && ((long)a.M * a.res2_ld + a.N) * 4 >= lim)) return false;
    }
    const int nkb = a.K / 16;
    if (a.seg_slices) {
        int kb = 0;
        for (int s = 0; s < a.nseg; ++s) {
            const int w = a.seg[s].width / 16;
            const int kbs = (a.seg_slices >> (1 + s)) & 1 ? (p.kbps + 1) / 2 : p.kbps;
            for (int c = 0; c < w; c += kbs) {
                if (S >= XMAXS) return false;
                xp->cut[S] = kb + c; xp->segof[S] = s;
                const int n = (w - c < kbs) ? w - c : kbs;
                maxn = n > maxn ? n : maxn;
                ++S;
            }
            kb += w;
        }
        xp->cut[S] = nkb;
    } else {
        // legacy uniform cuts: usable when no slice straddles a segment boundary
        S = p.S;
        if (S > XMAXS) return false;
        for (int i = 0; i < S; ++i) {
            const int k0 = i * p.kbps, k1 = (i + 1) * p.kbps < nkb ? (i + 1) * p.kbps : nkb;
            int so = -1, c0 = 0;
            for (int s = 0; s < a.nseg; ++s) {
                const int c1 = c0 + a.seg[s].width / 16;
                if (k0 >= c0 && k1 <= c1) so = s;
                c0 = c1;
            }
            if (so < 0) return false;
            xp->cut[i] = k0; xp->segof[i] = so;
            maxn = (k1 - k0) > maxn ? (k1 - k0) : maxn;
        }
        xp->cut[S] = nkb;
    }
    if (S != p.S) return false;
    int Jw = (maxn + NKG - 1) / NKG;
    bool gs32 = false;
    for (int s = 0; s < a.nseg; ++s) gs32 = gs32 || ((a.seg[s].pro == ES_PRO_GN || a.seg[s].pro == ES_PRO_GN_SILU) && a.seg[s].gs == 32);
    if (gs32 && (Jw & 1)) ++Jw;
    xp->uniform = true;
    for (int i = 0; i + 1 < S; ++i) xp->uniform = xp->uniform && (xp->cut[i + 1] - xp->cut[i] == xp->cut[1] - xp->cut[0]);
    if (p.lnattn && S != 1) return false;
    if (p.has_ln) {
        if (S > 2 || S * Jw > 4 || p.nsmax > 2) return false;
        if (S == 2 && xp->cut[2] - xp->cut[1] != xp->cut[1] - xp->cut[0]) return false;
    } else if (Jw > 12) return false;
    xp->S = S; xp->Jw = Jw;
    // per slice: k-blocks per wave (an even count under 32-channel GroupNorm groups: a group is two adjacent blocks of one wave)
    for (int i = 0; i < S; ++i) {
        const es_seg& sg = a.seg[xp->segof[i]];
        int j = (xp->cut[i + 1] - xp->cut[i] + NKG - 1) / NKG;
        if ((sg.pro == ES_PRO_GN || sg.pro == ES_PRO_GN_SILU) && sg.gs == 32 && (j & 1)) ++j;
        xp->jws[i] = p.has_ln ? Jw : j;
    }
    return true;
}

void x_fill(XProb& P, const RowsPrep& pr, const XPlan& xp) {
    const es_linear_args& a = pr.a;
    memset(&P, 0, sizeof(P));
    P.wpack = a.wpack; P.bias = a.bias; P.res = a.res; P.res2 = a.res2; P.out = a.out;
    P.res_step = a.res_step; P.res_step_stride = a.res_step_stride;
    P.M = a.M; P.N = a.N; P.inv_k = 1.0f / (float)a.K; P.nkb_total = a.K / 16; P.S = xp.S | (((32768 + xp.S - 1) / xp.S) << 16); P.Jw = xp.Jw; P.act = a.act;
    P.res_ld = a.res_ld; P.res_nslab = a.res_nslab > 1 ? a.res_nslab : 1; P.res_sstr = a.res_slab_stride;
    P.res2_ld = a.res2_ld; P.res2_nslab = a.res2_nslab > 1 ? a.res2_nslab : 1; P.res2_sstr = a.res2_slab_stride;
    if (pr.lnattn) P.res_nslab = P.res2_nslab = 0;        // res: the formed row's output, res2: an operand of the prologue -- the epilogue adds neither
    P.out_ld = a.out_ld; P.out_sstr = a.out_slab_stride;
    int segc0[3] = {0, 0, 0};
    for (int s = 1; s < a.nseg; ++s) segc0[s] = segc0[s - 1] + a.seg[s - 1].width;
    for (int i = 0; i < xp.S; ++i) {
        const es_seg& sg = a.seg[xp.segof[i]];
        XSlice& d = P.sl[i];
        const int col = xp.cut[i] * 16 - segc0[xp.segof[i]];          // first column of the slice inside its segment
        const bool ln = sg.pro == ES_PRO_LN || sg.pro == ES_PRO_LN_ATTN;
        d.a = sg.ptr + (ln ? 0 : col);
        d.idx = sg.mode == ES_SEG_GATHER ? sg.idx : nullptr;
        d.gamma = sg.gamma ? sg.gamma + (ln ? 0 : col) : nullptr;
        d.beta = sg.beta ? sg.beta + (ln ? 0 : col) : nullptr;
        d.ld = sg.ld; d.nslab = sg.nslab > 1 ? sg.nslab : 1; d.sstr = sg.slab_stride;
        d.flags = (sg.mode == ES_SEG_GATHER ? 1 : 0) | ((sg.pro == ES_PRO_GN || sg.pro == ES_PRO_GN_SILU) ? 2 : 0) |
                  (sg.pro == ES_PRO_GN_SILU ? 4 : 0) | (ln ? 8 : 0) | (sg.pre_act == ES_ACT_RELU ? 16 : 0) | (xp.jws[i] << 8);
        d.gs = sg.gs; d.eps = sg.eps; d.nkb = xp.cut[i + 1] - xp.cut[i]; d.kb0 = xp.cut[i];
    }
}

// the prefetch descriptor of the hinted next launch (w == NULL when there is none or it is not a single-problem k_rows_x launch)
XPre x_prefetch_of_next() {
    XPre pf;
    memset(&pf, 0, sizeof(pf));
    const es_linear_args* nx = g_rows_next;
    g_rows_next = nullptr;
    static const char* pre_env = getenv("ES_ROWS_PREFETCH");        // timing-only: 0 = no prefetch wave
    if (!nx || (pre_env && atoi(pre_env) == 0)) return pf;
    RowsPrep pr;
    if (rows_prepare(nx, &pr)) return pf;
    XPlan xp;
    if (!x_plan(pr, &xp)) return pf;
    const int nct = (pr.a.N + 15) / 16;
    const int nt = (pr.has_ln && pr.gepi && xp.S == 1 && nct % 2 == 0) ? 2 : 1;
    pf.w = (const char*)pr.a.wpack;
    pf.gx = nct / nt * xp.S;
    pf.smagic = xp.S | (((32768 + xp.S - 1) / xp.S) << 16);
    pf.nkb_total = pr.a.K / 16;
    pf.nt = nt;
    for (int i = 0; i <= xp.S; ++i) pf.cut[i] = xp.cut[i];
    if (pf.gx >= 5461) pf.w = nullptr;
    return pf;
}

template <int NP>
int x_launch_np(const void* fn, const XLaunch<NP>& L, dim3 grid, es_stream stream) {
    void* kargs[] = {(void*)&L};
    ES_CHECK_HIP(hipLaunchKernel(fn, grid, dim3(NTHREAD + (L.pf.w ? 64 : 0)), kargs, 0, (hipStream_t)stream));
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

// -1: not launched (the caller falls back to k_linear_rows)
int x_launch(const RowsPrep* pr, int n, es_stream stream, const RowsLaunch& RL, int gx, int gy) {
    const bool dry = n == 0;                              // n == 0: would pr[0] launch alone?  -2 = yes, -1 = no
    if (dry) n = 1;
    XPlan xp[3];
    int jmax = 1, nsmax = 1, proc = 0;
    bool has_ln = false;
    for (int i = 0; i < n; ++i) {
        if (!x_plan(pr[i], &xp[i])) return -1;
        jmax = xp[i].Jw > jmax ? xp[i].Jw : jmax;
        nsmax = pr[i].nsmax > nsmax ? pr[i].nsmax : nsmax;
        proc = pr[i].proc > proc ? pr[i].proc : proc;
        has_ln = has_ln || pr[i].has_ln;
    }
    if (proc > 1) return -1;
    bool gather = false;
    for (int i = 0; i < n; ++i) for (int sgi = 0; sgi < pr[i].a.nseg; ++sgi) gather = gather || pr[i].a.seg[sgi].mode == ES_SEG_GATHER;
    // kernel classes by load budget (k-blocks per wave, slabs): (2, 6), (4, 2), (8, 1)
    int cls = -1;
    if (jmax <= 2 && nsmax <= 4) cls = 0;
    else if (jmax <= 4 && nsmax <= 4) cls = 1;
    else if (jmax <= 8 && nsmax <= 2) cls = 2;
    else if (jmax <= 12 && nsmax <= 1 && (n == 1 || gather)) cls = 3;          // (the cross-attention-vector product: K 1280 in one slice)
    const void* fn = nullptr;
    dim3 grid;
    int nt = 1;
    if (has_ln) {
        if (n != 1 || gather) return -1;
        const RowsPrep& p0 = pr[0];
        const bool nt2 = p0.gepi && xp[0].S == 1 && ((p0.a.N + 15) / 16) % 2 == 0;
        nt = nt2 ? 2 : 1;
        if (p0.lnattn) fn = nt2 ? (const void*)k_rows_x<4, 2, 3, 1, 2, true, 1> : (const void*)k_rows_x<4, 2, 3, 1, 1, true, 1>;
        else if (xp[0].S == 2) fn = (const void*)k_rows_x<2, 2, 2, 2, 1, true, 1>;
        else fn = nt2 ? (const void*)k_rows_x<4, 2, 2, 1, 2, true, 1> : (const void*)k_rows_x<4, 2, 2, 1, 1, true, 1>;
    } else {
        if (cls < 0) return -1;
        // [class][GroupNorm compiled in]; the third class (long single-segment K ranges: the feed-forward output product, the
        // triple-row products of the GCNs) has no GroupNorm variant
        static const void* const tab1[4][2] = {
            {(const void*)k_rows_x<2, 4, 0, 1, 1, false, 1>, (const void*)k_rows_x<2, 4, 1, 1, 1, false, 1>},
            {(const void*)k_rows_x<4, 4, 0, 1, 1, false, 1>, (const void*)k_rows_x<4, 4, 1, 1, 1, false, 1>},
            {(const void*)k_rows_x<8, 2, 0, 1, 1, false, 1>, nullptr},
            {(const void*)k_rows_x<12, 1, 0, 1, 1, false, 1>, nullptr}};
        static const void* const tab3[3][2] = {
            {(const void*)k_rows_x<2, 4, 0, 1, 1, false, 3>, (const void*)k_rows_x<2, 4, 1, 1, 1, false, 3>},
            {(const void*)k_rows_x<4, 4, 0, 1, 1, false, 3>, (const void*)k_rows_x<4, 4, 1, 1, 1, false, 3>},
            {(const void*)k_rows_x<8, 2, 0, 1, 1, false, 3>, nullptr}};
        fn = n == 1 ? tab1[cls][proc] : cls < 3 ? tab3[cls][proc] : nullptr;
        if (gather) {        // gathered rows: the triple-row products of the GCNs (plain operands)
            if (proc != 0) return -1;
            if (jmax <= 8 && nsmax <= 2) fn = n == 1 ? (const void*)k_rows_x<8, 2, 0, 1, 1, false, 1, true> : (const void*)k_rows_x<8, 2, 0, 1, 1, false, 3, true>;
            else if (jmax <= 12 && nsmax <= 1)       // (the wide node vectors of the set-up GCNs: 1344 + 640 + 1344 columns)
                fn = n == 1 ? (const void*)k_rows_x<12, 1, 0, 1, 1, false, 1, true> : (const void*)k_rows_x<12, 1, 0, 1, 1, false, 3, true>;
            else return -1;
        }
        if (!fn) return -1;
    }
    if (dry) return -2;
    int launch_id = 0;
    (void)launch_id;
#ifdef ES_STAMP
    launch_id = g_rows_stamp ? g_rows_launch_id++ : 0;
#endif
    const XPre pf = x_prefetch_of_next();
    if (n == 1) {
        XLaunch<1> L;
        memset(&L, 0, sizeof(L));
        x_fill(L.p[0], pr[0], xp[0]);
        L.n = 1;
        L.pf = pf;
        grid = dim3((unsigned)(((pr[0].a.N + 15) / 16) / nt * xp[0].S), (unsigned)((pr[0].a.M + MT - 1) / MT), 1);
        L.p[0].ny = (int)grid.y; L.p[0].xw = (int)grid.x;
        if (grid.x >= 5461) return -1;
#ifdef ES_STAMP
        L.stamp = g_rows_stamp; L.launch_id = launch_id;
        if (g_rows_stamp && g_rows_log) fprintf(g_rows_log, "%d %d %d %d  x M%d K%d N%d S%d Jw%d ns%d proc%d%s\n", launch_id, (int)grid.x, (int)(grid.y * grid.z), n,
                                                pr[0].a.M, pr[0].a.K, pr[0].a.N, xp[0].S, xp[0].Jw, nsmax, proc, has_ln ? " ln" : "");
#endif
        return x_launch_np<1>(fn, L, grid, stream);
    }
    if (gx >= 5461) return -1;                       // (the slice lookup of a multi-problem launch divides by multiplication)
    XLaunch<3> L;
    memset(&L, 0, sizeof(L));
    for (int i = 0; i < n; ++i) {
        x_fill(L.p[i], pr[i], xp[i]);
        L.p[i].wg0 = RL.wg0[i]; L.p[i].ny = RL.ny[i]; L.p[i].xw = RL.xw[i]; L.p[i].fw = RL.fw[i];
    }
    for (int i = n; i < 3; ++i) L.p[i].wg0 = 0x7fffffff;
    L.n = n;
    L.pf = pf;
    grid = dim3((unsigned)gx, (unsigned)gy, 1);
#ifdef ES_STAMP
    L.stamp = g_rows_stamp; L.launch_id = launch_id;
    if (g_rows_stamp && g_rows_log) fprintf(g_rows_log, "%d %d %d %d  x M%d K%d N%d S%d Jw%d ns%d proc%d multi\n", launch_id, (int)grid.x, (int)grid.y, n,
                                            pr[0].a.M, pr[0].a.K, pr[0].a.N, xp[0].S, xp[0].Jw, nsmax, proc);
#endif
    return x_launch_np<3>(fn, L, grid, stream);
}

int rows_launch(const RowsPrep* pr, int n, es_stream stream) {
    RowsLaunch L;
    memset(&L, 0, sizeof(L));
    int nsmax = 1, proc = 0, kcmax = 16, gx = 0, gy = 1, wg_real = 0;
    bool csr = false, has_ln = false, gepi = false;
    for (int i = 0; i < n; ++i) {
        const int ny = (pr[i].a.M + MT - 1) / MT;
        gy = ny > gy ? ny : gy;
    }
    for (int i = 0; i < n; ++i) {
        L.p[i] = pr[i].a; L.S[i] = pr[i].S; L.kbps[i] = pr[i].kbps;
        L.wg0[i] = gx;
        L.ny[i] = (pr[i].a.M + MT - 1) / MT;
        L.xw[i] = ((pr[i].a.N + 15) / 16) * pr[i].S;
        L.fw[i] = (n > 1 && L.ny[i] < gy) ? (L.xw[i] * L.ny[i] + gy - 1) / gy : 0;      // fewer row tiles than the launch: folded
        gx += L.fw[i] > 0 ? L.fw[i] : L.xw[i];
        wg_real += L.xw[i] * L.ny[i];
        nsmax = pr[i].nsmax > nsmax ? pr[i].nsmax : nsmax;
        proc = pr[i].proc > proc ? pr[i].proc : proc;
        csr = csr || pr[i].csr; has_ln = has_ln || pr[i].has_ln; gepi = gepi || pr[i].gepi;
        const int kc = pr[i].kbps * 16 < KCH ? pr[i].kbps * 16 : KCH;
        kcmax = kc > kcmax ? kc : kcmax;
    }
    ES_REQUIRE(n == 1 || (!has_ln && !gepi && pr[0].nb == 1 && pr[1].nb == 1 && (n < 3 || pr[2].nb == 1)),
               "es_linear_rows_multi_f32: fused problems take no LayerNorm prologue, no GEGLU epilogue and no batching");
    L.n = n;
    L.ldx = kcmax + 8;
    // two column tiles per workgroup (NT = 2) for the GEGLU projection behind a LayerNorm: timing-only switch ES_ROWS_NT2 (0 = off)
    static const char* nt2_env = getenv("ES_ROWS_NT2");
    const bool nt2 = n == 1 && has_ln && gepi && pr[0].nb == 1 && pr[0].S == 1 && ((pr[0].a.N + 15) / 16) % 2 == 0 &&
                     !(nt2_env && atoi(nt2_env) == 0);
    if (g_rows_family == 1) {
        const int rc = x_launch(pr, n, stream, L, gx, gy);
        if (rc >= 0) return rc;
        if (n > 1) {
            // a group k_rows_x does not take as a whole: every problem takes the route it would take alone (a product's bits must not
            // depend on what it is fused with) -- unless none of them is a k_rows_x problem, then the group stays one k_linear_rows launch
            bool any = false;
            for (int i = 0; i < n; ++i) { XPlan xp; any = any || (x_plan(pr[i], &xp) && x_launch(pr + i, 0, stream, L, 0, 0) == -2); }
            if (any) {
                for (int i = 0; i < n; ++i) if (int rc1 = rows_launch(pr + i, 1, stream)) return rc1;
                return 0;
            }
        }
    }
    for (int i = 0; i < n; ++i)
        ES_REQUIRE(!pr[i].lnattn, "es_linear_rows_f32: ES_PRO_LN_ATTN runs on the register-operand kernel only (K=%d, %d slabs, kernel family %d): "
                                  "ask es_linear_rows_takes_ln_attn() at plan build", pr[i].a.K, pr[i].nsmax, g_rows_family);
    // k_linear_rows cuts K uniformly: segment-aligned cuts must coincide with that (every segment but the last a multiple of the slice)
    for (int i = 0; i < n; ++i) {
        const es_linear_args& a = pr[i].a;
        if (!a.seg_slices || pr[i].S == 1) continue;
        ES_REQUIRE((a.seg_slices >> 1) == 0, "es_linear_rows_f32: per-segment slice lengths need k_rows_x (the problem is not one it handles)");
        for (int sgi = 0; sgi + 1 < a.nseg; ++sgi)
            ES_REQUIRE((a.seg[sgi].width / 16) % pr[i].kbps == 0,
                       "es_linear_rows_f32: segment-aligned slices of %d k-blocks do not tile segment %d (%d columns) and the problem is not one k_rows_x handles", pr[i].kbps, sgi, a.seg[sgi].width);
    }
    if (nt2) gx /= 2;
    const size_t lds = (size_t)(MT * L.ldx + (nt2 ? 2 : 1) * NWAVE * 256) * sizeof(float);
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)pr[0].nb);
    // kernel table: lean instantiations for what the sampling path launches, one general kernel per slab bound for the rest
    static const void* const k_plain[4][2] = {
        {(const void*)k_linear_rows<1, 0, false, 0, false>, (const void*)k_linear_rows<1, 1, false, 0, false>},
        {(const void*)k_linear_rows<2, 0, false, 0, false>, (const void*)k_linear_rows<2, 1, false, 0, false>},
        {(const void*)k_linear_rows<4, 0, false, 0, false>, (const void*)k_linear_rows<4, 1, false, 0, false>},
        {(const void*)k_linear_rows<8, 0, false, 0, false>, (const void*)k_linear_rows<8, 1, false, 0, false>}};
    static const void* const k_general[4] = {
        (const void*)k_linear_rows<1, 2, true, 0, true>, (const void*)k_linear_rows<2, 2, true, 0, true>,
        (const void*)k_linear_rows<4, 2, true, 0, true>, (const void*)k_linear_rows<8, 2, true, 0, true>};
    static const void* const k_ln[2][2] = {
        {(const void*)k_linear_rows<1, 0, false, 4, true>, (const void*)k_linear_rows<2, 0, false, 4, true>},
        {(const void*)k_linear_rows<1, 0, false, 8, true>, (const void*)k_linear_rows<2, 0, false, 8, true>}};
    static const void* const k_ln_nt2[2][2] = {
        {(const void*)k_linear_rows<1, 0, false, 4, true, false, 2>, (const void*)k_linear_rows<2, 0, false, 4, true, false, 2>},
        {(const void*)k_linear_rows<1, 0, false, 8, true, false, 2>, (const void*)k_linear_rows<2, 0, false, 8, true, false, 2>}};
    static const void* const k_csr = (const void*)k_linear_rows<1, 0, true, 0, false>;
    // two-workgroups-per-CU variants (stage_chunk U1) of the plain family for launches of more than one round
    static const void* const k_plain_u1[3][2] = {
        {(const void*)k_linear_rows<1, 0, false, 0, false, true>, (const void*)k_linear_rows<1, 1, false, 0, false, true>},
        {(const void*)k_linear_rows<2, 0, false, 0, false, true>, (const void*)k_linear_rows<2, 1, false, 0, false, true>},
        {(const void*)k_linear_rows<4, 0, false, 0, false, true>, (const void*)k_linear_rows<4, 1, false, 0, false, true>}};
    const int nsi = nsmax <= 1 ? 0 : nsmax <= 2 ? 1 : nsmax <= 4 ? 2 : 3;
    static const char* u1_env = getenv("ES_ROWS_U1");            // timing-only A/B switch: 0 = never, 2 = every fused plain launch
    const int u1_mode = u1_env ? atoi(u1_env) : 1;
    // default rule: a FUSED launch of more than 256 workgroups whose ordinary variant holds one workgroup per CU (GroupNorm
    // prologue or 4-slab operands: 146-194 VGPRs): two per CU run 257-512 workgroups in ONE round instead of two
    const bool one_per_cu = proc == 1 || nsi == 2;
    const bool u1 = n > 1 && nsi <= 2 && (u1_mode == 2 || (u1_mode == 1 && one_per_cu && wg_real > 256));
    const void* fn = nullptr;
    if (has_ln) fn = (nt2 ? k_ln_nt2 : k_ln)[pr[0].a.K <= 512 ? 0 : 1][nsi];
    else if (csr && proc == 0 && nsmax == 1 && !gepi) fn = k_csr;
    else if (!csr && proc <= 1 && !gepi) fn = u1 ? k_plain_u1[nsi][proc] : k_plain[nsi][proc];
    else fn = k_general[nsi];
    {   // one-off per process, thread-safe: dynamic LDS limit (a 1024-column chunk needs 73 KiB)
        static std::once_flag once;
        static hipError_t attr_err = hipSuccess;
        std::call_once(once, [] {
            constexpr int bytes = (MT * (KCH + 8) + 2 * NWAVE * 256) * 4;
            auto set = [](const void* f) {
                const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
                if (e != hipSuccess && attr_err == hipSuccess) attr_err = e;
            };
            for (int i = 0; i < 4; ++i) { set(k_plain[i][0]); set(k_plain[i][1]); set(k_general[i]); }
            for (int i = 0; i < 3; ++i) { set(k_plain_u1[i][0]); set(k_plain_u1[i][1]); }
            for (int i = 0; i < 2; ++i) { set(k_ln[i][0]); set(k_ln[i][1]); set(k_ln_nt2[i][0]); set(k_ln_nt2[i][1]); }
            set(k_csr);
        });
        ES_REQUIRE(attr_err == hipSuccess, "es_linear_rows_f32: hipFuncSetAttribute failed: %s", hipGetErrorString(attr_err));
    }
    static const char* dbg_env = getenv("ES_ROWS_DBG");          // ablation switches of tools/microbench_rows.py (1: no weight loads, 2: no staging)
    L.dbg = dbg_env ? atoi(dbg_env) : 0;
#ifdef ES_STAMP
    L.stamp = g_rows_stamp;
    L.launch_id = g_rows_stamp ? g_rows_launch_id++ : 0;
    if (g_rows_stamp && g_rows_log) fprintf(g_rows_log, "%d %d %d %d  M%d K%d N%d S%d ns%d proc%d%s%s%s\n", L.launch_id, (int)grid.x, (int)grid.y, n,
                                            pr[0].a.M, pr[0].a.K, pr[0].a.N, pr[0].S, nsmax, proc, has_ln ? " ln" : "", csr ? " csr" : "", u1 ? " u1" : "");
#endif
    void* kargs[] = {(void*)&L};
    ES_CHECK_HIP(hipLaunchKernel(fn, grid, dim3(NTHREAD), kargs, lds, (hipStream_t)stream));
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}
}  // namespace

// 1 when es_linear_rows_f32(a) with an ES_PRO_LN_ATTN segment would run (the register-operand kernel takes it), 0 when not (the planner
// then keeps the self-attention product as its own launch), -1 on invalid arguments.  Host-only: launches nothing.
extern "C" int es_linear_rows_takes_ln_attn(const es_linear_args* a) {
    RowsPrep pr;
    if (rows_prepare(a, &pr)) return -1;
    if (!pr.lnattn || g_rows_family != 1) return 0;
    RowsLaunch L;
    memset(&L, 0, sizeof(L));
    return x_launch(&pr, 0, nullptr, L, 0, 0) == -2 ? 1 : 0;
}

extern "C" int es_rows_get_kernel_family(void) { return g_rows_family; }

extern "C" int es_rows_set_kernel_family(int family) {
    ES_REQUIRE(family == 0 || family == 1, "es_rows_set_kernel_family: %d", family);
    g_rows_family = family;
    return 0;
}

extern "C" int es_linear_rows_f32(const es_linear_args* a_in, es_stream stream) {
    RowsPrep pr;
    if (int rc = rows_prepare(a_in, &pr)) { g_rows_next = nullptr; return rc; }
    const int rc = rows_launch(&pr, 1, stream);
    g_rows_next = nullptr;          // the hint lives for ONE launch call, whichever kernel took it (it points into the caller's plan)
    return rc;
}

extern "C" int es_linear_rows_multi_f32(const es_linear_args* const* args, int n, es_stream stream) {
    ES_REQUIRE(args && n >= 1 && n <= 3, "es_linear_rows_multi_f32: n=%d (1..3)", n);
    RowsPrep pr[3];
    for (int i = 0; i < n; ++i) if (int rc = rows_prepare(args[i], &pr[i])) { g_rows_next = nullptr; return rc; }
    const int rc = rows_launch(pr, n, stream);
    g_rows_next = nullptr;
    return rc;
}

extern "C" int es_row_select(const es_rowsel_args* a, es_stream stream) {
    ES_REQUIRE(a->table && a->step && a->out && a->n > 0 && a->rows > 0, "es_row_select: bad args");
    hipLaunchKernelGGL(k_row_select, dim3((a->n + 255) / 256, a->rows < 64 ? a->rows : 64), dim3(256), 0, (hipStream_t)stream, *a);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_ddpm_update(const es_update_args* a, es_stream stream) {
    ES_REQUIRE(a->n > 0 && a->step && a->noise, "es_ddpm_update: bad args");
    if (a->n <= 4096) {
        hipLaunchKernelGGL(k_ddpm_update<true>, dim3(1), dim3(256), 0, (hipStream_t)stream, *a);
    } else {
        hipLaunchKernelGGL(k_ddpm_update<false>, dim3((a->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
        if (a->inc_step) hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, a->step);
    }
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_box_postprocess(float* boxes, int ld, const float* sincos, float* angle_out, const float* stats,
                                  int O, float angle_scale, es_stream stream) {
    ES_REQUIRE(O > 0 && (!boxes || (stats && ld >= 6)), "es_box_postprocess: bad args");
    hipLaunchKernelGGL(k_box_postprocess, dim3((O + 63) / 64), dim3(64), 0, (hipStream_t)stream, boxes, ld, sincos,
                       angle_out, stats, O, angle_scale, 6);
    ES_CHECK_HIP(hipGetLastError());
    return 0;
}

extern "C" int es_box_descale(float* boxes, int ld, int ncol, const float* stats, int O, es_stream stream) {
    ES_REQUIRE(O > 0 && boxes && stats && (ncol == 6 || ncol == 7) && ld >= ncol, "es_box_descale: bad args (ncol=%d, ld=%d)", ncol, ld);
    hipLaunchKernelGGL(k_box_postprocess, dim3((O + 63) / 64), dim3(64), 0, (hipStream_t)stream, boxes, ld, (const float*)nullptr,
                       (float*)nullptr, stats, O, 1.0f, ncol);
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
