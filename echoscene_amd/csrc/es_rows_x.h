// Round 5: the rows product as ONE short instruction stream (included by es_rows.hip inside its anonymous namespace).
//
// What round 5 measured first (profiles/r05_rows_stamps_before.txt, tools/probes/probe_l2_persist.hip, probe_icache.hip):
//   * a launch of the layout step takes 5.4 us on average: 1.0 launch boundary, 0.6 kernarg read, 2.4-3.6 "staging" in k_linear_rows;
//   * memory is NOT what the staging waits for: a cold 1 KiB wave load returns in 0.43 us, an L2 hit in 0.1 us, and L2 contents survive
//     kernel boundaries;
//   * a cold instruction cache costs nothing measurable either (the fetch streams), but every wave64 VALU instruction costs ~5 clocks
//     of issue: 1000 instructions = 2 us.  k_linear_rows (and the first straight-line rewrite of this round, k_rows_frag: generic
//     segment selects, 1150 instructions) are INSTRUCTION-COUNT bound.
// So this kernel is built to execute few instructions:
//   * the host cuts K so that a slice never straddles two segments and hands every slice its own 64-byte descriptor (base pointer at
//     the slice's first column, leading dimension, slab count / stride, prologue, affine pointers): a workgroup reads ONE descriptor,
//     no per-block segment selects, no generic loops;
//   * no LDS staging of the A operand: a wave loads the MFMA A fragments of its k-blocks straight from global memory -- ONE per-lane
//     byte offset per wave, blocks and slabs are immediate / scalar offsets of the buffer loads -- and applies the prologue in
//     registers (a GroupNorm group of 16 channels = the 4 lanes of one row and k-block: two v_permlane*_swap steps);
//   * every load (weights first: they are the HBM misses) is issued before the first wait; the only workgroup barrier is the one in
//     front of the fixed-order reduction over the 8 waves (LayerNorm: one more for each of the two row statistics);
//   * two accumulators per wave (even / odd k-steps) halve the dependent MFMA chain.
// Arithmetic: exact fp32 products on v_mfma_f32_16x16x4_f32; K order = (wave, k-block, k-step parity), a function of (K, N, slices)
// only, never of M: per-row results do not depend on the batch.
//
// Not handled here (k_linear_rows keeps them): the CSR poolings, SiLU / GEGLU prologues (one-off table builds, tests), batched
// launches, step-indexed segments, slices that straddle segments.

struct XSlice {                                   // 64 bytes
    const float* a;                               // A at the slice's first column (LayerNorm: at column 0 of the row)
    const int32_t* idx;                           // row gather index or NULL
    const float* gamma; const float* beta;        // affine of the norm prologue at the slice's first channel (LayerNorm: channel 0)
    int32_t ld, nslab, sstr, flags;               // flags: 1 gather, 2 GroupNorm, 4 SiLU after the norm, 8 LayerNorm, 16 ReLU on the slab sum; bits 8..: k-blocks per wave
    int32_t gs; float eps; int32_t nkb, kb0;      // k-blocks of the slice, first k-block (index into the weight image)
};
constexpr int XMAXS = 6;
struct XProb {                                    // 128 + 6 * 64 = 512 bytes
    const float* wpack; const float* bias; const float* res; const float* res2; float* out; const int32_t* res_step;
    int32_t M, N, nkb_total, S;
    int32_t res_ld, res_nslab, res_sstr, res2_ld;
    int32_t res2_nslab, res2_sstr, out_ld, out_sstr;
    int32_t wg0, ny, xw, fw;
    int32_t act, res_step_stride, Jw; float inv_k;       // inv_k = 1.0f / K formed on the host (IEEE division: the bits the device division gave)
    XSlice sl[XMAXS];
};
// What the NEXT rows launch of the plan will read of its weight image (single-problem launches): the extra wave of this launch pulls
// those bytes into the L2 of the XCD that will read them (L2 contents survive kernel boundaries, workgroup id -> XCD is the same function
// in every launch: tools/probes/probe_l2_persist.hip, tools/rows_stamps.py).  w == NULL: nothing to fetch.
struct XPre {
    const char* w;
    int32_t gx, smagic, nkb_total, nt;            // tiles of one grid row (column tiles x slices), S | ceil(2^15 / S) << 16, k-blocks of a weight tile, NT
    int32_t cut[XMAXS + 1], pad;                  // first k-block of every slice, and the end
};
template <int NP>
struct XLaunch {
    XProb p[NP];
    XPre pf;
    int32_t n, pad;
#ifdef ES_STAMP
    unsigned long long* stamp;
    int32_t launch_id, pad2;
#endif
};

__device__ __forceinline__ float quad_sum(float v) {          // sum over the 4 lanes (i16, q = 0..3) of a row: same value in all four
    float e, o;
    es_pair16(v, e, o); v = e + o;
    es_pair32(v, e, o); return e + o;
}
__device__ __forceinline__ float pair_sum(float v) {          // lanes q and q ^ 1
    float e, o;
    es_pair16(v, e, o); return e + o;
}

// Buffer descriptor over "everything behind p" (2 GiB window); p == NULL -> zero records: loads return 0 without touching memory.
constexpr unsigned XOOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, p ? (int)XOOB : 0, 0x00020000);      // (p is wave-uniform at every call site)
}
// the same window, or no records at all when `on` is false (wave-uniform): the load is issued either way and returns zeros -- the load
// phase of the kernel has NO branches, so every load is in flight before the first wait and vmcnt is counted exactly.  (Round 5, from
// the ISA: with `if (present) load` the register allocator treated the fragment arrays as one tuple and copied it around -- behind
// `s_waitcnt vmcnt(0)` -- between the loads: 2.8 us to ISSUE 16 loads in the LayerNorm variant.)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t x_rsrc_if(const void* p, bool on) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, (p && on) ? (int)XOOB : 0, 0x00020000);
}
__device__ __forceinline__ f4 x_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ f4 x_ld4_nt(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2));  // aux 2 = nt: weights are read once
}
__device__ __forceinline__ float x_ld1(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}

// Row statistics in ONE workgroup round (PROC 3; the two-round form -- mean, barrier, centred squares, barrier -- costs 1.1 us per
// LayerNorm in the stamp timelines, r05b_rows_stamps_fold.txt): every wave reduces its own blocks about its own pivot p_w (its local
// mean), publishes (S_w, Q_w) = (sum, sum of squares about p_w), and every thread combines the eight pairs with the identity that
// holds for ANY pivot:  sum (x - mu)^2 = Q_w - 2 (mu - p_w) (S_w - n_w p_w) + n_w (mu - p_w)^2.  Fixed order over the waves.
// a: the wave's blocks (absent blocks are zeros), nb = its number of existing blocks (<= NB), Jw / nkb: blocks per wave / of the row.
template <int NB>
__device__ __forceinline__ void x_row_stats(const f4 (&a)[NB], int nb, int Jw, int nkb, float inv_k, float eps, float* buf, int wave, int i16,
                                            int q, float& mean, float& rstd) {
    float sm = 0.f;
#pragma unroll
    for (int v = 0; v < NB; ++v) sm += (a[v][0] + a[v][1]) + (a[v][2] + a[v][3]);
    sm = quad_sum(sm);
    const float pw = nb > 0 ? sm * __builtin_amdgcn_rcpf(16.0f * (float)nb) : 0.f;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < NB; ++v) {
        const float mv = v < nb ? pw : 0.f;                        // (an absent block holds zeros: against a zero pivot it adds exact zeros)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = a[v][e] - mv; sq += d * d; }
    }
    sq = quad_sum(sq);
    if (q == 0) { buf[wave * 16 + i16] = sm; buf[NKG * 16 + wave * 16 + i16] = sq; }
    __syncthreads();
    float S[NKG], Q[NKG], tot = 0.f;
#pragma unroll
    for (int w = 0; w < NKG; ++w) { S[w] = buf[w * 16 + i16]; Q[w] = buf[NKG * 16 + w * 16 + i16]; tot += S[w]; }
    mean = tot * inv_k;
    float m2 = 0.f;
#pragma unroll
    for (int w = 0; w < NKG; ++w) {
        int nbw = nkb - w * Jw;
        nbw = nbw < 0 ? 0 : nbw > Jw ? Jw : nbw;
        const float n = 16.0f * (float)nbw;
        const float p = nbw > 0 ? S[w] * __builtin_amdgcn_rcpf(n) : 0.f;      // (the pivot the wave used: same inputs, same bits)
        const float d = mean - p;
        m2 += (Q[w] - 2.0f * d * (S[w] - n * p)) + n * d * d;
    }
    rstd = __builtin_amdgcn_rsqf(fmaxf(m2, 0.f) * inv_k + eps);
}

// JW: bound of the k-blocks per wave and slice; NS: bound of the slab counts; PROC 0: raw / ReLU'd operands, 1: + GroupNorm(+SiLU),
// 2: LayerNorm over the row (SLN = slices the row is cut into: the statistics need all of them); 3: LayerNorm over a row that is
// FORMED here (ES_PRO_LN_ATTN, the one-token self-attention of a transformer block folded into its input projection, plan.py: the
// producer wrote [t0 | u] with u = W1 P t0 through folded weights, P = I - 11^T / C the mean subtraction of LayerNorm1 -- linear, so
// it sits in the weights; the row is x = rstd(t0) u + t0 + cav, i.e. attn1(LayerNorm1(t0)) + t0 + attn2 with attn1's bias inside cav
// -- a dependent launch less per block; cav arrives as the launch's res2, and the workgroups of column tile 0 publish x through the
// launch's res pointer); NT: column tiles per workgroup;
// NP: problems the launch may carry (1: single-problem launches read a 3x smaller argument block and skip the problem lookup).
// GATHER: rows may be gathered through an index (a dependent round trip in front of the A loads: its own variants).
template <int JW, int NS, int PROC, int SLN, int NT, bool GEGLU_EPI, int NP, bool GATHER = false>
__global__ __launch_bounds__(NTHREAD + 64) void k_rows_x(const XLaunch<NP> L) {
    __shared__ __attribute__((aligned(16))) float red[NT * NKG * 256];
    __shared__ float lnx[PROC == 2 ? 2 * NKG * 16 : PROC == 3 ? 4 * NKG * 16 : 1];     // (PROC 3: one buffer per statistics round)
    static_assert(PROC == 2 || SLN == 1, "k_rows_x: only LayerNorm reads foreign slices");
#ifdef ES_STAMP
    unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    st_[0] = __builtin_amdgcn_s_memrealtime();
#endif
    kernarg_warm<sizeof(XLaunch<NP>)>();
    ES_RSTAMP(1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pi = 0;
    if (NP > 1) pi = (L.n > 1 && (int)blockIdx.x >= L.p[1].wg0 ? 1 : 0) + (L.n > 2 && (int)blockIdx.x >= L.p[NP > 2 ? 2 : 0].wg0 ? 1 : 0);
    const XProb& PP = L.p[pi];
    struct XHdr { const float* wpack; const float* bias; const float* res; const float* res2; float* out; const int32_t* res_step;
                  int32_t M, N, nkb_total, S, res_ld, res_nslab, res_sstr, res2_ld, res2_nslab, res2_sstr, out_ld, out_sstr, wg0, ny, xw, fw,
                          act, res_step_stride, Jw; float inv_k; };
    static_assert(sizeof(XHdr) == 128 && sizeof(XProb) == 128 + XMAXS * 64, "k_rows_x: descriptor layout");
    const XHdr P = *(const XHdr*)&PP;                      // by VALUE: the whole header in two wide scalar loads, here
    // tiles of a problem = (column tile, slice) pairs along blockIdx.x, slice fastest, from wg0 on; a problem with fewer row tiles than
    // its (multi-problem) launch is folded over all grid rows (see RowsLaunch)
    int bx = (int)blockIdx.x - P.wg0, by = (int)blockIdx.y;
    const int S = P.S & 0xffff;                            // (high half: ceil(2^15 / S))
    if (NP > 1 && P.fw > 0) {
        const int v = by * P.fw + bx, xw = P.xw;
        if (v >= xw * P.ny) return;
        by = v / xw;
        bx = v - by * xw;
    } else if (by >= P.ny) return;
    // bx / S without the integer-division sequence: floor(bx * ceil(2^15 / S) / 2^15) is exact for S <= 6, bx < 5461 (host-checked)
    const int ct = (int)(((unsigned)bx * ((unsigned)P.S >> 16)) >> 15);
    const int slice = bx - ct * S;
    bx = ct;
    if (wave == NKG) {
        // the PREFETCH wave (launched only when the host knows the next launch: 9 waves): fetch the weight bytes that the workgroups
        // of the next launch with this workgroup's id (mod 8: the same XCD) will read.  Its loads have their own vmcnt; nobody waits
        // for them but the end of this wave.
        const int id = (int)(blockIdx.y * gridDim.x + blockIdx.x), w8 = (int)(gridDim.x * gridDim.y) & ~7;
        f4 sink = {0.f, 0.f, 0.f, 0.f};
        const int pS = L.pf.smagic & 0xffff, pnt = L.pf.nt, pnkb = L.pf.nkb_total, pgx = L.pf.gx;
        const unsigned pmag = (unsigned)L.pf.smagic >> 16;
        for (int b = id; b < pgx && w8 > 0; b += w8) {
            const int pct = (int)(((unsigned)b * pmag) >> 15), psl = __builtin_amdgcn_readfirstlane(b - pct * pS);
            const int k0 = L.pf.cut[psl], nk = L.pf.cut[psl + 1] - k0;          // (scalar loads from the argument block)
            for (int t = 0; t < pnt; ++t) {
                const char* base = L.pf.w + ((size_t)(pct * pnt + t) * pnkb + k0) * 1024 + lane * 16;
                // the destination is an in / out operand: it stays allocated (loads in flight must not land in a register the compiler
                // has handed to something else) and the loads are ordered; nothing ever reads it
                for (int kb = 0; kb < nk; ++kb) asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(base + (size_t)kb * 1024));
            }
        }
#pragma unroll
        for (int i = 0; i < (PROC >= 2 ? 3 : 1); ++i) __syncthreads();
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(sink));
        return;
    }
    const XSlice SL = PP.sl[slice];                        // by VALUE: one wide scalar load
    const int Jw = SL.flags >> 8, M = P.M, N = P.N, nkb_total = P.nkb_total;     // k-blocks per wave in THIS slice (slices may differ in length)
    const int nt = bx * NT;
    const int m0 = by * MT;
    const int i16 = lane & 15, q = lane >> 4;
    const int m = m0 + i16;
    const int mc = m < M ? m : M - 1;
    const int nj = SL.nkb - wave * Jw;                    // k-blocks of this wave inside the slice (may be <= 0 or > Jw)

    // (0) gather: the row index first (a dependent round trip in front of the A loads)
    int row = mc;
    if (GATHER && (SL.flags & 1)) {
        row = SL.idx[mc];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(row) :: "memory");       // (before the weights are issued: vmcnt retires in order)
    }

    // (1) weights (the HBM misses): the wave's k-blocks x NT tiles, 1 KiB each, contiguous in the packed image
    const __amdgpu_buffer_rsrc_t rW = x_rsrc((const f4*)P.wpack + ((size_t)nt * nkb_total + SL.kb0 + wave * Jw) * 64);
    f4 bf[NT][JW];
    unsigned vj[JW];                                      // per-block byte offset of the lane inside the wave's range, or out of range
#pragma unroll
    for (int j = 0; j < JW; ++j) {
        const bool on = j < Jw && j < nj;
        vj[j] = on ? (unsigned)j * 64u : XOOB;
#pragma unroll
        for (int t = 0; t < NT; ++t) bf[t][j] = x_ld4_nt(rW, on ? (unsigned)lane * 16u + (unsigned)j * 1024u : XOOB, (unsigned)(t * nkb_total) * 1024u);
    }

    // (2) A fragments: lane (i16, q) = columns [16 kb + 4 q, + 4) of its row, every slab of the producer (absent slabs: descriptors
    // without records).  LayerNorm: the blocks at the wave's position in EVERY slice (block v = (slice v / JW, position v % JW)); own
    // blocks are those of slice `slice`.
    constexpr int NB = JW * SLN;
    f4 av[NB][NS], gav[PROC >= 1 ? JW : 1], bev[PROC >= 1 ? JW : 1];
    const unsigned colw = (unsigned)(wave * Jw * 16 + 4 * q);                 // first column of the lane inside the slice
    const unsigned voff = ((unsigned)row * (unsigned)SL.ld + colw) * 4u;
    const unsigned sstr4 = (unsigned)SL.sstr * 4u;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
        const __amdgpu_buffer_rsrc_t rA = x_rsrc_if(SL.a, u < SL.nslab);
#pragma unroll
        for (int v = 0; v < NB; ++v) {
            const int j = v % JW, sl = v / JW;
            const unsigned so = PROC == 2 ? (unsigned)(sl * SL.nkb) * 64u : 0u;     // LayerNorm: SL.a is column 0, slices are nkb blocks apart
            av[v][u] = x_ld4(rA, (sl < S ? voff : XOOB) + vj[j], so + (unsigned)u * sstr4);
        }
    }
    // PROC 3: the other operands of the formed row -- u at the same rows, SL.gs columns behind t0 in the producer's output (same
    // slabs); the cross-attention vector (the launch's res2: plain rows, one slab)
    f4 uv[PROC == 3 ? JW : 1][NS], cv[PROC == 3 ? JW : 1];
    if (PROC == 3) {
        const unsigned uo = (unsigned)SL.gs * 4u;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const __amdgpu_buffer_rsrc_t rA = x_rsrc_if(SL.a, u < SL.nslab);
#pragma unroll
            for (int j = 0; j < JW; ++j) uv[j][u] = x_ld4(rA, voff + uo + vj[j], (unsigned)u * sstr4);
        }
        const __amdgpu_buffer_rsrc_t rC = x_rsrc(P.res2);
        const unsigned co = ((unsigned)mc * (unsigned)P.res2_ld + colw) * 4u;
#pragma unroll
        for (int j = 0; j < JW; ++j) cv[j] = x_ld4(rC, co + vj[j], 0);
    }
    const bool aff = PROC >= 1 && PROC != 3 && SL.gamma != nullptr;     // NULL: the affine of the norm is folded into the weights (host)
    if (PROC >= 1 && PROC != 3) {
        const __amdgpu_buffer_rsrc_t rG = x_rsrc_if(SL.gamma, (SL.flags & (2 | 8)) != 0), rB = x_rsrc_if(SL.beta, (SL.flags & (2 | 8)) != 0);
        const unsigned go = colw * 4u, gso = PROC == 2 ? (unsigned)(slice * SL.nkb) * 64u : 0u;
#pragma unroll
        for (int j = 0; j < JW; ++j) { gav[j] = x_ld4(rG, go + vj[j], gso); bev[j] = x_ld4(rB, go + vj[j], gso); }
    }

    // (PROC 3, from the ISA: with 44 loads per lane the scheduler started summing the first slabs -- behind vmcnt waits -- before it had
    //  issued the rest; nothing may move across this point, so every load is in flight before the first wait, as in the other variants)
    if (PROC == 3) __builtin_amdgcn_sched_barrier(0);
    ES_RSTAMP(2);

    // (4) prologue in registers
    f4 a[NB];
    const float relu_lo = (SL.flags & 16) ? 0.f : -INFINITY;
#pragma unroll
    for (int v = 0; v < NB; ++v) {
        const int j = v % JW, sl = v / JW;
        (void)j; (void)sl;
        f4 y = av[v][0];                                       // (blocks / slabs that do not exist are zeros: x + 0 = x)
#pragma unroll
        for (int u = 1; u < NS; ++u) y += av[v][u];
        // ReLU on the slab sum without a branch per block: max(y, 0) or max(y, -inf)
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaxf(y[e], relu_lo);
        a[v] = y;
    }
    if (PROC == 1 && (SL.flags & 2)) {
        const int gs = SL.gs;
        const float inv_gs = __builtin_amdgcn_rcpf((float)gs);      // gs is a power of two: exact
        f4 o[JW];
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            if (!(j < Jw && j < nj)) continue;                     // wave-uniform
            const f4 y = a[j];
            float sm = (y[0] + y[1]) + (y[2] + y[3]);
            if (gs >= 8) sm = pair_sum(sm);
            if (gs >= 16) { float e, od; es_pair32(sm, e, od); sm = e + od; }
            if (gs >= 32) {                                      // two adjacent k-blocks of this wave (Jw is even then)
                const f4 z = a[(j ^ 1) < JW ? (j ^ 1) : j];
                float s2_ = (z[0] + z[1]) + (z[2] + z[3]);
                s2_ = quad_sum(s2_);
                sm = (j & 1) ? s2_ + sm : sm + s2_;
            }
            const float mean = sm * inv_gs;
            float sq = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = y[e] - mean; sq += d * d; }
            if (gs >= 8) sq = pair_sum(sq);
            if (gs >= 16) { float e, od; es_pair32(sq, e, od); sq = e + od; }
            if (gs >= 32) {
                const f4 z = a[(j ^ 1) < JW ? (j ^ 1) : j];
                float q2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = z[e] - mean; q2 += d * d; }
                q2 = quad_sum(q2);
                sq = (j & 1) ? q2 + sq : sq + q2;
            }
            const float rstd = __builtin_amdgcn_rsqf(sq * inv_gs + SL.eps);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float zz = (y[e] - mean) * rstd;
                if (aff) zz = zz * gav[j][e] + bev[j][e];
                if (SL.flags & 4) zz = es_silu(zz);
                o[j][e] = zz;
            }
        }
#pragma unroll
        for (int j = 0; j < JW; ++j) if (j < Jw && j < nj) a[j] = o[j];      // (a[] stayed raw for the partner block of a 32-channel group)
    }
    if (PROC == 3) {
        // statistics of the t0 row (LayerNorm 1 of the block: its affine and mean subtraction sit in u's weights): one round
        const int nbv = nj < 0 ? 0 : nj > Jw ? Jw : nj;
        float mean0, rstd0;
        x_row_stats<NB>(a, nbv, Jw, SL.nkb, P.inv_k, SL.eps, lnx, wave, i16, q, mean0, rstd0);
        (void)mean0;
#if defined(ES_STAMP) && ES_STAMP_P3 == 1
        ES_RSTAMP(3);                              // (instrumented builds: where the prologue of this variant spends its time)
#endif
        // x = rstd0 u + t0 + cav; column tile 0 publishes it (the feed-forward output product reads it as an operand)
        const __amdgpu_buffer_rsrc_t rX = x_rsrc_if(P.res, bx == 0);
        const unsigned xo = ((unsigned)mc * (unsigned)P.res_ld + colw) * 4u;
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            f4 uu = uv[j][0];
#pragma unroll
            for (int u = 1; u < NS; ++u) uu += uv[j][u];
            f4 y = a[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = (uu[e] * rstd0 + y[e]) + cv[j][e];
            a[j] = y;                                            // (blocks that do not exist: every operand is zero)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, y), rX, (int)((m < M && vj[j] != XOOB) ? xo + vj[j] : XOOB), 0, 0);
        }
#if defined(ES_STAMP) && ES_STAMP_P3 == 2
        ES_RSTAMP(3);
#endif
    }
    if (PROC >= 2) {
        // LayerNorm over the whole row: the 8 waves hold all k-blocks of the row between them
        float mean, rstd;
        if (PROC == 3) {
            const int nbv = nj < 0 ? 0 : nj > Jw ? Jw : nj;
            x_row_stats<NB>(a, nbv, Jw, SL.nkb, P.inv_k, SL.eps, lnx + 2 * NKG * 16, wave, i16, q, mean, rstd);
        } else {
        float sm = 0.f;
#pragma unroll
        for (int v = 0; v < NB; ++v) sm += (a[v][0] + a[v][1]) + (a[v][2] + a[v][3]);      // (blocks that do not exist are zeros)
        sm = quad_sum(sm);
        if (q == 0) lnx[wave * 16 + i16] = sm;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NKG; ++w) tot += lnx[w * 16 + i16];
        const float inv_k = P.inv_k;
        mean = tot * inv_k;
        float sq = 0.f;
#pragma unroll
        for (int v = 0; v < NB; ++v) {
            const int j = v % JW, sl = v / JW;
            // (a block that does not exist holds zeros: against a zero "mean" it adds exact zeros -- a scalar select, no branch)
            const float mv = (j < Jw && j < nj && sl < S) ? mean : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = a[v][e] - mv; sq += d * d; }
        }
        sq = quad_sum(sq);
        if (q == 0) lnx[NKG * 16 + wave * 16 + i16] = sq;
        __syncthreads();
        float tq = 0.f;
#pragma unroll
        for (int w = 0; w < NKG; ++w) tq += lnx[NKG * 16 + w * 16 + i16];
        rstd = __builtin_amdgcn_rsqf(tq * inv_k + SL.eps);
        }
        // the own slice's blocks, normalised, move to a[0 .. JW)
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            f4 y = a[j];
#pragma unroll
            for (int sl = 1; sl < SLN; ++sl) if (slice == sl) y = a[sl * JW + j];
            if (j < Jw && j < nj) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { y[e] = (y[e] - mean) * rstd; if (aff) y[e] = y[e] * gav[j][e] + bev[j][e]; }
            }
            a[j] = y;
        }
    }
    // (4b) epilogue operands, issued only now: only the slice-0 workgroups add bias and residuals, nothing needs them before the
    // reduction, and vmcnt retires in order -- issued with the A loads they would sit in front of the prologue's wait
    const int ml = (tid >> 4) & 15, nl = tid & 15;
    const int te = NT > 1 ? (tid >> 8) : 0;
    const int nt_e = nt + te;
    const int n_e = nt_e * 16 + nl, m_e = m0 + ml;
    const int act = P.act;
    const bool geglu = GEGLU_EPI && act == ES_ACT_GEGLU;
    const int nres = geglu ? nt_e * 8 + nl : n_e;
    const bool first = slice == 0;
    const bool ok_e = tid < 256 * NT && m_e < M && n_e < N;
    const bool ok_res = geglu ? (ok_e && nl < 8) : ok_e;
    float e_res = 0.f, e_res2 = 0.f, e_bias = 0.f;
    float rr1[XMAXS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rr2[XMAXS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    {
        // no branches here either (round 5, from the ISA: twelve `if (u < nslab) load` were ~130 instructions in front of the MFMAs):
        // slabs that do not exist and workgroups of the other slices load through descriptors without records
        const float* rp = P.res;
        // (the step counter through the scalar cache: a vector load here would wait for every load issued above)
        if (first && P.res_step) rp += (long)(*(const __attribute__((address_space(4))) int32_t*)(unsigned long)P.res_step) * P.res_step_stride;
        const unsigned ro1 = ok_res ? ((unsigned)m_e * (unsigned)P.res_ld + (unsigned)nres) * 4u : XOOB;
        const unsigned ro2 = ok_res ? ((unsigned)m_e * (unsigned)P.res2_ld + (unsigned)nres) * 4u : XOOB;
        const int n1 = first ? P.res_nslab : 0, n2 = first ? P.res2_nslab : 0;
#pragma unroll
        for (int u = 0; u < XMAXS; ++u) {
            rr1[u] = x_ld1(x_rsrc_if(rp, u < n1), ro1, (unsigned)(u * P.res_sstr) * 4u);
            rr2[u] = x_ld1(x_rsrc_if(P.res2, u < n2), ro2, (unsigned)(u * P.res2_sstr) * 4u);
        }
        e_bias = x_ld1(x_rsrc_if(P.bias, first), tid < 256 * NT && n_e < N ? (unsigned)n_e * 4u : XOOB, 0);
    }
#if !defined(ES_STAMP) || !defined(ES_STAMP_P3) || ES_STAMP_P3 == 0
    ES_RSTAMP(3);
#else
    if (PROC != 3) ES_RSTAMP(3);
#endif

    // (5) products: two accumulators (even / odd k-steps), combined once
    f4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            if (j < Jw && j < nj) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][0], bf[t][j][0], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][1], bf[t][j][1], c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][2], bf[t][j][2], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][3], bf[t][j][3], c1, 0, 0, 0);
            }
        }
        acc[t] = c0 + c1;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) *(f4*)&red[(t * NKG + wave) * 256 + lane * 4] = acc[t];
    ES_RSTAMP(4);
    __syncthreads();
    ES_RSTAMP(5);
    // D layout of mfma 16x16: lane = (row >> 2) * 16 + col holds D[row][col] in register row & 3
    const int off = ((ml >> 2) * 16 + nl) * 4 + (ml & 3);
    float sres = 0.f;
#pragma unroll
    for (int w = 0; w < NKG; ++w) sres += red[(te * NKG + w) * 256 + off];
    e_res = rr1[0]; e_res2 = rr2[0];
#pragma unroll
    for (int u = 1; u < XMAXS; ++u) { e_res += rr1[u]; e_res2 += rr2[u]; }         // (absent slabs are zeros)
    float* out = P.out + (long)slice * P.out_sstr;
    if (GEGLU_EPI && geglu) {
        // tile rows: [8 value | 8 gate]; lane nl < 8 holds the value of output column 8 * nt_e + nl, lane nl + 8 its gate
        float sb = sres + e_bias;
        const float gate = __shfl_xor(sb, 8, 16);
        if (ok_res) out[(long)m_e * P.out_ld + nres] = sb * es_gelu(gate) + e_res;
    } else if (ok_e) {
        if (first) {
            sres += e_bias;
            if (act == ES_ACT_RELU) sres = fmaxf(sres, 0.f);
            else if (act == ES_ACT_SILU) sres = es_silu(sres);
            else if (act == ES_ACT_SIGMOID) sres = 1.0f / (1.0f + expf(-sres));
            sres += e_res;
            sres += e_res2;
        }
        out[(long)m_e * P.out_ld + n_e] = sres;
    }
#ifdef ES_STAMP
    if (L.stamp && tid == 0) {
        const unsigned wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (wg < 1024) {
            st_[6] = __builtin_amdgcn_s_memrealtime();
            st_[7] = __builtin_amdgcn_s_getreg((3 << 11) | 20) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32);
            unsigned long long* d = L.stamp + ((size_t)L.launch_id * 1024 + wg) * 8;
            for (int k = 0; k < 8; ++k) d[k] = st_[k];
        }
    }
#endif
}
