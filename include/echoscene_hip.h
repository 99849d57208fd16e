/* echoscene_hip.h -- C ABI of libechoscene_hip.so (MI355X / gfx950 only).
 *
 * The reference (ymxlzgy/echoscene) has no FFI: its hot path is Python calling ATen ops
 * (SURVEY.md section 1).  This header is therefore the boundary the *build* defines
 * (last row): the host-side mirror of model/SGDiff.py
 * (echoscene_amd/model/ *.py) calls these entry points through ctypes; each entry point
 * cites the reference function(s) whose arithmetic it replaces.
 *
 * Conventions
 *  - every pointer is DEVICE memory unless its name starts with h_; row-major, contiguous
 *    unless an explicit leading dimension (ld, in elements) is given;
 *  - I/O dtype is IEEE fp32; index arrays are int32; "f16" buffers are IEEE binary16;
 *  - all work is enqueued on the caller's hipStream_t (pass torch's current stream);
 *    nothing synchronises the device unless stated;
 *  - return value 0 = ok; otherwise es_last_error() holds a thread-local message.
 *    The Python wrapper turns a non-zero status into RuntimeError.
 *  - inputs are borrowed for the duration of the enqueued work; outputs are caller-owned.
 */
#ifndef ECHOSCENE_HIP_H
#define ECHOSCENE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* es_stream;          /* hipStream_t */
typedef struct es_plan es_plan;   /* opaque: an ordered list of ops, optionally captured into a hipGraph */

#define ES_ABI_VERSION 10
int es_abi_version(void);
const char* es_last_error(void);
/* device name / CU count of the current device (diagnostics for bench.py) */
int es_device_info(char* name_out, int name_cap, int* cu_count);

/* ------------------------------------------------------------------------------------------
 * "rows" path -- fp32 node / triple matrices with M <= a few hundred rows.
 * Replaces: nn.Linear + BatchNorm1d(eval) + ReLU chains of build_mlp (model/layers.py:21-38),
 * GraphTripleConv's gather / scatter_add average pooling (model/graph.py:146-199), and -- for
 * the 1-D box denoiser whose signal length is 1 -- GroupNorm/SiLU/Conv1d(centre tap),
 * LayerNorm/attention-with-one-key/GEGLU (denoise_net.py:293-313, attention.py:172-245).
 *
 * One generic fused kernel:  out = act( prologue(A) @ W^T + bias ) + res
 *   A is the horizontal concatenation of up to 3 segments, each either direct rows, rows
 *   gathered through an index (obj_vecs[s_idx]), or a CSR segment-mean (the scatter_add/avg pool).
 *   Chip-wide parallelism for M = 32: grid = (16-column tiles x K slices, 16-row tiles); split-K partial sums are
 *   never reduced by a kernel of their own -- they are "slab tensors" summed by whoever reads them next.
 * ---------------------------------------------------------------------------------------- */
enum { ES_SEG_DIRECT = 0, ES_SEG_GATHER = 1, ES_SEG_CSRMEAN = 2, ES_SEG_CSRSUM = 3, ES_SEG_CSRWAVG = 4 };
/* CSRSUM: pooling='sum' (graph.py:186-199 without the division); CSRWAVG: pooling='wAvg' (graph.py:163-184): every entry is scaled by
 * its learned weight (es_seg.ent_wt) before the sum and the sum is divided by (sum of the weights + 1e-4) */
enum { ES_PRO_NONE = 0, ES_PRO_SILU = 1, ES_PRO_GN = 2, ES_PRO_GN_SILU = 3, ES_PRO_LN = 4, ES_PRO_GEGLU = 5, ES_PRO_LN_ATTN = 6 };
enum { ES_ACT_NONE = 0, ES_ACT_RELU = 1, ES_ACT_SILU = 2, ES_ACT_GEGLU = 3, ES_ACT_SIGMOID = 4 };   /* SIGMOID: WeightNetGCN's heads (graph.py:44-57) */
/* ES_ACT_GEGLU: W/bias rows are interleaved per 16-row tile as [8 value rows | 8 gate rows] (es_pack_linear_geglu_f32);
 * the kernel writes N/2 columns: value * gelu(gate)  (GEGLU.forward, attention.py:39-46). */

typedef struct es_seg {
    const float* ptr;      /* source matrix                                                     */
    const int32_t* idx;    /* GATHER: row index [M];  CSRMEAN: row pointer [M+1]                */
    const int32_t* ent_row;/* CSRMEAN: source row of each entry                                 */
    const int32_t* ent_off;/* CSRMEAN: column offset of each entry inside the source row        */
    const int32_t* step;   /* optional device scalar: ptr += (*step) * step_stride  (sampler loops) */
    int32_t step_stride;
    int32_t ld;            /* leading dimension of the source (0 = broadcast one row to all M)  */
    int32_t width;         /* columns this segment contributes to K (multiple of 4)             */
    int32_t mode;          /* ES_SEG_*                                                          */
    /* slab form (round 3): the source is the split-K output of another es_linear_rows_f32 launch -- its value is
     * sum_{j < nslab} ptr[j * slab_stride + ...] in fixed order j = 0 .. nslab-1, summed while this op stages its operand
     * (the launch-boundary reduce; 0 / 1 = an ordinary tensor).  slab_stride in floats, a multiple of 4.            */
    int32_t nslab, slab_stride;
    int32_t pre_act;       /* ES_ACT_NONE / ES_ACT_RELU applied to the slab sum (a producer that splits K cannot apply its
                              own ReLU; build_mlp's final_nonlinearity, model/layers.py:33-37)          */
    /* per-segment prologue (the op-level `prologue` below is shorthand for the same prologue on every segment):
     * ES_PRO_GN / GN_SILU: GroupNorm over THIS segment in groups of `gs` channels (4..32, power of two), affine gamma/beta
     * [width]; ES_PRO_LN: LayerNorm over the segment (must be the only one); ES_PRO_SILU; ES_PRO_GEGLU.             
     * ES_PRO_LN_ATTN (round 5; the only segment, direct, gamma = beta = NULL, launches the library routes to its register-operand
     * kernel -- ask es_linear_rows_takes_ln_attn()): LayerNorm over a row that the launch FORMS first from the producer's output
     * [t0 | u] (u = `gs` columns behind the segment's pointer, same slabs): x = rstd(t0) * u + t0 + res2, where u = W1 P t0 was
     * produced through folded weights (W1: the one-token self-attention's matrix with LayerNorm1's affine, P = I - 11^T / C its mean
     * subtraction: plan.py, attention.py:172-219 on one token) and res2 is the cross-attention vector plus the self-attention's
     * bias -- i.e. x = attn1(norm1(t0)) + t0 + attn2(norm2(.), ctx).  The launch's `res` is then an OUTPUT: x is written there
     * ([M, res_ld], by the workgroups of column tile 0) and neither res nor res2 is added in the epilogue.                          */
    int32_t pro;
    const float* gamma; const float* beta;
    float eps;
    int32_t gs;
    /* CSRWAVG: weights [rows of the source, 2]; an entry of source row t is scaled by ent_wt[2 t] when its column offset is 0 (the
     * subject slot of the triple) and by ent_wt[2 t + 1] otherwise (the object slot): s_weights / o_weights of graph.py:165-170 */
    const float* ent_wt;
} es_seg;

typedef struct es_linear_args {
    es_seg seg[3];
    int32_t nseg;
    int32_t M, K, N;          /* K = sum of widths (GEGLU: the source holds 2K columns: value | gate) */
    const float* wpack;       /* W[N,K] packed by es_pack_linear_f32 (MFMA 16x16x4 fragment order) */
    const float* bias;        /* [N] or NULL                                                   */
    int32_t prologue;         /* ES_PRO_*  (GN*: 32 groups over K; LN: over K)                 */
    const float* gamma;       /* [K] affine of the norm prologue                               */
    const float* beta;
    float eps;
    int32_t act;              /* ES_ACT_*                                                      */
    const float* res;         /* residual [M, N] added AFTER the activation, or NULL           */
    int32_t res_ld;
    int32_t res_nslab, res_slab_stride;   /* the residual may itself be a slab tensor (see es_seg)  */
    const float* res2;        /* optional second residual (cross-attention-with-one-key vector) */
    int32_t res2_ld;
    int32_t res2_nslab, res2_slab_stride;
    float* out;               /* [M, N]  (ES_ACT_GEGLU: [M, N/2])                              */
    int32_t out_ld;
    /* batched launch (grid.z = nbatch): batch z uses seg[0].ptr + z*a_bstride, the z-th packed weight image
     * (images of equal shape stored back to back), bias + z*N, out + z*out_bstride.  0/1 = single problem. */
    int32_t nbatch, a_bstride, out_bstride;
    /* K split over workgroups (round 3).  kb_per_slice = number of 16-column k-blocks per slice (0 = one slice); the
     * launch runs S = ceil(ceil(K/16) / kb_per_slice) slices per column tile and slice s writes its partial products to
     * out + s * out_slab_stride (slice 0 adds bias and residuals): the output is a slab tensor for its consumers.  Needs
     * act == ES_ACT_NONE and no batching.  Slices are cut at multiples of the largest GroupNorm group; the count depends on
     * (K, N) only -- never on M -- so a row's arithmetic does not depend on the batch it is in. */
    int32_t kb_per_slice;
    int32_t out_slab_stride;  /* floats between output slabs (>= M * out_ld)                    */
    /* plan hint: this op and the NEXT ES_OP_LINEAR of the plan are independent problems (e.g. net1's first Linear and the residual
     * projection of a GraphTripleConv layer, model/graph.py:146-211) -- the runtime launches them as ONE grid
     * (es_linear_rows_multi_f32; up to 3 problems), one dependent launch less per pair.  Ignored by es_linear_rows_f32 itself. */
    int32_t fuse_next;
    /* round 5: the residual may be a row of a per-schedule table selected by the device step counter -- res + (*res_step) *
     * res_step_stride floats (with res_ld = 0 the row is broadcast over the M rows): the 22 ResBlock time projections of the layout
     * denoiser are read in place by their consumers instead of being copied out by a row-select launch every step.  NULL = off. */
    const int32_t* res_step;
    int32_t res_step_stride;
    /* round 5: 1 = the K slices never straddle two segments -- every segment (a multiple of 16 columns wide) is cut into
     * ceil(width / (16 kb_per_slice)) slices of kb_per_slice k-blocks (the last slice of a segment may be shorter); the slab count is
     * es_linear_rows_slices().  What the low-latency kernel of round 5 (k_rows_x) needs: one workgroup reads ONE segment.
     * Bits 1..3: segment 0..2 is cut with HALF the slice length (a GroupNorm segment next to plain ones). */
    int32_t seg_slices;
} es_linear_args;

/* host-side helper: number of floats of the packed image of W[N,K], and the packing itself
 * (h_w, h_out are HOST pointers).  Layout: [ceil(N/16)][ceil(K/16)][64 lanes][4], lane = q*16+j
 * holds W[nt*16+j][kb*16+4q .. +3], zero padded. */
size_t es_pack_linear_f32_size(int N, int K);
int es_pack_linear_f32(const float* h_w, int N, int K, float* h_out);
/* the same image from a weight already on the device (what the Python host uses: the host loop is a strided gather) */
int es_pack_linear_f32_dev(const float* d_w, int N, int K, float* d_out, es_stream stream);
/* C[N, M] = A[N, K] B[K, M] in fp64 on the device, every element a left fold over k of fma(a, b, acc): the planners' weight folds
 * (plan.py mm64) when the model's parameters already live on the GPU -- a fixed summation order, so every process folds the same bits */
int es_matmul_f64(const double* d_a, const double* d_b, double* d_c, int N, int K, int M, es_stream stream);
/* GEGLU projection W[2*Nh,K] (value rows | gate rows): interleave per 16-row tile, then pack (ES_ACT_GEGLU) */
int es_pack_linear_geglu_f32(const float* h_w, const float* h_bias, int Nh, int K, float* h_out, float* h_bias_out);

int es_linear_rows_f32(const es_linear_args* args, es_stream stream);
/* n <= 3 independent problems as one launch (same kernel class: no LayerNorm prologue, no GEGLU epilogue, no batching) */
int es_linear_rows_multi_f32(const es_linear_args* const* args, int n, es_stream stream);
/* Kernel family of the rows products, process-wide: 1 (default) = k_rows_frag (round 5: A fragments loaded straight into registers,
 * one barrier per workgroup) wherever it applies, 0 = k_linear_rows (LDS-staged operand) for everything.  Both compute exact fp32
 * products with the same K slices; the order of additions inside a slice differs, so results agree to rounding, not bit for bit.
 * For A/B tools and tests -- set it before plans are captured. */
int es_rows_set_kernel_family(int family);
int es_rows_get_kernel_family(void);

/* Route options of the volume path (round 5).  Everything that decides where an fp32 sum is cut or which kernel family multiplies --
 * i.e. the bits of the results -- is a process-wide option with a constant default; the library reads NO environment variable for
 * them (until round 4 they were getenv switches: two ranks, or a saving and a replaying process, with different environments silently
 * disagreed).  Names: conv_tile (128: force 128-row tiles), conv_force256, conv_ws, conv_wssplit, conv_wss_target, conv_deep,
 * conv_tinysplit, gn_rg.  es_model_save records es_options_string() in the file; es_model_load refuses a file written under other values. */
int es_vol_set_option(const char* name, int value);
int es_vol_options(char* out, int cap);                 /* "name=value;..." of the volume-path options; returns the length needed */
int es_options_string(char* out, int cap);              /* + "rows_family=..." : every numerics-affecting option of the library */
int es_model_file_options(const char* path, char* out, int cap);   /* the options string recorded in a model file (no device needed) */
/* number of slices the launch will run for `args` (and the rounded kb_per_slice) -- the planner sizes the slab buffer with it */
int es_linear_rows_slices(const es_linear_args* args, int* kb_per_slice);
/* host-only: 1 when a launch with an ES_PRO_LN_ATTN segment (es_seg.pro) would run, 0 when the library has no kernel for its shape
 * (the planner then keeps the self-attention product as a launch of its own), -1 on invalid arguments */
int es_linear_rows_takes_ln_attn(const es_linear_args* args);
/* the library's default kb_per_slice for a [*, K] x [K, N] product whose slices must be multiples of kalign_cols columns
 * (~256 workgroups per 32 rows, >= 128 columns per slice, <= 8 slabs); 0 = do not split */
int es_linear_rows_auto_slices(int K, int N, int kalign_cols);

/* ------------------------------------------------------------------------------------------
 * Diffusion updates.
 * es_ddpm_update: GaussianDiffusion.p_mean_variance + p_sample_sg with eps-prediction,
 *   'fixedsmall' variance, clip_denoised=False  (diffusion_ddpm.py:220-264, 296-309):
 *     x0   = c[0]*x - c[1]*eps;  mean = c[2]*x0 + c[3]*x;  x <- mean + c[4]*noise
 *   with c = coef[5*step ..] = {sqrt_recip_ac, sqrt_recipm1_ac, post_coef1, post_coef2,
 *   (t!=0)*exp(0.5*post_logvar)} prepared on the host in fp32 exactly as the reference's tables.
 * es_ddim_update: DDIMSampler.p_sample_ddim  (samplers/ddim.py:236-262):
 *     pred_x0 = (x - c[0]*e)/c[1];  x <- c[2]*pred_x0 + c[3]*e  (+ c[4]*noise[step] when `noise` is given: eta != 0)
 *   with c = {sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev-sigma_t^2), sigma_t}; eta = 0 (the shipped call): sigma_t = 0,
 *   noise = NULL, coef_stride 4.
 * `step` is a device scalar; when inc_step != 0 the kernel increments it after use.
 * ---------------------------------------------------------------------------------------- */
typedef struct es_update_args {
    float* x;               /* [n] state, updated in place                                      */
    const float* eps;       /* [n] network output                                               */
    int32_t eps_nslab, eps_slab_stride;   /* eps as a slab tensor (the denoiser's last product may split K) */
    const float* noise;     /* noise + (*step)*noise_stride is this step's draw (DDPM: always; DDIM: eta != 0, else NULL) */
    int32_t noise_stride;
    const float* coef;      /* [n_steps][coef_stride]                                           */
    int32_t coef_stride;
    int32_t* step;
    int32_t n;
    int32_t inc_step;
    int32_t clip_x0;        /* DDPM: clip_denoised=True -- the predicted x0 is clamped to [-1, 1] before the posterior mean
                               (p_mean_variance, diffusion_ddpm.py:243-244; the shipped sampling call passes False, echo2layout.py:113) */
} es_update_args;
int es_ddpm_update(const es_update_args* args, es_stream stream);
/* Post-path box de-normalisation: descale_box_params (helpers/util.py:542-557;
 * boxes [O, ld>=6] = sizes|translations in [-1,1], updated IN PLACE like the reference does to its argument;
 * stats = 14 floats of the dataset's boxes_centered_stats file) and postprocess_sincos2arctan (:559-568;
 * angle_out[O] = atan2(sin, cos) * angle_scale).  Either half may be skipped with NULL pointers. */
int es_box_postprocess(float* boxes, int ld, const float* sincos, float* angle_out, const float* stats,
                       int O, float angle_scale, es_stream stream);
/* descale_box_params alone, with its ``angle`` flag (helpers/util.py:542-557): ncol = 6 -> sizes | translations (the call above),
 * ncol = 7 -> also column 6, a normalised angle, back to [stats[12], stats[13]] (angle=True, :553-555).  In place. */
int es_box_descale(float* boxes, int ld, int ncol, const float* stats, int O, es_stream stream);
int es_ddim_update(const es_update_args* args, es_stream stream);

/* ------------------------------------------------------------------------------------------
 * "volume" path -- the 3-D latent-SDF UNet (openai_model_3d.py:816-863).  Activations are
 * channels-last [O, D, H, W, C]; the residual stream is fp32, every contraction reads fp16
 * operands and accumulates in fp32 on MFMA (v_mfma_f32_16x16x32_f16).
 * ---------------------------------------------------------------------------------------- */
enum { ES_CONV_SAME = 0, ES_CONV_DOWN_HW = 1, ES_CONV_UP_HW = 2, ES_CONV_UP_DHW = 3, ES_CONV_DOWN_DHW = 4 };

typedef struct es_conv_args {
    const void* a;            /* f16 [O, D, Hi, Wi, Cin] (channels-last)                         */
    const void* w;            /* f16 weights packed by es_pack_conv_f16 (tiled LDS-image order); for N <= 4,
                                 Cin <= 64 3x3x3 convs: es_pack_conv_rows_f16 ([N][27][Cin]) -- LDS-tiled dot2 kernel */
    int32_t O, D, H, W;       /* OUTPUT spatial size                                             */
    int32_t Cin, N;           /* N = true number of output channels                              */
    int32_t taps;             /* 27 (3x3x3, pad 1) or 1 (1x1x1 / linear)                         */
    int32_t mode;             /* ES_CONV_*: SAME; DOWN_HW = stride (1,2,2) (Downsample, :188);
                                 UP_HW = nearest x2 on H,W folded into addressing (Upsample, :150-153);
                                 UP_DHW = nearest x2 on D,H,W (VQ-VAE Upsample, vqvae_modules.py:24-39; 'concat' UNet,
                                 dims=4, openai_model_3d.py:155-156); DOWN_DHW = stride 2 on D,H,W ('concat' UNet
                                 Downsample with dims=4, :188)                                              */
    /* optional second contraction accumulated into the same tile: the 1x1 skip_connection of a
       ResBlock whose channel count changes (out = conv2(h) + skip(x), :294-314) */
    const void* a2; const void* w2; int32_t Cin2;
    const float* bias;        /* [N] (sum of both biases when a2 is used) or NULL                */
    const float* rowvec;      /* [O, rowvec_ld] per-object vector broadcast over voxels (emb_layers output /
                                 cross-attention-with-one-key output), or NULL                   */
    int32_t rowvec_ld;
    const float* res;         /* fp32 residual [M, N] or NULL                                    */
    float* out_f32;           /* [M, N] or NULL                                                  */
    void* out_f16;            /* [M, N] or NULL                                                  */
    void* workspace;          /* optional split-K scratch: splitk * M * N floats (see splitk)       */
    int32_t splitk;           /* 0/1: none; S > 1: K is split over S workgroups per tile, partial sums go to
                                 `workspace` and a second kernel reduces them in a fixed order (deterministic)
                                 and applies the epilogue.  Used for the small-M 16x4x4 level (M = 8192 rows
                                 cannot fill 256 CUs otherwise).  -1: let the library choose; `workspace` must then hold
                                 8 * M * N floats, or 16 * M * N floats when M * N <= 2^22 (small outputs are split up to 16 ways) */
    int32_t out_ld;           /* leading dimension of both outputs and of res (>= N);
                                 out_ld < 0: out_f32 is written as NCDHW [O, N, D*H*W] (final eps conv)  */
    int32_t O_hint;           /* 0, or the object count of the WHOLE problem when this launch is one shard of it
                                 (multi-GPU object sharding): the split-K factor is then chosen as for the whole problem,
                                 so that the fp32 partial sums -- and the result -- are bit-identical to the unsharded run.
                                 < 0 (round 6, ABI 10): -O_hint is the object count of a REFERENCE shard; the launch cuts K
                                 where a launch of that many objects is cut best, whatever its own object count -- every rank
                                 of every world size (1 included) then leaves the same bits (ShapeDenoiser(deterministic=True)) */
    int32_t epilogue;         /* ES_EPI_NONE, or ES_EPI_GEGLU: GEGLU (attention.py:39-46) fused into the FeedForward
                                 proj: the weight rows are packed tile-interleaved -- packed row 16 k + c holds the value
                                 row of output 8 k + c for c < 8 and its gate row (4C + 8 k + c - 8) for c >= 8, i.e. every
                                 16-column MFMA tile carries 8 outputs' value and gate columns -- N = 8C (a multiple of
                                 224), the only output is out_f16 [M, 4C] = (value + b) * gelu(gate + b), out_ld = leading
                                 dim of that (a multiple of 8)                                              */
    float* gn_stats_out;      /* NULL, or [2][ceil(M/64)][N] fp32: per (64-row group, output column) the sum (plane 0) and
                                 the sum of squares (plane 1) of the FINAL fp32 output (after bias / rowvec / res), for the
                                 GroupNorm that reads this tensor next (es_gn_args.stats1 / stats2): the statistics pass
                                 of that GroupNorm -- a re-read of the whole tensor -- becomes a reduction of this 64x
                                 smaller array.  The 256-row producer/consumer kernel forms the sums in its epilogue from
                                 the values it stores (es_conv_emits_gn_stats() == 1); behind the other routes (64- /
                                 128-row tiles, k_linear_ws, split K, object chunks) a small kernel passes over the output
                                 instead.  One summation order for all routes (even rows of the group top to bottom, odd
                                 rows top to bottom, then the two halves): the same bits whichever route ran.
                                 Requires voxels per object % 64 == 0, channels-last output; out_f32 may be NULL (f16-only output)
                                 when es_conv_emits_gn_stats() == 1: the sums are formed from the fp32 values before rounding     */
    float* gn_part_out;       /* NULL, or the per-(object, voxel tile, group) partial sums [O][ntiles][gn_part_groups][2] that the
                                 statistics pass of the NEXT GroupNorm over this tensor alone (es_gn_args.part_in) would form: when
                                 the launch is split over K (es_conv_emits_gn_part() == 1) its reduction kernel -- which touches
                                 every output element anyway -- forms them from the values it stores, in that pass's own
                                 summation order (the same bits), and the GroupNorm runs its apply kernel only.  Ignored
                                 (nothing written) when es_conv_emits_gn_part() == 0: leave part_in NULL then             */
    int32_t gn_part_groups;   /* groups of that GroupNorm (N % groups == 0)                                                 */
} es_conv_args;
enum { ES_EPI_NONE = 0, ES_EPI_GEGLU = 1 };
int es_conv_mfma_f16(const es_conv_args* args, es_stream stream);
/* host-only: 1 when es_conv_mfma_f16(args) would form gn_stats_out inside its own epilogue, 0 when it would need the extra pass,
 * -1 on invalid arguments (es_last_error()).  Launches nothing. */
int es_conv_emits_gn_stats(const es_conv_args* args);
/* host-only: 1 when es_conv_mfma_f16(args) would write gn_part_out (a split-K launch with a channels-last fp32 output), else 0;
 * -1 on invalid arguments.  Launches nothing. */
int es_conv_emits_gn_part(const es_conv_args* args);
/* host-only: the number of fp32 slabs [S][M][N] es_conv_mfma_f16(args) writes into args->workspace -- 1 when it writes none (no
 * split, or K split inside the workgroup); -1 on invalid arguments.  With splitk = -1 the workspace must hold that many slabs (the
 * bounds stated at `splitk` are the maximum).  Launches nothing. */
int es_conv_split_of(const es_conv_args* args);
/* Split-operand image of an fp32 activation [M, C] (C % 4 == 0): out f16 [M, 3 C] = [hi | lo | hi], hi = f16(x), lo = f16(x - hi).
 * Against a weight image packed from [w_hi | w_hi | w_lo] (Cin = 3 C) es_conv_mfma_f16 accumulates hi w_hi + lo w_hi + hi w_lo in
 * fp32: the reference's fp32 arithmetic to ~2^-21 per product on the f16 matrix pipe (ShapeDenoiser(precision='fp32x')). */
int es_split_f16x3(const float* x, long M, int C, void* out, es_stream stream);
/* host helpers: pack a PyTorch conv/linear weight [N, Cin, kd,kh,kw] (or [N, Cin]) into the f16 image the kernel
 * streams: [n-tile of 224][K step = (Cin chunk of 32, tap)][256 rows x 64 B, swizzled] -- one contiguous 16 KiB
 * block per K step.  h_out holds uint16 bit patterns. */
size_t es_pack_conv_f16_size(int N, int Cin, int taps);
int es_pack_conv_f16(const float* h_w, int N, int Cin, int taps, uint16_t* h_out);
/* the same image (bit-identical: round to nearest even) from an fp32 weight [N][Cin][taps] already on the device */
int es_pack_conv_f16_dev(const float* d_w, int N, int Cin, int taps, uint16_t* d_out, es_stream stream);
int es_pack_conv_rows_f16(const float* h_w, int N, int Cin, int taps, uint16_t* h_out);

typedef struct es_gn_args {
    const float* x1; int32_t C1;     /* fp32 channels-last source 1 [O, V, C1]                  */
    const float* x2; int32_t C2;     /* optional source 2 (skip concat, th.cat([h, hs.pop()]))  */
    int32_t O, V;                    /* objects, voxels per object                              */
    int32_t groups; float eps;
    const float* gamma; const float* beta;   /* [C1+C2]                                         */
    int32_t silu;                    /* 0 none, 1 SiLU, 2 GELU (VQ-VAE norm_out, vqvae_modules.py:404-406) */
    float* stats;                    /* scratch, O*ceil(V/8)*groups*2 + O*groups*2 floats (per-tile partials, final statistics) */
    void* y_f16;                     /* normalised (+SiLU) output [O, V, C1+C2] f16              */
    void* raw_f16;                   /* optional un-normalised f16 copy of the concat (skip conv) */
    int32_t O_hint;                  /* 0, or the object count of the whole problem (sharding): partial-sum tiling as unsharded;
                                        < 0: -O_hint = the object count of the reference shard (es_conv_args.O_hint)          */
    const float* stats1;             /* NULL, or the [2][O*V/64][C1] row-group sums written by the conv that produced x1
                                        (es_conv_args.gn_stats_out); with x2, stats2 [2][O*V/64][C2] must be given too.  When
                                        present the statistics pass over x1 / x2 is skipped: (object, group) statistics are
                                        reduced from these sums (double, fixed order)                                */
    const float* stats2;
    int32_t x1_is_f16;               /* 1: x1 points at an f16 [O, V, C1] tensor -- the f16-ONLY output of the conv that produced it (a
                                        ResBlock's conv1 -> GroupNorm -> conv2 intermediate is never part of the residual stream: it is
                                        written once, as the operand precision the next contraction reads anyway, openai_model_3d.py:
                                        294-314).  Requires stats1 (the statistics are the conv's sums over its fp32 values), no x2   */
    int32_t y_is_f32;                /* 1: y_f16 / raw_f16 point at FP32 tensors (the fp32-operand validation route, es_conv_f32)       */
    const float* part_in;            /* NULL, or the partial sums the producing conv's split-K reduction left (es_conv_args.gn_part_out,
                                        same O / O_hint / V / groups, one source): the statistics pass is skipped             */
} es_gn_args;
/* GroupNorm32 (+SiLU) over channels-last volumes: ldm_diffusion_util.py:222-239, eps 1e-5 in
 * ResBlocks, 1e-6 in SpatialTransformer3D (attention.py:77-78). Two kernels: stats (a pass over x, or a reduction of
 * the producer's row-group sums), apply. */
int es_groupnorm_vol(const es_gn_args* args, es_stream stream);

typedef struct es_ln_args {
    const float* x; int32_t M, C; float eps; const float* gamma; const float* beta; void* y_f16;
    int32_t y_is_f32;     /* 1: y_f16 points at an fp32 tensor (fp32-operand validation route) */
} es_ln_args;
int es_layernorm_tokens(const es_ln_args* args, es_stream stream);      /* nn.LayerNorm, attention.py:229-231 */

typedef struct es_attn_args {
    const void* qkv;      /* f16 [B*Ntok, 3*C]: q | k | v, heads packed as (h d)                */
    int32_t B, Ntok, heads, dhead;
    float scale;          /* dhead^-0.5 (attention.py:158)                                      */
    void* out_f16;        /* [B*Ntok, C]                                                        */
} es_attn_args;
int es_attention_f16(const es_attn_args* args, es_stream stream);       /* CrossAttention.forward self-attn, attention.py:172-219 */

typedef struct es_geglu_args { const float* h_f32; int32_t M, C4; void* out_f16; int32_t out_is_f32; } es_geglu_args;  /* h: [M, 2*C4] value|gate; out_is_f32: fp32 result (validation route) */
int es_geglu_f16(const es_geglu_args* args, es_stream stream);          /* GEGLU: x * gelu(gate), attention.py:39-46 */

/* ------------------------------------------------------------------------------------------
 * fp32-OPERAND validation route of the volume path (csrc/es_vol32.hip).  The reference is fp32 everywhere (openai_model_3d.py:
 * 816-863, GroupNorm32 forced to fp32 at ldm_diffusion_util.py:237-239); the product multiplies fp16 operands.  These entry points run
 * the same argument structs with fp32 activations (a / a2 / res / out_f32 fp32 channels-last, Cin a multiple of 16) and fp32 weights
 * ([N][taps][Cin16], es_pack_conv_f32) on v_mfma_f32_16x16x4_f32 (exact fp32, 1/16 of the fp16 matrix rate), so that the effect of
 * operand rounding is a measurement.  No split K, no fused GEGLU, no statistics output; qkv / out of es_attention_f32 are fp32,
 * dhead <= 96.
 * ---------------------------------------------------------------------------------------- */
size_t es_pack_conv_f32_size(int N, int Cin, int taps);
int es_pack_conv_f32(const float* h_w, int N, int Cin, int taps, float* h_out);
int es_conv_f32(const es_conv_args* args, es_stream stream);
int es_attention_f32(const es_attn_args* args, es_stream stream);

/* VQVAE.decode_no_quant front end (vqvae_networks/network.py:95-103, quantizer.py:68-119): nearest
 * codebook entry per latent voxel; `lut` = codebook already mapped through post_quant_conv. */
typedef struct es_vq_args {
    const float* z;          /* [O,3,V] fp32 NCDHW latents                                       */
    const float* codebook;   /* [n_embed,3]                                                      */
    const float* lut;        /* [n_embed,3]                                                      */
    int32_t O, V, n_embed, Cpad;
    int32_t* idx_out;        /* optional [O*V] chosen indices                                    */
    void* out_f16;           /* [O*V, Cpad] channels-last f16                                    */
} es_vq_args;
int es_vq_lookup(const es_vq_args* args, es_stream stream);

/* NCDHW fp32 latent <-> channels-last helpers, the 3->32->64 conv-pool stem of
 * shape_messsage_passing (openai_model_3d.py:757-764). */
int es_latent_to_cl_f16(const float* x_ncdhw, int O, int C, int V, int Cpad, void* out_f16, es_stream s);
int es_latent_to_cl_f32(const float* x_ncdhw, int O, int C, int V, int Cpad, void* out_f32, es_stream s);
typedef struct es_stem_args {
    const float* x;        /* [O,Cin,16,16,16] fp32 NCDHW, object stride x_ostride floats            */
    const float* w0; const float* b0;   /* Conv3d(3,32,3)  weights [32,3,3,3,3]                   */
    const float* w1; const float* b1;   /* Conv3d(32,64,3) weights [64,32,3,3,3]                  */
    float* scratch;        /* [O,32,8,8,8] pooled stage-1 output                                 */
    float* out;            /* [O,512] = flatten(MaxPool3d(k2,s4)(conv2)) in NCDHW order          */
    int32_t O;
    int32_t Cin;           /* 3 ('crossattn': x_t) or 4 ('concat': x_t and the c_s channel); 0 = 3      */
    int32_t x_ostride;     /* floats between objects in x; 0 = Cin * 4096                              */
} es_stem_args;
int es_shape_stem(const es_stem_args* args, es_stream stream);


/* ------------------------------------------------------------------------------------------
 * Plans: an ordered op list enqueued by the native runtime (no Python between kernels), and
 * optionally captured once into a hipGraph and replayed per denoising step.
 * ---------------------------------------------------------------------------------------- */
enum {
    ES_OP_LINEAR = 1, ES_OP_DDPM = 2, ES_OP_DDIM = 3, ES_OP_COPY = 4, ES_OP_CONV = 5, ES_OP_GN = 6,
    ES_OP_LN = 7, ES_OP_ATTN = 8, ES_OP_GEGLU = 9, ES_OP_TO_CL = 10, ES_OP_STEM = 11, ES_OP_VQ = 12,
    ES_OP_FORK = 13, ES_OP_JOIN = 14, ES_OP_ROWSEL = 15,
    ES_OP_CONV_F32 = 16, ES_OP_ATTN_F32 = 17      /* the fp32-operand validation route: es_conv_f32 / es_attention_f32 on the same argument structs */
};
/* Row select: out[r, 0..n) = table[*step, 0..n) for r < rows.  The timestep-dependent but node-independent products of a
 * denoiser (time MLP, all ResBlock emb projections, box/shape time embedding) are tabulated once per schedule
 * [n_steps, n] and picked by the device-side step counter -- 4 launches and 92 MB of weights less per layout step. */
typedef struct es_rowsel_args {
    const float* table; int32_t stride;      /* floats between table rows */
    const int32_t* step;
    float* out; int32_t out_ld, rows, n;
} es_rowsel_args;
int es_row_select(const es_rowsel_args* args, es_stream stream);

typedef struct es_copy_args {      /* device-to-device copy: flat (rows <= 1) or 2-D (rows x bytes with pitches) */
    void* dst; const void* src; size_t bytes;
    int32_t rows; size_t dst_pitch, src_pitch;
} es_copy_args;
typedef struct es_tocl_args { const float* x; int32_t O, C, V, Cpad; void* out; int32_t out_is_f32; } es_tocl_args;
/* out_is_f32: 0 f16 / 1 fp32 channels-last copy of an NCDHW tensor; 2 (round 6): x is channels-last fp32 [O * V, C] already and out
 * is its split-operand image f16 [O * V, 3 C] = [hi | lo | hi] (es_split_f16x3, ShapeDenoiser(precision='fp32x')) */
typedef struct es_op {
    int32_t kind;
    int32_t lane;    /* execution lane (0 = main stream; >0 = side stream forked/joined with ES_OP_FORK/JOIN) */
    union {
        es_linear_args linear; es_update_args update; es_copy_args copy; es_conv_args conv;
        es_gn_args gn; es_ln_args ln; es_attn_args attn; es_geglu_args geglu; es_tocl_args tocl;
        es_stem_args stem; es_vq_args vq; es_rowsel_args rowsel;
    } u;
} es_op;

es_plan* es_plan_create(const es_op* ops, int n_ops);
void es_plan_destroy(es_plan* plan);
int es_plan_num_ops(const es_plan* plan);
/* enqueue every op once, in order, on `stream` (side lanes use internal streams + events) */
int es_plan_run(es_plan* plan, es_stream stream);
/* capture es_plan_run into a hipGraph (idempotent) */
int es_plan_capture(es_plan* plan, es_stream stream);
/* ------------------------------------------------------------------------------------------
 * Chamfer nearest-neighbour distance (offline consistency metric; the reference's only native kernels,
 * extension/old_chamfer/chamfer.cu:12-195, bound as chamfer.forward / chamfer.backward in chamfer_cuda.cpp:30-33).
 * xyz1 [batch,n,3], xyz2 [batch,m,3] fp32; dist = squared distance to the nearest point of the other cloud,
 * idx = its index (first minimum).  backward ACCUMULATES into gradxyz1/2 (caller zero-fills, as the reference).
 * ---------------------------------------------------------------------------------------- */
int es_chamfer_forward(const float* xyz1, const float* xyz2, int batch, int n, int m, float* dist1, int32_t* idx1,
                       float* dist2, int32_t* idx2, es_stream stream);
int es_chamfer_backward(const float* xyz1, const float* xyz2, int batch, int n, int m, const float* graddist1,
                        const float* graddist2, const int32_t* idx1, const int32_t* idx2, float* gradxyz1,
                        float* gradxyz2, es_stream stream);

/* ------------------------------------------------------------------------------------------
 * Marching cubes (SDF -> indexed triangle mesh): what eval_3dfront.py does right after the sampling path through
 * sdf_to_mesh (model/diff_utils/util_3d.py:194-236: mcubes.marching_cubes(sdf_i, level), PyMCubes, level 0.02).
 * sdf [O, n, n, n] fp32, array axes (i, j, k) = vertex coordinates (x, y, z) in grid units; inside = value < level.
 * tri_table: int8 [256][16] device table (echoscene_amd/mc_tables.py).  Two calls, because the mesh sizes are data dependent:
 *   count: counts[2*o] = vertices, counts[2*o+1] = triangles of object o (device int32[2*O]); `workspace`
 *          (es_marching_cubes_workspace bytes) carries the block scans to
 *   emit : verts [sum V, 3] fp32 and faces [sum T, 3] int32 (vertex ids local to the object), object o written at row
 *          vert_offset[o] / tri_offset[o] (device int64[O]).  Vertices are shared between cubes (one per crossing
 *          grid edge), output order = grid order (deterministic, no atomics).
 * ---------------------------------------------------------------------------------------- */
size_t es_marching_cubes_workspace(int O, int n);
int es_marching_cubes_count(const float* sdf, int O, int n, float level, const int8_t* tri_table, void* workspace,
                            int32_t* counts, es_stream stream);
int es_marching_cubes_emit(const float* sdf, int O, int n, float level, const int8_t* tri_table, void* workspace,
                           const int64_t* vert_offset, const int64_t* tri_offset, float* verts, int32_t* faces,
                           es_stream stream);

/* library-wide one-off device allocations (zero page); call once outside stream capture */
int es_init(void);
/* The sampling loops.  `step` is the device scalar the plan's ops read; it is set to first_step,
 * then the (captured) plan is launched n_steps times -- the plan's last update op increments it.
 *   layout: GaussianDiffusion.p_sample_loop_sg  (diffusion_ddpm.py:330-345), 1000 iterations
 *   shape : DDIMSampler.ddim_sampling           (samplers/ddim.py:127-181),   100 iterations   */
int es_sampler_run(es_plan* plan, int32_t* step, int first_step, int n_steps, int use_graph, es_stream stream);

/* ------------------------------------------------------------------------------------------
 * Model files: a plan built once (by the Python planner, echoscene_amd/plan*.py) serialised together with every device
 * buffer it references, so that a host WITHOUT Python can load it and run the sampling loops (the
 * coarse entry points es_layout_sample / es_shape_sample / es_vq_decode a non-Python binding of model/SGDiff.py would
 * call, SGDiff.py:87-95 -> echo2layout.py:96-124, echo2shape.py:484-525, vqvae_networks/network.py:95-103).
 *
 * File = header, buffer table, named regions, the op list with every pointer field rewritten as (buffer, offset), buffer
 * contents.  A buffer whose es_buffer_desc.bytes has bit 63 set is SCRATCH (every byte is written by the plan before it is read:
 * activations, split-K workspace, statistics): it is listed but not stored -- the loader allocates and zero-fills it -- so a model
 * file is about the size of the packed weights.  Header counts and sizes are validated on load (untrusted input); nothing throws
 * through the C boundary.  es_model_save is called by the planner (it knows the allocation ranges); es_model_load allocates the buffers with
 * hipMalloc, uploads the contents, rebases the pointers and creates the plan.  Named regions give the caller its I/O:
 *   layout model:  "x" [O,8] state, "noise" [T+1,O,8] (row 0 = x_T, row 1+i = draw of iteration i), "step" int32
 *   shape model :  "x" [O,3,16,16,16] latents, "step" int32
 *   both loops  :  "coef" = the schedule's coefficient table [n_steps][coef_stride]: es_model_run / es_layout_sample / es_shape_sample
 *                  refuse to run past its last row
 *   vq model    :  "z" [O,3,16,16,16] input latents, "sdf" [O,1,64,64,64] output
 * ---------------------------------------------------------------------------------------- */
typedef struct es_model es_model;
typedef struct es_buffer_desc { const void* ptr; size_t bytes; } es_buffer_desc;
typedef struct es_region_desc { char name[32]; const void* ptr; size_t bytes; } es_region_desc;
/* byte offsets (inside es_op) of the device-pointer fields of an op of `kind`; returns their number (<= cap) */
int es_op_pointer_offsets(int kind, size_t* offsets, int cap);
int es_model_save(const char* path, const es_plan* plan, const es_buffer_desc* buffers, int n_buffers,
                  const es_region_desc* regions, int n_regions);
es_model* es_model_load(const char* path);
void es_model_free(es_model* model);
int es_model_region(const es_model* model, const char* name, void** dev_ptr, size_t* bytes);
int es_model_num_ops(const es_model* model);
/* `n_steps` replays of the model's (captured) plan starting at loop iteration `first_step` (es_sampler_run on its own plan) */
int es_model_run(es_model* model, int first_step, int n_steps, es_stream stream);
/* GaussianDiffusion.p_sample_loop_sg (diffusion_ddpm.py:330-345): noise [n_rows >= n_steps + 1][O*8] device fp32 (row 0 = x_T);
 * x_out [O*8] device fp32 = x after n_steps ancestral steps */
int es_layout_sample(es_model* model, const float* noise, int noise_rows, int n_steps, float* x_out, es_stream stream);
/* DDIMSampler.ddim_sampling (samplers/ddim.py:127-181): z_T [O,3,16,16,16] device fp32 -> z_out after n_steps DDIM steps */
int es_shape_sample(es_model* model, const float* z_T, int n_steps, float* z_out, es_stream stream);
/* VQVAE.decode_no_quant (vqvae_networks/network.py:95-103): z [O,3,16,16,16] -> sdf_out [O,1,64,64,64], device fp32 */
int es_vq_decode(es_model* model, const float* z, float* sdf_out, es_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* ECHOSCENE_HIP_H */
