// Native runtime of the echoscene HIP library: error channel, plans (ordered op lists enqueued
// without returning to Python), hipGraph capture, and the two sampling loops.
//
// Why a plan executor: one layout denoising step is ~140 dependent kernels of a few microseconds;
// driving them from Python (ctypes, ~3-5 us per call) would be host-bound.  A plan is built once
// per (model, graph size) by echoscene_amd/plan.py, captured into a hipGraph on first use and
// replayed once per step; every step-dependent quantity (timestep-embedding row, noise row,
// schedule coefficients) is indexed on the device by a step counter, so the graph is immutable.
#include "es_common.h"
#include <vector>
#include <cstdlib>
#include <exception>

static thread_local char g_err[512] = "";

void es_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* es_last_error(void) { return g_err; }
extern "C" int es_abi_version(void) { return ES_ABI_VERSION; }

extern "C" int es_device_info(char* name_out, int name_cap, int* cu_count) {
    int dev = 0;
    ES_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    ES_CHECK_HIP(hipGetDeviceProperties(&p, dev));
    if (name_out && name_cap > 0) {
        snprintf(name_out, name_cap, "%s (%s)", p.name, p.gcnArchName);
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    return 0;
}

struct es_plan {
    std::vector<es_op> ops;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap = nullptr;
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
};

static int dispatch(const es_op& op, hipStream_t s) {
    switch (op.kind) {
        case ES_OP_LINEAR: return es_linear_rows_f32(&op.u.linear, s);
        case ES_OP_DDPM: return es_ddpm_update(&op.u.update, s);
        case ES_OP_DDIM: return es_ddim_update(&op.u.update, s);
        case ES_OP_COPY:
            if (op.u.copy.rows > 1)
                ES_CHECK_HIP(hipMemcpy2DAsync(op.u.copy.dst, op.u.copy.dst_pitch, op.u.copy.src, op.u.copy.src_pitch,
                                              op.u.copy.bytes, (size_t)op.u.copy.rows, hipMemcpyDeviceToDevice, s));
            else
                ES_CHECK_HIP(hipMemcpyAsync(op.u.copy.dst, op.u.copy.src, op.u.copy.bytes, hipMemcpyDeviceToDevice, s));
            return 0;
        case ES_OP_CONV: return es_conv_mfma_f16(&op.u.conv, s);
        case ES_OP_GN: return es_groupnorm_vol(&op.u.gn, s);
        case ES_OP_LN: return es_layernorm_tokens(&op.u.ln, s);
        case ES_OP_ATTN: return es_attention_f16(&op.u.attn, s);
        case ES_OP_GEGLU: return es_geglu_f16(&op.u.geglu, s);
        case ES_OP_TO_CL:
            if (op.u.tocl.out_is_f32 == 2)       // split-operand image of a channels-last fp32 tensor [O rows, C] (precision 'fp32x')
                return es_split_f16x3(op.u.tocl.x, (long)op.u.tocl.O * op.u.tocl.V, op.u.tocl.C, op.u.tocl.out, s);
            return op.u.tocl.out_is_f32 ? es_latent_to_cl_f32(op.u.tocl.x, op.u.tocl.O, op.u.tocl.C, op.u.tocl.V, op.u.tocl.Cpad, op.u.tocl.out, s)
                                        : es_latent_to_cl_f16(op.u.tocl.x, op.u.tocl.O, op.u.tocl.C, op.u.tocl.V, op.u.tocl.Cpad, op.u.tocl.out, s);
        case ES_OP_CONV_F32: return es_conv_f32(&op.u.conv, s);
        case ES_OP_ATTN_F32: return es_attention_f32(&op.u.attn, s);
        case ES_OP_STEM: return es_shape_stem(&op.u.stem, s);
        case ES_OP_VQ: return es_vq_lookup(&op.u.vq, s);
        case ES_OP_ROWSEL: return es_row_select(&op.u.rowsel, s);
        default: es_set_error("plan: unknown op kind %d", op.kind); return 3;
    }
}

int es_vol_init(void);
extern "C" int es_init(void) { return es_vol_init(); }

extern "C" es_plan* es_plan_create(const es_op* ops, int n_ops) {
    if (es_vol_init()) return nullptr;
    if (!ops || n_ops <= 0) { es_set_error("es_plan_create: empty op list"); return nullptr; }
    es_plan* p = new es_plan();
    p->ops.assign(ops, ops + n_ops);
    for (const es_op& op : p->ops)
        if (op.lane < 0 || op.lane > 4) { es_set_error("es_plan_create: lane %d out of range", op.lane); delete p; return nullptr; }
    return p;
}

extern "C" void es_plan_destroy(es_plan* p) {
    if (!p) return;
    if (p->exec) (void)hipGraphExecDestroy(p->exec);
    if (p->graph) (void)hipGraphDestroy(p->graph);
    if (p->cap) (void)hipStreamDestroy(p->cap);
    for (int i = 0; i < 4; ++i) {
        if (p->side[i]) (void)hipStreamDestroy(p->side[i]);
        if (p->ev_fork[i]) (void)hipEventDestroy(p->ev_fork[i]);
        if (p->ev_join[i]) (void)hipEventDestroy(p->ev_join[i]);
    }
    delete p;
}

extern "C" int es_plan_num_ops(const es_plan* p) { return p ? (int)p->ops.size() : 0; }

static int ensure_lane(es_plan* p, int lane) {
    const int i = lane - 1;
    if (!p->side[i]) {
        ES_CHECK_HIP(hipStreamCreateWithFlags(&p->side[i], hipStreamNonBlocking));
        ES_CHECK_HIP(hipEventCreateWithFlags(&p->ev_fork[i], hipEventDisableTiming));
        ES_CHECK_HIP(hipEventCreateWithFlags(&p->ev_join[i], hipEventDisableTiming));
    }
    return 0;
}

// Lanes: ops with lane==0 run on the caller's stream.  ES_OP_FORK(lane=L) makes side stream L wait
// for everything enqueued so far on the main stream; ops with lane==L then run concurrently with
// the main stream until ES_OP_JOIN(lane=L) makes the main stream wait for them.  Inside stream
// capture this becomes a fork/join in the graph.
extern "C" int es_plan_run(es_plan* p, es_stream stream) {
    ES_REQUIRE(p != nullptr, "es_plan_run: null plan");
    hipStream_t main = (hipStream_t)stream;
    int skip = 0;
    for (const es_op& op : p->ops) {
        if (skip > 0) { --skip; continue; }
        if (op.kind == ES_OP_FORK) {
            ES_REQUIRE(op.lane >= 1, "FORK needs lane >= 1");
            if (int rc = ensure_lane(p, op.lane)) return rc;
            ES_CHECK_HIP(hipEventRecord(p->ev_fork[op.lane - 1], main));
            ES_CHECK_HIP(hipStreamWaitEvent(p->side[op.lane - 1], p->ev_fork[op.lane - 1], 0));
            continue;
        }
        if (op.kind == ES_OP_JOIN) {
            ES_REQUIRE(op.lane >= 1 && p->side[op.lane - 1], "JOIN without FORK (lane %d)", op.lane);
            ES_CHECK_HIP(hipEventRecord(p->ev_join[op.lane - 1], p->side[op.lane - 1]));
            ES_CHECK_HIP(hipStreamWaitEvent(main, p->ev_join[op.lane - 1], 0));
            continue;
        }
        hipStream_t s = main;
        if (op.lane > 0) {
            ES_REQUIRE(p->side[op.lane - 1], "op on lane %d before FORK", op.lane);
            s = p->side[op.lane - 1];
        }
        static const char* dbg = getenv("ES_DEBUG_SYNC");
        if (dbg) {
            const long idx = &op - p->ops.data();
            fprintf(stderr, "[es] op %ld kind %d", idx, op.kind);
            if (op.kind == ES_OP_LINEAR)
                fprintf(stderr, " M=%d K=%d N=%d pro=%d nseg=%d modes=%d,%d,%d", op.u.linear.M, op.u.linear.K, op.u.linear.N,
                        op.u.linear.prologue, op.u.linear.nseg, op.u.linear.seg[0].mode, op.u.linear.seg[1].mode, op.u.linear.seg[2].mode);
            if (op.kind == ES_OP_CONV)
                fprintf(stderr, " taps=%d Cin=%d+%d N=%d O=%d(%d) %dx%dx%d mode=%d epi=%d splitk=%d ws=%d res=%d f32=%d f16=%d gnst=%d gnpart=%d", op.u.conv.taps, op.u.conv.Cin,
                        op.u.conv.a2 ? op.u.conv.Cin2 : 0, op.u.conv.N, op.u.conv.O, op.u.conv.O_hint, op.u.conv.D, op.u.conv.H, op.u.conv.W, op.u.conv.mode,
                        op.u.conv.epilogue, op.u.conv.splitk, op.u.conv.workspace != nullptr, op.u.conv.res != nullptr, op.u.conv.out_f32 != nullptr,
                        op.u.conv.out_f16 != nullptr, op.u.conv.gn_stats_out != nullptr, op.u.conv.gn_part_out != nullptr);
            if (op.kind == ES_OP_GN)
                fprintf(stderr, " C=%d+%d O=%d(%d) V=%d groups=%d f16src=%d part_in=%d stats1=%d", op.u.gn.C1, op.u.gn.C2, op.u.gn.O, op.u.gn.O_hint, op.u.gn.V, op.u.gn.groups,
                        op.u.gn.x1_is_f16, op.u.gn.part_in != nullptr, op.u.gn.stats1 != nullptr);
            fprintf(stderr, "\n");
            fflush(stderr);
        }
        if (op.kind == ES_OP_LINEAR) {
            // tell the rows launcher which (single) rows product the plan runs next -- wrapping around: a sampling loop replays the
            // plan -- so that this launch's extra wave can pull its weights into L2
            const es_op* base = p->ops.data();
            const size_t nops = p->ops.size();
            size_t k = (size_t)(&op - base);
            while (k < nops && base[k].kind == ES_OP_LINEAR && base[k].u.linear.fuse_next) ++k;      // end of this fused group
            const es_linear_args* nxt = nullptr;
            for (size_t step_ = 1; step_ <= nops; ++step_) {
                const es_op& o2 = base[(k + step_) % nops];
                if (o2.kind != ES_OP_LINEAR) continue;
                if (!o2.u.linear.fuse_next && &o2 != &op) nxt = &o2.u.linear;
                break;
            }
            es_rows_hint_next(nxt);
        }
        static const char* fuse_env = getenv("ES_ROWS_FUSE");       // A/B switch: 0 = one launch per product
        if (op.kind == ES_OP_LINEAR && op.u.linear.fuse_next && !(fuse_env && atoi(fuse_env) == 0)) {
            // independent row products marked by the planner: one launch for up to 3 of them
            const es_op* q = &op;
            const es_linear_args* group[3];
            int n = 0;
            while (n < 3 && q < p->ops.data() + p->ops.size() && q->kind == ES_OP_LINEAR && q->lane == op.lane) {
                group[n++] = &q->u.linear;
                if (!q->u.linear.fuse_next) break;
                ++q;
            }
            if (n > 1) {
                if (int rc = es_linear_rows_multi_f32(group, n, s)) return rc;
                if (dbg) ES_CHECK_HIP(hipStreamSynchronize(s));
                skip = n - 1;
                continue;
            }
        }
        if (int rc = dispatch(op, s)) { es_rows_hint_next(nullptr); return rc; }
        if (dbg) ES_CHECK_HIP(hipStreamSynchronize(s));
    }
    es_rows_hint_next(nullptr);      // (every rows launch call clears the hint itself; nothing of this plan may outlive the run)
    return 0;
}

extern "C" int es_plan_capture(es_plan* p, es_stream stream) {
    ES_REQUIRE(p != nullptr, "es_plan_capture: null plan");
    if (p->exec) return 0;
    (void)stream;
    // Capture on a private stream: the caller's stream is usually torch's legacy default stream,
    // which cannot be captured.  The graph is launched on the caller's stream afterwards.
    if (!p->cap) ES_CHECK_HIP(hipStreamCreateWithFlags(&p->cap, hipStreamNonBlocking));
    hipStream_t s = p->cap;
    // create side streams/events before capture starts (creation is not capturable)
    for (const es_op& op : p->ops)
        if (op.lane > 0) if (int rc = ensure_lane(p, op.lane)) return rc;
    ES_CHECK_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int rc = es_plan_run(p, (es_stream)s);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) { es_set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return 1; }
    p->graph = g;
    ES_CHECK_HIP(hipGraphInstantiate(&p->exec, p->graph, nullptr, nullptr, 0));
    return 0;
}

extern "C" int es_sampler_run(es_plan* p, int32_t* step, int first_step, int n_steps, int use_graph, es_stream stream) {
    ES_REQUIRE(p != nullptr && step != nullptr && n_steps >= 0, "es_sampler_run: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (use_graph) if (int rc = es_plan_capture(p, stream)) return rc;
    // step counter lives on the device; set by value (no host buffer whose lifetime an async copy would depend on)
    ES_CHECK_HIP(hipMemsetD32Async((hipDeviceptr_t)step, first_step, 1, s));
    for (int i = 0; i < n_steps; ++i) {
        if (use_graph) ES_CHECK_HIP(hipGraphLaunch(p->exec, s));
        else if (int rc = es_plan_run(p, stream)) return rc;
    }
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Model files (include/echoscene_hip.h): a plan + the device buffers it references, relocatable.
// ---------------------------------------------------------------------------------------------
#include <cstddef>
#include <string>
#include <algorithm>

#define ES_PTR(member) offsetof(es_op, u.member)

extern "C" int es_op_pointer_offsets(int kind, size_t* out, int cap) {
    std::vector<size_t> v;
    switch (kind) {
        case ES_OP_LINEAR:
            for (int s = 0; s < 3; ++s) {
                const size_t b = offsetof(es_op, u.linear.seg) + s * sizeof(es_seg);
                for (size_t f : {offsetof(es_seg, ptr), offsetof(es_seg, idx), offsetof(es_seg, ent_row), offsetof(es_seg, ent_off),
                                 offsetof(es_seg, step), offsetof(es_seg, gamma), offsetof(es_seg, beta), offsetof(es_seg, ent_wt)})
                    v.push_back(b + f);
            }
            for (size_t f : {ES_PTR(linear.wpack), ES_PTR(linear.bias), ES_PTR(linear.gamma), ES_PTR(linear.beta), ES_PTR(linear.res),
                             ES_PTR(linear.res2), ES_PTR(linear.out), ES_PTR(linear.res_step)})
                v.push_back(f);
            break;
        case ES_OP_DDPM: case ES_OP_DDIM:
            v = {ES_PTR(update.x), ES_PTR(update.eps), ES_PTR(update.noise), ES_PTR(update.coef), ES_PTR(update.step)};
            break;
        case ES_OP_COPY: v = {ES_PTR(copy.dst), ES_PTR(copy.src)}; break;
        case ES_OP_CONV: case ES_OP_CONV_F32:
            v = {ES_PTR(conv.a), ES_PTR(conv.w), ES_PTR(conv.a2), ES_PTR(conv.w2), ES_PTR(conv.bias), ES_PTR(conv.rowvec), ES_PTR(conv.res),
                 ES_PTR(conv.out_f32), ES_PTR(conv.out_f16), ES_PTR(conv.workspace), ES_PTR(conv.gn_stats_out), ES_PTR(conv.gn_part_out)};
            break;
        case ES_OP_GN:
            v = {ES_PTR(gn.x1), ES_PTR(gn.x2), ES_PTR(gn.gamma), ES_PTR(gn.beta), ES_PTR(gn.stats), ES_PTR(gn.y_f16), ES_PTR(gn.raw_f16), ES_PTR(gn.stats1), ES_PTR(gn.stats2), ES_PTR(gn.part_in)};
            break;
        case ES_OP_LN: v = {ES_PTR(ln.x), ES_PTR(ln.gamma), ES_PTR(ln.beta), ES_PTR(ln.y_f16)}; break;
        case ES_OP_ATTN: case ES_OP_ATTN_F32: v = {ES_PTR(attn.qkv), ES_PTR(attn.out_f16)}; break;
        case ES_OP_GEGLU: v = {ES_PTR(geglu.h_f32), ES_PTR(geglu.out_f16)}; break;
        case ES_OP_TO_CL: v = {ES_PTR(tocl.x), ES_PTR(tocl.out)}; break;
        case ES_OP_STEM:
            v = {ES_PTR(stem.x), ES_PTR(stem.w0), ES_PTR(stem.b0), ES_PTR(stem.w1), ES_PTR(stem.b1), ES_PTR(stem.scratch), ES_PTR(stem.out)};
            break;
        case ES_OP_VQ: v = {ES_PTR(vq.z), ES_PTR(vq.codebook), ES_PTR(vq.lut), ES_PTR(vq.idx_out), ES_PTR(vq.out_f16)}; break;
        case ES_OP_ROWSEL: v = {ES_PTR(rowsel.table), ES_PTR(rowsel.step), ES_PTR(rowsel.out)}; break;
        case ES_OP_FORK: case ES_OP_JOIN: break;
        default: return -1;
    }
    const int n = (int)v.size();
    for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
    return n;
}

namespace {
constexpr char ES_MODEL_MAGIC[8] = {'E', 'S', 'M', 'O', 'D', 'E', 'L', '2'};     // '2' (round 5): the route options follow the header
constexpr int ES_MODEL_OPT_BYTES = 512;
struct ModelHeader { char magic[8]; uint32_t abi, n_buffers, n_ops, n_regions, op_size, pad; };
struct RegionRec { char name[32]; uint32_t buffer, pad; uint64_t offset, bytes; };
constexpr uint64_t OFF_BITS = 40;

// encoded reference: 0 = NULL, else ((buffer + 1) << 40) | offset
bool encode_ptr(uint64_t p, const es_buffer_desc* bufs, int nb, uint64_t* out) {
    if (!p) { *out = 0; return true; }
    for (int i = 0; i < nb; ++i) {
        const uint64_t b = (uint64_t)bufs[i].ptr;
        if (p >= b && p < b + bufs[i].bytes) { *out = ((uint64_t)(i + 1) << OFF_BITS) | (p - b); return true; }
    }
    return false;
}
}  // namespace

// Buffer table entry = byte size; bit 63 set = SCRATCH: a buffer every byte of which the plan writes before it reads it
// (activations, split-K workspace, statistics) -- its contents are not stored, the loader allocates and zero-fills it.
constexpr uint64_t ES_BUF_SCRATCH = 1ull << 63;
constexpr uint32_t ES_MAX_BUFFERS = 1u << 20, ES_MAX_OPS = 1u << 20, ES_MAX_REGIONS = 4096;

struct es_model {
    std::vector<void*> bufs;
    std::vector<size_t> sizes;
    std::vector<RegionRec> regions;
    es_plan* plan = nullptr;
    long schedule_len = -1;      // loop iterations the coefficient table holds ("coef" region / the update op's stride); -1 = no loop
};

extern "C" int es_model_save(const char* path, const es_plan* plan, const es_buffer_desc* buffers, int n_buffers,
                             const es_region_desc* regions, int n_regions) {
    ES_REQUIRE(path && plan && buffers && n_buffers > 0, "es_model_save: bad args");
    // (es_buffer_desc.bytes with bit 63 set marks a scratch buffer: listed, not dumped)
    std::vector<es_buffer_desc> bd(buffers, buffers + n_buffers);
    std::vector<bool> scratch(n_buffers);
    for (int i = 0; i < n_buffers; ++i) {
        scratch[i] = (bd[i].bytes & ES_BUF_SCRATCH) != 0;
        bd[i].bytes &= ~ES_BUF_SCRATCH;
        ES_REQUIRE(bd[i].bytes < (1ull << OFF_BITS), "es_model_save: buffer %d too large", i);
    }
    buffers = bd.data();
    std::vector<es_op> ops = plan->ops;
    size_t offs[64];
    for (size_t k = 0; k < ops.size(); ++k) {
        const int n = es_op_pointer_offsets(ops[k].kind, offs, 64);
        ES_REQUIRE(n >= 0, "es_model_save: op %zu has unknown kind %d", k, ops[k].kind);
        for (int j = 0; j < n; ++j) {
            uint64_t* f = (uint64_t*)((char*)&ops[k] + offs[j]);
            uint64_t enc;
            ES_REQUIRE(encode_ptr(*f, buffers, n_buffers, &enc), "es_model_save: op %zu (kind %d) field at +%zu points outside every buffer (%p)",
                       k, ops[k].kind, offs[j], (void*)*f);
            *f = enc;
        }
    }
    FILE* fp = fopen(path, "wb");
    ES_REQUIRE(fp, "es_model_save: cannot open %s", path);
    ModelHeader h{};
    memcpy(h.magic, ES_MODEL_MAGIC, 8);
    h.abi = ES_ABI_VERSION; h.n_buffers = (uint32_t)n_buffers; h.n_ops = (uint32_t)ops.size(); h.n_regions = (uint32_t)n_regions;
    h.op_size = (uint32_t)sizeof(es_op);
    bool ok = fwrite(&h, sizeof(h), 1, fp) == 1;
    {   // the route options this plan was built (and validated) under: a replay under other values would cut fp32 sums elsewhere
        char opt[ES_MODEL_OPT_BYTES];
        memset(opt, 0, sizeof(opt));
        es_options_string(opt, ES_MODEL_OPT_BYTES);
        ok = ok && fwrite(opt, 1, ES_MODEL_OPT_BYTES, fp) == ES_MODEL_OPT_BYTES;
    }
    for (int i = 0; i < n_buffers && ok; ++i) { const uint64_t b = buffers[i].bytes | (scratch[i] ? ES_BUF_SCRATCH : 0); ok = fwrite(&b, 8, 1, fp) == 1; }
    for (int i = 0; i < n_regions && ok; ++i) {
        RegionRec r{};
        memcpy(r.name, regions[i].name, 32);
        r.name[31] = 0;
        uint64_t enc;
        if (!encode_ptr((uint64_t)regions[i].ptr, buffers, n_buffers, &enc) || !enc) { fclose(fp); ES_REQUIRE(false, "es_model_save: region %s outside every buffer", r.name); }
        r.buffer = (uint32_t)((enc >> OFF_BITS) - 1); r.offset = enc & ((1ull << OFF_BITS) - 1); r.bytes = regions[i].bytes;
        ok = fwrite(&r, sizeof(r), 1, fp) == 1;
    }
    ok = ok && fwrite(ops.data(), sizeof(es_op), ops.size(), fp) == ops.size();
    std::vector<char> host;
    for (int i = 0; i < n_buffers && ok; ++i) {
        if (scratch[i]) continue;
        host.resize(buffers[i].bytes);
        if (hipMemcpy(host.data(), buffers[i].ptr, buffers[i].bytes, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        ok = fwrite(host.data(), 1, host.size(), fp) == host.size();
    }
    fclose(fp);
    ES_REQUIRE(ok, "es_model_save: write to %s failed", path);
    return 0;
}

// every numerics-affecting route option of the library as one string: "rows_family=1;conv_tile=0;..."
extern "C" int es_options_string(char* out, int cap) {
    std::string s = "rows_family=" + std::to_string(es_rows_get_kernel_family()) + ";";
    char v[ES_MODEL_OPT_BYTES];
    es_vol_options(v, (int)sizeof(v));
    s += v;
    if (out && cap > 0) { strncpy(out, s.c_str(), (size_t)cap - 1); out[cap - 1] = 0; }
    return (int)s.size() + 1;
}

// the route options recorded in a model file (no device needed)
extern "C" int es_model_file_options(const char* path, char* out, int cap) {
    ES_REQUIRE(path && out && cap > 0, "es_model_file_options: bad args");
    FILE* fp = fopen(path, "rb");
    ES_REQUIRE(fp, "es_model_file_options: cannot open %s", path);
    ModelHeader h{};
    char opt[ES_MODEL_OPT_BYTES];
    const bool ok = fread(&h, sizeof(h), 1, fp) == 1 && !memcmp(h.magic, ES_MODEL_MAGIC, 8) && fread(opt, 1, ES_MODEL_OPT_BYTES, fp) == ES_MODEL_OPT_BYTES;
    fclose(fp);
    ES_REQUIRE(ok, "es_model_file_options: %s is not a model file of this format", path);
    opt[ES_MODEL_OPT_BYTES - 1] = 0;
    strncpy(out, opt, (size_t)cap - 1);
    out[cap - 1] = 0;
    return 0;
}

extern "C" void es_model_free(es_model* m) {
    if (!m) return;
    if (m->plan) es_plan_destroy(m->plan);
    for (void* p : m->bufs) if (p) (void)hipFree(p);
    delete m;
}

static es_model* model_load_impl(const char* path, FILE* fp, es_model* m) {
    auto fail = [&](const char* why) -> es_model* { es_set_error("es_model_load(%s): %s", path, why); return nullptr; };
    ModelHeader h{};
    if (fread(&h, sizeof(h), 1, fp) != 1 || memcmp(h.magic, ES_MODEL_MAGIC, 8)) return fail("not a model file");
    if (h.abi != ES_ABI_VERSION || h.op_size != sizeof(es_op)) return fail("written by another ABI version");
    // the header is untrusted input: bound every count before it sizes an allocation
    if (h.n_buffers == 0 || h.n_buffers > ES_MAX_BUFFERS || h.n_ops == 0 || h.n_ops > ES_MAX_OPS || h.n_regions > ES_MAX_REGIONS)
        return fail("implausible header counts");
    {
        char opt[ES_MODEL_OPT_BYTES], cur[ES_MODEL_OPT_BYTES];
        if (fread(opt, 1, ES_MODEL_OPT_BYTES, fp) != ES_MODEL_OPT_BYTES) return fail("truncated route options");
        opt[ES_MODEL_OPT_BYTES - 1] = 0;
        memset(cur, 0, sizeof(cur));
        es_options_string(cur, ES_MODEL_OPT_BYTES);
        if (strcmp(opt, cur)) {
            es_set_error("es_model_load(%s): written under route options '%s', this process runs '%s' (es_vol_set_option / es_rows_set_kernel_family): "
                         "the replay would cut its sums elsewhere", path, opt, cur);
            return nullptr;
        }
    }
    m->sizes.resize(h.n_buffers);
    std::vector<bool> scratch(h.n_buffers);
    size_t free_b = 0, total_b = 0, need = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return fail("hipMemGetInfo failed");
    for (uint32_t i = 0; i < h.n_buffers; ++i) {
        uint64_t b;
        if (fread(&b, 8, 1, fp) != 1) return fail("truncated buffer table");
        scratch[i] = (b & ES_BUF_SCRATCH) != 0;
        b &= ~ES_BUF_SCRATCH;
        if (b >= (1ull << OFF_BITS)) return fail("implausible buffer size");
        m->sizes[i] = (size_t)b;
        need += (size_t)b;
    }
    if (need > total_b) return fail("the buffers exceed the device memory");
    m->regions.resize(h.n_regions);
    if (h.n_regions && fread(m->regions.data(), sizeof(RegionRec), h.n_regions, fp) != h.n_regions) return fail("truncated region table");
    std::vector<es_op> ops(h.n_ops);
    if (fread(ops.data(), sizeof(es_op), h.n_ops, fp) != h.n_ops) return fail("truncated op list");
    m->bufs.assign(h.n_buffers, nullptr);
    std::vector<char> host;
    for (uint32_t i = 0; i < h.n_buffers; ++i) {
        if (hipMalloc(&m->bufs[i], m->sizes[i] ? m->sizes[i] : 1) != hipSuccess) return fail("hipMalloc failed");
        if (scratch[i]) {
            if (hipMemset(m->bufs[i], 0, m->sizes[i]) != hipSuccess) return fail("hipMemset failed");
            continue;
        }
        host.resize(m->sizes[i]);
        if (fread(host.data(), 1, host.size(), fp) != host.size()) return fail("truncated buffer contents");
        if (hipMemcpy(m->bufs[i], host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess) return fail("upload failed");
    }
    size_t offs[64];
    for (es_op& op : ops) {
        const int n = es_op_pointer_offsets(op.kind, offs, 64);
        if (n < 0) return fail("unknown op kind");
        for (int j = 0; j < n; ++j) {
            uint64_t* f = (uint64_t*)((char*)&op + offs[j]);
            if (!*f) continue;
            const uint64_t bi = (*f >> OFF_BITS) - 1, off = *f & ((1ull << OFF_BITS) - 1);
            if (bi >= h.n_buffers || off >= m->sizes[bi]) return fail("pointer reference out of range");
            *f = (uint64_t)m->bufs[bi] + off;
        }
    }
    for (RegionRec& r : m->regions) {
        r.name[31] = 0;
        if (r.buffer >= h.n_buffers || r.offset > m->sizes[r.buffer] || r.bytes > m->sizes[r.buffer] - r.offset) return fail("region out of range");
    }
    // length of the schedule the loop's coefficient table holds: es_model_run refuses to step past it
    for (const RegionRec& r : m->regions)
        if (!strncmp(r.name, "coef", 32))
            for (const es_op& op : ops)
                if ((op.kind == ES_OP_DDPM || op.kind == ES_OP_DDIM) && op.u.update.coef_stride > 0)
                    m->schedule_len = (long)(r.bytes / ((size_t)op.u.update.coef_stride * 4));
    m->plan = es_plan_create(ops.data(), (int)ops.size());
    return m->plan ? m : nullptr;
}

extern "C" es_model* es_model_load(const char* path) {
    FILE* fp = path ? fopen(path, "rb") : nullptr;
    if (!fp) { es_set_error("es_model_load: cannot open %s", path ? path : "(null)"); return nullptr; }
    es_model* m = nullptr;
    es_model* r = nullptr;
    try {                                         // nothing may unwind through the C boundary (std::bad_alloc from a resize, ...)
        m = new es_model();
        r = model_load_impl(path, fp, m);
    } catch (const std::exception& e) {
        es_set_error("es_model_load(%s): %s", path, e.what());
        r = nullptr;
    }
    fclose(fp);
    if (!r && m) es_model_free(m);
    return r;
}

extern "C" int es_model_region(const es_model* m, const char* name, void** dev_ptr, size_t* bytes) {
    ES_REQUIRE(m && name, "es_model_region: bad args");
    for (const RegionRec& r : m->regions)
        if (!strncmp(r.name, name, 32)) {
            if (dev_ptr) *dev_ptr = (char*)m->bufs[r.buffer] + r.offset;
            if (bytes) *bytes = (size_t)r.bytes;
            return 0;
        }
    ES_REQUIRE(false, "es_model_region: the model has no region '%s'", name);
}

extern "C" int es_model_num_ops(const es_model* m) { return m && m->plan ? (int)m->plan->ops.size() : 0; }

extern "C" int es_model_run(es_model* m, int first_step, int n_steps, es_stream stream) {
    ES_REQUIRE(m && m->plan, "es_model_run: null model");
    ES_REQUIRE(first_step >= 0 && n_steps >= 0, "es_model_run: first_step %d, n_steps %d", first_step, n_steps);
    ES_REQUIRE(m->schedule_len < 0 || (long)first_step + n_steps <= m->schedule_len,
               "es_model_run: iterations %d .. %d exceed the model's schedule (%ld steps)", first_step, first_step + n_steps, m->schedule_len);
    void* step = nullptr;
    if (int rc = es_model_region(m, "step", &step, nullptr)) return rc;
    return es_sampler_run(m->plan, (int32_t*)step, first_step, n_steps, 1, stream);
}

extern "C" int es_layout_sample(es_model* m, const float* noise, int noise_rows, int n_steps, float* x_out, es_stream stream) {
    ES_REQUIRE(m && noise && x_out && n_steps >= 0 && noise_rows >= n_steps + 1, "es_layout_sample: bad args (noise rows %d, steps %d)", noise_rows, n_steps);
    void *x = nullptr, *nz = nullptr;
    size_t xb = 0, nb = 0;
    if (int rc = es_model_region(m, "x", &x, &xb)) return rc;
    if (int rc = es_model_region(m, "noise", &nz, &nb)) return rc;
    ES_REQUIRE((size_t)noise_rows * xb <= nb, "es_layout_sample: %d noise rows exceed the model's schedule (%zu rows)", noise_rows, nb / xb);
    hipStream_t s = (hipStream_t)stream;
    ES_CHECK_HIP(hipMemcpyAsync(nz, noise, (size_t)noise_rows * xb, hipMemcpyDeviceToDevice, s));
    ES_CHECK_HIP(hipMemcpyAsync(x, noise, xb, hipMemcpyDeviceToDevice, s));               // x_T = row 0
    if (int rc = es_model_run(m, 0, n_steps, stream)) return rc;
    ES_CHECK_HIP(hipMemcpyAsync(x_out, x, xb, hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int es_shape_sample(es_model* m, const float* z_T, int n_steps, float* z_out, es_stream stream) {
    ES_REQUIRE(m && z_T && z_out && n_steps >= 0, "es_shape_sample: bad args");
    void* x = nullptr;
    size_t xb = 0;
    if (int rc = es_model_region(m, "x", &x, &xb)) return rc;
    hipStream_t s = (hipStream_t)stream;
    ES_CHECK_HIP(hipMemcpyAsync(x, z_T, xb, hipMemcpyDeviceToDevice, s));
    if (int rc = es_model_run(m, 0, n_steps, stream)) return rc;
    ES_CHECK_HIP(hipMemcpyAsync(z_out, x, xb, hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int es_vq_decode(es_model* m, const float* z, float* sdf_out, es_stream stream) {
    ES_REQUIRE(m && m->plan && z && sdf_out, "es_vq_decode: bad args");
    void *zi = nullptr, *so = nullptr;
    size_t zb = 0, sb = 0;
    if (int rc = es_model_region(m, "z", &zi, &zb)) return rc;
    if (int rc = es_model_region(m, "sdf", &so, &sb)) return rc;
    hipStream_t s = (hipStream_t)stream;
    ES_CHECK_HIP(hipMemcpyAsync(zi, z, zb, hipMemcpyDeviceToDevice, s));
    if (int rc = es_plan_run(m->plan, stream)) return rc;
    ES_CHECK_HIP(hipMemcpyAsync(sdf_out, so, sb, hipMemcpyDeviceToDevice, s));
    return 0;
}
