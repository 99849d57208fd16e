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
            return es_latent_to_cl_f16(op.u.tocl.x, op.u.tocl.O, op.u.tocl.C, op.u.tocl.V, op.u.tocl.Cpad, op.u.tocl.out, s);
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
    for (const es_op& op : p->ops) {
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
            fprintf(stderr, "\n");
            fflush(stderr);
        }
        if (int rc = dispatch(op, s)) return rc;
        if (dbg) ES_CHECK_HIP(hipStreamSynchronize(s));
    }
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
