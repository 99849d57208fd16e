/* A host WITHOUT Python: loads a model file written by the planner and runs one of the sampling loops through the C ABI only
 * (include/echoscene_hip.h).  Built by tests/test_hip_scene.py with hipcc against libechoscene_hip.so.
 * usage: replay_model <layout|shape|vq> <model file> <input .f32> <output .f32> <n_steps> */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime_api.h>
#include "echoscene_hip.h"

static float* read_f32(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    float* p = (float*)malloc(bytes);
    if (fread(p, 1, bytes, f) != (size_t)bytes) exit(2);
    fclose(f);
    *n = bytes / 4;
    return p;
}

#define CK(x) do { if ((x) != 0) { fprintf(stderr, "%s failed: %s\n", #x, es_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s <layout|shape|vq> model in.f32 out.f32 n_steps\n", argv[0]); return 2; }
    const char* kind = argv[1];
    const int n_steps = atoi(argv[5]);
    if (es_abi_version() != ES_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    es_model* m = es_model_load(argv[2]);
    if (!m) { fprintf(stderr, "es_model_load: %s\n", es_last_error()); return 1; }
    size_t n_in = 0;
    float* h_in = read_f32(argv[3], &n_in);
    float *d_in = NULL, *d_out = NULL;
    HK(hipMalloc((void**)&d_in, n_in * 4));
    HK(hipMemcpy(d_in, h_in, n_in * 4, hipMemcpyHostToDevice));
    void* reg = NULL;
    size_t out_bytes = 0;
    hipStream_t st;
    HK(hipStreamCreate(&st));
    if (!strcmp(kind, "layout")) {
        CK(es_model_region(m, "x", &reg, &out_bytes));
        HK(hipMalloc((void**)&d_out, out_bytes));
        const int rows = (int)(n_in * 4 / out_bytes);
        CK(es_layout_sample(m, d_in, rows, n_steps, d_out, (es_stream)st));
    } else if (!strcmp(kind, "shape")) {
        CK(es_model_region(m, "x", &reg, &out_bytes));
        if (out_bytes != n_in * 4) { fprintf(stderr, "latent size mismatch\n"); return 1; }
        HK(hipMalloc((void**)&d_out, out_bytes));
        CK(es_shape_sample(m, d_in, n_steps, d_out, (es_stream)st));
    } else {
        CK(es_model_region(m, "sdf", &reg, &out_bytes));
        HK(hipMalloc((void**)&d_out, out_bytes));
        CK(es_vq_decode(m, d_in, d_out, (es_stream)st));
    }
    HK(hipStreamSynchronize(st));
    float* h_out = (float*)malloc(out_bytes);
    HK(hipMemcpy(h_out, d_out, out_bytes, hipMemcpyDeviceToHost));
    FILE* f = fopen(argv[4], "wb");
    if (!f || fwrite(h_out, 1, out_bytes, f) != out_bytes) { fprintf(stderr, "cannot write %s\n", argv[4]); return 1; }
    fclose(f);
    printf("replay_model %s: %d ops, %zu output bytes\n", kind, es_model_num_ops(m), out_bytes);
    es_model_free(m);
    return 0;
}
