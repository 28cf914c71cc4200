/* A complete tiny BlobCtrl edit driven from plain C through the plan-level C ABI (include/blobctrl_hip.h): no Python, no torch.
 *
 *   plan_edit <tiny_edit.bcplan> <tiny_edit_io.bin>
 *
 * Loads the relocatable plan (weights and scheduler tables embedded), copies the edit's inputs into the plan's named buffers, runs
 * prologue + the denoise steps three ways - eager bc_step, per-segment hipGraphs (bc_plan_capture), and the WHOLE loop as one graph
 * (bc_plan_capture_loop) - and compares the final latents of each with the reference loop's (tests/golden/loop_tiny.npz, produced by
 * the reference's own BlobNet / UNet / DDIMScheduler classes).  Exit code 0 = all three match.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "blobctrl_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s failed (rc=%d): %s\n", #x, rc_, bc_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef struct { char name[33]; uint64_t bytes; void* data; } Blob;

static Blob* find(Blob* b, int n, const char* name) {
    for (int i = 0; i < n; ++i) if (!strcmp(b[i].name, name)) return &b[i];
    return NULL;
}

static int upload(BcPlan* plan, Blob* blobs, int n, const char* name) {
    Blob* b = find(blobs, n, name);
    void* dst = NULL; long long bytes = 0;
    if (!b) { fprintf(stderr, "io file has no record '%s'\n", name); return 1; }
    CHECK(bc_plan_buffer(plan, name, &dst, &bytes));
    if ((uint64_t)bytes != b->bytes) { fprintf(stderr, "%s: plan buffer has %lld bytes, input %llu\n", name, bytes, (unsigned long long)b->bytes); return 1; }
    HIP(hipMemcpy(dst, b->data, b->bytes, hipMemcpyHostToDevice));
    return 0;
}

static int reset_edit(BcPlan* plan, Blob* blobs, int n) {
    static const char* inputs[] = {"latents", "ctx", "fg_lat", "bg_lat", "bg_score", "fg_score", "feat", "feat16"};
    for (unsigned i = 0; i < sizeof(inputs) / sizeof(inputs[0]); ++i) if (upload(plan, blobs, n, inputs[i])) return 1;
    static const char* zeroed[] = {"step_idx", "hist"};
    for (unsigned i = 0; i < 2; ++i) {
        void* p = NULL; long long bytes = 0;
        CHECK(bc_plan_buffer(plan, zeroed[i], &p, &bytes));
        HIP(hipMemset(p, 0, (size_t)bytes));
    }
    return 0;
}

static int compare(BcPlan* plan, Blob* expect, const char* what) {
    void* lat = NULL; long long bytes = 0;
    CHECK(bc_plan_buffer(plan, "latents", &lat, &bytes));
    HIP(hipDeviceSynchronize());
    float* got = (float*)malloc((size_t)bytes);
    HIP(hipMemcpy(got, lat, (size_t)bytes, hipMemcpyDeviceToHost));
    const float* ref = (const float*)expect->data;
    double maxerr = 0, maxref = 0;
    for (long long i = 0; i < bytes / 4; ++i) {
        if (!isfinite(got[i])) { fprintf(stderr, "%s: non-finite latent at %lld\n", what, i); return 1; }
        double e = fabs((double)got[i] - ref[i]);
        if (e > maxerr) maxerr = e;
        if (fabs(ref[i]) > maxref) maxref = fabs(ref[i]);
    }
    free(got);
    printf("%-28s max-abs err %.4e (|ref| max %.3f, rel %.3e)\n", what, maxerr, maxref, maxerr / maxref);
    return maxerr / maxref < 1e-2 ? 0 : 1;        /* the stated bar (1e-2 of scale), as in the Python free-running tiny-loop tests */
}

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: %s plan.bcplan io.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[2], "rb");
    if (!f) { perror(argv[2]); return 2; }
    uint32_t n = 0;
    if (fread(&n, 4, 1, f) != 1 || n > 64) { fprintf(stderr, "bad io file\n"); return 2; }
    Blob blobs[64];
    for (uint32_t i = 0; i < n; ++i) {
        memset(blobs[i].name, 0, 33);
        if (fread(blobs[i].name, 1, 32, f) != 32 || fread(&blobs[i].bytes, 8, 1, f) != 1) { fprintf(stderr, "bad io record\n"); return 2; }
        blobs[i].data = malloc(blobs[i].bytes ? blobs[i].bytes : 1);
        if (fread(blobs[i].data, 1, blobs[i].bytes, f) != blobs[i].bytes) { fprintf(stderr, "truncated io record\n"); return 2; }
    }
    fclose(f);
    Blob* expect = find(blobs, (int)n, "expected_latents");
    Blob* seqb = find(blobs, (int)n, "sequence");
    if (!expect || !seqb) { fprintf(stderr, "io file lacks expected_latents / sequence\n"); return 2; }
    const int steps = (int)(seqb->bytes / 4);
    const int32_t* active = (const int32_t*)seqb->data;

    BcPlan* plan = NULL;
    CHECK(bc_plan_load(argv[1], &plan));
    const int pro = bc_plan_find_segment(plan, "prologue"), sa = bc_plan_find_segment(plan, "step_active"),
              si = bc_plan_find_segment(plan, "step_inactive");
    if (pro < 0 || sa < 0 || si < 0) { fprintf(stderr, "plan lacks a segment\n"); return 1; }
    printf("plan: prologue %d launches, active step %d, inactive step %d; %d denoise steps\n", bc_plan_num_launches(plan, pro),
           bc_plan_num_launches(plan, sa), bc_plan_num_launches(plan, si), steps);
    int fail = 0;

    /* 1. eager replay of the launch lists (streams: the plan's own) */
    if (reset_edit(plan, blobs, (int)n)) return 1;
    CHECK(bc_step(plan, pro, NULL, 0));
    for (int i = 0; i < steps; ++i) CHECK(bc_step(plan, active[i] ? sa : si, NULL, 0));
    fail |= compare(plan, expect, "eager bc_step");

    /* 2. one hipGraph per segment */
    HIP(hipDeviceSynchronize());
    CHECK(bc_plan_capture(plan, sa, NULL, 0));
    CHECK(bc_plan_capture(plan, si, NULL, 0));
    if (reset_edit(plan, blobs, (int)n)) return 1;
    CHECK(bc_step(plan, pro, NULL, 0));
    for (int i = 0; i < steps; ++i) CHECK(bc_step(plan, active[i] ? sa : si, NULL, 0));
    fail |= compare(plan, expect, "per-step graphs");

    /* 3. the whole edit (prologue + every step) as ONE graph; replayed twice (the device step counter is reset by the host) */
    int* seq = (int*)malloc(sizeof(int) * (size_t)(steps + 1));
    seq[0] = pro;
    for (int i = 0; i < steps; ++i) seq[i + 1] = active[i] ? sa : si;
    void* loop = NULL;
    hipStream_t s;
    HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipStream_t side, side2;
    HIP(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    HIP(hipStreamCreateWithFlags(&side2, hipStreamNonBlocking));
    bc_stream streams[3] = {s, side, side2};
    CHECK(bc_plan_release(plan, sa));
    CHECK(bc_plan_release(plan, si));
    CHECK(bc_plan_capture_loop(plan, seq, steps + 1, streams, 3, &loop));
    for (int rep = 0; rep < 2; ++rep) {
        if (reset_edit(plan, blobs, (int)n)) return 1;
        CHECK(bc_graph_launch(loop, s));
        HIP(hipStreamSynchronize(s));
        fail |= compare(plan, expect, rep ? "whole-loop graph (replay 2)" : "whole-loop graph");
    }
    CHECK(bc_graph_destroy(loop));
    CHECK(bc_plan_destroy(plan));
    printf(fail ? "FAILED\n" : "OK\n");
    return fail;
}
