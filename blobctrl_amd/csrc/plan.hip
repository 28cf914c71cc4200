// Plan runtime: a denoise step (or any other recorded forward pass) as a STATIC list of launches behind the C ABI.
//
// The reference runs its loop body as ~2000 ATen calls issued from Python every step (blobctrl/pipelines/pipeline_blobnet.py:1025-1123).
// Here a plan is compiled once per (batch, canvas, steps) configuration - by blobctrl_amd/launch.py in-process, or offline into a
// `.bcplan` file - and executed by this runtime with no Python in the loop:
//   bc_plan_create / bc_plan_segment / bc_plan_add_*   build segments (prologue, BlobNet-active step, UNet-only step, ...)
//   bc_step                                            replay one segment (eagerly, or its captured hipGraph) on the caller's streams
//   bc_plan_capture                                    capture one segment into a hipGraph (side streams join through fork / signal /
//                                                      wait events and become parallel branches)
//   bc_plan_capture_loop                               capture a whole SEQUENCE of segments (the 50-step loop) into ONE hipGraph
//   bc_plan_save / bc_plan_load                        relocatable serialisation: every pointer is (buffer id, offset); the loader
//                                                      allocates one arena, uploads the initialised buffers (weights, tables) and
//                                                      patches the launch records - a plain C host can run an edit (tests/c/plan_edit.c)
//   bc_plan_destroy
// Conventions as in blobctrl_hip.h: no internal threads, all work on the caller's streams, 0 on success + bc_last_error().
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "bc_common.h"
#include "plan_format.h"

void bc_gemm_set_probe(hipEvent_t e);      // gemm.hip (timing probe between a split-K GEMM's main kernel and its reducer)
bool bc_gemm_probe_hit();

using namespace bcplan;                    // the record / buffer structures and the file parser: plan_format.h (HIP-free)

namespace {

struct EventSet {                // events of a timed replay: destroyed on every exit path
    std::vector<hipEvent_t> ev;
    explicit EventSet(size_t n) : ev(n, nullptr) {}
    ~EventSet() { for (auto e : ev) if (e) (void)hipEventDestroy(e); }
    hipEvent_t& operator[](size_t i) { return ev[i]; }
};

struct Seg {
    std::string name;
    std::vector<Rec> recs;
    hipGraphExec_t graph = nullptr;
};


}  // namespace

struct BcPlan {
    std::vector<Seg> segs;
    std::vector<hipEvent_t> events;
    float* slab[kMaxStreams] = {};
    // loader-owned state
    void* arena = nullptr;
    std::vector<Buf> bufs;
    hipStream_t own_streams[kMaxStreams] = {};
    int n_own_streams = 0;
};

namespace {

inline float as_float(uint64_t v) { uint32_t u = (uint32_t)v; float f; memcpy(&f, &u, 4); return f; }
#define P(k) reinterpret_cast<void*>(r.a[k])
#define CP(T, k) reinterpret_cast<const T*>(r.a[k])
#define MP(T, k) reinterpret_cast<T*>(r.a[k])
#define I(k) (int)(int64_t)r.a[k]
#define L(k) (long long)r.a[k]
#define F(k) as_float(r.a[k])

int plan_event(BcPlan* pl, int e, hipEvent_t* out) {
    BC_CHECK_ARG(e >= 0 && e < (int)pl->events.size(), "plan: bad event id %d", e);
    if (!pl->events[e]) BC_CHECK_HIP(hipEventCreateWithFlags(&pl->events[e], hipEventDisableTiming));
    *out = pl->events[e];
    return 0;
}

int launch_rec(BcPlan* pl, Rec& r, hipStream_t* streams, int nstreams) {
    bc_stream s = streams[r.sid < nstreams ? r.sid : 0];
    switch (r.op) {
        case BC_OP_GEMM:
            if (r.g.splitk > 1) r.g.slab = pl->slab[r.sid];
            return bc_gemm(&r.g, s);
        case BC_OP_GN_STATS: return bc_gn_stats(CP(bc_half, 0), I(1), I(2), I(3), MP(unsigned long long, 4), s);
        case BC_OP_GN_FINALIZE:
            return bc_gn_finalize(CP(unsigned long long, 0), I(1), CP(unsigned long long, 2), I(3), I(4), I(5), I(6), F(7), CP(float, 8),
                                  CP(float, 9), MP(float, 10), s);
        case BC_OP_GN_APPLY_FUSED:
            return bc_gn_apply_fused(CP(unsigned long long, 0), I(1), CP(unsigned long long, 2), I(3), CP(bc_half, 4), CP(bc_half, 5), I(6),
                                     I(7), I(8), F(9), CP(float, 10), CP(float, 11), I(12), MP(bc_half, 13), s);
        case BC_OP_MEMSET_ZERO: return bc_memset_zero(P(0), L(1), s);
        case BC_OP_DUP_HALVES: return bc_dup_halves(P(0), L(1), P(2), L(3), P(4), L(5), P(6), L(7), P(8), L(9), P(10), L(11), s);
        case BC_OP_ROWCHAIN_MIDX:
            return bc_rowchain_midx(I(0), I(1), I(2), CP(bc_half, 3), CP(bc_half, 4), CP(bc_half, 5), CP(float, 6), CP(bc_half, 7), I(8), F(9),
                                    MP(bc_half, 10), MP(bc_half, 11), F(12), s);
        case BC_OP_ROWCHAIN_PACK_KV:
            return bc_rowchain_pack_kv(CP(bc_half, 0), I(1), CP(bc_half, 2), I(3), I(4), I(5), I(6), MP(bc_half, 7), s);
        case BC_OP_ROWCHAIN_SUM:
            return bc_rowchain_sum(I(0), I(1), I(2), CP(bc_half, 3), I(4), MP(bc_half, 5), MP(unsigned long long, 6), MP(bc_half, 7), s);
        case BC_OP_CTX_FOLD:
            return bc_ctx_fold(CP(bc_half, 0), I(1), CP(bc_half, 2), I(3), I(4), I(5), I(6), I(7), F(8), CP(bc_half, 9), CP(float, 10), CP(bc_half, 11),
                               MP(bc_half, 12), MP(float, 13), MP(float, 14), MP(bc_half, 15), s);
        case BC_OP_GN_APPLY:
            return bc_gn_apply(CP(bc_half, 0), I(1), CP(bc_half, 2), I(3), I(4), I(5), CP(float, 6), I(7), MP(bc_half, 8), s);
        case BC_OP_LAYERNORM:
            return bc_layernorm(CP(bc_half, 0), I(1), I(2), I(3), CP(float, 4), CP(float, 5), F(6), MP(bc_half, 7), I(8), s);
        case BC_OP_ATTENTION:
            return bc_attention(CP(bc_half, 0), CP(bc_half, 1), CP(bc_half, 2), MP(bc_half, 3), I(4), I(5), I(6), I(7), I(8), I(9), I(10),
                                I(11), I(12), L(13), L(14), L(15), L(16), F(17), s);
        case BC_OP_ATTENTION_CAUSAL:
            return bc_attention_causal(CP(bc_half, 0), CP(bc_half, 1), CP(bc_half, 2), MP(bc_half, 3), I(4), I(5), I(6), I(7), I(8), I(9),
                                       I(10), I(11), I(12), L(13), L(14), L(15), L(16), F(17), s);
        case BC_OP_ASSEMBLE_INPUT:
            return bc_assemble_input(CP(float, 0), I(1), CP(float, 2), CP(float, 3), CP(float, 4), I(5), I(6), I(7), I(8), I(9), I(10),
                                     I(11), MP(bc_half, 12), s);
        case BC_OP_TIMESTEP_EMBEDDING: return bc_timestep_embedding(CP(float, 0), CP(int, 1), F(2), I(3), I(4), MP(bc_half, 5), s);
        case BC_OP_TIMESTEP_EMBEDDING_TABLE: return bc_timestep_embedding_table(CP(float, 0), I(1), I(2), I(3), MP(bc_half, 4), s);
        case BC_OP_CFG_SCHEDULER_STEP:
            return bc_cfg_scheduler_step(CP(float, 0), MP(float, 1), CP(float, 2), MP(int, 3), MP(float, 4), F(5), I(6), I(7), I(8),
                                         MP(float, 9), I(10), s);
        case BC_OP_EMBED_TOKENS:
            return bc_embed_tokens(CP(long long, 0), CP(bc_half, 1), CP(float, 2), I(3), I(4), I(5), I(6), MP(bc_half, 7), s);
        case BC_OP_SOFTMAX_ROWS: return bc_softmax_rows(MP(bc_half, 0), I(1), I(2), I(3), s);
        case BC_OP_PATCHIFY: return bc_patchify(CP(float, 0), I(1), I(2), I(3), I(4), I(5), MP(bc_half, 6), s);
        case BC_OP_ADD_CLS_POS: return bc_add_cls_pos(CP(bc_half, 0), CP(float, 1), CP(float, 2), I(3), I(4), I(5), MP(bc_half, 6), s);
        case BC_OP_SILU: return bc_silu(CP(bc_half, 0), MP(bc_half, 1), L(2), s);
        case BC_OP_NCHW_TO_NHWC_F16: return bc_nchw_to_nhwc_f16(P(0), I(1), I(2), I(3), I(4), I(5), MP(bc_half, 6), s);
        case BC_OP_NHWC_TO_NCHW: return bc_nhwc_to_nchw(CP(bc_half, 0), I(1), I(2), I(3), I(4), P(5), I(6), s);
        case BC_OP_GAUSSIAN_SAMPLE: return bc_gaussian_sample(CP(bc_half, 0), CP(float, 1), I(2), I(3), I(4), F(5), MP(float, 6), s);
        case BC_OP_ASSEMBLE_IM2COL:
            return bc_assemble_input_im2col(CP(float, 0), I(1), CP(float, 2), CP(float, 3), I(4), I(5), I(6), I(7), I(8), MP(bc_half, 9), s);
        case BC_OP_ROWCHAIN:
            return bc_rowchain(I(0), I(1), I(2), I(3), CP(bc_half, 4), CP(float, 5), CP(unsigned long long, 6), CP(float, 7), CP(float, 8), I(9),
                               F(10), CP(bc_half, 11), CP(bc_half, 12), CP(bc_half, 13), I(14), I(15), I(16), CP(bc_half, 17), CP(float, 18),
                               MP(bc_half, 19), MP(bc_half, 20), MP(bc_half, 21), I(22), MP(unsigned long long, 23), F(24), F(25),
                               CP(float, 26), CP(int, 27), I(28), MP(float, 29), I(30), s);
        case BC_OP_SIGNAL: {
            hipEvent_t ev;
            int rc = plan_event(pl, I(0), &ev);
            if (rc) return rc;
            BC_CHECK_HIP(hipEventRecord(ev, reinterpret_cast<hipStream_t>(s)));
            return 0;
        }
        case BC_OP_WAIT: {
            hipEvent_t ev;
            int rc = plan_event(pl, I(0), &ev);
            if (rc) return rc;
            BC_CHECK_HIP(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(s), ev, 0));
            return 0;
        }
        default: bc_set_error("plan: unknown op %d", r.op); return 1;
    }
}
#undef P
#undef CP
#undef MP
#undef I
#undef L
#undef F

int run_eager(BcPlan* pl, Seg& sg, hipStream_t* streams, int n) {
    for (Rec& r : sg.recs) {
        if (!r.enabled) continue;
        int rc = launch_rec(pl, r, streams, n);
        if (rc) return rc;
    }
    return 0;
}

int get_streams(BcPlan* pl, const bc_stream* streams, int n, hipStream_t* out) {
    BC_CHECK_ARG(n >= 0 && n <= kMaxStreams, "plan: at most %d streams", kMaxStreams);
    if (n == 0) {                                            // loader-created streams (plain C hosts)
        if (pl->n_own_streams == 0) {
            for (int i = 0; i < 3; ++i) BC_CHECK_HIP(hipStreamCreateWithFlags(&pl->own_streams[i], hipStreamNonBlocking));
            pl->n_own_streams = 3;
        }
        for (int i = 0; i < kMaxStreams; ++i) out[i] = pl->own_streams[i < pl->n_own_streams ? i : 0];
        return 0;
    }
    for (int i = 0; i < kMaxStreams; ++i) out[i] = reinterpret_cast<hipStream_t>(streams[i < n ? i : 0]);
    return 0;
}

#define SEG(pl, s)                                                                                       \
    BC_CHECK_ARG((pl) != nullptr && (s) >= 0 && (s) < (int)(pl)->segs.size(), "plan: bad segment id %d", (s)); \
    Seg& sg = (pl)->segs[(s)]

// ---- file format helpers ----
struct Writer {
    FILE* f;
    bool ok = true;
    void raw(const void* p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void str(const std::string& s) { u32((uint32_t)s.size()); raw(s.data(), s.size()); }
};
// pointer -> (buffer index, offset); null stays null (index -1)
bool relocate_out(const std::vector<Buf>& bufs, uint64_t addr, int64_t& idx, uint64_t& off) {
    if (addr == 0) { idx = -1; off = 0; return true; }
    for (size_t i = 0; i < bufs.size(); ++i)
        if (addr >= bufs[i].addr && addr < bufs[i].addr + std::max<uint64_t>(bufs[i].bytes, 1)) {
            idx = (int64_t)i; off = addr - bufs[i].addr; return true;
        }
    return false;
}

}  // namespace

extern "C" int bc_plan_create(BcPlan** out) {
    BC_CHECK_ARG(out != nullptr, "bc_plan_create: null output");
    *out = new BcPlan();
    return 0;
}

extern "C" int bc_plan_destroy(BcPlan* pl) {
    if (!pl) return 0;
    for (Seg& s : pl->segs)
        if (s.graph) (void)hipGraphExecDestroy(s.graph);
    for (hipEvent_t e : pl->events)
        if (e) (void)hipEventDestroy(e);
    for (int i = 0; i < pl->n_own_streams; ++i) (void)hipStreamDestroy(pl->own_streams[i]);
    if (pl->arena) (void)hipFree(pl->arena);
    delete pl;
    return 0;
}

extern "C" int bc_plan_segment(BcPlan* pl, const char* name) {
    if (!pl) return -1;
    pl->segs.emplace_back();
    pl->segs.back().name = name ? name : "";
    return (int)pl->segs.size() - 1;
}

extern "C" int bc_plan_new_event(BcPlan* pl) {
    if (!pl) return -1;
    pl->events.push_back(nullptr);          // created on first use: plans can be compiled (and saved) on a machine without a GPU
    return (int)pl->events.size() - 1;
}

// (the two add functions return the launch INDEX, so errors are negative)
#define ADD_CHECK(cond, ...)                             \
    do {                                                 \
        if (!(cond)) { bc_set_error(__VA_ARGS__); return -1; } \
    } while (0)

extern "C" int bc_plan_add_gemm(BcPlan* pl, int seg, int stream_id, const BcGemm* g) {
    ADD_CHECK(pl != nullptr && seg >= 0 && seg < (int)pl->segs.size(), "bc_plan_add_gemm: bad segment id %d", seg);
    Seg& sg = pl->segs[seg];
    ADD_CHECK(g != nullptr && stream_id >= 0 && stream_id < kMaxStreams, "bc_plan_add_gemm: bad arguments");
    Rec r;
    r.op = BC_OP_GEMM;
    r.sid = stream_id;
    r.g = *g;
    sg.recs.push_back(r);
    return (int)sg.recs.size() - 1;
}

extern "C" int bc_plan_add_op(BcPlan* pl, int seg, int stream_id, int op, const uint64_t* args, int nargs) {
    ADD_CHECK(pl != nullptr && seg >= 0 && seg < (int)pl->segs.size(), "bc_plan_add_op: bad segment id %d", seg);
    Seg& sg = pl->segs[seg];
    const char* sig = op_signature(op);
    ADD_CHECK(sig != nullptr && (int)strlen(sig) == nargs && args != nullptr, "bc_plan_add_op: op %d takes %d arguments, got %d", op,
              sig ? (int)strlen(sig) : -1, nargs);
    ADD_CHECK(stream_id >= 0 && stream_id < kMaxStreams, "bc_plan_add_op: stream id %d out of range", stream_id);
    Rec r;
    r.op = op;
    r.sid = stream_id;
    r.a.assign(args, args + nargs);
    sg.recs.push_back(r);
    return (int)sg.recs.size() - 1;
}

extern "C" int bc_plan_set_slab(BcPlan* pl, int stream_id, float* slab) {
    BC_CHECK_ARG(pl && stream_id >= 0 && stream_id < kMaxStreams, "bc_plan_set_slab: bad stream id");
    pl->slab[stream_id] = slab;
    return 0;
}

extern "C" int bc_plan_enable(BcPlan* pl, int seg, int index, int enabled) {
    SEG(pl, seg);
    BC_CHECK_ARG(index >= 0 && index < (int)sg.recs.size(), "bc_plan_enable: bad launch index");
    sg.recs[index].enabled = enabled;
    return 0;
}

extern "C" int bc_plan_num_launches(BcPlan* pl, int seg) {
    if (!pl || seg < 0 || seg >= (int)pl->segs.size()) return -1;
    return (int)pl->segs[seg].recs.size();
}

extern "C" int bc_step(BcPlan* pl, int seg, const bc_stream* streams, int nstreams) {
    SEG(pl, seg);
    hipStream_t st[kMaxStreams];
    int rc = get_streams(pl, streams, nstreams, st);
    if (rc) return rc;
    if (sg.graph) {
        BC_CHECK_HIP(hipGraphLaunch(sg.graph, st[0]));
        return 0;
    }
    return run_eager(pl, sg, st, kMaxStreams);
}

extern "C" int bc_plan_capture(BcPlan* pl, int seg, const bc_stream* streams, int nstreams) {
    SEG(pl, seg);
    hipStream_t st[kMaxStreams];
    int rc = get_streams(pl, streams, nstreams, st);
    if (rc) return rc;
    if (sg.graph) { (void)hipGraphExecDestroy(sg.graph); sg.graph = nullptr; }
    BC_CHECK_HIP(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
    rc = run_eager(pl, sg, st, kMaxStreams);
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(st[0], &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    BC_CHECK_HIP(e);
    e = hipGraphInstantiate(&sg.graph, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    BC_CHECK_HIP(e);
    return 0;
}

extern "C" int bc_plan_release(BcPlan* pl, int seg) {
    SEG(pl, seg);
    if (sg.graph) { BC_CHECK_HIP(hipGraphExecDestroy(sg.graph)); sg.graph = nullptr; }
    return 0;
}

extern "C" int bc_plan_capture_loop(BcPlan* pl, const int* seg_sequence, int n, const bc_stream* streams, int nstreams,
                                    void** graph_exec_out) {
    BC_CHECK_ARG(pl && seg_sequence && n > 0 && graph_exec_out, "bc_plan_capture_loop: bad arguments");
    hipStream_t st[kMaxStreams];
    int rc = get_streams(pl, streams, nstreams, st);
    if (rc) return rc;
    BC_CHECK_HIP(hipStreamBeginCapture(st[0], hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n && !rc; ++i) {
        if (seg_sequence[i] < 0 || seg_sequence[i] >= (int)pl->segs.size()) { bc_set_error("bc_plan_capture_loop: bad segment id"); rc = 1; break; }
        rc = run_eager(pl, pl->segs[seg_sequence[i]], st, kMaxStreams);
    }
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(st[0], &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    BC_CHECK_HIP(e);
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    BC_CHECK_HIP(e);
    *graph_exec_out = exec;
    return 0;
}

extern "C" int bc_plan_run_timed(BcPlan* pl, int seg, bc_stream stream, float* ms_out) {
    SEG(pl, seg);
    BC_CHECK_ARG(ms_out != nullptr, "bc_plan_run_timed: null output");
    hipStream_t st[kMaxStreams];
    for (int i = 0; i < kMaxStreams; ++i) st[i] = reinterpret_cast<hipStream_t>(stream);   // serial: isolates every launch
    EventSet ev(2 * sg.recs.size());
    for (auto& e : ev.ev) BC_CHECK_HIP(hipEventCreate(&e));
    int rc = 0;
    for (size_t i = 0; i < sg.recs.size() && !rc; ++i) {
        BC_CHECK_HIP(hipEventRecord(ev[2 * i], st[0]));
        if (sg.recs[i].enabled) rc = launch_rec(pl, sg.recs[i], st, kMaxStreams);
        BC_CHECK_HIP(hipEventRecord(ev[2 * i + 1], st[0]));
    }
    if (!rc) {
        BC_CHECK_HIP(hipStreamSynchronize(st[0]));
        for (size_t i = 0; i < sg.recs.size(); ++i) BC_CHECK_HIP(hipEventElapsedTime(&ms_out[i], ev[2 * i], ev[2 * i + 1]));
    }
    return rc;
}

// Like bc_plan_run_timed, but a split-K GEMM's time is divided at an event recorded between its main kernel and its reducer:
// ms_main[i] is then the duration rocprofv3 reports for the main kernel of launch i, ms_reduce[i] that of its reducer (0 if none).
extern "C" int bc_plan_run_timed_kernels(BcPlan* pl, int seg, bc_stream stream, float* ms_main, float* ms_reduce) {
    SEG(pl, seg);
    BC_CHECK_ARG(ms_main != nullptr && ms_reduce != nullptr, "bc_plan_run_timed_kernels: null output");
    hipStream_t st[kMaxStreams];
    for (int i = 0; i < kMaxStreams; ++i) st[i] = reinterpret_cast<hipStream_t>(stream);
    const size_t n = sg.recs.size();
    EventSet ev(3 * n);
    for (auto& e : ev.ev) BC_CHECK_HIP(hipEventCreate(&e));
    std::vector<char> split(n, 0);
    int rc = 0;
    for (size_t i = 0; i < n && !rc; ++i) {
        BC_CHECK_HIP(hipEventRecord(ev[3 * i], st[0]));
        bc_gemm_set_probe(ev[3 * i + 1]);
        if (sg.recs[i].enabled) rc = launch_rec(pl, sg.recs[i], st, kMaxStreams);
        split[i] = bc_gemm_probe_hit();
        bc_gemm_set_probe(nullptr);
        BC_CHECK_HIP(hipEventRecord(ev[3 * i + 2], st[0]));
    }
    if (!rc) {
        BC_CHECK_HIP(hipStreamSynchronize(st[0]));
        for (size_t i = 0; i < n; ++i) {
            if (split[i]) {
                BC_CHECK_HIP(hipEventElapsedTime(&ms_main[i], ev[3 * i], ev[3 * i + 1]));
                BC_CHECK_HIP(hipEventElapsedTime(&ms_reduce[i], ev[3 * i + 1], ev[3 * i + 2]));
            } else {
                BC_CHECK_HIP(hipEventElapsedTime(&ms_main[i], ev[3 * i], ev[3 * i + 2]));
                ms_reduce[i] = 0.f;
            }
        }
    }
    return rc;
}

// Concurrent replay with sparse timestamps (round 5, tools/concurrent_timeline.py): the segment is replayed eagerly on its own streams -
// the two queues overlap as they do inside the graph - and a timing event is recorded on the launch's OWN stream behind each launch listed
// in `marks` (ascending launch indices); ms_out[k] = time from the start of the replay (an event on stream 0, in front of the fork) to
// that event.  A few dozen marks cost nothing measurable; an event behind EVERY launch would add ~5 us each and distort the overlap.
extern "C" int bc_plan_run_marked(BcPlan* pl, int seg, const bc_stream* streams, int nstreams, const int* marks, int nmarks, float* ms_out) {
    SEG(pl, seg);
    BC_CHECK_ARG(marks != nullptr && ms_out != nullptr && nmarks >= 0, "bc_plan_run_marked: bad arguments");
    hipStream_t st[kMaxStreams];
    int rc = get_streams(pl, streams, nstreams, st);
    if (rc) return rc;
    EventSet ev((size_t)nmarks + 1);
    for (auto& e : ev.ev) BC_CHECK_HIP(hipEventCreate(&e));
    BC_CHECK_HIP(hipEventRecord(ev[0], st[0]));
    int k = 0;
    for (size_t i = 0; i < sg.recs.size() && !rc; ++i) {
        Rec& r = sg.recs[i];
        if (r.enabled) rc = launch_rec(pl, r, st, kMaxStreams);
        while (k < nmarks && marks[k] == (int)i) {
            BC_CHECK_HIP(hipEventRecord(ev[1 + k], st[r.sid < kMaxStreams ? r.sid : 0]));
            ++k;
        }
    }
    BC_CHECK_ARG(rc || k == nmarks, "bc_plan_run_marked: marks must be ascending launch indices of the segment");
    if (!rc) {
        for (int i = 0; i < kMaxStreams; ++i) BC_CHECK_HIP(hipStreamSynchronize(st[i]));
        for (int j = 0; j < nmarks; ++j) BC_CHECK_HIP(hipEventElapsedTime(&ms_out[j], ev[0], ev[1 + j]));
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------- save / load
extern "C" int bc_plan_save(BcPlan* pl, const char* path, const BcPlanBuffer* bufs, int nbufs) {
    BC_CHECK_ARG(pl && path && bufs && nbufs > 0, "bc_plan_save: bad arguments");
    std::vector<Buf> tb(nbufs);
    for (int i = 0; i < nbufs; ++i) {
        tb[i].name = bufs[i].name ? bufs[i].name : "";
        tb[i].addr = (uint64_t)(uintptr_t)bufs[i].address;
        tb[i].bytes = (uint64_t)bufs[i].bytes;
    }
    FILE* f = fopen(path, "wb");
    BC_CHECK_ARG(f != nullptr, "bc_plan_save: cannot open %s", path);
    Writer w{f};
    w.u32(kMagic); w.u32(kVersion); w.u32((uint32_t)sizeof(BcGemm)); w.u32((uint32_t)nbufs);
    for (int i = 0; i < nbufs; ++i) {
        w.str(tb[i].name);
        w.u64(tb[i].bytes);
        const uint32_t has_data = bufs[i].host_data != nullptr;
        w.u32(has_data);
        if (has_data) w.raw(bufs[i].host_data, (size_t)tb[i].bytes);
    }
    bool ok = true;
    auto put_ptr = [&](uint64_t addr) {
        int64_t idx; uint64_t off;
        if (!relocate_out(tb, addr, idx, off)) { ok = false; idx = -1; off = 0; }
        w.u64((uint64_t)idx); w.u64(off);
    };
    w.u32((uint32_t)pl->events.size());
    for (int s = 0; s < kMaxStreams; ++s) put_ptr((uint64_t)(uintptr_t)pl->slab[s]);
    w.u32((uint32_t)pl->segs.size());
    for (Seg& sg : pl->segs) {
        w.str(sg.name);
        w.u32((uint32_t)sg.recs.size());
        for (Rec& r : sg.recs) {
            w.u32((uint32_t)r.op); w.u32((uint32_t)r.sid); w.u32((uint32_t)r.enabled);
            if (r.op == BC_OP_GEMM) {
                BcGemm g = r.g;
                g.slab = nullptr;
                w.raw(&g, sizeof(g));
                for (size_t fo : kGemmPtrFields) {
                    uint64_t addr;
                    memcpy(&addr, reinterpret_cast<const char*>(&r.g) + fo, 8);
                    put_ptr(addr);
                }
            } else {
                const char* sig = op_signature(r.op);
                w.u32((uint32_t)r.a.size());
                for (size_t k = 0; k < r.a.size(); ++k) {
                    if (sig[k] == 'p') put_ptr(r.a[k]);
                    else w.u64(r.a[k]);
                }
            }
        }
    }
    const bool wrote = w.ok;
    fclose(f);
    BC_CHECK_ARG(ok, "bc_plan_save: a launch references memory outside the %d declared buffers", nbufs);
    BC_CHECK_ARG(wrote, "bc_plan_save: write to %s failed", path);
    return 0;
}

extern "C" int bc_plan_load(const char* path, BcPlan** out) {
    BC_CHECK_ARG(path && out, "bc_plan_load: bad arguments");
    FILE* f = fopen(path, "rb");
    BC_CHECK_ARG(f != nullptr, "bc_plan_load: cannot open %s", path);
    BcPlan* pl = new BcPlan();
    PlanImage img;
    // the format and every check on it live in plan_format.h (also built as plain host C++ under the sanitizers)
    const std::string why = parse_plan(
        f, img,
        [&](uint64_t bytes) -> uint64_t {
            if (hipMalloc(&pl->arena, (size_t)bytes) != hipSuccess) { pl->arena = nullptr; return 0; }
            if (hipMemset(pl->arena, 0, (size_t)bytes) != hipSuccess) return 0;
            return (uint64_t)(uintptr_t)pl->arena;
        },
        [&](uint64_t off, const char* host, size_t n) {
            return hipMemcpy(static_cast<char*>(pl->arena) + off, host, n, hipMemcpyHostToDevice) == hipSuccess;
        });
    fclose(f);
    if (!why.empty()) {
        bc_plan_destroy(pl);
        bc_set_error("bc_plan_load(%s): %s", path, why.c_str());
        return 1;
    }
    pl->bufs = std::move(img.bufs);
    for (uint32_t i = 0; i < img.nevents; ++i)
        if (bc_plan_new_event(pl) < 0) { bc_plan_destroy(pl); bc_set_error("bc_plan_load(%s): event creation failed", path); return 1; }
    for (int s = 0; s < kMaxStreams; ++s) pl->slab[s] = reinterpret_cast<float*>(img.slab[s]);
    for (SegImage& si : img.segs) {
        pl->segs.emplace_back();
        pl->segs.back().name = std::move(si.name);
        pl->segs.back().recs = std::move(si.recs);
    }
    *out = pl;
    return 0;
}

extern "C" int bc_plan_buffer(BcPlan* pl, const char* name, void** ptr, long long* bytes) {
    BC_CHECK_ARG(pl && name && ptr, "bc_plan_buffer: bad arguments");
    for (Buf& b : pl->bufs)
        if (b.name == name) {
            *ptr = reinterpret_cast<void*>(b.addr);
            if (bytes) *bytes = (long long)b.bytes;
            return 0;
        }
    bc_set_error("bc_plan_buffer: no buffer named '%s'", name);
    return 1;
}

extern "C" int bc_plan_find_segment(BcPlan* pl, const char* name) {
    if (!pl || !name) return -1;
    for (size_t i = 0; i < pl->segs.size(); ++i)
        if (pl->segs[i].name == name) return (int)i;
    return -1;
}
