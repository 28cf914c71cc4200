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

void bc_gemm_set_probe(hipEvent_t e);      // gemm.hip (timing probe between a split-K GEMM's main kernel and its reducer)
bool bc_gemm_probe_hit();

namespace {

// argument kinds of the recordable entry points (stream argument excluded): p = device pointer, i = int, f = float, l = long long
const char* op_signature(int op) {
    switch (op) {
        case BC_OP_GN_STATS: return "piiip";
        case BC_OP_GN_FINALIZE: return "pipiiiifppp";
        case BC_OP_GN_APPLY_FUSED: return "pipippiiifppip";
        case BC_OP_GN_APPLY: return "pipiiipip";
        case BC_OP_LAYERNORM: return "piiippfpi";
        case BC_OP_ATTENTION:
        case BC_OP_ATTENTION_CAUSAL: return "ppppiiiiiiiiillllf";
        case BC_OP_ASSEMBLE_INPUT: return "pipppiiiiiiip";
        case BC_OP_TIMESTEP_EMBEDDING: return "ppfiip";
        case BC_OP_TIMESTEP_EMBEDDING_TABLE: return "piiip";
        case BC_OP_CFG_SCHEDULER_STEP: return "pppppfiiipi";
        case BC_OP_EMBED_TOKENS: return "pppiiiip";
        case BC_OP_SOFTMAX_ROWS: return "piii";
        case BC_OP_PATCHIFY: return "piiiiip";
        case BC_OP_ADD_CLS_POS: return "pppiiip";
        case BC_OP_SILU: return "ppl";
        case BC_OP_NCHW_TO_NHWC_F16: return "piiiiip";
        case BC_OP_NHWC_TO_NCHW: return "piiiipi";
        case BC_OP_GAUSSIAN_SAMPLE: return "ppiiifp";
        case BC_OP_SIGNAL:
        case BC_OP_WAIT: return "i";
        case BC_OP_ROWCHAIN: return "iiiipppppifpppiiipppppipffppipi";
        case BC_OP_ASSEMBLE_IM2COL: return "pippiiiiip";
        case BC_OP_MEMSET_ZERO: return "pl";
        default: return nullptr;
    }
}

// byte offsets of the pointer fields of BcGemm (relocated on save / load)
const size_t kGemmPtrFields[] = {
    offsetof(BcGemm, A), offsetof(BcGemm, A2), offsetof(BcGemm, W), offsetof(BcGemm, bias), offsetof(BcGemm, rowvec),
    offsetof(BcGemm, rowvec_idx), offsetof(BcGemm, colscale), offsetof(BcGemm, alpha_dev), offsetof(BcGemm, alpha_idx),
    offsetof(BcGemm, R), offsetof(BcGemm, R2), offsetof(BcGemm, C), offsetof(BcGemm, gn_tot), offsetof(BcGemm, a_affine),
    offsetof(BcGemm, a_tot1), offsetof(BcGemm, a_tot2), offsetof(BcGemm, a_gamma), offsetof(BcGemm, a_beta),
    offsetof(BcGemm, ln_colsum), offsetof(BcGemm, C_t)};

struct Rec {
    int op = 0, sid = 0, enabled = 1;
    std::vector<uint64_t> a;     // generic arguments (floats as their 32-bit pattern)
    BcGemm g;                    // BC_OP_GEMM
};

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

struct Buf {
    std::string name;
    uint64_t addr = 0;           // address the records were built against (device, or host when compiled without a GPU)
    uint64_t bytes = 0;
    uint64_t arena_off = 0;      // loader: offset inside the arena
};

constexpr int kMaxStreams = 8;

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
const uint32_t kMagic = 0x4E4C5042u;   // "BPLN"
const uint32_t kVersion = 2;        // 2: BcGemm grew ln_colsum / C_t (round 4)

struct Writer {
    FILE* f;
    bool ok = true;
    void raw(const void* p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void str(const std::string& s) { u32((uint32_t)s.size()); raw(s.data(), s.size()); }
};
struct Reader {
    FILE* f;
    bool ok = true;
    void raw(void* p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; }
    uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
    uint64_t u64() { uint64_t v = 0; raw(&v, 8); return v; }
    std::string str() { uint32_t n = u32(); std::string s(ok && n < (1u << 20) ? n : 0, '\0'); raw(&s[0], s.size()); return s; }
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
    Reader rd{f};
    if (fseek(f, 0, SEEK_END)) { fclose(f); bc_set_error("bc_plan_load(%s): seek failed", path); return 1; }
    const uint64_t file_bytes = (uint64_t)ftell(f);
    rewind(f);
    constexpr uint64_t kMaxArena = 1ull << 40;       // 1 TiB: far above any real plan, far below overflow of the running sum
    BcPlan* pl = new BcPlan();
    auto fail = [&](const char* why) { fclose(f); bc_plan_destroy(pl); bc_set_error("bc_plan_load(%s): %s", path, why); return 1; };
    if (rd.u32() != kMagic) return fail("not a plan file");
    if (rd.u32() != kVersion) return fail("unsupported plan version");
    if (rd.u32() != sizeof(BcGemm)) return fail("BcGemm layout differs from this library build");
    const uint32_t nb = rd.u32();
    if (!rd.ok || nb > (1u << 20)) return fail("corrupt header");
    // pass 1: sizes (data blobs are skipped), then one arena
    std::vector<long> data_pos(nb, -1);
    uint64_t total = 0;
    pl->bufs.resize(nb);
    for (uint32_t i = 0; i < nb; ++i) {
        Buf& b = pl->bufs[i];
        b.name = rd.str();
        b.bytes = rd.u64();
        if (!rd.ok || b.bytes > kMaxArena) return fail("corrupt buffer size");
        b.arena_off = total;
        total += (b.bytes + 255) & ~255ull;
        if (total > kMaxArena) return fail("buffer table larger than any device");
        if (rd.u32()) {
            data_pos[i] = ftell(f);
            if (data_pos[i] < 0 || (uint64_t)data_pos[i] + b.bytes > file_bytes) return fail("buffer data runs past the end of the file");
            if (fseek(f, (long)b.bytes, SEEK_CUR)) return fail("truncated buffer data");
        }
        if (!rd.ok) return fail("truncated buffer table");
    }
    if (hipMalloc(&pl->arena, (size_t)std::max<uint64_t>(total, 256)) != hipSuccess) return fail("hipMalloc of the plan arena failed");
    if (hipMemset(pl->arena, 0, (size_t)std::max<uint64_t>(total, 256)) != hipSuccess) return fail("hipMemset failed");
    const long after_table = ftell(f);
    std::vector<char> host;
    for (uint32_t i = 0; i < nb; ++i) {
        if (data_pos[i] < 0) continue;
        host.resize((size_t)pl->bufs[i].bytes);
        if (fseek(f, data_pos[i], SEEK_SET)) return fail("seek failed");
        rd.raw(host.data(), host.size());
        if (!rd.ok) return fail("truncated buffer data");
        if (hipMemcpy(static_cast<char*>(pl->arena) + pl->bufs[i].arena_off, host.data(), host.size(), hipMemcpyHostToDevice) != hipSuccess)
            return fail("hipMemcpy failed");
    }
    if (fseek(f, after_table, SEEK_SET)) return fail("seek failed");
    for (Buf& b : pl->bufs) b.addr = (uint64_t)(uintptr_t)(static_cast<char*>(pl->arena) + b.arena_off);
    bool bad_ptr = false;
    auto get_ptr = [&]() -> uint64_t {
        const int64_t idx = (int64_t)rd.u64();
        const uint64_t off = rd.u64();
        if (idx < 0) return 0;
        if ((uint64_t)idx >= pl->bufs.size() || off > pl->bufs[idx].bytes) { bad_ptr = true; return 0; }
        return pl->bufs[idx].addr + off;
    };
    const uint32_t nev = rd.u32();
    if (!rd.ok || nev > (1u << 20)) return fail("corrupt event count");
    for (uint32_t i = 0; i < nev; ++i)
        if (bc_plan_new_event(pl) < 0) return fail("event creation failed");
    for (int s = 0; s < kMaxStreams; ++s) pl->slab[s] = reinterpret_cast<float*>(get_ptr());
    const uint32_t nseg = rd.u32();
    if (!rd.ok || nseg > (1u << 16)) return fail("corrupt segment count");
    for (uint32_t si = 0; si < nseg; ++si) {
        pl->segs.emplace_back();
        Seg& sg = pl->segs.back();
        sg.name = rd.str();
        const uint32_t nr = rd.u32();
        if (!rd.ok || nr > (1u << 22)) return fail("corrupt launch count");
        sg.recs.resize(nr);
        for (Rec& r : sg.recs) {
            const uint32_t op = rd.u32(), sid = rd.u32(), enabled = rd.u32();
            if (!rd.ok || sid >= (uint32_t)kMaxStreams) return fail("stream id out of range");
            if (op != (uint32_t)BC_OP_GEMM && !op_signature((int)op)) return fail("unknown op code");
            r.op = (int)op; r.sid = (int)sid; r.enabled = enabled ? 1 : 0;
            if (r.op == BC_OP_GEMM) {
                rd.raw(&r.g, sizeof(r.g));
                for (size_t fo : kGemmPtrFields) {
                    const uint64_t addr = get_ptr();
                    memcpy(reinterpret_cast<char*>(&r.g) + fo, &addr, 8);
                }
                r.g.slab = nullptr;
            } else {
                const char* sig = op_signature(r.op);
                const uint32_t na = rd.u32();
                if (!sig || strlen(sig) != na) return fail("unknown op or argument count");
                r.a.resize(na);
                for (uint32_t k = 0; k < na; ++k) r.a[k] = sig[k] == 'p' ? get_ptr() : rd.u64();
            }
            if (!rd.ok) return fail("truncated launch record");
        }
    }
    fclose(f);
    if (bad_ptr) { bc_plan_destroy(pl); bc_set_error("bc_plan_load(%s): pointer outside its buffer", path); return 1; }
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
