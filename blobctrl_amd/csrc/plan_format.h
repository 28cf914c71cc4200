// The `.bcplan` file format and its PARSER, free of any HIP call so that it also builds as plain host C++ (tests/c/plan_parse_asan.cpp
// runs it under AddressSanitizer / UBSan over truncated and bit-flipped files: SURVEY section 5 "sanitizers").
// plan.hip supplies the device side through two callbacks (allocate the arena, upload one buffer); everything the file says about
// sizes, counts, op codes, stream ids and pointers is validated here before it is used.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <functional>
#include <string>
#include <vector>
#include "../../include/blobctrl_hip.h"

namespace bcplan {

constexpr int kMaxStreams = 8;
const uint32_t kMagic = 0x4E4C5042u;   // "BPLN"
const uint32_t kVersion = 5;           // 2: BcGemm grew ln_colsum / C_t, GroupNorm statistics totals; 3: + BC_OP_ROWCHAIN_MIDX / _PACK_KV (round 4);
                                       // 4: + BC_OP_ROWCHAIN_SUM, BC_CHAIN_OUT_FFP; 5: BcGemm grew w_bstride / vec_bstride / sm_group / sm_valid,
                                       //    + BC_OP_CTX_FOLD (round 5)
const uint32_t kOldestReadable = 5;    // (the records hold BcGemm by value: a file of another layout is refused by the size check below anyway)

// argument kinds of the recordable entry points (stream argument excluded): p = device pointer, i = int, f = float, l = long long
inline const char* op_signature(int op) {
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
        case BC_OP_ROWCHAIN_MIDX: return "iiipppppifppf";
        case BC_OP_ROWCHAIN_PACK_KV: return "pipiiiip";
        case BC_OP_ROWCHAIN_SUM: return "iiipippp";
        case BC_OP_CTX_FOLD: return "pipiiiiifppppppp";
        case BC_OP_DUP_HALVES: return "plplplplplpl";
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


struct Buf {
    std::string name;
    uint64_t addr = 0;           // address the records were built against (device, or host when compiled without a GPU)
    uint64_t bytes = 0;
    uint64_t arena_off = 0;      // loader: offset inside the arena
};

struct SegImage {
    std::string name;
    std::vector<Rec> recs;
};

struct PlanImage {               // what a plan file holds, with every pointer already relocated into the loader's arena
    std::vector<Buf> bufs;
    uint32_t nevents = 0;
    uint64_t slab[kMaxStreams] = {};
    std::vector<SegImage> segs;
    uint64_t arena_bytes = 0;
};

struct Reader {
    FILE* f;
    bool ok = true;
    void raw(void* p, size_t n) { if (ok && n && fread(p, 1, n, f) != n) ok = false; }
    uint32_t u32() { uint32_t v = 0; raw(&v, 4); return v; }
    uint64_t u64() { uint64_t v = 0; raw(&v, 8); return v; }
    std::string str() { uint32_t n = u32(); std::string s(ok && n < (1u << 20) ? n : 0, '\0'); raw(s.empty() ? nullptr : &s[0], s.size()); return s; }
};

// alloc(total_bytes) -> arena base address (0 = failure); upload(arena_offset, host bytes, n) -> false on failure.
// Returns "" on success, else the reason.  Nothing of `img` may be used after a failure.
inline std::string parse_plan(FILE* f, PlanImage& img, const std::function<uint64_t(uint64_t)>& alloc,
                              const std::function<bool(uint64_t, const char*, size_t)>& upload) {
    Reader rd{f};
    if (fseek(f, 0, SEEK_END)) return "seek failed";
    const long end = ftell(f);
    if (end < 0) return "seek failed";
    const uint64_t file_bytes = (uint64_t)end;
    rewind(f);
    constexpr uint64_t kMaxArena = 1ull << 40;       // 1 TiB: far above any real plan, far below overflow of the running sum
    if (rd.u32() != kMagic) return "not a plan file";
    { const uint32_t v = rd.u32(); if (v < kOldestReadable || v > kVersion) return "unsupported plan version"; }
    if (rd.u32() != sizeof(BcGemm)) return "BcGemm layout differs from this library build";
    const uint32_t nb = rd.u32();
    if (!rd.ok || nb > (1u << 20)) return "corrupt header";
    // pass 1: sizes (data blobs are skipped), then one arena
    std::vector<long> data_pos(nb, -1);
    uint64_t total = 0;
    img.bufs.resize(nb);
    for (uint32_t i = 0; i < nb; ++i) {
        Buf& b = img.bufs[i];
        b.name = rd.str();
        b.bytes = rd.u64();
        if (!rd.ok || b.bytes > kMaxArena) return "corrupt buffer size";
        b.arena_off = total;
        total += (b.bytes + 255) & ~255ull;
        if (total > kMaxArena) return "buffer table larger than any device";
        if (rd.u32()) {
            data_pos[i] = ftell(f);
            if (data_pos[i] < 0 || (uint64_t)data_pos[i] + b.bytes > file_bytes) return "buffer data runs past the end of the file";
            if (fseek(f, (long)b.bytes, SEEK_CUR)) return "truncated buffer data";
        }
        if (!rd.ok) return "truncated buffer table";
    }
    // (a file cannot describe more initialised bytes than it holds; the zero-filled workspace is bounded by kMaxArena above)
    img.arena_bytes = total > 256 ? total : 256;
    const uint64_t base = alloc(img.arena_bytes);
    if (!base) return "allocation of the plan arena failed";
    const long after_table = ftell(f);
    if (after_table < 0) return "seek failed";
    std::vector<char> host;
    for (uint32_t i = 0; i < nb; ++i) {
        if (data_pos[i] < 0) continue;
        host.resize((size_t)img.bufs[i].bytes);
        if (fseek(f, data_pos[i], SEEK_SET)) return "seek failed";
        rd.raw(host.data(), host.size());
        if (!rd.ok) return "truncated buffer data";
        if (!upload(img.bufs[i].arena_off, host.data(), host.size())) return "upload failed";
    }
    if (fseek(f, after_table, SEEK_SET)) return "seek failed";
    for (Buf& b : img.bufs) b.addr = base + b.arena_off;
    bool bad_ptr = false;
    auto get_ptr = [&]() -> uint64_t {
        const int64_t idx = (int64_t)rd.u64();
        const uint64_t off = rd.u64();
        if (idx < 0) return 0;
        if ((uint64_t)idx >= img.bufs.size() || off > img.bufs[(size_t)idx].bytes) { bad_ptr = true; return 0; }
        return img.bufs[(size_t)idx].addr + off;
    };
    img.nevents = rd.u32();
    if (!rd.ok || img.nevents > (1u << 20)) return "corrupt event count";
    for (int s = 0; s < kMaxStreams; ++s) img.slab[s] = get_ptr();
    const uint32_t nseg = rd.u32();
    if (!rd.ok || nseg > (1u << 16)) return "corrupt segment count";
    for (uint32_t si = 0; si < nseg; ++si) {
        img.segs.emplace_back();
        SegImage& sg = img.segs.back();
        sg.name = rd.str();
        const uint32_t nr = rd.u32();
        if (!rd.ok || nr > (1u << 22) || (uint64_t)nr * 12 > file_bytes) return "corrupt launch count";
        sg.recs.resize(nr);
        for (Rec& r : sg.recs) {
            const uint32_t op = rd.u32(), sid = rd.u32(), enabled = rd.u32();
            if (!rd.ok || sid >= (uint32_t)kMaxStreams) return "stream id out of range";
            if (op != (uint32_t)BC_OP_GEMM && !op_signature((int)op)) return "unknown op code";
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
                if (!sig || strlen(sig) != na) return "unknown op or argument count";
                r.a.resize(na);
                for (uint32_t k = 0; k < na; ++k) r.a[k] = sig[k] == 'p' ? get_ptr() : rd.u64();
            }
            if (!rd.ok) return "truncated launch record";
        }
    }
    if (bad_ptr) return "pointer outside its buffer";
    return "";
}

}  // namespace bcplan
