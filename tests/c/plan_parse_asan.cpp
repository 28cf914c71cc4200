// Host-only harness of the `.bcplan` parser (blobctrl_amd/csrc/plan_format.h) for AddressSanitizer / UBSan runs (SURVEY section 5,
// VERDICT r3 item 10):  g++ -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=all plan_parse_asan.cpp
// usage: plan_parse_asan FILE...   - parses every file with malloc / memcpy standing in for the device arena and prints one line per
// file: "ok <buffers> <segments> <launches>" or "rejected: <reason>".  Any sanitizer report or crash fails the test that runs it.
#include <stdlib.h>
#include "../../blobctrl_amd/csrc/plan_format.h"

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        FILE* f = fopen(argv[i], "rb");
        if (!f) { printf("rejected: cannot open\n"); continue; }
        bcplan::PlanImage img;
        char* arena = nullptr;
        uint64_t arena_bytes = 0;
        const std::string why = bcplan::parse_plan(
            f, img,
            [&](uint64_t bytes) -> uint64_t {
                if (bytes > (1ull << 32)) return 0;                  // (a test host does not allocate what a 288 GB device could)
                arena = static_cast<char*>(calloc(1, (size_t)bytes));
                arena_bytes = bytes;
                return (uint64_t)(uintptr_t)arena;
            },
            [&](uint64_t off, const char* host, size_t n) {
                if (off + n > arena_bytes) return false;
                memcpy(arena + off, host, n);
                return true;
            });
        fclose(f);
        if (!why.empty()) {
            printf("rejected: %s\n", why.c_str());
        } else {
            size_t launches = 0;
            for (auto& s : img.segs) launches += s.recs.size();
            // touch every relocated pointer's target byte: a pointer the parser accepted must lie inside the arena
            unsigned long long sum = 0;
            auto touch = [&](uint64_t addr) {
                if (!addr) return;
                const uint64_t base = (uint64_t)(uintptr_t)arena;
                if (addr < base || addr > base + arena_bytes) { printf("BAD POINTER\n"); abort(); }
                if (addr < base + arena_bytes) sum += (unsigned char)arena[addr - base];
            };
            for (auto& s : img.segs)
                for (auto& r : s.recs) {
                    if (r.op == BC_OP_GEMM) {
                        for (size_t fo : bcplan::kGemmPtrFields) { uint64_t a; memcpy(&a, reinterpret_cast<const char*>(&r.g) + fo, 8); touch(a); }
                    } else {
                        const char* sig = bcplan::op_signature(r.op);
                        for (size_t k = 0; k < r.a.size(); ++k) if (sig[k] == 'p') touch(r.a[k]);
                    }
                }
            printf("ok %zu %zu %zu %llu\n", img.bufs.size(), img.segs.size(), launches, sum & 1);
        }
        free(arena);
    }
    return 0;
}
