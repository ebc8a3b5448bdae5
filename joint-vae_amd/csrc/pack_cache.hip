// Per-step cache of the re-packed convolution weights.
//
// Every forward-type 5x5 kernel reads its weights in a packed operand layout (pack_elems.h): fp32 [c][tap][o], three split
// bf16 planes, or bf16 units.  The packs depend on the weights only, which change once per step (the optimiser), yet each
// convolution call used to launch its own pack kernel in front of itself: 18 launches of ~5 us per training step of
// BASELINE configs[1], all on the critical path (VERDICT r2 "small-launch diet").  With the cache the host brackets the span
// in which the weights are constant - evaluate() ... backward, or an evaluation pass:
//
//     jvae_pack_cache_begin(stream)   ONE kernel re-packs every known (weight, layout) entry from the CURRENT weights
//     ... convolutions                look their packed operand up (host-side table, no launch)
//     jvae_pack_cache_end()           the optimiser is about to change the weights: lookups fall back to per-call packs
//
// An entry is created the first time a convolution asks for it inside a bracket (that call packs into the new slot itself);
// from the next begin() on it is refreshed with all the others.  Outside a bracket nothing is cached: a stray convolution
// call always sees the weights of that moment.  No reference counterpart (PyTorch re-lays weights out inside cuDNN / MIOpen).
//
// The cache memory is the CALLER's (jvae_pack_cache_configure: a persistent device buffer); the entry table lives on the
// host and travels to the refresh kernel by value (kernel arguments: capture-safe, nothing to upload).
#include <mutex>
#include <string.h>
#include "common.h"
#include "jvae_internal.h"
#include "pack_elems.h"

namespace {

constexpr int MAX_ENTRIES = 48;

struct PackEntry {
    const float* w;
    void* dst;
    int C, O;
    int kind_swap_flip;      // kind | swap << 4 | flip << 5
    unsigned block0;         // first workgroup of this entry in the refresh launch
};
struct PackTable {
    PackEntry e[MAX_ENTRIES];
    int n;
    unsigned blocks;
};

constexpr int ELEMS_PER_BLOCK = 256 * 16;

__global__ __launch_bounds__(256) void pack_refresh_kernel(PackTable t) {
    // workgroup -> entry: the table is tiny and wave-uniform (scalar loads from the kernel arguments)
    int k = 0;
    while (k + 1 < t.n && blockIdx.x >= t.e[k + 1].block0) ++k;
    const PackEntry& en = t.e[k];
    const int kind = en.kind_swap_flip & 15, swap = en.kind_swap_flip >> 4 & 1, flip = en.kind_swap_flip >> 5 & 1;
    const long total = jvae_pack_elems(kind, en.C, en.O);
    const long i0 = (long)(blockIdx.x - en.block0) * ELEMS_PER_BLOCK;
    for (int j = 0; j < 16; ++j) {
        const long i = i0 + j * 256 + threadIdx.x;
        if (i >= total) break;
        if (kind == JVAE_PACK_F32) jvae_pack_f32_elem(en.w, (float*)en.dst, i, en.C, en.O, swap, flip);
        else if (kind == JVAE_PACK_X3) jvae_pack_x3_elem(en.w, (__bf16*)en.dst, i, en.C, en.O, swap, flip);
        else jvae_pack_b8_elem(en.w, (__bf16*)en.dst, i, en.C, en.O, swap, flip);
    }
}

struct State {
    std::mutex mu;
    unsigned char* buf = nullptr;
    size_t bytes = 0, used = 0;
    PackTable tab{};
    bool armed = false;
    long long owner = 0;
    bool fresh[MAX_ENTRIES] = {};     // entry holds the pack of the weights of the current bracket
    long long hits = 0, misses = 0, refreshes = 0;
} S;

}  // namespace

void* jvae_pack_cache_get(int kind, const float* w, int C, int O, int swap, int flip, bool* fresh) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (!S.buf || !S.armed) return nullptr;
    const int ksf = kind | (swap ? 16 : 0) | (flip ? 32 : 0);
    for (int k = 0; k < S.tab.n; ++k) {
        const PackEntry& en = S.tab.e[k];
        if (en.w == w && en.C == C && en.O == O && en.kind_swap_flip == ksf) {
            *fresh = S.fresh[k];
            S.fresh[k] = true;                       // a stale entry is re-packed by this very call
            *fresh ? ++S.hits : ++S.misses;
            return en.dst;
        }
    }
    const size_t need = (jvae_pack_bytes(kind, C, O) + 255) / 256 * 256;
    if (S.tab.n >= MAX_ENTRIES || S.used + need > S.bytes) return nullptr;
    PackEntry& en = S.tab.e[S.tab.n];
    en = PackEntry{w, S.buf + S.used, C, O, ksf, S.tab.blocks};
    S.used += need;
    S.tab.blocks += (unsigned)((jvae_pack_elems(kind, C, O) + ELEMS_PER_BLOCK - 1) / ELEMS_PER_BLOCK);
    S.fresh[S.tab.n] = true;
    ++S.tab.n;
    ++S.misses;
    *fresh = false;
    return en.dst;
}

extern "C" {

// buf: persistent device memory (256-byte aligned) for the packed weights, NULL / 0 switches the cache off.  Drops every entry.
int jvae_pack_cache_configure(void* buf, size_t bytes) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (buf && ((uintptr_t)buf & 255)) return JVAE_EINVAL;
    S.buf = (unsigned char*)buf;
    S.bytes = buf ? bytes : 0;
    S.used = 0;
    S.armed = false;
    memset(&S.tab, 0, sizeof(S.tab));
    memset(S.fresh, 0, sizeof(S.fresh));
    return 0;
}

// Start of a span with constant weights: re-pack every entry from the current weights (one launch on `stream`, none when the
// table is empty) and arm the lookups.  The convolutions of the span must run on `stream` or on streams ordered after it.
// owner: any value that changes whenever the SET OF WEIGHT ADDRESSES the caller is about to use changes (another model, parameters
// moved or re-allocated): entries are keyed by address and the refresh reads every registered address, so entries of another
// owner are dropped first.
int jvae_pack_cache_begin(void* stream, long long owner) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (!S.buf) return 0;
    if (owner != S.owner) {                          // other weights (another model, re-allocated parameters): the entries'
        S.used = 0;                                  // source addresses may be dead - forget them BEFORE the refresh launch
        memset(&S.tab, 0, sizeof(S.tab));
        memset(S.fresh, 0, sizeof(S.fresh));
        S.owner = owner;
    }
    S.armed = true;
    if (S.tab.n > 0) {
        hipLaunchKernelGGL(pack_refresh_kernel, dim3(S.tab.blocks), dim3(256), 0, (hipStream_t)stream, S.tab);
        JVAE_LAUNCH_CHECK();
        ++S.refreshes;
    }
    for (int k = 0; k < S.tab.n; ++k) S.fresh[k] = true;
    return 0;
}

// End of the span (the weights are about to change, or the caller no longer vouches for them): lookups return "not cached"
// until the next begin.  Entries stay registered - the next begin refreshes them.
int jvae_pack_cache_end(void) {
    std::lock_guard<std::mutex> lk(S.mu);
    S.armed = false;
    for (int k = 0; k < S.tab.n; ++k) S.fresh[k] = false;
    return 0;
}

// Forget every entry (parameters were re-allocated: their addresses are the keys).  The buffer stays configured.
int jvae_pack_cache_reset(void) {
    std::lock_guard<std::mutex> lk(S.mu);
    S.used = 0;
    S.armed = false;
    memset(&S.tab, 0, sizeof(S.tab));
    memset(S.fresh, 0, sizeof(S.fresh));
    return 0;
}

// Host-side counters (tests / diagnostics): entries, lookups served from the cache, lookups that packed, refresh launches.
int jvae_pack_cache_stats(int* entries, long long* hits, long long* misses, long long* refreshes) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (entries) *entries = S.tab.n;
    if (hits) *hits = S.hits;
    if (misses) *misses = S.misses;
    if (refreshes) *refreshes = S.refreshes;
    return 0;
}

}  // extern "C"
