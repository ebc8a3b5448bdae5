// Per-step cache of the re-packed convolution weights.
//
// Every forward-type 5x5 kernel reads its weights in a packed operand layout (pack_elems.h): fp32 [c][tap][o], three split
// bf16 planes, or bf16 units.  The packs depend on the weights only, which change once per step (the optimiser), yet each
// convolution call used to launch its own pack kernel in front of itself: 18 launches of ~5 us per training step of
// BASELINE configs[1], all on the critical path (VERDICT r2 "small-launch diet").  With the cache the host brackets the span
// in which the weights are constant - evaluate() ... end of backward, or an evaluation pass:
//
//     jvae_pack_cache_begin(stream, owner, weights, n)   ONE kernel re-packs every known (weight, layout) entry of this
//                                                        owner from the CURRENT weights; `weights` DECLARES the addresses
//                                                        the span may cache
//     ... convolutions                look their packed operand up (host-side table, no launch)
//     jvae_pack_cache_end()           backward is over / the optimiser is about to change the weights / the caller no longer
//                                     vouches for them: lookups fall back to per-call packs
//
// An entry is created the first time a convolution asks for a DECLARED weight inside a bracket (that call packs into the new
// slot itself); from the next begin() of the same owner on it is refreshed with the others.  A weight the owner did not declare
// (another model's forward, a direct op call with a temporary tensor - whose address may be re-used after it is freed) is never
// cached: its convolution packs for itself.  Outside a bracket nothing is cached.  No reference counterpart (PyTorch re-lays
// weights out inside cuDNN / MIOpen).
//
// Owners: every owner (= one set of weight addresses, i.e. one model on one flat buffer) has a REGION of its own - a fixed
// share of the cache memory and its own entry table.  A begin() of owner B never touches the slots or the table of owner A, so
// a HIP graph captured inside A's bracket (the table and the slot pointers are baked into it by value) stays valid while B runs
// eagerly in between; jvae_pack_cache_pin() keeps the armed owner's region from being recycled (graph_train_step calls it).
// With all regions taken by pinned or more recently used owners, the least recently used unpinned one is recycled; if every
// region is pinned the new owner simply runs uncached.
//
// The cache memory is the CALLER's (jvae_pack_cache_configure: a persistent device buffer); the entry tables live on the
// host and travel to the refresh kernel by value (kernel arguments: capture-safe, nothing to upload).
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
        else if (kind == JVAE_PACK_X3S) jvae_pack_x3s_elem(en.w, (__bf16*)en.dst, i, en.C, en.O, swap, flip);
        else if (kind == JVAE_PACK_SCI) jvae_pack_sci_elem(en.w, (float*)en.dst, i, en.C, en.O, swap, flip);
        else if (kind == JVAE_PACK_T2S) jvae_pack_t2s_elem(en.w, (__bf16*)en.dst, i, en.C, en.O, swap, flip);
        else jvae_pack_b8_elem(en.w, (__bf16*)en.dst, i, en.C, en.O, swap, flip);
    }
}

constexpr int MAX_OWNERS = 4;
constexpr int MAX_DECLARED = 64;

struct Region {
    bool used = false, pinned = false;
    long long owner = 0;
    unsigned long long stamp = 0;     // last begin(): least recently used first when a region has to be recycled
    size_t fill = 0;                  // bytes of this region's share handed out
    PackTable tab{};
    bool fresh[MAX_ENTRIES] = {};     // entry holds the pack of the weights of the current bracket
    const void* declared[MAX_DECLARED] = {};
    int ndecl = 0;
};

struct State {
    std::mutex mu;
    unsigned char* buf = nullptr;
    size_t bytes = 0;
    Region reg[MAX_OWNERS];
    int armed = -1;                   // region of the open bracket, -1: none
    unsigned long long clock = 0;
    long long hits = 0, misses = 0, refreshes = 0;
} S;

size_t region_bytes() { return S.bytes / MAX_OWNERS / 256 * 256; }

void drop_all() {
    for (Region& r : S.reg) r = Region{};
    S.armed = -1;
}

}  // namespace

void* jvae_pack_cache_get(int kind, const float* w, int C, int O, int swap, int flip, bool* fresh) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (!S.buf || S.armed < 0) return nullptr;
    Region& R = S.reg[S.armed];
    const int ksf = kind | (swap ? 16 : 0) | (flip ? 32 : 0);
    for (int k = 0; k < R.tab.n; ++k) {
        const PackEntry& en = R.tab.e[k];
        if (en.w == w && en.C == C && en.O == O && en.kind_swap_flip == ksf) {
            *fresh = R.fresh[k];
            R.fresh[k] = true;                       // a stale entry is re-packed by this very call
            *fresh ? ++S.hits : ++S.misses;
            return en.dst;
        }
    }
    bool declared = false;                           // only addresses the bracket's owner vouches for are ever registered
    for (int k = 0; k < R.ndecl && !declared; ++k) declared = R.declared[k] == (const void*)w;
    if (!declared) return nullptr;
    const size_t need = (jvae_pack_bytes(kind, C, O) + 255) / 256 * 256;
    if (R.tab.n >= MAX_ENTRIES || R.fill + need > region_bytes()) return nullptr;
    PackEntry& en = R.tab.e[R.tab.n];
    en = PackEntry{w, S.buf + (size_t)S.armed * region_bytes() + R.fill, C, O, ksf, R.tab.blocks};
    R.fill += need;
    R.tab.blocks += (unsigned)((jvae_pack_elems(kind, C, O) + ELEMS_PER_BLOCK - 1) / ELEMS_PER_BLOCK);
    R.fresh[R.tab.n] = true;
    ++R.tab.n;
    ++S.misses;
    *fresh = false;
    return en.dst;
}

extern "C" {

// buf: persistent device memory (256-byte aligned) for the packed weights, NULL / 0 switches the cache off.  Drops every entry
// of every owner (pinned ones too: graphs captured with the cache must not be replayed afterwards).
int jvae_pack_cache_configure(void* buf, size_t bytes) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (buf && ((uintptr_t)buf & 255)) return JVAE_EINVAL;
    S.buf = (unsigned char*)buf;
    S.bytes = buf ? bytes : 0;
    drop_all();
    return 0;
}

// Start of a span with constant weights: re-pack every entry of `owner` from the current weights (one launch on `stream`,
// none when its table is empty) and arm the lookups.  The convolutions of the span must run on `stream` or on streams ordered
// after it.
// owner: any value that changes whenever the SET OF WEIGHT ADDRESSES the caller is about to use changes (another model,
// parameters moved or re-allocated); weights[0..n): those addresses - the only ones the span may create entries for.
int jvae_pack_cache_begin(void* stream, long long owner, const void* const* weights, int n) {
    std::lock_guard<std::mutex> lk(S.mu);
    S.armed = -1;
    if (!S.buf || n < 0 || (n > 0 && !weights)) return (n < 0 || (n > 0 && !weights)) ? JVAE_EINVAL : 0;
    int r = -1;
    for (int k = 0; k < MAX_OWNERS; ++k)
        if (S.reg[k].used && S.reg[k].owner == owner) r = k;
    if (r < 0) {                                     // a new owner: a free region, else the least recently used unpinned one
        for (int k = 0; k < MAX_OWNERS && r < 0; ++k)
            if (!S.reg[k].used) r = k;
        if (r < 0)
            for (int k = 0; k < MAX_OWNERS; ++k)
                if (!S.reg[k].pinned && (r < 0 || S.reg[k].stamp < S.reg[r].stamp)) r = k;
        if (r < 0) return 0;                         // every region pinned by a captured graph: this owner runs uncached
        S.reg[r] = Region{};                         // the old owner's entries (their source addresses may be dead) are
        S.reg[r].used = true;                        // forgotten BEFORE any refresh launch could read them
        S.reg[r].owner = owner;
    }
    Region& R = S.reg[r];
    R.stamp = ++S.clock;
    R.ndecl = n < MAX_DECLARED ? n : MAX_DECLARED;   // beyond the table: those weights simply stay uncached
    for (int k = 0; k < R.ndecl; ++k) R.declared[k] = weights[k];
    S.armed = r;
    if (R.tab.n > 0) {
        hipLaunchKernelGGL(pack_refresh_kernel, dim3(R.tab.blocks), dim3(256), 0, (hipStream_t)stream, R.tab);
        JVAE_LAUNCH_CHECK();
        ++S.refreshes;
    }
    for (int k = 0; k < R.tab.n; ++k) R.fresh[k] = true;
    return 0;
}

// End of the span (backward is over, the weights are about to change, or the caller no longer vouches for them): lookups
// return "not cached" until the next begin.  Entries stay registered - the owner's next begin refreshes them.
int jvae_pack_cache_end(void) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (S.armed >= 0) {
        Region& R = S.reg[S.armed];
        for (int k = 0; k < R.tab.n; ++k) R.fresh[k] = false;
    }
    S.armed = -1;
    return 0;
}

// The armed owner's region is never recycled for another owner (a HIP graph captured in this bracket has the slot addresses
// and the refresh table baked in).  JVAE_EINVAL outside a bracket.
int jvae_pack_cache_pin(void) {
    std::lock_guard<std::mutex> lk(S.mu);
    if (!S.buf) return 0;
    if (S.armed < 0) return JVAE_EINVAL;
    S.reg[S.armed].pinned = true;
    return 0;
}

// Forget every entry of every owner (parameters were re-allocated: their addresses are the keys).  The buffer stays configured.
int jvae_pack_cache_reset(void) {
    std::lock_guard<std::mutex> lk(S.mu);
    drop_all();
    return 0;
}

// Host-side counters (tests / diagnostics): entries of the owner whose span was opened last, lookups served from the cache, lookups that packed,
// refresh launches.
int jvae_pack_cache_stats(int* entries, long long* hits, long long* misses, long long* refreshes) {
    std::lock_guard<std::mutex> lk(S.mu);
    const Region* last = nullptr;                    // the owner whose span was opened last
    for (const Region& r : S.reg)
        if (r.used && (!last || r.stamp > last->stamp)) last = &r;
    if (entries) *entries = last ? last->tab.n : 0;
    if (hits) *hits = S.hits;
    if (misses) *misses = S.misses;
    if (refreshes) *refreshes = S.refreshes;
    return 0;
}

}  // extern "C"
