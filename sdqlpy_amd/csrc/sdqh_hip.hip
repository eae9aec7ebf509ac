// sdqh_hip.hip — the C ABI (include/sdqh.h) over the gfx950 kernels of sdqh_kernels.hpp.
//
// One ctx = one GPU + one HIP stream.  Device memory comes from a per-ctx caching pool (tables ask
// for worst-case-sized buffers and touch only what the data needs: 288 GB of HBM3E makes virtual
// head-room cheap, and it lets every size decision stay on the device, so a query runs without a
// single host round trip until its result is fetched).  No torch types, no global mutable state.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <map>
#include <vector>

#include "sdqh_host.hpp"

using namespace sdqh_host;

static int ensure_index(sdqh_ctx* ctx, sdqh_table* tb);

namespace sdqh_host {

int fail(sdqh_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}
#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// ---- pool --------------------------------------------------------------------------------------
void* pool_alloc(sdqh_ctx* ctx, size_t bytes) {
    bytes = std::max<size_t>(256, (bytes + 255) & ~(size_t)255);
    int best = -1;
    for (size_t i = 0; i < ctx->pool.size(); ++i) {
        PoolBlock& b = ctx->pool[i];
        if (b.free && b.size >= bytes && b.size <= bytes * 2 + (1u << 20) && (best < 0 || b.size < ctx->pool[(size_t)best].size)) best = (int)i;
    }
    // (recording a plan graph: whatever the recorded calls allocate belongs to the graph — its kernels name those addresses at every
    //  replay — and is released with it, sdqh_graph_free; pool_free leaves such a block alone)
    if (best >= 0) { PoolBlock& b = ctx->pool[(size_t)best]; b.free = false; b.alloc_seq = ctx->launch_seq; b.fill_use = false; b.graph_owner = ctx->capturing ? ctx->capture_tag : nullptr; return b.ptr; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        // release cached free blocks and retry once
        for (auto& b : ctx->pool) if (b.free && b.ptr) { (void)hipFree(b.ptr); b.ptr = nullptr; }
        ctx->pool.erase(std::remove_if(ctx->pool.begin(), ctx->pool.end(), [](const PoolBlock& b) { return b.ptr == nullptr; }), ctx->pool.end());
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    { PoolBlock nb{p, bytes, false}; nb.alloc_seq = ctx->launch_seq; nb.graph_owner = ctx->capturing ? ctx->capture_tag : nullptr; ctx->pool.push_back(nb); }
    return p;
}
void pool_free(sdqh_ctx* ctx, void* p) {
    if (!p) return;
    for (auto& b : ctx->pool) if (b.ptr == p) { if (b.graph_owner) return; b.free = true; for (int h = 0; h < b.nhabits; ++h) b.habits[h].clean = false; if (!b.fill_use) b.nhabits = 0; return; }
}

void* attach_alloc(sdqh_ctx* ctx, const sdqh_column* c, size_t bytes) {
    sdqh_ctx* h = c->home ? c->home : ctx;
    if (h != ctx && h->stream) (void)hipStreamSynchronize(h->stream);
    return pool_alloc(h, bytes);
}
void attach_free(sdqh_ctx* ctx, const sdqh_column* c, void* p) { pool_free(c->home ? c->home : ctx, p); }

// ---- profiling / timing ------------------------------------------------------------------------
hipEvent_t next_event(sdqh_ctx* ctx) {
    if (ctx->event_next == ctx->event_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); ctx->event_pool.push_back(e); }
    return ctx->event_pool[ctx->event_next++];
}
void call_begin(sdqh_ctx* ctx) {
    if (ctx->nested) return;                               // an entry point implemented with others: one call for the profile and the call timer
    if (ctx->profiling != 2) { ctx->prof.clear(); ctx->event_next = 0; }
    ctx->call_timed = false;
    if (ctx->profiling != 1) return;                       // the call timer costs two barrier packets on the stream per call: only in the per-call mode
                                                           // (mode 2 records the launches it is asked for and nothing else: bench.py's timed region — 0.1 ms a step otherwise)
    (void)hipEventRecord(ctx->call_begin, ctx->stream);
}
void call_end(sdqh_ctx* ctx) {
    if (ctx->nested || ctx->profiling != 1) return;
    (void)hipEventRecord(ctx->call_end, ctx->stream);
    ctx->call_timed = true;
}
#define LAUNCH(ctx, name, kernel, grid, ...)                                         \
    do { KernelScope _ks(ctx, name); hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(TPB), 0, (ctx)->stream, __VA_ARGS__); } while (0)
#define LAUNCH_LDS(ctx, name, kernel, grid, lds_bytes, ...)                           \
    do { KernelScope _ks(ctx, name); hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(TPB), (lds_bytes), (ctx)->stream, __VA_ARGS__); } while (0)

int sync_stream(sdqh_ctx* ctx) {
    if (ctx->capturing) return fail(ctx, SDQH_ERR_UNSUPPORTED, "a call that waits for the device cannot be recorded into a plan graph");
    bool done = false;
    if (ctx->opt_spin_sync && !ctx->compile_only) {
        if (!ctx->sync_flag) {
            void* p = nullptr;
            if (hipHostMalloc(&p, 64, hipHostMallocDefault) == hipSuccess) { ctx->sync_flag = static_cast<volatile uint32_t*>(p); *ctx->sync_flag = 0; }
            else { (void)hipGetLastError(); ctx->opt_spin_sync = 0; }
        }
        if (ctx->sync_flag) {
            const uint32_t seq = ++ctx->sync_seq;
            if (hipStreamWriteValue32(ctx->stream, const_cast<uint32_t*>(ctx->sync_flag), seq, 0) == hipSuccess) {
                // poll; after 2 s of nothing (wall clock, read every 4096 spins: a spin COUNT is 40 - 100 s of a pegged core at 60 - 140 cycles
                // per PAUSE) fall back to the runtime's wait, which also reports a faulted kernel
                const auto t0 = std::chrono::steady_clock::now();
                for (uint64_t spins = 1;; ++spins) {
                    if (*ctx->sync_flag == seq) { done = true; break; }
                    __builtin_ia32_pause();
                    if ((spins & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
                }
            } else { (void)hipGetLastError(); ctx->opt_spin_sync = 0; }
        }
    }
    if (!done) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->profiling == 1) for (auto& e : ctx->prof) { float ms = 0; if (hipEventElapsedTime(&ms, e.e0, e.e1) == hipSuccess) e.ms = ms; }
    return SDQH_OK;
}

// One launch that sets up to FILL_MAX regions to a byte value each (see k_fill).
struct FillList {
    DevFillBig f; uint64_t most = 0;
    FillList() { std::memset(&f, 0, sizeof(f)); }
    DevFill pre() const {                                  // the short form a build kernel takes for its own fill (n <= FILL_MAX)
        DevFill d; std::memset(&d, 0, sizeof(d));
        for (int i = 0; i < f.n && i < FILL_MAX; ++i) { d.p[i] = f.p[i]; d.bytes[i] = f.bytes[i]; d.word[i] = f.word[i]; }
        d.n = std::min<int>(f.n, FILL_MAX);
        return d;
    }
    void add(void* p, size_t bytes, unsigned char byte) {
        if (!p || !bytes || f.n >= FILL_BIG) return;
        f.p[f.n] = p; f.bytes[f.n] = bytes; f.word[f.n] = 0x01010101u * byte; ++f.n;
        most = std::max<uint64_t>(most, bytes);
    }
};
// Fill-ahead.  A plan's builds each start with a fill of their table's header / bitmap / reference array: 6 us launches, three to
// five per join query.  The blocks come back from the pool run after run, so (1) a region that starts a pool block which was
// set to the same byte while it lay FREE, and that no kernel can have touched since it was handed out (no launch since), is
// dropped from the list; (2) a launch that has to happen anyway also sets the free blocks that were last used as fill regions
// ("habit"), so the later fills of the plan find theirs clean.  Results cannot depend on it: a region is skipped only when it
// provably holds the byte (`fill_ahead` 0 switches it off).
static PoolBlock* pool_block_of(sdqh_ctx* ctx, const void* p) {
    const char* c = static_cast<const char*>(p);
    for (auto& b : ctx->pool) if (c >= static_cast<const char*>(b.ptr) && c < static_cast<const char*>(b.ptr) + b.size) return &b;
    return nullptr;
}
static void prune_clean(sdqh_ctx* ctx, FillList* fl) {
    if (!ctx->opt_fill_ahead || ctx->capturing) return;      // (a recorded fill must fill at every replay: "clean" is a fact about THIS run)
    DevFillBig out; std::memset(&out, 0, sizeof(out));
    uint64_t most = 0;
    for (int i = 0; i < fl->f.n; ++i) {
        PoolBlock* b = pool_block_of(ctx, fl->f.p[i]);
        const int byte = (int)(fl->f.word[i] & 0xFFu);
        if (b && !b->free) {
            const size_t off = (size_t)(static_cast<const char*>(fl->f.p[i]) - static_cast<const char*>(b->ptr)), bytes = fl->f.bytes[i];
            // a habit of the same byte that covers the region serves it (blocks change hands between plans: the customers' bitmap of
            // one query is the parts' of the next, a little shorter or longer)
            int at = -1;
            for (int h = 0; h < b->nhabits; ++h) if (b->habits[h].byte == byte && b->habits[h].off <= off && b->habits[h].off + b->habits[h].bytes >= off + bytes) at = h;
            b->fill_use = true;
            if (at >= 0 && b->habits[at].clean && b->alloc_seq == ctx->launch_seq) { b->habits[at].clean = false; continue; }   // holds the byte already; its owner writes it next
            if (at < 0) {                                                   // a new habit: overlapping ones of the same byte grow into it, others are stale
                size_t lo = off, hi = off + bytes;
                int keep = 0;
                for (int h = 0; h < b->nhabits; ++h) {
                    const FillHabit& o = b->habits[h];
                    if (o.off + o.bytes <= off || off + bytes <= o.off) { b->habits[keep++] = o; continue; }
                    if (o.byte == byte) { lo = std::min(lo, o.off); hi = std::max(hi, o.off + o.bytes); }
                }
                b->nhabits = keep;
                if (b->nhabits < 4) b->habits[b->nhabits++] = FillHabit{lo, hi - lo, byte, false};
            } else b->habits[at].clean = false;
        }
        out.p[out.n] = fl->f.p[i]; out.bytes[out.n] = fl->f.bytes[i]; out.word[out.n] = fl->f.word[i]; ++out.n;
        most = std::max<uint64_t>(most, fl->f.bytes[i]);
    }
    fl->f = out; fl->most = most;
}
static void launch_fill(sdqh_ctx* ctx, const FillList& fl_in) {
    FillList fl = fl_in;
    prune_clean(ctx, &fl);
    if (!fl.f.n) return;
    if (ctx->opt_fill_ahead && !ctx->capturing) {              // (never record a fill of somebody else's free block into a graph)
        for (auto& b : ctx->pool) {
            if (!b.free) continue;
            for (int h = 0; h < b.nhabits && fl.f.n < FILL_BIG; ++h) {
                FillHabit& hb = b.habits[h];
                if (hb.clean || hb.off + hb.bytes > b.size || hb.bytes > ((size_t)64 << 20)) continue;
                fl.add(static_cast<char*>(b.ptr) + hb.off, hb.bytes, (unsigned char)hb.byte);
                hb.clean = true;
            }
        }
    }
    const uint64_t want = (fl.most / 16 + TPB - 1) / TPB;
    const unsigned grid = (unsigned)std::min<uint64_t>(std::max<uint64_t>(want, 1), (uint64_t)ctx->num_cu * 8);
    LAUNCH(ctx, "k_fill", k_fill, grid, fl.f);
}

// ---- argument conversion -------------------------------------------------------------------------
int tuple_nops(int shape) {
    switch (shape) {
        case SDQH_TUPLE_A: return 1; case SDQH_TUPLE_AB: return 2; case SDQH_TUPLE_A_1MB: return 2;
        case SDQH_TUPLE_PRICING: return 4; case SDQH_TUPLE_A_1MB_M_CD: return 4; case SDQH_TUPLE_COUNT: return 0;
        default: return -1;
    }
}
int tuple_nv(int shape) {
    switch (shape) {
        case SDQH_TUPLE_PRICING: return 4; case SDQH_TUPLE_COUNT: return 0;
        case SDQH_TUPLE_A: case SDQH_TUPLE_AB: case SDQH_TUPLE_A_1MB: case SDQH_TUPLE_A_1MB_M_CD: return 1;
        default: return -1;
    }
}

int check_col(sdqh_ctx* ctx, const sdqh_column* c, int dtype, int64_t nrows, const char* what) {
    if (!c) return fail(ctx, SDQH_ERR_INVALID, std::string(what) + ": null column");
    if (c->dtype != dtype) return fail(ctx, SDQH_ERR_INVALID, std::string(what) + ": wrong dtype");
    if (c->nrows < nrows) return fail(ctx, SDQH_ERR_INVALID, std::string(what) + ": column shorter than nrows");
    return SDQH_OK;
}

int make_tuple(sdqh_ctx* ctx, int64_t nrows, const sdqh_tuple* t, DevTuple* out) {
    if (!t) return fail(ctx, SDQH_ERR_INVALID, "null tuple");
    int nops = tuple_nops(t->shape);
    if (nops < 0) return fail(ctx, SDQH_ERR_UNSUPPORTED, "unknown tuple shape");
    const sdqh_column* ops[4] = {t->a, t->b, t->c, t->d};
    for (int j = 0; j < 4; ++j) out->op[j] = nullptr;
    for (int j = 0; j < nops; ++j) {
        if (int rc = check_col(ctx, ops[j], SDQH_F64, nrows, "tuple operand")) return rc;
        out->op[j] = static_cast<const double*>(ops[j]->data);
    }
    return SDQH_OK;
}

// f-predicates on a column that is also a value operand become ranges on the operand slot.
int make_filter(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* f, const sdqh_tuple* t, DevFilter* d) {
    std::memset(d, 0, sizeof(*d));
    if (!f) return SDQH_OK;
    if (f->n_ipred < 0 || f->n_ipred > SDQH_MAX_IPRED || f->n_fpred < 0 || f->n_fpred > SDQH_MAX_FPRED || f->n_spred < 0 || f->n_spred > SDQH_MAX_SPRED)
        return fail(ctx, SDQH_ERR_INVALID, "filter: predicate count out of range");
    for (int i = 0; i < f->n_ipred; ++i) {
        if (int rc = check_col(ctx, f->ipred[i].col, SDQH_I64, nrows, "ipred")) return rc;
        d->ic[d->ni] = static_cast<const int64_t*>(f->ipred[i].col->data); d->ilo[d->ni] = f->ipred[i].lo; d->ihi[d->ni] = f->ipred[i].hi; d->ni++;
    }
    const int nops = t ? tuple_nops(t->shape) : 0;
    const sdqh_column* ops[4] = {t ? t->a : nullptr, t ? t->b : nullptr, t ? t->c : nullptr, t ? t->d : nullptr};
    for (int i = 0; i < f->n_fpred; ++i) {
        if (int rc = check_col(ctx, f->fpred[i].col, SDQH_F64, nrows, "fpred")) return rc;
        int alias = -1;
        for (int j = 0; j < nops; ++j) if (ops[j] && ops[j]->data == f->fpred[i].col->data) { alias = j; break; }
        if (alias >= 0) {
            if ((d->omask >> alias) & 1u) { d->olo[alias] = std::max(d->olo[alias], f->fpred[i].lo); d->ohi[alias] = std::min(d->ohi[alias], f->fpred[i].hi); }
            else { d->omask |= 1u << alias; d->olo[alias] = f->fpred[i].lo; d->ohi[alias] = f->fpred[i].hi; }
        } else {
            d->fc[d->nf] = static_cast<const double*>(f->fpred[i].col->data); d->flo[d->nf] = f->fpred[i].lo; d->fhi[d->nf] = f->fpred[i].hi; d->nf++;
        }
    }
    if (f->n_cpred < 0 || f->n_cpred > SDQH_MAX_CPRED) return fail(ctx, SDQH_ERR_INVALID, "filter: too many column comparisons");
    d->nc = f->n_cpred;
    for (int i = 0; i < f->n_cpred; ++i) {
        const sdqh_cpred& c = f->cpred[i];
        if (!c.a || !c.b || c.a->dtype != c.b->dtype || c.a->dtype == SDQH_STR || c.a->nrows < nrows || c.b->nrows < nrows || c.op < SDQH_CMP_LT || c.op > SDQH_CMP_NE)
            return fail(ctx, SDQH_ERR_INVALID, "cpred: two I64 or two F64 columns covering nrows, op LT/LE/EQ/NE");
        d->ca[i] = static_cast<const int64_t*>(c.a->data); d->cb[i] = static_cast<const int64_t*>(c.b->data);
        d->cop[i] = c.op; d->cf64[i] = c.a->dtype == SDQH_F64 ? 1 : 0;
    }
    if (f->n_spred == 1) {
        if (int rc = check_col(ctx, f->spred[0].col, SDQH_STR, nrows, "spred")) return rc;
        if (f->spred[0].len < 0 || f->spred[0].len > SDQH_MAX_STR_CONST) return fail(ctx, SDQH_ERR_INVALID, "spred: constant too long");
        d->ns = 1; d->sc = static_cast<const uint32_t*>(f->spred[0].col->data); d->swidth = f->spred[0].col->width;
        d->slen = f->spred[0].len; d->sneg = f->spred[0].negate;
        std::memcpy(d->sval, f->spred[0].value, sizeof(uint32_t) * SDQH_MAX_STR_CONST);
    }
    return SDQH_OK;
}

int make_probes(sdqh_ctx* ctx, int64_t nrows, int nprobes, const sdqh_probe* probes, DevProbes* d) {
    std::memset(d, 0, sizeof(*d));
    if (nprobes < 0 || nprobes > SDQH_MAX_PROBE) return fail(ctx, SDQH_ERR_INVALID, "too many probes");
    for (int i = 0; i < nprobes; ++i) {
        if (!probes[i].table) return fail(ctx, SDQH_ERR_INVALID, "probe: null table");
        if (int rc = check_col(ctx, probes[i].key, SDQH_I64, nrows, "probe key")) return rc;
        sdqh_table* pt = const_cast<sdqh_table*>(probes[i].table);
        if (!pt->bm || pt->dev.bm_shift) { if (int rc = ensure_index(ctx, pt)) return rc; }   // hash layout: contains() walks the slots
        d->table[i] = pt->dev; d->key[i] = static_cast<const int64_t*>(probes[i].key->data);
    }
    d->n = nprobes;
    return SDQH_OK;
}

// Workgroups of `kernel` that are resident per CU.  Streaming kernels are launched with exactly
// num_cu * resident workgroups (a persistent grid striding over the tiles): a grid one wave of
// workgroups larger than what fits leaves a tail in which most of the chip idles.
// The occupancy API over-reports by one for SGPR-heavy 256-thread kernels on gfx950 / ROCm 7.2
// (MI355X_MICROARCH.md, residency); these kernels all sit in that band, hence the cap of 6.
template <class K>
int resident_per_cu(sdqh_ctx* ctx, K kernel) {
    const void* key = reinterpret_cast<const void*>(kernel);
    for (auto& e : ctx->occupancy) if (e.first == key) return std::min(e.second, ctx->opt_resident_cap);
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, TPB, 0) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 2; }
    ctx->occupancy.push_back({key, n});
    return std::min(n, ctx->opt_resident_cap);
}
template <class K>
unsigned stream_grid(sdqh_ctx* ctx, K kernel, int64_t nrows, int tile_rows = TILE_ROWS, bool pure_stream = false, int stream_resident = 0) {
    int64_t tiles = (nrows + tile_rows - 1) / tile_rows;
    if (pure_stream) tiles = (tiles + SDQH_TILE_CHUNK - 1) / SDQH_TILE_CHUNK;
    const int resident_stream = stream_resident > 0 ? std::max(stream_resident, ctx->opt_resident_stream) : ctx->opt_resident_stream;
    int64_t cap = (int64_t)ctx->num_cu * (pure_stream ? std::min(resident_per_cu(ctx, kernel), resident_stream) : resident_per_cu(ctx, kernel));
    return (unsigned)std::max<int64_t>(1, std::min(tiles, cap));
}

int ensure_minmax(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->have_minmax) return SDQH_OK;
    if (c->dtype != SDQH_I64) return fail(ctx, SDQH_ERR_INVALID, "minmax: needs an I64 column");
    // (every fact about a column that is measured on first request waits for the device.  Inside a recording that is not a refusal the
    //  runtime takes back — a wait on a capturing stream invalidates the capture and the stream stays lost, tools/exp/capture_abort.hip —
    //  so the measuring routines below give the answer that needs no measurement, uncached, and this one refuses BEFORE any call)
    if (ctx->capturing) return fail(ctx, SDQH_ERR_UNSUPPORTED, "minmax: a column's bounds would have to be measured inside a recording");
    if (!c->minmax_pending) {
        if (!c->d_minmax) { c->d_minmax = static_cast<long long*>(attach_alloc(ctx, c, 16)); if (!c->d_minmax) return fail(ctx, SDQH_ERR_NOMEM, "minmax: out of device memory"); }
        const long long init[2] = {INT64_MAX, INT64_MIN};
        HIP_TRY(ctx, hipMemcpyAsync(c->d_minmax, init, 16, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // `init` is on the stack
        if (c->nrows > 0) LAUNCH(ctx, "k_minmax", k_minmax, (unsigned)std::min<int64_t>((c->nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 4), static_cast<const int64_t*>(c->data), c->nrows, c->d_minmax);
        c->minmax_pending = true;
    }
    long long host[2];
    HIP_TRY(ctx, hipMemcpyAsync(host, c->d_minmax, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    c->mn = host[0]; c->mx = host[1]; c->have_minmax = true; c->minmax_pending = false;
    return SDQH_OK;
}

// Do neighbouring rows of an I64 column hold near-by values (a fact table clustered on this key)?  Sampled once
// per column on the host side of a tiny download: 2048 adjacent pairs spread over the column; "near" = within 4096.
bool column_is_clustered(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->clustered >= 0) return c->clustered == 1;
    if (c->dtype != SDQH_I64 || c->nrows < 4096) { c->clustered = 1; return true; }
    if (ctx->capturing) return true;                                      // (not measured inside a recording: the answer a failed download gives, uncached)
    const int samples = 2048;
    const int64_t step = (c->nrows - 2) / samples;
    int64_t* host = static_cast<int64_t*>(ctx->result_host);              // 64 KiB pinned: 2048 pairs = 32 KiB
    bool ok = true;
    for (int i = 0; i < samples && ok; ++i)
        ok = hipMemcpyAsync(host + 2 * i, static_cast<const int64_t*>(c->data) + (int64_t)i * step, 16, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    if (!ok || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); return true; }
    int near = 0;
    for (int i = 0; i < samples; ++i) { const int64_t d = host[2 * i + 1] - host[2 * i]; near += (d >= -4096 && d <= 4096) ? 1 : 0; }
    c->clustered = near * 10 >= samples * 9 ? 1 : 0;
    return c->clustered == 1;
}

// Is an I64 column strictly increasing?  One pass over the column the first time it is asked, cached like min / max.
bool column_is_increasing(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->increasing >= 0) return c->increasing == 1;
    if (c->dtype != SDQH_I64) { c->increasing = 0; return false; }
    if (c->nrows < 2) { c->increasing = 1; return true; }
    if (ctx->capturing) return false;                                     // (not measured inside a recording; uncached)
    int* flag = static_cast<int*>(pool_alloc(ctx, 64));
    if (!flag) return false;
    bool ok = hipMemsetAsync(flag, 0, 4, ctx->stream) == hipSuccess;
    if (ok) {
        const unsigned grid = (unsigned)std::min<int64_t>((c->nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 8);
        hipLaunchKernelGGL(k_check_increasing, dim3(grid), dim3(TPB), 0, ctx->stream, static_cast<const int64_t*>(c->data), c->nrows, flag);
        int* host = static_cast<int*>(ctx->result_host);
        ok = hipMemcpyAsync(host, flag, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (ok) c->increasing = host[0] == 0 ? 1 : 0;
    }
    if (!ok) (void)hipGetLastError();
    pool_free(ctx, flag);
    return ok && c->increasing == 1;
}

bool column_is_nondecreasing(sdqh_ctx* ctx, sdqh_column* c);
// ... or, as PAIRS with another column of the same table, no pair twice — strictly increasing row after row, or the first part never
// decreasing and the second parts of one run distinct (cached on `a`, one partner)?
bool columns_pair_increasing(sdqh_ctx* ctx, sdqh_column* a, sdqh_column* b) {
    if (a->dtype != SDQH_I64 || b->dtype != SDQH_I64 || a->nrows != b->nrows || a->transient || b->transient) return false;
    if (a->increasing == 1) return true;
    if (a->pair_uid == b->uid && a->pair_increasing >= 0) return a->pair_increasing == 1;
    if (ctx->capturing) return false;                                     // (the check waits for the device: inside a recording only what is known already counts)
    if (a->nondecreasing == 0) { a->pair_uid = b->uid; a->pair_increasing = 0; return false; }
    if (a->nrows < 2) return true;
    int* flag = static_cast<int*>(pool_alloc(ctx, 64));
    if (!flag) return false;
    bool ok = hipMemsetAsync(flag, 0, 4, ctx->stream) == hipSuccess, yes = false;
    if (ok) {
        const unsigned grid = (unsigned)std::min<int64_t>((a->nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 8);
        hipLaunchKernelGGL(k_check_pair_increasing, dim3(grid), dim3(TPB), 0, ctx->stream, static_cast<const int64_t*>(a->data), static_cast<const int64_t*>(b->data), a->nrows, flag);
        int* host = static_cast<int*>(ctx->result_host);
        ok = hipMemcpyAsync(host, flag, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (ok) yes = host[0] == 0;
        if (ok && !yes && column_is_nondecreasing(ctx, a)) {
            // not sorted as pairs, but stored in the order of the first part: distinct within every run of equal first parts will do
            // (partsupp: the four suppliers of a part in the generator's order)
            ok = hipMemsetAsync(flag, 0, 4, ctx->stream) == hipSuccess;
            if (ok) {
                hipLaunchKernelGGL(k_check_pair_distinct, dim3(grid), dim3(TPB), 0, ctx->stream, static_cast<const int64_t*>(a->data), static_cast<const int64_t*>(b->data), a->nrows, flag);
                ok = hipMemcpyAsync(host, flag, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
                if (ok) yes = host[0] == 0;
            }
        }
        if (ok) { a->pair_uid = b->uid; a->pair_increasing = yes ? 1 : 0; }
    }
    if (!ok) (void)hipGetLastError();
    pool_free(ctx, flag);
    return ok && yes;
}

// ... or never decreasing?  (Cached the same way; a strictly increasing column is.)
bool column_is_nondecreasing(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->nondecreasing >= 0) return c->nondecreasing == 1;
    if (c->increasing == 1) { c->nondecreasing = 1; return true; }
    if (c->dtype != SDQH_I64) { c->nondecreasing = 0; return false; }
    if (c->nrows < 2) { c->nondecreasing = 1; return true; }
    if (ctx->capturing) return false;                                     // (not measured inside a recording; uncached)
    int* flag = static_cast<int*>(pool_alloc(ctx, 64));
    if (!flag) return false;
    bool ok = hipMemsetAsync(flag, 0, 4, ctx->stream) == hipSuccess;
    if (ok) {
        const unsigned grid = (unsigned)std::min<int64_t>((c->nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 8);
        hipLaunchKernelGGL(k_check_nondecreasing, dim3(grid), dim3(TPB), 0, ctx->stream, static_cast<const int64_t*>(c->data), c->nrows, flag);
        int* host = static_cast<int*>(ctx->result_host);
        ok = hipMemcpyAsync(host, flag, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
        if (ok) c->nondecreasing = host[0] == 0 ? 1 : 0;
    }
    if (!ok) (void)hipGetLastError();
    pool_free(ctx, flag);
    return ok && c->nondecreasing == 1;
}

// The 4-byte twin of a streamed column, built and verified on first request (a pass over the column, like min / max).
// Returns the twin or nullptr (the column does not narrow exactly, or no memory: the caller uses the column itself).
const void* ensure_narrow(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->narrow_state >= 0) return c->narrow;
    if (ctx->capturing) return nullptr;                                   // (built and verified with a wait: not inside a recording; uncached)
    c->narrow_state = 0;
    if (c->transient) return nullptr;                                     // (a column that lives for one run — an exchange's unpacked chunk —: a pass and a wait per run would buy one use)
    if (c->nrows < 2) return nullptr;
    const int64_t units = c->dtype == SDQH_STR ? c->nrows * c->width : c->nrows;         // text: one byte per code unit
    int32_t* twin = static_cast<int32_t*>(attach_alloc(ctx, c, c->dtype == SDQH_STR ? (size_t)units + 64 : (size_t)c->nrows * 4 + 64));
    int* flag = static_cast<int*>(pool_alloc(ctx, 64));
    bool ok = twin && flag && hipMemsetAsync(flag, 0, 4, ctx->stream) == hipSuccess;
    if (ok) {
        const unsigned grid = (unsigned)std::min<int64_t>((units + TPB - 1) / TPB, (int64_t)ctx->num_cu * 8);
        if (c->dtype == SDQH_STR) hipLaunchKernelGGL(k_narrow_str1, dim3(grid), dim3(TPB), 0, ctx->stream, static_cast<const uint32_t*>(c->data), units, reinterpret_cast<uint8_t*>(twin), flag);
        else if (c->dtype == SDQH_I64) hipLaunchKernelGGL(k_narrow_i64, dim3(grid), dim3(TPB), 0, ctx->stream, static_cast<const int64_t*>(c->data), c->nrows, twin, flag);
        else hipLaunchKernelGGL(k_narrow_f64, dim3(grid), dim3(TPB), 0, ctx->stream, static_cast<const double*>(c->data), c->nrows, twin, flag);
        int* host = static_cast<int*>(ctx->result_host);
        ok = hipMemcpyAsync(host, flag, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess && host[0] == 0;
    }
    if (!ok) (void)hipGetLastError();
    if (flag) pool_free(ctx, flag);
    if (ok) { c->narrow = twin; c->narrow_state = 1; }
    else if (twin) attach_free(ctx, c, twin);
    return c->narrow;
}
// The LDS-staged string predicate through the text column's byte twin: 64 rows per wave and round (any width: 64 rows of bytes
// start on a 16-byte boundary), a quarter of the bytes streamed.  Returns the words of dynamic LDS per wave, 0 when there is no twin.
static bool stage_text_twin(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, DevFilter* f) {
    if (!ctx->opt_narrow || !f->ns || nrows < ctx->opt_feature_min_rows || f->swidth < 1 || f->swidth > 128) return false;
    const void* twin = ensure_narrow(ctx, const_cast<sdqh_column*>(filter->spred[0].col));
    if (!twin) return false;
    f->sc8 = static_cast<const uint8_t*>(twin); f->slds = 64;
    return true;
}
// Membership build on a key column in no row order: k_key_set_lds + k_or_slices instead of one device-scope atomic per row.
// false: not applicable (the caller launches k_key_set).
constexpr uint32_t KSL_PASS_WORDS = 32768;             // 128 KiB of LDS: one million keys of the range per pass over the rows
template <class FCT>
static bool launch_key_set_lds(sdqh_ctx* ctx, const DevFilter& f, const DevProbes& pr, const sdqh_column* key, int64_t nrows, int64_t lo, int64_t hi, sdqh_table* tb) {
    if (!ctx->opt_lds_key_set || f.ns || nrows < std::max<int64_t>(ctx->opt_feature_min_rows, 4096) || tb->nwords > (uint64_t)4 * KSL_PASS_WORDS) return false;
    if (column_is_clustered(ctx, const_cast<sdqh_column*>(key))) return false;
    auto kern = k_key_set_lds<FCT>;
    const uint32_t pass_words = (uint32_t)std::min<uint64_t>(tb->nwords, KSL_PASS_WORDS);
    const size_t lds = (size_t)pass_words * 4;
    int per_cu = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, KSL_BT, lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); return false; }
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + KSL_BT * KSL_U - 1) / (KSL_BT * KSL_U), (int64_t)ctx->num_cu * std::min(per_cu, 2)));
    uint32_t* slices = static_cast<uint32_t*>(pool_alloc(ctx, (size_t)grid * tb->nwords * 4 + 64));
    if (!slices) return false;
    { KernelScope ks(ctx, "k_key_set_lds"); hipLaunchKernelGGL(kern, dim3(grid), dim3(KSL_BT), lds, ctx->stream, f, pr, static_cast<const int64_t*>(key->data), nrows, lo, hi, slices, tb->nwords, pass_words); }
    const int nchunks = (int)std::max<uint64_t>(1, std::min<uint64_t>(grid / 8, 65536 / std::max<uint64_t>(tb->nwords, 1)));
    LAUNCH(ctx, "k_or_slices", k_or_slices, (unsigned)std::min<uint64_t>((tb->nwords * nchunks + TPB - 1) / TPB, (uint64_t)ctx->num_cu * 8), slices, (int)grid, tb->nwords, tb->bm, nchunks);
    pool_free(ctx, slices);                             // stream order: whoever gets the block next runs after these two
    return true;
}
// Swap every streamed column of a scan (integer / double predicates, tuple operands) for its twin; false (nothing changed)
// unless ALL of them have one.  Only for the instances that read nothing else by row (no string / column-pair predicates).
static bool narrow_streams(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, const sdqh_tuple* tuple, DevFilter* f, DevTuple* t) {
    if (!ctx->opt_narrow || nrows < ctx->opt_feature_min_rows || f->ns || f->nc) return false;
    const int nops = tuple ? std::max(0, tuple_nops(tuple->shape)) : 0;
    const sdqh_column* ops[4] = {tuple ? tuple->a : nullptr, tuple ? tuple->b : nullptr, tuple ? tuple->c : nullptr, tuple ? tuple->d : nullptr};
    const void* ni[SDQH_MAX_IPRED]; const void* nf[SDQH_MAX_FPRED]; const void* no[4];
    int fi = 0;
    for (int i = 0; i < f->ni; ++i) { ni[i] = ensure_narrow(ctx, const_cast<sdqh_column*>(filter->ipred[i].col)); if (!ni[i]) return false; }
    for (int i = 0; filter && i < filter->n_fpred; ++i) {                  // make_filter keeps a double predicate as a streamed column unless it aliases an operand
        bool alias = false;
        for (int j = 0; j < nops; ++j) alias = alias || (ops[j] && ops[j]->data == filter->fpred[i].col->data);
        if (alias) continue;
        if (fi >= f->nf) return false;
        nf[fi] = ensure_narrow(ctx, const_cast<sdqh_column*>(filter->fpred[i].col)); if (!nf[fi]) return false;
        ++fi;
    }
    if (fi != f->nf) return false;
    for (int j = 0; j < nops; ++j) { no[j] = ensure_narrow(ctx, const_cast<sdqh_column*>(ops[j])); if (!no[j]) return false; }
    for (int i = 0; i < f->ni; ++i) f->ic[i] = static_cast<const int64_t*>(ni[i]);
    for (int i = 0; i < f->nf; ++i) f->fc[i] = static_cast<const double*>(nf[i]);
    for (int j = 0; j < nops; ++j) t->op[j] = static_cast<const double*>(no[j]);
    return true;
}

// ---- dispatch onto the compiled menu of kernel instances -------------------------------------------
template <int S> using ShapeC = std::integral_constant<int, S>;
template <class Fn>
int with_shape(sdqh_ctx* ctx, int shape, Fn&& fn) {
    switch (shape) {
        case SDQH_TUPLE_A: return fn(ShapeC<SDQH_TUPLE_A>{});
        case SDQH_TUPLE_AB: return fn(ShapeC<SDQH_TUPLE_AB>{});
        case SDQH_TUPLE_A_1MB: return fn(ShapeC<SDQH_TUPLE_A_1MB>{});
        case SDQH_TUPLE_PRICING: return fn(ShapeC<SDQH_TUPLE_PRICING>{});
        case SDQH_TUPLE_A_1MB_M_CD: return fn(ShapeC<SDQH_TUPLE_A_1MB_M_CD>{});
        case SDQH_TUPLE_COUNT: return fn(ShapeC<SDQH_TUPLE_COUNT>{});
        default: return fail(ctx, SDQH_ERR_UNSUPPORTED, "unknown tuple shape");
    }
}
// filter layouts with their own instances; everything else runs on the generic instance
template <class Fn>
int with_scan_filter(const DevFilter& f, Fn&& fn) {          // K-A / K-C small / K-C large (no probes)
    if (f.nc) return fn(FGeneric{});
    if (f.ns == 0 && f.nf == 0 && f.ni == 1) return fn(FCfg<1, 0, 0, 0>{});
    if (f.ns == 0 && f.nf == 0 && f.ni == 0) return fn(FCfg<0, 0, 0, 0>{});
    if (f.ns == 0 && f.nf == 1 && f.ni == 1) return fn(FCfg<1, 1, 0, 0>{});
    return fn(FGeneric{});
}
template <class Fn>
int with_stage_filter(const DevFilter& f, int nprobes, Fn&& fn) {      // K-B staging
    if (f.nc == 1 && f.ns == 0 && f.nf == 0 && f.ni == 0 && nprobes == 0) return fn(FCfg<0, 0, 0, 0, 1>{});   // Q4: one column comparison
    if (f.nc) return fn(FGeneric{});
    if (f.ns == 0 && f.nf == 0 && f.ni == 1 && nprobes == 1) return fn(FCfg<1, 0, 0, 1>{});
    if (f.ns == 0 && f.nf == 0 && f.ni == 1 && nprobes == 0) return fn(FCfg<1, 0, 0, 0>{});
    if (f.ns == 1 && f.nf == 0 && f.ni == 0 && nprobes == 0) return fn(FCfg<0, 0, 1, 0>{});
    if (f.ns == 0 && f.nf == 0 && f.ni == 0 && nprobes == 0) return fn(FCfg<0, 0, 0, 0>{});
    if (f.ns == 0 && f.nf == 0 && f.ni == 0 && nprobes == 1) return fn(FCfg<0, 0, 0, 1>{});
    return fn(FGeneric{});
}
template <class Fn>
int with_group_keys(const DevGroupKeys& gk, Fn&& fn) {
    if (gk.nkeys == 2 && gk.is_str[0] && gk.is_str[1]) return fn(KCfg<1, 1>{});
    return fn(KGeneric{});
}
template <class Fn>
int with_groupby_filter(const DevFilter& f, Fn&& fn) {          // the register group-by kernels: one tuned layout + generic
    if (f.nc) return fn(FGeneric{});
    if (f.ns == 0 && f.nf == 0 && f.ni == 1) return fn(FCfg<1, 0, 0, 0>{});
    return fn(FGeneric{});
}

}  // namespace sdqh_host

// =================================================================================================
extern "C" {

int sdqh_abi_version(void) { return SDQH_ABI_VERSION; }
const char* sdqh_backend_name(void) { return "hip-gfx950"; }

int sdqh_create(int device, sdqh_ctx** out) {
    if (!out) return SDQH_ERR_INVALID;
    if (device == -1) {                                  // compile-only context (see sdqh.h): no HIP call is ever made through it
        sdqh_ctx* c = new sdqh_ctx();
        c->device = -1; c->compile_only = true;
        *out = c;
        return SDQH_OK;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SDQH_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return SDQH_ERR_DEVICE;
    sdqh_ctx* ctx = new sdqh_ctx();
    ctx->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->num_cu = prop.multiProcessorCount;
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipEventCreate(&ctx->call_begin) == hipSuccess && hipEventCreate(&ctx->call_end) == hipSuccess;
    for (int i = 0; i < 2 && ok; ++i) {
        ok = hipHostMalloc(&ctx->staging[i], STAGING_BYTES, hipHostMallocDefault) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&ctx->staging_done[i], hipEventDisableTiming) == hipSuccess;
    }
    ok = ok && hipHostMalloc(&ctx->result_host, RESULT_BYTES, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc(&ctx->result_dev, RESULT_BYTES) == hipSuccess;
    if (!ok) { sdqh_destroy(ctx); return SDQH_ERR_DEVICE; }
    // The runtime loads this library's code object (a few hundred kernels, 14 MB) at the FIRST launch of one of them: 35-40 ms inside
    // whatever call happens to be first — the first query's first kernel, behind its uploads (34 of Q1's first 135 ms at SF=10).  An
    // empty fill here moves that to the creation of the process's first context, where nothing waits for it.
    static std::atomic<bool> warmed{false};
    if (!warmed.exchange(true)) {
        DevFillBig none; std::memset(&none, 0, sizeof(none));
        hipLaunchKernelGGL(k_fill, dim3(1), dim3(TPB), 0, ctx->stream, none);
        (void)hipGetLastError();
    }
    *out = ctx;
    return SDQH_OK;
}

int sdqh_fork(sdqh_ctx* parent, sdqh_ctx** out) {
    if (!parent || !out) return SDQH_ERR_INVALID;
    if (parent->parent) return fail(parent, SDQH_ERR_INVALID, "fork: fork the family's first context");
    if (int rc = sdqh_create(parent->device, out)) return fail(parent, rc, "fork: no second context on this device");
    (*out)->parent = parent; (*out)->threads = parent->threads;
    parent->children.push_back(*out);
    return SDQH_OK;
}

void sdqh_destroy(sdqh_ctx* ctx) {
    if (!ctx) return;
    // A family's first context owns what its forks share: the columns' home pool (sdqh_column::home), their twins and dictionaries.
    // Destroyed while forks are alive it only marks itself: the last fork to go releases it (a fork that outlives its parent's
    // memory would free attachments into a dead pool).
    if (!ctx->children.empty()) { ctx->dying = true; return; }
    sdqh_ctx* orphaned_parent = nullptr;
    if (ctx->parent) {
        auto& ch = ctx->parent->children; ch.erase(std::remove(ch.begin(), ch.end(), ctx), ch.end());
        if (ctx->parent->dying && ch.empty()) orphaned_parent = ctx->parent;
    }
    struct Finish { sdqh_ctx* p; ~Finish() { if (p) sdqh_destroy(p); } } finish{orphaned_parent};
    if (ctx->compile_only) { delete ctx; return; }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (int i = 0; i < 2; ++i) {
        if (ctx->side[i]) { (void)hipStreamSynchronize(ctx->side[i]); (void)hipStreamDestroy(ctx->side[i]); }
        if (ctx->rs_copied[i]) (void)hipEventDestroy(ctx->rs_copied[i]);
        if (i == 0 && ctx->rs_ready) (void)hipEventDestroy(ctx->rs_ready);
        if (ctx->rs_dev[i]) (void)hipFree(ctx->rs_dev[i]);
    }
    for (auto& b : ctx->pool) if (b.ptr) (void)hipFree(b.ptr);
    for (int i = 0; i < 2; ++i) { if (ctx->staging[i]) (void)hipHostFree(ctx->staging[i]); if (ctx->staging_done[i]) (void)hipEventDestroy(ctx->staging_done[i]); }
    if (ctx->result_host) (void)hipHostFree(ctx->result_host);
    if (ctx->bulk_host) (void)hipHostFree(ctx->bulk_host);
    if (ctx->count_host) (void)hipHostFree(ctx->count_host);
    if (ctx->coarse_stat) (void)hipHostFree(ctx->coarse_stat);
    if (ctx->sync_flag) (void)hipHostFree(const_cast<uint32_t*>(ctx->sync_flag));
    if (ctx->result_dev) (void)hipFree(ctx->result_dev);
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->call_begin) (void)hipEventDestroy(ctx->call_begin);
    if (ctx->call_end) (void)hipEventDestroy(ctx->call_end);
    // a stream that was handed out (sdqh_stream: torch wraps it as an external stream and may still hold tensors allocated on it, events
    // recorded on it) is left to the process: destroying it under its other user crashes that user, leaking one stream harms nobody
    if (ctx->stream && !ctx->stream_exported) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* sdqh_last_error(const sdqh_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }
int sdqh_set_threads(sdqh_ctx* ctx, int threads) { if (!ctx || threads < 1) return SDQH_ERR_INVALID; ctx->threads = threads; return SDQH_OK; }
int sdqh_result_wait(sdqh_ctx* ctx) {
    if (!ctx) return SDQH_ERR_INVALID;
    if (!ctx->rs_pending) return SDQH_OK;
    ctx->rs_pending = false;
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < 2; ++i) if (ctx->rs_used[i]) HIP_TRY(ctx, hipEventSynchronize(ctx->rs_copied[i]));
    return SDQH_OK;
}
int sdqh_synchronize(sdqh_ctx* ctx) { if (!ctx) return SDQH_ERR_INVALID; if (int rc = sync_stream(ctx)) return rc; return sdqh_result_wait(ctx); }
int sdqh_last_device_ms(const sdqh_ctx* cctx, double* ms) {
    sdqh_ctx* ctx = const_cast<sdqh_ctx*>(cctx);
    if (!ctx || !ms) return SDQH_ERR_INVALID;
    if (!ctx->call_timed) { *ms = 0.0; return SDQH_OK; }
    HIP_TRY(ctx, hipEventSynchronize(ctx->call_end));
    float f = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&f, ctx->call_begin, ctx->call_end));
    *ms = f;
    return SDQH_OK;
}
int sdqh_set_profiling(sdqh_ctx* ctx, int mode) {
    if (!ctx || mode < 0 || mode > 2) return SDQH_ERR_INVALID;
    ctx->profiling = mode; ctx->prof.clear(); ctx->event_next = 0;
    return SDQH_OK;
}
int sdqh_set_profile_filter(sdqh_ctx* ctx, const char* kernel_name) {
    if (!ctx) return SDQH_ERR_INVALID;
    ctx->prof_filter = kernel_name ? kernel_name : "";
    return SDQH_OK;
}
int sdqh_profile_count(const sdqh_ctx* ctx) { return ctx ? (int)ctx->prof.size() : 0; }
int sdqh_profile_entry(const sdqh_ctx* cctx, int i, const char** name, double* ms) {
    sdqh_ctx* ctx = const_cast<sdqh_ctx*>(cctx);
    if (!ctx || i < 0 || i >= (int)ctx->prof.size() || !name || !ms) return SDQH_ERR_INVALID;
    ProfEntry& e = ctx->prof[(size_t)i];
    if (e.ms == 0.0) { (void)hipEventSynchronize(e.e1); float f = 0; if (hipEventElapsedTime(&f, e.e0, e.e1) == hipSuccess) e.ms = f; }
    *name = e.name; *ms = e.ms;
    return SDQH_OK;
}
int sdqh_profile_entry_bytes(const sdqh_ctx* ctx, int i, int64_t* model_bytes) {
    if (!ctx || i < 0 || i >= (int)ctx->prof.size() || !model_bytes) return SDQH_ERR_INVALID;
    *model_bytes = ctx->prof[(size_t)i].model_bytes;
    return SDQH_OK;
}
void* sdqh_stream(const sdqh_ctx* ctx) { if (ctx) const_cast<sdqh_ctx*>(ctx)->stream_exported = true; return ctx ? (void*)ctx->stream : nullptr; }
int sdqh_set_option(sdqh_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return SDQH_ERR_INVALID;
    const std::string n(name);
    if (n == "resident_cap" && value >= 1 && value <= 8) ctx->opt_resident_cap = (int)value;
    else if (n == "async_copies" && (value == 0 || value == 1)) ctx->opt_async_copies = (int)value;
    else if (n == "resident_stream" && value >= 1 && value <= 8) ctx->opt_resident_stream = (int)value;
    else if (n == "probe_chunk" && value >= 1 && value <= 64) ctx->opt_probe_chunk = (int)value;
    else if (n == "probe_unroll" && (value == 1 || value == 2 || value == 4)) ctx->opt_probe_unroll = (int)value;
    else if (n == "stage_batch" && (value == 2 || value == 4 || value == 8)) ctx->opt_stage_batch = (int)value;
    else if (n == "stage_eager" && (value == 0 || value == 1)) ctx->opt_stage_eager = (int)value;
    else if (n == "stage_eager_pay" && (value == 0 || value == 1)) ctx->opt_stage_eager_pay = (int)value;
    else if (n == "stage_waves_per_cu" && value >= 4 && value <= 64) ctx->opt_stage_waves_per_cu = (int)value;
    else if (n == "direct_index" && (value == 0 || value == 1)) ctx->opt_direct_index = (int)value;
    else if (n == "row_pack" && (value == 0 || value == 1)) ctx->opt_row_pack = (int)value;
    else if (n == "cluster_pack" && value >= 0 && value <= 2) ctx->opt_cluster_pack = (int)value;
    else if (n == "cluster_list" && (value == 0 || value == 1)) ctx->opt_cluster_list = (int)value;
    else if (n == "x_driven" && value >= 0 && value <= (1 << 20)) ctx->opt_x_driven = (int)value;
    else if (n == "delta8" && (value == 0 || value == 1)) ctx->opt_delta8 = (int)value;
    else if (n == "word_pairs" && (value == 0 || value == 1)) ctx->opt_word_pairs = (int)value;
    else if (n == "coarse_kb" && value >= 0 && value <= 96) ctx->opt_coarse_kb = (int)value;
    else if (n == "lookup_pipeline" && value >= -1 && value <= 1) ctx->opt_lookup_pipeline = (int)value;
    else if (n == "probe_pipeline" && value >= 0 && value <= 1) ctx->opt_probe_pipeline = (int)value;
    else if (n == "lookup_debug") ctx->opt_lookup_debug = (int)value;
    else if (n == "fuse_small" && value >= 0 && value <= 1) ctx->opt_fuse_small = (int)value;
    else if (n == "lds_key_set" && value >= 0 && value <= 1) ctx->opt_lds_key_set = (int)value;
    else if (n == "fill_ahead" && value >= 0 && value <= 1) { ctx->opt_fill_ahead = (int)value; for (auto& b : ctx->pool) b.nhabits = 0; }
    else if (n == "str_rows" && (value == 0 || value == 32 || value == 64)) ctx->opt_str_rows = (int)value;
    else if (n == "rank_increasing" && value >= 0 && value <= 1) ctx->opt_rank_increasing = (int)value;
    else if (n == "feature_min_rows" && value >= 0) ctx->opt_feature_min_rows = value;
    else if (n == "narrow" && value >= 0 && value <= 1) ctx->opt_narrow = (int)value;
    else if (n == "tight" && value >= 0 && value <= 1) ctx->opt_tight = (int)value;
    else if (n == "spin_sync" && value >= 0 && value <= 1) ctx->opt_spin_sync = (int)value;
    else if (n == "vstage" && value >= 0 && value <= 1) ctx->opt_vstage = (int)value;
    else if (n == "side_streams" && value >= 0 && value <= 1) ctx->opt_side_streams = (int)value;
    else if (n == "side_priority" && value >= 0 && value <= 1) ctx->opt_side_priority = (int)value;
    else if (n == "copy_kernel" && value >= 0 && value <= 4096) ctx->opt_copy_kernel = (int)value;      // workgroups of the library's own copy-out kernel; 0: the runtime's copy
    else if (n == "copy_nt" && value >= 0 && value <= 1) ctx->opt_copy_nt = (int)value;
    else if (n == "x_waves" && value >= 0 && value <= 256) ctx->opt_x_waves = (int)value;
    else if (n == "window" && value >= 0 && value <= 1) ctx->opt_window = (int)value;
    else if (n == "lane_int" && value >= 0 && value <= 1) ctx->opt_lane_int = (int)value;
    else if (n == "row_index" && value >= 0 && value <= 1) ctx->opt_row_index = (int)value;
    else if (n == "lane_resident" && value >= 1 && value <= 4) ctx->opt_lane_resident = (int)value;
    else if (n == "async_result" && value >= 0 && value <= 1) ctx->opt_async_result = (int)value;
    else if (n == "stage_pipeline" && value >= 0 && value <= 1) ctx->opt_stage_pipeline = (int)value;
    else if (n == "span_index" && value >= 0 && value <= 1) ctx->opt_span_index = (int)value;
    else if (n == "dense_increasing" && value >= 0 && value <= 1) ctx->opt_dense_increasing = (int)value;
    else if (n == "packed_slots" && (value == 0 || value == 1)) ctx->opt_packed_slots = (int)value;
    else if (n == "grouped_index" && (value == 0 || value == 1)) ctx->opt_grouped_index = (int)value;
    else if (n == "grouped_pairs" && (value == 0 || value == 1)) ctx->opt_grouped_pairs = (int)value;
    else if (n == "index_inline" && (value == 0 || value == 1)) ctx->opt_index_inline = (int)value;
    else if (n == "groupby_regs" && (value == 0 || value == 4 || value == 8)) ctx->opt_groupby_regs = (int)value;
    else if (n == "hash_filter" && (value == 0 || value == 1)) ctx->opt_hash_filter = (int)value;
    else if (n == "pack_ordered" && (value == 0 || value == 1)) ctx->opt_pack_ordered = (int)value;
    else if (n == "pool_trim" && value == 1) {
        // give the cached FREE blocks of this context's pool back to the runtime (the pool never shrinks by itself: 288 GB make head-room
        // cheap — until the next workload needs the memory another context's pool is sitting on).  Waits for the stream first.
        if (ctx->compile_only) return SDQH_OK;
        if (ctx->capturing) return fail(ctx, SDQH_ERR_UNSUPPORTED, "set_option: pool_trim inside a recording");
        (void)hipSetDevice(ctx->device);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, SDQH_ERR_DEVICE, "pool_trim: the stream failed"); }
        for (auto& b : ctx->pool) if (b.free && b.ptr && !b.graph_owner) { (void)hipFree(b.ptr); b.ptr = nullptr; }
        ctx->pool.erase(std::remove_if(ctx->pool.begin(), ctx->pool.end(), [](const PoolBlock& b) { return b.ptr == nullptr; }), ctx->pool.end());
    }
    else return fail(ctx, SDQH_ERR_INVALID, "set_option: unknown option or value out of range: " + n);
    return SDQH_OK;
}

int sdqh_memory_stats(sdqh_ctx* ctx, int64_t* out, int n) {
    if (!ctx || !out || n < 4) return fail(ctx, SDQH_ERR_INVALID, "memory_stats: bad arguments");
    int64_t used = 0, cached = 0, graphs = 0, blocks = 0;
    for (const auto& b : ctx->pool) {
        if (!b.ptr) continue;
        ++blocks;
        if (b.graph_owner) graphs += (int64_t)b.size;
        if (b.free && !b.graph_owner) cached += (int64_t)b.size; else used += (int64_t)b.size;
    }
    out[0] = used; out[1] = cached; out[2] = graphs; out[3] = blocks;
    return SDQH_OK;
}

// ---- columns -----------------------------------------------------------------------------------
static int new_column(sdqh_ctx* ctx, int64_t nrows, int dtype, int width, sdqh_column** out) {
    if (!ctx || !out || nrows < 0) return fail(ctx, SDQH_ERR_INVALID, "column: bad arguments");
    if (dtype != SDQH_I64 && dtype != SDQH_F64 && dtype != SDQH_STR) return fail(ctx, SDQH_ERR_INVALID, "column: bad dtype");
    if (dtype == SDQH_STR && width < 1) return fail(ctx, SDQH_ERR_INVALID, "column: STR needs width >= 1");
    sdqh_column* c = new sdqh_column();
    c->home = ctx;
    c->nrows = nrows; c->dtype = dtype; c->width = dtype == SDQH_STR ? width : 0;
    *out = c;
    return SDQH_OK;
}

int sdqh_column_alloc(sdqh_ctx* ctx, int64_t nrows, int dtype, int width, sdqh_column** out) {
    if (int rc = new_column(ctx, nrows, dtype, width, out)) return rc;
    sdqh_column* c = *out;
    (void)hipSetDevice(ctx->device);
    c->data = pool_alloc(ctx, (size_t)nrows * c->row_bytes() + 64);
    c->owned = true;
    if (!c->data) { delete c; *out = nullptr; return fail(ctx, SDQH_ERR_NOMEM, "column_alloc: out of device memory"); }
    return SDQH_OK;
}

// pageable -> pinned copy of a staging chunk on several host threads: one thread moves ~10 GB/s, the
// link takes ~50 (the first pass over SF=10's 3.9 GB of columns: 0.25 s -> PCIe-bound)
static void staging_copy(void* dst, const void* src, size_t n) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t parts = n < ((size_t)4 << 20) ? 1 : std::min<size_t>(8, std::max<unsigned>(1, hw / 2));
    if (parts <= 1) { std::memcpy(dst, src, n); return; }
    std::vector<std::thread> th;
    const size_t per = ((n / parts) + 4095) & ~(size_t)4095;
    for (size_t k = 1; k < parts; ++k) {
        const size_t b = std::min(n, k * per), e = std::min(n, (k + 1) * per);
        if (e > b) th.emplace_back([=] { std::memcpy(static_cast<char*>(dst) + b, static_cast<const char*>(src) + b, e - b); });
    }
    std::memcpy(dst, src, std::min(n, per));
    for (auto& t : th) t.join();
}

int sdqh_column_upload(sdqh_ctx* ctx, const void* host, int64_t nrows, int dtype, int width, sdqh_column** out) {
    if (nrows > 0 && !host) return fail(ctx, SDQH_ERR_INVALID, "column_upload: null host pointer");
    if (int rc = sdqh_column_alloc(ctx, nrows, dtype, width, out)) return rc;
    sdqh_column* c = *out;
    const size_t bytes = (size_t)nrows * c->row_bytes();
    // pinned staging ring: memcpy into buffer i while buffer 1-i is in flight to the device
    size_t off = 0; int i = 0;
    while (off < bytes) {
        const size_t n = std::min(STAGING_BYTES, bytes - off);
        if (ctx->staging_busy[i]) { HIP_TRY(ctx, hipEventSynchronize(ctx->staging_done[i])); ctx->staging_busy[i] = false; }
        staging_copy(ctx->staging[i], static_cast<const char*>(host) + off, n);
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<char*>(c->data) + off, ctx->staging[i], n, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipEventRecord(ctx->staging_done[i], ctx->stream));
        ctx->staging_busy[i] = true;
        off += n; i ^= 1;
    }
    for (int k = 0; k < 2; ++k) if (ctx->staging_busy[k]) { HIP_TRY(ctx, hipEventSynchronize(ctx->staging_done[k])); ctx->staging_busy[k] = false; }
    return SDQH_OK;
}

int sdqh_column_wrap(sdqh_ctx* ctx, void* device_ptr, int64_t nrows, int dtype, int width, sdqh_column** out) {
    if (nrows > 0 && (!device_ptr || (reinterpret_cast<uintptr_t>(device_ptr) & 15u))) return fail(ctx, SDQH_ERR_INVALID, "column_wrap: device pointer must be 16-byte aligned");
    if (int rc = new_column(ctx, nrows, dtype, width, out)) return rc;
    (*out)->data = device_ptr; (*out)->owned = false;
    return SDQH_OK;
}

int sdqh_column_download(sdqh_ctx* ctx, const sdqh_column* col, int64_t row0, int64_t nrows, void* host) {
    if (!ctx || !col || row0 < 0 || nrows < 0 || row0 + nrows > col->nrows || (nrows && !host)) return fail(ctx, SDQH_ERR_INVALID, "column_download: bad arguments");
    if (nrows == 0) return SDQH_OK;
    if (ctx->capturing) return fail(ctx, SDQH_ERR_UNSUPPORTED, "a call that waits for the device cannot be recorded into a plan graph");
    HIP_TRY(ctx, hipMemcpyAsync(host, static_cast<const char*>(col->data) + (size_t)row0 * col->row_bytes(), (size_t)nrows * col->row_bytes(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return SDQH_OK;
}
void* sdqh_column_data(const sdqh_column* col) { return col ? col->data : nullptr; }
int64_t sdqh_column_rows(const sdqh_column* col) { return col ? col->nrows : -1; }
int sdqh_column_dtype(const sdqh_column* col) { return col ? col->dtype : -1; }
int sdqh_column_width(const sdqh_column* col) { return col ? col->width : -1; }
int sdqh_column_minmax(sdqh_ctx* ctx, const sdqh_column* col, int64_t* mn, int64_t* mx) {
    if (!ctx || !col || !mn || !mx) return fail(ctx, SDQH_ERR_INVALID, "column_minmax: bad arguments");
    if (int rc = ensure_minmax(ctx, const_cast<sdqh_column*>(col))) return rc;
    *mn = col->mn; *mx = col->mx;
    return SDQH_OK;
}
void sdqh_column_free(sdqh_ctx* ctx, sdqh_column* col) {
    if (!col) return;
    if (ctx) {
        // a row pack does not outlive any of its columns — in whichever context of the family it was made
        sdqh_ctx* root = ctx->parent ? ctx->parent : ctx;
        std::vector<sdqh_ctx*> family{root};
        family.insert(family.end(), root->children.begin(), root->children.end());
        for (sdqh_ctx* m : family)
            for (size_t i = 0; i < m->packs.size();) {
                auto& pk = m->packs[i];
                if (std::find(pk.cols.begin(), pk.cols.end(), (const void*)col->data) != pk.cols.end() || pk.order_col == (const void*)col->data) {
                    pool_free(m, pk.data); if (pk.key32) pool_free(m, pk.key32); if (pk.lb) pool_free(m, pk.lb); m->packs.erase(m->packs.begin() + (long)i);
                }
                else ++i;
            }
        if (col->owned) attach_free(ctx, col, col->data);
        attach_free(ctx, col, col->d_minmax);
        if (col->narrow) attach_free(ctx, col, col->narrow);
        if (col->run_index) attach_free(ctx, col, col->run_index);
        if (col->delta8) attach_free(ctx, col, col->delta8);
        column_codes_release(ctx, col);
    }
    delete col;
}

// ---- K-A ---------------------------------------------------------------------------------------
int sdqh_scan_filter_sum(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, const sdqh_tuple* tuple,
                         double* out_values, int64_t* out_count) {
    if (!ctx || nrows < 0) return fail(ctx, SDQH_ERR_INVALID, "scan_filter_sum: bad arguments");
    (void)hipSetDevice(ctx->device);
    DevFilter f; DevTuple t;
    if (int rc = make_tuple(ctx, nrows, tuple, &t)) return rc;
    if (int rc = make_filter(ctx, nrows, filter, tuple, &f)) return rc;
    unsigned grid = 1;
    double* partial = nullptr;
    double* out_dev = static_cast<double*>(ctx->result_dev);
    rd_dirty(ctx);
    int lrc = with_shape(ctx, tuple->shape, [&](auto S) {
        return with_scan_filter(f, [&](auto FC) {
            using FCT = decltype(FC);
            auto launch = [&](auto kern, const DevFilter& lf, const DevTuple& lt) {
                grid = stream_grid(ctx, kern, nrows, TILE_ROWS, true);
                partial = static_cast<double*>(pool_alloc(ctx, (size_t)grid * 5 * sizeof(double)));
                if (!partial) return fail(ctx, SDQH_ERR_NOMEM, "scan_filter_sum: out of device memory");
                call_begin(ctx);
                LAUNCH(ctx, "k_scan_sum", kern, grid, lf, lt, nrows, partial);
                return (int)SDQH_OK;
            };
            if constexpr (FCT::NI >= 0 && FCT::NS == 0 && FCT::NC == 0) {        // the tuned layouts have a narrow-twin instance
                DevFilter nf = f; DevTuple nt = t;
                if (narrow_streams(ctx, nrows, filter, tuple, &nf, &nt)) return launch(k_scan_sum<decltype(S)::value, FCT, true>, nf, nt);
            }
            return launch(k_scan_sum<decltype(S)::value, FCT>, f, t);
        });
    });
    if (lrc) return lrc;
    (void)out_dev;
    LAUNCH(ctx, "k_sum_partials", k_sum_partials, 1, partial, (int)grid, static_cast<double*>(ctx->result_host));     // the fold writes the pinned host block itself
    call_end(ctx);
    int rc = sync_stream(ctx);
    pool_free(ctx, partial);
    if (rc) return rc;
    const double* h = static_cast<const double*>(ctx->result_host);
    const int nv = tuple_nv(tuple->shape);
    if (out_values) for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) out_values[k] = k < nv ? h[k] : 0.0;
    if (out_count) *out_count = reinterpret_cast<const int64_t*>(h)[4];
    return SDQH_OK;
}

int sdqh_scan_probe_sum(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, int nprobes, const sdqh_probe* probes,
                        const sdqh_tuple* tuple, double* out_values, int64_t* out_count) {
    if (!ctx || nrows < 0 || !tuple) return fail(ctx, SDQH_ERR_INVALID, "scan_probe_sum: bad arguments");
    if (nprobes == 0) return sdqh_scan_filter_sum(ctx, nrows, filter, tuple, out_values, out_count);
    (void)hipSetDevice(ctx->device);
    DevFilter f; DevTuple t; DevProbes pr;
    if (int rc = make_tuple(ctx, nrows, tuple, &t)) return rc;
    if (int rc = make_filter(ctx, nrows, filter, tuple, &f)) return rc;
    call_begin(ctx);                                   // a hash-layout probe table may have to build its index first
    if (int rc = make_probes(ctx, nrows, nprobes, probes, &pr)) return rc;
    double* out_dev = static_cast<double*>(ctx->result_dev);
    rd_dirty(ctx);
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + TPB * ROWS_PER_LOAD * 2 - 1) / (TPB * ROWS_PER_LOAD * 2), (int64_t)ctx->num_cu * ctx->opt_resident_cap));
    double* partial = static_cast<double*>(pool_alloc(ctx, (size_t)grid * 5 * sizeof(double)));
    if (!partial) return fail(ctx, SDQH_ERR_NOMEM, "scan_probe_sum: out of device memory");
    const bool tuned = f.ni == 1 && f.nf == 0 && f.ns == 0 && f.nc == 0 && nprobes == 1;
    int lrc = with_shape(ctx, tuple->shape, [&](auto S) {
        constexpr int SH = decltype(S)::value;
        if (tuned) { auto kern = k_scan_probe_sum<SH, FCfg<1, 0, 0, 1>>; LAUNCH(ctx, "k_scan_probe_sum", kern, grid, f, pr, t, nrows, partial); }
        else { auto kern = k_scan_probe_sum<SH, FGeneric>; LAUNCH(ctx, "k_scan_probe_sum", kern, grid, f, pr, t, nrows, partial); }
        return SDQH_OK;
    });
    if (lrc) { pool_free(ctx, partial); return lrc; }
    LAUNCH(ctx, "k_sum_partials", k_sum_partials, 1, partial, (int)grid, out_dev);
    call_end(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->result_host, out_dev, 5 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    int rc = sync_stream(ctx);
    pool_free(ctx, partial);
    if (rc) return rc;
    const double* h = static_cast<const double*>(ctx->result_host);
    const int nv = tuple_nv(tuple->shape);
    if (out_values) for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) out_values[k] = k < nv ? h[k] : 0.0;
    if (out_count) *out_count = reinterpret_cast<const int64_t*>(h)[4];
    return SDQH_OK;
}

// ---- K-C small ---------------------------------------------------------------------------------
int sdqh_groupby_small(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, int nkeys, const sdqh_column* const* keys,
                       const sdqh_tuple* tuple, int max_groups, int64_t* out_keys, double* out_values, int64_t* out_counts,
                       int32_t* out_ngroups) {
    if (!ctx || nrows < 0 || nkeys < 1 || nkeys > SDQH_MAX_GROUPKEYS || !keys || max_groups < 1 || max_groups > SDQH_MAX_SMALL_GROUPS || !out_ngroups)
        return fail(ctx, SDQH_ERR_INVALID, "groupby_small: bad arguments");
    (void)hipSetDevice(ctx->device);
    DevFilter f; DevTuple t; DevGroupKeys gk;
    std::memset(&gk, 0, sizeof(gk));
    if (int rc = make_tuple(ctx, nrows, tuple, &t)) return rc;
    if (int rc = make_filter(ctx, nrows, filter, tuple, &f)) return rc;
    for (int k = 0; k < nkeys; ++k) {
        if (!keys[k] || keys[k]->nrows < nrows) return fail(ctx, SDQH_ERR_INVALID, "groupby_small: bad key column");
        if (keys[k]->dtype == SDQH_STR) { if (keys[k]->width != 1) return fail(ctx, SDQH_ERR_UNSUPPORTED, "groupby_small: STR keys must have width 1"); gk.is_str[k] = 1; }
        else if (keys[k]->dtype == SDQH_I64) gk.is_str[k] = 0;
        else return fail(ctx, SDQH_ERR_UNSUPPORTED, "groupby_small: key dtype");
        gk.col[k] = keys[k]->data;
    }
    gk.nkeys = nkeys;
    constexpr int GREG = 8, GMAX = SDQH_MAX_SMALL_GROUPS;
    bool use_lds = true;
    for (int k = 0; k < nkeys; ++k) use_lds = use_lds && (ctx->lds_hint[k] == gk.col[k]);
    for (int k = nkeys; k < SDQH_MAX_GROUPKEYS; ++k) use_lds = use_lds && (ctx->lds_hint[k] == nullptr);
    // result block in ctx->result_dev: keys[64] | acc[64][4] | cnt[64] | ngroups | flags
    char* rd = static_cast<char*>(ctx->result_dev);
    unsigned long long* r_keys = reinterpret_cast<unsigned long long*>(rd);
    double* r_acc = reinterpret_cast<double*>(rd + GMAX * 8);
    int64_t* r_cnt = reinterpret_cast<int64_t*>(rd + GMAX * 40);
    int* r_ng = reinterpret_cast<int*>(rd + LG_SLOTS * 48);          // (group count, flags): where lookup_aggregate keeps its flags, so either call finds the other's tail reset
    int* r_flags = r_ng + 1;
    size_t rd_ff_before = 0;                                          // leading 0xFF bytes of the block this call found (its kernels touch the first GMAX * 8 only, and the merge resets those)
    const size_t rbytes = GMAX * 48 + 8;
    int rc = SDQH_OK;
    const char* h = static_cast<const char*>(ctx->result_host);
    // register kernel with 4 groups when these key columns produced <= 4 groups last time (or on
    // request), else 8; more than that falls back to the LDS kernel
    bool g4 = ctx->opt_groupby_regs == 4;
    if (ctx->opt_groupby_regs == 0) { g4 = true; for (int k = 0; k < SDQH_MAX_GROUPKEYS; ++k) g4 = g4 && ctx->g4_hint[k] == (k < nkeys ? gk.col[k] : nullptr); }
    for (int attempt = 0; attempt < 3; ++attempt) {
        unsigned grid = 1;
        char* blob = nullptr;
        double* pacc = nullptr; int64_t* pcnt = nullptr;
        auto carve = [&]() {
            const size_t nslots = (size_t)grid * GMAX;                 // every workgroup writes all GMAX global slots
            blob = static_cast<char*>(pool_alloc(ctx, nslots * 40 + 256));
            if (!blob) return fail(ctx, SDQH_ERR_NOMEM, "groupby_small: out of device memory");
            pacc = reinterpret_cast<double*>(blob);
            pcnt = reinterpret_cast<int64_t*>(blob + nslots * 32);
            call_begin(ctx);
            const bool clean = ctx->opt_fill_ahead && !ctx->capturing && ctx->rd_clean_ff >= (size_t)GMAX * 8 && ctx->rd_clean_zero_off == (int64_t)LG_SLOTS * 48;   // left so by the last merge
            rd_ff_before = clean ? ctx->rd_clean_ff : 0;
            rd_dirty(ctx);
            if (!clean) {
                FillList fl;
                fl.add(r_keys, GMAX * 8, 0xFF);                                      // every global group slot EMPTY_GROUP
                fl.add(r_ng, 8, 0);
                launch_fill(ctx, fl);
            }
            return SDQH_OK;
        };
        int lrc = with_shape(ctx, tuple->shape, [&](auto S) {
            constexpr int SH = decltype(S)::value;
            if (use_lds) {
                auto kern = k_groupby_lds<SH>;
                grid = stream_grid(ctx, kern, nrows);
                if (int c = carve()) return c;
                LAUNCH(ctx, "k_groupby_lds", kern, grid, f, t, gk, nrows, r_keys, pacc, pcnt, r_flags);
                return SDQH_OK;
            }
            return with_groupby_filter(f, [&](auto FC) {
                return with_group_keys(gk, [&](auto KC) {
                    using FCT = decltype(FC); using KCT = decltype(KC);
                    if constexpr (std::is_same_v<FCT, FCfg<1, 0, 0, 0>> && std::is_same_v<KCT, KCfg<1, 1>>) {      // the tuned family also has a 4-group form, and narrow-twin instances
                        DevFilter nf = f; DevTuple nt = t; DevGroupKeys ngk = gk;
                        bool narrow = narrow_streams(ctx, nrows, filter, tuple, &nf, &nt);
                        for (int k = 0; k < nkeys && narrow; ++k) {                 // the two string(1) keys through their one-byte twins
                            ngk.col[k] = keys[k]->dtype == SDQH_STR && keys[k]->width != 1 ? nullptr : ensure_narrow(ctx, const_cast<sdqh_column*>(keys[k]));
                            narrow = ngk.col[k] != nullptr;
                        }
                        auto launch = [&](auto kern, const DevFilter& lf, const DevTuple& lt) {
                            // through the twins a lane has half the bytes in flight: four resident workgroups per CU instead of two (0.293 -> 0.235 ms)
                            // The grid is taken from the 8-group instance for both forms: a run that learns "4 groups" and switches to the 4-group
                            // kernel must fold the same workgroup partials in the same order (sums bit-identical from run to run).
                            grid = narrow ? stream_grid(ctx, k_groupby_reg<SH, GREG, FCT, KCT, true>, nrows, TILE_ROWS, true, 4) : stream_grid(ctx, kern, nrows, TILE_ROWS, true);
                            if (int c = carve()) return c;
                            LAUNCH(ctx, "k_groupby_reg", kern, grid, lf, lt, narrow ? ngk : gk, nrows, r_keys, pacc, pcnt, r_flags);
                            return (int)SDQH_OK;
                        };
                        if (g4) return narrow ? launch(k_groupby_reg<SH, 4, FCT, KCT, true>, nf, nt) : launch(k_groupby_reg<SH, 4, FCT, KCT>, f, t);
                        if (narrow) return launch(k_groupby_reg<SH, GREG, FCT, KCT, true>, nf, nt);
                    }
                    auto kern = k_groupby_reg<SH, GREG, FCT, KCT>;
                    grid = stream_grid(ctx, kern, nrows, TILE_ROWS, true);
                    if (int c = carve()) return c;
                    LAUNCH(ctx, "k_groupby_reg", kern, grid, f, t, gk, nrows, r_keys, pacc, pcnt, r_flags);
                    return SDQH_OK;
                });
            });
        });
        if (lrc) { if (blob) pool_free(ctx, blob); return lrc; }
        {   // the merge writes the result block straight into the pinned host block (same layout as rd)
            char* hb = static_cast<char*>(ctx->result_host);
            LAUNCH(ctx, "k_groupby_merge", k_groupby_merge, GMAX, r_keys, pacc, pcnt, (int)grid, reinterpret_cast<double*>(hb + GMAX * 8), reinterpret_cast<int64_t*>(hb + GMAX * 40),
                   reinterpret_cast<unsigned long long*>(hb), r_ng, reinterpret_cast<int*>(hb + GMAX * 48), 1);
            ctx->rd_clean_ff = std::max<size_t>((size_t)GMAX * 8, rd_ff_before); ctx->rd_clean_zero_off = (int64_t)LG_SLOTS * 48;
        }
        call_end(ctx);
        (void)rbytes;
        rc = sync_stream(ctx);
        pool_free(ctx, blob);
        if (rc) break;
        const int flags = *reinterpret_cast<const int*>(h + GMAX * 48 + 4);
        if (flags & 2) { rc = fail(ctx, SDQH_ERR_UNSUPPORTED, "groupby_small: I64 key outside [0, 2^32-2]"); break; }
        if (flags & 1) {
            if (!use_lds && g4) { g4 = false; for (int k = 0; k < SDQH_MAX_GROUPKEYS; ++k) ctx->g4_hint[k] = nullptr; continue; }
            if (!use_lds) { use_lds = true; for (int k = 0; k < SDQH_MAX_GROUPKEYS; ++k) ctx->lds_hint[k] = k < nkeys ? gk.col[k] : nullptr; continue; }
            *out_ngroups = GMAX + 1;
            rc = fail(ctx, SDQH_ERR_OVERFLOW, "groupby_small: more groups than max_groups");
        }
        break;
    }
    if (rc) return rc;
    const unsigned long long* hk = reinterpret_cast<const unsigned long long*>(h);
    const double* ha = reinterpret_cast<const double*>(h + GMAX * 8);
    const int64_t* hc = reinterpret_cast<const int64_t*>(h + GMAX * 40);
    const int nv = tuple_nv(tuple->shape);
    int order[GMAX], ng = 0;
    for (int g = 0; g < GMAX; ++g) if (hk[g] != EMPTY_GROUP && hc[g] > 0) order[ng++] = g;
    std::sort(order, order + ng, [&](int a, int b) { return hk[a] < hk[b]; });       // global slots are claimed in racy order
    if (ng > max_groups) { *out_ngroups = ng; return fail(ctx, SDQH_ERR_OVERFLOW, "groupby_small: more groups than max_groups"); }
    for (int k = 0; k < SDQH_MAX_GROUPKEYS; ++k) ctx->g4_hint[k] = (ng <= 4 && k < nkeys) ? gk.col[k] : nullptr;   // next time: the 4-group kernel
    for (int i = 0; i < ng; ++i) {
        const int g = order[i];
        if (out_keys) for (int k = 0; k < nkeys; ++k) out_keys[i * nkeys + k] = (int64_t)((hk[g] >> (32 * k)) & 0xFFFFFFFFull);
        if (out_values) for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) out_values[i * SDQH_TUPLE_MAX_VALUES + k] = k < nv ? ha[g * 4 + k] : 0.0;
        if (out_counts) out_counts[i] = hc[g];
    }
    *out_ngroups = ng;
    return SDQH_OK;
}

// ---- K-B ---------------------------------------------------------------------------------------
static void table_release(sdqh_ctx* ctx, sdqh_table* t) {
    for (void* p : t->owned) pool_free(ctx, p);
    t->owned.clear();
}
static void* table_alloc(sdqh_ctx* ctx, sdqh_table* t, size_t bytes) {
    void* p = pool_alloc(ctx, bytes);
    if (p) t->owned.push_back(p);
    return p;
}

// stage layout shared by hash_build_unique and scan_compact
// `batch`: 128-row batches the staging kernel handles per loop step; a segment is a whole number
// of steps.  Small tables get short segments (down to one step per wave): their staging is bound
// by the dependent-load chain of a step times the steps per wave, not by bytes.
static int setup_stage(sdqh_ctx* ctx, sdqh_table* tb, int64_t nrows, const sdqh_column* key, int npay, const sdqh_column* const* payload, int batch, int waves_per_cu = 24) {
    DevStage& st = tb->stage;
    std::memset(&st, 0, sizeof(st));
    const int64_t target_segs = (int64_t)ctx->num_cu * waves_per_cu;                  // one segment per wave
    int64_t seg_rows = (nrows + target_segs - 1) / target_segs;
    const int64_t gran = (int64_t)WAVE * ROWS_PER_LOAD * std::max(1, batch);
    seg_rows = std::max<int64_t>(gran, (seg_rows + gran - 1) / gran * gran);
    if (seg_rows >= ((int64_t)1 << 31)) return fail(ctx, SDQH_ERR_UNSUPPORTED, "build: more than 2^31 rows per staging segment");   // k_build_lookup queues 32-bit offsets into its segment
    st.seg_rows = seg_rows;
    st.nseg = (int32_t)std::max<int64_t>(1, (nrows + seg_rows - 1) / seg_rows);
    st.npay = npay;
    const size_t col_bytes = (size_t)std::max<int64_t>(nrows, 1) * 8 + 64;
    st.key = static_cast<int64_t*>(table_alloc(ctx, tb, col_bytes));
    st.src_key = static_cast<const int64_t*>(key->data);
    bool ok = st.key != nullptr;
    for (int p = 0; p < npay && ok; ++p) {
        st.pay[p] = static_cast<int64_t*>(table_alloc(ctx, tb, col_bytes));
        st.src_pay[p] = static_cast<const int64_t*>(payload[p]->data);
        ok = st.pay[p] != nullptr;
    }
    st.seg_count = static_cast<uint32_t*>(table_alloc(ctx, tb, (size_t)st.nseg * 4 + 64));
    st.shits = static_cast<uint32_t*>(table_alloc(ctx, tb, (size_t)std::max<int64_t>(nrows, 1) * 4 + 64));
    ok = ok && st.seg_count != nullptr && st.shits != nullptr;
    st.acc_stride = 4;                                   // the tuple is not known yet: room for every shape
    if (ok && tb->accumulate) { st.sacc = static_cast<double*>(table_alloc(ctx, tb, (size_t)std::max<int64_t>(nrows, 1) * 32 + 64)); ok = st.sacc != nullptr; }
    if (!ok) return fail(ctx, SDQH_ERR_NOMEM, "out of device memory for the build stage");
    return SDQH_OK;
}

// Small direct-layout tables get their rank -> row array with the header and pre-filled with NO_ROW
// by the same fill launch, so the index build needs no separate (conditional) fill launch later.
static void prefill_refs(sdqh_ctx* ctx, sdqh_table* tb, FillList* fl) {
    if (!tb->bm || tb->dev.bm_shift != 0) return;
    const size_t rows = (size_t)std::max<int64_t>(tb->nrows_build, 1);
    if (rows * 4 > ((size_t)8 << 20)) return;
    uint32_t* dense = static_cast<uint32_t*>(table_alloc(ctx, tb, rows * 4 + 64));
    if (!dense) return;                                                    // ensure_index allocates (and reports) later
    tb->dev.dense_ref = dense; tb->refs_prefilled = true;
    fl->add(dense, (rows * 4 + 15) & ~(size_t)15, 0xFF);
    // owner by key offset for a plain key over a small range (supplier, nation: <= 512 K keys): lookups take the dense
    // layout's one-load path (Q9's drain asks 4 cache lines of the supplier table per row otherwise)
    const size_t range = (size_t)tb->nwords * 32;
    if (ctx->opt_span_index && tb->dev.lin_rb == 0 && range * 4 <= ((size_t)2 << 20)) {
        uint32_t* span = static_cast<uint32_t*>(table_alloc(ctx, tb, range * 4 + 64));
        if (span) { tb->span = span; fl->add(span, range * 4, 0xFF); }
    }
}

// The key -> stage-row index is built on first need: a table that is only ever used as a
// semi-join filter through its exact bitmap never pays for one.
static int ensure_index(sdqh_ctx* ctx, sdqh_table* tb) {
    if (tb->index_built || tb->bitmap_only) return SDQH_OK;
    if (tb->dev.grp_first) { tb->index_built = true; return SDQH_OK; }       // grouped layout: the stage kernel wrote the index (DevTable)
    const unsigned seg_grid = (unsigned)((tb->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE));
    const size_t rows = (size_t)std::max<int64_t>(tb->nrows_build, 1);
    if (tb->bm && tb->dev.bm_shift == 0 && tb->stage.wrow) {               // direct layout, ROW INDEX written by the stage kernel (sdqh_kernels.hpp: DevTable)
        LAUNCH(ctx, "k_wrow_fixup", k_wrow_fixup, (unsigned)((tb->stage.nseg + TPB - 1) / TPB), tb->stage, tb->wexc, tb->hdr);
        tb->dev.wprefix = tb->stage.wrow; tb->dev.dense_ref = nullptr; tb->dev.wexc = tb->wexc;
        hipError_t er = hipGetLastError();
        if (er != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string("table index launch: ") + hipGetErrorString(er));
        tb->index_built = true;
        return SDQH_OK;
    }
    if (tb->bm && tb->dev.bm_shift == 0) {                                 // direct layout
        const int nblocks = (int)((tb->nwords + RANK_BLOCK_WORDS - 1) / RANK_BLOCK_WORDS);
        uint32_t* wprefix = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64));
        uint32_t* dense = tb->refs_prefilled ? tb->dev.dense_ref : static_cast<uint32_t*>(table_alloc(ctx, tb, rows * 4 + 64));
        if (!wprefix || !dense) return fail(ctx, SDQH_ERR_NOMEM, "table index: out of device memory");
        tb->dev.wprefix = wprefix; tb->dev.dense_ref = dense;
        if (nblocks == 1 && seg_grid == 1 && tb->refs_prefilled && ctx->opt_fuse_small) {       // a tiny table: rank + insert in one launch
            LAUNCH(ctx, "k_index_small", k_index_small, 1u, tb->bm, tb->nwords, wprefix, tb->stage, tb->dev, tb->span);
            if (tb->span) tb->dev.dense_arr = tb->span;
            hipError_t es = hipGetLastError();
            if (es != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string("table index launch: ") + hipGetErrorString(es));
            tb->index_built = true;
            return SDQH_OK;
        }
        if (nblocks == 1 && ctx->opt_fuse_small && (tb->refs_prefilled || tb->keys_unique) && tb->nwords <= (uint64_t)RANK_BLOCK_WORDS) {
            // one rank block, many workgroups of rows: rank + insert in one launch, every workgroup ranking for itself (k_index_medium)
            LAUNCH(ctx, "k_index_medium", k_index_medium, seg_grid, tb->bm, (uint32_t)tb->nwords, wprefix, tb->stage, tb->dev, tb->span);
            if (tb->span) tb->dev.dense_arr = tb->span;
            hipError_t em = hipGetLastError();
            if (em != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string("table index launch: ") + hipGetErrorString(em));
            tb->index_built = true;
            return SDQH_OK;
        }
        LAUNCH(ctx, "k_rank_words", k_rank_words, (unsigned)nblocks, tb->bm, tb->nwords, wprefix, tb->stage.seg_count, tb->stage.nseg, tb->hdr);
        if (!tb->refs_prefilled && !tb->keys_unique) LAUNCH(ctx, "k_fill_refs", k_fill_refs, (unsigned)ctx->num_cu * 2, tb->stage, tb->dev);
        LAUNCH(ctx, "k_insert_direct", k_insert_direct, seg_grid, tb->stage, tb->dev, tb->span);
        if (tb->span) tb->dev.dense_arr = tb->span;                        // every later lookup: span[key - bm_lo] (the launch above ranked with its own copy of dev)
    } else {                                                               // hash layout
        // tables with payload get packed 32-byte slots (key, payload 0 / 1, owner row): one line per probe hit
        const bool packed = ctx->opt_packed_slots && tb->npay >= 1 && tb->capmax <= (1ull << 27);
        int64_t* keys = nullptr; uint32_t* rowref = nullptr; int64_t* slots = nullptr;
        if (packed) slots = static_cast<int64_t*>(table_alloc(ctx, tb, (tb->capmax + 2) * 32));
        else { keys = static_cast<int64_t*>(table_alloc(ctx, tb, (tb->capmax + 2) * 8)); rowref = static_cast<uint32_t*>(table_alloc(ctx, tb, (tb->capmax + 2) * 4)); }
        if (packed ? !slots : (!keys || !rowref)) return fail(ctx, SDQH_ERR_NOMEM, "table index: out of device memory");
        tb->dev.keys = keys; tb->dev.rowref = rowref; tb->dev.slots = slots;
        // the hashed filter in front of the slots (DevTable::hf): room for the largest capacity's filter, cleared and used at the size the
        // capacity the device settles on gives it
        uint32_t* hf = nullptr;
        if (ctx->opt_hash_filter) {
            const uint64_t bits = std::max<uint64_t>(std::min<uint64_t>(tb->capmax, HF_MAX_CAP) << HF_SHIFT, 1024);
            hf = static_cast<uint32_t*>(table_alloc(ctx, tb, (size_t)(bits / 8) + 64));
            if (!hf) (void)hipGetLastError();
        }
        tb->dev.hf = hf;
        LAUNCH(ctx, "k_clear", k_clear, (unsigned)ctx->num_cu * 4, tb->stage.seg_count, tb->stage.nseg, tb->capmax, tb->hdr, keys, rowref, slots, hf);
        LAUNCH(ctx, "k_insert", k_insert, seg_grid, tb->stage, tb->dev);
        LAUNCH(ctx, "k_insert_fixup", k_insert_fixup, seg_grid, tb->stage, tb->dev);
        if (packed) LAUNCH(ctx, "k_insert_repack", k_insert_repack, seg_grid, tb->stage, tb->dev);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string("table index launch: ") + hipGetErrorString(e));
    tb->index_built = true;
    return SDQH_OK;
}

// Dense layout (the reference's `dense(N, key)` arrays, ...generator_par.py:191-224, sized from the
// data instead of a hard-coded N): the build table's own columns are the stage, the index is one
// array over the key range.
static int build_dense(sdqh_ctx* ctx, int64_t nrows, const sdqh_column* key, int npayload, const sdqh_column* const* payload,
                       int64_t lo, int64_t hi, int accumulate, sdqh_table** out) {
    sdqh_table* tb = new sdqh_table();
    tb->npay = npayload; tb->nrows_build = nrows; tb->index_built = true; tb->accumulate = accumulate != 0;
    DevStage& st = tb->stage;
    std::memset(&st, 0, sizeof(st));
    const int64_t target_segs = (int64_t)ctx->num_cu * ctx->opt_stage_waves_per_cu;
    const int64_t gran = (int64_t)WAVE * ROWS_PER_LOAD * 8;
    int64_t seg_rows = (nrows + target_segs - 1) / target_segs;
    st.seg_rows = std::max<int64_t>(gran, (seg_rows + gran - 1) / gran * gran);
    st.nseg = (int32_t)std::max<int64_t>(1, (nrows + st.seg_rows - 1) / st.seg_rows);
    st.npay = npayload;
    st.key = const_cast<int64_t*>(static_cast<const int64_t*>(key->data));           // aliases, never written
    for (int p = 0; p < npayload; ++p) st.pay[p] = const_cast<int64_t*>(static_cast<const int64_t*>(payload[p]->data));
    const uint64_t range = (uint64_t)(hi - lo) + 1;
    st.seg_count = static_cast<uint32_t*>(table_alloc(ctx, tb, (size_t)st.nseg * 4 + 64));
    st.shits = static_cast<uint32_t*>(table_alloc(ctx, tb, (size_t)nrows * 4 + 64));
    tb->hdr = static_cast<TableHeader*>(table_alloc(ctx, tb, sizeof(TableHeader)));
    // accumulators (a dense group domain summed into in one pass: xplan's large group-bys over a small key range): one zeroed row of four per entry
    const size_t acc_bytes = accumulate ? (size_t)nrows * 32 : 0;
    if (accumulate) {
        st.acc_stride = 4;
        st.sacc = static_cast<double*>(table_alloc(ctx, tb, acc_bytes + 64));
        if (!st.sacc) { table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_NOMEM, "dense build: out of device memory"); }
        tb->dev.sacc = st.sacc; tb->dev.acc_stride = 4;
    }
    // a strictly increasing key column (orders by o_orderkey, any table by its primary key)
    const bool increasing = ctx->opt_dense_increasing && column_is_increasing(ctx, const_cast<sdqh_column*>(key));
    const int64_t* kc = static_cast<const int64_t*>(key->data);
    const unsigned grid = (unsigned)std::min<int64_t>((nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 8);
    if (increasing && ctx->opt_rank_increasing && (range * 4 > ((uint64_t)16 << 20) || ctx->opt_feature_min_rows == 0)) {      // (feature_min_rows 0: the suites run it on small tables too)
        // ... over a wide range: the direct layout with rank = row (k_rank_increasing): a bitmap and one row number per bitmap word
        // instead of four bytes per key of the range (Q9's orders at SF=10: 15 MB written instead of 240 MB)
        tb->nwords = (range + 31) / 32;
        tb->bm = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64));
        uint32_t* wprefix = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64));
        uint32_t* wpair = ctx->opt_word_pairs ? static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 8 + 64)) : nullptr;      // (none: lookups read the two arrays)
        if (!st.seg_count || !st.shits || !tb->hdr || !tb->bm || !wprefix) { table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_NOMEM, "dense build: out of device memory"); }
        st.hdr = tb->hdr;
        tb->dev.hdr = tb->hdr; tb->dev.shits = st.shits; tb->dev.bm_lo = lo; tb->dev.bm_hi = hi; tb->dev.bm = tb->bm; tb->dev.bm_shift = 0; tb->dev.wprefix = wprefix;
        tb->dev.wpair = reinterpret_cast<const unsigned long long*>(wpair);
        for (int p = 0; p < npayload; ++p) tb->dev.pay[p] = st.pay[p];
        call_begin(ctx);
        { FillList fl; fl.add(tb->hdr, sizeof(TableHeader), 0); fl.add(st.shits, (size_t)nrows * 4, 0); fl.add(tb->bm, tb->nwords * 4, 0); if (wpair) fl.add(wpair, tb->nwords * 8, 0); if (accumulate) fl.add(st.sacc, acc_bytes, 0); launch_fill(ctx, fl); }
        {
            const unsigned rgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + TPB * RANK_INC_NB - 1) / (TPB * RANK_INC_NB), (int64_t)ctx->num_cu * 32));
            const int32_t* k32 = (ctx->opt_narrow && nrows >= ctx->opt_feature_min_rows && !key->transient) ? static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(key))) : nullptr;
            if (k32) { auto kern = k_rank_increasing<int32_t>; LAUNCH(ctx, "k_rank_increasing", kern, rgrid, k32, nrows, lo, tb->bm, wprefix, tb->hdr, st.seg_count, st.nseg, st.seg_rows, wpair); }
            else { auto kern = k_rank_increasing<int64_t>; LAUNCH(ctx, "k_rank_increasing", kern, rgrid, kc, nrows, lo, tb->bm, wprefix, tb->hdr, st.seg_count, st.nseg, st.seg_rows, wpair); }
        }
        call_end(ctx);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) { table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_DEVICE, std::string("dense build launch: ") + hipGetErrorString(e2)); }
        *out = tb;
        return SDQH_OK;
    }
    uint32_t* arr = static_cast<uint32_t*>(table_alloc(ctx, tb, range * 4 + 64));
    if (!st.seg_count || !st.shits || !tb->hdr || !arr) { table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_NOMEM, "dense build: out of device memory"); }
    st.hdr = tb->hdr;
    tb->dev.hdr = tb->hdr; tb->dev.shits = st.shits; tb->dev.bm_lo = lo; tb->dev.bm_hi = hi; tb->dev.dense_arr = arr;
    for (int p = 0; p < npayload; ++p) tb->dev.pay[p] = st.pay[p];
    // ... over a narrow range fills the array in one pass: no prefill of the cells, no verification pass
    call_begin(ctx);
    if (increasing) {
        { FillList fl; fl.add(tb->hdr, sizeof(TableHeader), 0); fl.add(st.shits, (size_t)nrows * 4, 0); if (accumulate) fl.add(st.sacc, acc_bytes, 0); launch_fill(ctx, fl); }
        LAUNCH(ctx, "k_full_counts", k_full_counts, (unsigned)((st.nseg + TPB - 1) / TPB), st.seg_count, st.nseg, st.seg_rows, nrows);
        LAUNCH(ctx, "k_dense_fill_increasing", k_dense_fill_increasing, grid, kc, nrows, lo, arr, tb->hdr);
    } else {
        { FillList fl; fl.add(tb->hdr, sizeof(TableHeader), 0); fl.add(st.shits, (size_t)nrows * 4, 0); fl.add(arr, range * 4, 0xFF); if (accumulate) fl.add(st.sacc, acc_bytes, 0); launch_fill(ctx, fl); }
        LAUNCH(ctx, "k_full_counts", k_full_counts, (unsigned)((st.nseg + TPB - 1) / TPB), st.seg_count, st.nseg, st.seg_rows, nrows);
        LAUNCH(ctx, "k_dense_fill", k_dense_fill, grid, kc, nrows, lo, arr);
        LAUNCH(ctx, "k_dense_verify", k_dense_verify, grid, kc, nrows, lo, arr, tb->hdr);
        LAUNCH(ctx, "k_dense_fixup", k_dense_fixup, grid, kc, nrows, lo, arr, tb->hdr);
    }
    call_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_DEVICE, std::string("dense build launch: ") + hipGetErrorString(e)); }
    *out = tb;
    return SDQH_OK;
}

int sdqh_hash_build_unique(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, int nprobes, const sdqh_probe* probes,
                           const sdqh_column* key, int npayload, const sdqh_column* const* payload, int accumulate, sdqh_table** out) {
    if (!ctx || nrows < 0 || !out || npayload < 0 || npayload > SDQH_MAX_PAYLOAD) return fail(ctx, SDQH_ERR_INVALID, "hash_build_unique: bad arguments");
    if (nrows >= 0xFFFFFFFEll) return fail(ctx, SDQH_ERR_UNSUPPORTED, "hash_build_unique: build side limited to 2^32-2 rows per GPU");
    (void)hipSetDevice(ctx->device);
    DevFilter f; DevProbes pr;
    if (int rc = make_filter(ctx, nrows, filter, nullptr, &f)) return rc;
    if (int rc = make_probes(ctx, nrows, nprobes, probes, &pr)) return rc;
    if (int rc = check_col(ctx, key, SDQH_I64, nrows, "build key")) return rc;
    for (int p = 0; p < npayload; ++p)
        if (!payload || !payload[p] || payload[p]->nrows < nrows || payload[p]->dtype == SDQH_STR)
            return fail(ctx, SDQH_ERR_INVALID, "hash_build_unique: payload columns must be I64/F64 and cover nrows");
    // exact key bitmap when the key range is dense enough to be worth it
    int64_t lo = 0, hi = -1; bool want_bm = false;
    if (nrows > 0) {
        if (int rc = ensure_minmax(ctx, const_cast<sdqh_column*>(key))) return rc;
        lo = key->mn; hi = key->mx;
        if (hi >= lo && lo > INT64_MIN / 2 && hi < INT64_MAX / 2) {        // no overflow in hi - lo
            const uint64_t range = (uint64_t)(hi - lo) + 1;
            want_bm = range <= (1ull << 31) && range <= 64ull * (uint64_t)std::max<int64_t>(nrows, 1024);
        }
    }
    // dense layout: an unfiltered, unprobed build on a dense key range indexes the source columns in place
    const bool unfiltered = f.ni == 0 && f.nf == 0 && f.ns == 0 && f.nc == 0 && nprobes == 0;
    if (unfiltered && nrows > 0 && ctx->opt_direct_index && hi >= lo && lo > INT64_MIN / 2 && hi < INT64_MAX / 2 &&
        (uint64_t)(hi - lo) + 1 <= 16ull * (uint64_t)nrows && (uint64_t)(hi - lo) + 1 <= (1ull << 30))
        return build_dense(ctx, nrows, key, npayload, payload, lo, hi, accumulate, out);
    sdqh_table* tb = new sdqh_table();
    tb->npay = npayload; tb->accumulate = accumulate != 0; tb->nrows_build = nrows;
    if (nrows > 0 && column_is_increasing(ctx, const_cast<sdqh_column*>(key))) tb->keys_unique = true;
    // the tuned orders-like instance family runs opt_stage_batch batches per step, every other instance STAGE_BATCH
    const bool tuned_family = f.ns == 0 && f.nf == 0 && f.ni == 1 && nprobes == 1 && npayload == 2;
    const bool string_family = f.ns == 1 && f.nf == 0 && f.ni == 0 && nprobes == 0;
    // the streaming-heavy tuned family wants fewer, longer segments (DRAM page locality); everything else more waves (latency chains)
    int rc = setup_stage(ctx, tb, nrows, key, npayload, payload, tuned_family ? ctx->opt_stage_batch : (string_family ? 1 : STAGE_BATCH),
                         tuned_family ? ctx->opt_stage_waves_per_cu : 24);
    uint64_t capmax = 1024;
    while (capmax < 2 * (uint64_t)std::max<int64_t>(nrows, 1)) capmax <<= 1;
    tb->capmax = capmax;
    if (!rc) {
        tb->hdr = static_cast<TableHeader*>(table_alloc(ctx, tb, sizeof(TableHeader)));
        if (want_bm && ctx->opt_direct_index) { tb->nwords = ((uint64_t)(hi - lo) + 32) / 32; tb->bm = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64)); }
        if (!tb->hdr || (tb->nwords && !tb->bm)) rc = fail(ctx, SDQH_ERR_NOMEM, "hash_build_unique: out of device memory");
        else {
            tb->dev.hdr = tb->hdr; tb->dev.shits = tb->stage.shits; tb->dev.sacc = tb->stage.sacc; tb->dev.acc_stride = tb->stage.acc_stride;
            tb->dev.bm = tb->bm; tb->dev.bm_lo = lo; tb->dev.bm_hi = hi; tb->dev.bitmap_only = 0; tb->dev.bm_shift = 0;
            for (int p = 0; p < npayload; ++p) tb->dev.pay[p] = tb->stage.pay[p];
            tb->stage.bm = tb->bm; tb->stage.bm_lo = lo; tb->stage.bm_hi = hi; tb->stage.hdr = tb->hdr; tb->stage.bm_shift = 0;
            call_begin(ctx);
            const unsigned seg_grid = (unsigned)((tb->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE));
            FillList fl; fl.add(tb->hdr, sizeof(TableHeader), 0); if (tb->bm) fl.add(tb->bm, tb->nwords * 4, 0); prefill_refs(ctx, tb, &fl);
            DevFill pre; std::memset(&pre, 0, sizeof(pre));
            prune_clean(ctx, &fl);
            if (seg_grid == 1 && ctx->opt_fuse_small && fl.most <= ((uint64_t)1 << 20) && fl.f.n <= FILL_MAX) pre = fl.pre(); else launch_fill(ctx, fl);      // a tiny table: the staging kernel does its own fill
            // string predicate: fields staged through LDS, 64 rows per wave at a time (up to 64 KB per workgroup)
            const unsigned str_rows = (f.ns && f.swidth > 0 && f.swidth <= 128) ? (f.swidth <= 64 ? 64u : 32u) : 0u;
            f.slds = str_rows;
            stage_text_twin(ctx, nrows, filter, &f);
            const size_t stage_lds = (size_t)(TPB / WAVE) * str_lds_words(f) * 4 + (f.sc8 ? STR_LDS_SLACK : 0);       // at most 64 KB per workgroup
            with_stage_filter(f, nprobes, [&](auto FC) {
                using FCT = decltype(FC);
                if constexpr (std::is_same_v<FCT, FCfg<1, 0, 0, 1>>) {          // the tuned instance family (orders-like build side)
                    const int sb = ctx->opt_stage_batch, eg = ctx->opt_stage_eager;
                    if (npayload == 2) {
                        const int ep = ctx->opt_stage_eager_pay;
                        if (ctx->opt_stage_pipeline && eg == 1 && ep == 0) {
#define STAGE_PIPE(SB_) if (sb == SB_) { auto kern = k_stage<FCT, 2, SB_, true, false, true>; LAUNCH(ctx, "k_stage", kern, seg_grid, f, pr, tb->stage, nrows, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), pre); return SDQH_OK; }
                            STAGE_PIPE(2) STAGE_PIPE(4)
#undef STAGE_PIPE
                        }
                        if (ctx->opt_narrow && nrows >= ctx->opt_feature_min_rows && sb == 4 && eg == 1 && ep == 0) {       // first predicate + streamed probe key through their narrow twins
                            const int32_t* nkey = static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(probes[0].key)));
                            const int32_t* npred0 = nkey ? static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(filter->ipred[0].col))) : nullptr;
                            if (nkey && npred0) { auto kern = k_stage<FCT, 2, 4, true, false, false, true>; LAUNCH(ctx, "k_stage", kern, seg_grid, f, pr, tb->stage, nrows, nkey, npred0, pre); return SDQH_OK; }
                        }
#define STAGE_VARIANT(SB_, EG_, EP_) if (sb == SB_ && eg == EG_ && ep == EP_) { auto kern = k_stage<FCT, 2, SB_, EG_ != 0, EP_ != 0>; LAUNCH(ctx, "k_stage", kern, seg_grid, f, pr, tb->stage, nrows, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), pre); return SDQH_OK; }
                        STAGE_VARIANT(2, 1, 1) STAGE_VARIANT(4, 1, 1) STAGE_VARIANT(2, 1, 0) STAGE_VARIANT(4, 1, 0) STAGE_VARIANT(8, 1, 0) STAGE_VARIANT(4, 0, 0)
#undef STAGE_VARIANT
                    }
                }
                if constexpr (FCT::NS == 1) {                                   // string family: one batch per step (few registers, many waves)
                    if (npayload == 0) { auto kern = k_stage<FCT, 0, 1>; LAUNCH_LDS(ctx, "k_stage", kern, seg_grid, stage_lds, f, pr, tb->stage, nrows, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), pre); return SDQH_OK; }
                    auto kern = k_stage<FCT, -1, 1>;
                    LAUNCH_LDS(ctx, "k_stage", kern, seg_grid, stage_lds, f, pr, tb->stage, nrows, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), pre);
                    return SDQH_OK;
                } else {
                if (npayload == 0) { auto kern = k_stage<FCT, 0>; LAUNCH_LDS(ctx, "k_stage", kern, seg_grid, stage_lds, f, pr, tb->stage, nrows, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), pre); return SDQH_OK; }
                if (npayload == 1) { auto kern = k_stage<FCT, 1>; LAUNCH_LDS(ctx, "k_stage", kern, seg_grid, stage_lds, f, pr, tb->stage, nrows, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), pre); return SDQH_OK; }
                if (npayload == 2) { auto kern = k_stage<FCT, 2>; LAUNCH_LDS(ctx, "k_stage", kern, seg_grid, stage_lds, f, pr, tb->stage, nrows, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), pre); return SDQH_OK; }
                auto kern = k_stage<FCT, -1>;
                LAUNCH_LDS(ctx, "k_stage", kern, seg_grid, stage_lds, f, pr, tb->stage, nrows, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), pre);
                return SDQH_OK;
                }
            });
            call_end(ctx);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) rc = fail(ctx, SDQH_ERR_DEVICE, std::string("hash_build_unique launch: ") + hipGetErrorString(e));
        }
    }
    if (rc) { table_release(ctx, tb); delete tb; return rc; }
    *out = tb;
    return SDQH_OK;
}

int sdqh_build_key_set(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, int nprobes, const sdqh_probe* probes,
                       const sdqh_column* key, sdqh_table** out) {
    if (!ctx || nrows < 0 || !out) return fail(ctx, SDQH_ERR_INVALID, "build_key_set: bad arguments");
    (void)hipSetDevice(ctx->device);
    DevFilter f; DevProbes pr;
    if (int rc = make_filter(ctx, nrows, filter, nullptr, &f)) return rc;
    if (int rc = check_col(ctx, key, SDQH_I64, nrows, "build key")) return rc;
    int64_t lo = 0, hi = -1;
    if (nrows > 0) {
        if (int rc = ensure_minmax(ctx, const_cast<sdqh_column*>(key))) return rc;
        lo = key->mn; hi = key->mx;
        const bool fits = hi >= lo && lo > INT64_MIN / 2 && hi < INT64_MAX / 2 && (uint64_t)(hi - lo) + 1 <= (1ull << 31) &&
                          (uint64_t)(hi - lo) + 1 <= 64ull * (uint64_t)std::max<int64_t>(nrows, 1024);
        if (!fits) return fail(ctx, SDQH_ERR_UNSUPPORTED, "build_key_set: key range too wide or sparse for a bitmap");
    } else { lo = 0; hi = 0; }
    call_begin(ctx);
    if (int rc = make_probes(ctx, nrows, nprobes, probes, &pr)) return rc;
    sdqh_table* tb = new sdqh_table();
    tb->bitmap_only = true; tb->index_built = true; tb->nrows_build = nrows;
    tb->nwords = ((uint64_t)(hi - lo) + 32) / 32;
    tb->bm = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64));
    tb->hdr = static_cast<TableHeader*>(table_alloc(ctx, tb, sizeof(TableHeader)));
    if (!tb->bm || !tb->hdr) { table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_NOMEM, "build_key_set: out of device memory"); }
    tb->dev.bm = tb->bm; tb->dev.bm_lo = lo; tb->dev.bm_hi = hi; tb->dev.bitmap_only = 1; tb->dev.hdr = tb->hdr;
    FillList fl; fl.add(tb->bm, (tb->nwords * 4 + 15) & ~(uint64_t)15, 0); fl.add(tb->hdr, sizeof(TableHeader), 0);
    DevFill pre; std::memset(&pre, 0, sizeof(pre));
    const unsigned grid0 = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + TPB * ROWS_PER_LOAD * 2 - 1) / (TPB * ROWS_PER_LOAD * 2), (int64_t)ctx->num_cu * ctx->opt_resident_cap));
    prune_clean(ctx, &fl);
    if (nrows > 0 && grid0 == 1 && ctx->opt_fuse_small && fl.most <= ((uint64_t)1 << 20) && fl.f.n <= FILL_MAX) pre = fl.pre(); else launch_fill(ctx, fl);      // a tiny table clears its bitmap in the key-set kernel itself
    if (nrows > 0) {
        const int64_t* kc = static_cast<const int64_t*>(key->data);
        const unsigned grid = grid0;
        // 32 rows per round from 33 code units up: 64 rows of Q9's p_name are 14 KiB per wave, two workgroups per CU (0.110 -> 0.101 ms)
        unsigned str_rows = (f.ns && f.swidth > 0 && f.swidth <= 128) ? (f.swidth <= 32 ? 64u : 32u) : 0u;
        if (str_rows && ctx->opt_str_rows) str_rows = (unsigned)ctx->opt_str_rows;
        f.slds = str_rows;
        stage_text_twin(ctx, nrows, filter, &f);
        const size_t lds = (size_t)(TPB / WAVE) * str_lds_words(f) * 4 + (f.sc8 ? STR_LDS_SLACK : 0);
        with_stage_filter(f, nprobes, [&](auto FC) {
            if (!pre.n && launch_key_set_lds<decltype(FC)>(ctx, f, pr, key, nrows, lo, hi, tb)) return SDQH_OK;
            const int32_t* none = nullptr;
            if constexpr (std::is_same_v<decltype(FC), FCfg<0, 0, 0, 0, 1>>) {      // Q4's late lineitems: key and the two compared dates through their twins
                if (ctx->opt_narrow && nrows >= ctx->opt_feature_min_rows && !f.cf64[0]) {
                    const int32_t* nk = static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(key)));
                    const int32_t* na = nk ? static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(filter->cpred[0].a))) : nullptr;
                    const int32_t* nb = na ? static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(filter->cpred[0].b))) : nullptr;
                    if (nb) {
                        auto nkern = k_key_set<decltype(FC), true>;
                        LAUNCH_LDS(ctx, "k_key_set", nkern, grid, lds, f, pr, kc, nrows, lo, hi, tb->bm, pre, nk, na, nb);
                        return SDQH_OK;
                    }
                }
            }
            auto kern = k_key_set<decltype(FC)>;
            LAUNCH_LDS(ctx, "k_key_set", kern, grid, lds, f, pr, kc, nrows, lo, hi, tb->bm, pre, none, none, none);
            return SDQH_OK;
        });
    }
    call_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_DEVICE, std::string("build_key_set launch: ") + hipGetErrorString(e)); }
    *out = tb;
    return SDQH_OK;
}

int sdqh_table_size(sdqh_ctx* ctx, const sdqh_table* ctable, int64_t* entries) {
    sdqh_table* table = const_cast<sdqh_table*>(ctable);
    if (!ctx || !table || !entries) return fail(ctx, SDQH_ERR_INVALID, "table_size: bad arguments");
    (void)hipSetDevice(ctx->device);
    if (table->bitmap_only) {                                   // membership-only table: its size is the population count
        if (!table->hdr) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_size: a table made from an imported bitmap has no size");
        HIP_TRY(ctx, hipMemsetAsync(&table->hdr->counted, 0, 8, ctx->stream));
        LAUNCH(ctx, "k_popcount", k_popcount, (unsigned)ctx->num_cu * 4, table->bm, table->nwords, reinterpret_cast<unsigned long long*>(&table->hdr->counted));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->result_host, &table->hdr->counted, 8, hipMemcpyDeviceToHost, ctx->stream));
        if (int rc = sync_stream(ctx)) return rc;
        *entries = *static_cast<const int64_t*>(ctx->result_host);
        return SDQH_OK;
    }
    if (int rc = ensure_index(ctx, table)) return rc;
    HIP_TRY(ctx, hipMemsetAsync(&table->hdr->counted, 0, 8, ctx->stream));
    LAUNCH(ctx, "k_count", k_count, (unsigned)((table->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE)), table->stage, table->dev);
    TableHeader* h = static_cast<TableHeader*>(ctx->result_host);
    HIP_TRY(ctx, hipMemcpyAsync(h, table->hdr, sizeof(TableHeader), hipMemcpyDeviceToHost, ctx->stream));
    if (int rc = sync_stream(ctx)) return rc;
    *entries = (int64_t)h->counted;
    return SDQH_OK;
}

void sdqh_table_free(sdqh_ctx* ctx, sdqh_table* table) {
    if (!table) return;
    if (ctx) table_release(ctx, table);
    delete table;
}

// ---- K-C large ---------------------------------------------------------------------------------
int sdqh_hash_probe_aggregate(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, sdqh_table* table,
                              const sdqh_column* key, const sdqh_tuple* tuple) {
    if (!ctx || nrows < 0 || !table) return fail(ctx, SDQH_ERR_INVALID, "hash_probe_aggregate: bad arguments");
    if (!table->accumulate || table->bitmap_only) return fail(ctx, SDQH_ERR_INVALID, "hash_probe_aggregate: table was built without accumulators");
    (void)hipSetDevice(ctx->device);
    DevFilter f; DevTuple t;
    if (int rc = make_tuple(ctx, nrows, tuple, &t)) return rc;
    if (int rc = make_filter(ctx, nrows, filter, tuple, &f)) return rc;
    if (int rc = check_col(ctx, key, SDQH_I64, nrows, "probe key")) return rc;
    if (tuple_nv(tuple->shape) > table->stage.acc_stride) return fail(ctx, SDQH_ERR_INVALID, "hash_probe_aggregate: the table's entries have room for fewer values than this tuple has");
    table->compact_valid = false;
    table->nv = std::max(0, tuple_nv(tuple->shape));
    const int64_t* kc = static_cast<const int64_t*>(key->data);
    call_begin(ctx);
    if (int rc = ensure_index(ctx, table)) return rc;
    // software-pipelined streaming part (see k_lookup_agg): off by default — this loop streams two columns at 5.3 TB/s
    // already; measured at SF=10: Q3 0.175 -> 0.186 ms with it, Q18's sum per l_orderkey 0.518 -> 0.499 ms
    const int pipeline = ctx->opt_probe_pipeline;
    int lrc = with_shape(ctx, tuple->shape, [&](auto S) {
        return with_scan_filter(f, [&](auto FC) {
            constexpr int SH = decltype(S)::value; using FCT = decltype(FC);
            const int32_t* nkey = nullptr; const int32_t* npred0 = nullptr;
            auto launch = [&](auto kern, int pu) {
                const unsigned grid = stream_grid(ctx, kern, nrows, TPB * ROWS_PER_LOAD * pu * ctx->opt_probe_chunk);
                // the kernel queues candidate rows as 32-bit offsets from its current chunk and rebases them by one grid stride
                if ((int64_t)grid * ctx->opt_probe_chunk * (TPB * ROWS_PER_LOAD * pu) >= ((int64_t)1 << 31)) return fail(ctx, SDQH_ERR_UNSUPPORTED, "hash_probe_aggregate: probe_chunk too large for this grid");
                LAUNCH(ctx, "k_probe_agg", kern, grid, f, t, table->dev, kc, nrows, ctx->opt_probe_chunk, pipeline, nkey, npred0);
                return (int)SDQH_OK;
            };
            // narrow twins of the two streamed columns (key, first integer predicate): the layouts with at most one integer predicate
            if constexpr (std::is_same_v<FCT, FCfg<1, 0, 0, 0>> || std::is_same_v<FCT, FCfg<0, 0, 0, 0>>) {
                // (not for the row-keyed group-by, sdqh_groupby_key: every row hits there, the drain bounds it, and the conversions cost 6 %)
                if (ctx->opt_narrow && nrows >= ctx->opt_feature_min_rows && !ctx->in_groupby_key) {
                    nkey = static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(key)));
                    if (nkey && f.ni == 1) { npred0 = static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(filter->ipred[0].col))); if (!npred0) nkey = nullptr; }
                    if (nkey) return launch(k_probe_agg<SH, FCT, PROBE_UNROLL, true>, PROBE_UNROLL);
                }
            }
            if constexpr (SH == SDQH_TUPLE_A_1MB && std::is_same_v<FCT, FCfg<1, 0, 0, 0>>) {     // the tuned instance family
                if (ctx->opt_probe_unroll == 4) return launch(k_probe_agg<SH, FCT, 4>, 4);
                if (ctx->opt_probe_unroll == 1) return launch(k_probe_agg<SH, FCT, 1>, 1);
            }
            // Filters outside the instantiated layouts run the generic instance with ONE row pair per lane in flight: with two, its six
            // tuple shapes each spilled 64 bytes per lane to scratch (the run-time predicate counts keep every column pointer live).  The
            // planner prefers a row program for such loops anyway (Engine.program_routes); this keeps the fixed-shape ABI whole.
            if constexpr (std::is_same_v<FCT, FGeneric>) return launch(k_probe_agg<SH, FCT, 1>, 1);
            else return launch(k_probe_agg<SH, FCT>, PROBE_UNROLL);
        });
    });
    if (lrc) return lrc;
    call_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string("hash_probe_aggregate launch: ") + hipGetErrorString(e));
    return SDQH_OK;
}

// ---- K-C large domain on a row key; HAVING -------------------------------------------------------------
int sdqh_groupby_key(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, const sdqh_column* key, const sdqh_tuple* tuple, sdqh_table** out) {
    if (!ctx || nrows < 0 || !out || !tuple) return fail(ctx, SDQH_ERR_INVALID, "groupby_key: bad arguments");
    if (nrows >= 0xFFFFFFFEll) return fail(ctx, SDQH_ERR_UNSUPPORTED, "groupby_key: limited to 2^32-2 rows per GPU");
    (void)hipSetDevice(ctx->device);
    if (int rc = check_col(ctx, key, SDQH_I64, nrows, "group key")) return rc;
    int64_t lo = 0, hi = -1; bool dense = false;
    if (nrows > 0) {
        if (int rc = ensure_minmax(ctx, const_cast<sdqh_column*>(key))) return rc;
        lo = key->mn; hi = key->mx;
        dense = ctx->opt_direct_index && hi >= lo && lo > INT64_MIN / 2 && hi < INT64_MAX / 2 && (uint64_t)(hi - lo) + 1 <= (1ull << 31) &&
                (uint64_t)(hi - lo) + 1 <= 64ull * (uint64_t)std::max<int64_t>(nrows, 1024);
    }
    if (!dense) {                                      // any int64 keys: stage every passing row, index (first row owns the entry), add the rows
        sdqh_table* tb = nullptr;
        call_begin(ctx);
        ++ctx->nested;
        int rc = sdqh_hash_build_unique(ctx, nrows, filter, 0, nullptr, key, 0, nullptr, 1, &tb);
        if (!rc) { ctx->in_groupby_key = true; rc = sdqh_hash_probe_aggregate(ctx, nrows, filter, tb, key, tuple); ctx->in_groupby_key = false; if (rc) sdqh_table_free(ctx, tb); }
        --ctx->nested;
        call_end(ctx);
        if (rc) return rc;
        *out = tb;
        return SDQH_OK;
    }
    // dense key range: distinct keys from a bitmap, entries laid out by rank, rows added by the probe kernel
    DevFilter f; DevProbes pr; std::memset(&pr, 0, sizeof(pr));
    if (int rc = make_filter(ctx, nrows, filter, nullptr, &f)) return rc;
    sdqh_table* tb = new sdqh_table();
    tb->accumulate = true; tb->npay = 0; tb->nrows_build = nrows; tb->index_built = true;
    const uint64_t range = (uint64_t)(hi - lo) + 1;
    const uint64_t dmax = std::min<uint64_t>((uint64_t)nrows, range);           // at most this many distinct keys
    tb->nwords = (range + 31) / 32;
    DevStage& st = tb->stage;
    std::memset(&st, 0, sizeof(st));
    st.seg_rows = (int64_t)WAVE * ROWS_PER_LOAD * 16;                           // 2048 entries per compaction wave
    st.nseg = (int32_t)std::max<uint64_t>(1, (dmax + (uint64_t)st.seg_rows - 1) / (uint64_t)st.seg_rows);
    tb->bm = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64));
    tb->hdr = static_cast<TableHeader*>(table_alloc(ctx, tb, sizeof(TableHeader)));
    uint32_t* wprefix = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64));
    uint32_t* dense_ref = static_cast<uint32_t*>(table_alloc(ctx, tb, dmax * 4 + 64));
    st.key = static_cast<int64_t*>(table_alloc(ctx, tb, dmax * 8 + 64));
    st.shits = static_cast<uint32_t*>(table_alloc(ctx, tb, dmax * 4 + 64));
    st.acc_stride = std::max(1, tuple_nv(tuple->shape));               // the tuple is known here: no room for values it does not have
    st.sacc = static_cast<double*>(table_alloc(ctx, tb, dmax * 8 * (size_t)st.acc_stride + 64));
    st.seg_count = static_cast<uint32_t*>(table_alloc(ctx, tb, (size_t)st.nseg * 4 + 64));
    if (!tb->bm || !tb->hdr || !wprefix || !dense_ref || !st.key || !st.shits || !st.sacc || !st.seg_count) {
        table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_NOMEM, "groupby_key: out of device memory");
    }
    st.hdr = tb->hdr; st.bm = tb->bm; st.bm_lo = lo; st.bm_hi = hi;
    tb->dev.hdr = tb->hdr; tb->dev.bm = tb->bm; tb->dev.bm_lo = lo; tb->dev.bm_hi = hi; tb->dev.wprefix = wprefix; tb->dev.dense_ref = dense_ref;
    tb->dev.shits = st.shits; tb->dev.sacc = st.sacc; tb->dev.acc_stride = st.acc_stride;
    call_begin(ctx);
    { FillList fl; fl.add(tb->bm, (tb->nwords * 4 + 15) & ~(uint64_t)15, 0); fl.add(tb->hdr, sizeof(TableHeader), 0); launch_fill(ctx, fl); }
    const int64_t* kc = static_cast<const int64_t*>(key->data);
    const unsigned sgrid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + TPB * ROWS_PER_LOAD * 2 - 1) / (TPB * ROWS_PER_LOAD * 2), (int64_t)ctx->num_cu * ctx->opt_resident_cap));
    with_stage_filter(f, 0, [&](auto FC) {
        if (launch_key_set_lds<decltype(FC)>(ctx, f, pr, key, nrows, lo, hi, tb)) return SDQH_OK;
        auto kern = k_key_set<decltype(FC)>;
        { DevFill nofill; std::memset(&nofill, 0, sizeof(nofill)); LAUNCH_LDS(ctx, "k_key_set", kern, sgrid, 0, f, pr, kc, nrows, lo, hi, tb->bm, nofill, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr)); }
        return SDQH_OK;
    });
    const int nblocks = (int)((tb->nwords + RANK_BLOCK_WORDS - 1) / RANK_BLOCK_WORDS);
    LAUNCH(ctx, "k_rank_words", k_rank_words, (unsigned)nblocks, tb->bm, tb->nwords, wprefix, static_cast<const uint32_t*>(nullptr), 0, tb->hdr);
    LAUNCH(ctx, "k_gk_layout", k_gk_layout, (unsigned)ctx->num_cu * 8, st, tb->dev, lo, tb->nwords);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { call_end(ctx); table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_DEVICE, std::string("groupby_key launch: ") + hipGetErrorString(e)); }
    ++ctx->nested;
    ctx->in_groupby_key = true;
    const int prc = sdqh_hash_probe_aggregate(ctx, nrows, filter, tb, key, tuple);
    ctx->in_groupby_key = false;
    --ctx->nested;
    call_end(ctx);
    if (prc) { sdqh_table_free(ctx, tb); return prc; }
    *out = tb;
    return SDQH_OK;
}

int sdqh_table_share_groups(sdqh_ctx* ctx, sdqh_table* table, int nfields, const int32_t* fields, const int64_t* lo, const int64_t* span) {
    if (!ctx || !table || !fields || !lo || !span) return fail(ctx, SDQH_ERR_INVALID, "table_share_groups: bad arguments");
    if (!table->accumulate || table->bitmap_only) return fail(ctx, SDQH_ERR_INVALID, "table_share_groups: the table carries no accumulators");
    if (nfields < 1 || nfields > table->npay) return fail(ctx, SDQH_ERR_INVALID, "table_share_groups: 1..npayload fields");
    DevShare sh; std::memset(&sh, 0, sizeof(sh));
    sh.n = nfields;
    uint64_t cells = 1;
    for (int i = 0; i < nfields; ++i) {
        if (fields[i] < 0 || fields[i] >= table->npay) return fail(ctx, SDQH_ERR_INVALID, "table_share_groups: no such payload field");
        if (span[i] < 1) return fail(ctx, SDQH_ERR_INVALID, "table_share_groups: empty value range");
        if ((uint64_t)span[i] > (uint64_t)SDQH_MAX_SHARE_CELLS || cells * (uint64_t)span[i] > (uint64_t)SDQH_MAX_SHARE_CELLS)
            return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_share_groups: the fields' value rectangle exceeds SDQH_MAX_SHARE_CELLS");
        cells *= (uint64_t)span[i];
        sh.col[i] = table->stage.pay[fields[i]]; sh.lo[i] = lo[i]; sh.span[i] = span[i];
    }
    (void)hipSetDevice(ctx->device);
    call_begin(ctx);
    if (int rc = ensure_index(ctx, table)) return rc;
    const DevStage& st = table->stage;
    const uint64_t stage_rows = (uint64_t)st.nseg * (uint64_t)st.seg_rows;
    uint32_t* first = static_cast<uint32_t*>(pool_alloc(ctx, cells * 4 + 64));
    uint32_t* alias = static_cast<uint32_t*>(table_alloc(ctx, table, stage_rows * 4 + 64));
    if (!first || !alias) { if (first) pool_free(ctx, first); return fail(ctx, SDQH_ERR_NOMEM, "table_share_groups: out of device memory"); }
    { FillList fl; fl.add(first, (cells * 4 + 15) & ~(uint64_t)15, 0xFF); launch_fill(ctx, fl); }
    const unsigned grid = (unsigned)((st.nseg + TPB / WAVE - 1) / (TPB / WAVE));
    LAUNCH(ctx, "k_share_groups", k_share_groups<true>, grid, table->dev, st, sh, first, alias);
    LAUNCH(ctx, "k_share_groups", k_share_groups<false>, grid, table->dev, st, sh, first, alias);
    table->dev.alias = alias;
    table->compact_valid = false;
    call_end(ctx);
    hipError_t e = hipGetLastError();
    pool_free(ctx, first);                               // stream order: later users of the block run after the two kernels
    if (e != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string("table_share_groups launch: ") + hipGetErrorString(e));
    return SDQH_OK;
}

int sdqh_table_select_keys(sdqh_ctx* ctx, const sdqh_table* ctable, int64_t min_hits, int value_index, double lo, double hi, sdqh_table** out) {
    sdqh_table* table = const_cast<sdqh_table*>(ctable);
    if (!ctx || !table || !out || value_index < 0 || value_index >= SDQH_TUPLE_MAX_VALUES) return fail(ctx, SDQH_ERR_INVALID, "table_select_keys: bad arguments");
    if (!table->accumulate || table->bitmap_only) return fail(ctx, SDQH_ERR_INVALID, "table_select_keys: the table carries no accumulators");
    // (a key bitmap over [bm_lo, bm_hi] — the direct layout — or the dense layout's array over the same range)
    const bool dense_layout = !table->bm && table->dev.dense_arr && table->dev.bm_hi >= table->dev.bm_lo;
    if ((!table->bm && !dense_layout) || table->dev.bm_shift != 0 || table->dev.lin_rb != 0) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_select_keys: the table's keys have no dense range");
    (void)hipSetDevice(ctx->device);
    call_begin(ctx);
    if (int rc = ensure_index(ctx, table)) return rc;
    sdqh_table* tb = new sdqh_table();
    tb->bitmap_only = true; tb->index_built = true; tb->nrows_build = table->nrows_build;
    tb->nwords = dense_layout ? ((uint64_t)(table->dev.bm_hi - table->dev.bm_lo) + 32) / 32 : table->nwords;
    tb->bm = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64));
    tb->hdr = static_cast<TableHeader*>(table_alloc(ctx, tb, sizeof(TableHeader)));
    if (!tb->bm || !tb->hdr) { table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_NOMEM, "table_select_keys: out of device memory"); }
    tb->dev.bm = tb->bm; tb->dev.bm_lo = table->dev.bm_lo; tb->dev.bm_hi = table->dev.bm_hi; tb->dev.bitmap_only = 1; tb->dev.hdr = tb->hdr;
    { FillList fl; fl.add(tb->bm, (tb->nwords * 4 + 15) & ~(uint64_t)15, 0); fl.add(tb->hdr, sizeof(TableHeader), 0); launch_fill(ctx, fl); }
    const uint32_t mh = (uint32_t)std::min<int64_t>(std::max<int64_t>(min_hits, 0), 0xFFFFFFFFll);
    const unsigned seg_grid = (unsigned)((table->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE));
    LAUNCH(ctx, "k_select_keys", k_select_keys, seg_grid, table->dev, table->stage, mh, value_index, lo, hi, table->dev.bm_lo, tb->bm);
    call_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { table_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_DEVICE, std::string("table_select_keys launch: ") + hipGetErrorString(e)); }
    *out = tb;
    return SDQH_OK;
}

// ---- K-F ---------------------------------------------------------------------------------------
// Runs the compaction into device buffers (cached on the table) and returns the row count.
struct HostDest { int64_t* keys = nullptr; int64_t* payload = nullptr; double* values = nullptr; int64_t* hits = nullptr; int64_t capacity = 0; };
// The stream result copies are queued on: of the LOWEST priority the device offers — the copy is a blit kernel here, and a compute kernel
// of the next query that starts beside it should not be the one that waits ("side_priority" = 0: default priority).
static hipStream_t make_copy_stream(sdqh_ctx* ctx) {
    hipStream_t s = nullptr;
    int least = 0, greatest = 0;
    if (ctx->opt_side_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest &&
        hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) == hipSuccess) return s;
    (void)hipGetLastError();
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return s;
}
static bool in_host_block(sdqh_ctx* ctx, const void* p, size_t bytes) {
    const char* c = static_cast<const char*>(p);
    auto it = ctx->host_blocks.upper_bound(c);
    if (it == ctx->host_blocks.begin()) return false;
    --it;
    return c >= it->first && c + bytes <= it->first + it->second;
}
}  // extern "C" (reopened below)
namespace sdqh_host { bool host_block_contains(sdqh_ctx* ctx, const void* p, size_t bytes) { return in_host_block(ctx, p, bytes); } }
extern "C" {
// Runs the compaction and returns the row count in table->compact_n.  Without `dest` the rows go to
// device buffers cached on the table; with a device-visible `dest` that is large enough the write
// kernel stores them in the caller's arrays (*direct = true) and nothing is cached.
static int run_compact(sdqh_ctx* ctx, sdqh_table* table, int64_t min_hits, const HostDest* dest = nullptr, bool* direct = nullptr) {
    if (direct) *direct = false;
    if (table->compact_valid && table->compact_min_hits == min_hits) return SDQH_OK;
    DevCompactOut& o = table->compact;
    const size_t rows = (size_t)std::max<int64_t>(table->nrows_build, 1) + 1;
    if (!o.keys) {
        o.keys = static_cast<int64_t*>(table_alloc(ctx, table, rows * 8));
        o.hits = static_cast<int64_t*>(table_alloc(ctx, table, rows * 8));
        o.counter = static_cast<unsigned long long*>(table_alloc(ctx, table, 64));
        bool ok = o.keys && o.hits && o.counter;
        for (int p = 0; p < table->npay && ok; ++p) { o.pay[p] = static_cast<int64_t*>(table_alloc(ctx, table, rows * 8)); ok = o.pay[p] != nullptr; }
        if (table->accumulate) for (int k = 0; k < SDQH_TUPLE_MAX_VALUES && ok; ++k) { o.val[k] = static_cast<double*>(table_alloc(ctx, table, rows * 8)); ok = o.val[k] != nullptr; }
        if (!ok) return fail(ctx, SDQH_ERR_NOMEM, "table_compact: out of device memory");
    }
    o.npay = table->npay; o.nval = table->accumulate ? table->nv : 0;
    if (!ctx->count_host && hipHostMalloc(&ctx->count_host, 256, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->count_host = nullptr; }
    o.h_counter = static_cast<unsigned long long*>(ctx->count_host);
    o.direct = 0; o.host_rows = 0;
    o.h_keys = nullptr; o.h_hits = nullptr;
    for (int p = 0; p < SDQH_MAX_PAYLOAD; ++p) o.h_pay[p] = nullptr;
    for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) o.h_val[k] = nullptr;
    if (dest && dest->capacity > 0 && o.h_counter) {
        const size_t cb = (size_t)dest->capacity * 8;
        bool ok = (dest->keys || dest->payload || dest->values || dest->hits);
        if (dest->keys) ok = ok && in_host_block(ctx, dest->keys, cb);
        if (dest->payload) ok = ok && in_host_block(ctx, dest->payload, cb * (size_t)std::max(1, table->npay));
        if (dest->values) ok = ok && in_host_block(ctx, dest->values, cb * SDQH_TUPLE_MAX_VALUES);
        if (dest->hits) ok = ok && in_host_block(ctx, dest->hits, cb);
        if (ok) {
            o.direct = 1; o.host_rows = (uint64_t)dest->capacity;
            o.h_keys = dest->keys; o.h_hits = dest->hits;
            if (dest->payload) for (int p = 0; p < table->npay; ++p) o.h_pay[p] = dest->payload + (size_t)p * (size_t)dest->capacity;
            if (dest->values) for (int k = 0; k < o.nval; ++k) o.h_val[k] = dest->values + (size_t)k * (size_t)dest->capacity;
        }
    }
    call_begin(ctx);
    if (int rc = ensure_index(ctx, table)) return rc;
    uint32_t mh = (uint32_t)std::min<int64_t>(std::max<int64_t>(min_hits, 0), 0xFFFFFFFFll);
    const unsigned seg_grid = (unsigned)((table->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE));
    if (!table->seg_kept) { table->seg_kept = static_cast<uint32_t*>(table_alloc(ctx, table, (size_t)table->stage.nseg * 4 + 64)); if (!table->seg_kept) return fail(ctx, SDQH_ERR_NOMEM, "table_compact: out of device memory"); }
    LAUNCH(ctx, "k_compact_count", k_compact_count, seg_grid, table->dev, table->stage, o, mh, table->seg_kept);
    LAUNCH(ctx, "k_compact_scan", k_compact_scan, 1, table->seg_kept, table->stage.nseg, o.counter);
    LAUNCH(ctx, "k_compact_write", k_compact_write, seg_grid, table->dev, table->stage, o, mh, table->seg_kept);
    call_end(ctx);
    if (o.h_counter) {
        if (int rc = sync_stream(ctx)) return rc;
        table->compact_n = (int64_t)*o.h_counter;
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->result_host, o.counter, 8, hipMemcpyDeviceToHost, ctx->stream));
        if (int rc = sync_stream(ctx)) return rc;
        table->compact_n = (int64_t)*static_cast<const unsigned long long*>(ctx->result_host);
    }
    const bool went_direct = o.direct && (uint64_t)table->compact_n <= o.host_rows;
    table->compact_min_hits = min_hits; table->compact_valid = !went_direct;     // the device buffers hold the rows unless they went to the caller
    if (direct) *direct = went_direct;
    return SDQH_OK;
}

}  // extern "C" (reopened below)
namespace sdqh_host {
int table_compact_resident(sdqh_ctx* ctx, sdqh_table* tb, int64_t min_hits, bool want_zero_rows, int64_t* n) {
    (void)hipSetDevice(ctx->device);
    if (want_zero_rows && !tb->zero_rows) {
        const size_t bytes = ((size_t)std::max<int64_t>(tb->nrows_build, 1) + 1) * 8;
        tb->zero_rows = table_alloc(ctx, tb, bytes);
        if (!tb->zero_rows) return fail(ctx, SDQH_ERR_NOMEM, "table_columns: out of device memory");
        HIP_TRY(ctx, hipMemsetAsync(tb->zero_rows, 0, bytes, ctx->stream));
    }
    if (int rc = run_compact(ctx, tb, min_hits)) return rc;
    *n = tb->compact_n;
    return SDQH_OK;
}
}  // namespace sdqh_host
extern "C" {

int sdqh_table_compact(sdqh_ctx* ctx, const sdqh_table* ctable, int64_t min_hits, int64_t capacity,
                       int64_t* out_keys, int64_t* out_payload, double* out_values, int64_t* out_hits, int64_t* out_n) {
    sdqh_table* table = const_cast<sdqh_table*>(ctable);
    if (!ctx || !table || !out_n || capacity < 0) return fail(ctx, SDQH_ERR_INVALID, "table_compact: bad arguments");
    if (table->bitmap_only) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_compact: bitmap-only table");
    (void)hipSetDevice(ctx->device);
    HostDest dest; dest.keys = out_keys; dest.payload = out_payload; dest.values = out_values; dest.hits = out_hits; dest.capacity = capacity;
    bool direct = false;
    if (int rc = run_compact(ctx, table, min_hits, &dest, &direct)) return rc;
    const int64_t n = table->compact_n;
    *out_n = n;
    if (!out_keys && !out_payload && !out_values && !out_hits) return SDQH_OK;        // count-only call
    if (n > capacity) return fail(ctx, SDQH_ERR_OVERFLOW, "table_compact: capacity too small");
    if (n == 0) return SDQH_OK;
    const DevCompactOut& o = table->compact;
    const size_t nb = (size_t)n * 8;
    if (direct) {                                                                      // the kernel already wrote the rows
        if (out_values) for (int k = o.nval; k < SDQH_TUPLE_MAX_VALUES; ++k) std::memset(out_values + (size_t)k * (size_t)capacity, 0, nb);
        return SDQH_OK;
    }
    // D2H lands in pinned memory (a copy into pageable numpy memory is several times slower and
    // serialises inside the runtime), then one memcpy per array into the caller's buffers.
    const int nv = table->accumulate ? table->nv : 0;
    const int narr = (out_keys ? 1 : 0) + (out_payload ? table->npay : 0) + (out_values ? nv : 0) + (out_hits ? 1 : 0);
    const size_t need = nb * (size_t)narr;
    if (need > ctx->bulk_bytes && need <= ((size_t)1 << 30)) {
        if (ctx->bulk_host) (void)hipHostFree(ctx->bulk_host);
        ctx->bulk_host = nullptr; ctx->bulk_bytes = 0;
        size_t want = std::max<size_t>(need * 2, (size_t)8 << 20);
        if (hipHostMalloc(&ctx->bulk_host, want, hipHostMallocDefault) == hipSuccess) ctx->bulk_bytes = want; else (void)hipGetLastError();
    }
    const bool pinned = need <= ctx->bulk_bytes;
    char* land = static_cast<char*>(ctx->bulk_host);
    std::vector<std::pair<void*, const void*>> copies;       // (destination, pinned source)
    auto fetch = [&](void* dst, const void* dev) -> int {
        if (pinned) { HIP_TRY(ctx, hipMemcpyAsync(land, dev, nb, hipMemcpyDeviceToHost, ctx->stream)); copies.push_back({dst, land}); land += nb; }
        else HIP_TRY(ctx, hipMemcpyAsync(dst, dev, nb, hipMemcpyDeviceToHost, ctx->stream));
        return SDQH_OK;
    };
    if (out_keys) if (int rc = fetch(out_keys, o.keys)) return rc;
    if (out_payload) for (int p = 0; p < table->npay; ++p) if (int rc = fetch(out_payload + (size_t)p * (size_t)capacity, o.pay[p])) return rc;
    if (out_values) for (int k = 0; k < nv; ++k) if (int rc = fetch(out_values + (size_t)k * (size_t)capacity, o.val[k])) return rc;
    if (out_values) for (int k = nv; k < SDQH_TUPLE_MAX_VALUES; ++k) std::memset(out_values + (size_t)k * (size_t)capacity, 0, (size_t)n * 8);
    if (out_hits) if (int rc = fetch(out_hits, o.hits)) return rc;
    if (int rc = sync_stream(ctx)) return rc;
    for (auto& c : copies) std::memcpy(c.first, c.second, nb);
    return SDQH_OK;
}

// K-F with the rows delivered behind the call.  count -> write (each wave sums the counts before its segment) into one of two
// device staging buffers laid out like the caller's arrays; the stream is synchronised for the row count only; the rows leave
// by ONE device-to-host copy queued on a stream of its own, so the next call's kernels run beside it (writing 3.6 MB of Q3's
// result over PCIe from the kernel itself held the stream for 80 us of the query's 400).
int sdqh_table_compact_async(sdqh_ctx* ctx, const sdqh_table* ctable, int64_t min_hits, int64_t capacity,
                             int64_t* out_keys, int64_t* out_payload, double* out_values, int64_t* out_hits, int64_t* out_n) {
    sdqh_table* table = const_cast<sdqh_table*>(ctable);
    if (!ctx || !table || !out_n || capacity < 1 || !out_keys) return fail(ctx, SDQH_ERR_INVALID, "table_compact_async: bad arguments");
    if (table->bitmap_only) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_compact_async: bitmap-only table");
    (void)hipSetDevice(ctx->device);
    const int npay = out_payload ? table->npay : 0, nval = (out_values && table->accumulate) ? table->nv : 0;
    // the caller's arrays: keys | payload[npay] | values[TUPLE_MAX] | hits, each `capacity` rows, in ONE host block, in this order
    const size_t cb = (size_t)capacity * 8;
    char* base = reinterpret_cast<char*>(out_keys);
    const int narr = 1 + (out_payload ? table->npay : 0) + (out_values ? SDQH_TUPLE_MAX_VALUES : 0) + (out_hits ? 1 : 0);
    bool contiguous = in_host_block(ctx, base, cb * (size_t)narr);
    size_t at = cb;
    if (out_payload) { contiguous = contiguous && reinterpret_cast<char*>(out_payload) == base + at; at += cb * (size_t)table->npay; }
    if (out_values) { contiguous = contiguous && reinterpret_cast<char*>(out_values) == base + at; at += cb * SDQH_TUPLE_MAX_VALUES; }
    if (out_hits) { contiguous = contiguous && reinterpret_cast<char*>(out_hits) == base + at; at += cb; }
    if (!ctx->opt_async_result || !contiguous)                           // not the layout this path copies in one piece: the synchronous call
        return sdqh_table_compact(ctx, ctable, min_hits, capacity, out_keys, out_payload, out_values, out_hits, out_n);
    if (!ctx->side[1]) ctx->side[1] = make_copy_stream(ctx);
    if (!ctx->count_host && hipHostMalloc(&ctx->count_host, 256, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->count_host = nullptr; }
    const int b = ctx->rs_cur;
    if (!ctx->rs_copied[b] && hipEventCreateWithFlags(&ctx->rs_copied[b], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->rs_copied[b] = nullptr; }
    if (!ctx->side[1] || !ctx->count_host || !ctx->rs_copied[b]) return sdqh_table_compact(ctx, ctable, min_hits, capacity, out_keys, out_payload, out_values, out_hits, out_n);
    const size_t need = cb * (size_t)narr;
    if (ctx->rs_bytes[b] < need) {
        if (ctx->rs_used[b]) HIP_TRY(ctx, hipEventSynchronize(ctx->rs_copied[b]));
        if (ctx->rs_dev[b]) (void)hipFree(ctx->rs_dev[b]);
        ctx->rs_dev[b] = nullptr; ctx->rs_bytes[b] = 0;
        const size_t want = std::max<size_t>(need + need / 4, (size_t)4 << 20);
        if (hipMalloc(&ctx->rs_dev[b], want) != hipSuccess) { (void)hipGetLastError(); return sdqh_table_compact(ctx, ctable, min_hits, capacity, out_keys, out_payload, out_values, out_hits, out_n); }
        ctx->rs_bytes[b] = want;
    }
    call_begin(ctx);
    if (int rc = ensure_index(ctx, table)) return rc;
    if (ctx->rs_used[b]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->rs_copied[b], 0));       // the buffer's last copy has left it
    DevCompactOut o; std::memset(&o, 0, sizeof(o));
    char* dev = static_cast<char*>(ctx->rs_dev[b]);
    size_t off = 0;
    o.keys = reinterpret_cast<int64_t*>(dev); off += cb;
    for (int p = 0; p < (out_payload ? table->npay : 0); ++p) { o.pay[p] = reinterpret_cast<int64_t*>(dev + off); off += cb; }
    if (out_values) { for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) { if (k < nval) o.val[k] = reinterpret_cast<double*>(dev + off); off += cb; } }
    if (out_hits) { o.hits = reinterpret_cast<int64_t*>(dev + off); off += cb; }
    o.npay = npay; o.nval = nval;
    o.counter = reinterpret_cast<unsigned long long*>(static_cast<char*>(ctx->count_host) + 64);      // (device-visible pinned word: the kernel's own copy of the total)
    o.h_counter = static_cast<unsigned long long*>(ctx->count_host);
    o.host_rows = (uint64_t)capacity; o.bounded = 1;
    const uint32_t mh = (uint32_t)std::min<int64_t>(std::max<int64_t>(min_hits, 0), 0xFFFFFFFFll);
    const unsigned seg_grid = (unsigned)((table->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE));
    if (!table->seg_kept) { table->seg_kept = static_cast<uint32_t*>(table_alloc(ctx, table, (size_t)table->stage.nseg * 4 + 64)); if (!table->seg_kept) return fail(ctx, SDQH_ERR_NOMEM, "table_compact_async: out of device memory"); }
    table->compact_valid = false;
    LAUNCH(ctx, "k_compact_count", k_compact_count, seg_grid, table->dev, table->stage, o, mh, table->seg_kept);
    LAUNCH(ctx, "k_compact_write2", k_compact_write2, seg_grid, table->dev, table->stage, o, mh, table->seg_kept);
    call_end(ctx);
    if (int rc = sync_stream(ctx)) return rc;
    const int64_t n = (int64_t)*o.h_counter;
    *out_n = n;
    if (n > capacity) return fail(ctx, SDQH_ERR_OVERFLOW, "table_compact_async: capacity too small");
    if (n == 0) return SDQH_OK;
    // two copies at most: keys .. the last USED value array as one piece (the unused tails of the arrays in between ride along),
    // then the hit counts; value slots the tuple does not use are not part of either and are zeroed here
    const int lead = 1 + (out_payload ? table->npay : 0) + nval;
    HIP_TRY(ctx, hipMemcpyAsync(base, dev, cb * (size_t)(lead - 1) + (size_t)n * 8, hipMemcpyDeviceToHost, ctx->side[1]));
    if (out_hits) HIP_TRY(ctx, hipMemcpyAsync(out_hits, o.hits, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->side[1]));
    HIP_TRY(ctx, hipEventRecord(ctx->rs_copied[b], ctx->side[1]));
    ctx->rs_used[b] = true; ctx->rs_pending = true; ctx->rs_cur = b ^ 1;
    if (out_values) for (int k = nval; k < SDQH_TUPLE_MAX_VALUES; ++k) std::memset(out_values + (size_t)k * (size_t)capacity, 0, (size_t)n * 8);
    return SDQH_OK;
}

// (sdqh_table_compact_deferred — K-F with nothing waited for — lives in sdqh_aux.hip: it launches the two kernels through launch_compact_pair below)
int sdqh_table_topk(sdqh_ctx* ctx, const sdqh_table* ctable, int64_t min_hits, int k, int nsort, const sdqh_sort_key* sort,
                    int64_t* out_keys, int64_t* out_payload, double* out_values, int64_t* out_hits, int64_t* out_n) {
    sdqh_table* table = const_cast<sdqh_table*>(ctable);
    if (!ctx || !table || !out_n || !sort || nsort < 1 || nsort > SDQH_MAX_SORT_KEYS) return fail(ctx, SDQH_ERR_INVALID, "table_topk: bad arguments");
    if (k < 1 || k > SDQH_MAX_TOPK) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_topk: k must be 1..SDQH_MAX_TOPK");
    if (table->bitmap_only) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_topk: bitmap-only table");
    (void)hipSetDevice(ctx->device);
    DevTopSpec spec; std::memset(&spec, 0, sizeof(spec));
    spec.nsort = nsort; spec.k = k;
    const int nv = table->accumulate ? table->nv : 0;
    for (int i = 0; i < nsort; ++i) {
        const sdqh_sort_key& sk = sort[i];
        const bool ok = (sk.kind == SDQH_SORT_KEY) || (sk.kind == SDQH_SORT_PAYLOAD && sk.index >= 0 && sk.index < table->npay) ||
                        (sk.kind == SDQH_SORT_VALUE && sk.index >= 0 && sk.index < nv) || (sk.kind == SDQH_SORT_HITS && table->accumulate);
        if (!ok) return fail(ctx, SDQH_ERR_INVALID, "table_topk: sort key names a field the table does not have");
        spec.key[i].kind = sk.kind; spec.key[i].index = sk.index; spec.key[i].desc = sk.descending ? 1 : 0;
        spec.key[i].is_f64 = sk.kind == SDQH_SORT_VALUE ? 1 : (sk.kind == SDQH_SORT_PAYLOAD ? (sk.is_f64 ? 1 : 0) : 0);
    }
    call_begin(ctx);
    if (int rc = ensure_index(ctx, table)) return rc;
    // level 0 grid: enough workgroups to stream the stage, few enough that k padded candidates each stay small
    const unsigned g0 = (unsigned)std::max(1, std::min(table->stage.nseg, 4 * ctx->num_cu));
    const size_t cand = (size_t)g0 * (size_t)k;
    const size_t rec = 8 * 3 + 4;
    char* blob = static_cast<char*>(pool_alloc(ctx, 2 * (cand * rec + 64 * 4) + (size_t)k * 8 * 10 + 256));
    if (!blob) return fail(ctx, SDQH_ERR_NOMEM, "table_topk: out of device memory");
    auto carve = [&](char*& p, DevTopBuf& b) {
        b.k0 = reinterpret_cast<uint64_t*>(p); p += cand * 8; b.k1 = reinterpret_cast<uint64_t*>(p); p += cand * 8;
        b.k2 = reinterpret_cast<uint64_t*>(p); p += cand * 8; b.ref = reinterpret_cast<uint32_t*>(p); p += (cand * 4 + 63) / 64 * 64;
    };
    char* p = blob;
    DevTopBuf buf[2]; carve(p, buf[0]); carve(p, buf[1]);
    DevTopOut fin; std::memset(&fin, 0, sizeof(fin));
    int64_t* packed = reinterpret_cast<int64_t*>(p);                          // [count | keys k | hits k | pay 4k | val 4k]
    fin.count = reinterpret_cast<unsigned long long*>(packed);
    fin.keys = packed + 8; fin.hits = packed + 8 + k; fin.pay = packed + 8 + 2 * (size_t)k; fin.val = reinterpret_cast<double*>(packed + 8 + 6 * (size_t)k);
    fin.npay = table->npay; fin.nval = nv;
    const uint32_t mh = (uint32_t)std::min<int64_t>(std::max<int64_t>(min_hits, 0), 0xFFFFFFFFll);
    LAUNCH(ctx, "k_topk_scan", k_topk_scan, g0, table->dev, table->stage, spec, mh, buf[0], fin);
    int n_in = (int)cand, cur = 0;
    while (g0 > 1) {
        const unsigned g = (unsigned)((n_in + TOPK_CHUNK - 1) / TOPK_CHUNK);
        LAUNCH(ctx, "k_topk_reduce", k_topk_reduce, g, buf[cur], n_in, table->stage, spec, buf[cur ^ 1], fin);
        if (g == 1) break;
        n_in = (int)g * k; cur ^= 1;
    }
    call_end(ctx);
    const size_t packed_bytes = (8 + 10 * (size_t)k) * 8;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->result_host, packed, packed_bytes, hipMemcpyDeviceToHost, ctx->stream));
    int rc = sync_stream(ctx);
    pool_free(ctx, blob);
    if (rc) return rc;
    const int64_t* h = static_cast<const int64_t*>(ctx->result_host);
    const int64_t n = h[0];
    *out_n = n;
    const size_t nb = (size_t)n * 8;
    if (out_keys) std::memcpy(out_keys, h + 8, nb);
    if (out_hits) std::memcpy(out_hits, h + 8 + k, nb);
    if (out_payload) for (int q = 0; q < table->npay; ++q) std::memcpy(out_payload + (size_t)q * k, h + 8 + 2 * (size_t)k + (size_t)q * k, nb);
    if (out_values) for (int v = 0; v < SDQH_TUPLE_MAX_VALUES; ++v) std::memcpy(out_values + (size_t)v * k, h + 8 + 6 * (size_t)k + (size_t)v * k, nb);
    return SDQH_OK;
}

int sdqh_host_alloc(sdqh_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out || bytes == 0) return fail(ctx, SDQH_ERR_INVALID, "host_alloc: bad arguments");
    (void)hipSetDevice(ctx->device);
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, SDQH_ERR_NOMEM, "host_alloc: out of pinned host memory"); }
    ctx->host_blocks[static_cast<const char*>(p)] = bytes;
    *out = p;
    return SDQH_OK;
}
void sdqh_host_free(sdqh_ctx* ctx, void* block) {
    if (!block) return;
    if (ctx) ctx->host_blocks.erase(static_cast<const char*>(block));
    (void)hipHostFree(block);
}

int sdqh_table_entries(sdqh_ctx* ctx, const sdqh_table* ctable, sdqh_column** out_cols, int64_t* out_rows) {
    sdqh_table* table = const_cast<sdqh_table*>(ctable);
    if (!ctx || !table || !out_cols || !out_rows || table->bitmap_only) return fail(ctx, SDQH_ERR_INVALID, "table_entries: bad arguments");
    (void)hipSetDevice(ctx->device);
    if (int rc = run_compact(ctx, table, 0)) return rc;
    const int64_t n = table->compact_n;
    const DevCompactOut& o = table->compact;
    for (int c = 0; c < 1 + table->npay; ++c) {
        if (int rc = sdqh_column_alloc(ctx, n, SDQH_I64, 0, &out_cols[c])) return rc;
        if (n) HIP_TRY(ctx, hipMemcpyAsync(out_cols[c]->data, c == 0 ? o.keys : o.pay[c - 1], (size_t)n * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    *out_rows = n;
    return sync_stream(ctx);
}

// ---- generalised lookups (Q5 / Q9) ---------------------------------------------------------------
static int make_source(sdqh_ctx* ctx, const sdqh_source& s, int64_t nrows, int nl, const sdqh_lookup* lookups, int upto, const char* what, DevSource* d) {
    std::memset(d, 0, sizeof(*d));
    d->kind = s.kind;
    if (s.kind == SDQH_SRC_COLUMN) {
        if (!s.col || s.col->dtype == SDQH_STR || s.col->nrows < nrows) return fail(ctx, SDQH_ERR_INVALID, std::string(what) + ": column sources must be I64/F64 and cover nrows");
        d->col = static_cast<const int64_t*>(s.col->data);
        return SDQH_OK;
    }
    if (s.kind != SDQH_SRC_LOOKUP && s.kind != SDQH_SRC_LOOKUP_YEAR) return fail(ctx, SDQH_ERR_INVALID, std::string(what) + ": bad source kind");
    if (s.lookup < 0 || s.lookup >= upto || s.lookup >= nl) return fail(ctx, SDQH_ERR_INVALID, std::string(what) + ": source refers to a later or unknown lookup");
    const sdqh_table* t = lookups[s.lookup].table;
    if (t->bitmap_only || s.field < 0 || s.field >= t->npay) return fail(ctx, SDQH_ERR_INVALID, std::string(what) + ": no such payload field");
    d->lookup = s.lookup; d->field = s.field;
    // where the payload is read from, resolved here (the table's index exists: make_lookups ensured it) — see DevSource
    const bool direct = t->dev.dense_arr || (t->dev.bm && t->dev.bm_shift == 0) || t->dev.grp_first;
    if (t->dev.grp_first && t->dev.grp_kstride == 2 && s.field == 0) { d->pack = 3; d->col = t->dev.grp_key; return SDQH_OK; }
    if (t->dev.slots && !direct) {
        d->col = t->dev.slots;
        if (s.field < 2) d->pack = 1; else { d->pack = 2; d->col2 = t->dev.pay[s.field]; }
    } else { d->pack = 0; d->col = t->dev.pay[s.field]; }
    return SDQH_OK;
}

static int make_lookups(sdqh_ctx* ctx, int64_t nrows, int nl, const sdqh_lookup* lookups, DevLookups* L, bool* composite) {
    std::memset(L, 0, sizeof(*L));
    if (nl < 0 || nl > SDQH_MAX_LOOKUP || (nl && !lookups)) return fail(ctx, SDQH_ERR_INVALID, "too many lookups");
    for (int l = 0; l < nl; ++l) {
        if (!lookups[l].table || lookups[l].nkey < 1 || lookups[l].nkey > 2) return fail(ctx, SDQH_ERR_INVALID, "lookup: bad table / key arity");
        sdqh_table* t = const_cast<sdqh_table*>(lookups[l].table);
        if (int rc = ensure_index(ctx, t)) return rc;
        L->l[l].table = t->dev; L->l[l].nkey = lookups[l].nkey;
        if (lookups[l].nkey == 2) *composite = true;
        for (int k = 0; k < lookups[l].nkey; ++k) if (int rc = make_source(ctx, lookups[l].key[k], nrows, nl, lookups, l, "lookup key", &L->l[l].key[k])) return rc;
    }
    L->n = nl;
    return SDQH_OK;
}

int sdqh_build(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, int nlookups, const sdqh_lookup* lookups,
               int nkey, const sdqh_source* key, int npayload, const sdqh_source* payload, int accumulate, sdqh_table** out) {
    if (!ctx || nrows < 0 || !out || nkey < 1 || nkey > 2 || !key || npayload < 0 || npayload > SDQH_MAX_PAYLOAD || (npayload && !payload))
        return fail(ctx, SDQH_ERR_INVALID, "build: bad arguments");
    if (nrows >= 0xFFFFFFFEll) return fail(ctx, SDQH_ERR_UNSUPPORTED, "build: build side limited to 2^32-2 rows per GPU");
    (void)hipSetDevice(ctx->device);
    DevFilter f; DevLookups L; DevBuildSpec spec;
    std::memset(&spec, 0, sizeof(spec));
    bool composite = nkey == 2;
    if (int rc = make_filter(ctx, nrows, filter, nullptr, &f)) return rc;
    call_begin(ctx);                                   // the lookups may have to build their tables' indexes first
    if (int rc = make_lookups(ctx, nrows, nlookups, lookups, &L, &composite)) return rc;
    spec.nkey = nkey; spec.npay = npayload;
    for (int k = 0; k < nkey; ++k) if (int rc = make_source(ctx, key[k], nrows, nlookups, lookups, nlookups, "build key", &spec.key[k])) return rc;
    for (int p = 0; p < npayload; ++p) if (int rc = make_source(ctx, payload[p], nrows, nlookups, lookups, nlookups, "build payload", &spec.pay[p])) return rc;
    // bitmap over the key (single part: exact -> direct layout) or over its high part (composite:
    // a pre-filter in front of the hash layout), when that part is a plain column with a dense range
    int64_t lo = 0, hi = -1; bool want_bm = false;
    if (nrows > 0 && key[0].kind == SDQH_SRC_COLUMN && key[0].col->dtype == SDQH_I64) {
        if (int rc = ensure_minmax(ctx, const_cast<sdqh_column*>(key[0].col))) return rc;
        lo = key[0].col->mn; hi = key[0].col->mx;
        if (hi >= lo && lo > INT64_MIN / 2 && hi < INT64_MAX / 2) {
            const uint64_t range = (uint64_t)(hi - lo) + 1;
            want_bm = range <= (1ull << 31) && range <= 64ull * (uint64_t)std::max<int64_t>(nrows, 1024);
            if (nkey == 2 && (lo < 0 || hi > 0xFFFFFFFFll)) want_bm = false;
        }
    }
    // composite key whose two plain-column parts span a small rectangle: exact bitmap over the
    // linearised offset -> direct layout (no hash slots, no CAS) instead of hash + high-part pre-filter
    int64_t lin_rb = 0, lin_b0 = 0; uint64_t lin_bits = 0;
    if (want_bm && nkey == 2 && ctx->opt_direct_index && key[1].kind == SDQH_SRC_COLUMN && key[1].col->dtype == SDQH_I64) {
        if (int rc = ensure_minmax(ctx, const_cast<sdqh_column*>(key[1].col))) return rc;
        const int64_t blo = key[1].col->mn, bhi = key[1].col->mx;
        if (bhi >= blo && blo >= 0 && bhi <= 0xFFFFFFFFll) {
            const uint64_t ra = (uint64_t)(hi - lo) + 1, rb = (uint64_t)(bhi - blo) + 1;
            if (rb <= (1ull << 31) / ra && ra * rb <= 64ull * (uint64_t)std::max<int64_t>(nrows, 1024)) { lin_rb = (int64_t)rb; lin_b0 = blo; lin_bits = ra * rb; }
        }
    }
    // GROUPED layout (DevTable): a composite key whose first part is a plain column the table is stored in the order of (partsupp by part
    // key) — the entries of one first part are neighbours in the stage, the index is one stage row per first-part value, written while
    // staging: no hash slots, no clearing, no CAS insert (Q9's (part, supplier) table: 0.14 ms of staging + clear + insert -> the staging)
    const bool grouped = ctx->opt_grouped_index && ctx->opt_direct_index && want_bm && nkey == 2 && !lin_rb && nrows > 1 && nrows < ((int64_t)1 << 31) &&
                         hi < ((int64_t)1 << 31) && !key[0].col->transient && column_is_nondecreasing(ctx, const_cast<sdqh_column*>(key[0].col));
    sdqh_table* tb = new sdqh_table();
    tb->npay = npayload; tb->accumulate = accumulate != 0; tb->nrows_build = nrows;
    if (nkey == 1 && nrows > 0 && key[0].kind == SDQH_SRC_COLUMN && key[0].col->dtype == SDQH_I64 && column_is_increasing(ctx, const_cast<sdqh_column*>(key[0].col))) tb->keys_unique = true;
    // a composite key whose parts are plain columns that strictly increase as pairs (supplier by (s_suppkey, s_nationkey), partsupp by (ps_partkey,
    // ps_suppkey)): every staged row is an entry — the multi-GPU runner's device-sized replication may take the stage as it is
    if (nkey == 2 && nrows > 0 && key[0].kind == SDQH_SRC_COLUMN && key[1].kind == SDQH_SRC_COLUMN && key[0].col->nrows == nrows && key[1].col->nrows == nrows &&
        columns_pair_increasing(ctx, const_cast<sdqh_column*>(key[0].col), const_cast<sdqh_column*>(key[1].col))) tb->pack_unique = true;
    // stage without source columns: the kernel evaluates sources itself
    sdqh_column fake; fake.data = nullptr;
    const sdqh_column* fakes[SDQH_MAX_PAYLOAD] = {&fake, &fake, &fake, &fake};
    int rc = setup_stage(ctx, tb, nrows, &fake, npayload, fakes, nrows >= (int64_t)ctx->num_cu * 24 * 512 ? BUILD_LB : 1);   // whole steps for big tables; small ones keep one batch per wave (partial steps take the per-row path)
    uint64_t capmax = 1024;
    while (capmax < 2 * (uint64_t)std::max<int64_t>(nrows, 1)) capmax <<= 1;
    tb->capmax = capmax;
    int* flags = nullptr;
    if (!rc) {
        tb->hdr = static_cast<TableHeader*>(table_alloc(ctx, tb, sizeof(TableHeader)));
        flags = static_cast<int*>(table_alloc(ctx, tb, 64));
        if (want_bm && ctx->opt_direct_index) { tb->nwords = lin_rb ? (lin_bits + 31) / 32 : ((uint64_t)(hi - lo) + 32) / 32; tb->bm = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64)); }
        if (!tb->hdr || !flags || (tb->nwords && !tb->bm)) rc = fail(ctx, SDQH_ERR_NOMEM, "build: out of device memory");
    }
    uint32_t* grp_first = nullptr;
    const size_t grp_cells = grouped ? (size_t)(hi - lo) + 1 : 0;
    if (!rc && grouped && tb->bm) grp_first = static_cast<uint32_t*>(table_alloc(ctx, tb, grp_cells * 4 + 64));       // (no memory: the hash layout as before)
    if (!rc) {
        tb->dev.hdr = tb->hdr; tb->dev.shits = tb->stage.shits; tb->dev.sacc = tb->stage.sacc; tb->dev.acc_stride = tb->stage.acc_stride;
        if (grp_first) {
            tb->stage.grp_first = grp_first;
            tb->dev.grp_first = grp_first; tb->dev.grp_key = tb->stage.key; tb->dev.grp_kstride = 1; tb->dev.grp_seg_rows = tb->stage.seg_rows;
            if (npayload >= 1 && ctx->opt_grouped_pairs) {                  // (key, payload 0) side by side: a hit's first payload field comes with its key's line
                int64_t* kp = static_cast<int64_t*>(table_alloc(ctx, tb, ((size_t)nrows + 8) * 16 + 64));
                if (kp) { tb->stage.grp_kp = kp; tb->dev.grp_key = kp; tb->dev.grp_kstride = 2; }
            } tb->dev.grp_cap = nrows + 1;      // stage rows that can be read: the entries, and the end mark of a last segment that is all entries
        }
        const int shift = (nkey == 2 && !(tb->bm && lin_rb)) ? 32 : 0;
        tb->dev.bm = tb->bm; tb->dev.bm_lo = lo; tb->dev.bm_hi = hi; tb->dev.bitmap_only = 0; tb->dev.bm_shift = shift;
        for (int p = 0; p < npayload; ++p) tb->dev.pay[p] = tb->stage.pay[p];
        tb->stage.bm = tb->bm; tb->stage.bm_lo = lo; tb->stage.bm_hi = hi; tb->stage.hdr = tb->hdr; tb->stage.bm_shift = shift;
        if (tb->bm && lin_rb) { tb->dev.lin_rb = tb->stage.lin_rb = lin_rb; tb->dev.lin_b0 = tb->stage.lin_b0 = lin_b0; }
        const unsigned seg_grid = (unsigned)((tb->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE));
        // a tiny table (one workgroup of segments): the build kernel does its own fill (a launch less: each is ~8 us of dependent-launch latency)
        FillList fl; fl.add(tb->hdr, sizeof(TableHeader), 0); fl.add(flags, 8, 0); if (tb->bm) fl.add(tb->bm, tb->nwords * 4, 0); prefill_refs(ctx, tb, &fl);
        if (grp_first) fl.add(grp_first, grp_cells * 4, 0xFF);
        DevFill pre; std::memset(&pre, 0, sizeof(pre));
        prune_clean(ctx, &fl);
        if (seg_grid == 1 && ctx->opt_fuse_small && fl.most <= ((uint64_t)1 << 20) && fl.f.n <= FILL_MAX) pre = fl.pre(); else launch_fill(ctx, fl);
        // ... and its index too (k_index_small's work behind the staging, in the same workgroup): a launch less per tiny table
        DevIndexInline ix; std::memset(&ix, 0, sizeof(ix));
        if (seg_grid == 1 && ctx->opt_fuse_small && ctx->opt_index_inline && tb->bm && shift == 0 && tb->refs_prefilled && (tb->nwords + RANK_BLOCK_WORDS - 1) / RANK_BLOCK_WORDS == 1) {
            uint32_t* wprefix = static_cast<uint32_t*>(table_alloc(ctx, tb, tb->nwords * 4 + 64));
            if (wprefix) {
                tb->dev.wprefix = wprefix;
                ix.t = tb->dev; ix.wprefix = wprefix; ix.span = tb->span; ix.nwords = tb->nwords; ix.on = 1;
            }
        }
        hipError_t e;
        with_scan_filter(f, [&](auto FC) {
            using FCT = decltype(FC);
            if constexpr (std::is_same_v<FCT, FCfg<1, 0, 0, 0>> || std::is_same_v<FCT, FCfg<0, 0, 0, 0>>) {
                // the first lookup's streamed key and the first integer predicate through their narrow twins
                const bool eager0 = nlookups > 0 && lookups[0].key[0].kind == SDQH_SRC_COLUMN && lookups[0].key[0].col->dtype == SDQH_I64;
                if (ctx->opt_narrow && nrows >= ctx->opt_feature_min_rows && (eager0 || f.ni == 1)) {
                    const int32_t* nkey0 = eager0 ? static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(lookups[0].key[0].col))) : nullptr;
                    const int32_t* npred0 = f.ni == 1 ? static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(filter->ipred[0].col))) : nullptr;
                    if ((!eager0 || nkey0) && (f.ni != 1 || npred0)) {
                        auto kern = k_build_lookup<FCT, true>;
                        LAUNCH(ctx, "k_build_lookup", kern, seg_grid, f, L, spec, tb->stage, nrows, flags, nkey0, npred0, pre, ix);
                        return SDQH_OK;
                    }
                }
            }
            auto kern = k_build_lookup<FCT>;
            LAUNCH(ctx, "k_build_lookup", kern, seg_grid, f, L, spec, tb->stage, nrows, flags, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), pre, ix);
            return SDQH_OK;
        });
        if (ix.on) { if (tb->span) tb->dev.dense_arr = tb->span; tb->index_built = true; }      // (as ensure_index leaves a tiny table)
        call_end(ctx);
        e = hipGetLastError();
        if (e != hipSuccess) rc = fail(ctx, SDQH_ERR_DEVICE, std::string("build launch: ") + hipGetErrorString(e));
        // a key part outside [0, 2^32) cannot be packed: reported now — unless every packed part is a plain column whose minimum and
        // maximum say it cannot happen (then the host does not wait for this build: 35-45 us of an idle device inside Q5 and Q9)
        auto packs = [&](const sdqh_source& src) {
            if (src.kind != SDQH_SRC_COLUMN || !src.col || src.col->dtype != SDQH_I64) return false;
            sdqh_column* c = const_cast<sdqh_column*>(src.col);
            return ensure_minmax(ctx, c) == SDQH_OK && c->nrows > 0 && c->mn >= 0 && c->mx <= 0xFFFFFFFFll;
        };
        bool proven = composite;
        if (composite && !rc) {
            if (nkey == 2) proven = proven && packs(key[0]) && packs(key[1]);
            for (int l = 0; l < nlookups && proven; ++l) if (lookups[l].nkey == 2) proven = packs(lookups[l].key[0]) && packs(lookups[l].key[1]);
        }
        if (!rc && composite && !proven) {
            e = hipMemcpyAsync(ctx->result_host, flags, 4, hipMemcpyDeviceToHost, ctx->stream);
            if (e != hipSuccess) rc = fail(ctx, SDQH_ERR_DEVICE, hipGetErrorString(e));
            if (!rc) rc = sync_stream(ctx);
            if (!rc && (*static_cast<const int*>(ctx->result_host) & 2)) rc = fail(ctx, SDQH_ERR_UNSUPPORTED, "build: composite key part outside [0, 2^32)");
        }
    }
    if (rc) { table_release(ctx, tb); delete tb; return rc; }
    *out = tb;
    return SDQH_OK;
}

// The loop and its merge, launched: the groups land in `block` (an xgroupby block: pinned host memory with its completion word, or —
// device_block — device memory for sdqh_xgroupby_fold) or, without one, in the context's pinned result block.  Nothing is waited for.
static int lookup_aggregate_launch(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, int nlookups, const sdqh_lookup* lookups,
                                   int nkeys, const sdqh_source* keys, int tuple_shape, const sdqh_source* operands, void* block, bool device_block) {
    (void)hipSetDevice(ctx->device);
    const int nops = tuple_nops(tuple_shape), nv = tuple_nv(tuple_shape);
    if (nops < 0) return fail(ctx, SDQH_ERR_UNSUPPORTED, "unknown tuple shape");
    DevFilter f; DevLookups L; DevAggSpec spec;
    std::memset(&spec, 0, sizeof(spec));
    bool composite = false;
    if (int rc = make_filter(ctx, nrows, filter, nullptr, &f)) return rc;
    call_begin(ctx);
    if (int rc = make_lookups(ctx, nrows, nlookups, lookups, &L, &composite)) return rc;
    spec.nkeys = nkeys; spec.shape = tuple_shape;
    for (int k = 0; k < nkeys; ++k) if (int rc = make_source(ctx, keys[k], nrows, nlookups, lookups, nlookups, "group key", &spec.key[k])) return rc;
    for (int j = 0; j < nops; ++j) if (int rc = make_source(ctx, operands[j], nrows, nlookups, lookups, nlookups, "tuple operand", &spec.op[j])) return rc;
    L.debug = ctx->opt_lookup_debug;
    // software-pipelined streaming part (k_lookup_agg): pays where the first lookup's key column is clustered
    L.pipeline = 0;
    if (nlookups > 0 && lookups[0].key[0].kind == SDQH_SRC_COLUMN && lookups[0].key[0].col->dtype == SDQH_I64)
        L.pipeline = ctx->opt_lookup_pipeline >= 0 ? ctx->opt_lookup_pipeline : (column_is_clustered(ctx, const_cast<sdqh_column*>(lookups[0].key[0].col)) ? 1 : 0);
    // Row pack: every plain column the drain gathers (lookup key parts, group key parts, operands), interleaved once
    // and kept resident.  Only worth it for big scans with several gathered columns.
    // CLUSTERED (sdqh_aux.hip, cluster_pack_build): when the first lookup's key column comes in no row order, the scan has no predicate
    // (nothing else is streamed by row) and every gathered column is in the pack, the pack is built in the stable order of that key and
    // the loop runs over PACK rows: its streamed key is the twin in that order (clustered: bitmap words shared by neighbouring rows,
    // keys a step ahead, no coarse filter), survivors are runs of neighbouring pack rows, the first table is walked front to back.
    const int32_t* cluster_key32 = nullptr;
    size_t coarse_lds = 0;
    if (ctx->opt_row_pack && nlookups > 0 && nrows > 0 && nrows >= ctx->opt_feature_min_rows) {
        std::vector<DevSource*> srcs;
        for (int l = 0; l < nlookups; ++l) for (int k = 0; k < L.l[l].nkey; ++k) srcs.push_back(&L.l[l].key[k]);
        for (int k = 0; k < nkeys; ++k) srcs.push_back(&spec.key[k]);
        for (int j = 0; j < nops; ++j) srcs.push_back(&spec.op[j]);
        std::vector<const void*> cols;
        for (DevSource* sr : srcs) if (sr->kind == SDQH_SRC_COLUMN && std::find(cols.begin(), cols.end(), (const void*)sr->col) == cols.end()) cols.push_back(sr->col);
        if (cols.size() >= 5 && cols.size() <= (size_t)MAX_PACK && (size_t)nrows * cols.size() * 8 <= ((size_t)48 << 30)) {
            const int k = (int)((cols.size() + 1) & ~(size_t)1);               // even: rows stay 16-byte aligned
            // clustered?  (decided from facts about the key column that do not change from run to run)
            const void* order_col = nullptr;
            sdqh_column* kc0 = (lookups[0].key[0].kind == SDQH_SRC_COLUMN && lookups[0].key[0].col->dtype == SDQH_I64) ? const_cast<sdqh_column*>(lookups[0].key[0].col) : nullptr;
            const sdqh_table* t0 = lookups[0].table;
            const bool unfiltered = f.nc == 0 && f.ns == 0 && f.nf == 0 && f.ni == 0;
            const bool part_bm = t0->bm && !t0->dev.lin_rb && (lookups[0].nkey == 1 ? t0->dev.bm_shift == 0 : t0->dev.bm_shift != 0);
            if (ctx->opt_cluster_pack && ctx->opt_narrow && kc0 && unfiltered && part_bm && !kc0->transient && nrows < ((int64_t)1 << 32) &&
                (nrows >= 4 * ctx->opt_feature_min_rows || ctx->opt_feature_min_rows == 0) && (ctx->opt_cluster_pack == 2 || !column_is_clustered(ctx, kc0)) && ensure_minmax(ctx, kc0) == SDQH_OK)
                order_col = kc0->data;
            sdqh_ctx::RowPack* found = nullptr;
            for (auto& pk : ctx->packs) if (pk.cols == cols && pk.nrows == nrows && pk.order_col == order_col) found = &pk;
            if (!found && order_col && !ctx->capturing) {
                const int32_t* twin = static_cast<const int32_t*>(ensure_narrow(ctx, kc0));
                void* data = twin ? pool_alloc(ctx, (size_t)nrows * (size_t)k * 8 + 64) : nullptr;
                void* key32 = data ? pool_alloc(ctx, (size_t)nrows * 4 + 64) : nullptr;
                void* lb = key32 ? pool_alloc(ctx, ((size_t)(kc0->mx - kc0->mn) + 2) * 4 + 64) : nullptr;      // (none: the loop streams the ordered keys)
                if (key32 && cluster_pack_build(ctx, twin, kc0->mn, kc0->mx, nrows, cols.data(), (int)cols.size(), k, data, key32, lb) == SDQH_OK) {
                    sdqh_ctx::RowPack pk{cols, nrows, k, data}; pk.order_col = order_col; pk.key32 = key32; pk.lb = lb; pk.key_lo = kc0->mn; pk.key_hi = kc0->mx;
                    ctx->packs.push_back(pk);
                    found = &ctx->packs.back();
                } else {                                                       // (no twin, no memory: the pack in row order as before)
                    (void)hipGetLastError();
                    if (lb) pool_free(ctx, lb);
                    if (key32) pool_free(ctx, key32);
                    if (data) pool_free(ctx, data);
                }
            }
            if (!found && order_col) { order_col = nullptr; for (auto& pk : ctx->packs) if (pk.cols == cols && pk.nrows == nrows && !pk.order_col) found = &pk; }
            if (!found) {
                void* data = pool_alloc(ctx, (size_t)nrows * (size_t)k * 8 + 64);
                if (data) {
                    DevPackCols pc; std::memset(&pc, 0, sizeof(pc));
                    for (size_t j = 0; j < cols.size(); ++j) pc.col[j] = static_cast<const int64_t*>(cols[j]);
                    pc.ncols = (int)cols.size(); pc.k = k;
                    LAUNCH(ctx, "k_interleave", k_interleave, (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 16)), pc, nrows, static_cast<int64_t*>(data));
                    ctx->packs.push_back({cols, nrows, k, data});
                    found = &ctx->packs.back();
                }
            }
            if (found) {
                L.pack = static_cast<const int64_t*>(found->data); L.pack_k = found->k;
                for (DevSource* sr : srcs) if (sr->kind == SDQH_SRC_COLUMN)
                    sr->pack = 1 + (int)(std::find(cols.begin(), cols.end(), (const void*)sr->col) - cols.begin());
                if (found->order_col) {
                    cluster_key32 = static_cast<const int32_t*>(found->key32);
                    if (ctx->opt_lookup_pipeline < 0) L.pipeline = 1;
                    // the loop walks the first table's bitmap and the pack's runs instead of streaming the ordered keys (k_lookup_agg: RUN WALK)
                    if (found->lb && ctx->opt_cluster_list && nrows < ((int64_t)1 << 31) && t0->dev.bm_hi >= t0->dev.bm_lo && (uint64_t)(t0->dev.bm_hi - t0->dev.bm_lo) < 0xFFFFFFE0ull) {
                        L.run_lb = static_cast<const uint32_t*>(found->lb); L.run_lo = found->key_lo; L.run_hi = found->key_hi;
                    }
                }
            }
        }
    }
    // Coarse key filter in LDS for the first lookup when its keys come in no order and its bitmap does not fit L1
    if (!cluster_key32 && ctx->opt_coarse_kb > 0 && nlookups > 0 && nrows >= 4 * ctx->opt_feature_min_rows && lookups[0].key[0].kind == SDQH_SRC_COLUMN && lookups[0].key[0].col->dtype == SDQH_I64) {
        sdqh_table* t0 = const_cast<sdqh_table*>(lookups[0].table);
        const bool part_bitmap = t0->bm && !t0->dev.lin_rb && (lookups[0].nkey == 1 ? t0->dev.bm_shift == 0 : t0->dev.bm_shift != 0);
        // last time's density of this key column's coarse filter (see sdqh_ctx::coarse_stat): a filter that passed more than half the rows is left out
        const void* kcol0 = lookups[0].key[0].col->data;
        bool dense_last_time = false;
        if (ctx->coarse_stat && ctx->coarse_stat_col == kcol0 && ctx->coarse_stat_build_rows == t0->nrows_build && ctx->coarse_stat_bits_pending > 0) {
            dense_last_time = ctx->coarse_stat[0] * 2 > ctx->coarse_stat_bits_pending;
            // (the same key column may meet another table next time: measure again every 16th call)
            if (dense_last_time && ++ctx->coarse_skipped >= 16) { dense_last_time = false; ctx->coarse_skipped = 0; }
        }
        if (part_bitmap && !dense_last_time && (t0->nwords * 4 > (32u << 10) || ctx->opt_feature_min_rows == 0) && !column_is_clustered(ctx, const_cast<sdqh_column*>(lookups[0].key[0].col))) {
            if (!t0->coarse) {
                const uint64_t nbits = t0->nwords * 32, budget = (uint64_t)ctx->opt_coarse_kb * 1024 * 8;
                int shift = 0;
                while ((nbits >> shift) > budget) ++shift;
                const int cwords = (int)(((nbits >> shift) + 32) / 32);
                uint32_t* c = static_cast<uint32_t*>(table_alloc(ctx, t0, (size_t)cwords * 4 + 64));
                if (c) {
                    if (!ctx->coarse_stat) {
                        void* hp = nullptr;
                        if (hipHostMalloc(&hp, 64, hipHostMallocDefault) == hipSuccess) { ctx->coarse_stat = static_cast<uint64_t*>(hp); ctx->coarse_stat[0] = ctx->coarse_stat[1] = 0; }
                        else (void)hipGetLastError();
                        ctx->coarse_count_dev = static_cast<unsigned long long*>(pool_alloc(ctx, 64));
                    }
                    unsigned long long* cnt = (ctx->coarse_stat && ctx->coarse_count_dev) ? ctx->coarse_count_dev : nullptr;
                    if (cnt) (void)hipMemsetAsync(cnt, 0, 8, ctx->stream);
                    LAUNCH(ctx, "k_coarsen", k_coarsen, (unsigned)std::min<int>((cwords + TPB - 1) / TPB, ctx->num_cu * 4), t0->bm, nbits, shift, c, cwords, cnt);
                    if (cnt) {                                             // lands in pinned memory behind the launch; read by the next call
                        (void)hipMemcpyAsync(ctx->coarse_stat, cnt, 8, hipMemcpyDeviceToHost, ctx->stream);
                        ctx->coarse_stat_col = kcol0; ctx->coarse_stat_build_rows = t0->nrows_build; ctx->coarse_skipped = 0;
                        ctx->coarse_stat_bits_pending = (uint64_t)cwords * 32;
                    }
                    t0->coarse = c; t0->coarse_words = cwords; t0->coarse_shift = shift;
                }
            }
            if (t0->coarse && t0->coarse_shift > 0) {
                L.coarse = t0->coarse; L.coarse_words = t0->coarse_words; L.coarse_shift = t0->coarse_shift; coarse_lds = (size_t)t0->coarse_words * 4;
                if (ctx->opt_lookup_pipeline < 0) L.pipeline = 1;         // behind the filter the L2 is no longer the bound: keys a step ahead pay again (0.452 -> 0.434 ms)
            }
        }
    }
    // result block in ctx->result_dev: gkeys[LG_SLOTS] | acc[LG_SLOTS][4] | cnt[LG_SLOTS] | flags
    char* rd = static_cast<char*>(ctx->result_dev);
    unsigned long long* r_keys = reinterpret_cast<unsigned long long*>(rd);
    double* r_acc = reinterpret_cast<double*>(rd + LG_SLOTS * 8);
    int64_t* r_cnt = reinterpret_cast<int64_t*>(rd + LG_SLOTS * 40);
    int* r_flags = reinterpret_cast<int*>(rd + LG_SLOTS * 48);
    static_assert(LG_SLOTS * 48 + 8 <= RESULT_BYTES, "result block too small");
    auto rd_clean_lg = [&]() {                               // the group slots and the flags as the last merge left them: no fill; either way they are about to be written
        const bool clean = ctx->opt_fill_ahead && ctx->rd_clean_ff >= (size_t)LG_SLOTS * 8 && ctx->rd_clean_zero_off == (int64_t)LG_SLOTS * 48;
        rd_dirty(ctx);
        return clean;
    };
    // the merge writes the result block straight into the pinned host block (same layout): no copy-engine launch after it
    char* hb = static_cast<char*>(block ? block : ctx->result_host);
    unsigned long long* h_keys = reinterpret_cast<unsigned long long*>(hb);
    double* h_acc = reinterpret_cast<double*>(hb + LG_SLOTS * 8);
    int64_t* h_cnt = reinterpret_cast<int64_t*>(hb + LG_SLOTS * 40);
    int* h_tail = reinterpret_cast<int*>(hb + LG_SLOTS * 48);
    (void)r_acc; (void)r_cnt;
    // a result block's DONE word (as sdqh_xgroupby_async): cleared now, written by the stream behind the merge
    uint32_t* done = (block && !device_block) ? reinterpret_cast<uint32_t*>(hb + sdqh_xgroupby_block_bytes() - 64) : nullptr;
    if (done) host_init(ctx, done, 0, 4);
    unsigned grid = 1; char* blob = nullptr;
    int lrc = with_shape(ctx, tuple_shape, [&](auto S) {
        return with_scan_filter(f, [&](auto FC) {
            using FCT = decltype(FC);
            constexpr int SH = decltype(S)::value;
            // With the coarse key filter: 1024-thread workgroups (one copy of the filter per 16 waves), 4 row pairs per lane.
            // Compiled for unfiltered scans only (the shape it exists for: Q9); any other loop runs without the filter.
            constexpr bool BIG_OK = std::is_same_v<FCT, FCfg<0, 0, 0, 0, 0>>;
            constexpr int BIG_BT = 1024, BIG_PU = 4;
            if (coarse_lds && !BIG_OK) { L.coarse = nullptr; L.coarse_words = 0; L.coarse_shift = 0; coarse_lds = 0; }
            // narrow twin of the first lookup's streamed key column (unfiltered scans: the instances that exist with NW)
            const int32_t* nkey0 = nullptr;
            if (BIG_OK && ctx->opt_narrow && nrows >= ctx->opt_feature_min_rows && nlookups > 0 && lookups[0].key[0].kind == SDQH_SRC_COLUMN && lookups[0].key[0].col->dtype == SDQH_I64)
                nkey0 = cluster_key32 ? cluster_key32 : static_cast<const int32_t*>(ensure_narrow(ctx, const_cast<sdqh_column*>(lookups[0].key[0].col)));
            if (cluster_key32 && !nkey0) return fail(ctx, SDQH_ERR_DEVICE, "lookup_aggregate: clustered pack without its key twin");
            if constexpr (BIG_OK) if (coarse_lds) {
                auto big_raw = k_lookup_agg<SH, FCT, BIG_BT, BIG_PU>;
                auto big_nw = k_lookup_agg<SH, FCT, BIG_BT, BIG_PU, true>;
                auto big = nkey0 ? big_nw : big_raw;                             // (same signature)
                int per_cu = 0;
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)coarse_lds) != hipSuccess ||
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, big, BIG_BT, coarse_lds) != hipSuccess || per_cu < 1) {
                    (void)hipGetLastError();
                    L.coarse = nullptr; L.coarse_words = 0; L.coarse_shift = 0; coarse_lds = 0;        // does not fit: run without the filter
                } else {
                    const int64_t tile_rows = (int64_t)BIG_BT * ROWS_PER_LOAD * BIG_PU * ctx->opt_probe_chunk;
                    grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + tile_rows - 1) / tile_rows, (int64_t)ctx->num_cu * per_cu));
                    if ((int64_t)grid * tile_rows >= ((int64_t)1 << 31)) return fail(ctx, SDQH_ERR_UNSUPPORTED, "lookup_aggregate: probe_chunk too large for this grid");
                    const size_t nslots = (size_t)grid * LG_SLOTS;
                    blob = static_cast<char*>(pool_alloc(ctx, nslots * 40 + 256));
                    if (!blob) return fail(ctx, SDQH_ERR_NOMEM, "lookup_aggregate: out of device memory");
                    double* pacc = reinterpret_cast<double*>(blob);
                    int64_t* pcnt = reinterpret_cast<int64_t*>(blob + nslots * 32);
                    if (!rd_clean_lg()) { FillList fl; fl.add(r_keys, LG_SLOTS * 8, 0xFF); fl.add(r_flags, 8, 0); launch_fill(ctx, fl); }
                    { KernelScope _ks(ctx, "k_lookup_agg"); hipLaunchKernelGGL(big, dim3(grid), dim3(BIG_BT), coarse_lds, ctx->stream, f, L, spec, nrows, r_keys, pacc, pcnt, r_flags, ctx->opt_probe_chunk, nkey0); }
                    LAUNCH(ctx, "k_groupby_merge", k_groupby_merge, LG_SLOTS, r_keys, pacc, pcnt, (int)grid, h_acc, h_cnt, h_keys, r_flags, h_tail, 1);
                    ctx->rd_clean_ff = (size_t)LG_SLOTS * 8; ctx->rd_clean_zero_off = (int64_t)LG_SLOTS * 48;
                    call_end(ctx);
                    return SDQH_OK;
                }
            }
            auto kern = k_lookup_agg<SH, FCT>;
            if constexpr (BIG_OK) { if (nkey0) kern = k_lookup_agg<SH, FCT, TPB, LOOKUP_PU, true>; }
            if constexpr (BIG_OK) { if (L.run_lb) kern = k_lookup_agg<SH, FCT, TPB, LOOKUP_PU, true, true>; }      // the run walk: the instance without the streaming part
            grid = stream_grid(ctx, kern, nrows, TPB * ROWS_PER_LOAD * LOOKUP_PU * ctx->opt_probe_chunk);
            // the kernel queues candidate rows as 32-bit offsets from its current chunk and rebases them by one grid stride
            if ((int64_t)grid * ctx->opt_probe_chunk * (TPB * ROWS_PER_LOAD * LOOKUP_PU) >= ((int64_t)1 << 31)) return fail(ctx, SDQH_ERR_UNSUPPORTED, "lookup_aggregate: probe_chunk too large for this grid");
            const size_t nslots = (size_t)grid * LG_SLOTS;
            blob = static_cast<char*>(pool_alloc(ctx, nslots * 40 + 256));
            if (!blob) return fail(ctx, SDQH_ERR_NOMEM, "lookup_aggregate: out of device memory");
            double* pacc = reinterpret_cast<double*>(blob);
            int64_t* pcnt = reinterpret_cast<int64_t*>(blob + nslots * 32);
            if (!rd_clean_lg()) { FillList fl; fl.add(r_keys, LG_SLOTS * 8, 0xFF); fl.add(r_flags, 8, 0); launch_fill(ctx, fl); }
            LAUNCH(ctx, "k_lookup_agg", kern, grid, f, L, spec, nrows, r_keys, pacc, pcnt, r_flags, ctx->opt_probe_chunk, nkey0);
            LAUNCH(ctx, "k_groupby_merge", k_groupby_merge, LG_SLOTS, r_keys, pacc, pcnt, (int)grid, h_acc, h_cnt, h_keys, r_flags, h_tail, 1);
                    ctx->rd_clean_ff = (size_t)LG_SLOTS * 8; ctx->rd_clean_zero_off = (int64_t)LG_SLOTS * 48;
            call_end(ctx);
            return SDQH_OK;
        });
    });
    if (blob) pool_free(ctx, blob);                          // (stream order: whoever gets the block next runs after the merge)
    if (lrc) return lrc;
    if (done && stream_store32(ctx, ctx->stream, done, 1) != SDQH_OK) *done = 2;     // 2: no marker, collect synchronises
    return SDQH_OK;
}

int sdqh_lookup_aggregate(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, int nlookups, const sdqh_lookup* lookups,
                          int nkeys, const sdqh_source* keys, int tuple_shape, const sdqh_source* operands, int max_groups,
                          int64_t* out_keys, double* out_values, int64_t* out_counts, int32_t* out_ngroups) {
    if (!ctx || nrows < 0 || nkeys < 1 || nkeys > SDQH_MAX_GROUPKEYS || !keys || max_groups < 1 || max_groups > SDQH_MAX_LOOKUP_GROUPS || !out_ngroups)
        return fail(ctx, SDQH_ERR_INVALID, "lookup_aggregate: bad arguments");
    const int nv = tuple_nv(tuple_shape);
    if (int rc = lookup_aggregate_launch(ctx, nrows, filter, nlookups, lookups, nkeys, keys, tuple_shape, operands, nullptr, false)) return rc;
    if (int rc = sync_stream(ctx)) return rc;
    const char* h = static_cast<const char*>(ctx->result_host);
    const int flags = *reinterpret_cast<const int*>(h + LG_SLOTS * 48);
    if (flags & 2) return fail(ctx, SDQH_ERR_UNSUPPORTED, "lookup_aggregate: key part out of range");
    const unsigned long long* hk = reinterpret_cast<const unsigned long long*>(h);
    const double* ha = reinterpret_cast<const double*>(h + LG_SLOTS * 8);
    const int64_t* hc = reinterpret_cast<const int64_t*>(h + LG_SLOTS * 40);
    std::vector<int> order;
    for (int g = 0; g < LG_SLOTS; ++g) if (hk[g] != EMPTY_GROUP && hc[g] > 0) order.push_back(g);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return hk[a] < hk[b]; });
    const int ng = (int)order.size();
    if ((flags & 1) || ng > max_groups) { *out_ngroups = std::max(ng, max_groups + 1); return fail(ctx, SDQH_ERR_OVERFLOW, "lookup_aggregate: more groups than max_groups"); }
    for (int i = 0; i < ng; ++i) {
        const int g = order[(size_t)i];
        if (out_keys) for (int k = 0; k < nkeys; ++k) out_keys[i * nkeys + k] = (int64_t)((hk[g] >> (32 * k)) & 0xFFFFFFFFull);
        if (out_values) for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) out_values[i * SDQH_TUPLE_MAX_VALUES + k] = k < nv ? ha[g * 4 + k] : 0.0;
        if (out_counts) out_counts[i] = hc[g];
    }
    *out_ngroups = ng;
    return SDQH_OK;
}

// ... with its groups left in a block instead of returned (ABI 7): launched, not waited for.
int sdqh_lookup_aggregate_block(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, int nlookups, const sdqh_lookup* lookups,
                                int nkeys, const sdqh_source* keys, int tuple_shape, const sdqh_source* operands, void* block, int device_block) {
    if (!ctx || nrows < 0 || nkeys < 1 || nkeys > SDQH_MAX_GROUPKEYS || !keys || !block) return fail(ctx, SDQH_ERR_INVALID, "lookup_aggregate_block: bad arguments");
    if (!device_block && !in_host_block(ctx, block, sdqh_xgroupby_block_bytes()))
        return fail(ctx, SDQH_ERR_INVALID, "lookup_aggregate_block: the result block must come from sdqh_host_alloc (sdqh_xgroupby_block_bytes() bytes)");
    return lookup_aggregate_launch(ctx, nrows, filter, nlookups, lookups, nkeys, keys, tuple_shape, operands, block, device_block != 0);
}

// ---- multi-GPU helpers -------------------------------------------------------------------------
int sdqh_scan_compact(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, int nprobes, const sdqh_probe* probes,
                      int ncols, const sdqh_column* const* cols, sdqh_column** out_cols, int64_t* out_rows) {
    if (!ctx || nrows < 0 || ncols < 1 || ncols > SDQH_MAX_COMPACT_COLS || !cols || !out_cols || !out_rows)
        return fail(ctx, SDQH_ERR_INVALID, "scan_compact: bad arguments");
    (void)hipSetDevice(ctx->device);
    DevFilter f; DevProbes pr;
    if (int rc = make_filter(ctx, nrows, filter, nullptr, &f)) return rc;
    if (int rc = make_probes(ctx, nrows, nprobes, probes, &pr)) return rc;
    for (int c = 0; c < ncols; ++c) if (!cols[c] || cols[c]->nrows < nrows || cols[c]->dtype == SDQH_STR) return fail(ctx, SDQH_ERR_INVALID, "scan_compact: columns must be I64/F64 and cover nrows");
    // a throw-away stage: column 0 plays the key, the others the payload
    sdqh_table tmp;
    int rc = setup_stage(ctx, &tmp, nrows, cols[0], ncols - 1, cols + 1, STAGE_BATCH);
    uint64_t* seg_off = nullptr; unsigned long long* total = nullptr;
    sdqh_column* outs[SDQH_MAX_COMPACT_COLS] = {nullptr};
    if (!rc) {
        seg_off = static_cast<uint64_t*>(table_alloc(ctx, &tmp, (size_t)tmp.stage.nseg * 8 + 64));
        total = static_cast<unsigned long long*>(table_alloc(ctx, &tmp, 64));
        if (!seg_off || !total) rc = fail(ctx, SDQH_ERR_NOMEM, "scan_compact: out of device memory");
    }
    DevGather g; std::memset(&g, 0, sizeof(g)); g.ncols = ncols;
    for (int c = 0; c < ncols && !rc; ++c) {
        rc = sdqh_column_alloc(ctx, nrows, cols[c]->dtype, 0, &outs[c]);
        if (!rc) g.out[c] = static_cast<int64_t*>(outs[c]->data);
    }
    if (!rc) {
        const unsigned seg_grid = (unsigned)((tmp.stage.nseg + TPB / WAVE - 1) / (TPB / WAVE));
        call_begin(ctx);
        DevFill nofill_stage; std::memset(&nofill_stage, 0, sizeof(nofill_stage));
        with_stage_filter(f, nprobes, [&](auto FC) { auto kern = k_stage<decltype(FC), -1>; LAUNCH(ctx, "k_stage", kern, seg_grid, f, pr, tmp.stage, nrows, static_cast<const int32_t*>(nullptr), static_cast<const int32_t*>(nullptr), nofill_stage); return SDQH_OK; });
        LAUNCH(ctx, "k_seg_scan", k_seg_scan, 1, tmp.stage.seg_count, tmp.stage.nseg, seg_off, total);
        LAUNCH(ctx, "k_gather_segments", k_gather_segments, seg_grid, tmp.stage, seg_off, g);
        call_end(ctx);
        hipError_t e = hipMemcpyAsync(ctx->result_host, total, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, SDQH_ERR_DEVICE, hipGetErrorString(e));
        if (!rc) rc = sync_stream(ctx);
    }
    table_release(ctx, &tmp);
    if (rc) { for (int c = 0; c < ncols; ++c) if (outs[c]) sdqh_column_free(ctx, outs[c]); return rc; }
    const int64_t n = (int64_t)*static_cast<const unsigned long long*>(ctx->result_host);
    for (int c = 0; c < ncols; ++c) { outs[c]->nrows = n; out_cols[c] = outs[c]; }
    *out_rows = n;
    return SDQH_OK;
}

}  // extern "C" (reopened below)
namespace sdqh_host {
int stage_rows_out(sdqh_ctx* ctx, sdqh_table* tb, sdqh_column** out_cols, int64_t* out_rows) {
    const int ncols = 1 + tb->npay;
    if (ncols > SDQH_MAX_COMPACT_COLS || !tb->stage.seg_count || !tb->stage.key) return fail(ctx, SDQH_ERR_INVALID, "stage rows: not a staged table");
    uint64_t* seg_off = static_cast<uint64_t*>(tb_alloc(ctx, tb, (size_t)tb->stage.nseg * 8 + 64));
    unsigned long long* total = static_cast<unsigned long long*>(tb_alloc(ctx, tb, 64));
    if (!seg_off || !total) return fail(ctx, SDQH_ERR_NOMEM, "stage rows: out of device memory");
    // the survivors are counted first (the call waits for the total anyway) and the output columns sized by it: a probe side of which
    // half a per cent passes must not reserve 1 + npay columns of the SCANNED row count
    call_begin(ctx);
    LAUNCH(ctx, "k_seg_scan", k_seg_scan, 1, tb->stage.seg_count, tb->stage.nseg, seg_off, total);
    { hipError_t e = hipMemcpyAsync(ctx->result_host, total, 8, hipMemcpyDeviceToHost, ctx->stream);
      if (e != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, hipGetErrorString(e)); }
    if (int rc = sync_stream(ctx)) return rc;
    const int64_t n = (int64_t)*static_cast<const unsigned long long*>(ctx->result_host);
    sdqh_column* outs[SDQH_MAX_COMPACT_COLS] = {nullptr};
    DevGather g; std::memset(&g, 0, sizeof(g)); g.ncols = ncols;
    int rc = SDQH_OK;
    for (int c = 0; c < ncols && !rc; ++c) {
        rc = sdqh_column_alloc(ctx, n, SDQH_I64, 0, &outs[c]);
        if (!rc) { g.out[c] = static_cast<int64_t*>(outs[c]->data); sdqh_column_mark_transient(ctx, outs[c]); }
    }
    if (!rc && n > 0) {
        const unsigned seg_grid = (unsigned)((tb->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE));
        LAUNCH(ctx, "k_gather_segments", k_gather_segments, seg_grid, tb->stage, seg_off, g);
    }
    call_end(ctx);
    if (rc) { for (int c = 0; c < ncols; ++c) if (outs[c]) sdqh_column_free(ctx, outs[c]); return rc; }
    for (int c = 0; c < ncols; ++c) out_cols[c] = outs[c];
    *out_rows = n;
    return SDQH_OK;
}
}  // namespace sdqh_host
extern "C" {

int sdqh_partition_by_key(sdqh_ctx* ctx, int64_t nrows, const sdqh_column* key, int nparts, const int64_t* range_upper, int ncols,
                          const sdqh_column* const* cols, sdqh_column** out_cols, int64_t* counts) {
    if (!ctx || nrows < 0 || nparts < 1 || nparts > SDQH_MAX_PARTS || ncols < 1 || ncols > SDQH_MAX_COMPACT_COLS || !cols || !out_cols || !counts)
        return fail(ctx, SDQH_ERR_INVALID, "partition_by_key: bad arguments");
    (void)hipSetDevice(ctx->device);
    if (int rc = check_col(ctx, key, SDQH_I64, nrows, "partition key")) return rc;
    DevPartition pt; std::memset(&pt, 0, sizeof(pt)); pt.nparts = nparts; pt.by_range = range_upper ? 1 : 0;
    if (range_upper) for (int p = 0; p < nparts - 1; ++p) pt.upper[p] = range_upper[p];
    DevGather src, dst; std::memset(&src, 0, sizeof(src)); std::memset(&dst, 0, sizeof(dst)); src.ncols = dst.ncols = ncols;
    sdqh_column* outs[SDQH_MAX_COMPACT_COLS] = {nullptr};
    int rc = SDQH_OK;
    for (int c = 0; c < ncols && !rc; ++c) {
        if (!cols[c] || cols[c]->nrows < nrows || cols[c]->dtype == SDQH_STR) { rc = fail(ctx, SDQH_ERR_INVALID, "partition_by_key: columns must be I64/F64"); break; }
        rc = sdqh_column_alloc(ctx, nrows, cols[c]->dtype, 0, &outs[c]);
        if (!rc) { src.out[c] = static_cast<int64_t*>(cols[c]->data); dst.out[c] = static_cast<int64_t*>(outs[c]->data); }
    }
    unsigned long long* dcounts = static_cast<unsigned long long*>(pool_alloc(ctx, 2 * SDQH_MAX_PARTS * 8));
    if (!rc && !dcounts) rc = fail(ctx, SDQH_ERR_NOMEM, "partition_by_key: out of device memory");
    if (!rc) {
        unsigned long long* cursor = dcounts + SDQH_MAX_PARTS;
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 4));
        const int64_t* kc = static_cast<const int64_t*>(key->data);
        call_begin(ctx);
        hipError_t e = hipMemsetAsync(dcounts, 0, 2 * SDQH_MAX_PARTS * 8, ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, SDQH_ERR_DEVICE, hipGetErrorString(e));
        LAUNCH(ctx, "k_part_count", k_part_count, grid, kc, nrows, pt, dcounts);
        { KernelScope ks(ctx, "k_part_offsets"); hipLaunchKernelGGL(k_part_offsets, dim3(1), dim3(64), 0, ctx->stream, dcounts, nparts, cursor); }
        LAUNCH(ctx, "k_part_scatter", k_part_scatter, grid, kc, nrows, pt, cursor, src, dst);
        call_end(ctx);
        e = hipMemcpyAsync(ctx->result_host, dcounts, (size_t)nparts * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, SDQH_ERR_DEVICE, hipGetErrorString(e));
        if (!rc) rc = sync_stream(ctx);
    }
    pool_free(ctx, dcounts);
    if (rc) { for (int c = 0; c < ncols; ++c) if (outs[c]) sdqh_column_free(ctx, outs[c]); return rc; }
    for (int p = 0; p < nparts; ++p) counts[p] = (int64_t)static_cast<const unsigned long long*>(ctx->result_host)[p];
    for (int c = 0; c < ncols; ++c) out_cols[c] = outs[c];
    return SDQH_OK;
}

int sdqh_column_set_bounds(sdqh_ctx* ctx, sdqh_column* col, int64_t lo, int64_t hi) {
    if (!ctx || !col || col->dtype != SDQH_I64 || lo > hi) return fail(ctx, SDQH_ERR_INVALID, "column_set_bounds: bad arguments");
    col->mn = lo; col->mx = hi; col->have_minmax = true; col->minmax_pending = false;
    return SDQH_OK;
}
int sdqh_column_mark_transient(sdqh_ctx* ctx, sdqh_column* col) {
    if (!ctx || !col) return fail(ctx, SDQH_ERR_INVALID, "column_mark_transient: bad arguments");
    col->transient = true; col->narrow_state = 0; col->code_state = 0; col->clustered = 0; col->increasing = 0; col->span8 = 0;
    return SDQH_OK;
}

int sdqh_partition_pack(sdqh_ctx* ctx, int64_t nrows, const sdqh_column* key, int nparts, const int64_t* range_upper, int ncols,
                        const sdqh_column* const* cols, void* packed, int64_t* counts) {
    if (!ctx || nrows < 0 || nparts < 1 || nparts > SDQH_MAX_PARTS || ncols < 1 || ncols > SDQH_MAX_COMPACT_COLS || !cols || !counts || (nrows && !packed))
        return fail(ctx, SDQH_ERR_INVALID, "partition_pack: bad arguments");
    (void)hipSetDevice(ctx->device);
    if (int rc = check_col(ctx, key, SDQH_I64, nrows, "partition key")) return rc;
    DevPartition pt; std::memset(&pt, 0, sizeof(pt)); pt.nparts = nparts; pt.by_range = range_upper ? 1 : 0;
    if (range_upper) for (int p = 0; p < nparts - 1; ++p) pt.upper[p] = range_upper[p];
    DevGather src; std::memset(&src, 0, sizeof(src)); src.ncols = ncols;
    for (int c = 0; c < ncols; ++c) {
        if (!cols[c] || cols[c]->nrows < nrows || cols[c]->dtype == SDQH_STR) return fail(ctx, SDQH_ERR_INVALID, "partition_pack: columns must be I64/F64");
        src.out[c] = static_cast<int64_t*>(cols[c]->data);
    }
    unsigned long long* dcounts = static_cast<unsigned long long*>(pool_alloc(ctx, 3 * SDQH_MAX_PARTS * 8));
    if (!dcounts) return fail(ctx, SDQH_ERR_NOMEM, "partition_pack: out of device memory");
    unsigned long long *cursor = dcounts + SDQH_MAX_PARTS, *base = dcounts + 2 * SDQH_MAX_PARTS;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 4));
    const int64_t* kc = static_cast<const int64_t*>(key->data);
    int rc = SDQH_OK;
    call_begin(ctx);
    if (hipMemsetAsync(dcounts, 0, 3 * SDQH_MAX_PARTS * 8, ctx->stream) != hipSuccess) rc = fail(ctx, SDQH_ERR_DEVICE, "partition_pack: memset failed");
    if (!rc) {
        LAUNCH(ctx, "k_part_count", k_part_count, grid, kc, nrows, pt, dcounts);
        { KernelScope ks(ctx, "k_part_offsets"); hipLaunchKernelGGL(k_part_offsets, dim3(1), dim3(64), 0, ctx->stream, dcounts, nparts, cursor); }
        if (hipMemcpyAsync(base, cursor, (size_t)nparts * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) rc = fail(ctx, SDQH_ERR_DEVICE, "partition_pack: copy failed");
    }
    if (!rc && nrows > 0) LAUNCH(ctx, "k_part_scatter_packed", k_part_scatter_packed, grid, kc, nrows, pt, cursor, dcounts, base, src, static_cast<int64_t*>(packed));
    call_end(ctx);
    if (!rc && hipMemcpyAsync(ctx->result_host, dcounts, (size_t)nparts * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = fail(ctx, SDQH_ERR_DEVICE, "partition_pack: copy failed");
    if (!rc) rc = sync_stream(ctx);                       // the counts are what the caller sizes the exchange with
    pool_free(ctx, dcounts);
    if (rc) return rc;
    for (int p = 0; p < nparts; ++p) counts[p] = (int64_t)static_cast<const unsigned long long*>(ctx->result_host)[p];
    return SDQH_OK;
}

int sdqh_unpack_parts(sdqh_ctx* ctx, const void* packed, int nparts, const int64_t* part_rows, int ncols, const int* dtypes, sdqh_column** out_cols, int64_t* out_rows) {
    if (!ctx || nparts < 1 || nparts > SDQH_MAX_PARTS || !part_rows || ncols < 1 || ncols > SDQH_MAX_COMPACT_COLS || !dtypes || !out_cols || !out_rows)
        return fail(ctx, SDQH_ERR_INVALID, "unpack_parts: bad arguments");
    (void)hipSetDevice(ctx->device);
    DevUnpack u; std::memset(&u, 0, sizeof(u)); u.nparts = nparts; u.ncols = ncols;
    int64_t total = 0, most = 0;
    for (int s = 0; s < nparts; ++s) {
        if (part_rows[s] < 0) return fail(ctx, SDQH_ERR_INVALID, "unpack_parts: negative row count");
        u.rows[s] = part_rows[s]; u.dst_off[s] = total; u.src_off[s] = total * ncols; total += part_rows[s]; most = std::max(most, part_rows[s]);
    }
    if (total && !packed) return fail(ctx, SDQH_ERR_INVALID, "unpack_parts: no buffer");
    DevGather dst; std::memset(&dst, 0, sizeof(dst)); dst.ncols = ncols;
    sdqh_column* outs[SDQH_MAX_COMPACT_COLS] = {nullptr};
    for (int c = 0; c < ncols; ++c) {
        if (dtypes[c] != SDQH_I64 && dtypes[c] != SDQH_F64) { for (int j = 0; j < c; ++j) sdqh_column_free(ctx, outs[j]); return fail(ctx, SDQH_ERR_INVALID, "unpack_parts: columns are I64 / F64"); }
        if (int rc = sdqh_column_alloc(ctx, total, dtypes[c], 0, &outs[c])) { for (int j = 0; j < c; ++j) sdqh_column_free(ctx, outs[j]); return rc; }
        dst.out[c] = static_cast<int64_t*>(outs[c]->data);
        // rows that live for one run of a plan: no twins, no dictionaries, no order facts are gathered for them (each would be a pass
        // over the column — or a host round trip — paid on every run)
        outs[c]->transient = true; outs[c]->narrow_state = 0; outs[c]->code_state = 0; outs[c]->clustered = 0; outs[c]->increasing = 0; outs[c]->span8 = 0;
    }
    if (total > 0) {
        const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((most + TPB - 1) / TPB, 256));
        KernelScope ks(ctx, "k_unpack_parts");
        hipLaunchKernelGGL(k_unpack_parts, dim3(gx, (unsigned)(nparts * ncols)), dim3(TPB), 0, ctx->stream, static_cast<const int64_t*>(packed), u, dst);
    }
    for (int c = 0; c < ncols; ++c) out_cols[c] = outs[c];
    *out_rows = total;
    return ctx->opt_async_copies ? SDQH_OK : sync_stream(ctx);
}

int sdqh_column_copy_out(sdqh_ctx* ctx, const sdqh_column* col, int64_t row0, int64_t nrows, void* dst) {
    if (!ctx || !col || col->dtype == SDQH_STR || row0 < 0 || nrows < 0 || row0 + nrows > col->nrows || (nrows && !dst)) return fail(ctx, SDQH_ERR_INVALID, "column_copy_out: bad arguments");
    if (nrows == 0) return SDQH_OK;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipMemcpyAsync(dst, static_cast<const char*>(col->data) + (size_t)row0 * 8, (size_t)nrows * 8, hipMemcpyDefault, ctx->stream));      // (the caller's buffer: device memory, or pinned host memory — the multi-GPU runner's hybrid mode)
    return ctx->opt_async_copies ? SDQH_OK : sync_stream(ctx);
}
int sdqh_column_copy_in(sdqh_ctx* ctx, sdqh_column* col, int64_t row0, int64_t nrows, const void* src) {
    if (!ctx || !col || col->dtype == SDQH_STR || row0 < 0 || nrows < 0 || row0 + nrows > col->nrows || (nrows && !src)) return fail(ctx, SDQH_ERR_INVALID, "column_copy_in: bad arguments");
    if (nrows == 0) return SDQH_OK;
    (void)hipSetDevice(ctx->device);
    HIP_TRY(ctx, hipMemcpyAsync(static_cast<char*>(col->data) + (size_t)row0 * 8, src, (size_t)nrows * 8, hipMemcpyDefault, ctx->stream));
    // the contents changed: what was learnt about them is void.  A transient column (rows that live for one run: sdqh_column_mark_transient)
    // stays "no order, no twins" — unknown (-1) would make the next kernel that asks pay a pass and a host round trip, every run
    const int unknown = col->transient ? 0 : -1;
    col->have_minmax = false; col->minmax_pending = false; col->clustered = unknown; col->increasing = unknown; col->nondecreasing = unknown; col->span8 = unknown;
    if (col->narrow) { attach_free(ctx, col, col->narrow); col->narrow = nullptr; }
    col->narrow_state = unknown;
    if (col->run_index) { attach_free(ctx, col, col->run_index); col->run_index = nullptr; }
    col->run_index_state = -1;
    if (col->delta8) { attach_free(ctx, col, col->delta8); col->delta8 = nullptr; }
    col->delta8_state = -1;
    column_codes_release(ctx, col);
    if (col->transient) col->code_state = 0;
    return ctx->opt_async_copies ? SDQH_OK : sync_stream(ctx);
}

int sdqh_table_export_bitmap(sdqh_ctx* ctx, const sdqh_table* table, int64_t lo, int64_t hi, sdqh_column** out_words) {
    if (!ctx || !table || !out_words || hi < lo) return fail(ctx, SDQH_ERR_INVALID, "table_export_bitmap: bad arguments");
    (void)hipSetDevice(ctx->device);
    const uint64_t bits = (uint64_t)(hi - lo) + 1;
    if (bits > (1ull << 34)) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_export_bitmap: key range too wide");
    const int64_t words32 = (int64_t)((bits + 31) / 32), words64 = (words32 + 1) / 2;
    if (*out_words) { if ((*out_words)->dtype != SDQH_I64 || (*out_words)->nrows < words64) return fail(ctx, SDQH_ERR_INVALID, "table_export_bitmap: destination column too short"); }
    else if (int rc = sdqh_column_alloc(ctx, words64, SDQH_I64, 0, out_words)) return rc;
    // a table that has an exact bitmap of its own (a key set; the direct layout over a plain key): shifted word copies, no pass over the entries
    const bool own = table->bm && table->dev.bm_shift == 0 && table->dev.lin_rb == 0 && table->dev.bm_hi >= table->dev.bm_lo;
    if (own) {
        const int64_t nw = words64 * 2;
        LAUNCH(ctx, "k_export_bits", k_export_bits, (unsigned)std::max<int64_t>(1, std::min<int64_t>((nw + TPB - 1) / TPB, (int64_t)ctx->num_cu * 8)),
               table->bm, table->dev.bm_lo, table->dev.bm_hi, lo, hi, static_cast<uint32_t*>((*out_words)->data), nw);
    } else {
        if (table->bitmap_only) return fail(ctx, SDQH_ERR_INVALID, "table_export_bitmap: the table has neither entries nor a bitmap over its key");
        HIP_TRY(ctx, hipMemsetAsync((*out_words)->data, 0, (size_t)words64 * 8, ctx->stream));
        LAUNCH(ctx, "k_export_bitmap", k_export_bitmap, (unsigned)((table->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE)), table->stage, lo, hi, static_cast<uint32_t*>((*out_words)->data));
    }
    return ctx->opt_async_copies ? SDQH_OK : sync_stream(ctx);      // ("async_copies": queued; whatever reads the words next is ordered behind it on the stream)
}

int sdqh_column_unpack2(sdqh_ctx* ctx, const sdqh_column* packed, int64_t nrows, sdqh_column** out_hi, sdqh_column** out_lo) {
    if (!ctx || !packed || !out_hi || !out_lo || nrows < 0 || nrows > packed->nrows || packed->dtype != SDQH_I64) return fail(ctx, SDQH_ERR_INVALID, "column_unpack2: bad arguments");
    (void)hipSetDevice(ctx->device);
    sdqh_column *h = nullptr, *l = nullptr;
    if (int rc = sdqh_column_alloc(ctx, nrows, SDQH_I64, 0, &h)) return rc;
    if (int rc = sdqh_column_alloc(ctx, nrows, SDQH_I64, 0, &l)) { sdqh_column_free(ctx, h); return rc; }
    if (nrows > 0)
        LAUNCH(ctx, "k_unpack2", k_unpack2, (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 8)),
               static_cast<const int64_t*>(packed->data), nrows, static_cast<int64_t*>(h->data), static_cast<int64_t*>(l->data));
    *out_hi = h; *out_lo = l;
    return SDQH_OK;
}

int sdqh_table_from_bitmap(sdqh_ctx* ctx, const sdqh_column* words, int64_t lo, int64_t hi, sdqh_table** out) {
    if (!ctx || !words || !out || hi < lo || words->dtype != SDQH_I64) return fail(ctx, SDQH_ERR_INVALID, "table_from_bitmap: bad arguments");
    const uint64_t bits = (uint64_t)(hi - lo) + 1;
    if ((uint64_t)words->nrows * 64 < bits) return fail(ctx, SDQH_ERR_INVALID, "table_from_bitmap: bitmap too short");
    sdqh_table* tb = new sdqh_table();
    tb->bitmap_only = true; tb->index_built = true;
    tb->dev.bm = static_cast<const uint32_t*>(words->data); tb->dev.bm_lo = lo; tb->dev.bm_hi = hi; tb->dev.bitmap_only = 1;
    tb->bm = static_cast<uint32_t*>(words->data);          // borrowed: the caller keeps `words` alive
    tb->nwords = (bits + 31) / 32;
    *out = tb;
    return SDQH_OK;
}

}  // extern "C"

// ---- what sdqh_x.hip (row programs, run-time specialised kernels) uses from this translation unit -------
namespace sdqh_host {

void* tb_alloc(sdqh_ctx* ctx, sdqh_table* t, size_t bytes) { return ::table_alloc(ctx, t, bytes); }
void tb_release(sdqh_ctx* ctx, sdqh_table* t) { ::table_release(ctx, t); }
int index_ensure(sdqh_ctx* ctx, sdqh_table* tb) { return ::ensure_index(ctx, tb); }
int stage_setup_computed(sdqh_ctx* ctx, sdqh_table* tb, int64_t nrows, int npay, int batch) {
    sdqh_column fake; fake.data = nullptr;
    const sdqh_column* fakes[SDQH_MAX_PAYLOAD] = {&fake, &fake, &fake, &fake};
    return ::setup_stage(ctx, tb, nrows, &fake, npay, fakes, batch);
}
void fill_regions(sdqh_ctx* ctx, void* const* ptr, const size_t* bytes, const unsigned char* byte, int n) {
    FillList fl;
    for (int i = 0; i < n; ++i) fl.add(ptr[i], bytes[i], byte[i]);
    launch_fill(ctx, fl);
}
int prefill_direct_refs(sdqh_ctx* ctx, sdqh_table* tb, void** ptr, size_t* bytes) {     // up to 2 regions, all to be filled with 0xFF
    FillList fl;
    ::prefill_refs(ctx, tb, &fl);
    for (int i = 0; i < fl.f.n; ++i) { ptr[i] = fl.f.p[i]; bytes[i] = (size_t)fl.f.bytes[i]; }
    return fl.f.n;
}
void launch_sum_partials(sdqh_ctx* ctx, const double* partial, int nparts, double* out) {
    LAUNCH(ctx, "k_sum_partials", k_sum_partials, 1, partial, nparts, out);
}
void launch_groupby_merge_lg(sdqh_ctx* ctx, const unsigned long long* gkeys, const double* pacc, const int64_t* pcnt, int nparts, double* out_acc, int64_t* out_cnt) {
    LAUNCH(ctx, "k_groupby_merge", k_groupby_merge, LG_SLOTS, const_cast<unsigned long long*>(gkeys), pacc, pcnt, nparts, out_acc, out_cnt, static_cast<unsigned long long*>(nullptr), static_cast<int*>(nullptr), static_cast<int*>(nullptr), 0);
}
// The LG result block (group slots | sums | counts | flags in ctx->result_dev): did the last merge leave the slots EMPTY and the
// flags zero (the caller then launches no fill)?  Either way the caller is about to write the block: the claim is consumed.
bool rd_take_clean_lg(sdqh_ctx* ctx) {
    const bool clean = ctx->opt_fill_ahead && ctx->rd_clean_ff >= (size_t)LG_SLOTS * 8 && ctx->rd_clean_zero_off == (int64_t)LG_SLOTS * 48;
    rd_dirty(ctx);
    return clean;
}
// Fold the workgroups' partials straight into the pinned host block (same layout as the device block: no copy-engine launch
// after it), resetting the device block's group slots and flags for the next call.
void launch_groupby_merge_lg_host(sdqh_ctx* ctx, unsigned long long* r_keys, const double* pacc, const int64_t* pcnt, int nparts, int* r_flags, void* host_block) {
    char* hb = static_cast<char*>(host_block ? host_block : ctx->result_host);
    LAUNCH(ctx, "k_groupby_merge", k_groupby_merge, LG_SLOTS, r_keys, pacc, pcnt, nparts, reinterpret_cast<double*>(hb + LG_SLOTS * 8), reinterpret_cast<int64_t*>(hb + LG_SLOTS * 40),
           reinterpret_cast<unsigned long long*>(hb), r_flags, reinterpret_cast<int*>(hb + LG_SLOTS * 48), 1);
    ctx->rd_clean_ff = (size_t)LG_SLOTS * 8; ctx->rd_clean_zero_off = (int64_t)LG_SLOTS * 48;
}
// K-F's count + write pair into `o` (sdqh_aux.hip: sdqh_table_compact_deferred lays `o` out over a staging buffer of its own)
int launch_compact_pair(sdqh_ctx* ctx, sdqh_table* table, const DevCompactOut& o, uint32_t min_hits) {
    const unsigned seg_grid = (unsigned)((table->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE));
    if (!table->seg_kept) { table->seg_kept = static_cast<uint32_t*>(::table_alloc(ctx, table, (size_t)table->stage.nseg * 4 + 64)); if (!table->seg_kept) return fail(ctx, SDQH_ERR_NOMEM, "table_compact: out of device memory"); }
    table->compact_valid = false;
    LAUNCH(ctx, "k_compact_count", k_compact_count, seg_grid, table->dev, table->stage, o, min_hits, table->seg_kept);
    LAUNCH(ctx, "k_compact_write2", k_compact_write2, seg_grid, table->dev, table->stage, o, min_hits, table->seg_kept);
    return SDQH_OK;
}
hipStream_t copy_stream(sdqh_ctx* ctx) { if (!ctx->side[1]) ctx->side[1] = ::make_copy_stream(ctx); return ctx->side[1]; }
int column_minmax(sdqh_ctx* ctx, sdqh_column* c) { return ensure_minmax(ctx, c); }
bool column_increasing(sdqh_ctx* ctx, sdqh_column* c) { return ::column_is_increasing(ctx, c); }
bool column_nondecreasing(sdqh_ctx* ctx, sdqh_column* c) { return ::column_is_nondecreasing(ctx, c); }
const void* column_narrow(sdqh_ctx* ctx, sdqh_column* c) { return ensure_narrow(ctx, c); }
int new_owned_column(sdqh_ctx* ctx, int64_t nrows, int dtype, sdqh_column** out) { return sdqh_column_alloc(ctx, nrows, dtype, 0, out); }

}  // namespace sdqh_host
