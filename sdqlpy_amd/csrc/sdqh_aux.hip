// sdqh_aux.hip — the parts of the C ABI that keep the HOST out of a run (round 5):
//
//   sdqh_table_compact_deferred   K-F with nothing waited for (moved here from sdqh_hip.hip: it is the one ahead-of-time call a recorded
//                                 plan ends in, and this unit compiles in seconds)
//   sdqh_table_partition_pack /   a redistribution step of the partitioned join sized on the DEVICE: fixed-capacity chunks with their
//   sdqh_unpack_chunks            row counts in the chunk headers, so the counts travel with the data through one equal-split
//                                 all-to-all and the host never reads them (include/sdqh.h, ABI 5)
//   sdqh_graph_*                  a prepared plan's device calls recorded once into a hipGraph (stream capture) and replayed by one
//                                 call: the counterpart of the reference compiling a query into ONE function
//                                 (lib/sdql_ir_cpp_generator_par.py:839-890) — here the dozen calls of a query cost the host a third of
//                                 a step to issue
//
// Nothing here has a counterpart in the reference beyond those citations: it has one process, one address space, no device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>

#define SDQH_DECLS_ONLY 1            // argument structs and device helpers of the kernel header, not a second copy of its kernels
#include "sdqh_host.hpp"

using namespace sdqh_host;

#define HIP_TRYA(ctx, expr)                                                                             \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct sdqh_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    sdqh_ctx* ctx = nullptr;
    std::vector<sdqh_ctx::HostInit> inits;          // host words set again before every launch
    int nodes = 0;
};

namespace {

// one 32-bit word of device-visible host memory, stored by the stream itself: hipStreamWriteValue32 where the stream executes, a
// one-thread kernel where it is being recorded (the value-write has no graph node)
__global__ void k_store32(uint32_t* p, uint32_t v) { if (threadIdx.x == 0 && blockIdx.x == 0) { __atomic_store_n(p, v, __ATOMIC_RELEASE); } }

// ---- redistribution kernels ----------------------------------------------------------------------------------------------------------
constexpr int PP_SEGS = 4;                           // stage segments a workgroup of the partitioning kernel walks (ONE claim per part and workgroup)

// The staged rows of a table (DevStage: per-wave segments, seg_count[s] live rows each) scattered into nparts fixed-capacity chunks.
// A workgroup takes PP_SEGS segments; its four waves share every segment's rows (wave w: the 64-row batches w, w + 4, ... — a wave
// per segment left most of the chip idle: a stage has a few thousand segments of a few hundred live rows).  Pass 1 counts the
// workgroup's rows per part in LDS, one thread per part claims the workgroup's range in the chunk (the chunk's own header word is the
// cursor: it ends up holding every row MEANT for the chunk), pass 2 places the rows — rank inside the workgroup by a second LDS
// counter; the lanes of a wave that go to one part take consecutive places, so the stores are runs.
__global__ __launch_bounds__(TPB) void k_stage_part_pack(DevStage st, DevPartition pt, int ncols, int64_t chunk_rows, int64_t chunk_words, int64_t* __restrict__ packed) {
    __shared__ unsigned int s_hist[SDQH_MAX_PARTS];
    __shared__ unsigned int s_rank[SDQH_MAX_PARTS];
    __shared__ unsigned long long s_base[SDQH_MAX_PARTS];
    if (threadIdx.x < SDQH_MAX_PARTS) { s_hist[threadIdx.x] = 0; s_rank[threadIdx.x] = 0; }
    __syncthreads();
    const int wave = (int)(threadIdx.x / WAVE), lane = (int)(threadIdx.x & (WAVE - 1));
    const int seg0 = (int)blockIdx.x * PP_SEGS;
    uint32_t counts[PP_SEGS];
#pragma unroll
    for (int j = 0; j < PP_SEGS; ++j) counts[j] = seg0 + j < st.nseg ? st.seg_count[seg0 + j] : 0u;
    uint32_t most = 0;
#pragma unroll
    for (int j = 0; j < PP_SEGS; ++j) most = counts[j] > most ? counts[j] : most;
    for (int pass = 0; pass < 2; ++pass) {
        // round r: this wave's 64-row batch r of EVERY segment — the four key loads of a round are requested together (a round is one
        // memory round trip, not four: the kernel is a chain of round trips, a stage segment holds a few hundred live rows)
        for (uint32_t i0 = (uint32_t)wave * WAVE; i0 < most; i0 += TPB) {
            int64_t keys[PP_SEGS];
            bool lives[PP_SEGS];
#pragma unroll
            for (int j = 0; j < PP_SEGS; ++j) {
                lives[j] = i0 + lane < counts[j];
                keys[j] = lives[j] ? st.key[(int64_t)(seg0 + j) * st.seg_rows + i0 + lane] : 0;
            }
#pragma unroll
            for (int j = 0; j < PP_SEGS; ++j) {
                if (i0 >= counts[j]) continue;                            // (wave-uniform)
                const bool live = lives[j];
                const int64_t row = (int64_t)(seg0 + j) * st.seg_rows + i0 + lane;
                const int64_t key = keys[j];
                const int part = live ? part_of(pt, key) : -1;
                unsigned long long todo = __ballot(live);
                unsigned int place = 0;
                while (todo) {                                            // one round per distinct part among the wave's rows
                    const int leader = __builtin_ctzll(todo);
                    const int p = __shfl(part, leader, WAVE);
                    const unsigned long long mine = __ballot(live && part == p);
                    const unsigned int n = (unsigned int)__builtin_popcountll(mine);
                    unsigned int first = 0;
                    if (lane == leader) first = atomicAdd(pass == 0 ? &s_hist[p] : &s_rank[p], n);
                    first = (unsigned int)__shfl((int)first, leader, WAVE);
                    if (live && part == p) place = first + (unsigned int)__builtin_popcountll(mine & ((1ull << lane) - 1ull));
                    todo &= ~mine;
                }
                if (pass == 1 && live) {
                    const uint64_t at = s_base[part] + place;
                    if (at < (uint64_t)chunk_rows) {
                        int64_t* chunk = packed + (int64_t)part * chunk_words + 2;
                        chunk[at] = key;
#pragma unroll
                        for (int c = 1; c < SDQH_MAX_COMPACT_COLS; ++c) if (c < ncols) chunk[(int64_t)c * chunk_rows + at] = st.pay[c - 1][row];
                    }
                }
            }
        }
        if (pass == 0) {
            __syncthreads();
            if ((int)threadIdx.x < pt.nparts)
                s_base[threadIdx.x] = s_hist[threadIdx.x] ? atomicAdd(reinterpret_cast<unsigned long long*>(packed + (int64_t)threadIdx.x * chunk_words), (unsigned long long)s_hist[threadIdx.x]) : 0ull;
            __syncthreads();
        }
    }
}

// ---- the same scatter, DETERMINISTIC (round 6; round-5 advice: the kernel above places rows by racing atomics — across the four waves of a
// workgroup and, for the chunk cursors, across workgroups — so the order of a chunk's rows changed from run to run, and with it the order in
// which the received probe rows reach the f64 atomics of the join's sink).  Three launches: every WAVE counts its rows per part
// (k_stage_part_count: the rows it will also place — the 64-row batches w, w + 4, ... of its workgroup's segments), one workgroup per part
// scans the waves' counts (k_stage_part_scan: a wave's first place in the chunk, and the chunk's header = every row meant for it), every
// wave places its rows from its own cursor (k_stage_part_place: no atomics; the lanes of a batch that go to one part take consecutive places
// in lane order).  A chunk's rows are then in (workgroup, wave, batch, lane) order: the same bytes run after run.
template <bool PLACE>
__global__ __launch_bounds__(TPB) void k_stage_part_walk(DevStage st, DevPartition pt, int ncols, int64_t chunk_rows, int64_t chunk_words,
                                                         uint32_t* __restrict__ counts, const uint32_t* __restrict__ base, int nwaves, int64_t* __restrict__ packed) {
    __shared__ unsigned int s_cur[TPB / WAVE][SDQH_MAX_PARTS];
    const int wave = (int)(threadIdx.x / WAVE), lane = (int)(threadIdx.x & (WAVE - 1));
    const int gw = (int)blockIdx.x * (TPB / WAVE) + wave;                  // this wave's number among all the launch's waves
    if (lane < pt.nparts) s_cur[wave][lane] = PLACE ? base[(size_t)lane * nwaves + gw] : 0u;
    __builtin_amdgcn_wave_barrier();
    const int seg0 = (int)blockIdx.x * PP_SEGS;
    uint32_t cnt[PP_SEGS];
#pragma unroll
    for (int j = 0; j < PP_SEGS; ++j) cnt[j] = seg0 + j < st.nseg ? st.seg_count[seg0 + j] : 0u;
    uint32_t most = 0;
#pragma unroll
    for (int j = 0; j < PP_SEGS; ++j) most = cnt[j] > most ? cnt[j] : most;
    for (uint32_t i0 = (uint32_t)wave * WAVE; i0 < most; i0 += TPB) {
        int64_t keys[PP_SEGS];
        bool lives[PP_SEGS];
#pragma unroll
        for (int j = 0; j < PP_SEGS; ++j) {
            lives[j] = i0 + lane < cnt[j];
            keys[j] = lives[j] ? st.key[(int64_t)(seg0 + j) * st.seg_rows + i0 + lane] : 0;
        }
#pragma unroll
        for (int j = 0; j < PP_SEGS; ++j) {
            if (i0 >= cnt[j]) continue;                                    // (wave-uniform)
            const bool live = lives[j];
            const int64_t row = (int64_t)(seg0 + j) * st.seg_rows + i0 + lane;
            const int64_t key = keys[j];
            const int part = live ? part_of(pt, key) : -1;
            unsigned long long todo = __ballot(live);
            while (todo) {                                                 // one round per distinct part among the wave's rows
                const int leader = __builtin_ctzll(todo);
                const int p = __shfl(part, leader, WAVE);
                const unsigned long long mine = __ballot(live && part == p);
                const unsigned int first = s_cur[wave][p];                 // (this wave's own cell: plain LDS reads and writes, in program order)
                if (PLACE && live && part == p) {
                    const uint64_t at = (uint64_t)first + (unsigned int)__builtin_popcountll(mine & ((1ull << lane) - 1ull));
                    if (at < (uint64_t)chunk_rows) {
                        int64_t* chunk = packed + (int64_t)part * chunk_words + 2;
                        chunk[at] = key;
#pragma unroll
                        for (int c = 1; c < SDQH_MAX_COMPACT_COLS; ++c) if (c < ncols) chunk[(int64_t)c * chunk_rows + at] = st.pay[c - 1][row];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (lane == leader) s_cur[wave][p] = first + (unsigned int)__builtin_popcountll(mine);
                __builtin_amdgcn_wave_barrier();
                todo &= ~mine;
            }
        }
    }
    if (!PLACE && lane < pt.nparts) counts[(size_t)lane * nwaves + gw] = s_cur[wave][lane];
}
// one workgroup per part: the exclusive scan of the waves' counts for that part (in place: counts -> bases), the part's total into its chunk header
__global__ __launch_bounds__(TPB) void k_stage_part_scan(uint32_t* __restrict__ counts, int nwaves, int64_t chunk_words, int64_t* __restrict__ packed) {
    constexpr int PER = 16;
    __shared__ unsigned long long s_scan[TPB];
    __shared__ unsigned long long s_base;
    uint32_t* row = counts + (size_t)blockIdx.x * nwaves;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nwaves; i0 += TPB * PER) {
        const int first = i0 + (int)threadIdx.x * PER;
        uint32_t c[PER];
        unsigned long long mine = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { c[j] = first + j < nwaves ? row[first + j] : 0u; mine += c[j]; }
        s_scan[threadIdx.x] = mine;
        __syncthreads();
        for (int off = 1; off < TPB; off <<= 1) {
            const unsigned long long v = (int)threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0ull;
            __syncthreads();
            s_scan[threadIdx.x] += v;
            __syncthreads();
        }
        unsigned long long at = s_base + s_scan[threadIdx.x] - mine;
#pragma unroll
        for (int j = 0; j < PER; ++j) { if (first + j < nwaves) row[first + j] = (uint32_t)(at > 0xFFFFFFFFull ? 0xFFFFFFFFull : at); at += c[j]; }
        __syncthreads();
        if (threadIdx.x == 0) s_base += s_scan[TPB - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) { packed[(int64_t)blockIdx.x * chunk_words] = (int64_t)s_base; packed[(int64_t)blockIdx.x * chunk_words + 1] = 0; }
}

// ONE part (an all-gather's send buffer: this rank's entries, whole): the stage's segments back to back IN STAGE ORDER — a scan of the
// segments' live counts, then every wave copies its segment to its place.  Deterministic, and what was in row order stays in row
// order: the replica a receiver rebuilds from the ranks' chunks (rank after rank) sees the table's rows in the table's order.
constexpr int PS_PER = 16;                           // segments per thread of the scan: a workgroup takes TPB * PS_PER of them
__global__ __launch_bounds__(TPB) void k_stage_pack_scan(DevStage st, unsigned long long* __restrict__ offs, int64_t* __restrict__ packed) {
    // Workgroup b places the segments [b * 4096, (b + 1) * 4096).  What lies in front of them it sums for itself — a stage has tens of
    // thousands of segments for a 15 M row build: a few hundred KB at most, read coalesced — so no workgroup waits for another (one
    // workgroup walking all of them, a turn per 4096 with its barriers, was 26 - 31 us of an otherwise idle device in front of every
    // exchange: profiles/r06_distributed_host_profile_world1.txt).
    __shared__ unsigned long long s_scan[TPB];
    const int i0 = (int)blockIdx.x * TPB * PS_PER;
    unsigned long long before = 0;
    for (int i = (int)threadIdx.x; i < i0; i += TPB) before += st.seg_count[i];
    s_scan[threadIdx.x] = before;
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_scan[threadIdx.x] += s_scan[threadIdx.x + off];
        __syncthreads();
    }
    const unsigned long long base = s_scan[0];
    __syncthreads();
    const int first = i0 + (int)threadIdx.x * PS_PER;
    uint32_t c[PS_PER];
    unsigned long long mine = 0;
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) { c[j] = first + j < st.nseg ? st.seg_count[first + j] : 0u; mine += c[j]; }
    s_scan[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < TPB; off <<= 1) {
        const unsigned long long v = (int)threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0ull;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long at = base + s_scan[threadIdx.x] - mine;
#pragma unroll
    for (int j = 0; j < PS_PER; ++j) { if (first + j < st.nseg) offs[first + j] = at; at += c[j]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == TPB - 1) { packed[0] = (int64_t)(base + s_scan[TPB - 1]); packed[1] = 0; }
}
__global__ __launch_bounds__(TPB) void k_stage_pack_copy(DevStage st, const unsigned long long* __restrict__ offs, int ncols, int64_t chunk_rows, int64_t* __restrict__ packed) {
    const int seg = (int)blockIdx.x * (TPB / WAVE) + (int)(threadIdx.x / WAVE), lane = (int)(threadIdx.x & (WAVE - 1));
    if (seg >= st.nseg) return;
    const uint32_t n = st.seg_count[seg];
    const unsigned long long base = offs[seg];
    int64_t* chunk = packed + 2;
    for (uint32_t i = (uint32_t)lane; i < n; i += WAVE) {
        const unsigned long long at = base + i;
        if (at >= (unsigned long long)chunk_rows) break;
        const int64_t row = (int64_t)seg * st.seg_rows + i;
        chunk[at] = st.key[row];
#pragma unroll
        for (int c = 1; c < SDQH_MAX_COMPACT_COLS; ++c) if (c < ncols) chunk[(int64_t)c * chunk_rows + at] = st.pay[c - 1][row];
    }
}

struct DevChunkUnpack {
    const int64_t* packed; const int64_t* sent;
    int64_t* out[SDQH_MAX_COMPACT_COLS];
    int64_t* stat;
    int64_t chunk_rows, chunk_words, pad_key;
    int32_t nparts, ncols, self_part, slot;
};
// blockIdx.y = source * ncols + column: that source's rows of that column to their place behind the earlier sources' rows (every block
// sums the <= 64 headers before its source itself); blockIdx.y = nparts * ncols + column: the padding rows of that column.  The first
// block also records the step's figures in `stat`.
__global__ __launch_bounds__(TPB) void k_unpack_chunks(DevChunkUnpack u) {
    const int y = (int)blockIdx.y;
    const bool padding = y >= u.nparts * u.ncols;
    const int s = padding ? u.nparts : y / u.ncols, c = padding ? y - u.nparts * u.ncols : y % u.ncols;
    int64_t before = 0;
    for (int t = 0; t < s; ++t) { const int64_t n = u.packed[(int64_t)t * u.chunk_words]; before += n < u.chunk_rows ? n : u.chunk_rows; }
    int64_t* to = nullptr;
#pragma unroll
    for (int k = 0; k < SDQH_MAX_COMPACT_COLS; ++k) if (k == c) to = u.out[k];
    if (padding) {
        const int64_t cap = (int64_t)u.nparts * u.chunk_rows, v = c == 0 ? u.pad_key : 0;
        for (int64_t i = before + (int64_t)blockIdx.x * TPB + threadIdx.x; i < cap; i += (int64_t)gridDim.x * TPB) to[i] = v;
    } else {
        int64_t n = u.packed[(int64_t)s * u.chunk_words];
        n = n < u.chunk_rows ? n : u.chunk_rows;
        const int64_t* from = u.packed + (int64_t)s * u.chunk_words + 2 + (int64_t)c * u.chunk_rows;
        to += before;
        for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) to[i] = from[i];
    }
    if (y == 0 && blockIdx.x == 0 && threadIdx.x == 0 && u.stat) {
        int64_t most = 0, recv = 0, sent_all = 0, sent_self = 0;
        for (int t = 0; t < u.nparts; ++t) {
            const int64_t n = u.packed[(int64_t)t * u.chunk_words];
            most = n > most ? n : most; recv += n < u.chunk_rows ? n : u.chunk_rows;
            if (u.sent) {
                const int64_t m = u.sent[(int64_t)t * u.chunk_words];
                most = m > most ? m : most;
                const int64_t mc = m < u.chunk_rows ? m : u.chunk_rows;
                sent_all += mc; if (t == u.self_part) sent_self = mc;
            }
        }
        int64_t* d = u.stat + SDQH_STAT_DETAIL + 4 * u.slot;
        u.stat[SDQH_STAT_MAX_COUNT + u.slot] = most;
        d[0] = recv; d[1] = sent_all; d[2] = sent_self; d[3] = u.chunk_rows;
    }
}

// ---- clustered row pack (round 5): a loop's gathered columns laid out in the ORDER OF ITS FIRST LOOKUP'S KEY ------------------------------
// A final loop whose first lookup is keyed by a column that comes in no row order (Q9: l_partkey against the (part, supplier) table of the
// green parts) tests one bitmap word per ROW — its own L2 request each — and its survivors touch the looked-up table's index, runs and
// their own pack rows at random: 64-byte sectors for 8-16 useful bytes (2.07 GB moved for 0.4 GB of use at SF=10).  The row pack is a
// resident, loop-specific copy already; built in the order of that key instead of row order, the streamed key is clustered (a wave's
// 128 keys share a bitmap word or two), survivors are RUNS of neighbouring pack rows, and the first table's index and entries are walked
// front to back.  What stays random is what the other lookups reach (Q9: the orders index).  The order is a stable LSD radix sort of
// (key - lo, row) on 8-bit digits: per-wave digit counts -> exclusive scan (digit-major) -> every wave places its rows in row order, so
// equal keys keep their row order and the pack — with it the order of every floating-point sum over it — is the same in every run.
constexpr int RS_ROWS = 4096;                        // rows a wave counts / places per pass
constexpr int RS_SCAN = TPB * 8;                     // counters a workgroup of the scan folds

__global__ __launch_bounds__(TPB) void k_rs_init(const int32_t* __restrict__ twin, int64_t lo, int64_t n, uint32_t* __restrict__ key, uint32_t* __restrict__ row) {
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < n; r += (int64_t)gridDim.x * TPB) { key[r] = (uint32_t)((int64_t)twin[r] - lo); row[r] = (uint32_t)r; }
}
__global__ __launch_bounds__(TPB) void k_rs_hist(const uint32_t* __restrict__ key, int64_t n, int shift, uint32_t* __restrict__ hist, int64_t nw) {
    __shared__ unsigned int s_cnt[TPB / WAVE][256];
    const int wv = (int)(threadIdx.x / WAVE), lane = (int)(threadIdx.x & (WAVE - 1));
    for (int d = lane; d < 256; d += WAVE) s_cnt[wv][d] = 0;
    const int64_t w = (int64_t)blockIdx.x * (TPB / WAVE) + wv;
    if (w >= nw) return;
    const int64_t r0 = w * RS_ROWS, r1 = min(n, r0 + RS_ROWS);
    for (int64_t r = r0 + lane; r < r1; r += WAVE) atomicAdd(&s_cnt[wv][(key[r] >> shift) & 255u], 1u);
    for (int d = lane; d < 256; d += WAVE) hist[(int64_t)d * nw + w] = s_cnt[wv][d];
}
// exclusive scan of `total` counters: local part (in place) + one sum per workgroup ...
__global__ __launch_bounds__(TPB) void k_rs_scan_local(uint32_t* __restrict__ v, int64_t total, uint32_t* __restrict__ bsum) {
    __shared__ unsigned int s_part[TPB];
    const int64_t base = (int64_t)blockIdx.x * RS_SCAN + (int64_t)threadIdx.x * 8;
    uint32_t x[8]; uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = base + j < total ? v[base + j] : 0u; sum += x[j]; }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < TPB; off <<= 1) {
        const uint32_t a = (int)threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += a;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) { if (base + j < total) v[base + j] = run; run += x[j]; }
    if (threadIdx.x == TPB - 1) bsum[blockIdx.x] = s_part[TPB - 1];
}
// ... and the workgroup sums, by ONE workgroup, in place
__global__ __launch_bounds__(TPB) void k_rs_scan_top(uint32_t* __restrict__ bsum, int64_t nb) {
    __shared__ unsigned int s_part[TPB];
    const int64_t per = (nb + TPB - 1) / TPB, b0 = (int64_t)threadIdx.x * per, b1 = min(nb, b0 + per);
    uint32_t sum = 0;
    for (int64_t b = b0; b < b1; ++b) sum += bsum[b];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < TPB; off <<= 1) {
        const uint32_t a = (int)threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += a;
        __syncthreads();
    }
    uint32_t run = s_part[threadIdx.x] - sum;
    for (int64_t b = b0; b < b1; ++b) { const uint32_t c = bsum[b]; bsum[b] = run; run += c; }
}
// every wave places its rows, 64 at a time in row order: a lane's place = its digit's cursor + the lanes below it with the same digit
__global__ __launch_bounds__(TPB) void k_rs_scatter(const uint32_t* __restrict__ key, const uint32_t* __restrict__ row, int64_t n, int shift,
                                                    const uint32_t* __restrict__ hist, const uint32_t* __restrict__ bsum, int64_t nw,
                                                    uint32_t* __restrict__ key_out, uint32_t* __restrict__ row_out) {
    __shared__ unsigned int s_at[TPB / WAVE][256];
    const int wv = (int)(threadIdx.x / WAVE), lane = (int)(threadIdx.x & (WAVE - 1));
    const int64_t w = (int64_t)blockIdx.x * (TPB / WAVE) + wv;
    if (w >= nw) return;
    for (int d = lane; d < 256; d += WAVE) { const int64_t i = (int64_t)d * nw + w; s_at[wv][d] = hist[i] + bsum[i / RS_SCAN]; }
    const uint64_t lt = lanemask_lt();
    const int64_t r0 = w * RS_ROWS, r1 = min(n, r0 + RS_ROWS);
    for (int64_t b = r0; b < r1; b += WAVE) {
        const int64_t r = b + lane;
        const bool live = r < r1;
        const uint32_t k = live ? key[r] : 0u, id = live ? row[r] : 0u;
        const uint32_t d = (k >> shift) & 255u;
        uint64_t same = __ballot(live);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) { const uint64_t m = __ballot((d >> bit) & 1u); same &= ((d >> bit) & 1u) ? m : ~m; }
        if (live) {
            const uint32_t at = s_at[wv][d] + (uint32_t)__popcll(same & lt);
            key_out[at] = k; row_out[at] = id;
        }
        if (live && !(same & lt)) s_at[wv][d] += (uint32_t)__popcll(same);         // (the lowest lane of each digit; a wave's LDS accesses keep their order)
    }
}
// lb[v] = the first position of the sorted keys that holds a value >= v, for v = 0 .. nvals - 1 (a thread per value: a binary search)
__global__ __launch_bounds__(TPB) void k_lower_bounds(const uint32_t* __restrict__ sorted, int64_t n, int64_t nvals, uint32_t* __restrict__ lb) {
    for (int64_t v = (int64_t)blockIdx.x * TPB + threadIdx.x; v < nvals; v += (int64_t)gridDim.x * TPB) {
        int64_t lo = 0, hi = n;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if ((int64_t)sorted[mid] < v) lo = mid + 1; else hi = mid; }
        lb[v] = (uint32_t)lo;
    }
}
// pack[i * k + j] = col[j][row[i]] (j >= ncols: padding), key32[i] = twin[row[i]]
struct DevPackPerm { const int64_t* col[MAX_PACK]; int32_t ncols, k; };
__global__ __launch_bounds__(TPB) void k_interleave_perm(DevPackPerm c, const uint32_t* __restrict__ row, const int32_t* __restrict__ twin, int64_t n, int64_t* __restrict__ out, int32_t* __restrict__ key32) {
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
        const uint32_t r = row[i];
        int64_t v[MAX_PACK];
#pragma unroll
        for (int j = 0; j < MAX_PACK; ++j) v[j] = j < c.ncols ? c.col[j][r] : 0;
#pragma unroll
        for (int j = 0; j < MAX_PACK; ++j) if (j < c.k) out[i * c.k + j] = v[j];
        key32[i] = twin[r];
    }
}

}  // namespace

namespace sdqh_host {
int cluster_pack_build(sdqh_ctx* ctx, const int32_t* twin, int64_t lo, int64_t hi, int64_t n, const void* const* cols, int ncols, int k, void* pack_out, void* key32_out, void* lb_out) {
    if (n < 1 || n >= ((int64_t)1 << 32) || hi < lo || (uint64_t)(hi - lo) > 0xFFFFFFFFull || ncols > MAX_PACK || ctx->capturing) return SDQH_ERR_UNSUPPORTED;
    const int64_t nw = (n + RS_ROWS - 1) / RS_ROWS, total = nw * 256, nb = (total + RS_SCAN - 1) / RS_SCAN;
    uint32_t* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    for (auto& b : buf) b = static_cast<uint32_t*>(pool_alloc(ctx, (size_t)n * 4 + 64));
    uint32_t* hist = static_cast<uint32_t*>(pool_alloc(ctx, (size_t)total * 4 + 64));
    uint32_t* bsum = static_cast<uint32_t*>(pool_alloc(ctx, (size_t)nb * 4 + 64));
    auto release = [&]() { for (auto b : buf) if (b) pool_free(ctx, b); if (hist) pool_free(ctx, hist); if (bsum) pool_free(ctx, bsum); };
    if (!buf[0] || !buf[1] || !buf[2] || !buf[3] || !hist || !bsum) { release(); return SDQH_ERR_NOMEM; }
    const unsigned wide = (unsigned)std::max<int64_t>(1, std::min<int64_t>((n + TPB - 1) / TPB, (int64_t)ctx->num_cu * 16));
    const unsigned wgrid = (unsigned)((nw + TPB / WAVE - 1) / (TPB / WAVE));
    { KernelScope _ks(ctx, "k_rs_init"); hipLaunchKernelGGL(k_rs_init, dim3(wide), dim3(TPB), 0, ctx->stream, twin, lo, n, buf[0], buf[1]); }
    uint32_t *ka = buf[0], *ra = buf[1], *kb = buf[2], *rb = buf[3];
    const uint64_t range = (uint64_t)(hi - lo);
    for (int shift = 0; shift < 32 && (shift == 0 || (range >> shift) != 0); shift += 8) {
        { KernelScope _ks(ctx, "k_rs_hist"); hipLaunchKernelGGL(k_rs_hist, dim3(wgrid), dim3(TPB), 0, ctx->stream, ka, n, shift, hist, nw); }
        { KernelScope _ks(ctx, "k_rs_scan_local"); hipLaunchKernelGGL(k_rs_scan_local, dim3((unsigned)nb), dim3(TPB), 0, ctx->stream, hist, total, bsum); }
        { KernelScope _ks(ctx, "k_rs_scan_top"); hipLaunchKernelGGL(k_rs_scan_top, dim3(1), dim3(TPB), 0, ctx->stream, bsum, nb); }
        { KernelScope _ks(ctx, "k_rs_scatter"); hipLaunchKernelGGL(k_rs_scatter, dim3(wgrid), dim3(TPB), 0, ctx->stream, ka, ra, n, shift, hist, bsum, nw, kb, rb); }
        std::swap(ka, kb); std::swap(ra, rb);
    }
    if (lb_out) {                                                        // (hi - lo + 2 entries: the last one = n)
        const int64_t nvals = (int64_t)range + 2;
        KernelScope _ks(ctx, "k_lower_bounds");
        hipLaunchKernelGGL(k_lower_bounds, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((nvals + TPB - 1) / TPB, (int64_t)ctx->num_cu * 16))), dim3(TPB), 0, ctx->stream, ka, n, nvals, static_cast<uint32_t*>(lb_out));
    }
    DevPackPerm pc; std::memset(&pc, 0, sizeof(pc));
    for (int j = 0; j < ncols; ++j) pc.col[j] = static_cast<const int64_t*>(cols[j]);
    pc.ncols = ncols; pc.k = k;
    { KernelScope _ks(ctx, "k_interleave_perm"); hipLaunchKernelGGL(k_interleave_perm, dim3(wide), dim3(TPB), 0, ctx->stream, pc, ra, twin, n, static_cast<int64_t*>(pack_out), static_cast<int32_t*>(key32_out)); }
    const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;      // once per pack: its scratch goes back to a pool the side streams draw from too
    release();
    return ok ? SDQH_OK : SDQH_ERR_DEVICE;
}

int stream_store32(sdqh_ctx* ctx, hipStream_t s, uint32_t* word, uint32_t value) {
    if (ctx->capturing) {
        hipLaunchKernelGGL(k_store32, dim3(1), dim3(64), 0, s, word, value);
        return hipGetLastError() == hipSuccess ? SDQH_OK : SDQH_ERR_DEVICE;
    }
    if (hipStreamWriteValue32(s, word, value, 0) != hipSuccess) { (void)hipGetLastError(); return SDQH_ERR_DEVICE; }
    return SDQH_OK;
}
}  // namespace sdqh_host

extern "C" {

// K-F with NOTHING waited for: count -> write into a staging buffer -> the whole capacity-sized arrays copied out behind the kernels,
// the row count landing in *out_n (a cell of the caller's device-visible block, -1 until then).  The caller collects after
// sdqh_synchronize / sdqh_result_wait / the block's DONE word; a count above `capacity` means the rows beyond were dropped (fetch again,
// larger).  While a plan graph is recorded (sdqh_graph_begin) the staging buffer is the graph's own and the copy stream is forked
// from and joined to the ctx stream inside the recording.
int sdqh_table_compact_deferred(sdqh_ctx* ctx, const sdqh_table* ctable, int64_t min_hits, int64_t capacity,
                                int64_t* out_keys, int64_t* out_payload, double* out_values, int64_t* out_hits, int64_t* out_n) {
    sdqh_table* table = const_cast<sdqh_table*>(ctable);
    if (!ctx || !table || !out_n || capacity < 1 || !out_keys) return fail(ctx, SDQH_ERR_INVALID, "table_compact_deferred: bad arguments");
    if (table->bitmap_only || table->stage_only) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_compact_deferred: bitmap-only / stage-only table");
    (void)hipSetDevice(ctx->device);
    const int npay = out_payload ? table->npay : 0, nval = (out_values && table->accumulate) ? table->nv : 0;
    const size_t cb = (size_t)capacity * 8;
    char* base = reinterpret_cast<char*>(out_keys);
    const int narr = 1 + (out_payload ? table->npay : 0) + (out_values ? SDQH_TUPLE_MAX_VALUES : 0) + (out_hits ? 1 : 0);
    bool contiguous = host_block_contains(ctx, base, cb * (size_t)narr) && host_block_contains(ctx, out_n, 16);
    size_t at = cb;
    if (out_payload) { contiguous = contiguous && reinterpret_cast<char*>(out_payload) == base + at; at += cb * (size_t)table->npay; }
    if (out_values) { contiguous = contiguous && reinterpret_cast<char*>(out_values) == base + at; at += cb * SDQH_TUPLE_MAX_VALUES; }
    if (out_hits) { contiguous = contiguous && reinterpret_cast<char*>(out_hits) == base + at; at += cb; }
    if (!ctx->opt_async_result || !contiguous) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_compact_deferred: the result arrays must be one sdqh_host_alloc block laid out keys | payload | values | hits");
    hipStream_t side = copy_stream(ctx);
    if (!ctx->count_host && hipHostMalloc(&ctx->count_host, 256, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->count_host = nullptr; }
    const bool rec = ctx->capturing;
    const int b = ctx->rs_cur;
    if (!rec && !ctx->rs_copied[b] && hipEventCreateWithFlags(&ctx->rs_copied[b], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->rs_copied[b] = nullptr; }
    if (!ctx->rs_ready && hipEventCreateWithFlags(&ctx->rs_ready, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->rs_ready = nullptr; }
    if (!side || !ctx->count_host || (!rec && !ctx->rs_copied[b]) || !ctx->rs_ready) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_compact_deferred: no side stream");
    const size_t need = cb * (size_t)narr;
    char* dev = nullptr;
    if (rec) {
        dev = static_cast<char*>(pool_alloc(ctx, need + 64));          // the graph's own (pool blocks allocated while recording stay with the graph)
        if (!dev) return fail(ctx, SDQH_ERR_NOMEM, "table_compact_deferred: out of device memory");
    } else {
        if (ctx->rs_bytes[b] < need) {
            if (ctx->rs_used[b]) HIP_TRYA(ctx, hipEventSynchronize(ctx->rs_copied[b]));
            if (ctx->rs_dev[b]) (void)hipFree(ctx->rs_dev[b]);
            ctx->rs_dev[b] = nullptr; ctx->rs_bytes[b] = 0;
            const size_t want = std::max<size_t>(need + need / 4, (size_t)4 << 20);
            if (hipMalloc(&ctx->rs_dev[b], want) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, SDQH_ERR_NOMEM, "table_compact_deferred: out of device memory"); }
            ctx->rs_bytes[b] = want;
        }
        dev = static_cast<char*>(ctx->rs_dev[b]);
    }
    call_begin(ctx);
    if (int rc = index_ensure(ctx, table)) return rc;
    if (!rec && ctx->rs_used[b]) HIP_TRYA(ctx, hipStreamWaitEvent(ctx->stream, ctx->rs_copied[b], 0));       // the buffer's last copy has left it
    DevCompactOut o; std::memset(&o, 0, sizeof(o));
    size_t off = 0;
    o.keys = reinterpret_cast<int64_t*>(dev); off += cb;
    for (int p = 0; p < (out_payload ? table->npay : 0); ++p) { o.pay[p] = reinterpret_cast<int64_t*>(dev + off); off += cb; }
    if (out_values) { for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) { if (k < nval) o.val[k] = reinterpret_cast<double*>(dev + off); off += cb; } }
    if (out_hits) { o.hits = reinterpret_cast<int64_t*>(dev + off); off += cb; }
    o.npay = npay; o.nval = nval;
    o.counter = reinterpret_cast<unsigned long long*>(static_cast<char*>(ctx->count_host) + 64);
    host_init(ctx, &out_n[0], (uint64_t)-1ll, 8);
    host_init(ctx, &out_n[1], 0, 8);
    o.h_counter = reinterpret_cast<unsigned long long*>(out_n);                                      // the kernel's own store of the total, into the caller's block
    o.host_rows = (uint64_t)capacity; o.bounded = 1;
    const uint32_t mh = (uint32_t)std::min<int64_t>(std::max<int64_t>(min_hits, 0), 0xFFFFFFFFll);
    if (int rc = launch_compact_pair(ctx, table, o, mh)) return rc;
    call_end(ctx);
    // the copy waits for the kernels by an event, not the host: keys .. the last used value array in one piece, then the hit counts
    HIP_TRYA(ctx, hipEventRecord(ctx->rs_ready, ctx->stream));
    HIP_TRYA(ctx, hipStreamWaitEvent(side, ctx->rs_ready, 0));
    const int lead = 1 + (out_payload ? table->npay : 0) + nval;
    HIP_TRYA(ctx, hipMemcpyAsync(base, dev, cb * (size_t)lead, hipMemcpyDeviceToHost, side));
    if (out_hits) HIP_TRYA(ctx, hipMemcpyAsync(out_hits, o.hits, cb, hipMemcpyDeviceToHost, side));
    // out_n[1]: the DONE word of this result, written by the copy stream itself behind the copies (1; 2 = no marker: wait with sdqh_result_wait)
    if (stream_store32(ctx, side, reinterpret_cast<uint32_t*>(&out_n[1]), 1) != SDQH_OK) out_n[1] = 2;
    if (rec) {
        // join: a recording ends with every forked stream back on the origin (the next replay's kernels then start behind this replay's copy)
        HIP_TRYA(ctx, hipEventRecord(ctx->rs_ready, side));
        HIP_TRYA(ctx, hipStreamWaitEvent(ctx->stream, ctx->rs_ready, 0));
    } else {
        HIP_TRYA(ctx, hipEventRecord(ctx->rs_copied[b], side));
        ctx->rs_used[b] = true; ctx->rs_pending = true; ctx->rs_cur = b ^ 1;
    }
    // (value slots the tuple does not use are not written: the binding zeroes a block once, when it allocates it, and nothing on the
    //  device ever touches those slots — three capacity-sized host memsets per call were 40 us of a 60 us call on the launch path of every join query)
    return SDQH_OK;
}

// ---- device-sized redistribution --------------------------------------------------------------------------------------------------------
int64_t sdqh_chunk_words(int ncols, int64_t chunk_rows) { return (ncols < 1 || chunk_rows < 0) ? -1 : 2 + (int64_t)ncols * chunk_rows; }

static unsigned grid_of_pack(int nseg) { return (unsigned)std::max(1, (nseg + PP_SEGS - 1) / PP_SEGS); }
int sdqh_table_partition_pack(sdqh_ctx* ctx, const sdqh_table* table, int nparts, const int64_t* range_upper, int64_t chunk_rows, void* packed) {
    if (!ctx || !table || nparts < 1 || nparts > SDQH_MAX_PARTS || chunk_rows < 1 || !packed) return fail(ctx, SDQH_ERR_INVALID, "table_partition_pack: bad arguments");
    if (table->bitmap_only || !table->stage.seg_count || !table->stage.key) return fail(ctx, SDQH_ERR_INVALID, "table_partition_pack: not a staged table");
    if (!(table->stage_only || table->keys_unique || table->pack_unique)) return fail(ctx, SDQH_ERR_UNSUPPORTED, "table_partition_pack: the table's staged rows may repeat a key");
    const int ncols = 1 + table->npay;
    if (ncols > SDQH_MAX_COMPACT_COLS) return fail(ctx, SDQH_ERR_INVALID, "table_partition_pack: too many columns");
    (void)hipSetDevice(ctx->device);
    DevPartition pt; std::memset(&pt, 0, sizeof(pt)); pt.nparts = nparts; pt.by_range = range_upper ? 1 : 0;
    if (range_upper) for (int p = 0; p < nparts - 1; ++p) pt.upper[p] = range_upper[p];
    const int64_t cw = sdqh_chunk_words(ncols, chunk_rows);
    call_begin(ctx);
    if (nparts == 1) {
        // the whole stage into one chunk, in stage order (round 6: the send buffer of an all-gather — a replicated build's entries)
        unsigned long long* offs = static_cast<unsigned long long*>(pool_alloc(ctx, (size_t)table->stage.nseg * 8 + 64));
        if (!offs) return fail(ctx, SDQH_ERR_NOMEM, "table_partition_pack: out of device memory");
        { KernelScope ks(ctx, "k_stage_pack_scan");
          hipLaunchKernelGGL(k_stage_pack_scan, dim3((unsigned)std::max(1, (table->stage.nseg + TPB * PS_PER - 1) / (TPB * PS_PER))), dim3(TPB), 0, ctx->stream, table->stage, offs, static_cast<int64_t*>(packed)); }
        { KernelScope ks(ctx, "k_stage_pack_copy");
          hipLaunchKernelGGL(k_stage_pack_copy, dim3((unsigned)((table->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE))), dim3(TPB), 0, ctx->stream, table->stage, offs, ncols, chunk_rows, static_cast<int64_t*>(packed)); }
        pool_free(ctx, offs);                                          // (stream order: whoever gets the block next runs behind the copy)
        call_end(ctx);
        if (hipGetLastError() != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, "table_partition_pack: launch failed");
        return SDQH_OK;
    }
    if (ctx->opt_pack_ordered) {
        // deterministic placement: count per wave, scan per part, place (no atomics; the chunk headers are written by the scan)
        const int nwaves = (int)grid_of_pack(table->stage.nseg) * (TPB / WAVE);
        uint32_t* counts = static_cast<uint32_t*>(pool_alloc(ctx, (size_t)nparts * (size_t)nwaves * 4 + 64));
        if (!counts) return fail(ctx, SDQH_ERR_NOMEM, "table_partition_pack: out of device memory");
        const unsigned grid = grid_of_pack(table->stage.nseg);
        { KernelScope ks(ctx, "k_stage_part_count");
          hipLaunchKernelGGL(k_stage_part_walk<false>, dim3(grid), dim3(TPB), 0, ctx->stream, table->stage, pt, ncols, chunk_rows, cw, counts, static_cast<const uint32_t*>(counts), nwaves, static_cast<int64_t*>(packed)); }
        { KernelScope ks(ctx, "k_stage_part_scan");
          hipLaunchKernelGGL(k_stage_part_scan, dim3((unsigned)nparts), dim3(TPB), 0, ctx->stream, counts, nwaves, cw, static_cast<int64_t*>(packed)); }
        { KernelScope ks(ctx, "k_stage_part_place");
          hipLaunchKernelGGL(k_stage_part_walk<true>, dim3(grid), dim3(TPB), 0, ctx->stream, table->stage, pt, ncols, chunk_rows, cw, counts, static_cast<const uint32_t*>(counts), nwaves, static_cast<int64_t*>(packed)); }
        pool_free(ctx, counts);                                        // (stream order)
        call_end(ctx);
        if (hipGetLastError() != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, "table_partition_pack: launch failed");
        return SDQH_OK;
    }
    // the headers double as the kernel's cursors: 16 bytes cleared at the head of every chunk (one strided memset)
    HIP_TRYA(ctx, hipMemset2DAsync(packed, (size_t)cw * 8, 0, 16, (size_t)nparts, ctx->stream));
    const unsigned grid = (unsigned)std::max(1, (table->stage.nseg + PP_SEGS - 1) / PP_SEGS);
    { KernelScope ks(ctx, "k_stage_part_pack");
      hipLaunchKernelGGL(k_stage_part_pack, dim3(grid), dim3(TPB), 0, ctx->stream, table->stage, pt, ncols, chunk_rows, cw, static_cast<int64_t*>(packed)); }
    call_end(ctx);
    if (hipGetLastError() != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, "table_partition_pack: launch failed");
    return SDQH_OK;
}

int sdqh_unpack_chunks(sdqh_ctx* ctx, const void* packed, int nparts, int ncols, const int* dtypes, int64_t chunk_rows, int64_t pad_key,
                       const void* sent, int self_part, sdqh_column* stat, int slot, sdqh_column** out_cols) {
    if (!ctx || !packed || nparts < 1 || nparts > SDQH_MAX_PARTS || ncols < 1 || ncols > SDQH_MAX_COMPACT_COLS || !dtypes || chunk_rows < 1 || !out_cols || slot < 0 || slot > 3)
        return fail(ctx, SDQH_ERR_INVALID, "unpack_chunks: bad arguments");
    if (stat && (stat->dtype != SDQH_I64 || stat->nrows < SDQH_EXCHANGE_STAT_WORDS)) return fail(ctx, SDQH_ERR_INVALID, "unpack_chunks: stat must be an I64 column of SDQH_EXCHANGE_STAT_WORDS rows");
    (void)hipSetDevice(ctx->device);
    DevChunkUnpack u; std::memset(&u, 0, sizeof(u));
    u.packed = static_cast<const int64_t*>(packed); u.sent = static_cast<const int64_t*>(sent);
    u.stat = stat ? static_cast<int64_t*>(stat->data) : nullptr;
    u.chunk_rows = chunk_rows; u.chunk_words = sdqh_chunk_words(ncols, chunk_rows); u.pad_key = pad_key;
    u.nparts = nparts; u.ncols = ncols; u.self_part = self_part; u.slot = slot;
    const int64_t cap = (int64_t)nparts * chunk_rows;
    sdqh_column* outs[SDQH_MAX_COMPACT_COLS] = {nullptr};
    for (int c = 0; c < ncols; ++c) {
        int rc = (dtypes[c] != SDQH_I64 && dtypes[c] != SDQH_F64) ? fail(ctx, SDQH_ERR_INVALID, "unpack_chunks: columns are I64 / F64") : sdqh_column_alloc(ctx, cap, dtypes[c], 0, &outs[c]);
        if (rc) { for (int j = 0; j < c; ++j) sdqh_column_free(ctx, outs[j]); return rc; }
        u.out[c] = static_cast<int64_t*>(outs[c]->data);
        sdqh_column_mark_transient(ctx, outs[c]);                  // rows that live for one run: no twins, no dictionaries, no order facts
    }
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((chunk_rows + TPB - 1) / TPB, 128));
    { KernelScope ks(ctx, "k_unpack_chunks");
      hipLaunchKernelGGL(k_unpack_chunks, dim3(gx, (unsigned)((nparts + 1) * ncols)), dim3(TPB), 0, ctx->stream, u); }
    if (hipGetLastError() != hipSuccess) { for (int c = 0; c < ncols; ++c) sdqh_column_free(ctx, outs[c]); return fail(ctx, SDQH_ERR_DEVICE, "unpack_chunks: launch failed"); }
    for (int c = 0; c < ncols; ++c) out_cols[c] = outs[c];
    return SDQH_OK;
}

// ---- plan graphs -------------------------------------------------------------------------------------------------------------------------
int sdqh_graph_begin(sdqh_ctx* ctx) {
    if (!ctx || ctx->compile_only) return fail(ctx, SDQH_ERR_INVALID, "graph_begin: bad arguments");
    if (ctx->capturing) return fail(ctx, SDQH_ERR_INVALID, "graph_begin: already recording");
    if (ctx->profiling) return fail(ctx, SDQH_ERR_UNSUPPORTED, "graph_begin: profiling is on (its events are not part of a plan)");
    (void)hipSetDevice(ctx->device);
    (void)copy_stream(ctx);                                            // made outside the recording
    if (!ctx->rs_ready && hipEventCreateWithFlags(&ctx->rs_ready, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); ctx->rs_ready = nullptr; }
    sdqh_graph* g = new sdqh_graph();
    g->ctx = ctx;
    // relaxed: the recorded calls may allocate (hipMalloc for a pool block that is not there yet) — legal beside a capture in this mode
    if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); delete g; return fail(ctx, SDQH_ERR_UNSUPPORTED, "graph_begin: the stream cannot be captured"); }
    ctx->capturing = true; ctx->capture_tag = g; ctx->capture_inits.clear();
    return SDQH_OK;
}

static void release_graph_blocks(sdqh_ctx* ctx, const void* tag) {
    for (auto& b : ctx->pool) if (b.graph_owner == tag) { b.graph_owner = nullptr; b.free = true; b.nhabits = 0; }
}

int sdqh_graph_abort(sdqh_ctx* ctx) {
    if (!ctx) return SDQH_ERR_INVALID;
    if (!ctx->capturing) return SDQH_OK;
    hipGraph_t graph = nullptr;
    // A call that refused half way may have left a side stream forked into the capture.  This runtime does not take such a capture back
    // (hipStreamEndCapture: Unjoined, and every stream of it stays in capture mode for good — tools/exp/capture_abort.hip), so whatever is
    // still capturing beside the origin joins it first.
    for (hipStream_t s : {ctx->side[0], ctx->side[1]}) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (!s || hipStreamIsCapturing(s, &cs) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (cs != hipStreamCaptureStatusActive || !ctx->rs_ready) continue;
        if (hipEventRecord(ctx->rs_ready, s) != hipSuccess || hipStreamWaitEvent(ctx->stream, ctx->rs_ready, 0) != hipSuccess) (void)hipGetLastError();
    }
    const hipError_t ended = hipStreamEndCapture(ctx->stream, &graph);
    (void)hipGetLastError();
    if (graph) (void)hipGraphDestroy(graph);
    sdqh_graph* g = const_cast<sdqh_graph*>(static_cast<const sdqh_graph*>(ctx->capture_tag));
    ctx->capturing = false; ctx->capture_tag = nullptr; ctx->capture_inits.clear();
    release_graph_blocks(ctx, g);                                       // nothing was executed: the blocks are free at once
    rd_dirty(ctx);
    delete g;
    if (ended != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string("graph_abort: the recording could not be taken back (") + hipGetErrorName(ended) + "): the context's stream stays in capture mode and is lost");
    return SDQH_OK;
}

int sdqh_graph_end(sdqh_ctx* ctx, sdqh_graph** out) {
    if (!ctx || !out) return fail(ctx, SDQH_ERR_INVALID, "graph_end: bad arguments");
    if (!ctx->capturing) return fail(ctx, SDQH_ERR_INVALID, "graph_end: not recording");
    sdqh_graph* g = const_cast<sdqh_graph*>(static_cast<const sdqh_graph*>(ctx->capture_tag));
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
    ctx->capturing = false; ctx->capture_tag = nullptr;
    g->inits.swap(ctx->capture_inits);
    ctx->capture_inits.clear();
    rd_dirty(ctx);                                                      // the recorded calls' claims about the result block were about a run that never happened
    if (e != hipSuccess || !graph) {
        (void)hipGetLastError();
        if (graph) (void)hipGraphDestroy(graph);
        release_graph_blocks(ctx, g);
        delete g;
        return fail(ctx, SDQH_ERR_UNSUPPORTED, std::string("graph_end: the recording is not a graph: ") + hipGetErrorString(e));
    }
    g->graph = graph;
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) == hipSuccess) g->nodes = (int)n; else (void)hipGetLastError();
    if (hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipGraphDestroy(graph);
        release_graph_blocks(ctx, g);
        delete g;
        return fail(ctx, SDQH_ERR_UNSUPPORTED, "graph_end: the graph cannot be instantiated");
    }
    *out = g;
    return SDQH_OK;
}

int sdqh_graph_launch(sdqh_ctx* ctx, sdqh_graph* g) {
    if (!ctx || !g || g->ctx != ctx || !g->exec) return fail(ctx, SDQH_ERR_INVALID, "graph_launch: bad arguments");
    if (ctx->capturing) return fail(ctx, SDQH_ERR_INVALID, "graph_launch: recording");
    (void)hipSetDevice(ctx->device);
    for (const auto& h : g->inits) { if (h.bytes == 8) *static_cast<volatile uint64_t*>(h.p) = h.value; else *static_cast<volatile uint32_t*>(h.p) = (uint32_t)h.value; }
    ++ctx->launch_seq;
    rd_dirty(ctx);                                                      // the replayed kernels used the result block: no claim about it survives
    HIP_TRYA(ctx, hipGraphLaunch(g->exec, ctx->stream));
    return SDQH_OK;
}

int sdqh_graph_nodes(const sdqh_graph* g) { return g ? g->nodes : -1; }

void sdqh_graph_free(sdqh_ctx* ctx, sdqh_graph* g) {
    if (!g) return;
    sdqh_ctx* c = ctx ? ctx : g->ctx;
    if (c && !c->compile_only) {
        (void)hipSetDevice(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);           // a replay may still be running on the blocks about to be released
        if (c->side[1]) (void)hipStreamSynchronize(c->side[1]);
    }
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (c) release_graph_blocks(c, g);
    delete g;
}

}  // extern "C"
