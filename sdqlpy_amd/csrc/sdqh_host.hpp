// sdqh_host.hpp — host-side internals shared by the two translation units of libsdqlhip.so:
//   sdqh_hip.hip   the C ABI over the kernels compiled ahead of time (sdqh_kernels.hpp)
//   sdqh_x.hip     row programs (ABI 4): code generation, hiprtc specialisation, sdqh_x* entry points
// Not part of the public boundary (that is include/sdqh.h).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <string>
#include <vector>

#include "sdqh.h"
#include "sdqh_kernels.hpp"

namespace sdqh_host {

// fill-ahead (launch_fill): a `habit` = a fill this block received in its last use (where in the block, how long, which byte:
// what the block is used as run after run); `clean` = the block holds that byte there — set by a fill while the block was
// FREE, void as soon as its owner's fill has consumed it or the block is freed again.
struct FillHabit { size_t off, bytes; int byte; bool clean; };
struct PoolBlock {
    void* ptr; size_t size; bool free;
    FillHabit habits[4] = {};
    int nhabits = 0;
    uint64_t alloc_seq = 0;                        // the ctx's launch counter when the block was handed out
    bool fill_use = false;                         // this use of the block received (or was spared) a fill: its habits are alive
    const void* graph_owner = nullptr;             // allocated while a plan graph was being recorded (sdqh_graph_begin): reserved until sdqh_graph_free
};
struct ProfEntry { const char* name; hipEvent_t e0, e1; double ms; int64_t model_bytes; };      // model_bytes: HBM bytes this launch is MODELLED to move (0: not modelled), see sdqh_profile_entry_bytes

constexpr size_t STAGING_BYTES = 32u << 20;       // pinned H2D staging ring: 2 x 32 MiB
constexpr size_t RESULT_BYTES = 64u << 10;        // pinned buffer for small results

}  // namespace sdqh_host

using namespace sdqh;

struct sdqh_ctx {
    int device = 0;
    // sdqh_fork: contexts of one family share their columns (each has its own stream, pool and result blocks)
    sdqh_ctx* parent = nullptr;
    std::vector<sdqh_ctx*> children;
    const void* capture_tag = nullptr;             // the graph being recorded (tags the pool blocks it allocates)
    struct HostInit { void* p; uint64_t value; int bytes; };
    std::vector<HostInit> capture_inits;           // host words the recorded calls set before their launches (completion words): set again before every replay
    bool capturing = false;                        // the stream is in capture mode (sdqh_aux.hip: a prepared plan recorded into a hipGraph): nothing may wait for the device
    bool dying = false;                            // sdqh_destroy was called on a family's first context while forks were alive: the last fork releases it
    bool compile_only = false;                     // sdqh_create(-1): no GPU behind this ctx; sdqh_x* calls stop after specialising their kernel (build check)
    int num_cu = 256;
    hipStream_t stream = nullptr;
    bool stream_exported = false;                  // sdqh_stream handed it out: sdqh_destroy leaves it alone
    std::string err;
    std::vector<sdqh_host::PoolBlock> pool;
    void* staging[2] = {nullptr, nullptr};
    hipEvent_t staging_done[2] = {nullptr, nullptr};
    bool staging_busy[2] = {false, false};
    void* result_host = nullptr;                   // pinned
    void* result_dev = nullptr;
    void* bulk_host = nullptr;                     // pinned landing zone for result rows (grown on demand)
    size_t bulk_bytes = 0;
    std::map<const char*, size_t> host_blocks;     // sdqh_host_alloc blocks (base -> bytes): result arrays the device may write
    void* count_host = nullptr;                    // pinned line the compaction kernel writes its row count to
    hipEvent_t call_begin = nullptr, call_end = nullptr;
    bool call_timed = false;
    int profiling = 0;                             // 0 off, 1 per call, 2 accumulate across calls (read at the end)
    std::string prof_filter;                       // record only launches of this kernel (empty = all)
    std::vector<sdqh_host::ProfEntry> prof;
    int64_t next_model_bytes = 0;                  // set by a launcher right before its launch: the bytes that launch streams by construction (encodings x rows); consumed by KernelScope
    std::vector<hipEvent_t> event_pool;
    size_t event_next = 0;
    // hint: key columns whose group count overflowed the register kernel last time
    const void* lds_hint[SDQH_MAX_GROUPKEYS] = {nullptr, nullptr};
    int threads = 1;
    std::vector<std::pair<const void*, int>> occupancy;   // kernel -> resident workgroups per CU
    // tuning knobs (sdqh_set_option)
    int opt_resident_cap = 6;                      // probing kernels (latency chains to hide)
    int opt_async_copies = 0;                      // sdqh_column_copy_in/_out do not wait (the caller synchronises once per batch)
    int opt_probe_chunk = 1;
    int opt_resident_stream = 2;                   // pure streaming kernels: fewer concurrent DRAM streams run faster (tools/microbench_q1.hip)
    int opt_probe_unroll = PROBE_UNROLL;
    int opt_stage_batch = STAGE_BATCH;
    int opt_stage_eager = 1;
    int nested = 0;                                // > 0 while an entry point runs other entry points (call_begin / call_end)
    int opt_stage_eager_pay = 0;                   // measured after the queued stage output: gathers for the ~10 % survivors beat streaming every payload row
    int opt_stage_waves_per_cu = 12;               // tuned k_stage family: fewer, longer streams (12 x 256 x 5 columns) keep DRAM pages open; 24 was 15 % slower, 8 latency-bound
    int opt_direct_index = 1;
    int opt_packed_slots = 1;                      // hash-layout tables with payload: 32-byte slots { key, payload 0 / 1, owner row }
    int opt_lookup_pipeline = -1;                  // k_lookup_agg requests the next tile's first-lookup keys a step ahead: -1 = when that key column is clustered, 0 / 1 = never / always
    int opt_lookup_debug = 0;
    // what the result block (result_dev) is known to hold, left so by a self-resetting k_groupby_merge: rd_clean_ff leading bytes
    // of 0xFF and eight zero bytes at rd_clean_zero_off (-1: none).  Every other writer of the block voids it (rd_dirty).
    size_t rd_clean_ff = 0; int64_t rd_clean_zero_off = -1;
    int opt_fill_ahead = 1;                        // a fill launch also clears the free blocks later builds of the plan habitually need cleared: their fills find them clean
    uint64_t launch_seq = 0;                       // kernels launched through KernelScope (fill-ahead: no launch between a block's allocation and its fill)
    int opt_lds_key_set = 1;                       // membership builds on keys in no row order: bitmaps in LDS per workgroup, folded afterwards (no global atomics)
    int opt_fuse_small = 1;                        // tiny tables (one workgroup): fill inside the build kernel, rank + insert in one launch
    int opt_str_rows = 0;                          // rows of text a wave stages in LDS per round in k_key_set: 0 = by field width, 32, 64
    int opt_rank_increasing = 1;                   // whole-table builds on a strictly increasing key over a wide range: bitmap + row per word (rank = row), no dense array
    int64_t opt_feature_min_rows = 1 << 20;        // narrow twins / row pack (and, x 4, the coarse filter) are for scans of at least this many rows; the suites set 0 to
                                                   // run those instances on tiny and ragged inputs too
    bool in_groupby_key = false;                   // sdqh_groupby_key is running its probe-aggregate pass
    int opt_vstage = 1;                            // unique builds that qualify run on the value-queue stage kernel (x_vstage8)
    int opt_tight = 1;                             // register row programs stream their columns at the tightest exact encoding (dictionary codes of 1 / 2 bytes, 4-byte twins), 8 rows per lane (x_tight)
    hipStream_t side[2] = {nullptr, nullptr};      // side streams: independent build chains of a plan run beside the main stream (fork / join by events)
    hipEvent_t side_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int opt_side_streams = 1;
    int opt_side_priority = 1;          // result copies on a stream of the lowest priority
    int opt_copy_kernel = 0;            // > 0: sdqh_table_compact_deferred copies its rows out with that many workgroups of k_copy_out instead of the runtime's
                                        // copy (measured: 64 workgroups of write-through stores 0.85 ms a step against 0.79 — the runtime's blit kernel stays)
    int opt_copy_nt = 1;
    int opt_row_index = 1;              // unique builds keyed by a strictly increasing column: the stage kernel writes the word -> stage row index itself (no rank / insert passes)
    int opt_grouped_index = 1;                     // composite-key builds over a table stored in the order of the key's first part: the GROUPED layout (DevTable) instead of hash slots
    int opt_grouped_pairs = 1;                     // ... with (key, payload 0) pairs beside the stage (DevTable::grp_kstride)
    int opt_index_inline = 1;                      // a one-workgroup table's build kernel makes the direct index too (DevIndexInline) instead of a k_index_small launch at first use
    int opt_lane_int = 1;               // the per-lane group sink sums an integer-valued byte-coded column as an integer beside the row count (XGroupLane)
    int opt_lane_resident = 2;          // workgroups per CU the per-lane group sink's grid is sized for
    int opt_window = 0;                 // x_queue8's 32-bit prefilter tests a lane's 8 rows against ONE 16-byte window of the bitmap when the key column's 8-row spans allow (column_span8): measured
                                        // on par with 8 single-word requests once those are coalesced (Q3's probe 0.083 vs 0.081 ms) and slower where every row is tested (Q5's final loop 0.140 vs 0.130)
    int opt_x_driven = 64;              // the driven walk of x_queue8 is taken when (estimated entries of the first lookup's table) x this <= rows of the loop; 0 = never
    int opt_word_pairs = 0;             // whole-table builds in the rank = row layout also keep { first row, bits } pairs per bitmap word: a lookup's two requests on one line
                                        // (DevTable::wpair).  Measured on Q9 at SF=10: the final loop 0.264 -> 0.244 ms, the orders build 0.048 -> 0.068 (15 MB more to clear and write): off
    int opt_delta8 = 1;                 // queue programs stream a key column whose 8-row groups span at most 255 through its 12-bytes-per-8-rows delta twin
    int opt_x_waves = 0;                // > 0: wave segments per CU for the queue skeletons of row programs (0: each sink's own default)
    // K-F rows delivered behind the call (sdqh_table_compact_async): two device staging buffers in turn, a copy stream (side[1]),
    // the event of each buffer's last copy, and whether a copy may still be in flight
    void* rs_dev[2] = {nullptr, nullptr}; size_t rs_bytes[2] = {0, 0};
    hipEvent_t rs_copied[2] = {nullptr, nullptr}; bool rs_used[2] = {false, false};
    int rs_cur = 0; bool rs_pending = false;
    hipEvent_t rs_ready = nullptr;      // sdqh_table_compact_deferred: the compaction kernels are done (the side stream's copy waits for it, not the host)
    // sync_stream: a 32-bit sequence number written by the stream itself into pinned memory (hipStreamWriteValue32) and polled by
    // the host, instead of the runtime's wait (which sleeps on an interrupt: tens of microseconds per query on a 0.2 - 0.5 ms query)
    volatile uint32_t* sync_flag = nullptr; uint32_t sync_seq = 0; int opt_spin_sync = 1;
    int opt_async_result = 1;                      // K-F's rows reach the host by a copy queued behind the kernels; the caller waits when it reads them
    int opt_narrow = 1;                            // streaming kernels (k_scan_sum, k_groupby_reg) read predicates / operands through exact 4-byte twins when every one of them has one
    int opt_stage_pipeline = 0;                    // k_stage (tuned orders-like family): first-stage loads of the next step requested a step ahead (measured: Q3 orders 0.148 -> 0.151 ms, no gain: off)
    int opt_span_index = 1;                        // small direct tables also get an owner-by-key-offset array (one-load lookups)
    int opt_dense_increasing = 1;                  // dense layout over a strictly increasing key column is filled in one pass (no prefill, no verification)
    int opt_probe_pipeline = 0;                    // the same for k_probe_agg (keys + first predicate column a step ahead): 0 / 1
    // How dense did the coarse filter of a first lookup come out last time (set bits / bits, written by k_coarsen's count into pinned
    // memory behind the launch: read at the NEXT call, so nothing waits for it)?  A filter that passes most rows costs its LDS test and
    // its 1024-thread workgroups for nothing — Q9 at SF=100, 2.5 MB of part bitmap behind 64 KiB: 89 % pass — and is left out then: the
    // exact bitmap (L2-resident up to a few MB per XCD) is tested directly.  Keyed by the streamed key column and the build's row count.
    const void* coarse_stat_col = nullptr; uint64_t* coarse_stat = nullptr;      // pinned: set bits of the last filter built (coarse_stat_bits_pending = its size)
    unsigned long long* coarse_count_dev = nullptr; uint64_t coarse_stat_bits_pending = 0; int coarse_skipped = 0; int64_t coarse_stat_build_rows = -1;
    int opt_coarse_kb = 64;                        // LDS budget (KiB) of the coarse key filter in front of an unclustered first lookup; 0 = off.  One copy per
                                                   // 1024-thread workgroup (a copy per 256 threads cost more occupancy than it saved: 0.70 -> 0.67 ms at 32 KiB, 1.4 ms at 64 KiB)
    int opt_row_pack = 1;                          // final loops with lookups gather their columns from an interleaved row pack (see DevLookups)
    // order_col: null = rows in row order; else the pack is CLUSTERED — its rows stand in the (stable) order of that column's values, and key32
    // holds that column's 4-byte twin in the same order (sdqh_aux.hip: cluster_pack_build)
    struct RowPack { std::vector<const void*> cols; int64_t nrows; int k; void* data; const void* order_col = nullptr; void* key32 = nullptr;
                     void* lb = nullptr; int64_t key_lo = 0, key_hi = -1; };      // lb: per key value of [key_lo, key_hi + 1] the first pack row holding a key >= it
    std::vector<RowPack> packs;                    // resident row packs, by column set (and order)
    int opt_pack_ordered = 1;                      // sdqh_table_partition_pack places a chunk's rows deterministically (count / scan / place: three launches); 0: one launch, racing atomics
    int opt_hash_filter = 1;                       // hash-layout tables carry a hashed filter (DevTable::hf) that loops test on streamed registers in front of the slots
    int opt_cluster_list = 1;                      // ... and walks the first table's key bitmap and the pack's runs (k_lookup_agg: RUN WALK) instead of streaming the ordered keys
    int opt_cluster_pack = 1;                      // a final loop whose first lookup's key column comes in no row order, whose scan has no predicate and whose gathered columns are all
                                                   // in its row pack runs over a pack CLUSTERED by that key (Q9: l_partkey): 0 = never, 2 = whatever the key's order (the tests)
    int opt_groupby_regs = 0;                      // 0 = adaptive (4 when the last run of these key columns had <= 4 groups), 4, 8
    const void* g4_hint[SDQH_MAX_GROUPKEYS] = {nullptr, nullptr};
};

inline uint64_t sdqh_next_uid() { static std::atomic<uint64_t> next{1}; return next.fetch_add(1, std::memory_order_relaxed); }

struct sdqh_column {
    uint64_t uid = sdqh_next_uid();    // never reused (an address is): what a fact about TWO columns names its partner by
    uint64_t pair_uid = 0; int pair_increasing = -1;      // (this column, column `pair_uid`): no pair twice — strictly increasing as pairs row after row, or distinct within every run of this column's equal values? (one partner cached)
    sdqh_ctx* home = nullptr;          // the context that created the column: its attachments (twins, dictionaries, statistics) live in THAT pool, whichever context of the family builds them
    void* data = nullptr;
    int64_t nrows = 0;
    int dtype = SDQH_I64;
    int width = 0;
    bool owned = false;
    long long* d_minmax = nullptr;     // device [2], I64 only
    bool minmax_pending = false, have_minmax = false;
    int clustered = -1;                // -1 unknown; 1: neighbouring rows hold near-by values (sampled), 0: no order
    void* narrow = nullptr;            // 4-byte twin (int32 values / two-decimal doubles x 100), verified exact when built; see ensure_narrow
    int narrow_state = -1;             // -1 not tried, 0 the column does not narrow exactly, 1 twin present
    int increasing = -1;               // -1 unknown; 1: strictly increasing (sorted, no duplicates), 0: not — checked once on the device
    int nondecreasing = -1;            // -1 unknown; 1: never decreasing (the table is stored in this column's order: equal values are neighbours), 0: not
    int span8 = -1;                    // -1 unknown; 1: aligned groups of 8 consecutive rows span at most 96 values (sampled): a lane's 8 rows fit one 128-bit window of a key bitmap (sdqh_x.hip)
    int64_t mn = 0, mx = 0;
    // sorted-dictionary codes (sdqh_codes.hip): a column with at most 65 536 distinct values over a narrow integer / two-decimal
    // range also has a 1- or 2-byte twin holding, per row, the RANK of its value among the column's distinct values, and the
    // dictionary itself (ndict raw 8-byte values, ascending).  Order-preserving: `value < C` is `code < rank(C)`.
    void* code = nullptr;              // device: nrows codes of code_width bytes
    int code_width = 0;                // 1 or 2
    int code_state = -1;               // -1 not tried, 0 no codes (too many values / too wide a range / not exact), 1 present
    void* dict = nullptr;              // device: ndict raw 8-byte values (int64, or the bits of the doubles), ascending
    int ndict = 0;
    std::vector<int64_t> dict_host;    // the same on the host (bounds of comparisons are translated into code space at launch)
    // RUN INDEX of a never-decreasing column (sdqh_x.hip: column_run_index): per value v of [mn, mx] the first row that holds it (0xFFFFFFFF: none) —
    // a final loop whose first lookup is keyed by such a column and hits few of its values walks the TABLE's keys and their row runs
    // instead of streaming the column (x_queue8's driven walk)
    // DELTA twin (sdqh_x.hip: column_delta8): aligned groups of 8 consecutive rows as 12 bytes — the group's smallest value (int32) and eight
    // one-byte offsets from it — for a column whose groups span at most 255 (a key the table is stored in the order of, a foreign key of
    // such a table: o_orderkey, l_orderkey).  1.5 bytes per row where the 4-byte twin has 4; verified row by row when built.
    void* delta8 = nullptr;            // device: ceil(nrows / 8) records of 12 bytes
    int delta8_state = -1;             // -1 not tried, 0 none, 1 present
    void* run_index = nullptr;         // device: (mx - mn + 2) uint32
    int run_index_state = -1;          // -1 not tried, 0 none (not ordered / too wide a range / too many rows), 1 present
    bool transient = false;            // a view of a table's K-F buffers (sdqh_table_columns): lives for one run — no twins, no statistics gathered for it
    size_t row_bytes() const { return dtype == SDQH_STR ? (size_t)width * 4 : 8; }
};

struct sdqh_table {
    sdqh::DevTable dev{};
    DevStage stage{};
    TableHeader* hdr = nullptr;
    uint32_t* bm = nullptr;
    int npay = 0;
    bool accumulate = false;
    bool bitmap_only = false;
    int64_t nrows_build = 0;
    uint64_t capmax = 0;
    bool index_built = false;
    uint64_t nwords = 0;               // bitmap words (direct layout)
    std::vector<void*> owned;          // pool blocks to release
    // cached compaction (device buffers) for the two-step count / fetch protocol
    uint32_t* span = nullptr;                      // owner by key offset (small plain-key direct tables), becomes dev.dense_arr once the index is built
    bool pack_unique = false;                      // no two staged rows share a key although keys_unique does not say so (a composite key of two plain columns in which no pair comes twice: columns_pair_increasing): sdqh_table_partition_pack may take the stage as the entries
    bool keys_unique = false;                      // the build key is a strictly increasing column: no two staged rows share a key (k_fill_refs has nothing to do)
    bool refs_prefilled = false;                   // small direct tables: dense_ref was allocated and NO_ROW-filled with the header
    bool compact_valid = false;
    int64_t compact_min_hits = 0, compact_n = 0;
    DevCompactOut compact{};
    uint32_t* seg_kept = nullptr;
    void* zero_rows = nullptr;                     // nrows_build + 1 zeroed 8-byte rows: the accumulator columns a table does not have (sdqh_table_columns)
    int nv = SDQH_TUPLE_MAX_VALUES;    // value count of the tuple aggregated into the table
    WordExc* wexc = nullptr;                       // row index (DevTable): exception records, one slot per segment
    uint32_t* coarse = nullptr; int coarse_words = 0, coarse_shift = 0;     // coarse key filter (see DevLookups), built on first need
    bool stage_only = false;                       // sdqh_xstage: every passing row staged (equal keys included), never indexed — the source of a redistribution step, nothing else
};


namespace sdqh_host {

// ---- helpers defined in sdqh_hip.hip ----------------------------------------------------------------
int fail(sdqh_ctx* ctx, int code, const std::string& msg);
void* pool_alloc(sdqh_ctx* ctx, size_t bytes);
void pool_free(sdqh_ctx* ctx, void* p);
void call_begin(sdqh_ctx* ctx);
void call_end(sdqh_ctx* ctx);
hipEvent_t next_event(sdqh_ctx* ctx);
int sync_stream(sdqh_ctx* ctx);
void* tb_alloc(sdqh_ctx* ctx, sdqh_table* t, size_t bytes);
// memory attached to a column (see sdqh_column::home); a context other than the home waits for the home's stream first — a recycled block
// of that pool is only safe behind the work queued there
void* attach_alloc(sdqh_ctx* ctx, const sdqh_column* c, size_t bytes);
void attach_free(sdqh_ctx* ctx, const sdqh_column* c, void* p);
void tb_release(sdqh_ctx* ctx, sdqh_table* t);
// stage arrays of a build whose key / payload the kernel computes itself (no source columns)
int stage_setup_computed(sdqh_ctx* ctx, sdqh_table* tb, int64_t nrows, int npay, int batch);
int index_ensure(sdqh_ctx* ctx, sdqh_table* tb);
// every staged row of a freshly built table (no index is made, duplicate keys stay): key + payload as resident I64 columns
int stage_rows_out(sdqh_ctx* ctx, sdqh_table* tb, sdqh_column** out_cols, int64_t* out_rows);
// K-F into the table's own device buffers (tb->compact; rows in build-row order), *n = the number of entries kept; zero_rows made on request
int table_compact_resident(sdqh_ctx* ctx, sdqh_table* tb, int64_t min_hits, bool want_zero_rows, int64_t* n);
int launch_compact_pair(sdqh_ctx* ctx, sdqh_table* table, const sdqh::DevCompactOut& o, uint32_t min_hits);      // k_compact_count + k_compact_write2
hipStream_t copy_stream(sdqh_ctx* ctx);                         // the low-priority stream result copies are queued on (made on first use)
int column_minmax(sdqh_ctx* ctx, sdqh_column* c);
bool column_increasing(sdqh_ctx* ctx, sdqh_column* c);          // strictly increasing I64 column?  (one pass the first time, cached)
bool column_nondecreasing(sdqh_ctx* ctx, sdqh_column* c);       // never decreasing?
const void* column_narrow(sdqh_ctx* ctx, sdqh_column* c);      // the exact 4-byte twin of a streamed column (built on first request), or nullptr
bool column_codes(sdqh_ctx* ctx, sdqh_column* c);              // sdqh_codes.hip: the sorted-dictionary code twin (built on first request); false: none
void column_codes_release(sdqh_ctx* ctx, sdqh_column* c);
// launches of ahead-of-time kernels the run-time specialised path needs
void fill_regions(sdqh_ctx* ctx, void* const* ptr, const size_t* bytes, const unsigned char* byte, int n);
void launch_sum_partials(sdqh_ctx* ctx, const double* partial, int nparts, double* out);
void launch_groupby_merge_lg(sdqh_ctx* ctx, const unsigned long long* gkeys, const double* pacc, const int64_t* pcnt, int nparts, double* out_acc, int64_t* out_cnt);
bool rd_take_clean_lg(sdqh_ctx* ctx);
void launch_groupby_merge_lg_host(sdqh_ctx* ctx, unsigned long long* r_keys, const double* pacc, const int64_t* pcnt, int nparts, int* r_flags, void* host_block = nullptr);
bool host_block_contains(sdqh_ctx* ctx, const void* p, size_t bytes);      // inside a block from sdqh_host_alloc?
// small direct-layout tables: the rank -> row array is allocated up front; *ptr / *bytes = a region to fill with 0xFF (null: none)
int prefill_direct_refs(sdqh_ctx* ctx, sdqh_table* tb, void** ptr, size_t* bytes);   // regions (<= 2) to fill with 0xFF; returns their number

inline void rd_dirty(sdqh_ctx* c) { c->rd_clean_ff = 0; c->rd_clean_zero_off = -1; }

// ---- helpers defined in sdqh_aux.hip ----------------------------------------------------------------
// a row pack in the stable order of a key column's values (twin: the column's exact 4-byte twin, lo / hi: its minimum / maximum):
// pack_out[i * k + j] = cols[j][row_i], key32_out[i] = twin[row_i]; waits for the stream once (scratch returned to the pool)
// lb_out (optional): hi - lo + 2 uint32, lb[v - lo] = the first pack row whose key is >= v (the last entry: n)
int cluster_pack_build(sdqh_ctx* ctx, const int32_t* twin, int64_t lo, int64_t hi, int64_t n, const void* const* cols, int ncols, int k, void* pack_out, void* key32_out, void* lb_out);
// one 32-bit word of device-visible host memory stored by stream `s` itself, behind what is queued there: hipStreamWriteValue32 where
// the stream executes, a one-thread kernel where it is being recorded into a plan graph (the value-write has no graph node)
int stream_store32(sdqh_ctx* ctx, hipStream_t s, uint32_t* word, uint32_t value);
// a host word a call sets BEFORE its launches (a completion word cleared, a count set to -1): while a plan graph is recorded it is also
// noted, and set again before every replay (sdqh_graph_launch)
inline void host_init(sdqh_ctx* ctx, void* p, uint64_t value, int bytes) {
    if (bytes == 8) *static_cast<volatile uint64_t*>(p) = value; else *static_cast<volatile uint32_t*>(p) = (uint32_t)value;
    if (ctx->capturing) ctx->capture_inits.push_back(sdqh_ctx::HostInit{p, value, bytes});
}

// event pair around one launch when profiling is on (same bookkeeping as the LAUNCH macro)
struct KernelScope {
    sdqh_ctx* ctx; size_t idx = (size_t)-1;
    KernelScope(sdqh_ctx* c, const char* name) : ctx(c) {
        ++c->launch_seq;
        const int64_t model = c->next_model_bytes;
        c->next_model_bytes = 0;
        if (!c->profiling) return;
        if (!c->prof_filter.empty() && c->prof_filter != name) return;
        ProfEntry e{name, next_event(c), next_event(c), 0.0, model};
        (void)hipEventRecord(e.e0, c->stream);
        idx = c->prof.size(); c->prof.push_back(e);
    }
    ~KernelScope() { if (idx != (size_t)-1) (void)hipEventRecord(ctx->prof[idx].e1, ctx->stream); }
};

}  // namespace sdqh_host
