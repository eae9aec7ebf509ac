/*
 * sdqh.h — C ABI of the MI355X-native execution backend for sdqlpy's hot path.
 *
 * One header, two implementations with identical exports:
 *   - sdqlpy_amd/csrc/libsdqlhip.so   hand-written HIP kernels for gfx950 (the product)
 *   - oracle/libsdqloracle.so         CPU restatement of the reference plan (test infrastructure only)
 *
 * What each entry point replaces in the reference (edin-dal/sdqlpy, paths relative to its root):
 *
 *   The reference has no FFI for this path: `@sdql_compile` imports a generated CPython
 *   extension and calls `<fn>_compiled(db)` with db = list[table] of list[column] of numpy
 *   arrays (src/sdqlpy/sdql_lib.py:401-424).  The generated C++ binds every column to a raw
 *   typed pointer (src/sdqlpy/lib/sdql_compiler.py:638-672) and then runs one emitted loop per
 *   SumExpr (src/sdqlpy/lib/sdql_ir_cpp_generator_par.py:176-570).  This ABI cuts at exactly
 *   that level: columns in, one call per emitted loop shape, struct-of-arrays results out.
 *
 *   sdqh_column_upload        <- column binding, sdql_compiler.py:638-672 (long* / double* /
 *                                VarChar<n>* over numpy buffers; include/varchar.h:1-3)
 *   sdqh_set_threads          <- tbb::task_scheduler_init scheduler(N), sdql_compiler.py:485
 *   sdqh_scan_filter_sum      <- K-A  tbb::parallel_reduce loop, ...generator_par.py:258-291
 *   sdqh_groupby_small        <- K-C  aggregating-dict loop + AddMap merge for small key domains,
 *                                ...generator_par.py:402-440, include/map_helper.h:1-23,
 *                                include/tuple_helper.h:36-41,79-84
 *   sdqh_hash_build_unique    <- K-B  unique-dict build (joinBuild / joinProbe(...,False)),
 *                                ...generator_par.py:331-369, lib/sdql_ir.py:424-454,
 *                                lookups ...generator_par.py:85-96 (contains / at)
 *   sdqh_hash_probe_aggregate <- K-C  probe + group-by loop, ...generator_par.py:402-440
 *   sdqh_table_compact        <- K-F  finalise loop `out[tuple_cat(k,v)] = true`,
 *                                ...generator_par.py:520-568 and result hand-over 871-877
 *   sdqh_scan_compact / sdqh_partition_by_key / sdqh_table_* (bitmap)
 *                             <- no reference counterpart: the multi-GPU redistribution step
 *                                
 *
 * Conventions: plain C, opaque handles, every call returns an int status (0 = ok) and leaves a
 * message retrievable with sdqh_last_error().  Inputs are never written.  Host pointers are
 * only used for the duration of the call.  The library owns all device memory behind `ctx`.
 * One ctx per host thread; no global mutable state.
 */
#ifndef SDQH_H
#define SDQH_H

#ifndef __HIPCC_RTC__
#include <stddef.h>
#include <stdint.h>
#else   /* hiprtc (run-time kernel specialisation): no libc headers; the fixed-width types live in __hip_internal */
using __hip_internal::int8_t;  using __hip_internal::uint8_t;  using __hip_internal::int16_t; using __hip_internal::uint16_t;
using __hip_internal::int32_t; using __hip_internal::uint32_t; using __hip_internal::int64_t; using __hip_internal::uint64_t;
#ifndef INT64_MIN
#define INT64_MIN (-9223372036854775807ll - 1)
#define INT64_MAX 9223372036854775807ll
#endif
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define SDQH_ABI_VERSION 7   /* 2: sdqh_filter carries column-vs-column predicates; string modes 3 / 4.  3: sdqh_table_share_groups.  4: row programs (sdqh_x*).  5: device-sized redistribution, plan graphs.  6: sdqh_xgroupby_partial / _fold.  7: sdqh_lookup_aggregate_block */

/* ---- status codes ---------------------------------------------------------------------- */
#define SDQH_OK              0
#define SDQH_ERR_INVALID     1   /* bad argument (null, dtype mismatch, row-count mismatch) */
#define SDQH_ERR_UNSUPPORTED 2   /* shape outside this backend's vocabulary                 */
#define SDQH_ERR_DEVICE      3   /* HIP runtime error (message holds hipGetErrorString)     */
#define SDQH_ERR_OVERFLOW    4   /* small-domain group-by saw more keys than max_groups     */
#define SDQH_ERR_NOMEM       5

/* ---- column element types (the reference's three storage classes) ------------------------
 * I64: int / date columns, numpy int64, dates as yyyymmdd (sdql_lib.py:83-84)
 * F64: float columns, numpy float64
 * STR: string(n) columns, numpy '<U n' = n UCS4 code units per row, zero padded
 *      (include/varchar.h:1-3: VarChar<n>{ wchar_t data[n] })
 */
#define SDQH_I64 0
#define SDQH_F64 1
#define SDQH_STR 2

/* ---- value tuples ------------------------------------------------------------------------
 * The right-hand side of `+=` in an emitted loop is a tuple of doubles/longs computed from the
 * row (AdditionCodeGenerator, ...generator_par.py:712-795).  The kernels are specialised per
 * tuple shape; operands a..d are F64 columns bound at call time.  Every product is evaluated in
 * the reference's association order with FMA contraction off (test/test_all.py:52,171,293,480).
 */
#define SDQH_TUPLE_A            1  /* { a }                                      1 double          */
#define SDQH_TUPLE_AB           2  /* { a*b }                         (Q6)       1 double          */
#define SDQH_TUPLE_A_1MB        3  /* { a*(1.0-b) }                   (Q3,Q5)    1 double          */
#define SDQH_TUPLE_PRICING      4  /* { a, b, b*(1.0-c), (b*(1.0-c))*(1.0+d), 1 }   (Q1)
                                      4 doubles; the trailing COUNT is the row count output */
#define SDQH_TUPLE_A_1MB_M_CD   5  /* { a*(1.0-b) - c*d }             (Q9)       1 double          */
#define SDQH_TUPLE_COUNT        6  /* { 1 }                                      0 doubles         */
#define SDQH_TUPLE_MAX_VALUES   4

typedef struct sdqh_ctx    sdqh_ctx;
typedef struct sdqh_column sdqh_column;
typedef struct sdqh_table  sdqh_table;

/* ---- row filter: a conjunction, every term a closed range or a string equality ------------
 * `x < c` on ints is passed as hi = c-1; on doubles as hi = nextafter(c,-inf); NaN fails every
 * active range, as in C.  An empty filter passes every row.
 */
#define SDQH_MAX_IPRED 4
#define SDQH_MAX_FPRED 4
#define SDQH_MAX_SPRED 1
#define SDQH_MAX_STR_CONST 64

typedef struct sdqh_ipred { const sdqh_column* col; int64_t lo, hi; } sdqh_ipred;
typedef struct sdqh_fpred { const sdqh_column* col; double  lo, hi; } sdqh_fpred;
typedef struct sdqh_spred {                     /* col == value  (VarChar::operator==, include/varchar.h:61-77) */
    const sdqh_column* col;
    int32_t  len;                               /* code units in value, <= column width */
    int32_t  negate;                            /* 0: col == value, 1: col != value, 2: value is a substring of col
                                                   (`"x" in col` -> VarChar::contains / wcsstr, include/varchar.h:84-89),
                                                   3: col starts with value (startsWith, include/varchar.h:99-110),
                                                   4: col ends with value (endsWith, include/varchar.h:112-124; the text
                                                      is the field up to its first NUL; Python str.endswith semantics) */
    uint32_t value[SDQH_MAX_STR_CONST];
} sdqh_spred;

/* column-vs-column comparison `a[row] op b[row]` (e.g. Q4 `l_commitdate < l_receiptdate`): both I64 or both F64 */
#define SDQH_MAX_CPRED 2
#define SDQH_CMP_LT 0
#define SDQH_CMP_LE 1
#define SDQH_CMP_EQ 2
#define SDQH_CMP_NE 3
typedef struct sdqh_cpred { const sdqh_column* a; const sdqh_column* b; int32_t op; int32_t _pad; } sdqh_cpred;

typedef struct sdqh_filter {
    int32_t n_ipred, n_fpred, n_spred, n_cpred;
    sdqh_ipred ipred[SDQH_MAX_IPRED];
    sdqh_fpred fpred[SDQH_MAX_FPRED];
    sdqh_spred spred[SDQH_MAX_SPRED];
    sdqh_cpred cpred[SDQH_MAX_CPRED];
} sdqh_filter;

typedef struct sdqh_tuple {
    int32_t shape;                              /* SDQH_TUPLE_* */
    int32_t _pad;
    const sdqh_column* a;
    const sdqh_column* b;
    const sdqh_column* c;
    const sdqh_column* d;
} sdqh_tuple;

/* A semi-join step evaluated after the filter: keep the row iff table contains key[row]
 * (`(tbl).contains(k)`, ...generator_par.py:86-96). */
#define SDQH_MAX_PROBE 2
typedef struct sdqh_probe { const sdqh_table* table; const sdqh_column* key; } sdqh_probe;

/* ---- context ----------------------------------------------------------------------------- */
int         sdqh_abi_version(void);
const char* sdqh_backend_name(void);                    /* "hip-gfx950" or "cpu-oracle" */
int         sdqh_create(int device, sdqh_ctx** out);       /* HIP build, device = -1: a compile-only context for build checks on a host without a GPU —
                                                              columns can be wrapped (never dereferenced) and every sdqh_x* call stops after specialising its
                                                              kernel, returning SDQH_ERR_DEVICE with "kernel specialised" (sdqh_xbuild / sdqh_xkey_set: SDQH_OK and a
                                                              placeholder table later programs can name); every other call is invalid on it */
void        sdqh_destroy(sdqh_ctx* ctx);
/* A second context of the same FAMILY: its own stream, memory pool and result blocks on the parent's device, and the right to name
 * the columns of the family's other contexts in its calls (tables stay with the context that built them).  Calls on different
 * contexts of a family run concurrently on the device — independent queries share the chip instead of queueing behind one
 * another on one stream (the reference runs one query at a time; its TBB loops have the machine to themselves).  What the library
 * attaches to a column on first use (narrow twins, dictionaries, statistics) lives with the column and is complete when the call
 * that made it returns.  The caller keeps a column alive, and unchanged, while any context of the family has work in flight that
 * reads it (sdqh_synchronize each before sdqh_column_free / sdqh_column_copy_in); options are per context.  Fork the family's
 * first context only; destroy forks before it.  The concurrency is the DEVICE's: the host issues a family's calls from one thread
 * (or serialises them) — a column's attachments and its home pool are shared host-side state without locks. */
int         sdqh_fork(sdqh_ctx* parent, sdqh_ctx** out);
const char* sdqh_last_error(const sdqh_ctx* ctx);
int         sdqh_set_threads(sdqh_ctx* ctx, int threads);   /* CPU build: worker count; HIP build: accepted, ignored */
int         sdqh_synchronize(sdqh_ctx* ctx);
/* Milliseconds spent on the device by the most recent pattern call (HIP events on the ctx
 * stream, recorded while sdqh_set_profiling is on — 0.0 otherwise: the two events cost barrier
 * packets on the stream; CPU build: wall clock of the call). */
int         sdqh_last_device_ms(const sdqh_ctx* ctx, double* ms);
/* Name and duration (ms) of each kernel launch, measured with HIP events recorded on the ctx
 * stream around the launch.  mode 1: entries of the most recent pattern call; mode 2: entries
 * accumulate over all calls (events are only recorded, nothing synchronises) until the mode is set
 * again — read them with sdqh_profile_count / sdqh_profile_entry after the timed region. */
int         sdqh_set_profiling(sdqh_ctx* ctx, int mode);
/* Restrict the recording to launches of one kernel (exact name; NULL or "" = every kernel), so a
 * timed region can carry events around its dominant kernel only. */
int         sdqh_set_profile_filter(sdqh_ctx* ctx, const char* kernel_name);
int         sdqh_profile_count(const sdqh_ctx* ctx);
int         sdqh_profile_entry(const sdqh_ctx* ctx, int i, const char** name, double* ms);
/* HBM bytes launch i is MODELLED to move, from what the library itself decided for that launch: bytes per row of every streamed
 * column at the encoding chosen (8-byte column / 4-byte twin / 1- or 2-byte dictionary code) x rows, plus the key bitmap a
 * streamed prefilter reads and the stores the launch makes by construction.  Gathers by row and data-dependent stores (survivor
 * counts are on the device) are NOT in it, so for probing kernels it is a lower bound; for pure streaming kernels (K-A, small K-C)
 * it is what the PMC counters should show (bench.py prints it beside them: SURVEY.md 8(d) "physical bytes moved").  0: no model
 * for that kernel.  CPU build: always invalid (no launches). */
int         sdqh_profile_entry_bytes(const sdqh_ctx* ctx, int i, int64_t* model_bytes);
/* Raw hipStream_t the ctx launches on (NULL in the CPU build). */
void*       sdqh_stream(const sdqh_ctx* ctx);
/* Tuning knobs of the HIP build; results never depend on them (the parity suite runs with the defaults, A/B runs in
 * profiles/ flip them).  "resident_cap", "resident_stream", "probe_unroll", "probe_chunk", "stage_batch", "stage_eager",
 * "stage_eager_pay", "stage_waves_per_cu", "direct_index", "async_copies", "groupby_regs"; round 2: "narrow" (1: streamed
 * columns through exact 4-byte twins), "row_pack" (1), "packed_slots" (1), "coarse_kb" (64: LDS budget of the coarse key
 * filter, 0 = off), "lookup_pipeline" (-1 auto / 0 / 1), "probe_pipeline" (0), "stage_pipeline" (0), "span_index" (1),
 * "dense_increasing" (1), "rank_increasing" (1), "fuse_small" (1), "fill_ahead" (1: a fill launch also clears the free blocks later builds will want cleared), "lds_key_set" (1: membership builds on keys in no row
 * order through per-workgroup bitmaps in LDS), "feature_min_rows" (2^20: the row count from which twins, packs and the wide
 * instances are used; the tests set 0), "str_rows", "lookup_debug" (cut points of k_lookup_agg for measurements: results ARE
 * wrong with it); round 5: "cluster_pack" (1: the row pack of a final loop whose first lookup's key column comes in no row order is
 * built in the stable order of that key and the loop runs over pack rows; 0 = never, 2 = whatever the key's order), "cluster_list" (1: such
 * a loop walks the first table's key bitmap and the pack's runs instead of streaming the ordered keys), "x_driven" (64: a row program whose
 * first lookup is keyed by the column its table is stored in the order of walks the looked-up table's key bitmap and the column's run
 * index when (estimated keys of the table) x this <= rows of the loop; 0 = never, 1 = whenever the table holds fewer keys than rows),
 * "delta8" (1: queue programs stream a key column whose aligned 8-row groups span at most 255 through its delta twin, 12 bytes per 8 rows),
 * "word_pairs" (0: whole-table builds keyed by a strictly increasing column also keep { first row, bits } pairs per bitmap word);
 * round 6: "pack_ordered" (1: sdqh_table_partition_pack places rows deterministically, see there), "hash_filter" (1: a hash-layout table carries a hashed filter — 8 to 16 bits per key, two bits of one word per key, L2-sized — that
 * row programs test on streamed registers in front of the slots, as they test an exact key bitmap), "pool_trim" (an action, value 1: waits for the context's stream and returns the cached free blocks of its device-memory
 * pool to the runtime — a pool never shrinks by itself).
 * The CPU build accepts and ignores any name. */
int         sdqh_set_option(sdqh_ctx* ctx, const char* name, int64_t value);
/* What this context's device-memory pool holds (round 6: bench.py's `hbm` accounting — the reference keeps nothing resident, it reads
 * the caller's numpy buffers in place, sdql_compiler.py:638-672): out[0] bytes handed out (columns and what is attached to them —
 * twins, codes, dictionaries, delta twins, run indexes, row packs —, live tables, staging), out[1] bytes cached free (reused by the
 * next runs, returned by "pool_trim"), out[2] of out[0] + out[1]: bytes reserved by recorded plan graphs, out[3] blocks.  n >= 4.
 * A family's contexts have a pool each.  The CPU build reports zeros. */
int         sdqh_memory_stats(sdqh_ctx* ctx, int64_t* out, int n);

/* ---- columns ----------------------------------------------------------------------------- */
/* Copy a host column into device memory through a pinned staging ring (chunked, async H2D on
 * the ctx stream) and record min/max for I64 columns.  width = code units per row for STR, else 0. */
int     sdqh_column_upload(sdqh_ctx* ctx, const void* host, int64_t nrows, int dtype, int width, sdqh_column** out);
/* Wrap device memory owned by the caller (e.g. a torch tensor); 16-byte aligned. */
int     sdqh_column_wrap(sdqh_ctx* ctx, void* device_ptr, int64_t nrows, int dtype, int width, sdqh_column** out);
/* Uninitialised device column owned by the library. */
int     sdqh_column_alloc(sdqh_ctx* ctx, int64_t nrows, int dtype, int width, sdqh_column** out);
int     sdqh_column_download(sdqh_ctx* ctx, const sdqh_column* col, int64_t row0, int64_t nrows, void* host);
void*   sdqh_column_data(const sdqh_column* col);
int64_t sdqh_column_rows(const sdqh_column* col);
int     sdqh_column_dtype(const sdqh_column* col);
int     sdqh_column_width(const sdqh_column* col);
int     sdqh_column_minmax(sdqh_ctx* ctx, const sdqh_column* col, int64_t* min, int64_t* max);
void    sdqh_column_free(sdqh_ctx* ctx, sdqh_column* col);

/* ---- K-A: scan -> filter -> scalar / record reduce ------------------------------------------
 * out_values[SDQH_TUPLE_MAX_VALUES] receives the tuple's doubles, *out_count the rows that passed. */
int sdqh_scan_filter_sum(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
                         const sdqh_tuple* tuple, double* out_values, int64_t* out_count);

/* K-A with semi-join steps: as sdqh_scan_filter_sum, over the rows that also pass every probe
 * (`expr if tbl[key] != None else 0.0` inside a scalar sum — Q14's promo revenue,
 * test/test_all.py:703-711; lookups ...generator_par.py:85-96). */
int sdqh_scan_probe_sum(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
                        int nprobes, const sdqh_probe* probes, const sdqh_tuple* tuple,
                        double* out_values, int64_t* out_count);

/* ---- K-C small: scan -> filter -> group-by over a small key domain -------------------------
 * Keys: 1..2 columns, each STR of width 1 or I64 with values in [0, 2^32-2].  Results, one row
 * per group in unspecified order: out_keys[g*nkeys + k] (code point or integer),
 * out_values[g*SDQH_TUPLE_MAX_VALUES + v], out_counts[g].  SDQH_ERR_OVERFLOW if the data holds
 * more than max_groups (<= 64) distinct keys. */
#define SDQH_MAX_GROUPKEYS 2
#define SDQH_MAX_SMALL_GROUPS 64
int sdqh_groupby_small(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
                       int nkeys, const sdqh_column* const* keys, const sdqh_tuple* tuple,
                       int max_groups, int64_t* out_keys, double* out_values, int64_t* out_counts,
                       int32_t* out_ngroups);

/* ---- K-B: scan -> filter -> semi-join probes -> unique hash build ----------------------------
 * Builds key -> (payload columns of the same row); the first row (lowest index) wins on a
 * duplicate key, as `emplace`/`insert(range)` do in row order (...generator_par.py:366-367,766-773).
 * Payload columns are I64 or F64 (carried as 8 raw bytes).  With accumulate != 0 the table also
 * carries SDQH_TUPLE_MAX_VALUES double accumulators and a hit counter per entry for
 * sdqh_hash_probe_aggregate. */
#define SDQH_MAX_PAYLOAD 4
int sdqh_hash_build_unique(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
                           int nprobes, const sdqh_probe* probes,
                           const sdqh_column* key, int npayload, const sdqh_column* const* payload,
                           int accumulate, sdqh_table** out);
/* Membership-only build: the set of key[row] over the rows that pass filter + probes, kept as an
 * exact bitmap over the key range — what a `{unique(key): True}` build is for when it only ever
 * answers `tbl[k] != None` (EXISTS sub-queries: Q4's late-lineitem orders, test/test_all.py:183-193).
 * Nothing is staged: one streaming pass that ORs bits.  The table supports semi-join probes,
 * payload-less lookups and sdqh_table_size only (as tables from sdqh_table_from_bitmap).
 * SDQH_ERR_UNSUPPORTED when the key range is too wide or sparse for a bitmap (range > 2^31 or
 * > 64 x nrows): use sdqh_hash_build_unique then. */
int  sdqh_build_key_set(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
                        int nprobes, const sdqh_probe* probes, const sdqh_column* key, sdqh_table** out);
int  sdqh_table_size(sdqh_ctx* ctx, const sdqh_table* table, int64_t* entries);
void sdqh_table_free(sdqh_ctx* ctx, sdqh_table* table);

/* ---- K-C large: scan -> filter -> probe -> aggregate into the matched entry -------------------
 * For each passing row whose key is in `table`: entry.acc += tuple(row), entry.hits += 1.  This is
 * the reference's group-by on (probe key, fields of the matched entry), where the group is
 * functionally determined by the probe key (test/test_all.py:164-172). */
int sdqh_hash_probe_aggregate(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
                              sdqh_table* table, const sdqh_column* key, const sdqh_tuple* tuple);

/* ---- K-C with a large key domain, keyed by a column of the scanned row ---------------------------
 * `{row.key: tuple}` summed per distinct key of an int column (Q18's sum of l_quantity per
 * l_orderkey, test/test_all.py:877; emitted as the aggregating-dict loop ...generator_par.py:402-440).
 * The result is a table whose entries are the distinct keys of the passing rows, each with the
 * accumulated tuple and its row count: what sdqh_hash_build_unique(accumulate) on the rows followed
 * by sdqh_hash_probe_aggregate of the same rows would leave, in one call. */
int sdqh_groupby_key(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter, const sdqh_column* key,
                     const sdqh_tuple* tuple, sdqh_table** out);

/* HAVING: the keys of the entries with at least min_hits rows whose accumulator `value_index` lies
 * in [lo, hi], as a membership-only table (`{unique(k): True} if v > 300` over an aggregated
 * dictionary, test/test_all.py:879-885).  The source must be a table with a dense key range (what
 * sdqh_groupby_key and the direct layout of sdqh_hash_build_unique produce); SDQH_ERR_UNSUPPORTED otherwise. */
int sdqh_table_select_keys(sdqh_ctx* ctx, const sdqh_table* table, int64_t min_hits, int value_index,
                           double lo, double hi, sdqh_table** out);

/* The aggregating probe whose output key is made of fields of the matched entry only, not of the
 * probe key (Q10: lineitem probes the order -> customer-fields dictionary and sums revenue per
 * customer record, test/test_all.py:524-541; emitted as the aggregating-dict loop
 * ...generator_par.py:402-440 keyed by `indexedDictValue` fields): entries of an accumulating table
 * whose payload fields fields[0..nfields) are all equal form ONE group.  After this call rows that
 * match any entry of a group (sdqh_hash_probe_aggregate) are added to the group's first entry
 * (lowest build row); the other entries keep zero hits, so K-F / top-k with min_hits >= 1 return one
 * entry per group.  Field i takes values in [lo[i], lo[i] + span[i]) (row references, dictionary
 * codes); an entry with a value outside keeps its own accumulators.  The product of the spans is
 * limited to SDQH_MAX_SHARE_CELLS (SDQH_ERR_UNSUPPORTED beyond).  Call before the first aggregate. */
#define SDQH_MAX_SHARE_CELLS (1ll << 28)
int sdqh_table_share_groups(sdqh_ctx* ctx, sdqh_table* table, int nfields, const int32_t* fields,
                            const int64_t* lo, const int64_t* span);

/* ---- K-F: compact the entries that received at least min_hits rows into host arrays ---------
 * out_keys[i], out_payload[p*capacity + i] (8 raw bytes each), out_values[v*capacity + i],
 * out_hits[i]; *out_n = rows written.  min_hits = 0 returns every entry.  Any out_* may be NULL;
 * with all four NULL the call only counts (capacity ignored), and a following call with the same
 * min_hits copies the already-compacted rows out. */
int sdqh_table_compact(sdqh_ctx* ctx, const sdqh_table* table, int64_t min_hits, int64_t capacity,
                       int64_t* out_keys, int64_t* out_payload, double* out_values,
                       int64_t* out_hits, int64_t* out_n);
/* K-F with the rows delivered BEHIND the call — the reference's result object also defers its conversion to Python values to
 * `to_dict()` (src/sdqlpy/fastd.py:31-51; the map itself is filled by K-F, ...generator_par.py:520-568, 871-877).  Same arguments
 * and results as sdqh_table_compact, with out_keys | out_payload | out_values | out_hits laid out back to back (in this order,
 * `capacity` rows per array) in ONE sdqh_host_alloc block: the call returns once *out_n is known; the rows arrive by a
 * device-to-host copy queued behind the kernels on a stream of its own, so the kernels of the next call run beside it.  The
 * arrays must not be read (or the block reused) before sdqh_result_wait — or sdqh_synchronize — has returned.  Any other layout,
 * or the option "async_result" = 0, makes it the synchronous call.  The CPU build fills the arrays before it returns. */
int sdqh_table_compact_async(sdqh_ctx* ctx, const sdqh_table* table, int64_t min_hits, int64_t capacity,
                             int64_t* out_keys, int64_t* out_payload, double* out_values, int64_t* out_hits, int64_t* out_n);
/* K-F with NOTHING waited for: as sdqh_table_compact_async, but the row count arrives behind the call too — *out_n, which must
 * lie in the same sdqh_host_alloc block, is -1 until the kernels have run; every array is copied out at its full `capacity`.
 * After sdqh_result_wait (or sdqh_synchronize): *out_n rows are valid; *out_n > capacity means the rows beyond were dropped
 * (fetch again with the capacity it names).  SDQH_ERR_UNSUPPORTED for any other layout or with "async_result" = 0.  Value slots the
 * aggregated tuple does not use (out_values[k][..] for k >= the table's value count) are NOT written and their contents are UNDEFINED —
 * a block that served another result holds that result's rows there; a caller that wants zeros writes them (abi.py hands the
 * engine views of the first nv arrays only and zeroes nothing). */
int sdqh_table_compact_deferred(sdqh_ctx* ctx, const sdqh_table* table, int64_t min_hits, int64_t capacity,
                                int64_t* out_keys, int64_t* out_payload, double* out_values, int64_t* out_hits, int64_t* out_n);
/* out_n is TWO cells there: out_n[1] is the result's DONE word — the copy stream writes 1 into it behind the copies (2 at once if it
 * cannot: wait with sdqh_result_wait) — so one result can be waited for by itself: sdqh_host_wait_word(ctx, &out_n[1], 1). */
/* Block until a 32-bit word of sdqh_host_alloc memory that a stream writes holds `value` (spins; falls back to sdqh_synchronize). */
int sdqh_host_wait_word(sdqh_ctx* ctx, const void* word, uint32_t value);
/* Wait for every result copy queued by sdqh_table_compact_async / _deferred on this context. */
int sdqh_result_wait(sdqh_ctx* ctx);

/* Result memory the device can write: when every out_* array of a first sdqh_table_compact call
 * (no count-only call before it) lies inside one sdqh_host_alloc block, the compaction kernel
 * stores the rows there itself - no staging copy, one synchronisation.  If the result does not fit
 * `capacity` the call returns SDQH_ERR_OVERFLOW with *out_n set; retry with a larger block.  The
 * block belongs to the caller and may outlive the context (sdqh_host_free accepts ctx = NULL).
 * Replaces nothing in the reference, whose results are built in process memory
 * (...generator_par.py:871-877). */
int  sdqh_host_alloc(sdqh_ctx* ctx, size_t bytes, void** out);
void sdqh_host_free(sdqh_ctx* ctx, void* block);

/* ---- top-k of the entries (ORDER BY ... LIMIT k on top of K-F) ---------------------------------
 * The k first entries with at least min_hits rows in the order given by `sort` (1..SDQH_MAX_SORT_KEYS
 * keys: the entry key, payload field `index`, accumulator `index` or the hit count; ascending or
 * descending; is_f64 says a payload field holds a double), ties broken by build-row order — a total
 * order.  Outputs as sdqh_table_compact with capacity = k: out_payload[p*k + i], out_values[v*k + i];
 * *out_n = rows written (<= k).  Not in the reference, whose Q3 returns the unsorted set
 * (test/test_all.py:174); BASELINE config 3 names it ("hash joins + top-k"). */
#define SDQH_MAX_TOPK 128
#define SDQH_MAX_SORT_KEYS 3
#define SDQH_SORT_KEY 0
#define SDQH_SORT_PAYLOAD 1
#define SDQH_SORT_VALUE 2
#define SDQH_SORT_HITS 3
typedef struct sdqh_sort_key { int32_t kind, index, descending, is_f64; } sdqh_sort_key;
int sdqh_table_topk(sdqh_ctx* ctx, const sdqh_table* table, int64_t min_hits, int k,
                    int nsort, const sdqh_sort_key* sort,
                    int64_t* out_keys, int64_t* out_payload, double* out_values, int64_t* out_hits, int64_t* out_n);

/* The entries of a table (every owner row: key, then the npayload payload fields) as resident
 * columns, for re-distribution without a host round trip.  out_cols[1 + npayload]. */
int sdqh_table_entries(sdqh_ctx* ctx, const sdqh_table* table, sdqh_column** out_cols, int64_t* out_rows);

/* ---- generalised lookups: multi-join chains (Q5, Q9) -------------------------------------------
 * Where a value comes from inside an emitted loop body: a column of the scanned row, or a field of
 * the entry matched by an earlier lookup (`indexedDictValue.x`, `tbl[key].x` -> `.at(key)`,
 * ...generator_par.py:85-96), optionally through extractYear (sdql_lib.py:341-342).  All values are
 * 8 raw bytes (int64, or the bits of a double). */
#define SDQH_SRC_COLUMN      0
#define SDQH_SRC_LOOKUP      1
#define SDQH_SRC_LOOKUP_YEAR 2   /* payload / 10000 */
typedef struct sdqh_source {
    int32_t kind;                /* SDQH_SRC_* */
    int32_t lookup;              /* SDQH_SRC_LOOKUP*: index of the lookup step */
    int32_t field;               /* SDQH_SRC_LOOKUP*: payload field of the matched entry */
    int32_t _pad;
    const sdqh_column* col;      /* SDQH_SRC_COLUMN */
} sdqh_source;

/* A lookup step, evaluated after the filter in order: the row survives only if `table` contains
 * the key (joinProbe / `tbl[key] != None`).  nkey = 2: composite record key, the two parts (each in
 * [0, 2^32)) packed as (key[0] << 32) | key[1] — the same packing a two-part build uses. */
#define SDQH_MAX_LOOKUP 3
typedef struct sdqh_lookup {
    const sdqh_table* table;
    int32_t nkey, _pad;
    sdqh_source key[2];
} sdqh_lookup;

/* Generalised K-B (...generator_par.py:331-369 with lookups 85-96): filter -> lookups -> unique
 * build keyed by 1 or 2 sources, payload fields from sources.  First (lowest) row wins. */
int sdqh_build(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
               int nlookups, const sdqh_lookup* lookups,
               int nkey, const sdqh_source* key, int npayload, const sdqh_source* payload,
               int accumulate, sdqh_table** out);

/* Generalised K-C over a small group domain (...generator_par.py:402-440 with lookups): filter ->
 * lookups -> group key of 1..2 parts (each in [0, 2^32-2]) from sources -> value tuple whose
 * operands a..d come from sources (F64 columns or F64 payload bits).  Up to SDQH_MAX_LOOKUP_GROUPS
 * groups; outputs as sdqh_groupby_small. */
#define SDQH_MAX_LOOKUP_GROUPS 256
int sdqh_lookup_aggregate(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
                          int nlookups, const sdqh_lookup* lookups,
                          int nkeys, const sdqh_source* keys, int tuple_shape, const sdqh_source* operands,
                          int max_groups, int64_t* out_keys, double* out_values, int64_t* out_counts,
                          int32_t* out_ngroups);
/* ... launched and NOT waited for (ABI 7), its groups left in `block` (sdqh_xgroupby_block_bytes() bytes, opaque) instead of returned:
 * device_block 0 — a result block from sdqh_host_alloc, collected with sdqh_xgroupby_collect (the key it returns packs the parts:
 * part k in bits [32k, 32k + 32)); the plan's last loop, deferred like sdqh_xgroupby_async, recordable into a plan graph;
 * device_block 1 — device memory (CPU build: host memory): this rank's partial groups on their way through an all-gather to
 * sdqh_xgroupby_fold, which merges the ranks' blocks by packed key in rank order (the multi-GPU runner's Q9: the loop the reference
 * emits for the final aggregation of a join chain, ...generator_par.py:402-440, over a row shard).  What the data decides — too many
 * groups, a key part outside [0, 2^32-2] — is reported by sdqh_xgroupby_collect. */
int sdqh_lookup_aggregate_block(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
                                int nlookups, const sdqh_lookup* lookups,
                                int nkeys, const sdqh_source* keys, int tuple_shape, const sdqh_source* operands,
                                void* block, int device_block);

/* ---- row programs: the open expression / predicate vocabulary (ABI 4) -----------------------------
 * What the reference's generator prints as the C++ body of an emitted loop — any boolean expression
 * (`and` / `or` printed as `*` / `+`, src/sdqlpy/lib/sdql_compiler.py:277-292), conditional values
 * (IfExpr, lib/sdql_ir.py:294-303), arithmetic on columns and looked-up fields (AdditionCodeGenerator
 * targets, lib/sdql_ir_cpp_generator_par.py:712-795), lookups `.at(k)` / `contains(k)` (85-96), the
 * VarChar methods (include/varchar.h:61-124) — travels here as DATA: a list of typed operations in
 * single-assignment form, operation i computing value i from earlier values.  The HIP build
 * specialises a hand-written kernel skeleton on the program at run time (hiprtc; code objects are
 * cached by program structure, as the reference caches its compiled module, sdql_lib.py:377-387);
 * the CPU build interprets it row by row.  Constants and column / table bindings are arguments of the
 * specialised kernel, not part of its code.
 *
 * A row is processed as: gates in order (all must be true; a later gate is not evaluated once one
 * fails, so a gate may guard the lookups of the next), then `key` and `vals`.  A FIELD / ACC of a lookup
 * that missed reads as 0.  Arithmetic is evaluated exactly as written (no reassociation, no FMA).
 */
#define SDQH_T_I64  0
#define SDQH_T_F64  1
#define SDQH_T_BOOL 2

#define SDQH_X_COL     1   /* col[row]; the type is the column's (I64 / F64)                                              */
#define SDQH_X_ROWID   2   /* the row number, i64                                                                         */
#define SDQH_X_CONST   3   /* imm_i (i64, bool) or imm_f (f64), by `type`                                                 */
#define SDQH_X_LOOKUP  4   /* `table` looked up by key value a (i64; a PACK2 value for a two-part key): bool = found       */
#define SDQH_X_FIELD   5   /* payload field `aux` of the entry matched by LOOKUP op a, read as `type` (8 raw bytes)       */
#define SDQH_X_ACC     6   /* accumulator `aux` (f64) of the entry matched by LOOKUP op a; aux = -1: its row count (i64)  */
#define SDQH_X_ADD    10   /* a + b   (both i64 or both f64)                                                              */
#define SDQH_X_SUB    11
#define SDQH_X_MUL    12
#define SDQH_X_DIV    13   /* f64 only                                                                                   */
#define SDQH_X_NEG    14
#define SDQH_X_I2F    15   /* (double)a                                                                                  */
#define SDQH_X_YEAR   16   /* a / 10000 on a yyyymmdd integer (extractYear, sdql_lib.py:341-342)                          */
#define SDQH_X_PACK2  17   /* (a << 32) | b for a, b in [0, 2^32): the packing of a two-part key; a part outside fails the call (SDQH_ERR_UNSUPPORTED) */
#define SDQH_X_DIVI   18   /* a / imm_i on i64 with imm_i > 0 (C division); unpacks a mixed-radix or PACK2 key read back from a table (sdqh_table_columns) */
#define SDQH_X_MODI   19   /* a % imm_i, likewise.  The divisor of both is part of the program's structure, not a rebindable constant          */
#define SDQH_X_LT     20   /* a < b  (both i64 or both f64) -> bool; NaN compares false, as in C                           */
#define SDQH_X_LE     21
#define SDQH_X_GT     22
#define SDQH_X_GE     23
#define SDQH_X_EQ     24
#define SDQH_X_NE     25
#define SDQH_X_AND    30   /* bool, bool -> bool                                                                          */
#define SDQH_X_OR     31
#define SDQH_X_NOT    32
#define SDQH_X_SELECT 33   /* a ? b : c  (b, c of one type)                                                               */
#define SDQH_X_STR    40   /* predicate `aux` (SDQH_STR_*) of STR column `col` against the constant str[0..slen): bool     */
#define SDQH_X_STRIDX 41   /* firstIndex(col, constant): position of the first occurrence, -1 if none (varchar.h:91-97), i64 */
#define SDQH_X_CHAR   42   /* code unit `aux` of STR column `col` (0 past the text), i64 — substr() as group-key parts     */

#define SDQH_STR_EQ 0      /* the modes of sdqh_spred.negate */
#define SDQH_STR_NE 1
#define SDQH_STR_CONTAINS 2
#define SDQH_STR_PREFIX 3
#define SDQH_STR_SUFFIX 4

#define SDQH_MAX_XOPS    96
#define SDQH_MAX_XCOLS   16   /* distinct columns a program may read                   */
#define SDQH_MAX_XTABLES 6    /* distinct tables it may look up                        */
#define SDQH_MAX_XGATES  16
#define SDQH_MAX_XSTR    256  /* code units of all string constants together          */
#define SDQH_MAX_XCONST  48   /* CONST operations of one type (i64 + bool / f64)       */

typedef struct sdqh_xop {
    int32_t code, type;             /* SDQH_X_*, SDQH_T_* of the result */
    int32_t a, b, c;                /* operand operation indices, -1 = unused */
    int32_t aux;
    int64_t imm_i;
    double  imm_f;
    const sdqh_column* col;         /* COL / STR / STRIDX / CHAR */
    const sdqh_table*  table;       /* LOOKUP */
    const uint32_t*    str;         /* STR / STRIDX: the constant's code units (host memory, read during the call) */
    int32_t slen, _pad;
} sdqh_xop;

typedef struct sdqh_program {
    int32_t nops, ngates;
    const sdqh_xop* ops;
    const int32_t*  gates;          /* bool operations */
    int32_t key;                    /* i64 operation: group / build / set key; -1 = none */
    int32_t nvals;
    const int32_t* vals;            /* summed values (f64; an i64 value is summed as an integer count) or payload fields (8 raw bytes) */
} sdqh_program;

/* K-A (...generator_par.py:258-291): out_values[v] = sum of value v over the passing rows, *out_count = passing rows.
 * nvals <= SDQH_TUPLE_MAX_VALUES, every value f64. */
int sdqh_xscan_sum(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, double* out_values, int64_t* out_count);
/* K-C over a small group domain (402-440): groups keyed by the program's key (any i64 >= 0), <= max_groups
 * (<= SDQH_MAX_LOOKUP_GROUPS) of them; outputs as sdqh_groupby_small with nkeys = 1. */
int sdqh_xgroupby(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, int max_groups,
                  int64_t* out_keys, double* out_values, int64_t* out_counts, int32_t* out_ngroups);
/* The same K-C with the result delivered BEHIND the call (a query's last loop need not hold the host: the next query's kernels
 * are queued while this one's run — the reference's result object also converts on `to_dict()`, src/sdqlpy/fastd.py:31-51).
 * sdqh_xgroupby_async launches and returns; `result_block` is sdqh_xgroupby_block_bytes() bytes from sdqh_host_alloc, opaque,
 * written when the stream gets there.  sdqh_xgroupby_collect waits for THIS call's kernels (a completion word in the block that the
 * stream writes behind them), then reads the groups out of it exactly as
 * sdqh_xgroupby would have returned them (same errors: SDQH_ERR_OVERFLOW with *out_ngroups, SDQH_ERR_UNSUPPORTED for a negative
 * key); nvals = the program's value count.  The CPU build computes the groups in the first call. */
size_t sdqh_xgroupby_block_bytes(void);
int sdqh_xgroupby_async(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, void* result_block);
int sdqh_xgroupby_collect(sdqh_ctx* ctx, const void* result_block, int nvals, int max_groups,
                          int64_t* out_keys, double* out_values, int64_t* out_counts, int32_t* out_ngroups);
/* ---- the small group-by of a row-sharded table, folded over the ranks ON THE DEVICE (ABI 6; no reference counterpart) ----
 * sdqh_xgroupby_partial: sdqh_xgroupby_async with the group table written to DEVICE memory the caller owns (`device_block`:
 * sdqh_xgroupby_block_bytes() bytes; CPU build: host memory) instead of a result block — this rank's partial groups, on their way
 * into a collective; nothing is waited for and nothing can be collected from it.
 * sdqh_xgroupby_fold: `nblocks` (<= SDQH_MAX_PARTS) such blocks back to back in device memory (the all-gathered ranks' blocks, rank
 * order) merged by key into `result_block` (from sdqh_host_alloc, as for sdqh_xgroupby_async: collected with sdqh_xgroupby_collect).
 * A group's sums are added block by block, ((rank 0 + rank 1) + rank 2) ...: every rank folds the same blocks in the same order and
 * gets the same bits.  The keys must mean the same thing on every rank (the caller's business: same dictionaries, same packing).
 * More than SDQH_MAX_LOOKUP_GROUPS groups over all blocks, or a block that reported them: SDQH_ERR_OVERFLOW at collect. */
int sdqh_xgroupby_partial(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, void* device_block);
int sdqh_xgroupby_fold(sdqh_ctx* ctx, const void* device_blocks, int nblocks, void* result_block);
/* K-B (331-369): unique build keyed by the program's key, payload = its vals (<= SDQH_MAX_PAYLOAD); first row wins.
 * [key_lo, key_hi]: bounds of the key the caller knows from its sources (key_lo > key_hi: none known) — a
 * dense range gets the direct (bitmap + rank) index, anything else open addressing.  A key outside given
 * bounds fails the call (SDQH_ERR_UNSUPPORTED).  accumulate: 0 = the entries carry no accumulators; 1 = room for
 * SDQH_TUPLE_MAX_VALUES sums per entry (what a later sdqh_*probe_aggregate adds is not known yet); 16 + n = exactly n sums
 * (n <= SDQH_TUPLE_MAX_VALUES) — a caller that knows its plan says so and the build clears n doubles per entry instead of four. */
int sdqh_xbuild(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, int64_t key_lo, int64_t key_hi, int accumulate, sdqh_table** out);
/* Membership-only K-B: the set of keys of the passing rows (exact bitmap over [key_lo, key_hi], which the
 * caller knows from the key's sources); SDQH_ERR_UNSUPPORTED if a key falls outside. */
int sdqh_xkey_set(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, int64_t key_lo, int64_t key_hi, sdqh_table** out);
/* Compaction by a row program (the probe side of a re-distributed join): EVERY passing row — duplicate keys
 * included, nothing is indexed — as resident columns: the program's key, then its vals (<= SDQH_MAX_PAYLOAD), each an I64
 * column holding the operation's 8 bytes (a double value keeps its bit pattern).  out_cols[1 + nvals]; the rows keep the
 * scan's order within a segment of the scan, the segments follow one another.  Replaces the filter + materialise half of
 * the reference's probe loop (...generator_par.py:378-447) when the rows have to change GPUs before they can probe. */
int sdqh_xcompact(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, sdqh_column** out_cols, int64_t* out_rows);
/* K-C large: for every passing row, the entry of `table` matched by LOOKUP operation `lookup_op` (which must
 * be among the gates) gets acc += vals, hits += 1 (the group is the matched entry, test/test_all.py:164-172). */
int sdqh_xprobe_aggregate(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, int lookup_op, sdqh_table* table);
/* The entries of a table with at least min_hits rows as resident columns — key, the npayload payload
 * fields, the SDQH_TUPLE_MAX_VALUES accumulators (F64), the hit count — so that a sum over a result
 * dictionary (K-F / HAVING with lookups, ...generator_par.py:520-568) is a scan like any other.
 * out_cols[1 + npayload + SDQH_TUPLE_MAX_VALUES + 1]; tables without accumulators yield zero columns there.
 * The rows are in build-row order.  HIP build: the columns are VIEWS of the table's own K-F buffers (nothing is copied; the
 * only host round trip is the row count): they stay valid until the table is released or compacted with another min_hits,
 * and must be freed with sdqh_column_free like any column (which leaves the table's memory alone). */
int sdqh_table_columns(sdqh_ctx* ctx, const sdqh_table* table, int64_t min_hits, sdqh_column** out_cols, int64_t* out_rows);
/* HIP build: number of kernels specialised so far in this process / how many of them came from the on-disk
 * cache; CPU build: zeros.  Diagnostics for tests and the bench. */
int sdqh_jit_stats(sdqh_ctx* ctx, int64_t* compiled, int64_t* from_cache);
/* HIP build: compile the source of a specialised kernel (as the library generated it on an earlier run: SDQLPY_AMD_JIT_RECIPES keeps them)
 * into the on-disk cache, loading nothing — works on a compile-only context, i.e. on a host without a GPU.  The reference's
 * counterpart is its compiled mode's module cache (src/sdqlpy/sdql_lib.py:372-387).  CPU build: accepted, nothing to do. */
int sdqh_jit_compile(sdqh_ctx* ctx, const char* source);

/* ---- multi-GPU redistribution helpers (no reference counterpart) -------------- */
/* Filter + semi-join probes, then gather `ncols` columns of the surviving rows into freshly
 * allocated device columns (order of rows unspecified but identical across the ncols outputs). */
#define SDQH_MAX_COMPACT_COLS 6
int sdqh_scan_compact(sdqh_ctx* ctx, int64_t nrows, const sdqh_filter* filter,
                      int nprobes, const sdqh_probe* probes,
                      int ncols, const sdqh_column* const* cols, sdqh_column** out_cols, int64_t* out_rows);
/* Reorder rows so that all rows with part(key) == p are contiguous, p ascending (row order inside a
 * part unspecified, identical across the ncols outputs).  range_upper == NULL: part(key) =
 * mix64(key) % nparts (the same function in both builds).  Otherwise range_upper[nparts-1] holds
 * ascending inclusive upper bounds and part(key) = number of bounds below key (keys above the
 * last bound go to the last part).  counts[nparts] on host.  nparts <= 64. */
#define SDQH_MAX_PARTS 64
int sdqh_partition_by_key(sdqh_ctx* ctx, int64_t nrows, const sdqh_column* key, int nparts, const int64_t* range_upper,
                          int ncols, const sdqh_column* const* cols, sdqh_column** out_cols,
                          int64_t* counts);
/* Say that a column lives for one run of a plan (rows that arrived through a collective): no twins, no dictionaries, no order facts are
 * built for it — each would be a pass over the column, or a host round trip, paid on every run.  CPU build: accepted, nothing to do. */
int sdqh_column_mark_transient(sdqh_ctx* ctx, sdqh_column* col);
/* Tell the library bounds of an I64 column the caller knows (lo <= every value <= hi; a superset of the true range is fine): builds
 * size their bitmaps and decide whether key parts pack from them instead of a minimum / maximum pass over the column and the host
 * round trip that reads it back — rows that arrived through a collective carry the bounds of the columns they were made from. */
int sdqh_column_set_bounds(sdqh_ctx* ctx, sdqh_column* col, int64_t lo, int64_t hi);
/* sdqh_partition_by_key straight into ONE caller-owned buffer (device memory; CPU build: host) of nrows * ncols 8-byte elements, laid
 * out for an all-to-all: the chunk for part p starts at element ncols * (rows of the parts before p) and holds every column's rows of
 * that part, column after column — so ONE collective moves every column of a redistribution step and nothing is copied between the
 * partitioning pass and the collective's buffer.  counts[nparts] on host. */
int sdqh_partition_pack(sdqh_ctx* ctx, int64_t nrows, const sdqh_column* key, int nparts, const int64_t* range_upper,
                        int ncols, const sdqh_column* const* cols, void* packed, int64_t* counts);
/* The receiving side: a buffer of nparts chunks laid out as above (chunk s: part_rows[s] rows of each of the ncols columns) taken apart
 * into ncols freshly allocated columns of sum(part_rows) rows, rows grouped by source in source order.  dtypes[ncols]: SDQH_I64 /
 * SDQH_F64.  Queued on the ctx stream under "async_copies" (the caller keeps `packed` alive until it synchronises). */
int sdqh_unpack_parts(sdqh_ctx* ctx, const void* packed, int nparts, const int64_t* part_rows, int ncols, const int* dtypes,
                      sdqh_column** out_cols, int64_t* out_rows);
/* ---- redistribution sized on the DEVICE (ABI 5): a hash- / range-partitioned join step with no host wait in it -----------------------
 * The calls above return row counts to the host, which then sizes the all-to-all: a host round trip per exchange.  The calls below move
 * FIXED-CAPACITY chunks instead — the capacity is the caller's bound (the previous run's counts plus slack) — with the true row count in
 * each chunk's header, so the counts travel with the data and nothing waits; a chunk that overflowed is found out afterwards
 * (SDQH_STAT_MAX_COUNT) and the step is repeated with exact sizes by the calls above.  No reference counterpart.
 *
 * sdqh_xstage: the rows of a table loop that pass the program's gates, STAGED on the device — key and vals (<= SDQH_MAX_PAYLOAD), 8
 * bytes each, equal keys included, nothing indexed (the filter half of the probe loop, ...generator_par.py:378-447).  The table it
 * returns answers nothing but sdqh_table_partition_pack and sdqh_table_free; how many rows it holds stays on the device. */
int sdqh_xstage(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, sdqh_table** out);
/* 8-byte words of one chunk of `ncols` columns and `chunk_rows` rows: a 2-word header {rows meant for this chunk, 0}, then column after
 * column, chunk_rows rows each. */
int64_t sdqh_chunk_words(int ncols, int64_t chunk_rows);
/* The entries of `table` — a unique build over a strictly increasing key column (no two staged rows share a key), or a stage from
 * sdqh_xstage — partitioned by their key (mix64(key) % nparts, or range_upper as in sdqh_partition_by_key) into the nparts chunks of
 * `packed` (nparts * sdqh_chunk_words(1 + npayload, chunk_rows) words of device memory; CPU build: host), chunk p at word
 * p * sdqh_chunk_words(...): the layout of an equal-split all-to-all.  A chunk's header counts every row meant for it; rows beyond
 * chunk_rows are dropped.  Round 6: a chunk's rows are placed DETERMINISTICALLY — the same bytes run after run (option "pack_ordered", 1:
 * every wave counts its rows per part, one workgroup per part scans the counts, every wave places its rows from its own cursor: three
 * launches, no atomics; 0: one launch, rows placed by racing atomics) — and with nparts = 1 (the send buffer of an ALL-GATHER: a
 * replicated build's entries) in the order the build met them.  Queued on the ctx stream; nothing is waited for.
 * SDQH_ERR_UNSUPPORTED: a table whose staged rows may repeat a key (use sdqh_table_entries + sdqh_partition_pack) — a unique build
 * qualifies when its key is a strictly increasing column, or two plain columns that strictly increase as pairs. */
int sdqh_table_partition_pack(sdqh_ctx* ctx, const sdqh_table* table, int nparts, const int64_t* range_upper, int64_t chunk_rows, void* packed);
/* The receiving side: nparts chunks (as received from the ranks, in rank order) -> ncols freshly allocated columns of
 * nparts * chunk_rows rows: the sources' rows back to back (min(header, chunk_rows) of each), then PADDING rows up to the capacity —
 * column 0 (the key) holds pad_key there, the other columns 0 — so that whatever consumes the columns can be launched on the capacity
 * without knowing the count (the caller picks a pad_key its consumer drops: a key no table holds, a key a gate rejects).  dtypes as
 * for sdqh_unpack_parts; the columns are transient.  `sent` (may be null): this rank's own packed send buffer of the same step, read for
 * its header counts only.  `stat`: an I64 column of >= SDQH_EXCHANGE_STAT_WORDS rows in device memory; exchange number `slot` (0 .. 3)
 * overwrites its own words of it (the others are left alone: a caller may keep one block for every run of a plan)
 *   stat[SDQH_STAT_MAX_COUNT + slot]            the largest header count among the chunks received and sent (> chunk_rows: rows were lost)
 *   stat[SDQH_STAT_DETAIL + 4 * slot + 0 .. 3]  rows received, rows sent to all parts, rows sent to part `self_part`, chunk_rows
 * The first four words are meant to be all-reduced (MAX) over the ranks: every rank then knows whether any chunk of the step
 * overflowed anywhere, and the largest count is the next run's bound.  Queued on the ctx stream; nothing is waited for. */
#define SDQH_EXCHANGE_STAT_WORDS 32
#define SDQH_STAT_MAX_COUNT 0
#define SDQH_STAT_DETAIL 8
int sdqh_unpack_chunks(sdqh_ctx* ctx, const void* packed, int nparts, int ncols, const int* dtypes, int64_t chunk_rows, int64_t pad_key,
                       const void* sent, int self_part, sdqh_column* stat, int slot, sdqh_column** out_cols);

/* ---- plan graphs (ABI 5): a prepared plan's device calls recorded once, replayed by ONE call -------------------------------------------
 * The reference compiles a query into one function (...generator_par.py:839-890); here a query is a dozen calls, each of which
 * analyses its arguments and launches a kernel or two — a third of a 0.6 ms step is the host issuing them.  Between sdqh_graph_begin
 * and sdqh_graph_end every launch, copy and fill the calls on `ctx` make is RECORDED instead of executed (HIP stream capture); nothing
 * may wait for the device in between (such a call fails with SDQH_ERR_UNSUPPORTED and the recording is dropped: sdqh_graph_abort).
 * What the recorded calls allocate from the context's pool — tables, staging buffers — belongs to the graph until sdqh_graph_free:
 * freeing a recorded table releases its handle, not its memory.  sdqh_graph_launch queues the whole recording on the ctx stream:
 * same kernels, same arguments, same addresses; the host words the recorded calls initialise (the completion words of
 * sdqh_table_compact_deferred / sdqh_xgroupby_async result blocks) are reset first, so the caller collects from the same blocks
 * exactly as after the recorded calls.  The caller guarantees what the recording assumed: the same resident columns with the same
 * contents, the same bound constants, result blocks collected before the next launch of the same graph.  CPU build: begin / end
 * record nothing and sdqh_graph_end returns SDQH_ERR_UNSUPPORTED (the caller keeps issuing the calls).
 * Inside a recording nothing about a column is MEASURED (bounds, order, 4-byte twins are computed on first request, with a wait): the
 * calls use what is known already, answer the way that needs no measurement, or refuse before touching the stream — on ROCm 7 a wait
 * on a capturing stream, or a query FROM ANY THREAD of an event recorded on that stream before its capture began, invalidates the
 * capture and leaves the stream unusable for good (tools/exp/capture_abort.hip).  sdqh_graph_abort joins what a refused call left forked,
 * ends the capture and returns SDQH_ERR_DEVICE if the runtime could not take it back (the context's stream is then lost: close it).
 * A caller that shares the stream with another library (the multi-GPU runner with torch / RCCL) keeps that library's event polling
 * away from it: sdqlpy_amd/dist.py `_coll`. */
typedef struct sdqh_graph sdqh_graph;
int sdqh_graph_begin(sdqh_ctx* ctx);
int sdqh_graph_end(sdqh_ctx* ctx, sdqh_graph** out);
int sdqh_graph_abort(sdqh_ctx* ctx);
int sdqh_graph_launch(sdqh_ctx* ctx, sdqh_graph* graph);
int sdqh_graph_nodes(const sdqh_graph* graph);      /* kernels, copies and fills the recording holds (-1: no graph) */
void sdqh_graph_free(sdqh_ctx* ctx, sdqh_graph* graph);

/* Device-to-device (CPU build: memcpy) copy of rows [row0, row0+nrows) of an I64/F64 column to or
 * from caller-owned memory of the same kind (e.g. a torch tensor used as a collective buffer).
 * Returns after the copy has completed — unless the option "async_copies" is 1: then the copy is
 * only queued on the ctx stream and the caller calls sdqh_synchronize once per batch of copies (and
 * keeps a copy_in source alive until then). */
int sdqh_column_copy_out(sdqh_ctx* ctx, const sdqh_column* col, int64_t row0, int64_t nrows, void* dst);
int sdqh_column_copy_in(sdqh_ctx* ctx, sdqh_column* col, int64_t row0, int64_t nrows, const void* src);
/* Exact key bitmap of a table over [lo, hi] (bit i = key lo+i present), as device words.  If
 * *out_words is NULL on entry a column of ceil(bits/64) I64 rows is allocated; otherwise the given
 * I64 column (at least that long, e.g. a wrapped collective buffer) is cleared and filled.  Works on every table that knows its
 * keys: staged entries, or a bitmap of its own (key sets from sdqh_build_key_set / sdqh_xkey_set / sdqh_table_from_bitmap, the
 * direct layout).  With the option "async_copies" the export is only queued on the ctx stream, like the column copies above. */
int sdqh_table_export_bitmap(sdqh_ctx* ctx, const sdqh_table* table, int64_t lo, int64_t hi, sdqh_column** out_words);
/* The two parts of a column of packed composite keys ((hi << 32) | lo, both parts in [0, 2^32)) as columns of their own: a
 * replicated composite-key table is rebuilt from its all-gathered entries through sdqh_build with a two-part key, which gives it the
 * layouts the final loops are tuned for.  The first nrows rows; the caller frees both columns. */
int sdqh_column_unpack2(sdqh_ctx* ctx, const sdqh_column* packed, int64_t nrows, sdqh_column** out_hi, sdqh_column** out_lo);
/* Build a key-only membership table from a bitmap (device I64 column viewed as 32-bit words). */
int sdqh_table_from_bitmap(sdqh_ctx* ctx, const sdqh_column* words, int64_t lo, int64_t hi, sdqh_table** out);

#ifdef __cplusplus
}
#endif
#endif /* SDQH_H */
