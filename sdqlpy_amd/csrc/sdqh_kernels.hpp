// sdqh_kernels.hpp — hand-written HIP kernels for gfx950 (MI355X / CDNA4) behind the sdqh C ABI.
//
// Every kernel here is HBM-bound integer / f64 work: no MFMA.  The rules that matter are the
// streaming ones: 16-byte-per-lane coalesced loads (1 KiB per wave instruction), >= 1024
// workgroups of 256 threads (4 wave64 per workgroup) so all 256 CUs / 8 XCDs stay full, all loads
// of a tile issued before the first use, group-by state kept in registers / LDS, deterministic
// two-level reductions instead of global float atomics where a result is a handful of sums.
//
// Reference loop shapes these replace (edin-dal/sdqlpy, src/sdqlpy/lib/sdql_ir_cpp_generator_par.py):
//   k_scan_sum, k_scan_probe_sum     K-A 258-291 (the latter with `tbl[k] != None` conditions, lookups 85-96)
//   k_groupby_reg / k_groupby_lds    K-C 402-440, small key domain
//   k_stage (+ index kernels)        K-B 331-369, unique builds; k_key_set: builds that only answer membership
//   k_build_lookup                   K-B with keys / payloads derived through lookups (multi-join chains)
//   k_probe_agg                      K-C 402-440, group = matched entry; also the row-keyed group-by (k_gk_layout)
//   k_lookup_agg                     K-C 402-440 with chained lookups, <= 256 groups
//   k_compact_*, k_topk_*            K-F 520-568; ORDER BY ... LIMIT k on top of it (not in the reference)
//   k_select_keys                    a conditional sum over an aggregated dictionary (HAVING)
//   k_seg_scan ... k_export_bitmap   redistribution helpers of the multi-GPU plans (no reference counterpart)
#pragma once
#ifndef __HIPCC_RTC__           // hiprtc (the run-time specialiser, sdqh_jit.hpp) brings its own runtime declarations
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif
#include "sdqh.h"

// Kernels that are not templates are compiled by hipcc into the library.  When this header is parsed by
// hiprtc for a run-time specialised kernel (sdqh_jit.hpp) they become uninstantiated templates: parsed,
// never compiled, so a specialisation costs a fraction of a second.
#if defined(__HIPCC_RTC__) || defined(SDQH_DECLS_ONLY)      // SDQH_DECLS_ONLY: sdqh_x.hip wants the argument structs, not a second copy of the kernels
#define SDQH_KERNEL template <int SDQH_RTC_UNUSED = 0> __global__
#else
#define SDQH_KERNEL __global__
#endif

namespace sdqh {

constexpr int TPB = 256;                 // threads per workgroup = 4 wave64
constexpr int WAVE = 64;
constexpr int ROWS_PER_LOAD = 2;         // two 8-byte rows = one 16-byte load per lane
#ifndef SDQH_UNROLL
#define SDQH_UNROLL 2
#endif
#ifndef SDQH_NT_LOADS
#define SDQH_NT_LOADS 1                  // streamed columns are read once: non-temporal loads (+7 % on MI355X, tools/microbench_q1.hip)
#endif
#ifndef SDQH_TILE_CHUNK
#define SDQH_TILE_CHUNK 4                // consecutive tiles a workgroup takes per step of its stride loop (32 KiB runs per column)
#endif
constexpr int UNROLL = SDQH_UNROLL;      // 16-byte loads per column per thread per tile
constexpr int TILE_ROWS = TPB * ROWS_PER_LOAD * UNROLL;     // 1024 rows per workgroup step
constexpr int SUB_ROWS = TPB * ROWS_PER_LOAD;                // 512 rows per unrolled sub-step

constexpr int64_t EMPTY_KEY = INT64_MIN;  // empty-slot sentinel; a real key of this value uses the extra slot
constexpr uint32_t NO_ROW = 0xFFFFFFFFu;
constexpr uint64_t EMPTY_GROUP = ~0ull;

// ---- by-value kernel arguments -----------------------------------------------------------------
struct DevFilter {
    int32_t ni, nf, ns, sneg;
    const int64_t* ic[SDQH_MAX_IPRED]; int64_t ilo[SDQH_MAX_IPRED], ihi[SDQH_MAX_IPRED];
    const double* fc[SDQH_MAX_FPRED]; double flo[SDQH_MAX_FPRED], fhi[SDQH_MAX_FPRED];
    const uint32_t* sc; int32_t swidth, slen;
    const uint8_t* sc8;       // the text column's one-byte-per-code-unit twin (null: none); the LDS-staged predicate reads it instead of sc
    uint32_t sval[SDQH_MAX_STR_CONST];
    // ranges on tuple operand slots (an f-predicate whose column is also a value operand is
    // checked on the already-loaded operand instead of being loaded twice)
    // column-vs-column comparisons (generic filter instances only): a op b on raw 8-byte values
    int32_t nc, _padc; const int64_t* ca[SDQH_MAX_CPRED]; const int64_t* cb[SDQH_MAX_CPRED]; int32_t cop[SDQH_MAX_CPRED], cf64[SDQH_MAX_CPRED];
    uint32_t omask, slds;     // slds = 64 or 32: the launch carries slds * swidth words of dynamic LDS per wave (str_stage_mask)
    double olo[4], ohi[4];
};

struct DevTuple {
    const double* op[4];
};

struct TableHeader {          // lives in device memory: sized on the device, no host round trip
    uint64_t cap_mask;        // capacity - 1 (capacity is a power of two); slot[capacity] is the EMPTY_KEY slot
    uint64_t staged;          // rows that survived the build-side filter (>= distinct keys)
    uint64_t distinct;        // direct layout: number of distinct keys (set bits), written by k_rank_words
    uint32_t has_dups;        // a build row met its own key already in the table
    uint32_t _pad;
    uint64_t counted;         // filled by k_count (table_size)
    uint32_t _pad3;
    uint32_t _pad2;
};

// A built table maps a key to the stage index of the build row that owns the entry; everything else
// about an entry — payload, hit counter, accumulators — lives in the stage arrays at that index,
// which are dense per wave segment: they are initialised with coalesced stores while staging and
// the final compaction walks them instead of a slot array.  Two index layouts:
//
//   direct  (key range dense enough for an exact bitmap over [bm_lo, bm_hi]):
//           bm[w]       bit per key                         (set while staging)
//           wprefix[w]  rank of the first key of word w: set bits before it inside its 2048-word
//                       block + a base the block claimed with one atomicAdd (a bijection onto
//                       [0, distinct), not monotone across blocks)
//           dense_ref[rank(key)] = stage index                (plain stores)
//   dense   (an unfiltered build on a dense key range — the reference's `dense(N, key)` hint,
//           ...generator_par.py:191-224): dense_arr[key - lo] = build row, the source columns
//           themselves serve as the stage (nothing is copied), a lookup is one load
//   hash    (any int64 keys): open addressing, linear probing, capacity = pow2 >= 2 * entries
//           keys[s]     the key, EMPTY_KEY if free        (reset by k_clear)
//           rowref[s]   stage index of the owning row     (written by the claimer)
//
//   shits[i]    rows aggregated into entry i   (stage-indexed)
//   sacc[i*4+k] accumulators of entry i        (stage-indexed)
constexpr int RANK_BLOCK_WORDS = 4096;        // bitmap words per prefix block (one workgroup: 4 rows of 256 16-byte quads): every block claims its base with ONE
                                              // returning atomic on hdr->distinct, and those serialise on the one address (~20 ns each: 916 blocks of 2048 words
                                              // over Q3's 60 M-key range were 20 us of a kernel that moves 7.5 MB; 4096 against 8192: the same on the big bitmaps,
                                              // 9.5 -> 7 us on the small ones)

// ROW INDEX (builds keyed by a strictly increasing column, staged in row order: the entries of a segment carry increasing keys, and the
// segments' key ranges follow one another).  The stage kernel itself writes, for every bitmap word, the STAGE ROW of the word's first
// key (DevStage::wrow), so the stage row of a key is   wrow[word] + set bits below it in the word   and no rank pass, no owner array
// and no insert pass exist (k_rank_words + k_insert_direct: 16 us of Q3's 235, two launches).  The one complication: the keys of a word can
// belong to TWO consecutive segments (the end of one, the start of the next — never three: a segment's rows span at least 127 key
// values), whose stage rows do not follow one another.  For such a word wrow holds ROW_INDEX_EXC | s (s = the later segment) and
// wexc[s] says where each half lives; a lookup into it costs one more load (k_wrow_fixup finds these words after the build).
constexpr uint32_t ROW_INDEX_EXC = 0x80000000u;
struct WordExc { int64_t key0; uint32_t pos0, pos_prev, n0, _pad; };   // keys below key0: pos_prev + bits below; from key0 on: pos0 + (bits below - n0)
struct SegFirst { int64_t key; uint32_t pos, _pad; };                   // a segment's first entry (whether it opens its word is only known once the bitmap is complete)

struct DevTable {
    int64_t* keys;
    uint32_t* rowref;
    TableHeader* hdr;
    uint32_t* shits;
    double* sacc;             // null when the table carries no accumulators
    const uint32_t* bm;       // key bitmap over [bm_lo, bm_hi], or null
    const uint32_t* wprefix;
    uint32_t* dense_ref;
    int64_t bm_lo, bm_hi;
    int32_t bitmap_only;
    int32_t bm_shift;         // 0: bm is exact over the key (direct layout); 32: composite key, bm covers
                              //    the high part only (a pre-filter in front of the hash layout)
    const int64_t* pay[SDQH_MAX_PAYLOAD];   // stage payload arrays (entry payload by stage index)
    uint32_t* dense_arr;      // dense layout: dense_arr[key - bm_lo] = build row (NO_ROW if absent), bm == null
    int64_t lin_rb, lin_b0;   // lin_rb != 0: composite key (a << 32 | b) with a small a-range x b-range: bm is exact over the
                              //    linearised offset (a - bm_lo) * lin_rb + (b - lin_b0) (direct layout, bm_shift == 0)
    const uint32_t* alias;    // sdqh_table_share_groups: stage row -> the stage row whose accumulators it uses, or null
    int32_t acc_stride, _pad3; // doubles per entry in sacc: 4, or the tuple's value count when it is known at build time (sdqh_groupby_key)
    const struct WordExc* wexc; // direct layout whose wprefix[] holds STAGE ROWS (see DevStage::wrow), not ranks: the words that straddle two segments
    int64_t* slots;           // hash layout, packed form (tables with payload): slot h = { key, payload 0, payload 1, stage row } in 32
                              //    bytes, so a probe that hits finds the key, the owner and the first two payload fields in ONE
                              //    cache line instead of four (keys[], rowref[], pay[0][], pay[1][]); keys / rowref are null then
    // GROUPED layout (composite key (a << 32 | b) built from a table whose rows come in non-decreasing order of a — partsupp by part key):
    // the entries of one a are consecutive stage rows (a run; a run that crosses a wave segment continues at the next segment's base), so
    // the index is grp_first[a - bm_lo] = stage row of the run's first entry (NO_ROW: none), written by the stage kernel itself with a
    // fire-and-forget atomicMin — no hash slots, no CAS (a returning CAS on scattered lines runs at ~11 G/s on this chip whatever the
    // table's size: 50 us for Q9's 430 K entries, tools/microbench_insert.hip) — and a lookup walks the run comparing whole keys (first
    // match = lowest build row: the reference's first-insert-wins).  grp_key = the stage's key array; a segment that is not full ends in
    // an EMPTY_KEY entry: the walk goes on at the next segment's base.  bm (bm_shift 32) stays the pre-filter over a.
    const uint32_t* grp_first;
    const int64_t* grp_key;           // keys by stage row, grp_kstride apart: the stage's key array (1) or the (key, payload 0) pairs a build with payload writes
                                      //    beside it (2: a hit finds its first payload field in the line of its key — one line per lookup less in Q9's final loop)
    int64_t grp_kstride;
    int64_t grp_seg_rows, grp_cap;    // rows per stage segment; stage rows a walk may read (the build's rows + 1: the last segment's end mark)
    // rank = row layout (k_rank_increasing): per bitmap word { wprefix[w] (low half), bm[w] (high half) } side by side — a lookup's two requests are ONE
    // line where they were two (Q9's final loop fetches a 128-byte line per request: three per surviving row went to the orders table); bm and
    // wprefix stay what they are for everything that reads the bitmap alone.  Null: the table has no such copy.
    const unsigned long long* wpair;
    // HASHED FILTER in front of the hash layout (round 6): a bitmap of (capacity << HF_SHIFT) bits — 8 to 16 per staged key, L2-sized where the
    // slots are not — in which every key sets TWO bits of ONE word (hf_code below).  A loop tests it on streamed registers exactly like an
    // exact key bitmap (x_queue8's prefilter) and sends only the keys that pass — the hits and a few per cent of the rest — to the slots.
    // Sized with the capacity on the device (k_clear), filled by k_insert; null: none (direct / grouped layouts, option "hash_filter" 0).
    uint32_t* hf;
};
constexpr int HF_SHIFT = 2;
constexpr uint32_t HF_POS_MASK = 0x07FFFFFFu;                          // bit position: at most 2^27 bits (16 MB); the second bit's place in the word rides in the top five bits
// A filter helps while it stays in an XCD's L2 (4 MB) and its build's atomics stay cheap: tables of more than HF_MAX_CAP slots get the DEGENERATE
// filter — 128 bits, all set: every key passes, nothing is set per key (Q9's 15 M orders: 1.3 ms of scattered atomics for a 16 MB filter no L2 holds)
constexpr uint64_t HF_MAX_CAP = 1ull << 22;
constexpr uint32_t HF_DEGENERATE = 127u;
__device__ __forceinline__ uint32_t hf_mask_of(uint64_t cap_mask) { return cap_mask + 1 > HF_MAX_CAP ? HF_DEGENERATE : (uint32_t)(((cap_mask + 1) << HF_SHIFT) - 1); }
__device__ __forceinline__ uint32_t hf_raw(int64_t key);               // 32 hash bits of a key (defined behind mix64)
__device__ __forceinline__ uint32_t hf_code(uint32_t raw, uint32_t mask) { return (raw & mask) | (raw & ~HF_POS_MASK); }
__device__ __forceinline__ uint32_t hf_bits(uint32_t code) { return (1u << (code & 31u)) | (1u << (code >> 27)); }
__device__ __forceinline__ bool hf_test(uint32_t word, uint32_t code) { return ((word >> (code & 31u)) & (word >> (code >> 27)) & 1u) != 0u; }

// regions to set to a byte value each (k_fill; also the preamble of a build kernel that runs as ONE workgroup: see fill_in_block)
constexpr int FILL_MAX = 6;
struct DevFill { void* p[FILL_MAX]; uint64_t bytes[FILL_MAX]; uint32_t word[FILL_MAX]; int32_t n, _pad; };
// k_fill's own, longer list: a fill launch also carries the regions later builds of the plan will want cleared (fill-ahead, sdqh_hip.hip)
constexpr int FILL_BIG = 32;
struct DevFillBig { void* p[FILL_BIG]; uint64_t bytes[FILL_BIG]; uint32_t word[FILL_BIG]; int32_t n, _pad; };
// The fill of a tiny table's build, done by the build kernel's own (single) workgroup: a table over a few hundred rows is
// four launches of ~8 us each (fill, stage, rank, insert) for microseconds of work — two this way (the other pair: k_index_small)
__device__ __forceinline__ void fill_in_block(const DevFill& f) {
#pragma unroll
    for (int r = 0; r < FILL_MAX; ++r) {
        if (r >= f.n) break;
        const uint32_t w = f.word[r];
        const uint64_t n16 = f.bytes[r] / 16, n4 = f.bytes[r] / 4;
        uint4* p16 = static_cast<uint4*>(f.p[r]);
        for (uint64_t i = threadIdx.x; i < n16; i += blockDim.x) p16[i] = make_uint4(w, w, w, w);
        uint32_t* p4 = static_cast<uint32_t*>(f.p[r]);
        for (uint64_t i = n16 * 4 + threadIdx.x; i < n4; i += blockDim.x) p4[i] = w;
    }
}

// hash-layout accessors (separate arrays, or the packed 32-byte slots)
__device__ __forceinline__ int64_t slot_key(const DevTable& t, uint64_t h) { return t.slots ? t.slots[h * 4] : t.keys[h]; }
__device__ __forceinline__ uint32_t slot_row(const DevTable& t, uint64_t h) { return t.slots ? (uint32_t)t.slots[h * 4 + 3] : t.rowref[h]; }

// offset of `key` in a bitmap described by bm_lo / bm_hi / bm_shift / lin_rb / lin_b0 (DevTable or DevStage); false = out of range
template <class T>
__device__ __forceinline__ bool bm_locate(const T& t, int64_t key, uint64_t& off) {
    const int64_t rb = t.lin_rb;
    const int64_t v = (t.bm_shift || rb) ? (int64_t)((uint64_t)key >> 32) : key;
    if (v < t.bm_lo || v > t.bm_hi) return false;
    off = (uint64_t)(v - t.bm_lo);
    if (rb) {
        const int64_t b = (int64_t)((uint64_t)key & 0xFFFFFFFFull) - t.lin_b0;
        if (b < 0 || b >= rb) return false;
        off = off * (uint64_t)rb + (uint64_t)b;
    }
    return true;
}

template <int SHAPE> struct TupleTraits;
template <> struct TupleTraits<SDQH_TUPLE_A>          { static constexpr int NOPS = 1, NV = 1; };
template <> struct TupleTraits<SDQH_TUPLE_AB>         { static constexpr int NOPS = 2, NV = 1; };
template <> struct TupleTraits<SDQH_TUPLE_A_1MB>      { static constexpr int NOPS = 2, NV = 1; };
template <> struct TupleTraits<SDQH_TUPLE_PRICING>    { static constexpr int NOPS = 4, NV = 4; };
template <> struct TupleTraits<SDQH_TUPLE_A_1MB_M_CD> { static constexpr int NOPS = 4, NV = 1; };
template <> struct TupleTraits<SDQH_TUPLE_COUNT>      { static constexpr int NOPS = 0, NV = 0; };

// Products in the reference's association order; the build uses -ffp-contract=off so none of
// these becomes an FMA (reference test/test_all.py:52,171,293,480).
template <int SHAPE>
__device__ __forceinline__ void tuple_eval(const double (&x)[4], double (&o)[4]) {
    if constexpr (SHAPE == SDQH_TUPLE_A) { o[0] = x[0]; }
    else if constexpr (SHAPE == SDQH_TUPLE_AB) { o[0] = x[0] * x[1]; }
    else if constexpr (SHAPE == SDQH_TUPLE_A_1MB) { o[0] = x[0] * (1.0 - x[1]); }
    else if constexpr (SHAPE == SDQH_TUPLE_PRICING) {
        double dp = x[1] * (1.0 - x[2]);
        o[0] = x[0]; o[1] = x[1]; o[2] = dp; o[3] = dp * (1.0 + x[3]);
    }
    else if constexpr (SHAPE == SDQH_TUPLE_A_1MB_M_CD) { o[0] = x[0] * (1.0 - x[1]) - x[2] * x[3]; }
}

// ---- small helpers -----------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31; return x;
}
__device__ __forceinline__ uint32_t hash_key(int64_t k) {          // Fibonacci hashing, upper 32 bits
    return (uint32_t)(((uint64_t)k * 0x9E3779B97F4A7C15ull) >> 32);
}
__device__ __forceinline__ uint32_t hf_raw(int64_t key) { return (uint32_t)(mix64((uint64_t)key) >> 32); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

template <class T> struct Pair { T x, y; };

// ---- compile-time specialisation of the row filter / group key layout -----------------------------
// A kernel that decides at run time how many predicate columns it has ends up with its loads
// inside uniform branches, each followed by s_waitcnt vmcnt(0): the loads of a tile serialise and
// the stream stalls.  So the hot kernels are instantiated for the filter layouts the TPCH loops
// use (counts known at compile time: every load of a tile is issued up front) plus one generic
// instance (-1 = read the count from the arguments) that keeps every other query correct.
template <bool B> struct BoolC { static constexpr bool value = B; };
template <int NI_, int NF_, int NS_, int NP_, int NC_ = 0> struct FCfg { static constexpr int NI = NI_, NF = NF_, NS = NS_, NP = NP_, NC = NC_; };
using FGeneric = FCfg<-1, -1, -1, -1>;
template <class FC> __device__ __forceinline__ int cfg_ni(const int32_t n) { if constexpr (FC::NI >= 0) return FC::NI; else return n; }
template <class FC> __device__ __forceinline__ int cfg_nf(const int32_t n) { if constexpr (FC::NF >= 0) return FC::NF; else return n; }
template <class FC> __device__ __forceinline__ int cfg_ns(const int32_t n) { if constexpr (FC::NS >= 0) return FC::NS; else return n; }
template <class FC> __device__ __forceinline__ int cfg_np(const int32_t n) { if constexpr (FC::NP >= 0) return FC::NP; else return n; }
// column-vs-column predicates exist in the generic instance only (the dispatchers send filters that carry one there)
template <class FC> __device__ __forceinline__ int cfg_nc(const int32_t n) { if constexpr (FC::NI >= 0) return FC::NC; else return n; }
// group-key slots: 0 absent, 1 string(1) (UCS4 code unit), 2 int64, -1 decided at run time
template <int K0_, int K1_> struct KCfg { static constexpr int K0 = K0_, K1 = K1_; };
using KGeneric = KCfg<-1, -1>;

// Two consecutive 8-byte rows.  Full tiles: one aligned 16-byte load.  Tail: clamped scalar loads.
template <bool TAIL, class T>
__device__ __forceinline__ Pair<T> load2(const T* __restrict__ p, int64_t r, int64_t nrows) {
    Pair<T> v;
    if constexpr (!TAIL) {
        using V = T __attribute__((ext_vector_type(2)));
#if SDQH_NT_LOADS
        V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(p + r));
#else
        V t = *reinterpret_cast<const V*>(p + r);
#endif
        v.x = t.x; v.y = t.y;
    } else {
        int64_t r0 = r < nrows ? r : nrows - 1, r1 = r + 1 < nrows ? r + 1 : nrows - 1;
        v.x = p[r0]; v.y = p[r1];
    }
    return v;
}

// ---- narrow twins ------------------------------------------------------------------------------------------
// A streamed column may have a 4-byte twin on the device: an I64 column whose values fit int32 (dates, keys), an F64
// column whose every value v is reproduced BIT FOR BIT by narrow_decode(llrint(v * 100)) (prices, quantities, rates:
// two-decimal numbers).  The twin is built, and every row of it verified against the column with the very function
// the kernels decode with, the first time a streaming kernel wants it (sdqh_hip.hip: ensure_narrow); a column that
// fails keeps its 8-byte form.  The kernels' NW instances read 8 bytes per row pair instead of 16: the algorithmic
// bytes (the reference's widths) stay what they are, the physical bytes halve.
__device__ __forceinline__ double narrow_decode(int32_t n) {
    const double x = (double)n;
    const double q = x * 0.01;                                       // within an ulp or two of x / 100 ...
    const double r = __builtin_fma(-q, 100.0, x);                    // ... and one correction step lands on it (checked per row when the twin is built)
    return __builtin_fma(r, 0.01, q);
}
template <bool TAIL, bool NW, class T>
__device__ __forceinline__ Pair<T> loadc(const T* __restrict__ p, int64_t r, int64_t nrows) {
    if constexpr (!NW) return load2<TAIL>(p, r, nrows);
    else {
        const int32_t* __restrict__ q = reinterpret_cast<const int32_t*>(p);
        int32_t a, b;
        if constexpr (!TAIL) {
            using V = int32_t __attribute__((ext_vector_type(2)));
#if SDQH_NT_LOADS
            const V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(q + r));
#else
            const V t = *reinterpret_cast<const V*>(q + r);
#endif
            a = t.x; b = t.y;
        } else {
            const int64_t r0 = r < nrows ? r : nrows - 1, r1 = r + 1 < nrows ? r + 1 : nrows - 1;
            a = q[r0]; b = q[r1];
        }
        Pair<T> v;
        if constexpr (sizeof(T) == 8 && T(0.5) == T(0)) { v.x = (T)a; v.y = (T)b; }      // integer column
        else { v.x = (T)narrow_decode(a); v.y = (T)narrow_decode(b); }
        return v;
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}
__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}

// VarChar::operator==(const wchar_t*) (reference include/varchar.h:61-77): first `len` code units
// equal, the rest of the fixed-width field zero.
__device__ __forceinline__ bool str_equal(const uint32_t* __restrict__ s, int width, const uint32_t* val, int len) {
    if (len > width) return false;
    for (int k = 0; k < len; ++k) if (s[k] != val[k]) return false;
    for (int k = len; k < width; ++k) if (s[k] != 0u) return false;
    return true;
}
// VarChar::contains -> wcsstr (reference include/varchar.h:84-89): the field ends at its first NUL
__device__ __forceinline__ bool str_contains(const uint32_t* __restrict__ s, int width, const uint32_t* val, int len) {
    if (len == 0) return true;
    int n = 0;
    while (n < width && s[n] != 0u) ++n;
    for (int st = 0; st + len <= n; ++st) {
        if (s[st] != val[0]) continue;
        int k = 1;
        while (k < len && s[st + k] == val[k]) ++k;
        if (k == len) return true;
    }
    return false;
}
// the string predicate of a filter: mode 0 `==`, 1 `!=`, 2 substring
// startsWith (reference include/varchar.h:99-110): every unit of the needle matches and no NUL comes first
template <class T>
__device__ __forceinline__ bool str_prefix(const T* s, int width, const uint32_t* val, int len) {
    if (len > width) return false;
    for (int k = 0; k < len; ++k) if (s[k] == 0u || s[k] != val[k]) return false;
    return true;
}
// endsWith: the text (field up to its first NUL) ends with the needle
template <class T>
__device__ __forceinline__ bool str_suffix(const T* s, int width, const uint32_t* val, int len) {
    int n = 0;
    while (n < width && s[n] != 0u) ++n;
    if (len > n) return false;
    for (int k = 0; k < len; ++k) if (s[n - len + k] != val[k]) return false;
    return true;
}
__device__ __forceinline__ bool str_pred(const uint32_t* __restrict__ s, int width, const uint32_t* val, int len, int mode) {
    if (mode == 2) return str_contains(s, width, val, len);
    if (mode == 3) return str_prefix(s, width, val, len);
    if (mode == 4) return str_suffix(s, width, val, len);
    return str_equal(s, width, val, len) != (mode != 0);
}
// `a[r] op b[r]` on two columns of one type (ints compare as int64, doubles as double)
__device__ __forceinline__ bool col_cmp(const DevFilter& f, int i, int64_t x, int64_t y) {
    bool lt, eq;
    if (f.cf64[i]) { const double a = __longlong_as_double(x), b = __longlong_as_double(y); lt = a < b; eq = a == b; }
    else { lt = x < y; eq = x == y; }
    const int op = f.cop[i];
    return op == SDQH_CMP_LT ? lt : (op == SDQH_CMP_LE ? (lt || eq) : (op == SDQH_CMP_EQ ? eq : !eq));
}
__device__ __forceinline__ bool col_pred(const DevFilter& f, int i, int64_t r) { return col_cmp(f, i, f.ca[i][r], f.cb[i][r]); }
template <class FC>
__device__ __forceinline__ bool col_preds(const DevFilter& f, int64_t r) {
    bool p = true;
#pragma unroll
    for (int i = 0; i < SDQH_MAX_CPRED; ++i) if (i < cfg_nc<FC>(f.nc) && p) p = col_pred(f, i, r);
    return p;
}

// ---- the string predicate on fields staged in LDS -------------------------------------------------
// Branch-free over the whole fixed width, so every LDS read of a field is independent of the
// comparisons (the per-lane early-exit loops above chain one read latency per character).
// equality: first `len` units equal, the rest zero.
template <class T>
__device__ __forceinline__ bool lds_str_equal(const T* s, int width, const uint32_t* val, int len) {
    if (len > width) return false;
    uint32_t diff = 0;
#pragma unroll 4
    for (int k = 0; k < width; ++k) diff |= (uint32_t)s[k] ^ (k < len ? val[k] : 0u);
    return diff == 0;
}
// substring: one pass over the field builds two position masks per 32 units — "equals the first
// unit of the needle" and "is NUL" — with no branch and no dependence between the LDS reads; only
// the (few) first-unit hits before the first NUL are then verified.  wcsstr semantics: the field
// ends at its first NUL (reference include/varchar.h:84-89).
template <class T>
__device__ __forceinline__ bool lds_str_contains(const T* s, int width, const uint32_t* val, int len) {
    const uint32_t v0 = val[0];
    bool found = false, ended = false;
    for (int w0 = 0; w0 < width && !ended && !found; w0 += 32) {          // per-lane state, usually one or two rounds
        uint32_t first = 0, nul = 0;
        const int nk = min(32, width - w0);
        if (nk == 32) {
#pragma unroll
            for (int k = 0; k < 32; ++k) { const uint32_t c = s[w0 + k]; first |= (c == v0 ? 1u : 0u) << k; nul |= (c == 0u ? 1u : 0u) << k; }
        } else {
#pragma unroll 4
            for (int k = 0; k < nk; ++k) { const uint32_t c = s[w0 + k]; first |= (c == v0 ? 1u : 0u) << k; nul |= (c == 0u ? 1u : 0u) << k; }
            nul |= nk < 32 ? (1u << nk) : 0u;                              // the field ends with its width
        }
        if (nul) { first &= (nul & (0u - nul)) - 1u; ended = true; }      // only hits before the first NUL
        while (first) {
            const int pos = w0 + __builtin_ctz(first);
            first &= first - 1u;
            if (pos + len > width) break;
            int k = 1;
            while (k < len && s[pos + k] == val[k]) ++k;                  // a NUL inside stops the comparison: the needle has none
            if (k == len) { found = true; break; }
        }
    }
    return found;
}
// VarChar::firstIndex (reference include/varchar.h:91-97) on a field in LDS: position of the first occurrence of the needle
// in the text before the first NUL, or -1.  The same 32-position rounds as lds_str_contains.
template <class T>
__device__ __forceinline__ int64_t lds_first_index(const T* s, int width, const uint32_t* val, int len) {
    if (len == 0) return 0;
    const uint32_t v0 = val[0];
    bool ended = false;
    for (int w0 = 0; w0 < width && !ended; w0 += 32) {
        uint32_t first = 0, nul = 0;
        const int nk = min(32, width - w0);
        if (nk == 32) {
#pragma unroll
            for (int k = 0; k < 32; ++k) { const uint32_t c = s[w0 + k]; first |= (c == v0 ? 1u : 0u) << k; nul |= (c == 0u ? 1u : 0u) << k; }
        } else {
#pragma unroll 4
            for (int k = 0; k < nk; ++k) { const uint32_t c = s[w0 + k]; first |= (c == v0 ? 1u : 0u) << k; nul |= (c == 0u ? 1u : 0u) << k; }
            nul |= nk < 32 ? (1u << nk) : 0u;
        }
        if (nul) { first &= (nul & (0u - nul)) - 1u; ended = true; }
        while (first) {
            const int pos = w0 + __builtin_ctz(first);
            first &= first - 1u;
            if (pos + len > width) return -1;
            int k = 1;
            while (k < len && s[pos + k] == val[k]) ++k;                  // a NUL inside stops the comparison: the needle has none
            if (k == len) return pos;
        }
    }
    return -1;
}
template <class T>
__device__ __forceinline__ bool lds_str_pred(const T* s, int width, const uint32_t* val, int len, int mode) {
    if (mode == 2) {
        if (len == 0) return true;
        return lds_str_contains(s, width, val, len);
    }
    if (mode == 3) return str_prefix(s, width, val, len);          // at most `len` reads, early exit
    if (mode == 4) return str_suffix(s, width, val, len);
    return lds_str_equal(s, width, val, len) != (mode != 0);
}
// ---- the same scans on a byte twin in LDS ---------------------------------------------------------------
// A field of one-byte units starts at any byte offset of the staging region; it is read as aligned 32-bit words (one LDS read
// per four units), shifted into place, and the two position masks come from exact per-byte zero tests on whole words.
__device__ __forceinline__ uint32_t swar_zero_bytes(uint32_t x) { return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & 0x80808080u; }   // 0x80 where a byte is 0
__device__ __forceinline__ uint32_t swar_pack4(uint32_t z) { return (((z >> 7) * 0x00204081u) >> 21) & 0xFu; }                      // the four flags as bits 0..3
// units [0, nk) (nk <= 32) of the field at byte offset `off` of `region`: first = "equals v0", nul = "is NUL" (bits >= nk clear).
// Reads up to two words past the field (the neighbour's bytes, masked off; the region has that slack at its end).
__device__ __forceinline__ void lds8_scan(const uint32_t* region, int off, int nk, uint32_t v0x4, uint32_t& first, uint32_t& nul) {
    const uint32_t a8 = (uint32_t)(off & 3) * 8u;
    const uint32_t* wp = region + (off >> 2);
    const int nw = (nk + 3) >> 2;
    uint32_t w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = i <= nw ? wp[i] : 0u;
    first = 0; nul = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t x = __builtin_amdgcn_alignbit(w[i + 1], w[i], a8);
        nul |= swar_pack4(swar_zero_bytes(x)) << (4 * i);
        first |= swar_pack4(swar_zero_bytes(x ^ v0x4)) << (4 * i);
    }
    const uint32_t m = nk < 32 ? (1u << nk) - 1u : ~0u;
    first &= m; nul &= m;
}
// position of the first occurrence of the needle (len >= 1) in the text before the first NUL, or -1: VarChar::firstIndex /
// contains on the byte twin (a needle with a unit above 0xFF cannot occur in a column that has a byte twin)
__device__ __forceinline__ int lds8_find(const uint32_t* region, int off, int width, const uint32_t* val, int len) {
    uint32_t wide = 0;
    for (int k = 0; k < len; ++k) wide |= val[k];
    if (wide > 0xFFu) return -1;
    const uint8_t* s = reinterpret_cast<const uint8_t*>(region) + off;
    const uint32_t v0x4 = val[0] * 0x01010101u;
    bool ended = false;
    for (int w0 = 0; w0 < width && !ended; w0 += 32) {
        uint32_t first, nul;
        const int nk = min(32, width - w0);
        lds8_scan(region, off + w0, nk, v0x4, first, nul);
        nul |= nk < 32 ? (1u << nk) : 0u;                                  // the field ends with its width
        if (nul) { first &= (nul & (0u - nul)) - 1u; ended = true; }
        while (first) {
            const int pos = w0 + __builtin_ctz(first);
            first &= first - 1u;
            if (pos + len > width) return -1;
            int k = 1;
            while (k < len && s[pos + k] == val[k]) ++k;
            if (k == len) return pos;
        }
    }
    return -1;
}
// The string predicate over `rows` (64 or 32) consecutive rows (row0 + lane), staged through LDS:
// the wave copies the fixed-width fields with coalesced 16-byte loads (all in flight before the
// first LDS store), then lane i < rows scans field i out of LDS.  Per-lane character loops straight
// from global memory touch 64 different cache lines per instruction.  Returns the ballot of rows
// that pass.  row0 is a multiple of 32, so the block is 16-byte aligned; rows * swidth <= 4096.
// Only the rows below nrows are copied (the last vector may run up to 12 bytes into the column's slack).
// With a byte twin (f.sc8; the launch stages 64 rows, row0 a multiple of 64) the same fields are a quarter of the bytes.
__device__ __forceinline__ uint64_t str_stage_mask(const DevFilter& f, uint32_t* __restrict__ s_str, int64_t row0, int rows, int64_t nrows) {
    const int lane = lane_id();
    const int lim = (int)(nrows - row0 < (int64_t)rows ? nrows - row0 : (int64_t)rows);
    using V4 = uint32_t __attribute__((ext_vector_type(4)));
    const bool bytes = f.sc8 != nullptr;
    const int nvec = bytes ? (lim * f.swidth + 15) / 16 : (lim * f.swidth + 3) / 4;      // 16-byte vectors, <= 1024
    const V4* __restrict__ src = bytes ? reinterpret_cast<const V4*>(f.sc8 + row0 * f.swidth) : reinterpret_cast<const V4*>(f.sc + row0 * f.swidth);
    V4* dst = reinterpret_cast<V4*>(s_str);
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 < nvec; i0 += 8 * WAVE) {
        V4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * WAVE + lane; if (i < nvec) t[u] = __builtin_nontemporal_load(src + i); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * WAVE + lane; if (i < nvec) dst[i] = t[u]; }
    }
    __builtin_amdgcn_wave_barrier();
    bool ok = lane < lim;
    if (ok) {
        if (!bytes) ok = lds_str_pred(s_str + lane * f.swidth, f.swidth, f.sval, f.slen, f.sneg);
        else if (f.sneg == 2) ok = f.slen == 0 || lds8_find(s_str, lane * f.swidth, f.swidth, f.sval, f.slen) >= 0;
        else ok = lds_str_pred(reinterpret_cast<const uint8_t*>(s_str) + lane * f.swidth, f.swidth, f.sval, f.slen, f.sneg);
    }
    __builtin_amdgcn_wave_barrier();
    return __ballot(ok);
}

// 32-bit words of staging LDS per wave: slds rows of 4-byte units, or 64 rows of the byte twin
__host__ __device__ __forceinline__ size_t str_lds_words(const DevFilter& f) { return f.sc8 ? (size_t)16 * f.swidth : (size_t)f.slds * f.swidth; }
constexpr size_t STR_LDS_SLACK = 16;      // bytes after the last wave's region (lds8_scan reads whole words past a field)

__device__ __forceinline__ int64_t table_find(const DevTable& t, int64_t key, uint64_t cap_mask);

// contains(key): exact bitmap when the table has one, else open-addressing probe.
__device__ __forceinline__ bool table_contains(const DevTable& t, int64_t key, uint64_t cap_mask) {
    if (t.dense_arr || (t.bm && (t.bm_shift || t.lin_rb))) return table_find(t, key, cap_mask) >= 0;
    if (t.bm) {
        if (key < t.bm_lo || key > t.bm_hi) return false;
        uint64_t off = (uint64_t)(key - t.bm_lo);
        return (t.bm[off >> 5] >> (off & 31)) & 1u;
    }
    if (key == EMPTY_KEY) return slot_row(t, cap_mask + 1) != NO_ROW;
    uint64_t h = hash_key(key) & cap_mask;
    for (;;) {
        int64_t k = slot_key(t, h);
        if (k == key) return true;
        if (k == EMPTY_KEY) return false;
        h = (h + 1) & cap_mask;
    }
}

// position of `key` in the index (direct: its rank; hash: its slot), or -1; `word` = the bitmap
// word of the key when the caller has already fetched it (direct layout only)
__device__ __forceinline__ int64_t direct_rank(const DevTable& t, uint64_t off, uint32_t word) {
    const uint64_t w = off >> 5;
    return (int64_t)t.wprefix[w] + __popc(word & ((1u << (off & 31)) - 1u));
}
__device__ __forceinline__ int64_t table_find(const DevTable& t, int64_t key, uint64_t cap_mask) {
    if (t.dense_arr) {
        if (key < t.bm_lo || key > t.bm_hi) return -1;
        const uint32_t ref = t.dense_arr[key - t.bm_lo];
        return ref == NO_ROW ? -1 : (int64_t)ref;
    }
    // What follows the bitmap test is requested WITH the bitmap word, not after it (one memory round trip less in every
    // lookup of a latency-bound drain): the rank prefix of the word (direct layout), or the first hash slot (hash layout
    // behind a bitmap of the key's high part).  A miss wastes that request; the loops that call this mostly hit.
    const bool direct = t.bm && t.bm_shift == 0 && !t.bitmap_only;
    const bool hashed = !t.bitmap_only && !(t.bm && t.bm_shift == 0) && !t.grp_first;
    uint64_t h = 0;
    int64_t k_first = EMPTY_KEY;
    if (t.bm) {
        uint64_t off;
        if (!bm_locate(t, key, off)) return -1;
        uint32_t word, pre = 0;
        if (direct && t.wpair) { const unsigned long long pw = t.wpair[off >> 5]; pre = (uint32_t)pw; word = (uint32_t)(pw >> 32); }
        else { word = t.bm[off >> 5]; if (direct) pre = t.wprefix[off >> 5]; }
        if (t.grp_first) pre = t.grp_first[off];
        if (hashed) { h = hash_key(key) & cap_mask; k_first = slot_key(t, h); }
        if (!((word >> (off & 31)) & 1u)) return -1;
        if (t.bitmap_only) return 0;
        if (t.grp_first) {                                                   // grouped: walk the run of the key's high part
            int64_t p = (int64_t)pre;
            if (pre == NO_ROW) return -1;
            while (p < t.grp_cap) {
                // four entries of the run per round trip (most runs are that short); what lies behind a run's end is read and never looked at
                int64_t k[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) k[i] = t.grp_key[(p + i < t.grp_cap ? p + i : t.grp_cap - 1) * t.grp_kstride];
                bool jump = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (jump || p + i >= t.grp_cap) continue;
                    if (k[i] == key) return p + i;
                    if (k[i] == EMPTY_KEY) { p = ((p + i) / t.grp_seg_rows + 1) * t.grp_seg_rows; jump = true; continue; }
                    if (((uint64_t)k[i] >> 32) != ((uint64_t)key >> 32)) return -1;
                }
                if (!jump) p += 4;
            }
            return -1;
        }
        if (t.bm_shift == 0) {
            const uint32_t below = __popc(word & ((1u << (off & 31)) - 1u));
            if (t.wexc && (pre & ROW_INDEX_EXC)) {                           // row index, a word shared by two segments
                const WordExc e = t.wexc[pre & ~ROW_INDEX_EXC];
                return key < e.key0 ? (int64_t)e.pos_prev + below : (int64_t)e.pos0 + (below - e.n0);
            }
            return (int64_t)pre + below;
        }
    } else {
        h = hash_key(key) & cap_mask; k_first = slot_key(t, h);
    }
    if (key == EMPTY_KEY) return slot_row(t, cap_mask + 1) != NO_ROW ? (int64_t)(cap_mask + 1) : -1;
    int64_t k = k_first;
    for (;;) {
        if (k == key) return (int64_t)h;
        if (k == EMPTY_KEY) return -1;
        h = (h + 1) & cap_mask;
        k = slot_key(t, h);
    }
}
// stage index of the entry at an index position
__device__ __forceinline__ bool table_is_direct(const DevTable& t) { return t.dense_arr || (t.bm && t.bm_shift == 0) || t.grp_first; }     // no hash slots: cap_mask unused
// (direct layout without an owner array: the increasing-key form, rank = row — k_rank_increasing)
__device__ __forceinline__ uint32_t table_ref(const DevTable& t, int64_t pos) { return (t.dense_arr || t.grp_first) ? (uint32_t)pos : (t.bm && t.bm_shift == 0 ? (t.dense_ref ? t.dense_ref[pos] : (uint32_t)pos) : slot_row(t, (uint64_t)pos)); }

// ---- row filter on a pair of rows --------------------------------------------------------------
// Integer and double range predicates with their own columns, plus the optional string equality.
// Operand-slot ranges are applied by the caller on the loaded operands.
// The predicate columns of a row pair, fetched ahead of their use so that every load of a tile is
// in flight before the first one is consumed.
struct FilterRegs {
    Pair<int64_t> iv[SDQH_MAX_IPRED];
    Pair<double> fv[SDQH_MAX_FPRED];
};

template <class FC, bool TAIL, bool NW = false>
__device__ __forceinline__ void filter_load(const DevFilter& f, int64_t r, int64_t nrows, FilterRegs& fr) {
#pragma unroll
    for (int i = 0; i < SDQH_MAX_IPRED; ++i) if (i < cfg_ni<FC>(f.ni)) fr.iv[i] = loadc<TAIL, NW>(f.ic[i], r, nrows);
#pragma unroll
    for (int i = 0; i < SDQH_MAX_FPRED; ++i) if (i < cfg_nf<FC>(f.nf)) fr.fv[i] = loadc<TAIL, NW>(f.fc[i], r, nrows);
}

template <class FC, bool TAIL>
__device__ __forceinline__ void filter_eval(const DevFilter& f, const FilterRegs& fr, int64_t r, int64_t nrows, bool& p0, bool& p1) {
#pragma unroll
    for (int i = 0; i < SDQH_MAX_IPRED; ++i) {
        if (i < cfg_ni<FC>(f.ni)) {
            p0 &= (fr.iv[i].x >= f.ilo[i]) & (fr.iv[i].x <= f.ihi[i]);
            p1 &= (fr.iv[i].y >= f.ilo[i]) & (fr.iv[i].y <= f.ihi[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < SDQH_MAX_FPRED; ++i) {
        if (i < cfg_nf<FC>(f.nf)) {
            p0 &= (fr.fv[i].x >= f.flo[i]) & (fr.fv[i].x <= f.fhi[i]);
            p1 &= (fr.fv[i].y >= f.flo[i]) & (fr.fv[i].y <= f.fhi[i]);
        }
    }
    if (cfg_ns<FC>(f.ns)) {
        int64_t r0 = r < nrows ? r : nrows - 1, r1 = r + 1 < nrows ? r + 1 : nrows - 1;
        if (p0) p0 = str_pred(f.sc + r0 * f.swidth, f.swidth, f.sval, f.slen, f.sneg);
        if (p1) p1 = str_pred(f.sc + r1 * f.swidth, f.swidth, f.sval, f.slen, f.sneg);
    }
    if (cfg_nc<FC>(f.nc)) {
        int64_t r0 = r < nrows ? r : nrows - 1, r1 = r + 1 < nrows ? r + 1 : nrows - 1;
        if (p0) p0 = col_preds<FC>(f, r0);
        if (p1) p1 = col_preds<FC>(f, r1);
    }
}

template <class FC, bool TAIL>
__device__ __forceinline__ void filter_pair(const DevFilter& f, int64_t r, int64_t nrows, bool& p0, bool& p1) {
    FilterRegs fr;
    filter_load<FC, TAIL>(f, r, nrows, fr);
    filter_eval<FC, TAIL>(f, fr, r, nrows, p0, p1);
}

template <int NOPS>
__device__ __forceinline__ bool operand_ranges(const DevFilter& f, const double (&x)[4]) {
    bool p = true;
#pragma unroll
    for (int j = 0; j < NOPS; ++j)
        if ((f.omask >> j) & 1u) p &= (x[j] >= f.olo[j]) & (x[j] <= f.ohi[j]);
    return p;
}

// =================================================================================================
// K-A: scan -> filter -> sum of a value tuple.  Per-thread accumulators, wave shuffle + LDS block
// reduction, one partial per workgroup; k_sum_partials folds the partials in a fixed order, so the
// result is bit-reproducible run to run.
// partial layout: [grid][5] = NV doubles (padded to 4) + count (as int64 bits)
// =================================================================================================
template <int SHAPE, class FC, bool TAIL, bool NW = false>
__device__ __forceinline__ void scan_sum_tile(const DevFilter& f, const DevTuple& t, int64_t base, int64_t nrows,
                                              double (&acc)[4], int64_t& cnt) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV;
    Pair<double> xv[UNROLL][4];
    FilterRegs fr[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {                               // every load of the tile first
        const int64_t r = base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
#pragma unroll
        for (int j = 0; j < NOPS; ++j) xv[u][j] = loadc<TAIL, NW>(t.op[j], r, nrows);
        filter_load<FC, TAIL, NW>(f, r, nrows, fr[u]);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int64_t r = base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
        bool p0 = TAIL ? (r < nrows) : true, p1 = TAIL ? (r + 1 < nrows) : true;
        double x0[4] = {0, 0, 0, 0}, x1[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NOPS; ++j) { x0[j] = xv[u][j].x; x1[j] = xv[u][j].y; }
        filter_eval<FC, TAIL>(f, fr[u], r, nrows, p0, p1);
        p0 &= operand_ranges<NOPS>(f, x0);
        p1 &= operand_ranges<NOPS>(f, x1);
        double o0[4] = {0, 0, 0, 0}, o1[4] = {0, 0, 0, 0};
        tuple_eval<SHAPE>(x0, o0);
        tuple_eval<SHAPE>(x1, o1);
#pragma unroll
        for (int k = 0; k < NV; ++k) { acc[k] += p0 ? o0[k] : 0.0; acc[k] += p1 ? o1[k] : 0.0; }
        cnt += (int64_t)p0 + (int64_t)p1;
    }
}

template <int SHAPE, class FC, bool NW = false>          // NW: every streamed column (predicates, operands) is read through its narrow twin
__global__ __launch_bounds__(TPB) void k_scan_sum(DevFilter f, DevTuple t, int64_t nrows, double* __restrict__ partial) {
    constexpr int NV = TupleTraits<SHAPE>::NV;
    double acc[4] = {0, 0, 0, 0};
    int64_t cnt = 0;
    const int64_t full = nrows / TILE_ROWS;
    for (int64_t t0 = (int64_t)blockIdx.x * SDQH_TILE_CHUNK; t0 < full; t0 += (int64_t)gridDim.x * SDQH_TILE_CHUNK)
        for (int64_t tile = t0; tile < t0 + SDQH_TILE_CHUNK && tile < full; ++tile)
            scan_sum_tile<SHAPE, FC, false, NW>(f, t, tile * TILE_ROWS, nrows, acc, cnt);
    if (full * TILE_ROWS < nrows && blockIdx.x == (unsigned)(full % gridDim.x))
        scan_sum_tile<SHAPE, FC, true, NW>(f, t, full * TILE_ROWS, nrows, acc, cnt);

    __shared__ double s_acc[TPB / WAVE][4];
    __shared__ int64_t s_cnt[TPB / WAVE];
    const int w = threadIdx.x / WAVE;
#pragma unroll
    for (int k = 0; k < NV; ++k) { double v = wave_sum(acc[k]); if (lane_id() == 0) s_acc[w][k] = v; }
    { int64_t c = wave_sum_i64(cnt); if (lane_id() == 0) s_cnt[w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* out = partial + (size_t)blockIdx.x * 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double v = 0.0;
            if (k < NV) for (int i = 0; i < TPB / WAVE; ++i) v += s_acc[i][k];
            out[k] = v;
        }
        int64_t c = 0;
        for (int i = 0; i < TPB / WAVE; ++i) c += s_cnt[i];
        reinterpret_cast<int64_t*>(out)[4] = c;
    }
}

// out[0..3] doubles, out[4] count (int64 bits); one workgroup, fixed summation order
SDQH_KERNEL __launch_bounds__(TPB) void k_sum_partials(const double* __restrict__ partial, int nparts, double* __restrict__ out) {
    __shared__ double s[5][TPB];
    double a[4] = {0, 0, 0, 0};
    int64_t c = 0;
    for (int b = threadIdx.x; b < nparts; b += TPB) {
        const double* p = partial + (size_t)b * 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] += p[k];
        c += reinterpret_cast<const int64_t*>(p)[4];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k][threadIdx.x] = a[k];
    reinterpret_cast<int64_t*>(s[4])[threadIdx.x] = c;
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k][threadIdx.x] += s[k][threadIdx.x + off];
            reinterpret_cast<int64_t*>(s[4])[threadIdx.x] += reinterpret_cast<int64_t*>(s[4])[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = s[k][0];
        reinterpret_cast<int64_t*>(out)[4] = reinterpret_cast<int64_t*>(s[4])[0];
    }
}

// =================================================================================================
// K-C small: group-by over a small key domain.
//   Key: up to two 32-bit parts (UCS4 code unit of a string(1) column, or an int in [0, 2^32-2])
//   packed into one 64-bit word.  Each workgroup keeps its own key table in LDS (claimed by LDS CAS
//   in first-come order while it streams; doing that on a global table would serialise hundreds of
//   thousands of CAS on a handful of addresses).  k_groupby_reg keeps G x NV accumulators per lane
//   in registers and adds each row under a per-group predicate (no atomics, no divergence);
//   k_groupby_lds is the G <= 64 fall-back with LDS f64 atomics.  Only when a workgroup is done
//   does it map its few local slots to GLOBAL group slots (one agent-scope CAS per key, usually
//   just an atomic load) and write its partials at the global slot, so k_groupby_merge is a plain
//   fold of slot g over the workgroups in workgroup order: the sums are bit-reproducible run to
//   run (the order of the groups is not; the host sorts the handful of groups by key).
// partial layout (slot-major, coalesced for the merge): pacc[(g*nwg + wg)*4 + k], pcnt[g*nwg + wg]
// =================================================================================================
struct DevGroupKeys {
    const void* col[SDQH_MAX_GROUPKEYS];
    int32_t is_str[SDQH_MAX_GROUPKEYS];
    int32_t nkeys, _pad;
};

// one key slot of a row pair -> two 32-bit parts; KIND: 0 absent, 1 string(1), 2 int64, -1 run time
// NW (string(1) keys only): the column's narrow twin, one byte per code unit
template <int KIND, bool TAIL, bool NW = false>
__device__ __forceinline__ void load_key_part(const DevGroupKeys& gk, int j, int64_t r, int64_t nrows, uint32_t& a, uint32_t& b, bool& bad) {
    a = 0; b = 0;
    const bool present = KIND >= 0 ? KIND != 0 : j < gk.nkeys;
    if (!present) return;
    const bool is_str = KIND >= 0 ? KIND == 1 : gk.is_str[j] != 0;
    const int64_t r0 = (!TAIL || r < nrows) ? r : nrows - 1, r1 = (!TAIL || r + 1 < nrows) ? r + 1 : nrows - 1;
    if (is_str) {
        if constexpr (NW) {
            const uint8_t* c = static_cast<const uint8_t*>(gk.col[j]);
            if constexpr (!TAIL) { const uint16_t v = *reinterpret_cast<const uint16_t*>(c + r); a = v & 0xFFu; b = v >> 8; }
            else { a = c[r0]; b = c[r1]; }
        } else {
        const uint32_t* c = static_cast<const uint32_t*>(gk.col[j]);
        if constexpr (!TAIL) { uint2 v = *reinterpret_cast<const uint2*>(c + r); a = v.x; b = v.y; }
        else { a = c[r0]; b = c[r1]; }
        }
    } else {
        Pair<int64_t> v = load2<TAIL>(static_cast<const int64_t*>(gk.col[j]), r, nrows);
        bad |= (v.x < 0) | (v.x > 0xFFFFFFFEll) | (v.y < 0) | (v.y > 0xFFFFFFFEll);
        a = (uint32_t)v.x; b = (uint32_t)v.y;
    }
}

template <class KC, bool TAIL, bool NW = false>
__device__ __forceinline__ void load_group_keys(const DevGroupKeys& gk, int64_t r, int64_t nrows, uint64_t& k0, uint64_t& k1, bool& bad) {
    uint32_t a0, b0, a1, b1;
    load_key_part<KC::K0, TAIL, NW>(gk, 0, r, nrows, a0, b0, bad);
    load_key_part<KC::K1, TAIL, NW>(gk, 1, r, nrows, a1, b1, bad);
    k0 = (uint64_t)a0 | ((uint64_t)a1 << 32);
    k1 = (uint64_t)b0 | ((uint64_t)b1 << 32);
}

// claim-or-find in the workgroup's LDS key table; -1 when the table is full
template <int G>
__device__ __forceinline__ int lds_claim(unsigned long long* s_keys, uint64_t key) {
#pragma unroll 1
    for (int j = 0; j < G; ++j) {                            // cold path: keep it a real loop (unrolled copies thrash the I-cache)
        unsigned long long cur = s_keys[j];
        if (cur == key) return j;
        if (cur != EMPTY_GROUP) continue;
        unsigned long long old = atomicCAS(&s_keys[j], (unsigned long long)EMPTY_GROUP, (unsigned long long)key);
        if (old == EMPTY_GROUP || old == key) return j;
    }
    return -1;
}

// claim-or-find `key` in the global group table: atomic loads first (the key is almost always
// there already), one agent-scope CAS only on an empty entry.  -1 when all GMAX entries are taken.
__device__ __forceinline__ int global_group_slot(unsigned long long* gkeys, unsigned long long key) {
#pragma unroll 1
    for (int j = 0; j < SDQH_MAX_SMALL_GROUPS; ++j) {
        unsigned long long cur = __hip_atomic_load(&gkeys[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == EMPTY_GROUP) cur = atomicCAS(&gkeys[j], (unsigned long long)EMPTY_GROUP, key);
        if (cur == EMPTY_GROUP || cur == key) return j;
    }
    return -1;
}

// Workgroup epilogue shared by both group-by kernels: local slot -> global slot, then every one of
// the GMAX global slots of this workgroup's partial row is written (zeros where it saw no rows).
template <int G>
__device__ __forceinline__ void write_group_partials(const unsigned long long* s_keys, const double (*l_acc)[4], const int64_t* l_cnt,
                                                     unsigned long long* gkeys, double* __restrict__ pacc, int64_t* __restrict__ pcnt,
                                                     int* s_map, int* s_flags) {
    constexpr int GMAX = SDQH_MAX_SMALL_GROUPS;
    if (threadIdx.x < GMAX) s_map[threadIdx.x] = -1;
    __syncthreads();
    if (threadIdx.x < G && s_keys[threadIdx.x] != EMPTY_GROUP && l_cnt[threadIdx.x] > 0) {
        const int gs = global_group_slot(gkeys, s_keys[threadIdx.x]);
        if (gs < 0) atomicOr(&s_flags[0], 1); else s_map[gs] = threadIdx.x;
    }
    __syncthreads();
    if (threadIdx.x < GMAX) {
        const int l = s_map[threadIdx.x];
        const size_t e = (size_t)threadIdx.x * gridDim.x + blockIdx.x;
        pcnt[e] = l >= 0 ? l_cnt[l] : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) pacc[e * 4 + k] = l >= 0 ? l_acc[l][k] : 0.0;
    }
}

template <int SHAPE, int G, class FC, class KC, bool TAIL, bool NW = false>
__device__ __forceinline__ void groupby_reg_tile(const DevFilter& f, const DevTuple& t, const DevGroupKeys& gk, int64_t base, int64_t nrows,
                                                 unsigned long long* s_keys, int* s_flags,
                                                 double (&acc)[G][4], int32_t (&cnt)[G]) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV;
    Pair<double> xv[UNROLL][4];
    FilterRegs fr[UNROLL];
    uint64_t key[UNROLL][2];
    bool bad = false;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {                               // every load of the tile first
        const int64_t r = base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
#pragma unroll
        for (int j = 0; j < NOPS; ++j) xv[u][j] = loadc<TAIL, NW>(t.op[j], r, nrows);
        load_group_keys<KC, TAIL, NW && KC::K0 == 1 && KC::K1 == 1>(gk, r, nrows, key[u][0], key[u][1], bad);      // (the NW instances exist for two string(1) keys)
        filter_load<FC, TAIL, NW>(f, r, nrows, fr[u]);
    }
    uint64_t rk[G];
#pragma unroll
    for (int g = 0; g < G; ++g) rk[g] = s_keys[g];          // register copy of the key table (LDS broadcast reads)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        const int64_t r = base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
        bool p[2] = {TAIL ? (r < nrows) : true, TAIL ? (r + 1 < nrows) : true};
        filter_eval<FC, TAIL>(f, fr[u], r, nrows, p[0], p[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            double x[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < NOPS; ++j) x[j] = q == 0 ? xv[u][j].x : xv[u][j].y;
            p[q] &= operand_ranges<NOPS>(f, x);
            int slot = -1;
#pragma unroll
            for (int g = 0; g < G; ++g) slot = (rk[g] == key[u][q]) ? g : slot;
            if (p[q] && slot < 0) {                              // rare: first sight of a key in this workgroup
                slot = lds_claim<G>(s_keys, key[u][q]);
                if (slot < 0) atomicOr(&s_flags[0], 1);           // more than G groups
            }
            if (p[q] && bad) atomicOr(&s_flags[0], 2);            // int key outside [0, 2^32-2]
            double o[4] = {0, 0, 0, 0};
            tuple_eval<SHAPE>(x, o);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool m = p[q] && (slot == g);
#pragma unroll
                for (int k = 0; k < NV; ++k) acc[g][k] += m ? o[k] : 0.0;
                cnt[g] += m ? 1 : 0;
            }
        }
    }
}

template <int SHAPE, int G, class FC, class KC, bool NW = false>      // NW: predicates and operands through their narrow twins (the group keys stay as they are)
__global__ __launch_bounds__(TPB) void k_groupby_reg(DevFilter f, DevTuple t, DevGroupKeys gk, int64_t nrows,
                                                     unsigned long long* __restrict__ gkeys, double* __restrict__ pacc,
                                                     int64_t* __restrict__ pcnt, int* __restrict__ flags) {
    constexpr int NV = TupleTraits<SHAPE>::NV;
    __shared__ unsigned long long s_keys[G];
    __shared__ int s_flags[1];
    __shared__ double s_acc[TPB / WAVE][G][4];
    __shared__ int64_t s_cnt[TPB / WAVE][G];
    if (threadIdx.x < G) s_keys[threadIdx.x] = EMPTY_GROUP;
    if (threadIdx.x == 0) s_flags[0] = 0;
    __syncthreads();

    double acc[G][4];
    int32_t cnt[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { cnt[g] = 0; for (int k = 0; k < 4; ++k) acc[g][k] = 0.0; }

    const int64_t full = nrows / TILE_ROWS;
    for (int64_t t0 = (int64_t)blockIdx.x * SDQH_TILE_CHUNK; t0 < full; t0 += (int64_t)gridDim.x * SDQH_TILE_CHUNK)
        for (int64_t tile = t0; tile < t0 + SDQH_TILE_CHUNK && tile < full; ++tile)
            groupby_reg_tile<SHAPE, G, FC, KC, false, NW>(f, t, gk, tile * TILE_ROWS, nrows, s_keys, s_flags, acc, cnt);
    if (full * TILE_ROWS < nrows && blockIdx.x == (unsigned)(full % gridDim.x))
        groupby_reg_tile<SHAPE, G, FC, KC, true, NW>(f, t, gk, full * TILE_ROWS, nrows, s_keys, s_flags, acc, cnt);

    const int w = threadIdx.x / WAVE;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int k = 0; k < NV; ++k) { double v = wave_sum(acc[g][k]); if (lane_id() == 0) s_acc[w][g][k] = v; }
        int64_t c = wave_sum_i64((int64_t)cnt[g]);
        if (lane_id() == 0) s_cnt[w][g] = c;
    }
    __syncthreads();
    __shared__ double s_tot[G][4];
    __shared__ int64_t s_totc[G];
    __shared__ int s_map[SDQH_MAX_SMALL_GROUPS];
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        int64_t c = 0;
        for (int i = 0; i < TPB / WAVE; ++i) c += s_cnt[i][g];
        s_totc[g] = c;
        for (int k = 0; k < 4; ++k) {
            double v = 0.0;
            if (k < NV) for (int i = 0; i < TPB / WAVE; ++i) v += s_acc[i][g][k];
            s_tot[g][k] = v;
        }
    }
    __syncthreads();
    write_group_partials<G>(s_keys, s_tot, s_totc, gkeys, pacc, pcnt, s_map, s_flags);
    __syncthreads();
    if (threadIdx.x == 0 && s_flags[0]) atomicOr(flags, s_flags[0]);
}

// Fall-back for up to 64 groups: accumulators in LDS, f64 LDS atomics (ds_add_f64).
template <int SHAPE, bool TAIL>
__device__ __forceinline__ void groupby_lds_tile(const DevFilter& f, const DevTuple& t, const DevGroupKeys& gk, int64_t base, int64_t nrows,
                                                 unsigned long long* s_keys, double (*s_acc)[4],
                                                 unsigned long long* s_cnt, int* s_flags) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV, G = SDQH_MAX_SMALL_GROUPS;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        int64_t r = base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
        bool p[2] = {TAIL ? (r < nrows) : true, TAIL ? (r + 1 < nrows) : true};
        double x[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int j = 0; j < NOPS; ++j) { Pair<double> v = load2<TAIL>(t.op[j], r, nrows); x[0][j] = v.x; x[1][j] = v.y; }
        uint64_t key[2]; bool bad = false;
        load_group_keys<KGeneric, TAIL>(gk, r, nrows, key[0], key[1], bad);
        filter_pair<FGeneric, TAIL>(f, r, nrows, p[0], p[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            p[q] &= operand_ranges<NOPS>(f, x[q]);
            if (!p[q]) continue;
            if (bad) atomicOr(&s_flags[0], 2);
            const int slot = lds_claim<G>(s_keys, key[q]);
            if (slot < 0) { atomicOr(&s_flags[0], 1); continue; }
            double o[4] = {0, 0, 0, 0};
            tuple_eval<SHAPE>(x[q], o);
#pragma unroll
            for (int k = 0; k < NV; ++k) atomicAdd(&s_acc[slot][k], o[k]);
            atomicAdd(&s_cnt[slot], 1ull);
        }
    }
}

template <int SHAPE>
__global__ __launch_bounds__(TPB) void k_groupby_lds(DevFilter f, DevTuple t, DevGroupKeys gk, int64_t nrows,
                                                     unsigned long long* __restrict__ gkeys, double* __restrict__ pacc,
                                                     int64_t* __restrict__ pcnt, int* __restrict__ flags) {
    constexpr int G = SDQH_MAX_SMALL_GROUPS;
    __shared__ unsigned long long s_keys[G];
    __shared__ double s_acc[G][4];
    __shared__ unsigned long long s_cnt[G];
    __shared__ int s_flags[1];
    if (threadIdx.x < G) { s_keys[threadIdx.x] = EMPTY_GROUP; s_cnt[threadIdx.x] = 0; for (int k = 0; k < 4; ++k) s_acc[threadIdx.x][k] = 0.0; }
    if (threadIdx.x == 0) s_flags[0] = 0;
    __syncthreads();
    const int64_t full = nrows / TILE_ROWS;
    for (int64_t tile = blockIdx.x; tile < full; tile += gridDim.x)
        groupby_lds_tile<SHAPE, false>(f, t, gk, tile * TILE_ROWS, nrows, s_keys, s_acc, s_cnt, s_flags);
    if (full * TILE_ROWS < nrows && blockIdx.x == (unsigned)(full % gridDim.x))
        groupby_lds_tile<SHAPE, true>(f, t, gk, full * TILE_ROWS, nrows, s_keys, s_acc, s_cnt, s_flags);
    __syncthreads();
    __shared__ int64_t s_totc[G];
    __shared__ int s_map[SDQH_MAX_SMALL_GROUPS];
    if (threadIdx.x < G) s_totc[threadIdx.x] = (int64_t)s_cnt[threadIdx.x];
    __syncthreads();
    write_group_partials<G>(s_keys, s_acc, s_totc, gkeys, pacc, pcnt, s_map, s_flags);
    __syncthreads();
    if (threadIdx.x == 0 && s_flags[0]) atomicOr(flags, s_flags[0]);
}

// One workgroup per global group slot: fold that slot's partials over the workgroups in workgroup
// order (thread-strided, then an LDS tree).  out: acc[g*4+k], cnt[g]; the keys are gkeys[g].
// out_keys / out_tail (optional): the group keys and the 8 bytes after the block (group count / flags) are copied along, so that
// out_* can be the caller's pinned host block itself — the result is on the host when the stream is idle, without a copy-engine
// launch after the kernel (one API call and one DMA round trip less per group-by call)
// reset: the kernel is the last reader of the group-key slots and of the tail; it leaves them as the next call's fill would
// (EMPTY_GROUP / zero), so that call launches no fill (sdqh_hip.hip: rd_clean_*)
SDQH_KERNEL __launch_bounds__(TPB) void k_groupby_merge(unsigned long long* __restrict__ gkeys, const double* __restrict__ pacc,
                                                       const int64_t* __restrict__ pcnt, int nparts,
                                                       double* __restrict__ out_acc, int64_t* __restrict__ out_cnt,
                                                       unsigned long long* __restrict__ out_keys, int* __restrict__ d_tail, int* __restrict__ out_tail, int reset) {
    __shared__ double s_red[5][TPB];
    const int g = blockIdx.x;
    const unsigned long long gkey = gkeys[g];
    if (threadIdx.x == 0 && out_keys) out_keys[g] = gkey;
    if (g == 0 && threadIdx.x == 0 && out_tail) { out_tail[0] = d_tail[0]; out_tail[1] = d_tail[1]; if (reset) { d_tail[0] = 0; d_tail[1] = 0; } }
    if (gkey == EMPTY_GROUP) { if (threadIdx.x == 0) out_cnt[g] = 0; return; }
    double a[4] = {0, 0, 0, 0};
    int64_t c = 0;
    constexpr int BB = 4;
    for (int b0 = threadIdx.x; b0 < nparts; b0 += TPB * BB) {
        double4 v[BB]; int64_t n[BB];
#pragma unroll
        for (int i = 0; i < BB; ++i) {
            const int b = b0 + i * TPB;
            const size_t e = (size_t)g * nparts + (b < nparts ? b : 0);
            v[i] = *reinterpret_cast<const double4*>(pacc + e * 4);
            n[i] = b < nparts ? pcnt[e] : 0;
        }
#pragma unroll
        for (int i = 0; i < BB; ++i) if (n[i] > 0) { a[0] += v[i].x; a[1] += v[i].y; a[2] += v[i].z; a[3] += v[i].w; c += n[i]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) s_red[k][threadIdx.x] = a[k];
    reinterpret_cast<int64_t*>(s_red[4])[threadIdx.x] = c;
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s_red[k][threadIdx.x] += s_red[k][threadIdx.x + off];
            reinterpret_cast<int64_t*>(s_red[4])[threadIdx.x] += reinterpret_cast<int64_t*>(s_red[4])[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) out_acc[g * 4 + k] = s_red[k][0];
        out_cnt[g] = reinterpret_cast<int64_t*>(s_red[4])[0];
        if (reset) gkeys[g] = EMPTY_GROUP;                 // every thread of the block read its copy before the barriers above
    }
}

// =================================================================================================
// K-B: unique hash build.
//   k_stage   each wave64 owns a contiguous row segment; filter + semi-join probes; survivors are
//             compacted in row order with ballot + popcount prefix into the segment's own slice of
//             the stage arrays (no global atomics, no workgroup barrier).  Four 128-row batches
//             are in flight per wave so the dependent chain date -> probe key -> bitmap word is
//             overlapped four deep.  The entry state (hits, accumulators) is zeroed and the exact
//             key bitmap is set here, under the streaming reads.
//   k_clear   sizes the table on the device (capacity = pow2 >= 2 * staged rows) and resets keys
//   k_insert  wave per segment: one CAS per row claims the key, the claimer stores its stage
//             index.  Scattered atomics run at a fixed chip-wide rate, so the common case (unique
//             build keys) pays exactly one.  A row that meets its own key already present only
//             raises has_dups; k_insert_fixup (a no-op otherwise) then lowers every entry to its
//             lowest build row with atomicMin — first insert wins, as in the reference.
// =================================================================================================
struct DevProbes {
    DevTable table[SDQH_MAX_PROBE];
    const int64_t* key[SDQH_MAX_PROBE];
    int32_t n, _pad;
};

constexpr int MAX_STAGE_COLS = SDQH_MAX_COMPACT_COLS - 1;     // payload columns a stage can carry
constexpr uint32_t DEAD_ROW = 0xFFFFFFFFu;
constexpr int STAGE_BATCH = 4;                                // 128-row batches in flight per wave

struct DevStage {
    int64_t* key;                         // [nrows]
    int64_t* pay[MAX_STAGE_COLS];         // [nrows] each
    const int64_t* src_key;
    const int64_t* src_pay[MAX_STAGE_COLS];
    int32_t npay, nseg;
    uint32_t* seg_count;                  // [nseg]
    int64_t seg_rows;                     // rows per wave segment (multiple of 128 * STAGE_BATCH)
    uint32_t* shits;                      // [nrows] entry hit counters, zeroed while staging
    double* sacc;                         // [nrows*acc_stride] entry accumulators, zeroed while staging (or null)
    int32_t acc_stride, _pad3;
    uint32_t* bm;                         // key bitmap to fill, or null
    int64_t bm_lo, bm_hi;
    TableHeader* hdr;
    int32_t bm_shift, _pad2;              // 32: the bitmap covers the high part of a composite key
    int64_t lin_rb, lin_b0;               // linearised composite key (see DevTable)
    uint32_t* wrow;                       // ROW INDEX (see DevTable): per bitmap word the stage row of its first key, written while staging; or null
    SegFirst* seg_first;                  // [nseg] ... and every segment's first entry, for k_wrow_fixup
    uint32_t* grp_first;                  // GROUPED layout (see DevTable): first stage row per high key part, NO_ROW-filled before the build; or null
    int64_t* grp_kp;                      // ... and its (key, payload 0) pairs by stage row, or null
};

// Row index, the writer's side: the entries a wave is about to store (one per lane, `active` lanes, increasing keys, consecutive stage
// rows) — an entry whose key opens a bitmap word records its stage row there; the segment's very first entry goes to seg_first instead.
// `carry` = the word of the wave's previous entry (ROW_INDEX_NONE at the start of a segment); wave-converged.
constexpr uint32_t ROW_INDEX_NONE = 0xFFFFFFFFu;
__device__ __forceinline__ void row_index_note(const DevStage& st, int seg, bool active, int64_t key, int64_t pos, uint64_t active_mask, uint32_t& carry) {
    if (!st.wrow) return;
    active = active && key >= st.bm_lo && key <= st.bm_hi;                  // (a key outside the bounds fails the call: it is in no word)
    active_mask = __ballot(active);
    if (!active_mask) return;
    const int lane = (int)(threadIdx.x & (WAVE - 1));
    const uint32_t word = active ? (uint32_t)((uint64_t)(key - st.bm_lo) >> 5) : ROW_INDEX_NONE;
    const uint64_t lower = active_mask & ((1ull << lane) - 1ull);
    const int prev_lane = lower ? 63 - __builtin_clzll(lower) : lane;
    const uint32_t prev_word_lane = (uint32_t)__shfl((int)word, prev_lane, WAVE);
    const uint32_t prev = lower ? prev_word_lane : carry;
    if (active && word != prev) {
        if (prev == ROW_INDEX_NONE) { SegFirst f; f.key = key; f.pos = (uint32_t)pos; f._pad = 0; st.seg_first[seg] = f; }
        else st.wrow[word] = (uint32_t)pos;
    }
    carry = (uint32_t)__shfl((int)word, 63 - __builtin_clzll(active_mask), WAVE);
}

__device__ __forceinline__ void zero_acc(const DevStage& st, int64_t pos) {
    if (st.acc_stride == 4) { double4 z = {0, 0, 0, 0}; *reinterpret_cast<double4*>(st.sacc + pos * 4) = z; }
    else for (int k = 0; k < st.acc_stride; ++k) st.sacc[pos * st.acc_stride + k] = 0.0;
}

template <class FC>
__device__ __forceinline__ bool row_passes(const DevFilter& f, const DevProbes& pr, int64_t r, const uint64_t* cap_masks) {
    bool p = true;
#pragma unroll
    for (int i = 0; i < SDQH_MAX_IPRED; ++i) if (i < cfg_ni<FC>(f.ni) && p) { int64_t v = f.ic[i][r]; p = (v >= f.ilo[i]) & (v <= f.ihi[i]); }
#pragma unroll
    for (int i = 0; i < SDQH_MAX_FPRED; ++i) if (i < cfg_nf<FC>(f.nf) && p) { double v = f.fc[i][r]; p = (v >= f.flo[i]) & (v <= f.fhi[i]); }
    if (cfg_ns<FC>(f.ns) && p) p = str_pred(f.sc + r * f.swidth, f.swidth, f.sval, f.slen, f.sneg);
    if (cfg_nc<FC>(f.nc) && p) p = col_preds<FC>(f, r);
#pragma unroll
    for (int i = 0; i < SDQH_MAX_PROBE; ++i) if (i < cfg_np<FC>(pr.n) && p) p = table_contains(pr.table[i], pr.key[i][r], cap_masks[i]);
    return p;
}

// Filter + semi-join probes for NB pairs of rows at once.  Every stage of the dependent chain is
// issued for all NB pairs before the next stage consumes it: first predicate column with 16-byte
// loads for every row, the remaining predicates and the probe keys only by lanes still alive.
// The first stage of pass_pairs' loads (first probe's keys, first integer predicate column), requested a step ahead by a
// pipelined caller (k_stage<..., PIPE>)
template <int NB> struct PairsPre { Pair<int64_t> ek[NB], d[NB]; };
template <int NB, class FC>
__device__ __forceinline__ void pairs_preload(const DevFilter& f, const DevProbes& pr, int64_t row0, int64_t step_rows, int lane, int64_t nrows, PairsPre<NB>& pre) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int64_t r = row0 + (int64_t)j * step_rows + (int64_t)lane * ROWS_PER_LOAD;
        pre.ek[j] = load2<false>(pr.key[0], r, nrows);
        pre.d[j] = load2<false>(f.ic[0], r, nrows);
    }
}
// NW: the two first-stage columns (first probe's key when EAGER, first integer predicate) are read through their narrow twins nkey / npred0
template <int NB, class FC, bool EAGER = true, bool SKIP_FIRST = false, bool PRE = false, bool NW = false>      // SKIP_FIRST: the caller has applied the first integer predicate itself
__device__ __forceinline__ void pass_pairs(const DevFilter& f, const DevProbes& pr, const int64_t (&r)[NB], int64_t nrows,
                                           const uint64_t* cap_masks, bool (&p)[NB][2], uint32_t* s_str = nullptr, const PairsPre<NB>* pre = nullptr,
                                           const int32_t* nkey = nullptr, const int32_t* npred0 = nullptr) {
    // EAGER: the first probe's key column is streamed with 16-byte loads alongside the first
    // predicate instead of being fetched afterwards by the surviving lanes only.  When a good part
    // of the rows survive, every cache line of the key column is touched anyway, and one stage of
    // the dependent chain (predicate -> key -> bitmap word) disappears.
    Pair<int64_t> ek[NB];
    const bool eager = EAGER && cfg_np<FC>(pr.n) > 0;
    if (eager) {
#pragma unroll
        for (int j = 0; j < NB; ++j) { if constexpr (PRE) ek[j] = pre->ek[j]; else ek[j] = loadc<false, NW>(NW ? reinterpret_cast<const int64_t*>(nkey) : pr.key[0], r[j], nrows); }
    }
    if (!SKIP_FIRST && cfg_ni<FC>(f.ni) > 0) {
        Pair<int64_t> d[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) { if constexpr (PRE) d[j] = pre->d[j]; else d[j] = loadc<false, NW>(NW ? reinterpret_cast<const int64_t*>(npred0) : f.ic[0], r[j], nrows); }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            p[j][0] &= (d[j].x >= f.ilo[0]) & (d[j].x <= f.ihi[0]);
            p[j][1] &= (d[j].y >= f.ilo[0]) & (d[j].y <= f.ihi[0]);
        }
    }
#pragma unroll
    for (int i = 1; i < SDQH_MAX_IPRED; ++i) if (i < cfg_ni<FC>(f.ni)) {
        int64_t v[NB][2];
#pragma unroll
        for (int j = 0; j < NB; ++j) { v[j][0] = p[j][0] ? f.ic[i][r[j]] : f.ilo[i]; v[j][1] = p[j][1] ? f.ic[i][r[j] + 1] : f.ilo[i]; }
#pragma unroll
        for (int j = 0; j < NB; ++j) { p[j][0] &= (v[j][0] >= f.ilo[i]) & (v[j][0] <= f.ihi[i]); p[j][1] &= (v[j][1] >= f.ilo[i]) & (v[j][1] <= f.ihi[i]); }
    }
#pragma unroll
    for (int i = 0; i < SDQH_MAX_FPRED; ++i) if (i < cfg_nf<FC>(f.nf)) {
        double v[NB][2];
#pragma unroll
        for (int j = 0; j < NB; ++j) { v[j][0] = p[j][0] ? f.fc[i][r[j]] : f.flo[i]; v[j][1] = p[j][1] ? f.fc[i][r[j] + 1] : f.flo[i]; }
#pragma unroll
        for (int j = 0; j < NB; ++j) { p[j][0] &= (v[j][0] >= f.flo[i]) & (v[j][0] <= f.fhi[i]); p[j][1] &= (v[j][1] >= f.flo[i]) & (v[j][1] <= f.fhi[i]); }
    }
    if (cfg_ns<FC>(f.ns)) {
        if (s_str) {
            // lane L holds rows 2L, 2L+1 of each 128-row batch: rows 0..63 belong to lanes 0..31, 64..127 to lanes 32..63
            const int lane = lane_id();
            const int half = lane >> 5, sh = (2 * lane) & 63;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int64_t base = r[j] - 2 * lane;
                const uint64_t live = __ballot(p[j][0] | p[j][1]);
                uint64_t m[2] = {0, 0};
                if (f.slds == 64) {
                    if (live & 0x00000000FFFFFFFFull) m[0] = str_stage_mask(f, s_str, base, 64, nrows);
                    if (live & 0xFFFFFFFF00000000ull) m[1] = str_stage_mask(f, s_str, base + 64, 64, nrows);
                } else {                                                  // wide fields: 32 rows per staging keeps the LDS footprint of a wave small
                    if (live & 0x000000000000FFFFull) m[0] = str_stage_mask(f, s_str, base, 32, nrows);
                    if (live & 0x00000000FFFF0000ull) m[0] |= str_stage_mask(f, s_str, base + 32, 32, nrows) << 32;
                    if (live & 0x0000FFFF00000000ull) m[1] = str_stage_mask(f, s_str, base + 64, 32, nrows);
                    if (live & 0xFFFF000000000000ull) m[1] |= str_stage_mask(f, s_str, base + 96, 32, nrows) << 32;
                }
                const uint64_t mine = half ? m[1] : m[0];
                p[j][0] = p[j][0] && ((mine >> sh) & 1ull);
                p[j][1] = p[j][1] && ((mine >> (sh + 1)) & 1ull);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (p[j][0]) p[j][0] = str_pred(f.sc + r[j] * f.swidth, f.swidth, f.sval, f.slen, f.sneg);
                if (p[j][1]) p[j][1] = str_pred(f.sc + (r[j] + 1) * f.swidth, f.swidth, f.sval, f.slen, f.sneg);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < SDQH_MAX_CPRED; ++i) if (i < cfg_nc<FC>(f.nc)) {        // both sides streamed with 16-byte loads, all in flight first
        Pair<int64_t> a[NB], b[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (NW && FC::NI == 0 && FC::NP == 0 && FC::NC == 1) {      // the one-comparison instance (Q4): both integer columns through their twins
                a[j] = loadc<false, true>(reinterpret_cast<const int64_t*>(nkey), r[j], nrows); b[j] = loadc<false, true>(reinterpret_cast<const int64_t*>(npred0), r[j], nrows);
            } else { a[j] = load2<false>(f.ca[i], r[j], nrows); b[j] = load2<false>(f.cb[i], r[j], nrows); }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) { p[j][0] = p[j][0] && col_cmp(f, i, a[j].x, b[j].x); p[j][1] = p[j][1] && col_cmp(f, i, a[j].y, b[j].y); }
    }
#pragma unroll
    for (int i = 0; i < SDQH_MAX_PROBE; ++i) if (i < cfg_np<FC>(pr.n)) {
        int64_t k[NB][2];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (i == 0 && eager) { k[j][0] = ek[j].x; k[j][1] = ek[j].y; }
            else { k[j][0] = p[j][0] ? pr.key[i][r[j]] : 0; k[j][1] = p[j][1] ? pr.key[i][r[j] + 1] : 0; }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (p[j][0]) p[j][0] = table_contains(pr.table[i], k[j][0], cap_masks[i]);
            if (p[j][1]) p[j][1] = table_contains(pr.table[i], k[j][1], cap_masks[i]);
        }
    }
}

// payload column count known at compile time (NPAY >= 0) or read from the arguments (-1)
template <int NPAY> __device__ __forceinline__ int cfg_npay(const int32_t n) { if constexpr (NPAY >= 0) return NPAY; else return n; }

// (Round 3 tried setting the bitmap bits of a drain's 64 rows by the wave together — equal words of adjacent lanes ORed first by a
// segmented scan, one atomic per run, and plain stores for the runs strictly inside a wave's range of a strictly increasing key
// column.  Merged atomics alone: Q3's orders build 0.128 -> 0.144 ms, the scan costs more than the atomics it saves.  With the
// plain stores: wrong results at SF=1 — plain stores and memory-side atomics do not mix on one cache line.  One atomic per row stays.)
template <int NPAY>
__device__ __forceinline__ void stage_store(const DevStage& st, int64_t pos, int64_t key, const int64_t (&pay)[MAX_STAGE_COLS], bool opens_run = true) {
    st.key[pos] = key;
#pragma unroll
    for (int q = 0; q < MAX_STAGE_COLS; ++q) if (q < cfg_npay<NPAY>(st.npay)) st.pay[q][pos] = pay[q];
    if (st.shits) st.shits[pos] = 0;
    if (st.sacc) zero_acc(st, pos);
    if (st.bm) {
        uint64_t off;
        if (bm_locate(st, key, off)) {
#if defined(XS_EXP) && (XS_EXP & 1)
            if (key == -12345)                                // (timing experiment of the specialised builds: no bitmap atomics)
#endif
            atomicOr(&st.bm[off >> 5], 1u << (off & 31));     // fire and forget; duplicates show up as distinct < staged (k_rank_words)
            // grouped layout: off = the high part's offset (bm_shift 32, no rectangle).  opens_run false: the caller knows the entry stored just
            // before this one (the wave's previous kept lane) has the same high part — the run's first stage row is not this one
            if (st.grp_first && opens_run) atomicMin(&st.grp_first[off], (uint32_t)pos);
            if (st.grp_kp) { using V2 = long long __attribute__((ext_vector_type(2))); const V2 kp = {(long long)key, (long long)pay[0]}; reinterpret_cast<V2*>(st.grp_kp)[pos] = kp; }
        }
    }
}
// grouped layout: a segment that did not fill up ends in EMPTY_KEY (DevTable: the walk of a run goes on at the next segment's base)
__device__ __forceinline__ void stage_end_segment(const DevStage& st, int seg, int64_t begin, int64_t count) {
    if (st.grp_first && (count < st.seg_rows || seg == st.nseg - 1)) { st.key[begin + count] = EMPTY_KEY; if (st.grp_kp) st.grp_kp[2 * (begin + count)] = EMPTY_KEY; }      // (a last segment that is all entries: the mark sits in the array's slack)
}

// PIPE (needs EAGER, one integer predicate and one probe at least): the next step's first-stage loads are requested at the
// top of this step, so a step waits one memory round trip less (the build is bound by its chain of dependent round trips,
// not by bytes: see "Staging" in DESIGN.md §3)
template <class FC, int NPAY = -1, int SB = STAGE_BATCH, bool EAGER = true, bool EAGER_PAY = false, bool PIPE = false, bool NW = false>
__global__ __launch_bounds__(TPB) void k_stage(DevFilter f, DevProbes pr, DevStage st, int64_t nrows, const int32_t* __restrict__ nkey, const int32_t* __restrict__ npred0, DevFill pre) {
    if (pre.n) { fill_in_block(pre); __syncthreads(); }                  // a tiny table (grid of one workgroup) does its own fill
    extern __shared__ __align__(16) uint32_t s_dyn[];                   // f.slds * swidth words per wave (string predicate staging)
    __shared__ uint16_t s_queue[TPB / WAVE][WAVE * ROWS_PER_LOAD * SB];   // survivors of a step (row offsets), in row order
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    uint16_t* q_off = s_queue[threadIdx.x / WAVE];
    uint32_t* s_str = (cfg_ns<FC>(f.ns) && f.slds) ? s_dyn + (size_t)(threadIdx.x / WAVE) * str_lds_words(f) : nullptr;
    uint64_t cap_masks[SDQH_MAX_PROBE] = {0, 0};
#pragma unroll
    for (int i = 0; i < SDQH_MAX_PROBE; ++i) if (i < cfg_np<FC>(pr.n) && pr.table[i].hdr && !table_is_direct(pr.table[i]) && !pr.table[i].bitmap_only) cap_masks[i] = pr.table[i].hdr->cap_mask;
    const int64_t begin = (int64_t)seg * st.seg_rows;
    int64_t end = begin + st.seg_rows; if (end > nrows) end = nrows;
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    constexpr int64_t BATCH_ROWS = WAVE * ROWS_PER_LOAD;                 // 128
    int64_t out = begin;                                   // wave-uniform write cursor into the segment's stage slice
    PairsPre<SB> nxt;
    if constexpr (PIPE) { if (begin + BATCH_ROWS * SB <= end) pairs_preload<SB, FC>(f, pr, begin, BATCH_ROWS, lane, nrows, nxt); }
    for (int64_t b = begin; b < end; b += BATCH_ROWS * SB) {
        int64_t r[SB];
        bool p[SB][2];
#pragma unroll
        for (int j = 0; j < SB; ++j) r[j] = b + j * BATCH_ROWS + (int64_t)lane * ROWS_PER_LOAD;
        int64_t kx[SB][2], px[SB][2][MAX_STAGE_COLS];
        const bool full_step = b + BATCH_ROWS * SB <= end;
        if (full_step) {
            if constexpr (EAGER_PAY) {
                // key + payload streamed with 16-byte loads for every row, survivor or not: when a
                // tenth or more of the rows survive, most cache lines of these columns are touched
                // anyway, and sparse 8-byte gathers after the filter cost more than they save
#pragma unroll
                for (int j = 0; j < SB; ++j) {
                    Pair<int64_t> kk = load2<false>(st.src_key, r[j], nrows);
                    kx[j][0] = kk.x; kx[j][1] = kk.y;
#pragma unroll
                    for (int c = 0; c < MAX_STAGE_COLS; ++c) if (c < cfg_npay<NPAY>(st.npay)) {
                        Pair<int64_t> pp = load2<false>(st.src_pay[c], r[j], nrows);
                        px[j][0][c] = pp.x; px[j][1][c] = pp.y;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < SB; ++j) p[j][0] = p[j][1] = true;
            if constexpr (PIPE) {
                const PairsPre<SB> cur = nxt;
                // the next full step's rows; after the last full step this step's again (an unconditional request: see k_lookup_agg)
                const int64_t nb = b + 2 * BATCH_ROWS * SB <= end ? b + BATCH_ROWS * SB : b;
                pairs_preload<SB, FC>(f, pr, nb, BATCH_ROWS, lane, nrows, nxt);
                pass_pairs<SB, FC, EAGER, false, true>(f, pr, r, nrows, cap_masks, p, s_str, &cur);
            } else {
                pass_pairs<SB, FC, EAGER, false, false, NW>(f, pr, r, nrows, cap_masks, p, s_str, nullptr, nkey, npred0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                p[j][0] = r[j] < end; p[j][1] = r[j] + 1 < end;
                if (p[j][0]) p[j][0] = row_passes<FC>(f, pr, r[j], cap_masks);
                if (p[j][1]) p[j][1] = row_passes<FC>(f, pr, r[j] + 1, cap_masks);
            }
        }
        // Full step without eager payloads: the survivors of all SB batches are queued in LDS (row
        // order: ballot + popcount prefix) and drained 64 at a time — one gather and one store
        // instruction per column per 64 survivors.  Storing from inside the 2 x SB `if (survivor)`
        // regions instead issues every store with a tenth of its lanes; the vector-memory issue slot
        // per instruction, not bytes, bounded this kernel (same bytes, 17 % fewer: same time).
        if (!EAGER_PAY && full_step) {
            int qn = 0;
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const uint64_t b0 = __ballot(p[j][0]), b1 = __ballot(p[j][1]);
                const int at = qn + __popcll(b0 & lt) + __popcll(b1 & lt);
                const int off = j * (int)BATCH_ROWS + lane * ROWS_PER_LOAD;
                if (p[j][0]) q_off[at] = (uint16_t)off;
                if (p[j][1]) q_off[at + (p[j][0] ? 1 : 0)] = (uint16_t)(off + 1);
                qn += __popcll(b0) + __popcll(b1);
            }
            __builtin_amdgcn_wave_barrier();
            for (int i0 = 0; i0 < qn; i0 += WAVE) {
                const int i = i0 + lane;
                if (i < qn) {
                    const int64_t row = b + q_off[i];
                    int64_t pay[MAX_STAGE_COLS];
                    const int64_t key = st.src_key[row];
#pragma unroll
                    for (int c = 0; c < MAX_STAGE_COLS; ++c) pay[c] = c < cfg_npay<NPAY>(st.npay) ? st.src_pay[c][row] : 0;
                    stage_store<NPAY>(st, out + i, key, pay);
                }
            }
            __builtin_amdgcn_wave_barrier();
            out += qn;
            continue;
        }
        // Otherwise gather key + payload of every survivor of all SB batches before the first
        // store: one memory latency for the whole step, not one per divergent `if (survivor)` region.
        if (!(EAGER_PAY && full_step)) {
#pragma unroll
            for (int j = 0; j < SB; ++j) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    kx[j][q] = p[j][q] ? st.src_key[r[j] + q] : 0;
#pragma unroll
                    for (int c = 0; c < MAX_STAGE_COLS; ++c) px[j][q][c] = (c < cfg_npay<NPAY>(st.npay) && p[j][q]) ? st.src_pay[c][r[j] + q] : 0;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < SB; ++j) {
            const uint64_t b0 = __ballot(p[j][0]), b1 = __ballot(p[j][1]);
            const int64_t pos0 = out + __popcll(b0 & lt) + __popcll(b1 & lt);
            if (p[j][0]) stage_store<NPAY>(st, pos0, kx[j][0], px[j][0]);
            if (p[j][1]) stage_store<NPAY>(st, pos0 + (p[j][0] ? 1 : 0), kx[j][1], px[j][1]);
            out += __popcll(b0) + __popcll(b1);
        }
    }
    if (lane == 0) st.seg_count[seg] = (uint32_t)(out - begin);
}

// Every workgroup recomputes the staged total from the segment counts (a few KB from L2), so the
// capacity is agreed on without a grid barrier; workgroup 0 publishes the header for later kernels.
SDQH_KERNEL __launch_bounds__(TPB) void k_clear(const uint32_t* __restrict__ seg_count, int nseg, uint64_t capmax,
                                               TableHeader* __restrict__ hdr, int64_t* __restrict__ keys, uint32_t* __restrict__ rowref, int64_t* __restrict__ slots,
                                               uint32_t* __restrict__ hf) {
    __shared__ unsigned long long s_part[TPB];
    unsigned long long t = 0;
    for (int i = threadIdx.x; i < nseg; i += TPB) t += seg_count[i];
    s_part[threadIdx.x] = t;
    __syncthreads();
    for (int off = TPB / 2; off > 0; off >>= 1) { if ((int)threadIdx.x < off) s_part[threadIdx.x] += s_part[threadIdx.x + off]; __syncthreads(); }
    const uint64_t staged = s_part[0];
    uint64_t cap = 1024;
    while (cap < 2 * staged) cap <<= 1;
    if (cap > capmax) cap = capmax;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        hdr->cap_mask = cap - 1; hdr->staged = staged;
        if (slots) { slots[cap * 4] = EMPTY_KEY; slots[cap * 4 + 3] = (int64_t)NO_ROW; }
        else { keys[cap] = EMPTY_KEY; rowref[cap] = NO_ROW; }  // the extra slot that holds a real key == EMPTY_KEY
    }
    if (hf) {                                                  // the hashed filter's words (its size follows the capacity: hf_mask_of)
        using W = uint32_t __attribute__((ext_vector_type(4)));
        const uint32_t hm = hf_mask_of(cap - 1);
        const uint64_t n16 = ((uint64_t)hm + 1) / 128;
        const uint32_t z = hm == HF_DEGENERATE ? 0xFFFFFFFFu : 0u;
        const W zero = {z, z, z, z};
        W* h4 = reinterpret_cast<W*>(hf);
        for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * TPB) h4[i] = zero;
    }
    using V = long long __attribute__((ext_vector_type(2)));
    if (slots) {                                               // packed: { EMPTY_KEY, 0 } { 0, NO_ROW } per slot, two 16-byte stores
        const V a = {(long long)EMPTY_KEY, 0}, b = {0, (long long)NO_ROW};
        V* s2 = reinterpret_cast<V*>(slots);
        for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * TPB) { s2[2 * i] = a; s2[2 * i + 1] = b; }
        return;
    }
    const V empty = {(long long)EMPTY_KEY, (long long)EMPTY_KEY};
    V* k2 = reinterpret_cast<V*>(keys);
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < cap / 2; i += (uint64_t)gridDim.x * TPB) k2[i] = empty;
}

// ---- direct layout: prefix popcounts over the bitmap, then plain stores ---------------------------
// One workgroup per 2048 bitmap words: popcount, exclusive scan inside the block, and a base for
// the block claimed with one atomicAdd on hdr->distinct, folded into the per-word prefix.  Ranks
// are a bijection onto [0, distinct), not monotone in the key: nothing needs more (the rank only
// indexes dense_ref), and a claimed base needs no scan pass, no second launch and no device-wide
// fence (a release fence writes back a whole XCD L2 here).  hdr->staged is summed on the side;
// fewer distinct keys than staged rows means duplicate build keys (see direct_has_dups).
__device__ __forceinline__ void rank_words_body(const uint32_t* __restrict__ bm, uint64_t nwords, uint32_t* __restrict__ wprefix,
                                                const uint32_t* __restrict__ seg_count, int nseg, TableHeader* __restrict__ hdr) {
    // A block's 8192 words as 8 rows of 256 16-byte quads, quad (q, t) to thread t: every load and every store of a wave is 1 KiB of
    // consecutive bytes (32 consecutive words per thread — 128-byte strides between lanes, the prefixes stored word by word — made this
    // kernel 16 us for the 7.5 MB bitmap of Q3's orders).  The prefix runs in memory order: row after row, a wave scan inside each.
    __shared__ uint32_t s_wave[TPB / WAVE];
    __shared__ uint32_t s_base;
    constexpr int ROWS = RANK_BLOCK_WORDS / (TPB * 4);                 // 8 rows of quads
    const uint64_t q0 = (uint64_t)blockIdx.x * (RANK_BLOCK_WORDS / 4);  // first quad of the block
    const uint64_t nquads = (nwords + 3) / 4;                          // (the bitmap and the prefix array are allocated with slack: a last partial quad is whole memory)
    const int w = threadIdx.x / WAVE, lane = lane_id();
    uint4 quad[ROWS];
    uint32_t excl[ROWS];                                               // exclusive prefix of each quad inside the block
    uint32_t running = 0;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint64_t qi = q0 + (uint64_t)r * TPB + threadIdx.x;
        uint4 a = make_uint4(0u, 0u, 0u, 0u);
        if (qi < nquads) {
            a = *reinterpret_cast<const uint4*>(bm + qi * 4);
            const uint64_t wi = qi * 4;                                // words past the bitmap's end count nothing
            if (wi + 1 >= nwords) a.y = 0u;
            if (wi + 2 >= nwords) a.z = 0u;
            if (wi + 3 >= nwords) a.w = 0u;
        }
        quad[r] = a;
        const uint32_t mine = (uint32_t)(__popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w));
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) { const uint32_t v = __shfl_up(incl, off, WAVE); if (lane >= off) incl += v; }
        __syncthreads();                                               // (the previous row's totals have been read)
        if (lane == WAVE - 1) s_wave[w] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < TPB / WAVE; ++i) { if (i < w) before += s_wave[i]; total += s_wave[i]; }
        excl[r] = running + before + incl - mine;
        running += total;
    }
    // staged rows: every workgroup adds its share of the segment counts
    uint32_t stg = 0;
    for (int i = blockIdx.x * TPB + threadIdx.x; i < nseg; i += gridDim.x * TPB) stg += seg_count[i];
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) stg += __shfl_down(stg, off, WAVE);
    if (lane == 0 && stg) atomicAdd(reinterpret_cast<unsigned long long*>(&hdr->staged), (unsigned long long)stg);
    if (threadIdx.x == 0) s_base = running ? (uint32_t)atomicAdd(reinterpret_cast<unsigned long long*>(&hdr->distinct), (unsigned long long)running) : 0u;
    __syncthreads();
    const uint32_t base = s_base;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const uint64_t qi = q0 + (uint64_t)r * TPB + threadIdx.x;
        if (qi >= nquads) continue;
        uint4 p;
        p.x = base + excl[r];
        p.y = p.x + (uint32_t)__popc(quad[r].x);
        p.z = p.y + (uint32_t)__popc(quad[r].y);
        p.w = p.z + (uint32_t)__popc(quad[r].z);
        *reinterpret_cast<uint4*>(wprefix + qi * 4) = p;
    }
}
SDQH_KERNEL __launch_bounds__(TPB) void k_rank_words(const uint32_t* __restrict__ bm, uint64_t nwords, uint32_t* __restrict__ wprefix,
                                                    const uint32_t* __restrict__ seg_count, int nseg, TableHeader* __restrict__ hdr) {
    rank_words_body(bm, nwords, wprefix, seg_count, nseg, hdr);
}
// Row index, after the build (one thread per segment): does the segment's first entry open its bitmap word, or does the word already
// hold keys of the segment before (then it is an exception word, see DevTable)?  The bitmap is complete by now.
SDQH_KERNEL __launch_bounds__(TPB) void k_wrow_fixup(DevStage st, WordExc* __restrict__ exc, TableHeader* __restrict__ hdr) {
    const int s = (int)(blockIdx.x * TPB + threadIdx.x);
    if (s == 0) hdr->has_dups = 0;
    if (s >= st.nseg) return;
    const uint32_t cnt = st.seg_count[s];
    if (!cnt) return;
    const SegFirst f = st.seg_first[s];
    const uint64_t off0 = (uint64_t)(f.key - st.bm_lo);
    const uint32_t w0 = (uint32_t)(off0 >> 5), word = st.bm[w0];
    const uint32_t n_below = __popc(word & ((1u << (off0 & 31)) - 1u));
    if (n_below == 0) {
        // the word's first key is mine.  If a LATER segment has keys in this word too it will mark the word as an exception: then
        // (and only then) the word holds more keys than my segment has entries — the keys of the word are the smallest keys from mine on,
        // and my entries come in key order — and I leave it alone
        if (cnt >= (uint32_t)__popc(word)) st.wrow[w0] = f.pos;
        return;
    }
    int t = s - 1;
    while (t > 0 && st.seg_count[t] == 0) --t;                               // the segment before me that has entries: the n_below keys are its last ones
    WordExc e;
    e.key0 = f.key; e.pos0 = f.pos; e.n0 = n_below; e._pad = 0;
    e.pos_prev = (uint32_t)((int64_t)t * st.seg_rows + (int64_t)st.seg_count[t] - (int64_t)n_below);
    exc[s] = e;
    st.wrow[w0] = ROW_INDEX_EXC | (uint32_t)s;
}
// direct layout, after k_rank_words: duplicate build keys <=> fewer set bits than staged rows
__device__ __forceinline__ bool direct_has_dups(const TableHeader* hdr) { return hdr->staged != hdr->distinct; }
// dense_ref[rank(key)] = stage index.  Unique build keys: plain stores, every row its own rank.
// After a duplicate was seen while staging: atomicMin into a NO_ROW-filled array (lowest row wins).
SDQH_KERNEL __launch_bounds__(TPB) void k_fill_refs(DevStage st, DevTable t) {        // no-op unless duplicates
    if (!direct_has_dups(t.hdr)) return;
    const uint64_t n = t.hdr->distinct;
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) t.dense_ref[i] = NO_ROW;
}
// span (optional): the same owner by key offset, span[key - bm_lo] — for a table over a small key range a lookup is then
// ONE load (the dense layout's path) instead of bitmap word + rank prefix + dense_ref; NO_ROW-filled by the build's k_fill
__device__ __forceinline__ void insert_direct_body(const DevStage& st, const DevTable& t, uint32_t* __restrict__ span) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    const bool dups = direct_has_dups(t.hdr);
    if (blockIdx.x == 0 && threadIdx.x == 0) t.hdr->has_dups = dups ? 1u : 0u;     // for every later reader
    if (seg >= st.nseg) return;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    for (uint32_t i = lane_id(); i < count; i += WAVE) {
        const int64_t idx = base + i;
        const int64_t pos = table_find(t, st.key[idx], 0);
        if (pos < 0) continue;                                         // cannot happen: the key's bit was set while staging
        if (dups) atomicMin(&t.dense_ref[pos], (uint32_t)idx); else t.dense_ref[pos] = (uint32_t)idx;
        if (span) { uint32_t* cell = span + (st.key[idx] - t.bm_lo); if (dups) atomicMin(cell, (uint32_t)idx); else *cell = (uint32_t)idx; }
    }
}
SDQH_KERNEL __launch_bounds__(TPB) void k_insert_direct(DevStage st, DevTable t, uint32_t* __restrict__ span) { insert_direct_body(st, t, span); }
// rank + insert of a tiny table (one rank block, one workgroup of segments) in one launch
SDQH_KERNEL __launch_bounds__(TPB) void k_index_small(const uint32_t* __restrict__ bm, uint64_t nwords, uint32_t* __restrict__ wprefix, DevStage st, DevTable t, uint32_t* __restrict__ span) {
    rank_words_body(bm, nwords, wprefix, st.seg_count, st.nseg, t.hdr);
    __threadfence();
    __syncthreads();
    insert_direct_body(st, t, span);
}
// ... and of a table whose key bitmap is ONE rank block (<= 4096 words: a key range of 131 072 — the suppliers of Q5 / Q9, most
// dimension tables) but whose rows are many workgroups of segments: every workgroup ranks the whole bitmap for itself in LDS (16 KB
// out of L2; nobody waits for anybody), workgroup 0 also stores the prefixes and the header for the lookups to come, and each inserts
// its own segments with the ranks it holds.  dense_ref (and span) must be NO_ROW-filled already when keys can repeat (prefill_refs).
SDQH_KERNEL __launch_bounds__(TPB) void k_index_medium(const uint32_t* __restrict__ bm, uint32_t nwords, uint32_t* __restrict__ wprefix, DevStage st, DevTable t, uint32_t* __restrict__ span) {
    __shared__ uint32_t s_word[RANK_BLOCK_WORDS], s_pre[RANK_BLOCK_WORDS];
    __shared__ uint32_t s_wave[TPB / WAVE];
    __shared__ uint32_t s_staged;
    constexpr int PER = RANK_BLOCK_WORDS / TPB;                          // consecutive words per thread
    const int lane = lane_id(), w = threadIdx.x / WAVE;
    for (int i = threadIdx.x; i < RANK_BLOCK_WORDS; i += TPB) s_word[i] = (uint32_t)i < nwords ? bm[i] : 0u;
    if (threadIdx.x == 0) s_staged = 0u;
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) mine += (uint32_t)__popc(s_word[threadIdx.x * PER + j]);
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { const uint32_t v = __shfl_up(incl, off, WAVE); if (lane >= off) incl += v; }
    if (lane == WAVE - 1) s_wave[w] = incl;
    uint32_t stg = 0;
    for (int i = threadIdx.x; i < st.nseg; i += TPB) stg += st.seg_count[i];
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) stg += __shfl_down(stg, off, WAVE);
    if (lane == 0 && stg) atomicAdd(&s_staged, stg);
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int i = 0; i < TPB / WAVE; ++i) { if (i < w) before += s_wave[i]; total += s_wave[i]; }
    uint32_t run = before + incl - mine;
#pragma unroll
    for (int j = 0; j < PER; ++j) { s_pre[threadIdx.x * PER + j] = run; run += (uint32_t)__popc(s_word[threadIdx.x * PER + j]); }
    __syncthreads();
    const uint32_t staged = s_staged;
    const bool dups = staged != total;
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < nwords; i += TPB) wprefix[i] = s_pre[i];
        if (threadIdx.x == 0) { t.hdr->staged = staged; t.hdr->distinct = total; t.hdr->has_dups = dups ? 1u : 0u; }
    }
    const int seg = blockIdx.x * (TPB / WAVE) + w;
    if (seg >= st.nseg) return;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    for (uint32_t i = lane; i < count; i += WAVE) {
        const int64_t idx = base + i;
        const int64_t key = st.key[idx];
        uint64_t off;
        if (!bm_locate(t, key, off) || (off >> 5) >= (uint64_t)RANK_BLOCK_WORDS) continue;
        const uint32_t word = s_word[off >> 5];
        if (!((word >> (off & 31)) & 1u)) continue;                       // cannot happen: the key's bit was set while staging
        const uint32_t pos = s_pre[off >> 5] + (uint32_t)__popc(word & ((1u << (off & 31)) - 1u));
        if (dups) atomicMin(&t.dense_ref[pos], (uint32_t)idx); else t.dense_ref[pos] = (uint32_t)idx;
        if (span) { uint32_t* cell = span + (key - t.bm_lo); if (dups) atomicMin(cell, (uint32_t)idx); else *cell = (uint32_t)idx; }
    }
}

// ---- dense layout ---------------------------------------------------------------------------------------
// every row of the build table is an entry: segment counts are simply the segment lengths
SDQH_KERNEL __launch_bounds__(TPB) void k_full_counts(uint32_t* __restrict__ seg_count, int nseg, int64_t seg_rows, int64_t nrows) {
    const int s = blockIdx.x * TPB + threadIdx.x;
    if (s < nseg) { int64_t b = (int64_t)s * seg_rows, e = b + seg_rows; if (e > nrows) e = nrows; seg_count[s] = (uint32_t)(e - b); }
}
// dense_arr[key - lo] = row: plain stores (a unique key writes its own cell; sorted keys coalesce)
SDQH_KERNEL __launch_bounds__(TPB) void k_dense_fill(const int64_t* __restrict__ key, int64_t nrows, int64_t lo, uint32_t* __restrict__ arr) {
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * TPB) arr[key[r] - lo] = (uint32_t)r;
}
// Is the column strictly increasing (sorted, no duplicates)?  flag[0] |= 1 where it is not.  A property of the resident
// column, asked once (sdqh_column::sorted_unique) like its min / max.
SDQH_KERNEL __launch_bounds__(TPB) void k_check_increasing(const int64_t* __restrict__ key, int64_t nrows, int* __restrict__ flag) {
    bool bad = false;
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r + 1 < nrows; r += (int64_t)gridDim.x * TPB) bad |= key[r] >= key[r + 1];
    if (__ballot(bad) && lane_id() == 0) atomicOr(flag, 1);
}
// ... or at least never decreasing (a table stored in the order of this column: equal values are neighbours)?
SDQH_KERNEL __launch_bounds__(TPB) void k_check_nondecreasing(const int64_t* __restrict__ key, int64_t nrows, int* __restrict__ flag) {
    bool bad = false;
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r + 1 < nrows; r += (int64_t)gridDim.x * TPB) bad |= key[r] > key[r + 1];
    if (__ballot(bad) && lane_id() == 0) atomicOr(flag, 1);
}
// ... or two columns strictly increasing AS PAIRS (a, b), row after row (a composite key of a table stored in its order: no two rows share a key)?
SDQH_KERNEL __launch_bounds__(TPB) void k_check_pair_increasing(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t nrows, int* __restrict__ flag) {
    bool bad = false;
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r + 1 < nrows; r += (int64_t)gridDim.x * TPB) bad |= !(a[r] < a[r + 1] || (a[r] == a[r + 1] && b[r] < b[r + 1]));
    if (__ballot(bad) && lane_id() == 0) atomicOr(flag, 1);
}
// ... or, the first column never decreasing, no two rows of one run of equal first parts sharing their second part (partsupp by (ps_partkey,
// ps_suppkey): the suppliers of a part come in the generator's order, not sorted, but none twice).  A run longer than PAIR_RUN_MAX rows is
// not examined to its end: flag bit 2, "not known".
constexpr int PAIR_RUN_MAX = 32;
SDQH_KERNEL __launch_bounds__(TPB) void k_check_pair_distinct(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t nrows, int* __restrict__ flag) {
    int bad = 0;
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r + 1 < nrows; r += (int64_t)gridDim.x * TPB) {
        const int64_t ar = a[r], br = b[r];
        int64_t j = r + 1;
        for (; j < nrows && j <= r + PAIR_RUN_MAX && a[j] == ar; ++j) bad |= b[j] == br ? 1 : 0;
        if (j < nrows && j > r + PAIR_RUN_MAX && a[j] == ar) bad |= 2;
    }
    if (bad) atomicOr(flag, bad);
}
// Dense layout over a strictly increasing key column, in ONE pass: row r writes its own cell and NO_ROW into the cells up to
// the next key, so the array needs no prefill (240 MB for Q9's orders) and — no duplicates possible — no verification pass
// (another read of keys and cells).  Every cell in [lo, hi] is written exactly once; lo / hi are the column's min / max.
constexpr int DENSE_SPAN = 2048;                                      // cells a wave assembles in LDS per step (8 KiB)
SDQH_KERNEL __launch_bounds__(TPB) void k_dense_fill_increasing(const int64_t* __restrict__ key, int64_t nrows, int64_t lo, uint32_t* __restrict__ arr, TableHeader* __restrict__ hdr) {
    // A wave takes 128 consecutive rows, assembles the cells from its first key up to (not including) the first key of
    // the next 128 rows in LDS — NO_ROW everywhere, then each row number at its key — and writes them out with
    // 16-byte stores.  (One thread per row storing its own gap ran 3x slower: 4-byte stores at a stride of a few cells.)
    __shared__ __align__(16) uint32_t s_cells[TPB / WAVE][DENSE_SPAN];
    uint32_t* cells = s_cells[threadIdx.x / WAVE];
    const int lane = lane_id();
    constexpr int ROWS = 2 * WAVE;
    const int64_t nsteps = (nrows + ROWS - 1) / ROWS;
    const int64_t wave = (int64_t)blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE, nwaves = (int64_t)gridDim.x * (TPB / WAVE);
    for (int64_t step = wave; step < nsteps; step += nwaves) {
        const int64_t r0 = step * ROWS;
        const int64_t ra = r0 + lane, rb = r0 + WAVE + lane;
        const int64_t ka = ra < nrows ? key[ra] : 0, kb = rb < nrows ? key[rb] : 0;
        const int64_t first = __shfl(ka, 0, WAVE);
        const int64_t rend = r0 + ROWS;
        const int64_t last = rend < nrows ? key[rend] : key[nrows - 1] + 1;      // exclusive end of this wave's cells (uniform address)
        for (int64_t w0 = first; w0 < last; w0 += DENSE_SPAN) {                 // one window, except across a rare wide gap
            const int64_t span = last - w0 < DENSE_SPAN ? last - w0 : DENSE_SPAN;
            for (int i = lane * 4; i < span; i += WAVE * 4) *reinterpret_cast<uint4*>(&cells[i]) = make_uint4(NO_ROW, NO_ROW, NO_ROW, NO_ROW);
            __builtin_amdgcn_wave_barrier();
            if (ra < nrows && ka >= w0 && ka < w0 + span) cells[ka - w0] = (uint32_t)ra;
            if (rb < nrows && kb >= w0 && kb < w0 + span) cells[kb - w0] = (uint32_t)rb;
            __builtin_amdgcn_wave_barrier();
            uint32_t* out = arr + (w0 - lo);
            // head up to 16-byte alignment of the destination, 16-byte body, tail
            const int head = (int)((4 - (((unsigned long long)out >> 2) & 3)) & 3);
            if (lane < head && lane < span) out[lane] = cells[lane];
            const int64_t body = span > head ? (span - head) / 4 : 0;
            for (int64_t v = lane; v < body; v += WAVE) {
                const int c = head + (int)v * 4;
                *reinterpret_cast<uint4*>(out + c) = make_uint4(cells[c], cells[c + 1], cells[c + 2], cells[c + 3]);
            }
            const int64_t done = head + body * 4;
            if (done + lane < span) out[done + lane] = cells[done + lane];
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr->staged = (uint64_t)nrows; hdr->has_dups = 0; }
}
// Direct layout over a strictly increasing key column where EVERY row is an entry (a whole table joined on its sorted primary
// key): the rank of a key is its row number, so no owner array at all — the key bitmap and, per bitmap word, the row of the
// word's first key (wprefix; words without keys are never consulted).  A wave assembles the words of its 128 rows in LDS and
// stores the interior ones plainly (keys increase: no other wave has a key in them); the first and the last word may be shared
// with the neighbouring waves and are ORed in.  15 MB written for Q9's orders instead of the dense array's 240 MB.
constexpr int RANK_INC_WORDS = 1024;                                  // bitmap words a wave assembles per step (4 KiB)
constexpr int RANK_INC_NB = 8;                                        // blocks of 64 rows per wave step: every key of the step is requested before any is used
// KT: the key column itself (int64_t) or its exact 4-byte twin (int32_t: half the bytes of the one pass this build is)
template <class KT>
__global__ __launch_bounds__(TPB) void k_rank_increasing(const KT* __restrict__ key, int64_t nrows, int64_t lo, uint32_t* __restrict__ bm,
                                                        uint32_t* __restrict__ wprefix, TableHeader* __restrict__ hdr,
                                                        uint32_t* __restrict__ seg_count, int nseg, int64_t seg_rows, uint32_t* __restrict__ wpair) {
    // wpair (or null): { wprefix[w], bm[w] } pairs, zero-filled like bm — every 4-byte half is written as its array is: a word's row by the one
    // lane that holds its first key, its bits plainly where the wave owns the word and ORed in where two waves share it
    // (every row is an entry: a segment's count is its length — what k_full_counts wrote in a launch of its own)
    for (int sg = blockIdx.x * TPB + threadIdx.x; sg < nseg; sg += gridDim.x * TPB) {
        const int64_t b = (int64_t)sg * seg_rows, e = b + seg_rows < nrows ? b + seg_rows : nrows;
        seg_count[sg] = (uint32_t)(e - b);
    }
    __shared__ uint32_t s_words[TPB / WAVE][RANK_INC_WORDS];
    uint32_t* words = s_words[threadIdx.x / WAVE];
    const int lane = lane_id();
    constexpr int NB = RANK_INC_NB, ROWS = NB * WAVE;
    const int64_t nsteps = (nrows + ROWS - 1) / ROWS;
    const int64_t wave = (int64_t)blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE, nwaves = (int64_t)gridDim.x * (TPB / WAVE);
    for (int64_t step = wave; step < nsteps; step += nwaves) {
        const int64_t r0 = step * ROWS;
        uint64_t o[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) { const int64_t r = r0 + (int64_t)j * WAVE + lane; o[j] = r < nrows ? (uint64_t)((int64_t)key[r] - lo) : 0; }
        const int64_t rlast = (r0 + ROWS <= nrows ? r0 + ROWS : nrows) - 1;
        const uint64_t before = r0 > 0 ? (uint64_t)((int64_t)key[r0 - 1] - lo) : ~0ull;             // (uniform addresses)
        const uint64_t wl = (uint64_t)((int64_t)key[rlast] - lo) >> 5;
        const uint64_t wf = __shfl(o[0], 0, WAVE) >> 5;
        // a row that is the first key of its bitmap word records its row number there
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int64_t r = r0 + (int64_t)j * WAVE + lane;
            uint64_t prev = __shfl_up(o[j], 1, WAVE);
            const uint64_t carry = j == 0 ? before : __shfl(o[j > 0 ? j - 1 : 0], WAVE - 1, WAVE);
            if (lane == 0) prev = carry;
            if (r < nrows && (prev == ~0ull || (prev >> 5) != (o[j] >> 5))) { wprefix[o[j] >> 5] = (uint32_t)r; if (wpair) wpair[2 * (o[j] >> 5)] = (uint32_t)r; }
        }
        const uint64_t nw = wl - wf + 1;
        if (nw <= (uint64_t)RANK_INC_WORDS) {
            for (int i = lane; i < (int)nw; i += WAVE) words[i] = 0u;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < NB; ++j) { const int64_t r = r0 + (int64_t)j * WAVE + lane; if (r < nrows) atomicOr(&words[(o[j] >> 5) - wf], 1u << (o[j] & 31)); }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < (int)nw; i += WAVE) {
                const uint32_t w = words[i];
                if (i == 0 || i == (int)nw - 1) { if (w) { atomicOr(&bm[wf + i], w); if (wpair) atomicOr(&wpair[2 * (wf + i) + 1], w); } }
                else { bm[wf + i] = w; if (wpair) wpair[2 * (wf + i) + 1] = w; }
            }
            __builtin_amdgcn_wave_barrier();
        } else {                                                       // a wide gap inside the step: bit by bit
#pragma unroll
            for (int j = 0; j < NB; ++j) { const int64_t r = r0 + (int64_t)j * WAVE + lane; if (r < nrows) { atomicOr(&bm[o[j] >> 5], 1u << (o[j] & 31)); if (wpair) atomicOr(&wpair[2 * (o[j] >> 5) + 1], 1u << (o[j] & 31)); } }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr->staged = (uint64_t)nrows; hdr->distinct = (uint64_t)nrows; hdr->has_dups = 0; }
}
// a row that does not find itself in its cell lost to a duplicate key
SDQH_KERNEL __launch_bounds__(TPB) void k_dense_verify(const int64_t* __restrict__ key, int64_t nrows, int64_t lo, const uint32_t* __restrict__ arr, TableHeader* __restrict__ hdr) {
    bool dup = false;
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * TPB) dup |= arr[key[r] - lo] != (uint32_t)r;
    if (__ballot(dup) && lane_id() == 0) hdr->has_dups = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) hdr->staged = (uint64_t)nrows;
}
// only with duplicates: the lowest row wins
SDQH_KERNEL __launch_bounds__(TPB) void k_dense_fixup(const int64_t* __restrict__ key, int64_t nrows, int64_t lo, uint32_t* __restrict__ arr, const TableHeader* __restrict__ hdr) {
    if (hdr->has_dups == 0) return;
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * TPB) atomicMin(&arr[key[r] - lo], (uint32_t)r);
}

SDQH_KERNEL __launch_bounds__(TPB) void k_insert(DevStage st, DevTable t) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const uint64_t mask = t.hdr->cap_mask;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    const uint32_t hmask = hf_mask_of(mask);
    for (uint32_t i = lane_id(); i < count; i += WAVE) {
        const int64_t idx = base + i;
        const int64_t key = st.key[idx];
        if (t.hf && hmask != HF_DEGENERATE) { const uint32_t code = hf_code(hf_raw(key), hmask); atomicOr(&t.hf[(code & HF_POS_MASK) >> 5], hf_bits(code)); }      // (fire and forget: no value returns)
        if (key == EMPTY_KEY) {                                          // the sentinel value itself lives in the extra slot
            uint32_t* ref = t.slots ? reinterpret_cast<uint32_t*>(&t.slots[(mask + 1) * 4 + 3]) : &t.rowref[mask + 1];
            if (atomicMin(ref, (uint32_t)idx) != NO_ROW) t.hdr->has_dups = 1;
            else if (t.slots) { t.slots[(mask + 1) * 4 + 1] = st.npay > 0 ? st.pay[0][idx] : 0; t.slots[(mask + 1) * 4 + 2] = st.npay > 1 ? st.pay[1][idx] : 0; }
            continue;
        }
        uint64_t h = hash_key(key) & mask;
        for (;;) {
            unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(t.slots ? &t.slots[h * 4] : &t.keys[h]),
                                               (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (old == (unsigned long long)EMPTY_KEY) {                  // claimed a fresh slot
                if (t.slots) {
                    t.slots[h * 4 + 1] = st.npay > 0 ? st.pay[0][idx] : 0; t.slots[h * 4 + 2] = st.npay > 1 ? st.pay[1][idx] : 0;
                    t.slots[h * 4 + 3] = (int64_t)(uint32_t)idx;
                } else t.rowref[h] = (uint32_t)idx;
                break;
            }
            if ((int64_t)old == key) { t.hdr->has_dups = 1; break; }      // duplicate build key: settled by k_insert_fixup
            h = (h + 1) & mask;
        }
    }
}

// Only does work when k_insert saw a duplicate build key: every staged row then lowers its entry's
// rowref to its own stage index, so the lowest build row wins whatever the insertion order was.
// (Stage order is row order: segments are cut in row order and compacted in row order.)
SDQH_KERNEL __launch_bounds__(TPB) void k_insert_fixup(DevStage st, DevTable t) {
    if (t.hdr->has_dups == 0) return;
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const uint64_t mask = t.hdr->cap_mask;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    for (uint32_t i = lane_id(); i < count; i += WAVE) {
        const int64_t idx = base + i;
        const int64_t h = table_find(t, st.key[idx], mask);
        if (h >= 0) atomicMin(t.slots ? reinterpret_cast<uint32_t*>(&t.slots[h * 4 + 3]) : &t.rowref[h], (uint32_t)idx);
    }
}
// packed slots after a fix-up: the payload in the slot must be the owner's (its own launch: every atomicMin above has landed)
SDQH_KERNEL __launch_bounds__(TPB) void k_insert_repack(DevStage st, DevTable t) {
    if (t.hdr->has_dups == 0 || !t.slots) return;
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const uint64_t mask = t.hdr->cap_mask;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    for (uint32_t i = lane_id(); i < count; i += WAVE) {
        const int64_t idx = base + i;
        const int64_t h = table_find(t, st.key[idx], mask);
        if (h >= 0 && slot_row(t, (uint64_t)h) == (uint32_t)idx) {
            t.slots[h * 4 + 1] = st.npay > 0 ? st.pay[0][idx] : 0; t.slots[h * 4 + 2] = st.npay > 1 ? st.pay[1][idx] : 0;
        }
    }
}

// Is stage row idx the owner of its entry?  Always, unless the build met duplicate keys.
__device__ __forceinline__ bool stage_row_owns(const DevStage& st, const DevTable& t, int64_t idx, uint64_t mask) {
    if (t.hdr->has_dups == 0) return true;
    const int64_t h = table_find(t, st.key[idx], mask);
    return h >= 0 && table_ref(t, h) == (uint32_t)idx;
}

// distinct entries (table_size)
SDQH_KERNEL __launch_bounds__(TPB) void k_count(DevStage st, DevTable t) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const uint64_t mask = table_is_direct(t) ? 0 : t.hdr->cap_mask;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    unsigned long long n = 0;
    for (uint32_t i = lane_id(); i < count; i += WAVE) n += stage_row_owns(st, t, base + i, mask) ? 1 : 0;
    n = (unsigned long long)wave_sum_i64((int64_t)n);
    if (lane_id() == 0 && n) atomicAdd(reinterpret_cast<unsigned long long*>(&t.hdr->counted), n);
}

// sdqh_table_share_groups: entries whose payload fields agree become one group.  Pass 1 finds the
// lowest owning stage row of every cell of the fields' value rectangle (atomicMin), pass 2 points every
// stage row at it.  One wave per stage segment, as the compaction; 4 + 8 * nfields bytes per entry.
struct DevShare { const int64_t* col[SDQH_MAX_PAYLOAD]; int64_t lo[SDQH_MAX_PAYLOAD], span[SDQH_MAX_PAYLOAD]; int32_t n, _pad; };
template <bool FIRST>
__global__ __launch_bounds__(TPB) void k_share_groups(DevTable t, DevStage st, DevShare s, uint32_t* __restrict__ first, uint32_t* __restrict__ alias) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const uint64_t mask = table_is_direct(t) ? 0 : t.hdr->cap_mask;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    for (uint32_t i = lane_id(); i < count; i += WAVE) {
        const int64_t idx = base + i;
        bool inside = true;
        uint64_t cell = 0;
#pragma unroll
        for (int q = 0; q < SDQH_MAX_PAYLOAD; ++q) if (q < s.n) {
            const int64_t v = s.col[q][idx] - s.lo[q];
            inside = inside && v >= 0 && v < s.span[q];
            cell = cell * (uint64_t)s.span[q] + (uint64_t)v;
        }
        if (FIRST) {
            if (inside && stage_row_owns(st, t, idx, mask)) atomicMin(&first[cell], (uint32_t)idx);
        } else {
            const uint32_t a = inside ? first[cell] : NO_ROW;
            alias[idx] = a == NO_ROW ? (uint32_t)idx : a;
        }
    }
}

// =================================================================================================
// K-C large: probe + aggregate into the matched entry.  The date-like first predicate and the probe
// key are streamed with 16-byte loads, PROBE_UNROLL pairs per lane in flight; the bitmap word, the
// slot and the value operands are only read by the lanes that are still alive (the reference
// short-circuits the same way: `if (pred) if (contains) { ... += ep*(1.0-disc) }`).  Hits are rare
// and scattered, so native f64 global atomics are the right tool here.
// =================================================================================================
constexpr int PROBE_UNROLL = 2;
constexpr int PROBE_TILE = TPB * ROWS_PER_LOAD * PROBE_UNROLL;       // 2048 rows per workgroup step

// a row whose key is in the table: add its tuple to the owning entry
template <int SHAPE>
__device__ __forceinline__ void probe_add(const DevFilter& f, const DevTuple& t, const DevTable& tb, int64_t pos, int64_t r) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV;
    double x[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < NOPS; ++j) x[j] = t.op[j][r];
    if (!operand_ranges<NOPS>(f, x)) return;
    tuple_eval<SHAPE>(x, o);
    uint32_t idx = table_ref(tb, pos);
    if (tb.alias) idx = tb.alias[idx];
#pragma unroll
    for (int k = 0; k < NV; ++k) atomicAdd(&tb.sacc[(size_t)idx * tb.acc_stride + k], o[k]);
    atomicAdd(&tb.shits[idx], 1u);
}

// Rows that pass the filter (and, with the direct layout, the bitmap test) are rare and scattered
// over the lanes.  Handling each where it is found would make every `if (hit)` region run its own
// chain of dependent loads (index -> owner row -> operands -> atomics), one after another, in most
// waves.  Instead each wave appends its candidates to a small queue in LDS (ballot + popcount
// prefix) and keeps streaming; whenever 64 are queued they are processed one per lane, converged:
// one chain of latencies per 64 candidates.

// Candidates are queued in row order, so rows of one group (the lineitems of an order; every row of
// a group-by on a clustered key) sit in adjacent lanes: a segmented scan over runs of equal entries
// folds them before the atomics — one atomic per run instead of one per row (Q18's sum per
// l_orderkey: 4.5 -> ~1.2 ms for 60 M rows).  Head flags keep it exact for any order of entries.
template <int SHAPE>
__device__ __forceinline__ void probe_drain(const DevFilter& f, const DevTuple& t, const DevTable& tb, uint64_t mask,
                                            int64_t qbase, const int32_t* q_row, const int64_t* q_key, int first, int n) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV;
    const int lane = lane_id();
    uint32_t idx = NO_ROW, cnt = 0;
    double o[4] = {0, 0, 0, 0};
    if (lane < n) {
        const int64_t r = qbase + (int64_t)q_row[first + lane], key = q_key[first + lane];
        const int64_t pos = table_find(tb, key, mask);
        if (pos >= 0) {
            double x[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < NOPS; ++j) x[j] = t.op[j][r];
            if (operand_ranges<NOPS>(f, x)) {
                tuple_eval<SHAPE>(x, o); idx = table_ref(tb, pos); cnt = 1;
                if (tb.alias) idx = tb.alias[idx];                 // entries of one group share the first one's accumulators
            }
        }
    }
    const uint32_t prev = __shfl_up(idx, 1, WAVE);
    uint32_t head = (lane == 0 || prev != idx) ? 1u : 0u;            // this lane starts a run of equal entries
    const uint32_t next_head = __shfl_down(head, 1, WAVE);
    const bool tail = idx != NO_ROW && (lane == WAVE - 1 || next_head != 0u);
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t oc = __shfl_up(cnt, off, WAVE), oh = __shfl_up(head, off, WAVE);
        double ov[NV > 0 ? NV : 1];
#pragma unroll
        for (int k = 0; k < NV; ++k) ov[k] = __shfl_up(o[k], off, WAVE);
        if (lane >= off && !head) {
            cnt += oc;
#pragma unroll
            for (int k = 0; k < NV; ++k) o[k] += ov[k];
            head |= oh;
        }
    }
    if (tail) {
#pragma unroll
        for (int k = 0; k < NV; ++k) atomicAdd(&tb.sacc[(size_t)idx * tb.acc_stride + k], o[k]);
        atomicAdd(&tb.shits[idx], cnt);
    }
}

// NW: the streamed key column and the first integer predicate column are read through their narrow twins (nkey, npred0: int32);
// everything read by row (the tail, further predicates, the operands of a hit) stays on the columns themselves.
template <int SHAPE, class FC, int PU = PROBE_UNROLL, bool NW = false>
__global__ __launch_bounds__(TPB) void k_probe_agg(DevFilter f, DevTuple t, DevTable tb, const int64_t* __restrict__ keycol, int64_t nrows, int chunk, int pipeline,
                                                   const int32_t* __restrict__ nkey, const int32_t* __restrict__ npred0) {
    const int64_t* skey = NW ? reinterpret_cast<const int64_t*>(nkey) : keycol;                     // what the streaming part loads from (loadc<.., NW>)
    const int64_t* spred0 = NW ? reinterpret_cast<const int64_t*>(npred0) : f.ic[0];
    constexpr int TILE = TPB * ROWS_PER_LOAD * PU;
    constexpr int QCAP = 64 + PU * 128;                               // 63 left over + a whole tile's candidates
    __shared__ int32_t s_row[TPB / WAVE][QCAP];                       // row offsets from qbase (the chunk being produced), as in k_lookup_agg
    __shared__ int64_t s_key[TPB / WAVE][QCAP];
    int32_t* q_row = s_row[threadIdx.x / WAVE];
    int64_t* q_key = s_key[threadIdx.x / WAVE];
    int qn = 0;                                                       // wave-uniform queue length
    int64_t qbase = 0;
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    const uint64_t mask = table_is_direct(tb) ? 0 : tb.hdr->cap_mask;
    const bool direct_bm = tb.bm && tb.bm_shift == 0 && !tb.lin_rb;
    DevProbes none; none.n = 0;
    const uint64_t nomask[SDQH_MAX_PROBE] = {0, 0};
    const int64_t full = nrows / TILE;
    const bool tail_owner = full * TILE < nrows && blockIdx.x == (unsigned)(full % gridDim.x);
    int64_t t0 = (int64_t)blockIdx.x * chunk, tail_r0 = full * TILE;
    int c = 0, phase = t0 < full ? 0 : (tail_owner ? 1 : 2);                // 0: tiles, 1: tail, 2: last drain
    qbase = phase == 0 ? t0 * TILE : tail_r0;
    // Software pipeline (host's choice: a clustered key column): the keys and the first predicate column of the NEXT tile are
    // requested after this tile's bitmap words and before anything waits — a step costs one memory round trip, and the
    // requests stay in flight across the drain.
    constexpr bool HAS_I0 = FC::NI != 0;                                       // the filter may have a first integer predicate
    const bool first_pred = cfg_ni<FC>(f.ni) > 0;
    const bool pipe = pipeline != 0;
    Pair<int64_t> kvn[PU], dn[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) { kvn[u].x = kvn[u].y = 0; dn[u].x = dn[u].y = 0; }
    if (phase == 0 && pipe) {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int64_t rr = t0 * TILE + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
            kvn[u] = loadc<false, NW>(skey, rr, nrows);
            if (HAS_I0 && first_pred) dn[u] = loadc<false, NW>(spred0, rr, nrows);
        }
    }
    auto step = [&](auto PIPE_C) {
        constexpr bool PIPE = decltype(PIPE_C)::value;
        const int64_t tile = t0 + c;
        int64_t r[PU];
        Pair<int64_t> kv[PU];
        bool p[PU][2];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            r[u] = tile * TILE + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
            if constexpr (PIPE) kv[u] = kvn[u]; else kv[u] = loadc<false, NW>(skey, r[u], nrows);
            p[u][0] = p[u][1] = true;
        }
        if constexpr (PIPE || NW) {                                    // the first predicate from this kernel's own (prefetched / narrow) loads
            if (HAS_I0 && first_pred) {
                Pair<int64_t> d[PU];
#pragma unroll
                for (int u = 0; u < PU; ++u) { if constexpr (PIPE) d[u] = dn[u]; else d[u] = loadc<false, NW>(spred0, r[u], nrows); }
#pragma unroll
                for (int u = 0; u < PU; ++u) {
                    p[u][0] = (d[u].x >= f.ilo[0]) & (d[u].x <= f.ihi[0]);
                    p[u][1] = (d[u].y >= f.ilo[0]) & (d[u].y <= f.ihi[0]);
                }
            }
            pass_pairs<PU, FC, false, true>(f, none, r, nrows, nomask, p);
        } else {
            pass_pairs<PU, FC, false>(f, none, r, nrows, nomask, p);
        }
        uint32_t w[PU][2];
        if (direct_bm) {                                               // direct layout: all bitmap words requested before any is tested
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int64_t k0 = kv[u].x, k1 = kv[u].y;
                p[u][0] &= (k0 >= tb.bm_lo) & (k0 <= tb.bm_hi);
                p[u][1] &= (k1 >= tb.bm_lo) & (k1 <= tb.bm_hi);
                w[u][0] = tb.bm[p[u][0] ? (uint64_t)(k0 - tb.bm_lo) >> 5 : 0];
                w[u][1] = tb.bm[p[u][1] ? (uint64_t)(k1 - tb.bm_lo) >> 5 : 0];
            }
        }
        if constexpr (PIPE) {
            // (after the block's last tile: that tile again — an unconditional load keeps the step free of a branch the waits would pile up at)
            const int64_t nt = (c + 1 < chunk && tile + 1 < full) ? tile + 1 : (t0 + (int64_t)gridDim.x * chunk < full ? t0 + (int64_t)gridDim.x * chunk : tile);
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int64_t rr = nt * TILE + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
                kvn[u] = loadc<false, NW>(skey, rr, nrows);
                if (HAS_I0 && first_pred) dn[u] = loadc<false, NW>(spred0, rr, nrows);
            }
        }
        if (direct_bm) {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                p[u][0] = p[u][0] && ((w[u][0] >> ((uint64_t)(kv[u].x - tb.bm_lo) & 31)) & 1u);
                p[u][1] = p[u][1] && ((w[u][1] >> ((uint64_t)(kv[u].y - tb.bm_lo) & 31)) & 1u);
            }
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const uint64_t b0 = __ballot(p[u][0]), b1 = __ballot(p[u][1]);
            if (b0 | b1) {
                const int at = qn + __popcll(b0 & lt) + __popcll(b1 & lt);
                const int32_t off = (int32_t)(r[u] - qbase);
                if (p[u][0]) { q_row[at] = off; q_key[at] = kv[u].x; }
                if (p[u][1]) { const int a1 = at + (p[u][0] ? 1 : 0); q_row[a1] = off + 1; q_key[a1] = kv[u].y; }
                qn += __popcll(b0) + __popcll(b1);
            }
        }
    };
    // one loop, one drain site (see k_lookup_agg): a step produces candidates, then the full waves of the queue are drained
    for (;;) {
        if (phase == 0) {
            if (pipe) step(BoolC<true>{}); else step(BoolC<false>{});
            if (++c == chunk || t0 + c >= full) {                              // next chunk of this block, or the tail, or the end
                c = 0; t0 += (int64_t)gridDim.x * chunk;
                if (t0 >= full) phase = tail_owner ? 1 : 2;
                const int64_t nbase = phase == 0 ? t0 * TILE : tail_r0;
                if (phase != 2) {
                    const int32_t delta = (int32_t)(nbase - qbase);               // < 2^31: one stride of the grid (checked by the host)
                    for (int i = lane; i < qn; i += WAVE) q_row[i] -= delta;
                    qbase = nbase;
                }
            }
        } else if (phase == 1) {                                               // tail: one row per lane through the same queue
            const int64_t r = tail_r0 + threadIdx.x;
            const bool pass = r < nrows && row_passes<FC>(f, none, r, nomask);
            const uint64_t b = __ballot(pass);
            if (b) {
                if (pass) { const int at = qn + __popcll(b & lt); q_row[at] = (int32_t)(r - qbase); q_key[at] = keycol[r]; }
                qn += __popcll(b);
            }
            tail_r0 += TPB;
            if (tail_r0 >= nrows) phase = 2;
        }
        const bool last = phase == 2;
        while (qn >= WAVE || (last && qn > 0)) {
            const int n = qn >= WAVE ? WAVE : qn;
            qn -= n;
            probe_drain<SHAPE>(f, t, tb, mask, qbase, q_row, q_key, qn, n);
        }
        if (last) break;
    }
}

// Membership-only build (sdqh_build_key_set): stream filter + key, OR the survivors' bits.  A lane
// holds two consecutive rows; keys of neighbouring rows usually share a bitmap word (a fact table
// clustered on the key), so equal words of the pair are merged before the atomic.
// NW (the one-comparison instance only): key and both compared columns are read through their 4-byte twins nk / na / nb
template <class FC, bool NW = false>
__global__ __launch_bounds__(TPB) void k_key_set(DevFilter f, DevProbes pr, const int64_t* __restrict__ key, int64_t nrows,
                                                 int64_t lo, int64_t hi, uint32_t* __restrict__ bm, DevFill pre,
                                                 const int32_t* __restrict__ nk, const int32_t* __restrict__ na, const int32_t* __restrict__ nb) {
    if (pre.n) { fill_in_block(pre); __syncthreads(); }                  // a tiny table (grid of one workgroup) clears its own bitmap
    constexpr int PU = 2, TILE = TPB * ROWS_PER_LOAD * PU;
    extern __shared__ __align__(16) uint32_t s_dyn[];                   // f.slds * swidth words per wave (string predicate staging)
    uint32_t* s_str = (cfg_ns<FC>(f.ns) && f.slds) ? s_dyn + (size_t)(threadIdx.x / WAVE) * str_lds_words(f) : nullptr;
    uint64_t cap_masks[SDQH_MAX_PROBE] = {0, 0};
#pragma unroll
    for (int i = 0; i < SDQH_MAX_PROBE; ++i) if (i < cfg_np<FC>(pr.n) && pr.table[i].hdr && !table_is_direct(pr.table[i]) && !pr.table[i].bitmap_only) cap_masks[i] = pr.table[i].hdr->cap_mask;
    // One (word, bits) per lane, merged across the wave before the atomic: with keys in row order the
    // 128 rows of a wave step fall into a handful of bitmap words.  The lanes OR their bits into a
    // 64-word LDS window that starts at the first live lane's word (LDS atomics: a few instructions,
    // where a segmented OR-scan over the wave costs six shuffle rounds), then lane i flushes window word
    // i with one global atomic if it is set.  Words outside the window (keys not in row order) go
    // straight to memory.
    __shared__ uint32_t s_win[TPB / WAVE][WAVE];
    uint32_t* win = s_win[threadIdx.x / WAVE];
    win[lane_id()] = 0u;
    auto wave_set = [&](bool valid, uint32_t w, uint32_t bits) {
        const uint64_t live = __ballot(valid);
        if (!live) return;
        const int lane = lane_id();
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)w, __ffsll((long long)live) - 1);
        const uint32_t d = w - base;                                       // a word below the base wraps to a huge distance
        // the two atomics stay apart: as the arms of one if / else the compiler folds them into ONE flat
        // atomic on a selected address, and a flat access to LDS is not ordered with the ds_read below
        // Window word i is written by other lanes: it is read and reset with atomic accesses between
        // wavefront fences — a plain load would be a data race to the compiler, which then reads the word
        // only in the lanes that issued an OR themselves (and keeps "0" for the rest).
        if (valid && d < (uint32_t)WAVE) atomicOr(&win[d], bits);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t v = __hip_atomic_load(&win[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (v) { atomicOr(&bm[base + lane], v); __hip_atomic_store(&win[lane], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (valid && d >= (uint32_t)WAVE) atomicOr(&bm[w], bits);
    };
    auto set_pair = [&](bool p0, int64_t k0, bool p1, int64_t k1) {       // converged: every lane calls it
        p0 = p0 && k0 >= lo && k0 <= hi; p1 = p1 && k1 >= lo && k1 <= hi;
        const uint64_t o0 = (uint64_t)(k0 - lo), o1 = (uint64_t)(k1 - lo);
        const uint32_t w0 = (uint32_t)(o0 >> 5), w1 = (uint32_t)(o1 >> 5), b0 = 1u << (o0 & 31), b1 = 1u << (o1 & 31);
        const bool both = p0 && p1 && w0 == w1;
        wave_set(p0, w0, both ? (b0 | b1) : b0);
        if (__ballot(p1 && !both)) wave_set(p1 && !both, w1, b1);
    };
    const int64_t full = nrows / TILE;
    for (int64_t tile = blockIdx.x; tile < full; tile += gridDim.x) {
        int64_t r[PU];
        Pair<int64_t> kv[PU];
        bool p[PU][2];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            r[u] = tile * TILE + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
            kv[u] = loadc<false, NW>(NW ? reinterpret_cast<const int64_t*>(nk) : key, r[u], nrows);
            p[u][0] = p[u][1] = true;
        }
        pass_pairs<PU, FC, true, false, false, NW>(f, pr, r, nrows, cap_masks, p, s_str, nullptr, na, nb);
#pragma unroll
        for (int u = 0; u < PU; ++u) set_pair(p[u][0], kv[u].x, p[u][1], kv[u].y);
    }
    if (full * TILE < nrows && blockIdx.x == (unsigned)(full % gridDim.x))
        for (int64_t r0 = full * TILE; r0 < nrows; r0 += TPB) {
            const int64_t r = r0 + threadIdx.x;
            const bool ok = r < nrows && row_passes<FC>(f, pr, r, cap_masks);
            set_pair(ok, ok ? key[r] : lo, false, lo);
        }
}
// The same build for keys in NO row order (Q22: the customers of 15 M orders; Q15: the suppliers of a quarter's lineitems).
// Then nearly every bit is a device-scope atomic of its own, and those execute at the memory side, not in an XCD's L2: 26 G/s,
// 0.58 ms for Q22's 120 MB of keys.  Here a 1024-thread workgroup ORs the bits of ITS rows into a bitmap in LDS (ds_or) — up to
// pass_words words of the key range at a time, the rows streamed once per such part — and stores the words into its own slice;
// k_or_slices folds the slices into the table's bitmap with plain loads and stores.  No global atomic at all.
constexpr int KSL_BT = 1024, KSL_U = 4;
template <class FC>
__global__ __launch_bounds__(KSL_BT) void k_key_set_lds(DevFilter f, DevProbes pr, const int64_t* __restrict__ key, int64_t nrows, int64_t lo, int64_t hi,
                                                        uint32_t* __restrict__ slices, uint64_t nwords, uint32_t pass_words) {
    extern __shared__ __align__(16) uint32_t s_bits[];                  // pass_words words
    uint64_t cap_masks[SDQH_MAX_PROBE] = {0, 0};
#pragma unroll
    for (int i = 0; i < SDQH_MAX_PROBE; ++i) if (i < cfg_np<FC>(pr.n) && pr.table[i].hdr && !table_is_direct(pr.table[i]) && !pr.table[i].bitmap_only) cap_masks[i] = pr.table[i].hdr->cap_mask;
    const int64_t per = (nrows + gridDim.x - 1) / gridDim.x;
    const int64_t r_begin = (int64_t)blockIdx.x * per;
    const int64_t r_end = r_begin + per < nrows ? r_begin + per : nrows;
    uint32_t* __restrict__ mine = slices + (uint64_t)blockIdx.x * nwords;
    for (uint64_t w0 = 0; w0 < nwords; w0 += pass_words) {
        const uint32_t nw = (uint32_t)(nwords - w0 < (uint64_t)pass_words ? nwords - w0 : (uint64_t)pass_words);
        for (uint32_t i = threadIdx.x; i < nw; i += KSL_BT) s_bits[i] = 0u;
        __syncthreads();
        const int64_t plo = lo + (int64_t)(w0 * 32u);
        int64_t phi = plo + (int64_t)nw * 32 - 1; if (phi > hi) phi = hi;
        for (int64_t r0 = r_begin; r0 < r_end; r0 += (int64_t)KSL_BT * KSL_U) {
            int64_t k[KSL_U]; bool ok[KSL_U];
#pragma unroll
            for (int u = 0; u < KSL_U; ++u) { const int64_t r = r0 + (int64_t)u * KSL_BT + threadIdx.x; ok[u] = r < r_end; k[u] = ok[u] ? __builtin_nontemporal_load(key + r) : lo; }
#pragma unroll
            for (int u = 0; u < KSL_U; ++u) {
                const int64_t r = r0 + (int64_t)u * KSL_BT + threadIdx.x;
                if (ok[u] && k[u] >= plo && k[u] <= phi && row_passes<FC>(f, pr, r, cap_masks)) {
                    const uint64_t off = (uint64_t)(k[u] - plo);
                    atomicOr(&s_bits[off >> 5], 1u << (off & 31));
                }
            }
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nw; i += KSL_BT) mine[w0 + i] = s_bits[i];
        __syncthreads();
    }
}
// nchunks > 1 (a short bitmap: too few words to fill the device with one thread per word): a thread folds one chunk of the
// slices of its word and ORs the result into the (cleared) bitmap.
SDQH_KERNEL __launch_bounds__(TPB) void k_or_slices(const uint32_t* __restrict__ slices, int nslices, uint64_t nwords, uint32_t* __restrict__ bm, int nchunks) {
    if (nchunks == 1) {
        for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * TPB) {
            uint32_t v = 0;
#pragma unroll 8
            for (int g = 0; g < nslices; ++g) v |= slices[(uint64_t)g * nwords + i];
            bm[i] = v;
        }
        return;
    }
    const int per = (nslices + nchunks - 1) / nchunks;
    for (uint64_t idx = (uint64_t)blockIdx.x * TPB + threadIdx.x; idx < nwords * (uint64_t)nchunks; idx += (uint64_t)gridDim.x * TPB) {
        const uint64_t i = idx % nwords;
        const int c = (int)(idx / nwords), g1 = (c + 1) * per < nslices ? (c + 1) * per : nslices;
        uint32_t v = 0;
#pragma unroll 8
        for (int g = c * per; g < g1; ++g) v |= slices[(uint64_t)g * nwords + i];
        if (v) atomicOr(&bm[i], v);
    }
}

// population count of a bitmap (sdqh_table_size of a membership-only table)
SDQH_KERNEL __launch_bounds__(TPB) void k_popcount(const uint32_t* __restrict__ bm, uint64_t nwords, unsigned long long* __restrict__ out) {
    unsigned long long n = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * TPB) n += __popc(bm[i]);
    n = (unsigned long long)wave_sum_i64((int64_t)n);
    if (lane_id() == 0 && n) atomicAdd(out, n);
}

// ---- group-by on a row key with a large domain (sdqh_groupby_key) --------------------------------------
// The distinct keys come from k_key_set's bitmap, ranked by k_rank_words; entry i (= rank) is then
// laid out directly: key[i] from the bit position, zeroed accumulators, identity dense_ref, segment
// counts from the device-resident distinct count.  The rows are added by k_probe_agg.
SDQH_KERNEL __launch_bounds__(TPB) void k_gk_layout(DevStage st, DevTable t, int64_t lo, uint64_t nwords) {
    const uint64_t D = t.hdr->distinct;
    const uint64_t tid = (uint64_t)blockIdx.x * TPB + threadIdx.x, nth = (uint64_t)gridDim.x * TPB;
    if (tid == 0) { t.hdr->staged = D; t.hdr->has_dups = 0; }
    for (uint64_t s = tid; s < (uint64_t)st.nseg; s += nth) {
        const uint64_t b = s * (uint64_t)st.seg_rows;
        st.seg_count[s] = (uint32_t)(D > b ? min((uint64_t)st.seg_rows, D - b) : 0);
    }
    for (uint64_t i = tid; i < D; i += nth) {
        t.dense_ref[i] = (uint32_t)i;
        st.shits[i] = 0u;
        if (st.sacc) zero_acc(st, (int64_t)i);
    }
    for (uint64_t w = tid; w < nwords; w += nth) {                      // keys of the set bits of word w, at their ranks
        uint32_t bits = t.bm[w];
        uint64_t at = t.wprefix[w];
        while (bits) { const int bit = __builtin_ctz(bits); bits &= bits - 1u; st.key[at++] = lo + (int64_t)(w * 32 + (uint64_t)bit); }
    }
}
// HAVING (sdqh_table_select_keys): bits of the owner entries with hits >= min_hits and lo <= acc[v] <= hi
SDQH_KERNEL __launch_bounds__(TPB) void k_select_keys(DevTable t, DevStage st, uint32_t min_hits, int v, double lo, double hi,
                                                     int64_t key_lo, uint32_t* __restrict__ out_bm) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const bool dups = t.hdr->has_dups != 0;
    const uint64_t mask = (table_is_direct(t) || !dups) ? 0 : t.hdr->cap_mask;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    for (uint32_t i = lane_id(); i < count; i += WAVE) {
        const int64_t idx = base + i;
        if (st.shits[idx] < min_hits) continue;
        const double x = v < st.acc_stride ? st.sacc[(size_t)idx * st.acc_stride + v] : 0.0;    // a value the entries have no room for is 0
        if (!(x >= lo && x <= hi)) continue;
        if (dups && !stage_row_owns(st, t, idx, mask)) continue;
        const uint64_t off = (uint64_t)(st.key[idx] - key_lo);
        atomicOr(&out_bm[off >> 5], 1u << (off & 31));
    }
}

// K-A with semi-join probes (sdqh_scan_probe_sum): the streaming part of k_probe_agg — first
// predicate and first probe key with 16-byte loads, bitmap test — but survivors add their tuple to
// per-lane registers; workgroup partials are folded by k_sum_partials in a fixed order.
template <int SHAPE, class FC>
__global__ __launch_bounds__(TPB) void k_scan_probe_sum(DevFilter f, DevProbes pr, DevTuple t, int64_t nrows, double* __restrict__ partial) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV;
    constexpr int PU = 2, TILE = TPB * ROWS_PER_LOAD * PU;
    uint64_t cap_masks[SDQH_MAX_PROBE] = {0, 0};
#pragma unroll
    for (int i = 0; i < SDQH_MAX_PROBE; ++i) if (i < cfg_np<FC>(pr.n) && pr.table[i].hdr && !table_is_direct(pr.table[i]) && !pr.table[i].bitmap_only) cap_masks[i] = pr.table[i].hdr->cap_mask;
    double acc[4] = {0, 0, 0, 0};
    int64_t cnt = 0;
    auto add_row = [&](int64_t r) {
        double x[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NOPS; ++j) x[j] = t.op[j][r];
        if (!operand_ranges<NOPS>(f, x)) return;
        tuple_eval<SHAPE>(x, o);
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] += o[k];
        ++cnt;
    };
    const int64_t full = nrows / TILE;
    for (int64_t tile = blockIdx.x; tile < full; tile += gridDim.x) {
        int64_t r[PU];
        bool p[PU][2];
#pragma unroll
        for (int u = 0; u < PU; ++u) { r[u] = tile * TILE + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD; p[u][0] = p[u][1] = true; }
        pass_pairs<PU, FC, true>(f, pr, r, nrows, cap_masks, p);
#pragma unroll
        for (int u = 0; u < PU; ++u) { if (p[u][0]) add_row(r[u]); if (p[u][1]) add_row(r[u] + 1); }
    }
    if (full * TILE < nrows && blockIdx.x == (unsigned)(full % gridDim.x))
        for (int64_t r = full * TILE + threadIdx.x; r < nrows; r += TPB) if (row_passes<FC>(f, pr, r, cap_masks)) add_row(r);
    __shared__ double s_acc[TPB / WAVE][4];
    __shared__ int64_t s_cnt[TPB / WAVE];
    const int w = threadIdx.x / WAVE;
#pragma unroll
    for (int k = 0; k < NV; ++k) { double v = wave_sum(acc[k]); if (lane_id() == 0) s_acc[w][k] = v; }
    { int64_t c = wave_sum_i64(cnt); if (lane_id() == 0) s_cnt[w] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* out = partial + (size_t)blockIdx.x * 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) { double v = 0.0; if (k < NV) for (int i = 0; i < TPB / WAVE; ++i) v += s_acc[i][k]; out[k] = v; }
        int64_t c = 0;
        for (int i = 0; i < TPB / WAVE; ++i) c += s_cnt[i];
        reinterpret_cast<int64_t*>(out)[4] = c;
    }
}

// =================================================================================================
// Multi-join chains (Q5, Q9): lookups whose keys / payloads / operands come from earlier lookups.
// The streaming part of both kernels is the usual one — first predicate and, when it is a plain
// column, the first lookup's key streamed with 16-byte loads, pre-filtered by that table's bitmap —
// and everything that follows a surviving row (the chain of dependent lookups, payload gathers,
// key packing) runs in the converged drain of an LDS candidate queue, one row per lane.
// =================================================================================================
// pack: 1 + the column's position in the loop's row pack (DevLookups::pack), 0 = read the column itself
// pack: COLUMN sources: 1 + the column's position in the loop's row pack (DevLookups::pack), 0 = read the column itself.
//       LOOKUP sources: how the payload is read, resolved by the host (no selection among the tables' pointers on the device):
//       0: col[entry] (col = the table's stage payload array of `field`); 1: col[slot * 4 + 1 + field] (col = packed hash slots);
//       2: col2[(uint32_t)col[slot * 4 + 3]] (packed slots, a payload field beyond the two kept in the slot)
struct DevSource { int32_t kind, lookup, field, pack; const int64_t* col; const int64_t* col2; };
struct DevLookup { DevTable table; int32_t nkey, _pad; DevSource key[2]; };
// Row pack: the columns a loop GATHERS for its (sparse) surviving rows — key parts of later lookups, value
// operands — interleaved row-major, pack[r * pack_k + j], built once per column set (k_interleave) and kept
// resident.  A survivor then touches one or two cache lines instead of one per column: with 3-5 % of the
// rows surviving, the column gathers of Q5 / Q9's final loops pulled 40-60 % of every gathered column's lines
// for 8 useful bytes each.
constexpr int MAX_PACK = 8;
// Coarse key filter of the first lookup, held in LDS: bit j of `coarse` = "some key of the table lies in
// [lo + j * 2^shift, lo + (j + 1) * 2^shift)".  When the probe keys arrive in no order (l_partkey), every row's
// test of the table's exact key bitmap is its own L2 request — 60 M of them made Q9's final loop L2-request-bound
// (88 M L2 reads for 3.2 M useful rows) — while a test in LDS costs nothing; only the rows that pass it go on to
// the exact bitmap.  Built once per table (k_coarsen) and copied into LDS by every workgroup.
struct DevLookups { DevLookup l[SDQH_MAX_LOOKUP]; int32_t n, pack_k; const int64_t* pack; const uint32_t* coarse; int32_t coarse_words, coarse_shift; int32_t pipeline, debug;
                    // RUN WALK (clustered pack): run_lb[v - run_lo] = the first pack row whose key is >= v, for v in [run_lo, run_hi + 1] (sdqh_aux.hip:
                    // cluster_pack_build).  The loop then walks the first table's key BITMAP — every set bit a key, its pack rows a run — and streams nothing
                    const uint32_t* run_lb; int64_t run_lo, run_hi; };
struct PackRow { int64_t v0, v1, v2, v3, v4, v5, v6, v7; };      // named, not an array: a run-time pick must stay a select chain on registers

// all pack_k values of row r with 16-byte loads (pack_k is even: padded by the host), into registers
__device__ __forceinline__ void pack_load(const DevLookups& L, int64_t r, PackRow& pr) {
    using V = long long __attribute__((ext_vector_type(2)));
    const V* p = reinterpret_cast<const V*>(L.pack + r * L.pack_k);
    const V a = p[0];
    V b = {0, 0}, c = {0, 0}, d = {0, 0};
    if (L.pack_k > 2) b = p[1];
    if (L.pack_k > 4) c = p[2];
    if (L.pack_k > 6) d = p[3];
    pr.v0 = a.x; pr.v1 = a.y; pr.v2 = b.x; pr.v3 = b.y; pr.v4 = c.x; pr.v5 = c.y; pr.v6 = d.x; pr.v7 = d.y;
}
__device__ __forceinline__ int64_t pack_pick(const PackRow& pr, int idx) {
    return idx == 0 ? pr.v0 : idx == 1 ? pr.v1 : idx == 2 ? pr.v2 : idx == 3 ? pr.v3 : idx == 4 ? pr.v4 : idx == 5 ? pr.v5 : idx == 6 ? pr.v6 : pr.v7;
}

__device__ __forceinline__ uint32_t pick3(const uint32_t (&e)[SDQH_MAX_LOOKUP], int i) { return i == 0 ? e[0] : (i == 1 ? e[1] : e[2]); }

// PACKED: the row's gathered columns are already in registers (pr); false: pr is never read
template <bool PACKED = false>
__device__ __forceinline__ int64_t source_value(const DevSource& s, const DevLookups& L, int64_t r, const uint32_t (&ent)[SDQH_MAX_LOOKUP], const PackRow& pr) {
    if (s.kind == SDQH_SRC_COLUMN) { if constexpr (PACKED) { if (s.pack) return pack_pick(pr, s.pack - 1); } return s.col[r]; }
    const uint32_t e = pick3(ent, s.lookup);
    // one load at a selected index (no branch between the loads of a drain: they must all be in flight together);
    // only a packed table's third / fourth payload field pays a second, dependent load
    const uint64_t idx = s.pack == 0 ? (uint64_t)e : s.pack == 3 ? (uint64_t)e * 2 + 1 : (uint64_t)e * 4 + (s.pack == 1 ? 1 + (uint64_t)s.field : 3);      // (3: the grouped layout's key / payload pairs)
    int64_t v = s.col[idx];
    if (s.pack == 2) v = s.col2[(uint32_t)v];
    return s.kind == SDQH_SRC_LOOKUP_YEAR ? v / 10000 : v;
}
// 0 = ok; 1 = a part of a composite key is outside [0, 2^32)
__device__ __forceinline__ int pack_parts(int nkey, int64_t p0, int64_t p1, int64_t& key) {
    if (nkey == 1) { key = p0; return 0; }
    if (p0 < 0 || p0 > 0xFFFFFFFFll || p1 < 0 || p1 > 0xFFFFFFFFll) return 1;
    key = (int64_t)(((uint64_t)p0 << 32) | (uint64_t)p1);
    return 0;
}
// run every lookup for row r: 1 = all hit (ent filled with stage rows), 0 = a miss, -1 = bad key part
// (fully unrolled: `ent` is indexed statically and stays in registers)
// skip0: the caller has tested the first lookup's table already and that table is a key set (nothing to fetch from it)
template <bool PACKED = false>
__device__ __forceinline__ int run_lookups(const DevLookups& L, int64_t r, uint32_t (&ent)[SDQH_MAX_LOOKUP], const PackRow& pr, bool skip0 = false) {
#pragma unroll
    for (int l = 0; l < SDQH_MAX_LOOKUP; ++l) {
        if (l < L.n && !(l == 0 && skip0)) {
            const DevLookup& lk = L.l[l];
            const int64_t p0 = source_value<PACKED>(lk.key[0], L, r, ent, pr);
            const int64_t p1 = lk.nkey == 2 ? source_value<PACKED>(lk.key[1], L, r, ent, pr) : 0;
            int64_t key;
            if (pack_parts(lk.nkey, p0, p1, key)) return -1;
            const uint64_t mask = (table_is_direct(lk.table) || lk.table.bitmap_only) ? 0 : lk.table.hdr->cap_mask;
            const int64_t pos = table_find(lk.table, key, mask);
            if (pos < 0) return 0;
            ent[l] = lk.table.bitmap_only ? 0u : (lk.table.slots && !table_is_direct(lk.table) ? (uint32_t)pos : table_ref(lk.table, pos));
        }
    }
    return 1;
}
// cheap test on the first lookup alone, usable before queueing: false = the row cannot survive
__device__ __forceinline__ bool coarse_may_hit(const DevLookups& L, const uint32_t* s_coarse, int64_t part0) {
    const DevTable& t = L.l[0].table;
    if (part0 < t.bm_lo || part0 > t.bm_hi) return false;
    const uint64_t j = (uint64_t)(part0 - t.bm_lo) >> L.coarse_shift;
    return (s_coarse[j >> 5] >> (j & 31)) & 1u;
}
// coarse[w] bit b = any fine bit in [(32 w + b) << shift, (32 w + b + 1) << shift)
SDQH_KERNEL __launch_bounds__(TPB) void k_coarsen(const uint32_t* __restrict__ bm, uint64_t nbits, int shift, uint32_t* __restrict__ coarse, int cwords, unsigned long long* __restrict__ set_bits) {
    unsigned long long mine = 0;
    for (int w = blockIdx.x * TPB + threadIdx.x; w < cwords; w += gridDim.x * TPB) {
        uint32_t out = 0;
        for (int b = 0; b < 32; ++b) {
            const uint64_t f0 = ((uint64_t)w * 32 + (uint64_t)b) << shift, f1 = f0 + (1ull << shift);
            bool any = false;
            for (uint64_t f = f0; f < f1 && f < nbits && !any; ) {
                const uint32_t word = bm[f >> 5];
                const uint32_t lo = (uint32_t)(f & 31), span = (uint32_t)min((uint64_t)(32 - lo), f1 - f);
                const uint32_t mask = (span >= 32 ? 0xFFFFFFFFu : ((1u << span) - 1u)) << lo;
                any = (word & mask) != 0u;
                f += span;
            }
            out |= any ? (1u << b) : 0u;
        }
        coarse[w] = out;
        mine += (unsigned long long)__popc(out);
    }
    mine = (unsigned long long)wave_sum_i64((long long)mine);
    if (set_bits && lane_id() == 0 && mine) atomicAdd(set_bits, mine);
}
__device__ __forceinline__ bool first_lookup_may_hit(const DevLookups& L, int64_t part0) {
    const DevTable& t = L.l[0].table;
    if (!t.bm) return true;
    if (L.l[0].nkey == 2 && t.bm_shift == 0) return true;
    if (part0 < t.bm_lo || part0 > t.bm_hi) return false;
    const uint64_t off = (uint64_t)(part0 - t.bm_lo);
    return (t.bm[off >> 5] >> (off & 31)) & 1u;
}

// The same test in two halves, so that a streaming loop can put other loads between them: the bitmap word of the key
// (requested unconditionally, at word 0 when the key is out of range — no divergent branch around the load), and the
// test of its bit.  `none`: the table has no usable bitmap (every row goes on to the full lookup).
__device__ __forceinline__ bool first_lookup_none(const DevLookups& L) {
    const DevTable& t = L.l[0].table;
    return !t.bm || (L.l[0].nkey == 2 && t.bm_shift == 0);
}
__device__ __forceinline__ uint32_t first_lookup_word(const DevLookups& L, int64_t part0) {
    const DevTable& t = L.l[0].table;
    const bool in = part0 >= t.bm_lo && part0 <= t.bm_hi;
    const uint64_t off = in ? (uint64_t)(part0 - t.bm_lo) : 0;
    return t.bm[off >> 5];
}
__device__ __forceinline__ bool first_lookup_bit(const DevLookups& L, int64_t part0, uint32_t word) {
    const DevTable& t = L.l[0].table;
    const bool in = part0 >= t.bm_lo && part0 <= t.bm_hi;
    return in && ((word >> ((uint64_t)(part0 - t.bm_lo) & 31)) & 1u);
}

constexpr int BUILD_LB = 4;                                           // 128-row batches per step of k_build_lookup
constexpr int LOOKUP_PU = 2;                                          // row pairs per lane in flight in k_lookup_agg's streaming part.  Measured the same on Q9 (0.78 ms): 4 pairs;
                                                                      // streaming every gathered column through the LDS queue instead of gathering it (0.87 ms); key, row reference
                                                                      // and payloads of a hash entry in one 32-byte slot, i.e. one line per probe instead of four (0.79 ms); every
                                                                      // plain-column key part and operand of a candidate requested before the lookup chain (0.78 ms).  DESIGN.md §7
constexpr int LBQ_CAP = 64 + BUILD_LB * 128;                             // k_build_lookup's queue: 63 left over + a whole step's candidates

struct DevBuildSpec {                                                 // what a surviving row contributes to the build
    int32_t nkey, npay;
    DevSource key[2];
    DevSource pay[SDQH_MAX_PAYLOAD];
};

// Generalised staging: one wave per row segment (row order is kept: the queue is drained from the
// front), survivors compacted into the segment's stage slice exactly as k_stage does.
// A tiny table's direct index made by its build kernel's own workgroup, behind the staging (what k_index_small did in a launch of its
// own: 7 - 10 us of dependent round trips on an idle chip for a table of 25 rows): on = 1, `t` with wprefix / dense_ref set, span or null
struct DevIndexInline { DevTable t; uint32_t* wprefix; uint32_t* span; uint64_t nwords; int32_t on, _pad; };

template <class FC, bool NW>
__device__ __forceinline__ void build_lookup_segment(const DevFilter& f, const DevLookups& L, const DevBuildSpec& spec, const DevStage& st, int64_t nrows, int* __restrict__ flags,
                                                     const int32_t* __restrict__ nkey0, const int32_t* __restrict__ npred0, int32_t* q_row, const int seg) {
    const int64_t* skey0 = NW ? reinterpret_cast<const int64_t*>(nkey0) : L.l[0].key[0].col;
    const int64_t begin = (int64_t)seg * st.seg_rows;
    int64_t end = begin + st.seg_rows; if (end > nrows) end = nrows;
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    constexpr int64_t BATCH_ROWS = WAVE * ROWS_PER_LOAD;
    const bool eager0 = L.n > 0 && L.l[0].key[0].kind == SDQH_SRC_COLUMN;
    DevProbes none; none.n = 0;
    const uint64_t nomask[SDQH_MAX_PROBE] = {0, 0};
    int64_t out = begin;
    int qn = 0;
    bool bad = false;
    auto drain = [&](int first, int count) {                           // rows q_row[first .. first + count) one per lane, in row order
        bool keep = false; int64_t key = 0; int64_t pay[MAX_STAGE_COLS] = {0, 0, 0, 0, 0};
        if (lane < count) {
            const int64_t r = begin + (int64_t)q_row[first + lane];
            uint32_t ent[SDQH_MAX_LOOKUP] = {0, 0, 0};
            const PackRow nopack{};
            const int h = run_lookups(L, r, ent, nopack);
            if (h < 0) bad = true;
            if (h > 0) {
                const int64_t p0 = source_value(spec.key[0], L, r, ent, nopack);
                const int64_t p1 = spec.nkey == 2 ? source_value(spec.key[1], L, r, ent, nopack) : 0;
                if (pack_parts(spec.nkey, p0, p1, key)) bad = true;
                else {
                    keep = true;
#pragma unroll
                    for (int q = 0; q < SDQH_MAX_PAYLOAD; ++q) if (q < spec.npay) pay[q] = source_value(spec.pay[q], L, r, ent, nopack);
                }
            }
        }
        const uint64_t b = __ballot(keep);
        bool opens_run = true;
        if (st.grp_first) {                                               // the kept lane below mine holds the entry stored just before mine
            const uint64_t lower = b & lt;
            const int prev = lower ? 63 - __builtin_clzll(lower) : lane;
            const int64_t prev_key = __shfl(key, prev, WAVE);
            opens_run = !lower || ((uint64_t)prev_key >> 32) != ((uint64_t)key >> 32);
        }
        if (keep) stage_store<-1>(st, out + __popcll(b & lt), key, pay, opens_run);
        out += __popcll(b);
    };
    // queue entries: 32-bit offsets from the segment's first row (a segment is far shorter than 2^31 rows: the host checks)
    auto enqueue = [&](int64_t r, bool p0, bool p1) {
        const uint64_t b0 = __ballot(p0), b1 = __ballot(p1);
        if (b0 | b1) {
            const int at = qn + __popcll(b0 & lt) + __popcll(b1 & lt);
            const int32_t off = (int32_t)(r - begin);
            if (p0) q_row[at] = off;
            if (p1) q_row[at + (p0 ? 1 : 0)] = off + 1;
            qn += __popcll(b0) + __popcll(b1);
        }
    };
    // One loop, one drain site (the drain — three unrolled lookups over every table layout — is most of this kernel's
    // code; inlined at each enqueue it made 104 KB of it, past the instruction cache two CUs share).  Every step
    // produces candidates, then drains the FRONT full waves of the queue (row order) and moves what is left down.
    // A full step takes BUILD_LB batches of 128 rows: the first predicate and the first lookup's key of all of them
    // are loaded, and their bitmap words requested, before any is consumed (one dependent chain per step instead of
    // one per batch: the orders build of Q5 is latency-bound, 19 -> 5 steps per wave).
    for (int64_t b = begin;;) {
        const bool last = b >= end;
        if (last) {
        } else if (b + BATCH_ROWS * BUILD_LB <= end) {
            int64_t rr[BUILD_LB];
            bool p[BUILD_LB][2];
            Pair<int64_t> k0[BUILD_LB];
#pragma unroll
            for (int j = 0; j < BUILD_LB; ++j) {
                rr[j] = b + (int64_t)j * BATCH_ROWS + (int64_t)lane * ROWS_PER_LOAD;
                p[j][0] = p[j][1] = true;
                if (eager0) k0[j] = loadc<false, NW>(skey0, rr[j], nrows);
            }
            pass_pairs<BUILD_LB, FC, false, false, false, NW>(f, none, rr, nrows, nomask, p, nullptr, nullptr, nullptr, npred0);
            if (eager0) {
#pragma unroll
                for (int j = 0; j < BUILD_LB; ++j) { p[j][0] = p[j][0] && first_lookup_may_hit(L, k0[j].x); p[j][1] = p[j][1] && first_lookup_may_hit(L, k0[j].y); }
            }
#pragma unroll
            for (int j = 0; j < BUILD_LB; ++j) enqueue(rr[j], p[j][0], p[j][1]);
            b += BATCH_ROWS * BUILD_LB;
        } else {                                                         // short segments (small tables) and tails: one batch at a time
            const int64_t r = b + (int64_t)lane * ROWS_PER_LOAD;
            bool p1b[1][2];
            if (b + BATCH_ROWS <= end) {
                int64_t r1[1] = {r};
                p1b[0][0] = p1b[0][1] = true;
                Pair<int64_t> k1 = {0, 0};
                if (eager0) k1 = load2<false>(L.l[0].key[0].col, r, nrows);
                pass_pairs<1, FC, false>(f, none, r1, nrows, nomask, p1b);
                if (eager0) { p1b[0][0] = p1b[0][0] && first_lookup_may_hit(L, k1.x); p1b[0][1] = p1b[0][1] && first_lookup_may_hit(L, k1.y); }
            } else {
                p1b[0][0] = r < end; p1b[0][1] = r + 1 < end;
                if (p1b[0][0]) p1b[0][0] = row_passes<FC>(f, none, r, nomask);
                if (p1b[0][1]) p1b[0][1] = row_passes<FC>(f, none, r + 1, nomask);
            }
            enqueue(r, p1b[0][0], p1b[0][1]);
            b += BATCH_ROWS;
        }
        int head = 0;
        while (qn - head >= WAVE || (last && qn > head)) {
            const int n = qn - head >= WAVE ? WAVE : qn - head;
            drain(head, n);
            head += n;
        }
        if (last) break;
        if (head) {
            const int left = qn - head;                                   // < 64
            int32_t keepv = 0;
            if (lane < left) keepv = q_row[head + lane];
            if (lane < left) q_row[lane] = keepv;
            qn = left;
        }
    }
    if (lane == 0) {
        st.seg_count[seg] = (uint32_t)(out - begin);
        stage_end_segment(st, seg, begin, out - begin);
        // grouped layout: equal keys are not told apart while staging (a lookup meets the lowest build row first, which is the answer);
        // whoever needs an entry's OWNER (K-F, sizes) asks the index row by row, as after a build that met duplicates
        if (st.grp_first && seg == 0) st.hdr->has_dups = 1u;
    }
    if (__ballot(bad) && lane == 0) atomicOr(flags, 2);
}

template <class FC, bool NW = false>      // NW: the first lookup's streamed key and the first integer predicate through their narrow twins (full steps only)
__global__ __launch_bounds__(TPB) void k_build_lookup(DevFilter f, DevLookups L, DevBuildSpec spec, DevStage st, int64_t nrows, int* __restrict__ flags,
                                                      const int32_t* __restrict__ nkey0, const int32_t* __restrict__ npred0, DevFill pre, DevIndexInline ix) {
    if (pre.n) { fill_in_block(pre); __syncthreads(); }                  // (only ever with a grid of one workgroup)
    __shared__ int32_t s_row[TPB / WAVE][LBQ_CAP];
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg < st.nseg) build_lookup_segment<FC, NW>(f, L, spec, st, nrows, flags, nkey0, npred0, s_row[threadIdx.x / WAVE], seg);
    if (ix.on) {                                                         // (a grid of one workgroup: every wave gets here)
        __threadfence();
        __syncthreads();
        rank_words_body(st.bm, ix.nwords, ix.wprefix, st.seg_count, st.nseg, st.hdr);
        __threadfence();
        __syncthreads();
        insert_direct_body(st, ix.t, ix.span);
    }
}

// row pack builder: out[r * k + j] = col[j][r] (j >= ncols: padding)
struct DevPackCols { const int64_t* col[MAX_PACK]; int32_t ncols, k; };
SDQH_KERNEL __launch_bounds__(TPB) void k_interleave(DevPackCols c, int64_t nrows, int64_t* __restrict__ out) {
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * TPB) {
        int64_t v[MAX_PACK];
#pragma unroll
        for (int j = 0; j < MAX_PACK; ++j) v[j] = j < c.ncols ? c.col[j][r] : 0;
#pragma unroll
        for (int j = 0; j < MAX_PACK; ++j) if (j < c.k) out[r * c.k + j] = v[j];
    }
}

// ---- lookups -> group-by over a small domain ------------------------------------------------------------
constexpr int LG_SLOTS = 2 * SDQH_MAX_LOOKUP_GROUPS;                  // LDS / global group table slots (load factor <= 1/2)

struct DevAggSpec {
    int32_t nkeys, shape;
    DevSource key[SDQH_MAX_GROUPKEYS];
    DevSource op[4];
};

// claim-or-find with a hashed start; -1 when the table is full
__device__ __forceinline__ int group_slot(unsigned long long* keys, unsigned long long key, bool global) {
    int h = (int)(mix64(key) & (LG_SLOTS - 1));
#pragma unroll 1
    for (int i = 0; i < LG_SLOTS; ++i) {
        unsigned long long cur = global ? __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : keys[h];
        if (cur == EMPTY_GROUP) cur = atomicCAS(&keys[h], (unsigned long long)EMPTY_GROUP, key);
        if (cur == EMPTY_GROUP || cur == key) return h;
        h = (h + 1) & (LG_SLOTS - 1);
    }
    return -1;
}

// BT: threads per workgroup.  256 everywhere, except with the coarse key filter: 1024 (16 waves share ONE copy of the filter in LDS —
// a copy per 256 threads cost more occupancy than the filter saved), and then PU = 4 row pairs per lane in flight to make up for
// the fewer waves per CU.
// NW: the first lookup's streamed key column is read through its narrow twin (nkey0: int32); rows in the queue read the column itself
// WALK: the instance of the run walk (L.run_lb set by the host): the streaming part is not compiled — the registers it holds (four row pairs of keys,
// words and flags per lane) would set the kernel's occupancy for a part that never runs
template <int SHAPE, class FC, int BT = TPB, int PUV = LOOKUP_PU, bool NW = false, bool WALK = false>
__global__ __launch_bounds__(BT) void k_lookup_agg(DevFilter f, DevLookups L, DevAggSpec spec, int64_t nrows,
                                                    unsigned long long* __restrict__ gkeys, double* __restrict__ pacc,
                                                    int64_t* __restrict__ pcnt, int* __restrict__ flags, int chunk, const int32_t* __restrict__ nkey0) {
    const int64_t* skey0 = NW ? reinterpret_cast<const int64_t*>(nkey0) : L.l[0].key[0].col;
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV;
    constexpr int PU = PUV, TILE = BT * ROWS_PER_LOAD * PU, SUBR = BT * ROWS_PER_LOAD, QCAP = 64 + PU * 128;
    __shared__ unsigned long long s_keys[LG_SLOTS];
    constexpr int NVS = NV > 0 ? NV : 1;
    __shared__ double s_acc[LG_SLOTS][NVS];
    __shared__ unsigned long long s_cnt[LG_SLOTS];
    __shared__ int32_t s_row[BT / WAVE][QCAP];
    __shared__ int s_map[LG_SLOTS];
    __shared__ int s_flags[1];
    for (int i = threadIdx.x; i < LG_SLOTS; i += BT) { s_keys[i] = EMPTY_GROUP; s_cnt[i] = 0; for (int k = 0; k < NVS; ++k) s_acc[i][k] = 0.0; s_map[i] = -1; }
    // (this workgroup's column of the partial counts is zeroed now, under the streaming; the epilogue stores the slots it has rows for
    //  and nothing else — sdqh_xkernels.hpp XGroup::init; thread i zeroes and later writes slot i)
    for (int i = threadIdx.x; i < LG_SLOTS; i += BT) pcnt[(size_t)i * gridDim.x + blockIdx.x] = 0;
    if (threadIdx.x == 0) s_flags[0] = 0;
    __syncthreads();
    constexpr bool COARSE = BT != TPB;                                   // the coarse key filter exists in the wide-workgroup instance only (compile-time: a run-time
                                                                         // test around the word loads of the pipelined step broke its pipelining: Q5 0.118 -> 0.144 ms)
    extern __shared__ __align__(16) uint32_t s_coarse[];                 // L.coarse_words words
    if constexpr (COARSE) { for (int i = threadIdx.x; i < L.coarse_words; i += BT) s_coarse[i] = L.coarse[i]; __syncthreads(); }
    int32_t* q_row = s_row[threadIdx.x / WAVE];
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    const bool eager0 = L.n > 0 && L.l[0].key[0].kind == SDQH_SRC_COLUMN;
    DevProbes none; none.n = 0;
    const uint64_t nomask[SDQH_MAX_PROBE] = {0, 0};
    int qn = 0;
    // the streaming part's test of the first lookup is exact for a key set over a plain one-part key: its candidates skip that lookup in the drain
    const bool skip0 = eager0 && !first_lookup_none(L) && L.l[0].table.bitmap_only && L.l[0].nkey == 1 && L.l[0].table.bm_shift == 0 && L.l[0].table.lin_rb == 0 && !(L.debug & 2);
    int64_t qbase = 0;
    auto drain = [&](int first, int count) {
        if (lane >= count) return;
        if (L.debug & 1) return;
        const int64_t r = qbase + (int64_t)q_row[first + lane];
        uint32_t ent[SDQH_MAX_LOOKUP] = {0, 0, 0};
        PackRow prow{};
        if (L.pack) pack_load(L, r, prow);                            // every gathered column of the row from its pack: one or two lines
        double x[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};
        if (L.debug & 8) { if (prow.v0 == -12345 && prow.v5 == -777) atomicOr(&s_flags[0], 2); return; }
        const int h = run_lookups<true>(L, r, ent, prow, skip0);      // <true>: a source with pack != 0 reads the registers, any other its column
        if (h < 0) atomicOr(&s_flags[0], 2);
        if (h <= 0) return;
        if (L.debug & 4) { if (ent[0] + ent[1] + ent[2] == 0xFFFFFFF0u) atomicOr(&s_flags[0], 2); return; }
        const int64_t k0 = source_value<true>(spec.key[0], L, r, ent, prow);
        const int64_t k1 = spec.nkeys == 2 ? source_value<true>(spec.key[1], L, r, ent, prow) : 0;
#pragma unroll
        for (int j = 0; j < NOPS; ++j) x[j] = __longlong_as_double(source_value<true>(spec.op[j], L, r, ent, prow));
        if (k0 < 0 || k0 > 0xFFFFFFFEll || k1 < 0 || k1 > 0xFFFFFFFEll) { atomicOr(&s_flags[0], 2); return; }
        if (!operand_ranges<NOPS>(f, x)) return;
        tuple_eval<SHAPE>(x, o);
        const int slot = group_slot(s_keys, (unsigned long long)k0 | ((unsigned long long)k1 << 32), false);
        if (slot < 0) { atomicOr(&s_flags[0], 1); return; }
#pragma unroll
        for (int k = 0; k < NV; ++k) atomicAdd(&s_acc[slot][k], o[k]);
        atomicAdd(&s_cnt[slot], 1ull);
    };
    // One loop, one drain site: every step PRODUCES candidates (a whole tile of the block's current chunk, then — in the
    // block that owns it — one workgroup-width of the tail) and then drains the full waves of the queue; the last step
    // drains what is left.  The drain is the bulk of this kernel's code (three unrolled lookups over every table layout):
    // inlined at each of the four places that used to call it, the kernel outgrew the instruction cache two CUs share
    // (93 KB of code; Q5's final loop 0.222 -> 0.243 ms when the packed-slot paths were added).  Queue entries are
    // 32-bit offsets from `qbase` (the first row of the chunk being produced; what is left over from the previous chunk
    // is rebased), so 63 + PU * 128 of them cost less LDS than the 191 64-bit rows did.
    const int64_t full = nrows / TILE;
    const bool tail_owner = full * TILE < nrows && blockIdx.x == (unsigned)(full % gridDim.x);
    int64_t t0 = (int64_t)blockIdx.x * chunk, tail_r0 = full * TILE;
    int c = 0, phase = t0 < full ? 0 : (tail_owner ? 1 : 2);                // 0: tiles, 1: tail, 2: last drain, 3: rows from the list
    qbase = phase == 0 ? t0 * TILE : tail_r0;
    // RUN WALK (L.run_lb): a wave takes RW_WORDS words of the first table's bitmap per step and compacts their set bits — keys — into its list
    // in LDS; lanes take a key each (the two neighbouring entries of run_lb that bound its run: ONE round trip for up to 64 keys), and the
    // runs are queued ONE AFTER THE OTHER, a lane per row, 64 rows at most per turn of the loop: a drain's 64 rows are two or three whole
    // runs — neighbouring pack rows, two or three entries of the first table — as a list of the rows cut into 64s would give it.  No
    // launch in front, no list in memory, no atomics.  Measured on Q9's 3.2 M rows at SF=10 (0.300 ms with the ordered keys streamed):
    // a list written by a launch of its own 0.031-0.061 + 0.209 ms (one claim per wave step on ONE address, served ~9 ns apart); lanes
    // queueing a few rows of their own runs each turn — a drain's rows from 8-16 runs — 0.277-0.289; EVERY step claimed with a returning
    // atomic 0.311 (4096 first claims at the kernel's start, served one after the other); requesting a
    // step's pack lines and the later lookups' first words ahead 0.209 -> 0.309 ms (loads return in order, and what the early requests
    // bring is evicted before it is used).
    // A wave's FIRST step is dealt out (its number in the grid); every later one is claimed in order with a returning atomic on the word behind
    // the flags (zero when the kernel starts, reset by the merge): a step's keys are a handful, Poisson-distributed, and with steps dealt out
    // the waves that drew the most rows finished a third after the average — the list's even cut was worth 0.05 ms; claims spread over the
    // kernel's time do not queue up the way 4096 first claims at its start did.
    constexpr int RW_WORDS = 8;
    __shared__ uint32_t s_rw[BT / WAVE][RW_WORDS * 32];
    uint64_t rw_next = 0, rw_nwords = 0; uint32_t rw_st = 0, rw_len = 0, rw_off = 0; int rw_have = 0, rw_taken = 0, rw_nk = 0, rw_j = 0;
    bool rw_first = true;
    if (L.run_lb) {
        phase = 3; qbase = 0;
        rw_nwords = ((uint64_t)(L.l[0].table.bm_hi - L.l[0].table.bm_lo) + 32) >> 5;
        rw_next = ((uint64_t)blockIdx.x * (BT / WAVE) + threadIdx.x / WAVE) * RW_WORDS;
    }
    // Software pipeline of the streaming part: the first lookup's keys of the NEXT tile are requested after this tile's
    // bitmap words and before anything waits, so a step costs one memory round trip (keys of i+1, words and filter
    // columns of i in flight together), not the two dependent ones (keys, then words) it would otherwise.
    // Chosen by the host (L.pipeline) where the key column is clustered (Q5's l_orderkey: -23 %); with keys in no order
    // (Q9's l_partkey) every bitmap test is its own L2 request, the loop is bound by those, and the deeper queue of
    // requests costs 5-8 %.
    const bool pipe = L.pipeline && eager0 && !first_lookup_none(L);
    Pair<int64_t> k0n[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) { k0n[u].x = 0; k0n[u].y = 0; }
    if constexpr (!WALK) if (phase == 0 && pipe) {
#pragma unroll
        for (int u = 0; u < PU; ++u) k0n[u] = loadc<false, NW>(skey0, t0 * TILE + (int64_t)u * SUBR + (int64_t)threadIdx.x * ROWS_PER_LOAD, nrows);
    }
    // one streaming step over tile t0 + c: candidates into the queue.  PIPE: keys of this tile were requested a step ago
    auto step = [&](auto PIPE_C) {
        constexpr bool PIPE = decltype(PIPE_C)::value;
        const int64_t tile = t0 + c;
        int64_t r[PU];
        Pair<int64_t> k0[PU];
        uint32_t w[PU][2];
        bool p[PU][2];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            r[u] = tile * TILE + (int64_t)u * SUBR + (int64_t)threadIdx.x * ROWS_PER_LOAD;
            if constexpr (PIPE) k0[u] = k0n[u];
            else if (eager0) k0[u] = loadc<false, NW>(skey0, r[u], nrows);
            p[u][0] = p[u][1] = true;
            w[u][0] = w[u][1] = 0;
        }
        if constexpr (PIPE) {
            if constexpr (!COARSE) {                                       // (with the coarse filter only the rows that pass it ask for their exact word)
#pragma unroll
                for (int u = 0; u < PU; ++u) { w[u][0] = first_lookup_word(L, k0[u].x); w[u][1] = first_lookup_word(L, k0[u].y); }
            }
            // (after the block's last tile: that tile again — an unconditional load keeps the step free of a branch the waits would pile up at)
            const int64_t nt = (c + 1 < chunk && tile + 1 < full) ? tile + 1 : (t0 + (int64_t)gridDim.x * chunk < full ? t0 + (int64_t)gridDim.x * chunk : tile);
#pragma unroll
            for (int u = 0; u < PU; ++u) k0n[u] = loadc<false, NW>(skey0, nt * TILE + (int64_t)u * SUBR + (int64_t)threadIdx.x * ROWS_PER_LOAD, nrows);
        }
        pass_pairs<PU, FC, false>(f, none, r, nrows, nomask, p);
        if (eager0) {
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if constexpr (COARSE) { p[u][0] = p[u][0] && coarse_may_hit(L, s_coarse, k0[u].x); p[u][1] = p[u][1] && coarse_may_hit(L, s_coarse, k0[u].y); }
                if constexpr (PIPE && !COARSE) { p[u][0] = p[u][0] && first_lookup_bit(L, k0[u].x, w[u][0]); p[u][1] = p[u][1] && first_lookup_bit(L, k0[u].y, w[u][1]); }
                else if (L.debug & 2) { p[u][0] = p[u][0] && k0[u].x == -12345; p[u][1] = p[u][1] && k0[u].y == -12345; }
                else { p[u][0] = p[u][0] && first_lookup_may_hit(L, k0[u].x); p[u][1] = p[u][1] && first_lookup_may_hit(L, k0[u].y); }
            }
        }
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const uint64_t b0 = __ballot(p[u][0]), b1 = __ballot(p[u][1]);
            if (b0 | b1) {
                const int at = qn + __popcll(b0 & lt) + __popcll(b1 & lt);
                const int32_t off = (int32_t)(r[u] - qbase);
                if (p[u][0]) q_row[at] = off;
                if (p[u][1]) q_row[at + (p[u][0] ? 1 : 0)] = off + 1;
                qn += __popcll(b0) + __popcll(b1);
            }
        }
    };
    for (;;) {
        if (phase == 3) {
            if (rw_j >= rw_nk) {                                                    // every run in hand is queued
                uint32_t* keys = s_rw[threadIdx.x / WAVE];
                if (rw_taken >= rw_have) {                                            // ... and every key of the list taken: the wave's next words, or the end
                    if (!rw_first) {
                        unsigned int step_i = 0;
                        if (lane == 0) step_i = atomicAdd(reinterpret_cast<unsigned int*>(flags + 1), 1u);
                        rw_next = ((uint64_t)gridDim.x * (BT / WAVE) + (uint64_t)(unsigned int)__shfl((int)step_i, 0, WAVE)) * RW_WORDS;
                    }
                    rw_first = false;
                    if (rw_next >= rw_nwords) phase = 2;
                    else {
                        const uint64_t w = rw_next + (uint64_t)lane;
                        uint32_t word = (lane < RW_WORDS && w < rw_nwords) ? L.l[0].table.bm[w] : 0u;
                        const int cnt = __popc(word);
                        int incl = cnt;
#pragma unroll
                        for (int off = 1; off < RW_WORDS; off <<= 1) { const int o = __shfl_up(incl, off, WAVE); if (lane >= off) incl += o; }      // (only the first RW_WORDS lanes hold words)
                        rw_have = __shfl(incl, RW_WORDS - 1, WAVE);
                        rw_taken = 0;
                        int at = incl - cnt;
                        for (; word; word &= word - 1u) keys[at++] = (uint32_t)(w * 32) + (uint32_t)(__ffs((int)word) - 1);
                    }
                }
                rw_nk = 0; rw_j = 0; rw_off = 0;
                if (phase == 3 && rw_taken < rw_have) {                               // a key per lane: its run
                    const int i = rw_taken + lane;
                    rw_st = 0; rw_len = 0;
                    if (i < rw_have) {
                        const int64_t key = L.l[0].table.bm_lo + (int64_t)keys[i];
                        if (key >= L.run_lo && key <= L.run_hi) { rw_st = L.run_lb[key - L.run_lo]; rw_len = L.run_lb[key - L.run_lo + 1] - rw_st; }
                    }
                    rw_nk = rw_have - rw_taken < WAVE ? rw_have - rw_taken : WAVE;
                    rw_taken += WAVE;
                }
            }
            if (phase == 3 && rw_j < rw_nk) {                                        // up to 64 rows of run rw_j
                const uint32_t st = (uint32_t)__shfl((int)rw_st, rw_j, WAVE), len = (uint32_t)__shfl((int)rw_len, rw_j, WAVE);
                const uint32_t chunk = len - rw_off < (uint32_t)WAVE ? len - rw_off : (uint32_t)WAVE;
                if ((uint32_t)lane < chunk) q_row[qn + lane] = (int32_t)(st + rw_off + (uint32_t)lane);
                qn += (int)chunk;
                rw_off += chunk;
                if (rw_off >= len) { ++rw_j; rw_off = 0; }
            }
        } else if constexpr (!WALK) { if (phase == 0) {
            if (pipe) step(BoolC<true>{}); else step(BoolC<false>{});
            if (++c == chunk || t0 + c >= full) {                              // next chunk of this block, or the tail, or the end
                c = 0; t0 += (int64_t)gridDim.x * chunk;
                if (t0 >= full) phase = tail_owner ? 1 : 2;
                const int64_t nbase = phase == 0 ? t0 * TILE : tail_r0;
                if (phase != 2) {
                    const int32_t delta = (int32_t)(nbase - qbase);               // < 2^31: one stride of the grid (checked by the host)
                    for (int i = lane; i < qn; i += WAVE) q_row[i] -= delta;
                    qbase = nbase;
                }
            }
        } else if (phase == 1) {                                               // tail: one row per lane through the same queue
            const int64_t r = tail_r0 + threadIdx.x;
            bool pass = r < nrows && row_passes<FC>(f, none, r, nomask);
            if (pass && eager0) pass = first_lookup_may_hit(L, NW ? (int64_t)nkey0[r] : L.l[0].key[0].col[r]);      // (NW: the twin may stand in ANOTHER order than the column — a clustered pack)
            const uint64_t b = __ballot(pass);
            if (b) {
                if (pass) q_row[qn + __popcll(b & lt)] = (int32_t)(r - qbase);
                qn += __popcll(b);
            }
            tail_r0 += BT;
            if (tail_r0 >= nrows) phase = 2;
        } } else { phase = 2; }                                                // (WALK: launched with L.run_lb only)
        const bool last = phase == 2;
        while (qn >= WAVE || (last && qn > 0)) {
            const int n = qn >= WAVE ? WAVE : qn;
            qn -= n;
            drain(qn, n);
        }
        if (last) break;
    }
    __syncthreads();
    // publish this workgroup's groups at GLOBAL slots (slot-major partials, as k_groupby_*)
    for (int i = threadIdx.x; i < LG_SLOTS; i += BT) {
        if (s_keys[i] != EMPTY_GROUP && s_cnt[i] > 0) {
            const int gs = group_slot(gkeys, s_keys[i], true);
            if (gs < 0) atomicOr(&s_flags[0], 1); else s_map[gs] = i;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LG_SLOTS; i += BT) {
        const int l = s_map[i];
        if (l < 0) continue;                                              // (its count is zero since the prologue; the merge adds where the count is positive)
        const size_t e = (size_t)i * gridDim.x + blockIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) pacc[e * 4 + k] = k < NV ? s_acc[l][k < NVS ? k : 0] : 0.0;
        pcnt[e] = (int64_t)s_cnt[l];
    }
    if (threadIdx.x == 0 && s_flags[0]) atomicOr(flags, s_flags[0]);
}

// =================================================================================================
// K-F: compact the entries with hits >= min_hits.  The entries are the owning stage rows, dense
// per wave segment, so this reads 4 bytes per entry instead of scanning a slot array.  Three small
// kernels and no global atomics: survivors per segment, exclusive scan of the counts, write at
// ballot-prefix positions — the output keeps build-row order, so it is the same in every run.
// =================================================================================================
struct DevCompactOut {
    int64_t* keys; int64_t* pay[SDQH_MAX_PAYLOAD]; double* val[SDQH_TUPLE_MAX_VALUES]; int64_t* hits;
    unsigned long long* counter;
    int32_t npay, nval;
    // When the caller's result arrays are device-visible host memory and the result fits them, the
    // write kernel stores the rows there itself (coalesced stores over PCIe): no copy-engine
    // launches afterwards, one synchronisation for the whole query.
    int64_t* h_keys; int64_t* h_pay[SDQH_MAX_PAYLOAD]; double* h_val[SDQH_TUPLE_MAX_VALUES]; int64_t* h_hits;
    unsigned long long* h_counter;
    uint64_t host_rows;
    int32_t direct, bounded;                                          // bounded: keys / pay / val / hits hold host_rows rows each, rows beyond are dropped
};

constexpr int COMPACT_BATCH = 8;                                     // 64-entry groups whose hit counters are fetched together

// one survivor per lane, converged: every load of the copy is in flight before the first store
__device__ __forceinline__ void compact_copy(const DevStage& st, const DevCompactOut& o, int64_t idx, uint64_t at, uint32_t hits) {
    if (o.bounded && at >= o.host_rows) return;                       // a destination sized from the caller's capacity: the count still comes out right
    int64_t k = 0, pay[SDQH_MAX_PAYLOAD] = {0, 0, 0, 0};
    double val[SDQH_TUPLE_MAX_VALUES] = {0, 0, 0, 0};
    if (o.keys) k = st.key[idx];
#pragma unroll
    for (int q = 0; q < SDQH_MAX_PAYLOAD; ++q) if (q < o.npay && o.pay[q]) pay[q] = st.pay[q][idx];
#pragma unroll
    for (int v = 0; v < SDQH_TUPLE_MAX_VALUES; ++v) if (v < o.nval && o.val[v]) val[v] = st.sacc[(size_t)idx * st.acc_stride + v];
    if (o.keys) o.keys[at] = k;
#pragma unroll
    for (int q = 0; q < SDQH_MAX_PAYLOAD; ++q) if (q < o.npay && o.pay[q]) o.pay[q][at] = pay[q];
#pragma unroll
    for (int v = 0; v < SDQH_TUPLE_MAX_VALUES; ++v) if (v < o.nval && o.val[v]) o.val[v][at] = val[v];
    if (o.hits) o.hits[at] = (int64_t)hits;
}

// Survivors of one wave segment, counted (WRITE = false) or written from out0 on (WRITE = true).
// When writing, the survivors' stage indices are first queued in LDS (ballot + popcount prefix, in
// row order) and copied 64 at a time, one per lane — the same converged-drain idea as k_probe_agg.
constexpr int COMPACT_QCAP = 64 * (COMPACT_BATCH + 1);
template <bool WRITE>
__device__ __forceinline__ uint32_t compact_segment(const DevTable& t, const DevStage& st, const DevCompactOut& o, uint32_t min_hits,
                                                    int seg, uint64_t out0, uint32_t* q_idx, uint32_t* q_hits) {
    const int lane = lane_id();
    const bool dups = t.hdr->has_dups != 0;
    const uint64_t mask = (table_is_direct(t) || !dups) ? 0 : t.hdr->cap_mask;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    const uint64_t lt = lanemask_lt();
    uint32_t mine = 0;
    int qn = 0;
    uint64_t written = out0;
    for (uint32_t i0 = 0; i0 < count; i0 += WAVE * COMPACT_BATCH) {
        uint32_t h[COMPACT_BATCH];
#pragma unroll
        for (int j = 0; j < COMPACT_BATCH; ++j) { const uint32_t i = i0 + j * WAVE + lane; h[j] = i < count ? st.shits[base + i] : 0u; }
#pragma unroll
        for (int j = 0; j < COMPACT_BATCH; ++j) {
            const uint32_t i = i0 + j * WAVE + lane;
            bool keep = i < count && h[j] >= min_hits;
            if (dups && keep) keep = stage_row_owns(st, t, base + i, mask);
            const uint64_t b = __ballot(keep);
            if (WRITE && keep) { const int at = qn + __popcll(b & lt); q_idx[at] = i; q_hits[at] = h[j]; }
            qn += WRITE ? __popcll(b) : 0;
            mine += (uint32_t)__popcll(b);
        }
        if (WRITE) {
            int done = 0;
            while (qn - done >= WAVE) { compact_copy(st, o, base + q_idx[done + lane], written + lane, q_hits[done + lane]); done += WAVE; written += WAVE; }
            if (done) {                                               // move the leftover (< 64 entries) to the front
                const int left = qn - done;
                uint32_t a = 0, c = 0;
                if (lane < left) { a = q_idx[done + lane]; c = q_hits[done + lane]; }
                if (lane < left) { q_idx[lane] = a; q_hits[lane] = c; }
                qn = left;
            }
        }
    }
    if (WRITE && lane < qn) compact_copy(st, o, base + q_idx[lane], written + lane, q_hits[lane]);
    return mine;
}

// Several small regions set to a byte value by ONE launch (a hipMemsetAsync per region costs a
// launch each, and most builds need two or three).  Regions are 4-byte multiples, 16-byte aligned.
SDQH_KERNEL __launch_bounds__(TPB) void k_fill(DevFillBig f) {
    const uint64_t tid = (uint64_t)blockIdx.x * TPB + threadIdx.x, nth = (uint64_t)gridDim.x * TPB;
    for (int r = 0; r < FILL_BIG; ++r) {
        if (r >= f.n) break;
        const uint32_t w = f.word[r];
        const uint64_t n16 = f.bytes[r] / 16, n4 = f.bytes[r] / 4;
        uint4* p16 = static_cast<uint4*>(f.p[r]);
        for (uint64_t i = tid; i < n16; i += nth) p16[i] = make_uint4(w, w, w, w);
        uint32_t* p4 = static_cast<uint32_t*>(f.p[r]);
        for (uint64_t i = n16 * 4 + tid; i < n4; i += nth) p4[i] = w;
    }
}

// 1. survivors per segment
SDQH_KERNEL __launch_bounds__(TPB) void k_compact_count(DevTable t, DevStage st, DevCompactOut o, uint32_t min_hits, uint32_t* __restrict__ seg_kept) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const uint32_t n = compact_segment<false>(t, st, o, min_hits, seg, 0, nullptr, nullptr);
    if (lane_id() == 0) seg_kept[seg] = n;
}
// 2. one workgroup: exclusive scan of the per-segment counts, in place; total -> *o.counter
SDQH_KERNEL __launch_bounds__(TPB) void k_compact_scan(uint32_t* __restrict__ seg_kept, int nseg, unsigned long long* __restrict__ counter) {
    __shared__ unsigned long long s_part[TPB];
    const int per = (nseg + TPB - 1) / TPB;
    const int b0 = threadIdx.x * per, b1 = min(nseg, b0 + per);
    unsigned long long sum = 0;
    for (int b = b0; b < b1; ++b) sum += seg_kept[b];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over the 256 partial sums
    for (int off = 1; off < TPB; off <<= 1) {
        unsigned long long v = (int)threadIdx.x >= off ? s_part[threadIdx.x - off] : 0ull;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    if (threadIdx.x == TPB - 1) *counter = s_part[TPB - 1];
    unsigned long long run = s_part[threadIdx.x] - sum;              // exclusive
    // offsets are kept in 32 bits per segment + a 64-bit base per thread chunk would be needed past
    // 2^32 result rows; a build side is limited to 2^32-2 rows, so 32 bits are enough
    for (int b = b0; b < b1; ++b) { const uint32_t v = seg_kept[b]; seg_kept[b] = (uint32_t)run; run += v; }
}
// 3. write: output order = stage order = build-row order (deterministic)
SDQH_KERNEL __launch_bounds__(TPB) void k_compact_write(DevTable t, DevStage st, DevCompactOut o, uint32_t min_hits, const uint32_t* __restrict__ seg_off) {
    __shared__ uint32_t s_idx[TPB / WAVE][COMPACT_QCAP], s_hits[TPB / WAVE][COMPACT_QCAP];
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    const unsigned long long total = *o.counter;
    if (o.h_counter && blockIdx.x == 0 && threadIdx.x == 0) *o.h_counter = total;
    if (seg >= st.nseg) return;
    if (o.direct && total <= o.host_rows) {                            // the result fits the caller's arrays: write it there
        o.keys = o.h_keys; o.hits = o.h_hits;
#pragma unroll
        for (int q = 0; q < SDQH_MAX_PAYLOAD; ++q) o.pay[q] = o.h_pay[q];
#pragma unroll
        for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) o.val[k] = o.h_val[k];
    }
    compact_segment<true>(t, st, o, min_hits, seg, (uint64_t)seg_off[seg], s_idx[threadIdx.x / WAVE], s_hits[threadIdx.x / WAVE]);
}

// 3'. write, every wave finding its own offset: the counts of the segments before its own are summed by its 64 lanes (a few
// thousand 4-byte reads, L2-resident) — no scan launch between count and write.  The wave of the last segment publishes the
// total.  Rows go to the device staging arrays in o (bounded by o.host_rows); the host copies them out behind the kernels.
SDQH_KERNEL __launch_bounds__(TPB) void k_compact_write2(DevTable t, DevStage st, DevCompactOut o, uint32_t min_hits, const uint32_t* __restrict__ seg_kept) {
    __shared__ uint32_t s_idx[TPB / WAVE][COMPACT_QCAP], s_hits[TPB / WAVE][COMPACT_QCAP];
    __shared__ long long s_part[TPB / WAVE];
    const int wave = threadIdx.x / WAVE;
    const int seg0 = blockIdx.x * (TPB / WAVE), seg = seg0 + wave;
    // the counts before this WORKGROUP's first segment, summed once by all of its threads (every wave summing its own prefix read the
    // counts four times over: 58 MB of L2 reads for Q3's 6 K segments); then the one to three counts in between
    long long part = 0;
    for (int i = threadIdx.x; i < min(seg0, st.nseg); i += TPB) part += seg_kept[i];
    part = wave_sum_i64(part);
    if (lane_id() == 0) s_part[wave] = part;
    __syncthreads();
    if (seg >= st.nseg) return;
    long long before = 0;
#pragma unroll
    for (int w = 0; w < TPB / WAVE; ++w) before += s_part[w];
    for (int j = 0; j < wave; ++j) before += seg_kept[seg0 + j];
    if (seg == st.nseg - 1 && lane_id() == 0) {
        const unsigned long long total = (unsigned long long)before + seg_kept[seg];
        *o.counter = total;
        if (o.h_counter) *o.h_counter = total;
    }
    compact_segment<true>(t, st, o, min_hits, seg, (uint64_t)before, s_idx[threadIdx.x / WAVE], s_hits[threadIdx.x / WAVE]);
}

// K-F's rows out of the device staging buffer into the caller's pinned host block, by a kernel of this library's own on the copy stream
// (sdqh_table_compact_deferred).  The runtime's copy is a blit kernel whose stores to host memory sit dirty in L2: a compute kernel
// of the next query that ENDS while it runs waits for them in its own end-of-kernel write-back (Q5's one-workgroup nation build: 81 us
// instead of 9 beside Q3's 3.6 MB).  Here the stores are non-temporal (written through), a few workgroups, PCIe-bound either way.
typedef unsigned int sdqh_u4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(TPB) void k_copy_out(const sdqh_u4* __restrict__ src, sdqh_u4* __restrict__ dst, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * TPB) {
        if (NT) { const sdqh_u4 v = __builtin_nontemporal_load(src + i); __builtin_nontemporal_store(v, dst + i); }
        else dst[i] = src[i];
    }
}

// =================================================================================================
// Top-k over the entries of a table (ORDER BY ... LIMIT k after K-F; BASELINE config "Q3 ... + top-k",
// SURVEY.md §8f.2 — the reference has no such operator, its Q3 returns the whole set).
// Order: up to three sort keys (the entry's key, a payload field, an accumulator or its hit count,
// each ascending or descending), then build-row order — a total order, so the result is the same in
// every run and on every implementation of the ABI.
//   level 0  k_topk_scan    each workgroup walks its share of the stage segments, keeps the entries
//                           with hits >= min_hits in a 1024-slot LDS buffer and cuts it back to its k
//                           best whenever the next batch might not fit; writes k candidates (padded)
//   level i  k_topk_reduce  1024 candidates per workgroup -> k, until one workgroup is left; that
//                           one writes the result rows
// Selection inside a workgroup: k <= 16 -> k rounds of "argmin of the rest" by one wave; larger k ->
// a bitonic sort of the slot numbers by all four waves (top_block_select / top_block_sort).
// =================================================================================================
constexpr int TOPK_CHUNK = 1024;
constexpr int TOPK_PER_LANE = TOPK_CHUNK / WAVE;                       // 16 slots per lane of the selecting wave
struct DevSortKey { int32_t kind, index, desc, is_f64; };
struct DevTopSpec { DevSortKey key[SDQH_MAX_SORT_KEYS]; int32_t nsort, k; };
struct DevTopBuf { uint64_t* k0; uint64_t* k1; uint64_t* k2; uint32_t* ref; };
struct DevTopOut { int64_t* keys; int64_t* pay; double* val; int64_t* hits; unsigned long long* count; int32_t npay, nval; };   // pay[p*k+i], val[v*k+i]

// order-preserving map onto uint64 where smaller sorts first in the requested direction
__device__ __forceinline__ uint64_t sort_bits(int64_t raw, int is_f64, int desc) {
    uint64_t u = (uint64_t)raw;
    if (is_f64) u = (u >> 63) ? ~u : (u | (1ull << 63)); else u ^= (1ull << 63);
    return desc ? ~u : u;
}
__device__ __forceinline__ uint64_t top_sort_value(const DevSortKey& sk, const DevStage& st, int64_t idx, uint32_t hits) {
    int64_t raw = 0;
    if (sk.kind == SDQH_SORT_KEY) raw = st.key[idx];
    else if (sk.kind == SDQH_SORT_PAYLOAD) { const int64_t* p = sk.index == 0 ? st.pay[0] : (sk.index == 1 ? st.pay[1] : (sk.index == 2 ? st.pay[2] : st.pay[3])); raw = p[idx]; }
    else if (sk.kind == SDQH_SORT_VALUE) raw = __double_as_longlong(st.sacc[(size_t)idx * st.acc_stride + sk.index]);
    else raw = (int64_t)hits;
    return sort_bits(raw, sk.is_f64, sk.desc);
}

struct TopLds {
    uint64_t k0[TOPK_CHUNK], k1[TOPK_CHUNK], k2[TOPK_CHUNK];
    uint32_t ref[TOPK_CHUNK];
    uint64_t o0[SDQH_MAX_TOPK], o1[SDQH_MAX_TOPK], o2[SDQH_MAX_TOPK];     // the selection, in order
    uint32_t oref[SDQH_MAX_TOPK];
    uint16_t perm[TOPK_CHUNK];                                             // sorting network: slot numbers in sorted order
    int count;
};
__device__ __forceinline__ bool top_less(const TopLds& s, int a, int b) {       // slot a sorts strictly before slot b
    if (s.k0[a] != s.k0[b]) return s.k0[a] < s.k0[b];
    if (s.k1[a] != s.k1[b]) return s.k1[a] < s.k1[b];
    if (s.k2[a] != s.k2[b]) return s.k2[a] < s.k2[b];
    if (s.ref[a] != s.ref[b]) return s.ref[a] < s.ref[b];
    return a < b;                                   // padding slots are all alike: the slot number keeps the order strict, so every lane of the butterfly agrees on the winner
}
// best live slot of this lane (slots lane + 64*i): smallest primary key, equal primaries compared in LDS
__device__ __forceinline__ void top_lane_best(const TopLds& s, const uint64_t (&kk)[TOPK_PER_LANE], uint32_t live, int lane, int& best, uint64_t& bk) {
    best = -1; bk = ~0ull;
#pragma unroll
    for (int i = 0; i < TOPK_PER_LANE; ++i) {
        const bool lv = (live >> i) & 1u;
        bool lt = lv && (best < 0 || kk[i] < bk);
        if (lv && best >= 0 && kk[i] == bk) lt = top_less(s, lane + WAVE * i, best);      // rare: the branch is skipped when no lane ties
        if (lt) { best = lane + WAVE * i; bk = kk[i]; }
    }
}
// The k best of slots [0, count) moved to the front in order; returns min(count, k).  All threads
// call it (workgroup barriers inside).  The selection itself is done by ONE wave, barrier-free:
// each lane keeps the primary keys of its 16 slots in registers and its current best slot; a round
// is a 6-step butterfly on (primary key, slot) — equal primaries go back to LDS for the other keys —
// after which only the lane that owned the winner looks for its next best.
// Larger k: sort the slot numbers of [0, count) with a bitonic network (all four waves; 55 passes of
// two compare-exchanges per thread for 1024 slots, independent of k) and take the first k.
__device__ __forceinline__ int top_block_sort(TopLds& s, int count, int k) {
    const int rounds = min(count, k);
    int n2 = 2;
    while (n2 < count) n2 <<= 1;
    __syncthreads();
    for (int i = threadIdx.x; i < n2; i += TPB) s.perm[i] = i < count ? (uint16_t)i : (uint16_t)0xFFFF;
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < n2 / 2; t += TPB) {
                const int i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool up = (i & size) == 0;
                const int a = s.perm[i], b = s.perm[j];
                const bool a_lt_b = a != 0xFFFF && (b == 0xFFFF || top_less(s, a, b));      // padding sorts last
                const bool b_lt_a = b != 0xFFFF && (a == 0xFFFF || top_less(s, b, a));
                if (up ? b_lt_a : a_lt_b) { s.perm[i] = (uint16_t)b; s.perm[j] = (uint16_t)a; }
            }
            __syncthreads();
        }
    }
    for (int j = threadIdx.x; j < rounds; j += TPB) { const int p = s.perm[j]; s.o0[j] = s.k0[p]; s.o1[j] = s.k1[p]; s.o2[j] = s.k2[p]; s.oref[j] = s.ref[p]; }
    __syncthreads();
    for (int j = threadIdx.x; j < rounds; j += TPB) { s.k0[j] = s.o0[j]; s.k1[j] = s.o1[j]; s.k2[j] = s.o2[j]; s.ref[j] = s.oref[j]; }
    __syncthreads();
    return rounds;
}
__device__ __forceinline__ int top_block_select(TopLds& s, int count, int k) {
    if (k > 16) return top_block_sort(s, count, k);                      // k rounds of argmin cost more than one sort beyond that
    const int rounds = min(count, k);
    __syncthreads();
    if (threadIdx.x < WAVE) {
        const int lane = (int)threadIdx.x;
        uint64_t kk[TOPK_PER_LANE];
        uint32_t live = 0;                                                // bit i: slot lane + 64*i is a candidate not yet taken
#pragma unroll
        for (int i = 0; i < TOPK_PER_LANE; ++i) { const int j = lane + WAVE * i; kk[i] = j < count ? s.k0[j] : ~0ull; if (j < count) live |= 1u << i; }
        int mine; uint64_t mk;
        top_lane_best(s, kk, live, lane, mine, mk);
        for (int it = 0; it < rounds; ++it) {
            int best = mine; uint64_t bk = mk;
#pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const int ob = __shfl_xor(best, off, WAVE);
                const uint64_t ok = (uint64_t)__shfl_xor((long long)bk, off, WAVE);
                bool take = ob >= 0 && (best < 0 || ok < bk);
                if (ob >= 0 && best >= 0 && ok == bk && ob != best) take = top_less(s, ob, best);
                if (take) { best = ob; bk = ok; }
            }
            // every lane now holds the same winner (top_less is a strict total order on slots)
            if (lane == 0) { s.o0[it] = s.k0[best]; s.o1[it] = s.k1[best]; s.o2[it] = s.k2[best]; s.oref[it] = s.ref[best]; }
            if (best == mine) { live &= ~(1u << (best / WAVE)); top_lane_best(s, kk, live, lane, mine, mk); }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < rounds; j += TPB) { s.k0[j] = s.o0[j]; s.k1[j] = s.o1[j]; s.k2[j] = s.o2[j]; s.ref[j] = s.oref[j]; }
    __syncthreads();
    return rounds;
}
__device__ __forceinline__ void top_write_padded(const TopLds& s, int have, int k, const DevTopBuf& out, size_t at) {
    for (int j = threadIdx.x; j < k; j += TPB) {
        const bool real = j < have;
        out.k0[at + j] = real ? s.k0[j] : ~0ull; out.k1[at + j] = real ? s.k1[j] : ~0ull; out.k2[at + j] = real ? s.k2[j] : ~0ull;
        out.ref[at + j] = real ? s.ref[j] : NO_ROW;
    }
}
// result rows of the final selection, in order; padding slots are not rows
__device__ __forceinline__ void top_emit(const TopLds& s, int have, int k, const DevStage& st, const DevTopOut& o) {
    int rows = 0;
    for (int j = 0; j < have; ++j) if (s.ref[j] != NO_ROW) rows = j + 1;           // real entries sort before padding
    for (int j = threadIdx.x; j < rows; j += TPB) {
        const int64_t idx = (int64_t)s.ref[j];
        if (o.keys) o.keys[j] = st.key[idx];
#pragma unroll
        for (int p = 0; p < SDQH_MAX_PAYLOAD; ++p) if (p < o.npay && o.pay) o.pay[(size_t)p * k + j] = st.pay[p][idx];
#pragma unroll
        for (int v = 0; v < SDQH_TUPLE_MAX_VALUES; ++v) if (o.val) o.val[(size_t)v * k + j] = v < o.nval ? st.sacc[(size_t)idx * st.acc_stride + v] : 0.0;
        if (o.hits) o.hits[j] = st.shits ? (int64_t)st.shits[idx] : 0;
    }
    if (threadIdx.x == 0) *o.count = (unsigned long long)rows;
}

SDQH_KERNEL __launch_bounds__(TPB) void k_topk_scan(DevTable t, DevStage st, DevTopSpec spec, uint32_t min_hits, DevTopBuf out, DevTopOut fin) {
    __shared__ TopLds s;
    if (threadIdx.x == 0) s.count = 0;
    __syncthreads();
    const bool dups = t.hdr->has_dups != 0;
    const uint64_t mask = (table_is_direct(t) || !dups) ? 0 : t.hdr->cap_mask;
    const int per = (st.nseg + (int)gridDim.x - 1) / (int)gridDim.x;
    const int s0 = blockIdx.x * per, s1 = min(st.nseg, s0 + per);
    const uint64_t lt = lanemask_lt();
    for (int seg = s0; seg < s1; ++seg) {
        const int64_t base = (int64_t)seg * st.seg_rows;
        const uint32_t count = st.seg_count[seg];
        for (uint32_t i0 = 0; i0 < count; i0 += TPB) {                       // uniform across the workgroup
            const uint32_t i = i0 + threadIdx.x;
            uint32_t hits = 0;
            bool keep = i < count;
            if (keep) { hits = st.shits ? st.shits[base + i] : 0u; keep = hits >= min_hits; }
            if (keep && dups) keep = stage_row_owns(st, t, base + i, mask);
            const uint64_t b = __ballot(keep);
            int wbase = 0;
            if (lane_id() == 0 && b) wbase = atomicAdd(&s.count, __popcll(b));
            wbase = __shfl(wbase, 0, WAVE);
            if (keep) {
                const int at = wbase + __popcll(b & lt);
                const int64_t idx = base + i;
                s.k0[at] = top_sort_value(spec.key[0], st, idx, hits);
                s.k1[at] = spec.nsort > 1 ? top_sort_value(spec.key[1], st, idx, hits) : 0ull;
                s.k2[at] = spec.nsort > 2 ? top_sort_value(spec.key[2], st, idx, hits) : 0ull;
                s.ref[at] = (uint32_t)idx;
            }
            __syncthreads();
            if (s.count > TOPK_CHUNK - TPB) {                               // the next batch might not fit; uniform: everybody reads the same LDS word
                const int have = top_block_select(s, s.count, spec.k);
                if (threadIdx.x == 0) s.count = have;
                __syncthreads();
            }
        }
    }
    __syncthreads();
    const int have = top_block_select(s, s.count, spec.k);
    if (gridDim.x == 1) top_emit(s, have, spec.k, st, fin);
    else top_write_padded(s, have, spec.k, out, (size_t)blockIdx.x * spec.k);
}

SDQH_KERNEL __launch_bounds__(TPB) void k_topk_reduce(DevTopBuf in, int n_in, DevStage st, DevTopSpec spec, DevTopBuf out, DevTopOut fin) {
    __shared__ TopLds s;
    const int lo = blockIdx.x * TOPK_CHUNK, n = min(TOPK_CHUNK, n_in - lo);
    for (int j = threadIdx.x; j < n; j += TPB) { s.k0[j] = in.k0[lo + j]; s.k1[j] = in.k1[lo + j]; s.k2[j] = in.k2[lo + j]; s.ref[j] = in.ref[lo + j]; }
    __syncthreads();
    const int have = top_block_select(s, n, spec.k);
    if (gridDim.x == 1) top_emit(s, have, spec.k, st, fin);
    else top_write_padded(s, have, spec.k, out, (size_t)blockIdx.x * spec.k);
}

// =================================================================================================
// Redistribution helpers for the multi-GPU join (no reference counterpart).
// =================================================================================================
// exclusive scan of the per-segment counts into seg_off (separate array), total -> *total
SDQH_KERNEL __launch_bounds__(TPB) void k_seg_scan(const uint32_t* __restrict__ seg_count, int nseg, uint64_t* __restrict__ seg_off,
                                                  unsigned long long* __restrict__ total) {
    __shared__ unsigned long long s_part[TPB];
    const int per = (nseg + TPB - 1) / TPB;
    const int b0 = threadIdx.x * per, b1 = min(nseg, b0 + per);
    unsigned long long sum = 0;
    for (int b = b0; b < b1; ++b) sum += seg_count[b];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < TPB; off <<= 1) {
        unsigned long long v = (int)threadIdx.x >= off ? s_part[threadIdx.x - off] : 0ull;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    if (threadIdx.x == TPB - 1) *total = s_part[TPB - 1];
    unsigned long long run = s_part[threadIdx.x] - sum;
    for (int b = b0; b < b1; ++b) { seg_off[b] = run; run += seg_count[b]; }
}
// dense output columns from the segmented stage: wave per segment, coalesced copies
struct DevGather { int64_t* out[SDQH_MAX_COMPACT_COLS]; int32_t ncols, _pad; };
SDQH_KERNEL __launch_bounds__(TPB) void k_gather_segments(DevStage st, const uint64_t* __restrict__ seg_off, DevGather g) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    const uint64_t off = seg_off[seg];
    for (uint32_t i = lane_id(); i < count; i += WAVE) {
        g.out[0][off + i] = st.key[base + i];
#pragma unroll
        for (int c = 1; c < SDQH_MAX_COMPACT_COLS; ++c) if (c < g.ncols) g.out[c][off + i] = st.pay[c - 1][base + i];
    }
}

// partition function shared with the CPU build: mix64(key) % nparts, or range lookup
struct DevPartition { int64_t upper[SDQH_MAX_PARTS]; int32_t nparts, by_range; };
__device__ __forceinline__ int part_of(const DevPartition& pt, int64_t key) {
    if (!pt.by_range) return (int)(mix64((uint64_t)key) % (uint64_t)pt.nparts);
    int p = 0;
    while (p < pt.nparts - 1 && key > pt.upper[p]) ++p;
    return p;
}
SDQH_KERNEL __launch_bounds__(TPB) void k_part_count(const int64_t* __restrict__ key, int64_t nrows, DevPartition pt, unsigned long long* __restrict__ counts) {
    __shared__ unsigned int s_hist[SDQH_MAX_PARTS];
    if (threadIdx.x < SDQH_MAX_PARTS) s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * TPB) atomicAdd(&s_hist[part_of(pt, key[r])], 1u);
    __syncthreads();
    if ((int)threadIdx.x < pt.nparts && s_hist[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}
// counts -> exclusive offsets (cursor), one thread
SDQH_KERNEL void k_part_offsets(const unsigned long long* __restrict__ counts, int nparts, unsigned long long* __restrict__ cursor) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { unsigned long long run = 0; for (int p = 0; p < nparts; ++p) { cursor[p] = run; run += counts[p]; } }
}
// scatter: a workgroup takes 2048 rows at a time, ranks them inside their part with LDS atomics,
// reserves the part's output range with ONE global atomic per part, then writes
constexpr int PART_ROWS_PER_THREAD = 8;
SDQH_KERNEL __launch_bounds__(TPB) void k_part_scatter(const int64_t* __restrict__ key, int64_t nrows, DevPartition pt,
                                                      unsigned long long* __restrict__ cursor, DevGather src, DevGather dst) {
    __shared__ unsigned int s_hist[SDQH_MAX_PARTS];
    __shared__ unsigned long long s_base[SDQH_MAX_PARTS];
    constexpr int64_t CHUNK = (int64_t)TPB * PART_ROWS_PER_THREAD;
    for (int64_t c0 = (int64_t)blockIdx.x * CHUNK; c0 < nrows; c0 += (int64_t)gridDim.x * CHUNK) {
        if (threadIdx.x < SDQH_MAX_PARTS) s_hist[threadIdx.x] = 0;
        __syncthreads();
        int part[PART_ROWS_PER_THREAD]; unsigned int local[PART_ROWS_PER_THREAD];
#pragma unroll
        for (int j = 0; j < PART_ROWS_PER_THREAD; ++j) {
            const int64_t r = c0 + (int64_t)j * TPB + threadIdx.x;
            part[j] = r < nrows ? part_of(pt, key[r]) : -1;
        }
#pragma unroll
        for (int j = 0; j < PART_ROWS_PER_THREAD; ++j) local[j] = part[j] >= 0 ? atomicAdd(&s_hist[part[j]], 1u) : 0u;
        __syncthreads();
        if ((int)threadIdx.x < pt.nparts) s_base[threadIdx.x] = s_hist[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]) : 0ull;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PART_ROWS_PER_THREAD; ++j) {
            if (part[j] < 0) continue;
            const int64_t r = c0 + (int64_t)j * TPB + threadIdx.x;
            const uint64_t at = s_base[part[j]] + local[j];
#pragma unroll
            for (int c = 0; c < SDQH_MAX_COMPACT_COLS; ++c) if (c < src.ncols) dst.out[c][at] = src.out[c][r];
        }
        __syncthreads();
    }
}

// The same scatter into ONE buffer laid out for an all-to-all: the chunk of part p (every column's rows of that part, column after
// column) starts at element ncols * base[p]; counts / base = what k_part_count / k_part_offsets left.  One collective then moves
// every column of a redistribution step, and nothing is copied between the partitioning pass and the collective's buffer.
SDQH_KERNEL __launch_bounds__(TPB) void k_part_scatter_packed(const int64_t* __restrict__ key, int64_t nrows, DevPartition pt, unsigned long long* __restrict__ cursor,
                                                             const unsigned long long* __restrict__ counts, const unsigned long long* __restrict__ base, DevGather src, int64_t* __restrict__ packed) {
    __shared__ unsigned int s_hist[SDQH_MAX_PARTS];
    __shared__ unsigned long long s_base[SDQH_MAX_PARTS];
    constexpr int64_t CHUNK = (int64_t)TPB * PART_ROWS_PER_THREAD;
    for (int64_t c0 = (int64_t)blockIdx.x * CHUNK; c0 < nrows; c0 += (int64_t)gridDim.x * CHUNK) {
        if (threadIdx.x < SDQH_MAX_PARTS) s_hist[threadIdx.x] = 0;
        __syncthreads();
        int part[PART_ROWS_PER_THREAD]; unsigned int local[PART_ROWS_PER_THREAD];
#pragma unroll
        for (int j = 0; j < PART_ROWS_PER_THREAD; ++j) {
            const int64_t r = c0 + (int64_t)j * TPB + threadIdx.x;
            part[j] = r < nrows ? part_of(pt, key[r]) : -1;
        }
#pragma unroll
        for (int j = 0; j < PART_ROWS_PER_THREAD; ++j) local[j] = part[j] >= 0 ? atomicAdd(&s_hist[part[j]], 1u) : 0u;
        __syncthreads();
        if ((int)threadIdx.x < pt.nparts) s_base[threadIdx.x] = s_hist[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]) : 0ull;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PART_ROWS_PER_THREAD; ++j) {
            if (part[j] < 0) continue;
            const int64_t r = c0 + (int64_t)j * TPB + threadIdx.x;
            const uint64_t b = base[part[j]], n = counts[part[j]];
            const uint64_t at = (uint64_t)src.ncols * b + (s_base[part[j]] + local[j] - b);
#pragma unroll
            for (int c = 0; c < SDQH_MAX_COMPACT_COLS; ++c) if (c < src.ncols) packed[at + (uint64_t)c * n] = src.out[c][r];
        }
        __syncthreads();
    }
}
// ... and the received buffer (a chunk per source rank, laid out the same way) taken apart into contiguous columns: block (y) = (source, column)
struct DevUnpack { int64_t src_off[SDQH_MAX_PARTS]; int64_t dst_off[SDQH_MAX_PARTS]; int64_t rows[SDQH_MAX_PARTS]; int32_t nparts, ncols; };
SDQH_KERNEL __launch_bounds__(TPB) void k_unpack_parts(const int64_t* __restrict__ packed, DevUnpack u, DevGather dst) {
    const int s = (int)blockIdx.y / u.ncols, c = (int)blockIdx.y % u.ncols;
    const int64_t n = u.rows[s];
    const int64_t* from = packed + u.src_off[s] + (int64_t)c * n;
    int64_t* to = nullptr;
#pragma unroll
    for (int k = 0; k < SDQH_MAX_COMPACT_COLS; ++k) if (k == c) to = dst.out[k];
    to += u.dst_off[s];
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) to[i] = from[i];
}

// exact bitmap of a table's keys over [lo, hi] from its stage rows
SDQH_KERNEL __launch_bounds__(TPB) void k_export_bitmap(DevStage st, int64_t lo, int64_t hi, uint32_t* __restrict__ words) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    for (uint32_t i = lane_id(); i < count; i += WAVE) {
        const int64_t k = st.key[base + i];
        if (k >= lo && k <= hi) { const uint64_t off = (uint64_t)(k - lo); atomicOr(&words[off >> 5], 1u << (off & 31)); }
    }
}

// the same from a table's own exact bitmap (key sets, the direct layout): destination word w covers keys lo + 32 w ...; its bits are
// two source words shifted into place — no atomics, no pass over the entries
SDQH_KERNEL __launch_bounds__(TPB) void k_export_bits(const uint32_t* __restrict__ src, int64_t src_lo, int64_t src_hi, int64_t lo, int64_t hi, uint32_t* __restrict__ words, int64_t nwords) {
    const int64_t src_bits = src_hi - src_lo + 1;
    for (int64_t w = (int64_t)blockIdx.x * TPB + threadIdx.x; w < nwords; w += (int64_t)gridDim.x * TPB) {
        const int64_t first = lo + w * 32;                                   // key of this word's bit 0
        int64_t s = first - src_lo;                                          // its bit offset in the source (may be negative / beyond)
        uint32_t out = 0;
        if (s > -32 && s < src_bits) {
            const int64_t sw = s >= 0 ? s >> 5 : -1;                           // source word holding bit s (or the one before word 0)
            const int sh = (int)(s - sw * 32);                                 // 0 .. 31
            const int64_t last_word = (src_bits - 1) >> 5;
            const uint32_t a = (sw >= 0 && sw <= last_word) ? src[sw] : 0u;
            const uint32_t b = (sw + 1 >= 0 && sw + 1 <= last_word) ? src[sw + 1] : 0u;
            out = sh ? (a >> sh) | (b << (32 - sh)) : a;
        }
        const int64_t keys_left = hi - first + 1;                            // bits past `hi` stay clear
        if (keys_left < 32) out &= keys_left > 0 ? (0xFFFFFFFFu >> (32 - keys_left)) : 0u;
        words[w] = out;
    }
}

// (hi << 32 | lo) -> its two parts as columns of their own (replicated composite-key tables are rebuilt from their parts)
SDQH_KERNEL __launch_bounds__(TPB) void k_unpack2(const int64_t* __restrict__ src, int64_t nrows, int64_t* __restrict__ hi, int64_t* __restrict__ lo) {
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * TPB) {
        const uint64_t v = (uint64_t)src[r];
        hi[r] = (int64_t)(v >> 32); lo[r] = (int64_t)(v & 0xFFFFFFFFull);
    }
}

// ---- column statistics ---------------------------------------------------------------------------
// build + verify a narrow twin (flag |= 1: some row is not reproduced exactly -> the column keeps its 8-byte form only)
SDQH_KERNEL __launch_bounds__(TPB) void k_narrow_i64(const int64_t* __restrict__ src, int64_t nrows, int32_t* __restrict__ dst, int* __restrict__ flag) {
    bool bad = false;
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * TPB) {
        const int64_t v = src[r];
        bad |= v < -2147483647ll - 1 || v > 2147483647ll;
        dst[r] = (int32_t)v;
    }
    if (__ballot(bad) && lane_id() == 0) atomicOr(flag, 1);
}
SDQH_KERNEL __launch_bounds__(TPB) void k_narrow_str1(const uint32_t* __restrict__ src, int64_t nrows, uint8_t* __restrict__ dst, int* __restrict__ flag) {
    bool bad = false;
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * TPB) { const uint32_t v = src[r]; bad |= v > 0xFFu; dst[r] = (uint8_t)v; }
    if (__ballot(bad) && lane_id() == 0) atomicOr(flag, 1);
}
SDQH_KERNEL __launch_bounds__(TPB) void k_narrow_f64(const double* __restrict__ src, int64_t nrows, int32_t* __restrict__ dst, int* __restrict__ flag) {
    bool bad = false;
    for (int64_t r = (int64_t)blockIdx.x * TPB + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * TPB) {
        const double v = src[r];
        const double s100 = v * 100.0;
        int32_t n = 0;
        if (s100 > -2147483000.0 && s100 < 2147483000.0) n = (int32_t)__builtin_rint(s100); else bad = true;      // (NaN fails both comparisons)
        bad |= __double_as_longlong(narrow_decode(n)) != __double_as_longlong(v);                                    // bit for bit: -0.0, more decimals, huge values all fail
        dst[r] = n;
    }
    if (__ballot(bad) && lane_id() == 0) atomicOr(flag, 1);
}
SDQH_KERNEL __launch_bounds__(TPB) void k_minmax(const int64_t* __restrict__ col, int64_t nrows, long long* __restrict__ out /*[2]*/) {
    __shared__ long long s_lo[TPB / WAVE], s_hi[TPB / WAVE];
    long long lo = INT64_MAX, hi = INT64_MIN;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nrows; i += (int64_t)gridDim.x * TPB) {
        long long v = col[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi;
    }
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        long long l2 = __shfl_down(lo, off, WAVE), h2 = __shfl_down(hi, off, WAVE);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if (lane_id() == 0) { s_lo[threadIdx.x / WAVE] = lo; s_hi[threadIdx.x / WAVE] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {                                           // one atomic pair per workgroup
        for (int i = 1; i < TPB / WAVE; ++i) { lo = s_lo[i] < lo ? s_lo[i] : lo; hi = s_hi[i] > hi ? s_hi[i] : hi; }
        atomicMin(&out[0], lo); atomicMax(&out[1], hi);
    }
}

}  // namespace sdqh
