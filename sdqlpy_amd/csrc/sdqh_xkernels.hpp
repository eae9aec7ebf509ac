// sdqh_xkernels.hpp — kernel skeletons that are specialised at run time on a row program (ABI 4).
//
// The hand-written part of a specialised kernel lives here: how rows are streamed (16-byte loads
// of the columns the cheap leading conditions need, every load of a step issued before the first
// use), how survivors are queued in LDS in row order and drained 64 at a time, one row per lane,
// so that the dependent chain of a row (lookups -> payload gathers -> arithmetic) runs converged,
// and the five things a loop can do with a row that passed (the "sinks" below).  What the front
// end's expression compiler contributes is one small struct P generated from the program
// (sdqh_x.hip, codegen): the loads of the streamed columns, the conditions on them, and the
// evaluation of everything else for one row.  hiprtc compiles   skeleton<P, sink>   for gfx950;
// nothing here is ever compiled by hipcc into the library.
//
// Reference loop shapes (edin-dal/sdqlpy, src/sdqlpy/lib/sdql_ir_cpp_generator_par.py): K-A 258-291
// (XSum), K-C 402-440 (XGroup: small domain; XEntry: the group is the matched entry), K-B 331-369
// (XStage, XKeySet), with lookups 85-96 and conditions / values of any shape (712-795).
#pragma once
#include "sdqh_kernels.hpp"

// A/B switches of the skeletons (SDQLPY_AMD_X_DEFINES puts macros in front of a specialised source; defaults here)
#ifndef XGL_IVAL
#define XGL_IVAL -1                                                   // the per-lane group sink: index of the summed value that arrives as a small integer (see XGroupLane), or -1
#endif
#ifndef XE_EXP
#define XE_EXP 0                                                      // timing experiments of the entry sink (wrong results): 1 no atomics, 2 no count atomic
#endif
#ifndef X8_DRAIN_AT
#define X8_DRAIN_AT WAVE                                              // x_queue8 drains the front of its queue once this many rows wait (64: only whole waves)
#endif
#ifndef X8_EXP
#define X8_EXP 0                                                      // timing experiments of x_queue8 (wrong results): 1 no bitmap requests, 2 nothing queued, 3 queued but never drained, 4 drained rows evaluated twice, 5 nothing reaches the sink (final loops only; 6.. : XGroup's own, see there)
#endif
#ifndef X8_EXP_MINROWS
#define X8_EXP_MINROWS 0                                              // experiments 3 and 14 leave loops over fewer rows alone (a build that feeds the build under test)
#endif
#ifndef X8_EXP_BUILD
#define X8_EXP_BUILD 0                                                // 1: the timing experiments touch the BUILD loops (stage / key-set sinks) instead of the final ones
#endif
#ifndef X8_PIPE
#define X8_PIPE 0                                                     // x_queue8: the next double step's streamed loads requested behind this step's bitmap words (see there)
#endif
#ifndef XV_MASKED
#define XV_MASKED 0                                                   // x_vstage8: bitmap words requested by the lanes whose row passed the conditions so far only
#endif
#ifndef XV_OFFALL
#define XV_OFFALL 0
#endif
#ifndef XV_BITS
#define XV_BITS 0                                                     // x_vstage8: the key bitmap's words assembled per wave in LDS and stored, where the build key increases (see there: slower)
#endif
#ifndef XV_PIPE
#define XV_PIPE 1                                                     // x_vstage8: the next step's streamed loads requested behind this step's bitmap words (see there)
#endif

namespace sdqh {

constexpr int X_MAX_CONST = SDQH_MAX_XCONST;
template <bool B> struct XBool { static constexpr bool value = B; };

// by-value argument of every specialised kernel: the bindings of the program (columns, tables,
// constants).  The program's structure is in the code; these can change from run to run.
struct XArgs {
    const void* col[SDQH_MAX_XCOLS];
    const void* ncol[SDQH_MAX_XCOLS];           // exact 4-byte twins of the STREAMED columns that have one (sload reads them; everything by row reads col)
    int32_t width[SDQH_MAX_XCOLS];              // STR columns: code units per row
    const void* dcol[SDQH_MAX_XCOLS];           // delta twins (12 bytes per aligned group of 8 rows: smallest value + eight one-byte offsets) of the key columns a QUEUE program streams that way
    const void* code[SDQH_MAX_XCOLS];           // sorted-dictionary code twins (1 or 2 bytes per row) of the columns a TIGHT program streams that way
    const int64_t* dict[SDQH_MAX_XCOLS];        // their dictionaries: ndict raw 8-byte values, ascending
    int32_t ndict[SDQH_MAX_XCOLS];
    uint32_t cc[X_MAX_CONST];                   // constants of comparisons translated into code space (ranks), see sdqh_x.hip: tight_plan
    int64_t dlo[SDQH_MAX_XCOLS];                // coded integer columns whose distinct values are consecutive: value = code + dlo (no table)
    DevTable tab[SDQH_MAX_XTABLES];
    int64_t ci[X_MAX_CONST];
    double cf[X_MAX_CONST];
    uint32_t spool[SDQH_MAX_XSTR];              // string constants, back to back
    int32_t* flags;                             // |= 2: a key outside its bounds / an unpackable key part
    int64_t key_lo, key_hi;                     // bounds the key must respect (key_lo > key_hi: none)
    // DRIVEN walk (x_queue8; programs generated with P::DRIVEN): the run index of the prefilter's key column — per value v of
    // [run_lo, run_hi + 1] the first row holding a value >= v (sdqh_x.hip: column_run_index) — and the ratio the walk is chosen by (0: never)
    const uint32_t* run_index;
    int64_t run_lo, run_hi;
    int32_t driven_ratio, _pad_driven;
};
// does the program carry the members of the driven walk?  (generated only where it can be taken: programs made before it compile unchanged)
template <class...> using x_void_t = void;
template <class P, class = void> struct x_is_driven { static constexpr bool value = false; };
template <class P> struct x_is_driven<P, x_void_t<decltype(P::DRIVEN)>> { static constexpr bool value = P::DRIVEN; };
// ... and P::PHASH: the prefilter's bitmap is the HASHED filter of a hash-layout table (DevTable::hf), P::shdr(a) that table's header
template <class P, class = void> struct x_is_phash { static constexpr bool value = false; };
template <class P> struct x_is_phash<P, x_void_t<decltype(P::PHASH)>> { static constexpr bool value = P::PHASH; };

template <int NV> struct XOut {
    int64_t key;
    int64_t val[NV > 0 ? NV : 1];               // raw 8 bytes each (f64 values as bits)
    uint32_t ent;                               // stage row of the entry matched by the designated lookup (XEntry)
    bool bad;                                   // the key could not be formed (PACK2 part out of range)
};

__device__ __forceinline__ double x_f(int64_t bits) { return __longlong_as_double(bits); }
__device__ __forceinline__ int64_t x_bits(double v) { return __double_as_longlong(v); }

// a streamed row pair through the column's narrow twin (sdqh_kernels.hpp, loadc): raw 8-byte values as the registers hold them
template <bool TAIL> __device__ __forceinline__ Pair<int64_t> x_sload_narrow_i(const void* p, int64_t r, int64_t nrows) {
    return loadc<TAIL, true>(static_cast<const int64_t*>(p), r, nrows);
}
template <bool TAIL> __device__ __forceinline__ Pair<int64_t> x_sload_narrow_f(const void* p, int64_t r, int64_t nrows) {
    const Pair<double> d = loadc<TAIL, true>(static_cast<const double*>(p), r, nrows);
    Pair<int64_t> v; v.x = x_bits(d.x); v.y = x_bits(d.y);
    return v;
}
// VarChar::firstIndex (reference include/varchar.h:91-97), str.find on the text up to the first NUL
__device__ __forceinline__ int64_t x_first_index(const uint32_t* __restrict__ s, int width, const uint32_t* val, int len) {
    int n = 0;
    while (n < width && s[n] != 0u) ++n;
    for (int st = 0; st + len <= n; ++st) {
        if (len && s[st] != val[0]) continue;
        int k = len ? 1 : 0;
        while (k < len && s[st + k] == val[k]) ++k;
        if (k == len) return st;
    }
    return -1;
}
__device__ __forceinline__ int64_t x_char(const uint32_t* __restrict__ s, int width, int pos) {
    if (pos >= width) return 0;
    for (int k = 0; k < pos; ++k) if (s[k] == 0u) return 0;
    return (int64_t)s[pos];
}
// the text operations on a field of the byte twin staged in LDS (byte offset `off` of `region`)
__device__ __forceinline__ int64_t x8_first_index(const uint32_t* region, int off, int width, const uint32_t* val, int len) {
    return len == 0 ? 0 : (int64_t)lds8_find(region, off, width, val, len);
}
__device__ __forceinline__ bool x8_str_pred(const uint32_t* region, int off, int width, const uint32_t* val, int len, int mode) {
    if (mode == 2) return len == 0 || lds8_find(region, off, width, val, len) >= 0;
    return lds_str_pred(reinterpret_cast<const uint8_t*>(region) + off, width, val, len, mode);
}
__device__ __forceinline__ int64_t x8_char(const uint32_t* region, int off, int width, int pos) {
    const uint8_t* s = reinterpret_cast<const uint8_t*>(region) + off;
    if (pos >= width) return 0;
    for (int k = 0; k < pos; ++k) if (s[k] == 0u) return 0;
    return (int64_t)s[pos];
}
// one lookup: stage row of the matched entry, NO_ROW on a miss (key sets: 0 on a hit)
__device__ __forceinline__ uint32_t x_lookup(const DevTable& t, int64_t key, bool bad) {
    if (bad) return NO_ROW;
    const uint64_t mask = (table_is_direct(t) || t.bitmap_only) ? 0 : t.hdr->cap_mask;
    const int64_t pos = table_find(t, key, mask);
    if (pos < 0) return NO_ROW;
    return t.bitmap_only ? 0u : table_ref(t, pos);
}
// x_pin(a, b, ...): every one of these values is in its registers HERE.  Loads the program is going to need at the same depth of its
// dependency chain are written next to each other and pinned together: one `s_waitcnt` for the group, one memory round trip.  Without
// it the compiler SINKS each load into the block of its first use — behind the branch that tests the previous load's result — and a
// drain becomes a chain of single loads each waited for alone (Q5's final loop: eleven in a row where the data dependencies need five).
template <class A> __device__ __forceinline__ void x_pin(A& a) { asm volatile("" : "+v"(a)); }
template <class A, class B> __device__ __forceinline__ void x_pin(A& a, B& b) { asm volatile("" : "+v"(a), "+v"(b)); }
template <class A, class B, class C> __device__ __forceinline__ void x_pin(A& a, B& b, C& c) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c)); }
template <class A, class B, class C, class D> __device__ __forceinline__ void x_pin(A& a, B& b, C& c, D& d) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }
template <class A, class B, class C, class D, class E> __device__ __forceinline__ void x_pin(A& a, B& b, C& c, D& d, E& e) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e)); }
template <class A, class B, class C, class D, class E, class F> __device__ __forceinline__ void x_pin(A& a, B& b, C& c, D& d, E& e, F& f) { asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f)); }
#ifndef X_PIN
#define X_PIN 1                                                       // 0: no pinning inside the lookups (A/B)
#endif

// The same with the table's LAYOUT known when the kernel is specialised (XL_* bits, x_layout_of on the host: which of the index's
// arrays exist is decided when the table is built, so it is a property of the plan at a given data size like the columns' encodings).
// x_lookup decides every one of these at run time from the DevTable's fields — wave-uniform branches, each in a basic block of its own
// with its own scalar loads of the argument block and its own `s_waitcnt`: the ISA of Q5's drain was a maze of ~200 branches in which
// the bitmap word, the rank prefix and the owner were requested one after the other.  Straight-line here: what a layout reads at one
// depth is requested together.  L without XL_KNOWN: x_lookup.
constexpr uint32_t XL_DENSE_ARR = 1u, XL_BM = 2u, XL_SHIFT = 4u, XL_BITMAP_ONLY = 8u, XL_LIN = 16u, XL_GRP = 32u, XL_DENSE_REF = 64u, XL_WEXC = 128u,
                   XL_SLOTS = 256u, XL_ALIAS = 512u, XL_KNOWN = 0x80000000u;
__host__ __device__ inline uint32_t x_layout_of(const DevTable& t, bool bitmap_only) {
    uint32_t l = XL_KNOWN;
    if (t.dense_arr) l |= XL_DENSE_ARR;
    if (t.bm) l |= XL_BM;
    if (t.bm_shift) l |= XL_SHIFT;
    if (bitmap_only || t.bitmap_only) l |= XL_BITMAP_ONLY;
    if (t.lin_rb) l |= XL_LIN;
    if (t.grp_first) l |= XL_GRP;
    if (t.dense_ref) l |= XL_DENSE_REF;
    if (t.wexc) l |= XL_WEXC;
    if (t.slots) l |= XL_SLOTS;
    if (t.alias) l |= XL_ALIAS;
    if ((l & XL_GRP) && !(l & XL_BM)) return 0u;                              // (never built: left to the run-time form)
    if ((l & XL_SHIFT) && t.bm_shift != 32) return 0u;
    return l;
}
template <uint32_t L> __device__ __forceinline__ int64_t xl_slot_key(const DevTable& t, uint64_t h) { if constexpr ((L & XL_SLOTS) != 0) return t.slots[h * 4]; else return t.keys[h]; }
template <uint32_t L> __device__ __forceinline__ uint32_t xl_slot_row(const DevTable& t, uint64_t h) { if constexpr ((L & XL_SLOTS) != 0) return (uint32_t)t.slots[h * 4 + 3]; else return t.rowref[h]; }
template <uint32_t L>
__device__ __forceinline__ uint32_t x_lookup_l(const DevTable& t, int64_t key, bool bad) {
    if constexpr ((L & XL_KNOWN) == 0) return x_lookup(t, key, bad);
    else {
        if (bad) return NO_ROW;
        if constexpr ((L & XL_DENSE_ARR) != 0) {
            if (key < t.bm_lo || key > t.bm_hi) return NO_ROW;
            return t.dense_arr[key - t.bm_lo];
        } else if constexpr ((L & XL_BM) != 0) {
            const int64_t v = (L & (XL_SHIFT | XL_LIN)) ? (int64_t)((uint64_t)key >> 32) : key;
            if (v < t.bm_lo || v > t.bm_hi) return NO_ROW;
            uint64_t off = (uint64_t)(v - t.bm_lo);
            if constexpr ((L & XL_LIN) != 0) {
                const int64_t b = (int64_t)((uint64_t)key & 0xFFFFFFFFull) - t.lin_b0;
                if (b < 0 || b >= t.lin_rb) return NO_ROW;
                off = off * (uint64_t)t.lin_rb + (uint64_t)b;
            }
            uint32_t word = t.bm[off >> 5];
            if constexpr ((L & XL_BITMAP_ONLY) != 0) return ((word >> (off & 31)) & 1u) ? 0u : NO_ROW;
            else if constexpr ((L & XL_GRP) != 0) {
                uint32_t pre = t.grp_first[off];
#if X_PIN
                x_pin(word, pre);
#endif
                const bool bit = (word >> (off & 31)) & 1u;
                if (!bit || pre == NO_ROW) return NO_ROW;
                int64_t p = (int64_t)pre;
                while (p < t.grp_cap) {                                       // (table_find: four entries of the run per round trip)
                    int64_t k[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) k[i] = t.grp_key[(p + i < t.grp_cap ? p + i : t.grp_cap - 1) * t.grp_kstride];
#if X_PIN
                    x_pin(k[0], k[1], k[2], k[3]);
#endif
                    bool jump = false;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (jump || p + i >= t.grp_cap) continue;
                        if (k[i] == key) return (uint32_t)(p + i);
                        if (k[i] == EMPTY_KEY) { p = ((p + i) / t.grp_seg_rows + 1) * t.grp_seg_rows; jump = true; continue; }
                        if (((uint64_t)k[i] >> 32) != ((uint64_t)key >> 32)) return NO_ROW;
                    }
                    if (!jump) p += 4;
                }
                return NO_ROW;
            } else if constexpr ((L & XL_SHIFT) == 0) {                      // direct: rank of the key's bit
                uint32_t pre = t.wprefix[off >> 5];
#if X_PIN
                x_pin(word, pre);
#endif
                const bool bit = (word >> (off & 31)) & 1u;
                if (!bit) return NO_ROW;
                const uint32_t below = __popc(word & ((1u << (off & 31)) - 1u));
                int64_t pos = (int64_t)pre + below;
                if constexpr ((L & XL_WEXC) != 0) {
                    if (pre & ROW_INDEX_EXC) {
                        const WordExc e = t.wexc[pre & ~ROW_INDEX_EXC];
                        pos = key < e.key0 ? (int64_t)e.pos_prev + below : (int64_t)e.pos0 + (below - e.n0);
                    }
                }
                if constexpr ((L & XL_DENSE_REF) != 0) return t.dense_ref[pos]; else return (uint32_t)pos;
            } else {                                                          // hash layout behind a bitmap over the key's high part
                uint64_t mask = t.hdr->cap_mask;
#if X_PIN
                x_pin(word, mask);
#endif
                const bool bit = (word >> (off & 31)) & 1u;
                if (!bit) return NO_ROW;
                uint64_t h = hash_key(key) & mask;
                int64_t k = xl_slot_key<L>(t, h);
                if (key == EMPTY_KEY) return xl_slot_row<L>(t, mask + 1);
                for (;;) {
                    if (k == key) return xl_slot_row<L>(t, h);
                    if (k == EMPTY_KEY) return NO_ROW;
                    h = (h + 1) & mask;
                    k = xl_slot_key<L>(t, h);
                }
            }
        } else {                                                              // hash layout, no bitmap
            if constexpr ((L & XL_BITMAP_ONLY) != 0) return x_lookup(t, key, bad);
            else {
                const uint64_t mask = t.hdr->cap_mask;
                if (key == EMPTY_KEY) return xl_slot_row<L>(t, mask + 1);
                uint64_t h = hash_key(key) & mask;
                int64_t k = xl_slot_key<L>(t, h);
                for (;;) {
                    if (k == key) return xl_slot_row<L>(t, h);
                    if (k == EMPTY_KEY) return NO_ROW;
                    h = (h + 1) & mask;
                    k = xl_slot_key<L>(t, h);
                }
            }
        }
    }
}
template <uint32_t L> __device__ __forceinline__ double x_acc_l(const DevTable& t, int k, uint32_t ent) {
    if (ent == NO_ROW) return 0.0;
    if constexpr ((L & XL_KNOWN) == 0) { if (t.alias) ent = t.alias[ent]; }
    else if constexpr ((L & XL_ALIAS) != 0) ent = t.alias[ent];
    return k < t.acc_stride ? t.sacc[(size_t)ent * t.acc_stride + k] : 0.0;
}
template <uint32_t L> __device__ __forceinline__ int64_t x_hits_l(const DevTable& t, uint32_t ent) {
    if (ent == NO_ROW) return 0;
    if constexpr ((L & XL_KNOWN) == 0) { if (t.alias) ent = t.alias[ent]; }
    else if constexpr ((L & XL_ALIAS) != 0) ent = t.alias[ent];
    return (int64_t)t.shits[ent];
}
// cheap necessary condition for `key in table`, usable on a streamed key before the row is queued: the
// table's key bitmap when it has one (exact for key sets and the direct layout; over the high part of a
// composite key), else true
__device__ __forceinline__ bool x_may_hit(const DevTable& t, int64_t part0, bool composite) {
    if (!t.bm) {
        // hash layout: its hashed filter answers "maybe" for every key it holds (and a few per cent of the others)
        if (!t.hf || composite) return true;
        const uint32_t code = hf_code(hf_raw(part0), hf_mask_of(t.hdr->cap_mask));
        return hf_test(t.hf[(code & HF_POS_MASK) >> 5], code);
    }
    if (composite ? (t.bm_shift == 0 && !t.lin_rb) : (t.bm_shift != 0 || t.lin_rb != 0)) return true;     // the bitmap is not over this part
    if (composite && t.lin_rb) return true;
    if (part0 < t.bm_lo || part0 > t.bm_hi) return false;
    const uint64_t off = (uint64_t)(part0 - t.bm_lo);
    return (t.bm[off >> 5] >> (off & 31)) & 1u;
}
// the bitmap x_may_hit would test `part0` against, or null when it would answer "maybe" for every key (wave-uniform)
__device__ __forceinline__ const uint32_t* x_prefilter_bitmap(const DevTable& t, bool composite) {
    if (!t.bm) return nullptr;                                          // (a hash-layout table's hashed filter is named by the PHASH form itself: P::sbitmap)
    if (composite ? (t.bm_shift == 0 && !t.lin_rb) : (t.bm_shift != 0 || t.lin_rb != 0)) return nullptr;
    if (composite && t.lin_rb) return nullptr;
    return t.bm;
}
__device__ __forceinline__ int64_t x_field(const DevTable& t, int f, uint32_t ent) { return ent == NO_ROW ? 0 : t.pay[f][ent]; }
__device__ __forceinline__ double x_acc(const DevTable& t, int k, uint32_t ent) {
    if (ent == NO_ROW) return 0.0;
    if (t.alias) ent = t.alias[ent];
    return k < t.acc_stride ? t.sacc[(size_t)ent * t.acc_stride + k] : 0.0;
}
__device__ __forceinline__ int64_t x_hits(const DevTable& t, uint32_t ent) {
    if (ent == NO_ROW) return 0;
    if (t.alias) ent = t.alias[ent];
    return (int64_t)t.shits[ent];
}

// =================================================================================================
// Sinks.  consume() is called by all 64 lanes of a wave together (pass = false for lanes without a
// row), so a sink may use ballots and wave scans.
// =================================================================================================

// ---- K-A: sums of NV doubles + a row count; one partial per workgroup, folded by k_sum_partials ----
template <int NV> struct XSum {
    static constexpr bool FINAL = true;                               // (the timing experiments of the queue skeleton touch final loops only: a build they emptied would change the loop under test)
    static constexpr bool PIPELINED = false;                          // x_tight: two tiles in flight per step (many waves per SIMD cover the latency)
    struct Args { double* partial; };
    double acc[NV > 0 ? NV : 1];
    int64_t cnt;
    __device__ __forceinline__ void init(const Args&) { for (int k = 0; k < (NV > 0 ? NV : 1); ++k) acc[k] = 0.0; cnt = 0; }
    __device__ __forceinline__ void consume(const XArgs&, const Args&, bool pass, int64_t, const XOut<NV>& o) {
#pragma unroll
        for (int k = 0; k < NV; ++k) acc[k] += pass ? x_f(o.val[k]) : 0.0;
        cnt += pass ? 1 : 0;
    }
    __device__ __forceinline__ void finish(const XArgs&, const Args& s) {           // workgroup-converged
        __shared__ double s_acc[TPB / WAVE][4];
        __shared__ int64_t s_cnt[TPB / WAVE];
        const int w = threadIdx.x / WAVE;
#pragma unroll
        for (int k = 0; k < NV; ++k) { double v = wave_sum(acc[k]); if (lane_id() == 0) s_acc[w][k] = v; }
        { int64_t c = wave_sum_i64(cnt); if (lane_id() == 0) s_cnt[w] = c; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double* out = s.partial + (size_t)blockIdx.x * 5;
#pragma unroll
            for (int k = 0; k < 4; ++k) { double v = 0.0; if (k < NV) for (int i = 0; i < TPB / WAVE; ++i) v += s_acc[i][k]; out[k] = v; }
            int64_t c = 0;
            for (int i = 0; i < TPB / WAVE; ++i) c += s_cnt[i];
            reinterpret_cast<int64_t*>(out)[4] = c;
        }
    }
};

// ---- K-C over a small group domain: LDS hash table per workgroup, f64 LDS atomics, slot-major partials ----
template <int NV> struct XGroup {
    static constexpr bool FINAL = true;
    static constexpr bool PIPELINED = false;
    struct Args { unsigned long long* gkeys; double* pacc; int64_t* pcnt; int* flags; };
    static constexpr int NVS = NV > 0 ? NV : 1;
    unsigned long long* s_keys; double (*s_acc)[NVS]; unsigned long long* s_cnt; int* s_map; int* s_flags;
    __device__ __forceinline__ void init(const Args& s) {
        // This workgroup's column of the partial COUNTS is zeroed now, while the chip is busy streaming: the merge adds a partial only
        // where its count is positive, so the epilogue stores nothing but the slots this workgroup has rows for (thread i zeroes and
        // later writes slot i: one thread's stores to one address stay in order).  Zeros for all LG_SLOTS slots and their four sums,
        // stored by every workgroup at its end, were 2.6 M scattered 8-byte stores behind the last row: 0.023 of Q5's 0.111 ms.
        for (int i = threadIdx.x; i < LG_SLOTS; i += TPB) s.pcnt[(size_t)i * gridDim.x + blockIdx.x] = 0;
        __shared__ unsigned long long sk[LG_SLOTS];
        __shared__ double sa[LG_SLOTS][NVS];
        __shared__ unsigned long long sc[LG_SLOTS];
        __shared__ int sm[LG_SLOTS];
        __shared__ int sf[1];
        s_keys = sk; s_acc = sa; s_cnt = sc; s_map = sm; s_flags = sf;
        for (int i = threadIdx.x; i < LG_SLOTS; i += TPB) { sk[i] = EMPTY_GROUP; sc[i] = 0; for (int k = 0; k < NVS; ++k) sa[i][k] = 0.0; sm[i] = -1; }
        if (threadIdx.x == 0) sf[0] = 0;
        __syncthreads();
    }
    __device__ __forceinline__ void consume(const XArgs&, const Args&, bool pass, int64_t, const XOut<NV>& o) {
        if (!pass) return;
#if X8_EXP == 11
        { asm volatile("s_nop 0"); return; }                                 // (timing experiment: rows reach the sink, which does nothing; no epilogue)
#endif
#if X8_EXP == 12
        { asm volatile("s_nop 0" :: "v"(o.key), "v"(o.val[0])); return; }    // (timing experiment: ... but needs the row's key and value)
#endif
        if (o.bad || o.key < 0) { atomicOr(&s_flags[0], 2); return; }
#if X8_EXP == 8 || X8_EXP == 9
        const int slot = (int)((unsigned long long)o.key & (LG_SLOTS - 1));      // (timing experiment: no hashing, no probing; 9: and no epilogue)
        s_keys[slot] = (unsigned long long)o.key;
#else
        const int slot = group_slot(s_keys, (unsigned long long)o.key, false);
#endif
        if (slot < 0) { atomicOr(&s_flags[0], 1); return; }
#if X8_EXP == 7
        if (slot != 100000) return;                                         // (timing experiment: the slot found, nothing added)
#endif
#pragma unroll
        for (int k = 0; k < NV; ++k) atomicAdd(&s_acc[slot][k], x_f(o.val[k]));
        atomicAdd(&s_cnt[slot], 1ull);
    }
    __device__ __forceinline__ void finish(const XArgs&, const Args& s) {
#if X8_EXP == 6 || X8_EXP == 9 || X8_EXP == 10 || X8_EXP == 11 || X8_EXP == 12 || X8_EXP == 14
        if (s.gkeys != nullptr) return;                                     // (timing experiment: no epilogue)
#endif
        __syncthreads();
        for (int i = threadIdx.x; i < LG_SLOTS; i += TPB) {
            if (s_keys[i] != EMPTY_GROUP && s_cnt[i] > 0) {
                const int gs = group_slot(s.gkeys, s_keys[i], true);
                if (gs < 0) atomicOr(&s_flags[0], 1); else s_map[gs] = i;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < LG_SLOTS; i += TPB) {
            const int l = s_map[i];
            if (l < 0) continue;                                              // (its count is zero since init)
            const size_t e = (size_t)i * gridDim.x + blockIdx.x;
#pragma unroll
            for (int k = 0; k < 4; ++k) s.pacc[e * 4 + k] = k < NV ? s_acc[l][k < NVS ? k : 0] : 0.0;
            s.pcnt[e] = (int64_t)s_cnt[l];
        }
        if (threadIdx.x == 0 && s_flags[0]) atomicOr(s.flags, s_flags[0]);
    }
};

// ---- K-B: survivors compacted in row order into the wave segment's slice of the stage ----
// Round 6, measured and left OFF (XS_LINES=1 turns it on): the survivors through a ring of XS_RING rows per wave in LDS, reaching the stage
// arrays XS_FLUSH rows at a time — whole 128-byte lines of every array — instead of what each consume() call's 64 lanes kept (1.5 rows
// per call in Q5's orders build, where 2.3 % of the rows survive).  The suspicion was that the L2 fills every line it is handed a part
// of: that build reads 243 MB (2 x FETCH_SIZE) for a 92 MB stream.  tools/build_bytes.sh took it apart (profiles/r06_build_bytes_*.txt):
// 92 MB with the survivors queued and never drained; 93 MB drained and EVALUATED with nothing reaching the sink — the looked-up customer
// entries are L2 hits, not the 217 MB the round-5 review asked about —; 243 MB with the sink.  With whole-line stores: 242 MB and the same
// 0.062 ms.  So it is not the stage arrays: it is the sink's INDEX part — one memory-side atomicOr per entry into a 7.5 MB bitmap and one
// 4-byte row-index note into a 7.5 MB array, 340 K entries scattered over 60 M key values, each a 64-byte read-modify-write at the memory
// side (and FETCH_SIZE's doubling, right for wide coalesced reads, counts those scattered requests twice: the true figure lies between
// 75 and 150 MB).  Positions, order and the index part of a store are what they were either way.
#ifndef XS_LINES
#define XS_LINES 0
#endif
// the INDEX part of a staged entry alone (stage_store's: the key's bit, the grouped layout's first row of a run)
__device__ __forceinline__ void stage_index_part(const DevStage& st, int64_t pos, int64_t key) {
    if (st.bm) {
        uint64_t off;
        if (bm_locate(st, key, off)) {
#if defined(XS_EXP) && (XS_EXP & 1)
            if (key == -12345)                                // (byte-accounting experiment, tools/build_bytes.sh: no bitmap atomics — results ARE wrong)
#endif
            atomicOr(&st.bm[off >> 5], 1u << (off & 31));
            if (st.grp_first) atomicMin(&st.grp_first[off], (uint32_t)pos);
        }
    }
}
template <int NV> struct XStage {
    static constexpr bool FINAL = false;
    static constexpr int XS_RING = 128, XS_FLUSH = 32;                    // (a flush leaves < 32 rows behind, a consume adds <= 64)
    struct Args { DevStage st; };
    int64_t out, seg_begin, flushed;
    uint32_t carry;                                                       // row index: the bitmap word of the wave's previous entry
    __device__ __forceinline__ void init(const Args&) { out = 0; seg_begin = 0; flushed = 0; carry = ROW_INDEX_NONE; }
    __device__ __forceinline__ void begin_segment(int64_t begin) { out = seg_begin = flushed = begin; carry = ROW_INDEX_NONE; }
    __device__ __forceinline__ static long long (*ring())[XS_RING] {
        __shared__ long long s_ring[TPB / WAVE][NV + 1][XS_RING];
        return s_ring[threadIdx.x / WAVE];
    }
    // rows [flushed, flushed + n) of the ring to the stage arrays: lane i the row flushed + i
    __device__ __forceinline__ void flush(const DevStage& st, int n) {
        long long (*buf)[XS_RING] = ring();
        const int lane = lane_id();
        if (lane < n) {
            const int64_t pos = flushed + lane;
            const int slot = (int)((pos - seg_begin) & (XS_RING - 1));
            const long long key = buf[0][slot];
            st.key[pos] = key;
#pragma unroll
            for (int q = 0; q < MAX_STAGE_COLS; ++q) if (q < st.npay) st.pay[q][pos] = q < NV ? buf[1 + (q < NV ? q : 0)][slot] : 0ll;
            if (st.shits) st.shits[pos] = 0;
            if (st.sacc) zero_acc(st, pos);
            if (st.grp_kp) { using V2 = long long __attribute__((ext_vector_type(2))); const V2 kp = {key, NV > 0 ? buf[1][slot] : 0ll}; reinterpret_cast<V2*>(st.grp_kp)[pos] = kp; }
        }
        flushed += n;
    }
    __device__ __forceinline__ void consume(const XArgs& a, const Args& s, bool pass, int64_t, const XOut<NV>& o) {
        bool keep = pass;
        if (pass && (o.bad || (a.key_lo <= a.key_hi && (o.key < a.key_lo || o.key > a.key_hi)))) { atomicOr(a.flags, 2); keep = false; }
        const uint64_t b = __ballot(keep);
        const int64_t pos = out + __popcll(b & lanemask_lt());
#if XS_LINES
        if (!b) return;                                                   // (wave-uniform; nothing kept: nothing noted — row_index_note returns at once for an empty mask)
        if (keep) {
            long long (*buf)[XS_RING] = ring();
            const int slot = (int)((pos - seg_begin) & (XS_RING - 1));
            buf[0][slot] = o.key;
#pragma unroll
            for (int k = 0; k < NV; ++k) buf[1 + k][slot] = o.val[k];
            stage_index_part(s.st, pos, o.key);
        }
#if defined(XS_EXP) && (XS_EXP & 2)
        if (o.key == -12345)                                              // (experiment: no row-index notes)
#endif
        row_index_note(s.st, (int)(seg_begin / s.st.seg_rows), keep, o.key, pos, b, carry);
        out += __popcll(b);
        __builtin_amdgcn_wave_barrier();
#if defined(XS_EXP) && (XS_EXP & 4)
        if (o.key == -12345)                                              // (experiment: nothing stored to the stage arrays)
#endif
        while (out - flushed >= XS_FLUSH) flush(s.st, XS_FLUSH);
#if defined(XS_EXP) && (XS_EXP & 4)
        if (out - flushed >= XS_FLUSH) flushed = out - (out - flushed) % XS_FLUSH;
#endif
        __builtin_amdgcn_wave_barrier();
#else
        if (keep) {
            int64_t pay[MAX_STAGE_COLS] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NV; ++k) pay[k] = o.val[k];
            stage_store<-1>(s.st, pos, o.key, pay);
        }
        row_index_note(s.st, (int)(seg_begin / s.st.seg_rows), keep, o.key, pos, b, carry);
        out += __popcll(b);
#endif
    }
    __device__ __forceinline__ void end_segment(const Args& s, int seg, int64_t begin) {
#if XS_LINES
        __builtin_amdgcn_wave_barrier();
        while (out > flushed) flush(s.st, out - flushed >= WAVE ? WAVE : (int)(out - flushed));
#endif
        if (lane_id() == 0) s.st.seg_count[seg] = (uint32_t)(out - begin);
    }
    __device__ __forceinline__ void finish(const XArgs&, const Args&) {}
};

// ---- membership-only K-B: OR the bits of the passing keys ----
template <int NV> struct XKeySet {
    static constexpr bool FINAL = false;
    struct Args { uint32_t* bm; };
    __device__ __forceinline__ void init(const Args&) {}
    __device__ __forceinline__ void consume(const XArgs& a, const Args& s, bool pass, int64_t, const XOut<NV>& o) {
        if (!pass) return;
        if (o.bad || o.key < a.key_lo || o.key > a.key_hi) { atomicOr(a.flags, 2); return; }
        const uint64_t off = (uint64_t)(o.key - a.key_lo);
        atomicOr(&s.bm[off >> 5], 1u << (off & 31));
    }
    __device__ __forceinline__ void finish(const XArgs&, const Args&) {}
};

// ---- K-C where the group is the matched entry: segmented wave scan over runs of equal entries, then
//      one native f64 atomic per run and value (rows of one group sit in adjacent lanes when the probe side is
//      clustered on the key) ----
template <int NV> struct XEntry {
    static constexpr bool FINAL = true;
    struct Args { DevTable tb; };
    __device__ __forceinline__ void init(const Args&) {}
    __device__ __forceinline__ void consume(const XArgs&, const Args& s, bool pass, int64_t, const XOut<NV>& o) {
        const int lane = lane_id();
        uint32_t idx = NO_ROW, cnt = 0;
        double v[NV > 0 ? NV : 1];
#pragma unroll
        for (int k = 0; k < (NV > 0 ? NV : 1); ++k) v[k] = 0.0;
        if (pass && o.ent != NO_ROW) {
            idx = o.ent; cnt = 1;
            if (s.tb.alias) idx = s.tb.alias[idx];
#pragma unroll
            for (int k = 0; k < NV; ++k) v[k] = x_f(o.val[k]);
        }
        if (!__ballot(idx != NO_ROW)) return;
        const uint32_t prev = __shfl_up(idx, 1, WAVE);
        uint32_t head = (lane == 0 || prev != idx) ? 1u : 0u;
        const uint32_t next_head = __shfl_down(head, 1, WAVE);
        const bool tail = idx != NO_ROW && (lane == WAVE - 1 || next_head != 0u);
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
            const uint32_t oc = __shfl_up(cnt, off, WAVE), oh = __shfl_up(head, off, WAVE);
            double ov[NV > 0 ? NV : 1];
#pragma unroll
            for (int k = 0; k < NV; ++k) ov[k] = __shfl_up(v[k], off, WAVE);
            if (lane >= off && !head) {
                cnt += oc;
#pragma unroll
                for (int k = 0; k < NV; ++k) v[k] += ov[k];
                head |= oh;
            }
        }
#if XE_EXP == 1
        if (tail && v[0] == 1.2345e300) atomicAdd(&s.tb.shits[idx], cnt);                   // (timing experiment: no atomics)
#elif XE_EXP == 2
        if (tail) {                                                                            // (timing experiment: sums only)
#pragma unroll
            for (int k = 0; k < NV; ++k) atomicAdd(&s.tb.sacc[(size_t)idx * s.tb.acc_stride + k], v[k]);
        }
#else
        if (tail) {
#pragma unroll
            for (int k = 0; k < NV; ++k) atomicAdd(&s.tb.sacc[(size_t)idx * s.tb.acc_stride + k], v[k]);
            atomicAdd(&s.tb.shits[idx], cnt);
        }
#endif
    }
    __device__ __forceinline__ void finish(const XArgs&, const Args&) {}
};

// =================================================================================================
// DIRECT: every column the program reads is streamed with 16-byte loads and the whole program is
// evaluated in registers on both rows of a lane's pair — for programs without lookups / string
// operations feeding order-free sinks (XSum, XGroup).  Persistent grid over 1024-row tiles, as k_scan_sum.
// P::NS streamed columns; P::sload<TAIL>(a, r, nrows, regs); P::eval_regs<H>(a, regs, r, out) -> pass.
// =================================================================================================
template <class P, template <int> class SinkT>
__device__ __forceinline__ void x_direct(const XArgs& a, const typename SinkT<P::NV>::Args& sa, int64_t nrows) {
    using Sink = SinkT<P::NV>;
    constexpr int NS = P::NS > 0 ? P::NS : 1;
    Sink sink;
    sink.init(sa);
    auto tile = [&](int64_t base, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        Pair<int64_t> s[UNROLL][NS];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            P::template sload<TAIL>(a, base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD, nrows, s[u]);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t r = base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
            XOut<P::NV> o0, o1;
            bool p0 = TAIL ? (r < nrows) : true, p1 = TAIL ? (r + 1 < nrows) : true;
            p0 = p0 && P::template eval_regs<0>(a, s[u], r, o0);
            p1 = p1 && P::template eval_regs<1>(a, s[u], r + 1, o1);
            sink.consume(a, sa, p0, r, o0);
            sink.consume(a, sa, p1, r + 1, o1);
        }
    };
    const int64_t full = nrows / TILE_ROWS;
    for (int64_t t0 = (int64_t)blockIdx.x * SDQH_TILE_CHUNK; t0 < full; t0 += (int64_t)gridDim.x * SDQH_TILE_CHUNK)
        for (int64_t t = t0; t < t0 + SDQH_TILE_CHUNK && t < full; ++t) tile(t * TILE_ROWS, XBool<false>{});
    if (full * TILE_ROWS < nrows && blockIdx.x == (unsigned)(full % gridDim.x)) tile(full * TILE_ROWS, XBool<true>{});
    sink.finish(a, sa);
}

// =================================================================================================
// QUEUE: one wave per contiguous row segment.  Per step of X_LB batches of 128 rows: the columns of
// the cheap leading conditions are streamed with 16-byte loads (all in flight before the first is
// tested), survivors are appended to the wave's LDS queue in row order, and whenever 64 are queued
// the FRONT 64 are evaluated, one row per lane, converged.  P::sload / P::stest(a, regs, p0, p1) for
// the streamed part (NS may be 0: every row is queued), P::eval_row(a, r, out) -> pass for the rest.
// =================================================================================================
constexpr int X_LB = 4;
constexpr int XQ_CAP = 64 + X_LB * 128;                               // 63 left over + a whole step's candidates, as 32-bit offsets from the segment's first row
constexpr int XSTR_UNITS = 2048;                                      // code units of text a wave stages in LDS at a time (8 KiB)

// The text operations of a drain, column J of the program's text columns: the fields of the drain's rows are copied into
// LDS by the whole wave (consecutive code units by consecutive lanes: coalesced, where a lane scanning its own field in
// global memory touches a different cache line per lane on every step), R rows at a time, and each lane then scans its
// own field there (P::sops<J>: every text operation of the program on that column, results into sres).  Q13's two
// firstIndex conditions over o_comment (316 bytes per row): 9.9 -> see DESIGN.md §3b.
template <class P, int J>
__device__ __forceinline__ void x_stage_text(const XArgs& a, const int32_t* q_row, int64_t begin, int first, int count, uint32_t* s_str,
                                             int64_t (&sres)[P::NSOP > 0 ? P::NSOP : 1]) {
    constexpr int W = P::swidth(J);
    const int lane = lane_id();
    if constexpr (P::sbytes(J)) {
        // The column's byte twin: a field starts at any byte, so the wave copies the aligned 32-bit words that cover the fields
        // and every lane scans its field at a byte offset of the staging window (lds8_scan).  Consecutive rows are one
        // contiguous run of bytes; otherwise each row gets a slot of WP bytes holding its covering words.
        constexpr int WP = (W + 6) & ~3;
        constexpr int RB = (XSTR_UNITS * 4 - 8) / WP;
        constexpr int R = RB < WAVE ? RB : WAVE;
        static_assert(R >= 1, "a text field wider than the staging window");
        const uint8_t* __restrict__ col = static_cast<const uint8_t*>(a.ncol[P::scol(J)]);
        for (int base = 0; base < count; base += R) {
            const int nr = count - base < R ? count - base : R;
            __builtin_amdgcn_wave_barrier();
            constexpr int NL = (R * WP / 4 + WAVE - 1) / WAVE;
            uint32_t t[NL];
            const int q0 = q_row[first + base], q1 = q_row[first + base + nr - 1];
            const bool run = q1 - q0 == nr - 1;
            int off = 0, nw;
            if (run) {
                const unsigned long long p0 = (unsigned long long)(col + (begin + (int64_t)q0) * W);
                const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(p0 & ~3ull);
                nw = ((int)(p0 & 3ull) + nr * W + 3) >> 2;
#pragma unroll
                for (int i = 0; i < NL; ++i) { const int idx = lane + i * WAVE; t[i] = idx < nw ? src[idx] : 0u; }
                off = (int)(p0 & 3ull) + (lane - base) * W;
            } else {
                nw = nr * (WP / 4);
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    const int idx = lane + i * WAVE;
                    t[i] = 0u;
                    if (idx < nw) {
                        const int row = idx / (WP / 4), wi = idx - row * (WP / 4);
                        const unsigned long long pr = (unsigned long long)(col + (begin + (int64_t)q_row[first + base + row]) * W);
                        t[i] = reinterpret_cast<const uint32_t*>(pr & ~3ull)[wi];
                    }
                }
                if (lane >= base && lane < base + nr) off = (lane - base) * WP + (int)(((unsigned long long)(col + (begin + (int64_t)q_row[first + lane]) * W)) & 3ull);
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) { const int idx = lane + i * WAVE; if (idx < nw) s_str[idx] = t[i]; }
            __builtin_amdgcn_wave_barrier();
            if (lane >= base && lane < base + nr) P::template sops<J>(a, s_str, off, sres);
            __builtin_amdgcn_wave_barrier();
        }
    } else {
    constexpr int R = (XSTR_UNITS / W) < WAVE ? (XSTR_UNITS / W) : WAVE;
    static_assert(R >= 1, "a text field wider than the staging window");
    const uint32_t* __restrict__ col = static_cast<const uint32_t*>(a.col[P::scol(J)]);
    for (int base = 0; base < count; base += R) {
        const int nr = count - base < R ? count - base : R;
        __builtin_amdgcn_wave_barrier();
        // every load of the round in flight before the first LDS store (a load -> store loop ran one round trip per 64 units)
        constexpr int NL = (R * W + WAVE - 1) / WAVE;                    // <= 32 code units per lane
        uint32_t t[NL];
        const int q0 = q_row[first + base], q1 = q_row[first + base + nr - 1];
        if (q1 - q0 == nr - 1) {                                         // consecutive rows (the usual case): one contiguous block
            const uint32_t* __restrict__ src = col + (begin + (int64_t)q0) * W;
#pragma unroll
            for (int i = 0; i < NL; ++i) { const int idx = lane + i * WAVE; t[i] = idx < nr * W ? src[idx] : 0u; }
        } else {
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const int idx = lane + i * WAVE;
                t[i] = 0u;
                if (idx < nr * W) { const int row = idx / W, u = idx - row * W; t[i] = col[(begin + (int64_t)q_row[first + base + row]) * W + u]; }
            }
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) { const int idx = lane + i * WAVE; if (idx < nr * W) s_str[idx] = t[i]; }
        __builtin_amdgcn_wave_barrier();
        if (lane >= base && lane < base + nr) P::template sops<J>(a, s_str, (lane - base) * W * 4, sres);
        __builtin_amdgcn_wave_barrier();
    }
    }
}

template <class P, template <int> class SinkT, bool SEGMENTED>
__device__ __forceinline__ void x_queue(const XArgs& a, const typename SinkT<P::NV>::Args& sa, int64_t nrows, int64_t seg_rows, int nseg) {
    using Sink = SinkT<P::NV>;
    constexpr int NS = P::NS > 0 ? P::NS : 1;
    __shared__ int32_t s_row[TPB / WAVE][XQ_CAP];
    __shared__ uint32_t s_text[P::NSC > 0 ? TPB / WAVE : 1][P::NSC > 0 ? XSTR_UNITS : 1];
    Sink sink;
    sink.init(sa);
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    const bool live = seg < nseg;
    int32_t* q_row = s_row[threadIdx.x / WAVE];
    uint32_t* s_str = s_text[P::NSC > 0 ? threadIdx.x / WAVE : 0];
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    constexpr int64_t BATCH_ROWS = WAVE * ROWS_PER_LOAD;
    const int64_t begin = (int64_t)seg * seg_rows;                     // (a segment is far shorter than 2^31 rows: the host checks)
    int64_t end = begin + seg_rows; if (end > nrows) end = nrows;
    if constexpr (SEGMENTED) sink.begin_segment(begin);
    int qn = 0;
    auto drain = [&](int first, int count) {                           // rows q_row[first .. first + count), one per lane, in row order
        XOut<P::NV> o;
        int64_t sres[P::NSOP > 0 ? P::NSOP : 1] = {0};
        if constexpr (P::NSC > 0) x_stage_text<P, 0>(a, q_row, begin, first, count, s_str, sres);
        if constexpr (P::NSC > 1) x_stage_text<P, 1>(a, q_row, begin, first, count, s_str, sres);
        bool pass = false; int64_t r = 0;
        if (lane < count) { r = begin + (int64_t)q_row[first + lane]; pass = P::eval_row(a, r, sres, o); }
        sink.consume(a, sa, pass, r, o);
    };
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
    if (live) {
        // One loop, one drain site (as k_build_lookup: the drain is most of the code): a step produces candidates, then
        // the FRONT full waves of the queue are drained in row order and what is left moves down.
        for (int64_t b = begin;;) {
            const bool last = b >= end;
            if (last) {
            } else if (b + BATCH_ROWS * X_LB <= end) {
                Pair<int64_t> s[X_LB][NS];
                int64_t rr[X_LB];
#pragma unroll
                for (int j = 0; j < X_LB; ++j) { rr[j] = b + (int64_t)j * BATCH_ROWS + (int64_t)lane * ROWS_PER_LOAD; P::template sload<false>(a, rr[j], nrows, s[j]); }
#pragma unroll
                for (int j = 0; j < X_LB; ++j) { bool p0 = true, p1 = true; P::stest(a, s[j], p0, p1); enqueue(rr[j], p0, p1); }
                b += BATCH_ROWS * X_LB;
            } else {
                const int64_t r = b + (int64_t)lane * ROWS_PER_LOAD;
                Pair<int64_t> s1[NS];
                P::template sload<true>(a, r, end, s1);
                bool p0 = r < end, p1 = r + 1 < end;
                P::stest(a, s1, p0, p1);
                enqueue(r, p0, p1);
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
                const int left = qn - head;                               // < 64
                int32_t keepv = 0;
                if (lane < left) keepv = q_row[head + lane];
                if (lane < left) q_row[lane] = keepv;
                qn = left;
            }
        }
        if constexpr (SEGMENTED) sink.end_segment(sa, seg, begin);
    }
    sink.finish(a, sa);
}


// =================================================================================================
// TIGHT: the DIRECT shape over columns at their tightest exact encoding — sorted-dictionary codes of 1 or 2
// bytes (sdqh_codes.hip), the 4-byte twins, or the 8-byte column itself — XT_R = 8 consecutive rows per lane
// and load step: a code column is one 8- or 16-byte load per lane, a 4-byte twin two 16-byte loads.  What the
// codes buy beyond bytes: a comparison of a coded column with a constant is ONE integer comparison of the code
// with the constant's rank (the host translates the constant at launch: XArgs::cc), and a coded column's value
// is one LDS read of its dictionary (P::ND tables of 256 entries, loaded once per workgroup) — no decode
// arithmetic.  Q1 streams 11 bytes per row (22 through 4-byte twins, 48 in the reference's layout), Q6 8.
// P::Regs holds the packed words of a lane's 8 rows; P::sload<TAIL>; P::eval(a, regs, tabs, i, r, out) -> pass
// evaluates row i (i is a constant after unrolling: every extraction is a fixed bit slice).
// Measured in tools/microbench_tight.hip before it was built: DESIGN.md §3c.
// =================================================================================================
constexpr int XT_R = 8;
constexpr int XT_ROWS = TPB * XT_R;                                    // rows per workgroup tile

// the 8 rows of a lane: BPR bytes per row -> BPR * 2 32-bit words
template <int BPR, bool TAIL>
__device__ __forceinline__ void xt_load(const void* __restrict__ p, int64_t r, int64_t nrows, uint32_t (&w)[BPR * 2]) {
    if constexpr (!TAIL) {
        const char* q = static_cast<const char*>(p) + r * BPR;
        if constexpr (BPR == 1) {
            using V = uint32_t __attribute__((ext_vector_type(2)));
            const V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(q));
            w[0] = t.x; w[1] = t.y;
        } else {
            using V = uint32_t __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int i = 0; i < BPR / 2; ++i) {
                const V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(q) + i);
                w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < BPR * 2; ++i) w[i] = 0u;
#pragma unroll
        for (int i = 0; i < XT_R; ++i) {
            const int64_t ri = r + i < nrows ? r + i : nrows - 1;
            if constexpr (BPR == 1) w[i / 4] |= (uint32_t)static_cast<const uint8_t*>(p)[ri] << (8 * (i % 4));
            else if constexpr (BPR == 2) w[i / 2] |= (uint32_t)static_cast<const uint16_t*>(p)[ri] << (16 * (i % 2));
            else if constexpr (BPR == 4) w[i] = static_cast<const uint32_t*>(p)[ri];
            else { const uint64_t v = static_cast<const uint64_t*>(p)[ri]; w[2 * i] = (uint32_t)v; w[2 * i + 1] = (uint32_t)(v >> 32); }
        }
    }
}
// delta twin: the 12-byte record of the lane's 8 rows (r is a multiple of 8 in every skeleton; the twin is padded to whole groups)
__device__ __forceinline__ void xt_load_d8(const void* __restrict__ p, int64_t r, uint32_t (&w)[3]) {
    using V = uint32_t __attribute__((ext_vector_type(3)));
    typedef V __attribute__((aligned(4))) UV;
    const UV t = *reinterpret_cast<const UV*>(static_cast<const char*>(p) + (r >> 3) * 12);
    w[0] = t.x; w[1] = t.y; w[2] = t.z;
}
__device__ __forceinline__ int32_t xt_d8(const uint32_t (&w)[3], int i) { return (int32_t)w[0] + (int32_t)((w[1 + i / 4] >> (8 * (i % 4))) & 0xFFu); }
__device__ __forceinline__ uint32_t xt_u8(const uint32_t (&w)[2], int i) { return (w[i / 4] >> (8 * (i % 4))) & 0xFFu; }
__device__ __forceinline__ uint32_t xt_u16(const uint32_t (&w)[4], int i) { return (w[i / 2] >> (16 * (i % 2))) & 0xFFFFu; }
__device__ __forceinline__ int32_t xt_i32(const uint32_t (&w)[8], int i) { return (int32_t)w[i]; }
__device__ __forceinline__ int64_t xt_i64(const uint32_t (&w)[16], int i) { return (int64_t)((uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32)); }

// ---- K-C over a DENSE small key range [key_lo, key_hi]: every lane owns its accumulators in LDS ----
// acc[(slot * (NV + 1) + k) * TPB + thread]: no two lanes share a cell, so ds_add_f64 (no return) is a plain
// read-modify-write in the LDS unit: ONE DS instruction per summed value and row, where accumulators in registers
// cost G x NV predicated adds per row (G groups unknown at compile time) and made the kernel ALU-bound once the
// bytes were halved (tools/microbench_tight.hip: 0.20-0.25 ms in registers, 0.118 ms this way, SF=10 Q1).
// Deterministic: a lane adds its rows in row order, lanes are folded in a fixed order, workgroups by
// k_groupby_merge in workgroup order.  The slots are the key's offsets, so no key table and no claiming.
template <int NV> struct XGroupLane {
    static constexpr bool FINAL = true;
    static constexpr bool PIPELINED = true;                           // x_tight: its LDS cells leave two or three waves per SIMD: a wave hides its own latency
    struct Args { unsigned long long* gkeys; double* pacc; int64_t* pcnt; int* flags; int32_t nslots, _pad; };
    // XGL_IVAL = v: summed value v is a small non-negative INTEGER on every row (a byte-coded column whose dictionary is consecutive
    // integers: l_quantity) and arrives as that integer.  It shares the row counter's 8-byte cell — rows in the low half, the value's
    // sum in the high half, ONE 64-bit integer add for both — instead of a cell and a double-precision add of its own: a cell less
    // per slot (Q1: 49 KiB of cells per workgroup instead of 61: three workgroups per CU instead of two), an LDS instruction less per
    // row, no dictionary read for it; and exact — a sum of integers below 2^53 is the same number in any order and either type (the
    // launcher checks that a lane's share of the rows cannot carry the high half past 2^31).
    static constexpr int IV = XGL_IVAL;
    static constexpr int NF = (NV > 0 ? NV : 0) - (IV >= 0 ? 1 : 0);  // double-precision cells per slot
    static constexpr int NA = NF + 1;                                 // + the counter cell
    int nslots; bool bad;
    // (cells are addressed as 32-bit indices into the dynamic LDS array: through a generic pointer kept in the object the compiler
    //  did the address arithmetic in 64 bits, a multiply included, per value and row)
    __device__ __forceinline__ void init(const Args& s) {
        extern __shared__ double x_dyn[];
        nslots = s.nslots; bad = false;
        for (int i = threadIdx.x; i < nslots * NA * TPB; i += TPB) x_dyn[i] = 0.0;
    }
    __device__ __forceinline__ void consume(const XArgs& a, const Args&, bool pass, int64_t, const XOut<NV>& o) {
        extern __shared__ double x_dyn[];
        const uint64_t d = (uint64_t)(o.key - a.key_lo);
        const bool ok = !o.bad && d < (uint64_t)nslots;
        bad = bad || (pass && !ok);
        if (pass && ok) {
            const uint32_t cell = (uint32_t)d * (uint32_t)(NA * TPB) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (k == IV) continue;
                const int f = (IV >= 0 && k > IV) ? k - 1 : k;
                __hip_atomic_fetch_add(&x_dyn[cell + (uint32_t)(f * TPB)], x_f(o.val[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if constexpr (IV >= 0)
                __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(&x_dyn[cell + (uint32_t)(NF * TPB)]), ((unsigned long long)(uint32_t)o.val[IV >= 0 ? IV : 0] << 32) | 1ull,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else    // (a lane's own row count: 32 bits are plenty, and a 4-byte LDS add costs the LDS half of an 8-byte one)
                __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(&x_dyn[cell + (uint32_t)(NF * TPB)]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __device__ __forceinline__ void finish(const XArgs& a, const Args& s) {
        __shared__ double s_red[TPB / WAVE][NA + 1];
        const int w = threadIdx.x / WAVE, lane = lane_id();
        if (__ballot(bad) && lane == 0) atomicOr(s.flags, 2);
        __syncthreads();
        for (int g = 0; g < nslots; ++g) {
#pragma unroll
            for (int k = 0; k < NA; ++k) {
                extern __shared__ double x_dyn[];
                double v = x_dyn[(uint32_t)((g * NA + k) * TPB) + threadIdx.x];
                if (k == NA - 1) {
                    const uint64_t bits = (uint64_t)__double_as_longlong(v);
                    const int64_t c = wave_sum_i64((int64_t)(uint32_t)bits);
                    if (lane == 0) reinterpret_cast<int64_t*>(s_red[w])[k] = c;
                    if constexpr (IV >= 0) { const int64_t iv = wave_sum_i64((int64_t)(bits >> 32)); if (lane == 0) reinterpret_cast<int64_t*>(s_red[w])[NA] = iv; }
                } else { v = wave_sum(v); if (lane == 0) s_red[w][k] = v; }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                const size_t e = (size_t)g * gridDim.x + blockIdx.x;
                int64_t c = 0, iv = 0;
                for (int i = 0; i < TPB / WAVE; ++i) { c += reinterpret_cast<const int64_t*>(s_red[i])[NA - 1]; if (IV >= 0) iv += reinterpret_cast<const int64_t*>(s_red[i])[NA]; }
                s.pcnt[e] = c;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double v = 0.0;
                    if (k < NV) {
                        if (k == IV) v = (double)iv;
                        else { const int f = (IV >= 0 && k > IV) ? k - 1 : k; for (int i = 0; i < TPB / WAVE; ++i) v += s_red[i][f < NA ? f : 0]; }
                    }
                    s.pacc[e * 4 + k] = v;
                }
                if (blockIdx.x == 0) s.gkeys[g] = (unsigned long long)(a.key_lo + g);
            }
            __syncthreads();
        }
    }
};

template <class P, template <int> class SinkT>
__device__ __forceinline__ void x_tight(const XArgs& a, const typename SinkT<P::NV>::Args& sa, int64_t nrows) {
    using Sink = SinkT<P::NV>;
    __shared__ int64_t s_tab[P::ND > 0 ? P::ND : 1][256];
    P::load_dicts(a, s_tab);
    Sink sink;
    sink.init(sa);
    __syncthreads();
    auto consume_tile = [&](const typename P::Regs& s, int64_t base, auto tail_tag) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const int64_t r = base + (int64_t)threadIdx.x * XT_R;
#pragma unroll
        for (int i = 0; i < XT_R; ++i) {
            XOut<P::NV> o;
            bool p = TAIL ? (r + i < nrows) : true;
            p = p && P::eval(a, s, s_tab, i, r + i, o);
            sink.consume(a, sa, p, r + i, o);
        }
    };
    // Whole tiles, software-pipelined: the next tile's loads are requested before this tile's rows are consumed, so a wave's own memory
    // latency is covered by its own arithmetic — it matters here because the per-lane accumulators leave room for two waves per SIMD
    // only (tools/microbench_tight.hip: 0.119 -> 0.109 ms for Q1's shape).  The request is unconditional (after the workgroup's last
    // tile: that tile again) so that no wait piles up in front of a branch.
    // A sink that leaves room for many waves per SIMD (sums in registers) takes TWO tiles per step with all their loads in flight and
    // lets other waves cover the latency instead (Q6: 0.0766 ms that way, 0.0795 pipelined).
    const int64_t full = nrows / XT_ROWS;
    typename P::Regs cur, nxt;
    if constexpr (Sink::PIPELINED) {
        int64_t t = blockIdx.x;
        if (t < full) P::template sload<false>(a, t * XT_ROWS + (int64_t)threadIdx.x * XT_R, nrows, nxt);
        for (; t < full; t += gridDim.x) {
            cur = nxt;
            const int64_t tn = t + gridDim.x < full ? t + gridDim.x : t;
            P::template sload<false>(a, tn * XT_ROWS + (int64_t)threadIdx.x * XT_R, nrows, nxt);
            consume_tile(cur, t * XT_ROWS, XBool<false>{});
        }
    } else {
        const int64_t pairs = full / 2;
        for (int64_t t = blockIdx.x; t < pairs; t += gridDim.x) {
            P::template sload<false>(a, (2 * t) * XT_ROWS + (int64_t)threadIdx.x * XT_R, nrows, cur);
            P::template sload<false>(a, (2 * t + 1) * XT_ROWS + (int64_t)threadIdx.x * XT_R, nrows, nxt);
            consume_tile(cur, (2 * t) * XT_ROWS, XBool<false>{});
            consume_tile(nxt, (2 * t + 1) * XT_ROWS, XBool<false>{});
        }
        if ((full & 1) && blockIdx.x == (unsigned)(pairs % gridDim.x)) {                  // an odd whole tile
            P::template sload<false>(a, (full - 1) * XT_ROWS + (int64_t)threadIdx.x * XT_R, nrows, cur);
            consume_tile(cur, (full - 1) * XT_ROWS, XBool<false>{});
        }
    }
    if (full * XT_ROWS < nrows && blockIdx.x == (unsigned)(full % gridDim.x)) {        // the ragged last tile
        P::template sload<true>(a, full * XT_ROWS + (int64_t)threadIdx.x * XT_R, nrows, cur);
        consume_tile(cur, full * XT_ROWS, XBool<true>{});
    }
    sink.finish(a, sa);
}


// =================================================================================================
// QUEUE on tight encodings: the QUEUE shape with the streamed part of TIGHT — one wave per contiguous row
// segment, but a lane takes XT_R = 8 consecutive rows per step out of packed words (dictionary codes, 4-byte
// twins), tests the cheap leading conditions on them in code space (P::stest(a, regs, tab, i)) and appends its
// survivors to the wave's LDS queue IN ROW ORDER (lane-major: an exclusive prefix sum of the lanes' survivor
// counts); the drain (text staging, P::eval_row, the sinks) is QUEUE's own.  X8_U steps' loads are in flight
// before the first is tested.  Q3's probe streams 6 bytes per lineitem row this way (key through its 4-byte
// twin, date as a 2-byte code) where the two-rows-per-lane form read 8 and issued four times the loads.
// =================================================================================================
#ifndef X8_U_STEPS
#define X8_U_STEPS 2
#endif
constexpr int X8_U = X8_U_STEPS;                                       // 512-row steps whose loads are in flight together
constexpr int X8_STEP = WAVE * XT_R;                                  // 512 rows per wave step
constexpr int X8_CAP = 64 + X8_U * X8_STEP;
// four consecutive bitmap words from any word on (4-byte aligned: gfx950 serves a dwordx4 global load at dword alignment)
using XWindow = uint32_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ XWindow x_load_window(const uint32_t* p) {
    typedef XWindow __attribute__((aligned(4))) UW;                  // one global_load_dwordx4 (checked in the ISA), tracked by the compiler's waitcnt pass
    return *reinterpret_cast<const UW*>(p);
}

__device__ __forceinline__ int wave_excl_prefix(int v, int& total) {  // exclusive prefix sum over the 64 lanes; total = the wave's sum
    int incl = v;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) { const int o = __shfl_up(incl, off, WAVE); if (lane_id() >= off) incl += o; }
    total = __shfl(incl, WAVE - 1, WAVE);
    return incl - v;
}
// The same for counts of at most 15 (a lane's survivors among its 8 rows), without a cross-lane dependency chain: bit k of the
// counts as a ballot, lanes below counted by v_mbcnt — four ballots, eight mbcnt, no LDS-crossbar round trips (the shuffle form is six
// dependent ds_bpermute rounds per call, twice per 1024-row step: 8 us of Q3's 86 us probe went into queueing ~5 survivors per step).
__device__ __forceinline__ int wave_excl_prefix4(uint32_t v, int& total) {
    int below = 0, all = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint64_t b = __ballot((v >> k) & 1u);
        below += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u)) << k;
        all += __popcll(b) << k;
    }
    total = all;
    return below;
}

template <class Sink> __device__ __forceinline__ constexpr bool x_exp_on() { return X8_EXP_BUILD ? !Sink::FINAL : Sink::FINAL; }
template <class P, template <int> class SinkT, bool SEGMENTED>
__device__ __forceinline__ void x_queue8(const XArgs& a, const typename SinkT<P::NV>::Args& sa, int64_t nrows, int64_t seg_rows, int nseg) {
    using Sink = SinkT<P::NV>;
    __shared__ int32_t s_row[TPB / WAVE][X8_CAP];
    __shared__ uint32_t s_text[P::NSC > 0 ? TPB / WAVE : 1][P::NSC > 0 ? XSTR_UNITS : 1];
    __shared__ int64_t s_tab[P::ND > 0 ? P::ND : 1][256];
    P::load_dicts(a, s_tab);
    Sink sink;
    sink.init(sa);
    __syncthreads();
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    const bool live = seg < nseg;
    int32_t* q_row = s_row[threadIdx.x / WAVE];
    uint32_t* s_str = s_text[P::NSC > 0 ? threadIdx.x / WAVE : 0];
    const int lane = lane_id();
    // TILED walk (seg_rows == 0: the host asks for it for order-free sinks only): wave `seg` of `nseg` takes the 1024-row double steps
    // seg, seg + nseg, seg + 2 nseg ... — at any moment the chip's waves read NEIGHBOURING kilobytes, as the register skeletons' tile
    // walk does, instead of nseg separate streams that each advance through a slice of their own (Q5's final loop streams ONE 4-byte
    // column: 3.7 TB/s segment by segment).  Queue entries are then row numbers (begin = 0: the host checks nrows < 2^31); the rows
    // behind the last whole double step are wave 0's, walked in a second phase by the same loop.
    constexpr int64_t STEP2 = (int64_t)X8_STEP * X8_U;
    const bool tiled = !SEGMENTED && seg_rows == 0;
    const int64_t whole_end = tiled ? nrows / STEP2 * STEP2 : 0;
    const int64_t stride = tiled ? (int64_t)nseg * STEP2 : STEP2;
    int phase = tiled ? 1 : 0;
    const int64_t begin = tiled ? 0 : (int64_t)seg * seg_rows;        // (a segment is far shorter than 2^31 rows: the host checks)
    int64_t end = tiled ? whole_end : begin + seg_rows; if (end > nrows) end = nrows;
    if constexpr (SEGMENTED) sink.begin_segment(begin);
    int qn = 0;
    const uint32_t* pbm = P::sbitmap(a);                               // the prefilter's key bitmap, or null (wave-uniform)
    uint32_t hmask = 0;                                                // P::PHASH: the bitmap is the table's HASHED filter, this many bits - 1 (sized on the device with the table)
    if constexpr (x_is_phash<P>::value) hmask = hf_mask_of(P::shdr(a)->cap_mask);
    auto drain = [&](int first, int count) {                           // rows q_row[first .. first + count), one per lane, in row order
        XOut<P::NV> o;
        int64_t sres[P::NSOP > 0 ? P::NSOP : 1] = {0};
        if constexpr (P::NSC > 0) x_stage_text<P, 0>(a, q_row, begin, first, count, s_str, sres);
        if constexpr (P::NSC > 1) x_stage_text<P, 1>(a, q_row, begin, first, count, s_str, sres);
        bool pass = false; int64_t r = 0;
        if (lane < count) { r = begin + (int64_t)q_row[first + lane]; pass = P::eval_row(a, r, sres, o); }
        if constexpr (X8_EXP == 4 && x_exp_on<Sink>()) {                                                                                                        // (timing experiment: every drained row evaluated twice)
            if (lane < count) { XOut<P::NV> o2; const bool p2 = P::eval_row(a, r ^ 1, sres, o2); pass = pass && (p2 || (uint32_t)a.key_hi != 0x12345678u); }
        }
        if constexpr (X8_EXP == 14 && x_exp_on<Sink>()) pass = pass && (o.key == -12345 || nrows < (int64_t)X8_EXP_MINROWS);                                                                         // (no row reaches the sink, by its data)
        if constexpr ((X8_EXP == 5 || X8_EXP == 10) && x_exp_on<Sink>()) { int pi = pass ? 1 : 0; asm volatile("" : "+v"(pi)); pass = pi != 0 && (uint32_t)a.key_hi == 0x12345678u; }   // (evaluated, nothing reaches the sink; 10: and no epilogue)
        sink.consume(a, sa, pass, r, o);
    };
    auto enqueue8 = [&](int64_t r0, uint32_t m) {                      // m: bit i = row r0 + i of this lane survives
        if (!__ballot(m != 0)) return;
        int total;
        int at = qn + wave_excl_prefix4(__popc(m), total);
        const int32_t off = (int32_t)(r0 - begin);
        while (m) { q_row[at++] = off + (__ffs((int)m) - 1); m &= m - 1u; }        // as many rounds as the fullest lane has survivors (sparse: one or two)
        qn += total;
    };
    // DRIVEN walk (round 5).  A loop whose first lookup is keyed by a column the scanned table is STORED IN THE ORDER OF (l_orderkey) and
    // whose table holds few of that column's values (Q5: the orders of one year placed by one region's customers, 0.8 % of the keys)
    // streams 60 M keys to find the 3 % of the rows that can hit.  The table's key bitmap says which keys exist, the column's run index
    // (XArgs::run_index) where each key's rows are: the wave walks the BITMAP instead — 64 quads of four words per step, every set bit a
    // key, its rows run_index[key] .. run_index[key + 1) — and the rows go through the same queue into the same drain; nothing is
    // streamed.  Chosen by every wave alike, from the same 64-quad sample of the bitmap's density: (estimated keys) x driven_ratio <=
    // rows; order-free sinks only (the tiled walk's condition).  Measured on Q5's final loop at SF=10 (0.105 ms streamed): runs ended by
    // comparing the column's keys 0.081; the rows listed by a launch of their own in front (one returning atomic per wave step on ONE
    // address, 29 K of them ~9 ns apart) 0.4.
    bool driven = false;
    uint64_t d_nq = 0, d_q = 0, d_nwords = 0;
    XWindow d_w = {0u, 0u, 0u, 0u};
    uint32_t d_bit0 = 0, d_cur = 0, d_end = 0;
    auto d_quad = [&](uint64_t q) {                                      // quad q of the bitmap; words behind its end are zero (and not read)
        const uint64_t wb = q * 4;
        if (wb + 4 <= d_nwords) return x_load_window(pbm + wb);
        XWindow v = {0u, 0u, 0u, 0u};
        if (wb < d_nwords) v.x = pbm[wb];
        if (wb + 1 < d_nwords) v.y = pbm[wb + 1];
        if (wb + 2 < d_nwords) v.z = pbm[wb + 2];
        return v;
    };
    if constexpr (x_is_driven<P>::value) {
        if (live && tiled && pbm && a.run_index && a.driven_ratio > 0) {
            const DevTable& t = P::dtab(a);
            d_nwords = ((uint64_t)(t.bm_hi - t.bm_lo) + 32) >> 5;
            d_nq = (d_nwords + 3) >> 2;
            const uint64_t ns = d_nq < (uint64_t)WAVE ? d_nq : (uint64_t)WAVE;
            int bits = 0, tot = 0;
            if ((uint64_t)lane < ns) { const XWindow v = d_quad((uint64_t)lane * (d_nq / ns)); bits = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w); }
            (void)wave_excl_prefix(bits, tot);
            const double est = ns ? (double)tot * ((double)d_nq / (double)ns) : 0.0;
            driven = est * (double)a.driven_ratio <= (double)nrows;
            d_q = (uint64_t)seg * WAVE;
        }
    }
    if (live) {
#if X8_PIPE
        typename P::Regs pre[X8_U];
        {
            const int64_t b0 = tiled ? (int64_t)seg * STEP2 : begin;
            if (b0 + STEP2 <= end) {
#pragma unroll
                for (int u = 0; u < X8_U; ++u) P::template sload<false>(a, b0 + (int64_t)u * X8_STEP + (int64_t)lane * XT_R, nrows, pre[u]);
            }
        }
#endif
        for (int64_t b = tiled ? (int64_t)seg * STEP2 : begin;;) {
            bool last = false, produced = false;
            if constexpr (x_is_driven<P>::value) {
                if (driven) {
                    produced = true;
                    if (!__ballot((d_w.x | d_w.y | d_w.z | d_w.w) != 0u || d_cur < d_end)) {      // nothing in hand: the wave's next 64 quads, or the end
                        // (steps are dealt out.  Claimed in order with a returning atomic on ONE address, to even out their varying row counts: every
                        //  step 0.073 -> 0.167 ms — all waves' first claims at the kernel's start, served one after the other; every step but a
                        //  wave's first 0.128)
                        if (d_q >= d_nq) last = true;
                        else {
                            const uint64_t q = d_q + (uint64_t)lane;
                            d_w = q < d_nq ? d_quad(q) : XWindow{0u, 0u, 0u, 0u};
                            d_bit0 = (uint32_t)(q * 128);
                            d_q += (uint64_t)nseg * WAVE;
                        }
                    }
                    if (!last) {
                        // a round: every lane goes on with the run it is in, or opens the run of its next key; up to 8 rows of it are queued
                        if (d_cur >= d_end && (d_w.x | d_w.y | d_w.z | d_w.w) != 0u) {
                            const int j = d_w.x ? 0 : d_w.y ? 1 : d_w.z ? 2 : 3;
                            const uint32_t word = j == 0 ? d_w.x : j == 1 ? d_w.y : j == 2 ? d_w.z : d_w.w;
                            const uint32_t bp = (uint32_t)__ffs((int)word) - 1u, rest = word & (word - 1u);
                            if (j == 0) d_w.x = rest; else if (j == 1) d_w.y = rest; else if (j == 2) d_w.z = rest; else d_w.w = rest;
                            const int64_t key = P::dtab(a).bm_lo + (int64_t)(d_bit0 + (uint32_t)j * 32u + bp);
                            if (key >= a.run_lo && key <= a.run_hi) { d_cur = a.run_index[key - a.run_lo]; d_end = a.run_index[key - a.run_lo + 1]; }
                        }
                        const uint32_t n = d_end - d_cur < 8u ? d_end - d_cur : 8u;
                        const int64_t r0 = (int64_t)d_cur;
                        d_cur += n;
                        enqueue8(r0, (1u << n) - 1u);
                    }
                }
            }
            if (!produced) {
            if (phase == 1 && b >= end) {                                     // tiled: the whole double steps are done; wave 0 takes the rest of the rows
                phase = 2;
                if (seg == 0) { b = whole_end; end = nrows; }
            }
            last = b >= end;
            if (last) {
            } else if (b + (int64_t)X8_STEP * X8_U <= end) {
                // The NEXT double step's streamed loads are requested behind this step's bitmap words and before anything waits
                // (X8_PIPE): `s_waitcnt vmcnt` counts in issue order, so requested BEFORE the words (round 3's attempt: Q3's probe
                // 0.103 -> 0.110 ms) they have to land before the words can be tested and nothing overlaps; behind them they stay in
                // flight through the tests, the queueing and the drains of this step.  Unconditional (behind the segment's last double
                // step: that step again), so no wait piles up in front of a branch.
                typename P::Regs s[X8_U];
#if X8_PIPE
#pragma unroll
                for (int u = 0; u < X8_U; ++u) s[u] = pre[u];
#else
#pragma unroll
                for (int u = 0; u < X8_U; ++u) P::template sload<false>(a, b + (int64_t)u * X8_STEP + (int64_t)lane * XT_R, nrows, s[u]);
#endif
                uint32_t m[X8_U], off[X8_U][XT_R], w[X8_U][XT_R];
                uint32_t wb[X8_U]; XWindow win[X8_U];                           // (P::PWIN: the lane's window of the bitmap)
                if constexpr (P::PREF32) {
                    // 32-bit form (the key and the bitmap's range fit 32 bits — known when the kernel was specialised): a row costs
                    // a subtract, an unsigned compare, a select, a shift and a bit-field extract.  The 64-bit form below spent ~30
                    // vector instructions per row and left the kernel issue-bound at half of what its bytes allow (PMC, Q3's probe)
#pragma unroll
                    for (int u = 0; u < X8_U; ++u) {
                        m[u] = 0;
#pragma unroll
                        for (int i = 0; i < XT_R; ++i) { const bool p = P::spre32(a, s[u], s_tab, i, off[u][i]); m[u] |= p ? (1u << i) : 0u; }
                    }
                    if constexpr (X8_EXP == 1 && x_exp_on<Sink>()) {
                        // (timing experiment: no bitmap requests at all)
                    } else if constexpr (x_is_phash<P>::value) {
                        // HASHED filter of a hash-layout table: spre32 gave 32 hash bits of the key; the key's word is one request, its two
                        // bits one test — the keys that pass (the table's, and a few per cent of the others) are queued for the slots
#pragma unroll
                        for (int u = 0; u < X8_U; ++u)
#pragma unroll
                            for (int i = 0; i < XT_R; ++i) { off[u][i] = hf_code(off[u][i], hmask); w[u][i] = pbm[((m[u] >> i) & 1u) ? (off[u][i] & HF_POS_MASK) >> 5 : 0u]; }
                    } else if constexpr (P::PWIN) {
                        // WINDOW: a lane's 8 consecutive rows carry near-by keys when the key column is clustered (a foreign key of a table
                        // stored in the order of its parent: l_orderkey; column_span8 sampled it), so ONE 16-byte request per lane — the
                        // four bitmap words from the word of its smallest passing key on — answers all eight tests; a row whose key lies
                        // beyond those 128 bits asks for its own word afterwards (rare).  Every vector-memory instruction occupies the CU's
                        // address path for its 64 lanes whether or not they share a line: eight 4-byte requests per 8 rows cost Q3's probe
                        // 21 us of 96 (36 M cache accesses), the window 8.
#pragma unroll
                        for (int u = 0; u < X8_U; ++u) {
                            uint32_t lo = 0xFFFFFFFFu;
#pragma unroll
                            for (int i = 0; i < XT_R; ++i) { const uint32_t o = ((m[u] >> i) & 1u) ? off[u][i] : 0xFFFFFFFFu; lo = o < lo ? o : lo; }
                            wb[u] = m[u] ? lo >> 5 : 0u;
                            win[u] = x_load_window(pbm + wb[u]);
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < X8_U; ++u)
#pragma unroll
                            for (int i = 0; i < XT_R; ++i) w[u][i] = pbm[off[u][i] >> 5];
                    }
                } else if (pbm) {
                    // the prefilter's bitmap words of all 16 rows are requested before any is tested (a load inside each row's own
                    // `if (passes the cheap conditions)` region made a step sixteen dependent round trips)
#pragma unroll
                    for (int u = 0; u < X8_U; ++u) {
                        m[u] = 0;
#pragma unroll
                        for (int i = 0; i < XT_R; ++i) { uint32_t bit; const bool p = P::spre(a, s[u], s_tab, i, off[u][i], bit); m[u] |= p ? (1u << i) : 0u; off[u][i] |= bit << 27; }
                    }
#pragma unroll
                    for (int u = 0; u < X8_U; ++u)
#pragma unroll
                        for (int i = 0; i < XT_R; ++i) w[u][i] = pbm[off[u][i] & 0x07FFFFFFu];
                } else {
#pragma unroll
                    for (int u = 0; u < X8_U; ++u) {
                        m[u] = 0;
#pragma unroll
                        for (int i = 0; i < XT_R; ++i) m[u] |= P::stest(a, s[u], s_tab, i) ? (1u << i) : 0u;
                    }
                }
#if X8_PIPE
                {
                    __builtin_amdgcn_sched_barrier(0);
                    const int64_t adv = phase == 1 ? stride : STEP2;                    // (tiled: this wave's next double step is a stride away)
                    const int64_t bn = b + adv + STEP2 <= end ? b + adv : b;
#pragma unroll
                    for (int u = 0; u < X8_U; ++u) P::template sload<false>(a, bn + (int64_t)u * X8_STEP + (int64_t)lane * XT_R, nrows, pre[u]);
                    __builtin_amdgcn_sched_barrier(0);
                }
#endif
                if constexpr (P::PREF32) {
                    if constexpr (X8_EXP == 1 && x_exp_on<Sink>()) {
#pragma unroll
                        for (int u = 0; u < X8_U; ++u) m[u] = ((uint32_t)a.key_hi == 0x12345678u) ? m[u] : 0u;
                    } else if constexpr (x_is_phash<P>::value) {
#pragma unroll
                        for (int u = 0; u < X8_U; ++u) {
                            uint32_t hit = 0;
#pragma unroll
                            for (int i = 0; i < XT_R; ++i) hit |= (hf_test(w[u][i], off[u][i]) ? 1u : 0u) << i;
                            m[u] &= hit;
                        }
                    } else if constexpr (P::PWIN) {
#pragma unroll
                        for (int u = 0; u < X8_U; ++u) {
                            uint32_t hit = 0, beyond = 0;
#pragma unroll
                            for (int i = 0; i < XT_R; ++i) {
                                const uint32_t rel = off[u][i] - (wb[u] << 5);
                                const uint32_t k = rel >> 5;
                                const uint32_t word = k == 0 ? win[u].x : k == 1 ? win[u].y : k == 2 ? win[u].z : win[u].w;
                                hit |= (k < 4u ? __builtin_amdgcn_ubfe(word, rel & 31u, 1u) : 0u) << i;
                                beyond |= (k < 4u ? 0u : 1u) << i;
                            }
                            beyond &= m[u];
                            if (__ballot(beyond != 0)) {                                   // keys beyond the lane's window: their own words (wave-uniform branch per row slot)
#pragma unroll
                                for (int i = 0; i < XT_R; ++i) {
                                    const bool far = (beyond >> i) & 1u;
                                    if (__ballot(far)) { const uint32_t wd = pbm[far ? off[u][i] >> 5 : 0u]; hit |= (far ? __builtin_amdgcn_ubfe(wd, off[u][i] & 31u, 1u) : 0u) << i; }
                                }
                            }
                            m[u] &= hit;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < X8_U; ++u) {
                            uint32_t hit = 0;
#pragma unroll
                            for (int i = 0; i < XT_R; ++i) hit |= __builtin_amdgcn_ubfe(w[u][i], off[u][i] & 31u, 1u) << i;
                            m[u] &= hit;
                        }
                    }
                } else if (pbm) {
#pragma unroll
                    for (int u = 0; u < X8_U; ++u)
#pragma unroll
                        for (int i = 0; i < XT_R; ++i) m[u] &= ~(((~w[u][i] >> (off[u][i] >> 27)) & 1u) << i);
                }
                if constexpr (X8_EXP == 2 && x_exp_on<Sink>()) {
#pragma unroll
                    for (int u = 0; u < X8_U; ++u) m[u] = ((uint32_t)a.key_hi == 0x12345678u) ? m[u] : 0u;      // (timing experiment: tests done, nothing queued)
                }
#pragma unroll
                for (int u = 0; u < X8_U; ++u) enqueue8(b + (int64_t)u * X8_STEP + (int64_t)lane * XT_R, m[u]);
                b += phase == 1 ? stride : (int64_t)X8_STEP * X8_U;
            } else if (b + X8_STEP <= end) {                                  // one whole step left
                typename P::Regs s1;
                const int64_t r0 = b + (int64_t)lane * XT_R;
                P::template sload<false>(a, r0, nrows, s1);
                uint32_t m = 0;
#pragma unroll
                for (int i = 0; i < XT_R; ++i) m |= P::stest(a, s1, s_tab, i) ? (1u << i) : 0u;
                enqueue8(r0, m);
                b += X8_STEP;
            } else {
                const int64_t r0 = b + (int64_t)lane * XT_R;
                typename P::Regs s1;
                P::template sload<true>(a, r0, end, s1);
                uint32_t m = 0;
#pragma unroll
                for (int i = 0; i < XT_R; ++i) m |= (r0 + i < end && P::stest(a, s1, s_tab, i)) ? (1u << i) : 0u;
                enqueue8(r0, m);
                b += X8_STEP;
            }
            }   // !produced
            int head = 0;
            if constexpr (X8_EXP == 3 && x_exp_on<Sink>()) { if ((uint32_t)a.key_hi != 0x12345678u && nrows >= (int64_t)X8_EXP_MINROWS) qn = 0; }      // (timing experiment: survivors queued, never drained)
            while (qn - head >= X8_DRAIN_AT || (last && qn > head)) {
                const int n = qn - head >= WAVE ? WAVE : qn - head;
                drain(head, n);
                head += n;
            }
            if (last) break;
            if (head) {
                const int left = qn - head;                               // < 64
                int32_t keepv = 0;
                if (lane < left) keepv = q_row[head + lane];
                if (lane < left) q_row[lane] = keepv;
                qn = left;
            }
        }
        if constexpr (SEGMENTED) sink.end_segment(sa, seg, begin);
    }
    sink.finish(a, sa);
}



// =================================================================================================
// STAGE with a queue of VALUES (x_vstage8): a unique build whose every condition can be decided on the streamed registers —
// comparisons, arithmetic, and `tbl[k] != None` tests against tables that answer from an EXACT key bitmap over a 32-bit range —
// and whose key and payload are values of the scanned row.  A lane evaluates its 8 rows where they are (the bitmap words of a
// step's rows are requested together), and the survivors' KEY and PAYLOAD WORDS — not their row numbers — go to the wave's LDS
// queue at prefix positions (row order: lane-major); every full group of 64 is then stored to the stage by the whole wave.  Against
// the row-number queue: no second lookup and no gathers by row (Q3's orders build fetched most lines of three columns for a tenth
// of their values: 529 MB for 165 MB of streamed columns); against storing from the rows' own lanes (tried in round 3: every store
// instruction ran with a tenth of its lanes, 0.150 ms against 0.128): the stores are whole.
// P::gates / P::lkoff / P::lkbm / P::row; NL <= 2 lookups, NV <= 2 payload fields (the queue is (1 + NV) x 8 bytes x 576 per wave).
// =================================================================================================
constexpr int XV_CAP = 64 + X8_STEP;
template <bool Q32> struct XQueueWord { using type = int64_t; };
template <> struct XQueueWord<true> { using type = int32_t; };
template <class P>
__device__ __forceinline__ void x_vstage8(const XArgs& a, const typename XStage<P::NV>::Args& sa, int64_t nrows, int64_t seg_rows, int nseg) {
    constexpr int NQ = 1 + (P::NV > 0 ? P::NV : 0);
    // P::Q32: key and payload are integers that fit 32 bits (known from the columns' ranges when the kernel is specialised): the queue
    // holds 4-byte words — 27 KiB per workgroup instead of 55, five resident workgroups per CU instead of two (the kernel waits on memory
    // half of its cycles: PMC, Q3's orders build at 8 waves per CU took 2.4 rounds of waves)
    using QT = typename XQueueWord<P::Q32>::type;
    __shared__ QT s_q[TPB / WAVE][NQ][XV_CAP];
    __shared__ int64_t s_tab[P::ND > 0 ? P::ND : 1][256];
    P::load_dicts(a, s_tab);
    __syncthreads();
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= nseg) return;
    const int lane = lane_id();
    const DevStage& st = sa.st;
    QT (*q)[XV_CAP] = s_q[threadIdx.x / WAVE];
    const int64_t begin = (int64_t)seg * seg_rows;
    int64_t end = begin + seg_rows; if (end > nrows) end = nrows;
    int64_t out = begin;
    int qn = 0;
    constexpr int NL = P::NL > 0 ? P::NL : 1;
    const uint32_t* bm[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) bm[l] = P::NL > l ? P::lkbm(a, l) : nullptr;
    uint32_t carry = ROW_INDEX_NONE;                                          // row index: the bitmap word of the wave's previous entry
    // XV_BITS: a build keyed by a strictly increasing column (st.wrow: the row-index layout exists for nothing else) meets its survivors in
    // increasing key order, and the keys of another wave's rows lie outside [the wave's first row's key, its last row's key]: every bitmap
    // word strictly between the first and the last word the wave touches is the wave's alone.  The wave ORs its bits into a window of
    // XB_WORDS words in LDS and stores the words the window leaves behind — plainly, but for the first word it ever touched and, at the
    // end, the last: those two may be shared with a neighbour and are ORed in (one atomic each instead of one per entry; zero words are
    // not stored: the fill left them zero).  Right (the suite passes with it) and SLOWER: Q3's orders build 0.078 -> 0.091 ms — the ballots,
    // the LDS atomics and the window's stores cost more than 1.5 M fire-and-forget atomics do (round 3 found the same for atomics merged by
    // a segmented scan).  Off; kept as the measurement it is (SDQLPY_AMD_X_DEFINES="XV_BITS=1").
    constexpr int XB_WORDS = 128;
    __shared__ uint32_t s_bw[XV_BITS ? TPB / WAVE : 1][XV_BITS ? XB_WORDS : 1];
    uint32_t* bw = s_bw[XV_BITS ? threadIdx.x / WAVE : 0];
    const bool xb_on = XV_BITS && st.wrow && st.bm && st.bm_shift == 0 && st.lin_rb == 0 && !st.grp_first;
    uint32_t xb_base = 0xFFFFFFFFu, xb_first = 0xFFFFFFFFu, xb_last = 0;
    DevStage st_nobits = st;
    if (xb_on) { st_nobits.bm = nullptr; for (int i = lane; i < XB_WORDS; i += WAVE) bw[i] = 0u; }
    auto xb_flush = [&](uint32_t upto, bool final) {                         // the window's words below `upto` go to memory, the window is cleared
        for (int i = lane; i < XB_WORDS; i += WAVE) {
            const uint32_t wi = xb_base + (uint32_t)i, v = bw[i];
            if (v && wi < upto) { if (wi == xb_first || (final && wi == xb_last)) atomicOr(&st.bm[wi], v); else st.bm[wi] = v; }
            bw[i] = 0u;
        }
    };
    auto xb_add = [&](bool keep, int64_t key) {
        uint64_t off = 0;
        const bool in = keep && bm_locate(st, key, off);
        const uint32_t w = (uint32_t)(off >> 5);
        const uint64_t all = __ballot(in);
        uint64_t todo = all;
        while (todo) {
            const uint32_t wmin = (uint32_t)__shfl((int)w, __builtin_ctzll(todo), WAVE);
            if (xb_base == 0xFFFFFFFFu) { xb_base = wmin; xb_first = wmin; }
            if (wmin - xb_base >= (uint32_t)XB_WORDS) { xb_flush(0xFFFFFFFFu, false); xb_base = wmin; }
            const bool mine = in && ((todo >> lane) & 1ull) && w - xb_base < (uint32_t)XB_WORDS;
            if (mine) atomicOr(&bw[w - xb_base], 1u << (off & 31));
            todo &= ~__ballot(mine);
        }
        if (all) xb_last = (uint32_t)__shfl((int)w, 63 - __builtin_clzll(all), WAVE);
    };
    auto flush = [&](int first, int count) {                                 // queued entries [first, first + count), count <= 64, one per lane
        if (st.wrow) row_index_note(st, seg, lane < count, lane < count ? (int64_t)q[0][first + lane] : 0, out + lane, count >= 64 ? ~0ull : ((1ull << count) - 1ull), carry);
        if (xb_on) xb_add(lane < count, lane < count ? (int64_t)q[0][first + lane] : 0);
        if (lane < count) {
            const int64_t key = (int64_t)q[0][first + lane];
            int64_t pay[MAX_STAGE_COLS] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < P::NV; ++k) pay[k] = (int64_t)q[1 + k][first + lane];
            if (a.key_lo <= a.key_hi && (key < a.key_lo || key > a.key_hi)) atomicOr(a.flags, 2);      // (the call fails; what is stored is never read)
#ifdef XV_EXP
            {   // timing experiments only (SDQLPY_AMD_XV_EXP): bit 0 no bitmap atomics, bit 1 no accumulator zeroing, bit 2 no payload stores
                const int64_t pos = out + lane;
                st.key[pos] = key;
                if (!(XV_EXP & 4)) { for (int qq = 0; qq < P::NV; ++qq) st.pay[qq][pos] = pay[qq]; }
                if (st.shits) st.shits[pos] = 0;
                if (st.sacc && !(XV_EXP & 2)) zero_acc(st, pos);
                if (st.bm && !(XV_EXP & 1)) { uint64_t off; if (bm_locate(st, key, off)) atomicOr(&st.bm[off >> 5], 1u << (off & 31)); }
            }
#else
            stage_store<-1>(xb_on ? st_nobits : st, out + lane, key, pay);
#endif
        }
        out += count;
    };
    auto enqueue = [&](const typename P::Regs& s, int64_t r0, uint32_t m) {   // the lane's survivors (bits of m), in row order
        if (!__ballot(m != 0)) return;
        int total;
        int at = qn + wave_excl_prefix4(__popc(m), total);
#pragma unroll
        for (int i = 0; i < XT_R; ++i) {
            XOut<P::NV> o;
            P::row(a, s, s_tab, i, r0 + i, o);
            if ((m >> i) & 1u) {
                if (o.bad) atomicOr(a.flags, 2);
                q[0][at] = (QT)o.key;
#pragma unroll
                for (int k = 0; k < P::NV; ++k) q[1 + k][at] = (QT)o.val[k];
                ++at;
            }
        }
        qn += total;
        __builtin_amdgcn_wave_barrier();
        int head = 0;
        while (qn - head >= WAVE) { flush(head, WAVE); head += WAVE; }
        if (head) {                                                            // what is left (< 64) moves to the front
            const int left = qn - head;
            QT v[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) v[k] = lane < left ? q[k][head + lane] : (QT)0;
            __builtin_amdgcn_wave_barrier();
            if (lane < left) {
#pragma unroll
                for (int k = 0; k < NQ; ++k) q[k][lane] = v[k];
            }
            __builtin_amdgcn_wave_barrier();
            qn = left;
        }
    };
    // One step in three parts, so that the main loop can put the NEXT step's streamed loads between this step's bitmap requests and
    // their use: `s_waitcnt vmcnt` counts in issue order, so loads requested BEFORE the bitmap words (the pipeline tried in round 3)
    // have to land before the words can be tested — nothing overlaps; requested AFTER them they stay in flight through the tests,
    // the queueing and the flush stores of this step.
    auto prep = [&](const auto& s, int64_t b, auto u_tag, auto tail_tag, auto& m, auto& off) {        // conditions on the streamed registers, bitmap offsets
        constexpr int U = decltype(u_tag)::value ? X8_U : 1;
        constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            m[u] = 0;
            const int64_t r0 = b + (int64_t)u * X8_STEP + (int64_t)lane * XT_R;
#pragma unroll
            for (int i = 0; i < XT_R; ++i) {
                bool p = (!TAIL || r0 + i < end) && P::gates(a, s[u], s_tab, i, r0 + i);
#pragma unroll
                for (int l = 0; l < NL; ++l) {
                    off[l][u][i] = 0;
                    if (P::NL > l) {
#if XV_OFFALL
                        bool in = true;                                            // (A/B: the key's own offset whatever the earlier conditions say)
                        const uint32_t o32 = P::lkoff(a, s[u], s_tab, i, r0 + i, l, in);
                        p = p & in; off[l][u][i] = in ? o32 : 0u;
#else
                        const uint32_t o32 = P::lkoff(a, s[u], s_tab, i, r0 + i, l, p); off[l][u][i] = p ? o32 : 0u;
#endif
                    }
                }
                m[u] |= p ? (1u << i) : 0u;
            }
        }
    };
    auto request = [&](auto u_tag, const auto& m, const auto& off, auto& w) {                                         // the bitmap words of a step's rows, together
        constexpr int U = decltype(u_tag)::value ? X8_U : 1;
#pragma unroll
        for (int l = 0; l < NL; ++l) if (P::NL > l)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < XT_R; ++i) {
#if defined(XV_EXP) && (XV_EXP & 8)
                    w[l][u][i] = 0xFFFFFFFFu;                                      // (timing experiment: no bitmap requests, every key "found")
#elif XV_MASKED
                    w[l][u][i] = 0u;
                    if ((m[u] >> i) & 1u) w[l][u][i] = bm[l][off[l][u][i] >> 5];    // only the lanes whose row is still alive take part in the request
#else
                    w[l][u][i] = bm[l][off[l][u][i] >> 5];
#endif
                }
    };
    auto finish = [&](const auto& s, int64_t b, auto u_tag, auto& m, const auto& off, const auto& w) {   // bit tests, survivors queued, full groups stored
        constexpr int U = decltype(u_tag)::value ? X8_U : 1;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int l = 0; l < NL; ++l) if (P::NL > l) {
                uint32_t hit = 0;
#pragma unroll
                for (int i = 0; i < XT_R; ++i) hit |= __builtin_amdgcn_ubfe(w[l][u][i], off[l][u][i] & 31u, 1u) << i;
                m[u] &= hit;
            }
#if defined(XV_EXP) && (XV_EXP & 16)
            m[u] = ((uint32_t)a.key_hi == 0x12345678u) ? m[u] : 0u;                 // (timing experiment: nothing queued)
#endif
            enqueue(s[u], b + (int64_t)u * X8_STEP + (int64_t)lane * XT_R, m[u]);
        }
    };
    auto step = [&](int64_t b, auto u_tag, auto tail_tag) {
        constexpr int U = decltype(u_tag)::value ? X8_U : 1;
        constexpr bool TAIL = decltype(tail_tag)::value;
        typename P::Regs s[U];
        uint32_t m[U], off[NL][U][XT_R], w[NL][U][XT_R];
#pragma unroll
        for (int u = 0; u < U; ++u) P::template sload<TAIL>(a, b + (int64_t)u * X8_STEP + (int64_t)lane * XT_R, TAIL ? end : nrows, s[u]);
        prep(s, b, u_tag, tail_tag, m, off);
        request(u_tag, m, off, w);
        finish(s, b, u_tag, m, off, w);
    };
    int64_t b = begin;
#if XV_PIPE
    {
        constexpr int64_t STEP2 = (int64_t)X8_STEP * X8_U;
        const int64_t nfull = (end - begin) / STEP2;
        if (nfull > 0) {
            typename P::Regs cur[X8_U], nxt[X8_U];
            auto load2 = [&](int64_t bb, typename P::Regs (&s)[X8_U]) {
#pragma unroll
                for (int u = 0; u < X8_U; ++u) P::template sload<false>(a, bb + (int64_t)u * X8_STEP + (int64_t)lane * XT_R, nrows, s[u]);
            };
            auto piped = [&](typename P::Regs (&now)[X8_U], typename P::Regs (&next)[X8_U], int64_t bb, int64_t bnext) {
                uint32_t m[X8_U], off[NL][X8_U][XT_R], w[NL][X8_U][XT_R];
                prep(now, bb, XBool<true>{}, XBool<false>{}, m, off);
                request(XBool<true>{}, m, off, w);
                __builtin_amdgcn_sched_barrier(0);
                load2(bnext, next);                                               // unconditional (after the last step: that step again): no wait piles up in front of a branch
                __builtin_amdgcn_sched_barrier(0);
                finish(now, bb, XBool<true>{}, m, off, w);
            };
            load2(b, cur);
            int64_t k = 0;
            for (; k + 2 <= nfull; k += 2, b += 2 * STEP2) {
                piped(cur, nxt, b, b + STEP2);
                piped(nxt, cur, b + STEP2, k + 2 < nfull ? b + 2 * STEP2 : b + STEP2);
            }
            if (k < nfull) { piped(cur, nxt, b, b); b += STEP2; }
        }
    }
#endif
    for (; b + (int64_t)X8_STEP * X8_U <= end; b += (int64_t)X8_STEP * X8_U) step(b, XBool<true>{}, XBool<false>{});
    for (; b + X8_STEP <= end; b += X8_STEP) step(b, XBool<false>{}, XBool<false>{});
    if (b < end) step(b, XBool<false>{}, XBool<true>{});
    if (qn) flush(0, qn);
    if (xb_on && xb_base != 0xFFFFFFFFu) xb_flush(xb_last + 1u, true);
    if (lane == 0) st.seg_count[seg] = (uint32_t)(out - begin);
}

}  // namespace sdqh
