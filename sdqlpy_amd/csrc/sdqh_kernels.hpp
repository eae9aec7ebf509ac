// sdqh_kernels.hpp — hand-written HIP kernels for gfx950 (MI355X / CDNA4) behind the sdqh C ABI.
//
// Every kernel here is HBM-bound integer / f64 work: no MFMA.  The rules that matter are the
// streaming ones: 16-byte-per-lane coalesced loads (1 KiB per wave instruction), >= 1024
// workgroups of 256 threads (4 wave64 per workgroup) so all 256 CUs / 8 XCDs stay full, all loads
// of a tile issued before the first use, group-by state kept in registers / LDS, deterministic
// two-level reductions instead of global float atomics where a result is a handful of sums.
//
// Reference loop shapes these replace (edin-dal/sdqlpy, src/sdqlpy/lib/sdql_ir_cpp_generator_par.py):
//   k_scan_sum        K-A 258-291   k_groupby_reg / k_groupby_lds   K-C 402-440 (small key domain)
//   k_stage/k_insert  K-B 331-369   k_probe_agg                     K-C 402-440 (group = matched entry)
//   k_compact         K-F 520-568
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sdqh.h"

namespace sdqh {

constexpr int TPB = 256;                 // threads per workgroup = 4 wave64
constexpr int WAVE = 64;
constexpr int ROWS_PER_LOAD = 2;         // two 8-byte rows = one 16-byte load per lane
constexpr int UNROLL = 2;                // 16-byte loads per column per thread per tile
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
    uint32_t sval[SDQH_MAX_STR_CONST];
    // ranges on tuple operand slots (an f-predicate whose column is also a value operand is
    // checked on the already-loaded operand instead of being loaded twice)
    uint32_t omask, _pad;
    double olo[4], ohi[4];
};

struct DevTuple {
    const double* op[4];
};

struct Slot {                 // 16 bytes, one open-addressing slot
    int64_t key;
    uint32_t rowref;          // stage index of the winning (lowest) build row
    uint32_t hits;            // rows aggregated into this entry
};

struct TableHeader {          // lives in device memory: sized on the device, no host round trip
    uint64_t cap_mask;        // capacity - 1 (capacity is a power of two); slot[capacity] is the EMPTY_KEY slot
    uint64_t staged;          // rows that survived the build-side filter (>= distinct keys)
    uint64_t distinct;        // filled by k_count
    uint64_t _pad;
};

struct DevTable {
    Slot* slots;
    const TableHeader* hdr;
    double* acc;              // (capacity+1) * SDQH_TUPLE_MAX_VALUES, or null
    const uint32_t* bm;       // exact key bitmap over [bm_lo, bm_hi], or null
    int64_t bm_lo, bm_hi;
    int32_t bitmap_only, _pad;
};

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
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

template <class T> struct Pair { T x, y; };

// Two consecutive 8-byte rows.  Full tiles: one aligned 16-byte load.  Tail: clamped scalar loads.
template <bool TAIL, class T>
__device__ __forceinline__ Pair<T> load2(const T* __restrict__ p, int64_t r, int64_t nrows) {
    Pair<T> v;
    if constexpr (!TAIL) {
        using V = T __attribute__((ext_vector_type(2)));
        V t = *reinterpret_cast<const V*>(p + r);
        v.x = t.x; v.y = t.y;
    } else {
        int64_t r0 = r < nrows ? r : nrows - 1, r1 = r + 1 < nrows ? r + 1 : nrows - 1;
        v.x = p[r0]; v.y = p[r1];
    }
    return v;
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

// contains(key): exact bitmap when the table has one, else open-addressing probe.
__device__ __forceinline__ bool table_contains(const DevTable& t, int64_t key, uint64_t cap_mask) {
    if (t.bm) {
        if (key < t.bm_lo || key > t.bm_hi) return false;
        uint64_t off = (uint64_t)(key - t.bm_lo);
        return (t.bm[off >> 5] >> (off & 31)) & 1u;
    }
    if (key == EMPTY_KEY) return t.slots[cap_mask + 1].rowref != NO_ROW;
    uint64_t h = hash_key(key) & cap_mask;
    for (;;) {
        int64_t k = t.slots[h].key;
        if (k == key) return true;
        if (k == EMPTY_KEY) return false;
        h = (h + 1) & cap_mask;
    }
}

// slot index of key, or -1
__device__ __forceinline__ int64_t table_find(const DevTable& t, int64_t key, uint64_t cap_mask) {
    if (t.bm) {
        if (key < t.bm_lo || key > t.bm_hi) return -1;
        uint64_t off = (uint64_t)(key - t.bm_lo);
        if (!((t.bm[off >> 5] >> (off & 31)) & 1u)) return -1;
    }
    if (key == EMPTY_KEY) return t.slots[cap_mask + 1].rowref != NO_ROW ? (int64_t)(cap_mask + 1) : -1;
    uint64_t h = hash_key(key) & cap_mask;
    for (;;) {
        int64_t k = t.slots[h].key;
        if (k == key) return (int64_t)h;
        if (k == EMPTY_KEY) return -1;
        h = (h + 1) & cap_mask;
    }
}

// ---- row filter on a pair of rows --------------------------------------------------------------
// Integer and double range predicates with their own columns, plus the optional string equality.
// Operand-slot ranges are applied by the caller on the loaded operands.
template <bool TAIL>
__device__ __forceinline__ void filter_pair(const DevFilter& f, int64_t r, int64_t nrows, bool& p0, bool& p1) {
#pragma unroll
    for (int i = 0; i < SDQH_MAX_IPRED; ++i) {
        if (i < f.ni) {
            Pair<int64_t> v = load2<TAIL>(f.ic[i], r, nrows);
            p0 &= (v.x >= f.ilo[i]) & (v.x <= f.ihi[i]);
            p1 &= (v.y >= f.ilo[i]) & (v.y <= f.ihi[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < SDQH_MAX_FPRED; ++i) {
        if (i < f.nf) {
            Pair<double> v = load2<TAIL>(f.fc[i], r, nrows);
            p0 &= (v.x >= f.flo[i]) & (v.x <= f.fhi[i]);
            p1 &= (v.y >= f.flo[i]) & (v.y <= f.fhi[i]);
        }
    }
    if (f.ns) {
        int64_t r0 = r < nrows ? r : nrows - 1, r1 = r + 1 < nrows ? r + 1 : nrows - 1;
        if (p0) p0 = str_equal(f.sc + r0 * f.swidth, f.swidth, f.sval, f.slen) != (f.sneg != 0);
        if (p1) p1 = str_equal(f.sc + r1 * f.swidth, f.swidth, f.sval, f.slen) != (f.sneg != 0);
    }
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
template <int SHAPE, bool TAIL>
__device__ __forceinline__ void scan_sum_tile(const DevFilter& f, const DevTuple& t, int64_t base, int64_t nrows,
                                              double (&acc)[4], int64_t& cnt) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        int64_t r = base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
        bool p0 = TAIL ? (r < nrows) : true, p1 = TAIL ? (r + 1 < nrows) : true;
        double x0[4] = {0, 0, 0, 0}, x1[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NOPS; ++j) { Pair<double> v = load2<TAIL>(t.op[j], r, nrows); x0[j] = v.x; x1[j] = v.y; }
        filter_pair<TAIL>(f, r, nrows, p0, p1);
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

template <int SHAPE>
__global__ __launch_bounds__(TPB) void k_scan_sum(DevFilter f, DevTuple t, int64_t nrows, double* __restrict__ partial) {
    constexpr int NV = TupleTraits<SHAPE>::NV;
    double acc[4] = {0, 0, 0, 0};
    int64_t cnt = 0;
    const int64_t full = nrows / TILE_ROWS;
    for (int64_t tile = blockIdx.x; tile < full; tile += gridDim.x)
        scan_sum_tile<SHAPE, false>(f, t, tile * TILE_ROWS, nrows, acc, cnt);
    if (full * TILE_ROWS < nrows && blockIdx.x == (unsigned)(full % gridDim.x))
        scan_sum_tile<SHAPE, true>(f, t, full * TILE_ROWS, nrows, acc, cnt);

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
__global__ __launch_bounds__(TPB) void k_sum_partials(const double* __restrict__ partial, int nparts, double* __restrict__ out) {
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
//   packed into one 64-bit word.  The workgroup keeps its key table in LDS (claimed by CAS in
//   first-come order); k_groupby_reg keeps G x NV accumulators per lane in registers and adds each
//   row under a per-group predicate (no atomics, no divergence); k_groupby_lds is the G <= 64
//   fall-back with LDS f64 atomics.  Both write one partial per workgroup; k_groupby_merge maps the
//   workgroups' local slots to global groups and folds them in a fixed order.
// partial layout per workgroup: keys[G] (u64) | acc[G][4] (f64) | cnt[G] (i64)
// =================================================================================================
struct DevGroupKeys {
    const void* col[SDQH_MAX_GROUPKEYS];
    int32_t is_str[SDQH_MAX_GROUPKEYS];
    int32_t nkeys, _pad;
};

template <bool TAIL>
__device__ __forceinline__ void load_group_keys(const DevGroupKeys& gk, int64_t r, int64_t nrows, uint64_t& k0, uint64_t& k1, bool& bad) {
    uint32_t part0[2] = {0, 0}, part1[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < SDQH_MAX_GROUPKEYS; ++j) {
        if (j < gk.nkeys) {
            int64_t r0 = (!TAIL || r < nrows) ? r : nrows - 1, r1 = (!TAIL || r + 1 < nrows) ? r + 1 : nrows - 1;
            if (gk.is_str[j]) {
                const uint32_t* c = static_cast<const uint32_t*>(gk.col[j]);
                if constexpr (!TAIL) { uint2 v = *reinterpret_cast<const uint2*>(c + r); part0[j] = v.x; part1[j] = v.y; }
                else { part0[j] = c[r0]; part1[j] = c[r1]; }
            } else {
                Pair<int64_t> v = load2<TAIL>(static_cast<const int64_t*>(gk.col[j]), r, nrows);
                bad |= (v.x < 0) | (v.x > 0xFFFFFFFEll) | (v.y < 0) | (v.y > 0xFFFFFFFEll);
                part0[j] = (uint32_t)v.x; part1[j] = (uint32_t)v.y;
            }
        }
    }
    k0 = (uint64_t)part0[0] | ((uint64_t)part0[1] << 32);
    k1 = (uint64_t)part1[0] | ((uint64_t)part1[1] << 32);
}

// claim-or-find in the workgroup's LDS key table; -1 when the table is full
template <int G>
__device__ __forceinline__ int lds_claim(unsigned long long* s_keys, uint64_t key) {
    for (int j = 0; j < G; ++j) {
        unsigned long long old = atomicCAS(&s_keys[j], (unsigned long long)EMPTY_GROUP, (unsigned long long)key);
        if (old == EMPTY_GROUP || old == key) return j;
    }
    return -1;
}

template <int SHAPE, int G, bool TAIL>
__device__ __forceinline__ void groupby_reg_tile(const DevFilter& f, const DevTuple& t, const DevGroupKeys& gk, int64_t base, int64_t nrows,
                                                 unsigned long long* s_keys, int* s_flags,
                                                 double (&acc)[G][4], int32_t (&cnt)[G]) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV;
    uint64_t rk[G];
#pragma unroll
    for (int g = 0; g < G; ++g) rk[g] = s_keys[g];          // register copy of the key table (LDS broadcast reads)
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        int64_t r = base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
        bool p[2] = {TAIL ? (r < nrows) : true, TAIL ? (r + 1 < nrows) : true};
        double x[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int j = 0; j < NOPS; ++j) { Pair<double> v = load2<TAIL>(t.op[j], r, nrows); x[0][j] = v.x; x[1][j] = v.y; }
        uint64_t key[2]; bool bad = false;
        load_group_keys<TAIL>(gk, r, nrows, key[0], key[1], bad);
        filter_pair<TAIL>(f, r, nrows, p[0], p[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            p[q] &= operand_ranges<NOPS>(f, x[q]);
            int slot = -1;
#pragma unroll
            for (int g = 0; g < G; ++g) slot = (rk[g] == key[q]) ? g : slot;
            if (p[q] && slot < 0) {                              // rare: first sight of a key in this workgroup
                slot = lds_claim<G>(s_keys, key[q]);
                if (slot < 0) atomicOr(&s_flags[0], 1);           // more than G groups
            }
            if (p[q] && bad) atomicOr(&s_flags[0], 2);            // int key outside [0, 2^32-2]
            double o[4] = {0, 0, 0, 0};
            tuple_eval<SHAPE>(x[q], o);
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

template <int SHAPE, int G>
__global__ __launch_bounds__(TPB) void k_groupby_reg(DevFilter f, DevTuple t, DevGroupKeys gk, int64_t nrows,
                                                     unsigned long long* __restrict__ pkeys, double* __restrict__ pacc,
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
    for (int64_t tile = blockIdx.x; tile < full; tile += gridDim.x)
        groupby_reg_tile<SHAPE, G, false>(f, t, gk, tile * TILE_ROWS, nrows, s_keys, s_flags, acc, cnt);
    if (full * TILE_ROWS < nrows && blockIdx.x == (unsigned)(full % gridDim.x))
        groupby_reg_tile<SHAPE, G, true>(f, t, gk, full * TILE_ROWS, nrows, s_keys, s_flags, acc, cnt);

    const int w = threadIdx.x / WAVE;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int k = 0; k < NV; ++k) { double v = wave_sum(acc[g][k]); if (lane_id() == 0) s_acc[w][g][k] = v; }
        int64_t c = wave_sum_i64((int64_t)cnt[g]);
        if (lane_id() == 0) s_cnt[w][g] = c;
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        pkeys[(size_t)blockIdx.x * G + g] = s_keys[g];
        int64_t c = 0;
        for (int i = 0; i < TPB / WAVE; ++i) c += s_cnt[i][g];
        pcnt[(size_t)blockIdx.x * G + g] = c;
        for (int k = 0; k < 4; ++k) {
            double v = 0.0;
            if (k < NV) for (int i = 0; i < TPB / WAVE; ++i) v += s_acc[i][g][k];
            pacc[((size_t)blockIdx.x * G + g) * 4 + k] = v;
        }
    }
    if (threadIdx.x == 0 && s_flags[0]) atomicOr(flags, s_flags[0]);
}

// Fall-back for up to 64 groups: accumulators in LDS, f64 LDS atomics (ds_add_f64).
template <int SHAPE, bool TAIL>
__device__ __forceinline__ void groupby_lds_tile(const DevFilter& f, const DevTuple& t, const DevGroupKeys& gk, int64_t base, int64_t nrows,
                                                 unsigned long long* s_keys, double (*s_acc)[4], unsigned long long* s_cnt, int* s_flags) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV, G = SDQH_MAX_SMALL_GROUPS;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
        int64_t r = base + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD;
        bool p[2] = {TAIL ? (r < nrows) : true, TAIL ? (r + 1 < nrows) : true};
        double x[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int j = 0; j < NOPS; ++j) { Pair<double> v = load2<TAIL>(t.op[j], r, nrows); x[0][j] = v.x; x[1][j] = v.y; }
        uint64_t key[2]; bool bad = false;
        load_group_keys<TAIL>(gk, r, nrows, key[0], key[1], bad);
        filter_pair<TAIL>(f, r, nrows, p[0], p[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            p[q] &= operand_ranges<NOPS>(f, x[q]);
            if (!p[q]) continue;
            if (bad) atomicOr(&s_flags[0], 2);
            int slot = -1;
            for (int g = 0; g < G; ++g) { unsigned long long k = s_keys[g]; if (k == key[q]) { slot = g; break; } if (k == EMPTY_GROUP) break; }
            if (slot < 0) slot = lds_claim<G>(s_keys, key[q]);
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
                                                     unsigned long long* __restrict__ pkeys, double* __restrict__ pacc,
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
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        pkeys[(size_t)blockIdx.x * G + g] = s_keys[g];
        pcnt[(size_t)blockIdx.x * G + g] = (int64_t)s_cnt[g];
        for (int k = 0; k < 4; ++k) pacc[((size_t)blockIdx.x * G + g) * 4 + k] = s_acc[g][k];
    }
    if (threadIdx.x == 0 && s_flags[0]) atomicOr(flags, s_flags[0]);
}

// One workgroup.  Phase 1: map every (workgroup, local slot) with rows to a global group (LDS
// CAS table, <= 64 groups).  Phase 2: per global group, fold the partials in a fixed
// thread-strided order and tree-reduce in LDS.  out: keys[64] | acc[64][4] | cnt[64] | ngroups.
__global__ __launch_bounds__(TPB) void k_groupby_merge(const unsigned long long* __restrict__ pkeys, const double* __restrict__ pacc,
                                                       const int64_t* __restrict__ pcnt, int nparts, int G, int max_groups,
                                                       signed char* __restrict__ slotmap,
                                                       unsigned long long* __restrict__ out_keys, double* __restrict__ out_acc,
                                                       int64_t* __restrict__ out_cnt, int* __restrict__ out_ngroups, int* __restrict__ flags) {
    constexpr int GMAX = SDQH_MAX_SMALL_GROUPS;
    __shared__ unsigned long long s_keys[GMAX];
    __shared__ double s_red[5][TPB];
    __shared__ int s_over;
    if (threadIdx.x < GMAX) s_keys[threadIdx.x] = EMPTY_GROUP;
    if (threadIdx.x == 0) s_over = 0;
    __syncthreads();
    const int total = nparts * G;
    for (int e = threadIdx.x; e < total; e += TPB) {
        signed char gs = -1;
        if (pcnt[e] > 0) {
            int s = lds_claim<GMAX>(s_keys, pkeys[e]);
            if (s < 0) s_over = 1;
            gs = (signed char)s;
        }
        slotmap[e] = gs;
    }
    __syncthreads();
    int ng = 0;
    for (int g = 0; g < GMAX; ++g) if (s_keys[g] != EMPTY_GROUP) ng = g + 1;    // claims are dense from slot 0
    if (threadIdx.x == 0) {
        *out_ngroups = ng;
        if (s_over || ng > max_groups) atomicOr(flags, 1);
    }
    for (int g = 0; g < ng; ++g) {
        double a[4] = {0, 0, 0, 0};
        int64_t c = 0;
        for (int e = threadIdx.x; e < total; e += TPB) {
            if (slotmap[e] == g) {
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] += pacc[(size_t)e * 4 + k];
                c += pcnt[e];
            }
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
            out_keys[g] = s_keys[g];
#pragma unroll
            for (int k = 0; k < 4; ++k) out_acc[g * 4 + k] = s_red[k][0];
            out_cnt[g] = reinterpret_cast<int64_t*>(s_red[4])[0];
        }
        __syncthreads();
    }
}

// =================================================================================================
// K-B: unique hash build.
//   k_stage   each wave64 owns a contiguous row segment; filter + semi-join probes; survivors are
//             compacted in row order with ballot + popcount prefix into the segment's own slice of
//             the stage arrays (no global atomics, no workgroup barrier)
//   k_clear   sizes the table on the device (capacity = pow2 >= 2 * staged rows) and resets slots
//   k_insert  wave per segment: CAS claim of the key, atomicMin of the stage index so the lowest
//             build row wins a duplicate key, exact key bitmap by atomicOr
// =================================================================================================
struct DevProbes {
    DevTable table[SDQH_MAX_PROBE];
    const int64_t* key[SDQH_MAX_PROBE];
    int32_t n, _pad;
};

struct DevStage {
    int64_t* key;                         // [nrows]
    int64_t* pay[SDQH_MAX_PAYLOAD];       // [nrows] each
    const int64_t* src_key;
    const int64_t* src_pay[SDQH_MAX_PAYLOAD];
    int32_t npay, _pad;
    uint32_t* seg_count;                  // [nseg]
    int64_t seg_rows;                     // rows per wave segment (multiple of 128)
    int32_t nseg, _pad2;
};

__device__ __forceinline__ bool row_passes(const DevFilter& f, const DevProbes& pr, int64_t r, const uint64_t* cap_masks) {
    bool p = true;
#pragma unroll
    for (int i = 0; i < SDQH_MAX_IPRED; ++i) if (i < f.ni && p) { int64_t v = f.ic[i][r]; p = (v >= f.ilo[i]) & (v <= f.ihi[i]); }
#pragma unroll
    for (int i = 0; i < SDQH_MAX_FPRED; ++i) if (i < f.nf && p) { double v = f.fc[i][r]; p = (v >= f.flo[i]) & (v <= f.fhi[i]); }
    if (f.ns && p) p = str_equal(f.sc + r * f.swidth, f.swidth, f.sval, f.slen) != (f.sneg != 0);
#pragma unroll
    for (int i = 0; i < SDQH_MAX_PROBE; ++i) if (i < pr.n && p) p = table_contains(pr.table[i], pr.key[i][r], cap_masks[i]);
    return p;
}

// Generic pair evaluation for the probing kernels: the first integer predicate (typically the date
// filter that removes most rows) is read with 16-byte loads for every row; everything else is
// read lazily, only by lanes still alive.
template <bool TAIL>
__device__ __forceinline__ void pass_pair(const DevFilter& f, const DevProbes& pr, int64_t r, int64_t nrows,
                                          const uint64_t* cap_masks, bool& p0, bool& p1) {
    int first = 0;
    if (f.ni > 0) {
        Pair<int64_t> v = load2<TAIL>(f.ic[0], r, nrows);
        p0 &= (v.x >= f.ilo[0]) & (v.x <= f.ihi[0]);
        p1 &= (v.y >= f.ilo[0]) & (v.y <= f.ihi[0]);
        first = 1;
    }
#pragma unroll
    for (int i = 1; i < SDQH_MAX_IPRED; ++i) if (i >= first && i < f.ni) {
        if (p0) { int64_t v = f.ic[i][r]; p0 = (v >= f.ilo[i]) & (v <= f.ihi[i]); }
        if (p1) { int64_t v = f.ic[i][r + 1]; p1 = (v >= f.ilo[i]) & (v <= f.ihi[i]); }
    }
#pragma unroll
    for (int i = 0; i < SDQH_MAX_FPRED; ++i) if (i < f.nf) {
        if (p0) { double v = f.fc[i][r]; p0 = (v >= f.flo[i]) & (v <= f.fhi[i]); }
        if (p1) { double v = f.fc[i][r + 1]; p1 = (v >= f.flo[i]) & (v <= f.fhi[i]); }
    }
    if (f.ns) {
        if (p0) p0 = str_equal(f.sc + r * f.swidth, f.swidth, f.sval, f.slen) != (f.sneg != 0);
        if (p1) p1 = str_equal(f.sc + (r + 1) * f.swidth, f.swidth, f.sval, f.slen) != (f.sneg != 0);
    }
#pragma unroll
    for (int i = 0; i < SDQH_MAX_PROBE; ++i) if (i < pr.n) {
        if (p0) p0 = table_contains(pr.table[i], pr.key[i][r], cap_masks[i]);
        if (p1) p1 = table_contains(pr.table[i], pr.key[i][r + 1], cap_masks[i]);
    }
}

__global__ __launch_bounds__(TPB) void k_stage(DevFilter f, DevProbes pr, DevStage st, int64_t nrows) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    uint64_t cap_masks[SDQH_MAX_PROBE] = {0, 0};
#pragma unroll
    for (int i = 0; i < SDQH_MAX_PROBE; ++i) if (i < pr.n && pr.table[i].hdr) cap_masks[i] = pr.table[i].hdr->cap_mask;
    const int64_t begin = (int64_t)seg * st.seg_rows;
    int64_t end = begin + st.seg_rows; if (end > nrows) end = nrows;
    const int lane = lane_id();
    const uint64_t lt = lanemask_lt();
    int64_t out = begin;                                   // wave-uniform write cursor into the segment's stage slice
    for (int64_t b = begin; b < end; b += WAVE * ROWS_PER_LOAD) {
        const int64_t r = b + (int64_t)lane * ROWS_PER_LOAD;
        bool p0, p1;
        if (b + WAVE * ROWS_PER_LOAD <= end) { p0 = p1 = true; pass_pair<false>(f, pr, r, nrows, cap_masks, p0, p1); }
        else {
            p0 = r < end; p1 = r + 1 < end;
            if (p0) p0 = row_passes(f, pr, r, cap_masks);
            if (p1) p1 = row_passes(f, pr, r + 1, cap_masks);
        }
        const uint64_t b0 = __ballot(p0), b1 = __ballot(p1);
        const int64_t pos0 = out + __popcll(b0 & lt) + __popcll(b1 & lt);
        const int64_t pos1 = pos0 + (p0 ? 1 : 0);
        if (p0) {
            st.key[pos0] = st.src_key[r];
#pragma unroll
            for (int q = 0; q < SDQH_MAX_PAYLOAD; ++q) if (q < st.npay) st.pay[q][pos0] = st.src_pay[q][r];
        }
        if (p1) {
            st.key[pos1] = st.src_key[r + 1];
#pragma unroll
            for (int q = 0; q < SDQH_MAX_PAYLOAD; ++q) if (q < st.npay) st.pay[q][pos1] = st.src_pay[q][r + 1];
        }
        out += __popcll(b0) + __popcll(b1);
    }
    if (lane == 0) st.seg_count[seg] = (uint32_t)(out - begin);
}

// Every workgroup recomputes the staged total from the segment counts (a few KB from L2), so the
// capacity is agreed on without a grid barrier; workgroup 0 publishes the header for later kernels.
__global__ __launch_bounds__(TPB) void k_clear(const uint32_t* __restrict__ seg_count, int nseg, uint64_t capmax,
                                               TableHeader* __restrict__ hdr, Slot* __restrict__ slots) {
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
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr->cap_mask = cap - 1; hdr->staged = staged; hdr->distinct = 0; hdr->_pad = 0; }
    const uint4 empty = {0u, 0x80000000u, NO_ROW, 0u};          // key = INT64_MIN, rowref = NO_ROW, hits = 0
    uint4* s4 = reinterpret_cast<uint4*>(slots);
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i <= cap; i += (uint64_t)gridDim.x * TPB) s4[i] = empty;
}

__global__ __launch_bounds__(TPB) void k_insert(DevStage st, DevTable t, uint32_t* __restrict__ bm, int zero_acc) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= st.nseg) return;
    const uint64_t mask = t.hdr->cap_mask;
    const int64_t base = (int64_t)seg * st.seg_rows;
    const uint32_t count = st.seg_count[seg];
    for (uint32_t i = lane_id(); i < count; i += WAVE) {
        const int64_t idx = base + i;
        const int64_t key = st.key[idx];
        uint64_t h;
        bool fresh = false;
        if (key == EMPTY_KEY) h = mask + 1;                             // the sentinel value itself lives in the extra slot
        else {
            h = hash_key(key) & mask;
            for (;;) {
                unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&t.slots[h].key),
                                                   (unsigned long long)EMPTY_KEY, (unsigned long long)key);
                if (old == (unsigned long long)EMPTY_KEY) { fresh = true; break; }   // claimed a fresh slot
                if ((int64_t)old == key) break;                                       // duplicate key: lowest row wins below
                h = (h + 1) & mask;
            }
        }
        const uint32_t prev = atomicMin(&t.slots[h].rowref, (uint32_t)idx);
        if (key == EMPTY_KEY) fresh = (prev == NO_ROW);
        if (fresh && zero_acc) { double4 z = {0, 0, 0, 0}; *reinterpret_cast<double4*>(t.acc + h * 4) = z; }
        if (bm && key >= t.bm_lo && key <= t.bm_hi) { uint64_t off = (uint64_t)(key - t.bm_lo); atomicOr(&bm[off >> 5], 1u << (off & 31)); }
    }
}

// distinct entries (table_size)
__global__ __launch_bounds__(TPB) void k_count(const Slot* __restrict__ slots, TableHeader* __restrict__ hdr) {
    const uint64_t cap = hdr->cap_mask + 1;
    unsigned long long n = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i <= cap; i += (uint64_t)gridDim.x * TPB)
        n += (slots[i].rowref != NO_ROW) ? 1 : 0;
    n = (unsigned long long)wave_sum_i64((int64_t)n);
    if (lane_id() == 0 && n) atomicAdd(reinterpret_cast<unsigned long long*>(&hdr->distinct), n);
}

// =================================================================================================
// K-C large: probe + aggregate into the matched entry.  The date-like first predicate and the probe
// key are streamed with 16-byte loads; the value operands are only read by the lanes that hit (the
// reference short-circuits the same way: `if (pred) if (contains) { ... += ep*(1.0-disc) }`).
// Hits are rare and scattered, so native f64 global atomics are the right tool here.
// =================================================================================================
template <int SHAPE, bool TAIL>
__device__ __forceinline__ void probe_agg_pair(const DevFilter& f, const DevTuple& t, const DevTable& tb, const int64_t* __restrict__ keycol,
                                               uint64_t mask, int64_t r, int64_t nrows) {
    constexpr int NOPS = TupleTraits<SHAPE>::NOPS, NV = TupleTraits<SHAPE>::NV;
    bool p[2] = {TAIL ? (r < nrows) : true, TAIL ? (r + 1 < nrows) : true};
    DevProbes none; none.n = 0;
    uint64_t nomask[SDQH_MAX_PROBE] = {0, 0};
    Pair<int64_t> kv = load2<TAIL>(keycol, r, nrows);
    if constexpr (!TAIL) pass_pair<false>(f, none, r, nrows, nomask, p[0], p[1]);
    else {
        if (p[0]) p[0] = row_passes(f, none, r, nomask);
        if (p[1]) p[1] = row_passes(f, none, r + 1, nomask);
    }
    const int64_t key[2] = {kv.x, kv.y};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (!p[q]) continue;
        const int64_t slot = table_find(tb, key[q], mask);
        if (slot < 0) continue;
        double x[4] = {0, 0, 0, 0}, o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NOPS; ++j) x[j] = t.op[j][r + q];
        if (!operand_ranges<NOPS>(f, x)) continue;
        tuple_eval<SHAPE>(x, o);
#pragma unroll
        for (int k = 0; k < NV; ++k) atomicAdd(&tb.acc[slot * 4 + k], o[k]);
        atomicAdd(&tb.slots[slot].hits, 1u);
    }
}

template <int SHAPE>
__global__ __launch_bounds__(TPB) void k_probe_agg(DevFilter f, DevTuple t, DevTable tb, const int64_t* __restrict__ keycol, int64_t nrows) {
    const uint64_t mask = tb.hdr->cap_mask;
    const int64_t full = nrows / TILE_ROWS;
    for (int64_t tile = blockIdx.x; tile < full; tile += gridDim.x) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            probe_agg_pair<SHAPE, false>(f, t, tb, keycol, mask, tile * TILE_ROWS + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD, nrows);
    }
    if (full * TILE_ROWS < nrows && blockIdx.x == (unsigned)(full % gridDim.x)) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
            probe_agg_pair<SHAPE, true>(f, t, tb, keycol, mask, full * TILE_ROWS + (int64_t)u * SUB_ROWS + (int64_t)threadIdx.x * ROWS_PER_LOAD, nrows);
    }
}

// =================================================================================================
// K-F: compact the entries with hits >= min_hits.  Each workgroup scans a contiguous run of slots,
// counts its survivors (ballot + LDS), reserves its output range with ONE global atomic, and writes
// key / payload (gathered from the stage arrays through rowref) / accumulators / hits.
// =================================================================================================
struct DevCompactOut {
    int64_t* keys; int64_t* pay[SDQH_MAX_PAYLOAD]; double* val[SDQH_TUPLE_MAX_VALUES]; int64_t* hits;
    unsigned long long* counter;
    int32_t npay, nval;
};
constexpr int COMPACT_SLOTS_PER_THREAD = 8;

__global__ __launch_bounds__(TPB) void k_compact(DevTable t, DevStage st, DevCompactOut o, uint32_t min_hits) {
    __shared__ uint32_t s_wave[TPB / WAVE];
    __shared__ unsigned long long s_base;
    const uint64_t cap = t.hdr->cap_mask + 1;                       // slots [0, cap] inclusive (cap = EMPTY_KEY slot)
    const uint64_t chunk = (uint64_t)TPB * COMPACT_SLOTS_PER_THREAD;
    for (uint64_t c0 = (uint64_t)blockIdx.x * chunk; c0 <= cap; c0 += (uint64_t)gridDim.x * chunk) {
        bool keep[COMPACT_SLOTS_PER_THREAD];
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < COMPACT_SLOTS_PER_THREAD; ++j) {
            const uint64_t s = c0 + (uint64_t)j * TPB + threadIdx.x;
            bool k = false;
            if (s <= cap) { const Slot sl = t.slots[s]; k = (sl.rowref != NO_ROW) && (sl.hits >= min_hits); }
            keep[j] = k; mine += k ? 1u : 0u;
        }
        // exclusive prefix of `mine` over the workgroup
        uint32_t incl = mine;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) { uint32_t v = __shfl_up(incl, off, WAVE); if (lane_id() >= off) incl += v; }
        const int w = threadIdx.x / WAVE;
        if (lane_id() == WAVE - 1) s_wave[w] = incl;
        __syncthreads();
        uint32_t wbase = 0, total = 0;
        for (int i = 0; i < TPB / WAVE; ++i) { if (i < w) wbase += s_wave[i]; total += s_wave[i]; }
        if (threadIdx.x == 0) s_base = total ? atomicAdd(o.counter, (unsigned long long)total) : 0ull;
        __syncthreads();
        uint64_t pos = s_base + wbase + (incl - mine);
#pragma unroll
        for (int j = 0; j < COMPACT_SLOTS_PER_THREAD; ++j) {
            if (!keep[j]) continue;
            const uint64_t s = c0 + (uint64_t)j * TPB + threadIdx.x;
            const Slot sl = t.slots[s];
            if (o.keys) o.keys[pos] = (s == cap) ? EMPTY_KEY : sl.key;
#pragma unroll
            for (int q = 0; q < SDQH_MAX_PAYLOAD; ++q) if (q < o.npay && o.pay[q]) o.pay[q][pos] = st.pay[q][sl.rowref];
#pragma unroll
            for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) if (k < o.nval && o.val[k]) o.val[k][pos] = t.acc[s * 4 + k];
            if (o.hits) o.hits[pos] = (int64_t)sl.hits;
            ++pos;
        }
        __syncthreads();
    }
}

// ---- column statistics ---------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_minmax(const int64_t* __restrict__ col, int64_t nrows, long long* __restrict__ out /*[2]*/) {
    long long lo = INT64_MAX, hi = INT64_MIN;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nrows; i += (int64_t)gridDim.x * TPB) {
        long long v = col[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi;
    }
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
        long long l2 = __shfl_down(lo, off, WAVE), h2 = __shfl_down(hi, off, WAVE);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if (lane_id() == 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}

}  // namespace sdqh
