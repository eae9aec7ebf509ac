// sdqh_codes.hip — sorted-dictionary code twins of streamed columns (round 3).
//
// A column a kernel STREAMS may, besides its exact 4-byte twin (sdqh_hip.hip: ensure_narrow), have a 1- or 2-byte twin:
// per row the RANK of the row's value among the column's distinct values, plus the dictionary (the distinct values,
// ascending, 8 raw bytes each).  TPCH's dates (2 526 days), quantities (50), discounts (11), taxes (9), flags (2-3) all
// qualify; a price column (a million distinct values) does not and keeps its 4-byte form.  What it buys the streaming
// kernels (sdqh_xkernels.hpp: x_tight): Q1 reads 11 bytes per row instead of 22, Q6 8 instead of 16 — and because ranks
// preserve order, a comparison with a constant is an integer comparison of the code with the constant's rank, and a
// value is one LDS read of the dictionary: no decode arithmetic at all.  Exact by construction: the dictionary holds the
// column's own values (for doubles: the bits narrow_decode reproduces, verified per row when the 4-byte twin was built).
//
// Built from the 4-byte twin (or the byte twin of a string(1) column, or an I64 column itself) in three small kernels:
//   k_code_presence   every workgroup marks the values of its rows in an LDS bitmap over [lo, lo + range) and ORs the
//                     non-zero words into the global bitmap (per-row global atomics on a handful of words would serialise)
//   k_code_rank       ONE workgroup: exclusive prefix of the words' popcounts -> rank of each word's first value, the
//                     number of distinct values, and the dictionary
//   k_code_write      code[row] = prefix[word] + popcount(bits below)
// Nothing of this has a counterpart in the reference (it streams 8-byte / UCS-4 columns, include/varchar.h:1-3).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#define SDQH_DECLS_ONLY 1            // argument structs and helpers of the kernel header, not a second copy of its kernels
#include "sdqh_host.hpp"

using namespace sdqh_host;

namespace {

constexpr int64_t CODE_MAX_RANGE = 1 << 19;          // bits of the presence bitmap: 64 KiB of LDS per workgroup

// the integer a row's value is ranked by: the I64 value, the cents of a two-decimal double (its 4-byte twin), a code unit
template <int KIND> __device__ __forceinline__ int64_t code_source(const void* src, int64_t r) {
    if constexpr (KIND == 0) return static_cast<const int64_t*>(src)[r];
    else if constexpr (KIND == 1) return (int64_t)static_cast<const int32_t*>(src)[r];
    else return (int64_t)static_cast<const uint8_t*>(src)[r];
}

template <int KIND>
__global__ __launch_bounds__(1024) void k_code_presence(const void* __restrict__ src, int64_t nrows, int64_t lo, int nwords, uint32_t* __restrict__ bm) {
    extern __shared__ uint32_t s_bm[];
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) s_bm[i] = 0u;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t off = (uint64_t)(code_source<KIND>(src, r) - lo);
        const uint32_t bit = 1u << (off & 31);
        if (!(s_bm[off >> 5] & bit)) atomicOr(&s_bm[off >> 5], bit);         // a handful of distinct values: almost always set already
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) { const uint32_t w = s_bm[i]; if (w) atomicOr(&bm[i], w); }
}

// one workgroup of 1024: prefix[w] = set bits in words before w; *ndict = total; dict[rank] = raw value of that rank
template <int KIND>
__global__ __launch_bounds__(1024) void k_code_rank(const uint32_t* __restrict__ bm, int nwords, int64_t lo, uint32_t* __restrict__ prefix,
                                                     int64_t* __restrict__ dict, int dict_cap, int* __restrict__ ndict) {
    __shared__ uint32_t s_part[1024];
    const int per = (nwords + 1023) / 1024;
    const int w0 = threadIdx.x * per, w1 = min(nwords, w0 + per);
    uint32_t sum = 0;
    for (int w = w0; w < w1; ++w) sum += (uint32_t)__popc(bm[w]);
    s_part[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = (int)threadIdx.x >= off ? s_part[threadIdx.x - off] : 0u;
        __syncthreads();
        s_part[threadIdx.x] += v;
        __syncthreads();
    }
    if (threadIdx.x == 1023) *ndict = (int)s_part[1023];
    uint32_t run = s_part[threadIdx.x] - sum;
    for (int w = w0; w < w1; ++w) {
        prefix[w] = run;
        uint32_t bits = bm[w];
        while (bits) {
            const int b = __ffs((int)bits) - 1;
            bits &= bits - 1;
            if ((int)run < dict_cap) {
                const int64_t v = lo + (int64_t)w * 32 + b;
                dict[run] = KIND == 1 ? __double_as_longlong(narrow_decode((int32_t)v)) : v;      // KIND 1: the double the cents stand for
            }
            ++run;
        }
    }
}

template <int KIND, class CT>
__global__ __launch_bounds__(TPB) void k_code_write(const void* __restrict__ src, int64_t nrows, int64_t lo, const uint32_t* __restrict__ bm,
                                                    const uint32_t* __restrict__ prefix, CT* __restrict__ code) {
    // four consecutive rows per thread and step: one 4- / 8-byte store
    for (int64_t r0 = ((int64_t)blockIdx.x * TPB + threadIdx.x) * 4; r0 < nrows; r0 += (int64_t)gridDim.x * TPB * 4) {
        CT out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = r0 + i < nrows ? r0 + i : nrows - 1;
            const uint64_t off = (uint64_t)(code_source<KIND>(src, r) - lo);
            const uint32_t w = bm[off >> 5];
            out[i] = (CT)(prefix[off >> 5] + (uint32_t)__popc(w & ((1u << (off & 31)) - 1u)));
        }
        if (r0 + 4 <= nrows) {
            if constexpr (sizeof(CT) == 1) *reinterpret_cast<uint32_t*>(code + r0) = (uint32_t)out[0] | ((uint32_t)out[1] << 8) | ((uint32_t)out[2] << 16) | ((uint32_t)out[3] << 24);
            else *reinterpret_cast<uint2*>(code + r0) = make_uint2((uint32_t)out[0] | ((uint32_t)out[1] << 16), (uint32_t)out[2] | ((uint32_t)out[3] << 16));
        } else {
            for (int i = 0; i < 4 && r0 + i < nrows; ++i) code[r0 + i] = out[i];
        }
    }
}

__global__ __launch_bounds__(TPB) void k_minmax32(const int32_t* __restrict__ col, int64_t nrows, int* __restrict__ out /*[2]: min, max*/) {
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < nrows; i += (int64_t)gridDim.x * TPB) { const int v = col[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) { const int l2 = __shfl_down(lo, off, WAVE), h2 = __shfl_down(hi, off, WAVE); lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
    if (lane_id() == 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}

template <int KIND>
bool build_codes(sdqh_ctx* ctx, sdqh_column* c, const void* src, int64_t lo, int64_t hi) {
    const int64_t range = hi - lo + 1;
    if (range < 1 || range > CODE_MAX_RANGE) return false;
    const int nwords = (int)((range + 31) / 32);
    uint32_t* bm = static_cast<uint32_t*>(pool_alloc(ctx, (size_t)nwords * 4 + 64));
    uint32_t* prefix = static_cast<uint32_t*>(pool_alloc(ctx, (size_t)nwords * 4 + 64));
    int64_t* dict = static_cast<int64_t*>(attach_alloc(ctx, c, (size_t)65536 * 8));
    int* nd = static_cast<int*>(pool_alloc(ctx, 64));
    bool ok = bm && prefix && dict && nd && hipMemsetAsync(bm, 0, (size_t)nwords * 4, ctx->stream) == hipSuccess;
    int ndict = 0;
    if (ok) {
        const unsigned grid = (unsigned)std::min<int64_t>((c->nrows + 1023) / 1024, (int64_t)ctx->num_cu * 2);
        hipLaunchKernelGGL(k_code_presence<KIND>, dim3(grid), dim3(1024), (size_t)nwords * 4, ctx->stream, src, c->nrows, lo, nwords, bm);
        hipLaunchKernelGGL(k_code_rank<KIND>, dim3(1), dim3(1024), 0, ctx->stream, bm, nwords, lo, prefix, dict, 65536, nd);
        int* host = static_cast<int*>(ctx->result_host);
        ok = hipMemcpyAsync(host, nd, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
        ndict = ok ? host[0] : 0;
        ok = ok && ndict >= 1 && ndict <= 65536;
    }
    void* code = nullptr;
    if (ok) {
        const int width = ndict <= 256 ? 1 : 2;
        code = attach_alloc(ctx, c, (size_t)c->nrows * width + 256);               // padded: the streaming kernels load whole 16-byte groups
        ok = code != nullptr;
        if (ok) {
            const unsigned grid = (unsigned)std::min<int64_t>((c->nrows / 4 + TPB) / TPB, (int64_t)ctx->num_cu * 8);
            if (width == 1) hipLaunchKernelGGL((k_code_write<KIND, uint8_t>), dim3(grid), dim3(TPB), 0, ctx->stream, src, c->nrows, lo, bm, prefix, static_cast<uint8_t*>(code));
            else hipLaunchKernelGGL((k_code_write<KIND, uint16_t>), dim3(grid), dim3(TPB), 0, ctx->stream, src, c->nrows, lo, bm, prefix, static_cast<uint16_t*>(code));
            c->dict_host.assign((size_t)ndict, 0);
            ok = hipMemsetAsync(static_cast<char*>(code) + (size_t)c->nrows * width, 0, 256, ctx->stream) == hipSuccess &&
                 hipMemcpyAsync(c->dict_host.data(), dict, (size_t)ndict * 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                 hipStreamSynchronize(ctx->stream) == hipSuccess && hipGetLastError() == hipSuccess;
            if (ok) { c->code = code; c->code_width = width; c->dict = dict; c->ndict = ndict; c->code_state = 1; dict = nullptr; code = nullptr; }
        }
    }
    if (!ok) { (void)hipGetLastError(); c->dict_host.clear(); }
    if (code) attach_free(ctx, c, code);
    if (dict) attach_free(ctx, c, dict);
    if (nd) pool_free(ctx, nd);
    if (prefix) pool_free(ctx, prefix);
    if (bm) pool_free(ctx, bm);
    return ok;
}

}  // namespace

namespace sdqh_host {

bool column_codes(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->code_state >= 0) return c->code_state == 1;
    c->code_state = 0;
    if (c->nrows < 2 || ctx->compile_only) return false;
    if (c->dtype == SDQH_I64) {
        if (column_minmax(ctx, c)) return false;
        if (c->mx - c->mn >= CODE_MAX_RANGE || c->mx - c->mn < 0) return false;
        return build_codes<0>(ctx, c, c->data, c->mn, c->mx);
    }
    if (c->dtype == SDQH_F64) {
        const int32_t* twin = static_cast<const int32_t*>(column_narrow(ctx, c));
        if (!twin) return false;
        int* mm = static_cast<int*>(pool_alloc(ctx, 64));
        if (!mm) return false;
        int init[2] = {INT32_MAX, INT32_MIN};
        int* host = static_cast<int*>(ctx->result_host);
        std::memcpy(host, init, sizeof(init));
        bool ok = hipMemcpyAsync(mm, host, 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
        if (ok) {
            hipLaunchKernelGGL(k_minmax32, dim3((unsigned)std::min<int64_t>((c->nrows + TPB - 1) / TPB, (int64_t)ctx->num_cu * 8)), dim3(TPB), 0, ctx->stream, twin, c->nrows, mm);
            ok = hipMemcpyAsync(host, mm, 8, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
        }
        const int64_t lo = host[0], hi = host[1];
        pool_free(ctx, mm);
        if (!ok) { (void)hipGetLastError(); return false; }
        return build_codes<1>(ctx, c, twin, lo, hi);
    }
    if (c->dtype == SDQH_STR && c->width == 1) {
        const void* twin = column_narrow(ctx, c);
        return twin ? build_codes<2>(ctx, c, twin, 0, 255) : false;
    }
    return false;
}

void column_codes_release(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->code) attach_free(ctx, c, c->code);
    if (c->dict) attach_free(ctx, c, c->dict);
    c->code = nullptr; c->dict = nullptr; c->code_width = 0; c->ndict = 0; c->code_state = -1; c->dict_host.clear();
}

}  // namespace sdqh_host
