// Standalone design harness for the round-3 streaming kernels: Q1's group-by and Q6's scan over TIGHT column encodings
// (sorted-dictionary codes of 1 / 2 bytes for dates, quantities, rates and flags; int32 cents for prices) with R rows
// per lane per load step, and three ways of keeping the group sums:
//   reg    G x NV f64 accumulators per lane in registers, every row added under a per-group predicate (round 2's scheme)
//   ldsa   per-lane private accumulators in LDS, ds_add_f64 (no return): one DS instruction per value
//   ldsrw  per-lane private accumulators in LDS, read - add - write
// Synthetic SF=10-shaped columns generated on the device; kernels timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics tools/microbench_tight.hip -o tools/mb_tight && tools/mb_tight
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int TPB = 256, WAVE = 64;

__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }
__device__ __forceinline__ double narrow_decode(int32_t n) {
    const double x = (double)n; const double q = x * 0.01; const double r = __builtin_fma(-q, 100.0, x); return __builtin_fma(r, 0.01, q);
}

__global__ void gen(int64_t n, uint16_t* ship, uint8_t* qty, int32_t* ep, uint8_t* disc, uint8_t* tax, uint8_t* rf, uint8_t* ls) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t h = mix64((uint64_t)i * 0x9E3779B97F4A7C15ull + 12345);
        ship[i] = (uint16_t)(h % 2526); qty[i] = (uint8_t)((h >> 24) % 50); ep[i] = 90000 + (int32_t)((h >> 32) % 10000000);
        disc[i] = (uint8_t)((h >> 40) % 11); tax[i] = (uint8_t)((h >> 44) % 9);
        rf[i] = (uint8_t)((h >> 48) % 3); ls[i] = (uint8_t)((h >> 52) & 1);
    }
}

struct Q1Args {
    const uint16_t* ship; const uint8_t* qty; const int32_t* ep; const uint8_t* disc; const uint8_t* tax; const uint8_t* rf; const uint8_t* ls;
    const double* dqty; const double* ddisc; const double* dtax;      // dictionaries (sorted distinct values)
    uint32_t ship_hi;                                                  // predicate in code space: ship <= ship_hi
    int64_t nrows; double* out;                                        // out[grid][G][5]
};

template <int R> struct Bytes;      // R one-byte codes per lane
template <> struct Bytes<4> { uint32_t w[1]; };
template <> struct Bytes<8> { uint32_t w[2]; };
template <> struct Bytes<16> { uint32_t w[4]; };
template <int R> __device__ __forceinline__ Bytes<R> ld8(const uint8_t* p, int64_t r) {
    Bytes<R> b;
    if constexpr (R == 4) b.w[0] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p + r));
    else if constexpr (R == 8) { using V = uint32_t __attribute__((ext_vector_type(2))); V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(p + r)); b.w[0] = t.x; b.w[1] = t.y; }
    else { using V = uint32_t __attribute__((ext_vector_type(4))); V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(p + r)); b.w[0] = t.x; b.w[1] = t.y; b.w[2] = t.z; b.w[3] = t.w; }
    return b;
}
template <int R> struct Halfs { uint32_t w[R / 2]; };
template <int R> __device__ __forceinline__ Halfs<R> ld16(const uint16_t* p, int64_t r) {
    Halfs<R> b;
    if constexpr (R == 4) { using V = uint32_t __attribute__((ext_vector_type(2))); V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(p + r)); b.w[0] = t.x; b.w[1] = t.y; }
    else {
        using V = uint32_t __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < R / 8; ++i) { V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(p + r) + i); b.w[4 * i] = t.x; b.w[4 * i + 1] = t.y; b.w[4 * i + 2] = t.z; b.w[4 * i + 3] = t.w; }
    }
    return b;
}
template <int R> struct Words { int32_t w[R]; };
template <int R> __device__ __forceinline__ Words<R> ld32(const int32_t* p, int64_t r) {
    Words<R> b;
    using V = int32_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < R / 4; ++i) { V t = __builtin_nontemporal_load(reinterpret_cast<const V*>(p + r) + i); b.w[4 * i] = t.x; b.w[4 * i + 1] = t.y; b.w[4 * i + 2] = t.z; b.w[4 * i + 3] = t.w; }
    return b;
}
template <int R> __device__ __forceinline__ uint32_t byte_of(const Bytes<R>& b, int i) { return (b.w[i / 4] >> (8 * (i % 4))) & 0xFFu; }
template <int R> __device__ __forceinline__ uint32_t half_of(const Halfs<R>& b, int i) { return (b.w[i / 2] >> (16 * (i % 2))) & 0xFFFFu; }

constexpr int G = 6, NV = 4;
enum Mode { REG = 0, LDSA = 1, LDSRW = 2, LDSP = 3 };      // LDSP: LDSA with the next step's loads requested before this step is consumed

template <int R, int U, int MODE, int GS = G, bool C32 = false>
__global__ __launch_bounds__(TPB) void k_q1(Q1Args a) {
    __shared__ double s_qty[64], s_1md[16], s_1pt[16];
    extern __shared__ double s_dyn[];                                    // LDS modes: acc[(g * 5 + k) * TPB + tid]
    if (threadIdx.x < 50) s_qty[threadIdx.x] = a.dqty[threadIdx.x];
    if (threadIdx.x < 11) s_1md[threadIdx.x] = 1.0 - a.ddisc[threadIdx.x];
    if (threadIdx.x < 9) s_1pt[threadIdx.x] = 1.0 + a.dtax[threadIdx.x];
    double acc[MODE == REG ? G : 1][NV];
    int32_t cnt[MODE == REG ? G : 1];
    if constexpr (MODE == REG) {
#pragma unroll
        for (int g = 0; g < G; ++g) { cnt[g] = 0; for (int k = 0; k < NV; ++k) acc[g][k] = 0.0; }
    } else {
        for (int i = threadIdx.x; i < GS * 5 * TPB; i += TPB) s_dyn[i] = 0.0;
    }
    __syncthreads();
    constexpr int64_t TILE = (int64_t)TPB * R;
    const int64_t full = a.nrows / (TILE * U);
    Halfs<R> nship[U]; Bytes<R> nqty[U], ndisc[U], ntax[U], nrf[U], nls[U]; Words<R> nep[U];
    auto request = [&](int64_t t) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = (t * U + u) * TILE + (int64_t)threadIdx.x * R;
            nship[u] = ld16<R>(a.ship, r); nqty[u] = ld8<R>(a.qty, r); ndisc[u] = ld8<R>(a.disc, r); ntax[u] = ld8<R>(a.tax, r);
            nrf[u] = ld8<R>(a.rf, r); nls[u] = ld8<R>(a.ls, r); nep[u] = ld32<R>(a.ep, r);
        }
    };
    if constexpr (MODE == LDSP) { if ((int64_t)blockIdx.x < full) request(blockIdx.x); }
    for (int64_t t = blockIdx.x; t < full; t += gridDim.x) {
        Halfs<R> ship[U]; Bytes<R> qty[U], disc[U], tax[U], rf[U], ls[U]; Words<R> ep[U];
        if constexpr (MODE == LDSP) {
#pragma unroll
            for (int u = 0; u < U; ++u) { ship[u] = nship[u]; qty[u] = nqty[u]; disc[u] = ndisc[u]; tax[u] = ntax[u]; rf[u] = nrf[u]; ls[u] = nls[u]; ep[u] = nep[u]; }
            request(t + gridDim.x < full ? t + gridDim.x : t);               // (unconditional: after the last step, this one again)
        } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = (t * U + u) * TILE + (int64_t)threadIdx.x * R;
            ship[u] = ld16<R>(a.ship, r); qty[u] = ld8<R>(a.qty, r); disc[u] = ld8<R>(a.disc, r); tax[u] = ld8<R>(a.tax, r);
            rf[u] = ld8<R>(a.rf, r); ls[u] = ld8<R>(a.ls, r); ep[u] = ld32<R>(a.ep, r);
        }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const bool p = half_of<R>(ship[u], i) <= a.ship_hi;
                int g = (int)(byte_of<R>(rf[u], i) * 2 + byte_of<R>(ls[u], i));
                if constexpr (GS < G) g = g >= 4 ? g - 2 : g;            // (a 4-slot table: the harness folds slots 4, 5 onto 2, 3)
                const double q = s_qty[byte_of<R>(qty[u], i)];
                const double e = narrow_decode(ep[u].w[i]);
                const double dp = e * s_1md[byte_of<R>(disc[u], i)];
                const double ch = dp * s_1pt[byte_of<R>(tax[u], i)];
                if constexpr (MODE == REG) {
#pragma unroll
                    for (int gg = 0; gg < G; ++gg) {
                        const bool m = p && g == gg;
                        acc[gg][0] += m ? q : 0.0; acc[gg][1] += m ? e : 0.0; acc[gg][2] += m ? dp : 0.0; acc[gg][3] += m ? ch : 0.0;
                        cnt[gg] += m ? 1 : 0;
                    }
                } else if constexpr (MODE == LDSA || MODE == LDSP) {
                    if (p) {
                        double* base = s_dyn + (size_t)g * 5 * TPB + threadIdx.x;
                        __hip_atomic_fetch_add(base, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(base + TPB, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(base + 2 * TPB, dp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(base + 3 * TPB, ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if constexpr (C32) __hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(base + 4 * TPB), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        else __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(base + 4 * TPB), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                } else {
                    if (p) {
                        double* base = s_dyn + (size_t)g * 5 * TPB + threadIdx.x;
                        base[0] += q; base[TPB] += e; base[2 * TPB] += dp; base[3 * TPB] += ch;
                        reinterpret_cast<unsigned long long*>(base)[4 * TPB] += 1ull;
                    }
                }
            }
        }
    }
    // (tail rows beyond full * TILE * U are left out: a timing harness)
    __shared__ double s_red[TPB / WAVE][G][5];
    const int w = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
    for (int g = 0; g < (MODE == REG ? G : GS); ++g)
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            double v;
            if constexpr (MODE == REG) v = k < 4 ? acc[g][k] : (double)cnt[g];
            else { v = s_dyn[(size_t)(g * 5 + k) * TPB + threadIdx.x]; if (k == 4) v = (double)__double_as_longlong(v); }
#pragma unroll
            for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
            if (lane == 0) s_red[w][g][k] = v;
        }
    __syncthreads();
    if (threadIdx.x < G * 5) {
        const int g = threadIdx.x / 5, k = threadIdx.x % 5;
        double v = 0; for (int i = 0; i < TPB / WAVE; ++i) v += s_red[i][g][k];
        a.out[((size_t)blockIdx.x * G + g) * 5 + k] = v;
    }
}

// Q6: sum(ep * disc) where ship in [lo, hi], disc in [dlo, dhi], qty < qhi — all conditions in code space
struct Q6Args { const uint16_t* ship; const uint8_t* qty; const int32_t* ep; const uint8_t* disc; const double* ddisc; uint32_t slo, shi, dlo, dhi, qhi; int64_t nrows; double* out; };
template <int R, int U>
__global__ __launch_bounds__(TPB) void k_q6(Q6Args a) {
    __shared__ double s_disc[16];
    if (threadIdx.x < 11) s_disc[threadIdx.x] = a.ddisc[threadIdx.x];
    __syncthreads();
    double acc = 0.0; int64_t cnt = 0;
    constexpr int64_t TILE = (int64_t)TPB * R;
    const int64_t full = a.nrows / (TILE * U);
    for (int64_t t = blockIdx.x; t < full; t += gridDim.x) {
        Halfs<R> ship[U]; Bytes<R> qty[U], disc[U]; Words<R> ep[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = (t * U + u) * TILE + (int64_t)threadIdx.x * R;
            ship[u] = ld16<R>(a.ship, r); qty[u] = ld8<R>(a.qty, r); disc[u] = ld8<R>(a.disc, r); ep[u] = ld32<R>(a.ep, r);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const uint32_t s = half_of<R>(ship[u], i), d = byte_of<R>(disc[u], i), q = byte_of<R>(qty[u], i);
                const bool p = s >= a.slo && s <= a.shi && d >= a.dlo && d <= a.dhi && q < a.qhi;
                const double v = narrow_decode(ep[u].w[i]) * s_disc[d];
                acc += p ? v : 0.0; cnt += p ? 1 : 0;
            }
    }
    __shared__ double s_red[TPB / WAVE]; __shared__ int64_t s_c[TPB / WAVE];
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) { acc += __shfl_down(acc, off, WAVE); cnt += __shfl_down(cnt, off, WAVE); }
    if (threadIdx.x % WAVE == 0) { s_red[threadIdx.x / WAVE] = acc; s_c[threadIdx.x / WAVE] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) { double v = 0; int64_t c = 0; for (int i = 0; i < TPB / WAVE; ++i) { v += s_red[i]; c += s_c[i]; } a.out[blockIdx.x * 2] = v; a.out[blockIdx.x * 2 + 1] = (double)c; }
}

template <class K, class A>
static void time_kernel(const char* name, K kern, A args, unsigned grid, size_t lds, double bytes, double* out, int nout) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 14; ++it) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kern, dim3(grid), dim3(TPB), lds, 0, args); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); if (it >= 4) ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    std::vector<double> h((size_t)grid * nout);
    CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
    double chk[8] = {0}; for (unsigned b = 0; b < grid; ++b) for (int j = 0; j < nout && j < 8; ++j) chk[j] += h[(size_t)b * nout + j];
    printf("%-28s grid %5u lds %6zu  median %.4f ms  min %.4f  -> %.0f GB/s   check %.6e %.6e %.0f\n", name, grid, lds, ms[ms.size() / 2], ms[0], bytes / (ms[ms.size() / 2] * 1e-3) / 1e9,
           chk[0], chk[nout > 3 ? 3 : 0], chk[nout > 4 ? 4 : 1]);
}

int main() {
    const int64_t n = 60003415;
    uint16_t* ship; uint8_t *qty, *disc, *tax, *rf, *ls; int32_t* ep;
    CK(hipMalloc(&ship, n * 2 + 256)); CK(hipMalloc(&qty, n + 256)); CK(hipMalloc(&disc, n + 256)); CK(hipMalloc(&tax, n + 256)); CK(hipMalloc(&rf, n + 256)); CK(hipMalloc(&ls, n + 256)); CK(hipMalloc(&ep, n * 4 + 256));
    hipLaunchKernelGGL(gen, dim3(4096), dim3(256), 0, 0, n, ship, qty, ep, disc, tax, rf, ls);
    double hq[50], hd[11], ht[9];
    for (int i = 0; i < 50; ++i) hq[i] = 1.0 + i;
    for (int i = 0; i < 11; ++i) hd[i] = i / 100.0;
    for (int i = 0; i < 9; ++i) ht[i] = i / 100.0;
    double *dq, *dd, *dt, *out;
    CK(hipMalloc(&dq, sizeof(hq))); CK(hipMalloc(&dd, sizeof(hd))); CK(hipMalloc(&dt, sizeof(ht))); CK(hipMalloc(&out, (size_t)8192 * G * 5 * 8));
    CK(hipMemcpy(dq, hq, sizeof(hq), hipMemcpyHostToDevice)); CK(hipMemcpy(dd, hd, sizeof(hd), hipMemcpyHostToDevice)); CK(hipMemcpy(dt, ht, sizeof(ht), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    Q1Args a{ship, qty, ep, disc, tax, rf, ls, dq, dd, dt, 2487u, n, out};
    const double b1 = 11.0 * n, b6 = 8.0 * n;
    const size_t lds = (size_t)G * 5 * TPB * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_q1<8, 1, LDSA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#define RUN1(R, U, MODE, res) time_kernel("q1 R=" #R " U=" #U " " #MODE " x" #res, k_q1<R, U, MODE>, a, 256u * res, MODE == REG ? 0 : lds, b1, out, G * 5)
    RUN1(8, 1, REG, 2); RUN1(8, 1, REG, 4); RUN1(8, 2, REG, 2); RUN1(16, 1, REG, 2); RUN1(4, 2, REG, 4);
    RUN1(8, 1, LDSA, 2); RUN1(8, 2, LDSA, 2); RUN1(8, 1, LDSA, 1); RUN1(16, 1, LDSA, 2); RUN1(4, 2, LDSA, 2); RUN1(8, 2, LDSA, 1);
    RUN1(8, 1, LDSRW, 2); RUN1(8, 2, LDSRW, 2); RUN1(16, 1, LDSRW, 2);
#define RUNG(R, U, GS, C32, res) time_kernel("q1 R=" #R " U=" #U " LDSA GS=" #GS " C32=" #C32 " x" #res, k_q1<R, U, LDSA, GS, C32>, a, 256u * res, (size_t)GS * 5 * TPB * 8, b1, out, G * 5)
    RUN1(8, 1, LDSP, 2); RUN1(8, 2, LDSP, 2); RUN1(8, 1, LDSP, 1); RUN1(4, 2, LDSP, 2);
    RUNG(8, 2, 6, true, 2); RUNG(8, 2, 4, false, 2); RUNG(8, 2, 4, false, 3); RUNG(8, 2, 4, true, 3); RUNG(8, 1, 4, true, 3); RUNG(8, 1, 4, true, 4); RUNG(8, 2, 4, true, 4);
    Q6Args q{ship, qty, ep, disc, dd, 730u, 1094u, 5u, 7u, 23u, n, out};
#define RUN6(R, U, res) time_kernel("q6 R=" #R " U=" #U " x" #res, k_q6<R, U>, q, 256u * res, 0, b6, out, 2)
    RUN6(8, 1, 2); RUN6(8, 1, 4); RUN6(8, 2, 2); RUN6(8, 2, 4); RUN6(16, 1, 2); RUN6(16, 1, 4); RUN6(4, 2, 4); RUN6(4, 4, 4); RUN6(8, 4, 2);
    return 0;
}
