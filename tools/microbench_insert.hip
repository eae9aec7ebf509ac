// Standalone harness for the hash-layout insert pass (k_insert: one CAS per staged row claims a 32-byte packed slot, the claimer
// stores payload and stage row): why does it take ~50 us whether 100 K (Q5's suppliers as a hash table) or 432 K (Q9's green
// part-supplier pairs) rows are inserted?  Variants of the claim, timed with HIP events on synthetic segments:
//   cas64   device-scope 64-bit CAS on the slot's key + three 8-byte stores (the shipped kernel)
//   casonly the CAS alone
//   stores  plain stores only (no claim: wrong, timing)
//   read    the staged keys read, nothing written
//   wg      the CAS at workgroup scope (wrong across workgroups, timing: does the scope decide where the atomic executes?)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_insert.hip -o tools/mb_insert && tools/mb_insert
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int TPB = 256, WAVE = 64;
constexpr int64_t EMPTY_KEY = INT64_MIN;

__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31; return x; }
__device__ __forceinline__ uint32_t hash_key(int64_t k) { return (uint32_t)(((uint64_t)k * 0x9E3779B97F4A7C15ull) >> 32); }

__global__ void gen(int64_t* key, int64_t* p0, int64_t* p1, uint32_t* seg_count, int nseg, int64_t seg_rows, uint32_t per_seg) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= nseg) return;
    const int lane = threadIdx.x % WAVE;
    for (uint32_t i = lane; i < per_seg; i += WAVE) {
        const int64_t idx = (int64_t)seg * seg_rows + i;
        const uint64_t a = 1 + mix64((uint64_t)seg * 977 + 5) % 2000000, b = 1 + (uint64_t)i * 25003 % 100000;
        key[idx] = (int64_t)((a << 32) | b) + seg;            // distinct per (seg, i)
        p0[idx] = idx; p1[idx] = ~idx;
    }
    if (lane == 0) seg_count[seg] = per_seg;
}

__global__ void clear(int64_t* slots, uint64_t cap) {
    using V = long long __attribute__((ext_vector_type(2)));
    const V a = {(long long)EMPTY_KEY, 0}, b = {0, -1};
    V* s2 = reinterpret_cast<V*>(slots);
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < cap; i += (uint64_t)gridDim.x * TPB) { s2[2 * i] = a; s2[2 * i + 1] = b; }
}

template <int MODE>
__global__ __launch_bounds__(TPB) void insert(const int64_t* __restrict__ key, const int64_t* __restrict__ p0, const int64_t* __restrict__ p1,
                                              const uint32_t* __restrict__ seg_count, int nseg, int64_t seg_rows, int64_t* slots, uint64_t mask, int* sink) {
    const int seg = blockIdx.x * (TPB / WAVE) + threadIdx.x / WAVE;
    if (seg >= nseg) return;
    const int64_t base = (int64_t)seg * seg_rows;
    const uint32_t count = seg_count[seg];
    int acc = 0;
    for (uint32_t i = threadIdx.x % WAVE; i < count; i += WAVE) {
        const int64_t idx = base + i;
        const int64_t k = key[idx];
        uint64_t h = hash_key(k) & mask;
        if (MODE == 3) { acc += (int)h; continue; }
        if (MODE == 5) {                                           // 32-bit CAS on the slot's row reference; the key is compared through the stage
            uint32_t* ref = reinterpret_cast<uint32_t*>(&slots[h * 4 + 3]);
            for (;;) {
                const uint32_t old = atomicCAS(ref, 0xFFFFFFFFu, (uint32_t)idx);
                if (old == 0xFFFFFFFFu) { slots[h * 4] = k; slots[h * 4 + 1] = p0[idx]; slots[h * 4 + 2] = p1[idx]; break; }
                if (key[old] == k) { acc++; break; }
                h = (h + 1) & mask; ref = reinterpret_cast<uint32_t*>(&slots[h * 4 + 3]);
            }
            continue;
        }
        if (MODE == 6) { atomicMin(reinterpret_cast<uint32_t*>(slots) + (h & (mask >> 2)), (uint32_t)idx); continue; }     // non-returning 32-bit atomic into an array of cap / 4 words
        if (MODE == 7) { atomicMin(reinterpret_cast<unsigned long long*>(&slots[h * 4]), (unsigned long long)k); continue; }   // non-returning 64-bit atomic on the slot
        if (MODE == 2) { slots[h * 4] = k; slots[h * 4 + 1] = p0[idx]; slots[h * 4 + 2] = p1[idx]; slots[h * 4 + 3] = idx; continue; }
        for (;;) {
            unsigned long long old;
            if (MODE == 4) {
                unsigned long long expect = (unsigned long long)EMPTY_KEY;
                __hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long*>(&slots[h * 4]), &expect, (unsigned long long)k,
                                                     __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = expect;
            } else old = atomicCAS(reinterpret_cast<unsigned long long*>(&slots[h * 4]), (unsigned long long)EMPTY_KEY, (unsigned long long)k);
            if (old == (unsigned long long)EMPTY_KEY) {
                if (MODE != 1) { slots[h * 4 + 1] = p0[idx]; slots[h * 4 + 2] = p1[idx]; slots[h * 4 + 3] = idx; }
                break;
            }
            if ((int64_t)old == k) { acc++; break; }
            h = (h + 1) & mask;
        }
    }
    if (acc == 0x7fffffff) *sink = acc;
}

int main() {
    int dev = 0; CK(hipSetDevice(dev));
    struct Case { const char* what; int64_t nrows; uint32_t per_seg_pct; };
    const Case cases[] = {{"partsupp 8M rows, 5.4% staged", 8000000, 54}, {"supplier 100K rows, all staged", 100000, 1000}, {"8M rows, all staged", 8000000, 1000}};
    for (const Case& c : cases) {
        const int64_t target = 256 * 24, gran = 512;
        int64_t seg_rows = (c.nrows + target - 1) / target; seg_rows = std::max<int64_t>(gran, (seg_rows + gran - 1) / gran * gran);
        const int nseg = (int)((c.nrows + seg_rows - 1) / seg_rows);
        const uint32_t per_seg = (uint32_t)(seg_rows * c.per_seg_pct / 1000);
        const uint64_t staged = (uint64_t)per_seg * nseg;
        uint64_t cap = 1024; while (cap < 2 * staged) cap <<= 1;
        int64_t *key, *p0, *p1, *slots; uint32_t* sc; int* sink;
        CK(hipMalloc(&key, c.nrows * 8 + seg_rows * 8)); CK(hipMalloc(&p0, c.nrows * 8 + seg_rows * 8)); CK(hipMalloc(&p1, c.nrows * 8 + seg_rows * 8));
        CK(hipMalloc(&slots, (cap + 1) * 32)); CK(hipMalloc(&sc, nseg * 4 + 64)); CK(hipMalloc(&sink, 64));
        const unsigned grid = (nseg + 3) / 4;
        gen<<<grid, TPB>>>(key, p0, p1, sc, nseg, seg_rows, per_seg);
        CK(hipDeviceSynchronize());
        printf("%s: nseg %d seg_rows %lld staged %llu cap %llu (%.1f MB of slots)\n", c.what, nseg, (long long)seg_rows, (unsigned long long)staged, (unsigned long long)cap, cap * 32 / 1e6);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const char* names[] = {"cas64+stores", "cas only", "stores only", "read only", "wg-scope cas+stores", "cas32 on rowref+stores", "atomicMin32 no return", "atomicMin64 no return"};
        for (int mode = 0; mode < 8; ++mode) {
            float best = 1e9f, sum = 0; const int reps = 12;
            for (int r = 0; r < reps; ++r) {
                clear<<<1024, TPB>>>(slots, cap);
                CK(hipEventRecord(e0));
                switch (mode) {
                    case 0: insert<0><<<grid, TPB>>>(key, p0, p1, sc, nseg, seg_rows, slots, cap - 1, sink); break;
                    case 1: insert<1><<<grid, TPB>>>(key, p0, p1, sc, nseg, seg_rows, slots, cap - 1, sink); break;
                    case 2: insert<2><<<grid, TPB>>>(key, p0, p1, sc, nseg, seg_rows, slots, cap - 1, sink); break;
                    case 3: insert<3><<<grid, TPB>>>(key, p0, p1, sc, nseg, seg_rows, slots, cap - 1, sink); break;
                    case 5: insert<5><<<grid, TPB>>>(key, p0, p1, sc, nseg, seg_rows, slots, cap - 1, sink); break;
                    case 6: insert<6><<<grid, TPB>>>(key, p0, p1, sc, nseg, seg_rows, slots, cap - 1, sink); break;
                    case 7: insert<7><<<grid, TPB>>>(key, p0, p1, sc, nseg, seg_rows, slots, cap - 1, sink); break;
                    default: insert<4><<<grid, TPB>>>(key, p0, p1, sc, nseg, seg_rows, slots, cap - 1, sink); break;
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) { best = std::min(best, ms); sum += ms; }
            }
            printf("   %-22s avg %.4f ms  min %.4f ms\n", names[mode], sum / 11, best);
        }
        CK(hipFree(key)); CK(hipFree(p0)); CK(hipFree(p1)); CK(hipFree(slots)); CK(hipFree(sc)); CK(hipFree(sink));
    }
    return 0;
}
