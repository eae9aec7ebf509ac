// How a wave's streamed loads are shaped against what HBM delivers (MI355X): one 4-byte column of N rows read once by
// 256 CUs x 16 waves, every wave two 512-row steps (4 KiB) in flight, summed so nothing is dropped.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_shapes tools/stream_shapes.hip && /tmp/stream_shapes
// Shapes:  A  lane = 16 B, the wave's instruction covers 1 KiB contiguous (the float4 copy's shape), four in flight
//          B  lane = 32 B as two adjacent 16-byte loads (8 consecutive rows per lane: the tight skeletons' shape for 4-byte columns)
//          C  A with two double steps (eight loads, 8 KiB per wave) in flight
//          D  B, every wave walking a contiguous segment of its own instead of the tiled walk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

using U4 = uint32_t __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ uint32_t fold(U4 v) { return v.x + v.y + v.z + v.w; }

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const uint32_t* __restrict__ col, int64_t n, uint32_t* out, int waves, int how) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t steps = n / 1024;                                    // 1024-row double steps
    uint32_t acc = 0;
    if (SHAPE == 3) {
        const int64_t per = (steps + waves - 1) / waves;
        int64_t s0 = (int64_t)w * per, s1 = s0 + per; if (s1 > steps) s1 = steps;
        for (int64_t s = s0; s < s1; ++s) {
            const U4* p = reinterpret_cast<const U4*>(col + s * 1024);
            U4 a0 = p[2 * lane], a1 = p[2 * lane + 1], b0 = p[128 + 2 * lane], b1 = p[128 + 2 * lane + 1];
            acc += fold(a0) + fold(a1) + fold(b0) + fold(b1);
        }
    } else {
        for (int64_t s = w; s < steps; s += waves) {
            const U4* p = reinterpret_cast<const U4*>(col + s * 1024);  // 256 U4 per double step
            U4 a0, a1, b0, b1;
            if (SHAPE == 0) { a0 = p[lane]; a1 = p[64 + lane]; b0 = p[128 + lane]; b1 = p[192 + lane]; }
            if (SHAPE == 1 || SHAPE == 4) { a0 = p[2 * lane]; a1 = p[2 * lane + 1]; b0 = p[128 + 2 * lane]; b1 = p[128 + 2 * lane + 1]; }
            if (SHAPE == 2) {                                           // eight loads in flight: this double step and the one `waves` further on
                a0 = p[lane]; a1 = p[64 + lane]; b0 = p[128 + lane]; b1 = p[192 + lane];
                if (s + waves < steps) { const U4* q = reinterpret_cast<const U4*>(col + (s + waves) * 1024);
                    U4 c0 = q[lane], c1 = q[64 + lane], d0 = q[128 + lane], d1 = q[192 + lane]; acc += fold(c0) + fold(c1) + fold(d0) + fold(d1); }
                s += waves;
            }
            acc += fold(a0) + fold(a1) + fold(b0) + fold(b1);
        }
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (how == 0) { if (lane == 0) atomicAdd(out, acc); }                           // every wave one atomic on ONE address
    else if (how == 1) { if (lane == 0) out[1 + w] = acc; }                        // every wave a word of its own
    else if (how == 2) { if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(out) + 1 + (w & 7), (unsigned long long)acc); }   // 8 addresses, 64-bit
    else if (how == 3) {                                                           // one atomic per workgroup (LDS first)
        __shared__ uint32_t s; if (threadIdx.x == 0) s = 0; __syncthreads();
        if (lane == 0) atomicAdd(&s, acc); __syncthreads();
        if (threadIdx.x == 0) atomicAdd(out, s);
    }
}

int main() {
    const int64_t n = 60003415 / 1024 * 1024;
    uint32_t *col, *out;
    hipMalloc(&col, n * 4); hipMalloc(&out, 4 << 16);
    std::vector<uint32_t> h(n); for (int64_t i = 0; i < n; ++i) h[i] = (uint32_t)(i * 2654435761u >> 7);
    hipMemcpy(col, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"A 16B/lane contiguous x4", "B 32B/lane adjacent pairs", "C A, eight loads in flight", "D B, segment walk"};
    const char* hows[] = {"1 atomic/wave, one address", "a word per wave", "1 atomic/wave, 8 addresses (64-bit)", "1 atomic/workgroup"};
    for (int how = 0; how < 4; ++how)
    for (int wpc : {4, 8, 16, 32}) {
        const int waves = 256 * wpc, blocks = waves / 4;
        for (int shape = 0; shape < 4; ++shape) {
            if (how != 1 && shape != 1) continue;
            float best = 1e9f;
            for (int it = 0; it < 12; ++it) {
                hipMemsetAsync(out, 0, 4, 0);
                hipEventRecord(e0, 0);
                if (shape == 0) k<0><<<blocks, 256>>>(col, n, out, waves, how);
                if (shape == 1) k<1><<<blocks, 256>>>(col, n, out, waves, how);
                if (shape == 2) k<2><<<blocks, 256>>>(col, n, out, waves, how);
                if (shape == 3) k<3><<<blocks, 256>>>(col, n, out, waves, how);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 2 && ms < best) best = ms;
            }
            uint32_t r; hipMemcpy(&r, out, 4, hipMemcpyDeviceToHost);
            printf("%-36s waves/CU %2d  %-28s %.4f ms  %.0f GB/s  (sum %u)\n", hows[how], wpc, names[shape], best, n * 4 / best * 1e-6, r);
        }
    }
    return 0;
}
