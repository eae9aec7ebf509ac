// Launch floor of a dependent chain of tiny kernels on one stream (what a join query's small builds pay):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_launch tools/microbench_launch.hip && /tmp/mb_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { char b[2048]; };
__global__ void k_empty() {}
__global__ void k_args(Big a, int* out) { if (a.b[0] == 77 && out) *out = 1; }
__global__ void k_touch(int* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1; }
int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 4096);
    hipMemset(d, 0, 4096);
    Big big{};
    for (int chain : {1, 2, 22, 100}) {
        for (int variant = 0; variant < 3; ++variant) {
            double best = 1e9;
            for (int rep = 0; rep < 50; ++rep) {
                hipStreamSynchronize(s);
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < chain; ++i) {
                    if (variant == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
                    else if (variant == 1) hipLaunchKernelGGL(k_args, dim3(1), dim3(64), 0, s, big, d);
                    else hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, s, d);
                }
                auto t1 = std::chrono::steady_clock::now();
                hipStreamSynchronize(s);
                auto t2 = std::chrono::steady_clock::now();
                const double submit = std::chrono::duration<double, std::micro>(t1 - t0).count(), total = std::chrono::duration<double, std::micro>(t2 - t0).count();
                if (total < best) { best = total; if (rep == 49 || true) { } }
                if (rep == 49) std::printf("chain %3d  %-8s  best total %7.1f us  (%.2f us per launch)   last: submit %.1f us, total %.1f us\n", chain,
                                           variant == 0 ? "empty" : variant == 1 ? "2KB args" : "touch", best, best / chain, submit, total);
            }
        }
    }
    return 0;
}
