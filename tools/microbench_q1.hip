// Standalone tuning harness for the two headline streaming kernels (Q1 group-by, Q6 scan):
// synthetic SF=10-shaped columns generated on the device, kernels timed with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -Iinclude -Isdqlpy_amd/csrc
//         [-DSDQH_UNROLL=4] [-DSDQH_NT_LOADS=1] [-DSDQH_TILE_CHUNK=4] tools/microbench_q1.hip -o mb && ./mb [resident_per_cu]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "sdqh_kernels.hpp"
using namespace sdqh;

__global__ void gen(int64_t n, int64_t* ship, double* qty, double* ep, double* disc, double* tax, uint32_t* rf, uint32_t* ls) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        uint64_t h = mix64((uint64_t)i * 0x9E3779B97F4A7C15ull + 12345);
        int year = 1992 + (int)(h % 7), mon = 1 + (int)((h >> 8) % 12), day = 1 + (int)((h >> 16) % 28);
        ship[i] = (int64_t)year * 10000 + mon * 100 + day;
        qty[i] = 1 + (double)((h >> 24) % 50); ep[i] = 900.0 + (double)((h >> 32) % 100000) / 100.0;
        disc[i] = (double)((h >> 40) % 11) / 100.0; tax[i] = (double)((h >> 44) % 9) / 100.0;
        const uint32_t flags[3] = {'A', 'N', 'R'};
        rf[i] = flags[(h >> 48) % 3]; ls[i] = ((h >> 52) & 1) ? 'O' : 'F';
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int64_t n = 60003415;
    int resident = argc > 1 ? atoi(argv[1]) : 0;
    int64_t* ship; double *qty, *ep, *disc, *tax; uint32_t *rf, *ls;
    CK(hipMalloc(&ship, n * 8 + 64)); CK(hipMalloc(&qty, n * 8 + 64)); CK(hipMalloc(&ep, n * 8 + 64)); CK(hipMalloc(&disc, n * 8 + 64));
    CK(hipMalloc(&tax, n * 8 + 64)); CK(hipMalloc(&rf, n * 4 + 64)); CK(hipMalloc(&ls, n * 4 + 64));
    hipLaunchKernelGGL(gen, dim3(4096), dim3(256), 0, 0, n, ship, qty, ep, disc, tax, rf, ls);
    CK(hipDeviceSynchronize());
    using FC = FCfg<1, 0, 0, 0>; using KC = KCfg<1, 1>;
    auto kq1 = k_groupby_reg<SDQH_TUPLE_PRICING, 8, FC, KC>;
    auto kq6 = k_scan_sum<SDQH_TUPLE_AB, FCfg<1, 1, 0, 0>>;
    int occ1 = 0, occ6 = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ1, kq1, TPB, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ6, kq6, TPB, 0));
    const int r1 = resident ? resident : std::min(occ1, 6), r6 = resident ? resident : std::min(occ6, 6);
    const unsigned g1 = 256 * r1, g6 = 256 * r6;
    DevFilter f1{}; f1.ni = 1; f1.ic[0] = ship; f1.ilo[0] = INT64_MIN; f1.ihi[0] = 19980902;
    DevTuple t1{}; t1.op[0] = qty; t1.op[1] = ep; t1.op[2] = disc; t1.op[3] = tax;
    DevGroupKeys gk{}; gk.col[0] = rf; gk.col[1] = ls; gk.is_str[0] = gk.is_str[1] = 1; gk.nkeys = 2;
    DevFilter f6{}; f6.ni = 1; f6.ic[0] = ship; f6.ilo[0] = 19940101; f6.ihi[0] = 19941231; f6.nf = 1; f6.fc[0] = qty; f6.flo[0] = -1e300; f6.fhi[0] = 23.999;
    f6.omask = 2; f6.olo[1] = 0.05; f6.ohi[1] = 0.07;
    DevTuple t6{}; t6.op[0] = ep; t6.op[1] = disc;
    unsigned long long* gkeys; double* pacc; int64_t* pcnt; int* flags; double* partial;
    CK(hipMalloc(&gkeys, 64 * 8)); CK(hipMalloc(&pacc, (size_t)4096 * 64 * 32)); CK(hipMalloc(&pcnt, (size_t)4096 * 64 * 8)); CK(hipMalloc(&flags, 64)); CK(hipMalloc(&partial, 4096 * 5 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> m1, m6;
    for (int it = 0; it < 24; ++it) {
        CK(hipMemset(gkeys, 0xFF, 64 * 8)); CK(hipMemset(flags, 0, 8));
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kq1, dim3(g1), dim3(TPB), 0, 0, f1, t1, gk, n, gkeys, pacc, pcnt, flags); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 4) m1.push_back(ms);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(kq6, dim3(g6), dim3(TPB), 0, 0, f6, t6, n, partial); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1)); if (it >= 4) m6.push_back(ms);
    }
    std::sort(m1.begin(), m1.end()); std::sort(m6.begin(), m6.end());
    printf("UNROLL=%d NT=%d CHUNK=%d | q1 groupby: occ %d grid %u  median %.4f ms min %.4f  -> %.0f GB/s | q6 scan: occ %d grid %u median %.4f min %.4f -> %.0f GB/s\n",
           SDQH_UNROLL, SDQH_NT_LOADS, SDQH_TILE_CHUNK, occ1, g1, m1[m1.size() / 2], m1[0], 48.0 * n / (m1[m1.size() / 2] * 1e-3) / 1e9,
           occ6, g6, m6[m6.size() / 2], m6[0], 32.0 * n / (m6[m6.size() / 2] * 1e-3) / 1e9);
    return 0;
}
