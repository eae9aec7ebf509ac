#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 1, ND = 2;
    struct Regs { uint32_t c0[2]; uint32_t c1[2]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
        for (int i = threadIdx.x; i < 256; i += TPB) {
            int64_t cell = 0;
            if (i < a.ndict[1]) {
                const int64_t dv = a.dict[1][i];
        const int64_t v3 = dv;
                cell = v3;
            }
            tab[0][i] = cell;
        }
        for (int i = threadIdx.x; i < 256; i += TPB) {
            int64_t cell = 0;
            if (i < a.ndict[0]) {
                const int64_t dv = a.dict[0][i];
        const double v0 = x_f(dv);
        const double v4 = a.cf[1];
        const double v5 = (v4 - v0);
        const double v6 = (v0 * v5);
                cell = x_bits(v6);
            }
            tab[1][i] = cell;
        }
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<1, TAIL>(a.code[0], r, nrows, s.c0);
        xt_load<1, TAIL>(a.code[1], r, nrows, s.c1);
    }
    __device__ __forceinline__ static bool eval(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        bool pass = true;
        const bool v2 = (xt_u8(s.c0, i) >= a.cc[0]);
        pass = pass & v2;
        const int64_t v3 = tab[0][xt_u8(s.c1, i)];
        o.key = v3; o.bad = false;
        const double v6 = x_f(tab[1][xt_u8(s.c0, i)]);
        o.val[0] = x_bits(v6);
        o.ent = NO_ROW;
        return pass;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_group_lane_tight(XArgs a, XGroupLane<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_tight<P, XGroupLane>(a, s, nrows);
}
