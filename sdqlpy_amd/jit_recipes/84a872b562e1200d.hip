#define XGL_IVAL 0
#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 4, ND = 2;
    struct Regs { uint32_t c0[4]; uint32_t c1[2]; uint32_t c2[2]; uint32_t c3[2]; uint32_t c4[8]; uint32_t c5[2]; uint32_t c6[2]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
        for (int i = threadIdx.x; i < 256; i += TPB) {
            int64_t cell = 0;
            if (i < a.ndict[5]) {
                const int64_t dv = a.dict[5][i];
        const double v10 = a.cf[0];
        const double v11 = x_f(dv);
        const double v12 = (v10 - v11);
                cell = x_bits(v12);
            }
            tab[0][i] = cell;
        }
        for (int i = threadIdx.x; i < 256; i += TPB) {
            int64_t cell = 0;
            if (i < a.ndict[6]) {
                const int64_t dv = a.dict[6][i];
        const double v10 = a.cf[0];
        const double v14 = x_f(dv);
        const double v15 = (v10 + v14);
                cell = x_bits(v15);
            }
            tab[1][i] = cell;
        }
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<2, TAIL>(a.code[0], r, nrows, s.c0);
        xt_load<1, TAIL>(a.code[1], r, nrows, s.c1);
        xt_load<1, TAIL>(a.code[2], r, nrows, s.c2);
        xt_load<1, TAIL>(a.code[3], r, nrows, s.c3);
        xt_load<4, TAIL>(a.ncol[4], r, nrows, s.c4);
        xt_load<1, TAIL>(a.code[5], r, nrows, s.c5);
        xt_load<1, TAIL>(a.code[6], r, nrows, s.c6);
    }
    __device__ __forceinline__ static bool eval(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        bool pass = true;
        const bool v2 = (xt_u16(s.c0, i) < a.cc[0]);
        pass = pass & v2;
        const int64_t v3 = ((int64_t)xt_u8(s.c1, i) + a.dlo[1]);
        const int64_t v5 = a.ci[1];
        const int64_t v6 = (int64_t)__mul24((int)v3, (int)v5);
        const int64_t v4 = ((int64_t)xt_u8(s.c2, i) + a.dlo[2]);
        const int64_t v7 = (int64_t)((int32_t)v6 + (int32_t)v4);
        o.key = v7; o.bad = false;
        o.val[0] = (int64_t)xt_u8(s.c3, i) + a.dlo[3];
        const double v9 = narrow_decode(xt_i32(s.c4, i));
        o.val[1] = x_bits(v9);
        const double v12 = x_f(tab[0][xt_u8(s.c5, i)]);
        const double v13 = (v9 * v12);
        o.val[2] = x_bits(v13);
        const double v15 = x_f(tab[1][xt_u8(s.c6, i)]);
        const double v16 = (v13 * v15);
        o.val[3] = x_bits(v16);
        o.ent = NO_ROW;
        return pass;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_group_lane_tight(XArgs a, XGroupLane<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_tight<P, XGroupLane>(a, s, nrows);
}
