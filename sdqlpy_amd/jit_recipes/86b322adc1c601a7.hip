#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 1, ND = 1;
    struct Regs { uint32_t c0[4]; uint32_t c1[2]; uint32_t c2[2]; uint32_t c3[8]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
        for (int i = threadIdx.x; i < 256; i += TPB) {
            int64_t cell = 0;
            if (i < a.ndict[1]) {
                const int64_t dv = a.dict[1][i];
        const double v5 = x_f(dv);
                cell = x_bits(v5);
            }
            tab[0][i] = cell;
        }
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<2, TAIL>(a.code[0], r, nrows, s.c0);
        xt_load<1, TAIL>(a.code[1], r, nrows, s.c1);
        xt_load<1, TAIL>(a.code[2], r, nrows, s.c2);
        xt_load<4, TAIL>(a.ncol[3], r, nrows, s.c3);
    }
    __device__ __forceinline__ static bool eval(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        bool pass = true;
        const bool v2 = (xt_u16(s.c0, i) >= a.cc[0]);
        pass = pass & v2;
        const bool v4 = (xt_u16(s.c0, i) < a.cc[1]);
        pass = pass & v4;
        const bool v7 = (xt_u8(s.c1, i) >= a.cc[2]);
        pass = pass & v7;
        const bool v9 = (xt_u8(s.c1, i) < a.cc[3]);
        pass = pass & v9;
        const bool v12 = (xt_u8(s.c2, i) < a.cc[4]);
        pass = pass & v12;
        o.key = 0; o.bad = false;
        const double v13 = narrow_decode(xt_i32(s.c3, i));
        const double v5 = x_f(tab[0][xt_u8(s.c1, i)]);
        const double v14 = (v13 * v5);
        o.val[0] = x_bits(v14);
        o.ent = NO_ROW;
        return pass;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_sum_tight(XArgs a, XSum<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_tight<P, XSum>(a, s, nrows);
}
