#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 4, ND = 0;
    struct Regs { uint32_t c0[16]; uint32_t c1[16]; uint32_t c2[16]; uint32_t c3[16]; uint32_t c4[16]; uint32_t c5[16]; uint32_t c6[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
        xt_load<8, TAIL>(a.col[1], r, nrows, s.c1);
        xt_load<8, TAIL>(a.col[2], r, nrows, s.c2);
        xt_load<8, TAIL>(a.col[3], r, nrows, s.c3);
        xt_load<8, TAIL>(a.col[4], r, nrows, s.c4);
        xt_load<8, TAIL>(a.col[5], r, nrows, s.c5);
        xt_load<8, TAIL>(a.col[6], r, nrows, s.c6);
    }
    __device__ __forceinline__ static bool eval(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        bool pass = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 <= v1);
        pass = pass & v2;
        const int64_t v3 = xt_i64(s.c1, i);
        const int64_t v5 = a.ci[1];
        const int64_t v6 = (v3 * v5);
        const int64_t v4 = xt_i64(s.c2, i);
        const int64_t v7 = (v6 + v4);
        o.key = v7; o.bad = false;
        const double v8 = x_f(xt_i64(s.c3, i));
        o.val[0] = x_bits(v8);
        const double v9 = x_f(xt_i64(s.c4, i));
        o.val[1] = x_bits(v9);
        const double v10 = a.cf[0];
        const double v11 = x_f(xt_i64(s.c5, i));
        const double v12 = (v10 - v11);
        const double v13 = (v9 * v12);
        o.val[2] = x_bits(v13);
        const double v14 = x_f(xt_i64(s.c6, i));
        const double v15 = (v10 + v14);
        const double v16 = (v13 * v15);
        o.val[3] = x_bits(v16);
        o.ent = NO_ROW;
        return pass;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_group_lane_tight(XArgs a, XGroupLane<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_tight<P, XGroupLane>(a, s, nrows);
}
