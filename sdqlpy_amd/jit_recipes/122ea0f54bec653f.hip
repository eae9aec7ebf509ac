#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 1, ND = 0;
    struct Regs { uint32_t c0[16]; uint32_t c1[16]; uint32_t c2[16]; uint32_t c3[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
        xt_load<8, TAIL>(a.col[1], r, nrows, s.c1);
        xt_load<8, TAIL>(a.col[2], r, nrows, s.c2);
        xt_load<8, TAIL>(a.col[3], r, nrows, s.c3);
    }
    __device__ __forceinline__ static bool eval(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        bool pass = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 >= v1);
        pass = pass & v2;
        const int64_t v3 = a.ci[1];
        const bool v4 = (v0 < v3);
        pass = pass & v4;
        const double v5 = x_f(xt_i64(s.c1, i));
        const double v6 = a.cf[0];
        const bool v7 = (v5 >= v6);
        pass = pass & v7;
        const double v8 = a.cf[1];
        const bool v9 = (v5 <= v8);
        pass = pass & v9;
        const double v10 = x_f(xt_i64(s.c2, i));
        const double v11 = a.cf[2];
        const bool v12 = (v10 < v11);
        pass = pass & v12;
        o.key = 0; o.bad = false;
        const double v13 = x_f(xt_i64(s.c3, i));
        const double v14 = (v13 * v5);
        o.val[0] = x_bits(v14);
        o.ent = NO_ROW;
        return pass;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_sum_tight(XArgs a, XSum<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_tight<P, XSum>(a, s, nrows);
}
