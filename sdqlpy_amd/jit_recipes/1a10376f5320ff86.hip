#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 1, ND = 0;
    struct Regs { uint32_t c0[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
    }
    __device__ __forceinline__ static bool eval(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        bool pass = true;
        const double v0 = x_f(xt_i64(s.c0, i));
        const double v1 = a.cf[0];
        const bool v2 = (v0 > v1);
        pass = pass & v2;
        o.key = 0; o.bad = false;
        o.val[0] = x_bits(v0);
        o.ent = NO_ROW;
        return pass;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_sum_tight(XArgs a, XSum<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_tight<P, XSum>(a, s, nrows);
}
