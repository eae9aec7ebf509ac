#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 0, ND = 0;
    struct Regs { uint32_t c0[16]; uint32_t c1[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
        xt_load<8, TAIL>(a.col[1], r, nrows, s.c1);
    }
    __device__ __forceinline__ static bool eval(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        bool pass = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = (v0 / (int64_t)750000ll);
        const int64_t v11 = a.ci[1];
        const int64_t v12 = (v1 * v11);
        const int64_t v2 = (v0 / (int64_t)5000ll);
        const int64_t v3 = (v2 % (int64_t)150ll);
        const int64_t v13 = (v12 + v3);
        const int64_t v15 = a.ci[2];
        const int64_t v16 = (v13 * v15);
        const int64_t v4 = (v0 / (int64_t)100ll);
        const int64_t v5 = (v4 % (int64_t)50ll);
        const int64_t v6 = a.ci[0];
        const int64_t v7 = (v5 + v6);
        const int64_t v14 = (v7 - v6);
        const int64_t v17 = (v16 + v14);
        o.key = v17; o.bad = false;
        o.ent = NO_ROW;
        return pass;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_group_tight(XArgs a, XGroup<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_tight<P, XGroup>(a, s, nrows);
}
