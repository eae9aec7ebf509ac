#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 2, ND = 0;
    struct Regs { uint32_t c0[16]; uint32_t c1[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
        xt_load<8, TAIL>(a.col[1], r, nrows, s.c1);
    }
    __device__ __forceinline__ static bool eval(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        bool pass = true;
        const double v0 = x_f(xt_i64(s.c0, i));
        const double v1 = a.cf[0];
        const bool v2 = (v0 > v1);
        pass = pass & v2;
        const int64_t v3 = xt_i64(s.c1, i);
        const int64_t v4 = a.ci[0];
        const bool v6 = (v3 >= v4);
        const int64_t v5 = a.ci[1];
        const bool v7 = (v3 <= v5);
        const bool v8 = (v6 && v7);
        const int64_t v9 = a.ci[2];
        const bool v11 = (v3 >= v9);
        const int64_t v10 = a.ci[3];
        const bool v12 = (v3 <= v10);
        const bool v13 = (v11 && v12);
        const bool v39 = (v8 || v13);
        const int64_t v14 = a.ci[4];
        const bool v16 = (v3 >= v14);
        const int64_t v15 = a.ci[5];
        const bool v17 = (v3 <= v15);
        const bool v18 = (v16 && v17);
        const bool v40 = (v39 || v18);
        const int64_t v19 = a.ci[6];
        const bool v21 = (v3 >= v19);
        const int64_t v20 = a.ci[7];
        const bool v22 = (v3 <= v20);
        const bool v23 = (v21 && v22);
        const bool v41 = (v40 || v23);
        const int64_t v24 = a.ci[8];
        const bool v26 = (v3 >= v24);
        const int64_t v25 = a.ci[9];
        const bool v27 = (v3 <= v25);
        const bool v28 = (v26 && v27);
        const bool v42 = (v41 || v28);
        const int64_t v29 = a.ci[10];
        const bool v31 = (v3 >= v29);
        const int64_t v30 = a.ci[11];
        const bool v32 = (v3 <= v30);
        const bool v33 = (v31 && v32);
        const bool v43 = (v42 || v33);
        const int64_t v34 = a.ci[12];
        const bool v36 = (v3 >= v34);
        const int64_t v35 = a.ci[13];
        const bool v37 = (v3 <= v35);
        const bool v38 = (v36 && v37);
        const bool v44 = (v43 || v38);
        pass = pass & v44;
        o.key = 0; o.bad = false;
        o.val[0] = x_bits(v0);
        const double v45 = a.cf[1];
        o.val[1] = x_bits(v45);
        o.ent = NO_ROW;
        return pass;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_sum_tight(XArgs a, XSum<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_tight<P, XSum>(a, s, nrows);
}
