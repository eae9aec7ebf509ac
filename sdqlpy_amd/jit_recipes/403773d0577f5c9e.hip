#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 1, NV = 1, NSC = 0, NSOP = 0, ND = 0;
    struct Regs { uint32_t c0[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
    }
    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {
        bool p = true;
        const int64_t v0 = xt_i64(s.c0, i);
        p = p && x_may_hit(a.tab[0], v0, false);
        return p;
    }
    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {
        bool p = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const bool in = (v0 >= a.tab[0].bm_lo) & (v0 <= a.tab[0].bm_hi);
        p = p & in;
        const uint64_t off = p ? (uint64_t)(v0 - a.tab[0].bm_lo) : 0ull;
        widx = (uint32_t)(off >> 5); bit = (uint32_t)off & 31u;
        return p;
    }
    static constexpr bool PREF32 = false, PWIN = false;
    __device__ __forceinline__ static bool spre32(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& off) {
        off = 0; return false;
    }
    __device__ __forceinline__ static const uint32_t* sbitmap(const XArgs& a) { return x_prefilter_bitmap(a.tab[0], false); }
    template <int H> __device__ __forceinline__ static bool eval_regs(const XArgs& a, const Pair<int64_t> (&s)[1], int64_t r, XOut<NV>& o) {
        return false;
    }
    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[1], XOut<NV>& o) {
        int64_t v0 = static_cast<const int64_t*>(a.col[0])[r];
        int64_t v2 = static_cast<const int64_t*>(a.col[1])[r];
        int64_t v5 = static_cast<const int64_t*>(a.col[2])[r];
        double v12 = static_cast<const double*>(a.col[3])[r];
        x_pin(v0, v2, v5, v12);
        const uint32_t e1 = x_lookup_l<0x80000082u>(a.tab[0], v0, false);
        const bool v1 = (e1 != NO_ROW);
        int64_t v10 = x_field(a.tab[0], 0, e1);
        x_pin(v10);
        if (!v1) return false;
        const int64_t v3 = a.ci[0];
        const bool v4 = (v2 == v3);
        if (!v4) return false;
        const int64_t v6 = a.ci[1];
        const bool v7 = (v5 == v6);
        const bool v8 = (a.ci[2] != 0);
        const bool v9 = (v7 || v8);
        if (!v9) return false;
        const bool v11 = (v10 == v3);
        const double v13 = (double)v3;
        const bool v14 = (v12 >= v13);
        const bool v18 = (v11 && v14);
        const int64_t v15 = a.ci[3];
        const double v16 = (double)v15;
        const bool v17 = (v12 <= v16);
        const bool v19 = (v18 && v17);
        const int64_t v20 = a.ci[4];
        const bool v21 = (v10 == v20);
        const int64_t v22 = a.ci[5];
        const double v23 = (double)v22;
        const bool v24 = (v12 >= v23);
        const bool v28 = (v21 && v24);
        const int64_t v25 = a.ci[6];
        const double v26 = (double)v25;
        const bool v27 = (v12 <= v26);
        const bool v29 = (v28 && v27);
        const bool v39 = (v19 || v29);
        const int64_t v30 = a.ci[7];
        const bool v31 = (v10 == v30);
        const double v32 = (double)v25;
        const bool v33 = (v12 >= v32);
        const bool v37 = (v31 && v33);
        const int64_t v34 = a.ci[8];
        const double v35 = (double)v34;
        const bool v36 = (v12 <= v35);
        const bool v38 = (v37 && v36);
        const bool v40 = (v39 || v38);
        if (!v40) return false;
        double v41 = static_cast<const double*>(a.col[4])[r];
        double v43 = static_cast<const double*>(a.col[5])[r];
        x_pin(v41, v43);
        o.key = 0; o.bad = false;
        const double v42 = a.cf[0];
        const double v44 = (v42 - v43);
        const double v45 = (v41 * v44);
        o.val[0] = x_bits(v45);
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_sum_tight(XArgs a, XSum<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue8<P, XSum, false>(a, s, nrows, seg_rows, nseg);
}
