#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 4, NV = 2, NSC = 0, NSOP = 0, ND = 0;
    struct Regs { uint32_t c0[16]; uint32_t c1[16]; uint32_t c2[16]; uint32_t c3[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
        xt_load<8, TAIL>(a.col[1], r, nrows, s.c1);
        xt_load<8, TAIL>(a.col[2], r, nrows, s.c2);
        xt_load<8, TAIL>(a.col[3], r, nrows, s.c3);
    }
    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {
        bool p = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 >= v1);
        p = p & v2;
        const int64_t v3 = a.ci[1];
        const bool v4 = (v0 < v3);
        p = p & v4;
        const int64_t v5 = xt_i64(s.c1, i);
        const int64_t v6 = xt_i64(s.c2, i);
        const bool v7 = (v5 < v6);
        p = p & v7;
        const bool v8 = (v6 < v0);
        p = p & v8;
        const int64_t v9 = xt_i64(s.c3, i);
        p = p && x_may_hit(a.tab[0], v9, false);
        return p;
    }
    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {
        bool p = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 >= v1);
        p = p & v2;
        const int64_t v3 = a.ci[1];
        const bool v4 = (v0 < v3);
        p = p & v4;
        const int64_t v5 = xt_i64(s.c1, i);
        const int64_t v6 = xt_i64(s.c2, i);
        const bool v7 = (v5 < v6);
        p = p & v7;
        const bool v8 = (v6 < v0);
        p = p & v8;
        const int64_t v9 = xt_i64(s.c3, i);
        const bool in = (v9 >= a.tab[0].bm_lo) & (v9 <= a.tab[0].bm_hi);
        p = p & in;
        const uint64_t off = p ? (uint64_t)(v9 - a.tab[0].bm_lo) : 0ull;
        widx = (uint32_t)(off >> 5); bit = (uint32_t)off & 31u;
        return p;
    }
    static constexpr bool PREF32 = false, PWIN = false;
    __device__ __forceinline__ static bool spre32(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& off) {
        off = 0; return false;
    }
    __device__ __forceinline__ static const uint32_t* sbitmap(const XArgs& a) { return x_prefilter_bitmap(a.tab[0], false); }
    template <int H> __device__ __forceinline__ static bool eval_regs(const XArgs& a, const Pair<int64_t> (&s)[4], int64_t r, XOut<NV>& o) {
        return false;
    }
    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[1], XOut<NV>& o) {
        int64_t v9 = static_cast<const int64_t*>(a.col[3])[r];
        int64_t v11 = static_cast<const int64_t*>(a.col[4])[r];
        x_pin(v9, v11);
        const uint32_t e10 = x_lookup_l<0x80000001u>(a.tab[0], v9, false);
        const bool v10 = (e10 != NO_ROW);
        if (!v10) return false;
        const int64_t v12 = a.ci[2];
        const bool v13 = (v11 == v12);
        const int64_t v14 = a.ci[3];
        const bool v15 = (v11 == v14);
        const bool v16 = (v13 || v15);
        if (!v16) return false;
        int64_t v17 = x_field(a.tab[0], 0, e10);
        x_pin(v17);
        o.key = v11; o.bad = false;
        const int64_t v18 = a.ci[4];
        const bool v19 = (v17 == v18);
        const int64_t v20 = a.ci[5];
        const bool v21 = (v17 == v20);
        const bool v22 = (v19 || v21);
        const bool b23 = v22 ? false : false;
        const int64_t v23 = (v22 ? v20 : v18);
        const double v24 = (double)v23;
        o.val[0] = x_bits(v24);
        const bool v26 = (v17 >= v20);
        const int64_t v25 = a.ci[6];
        const bool v27 = (v17 <= v25);
        const bool v28 = (v26 && v27);
        const bool v29 = (v17 == v18);
        const bool v30 = (v17 == v12);
        const bool v34 = (v29 || v30);
        const int64_t v31 = a.ci[7];
        const bool v32 = (v17 == v31);
        const bool v35 = (v34 || v32);
        const bool v33 = (v17 == v25);
        const bool v36 = (v35 || v33);
        const bool v37 = (v28 && v36);
        const bool b38 = v37 ? false : false;
        const int64_t v38 = (v37 ? v20 : v18);
        const double v39 = (double)v38;
        o.val[1] = x_bits(v39);
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_group_tight(XArgs a, XGroup<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue8<P, XGroup, false>(a, s, nrows, seg_rows, nseg);
}
