#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 2, NV = 1, NSC = 0, NSOP = 0, ND = 0;
    struct Regs { uint32_t c0[16]; uint32_t c1[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
        xt_load<8, TAIL>(a.col[1], r, nrows, s.c1);
    }
    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {
        bool p = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 >= v1);
        p = p & v2;
        const int64_t v3 = a.ci[1];
        const bool v4 = (v0 <= v3);
        p = p & v4;
        const int64_t v5 = xt_i64(s.c1, i);
        p = p && x_may_hit(a.tab[0], v5, false);
        return p;
    }
    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {
        bool p = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 >= v1);
        p = p & v2;
        const int64_t v3 = a.ci[1];
        const bool v4 = (v0 <= v3);
        p = p & v4;
        const int64_t v5 = xt_i64(s.c1, i);
        const bool in = (v5 >= a.tab[0].bm_lo) & (v5 <= a.tab[0].bm_hi);
        p = p & in;
        const uint64_t off = p ? (uint64_t)(v5 - a.tab[0].bm_lo) : 0ull;
        widx = (uint32_t)(off >> 5); bit = (uint32_t)off & 31u;
        return p;
    }
    static constexpr bool PREF32 = false, PWIN = false;
    __device__ __forceinline__ static bool spre32(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& off) {
        off = 0; return false;
    }
    __device__ __forceinline__ static const uint32_t* sbitmap(const XArgs& a) { return x_prefilter_bitmap(a.tab[0], false); }
    template <int H> __device__ __forceinline__ static bool eval_regs(const XArgs& a, const Pair<int64_t> (&s)[2], int64_t r, XOut<NV>& o) {
        return false;
    }
    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[1], XOut<NV>& o) {
        int64_t v5 = static_cast<const int64_t*>(a.col[1])[r];
        int64_t v7 = static_cast<const int64_t*>(a.col[2])[r];
        x_pin(v5, v7);
        const uint32_t e6 = x_lookup_l<0x80000043u>(a.tab[0], v5, false);
        const bool v6 = (e6 != NO_ROW);
        int64_t v12 = x_field(a.tab[0], 0, e6);
        x_pin(v12);
        if (!v6) return false;
        const uint32_t e8 = x_lookup_l<0x80000043u>(a.tab[1], v7, false);
        const bool v8 = (e8 != NO_ROW);
        int64_t v9 = x_field(a.tab[1], 0, e8);
        x_pin(v9);
        if (!v8) return false;
        const int64_t v10 = a.ci[2];
        const bool v11 = (v9 == v10);
        const int64_t v13 = a.ci[3];
        const bool v14 = (v12 == v13);
        const bool v15 = (v11 && v14);
        const bool v16 = (v9 == v13);
        const bool v17 = (v12 == v10);
        const bool v18 = (v16 && v17);
        const bool v19 = (v15 || v18);
        if (!v19) return false;
        int64_t v0 = static_cast<const int64_t*>(a.col[0])[r];
        double v28 = static_cast<const double*>(a.col[3])[r];
        double v30 = static_cast<const double*>(a.col[4])[r];
        x_pin(v0, v28, v30);
        const int64_t v21 = a.ci[4];
        const int64_t v22 = (v9 * v21);
        const int64_t v23 = (v22 + v12);
        const int64_t v26 = (v23 * v10);
        const int64_t v20 = (v0 / 10000);
        const int64_t v24 = a.ci[5];
        const int64_t v25 = (v20 - v24);
        const int64_t v27 = (v26 + v25);
        o.key = v27; o.bad = false;
        const double v29 = a.cf[0];
        const double v31 = (v29 - v30);
        const double v32 = (v28 * v31);
        o.val[0] = x_bits(v32);
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_group_tight(XArgs a, XGroup<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue8<P, XGroup, false>(a, s, nrows, seg_rows, nseg);
}
