#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 1, NV = 1, NSC = 0, NSOP = 0, ND = 0;
    struct Regs { uint32_t c0[8]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<4, TAIL>(a.ncol[0], r, nrows, s.c0);
    }
    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {
        bool p = true;
        const int64_t v0 = (int64_t)xt_i32(s.c0, i);
        p = p && x_may_hit(a.tab[0], v0, false);
        return p;
    }
    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {
        bool p = true;
        const int64_t v0 = (int64_t)xt_i32(s.c0, i);
        const bool in = (v0 >= a.tab[0].bm_lo) & (v0 <= a.tab[0].bm_hi);
        p = p & in;
        const uint64_t off = in ? (uint64_t)(v0 - a.tab[0].bm_lo) : 0ull;
        widx = (uint32_t)(off >> 5); bit = (uint32_t)off & 31u;
        return p;
    }
    static constexpr bool PREF32 = true, PWIN = false;
    __device__ __forceinline__ static bool spre32(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& off) {
        bool p = true;
        const int64_t v0 = (int64_t)xt_i32(s.c0, i);
        const uint32_t o32 = (uint32_t)((int32_t)v0 - (int32_t)a.tab[0].bm_lo);
        const bool in = o32 <= (uint32_t)(a.tab[0].bm_hi - a.tab[0].bm_lo);
        p = p & in;
        off = in ? o32 : 0u;
        return p;
    }
    __device__ __forceinline__ static const uint32_t* sbitmap(const XArgs& a) { return x_prefilter_bitmap(a.tab[0], false); }
    template <int H> __device__ __forceinline__ static bool eval_regs(const XArgs& a, const Pair<int64_t> (&s)[1], int64_t r, XOut<NV>& o) {
        return false;
    }
    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[1], XOut<NV>& o) {
        int64_t v0 = (int64_t)static_cast<const int32_t*>(a.ncol[0])[r];
        int64_t v2 = (int64_t)static_cast<const int32_t*>(a.ncol[1])[r];
        x_pin(v0, v2);
        const uint32_t e1 = x_lookup_l<0x80000082u>(a.tab[0], v0, false);
        const bool v1 = (e1 != NO_ROW);
        int64_t v3 = x_field(a.tab[0], 1, e1);
        x_pin(v3);
        if (!v1) return false;
        const bool b4 = false || false || v2 < 0 || v2 > 0xFFFFFFFFll || v3 < 0 || v3 > 0xFFFFFFFFll;
        const int64_t v4 = (int64_t)(((uint64_t)v2 << 32) | ((uint64_t)v3 & 0xFFFFFFFFull));
        const uint32_t e5 = x_lookup_l<0x80000052u>(a.tab[1], v4, b4);
        const bool v5 = (e5 != NO_ROW);
        if (!v5) return false;
        int64_t v6 = x_field(a.tab[0], 0, e1);
        double v7 = narrow_decode(static_cast<const int32_t*>(a.ncol[2])[r]);
        double v9 = narrow_decode(static_cast<const int32_t*>(a.ncol[3])[r]);
        x_pin(v6, v7, v9);
        o.key = v6; o.bad = false;
        const double v8 = a.cf[0];
        const double v10 = (v8 - v9);
        const double v11 = (v7 * v10);
        o.val[0] = x_bits(v11);
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_group_tight(XArgs a, XGroup<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue8<P, XGroup, false>(a, s, nrows, seg_rows, nseg);
}
