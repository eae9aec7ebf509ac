#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 2, NV = 2, NSC = 0, NSOP = 0, ND = 0;
    struct Regs { uint32_t c0[4]; uint32_t c1[8]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<2, TAIL>(a.code[0], r, nrows, s.c0);
        xt_load<4, TAIL>(a.ncol[1], r, nrows, s.c1);
    }
    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {
        bool p = true;
        const bool v2 = (xt_u16(s.c0, i) >= a.cc[0]);
        p = p & v2;
        const bool v4 = (xt_u16(s.c0, i) < a.cc[1]);
        p = p & v4;
        const int64_t v5 = (int64_t)xt_i32(s.c1, i);
        p = p && x_may_hit(a.tab[0], v5, false);
        return p;
    }
    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {
        bool p = true;
        const bool v2 = (xt_u16(s.c0, i) >= a.cc[0]);
        p = p & v2;
        const bool v4 = (xt_u16(s.c0, i) < a.cc[1]);
        p = p & v4;
        const int64_t v5 = (int64_t)xt_i32(s.c1, i);
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
        int64_t v5 = (int64_t)static_cast<const int32_t*>(a.ncol[1])[r];
        x_pin(v5);
        const uint32_t e6 = x_lookup_l<0x80000100u>(a.tab[0], v5, false);
        const bool v6 = (e6 != NO_ROW);
        if (!v6) return false;
        int64_t v7 = (int64_t)static_cast<const int32_t*>(a.ncol[2])[r];
        int64_t v8 = x_field(a.tab[0], 0, e6);
        int64_t v9 = x_field(a.tab[0], 1, e6);
        x_pin(v7, v8, v9);
        o.key = v7; o.bad = false;
        o.val[0] = v8;
        o.val[1] = v9;
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_build_tight(XArgs a, XStage<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue8<P, XStage, true>(a, s, nrows, seg_rows, nseg);
}
