#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 1, NV = 2, NSC = 0, NSOP = 0, ND = 0;
    struct Regs { uint32_t c0[2]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<1, TAIL>(a.code[0], r, nrows, s.c0);
    }
    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {
        bool p = true;
        const int64_t v0 = ((int64_t)xt_u8(s.c0, i) + a.dlo[0]);
        p = p && x_may_hit(a.tab[0], v0, false);
        return p;
    }
    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {
        bool p = true;
        const int64_t v0 = ((int64_t)xt_u8(s.c0, i) + a.dlo[0]);
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
        x_pin(v0);
        const uint32_t e1 = x_lookup_l<0x80000100u>(a.tab[0], v0, false);
        const bool v1 = (e1 != NO_ROW);
        if (!v1) return false;
        int64_t v2 = (int64_t)static_cast<const int32_t*>(a.ncol[1])[r];
        int64_t v3 = x_field(a.tab[0], 0, e1);
        x_pin(v2, v3);
        o.key = v2; o.bad = false;
        o.val[0] = v3;
        o.val[1] = v0;
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_build_tight(XArgs a, XStage<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue8<P, XStage, true>(a, s, nrows, seg_rows, nseg);
}
