#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 2, ND = 1, NL = 1;
    static constexpr bool Q32 = false;
    struct Regs { uint32_t c0[4]; uint32_t c1[3]; uint32_t c2[8]; uint32_t c3[2]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
        for (int i = threadIdx.x; i < 256; i += TPB) {
            int64_t cell = 0;
            if (i < a.ndict[3]) {
                const int64_t dv = a.dict[3][i];
        const double v6 = x_f(dv);
                cell = x_bits(v6);
            }
            tab[0][i] = cell;
        }
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<2, TAIL>(a.code[0], r, nrows, s.c0);
        xt_load_d8(a.dcol[1], r, s.c1);
        xt_load<4, TAIL>(a.ncol[2], r, nrows, s.c2);
        xt_load<1, TAIL>(a.code[3], r, nrows, s.c3);
    }
    __device__ __forceinline__ static bool gates(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r) {
        bool p = true;
        const bool v2 = (xt_u16(s.c0, i) >= a.cc[0]);
        p = p & v2;
        return p;
    }
    __device__ __forceinline__ static uint32_t lkoff(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, const int l, bool& p) {
        if (l == 0) {
        const int64_t v3 = (int64_t)xt_d8(s.c1, i);
            const uint32_t o32 = (uint32_t)((int32_t)v3 - (int32_t)a.tab[0].bm_lo);
            p = p & (o32 <= (uint32_t)(a.tab[0].bm_hi - a.tab[0].bm_lo));
            return o32;
        }
        return 0u;
    }
    __device__ __forceinline__ static const uint32_t* lkbm(const XArgs& a, int l) { return a.tab[0].bm; }
    __device__ __forceinline__ static void row(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        const int64_t v3 = (int64_t)xt_d8(s.c1, i);
        o.key = v3; o.bad = false;
        const double v5 = narrow_decode(xt_i32(s.c2, i));
        o.val[0] = x_bits(v5);
        const double v6 = x_f(tab[0][xt_u8(s.c3, i)]);
        o.val[1] = x_bits(v6);
        o.ent = NO_ROW;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_build_values(XArgs a, XStage<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_vstage8<P>(a, s, nrows, seg_rows, nseg);
}
