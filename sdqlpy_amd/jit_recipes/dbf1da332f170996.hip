#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NV = 2, ND = 0, NL = 0;
    static constexpr bool Q32 = false;
    struct Regs { uint32_t c0[16]; uint32_t c1[16]; uint32_t c2[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
        xt_load<8, TAIL>(a.col[1], r, nrows, s.c1);
        xt_load<8, TAIL>(a.col[2], r, nrows, s.c2);
    }
    __device__ __forceinline__ static bool gates(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r) {
        bool p = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 >= v1);
        p = p & v2;
        return p;
    }
    __device__ __forceinline__ static uint32_t lkoff(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, const int l, bool& p) {
        return 0u;
    }
    __device__ __forceinline__ static const uint32_t* lkbm(const XArgs& a, int l) { return nullptr; }
    __device__ __forceinline__ static void row(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {
        const int64_t v0 = xt_i64(s.c0, i);
        o.key = v0; o.bad = false;
        const int64_t v3 = xt_i64(s.c1, i);
        o.val[0] = v3;
        const int64_t v4 = xt_i64(s.c2, i);
        o.val[1] = v4;
        o.ent = NO_ROW;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_build_values(XArgs a, XStage<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_vstage8<P>(a, s, nrows, seg_rows, nseg);
}
