#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 0, NV = 0, NSC = 0, NSOP = 0, ND = 0;
    struct Regs { };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
    }
    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {
        bool p = true;
        return p;
    }
    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {
        bool p = true;
        widx = 0; bit = 0;
        return p;
    }
    static constexpr bool PREF32 = false, PWIN = false;
    __device__ __forceinline__ static bool spre32(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& off) {
        off = 0; return false;
    }
    __device__ __forceinline__ static const uint32_t* sbitmap(const XArgs& a) { return nullptr; }
    template <int H> __device__ __forceinline__ static bool eval_regs(const XArgs& a, const Pair<int64_t> (&s)[1], int64_t r, XOut<NV>& o) {
        return false;
    }
    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[1], XOut<NV>& o) {
        int64_t v0 = static_cast<const int64_t*>(a.col[0])[r];
        x_pin(v0);
        o.key = v0; o.bad = false;
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_build_tight(XArgs a, XStage<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue8<P, XStage, true>(a, s, nrows, seg_rows, nseg);
}
