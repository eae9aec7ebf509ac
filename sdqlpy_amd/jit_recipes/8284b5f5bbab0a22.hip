#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 0, NV = 0, NSC = 1, NSOP = 2, ND = 0;
    __device__ __forceinline__ static constexpr int scol(int j) { return j == 0 ? 0 : 0; }
    __device__ __forceinline__ static constexpr int swidth(int j) { return j == 0 ? 101 : 101; }
    __device__ __forceinline__ static constexpr bool sbytes(int j) { return j == 0 ? false : false; }
    template <int J> __device__ __forceinline__ static void sops(const XArgs& a, const uint32_t* region, int off, int64_t (&sres)[2]) {
        if constexpr (J == 0) {
            const uint32_t* f = region + (off >> 2);
            sres[0] = (int64_t)lds_first_index(f, 101, a.spool + 0, 8);
            sres[1] = (int64_t)lds_first_index(f, 101, a.spool + 8, 10);
        }
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Pair<int64_t> (&s)[1]) {
    }
    __device__ __forceinline__ static void stest(const XArgs& a, const Pair<int64_t> (&s)[1], bool& p0, bool& p1) {
      {
      }
      {
      }
    }
    template <int H> __device__ __forceinline__ static bool eval_regs(const XArgs& a, const Pair<int64_t> (&s)[1], int64_t r, XOut<NV>& o) {
        return false;
    }
    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[2], XOut<NV>& o) {
        const int64_t v0 = sres[0];
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 != v1);
        if (!v2) return false;
        const int64_t v3 = sres[1];
        const int64_t v4 = a.ci[1];
        const int64_t v5 = (v0 + v4);
        const bool v6 = (v3 > v5);
        if (!v6) return false;
        int64_t v7 = static_cast<const int64_t*>(a.col[1])[r];
        x_pin(v7);
        o.key = v7; o.bad = false;
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_keyset_queue(XArgs a, XKeySet<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue<P, XKeySet, false>(a, s, nrows, seg_rows, nseg);
}
