#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 0, NV = 3, NSC = 1, NSOP = 1, ND = 0;
    __device__ __forceinline__ static constexpr int scol(int j) { return j == 0 ? 1 : 1; }
    __device__ __forceinline__ static constexpr int swidth(int j) { return j == 0 ? 10 : 10; }
    __device__ __forceinline__ static constexpr bool sbytes(int j) { return j == 0 ? false : false; }
    template <int J> __device__ __forceinline__ static void sops(const XArgs& a, const uint32_t* region, int off, int64_t (&sres)[1]) {
        if constexpr (J == 0) {
            const uint32_t* f = region + (off >> 2);
            sres[0] = (int64_t)lds_str_pred(f, 10, a.spool + 0, 8, 1);
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
    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[1], XOut<NV>& o) {
        int64_t v2 = static_cast<const int64_t*>(a.col[2])[r];
        int64_t v9 = static_cast<const int64_t*>(a.col[3])[r];
        x_pin(v2, v9);
        const bool v1 = (sres[0] != 0);
        if (!v1) return false;
        const int64_t v3 = a.ci[0];
        const bool v5 = (v2 >= v3);
        const int64_t v4 = a.ci[1];
        const bool v6 = (v2 <= v4);
        const bool v7 = (v5 && v6);
        const bool v8 = (!v7);
        if (!v8) return false;
        const int64_t v10 = a.ci[2];
        const bool v11 = (v9 == v10);
        const int64_t v12 = a.ci[3];
        const bool v13 = (v9 == v12);
        const bool v26 = (v11 || v13);
        const int64_t v14 = a.ci[4];
        const bool v15 = (v9 == v14);
        const bool v27 = (v26 || v15);
        const int64_t v16 = a.ci[5];
        const bool v17 = (v9 == v16);
        const bool v28 = (v27 || v17);
        const int64_t v18 = a.ci[6];
        const bool v19 = (v9 == v18);
        const bool v29 = (v28 || v19);
        const int64_t v20 = a.ci[7];
        const bool v21 = (v9 == v20);
        const bool v30 = (v29 || v21);
        const int64_t v22 = a.ci[8];
        const bool v23 = (v9 == v22);
        const bool v31 = (v30 || v23);
        const int64_t v24 = a.ci[9];
        const bool v25 = (v9 == v24);
        const bool v32 = (v31 || v25);
        if (!v32) return false;
        int64_t v0 = static_cast<const int64_t*>(a.col[0])[r];
        int64_t v33 = static_cast<const int64_t*>(a.col[4])[r];
        x_pin(v0, v33);
        o.key = v33; o.bad = false;
        o.val[0] = v0;
        o.val[1] = v2;
        o.val[2] = v9;
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_build_queue(XArgs a, XStage<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue<P, XStage, true>(a, s, nrows, seg_rows, nseg);
}
