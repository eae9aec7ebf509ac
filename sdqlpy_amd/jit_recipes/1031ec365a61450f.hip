#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 1, NV = 1, NSC = 1, NSOP = 2, ND = 0;
    struct Regs { uint32_t c0[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
    }
    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {
        bool p = true;
        const double v0 = x_f(xt_i64(s.c0, i));
        const double v1 = a.cf[0];
        const bool v2 = (v0 > v1);
        p = p & v2;
        return p;
    }
    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {
        bool p = true;
        const double v0 = x_f(xt_i64(s.c0, i));
        const double v1 = a.cf[0];
        const bool v2 = (v0 > v1);
        p = p & v2;
        widx = 0; bit = 0;
        return p;
    }
    static constexpr bool PREF32 = false, PWIN = false;
    __device__ __forceinline__ static bool spre32(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& off) {
        off = 0; return false;
    }
    __device__ __forceinline__ static const uint32_t* sbitmap(const XArgs& a) { return nullptr; }
    __device__ __forceinline__ static constexpr int scol(int j) { return j == 0 ? 3 : 3; }
    __device__ __forceinline__ static constexpr int swidth(int j) { return j == 0 ? 15 : 15; }
    __device__ __forceinline__ static constexpr bool sbytes(int j) { return j == 0 ? false : false; }
    template <int J> __device__ __forceinline__ static void sops(const XArgs& a, const uint32_t* region, int off, int64_t (&sres)[2]) {
        if constexpr (J == 0) {
            const uint32_t* f = region + (off >> 2);
            sres[0] = (int64_t)x_char(f, 15, 0);
            sres[1] = (int64_t)x_char(f, 15, 1);
        }
    }
    template <int H> __device__ __forceinline__ static bool eval_regs(const XArgs& a, const Pair<int64_t> (&s)[1], int64_t r, XOut<NV>& o) {
        return false;
    }
    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[2], XOut<NV>& o) {
        int64_t v3 = static_cast<const int64_t*>(a.col[1])[r];
        int64_t v6 = static_cast<const int64_t*>(a.col[2])[r];
        x_pin(v3, v6);
        const uint32_t e4 = x_lookup_l<0x8000000au>(a.tab[0], v3, false);
        const bool v4 = (e4 != NO_ROW);
        const bool v5 = (!v4);
        if (!v5) return false;
        const int64_t v7 = a.ci[0];
        const bool v9 = (v6 >= v7);
        const int64_t v8 = a.ci[1];
        const bool v10 = (v6 <= v8);
        const bool v11 = (v9 && v10);
        const int64_t v12 = a.ci[2];
        const bool v14 = (v6 >= v12);
        const int64_t v13 = a.ci[3];
        const bool v15 = (v6 <= v13);
        const bool v16 = (v14 && v15);
        const bool v42 = (v11 || v16);
        const int64_t v17 = a.ci[4];
        const bool v19 = (v6 >= v17);
        const int64_t v18 = a.ci[5];
        const bool v20 = (v6 <= v18);
        const bool v21 = (v19 && v20);
        const bool v43 = (v42 || v21);
        const int64_t v22 = a.ci[6];
        const bool v24 = (v6 >= v22);
        const int64_t v23 = a.ci[7];
        const bool v25 = (v6 <= v23);
        const bool v26 = (v24 && v25);
        const bool v44 = (v43 || v26);
        const int64_t v27 = a.ci[8];
        const bool v29 = (v6 >= v27);
        const int64_t v28 = a.ci[9];
        const bool v30 = (v6 <= v28);
        const bool v31 = (v29 && v30);
        const bool v45 = (v44 || v31);
        const int64_t v32 = a.ci[10];
        const bool v34 = (v6 >= v32);
        const int64_t v33 = a.ci[11];
        const bool v35 = (v6 <= v33);
        const bool v36 = (v34 && v35);
        const bool v46 = (v45 || v36);
        const int64_t v37 = a.ci[12];
        const bool v39 = (v6 >= v37);
        const int64_t v38 = a.ci[13];
        const bool v40 = (v6 <= v38);
        const bool v41 = (v39 && v40);
        const bool v47 = (v46 || v41);
        if (!v47) return false;
        double v0 = static_cast<const double*>(a.col[0])[r];
        x_pin(v0);
        const int64_t v48 = sres[0];
        const int64_t v50 = a.ci[14];
        const int64_t v51 = (v48 * v50);
        const int64_t v49 = sres[1];
        const int64_t v52 = (v51 + v49);
        o.key = v52; o.bad = false;
        o.val[0] = x_bits(v0);
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_group_tight(XArgs a, XGroup<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue8<P, XGroup, false>(a, s, nrows, seg_rows, nseg);
}
