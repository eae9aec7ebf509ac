#include "sdqh_xkernels.hpp"
using namespace sdqh;
struct P {
    static constexpr int NS = 3, NV = 1, NSC = 0, NSOP = 0, ND = 0;
    struct Regs { uint32_t c0[16]; uint32_t c1[16]; uint32_t c2[16]; };
    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {
    }
    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {
        xt_load<8, TAIL>(a.col[0], r, nrows, s.c0);
        xt_load<8, TAIL>(a.col[1], r, nrows, s.c1);
        xt_load<8, TAIL>(a.col[2], r, nrows, s.c2);
    }
    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {
        bool p = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 == v1);
        const int64_t v3 = xt_i64(s.c1, i);
        const bool v4 = (v3 >= v1);
        const bool v19 = (v2 && v4);
        const int64_t v5 = a.ci[1];
        const bool v6 = (v3 <= v5);
        const bool v20 = (v19 && v6);
        const int64_t v7 = xt_i64(s.c2, i);
        const int64_t v8 = a.ci[2];
        const bool v9 = (v7 == v8);
        const int64_t v10 = a.ci[3];
        const bool v11 = (v7 == v10);
        const bool v16 = (v9 || v11);
        const int64_t v12 = a.ci[4];
        const bool v13 = (v7 == v12);
        const bool v17 = (v16 || v13);
        const int64_t v14 = a.ci[5];
        const bool v15 = (v7 == v14);
        const bool v18 = (v17 || v15);
        const bool v21 = (v20 && v18);
        const int64_t v22 = a.ci[6];
        const bool v23 = (v0 == v22);
        const bool v37 = (v23 && v4);
        const int64_t v24 = a.ci[7];
        const bool v25 = (v3 <= v24);
        const bool v38 = (v37 && v25);
        const int64_t v26 = a.ci[8];
        const bool v27 = (v7 == v26);
        const int64_t v28 = a.ci[9];
        const bool v29 = (v7 == v28);
        const bool v34 = (v27 || v29);
        const int64_t v30 = a.ci[10];
        const bool v31 = (v7 == v30);
        const bool v35 = (v34 || v31);
        const int64_t v32 = a.ci[11];
        const bool v33 = (v7 == v32);
        const bool v36 = (v35 || v33);
        const bool v39 = (v38 && v36);
        const bool v57 = (v21 || v39);
        const int64_t v40 = a.ci[12];
        const bool v41 = (v0 == v40);
        const bool v54 = (v41 && v4);
        const int64_t v42 = a.ci[13];
        const bool v43 = (v3 <= v42);
        const bool v55 = (v54 && v43);
        const int64_t v44 = a.ci[14];
        const bool v45 = (v7 == v44);
        const int64_t v46 = a.ci[15];
        const bool v47 = (v7 == v46);
        const bool v51 = (v45 || v47);
        const int64_t v48 = a.ci[16];
        const bool v49 = (v7 == v48);
        const bool v52 = (v51 || v49);
        const bool v50 = (v7 == v42);
        const bool v53 = (v52 || v50);
        const bool v56 = (v55 && v53);
        const bool v58 = (v57 || v56);
        p = p & v58;
        return p;
    }
    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {
        bool p = true;
        const int64_t v0 = xt_i64(s.c0, i);
        const int64_t v1 = a.ci[0];
        const bool v2 = (v0 == v1);
        const int64_t v3 = xt_i64(s.c1, i);
        const bool v4 = (v3 >= v1);
        const bool v19 = (v2 && v4);
        const int64_t v5 = a.ci[1];
        const bool v6 = (v3 <= v5);
        const bool v20 = (v19 && v6);
        const int64_t v7 = xt_i64(s.c2, i);
        const int64_t v8 = a.ci[2];
        const bool v9 = (v7 == v8);
        const int64_t v10 = a.ci[3];
        const bool v11 = (v7 == v10);
        const bool v16 = (v9 || v11);
        const int64_t v12 = a.ci[4];
        const bool v13 = (v7 == v12);
        const bool v17 = (v16 || v13);
        const int64_t v14 = a.ci[5];
        const bool v15 = (v7 == v14);
        const bool v18 = (v17 || v15);
        const bool v21 = (v20 && v18);
        const int64_t v22 = a.ci[6];
        const bool v23 = (v0 == v22);
        const bool v37 = (v23 && v4);
        const int64_t v24 = a.ci[7];
        const bool v25 = (v3 <= v24);
        const bool v38 = (v37 && v25);
        const int64_t v26 = a.ci[8];
        const bool v27 = (v7 == v26);
        const int64_t v28 = a.ci[9];
        const bool v29 = (v7 == v28);
        const bool v34 = (v27 || v29);
        const int64_t v30 = a.ci[10];
        const bool v31 = (v7 == v30);
        const bool v35 = (v34 || v31);
        const int64_t v32 = a.ci[11];
        const bool v33 = (v7 == v32);
        const bool v36 = (v35 || v33);
        const bool v39 = (v38 && v36);
        const bool v57 = (v21 || v39);
        const int64_t v40 = a.ci[12];
        const bool v41 = (v0 == v40);
        const bool v54 = (v41 && v4);
        const int64_t v42 = a.ci[13];
        const bool v43 = (v3 <= v42);
        const bool v55 = (v54 && v43);
        const int64_t v44 = a.ci[14];
        const bool v45 = (v7 == v44);
        const int64_t v46 = a.ci[15];
        const bool v47 = (v7 == v46);
        const bool v51 = (v45 || v47);
        const int64_t v48 = a.ci[16];
        const bool v49 = (v7 == v48);
        const bool v52 = (v51 || v49);
        const bool v50 = (v7 == v42);
        const bool v53 = (v52 || v50);
        const bool v56 = (v55 && v53);
        const bool v58 = (v57 || v56);
        p = p & v58;
        widx = 0; bit = 0;
        return p;
    }
    static constexpr bool PREF32 = false, PWIN = false;
    __device__ __forceinline__ static bool spre32(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& off) {
        off = 0; return false;
    }
    __device__ __forceinline__ static const uint32_t* sbitmap(const XArgs& a) { return nullptr; }
    template <int H> __device__ __forceinline__ static bool eval_regs(const XArgs& a, const Pair<int64_t> (&s)[3], int64_t r, XOut<NV>& o) {
        return false;
    }
    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[1], XOut<NV>& o) {
        int64_t v0 = static_cast<const int64_t*>(a.col[0])[r];
        int64_t v59 = static_cast<const int64_t*>(a.col[3])[r];
        x_pin(v0, v59);
        o.key = v59; o.bad = false;
        o.val[0] = v0;
        o.ent = NO_ROW;
        return true;
    }
};
extern "C" __global__ __launch_bounds__(256) void xk_build_tight(XArgs a, XStage<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {
    x_queue8<P, XStage, true>(a, s, nrows, seg_rows, nseg);
}
