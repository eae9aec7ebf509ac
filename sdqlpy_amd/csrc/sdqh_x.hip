// sdqh_x.hip — row programs (ABI 4, include/sdqh.h): the open expression / predicate vocabulary.
//
// The reference compiles every query file into C++ of its own (src/sdqlpy/lib/sdql_ir_cpp_generator_par.py
// prints the loop bodies; sdql_lib.py:372-387 builds and caches the module).  The MI355X counterpart of that
// step is here: a program — the conditions, lookups and arithmetic of ONE loop, as data — is turned into a
// small struct of device functions, and hiprtc specialises a hand-written kernel skeleton
// (sdqh_xkernels.hpp: streaming, LDS survivor queue, converged drain, sinks) on it for gfx950.  Code
// objects are cached in memory and on disk by a hash of the generated source, so a plan pays for its
// specialisation once (about half a second), like the reference's mode 2 reuses its compiled module.
// There is no interpreter and no CPU path in this library: if hiprtc or its headers are missing the
// sdqh_x* calls fail with SDQH_ERR_DEVICE and the message says why.
#define SDQH_DECLS_ONLY 1            // argument structs and constants of the kernel headers, not the kernels
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>
#include <string>
#include <vector>

#include "sdqh_host.hpp"
#include "sdqh_xkernels.hpp"

using namespace sdqh_host;

namespace {

#define HIP_TRYX(ctx, expr)                                                                             \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

enum Sink { SINK_SUM, SINK_GROUP, SINK_STAGE, SINK_KEYSET, SINK_ENTRY, SINK_GROUP_LANE };
const char* sink_name(Sink s) { return s == SINK_SUM ? "XSum" : s == SINK_GROUP ? "XGroup" : s == SINK_STAGE ? "XStage" : s == SINK_KEYSET ? "XKeySet" : s == SINK_ENTRY ? "XEntry" : "XGroupLane"; }
// the kernel's symbol: what rocprofv3 lists it under (one name per skeleton and sink, not one per program)
std::string entry_name(Sink s, bool direct, bool tight) {
    const char* sk = s == SINK_SUM ? "sum" : s == SINK_GROUP ? "group" : s == SINK_STAGE ? "build" : s == SINK_KEYSET ? "keyset" : s == SINK_ENTRY ? "probe_agg" : "group_lane";
    return std::string("xk_") + sk + (tight ? "_tight" : direct ? "_direct" : "_queue");
}
const char* VSTAGE_ENTRY = "xk_build_values";
enum Enc { ENC_D8 = -1, ENC_RAW = 0, ENC_N32 = 1, ENC_C16 = 2, ENC_C8 = 3 };      // (D8: the delta twin, queue programs only; ordered so that `>= ENC_C16` means "coded")
// the same name as the profiling label of a launch (a pointer that stays valid): bench.py's per-kernel table and rocprofv3's agree on names
const char* launch_label(Sink s, bool direct, bool tight) {
    static std::mutex mu; static std::vector<std::string*> names;
    const std::string n = entry_name(s, direct, tight);
    std::lock_guard<std::mutex> lock(mu);
    for (auto* have : names) if (*have == n) return have->c_str();
    names.push_back(new std::string(n));
    return names.back()->c_str();
}

// ---- program analysis -------------------------------------------------------------------------------
struct XInfo {
    const sdqh_program* p = nullptr;
    int ncols = 0; const sdqh_column* cols[SDQH_MAX_XCOLS]; int col_of[SDQH_MAX_XOPS];
    int ntabs = 0; sdqh_table* tabs[SDQH_MAX_XTABLES]; int tab_of[SDQH_MAX_XOPS];
    int nci = 0, ncf = 0; int const_of[SDQH_MAX_XOPS]; int64_t ci[X_MAX_CONST]; double cf[X_MAX_CONST];
    int nstr = 0; int str_off[SDQH_MAX_XOPS]; uint32_t spool[SDQH_MAX_XSTR];
    bool direct = false;                     // every operation is register arithmetic on numeric columns
    int nstream_gates = 0;                   // leading gates that depend on numeric columns / constants only
    std::vector<int> scols;                  // streamed columns (indices into cols)
    uint32_t narrow_mask = 0;                // streamed columns read through their exact 4-byte twin (bit = index into cols)
    int probe_op = -1;                       // XEntry: the LOOKUP whose entry receives the values
    int prefilter_op = -1;                   // first LOOKUP gate after the streamed ones whose key is made of plain columns:
    int prefilter_part0 = -1;                //   its table's key bitmap is tested on the streamed key (operation of the first key part)
    bool prefilter_composite = false;
    // TIGHT (x_tight): a register program whose columns are streamed at their tightest exact encoding, 8 rows per lane
    bool tight = false;
    int enc[SDQH_MAX_XCOLS] = {};            // Enc per column
    int dict_slot[SDQH_MAX_XCOLS];           // LDS dictionary table of a coded column whose VALUE the program uses, or -1
    int nd = 0;
    int cmp_cc[SDQH_MAX_XOPS];               // comparison of a coded column with a constant, rewritten in code space: its slot in cc[], or -1
    int cmp_kind[SDQH_MAX_XOPS];             // 0: code < t   1: code >= t   2: code == t   3: code != t
    int cmp_col[SDQH_MAX_XOPS];              // the coded column
    uint32_t cc[X_MAX_CONST]; int ncc = 0;
    bool affine[SDQH_MAX_XCOLS] = {};        // a coded integer column whose distinct values are consecutive integers: value = code + dlo, no table
    int64_t dlo[SDQH_MAX_XCOLS] = {};
    signed char irange[SDQH_MAX_XOPS] = {};  // i64 operations over the columns' actual ranges: 0 unknown / wide, 1 fits int32, 2 fits 24 bits (32-bit arithmetic, v_mul_i32_i24)
    bool fake = false;
    bool vstage = false; int nlk = 0; int lk_op[2] = {-1, -1};      // a build that x_vstage8 can run: every gate on registers, lookups answered by exact 32-bit-range bitmaps
    bool pref32 = false;                     // the prefilter's key and its table's bitmap range fit 32 bits: the streamed test is 32-bit arithmetic
    int ival = -1;                           // the per-lane group sink: summed value that is a small integer on every row (byte-coded column, consecutive integral dictionary), or -1
    bool pnear = false;                      // ... and a lane's 8 consecutive rows carry near-by keys (column_span8): a row that fails an earlier condition still asks for ITS key's word
    bool pwin = false;                       // ... tested against one 128-bit window of the bitmap: one 16-byte request per lane and 8 rows (option "window", off: measured slower)
    bool phash = false;                         // the prefilter's table is a hash layout with a hashed filter (DevTable::hf): its key is tested against that on streamed registers (x_queue8: P::PHASH)
    bool want_driven = false;                   // set by the caller whose sink can take the walk (the group sink): no other loop pays for a run index
    bool driven = false; int driven_col = -1;   // the driven walk of x_queue8 can be taken (order-free sinks): the prefilter's key column is stored in its own order and has a run index
    uint32_t gather32 = 0;                   // queue programs: numeric columns read BY ROW (the drain's gathers) through their 4-byte twins: half the bytes of every touched line
    mutable uint32_t lay[SDQH_MAX_XTABLES] = {};   // XL_* layout bits of every table (kernel_for: known once the tables' indexes are made), 0: decided at run time
    std::vector<char> scope;                 // operations evaluated on the streamed registers (register programs: all; queue programs: the streamed gates + the prefilter's key)
};

bool op_is_light(const sdqh_xop& o) {        // evaluable on streamed registers
    switch (o.code) {
        case SDQH_X_LOOKUP: case SDQH_X_FIELD: case SDQH_X_ACC: case SDQH_X_STR: case SDQH_X_STRIDX: case SDQH_X_CHAR: return false;
        default: return true;
    }
}
void closure(const sdqh_program* p, int k, std::vector<char>& seen) {       // the operations value k depends on (itself included)
    if (k < 0 || k >= p->nops || seen[(size_t)k]) return;
    seen[(size_t)k] = 1;
    const sdqh_xop& o = p->ops[k];
    if (o.code == SDQH_X_COL || o.code == SDQH_X_ROWID || o.code == SDQH_X_CONST || o.code == SDQH_X_STR || o.code == SDQH_X_STRIDX || o.code == SDQH_X_CHAR) return;
    if (o.a < k) closure(p, o.a, seen);
    if (o.b < k) closure(p, o.b, seen);
    if (o.c < k) closure(p, o.c, seen);
}

void tight_plan(sdqh_ctx* ctx, int64_t nrows, XInfo* x);

// the ABI's rules for a program (the CPU implementation applies the same ones)
int analyse(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* p, int max_vals, bool need_key, bool vals_f64, XInfo* x) {
    if (!p || p->nops < 0 || p->nops > SDQH_MAX_XOPS || (p->nops && !p->ops) || p->ngates < 0 || p->ngates > SDQH_MAX_XGATES || (p->ngates && !p->gates) ||
        p->nvals < 0 || p->nvals > max_vals || (p->nvals && !p->vals))
        return fail(ctx, SDQH_ERR_INVALID, "program: bad counts");
    x->p = p;
    for (int k = 0; k < p->nops; ++k) {
        const sdqh_xop& o = p->ops[k];
        x->col_of[k] = x->tab_of[k] = x->const_of[k] = x->str_off[k] = -1;
        auto ty = [&](int j) { return (j >= 0 && j < k) ? p->ops[j].type : -1; };
        auto need = [&](bool ok, const char* what) { return ok ? SDQH_OK : fail(ctx, SDQH_ERR_INVALID, std::string("program: operation ") + std::to_string(k) + ": " + what); };
        int rc = SDQH_OK;
        switch (o.code) {
            case SDQH_X_COL:
                rc = need(o.col && o.col->dtype != SDQH_STR && o.col->nrows >= nrows && o.type == (o.col->dtype == SDQH_F64 ? SDQH_T_F64 : SDQH_T_I64), "COL needs an I64 / F64 column covering nrows, typed alike"); break;
            case SDQH_X_ROWID: rc = need(o.type == SDQH_T_I64, "ROWID is i64"); break;
            case SDQH_X_CONST: rc = need(o.type >= SDQH_T_I64 && o.type <= SDQH_T_BOOL, "CONST type"); break;
            case SDQH_X_LOOKUP: rc = need(o.table && ty(o.a) == SDQH_T_I64 && o.type == SDQH_T_BOOL, "LOOKUP needs a table and an i64 key"); break;
            case SDQH_X_FIELD:
                rc = need(o.a >= 0 && o.a < k && p->ops[o.a].code == SDQH_X_LOOKUP && !p->ops[o.a].table->bitmap_only && o.aux >= 0 && o.aux < p->ops[o.a].table->npay &&
                          (o.type == SDQH_T_I64 || o.type == SDQH_T_F64), "FIELD needs an earlier LOOKUP and one of its payload fields"); break;
            case SDQH_X_ACC:
                rc = need(o.a >= 0 && o.a < k && p->ops[o.a].code == SDQH_X_LOOKUP && p->ops[o.a].table->accumulate && !p->ops[o.a].table->bitmap_only && o.aux >= -1 && o.aux < SDQH_TUPLE_MAX_VALUES &&
                          o.type == (o.aux < 0 ? SDQH_T_I64 : SDQH_T_F64), "ACC needs an earlier LOOKUP into a table with accumulators"); break;
            case SDQH_X_ADD: case SDQH_X_SUB: case SDQH_X_MUL:
                rc = need(ty(o.a) == ty(o.b) && (ty(o.a) == SDQH_T_I64 || ty(o.a) == SDQH_T_F64) && o.type == ty(o.a), "arithmetic needs two operands of one numeric type"); break;
            case SDQH_X_DIV: rc = need(ty(o.a) == SDQH_T_F64 && ty(o.b) == SDQH_T_F64 && o.type == SDQH_T_F64, "DIV is f64"); break;
            case SDQH_X_NEG: rc = need((ty(o.a) == SDQH_T_I64 || ty(o.a) == SDQH_T_F64) && o.type == ty(o.a), "NEG operand"); break;
            case SDQH_X_I2F: rc = need(ty(o.a) == SDQH_T_I64 && o.type == SDQH_T_F64, "I2F operand"); break;
            case SDQH_X_YEAR: rc = need(ty(o.a) == SDQH_T_I64 && o.type == SDQH_T_I64, "YEAR operand"); break;
            case SDQH_X_DIVI: case SDQH_X_MODI: rc = need(ty(o.a) == SDQH_T_I64 && o.type == SDQH_T_I64 && o.imm_i > 0, "DIVI / MODI need an i64 operand and a positive divisor"); break;
            case SDQH_X_PACK2: rc = need(ty(o.a) == SDQH_T_I64 && ty(o.b) == SDQH_T_I64 && o.type == SDQH_T_I64, "PACK2 operands"); break;
            case SDQH_X_LT: case SDQH_X_LE: case SDQH_X_GT: case SDQH_X_GE: case SDQH_X_EQ: case SDQH_X_NE:
                rc = need(ty(o.a) == ty(o.b) && (ty(o.a) == SDQH_T_I64 || ty(o.a) == SDQH_T_F64 || (ty(o.a) == SDQH_T_BOOL && (o.code == SDQH_X_EQ || o.code == SDQH_X_NE))) && o.type == SDQH_T_BOOL,
                          "comparison needs two operands of one type"); break;
            case SDQH_X_AND: case SDQH_X_OR: rc = need(ty(o.a) == SDQH_T_BOOL && ty(o.b) == SDQH_T_BOOL && o.type == SDQH_T_BOOL, "boolean operands"); break;
            case SDQH_X_NOT: rc = need(ty(o.a) == SDQH_T_BOOL && o.type == SDQH_T_BOOL, "boolean operand"); break;
            case SDQH_X_SELECT: rc = need(ty(o.a) == SDQH_T_BOOL && ty(o.b) == ty(o.c) && ty(o.b) >= 0 && o.type == ty(o.b), "SELECT needs a bool and two values of one type"); break;
            case SDQH_X_STR: case SDQH_X_STRIDX:
                rc = need(o.col && o.col->dtype == SDQH_STR && o.col->nrows >= nrows && o.slen >= 0 && o.slen <= SDQH_MAX_STR_CONST && (o.slen == 0 || o.str) &&
                          o.type == (o.code == SDQH_X_STR ? SDQH_T_BOOL : SDQH_T_I64) && (o.code != SDQH_X_STR || (o.aux >= SDQH_STR_EQ && o.aux <= SDQH_STR_SUFFIX)), "string operation needs a STR column and a constant");
                if (!rc) {
                    if (x->nstr + o.slen > SDQH_MAX_XSTR) return fail(ctx, SDQH_ERR_UNSUPPORTED, "program: string constants exceed SDQH_MAX_XSTR code units");
                    x->str_off[k] = x->nstr;
                    for (int i = 0; i < o.slen; ++i) x->spool[x->nstr++] = o.str[i];
                }
                break;
            case SDQH_X_CHAR: rc = need(o.col && o.col->dtype == SDQH_STR && o.col->nrows >= nrows && o.type == SDQH_T_I64 && o.aux >= 0, "CHAR needs a STR column"); break;
            default: rc = fail(ctx, SDQH_ERR_UNSUPPORTED, "program: unknown operation code " + std::to_string(o.code));
        }
        if (rc) return rc;
        if (o.col) {
            int j = 0; while (j < x->ncols && x->cols[j] != o.col) ++j;
            if (j == x->ncols) { if (x->ncols == SDQH_MAX_XCOLS) return fail(ctx, SDQH_ERR_UNSUPPORTED, "program: more than SDQH_MAX_XCOLS columns"); x->cols[x->ncols++] = o.col; }
            x->col_of[k] = j;
        }
        if (o.code == SDQH_X_LOOKUP) {
            sdqh_table* t = const_cast<sdqh_table*>(o.table);
            int j = 0; while (j < x->ntabs && x->tabs[j] != t) ++j;
            if (j == x->ntabs) { if (x->ntabs == SDQH_MAX_XTABLES) return fail(ctx, SDQH_ERR_UNSUPPORTED, "program: more than SDQH_MAX_XTABLES tables"); x->tabs[x->ntabs++] = t; }
            x->tab_of[k] = j;
        }
        if (o.code == SDQH_X_CONST) {
            if (o.type == SDQH_T_F64) { if (x->ncf == X_MAX_CONST) return fail(ctx, SDQH_ERR_UNSUPPORTED, "program: too many constants"); x->cf[x->ncf] = o.imm_f; x->const_of[k] = x->ncf++; }
            else { if (x->nci == X_MAX_CONST) return fail(ctx, SDQH_ERR_UNSUPPORTED, "program: too many constants"); x->ci[x->nci] = o.imm_i; x->const_of[k] = x->nci++; }
        }
    }
    for (int g = 0; g < p->ngates; ++g)
        if (p->gates[g] < 0 || p->gates[g] >= p->nops || p->ops[p->gates[g]].type != SDQH_T_BOOL) return fail(ctx, SDQH_ERR_INVALID, "program: a gate must be a bool operation");
    if (need_key ? !(p->key >= 0 && p->key < p->nops && p->ops[p->key].type == SDQH_T_I64) : p->key != -1) return fail(ctx, SDQH_ERR_INVALID, "program: key");
    for (int v = 0; v < p->nvals; ++v) {
        if (p->vals[v] < 0 || p->vals[v] >= p->nops) return fail(ctx, SDQH_ERR_INVALID, "program: value index");
        const int t = p->ops[p->vals[v]].type;
        if (vals_f64 ? t != SDQH_T_F64 : (t != SDQH_T_I64 && t != SDQH_T_F64)) return fail(ctx, SDQH_ERR_INVALID, "program: value type");
    }
    // DIRECT: nothing but register arithmetic on numeric columns
    x->direct = x->ncols <= 12;
    for (int k = 0; k < p->nops; ++k) x->direct = x->direct && op_is_light(p->ops[k]);
    // leading gates that only need numeric columns: tested on the streamed registers
    std::vector<char> scol((size_t)SDQH_MAX_XCOLS, 0);
    for (int g = 0; g < p->ngates; ++g) {
        std::vector<char> seen((size_t)p->nops, 0);
        closure(p, p->gates[g], seen);
        bool light = true; std::vector<int> cols;
        for (int k = 0; k < p->nops; ++k) if (seen[(size_t)k]) { light = light && op_is_light(p->ops[k]); if (p->ops[k].code == SDQH_X_COL) cols.push_back(x->col_of[k]); }
        int extra = 0; for (int c : cols) if (!scol[(size_t)c]) ++extra;
        int have = 0; for (char c : scol) have += c;
        if (!light || have + extra > 6) break;
        for (int c : cols) scol[(size_t)c] = 1;
        x->nstream_gates = g + 1;
    }
    // the next gate, when it is a lookup keyed by plain int columns: stream its key and test the table's key
    // bitmap (exact for key sets / the direct layout, the high part's for composite keys) before queueing
    if (x->nstream_gates < p->ngates) {
        const sdqh_xop& g = p->ops[p->gates[x->nstream_gates]];
        if (g.code == SDQH_X_LOOKUP) {
            const sdqh_xop& k = p->ops[g.a];
            int part0 = -1, part1 = -1;
            if (k.code == SDQH_X_COL && k.type == SDQH_T_I64) part0 = g.a;
            else if (k.code == SDQH_X_PACK2 && p->ops[k.a].code == SDQH_X_COL && p->ops[k.b].code == SDQH_X_COL) { part0 = k.a; part1 = k.b; }
            int have = 0; for (char c : scol) have += c;
            if (part0 >= 0 && have + (scol[(size_t)x->col_of[part0]] ? 0 : 1) <= 7) {
                scol[(size_t)x->col_of[part0]] = 1;
                x->prefilter_op = p->gates[x->nstream_gates]; x->prefilter_part0 = part0; x->prefilter_composite = part1 >= 0;
            }
        }
    }
    for (int c = 0; c < x->ncols; ++c) if (scol[(size_t)c]) x->scols.push_back(c);
    // narrow twins of what is streamed (register programs stream every column): decided per column, part of the kernel's structure
    if (ctx->opt_narrow && !ctx->compile_only && nrows >= ctx->opt_feature_min_rows) {
        for (int c = 0; c < x->ncols; ++c) {
            if (!(x->direct || scol[(size_t)c]) || x->cols[c]->dtype == SDQH_STR) continue;
            if (column_narrow(ctx, const_cast<sdqh_column*>(x->cols[c]))) x->narrow_mask |= 1u << c;
        }
        // text columns a queue program scans (staged in LDS by the drain): through their byte twin
        if (!x->direct) for (int k = 0; k < x->p->nops; ++k) {
            const int code = x->p->ops[k].code;
            if (code != SDQH_X_STR && code != SDQH_X_STRIDX && code != SDQH_X_CHAR) continue;
            const int c = x->col_of[k];
            if (c < 0 || x->cols[c]->dtype != SDQH_STR || x->cols[c]->width > 1024 || ((x->narrow_mask >> c) & 1u)) continue;
            if (column_narrow(ctx, const_cast<sdqh_column*>(x->cols[c]))) x->narrow_mask |= 1u << c;
        }
    }
    tight_plan(ctx, nrows, x);
    return SDQH_OK;
}


// ---- TIGHT plans ------------------------------------------------------------------------------------------
bool is_cmp(int code) { return code >= SDQH_X_LT && code <= SDQH_X_NE; }

// [lo, hi] of an i64 operation over the columns' actual value ranges (min / max are cached per column; a coded column's
// range is its dictionary's ends).  false: unknown (a product that leaves 62 bits, an operation without a range).
bool op_interval(sdqh_ctx* ctx, const XInfo& x, int k, int64_t* lo, int64_t* hi) {
    const sdqh_xop& o = x.p->ops[k];
    if (o.type != SDQH_T_I64) return false;
    const __int128 LIM = (__int128)1 << 62;
    auto fits = [&](__int128 a, __int128 b) { return a > -LIM && a < LIM && b > -LIM && b < LIM; };
    int64_t al, ah, bl, bh;
    switch (o.code) {
        case SDQH_X_CONST: *lo = *hi = o.imm_i; return true;
        case SDQH_X_COL: {
            sdqh_column* c = const_cast<sdqh_column*>(o.col);
            if (x.fake) { if (c->dtype != SDQH_I64) return false; *lo = 0; *hi = 3; return true; }
            // (a transient column — rows that live for one run — gets no minimum / maximum pass; bounds it was TOLD, sdqh_column_set_bounds, count)
            if (c->dtype != SDQH_I64 || c->nrows < 1 || (c->transient && !c->have_minmax)) return false;
            if (c->code_state == 1 && !c->dict_host.empty()) { *lo = c->dict_host.front(); *hi = c->dict_host.back(); return true; }
            if (column_minmax(ctx, c)) return false;
            *lo = c->mn; *hi = c->mx; return true;
        }
        case SDQH_X_ADD: case SDQH_X_SUB: case SDQH_X_MUL: {
            if (!op_interval(ctx, x, o.a, &al, &ah) || !op_interval(ctx, x, o.b, &bl, &bh)) return false;
            __int128 l, h;
            if (o.code == SDQH_X_ADD) { l = (__int128)al + bl; h = (__int128)ah + bh; }
            else if (o.code == SDQH_X_SUB) { l = (__int128)al - bh; h = (__int128)ah - bl; }
            else {
                const __int128 c4[4] = {(__int128)al * bl, (__int128)al * bh, (__int128)ah * bl, (__int128)ah * bh};
                l = h = c4[0];
                for (int i = 1; i < 4; ++i) { l = c4[i] < l ? c4[i] : l; h = c4[i] > h ? c4[i] : h; }
            }
            if (!fits(l, h)) return false;
            *lo = (int64_t)l; *hi = (int64_t)h; return true;
        }
        case SDQH_X_NEG: if (!op_interval(ctx, x, o.a, &al, &ah)) return false; *lo = -ah; *hi = -al; return true;
        case SDQH_X_YEAR: if (!op_interval(ctx, x, o.a, &al, &ah) || al < 0) return false; *lo = al / 10000; *hi = ah / 10000; return true;
        case SDQH_X_DIVI: if (!op_interval(ctx, x, o.a, &al, &ah) || al < 0) return false; *lo = al / o.imm_i; *hi = ah / o.imm_i; return true;
        case SDQH_X_MODI: if (!op_interval(ctx, x, o.a, &al, &ah) || al < 0) return false; *lo = 0; *hi = std::min<int64_t>(ah, o.imm_i - 1); return true;
        case SDQH_X_SELECT:
            if (!op_interval(ctx, x, o.b, &al, &ah) || !op_interval(ctx, x, o.c, &bl, &bh)) return false;
            *lo = std::min(al, bl); *hi = std::max(ah, bh); return true;
        default: return false;
    }
}

// rank of a constant among a column's distinct values, for `value OP constant` in code space.  The dictionary is ascending in
// the column's own order (int64, or doubles: the two-decimal values are ordered like their cents).
template <class T> void translate_cmp(const std::vector<int64_t>& dict, bool f64, T cst, int op, int* kind, uint32_t* t) {
    auto val = [&](size_t i) -> T { if constexpr (sizeof(T) == 8 && T(0.5) != T(0)) { double d; std::memcpy(&d, &dict[i], 8); return (T)d; } else return (T)dict[i]; };
    (void)f64;
    const size_t n = dict.size();
    size_t lb = 0, ub = n;                                             // first index with value >= cst / > cst (binary searches: ascending values)
    { size_t lo = 0, hi = n; while (lo < hi) { const size_t mid = (lo + hi) / 2; if (val(mid) < cst) lo = mid + 1; else hi = mid; } lb = lo; }
    { size_t lo = lb, hi = n; while (lo < hi) { const size_t mid = (lo + hi) / 2; if (!(cst < val(mid))) lo = mid + 1; else hi = mid; } ub = lo; }
    const bool nan = cst != cst;
    switch (op) {
        case SDQH_X_LT: *kind = 0; *t = nan ? 0u : (uint32_t)lb; break;           // value <  cst  <=>  code <  lb
        case SDQH_X_LE: *kind = 0; *t = nan ? 0u : (uint32_t)ub; break;           // value <= cst  <=>  code <  ub
        case SDQH_X_GT: *kind = 1; *t = nan ? 0xFFFFFFFFu : (uint32_t)ub; break;  // value >  cst  <=>  code >= ub
        case SDQH_X_GE: *kind = 1; *t = nan ? 0xFFFFFFFFu : (uint32_t)lb; break;  // value >= cst  <=>  code >= lb
        case SDQH_X_EQ: *kind = 2; *t = (!nan && lb < ub) ? (uint32_t)lb : 0xFFFFFFFFu; break;
        default:        *kind = 3; *t = (!nan && lb < ub) ? (uint32_t)lb : 0xFFFFFFFFu; break;
    }
}
int mirrored(int op) { return op == SDQH_X_LT ? SDQH_X_GT : op == SDQH_X_LE ? SDQH_X_GE : op == SDQH_X_GT ? SDQH_X_LT : op == SDQH_X_GE ? SDQH_X_LE : op; }

// Do aligned groups of 8 consecutive rows of an I64 column hold values within a span of 96 (a foreign key of a table stored in its
// parent's order: l_orderkey)?  Then the 8 rows a lane of the tight skeletons takes meet at most four consecutive words of a key
// bitmap.  Sampled once per column (2048 groups spread over the column, one small download), cached; at most 2 outliers allowed —
// a wave tests 128 groups per step and every outlier costs it eight single-word requests.
bool column_span8(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->span8 >= 0) return c->span8 == 1;
    if (c->dtype != SDQH_I64 || c->transient || c->nrows < 64) { c->span8 = 0; return false; }
    if (ctx->capturing) return false;                                     // (sampled with a wait: not inside a recording; uncached)
    const int samples = (int)std::min<int64_t>(1024, c->nrows / 8);        // 1024 groups x 64 bytes = the 64 KiB pinned result block
    const int64_t groups = c->nrows / 8, step = std::max<int64_t>(1, groups / samples);
    int64_t* host = static_cast<int64_t*>(ctx->result_host);
    bool ok = true;
    for (int i = 0; i < samples && ok; ++i)
        ok = hipMemcpyAsync(host + 8 * i, static_cast<const int64_t*>(c->data) + (int64_t)i * step * 8, 64, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess;
    if (!ok || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); c->span8 = 0; return false; }
    int wide = 0;
    for (int i = 0; i < samples; ++i) {
        int64_t lo = host[8 * i], hi = lo;
        for (int j = 1; j < 8; ++j) { lo = std::min(lo, host[8 * i + j]); hi = std::max(hi, host[8 * i + j]); }
        wide += (hi - lo > 96) ? 1 : 0;
    }
    c->span8 = wide <= 1 ? 1 : 0;
    return c->span8 == 1;
}

// RUN INDEX of a never-decreasing I64 column (sdqh_column::run_index): per value v of [mn, mx + 1] the first row holding a value >= v
// (mx - mn + 2 entries; the rows of v are index[v - mn] .. index[v - mn + 1]).  Built once per column on first need (a binary search
// per value over the 4-byte twin), kept with the column.  Only where it pays its memory: fewer than 2^31 rows, a value range of at
// most 64 x the rows.
__global__ __launch_bounds__(256) void k_run_index(const int32_t* __restrict__ twin, int64_t n, int64_t lo, int64_t nvals, uint32_t* __restrict__ ridx) {
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvals; v += (int64_t)gridDim.x * 256) {
        int64_t a = 0, b = n;
        while (a < b) { const int64_t mid = (a + b) >> 1; if ((int64_t)twin[mid] - lo < v) a = mid + 1; else b = mid; }
        ridx[v] = (uint32_t)a;
    }
}
// DELTA twin (sdqh_column::delta8): per aligned group of 8 rows the smallest value of the group's 4-byte twin and eight one-byte offsets from it;
// a group that spans more than 255 sets the flag and the column keeps its 4-byte twin.  Rows behind the end repeat the last row.
__global__ __launch_bounds__(256) void k_delta8(const int32_t* __restrict__ twin, int64_t n, int64_t ngroups, uint32_t* __restrict__ out, int* __restrict__ flag) {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (int64_t)gridDim.x * 256) {
        int32_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int64_t r = g * 8 + i; v[i] = twin[r < n ? r : n - 1]; }
        int32_t lo = v[0], hi = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) { lo = v[i] < lo ? v[i] : lo; hi = v[i] > hi ? v[i] : hi; }
        if ((int64_t)hi - (int64_t)lo > 255) { *flag = 1; continue; }
        uint32_t w1 = 0, w2 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) { w1 |= (uint32_t)(v[i] - lo) << (8 * i); w2 |= (uint32_t)(v[4 + i] - lo) << (8 * i); }
        out[g * 3] = (uint32_t)lo; out[g * 3 + 1] = w1; out[g * 3 + 2] = w2;
    }
}
const void* column_delta8(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->delta8_state >= 0) return c->delta8_state == 1 ? c->delta8 : nullptr;
    if (ctx->capturing) return nullptr;
    c->delta8_state = 0;
    if (c->dtype != SDQH_I64 || c->transient || c->nrows < 8) return nullptr;
    const int32_t* twin = static_cast<const int32_t*>(column_narrow(ctx, c));
    if (!twin) return nullptr;
    const int64_t ngroups = (c->nrows + 7) / 8;
    uint32_t* out = static_cast<uint32_t*>(attach_alloc(ctx, c, (size_t)ngroups * 12 + 64 * 12 + 64));   // (+ 64 groups: in a skeleton's last step every lane loads its record, also the lanes behind the last row — masked afterwards, but read)
    int* flag = static_cast<int*>(pool_alloc(ctx, 64));
    bool ok = out && flag && hipMemsetAsync(flag, 0, 4, ctx->stream) == hipSuccess;
    if (ok) {
        const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((ngroups + 255) / 256, (int64_t)ctx->num_cu * 16));
        hipLaunchKernelGGL(k_delta8, dim3(grid), dim3(256), 0, ctx->stream, twin, c->nrows, ngroups, out, flag);
        int* host = static_cast<int*>(ctx->result_host);
        ok = hipMemcpyAsync(host, flag, 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess && host[0] == 0;
    }
    if (!ok) (void)hipGetLastError();
    if (flag) pool_free(ctx, flag);
    if (ok) { c->delta8 = out; c->delta8_state = 1; }
    else if (out) attach_free(ctx, c, out);
    return ok ? out : nullptr;
}
const uint32_t* column_run_index(sdqh_ctx* ctx, sdqh_column* c) {
    if (c->run_index_state >= 0) return c->run_index_state == 1 ? static_cast<const uint32_t*>(c->run_index) : nullptr;
    if (ctx->capturing) return nullptr;                                   // (decided outside a recording)
    c->run_index_state = 0;
    if (c->dtype != SDQH_I64 || c->transient || c->nrows < 2 || c->nrows >= ((int64_t)1 << 31)) return nullptr;
    if (!column_nondecreasing(ctx, c) || column_minmax(ctx, c) != SDQH_OK || c->mx < c->mn) return nullptr;
    const uint64_t range = (uint64_t)(c->mx - c->mn) + 1;
    if (range > 0xFFFFFFF0ull || range > 64ull * (uint64_t)c->nrows) return nullptr;
    const int32_t* twin = static_cast<const int32_t*>(column_narrow(ctx, c));
    if (!twin) return nullptr;
    uint32_t* ridx = static_cast<uint32_t*>(attach_alloc(ctx, c, ((size_t)range + 1) * 4 + 64));
    if (!ridx) return nullptr;
    const int64_t nvals = (int64_t)range + 1;
    const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nvals + 255) / 256, (int64_t)ctx->num_cu * 16));
    { KernelScope ks(ctx, "k_run_index"); hipLaunchKernelGGL(k_run_index, dim3(grid), dim3(256), 0, ctx->stream, twin, c->nrows, c->mn, nvals, ridx); }
    // the column is shared by every lane (context) of the family: publish the index when it is WRITTEN, not when it is queued — another lane's
    // walk on another stream would read it half-built (delta twins and narrow twins wait the same way)
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); attach_free(ctx, c, ridx); return nullptr; }
    c->run_index = ridx; c->run_index_state = 1;
    return ridx;
}

// Decide, per column of a register program, the tightest exact encoding it can be streamed in, and rewrite the comparisons
// of coded columns with constants.  The decisions are part of the kernel's structure; the translated constants are arguments
// (recomputed at every call: cheap, and they follow the constants).
void tight_plan(sdqh_ctx* ctx, int64_t nrows, XInfo* x) {
    const sdqh_program* p = x->p;
    for (int k = 0; k < p->nops; ++k) x->cmp_cc[k] = x->cmp_kind[k] = x->cmp_col[k] = -1;
    for (int c = 0; c < SDQH_MAX_XCOLS; ++c) { x->enc[c] = ENC_RAW; x->dict_slot[c] = -1; }
    x->tight = false; x->nd = 0; x->ncc = 0; x->gather32 = 0; x->pref32 = false; x->pnear = false; x->pwin = false; x->driven = false; x->driven_col = -1; x->phash = false;
    // (a compile-only context has no columns to code; SDQLPY_AMD_FAKE_CODES makes it pretend every column is coded — 2 bytes where
    //  only compared, 1 byte where its value is used — so that build() proves on a host without a GPU that the generator's tight
    //  output compiles for gfx950)
    const bool fake = ctx->compile_only && std::getenv("SDQLPY_AMD_FAKE_CODES") != nullptr;
    static const std::vector<int64_t> fake_dict = {0, 1, 2, 3};
    const bool regs_all = x->direct || x->vstage;                         // every column is streamed and every operation evaluated on registers
    if (!ctx->opt_tight || x->ncols < 1 || (!regs_all && x->scols.empty())) return;
    // Every register program runs on the tight skeleton, and every queue program that streams anything on the tight queue,
    // whatever its columns' encodings: rows meet lanes, partial sums are folded and queues are drained in the same order with and
    // without twins, so switching the twins off changes no bit of a sum (tests/test_hip_parity.py).
    x->tight = true;
    x->fake = fake;
    uint32_t text_twins = 0;
    for (int c = 0; c < x->ncols; ++c) if (x->cols[c]->dtype == SDQH_STR) text_twins |= x->narrow_mask & (1u << c);
    x->narrow_mask = text_twins;                                           // (the 4-byte twins analyse chose were for the two-rows-per-lane skeletons)
    x->scope.assign((size_t)p->nops, regs_all ? 1 : 0);
    std::vector<char> streamed((size_t)x->ncols, regs_all ? 1 : 0);
    if (!regs_all) {
        for (int g = 0; g < x->nstream_gates; ++g) closure(p, p->gates[g], x->scope);
        if (x->prefilter_part0 >= 0) closure(p, x->prefilter_part0, x->scope);
        for (int c : x->scols) streamed[(size_t)c] = 1;
    }
    if (!ctx->opt_narrow || (ctx->compile_only && !fake) || (!fake && nrows < ctx->opt_feature_min_rows)) return;
    // how every streamed COL operation is used by what is evaluated on the streamed registers
    std::vector<char> cmp_only((size_t)x->ncols, 1), used((size_t)x->ncols, 0);
    auto direct_ref = [&](int k) {
        if (!regs_all) {                                                   // a streamed gate that IS the column cannot be (gates are bool); the prefilter reads its key's value
            return k == x->prefilter_part0;
        }
        if (p->key == k) return true;
        for (int g = 0; g < p->ngates; ++g) if (p->gates[g] == k) return true;
        for (int v = 0; v < p->nvals; ++v) if (p->vals[v] == k) return true;
        return false;
    };
    for (int k = 0; k < p->nops; ++k) {
        if (p->ops[k].code != SDQH_X_COL || !x->scope[(size_t)k]) continue;
        const int c = x->col_of[k];
        if (!streamed[(size_t)c]) continue;
        used[(size_t)c] = 1;
        if (direct_ref(k)) cmp_only[(size_t)c] = 0;
        for (int j = k + 1; j < p->nops; ++j) {
            const sdqh_xop& u = p->ops[j];
            if (!x->scope[(size_t)j] || u.code == SDQH_X_COL || u.code == SDQH_X_CONST || u.code == SDQH_X_ROWID) continue;
            const bool reads = u.a == k || u.b == k || (u.code == SDQH_X_SELECT && u.c == k);
            if (!reads) continue;
            const int other = u.a == k ? u.b : u.a;
            if (!(is_cmp(u.code) && other >= 0 && other < j && p->ops[other].code == SDQH_X_CONST && p->ops[other].type == p->ops[k].type)) cmp_only[(size_t)c] = 0;
        }
    }
    bool any = false;
    for (int c = 0; c < x->ncols; ++c) {
        if (!used[(size_t)c]) continue;
        sdqh_column* col = const_cast<sdqh_column*>(x->cols[c]);
        if (col->dtype == SDQH_STR) continue;                              // (text is never streamed)
        if (fake) {
            x->enc[c] = cmp_only[(size_t)c] ? ENC_C16 : (c % 3 == 2 ? ENC_N32 : ENC_C8);
            if (x->enc[c] == ENC_C8) x->dict_slot[c] = x->nd++;
        } else if (column_codes(ctx, col) && (cmp_only[(size_t)c] || col->code_width == 1)) {
            x->enc[c] = col->code_width == 1 ? ENC_C8 : ENC_C16;
            if (!cmp_only[(size_t)c]) x->dict_slot[c] = x->nd++;
        } else if (column_narrow(ctx, col)) x->enc[c] = ENC_N32;
        any = any || x->enc[c] != ENC_RAW;
    }
    if (!regs_all && !fake) {
        // columns the drain reads by row (keys and payloads of a build, operands of a probe hit): survivors are a few per cent of the
        // rows, scattered, so a gather moves whole lines for single values — through the 4-byte twin the same lines are half as many
        std::vector<char> by_row((size_t)x->ncols, 0);
        for (int k = 0; k < p->nops; ++k) if (p->ops[k].code == SDQH_X_COL && !x->scope[(size_t)k]) by_row[(size_t)x->col_of[k]] = 1;
        for (int c = 0; c < x->ncols; ++c)
            if (by_row[(size_t)c] && x->cols[c]->dtype != SDQH_STR && column_narrow(ctx, const_cast<sdqh_column*>(x->cols[c]))) { x->gather32 |= 1u << c; any = true; }
    }
    if (!any) return;
    for (int c = 0; c < x->ncols; ++c) if (x->enc[c] == ENC_N32) x->narrow_mask |= 1u << c;
    for (int c = 0; c < x->ncols; ++c) {
        const sdqh_column* col = x->cols[c];
        if (x->enc[c] < ENC_C16 || col->dtype != SDQH_I64) continue;
        if (fake) { x->affine[c] = c % 2 == 0; x->dlo[c] = 0; continue; }
        const std::vector<int64_t>& d = col->dict_host;
        x->affine[c] = !d.empty() && d.back() - d.front() == (int64_t)d.size() - 1;        // (ascending and distinct: consecutive)
        x->dlo[c] = d.empty() ? 0 : d.front();
    }
    for (int k = 0; k < p->nops; ++k) {
        int64_t lo = 0, hi = 0;
        if (x->scope[(size_t)k] && p->ops[k].type == SDQH_T_I64 && op_interval(ctx, *x, k, &lo, &hi))
            x->irange[k] = (lo >= -(1 << 23) && hi < (1 << 23)) ? 2 : (lo >= INT32_MIN && hi <= INT32_MAX) ? 1 : 0;
    }
    if (!regs_all && x->prefilter_op >= 0 && !x->prefilter_composite && !fake) {
        const sdqh_table* t = x->tabs[x->tab_of[x->prefilter_op]];
        const bool applicable = t->dev.bm && t->dev.bm_shift == 0 && t->dev.lin_rb == 0;
        x->pref32 = applicable && x->irange[x->prefilter_part0] >= 1 && t->dev.bm_lo >= INT32_MIN && t->dev.bm_hi <= INT32_MAX && t->dev.bm_hi >= t->dev.bm_lo;
        // a hash-layout table (keys far apart, in no order, or the direct layouts switched off): its hashed filter, whatever the key's width
        x->phash = !x->pref32 && !t->dev.bm && t->dev.hf != nullptr && !t->bitmap_only;
    }
    if (fake && !regs_all && x->prefilter_op >= 0 && !x->prefilter_composite) x->pref32 = true;
    static const bool xdebug = std::getenv("SDQLPY_AMD_X_DEBUG") != nullptr;
    if (xdebug && !regs_all && x->prefilter_op >= 0) {
        const sdqh_table* t = x->tabs[x->tab_of[x->prefilter_op]];
        std::fprintf(stderr, "[x] prefilter: composite %d bm %p shift %d lin_rb %lld range [%lld, %lld] irange %d pref32 %d gates %d rows %lld\n", (int)x->prefilter_composite, (const void*)t->dev.bm,
                     (int)t->dev.bm_shift, (long long)t->dev.lin_rb, (long long)t->dev.bm_lo, (long long)t->dev.bm_hi, (int)x->irange[x->prefilter_part0], (int)x->pref32, x->nstream_gates, (long long)nrows);
    }
    if (x->pref32) {
        const sdqh_xop& ko = p->ops[x->prefilter_part0];
        x->pnear = fake ? true : (ko.code == SDQH_X_COL && column_span8(ctx, const_cast<sdqh_column*>(ko.col)));
        x->pwin = fake ? std::getenv("SDQLPY_AMD_FAKE_WINDOW") != nullptr : (ctx->opt_window != 0 && x->pnear);
        // the driven walk (x_queue8): no other streamed condition, the key a plain column stored in its own order
        if (!fake && x->want_driven && ctx->opt_x_driven > 0 && x->nstream_gates == 0 && ko.code == SDQH_X_COL) {
            const int c = x->col_of[x->prefilter_part0];
            sdqh_column* kc = const_cast<sdqh_column*>(x->cols[c]);
            if (kc->dtype == SDQH_I64 && kc->nrows == nrows && nrows < ((int64_t)1 << 31) && column_run_index(ctx, kc)) { x->driven = true; x->driven_col = c; }      // (whatever encoding the key is streamed in: the walk does not read it)
            if (xdebug) std::fprintf(stderr, "[x] driven walk: enc %d column rows %lld run index %d -> %d\n", x->enc[c], (long long)kc->nrows, kc->run_index_state, (int)x->driven);
        }
    }
    // queue programs: the prefilter's key through its delta twin where it has one (the stream's widest column: 4 -> 1.5 bytes per row); the rows
    // that are drained go on reading the 4-byte twin by row (gather32)
    if (!regs_all && !fake && !x->vstage && ctx->opt_delta8 && x->pref32 && !x->driven) {
        const sdqh_xop& ko = p->ops[x->prefilter_part0];
        if (ko.code == SDQH_X_COL) {
            const int c = x->col_of[x->prefilter_part0];
            sdqh_column* kc = const_cast<sdqh_column*>(x->cols[c]);
            if (x->enc[c] == ENC_N32 && kc->nrows == nrows && x->pnear && column_delta8(ctx, kc)) { x->enc[c] = ENC_D8; x->narrow_mask &= ~(1u << c); x->gather32 |= 1u << c; }
        }
    }
    // ... and a value-queue build (x_vstage8: every column is streamed, nothing is read by row) every 4-byte integer column whose 8-row groups
    // are narrow (the build's key on a table stored in its order: o_orderkey)
    if (x->vstage && !fake && ctx->opt_delta8) {
        for (int c = 0; c < x->ncols; ++c) {
            sdqh_column* kc = const_cast<sdqh_column*>(x->cols[c]);
            if (x->enc[c] == ENC_N32 && kc->dtype == SDQH_I64 && kc->nrows == nrows && column_span8(ctx, kc) && column_delta8(ctx, kc)) { x->enc[c] = ENC_D8; x->narrow_mask &= ~(1u << c); }
        }
    }
    // comparisons of a coded column with a constant, in code space
    for (int j = 0; j < p->nops; ++j) {
        const sdqh_xop& u = p->ops[j];
        if (!is_cmp(u.code) || !x->scope[(size_t)j]) continue;
        int kc = -1, kk = -1, op = u.code;
        if (p->ops[u.a].code == SDQH_X_COL && p->ops[u.b].code == SDQH_X_CONST) { kc = u.a; kk = u.b; }
        else if (p->ops[u.b].code == SDQH_X_COL && p->ops[u.a].code == SDQH_X_CONST) { kc = u.b; kk = u.a; op = mirrored(op); }
        if (kc < 0) continue;
        const int c = x->col_of[kc];
        if (x->enc[c] < ENC_C16 || p->ops[kk].type != p->ops[kc].type || x->ncc >= X_MAX_CONST) continue;
        const sdqh_column* col = x->cols[c];
        int kind = 0; uint32_t t = 0;
        const std::vector<int64_t>& dict = fake ? fake_dict : col->dict_host;
        if (col->dtype == SDQH_F64) translate_cmp<double>(dict, true, p->ops[kk].imm_f, op, &kind, &t);
        else translate_cmp<int64_t>(dict, false, p->ops[kk].imm_i, op, &kind, &t);
        x->cmp_cc[j] = x->ncc; x->cmp_kind[j] = kind; x->cmp_col[j] = c; x->cc[x->ncc++] = t;
    }
}

// ---- code generation ----------------------------------------------------------------------------------
struct Gen {
    const XInfo& x; std::ostringstream os; std::vector<char> done; int mode = 0;      // mode 0: columns gathered by row r; 1: from streamed registers, half H; 2: stest (explicit .x / .y)
    const char* half = "";
    std::vector<int> slot_of;                  // column -> slot in the streamed register array
    std::vector<int> sres_of;                  // text operation -> its slot in the drain's staged results (-1: reads the column in global memory)
    // TIGHT (mode 3): an operation that depends on ONE byte-coded column and constants only is a table of <= 256 entries, filled once per
    // workgroup by evaluating it on the dictionary (mode 4: the column reads as the dictionary entry `dv`): `1.0 - l_discount` costs the
    // row one LDS read, like the column's value itself
    std::vector<std::pair<int, int>> tabs;     // table -> (operation, column)
    std::vector<int> tab_of, single_memo;
    int dict_col = -1;                         // mode 4: the column being tabulated
    static constexpr int MAX_TABS = 16, MAX_DERIVED = 12;
    // mode 0 (a drained row, evaluated by its own lane): the loads the program needs at one depth of its dependency chain are emitted
    // together and pinned (x_pin): `want` = the operations the gates depend on; after a LOOKUP, the payload fields of its entry that are
    // wanted are read at once
    bool pinning = false; std::vector<char> want; std::vector<std::string> pins;
    void flush_pins() {
        for (size_t i = 0; i < pins.size(); i += 6) {
            os << "        x_pin(";
            for (size_t j = i; j < pins.size() && j < i + 6; ++j) os << (j > i ? ", " : "") << pins[j];
            os << ");\n";
        }
        pins.clear();
    }
    bool is_row_load(int k) const {                                     // a by-row load of a numeric column / a payload field / an accumulator
        const sdqh_xop& o = x.p->ops[k];
        if (o.code == SDQH_X_COL) return x.cols[x.col_of[k]]->dtype != SDQH_STR;
        return o.code == SDQH_X_FIELD || o.code == SDQH_X_ACC;
    }
    std::string lay_lit(int t) const { char b[24]; std::snprintf(b, sizeof(b), "0x%08xu", (unsigned)x.lay[t]); return b; }      // the table's layout as a template argument
    explicit Gen(const XInfo& xi) : x(xi), done((size_t)xi.p->nops, 0), slot_of((size_t)SDQH_MAX_XCOLS, -1), sres_of((size_t)xi.p->nops, -1),
                                    tab_of((size_t)xi.p->nops, -1), single_memo((size_t)xi.p->nops, -2) {}
    // the one byte-coded column operation k depends on (nothing else but constants), or -1
    int single_col(int k) {
        if (k < 0) return -1;
        if (single_memo[(size_t)k] != -2) return single_memo[(size_t)k];
        const sdqh_xop& o = x.p->ops[k];
        int r = -1;
        if (o.code == SDQH_X_COL) r = (x.enc[x.col_of[k]] == ENC_C8 && !x.affine[x.col_of[k]]) ? x.col_of[k] : -1;      // (consecutive values: code + offset, no table)
        else if (o.code == SDQH_X_CONST) r = -3;                              // no column at all
        else if (o.code == SDQH_X_ROWID || o.code == SDQH_X_PACK2 || !op_is_light(o)) r = -1;       // (PACK2 can fail per row: never a table)
        else {
            r = -3;
            const int opnd[3] = {o.a, o.b, o.code == SDQH_X_SELECT ? o.c : -1};
            for (int j : opnd) {
                if (j < 0 || j >= k) continue;
                const int c = single_col(j);
                if (c == -1 || (c >= 0 && r >= 0 && c != r)) { r = -1; break; }
                if (c >= 0) r = c;
            }
        }
        return single_memo[(size_t)k] = r;
    }
    // the text operation itself, on the field at `field` ("pointer, width"); lds: the field was staged in LDS
    std::string text_op(int k, const std::string& field, bool lds = false, bool bytes = false) const {
        const sdqh_xop& o = x.p->ops[k];
        if (bytes) {                                                        // field = "region, off, width": the byte twin staged in LDS
            if (o.code == SDQH_X_STR) return "x8_str_pred(" + field + ", a.spool + " + std::to_string(x.str_off[k]) + ", " + std::to_string(o.slen) + ", " + std::to_string(o.aux) + ")";
            if (o.code == SDQH_X_STRIDX) return "x8_first_index(" + field + ", a.spool + " + std::to_string(x.str_off[k]) + ", " + std::to_string(o.slen) + ")";
            return "x8_char(" + field + ", " + std::to_string(o.aux) + ")";
        }
        if (lds && o.code == SDQH_X_STR) return "lds_str_pred(" + field + ", a.spool + " + std::to_string(x.str_off[k]) + ", " + std::to_string(o.slen) + ", " + std::to_string(o.aux) + ")";
        if (lds && o.code == SDQH_X_STRIDX) return "lds_first_index(" + field + ", a.spool + " + std::to_string(x.str_off[k]) + ", " + std::to_string(o.slen) + ")";
        if (o.code == SDQH_X_STR) return "str_pred(" + field + ", a.spool + " + std::to_string(x.str_off[k]) + ", " + std::to_string(o.slen) + ", " + std::to_string(o.aux) + ")";
        if (o.code == SDQH_X_STRIDX) return "x_first_index(" + field + ", a.spool + " + std::to_string(x.str_off[k]) + ", " + std::to_string(o.slen) + ")";
        return "x_char(" + field + ", " + std::to_string(o.aux) + ")";
    }

    static const char* ctype(int t) { return t == SDQH_T_F64 ? "double" : t == SDQH_T_BOOL ? "bool" : "int64_t"; }
    std::string bad(int k) const {
        const int c = x.p->ops[k].code;
        return (c == SDQH_X_PACK2 || c == SDQH_X_SELECT) ? "b" + std::to_string(k) : std::string("false");
    }
    std::string col_expr(int k) {
        const sdqh_xop& o = x.p->ops[k];
        const int c = x.col_of[k];
        std::string raw;
        // (a column the loop also STREAMS through its 4-byte twin — the prefilter's key — is read again from that twin by the rows that
        //  are drained: the lines the wave has just streamed, still in L2, not the 8-byte original's)
        if (mode == 0 && (((x.gather32 | x.narrow_mask) >> c) & 1u))
            return o.type == SDQH_T_F64 ? "narrow_decode(static_cast<const int32_t*>(a.ncol[" + std::to_string(c) + "])[r])" : "(int64_t)static_cast<const int32_t*>(a.ncol[" + std::to_string(c) + "])[r]";
        if (mode == 0) return std::string("static_cast<const ") + (o.type == SDQH_T_F64 ? "double" : "int64_t") + "*>(a.col[" + std::to_string(c) + "])[r]";
        if (mode == 4) return o.type == SDQH_T_F64 ? "x_f(dv)" : "dv";         // tabulating: the dictionary entry
        if (mode == 3) {                                                   // TIGHT: row i of the lane's 8, out of the packed words s.c<slot>
            const std::string w = "s.c" + std::to_string(slot_of[(size_t)c]);
            if (x.affine[c] && x.enc[c] >= ENC_C16) return std::string("((int64_t)") + (x.enc[c] == ENC_C8 ? "xt_u8(" : "xt_u16(") + w + ", i) + a.dlo[" + std::to_string(c) + "])";
            switch (x.enc[c]) {
                case ENC_C8: {                                               // (no table left: the dictionary in global memory, L1-resident)
                    const std::string e = "a.dict[" + std::to_string(c) + "][xt_u8(" + w + ", i)]";
                    return o.type == SDQH_T_F64 ? "x_f(" + e + ")" : e;
                }
                case ENC_D8: return "(int64_t)xt_d8(" + w + ", i)";
                case ENC_N32: return o.type == SDQH_T_F64 ? "narrow_decode(xt_i32(" + w + ", i))" : "(int64_t)xt_i32(" + w + ", i)";
                case ENC_RAW: return o.type == SDQH_T_F64 ? "x_f(xt_i64(" + w + ", i))" : "xt_i64(" + w + ", i)";
                default: return "0 /* a 16-bit code has no value form */";
            }
        }
        if (mode == 1) raw = "(H == 0 ? s[" + std::to_string(slot_of[(size_t)c]) + "].x : s[" + std::to_string(slot_of[(size_t)c]) + "].y)";
        else raw = "s[" + std::to_string(slot_of[(size_t)c]) + "]." + half;
        return o.type == SDQH_T_F64 ? "x_f(" + raw + ")" : raw;
    }
    void emit(int k) {
        if (k < 0 || done[(size_t)k]) return;
        const sdqh_xop& o = x.p->ops[k];
        auto v = [](int j) { return "v" + std::to_string(j); };
        const std::string K = std::to_string(k);
        std::string e;
        if (mode == 3 && !(is_cmp(o.code) && x.cmp_cc[k] >= 0)) {
            const int c = single_col(k);
            const bool bare = o.code == SDQH_X_COL;
            if (c >= 0 && (int)tabs.size() < (bare ? MAX_TABS : MAX_DERIVED)) {
                if (tab_of[(size_t)k] < 0) { tab_of[(size_t)k] = (int)tabs.size(); tabs.push_back({k, c}); }
                const std::string cell = "tab[" + std::to_string(tab_of[(size_t)k]) + "][xt_u8(s.c" + std::to_string(slot_of[(size_t)c]) + ", i)]";
                e = o.type == SDQH_T_F64 ? "x_f(" + cell + ")" : o.type == SDQH_T_BOOL ? "(" + cell + " != 0)" : cell;
                os << "        const " << ctype(o.type) << " v" << K << " = " << e << ";\n";
                if (o.code == SDQH_X_SELECT) os << "        const bool b" << K << " = false;\n";      // (nothing under a tabulated SELECT can fail: PACK2 is never tabulated)
                done[(size_t)k] = 1;
                return;
            }
        }
        switch (o.code) {
            case SDQH_X_COL: e = col_expr(k); break;
            case SDQH_X_ROWID: e = "r"; break;
            case SDQH_X_CONST:
                e = o.type == SDQH_T_F64 ? "a.cf[" + std::to_string(x.const_of[k]) + "]" : (o.type == SDQH_T_BOOL ? "(a.ci[" + std::to_string(x.const_of[k]) + "] != 0)" : "a.ci[" + std::to_string(x.const_of[k]) + "]");
                break;
            case SDQH_X_LOOKUP:
                emit(o.a);
                os << "        const uint32_t e" << K << " = x_lookup_l<" << lay_lit(x.tab_of[k]) << ">(a.tab[" << x.tab_of[k] << "], " << v(o.a) << ", " << bad(o.a) << ");\n";
                e = "(e" + K + " != NO_ROW)";
                break;
            case SDQH_X_FIELD:
                emit(o.a);
                e = "x_field(a.tab[" + std::to_string(x.tab_of[o.a]) + "], " + std::to_string(o.aux) + ", e" + std::to_string(o.a) + ")";
                if (o.type == SDQH_T_F64) e = "x_f(" + e + ")";
                break;
            case SDQH_X_ACC:
                emit(o.a);
                e = o.aux < 0 ? "x_hits_l<" + lay_lit(x.tab_of[o.a]) + ">(a.tab[" + std::to_string(x.tab_of[o.a]) + "], e" + std::to_string(o.a) + ")"
                              : "x_acc_l<" + lay_lit(x.tab_of[o.a]) + ">(a.tab[" + std::to_string(x.tab_of[o.a]) + "], " + std::to_string(o.aux) + ", e" + std::to_string(o.a) + ")";
                break;
            case SDQH_X_ADD: case SDQH_X_SUB: case SDQH_X_MUL: {
                emit(o.a); emit(o.b);
                const char* sym = o.code == SDQH_X_ADD ? " + " : o.code == SDQH_X_SUB ? " - " : " * ";
                // integers whose ranges (from the columns' own minima / maxima) fit 32 bits: 32-bit arithmetic, 24-bit multiplies
                if (mode == 3 && o.type == SDQH_T_I64 && x.irange[k] >= 1 && x.irange[o.a] >= 1 && x.irange[o.b] >= 1) {
                    if (o.code == SDQH_X_MUL && x.irange[o.a] == 2 && x.irange[o.b] == 2) e = "(int64_t)__mul24((int)" + v(o.a) + ", (int)" + v(o.b) + ")";
                    else e = "(int64_t)((int32_t)" + v(o.a) + sym + "(int32_t)" + v(o.b) + ")";
                } else e = "(" + v(o.a) + sym + v(o.b) + ")";
                break;
            }
            case SDQH_X_DIV: emit(o.a); emit(o.b); e = "(" + v(o.a) + " / " + v(o.b) + ")"; break;
            case SDQH_X_NEG: emit(o.a); e = "(-" + v(o.a) + ")"; break;
            case SDQH_X_I2F: emit(o.a); e = "(double)" + v(o.a); break;
            case SDQH_X_YEAR: emit(o.a); e = "(" + v(o.a) + " / 10000)"; break;
            case SDQH_X_DIVI: case SDQH_X_MODI:                             // the divisor is a literal: the compiler turns it into a multiply / a mask
                emit(o.a); e = "(" + v(o.a) + (o.code == SDQH_X_DIVI ? " / " : " % ") + "(int64_t)" + std::to_string((long long)o.imm_i) + "ll)"; break;
            case SDQH_X_PACK2:
                emit(o.a); emit(o.b);
                os << "        const bool b" << K << " = " << bad(o.a) << " || " << bad(o.b) << " || " << v(o.a) << " < 0 || " << v(o.a) << " > 0xFFFFFFFFll || " << v(o.b) << " < 0 || " << v(o.b) << " > 0xFFFFFFFFll;\n";
                e = "(int64_t)(((uint64_t)" + v(o.a) + " << 32) | ((uint64_t)" + v(o.b) + " & 0xFFFFFFFFull))";
                break;
            case SDQH_X_LT: case SDQH_X_LE: case SDQH_X_GT: case SDQH_X_GE: case SDQH_X_EQ: case SDQH_X_NE:
                if (mode == 3 && x.cmp_cc[k] >= 0) {                        // a coded column against a constant: the code against the constant's rank
                    const int c = x.cmp_col[k];
                    const std::string code = std::string(x.enc[c] == ENC_C8 ? "xt_u8" : "xt_u16") + "(s.c" + std::to_string(slot_of[(size_t)c]) + ", i)";
                    static const char* rel[4] = {" < ", " >= ", " == ", " != "};
                    e = "(" + code + rel[x.cmp_kind[k]] + "a.cc[" + std::to_string(x.cmp_cc[k]) + "])";
                    break;
                }
                emit(o.a); emit(o.b);
                e = "(" + v(o.a) + (o.code == SDQH_X_LT ? " < " : o.code == SDQH_X_LE ? " <= " : o.code == SDQH_X_GT ? " > " : o.code == SDQH_X_GE ? " >= " : o.code == SDQH_X_EQ ? " == " : " != ") + v(o.b) + ")";
                break;
            case SDQH_X_AND: emit(o.a); emit(o.b); e = "(" + v(o.a) + " && " + v(o.b) + ")"; break;
            case SDQH_X_OR: emit(o.a); emit(o.b); e = "(" + v(o.a) + " || " + v(o.b) + ")"; break;
            case SDQH_X_NOT: emit(o.a); e = "(!" + v(o.a) + ")"; break;
            case SDQH_X_SELECT:
                emit(o.a); emit(o.b); emit(o.c);
                os << "        const bool b" << K << " = " << v(o.a) << " ? " << bad(o.b) << " : " << bad(o.c) << ";\n";
                e = "(" + v(o.a) + " ? " + v(o.b) + " : " + v(o.c) + ")";
                break;
            case SDQH_X_STR: case SDQH_X_STRIDX: case SDQH_X_CHAR: {
                if (sres_of[(size_t)k] >= 0) {                             // computed by the drain from the field staged in LDS
                    e = o.code == SDQH_X_STR ? "(sres[" + std::to_string(sres_of[(size_t)k]) + "] != 0)" : "sres[" + std::to_string(sres_of[(size_t)k]) + "]";
                    break;
                }
                const std::string c = std::to_string(x.col_of[k]);
                e = text_op(k, "static_cast<const uint32_t*>(a.col[" + c + "]) + r * (int64_t)a.width[" + c + "], a.width[" + c + "]");
                break;
            }
            default: e = "0";
        }
        if (done[(size_t)k]) return;                                          // (emitted meanwhile: a FIELD whose LOOKUP, emitted for it just now, read its wanted fields at once)
        const bool pin = pinning && mode == 0 && is_row_load(k);
        os << "        " << (pin ? "" : "const ") << ctype(o.type) << " v" << K << " = " << e << ";\n";
        if (pin) pins.push_back("v" + K);
        done[(size_t)k] = 1;
        if (pinning && mode == 0 && o.code == SDQH_X_LOOKUP && !want.empty()) {
            flush_pins();
            for (int j = k + 1; j < x.p->nops; ++j) {
                const sdqh_xop& f = x.p->ops[j];
                if ((f.code == SDQH_X_FIELD || f.code == SDQH_X_ACC) && f.a == k && want[(size_t)j]) emit(j);
            }
            flush_pins();
        }
    }
    void reset() { std::fill(done.begin(), done.end(), 0); pins.clear(); }
};

// A TIGHT program: struct P for x_tight (sdqh_xkernels.hpp).  Regs = the packed words of a lane's 8 rows per streamed
// column; eval(i) = the whole program on row i, comparisons of coded columns in code space, coded values through the
// dictionary tables in LDS.
std::string generate_tight(const XInfo& x, Sink sink) {
    const sdqh_program* p = x.p;
    Gen g(x);
    std::vector<int> scols;
    for (int c = 0; c < x.ncols; ++c) scols.push_back(c);
    for (size_t i = 0; i < scols.size(); ++i) g.slot_of[(size_t)scols[i]] = (int)i;
    auto bpr = [&](int c) { return x.enc[c] == ENC_C8 ? 1 : x.enc[c] == ENC_C16 ? 2 : x.enc[c] == ENC_N32 ? 4 : 8; };
    std::ostringstream out;
    out << "#include \"sdqh_xkernels.hpp\"\nusing namespace sdqh;\n";
    // the row function first: it decides which operations become tables
    // No early exit on a failed gate: everything here is register arithmetic (nothing a failed gate has to guard), and a branch per
    // gate and row — exec-mask save / restore, a wait in front of each — costs more than the few operations it skips; the sink
    // applies `pass` once.
    g.reset(); g.mode = 3; g.os.str("");
    g.os << "        bool pass = true;\n";
    for (int q = 0; q < p->ngates; ++q) { g.emit(p->gates[q]); g.os << "        pass = pass & v" << p->gates[q] << ";\n"; }
    if (p->key >= 0) { g.emit(p->key); g.os << "        o.key = v" << p->key << "; o.bad = " << g.bad(p->key) << ";\n"; }
    else g.os << "        o.key = 0; o.bad = false;\n";
    for (int v = 0; v < p->nvals; ++v) {
        if (sink == SINK_GROUP_LANE && v == x.ival) {                     // the value AS AN INTEGER: code + the dictionary's first value (XGroupLane adds it beside the row count)
            const int c = x.col_of[p->vals[v]];
            g.os << "        o.val[" << v << "] = (int64_t)xt_u8(s.c" << g.slot_of[(size_t)c] << ", i) + a.dlo[" << c << "];\n";
            continue;
        }
        g.emit(p->vals[v]);
        g.os << "        o.val[" << v << "] = " << (p->ops[p->vals[v]].type == SDQH_T_F64 ? "x_bits(v" + std::to_string(p->vals[v]) + ")" : "v" + std::to_string(p->vals[v])) << ";\n";
    }
    g.os << "        o.ent = NO_ROW;\n        return pass;\n";
    const std::string row_fn = g.os.str();
    if (sink == SINK_GROUP_LANE && x.ival >= 0) out.str("#define XGL_IVAL " + std::to_string(x.ival) + "\n" + out.str()), out.seekp(0, std::ios::end);
    const std::vector<std::pair<int, int>> tabs = g.tabs;
    out << "struct P {\n    static constexpr int NV = " << p->nvals << ", ND = " << tabs.size() << ";\n    struct Regs {";
    for (size_t i = 0; i < scols.size(); ++i) out << " uint32_t c" << i << "[" << bpr(scols[i]) * 2 << "];";
    out << " };\n";
    out << "    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {\n";
    for (size_t j = 0; j < tabs.size(); ++j) {
        const int k = tabs[j].first, c = tabs[j].second;
        Gen t(x);
        t.mode = 4; t.dict_col = c;
        t.emit(k);
        const int ty = p->ops[k].type;
        out << "        for (int i = threadIdx.x; i < 256; i += TPB) {\n            int64_t cell = 0;\n            if (i < a.ndict[" << c << "]) {\n                const int64_t dv = a.dict[" << c << "][i];\n";
        out << t.os.str();
        out << "                cell = " << (ty == SDQH_T_F64 ? "x_bits(v" + std::to_string(k) + ")" : ty == SDQH_T_BOOL ? "(v" + std::to_string(k) + " ? 1 : 0)" : "v" + std::to_string(k)) << ";\n";
        out << "            }\n            tab[" << j << "][i] = cell;\n        }\n";
    }
    out << "    }\n";
    out << "    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {\n";
    for (size_t i = 0; i < scols.size(); ++i) {
        const int c = scols[i];
        const char* src = x.enc[c] >= ENC_C16 ? "a.code[" : x.enc[c] == ENC_N32 ? "a.ncol[" : "a.col[";
        out << "        xt_load<" << bpr(c) << ", TAIL>(" << src << c << "], r, nrows, s.c" << i << ");\n";
    }
    out << "    }\n";
    out << "    __device__ __forceinline__ static bool eval(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {\n";
    out << row_fn << "    }\n};\n";
    const std::string sn = sink_name(sink);
    out << "extern \"C\" __global__ __launch_bounds__(256) void " << entry_name(sink, true, true) << "(XArgs a, " << sn << "<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {\n";
    out << "    x_tight<P, " << sn << ">(a, s, nrows);\n}\n";
    return out.str();
}



// Can x_vstage8 run this build?  Every operation on registers; lookups only as gates of their own, keyed by one plain integer column
// that fits 32 bits, into tables that answer membership from an exact bitmap over a 32-bit range; at most two payload fields.
void vstage_plan(sdqh_ctx* ctx, int64_t nrows, XInfo* x) {
    const sdqh_program* p = x->p;
    x->vstage = false; x->nlk = 0;
    const bool fake = ctx->compile_only && std::getenv("SDQLPY_AMD_FAKE_CODES") != nullptr;
    if (!ctx->opt_tight || !ctx->opt_narrow || !ctx->opt_vstage || x->ncols < 1 || x->ncols > 12 || p->nvals > 2 ||
        (ctx->compile_only && !fake) || (!fake && nrows < ctx->opt_feature_min_rows)) return;
    for (int c = 0; c < x->ncols; ++c) if (x->cols[c]->dtype == SDQH_STR) return;
    for (int k = 0; k < p->nops; ++k) {
        const sdqh_xop& o = p->ops[k];
        if (op_is_light(o)) continue;
        if (o.code != SDQH_X_LOOKUP || x->nlk == 2) return;
        bool gate = false, used = p->key == k;
        for (int g = 0; g < p->ngates; ++g) gate = gate || p->gates[g] == k;
        for (int j = k + 1; j < p->nops; ++j) used = used || p->ops[j].a == k || p->ops[j].b == k || (p->ops[j].code == SDQH_X_SELECT && p->ops[j].c == k);
        for (int v = 0; v < p->nvals; ++v) used = used || p->vals[v] == k;
        if (!gate || used || p->ops[o.a].code != SDQH_X_COL || p->ops[o.a].type != SDQH_T_I64) return;
        if (!fake) {
            const sdqh_table* t = o.table;
            if (!(t->dev.bm && t->dev.bm_shift == 0 && t->dev.lin_rb == 0 && t->dev.bm_lo >= INT32_MIN && t->dev.bm_hi <= INT32_MAX && t->dev.bm_hi >= t->dev.bm_lo)) return;
        }
        x->lk_op[x->nlk++] = k;
    }
    x->vstage = true;
    tight_plan(ctx, nrows, x);                                             // again, now with every column streamed
    bool ok = x->tight;
    for (int l = 0; l < x->nlk && ok; ++l) ok = x->irange[p->ops[x->lk_op[l]].a] >= 1;      // the lookups' keys fit 32 bits
    if (!ok) { x->vstage = false; tight_plan(ctx, nrows, x); }
}

std::string generate_vstage(const XInfo& x) {
    const sdqh_program* p = x.p;
    Gen g(x);
    std::vector<int> scols;
    for (int c = 0; c < x.ncols; ++c) scols.push_back(c);
    for (size_t i = 0; i < scols.size(); ++i) g.slot_of[(size_t)scols[i]] = (int)i;
    auto bpr = [&](int c) { return x.enc[c] == ENC_C8 ? 1 : x.enc[c] == ENC_C16 ? 2 : x.enc[c] == ENC_N32 ? 4 : 8; };
    auto is_lk = [&](int k) { for (int l = 0; l < x.nlk; ++l) if (x.lk_op[l] == k) return true; return false; };
    g.reset(); g.mode = 3; g.os.str("");
    g.os << "        bool p = true;\n";
    for (int q = 0; q < p->ngates; ++q) { if (is_lk(p->gates[q])) continue; g.emit(p->gates[q]); g.os << "        p = p & v" << p->gates[q] << ";\n"; }
    g.os << "        return p;\n";
    const std::string gates = g.os.str();
    std::string lkoff;
    for (int l = 0; l < x.nlk; ++l) {
        g.reset(); g.os.str("");
        const int kop = p->ops[x.lk_op[l]].a;
        g.emit(kop);
        const std::string t = "a.tab[" + std::to_string(x.tab_of[x.lk_op[l]]) + "]";
        lkoff += "        if (l == " + std::to_string(l) + ") {\n" + g.os.str();
        lkoff += "            const uint32_t o32 = (uint32_t)((int32_t)v" + std::to_string(kop) + " - (int32_t)" + t + ".bm_lo);\n";
        lkoff += "            p = p & (o32 <= (uint32_t)(" + t + ".bm_hi - " + t + ".bm_lo));\n            return o32;\n        }\n";
    }
    lkoff += "        return 0u;\n";
    g.reset(); g.os.str("");
    g.emit(p->key);
    g.os << "        o.key = v" << p->key << "; o.bad = " << g.bad(p->key) << ";\n";
    for (int v = 0; v < p->nvals; ++v) {
        g.emit(p->vals[v]);
        g.os << "        o.val[" << v << "] = " << (p->ops[p->vals[v]].type == SDQH_T_F64 ? "x_bits(v" + std::to_string(p->vals[v]) + ")" : "v" + std::to_string(p->vals[v])) << ";\n";
    }
    g.os << "        o.ent = NO_ROW;\n";
    const std::string row = g.os.str();
    const std::vector<std::pair<int, int>> tabs = g.tabs;
    std::ostringstream out;
    if (const char* e = std::getenv("SDQLPY_AMD_XV_EXP")) out << "#define XV_EXP " << std::atoi(e) << "\n";      // (timing experiments: x_vstage8)
    out << "#include \"sdqh_xkernels.hpp\"\nusing namespace sdqh;\n";
    bool q32 = x.irange[p->key] >= 1;                                     // key and payload fit 32 bits: a queue of 4-byte words
    for (int v = 0; v < p->nvals; ++v) q32 = q32 && p->ops[p->vals[v]].type == SDQH_T_I64 && x.irange[p->vals[v]] >= 1;
    out << "struct P {\n    static constexpr int NV = " << p->nvals << ", ND = " << tabs.size() << ", NL = " << x.nlk << ";\n    static constexpr bool Q32 = " << (q32 ? "true" : "false") << ";\n    struct Regs {";
    for (size_t i = 0; i < scols.size(); ++i) out << " uint32_t c" << i << "[" << (x.enc[scols[i]] == ENC_D8 ? 3 : bpr(scols[i]) * 2) << "];";
    out << " };\n";
    out << "    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {\n";
    for (size_t j = 0; j < tabs.size(); ++j) {
        const int k = tabs[j].first, c = tabs[j].second;
        Gen t(x);
        t.mode = 4; t.dict_col = c;
        t.emit(k);
        const int ty = p->ops[k].type;
        out << "        for (int i = threadIdx.x; i < 256; i += TPB) {\n            int64_t cell = 0;\n            if (i < a.ndict[" << c << "]) {\n                const int64_t dv = a.dict[" << c << "][i];\n";
        out << t.os.str();
        out << "                cell = " << (ty == SDQH_T_F64 ? "x_bits(v" + std::to_string(k) + ")" : ty == SDQH_T_BOOL ? "(v" + std::to_string(k) + " ? 1 : 0)" : "v" + std::to_string(k)) << ";\n";
        out << "            }\n            tab[" << j << "][i] = cell;\n        }\n";
    }
    out << "    }\n";
    out << "    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {\n";
    for (size_t i = 0; i < scols.size(); ++i) {
        const int c = scols[i];
        const char* src = x.enc[c] >= ENC_C16 ? "a.code[" : x.enc[c] == ENC_N32 ? "a.ncol[" : "a.col[";
        if (x.enc[c] == ENC_D8) out << "        xt_load_d8(a.dcol[" << c << "], r, s.c" << i << ");\n";
        else out << "        xt_load<" << bpr(c) << ", TAIL>(" << src << c << "], r, nrows, s.c" << i << ");\n";
    }
    out << "    }\n";
    out << "    __device__ __forceinline__ static bool gates(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r) {\n" << gates << "    }\n";
    out << "    __device__ __forceinline__ static uint32_t lkoff(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, const int l, bool& p) {\n" << lkoff << "    }\n";
    out << "    __device__ __forceinline__ static const uint32_t* lkbm(const XArgs& a, int l) { return ";
    if (x.nlk == 0) out << "nullptr";
    else if (x.nlk == 1) out << "a.tab[" << x.tab_of[x.lk_op[0]] << "].bm";
    else out << "l == 0 ? a.tab[" << x.tab_of[x.lk_op[0]] << "].bm : a.tab[" << x.tab_of[x.lk_op[1]] << "].bm";
    out << "; }\n";
    out << "    __device__ __forceinline__ static void row(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, const int64_t r, XOut<NV>& o) {\n" << row << "    }\n};\n";
    out << "extern \"C\" __global__ __launch_bounds__(256) void " << VSTAGE_ENTRY << "(XArgs a, XStage<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {\n";
    out << "    x_vstage8<P>(a, s, nrows, seg_rows, nseg);\n}\n";
    return out.str();
}

std::string generate(const XInfo& x, Sink sink, bool direct) {
    if (x.vstage && sink == SINK_STAGE) return generate_vstage(x);
    if (x.tight && direct) return generate_tight(x, sink);
    const bool q8 = x.tight && !direct;                                 // the queue skeleton with the tight streamed part (x_queue8)
    const sdqh_program* p = x.p;
    Gen g(x);
    std::vector<int> scols = x.scols;
    int first_gate = x.nstream_gates;
    if (direct) { scols.clear(); for (int c = 0; c < x.ncols; ++c) scols.push_back(c); first_gate = 0; }
    for (size_t i = 0; i < scols.size(); ++i) g.slot_of[(size_t)scols[i]] = (int)i;
    const int ns = (int)scols.size(), nsx = std::max(1, ns);
    std::ostringstream out;
    out << "#include \"sdqh_xkernels.hpp\"\nusing namespace sdqh;\n";
    // text operations: per text column (at most two, fields of at most XSTR_UNITS code units), evaluated by the drain on
    // fields staged in LDS; any other program scans its fields in global memory as before
    std::vector<int> tcols, twidth; std::vector<std::vector<int>> tops;
    bool stage_text = !direct;
    for (int k = 0; k < p->nops && stage_text; ++k) {
        const int code = p->ops[k].code;
        if (code != SDQH_X_STR && code != SDQH_X_STRIDX && code != SDQH_X_CHAR) continue;
        const int c = x.col_of[k], w = p->ops[k].col ? p->ops[k].col->width : 0;
        size_t j = 0;
        while (j < tcols.size() && tcols[j] != c) ++j;
        if (j == tcols.size()) { tcols.push_back(c); twidth.push_back(w); tops.emplace_back(); }
        tops[j].push_back(k);
        if (tcols.size() > 2 || w < 1 || w > XSTR_UNITS) stage_text = false;
    }
    if (!stage_text) { tcols.clear(); twidth.clear(); tops.clear(); }
    int nsop = 0;
    for (auto& ops : tops) for (int k : ops) g.sres_of[(size_t)k] = nsop++;
    const int nsopx = std::max(1, nsop);
    // q8: the streamed conditions first — their emission decides which operations become dictionary tables
    std::string stest8, spre8, spre32;
    std::vector<std::pair<int, int>> tabs8;
    auto bpr = [&](int c) { return x.enc[c] == ENC_C8 ? 1 : x.enc[c] == ENC_C16 ? 2 : x.enc[c] == ENC_N32 ? 4 : 8; };
    if (q8) {
        g.reset(); g.mode = 3; g.os.str("");
        g.os << "        bool p = true;\n";
        for (int q = 0; q < x.nstream_gates; ++q) { g.emit(p->gates[q]); g.os << "        p = p & v" << p->gates[q] << ";\n"; }
        if (x.prefilter_op >= 0) {
            g.emit(x.prefilter_part0);
            g.os << "        p = p && x_may_hit(a.tab[" << x.tab_of[x.prefilter_op] << "], v" << x.prefilter_part0 << ", " << (x.prefilter_composite ? "true" : "false") << ");\n";
        }
        g.os << "        return p;\n";
        stest8 = g.os.str();
        // the same in two halves for the batched form: the cheap conditions and where the key's bit lives (word index, bit), the
        // word loads and the bit tests being the skeleton's
        g.reset(); g.mode = 3; g.os.str("");
        g.os << "        bool p = true;\n";
        for (int q = 0; q < x.nstream_gates; ++q) { g.emit(p->gates[q]); g.os << "        p = p & v" << p->gates[q] << ";\n"; }
        if (x.prefilter_op >= 0) {
            g.emit(x.prefilter_part0);
            const std::string t = "a.tab[" + std::to_string(x.tab_of[x.prefilter_op]) + "]";
            g.os << "        const bool in = (v" << x.prefilter_part0 << " >= " << t << ".bm_lo) & (v" << x.prefilter_part0 << " <= " << t << ".bm_hi);\n";
            g.os << "        p = p & in;\n";
            g.os << "        const uint64_t off = " << (x.pnear ? "in" : "p") << " ? (uint64_t)(v" << x.prefilter_part0 << " - " << t << ".bm_lo) : 0ull;\n";
            g.os << "        widx = (uint32_t)(off >> 5); bit = (uint32_t)off & 31u;\n";
        } else g.os << "        widx = 0; bit = 0;\n";
        g.os << "        return p;\n";
        spre8 = g.os.str();
        if (x.phash) {
            // the hashed filter: 32 hash bits of the (64-bit) key; the skeleton cuts them to the filter's size and tests the key's two bits
            g.reset(); g.mode = 3; g.os.str("");
            g.os << "        bool p = true;\n";
            for (int q = 0; q < x.nstream_gates; ++q) { g.emit(p->gates[q]); g.os << "        p = p & v" << p->gates[q] << ";\n"; }
            g.emit(x.prefilter_part0);
            g.os << "        off = p ? hf_raw((int64_t)v" << x.prefilter_part0 << ") : 0u;\n        return p;\n";
            spre32 = g.os.str();
        }
        if (x.pref32) {
            g.reset(); g.mode = 3; g.os.str("");
            g.os << "        bool p = true;\n";
            for (int q = 0; q < x.nstream_gates; ++q) { g.emit(p->gates[q]); g.os << "        p = p & v" << p->gates[q] << ";\n"; }
            g.emit(x.prefilter_part0);
            const std::string t = "a.tab[" + std::to_string(x.tab_of[x.prefilter_op]) + "]";
            g.os << "        const uint32_t o32 = (uint32_t)((int32_t)v" << x.prefilter_part0 << " - (int32_t)" << t << ".bm_lo);\n";
            // (the offset of a row that fails an earlier condition is still its key's: neighbouring lanes then ask for neighbouring words
            //  whatever the conditions say — with the failing half of Q3's rows sent to word 0 instead, each of the step's 16 requests
            //  straddled two far-apart lines lane by lane and cost three times the address-path cycles of Q5's, where every row passes)
            //  Only where the key column is clustered (pnear): with keys in no order every lane that still asks is one more line, and the
            //  failing rows are better sent to word 0 together — Q5's orders build, 85 % of its rows failing the date: 0.063 -> 0.104 ms.
            g.os << "        const bool in = o32 <= (uint32_t)(" << t << ".bm_hi - " << t << ".bm_lo);\n";
            g.os << "        p = p & in;\n        off = " << (x.pnear ? "in" : "p") << " ? o32 : 0u;\n        return p;\n";
            spre32 = g.os.str();
        }
        tabs8 = g.tabs;
    }
    out << "struct P {\n    static constexpr int NS = " << ns << ", NV = " << p->nvals << ", NSC = " << tcols.size() << ", NSOP = " << nsop << ", ND = " << tabs8.size() << ";\n";
    if (q8) {
        out << "    struct Regs {";
        for (int i = 0; i < ns; ++i) out << " uint32_t c" << i << "[" << (x.enc[scols[(size_t)i]] == ENC_D8 ? 3 : bpr(scols[(size_t)i]) * 2) << "];";
        out << " };\n";
        out << "    __device__ __forceinline__ static void load_dicts(const XArgs& a, int64_t (*tab)[256]) {\n";
        for (size_t j = 0; j < tabs8.size(); ++j) {
            const int k = tabs8[j].first, c = tabs8[j].second;
            Gen t(x);
            t.mode = 4; t.dict_col = c;
            t.emit(k);
            const int ty = p->ops[k].type;
            out << "        for (int i = threadIdx.x; i < 256; i += TPB) {\n            int64_t cell = 0;\n            if (i < a.ndict[" << c << "]) {\n                const int64_t dv = a.dict[" << c << "][i];\n";
            out << t.os.str();
            out << "                cell = " << (ty == SDQH_T_F64 ? "x_bits(v" + std::to_string(k) + ")" : ty == SDQH_T_BOOL ? "(v" + std::to_string(k) + " ? 1 : 0)" : "v" + std::to_string(k)) << ";\n";
            out << "            }\n            tab[" << j << "][i] = cell;\n        }\n";
        }
        out << "    }\n";
        out << "    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Regs& s) {\n";
        for (int i = 0; i < ns; ++i) {
            const int c = scols[(size_t)i];
            const char* src = x.enc[c] >= ENC_C16 ? "a.code[" : x.enc[c] == ENC_N32 ? "a.ncol[" : "a.col[";
            if (x.enc[c] == ENC_D8) out << "        xt_load_d8(a.dcol[" << c << "], r, s.c" << i << ");\n";
            else out << "        xt_load<" << bpr(c) << ", TAIL>(" << src << c << "], r, nrows, s.c" << i << ");\n";
        }
        out << "    }\n";
        out << "    __device__ __forceinline__ static bool stest(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i) {\n" << stest8 << "    }\n";
        out << "    __device__ __forceinline__ static bool spre(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& widx, uint32_t& bit) {\n" << spre8 << "    }\n";
        out << "    static constexpr bool PREF32 = " << ((x.pref32 || x.phash) ? "true" : "false") << ", PWIN = " << (x.pwin ? "true" : "false") << ";\n";
        out << "    __device__ __forceinline__ static bool spre32(const XArgs& a, const Regs& s, const int64_t (*tab)[256], const int i, uint32_t& off) {\n"
            << ((x.pref32 || x.phash) ? spre32 : std::string("        off = 0; return false;\n")) << "    }\n";
        if (x.phash) {
            out << "    static constexpr bool PHASH = true;\n";
            out << "    __device__ __forceinline__ static const TableHeader* shdr(const XArgs& a) { return a.tab[" << x.tab_of[x.prefilter_op] << "].hdr; }\n";
            out << "    __device__ __forceinline__ static const uint32_t* sbitmap(const XArgs& a) { return a.tab[" << x.tab_of[x.prefilter_op] << "].hf; }\n";
        } else
        out << "    __device__ __forceinline__ static const uint32_t* sbitmap(const XArgs& a) { return "
            << (x.prefilter_op >= 0 ? "x_prefilter_bitmap(a.tab[" + std::to_string(x.tab_of[x.prefilter_op]) + "], " + (x.prefilter_composite ? "true" : "false") + ")" : std::string("nullptr")) << "; }\n";
        if (x.driven && sink == SINK_GROUP) {       // (the group sink's launch asks for the tiled walk)
            out << "    static constexpr bool DRIVEN = true;\n";
            out << "    __device__ __forceinline__ static const DevTable& dtab(const XArgs& a) { return a.tab[" << x.tab_of[x.prefilter_op] << "]; }\n";
        }
    }
    if (!tcols.empty()) {
        out << "    __device__ __forceinline__ static constexpr int scol(int j) { return j == 0 ? " << tcols[0] << " : " << (tcols.size() > 1 ? tcols[1] : tcols[0]) << "; }\n";
        out << "    __device__ __forceinline__ static constexpr int swidth(int j) { return j == 0 ? " << twidth[0] << " : " << (twidth.size() > 1 ? twidth[1] : twidth[0]) << "; }\n";
        auto tb = [&](size_t j) { return ((x.narrow_mask >> tcols[j]) & 1u) != 0; };      // staged from the column's byte twin
        out << "    __device__ __forceinline__ static constexpr bool sbytes(int j) { return j == 0 ? " << (tb(0) ? "true" : "false") << " : " << (tb(tcols.size() > 1 ? 1 : 0) ? "true" : "false") << "; }\n";
        // a field of the staging window: `off` is its byte offset (4-byte units: off / 4 words in)
        out << "    template <int J> __device__ __forceinline__ static void sops(const XArgs& a, const uint32_t* region, int off, int64_t (&sres)[" << nsopx << "]) {\n";
        for (size_t j = 0; j < tcols.size(); ++j) {
            out << "        if constexpr (J == " << j << ") {\n";
            if (!tb(j)) out << "            const uint32_t* f = region + (off >> 2);\n";
            for (int k : tops[j]) out << "            sres[" << g.sres_of[(size_t)k] << "] = (int64_t)" << (tb(j) ? g.text_op(k, "region, off, " + std::to_string(twidth[j]), true, true) : g.text_op(k, "f, " + std::to_string(twidth[j]), true)) << ";\n";
            out << "        }\n";
        }
        out << "    }\n";
    }
    if (!q8) {
    out << "    template <bool TAIL> __device__ __forceinline__ static void sload(const XArgs& a, int64_t r, int64_t nrows, Pair<int64_t> (&s)[" << nsx << "]) {\n";
    for (int i = 0; i < ns; ++i) {
        const int c = scols[(size_t)i];
        if (((x.narrow_mask >> c) & 1u) && x.cols[c]->dtype != SDQH_STR) out << "        s[" << i << "] = x_sload_narrow_" << (x.cols[c]->dtype == SDQH_F64 ? "f" : "i") << "<TAIL>(a.ncol[" << c << "], r, nrows);\n";
        else out << "        s[" << i << "] = load2<TAIL>(static_cast<const int64_t*>(a.col[" << c << "]), r, nrows);\n";
    }
    out << "    }\n";
    // streamed conditions, both rows of the pair
    out << "    __device__ __forceinline__ static void stest(const XArgs& a, const Pair<int64_t> (&s)[" << nsx << "], bool& p0, bool& p1) {\n";
    if (!direct) for (int h = 0; h < 2; ++h) {
        g.reset(); g.mode = 2; g.half = h ? "y" : "x"; g.os.str("");
        out << "      {\n";
        for (int q = 0; q < x.nstream_gates; ++q) { g.emit(p->gates[q]); g.os << "        p" << h << " = p" << h << " && v" << p->gates[q] << ";\n"; }
        if (x.prefilter_op >= 0) {
            g.emit(x.prefilter_part0);
            g.os << "        p" << h << " = p" << h << " && x_may_hit(a.tab[" << x.tab_of[x.prefilter_op] << "], v" << x.prefilter_part0 << ", " << (x.prefilter_composite ? "true" : "false") << ");\n";
        }
        out << g.os.str() << "      }\n";
    }
    out << "    }\n";
    }   // !q8
    auto body = [&](int mode, int from_gate) {
        g.reset(); g.mode = mode; g.os.str("");
        // A drained row has passed the streamed prefilter (the looked-up table's key bitmap), so nearly every one of them goes on to its
        // values: the by-row loads of the columns the values read are requested up front, beside the lookup's chain of dependent loads,
        // not behind the branch that ends it (a round trip less per drain)
        static const bool no_levels = std::getenv("SDQLPY_AMD_X_NOLEVELS") != nullptr;
        g.pinning = mode == 0 && !no_levels;
        g.want.assign((size_t)p->nops, 0);
        if (g.pinning) {
            // the columns the GATES read by row (key parts of their lookups, operands of their comparisons): requested together, up front
            for (int q = from_gate; q < p->ngates; ++q) closure(p, p->gates[q], g.want);
            for (int k = 0; k < p->nops; ++k) if (g.want[(size_t)k] && p->ops[k].code == SDQH_X_COL && g.is_row_load(k)) g.emit(k);
            g.flush_pins();
        }
        for (int q = from_gate; q < p->ngates; ++q) { g.emit(p->gates[q]); g.flush_pins(); g.os << "        if (!v" << p->gates[q] << ") return false;\n"; }
        if (g.pinning) {
            // what is left reads the rows that passed every gate (few, in the loops that have lookups): the lookups the values still need,
            // then every column, payload field and accumulator of the key and the values in ONE group
            std::vector<char> rest((size_t)p->nops, 0);
            if (p->key >= 0) closure(p, p->key, rest);
            for (int v = 0; v < p->nvals; ++v) closure(p, p->vals[v], rest);
            if (x.probe_op >= 0) closure(p, x.probe_op, rest);
            g.want = rest;
            for (int k = 0; k < p->nops; ++k) if (rest[(size_t)k] && p->ops[k].code == SDQH_X_LOOKUP) g.emit(k);
            g.flush_pins();
            for (int k = 0; k < p->nops; ++k) if (rest[(size_t)k] && g.is_row_load(k)) g.emit(k);
            g.flush_pins();
        }
        if (p->key >= 0) { g.emit(p->key); g.os << "        o.key = v" << p->key << "; o.bad = " << g.bad(p->key) << ";\n"; }
        else g.os << "        o.key = 0; o.bad = false;\n";
        for (int v = 0; v < p->nvals; ++v) {
            g.emit(p->vals[v]);
            g.os << "        o.val[" << v << "] = " << (p->ops[p->vals[v]].type == SDQH_T_F64 ? "x_bits(v" + std::to_string(p->vals[v]) + ")" : "v" + std::to_string(p->vals[v])) << ";\n";
        }
        if (x.probe_op >= 0) { g.emit(x.probe_op); g.os << "        o.ent = e" << x.probe_op << ";\n"; } else g.os << "        o.ent = NO_ROW;\n";
        g.os << "        return true;\n";
        return g.os.str();
    };
    out << "    template <int H> __device__ __forceinline__ static bool eval_regs(const XArgs& a, const Pair<int64_t> (&s)[" << nsx << "], int64_t r, XOut<NV>& o) {\n";
    if (direct) out << body(1, 0); else out << "        return false;\n";
    out << "    }\n";
    out << "    __device__ __forceinline__ static bool eval_row(const XArgs& a, int64_t r, const int64_t (&sres)[" << nsopx << "], XOut<NV>& o) {\n";
    if (!direct) out << body(0, first_gate); else out << "        return false;\n";
    out << "    }\n};\n";
    const std::string sn = sink_name(sink);
    out << "extern \"C\" __global__ __launch_bounds__(256) void " << entry_name(sink, direct, q8) << "(XArgs a, " << sn << "<P::NV>::Args s, int64_t nrows, int64_t seg_rows, int nseg) {\n";
    if (direct) out << "    x_direct<P, " << sn << ">(a, s, nrows);\n";
    else out << "    " << (q8 ? "x_queue8" : "x_queue") << "<P, " << sn << ", " << (sink == SINK_STAGE ? "true" : "false") << ">(a, s, nrows, seg_rows, nseg);\n";
    out << "}\n";
    return out.str();
}

// ---- specialisation: hiprtc + caches --------------------------------------------------------------------
struct JitState {
    std::mutex mu;
    bool headers_loaded = false; std::string h_abi, h_kernels, h_xkernels, src_dir, cache_dir, why_unusable;
    std::map<std::string, hipFunction_t> kernels;          // hash (+ device) -> function
    int64_t compiled = 0, from_disk = 0;
};
JitState& jit() { static JitState s; return s; }

std::string slurp(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return std::string();
    std::stringstream ss; ss << f.rdbuf();
    return ss.str();
}
uint64_t fnv1a(const std::string& s, uint64_t h = 1469598103934665603ull) {
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}

bool load_headers(JitState& J) {
    if (J.headers_loaded) return J.why_unusable.empty();
    J.headers_loaded = true;
    std::string dir;
    if (const char* env = std::getenv("SDQLPY_AMD_CSRC")) dir = env;
    else {
        Dl_info info;
        if (dladdr(reinterpret_cast<const void*>(&sdqh_jit_stats), &info) && info.dli_fname) { dir = info.dli_fname; const size_t at = dir.rfind('/'); dir = at == std::string::npos ? "." : dir.substr(0, at); }
    }
    J.src_dir = dir;
    J.h_kernels = slurp(dir + "/sdqh_kernels.hpp");
    J.h_xkernels = slurp(dir + "/sdqh_xkernels.hpp");
    J.h_abi = slurp(dir + "/../../include/sdqh.h");
    if (J.h_abi.empty()) J.h_abi = slurp(dir + "/sdqh.h");
    if (J.h_kernels.empty() || J.h_xkernels.empty() || J.h_abi.empty()) {
        J.why_unusable = "run-time specialisation needs sdqh_kernels.hpp, sdqh_xkernels.hpp next to libsdqlhip.so and include/sdqh.h (looked in " + dir + ")";
        return false;
    }
    if (const char* env = std::getenv("SDQLPY_AMD_JIT_CACHE")) J.cache_dir = env;
    else J.cache_dir = dir + "/../jit_cache";
    (void)mkdir(J.cache_dir.c_str(), 0777);
    return true;
}

// The kernel of a program depends on its STRUCTURE only (operations, operand wiring, which operations share a
// column / table / constant slot, string constants, outputs, sink): a cheap hash of that finds the function of
// a plan that has run before without regenerating its source.
uint64_t structure_hash(const XInfo& x, Sink sink, bool direct) {
    const sdqh_program* p = x.p;
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h ^= v; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29; };      // (a word at a time: this runs on every call)
    mix((uint64_t)sink * 2 + (direct ? 1 : 0)); mix((uint64_t)x.narrow_mask);
    if (x.tight) {
        mix(0x7167ull); mix((uint64_t)x.gather32); mix((x.pref32 ? 1ull : 0ull) | (x.vstage ? 2ull : 0ull) | (x.pwin ? 4ull : 0ull) | (x.pnear ? 8ull : 0ull) | (x.driven ? 16ull : 0ull) | (x.phash ? 32ull : 0ull) | ((uint64_t)(x.ival + 1) << 8));
        for (int c = 0; c < x.ncols; ++c) mix(((uint64_t)(uint32_t)x.enc[c] << 32) | ((uint32_t)(x.affine[c] ? 1 : 0) << 16) | (uint32_t)(x.dict_slot[c] & 0xFFFF));
        for (int k = 0; k < x.p->nops; ++k) mix((uint64_t)(uint8_t)x.irange[k]);
        for (int k = 0; k < x.p->nops; ++k) mix(((uint64_t)(uint32_t)x.cmp_cc[k] << 32) | ((uint32_t)x.cmp_kind[k] << 8) | (uint32_t)(x.cmp_col[k] & 0xFF));
    } mix((uint64_t)p->nops); mix((uint64_t)(int64_t)p->key); mix((uint64_t)(int64_t)x.probe_op);
    for (int k = 0; k < p->nops; ++k) {
        const sdqh_xop& o = p->ops[k];
        mix(((uint64_t)(uint32_t)o.code << 32) | (uint32_t)o.type); mix(((uint64_t)(uint32_t)o.a << 32) | (uint32_t)o.b); mix(((uint64_t)(uint32_t)o.c << 32) | (uint32_t)o.aux);
        mix(((uint64_t)(uint32_t)x.col_of[k] << 32) | (uint32_t)x.tab_of[k]); mix(((uint64_t)(uint32_t)x.const_of[k] << 32) | (uint32_t)x.str_off[k]);
        if (o.code == SDQH_X_DIVI || o.code == SDQH_X_MODI) mix((uint64_t)o.imm_i);
        if (o.code == SDQH_X_STR || o.code == SDQH_X_STRIDX) { mix((uint64_t)o.slen); for (int i = 0; i < o.slen; ++i) mix(o.str[i]); }
        if (o.code == SDQH_X_STR || o.code == SDQH_X_STRIDX || o.code == SDQH_X_CHAR) mix((uint64_t)(o.col ? o.col->width : 0));      // the staged field width is a compile-time constant
    }
    mix((uint64_t)p->ngates); for (int g = 0; g < p->ngates; ++g) mix((uint64_t)p->gates[g]);
    mix((uint64_t)p->nvals); for (int v = 0; v < p->nvals; ++v) mix((uint64_t)p->vals[v]);
    return h;
}

int specialise(sdqh_ctx* ctx, const std::string& source, const std::string& entry, hipFunction_t* fn);

int kernel_for(sdqh_ctx* ctx, const XInfo& x, Sink sink, bool direct, hipFunction_t* fn) {
    JitState& J = jit();
    // the tables' layouts are part of the kernel (x_lookup_l): their indexes are made now (fill_xargs would, a moment later)
    static const bool no_layouts = std::getenv("SDQLPY_AMD_X_NOLAYOUT") != nullptr;
    uint64_t lay_hash = 0;
    for (int t = 0; t < x.ntabs; ++t) {
        if (ctx->compile_only) { x.lay[t] = 0u; continue; }                 // (no device: nothing is built, the run-time form of the lookups is what gets compiled)
        if (int rc = index_ensure(ctx, x.tabs[t])) return rc;
        x.lay[t] = no_layouts ? 0u : x_layout_of(x.tabs[t]->dev, x.tabs[t]->bitmap_only);
        lay_hash = (lay_hash ^ x.lay[t]) * 0x9E3779B97F4A7C15ull + (uint64_t)t;
    }
    char name[64];
    std::snprintf(name, sizeof(name), "s%016llx@%d", (unsigned long long)(structure_hash(x, sink, direct) ^ lay_hash), ctx->device);
    if (!ctx->compile_only) {
        std::lock_guard<std::mutex> lock(J.mu);
        auto hit = J.kernels.find(name);
        if (hit != J.kernels.end()) { *fn = hit->second; return SDQH_OK; }
    }
    if (int rc = specialise(ctx, generate(x, sink, direct), (x.vstage && sink == SINK_STAGE) ? std::string(VSTAGE_ENTRY) : entry_name(sink, direct, x.tight), fn)) return rc;
    std::lock_guard<std::mutex> lock(J.mu);
    J.kernels[name] = *fn;
    return SDQH_OK;
}

int specialise(sdqh_ctx* ctx, const std::string& source_in, const std::string& entry, hipFunction_t* fn) {
    // SDQLPY_AMD_X_DEFINES="NAME=VALUE NAME=VALUE": macros put in front of every specialised source (A/B switches of the skeletons,
    // e.g. X8_PIPE=1; part of the source, so of the cache key)
    std::string source = source_in;
    if (const char* defs = std::getenv("SDQLPY_AMD_X_DEFINES")) {
        std::istringstream in(defs); std::string tok, head;
        while (in >> tok) { const size_t eq = tok.find('='); head += "#define " + (eq == std::string::npos ? tok + " 1" : tok.substr(0, eq) + " " + tok.substr(eq + 1)) + "\n"; }
        source = head + source;
    }
    JitState& J = jit();
    std::lock_guard<std::mutex> lock(J.mu);
    if (!load_headers(J)) return fail(ctx, SDQH_ERR_DEVICE, J.why_unusable);
    int rtc_major = 0, rtc_minor = 0;
    (void)hiprtcVersion(&rtc_major, &rtc_minor);
    uint64_t h = fnv1a(source);
    h = fnv1a(J.h_xkernels, h); h = fnv1a(J.h_kernels, h); h = fnv1a(J.h_abi, h);
    h = fnv1a("gfx950 rtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor), h);
    char name[64];
    std::snprintf(name, sizeof(name), "%016llx", (unsigned long long)h);
    const std::string key = std::string(name) + "@" + std::to_string(ctx->device);
    auto hit = J.kernels.find(key);
    if (hit != J.kernels.end()) { *fn = hit->second; return SDQH_OK; }
    if (const char* dump = std::getenv("SDQLPY_AMD_JIT_DUMP")) if (dump[0] == '2') std::fprintf(stderr, "---- specialised source (%s) ----\n%s\n", entry.c_str(), source.c_str());
    const std::string path = J.cache_dir + "/" + name + ".hsaco";
    std::string code = slurp(path);
    if (!code.empty()) ++J.from_disk;
    else {
        hiprtcProgram prog;
        const char* hs[3] = {J.h_abi.c_str(), J.h_kernels.c_str(), J.h_xkernels.c_str()};
        const char* hn[3] = {"sdqh.h", "sdqh_kernels.hpp", "sdqh_xkernels.hpp"};
        if (hiprtcCreateProgram(&prog, source.c_str(), "sdqh_specialised.hip", 3, hs, hn) != HIPRTC_SUCCESS)
            return fail(ctx, SDQH_ERR_DEVICE, "hiprtcCreateProgram failed");
        const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics"};
        const hiprtcResult r = hiprtcCompileProgram(prog, 5, opts);
        if (r != HIPRTC_SUCCESS) {
            size_t n = 0; (void)hiprtcGetProgramLogSize(prog, &n);
            std::string log(n + 1, '\0');
            if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
            (void)hiprtcDestroyProgram(&prog);
            if (std::getenv("SDQLPY_AMD_JIT_DUMP")) std::fprintf(stderr, "---- specialised source ----\n%s\n", source.c_str());
            return fail(ctx, SDQH_ERR_DEVICE, std::string("hiprtc: ") + hiprtcGetErrorString(r) + "\n" + log.substr(0, 4000));
        }
        size_t n = 0; (void)hiprtcGetCodeSize(prog, &n);
        code.resize(n);
        (void)hiprtcGetCode(prog, &code[0]);
        (void)hiprtcDestroyProgram(&prog);
        ++J.compiled;
        // SDQLPY_AMD_JIT_RECIPES=<dir>: keep the generated source of every kernel that had to be compiled (named by a hash of the source
        // alone).  Committed under sdqlpy_amd/jit_recipes/, build() compiles them ahead of time — hipcc-style cross compilation, no GPU
        // needed — so a fresh box finds its kernels in the cache instead of paying 0.3 - 0.6 s of hiprtc per kernel on the first run.
        if (const char* rdir = std::getenv("SDQLPY_AMD_JIT_RECIPES")) {
            char rn[32]; std::snprintf(rn, sizeof(rn), "%016llx", (unsigned long long)fnv1a(source));
            (void)mkdir(rdir, 0777);
            std::ofstream rf(std::string(rdir) + "/" + rn + ".hip", std::ios::binary);
            rf << source;
        }
        const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
        { std::ofstream f(tmp, std::ios::binary); f.write(code.data(), (std::streamsize)code.size()); }
        (void)std::rename(tmp.c_str(), path.c_str());                  // atomic: a peer rank never sees half a file
    }
    if (ctx->compile_only) return fail(ctx, SDQH_ERR_DEVICE, "compile-only context: kernel specialised (" + std::to_string(code.size()) + " bytes of gfx950 code), not run");
    hipModule_t mod;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, SDQH_ERR_DEVICE, "hipModuleLoadData failed for a specialised kernel"); }
    hipFunction_t f;
    if (hipModuleGetFunction(&f, mod, entry.c_str()) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, SDQH_ERR_DEVICE, "specialised kernel has no entry point"); }
    J.kernels[key] = f;
    *fn = f;
    return SDQH_OK;
}

// ---- launch --------------------------------------------------------------------------------------------
template <class SA> struct Packed { XArgs a; SA s; int64_t nrows; int64_t seg_rows; int32_t nseg; };

int fill_xargs(sdqh_ctx* ctx, const XInfo& x, XArgs* a, int32_t* flags, int64_t key_lo, int64_t key_hi) {
    std::memset(a, 0, sizeof(*a));
    for (int c = 0; c < x.ncols; ++c) { a->col[c] = x.cols[c]->data; a->width[c] = x.cols[c]->width; a->ncol[c] = (((x.narrow_mask | x.gather32) >> c) & 1u) ? x.cols[c]->narrow : nullptr; }
    for (int t = 0; t < x.ntabs; ++t) {
        if (int rc = index_ensure(ctx, x.tabs[t])) return rc;
        a->tab[t] = x.tabs[t]->dev;
    }
    if (x.tight) for (int c = 0; c < x.ncols; ++c) a->dlo[c] = x.dlo[c];
    if (x.tight) for (int c = 0; c < x.ncols; ++c) if (x.enc[c] == ENC_D8) a->dcol[c] = x.cols[c]->delta8;
    if (x.tight) for (int c = 0; c < x.ncols; ++c) if (x.enc[c] >= ENC_C16) { a->code[c] = x.cols[c]->code; a->dict[c] = static_cast<const int64_t*>(x.cols[c]->dict); a->ndict[c] = x.cols[c]->ndict; }
    std::memcpy(a->cc, x.cc, sizeof(x.cc[0]) * (size_t)x.ncc);
    std::memcpy(a->ci, x.ci, sizeof(x.ci[0]) * (size_t)x.nci);
    std::memcpy(a->cf, x.cf, sizeof(x.cf[0]) * (size_t)x.ncf);
    std::memcpy(a->spool, x.spool, sizeof(uint32_t) * (size_t)x.nstr);
    a->flags = flags; a->key_lo = key_lo; a->key_hi = key_hi;
    if (x.driven && x.driven_col >= 0 && x.cols[x.driven_col]->run_index_state == 1) {
        const sdqh_column* kc = x.cols[x.driven_col];
        a->run_index = static_cast<const uint32_t*>(kc->run_index); a->run_lo = kc->mn; a->run_hi = kc->mx; a->driven_ratio = ctx->opt_x_driven;
    }
    return SDQH_OK;
}

struct Geometry { unsigned grid; int64_t seg_rows; int nseg; };
Geometry geometry_tight(sdqh_ctx* ctx, int64_t nrows, int resident, bool pipelined = false) {
    const int64_t steps = std::max<int64_t>(1, nrows / ((int64_t)XT_ROWS * (pipelined ? 1 : 2)));
    return Geometry{(unsigned)std::min<int64_t>(steps, (int64_t)ctx->num_cu * resident), 0, 0};
}
// tiled: the sink does not care in which order the rows reach it (sums into groups / entries, key bits) — the tight queue skeleton then
// walks the rows in interleaved 1024-row double steps (x_queue8: seg_rows == 0) instead of one contiguous segment per wave
Geometry geometry(sdqh_ctx* ctx, int64_t nrows, bool direct, int waves_per_cu, bool tight = false, bool tiled = false) {
    Geometry g{1, 0, 0};
    static const bool no_tiles = getenv("SDQLPY_AMD_X_NOTILED") != nullptr;
    if (tiled && tight && !direct && !no_tiles && nrows < ((int64_t)1 << 31) && nrows >= (int64_t)X8_STEP * X8_U) {
        if (ctx->opt_x_waves > 0) waves_per_cu = ctx->opt_x_waves;
        const int64_t steps = nrows / ((int64_t)X8_STEP * X8_U);
        g.seg_rows = 0;
        g.nseg = (int)std::max<int64_t>(1, std::min<int64_t>(steps, (int64_t)ctx->num_cu * waves_per_cu));
        g.grid = (unsigned)((g.nseg + TPB / WAVE - 1) / (TPB / WAVE));
        return g;
    }
    if (direct) {
        const int64_t tiles = ((nrows + TILE_ROWS - 1) / TILE_ROWS + SDQH_TILE_CHUNK - 1) / SDQH_TILE_CHUNK;
        g.grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(tiles, (int64_t)ctx->num_cu * ctx->opt_resident_stream));
        return g;
    }
    const int64_t gran = tight ? (int64_t)X8_STEP * X8_U : (int64_t)WAVE * ROWS_PER_LOAD * X_LB;      // a wave's segment is whole steps of its skeleton
    if (ctx->opt_x_waves > 0) waves_per_cu = ctx->opt_x_waves;                                        // (tuning: segments per CU of the queue skeletons)
    const int64_t target = (int64_t)ctx->num_cu * waves_per_cu;
    int64_t seg_rows = (nrows + target - 1) / target;
    seg_rows = std::max<int64_t>(gran, (seg_rows + gran - 1) / gran * gran);
    g.seg_rows = seg_rows;
    g.nseg = (int)std::max<int64_t>(1, (nrows + seg_rows - 1) / seg_rows);
    g.grid = (unsigned)((g.nseg + TPB / WAVE - 1) / (TPB / WAVE));
    return g;
}

template <class SA>
int launch(sdqh_ctx* ctx, hipFunction_t fn, const char* name, const XArgs& a, const SA& sa, int64_t nrows, const Geometry& g, unsigned lds_bytes = 0) {
    Packed<SA> pk;
    std::memset(&pk, 0, sizeof(pk));
    pk.a = a; pk.s = sa; pk.nrows = nrows; pk.seg_rows = g.seg_rows; pk.nseg = g.nseg;
    size_t size = sizeof(pk);
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &pk, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    KernelScope ks(ctx, name);
    HIP_TRYX(ctx, hipModuleLaunchKernel(fn, g.grid, 1, 1, TPB, 1, 1, lds_bytes, ctx->stream, nullptr, config));
    return SDQH_OK;
}

// HBM bytes a launch of this program streams BY CONSTRUCTION (sdqh_profile_entry_bytes): every streamed column at the encoding this
// call chose x rows, and the key bitmaps its streamed membership tests read (once: they stay in L2).  Gathers by row and
// survivor-dependent stores are not modelled; nor is the tile every workgroup of the pipelined tight skeleton requests a second
// time after its last (an L2 hit: it never reaches the memory side).
int64_t model_stream_bytes(const XInfo& x, int64_t nrows, bool regs_all) {
    int64_t per_row = 0, per_8rows = 0;
    for (int c = 0; c < x.ncols; ++c) {
        bool streamed = regs_all;
        if (!streamed) for (int sc : x.scols) streamed = streamed || sc == c;
        if (!streamed || x.cols[c]->dtype == SDQH_STR) continue;
        if (x.tight && x.enc[c] == ENC_D8) { per_8rows += 12; continue; }
        if (x.tight) per_row += x.enc[c] == ENC_C8 ? 1 : x.enc[c] == ENC_C16 ? 2 : x.enc[c] == ENC_N32 ? 4 : 8;
        else per_row += ((x.narrow_mask >> c) & 1u) ? 4 : 8;
    }
    int64_t bytes = per_row * nrows + per_8rows * ((nrows + 7) / 8);
    auto bitmap = [&](const sdqh_table* t) { if (t && t->dev.bm && t->dev.bm_hi >= t->dev.bm_lo) bytes += (int64_t)(((uint64_t)(t->dev.bm_hi - t->dev.bm_lo) >> t->dev.bm_shift) / 8 + 4); };
    if (x.vstage) for (int l = 0; l < x.nlk; ++l) bitmap(x.tabs[x.tab_of[x.lk_op[l]]]);
    else if (!regs_all && x.prefilter_op >= 0) bitmap(x.tabs[x.tab_of[x.prefilter_op]]);
    return bytes;
}

int read_flags(sdqh_ctx* ctx, const int32_t* d_flags, int* out) {
    HIP_TRYX(ctx, hipMemcpyAsync(ctx->result_host, d_flags, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (int rc = sync_stream(ctx)) return rc;
    *out = *static_cast<const int*>(ctx->result_host);
    return SDQH_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int sdqh_jit_stats(sdqh_ctx* ctx, int64_t* compiled, int64_t* from_cache) {
    if (!ctx) return SDQH_ERR_INVALID;
    JitState& J = jit();
    std::lock_guard<std::mutex> lock(J.mu);
    if (compiled) *compiled = J.compiled;
    if (from_cache) *from_cache = J.from_disk;
    return SDQH_OK;
}

int sdqh_jit_compile(sdqh_ctx* ctx, const char* source) {
    if (!ctx || !source) return fail(ctx, SDQH_ERR_INVALID, "jit_compile: bad arguments");
    hipFunction_t fn;
    const bool was = ctx->compile_only;
    ctx->compile_only = true;                                  // compile into the cache, load nothing
    const int rc = specialise(ctx, source, "", &fn);
    ctx->compile_only = was;
    if (rc == SDQH_ERR_DEVICE && ctx->err.find("kernel specialised") != std::string::npos) return SDQH_OK;
    return rc;
}

int sdqh_xscan_sum(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, double* out_values, int64_t* out_count) {
    if (!ctx || nrows < 0) return fail(ctx, SDQH_ERR_INVALID, "xscan_sum: bad arguments");
    if (!ctx->compile_only) (void)hipSetDevice(ctx->device);
    XInfo x;
    if (int rc = analyse(ctx, nrows, prog, SDQH_TUPLE_MAX_VALUES, false, true, &x)) return rc;
    hipFunction_t fn;
    if (int rc = kernel_for(ctx, x, SINK_SUM, x.direct, &fn)) return rc;
    call_begin(ctx);
    XArgs a;
    rd_dirty(ctx);
    int32_t* d_flags = reinterpret_cast<int32_t*>(static_cast<char*>(ctx->result_dev) + 1024);
    if (int rc = fill_xargs(ctx, x, &a, d_flags, 1, 0)) return rc;
    const Geometry g = (x.tight && x.direct) ? geometry_tight(ctx, nrows, 2) : geometry(ctx, nrows, x.direct, 16, x.tight, true);
    double* partial = static_cast<double*>(pool_alloc(ctx, (size_t)g.grid * 5 * sizeof(double)));
    if (!partial) return fail(ctx, SDQH_ERR_NOMEM, "xscan_sum: out of device memory");
    XSum<1>::Args sa{partial};
    ctx->next_model_bytes = model_stream_bytes(x, nrows, x.direct);
    int rc = launch(ctx, fn, launch_label(SINK_SUM, x.direct, x.tight), a, sa, nrows, g);
    if (!rc) {
        launch_sum_partials(ctx, partial, (int)g.grid, static_cast<double*>(ctx->result_host));      // the fold writes the pinned host block: no copy-engine launch
        call_end(ctx);
        rc = sync_stream(ctx);
    }
    pool_free(ctx, partial);
    if (rc) return rc;
    const double* h = static_cast<const double*>(ctx->result_host);
    if (out_values) for (int k = 0; k < prog->nvals; ++k) out_values[k] = h[k];
    if (out_count) *out_count = reinterpret_cast<const int64_t*>(h)[4];
    return SDQH_OK;
}

// The completion word of an xgroupby_async block: on a cache line the merge kernel never touches (it stores the two tail words right
// behind the counts — flags and a second word that only happens to be zero today; a kernel store there must never read as completion)
constexpr size_t XGROUPBY_DONE_OFFSET = (((size_t)LG_SLOTS * 48 + 8 + 63) & ~(size_t)63);
// K-C small, launched: the group table of the whole call lands in `host_block` (device-visible host memory laid out like the device
// result block: keys[LG_SLOTS] | acc[LG_SLOTS][4] | cnt[LG_SLOTS] | flags) when the stream gets there; nothing is waited for.
// device_block: `host_block` is DEVICE memory (sdqh_xgroupby_partial: one rank's group table on its way into a collective) — same layout,
// no completion word (whoever folds the ranks' blocks writes the result block's).
static int xgroupby_launch(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, void* host_block, bool device_block = false) {
    XInfo x;
    x.want_driven = true;
    if (int rc = analyse(ctx, nrows, prog, SDQH_TUPLE_MAX_VALUES, true, true, &x)) return rc;
    // a key over a dense small range (known from the columns' own ranges): every lane keeps its groups' sums in LDS cells of its own
    Sink sink = SINK_GROUP;
    int nslots = 0;
    int64_t klo = 1, khi = 0;
    if (x.tight && x.direct) {
        int64_t lo = 0, hi = -1;
        const int na = prog->nvals + 1;
        if (ctx->compile_only) { sink = SINK_GROUP_LANE; nslots = 1; klo = khi = 0; }          // (build check: compile the lane sink)
        else if (op_interval(ctx, x, prog->key, &lo, &hi) && lo >= 0 && hi >= lo && hi - lo < 32 && (hi - lo + 1) * na <= 32) { sink = SINK_GROUP_LANE; nslots = (int)(hi - lo + 1); klo = lo; khi = hi; }
    }
    // (the per-lane sink's grid is the same with and without the encodings: partial sums are folded in workgroup order, and switching
    //  the twins off must not change a bit of a sum — tests/test_hip_parity.py)
    const Geometry g = (x.tight && x.direct) ? geometry_tight(ctx, nrows, sink == SINK_GROUP_LANE ? ctx->opt_lane_resident : 2, sink == SINK_GROUP_LANE) : geometry(ctx, nrows, x.direct, 16, x.tight, true);
    if (sink == SINK_GROUP_LANE && !ctx->compile_only && ctx->opt_lane_int) {
        // a summed value that is a small integer on every row — the value of a byte-coded column whose dictionary is consecutive
        // non-negative integers (l_quantity: 1.0 ... 50.0) — is summed as an integer beside the row count (XGroupLane); its lane sums
        // must stay below 2^31: largest value x the rows a lane can meet
        const int64_t lane_rows = ((nrows / XT_ROWS + g.grid) / std::max(1u, g.grid) + 2) * XT_R;
        for (int v = 0; v < prog->nvals && x.ival < 0; ++v) {
            const sdqh_xop& o = prog->ops[prog->vals[v]];
            if (o.code != SDQH_X_COL || o.type != SDQH_T_F64) continue;
            const int c = x.col_of[prog->vals[v]];
            const sdqh_column* col = x.cols[c];
            if (x.enc[c] != ENC_C8 || col->dtype != SDQH_F64 || col->dict_host.empty()) continue;
            bool ok = true; int64_t base = 0;
            for (size_t i = 0; i < col->dict_host.size() && ok; ++i) {
                double d; std::memcpy(&d, &col->dict_host[i], 8);
                const int64_t iv = (int64_t)d;
                ok = (double)iv == d && iv >= 0 && iv <= (1 << 20) && !(iv == 0 && std::signbit(d));
                if (i == 0) base = iv; else ok = ok && iv == base + (int64_t)i;
            }
            if (ok && (base + (int64_t)col->dict_host.size()) * lane_rows < ((int64_t)1 << 31)) { x.ival = v; x.dlo[c] = base; }
        }
    }
    hipFunction_t fn;
    if (int rc = kernel_for(ctx, x, sink, x.direct, &fn)) return rc;
    call_begin(ctx);
    // result block in ctx->result_dev: gkeys[LG_SLOTS] | acc[LG_SLOTS][4] | cnt[LG_SLOTS] | flags
    char* rd = static_cast<char*>(ctx->result_dev);
    unsigned long long* r_keys = reinterpret_cast<unsigned long long*>(rd);
    int* r_flags = reinterpret_cast<int*>(rd + LG_SLOTS * 48);
    static_assert(LG_SLOTS * 48 + 8 <= RESULT_BYTES, "result block too small");
    XArgs a;
    if (int rc = fill_xargs(ctx, x, &a, r_flags, klo, khi)) return rc;
    const size_t npart = (size_t)g.grid * LG_SLOTS;
    char* blob = static_cast<char*>(pool_alloc(ctx, npart * 40 + 256));
    if (!blob) return fail(ctx, SDQH_ERR_NOMEM, "xgroupby: out of device memory");
    double* pacc = reinterpret_cast<double*>(blob);
    int64_t* pcnt = reinterpret_cast<int64_t*>(blob + npart * 32);
    if (!rd_take_clean_lg(ctx)) { void* ptr[2] = {r_keys, r_flags}; size_t bytes[2] = {LG_SLOTS * 8, 8}; unsigned char byte[2] = {0xFF, 0}; fill_regions(ctx, ptr, bytes, byte, 2); }
    int rc;
    if (sink == SINK_GROUP_LANE) {
        XGroupLane<1>::Args sa{r_keys, pacc, pcnt, r_flags, nslots, 0};
        ctx->next_model_bytes = model_stream_bytes(x, nrows, true) + (int64_t)nslots * g.grid * 40;
        rc = launch(ctx, fn, launch_label(SINK_GROUP_LANE, true, true), a, sa, nrows, g, (unsigned)((size_t)nslots * (size_t)(prog->nvals + 1 - (x.ival >= 0 ? 1 : 0)) * TPB * 8));
    } else {
        XGroup<1>::Args sa{r_keys, pacc, pcnt, r_flags};
        // (a loop that may WALK — decided by its waves at run time from the looked-up table's own bitmap — streams nothing when it does, and what it
        //  gathers instead only the data says: no model, rather than the stream's bytes quoted for a kernel that does not stream)
        ctx->next_model_bytes = x.driven ? 0 : model_stream_bytes(x, nrows, x.direct) + (int64_t)npart * 40;
        rc = launch(ctx, fn, launch_label(SINK_GROUP, x.direct, x.tight), a, sa, nrows, g);
    }
    if (!rc) {
        // the block's DONE word: written by the stream itself once the merge has finished — collect waits for its own result, not for
        // whatever was queued behind it.  Cleared before the merge is launched.
        uint32_t* done = (host_block && !device_block) ? reinterpret_cast<uint32_t*>(static_cast<char*>(host_block) + XGROUPBY_DONE_OFFSET) : nullptr;
        if (done) host_init(ctx, done, 0, 4);
        launch_groupby_merge_lg_host(ctx, r_keys, pacc, pcnt, (int)g.grid, r_flags, host_block);       // writes the pinned host block, leaves the device block clean
        call_end(ctx);
        if (done && stream_store32(ctx, ctx->stream, done, 1) != SDQH_OK) *done = 2;     // 2: no marker, collect synchronises
    }
    pool_free(ctx, blob);                                   // stream order: whoever gets the block next runs after the merge
    return rc;
}

// wait until a word of device-visible host memory that the stream writes (hipStreamWriteValue32) holds `value`
static int wait_word(sdqh_ctx* ctx, const volatile uint32_t* word, uint32_t value) {
    if (ctx->capturing) return fail(ctx, SDQH_ERR_UNSUPPORTED, "a call that waits for the device cannot be recorded into a plan graph");
    // bounded by wall time (2 s), not by a spin count; then the runtime's own wait on the stream — it reports a faulted kernel at once,
    // where re-entering the library's spinning synchronise would stall for its bound a second time
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 1;; ++spins) {
        if (*word == value) return SDQH_OK;
        __builtin_ia32_pause();
        if ((spins & 4095u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) break;
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); return fail(ctx, SDQH_ERR_DEVICE, "the device failed while a result was waited for"); }
    if (*word == value) return SDQH_OK;
    if (int rc = sdqh_synchronize(ctx)) return rc;                             // (K-F's word is written by the copy stream: wait for the copies too)
    return *word == value ? SDQH_OK : fail(ctx, SDQH_ERR_DEVICE, "a result's completion word was never written");
}

// The groups of a finished call, out of its host block: ascending keys, at most max_groups.
static int xgroupby_collect(sdqh_ctx* ctx, const void* host_block, int nvals, int max_groups, int64_t* out_keys, double* out_values, int64_t* out_counts, int32_t* out_ngroups) {
    const char* h = static_cast<const char*>(host_block);
    const int flags = *reinterpret_cast<const int*>(h + LG_SLOTS * 48);
    if (flags & 2) return fail(ctx, SDQH_ERR_UNSUPPORTED, "xgroupby: negative group key (or a key outside the range its columns span)");
    const unsigned long long* hk = reinterpret_cast<const unsigned long long*>(h);
    const double* ha = reinterpret_cast<const double*>(h + LG_SLOTS * 8);
    const int64_t* hc = reinterpret_cast<const int64_t*>(h + LG_SLOTS * 40);
    std::vector<int> order;
    for (int s = 0; s < LG_SLOTS; ++s) if (hk[s] != EMPTY_GROUP && hc[s] > 0) order.push_back(s);
    std::sort(order.begin(), order.end(), [&](int p, int q) { return hk[p] < hk[q]; });
    const int ng = (int)order.size();
    if ((flags & 1) || ng > max_groups) { *out_ngroups = std::max(ng, max_groups + 1); return fail(ctx, SDQH_ERR_OVERFLOW, "xgroupby: more groups than max_groups"); }
    for (int i = 0; i < ng; ++i) {
        const int s = order[(size_t)i];
        if (out_keys) out_keys[i] = (int64_t)hk[s];
        if (out_values) for (int k = 0; k < SDQH_TUPLE_MAX_VALUES; ++k) out_values[i * SDQH_TUPLE_MAX_VALUES + k] = k < nvals ? ha[s * 4 + k] : 0.0;
        if (out_counts) out_counts[i] = hc[s];
    }
    *out_ngroups = ng;
    return SDQH_OK;
}

int sdqh_xgroupby(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, int max_groups,
                  int64_t* out_keys, double* out_values, int64_t* out_counts, int32_t* out_ngroups) {
    if (!ctx || nrows < 0 || max_groups < 1 || max_groups > SDQH_MAX_LOOKUP_GROUPS || !out_ngroups) return fail(ctx, SDQH_ERR_INVALID, "xgroupby: bad arguments");
    if (!ctx->compile_only) (void)hipSetDevice(ctx->device);
    if (int rc = xgroupby_launch(ctx, nrows, prog, nullptr)) return rc;
    if (int rc = sync_stream(ctx)) return rc;
    return xgroupby_collect(ctx, ctx->result_host, prog->nvals, max_groups, out_keys, out_values, out_counts, out_ngroups);
}

size_t sdqh_xgroupby_block_bytes(void) { return XGROUPBY_DONE_OFFSET + 64; }

int sdqh_xgroupby_async(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, void* result_block) {
    if (!ctx || nrows < 0 || !prog || !result_block) return fail(ctx, SDQH_ERR_INVALID, "xgroupby_async: bad arguments");
    if (ctx->compile_only) return fail(ctx, SDQH_ERR_UNSUPPORTED, "xgroupby_async: compile-only context");
    if (!host_block_contains(ctx, result_block, sdqh_xgroupby_block_bytes())) return fail(ctx, SDQH_ERR_INVALID, "xgroupby_async: the result block must come from sdqh_host_alloc (sdqh_xgroupby_block_bytes() bytes)");
    (void)hipSetDevice(ctx->device);
    return xgroupby_launch(ctx, nrows, prog, result_block);
}

int sdqh_xgroupby_collect(sdqh_ctx* ctx, const void* result_block, int nvals, int max_groups,
                          int64_t* out_keys, double* out_values, int64_t* out_counts, int32_t* out_ngroups) {
    if (!ctx || !result_block || nvals < 0 || nvals > SDQH_TUPLE_MAX_VALUES || max_groups < 1 || max_groups > SDQH_MAX_LOOKUP_GROUPS || !out_ngroups) return fail(ctx, SDQH_ERR_INVALID, "xgroupby_collect: bad arguments");
    const volatile uint32_t* done = reinterpret_cast<const volatile uint32_t*>(static_cast<const char*>(result_block) + XGROUPBY_DONE_OFFSET);
    if (*done == 2) { if (int rc = sdqh_synchronize(ctx)) return rc; }
    else if (int rc = wait_word(ctx, done, 1)) return rc;
    return xgroupby_collect(ctx, result_block, nvals, max_groups, out_keys, out_values, out_counts, out_ngroups);
}

// ---- partial groups of several ranks folded on the device (ABI 6; no reference counterpart: SURVEY.md 8e) --------------------------------
// One workgroup: the group tables of `nblocks` ranks (each laid out like the result block: keys | sums | counts | flags, `stride` bytes
// apart) folded into `out` BLOCK BY BLOCK — a group's sums are ((rank 0 + rank 1) + rank 2) ..., the order the host-side merge of the
// distributed runner uses, so a result does not depend on which of the two folded it.  Inside one block every key sits in one slot, so
// the thread that owns the slot is the only one that adds to its group during that block's turn: plain LDS read-modify-write, no atomics
// but the claim of a slot.  More than LG_SLOTS / 2 distinct keys over all ranks: flag 1 (collect: SDQH_ERR_OVERFLOW), as for one rank.
__global__ __launch_bounds__(TPB) void k_groups_fold(const char* __restrict__ blocks, int nblocks, size_t stride, char* __restrict__ out) {
    __shared__ unsigned long long s_keys[LG_SLOTS];
    __shared__ double s_acc[LG_SLOTS * 4];
    __shared__ long long s_cnt[LG_SLOTS];
    __shared__ int s_flags, s_groups;
    for (int s = threadIdx.x; s < LG_SLOTS; s += TPB) { s_keys[s] = EMPTY_GROUP; s_cnt[s] = 0; s_acc[s * 4] = s_acc[s * 4 + 1] = s_acc[s * 4 + 2] = s_acc[s * 4 + 3] = 0.0; }
    if (threadIdx.x == 0) { s_flags = 0; s_groups = 0; }
    __syncthreads();
    for (int b = 0; b < nblocks; ++b) {
        const char* blk = blocks + (size_t)b * stride;
        const unsigned long long* bk = reinterpret_cast<const unsigned long long*>(blk);
        const double* ba = reinterpret_cast<const double*>(blk + LG_SLOTS * 8);
        const long long* bc = reinterpret_cast<const long long*>(blk + LG_SLOTS * 40);
        if (threadIdx.x == 0) { const int f = *reinterpret_cast<const int*>(blk + LG_SLOTS * 48); if (f) atomicOr(&s_flags, f); }
        for (int s = threadIdx.x; s < LG_SLOTS; s += TPB) {
            const unsigned long long key = bk[s];
            const long long c = bc[s];
            if (key == EMPTY_GROUP || c <= 0) continue;
            int h = (int)((key * 0x9E3779B97F4A7C15ull) >> 55) & (LG_SLOTS - 1);
            int at = -1;
            for (int probe = 0; probe < LG_SLOTS; ++probe, h = (h + 1) & (LG_SLOTS - 1)) {
                unsigned long long cur = s_keys[h];
                if (cur == EMPTY_GROUP) { cur = atomicCAS(&s_keys[h], (unsigned long long)EMPTY_GROUP, key); if (cur == EMPTY_GROUP) atomicAdd(&s_groups, 1); }
                if (cur == EMPTY_GROUP || cur == key) { at = h; break; }
            }
            if (at < 0) { atomicOr(&s_flags, 1); continue; }
#pragma unroll
            for (int k = 0; k < 4; ++k) s_acc[at * 4 + k] += ba[s * 4 + k];
            s_cnt[at] += c;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_groups > LG_SLOTS / 2) s_flags |= 1;
    __syncthreads();
    unsigned long long* ok = reinterpret_cast<unsigned long long*>(out);
    double* oa = reinterpret_cast<double*>(out + LG_SLOTS * 8);
    long long* oc = reinterpret_cast<long long*>(out + LG_SLOTS * 40);
    for (int s = threadIdx.x; s < LG_SLOTS; s += TPB) {
        ok[s] = s_keys[s]; oc[s] = s_cnt[s];
#pragma unroll
        for (int k = 0; k < 4; ++k) oa[s * 4 + k] = s_acc[s * 4 + k];
    }
    if (threadIdx.x == 0) { int* tail = reinterpret_cast<int*>(out + LG_SLOTS * 48); tail[0] = s_flags; tail[1] = 0; }
}

int sdqh_xgroupby_partial(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, void* device_block) {
    if (!ctx || nrows < 0 || !prog || !device_block) return fail(ctx, SDQH_ERR_INVALID, "xgroupby_partial: bad arguments");
    if (ctx->compile_only) return fail(ctx, SDQH_ERR_UNSUPPORTED, "xgroupby_partial: compile-only context");
    (void)hipSetDevice(ctx->device);
    return xgroupby_launch(ctx, nrows, prog, device_block, true);
}

int sdqh_xgroupby_fold(sdqh_ctx* ctx, const void* device_blocks, int nblocks, void* result_block) {
    if (!ctx || !device_blocks || nblocks < 1 || nblocks > SDQH_MAX_PARTS || !result_block) return fail(ctx, SDQH_ERR_INVALID, "xgroupby_fold: bad arguments");
    if (ctx->compile_only) return fail(ctx, SDQH_ERR_UNSUPPORTED, "xgroupby_fold: compile-only context");
    if (!host_block_contains(ctx, result_block, sdqh_xgroupby_block_bytes())) return fail(ctx, SDQH_ERR_INVALID, "xgroupby_fold: the result block must come from sdqh_host_alloc (sdqh_xgroupby_block_bytes() bytes)");
    (void)hipSetDevice(ctx->device);
    call_begin(ctx);
    uint32_t* done = reinterpret_cast<uint32_t*>(static_cast<char*>(result_block) + XGROUPBY_DONE_OFFSET);
    host_init(ctx, done, 0, 4);
    { KernelScope ks(ctx, "k_groups_fold");
      hipLaunchKernelGGL(k_groups_fold, dim3(1), dim3(TPB), 0, ctx->stream, static_cast<const char*>(device_blocks), nblocks, sdqh_xgroupby_block_bytes(), static_cast<char*>(result_block)); }
    call_end(ctx);
    if (hipGetLastError() != hipSuccess) return fail(ctx, SDQH_ERR_DEVICE, "xgroupby_fold: launch failed");
    if (stream_store32(ctx, ctx->stream, done, 1) != SDQH_OK) *done = 2;
    return SDQH_OK;
}

int sdqh_host_wait_word(sdqh_ctx* ctx, const void* word, uint32_t value) {
    if (!ctx || !word) return fail(ctx, SDQH_ERR_INVALID, "host_wait_word: bad arguments");
    return wait_word(ctx, static_cast<const volatile uint32_t*>(word), value);
}

int sdqh_xbuild(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, int64_t key_lo, int64_t key_hi, int accumulate, sdqh_table** out) {
    if (!ctx || nrows < 0 || !out) return fail(ctx, SDQH_ERR_INVALID, "xbuild: bad arguments");
    if (nrows >= 0xFFFFFFFEll) return fail(ctx, SDQH_ERR_UNSUPPORTED, "xbuild: build side limited to 2^32-2 rows per GPU");
    if (!ctx->compile_only) (void)hipSetDevice(ctx->device);
    static const bool phase_times = getenv("SDQLPY_AMD_XBUILD_TIMES") != nullptr;      // (host time per phase of this call, to stderr: tuning aid)
    auto t_now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = t_now();
    auto lap = [&](const char* what) { if (phase_times) { auto t1 = t_now(); fprintf(stderr, "xbuild %-12s %6.1f us\n", what, std::chrono::duration<double, std::micro>(t1 - t0).count()); t0 = t1; } };
    XInfo x;
    if (int rc = analyse(ctx, nrows, prog, SDQH_MAX_PAYLOAD, true, false, &x)) return rc;
    lap("analyse");
    vstage_plan(ctx, nrows, &x);
    lap("vstage_plan");
    hipFunction_t fn;
    if (int rc = kernel_for(ctx, x, SINK_STAGE, false, &fn)) {
        if (!ctx->compile_only || ctx->err.find("kernel specialised") == std::string::npos) return rc;
        sdqh_table* dummy = new sdqh_table();                 // compile-only: a table later programs can name (never dereferenced)
        dummy->npay = prog->nvals; dummy->accumulate = accumulate != 0; dummy->stage.acc_stride = 4;
        *out = dummy;
        return SDQH_OK;
    }
    const bool bounded = key_lo <= key_hi && key_lo > INT64_MIN / 2 && key_hi < INT64_MAX / 2;
    bool want_bm = false;
    if (bounded && ctx->opt_direct_index) {
        const uint64_t range = (uint64_t)(key_hi - key_lo) + 1;
        want_bm = range <= (1ull << 31) && range <= 64ull * (uint64_t)std::max<int64_t>(nrows, 1024);
    }
    sdqh_table* tb = new sdqh_table();
    tb->npay = prog->nvals; tb->accumulate = accumulate != 0; tb->nrows_build = nrows;
    { const sdqh_xop& ko = prog->ops[prog->key];                       // a strictly increasing key column (a primary key): no duplicate keys whatever the gates pass
      if (ko.code == SDQH_X_COL && ko.col->dtype == SDQH_I64 && !ko.col->transient && nrows > 0 && column_increasing(ctx, const_cast<sdqh_column*>(ko.col))) tb->keys_unique = true; }
    call_begin(ctx);
    int stage_batch = x.tight ? (X8_STEP * X8_U) / 128 : X_LB;
    if (ctx->opt_x_waves > 0) {                                           // (tuning: wave segments per CU — the stage cuts segments in whole batches of 128 rows)
        const int64_t seg = (nrows + (int64_t)ctx->num_cu * ctx->opt_x_waves - 1) / ((int64_t)ctx->num_cu * ctx->opt_x_waves);
        const int64_t per = (int64_t)128 * stage_batch;
        stage_batch = (int)std::min<int64_t>(1 << 20, std::max<int64_t>(1, (seg + per - 1) / per) * stage_batch);
    }
    lap("kernel_for");
    int rc = stage_setup_computed(ctx, tb, nrows, prog->nvals, stage_batch);
    lap("stage_setup");

    uint64_t capmax = 1024;
    while (capmax < 2 * (uint64_t)std::max<int64_t>(nrows, 1)) capmax <<= 1;
    tb->capmax = capmax;
    int32_t* flags = nullptr;
    if (!rc && accumulate >= 16) tb->stage.acc_stride = std::max(1, std::min(accumulate - 16, SDQH_TUPLE_MAX_VALUES));      // the caller's plan knows how many sums an entry gets
    if (!rc) {
        tb->hdr = static_cast<TableHeader*>(tb_alloc(ctx, tb, sizeof(TableHeader)));
        flags = static_cast<int32_t*>(tb_alloc(ctx, tb, 64));
        if (want_bm) { tb->nwords = ((uint64_t)(key_hi - key_lo) + 32) / 32; tb->bm = static_cast<uint32_t*>(tb_alloc(ctx, tb, tb->nwords * 4 + 64)); }
        if (!tb->hdr || !flags || (tb->nwords && !tb->bm)) rc = fail(ctx, SDQH_ERR_NOMEM, "xbuild: out of device memory");
    }
    if (!rc) {
        tb->dev.hdr = tb->hdr; tb->dev.shits = tb->stage.shits; tb->dev.sacc = tb->stage.sacc; tb->dev.acc_stride = tb->stage.acc_stride;
        tb->dev.bm = tb->bm; tb->dev.bm_lo = key_lo; tb->dev.bm_hi = key_hi; tb->dev.bitmap_only = 0; tb->dev.bm_shift = 0;
        for (int p = 0; p < prog->nvals; ++p) tb->dev.pay[p] = tb->stage.pay[p];
        tb->stage.bm = tb->bm; tb->stage.bm_lo = key_lo; tb->stage.bm_hi = key_hi; tb->stage.hdr = tb->hdr; tb->stage.bm_shift = 0;
        void* ptr[6]; size_t bytes[6]; unsigned char byte[6]; int n = 0;
        ptr[n] = tb->hdr; bytes[n] = sizeof(TableHeader); byte[n++] = 0;
        ptr[n] = flags; bytes[n] = 8; byte[n++] = 0;
        if (tb->bm) { ptr[n] = tb->bm; bytes[n] = tb->nwords * 4; byte[n++] = 0; }
        // ROW INDEX: keys strictly increasing with the rows, an exact bitmap over the key -> the stage kernel writes, per bitmap word, the
        // stage row of the word's first key; nothing is ranked or inserted afterwards (sdqh_kernels.hpp: DevTable)
        const bool row_index = ctx->opt_row_index && tb->bm && tb->keys_unique && nrows < ((int64_t)1 << 31) && tb->stage.nseg < (1 << 30);
        if (row_index) {
            tb->stage.wrow = static_cast<uint32_t*>(tb_alloc(ctx, tb, tb->nwords * 4 + 64));
            tb->stage.seg_first = static_cast<SegFirst*>(tb_alloc(ctx, tb, (size_t)tb->stage.nseg * sizeof(SegFirst) + 64));
            tb->wexc = static_cast<WordExc*>(tb_alloc(ctx, tb, (size_t)tb->stage.nseg * sizeof(WordExc) + 64));
            if (!tb->stage.wrow || !tb->stage.seg_first || !tb->wexc) { tb->stage.wrow = nullptr; tb->stage.seg_first = nullptr; tb->wexc = nullptr; }
        }
        if (!tb->stage.wrow) { void* rp[2]; size_t rb[2]; const int nr = prefill_direct_refs(ctx, tb, rp, rb); for (int i = 0; i < nr; ++i) { ptr[n] = rp[i]; bytes[n] = rb[i]; byte[n++] = 0xFF; } }
        lap("allocs");
        fill_regions(ctx, ptr, bytes, byte, n);
        lap("fill");
        XArgs a;
        rc = fill_xargs(ctx, x, &a, flags, bounded ? key_lo : 1, bounded ? key_hi : 0);
        lap("fill_xargs");
        if (!rc) {
            // the segments were cut by stage_setup_computed: the kernel's geometry must be the stage's
            Geometry g{(unsigned)((tb->stage.nseg + TPB / WAVE - 1) / (TPB / WAVE)), tb->stage.seg_rows, tb->stage.nseg};
            XStage<1>::Args sa{tb->stage};
            ctx->next_model_bytes = model_stream_bytes(x, nrows, x.vstage);
            rc = launch(ctx, fn, x.vstage ? VSTAGE_ENTRY : launch_label(SINK_STAGE, false, x.tight), a, sa, nrows, g);
        }
        lap("launch");
        call_end(ctx);
        int f = 0;
        // Can the kernel raise a flag at all (a key outside the bounds, a key part that does not pack)?  Not when the key's own
        // interval — from the minima / maxima of the columns it is computed from — lies inside: then there is nothing to read back,
        // and the host does not stand between this build and the next launch (30-40 us of an idle device in the middle of Q3).
        const sdqh_xop& ko = prog->ops[prog->key];
        bool can_fail = bounded || ko.code == SDQH_X_PACK2 || ko.code == SDQH_X_SELECT;
        if (!rc && can_fail && ko.code != SDQH_X_SELECT) {
            int64_t lo = 0, hi = -1, alo = 0, ahi = -1, blo = 0, bhi = -1;
            const bool packs = ko.code != SDQH_X_PACK2 || (op_interval(ctx, x, ko.a, &alo, &ahi) && op_interval(ctx, x, ko.b, &blo, &bhi) &&
                                                           alo >= 0 && ahi <= 0xFFFFFFFFll && blo >= 0 && bhi <= 0xFFFFFFFFll);
            const bool inside = !bounded || (ko.code == SDQH_X_PACK2 ? packs && (int64_t)(((uint64_t)alo << 32) | (uint64_t)blo) >= key_lo && (int64_t)(((uint64_t)ahi << 32) | (uint64_t)bhi) <= key_hi
                                                                      : op_interval(ctx, x, prog->key, &lo, &hi) && lo >= key_lo && hi <= key_hi);
            if (packs && inside) can_fail = false;
        }
        lap("interval");
        if (!rc && can_fail) rc = read_flags(ctx, flags, &f);
        lap("flags");
        if (!rc && (f & 2)) rc = fail(ctx, SDQH_ERR_UNSUPPORTED, "xbuild: a key outside the given bounds / a key part outside [0, 2^32)");
    }
    if (rc) { tb_release(ctx, tb); delete tb; return rc; }
    *out = tb;
    return SDQH_OK;
}

int sdqh_xstage(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, sdqh_table** out) {
    if (!ctx || !prog || !out) return fail(ctx, SDQH_ERR_INVALID, "xstage: bad arguments");
    if (1 + prog->nvals > SDQH_MAX_COMPACT_COLS) return fail(ctx, SDQH_ERR_INVALID, "xstage: too many columns");
    // the staging half of a build: the stage sink keeps every passing row in its wave's segment, equal keys included; no bounds, no
    // bitmap, and the index a build would make lazily is never asked for — the row count stays in seg_count[] on the device
    // Which stage kernel: the VALUE queue (x_vstage8) streams every column of every row — right for a build that keeps a tenth of its
    // rows; the probe side of a join keeps a fraction of a per cent, and the queue skeleton, which streams the tested columns and
    // gathers the values of the few survivors by row, moves half the bytes (Q3's lineitem at SF=10: 0.15 ms against 0.10).
    // SDQLPY_AMD_XSTAGE_VALUES=1 is the A/B switch.
    static const bool by_values = [] { const char* e = getenv("SDQLPY_AMD_XSTAGE_VALUES"); return e && e[0] == '1'; }();
    const int was = ctx->opt_vstage;
    if (!by_values) ctx->opt_vstage = 0;
    sdqh_table* tb = nullptr;
    const int rc = sdqh_xbuild(ctx, nrows, prog, 1, 0, 0, &tb);
    ctx->opt_vstage = was;
    if (rc) return rc;
    tb->stage_only = true;
    *out = tb;
    return SDQH_OK;
}

int sdqh_xcompact(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, sdqh_column** out_cols, int64_t* out_rows) {
    if (!ctx || !prog || !out_cols || !out_rows) return fail(ctx, SDQH_ERR_INVALID, "xcompact: bad arguments");
    if (1 + prog->nvals > SDQH_MAX_COMPACT_COLS) return fail(ctx, SDQH_ERR_INVALID, "xcompact: too many columns");
    // the staging half of a build (the stage sink keeps every passing row in its segment); the index a build would make lazily is never asked for
    sdqh_table* tb = nullptr;
    if (int rc = sdqh_xbuild(ctx, nrows, prog, 1, 0, 0, &tb)) return rc;
    int rc = SDQH_OK;
    if (ctx->compile_only) { for (int c = 0; c < 1 + prog->nvals; ++c) out_cols[c] = nullptr; *out_rows = 0; }
    else rc = stage_rows_out(ctx, tb, out_cols, out_rows);
    sdqh_table_free(ctx, tb);
    return rc;
}

int sdqh_xkey_set(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, int64_t key_lo, int64_t key_hi, sdqh_table** out) {
    if (!ctx || nrows < 0 || !out) return fail(ctx, SDQH_ERR_INVALID, "xkey_set: bad arguments");
    if (!ctx->compile_only) (void)hipSetDevice(ctx->device);
    XInfo x;
    if (int rc = analyse(ctx, nrows, prog, 0, true, false, &x)) return rc;
    if (key_lo > key_hi) { if (nrows > 0) return fail(ctx, SDQH_ERR_UNSUPPORTED, "xkey_set: needs the key's bounds"); key_lo = key_hi = 0; }
    if (key_lo <= INT64_MIN / 2 || key_hi >= INT64_MAX / 2 || (uint64_t)(key_hi - key_lo) + 1 > (1ull << 31))
        return fail(ctx, SDQH_ERR_UNSUPPORTED, "xkey_set: key range too wide for a bitmap");
    hipFunction_t fn;
    if (int rc = kernel_for(ctx, x, SINK_KEYSET, false, &fn)) {
        if (!ctx->compile_only || ctx->err.find("kernel specialised") == std::string::npos) return rc;
        sdqh_table* dummy = new sdqh_table();
        dummy->bitmap_only = true; dummy->index_built = true;
        *out = dummy;
        return SDQH_OK;
    }
    call_begin(ctx);
    sdqh_table* tb = new sdqh_table();
    tb->bitmap_only = true; tb->index_built = true; tb->nrows_build = nrows;
    tb->nwords = ((uint64_t)(key_hi - key_lo) + 32) / 32;
    tb->bm = static_cast<uint32_t*>(tb_alloc(ctx, tb, tb->nwords * 4 + 64));
    tb->hdr = static_cast<TableHeader*>(tb_alloc(ctx, tb, sizeof(TableHeader) + 64));
    if (!tb->bm || !tb->hdr) { tb_release(ctx, tb); delete tb; return fail(ctx, SDQH_ERR_NOMEM, "xkey_set: out of device memory"); }
    int32_t* flags = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(tb->hdr) + sizeof(TableHeader));
    tb->dev.bm = tb->bm; tb->dev.bm_lo = key_lo; tb->dev.bm_hi = key_hi; tb->dev.bitmap_only = 1; tb->dev.hdr = tb->hdr;
    { void* ptr[2] = {tb->bm, tb->hdr}; size_t bytes[2] = {(size_t)((tb->nwords * 4 + 15) & ~(uint64_t)15), sizeof(TableHeader) + 16}; unsigned char byte[2] = {0, 0}; fill_regions(ctx, ptr, bytes, byte, 2); }
    XArgs a;
    int rc = fill_xargs(ctx, x, &a, flags, key_lo, key_hi);
    if (!rc && nrows > 0) {
        const Geometry g = geometry(ctx, nrows, false, 16, x.tight);
        XKeySet<1>::Args sa{tb->bm};
        ctx->next_model_bytes = model_stream_bytes(x, nrows, false);
        rc = launch(ctx, fn, launch_label(SINK_KEYSET, false, x.tight), a, sa, nrows, g);
    }
    call_end(ctx);
    int f = 0;
    if (!rc) rc = read_flags(ctx, flags, &f);
    if (!rc && (f & 2)) rc = fail(ctx, SDQH_ERR_UNSUPPORTED, "xkey_set: a key outside the given bounds");
    if (rc) { tb_release(ctx, tb); delete tb; return rc; }
    *out = tb;
    return SDQH_OK;
}

int sdqh_xprobe_aggregate(sdqh_ctx* ctx, int64_t nrows, const sdqh_program* prog, int lookup_op, sdqh_table* table) {
    if (!ctx || nrows < 0 || !table) return fail(ctx, SDQH_ERR_INVALID, "xprobe_aggregate: bad arguments");
    if (!table->accumulate || table->bitmap_only) return fail(ctx, SDQH_ERR_INVALID, "xprobe_aggregate: table was built without accumulators");
    if (!ctx->compile_only) (void)hipSetDevice(ctx->device);
    XInfo x;
    if (int rc = analyse(ctx, nrows, prog, SDQH_TUPLE_MAX_VALUES, false, true, &x)) return rc;
    bool gated = false;
    for (int g = 0; g < prog->ngates; ++g) gated = gated || prog->gates[g] == lookup_op;
    if (lookup_op < 0 || lookup_op >= prog->nops || prog->ops[lookup_op].code != SDQH_X_LOOKUP || prog->ops[lookup_op].table != table || !gated)
        return fail(ctx, SDQH_ERR_INVALID, "xprobe_aggregate: lookup_op must be a gate that looks `table` up");
    if (prog->nvals > table->stage.acc_stride) return fail(ctx, SDQH_ERR_INVALID, "xprobe_aggregate: the table's entries have room for fewer values than the program sums");
    x.probe_op = lookup_op;
    hipFunction_t fn;
    if (int rc = kernel_for(ctx, x, SINK_ENTRY, false, &fn)) return rc;
    table->compact_valid = false;
    table->nv = prog->nvals;
    call_begin(ctx);
    XArgs a;
    // (the flag word nothing reads here lives past every result block of ctx->result_dev: the group-by block a neighbouring call left
    // clean stays clean — Q5's xgroupby after Q3's probe-aggregate needed a fill launch for it)
    int32_t* d_flags = reinterpret_cast<int32_t*>(static_cast<char*>(ctx->result_dev) + RESULT_BYTES - 64);
    if (int rc = fill_xargs(ctx, x, &a, d_flags, 1, 0)) return rc;
    // (NOT the tiled walk, although the sink takes rows in any order: with one contiguous segment per wave the rows of a group — the
    //  lineitems of an order — meet in one wave, folded by its drain's scan into one atomic per run, and a sum is the same bits run after
    //  run; 1024-row tiles cut ten times as many groups in two, each half added by another wave in racing order — Q3 at SF=100 then differs
    //  in the last bit between two runs (test_sf100_on_one_gpu_q3_q6) for 6 % of the kernel's time)
    const Geometry g = geometry(ctx, nrows, false, 24, x.tight, false);
    XEntry<1>::Args sa{table->dev};
    ctx->next_model_bytes = model_stream_bytes(x, nrows, false);
    int rc = launch(ctx, fn, launch_label(SINK_ENTRY, false, x.tight), a, sa, nrows, g);
    call_end(ctx);
    return rc;
}

int sdqh_table_columns(sdqh_ctx* ctx, const sdqh_table* ctable, int64_t min_hits, sdqh_column** out_cols, int64_t* out_rows) {
    if (!ctx || !ctable || !out_cols || !out_rows || ctable->bitmap_only) return fail(ctx, SDQH_ERR_INVALID, "table_columns: bad arguments");
    // O(entries), like K-F itself, and all on the device: the table's own K-F buffers (build-row order) ARE the columns — the
    // only host round trip is the row count.  The columns are views: valid until the table is released or compacted again.
    sdqh_table* table = const_cast<sdqh_table*>(ctable);
    const int npay = table->npay, ncols = 1 + npay + SDQH_TUPLE_MAX_VALUES + 1, nval = table->accumulate ? table->nv : 0;
    int64_t n = 0;
    if (int rc = table_compact_resident(ctx, table, min_hits, nval < SDQH_TUPLE_MAX_VALUES, &n)) return rc;
    const DevCompactOut& o = table->compact;
    for (int c = 0; c < ncols; ++c) {
        const int acc = c - 1 - npay;
        const bool is_acc = acc >= 0 && acc < SDQH_TUPLE_MAX_VALUES;
        void* src = c == 0 ? (void*)o.keys : c <= npay ? (void*)o.pay[c - 1] : is_acc ? (acc < nval ? (void*)o.val[acc] : table->zero_rows) : (void*)o.hits;
        sdqh_column* col = new (std::nothrow) sdqh_column();
        if (!col) { for (int j = 0; j < c; ++j) { delete out_cols[j]; out_cols[j] = nullptr; } return fail(ctx, SDQH_ERR_NOMEM, "table_columns: out of host memory"); }
        col->home = ctx; col->data = src; col->nrows = n; col->dtype = is_acc ? SDQH_F64 : SDQH_I64; col->owned = false; col->transient = true;
        col->narrow_state = 0; col->code_state = 0; col->clustered = 0; col->increasing = 0;
        out_cols[c] = col;
    }
    *out_rows = n;
    return SDQH_OK;
}

}  // extern "C"
