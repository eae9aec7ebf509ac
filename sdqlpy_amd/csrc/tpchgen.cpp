// tpchgen.cpp — seed-deterministic TPCH-shaped column generator (CPU, threaded).
//
// No dbgen is available offline, so the benchmark and parity inputs come from this generator.
// It reproduces the TPCH properties the hot-path queries are sensitive to:
// sparse sorted o_orderkey (8 used of every 32), 1..7 contiguous lineitems per order, shipdate /
// commitdate / receiptdate offsets from o_orderdate, returnflag / linestatus derived from dates,
// 2-decimal money columns that round-trip through text, spec nation->region table, the
// 4-suppliers-per-part formula, 5-of-92-colour part names.
//
// Every value is a pure function of (seed, table, row index, field) through a counter-based mixer,
// so any row range of any table can be produced independently (multi-GPU shards, threads) and the
// same rows are identical however they are produced.  Dates are yyyymmdd int64, strings are UCS4
// fixed-width (numpy '<U n'), exactly the column layout the reference hands to compiled code
// (reference src/sdqlpy/sdql_lib.py:69-115).
//
// Plain C ABI, loaded with ctypes from sdqlpy_amd/tpch.py.  Null output pointers are skipped.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {

inline uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31; return x;
}
// table ids
enum { T_REGION = 1, T_NATION, T_SUPPLIER, T_CUSTOMER, T_PART, T_PARTSUPP, T_ORDERS, T_LINEITEM };

inline uint64_t rnd(uint64_t seed, int table, int field, uint64_t idx) {
    return mix64(mix64(seed + 0x9E3779B97F4A7C15ull * (uint64_t)(table * 64 + field)) ^ (idx * 0xD6E8FEB86659FD93ull));
}
inline int64_t uniform(uint64_t r, int64_t lo, int64_t hi) {   // inclusive
    return lo + (int64_t)(r % (uint64_t)(hi - lo + 1));
}

// ---- calendar: day index 0 = 1992-01-01 ---------------------------------------------------
struct Calendar {
    std::vector<int64_t> ymd;
    Calendar() {
        static const int mdays[12] = {31, 28, 31, 30, 31, 30, 31, 31, 30, 31, 30, 31};
        for (int y = 1992; y <= 1999; ++y) {
            bool leap = (y % 4 == 0 && y % 100 != 0) || (y % 400 == 0);
            for (int m = 0; m < 12; ++m) {
                int nd = mdays[m] + ((m == 1 && leap) ? 1 : 0);
                for (int d = 1; d <= nd; ++d) ymd.push_back((int64_t)y * 10000 + (m + 1) * 100 + d);
            }
        }
    }
};
const Calendar& cal() { static Calendar c; return c; }
const int ORDERDATE_MAX_DAY = 2405;   // 1998-08-02

int64_t scaled(double sf, int64_t base) {
    int64_t n = (int64_t)std::llround(sf * (double)base);
    return n < 1 ? 1 : n;
}

int64_t rows_of(int table, double sf) {
    switch (table) {
        case T_REGION: return 5;
        case T_NATION: return 25;
        case T_SUPPLIER: return scaled(sf, 10000);
        case T_CUSTOMER: return scaled(sf, 150000);
        case T_PART: return scaled(sf, 200000);
        case T_PARTSUPP: return 4 * scaled(sf, 200000);
        case T_ORDERS: return scaled(sf, 1500000);
        default: return -1;
    }
}

void put_str(uint32_t* dst, int width, const char* s) {
    int i = 0;
    for (; s[i] && i < width; ++i) dst[i] = (uint32_t)(unsigned char)s[i];
    for (; i < width; ++i) dst[i] = 0;
}

const char* REGIONS[5] = {"AFRICA", "AMERICA", "ASIA", "EUROPE", "MIDDLE EAST"};
struct NationDef { const char* name; int region; };
const NationDef NATIONS[25] = {
    {"ALGERIA", 0}, {"ARGENTINA", 1}, {"BRAZIL", 1}, {"CANADA", 1}, {"EGYPT", 4}, {"ETHIOPIA", 0},
    {"FRANCE", 3}, {"GERMANY", 3}, {"INDIA", 2}, {"INDONESIA", 2}, {"IRAN", 4}, {"IRAQ", 4},
    {"JAPAN", 2}, {"JORDAN", 4}, {"KENYA", 0}, {"MOROCCO", 0}, {"MOZAMBIQUE", 0}, {"PERU", 1},
    {"CHINA", 2}, {"ROMANIA", 3}, {"SAUDI ARABIA", 4}, {"VIETNAM", 2}, {"RUSSIA", 3},
    {"UNITED KINGDOM", 3}, {"UNITED STATES", 1}};
const char* SEGMENTS[5] = {"AUTOMOBILE", "BUILDING", "FURNITURE", "MACHINERY", "HOUSEHOLD"};
const char* COLOURS[92] = {
    "almond", "antique", "aquamarine", "azure", "beige", "bisque", "black", "blanched", "blue", "blush",
    "brown", "burlywood", "burnished", "chartreuse", "chiffon", "chocolate", "coral", "cornflower",
    "cornsilk", "cream", "cyan", "dark", "deep", "dim", "dodger", "drab", "firebrick", "floral", "forest",
    "frosted", "gainsboro", "ghost", "goldenrod", "green", "grey", "honeydew", "hot", "indian", "ivory",
    "khaki", "lace", "lavender", "lawn", "lemon", "light", "lime", "linen", "magenta", "maroon", "medium",
    "metallic", "midnight", "mint", "misty", "moccasin", "navajo", "navy", "olive", "orange", "orchid",
    "pale", "papaya", "peach", "peru", "pink", "plum", "powder", "puff", "purple", "red", "rose", "rosy",
    "royal", "saddle", "salmon", "sandy", "seashell", "sienna", "sky", "slate", "smoke", "snow", "spring",
    "steel", "tan", "thistle", "tomato", "turquoise", "violet", "wheat", "white", "yellow"};

inline int64_t retail_cents(int64_t partkey) {   // TPCH 4.2.3: P_RETAILPRICE
    return 90000 + ((partkey / 10) % 20001) + 100 * (partkey % 1000);
}
inline int64_t partsupp_suppkey(int64_t partkey, int j, int64_t S) {   // TPCH 4.2.3: PS_SUPPKEY
    return (partkey + j * (S / 4 + (partkey - 1) / S)) % S + 1;
}
inline int64_t order_key(int64_t i) { return (i / 8) * 32 + (i % 8) + 1; }   // sparse: 8 of every 32
inline int lines_of(uint64_t seed, int64_t order_idx) { return 1 + (int)(rnd(seed, T_ORDERS, 7, (uint64_t)order_idx) % 7); }
inline int order_day(uint64_t seed, int64_t order_idx) { return (int)(rnd(seed, T_ORDERS, 2, (uint64_t)order_idx) % (ORDERDATE_MAX_DAY + 1)); }

template <class F>
void parallel_ranges(int64_t begin, int64_t end, int threads, F f) {
    int64_t n = end - begin;
    if (threads < 1) threads = 1;
    if (n < 4096 || threads == 1) { f(begin, end, 0); return; }
    std::vector<std::thread> pool;
    int64_t chunk = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        int64_t b = begin + t * chunk, e = std::min(end, b + chunk);
        if (b >= e) break;
        pool.emplace_back([=] { f(b, e, t); });
    }
    for (auto& th : pool) th.join();
}

}  // namespace

extern "C" {

int64_t tpchgen_rows(int table, double sf) { return rows_of(table, sf); }

// Number of lineitem rows belonging to orders [order_begin, order_end).
int64_t tpchgen_lineitem_rows(double sf, uint64_t seed, int64_t order_begin, int64_t order_end, int threads) {
    (void)sf;
    if (threads < 1) threads = 1;
    std::vector<int64_t> part((size_t)threads, 0);
    parallel_ranges(order_begin, order_end, threads, [&](int64_t b, int64_t e, int t) {
        int64_t s = 0;
        for (int64_t i = b; i < e; ++i) s += lines_of(seed, i);
        part[(size_t)t] += s;
    });
    int64_t total = 0;
    for (auto v : part) total += v;
    return total;
}

void tpchgen_region(int64_t* r_regionkey, uint32_t* r_name /*U25*/) {
    for (int i = 0; i < 5; ++i) {
        if (r_regionkey) r_regionkey[i] = i;
        if (r_name) put_str(r_name + 25 * i, 25, REGIONS[i]);
    }
}

void tpchgen_nation(int64_t* n_nationkey, uint32_t* n_name /*U25*/, int64_t* n_regionkey) {
    for (int i = 0; i < 25; ++i) {
        if (n_nationkey) n_nationkey[i] = i;
        if (n_name) put_str(n_name + 25 * i, 25, NATIONS[i].name);
        if (n_regionkey) n_regionkey[i] = NATIONS[i].region;
    }
}

void tpchgen_supplier(double sf, uint64_t seed, int64_t begin, int64_t end,
                      int64_t* s_suppkey, int64_t* s_nationkey, double* s_acctbal, int threads) {
    (void)sf;
    parallel_ranges(begin, end, threads, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            int64_t o = i - begin;
            if (s_suppkey) s_suppkey[o] = i + 1;
            if (s_nationkey) s_nationkey[o] = uniform(rnd(seed, T_SUPPLIER, 1, (uint64_t)i), 0, 24);
            if (s_acctbal) s_acctbal[o] = (double)uniform(rnd(seed, T_SUPPLIER, 2, (uint64_t)i), -99999, 999999) / 100.0;
        }
    });
}

void tpchgen_customer(double sf, uint64_t seed, int64_t begin, int64_t end,
                      int64_t* c_custkey, int64_t* c_nationkey, double* c_acctbal,
                      uint32_t* c_mktsegment /*U10*/, int threads) {
    (void)sf;
    parallel_ranges(begin, end, threads, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            int64_t o = i - begin;
            if (c_custkey) c_custkey[o] = i + 1;
            if (c_nationkey) c_nationkey[o] = uniform(rnd(seed, T_CUSTOMER, 1, (uint64_t)i), 0, 24);
            if (c_acctbal) c_acctbal[o] = (double)uniform(rnd(seed, T_CUSTOMER, 2, (uint64_t)i), -99999, 999999) / 100.0;
            if (c_mktsegment) put_str(c_mktsegment + 10 * o, 10, SEGMENTS[rnd(seed, T_CUSTOMER, 3, (uint64_t)i) % 5]);
        }
    });
}

void tpchgen_part(double sf, uint64_t seed, int64_t begin, int64_t end,
                  int64_t* p_partkey, uint32_t* p_name /*U55*/, double* p_retailprice, int64_t* p_size, int threads) {
    (void)sf;
    parallel_ranges(begin, end, threads, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            int64_t o = i - begin, pk = i + 1;
            if (p_partkey) p_partkey[o] = pk;
            if (p_retailprice) p_retailprice[o] = (double)retail_cents(pk) / 100.0;
            if (p_size) p_size[o] = uniform(rnd(seed, T_PART, 2, (uint64_t)i), 1, 50);
            if (p_name) {
                // five distinct colours, space separated (TPCH 4.2.3: P_NAME)
                int pick[5];
                for (int k = 0; k < 5; ++k) {
                    int c = (int)(rnd(seed, T_PART, 8 + k, (uint64_t)i) % (uint64_t)(92 - k));
                    // k-th pick among the colours not picked yet
                    for (int a = 0; a < 92; ++a) {
                        bool used = false;
                        for (int q = 0; q < k; ++q) used |= (pick[q] == a);
                        if (used) continue;
                        if (c == 0) { pick[k] = a; break; }
                        --c;
                    }
                }
                char buf[128]; int len = 0;
                for (int k = 0; k < 5; ++k) {
                    if (k) buf[len++] = ' ';
                    const char* w = COLOURS[pick[k]];
                    while (*w) buf[len++] = *w++;
                }
                buf[len] = 0;
                put_str(p_name + 55 * o, 55, buf);
            }
        }
    });
}

// partsupp rows [begin,end): row r = part (r/4), supplier slot (r%4)
void tpchgen_partsupp(double sf, uint64_t seed, int64_t begin, int64_t end,
                      int64_t* ps_partkey, int64_t* ps_suppkey, double* ps_availqty, double* ps_supplycost, int threads) {
    const int64_t S = rows_of(T_SUPPLIER, sf);
    parallel_ranges(begin, end, threads, [&](int64_t b, int64_t e, int) {
        for (int64_t r = b; r < e; ++r) {
            int64_t o = r - begin, pk = r / 4 + 1; int j = (int)(r % 4);
            if (ps_partkey) ps_partkey[o] = pk;
            if (ps_suppkey) ps_suppkey[o] = partsupp_suppkey(pk, j, S);
            if (ps_availqty) ps_availqty[o] = (double)uniform(rnd(seed, T_PARTSUPP, 1, (uint64_t)r), 1, 9999);
            if (ps_supplycost) ps_supplycost[o] = (double)uniform(rnd(seed, T_PARTSUPP, 2, (uint64_t)r), 100, 100000) / 100.0;
        }
    });
}

void tpchgen_orders(double sf, uint64_t seed, int64_t begin, int64_t end,
                    int64_t* o_orderkey, int64_t* o_custkey, int64_t* o_orderdate,
                    int64_t* o_shippriority, double* o_totalprice, int threads) {
    const int64_t C = rows_of(T_CUSTOMER, sf);
    const Calendar& c = cal();
    parallel_ranges(begin, end, threads, [&](int64_t b, int64_t e, int) {
        for (int64_t i = b; i < e; ++i) {
            int64_t o = i - begin;
            if (o_orderkey) o_orderkey[o] = order_key(i);
            if (o_custkey) {
                // customers with custkey % 3 == 0 never order (TPCH 4.2.3: O_CUSTKEY)
                // the j-th key of the sequence 1,2,4,5,7,8,... is j + j/2 + 1
                int64_t valid = C - C / 3;
                int64_t j = uniform(rnd(seed, T_ORDERS, 1, (uint64_t)i), 0, valid - 1);
                o_custkey[o] = j + j / 2 + 1;
            }
            if (o_orderdate) o_orderdate[o] = c.ymd[(size_t)order_day(seed, i)];
            if (o_shippriority) o_shippriority[o] = 0;
            if (o_totalprice) o_totalprice[o] = (double)uniform(rnd(seed, T_ORDERS, 3, (uint64_t)i), 85000, 55000000) / 100.0;
        }
    });
}

// Lineitems of orders [order_begin, order_end); nrows must equal tpchgen_lineitem_rows(...).
// Flags are UCS4 width-1 strings.  Returns rows written (or -1 on a count mismatch).
int64_t tpchgen_lineitem(double sf, uint64_t seed, int64_t order_begin, int64_t order_end, int64_t nrows,
                         int64_t* l_orderkey, int64_t* l_partkey, int64_t* l_suppkey, int64_t* l_linenumber,
                         double* l_quantity, double* l_extendedprice, double* l_discount, double* l_tax,
                         uint32_t* l_returnflag, uint32_t* l_linestatus,
                         int64_t* l_shipdate, int64_t* l_commitdate, int64_t* l_receiptdate, int threads) {
    const int64_t P = rows_of(T_PART, sf), S = rows_of(T_SUPPLIER, sf);
    const Calendar& c = cal();
    if (threads < 1) threads = 1;
    int64_t norders = order_end - order_begin;
    if (norders <= 0) return 0;
    // per-thread order chunks and their starting lineitem offsets
    int T = (int)std::min<int64_t>(threads, std::max<int64_t>(1, norders / 1024));
    std::vector<int64_t> cb((size_t)T + 1), off((size_t)T + 1, 0);
    for (int t = 0; t <= T; ++t) cb[(size_t)t] = order_begin + norders * t / T;
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back([&, t] {
            int64_t s = 0;
            for (int64_t i = cb[(size_t)t]; i < cb[(size_t)t + 1]; ++i) s += lines_of(seed, i);
            off[(size_t)t + 1] = s;
        });
        for (auto& th : pool) th.join();
    }
    for (int t = 0; t < T; ++t) off[(size_t)t + 1] += off[(size_t)t];
    if (off[(size_t)T] != nrows) return -1;
    const int64_t CUTOFF = 19950617;
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t) pool.emplace_back([&, t] {
        int64_t r = off[(size_t)t];
        for (int64_t i = cb[(size_t)t]; i < cb[(size_t)t + 1]; ++i) {
            int nl = lines_of(seed, i);
            int od = order_day(seed, i);
            int64_t ok = order_key(i);
            for (int ln = 0; ln < nl; ++ln, ++r) {
                uint64_t idx = (uint64_t)i * 8 + (uint64_t)ln;
                int64_t pk = uniform(rnd(seed, T_LINEITEM, 1, idx), 1, P);
                int j = (int)(rnd(seed, T_LINEITEM, 2, idx) % 4);
                int64_t qty = uniform(rnd(seed, T_LINEITEM, 3, idx), 1, 50);
                int64_t disc = uniform(rnd(seed, T_LINEITEM, 4, idx), 0, 10);
                int64_t tax = uniform(rnd(seed, T_LINEITEM, 5, idx), 0, 8);
                int sd = od + (int)uniform(rnd(seed, T_LINEITEM, 6, idx), 1, 121);
                int cd = od + (int)uniform(rnd(seed, T_LINEITEM, 7, idx), 30, 90);
                int rd = sd + (int)uniform(rnd(seed, T_LINEITEM, 8, idx), 1, 30);
                int64_t ship = c.ymd[(size_t)sd], commit = c.ymd[(size_t)cd], receipt = c.ymd[(size_t)rd];
                if (l_orderkey) l_orderkey[r] = ok;
                if (l_partkey) l_partkey[r] = pk;
                if (l_suppkey) l_suppkey[r] = partsupp_suppkey(pk, j, S);
                if (l_linenumber) l_linenumber[r] = ln + 1;
                if (l_quantity) l_quantity[r] = (double)qty;
                if (l_extendedprice) l_extendedprice[r] = (double)(qty * retail_cents(pk)) / 100.0;
                if (l_discount) l_discount[r] = (double)disc / 100.0;
                if (l_tax) l_tax[r] = (double)tax / 100.0;
                if (l_returnflag) l_returnflag[r] = (receipt <= CUTOFF) ? ((rnd(seed, T_LINEITEM, 9, idx) & 1) ? 'R' : 'A') : 'N';
                if (l_linestatus) l_linestatus[r] = (ship > CUTOFF) ? 'O' : 'F';
                if (l_shipdate) l_shipdate[r] = ship;
                if (l_commitdate) l_commitdate[r] = commit;
                if (l_receiptdate) l_receiptdate[r] = receipt;
            }
        }
    });
    for (auto& th : pool) th.join();
    return nrows;
}

}  // extern "C"
