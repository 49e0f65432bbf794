// psa_host.cpp — host side of libpsa_amd: the 27x27 table build, the
// cross-shard fold, the shard split and the synthetic generator.  No HIP here.
//
// The reference evaluates the substitution rules per char pair inside its hot
// loop (cpu_funcs.c:277-285 -> cuda_funcs.cu:310-421).  Everything those
// functions return depends only on (c1, c2, weights, is_max), so this file
// evaluates them once per 27x27 pair and hands the kernels three tables:
// pair score P, substitute SUB, and the rank of the substitution delta.
#include "psa_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace psa {

namespace {

thread_local std::string g_last_error;

// (the group lists: psa_internal.hpp)
struct GroupMasks {
    uint32_t cons[PSA_ALPHABET];
    uint32_t semi[PSA_ALPHABET];
    GroupMasks()
    {
        std::memset(cons, 0, sizeof cons);
        std::memset(semi, 0, sizeof semi);
        for (size_t g = 0; g < sizeof kConservative / sizeof *kConservative; ++g)
            for (const char* p = kConservative[g]; *p; ++p) cons[*p - 'A'] |= 1u << g;
        for (size_t g = 0; g < sizeof kSemiConservative / sizeof *kSemiConservative; ++g)
            for (const char* p = kSemiConservative[g]; *p; ++p) semi[*p - 'A'] |= 1u << g;
    }
};

const GroupMasks& masks()
{
    static const GroupMasks m;
    return m;
}

// get_weight, cuda_funcs.cu:442-452
double weight_of(char sign, const double* w)
{
    switch (sign) {
    case PSA_SIGN_STAR:  return w[0];
    case PSA_SIGN_COLON: return -w[1];
    case PSA_SIGN_DOT:   return -w[2];
    case PSA_SIGN_SPACE: return -w[3];
    }
    return 0.0;
}

int class_of(char sign)
{
    switch (sign) {
    case PSA_SIGN_STAR:  return 0;
    case PSA_SIGN_COLON: return 1;
    case PSA_SIGN_DOT:   return 2;
    default:             return 3;
    }
}

// get_substitute_by_sign_with_restrictions, cuda_funcs.cu:412-421.
// Returns a code 0..25 or -1.
int first_allowed(int by, char sign, int replaced)
{
    for (int ch = 0; ch < PSA_ALPHABET; ++ch)
        if (sign_of(by, ch) == sign && sign_of(replaced, ch) != PSA_SIGN_COLON) return ch;
    return -1;
}

// get_optimal_substitute, cuda_funcs.cu:396-409 (>= for max, <= for min: the
// first-listed option wins ties).
int choose(bool is_max, double d1, int s1, double d2, int s2)
{
    if ((is_max && d1 >= d2) || (!is_max && d1 <= d2))
        if (s1 >= 0) return s1;
    if (s2 >= 0) return s2;
    return s1;
}

// get_substitute, cuda_funcs.cu:310-393, on codes.  -1 = no substitution.
int substitute(int c1, int c2, const double* w, bool is_max)
{
    const char sign = sign_of(c1, c2);
    if (is_max) {
        if (sign == PSA_SIGN_DOT || sign == PSA_SIGN_SPACE) return c1;   // -> '*'
        double d_dot, d_blank;
        if (sign == PSA_SIGN_STAR) {
            d_dot = -w[0] - w[2];
            d_blank = -w[0] - w[3];
        } else {
            d_dot = w[1] - w[2];
            d_blank = w[1] - w[3];
        }
        return choose(true, d_dot, first_allowed(c1, PSA_SIGN_DOT, c2),
                      d_blank, first_allowed(c1, PSA_SIGN_SPACE, c2));
    }
    const int s_colon = first_allowed(c1, PSA_SIGN_COLON, c2);
    const int s_dot = first_allowed(c1, PSA_SIGN_DOT, c2);
    const int s_blank = first_allowed(c1, PSA_SIGN_SPACE, c2);
    double d1, d2;
    int s1, s2;
    switch (sign) {
    case PSA_SIGN_STAR:  d1 = -w[0] - w[2]; s1 = s_dot;   d2 = -w[0] - w[3]; s2 = s_blank; break;
    case PSA_SIGN_COLON: d1 = w[1] - w[2];  s1 = s_dot;   d2 = w[1] - w[3];  s2 = s_blank; break;
    case PSA_SIGN_DOT:   d1 = w[2] - w[1];  s1 = s_colon; d2 = w[2] - w[3];  s2 = s_blank; break;
    default:             d1 = w[3] - w[1];  s1 = s_colon; d2 = w[3] - w[2];  s2 = s_dot;   break;
    }
    const int got = choose(false, d1, s1, d2, s2);
    if ((sign == PSA_SIGN_DOT || sign == PSA_SIGN_SPACE) && got < 0) return c1;   // cuda_funcs.cu:389-390
    return got;
}

char symbol_of(int code) { return code == kCodeHyphen ? PSA_HYPHEN : static_cast<char>('A' + code); }

}  // namespace

void set_last_error(const std::string& msg) { g_last_error = msg; }

// get_hashtable_sign (cuda_funcs.cu:424-439) over get_pair_sign
// (cuda_funcs.cu:495-502), on codes 0..26.
char sign_of(int c1, int c2)
{
    if (c1 == kCodeHyphen && c2 == kCodeHyphen) return PSA_SIGN_STAR;
    if (c1 == kCodeHyphen || c2 == kCodeHyphen) return PSA_SIGN_SPACE;
    if (c1 == c2) return PSA_SIGN_STAR;
    const GroupMasks& m = masks();
    if (m.cons[c1] & m.cons[c2]) return PSA_SIGN_COLON;
    if (m.semi[c1] & m.semi[c2]) return PSA_SIGN_DOT;
    return PSA_SIGN_SPACE;
}

bool build_tables(const double weights[4], int is_max, Tables* t, std::string* err)
{
    for (int k = 0; k < 4; ++k)
        if (!std::isfinite(weights[k])) {
            if (err) *err = "weights must be finite";
            return false;
        }
    std::memset(t, 0, sizeof *t);
    std::memcpy(t->weights, weights, sizeof t->weights);
    t->is_max = is_max ? 1 : 0;
    const bool mx = is_max != 0;
    constexpr int kSyms = PSA_ALPHABET + 1;

    double delta[kSyms][kSyms];
    bool has[kSyms][kSyms];
    std::vector<double> distinct;
    for (int c1 = 0; c1 < kSyms; ++c1)
        for (int c2 = 0; c2 < kSyms; ++c2) {
            const char sign = sign_of(c1, c2);
            const double p = weight_of(sign, weights);
            const int sub = substitute(c1, c2, weights, mx);
            t->pair[c2 * kRow + c1] = p;
            t->cls[c2 * kRow + c1] = static_cast<unsigned char>(class_of(sign));
            has[c1][c2] = sub >= 0;
            if (sub >= 0) {
                // cpu_funcs.c:285: weight(sign(c1, sub)) - pair_score, in that order
                delta[c1][c2] = weight_of(sign_of(c1, sub), weights) - p;
                t->sub[c2 * kRow + c1] = static_cast<unsigned char>(symbol_of(sub));
                if (std::find(distinct.begin(), distinct.end(), delta[c1][c2]) == distinct.end())
                    distinct.push_back(delta[c1][c2]);
            }
        }
    // padding rows/columns: +0.0 score, class 4, never a substitution
    for (int r = 0; r < kRow; ++r)
        for (int c = 0; c < kRow; ++c)
            if (r >= kSyms || c >= kSyms) t->cls[r * kRow + c] = 4;

    // rank: higher = better for the requested mode; ties in value share a
    // rank so "strictly better" (cpu_funcs.c:287-288) is "strictly higher rank".
    std::sort(distinct.begin(), distinct.end());
    if (!mx) std::reverse(distinct.begin(), distinct.end());
    if (static_cast<int>(distinct.size()) >= kMaxRanks) {
        if (err) *err = "internal: more distinct substitution deltas than expected";
        return false;
    }
    t->nranks = static_cast<int>(distinct.size());
    t->delta_of_rank[0] = mx ? -INFINITY : INFINITY;
    for (int r = 0; r < t->nranks; ++r) t->delta_of_rank[r + 1] = distinct[r];
    for (int k = 0; k < 4; ++k) {
        t->class_best_delta[k] = mx ? -INFINITY : INFINITY;
        t->class_worst_delta[k] = mx ? INFINITY : -INFINITY;
    }
    for (int c1 = 0; c1 < kSyms; ++c1)
        for (int c2 = 0; c2 < kSyms; ++c2) {
            const int k = t->cls[c2 * kRow + c1];
            if (!has[c1][c2]) {
                // a pair with no substitution cannot raise the per-offset best
                t->class_worst_delta[k] = mx ? -INFINITY : INFINITY;
                continue;
            }
            const int r = static_cast<int>(std::find(distinct.begin(), distinct.end(), delta[c1][c2]) -
                                           distinct.begin()) + 1;
            t->rank[c2 * kRow + c1] = static_cast<unsigned char>(r);
            const double d = delta[c1][c2];
            if (mx) {
                t->class_best_delta[k] = std::max(t->class_best_delta[k], d);
                t->class_worst_delta[k] = std::min(t->class_worst_delta[k], d);
            } else {
                t->class_best_delta[k] = std::min(t->class_best_delta[k], d);
                t->class_worst_delta[k] = std::max(t->class_worst_delta[k], d);
            }
        }
    return true;
}

}  // namespace psa

// ---------------------------------------------------------------------------
// C ABI, host-only entry points
// ---------------------------------------------------------------------------
extern "C" {

const char* psa_last_error(void) { return psa::g_last_error.c_str(); }

const char* psa_version(void) { return "psa_amd 0.3 (gfx950)"; }

psa_status psa_build_tables(const double weights[4], int is_max,
                            double* P, unsigned char* SUB, double* DELTA, char* SIGN)
{
    if (!weights) {
        psa::set_last_error("psa_build_tables: weights is NULL");
        return PSA_ERR_ARG;
    }
    psa::Tables t;
    std::string err;
    if (!psa::build_tables(weights, is_max, &t, &err)) {
        psa::set_last_error(err);
        return PSA_ERR_ARG;
    }
    for (int c1 = 0; c1 < psa::kRow; ++c1)
        for (int c2 = 0; c2 < psa::kRow; ++c2) {
            const int dev = c2 * psa::kRow + c1, out = c1 * psa::kRow + c2;
            const bool real = c1 <= psa::kCodeHyphen && c2 <= psa::kCodeHyphen;
            if (P) P[out] = t.pair[dev];
            if (SUB) SUB[out] = t.sub[dev];
            if (DELTA) DELTA[out] = t.rank[dev] ? t.delta_of_rank[t.rank[dev]] : 0.0;
            if (SIGN) SIGN[out] = real ? psa::sign_of(c1, c2) : '\0';
        }
    return PSA_OK;
}

// is_swapable, cuda_funcs.cu:290-307, as a fold over shard records.  Records
// with offset < 0 ("nothing found", cuda_funcs.cu:158-161) never win.
psa_status psa_fold(const psa_result* records, int count, int is_max, psa_result* out)
{
    if (!records || !out || count < 0) {
        psa::set_last_error("psa_fold: bad arguments");
        return PSA_ERR_ARG;
    }
    psa_result best;
    best.score = is_max ? -INFINITY : INFINITY;
    best.offset = -1;
    best.char_offset = -1;
    best.ch = 0;
    best.shard = -1;
    best.pairs = 0;
    int64_t pairs = 0;
    for (int k = 0; k < count; ++k) {
        const psa_result& r = records[k];
        pairs += r.pairs;
        if (r.offset < 0) continue;
        bool take;
        if (best.offset < 0) take = true;
        else if (is_max ? r.score > best.score : r.score < best.score) take = true;
        else if (r.score == best.score)
            take = r.offset < best.offset || (r.offset == best.offset && r.char_offset < best.char_offset);
        else take = false;
        if (take) best = r;
    }
    best.pairs = pairs;
    *out = best;
    return PSA_OK;
}

// cpu_funcs.c:128-133
void psa_shard_range(int total_offsets, int nshards, int rank, int* first_offset, int* last_offset)
{
    if (nshards < 1) nshards = 1;
    const int per = total_offsets / nshards;
    int lo = per * rank;
    int hi = lo + per;
    if (rank == nshards - 1) hi += total_offsets % nshards;
    if (first_offset) *first_offset = lo;
    if (last_offset) *last_offset = hi;
}

// Queries of a batch over shards: blocks as even as possible (no reference rule: the reference has no batch)
void psa_query_block(int nq, int nshards, int rank, int* first_query, int* last_query)
{
    if (nshards < 1) nshards = 1;
    if (nq < 0) nq = 0;
    const int base = nq / nshards, rem = nq % nshards;
    const int lo = base * rank + (rank < rem ? rank : rem);
    const int hi = lo + base + (rank < rem ? 1 : 0);
    if (first_query) *first_query = lo;
    if (last_query) *last_query = hi;
}

int psa_query_block_max(int nq, int nshards)
{
    if (nshards < 1) nshards = 1;
    if (nq < 0) nq = 0;
    return (nq + nshards - 1) / nshards;
}

psa_status psa_batch_unpack(const psa_result* gathered, int nq, int nshards, psa_result* results, psa_result* results_min)
{
    if (!gathered || !results || nq < 1 || nshards < 1) {
        psa::set_last_error("psa_batch_unpack: bad arguments");
        return PSA_ERR_ARG;
    }
    const int per = psa_query_block_max(nq, nshards);
    for (int i = 0; i < nshards; ++i) {
        int lo, hi;
        psa_query_block(nq, nshards, i, &lo, &hi);
        const psa_result* block = gathered + (size_t)i * 2 * per;
        for (int q = lo; q < hi; ++q) {
            results[q] = block[q - lo];
            results[q].shard = q;
            if (results_min) {
                results_min[q] = block[per + (q - lo)];
                results_min[q].shard = q;
            }
        }
    }
    return PSA_OK;
}

void psa_generate_sequence(uint64_t seed, int n, char* out)
{
    for (int i = 0; i < n; ++i) {
        uint64_t z = seed + (static_cast<uint64_t>(i) + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        out[i] = static_cast<char>('A' + (((z >> 32) * 26u) >> 32));
    }
    out[n] = '\0';
}

}  // extern "C"
