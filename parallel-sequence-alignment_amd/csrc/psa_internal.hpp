// psa_internal.hpp — shared declarations between the host-side table builder
// (psa_host.cpp) and the device side (psa_device.hip).  Not installed.
#pragma once

#include <cstdint>
#include <string>

#include "psa.h"

// Build switches.  The product library is built with none of them.  PSA_EXPERIMENT marks a debug build
// (libpsa_debug.so: getenv knobs, fault injection, the test hooks of include/psa_debug.h); every other switch changes
// what the kernels do or measure and is refused unless the build says it is an experiment; retired A/B switches
// (whose branches computed wrong results) are refused always.
#if defined(PSA_EXP_NOLDS) || defined(PSA_EXP_NOWAIT) || defined(PSA_TAIL_TWICE) || defined(PSA_NO_BEST_SPLIT)
#error "PSA_EXP_NOLDS / PSA_EXP_NOWAIT / PSA_TAIL_TWICE / PSA_NO_BEST_SPLIT were timing experiments (HISTORY.md); their branches are gone"
#endif
#if !defined(PSA_EXPERIMENT) && (defined(PSA_PHASE_TIMING) || defined(PSA_RESCORE_EPL) || defined(PSA_PLANES_INLINE))
#error "PSA_PHASE_TIMING / PSA_RESCORE_EPL / PSA_PLANES_INLINE change the kernels: build them as an experiment (make variant V=... EXTRA=\"-DPSA_EXPERIMENT ...\")"
#endif

namespace psa {

// Symbol codes used on the device: 'A'..'Z' -> 0..25, '-' -> 26,
// 27 = padding (scores +0.0, never the best substitution), 255 = invalid.
constexpr int kCodeHyphen = 26;
constexpr int kCodePad    = 27;
constexpr int kCodeBad    = 255;
constexpr int kRow        = 32;   // row stride of every [c2][c1] table
constexpr int kMaxRanks   = 16;   // distinct substitution deltas incl. rank 0 = "no substitution"

// Group lists: cpu_funcs.c:19-20 / cuda_funcs.h:21-22 (README.md:26 lists MSGNDILV for the 4th semi-conservative group;
// the code says SGND — the code is what the reference executes).  constexpr: the host's tables (psa_host.cpp) and the
// count kernels' plane build (psa_filter_kernels.hpp, which turns the relation into straight-line code) read the same lists.
inline constexpr const char* kConservative[] = {"NDEQ", "NEQK", "STA", "MILV", "QHRK", "NHQK", "FYW", "HY", "MILF"};
inline constexpr const char* kSemiConservative[] = {"SAG", "ATV", "CSA", "SGND", "STPA", "STNK",
                                                    "NEQHRK", "NDEQHK", "SNDEQK", "HFY", "FVLIM"};
constexpr int kNumConservative = (int)(sizeof kConservative / sizeof *kConservative);
constexpr int kNumSemiConservative = (int)(sizeof kSemiConservative / sizeof *kSemiConservative);

constexpr bool shares_group(const char* const* groups, int count, int c1, int c2)
{
    for (int g = 0; g < count; ++g) {
        bool has1 = false, has2 = false;
        for (const char* p = groups[g]; *p; ++p) {
            has1 = has1 || *p - 'A' == c1;
            has2 = has2 || *p - 'A' == c2;
        }
        if (has1 && has2) return true;
    }
    return false;
}

// sign of a pair of codes (0..26): get_hashtable_sign (cuda_funcs.cu:424-439) over get_pair_sign (cuda_funcs.cu:495-502)
constexpr char sign_of_codes(int c1, int c2)
{
    if (c1 == kCodeHyphen && c2 == kCodeHyphen) return PSA_SIGN_STAR;
    if (c1 == kCodeHyphen || c2 == kCodeHyphen) return PSA_SIGN_SPACE;
    if (c1 == c2) return PSA_SIGN_STAR;
    if (shares_group(kConservative, kNumConservative, c1, c2)) return PSA_SIGN_COLON;
    if (shares_group(kSemiConservative, kNumSemiConservative, c1, c2)) return PSA_SIGN_DOT;
    return PSA_SIGN_SPACE;
}

inline int encode_symbol(unsigned char c)
{
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c == '-') return kCodeHyphen;
    return kCodeBad;
}

// Everything the kernels need that depends only on (weights, is_max).
// Device tables are indexed [c2][c1] (row = the wave-uniform Seq2 symbol,
// column = the per-lane Seq1 symbol).
struct Tables {
    double        pair[kRow * kRow];     // P: pair score, +0.0 for padding
    unsigned char rank[kRow * kRow];     // 0 = no substitution, else 1..nranks; higher = better
    unsigned char sub[kRow * kRow];      // substituted character (ASCII), 0 = none
    unsigned char cls[kRow * kRow];      // sign class: 0 '*', 1 ':', 2 '.', 3 '_', 4 padding
    double        delta_of_rank[kMaxRanks];  // delta value of each rank; [0] = -INF (max) / +INF (min)
    int           nranks;                // number of real ranks
    // best / worst delta reachable inside each sign class (filter bounds)
    double        class_best_delta[4];
    double        class_worst_delta[4];
    double        weights[4];
    int           is_max;
};

// Builds Tables from the reference's rules.  Returns false (and sets err)
// when a weight is not finite.
bool build_tables(const double weights[4], int is_max, Tables* out, std::string* err);

// sign of a pair of codes (0..26): get_hashtable_sign, cuda_funcs.cu:424-439
char sign_of(int c1, int c2);

void set_last_error(const std::string& msg);

}  // namespace psa
