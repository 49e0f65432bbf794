/*
 * psa_debug.h — test hooks of libpsa_debug.so (the library built with -DPSA_EXPERIMENT).
 *
 * NOT part of the drop-in boundary: libpsa_amd.so exports none of these (nm -D shows the reference's
 * gpu_run_program, its C++-mangled name and the psa_* product calls of psa.h only).  The debug library is the same
 * sources plus these entry points and the getenv experiment / fault-injection knobs the tests use
 * (PSA_SEQ_R, PSA_DUO_FORCE_TIMEOUT, PSA_STREAM_FORCE_TIMEOUT, PSA_RAW_FORCE_TIMEOUT, PSA_SELECT_FROZEN,
 * PSA_RESCORE_WALK, PSA_RESCORE_CHAIN, ...).
 */
#ifndef PSA_DEBUG_H
#define PSA_DEBUG_H

#include "psa.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/* The filter engine's exact per-offset class counts (how many of the m pairs are '*', ':' and '.'): what
 * calc_mutants_scores (cuda_funcs.cu:149-199) would classify pair by pair.  Arrays are host memory of last-first
 * entries; any may be NULL. */
psa_status psa_debug_counts(psa_context* ctx, int first_offset, int last_offset,
                            uint32_t* n_star, uint32_t* n_colon, uint32_t* n_dot);

/* The filter engine's exact re-score (total + best delta, cpu_funcs.c:299) of up to 2048 arbitrary offsets,
 * without the filter in front of it. */
psa_status psa_debug_rescore(psa_context* ctx, const int32_t* offsets, int count, double* scores);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PSA_DEBUG_H */
