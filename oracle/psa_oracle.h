/*
 * psa_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's CPU search path (cpu_funcs.c +
 * the host-side primitives in cuda_funcs.cu:289-545).  It exists so tests,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg have something to
 * compare the HIP path against.  Nothing in the shipped library includes,
 * links or calls this.
 *
 * Parity pin: the reference ships no unit tests; the oracle is pinned by the
 * fixtures the reference does hold (tests/test_oracle_golden.py):
 *   - input.txt record 0 -> output.txt  (byte-for-byte)
 *   - README.md:105-133 27x27 sign table
 *   - README.md:29-34, 51-70 sign string and alignment scores
 *   - README.md:75-79 substitution rule examples
 * plus the survey-stage known answers.
 * The reference itself is not buildable in this image without writing
 * stand-ins for the CUDA headers it includes by absolute path
 * (cuda_funcs.h:4-5), helper_cuda.h (cuda_funcs.cu:1) and mpi.h, so there is
 * no oracle/_ref build; see DESIGN.md.
 */
#ifndef PSA_ORACLE_H
#define PSA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_mutant {      /* mutant.h:6-10 */
    int  offset;
    int  char_offset;
    char ch;
} orc_mutant;

/* sign of a letter pair from the group lists (get_pair_sign, cuda_funcs.cu:495-502) */
char   orc_pair_sign(char a, char b);
/* table lookup with '-' and range rules (get_hashtable_sign, cuda_funcs.cu:424-439) */
char   orc_table_sign(char c1, char c2);
/* get_weight, cuda_funcs.cu:442-452 */
double orc_weight(char sign, const double* w);
/* get_substitute, cuda_funcs.cu:310-317 (and 320-421 below it) */
char   orc_substitute(char c1, char c2, const double* w, int is_max);
/* is_swapable, cuda_funcs.cu:290-307 */
int    orc_is_swapable(const orc_mutant* m1, const orc_mutant* m2, double s1, double s2, int is_max);

/* find_best_mutant_offset, cpu_funcs.c:257-300 */
double orc_score_offset(const char* seq1, const char* seq2, const double* w, int is_max,
                        int offset, orc_mutant* mt);
/* get_score_and_signs, cpu_funcs.c:429-441 (no mutation) */
double orc_score_and_signs(const char* seq1, const char* seq2, const double* w, int offset, char* signs);

/* find_best_mutant_cpu over [first,last) on one thread, cpu_funcs.c:222-253 */
double orc_search_range(const char* seq1, const char* seq2, const double* w, int is_max,
                        int first, int last, orc_mutant* best);
/* OpenMP split of [first,last) exactly like cpu_funcs.c:174-199 (static
 * contiguous slices, last thread takes the remainder, critical merge) */
double orc_search_omp(const char* seq1, const char* seq2, const double* w, int is_max,
                      int first, int last, int threads, orc_mutant* best);
/* whole program for `nproc` simulated ranks: rank slices (cpu_funcs.c:128-133),
 * per-rank search, MAXLOC/MINLOC fold with lowest-rank ties (cpu_funcs.c:71-94) */
double orc_search_ranks(const char* seq1, const char* seq2, const double* w, int is_max,
                        int nproc, int threads, orc_mutant* best);

/* per-offset dump for parity tests: scores[o-first], char_offsets, chs */
void   orc_scores(const char* seq1, const char* seq2, const double* w, int is_max,
                  int first, int last, double* scores, int* char_offsets, char* chs);

/* input.txt / output.txt (cpu_funcs.c:353-368, 373-378). read returns 1 on
 * success and mallocs both sequences (caller frees); `record` selects the k-th
 * stacked 7-token record (the reference only ever reads record 0). */
int    orc_read_record(const char* path, int record, double* w, char** seq1, char** seq2, int* is_max);
int    orc_write_result(const char* path, const char* seq2, const orc_mutant* mt, double score);

#ifdef __cplusplus
}
#endif
#endif
