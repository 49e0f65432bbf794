/*
 * psa_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see psa_oracle.h).
 *
 * A CPU restatement of the reference's CPU search path, written from the
 * behaviour of cpu_funcs.c and cuda_funcs.cu:289-545.  The per-pair work is
 * deliberately kept as heavy as the reference's (alphabet scans inside
 * get_substitute for every char pair) because bench.py times this as the
 * "reference OpenMP path" baseline on the GPU box's host cores.
 */
#include "psa_oracle.h"

#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { NLET = 26 };
static const char STAR = '*', COLON = ':', DOT = '.', BLANK = '_', HYPHEN = '-', NONE = '\0';

/* group lists: cpu_funcs.c:19-20 (identical to cuda_funcs.h:21-22) */
static const char* const CONSERVATIVE[] = {"NDEQ", "NEQK", "STA", "MILV", "QHRK", "NHQK", "FYW", "HY", "MILF"};
static const char* const SEMI_CONSERVATIVE[] = {"SAG", "ATV", "CSA", "SGND", "STPA", "STNK",
                                                "NEQHRK", "NDEQHK", "SNDEQK", "HFY", "FVLIM"};

static char g_sign[NLET][NLET];
static int  g_sign_ready = 0;

static int both_in(const char* group, char a, char b)
{
    return strchr(group, a) != NULL && strchr(group, b) != NULL;
}

/* get_pair_sign, cuda_funcs.cu:495-502: equality first, then conservative,
 * then semi-conservative membership. */
char orc_pair_sign(char a, char b)
{
    size_t g;
    if (a == b) return STAR;
    for (g = 0; g < sizeof CONSERVATIVE / sizeof *CONSERVATIVE; g++)
        if (both_in(CONSERVATIVE[g], a, b)) return COLON;
    for (g = 0; g < sizeof SEMI_CONSERVATIVE / sizeof *SEMI_CONSERVATIVE; g++)
        if (both_in(SEMI_CONSERVATIVE[g], a, b)) return DOT;
    return BLANK;
}

/* fill_hash, cpu_funcs.c:304-318.  The reference fills only the lower
 * triangle and reads [max][min]; the relation is symmetric so a full table
 * is the same function (its out-of-bounds column-26 write is not restated). */
static void sign_table_init(void)
{
#pragma omp critical(orc_sign_init)
    {
        if (!g_sign_ready) {
            int r, c;
            for (r = 0; r < NLET; r++)
                for (c = 0; c < NLET; c++)
                    g_sign[r][c] = orc_pair_sign((char)('A' + r), (char)('A' + c));
            g_sign_ready = 1;
        }
    }
}

/* get_hashtable_sign, cuda_funcs.cu:424-439 */
char orc_table_sign(char c1, char c2)
{
    if (c1 == HYPHEN && c2 == HYPHEN) return STAR;
    if (c1 == HYPHEN || c2 == HYPHEN) return BLANK;
    if (c1 >= 'A' + NLET || c2 >= 'A' + NLET) return NONE;
    if (c1 < 'A' || c2 < 'A') return NONE;
    if (!g_sign_ready) sign_table_init();
    return c1 >= c2 ? g_sign[c1 - 'A'][c2 - 'A'] : g_sign[c2 - 'A'][c1 - 'A'];
}

/* get_weight, cuda_funcs.cu:442-452 */
double orc_weight(char sign, const double* w)
{
    switch (sign) {
    case '*': return w[0];
    case ':': return -w[1];
    case '.': return -w[2];
    case '_': return -w[3];
    }
    return 0;
}

/* get_substitute_by_sign_with_restrictions, cuda_funcs.cu:412-421: first
 * letter (alphabetical) that gives `sign` against `by` and is not
 * conservative-related to the letter being replaced. */
static char first_allowed(char by, char sign, char replaced)
{
    char ch;
    for (ch = 'A'; ch < 'A' + NLET; ch++)
        if (orc_table_sign(by, ch) == sign && orc_table_sign(replaced, ch) != COLON)
            return ch;
    return NONE;
}

/* get_optimal_substitute, cuda_funcs.cu:396-409 */
static char pick(int is_max, double d1, char s1, double d2, char s2)
{
    if ((is_max && d1 >= d2) || (!is_max && d1 <= d2))
        if (s1 != NONE) return s1;
    if (s2 != NONE) return s2;
    return s1;
}

/* get_max_substitute, cuda_funcs.cu:320-345 */
static char max_substitute(char c1, char c2, char sign, const double* w)
{
    double dot_diff, blank_diff;
    if (sign == DOT || sign == BLANK) return c1;      /* becomes '*' */
    if (sign == STAR) {
        dot_diff = -w[0] - w[2];
        blank_diff = -w[0] - w[3];
    } else if (sign == COLON) {
        dot_diff = w[1] - w[2];
        blank_diff = w[1] - w[3];
    } else {
        return NONE;   /* outside [A-Z-]: undefined in the reference (quirk 16) */
    }
    return pick(1, dot_diff, first_allowed(c1, DOT, c2), blank_diff, first_allowed(c1, BLANK, c2));
}

/* get_min_substitute, cuda_funcs.cu:348-393 */
static char min_substitute(char c1, char c2, char sign, const double* w)
{
    const char colon_sub = first_allowed(c1, COLON, c2);
    const char dot_sub = first_allowed(c1, DOT, c2);
    const char blank_sub = first_allowed(c1, BLANK, c2);
    double d1, d2;
    char s1, s2, chosen;

    if (sign == STAR)       { d1 = -w[0] - w[2]; s1 = dot_sub;   d2 = -w[0] - w[3]; s2 = blank_sub; }
    else if (sign == COLON) { d1 = w[1] - w[2];  s1 = dot_sub;   d2 = w[1] - w[3];  s2 = blank_sub; }
    else if (sign == DOT)   { d1 = w[2] - w[1];  s1 = colon_sub; d2 = w[2] - w[3];  s2 = blank_sub; }
    else if (sign == BLANK) { d1 = w[3] - w[1];  s1 = colon_sub; d2 = w[3] - w[2];  s2 = dot_sub;   }
    else return NONE;

    chosen = pick(0, d1, s1, d2, s2);
    if ((sign == DOT || sign == BLANK) && chosen == NONE)
        return c1;          /* '*' is always reachable */
    return chosen;
}

/* get_substitute, cuda_funcs.cu:310-317 */
char orc_substitute(char c1, char c2, const double* w, int is_max)
{
    const char sign = orc_table_sign(c1, c2);
    return is_max ? max_substitute(c1, c2, sign, w) : min_substitute(c1, c2, sign, w);
}

/* is_swapable, cuda_funcs.cu:290-307 */
int orc_is_swapable(const orc_mutant* m1, const orc_mutant* m2, double s1, double s2, int is_max)
{
    if ((is_max && s2 > s1) || (!is_max && s2 < s1)) return 1;
    if (s2 == s1) {
        if (m2->offset < m1->offset) return 1;
        if (m2->offset == m1->offset && m2->char_offset < m1->char_offset) return 1;
    }
    return 0;
}

/* find_best_mutant_offset, cpu_funcs.c:257-300.  The running total is a
 * plain double accumulated in index order — the order is part of the
 * contract (near-tied offsets are separated by their rounding). */
double orc_score_offset(const char* seq1, const char* seq2, const double* w, int is_max,
                        int offset, orc_mutant* mt)
{
    const int len = (int)strlen(seq2);
    double total = 0;
    double best_diff = is_max ? -INFINITY : INFINITY;
    int i;

    mt->offset = -1;
    mt->char_offset = -1;
    mt->ch = NONE;

    for (i = 0; i < len; i++) {
        const char c1 = seq1[offset + i];
        const char c2 = seq2[i];
        const double pair = orc_weight(orc_table_sign(c1, c2), w);
        double diff;
        char sub;

        total += pair;
        sub = orc_substitute(c1, c2, w, is_max);
        if (sub == NONE) continue;
        diff = orc_weight(orc_table_sign(c1, sub), w) - pair;
        if ((is_max && diff > best_diff) || (!is_max && diff < best_diff)) {
            best_diff = diff;
            mt->ch = sub;
            mt->char_offset = i;
            mt->offset = offset;
        }
    }
    if (mt->ch == NONE) return best_diff;
    return total + best_diff;
}

/* get_score_and_signs, cpu_funcs.c:429-441 */
double orc_score_and_signs(const char* seq1, const char* seq2, const double* w, int offset, char* signs)
{
    const int len = (int)strlen(seq2);
    double score = 0;
    int i;
    for (i = 0; i < len; i++) {
        const char s = orc_table_sign(seq1[offset + i], seq2[i]);
        if (signs) signs[i] = s;
        score += orc_weight(s, w);
    }
    if (signs) signs[len] = '\0';
    return score;
}

/* find_best_mutant_cpu, cpu_funcs.c:222-253 (one thread's loop) */
double orc_search_range(const char* seq1, const char* seq2, const double* w, int is_max,
                        int first, int last, orc_mutant* best)
{
    double best_score = is_max ? -INFINITY : INFINITY;
    orc_mutant best_m = {-1, -1, NONE}, cur;
    int o;
    for (o = first; o < last; o++) {
        const double s = orc_score_offset(seq1, seq2, w, is_max, o, &cur);
        if (orc_is_swapable(&best_m, &cur, best_score, s, is_max)) {
            best_m = cur;
            best_m.offset = o;
            best_score = s;
        }
    }
    *best = best_m;
    return best_score;
}

/* The parallel region of divide_execute_tasks, cpu_funcs.c:174-199, with the
 * thread count a parameter instead of THREADS_COUNT (def.h:18). */
double orc_search_omp(const char* seq1, const char* seq2, const double* w, int is_max,
                      int first, int last, int threads, orc_mutant* best)
{
    double shared_score = is_max ? -INFINITY : INFINITY;
    orc_mutant shared_m = {-1, -1, NONE};
    const int span = last - first;

    if (threads < 1) threads = 1;
    if (!g_sign_ready) sign_table_init();

#pragma omp parallel num_threads(threads)
    {
        const int nt = omp_get_num_threads();
        const int tid = omp_get_thread_num();
        const int per = span / nt;
        int lo = first + per * tid;
        int hi = lo + per;
        orc_mutant mine;
        double mine_score;
        if (span % nt != 0 && tid == nt - 1) hi += span % nt;
        mine_score = orc_search_range(seq1, seq2, w, is_max, lo, hi, &mine);
#pragma omp critical(orc_merge)
        {
            if (orc_is_swapable(&shared_m, &mine, shared_score, mine_score, is_max)) {
                shared_score = mine_score;
                shared_m = mine;
            }
        }
    }
    *best = shared_m;
    return shared_score;
}

/* initiate_program's distributed part, cpu_funcs.c:57-94, simulated in one
 * process: rank slices (cpu_funcs.c:128-133), then MAXLOC/MINLOC — best
 * score, lowest rank on equal scores. */
double orc_search_ranks(const char* seq1, const char* seq2, const double* w, int is_max,
                        int nproc, int threads, orc_mutant* best)
{
    const int total = (int)strlen(seq1) - (int)strlen(seq2) + 1;
    const int per = total / nproc;
    double win_score = 0;
    orc_mutant win = {-1, -1, NONE};
    int rank;
    for (rank = 0; rank < nproc; rank++) {
        int lo = per * rank, hi = lo + per;
        orc_mutant m;
        double s;
        if (rank == nproc - 1) hi += total % nproc;
        s = orc_search_omp(seq1, seq2, w, is_max, lo, hi, threads, &m);
        if (rank == 0 || (is_max ? s > win_score : s < win_score)) {
            win_score = s;
            win = m;
        }
    }
    *best = win;
    return win_score;
}

void orc_scores(const char* seq1, const char* seq2, const double* w, int is_max,
                int first, int last, double* scores, int* char_offsets, char* chs)
{
    int o;
    if (!g_sign_ready) sign_table_init();
#pragma omp parallel for schedule(static)
    for (o = first; o < last; o++) {
        orc_mutant m;
        const double s = orc_score_offset(seq1, seq2, w, is_max, o, &m);
        if (scores) scores[o - first] = s;
        if (char_offsets) char_offsets[o - first] = m.char_offset;
        if (chs) chs[o - first] = m.ch;
    }
}

/* ---- input.txt / output.txt ------------------------------------------------ */

static char* read_token(FILE* f)
{
    size_t cap = 64, len = 0;
    char* buf;
    int c;
    do { c = fgetc(f); } while (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v');
    if (c == EOF) return NULL;
    buf = (char*)malloc(cap);
    while (c != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r' && c != '\f' && c != '\v') {
        if (len + 1 >= cap) { cap *= 2; buf = (char*)realloc(buf, cap); }
        buf[len++] = (char)c;
        c = fgetc(f);
    }
    buf[len] = '\0';
    return buf;
}

/* read_seq_and_weights_from_file, cpu_funcs.c:353-368: "%lf %lf %lf %lf",
 * "%s", "%s", "%s"; is_max iff the last token is exactly "maximum". */
int orc_read_record(const char* path, int record, double* w, char** seq1, char** seq2, int* is_max)
{
    FILE* f = fopen(path, "r");
    int k, ok = 0;
    char* tok;
    if (!f) return 0;
    for (k = 0; k < record * 7; k++) {
        tok = read_token(f);
        if (!tok) { fclose(f); return 0; }
        free(tok);
    }
    for (k = 0; k < 4; k++) {
        char* end;
        tok = read_token(f);
        if (!tok) goto done;
        w[k] = strtod(tok, &end);
        if (end == tok) { free(tok); goto done; }
        free(tok);
    }
    *seq1 = read_token(f);
    *seq2 = read_token(f);
    tok = read_token(f);
    if (!*seq1 || !*seq2 || !tok) { free(tok); goto done; }
    *is_max = strcmp(tok, "maximum") == 0 ? 1 : 0;
    free(tok);
    ok = 1;
done:
    fclose(f);
    return ok;
}

/* initiate_program's tail + write_results_to_file, cpu_funcs.c:96-98, 373-378:
 * "<mutated seq2>\n<offset> <score %g>", no trailing newline. */
int orc_write_result(const char* path, const char* seq2, const orc_mutant* mt, double score)
{
    FILE* f = fopen(path, "w");
    char* mut;
    int ok;
    if (!f) return 0;
    mut = strdup(seq2);
    if (mt->char_offset >= 0) mut[mt->char_offset] = mt->ch;
    ok = fprintf(f, "%s\n%d %g", mut, mt->offset, score) > 0;
    free(mut);
    fclose(f);
    return ok;
}
