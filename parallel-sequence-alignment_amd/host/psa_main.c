/*
 * psa_main.c — the drop-in program: same files, arguments and stdout lines as
 * the reference's mpiCudaOpenMP binary, with the search done on the MI355X
 * through gpu_run_program (include/psa.h).
 *
 * Mirrors main.c:13-56 and initiate_program, cpu_funcs.c:25-121:
 *   reads ./input.txt  ("%lf %lf %lf %lf" weights, seq1, seq2, maximum|minimum;
 *                       cpu_funcs.c:353-368),
 *   writes ./output.txt ("<mutated seq2>\n<offset> <score %g>", no trailing
 *                       newline; cpu_funcs.c:96-98, 373-378),
 *   prints the three stdout lines (cpu_funcs.c:55,153, main.c:47),
 *   exit code 2 on I/O errors (MPI_Abort(..., 2), cpu_funcs.c:38,44,104,110).
 * argv[1] (CUDA percentage, main.c:30-42) keeps the reference's meaning
 * (divide_execute_tasks, cpu_funcs.c:123-218): the first `percentage` percent
 * of the offsets go to the GPU through gpu_run_program on OpenMP thread 0, the
 * rest to the host threads (a table-driven scorer over psa_build_tables: the
 * same additions in the same order as find_best_mutant_offset,
 * cpu_funcs.c:257-300), the GPU result wins unless the CPU's is strictly better
 * (cpu_funcs.c:207-214).  -100 = one host thread, no GPU (main.c:33-37).  No
 * argument, or one outside [0,100] ("auto", main.c:38-41): every offset runs on
 * the GPU — the reference's 20 % rule (cpu_funcs.c:137-142, a macro-expansion
 * accident: quirk 9) is not reproduced, an MI355X wins from a few 10^4 pairs up
 * (DESIGN.md).  The host scorer lives in THIS program only; libpsa_amd.so has no
 * CPU search path.
 *
 * Extension: `psa_main --all` reads EVERY stacked 7-token
 * record of input.txt (the reference's file holds 15, it only ever reads the
 * first) and writes one "<mutated seq2>\n<offset> <score>" block per record,
 * blocks separated by a newline.
 *
 * `psa_main --table` prints the 27x27 sign table (A..Z and '-') in the layout of
 * the reference's print_hash (cpu_funcs.c:322-349; README.md:105-133 shows it
 * with '_' as blanks) and exits: the table comes from the library's host-side
 * builder (psa_build_tables), no device needed.
 *
 * Extension: `psa_main --pretty [percentage]` also prints the
 * reference's human-checkable view of the answer (pretty_print,
 * cpu_funcs.c:382-426, its DEBUG_PRINT build): mode, weights, the score and
 * sign string of the winning offset before and after the mutation, with the
 * mutated position in red.  The scores of that view come from a second,
 * independent host-side scorer (get_score_and_signs, cpu_funcs.c:430-443)
 * over the library's sign table.
 */
#include <limits.h>
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "psa.h"

static int read_token(FILE* f, char* dst, size_t cap)
{
    size_t len = 0;
    int c;
    do { c = fgetc(f); } while (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v');
    if (c == EOF) return 0;
    while (c != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r' && c != '\f' && c != '\v') {
        if (len + 1 >= cap) return -1;   /* longer than the struct's array: the reference overflows here */
        dst[len++] = (char)c;
        c = fgetc(f);
    }
    dst[len] = '\0';
    return 1;
}

static ProgramData g_data;   /* ~1 MB: kept off the stack */

/* ---- the verification view (pretty_print, cpu_funcs.c:382-461) ---- */
static char   g_sign[32 * 32];
static double g_pair[32 * 32];

static int symbol_index(char c) { return c == PSA_HYPHEN ? 26 : (c >= 'A' && c <= 'Z' ? c - 'A' : -1); }

/* ---- the CPU share of a hybrid run (cpu_funcs.c:222-300), table-driven ----
 * Everything find_best_mutant_offset evaluates per pair depends on (c1, c2, weights, mode) only; the library's
 * host-side table builder (psa_build_tables: the reference's get_hashtable_sign / get_weight / get_substitute
 * rules) gives the pair score P, the substitute SUB and its delta, and this scorer adds them in the reference's
 * order.  It is the reference-side role of cpu_funcs.c in this program, not a fallback of the library. */
#define PSA_THREADS_COUNT 4            /* THREADS_COUNT, def.h:18 */
static double        g_P[32 * 32], g_DELTA[32 * 32];
static unsigned char g_SUB[32 * 32];

/* is_swapable, cuda_funcs.cu:290-307 */
static int is_swapable(const Mutant* m1, const Mutant* m2, double score1, double score2, int is_max)
{
    if ((is_max && score2 > score1) || (!is_max && score2 < score1)) return 1;
    if (score2 == score1) {
        if (m2->offset < m1->offset) return 1;
        if (m2->offset == m1->offset && m2->char_offset < m1->char_offset) return 1;
    }
    return 0;
}

/* find_best_mutant_offset, cpu_funcs.c:257-300 */
static double cpu_best_mutant_offset(const ProgramData* data, int iterations, int offset, Mutant* mt)
{
    double total_score = 0;
    double best_mutant_diff = data->is_max ? -INFINITY : INFINITY;
    int i;
    mt->offset = -1;
    mt->char_offset = -1;
    mt->ch = PSA_NO_CHAR;
    for (i = 0; i < iterations; i++) {
        const int k = symbol_index(data->seq1[offset + i]) * 32 + symbol_index(data->seq2[i]);
        const double pair_score = g_P[k];
        total_score += pair_score;                     /* sequential fp64 sum: the order is part of the contract */
        if (g_SUB[k] == 0) continue;
        if ((data->is_max && g_DELTA[k] > best_mutant_diff) || (!data->is_max && g_DELTA[k] < best_mutant_diff)) {
            best_mutant_diff = g_DELTA[k];
            mt->ch = (char)g_SUB[k];
            mt->char_offset = i;
            mt->offset = offset;
        }
    }
    if (mt->ch == PSA_NO_CHAR) return best_mutant_diff;
    return total_score + best_mutant_diff;
}

/* find_best_mutant_cpu, cpu_funcs.c:222-253 (one thread's slice; the caller merges under a critical section) */
static double cpu_best_mutant_range(const ProgramData* data, int iterations, int first_offset, int last_offset, Mutant* best)
{
    double best_score = data->is_max ? -INFINITY : INFINITY;
    Mutant best_mutant = {-1, -1, PSA_NO_CHAR}, tmp;
    int o;
    for (o = first_offset; o < last_offset; o++) {
        const double score = cpu_best_mutant_offset(data, iterations, o, &tmp);
        if (is_swapable(&best_mutant, &tmp, best_score, score, data->is_max)) {
            best_mutant = tmp;
            best_mutant.offset = o;
            best_score = score;
        }
    }
    *best = best_mutant;
    return best_score;
}

/* divide_execute_tasks for one process, cpu_funcs.c:123-218 */
static double divide_execute_tasks(const ProgramData* data, int cuda_percentage, int threads, Mutant* mt)
{
    const int total_chars = (int)strlen(data->seq2);
    const int total_offsets = (int)strlen(data->seq1) - total_chars + 1;
    const int gpu_tasks = (int)((long long)total_offsets * cuda_percentage / 100);
    const int gpu_first = 0, gpu_last = gpu_tasks;
    const int cpu_tasks = total_offsets - gpu_tasks;
    const int cpu_first = gpu_last, cpu_last = cpu_first + cpu_tasks;
    double gpu_best_score = data->is_max ? -INFINITY : INFINITY;
    double cpu_best_score = data->is_max ? -INFINITY : INFINITY;
    Mutant gpu_mutant = {-1, -1, PSA_NO_CHAR}, cpu_mutant = {-1, -1, PSA_NO_CHAR};

    if (cpu_tasks > 0) {
        size_t i;
        if (psa_build_tables(data->weights, data->is_max, g_P, g_SUB, g_DELTA, NULL) != PSA_OK) {
            fprintf(stderr, "Failed to build the score tables - %s\n", psa_last_error());
            exit(EXIT_FAILURE);
        }
        for (i = 0; data->seq1[i]; ++i)
            if (symbol_index(data->seq1[i]) < 0) { fprintf(stderr, "seq1 contains a character outside [A-Z-]\n"); exit(EXIT_FAILURE); }
        for (i = 0; data->seq2[i]; ++i)
            if (symbol_index(data->seq2[i]) < 0) { fprintf(stderr, "seq2 contains a character outside [A-Z-]\n"); exit(EXIT_FAILURE); }
    }
    if (gpu_tasks > 0 && cpu_tasks == 0) threads = 1;          /* nothing for the other threads to do */
    if (cpu_tasks > 0 && gpu_tasks > 0 && threads < 2) threads = 2;

#pragma omp parallel num_threads(threads)
    {
        int num_threads = omp_get_num_threads();
        int tid = omp_get_thread_num();
        if (gpu_tasks > 0 && tid == 0) gpu_best_score = gpu_run_program(data, &gpu_mutant, gpu_first, gpu_last);
        if (cpu_tasks > 0) {
            if (gpu_tasks > 0) {        /* thread 0 waits for the GPU: the others share the CPU slice (cpu_funcs.c:185-189) */
                num_threads -= 1;
                tid--;
            }
            if (tid != -1) {
                const int per_thread = (cpu_last - cpu_first) / num_threads;
                const int start = cpu_first + per_thread * tid;
                int end = start + per_thread;
                Mutant mine;
                double score;
                if (tid == num_threads - 1) end = cpu_last;             /* the last thread takes the remainder */
                score = cpu_best_mutant_range(data, total_chars, start, end, &mine);
#pragma omp critical
                {
                    if (is_swapable(&cpu_mutant, &mine, cpu_best_score, score, data->is_max)) {
                        cpu_best_score = score;
                        cpu_mutant = mine;
                    }
                }
            }
        }
    }
    /* the GPU result, unless the CPU found a strictly better one (cpu_funcs.c:207-214) */
    *mt = gpu_mutant;
    if (gpu_tasks == 0 || (data->is_max && cpu_best_score > gpu_best_score) || (!data->is_max && cpu_best_score < gpu_best_score)) {
        *mt = cpu_mutant;
        return cpu_best_score;
    }
    return gpu_best_score;
}

/* score and sign string of seq2 under seq1 at `offset` (get_score_and_signs, cpu_funcs.c:430-443) */
static double score_and_signs(const char* seq1, const char* seq2, int offset, char* signs)
{
    double score = 0.0;
    size_t i;
    for (i = 0; seq2[i] != '\0'; ++i) {
        const int a = symbol_index(seq1[offset + i]), b = symbol_index(seq2[i]);
        if (a < 0 || b < 0) { signs[i] = '?'; continue; }
        signs[i] = g_sign[a * 32 + b];
        score += g_pair[a * 32 + b];
    }
    signs[i] = '\0';
    return score;
}

/* the string shifted right by `offset`, the char at char_offset in red (print_with_offset, cpu_funcs.c:446-461) */
static void print_with_offset(const char* chrs, int offset, int char_offset)
{
    const size_t len = strlen(chrs);
    size_t i;
    if (char_offset < 0) char_offset = 0;
    for (i = 0; i < (size_t)offset; ++i) putchar(' ');
    for (i = 0; i < (size_t)char_offset && i < len; ++i) putchar(chrs[i]);
    if ((size_t)char_offset < len) printf("\033[0;31m%c\033[0m", chrs[char_offset]);
    for (i = (size_t)char_offset + 1; i < len; ++i) putchar(chrs[i]);
}

static void pretty_print(const ProgramData* data, const char* mut, int offset, int char_offset)
{
    static char signs[PSA_SEQ2_MAX_LEN];
    double score;
    int i;
    if (psa_build_tables(data->weights, data->is_max, g_pair, NULL, NULL, g_sign) != PSA_OK) return;
    printf("\033[0;31m%s\033[0m problem\n", data->is_max ? "Maximum" : "Minimum");
    printf("Weights: ");
    for (i = 0; i < PSA_NUM_WEIGHTS; ++i) printf("%g ", data->weights[i]);
    if (offset == -1) {
        printf("\n\033[0;31mThere are no mutations found\033[0m\n");
        printf("%s\n%s\n", data->seq1, data->seq2);
        return;
    }
    score = score_and_signs(data->seq1, data->seq2, offset, signs);
    printf("\nOriginal Score: %g\n", score);
    print_with_offset(signs, offset, char_offset);
    printf("\n");
    print_with_offset(data->seq2, offset, char_offset);
    printf("\n");
    printf("%s\n", data->seq1);
    print_with_offset(mut, offset, char_offset);
    printf("\n");
    score = score_and_signs(data->seq1, mut, offset, signs);
    print_with_offset(signs, offset, char_offset);
    printf("\n");
    printf("Mutation Score: %g\n", score);
    printf("Seq offset=%3d, Char offset=%3d\n", offset, char_offset);
}

/* the sign table as print_hash lays it out (cpu_funcs.c:322-349): header, rule, one row per symbol */
static int print_sign_table(void)
{
    static const char symbols[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZ-";
    const double unit[PSA_NUM_WEIGHTS] = {1, 1, 1, 1};     /* the signs do not depend on the weights */
    const int count = (int)(sizeof symbols - 1);
    int r, c;
    if (psa_build_tables(unit, 1, NULL, NULL, NULL, g_sign) != PSA_OK) {
        fprintf(stderr, "%s\n", psa_last_error());
        return 1;
    }
    printf("   ");
    for (c = 0; c < count; ++c) printf(c + 1 < count ? "%c " : "%c\n", symbols[c]);
    printf("   ");
    for (c = 0; c < count; ++c) printf("__");
    printf("\n");
    for (r = 0; r < count; ++r) {
        printf("%c |", symbols[r]);
        for (c = 0; c < count; ++c) printf("%c ", g_sign[symbol_index(symbols[r]) * 32 + symbol_index(symbols[c])]);
        printf("\n");
    }
    return 0;
}

static int run_all_records(void)
{
    FILE* in = fopen(PSA_INPUT_FILE, "r");
    FILE* out;
    static char mut[PSA_SEQ2_MAX_LEN];
    char word[16];
    int records = 0;
    if (!in) {
        printf("Error open input file `%s`\n", PSA_INPUT_FILE);
        return 2;
    }
    out = fopen(PSA_OUTPUT_FILE, "w");
    if (!out) {
        printf("Error open or write to the output file %s\n", PSA_OUTPUT_FILE);
        fclose(in);
        return 2;
    }
    while (fscanf(in, "%lf %lf %lf %lf", &g_data.weights[0], &g_data.weights[1], &g_data.weights[2],
                  &g_data.weights[3]) == 4) {
        Mutant mutant;
        double score;
        int total_offsets;
        if (read_token(in, g_data.seq1, sizeof g_data.seq1) != 1 || read_token(in, g_data.seq2, sizeof g_data.seq2) != 1 ||
            read_token(in, word, sizeof word) != 1)
            break;
        g_data.is_max = strcmp(word, PSA_MAXIMUM_STR) == 0 ? 1 : 0;
        total_offsets = (int)strlen(g_data.seq1) - (int)strlen(g_data.seq2) + 1;
        if (total_offsets < 1 || g_data.seq2[0] == '\0') break;
        score = gpu_run_program(&g_data, &mutant, 0, total_offsets);
        strcpy(mut, g_data.seq2);
        if (mutant.char_offset >= 0) mut[mutant.char_offset] = mutant.ch;
        fprintf(out, "%s%s\n%d %g", records ? "\n" : "", mut, mutant.offset, score);
        records++;
    }
    fclose(in);
    fclose(out);
    if (!records) {
        printf("Error reading input file `%s`\n", PSA_INPUT_FILE);
        return 2;
    }
    printf("records: %d\n", records);
    return 0;
}

int main(int argc, char* argv[])
{
    FILE* in;
    FILE* out;
    char word[16];
    int cuda_percentage = -1;          /* main.c:18 */
    int threads = PSA_THREADS_COUNT;   /* main.c:28 */
    int total_offsets;
    Mutant mutant;
    double score;
    struct timespec t0, t1;
    static char mut[PSA_SEQ2_MAX_LEN];

    int pretty = 0;
    const char* percentage = argc == 2 ? argv[1] : NULL;

    if (argc == 2 && strcmp(argv[1], "--all") == 0) return run_all_records();
    if (argc == 2 && strcmp(argv[1], "--table") == 0) return print_sign_table();
    if (argc >= 2 && strcmp(argv[1], "--pretty") == 0) {
        pretty = 1;
        percentage = argc == 3 ? argv[2] : NULL;
    }
    if (percentage) {                  /* main.c:30-42 */
        cuda_percentage = atoi(percentage);
        if (cuda_percentage == -100) {
            threads = 1;
            cuda_percentage = 0;
        }
        if (cuda_percentage < 0 || cuda_percentage > 100) cuda_percentage = -1;
    }
    if (getenv("PSA_THREADS") && atoi(getenv("PSA_THREADS")) > 0) threads = atoi(getenv("PSA_THREADS"));
    /* auto: the whole search on the GPU (see the header comment; the reference's rule is cpu_funcs.c:137-142) */
    if (cuda_percentage == -1) cuda_percentage = 100;

    in = fopen(PSA_INPUT_FILE, "r");
    if (!in) {
        printf("Error open input file `%s`\n", PSA_INPUT_FILE);
        return 2;
    }
    if (fscanf(in, "%lf %lf %lf %lf", &g_data.weights[0], &g_data.weights[1], &g_data.weights[2],
               &g_data.weights[3]) != 4 ||
        read_token(in, g_data.seq1, sizeof g_data.seq1) != 1 ||
        read_token(in, g_data.seq2, sizeof g_data.seq2) != 1 ||
        read_token(in, word, sizeof word) != 1) {
        printf("Error reading input file `%s`\n", PSA_INPUT_FILE);
        fclose(in);
        return 2;
    }
    fclose(in);
    g_data.is_max = strcmp(word, PSA_MAXIMUM_STR) == 0 ? 1 : 0;

    total_offsets = (int)strlen(g_data.seq1) - (int)strlen(g_data.seq2) + 1;
    if (total_offsets < 1 || g_data.seq2[0] == '\0') {
        printf("Error reading input file `%s`\n", PSA_INPUT_FILE);   /* n < m is undefined in the reference */
        return 2;
    }

    /* "processes" = the reference's MPI ranks (cpu_funcs.c:55): here the GPUs the library shards the offsets
     * over, PSA_DEVICES=0,1,... (one RCCL exchange instead of MPI_Allreduce + Send/Recv, cpu_funcs.c:71-94) */
    {
        int devices = 1;
        const char* list = getenv("PSA_DEVICES");
        if (list && *list) {
            const char* p;
            for (p = list; *p; ++p)
                if (*p == ',') devices++;
        }
        printf("threads=%2d, processes=%2d\n", threads, devices);
    }
    printf("CUDA percentage set to %d\n", cuda_percentage);

    clock_gettime(CLOCK_MONOTONIC, &t0);
    score = divide_execute_tasks(&g_data, cuda_percentage, threads, &mutant);
    clock_gettime(CLOCK_MONOTONIC, &t1);

    strcpy(mut, g_data.seq2);
    if (mutant.char_offset >= 0) mut[mutant.char_offset] = mutant.ch;

    out = fopen(PSA_OUTPUT_FILE, "w");
    if (!out) {
        printf("Error open or write to the output file %s\n", PSA_OUTPUT_FILE);
        return 2;
    }
    if (fprintf(out, "%s\n%d %g", mut, mutant.offset, score) <= 0) {
        printf("Error write to the output file %s\n", PSA_OUTPUT_FILE);
        fclose(out);
        return 2;
    }
    fclose(out);

    printf("total time: %g\n", (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
    if (pretty) pretty_print(&g_data, mut, mutant.offset, mutant.char_offset);
    return 0;
}
