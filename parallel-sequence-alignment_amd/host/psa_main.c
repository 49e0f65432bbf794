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

/* ---- the CPU share of a hybrid run, table-driven ---------------------------------------------
 * What the reference's host threads compute for their offsets (find_best_mutant_cpu / find_best_mutant_offset,
 * cpu_funcs.c:222-300) depends per pair on (c1, c2, weights, mode) only.  The library's host-side table builder
 * (psa_build_tables: the reference's sign, weight and substitution rules) gives the pair score, the substitute and
 * its delta per symbol pair; both sequences are turned into table indices once, and an offset is one pass over
 * `row[i] + col[o + i]`.  The pair scores are added one by one in index order (the order decides how near-tied
 * offsets resolve, cpu_funcs.c:278); a substitution replaces the running choice only when its delta is strictly
 * better (cpu_funcs.c:287-294: the first position wins).  This is the role cpu_funcs.c plays in the reference's
 * program; it lives in this program only — libpsa_amd.so has no CPU search. */
#define PSA_THREADS_COUNT 4            /* THREADS_COUNT, def.h:18 */

typedef struct {
    double score;          /* total + best delta; -INF / +INF when nothing could be substituted */
    int    offset, char_offset;
    char   ch;
} Candidate;

static double        g_score_of[32 * 32], g_delta_of[32 * 32];
static unsigned char g_sub_of[32 * 32];
static int*          g_col;            /* 32 * symbol of seq1[j] */
static int*          g_row;            /* symbol of seq2[i] */

static Candidate no_candidate(int is_max)
{
    Candidate c = {is_max ? -INFINITY : INFINITY, -1, -1, PSA_NO_CHAR};
    return c;
}

/* the reference's total order of results (cuda_funcs.cu:290-307): better score, then lower offset, then lower position */
static int candidate_beats(const Candidate* challenger, const Candidate* holder, int is_max)
{
    if (challenger->score != holder->score) return is_max ? challenger->score > holder->score : challenger->score < holder->score;
    if (challenger->offset != holder->offset) return challenger->offset < holder->offset;
    return challenger->char_offset < holder->char_offset;
}

static Candidate score_one_offset(int is_max, int chars, int offset)
{
    const int* col = g_col + offset;
    Candidate c = no_candidate(is_max);
    double sum = 0.0, best_delta = c.score;
    int i;
    for (i = 0; i < chars; ++i) {
        const int pair = col[i] + g_row[i];
        sum += g_score_of[pair];
        if (g_sub_of[pair] && (is_max ? g_delta_of[pair] > best_delta : g_delta_of[pair] < best_delta)) {
            best_delta = g_delta_of[pair];
            c.char_offset = i;
            c.ch = (char)g_sub_of[pair];
        }
    }
    if (c.ch != PSA_NO_CHAR) {
        c.offset = offset;
        c.score = sum + best_delta;
    }
    return c;
}

static Candidate best_of_offsets(int is_max, int chars, int from, int to)
{
    Candidate best = no_candidate(is_max);
    int o;
    for (o = from; o < to; ++o) {
        Candidate c = score_one_offset(is_max, chars, o);
        if (c.offset < 0) c.offset = o;                 /* (never for sequences over [A-Z-]) */
        if (candidate_beats(&c, &best, is_max) || best.offset < 0) best = c;
    }
    return best;
}

static void index_sequences(const ProgramData* data)
{
    const size_t n = strlen(data->seq1), m = strlen(data->seq2);
    size_t k;
    if (psa_build_tables(data->weights, data->is_max, g_score_of, g_sub_of, g_delta_of, NULL) != PSA_OK) {
        fprintf(stderr, "Failed to build the score tables - %s\n", psa_last_error());
        exit(EXIT_FAILURE);
    }
    free(g_col);
    free(g_row);
    g_col = (int*)malloc(sizeof(int) * (n + 1));
    g_row = (int*)malloc(sizeof(int) * (m + 1));
    if (!g_col || !g_row) { fprintf(stderr, "out of memory\n"); exit(EXIT_FAILURE); }
    for (k = 0; k < n; ++k) {
        const int sym = symbol_index(data->seq1[k]);
        if (sym < 0) { fprintf(stderr, "seq1 contains a character outside [A-Z-]\n"); exit(EXIT_FAILURE); }
        g_col[k] = 32 * sym;
    }
    for (k = 0; k < m; ++k) {
        const int sym = symbol_index(data->seq2[k]);
        if (sym < 0) { fprintf(stderr, "seq2 contains a character outside [A-Z-]\n"); exit(EXIT_FAILURE); }
        g_row[k] = sym;
    }
}

/* The reference's split of one process's offsets (divide_execute_tasks, cpu_funcs.c:123-218): the first
 * `cuda_percentage` percent go to the GPU (thread 0 blocks in gpu_run_program), the rest to the other threads, and
 * the GPU's result stands unless the CPU's is strictly better (cpu_funcs.c:207-214; CPU offsets are the higher
 * ones, so this is also the lowest-offset rule on ties). */
static double run_split(const ProgramData* data, int cuda_percentage, int threads, Mutant* winner)
{
    const int chars = (int)strlen(data->seq2);
    const int offsets = (int)strlen(data->seq1) - chars + 1;
    const int on_gpu = (int)((long long)offsets * cuda_percentage / 100);
    const int on_cpu = offsets - on_gpu;
    const int is_max = data->is_max;
    Candidate gpu = no_candidate(is_max), cpu = no_candidate(is_max);
    Mutant gpu_mutant = {-1, -1, PSA_NO_CHAR};

    if (on_cpu > 0) index_sequences(data);
    if (on_cpu == 0) threads = 1;                                  /* only the GPU's driver thread has work */
    else if (on_gpu > 0 && threads < 2) threads = 2;               /* one drives the GPU, at least one scores */

#pragma omp parallel num_threads(threads)
    {
        const int team = omp_get_num_threads();
        const int me = omp_get_thread_num();
        const int scorers = on_gpu > 0 ? team - 1 : team;          /* thread 0 waits for the GPU (cpu_funcs.c:185-189) */
        const int slot = on_gpu > 0 ? me - 1 : me;
        if (on_gpu > 0 && me == 0) {
            gpu.score = gpu_run_program(data, &gpu_mutant, 0, on_gpu);
        } else if (on_cpu > 0 && slot >= 0 && slot < scorers) {
            const int from = on_gpu + (int)((long long)on_cpu * slot / scorers);
            const int to = on_gpu + (int)((long long)on_cpu * (slot + 1) / scorers);
            const Candidate mine = best_of_offsets(is_max, chars, from, to);
#pragma omp critical
            if (mine.offset >= 0 && (cpu.offset < 0 || candidate_beats(&mine, &cpu, is_max))) cpu = mine;
        }
    }
    if (on_gpu > 0 && !(on_cpu > 0 && (is_max ? cpu.score > gpu.score : cpu.score < gpu.score))) {
        *winner = gpu_mutant;
        return gpu.score;
    }
    winner->offset = cpu.ch != PSA_NO_CHAR ? cpu.offset : -1;
    winner->char_offset = cpu.char_offset;
    winner->ch = cpu.ch;
    return cpu.score;
}

/* ---- the verification view: what the reference prints under DEBUG_PRINT (cpu_funcs.c:382-461) ------------------
 * Five lines under each other — the signs of Seq2 against Seq1 at the winning offset, Seq2, Seq1, the mutated Seq2,
 * its signs — every line but Seq1 indented by the offset, the mutated position in red.  The two scores come from a
 * scorer of their own (pair table from psa_build_tables, plain sum), independent of the search. */
static double signs_and_score(const char* seq1, const char* seq2, int offset, char* signs)
{
    const char* under = seq1 + offset;
    double score = 0.0;
    size_t i = 0;
    for (; seq2[i]; ++i) {
        const int a = symbol_index(under[i]), b = symbol_index(seq2[i]);
        signs[i] = a < 0 || b < 0 ? '?' : g_sign[a * 32 + b];
        if (a >= 0 && b >= 0) score += g_pair[a * 32 + b];
    }
    signs[i] = '\0';
    return score;
}

static void aligned_row(const char* text, int indent, int red)
{
    const int len = (int)strlen(text);
    if (red < 0) red = 0;
    printf("%*s", indent, "");
    if (red < len) printf("%.*s\033[0;31m%c\033[0m%s", red, text, text[red], text + red + 1);
    else printf("%s\033[0;31m\033[0m", text);
    putchar('\n');
}

static void pretty_print(const ProgramData* data, const char* mut, int offset, int char_offset)
{
    static char signs[PSA_SEQ2_MAX_LEN];
    int i;
    if (psa_build_tables(data->weights, data->is_max, g_pair, NULL, NULL, g_sign) != PSA_OK) return;
    printf("\033[0;31m%s\033[0m problem\nWeights: ", data->is_max ? "Maximum" : "Minimum");
    for (i = 0; i < PSA_NUM_WEIGHTS; ++i) printf("%g ", data->weights[i]);
    if (offset < 0) {
        printf("\n\033[0;31mThere are no mutations found\033[0m\n%s\n%s\n", data->seq1, data->seq2);
        return;
    }
    printf("\nOriginal Score: %g\n", signs_and_score(data->seq1, data->seq2, offset, signs));
    aligned_row(signs, offset, char_offset);
    aligned_row(data->seq2, offset, char_offset);
    printf("%s\n", data->seq1);
    aligned_row(mut, offset, char_offset);
    {
        const double after = signs_and_score(data->seq1, mut, offset, signs);
        aligned_row(signs, offset, char_offset);
        printf("Mutation Score: %g\nSeq offset=%3d, Char offset=%3d\n", after, offset, char_offset);
    }
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
    score = run_split(&g_data, cuda_percentage, threads, &mutant);
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
