/*
 * stock_caller.cpp — the same caller as stock_caller.c, built the way the reference builds its files: as C++
 * (Makefile:9-12 compiles every .c with mpicxx) and with NO extern "C" anywhere, so the call at cpu_funcs.c:180
 * refers to the Itanium-mangled _Z15gpu_run_programP5_dataP7_mutantii (cuda_funcs.h:33).  The reference's own
 * data contract is restated (def.h:35-36, program_data.h:6-11, mutant.h:6-10); none of this repo's headers are
 * included.  Linked against libpsa_amd.so — which exports that mangled name next to the C one — it must print the
 * answer of the reference's output.txt for the bundled input.txt.
 *
 *   stock_caller_cxx <input.txt>     ->  "<offset> <char_offset> <ch> <score %g>" per call
 */
#include <stdio.h>
#include <string.h>

#define SEQ1_MAX_LEN 10000 + 1            /* def.h:35 (unparenthesised there too) */
#define SEQ2_MAX_LEN 5000 + 1             /* def.h:36 */
#define WEIGHTS_COUNT 4

typedef struct _data {                    /* program_data.h:6-11 */
    int    is_max;
    double weights[WEIGHTS_COUNT];
    char   seq1[SEQ1_MAX_LEN];
    char   seq2[SEQ2_MAX_LEN];
} ProgramData;

typedef struct _mutant {                  /* mutant.h:6-10 */
    int  offset;
    int  char_offset;
    char ch;
} Mutant;

/* cuda_funcs.h:33 — C++ linkage, as in the reference */
double gpu_run_program(ProgramData* cpu_data, Mutant* returned_mutant, int first_offset, int last_offset);

int main(int argc, char** argv)
{
    static ProgramData data;
    char mode[16];
    Mutant m;
    FILE* f = argc > 1 ? fopen(argv[1], "r") : NULL;
    if (!f) return 2;
    /* read_seq_and_weights_from_file, cpu_funcs.c:353-368 */
    if (fscanf(f, "%lf %lf %lf %lf", &data.weights[0], &data.weights[1], &data.weights[2], &data.weights[3]) != 4 ||
        fscanf(f, "%10000s", data.seq1) != 1 || fscanf(f, "%5000s", data.seq2) != 1 || fscanf(f, "%15s", mode) != 1)
        return 2;
    fclose(f);
    data.is_max = strcmp(mode, "maximum") == 0;
    if (sizeof(ProgramData) != 15048) return 3;
    const int total = (int)strlen(data.seq1) - (int)strlen(data.seq2) + 1;
    double score = gpu_run_program(&data, &m, 0, total);
    printf("%d %d %c %g\n", m.offset, m.char_offset, m.ch, score);
    score = gpu_run_program(&data, &m, total / 2, total);            /* the upper half, as a rank or a thread share would get */
    printf("%d %d %c %g\n", m.offset, m.char_offset, m.ch, score);
    return 0;
}
