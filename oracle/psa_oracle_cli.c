/*
 * psa_oracle_cli.c — TEST INFRASTRUCTURE (see psa_oracle.h).
 * Runs the oracle on one record of an input.txt-format file, the way the
 * reference's main does (main.c:13-56 -> initiate_program), and writes an
 * output.txt-format file.
 *   psa_oracle_cli <input> <output> [record=0] [threads=4] [ranks=1]
 * Prints one line: offset char_offset ch score(%.17g) seconds
 */
#include "psa_oracle.h"
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv)
{
    double w[4], score, t0, t1;
    char *s1 = NULL, *s2 = NULL;
    int is_max = 0, record = 0, threads = 4, ranks = 1;
    orc_mutant m;
    if (argc < 3) {
        fprintf(stderr, "usage: %s <input> <output> [record] [threads] [ranks]\n", argv[0]);
        return 1;
    }
    if (argc > 3) record = atoi(argv[3]);
    if (argc > 4) threads = atoi(argv[4]);
    if (argc > 5) ranks = atoi(argv[5]);
    if (!orc_read_record(argv[1], record, w, &s1, &s2, &is_max)) {
        printf("Error reading input file `%s`\n", argv[1]);
        return 2;
    }
    t0 = omp_get_wtime();
    score = orc_search_ranks(s1, s2, w, is_max, ranks, threads, &m);
    t1 = omp_get_wtime();
    if (!orc_write_result(argv[2], s2, &m, score)) {
        printf("Error write to the output file %s\n", argv[2]);
        return 2;
    }
    printf("%d %d %c %.17g %g\n", m.offset, m.char_offset, m.ch ? m.ch : '?', score, t1 - t0);
    free(s1);
    free(s2);
    return 0;
}
