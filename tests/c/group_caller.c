/* A plain C caller of the device-group API (include/psa.h): what INTEGRATION.md section 4 shows, compiled and run by
 * tests/test_gpu_parity.py::test_group_api_from_c.  The reference's MPI layer (ranks, MPI_Bcast of the sequences,
 * MPI_Allreduce(MAXLOC|MINLOC) + MPI_Send/Recv of the Mutant: cpu_funcs.c:49-94, rank split cpu_funcs.c:128-133) as one
 * process: here N contexts on device 0 with the host exchange (a one-GPU box); with N GPUs: psa_group_create(devs, N, &g).
 *   usage: group_caller <ndev>      prints "ok <offset> <char_offset> <ch> <score>" or a message on stderr, exit 1 */
#include <stdio.h>
#include <stdlib.h>
#include "psa.h"

#define N1 200000
#define M2 3000

static char seq1[N1 + 1], seq2[M2 + 1], queries[5][M2];

static int fail(const char* what)
{
    fprintf(stderr, "%s: %s\n", what, psa_last_error());
    return 1;
}

int main(int argc, char** argv)
{
    const int ndev = argc > 1 ? atoi(argv[1]) : 2;
    const double w[4] = {1.5, 2.6, 0.3, 0.2};
    int devs[16] = {0};
    psa_group* g = NULL;
    psa_context* ctx = NULL;
    psa_result one, sharded, per_device[16], batch[5], batch_min[5], single[5];
    double issue_us = 0.0, total_us = 0.0;
    int q, i;
    if (ndev < 1 || ndev > 16) return 2;
    psa_generate_sequence(11, N1, seq1);
    psa_generate_sequence(12, M2, seq2);
    for (q = 0; q < 5; ++q) {
        static char tmp[M2 + 1];
        psa_generate_sequence(100 + (uint64_t)q, M2, tmp);
        for (i = 0; i < M2; ++i) queries[q][i] = tmp[i];
    }
    /* one context: the answer to compare with */
    if (psa_create(0, NULL, &ctx) != PSA_OK) return fail("psa_create");
    if (psa_set_weights(ctx, w, 1) != PSA_OK || psa_set_sequences(ctx, seq1, N1, seq2, M2) != PSA_OK) return fail("setup");
    if (psa_search(ctx, 0, N1 - M2 + 1, &one) != PSA_OK) return fail("psa_search");
    if (psa_search_batch(ctx, &queries[0][0], 5, M2, M2, 0, N1 - M2 + 1, single) != PSA_OK) return fail("psa_search_batch");
    /* the group */
    if (psa_group_create_ex(devs, ndev, PSA_EXCHANGE_HOST, &g) != PSA_OK) return fail("psa_group_create_ex");
    if (psa_group_size(g) != ndev) return fail("psa_group_size");
    if (psa_group_set_weights(g, w, 1) != PSA_OK || psa_group_set_seq1(g, seq1, N1) != PSA_OK ||
        psa_group_set_seq2(g, seq2, M2) != PSA_OK)
        return fail("group setup");
    for (i = 0; i < 3; ++i)
        if (psa_search_sharded(g, 0, N1 - M2 + 1, &sharded, per_device) != PSA_OK) return fail("psa_search_sharded");
    if (psa_group_last_timing(g, &issue_us, &total_us) != PSA_OK || !(issue_us > 0.0 && issue_us <= total_us)) return fail("timing");
    if (sharded.score != one.score || sharded.offset != one.offset || sharded.char_offset != one.char_offset || sharded.ch != one.ch ||
        sharded.pairs != one.pairs) {
        fprintf(stderr, "sharded search differs from the single context's\n");
        return 1;
    }
    if (psa_search_batch_sharded(g, &queries[0][0], 5, M2, M2, 0, N1 - M2 + 1, batch, batch_min) != PSA_OK)
        return fail("psa_search_batch_sharded");
    for (q = 0; q < 5; ++q)
        if (batch[q].score != single[q].score || batch[q].offset != single[q].offset || batch[q].ch != single[q].ch || batch_min[q].offset < 0) {
            fprintf(stderr, "sharded batch differs from the single context's (query %d)\n", q);
            return 1;
        }
    printf("ok %d %d %c %.17g\n", sharded.offset, sharded.char_offset, (char)sharded.ch, sharded.score);
    psa_group_destroy(g);
    psa_destroy(ctx);
    return 0;
}
