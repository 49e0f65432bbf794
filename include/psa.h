/*
 * psa.h — C ABI of libpsa_amd.so, the MI355X (gfx950) implementation of the
 * reference's GPU search engine.
 *
 * Every entry point is plain C: pointers, sizes, ints and doubles only.  The
 * reference interface each one stands in for is cited as file:line into
 * GuyKabiri/Parallel-Sequence-Alignment.
 *
 * Error model: the legacy symbol keeps the reference's print-and-exit
 * behaviour (cuda_funcs.cu:44-48 pattern); everything else returns a
 * psa_status and leaves a message in psa_last_error().  There is no CPU
 * fallback anywhere in this library: without a usable HIP device every
 * compute call fails with PSA_ERR_DEVICE.
 */
#ifndef PSA_H
#define PSA_H

#include <stddef.h>
#include <stdint.h>
#include "psa_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: only what this header declares (and the reference's C++-mangled
 * name of gpu_run_program, see below) is exported. */
#define PSA_EXPORT __attribute__((visibility("default")))
#pragma GCC visibility push(default)

typedef enum psa_status {
    PSA_OK            = 0,
    PSA_ERR_ARG       = 1,  /* null pointer, empty/negative range, n < m ... */
    PSA_ERR_ALPHABET  = 2,  /* a character outside [A-Z-] (reference quirk 16: undefined there) */
    PSA_ERR_DEVICE    = 3,  /* HIP runtime error or no device */
    PSA_ERR_STATE     = 4   /* call order: sequences / weights not set yet */
} psa_status;

/* Search engines. All give identical answers; they differ in speed. */
typedef enum psa_engine {
    PSA_ENGINE_AUTO       = 0,
    PSA_ENGINE_SEQUENTIAL = 1, /* lane-per-offset, fp64 sum in reference order for every offset */
    PSA_ENGINE_FILTER     = 2  /* exact integer class counts -> candidate set -> sequential re-score */
} psa_engine;

/*
 * One search result.  32 bytes, no padding holes; this is also the record
 * that ranks exchange in the multi-GPU fold (the reference sends
 * {score, pid} through MPI_Allreduce(MAXLOC|MINLOC) and then the Mutant by
 * MPI_Send/Recv, cpu_funcs.c:71-94 — here it is one record).
 */
typedef struct psa_result {
    double  score;        /* total + best substitution delta; -INF (max) / +INF (min) if none */
    int32_t offset;       /* absolute offset into seq1, -1 if none      (Mutant.offset)      */
    int32_t char_offset;  /* index into seq2, -1 if none                (Mutant.char_offset) */
    int32_t ch;           /* substituted character, 0 if none           (Mutant.ch)          */
    int32_t shard;        /* which shard / rank produced it (caller-defined), -1 if unset    */
    int64_t pairs;        /* char pairs scored to produce this record   (metric numerator)   */
} psa_result;

typedef struct psa_context psa_context;

/* ---------------------------------------------------------------------------
 * Legacy drop-in: same name, argument meaning and blocking behaviour as
 *   double gpu_run_program(ProgramData*, Mutant*, int first_offset, int last_offset)
 * (cuda_funcs.h:33, cuda_funcs.cu:6-146).  Searches offsets
 * [first_offset, last_offset), returns the best total+diff, fills
 * returned_mutant with absolute offsets; returns -INFINITY (max) / +INFINITY
 * (min) with {-1,-1,'\0'} when nothing is found (cuda_funcs.cu:143-145,
 * 158-161).  Never writes cpu_data, retains no pointer.  On a device error:
 * message on stderr + exit(EXIT_FAILURE), like the reference.
 *
 * Layout.  A plain symbol cannot see how its caller compiled ProgramData, so the
 * library's gpu_run_program reads the reference's STOCK struct (def.h:35-36:
 * seq1[10001], seq2[5001], 15,048 bytes).  A caller compiled with other limits
 * (this repo's own program uses 1,000,001 / 10,001) must go through
 * gpu_run_program_layout; including this header does that automatically (the
 * inline below), so `gpu_run_program(&data, ...)` is right for whatever
 * PSA_SEQ*_MAX_LEN the including translation unit was built with.  A caller that
 * binds the plain symbol by name (dlsym, ctypes, its own prototype) with another
 * layout is detected where that is possible — no NUL inside seq1[10001] /
 * seq2[5001]: message on stderr and exit(EXIT_FAILURE); an empty seq2 at the stock
 * offset: message on stderr, "nothing found" returned.
 *
 * Linkage.  The reference builds every file as C++ (Makefile:9-12) and declares no
 * extern "C", so its objects refer to the Itanium-mangled
 * _Z15gpu_run_programP5_dataP7_mutantii (cuda_funcs.h:33, call site
 * cpu_funcs.c:180).  The library exports that name too (stock layout, same entry):
 * an unmodified reference cpu_funcs.o links against libpsa_amd.so without an edit
 * (tests/c/stock_caller.cpp).  C callers and this header use the plain C name.
 * ------------------------------------------------------------------------- */

/* For a caller whose ProgramData was compiled with array limits seq1_cap /
 * seq2_cap (its SEQ1_MAX_LEN / SEQ2_MAX_LEN); field offsets are derived from
 * them (is_max@0, weights@8, seq1@40, seq2@40+seq1_cap). */
double gpu_run_program_layout(const void* cpu_data, size_t seq1_cap, size_t seq2_cap,
                              Mutant* returned_mutant, int first_offset, int last_offset);

#if defined(PSA_BUILDING_LIBRARY) || \
    (PSA_SEQ1_MAX_LEN == PSA_REF_SEQ1_MAX_LEN && PSA_SEQ2_MAX_LEN == PSA_REF_SEQ2_MAX_LEN)
/* the exported symbol: stock reference layout */
double gpu_run_program(const ProgramData* cpu_data, Mutant* returned_mutant,
                       int first_offset, int last_offset);
#else
static inline double gpu_run_program(const ProgramData* cpu_data, Mutant* returned_mutant,
                                     int first_offset, int last_offset)
{
    return gpu_run_program_layout(cpu_data, PSA_SEQ1_MAX_LEN, PSA_SEQ2_MAX_LEN, returned_mutant,
                                  first_offset, last_offset);
}
#endif

/* ---------------------------------------------------------------------------
 * Persistent context: owns the device buffers the reference allocates and
 * frees on every call (cuda_funcs.cu:43-64, 122-141).  `stream` is a
 * hipStream_t (or NULL for a stream owned by the context).
 * ------------------------------------------------------------------------- */
psa_status psa_create(int device, void* stream, psa_context** out);
void       psa_destroy(psa_context* ctx);

/* weights[4] in reference order (def.h:12-15), is_max as ProgramData.is_max */
psa_status psa_set_weights(psa_context* ctx, const double weights[4], int is_max);
/* Host sequences (copied to the device; ASCII, not necessarily NUL-terminated).  1 <= n <= PSA_MAX_SEQ1 (2^30, psa_types.h),
 * in every call that takes a Seq1: longer -> PSA_ERR_ARG before anything is read (tested at 2^30 x 10,000) */
psa_status psa_set_seq1(psa_context* ctx, const char* seq1, int n);
psa_status psa_set_seq2(psa_context* ctx, const char* seq2, int m);
/* Device-resident sequences (raw ASCII bytes already in HBM, e.g. a torch
 * uint8 tensor's data_ptr); encoded in place on the context's stream. */
psa_status psa_set_seq1_device(psa_context* ctx, const void* d_seq1, int n);
psa_status psa_set_seq2_device(psa_context* ctx, const void* d_seq2, int m);

/* Both sequences of a search at once: ONE encode launch (and, for host sequences that lie
 * side by side like the two arrays of a ProgramData, one copy command). */
psa_status psa_set_sequences(psa_context* ctx, const char* seq1, int n, const char* seq2, int m);
psa_status psa_set_sequences_device(psa_context* ctx, const void* d_seq1, int n, const void* d_seq2, int m);
/* Host sequences of 256 KB and more (psa_set_sequences, gpu_run_program) are STREAMED: the encode launch goes first and
 * reads every 64 KB chunk over PCIe as soon as the host has copied it into pinned staging, so the launch latency and
 * the host's copy run under the transfer (cuda_funcs.cu:50's cudaMemcpy of the whole ProgramData, here ~30 us per MB
 * instead of ~46).  How many transfers went that way, and how many of those were redone the plain way because the
 * kernel gave up waiting for the host (never expected).  ctx == NULL: the context behind gpu_run_program. */
int psa_streamed_count(psa_context* ctx);
int psa_stream_retry_count(psa_context* ctx);

psa_status psa_set_engine(psa_context* ctx, psa_engine engine);
/* Batches: count launches stay on one stream (default 1: no two launches overlap, a profiler sees each launch's
 * own duration) or alternate between two (2: consecutive launches overlap at their edges). */
psa_status psa_set_batch_streams(psa_context* ctx, int streams);

/* Search offsets [first_offset, last_offset) — the body of gpu_run_program.
 * psa_search blocks; psa_search_async enqueues on the context's stream and
 * psa_search_wait waits for the record and returns it.
 * "Blocks" means: until the RESULT is on the host.  The search's last kernel writes its record into pinned host memory
 * and the library watches that word; the kernel's last instructions (and the stream's completion signal) may follow a
 * few microseconds after psa_search / psa_search_wait / psa_search_device have returned.  Everything the library itself
 * does next is ordered behind it by the context's stream.  A caller that passed device buffers (psa_set_*_device,
 * psa_search_device, psa_search_batch*_device) and wants to overwrite or free them from ANOTHER stream must
 * synchronise with the context's stream first (hipStreamSynchronize, or an event). */
psa_status psa_search(psa_context* ctx, int first_offset, int last_offset, psa_result* out);
psa_status psa_search_async(psa_context* ctx, int first_offset, int last_offset);
psa_status psa_search_wait(psa_context* ctx, psa_result* out);

/* One-shot search from raw ASCII in device memory — the whole of gpu_run_program's device work as ONE kernel launch
 * (no encode launch in front: the count kernel reads the ASCII itself, its last workgroup re-scores the survivors and
 * writes the result).  Blocking.  Replaces the context's sequences (psa_set_seq*): after the call they are unset, or —
 * when the one launch could not settle the search and the call went through psa_set_sequences_device + psa_search —
 * set to these.  Same results and errors as that pair of calls. */
psa_status psa_search_device(psa_context* ctx, const void* d_seq1, int n, const void* d_seq2, int m,
                             int first_offset, int last_offset, psa_result* out);
/* one-shot searches redone through the two-launch path (a character outside [A-Z-], > 16 surviving candidates, ...) */
int psa_raw_retry_count(psa_context* ctx);

/* Batch of nq queries (BASELINE.json configs[4]): nq Seq2 sequences of m characters
 * each, `stride` bytes apart in one buffer (host memory for psa_search_batch, device
 * memory for psa_search_batch_device), all searched over offsets [first,last) of the
 * context's Seq1 with its weights and mode.  results[q] (host memory, shard = q) is
 * what psa_search would return for query q.  Blocking.  Inside the call, query q+1's
 * scoring kernel overlaps query q's candidate re-score and read-back on a second
 * stream.  Replaces the Seq2 set by psa_set_seq2*. */
psa_status psa_search_batch(psa_context* ctx, const char* queries, int nq, int m, size_t stride,
                            int first_offset, int last_offset, psa_result* results);
psa_status psa_search_batch_device(psa_context* ctx, const void* d_queries, int nq, int m, size_t stride,
                                   int first_offset, int last_offset, psa_result* results);

/* MIN and MAX of every query in one pass (configs[4] "MIN+MAX"): what two batches with the
 * context's weights and is_max = 1 / 0 would return (the reference runs its program once per
 * mode: one maximum/minimum line per input.txt, cpu_funcs.c:365).  The sign class of a char
 * pair does not depend on the mode, so both searches of a query share the class-count pass;
 * bounds, candidates and the exact re-score are per mode.  The context's own mode is
 * unchanged by the call. */
psa_status psa_search_batch_minmax(psa_context* ctx, const char* queries, int nq, int m, size_t stride,
                                   int first_offset, int last_offset, psa_result* results_max,
                                   psa_result* results_min);
psa_status psa_search_batch_minmax_device(psa_context* ctx, const void* d_queries, int nq, int m, size_t stride,
                                          int first_offset, int last_offset, psa_result* results_max,
                                          psa_result* results_min);

/* Per-offset view for tests: scores[o-first] = total+diff of offset o,
 * char_offsets / chs the chosen substitution (find_best_mutant_offset,
 * cpu_funcs.c:257-300).  Arrays are host memory of last-first entries; any
 * may be NULL. */
psa_status psa_scores(psa_context* ctx, int first_offset, int last_offset,
                      double* scores, int32_t* char_offsets, int32_t* chs);

/* Health counters of the filter engine: how many searches of this context overflowed the candidate list and were
 * redone by the sequential engine.  (The test hooks psa_debug_counts / psa_debug_rescore are not part of this
 * library: include/psa_debug.h, libpsa_debug.so.) */
int psa_fallback_count(psa_context* ctx);
/* Searches / batches of this context that the sequential engine (20x slower on such shapes) answered because the filter
 * engine's per-chunk counts — a Seq2 of several char chunks over very many offsets — would have exceeded the context's
 * cap on them: 16 GB, half of the device's free memory (looked up by the first search that asks for more than 256 MB),
 * less after an allocation of them has failed (psa_scratch_shrink_count: a device without the memory makes such a
 * search run in pieces or sequentially, not fail).  A blocking psa_search over the cap runs in pieces and is not counted. */
int psa_scratch_fallback_count(psa_context* ctx);
int psa_scratch_shrink_count(psa_context* ctx);
/* Offset shards (a fused search of few tiles under several char chunks) let every chunk workgroup of a tile take a slice
 * of the tile's combine + selection.  A workgroup whose peers do not show up in time (they are not resident: the GPU is
 * shared with other work) leaves, and the tile's last workgroup takes its slice over: this counts those slices — 0 on a
 * GPU the context has to itself.  Synchronises the context's stream; -1 on a device error. */
long long psa_share_sweep_count(psa_context* ctx);
/* A search more than half of whose slices had to be taken over tells the context that its GPU is shared: it stops sharing
 * combines for its next 64 such launches (each then costs what round 4's did) and tries again.  How often that happened. */
int psa_share_pause_count(psa_context* ctx);
/* Filter searches of this context whose candidate selection ran twice: candidates are picked
 * against a running bound while the tiles finish; when that overflows the list, the selection
 * is repeated against the final bound (only a second overflow goes to the sequential engine). */
int psa_reselect_count(psa_context* ctx);
/* Single filter searches that needed a second launch (psa_rescore_kernel): a single search normally ends inside the
 * count kernel, whose last workgroup re-scores up to 16 surviving candidates itself. */
int psa_unfused_count(psa_context* ctx);
/* Batches redone with the per-query count kernel because a wave-group barrier of the batch
 * count kernel gave up (never expected on a healthy device). */
int psa_duo_retry_count(psa_context* ctx);
/* ---------------------------------------------------------------------------
 * Device group: ONE search sharded over several GPUs of this node — the reference's
 * MPI layer (ranks, MPI_Bcast of ProgramData, MPI_Allreduce(MAXLOC|MINLOC) +
 * MPI_Send/Recv of the Mutant: cpu_funcs.c:49-94, split rule cpu_funcs.c:128-133)
 * as one process: one context per device, psa_shard_range split, all shards searched
 * concurrently, ONE RCCL all-gather of the 32-byte psa_result records over xGMI,
 * psa_fold.  librccl is loaded when the first group is created.  The legacy entry
 * points use a group when PSA_DEVICES=0,1,... lists more than one device.
 * ------------------------------------------------------------------------- */
typedef struct psa_group psa_group;
/* How the devices' records meet.  RCCL: one ncclAllGather over xGMI, one rank per GPU (a device may be listed
 * once).  HOST: every device's record is copied to host memory — no RCCL; a device may be listed several times
 * (one context each), which rehearses an N-way split on fewer GPUs (tests do on the one-GPU box). */
#define PSA_EXCHANGE_RCCL 0
#define PSA_EXCHANGE_HOST 1
psa_status psa_group_create(const int* devices, int ndev, psa_group** out);             /* RCCL exchange */
psa_status psa_group_create_ex(const int* devices, int ndev, int exchange, psa_group** out);
void       psa_group_destroy(psa_group* g);
int        psa_group_size(const psa_group* g);
psa_status psa_group_set_weights(psa_group* g, const double weights[4], int is_max);
psa_status psa_group_set_seq1(psa_group* g, const char* seq1, int n);   /* every device gets Seq1 (MPI_Bcast) */
psa_status psa_group_set_seq2(psa_group* g, const char* seq2, int m);
/* Offsets [first_offset,last_offset) over the group.  shard_records (optional, [ndev]) receives the
 * exchanged per-device records (shard = index in the group; offset -1 for an empty shard).  The last kernel of every
 * device's search writes its record into the device's own slot of its receive buffer (the all-gather's in-place form),
 * the all-gather follows on the same stream, device 0 publishes the gathered records in pinned host memory and the
 * host polls their flag word — one wait, with a deadline (psa_group_set_timeout).  The group entry points put the calling thread's current
 * device back before they return. */
psa_status psa_search_sharded(psa_group* g, int first_offset, int last_offset, psa_result* out,
                              psa_result* shard_records);

/* No group call hangs on a device that does not answer (the reference: MPI_Abort, cpu_funcs.c:38-44).  Whatever fails
 * on one device — selecting it, its launch, a copy — is recorded and that device still takes its part in the exchange, so
 * no other device's collective waits for a rank that gave up; and every wait has a deadline: `seconds` (default 60) for
 * device 0's wait for the exchanged records and, with the HOST exchange, for every device's own search.  A call that
 * gives up returns PSA_ERR_DEVICE, psa_last_error() naming the devices that had not answered.  Until the abandoned work
 * has drained from the devices' streams the group's next calls fail at once with PSA_ERR_DEVICE (they never queue behind a
 * stalled device); afterwards the group works as before. */
psa_status psa_group_set_timeout(psa_group* g, double seconds);

/* Host time of the group's last psa_search_sharded / psa_search_batch_sharded, microseconds from the call's entry:
 * issue_us = until the last device's search launch had been enqueued (every device has a persistent launcher thread, so
 * the devices are served side by side, not one after the other; for a batch: until the last device's block — a
 * blocking psa_search_batch on its thread — had returned and its exchange was enqueued), total_us = until the call
 * returned.  Either may be NULL. */
psa_status psa_group_last_timing(const psa_group* g, double* issue_us, double* total_us);

/* A batch of queries over the group (BASELINE.json configs[4] on several GPUs): nq queries of m characters, `stride`
 * bytes apart in host memory, dealt to the devices in contiguous blocks, each block through psa_search_batch (or
 * psa_search_batch_minmax when results_min != NULL: results = MAX, results_min = MIN) on its device, ONE RCCL
 * all-gather of the records.  Needs psa_group_set_weights and psa_group_set_seq1. */
psa_status psa_search_batch_sharded(psa_group* g, const char* queries, int nq, int m, size_t stride,
                                    int first_offset, int last_offset, psa_result* results, psa_result* results_min);

/* Host-only: how nq queries are dealt to nshards shards (psa_search_batch_sharded, bench.py): contiguous blocks as
 * even as possible, the first nq % nshards shards take one query more.  (The reference has no batch; its offset rule
 * — the last rank takes the remainder, psa_shard_range — would leave the last GPU up to nshards-1 extra queries.)
 * psa_query_block_max = the largest block = records per shard and mode in the batch exchange. */
void psa_query_block(int nq, int nshards, int rank, int* first_query, int* last_query);
int  psa_query_block_max(int nq, int nshards);
/* Host-only: the batch exchange's layout.  Every shard contributes a block of 2 * per records, [mode][per] with
 * per = psa_query_block_max (unused entries are padding); `gathered` = the nshards blocks side by side.  Puts mode 0
 * into results[nq] and mode 1 into results_min[nq] (may be NULL) in query order, shard = query index. */
psa_status psa_batch_unpack(const psa_result* gathered, int nq, int nshards, psa_result* results, psa_result* results_min);

/* Fold of per-shard records with the reference's merge order
 * (is_swapable, cuda_funcs.cu:290-307: strictly better score, then lower
 * offset, then lower char_offset).  Host-only, no device needed. */
psa_status psa_fold(const psa_result* records, int count, int is_max, psa_result* out);

/* Offset slice of shard `rank` of `nshards` over `total_offsets`
 * (cpu_funcs.c:128-133: equal slices, the last shard takes the remainder). */
void psa_shard_range(int total_offsets, int nshards, int rank, int* first_offset, int* last_offset);

/* Host-built 27x27 tables (index [c1][c2] with 'A'..'Z' -> 0..25, '-' -> 26,
 * row stride 32): pair score P (get_weight(get_hashtable_sign), cuda_funcs.cu:
 * 424-452), substitute SUB (get_substitute, cuda_funcs.cu:310-421) and
 * DELTA = weight(sign(c1,SUB)) - P (cpu_funcs.c:285).  Any may be NULL. */
psa_status psa_build_tables(const double weights[4], int is_max,
                            double* P, unsigned char* SUB, double* DELTA, char* SIGN);

/* Timing of the last search's kernels, measured with HIP events on the
 * context's stream when profiling is on: which = 0 whole search, 1 dominant
 * kernel.  kernel_name may be NULL. */
psa_status psa_set_profiling(psa_context* ctx, int on);
psa_status psa_last_timing(psa_context* ctx, int which, float* milliseconds, const char** kernel_name);
/* Queries covered by the kernel launch that psa_last_timing(which = 1) timed: 1 for a single
 * search; for a batch run with profiling on, which = 0 is the whole batch and which = 1 its first
 * count launch (one launch serves several queries). */
int psa_last_kernel_queries(psa_context* ctx);
/* Sums over every search completed (psa_search / psa_search_wait) since
 * profiling was switched on: whole-search ms, dominant-kernel ms, count. */
psa_status psa_profile_totals(psa_context* ctx, double* search_ms, double* kernel_ms, int* searches);

/* Deterministic synthetic sequence: n letters uniform over 'A'..'Z'
 * (the distribution of the reference's own generator, main.c:73,81) from a
 * counter-based SplitMix64 stream; NUL-terminates out[n]. */
void psa_generate_sequence(uint64_t seed, int n, char* out);

const char* psa_last_error(void);
const char* psa_version(void);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* PSA_H */
