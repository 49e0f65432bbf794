// psa_device.hip — gfx950 kernels and the device half of the C ABI.
//
// Engine "sequential" (this file, round 1): one lane per offset (R offsets
// per lane), the char axis walked in index order, the pair score looked up in
// an LDS-resident 32x32 fp64 table and added with a plain v_add_f64 — the
// same additions in the same order as find_best_mutant_offset
// (cpu_funcs.c:271-295), so per-offset totals are bit-identical to the
// reference CPU path and near-tied offsets resolve the same way.
// The substitution delta has at most 10 distinct values per (weights, mode);
// the kernel tracks only the best delta *rank* per offset (integer max), the
// substituted position/character is recovered for the winner alone.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl is loaded at run time, by psa_group_create

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "psa_internal.hpp"
#include "psa_group_workers.hpp"
#ifdef PSA_EXPERIMENT
#include "psa_debug.h"
#endif

namespace psa {

constexpr int kBlock = 256;


struct SearchTail {   // everything the host reads back after a search, one D2H copy
    psa_result result;
    int        bad;       // alphabet flag of the encode kernels (bit 0 seq1, bit 1 seq2)
    int        overflow;  // filter engine: the candidate list did not fit
    int        seq;       // echo of the search's sequence number: proves this tail belongs to it
    int        swept;     // fused searches with a shared combine: the slot's sweep counter (low 32 bits) when the search ended
};

// The tail lives in pinned host memory the host may be POLLING (psa_search_wait does not wait for the kernel's end
// signal): everything else first, then the search's number as a system-scope release store.
__device__ __forceinline__ void publish_tail(SearchTail* out, const psa_result& r, int bad, int overflow, int seq, int swept = 0)
{
    out->result = r;
    out->bad = bad;
    out->overflow = overflow;
    out->swept = swept;
    __hip_atomic_store(&out->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct BlockRec {     // per-block / per-shard arg-best
    double score;
    int    offset;    // INT_MAX = none
    int    rank;
};

// ---------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ bool rec_better(bool is_max, double s_new, int o_new, double s_old, int o_old)
{
    // is_swapable (cuda_funcs.cu:290-307) restricted to (score, offset):
    // strictly better score, or equal score and lower offset.
    if (is_max ? (s_new > s_old) : (s_new < s_old)) return true;
    return s_new == s_old && o_new < o_old;
}

__device__ __forceinline__ void wave_fold(bool is_max, double& s, int& o, int& r)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const double s2 = __shfl_xor(s, d, 64);
        const int o2 = __shfl_xor(o, d, 64);
        const int r2 = __shfl_xor(r, d, 64);
        if (rec_better(is_max, s2, o2, s, o)) { s = s2; o = o2; r = r2; }
    }
}

// Block-wide fold; result valid in thread 0.  scratch: kBlock/64 records.
__device__ __forceinline__ void block_fold(bool is_max, double& s, int& o, int& r, BlockRec* scratch)
{
    wave_fold(is_max, s, o, r);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { scratch[wave].score = s; scratch[wave].offset = o; scratch[wave].rank = r; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            if (rec_better(is_max, scratch[w].score, scratch[w].offset, s, o)) {
                s = scratch[w].score; o = scratch[w].offset; r = scratch[w].rank;
            }
    }
}

}  // namespace psa
#include "psa_filter_kernels.hpp"
namespace psa {

// ---------------------------------------------------------------------------
// sequential engine
// ---------------------------------------------------------------------------
struct SeqArgs {
    const unsigned char* s1;      // codes, padded with kCodePad
    const unsigned int*  s2rows;  // Seq2 symbols << kRowShift, padded to a multiple of 4 with kCodePad
    const double*        pair;    // [32][32] fp64, [c2][c1]
    const unsigned char* rank;    // [32][32]
    const double*        delta_of_rank;  // [kMaxRanks]
    BlockRec*            block_out;      // one record per block
    double*              per_score;      // optional per-offset outputs (index o - first)
    unsigned char*       per_rank;
    int m;        // chars
    int m4;       // m rounded up to a multiple of 4
    int first;    // offset range [first, last)
    int last;
    int base;     // first rounded down to a multiple of 4 (lane 0's first offset)
    int is_max;
};

template <int R>
__global__ __launch_bounds__(kBlock) void psa_seq_kernel(SeqArgs a)
{
    __shared__ double        s_pair[kRow * kRow];
    __shared__ unsigned char s_rank[kRow * kRow];
    __shared__ double        s_delta[kMaxRanks];
    __shared__ BlockRec      s_fold[kBlock / 64];

    for (int k = threadIdx.x; k < kRow * kRow; k += kBlock) {
        s_pair[k] = a.pair[k];
        s_rank[k] = a.rank[k];
    }
    if (threadIdx.x < kMaxRanks) s_delta[threadIdx.x] = a.delta_of_rank[threadIdx.x];
    __syncthreads();

    const int lane_id = blockIdx.x * kBlock + threadIdx.x;
    const int o0 = a.base + R * lane_id;              // this lane's offsets: o0 .. o0+R-1
    const unsigned char* p = a.s1 + o0;

    double acc[R];
    unsigned int best[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc[r] = 0.0; best[r] = 0u; }

    // Window of Seq1 codes for chars i0..i0+3 of all R offsets: bytes
    // o0+i0 .. o0+i0+R+2, held in (d0, d1).
    // (p is only byte-aligned for R = 1, 2: memcpy keeps the 4-byte loads well-defined; one global_load_dword each)
    unsigned int d0;
    __builtin_memcpy(&d0, p, 4);
    for (int i0 = 0; i0 < a.m4; i0 += 4) {
        unsigned int d1;
        __builtin_memcpy(&d1, p + i0 + 4, 4);
        const uint4 rows = *reinterpret_cast<const uint4*>(a.s2rows + i0);   // wave-uniform -> scalar load
        const unsigned long long win = ((unsigned long long)d1 << 32) | d0;
        const unsigned int row[4] = {rows.x >> kRowShift, rows.y >> kRowShift, rows.z >> kRowShift, rows.w >> kRowShift};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const unsigned int c1 = (unsigned int)(win >> (8 * (r + k))) & 0xffu;
                const unsigned int idx = row[k] * kRow + c1;
                acc[r] += s_pair[idx];                       // reference order: i ascending
                const unsigned int rk = s_rank[idx];
                best[r] = rk > best[r] ? rk : best[r];
            }
        }
        d0 = d1;
    }

    // per-offset score = total + best delta (cpu_funcs.c:297-299)
    const bool is_max = a.is_max != 0;
    double my_s = is_max ? -INFINITY : INFINITY;
    int my_o = INT_MAX, my_r = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int o = o0 + r;
        if (o >= a.first && o < a.last) {
            const double s = best[r] ? acc[r] + s_delta[best[r]] : s_delta[0];
            if (a.per_score) a.per_score[o - a.first] = s;
            if (a.per_rank) a.per_rank[o - a.first] = (unsigned char)best[r];
            if (rec_better(is_max, s, o, my_s, my_o)) { my_s = s; my_o = o; my_r = (int)best[r]; }
        }
    }
    block_fold(is_max, my_s, my_o, my_r, s_fold);
    if (threadIdx.x == 0) {
        a.block_out[blockIdx.x].score = my_s;
        a.block_out[blockIdx.x].offset = my_o;
        a.block_out[blockIdx.x].rank = my_r;
    }
}

// Folds the per-block records, then recovers the winner's substitution:
// the first i whose delta rank equals the offset's best rank
// (cpu_funcs.c:287-294: strict compare keeps the lowest i).
struct FinalArgs {
    const BlockRec*      block_in;
    int                  nblocks;
    const unsigned char* s1;
    const unsigned int*  s2rows;
    const unsigned char* rank;
    const unsigned char* sub;
    SearchTail*          out;
    const int*           bad;
    int m;
    int is_max;
    int seq;
    long long pairs;
    psa_result* record_out;      // device groups: the record also goes to device memory (shard = record_shard); or NULL
    int record_shard;
};

__device__ __forceinline__ void finalize_write(const FinalArgs& a, psa_result r)
{
    r.shard = -1;
    r.pairs = a.pairs;
    if (a.record_out) {
        psa_result x = r;
        x.shard = a.record_shard;
        *a.record_out = x;
    }
    publish_tail(a.out, r, *a.bad, 0, a.seq);
}

__global__ __launch_bounds__(kBlock) void psa_finalize_kernel(FinalArgs a)
{
    __shared__ BlockRec s_fold[kBlock / 64];
    __shared__ BlockRec s_win;
    __shared__ int s_first_i;
    const bool is_max = a.is_max != 0;

    double s = is_max ? -INFINITY : INFINITY;
    int o = INT_MAX, r = 0;
    for (int k = threadIdx.x; k < a.nblocks; k += kBlock) {
        const BlockRec b = a.block_in[k];
        if (rec_better(is_max, b.score, b.offset, s, o)) { s = b.score; o = b.offset; r = b.rank; }
    }
    block_fold(is_max, s, o, r, s_fold);
    if (threadIdx.x == 0) {
        s_win.score = s; s_win.offset = o; s_win.rank = r;
        s_first_i = INT_MAX;
    }
    __syncthreads();
    const BlockRec w = s_win;
    if (w.offset == INT_MAX || w.rank == 0) {
        if (threadIdx.x == 0) {
            psa_result r;
            r.score = is_max ? -INFINITY : INFINITY;
            r.offset = -1;
            r.char_offset = -1;
            r.ch = 0;
            finalize_write(a, r);
        }
        return;
    }
    int mine = INT_MAX;
    for (int i = threadIdx.x; i < a.m; i += kBlock) {
        const unsigned int idx = (a.s2rows[i] >> kRowShift) * kRow + a.s1[w.offset + i];
        if (a.rank[idx] == w.rank) { mine = i; break; }
    }
    atomicMin(&s_first_i, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int i = s_first_i;
        const unsigned int idx = (a.s2rows[i] >> kRowShift) * kRow + a.s1[w.offset + i];
        psa_result r;
        r.score = w.score;
        r.offset = w.offset;
        r.char_offset = i;
        r.ch = a.sub[idx];
        finalize_write(a, r);
    }
}

// test view: substitution of every offset (first i of the offset's best rank)
__global__ void psa_first_index_kernel(const unsigned char* __restrict__ s1, const unsigned int* __restrict__ s2rows,
                                       const unsigned char* __restrict__ rank, const unsigned char* __restrict__ sub,
                                       const unsigned char* __restrict__ per_rank, int first, int last, int m,
                                       int* __restrict__ char_offsets, int* __restrict__ chs)
{
    const int o = first + blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= last) return;
    const unsigned int want = per_rank[o - first];
    int found = -1, ch = 0;
    if (want)
        for (int i = 0; i < m; ++i) {
            const unsigned int idx = (s2rows[i] >> kRowShift) * kRow + s1[o + i];
            if (rank[idx] == want) { found = i; ch = sub[idx]; break; }
        }
    char_offsets[o - first] = found;
    chs[o - first] = ch;
}

}  // namespace psa

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
using psa::set_last_error;

#define PSA_HIP(call)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            set_last_error(std::string(#call) + ": " + hipGetErrorString(e_));                 \
            return PSA_ERR_DEVICE;                                                             \
        }                                                                                      \
    } while (0)

#define PSA_TRY(expr)                                                                          \
    do {                                                                                       \
        psa_status st_ = (expr);                                                               \
        if (st_ != PSA_OK) return st_;                                                         \
    } while (0)

namespace {

// debug library, PSA_TRACE_INIT=1: where a process's first call spends its time (tools/cold_start.py)
#ifdef PSA_EXPERIMENT
void init_trace(const char* stage)
{
    static const bool on = std::getenv("PSA_TRACE_INIT") != nullptr;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    static auto last = t0;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[psa init] %9.1f us  (+%9.1f)  %s\n", std::chrono::duration<double, std::micro>(now - t0).count(),
                 std::chrono::duration<double, std::micro>(now - last).count(), stage);
    last = now;
}
#else
inline void init_trace(const char*) {}
#endif

// debug library, PSA_TRACE_SEARCH=1: host-side stamps of one blocking search (printed by psa_search_wait)
#ifdef PSA_EXPERIMENT
struct SearchTrace {
    bool on = std::getenv("PSA_TRACE_SEARCH") != nullptr;
    std::chrono::steady_clock::time_point t[8];
    void stamp(int k) { if (on) t[k] = std::chrono::steady_clock::now(); }
    void print() const
    {
        if (!on) return;
        auto us = [&](int a, int b) { return std::chrono::duration<double, std::micro>(t[b] - t[a]).count(); };
        std::fprintf(stderr, "[psa search] entry->planned %.2f, ->args built %.2f, launch call %.2f, ->wait entered %.2f, tail seen after %.2f, settle %.2f us\n",
                     us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6));
    }
};
SearchTrace g_search_trace;
#define PSA_STRACE(k) g_search_trace.stamp(k)
#define PSA_STRACE_PRINT() g_search_trace.print()
#else
#define PSA_STRACE(k) do { } while (0)
#define PSA_STRACE_PRINT() do { } while (0)
#endif

// The library selects the device of the context it works on; an entry point puts the caller's current device back
// when it returns (a host program with its own HIP / torch work on another GPU keeps launching there).
struct DeviceScope {
    int prev = -1;
    DeviceScope() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

constexpr int kLdsLimit = 160 * 1024 - 1024;   // dynamic LDS a workgroup may ask for (the kernels keep < 1 KB static)
constexpr int kCandCap = 2048;        // candidates re-scored per search before falling back
constexpr int kSeq1Slack = 49152;     // code-27 tail behind Seq1: tile + chunk rounding + staged-window over-read
constexpr int kSeq2Slack = 32768 + 64;
// The filter engine splits a long Seq2 into char chunks (grid.y).  A chunk workgroup stages kTile / 32 + class_chars + 1
// slots per plane row of kRowSlots: class_chars <= 383, i.e. 11 rounds of 32 chars per shift class = 11,264 chars per chunk
// (the waves' counters would hold 15 rounds).  Up to 186 chunks = 2,095,104 chars: the per-offset counts travel as 21-bit
// fields.  Longer: the sequential engine.
constexpr int kMaxChunks = 186;
constexpr int kChunkRoundsMax = std::min(psa::kMaxWaveChars / 2 / psa::kUnroll, (psa::kRowSlots - psa::kTile / 32 - 1) / psa::kUnroll);
static_assert(kChunkRoundsMax == 11 && (long long)kMaxChunks * kChunkRoundsMax * psa::kClasses * psa::kUnroll < (1 << 21), "21-bit counts");
// room behind a Seq2 of m symbols: the class-order buffer holds nchunks * 1024 * ceil(rounds / nchunks) <= 1024 * (rounds +
// nchunks) symbols (rounds = m in units of 1024, rounded up) + one round of look-ahead, whatever chunk count
// (<= min(kMaxChunks, rounds)) the planner picks.  Up to 31 rounds (31,744 chars) this is the fixed kSeq2Slack.
inline int seq2_slack(int m)
{
    const int rounds = ((m + psa::kClasses - 1) / psa::kClasses + psa::kUnroll - 1) / psa::kUnroll;
    const int chunks = std::min(kMaxChunks, rounds);
    return std::max(kSeq2Slack, (chunks + 1) * psa::kClasses * psa::kUnroll + psa::kUnroll + 64);
}
constexpr int kSlots = 2;             // searches in flight inside a batch

struct Scalars {                      // small device-side state of one filter search (zero between searches)
    unsigned long long global_best;   // ordered-u64 of the best pessimistic bound seen so far
    int cand_count;
    int done;                         // candidate workgroups that have finished re-scoring
};

struct FilterPlan {
    int waves = 0;        // waves per workgroup (16)
    int nchunks = 0;      // char chunks (grid.y)
    int class_chars = 0;  // chars of every shift class per chunk (a chunk covers 32 x class_chars chars)
    int lds_slots = 0;
    int lds_bytes = 0;
    int ntiles = 0;       // offset tiles (grid.x)
};

template <typename T>
psa_status dev_alloc(T** ptr, size_t count)
{
    if (*ptr) PSA_HIP(hipFree(*ptr));
    *ptr = nullptr;
    PSA_HIP(hipMalloc(reinterpret_cast<void**>(ptr), sizeof(T) * count));
    return PSA_OK;
}

// Everything one search writes.  The single-search API uses slot 0; a batch
// alternates between the slots so the tail of query q (re-score, read-back)
// overlaps the count kernel of query q+1.
struct SearchSlot {
    int m = -1, m_cap = 0;
    unsigned int*       d_s2rows = nullptr;    // owned: Seq2 codes as u32 (sequential engine: scalar loads)
    unsigned char*      d_s2bytes = nullptr;   // owned: Seq2 codes as bytes (filter engine)
    unsigned int*       d_s2cls = nullptr;     // owned: Seq2 codes as u32 in the count kernels' class order
    int                 cls_chars = 0, cls_chunks = 0;   // the chunking d_s2cls is laid out for
    const unsigned int*  s2rows = nullptr;     // the codes this slot searches: its own buffers, or one
    const unsigned char* s2bytes = nullptr;    // query of the batch buffer
    const unsigned int*  s2cls = nullptr;
    bool                inline_select = false; // the count kernel picked the candidates itself
    bool                scal_dirty = false;    // a filter search was enqueued and its self-reset not confirmed yet
    psa::BlockRec*      d_blocks = nullptr;    int blocks_cap = 0;      // sequential engine
    unsigned long long* d_partial = nullptr;   size_t partial_cap = 0;   // [nchunks][span_pad] counts of a multi-chunk search
    int*                d_arrivals = nullptr;  int arrivals_cap = 0;     // [ntiles], zero between searches
    // shared combine of a fused multi-chunk search: per tile a "ready" tag, per (tile, chunk) a "claimed" tag (launch numbers)
    unsigned long long* d_share = nullptr;
    unsigned long long  share_launches = 0;      // sharing launches on this slot so far: the next one's tag
    Scalars*            d_scal = nullptr;
    unsigned long long* d_best_split = nullptr;   // kBestSplit words of the running bound, a cache line each (zero between searches)
    int*                d_cand = nullptr;
    double*             d_cand_upper = nullptr;
    psa::CandRec*       d_rec = nullptr;
    psa::SearchTail*    d_tail = nullptr;      // not owned: the search's entry of the context's host-mapped tails
    hipEvent_t          front_done = nullptr;  // count + select enqueued
    hipEvent_t          back_done = nullptr;   // re-score + read-back enqueued
};

}  // namespace

struct psa_context {
    int         device = 0;
    hipStream_t stream = nullptr;      // main stream (caller's or owned)
    hipStream_t aux = nullptr;         // owned: tails of batched searches
    hipStream_t twin = nullptr;        // owned: every other count kernel of a pooled batch (kernels overlap at their edges)
    int batch_streams = 1;             // count launches of a batch on one stream (2: alternate between two, psa_set_batch_streams)
    int batch_rescore_grid = 8;        // re-score workgroups per query inside a batch (the CUs are busy counting)
    // dynamic-LDS grants of the count kernel's instances on this context's device: [raw][chunked]
    bool granted16[2][2] = {{false, false}, {false, false}};
    // Shares of a SIMD's four waves in the single-search count kernel, by age (oldest first), in permille.  The hardware
    // issues the oldest wave of a SIMD first: with equal shares it finished 25 us before the youngest and the SIMD ended on
    // one wave that cannot fill its issue slots.  Round 3, first form (every wave converted its own counts, the older ones
    // early, under the younger waves' loops), kernel us at 1M x 10k: 2/2/2/2 classes 87.8, 3/2/2/1 81.3, 2/3/2/1 84.5,
    // 3/3/1/1 86.4, 3/3/2/0 86.0, 1/2/2/3 99.0; in rounds of 32 chars 340/270/250/140 another 1.7 us.  Since the waves'
    // counters are summed as bit planes behind a barrier (sum_dump / sum_take / sum_columns) nobody has conversion work to
    // hide and the optimum is flat: same box, warm, kernel us: 350/270/240/140 76.8-77.3, 360/260/240/140 77.2,
    // 320/270/250/160 77.3-77.5, 330/270/250/150 77.4, 340/270/250/140 77.8-78.3, 330/280/240/150 78.8-79.4,
    // 310/280/260/150 79.1; another box: 300/270/250/180 81.0 and 250/250/250/250 83.9 against 78.2
    // (tools/deal_sweep.py, gpurun_out/deal_sweep_r5l.txt, _r5j.txt).
    int  single_deal[4] = {350, 270, 240, 140};
    // Batch kernel: share of a group's four older waves (the younger ones take the rest).  Its waves run with the younger
    // ones at the higher priority, so THEY finish first: measured per 256-query step 500 -> 15.27 ms, 375 -> 14.08 ms,
    // 250 -> 16.3 ms, 625 -> slower still; per 32-query launch 375 2.006 ms, 350 2.04, 400 2.07, 325 2.09, 425 2.10, 450 2.11
    // (shares that split a class pay a counter shift and a restart more)
    int  duo_older_permille = 375;
    int  duo_prio_mode = 0;            // debug library: other s_setprio schemes of the batch kernel's main loop
    bool granted_duo = false;
    bool duo = true;                   // batches: one workgroup per tile serves several queries (psa_count_duo_kernel)
    int  duo_queries = 128;            // queries per launch of it (measured: 32 -> 128 per launch -1.1 % per 256-query batch;
                                       // the re-score of a launch still overlaps the next launch's counting)
    int  duo_skew_ticks = 1300;        // 100 MHz ticks group 1 starts late (one epilogue)
    bool duo_force_timeout = false;    // debug library only: the group barriers of the batch kernel give up at once
    int  duo_retries = 0;              // batches redone without the batch kernel after a group barrier timed out
    bool granted_rescore = false;
    bool rescore_pad = true;
    bool        own_stream = false;

    // Everything of fixed size lives in TWO allocations made by psa_create (round 4; before: ~25 hipMalloc / hipHostMalloc
    // calls and 3 streams, of which a single gpu_run_program needs one): a device arena — score tables of both modes,
    // symbol planes, alphabet flag, control words, the slots' search scalars, candidate lists and records — zeroed by
    // one fill, and a pinned host-coherent arena (control word, the single search's result tail, the flag's host copy).
    // The second and third stream, the inter-stream events and the timing events are created by the first batch /
    // psa_set_profiling(1) (ensure_batch_objects, ensure_timing_events).
    unsigned char* d_arena = nullptr;
    unsigned char* h_arena = nullptr;     // pinned, mapped, coherent
    bool        tails_in_arena = true;

    psa::Tables tables;
    bool        have_weights = false;
    int         n = -1;
    int         n_cap = 0;

    unsigned char* d_raw = nullptr;  size_t raw_cap = 0;   // staging for host sequences
    // pinned, host-coherent staging of a search's two sequences: the host copies them in (85 GB/s), the encode kernel reads
    // them over PCIe itself — 12 + 25 us for 1 MB instead of the 46 us of a pageable hipMemcpyAsync (tools/ubench/h2d.hip)
    unsigned char* h_stage = nullptr;  size_t stage_cap = 0;
    unsigned char* hd_stage = nullptr;    // its device-side alias
    size_t         stage_second = 0;
    // Long sequences are STREAMED through it (stream_pair): the encode launch goes first and reads each chunk as soon as
    // the host has copied and published it in this control word (pinned, host-coherent; psa_encode_pair_streamed_kernel)
    unsigned long long* h_ctl = nullptr;
    unsigned long long* hd_ctl = nullptr;
    unsigned long long* d_ctl = nullptr;  // device memory: the poller workgroup forwards the host's word to it
    unsigned       stream_ticket = 0;
    bool           stream_force_timeout = false;   // debug library only: the host publishes nothing, the kernel gives up at once
    int            streamed = 0;          // transfers that went that way
    int            stream_retries = 0;    // ... and had to be redone the plain way (the kernel gave up waiting for the host)
    bool           stream_gave_up = false; // report_bad saw that flag: the caller of stream_pair redoes the transfer
    unsigned char* d_s1 = nullptr;        // Seq1 codes, code-27 padded to n_cap
    int            nslots = 0;            // 32-symbol slots of d_s1
    double*        d_pair = nullptr;
    unsigned char* d_rank = nullptr;
    unsigned char* d_sub = nullptr;
    double*        d_delta = nullptr;
    unsigned char* d_clsrank = nullptr;   // class | rank << 3 per pair (rescore kernel)
    // the same weights in the opposite mode (MIN+MAX batches); built on first use
    psa::Tables    alt_tables;
    bool           have_alt = false;
    unsigned char* d_alt_rank = nullptr;
    unsigned char* d_alt_sub = nullptr;
    double*        d_alt_delta = nullptr;
    unsigned char* d_alt_clsrank = nullptr;
    int*           d_bad = nullptr;
    int*           h_bad = nullptr;       // pinned
    // pinned, host-coherent, one per search of a batch (>= 1).  The kernels write the 48-byte tail
    // straight into it (hd_tails = the device-side alias): no D2H copy command per search.
    psa::SearchTail* h_tails = nullptr;   int tails_cap = 0;
    psa::SearchTail* hd_tails = nullptr;
    unsigned int*  d_batch_rows = nullptr;   // codes of all queries of a batch
    unsigned int*  d_batch_cls = nullptr;    // the same in the count kernels' class order
    unsigned char* d_batch_bytes = nullptr;  size_t batch_cap = 0;
    // per-query search state of a batch (no buffer is reused inside a batch, so the main
    // stream never has to wait for the re-score stream)
    Scalars*         d_pool_scal = nullptr;
    int*             d_pool_cand = nullptr;
    double*          d_pool_cand_upper = nullptr;
    psa::CandRec*    d_pool_rec = nullptr;
    int              pool_cap = 0;
    std::vector<hipEvent_t> pool_events;
    // count-kernel scratch of multi-chunk plans inside a batch: kernels q and q+2 share a stream, so two sets do
    unsigned long long* d_ring_partial[2] = {nullptr, nullptr};   size_t ring_partial_cap = 0;
    int*                d_ring_arrivals[2] = {nullptr, nullptr};  int ring_arrivals_cap = 0;

    SearchSlot  slot[kSlots];

    psa_engine  engine = PSA_ENGINE_AUTO;
    int         forced_r = 0;

    bool        profiling = false;
    hipEvent_t  ev[4] = {nullptr, nullptr, nullptr, nullptr};   // search start/stop, kernel start/stop
    bool        timing_valid = false;
    const char* last_kernel = "";
    bool        pending = false;
    bool        pending_filter = false;      // the search in flight used the filter engine
    int         pending_first = 0, pending_last = 0;
    double      sum_search_ms = 0.0, sum_kernel_ms = 0.0;   // over profiled searches since profiling was switched on
    int         profiled = 0;
    int         last_kernel_queries = 1;     // queries covered by the launch timed between ev[2] and ev[3]
    int         fallbacks = 0;               // filter searches redone by the sequential engine
    int         reselects = 0;               // filter searches whose selection ran twice (overflow against the running bound)
    int         unfused = 0;                 // single filter searches finished by psa_rescore_kernel, not inside the count kernel
    int         raw_retries = 0;             // one-shot searches (psa_search_device) redone through the two-launch path
    // one-shot searches: arrival counter of the converter workgroups that write Seq2's class order in front of the tiles
    // (monotonic; raw_arrived = what it holds when no launch is in flight), and how many of them a launch may have
    unsigned*   d_raw_arrivals = nullptr;
    unsigned    raw_arrived = 0;
    int         raw_converters = 16;         // at most (as many as the tiles leave CUs free)
    bool        raw_force_timeout = false;   // debug library only: the tiles wait for one converter more than there is
    bool        select_frozen = false;       // debug library only (PSA_SELECT_FROZEN): the first selection of every search runs
    bool        select_frozen_next = false;  // against no bound at all, so the overflow -> re-selection path is taken for sure
    int         cus = 256;                   // CUs of the device
    int         seq = 0;                     // sequence number of the last search enqueued
    // device groups: the final kernel of a single search also writes its 32-byte record here (device memory)
    psa_result* record_out = nullptr;
    int         record_shard = -1;
    // bytes of per-chunk counts one search may take: at most 16 GB, and at most half of what the device had free when the
    // context was created; lowered when an allocation of them fails (reserve_chunk_scratch).  Debug library: PSA_SCRATCH_CAP
    size_t      scratch_cap = (size_t)16 << 30;
    // A shared GPU makes the sharing a loss (the tiles' workgroups are not resident together: +8 us per search): a search
    // more than half of whose slices had to be taken over pauses the sharing for the context's next kSharePause launches
    int         share_pause = 0;
    unsigned    share_swept_seen = 0;        // the sweep counter's low 32 bits after the last sharing search
    int         pending_share_slices = 0;    // slices other than the last arrivers' own of the sharing search in flight (0: none)
    int         share_pauses = 0;            // how often that happened
    bool        share_may_pause = true;      // debug library: PSA_SHARE_PAUSE=0 keeps sharing whatever happens (the sweep path's tests)
    bool        share_combine = true;        // fused multi-chunk searches whose workgroups fit the CUs share the tiles' combine (debug library: PSA_SHARE=0)
    int         share_spins = psa::kShareSpins;   // debug library: PSA_SHARE_SPINS=0 makes every waiting workgroup leave at once (the last arriver sweeps)
    bool        scratch_cap_fitted = false;  // the device's free memory has been looked at (first request beyond 256 MB)
    int         scratch_fallbacks = 0;       // searches / batches the sequential engine answered because their chunk counts exceed the cap
    int         scratch_shrinks = 0;         // times the cap was lowered after a failed allocation
    // ... and the kernels that finish the queries of a batch write theirs here ([0]: the results / MAX block, [1]: the MIN
    // block of a MIN+MAX batch; device memory, query q at + q).  batch_records_final: the batch ended without the host
    // touching a result (no re-selection, no sequential-engine fallback), so the device blocks are what the host holds.
    psa_result* batch_records[2] = {nullptr, nullptr};
    bool        batch_records_final = false;
};

namespace {

psa_status ensure_device(psa_context* c)
{
    PSA_HIP(hipSetDevice(c->device));
    return PSA_OK;
}

// The tables of one search mode: the context's own (psa_set_weights) or the opposite mode's.
// The pair-score table d_pair depends on the weights only and is shared.
struct TableSet {
    const psa::Tables*   t;
    const unsigned char* d_rank;
    const unsigned char* d_sub;
    const double*        d_delta;
    const unsigned char* d_clsrank;
};

TableSet primary_tables(const psa_context* c)
{
    return TableSet{&c->tables, c->d_rank, c->d_sub, c->d_delta, c->d_clsrank};
}

TableSet alternate_tables(const psa_context* c)
{
    return TableSet{&c->alt_tables, c->d_alt_rank, c->d_alt_sub, c->d_alt_delta, c->d_alt_clsrank};
}

psa_status upload_mode_tables(psa_context* c, const psa::Tables& t, unsigned char* d_rank, unsigned char* d_sub,
                              double* d_delta, unsigned char* d_clsrank)
{
    PSA_HIP(hipMemcpyAsync(d_rank, t.rank, sizeof t.rank, hipMemcpyHostToDevice, c->stream));
    PSA_HIP(hipMemcpyAsync(d_sub, t.sub, sizeof t.sub, hipMemcpyHostToDevice, c->stream));
    PSA_HIP(hipMemcpyAsync(d_delta, t.delta_of_rank, sizeof t.delta_of_rank, hipMemcpyHostToDevice, c->stream));
    unsigned char clsrank[psa::kRow * psa::kRow];
    for (int k = 0; k < psa::kRow * psa::kRow; ++k) clsrank[k] = (unsigned char)(t.cls[k] | (t.rank[k] << 3));
    PSA_HIP(hipMemcpyAsync(d_clsrank, clsrank, sizeof clsrank, hipMemcpyHostToDevice, c->stream));
    PSA_HIP(hipStreamSynchronize(c->stream));   // host arrays go out of scope right after return
    return PSA_OK;
}

psa_status upload_tables(psa_context* c)
{
    const psa::Tables& t = c->tables;
    PSA_HIP(hipMemcpyAsync(c->d_pair, t.pair, sizeof t.pair, hipMemcpyHostToDevice, c->stream));
    return upload_mode_tables(c, t, c->d_rank, c->d_sub, c->d_delta, c->d_clsrank);
}

// tables of the opposite mode for the same weights
psa_status ensure_alt_tables(psa_context* c)
{
    if (c->have_alt) return PSA_OK;
    std::string err;
    if (!psa::build_tables(c->tables.weights, c->tables.is_max ? 0 : 1, &c->alt_tables, &err)) {
        set_last_error(err);
        return PSA_ERR_ARG;
    }
    PSA_TRY(upload_mode_tables(c, c->alt_tables, c->d_alt_rank, c->d_alt_sub, c->d_alt_delta, c->d_alt_clsrank));
    c->have_alt = true;
    return PSA_OK;
}

// Reads the encode kernels' flag.  Blocking; used by the host-sequence setters
// (device-sequence setters defer the check to the end of the search).
psa_status check_bad_now(psa_context* c)
{
    PSA_HIP(hipMemcpyAsync(c->h_bad, c->d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PSA_HIP(hipStreamSynchronize(c->stream));
    return PSA_OK;
}

psa_status report_bad(psa_context* c)
{
    const int bad = *c->h_bad;
    if (!bad) return PSA_OK;
    PSA_HIP(hipMemsetAsync(c->d_bad, 0, sizeof(int), c->stream));
    *c->h_bad = 0;
    if (bad & psa::kStreamBad) {   // the streamed encode launch gave up waiting for the host's copy: whatever it encoded is void
        c->n = -1;
        for (auto& s : c->slot) s.m = -1;
        c->stream_gave_up = true;
        c->stream_retries += 1;
        set_last_error("internal: the encode launch gave up waiting for the host's copy");
        return PSA_ERR_DEVICE;
    }
    if (bad & 4) {      // (batches retry without the batch kernel first: psa_search_batch*)
        set_last_error("internal: a wave-group barrier of the batch count kernel timed out");
        return PSA_ERR_DEVICE;
    }
    if (bad & 1) c->n = -1;
    if (bad & 2) for (auto& s : c->slot) s.m = -1;
    set_last_error(std::string(bad & 1 ? "seq1" : "seq2") + " contains a character outside [A-Z-]");
    return PSA_ERR_ALPHABET;
}

// The pinned control words of the encode launches and their device-side companions: pinned word 0 = the host's copy
// progress of a streamed transfer, pinned word 8 (a cache line of its own) = an encode launch's "done" word (ticket and
// alphabet flag, written by its last workgroup); device word 0 = the poller's copy of the progress, word 8 = the
// workgroups' arrival counter.
psa_status ensure_ctl(psa_context* c)
{
    if (c->h_ctl) return PSA_OK;                  // (psa_create carves them out of the arenas)
    set_last_error("internal: no control words");
    return PSA_ERR_STATE;
}

unsigned next_ticket(psa_context* c)
{
    c->stream_ticket = (c->stream_ticket + 1) & 0xffffffu;
    if (c->stream_ticket == 0) c->stream_ticket = 1;
    return c->stream_ticket;
}

// `report`: the launch's last workgroup writes its ticket and the alphabet flag to the pinned "done" word (wait_done)
psa_status done_word(psa_context* c, bool report, psa::DoneWord* d)
{
    d->arrivals = nullptr;
    d->host_done = nullptr;
    d->bad = c->d_bad;
    d->ticket = 0u;
    if (!report) return PSA_OK;
    PSA_TRY(ensure_ctl(c));
    d->arrivals = reinterpret_cast<unsigned*>(c->d_ctl + 8);
    d->host_done = c->hd_ctl + 8;
    d->ticket = next_ticket(c);
    return PSA_OK;
}

// The encode launch in flight reports through the "done" word: watching it returns a few microseconds before a copy
// of the flag + the stream's completion signal would (a launch that takes longer than 2 ms falls back to exactly that).
psa_status wait_done(psa_context* c)
{
    const volatile unsigned long long* done = c->h_ctl + 8;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int k = 0; k < 256; ++k) {
            const unsigned long long w = *done;
            if ((unsigned)(w >> 40) == c->stream_ticket) {
                std::atomic_thread_fence(std::memory_order_acquire);
                *c->h_bad = (int)(w & 0xffffffu);
                return PSA_OK;
            }
            __builtin_ia32_pause();
        }
        if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 2000.0) break;
    }
    return check_bad_now(c);
}

// Seq1: ASCII -> symbol codes (the class planes are built per workgroup inside the count kernel)
psa_status seq1_args(psa_context* c, const void* d_raw, int n, psa::PrepareArgs* a)
{
    if (n > PSA_MAX_SEQ1) { set_last_error("seq1 is longer than PSA_MAX_SEQ1 (2^30)"); return PSA_ERR_ARG; }
    const int need = (n + kSeq1Slack + 63) & ~63;
    if (need > c->n_cap) {
        PSA_TRY(dev_alloc(&c->d_s1, (size_t)need));
        c->n_cap = need;
        c->nslots = need / 32;
    }
    c->n = -1;
    a->raw = static_cast<const unsigned char*>(d_raw);
    a->codes = c->d_s1;
    a->bad = c->d_bad;
    a->n = n;
    a->nslots = c->nslots;
    return PSA_OK;
}

psa_status encode_seq1(psa_context* c, const void* d_raw, int n, bool report = false)
{
    psa::PrepareArgs a;
    psa::DoneWord done;
    PSA_TRY(seq1_args(c, d_raw, n, &a));
    PSA_TRY(done_word(c, report, &done));
    hipLaunchKernelGGL(psa::psa_prepare_seq1_kernel, dim3((c->nslots + 255) / 256), dim3(256), 0, c->stream, a, done);
    PSA_HIP(hipGetLastError());
    c->n = n;
    return PSA_OK;
}

bool plan_filter(int first, int last, int m, FilterPlan* best, int max_chunks = 1 << 30);

// the slot's three Seq2 buffers hold m symbols (+ the planners' slack)
psa_status ensure_seq2_capacity(SearchSlot& s, int m)
{
    const int need = (m + seq2_slack(m) + 63) & ~63;
    if (need > s.m_cap) {
        s.m_cap = 0;
        PSA_TRY(dev_alloc(&s.d_s2rows, (size_t)need));
        PSA_TRY(dev_alloc(&s.d_s2bytes, (size_t)need));
        PSA_TRY(dev_alloc(&s.d_s2cls, (size_t)need));
        s.m_cap = need;
    }
    return PSA_OK;
}

// `n` = the Seq1 length the class order is planned for (the context's, or the one being set in the same launch)
psa_status seq2_args(SearchSlot& s, const void* d_raw, int m, int n, int* d_bad, psa::Seq2Args* a, FilterPlan* p)
{
    PSA_TRY(ensure_seq2_capacity(s, m));
    s.m = -1;
    // class order for the chunking a search over every offset will use (a search of another range
    // that plans differently re-orders them: ensure_class_order)
    const int total = n >= m ? n - m + 1 : psa::kTileStride * 256;
    if (!plan_filter(0, total, m, p)) { p->class_chars = 0; p->nchunks = 0; }
    a->raw = static_cast<const unsigned char*>(d_raw);
    a->rows = s.d_s2rows;
    a->bytes = s.d_s2bytes;
    a->classes = s.d_s2cls;
    a->bad = d_bad;
    a->m = m;
    a->capacity = s.m_cap;
    a->class_chars = p->class_chars;
    a->class_total = p->nchunks * psa::kClasses * p->class_chars;
    return PSA_OK;
}

void seq2_encoded(SearchSlot& s, int m, const FilterPlan& p)
{
    s.m = m;
    s.cls_chars = p.class_chars;
    s.cls_chunks = p.nchunks;
    s.s2rows = s.d_s2rows;
    s.s2bytes = s.d_s2bytes;
    s.s2cls = s.d_s2cls;
}

psa_status encode_seq2(psa_context* c, SearchSlot& s, const void* d_raw, int m, hipStream_t stream, bool report = false)
{
    psa::Seq2Args a;
    psa::DoneWord done;
    FilterPlan p;
    PSA_TRY(seq2_args(s, d_raw, m, c->n, c->d_bad, &a, &p));
    PSA_TRY(done_word(c, report, &done));
    hipLaunchKernelGGL(psa::psa_encode_seq2_kernel, dim3((s.m_cap / 16 + 255) / 256), dim3(256), 0, stream, a, done);
    PSA_HIP(hipGetLastError());
    seq2_encoded(s, m, p);
    return PSA_OK;
}

// both sequences of a search (slot 0) by one launch
psa_status encode_pair(psa_context* c, const void* d_raw1, int n, const void* d_raw2, int m, bool report = false)
{
    SearchSlot& s = c->slot[0];
    psa::PrepareArgs a1;
    psa::Seq2Args a2;
    psa::DoneWord done;
    FilterPlan p;
    PSA_TRY(seq1_args(c, d_raw1, n, &a1));
    PSA_TRY(seq2_args(s, d_raw2, m, n, c->d_bad, &a2, &p));
    PSA_TRY(done_word(c, report, &done));
    const int blocks1 = (c->nslots + 255) / 256, blocks2 = (s.m_cap / 16 + 255) / 256;
    hipLaunchKernelGGL(psa::psa_encode_pair_kernel, dim3(blocks1 + blocks2), dim3(256), 0, c->stream, a1, a2, blocks1, done);
    PSA_HIP(hipGetLastError());
    c->n = n;
    seq2_encoded(s, m, p);
    return PSA_OK;
}

// the slot's own Seq2 in the class order of plan `p`
psa_status ensure_class_order(SearchSlot& s, const FilterPlan& p, hipStream_t stream)
{
    if (s.s2cls != s.d_s2cls || (s.cls_chars == p.class_chars && s.cls_chunks == p.nchunks)) return PSA_OK;
    const int class_total = p.nchunks * psa::kClasses * p.class_chars;
    hipLaunchKernelGGL(psa::psa_class_order_kernel, dim3((class_total + psa::kUnroll + 255) / 256), dim3(256), 0, stream,
                       s.d_s2bytes, s.m, s.d_s2cls, p.class_chars, class_total);
    PSA_HIP(hipGetLastError());
    s.cls_chars = p.class_chars;
    s.cls_chunks = p.nchunks;
    return PSA_OK;
}

psa_status stage_host(psa_context* c, const char* host, size_t len)
{
    if (len > c->raw_cap) {
        c->raw_cap = 0;
        PSA_TRY(dev_alloc(&c->d_raw, len + 64));
        c->raw_cap = len + 64;
    }
    PSA_HIP(hipMemcpyAsync(c->d_raw, host, len, hipMemcpyHostToDevice, c->stream));
    return PSA_OK;
}

psa_status ensure_stage(psa_context* c, size_t len)
{
    if (len <= c->stage_cap) return PSA_OK;
    if (c->h_stage) PSA_HIP(hipHostFree(c->h_stage));
    c->h_stage = c->hd_stage = nullptr;
    c->stage_cap = 0;
    const size_t cap = len + len / 4 + 4096;
    PSA_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_stage), cap, hipHostMallocMapped | hipHostMallocCoherent));
    PSA_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->hd_stage), c->h_stage, 0));
    c->stage_cap = cap;
    return PSA_OK;
}

// Both sequences of a search into the pinned staging buffer (host memcpy); the encode kernel reads them from there.
// `one_object`: the caller knows that both lie inside ONE object (the two arrays of a ProgramData), so the bytes between
// them may be read: they keep their distance and the copy is one memcpy when the gap is small.  Arbitrary caller
// buffers (psa_set_sequences) are never compared or bridged: two copies.
psa_status stage_pinned_pair(psa_context* c, const char* a, size_t na, const char* b, size_t nb, bool one_object = false)
{
    const bool adjacent = one_object && b >= a + na && (size_t)(b - a) <= na + 4096;
    const size_t second = adjacent ? (size_t)(b - a) : ((na + 63) & ~(size_t)63);
    const size_t len = second + nb;
    PSA_TRY(ensure_stage(c, len));
    if (adjacent) std::memcpy(c->h_stage, a, len);
    else {
        std::memcpy(c->h_stage, a, na);
        std::memcpy(c->h_stage + second, b, nb);
    }
    c->stage_second = second;
    return PSA_OK;
}

// ---- long sequences from pageable host memory: the encode launch FOLLOWS the host's copy ---------------------------
// stage_pinned_pair + encode_pair put three things one after the other: the host's copy into pinned staging (12 us per
// MB), the launch (4 us until the kernel runs) and the kernel's reads over PCIe (25 us per MB).  Here the launch goes
// first; the host then copies Seq1 in 64 KB chunks and publishes its progress in a control word after each, and every
// workgroup of psa_encode_pair_streamed_kernel reads its 8 KB as soon as they are there: launch latency and copy run
// under the PCIe transfer.  `find_nul`: Seq1's length is not known (gpu_run_program gets a ProgramData, not lengths):
// the copy stops at the first NUL inside `limit1` bytes and the kernel learns the length from the final publication.
// *n_out = Seq1's length (== limit1 when find_nul met no NUL).  The caller has checked m >= 1.
// Measured, 1M x 10k on one MI355X (tools/stream_ab.sh, same box): psa_set_sequences 55.4 -> 44.4 us, gpu_run_program
// 138.5 -> 122.2 us; chunks of 16 KB / 256 KB / 1 MB: 54 / 48 / 53 us; non-temporal stores for the copy: no gain (the
// copy itself takes twice as long).  The kernel's last workgroup finishes ~31 us after the first started, against the
// 25 us the PCIe reads alone take (tools/stream_phase.py).
constexpr size_t kStreamChunk = 64 * 1024;
constexpr size_t kStreamMin = 256 * 1024;        // shorter transfers: nothing to hide a poll over PCIe behind
size_t stream_min()
{
#ifdef PSA_EXPERIMENT
    // experiment knob (debug library): another threshold, e.g. a huge one to time the plain transfer on the same GPU
    static const size_t v = std::getenv("PSA_STREAM_MIN") ? (size_t)std::atoll(std::getenv("PSA_STREAM_MIN")) : kStreamMin;
    return v;
#else
    return kStreamMin;
#endif
}

void stream_publish(psa_context* c, unsigned ticket, size_t bytes, bool final)
{
    __builtin_ia32_sfence();                      // (a libc memcpy may use non-temporal stores, which nothing else orders)
    __atomic_store_n(c->h_ctl, ((unsigned long long)ticket << 40) | (final ? psa::kStreamFinal : 0ull) |
                                   ((unsigned long long)bytes & psa::kStreamBytes), __ATOMIC_RELEASE);
}

struct StreamKnobs { size_t chunk; bool trace; };   // experiment knobs of the debug library
const StreamKnobs& stream_knobs()
{
#ifdef PSA_EXPERIMENT
    static const StreamKnobs k = {std::getenv("PSA_STREAM_CHUNK") ? (size_t)std::atoll(std::getenv("PSA_STREAM_CHUNK")) : kStreamChunk,
                                  std::getenv("PSA_STREAM_TRACE") != nullptr};
#else
    static const StreamKnobs k = {kStreamChunk, false};
#endif
    return k;
}

psa_status stream_pair(psa_context* c, const char* seq1, size_t limit1, bool find_nul, const char* seq2, size_t m, int* n_out)
{
    SearchSlot& s = c->slot[0];
    const auto t_enter = std::chrono::steady_clock::now();
    PSA_TRY(ensure_ctl(c));
    // Seq2 starts on a 256-byte boundary at least one whole line behind Seq1's last byte: no 128-byte line holds bytes
    // of both, so a line a Seq2 workgroup has read (before Seq1's tail was copied) never carries Seq1 bytes
    const size_t second = ((limit1 + 255) & ~(size_t)255) + 256;
    PSA_TRY(ensure_stage(c, second + m));
    c->stage_second = second;
    const unsigned ticket = next_ticket(c);
    std::memcpy(c->h_stage + second, seq2, m);
    if (!c->stream_force_timeout) stream_publish(c, ticket, 0, false);

    psa::PrepareArgs a1;
    psa::Seq2Args a2;
    FilterPlan p;
    PSA_TRY(seq1_args(c, c->hd_stage, (int)limit1, &a1));        // (room for the longest Seq1 the caller may hold)
    PSA_TRY(seq2_args(s, c->hd_stage + second, (int)m, (int)limit1, c->d_bad, &a2, &p));
    const int blocks1 = (c->nslots + 255) / 256, blocks2 = (s.m_cap / 16 + 255) / 256;
    hipLaunchKernelGGL(psa::psa_encode_pair_streamed_kernel, dim3(1 + blocks1 + blocks2), dim3(256), 0, c->stream, a1, a2, blocks1,
                       c->hd_ctl, c->d_ctl, ticket, c->stream_force_timeout ? 0 : (1 << 16),   // 2^16 polls of ~2 us
                       psa::DoneWord{reinterpret_cast<unsigned*>(c->d_ctl + 8), c->hd_ctl + 8, c->d_bad, ticket});
    const hipError_t launched = hipGetLastError();
    // (whatever happens from here on, the kernel is told that nothing more will come)
    size_t done = 0;
    bool end = !find_nul && limit1 == 0;
    const StreamKnobs& knobs = stream_knobs();
    const auto t_launched = std::chrono::steady_clock::now();
    while (!end && launched == hipSuccess) {
        size_t k = std::min(knobs.chunk, limit1 - done);
        if (find_nul) {
            const void* z = std::memchr(seq1 + done, 0, k);
            if (z) { k = (size_t)(static_cast<const char*>(z) - (seq1 + done)); end = true; }
        }
        std::memcpy(c->h_stage + done, seq1 + done, k);
        done += k;
        if (done == limit1) end = true;
        if (!end && !c->stream_force_timeout) stream_publish(c, ticket, done, false);
    }
    if (!c->stream_force_timeout) stream_publish(c, ticket, done, true);
    if (knobs.trace) {
        const auto t_end = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(c->stream);
        const auto t_sync = std::chrono::steady_clock::now();
        std::fprintf(stderr, "stream_pair: launch %.1f us, copy %.1f us, kernel end %.1f us after the last publication\n",
                     std::chrono::duration<double, std::micro>(t_launched - t_enter).count(),
                     std::chrono::duration<double, std::micro>(t_end - t_launched).count(),
                     std::chrono::duration<double, std::micro>(t_sync - t_end).count());
    }
    if (launched != hipSuccess) { set_last_error(std::string("encode launch: ") + hipGetErrorString(launched)); return PSA_ERR_DEVICE; }
    *n_out = (int)done;
    c->streamed += 1;
    if (done >= 1) {
        c->n = (int)done;
        seq2_encoded(s, (int)m, p);
    }
    return PSA_OK;
}

psa_status ensure_tails(psa_context* c, int count)
{
    if (count <= c->tails_cap) return PSA_OK;
    if (c->h_tails && !c->tails_in_arena) PSA_HIP(hipHostFree(c->h_tails));   // (the first tail is part of the pinned arena)
    c->tails_in_arena = false;
    c->h_tails = nullptr;
    c->tails_cap = 0;
    PSA_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_tails), sizeof(psa::SearchTail) * (size_t)count,
                          hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->h_tails, 0, sizeof(psa::SearchTail) * (size_t)count);
    PSA_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->hd_tails), c->h_tails, 0));
    for (auto& s : c->slot) s.d_tail = c->hd_tails;
    c->tails_cap = count;
    return PSA_OK;
}

// ---- sequential engine ---------------------------------------------------------
int pick_r(const psa_context* c, int span)
{
    if (c->forced_r == 1 || c->forced_r == 2 || c->forced_r == 4) return c->forced_r;
    // 1024 SIMDs: keep several waves per SIMD before widening the per-lane tile
    if (span >= 2 * 64 * 3072) return 2;
    return 1;
}

psa_status launch_sequential(psa_context* c, SearchSlot& s, int first, int last, double* per_score,
                             unsigned char* per_rank, hipStream_t stream, const TableSet& ts)
{
    const int span = last - first;
    const int r = pick_r(c, span);
    const int base = first & ~3;
    const int lanes = (last - base + r - 1) / r;
    const int nblocks = (lanes + psa::kBlock - 1) / psa::kBlock;
    if (nblocks > s.blocks_cap) {
        s.blocks_cap = 0;
        PSA_TRY(dev_alloc(&s.d_blocks, (size_t)nblocks));
        s.blocks_cap = nblocks;
    }
    psa::SeqArgs a;
    a.s1 = c->d_s1;
    a.s2rows = s.s2rows;
    a.pair = c->d_pair;
    a.rank = ts.d_rank;
    a.delta_of_rank = ts.d_delta;
    a.block_out = s.d_blocks;
    a.per_score = per_score;
    a.per_rank = per_rank;
    a.m = s.m;
    a.m4 = (s.m + 3) & ~3;
    a.first = first;
    a.last = last;
    a.base = base;
    a.is_max = ts.t->is_max;

    if (c->profiling) PSA_HIP(hipEventRecord(c->ev[2], stream));
    switch (r) {
    case 4:
        hipLaunchKernelGGL(psa::psa_seq_kernel<4>, dim3(nblocks), dim3(psa::kBlock), 0, stream, a);
        c->last_kernel = "psa_seq_kernel<4>";
        break;
    case 2:
        hipLaunchKernelGGL(psa::psa_seq_kernel<2>, dim3(nblocks), dim3(psa::kBlock), 0, stream, a);
        c->last_kernel = "psa_seq_kernel<2>";
        break;
    default:
        hipLaunchKernelGGL(psa::psa_seq_kernel<1>, dim3(nblocks), dim3(psa::kBlock), 0, stream, a);
        c->last_kernel = "psa_seq_kernel<1>";
        break;
    }
    PSA_HIP(hipGetLastError());
    if (c->profiling) PSA_HIP(hipEventRecord(c->ev[3], stream));

    psa::FinalArgs f;
    f.block_in = s.d_blocks;
    f.nblocks = nblocks;
    f.s1 = c->d_s1;
    f.s2rows = s.s2rows;
    f.rank = ts.d_rank;
    f.sub = ts.d_sub;
    f.out = s.d_tail;
    f.bad = c->d_bad;
    f.m = s.m;
    f.is_max = ts.t->is_max;
    f.seq = c->seq;
    f.pairs = (long long)span * (long long)s.m;
    f.record_out = c->record_out;
    f.record_shard = c->record_shard;
    hipLaunchKernelGGL(psa::psa_finalize_kernel, dim3(1), dim3(psa::kBlock), 0, stream, f);
    PSA_HIP(hipGetLastError());
    return PSA_OK;
}

// ---- filter engine -------------------------------------------------------------
// Largest chunk count a Seq2 class buffer is sized for (every chunk rounds its classes up to whole rounds)
static_assert((24 + 1) * psa::kClasses * psa::kUnroll + psa::kUnroll <= kSeq2Slack, "class-order buffer of up to 24 chunks inside the fixed slack");

bool plan_filter(int first, int last, int m, FilterPlan* best, int max_chunks)
{
    const int ntiles = (last - (first & ~31) + psa::kTileStride - 1) / psa::kTileStride;
    const int rounds_total = ((m + psa::kClasses - 1) / psa::kClasses + psa::kUnroll - 1) / psa::kUnroll;   // per class
    double best_cost = 0.0;
    bool found = false;
#ifdef PSA_EXPERIMENT
    // experiment knobs (not part of the API, compiled out of the product build): force the workgroup shape / chunk count
    static const int force_chunks = std::getenv("PSA_FILTER_CHUNKS") ? std::atoi(std::getenv("PSA_FILTER_CHUNKS")) : 0;
#else
    constexpr int force_chunks = 0;
#endif
    {
        constexpr int waves = 16;
        const int wave_classes = psa::kClasses / waves;
        for (int nchunks = 1; nchunks <= kMaxChunks && nchunks <= rounds_total && nchunks <= max_chunks; ++nchunks) {
            if (force_chunks && nchunks != force_chunks) continue;
            const int rounds = (rounds_total + nchunks - 1) / nchunks;
            if (nchunks > 1 && rounds * (nchunks - 1) >= rounds_total) continue;   // last chunk would be empty
            const int class_chars = rounds * psa::kUnroll;
            if (wave_classes * class_chars > psa::kMaxWaveChars) continue;
            const int slots = psa::kTile / 32 + class_chars + 1;
            if (slots > psa::kRowSlots) continue;             // rows are kRowSlots slots apart in LDS
            const int lds = psa::kCountLdsBytes;              // plane rows + count cells: one workgroup per CU
            if (lds > kLdsLimit) continue;
            const long long blocks = (long long)ntiles * nchunks;
            const long long rounds_cu = (blocks + 255) / 256;        // waves of co-resident workgroups
            // Microseconds, fitted on MI355X (tools/plan_sweep.py, gpurun_out/plan_sweep_*.txt): the youngest wave of a
            // workgroup needs ~0.17 us per char of a class, a workgroup ~8 us around its main loop, the launch ~11 us;
            // the chunk workgroups of a tile meet through memory (write-through stores, arrival counter, the others'
            // counts read back).  The chunk term is round 4's fit (the tile's last arriver read the other chunks' counts in
            // a rolled loop, a memory round trip each, and selected alone).  Round 5 (loads issued ahead, 16-byte publish
            // stores, plans that fit the CUs share the tile's combine: psa_count_kernel) took 2-2.5 us off every such plan
            // without changing their order — gpurun_out/plan_sweep_r5.txt, kernel us for 1M x 10k, chunks 3 / 4 / 5 / 10:
            // 123,750 offsets 52.4 / 49.0 / 44.0 / 62.7 (two waves of workgroups), 247,500: 55.8 / 54.2 / 65.5 / 85.4,
            // 8,000: 50.9 / 47.0 / 40.9 / 41.2 — so the fit stays: it ranks the plans right, its absolute value is high
            // by those 2 us.  A chunked search of TWO tiles takes 40.7 us: launch, planes, counter sums, hand-off,
            // selection and the search's tail (12 us) are what a shard cannot shed.
            const double cost = (double)rounds_cu * (0.17 * class_chars + 8.0) + 11.0 +
                                (nchunks > 1 ? 2.3 + 1.7 * nchunks + 0.02 * (double)blocks : 0.0);
            if (!found || cost < best_cost) {
                found = true;
                best_cost = cost;
                best->waves = waves;
                best->nchunks = nchunks;
                best->class_chars = class_chars;
                best->lds_slots = slots;
                best->lds_bytes = lds;
                best->ntiles = ntiles;
            }
        }
    }
    return found;
}

double filter_slack(const psa::Tables& t, int m)
{
    // |sequential fp64 sum - exact sum| <= (m-1) u sum|p_i| / (1 - (m-1) u)   (u = 2^-53),
    // the count formula and the "+ delta" add a handful of roundings more.
    // Doubled, so that a borderline offset is always re-scored rather than dropped.
    double wmax = 0.0, dmax = 0.0;
    for (int k = 0; k < 4; ++k) wmax = std::fmax(wmax, std::fabs(t.weights[k]));
    for (int r = 1; r <= t.nranks; ++r) dmax = std::fmax(dmax, std::fabs(t.delta_of_rank[r]));
    const double u = 1.1102230246251565e-16;
    const double mag = (double)m * wmax + dmax;
    return 2.0 * 1.05 * ((double)m + 32.0) * u * mag;
}

psa::BoundParams bound_params(const psa::Tables& t, int m)
{
    psa::BoundParams b;
    for (int i = 0; i < 4; ++i) {
        b.class_best[i] = t.class_best_delta[i];
        b.class_worst[i] = t.class_worst_delta[i];
    }
    b.slack = filter_slack(t, m);
    {   // offset_bounds' fold over the classes, for an offset in which all four occur (same operations, same order)
        const bool mx = t.is_max != 0;
        double d_opt = mx ? -INFINITY : INFINITY, d_pes = d_opt;
        for (int k = 0; k < 4; ++k) {
            d_opt = mx ? std::fmax(d_opt, b.class_best[k]) : std::fmin(d_opt, b.class_best[k]);
            d_pes = mx ? std::fmax(d_pes, b.class_worst[k]) : std::fmin(d_pes, b.class_worst[k]);
        }
        b.opt_all = d_opt;
        b.pes_all = d_pes;
        const double sgn = mx ? 1.0 : -1.0;
        b.c[0] = sgn * (t.weights[0] + t.weights[3]);
        b.c[1] = sgn * (t.weights[3] - t.weights[1]);
        b.c[2] = sgn * (t.weights[3] - t.weights[2]);
        b.c_m = -sgn * t.weights[3] * (double)m;
        b.up_all = sgn * d_opt + b.slack;
        b.pes_all_add = sgn * d_pes - b.slack;
    }
    b.m = m;
    b.is_max = t.is_max;
    return b;
}

// What the exact re-score of a search's candidates needs — psa_rescore_kernel's arguments, also handed to the count
// kernel of a single search, whose last workgroup does that work itself (finish_search).
psa::RescoreArgs rescore_args(const psa_context* c, const SearchSlot& s, long long pairs, const TableSet& ts)
{
    const psa::Tables& t = *ts.t;
    psa::RescoreArgs r;
    r.s1 = c->d_s1;
    r.s2 = s.s2bytes;
    r.clsrank = ts.d_clsrank;
    r.sub = ts.d_sub;
    r.delta_of_rank = ts.d_delta;
    r.cand = s.d_cand;
    r.cand_upper = s.inline_select ? s.d_cand_upper : nullptr;
    r.global_best = &s.d_scal->global_best;
    r.best_split = nullptr;                   // (launch_count: single searches that end inside the count kernel)
    r.cand_count = &s.d_scal->cand_count;
    r.rec = s.d_rec;
    r.done = &s.d_scal->done;
    r.out = s.d_tail;
    r.bad = c->d_bad;
    r.pairs = pairs;
    r.seq = c->seq;
    r.share_swept = nullptr;                  // (launch_count: searches that share their tiles' combines)
    r.class_value[0] = t.weights[0];       // get_weight, cuda_funcs.cu:442-452
    r.class_value[1] = -t.weights[1];
    r.class_value[2] = -t.weights[2];
    r.class_value[3] = -t.weights[3];
    for (int i = 4; i < 8; ++i) r.class_value[i] = 0.0;
    r.m = s.m;
    r.cap = kCandCap;
    r.is_max = t.is_max;
    {
        // the integer walk needs normal (or zero) addends and a running sum at least two
        // binades above the largest of them
        int max_exp = 0;
        bool ok = true;
        for (int i = 0; i < 4; ++i) {
            unsigned long long bits;
            std::memcpy(&bits, &r.class_value[i], sizeof bits);
            const int e = (int)((bits >> 52) & 0x7ff);
            if (e == 0 && (bits << 1) != 0) ok = false;          // subnormal weight
            if (e > max_exp) max_exp = e;
        }
#ifdef PSA_EXPERIMENT
        const bool chain_only = std::getenv("PSA_RESCORE_CHAIN") != nullptr;   // experiment knob
#else
        constexpr bool chain_only = false;
#endif
        r.fast_ok = ok && !chain_only;
        r.fast_min_exp = max_exp + 2;
#ifdef PSA_EXPERIMENT
        const bool walk_only = std::getenv("PSA_RESCORE_WALK") != nullptr;     // experiment knob: no certified sums
#else
        constexpr bool walk_only = false;
#endif
        // |sequential fp64 prefix sum - exact prefix sum| <= (k-1) u sum|p_i| / (1 - k u) <= m u m wmax; doubled, and the
        // order-free evaluation the certificate compares with adds a few roundings of its own (far inside the factor)
        double wmax = 0.0;
        for (int i = 0; i < 4; ++i) wmax = std::fmax(wmax, std::fabs(r.class_value[i]));
        r.walk_slack = 2.1 * ((double)s.m + 32.0) * 1.1102230246251565e-16 * ((double)s.m * wmax);
        r.certify = r.fast_ok && !walk_only && s.m < 65536 && std::isfinite(r.walk_slack);
    }
    r.batched = 0;
    r.s2_stride = 0;
    r.record_out = nullptr;
    r.record_shard = -1;
    r.raw = 0;
    r.n1 = c->n;
    return r;
}

// first/span describe the searched range when the kernel should also produce the score
// bounds (single-chunk plans); span = 0 writes raw partial counts instead.
// `s2`/`ts2` (optional): a second search of the same Seq2 in another mode, selected in the same pass.
// `fused`: a single search that ends inside this launch (the last tile workgroup re-scores the candidates and writes
// the tail); otherwise psa_rescore_kernel follows.
// ---- who counts what: the waves' shares of a workgroup's symbol stream ------------------------------------------
// The stream of a chunk is 32 classes x R rounds (R = class_chars / 32); the four waves that share a SIMD (ages 0..3:
// waves s, s+4, s+8, s+12 of the workgroup) count 8 R rounds between them.  The hardware does not treat them alike —
// it issues the oldest wave of a SIMD first — so equal shares leave the SIMD on one wave that cannot fill its issue
// slots for the last third of the loop.  `permille[g]` = the share of age g; every share is capped by what a wave's
// counters hold (`cap` rounds), the rest goes to the other ages.
void deal_rounds(int rounds_per_class, const int permille[4], int cap, int q[4])
{
    const int total = 8 * rounds_per_class;
    int left = total;
    for (int g = 0; g < 4; ++g) {
        q[g] = std::min(cap, (total * permille[g] + 500) / 1000);
        left -= q[g];
    }
    for (int pass = 0; pass < 64 && left != 0; ++pass)          // rounding and caps: hand the difference round by round
        for (int g = 3; g >= 0 && left != 0; --g) {
            if (left > 0 && q[g] < cap) { ++q[g]; --left; }
            else if (left < 0 && q[g] > 0) { --q[g]; ++left; }
        }
}

// psa_count_kernel<16>: wave w has age w / 4 and SIMD w % 4; the ages' shares lie one after the other in the stream
bool deal_single(const psa_context* c, int class_chars, unsigned short lo[16], unsigned short hi[16])
{
    const int R = class_chars / psa::kUnroll;
    int q[4];
    deal_rounds(R, c->single_deal, psa::kMaxWaveChars / psa::kUnroll, q);
    if (q[0] + q[1] + q[2] + q[3] != 8 * R) return false;
    int at = 0;
    for (int g = 0; g < 4; ++g)
        for (int s = 0; s < 4; ++s) {
            lo[4 * g + s] = (unsigned short)at;
            at += q[g];
            hi[4 * g + s] = (unsigned short)at;
        }
    return at == psa::kClasses * R;
}

// psa_count_duo_kernel: a group's waves 0..3 are the older ones of their SIMDs, 4..7 the younger ones
bool deal_duo(const psa_context* c, int class_chars, unsigned short lo[8], unsigned short hi[8])
{
    const int R = class_chars / psa::kUnroll;
    const int cap = psa::kDuoMaxWaveChars / psa::kUnroll;
    int older = std::min(cap, (8 * R * c->duo_older_permille + 500) / 1000);
    int younger = 8 * R - older;
    if (younger > cap) { younger = cap; older = 8 * R - younger; }
    if (older > cap || older < 0) return false;
    int at = 0;
    for (int w = 0; w < 8; ++w) {
        lo[w] = (unsigned short)at;
        at += w < 4 ? older : younger;
        hi[w] = (unsigned short)at;
    }
    return at == psa::kClasses * R;
}

struct RawSequences {                 // a one-shot search's inputs: the caller's ASCII in device memory
    const unsigned char* seq1;
    int                  n;
    const unsigned char* seq2;
    int                  m;
};

// `raw`: the launch reads the caller's ASCII itself (psa_count_kernel<16, true>: no encode launch in front of it).
psa_status launch_count(psa_context* c, SearchSlot& s, const FilterPlan& p, int base, int span_pad, int first,
                        int span, bool inline_select, hipStream_t stream, const TableSet& ts,
                        const SearchSlot* s2 = nullptr, const TableSet* ts2 = nullptr, bool fused = false,
                        const RawSequences* raw = nullptr)
{
    if (p.nchunks > 1 && inline_select && !s.d_arrivals) { set_last_error("internal: no arrival counters"); return PSA_ERR_STATE; }
    psa::CountArgs a;
    a.s1 = c->d_s1;
    a.classes = s.s2cls;
    a.partial = s.d_partial;
    a.base = base;
    a.span_pad = span_pad;
    a.class_chars = p.class_chars;
    a.lds_slots = p.lds_slots;
    a.tile_arrivals = s.d_arrivals;
    a.share = 0;
    a.share_spins = c->share_spins;
    a.share_tag = 0ull;
    a.tile_ready = a.slice_claim = a.share_swept = nullptr;
    a.done_target = p.ntiles;
    if (!raw) c->pending_share_slices = 0;
    if (fused && inline_select && c->share_combine && c->share_pause > 0 && p.nchunks > 1) --c->share_pause;
    else if (fused && inline_select && c->share_combine && p.nchunks > 1 && p.nchunks <= psa::kShareMaxChunks && p.ntiles <= psa::kShareMaxTiles &&
        (long long)p.ntiles * p.nchunks <= c->cus) {
        // (every chunk workgroup of every tile is resident at once: they share the tiles' combine + selection)
        constexpr size_t kShareWords = (size_t)psa::kShareMaxTiles * (1 + psa::kShareMaxChunks) + 1;     // (+ the sweep counter)
        if (!s.d_share) {
            PSA_TRY(dev_alloc(&s.d_share, kShareWords));
            PSA_HIP(hipMemsetAsync(s.d_share, 0, sizeof(unsigned long long) * kShareWords, stream));     // (launch numbers start at 1)
        }
        a.share = 1;
        a.share_tag = ++s.share_launches;
        c->pending_share_slices = p.ntiles * (p.nchunks - 1);
        if (a.share_spins > 0) {
            // the last chunk of a plan may be shorter than the others (10 rounds as 3 + 3 + 3 + 1): its workgroups arrive that
            // much earlier and must not give up on the others for it — 0.17 us per char of a class (the planner's figure),
            // a poll every ~0.75 us
            const int rounds_total = ((s.m + psa::kClasses - 1) / psa::kClasses + psa::kUnroll - 1) / psa::kUnroll;
            const int last_chars = (rounds_total - (p.nchunks - 1) * (p.class_chars / psa::kUnroll)) * psa::kUnroll;
            if (last_chars > 0 && last_chars < p.class_chars) a.share_spins += (int)(0.17 * (p.class_chars - last_chars) / 0.75) + 1;
        }
        a.tile_ready = s.d_share;
        a.slice_claim = s.d_share + psa::kShareMaxTiles;
        a.share_swept = s.d_share + kShareWords - 1;
        a.done_target = p.ntiles * p.nchunks;
    }
    a.inline_select = inline_select ? 1 : 0;
    a.nmodes = s2 ? 2 : 1;
    a.cap = kCandCap;
    a.first = first;
    a.lead = first - base;
    a.span = span;
    a.sel[0].global_best = &s.d_scal->global_best;
    a.sel[0].cand = s.d_cand;
    a.sel[0].cand_upper = s.d_cand_upper;
    a.sel[0].cand_count = &s.d_scal->cand_count;
    a.sel[0].bounds = bound_params(*ts.t, s.m);
    a.sel[1] = a.sel[0];
    if (s2) {
        a.sel[1].global_best = &s2->d_scal->global_best;
        a.sel[1].cand = s2->d_cand;
        a.sel[1].cand_upper = s2->d_cand_upper;
        a.sel[1].cand_count = &s2->d_scal->cand_count;
        a.sel[1].bounds = bound_params(*ts2->t, s.m);
    }
    a.fused = fused ? 1 : 0;
    if (!deal_single(c, p.class_chars, a.wave_lo, a.wave_hi)) { set_last_error("internal: no deal of the symbol stream"); return PSA_ERR_STATE; }
    a.lds_bytes = p.lds_bytes;
    a.tail = rescore_args(c, s, (long long)span * s.m, ts);
    a.tail.record_out = c->record_out;
    a.tail.record_shard = c->record_shard;
    a.tail.share_swept = a.share ? a.share_swept : nullptr;       // (the tail echoes the slot's sweep counter to the host)
    if (fused && s.d_scal == c->slot[0].d_scal) a.tail.best_split = c->slot[0].d_best_split;     // (a slot's own scalars, not a pool entry)
    a.raw1 = a.raw2 = nullptr;
    a.classes_w = nullptr;
    a.bad = c->d_bad;
    a.n1 = a.m2 = a.class_total = 0;
    a.raw_arrivals = nullptr;
    a.raw_base = 0u;
    a.raw_converters = a.raw_expected = 0;
    a.raw_scan_from = 0;
#ifdef PSA_EXPERIMENT
    a.select_frozen = c->select_frozen_next ? 1 : 0;
    c->select_frozen_next = false;
#endif
    const dim3 grid(p.ntiles, p.nchunks);
    if (raw) {
        if (!fused || s2 || s.s2cls != s.d_s2cls) { set_last_error("internal: a raw launch is a fused single search"); return PSA_ERR_STATE; }
        a.raw1 = raw->seq1;
        a.n1 = raw->n;
        a.raw2 = raw->seq2;
        a.m2 = raw->m;
        a.classes_w = s.d_s2cls;
        a.class_total = p.nchunks * psa::kClasses * p.class_chars;
        a.tail.raw = 1;
        a.tail.n1 = raw->n;
        a.tail.s1 = raw->seq1;
        a.tail.s2 = raw->seq2;
        // what the tiles' plane builds look at (and validate) of Seq1: [base, raw_scan_from); the converters take the rest
        a.raw_scan_from = base + 32 * ((p.ntiles - 1) * (psa::kTileStride / 32) + p.nchunks * p.class_chars + psa::kTile / 32 + 1);
        {
            // converter workgroups in front of the tiles: on the CUs the tiles leave free (at most 16), at least as many
            // as keep a thread's share within kRawPerThread positions.  A search of a sub-range also has them validate the
            // Seq1 bytes outside the tiles' windows (the tiles wait for the converters' arrival): one more converter per
            // 128 KB of those, as far as the spare CUs go — 16 workgroups scanned 0.26 GB/ms, so the latency of a small
            // range under a long Seq1 grew with n, not with the range.
            const long long tiles = (long long)p.ntiles * p.nchunks;
            const int need = (a.class_total + psa::kUnroll + 1024 * psa::kRawPerThread - 1) / (1024 * psa::kRawPerThread);
            const long long spare = (long long)c->cus - tiles;
            const long long outside = (long long)base + std::max(0, raw->n - a.raw_scan_from);
            const long long want = std::max<long long>(c->raw_converters, (outside + (128 << 10) - 1) / (128 << 10));
            int conv = (int)std::max<long long>(need, std::min<long long>(want, spare > 0 ? spare : 8));
            a.raw_converters = conv;
            a.raw_expected = conv + (c->raw_force_timeout ? 1 : 0);
            a.raw_arrivals = c->d_raw_arrivals;
            a.raw_base = c->raw_arrived;
        }
        const bool chunked = p.nchunks > 1 || !inline_select;
        auto kernel = chunked ? &psa::psa_count_kernel<16, true, true> : &psa::psa_count_kernel<16, true, false>;
        if (!c->granted16[1][chunked]) {
            PSA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, psa::kCountLdsBytes));   // (+ 2.3 KB static)
            c->granted16[1][chunked] = true;
        }
        hipLaunchKernelGGL(kernel, dim3(p.ntiles + a.raw_converters, p.nchunks), dim3(16 * 64), p.lds_bytes, stream, a);
        c->last_kernel = "psa_count_kernel<16, raw>";
        PSA_HIP(hipGetLastError());
        c->raw_arrived += (unsigned)a.raw_converters;           // every converter arrives, however the search ends
        return PSA_OK;
    }
    // dynamic LDS above 64 KB has to be granted per kernel instance (and device: the flags live in the context).
    // One char chunk per tile, selected inline (every full-size single search): the instance without the chunk hand-off.
    const bool chunked = p.nchunks > 1 || !inline_select;
    auto kernel = chunked ? &psa::psa_count_kernel<16, false, true> : &psa::psa_count_kernel<16, false, false>;
    if (!c->granted16[0][chunked]) {
        PSA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit));
        c->granted16[0][chunked] = true;
    }
    PSA_STRACE(2);
    hipLaunchKernelGGL(kernel, grid, dim3(16 * 64), p.lds_bytes, stream, a);
    c->last_kernel = "psa_count_kernel<16>";
    PSA_HIP(hipGetLastError());
    PSA_STRACE(3);
    return PSA_OK;
}

// Batches: queries [q, q+count) of a pooled batch through psa_count_duo_kernel.  `scal`, `cand`, `cand_upper`
// are the pool entries of query q in each mode.
bool duo_fits(int m, int* class_chars, int* lds_slots)
{
    const int rounds = ((m + psa::kClasses - 1) / psa::kClasses + psa::kUnroll - 1) / psa::kUnroll;
    const int cc = rounds * psa::kUnroll;
    const int slots = psa::kTile / 32 + cc + 1;
    if ((psa::kClasses / psa::kDuoGroupWaves) * cc > psa::kDuoMaxWaveChars || slots > psa::kRowSlots) return false;
    *class_chars = cc;
    *lds_slots = slots;
    return true;
}

// With fewer tiles than CUs the queries of a launch are spread over grid.y: (tile, y) workgroups of `qpb` queries each
// (an even number: one query per wave group at a time), about one workgroup per CU.
int duo_queries_per_workgroup(int ntiles, int count, int* yblocks)
{
    int want = (256 + ntiles - 1) / ntiles;                 // y blocks that fill 256 CUs
    if (want < 1) want = 1;
    int qpb = (count + want - 1) / want;
    qpb = (qpb + 1) & ~1;
    if (qpb < 2) qpb = 2;
    *yblocks = (count + qpb - 1) / qpb;
    return qpb;
}

// queries per launch of the batch kernel: 32 for a long Seq1 (one workgroup per tile and launch, the re-score of a launch
// overlaps the next launch), more when there are few tiles, so that tiles x y-blocks still fills the GPU
int duo_queries_per_launch(const psa_context* c, int ntiles)
{
    const int want = (256 + ntiles - 1) / ntiles;           // y blocks, two queries each
    return std::max(c->duo_queries, std::min(4096, 2 * want));
}

psa_status launch_count_duo(psa_context* c, const unsigned* s2cls, size_t s2_stride, int m, int count, int first, int last,
                            int nmodes, const TableSet* tables, Scalars* const scal[2], int* const cand[2],
                            double* const cand_upper[2], hipStream_t stream)
{
    static_assert(sizeof(Scalars) == psa::kScalStride && offsetof(Scalars, cand_count) == psa::kScalCountOffset,
                  "psa_count_duo_kernel addresses the pool scalars by these strides");
    psa::DuoArgs a;
    if (!duo_fits(m, &a.class_chars, &a.lds_slots)) { set_last_error("internal: duo kernel does not fit"); return PSA_ERR_ARG; }
    const int base = first & ~31;
    a.s1 = c->d_s1;
    a.classes = s2cls;
    a.s2_stride = s2_stride;
    a.base = base;
    a.nq = count;
    a.nmodes = nmodes;
    a.cap = kCandCap;
    a.first = first;
    a.lead = first - base;
    a.span = last - first;
    a.skew_ticks = (unsigned)c->duo_skew_ticks;
    a.spin_limit = c->duo_force_timeout ? 0 : (1 << 21);
    a.prio_mode = c->duo_prio_mode;
#ifdef PSA_EXPERIMENT
    a.select_frozen = c->select_frozen ? 1 : 0;
#endif
    if (!deal_duo(c, a.class_chars, a.wave_lo, a.wave_hi)) { set_last_error("internal: no deal of the symbol stream"); return PSA_ERR_STATE; }
    a.bad = c->d_bad;
    for (int k = 0; k < 2; ++k) {
        const int src = k < nmodes ? k : 0;
        a.sel[k].scalars = reinterpret_cast<char*>(scal[src]);
        a.sel[k].cand = cand[src];
        a.sel[k].cand_upper = cand_upper[src];
        a.sel[k].bounds = bound_params(*tables[src].t, m);
    }
    if (!c->granted_duo) {
        PSA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&psa::psa_count_duo_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, psa::kDuoLdsBytes));
        c->granted_duo = true;
    }
    const int ntiles = (last - base + psa::kTileStride - 1) / psa::kTileStride;
    int yblocks = 0;
    a.qpb = duo_queries_per_workgroup(ntiles, count, &yblocks);
    hipLaunchKernelGGL(psa::psa_count_duo_kernel, dim3(ntiles, yblocks), dim3(1024), psa::kDuoLdsBytes, stream, a);
    PSA_HIP(hipGetLastError());
    return PSA_OK;
}

// Scratch of a multi-chunk plan: per-chunk counts of every offset and the tiles' arrival counters.
psa_status ensure_chunk_scratch(unsigned long long** partial, size_t* partial_cap, int** arrivals, int* arrivals_cap,
                                const FilterPlan& p, hipStream_t stream)
{
    const size_t need = (size_t)p.nchunks * p.ntiles * psa::kTile;
#ifdef PSA_EXPERIMENT
    // debug library: allocations of chunk counts beyond PSA_SCRATCH_ALLOC_FAIL_ABOVE bytes fail, as on a device without the memory
    if (const char* r = std::getenv("PSA_SCRATCH_ALLOC_FAIL_ABOVE"))
        if (need > *partial_cap && need * sizeof(unsigned long long) > (size_t)std::atoll(r)) {
            set_last_error("injected fault: hipMalloc of the chunk counts fails");
            return PSA_ERR_DEVICE;
        }
#endif
    if (need > *partial_cap) {
        *partial_cap = 0;
        PSA_TRY(dev_alloc(partial, need));
        *partial_cap = need;
    }
    if (p.ntiles > *arrivals_cap) {
        *arrivals_cap = 0;
        PSA_TRY(dev_alloc(arrivals, (size_t)p.ntiles));
        PSA_HIP(hipMemsetAsync(*arrivals, 0, sizeof(int) * (size_t)p.ntiles, stream));   // the kernels leave them zero
        *arrivals_cap = p.ntiles;
    }
    return PSA_OK;
}

// the plan of a search on a slot's own scratch buffers
psa_status plan_on_slot(SearchSlot& s, int first, int last, hipStream_t stream, FilterPlan* p)
{
    if (!plan_filter(first, last, s.m, p)) {
        set_last_error("internal: no filter plan fits the LDS");
        return PSA_ERR_ARG;
    }
    if (p->nchunks > 1) PSA_TRY(ensure_chunk_scratch(&s.d_partial, &s.partial_cap, &s.d_arrivals, &s.arrivals_cap, *p, stream));
    return ensure_class_order(s, *p, stream);
}

// count + bounds + candidate selection: one kernel (chunk workgroups of a tile combine in the kernel)
psa_status launch_filter_front(psa_context* c, SearchSlot& s, const FilterPlan& p, int first, int last, hipStream_t stream,
                               bool profile, const TableSet& ts, SearchSlot* s2 = nullptr, const TableSet* ts2 = nullptr,
                               bool fused = false, const RawSequences* raw = nullptr)
{
    const int span = last - first;
    const int base = first & ~31;
    const int span_pad = p.ntiles * psa::kTile;
    s.inline_select = true;
    if (s2) s2->inline_select = true;
    if (profile) PSA_HIP(hipEventRecord(c->ev[2], stream));
    PSA_TRY(launch_count(c, s, p, base, span_pad, first, span, true, stream, ts, s2, ts2, fused, raw));
    if (profile) PSA_HIP(hipEventRecord(c->ev[3], stream));
    return PSA_OK;
}

// exact re-score of the candidates + fold into the slot's tail
// `batch` > 0: s describes the first of `batch` consecutive pool entries (queries) re-scored by one launch
// `record`: where the finishing kernel also writes the 32-byte record in device memory (a batch: the first query's,
// query q at + q); default for a single search: the context's (device groups)
psa_status launch_filter_back(psa_context* c, SearchSlot& s, long long pairs, hipStream_t stream, const TableSet& ts,
                              int grid = psa::kRescoreGrid, int batch = 0, size_t s2_stride = 0, psa_result* record = nullptr)
{
    psa::RescoreArgs r = rescore_args(c, s, pairs, ts);
    if (record) {
        r.record_out = record;
        r.record_shard = -1;
    } else if (batch == 0) {
        r.record_out = c->record_out;
        r.record_shard = c->record_shard;
    }
    if (!c->granted_rescore) {
        PSA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&psa::psa_rescore_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, psa::kRescorePadLds));
        c->granted_rescore = true;
    }
    r.batched = batch > 0 ? 1 : 0;
    r.s2_stride = s2_stride;
    hipLaunchKernelGGL(psa::psa_rescore_kernel, dim3(grid, batch > 0 ? batch : 1), dim3(psa::kRescoreThreads),
                       c->rescore_pad ? psa::kRescorePadLds : 0, stream, r);
    PSA_HIP(hipGetLastError());
    return PSA_OK;
}

// A multi-chunk plan keeps every chunk's counts of every offset in memory until the tile's last chunk workgroup adds
// them up: nchunks x ntiles x 32 KB.  One search (or one ring buffer of a batch) takes at most the context's scratch_cap (16 GB) of it: a
// blocking psa_search over more offsets than that runs in pieces (search_in_pieces); every other path answers such a
// search with the sequential engine.
size_t chunk_scratch_bytes(const FilterPlan& p)
{
    return p.nchunks > 1 ? (size_t)p.nchunks * p.ntiles * psa::kTile * sizeof(unsigned long long) : 0;
}

bool use_filter(const psa_context* c, int first, int last, int m)
{
    if (c->engine == PSA_ENGINE_SEQUENTIAL) return false;
    FilterPlan p;
    if (!plan_filter(first, last, m, &p)) return false;      // Seq2 beyond what the count kernels chunk
    if (chunk_scratch_bytes(p) > c->scratch_cap) return false;
    if (c->engine == PSA_ENGINE_FILTER) return true;
    // Measured on MI355X (tools/engine_threshold.py, round 3 kernels: gpurun_out/engine_threshold_r3.txt), microseconds
    // per blocking search: the sequential engine is a serial chain of m additions per lane, the filter engine has the
    // larger fixed cost (plane build, epilogue, the search's tail) and ~30x the throughput.
    const double pairs = (double)(last - first) * m;
    const double t_sequential = 16.0 + 0.034 * m + pairs / 8.5e6;
    const double t_filter = 36.0 + 0.002 * m + pairs / 2.6e8;
    return t_filter < t_sequential;
}

// The chunk counts of the plan this search would use, allocated NOW if they are within the cap and not there yet — so that
// a device without that much free memory is found out before the engine is chosen: a failed allocation lowers the cap
// below the request (the search then runs in pieces or on the sequential engine, like any search over the cap) instead of
// failing the search with PSA_ERR_DEVICE.  `ring`: the two ring buffers of a batch instead of the slot's own scratch.
static void reserve_chunk_scratch(psa_context* c, int first, int last, int m, bool ring)
{
    if (c->engine == PSA_ENGINE_SEQUENTIAL || m < 1 || first >= last) return;
    FilterPlan p;
    if (!plan_filter(first, last, m, &p) || p.nchunks < 2) return;
    const size_t bytes = chunk_scratch_bytes(p);
    if (hipSetDevice(c->device) != hipSuccess) return;
    if (!c->scratch_cap_fitted && bytes > ((size_t)256 << 20)) {
        // (looked up once, by the first search that asks for much: psa_create stays as cheap as it is)
        size_t free_bytes = 0, total_bytes = 0;
        if (hipMemGetInfo(&free_bytes, &total_bytes) == hipSuccess) c->scratch_cap = std::min(c->scratch_cap, free_bytes / 2);
        else (void)hipGetLastError();
        c->scratch_cap_fitted = true;
    }
    if (bytes > c->scratch_cap) return;
    bool ok = true;
    if (ring) {
        for (int k = 0; k < 2 && ok; ++k) {
            size_t cap = c->ring_partial_cap;
            int acap = c->ring_arrivals_cap;
            ok = ensure_chunk_scratch(&c->d_ring_partial[k], &cap, &c->d_ring_arrivals[k], &acap, p, c->stream) == PSA_OK;
            if (ok && k == 1) { c->ring_partial_cap = cap; c->ring_arrivals_cap = acap; }
        }
        if (!ok) c->ring_partial_cap = 0, c->ring_arrivals_cap = 0;     // (whatever half of the pair exists is re-made next time)
    } else {
        SearchSlot& s = c->slot[0];
        ok = ensure_chunk_scratch(&s.d_partial, &s.partial_cap, &s.d_arrivals, &s.arrivals_cap, p, c->stream) == PSA_OK;
    }
    if (!ok) {
        (void)hipGetLastError();                   // (the out-of-memory error is handled here)
        c->scratch_cap = bytes > 1 ? bytes / 2 : 0;
        c->scratch_shrinks += 1;
    }
}

// The filter engine would take this search but for the cap on its chunk counts (health counter: psa_scratch_fallback_count)
static bool capped_out(const psa_context* c, int first, int last, int m)
{
    if (c->engine == PSA_ENGINE_SEQUENTIAL) return false;
    FilterPlan p;
    return plan_filter(first, last, m, &p) && chunk_scratch_bytes(p) > c->scratch_cap;
}


// (seq1_args refuses it too; the host variants ask before they stage a gigabyte)
psa_status check_seq1_length(int n)
{
    if (n > PSA_MAX_SEQ1) { set_last_error("seq1 is longer than PSA_MAX_SEQ1 (2^30)"); return PSA_ERR_ARG; }
    return PSA_OK;
}

psa_status check_ready(psa_context* c, int m, int first, int last)
{
    if (!c) { set_last_error("context is NULL"); return PSA_ERR_ARG; }
    if (!c->have_weights || c->n < 0 || m < 0) {
        set_last_error("weights, seq1 and seq2 must be set before a search");
        return PSA_ERR_STATE;
    }
    const long long total = (long long)c->n - m + 1;
    if (m < 1 || total < 1) {
        set_last_error("need 1 <= len(seq2) <= len(seq1)");
        return PSA_ERR_ARG;
    }
    if (first < 0 || last > total || first >= last) {
        set_last_error("offset range must satisfy 0 <= first < last <= len(seq1)-len(seq2)+1");
        return PSA_ERR_ARG;
    }
    return PSA_OK;
}

psa_status ensure_pool(psa_context* c, int nq)
{
    if (nq > c->pool_cap) {
        c->pool_cap = 0;
        PSA_TRY(dev_alloc(&c->d_pool_scal, (size_t)nq));
        PSA_TRY(dev_alloc(&c->d_pool_cand, (size_t)nq * kCandCap));
        PSA_TRY(dev_alloc(&c->d_pool_cand_upper, (size_t)nq * kCandCap));
        PSA_TRY(dev_alloc(&c->d_pool_rec, (size_t)nq * kCandCap));
        c->pool_cap = nq;
    }
    while ((int)c->pool_events.size() < nq) {
        hipEvent_t ev;
        PSA_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c->pool_events.push_back(ev);
    }
    return PSA_OK;
}

// The second and third stream (re-scores of a batch, alternate count launches) and the events between the streams:
// only batches use them, so the first batch creates them (a stream costs ~8 ms to create on this runtime — a fresh
// process that makes one gpu_run_program call does not pay for two it never uses).
psa_status ensure_batch_objects(psa_context* c)
{
    // (every object on its own: a creation that failed half-way is taken up where it stopped)
    if (!c->aux) PSA_HIP(hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking));
    if (!c->twin) PSA_HIP(hipStreamCreateWithFlags(&c->twin, hipStreamNonBlocking));
    for (auto& s : c->slot) {
        if (!s.front_done) PSA_HIP(hipEventCreateWithFlags(&s.front_done, hipEventDisableTiming));
        if (!s.back_done) PSA_HIP(hipEventCreateWithFlags(&s.back_done, hipEventDisableTiming));
    }
    return PSA_OK;
}

psa_status ensure_timing_events(psa_context* c)
{
    for (auto& ev : c->ev)
        if (!ev) PSA_HIP(hipEventCreate(&ev));
    return PSA_OK;
}

void free_slot(SearchSlot& s)
{
    (void)hipFree(s.d_s2rows);
    (void)hipFree(s.d_s2bytes);
    (void)hipFree(s.d_s2cls);
    (void)hipFree(s.d_blocks);
    (void)hipFree(s.d_partial);
    (void)hipFree(s.d_arrivals);
    (void)hipFree(s.d_share);
    if (s.front_done) (void)hipEventDestroy(s.front_done);       // (scalars, candidate lists, records: the device arena)
    if (s.back_done) (void)hipEventDestroy(s.back_done);
}

}  // namespace

extern "C" {

psa_status psa_create(int device, void* stream, psa_context** out)
{
    if (!out) { set_last_error("psa_create: out is NULL"); return PSA_ERR_ARG; }
    *out = nullptr;
    int count = 0;
    init_trace("psa_create: enter");
    hipError_t e = hipGetDeviceCount(&count);
    init_trace("psa_create: hipGetDeviceCount (HIP runtime initialised)");
    if (e != hipSuccess || count < 1) {
        set_last_error(std::string("no HIP device available: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0"));
        return PSA_ERR_DEVICE;
    }
    if (device < 0 || device >= count) { set_last_error("psa_create: device index out of range"); return PSA_ERR_ARG; }
    psa_context* c = new psa_context();
    c->device = device;
    DeviceScope device_scope;
    psa_status st = [&]() -> psa_status {
        PSA_HIP(hipSetDevice(device));
        init_trace("psa_create: hipSetDevice");
        if (stream) c->stream = static_cast<hipStream_t>(stream);
#ifdef PSA_EXPERIMENT
        else if (const char* how = std::getenv("PSA_MAIN_STREAM")) {      // cold-start experiment: what a created stream costs
            c->stream = std::strcmp(how, "perthread") == 0 ? hipStreamPerThread : nullptr;
            init_trace(std::strcmp(how, "perthread") == 0 ? "psa_create: hipStreamPerThread" : "psa_create: the null stream");
        }
#endif
        else {
            PSA_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
            c->own_stream = true;
        }
        init_trace("psa_create: main stream");
        if (hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->cus < 1) c->cus = 256;
        // ---- the device arena: every fixed-size device buffer of the context, one allocation, one fill
        {
            size_t at = 0;
            auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) & ~(size_t)255; return o; };
            const size_t o_pair = take(sizeof(double) * psa::kRow * psa::kRow);
            const size_t o_rank = take(psa::kRow * psa::kRow), o_sub = take(psa::kRow * psa::kRow), o_clsrank = take(psa::kRow * psa::kRow);
            const size_t o_delta = take(sizeof(double) * psa::kMaxRanks);
            const size_t o_arank = take(psa::kRow * psa::kRow), o_asub = take(psa::kRow * psa::kRow), o_aclsrank = take(psa::kRow * psa::kRow);
            const size_t o_adelta = take(sizeof(double) * psa::kMaxRanks);
            const size_t o_bad = take(sizeof(int));
            const size_t o_ctl = take(128);
            const size_t o_rawarr = take(64);
            size_t o_scal[kSlots], o_split[kSlots], o_cand[kSlots], o_upper[kSlots], o_rec[kSlots];
            for (int k = 0; k < kSlots; ++k) {
                o_scal[k] = take(sizeof(Scalars));
                o_split[k] = take(sizeof(unsigned long long) * psa::kBestSplit * psa::kBestStride);
                o_cand[k] = take(sizeof(int) * kCandCap);
                o_upper[k] = take(sizeof(double) * kCandCap);
                o_rec[k] = take(sizeof(psa::CandRec) * kCandCap);
            }
            PSA_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_arena), at));
            init_trace("psa_create: device arena (one hipMalloc)");
            PSA_HIP(hipMemsetAsync(c->d_arena, 0, at, c->stream));       // flag, control words, counters, search scalars: zero
            unsigned char* d = c->d_arena;
            c->d_pair = reinterpret_cast<double*>(d + o_pair);
            c->d_rank = d + o_rank;  c->d_sub = d + o_sub;  c->d_clsrank = d + o_clsrank;
            c->d_delta = reinterpret_cast<double*>(d + o_delta);
            c->d_alt_rank = d + o_arank;  c->d_alt_sub = d + o_asub;  c->d_alt_clsrank = d + o_aclsrank;
            c->d_alt_delta = reinterpret_cast<double*>(d + o_adelta);
            c->d_bad = reinterpret_cast<int*>(d + o_bad);
            c->d_ctl = reinterpret_cast<unsigned long long*>(d + o_ctl);
            c->d_raw_arrivals = reinterpret_cast<unsigned*>(d + o_rawarr);
            c->raw_arrived = 0u;
            for (int k = 0; k < kSlots; ++k) {
                SearchSlot& sl = c->slot[k];
                sl.d_scal = reinterpret_cast<Scalars*>(d + o_scal[k]);
                sl.d_best_split = reinterpret_cast<unsigned long long*>(d + o_split[k]);
                sl.d_cand = reinterpret_cast<int*>(d + o_cand[k]);
                sl.d_cand_upper = reinterpret_cast<double*>(d + o_upper[k]);
                sl.d_rec = reinterpret_cast<psa::CandRec*>(d + o_rec[k]);
            }
        }
        // ---- the pinned arena (host-coherent, mapped): control words of the encode launches (two cache lines), the
        // single search's result tail, the host copy of the alphabet flag
        {
            constexpr size_t kHostArena = 128 + 128 + 64;
            static_assert(sizeof(psa::SearchTail) <= 128, "pinned arena");
            PSA_HIP(hipHostMalloc(reinterpret_cast<void**>(&c->h_arena), kHostArena, hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(c->h_arena, 0, kHostArena);
            unsigned char* hd = nullptr;
            PSA_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), c->h_arena, 0));
            c->h_ctl = reinterpret_cast<unsigned long long*>(c->h_arena);
            c->hd_ctl = reinterpret_cast<unsigned long long*>(hd);
            c->h_tails = reinterpret_cast<psa::SearchTail*>(c->h_arena + 128);
            c->hd_tails = reinterpret_cast<psa::SearchTail*>(hd + 128);
            c->tails_cap = 1;
            c->tails_in_arena = true;
            for (auto& sl : c->slot) sl.d_tail = c->hd_tails;
            c->h_bad = reinterpret_cast<int*>(c->h_arena + 256);
            init_trace("psa_create: pinned arena (one hipHostMalloc)");
        }
        // The count kernels' plane build has the pair relation compiled in (kPlaneSets, from the group lists of
        // psa_internal.hpp); the score tables come from psa::sign_of at run time: they must describe the same relation.
        for (int a = 0; a <= psa::kCodeHyphen; ++a)
            for (int b = 0; b <= psa::kCodeHyphen; ++b) {
                const char sign = psa::sign_of(a, b);
                const bool lo = sign == PSA_SIGN_STAR || sign == PSA_SIGN_DOT, hi = sign == PSA_SIGN_COLON || sign == PSA_SIGN_DOT;
                if (lo != (((psa::kPlaneSets.lo[b] >> a) & 1u) != 0u) || hi != (((psa::kPlaneSets.hi[b] >> a) & 1u) != 0u)) {
                    set_last_error("internal: the kernels' compiled-in pair relation differs from the host tables'");
                    return PSA_ERR_STATE;
                }
            }
        PSA_HIP(hipStreamSynchronize(c->stream));
        init_trace("psa_create: arena fill enqueued, stream synchronised");
        return PSA_OK;
    }();
    if (st != PSA_OK) { psa_destroy(c); return st; }
#ifdef PSA_EXPERIMENT
    // Experiment / fault-injection knobs: only in the debug library (libpsa_debug.so, -DPSA_EXPERIMENT);
    // the product library reads no environment variable except PSA_DEVICE (legacy entry point).
    if (const char* r = std::getenv("PSA_SEQ_R")) c->forced_r = std::atoi(r);
    if (const char* r = std::getenv("PSA_DUO_PRIO")) c->duo_prio_mode = std::atoi(r);
    if (const char* r = std::getenv("PSA_DUO_OLDER")) c->duo_older_permille = std::max(0, std::min(1000, std::atoi(r)));
    if (const char* r = std::getenv("PSA_DEAL")) {              // e.g. 375,250,250,125: permille by age, oldest first
        int v[4] = {0, 0, 0, 0};
        if (std::sscanf(r, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4)
            for (int g = 0; g < 4; ++g) c->single_deal[g] = v[g];
    }
    if (const char* r = std::getenv("PSA_BATCH_STREAMS")) c->batch_streams = std::atoi(r) >= 2 ? 2 : 1;
    if (const char* r = std::getenv("PSA_RESCORE_GRID")) c->batch_rescore_grid = std::max(1, std::min(1024, std::atoi(r)));
    if (const char* r = std::getenv("PSA_RESCORE_PAD")) c->rescore_pad = std::atoi(r) != 0;
    if (const char* r = std::getenv("PSA_DUO")) c->duo = std::atoi(r) != 0;
    if (const char* r = std::getenv("PSA_DUO_QUERIES")) c->duo_queries = std::max(2, std::min(4096, std::atoi(r)));
    if (const char* r = std::getenv("PSA_DUO_SKEW")) c->duo_skew_ticks = std::max(0, std::atoi(r));
    if (const char* r = std::getenv("PSA_DUO_FORCE_TIMEOUT")) c->duo_force_timeout = std::atoi(r) != 0;
    if (const char* r = std::getenv("PSA_STREAM_FORCE_TIMEOUT")) c->stream_force_timeout = std::atoi(r) != 0;
    if (const char* r = std::getenv("PSA_RAW_CONVERTERS")) c->raw_converters = std::max(1, std::min(1024, std::atoi(r)));
    if (const char* r = std::getenv("PSA_RAW_FORCE_TIMEOUT")) c->raw_force_timeout = std::atoi(r) != 0;
    if (const char* r = std::getenv("PSA_SELECT_FROZEN")) c->select_frozen = std::atoi(r) != 0;
    if (const char* r = std::getenv("PSA_SHARE")) c->share_combine = std::atoi(r) != 0;
    if (const char* r = std::getenv("PSA_SHARE_SPINS")) c->share_spins = std::max(0, std::atoi(r));
    if (const char* r = std::getenv("PSA_SHARE_PAUSE")) c->share_may_pause = std::atoi(r) != 0;
    if (const char* r = std::getenv("PSA_SCRATCH_CAP")) c->scratch_cap = (size_t)std::atoll(r);     // bytes; sends multi-chunk searches through the piece / sequential paths
#endif
    *out = c;
    return PSA_OK;
}

// Searches / batches of this context that the sequential engine answered because the filter engine's per-chunk counts
// would have exceeded the context's cap (16 GB, half of the device's free memory at psa_create, less after a failed
// allocation), and how often a failed allocation lowered that cap.
// Slices of shared combines (offset shards: psa_count_kernel) that a tile's last arriver had to take over because their
// own workgroup had waited in vain — 0 on a GPU this context has to itself; a shared GPU makes it grow (and such searches
// slower: the tile's workgroups were not resident together).  Reads one word from the device (synchronises the stream).
long long psa_share_sweep_count(psa_context* c)
{
    if (!c || !c->slot[0].d_share) return 0;
    DeviceScope device_scope;
    if (hipSetDevice(c->device) != hipSuccess) return -1;
    unsigned long long v = 0;
    constexpr size_t last = (size_t)psa::kShareMaxTiles * (1 + psa::kShareMaxChunks);
    if (hipStreamSynchronize(c->stream) != hipSuccess ||
        hipMemcpy(&v, c->slot[0].d_share + last, sizeof v, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (long long)v;
}
// How often this context has paused the sharing (64 launches each) because more than half of a search's slices had to be
// taken over: the GPU was shared with other work.
int psa_share_pause_count(psa_context* c) { return c ? c->share_pauses : 0; }
int psa_scratch_fallback_count(psa_context* c) { return c ? c->scratch_fallbacks : 0; }
int psa_scratch_shrink_count(psa_context* c) { return c ? c->scratch_shrinks : 0; }

void psa_destroy(psa_context* c)
{
    if (!c) return;
    DeviceScope device_scope;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->aux) (void)hipStreamSynchronize(c->aux);
    if (c->twin) (void)hipStreamSynchronize(c->twin);
    (void)hipFree(c->d_raw);
    (void)hipFree(c->d_s1);
    (void)hipFree(c->d_arena);               // tables, flag, control words, the slots' scalars / lists / records
    for (auto& s : c->slot) free_slot(s);
    (void)hipFree(c->d_batch_rows);
    (void)hipFree(c->d_batch_cls);
    (void)hipFree(c->d_batch_bytes);
    (void)hipFree(c->d_pool_scal);
    (void)hipFree(c->d_pool_cand);
    (void)hipFree(c->d_pool_cand_upper);
    (void)hipFree(c->d_pool_rec);
    for (int k = 0; k < 2; ++k) {
        (void)hipFree(c->d_ring_partial[k]);
        (void)hipFree(c->d_ring_arrivals[k]);
    }
    for (auto& ev : c->pool_events) (void)hipEventDestroy(ev);
    if (c->h_tails && !c->tails_in_arena) (void)hipHostFree(c->h_tails);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->h_arena) (void)hipHostFree(c->h_arena);
    for (auto& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (c->aux) (void)hipStreamDestroy(c->aux);
    if (c->twin) (void)hipStreamDestroy(c->twin);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

psa_status psa_set_weights(psa_context* c, const double weights[4], int is_max)
{
    if (!c || !weights) { set_last_error("psa_set_weights: NULL argument"); return PSA_ERR_ARG; }
    // the same weights (bit for bit) and mode as last time: the tables on the device are current
    if (c->have_weights && (c->tables.is_max != 0) == (is_max != 0) &&
        std::memcmp(c->tables.weights, weights, sizeof(double) * 4) == 0)
        return PSA_OK;
    std::string err;
    psa::Tables t;
    if (!psa::build_tables(weights, is_max, &t, &err)) { set_last_error(err); return PSA_ERR_ARG; }
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    c->tables = t;
    c->have_alt = false;
    PSA_TRY(upload_tables(c));
    c->have_weights = true;
    return PSA_OK;
}

psa_status psa_set_seq1_device(psa_context* c, const void* d_seq1, int n)
{
    if (!c || !d_seq1 || n < 1) { set_last_error("psa_set_seq1: bad arguments"); return PSA_ERR_ARG; }
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    return encode_seq1(c, d_seq1, n);      // alphabet errors surface at psa_search_wait
}

psa_status psa_set_seq2_device(psa_context* c, const void* d_seq2, int m)
{
    if (!c || !d_seq2 || m < 1) { set_last_error("psa_set_seq2: bad arguments"); return PSA_ERR_ARG; }
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    return encode_seq2(c, c->slot[0], d_seq2, m, c->stream);
}

psa_status psa_set_seq1(psa_context* c, const char* seq1, int n)
{
    if (!c || !seq1 || n < 1) { set_last_error("psa_set_seq1: bad arguments"); return PSA_ERR_ARG; }
    PSA_TRY(check_seq1_length(n));
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    PSA_TRY(stage_pinned_pair(c, seq1, (size_t)n, seq1 + n, 0));
    PSA_TRY(encode_seq1(c, c->hd_stage, n, true));          // reads the pinned host copy over PCIe
    PSA_TRY(wait_done(c));
    return report_bad(c);
}

psa_status psa_set_seq2(psa_context* c, const char* seq2, int m)
{
    if (!c || !seq2 || m < 1) { set_last_error("psa_set_seq2: bad arguments"); return PSA_ERR_ARG; }
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    PSA_TRY(stage_pinned_pair(c, seq2, (size_t)m, seq2 + m, 0));
    PSA_TRY(encode_seq2(c, c->slot[0], c->hd_stage, m, c->stream, true));
    PSA_TRY(wait_done(c));
    return report_bad(c);
}

// Both sequences of a search at once: one copy command each (host variant) and ONE encode launch.
psa_status psa_set_sequences_device(psa_context* c, const void* d_seq1, int n, const void* d_seq2, int m)
{
    if (!c || !d_seq1 || !d_seq2 || n < 1 || m < 1) { set_last_error("psa_set_sequences: bad arguments"); return PSA_ERR_ARG; }
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    return encode_pair(c, d_seq1, n, d_seq2, m);      // alphabet errors surface at psa_search_wait
}

psa_status psa_set_sequences(psa_context* c, const char* seq1, int n, const char* seq2, int m)
{
    if (!c || !seq1 || !seq2 || n < 1 || m < 1) { set_last_error("psa_set_sequences: bad arguments"); return PSA_ERR_ARG; }
    PSA_TRY(check_seq1_length(n));
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    if ((size_t)n >= stream_min()) {
        // long sequences: the encode launch goes first and follows the host's copy chunk by chunk (stream_pair)
        int got = 0;
        c->stream_gave_up = false;
        PSA_TRY(stream_pair(c, seq1, (size_t)n, false, seq2, (size_t)m, &got));
        PSA_TRY(wait_done(c));
        const psa_status st = report_bad(c);
        if (!c->stream_gave_up) return st;
        c->stream_gave_up = false;                  // (never expected: the host thread stalled for > 0.1 s mid-copy)
    }
    PSA_TRY(stage_pinned_pair(c, seq1, (size_t)n, seq2, (size_t)m));
    PSA_TRY(encode_pair(c, c->hd_stage, n, c->hd_stage + c->stage_second, m, true));     // reads the host copy over PCIe
    PSA_TRY(wait_done(c));
    return report_bad(c);
}

// Count launches of a batch run on one stream by default; 2 alternates them between two streams (the next launch's
// workgroups fill the CUs the running one leaves idle at its edges: worth 9 % with the round-1 kernels, within
// noise with the round-2 ones, whose launches are back to back on 244 of 256 CUs).
psa_status psa_set_batch_streams(psa_context* c, int streams)
{
    if (!c || streams < 1 || streams > 2) { set_last_error("psa_set_batch_streams: 1 or 2"); return PSA_ERR_ARG; }
    c->batch_streams = streams;
    return PSA_OK;
}

psa_status psa_set_engine(psa_context* c, psa_engine engine)
{
    if (!c) { set_last_error("context is NULL"); return PSA_ERR_ARG; }
    c->engine = engine;
    return PSA_OK;
}

// One filter search on slot 0 = ONE launch: the count kernel's last tile workgroup re-scores the candidates and
// writes the tail (finish_search).  Only a Seq2 whose class codes do not fit that workgroup's LDS is followed by
// psa_rescore_kernel right away.
static psa_status enqueue_filter_search(psa_context* c, int first, int last, bool profile)
{
    SearchSlot& s = c->slot[0];
    FilterPlan plan;
    PSA_TRY(plan_on_slot(s, first, last, c->stream, &plan));
    PSA_STRACE(1);
    const bool fused = psa::fused_candidates_at_a_time(plan.lds_bytes, s.m) >= 1;
    PSA_TRY(launch_filter_front(c, s, plan, first, last, c->stream, profile, primary_tables(c), nullptr, nullptr, fused));
    if (!fused) {
        c->unfused += 1;
        PSA_TRY(launch_filter_back(c, s, (long long)(last - first) * s.m, c->stream, primary_tables(c)));
    }
    return PSA_OK;
}

static psa_status search_async_impl(psa_context* c, int first, int last, psa_result* record_out, int record_shard)
{
    if (!c) { set_last_error("context is NULL"); return PSA_ERR_ARG; }
    PSA_STRACE(0);
    SearchSlot& s = c->slot[0];
    PSA_TRY(check_ready(c, s.m, first, last));
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    c->timing_valid = false;
    c->seq += 1;
    s.d_tail = c->hd_tails;
    c->record_out = record_out;
    c->record_shard = record_shard;
    if (c->profiling) PSA_HIP(hipEventRecord(c->ev[0], c->stream));
    c->last_kernel_queries = 1;
    reserve_chunk_scratch(c, first, last, s.m, false);
    c->pending_filter = use_filter(c, first, last, s.m);
    if (!c->pending_filter && capped_out(c, first, last, s.m)) c->scratch_fallbacks += 1;
    c->pending_first = first;
    c->pending_last = last;
    if (c->pending_filter) {
        // the search's last workgroup leaves the scalars zeroed; only a search that was enqueued but never
        // waited for, failed or fell back can have left them dirty
        if (s.scal_dirty) {
            PSA_HIP(hipMemsetAsync(s.d_scal, 0, sizeof(Scalars), c->stream));
            PSA_HIP(hipMemsetAsync(s.d_best_split, 0, sizeof(unsigned long long) * psa::kBestSplit * psa::kBestStride, c->stream));
        }
        s.scal_dirty = true;
        c->select_frozen_next = c->select_frozen;      // (debug library; the re-selection of settle_search runs normally)
        PSA_TRY(enqueue_filter_search(c, first, last, c->profiling));
    } else {
        PSA_TRY(launch_sequential(c, s, first, last, nullptr, nullptr, c->stream, primary_tables(c)));
    }
    if (c->profiling) {
        PSA_HIP(hipEventRecord(c->ev[1], c->stream));
        c->timing_valid = true;
    }
    c->pending = true;
    return PSA_OK;
}

psa_status psa_search_async(psa_context* c, int first, int last)
{
    return search_async_impl(c, first, last, nullptr, -1);
}

// The search in flight has been synchronised with: look at its tail and do what it asks for.
//   overflow 2  more surviving candidates than the count kernel's last workgroup takes: psa_rescore_kernel on the lists.
//   overflow 1  the candidate list did not hold everything picked against the RUNNING bound.  The bound left in the
//               search scalars is final now, so the selection runs once more against it; only if that overflows too
//               (more true near-ties than the list holds, e.g. a homopolymer) every offset is scored in reference
//               order by the sequential engine.
//   a tail that does not echo this search's number: no workgroup finished it (cannot happen for finite scores):
//               sequential engine as well.
// A sharing search has ended normally: how many of its slices had to be taken over by the tiles' last arrivers (the tail
// echoes the slot's sweep counter)?  More than half: the GPU is shared with other work — no sharing for a while.
constexpr int kSharePause = 64;
static void note_share_outcome(psa_context* c)
{
    if (c->pending_share_slices <= 0) return;
    const unsigned now = (unsigned)c->h_tails[0].swept;
    const unsigned taken = now - c->share_swept_seen;
    c->share_swept_seen = now;
    if (c->share_may_pause && (long long)taken * 2 > c->pending_share_slices) {
        c->share_pause = kSharePause;
        c->share_pauses += 1;
    }
    c->pending_share_slices = 0;
}

static psa_status settle_search(psa_context* c)
{
    SearchSlot& s = c->slot[0];
    const long long pairs = (long long)(c->pending_last - c->pending_first) * s.m;
    *c->h_bad = c->h_tails[0].bad;
    PSA_TRY(report_bad(c));                   // device-resident sequences are validated here
    if (!c->pending_filter) return PSA_OK;
    auto stale = [&]() { return c->h_tails[0].seq != c->seq; };
    auto rescore_if_asked = [&]() -> psa_status {
        if (stale() || c->h_tails[0].overflow != 2) return PSA_OK;
        c->timing_valid = false;
        c->unfused += 1;
        PSA_TRY(launch_filter_back(c, s, pairs, c->stream, primary_tables(c)));
        PSA_HIP(hipStreamSynchronize(c->stream));
        return PSA_OK;
    };
    PSA_TRY(rescore_if_asked());
    if (!stale() && c->h_tails[0].overflow == 1) {
        c->reselects += 1;
        c->seq += 1;
        c->timing_valid = false;
        PSA_TRY(enqueue_filter_search(c, c->pending_first, c->pending_last, false));
        PSA_HIP(hipStreamSynchronize(c->stream));
        PSA_TRY(rescore_if_asked());
    }
    if (stale() || c->h_tails[0].overflow) {
        c->fallbacks += 1;                        // (the bound is still in the scalars: scal_dirty stays set)
        c->seq += 1;
        c->timing_valid = false;
        PSA_TRY(launch_sequential(c, s, c->pending_first, c->pending_last, nullptr, nullptr, c->stream, primary_tables(c)));
        PSA_HIP(hipStreamSynchronize(c->stream));
    } else {
        s.scal_dirty = false;                     // the search ran to its end: its last workgroup zeroed the scalars
        note_share_outcome(c);
    }
    return PSA_OK;
}

// The search's last kernel writes its 48-byte tail straight into pinned host memory, its sequence number last (a
// system-scope release).  Watching that word returns the result a few microseconds before the stream's completion
// signal would (hipStreamSynchronize: kernel end, signal, runtime wake-up); a search that takes longer than a
// short spin falls back to it.  Everything the host touches afterwards is ordered behind the kernel by the stream.
// How long to watch the tail word before handing the wait to the runtime: a search the cost model (use_filter) puts far
// beyond the spin budget is not spun for at all (a 50 ms sequential-engine search would burn a core for nothing).
static double spin_budget_us(const psa_context* c, int first, int last, int m, bool filter)
{
    (void)c;
    const double pairs = (double)(last - first) * m;
    const double expect = filter ? 36.0 + 0.002 * m + pairs / 2.6e8 : 16.0 + 0.034 * m + pairs / 8.5e6;
    return expect <= 300.0 ? 400.0 : 0.0;
}

static bool tail_arrived(const psa_context* c, double spin_us)
{
    if (spin_us <= 0.0) return false;
    const volatile int* seq = &c->h_tails[0].seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int k = 0; k < 256; ++k) {
            if (*seq == c->seq) {
                std::atomic_thread_fence(std::memory_order_acquire);
                return true;
            }
            __builtin_ia32_pause();
        }
        if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_us) return false;
    }
}

// The stream's work has ended, or `deadline_s` seconds have passed (false).  hipStreamQuery between short sleeps: a wait
// that can be given up, unlike hipStreamSynchronize (device groups: a stalled device must not hang the call).
static bool stream_idle_within(hipStream_t stream, double deadline_s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned looks = 0;; ++looks) {
        const hipError_t e = hipStreamQuery(stream);
        if (e != hipErrorNotReady) return e == hipSuccess;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > deadline_s) return false;
        if (looks < 64u) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// deadline_s <= 0: wait as long as it takes (hipStreamSynchronize).  Otherwise the search in flight is given that long to
// end; PSA_ERR_DEVICE when it has not (the context keeps its `pending` mark: the kernel is still on the stream).
static psa_status search_wait_impl(psa_context* c, psa_result* out, double deadline_s)
{
    if (!c || !out) { set_last_error("psa_search_wait: NULL argument"); return PSA_ERR_ARG; }
    if (!c->pending) { set_last_error("psa_search_wait: no search in flight"); return PSA_ERR_STATE; }
    PSA_STRACE(4);
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));                // (a group waits for several devices' searches in turn; the retry
                                              // paths launch kernels)
    if (c->profiling ||
        !tail_arrived(c, spin_budget_us(c, c->pending_first, c->pending_last, c->slot[0].m, c->pending_filter))) {
        if (deadline_s <= 0.0) {
            PSA_HIP(hipStreamSynchronize(c->stream));
        } else if (!stream_idle_within(c->stream, deadline_s)) {
            set_last_error("the search on device " + std::to_string(c->device) + " has not ended within " +
                           std::to_string(deadline_s) + " s");
            return PSA_ERR_DEVICE;
        }
    }
    PSA_STRACE(5);
    c->pending = false;
    const psa_status st = settle_search(c);
    c->record_out = nullptr;
    c->record_shard = -1;
    PSA_TRY(st);
    *out = c->h_tails[0].result;
    PSA_STRACE(6);
    PSA_STRACE_PRINT();
    if (c->profiling && c->timing_valid) {
        float a = 0.f, b = 0.f;
        PSA_HIP(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
        PSA_HIP(hipEventElapsedTime(&b, c->ev[2], c->ev[3]));
        c->sum_search_ms += a;
        c->sum_kernel_ms += b;
        c->profiled += 1;
    }
    return PSA_OK;
}

psa_status psa_search_wait(psa_context* c, psa_result* out) { return search_wait_impl(c, out, 0.0); }

// One-shot search from raw ASCII in device memory: ONE launch, no encode launch in front of it — the count kernel
// converts Seq1's windows while it builds its planes, every workgroup writes Seq2's class-ordered symbols itself, and
// the search's last workgroup re-scores the survivors from the ASCII.  Anything the one launch does not settle (a
// character outside [A-Z-], candidate-list overflow, more than 16 survivors, the sequential engine being the better
// one for the shape) goes through psa_set_sequences_device + psa_search.  The context's own sequences are replaced
// (forgotten after a one-launch search, set after a fallback).
psa_status psa_search_device(psa_context* c, const void* d_seq1, int n, const void* d_seq2, int m,
                             int first, int last, psa_result* out)
{
    if (!c || !d_seq1 || !d_seq2 || !out || n < 1 || m < 1) { set_last_error("psa_search_device: bad arguments"); return PSA_ERR_ARG; }
    if (!c->have_weights) { set_last_error("weights must be set before a search"); return PSA_ERR_STATE; }
    if (c->pending) { set_last_error("psa_search_device: a psa_search_async is still in flight"); return PSA_ERR_STATE; }
    const long long total = (long long)n - m + 1;
    if (total < 1) { set_last_error("need 1 <= len(seq2) <= len(seq1)"); return PSA_ERR_ARG; }
    if (n > PSA_MAX_SEQ1) { set_last_error("seq1 is longer than PSA_MAX_SEQ1 (2^30)"); return PSA_ERR_ARG; }
    if (first < 0 || last > total || first >= last) {
        set_last_error("offset range must satisfy 0 <= first < last <= len(seq1)-len(seq2)+1");
        return PSA_ERR_ARG;
    }
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    SearchSlot& s = c->slot[0];
    auto two_launches = [&]() -> psa_status {
        PSA_TRY(encode_pair(c, d_seq1, n, d_seq2, m));
        return psa_search(c, first, last, out);
    };
    FilterPlan plan;
    reserve_chunk_scratch(c, first, last, m, false);
    if (c->profiling || !use_filter(c, first, last, m) || !plan_filter(first, last, m, &plan) ||
        psa::fused_candidates_at_a_time(plan.lds_bytes, m) < 1)
        return two_launches();
    PSA_TRY(ensure_seq2_capacity(s, m));
    c->n = -1;                                     // what the context held is gone either way
    s.m = m;
    s.s2rows = s.d_s2rows;
    s.s2bytes = s.d_s2bytes;
    s.s2cls = s.d_s2cls;
    s.cls_chars = plan.class_chars;                // (the launch writes the class order itself)
    s.cls_chunks = plan.nchunks;
    if (plan.nchunks > 1) PSA_TRY(ensure_chunk_scratch(&s.d_partial, &s.partial_cap, &s.d_arrivals, &s.arrivals_cap, plan, c->stream));
    c->timing_valid = false;
    c->seq += 1;
    s.d_tail = c->hd_tails;
    c->record_out = nullptr;
    c->record_shard = -1;
    c->last_kernel_queries = 1;
    if (s.scal_dirty) {
        PSA_HIP(hipMemsetAsync(s.d_scal, 0, sizeof(Scalars), c->stream));
        PSA_HIP(hipMemsetAsync(s.d_best_split, 0, sizeof(unsigned long long) * psa::kBestSplit * psa::kBestStride, c->stream));
    }
    s.scal_dirty = true;
    const RawSequences raw = {static_cast<const unsigned char*>(d_seq1), n, static_cast<const unsigned char*>(d_seq2), m};
    const psa_status launched = launch_filter_front(c, s, plan, first, last, c->stream, false, primary_tables(c), nullptr, nullptr,
                                                    true, &raw);
    s.m = -1;                                      // the slot holds no encoded Seq2
    PSA_TRY(launched);
    if (!tail_arrived(c, spin_budget_us(c, first, last, m, true))) PSA_HIP(hipStreamSynchronize(c->stream));
    const psa::SearchTail& t = c->h_tails[0];
    if (t.seq == c->seq && t.overflow == 0 && t.bad == 0) {
        s.scal_dirty = false;                      // the search ran to its end: its last workgroup zeroed the scalars
        *out = t.result;
        return PSA_OK;
    }
    // not settled by the one launch: the two-launch path decides (and reports alphabet errors)
    PSA_HIP(hipStreamSynchronize(c->stream));
    PSA_HIP(hipMemsetAsync(c->d_bad, 0, sizeof(int), c->stream));
    *c->h_bad = 0;
    c->raw_retries += 1;
    return two_launches();
}

// A long Seq2 (several char chunks) over so many offsets that the chunks' counts would not fit the context's scratch_cap: the range
// is searched piece by piece (each a complete filter search) and the pieces' winners are folded in the is_swapable order
// (psa_fold: ties go to the lower offset, i.e. the earlier piece).  0 = no pieces needed.
static int piece_offsets(const psa_context* c, int first, int last)
{
    if (c->engine == PSA_ENGINE_SEQUENTIAL) return 0;
    const int m = c->slot[0].m;
    FilterPlan p;
    if (m < 1 || !plan_filter(first, last, m, &p) || chunk_scratch_bytes(p) <= c->scratch_cap) return 0;
    long long piece = (long long)last - first;
    while (piece > psa::kTileStride) {
        piece = ((piece / 2 + psa::kTileStride - 1) / psa::kTileStride) * psa::kTileStride;
        if (plan_filter(first, first + (int)piece, m, &p) && chunk_scratch_bytes(p) <= c->scratch_cap) return (int)piece;
    }
    return 0;                                      // (one tile's chunks beyond the cap: cannot happen; the sequential engine)
}

psa_status psa_search(psa_context* c, int first, int last, psa_result* out)
{
    if (c && out && !c->pending && c->have_weights && c->n >= 0 && check_ready(c, c->slot[0].m, first, last) == PSA_OK) {
        reserve_chunk_scratch(c, first, last, c->slot[0].m, false);
        const int piece = piece_offsets(c, first, last);
        if (piece > 0) {
            psa_result best;
            bool have = false;
            for (long long lo = first; lo < last; lo += piece) {
                const int hi = (int)std::min<long long>(lo + piece, last);
                psa_result r[2];
                PSA_TRY(psa_search_async(c, (int)lo, hi));
                PSA_TRY(psa_search_wait(c, &r[1]));
                if (have) { r[0] = best; PSA_TRY(psa_fold(r, 2, c->tables.is_max, &best)); }
                else best = r[1];
                have = true;
            }
            *out = best;
            return PSA_OK;
        }
    }
    PSA_TRY(psa_search_async(c, first, last));
    return psa_search_wait(c, out);
}

// nq queries of m characters each, `stride` bytes apart in one device buffer, against the
// context's Seq1 and weights; results in host memory.  results1 != NULL asks for the opposite
// mode as well (MIN next to MAX): the class counts of an offset do not depend on the mode, so
// both searches of a query share one count pass and differ only in bounds, candidates and
// re-score.  Count kernels of consecutive queries alternate between two streams; the re-score
// and read-back of every query run on a third.
static psa_status batch_once(psa_context* c, const void* d_queries, int nq, int m, size_t stride,
                             int first, int last, psa_result* results0, psa_result* results1,
                             bool allow_duo, bool* duo_timed_out)
{
    *duo_timed_out = false;
    if (!c || !d_queries || !results0 || nq < 1 || stride < (size_t)m) {
        set_last_error("psa_search_batch: bad arguments");
        return PSA_ERR_ARG;
    }
    if (c->pending) { set_last_error("psa_search_batch: a psa_search_async is still in flight"); return PSA_ERR_STATE; }
    PSA_TRY(check_ready(c, m, first, last));
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    PSA_TRY(ensure_batch_objects(c));
    const int nmodes = results1 ? 2 : 1;
    if (nmodes == 2) PSA_TRY(ensure_alt_tables(c));
    const TableSet tables[2] = {primary_tables(c), alternate_tables(c)};
    psa_result* const results[2] = {results0, results1};
    // device-side record blocks of a group batch (mode 0 = the context's own mode; the blocks are [MAX or only, MIN])
    const bool swap_blocks = nmodes == 2 && c->tables.is_max == 0;
    psa_result* const dev_records[2] = {c->batch_records[swap_blocks ? 1 : 0], nmodes == 2 ? c->batch_records[swap_blocks ? 0 : 1] : nullptr};
    bool host_touched = false;
    c->batch_records_final = false;
    PSA_TRY(ensure_tails(c, nq * nmodes));
    const unsigned char* base = static_cast<const unsigned char*>(d_queries);
    {
        FilterPlan whole;                         // (a batch of >= 2 queries prefers a plan without char chunks: nothing to reserve)
        if (!(nq >= 2 && plan_filter(first, last, m, &whole, 1))) reserve_chunk_scratch(c, first, last, m, true);
    }
    const bool filter = use_filter(c, first, last, m);
    if (!filter && capped_out(c, first, last, m)) c->scratch_fallbacks += 1;
    const long long pairs = (long long)(last - first) * m;
    const int seq0 = c->seq + 1;
    c->timing_valid = false;
    if (c->profiling) PSA_HIP(hipEventRecord(c->ev[0], c->stream));
    // all queries encoded by one launch: [nq][ostride] codes as u32 rows and as bytes
    const int ostride = (m + seq2_slack(m) + 63) & ~63;
    if ((size_t)nq * ostride > c->batch_cap) {
        c->batch_cap = 0;
        PSA_TRY(dev_alloc(&c->d_batch_rows, (size_t)nq * ostride));
        PSA_TRY(dev_alloc(&c->d_batch_cls, (size_t)nq * ostride));
        PSA_TRY(dev_alloc(&c->d_batch_bytes, (size_t)nq * ostride));
        c->batch_cap = (size_t)nq * ostride;
    }
    // filter searches keep all per-query state in pools (entry mode*nq + q): nothing is
    // reused inside the batch and the count kernels never wait for the re-score stream
    // A batch spreads over the GPU by its queries (the batch kernel's grid is tiles x query blocks), so it keeps all
    // chars of a query in one workgroup whenever they fit; char chunks are for a Seq2 too long for that.
    FilterPlan plan;
    bool pooled = filter && nq >= 2 && plan_filter(first, last, m, &plan, 1);
    if (!pooled) pooled = filter && plan_filter(first, last, m, &plan);      // filter => a plan exists (use_filter)
    static_assert(sizeof(Scalars) % sizeof(unsigned long long) == 0, "the encode kernel zeroes the pool in 8-byte words");
    int zero_words = 0;
    if (pooled) {
        PSA_TRY(ensure_pool(c, nq * nmodes));
        zero_words = (int)(sizeof(Scalars) / sizeof(unsigned long long)) * nq * nmodes;     // << threads of the encode grid
        if (plan.nchunks > 1)
            for (int k = 0; k < 2; ++k) {
                size_t cap = c->ring_partial_cap;
                int acap = c->ring_arrivals_cap;
                PSA_TRY(ensure_chunk_scratch(&c->d_ring_partial[k], &cap, &c->d_ring_arrivals[k], &acap, plan, c->stream));
                if (k == 1) { c->ring_partial_cap = cap; c->ring_arrivals_cap = acap; }
            }
    } else {
        for (auto& s : c->slot) PSA_HIP(hipMemsetAsync(s.d_scal, 0, sizeof(Scalars), c->stream));
    }
    hipLaunchKernelGGL(psa::psa_encode_batch_kernel, dim3((ostride + 255) / 256, nq), dim3(256), 0, c->stream,
                       base, nq, m, stride, c->d_batch_rows, c->d_batch_bytes, c->d_batch_cls, ostride,
                       pooled ? plan.class_chars : 0, pooled ? plan.nchunks * psa::kClasses * plan.class_chars : 0, c->d_bad,
                       reinterpret_cast<unsigned long long*>(c->d_pool_scal), zero_words);
    PSA_HIP(hipGetLastError());
    // the other streams must see everything already enqueued on the main stream (Seq1 codes, tables, query codes)
    hipEvent_t& gate = c->slot[0].front_done;
    PSA_HIP(hipEventRecord(gate, c->stream));
    PSA_HIP(hipStreamWaitEvent(c->aux, gate, 0));
    // pooled queries share nothing, so their count kernels alternate between two streams: the next
    // kernel's workgroups fill the CUs the previous one leaves idle (245 tiles on 256 CUs, launch gaps)
    const bool twin = pooled && c->batch_streams == 2;
    if (twin) PSA_HIP(hipStreamWaitEvent(c->twin, gate, 0));
    // sequence number echoed by the tail of (mode, q)
    auto seq_of = [&](int mode, int q) { return pooled ? seq0 + q : seq0 + mode * nq + q; };
    auto pool_view = [&](int mode, int q) {
        SearchSlot v = c->slot[0];            // pointers only; the pool overrides the per-search buffers
        const size_t e = (size_t)mode * nq + q;
        v.d_scal = c->d_pool_scal + e;
        v.d_cand = c->d_pool_cand + e * kCandCap;
        v.d_cand_upper = c->d_pool_cand_upper + e * kCandCap;
        v.d_rec = c->d_pool_rec + e * kCandCap;
        v.d_tail = c->hd_tails + e;
        v.front_done = c->pool_events[e];
        v.m = m;
        v.s2rows = c->d_batch_rows + (size_t)q * ostride;
        v.s2bytes = c->d_batch_bytes + (size_t)q * ostride;
        v.s2cls = c->d_batch_cls + (size_t)q * ostride;
        v.d_partial = c->d_ring_partial[q & 1];          // queries q and q+2 run on the same stream
        v.d_arrivals = c->d_ring_arrivals[q & 1];
        return v;
    };
    int duo_wc = 0, duo_slots = 0;
    const bool duo = allow_duo && pooled && c->duo && nq >= 2 && plan.nchunks == 1 && plan.waves == 16 &&
                     duo_fits(m, &duo_wc, &duo_slots);
    if (duo) {
        // several queries per count launch (two wave groups per tile share the planes and alternate between
        // main loop and epilogue); launches alternate between the two streams, the re-scores follow on the third
        int launch = 0;
        const int per_launch = duo_queries_per_launch(c, plan.ntiles);
        for (int q0 = 0; q0 < nq; q0 += per_launch, ++launch) {
            const int count = std::min(per_launch, nq - q0);
            hipStream_t front = twin && (launch & 1) ? c->twin : c->stream;
            Scalars* scal[2];
            int* cand[2];
            double* cand_upper[2];
            for (int mode = 0; mode < 2; ++mode) {
                const size_t e = (size_t)(mode < nmodes ? mode : 0) * nq + q0;
                scal[mode] = c->d_pool_scal + e;
                cand[mode] = c->d_pool_cand + e * kCandCap;
                cand_upper[mode] = c->d_pool_cand_upper + e * kCandCap;
            }
            const bool timed = c->profiling && launch == 0;          // launch 0 runs on the main stream
            if (timed) PSA_HIP(hipEventRecord(c->ev[2], front));
            PSA_TRY(launch_count_duo(c, c->d_batch_cls + (size_t)q0 * ostride, (size_t)ostride, m, count, first, last,
                                     nmodes, tables, scal, cand, cand_upper, front));
            if (timed) {
                PSA_HIP(hipEventRecord(c->ev[3], front));
                c->last_kernel = "psa_count_duo_kernel";
                c->last_kernel_queries = count;
            }
            hipEvent_t done = c->pool_events[q0];
            PSA_HIP(hipEventRecord(done, front));
            PSA_HIP(hipStreamWaitEvent(c->aux, done, 0));
            // one re-score launch per mode for the whole group of queries (grid.y = query)
            c->seq = seq0 + q0;
            for (int mode = 0; mode < nmodes; ++mode) {
                SearchSlot s = pool_view(mode, q0);
                s.inline_select = true;
                PSA_TRY(launch_filter_back(c, s, pairs, c->aux, tables[mode], c->batch_rescore_grid, count, (size_t)ostride,
                                           dev_records[mode] ? dev_records[mode] + q0 : nullptr));
            }
            c->seq = seq0 + q0 + count - 1;
        }
    } else if (pooled) {
        for (int q = 0; q < nq; ++q) {
            hipStream_t front = twin && (q & 1) ? c->twin : c->stream;
            SearchSlot s0 = pool_view(0, q);
            SearchSlot s1 = pool_view(nmodes - 1, q);
            c->seq = seq0 + q;
            PSA_TRY(launch_filter_front(c, s0, plan, first, last, front, c->profiling && q == 0, tables[0],
                                        nmodes == 2 ? &s1 : nullptr, &tables[1]));
            if (q == 0) c->last_kernel_queries = 1;
            PSA_HIP(hipEventRecord(s0.front_done, front));
            PSA_HIP(hipStreamWaitEvent(c->aux, s0.front_done, 0));
            PSA_TRY(launch_filter_back(c, s0, pairs, c->aux, tables[0], c->batch_rescore_grid, 0, 0,
                                       dev_records[0] ? dev_records[0] + q : nullptr));
            if (nmodes == 2) {
                PSA_TRY(launch_filter_back(c, s1, pairs, c->aux, tables[1], c->batch_rescore_grid, 0, 0,
                                           dev_records[1] ? dev_records[1] + q : nullptr));
            }
        }
    } else {
        for (int mode = 0; mode < nmodes; ++mode)
            for (int q = 0; q < nq; ++q) {
                const int k = mode * nq + q;
                SearchSlot& s = c->slot[k % kSlots];
                if (k >= kSlots) PSA_HIP(hipStreamWaitEvent(c->stream, s.back_done, 0));   // slot free again
                s.m = m;
                s.d_tail = c->hd_tails + k;
                s.s2rows = c->d_batch_rows + (size_t)q * ostride;
                s.s2bytes = c->d_batch_bytes + (size_t)q * ostride;
                c->seq = seq_of(mode, q);
                PSA_TRY(launch_sequential(c, s, first, last, nullptr, nullptr, c->stream, tables[mode]));
                PSA_HIP(hipEventRecord(s.front_done, c->stream));
                PSA_HIP(hipStreamWaitEvent(c->aux, s.front_done, 0));
                PSA_HIP(hipEventRecord(s.back_done, c->aux));
            }
    }
    PSA_HIP(hipStreamSynchronize(c->aux));
    if (twin) PSA_HIP(hipStreamSynchronize(c->twin));
    if (c->profiling && pooled) {
        PSA_HIP(hipEventRecord(c->ev[1], c->stream));      // whole batch; ev[2..3]: its first count launch
        c->timing_valid = true;
    }
    PSA_HIP(hipStreamSynchronize(c->stream));
    int bad = 0;
    for (int k = 0; k < nq * nmodes; ++k) bad |= c->h_tails[k].bad;
    if (duo && (bad & 4)) {
        // A group barrier of the batch kernel gave up (a hand-rolled LDS barrier with a spin limit: a wave was
        // held back far beyond any healthy delay).  Its counts cannot be trusted: the caller runs the batch
        // again with the per-query count kernel, whose barriers are the hardware's.
        PSA_HIP(hipMemsetAsync(c->d_bad, 0, sizeof(int), c->stream));
        PSA_HIP(hipStreamSynchronize(c->stream));
        *c->h_bad = 0;
        if (bad & 3) { *c->h_bad = bad & 3; return report_bad(c); }       // alphabet errors are real either way
        *duo_timed_out = true;
        return PSA_OK;
    }
    *c->h_bad = bad;
    PSA_TRY(report_bad(c));
    for (int mode = 0; mode < nmodes; ++mode)
        for (int q = 0; q < nq; ++q) {
            const int k = mode * nq + q;
            bool stale = filter && c->h_tails[k].seq != seq_of(mode, q);
            if (pooled && c->h_tails[k].overflow && !stale) {
                // overflow against the running bound: select once more against the final one (still in the
                // query's pool entry), with the per-query count kernel
                SearchSlot v = pool_view(mode, q);
                host_touched = true;
                c->reselects += 1;
                c->seq += 1;
                PSA_TRY(launch_filter_front(c, v, plan, first, last, c->stream, false, tables[mode]));
                PSA_TRY(launch_filter_back(c, v, pairs, c->stream, tables[mode]));
                PSA_HIP(hipStreamSynchronize(c->stream));
                stale = c->h_tails[k].seq != c->seq;
            }
            if (filter && (c->h_tails[k].overflow || stale)) {
                SearchSlot& s = c->slot[0];
                host_touched = true;
                c->fallbacks += 1;
                c->seq += 1;
                for (auto& sl : c->slot) sl.scal_dirty = true;
                s.m = m;
                s.d_tail = c->hd_tails + k;
                s.s2rows = c->d_batch_rows + (size_t)q * ostride;
                s.s2bytes = c->d_batch_bytes + (size_t)q * ostride;
                PSA_TRY(launch_sequential(c, s, first, last, nullptr, nullptr, c->stream, tables[mode]));
                PSA_HIP(hipStreamSynchronize(c->stream));
            }
            results[mode][q] = c->h_tails[k].result;
            results[mode][q].shard = q;
        }
    for (auto& s : c->slot) {
        s.m = -1;                          // the single-search Seq2 is gone
        s.d_tail = c->hd_tails;
    }
    c->batch_records_final = pooled && dev_records[0] != nullptr && !host_touched;
    return PSA_OK;
}

static psa_status batch_impl(psa_context* c, const void* d_queries, int nq, int m, size_t stride,
                             int first, int last, psa_result* results0, psa_result* results1)
{
    bool timed_out = false;
    PSA_TRY(batch_once(c, d_queries, nq, m, stride, first, last, results0, results1, true, &timed_out));
    if (!timed_out) return PSA_OK;
    c->duo_retries += 1;
    return batch_once(c, d_queries, nq, m, stride, first, last, results0, results1, false, &timed_out);
}

psa_status psa_search_batch_device(psa_context* c, const void* d_queries, int nq, int m, size_t stride,
                                   int first, int last, psa_result* results)
{
    return batch_impl(c, d_queries, nq, m, stride, first, last, results, nullptr);
}

psa_status psa_search_batch(psa_context* c, const char* queries, int nq, int m, size_t stride,
                            int first, int last, psa_result* results)
{
    if (!c || !queries || nq < 1 || stride < (size_t)m) { set_last_error("psa_search_batch: bad arguments"); return PSA_ERR_ARG; }
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    // (a DMA copy, not the pinned staging the single-search setters use: the batch encode kernel reads its queries a byte
    // per lane, which over PCIe would be a request per byte)
    PSA_TRY(stage_host(c, queries, (size_t)(nq - 1) * stride + (size_t)m));
    PSA_HIP(hipStreamSynchronize(c->stream));
    return batch_impl(c, c->d_raw, nq, m, stride, first, last, results, nullptr);
}

psa_status psa_search_batch_minmax_device(psa_context* c, const void* d_queries, int nq, int m, size_t stride,
                                          int first, int last, psa_result* results_max, psa_result* results_min)
{
    if (!c || !results_max || !results_min) { set_last_error("psa_search_batch_minmax: NULL argument"); return PSA_ERR_ARG; }
    if (!c->have_weights) { set_last_error("weights, seq1 and seq2 must be set before a search"); return PSA_ERR_STATE; }
    const bool mx = c->tables.is_max != 0;      // results0 is the context's own mode
    return batch_impl(c, d_queries, nq, m, stride, first, last, mx ? results_max : results_min, mx ? results_min : results_max);
}

psa_status psa_search_batch_minmax(psa_context* c, const char* queries, int nq, int m, size_t stride,
                                   int first, int last, psa_result* results_max, psa_result* results_min)
{
    if (!c || !queries || nq < 1 || stride < (size_t)m) { set_last_error("psa_search_batch_minmax: bad arguments"); return PSA_ERR_ARG; }
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    PSA_TRY(stage_host(c, queries, (size_t)(nq - 1) * stride + (size_t)m));
    PSA_HIP(hipStreamSynchronize(c->stream));
    return psa_search_batch_minmax_device(c, c->d_raw, nq, m, stride, first, last, results_max, results_min);
}

psa_status psa_scores(psa_context* c, int first, int last, double* scores, int32_t* char_offsets, int32_t* chs)
{
    if (!c) { set_last_error("context is NULL"); return PSA_ERR_ARG; }
    SearchSlot& s = c->slot[0];
    PSA_TRY(check_ready(c, s.m, first, last));
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    const int span = last - first;
    double* d_score = nullptr;
    unsigned char* d_prank = nullptr;
    int* d_ci = nullptr;
    int* d_ch = nullptr;
    psa_status st = [&]() -> psa_status {
        PSA_HIP(hipMalloc(reinterpret_cast<void**>(&d_score), sizeof(double) * (size_t)span));
        PSA_HIP(hipMalloc(reinterpret_cast<void**>(&d_prank), (size_t)span));
        PSA_HIP(hipMalloc(reinterpret_cast<void**>(&d_ci), sizeof(int) * (size_t)span));
        PSA_HIP(hipMalloc(reinterpret_cast<void**>(&d_ch), sizeof(int) * (size_t)span));
        PSA_TRY(launch_sequential(c, s, first, last, d_score, d_prank, c->stream, primary_tables(c)));
        hipLaunchKernelGGL(psa::psa_first_index_kernel, dim3((span + 255) / 256), dim3(256), 0, c->stream,
                           c->d_s1, s.s2rows, c->d_rank, c->d_sub, d_prank, first, last, s.m, d_ci, d_ch);
        PSA_HIP(hipGetLastError());
        if (scores) PSA_HIP(hipMemcpyAsync(scores, d_score, sizeof(double) * (size_t)span, hipMemcpyDeviceToHost, c->stream));
        if (char_offsets) PSA_HIP(hipMemcpyAsync(char_offsets, d_ci, sizeof(int) * (size_t)span, hipMemcpyDeviceToHost, c->stream));
        if (chs) PSA_HIP(hipMemcpyAsync(chs, d_ch, sizeof(int) * (size_t)span, hipMemcpyDeviceToHost, c->stream));
        PSA_HIP(hipMemcpyAsync(c->h_bad, c->d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        PSA_HIP(hipStreamSynchronize(c->stream));
        return report_bad(c);
    }();
    (void)hipFree(d_score);
    (void)hipFree(d_prank);
    (void)hipFree(d_ci);
    (void)hipFree(d_ch);
    return st;
}

#ifdef PSA_EXPERIMENT
// (test hooks: exported by libpsa_debug.so only, declared in include/psa_debug.h)
// Diagnostic view of the filter engine's exact class counts: n_star / n_colon /
// n_dot of every offset in [first,last).
psa_status psa_debug_counts(psa_context* c, int first, int last, uint32_t* n_star, uint32_t* n_colon,
                            uint32_t* n_dot)
{
    if (!c) { set_last_error("context is NULL"); return PSA_ERR_ARG; }
    SearchSlot& s = c->slot[0];
    PSA_TRY(check_ready(c, s.m, first, last));
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    const int span = last - first;
    const int base = first & ~31;
    const int lead = first - base;
    FilterPlan p;
    if (!plan_filter(first, last, s.m, &p)) { set_last_error("internal: no filter plan"); return PSA_ERR_ARG; }
    const int span_pad = p.ntiles * psa::kTile;
    const size_t need = (size_t)p.nchunks * span_pad;
    if (need > s.partial_cap) {
        s.partial_cap = 0;
        PSA_TRY(dev_alloc(&s.d_partial, need));
        s.partial_cap = need;
    }
    PSA_TRY(ensure_class_order(s, p, c->stream));
    PSA_TRY(launch_count(c, s, p, base, span_pad, first, span, false, c->stream, primary_tables(c)));
    std::vector<unsigned long long> host(need);
    PSA_HIP(hipMemcpyAsync(host.data(), s.d_partial, need * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    PSA_HIP(hipMemcpyAsync(c->h_bad, c->d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PSA_HIP(hipStreamSynchronize(c->stream));
    PSA_TRY(report_bad(c));
    for (int o = 0; o < span; ++o) {
        unsigned long long sum = 0;
        const int rel = lead + o;               // tile t answers for kTileStride offsets, stored kTile apart
        const size_t at = (size_t)(rel / psa::kTileStride) * psa::kTile + (size_t)(rel % psa::kTileStride);
        for (int k = 0; k < p.nchunks; ++k) sum += host[(size_t)k * span_pad + at];
        const uint32_t n_lo = (uint32_t)(sum & 0x1fffff), n_hi = (uint32_t)((sum >> 21) & 0x1fffff);
        const uint32_t n_both = (uint32_t)((sum >> 42) & 0x1fffff);
        if (n_star) n_star[o] = n_lo - n_both;
        if (n_colon) n_colon[o] = n_hi - n_both;
        if (n_dot) n_dot[o] = n_both;
    }
    return PSA_OK;
}
#endif  // PSA_EXPERIMENT

#ifdef PSA_PHASE_TIMING
// debug build only (tools/phase_timing.py): the count kernel's phase timestamps
PSA_EXPORT int psa_debug_phase_ts(unsigned long long* out /* [1024][8] */)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(psa::g_phase_ts), sizeof(unsigned long long) * 1024 * 8);
}
PSA_EXPORT int psa_debug_chunk_ts(unsigned long long* out /* [1024][16] */)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(psa::g_chunk_ts), sizeof(unsigned long long) * 1024 * 16);
}
#endif

#ifdef PSA_EXPERIMENT
// Diagnostic: the filter engine's exact re-score of arbitrary offsets
// (find_best_mutant_offset's return value, cpu_funcs.c:299) without the filter.
psa_status psa_debug_rescore(psa_context* c, const int32_t* offsets, int count, double* scores)
{
    if (!c || !offsets || !scores || count < 1 || count > kCandCap) {
        set_last_error("psa_debug_rescore: bad arguments (1 <= count <= 2048)");
        return PSA_ERR_ARG;
    }
    SearchSlot& s = c->slot[0];
    PSA_TRY(check_ready(c, s.m, 0, 1));
    DeviceScope device_scope;
    PSA_TRY(ensure_device(c));
    const int total = c->n - s.m + 1;
    for (int k = 0; k < count; ++k)
        if (offsets[k] < 0 || offsets[k] >= total) { set_last_error("psa_debug_rescore: offset out of range"); return PSA_ERR_ARG; }
    Scalars sc;
    sc.global_best = 0;
    sc.cand_count = count;
    sc.done = 0;
    s.inline_select = false;
    PSA_HIP(hipMemcpyAsync(s.d_scal, &sc, sizeof sc, hipMemcpyHostToDevice, c->stream));
    PSA_HIP(hipMemcpyAsync(s.d_cand, offsets, sizeof(int) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    PSA_TRY(launch_filter_back(c, s, 0, c->stream, primary_tables(c)));
    std::vector<psa::CandRec> host((size_t)count);
    PSA_HIP(hipMemcpyAsync(host.data(), s.d_rec, sizeof(psa::CandRec) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    PSA_HIP(hipMemcpyAsync(c->h_bad, c->d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PSA_HIP(hipStreamSynchronize(c->stream));
    PSA_TRY(report_bad(c));
    for (int k = 0; k < count; ++k) scores[k] = host[(size_t)k].score;
    return PSA_OK;
}
#endif  // PSA_EXPERIMENT

psa_status psa_set_profiling(psa_context* c, int on)
{
    if (!c) { set_last_error("context is NULL"); return PSA_ERR_ARG; }
    if (on) {
        DeviceScope device_scope;
        PSA_TRY(ensure_device(c));
        PSA_TRY(ensure_timing_events(c));
    }
    c->profiling = on != 0;
    c->timing_valid = false;
    c->sum_search_ms = c->sum_kernel_ms = 0.0;
    c->profiled = 0;
    return PSA_OK;
}

psa_status psa_profile_totals(psa_context* c, double* search_ms, double* kernel_ms, int* searches)
{
    if (!c) { set_last_error("context is NULL"); return PSA_ERR_ARG; }
    if (search_ms) *search_ms = c->sum_search_ms;
    if (kernel_ms) *kernel_ms = c->sum_kernel_ms;
    if (searches) *searches = c->profiled;
    return PSA_OK;
}

psa_status psa_last_timing(psa_context* c, int which, float* ms, const char** kernel_name)
{
    if (!c || !ms) { set_last_error("psa_last_timing: NULL argument"); return PSA_ERR_ARG; }
    if (!c->timing_valid) { set_last_error("psa_last_timing: profiling was off for the last search"); return PSA_ERR_STATE; }
    PSA_HIP(hipEventSynchronize(c->ev[1]));
    if (which == 0) PSA_HIP(hipEventElapsedTime(ms, c->ev[0], c->ev[1]));
    else PSA_HIP(hipEventElapsedTime(ms, c->ev[2], c->ev[3]));
    if (kernel_name) *kernel_name = c->last_kernel;
    return PSA_OK;
}

// Queries covered by the kernel launch psa_last_timing(which = 1) timed: 1 for a single search, the
// queries of the first count launch for a profiled batch.
int psa_last_kernel_queries(psa_context* c) { return c ? c->last_kernel_queries : 0; }

// How many filter searches of this context overflowed the candidate list and
// were redone by the sequential engine (tests; adversarial inputs only).
int psa_fallback_count(psa_context* c) { return c ? c->fallbacks : 0; }

// How many filter searches ran their candidate selection twice: the first pass picks against a running
// bound and overflowed the list, the second against the final bound (tests).
int psa_reselect_count(psa_context* c) { return c ? c->reselects : 0; }

// Single filter searches of this context that psa_rescore_kernel finished (a second launch): more surviving candidates
// than the count kernel's last workgroup re-scores itself (16), or a Seq2 whose class codes do not fit its LDS.
int psa_unfused_count(psa_context* c) { return c ? c->unfused : 0; }

// One-shot searches (psa_search_device) that the one launch did not settle and the two-launch path redid.
int psa_raw_retry_count(psa_context* c) { return c ? c->raw_retries : 0; }

// Batches this context ran a second time without the batch count kernel because one of its wave-group
// barriers timed out (never expected; the debug library can force it: PSA_DUO_FORCE_TIMEOUT).
int psa_duo_retry_count(psa_context* c) { return c ? c->duo_retries : 0; }


}  // extern "C"

// ---------------------------------------------------------------------------
// device group: one search sharded over several GPUs of the node
// (the reference's MPI split and merge, cpu_funcs.c:49-94 and 128-133, in one process)
// ---------------------------------------------------------------------------
namespace {

// RCCL entry points, resolved from librccl.so.1 when the first group is created: the single-GPU
// product path neither needs nor loads RCCL.
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll)  comm_init_all = nullptr;
    decltype(&ncclCommDestroy)  comm_destroy = nullptr;
    decltype(&ncclCommAbort)    comm_abort = nullptr;     // (optional: a group that gave up on a device)
    decltype(&ncclAllGather)    all_gather = nullptr;
    decltype(&ncclGroupStart)   group_start = nullptr;
    decltype(&ncclGroupEnd)     group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

bool load_rccl(std::string* err)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return true;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) { *err = std::string("cannot load librccl: ") + dlerror(); return false; }
    RcclApi api;
    api.handle = h;
    api.comm_init_all = reinterpret_cast<decltype(api.comm_init_all)>(dlsym(h, "ncclCommInitAll"));
    api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    api.comm_abort = reinterpret_cast<decltype(api.comm_abort)>(dlsym(h, "ncclCommAbort"));
    api.all_gather = reinterpret_cast<decltype(api.all_gather)>(dlsym(h, "ncclAllGather"));
    api.group_start = reinterpret_cast<decltype(api.group_start)>(dlsym(h, "ncclGroupStart"));
    api.group_end = reinterpret_cast<decltype(api.group_end)>(dlsym(h, "ncclGroupEnd"));
    api.error_string = reinterpret_cast<decltype(api.error_string)>(dlsym(h, "ncclGetErrorString"));
    if (!api.comm_init_all || !api.comm_destroy || !api.all_gather || !api.group_start || !api.group_end || !api.error_string) {
        *err = "librccl lacks an expected entry point";
        dlclose(h);
        return false;
    }
    g_rccl = api;
    return true;
}

#define PSA_NCCL(call)                                                                         \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            set_last_error(std::string(#call) + ": " + g_rccl.error_string(r_));               \
            return PSA_ERR_DEVICE;                                                             \
        }                                                                                      \
    } while (0)

}  // namespace

// ---- per-device launcher threads: csrc/psa_group_workers.hpp (host-only; ThreadSanitizer-tested on the CPU build) ----
using psa::GroupWorkers;

// ---- the exchange's last step on device 0, and the knobs of the fault tests ------------------------------------------
namespace {

// Device 0's copy of the gathered records (device memory, complete: the all-gather precedes this launch on the stream)
// -> pinned host memory the host is POLLING, then the exchange's number as a system-scope release store.  The host sees
// the records a stream-completion signal earlier than a D2H copy + hipStreamSynchronize would show them (round 4: +15 us
// per group call), and the wait can be given up: a stalled device cannot hang the caller.  Why not let RCCL receive
// straight into the pinned buffer: a record written by a library's kernel carries no "last word" the host could trust
// (a half-written 32-byte record must never be folded); here the flag is ordered behind every record.
__global__ void __launch_bounds__(256) psa_group_publish_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int pieces,
                                                                unsigned* flag, unsigned seq)
{
    for (int k = (int)threadIdx.x; k < pieces; k += (int)blockDim.x) dst[k] = src[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The flag word shows `want`, or deadline_s seconds have passed (false).
bool flag_arrived(const unsigned* flag, unsigned want, double deadline_s)
{
    const volatile unsigned* f = flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int k = 0; k < 256; ++k) {
            if (*f == want) {
                std::atomic_thread_fence(std::memory_order_acquire);
                return true;
            }
            __builtin_ia32_pause();
        }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > deadline_s) return false;
        if (waited > 2e-3) std::this_thread::sleep_for(std::chrono::microseconds(50));   // (a batch: not worth a core)
    }
}

// Fault injection of the group calls — the debug library only (libpsa_debug.so, -DPSA_EXPERIMENT), read per call:
//   PSA_GROUP_FAULT=launch:K      device K's job fails before its launch
//   PSA_GROUP_FAULT=exchange:K    ... after its launch, before its part of the exchange
//   PSA_GROUP_FAULT=stall:K:MS    a kernel that idles for MS milliseconds (bounded: it watches the wall clock) runs on
//                                 device K's stream in front of its search: a device that does not answer in time
struct GroupFault {
    int kind = 0;          // 1 launch, 2 exchange, 3 stall
    int device = -1;
    double stall_ms = 0.0;
    bool hits(int what, int i) const { return kind == what && device == i; }
};

#ifdef PSA_EXPERIMENT
__global__ void psa_debug_stall_kernel(unsigned long long ticks)          // 100 MHz ticks
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
#endif

GroupFault group_fault()
{
    GroupFault f;
#ifdef PSA_EXPERIMENT
    if (const char* e = std::getenv("PSA_GROUP_FAULT")) {
        char what[16] = {0};
        int k = -1;
        double ms = 0.0;
        if (std::sscanf(e, "%15[a-z]:%d:%lf", what, &k, &ms) >= 2) {
            f.kind = !std::strcmp(what, "launch") ? 1 : !std::strcmp(what, "exchange") ? 2 : !std::strcmp(what, "stall") ? 3 : 0;
            f.device = k;
            f.stall_ms = ms > 0.0 ? (ms < 5000.0 ? ms : 5000.0) : 200.0;
        }
    }
#endif
    return f;
}

}  // namespace

struct psa_group {
    std::vector<int>          devices;
    std::vector<psa_context*> ctx;
    int                       exchange = PSA_EXCHANGE_RCCL;
    std::vector<ncclComm_t>   comm;       // RCCL exchange only
    // per device: all ndev records after the exchange, and — INSIDE it, at the device's own rank (the all-gather's
    // in-place form: no local copy) — the 32-byte record the search's last kernel writes
    std::vector<psa_result*>  d_recv;
    std::vector<psa_result*>  d_send;     // = d_recv[i] + i
    psa_result*               h_send = nullptr;   // pinned [ndev]: records of empty shards (written by the host)
    // pinned, host-coherent, mapped on device 0: [0] the exchange's flag word (64 bytes), then the exchanged records [ndev],
    // written by psa_group_publish_kernel (RCCL) or by the launcher threads (HOST) and read by the caller
    unsigned char*            h_pub = nullptr;
    unsigned*                 h_flag = nullptr;   unsigned*   hd_flag = nullptr;
    psa_result*               h_recv = nullptr;   psa_result* hd_recv = nullptr;
    // batches: per device all ndev blocks of `per` records per mode after the exchange; a device's own block is written
    // INSIDE it, at the device's rank
    std::vector<psa_result*>  d_brecv;
    psa_result*               h_bsend = nullptr;  // pinned [ndev][2 * per]
    psa_result*               h_brecv = nullptr;  psa_result* hd_brecv = nullptr;   // pinned [ndev][2 * per], mapped on device 0
    int                       batch_per = 0;
    int is_max = 1;
    bool have_weights = false;
    int n = -1, m = -1;
    std::unique_ptr<GroupWorkers> workers;        // one launcher thread per device beyond the first
    // the last group call: microseconds from its entry until the last device's work had been enqueued, and until it returned
    double last_issue_us = 0.0, last_total_us = 0.0;
    // ---- no call hangs on a device that does not answer
    unsigned xseq = 0;                            // number of the last exchange (what its flag word will show)
    double   timeout_s = 60.0;                    // how long a call waits for the devices (psa_group_set_timeout)
    // how far every device's job of the running call has come: 0 nothing enqueued, 1 its search, 2 its part of the
    // exchange, -1 its all-gather could not be enqueued (written by its launcher thread, read by device 0's on a timeout)
    std::unique_ptr<std::atomic<int>[]> stage;
    // a call gave up waiting: work of it may still be on the devices' streams.  The next call looks (hipStreamQuery) and
    // goes ahead once every stream has drained; until then it fails at once instead of queueing behind a stalled device.
    std::atomic<bool> draining{false};
};

// Where every device of a timed-out call stands (device 0's thread; the stage words order what it reads of the others)
static std::string stalled_devices(const psa_group* g, bool single_search)
{
    std::string s;
    bool any = false;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        const int st = g->stage[i].load(std::memory_order_acquire);
        const psa_context* c = g->ctx[i];
        const char* what = nullptr;
        if (st == 0) what = "had not enqueued its work";
        else if (st == -1) what = "could not enqueue its all-gather";
        else if (single_search && c->pending && *static_cast<const volatile int*>(&c->h_tails[0].seq) != c->seq) what = "has not finished its search";
        if (what) {
            s += std::string(any ? "; " : "") + "device " + std::to_string(g->devices[i]) + " (shard " + std::to_string(i) + ") " + what;
            any = true;
        }
    }
    return any ? s : std::string("every device had joined: the all-gather itself has not completed");
}

// A group whose last call timed out takes the next one only when every stream has drained.
static psa_status group_ready(psa_group* g)
{
    if (!g->draining.load()) return PSA_OK;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        psa_context* c = g->ctx[i];
        PSA_TRY(ensure_device(c));
        // (a first hipStreamQuery of a stream nobody has waited on may say "not ready" for work that has ended — the
        // runtime only then asks for its completion: look for a millisecond before calling the device busy)
        if (!stream_idle_within(c->stream, 1e-3)) {
            set_last_error("device " + std::to_string(g->devices[i]) + " (shard " + std::to_string(i) +
                           ") is still busy with a group call that timed out");
            return PSA_ERR_DEVICE;
        }
    }
    for (psa_context* c : g->ctx) {
        c->pending = false;
        c->record_out = nullptr;
        c->record_shard = -1;
        for (auto& sl : c->slot) sl.scal_dirty = true;       // (whatever the abandoned search left in the search scalars)
    }
    g->draining.store(false);
    return PSA_OK;
}

// Device 0's end of an RCCL exchange: the publish launch behind the all-gather, then the bounded wait for its flag.
static psa_status publish_and_wait(psa_group* g, const psa_result* d_src, psa_result* hd_dst, size_t bytes, unsigned xseq,
                                   bool single_search, const char* who)
{
    psa_context* c = g->ctx[0];
    PSA_TRY(ensure_device(c));
    const int pieces = (int)(bytes / sizeof(uint4));
    hipLaunchKernelGGL(psa_group_publish_kernel, dim3(1), dim3(pieces <= 64 ? 64 : 256), 0, c->stream,
                       reinterpret_cast<const uint4*>(d_src), reinterpret_cast<uint4*>(hd_dst), pieces, g->hd_flag, xseq);
    PSA_HIP(hipGetLastError());
    if (!flag_arrived(g->h_flag, xseq, g->timeout_s)) {
        g->draining.store(true);
        set_last_error(std::string(who) + ": the exchange has not completed within " + std::to_string(g->timeout_s) + " s: " +
                       stalled_devices(g, single_search));
        return PSA_ERR_DEVICE;
    }
    return PSA_OK;
}

extern "C" {

psa_status psa_group_create_ex(const int* devices, int ndev, int exchange, psa_group** out)
{
    if (!out || !devices || ndev < 1 || (exchange != PSA_EXCHANGE_RCCL && exchange != PSA_EXCHANGE_HOST)) {
        set_last_error("psa_group_create: bad arguments");
        return PSA_ERR_ARG;
    }
    *out = nullptr;
    if (exchange == PSA_EXCHANGE_RCCL)          // (RCCL wants one rank per GPU; the host exchange takes several contexts on one)
        for (int i = 0; i < ndev; ++i)
            for (int k = 0; k < i; ++k)
                if (devices[i] == devices[k]) { set_last_error("psa_group_create: a device is listed twice"); return PSA_ERR_ARG; }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) { set_last_error("no HIP device available"); return PSA_ERR_DEVICE; }
    std::string err;
    if (exchange == PSA_EXCHANGE_RCCL && !load_rccl(&err)) { set_last_error(err); return PSA_ERR_DEVICE; }
    DeviceScope device_scope;
    psa_group* g = new psa_group();
    g->devices.assign(devices, devices + ndev);
    g->exchange = exchange;
    g->stage.reset(new std::atomic<int>[(size_t)ndev]);
    for (int i = 0; i < ndev; ++i) g->stage[(size_t)i].store(0);
#ifdef PSA_EXPERIMENT
    if (const char* t = std::getenv("PSA_GROUP_TIMEOUT_MS")) g->timeout_s = std::atof(t) / 1e3;
#endif
    psa_status st = [&]() -> psa_status {
        for (int i = 0; i < ndev; ++i) {
            psa_context* c = nullptr;
            PSA_TRY(psa_create(devices[i], nullptr, &c));
            g->ctx.push_back(c);
            // a GPU listed several times (host exchange: an N-way split rehearsed on fewer GPUs) runs several shards at
            // once: the chunk workgroups of a tile are not resident together there, and waiting for each other only holds
            // CUs the others need (8 contexts on one GPU: 260 us per call instead of 175) — the last arriver does the tile
            int listed = 0;
            for (int k = 0; k < ndev; ++k) listed += devices[k] == devices[i];
            if (listed > 1) c->share_combine = false;
        }
        if (exchange == PSA_EXCHANGE_RCCL) {
            g->comm.assign((size_t)ndev, nullptr);
            PSA_NCCL(g_rccl.comm_init_all(g->comm.data(), ndev, devices));      // one communicator over the listed GPUs (xGMI)
        }
        g->d_send.assign((size_t)ndev, nullptr);
        g->d_recv.assign((size_t)ndev, nullptr);
        for (int i = 0; i < ndev; ++i) {
            PSA_HIP(hipSetDevice(devices[i]));
            PSA_HIP(hipMalloc(reinterpret_cast<void**>(&g->d_recv[i]), sizeof(psa_result) * (size_t)ndev));
            PSA_HIP(hipMemset(g->d_recv[i], 0, sizeof(psa_result) * (size_t)ndev));
            g->d_send[i] = g->d_recv[i] + i;
        }
        PSA_HIP(hipSetDevice(devices[0]));
        PSA_HIP(hipHostMalloc(reinterpret_cast<void**>(&g->h_send), sizeof(psa_result) * (size_t)ndev, hipHostMallocPortable));
        PSA_HIP(hipHostMalloc(reinterpret_cast<void**>(&g->h_pub), 64 + sizeof(psa_result) * (size_t)ndev,
                              hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(g->h_pub, 0, 64 + sizeof(psa_result) * (size_t)ndev);
        unsigned char* hd = nullptr;
        PSA_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), g->h_pub, 0));
        g->h_flag = reinterpret_cast<unsigned*>(g->h_pub);
        g->hd_flag = reinterpret_cast<unsigned*>(hd);
        g->h_recv = reinterpret_cast<psa_result*>(g->h_pub + 64);
        g->hd_recv = reinterpret_cast<psa_result*>(hd + 64);
        g->workers.reset(new GroupWorkers(g->devices, [](int device) { (void)hipSetDevice(device); }));
        return PSA_OK;
    }();
    if (st != PSA_OK) { psa_group_destroy(g); return st; }
    *out = g;
    return PSA_OK;
}

psa_status psa_group_create(const int* devices, int ndev, psa_group** out)
{
    return psa_group_create_ex(devices, ndev, PSA_EXCHANGE_RCCL, out);
}

void psa_group_destroy(psa_group* g)
{
    if (!g) return;
    DeviceScope device_scope;
    g->workers.reset();                          // the launcher threads end before anything they use is freed
    // a group that gave up on a device: its streams get a last, bounded chance to drain; what is still busy then is
    // aborted (RCCL) and its memory is left to the process's end rather than freed under a running kernel
    bool busy = false;
    if (g->draining.load())
        for (psa_context* c : g->ctx)
            if (hipSetDevice(c->device) != hipSuccess || !stream_idle_within(c->stream, 1.0)) busy = true;
    for (size_t i = 0; i < g->comm.size(); ++i)
        if (g->comm[i]) (void)(busy && g_rccl.comm_abort ? g_rccl.comm_abort(g->comm[i]) : g_rccl.comm_destroy(g->comm[i]));
    if (busy) {
        // (contexts, buffers and pinned memory of a stalled device stay allocated: freeing them would wait for the device)
        delete g;
        return;
    }
    for (size_t i = 0; i < g->d_recv.size(); ++i) {
        (void)hipSetDevice(g->devices[i]);
        (void)hipFree(g->d_recv[i]);
    }
    for (size_t i = 0; i < g->d_brecv.size(); ++i) {
        (void)hipSetDevice(g->devices[i]);
        (void)hipFree(g->d_brecv[i]);
    }
    if (g->h_send) (void)hipHostFree(g->h_send);
    if (g->h_pub) (void)hipHostFree(g->h_pub);
    if (g->h_bsend) (void)hipHostFree(g->h_bsend);
    if (g->h_brecv) (void)hipHostFree(g->h_brecv);
    for (psa_context* c : g->ctx) psa_destroy(c);
    delete g;
}

int psa_group_size(const psa_group* g) { return g ? (int)g->ctx.size() : 0; }

// How long a group call waits for the devices before it gives up with PSA_ERR_DEVICE (default 60 s): device 0's wait for
// the exchanged records, and (HOST exchange) every launcher thread's wait for its own search.
psa_status psa_group_set_timeout(psa_group* g, double seconds)
{
    if (!g || !(seconds > 0.0)) { set_last_error("psa_group_set_timeout: bad arguments"); return PSA_ERR_ARG; }
    g->timeout_s = seconds;
    return PSA_OK;
}

psa_status psa_group_set_weights(psa_group* g, const double weights[4], int is_max)
{
    if (!g) { set_last_error("group is NULL"); return PSA_ERR_ARG; }
    DeviceScope device_scope;
    PSA_TRY(group_ready(g));
    for (psa_context* c : g->ctx) PSA_TRY(psa_set_weights(c, weights, is_max));
    g->is_max = is_max ? 1 : 0;
    g->have_weights = true;
    return PSA_OK;
}

// Every device receives the whole sequences (the reference broadcasts ProgramData to every rank,
// cpu_funcs.c:49-52; a shard only reads its offsets' window).  Copies and encodes are enqueued on all
// devices first and synchronised afterwards, so the uploads of the devices overlap.
static psa_status group_set_seq(psa_group* g, const char* seq, int len, bool first_seq)
{
    if (!g || !seq || len < 1) { set_last_error("psa_group_set_seq: bad arguments"); return PSA_ERR_ARG; }
    if (first_seq) PSA_TRY(check_seq1_length(len));
    DeviceScope device_scope;
    PSA_TRY(group_ready(g));
    for (psa_context* c : g->ctx) {
        PSA_TRY(ensure_device(c));
        PSA_TRY(stage_host(c, seq, (size_t)len));
        if (first_seq) PSA_TRY(encode_seq1(c, c->d_raw, len));
        else PSA_TRY(encode_seq2(c, c->slot[0], c->d_raw, len, c->stream));
    }
    for (psa_context* c : g->ctx) {
        PSA_TRY(ensure_device(c));
        PSA_TRY(check_bad_now(c));
        PSA_TRY(report_bad(c));
    }
    (first_seq ? g->n : g->m) = len;
    return PSA_OK;
}

psa_status psa_group_set_seq1(psa_group* g, const char* seq1, int n) { return group_set_seq(g, seq1, n, true); }
psa_status psa_group_set_seq2(psa_group* g, const char* seq2, int m) { return group_set_seq(g, seq2, m, false); }

// Offsets [first,last) split over the group's devices with the reference's rank rule (psa_shard_range,
// cpu_funcs.c:128-133), all shards searched concurrently, ONE exchange of the 32-byte records (RCCL all-gather in place
// of MPI_Allreduce(MAXLOC|MINLOC) + MPI_Send/Recv of the Mutant, cpu_funcs.c:71-94), then the is_swapable fold (ties:
// lowest offset = lowest shard, like MPI_MAXLOC's lowest rank).
// Every device's share of the call runs on that device's launcher thread (GroupWorkers), all at once:
//   RCCL  the search launch (its last kernel writes the record straight into the device's slot of its receive buffer:
//         the all-gather's in-place form), the device's ncclAllGather on the same stream — every thread on its own
//         communicator, the multi-thread form of a one-process communicator set — and, on device 0, a one-workgroup
//         kernel that publishes the gathered records in pinned host memory, its flag word last; the caller polls that
//         word.  A record tagged "look again" (candidate-list overflow, too many survivors) or not carrying this search's
//         tag sends that device through the single-context retry path afterwards.
//   HOST  the search launch and the wait for that device's result tail (pinned host memory, written by the search's
//         last kernel and polled: psa_search_wait, retry paths included); the records meet in host memory.
// NO CALL HANGS (the reference: MPI_Abort, cpu_funcs.c:38-44).  (1) Whatever fails on a device's thread — selecting the
// device, the launch, the copy of an empty shard's record — is recorded, and the thread still reaches its all-gather (with
// whatever its send slot holds: the tag check rejects it): no other device's collective is left waiting for a rank that
// returned early.  (2) Every wait has a deadline (psa_group_set_timeout, 60 s): device 0's wait for the flag word, a HOST
// thread's wait for its search.  A call that gives up returns PSA_ERR_DEVICE naming the devices that had not answered
// and marks the group "draining": the next call looks at the streams and fails at once while one of them is still busy
// (it never queues behind a stalled device), and goes ahead as soon as they have drained.
psa_status psa_search_sharded(psa_group* g, int first, int last, psa_result* out, psa_result* shard_records)
{
    if (!g || !out) { set_last_error("psa_search_sharded: NULL argument"); return PSA_ERR_ARG; }
    if (!g->have_weights || g->n < 0 || g->m < 0) {
        set_last_error("weights, seq1 and seq2 must be set before a search");
        return PSA_ERR_STATE;
    }
    const auto t_enter = std::chrono::steady_clock::now();
    const int ndev = (int)g->ctx.size();
    PSA_TRY(check_ready(g->ctx[0], g->m, first, last));
    DeviceScope device_scope;
    PSA_TRY(group_ready(g));
    const int total = last - first;
    const bool rccl = g->exchange == PSA_EXCHANGE_RCCL;
    const GroupFault fault = group_fault();
    const unsigned xseq = ++g->xseq;
    std::vector<int> lo((size_t)ndev), hi((size_t)ndev), tag((size_t)ndev);
    std::vector<psa_status> status((size_t)ndev, PSA_OK);
    std::vector<std::string> errors((size_t)ndev);
    std::vector<std::chrono::steady_clock::time_point> issued((size_t)ndev, t_enter);
    // whatever happens below, no context is left with a search "in flight" (an error half-way would otherwise block
    // the next call on the devices that had been started)
    struct Settle {
        psa_group* g;
        ~Settle()
        {
            const bool draining = g->draining.load();         // (a device that does not answer is not waited for here either)
            for (psa_context* c : g->ctx) {
                if (c->pending) {
                    if (!draining) {
                        (void)hipSetDevice(c->device);
                        (void)hipStreamSynchronize(c->stream);
                    }
                    c->slot[0].scal_dirty = true;
                }
                c->pending = false;
                c->record_out = nullptr;
                c->record_shard = -1;
            }
        }
    } settle_on_exit{g};
    for (int i = 0; i < ndev; ++i) {
        psa_shard_range(total, ndev, i, &lo[i], &hi[i]);
        lo[i] += first;
        hi[i] += first;
        // what the record must carry in `shard` to count as this search's: the shard and the search's number
        tag[i] = i | (((g->ctx[i]->seq + 1) & 0x3fffff) << 8);
        g->stage[(size_t)i].store(0);
    }
    auto device_job = [&](int i) {
        psa_context* c = g->ctx[(size_t)i];
        DeviceScope scope;                       // (the caller's thread serves device 0: its current device is put back)
        psa_status st = PSA_OK;
        std::string err;
        auto note = [&](psa_status s) {          // the first failure is the one reported; the job goes on to its exchange
            if (s != PSA_OK && st == PSA_OK) { st = s; err = psa_last_error(); }
        };
        // ---- this device's search
        if (fault.hits(1, i)) {
            set_last_error("injected fault: device fails before its launch");
            note(PSA_ERR_DEVICE);
        } else if (hi[i] > lo[i]) {
#ifdef PSA_EXPERIMENT
            if (fault.hits(3, i) && ensure_device(c) == PSA_OK)
                hipLaunchKernelGGL(psa_debug_stall_kernel, dim3(1), dim3(1), 0, c->stream, (unsigned long long)(fault.stall_ms * 1e5));
#endif
            note(search_async_impl(c, lo[i], hi[i], rccl ? g->d_send[i] : nullptr, tag[i]));
        } else {
            psa_result r;                                    // an empty shard: "nothing found" (cuda_funcs.cu:158-161)
            r.score = g->is_max ? -INFINITY : INFINITY;
            r.offset = -1; r.char_offset = -1; r.ch = 0; r.pairs = 0;
            r.shard = tag[i];
            c->seq += 1;
            if (rccl) {
                g->h_send[i] = r;
                note(ensure_device(c));
                if (st == PSA_OK &&
                    hipMemcpyAsync(g->d_send[i], &g->h_send[i], sizeof(psa_result), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                    set_last_error("psa_search_sharded: copy of an empty shard's record failed");
                    note(PSA_ERR_DEVICE);
                }
            } else {
                g->h_recv[i] = r;
            }
        }
        issued[(size_t)i] = std::chrono::steady_clock::now();
        g->stage[(size_t)i].store(1, std::memory_order_release);
        if (fault.hits(2, i)) {
            set_last_error("injected fault: device fails before its part of the exchange");
            note(PSA_ERR_DEVICE);
        }
        // ---- its part of the exchange: every path above arrives here
        if (rccl) {
            const ncclResult_t r = g_rccl.all_gather(g->d_send[i], g->d_recv[i], sizeof(psa_result), ncclChar, g->comm[i], c->stream);
            if (r != ncclSuccess) {
                set_last_error(std::string("ncclAllGather: ") + g_rccl.error_string(r));
                note(PSA_ERR_DEVICE);
                g->stage[(size_t)i].store(-1, std::memory_order_release);
            } else {
                g->stage[(size_t)i].store(2, std::memory_order_release);
            }
            if (i == 0) note(publish_and_wait(g, g->d_recv[0], g->hd_recv, sizeof(psa_result) * (size_t)ndev, xseq, true, "psa_search_sharded"));
        } else if (st == PSA_OK && hi[i] > lo[i]) {
            psa_result r;
            const psa_status w = search_wait_impl(c, &r, g->timeout_s);     // this device's tail, retry paths included
            note(w);
            if (w == PSA_OK) {
                r.shard = tag[i];
                g->h_recv[i] = r;
            } else if (c->pending) {
                g->draining.store(true);                                     // (gave up: the search is still on the stream)
            }
            g->stage[(size_t)i].store(2, std::memory_order_release);
        }
        status[(size_t)i] = st;
        errors[(size_t)i] = err;
    };
    g->workers->run(device_job);
    {
        auto last_issue = t_enter;
        for (const auto& t : issued) last_issue = std::max(last_issue, t);
        g->last_issue_us = std::chrono::duration<double, std::micro>(last_issue - t_enter).count();
    }
    psa_status st = PSA_OK;
    for (int i = 0; i < ndev && st == PSA_OK; ++i)
        if (status[(size_t)i] != PSA_OK) { set_last_error(errors[(size_t)i]); st = status[(size_t)i]; }
    for (int i = 0; i < ndev && !g->draining.load(); ++i) {
        psa_context* c = g->ctx[i];
        if (c->pending && st == PSA_OK && g->h_recv[i].shard != tag[i]) {
            // this device's search asks for the host (or its record is not this search's): the single-context path
            st = ensure_device(c);
            if (st == PSA_OK && hipStreamSynchronize(c->stream) != hipSuccess) {
                set_last_error("psa_search_sharded: stream synchronisation failed");
                st = PSA_ERR_DEVICE;
            }
            if (st == PSA_OK) st = settle_search(c);
            if (st == PSA_OK) g->h_recv[i] = c->h_tails[0].result;
        } else if (c->pending && c->pending_filter && st == PSA_OK) {
            c->slot[0].scal_dirty = false;               // its last workgroup zeroed the search scalars
            if (c->h_tails[0].seq == c->seq) note_share_outcome(c);      // (the device's own tail has landed with its record)
        }
        if (st == PSA_OK) {
            c->pending = false;
            c->record_out = nullptr;
            c->record_shard = -1;
            g->h_recv[i].shard = i;
        }
    }
    g->last_total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count();
    PSA_TRY(st);
    if (shard_records) std::memcpy(shard_records, g->h_recv, sizeof(psa_result) * (size_t)ndev);
    return psa_fold(g->h_recv, ndev, g->is_max, out);
}

// Host time of the last psa_search_sharded / psa_search_batch_sharded of this group, microseconds from the call's entry:
// until the last device's work had been enqueued (the launcher threads work side by side), and until the call returned.
psa_status psa_group_last_timing(const psa_group* g, double* issue_us, double* total_us)
{
    if (!g) { set_last_error("group is NULL"); return PSA_ERR_ARG; }
    if (issue_us) *issue_us = g->last_issue_us;
    if (total_us) *total_us = g->last_total_us;
    return PSA_OK;
}


// A batch of queries over the group (BASELINE.json configs[4] on several GPUs): the queries are dealt to the devices
// in contiguous blocks (psa_query_block: as even as possible); every device's launcher thread runs its block through
// psa_search_batch / _minmax (the records also land in the device's block of its receive buffer — the all-gather's
// in-place form — written by the kernels that finish each query) and enqueues its all-gather on the same stream;
// device 0's thread publishes the gathered blocks in pinned host memory (psa_group_publish_kernel) and polls the flag —
// one wait, with the group's deadline.  psa_batch_unpack puts them in query order.  A block with a query the host had to
// finish (re-selection, sequential-engine fallback) is copied to the device again before the exchange.  As in
// psa_search_sharded a device's thread reaches its all-gather whatever failed before.
// results_min == NULL: one mode (the group's).  HOST exchange: the blocks meet in pinned host memory.
psa_status psa_search_batch_sharded(psa_group* g, const char* queries, int nq, int m, size_t stride,
                                    int first, int last, psa_result* results, psa_result* results_min)
{
    if (!g || !queries || !results || nq < 1 || stride < (size_t)m) {
        set_last_error("psa_search_batch_sharded: bad arguments");
        return PSA_ERR_ARG;
    }
    if (!g->have_weights || g->n < 0) { set_last_error("weights and seq1 must be set before a search"); return PSA_ERR_STATE; }
    const auto t_enter = std::chrono::steady_clock::now();
    DeviceScope device_scope;
    PSA_TRY(group_ready(g));
    const int ndev = (int)g->ctx.size();
    const int nmodes = results_min ? 2 : 1;
    const bool rccl = g->exchange == PSA_EXCHANGE_RCCL;
    const int per = psa_query_block_max(nq, ndev);          // records per device and mode in the exchange (the largest block)
    if (per > g->batch_per) {
        g->batch_per = 0;
        for (size_t i = 0; i < g->d_brecv.size(); ++i) {
            PSA_HIP(hipSetDevice(g->devices[i]));
            (void)hipFree(g->d_brecv[i]);
            g->d_brecv[i] = nullptr;                         // (an error return below leaves nothing to free twice)
        }
        if (g->h_bsend) (void)hipHostFree(g->h_bsend);
        if (g->h_brecv) (void)hipHostFree(g->h_brecv);
        g->h_bsend = g->h_brecv = g->hd_brecv = nullptr;
        g->d_brecv.assign((size_t)ndev, nullptr);
        for (int i = 0; i < ndev; ++i) {
            PSA_HIP(hipSetDevice(g->devices[i]));
            PSA_HIP(hipMalloc(reinterpret_cast<void**>(&g->d_brecv[i]), sizeof(psa_result) * 2 * (size_t)per * ndev));
            PSA_HIP(hipMemset(g->d_brecv[i], 0, sizeof(psa_result) * 2 * (size_t)per * ndev));     // (padding entries are never unpacked)
        }
        PSA_HIP(hipSetDevice(g->devices[0]));
        PSA_HIP(hipHostMalloc(reinterpret_cast<void**>(&g->h_bsend), sizeof(psa_result) * 2 * (size_t)per * ndev, hipHostMallocPortable));
        PSA_HIP(hipHostMalloc(reinterpret_cast<void**>(&g->h_brecv), sizeof(psa_result) * 2 * (size_t)per * ndev,
                              hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
        PSA_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&g->hd_brecv), g->h_brecv, 0));
        g->batch_per = per;
    }
    // the devices' blocks: device i searches queries [lo_i, hi_i) and fills block i of the exchange, [mode][per]
    std::vector<int> lo((size_t)ndev), hi((size_t)ndev);
    std::vector<psa_status> status((size_t)ndev, PSA_OK);
    std::vector<std::string> errors((size_t)ndev);
    std::vector<std::chrono::steady_clock::time_point> issued((size_t)ndev, t_enter);
    const size_t block = sizeof(psa_result) * 2 * (size_t)per;             // (a batch smaller than the buffers uses their front)
    // device i's send block INSIDE its receive buffer, at its own rank: the all-gather's in-place form (no local copy)
    std::vector<psa_result*> d_bsend((size_t)ndev);
    for (int i = 0; i < ndev; ++i) d_bsend[(size_t)i] = g->d_brecv[(size_t)i] + (size_t)i * 2 * per;
    psa_result none;
    none.score = 0.0; none.offset = -1; none.char_offset = -1; none.ch = 0; none.shard = -1; none.pairs = 0;
    for (int i = 0; i < ndev; ++i) {
        psa_query_block(nq, ndev, i, &lo[i], &hi[i]);
        if (hi[i] - lo[i] > per) { set_last_error("internal: a query block exceeds the exchange block"); return PSA_ERR_STATE; }
        g->stage[(size_t)i].store(0);
    }
    const GroupFault fault = group_fault();
    const unsigned xseq = ++g->xseq;
    auto device_job = [&](int i) {
        psa_context* c = g->ctx[(size_t)i];
        DeviceScope scope;
        psa_status st = PSA_OK;
        std::string err;
        auto note = [&](psa_status s) {
            if (s != PSA_OK && st == PSA_OK) { st = s; err = psa_last_error(); }
        };
        psa_result* mine = g->h_bsend + (size_t)i * 2 * per;
        for (int k = 0; k < 2 * per; ++k) mine[k] = none;
        const int count = hi[i] - lo[i];
        bool on_device = false;                // the kernels have written this block's records to d_bsend[i]
        if (fault.hits(1, i)) {
            set_last_error("injected fault: device fails before its launch");
            note(PSA_ERR_DEVICE);
        } else if (count > 0) {
            const char* q0 = queries + (size_t)lo[i] * stride;
            // (the group's mode first: results / results_min are MAX / MIN)
            c->batch_records[0] = rccl ? d_bsend[(size_t)i] : nullptr;
            c->batch_records[1] = rccl && nmodes == 2 ? d_bsend[(size_t)i] + per : nullptr;
            c->batch_records_final = false;
#ifdef PSA_EXPERIMENT
            if (fault.hits(3, i) && ensure_device(c) == PSA_OK)
                hipLaunchKernelGGL(psa_debug_stall_kernel, dim3(1), dim3(1), 0, c->stream, (unsigned long long)(fault.stall_ms * 1e5));
#endif
            if (nmodes == 2) note(psa_search_batch_minmax(c, q0, count, m, stride, first, last, mine, mine + per));
            else note(psa_search_batch(c, q0, count, m, stride, first, last, mine));
            on_device = rccl && st == PSA_OK && c->batch_records_final;
            c->batch_records[0] = c->batch_records[1] = nullptr;
        }
        issued[(size_t)i] = std::chrono::steady_clock::now();
        g->stage[(size_t)i].store(1, std::memory_order_release);
        if (fault.hits(2, i)) {
            set_last_error("injected fault: device fails before its part of the exchange");
            note(PSA_ERR_DEVICE);
        }
        if (rccl) {
            // ---- its part of the exchange: every path above arrives here
            if (!on_device) {
                if (ensure_device(c) != PSA_OK ||
                    hipMemcpyAsync(d_bsend[(size_t)i], mine, block, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
                    set_last_error("psa_search_batch_sharded: copy of a device's block failed");
                    note(PSA_ERR_DEVICE);
                }
            }
            const ncclResult_t r = g_rccl.all_gather(d_bsend[(size_t)i], g->d_brecv[i], block, ncclChar, g->comm[i], c->stream);
            if (r != ncclSuccess) {
                set_last_error(std::string("ncclAllGather: ") + g_rccl.error_string(r));
                note(PSA_ERR_DEVICE);
                g->stage[(size_t)i].store(-1, std::memory_order_release);
            } else {
                g->stage[(size_t)i].store(2, std::memory_order_release);
            }
            if (i == 0) note(publish_and_wait(g, g->d_brecv[0], g->hd_brecv, block * (size_t)ndev, xseq, false, "psa_search_batch_sharded"));
        }
        status[(size_t)i] = st;
        errors[(size_t)i] = err;
    };
    g->workers->run(device_job);
    {
        auto last_issue = t_enter;
        for (const auto& t : issued) last_issue = std::max(last_issue, t);
        g->last_issue_us = std::chrono::duration<double, std::micro>(last_issue - t_enter).count();
        g->last_total_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_enter).count();
    }
    for (int i = 0; i < ndev; ++i)
        if (status[(size_t)i] != PSA_OK) { set_last_error(errors[(size_t)i]); return status[(size_t)i]; }
    // host exchange: the blocks already lie side by side in pinned host memory
    return psa_batch_unpack(rccl ? g->h_brecv : g->h_bsend, nq, ndev, results, results_min);
}
}  // extern "C"

extern "C" {

// ---------------------------------------------------------------------------
// legacy drop-in (cuda_funcs.h:33)
// ---------------------------------------------------------------------------
static std::mutex g_legacy_mutex;
static psa_context* g_legacy_ctx = nullptr;
static psa_group*   g_legacy_group = nullptr;

// Host-to-device transfers of this context (NULL: of the context behind gpu_run_program) that were streamed — the
// encode launch following the host's copy chunk by chunk — and how many of those had to be redone the plain way.
int psa_streamed_count(psa_context* c) { if (!c) c = g_legacy_ctx; return c ? c->streamed : 0; }
int psa_stream_retry_count(psa_context* c) { if (!c) c = g_legacy_ctx; return c ? c->stream_retries : 0; }

static void legacy_die(const char* what)
{
    // cuda_funcs.cu:44-48 pattern: message on stderr, exit(EXIT_FAILURE)
    std::fprintf(stderr, "%s - %s\n", what, psa_last_error());
    std::exit(EXIT_FAILURE);
}

double gpu_run_program_layout(const void* cpu_data, size_t seq1_cap, size_t seq2_cap,
                              Mutant* returned_mutant, int first_offset, int last_offset)
{
    std::lock_guard<std::mutex> lock(g_legacy_mutex);
    init_trace("gpu_run_program: enter");
    const char* base = static_cast<const char*>(cpu_data);
    int is_max;
    double weights[4];
    std::memcpy(&is_max, base, sizeof is_max);
    std::memcpy(weights, base + 8, sizeof weights);
    const char* seq1 = base + 40;
    const char* seq2 = base + 40 + seq1_cap;
    // A long Seq1 (no NUL in its first 256 KB) is not measured up front: its end is found while it is copied to the
    // device (stream_pair), the encode launch already under way; n = -1 until then.
    if (seq1_cap > (size_t)PSA_MAX_SEQ1 + 1) {
        std::fprintf(stderr, "gpu_run_program: seq1[%zu] is longer than this library takes (PSA_MAX_SEQ1 = 2^30)\n", seq1_cap);
        std::exit(EXIT_FAILURE);
    }
    const size_t peek = std::min(seq1_cap, stream_min());
    const void* nul = std::memchr(seq1, 0, peek);
    bool stream = !nul && seq1_cap > peek;
    int n = nul ? (int)(static_cast<const char*>(nul) - seq1) : (stream ? -1 : (int)seq1_cap);
    const int m = (int)strnlen(seq2, seq2_cap);
    auto no_nul = [&]() {
        std::fprintf(stderr, "gpu_run_program: no NUL inside seq1[%zu] / seq2[%zu] - the caller's ProgramData was compiled with "
                             "other array limits than this entry point reads (the plain symbol reads the reference's stock "
                             "10001 / 5001 layout; other limits: gpu_run_program_layout, or include psa.h)\n", seq1_cap, seq2_cap);
        std::exit(EXIT_FAILURE);
    };

    returned_mutant->offset = -1;          // cuda_funcs.cu:158-160
    returned_mutant->char_offset = -1;
    returned_mutant->ch = PSA_NO_CHAR;
    const double none = is_max ? -INFINITY : INFINITY;
    // A plain symbol cannot see how its caller compiled ProgramData.  What a mismatch looks like from here: no NUL
    // inside an array, or an empty Seq2 (the caller's seq2 lies elsewhere).  Say so instead of answering from the
    // wrong bytes (the reference prints and exits on every error, cuda_funcs.cu:44-48).
    if ((!stream && (size_t)n >= seq1_cap) || (size_t)m >= seq2_cap) no_nul();
    if (m < 1 && (stream || n >= 1) && last_offset > first_offset) {
        std::fprintf(stderr, "gpu_run_program: seq2 is empty at offset %zu of ProgramData - if the caller was compiled with array "
                             "limits other than %zu / %zu its seq2 lies elsewhere (use gpu_run_program_layout, or include psa.h); "
                             "nothing searched\n", (size_t)40 + seq1_cap, seq1_cap, seq2_cap);
    }
    if (m < 1 || last_offset <= first_offset) return none;
    DeviceScope device_scope;

    // PSA_DEVICES=0,1,...: the offsets are sharded over those GPUs (psa_search_sharded: the reference's MPI
    // ranks, cpu_funcs.c:49-94, as one process and one RCCL exchange); otherwise one GPU, PSA_DEVICE (default 0)
    std::vector<int> devices;
    if (const char* list = std::getenv("PSA_DEVICES")) {
        for (const char* p = list; *p;) {
            char* end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            devices.push_back((int)v);
            p = *end == ',' ? end + 1 : end;
            if (*end != ',' && *end != '\0') break;
        }
    }
    psa_result r;
    if (devices.size() > 1) {
        if (stream) {                        // (the group's setters want the length)
            stream = false;
            n = (int)strnlen(seq1, seq1_cap);
            if ((size_t)n >= seq1_cap) no_nul();
        }
        if (n < 1) legacy_die("Failed to copy data from host to device (seq1)");
        // PSA_EXCHANGE=host: the records meet in host memory instead of an RCCL all-gather (a device may then be listed
        // more than once: an N-way split rehearsed on fewer GPUs)
        const char* ex = std::getenv("PSA_EXCHANGE");
        const int exchange = ex && std::strcmp(ex, "host") == 0 ? PSA_EXCHANGE_HOST : PSA_EXCHANGE_RCCL;
        if (!g_legacy_group && psa_group_create_ex(devices.data(), (int)devices.size(), exchange, &g_legacy_group) != PSA_OK)
            legacy_die("Failed to initialise the device group (PSA_DEVICES)");
        psa_group* g = g_legacy_group;
        if (psa_group_set_weights(g, weights, is_max) != PSA_OK) legacy_die("Failed to build the score tables");
        if (psa_group_set_seq1(g, seq1, n) != PSA_OK) legacy_die("Failed to copy data from host to device (seq1)");
        if (psa_group_set_seq2(g, seq2, m) != PSA_OK) legacy_die("Failed to copy data from host to device (seq2)");
        if (psa_search_sharded(g, first_offset, last_offset, &r, nullptr) != PSA_OK) legacy_die("Failed to run the search");
    } else {
        if (!g_legacy_ctx) {
            int device = devices.size() == 1 ? devices[0] : 0;
            if (devices.empty())
                if (const char* d = std::getenv("PSA_DEVICE")) device = std::atoi(d);
            if (psa_create(device, nullptr, &g_legacy_ctx) != PSA_OK) legacy_die("Failed to initialise the device context");
        }
        psa_context* c = g_legacy_ctx;
        init_trace("gpu_run_program: context ready");
        if (psa_set_weights(c, weights, is_max) != PSA_OK) legacy_die("Failed to build the score tables");
        init_trace("gpu_run_program: weights (tables built, 5 copies, synchronise)");
        // The two arrays of a ProgramData lie side by side: one host memcpy into pinned staging (the reference copies the
        // whole struct, cuda_funcs.cu:50), an encode launch that reads it over PCIe, one search launch; the alphabet flag
        // comes back with the result tail.
        bool searched = false;
        if (stream) {
            // long sequences: the encode launch first, the host's copy (and its search for Seq1's end) under the PCIe
            // transfer, the search launch behind it
            c->stream_gave_up = false;
            if (ensure_device(c) != PSA_OK || stream_pair(c, seq1, seq1_cap, true, seq2, (size_t)m, &n) != PSA_OK)
                legacy_die("Failed to copy data from host to device");
            if ((size_t)n >= seq1_cap) no_nul();
            init_trace("gpu_run_program: streamed transfer enqueued (first kernel launch: code object load)");
            searched = psa_search(c, first_offset, last_offset, &r) == PSA_OK;
            init_trace("gpu_run_program: search returned");
            if (!searched && !c->stream_gave_up) legacy_die("Failed to run the search");
            c->stream_gave_up = false;       // (gave up: the host stalled mid-copy; once more the plain way)
        }
        if (!searched) {
            if (n < 1 || ensure_device(c) != PSA_OK || stage_pinned_pair(c, seq1, (size_t)n, seq2, (size_t)m, true) != PSA_OK ||
                encode_pair(c, c->hd_stage, n, c->hd_stage + c->stage_second, m) != PSA_OK)
                legacy_die("Failed to copy data from host to device");
            if (psa_search(c, first_offset, last_offset, &r) != PSA_OK) legacy_die("Failed to run the search");
        }
    }
    init_trace("gpu_run_program: done");
    if (r.offset < 0 || r.ch == 0) return none;   // cuda_funcs.cu:143-144
    returned_mutant->offset = r.offset;
    returned_mutant->char_offset = r.char_offset;
    returned_mutant->ch = (char)r.ch;
    return r.score;
}

// The exported legacy symbol reads the reference's STOCK ProgramData (def.h:35-36: 10001 / 5001, 15,048 bytes):
// that is what an unmodified reference build passes.  Callers built with other limits use
// gpu_run_program_layout (psa.h does it for them).
double gpu_run_program(const ProgramData* cpu_data, Mutant* returned_mutant, int first_offset, int last_offset)
{
    return gpu_run_program_layout(cpu_data, PSA_REF_SEQ1_MAX_LEN, PSA_REF_SEQ2_MAX_LEN, returned_mutant,
                                  first_offset, last_offset);
}

// The reference compiles every file as C++ (Makefile:9-12) and declares no extern "C": its cpu_funcs.o refers to
// gpu_run_program(ProgramData*, Mutant*, int, int) by its Itanium-mangled name (cuda_funcs.h:33, call site
// cpu_funcs.c:180; the struct tags are _data and _mutant, program_data.h:6, mutant.h:6).  The same entry under
// that name: an unmodified reference object links without an edit.
PSA_EXPORT double psa_gpu_run_program_cxx(ProgramData* cpu_data, Mutant* returned_mutant, int first_offset, int last_offset)
    __asm__("_Z15gpu_run_programP5_dataP7_mutantii");
double psa_gpu_run_program_cxx(ProgramData* cpu_data, Mutant* returned_mutant, int first_offset, int last_offset)
{
    return gpu_run_program_layout(cpu_data, PSA_REF_SEQ1_MAX_LEN, PSA_REF_SEQ2_MAX_LEN, returned_mutant,
                                  first_offset, last_offset);
}

}  // extern "C"
