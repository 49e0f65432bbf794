// psa_filter_kernels.hpp — the "filter" engine (included by psa_device.hip).
//
// Idea.  A pair's score depends only on its sign class, so the
// order-independent ("mathematical") score of an offset is fixed by three
// integers: how many of its m pairs are '*', ':' and '.' (cpu_funcs.c:277,
// cuda_funcs.cu:442-452).  Those counts are computed EXACTLY, for every
// offset, with bit-sliced arithmetic.  The class of (Seq1[j], b) is stored as
// two bit-planes over the Seq1 positions j, for each of the 27 symbols b:
//
//   lo[b][j] = class is '*' or '.'        hi[b][j] = class is ':' or '.'   (built in LDS per workgroup)
//   n_lo(o)   = sum_i lo[Seq2[i]][o+i]    (= n_star + n_dot)
//   n_hi(o)   = sum_i hi[Seq2[i]][o+i]    (= n_colon + n_dot)
//   n_both(o) = sum_i (lo & hi)[...]      (= n_dot)
//
// One lane owns 64 positions (two 32-bit words) and adds plane words into per-lane
// bit-plane counters with carry-save adders built from v_bitop3_b32 (xor3 /
// majority: 2 VALU ops per 32 pairs and counter).  Char i of Seq2 pairs offset o
// with Seq1 position o+i, i.e. the planes of Seq2[i] shifted by i bits.  The chars
// are therefore grouped into 32 SHIFT CLASSES J = i mod 32: inside a class
// (i = 32k + J) char k reads the plane word one whole slot further on, with no
// bit shift at all — two aligned ds_read_b64 per char and no funnel shift.  With
//     D_J[x] = sum_k plane[Seq2[32k+J]][x + 32k]      (x = position in the tile)
// the count of offset o is sum_J D_J[o + J].  A wave walks its classes from the
// highest J down and shifts its COUNTER planes by one bit between two classes
// (Horner: C_J[x] = D_J[x] + C_{J+1}[x+1]; the neighbour word comes from the next
// lane by DPP), then by its lowest J once at the end: ~170 instructions per class
// instead of 4 half-rate v_alignbit_b32 per char.  The bits shifted in at the top
// of the tile's last word are not the tile's: a tile of 128 words answers for its
// first 127 words (kTileStride offsets).  From the counts, a rounding bound of
// the reference's sequential fp64 sum (cpu_funcs.c:278) gives a score interval
// per offset; only offsets whose interval can reach the best one are re-scored
// in the reference's exact order (psa_rescore_kernel).  The answer is the
// sequential engine's, bit for bit, for any input; inputs where too many
// offsets tie (homopolymers) fall back to the sequential engine.
#pragma once

#include <utility>

namespace psa {

constexpr int kTile = 4096;        // positions per workgroup: 64 lanes x 2 words x 32 bits
constexpr int kTileStride = kTile - 32;   // offsets a tile answers for (its last word takes the class shifts' spill)
constexpr int kClasses = 32;       // shift classes: chars i with the same i mod 32
constexpr int kUnroll = 32;        // chars per carry-save round (2^kLevels)
constexpr int kLevels = 5;
constexpr int kPlanes = 10;        // per-wave counters hold up to 1023 chars
constexpr int kMaxWaveChars = 992; // multiple of kUnroll below 2^kPlanes
constexpr int kMaskRows = 28;      // 27 symbols + the all-zero row of the padding symbol
constexpr int kCounters = 3;       // n_lo, n_hi, n_both
constexpr int kRowShift = 12;      // LDS rows of the count kernel are 4096 bytes apart (512 slots)
constexpr int kRowSlots = 1 << (kRowShift - 3);

#ifdef PSA_PHASE_TIMING
// debug build only: 100 MHz timestamps of the count kernel's phases, [block][phase]; rows 1000.. = re-score / tail
__device__ unsigned long long g_phase_ts[1024][8];
__device__ unsigned long long g_chunk_ts[1024][16];     // psa_count_kernel: sub-phases of the chunk hand-off and the selection (PSA_CPHASE)
#define PSA_FPHASE(k) do { if (tid == 0) g_phase_ts[1002][k] = wall_clock64(); } while (0)
#else
#define PSA_FPHASE(k) do { } while (0)
#endif

// ---- carry-save arithmetic on 32 independent bit lanes ----------------------
// v_bitop3_b32 (gfx950): any 3-input boolean function, full VALU rate
// (v_bfi_b32 and v_alignbit_b32 issue at half rate: tools/ubench/valu_rate.hip).
__device__ __forceinline__ unsigned xor3(unsigned a, unsigned b, unsigned c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
__device__ __forceinline__ unsigned maj3(unsigned a, unsigned b, unsigned c)
{
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xe8);
}
// full adder: sum -> a, carry returned
__device__ __forceinline__ unsigned csa(unsigned& a, unsigned b, unsigned c)
{
    const unsigned carry = maj3(a, b, c);
    a = xor3(a, b, c);
    return carry;
}

// One bit-sliced counter: planes[p] holds bit p of 32 independent counts.
// Values are pushed one per char; level L keeps one pending input so that
// every second arrival triggers a full adder whose carry moves up a level.
// Above the tree there is one lazy level: rounds are walked in pairs, the carry of the first round
// of a pair (weight 2^kLevels) is held and added together with the second round's by one more full
// adder — one ripple through the upper planes per two rounds.  HOW says what a round does with its
// carry: kHold (first of a pair), kJoin (second of a pair), kAlone (a class's odd last round).
enum CarryUse { kHold = 0, kJoin = 1, kAlone = 2 };

template <int PLANES>
struct SlicedCounterT {
    unsigned plane[PLANES];
    unsigned pend[kLevels];
    unsigned held;

    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int p = 0; p < PLANES; ++p) plane[p] = 0u;
    }

    // v of weight 2^P: ripple into the planes from P up (half adders)
    template <int P>
    __device__ __forceinline__ void ripple(unsigned v)
    {
#pragma unroll
        for (int p = P; p < PLANES - 1; ++p) {
            const unsigned c = plane[p] & v;
            plane[p] ^= v;
            v = c;
        }
        plane[PLANES - 1] ^= v;
    }

    // T = index of the char inside the round, L = level (both compile-time so the
    // whole carry-save tree is straight-line code)
    template <int L, int T, int HOW>
    __device__ __forceinline__ void push_level(unsigned v)
    {
        if constexpr (L == kLevels) {
            // the round's one carry of weight 2^kLevels
            if constexpr (HOW == kHold) held = v;
            else if constexpr (HOW == kJoin) ripple<kLevels + 1>(csa(plane[kLevels], held, v));
            else ripple<kLevels>(v);
        } else if constexpr ((T & 1) == 0) {
            pend[L] = v;
        } else {
            push_level<L + 1, (T >> 1), HOW>(csa(plane[L], pend[L], v));
        }
    }

    template <int T, int HOW>
    __device__ __forceinline__ void push(unsigned v) { push_level<0, T, HOW>(v); }
};
using SlicedCounter = SlicedCounterT<kPlanes>;

// ---- Seq2 in the order the count kernels read it -------------------------------
// Char i = 32k + J belongs to shift class J.  A char chunk holds `class_chars` chars k of
// every class; inside a chunk the classes lie from 31 down to 0 (a wave walks its classes
// downwards, so its symbols are one ascending run and the scalar prefetch simply runs on).
__host__ __device__ __forceinline__ size_t class_index(int i, int class_chars)
{
    const int J = i & (kClasses - 1), k = i >> 5;
    const int c = k / class_chars, kk = k - c * class_chars;
    return ((size_t)c * kClasses + (size_t)(kClasses - 1 - J)) * (size_t)class_chars + (size_t)kk;
}

// ---- Seq2 queries of a batch: ASCII -> codes, all queries in one launch -------
__global__ __launch_bounds__(256) void psa_encode_batch_kernel(const unsigned char* __restrict__ raw, int nq, int m,
                                                               size_t in_stride, unsigned int* __restrict__ rows,
                                                               unsigned char* __restrict__ bytes,
                                                               unsigned int* __restrict__ classes, int out_stride,
                                                               int class_chars, int class_total,
                                                               int* __restrict__ bad,
                                                               unsigned long long* __restrict__ zero, int zero_words)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int q = blockIdx.y;
    // the batch's per-query search scalars start from zero (saves a fill command in front of the batch)
    const long long g = ((long long)q * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
    if (g < zero_words) zero[g] = 0ull;
    if (i >= out_stride || q >= nq) return;
    int code = kCodePad;
    if (i < m) {
        const unsigned char c = raw[(size_t)q * in_stride + i];
        code = (c >= 'A' && c <= 'Z') ? c - 'A' : (c == '-' ? kCodeHyphen : kCodeBad);
        if (code == kCodeBad) { atomicOr(bad, 2); code = kCodePad; }
    }
    rows[(size_t)q * out_stride + i] = (unsigned int)code << kRowShift;   // symbol as a 4 KiB row offset
    bytes[(size_t)q * out_stride + i] = (unsigned char)code;
    // class order (count kernels); the kUnroll entries behind it are the last prefetch's padding
    if (i < class_total) classes[(size_t)q * out_stride + class_index(i, class_chars)] = (unsigned int)code << kRowShift;
    else if (i < class_total + kUnroll) classes[(size_t)q * out_stride + i] = (unsigned int)kCodePad << kRowShift;
}

// The class order of one Seq2 for another chunking than the one it was encoded for
__global__ __launch_bounds__(256) void psa_class_order_kernel(const unsigned char* __restrict__ bytes, int m,
                                                              unsigned int* __restrict__ classes, int class_chars,
                                                              int class_total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < class_total) classes[class_index(i, class_chars)] = (unsigned int)(i < m ? bytes[i] : kCodePad) << kRowShift;
    else if (i < class_total + kUnroll) classes[i] = (unsigned int)kCodePad << kRowShift;
}

// ---- Seq1: ASCII -> symbol codes (32 positions per thread), alphabet validation ----
struct PrepareArgs {
    const unsigned char* raw;         // Seq1 as ASCII (device), n bytes
    unsigned char*       codes;       // [32*nslots]: symbol codes, kCodePad behind n
    int*                 bad;
    int                  n;
    int                  nslots;
};

__device__ __forceinline__ void prepare_seq1_slot(const PrepareArgs& a, int slot)
{
    if (slot >= a.nslots) return;
    const int p0 = slot * 32;
    unsigned w[8];
    if (p0 + 32 <= a.n && (reinterpret_cast<size_t>(a.raw) & 15) == 0) {
        const uint4* src = reinterpret_cast<const uint4*>(a.raw + p0);
        const uint4 c0 = src[0], c1 = src[1];
        w[0] = c0.x; w[1] = c0.y; w[2] = c0.z; w[3] = c0.w; w[4] = c1.x; w[5] = c1.y; w[6] = c1.z; w[7] = c1.w;
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            unsigned v = 0u;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = p0 + 4 * q + r;
                v |= (p < a.n ? (unsigned)a.raw[p] : (unsigned)'\xff') << (8 * r);
            }
            w[q] = v;
        }
    }
    unsigned cw[8];
    bool any_bad = false;
#pragma unroll
    for (int t = 0; t < 32; ++t) {
        const unsigned c = (w[t >> 2] >> (8 * (t & 3))) & 0xffu;
        const unsigned code = c - 'A' < 26u ? c - 'A' : (c == '-' ? (unsigned)kCodeHyphen : (unsigned)kCodePad);
        if (code == (unsigned)kCodePad && p0 + t < a.n) any_bad = true;      // a real character outside [A-Z-]
        if ((t & 3) == 0) cw[t >> 2] = 0u;
        cw[t >> 2] |= code << (8 * (t & 3));
    }
    if (any_bad) atomicOr(a.bad, 1);
    uint4* dst = reinterpret_cast<uint4*>(a.codes + p0);
    dst[0] = make_uint4(cw[0], cw[1], cw[2], cw[3]);
    dst[1] = make_uint4(cw[4], cw[5], cw[6], cw[7]);
}

// The host-sequence setters have to report a character outside [A-Z-] before they return.  Instead of copying the flag
// back and waiting for the stream (a copy command, its completion signal, the runtime's wake-up: ~8 us), the encode
// launch's last workgroup to finish — `arrivals` (device, zero between launches) counts them — writes ticket << 40 | flag
// to `host_done` in pinned memory, which the host polls.  host_done == NULL: nobody is waiting (device-sequence setters).
struct DoneWord {
    unsigned*           arrivals;
    unsigned long long* host_done;
    const int*          bad;
    unsigned            ticket;
};

__device__ __forceinline__ void report_done(const DoneWord& d)
{
    if (!d.host_done) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's stores and flag updates have been performed,
    __syncthreads();                                      // then ONE lane's arrival
    if (threadIdx.x == 0) {
        const unsigned arrived = __hip_atomic_fetch_add(d.arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x * gridDim.y - 1) {
            __hip_atomic_store(d.arrivals, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned flag = (unsigned)__hip_atomic_load(d.bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(d.host_done, ((unsigned long long)d.ticket << 40) | flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ __launch_bounds__(256) void psa_prepare_seq1_kernel(PrepareArgs a, DoneWord done)
{
    prepare_seq1_slot(a, blockIdx.x * 256 + threadIdx.x);
    report_done(done);
}

// ---- Seq2 of a single search: ASCII -> the three encodings the engines read ----------
struct Seq2Args {
    const unsigned char* raw;       // ASCII (device), m bytes
    unsigned int*        rows;      // [capacity] code << kRowShift, natural order (sequential engine: scalar loads)
    unsigned char*       bytes;     // [capacity] codes (re-score)
    unsigned int*        classes;   // code << kRowShift in the count kernels' class order (class_index)
    int*                 bad;
    int m, capacity;
    int class_chars, class_total;   // the chunking the class order is laid out for
};

// sixteen chars per thread: the ASCII is read 16 bytes at a time (the sequence may lie in pinned HOST memory, where
// every load is a PCIe request), the byte codes and row offsets are stored as whole vectors
__device__ __forceinline__ void encode_seq2_16(const Seq2Args& a, int i0)
{
    if (i0 >= a.capacity) return;                        // (capacity is a multiple of 64)
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (i0 < a.m) {
        if (i0 + 16 <= a.m && ((reinterpret_cast<size_t>(a.raw) + (size_t)i0) & 15) == 0) {
            const uint4 q = *reinterpret_cast<const uint4*>(a.raw + i0);
            w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (i0 + 4 * q + r < a.m) w[q] |= (unsigned)a.raw[i0 + 4 * q + r] << (8 * r);
        }
    }
    unsigned codes[4];
    bool any_bad = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        codes[q] = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + 4 * q + r;
            const unsigned c = (w[q] >> (8 * r)) & 0xffu;
            unsigned code = kCodePad;
            if (i < a.m) {
                code = c - 'A' < 26u ? c - 'A' : (c == (unsigned)'-' ? (unsigned)kCodeHyphen : (unsigned)kCodePad);
                any_bad |= code == (unsigned)kCodePad;
            }
            codes[q] |= code << (8 * r);
            a.rows[i] = code << kRowShift;               // symbol as a 4 KiB row offset; the sequential engine shifts it back
            // the same symbols in the count kernels' class order (class_index); kUnroll entries of padding behind them
            if (i < a.class_total) a.classes[class_index(i, a.class_chars)] = code << kRowShift;
            else if (i < a.class_total + kUnroll) a.classes[i] = (unsigned)kCodePad << kRowShift;
        }
    }
    *reinterpret_cast<uint4*>(a.bytes + i0) = make_uint4(codes[0], codes[1], codes[2], codes[3]);
    if (any_bad) atomicOr(a.bad, 2);
}

__global__ __launch_bounds__(256) void psa_encode_seq2_kernel(Seq2Args a, DoneWord done)
{
    encode_seq2_16(a, (blockIdx.x * 256 + threadIdx.x) * 16);
    report_done(done);
}

// both sequences of a search in ONE launch (psa_set_sequences*, gpu_run_program): blocks [0, blocks1) take Seq1
__global__ __launch_bounds__(256) void psa_encode_pair_kernel(PrepareArgs a1, Seq2Args a2, int blocks1, DoneWord done)
{
    if ((int)blockIdx.x < blocks1) prepare_seq1_slot(a1, blockIdx.x * 256 + threadIdx.x);
    else encode_seq2_16(a2, (((int)blockIdx.x - blocks1) * 256 + threadIdx.x) * 16);
    report_done(done);
}

// ---- the same launch FOLLOWING the host's copy (psa_set_sequences, gpu_run_program on long sequences) -------------
// The sequences lie in pageable host memory; the host copies them into pinned, host-coherent staging in chunks and this
// kernel — launched BEFORE the copy starts — reads every chunk over PCIe as soon as the host has published it, so the
// launch latency and the host's copy run under the PCIe transfer instead of in front of it.  One 64-bit control word
// in the same pinned memory: ticket (bits 40-63: this launch's number), final (bit 39: Seq1's end is known and
// everything has been copied; the byte count is then Seq1's length — the caller of gpu_run_program does not pass it,
// the host finds the NUL while it copies), bytes (bits 0-38: bytes of Seq1 copied so far).  Seq2 is copied before the
// first publication.
// Workgroup 0 is the POLLER: one lane reads the host's word over PCIe and forwards every new value to a word in device
// memory, which the other workgroups watch.  (Every workgroup polling the host's word itself was measured: 130 reads of
// one cache line that the copying core keeps modified are answered one snoop at a time — the first workgroup was let go
// 11 us after its chunk had been published, the last 52 us after the launch.)  Nobody waits for anything but the host
// and the poller, which is dispatched first; the poller gives up after `spin_limit` polls (bad |= 8, "nothing arrived":
// the host redoes the transfer the plain way), the watchers after four times as many.
constexpr unsigned long long kStreamFinal = 1ull << 39;
constexpr unsigned long long kStreamBytes = kStreamFinal - 1;
constexpr int kStreamBad = 8;

__device__ __forceinline__ void stream_poller(const unsigned long long* host_word, unsigned long long* dev_word,
                                              unsigned ticket, int spin_limit, int* bad)
{
    if (threadIdx.x != 0) return;
    unsigned long long last = 0ull;
    for (int spins = 0;; ++spins) {
        const unsigned long long w = __hip_atomic_load(host_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(w >> 40) == ticket && w != last) {
            __hip_atomic_store(dev_word, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = w;
            if (w & kStreamFinal) return;
        }
        if (spins >= spin_limit) {
            atomicOr(bad, kStreamBad);
            __hip_atomic_store(dev_word, ((unsigned long long)ticket << 40) | kStreamFinal, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);           // "Seq1 is empty": every symbol is padding
            return;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// the forwarded word once it belongs to this launch and says that `need_bytes` of Seq1 are there (or that all is)
__device__ __forceinline__ unsigned long long stream_wait(const unsigned long long* dev_word, unsigned ticket,
                                                          unsigned long long need_bytes, int spin_limit, int* bad)
{
    __shared__ unsigned long long s_word;
    if (threadIdx.x == 0) {
        unsigned long long w = 0ull;
        for (long long spins = 0;; ++spins) {
            w = __hip_atomic_load(dev_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 40) == ticket && ((w & kStreamFinal) != 0ull || (w & kStreamBytes) >= need_bytes)) break;
            if (spins >= 4ll * spin_limit) {
                atomicOr(bad, kStreamBad);
                w = ((unsigned long long)ticket << 40) | kStreamFinal;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        s_word = w;
    }
    __syncthreads();
    // No cache maintenance: the polls are relaxed loads and there is no acquire fence (a system-scope acquire invalidates
    // the L2 — per poll and workgroup that was measured to double the kernel's time).  None is needed: the staging
    // buffer is fine-grained host memory, which the GPU does not cache, and no lane of this launch has touched the bytes
    // it is about to read; the loads below are issued after the poll that saw the publication has returned.
    asm volatile("" ::: "memory");
    return s_word;
}

__global__ __launch_bounds__(256) void psa_encode_pair_streamed_kernel(PrepareArgs a1, Seq2Args a2, int blocks1,
                                                                       const unsigned long long* host_word,
                                                                       unsigned long long* dev_word, unsigned ticket,
                                                                       int spin_limit, DoneWord done)
{
#ifdef PSA_PHASE_TIMING
#define PSA_EPHASE(k) do { if (threadIdx.x == 0 && blockIdx.x < 1000) g_phase_ts[blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define PSA_EPHASE(k) do { } while (0)
#endif
    PSA_EPHASE(0);
    const int block = (int)blockIdx.x - 1;
    if (blockIdx.x == 0) {
        stream_poller(host_word, dev_word, ticket, spin_limit, a1.bad);
        PSA_EPHASE(1);
    } else if (block < blocks1) {
        const unsigned long long end = ((unsigned long long)block + 1ull) * 256ull * 32ull;       // bytes this block reads
        const unsigned long long w = stream_wait(dev_word, ticket, end, spin_limit, a1.bad);
        // not final: the host has copied `end` bytes without meeting Seq1's end, so all of this block's are symbols
        a1.n = (w & kStreamFinal) ? (int)(w & kStreamBytes) : (int)end;
        PSA_EPHASE(1);
        prepare_seq1_slot(a1, block * 256 + threadIdx.x);
        PSA_EPHASE(2);
    } else {
        (void)stream_wait(dev_word, ticket, 0ull, spin_limit, a2.bad);     // Seq2 is in place before the first publication
        encode_seq2_16(a2, ((block - blocks1) * 256 + threadIdx.x) * 16);
    }
    report_done(done);
}

#ifndef PSA_PLANES_INLINE
#define PSA_PLANES_INLINE __attribute__((noinline))
#endif
// ---- class planes of a slot (32 positions of Seq1) without table lookups -----------------------------------------
// Row b of a slot: bit t of `lo` = class(Seq1[32 slot + t], b) in {'*', '.'}, of `hi` = class in {':', '.'}.  Round 4's
// first form looked 32 symbols up in a 64-word table and transposed the 32x32 bit matrix, once for lo and once for hi
// (two threads per slot, ~450 instructions each: the plane build was 3.7 us of VALU work in every workgroup).  Here one
// thread turns the slot's 32 bytes into BIT PLANES of the symbol code (byte shuffle + three butterfly stages on 8 dwords:
// an 8x8 bit transpose in four byte lanes at once), forms the 27 masks "position holds symbol a" as products of two
// minterms, and ORs them per row — which symbols belong to a row is the reference's group relation, a compile-time
// constant (psa_internal.hpp), so the ORs are straight-line code: ~230 instructions for both planes of a slot.
constexpr unsigned plane_symbols(int b, bool hi)        // Seq1 symbols a (bit a) whose pair with Seq2 symbol b is in the plane
{
    unsigned m = 0u;
    if (b > kCodeHyphen) return 0u;                     // the padding symbol's row is empty
    for (int a = 0; a <= kCodeHyphen; ++a) {
        const char s = sign_of_codes(a, b);
        if (hi ? (s == PSA_SIGN_COLON || s == PSA_SIGN_DOT) : (s == PSA_SIGN_STAR || s == PSA_SIGN_DOT)) m |= 1u << a;
    }
    return m;
}

// w[q] = bytes 4q..4q+3 of the slot -> plane[j] bit t = bit j of byte t
__device__ __forceinline__ void byte_bit_planes(const unsigned (&w)[8], unsigned (&plane)[8])
{
    // dword u <- bytes u, u + 8, u + 16, u + 24 (v_perm_b32: selector bytes 0-3 name the second source, 4-7 the first)
    unsigned v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int q = u >> 2, r = u & 3;
        const unsigned sel = (unsigned)r | ((unsigned)(4 + r) << 8) | 0x0c0c0000u;
        const unsigned x = __builtin_amdgcn_perm(w[q + 2], w[q], sel);          // bytes u, u + 8
        const unsigned y = __builtin_amdgcn_perm(w[q + 6], w[q + 4], sel);      // bytes u + 16, u + 24
        v[u] = __builtin_amdgcn_perm(y, x, 0x05040100u);
    }
    // 8x8 bit transpose of (dword index, bit inside the byte), all four byte lanes at once
    unsigned m = 0x0f0f0f0fu;
#pragma unroll
    for (int j = 4; j != 0; j >>= 1, m ^= (m << j)) {
#pragma unroll
        for (int k = 0; k < 8; k = (k + j + 1) & ~j) {
            const unsigned x = v[k], y = v[k + j];
            v[k] = __builtin_amdgcn_bitop3_b32(x, y << j, m << j, 0xd8);
            v[k + j] = __builtin_amdgcn_bitop3_b32(y, x >> j, m, 0xd8);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) plane[j] = v[j];
}

// the rows' symbol sets, evaluated by the compiler (psa_create compares them with the host tables' relation)
struct PlaneSets { unsigned lo[kMaskRows], hi[kMaskRows]; };
constexpr PlaneSets make_plane_sets()
{
    PlaneSets p{};
    for (int b = 0; b < kMaskRows; ++b) { p.lo[b] = plane_symbols(b, false); p.hi[b] = plane_symbols(b, true); }
    return p;
}
inline constexpr PlaneSets kPlaneSets = make_plane_sets();

// v_bitop3_b32 looks its result up at index 4 S0 + 2 S1 + S2.  out[x] = "(a, b, c) spell x" (a = bit 0 of x)
template <int... X>
__device__ __forceinline__ void minterms3(unsigned a, unsigned b, unsigned c, unsigned (&out)[8], std::integer_sequence<int, X...>)
{
    ((out[X] = __builtin_amdgcn_bitop3_b32(a, b, c, 1u << (((X & 1) << 2) | (X & 2) | ((X >> 2) & 1)))), ...);
}
// out[y] = "(a, b) spell y" and gate == GATE
template <int GATE, int... Y>
__device__ __forceinline__ void minterms2(unsigned a, unsigned b, unsigned gate, unsigned (&out)[4], std::integer_sequence<int, Y...>)
{
    ((out[Y] = __builtin_amdgcn_bitop3_b32(a, b, gate, 1u << (((Y & 1) << 2) | (Y & 2) | GATE))), ...);
}

// The 27 masks eq[a]: bit t set where position t of the slot holds symbol a.  RAW: the bytes are the caller's ASCII
// ('A'..'Z', '-'; *other = positions holding anything else, 0 bytes included), else symbol codes 0..27 (27 = padding: in
// no mask).
template <bool RAW>
__device__ __forceinline__ void symbol_masks(const unsigned (&w)[8], unsigned (&eq)[27], unsigned* other)
{
    unsigned p[8];
    byte_bit_planes(w, p);
    unsigned low[8];                                    // minterms of bits 0-2
    minterms3(p[0], p[1], p[2], low, std::make_integer_sequence<int, 8>{});
    if constexpr (!RAW) {
        unsigned high[4];                               // minterms of bits 3-4 (bits 5-7 of a code are zero)
        minterms2<0>(p[3], p[4], 0u, high, std::make_integer_sequence<int, 4>{});
#pragma unroll
        for (int a = 0; a <= kCodeHyphen; ++a) eq[a] = low[a & 7] & high[a >> 3];
    } else {
        // letters 0x41..0x5a: bits 7-5 = 010, bits 4-0 = 1..26;  '-' = 0x2d: bits 7-5 = 001, bits 4-0 = 01101
        const unsigned letter = __builtin_amdgcn_bitop3_b32(p[5], p[6], p[7], 1u << 2);       // !b5 & b6 & !b7
        const unsigned hyphen = __builtin_amdgcn_bitop3_b32(p[5], p[6], p[7], 1u << 4);       //  b5 & !b6 & !b7
        unsigned high[4];                               // minterms of bits 3-4, and a letter
        minterms2<1>(p[3], p[4], letter, high, std::make_integer_sequence<int, 4>{});
#pragma unroll
        for (int a = 0; a < kCodeHyphen; ++a) eq[a] = low[(a + 1) & 7] & high[(a + 1) >> 3];
        eq[kCodeHyphen] = low[5] & __builtin_amdgcn_bitop3_b32(p[3], p[4], hyphen, 1u << ((1 << 2) | 1));   // b3 & !b4 & hyphen
        unsigned any = 0u;
#pragma unroll
        for (int a = 0; a <= kCodeHyphen; ++a) any |= eq[a];
        *other = ~any;
    }
}

// rows 0..kMaskRows-1 of one slot into the plane rows in LDS: {lo, hi} at dst[(b * row_slots + slot) * 2]
template <int B = 0>
__device__ __forceinline__ void store_plane_rows(const unsigned (&eq)[27], uint2* dst, int row_slots, int slot)
{
    if constexpr (B < kMaskRows) {
        constexpr unsigned lo_set = kPlaneSets.lo[B], hi_set = kPlaneSets.hi[B];
        unsigned lo = 0u, hi = 0u;
#pragma unroll
        for (int a = 0; a <= kCodeHyphen; ++a) {
            if ((lo_set >> a) & 1u) lo |= eq[a];
            if ((hi_set >> a) & 1u) hi |= eq[a];
        }
        dst[B * row_slots + slot] = make_uint2(lo, hi);
        store_plane_rows<B + 1>(eq, dst, row_slots, slot);
    }
}

// The class planes of a tile's window: slots [slot0, slot0 + lds_slots) of Seq1 -> rows in LDS, one thread per slot.
// RAW: from the caller's ASCII (never read behind its end: positions from n1 on count as byte 0 and are left out of the
// alphabet check), else from the encoded Seq1.  Not inlined: the count loop's schedule and registers must not depend on
// this code (a 2 % difference of the whole loop was measured between two forms of the plane build).
template <int NT, bool RAW>
__device__ PSA_PLANES_INLINE void build_tile_planes(const unsigned char* __restrict__ s1, const unsigned char* __restrict__ raw1, int n1,
                                                    int* bad, int slot0, int lds_slots, uint2* rows, int tid)
{
    const uint4* src = reinterpret_cast<const uint4*>(s1 + ((size_t)slot0 << 5));
    const bool raw_aligned = RAW && (reinterpret_cast<size_t>(raw1) & 15) == 0;
    bool bad_seen = false;
    for (int slot = tid; slot < lds_slots; slot += NT) {               // (at most kRowSlots slots: one per thread)
        unsigned w[8];
        if constexpr (!RAW) {
            const uint4 c0 = src[2 * slot], c1 = src[2 * slot + 1];
            w[0] = c0.x; w[1] = c0.y; w[2] = c0.z; w[3] = c0.w; w[4] = c1.x; w[5] = c1.y; w[6] = c1.z; w[7] = c1.w;
        } else {
            const int p0 = (slot0 + slot) << 5;
            if (raw_aligned && p0 + 32 <= n1) {
                const uint4* rs = reinterpret_cast<const uint4*>(raw1 + p0);
                const uint4 c0 = rs[0], c1 = rs[1];
                w[0] = c0.x; w[1] = c0.y; w[2] = c0.z; w[3] = c0.w; w[4] = c1.x; w[5] = c1.y; w[6] = c1.z; w[7] = c1.w;
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    unsigned v = 0u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int p = p0 + 4 * q + r;
                        v |= (p < n1 ? (unsigned)raw1[p] : 0u) << (8 * r);
                    }
                    w[q] = v;
                }
            }
        }
        unsigned eq[27], other = 0u;
        symbol_masks<RAW>(w, eq, &other);
        if constexpr (RAW) {
            const int in_range = n1 - ((slot0 + slot) << 5);                // chars of this slot inside Seq1
            const unsigned range = in_range >= 32 ? ~0u : in_range > 0 ? (1u << in_range) - 1u : 0u;
            bad_seen |= (other & range) != 0u;
        }
        store_plane_rows(eq, rows, kRowSlots, slot);
    }
    if constexpr (RAW) {
        if (bad_seen) atomicOr(bad, 1);                 // a character of Seq1 outside [A-Z-]
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's symbol stores have been performed
    }
}

// ---- LDS reads of the count loop, issued one char ahead ------------------------
// Two plain ds_read_b64 per char (slots l+k and l+64+k of the symbol's row): 2 LDS cycles each,
// consecutive lanes on consecutive 8-byte slots (conflict-free).  Left to the compiler they would
// be merged into ds_read2_b64 (8 cycles) and waited for right after issue; here they are issued
// for char T+1 before char T is processed and retired by an explicit s_waitcnt (inline asm: hipcc
// does not count these loads, the wait statement names every destination so no consumer can be
// scheduled above it).
struct SlotRegs {
    unsigned long long v[2];
};

template <int IMM>
__device__ __forceinline__ void lds_issue(SlotRegs& n, unsigned addr)
{
    asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"
                 : "=&v"(n.v[0]), "=&v"(n.v[1])
                 : "v"(addr), "i"(IMM), "i"(IMM + 512));
}

__device__ __forceinline__ void lds_retire(SlotRegs& n)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n.v[0]), "+v"(n.v[1]));
}

// One char of a shift class: T is its index inside the round.  Char k of the class reads slot
// l + k (word l) and l + 64 + k (word l+64) of its symbol's row: the round's base address moves
// on by kUnroll slots per round, T is an immediate offset.  `cur` holds this char's slots; the
// next char's are requested first (T == 31: char 0 of the next round).
// BASE: byte offset of the round's first slot from lane_addr (an immediate: the second round of a pair reads
// kUnroll slots further on without another address add).
template <int T, int HOW, int BASE, typename Counter>
__device__ __forceinline__ void class_step(Counter (&cnt)[2][kCounters], SlotRegs& cur,
                                           const unsigned (&rowoff)[kUnroll], unsigned next_rowoff,
                                           unsigned lane_addr)
{
    SlotRegs nxt;
    // Seq2 symbol of the next char as a wave-uniform LDS row offset (symbol << kRowShift)
    if constexpr (T + 1 < kUnroll) lds_issue<BASE + 8 * (T + 1)>(nxt, lane_addr + rowoff[(T + 1) % kUnroll]);
    else lds_issue<BASE + 8 * kUnroll>(nxt, lane_addr + next_rowoff);
    const unsigned lo0 = (unsigned)cur.v[0], hi0 = (unsigned)(cur.v[0] >> 32);
    const unsigned lo1 = (unsigned)cur.v[1], hi1 = (unsigned)(cur.v[1] >> 32);
    cnt[0][0].template push<T, HOW>(lo0);
    cnt[0][1].template push<T, HOW>(hi0);
    cnt[0][2].template push<T, HOW>(lo0 & hi0);
    cnt[1][0].template push<T, HOW>(lo1);
    cnt[1][1].template push<T, HOW>(hi1);
    cnt[1][2].template push<T, HOW>(lo1 & hi1);
    lds_retire(nxt);
    cur = nxt;
}

template <int HOW, int BASE, typename Counter, int... Ts>
__device__ __forceinline__ void class_round(Counter (&cnt)[2][kCounters], SlotRegs& cur,
                                            const unsigned (&rowoff)[kUnroll], unsigned next_rowoff,
                                            unsigned lane_addr, std::integer_sequence<int, Ts...>)
{
    (class_step<Ts, HOW, BASE>(cnt, cur, rowoff, next_rowoff, lane_addr), ...);
}

// Counter planes moved down by `s` bit positions (C[x] <- C[x + s], 1 <= s <= 31): word w takes
// its upper bits from word w+1, which the next lane holds (DPP wave rotate; lane 63's word 63 is
// followed by lane 0's word 64).  Word 127 has no successor in the tile: what lane 63 receives
// for it (lane 0's word 64) is spill — offsets of the tile's last word are not used (kTileStride).
template <typename Counter>
__device__ __forceinline__ void shift_planes(Counter (&cnt)[2][kCounters], unsigned s, int lane)
{
    const bool last_lane = lane == 63;
#pragma unroll
    for (int k = 0; k < kCounters; ++k)
#pragma unroll
        for (int p = 0; p < (int)(sizeof(cnt[0][k].plane) / sizeof(unsigned)); ++p) {
            const unsigned a = cnt[0][k].plane[p], b = cnt[1][k].plane[p];
            // wave_rol:1: lane l <- lane l+1.  Every lane is written, so the DPP's tied `old` operand is a value
            // nobody computes (an empty asm names a register): no copy in front of the DPP mov, no hazard nop.
            int ua, ub;
            asm volatile("" : "=v"(ua));          // (volatile: two distinct registers, not one merged value)
            asm volatile("" : "=v"(ub));
            const unsigned ra = (unsigned)__builtin_amdgcn_update_dpp(ua, (int)a, 0x134, 0xf, 0xf, false);
            const unsigned rb = (unsigned)__builtin_amdgcn_update_dpp(ub, (int)b, 0x134, 0xf, 0xf, false);
            cnt[0][k].plane[p] = __builtin_amdgcn_alignbit(last_lane ? rb : ra, a, s);
            cnt[1][k].plane[p] = __builtin_amdgcn_alignbit(rb, b, s);
        }
}

// The main loop of one wave: `chars` consecutive chars (a multiple of kUnroll) of the workgroup's class-ordered
// symbol stream (class_index: class 31 first, `class_chars` chars per class), starting at char `first_k` of a class —
// a wave's share need not be whole classes (the hardware does not treat the waves of a SIMD alike, see the kernels'
// deals).  `sym` = the symbols from the share's first char on, `lane_base` = LDS byte address of slot `lane` of the
// chunk's first column, `low_class` = the lowest class the share touches.  On return cnt holds the share's part of
// sum_J D_J[x + J], shifted down: the wave's contribution to the counts.
template <typename Counter, typename RowOf>
__device__ __forceinline__ void count_stream(Counter (&cnt)[2][kCounters],
                                             const __attribute__((address_space(4))) unsigned* sym,
                                             int first_k, int chars, int class_chars, unsigned low_class,
                                             unsigned lane_base, int lane, RowOf row_of)
{
    // the row offsets of a round are fetched one round ahead, the LDS slots one char ahead;
    // the last prefetches of a wave run past its range into valid padding and are never used
    // two sets of row offsets: a round reads one while the next round's arrive in the other (no copies;
    // the code-byte budget of this loop is what bounds it)
    unsigned rowoff_a[kUnroll], rowoff_b[kUnroll];
#pragma unroll
    for (int q = 0; q < kUnroll; ++q) rowoff_a[q] = row_of(sym[q]);
    unsigned lane_addr;
    SlotRegs cur;
    bool first = true;
    while (chars > 0) {
        const int room = class_chars - first_k;
        const int n = chars < room ? chars : room;       // this class's part of the share
        if (!first) shift_planes(cnt, 1u, lane);         // Horner step: C[x] <- C[x+1], then add the next class
        first = false;
        lane_addr = lane_base + 8u * (unsigned)first_k;
        lds_issue<0>(cur, lane_addr + rowoff_a[0]);
        lds_retire(cur);
        int g = 0;
        for (; g + 2 * kUnroll <= n; g += 2 * kUnroll) {
            sym += kUnroll;
#pragma unroll
            for (int q = 0; q < kUnroll; ++q) rowoff_b[q] = row_of(sym[q]);   // wave-uniform
            class_round<kHold, 0>(cnt, cur, rowoff_a, rowoff_b[0], lane_addr, std::make_integer_sequence<int, kUnroll>{});
            sym += kUnroll;
#pragma unroll
            for (int q = 0; q < kUnroll; ++q) rowoff_a[q] = row_of(sym[q]);
            class_round<kJoin, 8 * kUnroll>(cnt, cur, rowoff_b, rowoff_a[0], lane_addr, std::make_integer_sequence<int, kUnroll>{});
            lane_addr += 16u * kUnroll;
        }
        if (g < n) {                                     // an odd last round
            sym += kUnroll;
#pragma unroll
            for (int q = 0; q < kUnroll; ++q) rowoff_b[q] = row_of(sym[q]);
            class_round<kAlone, 0>(cnt, cur, rowoff_a, rowoff_b[0], lane_addr, std::make_integer_sequence<int, kUnroll>{});
#pragma unroll
            for (int q = 0; q < kUnroll; ++q) rowoff_a[q] = rowoff_b[q];
        }
        chars -= n;
        first_k = 0;
    }
    if (low_class) shift_planes(cnt, low_class, lane);
}

// A wave's share [lo, hi) of the stream in rounds of kUnroll chars -> count_stream's arguments
struct WaveShare {
    int sym_at;        // first symbol of the share (offset into the chunk's stream)
    int first_k;       // char inside its first class
    int chars;
    unsigned low_class;
};
__device__ __forceinline__ WaveShare wave_share(int lo_round, int hi_round, int class_chars)
{
    WaveShare s;
    s.sym_at = lo_round * kUnroll;
    s.chars = (hi_round - lo_round) * kUnroll;
    const int cls0 = s.sym_at / class_chars;
    s.first_k = s.sym_at - cls0 * class_chars;
    const int last = s.chars > 0 ? (hi_round * kUnroll - 1) / class_chars : cls0;
    s.low_class = (unsigned)(kClasses - 1 - last);
    return s;
}

// ---- counter planes -> integer counts, summed over the waves in LDS ----------------
// cells[counter][t][word] (dwords): bits 0-15 = the count of position 32*word + t, bits 16-31 that of
// position 32*word + 16 + t.  A workgroup's counts stay below 2^16 (at most 16 x 1023 or 8 x 2047 chars),
// so the two halves never carry into each other.
constexpr int kTileWords = kTile / 32;
constexpr int kCellDwords = kCounters * 16 * kTileWords;
constexpr int kPlaneBytes = kMaskRows * kRowSlots * 8;          // the class-plane rows of a tile in LDS
// psa_count_kernel sums its 16 waves' counter planes BIT-SLICED before anything is converted (sum_dump / sum_take / sum_columns): the
// exchange area, 8 slots of one wave's six 11-plane counters each, lies over the plane rows (dead by then) and a little
// beyond them; the count cells follow it.
constexpr int kXPairs = kPlanes / 2;                            // plane pairs of a (counter, word), one ds_*_b64 each
constexpr int kXKwDwords = kXPairs * 128 + 64;                  // one (counter, word) of a wave: 5 pairs x 64 lanes x 2, plane 10 x 64 lanes
constexpr int kXSlotDwords = 2 * kCounters * kXKwDwords;        // one wave: 4224 dwords
constexpr int kXBytes = 8 * kXSlotDwords * 4;                   // 135,168
static_assert(kPlanes == 10 && kXBytes >= kPlaneBytes, "exchange area: ten-plane counters, behind the plane rows' end");
constexpr int kCellOffset = kXBytes;
constexpr int kRawPerThread = 4;    // raw launches: positions of Seq2's class order one thread of a converter workgroup writes
constexpr int kCountLdsBytes = kCellOffset + kCellDwords * 4;   // dynamic LDS of psa_count_kernel: 159,744 bytes

// 16 plane words (planes 0..15 of 32 positions) -> 16 dwords, dword t = count(t) | count(t+16) << 16:
// the lower four stages of the 32x32 bit-matrix transpose (Hacker's Delight 7-3), applied to both
// 16-bit halves at once
__device__ __forceinline__ void planes_to_counts(unsigned (&a)[16])
{
    unsigned m = 0x00ff00ffu;
#pragma unroll
    for (int j = 8; j != 0; j >>= 1, m ^= (m << j)) {
#pragma unroll
        for (int k = 0; k < 16; k = (k + j + 1) & ~j) {
            const unsigned x = a[k], y = a[k + j];              // (butterflies as in byte_bit_planes: shift + select)
            a[k] = __builtin_amdgcn_bitop3_b32(x, y << j, m << j, 0xd8);
            a[k + j] = __builtin_amdgcn_bitop3_b32(y, x >> j, m, 0xd8);
        }
    }
}

// `words` = words per row of the cell array (kTileWords, or kTileWords - 1 where the unused last word is
// not stored); `all_words` = false skips lane 63's second word (word 127)
template <typename Counter>
__device__ __forceinline__ void add_counts_to_cells(Counter (&cnt)[2][kCounters], unsigned* cells, int words,
                                                    int lane, bool all_words)
{
    constexpr int kHave = (int)(sizeof(cnt[0][0].plane) / sizeof(unsigned));
    static_assert(kHave <= 16, "a wave's counts fit 16 bits");
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const int word = w * 64 + lane;
        if (!all_words && word >= words) continue;
#pragma unroll
        for (int k = 0; k < kCounters; ++k) {
            unsigned v[16];
#pragma unroll
            for (int p = 0; p < 16; ++p) v[p] = p < kHave ? cnt[w][k].plane[p] : 0u;
            planes_to_counts(v);
#pragma unroll
            for (int t = 0; t < 16; ++t)
                __hip_atomic_fetch_add(cells + (k * 16 + t) * words + word, v[t], __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// ---- psa_count_kernel: the 16 waves' counters summed as bit planes, converted ONCE ------------------------------
// add_counts_to_cells costs a wave ~1,050 instructions (six 16x16 bit transposes and 96 ds_adds): sixteen waves doing it
// were 8.6 % of the kernel's instructions.  Adding two bit-sliced counters is one full adder per plane (20 instructions
// per counter word), so the waves' planes are added first and only the totals are converted:
//   1. waves 8-15 store their planes to the exchange area (sum_dump), barrier;
//   2. wave w < 8 adds wave w + 8's planes to its own registers (sum_take: 11 planes) and stores the sum over what it
//      has just read, barrier;
//   3. 768 threads, one per (counter, word, lane, half): each adds four of the eight sums (13 planes), the halves meet
//      through v_permlane32_swap (14 planes), the lower half converts its column (one 16x16 transpose) and stores the
//      integer counts to the cells (plain stores: nothing else adds to them), barrier.
// ~5,800 instructions per workgroup instead of 16,900.  The area lies over the plane rows, so step 1 starts behind a
// barrier that every wave's main loop has passed.
template <typename Counter>
__device__ __forceinline__ void sum_dump(const Counter (&cnt)[2][kCounters], const unsigned (*top)[kCounters], unsigned* slot, int lane)
{
#pragma unroll
    for (int k = 0; k < kCounters; ++k)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            unsigned* at = slot + (k * 2 + w) * kXKwDwords;
#pragma unroll
            for (int pr = 0; pr < kXPairs; ++pr)
                *reinterpret_cast<uint2*>(at + pr * 128 + lane * 2) = make_uint2(cnt[w][k].plane[2 * pr], cnt[w][k].plane[2 * pr + 1]);
            if (top) at[kXPairs * 128 + lane] = top[w][k];
        }
}

// cnt += the ten planes another wave stored in `slot`; the carry out of plane 9 goes to top (plane 10)
template <typename Counter>
__device__ __forceinline__ void sum_take(Counter (&cnt)[2][kCounters], unsigned (&top)[2][kCounters], const unsigned* slot, int lane)
{
#pragma unroll
    for (int k = 0; k < kCounters; ++k)
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const unsigned* at = slot + (k * 2 + w) * kXKwDwords;
            unsigned in[kPlanes];
#pragma unroll
            for (int pr = 0; pr < kXPairs; ++pr) {
                const uint2 v = *reinterpret_cast<const uint2*>(at + pr * 128 + lane * 2);
                in[2 * pr] = v.x;
                in[2 * pr + 1] = v.y;
            }
            unsigned carry = 0u;
#pragma unroll
            for (int p = 0; p < kPlanes; ++p) carry = csa(cnt[w][k].plane[p], in[p], carry);
            top[w][k] = carry;
        }
}

// step 3 (threads 0..767 of the workgroup; X holds the eight 11-plane sums)
__device__ __forceinline__ void sum_columns(const unsigned* X, unsigned* cells, int tid)
{
    const int lane = tid & 63, wv = tid >> 6;               // wv 0..11
    const int half = lane >> 5;                             // which four of the eight sums
    const int l = ((wv & 1) << 5) | (lane & 31);            // lane of the main loop = word l (w = 0) or 64 + l (w = 1)
    const int kw = wv >> 1;                                 // counter k = kw >> 1, word half w = kw & 1
    const unsigned* base = X + kw * kXKwDwords;
    unsigned acc[14];
    auto load = [&](int src, unsigned (&in)[11]) {
        const unsigned* at = base + src * kXSlotDwords;
#pragma unroll
        for (int pr = 0; pr < kXPairs; ++pr) {
            const uint2 v = *reinterpret_cast<const uint2*>(at + pr * 128 + l * 2);
            in[2 * pr] = v.x;
            in[2 * pr + 1] = v.y;
        }
        in[10] = at[kXPairs * 128 + l];
    };
    {
        unsigned in[11];
        load(4 * half, in);
#pragma unroll
        for (int p = 0; p < 11; ++p) acc[p] = in[p];
        acc[11] = acc[12] = acc[13] = 0u;
    }
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        unsigned in[11];
        load(4 * half + q, in);
        unsigned carry = 0u;
#pragma unroll
        for (int p = 0; p < 11; ++p) carry = csa(acc[p], in[p], carry);
#pragma unroll
        for (int p = 11; p < 13; ++p) {                     // half adders: four 11-bit numbers stay below 2^13
            const unsigned t = acc[p] & carry;
            acc[p] ^= carry;
            carry = t;
        }
    }
    // the other half's sum: lanes 0-31 receive what lanes 32-63 hold (v_permlane32_swap, gfx950)
    unsigned carry = 0u;
#pragma unroll
    for (int p = 0; p < 13; ++p) {
        const auto r = __builtin_amdgcn_permlane32_swap(acc[p], acc[p], false, false);
        carry = csa(acc[p], (unsigned)r[1], carry);
    }
    acc[13] = carry;
    if (half == 0) {
        unsigned v[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) v[p] = p < 14 ? acc[p] : 0u;
        planes_to_counts(v);
        unsigned* out = cells + (kw >> 1) * 16 * kTileWords + (kw & 1) * 64 + l;
#pragma unroll
        for (int t = 0; t < 16; ++t) out[t * kTileWords] = v[t];
    }
}

// counts of the PER consecutive positions starting at o_local (PER divides 16: one word, one half)
template <int PER>
__device__ __forceinline__ void read_cells(unsigned (&n)[kCounters][PER], const unsigned* cells, int words, int o_local)
{
    const int word = o_local >> 5;
    const int t0 = o_local & 15;
    const unsigned sh = (unsigned)(o_local & 16);
#pragma unroll
    for (int k = 0; k < kCounters; ++k)
#pragma unroll
        for (int e = 0; e < PER; ++e) n[k][e] = word < words ? (cells[(k * 16 + t0 + e) * words + word] >> sh) & 0xffffu : 0u;
}

// ---- score interval of one offset from its exact counts ------------------------
struct BoundParams {
    double class_best[4];    // most favourable substitution delta inside each class
    double class_worst[4];   // least favourable delta inside each class
    double slack;            // rounding bound of the sequential sum + this formula
    double opt_all, pes_all; // the best / worst delta over all four classes: what offset_bounds finds whenever every
                             // class occurs at least once in an offset (always, for long random sequences)
    // the order-free score in "bigger is better" form as three multiply-adds of the counts: with n_ = m - n* - n: - n.,
    //   sgn (w1 n* - w2 n: - w3 n. - w4 n_) = c[0] n* + c[1] n: + c[2] n. + c_m      (host: bound_params)
    double c[3], c_m;
    double up_all, pes_all_add;   // sgn opt_all + slack, sgn pes_all - slack: the interval's ends relative to that score
    int m;
    int is_max;
};

// monotone map double -> u64 (total order, -0 < +0)
__device__ __forceinline__ unsigned long long order_key(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_order(unsigned long long k)
{
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// optimistic ("upper") and pessimistic score of an offset in "bigger is better" form
// (MIN searches are negated), each widened by the rounding slack
// ALL_PRESENT: the caller has checked that all four classes occur in this offset (then the fold over the
// classes that occur is the fold over all of them, precomputed by the host with the same operations).
template <bool ALL_PRESENT>
__device__ __forceinline__ void offset_bounds(const BoundParams& b, int n_lo, int n_hi, int n_both,
                                              double& upper, double& pess)
{
    const bool mx = b.is_max != 0;
    const int n0 = n_lo - n_both, n1 = n_hi - n_both, n2 = n_both;   // stars, colons, dots
    // (three fused multiply-adds; the rounding of the pre-added coefficients, <= 6 u m max|w|, is far inside the slack,
    // which allows (m + 32) u m max|w| twice over: filter_slack)
    const double total = fma(b.c[0], (double)n0, fma(b.c[1], (double)n1, fma(b.c[2], (double)n2, b.c_m)));
    if constexpr (ALL_PRESENT) {
        upper = total + b.up_all;
        pess = total + b.pes_all_add;
    } else {
        const int n[4] = {n0, n1, n2, b.m - n0 - n1 - n2};
        double d_opt, d_pes;
        d_opt = d_pes = mx ? -INFINITY : INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (n[k] > 0) {
                d_opt = mx ? fmax(d_opt, b.class_best[k]) : fmin(d_opt, b.class_best[k]);
                d_pes = mx ? fmax(d_pes, b.class_worst[k]) : fmin(d_pes, b.class_worst[k]);
            }
        const double sgn = mx ? 1.0 : -1.0;
        upper = (total + sgn * d_opt) + b.slack;
        pess = (total + sgn * d_pes) - b.slack;
    }
}

// does every class occur in the offset with these counts?
__device__ __forceinline__ bool all_classes_present(int m, unsigned n_lo, unsigned n_hi, unsigned n_both)
{
    return n_lo > n_both && n_hi > n_both && n_both > 0u && n_lo + n_hi < (unsigned)m + n_both;
}

// Score intervals of a thread's PER offsets (index o0 + e relative to the searched range, valid while below
// `limit` offsets of the tile and inside [0, span)); returns the thread's best pessimistic bound.  The whole
// wave takes the short path when every class occurs in every one of its offsets.
template <int PER>
__device__ __forceinline__ double thread_bounds(const BoundParams& b, const unsigned (&n)[kCounters][PER], int o0, int span,
                                                int valid, double (&up)[PER])
{
    bool all = true;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        const bool in = o0 + e >= 0 && o0 + e < span && e < valid;
        all &= !in || all_classes_present(b.m, n[0][e], n[1][e], n[2][e]);
    }
    const bool fast = __ballot(!all) == 0ull;          // wave-uniform
    double pess = -INFINITY;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
        up[e] = -INFINITY;
        if (o0 + e >= 0 && o0 + e < span && e < valid) {
            double pe;
            if (fast) offset_bounds<true>(b, (int)n[0][e], (int)n[1][e], (int)n[2][e], up[e], pe);
            else offset_bounds<false>(b, (int)n[0][e], (int)n[1][e], (int)n[2][e], up[e], pe);
            pess = fmax(pess, pe);
        }
    }
    return pess;
}

// ---- values other workgroups wrote / will read inside the same launch ----------------
// Write-through (sc1) stores and agent-scope loads: they pass the XCD's L2, so the hand-off needs no L2 write-back
// (MI355X_MICROARCH.md, "Valid forms": every storing wave drains its stores, workgroup barrier, ONE lane's arrival).
__device__ __forceinline__ void store_agent(int* p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_agent(double* p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int load_agent(const int* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long load_agent(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_agent(const double* p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// ---- rescore: the reference's sequential fp64 sum for each candidate -----------
// One wave per candidate.  The additions of cpu_funcs.c:278 are a serial chain of
// m roundings, but while the running sum stays inside one binade every addend is
// rounded to that binade's ulp, i.e. the chain is an INTEGER walk in units of the
// ulp:  S += R_e[class]  with R_e[c] = round(p_c / ulp_e).  Integer adds commute,
// so a batch of 512 chars is summed with a wave-wide prefix scan and accepted when
// every prefix stays strictly inside the binade (then each partial sum is provably
// the reference's, see DESIGN.md "exact fast path"); the char that leaves the
// binade, binades where an addend falls exactly between two grid points (a
// round-half-even tie) and the first chars (|sum| comparable to the addends) run
// the plain chain of v_add_f64.
struct CandRec {                    // one re-scored candidate
    double score;                   // exact total + best delta
    int    offset;
    int    char_offset;             // first char with the best delta rank
    int    rank;                    // 0 = no substitution possible
    int    ch;                      // substituted character
};

struct RescoreArgs {
    const unsigned char* s1;
    const unsigned char* s2;        // byte codes
    const unsigned char* clsrank;   // [32][32]: class (bits 0-2) | delta rank (bits 3-7), [c2][c1]
    const unsigned char* sub;       // [32][32]
    const double*        delta_of_rank;
    const int*           cand;
    const double*        cand_upper;  // optimistic bounds of the candidates, or NULL (already filtered)
    unsigned long long*  global_best; // final best pessimistic bound (ordered-u64)
    // Single searches that end inside the count kernel keep the RUNNING bound in kBestSplit words (a tile uses word
    // tile % kBestSplit, kBestStride words apart: a cache line each): all 244 tiles of a 1M x 10k search reach their
    // atomicMax within 0.3 us of each other, and 244 atomics on one word were served one after the other (1.7 us for the
    // last arrivers against 0.3 uncontended).  A tile then knows the bounds of its own word's tiles only — a weaker
    // lower bound, a few more listed candidates, the same survivors (whoever re-scores re-checks against the final bound =
    // max of the words and of global_best).  NULL: one word (batches, searches a second launch finishes).
    unsigned long long*  best_split;
    int*                 cand_count;
    CandRec*             rec;       // [cap]
    int*                 done;      // arrival counter of the candidate workgroups
    SearchTail*          out;       // written by the last workgroup to arrive
    const int*           bad;       // the encode kernels' alphabet flag
    long long            pairs;
    int                  seq;       // sequence number of the search, echoed in the tail
    const unsigned long long* share_swept;   // shared combines: the slot's sweep counter (echoed in the tail), or NULL
    double class_value[8];          // pair score of class 0..3, +0.0 for the padding class 4
    int m;
    int cap;
    int is_max;
    int fast_ok;                    // all weights normal doubles or zero
    int fast_min_exp;               // smallest biased exponent of the running sum the walk handles
    int raw;                        // 1 (one-shot searches): s1 / s2 are the caller's raw ASCII (n1 / m bytes), not codes
    int n1;
    int certify;                    // 1: try the certified segment sum first (cert_sum), the walk is its fallback
    double walk_slack;              // rigorous bound of |sequential fp64 prefix sum - order-free prefix sum| over the m chars
    // blockIdx.y = query of a batch launch: its state lies query_stride... further on (all 0 for one search)
    size_t s2_stride;               // bytes between the queries' codes
    int    batched;                 // 1: cand/cand_upper/rec advance by cap, the scalars by kScalStride, out and seq by 1
    // single searches of a device group: the 32-byte record also goes to device memory (what the devices exchange)
    psa_result* record_out;         // or NULL
    int         record_shard;
};

#ifndef PSA_RESCORE_EPL
#define PSA_RESCORE_EPL 8
#endif
constexpr int kBestSplit = 8;           // words of the split running bound
constexpr int kBestStride = 16;         // u64 between them (128 bytes)
constexpr int kRescoreEpl = PSA_RESCORE_EPL;   // chars per lane and batch on the integer walk
constexpr int kRescoreThreads = 512;
constexpr int kRescoreChunk = 32768;    // chars whose class codes are staged in LDS at a time
constexpr int kRescoreGrid = 64;        // workgroups; each takes candidates ci = block, block + 64, ...
constexpr int kRescorePadLds = 8 * 1024;    // + ~68 KB static: two rescore workgroups per CU, never one beside a count workgroup

// Inclusive prefix sum of a 64-bit value over the 64 lanes, by DPP row shifts and row
// broadcasts (register-to-register; no LDS round trips as __shfl_up would need).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ long long dpp_add(long long v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return v + (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

__device__ __forceinline__ long long wave_scan_inclusive(long long v)
{
    v = dpp_add<0x111, 0xf>(v);   // row_shr:1
    v = dpp_add<0x112, 0xf>(v);   // row_shr:2
    v = dpp_add<0x114, 0xf>(v);   // row_shr:4
    v = dpp_add<0x118, 0xf>(v);   // row_shr:8   -> inclusive scan inside each row of 16 lanes
    v = dpp_add<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return v;
}

// Maximum of an unsigned 64-bit value over the 64 lanes, the same DPP steps as the scan (lanes without a source
// take 0, the identity); every lane gets the result.  (A __shfl_xor butterfly moves 64-bit values through the LDS
// crossbar: twelve ds_bpermute round trips instead of twelve DPP moves.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o > v ? o : v;
}

__device__ __forceinline__ long long readlane64(long long v, int l);

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
    v = dpp_max_u64<0x111, 0xf>(v);   // row_shr:1
    v = dpp_max_u64<0x112, 0xf>(v);   // row_shr:2
    v = dpp_max_u64<0x114, 0xf>(v);   // row_shr:4
    v = dpp_max_u64<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row's maximum
    v = dpp_max_u64<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_max_u64<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's maximum
    return (unsigned long long)readlane64((long long)v, 63);
}

__device__ __forceinline__ long long readlane64(long long v, int l)
{
    const unsigned lo = __builtin_amdgcn_readlane((int)v, l);
    const unsigned hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return (long long)(((unsigned long long)hi << 32) | lo);
}


// ---- building blocks shared by psa_rescore_kernel and the fused tail of psa_count_kernel ----
// Class | rank codes of chars [c0, c1) of one candidate (Seq1 window s1 = codes of Seq1 from the candidate's
// offset) into LDS, chunk-relative, four chars per thread and step; positions from c1 up to the walk's
// over-read are padding (class 4, addend +0.0).  `key` collects max(rank << 32 | ~i): best delta rank, then
// the lowest char index (cpu_funcs.c:287-294: strict compare keeps the first i).
// ASCII -> symbol code (psa_internal.hpp): 'A'..'Z' -> 0..25, '-' -> 26, anything else -> the padding symbol
__device__ __forceinline__ unsigned ascii_code(unsigned c)
{
    return c - 'A' < 26u ? c - 'A' : (c == (unsigned)'-' ? (unsigned)kCodeHyphen : (unsigned)kCodePad);
}

// Four consecutive symbols of a sequence as codes, one per byte.  Encoded sequences (RAW = false) are padded with the
// padding symbol far beyond their end; raw ASCII (a caller's buffer of `limit` bytes) is not read behind its end.
template <bool RAW>
__device__ __forceinline__ unsigned load_codes4(const unsigned char* __restrict__ seq, int i, int limit)
{
    unsigned w;
    if constexpr (!RAW) {
        (void)limit;
        __builtin_memcpy(&w, seq + i, 4);
        return w;
    } else {
        if (i + 4 <= limit) __builtin_memcpy(&w, seq + i, 4);
        else {
            w = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) w |= (i + j < limit ? (unsigned)seq[i + j] : 0u) << (8 * j);
        }
        unsigned codes = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) codes |= ascii_code((w >> (8 * j)) & 0xffu) << (8 * j);
        return codes;
    }
}

template <bool RAW = false>
__device__ __forceinline__ void classify_chars(const unsigned char* __restrict__ s1, const unsigned char* __restrict__ s2,
                                               const unsigned char* s_clsrank, unsigned char* s_codes, int c0, int c1,
                                               int tid, int nthreads, unsigned long long& key, int limit1 = 0, int limit2 = 0)
{
    const int groups = (c1 - c0 + 64 * kRescoreEpl + 3) >> 2;
#pragma unroll 4
    for (int g = tid; g < groups; g += nthreads) {
        const int i = c0 + 4 * g;                         // c0 and the Seq2 code buffer are 4-byte aligned
        const unsigned w2 = load_codes4<RAW>(s2, i, limit2);             // encoded buffers are padded with
        const unsigned w1 = load_codes4<RAW>(s1, i, limit1);             // kCodePad (class 4, rank 0)
        unsigned packed = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned cr = 4u;
            if (i + j < c1) {
                cr = s_clsrank[((w2 >> (8 * j)) & 0xffu) * kRow + ((w1 >> (8 * j)) & 0xffu)];
                const unsigned long long k2 = ((unsigned long long)(cr >> 3) << 32) | (0xffffffffu - (unsigned)(i + j));
                key = k2 > key ? k2 : key;
            }
            packed |= cr << (8 * j);
        }
        *reinterpret_cast<unsigned*>(s_codes + 4 * g) = packed;
    }
}

// s + v(lane 0) + v(lane 1) + ... + v(lane 63), added in that order (cpu_funcs.c:278): every lane's addend reaches the
// adder as a scalar operand (v_readlane), so the chain is 64 dependent v_add_f64 and nothing else — read back from LDS
// as broadcast ds_read_b128 the adds waited for the LDS every other step (1.24 us per 64 chars instead of ~0.4).
__device__ __forceinline__ double chain64(double s, double mine)
{
    const long long bits = __double_as_longlong(mine);
#pragma unroll
    for (int l = 0; l < 64; ++l) s += __longlong_as_double(readlane64(bits, l));
    return s;
}

struct WalkScratch {                // per walking wave, in LDS
    long long round[8];             // the integer walk's R_e[class]
};
struct WalkStats { int chain, batches, left; };     // how a walk went (phase-timing builds print them)

// The reference's sum over chars [c0, c1) continued from `s` (cpu_funcs.c:278), by ONE wave (all 64 lanes call
// this with the same arguments); s_codes = the chars' class codes from classify_chars (index k - c0), s_class =
// the class values.  Returns the new running sum, bit for bit what the m sequential fp64 additions give.
__device__ __forceinline__ double walk_sum(double s, const unsigned char* s_codes, int c0, int c1, const double* s_class,
                                           WalkScratch* ws, int fast_ok, int fast_min_exp, int lane, WalkStats& st)
{
    int k = c0;
    while (k < c1) {
        const unsigned long long sb = (unsigned long long)__double_as_longlong(s);
        const int e = (int)((sb >> 52) & 0x7ff);
        bool walk = fast_ok && e >= fast_min_exp && e < 0x7ff;
        if (walk) {
            // R_e[c] for the five classes, and whether any addend ties in this binade
            long long r = 0;
            bool tie = false;
            if (lane < 5) {
                const unsigned long long pb = (unsigned long long)__double_as_longlong(s_class[lane]);
                const int ep = (int)((pb >> 52) & 0x7ff);
                if (ep != 0) {                           // zero stays 0 (subnormals excluded by fast_ok)
                    const unsigned long long mp = (pb & 0xfffffffffffffull) | (1ull << 52);
                    const int d = e - ep;                // >= 2 by fast_min_exp
                    if (d <= 53) {
                        const unsigned long long half = 1ull << (d - 1);
                        tie = (mp & ((half << 1) - 1)) == half;
                        const long long mag = (long long)((mp + half) >> d);
                        r = (pb >> 63) ? -mag : mag;
                    }
                }
                ws->round[lane] = r;
            }
            if (__ballot(tie) != 0ull) walk = false;
        }
        if (!walk) {
            ++st.chain;
            // plain chain over the next <= 64 chars; lanes past the chunk hold +0.0 (exact to add)
            const int nb = c1 - k < 64 ? c1 - k : 64;
            s = chain64(s, lane < nb ? s_class[s_codes[k - c0 + lane] & 7u] : 0.0);   // reference order (cpu_funcs.c:278)
            k += nb;
            continue;
        }
        // integer walk inside binade e
        __builtin_amdgcn_wave_barrier();
        long long S = (long long)((sb & 0xfffffffffffffull) | (1ull << 52));
        const bool neg = (sb >> 63) != 0;
        if (neg) S = -S;
        // |q| strictly inside (2^52, 2^53)  <=>  q strictly inside (bound_lo, bound_hi)
        const long long bound_lo = neg ? -(1ll << 53) : (1ll << 52);
        const long long bound_hi = neg ? -(1ll << 52) : (1ll << 53);
        static_assert(kRescoreEpl % 8 == 0 && kRescoreEpl <= 16, "a lane's codes of one batch are 8-byte LDS words");
        constexpr int kWords = kRescoreEpl / 8;
        bool left = false;
        while (k < c1 && !left) {
            // The batch starts at the kRescoreEpl-aligned position at or below k; chars below k are already
            // in S and count as padding (class 4, addend 0), like the staged positions past c1.
            const int k_al = k & ~(kRescoreEpl - 1);
            unsigned long long w[kWords];
#pragma unroll
            for (int x = 0; x < kWords; ++x)
                w[x] = *reinterpret_cast<const unsigned long long*>(s_codes + (k_al - c0) + lane * kRescoreEpl + 8 * x);
            if (lane == 0 && k != k_al) {
#pragma unroll
                for (int x = 0; x < kWords; ++x) {
                    const int skip = k - k_al - 8 * x;                  // chars of this word already summed
                    if (skip >= 8) w[x] = 0x0404040404040404ull;
                    else if (skip > 0) {
                        const unsigned long long keep = ~0ull << (8 * skip);
                        w[x] = (w[x] & keep) | (0x0404040404040404ull & ~keep);
                    }
                }
            }
            long long rr[kRescoreEpl];
            long long t = 0;
#pragma unroll
            for (int j = 0; j < kRescoreEpl; ++j) {
                rr[j] = ws->round[(unsigned)(w[j >> 3] >> (8 * (j & 7))) & 7u];
                t += rr[j];
            }
            ++st.batches;
            const long long incl = wave_scan_inclusive(t);
            const long long q0 = S + (incl - t);          // sum before this lane's first char
            long long q = q0;
            bool inside = true;
#pragma unroll
            for (int j = 0; j < kRescoreEpl; ++j) {
                q += rr[j];
                inside &= (q > bound_lo) & (q < bound_hi);
            }
            const unsigned long long bad_lanes = __ballot(!inside);
            if (bad_lanes == 0ull) {
                S = readlane64(q, 63);
                k = k_al + 64 * kRescoreEpl < c1 ? k_al + 64 * kRescoreEpl : c1;
            } else {
                // rare (once per binade change): find the first char that leaves the binade
                int bad = kRescoreEpl;
                long long prev = q0, run = q0;
#pragma unroll
                for (int j = 0; j < kRescoreEpl; ++j) {
                    const long long nxt = run + rr[j];
                    if (!((nxt > bound_lo) & (nxt < bound_hi)) && bad == kRescoreEpl) { bad = j; prev = run; }
                    run = nxt;
                }
                const int L = __ffsll((long long)bad_lanes) - 1;
                const int j = __builtin_amdgcn_readlane(bad, L);
                S = readlane64(prev, L);                  // the sum before the char that leaves the binade
                const int p = k_al + L * kRescoreEpl + j;
                k = p > k ? p : k;                        // (a skipped char can only "fail" with S on the binade's edge)
                left = true;
            }
        }
        const unsigned long long mag = (unsigned long long)(S < 0 ? -S : S);
        s = __longlong_as_double((long long)((neg ? (1ull << 63) : 0ull) | ((unsigned long long)e << 52) |
                                             (mag & 0xfffffffffffffull)));
        if (left) {
            ++st.left;
            // this char moves the sum out of the binade: one true fp64 add
            s += s_class[s_codes[k - c0] & 7u];
            k += 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return s;
}

// ---- the same sum without walking: certified segments ---------------------------------
// walk_sum is a chain of ~30 dependent wave steps (14 us for 10,000 chars).  Most of it can be known without walking.
// Let P_k be the exact (real-number) prefix sum after char k and s_k the reference's sequentially rounded one:
// |s_k - P_k| <= slack for every k (the standard bound (k-1) u sum|p_i|; the host passes it doubled).  A char k is
// INTERIOR when an order-free evaluation of P_{k-1} and P_k puts both at least `slack` inside the same binade
// (2^e, 2^(e+1)) of the same sign, e large enough for the integer walk: then s_{k-1} and s_k certainly lie in that
// binade, and the reference's addition fl(s_{k-1} + p_k) is the integer step S += R_e[class] (both multiples of
// ulp_e, no tie — walk_sum's argument).  Every other char (a binade change, a prefix sum within slack of a power of
// two, the first chars near zero) is a TRUE fp64 addition.
constexpr int kCertChain = 64;            // chars summed as a plain chain first (the sum is still near zero there)
constexpr int kCertSeg = 16;              // chars per segment: one thread settles a segment, or one 16-lane row its chars

// scans inside the 16-lane rows of a wave (DPP row shifts; lanes without a source add nothing)
template <int CTRL>
__device__ __forceinline__ double row_add_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, false);
    return v + __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));   // (+0.0 where no source)
}
__device__ __forceinline__ double row_scan_inclusive_f64(double v)
{
    v = row_add_f64<0x111>(v);
    v = row_add_f64<0x112>(v);
    v = row_add_f64<0x114>(v);
    v = row_add_f64<0x118>(v);
    return v;
}
// |x| at least `margin` inside its binade (2^e, 2^(e+1)), e in the integer walk's range?  With lo = 2^e:
// min(|x| - lo, lo - (|x| - lo)) >= margin (both differences are exact).  hi32 = sign, exponent and upper mantissa.
__device__ __forceinline__ bool cert_inside(double x, double margin, double amin, double amax, unsigned& hi32)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    hi32 = (unsigned)(b >> 32);
    const double ax = __longlong_as_double((long long)(b & 0x7fffffffffffffffull));
    const double lo = __longlong_as_double((long long)(b & 0x7ff0000000000000ull));
    const double d = ax - lo;
    return fmin(d, lo - d) >= margin && ax >= amin && ax < amax;
}

// order-free prefix sum from class counts (16 bits per class)
__device__ __forceinline__ double cert_value(unsigned long long counts, const double* s_class)
{
    double p = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) p += s_class[c] * (double)(unsigned)((counts >> (16 * c)) & 0xffffull);
    return p;
}

// =====================================================================================================================
// The certificate and the certified sum (round 4 form; round 3's kept an unordered list of at most 64 true additions,
// which one wave sorted, and rebuilt the sum's mantissa per event):
//   * inputs whose running sum hovers near zero or near a power of two (low-entropy alphabets: ACGT with the README
//     weights has 154 true additions behind the first 64 chars) used to exceed the list and were WALKED (64-130 us);
//     now up to kCertMaxEv true additions are applied, 64 at a time;
//   * a shorter critical path in the search's tail: classification and the certificate's first level are ONE pass over
//     a thread's 16 chars, the head chain runs on a spare wave beside level 2, an event is two v_add_f64.
// Data of one candidate (LDS, behind its class codes; CertLayout):
//   seg_before[seg]  class counts (16 bits per class) of the chars in front of 16-char segment `seg`
//   bitmap           bit k set = char k is a TRUE fp64 addition (k >= kCertChain); level 2 sets the bits, so the true
//                    additions come out in ascending order by themselves (the first form sorted an unordered list)
//   fail[]           segments level 1 could not settle (any order)
//   ev[]             positions of the true additions, ascending (compacted from the bitmap by the summing wave)
// and per search the table rt[e - emin][class] = round(p_class / ulp_e) with a "tie" bit per binade, filled by spare
// threads while the others classify (the first form recomputed it per event with 64-bit shifts).
// An EVENT = the interior run in front of a true addition + that addition.  The run's sum is an integer number of
// ulp_e, D = sum_c cnt_c * rt[e][c] (exact, wrapping 64-bit; the certified total fits 53 bits); state and D are both
// multiples of ulp_e and the certificate says state + D stays inside binade e, so ONE v_add_f64 adds the run EXACTLY
// (no rounding can occur), a second one is the reference's own addition of the true char: two dependent adds and a
// check of the exponents per event instead of the first form's mantissa rebuild (87 -> ~30 ns per event).
// Nothing is assumed: a state or a state + D whose sign/exponent is not the certified one, a binade with a tie or
// outside the table, more than kCertMaxEv true additions or kCertMaxFail unsettled segments send the candidate to
// walk_sum.  Tested bit for bit against the oracle for every offset (test_rescore_walk_is_bit_exact, the soak).
constexpr int kCertMaxFail = 1024;       // unsettled 16-char segments per candidate
constexpr int kCertMaxEv = 1024;         // true additions per candidate the summing wave applies (64 at a time)
constexpr int kCertRows = 48;            // binades of the rounded-class-value table

struct CertTable {                       // per search
    long long rt[kCertRows][4];
    unsigned tie[kCertRows];             // != 0: an addend ties in that binade (round-half-even: the walk's business); a byte per class
    int emin, pad;
};
struct CertHead {                        // per candidate
    unsigned long long total;             // class counts of all m chars
    double head_sum;                      // the reference's sum over the first kCertChain chars
    int nfail;
    int pad;
};
constexpr int kCertSummary = 64;           // words of the bitmap's summary (bit w: bitmap word w is not zero): 65,536 chars
struct CertLayout { int seg_before, segbits, bitmap, summary, fail, ev, bytes, nseg, nwords; };
__host__ __device__ inline CertLayout cert_layout(int m)
{
    CertLayout L;
    L.nseg = (m + kCertSeg - 1) / kCertSeg;
    L.nwords = (m + 31) / 32;
    int at = (int)sizeof(CertHead);
    L.seg_before = at;  at += 8 * (L.nseg + 1);
    L.segbits = at;     at += 4 * ((L.nseg + 1) & ~1);       // per segment: bit t = low class bit of char t, bit 16 + t = high
    L.bitmap = at;      at += 4 * ((L.nwords + 1) & ~1);
    L.summary = at;     at += 4 * kCertSummary;
    L.fail = at;        at += 2 * kCertMaxFail;             // (pass 4 reuses it for the list of non-zero bitmap words)
    L.ev = at;          at += 2 * kCertMaxEv;
    L.bytes = (at + 15) & ~15;
    return L;
}
static_assert(sizeof(CertHead) == 24, "cert_layout");

// the table (any kCertRows * 4 threads; `idx` = this thread's entry, `value` = that class's pair score)
__device__ __forceinline__ void cert_fill_table(CertTable* tb, double value, int fast_min_exp, int idx)
{
    const int row = idx >> 2, c = idx & 3;
    const int e = fast_min_exp + row;
    const unsigned long long pb = (unsigned long long)__double_as_longlong(value);
    const int ep = (int)((pb >> 52) & 0x7ff);
    long long r = 0;
    bool tie = false;
    if (ep != 0) {                                           // zero stays 0 (subnormal weights: no certificate at all)
        const unsigned long long mp = (pb & 0xfffffffffffffull) | (1ull << 52);
        const int d = e - ep;                                // >= 2: fast_min_exp is two binades above the largest addend
        if (d < 2 || e >= 0x7fe) tie = true;                 // (never a certified binade)
        else if (d <= 53) {
            const unsigned long long half = 1ull << (d - 1);
            tie = (mp & ((half << 1) - 1)) == half;
            const long long mag = (long long)((mp + half) >> d);
            r = (pb >> 63) ? -mag : mag;
        }
    }
    tb->rt[row][c] = r;
    reinterpret_cast<unsigned char*>(tb->tie)[idx] = tie ? 1 : 0;
    if (idx == 0) tb->emin = fast_min_exp;
}

// class counts (16 bits per class) of the first `r` chars of a segment from its class bits
__device__ __forceinline__ unsigned long long cert_counts_of_bits(unsigned bits, int r)
{
    const unsigned pm = (1u << r) - 1u;                      // r <= 16
    const unsigned b0 = bits & pm, b1 = (bits >> 16) & pm;
    const unsigned n1 = __popc(b0 & ~b1), n2 = __popc(b1 & ~b0), n3 = __popc(b0 & b1);
    const unsigned n0 = (unsigned)r - n1 - n2 - n3;
    return (unsigned long long)(n0 | (n1 << 16)) | ((unsigned long long)(n2 | (n3 << 16)) << 32);
}

// Returns the WAVE's class counts in front of this thread's first segment + the thread's own (an inclusive wave scan of
// the threads' counts; lane 63 holds the wave's total): seg_before[] is left relative to the wave's first segment, pass
// 2 adds the waves in front.  The code is kept SMALL (the four dwords of a segment in a rolled loop): the tail runs once
// per search on a workgroup whose instruction cache holds the count loop — every byte of it is a cold fetch.
template <int NT, bool RAW>
__device__ __forceinline__ unsigned long long cert_classify(const unsigned char* __restrict__ s1, const unsigned char* __restrict__ s2,
                                                             const unsigned char* s_clsrank, unsigned char* s_codes, int m,
                                                             unsigned long long* seg_before, unsigned* segbits, int tid,
                                                             unsigned long long& key, int limit1, int limit2)
{
    const int nseg = (m + kCertSeg - 1) / kCertSeg;
    const int nseg_pad = (m + 64 * kRescoreEpl + 16 + kCertSeg - 1) / kCertSeg;      // codes are written up to here
    const int per = (nseg + NT - 1) / NT;                    // 1 up to 16,384 chars with 1,024 threads
    const int per_pad = per + (nseg_pad - nseg + NT - 1) / NT;
    unsigned long long mine = 0ull;
#pragma unroll 1
    for (int q = 0; q < per_pad; ++q) {
        const int seg = q < per ? tid * per + q : nseg + (q - per) * NT + tid;      // real segments first, then the padding's
        const int k0 = seg * kCertSeg;
        if (q < per && seg < nseg) {
            unsigned w1[4], w2[4];
#ifdef PSA_PHASE_TIMING
            if (tid == 0) g_phase_ts[1005][0] = wall_clock64();
#endif
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                w2[x] = load_codes4<RAW>(s2, k0 + 4 * x, limit2);      // encoded buffers are padded with kCodePad (class 4,
                w1[x] = load_codes4<RAW>(s1, k0 + 4 * x, limit1);      // rank 0) far beyond their ends, raw ones read as such
            }
#ifdef PSA_PHASE_TIMING
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(w1[0]), "+v"(w1[1]), "+v"(w1[2]), "+v"(w1[3]), "+v"(w2[0]), "+v"(w2[1]), "+v"(w2[2]), "+v"(w2[3]));
            if (tid == 0) g_phase_ts[1005][1] = wall_clock64();
#endif
            // per char: one table byte (class | rank << 3; chars behind m come out as class 4, rank 0 by themselves), its
            // place in the packed codes and the two class bits; the best (rank, lowest char) of the segment is taken from
            // the packed codes four chars at a time (bytes rank << 4 | 15 - t, maxima as 16-bit pairs) — 32-bit ops only
            unsigned packed[4], b0 = 0u, b1 = 0u;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                packed[x] = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int t = 4 * x + j;
                    const unsigned cr = s_clsrank[((w2[x] >> (8 * j)) & 0xffu) * kRow + ((w1[x] >> (8 * j)) & 0xffu)];
                    packed[x] |= cr << (8 * j);
                    b0 |= (cr & 1u) << t;
                    b1 |= ((cr >> 1) & 1u) << t;
                }
            }
            *reinterpret_cast<uint4*>(s_codes + k0) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
            unsigned best;
            {
                // byte t of v[x]: rank << 4 | 15 - t (ranks are 4 bits: kMaxRanks = 16)
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                auto pk_max = [](unsigned a, unsigned b) {
                    const u16x2 r = __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
                    return __builtin_bit_cast(unsigned, r);
                };
                unsigned mx = 0u;
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const unsigned v = ((packed[x] << 1) & 0xf0f0f0f0u) | (0x0c0d0e0fu - 0x04040404u * (unsigned)x);
                    mx = pk_max(mx, pk_max(v & 0x00ff00ffu, (v >> 8) & 0x00ff00ffu));
                }
                best = (mx & 0xffffu) > (mx >> 16) ? (mx & 0xffffu) : (mx >> 16);
            }
#ifdef PSA_PHASE_TIMING
            if (tid == 0) g_phase_ts[1005][2] = wall_clock64();
#endif
            // (cpu_funcs.c:287-294: strictly better delta only, so the first char of the best rank stands)
            const unsigned long long k2 = ((unsigned long long)(best >> 4) << 32) | (0xffffffffu - (unsigned)(k0 + 15 - (int)(best & 15u)));
            key = k2 > key ? k2 : key;
            if (seg_before) {
                const int valid = m - k0 < kCertSeg ? m - k0 : kCertSeg;
                const unsigned bits = b0 | (b1 << 16);
                segbits[seg] = bits;
                if (per > 1) seg_before[seg] = mine;         // (relative to the thread's first segment: see below)
                mine += cert_counts_of_bits(bits, valid);
            }
        } else if (q >= per && seg < nseg_pad) {
            *reinterpret_cast<uint4*>(s_codes + k0) = make_uint4(0x04040404u, 0x04040404u, 0x04040404u, 0x04040404u);
        }
    }
    if (!seg_before) return 0ull;
    const unsigned long long incl = (unsigned long long)wave_scan_inclusive((long long)mine);
    const unsigned long long excl = incl - mine;
    // seg_before relative to the wave's first segment: one store for a thread's one segment (Seq2 up to 16,384 chars with
    // 1,024 threads), otherwise the thread's segments again (its own stores, no barrier needed)
    if (per == 1) {
        if (tid < nseg) seg_before[tid] = excl;
    } else if (excl != 0ull) {
#pragma unroll 1
        for (int q = 0; q < per; ++q) {
            const int seg = tid * per + q;
            if (seg < nseg) seg_before[seg] += excl;
        }
    }
    return incl;
}

// ---- pass 2 (every thread, behind the barrier that follows the block scan's wave totals) ----------------------------
// `before` = class counts of the WAVES in front of this thread's (their totals: lane 63 of cert_classify's result).
// Level 1: all prefix sums of a segment lie within A = sum |p_i| of the sum in front of it; that sum A + slack inside its
// binade settles the segment.  The others go to fail[].  head->nfail must be zero (and the bitmap clear) by now.
template <int NT>
__device__ __forceinline__ void cert_level1(const unsigned char* s_codes, int m, const double* s_class, double slack,
                                             int fast_min_exp, unsigned char* cert, unsigned long long before, int tid)
{
    const CertLayout L = cert_layout(m);
    CertHead* head = reinterpret_cast<CertHead*>(cert);
    unsigned long long* seg_before = reinterpret_cast<unsigned long long*>(cert + L.seg_before);
    const unsigned* segbits = reinterpret_cast<const unsigned*>(cert + L.segbits);
    unsigned short* fail = reinterpret_cast<unsigned short*>(cert + L.fail);
    (void)s_codes;
    const int per = (L.nseg + NT - 1) / NT;
    const double amin = __longlong_as_double((long long)((unsigned long long)(fast_min_exp < 1 ? 1 : fast_min_exp) << 52));
    const double amax = __longlong_as_double(0x7fd0000000000000ll);
    for (int q = 0; q < per; ++q) {
        const int seg = tid * per + q;
        if (seg >= L.nseg) break;
        const int k0 = seg * kCertSeg;
        const unsigned long long b = before + seg_before[seg];            // (relative to the thread's first segment so far)
        seg_before[seg] = b;
        if (k0 + kCertSeg > kCertChain) {
            // the segment's own counts: from its class bits
            const unsigned long long n = cert_counts_of_bits(segbits[seg], m - k0 < kCertSeg ? m - k0 : kCertSeg);
            double reach = 0.0;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) reach += fabs(s_class[cc]) * (double)(unsigned)((n >> (16 * cc)) & 0xffffull);
            unsigned hi32;
            if (!cert_inside(cert_value(b, s_class), reach * 1.0000001 + slack, amin, amax, hi32)) {
                const int slot = atomicAdd(&head->nfail, 1);
                if (slot < kCertMaxFail) fail[slot] = (unsigned short)seg;
            }
        }
    }
}

// ---- pass 3 (every thread, behind a barrier): level 2, a 16-lane row per unsettled segment, a lane per char ----------
template <int NT>
__device__ __forceinline__ void cert_level2(const unsigned char* s_codes, int m, const double* s_class, double slack,
                                             int fast_min_exp, unsigned char* cert, int tid)
{
    const CertLayout L = cert_layout(m);
    const CertHead* head = reinterpret_cast<const CertHead*>(cert);
    const unsigned long long* seg_before = reinterpret_cast<const unsigned long long*>(cert + L.seg_before);
    const unsigned short* fail = reinterpret_cast<const unsigned short*>(cert + L.fail);
    unsigned* bitmap = reinterpret_cast<unsigned*>(cert + L.bitmap);
    unsigned* summary = reinterpret_cast<unsigned*>(cert + L.summary);
    const int lane = tid & 63, wave = tid >> 6;
    const int nfail = head->nfail;
    if (nfail > kCertMaxFail) return;
    const double amin = __longlong_as_double((long long)((unsigned long long)(fast_min_exp < 1 ? 1 : fast_min_exp) << 52));
    const double amax = __longlong_as_double(0x7fd0000000000000ll);
    const int row = lane >> 4, col = lane & 15;
    for (int f0 = wave * 4; f0 < nfail; f0 += (NT / 64) * 4) {
        const int f = f0 + row;
        const bool live = f < nfail;
        const int seg = live ? (int)fail[f] : 0;
        const int k = seg * kCertSeg + col;
        const unsigned long long cb = live ? seg_before[seg] : 0ull;
        const unsigned cls = s_codes[k] & 7u;
        const double v = s_class[cls];                                           // (+0.0 behind m)
        const double p_front = cert_value(cb, s_class);
        const double p = p_front + row_scan_inclusive_f64(v);                    // after this lane's char
        // the sum in front of this lane's char: the left neighbour's, the segment's own for the first lane
        const long long pb = __double_as_longlong(p);
        const int nlo = __builtin_amdgcn_update_dpp(0, (int)pb, 0x111, 0xf, 0xf, false);
        const int nhi = __builtin_amdgcn_update_dpp(0, (int)(pb >> 32), 0x111, 0xf, 0xf, false);
        const double p_prev = col == 0 ? p_front
                                       : __longlong_as_double((long long)(((unsigned long long)(unsigned)nhi << 32) | (unsigned)nlo));
        unsigned hi, hi_prev;
        const bool safe = cert_inside(p, slack, amin, amax, hi);
        const bool safe_prev = cert_inside(p_prev, slack, amin, amax, hi_prev);
        const bool interior = safe && safe_prev && ((hi ^ hi_prev) >> 20) == 0u;   // same sign and exponent
        const bool is_true = live && !interior && k >= kCertChain && k < m;
        const unsigned long long mask = __ballot(is_true);
        const unsigned bits = (unsigned)(mask >> (16 * row)) & 0xffffu;
        if (col == 0 && bits) {
            atomicOr(&bitmap[seg >> 1], bits << (16 * (seg & 1)));
            atomicOr(&summary[seg >> 6], 1u << ((seg >> 1) & 31));
        }
    }
}

// class counts of chars [0, x), 0 <= x <= m: the segment's prefix + the chars of its segment in front of x
__device__ __forceinline__ unsigned long long cert_counts_before(const unsigned* segbits, const unsigned long long* seg_before,
                                                                  int x, int m, unsigned long long total)
{
    if (x >= m) return total;
    const int seg = x >> 4;
    return seg_before[seg] + cert_counts_of_bits(segbits[seg], x & 15);
}

// ---- the sum: ONE wave (all 64 lanes, same arguments), behind the barrier that follows level 2 ------------------------
// Returns false when the candidate has to be walked instead.
__device__ __forceinline__ bool cert_sum(const unsigned char* s_codes, int m, const double* s_class, unsigned char* cert,
                                          const CertTable* tb, int lane, double& s_out)
{
#ifdef PSA_PHASE_TIMING
#define PSA_SPHASE(k) do { if (lane == 0) g_phase_ts[1004][k] = wall_clock64(); } while (0)
#else
#define PSA_SPHASE(k) do { } while (0)
#endif
    const CertLayout L = cert_layout(m);
    const CertHead* head = reinterpret_cast<const CertHead*>(cert);
    const unsigned long long* seg_before = reinterpret_cast<const unsigned long long*>(cert + L.seg_before);
    const unsigned* segbits = reinterpret_cast<const unsigned*>(cert + L.segbits);
    const unsigned* bitmap = reinterpret_cast<const unsigned*>(cert + L.bitmap);
    const unsigned* summary = reinterpret_cast<const unsigned*>(cert + L.summary);
    unsigned short* nzw = reinterpret_cast<unsigned short*>(cert + L.fail);     // (the fail list is dead behind level 2)
    unsigned short* ev = reinterpret_cast<unsigned short*>(cert + L.ev);
    if (head->nfail > kCertMaxFail) return false;
    PSA_SPHASE(0);
    // ---- the true additions in ascending order, compacted from the bitmap in two steps: the summary (a lane per 32
    // bitmap words) gives the non-zero words in order, then a lane per non-zero word gives the positions
    int nzw_count;
    {
        const unsigned sw = lane < (L.nwords + 31) / 32 ? summary[lane] : 0u;
        const int mine = __popc(sw);
        const int incl = (int)wave_scan_inclusive((long long)mine);
        nzw_count = __builtin_amdgcn_readlane(incl, 63);
        if (nzw_count > kCertMaxFail) return false;
        int at = incl - mine;
        unsigned w = sw;
        while (w) {
            nzw[at++] = (unsigned short)(32 * lane + __ffs((int)w) - 1);
            w &= w - 1u;
        }
    }
    __builtin_amdgcn_wave_barrier();
    int nev = 0;
    for (int c0 = 0; c0 < nzw_count; c0 += 64) {
        const int wi = c0 + lane < nzw_count ? (int)nzw[c0 + lane] : -1;
        unsigned w = wi >= 0 ? bitmap[wi] : 0u;
        const int mine = __popc(w);
        const int incl = (int)wave_scan_inclusive((long long)mine);
        int at = nev + incl - mine;
        nev += __builtin_amdgcn_readlane(incl, 63);
        if (nev > kCertMaxEv) return false;
        while (w) {
            ev[at++] = (unsigned short)(32 * wi + __ffs((int)w) - 1);
            w &= w - 1u;
        }
    }
    __builtin_amdgcn_wave_barrier();
    PSA_SPHASE(1);
    const int head_n = m < kCertChain ? m : kCertChain;
    const unsigned long long total = head->total;
    const unsigned long long at_head = cert_counts_before(segbits, seg_before, head_n, m, total);
    double s = head->head_sum;                            // the first chars: a plain chain, summed beside level 2
    PSA_SPHASE(2);
    // ---- events, 64 at a time: a lane prepares event i = the interior run behind true addition i - 1 (or the head) and
    // true addition i; the last event (i == nev) is the run up to m alone.  Then the wave applies them in order.
    for (int i0 = 0; i0 <= nev; i0 += 64) {
        const int i = i0 + lane;
        double dv = 0.0, v = 0.0;
        unsigned keyhi = 0u, kmask = 0u;
        bool bad = false;
        if (i <= nev) {
            int at = m, from = head_n;
            unsigned long long ca = at_head;
            if (i < nev) {
                at = (int)ev[i];
                v = s_class[s_codes[at] & 7u];
            }
            if (i > 0) {
                from = (int)ev[i - 1] + 1;
                ca = cert_counts_before(segbits, seg_before, from, m, total);
            }
            if (at > from) {
                const unsigned long long cb = cert_counts_before(segbits, seg_before, at, m, total);
                const unsigned long long cnt = cb - ca;                    // per class, no borrows: counts only grow
                // the run's binade: the certified one of the sum in front of it
                const unsigned key = (unsigned)((unsigned long long)__double_as_longlong(cert_value(ca, s_class)) >> 52);
                const int e = (int)(key & 0x7ffu);
                const int row = e - tb->emin;
                if (row < 0 || row >= kCertRows || tb->tie[row] != 0u) bad = true;
                else {
                    unsigned long long delta = 0ull;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        delta += ((cnt >> (16 * c)) & 0xffffull) * (unsigned long long)tb->rt[row][c];   // (wraps; the certified total fits)
                    dv = ldexp((double)(long long)delta, e - 1075);        // D = delta ulp_e, exact: |delta| < 2^53
                }
                keyhi = key << 20;
                kmask = 0xfff00000u;
            }
        }
        if (__ballot(bad) != 0ull) return false;
        PSA_SPHASE(3);
        const int count = nev + 1 - i0 < 64 ? nev + 1 - i0 : 64;
        // Applied in order WITHOUT broadcasting the events: lane j keeps event j, and the running sum travels from lane
        // to lane.  Every lane executes every step — "take the left neighbour's state (wave_ror:1), add my run, add my
        // char" — but only lane j's step j is real: it reads lane j-1's state of step j-1 (by induction the true sum
        // behind event j-1; step 0: lane 0 reads lane 63, which like every lane starts from the chunk's initial sum).
        // Whatever the other lanes compute in that step is never read at the moment it would matter.  Per event: two DPP
        // moves, v_add_f64 (the run: exact), v_add_f64 (the true addition), the exponent check kept for lane j at step j —
        // 9 instructions and no v_readlane (the broadcast form took 12.5 and 46 ns per event on the lone wave).
        unsigned s_lo = (unsigned)(unsigned long long)__double_as_longlong(s);
        unsigned s_hi = (unsigned)((unsigned long long)__double_as_longlong(s) >> 32);
        unsigned chk_mine = 0u;
        auto step = [&](int j) {
            const unsigned t_lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s_lo, 0x13c, 0xf, 0xf, false);   // wave_ror:1:
            const unsigned t_hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s_hi, 0x13c, 0xf, 0xf, false);   // lane l <- lane l-1
            const double s0 = __longlong_as_double((long long)(((unsigned long long)t_hi << 32) | t_lo));
            const double s1 = s0 + dv;                     // the interior run: exact (both multiples of ulp_e, inside binade e)
            const unsigned h1 = (unsigned)((unsigned long long)__double_as_longlong(s1) >> 32);
            const unsigned term = ((t_hi ^ keyhi) | (h1 ^ keyhi)) & kmask;      // state and state + run: the certified sign and exponent
            chk_mine = lane == j ? term : chk_mine;
            const double x = s1 + v;                       // the reference's own addition (+0.0 behind the last one)
            s_lo = (unsigned)(unsigned long long)__double_as_longlong(x);
            s_hi = (unsigned)((unsigned long long)__double_as_longlong(x) >> 32);
        };
        int j = 0;
        for (; j + 2 <= count; j += 2) {
            step(j);
            step(j + 1);
        }
        if (j < count) step(j);
        s = __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)s_hi, count - 1) << 32) |
                                             (unsigned)__builtin_amdgcn_readlane((int)s_lo, count - 1)));
        if (__ballot(lane < count && chk_mine != 0u) != 0ull) return false;
    }
    PSA_SPHASE(4);
    s_out = s;
    return true;
}

// what the host reads after a search (pinned, host-coherent memory written over PCIe by one lane)
// `bad`: the alphabet flag when the caller has it already (>= 0), else it is loaded here
__device__ __forceinline__ void write_search_tail(const RescoreArgs& a, const CandRec* best, int status, int bad = -1, int swept = 0)
{
    const bool is_max = a.is_max != 0;
    psa_result r;
    if (best) {
        r.score = best->score;
        r.offset = best->offset;
        r.char_offset = best->char_offset;
        r.ch = best->ch;
    } else {
        r.score = status ? 0.0 : (is_max ? -INFINITY : INFINITY);
        r.offset = -1;
        r.char_offset = -1;
        r.ch = 0;
    }
    r.shard = -1;
    r.pairs = a.pairs;
    if (a.record_out) {
        // the record a device group exchanges: shard = the shard's index, or -2 - status when the host has to look
        // at this search again (candidate list overflow, separate re-score launch)
        psa_result x = r;
        x.shard = status ? -2 - status : a.record_shard;
        *a.record_out = x;
    }
    publish_tail(a.out, r, bad >= 0 ? bad : load_agent(a.bad), status, a.seq, swept);
}

// ---- end of a single search inside the count kernel ---------------------------------
// Run by the last tile workgroup to arrive (all of its waves).  The candidate list is complete and the bound in
// global_best final: candidates it rules out are dropped, the survivors (a handful) are re-scored in the reference's
// summation order and folded in the is_swapable order, the tail is written.  Up to kFusedRound survivors at a time:
//   pass 1  every thread classifies its 16-char segments of every survivor (codes into LDS, best delta rank and its first
//           char, the segments' class counts), wave scans of the counts; spare threads fill the rounded-class-value table
//   pass 2  block prefix of the counts, the certificate's first level (a thread per segment)
//   pass 3  the certificate's second level (a 16-lane row per unsettled segment); the head chain of survivor j on wave
//           WAVES - 1 - j beside it
//   pass 4  one wave per survivor: true additions compacted, events prepared and applied (cert_sum; walk_sum if that
//           declines)
// four workgroup barriers per round.  More survivors than kFusedMax, or a Seq2 whose codes do not fit the LDS: status 2,
// the host launches psa_rescore_kernel on the untouched lists.
constexpr int kFusedMax = 16;
constexpr int kFusedWaves = 16;
constexpr int kFusedRound = 4;            // survivors of one round (their per-thread state is kept in registers)
// LDS in front of the candidates' class codes (finish_search's carve-up) and how many candidates' regions fit at a time
constexpr int kFusedFixedBytes = (2 * kRow * kRow + (8 + kMaxRanks) * (int)sizeof(double) +
                                  kFusedWaves * (int)sizeof(WalkScratch) + 2 * kFusedRound * kFusedWaves * 8 +
                                  kFusedMax * (int)sizeof(CandRec) + kFusedMax * 4 + 16 + (int)sizeof(CertTable) + 15) & ~15;
// LDS of one candidate: its class codes + the walk's over-read, then the certificate's data (Seq2 below 65,536 chars)
__host__ __device__ inline int fused_codes_bytes(int m) { return (m + 64 * kRescoreEpl + 16 + 15) & ~15; }
__host__ __device__ inline int fused_region_bytes(int m) { return fused_codes_bytes(m) + (m < 65536 ? cert_layout(m).bytes : 0); }
__host__ __device__ inline int fused_candidates_at_a_time(int lds_bytes, int m)
{
    const int g = (lds_bytes - kFusedFixedBytes) / fused_region_bytes(m);
    return g > kFusedRound ? kFusedRound : g;
}

// where finish_search keeps its survivor counter (the caller zeroes it BEFORE the arrival's barrier, so the survivors can
// be collected without a barrier of their own)
constexpr int kFusedCounterOffset = 2 * kRow * kRow + (8 + kMaxRanks) * (int)sizeof(double) + kFusedWaves * (int)sizeof(WalkScratch) +
                                    2 * kFusedRound * kFusedWaves * 8 + kFusedMax * (int)sizeof(CandRec) + kFusedMax * 4;

template <int WAVES, bool RAW>
__device__ __forceinline__ void finish_search(const RescoreArgs& r, unsigned char* lds, int lds_bytes, int tid)
{
    static_assert(WAVES == kFusedWaves && kFusedRound <= WAVES && kMaxRanks == 16, "kFusedFixedBytes is this carve-up for 16 waves");
    static_assert(sizeof(WalkScratch) % 16 == 0 && sizeof(CandRec) == 24 && sizeof(CertTable) % 8 == 0, "LDS carve-up below");
    constexpr int T = WAVES * 64;
    static_assert(T >= 4 * kCertRows, "the table is filled by the workgroup's last threads");
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* p = lds;
    unsigned char* s_clsrank = p;                                             p += kRow * kRow;
    unsigned char* s_sub = p;                                                 p += kRow * kRow;
    double* s_class = reinterpret_cast<double*>(p);                           p += 8 * sizeof(double);
    double* s_delta = reinterpret_cast<double*>(p);                           p += kMaxRanks * sizeof(double);
    WalkScratch* s_walk = reinterpret_cast<WalkScratch*>(p);                  p += WAVES * sizeof(WalkScratch);
    unsigned long long* s_keys = reinterpret_cast<unsigned long long*>(p);    p += kFusedRound * WAVES * 8;   // [candidate of the round][wave]
    unsigned long long* s_wsum = reinterpret_cast<unsigned long long*>(p);    p += kFusedRound * WAVES * 8;   // the same shape: wave totals of the counts
    CandRec* s_rec = reinterpret_cast<CandRec*>(p);                           p += kFusedMax * sizeof(CandRec);
    int* s_surv = reinterpret_cast<int*>(p);                                  p += kFusedMax * 4;
    int* s_n = reinterpret_cast<int*>(p);                                     p += 16;   // (= lds + kFusedCounterOffset, zero on entry)
    CertTable* s_tab = reinterpret_cast<CertTable*>(p);
    unsigned char* s_codes = lds + kFusedFixedBytes;
    const int m = r.m;
    const bool certify = r.certify != 0;                                      // (host: only for m < 65,536)
    const int codes_bytes = fused_codes_bytes(m);
    const int region = fused_region_bytes(m);                                 // per candidate: codes, certificate
    const CertLayout L = cert_layout(m < 65536 ? m : 0);
    const int G = fused_candidates_at_a_time(lds_bytes, m);                   // candidates of a round

    PSA_FPHASE(0);
    // ---- everything the tail reads from memory before it knows the survivors, in flight together: the search scalars,
    // the alphabet flag (every tile and converter has arrived: it is final), the whole candidate list (entries behind
    // the count are never looked at) and the small tables
    // The search's scalars by ONE 8-byte load per wave, a lane each: lanes 0..7 the words of the running bound, lane 8
    // global_best, lane 9 the candidate count (and the arrival counter behind it), lanes 10.. the alphabet flag (every
    // tile and converter has arrived: it is final).  (Every lane loading every scalar was 13 requests per wave to the
    // same few lines: the 16 waves' 200 requests queued for 2 us.)
    static_assert(kBestSplit == 8, "lane map of the scalar load");
    const unsigned long long* scalar_at =
        lane < kBestSplit ? (r.best_split ? r.best_split + lane * kBestStride : r.global_best)
        : lane == 8       ? r.global_best
        : lane == 9       ? reinterpret_cast<const unsigned long long*>(r.cand_count)
        : lane == 11 && r.share_swept ? r.share_swept         // (the slot's sweep counter: the host watches how the sharing goes)
                          : reinterpret_cast<const unsigned long long*>(r.bad);
    const unsigned long long scalar = load_agent(scalar_at);
    double up[2];
    int off[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ci = tid + e * T;
        up[e] = ci < r.cap ? load_agent(r.cand_upper + ci) : 0.0;
        off[e] = ci < r.cap ? load_agent(r.cand + ci) : 0;
    }
    if (tid < kRow * kRow / 4) reinterpret_cast<unsigned*>(s_clsrank)[tid] = reinterpret_cast<const unsigned*>(r.clsrank)[tid];
    else if (tid < kRow * kRow / 2) reinterpret_cast<unsigned*>(s_sub)[tid - kRow * kRow / 4] = reinterpret_cast<const unsigned*>(r.sub)[tid - kRow * kRow / 4];
    else if (tid < kRow * kRow / 2 + kMaxRanks) s_delta[tid - kRow * kRow / 2] = r.delta_of_rank[tid - kRow * kRow / 2];
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_class[k] = r.class_value[k];
    }
    // (the rounded-class-value table, by the workgroup's last threads while the loads above are in flight; pass 4 reads it)
    if (certify && tid >= T - 4 * kCertRows) {
        const int idx = tid - (T - 4 * kCertRows), c = idx & 3;
        const double value = c == 0 ? r.class_value[0] : c == 1 ? r.class_value[1] : c == 2 ? r.class_value[2] : r.class_value[3];
        cert_fill_table(s_tab, value, r.fast_min_exp, idx);
    }
    const int count = __builtin_amdgcn_readlane((int)scalar, 9);
    const int bad_flag = __builtin_amdgcn_readlane((int)scalar, 10);
    const int swept = r.share_swept ? __builtin_amdgcn_readlane((int)scalar, 11) : 0;
    const unsigned long long best_key = wave_max_u64(lane <= kBestSplit ? scalar : 0ull);      // the final bound
    if (count > r.cap) {
        // overflow against the running bound: the bound stays in global_best, the host selects once more against it
        if (tid == 0) {
            write_search_tail(r, nullptr, 1, bad_flag);
            store_agent(r.cand_count, 0);
            store_agent(r.done, 0);
            __hip_atomic_store(r.global_best, best_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (every word of the split bound: the re-selection launch's tiles then select against the final bound at once)
        if (r.best_split && tid < kBestSplit) __hip_atomic_store(r.best_split + tid * kBestStride, best_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const double final_best = key_order(best_key);
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if (tid + e * T < count && up[e] >= final_best) {
            const int k = atomicAdd(&s_n[0], 1);
            if (k < kFusedMax) s_surv[k] = off[e];
        }
    __syncthreads();
    PSA_FPHASE(1);
    const int nsurv = s_n[0];
    if (nsurv > kFusedMax || G < 1 || r.cap > 2 * T) {
        if (tid == 0) {
            write_search_tail(r, nullptr, 2, bad_flag);
            store_agent(r.done, 0);                      // psa_rescore_kernel counts its own arrivals in it
            __hip_atomic_store(r.global_best, best_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ... and reads the final bound here
        }
        if (r.best_split && tid < kBestSplit) __hip_atomic_store(r.best_split + tid * kBestStride, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    for (int base = 0; base < nsurv; base += G) {
        const int cnt = nsurv - base < G ? nsurv - base : G;
        // ---- pass 1
#pragma unroll 1
        for (int j = 0; j < cnt; ++j) {
            unsigned char* codes = s_codes + (size_t)j * region;
            unsigned char* cert = codes + codes_bytes;
            const int o = s_surv[base + j];
            unsigned long long key = 0ull;
            const unsigned long long incl =
                cert_classify<T, RAW>(r.s1 + o, r.s2, s_clsrank, codes, m,
                                       certify ? reinterpret_cast<unsigned long long*>(cert + L.seg_before) : nullptr,
                                       reinterpret_cast<unsigned*>(cert + L.segbits), tid, key, r.n1 - o, m);
            if (certify) {
                // (level 2 sets the bits, behind two barriers) bitmap and summary lie side by side: one clearing loop
                unsigned* bitmap = reinterpret_cast<unsigned*>(cert + L.bitmap);
                const int words = (L.summary - L.bitmap) / 4 + kCertSummary;
                for (int w = tid; w < words; w += T) bitmap[w] = 0u;
                if (tid == 0) reinterpret_cast<CertHead*>(cert)->nfail = 0;
            }
#ifdef PSA_PHASE_TIMING
            if (tid == 0) g_phase_ts[1005][3] = wall_clock64();
#endif
            key = wave_max_u64(key);
            if (lane == 0) s_keys[j * WAVES + wave] = key;
            if (lane == 63) s_wsum[j * WAVES + wave] = incl;
#ifdef PSA_PHASE_TIMING
            if (tid == 0) g_phase_ts[1005][4] = wall_clock64();
            if (tid == T - 1) g_phase_ts[1005][5] = wall_clock64();
#endif
        }
        __syncthreads();
        PSA_FPHASE(2);
        // ---- pass 2
        if (certify) {
#pragma unroll 1
            for (int j = 0; j < cnt; ++j) {
                unsigned char* codes = s_codes + (size_t)j * region;
                unsigned char* cert = codes + codes_bytes;
                // the waves' totals: lane w holds wave w's; an exclusive scan gives the counts in front of every wave
                const unsigned long long tot = lane < WAVES ? s_wsum[j * WAVES + lane] : 0ull;
                const unsigned long long scan = (unsigned long long)wave_scan_inclusive((long long)tot);
                const unsigned long long before = (unsigned long long)readlane64((long long)(scan - tot), wave);
                const unsigned long long all = (unsigned long long)readlane64((long long)scan, WAVES - 1);
                if (tid == 0) reinterpret_cast<CertHead*>(cert)->total = all;
                cert_level1<T>(codes, m, s_class, r.walk_slack, r.fast_min_exp, cert, before, tid);
            }
            __syncthreads();
            PSA_FPHASE(3);
            // ---- pass 3
            for (int j = 0; j < cnt; ++j) {
                unsigned char* codes = s_codes + (size_t)j * region;
                unsigned char* cert = codes + codes_bytes;
                cert_level2<T>(codes, m, s_class, r.walk_slack, r.fast_min_exp, cert, tid);
                if (wave == WAVES - 1 - j) {
                    // the first chars: a plain chain (cpu_funcs.c:278), lanes past the end hold +0.0
                    const int head_n = m < kCertChain ? m : kCertChain;
                    const double hs = chain64(0.0, lane < head_n ? s_class[codes[lane] & 7u] : 0.0);
                    if (lane == 0) reinterpret_cast<CertHead*>(cert)->head_sum = hs;
                }
            }
            __syncthreads();
        }
        PSA_FPHASE(4);
        // ---- pass 4
        if (wave < cnt) {
            const int o = s_surv[base + wave];
            unsigned char* codes = s_codes + (size_t)wave * region;
            // the best delta rank and its first char; the two symbols there are asked for now and used after the sum
            unsigned long long key = lane < WAVES ? s_keys[wave * WAVES + lane] : 0ull;
            key = wave_max_u64(key);
            const int rank = (int)(key >> 32);
            const int first_i = rank ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffu)) : 0;
            unsigned sym2 = r.s2[first_i], sym1 = r.s1[o + first_i];
            if constexpr (RAW) { sym2 = ascii_code(sym2); sym1 = ascii_code(sym1); }
            WalkStats st = {0, 0, 0};
            double s = 0.0;
            if (!certify || !cert_sum(codes, m, s_class, codes + codes_bytes, s_tab, lane, s))
                s = walk_sum(0.0, codes, 0, m, s_class, &s_walk[wave], r.fast_ok, r.fast_min_exp, lane, st);
            if (lane == 0) {
                CandRec c;
                c.offset = o;
                c.rank = rank;
                if (rank) {
                    c.score = s + s_delta[rank];
                    c.char_offset = first_i;
                    c.ch = s_sub[sym2 * kRow + sym1];
                } else {
                    c.score = s_delta[0];
                    c.char_offset = -1;
                    c.ch = 0;
                }
                s_rec[base + wave] = c;
            }
        }
        __syncthreads();
        PSA_FPHASE(5);
    }
    if (tid == 0) {
        const bool is_max = r.is_max != 0;
        int idx = -1;
        for (int k = 0; k < nsurv; ++k)
            if (s_rec[k].rank != 0 &&
                (idx < 0 || rec_better(is_max, s_rec[k].score, s_rec[k].offset, s_rec[idx].score, s_rec[idx].offset)))
                idx = k;
        if (idx >= 0) {
            const CandRec b = s_rec[idx];
            write_search_tail(r, &b, 0, bad_flag, swept);
        } else {
            write_search_tail(r, nullptr, 0, bad_flag, swept);
        }
        // the search scalars are clean for the next search on this slot
        store_agent(r.cand_count, 0);
        store_agent(r.done, 0);
        __hip_atomic_store(r.global_best, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (r.best_split && tid < kBestSplit) __hip_atomic_store(r.best_split + tid * kBestStride, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    PSA_FPHASE(6);
#ifdef PSA_PHASE_TIMING
    if (tid == 0) {
        g_phase_ts[1003][0] = (unsigned long long)count;
        g_phase_ts[1003][1] = (unsigned long long)nsurv;
        g_phase_ts[1003][2] = certify ? (unsigned long long)reinterpret_cast<const CertHead*>(s_codes + codes_bytes)->nfail : 0ull;
    }
#endif
}

// ---- count kernel --------------------------------------------------------------
constexpr int kShareMaxChunks = 32;       // chunk workgroups of a tile that may share its combine (one claim word each)
constexpr int kShareMaxTiles = 256;       // ... of at most this many tiles (tiles x chunks <= the CUs)
constexpr unsigned long long kShareLeft = 1ull << 63;
constexpr int kShareSpins = 16;           // polls (a memory round trip each, ~0.7 us) of a tile's "ready" tag before a chunk workgroup leaves
struct CountArgs {
    const unsigned char* s1;        // Seq1 symbol codes, padded with kCodePad (32 per slot)
    const unsigned*      classes;   // Seq2 symbols as LDS row offsets (code << kRowShift) in class order (class_index)
    unsigned long long*  partial;   // [nchunks][span_pad]: n_lo | n_hi<<21 | n_both<<42; tile t at t*kTile
    int base;         // first offset of tile 0: the range's first offset rounded down to a multiple of 32
    int span_pad;     // ntiles*kTile
    int class_chars;  // chars of every shift class per chunk (multiple of kUnroll); a chunk = 32 x class_chars chars
    int lds_slots;    // slots per staged row (kTile/32 + class_chars + 1)
    int* tile_arrivals;   // [ntiles], zero between searches: chunk workgroups of a tile that have published their counts
    // shared combine (fused single searches whose tiles x chunks fit the CUs): every chunk workgroup takes a slice of its tile
    int  share;           // 0: the tile's last arriver combines and selects alone
    int  share_spins;     // polls of the tile's "ready" tag before a workgroup leaves (debug library: 0 = leave at once)
    unsigned long long  share_tag;    // this launch's number among the slot's sharing launches (64 bits: never repeats)
    unsigned long long* tile_ready;   // [ntiles] the tag of the launch whose chunk counts of this tile are complete
    unsigned long long* slice_claim;  // [ntiles][kShareMaxChunks] tag | kShareLeft: the slice's owner left; tag: it has a taker
    unsigned long long* share_swept;  // health counter: slices the tiles' last arrivers had to take over (psa_share_sweep_count)
    int  done_target;     // arrivals on tail.done that end the search: tiles (x chunks when sharing)
    // single-chunk searches finish here: score bounds per offset, running best pessimistic
    // bound (atomic max), candidates = offsets whose optimistic bound reaches it.  The counts do
    // not depend on the search mode, so one pass can select for several modes (MAX and MIN of the
    // same weights): one CountSelect each.
    int inline_select;                // 0: only write the raw per-chunk counts to `partial` (psa_debug_counts)
    int nmodes;                       // 1 or 2
    int cap;
    int first;                        // first offset of the searched range
    int lead;                         // offsets of tile 0 in front of the range (first - base)
    int span;
    struct CountSelect {
        unsigned long long* global_best;  // ordered-u64, 0 before the search
        int*                cand;         // [cap] absolute offsets
        double*             cand_upper;   // [cap] their optimistic bounds (re-checked against the final best)
        int*                cand_count;
        BoundParams         bounds;
    } sel[2];
    // Single searches are ONE launch: the last tile workgroup to finish its selection ends the search (finish_search:
    // final bound, surviving candidates, exact re-score, fold, tail).  `tail` = what psa_rescore_kernel would be given.
    int         fused;
    unsigned short wave_lo[16], wave_hi[16];   // every wave's share of the chunk's symbol stream, in rounds of kUnroll chars
    int         lds_bytes;            // dynamic LDS of the launch (the tail carves its buffers out of the dead planes)
    // psa_count_kernel<16, true> (one-shot searches from raw ASCII, no encode launch in front): the caller's sequences;
    // every workgroup converts Seq1's window while it builds the planes and writes Seq2's class-ordered symbols itself
    const unsigned char* raw1;        // n1 bytes
    const unsigned char* raw2;        // m2 bytes
    unsigned*   classes_w;            // = classes, writable
    int*        bad;                  // alphabet flag (bit 0 Seq1, bit 1 Seq2)
    int         n1, m2, class_total;
    // raw launches: the grid's first `raw_converters` (>= 1) workgroups (row 0) are not tiles: they write a slice of Seq2's
    // class order each and arrive on `raw_arrivals` (monotonic: `raw_base` = its value before this launch)
    unsigned*   raw_arrivals;
    unsigned    raw_base;
    int         raw_converters;
    int         raw_expected;         // = raw_converters (the debug library can ask for one more: the wait then gives up)
    // The tiles validate the Seq1 bytes of their own windows, [base, raw_scan_from) all in all.  psa_search_device promises
    // the errors of psa_set_sequences_device + psa_search, which encodes all n1 bytes: the converters look at the rest,
    // [0, base) and [raw_scan_from, n1) — nothing for a search over every offset.
    int         raw_scan_from;
#ifdef PSA_EXPERIMENT
    int         select_frozen;        // debug library: the tiles select against "no bound at all" (forces the overflow path)
#endif
    RescoreArgs tail;
};

#ifdef PSA_PHASE_TIMING
#define PSA_PHASE(k) do { if (tid == 0 && blockIdx.y * gridDim.x + blockIdx.x < 1000) g_phase_ts[blockIdx.y * gridDim.x + blockIdx.x][k] = wall_clock64(); } while (0)
#define PSA_CPHASE(k) do { if (tid == 0 && blockIdx.y * gridDim.x + blockIdx.x < 1000) g_chunk_ts[blockIdx.y * gridDim.x + blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define PSA_PHASE(k) do { } while (0)
#define PSA_CPHASE(k) do { } while (0)
#endif

// CHUNKED = false: the launch has ONE char chunk per tile and selects inline (grid.y == 1: every full-size single search) —
// the chunk hand-off, the shared combine and their loops are compiled out of that instance: its tail is straight-line code.
template <int WAVES, bool RAW = false, bool CHUNKED = true>
__global__ __launch_bounds__(WAVES * 64, 4) void psa_count_kernel(CountArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint2 lds[];   // [kMaskRows][lds_slots] planes, the count cells; the tail's buffers
                                                                  // (16-byte aligned: the tail reads its staged codes as uint4)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Raw launches: the first raw_converters workgroups of the grid's row 0 are not tiles — they
    // write Seq2's symbols in the count loop's class order, ONCE for everybody (a byte load and a write-through store per
    // position, ~3 us on CUs the tiles leave free: 244 tiles on 256 CUs), arrive on raw_arrivals and leave; they are the
    // first to be dispatched and wait for nobody.  The tile workgroups build their planes meanwhile and then wait for the
    // arrivals, which are there by then.  (First form: every tile workgroup wrote all 40 KB itself, staged through LDS —
    // 4.3 us per workgroup, prologue 8.6 us instead of 5.2.  Tried in between: the slices written by one wave of each of
    // 128 tile workgroups — 128 arrivals on one counter under 244 pollers took 6 us to become visible.)
    const int conv_wgs = RAW ? a.raw_converters : 0;
    const int tile_x = (int)blockIdx.x - conv_wgs;
    if constexpr (RAW) if (tile_x < 0) {
        if (blockIdx.y != 0) return;
        const int total = a.class_total + kUnroll;
        const int per = (total + conv_wgs - 1) / conv_wgs;
        const int j0 = (int)blockIdx.x * per;
        const int j1 = j0 + per < total ? j0 + per : total;
        const int cc = a.class_chars;
        const float rcc = 1.0f / (float)cc;
        // position j of the class order holds char i = 32 (c cc + kk) + 31 - cs with j = ((32 c + cs) cc + kk); up to
        // kRawPerThread positions per thread (host), all their loads in flight together
        int at[kRawPerThread];
        unsigned char ch[kRawPerThread];
#pragma unroll
        for (int t = 0; t < kRawPerThread; ++t) {
            const int j = j0 + tid + WAVES * 64 * t;
            at[t] = -1;
            ch[t] = 0;
            if (j < j1 && j < a.class_total) {
                int q = (int)((float)j * rcc);
                int kk = j - q * cc;
                if (kk < 0) { kk += cc; --q; }
                if (kk >= cc) { kk -= cc; ++q; }
                const int i = (((q >> 5) * cc + kk) << 5) + (kClasses - 1 - (q & 31));
                if (i < a.m2) { at[t] = i; ch[t] = a.raw2[i]; }
            }
        }
#pragma unroll
        for (int t = 0; t < kRawPerThread; ++t) {
            const int j = j0 + tid + WAVES * 64 * t;
            if (j < j1) {
                unsigned code = kCodePad;
                if (at[t] >= 0) {
                    code = ascii_code(ch[t]);
                    if (code == (unsigned)kCodePad) atomicOr(a.bad, 2);              // a character of Seq2 outside [A-Z-]
                }
                __hip_atomic_store(a.classes_w + j, code << kRowShift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // Seq1 outside the tiles' windows (a search of a sub-range): 16 bytes per thread and step, all converters together
        {
            const int lo_units = a.base >> 4;                                   // [0, base): base is a multiple of 32
            const int hi_units = a.raw_scan_from < a.n1 ? (a.n1 - a.raw_scan_from + 15) >> 4 : 0;
            const bool aligned = ((reinterpret_cast<size_t>(a.raw1) | (size_t)a.raw_scan_from) & 15) == 0;
            bool any_bad = false;
#pragma unroll 4
            for (int u = (int)blockIdx.x * WAVES * 64 + tid; u < lo_units + hi_units; u += conv_wgs * WAVES * 64) {
                const int p0 = u < lo_units ? u << 4 : a.raw_scan_from + ((u - lo_units) << 4);
                unsigned w[4];
                if (aligned && p0 + 16 <= a.n1) {
                    const uint4 q = *reinterpret_cast<const uint4*>(a.raw1 + p0);
                    w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        w[q] = 0x41414141u;                                     // behind the end: 'A'
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (p0 + 4 * q + r < a.n1) w[q] = (w[q] & ~(0xffu << (8 * r))) | ((unsigned)a.raw1[p0 + 4 * q + r] << (8 * r));
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) any_bad |= ascii_code((w[q] >> (8 * r)) & 0xffu) == (unsigned)kCodePad;
            }
            if (any_bad) atomicOr(a.bad, 1);                                    // a character of Seq1 outside [A-Z-]
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave's stores have been performed,
        __syncthreads();                                            // then ONE lane's arrival
        if (tid == 0) __hip_atomic_fetch_add(a.raw_arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int tile_o0 = a.base + tile_x * kTileStride;         // absolute offset of word 0, bit 0 (multiple of 32)
    const int chunk_k0 = blockIdx.y * a.class_chars;           // first char k of every class in this chunk
    const int slot0 = (tile_o0 >> 5) + chunk_k0;               // first staged slot
    // rows are 4 KiB apart: a Seq2 symbol, stored as code << 12, is its row's byte offset.  (A packed-row variant
    // with 8 waves and two workgroups per CU was measured slower than this kernel for every shape: removed.)
    static_assert(WAVES == 16, "one shape: 16 waves, one workgroup per CU");
    auto row_of = [](unsigned sym) -> unsigned { return sym; };

    // Build the class planes of the window [32*slot0, +kTile+32*class_chars) in LDS: row b
    // (Seq2 symbol), slot s holds {lo, hi}: bit t of lo = class(Seq1[32*(slot0+s)+t], b) in
    // {'*','.'}, of hi = class in {':','.'}.  One thread per slot: its 32 symbols (codes from the encoded Seq1, or the
    // caller's ASCII) -> the 27 "holds symbol a" masks -> the rows (symbol_masks / store_plane_rows above) — 14 KB of
    // codes read per workgroup instead of 100 KB of planes, no table, ~230 instructions per slot.
    // The tile's count cells lie behind the exchange area of the waves' counter sums (which covers the plane rows); they
    // are written once, by sum_columns, with plain stores.
    unsigned* const cells = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds) + kCellOffset);
    PSA_PHASE(0);
    build_tile_planes<WAVES * 64, RAW>(a.s1, a.raw1, a.n1, a.bad, slot0, a.lds_slots, lds, tid);
    __syncthreads();
    if constexpr (RAW) {
        // everybody waits until all converters have arrived: the symbols are complete (and, written through, in memory)
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(a.raw_arrivals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.raw_base <
                   (unsigned)a.raw_expected) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 17)) {                // ~0.1 s: never expected; the host redoes the search with an
                    atomicOr(a.bad, 16);                  // encode launch in front (psa_raw_retry_count)
                    break;
                }
            }
        }
        __syncthreads();
    }
    PSA_PHASE(1);

    SlicedCounter cnt[2][kCounters];     // [word][n_lo, n_hi, n_both]
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int k = 0; k < kCounters; ++k) cnt[w][k].clear();

    // this wave's share of the chunk's symbol stream (host: deal_single — the hardware favours the oldest wave of a
    // SIMD (waves w, w+4, w+8, w+12 share one), so the shares differ by age and all four finish together)
    const int class_chars_u = __builtin_amdgcn_readfirstlane(a.class_chars);
    const WaveShare share = wave_share(__builtin_amdgcn_readfirstlane((int)a.wave_lo[wave]),
                                       __builtin_amdgcn_readfirstlane((int)a.wave_hi[wave]), class_chars_u);
    // constant address space: keeps these wave-uniform loads scalar (s_load) although the loop
    // contains volatile asm
    using const_u32 = const __attribute__((address_space(4))) unsigned;
    // (32-bit index arithmetic: a 64-bit multiply would go through the VALU and make these vector loads)
    const int class_chars = class_chars_u;
    const int sym_at = __builtin_amdgcn_readfirstlane((int)blockIdx.y * kClasses * class_chars + share.sym_at);
    const_u32* sym = (const_u32*)(size_t)(a.classes + sym_at);
    if constexpr (RAW) {
        // the symbols were stored in this launch (behind the barrier above): drop what the scalar cache holds, and tie
        // the pointer to that point so that no scalar load of the symbols can be scheduled in front of it
        asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" : "+s"(sym) : : "memory");
    }
    const unsigned lane_base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds + (unsigned)lane * 8u;
    count_stream(cnt, sym, share.first_k, share.chars, class_chars, share.low_class, lane_base, lane, row_of);

    // ---- epilogue: the waves' counter planes are summed bit-sliced (sum_dump / sum_take / sum_columns), the totals turned
    // into integers (bit-matrix transpose in registers, two 16-bit counts per dword) and stored to the tile's cells;
    // each thread then picks up the counts of its offsets.
#ifdef PSA_PHASE_TIMING
    // (when the workgroup's LAST wave left its main loop; zeroed by the host between searches it is not: the maximum
    // of 100 MHz timestamps only grows)
    if (lane == 0 && blockIdx.y * gridDim.x + blockIdx.x < 1000) atomicMax(&g_phase_ts[blockIdx.y * gridDim.x + blockIdx.x][2], wall_clock64());
#endif
    {
        unsigned* const X = reinterpret_cast<unsigned*>(lds);
        __syncthreads();                                  // every wave's main loop has ended: the plane rows are dead
        if (wave >= 8) sum_dump(cnt, nullptr, X + (wave - 8) * kXSlotDwords, lane);
        __syncthreads();
        PSA_PHASE(3);
        if (wave < 8) {
            unsigned top[2][kCounters];
            sum_take(cnt, top, X + wave * kXSlotDwords, lane);
            sum_dump(cnt, top, X + wave * kXSlotDwords, lane);   // over what this wave has just read (a wave's LDS
        }                                                        // operations execute in order)
        __syncthreads();
        PSA_PHASE(4);
        if (tid < 768) sum_columns(X, cells, tid);
    }
    __syncthreads();

    // each thread handles kTile / (WAVES*64) consecutive offsets of one word
    constexpr int kPer = kTile / (WAVES * 64);
    static_assert(kPer >= 1 && 16 % kPer == 0, "tile/threads");
    int o_local = tid * kPer;
    unsigned n[kCounters][kPer];
    read_cells<kPer>(n, cells, kTileWords, o_local);
    PSA_CPHASE(0);
    __shared__ unsigned long long s_best[2][WAVES + 1];
    __shared__ int s_flag;
    __shared__ int s_last_tile;
    const int nchunks = CHUNKED ? (int)gridDim.y : 1;
    const unsigned long long tag = a.share_tag;         // this launch's number: what "ready" and "claimed" words must show
    bool last_chunk = true;
    if (CHUNKED && (!a.inline_select || nchunks > 1)) {
        // ---- several char chunks per tile (or psa_debug_counts): every chunk workgroup publishes its counts
        unsigned long long* dst = a.partial + (size_t)blockIdx.y * a.span_pad + (size_t)tile_x * kTile + o_local;
        // write-through (sc1) stores: they leave the XCD's L2 at once, so the hand-off below needs no L2 write-back
        // (16 bytes each: half the requests of 8-byte atomic stores)
        static_assert(kPer == 4, "two 16-byte stores per thread");
        unsigned long long pk[kPer];
#pragma unroll
        for (int e = 0; e < kPer; ++e)
            pk[e] = (unsigned long long)n[0][e] | ((unsigned long long)n[1][e] << 21) | ((unsigned long long)n[2][e] << 42);
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u q0 = {(unsigned)pk[0], (unsigned)(pk[0] >> 32), (unsigned)pk[1], (unsigned)(pk[1] >> 32)};
        const v4u q1 = {(unsigned)pk[2], (unsigned)(pk[2] >> 32), (unsigned)pk[3], (unsigned)(pk[3] >> 32)};
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1"
                     : : "v"(dst), "v"(q0), "v"(q1) : "memory");
        if (!a.inline_select) return;
        // The chunk workgroups of a tile meet through memory; they may sit on different XCDs (separate L2s).  Producer:
        // every byte is stored sc1, every storing wave drains its stores (vmcnt(0)), the workgroup barrier, then ONE
        // lane's arrival (MI355X_MICROARCH.md, "Valid forms": no release fence — an L2 write-back costs ~6 us per
        // workgroup here).  Consumer: agent-scope acquire behind the arrival counter (or the tile's "ready" tag), barrier,
        // agent-scope loads.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PSA_CPHASE(1);
        __syncthreads();
        PSA_CPHASE(2);
        if (tid == 0) {
            const int arrived = __hip_atomic_fetch_add(a.tile_arrivals + tile_x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = arrived == nchunks - 1;
            s_flag = last;
            if (last) {
                __hip_atomic_store(a.tile_arrivals + tile_x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.share) __hip_atomic_store(a.tile_ready + tile_x, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        PSA_CPHASE(3);
        __syncthreads();
        PSA_CPHASE(4);
        last_chunk = __builtin_amdgcn_readfirstlane(s_flag) != 0;
        if (!last_chunk) {
            // Without sharing the tile's last chunk workgroup to arrive adds the others' counts to its own and goes on
            // alone (plans of more workgroups than CUs, batches: the others have no reason to be resident still).
            if (!a.share) return;
            // sharing: wait — with a bound — for the tile's "ready" tag.  A workgroup that has waited in vain marks its slice
            // "left" and looks once more (the tag may have come in between: then it takes the slice back unless the tile's
            // last arriver has taken it already).
            __syncthreads();
            if (tid == 0) {
                int go = 0;
                for (int spins = 0; spins < a.share_spins; ++spins) {
                    if (__hip_atomic_load(a.tile_ready + tile_x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) { go = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!go) {
                    unsigned long long* const mine = a.slice_claim + (size_t)tile_x * kShareMaxChunks + blockIdx.y;
                    (void)__hip_atomic_exchange(mine, tag | kShareLeft, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the mark is in memory before the second look
                    if (__hip_atomic_load(a.tile_ready + tile_x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) {
                        unsigned long long expect = tag | kShareLeft;
                        go = __hip_atomic_compare_exchange_strong(mine, &expect, tag, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1 : 0;
                    }
                }
                if (go) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                s_flag = go;
            }
            __syncthreads();
            if (!__builtin_amdgcn_readfirstlane(s_flag)) return;     // left: the tile's last arriver takes this slice
        }
    }
    // ---- SHARED COMBINE (a fused single search whose tiles x chunks fit the GPU's CUs: an offset shard).  One workgroup
    // adding up (C - 1) x 32 KB of other workgroups' counts is bound by what ONE CU keeps in flight (measured, 5 chunks:
    // 3 us for its first wave + 5.4 us until its last, then bounds and selection for all 4064 offsets alone) while the
    // tile's other C - 1 CUs have gone idle.  Instead every chunk workgroup waits for the tile's "ready" tag (the last
    // arriver's store) and takes the C-th part of the tile's offsets — slice c = blockIdx.y: its own counts from LDS, the
    // others' from memory, bounds, selection, and its own arrival on the search's counter (done_target = tiles x chunks).
    // NOBODY WAITS WITHOUT A BOUND: a workgroup that does not see the tag within share_spins polls marks its slice "left"
    // and goes (its peers are not resident: the GPU is shared).  The tile's last arriver, which waits for nobody, takes its
    // own slice and then every slice it finds marked "left" (compare-and-swap left -> taken) — so each slice is processed
    // exactly once: by its owner if the owner saw the tag (no atomic on that path) or won its slice back, else by the last
    // arriver, whose sweep comes after the tag is in memory (an owner that marks "left" later sees the tag at its second
    // look).  Tags are 64-bit launch numbers: no word needs zeroing between launches.
    // Without sharing there is one "slice": the whole tile, this workgroup's.
    const bool sharing = CHUNKED && nchunks > 1 && a.share != 0;
    const int nslices = sharing ? nchunks : 1;
    const int groups = (sharing ? kTileStride : kTile) / kPer;           // groups of kPer offsets that are dealt out
    unsigned long long* const claim = a.slice_claim + (size_t)tile_x * kShareMaxChunks;
    bool finish = false;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        unsigned long long mine = 1ull;
        if (!sharing) {
            if (pass == 1) break;
        } else {
            if (pass == 0) {
                mine = 1ull << blockIdx.y;                 // its own slice
            } else {
                if (!last_chunk) break;
                // the tile's last arriver: every other slice whose owner has left — one lane per slice, all at once
                __syncthreads();
                if (wave == 0) {
                    bool got = false;
                    if (lane < nchunks && lane != (int)blockIdx.y) {
                        unsigned long long expect = tag | kShareLeft;
                        got = __hip_atomic_compare_exchange_strong(claim + lane, &expect, tag, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const unsigned long long all = __ballot(got);
                    if (lane == 0) s_best[0][WAVES] = all;
                }
                __syncthreads();
                const unsigned long long all = s_best[0][WAVES];
                mine = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)all) |
                       ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(all >> 32)) << 32);
                __syncthreads();
            }
        }
        if (mine == 0ull) break;                         // (pass 1: every other owner took its slice itself)
        {
            // pass 0: ONE slice — this workgroup's (without sharing: the whole tile) — dealt to threads 0 .. its size;
            // pass 1 (the tile's last arriver): ALL the slices it swept in one go, every thread at its place in the tile
            // (all owners leaving is how it goes when the GPU is shared: five slices one after the other cost 40 us)
            const int taken = __builtin_popcountll(mine);
            bool active;
            if (pass == 0) {
                const int slice = __builtin_ctzll(mine);
                const int g0 = slice * groups / nslices, g1 = (slice + 1) * groups / nslices;
                active = tid < g1 - g0;
                if constexpr (CHUNKED) o_local = (active ? g0 + tid : g0) * kPer;
            } else {
                int slice = tid * nslices / groups;                         // the slice whose [g0, g1) holds group `tid`
                if (slice + 1 < nslices && tid >= (slice + 1) * groups / nslices) ++slice;
                active = tid < groups && ((mine >> slice) & 1ull) != 0ull;
                o_local = tid * kPer;
                if (tid == 0) __hip_atomic_fetch_add(a.share_swept, (unsigned long long)taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if constexpr (CHUNKED) read_cells<kPer>(n, cells, kTileWords, o_local);
            if (CHUNKED && nchunks > 1) {
                // the other chunks' counts of this thread's kPer offsets.  The loads of kCombineAhead chunks are all issued
                // before the first is used (a slot beyond the last chunk reads this workgroup's own counts and is masked
                // out: straight-line code, no branch between the loads).  A rolled loop waited one memory round trip per
                // chunk, 1.7 us each: the planner's former "chunk term".
                constexpr int kCombineAhead = 4;
                const int nothers = nchunks - 1;
                const unsigned long long* const at = a.partial + (size_t)tile_x * kTile + o_local;
                for (int j0 = 0; j0 < nothers; j0 += kCombineAhead) {
                    unsigned long long v[kCombineAhead][kPer];
#pragma unroll
                    for (int j = 0; j < kCombineAhead; ++j) {
                        const bool take = j0 + j < nothers;
                        const int c = j0 + j < (int)blockIdx.y ? j0 + j : j0 + j + 1;      // the chunks but this one, in order
                        const unsigned long long* src = at + (size_t)(take ? c : (int)blockIdx.y) * a.span_pad;
#pragma unroll
                        for (int e = 0; e < kPer; ++e) {
                            v[j][e] = __hip_atomic_load(src + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v[j][e] = take ? v[j][e] : 0ull;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < kCombineAhead; ++j)
#pragma unroll
                        for (int e = 0; e < kPer; ++e) {
                            n[0][e] += (unsigned)(v[j][e] & 0x1fffffu);
                            n[1][e] += (unsigned)((v[j][e] >> 21) & 0x1fffffu);
                            n[2][e] += (unsigned)((v[j][e] >> 42) & 0x1fffffu);
                        }
                }
            }
            // the counts are final
            PSA_PHASE(5);
            PSA_CPHASE(5);
            const int valid = active ? kTileStride - o_local : 0;
            double up[2][kPer];
            for (int mode = 0; mode < a.nmodes; ++mode) {
                // offsets relative to the range's first one; the tile answers for its first kTileStride positions
                const double pess = thread_bounds<kPer>(a.sel[mode].bounds, n, tile_x * kTileStride + o_local - a.lead,
                                                        a.span, valid, up[mode]);
                unsigned long long key = order_key(pess);
                key = wave_max_u64(key);
                if (lane == 0) s_best[mode][wave] = key;
            }
            PSA_CPHASE(6);
            __syncthreads();
            PSA_CPHASE(7);
            if (tid < a.nmodes) {
                // running best over the tiles finished so far: a lower bound of the final best, so selecting against it
                // keeps a superset of the true candidates (whoever re-scores re-checks against the final bound)
                unsigned long long key = s_best[tid][0];
                for (int w = 1; w < WAVES; ++w) key = s_best[tid][w] > key ? s_best[tid][w] : key;
#ifdef PSA_PHASE_TIMING
                if (tid == 0 && blockIdx.x < 244) g_phase_ts[700 + blockIdx.x][0] = wall_clock64();
#endif
                unsigned long long old;
                if (a.fused && a.tail.best_split) {
                    // (fused: one mode.  In a re-selection launch every word starts from the first pass's final bound.)
                    old = atomicMax(a.tail.best_split + (tile_x & (kBestSplit - 1)) * kBestStride, key);
                } else {
                    old = atomicMax(a.sel[tid].global_best, key);
                }
                s_best[tid][WAVES] = old > key ? old : key;
#ifdef PSA_PHASE_TIMING
                if (tid == 0 && blockIdx.x < 244) g_phase_ts[700 + blockIdx.x][1] = wall_clock64();
#endif
            }
            PSA_CPHASE(8);
            __syncthreads();
            PSA_CPHASE(9);
            for (int mode = 0; mode < a.nmodes; ++mode) {
                const CountArgs::CountSelect& sel = a.sel[mode];
                double need = key_order(s_best[mode][WAVES]);
#ifdef PSA_EXPERIMENT
                if (a.select_frozen) need = -1.7976931348623157e308;    // every real offset is a candidate (not the -inf of the others)
#endif
                // (a SLICE may lie wholly outside the searched range — the last tile's end — and run before any bound exists:
                // its key and `need` are then -inf, and "-inf >= -inf" must not make candidates of offsets that do not exist:
                // 4,096 of them overflowed the list and sent the search through a re-selection launch, now and then)
#pragma unroll
                for (int e = 0; e < kPer; ++e)
                    if (up[mode][e] >= need && up[mode][e] != -INFINITY) {
                        const int slot = atomicAdd(sel.cand_count, 1);
                        if (slot < a.cap) {
                            store_agent(sel.cand + slot, a.first + tile_x * kTileStride + o_local + e - a.lead);
                            store_agent(sel.cand_upper + slot, up[mode][e]);
                        }
                    }
            }
            PSA_PHASE(6);
            if (!a.fused) return;
            // ---- a single search ends in this launch: the last of `done_target` selections to get here finishes it.
            // Producer side of the hand-off: the candidate stores above are write-through, every wave drains its own, the
            // workgroup barrier, then ONE lane's arrival; the last arriver acquires.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // (finish_search's survivor counter: the plane rows / exchange area it lies in are dead since the counter sums)
            if (tid == 0) *reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(lds) + kFusedCounterOffset) = 0;
            __syncthreads();
            if (tid == 0) {
#ifdef PSA_PHASE_TIMING
                if (blockIdx.x < 244) g_phase_ts[700 + blockIdx.x][2] = wall_clock64();
#endif
                const int arrived = __hip_atomic_fetch_add(a.tail.done, taken, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_last_tile = arrived + taken == a.done_target;
#ifdef PSA_PHASE_TIMING
                if (blockIdx.x < 244) { g_phase_ts[700 + blockIdx.x][3] = wall_clock64(); g_phase_ts[700 + blockIdx.x][4] = (unsigned long long)arrived; }
#endif
                if (s_last_tile) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            finish = __builtin_amdgcn_readfirstlane(s_last_tile) != 0 || finish;
        }
    }
    if (!finish) return;
    finish_search<WAVES, RAW>(a.tail, reinterpret_cast<unsigned char*>(lds), a.lds_bytes, tid);
    PSA_PHASE(7);
}

// ---- duo count kernel (batches) ------------------------------------------------
// One workgroup per tile handles SEVERAL queries of a batch.  The tile's class planes depend on
// Seq1 only: they are built once and shared by two groups of 8 waves.  Group g takes queries
// g, g+2, ...: main loop over all chars of its query (8 waves x 4 shift classes), then its own epilogue
// (counts into its own LDS cells, bounds, candidates), synchronised by a group-local barrier (an
// LDS counter; s_barrier would stop the other group too).  Group 1 starts one epilogue late, so
// while one group runs the latency-bound epilogue the other keeps the VALUs busy with its main
// loop — in psa_count_kernel those phases (about 15 % of the kernel) leave the CU idle.
constexpr int kDuoGroupWaves = 8;
constexpr int kDuoPlanes = 11;             // per-wave counters: up to 2047 chars
constexpr int kDuoMaxWaveChars = 2016;     // multiple of kUnroll below 2^kDuoPlanes
constexpr int kDuoPer = kTile / (kDuoGroupWaves * 64);     // offsets per thread in the epilogue
// per group: the count cells of the 127 words a tile answers for (the planes take 112 of the CU's 160 KiB)
constexpr int kDuoCellWords = kTileStride / 32;
constexpr int kDuoCellDwords = kCounters * 16 * kDuoCellWords;
constexpr int kDuoScratchBytes = kDuoCellDwords * 4;
static_assert(kDuoGroupWaves * kDuoMaxWaveChars < 65536, "a group's counts fit 16 bits");
constexpr int kDuoLdsBytes = kMaskRows * kRowSlots * 8 + 2 * kDuoScratchBytes;
static_assert(kDuoLdsBytes + 256 <= 160 * 1024, "planes + two groups' cells + the kernel's static LDS fit one CU");
constexpr int kScalStride = 16;            // bytes between the search scalars of consecutive queries (host: Scalars)
constexpr int kScalCountOffset = 8;        // cand_count inside them

struct DuoArgs {
    const unsigned char* s1;
    const unsigned*      classes;      // class-ordered symbols of the launch's first query; query q at + q * s2_stride
    size_t               s2_stride;    // in elements
    int base;             // first offset of tile 0 (multiple of 32)
    int class_chars;      // chars per shift class (multiple of kUnroll); a wave walks 4 classes (<= kDuoMaxWaveChars chars)
    int lds_slots;        // slots per staged row
    int nq;               // queries of this launch
    int qpb;              // queries per workgroup: workgroup (tile, y) serves queries [y * qpb, (y + 1) * qpb) of the launch.
                          // Long Seq1: one workgroup per tile serves the whole launch; few tiles: the queries are spread
                          // over grid.y so that the launch still fills the GPU
    int nmodes;
    int cap;
    int first, lead, span;
    unsigned skew_ticks;  // 100 MHz ticks group 1 waits before its first query
    int spin_limit;       // polls a group barrier waits before it gives up (debug library: 0 forces the give-up path)
    unsigned short wave_lo[8], wave_hi[8];   // the shares of a group's eight waves of a query's symbol stream, in rounds
    int prio_mode;        // 0: the younger waves of a group at the higher priority (product); 1-3: experiments
#ifdef PSA_EXPERIMENT
    int select_frozen;    // debug library: select against "no bound at all" (forces the overflow path)
#endif
    int* bad;             // |= 4 if a group barrier timed out (never expected): the host redoes the batch without this kernel
    struct Select {
        char*   scalars;      // search scalars of the launch's first query (kScalStride apart)
        int*    cand;         // candidate offsets, `cap` per query
        double* cand_upper;
        BoundParams bounds;
    } sel[2];
};

// barrier of the 8 waves of one group: arrival counter in LDS, monotonic (target grows by 8 per barrier)
__device__ __forceinline__ void duo_barrier(unsigned* ctr, unsigned& target, int lane, int* bad, int spin_limit)
{
    target += (unsigned)kDuoGroupWaves;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // this wave's LDS traffic is complete
    if (lane == 0) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    int spins = 0;
    while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > spin_limit) {                                  // 2^21 polls ~ 0.1 s: give up loudly instead of hanging
            if (lane == 0) atomicOr(bad, 4);
            break;
        }
    }
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(1024) void psa_count_duo_kernel(DuoArgs a)
{
    extern __shared__ uint2 lds[];                  // [kMaskRows][kRowSlots] planes, then one scratch per group
    __shared__ unsigned s_bar[2];
    __shared__ unsigned long long s_best[2][kDuoGroupWaves + 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 3;
    const int gw = wave & 7;                         // wave inside its group
    const int gtid = tid & (kDuoGroupWaves * 64 - 1);
    const int tile_o0 = a.base + blockIdx.x * kTileStride;
    const int slot0 = tile_o0 >> 5;

    // ---- planes of the window [tile_o0, +kTile+32*class_chars): as psa_count_kernel<16>
    if (tid < 2) s_bar[tid] = 0u;
    build_tile_planes<1024, false>(a.s1, nullptr, 0, nullptr, slot0, a.lds_slots, lds, tid);
    __syncthreads();
    {   // both groups' count cells start from zero
        unsigned* z = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds) + kMaskRows * kRowSlots * 8);
        for (int i = tid; i < 2 * kDuoCellDwords; i += 1024) z[i] = 0u;
    }
    __syncthreads();                                 // the last workgroup-wide barrier: the groups part here

    unsigned* const cells = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(lds) + kMaskRows * kRowSlots * 8 +
                                                        group * kDuoScratchBytes);
    unsigned* const bar = &s_bar[group];
    unsigned bar_target = 0u;
    const int q_begin = (int)blockIdx.y * a.qpb;
    const int q_end = q_begin + a.qpb < a.nq ? q_begin + a.qpb : a.nq;
    if (group == 1 && q_end - q_begin > 2) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < a.skew_ticks) __builtin_amdgcn_s_sleep(8);
    }
    using const_u32 = const __attribute__((address_space(4))) unsigned;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds;

#ifdef PSA_PHASE_TIMING
#define PSA_DPHASE(k) do { if (gtid == 0 && blockIdx.x == 7 && (q >> 1) < 16) g_phase_ts[600 + group * 16 + (q >> 1)][k] = wall_clock64(); } while (0)
#else
#define PSA_DPHASE(k) do { } while (0)
#endif
    for (int q = q_begin + group; q < q_end; q += 2) {
        PSA_DPHASE(0);
        // younger waves of the group first: the hardware favours the oldest wave, which would finish its share
        // long before the others and wait for them at the first epilogue barrier
        {
            int prio = (gw >> 1) + group;          // and the younger group a notch above the older one
            if (a.prio_mode == 1) prio = 0;                                   // experiment: no priorities in the main loop
            else if (a.prio_mode == 2) prio = 3 - (gw >> 1) - group;          // experiment: the older waves first
            else if (a.prio_mode == 3) prio = (gw >> 2) + group;              // experiment: two levels per group
            switch (prio) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
        // ---- main loop: this wave's chars of query q against the whole tile
        SlicedCounterT<kDuoPlanes> cnt[2][kCounters];
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int k = 0; k < kCounters; ++k) cnt[w][k].clear();
        // this wave's share of the query's symbol stream (host: deal_duo)
        const int class_chars = __builtin_amdgcn_readfirstlane(a.class_chars);
        const WaveShare share = wave_share(__builtin_amdgcn_readfirstlane((int)a.wave_lo[gw]),
                                           __builtin_amdgcn_readfirstlane((int)a.wave_hi[gw]), class_chars);
        const int sym_at = __builtin_amdgcn_readfirstlane(share.sym_at);   // scalar unit
        const_u32* sym = (const_u32*)(size_t)(a.classes + (size_t)q * a.s2_stride + sym_at);
        count_stream(cnt, sym, share.first_k, share.chars, class_chars, share.low_class, lds_base + (unsigned)lane * 8u, lane,
                     [](unsigned sym_off) { return sym_off; });

        PSA_DPHASE(1);
        // the epilogue is a chain of short, latency-bound steps: let its instructions pass the other group's main loop
        __builtin_amdgcn_s_setprio(3);
        // ---- every wave adds its counts to the group's cells; then each thread picks up its offsets'
        // counts (kDuoPer consecutive offsets of one word) and the cells are zeroed for the next query
        add_counts_to_cells(cnt, cells, kDuoCellWords, lane, false);
        duo_barrier(bar, bar_target, lane, a.bad, a.spin_limit);
        PSA_DPHASE(2);
        const int o_local = gtid * kDuoPer;
        unsigned n[kCounters][kDuoPer];
        read_cells<kDuoPer>(n, cells, kDuoCellWords, o_local);
        duo_barrier(bar, bar_target, lane, a.bad, a.spin_limit);
        for (int i = gtid; i < kDuoCellDwords; i += kDuoGroupWaves * 64) cells[i] = 0u;
        PSA_DPHASE(3);
        // ---- bounds, running best, candidates (per mode), as psa_count_kernel
        for (int mode = 0; mode < a.nmodes; ++mode) {
            const DuoArgs::Select& sel = a.sel[mode];
            unsigned long long* global_best = reinterpret_cast<unsigned long long*>(sel.scalars + (size_t)q * kScalStride);
            int* cand_count = reinterpret_cast<int*>(sel.scalars + (size_t)q * kScalStride + kScalCountOffset);
            int* cand = sel.cand + (size_t)q * a.cap;
            double* cand_upper = sel.cand_upper + (size_t)q * a.cap;
            double up[kDuoPer];
            const double pess = thread_bounds<kDuoPer>(sel.bounds, n, (int)blockIdx.x * kTileStride + o_local - a.lead, a.span,
                                                       kTileStride - o_local, up);
            unsigned long long key = order_key(pess);
            key = wave_max_u64(key);
            duo_barrier(bar, bar_target, lane, a.bad, a.spin_limit);          // the previous mode's s_best has been read
            if (lane == 0) s_best[group][gw] = key;
            duo_barrier(bar, bar_target, lane, a.bad, a.spin_limit);
            if (gtid == 0) {
                for (int w = 1; w < kDuoGroupWaves; ++w) key = s_best[group][w] > key ? s_best[group][w] : key;
                const unsigned long long old = atomicMax(global_best, key);
                s_best[group][kDuoGroupWaves] = old > key ? old : key;
            }
            duo_barrier(bar, bar_target, lane, a.bad, a.spin_limit);
            double need = key_order(s_best[group][kDuoGroupWaves]);
#ifdef PSA_EXPERIMENT
            if (a.select_frozen) need = -1.7976931348623157e308;
#endif
#pragma unroll
            for (int e = 0; e < kDuoPer; ++e)
                if (up[e] >= need) {
                    const int slot = atomicAdd(cand_count, 1);
                    if (slot < a.cap) {
                        cand[slot] = a.first + (int)blockIdx.x * kTileStride + o_local + e - a.lead;
                        cand_upper[slot] = up[e];
                    }
                }
        }
        duo_barrier(bar, bar_target, lane, a.bad, a.spin_limit);              // the scratch is free for the next query
        PSA_DPHASE(4);
    }
}

// One workgroup per candidate.  All waves classify the chars of a chunk into LDS
// (class | rank codes; loads of the whole chunk in flight together) and track the best
// delta rank and its first position; wave 0 then runs the summation over the staged codes.
__global__ __launch_bounds__(kRescoreThreads) void psa_rescore_kernel(RescoreArgs a)
{
    __shared__ unsigned char      s_clsrank[kRow * kRow];
    __shared__ __attribute__((aligned(16))) unsigned char s_codes[kRescoreChunk + 64 * kRescoreEpl + 16];
    __shared__ WalkScratch        s_walk;
    __shared__ double             s_class[8];     // the class values, indexable (a by-value kernel argument indexed
                                                  // at run time would be copied to scratch by every lane)
    __shared__ unsigned long long s_key[kRescoreThreads / 64];
    // the certificate of a Seq2 that is staged whole (m <= kRescoreChunk): cert_layout(kRescoreChunk) bytes
    constexpr int kCertBytes = ((int)sizeof(CertHead) + 8 * (kRescoreChunk / kCertSeg + 1) + 4 * (kRescoreChunk / kCertSeg + 2) +
                                4 * (kRescoreChunk / 32 + 2) + 4 * kCertSummary + 2 * kCertMaxFail + 2 * kCertMaxEv + 15) & ~15;
    __shared__ __attribute__((aligned(16))) unsigned char s_cert[kCertBytes];
    __shared__ CertTable         s_tab;
    __shared__ unsigned long long s_wsum[kRescoreThreads / 64];
    __shared__ int s_last;
    extern __shared__ unsigned char s_pad[];      // never touched: only makes the workgroup too big to share a
    (void)s_pad;                                  // CU with a count-kernel workgroup (LDS), see launch site
    if (a.batched) {
        const size_t q = blockIdx.y;
        a.s2 += q * a.s2_stride;
        a.cand += q * (size_t)a.cap;
        if (a.cand_upper) a.cand_upper += q * (size_t)a.cap;
        a.rec += q * (size_t)a.cap;
        a.global_best = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.global_best) + q * kScalStride);
        a.cand_count = reinterpret_cast<int*>(reinterpret_cast<char*>(a.cand_count) + q * kScalStride);
        a.done = reinterpret_cast<int*>(reinterpret_cast<char*>(a.done) + q * kScalStride);
        a.out += q;
        a.seq += (int)q;
        if (a.record_out) a.record_out += q;
    }
#ifdef PSA_PHASE_TIMING
#define PSA_RPHASE(k) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_phase_ts[1000][k] = wall_clock64(); } while (0)
#else
#define PSA_RPHASE(k) do { } while (0)
#endif
    PSA_RPHASE(0);
    const int count = *a.cand_count;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (count > a.cap) {
        // The candidate list overflowed: candidates were picked against a RUNNING best, so early tiles can
        // contribute far more than survive.  Nothing is re-scored; the bound in global_best is final now and
        // stays, the host runs the selection once more against it (only a second overflow — more true
        // near-ties than the list holds — goes to the sequential engine).
        if (blockIdx.x == 0 && tid == 0) {
            write_search_tail(a, nullptr, 1);
            *a.cand_count = 0;
            *a.done = 0;
        }
        return;
    }
    if ((int)blockIdx.x >= count) return;
    for (int k = tid; k < kRow * kRow; k += kRescoreThreads) s_clsrank[k] = a.clsrank[k];
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_class[k] = a.class_value[k];
    }
    static_assert(kRescoreThreads >= 4 * kCertRows, "the table is filled by the workgroup's last threads");
    if (a.certify && tid >= kRescoreThreads - 4 * kCertRows) {     // (read behind the barriers of the first candidate's passes)
        const int idx = tid - (kRescoreThreads - 4 * kCertRows), c = idx & 3;
        const double value = c == 0 ? a.class_value[0] : c == 1 ? a.class_value[1] : c == 2 ? a.class_value[2] : a.class_value[3];
        cert_fill_table(&s_tab, value, a.fast_min_exp, idx);
    }
    __syncthreads();
    const double final_best = key_order(*a.global_best);
    PSA_RPHASE(1);
    for (int ci = blockIdx.x; ci < count; ci += gridDim.x) {
    // candidates were picked against a running best; drop those the final best rules out
    const bool pruned = a.cand_upper != nullptr && a.cand_upper[ci] < final_best;
    if (!pruned) {
    const int o = a.cand[ci];
    const unsigned char* s1 = a.s1 + o;
    const int m = a.m;

    double s = 0.0;                 // running sum (meaningful in wave 0), wave-uniform
    unsigned long long key = 0ull;  // per thread: max of (rank << 32 | ~i): best rank, then lowest i
    WalkStats st = {0, 0, 0};

    if (a.certify && m <= kRescoreChunk) {
        // ---- the whole Seq2 is staged: classification + certificate in the passes of finish_search (one candidate),
        // the certified sum on wave 0, the head chain on the last wave beside level 2
        static_assert(kRescoreChunk < 65536, "cert_layout");
        const CertLayout L = cert_layout(m);
        CertHead* head = reinterpret_cast<CertHead*>(s_cert);
        {
            unsigned* bitmap = reinterpret_cast<unsigned*>(s_cert + L.bitmap);
            const int words = (L.summary - L.bitmap) / 4 + kCertSummary;
            for (int w = tid; w < words; w += kRescoreThreads) bitmap[w] = 0u;
            if (tid == 0) head->nfail = 0;
        }
        const unsigned long long incl = cert_classify<kRescoreThreads, false>(
            s1, a.s2, s_clsrank, s_codes, m, reinterpret_cast<unsigned long long*>(s_cert + L.seg_before),
            reinterpret_cast<unsigned*>(s_cert + L.segbits), tid, key, 0, 0);
        if (lane == 63) s_wsum[tid >> 6] = incl;
        __syncthreads();
        if (ci == (int)blockIdx.x) PSA_RPHASE(2);
        {
            constexpr int kWaves = kRescoreThreads / 64;
            const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            const unsigned long long tot = lane < kWaves ? s_wsum[lane] : 0ull;
            const unsigned long long scan = (unsigned long long)wave_scan_inclusive((long long)tot);
            const unsigned long long before = (unsigned long long)readlane64((long long)(scan - tot), wave);
            const unsigned long long all = (unsigned long long)readlane64((long long)scan, kWaves - 1);
            if (tid == 0) head->total = all;
            cert_level1<kRescoreThreads>(s_codes, m, s_class, a.walk_slack, a.fast_min_exp, s_cert, before, tid);
        }
        __syncthreads();
        cert_level2<kRescoreThreads>(s_codes, m, s_class, a.walk_slack, a.fast_min_exp, s_cert, tid);
        if ((tid >> 6) == kRescoreThreads / 64 - 1) {
            const int head_n = m < kCertChain ? m : kCertChain;
            const double hs = chain64(0.0, lane < head_n ? s_class[s_codes[lane] & 7u] : 0.0);
            if (lane == 0) head->head_sum = hs;
        }
        __syncthreads();
        if (tid < 64) {
            if (!cert_sum(s_codes, m, s_class, s_cert, &s_tab, lane, s)) {
                st.left += 1000;                        // (phase-timing builds: the walk was needed)
                s = walk_sum(0.0, s_codes, 0, m, s_class, &s_walk, a.fast_ok, a.fast_min_exp, lane, st);
            }
        }
        if (ci == (int)blockIdx.x) PSA_RPHASE(3);
        __syncthreads();            // the next candidate overwrites the staged codes
    } else
    for (int c0 = 0; c0 < m; c0 += kRescoreChunk) {
        const int c1 = c0 + kRescoreChunk < m ? c0 + kRescoreChunk : m;
        classify_chars(s1, a.s2, s_clsrank, s_codes, c0, c1, tid, kRescoreThreads, key);
        __syncthreads();
        if (ci == (int)blockIdx.x) PSA_RPHASE(2);
        // ---- sum the chunk in reference order (wave 0): the integer walk, the plain chain where that does not apply
        if (tid < 64) {
            st.left += 1000;
            s = walk_sum(s, s_codes, c0, c1, s_class, &s_walk, a.fast_ok, a.fast_min_exp, lane, st);
        }
        if (ci == (int)blockIdx.x) PSA_RPHASE(3);
        __syncthreads();            // the next chunk overwrites s_codes
    }
#ifdef PSA_PHASE_TIMING
    if (tid == 0 && blockIdx.x == 0 && ci == 0) {      // how the walk of the first candidate went
        g_phase_ts[1001][0] = (unsigned long long)st.chain;
        g_phase_ts[1001][1] = (unsigned long long)st.batches;
        g_phase_ts[1001][2] = (unsigned long long)st.left;
    }
#endif

    // best rank and its first position over all chars
    key = wave_max_u64(key);
    if (lane == 0) s_key[tid >> 6] = key;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kRescoreThreads / 64; ++w) key = s_key[w] > key ? s_key[w] : key;
        const int rank = (int)(key >> 32);
        CandRec r;
        r.offset = o;
        r.rank = rank;
        if (rank) {
            const int i = (int)(0xffffffffu - (unsigned)(key & 0xffffffffu));
            r.score = s + a.delta_of_rank[rank];
            r.char_offset = i;
            r.ch = a.sub[(unsigned)a.s2[i] * kRow + s1[i]];
        } else {
            r.score = a.delta_of_rank[0];
            r.char_offset = -1;
            r.ch = 0;
        }
        a.rec[ci] = r;
    }
    } else if (tid == 0) {
        CandRec r;
        r.score = 0.0; r.offset = -1; r.char_offset = -1; r.rank = 0; r.ch = 0;     // rank 0: ignored by the fold
        a.rec[ci] = r;
    }
    if (ci == (int)blockIdx.x) PSA_RPHASE(4);
    // ---- the last candidate workgroup to finish folds all records (is_swapable order).
    // Hand-off per the release/acquire rule: record store -> agent-scope release ->
    // relaxed atomic arrival; the last arriver acquires before reading the records.
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int arrived = __hip_atomic_fetch_add(a.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = arrived == count - 1;
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (s_last && tid < 64) {
        const bool is_max = a.is_max != 0;
        double bs = is_max ? -INFINITY : INFINITY;
        int bo = INT_MAX, idx = -1;
        for (int k = lane; k < count; k += 64) {
            const CandRec b = a.rec[k];
            if (b.rank != 0 && rec_better(is_max, b.score, b.offset, bs, bo)) { bs = b.score; bo = b.offset; idx = k; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double s2 = __shfl_xor(bs, d, 64);
            const int o2 = __shfl_xor(bo, d, 64);
            const int i2 = __shfl_xor(idx, d, 64);
            if (rec_better(is_max, s2, o2, bs, bo)) { bs = s2; bo = o2; idx = i2; }
        }
        if (lane == 0) {
            if (idx >= 0) {
                const CandRec b = a.rec[idx];
                write_search_tail(a, &b, 0);
            } else {
                write_search_tail(a, nullptr, 0);
            }
            // leave the search scalars clean for the next search on this slot
            *a.cand_count = 0;
            *a.done = 0;
            *a.global_best = 0ull;
        }
    }
    __syncthreads();            // the next candidate reuses the shared buffers
    if (ci == (int)blockIdx.x) PSA_RPHASE(5);
    }
}

}  // namespace psa
