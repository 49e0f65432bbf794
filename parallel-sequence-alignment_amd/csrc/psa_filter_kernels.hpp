// psa_filter_kernels.hpp — the "filter" engine (included by psa_device.hip).
//
// Idea.  A pair's score depends only on its sign class, so the
// order-independent ("mathematical") score of an offset is fixed by three
// integers: how many of its m pairs are '*', ':' and '.' (cpu_funcs.c:277,
// cuda_funcs.cu:442-452).  Those counts are computed EXACTLY, for every
// offset, with bit-sliced arithmetic.  The class of (Seq1[j], b) is stored as
// two bit-planes over the Seq1 positions j, for each of the 27 symbols b:
//
//   lo[b][j] = class is '*' or '.'        hi[b][j] = class is ':' or '.'
//   n_lo(o)   = sum_i lo[Seq2[i]][o+i]    (= n_star + n_dot)
//   n_hi(o)   = sum_i hi[Seq2[i]][o+i]    (= n_colon + n_dot)
//   n_both(o) = sum_i (lo & hi)[...]      (= n_dot)
//
// One lane owns 64 offsets (two 32-bit words); for char i the planes of
// Seq2[i] are funnel-shifted by i and added into per-lane bit-plane counters
// with carry-save adders built from v_bitop3_b32 (xor3 / majority: 2 full-rate
// VALU ops per 32 pairs and counter).  From the counts, a rounding bound of
// the reference's sequential fp64 sum (cpu_funcs.c:278) gives a score interval
// per offset; only offsets whose interval can reach the best one are re-scored
// in the reference's exact order (psa_rescore_kernel).  The answer is the
// sequential engine's, bit for bit, for any input; inputs where too many
// offsets tie (homopolymers) fall back to the sequential engine.
#pragma once

#include <utility>

namespace psa {

constexpr int kTile = 4096;        // offsets per workgroup: 64 lanes x 2 words x 32 bits
constexpr int kUnroll = 32;        // chars per carry-save round (2^kLevels)
constexpr int kLevels = 5;
constexpr int kPlanes = 10;        // per-wave counters hold up to 1023 chars
constexpr int kMaxWaveChars = 992; // multiple of kUnroll below 2^kPlanes
constexpr int kMergedPlanes = 15;  // 16 waves x 1023 < 2^15
constexpr int kMaskRows = 28;      // 27 symbols + the all-zero row of the padding symbol
constexpr int kCounters = 3;       // n_lo, n_hi, n_both

// ---- carry-save arithmetic on 32 independent bit lanes ----------------------
// v_bitop3_b32 (gfx950): any 3-input boolean function, full VALU rate
// (v_bfi_b32 and v_alignbit_b32 issue at half rate: tools/ubench/valu_rate.hip).
__device__ __forceinline__ unsigned xor3(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned maj3(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe8" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
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
struct SlicedCounter {
    unsigned plane[kPlanes];
    unsigned pend[kLevels];

    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int p = 0; p < kPlanes; ++p) plane[p] = 0u;
    }

    // T = index of the char inside the round, L = level (both compile-time so the
    // whole carry-save tree is straight-line code)
    template <int L, int T>
    __device__ __forceinline__ void push_level(unsigned v)
    {
        if constexpr (L == kLevels) {
            // one carry of weight 2^kLevels per round: ripple into the upper planes
#pragma unroll
            for (int p = kLevels; p < kPlanes - 1; ++p) {
                const unsigned c = plane[p] & v;
                plane[p] ^= v;
                v = c;
            }
            plane[kPlanes - 1] ^= v;
        } else if constexpr ((T & 1) == 0) {
            pend[L] = v;
        } else {
            push_level<L + 1, (T >> 1)>(csa(plane[L], pend[L], v));
        }
    }

    template <int T>
    __device__ __forceinline__ void push(unsigned v) { push_level<0, T>(v); }
};

// ---- Seq1 class planes --------------------------------------------------------
// letter_masks[b*2 + 0] = 27-bit set of Seq1 symbols a with class(a, b) in {'*', '.'}
// letter_masks[b*2 + 1] = ... with class(a, b) in {':', '.'}
struct MaskArgs {
    const unsigned char* s1;          // codes, padded with kCodePad to 32*nslots
    uint2*               masks;       // [kMaskRows][nslots]: {lo, hi} of 32 Seq1 positions
    const unsigned*      letter_masks;  // [kMaskRows*2]
    int                  nslots;
};

__global__ __launch_bounds__(256) void psa_build_masks_kernel(MaskArgs a)
{
    __shared__ unsigned s_letter[kMaskRows * 2];
    if (threadIdx.x < kMaskRows * 2) s_letter[threadIdx.x] = a.letter_masks[threadIdx.x];
    __syncthreads();
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= a.nslots) return;
    const uint4* src = reinterpret_cast<const uint4*>(a.s1 + (size_t)slot * 32);
    const uint4 c0 = src[0], c1 = src[1];
    const unsigned w[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    // one-hot of each position's symbol, then every symbol row is a 32-way bit gather
    for (int b = 0; b < kMaskRows; ++b) {
        const unsigned klo = s_letter[b * 2 + 0], khi = s_letter[b * 2 + 1];
        unsigned lo = 0u, hi = 0u;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const unsigned code = (w[t >> 2] >> (8 * (t & 3))) & 0xffu;   // 0..27
            lo |= ((klo >> code) & 1u) << t;
            hi |= ((khi >> code) & 1u) << t;
        }
        a.masks[(size_t)b * a.nslots + slot] = make_uint2(lo, hi);
    }
}

// One char of the count loop: J is its index inside the round.  Tiles start on
// a 32-offset boundary and rounds on a 32-char boundary, so the funnel shift of
// char J is the compile-time constant J and all chars of a round read the same
// slot column.  Lane l owns words l and l+64 of the tile: consecutive lanes read
// consecutive 8-byte slots (conflict-free ds_read_b64).
template <int J>
__device__ __forceinline__ void count_step(SlicedCounter (&cnt)[2][kCounters],
                                           const unsigned (&packed)[kUnroll / 4], const char* lds_lane,
                                           int row_bytes)
{
    const unsigned b = (packed[J >> 2] >> (8 * (J & 3))) & 0xffu;   // Seq2 symbol: wave-uniform
    const uint2* src = reinterpret_cast<const uint2*>(lds_lane + b * row_bytes);
    unsigned lo0, hi0, lo1, hi1;
    if constexpr (J == 0) {
        const uint2 s0 = src[0], s2 = src[64];
        lo0 = s0.x; hi0 = s0.y; lo1 = s2.x; hi1 = s2.y;
    } else {
        const uint2 s0 = src[0], s1 = src[1], s2 = src[64], s3 = src[65];
        lo0 = __builtin_amdgcn_alignbit(s1.x, s0.x, J);
        hi0 = __builtin_amdgcn_alignbit(s1.y, s0.y, J);
        lo1 = __builtin_amdgcn_alignbit(s3.x, s2.x, J);
        hi1 = __builtin_amdgcn_alignbit(s3.y, s2.y, J);
    }
    cnt[0][0].template push<J>(lo0);
    cnt[0][1].template push<J>(hi0);
    cnt[0][2].template push<J>(lo0 & hi0);
    cnt[1][0].template push<J>(lo1);
    cnt[1][1].template push<J>(hi1);
    cnt[1][2].template push<J>(lo1 & hi1);
}

template <int... Js>
__device__ __forceinline__ void count_round(SlicedCounter (&cnt)[2][kCounters],
                                            const unsigned (&packed)[kUnroll / 4], const char* lds_lane,
                                            int row_bytes, std::integer_sequence<int, Js...>)
{
    (count_step<Js>(cnt, packed, lds_lane, row_bytes), ...);
}

// ---- count kernel --------------------------------------------------------------
struct CountArgs {
    const uint2*         masks;     // [kMaskRows][nslots]
    const unsigned char* s2;        // codes of Seq2 as bytes, padded with kCodePad
    unsigned long long*  partial;   // [nchunks][span_pad]: n_lo | n_hi<<21 | n_both<<42
    int nslots;
    int base;         // first offset of tile 0: the range's first offset rounded down to a multiple of 32
    int span_pad;     // offsets covered = ntiles*kTile
    int chunk_chars;  // chars per chunk (multiple of kUnroll * waves)
    int wave_chars;   // chars per wave   (multiple of kUnroll, <= kMaxWaveChars)
    int lds_slots;    // slots per staged row
};

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void psa_count_kernel(CountArgs a)
{
    extern __shared__ uint2 lds[];                  // [kMaskRows][lds_slots]; reused by the epilogue
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile_o0 = a.base + blockIdx.x * kTile;           // absolute offset of word 0, bit 0 (multiple of 32)
    const int chunk_c0 = blockIdx.y * a.chunk_chars;           // first char of this chunk (multiple of 32)
    const int slot0 = (tile_o0 + chunk_c0) >> 5;               // first staged slot

    // stage the 28 plane rows of the window [tile_o0+chunk_c0, +kTile+chunk_chars): rows dealt to waves
    for (int b = wave; b < kMaskRows; b += WAVES) {
        const uint2* src = a.masks + (size_t)b * a.nslots + slot0;
        uint2* dst = lds + b * a.lds_slots;
        for (int s = lane; s < a.lds_slots; s += 64) dst[s] = src[s];
    }
    __syncthreads();

    SlicedCounter cnt[2][kCounters];     // [word][n_lo, n_hi, n_both]
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int k = 0; k < kCounters; ++k) cnt[w][k].clear();

    const int c_begin = chunk_c0 + wave * a.wave_chars;
    const unsigned* codes32 = reinterpret_cast<const unsigned*>(a.s2 + c_begin);
    const int row_bytes = a.lds_slots * 8;
    // slot column of this wave's first round; one column further per round
    const char* lds_lane = reinterpret_cast<const char*>(lds) + lane * 8 + (size_t)((c_begin - chunk_c0) >> 5) * 8;

    for (int g = 0; g < a.wave_chars; g += kUnroll) {
        unsigned packed[kUnroll / 4];
#pragma unroll
        for (int q = 0; q < kUnroll / 4; ++q) packed[q] = codes32[(g >> 2) + q];   // wave-uniform
        count_round(cnt, packed, lds_lane, row_bytes, std::make_integer_sequence<int, kUnroll>{});
        lds_lane += 8;
    }

    // ---- epilogue: sum the WAVES partial counters (bit-sliced ripple adds through LDS),
    // then expand the merged planes to integers and store them.
    __syncthreads();                                 // rows no longer needed: LDS is scratch now
    unsigned* scratch = reinterpret_cast<unsigned*>(lds);
    // merged[word][counter][plane]: grows by one plane per round
    unsigned merged[2][kCounters][kMergedPlanes];
#pragma unroll
    for (int w = 0; w < 2; ++w)
#pragma unroll
        for (int k = 0; k < kCounters; ++k)
#pragma unroll
            for (int p = 0; p < kMergedPlanes; ++p) merged[w][k][p] = p < kPlanes ? cnt[w][k].plane[p] : 0u;

    // Round r: the upper half of the surviving waves hands its counters (kPlanes + r
    // planes each) to the lower half, which adds them with a bit-sliced ripple adder.
    {
        int r = 0;
#pragma unroll
        for (int half = WAVES / 2; half >= 1; half >>= 1, ++r) {
            const int np = kPlanes + r;                 // planes in flight this round
            const int rec = 2 * kCounters * np;         // dwords per lane in the exchange buffer
            if (wave >= half && wave < 2 * half) {
                unsigned* dst = scratch + (size_t)(wave - half) * rec * 64 + lane;
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int k = 0; k < kCounters; ++k)
#pragma unroll
                        for (int p = 0; p < kMergedPlanes; ++p)
                            if (p < np) dst[((w * kCounters + k) * np + p) * 64] = merged[w][k][p];
            }
            __syncthreads();
            if (wave < half) {
                const unsigned* src = scratch + (size_t)wave * rec * 64 + lane;
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int k = 0; k < kCounters; ++k) {
                        unsigned carry = 0u;
#pragma unroll
                        for (int p = 0; p < kMergedPlanes; ++p)
                            if (p < np) {
                                const unsigned other = src[((w * kCounters + k) * np + p) * 64];
                                const unsigned x = merged[w][k][p];
                                merged[w][k][p] = xor3(x, other, carry);
                                carry = maj3(x, other, carry);
                            } else if (p == np) {
                                merged[w][k][p] = carry;
                            }
                    }
            }
            __syncthreads();
        }
    }

    // wave 0 publishes the merged planes: layout [counter][plane][word(128)]
    unsigned* pub = scratch;
    if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int k = 0; k < kCounters; ++k)
#pragma unroll
                for (int p = 0; p < kMergedPlanes; ++p) pub[(k * kMergedPlanes + p) * 128 + w * 64 + lane] = merged[w][k][p];
    }
    __syncthreads();

    // expand: each thread handles kTile / (WAVES*64) consecutive offsets of one word
    constexpr int kPer = kTile / (WAVES * 64);
    static_assert(kPer >= 1 && 32 % kPer == 0, "tile/threads");
    const int o_local = tid * kPer;
    const int word = o_local >> 5;
    const int bit0 = o_local & 31;
    unsigned long long out[kPer];
#pragma unroll
    for (int e = 0; e < kPer; ++e) out[e] = 0ull;
#pragma unroll
    for (int k = 0; k < kCounters; ++k) {
        unsigned n[kPer];
#pragma unroll
        for (int e = 0; e < kPer; ++e) n[e] = 0u;
#pragma unroll
        for (int p = 0; p < kMergedPlanes; ++p) {
            const unsigned v = pub[(k * kMergedPlanes + p) * 128 + word] >> bit0;
#pragma unroll
            for (int e = 0; e < kPer; ++e) n[e] |= ((v >> e) & 1u) << p;
        }
#pragma unroll
        for (int e = 0; e < kPer; ++e) out[e] |= (unsigned long long)n[e] << (21 * k);
    }
    unsigned long long* dst = a.partial + (size_t)blockIdx.y * a.span_pad + (size_t)blockIdx.x * kTile + o_local;
#pragma unroll
    for (int e = 0; e < kPer; ++e) dst[e] = out[e];
}

// ---- keys: counts -> score interval per offset; global best bound --------------
struct KeyArgs {
    const unsigned long long* partial;   // [nchunks][span_pad]
    double*                   upper;     // [span]: optimistic score of every offset
    unsigned long long*       block_best;  // [gridDim.x]: ordered-u64 of each block's best pessimistic score
    double w[4];
    double class_best[4];    // most favourable delta inside each class
    double class_worst[4];   // least favourable delta inside each class
    double slack;            // rounding bound of the sequential sum + this formula
    int nchunks;
    int span_pad;
    int span;
    int lead;         // offsets of tile 0 in front of the range (first - base)
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

constexpr int kKeyBlocks = 1024;   // grid of the keys / select kernels (grid-stride over the offsets)

__global__ __launch_bounds__(256) void psa_keys_kernel(KeyArgs a)
{
    __shared__ unsigned long long s_best[4];
    const bool mx = a.is_max != 0;
    // in MIN mode everything is negated so that "best" is always the maximum
    const double sgn = mx ? 1.0 : -1.0;
    double pess = -INFINITY;
    for (int o = blockIdx.x * 256 + threadIdx.x; o < a.span; o += gridDim.x * 256) {
        unsigned long long sum = 0ull;
        for (int c = 0; c < a.nchunks; ++c) sum += a.partial[(size_t)c * a.span_pad + a.lead + o];
        const int n_lo = (int)(sum & 0x1fffff), n_hi = (int)((sum >> 21) & 0x1fffff), n2 = (int)((sum >> 42) & 0x1fffff);
        const int n0 = n_lo - n2, n1 = n_hi - n2;          // stars, colons; n2 = dots
        const int n3 = a.m - n0 - n1 - n2;
        const double total = ((a.w[0] * n0 - a.w[1] * n1) - a.w[2] * n2) - a.w[3] * n3;
        const int n[4] = {n0, n1, n2, n3};
        double d_opt = mx ? -INFINITY : INFINITY, d_pes = mx ? -INFINITY : INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (n[k] > 0) {
                d_opt = mx ? fmax(d_opt, a.class_best[k]) : fmin(d_opt, a.class_best[k]);
                d_pes = mx ? fmax(d_pes, a.class_worst[k]) : fmin(d_pes, a.class_worst[k]);
            }
        a.upper[o] = sgn * (total + d_opt) + a.slack;            // optimistic, in "bigger is better" form
        pess = fmax(pess, sgn * (total + d_pes) - a.slack);      // pessimistic
    }
    unsigned long long key = order_key(pess);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long other = __shfl_xor(key, d, 64);
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) key = s_best[w] > key ? s_best[w] : key;
        a.block_best[blockIdx.x] = key;
    }
}

// ---- select: offsets whose optimistic score reaches the best pessimistic one ----
struct SelectArgs {
    const double*             upper;
    const unsigned long long* block_best;   // [nbest] from the keys kernel
    int*                      cand;         // [cap]
    int*                      cand_count;
    int nbest;
    int span;
    int first;
    int cap;
};

__global__ __launch_bounds__(256) void psa_select_kernel(SelectArgs a)
{
    __shared__ unsigned long long s_best[4];
    unsigned long long key = 0ull;
    for (int k = threadIdx.x; k < a.nbest; k += 256) {
        const unsigned long long v = a.block_best[k];
        key = v > key ? v : key;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long other = __shfl_xor(key, d, 64);
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) s_best[threadIdx.x >> 6] = key;
    __syncthreads();
    key = s_best[0];
    for (int w = 1; w < 4; ++w) key = s_best[w] > key ? s_best[w] : key;
    const double need = key_order(key);
    for (int o = blockIdx.x * 256 + threadIdx.x; o < a.span; o += gridDim.x * 256) {
        if (a.upper[o] >= need) {
            const int slot = atomicAdd(a.cand_count, 1);
            if (slot < a.cap) a.cand[slot] = a.first + o;
        }
    }
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
struct RescoreArgs {
    const unsigned char* s1;
    const unsigned char* s2;        // byte codes
    const unsigned char* clsrank;   // [32][32]: class (bits 0-2) | delta rank (bits 3-7), [c2][c1]
    const double*        delta_of_rank;
    const int*           cand;
    const int*           cand_count;
    BlockRec*            rec;       // [cap]: exact score, offset, best rank
    double class_value[8];          // pair score of class 0..3, +0.0 for the padding class 4
    int m;
    int cap;
    int is_max;
    int fast_ok;                    // all weights normal doubles or zero
    int fast_min_exp;               // smallest biased exponent of the running sum the walk handles
};

constexpr int kRescoreEpl = 8;      // chars per lane and batch on the integer walk

__device__ __forceinline__ long long wave_scan_inclusive(long long v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__global__ __launch_bounds__(64) void psa_rescore_kernel(RescoreArgs a)
{
    __shared__ unsigned char s_clsrank[kRow * kRow];
    __shared__ double        s_vals[64];
    __shared__ long long     s_round[8];
    int count = *a.cand_count;
    if (count > a.cap) count = a.cap;
    if ((int)blockIdx.x >= count) return;
    const int lane = threadIdx.x;
    for (int k = lane; k < kRow * kRow; k += 64) s_clsrank[k] = a.clsrank[k];
    __syncthreads();
    const int o = a.cand[blockIdx.x];
    const unsigned char* s1 = a.s1 + o;
    const int m = a.m;

    double s = 0.0;             // running sum, wave-uniform
    unsigned best = 0u;         // per lane: max of (rank << 3 | class) over its chars
    int k = 0;
    while (k < m) {
        const unsigned long long sb = (unsigned long long)__double_as_longlong(s);
        const int e = (int)((sb >> 52) & 0x7ff);
        bool walk = a.fast_ok && e >= a.fast_min_exp && e < 0x7ff;
        if (walk) {
            // R_e[c] for the five classes, and whether any addend ties in this binade
            long long r = 0;
            bool tie = false;
            if (lane < 5) {
                const unsigned long long pb = (unsigned long long)__double_as_longlong(a.class_value[lane]);
                const int ep = (int)((pb >> 52) & 0x7ff);
                if (ep != 0) {                                   // zero stays 0 (subnormals excluded by fast_ok)
                    const unsigned long long mp = (pb & 0xfffffffffffffull) | (1ull << 52);
                    const int d = e - ep;                        // >= 2 by fast_min_exp
                    if (d <= 53) {
                        const unsigned long long half = 1ull << (d - 1);
                        tie = (mp & ((half << 1) - 1)) == half;
                        const long long mag = (long long)((mp + half) >> d);
                        r = (pb >> 63) ? -mag : mag;
                    }
                }
                s_round[lane] = r;
            }
            if (__ballot(tie) != 0ull) walk = false;
        }
        if (!walk) {
            // plain chain over the next <= 64 chars
            const int nb = m - k < 64 ? m - k : 64;
            double p = 0.0;
            if (lane < nb) {
                const unsigned cr = s_clsrank[(unsigned)a.s2[k + lane] * kRow + s1[k + lane]];
                best = cr > best ? cr : best;
                p = a.class_value[cr & 7u];
            }
            s_vals[lane] = p;                                     // lanes past the end hold +0.0: adding it is exact
            __builtin_amdgcn_wave_barrier();
            const double2* v = reinterpret_cast<const double2*>(s_vals);   // same address in every lane: broadcast
#pragma unroll
            for (int l = 0; l < 32; ++l) {
                const double2 x = v[l];
                s += x.x;                                         // reference order (cpu_funcs.c:278)
                s += x.y;
            }
            __builtin_amdgcn_wave_barrier();
            k += nb;
            continue;
        }
        // integer walk inside binade e
        __builtin_amdgcn_wave_barrier();
        long long S = (long long)((sb & 0xfffffffffffffull) | (1ull << 52));
        const bool neg = (sb >> 63) != 0;
        if (neg) S = -S;
        const long long lo = 1ll << 52, hi = 1ll << 53;
        bool left = false;
        while (k < m && !left) {
            const int base = k + lane * kRescoreEpl;
            long long rr[kRescoreEpl];
            long long t = 0;
#pragma unroll
            for (int j = 0; j < kRescoreEpl; ++j) {
                unsigned cls = 4u;
                if (base + j < m) {
                    const unsigned cr = s_clsrank[(unsigned)a.s2[base + j] * kRow + s1[base + j]];
                    best = cr > best ? cr : best;
                    cls = cr & 7u;
                }
                rr[j] = s_round[cls];
                t += rr[j];
            }
            const long long incl = wave_scan_inclusive(t, lane);
            long long q = S + (incl - t);                 // sum before this lane's first char
            long long before[kRescoreEpl];
            int bad = kRescoreEpl;
#pragma unroll
            for (int j = kRescoreEpl - 1; j >= 0; --j) before[j] = 0;
#pragma unroll
            for (int j = 0; j < kRescoreEpl; ++j) {
                before[j] = q;
                q += rr[j];
                const long long mag = neg ? -q : q;
                if (!(mag > lo && mag < hi) && bad == kRescoreEpl) bad = j;
            }
            const unsigned long long bad_lanes = __ballot(bad < kRescoreEpl);
            if (bad_lanes == 0ull) {
                S = __shfl(q, 63, 64);
                const int nb = m - k < 64 * kRescoreEpl ? m - k : 64 * kRescoreEpl;
                k += nb;
            } else {
                const int L = __ffsll((long long)bad_lanes) - 1;
                const int j = __shfl(bad, L, 64);
                long long prev = before[0];
#pragma unroll
                for (int jj = 1; jj < kRescoreEpl; ++jj) prev = jj == j ? before[jj] : prev;
                S = __shfl(prev, L, 64);                  // the sum before the char that leaves the binade
                k += L * kRescoreEpl + j;
                left = true;
            }
        }
        const unsigned long long mag = (unsigned long long)(S < 0 ? -S : S);
        s = __longlong_as_double((long long)((neg ? (1ull << 63) : 0ull) | ((unsigned long long)e << 52) |
                                             (mag & 0xfffffffffffffull)));
        if (left) {
            // this char moves the sum out of the binade: one true fp64 add (its class was
            // already folded into `best` by the lane that owned it)
            const unsigned cr = s_clsrank[(unsigned)a.s2[k] * kRow + s1[k]];
            s += a.class_value[cr & 7u];
            k += 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned other = __shfl_xor(best, d, 64);
        best = other > best ? other : best;
    }
    if (lane == 0) {
        const int rank = (int)(best >> 3);
        BlockRec r;
        r.score = rank ? s + a.delta_of_rank[rank] : a.delta_of_rank[0];
        r.offset = o;
        r.rank = rank;
        a.rec[blockIdx.x] = r;
    }
}

// ---- final fold of the candidate records (is_swapable order) + the winner's substitution
struct FoldArgs {
    const BlockRec*      rec;
    const int*           cand_count;
    const unsigned char* s1;
    const unsigned char* s2;        // byte codes
    const unsigned char* rank;      // [32][32]
    const unsigned char* sub;
    psa_result*          out;
    int*                 overflow;  // set when the candidate list did not fit
    int m;
    int cap;
    int is_max;
    long long pairs;
};

__global__ __launch_bounds__(256) void psa_fold_kernel(FoldArgs a)
{
    __shared__ BlockRec s_fold[4];
    __shared__ BlockRec s_win;
    __shared__ int s_first_i;
    const bool is_max = a.is_max != 0;
    int count = *a.cand_count;
    if (threadIdx.x == 0) *a.overflow = count > a.cap ? 1 : 0;
    if (count > a.cap) count = a.cap;
    double s = is_max ? -INFINITY : INFINITY;
    int o = INT_MAX, r = 0;
    for (int k = threadIdx.x; k < count; k += 256) {
        const BlockRec b = a.rec[k];
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
            a.out->score = is_max ? -INFINITY : INFINITY;
            a.out->offset = -1;
            a.out->char_offset = -1;
            a.out->ch = 0;
            a.out->shard = -1;
            a.out->pairs = a.pairs;
        }
        return;
    }
    // first char whose delta rank is the offset's best (cpu_funcs.c:287-294: strict compare keeps the lowest i)
    int mine = INT_MAX;
    for (int i = threadIdx.x; i < a.m; i += 256) {
        const unsigned idx = (unsigned)a.s2[i] * kRow + a.s1[w.offset + i];
        if (a.rank[idx] == w.rank) { mine = i; break; }
    }
    atomicMin(&s_first_i, mine);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int i = s_first_i;
        const unsigned idx = (unsigned)a.s2[i] * kRow + a.s1[w.offset + i];
        a.out->score = w.score;
        a.out->offset = w.offset;
        a.out->char_offset = i;
        a.out->ch = a.sub[idx];
        a.out->shard = -1;
        a.out->pairs = a.pairs;
    }
}

}  // namespace psa
