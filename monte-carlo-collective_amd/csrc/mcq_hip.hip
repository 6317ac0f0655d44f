// mcq_hip.hip -- hand-written HIP kernels (gfx950 / CDNA4) + the C-ABI of include/mcq.h.
//
// What it replaces: the reference's per-chain Metropolis sweep and its process-pool fan-out
//   metropolis_mcmc_board   experiments.py:282-376   (board chain)
//   metropolis_mcmc         experiments.py:199-279   (full_3d chain)
//   State3DQueensBoard      mcmc_board.py:5-193      (init modes, energy, conflicts_for_position)
//   State3DQueens           mcmc.py:5-226            (init modes, energy, conflicts_for_queen)
//   beta schedules          experiments.py:13-77
//   run_experiment fan-out  experiments.py:507-546   (chain r seeded with base_seed + r)
//
// Design (DESIGN.md has the long form).  A wavefront of 64 lanes is split into groups of G
// lanes (4, 8 or 16; default 4 for boards up to N = 12, 8 otherwise); one group runs one chain,
// so a wavefront advances 64/G chains in lockstep, one Metropolis step per loop iteration, one
// wavefront per workgroup, no barrier anywhere.  What is serial inside a chain (the NumPy-legacy
// MT19937 stream with its data-dependent word consumption, the proposal, the accept test) is
// computed redundantly by the G lanes of the group; the attack count and the stream generation
// are spread over the lanes.
//   * MT19937: the 624 raw state words of a chain stay in its record in global memory and are
//     regenerated in place in blocks of 16 (16/G per lane), requested one or more steps before
//     they are needed; the block is twisted, tempered and appended to a 64-slot ring of ready
//     words in LDS (its first 32 slots mirrored behind it), together with one flag per word that
//     says whether it passes the masked-rejection test of randint(0, N); the flags live in a
//     64-bit register pair indexed from the read position.  This upkeep is demand-driven and
//     runs for all chains of the wavefront together, so the lanes stay converged.
//   * proposal (board): the positions of the next five accepted words come from bit tricks on the
//     low word of the flags (no rejection loop, no divergence); i, j and three candidates for
//     new_k are fetched in one batch, the uniform's two words behind the candidate that is taken.
//     Anything unusual (ring nearly empty, all candidates equal to old_k, rejection run longer
//     than the view) takes a sequential fallback that draws word by word -- same stream, same results.
//   * dE (board): only columns on the row, column and two diagonals of (i,j) in the ij-plane
//     can attack cell (i,j,k), at most 4N of them; a column at in-plane distance d with height
//     h attacks iff |h-k| is 0 or d, i.e. iff bit h of (B | B<<d | B>>d), B = 1<<k, is set.
//     One lane per (direction, position) probe, old and new height packed into one register for
//     N <= 16, then a DPP all-reduce over the group.  full_3d: the same with one occupancy word
//     per column and popcounts.
//   * accept: u < exp(-beta dE) is decided in float32 from the top 27 bits of u whenever u is
//     outside a 2^-10 relative bracket around the float32 estimate; inside the bracket the
//     float64 exp and the 53-bit u decide, so every decision equals the all-float64 decision.
//     (MCQ_FLAG_EXACT_EXP disables the bracket.)
//   * beta(step) is evaluated on the device in float64 by a small kernel into a table that the
//     sweep reads with scalar loads; strict IEEE (this file is built with -ffp-contract=off).
//   * energy_history: each chain stages 16 entries in LDS and stores them as one aligned 64-byte
//     segment; accept bits are flushed as 32-bit words.
//   * pacing: the wavefronts of a SIMD compare their progress through a small table and set
//     s_setprio so that they finish together (see set_priority below).
//   * the step is bound by instruction issue (four wavefronts per SIMD): the hot
//     loop is written for few instructions per step, and the sizes of BASELINE's configs (N = 12,
//     N = 24) have instantiations with N as a compile-time constant.
//   * one chain that leaves the batched draw stalls the 15 (7) that share its wavefront, so the rare paths are kept
//     short: the word-by-word draw goes on from what the batched attempt established (i, j and old_k; q and the
//     examined triples) instead of starting over, and boards up to N = 5 look at five candidates for new_k.
//   * beyond the reference's drivers: full_3d with Q != N^2 queens (mcq_params.n_queens), boards up to N = 128
//     (compare-based probes beyond 32), and -- not modes of the reference, never defaults -- the Philox stream and
//     replica exchange between the chains of a wavefront (mcq_params.exchange_every).
//
// Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <vector>

#include "../../include/mcq.h"

extern "C" int32_t mcq_default_lanes(int32_t mode);
extern "C" int32_t mcq_default_lanes_n(int32_t mode, int32_t N);

namespace {

// Diagnostic build (-DMCQ_STAMPS, tools/stamp_profile.sh): s_memtime stamps around the sections of a Metropolis
// step; the per-section cycle sums of every wavefront go to a debug buffer.  Never defined in the shipped library.
// Diagnostic build (-DMCQ_WAVE_TIMES): start / end time (100 MHz s_memrealtime) and placement of every wavefront of the sweep.
#ifdef MCQ_WAVE_TIMES
#define WAVE_T0 const unsigned long long wt0 = __builtin_amdgcn_s_memrealtime()
#define WAVE_T1(buf)                                                                                         \
    do {                                                                                                     \
        if (threadIdx.x == 0 && (buf)) {                                                                     \
            unsigned long long* r = (buf) + 4ull * blockIdx.x;                                               \
            r[0] = wt0, r[1] = __builtin_amdgcn_s_memrealtime();                                             \
            r[2] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20), r[3] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); \
        }                                                                                                    \
    } while (0)
#else
#define WAVE_T0
#define WAVE_T1(buf)
#endif

#ifdef MCQ_STAMPS
#define STAMP_DECL unsigned long long st_prev = __builtin_amdgcn_s_memtime(), st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k)                                                        \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        const unsigned long long st_now = __builtin_amdgcn_s_memtime(); \
        st_acc[k] += st_now - st_prev;                                  \
        st_prev = st_now;                                               \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)
#define STAMP_FLUSH(dbg)                                                      \
    do {                                                                      \
        if (threadIdx.x == 0)                                                 \
            for (int k_ = 0; k_ < 12; k_++) atomicAdd(&(dbg)[k_], st_acc[k_]); \
    } while (0)
#define STAMP_COUNT(k) (st_acc[k] += 1)
#else
#define STAMP_DECL
#define STAMP(k)
#define STAMP_FLUSH(dbg)
#define STAMP_COUNT(k)
#endif

constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int RING = 64;          // ready (tempered) words per chain
constexpr int RING_MIRROR = 32;   // slots 0..31 are repeated behind the ring: a proposal reads up to 31 slots past pos without wrapping
#ifndef MCQ_RED_STRIPES
#define MCQ_RED_STRIPES 8
#endif
constexpr int RED_STRIPES = MCQ_RED_STRIPES;  // trace == REDUCED: independent accumulator copies, so one address sees few atomics
constexpr int REC_MIRROR = 624;   // record word: copy of MT word 0, so that words i+1 and i+397.. of a block never wrap inside a lane's run
constexpr int REC_POS = 625;      // record word: MT index of the next word to consume
constexpr int REC_GEN_END = 626;  // record word: words [0, gen_end) belong to the current generation
constexpr int REC_E0 = 627;       // record word: initial energy
constexpr int REC_STATE = 628;    // first word of the state bytes (heights or (i,j,k) triplets)

struct KArgs {
    int N, Q, mode, init, sched, rng;  // Q queens: N * N, or mcq_params.n_queens (full_3d, random init)
    int NN;                 // N * N: columns of the board
    unsigned flags;
    unsigned maskN, maskQ;  // smallest 2^b - 1 >= N-1 / Q-1 (masked rejection)
    int klarner_M;          // 0: exact Klarner (gcd(N,210)==1); else core edge M
    int state_bytes;
    int rec_words;          // words per chain record in the workspace
    int chain_lds_words;    // words of LDS per chain in the sweep kernel
    long long chains_per_set;  // schedule sets: chains per set (0: one set)
    long long tab_stride;      // schedule sets: elements between the tables of consecutive sets (beta_tab and c32_tab alike)
    long long red_set_stride;  // schedule sets, trace == REDUCED: accumulator words per set
    uint32_t* pace;         // progress table: one row of 16 words per SIMD of the device, word = step reached by the wavefront in that slot
    int full_pad;           // full_3d: spare column words on either side of the column table (>= N-1: out-of-board diagonal probes)
    double beta_const, beta_start, beta_end;
    long long n_steps, n_chains, patience, hist_stride, bits_stride;
    uint32_t* ws;           // chain records
    double* beta_tab;       // [n_steps] beta(step), filled by mcq_beta_kernel
    float* c32_tab;         // [n_steps] (float)(-beta(step) * log2(e)): exp(-beta dE) = exp2(dE * c32)
    const uint32_t* seeds;
    mcq_outputs out;
    unsigned long long* red;  // trace == REDUCED: [RED_STRIPES][3][red_len] per-entry sums (E, E^2, accepted | chains << 32)
    long long red_len;
    unsigned long long* dbg;  // MCQ_STAMPS diagnostic build only: per-section cycle sums
    long long exch_every;     // replica exchange: period in steps (0: off)
    int exch_R;               // rungs of a ladder
    int low_water;            // stream upkeep runs when some chain of the wavefront holds fewer ready words than this
    int init_words;           // init kernel: LDS words per chain
    const double* exch_ladder;  // [exch_R] beta multipliers per rung (workspace)
    int dry;                  // launch_sweep: check that the chosen variant fits the device and return without launching
    uint16_t* qtab;           // full_3d: the queens of every chain as i | j << 5 | k << 10, [n_chains][qtab_stride] (workspace; the sweep variants that
    int qtab_stride;          // keep their queen table out of LDS work on it, the init kernel fills it).  Beyond N = 32: uint32 entries i | j << 8 | k << 16
    uint32_t* perm;           // full_3d beyond N = 32, random init: the N^3 cells np.random.choice permutes, one slice per chain of an init launch (workspace)
    long long chain0;         // init kernel: first chain of this launch (the launches of one run share the `perm` slices)
    const uint32_t* stream;   // mcq_params.stream_states in the layout the kernels stream from, [n_chains][626]: MT words, position, words of the current generation (workspace)
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// One word of the next MT19937 generation.  Word i depends on words i, i+1 and (i+397) mod 624,
// the last taken from the NEW generation when i >= 227 and the middle one when i == 623;
// regenerating blocks of <= 64 consecutive words in increasing order on demand therefore yields
// exactly the words of the all-at-once twist (NumPy: _mt19937/mt19937.c, mt19937_gen) -- every
// lane of a block reads its three inputs before any lane of the block writes.
__device__ __forceinline__ uint32_t mt_twist_word(const uint32_t* mt, int i) {
    const uint32_t a = mt[i];
    const uint32_t b = mt[i + 1 == MT_N ? 0 : i + 1];
    const uint32_t c = mt[i + MT_M >= MT_N ? i + MT_M - MT_N : i + MT_M];
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ unsigned mask_for(unsigned m) {
    unsigned mask = m;
    mask |= mask >> 1, mask |= mask >> 2, mask |= mask >> 4, mask |= mask >> 8, mask |= mask >> 16;
    return mask;
}

// Philox-4x32-10 (Salmon et al., SC'11) with counter (c0, c1, 0, 0) and key (k0, 0): the block function of
// mcq_params.rng == MCQ_RNG_PHILOX4X32_10 (include/mcq.h); oracle/mcq_oracle.c holds the same function and its known answers.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t (&out)[4]) {
    uint32_t c2 = 0u, c3 = 0u, k1 = 0u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

// Stream used by the init kernel: L lanes per chain (64 / L chains per wavefront), the current L-word block kept in a
// register per lane, one ds_bpermute per draw.
struct InitRng {
    uint32_t* mt;
    uint32_t win;
    int pos, gen_end, sub, gbase, L;  // sub: lane inside the chain's group; gbase: first lane of the group
    int wraps;                        // generations finished (mcq_outputs.stream_words)
    bool philox;
    uint32_t key;

    __device__ __forceinline__ void fill(int base) {
        if (philox) {  // word base + sub of the chain's Philox stream (an init draws far fewer than 2^32 words)
            uint32_t o[4];
            const uint32_t w = (uint32_t)(base + sub);
            philox4x32_10(w >> 2, 0u, key, o);
            win = (w & 2u) ? ((w & 1u) ? o[3] : o[2]) : ((w & 1u) ? o[1] : o[0]);
            return;
        }
        const int i = base + sub;
        if (i < MT_N) {
            uint32_t v;
            if (base < gen_end) {
                v = mt[i];
            } else {
                v = mt_twist_word(mt, i);
                mt[i] = v;
            }
            win = mt_temper(v);
        }
        if (base >= gen_end) gen_end = base + L > MT_N ? MT_N : base + L;
    }
    __device__ __forceinline__ uint32_t next() {
        const int off = pos & (L - 1);
        if (off == 0) fill(pos);
        const uint32_t w = (uint32_t)__shfl((int)win, gbase + off, 64);
        pos++;
        if (!philox && pos == MT_N) pos = 0, gen_end = 0, wraps++;
        return w;
    }
    // RandomState.randint(0, m + 1) / shuffle's random_interval: masked rejection on 32-bit words;
    // m == 0 consumes nothing.
    __device__ __forceinline__ int bounded(unsigned m, unsigned mask) {
        if (m == 0) return 0;
        unsigned v;
        do {
            v = next() & mask;
        } while (v > m);
        return (int)v;
    }
};

// experiments.py:13-77, evaluation order kept, float64, no contraction.
__device__ double beta_at(const KArgs& a, long long step) {
    const double bs = a.beta_start, be = a.beta_end;
    const long long n = a.n_steps;
    switch (a.sched) {
    case MCQ_SCHED_CONSTANT:
        return a.beta_const;
    case MCQ_SCHED_LINEAR: {
        if (n <= 1) return be;
        const double frac = (double)step / (double)(n - 1);
        return bs + frac * (be - bs);
    }
    case MCQ_SCHED_EXPONENTIAL: {
        if (n <= 1) return be;
        const double log_ratio = log(be / bs);
        const long long c = step < 0 ? 0 : (step > n - 1 ? n - 1 : step);
        const double t = (double)c / (double)(n - 1);
        return bs * exp(log_ratio * t);
    }
    case MCQ_SCHED_LOGARITHMIC: {
        if (n <= 1) return be;
        const double log_norm = log((double)(1 + n));
        const long long c = step < 0 ? 0 : (step > n ? n : step);
        return bs + (be - bs) * (log((double)(1 + c)) / log_norm);
    }
    default: {
        if (n <= 1) return be;
        const long long c = step < 0 ? 0 : (step > n ? n : step);
        const double x = 3.141592653589793 * (double)c / (double)n;
        return bs + (be - bs) * (1.0 - cos(x)) / 2.0;
    }
    }
}

__global__ __launch_bounds__(256) void mcq_beta_kernel(KArgs a) {
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    if (s < a.n_steps) {
        const double b = beta_at(a, s);
        a.beta_tab[s] = b;
        if (a.c32_tab) a.c32_tab[s] = (float)(-b * 1.4426950408889634);
    }
}

// caller-supplied beta values (mcq_params.beta_table): copy into the workspace table and derive the float32 factor
__global__ __launch_bounds__(256) void mcq_beta_copy_kernel(const double* __restrict__ src, double* __restrict__ beta_tab, float* __restrict__ c32_tab,
                                                            long long n_steps) {
    const long long s = (long long)blockIdx.x * 256 + threadIdx.x;
    if (s < n_steps) {
        const double b = src[s];
        beta_tab[s] = b;
        c32_tab[s] = (float)(-b * 1.4426950408889634);
    }
}

// sum over the 64 lanes of a wavefront, result in every lane
__device__ __forceinline__ int wave_sum(int v) {
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// index of the lowest set bit, -1 for zero, in one instruction (what __ffs(x) - 1 means, without the zero test the
// compiler wraps around it)
__device__ __forceinline__ int lowest_bit(uint32_t x) {
    int p;
    asm("v_ffbl_b32 %0, %1" : "=v"(p) : "v"(x));
    return p;
}

// |a - b| of two small non-negative integers in one instruction
__device__ __forceinline__ uint32_t abs_diff(int a, int b) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// x * N for a coordinate x; with a compile-time N (NC != 0) through the 24-bit multiply with an inline constant
template <int NC>
__device__ __forceinline__ int times_N(int x, int N) {
    if constexpr (NC != 0) {
        uint32_t d;
        asm("v_mul_u32_u24 %0, %1, %2" : "=v"(d) : "v"(x), "n"(NC));
        return (int)d;
    } else {
        return __mul24(x, N);
    }
}

// |a - b| - 1 (wrapping: 0xffffffff for a == b) in one instruction
__device__ __forceinline__ uint32_t abs_diff_minus_1(int a, int b) {
    uint32_t d;
    asm("v_sad_u32 %0, %1, %2, -1" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Two distinct cells attack each other iff they share one of the 13 lines: every non-zero coordinate offset has the same
// magnitude.  With a, b, c = the three magnitudes minus one (a zero offset becomes 0xffffffff = -1): the largest magnitude is the
// SIGNED maximum, the smallest non-zero one the UNSIGNED minimum, and the cells share a line iff the two are equal.
__device__ __forceinline__ bool on_a_line_m1(uint32_t a, uint32_t b, uint32_t c) {
    const int hi = max((int)a, max((int)b, (int)c));
    const uint32_t lo = min(a, min(b, c));
    return (uint32_t)hi == lo;
}

// ------------------------------------------------------------------------------------------------
// init kernel: CI chains per wavefront, L = 64 / CI lanes each.  Seeds the stream (np.random.seed, experiments.py:201/288),
// builds the initial state (mcmc_board.py:26-59, mcmc.py:20-104), counts E0 and writes the chain
// record {mt[624], pos, gen_end, E0, state bytes} to the workspace.
// Nearly all of it is serial per chain (the seeding recurrence, the sequential draws, the Fisher-Yates swaps), so a wavefront
// that carried one chain issued every instruction for one useful lane and the kernel was bound by instruction issue: four
// chains per wavefront (round 3) run the same instruction stream for four chains (65 536 chains of N = 12: board 1.07 ->
// 0.65 ms; full_3d random, whose Fisher-Yates loop is bound by the LDS round trip per swap and by the chains a CU's LDS holds,
// 7.8 -> 6.7 ms at two chains per wavefront; profiles/r03_init_kernel.txt).  Chains diverge only inside the rejection loops.
// LDS per chain (init_words): mt[624] | state bytes | (full_3d random only) uint16 perm[N^3], later the line counters of E0
// ------------------------------------------------------------------------------------------------
template <int CI>
__global__ __launch_bounds__(64) void mcq_init_kernel(KArgs a) {
    extern __shared__ uint32_t lds[];
    constexpr int L = 64 / CI;
    const int lane = threadIdx.x, sub = lane & (L - 1), grp = lane / L;
    const long long slot = (long long)blockIdx.x * CI + grp, mine = a.chain0 + slot;
    const bool valid = mine < a.n_chains;
    const long long chain = valid ? mine : a.n_chains - 1;  // (an idle group repeats the last chain in its own LDS slice and writes nothing)
    const int N = a.N, Q = a.Q;
    uint32_t* mt = lds + (size_t)grp * a.init_words;
    uint8_t* st = (uint8_t*)(mt + MT_N);
    uint16_t* perm = (uint16_t*)(st + ((a.state_bytes + 3) & ~3));
    // any / sum over the L lanes of a chain
    auto group_any = [&](bool p) { return ((__ballot(p) >> (grp * L)) & (L == 64 ? ~0ull : ((1ull << L) - 1ull))) != 0ull; };
    auto group_sum = [&](int v) {
        for (int o = 1; o < L; o <<= 1) v += __shfl_xor(v, o, 64);
        return v;
    };

    InitRng rng;
    rng.mt = mt, rng.win = 0, rng.pos = 0, rng.gen_end = 0, rng.sub = sub, rng.gbase = lane - sub, rng.L = L;  // NumPy's pos == 624: the first draw starts a generation
    rng.philox = a.rng == MCQ_RNG_PHILOX4X32_10, rng.key = a.seeds[chain], rng.wraps = 0;
    if (a.stream) {
        // the chain continues a caller's stream (mcq_params.stream_states; seed=None in the reference): words [0, gen_end) of the record belong to the
        // current generation, the rest to the one before (stream_layout() on the host rewound them), gen_end a multiple of 64 or 624
        const uint32_t* g = a.stream + chain * 626LL;
        for (int p = sub; p < MT_N; p += L) mt[p] = g[p];
        rng.pos = (int)g[624], rng.gen_end = (int)g[625];
        __syncthreads();  // (a workgroup of one wavefront: orders the LDS writes before the window's reads for the compiler as well)
        if (rng.pos & (L - 1)) rng.fill(rng.pos & ~(L - 1));  // the window next() refills at every L-th word
    } else {  // init_genrand: key[p] = s; s = 1812433253 * (s ^ (s >> 30)) + p + 1
        uint32_t s = a.seeds[chain];
        for (int p = 0; p < MT_N; p++) {
            if (sub == (p & (L - 1))) mt[p] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)p + 1u;
        }
    }
    const int pos_start = rng.pos;

    const unsigned mN = (unsigned)(N - 1);
    if (a.mode == MCQ_MODE_BOARD) {
        if (a.init == MCQ_INIT_RANDOM) {  // N*N sequential draws, row-major
            for (int c = 0; c < Q; c++) {
                const int h = rng.bounded(mN, a.maskN);
                if (sub == 0) st[c] = (uint8_t)h;
            }
        } else if (a.init == MCQ_INIT_LATIN) {
            for (int c = sub; c < Q; c += L) st[c] = (uint8_t)((c / N + c % N) % N);
        } else if (a.klarner_M == 0) {
            for (int c = sub; c < Q; c += L) st[c] = (uint8_t)((3 * (c / N) + 5 * (c % N)) % N);
        } else {  // Klarner core M x M, the other cells drawn in row-major order
            const int M = a.klarner_M;
            for (int c = 0; c < Q; c++) {
                const int i = c / N, j = c % N;
                int h;
                if (i < M && j < M) h = (3 * i + 5 * j) % M;
                else h = rng.bounded(mN, a.maskN);
                if (sub == 0) st[c] = (uint8_t)h;
            }
        }
    } else {
        if (a.init == MCQ_INIT_LATIN) {
            for (int c = sub; c < Q; c += L) {
                const int i = c / N, j = c % N;
                st[3 * c] = (uint8_t)i, st[3 * c + 1] = (uint8_t)j, st[3 * c + 2] = (uint8_t)((i + j) % N);
            }
        } else if (a.init == MCQ_INIT_KLARNER && a.klarner_M == 0) {
            for (int c = sub; c < Q; c += L) {
                const int i = c / N, j = c % N;
                st[3 * c] = (uint8_t)i, st[3 * c + 1] = (uint8_t)j, st[3 * c + 2] = (uint8_t)((3 * i + 5 * j) % N);
            }
        } else if (a.init == MCQ_INIT_KLARNER) {
            // core in row-major order, then (i,j,k) triples rejected while already used (mcmc.py:63-88)
            const int M = a.klarner_M;
            for (int c = sub; c < M * M; c += L) {
                const int i = c / M, j = c % M;
                st[3 * c] = (uint8_t)i, st[3 * c + 1] = (uint8_t)j, st[3 * c + 2] = (uint8_t)((3 * i + 5 * j) % M);
            }
            int n = M * M;
            while (__any(n < Q)) {  // (the chains of a wavefront accept at different rates: a chain that is done draws nothing more)
                if (n < Q) {
                    const int i = rng.bounded(mN, a.maskN);
                    const int j = rng.bounded(mN, a.maskN);
                    const int k = rng.bounded(mN, a.maskN);
                    bool used = false;
                    for (int c = sub; c < n; c += L) used |= (st[3 * c] == i && st[3 * c + 1] == j && st[3 * c + 2] == k);
                    if (!group_any(used)) {
                        if (sub == 0) st[3 * n] = (uint8_t)i, st[3 * n + 1] = (uint8_t)j, st[3 * n + 2] = (uint8_t)k;
                        n++;
                    }
                }
            }
        } else {
            // np.random.choice(N^3, Q, replace=False) = permutation(N^3)[:Q]: identity array, then for
            // t = n-1 .. 1 swap(arr[t], arr[bounded(t)]) (mcmc.py:97); cells decoded k fastest.
            const int n = N * N * N;
            if (a.perm) {
                // beyond N = 32 the cells no longer fit LDS (nor, from N = 41, 16 bits): the array lives in global memory, a slice per
                // chain of this launch.  Every lane of the chain reads the two entries of a swap (one request), its first lane writes them;
                // a wavefront's accesses to one address stay in program order.  ~1.5 us per swap: N = 64 takes 0.4 s, once per run.
                uint32_t* gp = a.perm + slot * (long long)n;
                for (int t = sub; t < n; t += L) gp[t] = (uint32_t)t;
                for (int t = n - 1; t >= 1; t--) {
                    const int s = rng.bounded((unsigned)t, mask_for((unsigned)t));
                    const uint32_t at = gp[t], as = gp[s];
                    if (sub == 0) gp[t] = as, gp[s] = at;
                }
                for (int c = sub; c < Q; c += L) {
                    const int f = (int)gp[c];
                    st[3 * c] = (uint8_t)(f / (N * N)), st[3 * c + 1] = (uint8_t)((f / N) % N), st[3 * c + 2] = (uint8_t)(f % N);
                }
            } else {
                for (int t = sub; t < n; t += L) perm[t] = (uint16_t)t;
                for (int t = n - 1; t >= 1; t--) {
                    const int s = rng.bounded((unsigned)t, mask_for((unsigned)t));
                    const uint16_t at = perm[t], as = perm[s];
                    if (sub == 0) perm[t] = as, perm[s] = at;
                }
                for (int c = sub; c < Q; c += L) {
                    const int f = perm[c];
                    st[3 * c] = (uint8_t)(f / (N * N)), st[3 * c + 1] = (uint8_t)((f / N) % N), st[3 * c + 2] = (uint8_t)(f % N);
                }
            }
        }
    }

    // E0 = number of unordered attacking pairs (mcmc_board.py:82-122, mcmc.py:134-169).  Two distinct cells attack iff they
    // share one of the 13 lines through a cell, and no two cells share more than one, so E0 = sum over lines of c (c - 1) / 2
    // with c the queens on the line: one byte counter per line (c <= N), O(Q) increments instead of Q^2 / 2 pair tests.
    //   N^2 lines each:        (j, k) along i | (i, k) along j | (i, j) along k
    //   N (2N - 1) lines each: (k, i - j), (k, i + j) | (j, i - k), (j, i + k) | (i, j - k), (i, j + k)      planar diagonals
    //   (2N - 1)^2 lines each: (i - j, i - k), (i - j, i + k), (i + j, i - k), (i + j, i + k)                space diagonals
    // One family at a time in a (2N - 1)^2-byte array that takes the place of the permutation array (no longer needed): all 13
    // at once are 30 N^2 bytes, which keeps a CU to few chains at N = 12 and does not fit at all beyond N = 70.
    int e = 0;
    {
        uint32_t* cnt = (uint32_t*)perm;
        const int D = 2 * N - 1, o = N - 1;
        const bool board = a.mode == MCQ_MODE_BOARD;
        for (int f = 0; f < 13; f++) {
            if (f == 2 && board) continue;  // lines along k: on a board the column (i, j) itself, one queen each
            const int lines = f < 3 ? N * N : f < 9 ? N * D : D * D;
            for (int w = sub; w < (lines + 3) / 4; w += L) cnt[w] = 0;
            __syncthreads();  // (a workgroup of one wavefront: orders the phases for the compiler as well)
            for (int c = sub; c < Q; c += L) {
                int i, j, k;
                if (board) i = c / N, j = c % N, k = st[c];
                else i = st[3 * c], j = st[3 * c + 1], k = st[3 * c + 2];
                int line;
                switch (f) {
                case 0: line = j * N + k; break;
                case 1: line = i * N + k; break;
                case 2: line = i * N + j; break;
                case 3: line = k * D + (i - j + o); break;
                case 4: line = k * D + (i + j); break;
                case 5: line = j * D + (i - k + o); break;
                case 6: line = j * D + (i + k); break;
                case 7: line = i * D + (j - k + o); break;
                case 8: line = i * D + (j + k); break;
                case 9: line = (i - j + o) * D + (i - k + o); break;
                case 10: line = (i - j + o) * D + (i + k); break;
                case 11: line = (i + j) * D + (i - k + o); break;
                default: line = (i + j) * D + (i + k); break;
                }
                atomicAdd(&cnt[line >> 2], 1u << (8 * (line & 3)));
            }
            __syncthreads();
            for (int w = sub; w < (lines + 3) / 4; w += L) {
                const uint32_t x = cnt[w];
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const int c = (int)((x >> (8 * b)) & 0xffu);
                    e += c * (c - 1) / 2;
                }
            }
            __syncthreads();
        }
    }
    e = group_sum(e);

    if (!valid) return;
    uint32_t* rec = a.ws + chain * (long long)a.rec_words;
    for (int w = sub; w < MT_N; w += L) rec[w] = mt[w];
    if (sub == 0) rec[REC_MIRROR] = mt[0], rec[REC_POS] = (uint32_t)rng.pos, rec[REC_GEN_END] = (uint32_t)rng.gen_end, rec[REC_E0] = (uint32_t)e;
    if (sub == 0 && a.out.stream_words) a.out.stream_words[chain] = rng.philox ? 0u : (uint32_t)(rng.wraps * MT_N + rng.pos - pos_start);  // the sweep adds its own
    uint8_t* rst = (uint8_t*)(rec + REC_STATE);
    for (int c = sub; c < a.state_bytes; c += L) rst[c] = st[c];
    if (a.mode == MCQ_MODE_FULL3D && a.qtab) {
        if (N > 32) {
            uint32_t* qt = (uint32_t*)a.qtab + chain * (long long)a.qtab_stride;
            for (int c = sub; c < Q; c += L) qt[c] = (uint32_t)st[3 * c] | ((uint32_t)st[3 * c + 1] << 8) | ((uint32_t)st[3 * c + 2] << 16);
        } else {
            uint16_t* qt = a.qtab + chain * (long long)a.qtab_stride;
            for (int c = sub; c < Q; c += L) qt[c] = (uint16_t)(st[3 * c] | (st[3 * c + 1] << 5) | (st[3 * c + 2] << 10));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// sweep kernel
// ------------------------------------------------------------------------------------------------
// The SIMD's instruction arbiter serves the highest priority first and, among equals, the oldest wavefront.  Left alone,
// the first wavefront of a SIMD runs almost unimpeded and the last one gets the leftovers: the four wavefronts of a SIMD
// finish 16.6 / 21.3 / 26.8 / 32.6 ms into a 33 ms sweep, and the SIMD spends the second half of the kernel partly empty.
// The sweep therefore paces itself: every 64 steps a wavefront compares its progress with that of the wavefronts sharing
// its SIMD (a 16-word row per SIMD in the workspace) and sets its priority to the number of them that are ahead of it.
// All 4 096 wavefronts of the headline run then finish within 5 % of each other (27.4 .. 28.7 ms).
__device__ __forceinline__ void set_priority(int k) {  // s_setprio takes an immediate
    switch (k) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// wave-uniform "some lane has p": the ballot is compared in the scalar unit (no per-lane 0/1 materialised)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// Attack masks of an (old, new) height pair in one register, N <= 16: low half B_old, high half B_new, each half
// becomes B | B << d | B >> d within its 16 bits (packed 16-bit shifts; bits >= 16 of a half are never probed).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_star(uint32_t bb, uint32_t d) {
    const u16x2 v = __builtin_bit_cast(u16x2, bb);
    const u16x2 dd = {(unsigned short)d, (unsigned short)d};
    const u16x2 l = v << dd, r = v >> dd;
    return bb | __builtin_bit_cast(uint32_t, l) | __builtin_bit_cast(uint32_t, r);
}

// the same with a shift distance per half (dd = d_low | d_high << 16)
__device__ __forceinline__ uint32_t pk_star2(uint32_t bb, uint32_t dd) {
    const u16x2 v = __builtin_bit_cast(u16x2, bb), d = __builtin_bit_cast(u16x2, dd);
    const u16x2 l = v << d, r = v >> d;
    return bb | __builtin_bit_cast(uint32_t, l) | __builtin_bit_cast(uint32_t, r);
}

// DPP reductions over the G lanes of a group (G = 2, 4, 8 or 16), result in every lane of the group.
template <int G>
__device__ __forceinline__ int group_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);                // quad_perm [1,0,3,2]
    if (G >= 4) v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);    // quad_perm [2,3,0,1]
    if (G >= 8) v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);   // row_half_mirror
    if (G >= 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);  // row_mirror
    return v;
}
template <int G>
__device__ __forceinline__ uint32_t group_or(uint32_t u) {
    int v = (int)u;
    v |= __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false);
    if (G >= 4) v |= __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false);
    if (G >= 8) v |= __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false);
    if (G >= 16) v |= __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);
    return (uint32_t)v;
}

// Per-chain stream state, replicated in the G lanes of the group.
//
// The 624 raw MT19937 words of a chain stay in its record in GLOBAL memory (2.5 KB per chain would
// cap a CU at ~56 chains if they lived in LDS).  They are regenerated in blocks of 16 words, 16/G
// words per lane: the block's inputs are loaded (`issue`) one or more Metropolis steps before they are
// consumed (`complete`), so the memory latency is covered by the steps in between; `complete`
// twists, writes the new raw words back in place, tempers them and appends them to a 64-slot ring
// of ready words in LDS and records one accept flag per word ((w & maskN) <= N-1, the masked
// rejection test of randint(0, N)) in the register pair `ok`.
//   pos, gen   absolute counters of consumed / generated words; ring slot = counter & 63
//   gi         MT index of the next block to generate (multiple of 16, wraps at 624)
//
// PHILOX (mcq_params.rng == MCQ_RNG_PHILOX4X32_10): the same ring, but a block of 16 words is computed from the chain's
// word counter (`generate`) instead of being regenerated from a state in memory: no issue / complete, no record traffic.
// MIRROR = false: a ring without the mirrored slots (readers wrap their slot indices themselves): 128 bytes of LDS per chain less.
template <int G, bool HASQ, bool PHILOX = false, bool MIRROR = true>
struct Stream {
    static constexpr int WPL = 16 / G;  // words per lane in a block
    char* wbase;    // wave-uniform: record of the wavefront's first chain (kept in scalar registers)
    uint32_t coff;  // byte offset of this chain's record from wbase
    uint32_t* ring;
    uint32_t pos, gen;
    int gi;
    // Accept flags of the ready words, RELATIVE to the read position: bit b of `ok` says that word pos + b passes (w & maskN) <= N-1
    // (randint(0, N)); `okq` (HASQ, full_3d) the same for (w & maskQ) <= Q-1 (randint(0, Q)).  Consuming n words shifts them down
    // by n, a new block is OR-ed in at gen - pos: the low word is the view a batched draw works on, and no bit survives its word.
    uint64_t ok, okq;
    bool pending;
    uint32_t pa[WPL], pn, px[WPL];
    int gl;
    unsigned maskN, mN, maskQ, mQ;
    uint32_t tc1, tc2;  // tempering masks, kept in scalar registers so that (y << s) & c ^ y is one 3-input op
    uint32_t tlow, tmat;   // 0x7fffffff and the twist matrix 0x9908b0df, scalar registers as well
    uint32_t okM4, okK4;   // maskN and 0x80 + (N - 1) in each byte: four randint(0, N) accept tests in one subtraction
    uint32_t pkey, gen_hi;  // PHILOX: the chain's key (its seed); bits 32.. of the generated-word counter

    __device__ __forceinline__ uint32_t temper(uint32_t y) const {
        y ^= y >> 11;
        y = __builtin_amdgcn_bitop3_b32(y << 7, tc1, y, 0x6a);   // ((y << 7) & c1) ^ y
        y = __builtin_amdgcn_bitop3_b32(y << 15, tc2, y, 0x6a);  // ((y << 15) & c2) ^ y
        y ^= y >> 18;
        return y;
    }
    // new raw word from words i (top bit), i+1 (low 31 bits) and i+397
    __device__ __forceinline__ uint32_t twist(uint32_t cur, uint32_t nxt, uint32_t x) const {
        uint32_t y;
        asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y) : "s"(tlow), "v"(nxt), "v"(cur));  // (nxt & 0x7fffffff) | (cur & 0x80000000)
        const uint32_t odd = (uint32_t)__builtin_amdgcn_sbfe((int)nxt, 0, 1);       // all ones iff y is odd
        return __builtin_amdgcn_bitop3_b32(odd, tmat, y >> 1, 0x6a) ^ x;           // ((odd & matrix) ^ (y >> 1)) ^ x
    }

    // the next n words have been used
    __device__ __forceinline__ void consume(uint32_t n) {
        pos += n;
        ok >>= n;
        if (HASQ) okq >>= n;
    }
    // flags of the block that lands at word `gen` (16 bits; called before gen advances)
    __device__ __forceinline__ void add_flags(uint32_t bits16, uint32_t bitsq16) {
        const uint32_t rel = gen - pos;  // <= 48
        ok |= (uint64_t)bits16 << rel;
        if (HASQ) okq |= (uint64_t)bitsq16 << rel;
    }

    // randint(0, N) accept flags of four words (bit w: word w passes): the low bytes side by side; per byte
    // (0x80 + N - 1) - (t & maskN) keeps bit 7 iff the word passes; a dot product with (1, 2, 4, 8) collects the four flags
    __device__ __forceinline__ uint32_t flags4(uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) const {
        const uint32_t b01 = __builtin_amdgcn_perm(t1, t0, 0x0c0c0400u), b23 = __builtin_amdgcn_perm(t3, t2, 0x04000c0cu);
        const uint32_t d = okK4 - ((b01 | b23) & okM4);
        return __builtin_amdgcn_udot4((d >> 7) & 0x01010101u, 0x08040201u, 0u, false);
    }

    // MT word `idx` of this chain: uniform base + 32-bit offset, so the address needs no 64-bit vector math
    __device__ __forceinline__ uint32_t* word(int idx) const { return (uint32_t*)(wbase + (coff + 4u * (uint32_t)idx)); }
    // Timing experiments (tools/exp_build.sh; never defined in the shipped library, results are wrong with any of them):
    // MCQ_EXP_HOT_LOADS / MCQ_EXP_HOT_STORES fold the block's reads / writes onto the first 64 words of the record (cache-resident),
    // MCQ_EXP_NO_STORE drops the write-back, MCQ_EXP_NT_STORE / MCQ_EXP_NT_XLOAD mark the stream's accesses non-temporal.
#ifdef MCQ_EXP_HOT_LOADS
    __device__ __forceinline__ uint32_t* lword(int idx) const { return word(idx & 63); }
#else
    __device__ __forceinline__ uint32_t* lword(int idx) const { return word(idx); }
#endif
#ifdef MCQ_EXP_HOT_STORES
    __device__ __forceinline__ uint32_t* sword(int idx) const { return word(idx & 63); }
#else
    __device__ __forceinline__ uint32_t* sword(int idx) const { return word(idx); }
#endif

    // load the inputs of block gi: words i, i+1 and (i+397) mod 624 for the lane's WPL words.  Word 624 of the record
    // mirrors word 0 (the NEW word 0, regenerated earlier in the same pass), which is what index 624 stands for in both
    // roles; the lane's run of (i+397) mod 624 starts at 1 mod 4, so it can only run over the end by that one word.
    __device__ __forceinline__ void issue() {
        struct __attribute__((packed, aligned(4))) W4 { uint32_t x, y, z, w; };
        struct __attribute__((packed, aligned(4))) W2 { uint32_t x, y; };
        const int i0 = gi + gl * WPL;
        const int ix0 = i0 + MT_M >= MT_N ? i0 + MT_M - MT_N : i0 + MT_M;
        if constexpr (WPL == 8) {
            // two runs of four: each half of (i+397) mod 624 wraps on its own (the first can run over the end by the one mirrored
            // word, at i0 = 224; the second then starts at word 1)
            const int ix1 = i0 + 4 + MT_M >= MT_N ? i0 + 4 + MT_M - MT_N : i0 + 4 + MT_M;
            const uint4 q0 = *(const uint4*)lword(i0), q1 = *(const uint4*)lword(i0 + 4);
            pa[0] = q0.x, pa[1] = q0.y, pa[2] = q0.z, pa[3] = q0.w, pa[4] = q1.x, pa[5] = q1.y, pa[6] = q1.z, pa[7] = q1.w;
            const W4 x0 = *(const W4*)lword(ix0), x1 = *(const W4*)lword(ix1);
            px[0] = x0.x, px[1] = x0.y, px[2] = x0.z, px[3] = x0.w, px[4] = x1.x, px[5] = x1.y, px[6] = x1.z, px[7] = x1.w;
        } else if constexpr (WPL == 4) {
#ifdef MCQ_EXP_NT_CUR
            typedef uint32_t u32x4a __attribute__((ext_vector_type(4)));
            const u32x4a q = __builtin_nontemporal_load((const u32x4a*)lword(i0));
#else
            const uint4 q = *(const uint4*)lword(i0);
#endif
            pa[0] = q.x, pa[1] = q.y, pa[2] = q.z, pa[3] = q.w;
#ifdef MCQ_EXP_NT_XLOAD
            typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));
            const u32x4u xx = __builtin_nontemporal_load((const u32x4u*)lword(ix0));
            px[0] = xx.x, px[1] = xx.y, px[2] = xx.z, px[3] = xx.w;
#else
            const W4 x = *(const W4*)lword(ix0);
            px[0] = x.x, px[1] = x.y, px[2] = x.z, px[3] = x.w;
#endif
        } else if constexpr (WPL == 2) {
            const uint2 q = *(const uint2*)lword(i0);
            pa[0] = q.x, pa[1] = q.y;
            const W2 x = *(const W2*)lword(ix0);
            px[0] = x.x, px[1] = x.y;
        } else {
            pa[0] = *lword(i0);
            px[0] = *lword(ix0);
        }
        pn = *lword(i0 + WPL);
        pending = true;
    }

    // PHILOX: the next 16 words of the chain's stream straight from the counter; append to the ring.  Needs gen - pos <= 48.
    // Word w of the stream is element w % 4 of block w / 4; a lane computes the block that holds its WPL words.
    __device__ __forceinline__ void generate() {
        uint32_t o[4], t[WPL];
        const uint32_t blk = (gen >> 2) | (gen_hi << 30);  // gen is a multiple of 16: the low two bits of blk are free for the lane's part
        philox4x32_10(blk | ((uint32_t)(gl * WPL) >> 2), gen_hi >> 2, pkey, o);
        if constexpr (WPL == 8) {  // two blocks per lane
            t[0] = o[0], t[1] = o[1], t[2] = o[2], t[3] = o[3];
            philox4x32_10(blk | (((uint32_t)(gl * WPL) >> 2) + 1u), gen_hi >> 2, pkey, o);
            t[4] = o[0], t[5] = o[1], t[6] = o[2], t[7] = o[3];
        } else if constexpr (WPL == 4) {
            t[0] = o[0], t[1] = o[1], t[2] = o[2], t[3] = o[3];
        } else if constexpr (WPL == 2) {
            t[0] = (gl & 1) ? o[2] : o[0], t[1] = (gl & 1) ? o[3] : o[1];
        } else {
            t[0] = (gl & 2) ? ((gl & 1) ? o[3] : o[2]) : ((gl & 1) ? o[1] : o[0]);
        }
        append(t);
        gen_hi += gen == 0u ? 1u : 0u;  // gen has just advanced by 16
    }

    // tempered words of a block -> ring slots [gen, gen + 16) (+ mirror) and their accept bits; advances gen
    __device__ __forceinline__ void append(const uint32_t (&t)[WPL]) {
        const int so = gen & (RING - 1);
        uint32_t bits = 0, bitsq = 0;
        uint32_t* slot = ring + so + gl * WPL;
        uint32_t* mirror = ring + (so < RING_MIRROR ? so + RING : so) + gl * WPL;  // mirror of slots 0..31 (otherwise the same store again)
        if constexpr (WPL == 8) {
            *(uint4*)slot = make_uint4(t[0], t[1], t[2], t[3]), *(uint4*)(slot + 4) = make_uint4(t[4], t[5], t[6], t[7]);
            if constexpr (MIRROR) *(uint4*)mirror = make_uint4(t[0], t[1], t[2], t[3]), *(uint4*)(mirror + 4) = make_uint4(t[4], t[5], t[6], t[7]);
            bits = (flags4(t[0], t[1], t[2], t[3]) | (flags4(t[4], t[5], t[6], t[7]) << 4)) << (gl * 8);
        } else if constexpr (WPL == 4) {
            *(uint4*)slot = make_uint4(t[0], t[1], t[2], t[3]);
            if constexpr (MIRROR) *(uint4*)mirror = make_uint4(t[0], t[1], t[2], t[3]);
            bits = flags4(t[0], t[1], t[2], t[3]) << (gl * 4);
        } else {
#pragma unroll
            for (int w = 0; w < WPL; w++) {
                slot[w] = t[w];
                if constexpr (MIRROR) mirror[w] = t[w];
                bits |= ((t[w] & maskN) <= mN ? 1u : 0u) << (gl * WPL + w);
            }
        }
        if (HASQ) {
#pragma unroll
            for (int w = 0; w < WPL; w++) bitsq |= ((t[w] & maskQ) <= mQ ? 1u : 0u) << (gl * WPL + w);
            const uint32_t both = group_or<G>(bits | (bitsq << 16));
            add_flags(both & 0xffffu, both >> 16);
        } else {
            add_flags(group_or<G>(bits), 0u);
        }
        gen += 16;
    }

    // twist + temper the block loaded by issue(); append to the ring.  Needs gen - pos <= 48.
    __device__ __forceinline__ void complete() {
        const int i0 = gi + gl * WPL;
        const int so = gen & (RING - 1);
        uint32_t v[WPL], t[WPL], bits = 0, bitsq = 0;
#pragma unroll
        for (int w = 0; w < WPL; w++) {
            v[w] = twist(pa[w], w + 1 < WPL ? pa[(w + 1) % WPL] : pn, px[w]);
            t[w] = temper(v[w]);
        }
        uint32_t* slot = ring + so + gl * WPL;
        uint32_t* mirror = ring + (so < RING_MIRROR ? so + RING : so) + gl * WPL;  // mirror of slots 0..31 (otherwise the same store again)
        if constexpr (WPL == 8) {
            *(uint4*)sword(i0) = make_uint4(v[0], v[1], v[2], v[3]), *(uint4*)sword(i0 + 4) = make_uint4(v[4], v[5], v[6], v[7]);
            *(uint4*)slot = make_uint4(t[0], t[1], t[2], t[3]), *(uint4*)(slot + 4) = make_uint4(t[4], t[5], t[6], t[7]);
            if constexpr (MIRROR) *(uint4*)mirror = make_uint4(t[0], t[1], t[2], t[3]), *(uint4*)(mirror + 4) = make_uint4(t[4], t[5], t[6], t[7]);
            bits = (flags4(t[0], t[1], t[2], t[3]) | (flags4(t[4], t[5], t[6], t[7]) << 4)) << (gl * 8);
        } else if constexpr (WPL == 4) {
#if defined(MCQ_EXP_NO_STORE)
#elif defined(MCQ_EXP_NT_STORE)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store((u32x4){v[0], v[1], v[2], v[3]}, (u32x4*)sword(i0));
#elif defined(MCQ_EXP_LAST_TOUCH)
            // timing experiment (profiles/r03_last_touch.txt): records are 128-byte aligned in this build, so the block with gi & 16
            // is the second half of its line -- after its write-back the line is dead for ~100 steps: a streaming store for that one
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            if (gi & 16) __builtin_nontemporal_store((u32x4){v[0], v[1], v[2], v[3]}, (u32x4*)sword(i0));
            else *(uint4*)sword(i0) = make_uint4(v[0], v[1], v[2], v[3]);
#else
            *(uint4*)sword(i0) = make_uint4(v[0], v[1], v[2], v[3]);
#endif
            *(uint4*)slot = make_uint4(t[0], t[1], t[2], t[3]);
            if constexpr (MIRROR) *(uint4*)mirror = make_uint4(t[0], t[1], t[2], t[3]);
            bits = flags4(t[0], t[1], t[2], t[3]) << (gl * 4);
        } else {
            if constexpr (WPL == 2) *(uint2*)word(i0) = make_uint2(v[0], v[1]);
            else *word(i0) = v[0];
#pragma unroll
            for (int w = 0; w < WPL; w++) {
                slot[w] = t[w];
                if constexpr (MIRROR) mirror[w] = t[w];
                bits |= ((t[w] & maskN) <= mN ? 1u : 0u) << (gl * WPL + w);
            }
        }
        if (HASQ) {
#pragma unroll
            for (int w = 0; w < WPL; w++) bitsq |= ((t[w] & maskQ) <= mQ ? 1u : 0u) << (gl * WPL + w);
        }
        if (i0 == 0) *word(REC_MIRROR) = v[0];
        if (HASQ) {  // both 16-bit masks through one reduction over the group
            const uint32_t both = group_or<G>(bits | (bitsq << 16));
            add_flags(both & 0xffffu, both >> 16);
        } else {
            add_flags(group_or<G>(bits), 0u);
        }
        gen += 16;
        gi = gi + 16 == MT_N ? 0 : gi + 16;
        pending = false;
    }

    // PHILOX: continue the stream at word rpos (the init kernel consumed the words before it): the 16-word group that holds
    // rpos is generated again and the words in front of rpos are skipped.
    __device__ __forceinline__ void attach_philox(uint32_t key, uint32_t rpos) {
        pkey = key, pos = gen = rpos & ~15u, gen_hi = 0, gi = 0, ok = okq = 0;
        generate();
        generate();
        consume(rpos & 15u);
    }

    // continue the stream of a chain record: words [rpos, rge) of the current generation are already
    // twisted but not consumed (fewer than 64 of them); temper them into the ring.
    __device__ __forceinline__ void attach(char* wave_base, uint32_t chain_off, uint32_t* lds_ring, int rpos, int rge, int gl_, unsigned maskN_,
                                           unsigned mN_, unsigned maskQ_, unsigned mQ_, uint32_t c1, uint32_t c2) {
        wbase = wave_base, coff = chain_off, ring = lds_ring, gl = gl_, maskN = maskN_, mN = mN_, maskQ = maskQ_, mQ = mQ_, tc1 = c1, tc2 = c2;
        tlow = 0x7fffffffu, tmat = 0x9908b0dfu;
        asm volatile("" : "+s"(tlow), "+s"(tmat));  // opaque scalars, like tc1 / tc2
        okM4 = maskN_ * 0x01010101u, okK4 = (0x80u + mN_) * 0x01010101u;  // maskN, N - 1 <= 31
        ok = okq = 0;
        pos = (uint32_t)rpos, gen = (uint32_t)rge, gi = rge == MT_N ? 0 : rge;
        pending = false, pn = 0;
        pkey = 0, gen_hi = 0;
#pragma unroll
        for (int w = 0; w < WPL; w++) pa[w] = px[w] = 0;
        if constexpr (PHILOX) return;  // attach_philox() follows
        for (int t0 = rpos & ~15; t0 < rge; t0 += 16) {
            uint32_t bits = 0, bitsq = 0;
#pragma unroll
            for (int w = 0; w < WPL; w++) {
                const int t = t0 + gl * WPL + w;
                if (t >= rpos && t < rge) {
                    const uint32_t x = temper(*word(t));
                    ring[t & (RING - 1)] = x;
                    if (MIRROR && (t & (RING - 1)) < RING_MIRROR) ring[(t & (RING - 1)) + RING] = x;
                    bits |= ((x & maskN) <= mN ? 1u : 0u) << (gl * WPL + w);
                    bitsq |= ((x & maskQ) <= mQ ? 1u : 0u) << (gl * WPL + w);
                }
            }
            // block t0 sits at word t0 - rpos of the view (the first block may start in front of rpos: those flags are not set)
            const uint64_t f = group_or<G>(bits), fq = HASQ ? group_or<G>(bitsq) : 0u;
            ok |= t0 >= rpos ? f << (t0 - rpos) : f >> (rpos - t0);
            if (HASQ) okq |= t0 >= rpos ? fq << (t0 - rpos) : fq >> (rpos - t0);
        }
    }
};

// exact accept test: u < min(1, exp(x)), x = -beta * dE in float64 (experiments.py:238-239, 326-327).
// min(1.0, e) keeps 1.0 unless e < 1.0, so a NaN e accepts, like the reference.
// Returns bit 0 = accepted, bit 1 = u within 4 ulp of the probability (a "near tie").
__device__ __forceinline__ int accept_exact(double beta, int dE, uint32_t w1, uint32_t w2) {
    const double x = -beta * (double)dE;
    if (!(x < 0.0)) return 1;
    const double u = ((double)(w1 >> 5) * 67108864.0 + (double)(w2 >> 6)) * 1.1102230246251565e-16;  // exact: * 2^-53
    const double e = exp(x);
    if (e < 1.0) {
        const long long d = __double_as_longlong(u) - __double_as_longlong(e);
        return (u < e ? 1 : 0) | ((d < 0 ? -d : d) <= 4 ? 2 : 0);
    }
    return 1;
}

// Replica exchange, the pair's decision (include/mcq.h): the chains on rungs t and t + 1 trade rungs iff u < min(1, exp(x)),
// x = (beta_a - beta_b) (E_a - E_b), u from the two words the lower chain draws -- experiments.py:326-327 applied to the pair.
// Returns bit 0 = swap, bit 1 = u within 4 ulp of the probability.
__device__ __forceinline__ int exchange_decide(double beta_a, double beta_b, int dEab, uint32_t w1, uint32_t w2) {
    const double x = (beta_a - beta_b) * (double)dEab;
    const double u = ((double)(w1 >> 5) * 67108864.0 + (double)(w2 >> 6)) * 1.1102230246251565e-16;
    const double e = exp(x);
    if (e < 1.0) {
        const long long d = __double_as_longlong(u) - __double_as_longlong(e);
        return (u < e ? 1 : 0) | ((d < 0 ? -d : d) <= 4 ? 2 : 0);
    }
    return 1;  // min(1.0, e) == 1.0 > u
}

// LDS per chain: stage[16] | cold[4] | ring[64 + 32 mirrored] | board: heights bytes, pad (pad = (N+2)/4 words >= N-1 bytes) |
//                full_3d: pad[full_pad], column words [Q], pad[full_pad], queens uint16 [Q]; column words are uint16 in the
//                unrolled variants (N <= 16), uint32 otherwise
// (the staging block and the cold scalars sit in front of the ring: ring[-1], which an unused draw position may address -- as a
// READ -- is one of the chain's own cold scalars)
constexpr int LDS_STAGE = 0;                          // word offset of the energy_history staging block
constexpr int LDS_COLD = 16;                          // four per-chain scalars that change rarely: steps_to_best, n_accepted, near ties, history length
constexpr int LDS_RING = 20;                          // word offset of the ring
constexpr int LDS_STATE = LDS_RING + RING + RING_MIRROR;  // word offset of the state

// trace == REDUCED: add the block of 16 history entries [e0, e0 + 16) of the wavefront's chains to the per-entry accumulators.
// A stage word is E | valid << 30 | accepted << 31: `valid` marks an appended history entry, `accepted` that the entry's step
// was accepted.  A chain that stops early (experiments.py:349-353) leaves, at the entry it did not append, a word with the
// accepted flag alone (the step was executed and sits in accepted_steps / rejected_steps all the same, experiments.py:329-332),
// zeroes the rest of its block, and zeroes the whole block after every later reduction, so nothing is counted twice.
// Lane L sums entry L & 15 over the chains of quarter L >> 4, two butterfly steps join the quarters, and lanes 0..15 issue one
// 128-byte atomic instruction per accumulator (E, E^2, accepted | chains << 32): 3 per block of 16 steps instead of 64.
// Every lane of the wavefront takes part.
template <int G>
__device__ __forceinline__ void reduce_block(const uint32_t* wave_stage, int chain_lds_words, int lane, int e0, long long n_entries,
                                             unsigned long long* red, long long red_len) {
    constexpr int CPQ = (64 / G) / 4;  // chains per quarter of the wavefront
    const int ent = lane & 15, q = lane >> 4;
    uint32_t se = 0, ac = 0, cn = 0;
    unsigned long long sq = 0;
#pragma unroll
    for (int c = 0; c < CPQ; c++) {
        const uint32_t x = wave_stage[(q * CPQ + c) * chain_lds_words + ent];
        const bool valid = (x & 0x40000000u) != 0;
        const uint32_t e = valid ? x & 0x3fffffffu : 0u;
        se += e, sq += (unsigned long long)e * e, ac += x >> 31, cn += valid ? 1u : 0u;
    }
    for (int off = 16; off < 64; off <<= 1) {
        se += (uint32_t)__shfl_xor((int)se, off, 64);
        ac += (uint32_t)__shfl_xor((int)ac, off, 64);
        cn += (uint32_t)__shfl_xor((int)cn, off, 64);
        sq += ((unsigned long long)(uint32_t)__shfl_xor((int)(sq >> 32), off, 64) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)sq, off, 64);
    }
    const long long idx = (long long)e0 + ent;
#ifdef MCQ_EXP_NO_ATOMICS
    if (lane < 16 && idx < n_entries && (ac | cn) == 0xffffffffu) {
#else
    if (lane < 16 && idx < n_entries && (ac | cn) != 0) {
#endif
        atomicAdd(red + idx, (unsigned long long)se);
        atomicAdd(red + red_len + idx, sq);
        atomicAdd(red + 2 * red_len + idx, (unsigned long long)ac | ((unsigned long long)cn << 32));
    }
}

// sum of the stripes -> the caller's arrays
__global__ __launch_bounds__(256) void mcq_reduced_finalize_kernel(const unsigned long long* __restrict__ red, long long red_len, long long n_entries,
                                                                   long long* __restrict__ sum, long long* __restrict__ sumsq,
                                                                   long long* __restrict__ accepted, long long* __restrict__ count) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_entries) return;
    unsigned long long t[3] = {0, 0, 0};
    for (int st = 0; st < RED_STRIPES; st++)
        for (int k = 0; k < 3; k++) t[k] += red[((long long)st * 3 + k) * red_len + e];
    sum[e] = (long long)t[0], sumsq[e] = (long long)t[1], accepted[e] = (long long)(t[2] & 0xffffffffull), count[e] = (long long)(t[2] >> 32);
}

// A chain's state to the caller's best_state / final_state row, by the G lanes of its group.  A strict improvement copies the
// board (experiments.py:340-343 `best_state = state.copy()`): ~400 times per chain in a 10^5-step run and 1 600 times at N = 24,
// and with 8 or 16 chains per wavefront some chain improves in every tenth step -- byte stores made that loop 29 % of an N = 24
// step (`apply+history`, profiles/r02_stamp_shares.txt).  Rows are 16-byte aligned whenever Q % 16 == 0 (the host side checks the
// base), so the copy moves 16 bytes per lane and instruction; odd sizes fall back to dwords or bytes.
template <int G>
__device__ __forceinline__ void copy_board_out(uint8_t* dst, const uint8_t* hts, int Q, int gl) {
    if ((Q & 15) == 0) {
        for (int c = gl; c < (Q >> 4); c += G) ((uint4*)dst)[c] = ((const uint4*)hts)[c];
    } else if ((Q & 3) == 0) {
        for (int c = gl; c < (Q >> 2); c += G) ((uint32_t*)dst)[c] = ((const uint32_t*)hts)[c];
    } else {
        for (int c = gl; c < Q; c += G) dst[c] = hts[c];
    }
}
// full_3d: queens are packed i | j << 5 | k << 10 in LDS and (i, j, k) bytes in the caller's row: four queens = three dwords
template <int G>
__device__ __forceinline__ void copy_queens_out(uint8_t* dst, const uint32_t* qn, int Q, int gl) {  // beyond N = 32: i | j << 8 | k << 16
    for (int c = gl; c < Q; c += G) {
        const uint32_t pq = qn[c];
        dst[3 * c] = (uint8_t)pq, dst[3 * c + 1] = (uint8_t)(pq >> 8), dst[3 * c + 2] = (uint8_t)(pq >> 16);
    }
}
template <int G>
__device__ __forceinline__ void copy_queens_out(uint8_t* dst, const uint16_t* qn, int Q, int gl) {
    if ((Q & 3) == 0) {
        for (int c = gl; c < (Q >> 2); c += G) {
            const uint2 p = *(const uint2*)(qn + 4 * c);  // queens 4c .. 4c+3
            const uint32_t q0 = p.x & 0xffffu, q1 = p.x >> 16, q2 = p.y & 0xffffu, q3 = p.y >> 16;
            auto I = [](uint32_t q) { return q & 31u; };
            auto J = [](uint32_t q) { return (q >> 5) & 31u; };
            auto K = [](uint32_t q) { return (q >> 10) & 31u; };
            uint32_t* d = (uint32_t*)(dst + 12 * c);
            d[0] = I(q0) | (J(q0) << 8) | (K(q0) << 16) | (I(q1) << 24);
            d[1] = J(q1) | (K(q1) << 8) | (I(q2) << 16) | (J(q2) << 24);
            d[2] = K(q2) | (I(q3) << 8) | (J(q3) << 16) | (K(q3) << 24);
        }
    } else {
        for (int c = gl; c < Q; c += G) {
            const uint32_t pq = qn[c];
            dst[3 * c] = (uint8_t)(pq & 31), dst[3 * c + 1] = (uint8_t)((pq >> 5) & 31), dst[3 * c + 2] = (uint8_t)((pq >> 10) & 31);
        }
    }
}

// The same as real calls: the kernels whose step exists twice (early stop, reduced trace) would otherwise carry four inlined
// copies of these loops and spill; a call in the rare improvement path costs them nothing measurable and frees ~15 VGPRs.
template <int G>
__device__ __attribute__((noinline)) void copy_board_out_call(uint8_t* dst, const uint8_t* hts, int Q, int gl) { copy_board_out<G>(dst, hts, Q, gl); }
template <int G, typename QN>
__device__ __attribute__((noinline)) void copy_queens_out_call(uint8_t* dst, const QN* qn, int Q, int gl) { copy_queens_out<G>(dst, qn, Q, gl); }
template <int MODE, int G, bool CALL, typename QN>
__device__ __forceinline__ void copy_state_out(uint8_t* dst, const uint8_t* hts, const QN* qn, int Q, int gl) {
    if constexpr (MODE == MCQ_MODE_BOARD) {
        if constexpr (CALL) copy_board_out_call<G>(dst, hts, Q, gl);
        else copy_board_out<G>(dst, hts, Q, gl);
    } else {
        if constexpr (CALL) copy_queens_out_call<G>(dst, qn, Q, gl);
        else copy_queens_out<G>(dst, qn, Q, gl);
    }
}

// NT > 0: ceil(N / G) is a compile-time constant, so the dE probes of a step are issued as one straight-line block
// (all their LDS reads in flight together); NT == 0: run-time loop over the probe passes.
// REDUCED: trace == MCQ_TRACE_REDUCED (per-entry sums accumulated in the sweep); a separate instantiation so that the
// default kernels carry none of its code.
// PHILOX: mcq_params.rng == MCQ_RNG_PHILOX4X32_10 (the stream is computed, not read from the chain record).
// (full_3d at 4 lanes per chain, unrolled: 16 chains of 1 KB per wavefront leave room for 2-3 wavefronts per SIMD, so that variant may
// use the registers of a 2-per-SIMD kernel)
// NC: the board size as a compile-time constant (0: taken from the arguments).  The probe addresses of the later passes are the
// first pass's plus multiples of N: with N known they become immediate offsets of the LDS reads instead of an addition each
// (12 vector instructions of ~208 per step on the headline problem); instantiated for N = 12, the size of BASELINE configs 2 and 3,
// and for config 5's N = 24 (8 lanes, reduced trace).
// EXCH: replica exchange between the chains of a ladder (mcq_params.exchange_every > 0; never with PATIENCE or REDUCED).
// CAND5: board, five candidates for new_k instead of three (N <= 5, where all three equal old_k too often).
// EARLYU: the unpacked probe passes (boards beyond N = 16) request their heights together with the old height; for launches that
// leave the device at most half full (below).
// SLIM (full_3d, unrolled 16-bit column words): the layout that lets BASELINE configs[2]'s 65 536 chains of N = 12 run at 4 lanes per chain in ONE
// resident round (16 chains per wavefront x 16 wavefronts per CU need <= 640 B of LDS per chain): the queen table lives in global memory
// (KArgs::qtab: one 2-byte read per step once q is known, one write per accepted move), the ring has no mirror (readers wrap their
// slot indices) and the column table no pads (an out-of-board diagonal probe reads -- and discards -- ring words in front of the
// table and the next chain's staging block behind it; the workgroup's allocation ends in a spare pad for its last chain):
// stage[16] | cold[4] | ring[64] | column words = 156 words at N = 12, the 624 B of a board chain.
// CNT (board, 4 lanes per chain, N <= 8; mcq_params.flags & MCQ_FLAG_LINE_COUNTERS): dE from per-line occupancy counters in LDS -- the
// formulation BASELINE's north star names -- instead of bit-mask probes of the heights.  Two distinct cells attack each other iff they
// share one of the 13 lines through a cell, and on a board the line along k holds the column's own queen only, so with one byte
// counter per line of the other 12 families
//     dE = sum_f cnt[line_f(i, j, new_k)] - sum_f cnt[line_f(i, j, old_k)] + 12
// (the moving queen sits on every line through its old cell and on none through the new one: every key below contains k).  Every
// line index is linear in (i, j, k); lane gl of a chain owns the families gl, gl + 4, gl + 8: three counters read for the old cell,
// three for the new one, and an accepted move writes those six back (+-1) -- 2 N^2 + 6 N (2N - 1) + 4 (2N - 1)^2 bytes per chain
// (N = 8: 1 748, N = 4: 396), which is what keeps this a variant for small boards (profiles/r04_line_counters.txt).
// WIDE (full_3d beyond N = 32, run-time probe loop, 16 lanes per chain): 64-bit column words (N = 64: 32 KB per chain, four chains per
// wavefront), and the queens -- i | j << 8 | k << 16 in 32-bit entries -- stay in the workspace's table like SLIM's.
template <int MODE, int G, bool PATIENCE, int NT, bool REDUCED, bool PHILOX = false, int NC = 0, bool EXCH = false, bool CAND5 = false, bool EARLYU = false, bool SLIM = false, bool CNT = false,
          bool WIDE = false>
#ifndef MCQ_EXP_WAVES  // experiment (profiles/r03_occupancy5.txt): the register budget of more wavefronts per SIMD
#define MCQ_EXP_WAVES 4
#endif
// (G = 2: 32 chains per wavefront take twice the LDS of 16, so a CU holds two of those wavefronts per SIMD, and each may use the registers of two)
#ifndef MCQ_G2_WAVES
#define MCQ_G2_WAVES 2
#endif
__global__ __launch_bounds__(64, G == 2 ? MCQ_G2_WAVES : (MODE == MCQ_MODE_FULL3D && G == 4 && NT > 0 && !SLIM) ? 2 : (SLIM && NT > 6) ? 3 : MCQ_EXP_WAVES) void mcq_sweep_kernel(KArgs a) {
    static_assert(!SLIM || (MODE == MCQ_MODE_FULL3D && NT > 0 && !PATIENCE), "the slim layout exists for the unrolled full_3d kernels");
    static_assert(!WIDE || (MODE == MCQ_MODE_FULL3D && NT == 0 && NC == 0 && G == 16 && !SLIM), "64-bit column words: the run-time-loop full_3d kernel at 16 lanes per chain");
    static_assert(!CNT || (MODE == MCQ_MODE_BOARD && G == 4 && NT == 0 && !EXCH), "line counters: boards at 4 lanes per chain");
    static_assert(G >= 4 || MODE == MCQ_MODE_BOARD, "two lanes per chain: boards only (full_3d splits a chain's lanes between two cells)");
    static_assert(!EXCH || (!PATIENCE && !REDUCED), "replica exchange runs without early stop and with trace none / i32");
    WAVE_T0;
    // where this wavefront runs: HW_ID = wave slot [3:0], SIMD [5:4], CU [11:8], SE [14:13]; XCC_ID [3:0]
    const uint32_t hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    const int wave_slot = (int)(hw_id & 15u);
    uint32_t* pace_row = a.pace + 16u * (((xcc_id & 7u) << 8) | (((hw_id >> 13) & 3u) << 6) | (((hw_id >> 8) & 15u) << 2) | ((hw_id >> 4) & 3u));
    extern __shared__ uint32_t lds[];
    constexpr int CPW = 64 / G;
    constexpr int SB = 16;             // history entries staged per chain between two flushes
    constexpr int WPL = SB / G;        // ... and how many of them a lane stores in a flush
    // LDS slice of a chain (word offsets): stage[SB] | cold[4] | ring[64 (+ 32 mirrored)] | state
    constexpr int L_STAGE = 0, L_COLD = SB, L_RING = SB + 4, L_STATE = L_RING + RING + (SLIM ? 0 : RING_MIRROR);
    static_assert(SLIM || (L_STAGE == LDS_STAGE && L_COLD == LDS_COLD && L_RING == LDS_RING && L_STATE == LDS_STATE), "layout constants");
    constexpr int LAST = MODE == MCQ_MODE_BOARD ? 5 : 6;  // sequential-draw stages of one proposal
    const int lane = threadIdx.x;
    const int gl = lane & (G - 1), grp = lane / G;
    const long long chain = (long long)blockIdx.x * CPW + grp;
    const int N = NC ? NC : a.N, NN = N * N, Q = NC ? NC * NC : a.Q;  // (the compile-time-N variants are dispatched for Q = N^2 only)
    const int full_pad = SLIM ? 0 : NC ? (NC + 3) & ~3 : a.full_pad;
    const int state_bytes = NC ? (MODE == MCQ_MODE_BOARD ? NC * NC : 3 * NC * NC) : a.state_bytes;
    bool active = chain < a.n_chains;
    constexpr bool reduced = REDUCED;
    if (!PATIENCE && !active && !reduced) return;  // no wave-wide operation below: idle groups of the last wavefront can leave
    const long long crow = active ? chain : 0;

    uint32_t* base = lds + grp * a.chain_lds_words;
    uint32_t* ring = base + L_RING;
    uint32_t* stage = base + L_STAGE;
    // board: the diagonal probes read (and discard) up to N-1 bytes outside either end of the heights: in front of them lies the
    // ring's mirror, behind them (N+2)/4 spare words -- both the chain's own
    uint8_t* hts = (uint8_t*)(base + L_STATE);
    // full_3d: colw[i*N+j] = occupancy word of column (i,j) (bit k set: a queen at (i,j,k)), padded on either
    // side for the out-of-board diagonal probes; qn[q] = queen q as i | j<<5 | k<<10
    constexpr bool NARROW = MODE == MCQ_MODE_FULL3D && NT > 0;  // N <= 16: 16-bit column words
    typedef typename std::conditional<WIDE, uint64_t, typename std::conditional<NARROW, uint16_t, uint32_t>::type>::type colw_t;
    typedef typename std::conditional<WIDE, uint64_t, uint32_t>::type cword_t;  // a column word in a register
    typedef typename std::conditional<WIDE, uint32_t, uint16_t>::type qn_t;     // a queen: i | j << QS | k << 2 QS
    constexpr int QS = WIDE ? 8 : 5;
    constexpr uint32_t QM = WIDE ? 63u : 31u;
    colw_t* colw = (colw_t*)(base + L_STATE) + full_pad;
    qn_t* const qn = [&]() {
        if constexpr (SLIM || WIDE) return (qn_t*)a.qtab + crow * (long long)a.qtab_stride;  // global memory, filled by the init kernel
        else return (qn_t*)(colw + NN + full_pad);
    }();

    // ---- load the chain record ----
    uint32_t* rec = a.ws + crow * (long long)a.rec_words;
    const uint8_t* rst = (const uint8_t*)(rec + REC_STATE);
    // CNT: the counters lie behind the heights and their pad; family f starts at cf_base, and line_f(i, j, k) = cf_base + cf_a i + cf_b j + cf_c k
    uint8_t* const cnt = (uint8_t*)(base + L_STATE + (NN + 3) / 4 + (N + 2) / 4);
    int cf_a[3] = {0, 0, 0}, cf_b[3] = {0, 0, 0}, cf_c[3] = {0, 0, 0}, cf_base[3] = {0, 0, 0};
    if constexpr (CNT) {
        const int D = 2 * N - 1, o = N - 1;
        //                   (j,k)  (i,k) (k,i-j) (k,i+j) (j,i-k) (j,i+k) (i,j-k) (i,j+k) (i-j,i-k)  (i-j,i+k) (i+j,i-k) (i+j,i+k)
        const int fa[12] = {0,     N,    1,      1,      1,      1,      D,      D,      D + 1,     D + 1,    D + 1,    D + 1};
        const int fb[12] = {N,     0,    -1,     1,      D,      D,      1,      1,      -D,        -D,       D,        D};
        const int fc[12] = {1,     1,    D,      D,      -1,     1,      -1,     1,      -1,        1,        -1,       1};
        const int fo[12] = {0,     0,    o,      0,      o,      0,      o,      0,      o * D + o, o * D,    o,        0};
        const int fs[12] = {NN,    NN,   N * D,  N * D,  N * D,  N * D,  N * D,  N * D,  D * D,     D * D,    D * D,    D * D};
        int start = 0;
#pragma unroll
        for (int f = 0; f < 12; f++) {
            if ((f & 3) == gl) cf_a[f >> 2] = fa[f], cf_b[f >> 2] = fb[f], cf_c[f >> 2] = fc[f], cf_base[f >> 2] = start + fo[f];
            start += fs[f];
        }
    }
    if (MODE == MCQ_MODE_BOARD) {
        for (int c = gl; c < Q; c += G) hts[c] = rst[c];
        if constexpr (CNT) {  // count the initial board: every lane walks all columns for its own three families (disjoint bytes: plain read-modify-write)
            const int total = 2 * NN + 6 * N * (2 * N - 1) + 4 * (2 * N - 1) * (2 * N - 1);
            for (int w = gl; w < (total + 3) / 4; w += G) ((uint32_t*)cnt)[w] = 0;
            for (int c = 0; c < Q; c++) {
                const int ci = c / N, cj = c - ci * N, ck = rst[c];
#pragma unroll
                for (int f = 0; f < 3; f++) cnt[cf_base[f] + cf_a[f] * ci + cf_b[f] * cj + cf_c[f] * ck] += 1;
            }
        }
    } else {
        uint32_t* cw32 = base + L_STATE;  // the column table and its pads as 32-bit words
        const int cwords = (int)(((2 * full_pad + NN) * sizeof(colw_t) + 3) / 4);  // (rounded up: an odd count of 16-bit words ends in half a word)
        for (int w = gl; w < cwords; w += G) cw32[w] = 0;
        for (int c = gl; c < Q; c += G) {
            const uint32_t qi_ = rst[3 * c], qj_ = rst[3 * c + 1], qk_ = rst[3 * c + 2];
            if constexpr (!SLIM && !WIDE) qn[c] = (uint16_t)(qi_ | (qj_ << 5) | (qk_ << 10));
            const uint32_t e = (uint32_t)full_pad + qi_ * N + qj_;  // element index from the start of the table
            if constexpr (WIDE) atomicOr((unsigned long long*)cw32 + e, 1ull << qk_);
            else if (NARROW) atomicOr(&cw32[e >> 1], (1u << qk_) << ((e & 1u) * 16u));
            else atomicOr(&cw32[e], 1u << qk_);
        }
    }
    const unsigned mN = (unsigned)(N - 1), mQ = (unsigned)(Q - 1);
    const unsigned maskN = NC ? mask_for((unsigned)(NC - 1)) : a.maskN, maskQ = NC ? mask_for((unsigned)(NC * NC - 1)) : a.maskQ;
    uint32_t tc1 = 0x9d2c5680u, tc2 = 0xefc60000u;
    asm volatile("" : "+s"(tc1), "+s"(tc2));  // opaque scalars: no literal operands in the tempering
    const uint32_t rec_bytes = (uint32_t)a.rec_words * 4u;
    char* wave_base = (char*)(a.ws + (long long)blockIdx.x * CPW * (long long)a.rec_words);
    Stream<G, MODE == MCQ_MODE_FULL3D, PHILOX, !SLIM> rng;
    rng.attach(wave_base, active ? (uint32_t)grp * rec_bytes : 0u, ring, (int)rec[REC_POS], (int)rec[REC_GEN_END], gl, maskN, mN, maskQ, mQ, tc1, tc2);
    if constexpr (PHILOX) rng.attach_philox(a.seeds[crow], rec[REC_POS]);

    int E = (int)rec[REC_E0];
    int best = E;
    // steps_to_best, n_accepted, near ties and the history length change rarely: they live in LDS, not in registers (every lane of
    // a group performs the same read-modify-write in lockstep)
    int* cold = (int*)(base + L_COLD);
    enum { C_BEST_STEP = 0, C_N_ACC = 1, C_TIES = 2, C_HIST_LEN = 3 };
    cold[C_BEST_STEP] = 0, cold[C_N_ACC] = 0, cold[C_TIES] = 0, cold[C_HIST_LEN] = active ? (int)a.n_steps + 1 : 0;
    unsigned long long* red = reduced ? a.red + (a.chains_per_set > 0 ? ((long long)blockIdx.x * CPW) / a.chains_per_set : 0) * a.red_set_stride +
                                            (long long)(blockIdx.x & (RED_STRIPES - 1)) * 3 * a.red_len
                                      : nullptr;
    uint32_t accw = 0;  // accept bits of the current block of 32 steps
    const bool exact_only = (a.flags & MCQ_FLAG_EXACT_EXP) != 0;
    // half-width of the float32 bracket around exp(x) 2^27 as w = e27 * w_scale + w_bias; MCQ_FLAG_EXACT_EXP makes it cover everything
#ifndef MCQ_W_SCALE_BITS
#define MCQ_W_SCALE_BITS 0x3a800000u /* 2^-10 */
#endif
    uint32_t w_scale_bits = exact_only ? 0u : MCQ_W_SCALE_BITS, w_bias_bits = exact_only ? 0x7f61b1e6u /* 3.0e38 */ : 0x3f000000u /* 0.5 */;
    asm volatile("" : "+s"(w_scale_bits), "+s"(w_bias_bits));  // two scalars, no select per step
    const float w_scale = __uint_as_float(w_scale_bits), w_bias = __uint_as_float(w_bias_bits);
    const bool force_slow = (a.flags & MCQ_FLAG_SEQUENTIAL_DRAWS) != 0;
    uint32_t batch_mask = force_slow ? 0u : 0xffffffffu;  // wave-uniform
    asm volatile("" : "+s"(batch_mask));
    uint32_t zero_mark = 0u;  // a zero the compiler cannot see through (load_probes)
    asm volatile("" : "+s"(zero_mark));
    const bool trace = a.out.energy_hist != nullptr;
    int flush_at = trace ? SB - 1 : 99;
    asm volatile("" : "+s"(flush_at));  // an opaque scalar: one compare per step, whatever the compiler could derive from the 99
    // trace rows: wave-uniform address of the wavefront's first row (scalar registers) + a 32-bit byte offset per lane, so that no
    // 64-bit pointer is held in vector registers (16 rows of hist_stride < 2^24 entries span < 2^30 bytes: checked by the host side)
    char* const hist_base = trace ? (char*)(a.out.energy_hist + (long long)blockIdx.x * CPW * a.hist_stride) : nullptr;
    const uint32_t hist_off = ((uint32_t)(active ? grp : 0) * (uint32_t)a.hist_stride + (uint32_t)(gl * WPL)) * 4u;
    auto hist_at = [&](int entry) { return (int32_t*)(hist_base + (hist_off + 4u * (uint32_t)entry)); };  // this lane's WPL entries from `entry` on
    const bool have_bits = a.out.accept_bits != nullptr;
    char* const bits_base = have_bits ? (char*)((uint32_t*)a.out.accept_bits + (long long)blockIdx.x * CPW * a.bits_stride * 2) : nullptr;
    const uint32_t bits_off = (uint32_t)(active ? grp : 0) * (uint32_t)a.bits_stride * 8u;
    auto bits_at = [&](int word) { return (uint32_t*)(bits_base + (bits_off + 4u * (uint32_t)word)); };  // 32-bit word `word` of the chain's row
    // Early stop (experiments.py:343-353): no_improvement_steps is reset by a strict improvement and grows by one on every other
    // step, so after step s it equals s - (step of the last improvement, -1 before any): the chain stops at the first step
    // s >= deadline, deadline = last improvement + patience -- a value that only changes in the (rare) improvement path, so the
    // common path pays one compare.  A patience beyond n_steps can never trigger and is clamped, which keeps the sum inside 32 bits.
    const uint32_t patience = a.patience < 0 || a.patience > a.n_steps ? (uint32_t)a.n_steps + 1u : (uint32_t)a.patience;
    uint32_t deadline = (patience ? patience : 1u) - 1u;  // patience 0 stops at step 0 whatever happens there

    if constexpr (REDUCED) {  // stage words carry a `valid` bit (reduce_block): nothing is valid yet, idle groups never are
#pragma unroll
        for (int w = 0; w < WPL; w++) stage[gl * WPL + w] = 0u;
        stage[0] = active ? (uint32_t)E | 0x40000000u : 0u;
    } else {
        stage[0] = (uint32_t)E;  // history entry 0 = E0 (every lane of the group writes the same word)
    }
    if (active) {
        if (gl == 0 && a.out.initial_energy) a.out.initial_energy[chain] = E;
        if (a.out.best_state) {
            uint8_t* bo = a.out.best_state + crow * (long long)state_bytes;
            for (int c = gl; c < state_bytes; c += G) bo[c] = rst[c];
        }
    }

    // the tables were written by an earlier kernel and are read-only here: the constant address space
    // lets the compiler fetch them with scalar loads (one s_load per step instead of a vector load)
    typedef const __attribute__((address_space(4))) float* const_f32_ptr;
    typedef const __attribute__((address_space(4))) double* const_f64_ptr;
    // schedule sets: every chain of a wavefront belongs to the same set (chains_per_set is a multiple of 16)
    const long long set_idx = a.chains_per_set > 0 ? ((long long)blockIdx.x * CPW) / a.chains_per_set : 0;
    const const_f32_ptr c32_tab = (const_f32_ptr)(unsigned long long)(a.c32_tab + set_idx * a.tab_stride);
    const const_f64_ptr beta_tab = (const_f64_ptr)(unsigned long long)(a.beta_tab + set_idx * a.tab_stride);

    // Packed dE probes (N <= 16 <=> NT <= 4): lane constants of the NT columns / rows this lane probes.  Board: the G
    // lanes of a chain share the 4N probes; full_3d: half of the lanes probe around the new cell, the other half around the old.
    //   pm      probe index m, clamped into the board (a clamped probe has all-zero selectors below)
    //   krc     selector of the row / column probe (board: one bit per half; full_3d: everything)
    //   vdm     bit (j - i + 16) set iff the diagonal probe (m, m - i + j) is on the board
    //   vam     bit (i + j) set iff the anti-diagonal probe (m, i + j - m) is
    constexpr bool PACKED = (MODE == MCQ_MODE_BOARD && NT >= 1 && NT * G <= 16) || NARROW;  // N <= 16
    constexpr int PG = NARROW ? G / 2 : G;  // lanes that share one set of probes
    constexpr bool UNROLLED = PACKED || (MODE == MCQ_MODE_BOARD && NT >= 1 && NT <= 4);  // probe passes with per-lane constants (boards beyond N = 16: pm and krc only; five or six passes would spill)
    constexpr int NTP = UNROLLED ? NT : 1;
    int pm[NTP];
    uint32_t krc[NTP], vdm[NTP], vam[NTP];
#pragma unroll
    for (int t = 0; t < NTP; t++) {
        const int m = (gl & (PG - 1)) + t * PG;
        // only the last pass can leave the board: a variant with NT passes runs for ceil(N / PG) == NT, i.e. N > (NT - 1) * PG
        const bool inb = t + 1 < NTP || (NC != 0 && NTP * PG <= NC) || m < N;
        const uint32_t full = N >= 32 ? 0xffffffffu : (1u << N) - 1u;
        pm[t] = inb ? m : N - 1;
        krc[t] = inb ? (NARROW ? 0xffffffffu : 0x00010001u) : 0u;
        vdm[t] = inb && PACKED ? full << ((16 - pm[t]) & 31) : 0u;
        vam[t] = inb && PACKED ? full << (pm[t] & 31) : 0u;
    }

    // Stream upkeep cadence: demand-driven.  The upkeep code (finish the block in flight, request the next one) runs in a step
    // when some chain of the wavefront is down to fewer than LOW_WATER ready words -- every chain with work pending is served then -- and
    // is skipped otherwise.  A board step uses 3 (mask + 1) / N + 2 words on average (6.1 at N = 12, 7.7 at N = 17), a full_3d
    // step 8.1, and a block brings 16, so this runs every second or third step with most lanes busy, instead of on a fixed
    // cadence with half of them idle (full_3d +9 %, board N = 17 +10 %, N = 16 +5 %).  The ring can never be overrun: a block
    // needs gen - pos <= 48 when it lands, and it is requested at <= 53 (board, >= 5 words used per step) or <= 54 (full_3d, >= 6)
    // at least one step earlier; running dry is handled by the sequential path, which services the stream itself.
    // (PHILOX: the block is generated on the spot, so it needs its room right away: gen - pos <= 48)
    const uint32_t room_limit = PHILOX ? 48u : MODE == MCQ_MODE_BOARD ? 53u : 54u;
    // (A/B on one box, 100 000 steps: board N = 12 low water 16 / 20 / 24 / 28 / 32 / 36 -> 140.3 / 137.1 / 137.4 / 138.3 / 137.9* / 140.5* ms,
    // N = 24 the same shape; full_3d 20 / 24 / 28 / 32 -> 408.9 / 374.8 / 369.8 / 373.9 ms.  * = another box, shipped 135.4 there.)
    // Sizes whose randint(0, N) rejects many words (N / (mask + 1) < 0.6: N = 9, 17, 18, 19) look further ahead for their five
    // accepted words and run into a short ring more often: their mark is 28 (lone wavefront, 20 000 steps: N = 9 13.4 -> 12.7 ms,
    // N = 17 19.4 -> 18.1 ms, N = 18 18.1 -> 17.5 ms; every other size loses 1-2 % at 28: profiles/r03_low_water.txt).  A run-time scalar.
#ifdef MCQ_LOW_WATER
    const uint32_t LOW_WATER = MCQ_LOW_WATER;
#else
    const uint32_t LOW_WATER = (uint32_t)a.low_water;
#endif

    // replica exchange: the chain's rung, the float32 image of its beta multiplier, accepted swaps; wave-uniform countdown and parity
    int rung = EXCH ? grp % a.exch_R : 0, n_exch = 0;
    float m32 = EXCH && active ? (float)a.exch_ladder[rung] : 1.0f;
    int xcount = EXCH ? (int)(a.exch_every > 2147483647LL ? 2147483647LL : a.exch_every) : 0;
    int xpar = 1;  // exchange n = 1, 2, ... offers the rung pairs (t, t + 1) with t = n & 1 (mod 2)

    // A launch that is too small to pace itself (a.pace is null then) runs at the priority its caller names: a job list of launches
    // side by side gives its long launches precedence, so that they run at the pace of a lone wavefront while the short ones fill
    // the gaps, instead of everybody sharing alike and the long ones finishing on an empty device (jobs.JobSet; MCQ_FLAG_PRIORITY).
    if (!a.pace) set_priority((int)((a.flags >> MCQ_FLAG_PRIORITY_SHIFT) & 3u));

    STAMP_DECL;
    const int n_steps = (int)a.n_steps;
    int last_entry = n_steps;  // wave-uniform: the last history entry any chain of the wavefront can have reached
    // One Metropolis step of the wavefront's chains.  Where lanes can sit a step out -- chains that have stopped early, the idle
    // groups a reduced-trace wavefront keeps for its reductions -- the whole step becomes a divergent region: ~27 scalar and ~11
    // vector instructions more per step for exec-mask bookkeeping (PMC: 64 / 233 against 38 / 222), and a loop counter the compiler
    // no longer sees as uniform.  So the step exists twice: ALL = true assumes every lane takes part (no `if (active)`, plain scalar
    // counter) and runs while that holds -- for the API's default early_stop_patience = 100000 it always does; ALL = false is the
    // general form, its counter kept scalar through readfirstlane.  Returns STEP_FINISHED when the wavefront has nothing left to do.
    int vstep = 0;
    enum { STEP_GO_ON = 0, STEP_FINISHED = 1 };
    auto metropolis_step = [&](auto all_tag) __attribute__((always_inline)) -> int {
        constexpr bool ALL = decltype(all_tag)::value;
        STAMP(0);  // loop overhead + previous step's tail
        const int step = ALL ? vstep : __builtin_amdgcn_readfirstlane(vstep);
        vstep = step + 1;
        const float c32s = c32_tab[(uint32_t)step];  // exp(-beta dE) = exp2(dE * c32)  (unsigned index: a scalar load with a 32-bit offset register)
        const float c32 = EXCH ? c32s * m32 : c32s;  // replica exchange: the chain runs at beta(step) * ladder[rung]
        // what closes a step for the whole wavefront (stopped and idle lanes included): REDUCED adds every 16th block of entries
        // to the accumulators.  SOME_INACTIVE = false: every lane is a live chain (the ALL form's common path).
        auto end_of_step = [&](auto some_inactive) __attribute__((always_inline)) {
            STAMP(5);  // apply + history
            if (reduced && ((step + 1) & 15) == 15) {
                reduce_block<G>(lds + LDS_STAGE, a.chain_lds_words, lane, (step + 1) & ~15, a.n_steps + 1, red, a.red_len);
                if constexpr (decltype(some_inactive)::value) {
                    if (!active) {  // a stopped (or idle) chain's block must not be counted again
#pragma unroll
                        for (int w = 0; w < WPL; w++) stage[gl * WPL + w] = 0u;
                    }
                }
            }
        };

        if (ALL || active) {
            // ---- proposal draws -----------------------------------------------------------------
            // board   (experiments.py:311-327): i, j, new_k (redrawn while == old_k), then random()
            // full_3d (experiments.py:221-239): q, (i, j, k) redrawn while the cell is occupied, random()
            int pa = 0, pb = 0, pc = 0;  // board: i, j, new_k      full_3d: ni, nj, nk
            int cell = 0, old_k = 0;     // board
            int qi = 0;                  // full_3d
            uint32_t oldp = 0;
            uint32_t uw1 = 0, uw2 = 0;   // the two words of random()
            int stage_no = 0;
            int redraw_from = -1;        // board: words to skip when the word-by-word draw can go on from what the batched attempt established
            uint32_t seen = 0;           // full_3d: what the batched attempt established (words to skip | stage to go on from << 8; 0: nothing)
            // stream upkeep runs for every chain of the wavefront together (cadence: see LOW_WATER above)
            auto upkeep = [&]() {
                STAMP(0);
                STAMP_COUNT(11);  // (diagnostic build: upkeep events of the wavefront)
                if constexpr (PHILOX) {
                    if (rng.gen - rng.pos <= room_limit) rng.generate();
                } else {
                    if (rng.pending) rng.complete();
                    // the block lands at a later upkeep, after at least one more step's words were consumed, or earlier only
                    // if the ring ran dry: there is room for its 16 words
                    if (rng.gen - rng.pos <= room_limit) rng.issue();
                }
                STAMP(1);  // stream upkeep: complete + issue
            };
            // sequential draws: one word at a time from what the ring holds, topping it up when it runs dry
            auto sequential = [&]() {
                bool service = false;
                for (;;) {
                    if (service) upkeep();
                    service = true;  // a second pass means the ring ran dry: finish the block in flight now
                    while (rng.pos != rng.gen && stage_no < LAST) {
                        const uint32_t w = ring[rng.pos & (RING - 1)];
                        rng.consume(1u);
                        const int vN = (int)(w & maskN);
                        const bool okN = (unsigned)vN <= mN;
                        if (MODE == MCQ_MODE_BOARD) {
                            if (stage_no == 0) {
                                if (okN) pa = vN, stage_no = 1;
                            } else if (stage_no == 1) {
                                if (okN) pb = vN, cell = pa * N + pb, old_k = hts[cell], stage_no = 2;
                            } else if (stage_no == 2) {
                                if (okN && vN != old_k) pc = vN, stage_no = 3;
                            } else if (stage_no == 3) {
                                uw1 = w, stage_no = 4;
                            } else {
                                uw2 = w, stage_no = 5;
                            }
                        } else {
                            if (stage_no == 0) {
                                const unsigned vQ = w & maskQ;
                                if (vQ <= mQ) qi = (int)vQ, oldp = qn[qi], stage_no = 1;
                            } else if (stage_no == 1) {
                                if (okN) pa = vN, stage_no = 2;
                            } else if (stage_no == 2) {
                                if (okN) pb = vN, stage_no = 3;
                            } else if (stage_no == 3) {
                                if (okN) {
                                    pc = vN;
                                    stage_no = ((colw[pa * N + pb] >> pc) & 1u) ? 1 : 4;  // occupied: draw the triple again
                                }
                            } else if (stage_no == 4) {
                                uw1 = w, stage_no = 5;
                            } else {
                                uw2 = w, stage_no = 6;
                            }
                        }
                    }
                    if (stage_no == LAST) break;
                }
            };

            // packed dE probes: the probed heights depend on (i, j) only; with three passes they are requested together with
            // the old height (with four, the 16 extra live registers would spill)
            // (Boards beyond N = 16 at 8 lanes -- three unpacked passes -- do the same where it pays, EARLYU: their twelve byte reads
            // otherwise sit behind the rare-path branch, one LDS round trip later than they have to.  A wavefront alone on its SIMD gains
            // 7.5 % (N = 24: 14.14 -> 13.08 ms per 20 000 steps), two per SIMD 2.6 % (config 5's per-GPU shape: 89.2 -> 86.9 ms), a full
            // device LOSES 4 % (552 against 531 ms: twelve more live registers across the draw), so the launcher picks the variant by the
            // wavefronts the launch puts on a SIMD.  With the reduced trace the packed variants request early too since round 3:
            // +0.9 % at N = 12.  profiles/r03_early_probes.txt)
#ifdef MCQ_EXP_NO_EARLY  // timing experiment: no early requests at all
            constexpr bool EARLY_PROBES = false;
#else
#ifndef MCQ_G2_EARLY_NT  // (two lanes per chain: the register budget of two wavefronts per SIMD holds the probed heights of up to eight passes)
#define MCQ_G2_EARLY_NT 8
#endif
            constexpr bool EARLY_PROBES = MODE == MCQ_MODE_BOARD && NT >= 1 && NT <= (G == 2 ? MCQ_G2_EARLY_NT : 3) && (PACKED || EARLYU);
#endif
            uint32_t ph[4 * NTP];
            // (`mark`: 0 where the probes are requested early, an opaque zero in the rare path's second request.  With two plain requests the
            // compiler merges the byte loads as 8-bit values and widens them again behind the merge: twelve `v_and 0xff` per step in the common
            // path of the unpacked early variants, 5 % of their vector instructions.  The OR keeps the merge 32 bits wide and costs the rare path only.)
            auto load_probes = [&](uint32_t mark) {
                const uint8_t* hrow = hts + __mul24(pa, N);
                const uint8_t* hj = hts + pb;
#pragma unroll
                for (int t = 0; t < NTP; t++) {
                    const int m = pm[t], mN_ = __mul24(m, N);
                    ph[4 * t] = hrow[m] | mark, ph[4 * t + 1] = hj[mN_] | mark, ph[4 * t + 2] = hj[mN_ + m - pa] | mark, ph[4 * t + 3] = hj[mN_ - m + pa] | mark;
                }
            };

            const bool upkeep_now = wave_any(rng.gen - rng.pos < LOW_WATER);
            bool batched;  // the batched draw below succeeded for this chain
            if constexpr (MODE == MCQ_MODE_BOARD) {
                // Straight-line for every chain of the wavefront (no divergent branch): positions of the next five
                // accepted words inside the 32 ring slots that follow pos, i / j / three candidates for new_k, the old
                // height, and the uniform's two words behind the chosen candidate.  Where the attempt is not valid the
                // fetched values are simply not used (every address is inside the chain's LDS slice).
                const uint32_t s = rng.pos & (RING - 1);
                // (positions 0..29 from the second word on -- the mask rides in the three-input AND: whichever candidate is taken, its
                // two followers are inside the view; a first word at 30 / 31 leaves no second one and the attempt is not used)
                const uint32_t v1 = (uint32_t)rng.ok;
                const uint32_t v2 = v1 & (v1 - 1) & 0x3fffffffu, v3 = v2 & (v2 - 1), v4 = v3 & (v3 - 1), v5 = v4 & (v4 - 1);
                const uint32_t avail = rng.gen - rng.pos;  // ring slots [pos, pos + avail) hold words
                // an empty mask gives position -1 (the attempt is not used then): the fetches below read ring[s - 1 ..], still the chain's own LDS
                const int p1 = lowest_bit(v1), p2 = lowest_bit(v2), p3 = lowest_bit(v3), p4 = lowest_bit(v4), p5 = lowest_bit(v5);
                // The candidate that becomes new_k must leave its two followers (the uniform's words) inside the generated words:
                // kp + 2 < avail.  As ONE unsigned compare, kp < draw_limit: a position that
                // does not exist (-1 = 0xffffffff) fails it, and so does everything under MCQ_FLAG_SEQUENTIAL_DRAWS (limit 0).  (Flags
                // exist for generated words only, so a real position is < avail and avail - 2 cannot have wrapped for it.)  The test is
                // on the candidate that is TAKEN -- the first one, (N - 1) / N of the time -- not on the last one that might be: sizes
                // whose randint rejects many words (N = 9, 17: 44 % rejected) found five accepted words inside a short ring far less
                // often than three.
                const uint32_t draw_limit = (avail - 2u) & batch_mask;
                // no wrap-around: s <= 63 and every position is <= 31, inside the mirrored ring.  The uniform's two words follow the
                // candidate that becomes new_k: they are fetched once that is known (the step is bound by instruction issue, not by
                // this round trip: fetching them behind all three candidates up front cost 6 selects and 2 LDS instructions more).
                const uint32_t* rs = ring + s;
                const uint32_t w1 = rs[p1], w2 = rs[p2], w3 = rs[p3], w4 = rs[p4], w5 = rs[p5];
                // CAND5 (N <= 5): two more candidates.  All candidates equal old_k with probability 1 / N^3 per chain -- 4 % at N = 3,
                // so in 45 % of the steps of a 16-chain wavefront some chain went word by word; with five it is 1 / N^5.
                int p6 = -1, p7 = -1;
                uint32_t w6 = 0, w7 = 0;
                if constexpr (CAND5) {
                    const uint32_t v6 = v5 & (v5 - 1), v7 = v6 & (v6 - 1);
                    p6 = lowest_bit(v6), p7 = lowest_bit(v7);
                    w6 = rs[p6], w7 = rs[p7];
                }
                // the stream upkeep runs while those reads are in flight; it appends behind the words of this view
                if (upkeep_now) upkeep();
                pa = (int)(w1 & maskN), pb = (int)(w2 & maskN);
                const int c3 = (int)(w3 & maskN), c4 = (int)(w4 & maskN), c5 = (int)(w5 & maskN);
                cell = __mul24(pa, N) + pb;  // < 2^10 bytes: whatever the words were, inside the workgroup's LDS
                old_k = hts[cell];
                if constexpr (EARLY_PROBES) load_probes(0u);
                const bool use3 = c3 != old_k, use4 = c4 != old_k;  // new_k is redrawn while it equals old_k (experiments.py:318-319)
                int kp, p_last;
                if constexpr (CAND5) {
                    const int c6 = (int)(w6 & maskN), c7 = (int)(w7 & maskN);
                    const bool use5 = c5 != old_k, use6 = c6 != old_k;
                    pc = use3 ? c3 : use4 ? c4 : use5 ? c5 : use6 ? c6 : c7;
                    kp = use3 ? p3 : use4 ? p4 : use5 ? p5 : use6 ? p6 : p7;
                    p_last = p7;
                } else {
                    pc = use3 ? c3 : use4 ? c4 : c5;
                    kp = use3 ? p3 : use4 ? p4 : p5;
                    p_last = p5;
                }
                uw1 = rs[kp + 1], uw2 = rs[kp + 2];  // (an unused attempt reads some words of the chain's ring: kp >= -1)
                // every candidate equal to old_k (1/N^3): word by word instead
                uint32_t taken = pc != old_k ? (uint32_t)kp : 0xffffffffu;
                asm volatile("" : "+v"(taken));  // (keeps it one compare: the compiler would take the select apart into mask logic again)
                batched = taken < draw_limit;
                rng.consume(batched ? (uint32_t)kp + 3u : 0u);
                redraw_from = (uint32_t)p_last < draw_limit ? p_last + 1 : -1;  // (dead outside the rare path below)
            } else {
                // q = first word accepted for randint(0, Q); then a candidate (i, j, k) triple from the words accepted for
                // randint(0, N) after it, and a second triple when the first cell is occupied; the uniform's two words follow the
                // chosen triple.
                const uint32_t s = rng.pos & (RING - 1);
                const uint32_t vq = (uint32_t)rng.okq;
                const int pq = lowest_bit(vq);  // an empty mask gives -1 (and leaves n1 empty), like the positions below
                const uint32_t n1 = (uint32_t)rng.ok & (0xFFFFFFFEu << ((uint32_t)pq & 31u));
                const uint32_t n2 = n1 & (n1 - 1), n3 = n2 & (n2 - 1);
                const uint32_t avail = rng.gen - rng.pos;
                // the third word must leave its two followers (the uniform's) inside the view: positions 0..29 (none there: p3 = -1)
                const int p1 = lowest_bit(n1), p2 = lowest_bit(n2), p3 = lowest_bit(n3 & 0x3fffffffu);
                const uint32_t draw_limit = (avail - 2u) & batch_mask;  // p + 2 < avail as p < draw_limit (see the board branch)
                // no wrap-around: s <= 63 and every position is <= 31, inside the mirrored ring (SLIM: no mirror, the slot index wraps)
                const uint32_t* rs = ring + s;
                auto rd = [&](int p) __attribute__((always_inline)) -> uint32_t {
                    if constexpr (SLIM) return ring[(s + (uint32_t)p) & (RING - 1)];
                    else return rs[p];
                };
                const uint32_t wq = rd(pq);
                const uint32_t w1 = rd(p1), w2 = rd(p2), w3 = rd(p3);
                if (upkeep_now) upkeep();  // while those reads are in flight
                const uint32_t vqi = wq & maskQ;
                qi = (int)min(vqi, mQ);  // (a used attempt has vqi <= mQ; an unused one must still index inside the queen table)
                // (SLIM: a 2-byte read from global memory at the head of the step's dependency chain.  Requesting it a whole step ahead --
                // the position of the next step's first randint(0, Q) word is known once a step's draws are done -- measured nothing at four
                // wavefronts per SIMD: 232.8 against 232.4 ms, profiles/r04_full3d_slim.txt)
#if defined(MCQ_EXP_QN_SC1)  // timing experiment: the queen read with agent scope (sc1): does the L2 then ask the fabric for less than a 128-byte line?
                if constexpr (SLIM) oldp = __hip_atomic_load(qn + qi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else oldp = qn[qi];
#elif defined(MCQ_EXP_QN_NT)
                if constexpr (SLIM) oldp = __builtin_nontemporal_load(qn + qi);
                else oldp = qn[qi];
#else
                oldp = qn[qi];
#endif
                pa = (int)(w1 & maskN), pb = (int)(w2 & maskN), pc = (int)((w3 & maskN) & QM);
                // (WIDE: an attempt that is not used may hold any value up to the mask, 63 -- at N = 33 a column index far behind the chain's table,
                // and for the wavefront's last chain behind the workgroup's LDS: clamped, here and for the other two triples)
                if constexpr (WIDE) pa = min(pa, (int)mN), pb = min(pb, (int)mN);
                const cword_t cw1 = colw[__mul24(pa, N) + pb];  // word index < 2^10 (WIDE: 2^12): inside the workgroup's LDS
                const bool free1 = !((cw1 >> pc) & 1u);
                int pu = p3;
                uint32_t third = free1 ? (uint32_t)p3 : 0xffffffffu;
                asm volatile("" : "+v"(third));  // (keeps it one compare, as in the board branch)
                batched = third < draw_limit;
                // The cell of the first triple is taken with probability Q / N^3 (1/12 at N = 12): only then -- for some chain of the
                // wavefront, so in every second step of 8 chains -- is the second triple looked at (it was fetched up front before:
                // ~30 instructions per step; the step is bound by instruction issue, not by this extra round trip).
                if (wave_any(!batched)) {
                    STAMP_COUNT(10);
                    const uint32_t n4 = n3 & (n3 - 1), n5 = n4 & (n4 - 1), n6 = n5 & (n5 - 1);
                    const int p4 = lowest_bit(n4), p5 = lowest_bit(n5), p6 = lowest_bit(n6 & 0x3fffffffu);
                    const uint32_t w4 = rd(p4), w5 = rd(p5), w6 = rd(p6);
                    int i2 = (int)(w4 & maskN), j2 = (int)(w5 & maskN);
                    const int k2 = (int)((w6 & maskN) & QM);
                    if constexpr (WIDE) i2 = min(i2, (int)mN), j2 = min(j2, (int)mN);
                    const cword_t cw2 = colw[__mul24(i2, N) + j2];
                    // (a position counts only when its word has been generated; nothing does under MCQ_FLAG_SEQUENTIAL_DRAWS)
                    const uint32_t availm = avail & batch_mask;
                    const bool valid3 = (uint32_t)p3 < availm, valid6 = (uint32_t)p6 < availm, taken2 = ((cw2 >> k2) & 1u) != 0;
                    const bool use2 = !batched && !free1 && valid6 && !taken2;  // the second triple's cell is the new cell
                    const bool second = use2 && (uint32_t)p6 < draw_limit;      // ... and its uniform's words are there
                    pa = use2 ? i2 : pa, pb = use2 ? j2 : pb, pc = use2 ? k2 : pc, pu = second ? p6 : pu;
                    batched = batched || second;
                    if (wave_any(!batched)) {  // (one time in nine: some chain's second cell is taken too, or its ring is short)
                    // What the word-by-word path needs to know should it be taken (see below): the words to skip and the stage to go on
                    // from -- a new triple (1) or, the last triple's cell being free, only the uniform (4).  (valid6 implies valid3.)
                    const uint32_t skip = (uint32_t)((free1 || !valid6) ? p3 : p6) + 1u;
                    seen = valid3 ? skip | ((free1 || (valid6 && !taken2)) ? 4u << 8 : 1u << 8) : 0u;
                    // Both cells taken (Q^2 / N^6 per chain: 5.5 % of the steps of an 8-chain wavefront at N = 12): a third triple, the
                    // same way, before the chain goes word by word -- that path stalled the wavefront for 6.6 % of config 3's time
                    // (timing-only build that treats the second cell as free: 62.49 -> 58.37 ms per 20 000 steps).
                    const bool both = !free1 && valid6 && taken2;
                    if (wave_any(both)) {
                        const uint32_t n7 = n6 & (n6 - 1), n8 = n7 & (n7 - 1), n9 = n8 & (n8 - 1);
                        const int p7 = lowest_bit(n7), p8 = lowest_bit(n8), p9 = lowest_bit(n9 & 0x3fffffffu);
                        const uint32_t w7 = rd(p7), w8 = rd(p8), w9 = rd(p9);
                        int i3 = (int)(w7 & maskN), j3 = (int)(w8 & maskN);
                        const int k3 = (int)((w9 & maskN) & QM);
                        if constexpr (WIDE) i3 = min(i3, (int)mN), j3 = min(j3, (int)mN);
                        const cword_t cw3 = colw[__mul24(i3, N) + j3];
                        const bool look3 = both && (uint32_t)p9 < availm, taken3 = ((cw3 >> k3) & 1u) != 0;
                        const bool use3 = look3 && !taken3;
                        const bool third_ok = use3 && (uint32_t)p9 < draw_limit;
                        pa = use3 ? i3 : pa, pb = use3 ? j3 : pb, pc = use3 ? k3 : pc, pu = third_ok ? p9 : pu;
                        batched = batched || third_ok;
                        seen = look3 ? ((uint32_t)p9 + 1u) | (taken3 ? 1u << 8 : 4u << 8) : seen;
                    }
                    }
                }
                uw1 = rd(pu + 1), uw2 = rd(pu + 2);  // the uniform's words follow the chosen triple (pu >= -1: inside the chain's ring)
                rng.consume(batched ? (uint32_t)pu + 3u : 0u);
            }
            STAMP_COUNT(7);  // (diagnostic build: wavefront-steps, and those of them that take the word-by-word path)
            if (__builtin_expect(wave_any(!batched), 0)) {  // wave-uniform guard of the rare path
                STAMP_COUNT(6);
                if (!batched) {
                    if constexpr (MODE == MCQ_MODE_BOARD) {
                        // Three candidates in a row equal to old_k, all of them inside the generated words (1/N^3 per chain: 4 % at
                        // N = 3, so 45 % of the steps of a 16-chain wavefront): i, j and old_k stand, and the word-by-word draw goes
                        // on behind the third candidate instead of starting the proposal over.
                        if (pc == old_k && redraw_from >= 0) {
                            rng.consume((uint32_t)redraw_from);
                            stage_no = 2;
                        }
                    } else if (seen) {
                        // The cells of the batched triples are all taken, or the ring was short: q and the triples whose words are all
                        // there stand.  The draw goes on behind the last of them -- with a new triple (stage 1) or, its cell being free,
                        // only the uniform (stage 4) -- instead of starting over.
                        rng.consume(seen & 0xffu);
                        stage_no = (int)(seen >> 8);
                    }
                    sequential();
                    if constexpr (SLIM) asm volatile("" : "+v"(oldp));  // (a queen read from global memory in there is claimed in here: no wait at the merge, where the common path has the stream's block in flight)
                    if constexpr (EARLY_PROBES) load_probes(PACKED ? 0u : zero_mark);  // (the packed variants shift by the byte: no widening to pay for)
                }
            }

            STAMP(2);  // proposal draws
            // ---- dE -------------------------------------------------------------------------------
            int dE;
            uint32_t newp = 0;
            cword_t cw_new = 0, cw_old = 0;  // full_3d: occupancy words of the new and of the old cell's column
            int cix[6] = {0, 0, 0, 0, 0, 0};  // CNT: the lane's three old-cell and three new-cell lines ...
            int cvl[6] = {0, 0, 0, 0, 0, 0};  // ... and their counts, kept for the write-back of an accepted move
            if constexpr (CNT) {
                // conflicts_for_position(i, j, new_k) - conflicts_for_position(i, j, old_k) (mcmc_board.py:147-193) from the line counters
                int part = 0;
#pragma unroll
                for (int f = 0; f < 3; f++) {
                    const int t = cf_base[f] + __mul24(cf_a[f], pa) + __mul24(cf_b[f], pb);
                    cix[f] = t + __mul24(cf_c[f], old_k), cix[3 + f] = t + __mul24(cf_c[f], pc);
                    cvl[f] = cnt[cix[f]], cvl[3 + f] = cnt[cix[3 + f]];
                    part += cvl[3 + f] - cvl[f];
                }
                dE = group_sum<G>(part) + 12;
            } else if (MODE == MCQ_MODE_BOARD) {
                // dE = conflicts(new_k) - conflicts(old_k) (mcmc_board.py:147-193).  Probe the columns on the
                // four lines of the ij-plane through (i, j).  Column (i2, j2) at distance d holds height h; it
                // attacks (i, j, k) iff h - k is 0 or +-d.  The cell (i, j) itself is probed once per direction
                // and always scores -1 (it holds old_k): +4 below.
                const int i = pa, j = pb;
                const uint32_t Bo = 1u << old_k, Bn = 1u << pc;
                // LDS byte addresses: row i starts at hrow; (m, j) = hj + m*N; (m, m-i+j) = hd + m*(N+1); (m, i+j-m) = ha + m*(N-1)
                const uint8_t* hrow = hts + __mul24(i, N);
                const uint8_t* hj = hts + j;
                const uint8_t* hd = hj - i;
                const uint8_t* ha = hj + i;
                const int dji = j - i, sij = i + j;
                int acc = 0;
                auto probe = [&](int m, bool in_board, uint32_t hr, uint32_t hc, uint32_t hdg, uint32_t han) {
                    const uint32_t dr = abs_diff(m, j), dc = abs_diff(m, i);
                    const uint32_t Mor = Bo | (Bo << dr) | (Bo >> dr), Mnr = Bn | (Bn << dr) | (Bn >> dr);
                    const uint32_t Moc = Bo | (Bo << dc) | (Bo >> dc), Mnc = Bn | (Bn << dc) | (Bn >> dc);
                    // +1 for a hit on the new height, -1 (sign-extended bit) for a hit on the old one
                    const int cr = (int)__builtin_amdgcn_ubfe(Mnr, hr, 1) + __builtin_amdgcn_sbfe((int)Mor, hr, 1);
                    const int cc = (int)__builtin_amdgcn_ubfe(Mnc, hc, 1) + __builtin_amdgcn_sbfe((int)Moc, hc, 1);
                    // diagonals: out-of-board probes read some other byte of the chain's LDS slice and are discarded
                    const int cd = (int)__builtin_amdgcn_ubfe(Mnc, hdg, 1) + __builtin_amdgcn_sbfe((int)Moc, hdg, 1);
                    const int ca = (int)__builtin_amdgcn_ubfe(Mnc, han, 1) + __builtin_amdgcn_sbfe((int)Moc, han, 1);
                    acc += in_board ? cr + cc : 0;
                    acc += in_board && (unsigned)(m + dji) < (unsigned)N ? cd : 0;
                    acc += in_board && (unsigned)(sij - m) < (unsigned)N ? ca : 0;
                };
                if constexpr (PACKED) {
                    // both heights ride in one register (pk_star); a probe shifts the pair of masks right by the probed
                    // height, which leaves "attacks old_k" in bit 0 and "attacks new_k" in bit 16, and the selected bits
                    // are summed as two 16-bit counters.  No select depends on a loaded value and none guards a load, so
                    // all 4*NT byte reads are in flight together.
                    const uint32_t BB = Bo | (Bn << 16);
                    const uint32_t shd = (uint32_t)(dji + 16), sha = (uint32_t)sij;
                    uint32_t accp = 0;
                    if constexpr (!EARLY_PROBES) load_probes(0u);
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const int m = pm[t];
                        const uint32_t hr = ph[4 * t], hc = ph[4 * t + 1], hdg = ph[4 * t + 2], han = ph[4 * t + 3];
                        const uint32_t Mr = pk_star(BB, abs_diff(m, j)), Mc = pk_star(BB, abs_diff(m, i));
                        const uint32_t vd = (uint32_t)__builtin_amdgcn_sbfe((int)vdm[t], shd, 1) & 0x00010001u;
                        const uint32_t va = (uint32_t)__builtin_amdgcn_sbfe((int)vam[t], sha, 1) & 0x00010001u;
                        accp += ((Mr >> (hr & 31u)) & krc[t]) + ((Mc >> (hc & 31u)) & krc[t]);
                        accp += ((Mc >> (hdg & 31u)) & vd) + ((Mc >> (han & 31u)) & va);
                    }
                    const uint32_t both = (uint32_t)group_sum<G>((int)accp);  // at most 4 * N hits per half: no carry
                    dE = (int)(both >> 16) - (int)(both & 0xffffu) + 4;
                } else {
                    if constexpr (UNROLLED) {  // the heights were requested by load_probes (a clamped probe reads cell N - 1 and is discarded)
                        if constexpr (!EARLY_PROBES) load_probes(0u);
#pragma unroll
                        for (int t = 0; t < NT; t++) probe(pm[t], t + 1 < NT || krc[t] != 0u, ph[4 * t], ph[4 * t + 1], ph[4 * t + 2], ph[4 * t + 3]);
                    } else if constexpr (NT > 0) {
#pragma unroll
                        for (int t = 0; t < NT; t++) {
                            const int m = gl + t * G, mN_ = __mul24(m, N);
                            probe(m, t + 1 < NT || m < N, hrow[m], hj[mN_], hd[mN_ + m], ha[mN_ - m]);
                        }
                    } else if (N <= 32) {
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                        for (int m = gl; m < N; m += G) {
                            const int mN_ = __mul24(m, N);
                            probe(m, true, hrow[m], hj[mN_], hd[mN_ + m], ha[mN_ - m]);
                        }
                    } else {
                        // boards beyond N = 32 (up to MCQ_MAX_N_BOARD): heights no longer index a 32-bit mask, so a probed height h at
                        // in-plane distance d is compared: it attacks k iff |h - k| is 0 or d (mcmc_board.py:177-191).  Slower per probe;
                        // these sizes are outside every BASELINE config.
                        auto hits = [&](uint32_t h, uint32_t d) {
                            const uint32_t xn = abs_diff((int)h, pc), xo = abs_diff((int)h, old_k);
                            return (int)(xn == 0u || xn == d) - (int)(xo == 0u || xo == d);
                        };
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                        for (int m = gl; m < N; m += G) {
                            const int mN_ = __mul24(m, N);
                            const uint32_t dr = abs_diff(m, j), dc = abs_diff(m, i);
                            acc += hits(hrow[m], dr) + hits(hj[mN_], dc);
                            acc += (unsigned)(m + dji) < (unsigned)N ? hits(hd[mN_ + m], dc) : 0;
                            acc += (unsigned)(sij - m) < (unsigned)N ? hits(ha[mN_ - m], dc) : 0;
                        }
                    }
                    dE = group_sum<G>(acc) + 4;
                }
            } else {
                // conflicts_for_queen(q, new) - conflicts_for_queen(q) (mcmc.py:185-226) without visiting the queens:
                // the 13 lines through a cell (ci,cj,ck) are the column itself plus, for every column (i2,j2) on the
                // four lines of the ij-plane through (ci,cj) at distance d, the heights ck and ck +- d.  With W the
                // occupancy word of that column, the attackers in it are popc(W & (B | B<<d | B>>d)), B = 1<<ck.
                const int ni = pa, nj = pb, nk = pc;
                const int oi = oldp & QM, oj = (oldp >> QS) & QM, ok_ = (oldp >> (2 * QS)) & QM;
                newp = (uint32_t)ni | ((uint32_t)nj << QS) | ((uint32_t)nk << (2 * QS));
                const cword_t Bo = (cword_t)1 << ok_, Bn = (cword_t)1 << nk;
                auto popc_w = [](cword_t x) -> int {
                    if constexpr (WIDE) return __popcll(x);
                    else return __popc(x);
                };
                auto star = [&](int m, bool in_board, int ci, int cj, cword_t B) {
                    const int mN_ = __mul24(m, N);
                    const colw_t* cw = colw + mN_;
                    const int jd = m - ci + cj, ja = ci + cj - m;
                    const cword_t wr = colw[__mul24(ci, N) + m], wc = cw[cj], wd = cw[jd], wa = cw[ja];
                    const uint32_t dr = abs_diff(m, cj), dc = abs_diff(m, ci);
                    const cword_t Mr = B | (B << dr) | (B >> dr), Mc = B | (B << dc) | (B >> dc);
                    int c = popc_w(wr & Mr) + popc_w(wc & Mc);
                    c += (unsigned)jd < (unsigned)N ? popc_w(wd & Mc) : 0;  // out-of-board probes read padding or another column: discarded
                    c += (unsigned)ja < (unsigned)N ? popc_w(wa & Mc) : 0;
                    return in_board ? c : 0;
                };
                int part = 0;
                if constexpr (NARROW) {
                    // Half of the chain's lanes probe around the new cell, the other half around the old one (which counts
                    // negative).  A lane packs the row and the column probe of its index m into one register (masks from one
                    // pk_star2 with the two distances), and the two diagonal probes into another (both at the column distance).
                    const bool oldside = gl >= PG;
                    const int ci = oldside ? oi : ni, cj = oldside ? oj : nj;
                    const uint32_t B = oldside ? Bo : Bn, BB = B | (B << 16);
                    const uint32_t shd = (uint32_t)(cj - ci + 16), sha = (uint32_t)(ci + cj);
                    const colw_t* crow = colw + times_N<NC>(ci, N);
                    uint32_t cnt = 0;
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const int m = pm[t];
                        const colw_t* cw = colw + __mul24(m, N);
                        const uint32_t wr = crow[m], wc = cw[cj], wd = cw[m - ci + cj], wa = cw[ci + cj - m];
                        const uint32_t Mrc = pk_star2(BB, abs_diff(m, cj) | (abs_diff(m, ci) << 16));
                        const uint32_t Mcc = __builtin_amdgcn_perm(Mrc, Mrc, 0x03020302u);  // the column-distance mask in both halves
                        const uint32_t vd = (uint32_t)__builtin_amdgcn_sbfe((int)vdm[t], shd, 1), va = (uint32_t)__builtin_amdgcn_sbfe((int)vam[t], sha, 1);
                        cnt += __popc((wr | (wc << 16)) & Mrc & krc[t]);
                        cnt += __popc(((wd & vd) | ((wa & va) << 16)) & Mcc);
                    }
                    part = oldside ? -(int)cnt : (int)cnt;
                } else if constexpr (NT > 0) {
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        const int m = gl + t * G;
                        const bool in_board = t + 1 < NT || m < N;
                        part += star(m, in_board, ni, nj, Bn) - star(m, in_board, oi, oj, Bo);
                    }
                } else {
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
                    for (int m = gl; m < N; m += G) part += star(m, true, ni, nj, Bn) - star(m, true, oi, oj, Bo);
                }
                // The old cell itself is probed once per planar direction (d = 0) and holds the moving queen: -4 on the old
                // side, i.e. +4.  Own columns (the k-axis lines): popc(W) minus the moving queen on the old side.  On the
                // new side the moving queen still sits at the old cell; if that lies on a line through the new cell it was
                // counted and is removed.
                cw_new = colw[times_N<NC>(ni, N) + nj], cw_old = colw[times_N<NC>(oi, N) + oj];  // kept: an accepted move rewrites them without reading again
                const int own_new = popc_w(cw_new), own_old = popc_w(cw_old) - 1;
                const int moving = on_a_line_m1(abs_diff_minus_1(oi, ni), abs_diff_minus_1(oj, nj), abs_diff_minus_1(ok_, nk)) ? 1 : 0;
                dE = group_sum<G>(part) + 4 + own_new - own_old - moving;
            }

            STAMP(3);  // dE probes + reduce
            // ---- accept iff u < min(1, exp(-beta dE)); the uniform is always drawn -------------------
            // x = -beta dE < 0 iff beta and dE have the same sign (c32 has the sign of -beta); otherwise
            // the probability is 1 (also for a NaN beta, like min(1.0, nan) in the reference).  The sign comes from the float32
            // product dE * c32: no rounding can change it (|dE| >= 1 or the product is an exact zero), and a c32 that underflowed to
            // zero stands for a |x| far below 2^-54, where exp(x) rounds to 1.0 in float64 as well.
            const float fdE = (float)dE;
            const bool xneg = fdE * c32 < 0.0f;  // exact in sign; false for a zero or NaN c32 (and for inf * 0), like !(x < 0.0) in float64
            // u * 2^27 lies in [a27, a27 + 1), a27 = the top 27 bits of u; e27 = exp(x) * 2^27 within 3e-5 relative.
            // d = a27 + 1/2 - e27 and the half-width w = e27 * 2^-10 + 1/2: d < -w accepts for sure (a27 + 1 < e27 (1 - 2^-10)),
            // d > w rejects for sure (a27 > e27 (1 + 2^-10)), in between the float64 path decides.  Both tests end in one
            // compare whose operand carries the x < 0 condition, so each result is a lane mask straight from the compare.
            const float fa = (float)(uw1 >> 5);
            const float e27 = __builtin_amdgcn_exp2f(fmaf(fdE, c32, 27.0f));
            const float d = (fa + 0.5f) - e27;
            const float w = fmaf(e27, w_scale, w_bias);  // (x < 0 here whenever w is used: e27 <= 2^27)
            bool exact = __builtin_fabsf(d) <= (xneg ? w : -1.0f);
            uint32_t acc = (xneg ? d : -1.0f) < 0.0f ? 1u : 0u;  // 0 / 1 in a vector register: the rare branch below may rewrite it
            if (__builtin_expect(wave_any(exact), 0)) {  // ~0.1 % of the steps of a chain
                STAMP_COUNT(8);
                if (exact) {
                    const int r = accept_exact(EXCH ? beta_tab[step] * a.exch_ladder[rung] : beta_tab[step], dE, uw1, uw2);
                    acc = (uint32_t)r & 1u;
                    if (r >> 1) cold[C_TIES] += 1;
                }
            }

            STAMP(4);  // accept test
            accw = __builtin_amdgcn_alignbit(acc, accw, 1);  // the flag enters at bit 31: after 32 steps the first of them sits in bit 0
            if (MODE == MCQ_MODE_BOARD) {
                hts[cell] = (uint8_t)(acc ? pc : old_k);  // every lane of the group writes the same byte; a rejected move rewrites the old height
                if constexpr (CNT) {
                    if (acc) {  // the queen leaves its twelve old lines and enters the twelve new ones (all 24 distinct: every key contains k)
#pragma unroll
                        for (int f = 0; f < 3; f++) cnt[cix[f]] = (uint8_t)(cvl[f] - 1), cnt[cix[3 + f]] = (uint8_t)(cvl[3 + f] + 1);
                    }
                }
            } else if (acc) {
                // mcmc.py:171-183; every lane of the group performs the same writes.  The two column words were read for the own-column
                // counts of dE: no second LDS round trip here.  A move inside one column changes one word twice: the second
                // store carries both changes (LDS stores of a wavefront land in order).
                const uint32_t op_i = oldp & QM, op_j = (oldp >> QS) & QM;
                const cword_t cleared = cw_old & ~((cword_t)1 << ((oldp >> (2 * QS)) & QM));
                const bool same_column = op_i == (uint32_t)pa && op_j == (uint32_t)pb;
                colw[times_N<NC>((int)op_i, N) + (int)op_j] = (colw_t)cleared;
                colw[times_N<NC>(pa, N) + pb] = (colw_t)((same_column ? cleared : cw_new) | ((cword_t)1 << pc));
#if defined(MCQ_EXP_QN_SC1)
                if constexpr (SLIM) __hip_atomic_store(qn + qi, (uint16_t)newp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else qn[qi] = (qn_t)newp;
#elif defined(MCQ_EXP_QN_NT)
                if constexpr (SLIM) __builtin_nontemporal_store((uint16_t)newp, qn + qi);
                else qn[qi] = (qn_t)newp;
#else
                qn[qi] = (qn_t)newp;
#endif
            }
            E += __mul24((int)acc, dE);  // E += acc ? dE : 0  (|dE| <= 8 N)
            const bool improved = E < best;  // only an accepted move can get below the best so far (E >= best otherwise)
            best = min(best, E);
            const int e = step + 1;
            if (__builtin_expect(wave_any(improved), 0)) {  // rare after the first few hundred steps
                STAMP_COUNT(9);
                if (improved) {
                    // first index of the minimum of energy_history (experiments.py:364-365); with patience 0 the chain stops
                    // right here without appending this entry (no_improvement_steps = 0 >= 0), so the index stays
                    if (!PATIENCE || patience > 0) cold[C_BEST_STEP] = e;
                    if (PATIENCE) deadline = (uint32_t)step + patience;
                    uint8_t* bo = a.out.best_state ? a.out.best_state + chain * (long long)state_bytes : nullptr;
                    if (bo) copy_state_out<MODE, G, PATIENCE || REDUCED>(bo, hts, qn, Q, gl);
                }
            }

            // the common tail of a step: append the entry, flush full blocks, pace
            auto append_entry = [&]() {
                stage[e & (SB - 1)] = (uint32_t)E | (reduced ? 0x40000000u | (acc << 31) : 0u);  // REDUCED: bit 30 valid entry, bit 31 its step was accepted
                if ((e & (SB - 1)) == flush_at) {  // one aligned 64-byte segment per chain (flush_at = SB - 1, or out of reach without a trace)
                    // non-temporal: the trace is written once and never read here, so its lines should not push the MT19937 state's
                    // lines out of the L2 (reads 68.0 -> 65.6 B/move, time unchanged; profiles/r02_nt_trace_experiment.txt)
                    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    if constexpr (WPL == 8) {
                        __builtin_nontemporal_store(*(const u32x4*)(stage + gl * 8), (u32x4*)hist_at(e - (SB - 1)));
                        __builtin_nontemporal_store(*(const u32x4*)(stage + gl * 8 + 4), (u32x4*)hist_at(e - (SB - 1)) + 1);
                    } else if constexpr (WPL == 4) __builtin_nontemporal_store(*(const u32x4*)(stage + gl * 4), (u32x4*)hist_at(e - (SB - 1)));
                    else if constexpr (WPL == 2) __builtin_nontemporal_store(*(const u32x2*)(stage + gl * 2), (u32x2*)hist_at(e - (SB - 1)));
                    else __builtin_nontemporal_store((int)stage[gl], hist_at(e - (SB - 1)));
                }
                if ((e & 31) == 0) {
                    cold[C_N_ACC] += __popc(accw);  // accepted moves are counted from the bit words
                    if (have_bits && gl == 0) __builtin_nontemporal_store(accw, bits_at((e >> 5) - 1));
                    accw = 0;
#ifndef MCQ_PACE_MASK
#define MCQ_PACE_MASK 63
#endif
                    if ((e & MCQ_PACE_MASK) == 0 && a.pace) {
                        // Pacing: publish this wavefront's progress, read the row of its SIMD, and take a priority that grows with
                        // the number of co-resident wavefronts that are further along (ties fall to the arbiter's age order).
                        const uint32_t mine = (uint32_t)e;
                        if (lane == 0) __hip_atomic_store(pace_row + wave_slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint32_t other = lane < 16 ? __hip_atomic_load(pace_row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                        set_priority(min(3, (int)__popcll(__ballot(other > mine))));
                    }
                }
            };
            if constexpr (PATIENCE && !ALL) {
                // Early stop (experiments.py:349-353).  A chain stops in few steps of a run, so the test is one wave-wide ballot and
                // the step's tail runs undivided unless some chain of the wavefront stops right now.  (The ALL form never gets here with
                // a chain that stops: its loop hands a step in which one might over to this form before executing it.)
                const bool stop = (uint32_t)step >= deadline;
                if (__builtin_expect(wave_any(stop), 0)) {
                    if (stop) {
                        // break BEFORE the append: entries 0..step are valid
                        active = false;
                        cold[C_HIST_LEN] = e;
                        if constexpr (REDUCED) {  // this step counts as executed / accepted but appends no entry; the rest of the block is stale
#pragma unroll
                            for (int w = 0; w < WPL; w++)
                                if (gl * WPL + w >= (e & (SB - 1))) stage[gl * WPL + w] = gl * WPL + w == (e & (SB - 1)) && acc ? 0x80000000u : 0u;
                        }
                        if (trace)
                            for (int w = 0; w < WPL; w++)
                                if (gl * WPL + w <= (step & (SB - 1))) hist_at(step & ~(SB - 1))[w] = (int)stage[gl * WPL + w];
                        cold[C_N_ACC] += __popc(accw);
                        if (have_bits && gl == 0) *bits_at(step >> 5) = accw >> (31 - (step & 31));  // (step & 31) + 1 flags so far
                    } else {
                        append_entry();
                    }
                } else {
                    append_entry();
                }
            } else {
                append_entry();
            }
            if constexpr (EXCH) {
                if (--xcount == 0) {  // wave-uniform: every K-th step
                    xcount = (int)(a.exch_every > 2147483647LL ? 2147483647LL : a.exch_every);
                    const int R = a.exch_R;
                    const int cl = grp & (R - 1);                  // this chain's place in its ladder (R is a power of two dividing 64 / G)
                    const int lb = (grp - cl) * G + gl;            // the same lane of the ladder's first chain
                    const bool lower = ((rung ^ xpar) & 1) == 0;   // pairs (t, t + 1), t = xpar (mod 2): the lower rung decides
                    xpar ^= 1;
                    const int prt = lower ? rung + 1 : rung - 1;   // the partner's rung
                    const bool paired = (unsigned)prt < (unsigned)R;
                    // rung -> chain: every chain drops its place at the lane group of its rung (the rungs of a ladder are a
                    // permutation: no two lanes write one destination), then reads who sits on the partner's rung
                    const int on_rung = __builtin_amdgcn_ds_permute((lb + rung * G) * 4, cl);
                    const int pcl = __builtin_amdgcn_ds_bpermute((lb + (paired ? prt : rung) * G) * 4, on_rung);
                    const int pE = __builtin_amdgcn_ds_bpermute((lb + pcl * G) * 4, E);
                    // the lower chain draws random() from its stream, always (like the uniform of a step): two ready words
                    // (a chain that has run dry is topped up on the spot; the others are left alone: a block requested here could
                    // land at the next step's upkeep with no step's words consumed in between, i.e. into a ring that is still full)
                    while (wave_any(rng.gen - rng.pos < 2u)) {
                        if (rng.gen - rng.pos < 2u) {
                            if constexpr (PHILOX) {
                                rng.generate();
                            } else {
                                if (!rng.pending) rng.issue();
                                rng.complete();
                            }
                        }
                    }
                    const uint32_t xw1 = ring[rng.pos & (RING - 1)], xw2 = ring[(rng.pos + 1u) & (RING - 1)];
                    int dec = 0;
                    if (lower && paired) {
                        rng.consume(2u);
                        const double b = beta_tab[step];
                        dec = exchange_decide(b * a.exch_ladder[rung], b * a.exch_ladder[prt], E - pE, xw1, xw2);
                        if (dec >> 1) cold[C_TIES] += 1;
                    }
                    const int pdec = __builtin_amdgcn_ds_bpermute((lb + pcl * G) * 4, dec);
                    if (paired && (((lower ? dec : pdec) & 1) != 0)) {
                        rung = prt;
                        n_exch++;
                        m32 = (float)a.exch_ladder[rung];
                    }
                }
            }
        }
        end_of_step(std::integral_constant<bool, !ALL>());
        if constexpr (PATIENCE && !ALL) {
            if (!wave_any(active)) {  // every lane takes part in this ballot, stopped chains included
                last_entry = step + 1;
                return STEP_FINISHED;
            }
        }
        return STEP_GO_ON;
    };
    if constexpr (PATIENCE || REDUCED) {
        // The ALL form runs while every lane is a live chain and none can stop in the step at hand: a chain stops at the first step
        // >= its deadline, and a step can only move the deadline further away (an improvement), so `step >= deadline` before the step
        // is the condition -- or a false alarm, which costs one step in the general form.  That step and everything after it goes to
        // the general form: the ALL form holds no early-stop code at all, only this compare in front of each step.
        if (!wave_any(!active))  // (the last wavefront of a launch may carry idle groups: general form from the start)
        {
            // (one condition, evaluated without a short circuit, and one exit at the bottom: with two exits the compiler carries shadow
            // copies of the per-lane flags it keeps in scalar registers -- eight scalar instructions per step)
            auto all_go_on = [&]() { return ((vstep < n_steps ? 1 : 0) & (PATIENCE && wave_any((uint32_t)vstep >= deadline) ? 0 : 1)) != 0; };
            if (all_go_on()) {  // written as a guarded do-while: the ballot is a convergent operation, which the compiler will not duplicate to rotate a while loop
                do {
                    (void)metropolis_step(std::true_type());
                } while (all_go_on());
            }
        }
        while (vstep < n_steps && metropolis_step(std::false_type()) == STEP_GO_ON) {}
    } else {  // idle groups have left: every lane is a live chain to the end
        while (vstep < n_steps) (void)metropolis_step(std::true_type());
    }
    STAMP_FLUSH(a.dbg);
    WAVE_T1(a.dbg);
    if (lane == 0 && a.pace) __hip_atomic_store(pace_row + wave_slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // a finished wavefront is ahead of nobody
    if (reduced && (last_entry & 15) != 15) {  // partial last block: words behind the last entry are left from the block before
        if (active) {
#pragma unroll
            for (int w = 0; w < WPL; w++)
                if (gl * WPL + w > (last_entry & 15)) stage[gl * WPL + w] = 0u;
        }
        reduce_block<G>(lds + LDS_STAGE, a.chain_lds_words, lane, last_entry & ~15, a.n_steps + 1, red, a.red_len);
    }

    if (active) {  // ran to n_steps: flush the partial last block and word
        if (trace && (n_steps & (SB - 1)) != SB - 1)
            for (int w = 0; w < WPL; w++)
                if (gl * WPL + w <= (n_steps & (SB - 1))) hist_at(n_steps & ~(SB - 1))[w] = (int)stage[gl * WPL + w];
        if ((n_steps & 31) != 0) cold[C_N_ACC] += __popc(accw);
        if (have_bits && gl == 0 && (n_steps & 31) != 0) *bits_at(n_steps >> 5) = accw >> (32 - (n_steps & 31));  // n_steps & 31 flags in the last word
    }
    if (chain < a.n_chains) {
        if (gl == 0) {
            const int hist_len = cold[C_HIST_LEN];
            if (a.out.hist_len) a.out.hist_len[chain] = hist_len;
            if (a.out.steps_executed) a.out.steps_executed[chain] = hist_len == n_steps + 1 ? n_steps : hist_len;
            if (a.out.best_energy) a.out.best_energy[chain] = best;
            if (a.out.final_energy) a.out.final_energy[chain] = E;
            if (a.out.steps_to_best) a.out.steps_to_best[chain] = cold[C_BEST_STEP];
            if (a.out.n_accepted) a.out.n_accepted[chain] = cold[C_N_ACC];
            if (a.out.near_ties) a.out.near_ties[chain] = cold[C_TIES];
            // mcq_outputs.stream_words: where the chain's stream stands, left in the record's gen_end word (read once, when the stream was attached); the init
            // kernel wrote the words IT took and mcq_stream_words_kernel adds this position minus the one the sweep started from.  (Through the stream's own
            // base + offset registers: a store to a.out.stream_words here would keep three more kernel arguments alive across the loop -- 2 VGPRs and 3-17
            // spilled SGPRs more in every variant.)
            if constexpr (!PHILOX) *rng.word(REC_GEN_END) = rng.pos;  // (a Philox stream is addressed by position, there is nothing to hand back: stream_words is 0)
            if (EXCH && a.out.exchange_rung) a.out.exchange_rung[chain] = rung;
            if (EXCH && a.out.n_exchanges) a.out.n_exchanges[chain] = n_exch;
        }
        if (a.out.final_state) {
            uint8_t* fo = a.out.final_state + chain * (long long)state_bytes;
            copy_state_out<MODE, G, PATIENCE || REDUCED>(fo, hts, qn, Q, gl);
        }
    }
}

// mcq_outputs.stream_words = the init kernel's count (already there) + the words the sweep took: final position (left in the record's gen_end word by
// the sweep's epilogue) minus the position the init kernel handed over, modulo 2^32
__global__ __launch_bounds__(256) void mcq_stream_words_kernel(uint32_t* __restrict__ words, const uint32_t* __restrict__ ws, int rec_words, long long n_chains) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r < n_chains) words[r] += ws[r * rec_words + REC_GEN_END] - ws[r * rec_words + REC_POS];
}

// ------------------------------------------------------------------------------------------------
// trace statistics: what plot_energy_histories / plot_acceptance_rates_binned consume
// (experiments.py:593-595 per-step mean/std over runs, 660-695 binned acceptance), computed from the
// HBM-resident trace so that it never has to cross PCIe.
// ------------------------------------------------------------------------------------------------
// One workgroup per block of 64 history entries: thread (x = entry, y = chain slice) walks the chains,
// reading 256-byte row segments (coalesced along x), and accumulates sum, sum of squares and count.
__global__ __launch_bounds__(256) void mcq_step_stats_kernel(const int32_t* __restrict__ hist, const int64_t* __restrict__ hist_len,
                                                             long long n_chains, long long hist_stride, long long n_entries,
                                                             long long* __restrict__ sum, long long* __restrict__ sumsq,
                                                             long long* __restrict__ count) {
    __shared__ long long red[3][4][64];
    const int x = threadIdx.x & 63, y = threadIdx.x >> 6;
    const long long e = (long long)blockIdx.x * 64 + x;
    long long s = 0, q = 0, c = 0;
    if (e < n_entries)
        for (long long r = y; r < n_chains; r += 4) {
            if (e < hist_len[r]) {  // ragged after an early stop
                const long long v = hist[r * hist_stride + e];
                s += v, q += v * v, c++;
            }
        }
    red[0][y][x] = s, red[1][y][x] = q, red[2][y][x] = c;
    __syncthreads();
    if (y == 0 && e < n_entries) {
        sum[e] = red[0][0][x] + red[0][1][x] + red[0][2][x] + red[0][3][x];
        sumsq[e] = red[1][0][x] + red[1][1][x] + red[1][2][x] + red[1][3][x];
        count[e] = red[2][0][x] + red[2][1][x] + red[2][2][x] + red[2][3][x];
    }
}

// One thread per (chain, bin): accepted and proposed steps of the chain inside [bin_lo[b], bin_lo[b+1]).
__global__ __launch_bounds__(256) void mcq_accept_bins_kernel(const unsigned long long* __restrict__ bits, const int64_t* __restrict__ executed,
                                                              long long n_chains, long long bits_stride, int n_bins,
                                                              const int64_t* __restrict__ bin_lo, unsigned long long* __restrict__ accepted,
                                                              unsigned long long* __restrict__ proposed) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_chains * n_bins) return;
    const long long r = t / n_bins;
    const int b = (int)(t % n_bins);
    const long long ex = executed[r];
    long long lo = bin_lo[b], hi = bin_lo[b + 1];
    if (hi > ex) hi = ex;
    if (lo >= hi) return;
    const unsigned long long* row = bits + r * bits_stride;
    long long acc = 0;
    for (long long w = lo >> 6; w <= (hi - 1) >> 6; w++) {
        unsigned long long m = row[w];
        const long long base = w << 6;
        if (lo > base) m &= ~0ull << (lo - base);
        if (hi < base + 64) m &= ~0ull >> (base + 64 - hi);
        acc += __popcll(m);
    }
    atomicAdd(&accepted[b], (unsigned long long)acc);
    atomicAdd(&proposed[b], (unsigned long long)(hi - lo));
}

// ------------------------------------------------------------------------------------------------
// the packed node summary of a launch (mcq_pack_summary_device): one workgroup per schedule set reduces the set's chains on this rank
// and fills its job's counters, minimum slot and per-chain slots; a second kernel copies the four per-entry arrays of the reduced trace.
// ------------------------------------------------------------------------------------------------
constexpr int PACK_SETS = 32;  // slots per launch of the pack kernels (more sets: several launches)
struct PackArgs {
    mcq_pack_slot slot[PACK_SETS];
    const int32_t* best;
    const int64_t *stb, *acc, *exe, *hist_len;
    const int64_t *step_sum, *step_sumsq, *step_accepted, *step_count;
    long long cps, n_local, n_steps, first_set;
    int64_t* packed;
};

__global__ __launch_bounds__(256) void mcq_pack_kernel(PackArgs a) {
    __shared__ long long red[6][256];
    const mcq_pack_slot sl = a.slot[blockIdx.x];
    const long long c0 = (a.first_set + blockIdx.x) * a.cps;
    long long s_acc = 0, s_exe = 0, s_best = 0, s_sq = 0, s_stb = 0, mn = 0x7fffffffffffffffLL;
    for (long long r = threadIdx.x; r < a.n_local; r += 256) {
        const long long be = a.best[c0 + r], sb = a.stb[c0 + r];
        s_acc += a.acc[c0 + r], s_exe += a.exe[c0 + r], s_best += be, s_sq += be * be, s_stb += sb;
        mn = be < mn ? be : mn;
        if (sl.best >= 0) a.packed[sl.best + r] = be;
        if (sl.stb >= 0) a.packed[sl.stb + r] = sb;
        if (sl.stats >= 0) {  // a chain that stopped early did so at the entry it did not append (experiments.py:349-353)
            const long long hl = a.hist_len[c0 + r];
            if (hl <= a.n_steps) atomicAdd((unsigned long long*)&a.packed[sl.stats + 4 * (a.n_steps + 1) + hl], 1ull);
        }
    }
    red[0][threadIdx.x] = s_acc, red[1][threadIdx.x] = s_exe, red[2][threadIdx.x] = s_best, red[3][threadIdx.x] = s_sq, red[4][threadIdx.x] = s_stb, red[5][threadIdx.x] = mn;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            for (int k = 0; k < 5; k++) red[k][threadIdx.x] += red[k][threadIdx.x + o];
            red[5][threadIdx.x] = red[5][threadIdx.x + o] < red[5][threadIdx.x] ? red[5][threadIdx.x + o] : red[5][threadIdx.x];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int64_t* c = a.packed + sl.counters;
        c[0] = a.n_local;
        if (a.n_local > 0) {
            c[1] = red[0][0], c[2] = red[1][0], c[3] = red[2][0], c[4] = red[3][0], c[5] = red[4][0];
            a.packed[sl.min_slot] = red[5][0] + 1;
        }
    }
}

__global__ __launch_bounds__(256) void mcq_pack_stats_kernel(PackArgs a) {
    const mcq_pack_slot sl = a.slot[blockIdx.y];
    if (sl.stats < 0) return;
    const long long L = a.n_steps + 1, e = (long long)blockIdx.x * 256 + threadIdx.x, src = (a.first_set + blockIdx.y) * L + e;
    if (e >= L) return;
    a.packed[sl.stats + e] = a.step_sum[src];
    a.packed[sl.stats + L + e] = a.step_sumsq[src];
    a.packed[sl.stats + 2 * L + e] = a.step_accepted[src];
    a.packed[sl.stats + 3 * L + e] = a.step_count[src];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local char g_err[512];
#if defined(MCQ_STAMPS) || defined(MCQ_WAVE_TIMES)
unsigned long long* g_dbg = nullptr;
#endif

int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return fail(MCQ_EDEVICE, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

int gcd_int(int a, int b) {
    while (b) {
        int t = a % b;
        a = b, b = t;
    }
    return a;
}

unsigned host_mask(unsigned m) {
    unsigned mask = m;
    mask |= mask >> 1, mask |= mask >> 2, mask |= mask >> 4, mask |= mask >> 8, mask |= mask >> 16;
    return mask;
}

int validate(const mcq_params* p) {
    if (!p) return fail(MCQ_EINVAL, "null params");
    if (p->abi_version != MCQ_ABI_VERSION) return fail(MCQ_EINVAL, "abi_version mismatch");
    if (p->mode != MCQ_MODE_BOARD && p->mode != MCQ_MODE_FULL3D) return fail(MCQ_EINVAL, "unknown mcmc_type");
    if (p->N < MCQ_MIN_N || p->N > (p->mode == MCQ_MODE_BOARD ? MCQ_MAX_N_BOARD : MCQ_MAX_N))
        return fail(MCQ_EINVAL, "N out of range [2, 64] (full_3d) / [2, 128] (board)");
    if (p->mode == MCQ_MODE_FULL3D && p->N > 32) {  // the variant with 64-bit column words (mcq_sweep_kernel: WIDE)
        if (p->lanes_per_chain != 0 && p->lanes_per_chain != 16) return fail(MCQ_EINVAL, "full_3d beyond N = 32 runs at 16 lanes per chain (lanes_per_chain 0 or 16)");
        if (p->rng != MCQ_RNG_MT19937_NUMPY) return fail(MCQ_EINVAL, "full_3d beyond N = 32 runs with NumPy's MT19937 stream only");
        if (p->exchange_every > 0) return fail(MCQ_EINVAL, "full_3d beyond N = 32 runs without replica exchange");
    }
    if (p->init < MCQ_INIT_RANDOM || p->init > MCQ_INIT_KLARNER) return fail(MCQ_EINVAL, "Unknown init_mode");
    if (p->sched < MCQ_SCHED_CONSTANT || p->sched > MCQ_SCHED_SINUSOIDAL) return fail(MCQ_EINVAL, "Unknown betta_scheduling type");
    if (p->rng != MCQ_RNG_MT19937_NUMPY && p->rng != MCQ_RNG_PHILOX4X32_10) return fail(MCQ_EINVAL, "unknown rng");
    if (p->trace != MCQ_TRACE_NONE && p->trace != MCQ_TRACE_I32 && p->trace != MCQ_TRACE_REDUCED) return fail(MCQ_EINVAL, "unknown trace mode");
    if (p->n_steps < 0 || p->n_steps > 2147483000LL) return fail(MCQ_EINVAL, "n_steps out of range [0, 2^31)");
    if (p->n_chains < 0) return fail(MCQ_EINVAL, "negative n_chains");
    if (p->n_chains > 2147483647LL) return fail(MCQ_EINVAL, "n_chains out of range [0, 2^31)");  // one workgroup per chain in the init kernel
    if (p->lanes_per_chain != 0 && p->lanes_per_chain != 2 && p->lanes_per_chain != 4 && p->lanes_per_chain != 8 && p->lanes_per_chain != 16)
        return fail(MCQ_EINVAL, "lanes_per_chain must be 0, 2, 4, 8 or 16");
    if (p->lanes_per_chain == 2) {  // 32 chains per wavefront
        if (p->mode != MCQ_MODE_BOARD) return fail(MCQ_EINVAL, "lanes_per_chain 2 applies to mcmc_type board (full_3d splits a chain's lanes between the old and the new cell)");
        if (p->n_sets > 1 && p->chains_per_set % 32 != 0) return fail(MCQ_EINVAL, "lanes_per_chain 2 needs chains_per_set to be a multiple of 32 (a wavefront belongs to one set)");
        if (p->trace == MCQ_TRACE_I32 && p->bits_stride >= (1LL << 23)) return fail(MCQ_EINVAL, "lanes_per_chain 2 needs bits_stride < 2^23");
    }
    if (p->n_sets < 0) return fail(MCQ_EINVAL, "negative n_sets");
    if (p->n_sets > 1) {
        if (!p->sets) return fail(MCQ_EINVAL, "n_sets > 1 without sets");
        if (p->chains_per_set <= 0 || p->chains_per_set % 16 != 0) return fail(MCQ_EINVAL, "chains_per_set must be a positive multiple of 16");
        if (p->n_chains != p->n_sets * p->chains_per_set) return fail(MCQ_EINVAL, "n_chains must equal n_sets * chains_per_set");
        for (int64_t t = 0; t < p->n_sets; t++) {
            if (p->sets[t].sched < MCQ_SCHED_CONSTANT || p->sets[t].sched > MCQ_SCHED_SINUSOIDAL) return fail(MCQ_EINVAL, "unknown schedule type in sets");
            if (p->sets[t].init_plus1 < 0 || p->sets[t].init_plus1 > MCQ_INIT_KLARNER + 1) return fail(MCQ_EINVAL, "Unknown init_mode in sets");
        }
    }
    if (p->n_queens < 0) return fail(MCQ_EINVAL, "negative n_queens");
    if (p->n_queens > 0 && p->n_queens != p->N * p->N) {  // mcmc.py:6-18, 92-101: any Q <= N^3 with the random init; latin / klarner assume Q = N^2 (mcmc.py:21-25)
        if (p->mode != MCQ_MODE_FULL3D) return fail(MCQ_EINVAL, "n_queens applies to mcmc_type full_3d (a board has one queen per column)");
        bool other_init = p->init != MCQ_INIT_RANDOM;
        for (int64_t t = 0; t < p->n_sets && p->n_sets > 1; t++) other_init |= p->sets[t].init_plus1 != 0 && p->sets[t].init_plus1 != MCQ_INIT_RANDOM + 1;
        if (other_init) return fail(MCQ_EINVAL, "latin / klarner initialization assumes Q = N^2");
        if (p->n_queens < 2) return fail(MCQ_EINVAL, "n_queens must be at least 2 in this build");
        if ((int64_t)p->n_queens >= (int64_t)p->N * p->N * p->N) return fail(MCQ_EINVAL, "n_queens must leave a free cell: Q < N^3 (the reference raises for Q > N^3 and never returns for Q = N^3)");
        if (p->n_queens > 32767) return fail(MCQ_EINVAL, "n_queens above 32767 is not supported by this build");
    }
    if (p->stream_states) {
        if (p->rng != MCQ_RNG_MT19937_NUMPY) return fail(MCQ_EINVAL, "stream_states continues an MT19937 stream (a Philox stream is a function of its seed)");
        for (int64_t r = 0; r < p->n_chains; r++)
            if (p->stream_states[r * 625 + 624] > 624u) return fail(MCQ_EINVAL, "stream_states: MT19937 position out of range");
    }
    if (p->exchange_every < 0) return fail(MCQ_EINVAL, "negative exchange_every");
    if (p->exchange_every > 0) {
        const int R = p->exchange_replicas;
        if (R != 2 && R != 4 && R != 8 && R != 16) return fail(MCQ_EINVAL, "exchange_replicas must be 2, 4, 8 or 16");
        if (!p->exchange_ladder) return fail(MCQ_EINVAL, "exchange_every > 0 without exchange_ladder");
        for (int t = 0; t < R; t++)  // a rung runs at beta(step) * ladder[t]: a multiplier that is not a positive finite number has no meaning
            if (!(p->exchange_ladder[t] > 0.0) || p->exchange_ladder[t] > 1.7976931348623157e308) return fail(MCQ_EINVAL, "exchange_ladder entries must be finite and positive");
        if (p->n_chains % R != 0 || (p->n_sets > 1 && p->chains_per_set % R != 0)) return fail(MCQ_EINVAL, "n_chains (and chains_per_set) must be multiples of exchange_replicas");
        if (p->mode == MCQ_MODE_BOARD && p->patience >= 0 && p->patience <= p->n_steps) return fail(MCQ_EINVAL, "replica exchange needs early stopping off (early_stop_patience None)");
        if (p->trace == MCQ_TRACE_REDUCED) return fail(MCQ_EINVAL, "replica exchange runs with trace none or i32");
        if (p->lanes_per_chain != 0 && 64 / p->lanes_per_chain < R) return fail(MCQ_EINVAL, "lanes_per_chain too wide: a ladder of exchange_replicas chains must fit one wavefront");
    }
    return MCQ_OK;
}

// queens of a full_3d chain: N * N unless the caller names a count (mcmc.py:6-18 State3DQueens(N, Q=...), random init only)
int queens_of(const mcq_params* p) { return p->mode == MCQ_MODE_FULL3D && p->n_queens > 0 ? p->n_queens : p->N * p->N; }
size_t state_bytes_of(const mcq_params* p) { return p->mode == MCQ_MODE_BOARD ? (size_t)p->N * p->N : (size_t)3 * queens_of(p); }

// 64-byte multiple: the sweep reads and writes the MT words of a record in aligned 64-byte blocks
#if defined(MCQ_EXP_LAST_TOUCH) || defined(MCQ_EXP_REC128)  // timing experiment: 128-byte records, so that a block's half of its line is known from its index
int rec_words_for(const mcq_params* p) { return (REC_STATE + (int)((state_bytes_of(p) + 3) / 4) + 31) & ~31; }
#else
int rec_words_for(const mcq_params* p) { return (REC_STATE + (int)((state_bytes_of(p) + 3) / 4) + 15) & ~15; }
#endif

// full_3d: entries per chain of the packed queen table in the workspace (64-byte rows; uint16 entries, uint32 beyond N = 32); 0 for boards
int qtab_stride_for(const mcq_params* p) { return p->mode == MCQ_MODE_FULL3D ? (queens_of(p) + 31) & ~31 : 0; }
size_t qtab_bytes_for(const mcq_params* p) { return (size_t)(p->n_chains > 0 ? p->n_chains : 1) * (size_t)qtab_stride_for(p) * (p->N > 32 ? 4 : 2); }
// full_3d beyond N = 32 with a random init: the init kernel's permutation arrays (N^3 uint32 per chain) for as many chains at a time as 1 GiB holds
// (N = 33: 7 468, N = 64: 1 024) -- one chain per wavefront, so a round should bring a wavefront for every SIMD -- and never fewer than 256
constexpr long long PERM_SLOTS_MIN = 256;
constexpr long long PERM_BUDGET_BYTES = 1LL << 30;
bool any_random_init(const mcq_params* p) {
    bool any = p->init == MCQ_INIT_RANDOM;
    for (int64_t t = 0; t < p->n_sets && p->n_sets > 1; t++) any |= p->sets[t].init_plus1 == MCQ_INIT_RANDOM + 1;
    return any;
}
long long perm_slots_for(const mcq_params* p) {
    if (p->mode != MCQ_MODE_FULL3D || p->N <= 32 || !any_random_init(p)) return 0;
    const long long per_chain = (long long)p->N * p->N * p->N * 4, fit = (PERM_BUDGET_BYTES / per_chain) & ~3LL, cap = fit > PERM_SLOTS_MIN ? fit : PERM_SLOTS_MIN;
    return p->n_chains < cap ? ((p->n_chains > 0 ? p->n_chains : 1) + 3) & ~3LL : cap;  // (a multiple of the init kernel's chains per wavefront)
}
size_t perm_bytes_for(const mcq_params* p) { return (size_t)perm_slots_for(p) * (size_t)p->N * p->N * p->N * 4; }
// mcq_params.stream_states in the kernels' layout: 626 words per chain (stream_layout), behind everything else in the workspace
size_t stream_bytes_for(const mcq_params* p) { return p->stream_states ? (size_t)(p->n_chains > 0 ? p->n_chains : 1) * 626 * 4 : 0; }

// An MT19937 state as NumPy holds it -- 624 key words of the CURRENT generation, `pos` of them consumed -- in the layout the kernels stream from:
// words [0, ge) of the current generation, words [ge, 624) of the one BEFORE (the kernels twist a generation block by block, in place, as its words
// are needed), ge = pos rounded up to a multiple of 64 (at most 63 ready words: what the sweep's ring takes over), out[624] = pos, out[625] = ge.
// Rewinding word j of a generation: the twist made new[i] = x_i ^ (y_i >> 1) ^ (y_i odd ? A : 0) with y_i = (old[i] & UPPER) | (old[i+1] & LOWER) and
// x_i = old[i + 397] (i < 227) or new[i - 227]; A has its top bit set and y_i >> 1 has not, so new[i] ^ x_i gives y_i back, and old[j] = (y_j & UPPER) |
// (y_{j-1} & LOWER).  (old[0]'s low bits went into nothing -- y_623 took new[0]'s -- and nothing reads them: the key's own are kept.)
void stream_layout(const uint32_t* st, uint32_t* out) {
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
    const int pos = (int)st[624];
    memcpy(out, st, 624 * sizeof(uint32_t));
    if (pos >= 624) {  // nothing of the key's generation is left: it is the "one before" of the generation the next draw starts
        out[624] = 0, out[625] = 0;
        return;
    }
    const int ge = (pos + 63) & ~63;
    out[624] = (uint32_t)pos, out[625] = (uint32_t)(ge > 624 ? 624 : ge);
    if (ge >= 624) return;
    uint32_t y_next = 0;  // y_{i+1} while walking down
    for (int i = 623; i >= (ge > 0 ? ge - 1 : 0); i--) {
        const uint32_t x = i < 227 ? out[i + 397] : st[i - 227];  // (i < 227: old[i + 397], rewound earlier in this walk since i + 397 > i >= ge - 1)
        uint32_t t = st[i] ^ x;
        const uint32_t odd = t >> 31;
        if (odd) t ^= A;
        const uint32_t y = (t << 1) | odd;
        if (i + 1 <= 623 && i + 1 >= ge) out[i + 1] = (y_next & UP) | (y & LO);  // old[i + 1] = upper bit from y_{i+1}, the rest from y_i
        y_next = y;
    }
    if (ge == 0) out[0] = (y_next & UP) | (st[0] & LO);  // old[0]: only its upper bit is ever read
}

size_t n_sets_of(const mcq_params* p) { return p->n_sets > 1 ? (size_t)p->n_sets : 1; }
// one table per schedule set, tab_stride elements apart (a 256-byte multiple for either element size)
size_t tab_stride_for(const mcq_params* p) { return ((size_t)(p->n_steps > 0 ? p->n_steps : 1) + 63) & ~(size_t)63; }
size_t beta_tab_bytes(const mcq_params* p) { return n_sets_of(p) * tab_stride_for(p) * 8; }
size_t c32_tab_bytes(const mcq_params* p) { return n_sets_of(p) * tab_stride_for(p) * 4; }
long long red_len_for(const mcq_params* p) { return (p->n_steps + 1 + 31) & ~31LL; }
size_t red_set_bytes(const mcq_params* p) { return p->trace == MCQ_TRACE_REDUCED ? (size_t)RED_STRIPES * 3 * red_len_for(p) * 8 : 0; }
size_t red_bytes(const mcq_params* p) { return n_sets_of(p) * red_set_bytes(p); }
constexpr size_t PACE_BYTES = 2048 * 16 * 4;  // 8 XCC x 4 SE x 16 CU x 4 SIMD rows of 16 wave slots
constexpr size_t LADDER_BYTES = 16 * 8;       // replica exchange: the beta multipliers of a ladder

// LDS words per chain.  board: the diagonal probes read up to N-1 bytes before / after the heights, (N+2)/4 spare words on each
// side keep those (discarded) reads inside the chain's own slice; full_3d: pad | column words | pad | queens (uint16).
int chain_lds_words_for(int N, int mode, bool narrow, int Q = 0, bool slim = false, bool counters = false, bool wide = false) {
    const int NN = N * N, pad = (N + 3) & ~3;
    if (Q <= 0) Q = NN;  // full_3d: the queens (mcq_params.n_queens); N * N by default
    int w = LDS_STATE;
    if (mode == MCQ_MODE_BOARD) w += (NN + 3) / 4 + (N + 2) / 4 + (counters ? (2 * NN + 6 * N * (2 * N - 1) + 4 * (2 * N - 1) * (2 * N - 1) + 3) / 4 : 0);
    else if (wide) w += 2 * (2 * pad + NN);  // WIDE: pad | 64-bit column words | pad; the queens are in global memory
    else if (slim) w = 16 + 4 + RING + (NN + 1) / 2;  // SLIM: stage[16] | cold[4] | ring[64] | 16-bit column words, no pads; the queens are in global memory
    else w += (narrow ? (2 * pad + NN + 1) / 2 : 2 * pad + NN) + (Q + 1) / 2;
    w = (w + 3) & ~3;  // 16-byte multiple: the staging block and the ring are accessed with 128-bit LDS operations
    // The chains of a wavefront make many accesses at the SAME offset of their slices (history staging, cold scalars, ring appends):
    // a stride of 4 mod 8 words puts the 8 chains of a 32-lane access group on 8 different banks; 0 mod 8 would serialise them
    // (A/B on the headline problem, same box: profiles/r02_lds_stride_ab.txt, r02_lds_stride_pmc.txt).
#ifdef MCQ_EXP_LDS_STRIDE_0MOD8  // timing experiment (tools/exp_build.sh): the conflicting stride, for the A/B in profiles/
    w = (w + 7) & ~7;
#else
    if (w % 8 == 0) w += 4;
#endif
    return w;
}

int build_args(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, void* ws, KArgs* a) {
    memset(a, 0, sizeof *a);
    a->N = p->N, a->Q = queens_of(p), a->NN = p->N * p->N, a->mode = p->mode, a->init = p->init, a->sched = p->sched, a->flags = p->flags, a->rng = p->rng;
    a->maskN = host_mask((unsigned)(p->N - 1)), a->maskQ = host_mask((unsigned)(a->Q - 1));
    a->klarner_M = 0;
    if (p->init == MCQ_INIT_KLARNER && gcd_int(p->N, 210) != 1) {
        for (int m = p->N - 1; m > 0; m--)
            if (gcd_int(m, 210) == 1) {
                a->klarner_M = m;
                break;
            }
        if (a->klarner_M == 0) return fail(MCQ_EINVAL, "no Klarner core below N");
    }
    a->state_bytes = (int)state_bytes_of(p);
    a->rec_words = rec_words_for(p);
    // board: the diagonal probes read up to N-1 bytes before / after the heights; (N+2)/4 spare words on each
    // side keep those (discarded) reads inside the chain's own LDS slice.
    a->full_pad = (p->N + 3) & ~3;
    a->chain_lds_words = chain_lds_words_for(p->N, p->mode, false, a->Q);  // full_3d: the launcher picks the 16-bit layout where it applies
    a->beta_const = p->beta_const, a->beta_start = p->beta_start, a->beta_end = p->beta_end;
    a->n_steps = p->n_steps, a->n_chains = p->n_chains;
    a->patience = p->mode == MCQ_MODE_BOARD ? p->patience : -1;  // full_3d ignores early_stop_patience (experiments.py:199-279)
    a->hist_stride = p->hist_stride, a->bits_stride = p->bits_stride;
    a->beta_tab = (double*)ws;
    a->c32_tab = (float*)((char*)ws + beta_tab_bytes(p));
    a->red = p->trace == MCQ_TRACE_REDUCED ? (unsigned long long*)((char*)ws + beta_tab_bytes(p) + c32_tab_bytes(p)) : nullptr;
    a->red_len = red_len_for(p);
    a->chains_per_set = p->n_sets > 1 ? p->chains_per_set : 0;
    a->tab_stride = (long long)tab_stride_for(p);
    a->red_set_stride = (long long)(red_set_bytes(p) / 8);
    a->pace = (uint32_t*)((char*)ws + beta_tab_bytes(p) + c32_tab_bytes(p) + red_bytes(p));
    a->low_water = p->mode == MCQ_MODE_BOARD ? ((double)p->N / (double)(a->maskN + 1u) < 0.6 ? 28 : 24) : 28;
    a->exch_every = p->exchange_every, a->exch_R = p->exchange_every > 0 ? p->exchange_replicas : 1;
    a->exch_ladder = (const double*)((char*)ws + beta_tab_bytes(p) + c32_tab_bytes(p) + red_bytes(p) + PACE_BYTES);
    a->ws = (uint32_t*)((char*)ws + beta_tab_bytes(p) + c32_tab_bytes(p) + red_bytes(p) + PACE_BYTES + LADDER_BYTES);
    a->qtab_stride = qtab_stride_for(p);
    a->qtab = a->qtab_stride ? (uint16_t*)(a->ws + (size_t)(p->n_chains > 0 ? p->n_chains : 1) * a->rec_words) : nullptr;  // behind the chain records
    a->perm = perm_slots_for(p) ? (uint32_t*)((char*)a->qtab + qtab_bytes_for(p)) : nullptr;                                   // ... and behind that (qtab rows are 64-byte multiples)
    a->stream = p->stream_states ? (const uint32_t*)((char*)(a->ws + (size_t)(p->n_chains > 0 ? p->n_chains : 1) * a->rec_words) + qtab_bytes_for(p) + perm_bytes_for(p)) : nullptr;
    a->seeds = seeds, a->out = *out;
#if defined(MCQ_STAMPS) || defined(MCQ_WAVE_TIMES)
    a->dbg = g_dbg;
#endif
    if (p->trace != MCQ_TRACE_I32) a->out.energy_hist = nullptr, a->out.accept_bits = nullptr;
    return MCQ_OK;
}

// SIMDs of the current device (4 per compute unit)
int device_simds() {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 1024;
    return 4 * cus;
}

// lanes of a wavefront per chain a launch runs with: the caller's choice, or the default for the board size -- except that a board
// launch on its own that leaves SIMDs empty is spread over twice, then four times as many wavefronts while every wavefront
// still gets a SIMD to itself (a lone wavefront is bound by its own latency, and a step is shorter with more lanes:
// profiles/r03_lane_table.txt).  Boards from N = 20 stop at 8 lanes (16 take the run-time probe loop and are slower); boards up
// to N = 8 stay at 4 (one probe pass or two: more lanes shorten nothing, and the 4-lane kernels are the specialised ones).
// (A caller that runs several launches side by side knows better and says so: jobs.plan_lanes.)
int effective_lanes(const mcq_params* p) {
    int G = p->lanes_per_chain;
    if (!G) {
        G = mcq_default_lanes_n(p->mode, p->N);
        if (p->mode == MCQ_MODE_BOARD) {
            const long long room = device_simds();
            const int top = p->N <= 8 ? 4 : p->N >= 20 ? 8 : 16;
            while (G < top && (p->n_chains * (2 * G) + 63) / 64 <= room) G *= 2;
        }
        // full_3d N = 9..12, NumPy's stream, no exchange: the slim 4-lane kernels (16 chains per wavefront; every trace mode)
        if (p->mode == MCQ_MODE_FULL3D && p->N > 8 && p->N <= 12 && p->rng == MCQ_RNG_MT19937_NUMPY && p->exchange_every == 0) G = 4;
        // replica exchange: a ladder lives in one wavefront (its chains swap through cross-lane moves, no barrier)
        if (p->exchange_every > 0 && 64 / G < p->exchange_replicas) G = 64 / p->exchange_replicas;
        if (p->mode == MCQ_MODE_FULL3D && p->N > 32) G = 16;  // 64-bit column words: four chains per wavefront (N = 64: 33 KB each)
    }
    return G;
}

template <int MODE, int G, bool PATIENCE, int NT, bool REDUCED, bool PHILOX = false, int NC = 0, bool EXCH = false, bool CAND5 = false, bool EARLYU = false, bool SLIM = false, bool CNT = false,
          bool WIDE = false>
int launch_sweep(const KArgs& a0, hipStream_t s) {
    constexpr int CPB = 64 / G;  // one wavefront per workgroup: chains never interact, so no barrier exists
    KArgs a = a0;
    // Pacing pays when a SIMD holds several wavefronts of this launch (they can only see each other); a launch that puts less
    // than two wavefronts on a SIMD would pay for the checkpoints and gain nothing.
    const long long waves = (a.n_chains + CPB - 1) / CPB;
    if (waves < 2LL * device_simds() && !(a.flags & MCQ_FLAG_SHARED_PACING)) a.pace = nullptr;
    const size_t lds = (size_t)CPB * a.chain_lds_words * 4 + (SLIM ? 64 : 0);  // (SLIM: the pad behind the last chain's column table)
    if (lds > 160 * 1024)
        return fail(MCQ_EINVAL, MODE == MCQ_MODE_FULL3D && a.Q != a.NN ? "chain state does not fit in LDS (n_queens: the queen table of %s chains per wavefront exceeds 160 KB; more lanes_per_chain halve it)"
                                                                        : "chain state does not fit in LDS (%s chains per wavefront)", G == 2 ? "32" : G == 4 ? "16" : G == 8 ? "8" : "4");
    if (a.dry) return MCQ_OK;
    HIP_TRY(hipFuncSetAttribute((const void*)mcq_sweep_kernel<MODE, G, PATIENCE, NT, REDUCED, PHILOX, NC, EXCH, CAND5, EARLYU, SLIM, CNT, WIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((a.n_chains + CPB - 1) / CPB);
    hipLaunchKernelGGL((mcq_sweep_kernel<MODE, G, PATIENCE, NT, REDUCED, PHILOX, NC, EXCH, CAND5, EARLYU, SLIM, CNT, WIDE>), dim3(grid), dim3(64), lds, s, a);
    HIP_TRY(hipGetLastError());
    return MCQ_OK;
}

// Philox mode (evidence / fast mode, never the reference's stream): the variants of BASELINE's two single_N shapes are
// specialised (board N = 9..12 at 4 lanes, full_3d N = 9..12 at 8 lanes, no early stop, full or no trace); everything else takes
// the run-time probe loop.
template <int MODE, int G>
int launch_sweep_philox(const KArgs& a, hipStream_t s) {
    if constexpr (MODE == MCQ_MODE_FULL3D) {
        if constexpr (G == 8) {
            if (!a.red && (a.N + 3) / 4 == 3) {
                KArgs b = a;
                b.chain_lds_words = chain_lds_words_for(a.N, MCQ_MODE_FULL3D, true, a.Q);
                return a.N == 12 && a.Q == 144 ? launch_sweep<MODE, G, false, 3, false, true, 12>(b, s) : launch_sweep<MODE, G, false, 3, false, true>(b, s);
            }
        }
        return a.red ? launch_sweep<MODE, G, false, 0, true, true>(a, s) : launch_sweep<MODE, G, false, 0, false, true>(a, s);
    } else {
        const bool pat = a.patience >= 0 && a.patience <= a.n_steps;  // a patience beyond n_steps can never stop a chain: the plain variants give the same results
        if constexpr (G == 4) {
            if (!pat && !a.red && a.N == 12) return launch_sweep<MODE, G, false, 3, false, true, 12>(a, s);
            if (!pat && !a.red && (a.N + 3) / 4 == 3) return launch_sweep<MODE, G, false, 3, false, true>(a, s);
        }
        if (a.red) return pat ? launch_sweep<MODE, G, true, 0, true, true>(a, s) : launch_sweep<MODE, G, false, 0, true, true>(a, s);
        return pat ? launch_sweep<MODE, G, true, 0, false, true>(a, s) : launch_sweep<MODE, G, false, 0, false, true>(a, s);
    }
}

// Replica exchange (never a default, not a mode of the reference): the run-time probe loop for every size, plus the two
// single_N shapes of BASELINE (board N = 12 at 4 lanes, full_3d N = 12 at 8) with their unrolled steps.
template <int MODE, int G>
int launch_sweep_exchange(const KArgs& a, hipStream_t s) {
    if (a.rng == MCQ_RNG_PHILOX4X32_10) return launch_sweep<MODE, G, false, 0, false, true, 0, true>(a, s);
    if constexpr (MODE == MCQ_MODE_BOARD && G == 4) {
        if (a.N == 12) return launch_sweep<MODE, G, false, 3, false, false, 12, true>(a, s);
    }
    if constexpr (MODE == MCQ_MODE_FULL3D && G == 8) {
        if (a.N == 12 && a.Q == 144) {
            KArgs b = a;
            b.chain_lds_words = chain_lds_words_for(a.N, MCQ_MODE_FULL3D, true, a.Q);
            return launch_sweep<MODE, G, false, 3, false, false, 12, true>(b, s);
        }
    }
    return launch_sweep<MODE, G, false, 0, false, false, 0, true>(a, s);
}

template <int MODE, int G>
int launch_sweep_g(const KArgs& a, hipStream_t s) {
    if (a.exch_every > 0) return launch_sweep_exchange<MODE, G>(a, s);
    if (a.rng == MCQ_RNG_PHILOX4X32_10) return launch_sweep_philox<MODE, G>(a, s);
    if constexpr (MODE == MCQ_MODE_FULL3D) {  // no early stop (experiments.py:199-279)
        if (a.N > 32) {  // 64-bit column words (validate(): 16 lanes per chain, NumPy's stream, no exchange)
            if constexpr (G == 16) {
                KArgs b = a;
                b.chain_lds_words = chain_lds_words_for(a.N, MCQ_MODE_FULL3D, false, a.Q, false, false, true);
                return a.red ? launch_sweep<MODE, G, false, 0, true, false, 0, false, false, false, false, false, true>(b, s)
                             : launch_sweep<MODE, G, false, 0, false, false, 0, false, false, false, false, false, true>(b, s);
            } else {
                return fail(MCQ_EINVAL, "full_3d beyond N = 32 runs at 16 lanes per chain");
            }
        }
        if constexpr (G == 8) {  // N <= 16: 16-bit column words, four lanes around each of the two cells
            const int nt = (a.N + 3) / 4;
            if (a.red && nt == 3) {  // BASELINE config 3's shape with the reduced trace
                KArgs b = a;
                b.chain_lds_words = chain_lds_words_for(a.N, MCQ_MODE_FULL3D, true, a.Q);
                return a.N == 12 && a.Q == 144 ? launch_sweep<MODE, G, false, 3, true, false, 12>(b, s) : launch_sweep<MODE, G, false, 3, true>(b, s);
            }
            if (!a.red && nt <= 4) {
                KArgs b = a;
                b.chain_lds_words = chain_lds_words_for(a.N, MCQ_MODE_FULL3D, true, a.Q);
                switch (nt) {
                case 1: return launch_sweep<MODE, G, false, 1, false>(b, s);
                case 2: return launch_sweep<MODE, G, false, 2, false>(b, s);
                case 3: return a.N == 12 && a.Q == 144 ? launch_sweep<MODE, G, false, 3, false, false, 12>(b, s) : launch_sweep<MODE, G, false, 3, false>(b, s);
                default: return launch_sweep<MODE, G, false, 4, false>(b, s);
                }
            }
        }
        if constexpr (G == 4) {
            // the slim layout (N = 9..12): two lanes around each of the two cells, ceil(N / 2) unrolled passes, the queens in global memory.
            // (65 536 chains x 20 000 steps against 8 lanes: N = 9 53.2 / 65.4 ms, N = 10 49.6 / 62.6, N = 11 60.3 / 62.9, N = 12 52.1 / 59.8;
            // seven and eight passes spill and lose -- N = 13 129 / 68 ms, N = 16 109 / 91: profiles/r04_full3d_slim.txt)
#ifdef MCQ_EXP_SLIM16  // experiment: seven and eight passes (N = 13..16) at the register budget of three wavefronts per SIMD, which is all their LDS allows anyway
            if (!a.red && a.N > 12 && a.N <= 16) {
                KArgs b = a;
                b.chain_lds_words = chain_lds_words_for(a.N, MCQ_MODE_FULL3D, true, a.Q, true);
                return (a.N + 1) / 2 == 7 ? launch_sweep<MODE, G, false, 7, false, false, 0, false, false, false, true>(b, s) : launch_sweep<MODE, G, false, 8, false, false, 0, false, false, false, true>(b, s);
            }
#endif
            if (a.N > 8 && a.N <= 12) {
                KArgs b = a;
                b.chain_lds_words = chain_lds_words_for(a.N, MCQ_MODE_FULL3D, true, a.Q, true);
                if (a.red) {  // the reduced trace (what the drivers' statistics runs take)
                    if ((a.N + 1) / 2 == 5) return launch_sweep<MODE, G, false, 5, true, false, 0, false, false, false, true>(b, s);
                    return a.N == 12 && a.Q == 144 ? launch_sweep<MODE, G, false, 6, true, false, 12, false, false, false, true>(b, s)
                                                   : launch_sweep<MODE, G, false, 6, true, false, 0, false, false, false, true>(b, s);
                }
                if ((a.N + 1) / 2 == 5) return launch_sweep<MODE, G, false, 5, false, false, 0, false, false, false, true>(b, s);
                return a.N == 12 && a.Q == 144 ? launch_sweep<MODE, G, false, 6, false, false, 12, false, false, false, true>(b, s)
                                               : launch_sweep<MODE, G, false, 6, false, false, 0, false, false, false, true>(b, s);
            }
        }
        return a.red ? launch_sweep<MODE, G, false, 0, true>(a, s) : launch_sweep<MODE, G, false, 0, false>(a, s);
    } else {
        const bool pat = a.patience >= 0 && a.patience <= a.n_steps;  // a patience beyond n_steps can never stop a chain: the plain variants give the same results
        if constexpr (G == 2) {  // 32 chains per wavefront, ceil(N / 2) packed probe passes: the size of BASELINE config 2
            if (a.N == 12) {
                if (!pat) return a.red ? launch_sweep<MODE, G, false, 6, true, false, 12>(a, s) : launch_sweep<MODE, G, false, 6, false, false, 12>(a, s);
                if (!a.red) return launch_sweep<MODE, G, true, 6, false, false, 12>(a, s);
            }
            // the other sizes up to N = 12 without early stop and with a full trace or none: what a job list takes for its SHORT launches when it
            // balances the lanes of launches that run side by side (half the wavefronts of 4 lanes, a longer step: jobs.plan_lanes)
            if (!pat && !a.red && a.N <= 12) {
                if (a.N <= 4) return launch_sweep<MODE, G, false, 2, false, false, 0, false, true>(a, s);  // (five candidates for new_k, like the 4-lane kernels of N <= 5)
                if (a.N == 5) return launch_sweep<MODE, G, false, 3, false, false, 0, false, true>(a, s);
                switch ((a.N + 1) / 2) {
                case 3: return launch_sweep<MODE, G, false, 3, false>(a, s);
                case 4: return launch_sweep<MODE, G, false, 4, false>(a, s);
                case 5: return launch_sweep<MODE, G, false, 5, false>(a, s);
                default: return launch_sweep<MODE, G, false, 6, false>(a, s);
                }
            }
        }
        if constexpr (G == 4) {
            // dE from line counters (MCQ_FLAG_LINE_COUNTERS; N <= 8, NumPy's stream, plain or early-stop, full or no trace): see CNT at the kernel
            if ((a.flags & MCQ_FLAG_LINE_COUNTERS) && a.N <= 8 && !a.red) {
                KArgs b = a;
                b.chain_lds_words = chain_lds_words_for(a.N, MCQ_MODE_BOARD, false, 0, false, true);
                if (a.N <= 5) return pat ? launch_sweep<MODE, G, true, 0, false, false, 0, false, true, false, false, true>(b, s) : launch_sweep<MODE, G, false, 0, false, false, 0, false, true, false, false, true>(b, s);
                return pat ? launch_sweep<MODE, G, true, 0, false, false, 0, false, false, false, false, true>(b, s) : launch_sweep<MODE, G, false, 0, false, false, 0, false, false, false, false, true>(b, s);
            }
        }
        if constexpr (G == 4) {  // straight-line probe blocks for the common board sizes
            // (early stopping -- the reference's default early_stop_patience = 100000 -- has its own unrolled variants for the
            // sizes that default to 4 lanes)
            if (pat && !a.red && a.N <= 5)  // (five candidates for new_k, like the plain variants below)
                return a.N <= 4 ? launch_sweep<MODE, G, true, 1, false, false, 0, false, true>(a, s) : launch_sweep<MODE, G, true, 2, false, false, 0, false, true>(a, s);
            if (pat && !a.red) switch ((a.N + G - 1) / G) {
                case 1: return launch_sweep<MODE, G, true, 1, false>(a, s);
                case 2: return launch_sweep<MODE, G, true, 2, false>(a, s);
                case 3: return a.N == 12 ? launch_sweep<MODE, G, true, 3, false, false, 12>(a, s) : launch_sweep<MODE, G, true, 3, false>(a, s);
                default: break;
                }
            if (!pat && !a.red && a.N <= 5)  // five candidates for new_k: the small cells of measure_min_energy_vs_N (BASELINE config 4)
                return a.N <= 4 ? launch_sweep<MODE, G, false, 1, false, false, 0, false, true>(a, s) : launch_sweep<MODE, G, false, 2, false, false, 0, false, true>(a, s);
            if (!pat && a.N == 12)  // the size of BASELINE config 2: N as a compile-time constant
                return a.red ? launch_sweep<MODE, G, false, 3, true, false, 12>(a, s) : launch_sweep<MODE, G, false, 3, false, false, 12>(a, s);
#ifndef MCQ_EXP_NO_NC5  // (timing experiment: without these instantiations)
            // the long cells of measure_min_energy_vs_N (BASELINE configs[3]): N as a compile-time constant (125 / 119 / 118 VGPRs and 10 / 7 / 4
            // spilled SGPRs against 128 / 40 of the generic five-pass variant; N = 19 is left out: its instantiation spills 324 VGPRs)
            if (!pat && !a.red && (a.N == 17 || a.N == 18 || a.N == 20)) switch (a.N) {
                case 17: return launch_sweep<MODE, G, false, 5, false, false, 17>(a, s);
                case 18: return launch_sweep<MODE, G, false, 5, false, false, 18>(a, s);
                default: return launch_sweep<MODE, G, false, 5, false, false, 20>(a, s);
                }
#endif
            if (!pat) switch ((a.N + G - 1) / G) {
#define MCQ_NT_CASE(nt) case nt: return a.red ? launch_sweep<MODE, G, false, (nt >= 4 ? 0 : nt), true>(a, s) : launch_sweep<MODE, G, false, nt, false>(a, s)
                MCQ_NT_CASE(1);  // N = 2..4
                MCQ_NT_CASE(2);  // N = 5..8
                MCQ_NT_CASE(3);  // N = 9..12
                MCQ_NT_CASE(4);  // N = 13..16 (up to here: packed 16-bit masks); from four passes on the reduced-trace variants would spill and take the loop
                MCQ_NT_CASE(5);  // N = 17..20
                MCQ_NT_CASE(6);  // N = 21..24
#undef MCQ_NT_CASE
                default: break;
                }
        }
        if constexpr (G == 8) {  // larger boards run 8 lanes per chain by default: 3 or 4 straight-line probe passes
            // (a launch that leaves the device at most half full -- two wavefronts per SIMD -- takes the variants that request the probe
            // heights early: see EARLY_PROBES in the kernel)
            const bool roomy = (a.n_chains + 7) / 8 <= 2LL * device_simds();
            if (!pat && a.red && a.N == 24)  // BASELINE config 5: the beta-pair driver's shape (N = 24, reduced trace)
                return roomy ? launch_sweep<MODE, G, false, 3, true, false, 24, false, false, true>(a, s) : launch_sweep<MODE, G, false, 3, true, false, 24>(a, s);
            if (!pat && a.red && a.N > 16 && a.N <= 24) return roomy ? launch_sweep<MODE, G, false, 3, true, false, 0, false, false, true>(a, s) : launch_sweep<MODE, G, false, 3, true>(a, s);
            if (!pat && a.red && a.N > 8 && a.N <= 16) return launch_sweep<MODE, G, false, 2, true>(a, s);
            if (!pat && !a.red && a.N > 16 && a.N <= 24) return roomy ? launch_sweep<MODE, G, false, 3, false, false, 0, false, false, true>(a, s) : launch_sweep<MODE, G, false, 3, false>(a, s);
            if (!pat && !a.red && a.N > 24 && a.N <= 32) return launch_sweep<MODE, G, false, 4, false>(a, s);
            if (!pat && !a.red && a.N > 8 && a.N <= 16) return launch_sweep<MODE, G, false, 2, false>(a, s);  // N = 9..16: two packed passes
            if (pat && !a.red && a.N > 16 && a.N <= 24) return roomy ? launch_sweep<MODE, G, true, 3, false, false, 0, false, false, true>(a, s) : launch_sweep<MODE, G, true, 3, false>(a, s);
            if (pat && !a.red && a.N > 24 && a.N <= 32) return launch_sweep<MODE, G, true, 4, false>(a, s);
            if (pat && !a.red && a.N > 8 && a.N <= 16) return launch_sweep<MODE, G, true, 2, false>(a, s);
        }
        if constexpr (G == 16) {  // 4 chains per wavefront: one packed probe pass up to N = 16, two unpacked ones up to N = 32
            // (what a launch far below the device's capacity takes -- 16 384 chains of N = 24, the per-GPU shape of BASELINE configs[4], are
            // four wavefronts per SIMD this way; before round 4 these widths ran the run-time probe loop)
            const bool roomy = (a.n_chains + 3) / 4 <= 2LL * device_simds();
            if (!pat && a.N <= 16) return a.red ? launch_sweep<MODE, G, false, 1, true>(a, s) : launch_sweep<MODE, G, false, 1, false>(a, s);
            if (!pat && a.red && a.N == 24) return launch_sweep<MODE, G, false, 2, true, false, 24>(a, s);
            if (!pat && a.N <= 32) {
                if (a.red) return launch_sweep<MODE, G, false, 2, true>(a, s);
                return roomy ? launch_sweep<MODE, G, false, 2, false, false, 0, false, false, true>(a, s) : launch_sweep<MODE, G, false, 2, false>(a, s);
            }
        }
        if (a.red) return pat ? launch_sweep<MODE, G, true, 0, true>(a, s) : launch_sweep<MODE, G, false, 0, true>(a, s);
        return pat ? launch_sweep<MODE, G, true, 0, false>(a, s) : launch_sweep<MODE, G, false, 0, false>(a, s);
    }
}

template <int MODE>
int launch_sweep_mode(const KArgs& a, int G, hipStream_t s) {
    if (G == 2) {
        if constexpr (MODE == MCQ_MODE_BOARD) return launch_sweep_g<MODE, 2>(a, s);
        else return fail(MCQ_EINVAL, "lanes_per_chain 2 applies to mcmc_type board");
    }
    if (G == 4) return launch_sweep_g<MODE, 4>(a, s);
    if (G == 8) return launch_sweep_g<MODE, 8>(a, s);
    return launch_sweep_g<MODE, 16>(a, s);
}

int run_device_impl(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, void* workspace,
                    size_t workspace_bytes, void* hip_stream, hipEvent_t* ev, const double* beta_table_dev) {
    int rc = validate(p);
    if (rc != MCQ_OK) return rc;
    if (!seeds || !out || !workspace) return fail(MCQ_EINVAL, "null argument");
    if (workspace_bytes < mcq_workspace_bytes(p)) return fail(MCQ_ENOMEM, "workspace too small");
    // the sweep reads the MT words of a record with 16-byte loads and flushes the trace in aligned 64-byte segments
    if (((uintptr_t)workspace & 63u) != 0) return fail(MCQ_EINVAL, "workspace must be 64-byte aligned");
    if ((((uintptr_t)out->best_state | (uintptr_t)out->final_state) & 15u) != 0) return fail(MCQ_EINVAL, "best_state / final_state must be 16-byte aligned");
    if (p->trace == MCQ_TRACE_I32) {
        if (!out->energy_hist || !out->accept_bits) return fail(MCQ_EINVAL, "trace requested without buffers");
        if (p->hist_stride < p->n_steps + 1) return fail(MCQ_EINVAL, "hist_stride too small");
        if (p->hist_stride % 16 != 0) return fail(MCQ_EINVAL, "hist_stride must be a multiple of 16");
        if (p->hist_stride >= (1LL << 24) || p->bits_stride >= (1LL << 24))
            return fail(MCQ_EINVAL, "a full trace takes at most 2^24 - 16 entries per chain; use trace = reduced (or none) for longer runs");
        if (((uintptr_t)out->energy_hist & 63u) != 0) return fail(MCQ_EINVAL, "energy_hist must be 64-byte aligned");
        if (p->bits_stride < (p->n_steps + 63) / 64) return fail(MCQ_EINVAL, "bits_stride too small");
    }
    if (p->trace == MCQ_TRACE_REDUCED && (!out->step_sum || !out->step_sumsq || !out->step_accepted || !out->step_count))
        return fail(MCQ_EINVAL, "reduced trace requested without step_sum / step_sumsq / step_accepted / step_count");
    hipStream_t s = (hipStream_t)hip_stream;
    if (p->n_chains == 0) {
        if (p->trace == MCQ_TRACE_REDUCED)
            for (int64_t* arr : {out->step_sum, out->step_sumsq, out->step_accepted, out->step_count})
                HIP_TRY(hipMemsetAsync(arr, 0, n_sets_of(p) * (size_t)(p->n_steps + 1) * 8, s));
        if (ev)
            for (int t = 0; t < 3; t++) HIP_TRY(hipEventRecord(ev[t], s));
        return MCQ_OK;
    }
    KArgs a;
    rc = build_args(p, seeds, out, workspace, &a);
    if (rc != MCQ_OK) return rc;

    const int G = effective_lanes(p);
    {  // the variant this launch takes must fit the LDS: found out before anything is enqueued
        KArgs d = a;
        d.dry = 1;
        rc = p->mode == MCQ_MODE_BOARD ? launch_sweep_mode<MCQ_MODE_BOARD>(d, G, s) : launch_sweep_mode<MCQ_MODE_FULL3D>(d, G, s);
        if (rc != MCQ_OK) return rc;
    }
    if (a.out.accept_bits)  // chains that stop early leave their later words untouched
        HIP_TRY(hipMemsetAsync(a.out.accept_bits, 0, (size_t)p->n_chains * p->bits_stride * 8, s));

    if (a.red) HIP_TRY(hipMemsetAsync(a.red, 0, red_bytes(p), s));
    if (p->flags & MCQ_FLAG_SHARED_PACING) {
        // The launches of a job list that run side by side pace their wavefronts against EACH OTHER: one progress table per device for all
        // launches that set the flag (a wavefront's row and slot come from the hardware's SIMD / wave-slot ids, which are unique whatever
        // kernel a wavefront belongs to), never cleared in between -- a wavefront that ends writes a zero, a new one publishes after 64 steps.
        // (allocated and cleared once per device, under a lock: the first flagged call of a process synchronises with the device, the later ones do not)
        static uint32_t* shared_pace[64] = {};
        static std::mutex shared_pace_lock;
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64) return fail(MCQ_EDEVICE, "device ordinal out of range");
        {
            std::lock_guard<std::mutex> guard(shared_pace_lock);
            if (!shared_pace[dev]) {
                uint32_t* t = nullptr;
                HIP_TRY(hipMalloc((void**)&t, PACE_BYTES));
                if (hipMemset(t, 0, PACE_BYTES) != hipSuccess) {
                    (void)hipFree(t);
                    return fail(MCQ_EDEVICE, "hipMemset of the shared progress table failed");
                }
                shared_pace[dev] = t;
            }
            a.pace = shared_pace[dev];
        }
    } else {
        HIP_TRY(hipMemsetAsync(a.pace, 0, PACE_BYTES, s));
    }
    if (p->exchange_every > 0)  // 16 doubles at most, from the caller's (host) array
        HIP_TRY(hipMemcpyAsync((void*)a.exch_ladder, p->exchange_ladder, (size_t)p->exchange_replicas * 8, hipMemcpyHostToDevice, s));
    if (a.stream) {  // the caller's MT19937 states in the kernels' layout; the staging array lives until the copy has run (hence the synchronisation)
        std::vector<uint32_t> staged((size_t)p->n_chains * 626);
        for (int64_t r = 0; r < p->n_chains; r++) stream_layout(p->stream_states + r * 625, staged.data() + r * 626);
        HIP_TRY(hipMemcpyAsync((void*)a.stream, staged.data(), staged.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
    // init kernel: LDS per chain = the MT19937 words, the state, and behind it the permutation array of np.random.choice (full_3d random
    // init), later one family of E0 line counters; four chains per wavefront while a CU still holds 8 wavefronts of them, else two or one (full_3d N = 12: 6.7 ms at two, 8.2 ms at four, 7.8 ms at one)
    size_t init_lds = (size_t)MT_N * 4 + ((a.state_bytes + 3) & ~3);
    {
        const size_t D = 2 * (size_t)p->N - 1, lines = ((D * D + 3) / 4) * 4;
        const size_t perm = p->mode == MCQ_MODE_FULL3D && any_random_init(p) && !a.perm ? (size_t)p->N * p->N * p->N * 2 : 0;  // (beyond N = 32: in global memory)
        init_lds += perm > lines ? perm : lines;
        init_lds = (init_lds + 15) & ~(size_t)15;
    }
    if (init_lds > 160 * 1024) return fail(MCQ_EINVAL, "initial state does not fit in LDS (N^3 permutation array + 3 Q state bytes)");
    const int init_ci = 4 * init_lds * 8 <= 160 * 1024 ? 4 : 2 * init_lds * 8 <= 160 * 1024 ? 2 : 1;
    a.init_words = (int)(init_lds / 4);
    auto launch_init_part = [&](const KArgs& k, long long count) -> hipError_t {  // chains k.chain0 .. k.chain0 + count - 1
        const size_t bytes = (size_t)init_ci * init_lds;
        const unsigned grid = (unsigned)((count + init_ci - 1) / init_ci);
        hipError_t e;
        if (init_ci == 4) {
            e = hipFuncSetAttribute((const void*)mcq_init_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) hipLaunchKernelGGL(mcq_init_kernel<4>, dim3(grid), dim3(64), bytes, s, k);
        } else if (init_ci == 2) {
            e = hipFuncSetAttribute((const void*)mcq_init_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) hipLaunchKernelGGL(mcq_init_kernel<2>, dim3(grid), dim3(64), bytes, s, k);
        } else {
            e = hipFuncSetAttribute((const void*)mcq_init_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) hipLaunchKernelGGL(mcq_init_kernel<1>, dim3(grid), dim3(64), bytes, s, k);
        }
        return e;
    };
    auto launch_init = [&](const KArgs& k0) -> hipError_t {
        if (!k0.perm) return launch_init_part(k0, k0.n_chains);
        const long long slots = perm_slots_for(p);  // the permutation slices are shared: one launch after the other on the stream
        KArgs k = k0;
        for (k.chain0 = 0; k.chain0 < k0.n_chains; k.chain0 += slots) {
            const hipError_t e = launch_init_part(k, k0.n_chains - k.chain0 < slots ? k0.n_chains - k.chain0 : slots);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    if (ev) HIP_TRY(hipEventRecord(ev[0], s));
    if (p->n_steps > 0) {
        for (size_t t = 0; t < n_sets_of(p); t++) {  // one table pair per schedule set
            KArgs b = a;
            if (p->n_sets > 1) b.sched = p->sets[t].sched, b.beta_const = p->sets[t].beta_const, b.beta_start = p->sets[t].beta_start, b.beta_end = p->sets[t].beta_end;
            b.beta_tab = a.beta_tab + t * a.tab_stride, b.c32_tab = a.c32_tab + t * a.tab_stride;
            if (beta_table_dev)
                hipLaunchKernelGGL(mcq_beta_copy_kernel, dim3((unsigned)((p->n_steps + 255) / 256)), dim3(256), 0, s,
                                   beta_table_dev + t * (size_t)p->n_steps, b.beta_tab, b.c32_tab, (long long)p->n_steps);
            else
                hipLaunchKernelGGL(mcq_beta_kernel, dim3((unsigned)((p->n_steps + 255) / 256)), dim3(256), 0, s, b);
        }
    }
    {  // sets may start from different init modes (one launch of the init kernel per distinct run of sets)
        bool mixed = false;
        for (size_t t = 0; t < n_sets_of(p) && p->n_sets > 1; t++) mixed |= p->sets[t].init_plus1 != 0;
        if (!mixed) {
            HIP_TRY(launch_init(a));
        } else {
            for (size_t t = 0; t < n_sets_of(p); t++) {
                KArgs b = a;
                b.init = p->sets[t].init_plus1 ? p->sets[t].init_plus1 - 1 : p->init;
                b.klarner_M = 0;
                if (b.init == MCQ_INIT_KLARNER && gcd_int(p->N, 210) != 1)
                    for (int m = p->N - 1; m > 0; m--)
                        if (gcd_int(m, 210) == 1) {
                            b.klarner_M = m;
                            break;
                        }
                b.seeds = a.seeds + t * (size_t)p->chains_per_set;
                b.ws = a.ws + t * (size_t)p->chains_per_set * a.rec_words;
                if (a.qtab) b.qtab = a.qtab + t * (size_t)p->chains_per_set * a.qtab_stride * (p->N > 32 ? 2 : 1);
                if (a.stream) b.stream = a.stream + t * (size_t)p->chains_per_set * 626;
                if (a.out.stream_words) b.out.stream_words = a.out.stream_words + t * (size_t)p->chains_per_set;
                b.n_chains = p->chains_per_set;
                HIP_TRY(launch_init(b));
            }
        }
    }
    HIP_TRY(hipGetLastError());
    if (ev) HIP_TRY(hipEventRecord(ev[1], s));

    rc = p->mode == MCQ_MODE_BOARD ? launch_sweep_mode<MCQ_MODE_BOARD>(a, G, s) : launch_sweep_mode<MCQ_MODE_FULL3D>(a, G, s);
    if (rc != MCQ_OK) return rc;
    if (out->stream_words && p->rng == MCQ_RNG_MT19937_NUMPY) {
        hipLaunchKernelGGL(mcq_stream_words_kernel, dim3((unsigned)((p->n_chains + 255) / 256)), dim3(256), 0, s, out->stream_words, a.ws, a.rec_words, (long long)p->n_chains);
        HIP_TRY(hipGetLastError());
    }
    if (a.red) {
        const long long n_entries = p->n_steps + 1;
        for (size_t t = 0; t < n_sets_of(p); t++)
            hipLaunchKernelGGL(mcq_reduced_finalize_kernel, dim3((unsigned)((n_entries + 255) / 256)), dim3(256), 0, s, a.red + t * a.red_set_stride,
                               a.red_len, n_entries, (long long*)out->step_sum + t * n_entries, (long long*)out->step_sumsq + t * n_entries,
                               (long long*)out->step_accepted + t * n_entries, (long long*)out->step_count + t * n_entries);
        HIP_TRY(hipGetLastError());
    }
    if (ev) HIP_TRY(hipEventRecord(ev[2], s));
    return MCQ_OK;
}

}  // namespace

extern "C" {

int mcq_abi_version(void) { return MCQ_ABI_VERSION; }

const char* mcq_last_error(void) { return g_err; }

int mcq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int32_t mcq_default_lanes(int32_t mode) { return mode == MCQ_MODE_BOARD ? 4 : 8; }

// board: 4 lanes up to N = 12, where 16 chains per wavefront still fit 16 wavefronts per CU (<= 640 B of LDS per chain), and 8
// beyond (65 536 chains x 20 000 steps: N = 13 42 ms against 49 ms, N = 16 42 / 52, N = 20 56 / 70, N = 24 56 / 73); full_3d: 8
int32_t mcq_default_lanes_n(int32_t mode, int32_t N) { return mode == MCQ_MODE_BOARD && N > 12 ? 8 : mode == MCQ_MODE_FULL3D && N > 32 ? 16 : mcq_default_lanes(mode); }

int32_t mcq_effective_lanes(const mcq_params* p) { return validate(p) == MCQ_OK ? effective_lanes(p) : 0; }

void mcq_stream_layout(const uint32_t* numpy_state, uint32_t* out) { stream_layout(numpy_state, out); }

int32_t mcq_device_simds(void) { return device_simds(); }

size_t mcq_state_bytes(int32_t N, int32_t mode) {
    if (N < MCQ_MIN_N || N > (mode == MCQ_MODE_BOARD ? MCQ_MAX_N_BOARD : MCQ_MAX_N)) return 0;
    return mode == MCQ_MODE_BOARD ? (size_t)N * N : (size_t)3 * N * N;
}

size_t mcq_state_bytes_for(const mcq_params* p) { return validate(p) == MCQ_OK ? state_bytes_of(p) : 0; }

size_t mcq_workspace_bytes(const mcq_params* p) {
    if (validate(p) != MCQ_OK) return 0;
    const size_t chains = (size_t)(p->n_chains > 0 ? p->n_chains : 1);
    return beta_tab_bytes(p) + c32_tab_bytes(p) + red_bytes(p) + PACE_BYTES + LADDER_BYTES + chains * rec_words_for(p) * 4 + qtab_bytes_for(p) + perm_bytes_for(p) + stream_bytes_for(p);
}

int mcq_run_device(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, void* workspace,
                   size_t workspace_bytes, void* hip_stream) {
    return run_device_impl(p, seeds, out, workspace, workspace_bytes, hip_stream, nullptr, p ? p->beta_table : nullptr);
}

int mcq_run_device_timed(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, void* workspace,
                         size_t workspace_bytes, void* hip_stream, float* init_ms, float* sweep_ms) {
    struct Events {  // destroyed on every return path
        hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
        ~Events() {
            for (auto& e : ev)
                if (e) (void)hipEventDestroy(e);
        }
    } evs;
    hipEvent_t* ev = evs.ev;
    for (int t = 0; t < 3; t++) HIP_TRY(hipEventCreate(&ev[t]));
#ifdef MCQ_STAMPS
    if (!g_dbg) HIP_TRY(hipMalloc((void**)&g_dbg, 96));
    HIP_TRY(hipMemset(g_dbg, 0, 96));
#endif
#ifdef MCQ_WAVE_TIMES
    const size_t wt_waves = 1 << 16;
    if (!g_dbg) HIP_TRY(hipMalloc((void**)&g_dbg, wt_waves * 32));
    HIP_TRY(hipMemset(g_dbg, 0, wt_waves * 32));
#endif
    int rc = run_device_impl(p, seeds, out, workspace, workspace_bytes, hip_stream, ev, p ? p->beta_table : nullptr);
#ifdef MCQ_STAMPS
    if (rc == MCQ_OK) {
        unsigned long long h[12];
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h, g_dbg, 96, hipMemcpyDeviceToHost));
        const char* names[8] = {"loop", "stream_upkeep", "draws", "dE", "accept", "apply+history", "-", "-"};
        fprintf(stderr, "STAMP word-by-word draws: %llu of %llu wavefront-steps (%.3f %%)\n", h[6], h[7], 100.0 * h[6] / (h[7] ? h[7] : 1));
        fprintf(stderr, "STAMP float64 accept path %.3f %%, improvement path %.3f %%, second triple (full_3d) %.3f %%, stream upkeep %.1f %% of the wavefront-steps\n",
                100.0 * h[8] / (h[7] ? h[7] : 1), 100.0 * h[9] / (h[7] ? h[7] : 1), 100.0 * h[10] / (h[7] ? h[7] : 1), 100.0 * h[11] / (h[7] ? h[7] : 1));
        unsigned long long tot = 0;
        for (int k = 0; k < 6; k++) tot += h[k];
        for (int k = 0; k < 6; k++) fprintf(stderr, "STAMP %-14s %14llu  %5.1f %%\n", names[k], h[k], 100.0 * h[k] / (tot ? tot : 1));
    }
#endif
#ifdef MCQ_WAVE_TIMES
    if (rc == MCQ_OK) {  // one line per wavefront: start and end in 10 ns ticks from the first start, XCC, SE, CU, SIMD
        HIP_TRY(hipDeviceSynchronize());
        unsigned long long* h = (unsigned long long*)malloc(wt_waves * 32);
        if (!h) return fail(MCQ_ENOMEM, "out of host memory");
        const hipError_t ce = hipMemcpy(h, g_dbg, wt_waves * 32, hipMemcpyDeviceToHost);
        if (ce != hipSuccess) {
            free(h);
            return fail(MCQ_EDEVICE, "hipMemcpy: %s", hipGetErrorString(ce));
        }
        unsigned long long t0 = ~0ull;
        for (size_t w = 0; w < wt_waves; w++)
            if (h[4 * w + 1] && h[4 * w] < t0) t0 = h[4 * w];
        if (FILE* f = fopen(getenv("MCQ_WAVE_TIMES_OUT") ? getenv("MCQ_WAVE_TIMES_OUT") : "/tmp/mcq_wave_times.txt", "w")) {
            for (size_t w = 0; w < wt_waves; w++)
                if (h[4 * w + 1]) {
                    const unsigned hw = (unsigned)h[4 * w + 3];
                    fprintf(f, "%zu %llu %llu %u %u %u %u %u\n", w, h[4 * w] - t0, h[4 * w + 1] - t0, (unsigned)h[4 * w + 2], (hw >> 13) & 3, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15);
                }
            fclose(f);
        }
        free(h);
    }
#endif
    float a = 0.f, b = 0.f;
    if (rc == MCQ_OK) {
        hipError_t e = hipEventSynchronize(ev[2]);
        if (e == hipSuccess) e = hipEventElapsedTime(&a, ev[0], ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&b, ev[1], ev[2]);
        if (e != hipSuccess) rc = fail(MCQ_EDEVICE, "event timing: %s", hipGetErrorString(e));
    }
    if (init_ms) *init_ms = a;
    if (sweep_ms) *sweep_ms = b;
    return rc;
}

int mcq_trace_stats_device(const mcq_params* p, const mcq_outputs* out, int64_t* step_sum, int64_t* step_sumsq, int64_t* step_count,
                           int32_t n_bins, const int64_t* bin_lo, uint64_t* bin_accepted, uint64_t* bin_proposed, void* hip_stream) {
    int rc = validate(p);
    if (rc != MCQ_OK) return rc;
    if (!out || !out->energy_hist || !out->accept_bits || !out->hist_len || !out->steps_executed)
        return fail(MCQ_EINVAL, "trace statistics need energy_hist, accept_bits, hist_len and steps_executed");
    if (p->hist_stride < p->n_steps + 1 || p->bits_stride < (p->n_steps + 63) / 64) return fail(MCQ_EINVAL, "stride too small");
    hipStream_t s = (hipStream_t)hip_stream;
    const long long n_entries = p->n_steps + 1;
    if (step_sum) {
        if (!step_sumsq || !step_count) return fail(MCQ_EINVAL, "step_sum, step_sumsq and step_count go together");
        hipLaunchKernelGGL(mcq_step_stats_kernel, dim3((unsigned)((n_entries + 63) / 64)), dim3(256), 0, s, out->energy_hist, out->hist_len,
                           (long long)p->n_chains, (long long)p->hist_stride, n_entries, (long long*)step_sum, (long long*)step_sumsq,
                           (long long*)step_count);
        HIP_TRY(hipGetLastError());
    }
    if (n_bins > 0) {
        if (!bin_lo || !bin_accepted || !bin_proposed) return fail(MCQ_EINVAL, "bin_lo, bin_accepted and bin_proposed go together");
        HIP_TRY(hipMemsetAsync(bin_accepted, 0, (size_t)n_bins * 8, s));
        HIP_TRY(hipMemsetAsync(bin_proposed, 0, (size_t)n_bins * 8, s));
        const long long total = (long long)p->n_chains * n_bins;
        if (total > 0) {
            hipLaunchKernelGGL(mcq_accept_bins_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                               (const unsigned long long*)out->accept_bits, out->steps_executed, (long long)p->n_chains,
                               (long long)p->bits_stride, (int)n_bins, bin_lo, (unsigned long long*)bin_accepted,
                               (unsigned long long*)bin_proposed);
            HIP_TRY(hipGetLastError());
        }
    }
    return MCQ_OK;
}

int mcq_pack_summary_device(const mcq_params* p, const mcq_outputs* out, int64_t n_local, const mcq_pack_slot* slots, int64_t* packed, void* hip_stream) {
    int rc = validate(p);
    if (rc != MCQ_OK) return rc;
    if (!out || !slots || !packed) return fail(MCQ_EINVAL, "null argument");
    if (!out->best_energy || !out->steps_to_best || !out->n_accepted || !out->steps_executed || !out->hist_len)
        return fail(MCQ_EINVAL, "the packed summary needs best_energy, steps_to_best, n_accepted, steps_executed and hist_len");
    const long long sets = (long long)n_sets_of(p), cps = p->n_sets > 1 ? p->chains_per_set : p->n_chains;
    if (n_local < 0 || n_local > cps) return fail(MCQ_EINVAL, "n_local out of range");
    bool stats = false;
    for (long long t = 0; t < sets; t++) {
        if (slots[t].counters < 0 || slots[t].min_slot < 0) return fail(MCQ_EINVAL, "a pack slot needs its counters and its minimum slot");
        stats |= slots[t].stats >= 0;
    }
    if (stats && (p->trace != MCQ_TRACE_REDUCED || !out->step_sum || !out->step_sumsq || !out->step_accepted || !out->step_count))
        return fail(MCQ_EINVAL, "per-entry statistics are packed from the reduced trace (trace == REDUCED with its four step_* arrays)");
    hipStream_t s = (hipStream_t)hip_stream;
    for (long long t0 = 0; t0 < sets; t0 += PACK_SETS) {
        PackArgs a;
        memset(&a, 0, sizeof a);
        const int n = (int)(sets - t0 < PACK_SETS ? sets - t0 : PACK_SETS);
        for (int k = 0; k < n; k++) a.slot[k] = slots[t0 + k];
        a.best = out->best_energy, a.stb = out->steps_to_best, a.acc = out->n_accepted, a.exe = out->steps_executed, a.hist_len = out->hist_len;
        a.step_sum = out->step_sum, a.step_sumsq = out->step_sumsq, a.step_accepted = out->step_accepted, a.step_count = out->step_count;
        a.cps = cps, a.n_local = n_local, a.n_steps = p->n_steps, a.first_set = t0, a.packed = packed;
        hipLaunchKernelGGL(mcq_pack_kernel, dim3((unsigned)n), dim3(256), 0, s, a);
        if (stats) hipLaunchKernelGGL(mcq_pack_stats_kernel, dim3((unsigned)((p->n_steps + 1 + 255) / 256), (unsigned)n), dim3(256), 0, s, a);
    }
    HIP_TRY(hipGetLastError());
    return MCQ_OK;
}

int mcq_beta_table_device(const mcq_params* p, double* beta_out, float* c32_out, void* hip_stream) {
    int rc = validate(p);
    if (rc != MCQ_OK) return rc;
    if (!beta_out) return fail(MCQ_EINVAL, "null argument");
    if (p->n_steps == 0) return MCQ_OK;
    hipStream_t s = (hipStream_t)hip_stream;
    KArgs a;
    memset(&a, 0, sizeof a);
    a.n_steps = p->n_steps;
    for (size_t t = 0; t < n_sets_of(p); t++) {
        if (p->n_sets > 1) a.sched = p->sets[t].sched, a.beta_const = p->sets[t].beta_const, a.beta_start = p->sets[t].beta_start, a.beta_end = p->sets[t].beta_end;
        else a.sched = p->sched, a.beta_const = p->beta_const, a.beta_start = p->beta_start, a.beta_end = p->beta_end;
        a.beta_tab = beta_out + t * (size_t)p->n_steps;
        a.c32_tab = c32_out ? c32_out + t * (size_t)p->n_steps : nullptr;
        hipLaunchKernelGGL(mcq_beta_kernel, dim3((unsigned)((p->n_steps + 255) / 256)), dim3(256), 0, s, a);  // always the device's own evaluation
    }
    HIP_TRY(hipGetLastError());
    return MCQ_OK;
}

int mcq_run_host(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, double* kernel_seconds) {
    int rc = validate(p);
    if (rc != MCQ_OK) return rc;
    if (!seeds || !out) return fail(MCQ_EINVAL, "null argument");
    if (p->device >= 0) HIP_TRY(hipSetDevice(p->device));
    if (kernel_seconds) *kernel_seconds = 0.0;
    if (p->n_chains == 0) return MCQ_OK;

    const size_t n = (size_t)p->n_chains, sb = state_bytes_of(p);
    struct Buf {
        void** dev;
        void* host;
        size_t bytes;
    };
    mcq_outputs d;
    memset(&d, 0, sizeof d);
    const bool tr = p->trace == MCQ_TRACE_I32;
    Buf bufs[] = {
        {(void**)&d.energy_hist, tr ? out->energy_hist : nullptr, n * (size_t)p->hist_stride * 4},
        {(void**)&d.accept_bits, tr ? out->accept_bits : nullptr, n * (size_t)p->bits_stride * 8},
        {(void**)&d.hist_len, out->hist_len, n * 8},
        {(void**)&d.steps_executed, out->steps_executed, n * 8},
        {(void**)&d.initial_energy, out->initial_energy, n * 4},
        {(void**)&d.best_energy, out->best_energy, n * 4},
        {(void**)&d.final_energy, out->final_energy, n * 4},
        {(void**)&d.steps_to_best, out->steps_to_best, n * 8},
        {(void**)&d.n_accepted, out->n_accepted, n * 8},
        {(void**)&d.near_ties, out->near_ties, n * 8},
        {(void**)&d.stream_words, out->stream_words, n * 4},
        {(void**)&d.exchange_rung, p->exchange_every > 0 ? out->exchange_rung : nullptr, n * 4},
        {(void**)&d.n_exchanges, p->exchange_every > 0 ? out->n_exchanges : nullptr, n * 8},
        {(void**)&d.best_state, out->best_state, n * sb},
        {(void**)&d.final_state, out->final_state, n * sb},
        {(void**)&d.step_sum, p->trace == MCQ_TRACE_REDUCED ? out->step_sum : nullptr, n_sets_of(p) * (size_t)(p->n_steps + 1) * 8},
        {(void**)&d.step_sumsq, p->trace == MCQ_TRACE_REDUCED ? out->step_sumsq : nullptr, n_sets_of(p) * (size_t)(p->n_steps + 1) * 8},
        {(void**)&d.step_accepted, p->trace == MCQ_TRACE_REDUCED ? out->step_accepted : nullptr, n_sets_of(p) * (size_t)(p->n_steps + 1) * 8},
        {(void**)&d.step_count, p->trace == MCQ_TRACE_REDUCED ? out->step_count : nullptr, n_sets_of(p) * (size_t)(p->n_steps + 1) * 8},
    };
    uint32_t* d_seeds = nullptr;
    void* d_ws = nullptr;
    double* d_beta = nullptr;
    const size_t ws_bytes = mcq_workspace_bytes(p);
    auto cleanup = [&]() {
        for (auto& b : bufs)
            if (*b.dev) (void)hipFree(*b.dev);
        if (d_seeds) (void)hipFree(d_seeds);
        if (d_ws) (void)hipFree(d_ws);
        if (d_beta) (void)hipFree(d_beta);
    };
#define HOST_TRY(expr)                                                                                              \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) {                                                                                     \
            cleanup();                                                                                              \
            return fail(e_ == hipErrorOutOfMemory ? MCQ_ENOMEM : MCQ_EDEVICE, #expr ": %s", hipGetErrorString(e_)); \
        }                                                                                                           \
    } while (0)
    for (auto& b : bufs)
        if (b.host) HOST_TRY(hipMalloc(b.dev, b.bytes));
    HOST_TRY(hipMalloc((void**)&d_seeds, n * 4));
    HOST_TRY(hipMalloc(&d_ws, ws_bytes));
    HOST_TRY(hipMemcpy(d_seeds, seeds, n * 4, hipMemcpyHostToDevice));
    mcq_params pd = *p;
    if (p->beta_table && p->n_steps > 0) {  // the host table goes to the device
        const size_t tb = n_sets_of(p) * (size_t)p->n_steps * 8;
        HOST_TRY(hipMalloc((void**)&d_beta, tb));
        HOST_TRY(hipMemcpy(d_beta, p->beta_table, tb, hipMemcpyHostToDevice));
    }
    pd.beta_table = d_beta;
    float i_ms = 0.f, s_ms = 0.f;
    rc = mcq_run_device_timed(&pd, d_seeds, &d, d_ws, ws_bytes, nullptr, &i_ms, &s_ms);
    if (rc != MCQ_OK) {
        cleanup();
        return rc;
    }
    if (kernel_seconds) *kernel_seconds = (i_ms + s_ms) * 1e-3;
    for (auto& b : bufs)
        if (b.host) HOST_TRY(hipMemcpy(b.host, *b.dev, b.bytes, hipMemcpyDeviceToHost));
    cleanup();
    return MCQ_OK;
#undef HOST_TRY
}

}  // extern "C"
