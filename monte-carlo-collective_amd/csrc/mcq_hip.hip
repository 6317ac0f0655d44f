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
// Design: a wavefront of 64 lanes is split into groups of G lanes (G = 16,
// 32 or 64); one group runs one chain.  Everything that is serial in a chain (the NumPy-legacy
// MT19937 stream with its data-dependent word consumption, the proposal, the accept test) is
// computed redundantly by the G lanes of the group; the attack count is spread over the lanes.
//   * MT19937 state: 624 words per chain in LDS, regenerated lazily G words at a time (each
//     lane twists one word, tempers it and keeps it in a register "window"); a draw is one
//     ds_bpermute from the window.
//   * board dE: the only columns that can attack cell (i,j,k) lie on the row, the column and
//     the two diagonals of (i,j) in the ij-plane, at most 4N of them; a column at in-plane
//     distance d attacks iff |h - k| is 0 or d.  4N column probes over G lanes + a DPP
//     all-reduce give conflicts(new) - conflicts(old) without any per-line counters.
//   * energy_history: each group stages G consecutive entries in one register and stores
//     them as one aligned G*4-byte segment; accept bits as one 64-bit word per 64 steps.
//   * beta(step) is evaluated on device in float64, 64 steps at a time (lane L computes
//     step0 + L), strict IEEE (compiled with -ffp-contract=off).
//   * exp(-beta*dE) is bracketed by a float32 estimate; only when the uniform falls inside
//     the bracket (|u/p - 1| < 2^-10) is the float64 exp evaluated, so every decision equals
//     the all-float64 decision (MCQ_FLAG_EXACT_EXP forces float64 on every step for testing).
//
// This file is compiled for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../include/mcq.h"

namespace {

constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int REC_POS = 624;      // record word: index of the next MT word to consume
constexpr int REC_GEN_END = 625;  // record word: words [0, gen_end) belong to the current generation
constexpr int REC_E0 = 626;       // record word: initial energy
constexpr int REC_STATE = 628;    // first word of the state bytes (heights or (i,j,k) triplets)

struct KArgs {
    int N, Q, mode, init, sched;
    unsigned flags;
    unsigned maskN, maskQ;  // smallest 2^b - 1 >= N-1 / Q-1 (masked rejection)
    int klarner_M;          // 0: exact Klarner (gcd(N,210)==1); else core edge M
    int state_bytes;
    int rec_words;          // words per chain record in the workspace
    int chain_lds_words;    // words of LDS per chain in the sweep kernel
    double beta_const, beta_start, beta_end;
    long long n_steps, n_chains, patience, hist_stride, bits_stride;
    uint32_t* ws;
    const uint32_t* seeds;
    mcq_outputs out;
};

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// NumPy-legacy MT19937 stream of one chain, shared by the G lanes of its group.
//   mt       LDS, 624 raw state words
//   pos      next word to consume, 0..623
//   gen_end  words [0, gen_end) already belong to the current generation (multiple of 64, or 624)
//   win      tempered word (chunk base + lane-in-group) of the chunk that contains pos
// Word i of a generation depends on words i, i+1 and (i+397) mod 624, the last one from the
// current generation when i >= 227 and the middle one when i == 623; regenerating chunks of
// G <= 64 consecutive words in increasing order on demand therefore yields exactly the words
// of the all-at-once twist in mt19937_gen (NumPy: _mt19937/mt19937.c) -- every lane reads its
// three inputs before any lane of the chunk writes.
template <int G>
struct Rng {
    uint32_t* mt;
    uint32_t win;
    int pos, gen_end;
    int gl;     // lane within the group
    int gbase;  // wave lane of the group's lane 0

    __device__ __forceinline__ void fill(int base) {
        const int i = base + gl;
        if (i < MT_N) {
            uint32_t v;
            if (base < gen_end) {
                v = mt[i];
            } else {
                const uint32_t a = mt[i];
                const uint32_t b = mt[i + 1 == MT_N ? 0 : i + 1];
                const uint32_t c = mt[i + MT_M >= MT_N ? i + MT_M - MT_N : i + MT_M];
                const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
                v = c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
                mt[i] = v;
            }
            win = mt_temper(v);
        }
        if (base >= gen_end) gen_end = base + G > MT_N ? MT_N : base + G;
    }

    __device__ __forceinline__ void attach(uint32_t* lds_mt, int p, int ge, int gl_, int gbase_) {
        mt = lds_mt, pos = p, gen_end = ge, gl = gl_, gbase = gbase_, win = 0;
        const int off = pos & (G - 1);
        if (off != 0) fill(pos - off);  // resume in the middle of a chunk: always below gen_end
    }

    __device__ __forceinline__ uint32_t next() {
        const int off = pos & (G - 1);
        if (off == 0) fill(pos);
        const uint32_t w = (uint32_t)__shfl((int)win, gbase + off, 64);
        pos++;
        if (pos == MT_N) pos = 0, gen_end = 0;
        return w;
    }

    // RandomState.randint(0, m + 1) / shuffle's random_interval: masked rejection on 32-bit
    // words; m == 0 consumes nothing.
    __device__ __forceinline__ int bounded(unsigned m, unsigned mask) {
        if (m == 0) return 0;
        unsigned v;
        do {
            v = next() & mask;
        } while (v > m);
        return (int)v;
    }

    // RandomState.random(): (a * 2^26 + b) / 2^53 with a = w1 >> 5, b = w2 >> 6.
    __device__ __forceinline__ double uniform() {
        const uint32_t a = next() >> 5;
        const uint32_t b = next() >> 6;
        return ((double)a * 67108864.0 + (double)b) * 1.1102230246251565e-16;  // exact: * 2^-53
    }
};

__device__ __forceinline__ unsigned mask_for(unsigned m) {
    unsigned mask = m;
    mask |= mask >> 1, mask |= mask >> 2, mask |= mask >> 4, mask |= mask >> 8, mask |= mask >> 16;
    return mask;
}

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

// accept iff u < min(1, exp(x)), x = -beta * dE  (experiments.py:238-239, 326-327).
// min(1.0, e) keeps 1.0 unless e < 1.0, so a NaN e accepts, like the reference.
__device__ __forceinline__ bool accept_test(double x, double u, bool exact_only, int& ties) {
    if (!(x < 0.0)) return true;  // e >= 1 (or NaN): probability 1, and u < 1 always
    if (!exact_only) {
        const float e32 = __expf((float)x);  // relative error < 2e-5 over the whole range
        const double lo = (double)(e32 * 0.9990234375f), hi = (double)(e32 * 1.0009765625f);
        if (u < lo) return true;
        if (u > hi) return false;
    }
    const double e = exp(x);
    if (e < 1.0) {
        const long long d = __double_as_longlong(u) - __double_as_longlong(e);
        if ((d < 0 ? -d : d) <= 4) ties++;
        return u < e;
    }
    return true;
}

// sum over the G lanes of a group, result in every lane of the group
template <int G>
__device__ __forceinline__ int group_sum(int v) {
    // rotations inside a row of 16 lanes (DPP row_ror), then across rows
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false);  // row_ror:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xf, 0xf, false);  // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x122, 0xf, 0xf, false);  // row_ror:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);  // row_ror:1
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// two distinct cells attack each other iff they share one of the 13 lines: every non-zero
// coordinate offset has the same magnitude.
__device__ __forceinline__ bool on_a_line(int di, int dj, int dk) {
    di = di < 0 ? -di : di, dj = dj < 0 ? -dj : dj, dk = dk < 0 ? -dk : dk;
    const int m = max(di, max(dj, dk));
    return (di == 0 || di == m) && (dj == 0 || dj == m) && (dk == 0 || dk == m);
}

// ------------------------------------------------------------------------------------------------
// init kernel: one wavefront per chain.  Seeds the stream (np.random.seed, experiments.py:201/288),
// builds the initial state (mcmc_board.py:26-59, mcmc.py:20-104), counts E0 and writes the chain
// record {mt[624], pos, gen_end, E0, state bytes} to the workspace.
// LDS: mt[624] | state bytes | (full_3d random only) uint16 perm[N^3]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mcq_init_kernel(KArgs a) {
    extern __shared__ uint32_t lds[];
    const long long chain = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = a.N, Q = a.Q;
    uint32_t* mt = lds;
    uint8_t* st = (uint8_t*)(lds + MT_N);
    uint16_t* perm = (uint16_t*)(st + ((a.state_bytes + 3) & ~3));

    {  // init_genrand: key[p] = s; s = 1812433253 * (s ^ (s >> 30)) + p + 1
        uint32_t s = a.seeds[chain];
        for (int p = 0; p < MT_N; p++) {
            if (lane == (p & 63)) mt[p] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)p + 1u;
        }
    }
    Rng<64> rng;
    rng.attach(mt, 0, 0, lane, 0);  // NumPy's pos == 624: the first draw starts a new generation

    const unsigned mN = (unsigned)(N - 1);
    if (a.mode == MCQ_MODE_BOARD) {
        if (a.init == MCQ_INIT_RANDOM) {  // N*N sequential draws, row-major
            for (int c = 0; c < Q; c++) {
                const int h = rng.bounded(mN, a.maskN);
                if (lane == 0) st[c] = (uint8_t)h;
            }
        } else if (a.init == MCQ_INIT_LATIN) {
            for (int c = lane; c < Q; c += 64) st[c] = (uint8_t)((c / N + c % N) % N);
        } else if (a.klarner_M == 0) {
            for (int c = lane; c < Q; c += 64) st[c] = (uint8_t)((3 * (c / N) + 5 * (c % N)) % N);
        } else {  // Klarner core M x M, the other cells drawn in row-major order
            const int M = a.klarner_M;
            for (int c = 0; c < Q; c++) {
                const int i = c / N, j = c % N;
                int h;
                if (i < M && j < M) h = (3 * i + 5 * j) % M;
                else h = rng.bounded(mN, a.maskN);
                if (lane == 0) st[c] = (uint8_t)h;
            }
        }
    } else {
        if (a.init == MCQ_INIT_LATIN) {
            for (int c = lane; c < Q; c += 64) {
                const int i = c / N, j = c % N;
                st[3 * c] = (uint8_t)i, st[3 * c + 1] = (uint8_t)j, st[3 * c + 2] = (uint8_t)((i + j) % N);
            }
        } else if (a.init == MCQ_INIT_KLARNER && a.klarner_M == 0) {
            for (int c = lane; c < Q; c += 64) {
                const int i = c / N, j = c % N;
                st[3 * c] = (uint8_t)i, st[3 * c + 1] = (uint8_t)j, st[3 * c + 2] = (uint8_t)((3 * i + 5 * j) % N);
            }
        } else if (a.init == MCQ_INIT_KLARNER) {
            // core in row-major order, then (i,j,k) triples rejected while already used (mcmc.py:63-88)
            const int M = a.klarner_M;
            for (int c = lane; c < M * M; c += 64) {
                const int i = c / M, j = c % M;
                st[3 * c] = (uint8_t)i, st[3 * c + 1] = (uint8_t)j, st[3 * c + 2] = (uint8_t)((3 * i + 5 * j) % M);
            }
            int n = M * M;
            while (n < Q) {
                const int i = rng.bounded(mN, a.maskN);
                const int j = rng.bounded(mN, a.maskN);
                const int k = rng.bounded(mN, a.maskN);
                bool used = false;
                for (int c = lane; c < n; c += 64) used |= (st[3 * c] == i && st[3 * c + 1] == j && st[3 * c + 2] == k);
                if (!__any(used)) {
                    if (lane == 0) st[3 * n] = (uint8_t)i, st[3 * n + 1] = (uint8_t)j, st[3 * n + 2] = (uint8_t)k;
                    n++;
                }
            }
        } else {
            // np.random.choice(N^3, Q, replace=False) = permutation(N^3)[:Q]: identity array, then for
            // t = n-1 .. 1 swap(arr[t], arr[bounded(t)]) (mcmc.py:97); cells decoded k fastest.
            const int n = N * N * N;
            for (int t = lane; t < n; t += 64) perm[t] = (uint16_t)t;
            for (int t = n - 1; t >= 1; t--) {
                const int s = rng.bounded((unsigned)t, mask_for((unsigned)t));
                const uint16_t at = perm[t], as = perm[s];
                if (lane == 0) perm[t] = as, perm[s] = at;
            }
            for (int c = lane; c < Q; c += 64) {
                const int f = perm[c];
                st[3 * c] = (uint8_t)(f / (N * N)), st[3 * c + 1] = (uint8_t)((f / N) % N), st[3 * c + 2] = (uint8_t)(f % N);
            }
        }
    }

    // E0 = number of unordered attacking pairs (mcmc_board.py:82-122, mcmc.py:134-169)
    int e = 0;
    for (int p = lane; p < Q * Q; p += 64) {
        const int x = p / Q, y = p % Q;
        if (x < y) {
            int xi, xj, xk, yi, yj, yk;
            if (a.mode == MCQ_MODE_BOARD) {
                xi = x / N, xj = x % N, xk = st[x], yi = y / N, yj = y % N, yk = st[y];
            } else {
                xi = st[3 * x], xj = st[3 * x + 1], xk = st[3 * x + 2], yi = st[3 * y], yj = st[3 * y + 1], yk = st[3 * y + 2];
            }
            e += on_a_line(xi - yi, xj - yj, xk - yk) ? 1 : 0;
        }
    }
    e = group_sum<64>(e);

    uint32_t* rec = a.ws + chain * (long long)a.rec_words;
    for (int w = lane; w < MT_N; w += 64) rec[w] = mt[w];
    if (lane == 0) rec[REC_POS] = (uint32_t)rng.pos, rec[REC_GEN_END] = (uint32_t)rng.gen_end, rec[REC_E0] = (uint32_t)e;
    uint8_t* rst = (uint8_t*)(rec + REC_STATE);
    for (int c = lane; c < a.state_bytes; c += 64) rst[c] = st[c];
}

// ------------------------------------------------------------------------------------------------
// sweep kernel: G lanes per chain, 64 / G chains per wavefront.
// LDS per chain: mt[624] | board: heights bytes | full_3d: queens packed (i | j<<8 | k<<16) [Q], occupancy bits
// ------------------------------------------------------------------------------------------------
constexpr int SWEEP_WAVES = 1;  // wavefronts per workgroup; chains never interact, so no barrier exists

template <int MODE, int G>
__global__ __launch_bounds__(64 * SWEEP_WAVES) void mcq_sweep_kernel(KArgs a) {
    extern __shared__ uint32_t lds[];
    constexpr int CPW = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gl = lane & (G - 1), grp = lane / G, gbase = lane - gl;
    const long long chain = ((long long)blockIdx.x * SWEEP_WAVES + wave) * CPW + grp;
    const int N = a.N, Q = a.Q;
    bool active = chain < a.n_chains;
    const long long crow = active ? chain : 0;

    uint32_t* mt = lds + (wave * CPW + grp) * a.chain_lds_words;
    uint8_t* hts = (uint8_t*)(mt + MT_N);             // board
    uint32_t* qn = mt + MT_N;                          // full_3d: packed queens
    uint32_t* occ = qn + Q;                            // full_3d: N^3 occupancy bits

    // ---- load the chain record ----
    const uint32_t* rec = a.ws + crow * (long long)a.rec_words;
    for (int w = gl; w < MT_N; w += G) mt[w] = rec[w];
    const uint8_t* rst = (const uint8_t*)(rec + REC_STATE);
    if (MODE == MCQ_MODE_BOARD) {
        for (int c = gl; c < Q; c += G) hts[c] = rst[c];
    } else {
        const int occ_words = (N * N * N + 31) >> 5;
        for (int w = gl; w < occ_words; w += G) occ[w] = 0;
        for (int c = gl; c < Q; c += G) qn[c] = (uint32_t)rst[3 * c] | ((uint32_t)rst[3 * c + 1] << 8) | ((uint32_t)rst[3 * c + 2] << 16);
        for (int c = gl; c < Q; c += G) {
            const int f = (rst[3 * c] * N + rst[3 * c + 1]) * N + rst[3 * c + 2];
            atomicOr(&occ[f >> 5], 1u << (f & 31));
        }
    }
    Rng<G> rng;
    rng.attach(mt, (int)rec[REC_POS], (int)rec[REC_GEN_END], gl, gbase);

    int E = (int)rec[REC_E0];
    int best = E, best_step = 0, n_acc = 0, no_imp = 0, ties = 0;
    long long hist_len = a.n_steps + 1, executed = a.n_steps;
    unsigned long long accw = 0;
    int hv = E;  // lane gl stages history entry (block base + gl); entry 0 = E0
    const bool exact_only = (a.flags & MCQ_FLAG_EXACT_EXP) != 0;
    const bool trace = a.out.energy_hist != nullptr;
    int32_t* hist = trace ? a.out.energy_hist + crow * a.hist_stride : nullptr;
    unsigned long long* bits = a.out.accept_bits ? (unsigned long long*)a.out.accept_bits + crow * a.bits_stride : nullptr;
    uint8_t* best_out = a.out.best_state ? a.out.best_state + crow * (long long)a.state_bytes : nullptr;
    const unsigned mN = (unsigned)(N - 1), mQ = (unsigned)(Q - 1);

    if (active) {
        if (gl == 0 && a.out.initial_energy) a.out.initial_energy[chain] = E;
        if (best_out)
            for (int c = gl; c < a.state_bytes; c += G) best_out[c] = rst[c];
    }

    double bvec = 0.0;  // lane L: beta(step0 + L) for the current block of 64 steps
    const int n_steps = (int)a.n_steps;
    for (int step = 0; step < n_steps; step++) {
        if ((step & 63) == 0) bvec = beta_at(a, (long long)step + lane);
        const int bsel = step & 63;
        const double beta = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(bvec), bsel),
                                             __builtin_amdgcn_readlane(__double2loint(bvec), bsel));
        if (active) {
            int dE;
            int cell = 0, new_k = 0;          // board move
            int qi = 0;                       // full_3d move
            uint32_t oldp = 0, newp = 0;
            if (MODE == MCQ_MODE_BOARD) {
                // experiments.py:311-321
                const int i = rng.bounded(mN, a.maskN);
                const int j = rng.bounded(mN, a.maskN);
                cell = i * N + j;
                const int old_k = hts[cell];
                do {
                    new_k = rng.bounded(mN, a.maskN);
                } while (new_k == old_k);
                // conflicts(new_k) - conflicts(old_k) over the <= 4N columns that share a line of the
                // ij-plane with (i, j): row, column, diagonal, anti-diagonal (mcmc_board.py:177-191).
                int part = 0;
                for (int p = gl; p < 4 * N; p += G) {
                    const int dir = (p >= N) + (p >= 2 * N) + (p >= 3 * N);
                    const int m = p - dir * N;
                    const int i2 = dir == 0 ? i : m;
                    const int j2 = dir == 0 ? m : dir == 1 ? j : dir == 2 ? m - i + j : i + j - m;
                    const bool ok = (unsigned)j2 < (unsigned)N && !(i2 == i && j2 == j);
                    const int d = max(abs(i2 - i), abs(j2 - j));
                    const int h = hts[ok ? i2 * N + j2 : 0];
                    const int ao = abs(h - old_k), an = abs(h - new_k);
                    const int c = (int)(an == 0 || an == d) - (int)(ao == 0 || ao == d);
                    part += ok ? c : 0;
                }
                dE = group_sum<G>(part);
            } else {
                // experiments.py:221-235
                qi = rng.bounded(mQ, a.maskQ);
                oldp = qn[qi];
                int ni, nj, nk;
                for (;;) {
                    ni = rng.bounded(mN, a.maskN);
                    nj = rng.bounded(mN, a.maskN);
                    nk = rng.bounded(mN, a.maskN);
                    const int f = (ni * N + nj) * N + nk;
                    if (!((occ[f >> 5] >> (f & 31)) & 1u)) break;
                }
                newp = (uint32_t)ni | ((uint32_t)nj << 8) | ((uint32_t)nk << 16);
                const int oi = oldp & 255, oj = (oldp >> 8) & 255, ok_ = (oldp >> 16) & 255;
                int part = 0;
                for (int c = gl; c < Q; c += G) {  // every other queen against both cells (mcmc.py:185-226)
                    const uint32_t pc = qn[c];
                    const int ci = pc & 255, cj = (pc >> 8) & 255, ck = (pc >> 16) & 255;
                    const int v = (int)on_a_line(ci - ni, cj - nj, ck - nk) - (int)on_a_line(ci - oi, cj - oj, ck - ok_);
                    part += c != qi ? v : 0;
                }
                dE = group_sum<G>(part);
            }

            const double u = rng.uniform();
            const bool acc = accept_test(-beta * (double)dE, u, exact_only, ties);
            bool improved = false;
            if (acc) {
                accw |= 1ull << (step & 63);
                if (MODE == MCQ_MODE_BOARD) {
                    if (gl == 0) hts[cell] = (uint8_t)new_k;
                } else if (gl == 0) {
                    const int fo = ((int)(oldp & 255) * N + (int)((oldp >> 8) & 255)) * N + (int)((oldp >> 16) & 255);
                    const int fn = ((int)(newp & 255) * N + (int)((newp >> 8) & 255)) * N + (int)((newp >> 16) & 255);
                    occ[fo >> 5] &= ~(1u << (fo & 31));
                    occ[fn >> 5] |= 1u << (fn & 31);
                    qn[qi] = newp;
                }
                E += dE;
                n_acc++;
                if (E < best) {
                    best = E, no_imp = 0, improved = true;
                    if (best_out) {
                        if (MODE == MCQ_MODE_BOARD) {
                            for (int c = gl; c < Q; c += G) best_out[c] = hts[c];
                        } else {
                            for (int c = gl; c < Q; c += G) {
                                const uint32_t pc = qn[c];
                                best_out[3 * c] = (uint8_t)pc, best_out[3 * c + 1] = (uint8_t)(pc >> 8), best_out[3 * c + 2] = (uint8_t)(pc >> 16);
                            }
                        }
                    }
                } else {
                    no_imp++;
                }
            } else {
                no_imp++;
            }

            if (MODE == MCQ_MODE_BOARD && a.patience >= 0 && no_imp >= a.patience) {
                // break BEFORE the append (experiments.py:349-353): entries 0..step are valid
                active = false;
                hist_len = step + 1, executed = step + 1;
                if (trace && gl <= (step & (G - 1))) hist[(step & ~(G - 1)) + gl] = hv;
                if (bits && gl == 0) bits[step >> 6] = accw;
            } else {
                const int e = step + 1;
                if ((e & (G - 1)) == gl) hv = E;
                if (improved) best_step = e;
            }
        }
        const int e = step + 1;
        if ((e & (G - 1)) == G - 1 && trace && active) hist[e - (G - 1) + gl] = hv;
        if ((step & 63) == 63) {
            if (bits && active && gl == 0) bits[step >> 6] = accw;
            accw = 0;
        }
        if (!__any(active)) break;
    }

    if (active) {  // ran to n_steps: flush the partial last block and word
        if (trace && (n_steps & (G - 1)) != G - 1 && gl <= (n_steps & (G - 1))) hist[(n_steps & ~(G - 1)) + gl] = hv;
        if (bits && gl == 0 && (n_steps & 63) != 0) bits[n_steps >> 6] = accw;
    }
    if (chain < a.n_chains) {
        if (gl == 0) {
            if (a.out.hist_len) a.out.hist_len[chain] = hist_len;
            if (a.out.steps_executed) a.out.steps_executed[chain] = executed;
            if (a.out.best_energy) a.out.best_energy[chain] = best;
            if (a.out.final_energy) a.out.final_energy[chain] = E;
            if (a.out.steps_to_best) a.out.steps_to_best[chain] = best_step;
            if (a.out.n_accepted) a.out.n_accepted[chain] = n_acc;
            if (a.out.near_ties) a.out.near_ties[chain] = ties;
        }
        if (a.out.final_state) {
            uint8_t* fo = a.out.final_state + chain * (long long)a.state_bytes;
            if (MODE == MCQ_MODE_BOARD) {
                for (int c = gl; c < Q; c += G) fo[c] = hts[c];
            } else {
                for (int c = gl; c < Q; c += G) {
                    const uint32_t pc = qn[c];
                    fo[3 * c] = (uint8_t)pc, fo[3 * c + 1] = (uint8_t)(pc >> 8), fo[3 * c + 2] = (uint8_t)(pc >> 16);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local char g_err[512];

int fail(int code, const char* fmt, const char* detail = "") {
    snprintf(g_err, sizeof g_err, fmt, detail);
    return code;
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
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
    if (p->N < MCQ_MIN_N || p->N > MCQ_MAX_N) return fail(MCQ_EINVAL, "N out of range [2, 32]");
    if (p->mode != MCQ_MODE_BOARD && p->mode != MCQ_MODE_FULL3D) return fail(MCQ_EINVAL, "unknown mcmc_type");
    if (p->init < MCQ_INIT_RANDOM || p->init > MCQ_INIT_KLARNER) return fail(MCQ_EINVAL, "Unknown init_mode");
    if (p->sched < MCQ_SCHED_CONSTANT || p->sched > MCQ_SCHED_SINUSOIDAL) return fail(MCQ_EINVAL, "Unknown betta_scheduling type");
    if (p->rng != MCQ_RNG_MT19937_NUMPY) return fail(MCQ_EINVAL, "unknown rng");
    if (p->trace != MCQ_TRACE_NONE && p->trace != MCQ_TRACE_I32) return fail(MCQ_EINVAL, "unknown trace mode");
    if (p->n_steps < 0 || p->n_steps > 2147483000LL) return fail(MCQ_EINVAL, "n_steps out of range [0, 2^31)");
    if (p->n_chains < 0) return fail(MCQ_EINVAL, "negative n_chains");
    if (p->lanes_per_chain != 0 && p->lanes_per_chain != 16 && p->lanes_per_chain != 32 && p->lanes_per_chain != 64)
        return fail(MCQ_EINVAL, "lanes_per_chain must be 0, 16, 32 or 64");
    return MCQ_OK;
}

int rec_words_for(const mcq_params* p) { return REC_STATE + (int)((mcq_state_bytes(p->N, p->mode) + 3) / 4); }

int build_args(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, void* ws, KArgs* a) {
    memset(a, 0, sizeof *a);
    a->N = p->N, a->Q = p->N * p->N, a->mode = p->mode, a->init = p->init, a->sched = p->sched, a->flags = p->flags;
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
    a->state_bytes = (int)mcq_state_bytes(p->N, p->mode);
    a->rec_words = rec_words_for(p);
    a->chain_lds_words = MT_N + (p->mode == MCQ_MODE_BOARD ? (a->Q + 3) / 4 : a->Q + (p->N * p->N * p->N + 31) / 32);
    a->beta_const = p->beta_const, a->beta_start = p->beta_start, a->beta_end = p->beta_end;
    a->n_steps = p->n_steps, a->n_chains = p->n_chains;
    a->patience = p->mode == MCQ_MODE_BOARD ? p->patience : -1;  // full_3d ignores early_stop_patience (experiments.py:199-279)
    a->hist_stride = p->hist_stride, a->bits_stride = p->bits_stride;
    a->ws = (uint32_t*)ws, a->seeds = seeds, a->out = *out;
    if (p->trace == MCQ_TRACE_NONE) a->out.energy_hist = nullptr, a->out.accept_bits = nullptr;
    return MCQ_OK;
}

template <int MODE, int G>
int launch_sweep(const KArgs& a, hipStream_t s) {
    constexpr int CPB = SWEEP_WAVES * (64 / G);
    const size_t lds = (size_t)CPB * a.chain_lds_words * 4;
    if (lds > 160 * 1024) return fail(MCQ_EINVAL, "chain state does not fit in LDS");
    HIP_TRY(hipFuncSetAttribute((const void*)mcq_sweep_kernel<MODE, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)((a.n_chains + CPB - 1) / CPB);
    hipLaunchKernelGGL((mcq_sweep_kernel<MODE, G>), dim3(grid), dim3(64 * SWEEP_WAVES), lds, s, a);
    HIP_TRY(hipGetLastError());
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

size_t mcq_state_bytes(int32_t N, int32_t mode) {
    if (N < MCQ_MIN_N || N > MCQ_MAX_N) return 0;
    return mode == MCQ_MODE_BOARD ? (size_t)N * N : (size_t)3 * N * N;
}

size_t mcq_workspace_bytes(const mcq_params* p) {
    if (validate(p) != MCQ_OK) return 0;
    return (size_t)(p->n_chains > 0 ? p->n_chains : 1) * rec_words_for(p) * 4;
}

static int run_device_impl(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, void* workspace,
                           size_t workspace_bytes, void* hip_stream, hipEvent_t* ev) {
    int rc = validate(p);
    if (rc != MCQ_OK) return rc;
    if (!seeds || !out || !workspace) return fail(MCQ_EINVAL, "null argument");
    if (workspace_bytes < mcq_workspace_bytes(p)) return fail(MCQ_ENOMEM, "workspace too small");
    if (p->trace == MCQ_TRACE_I32) {
        if (!out->energy_hist || !out->accept_bits) return fail(MCQ_EINVAL, "trace requested without buffers");
        if (p->hist_stride < p->n_steps + 1) return fail(MCQ_EINVAL, "hist_stride too small");
        if (p->bits_stride < (p->n_steps + 63) / 64) return fail(MCQ_EINVAL, "bits_stride too small");
    }
    if (p->n_chains == 0) {
        if (ev) for (int t = 0; t < 3; t++) HIP_TRY(hipEventRecord(ev[t], (hipStream_t)hip_stream));
        return MCQ_OK;
    }
    KArgs a;
    rc = build_args(p, seeds, out, workspace, &a);
    if (rc != MCQ_OK) return rc;
    hipStream_t s = (hipStream_t)hip_stream;

    if (a.out.accept_bits)  // chains that stop early leave their later words untouched
        HIP_TRY(hipMemsetAsync(a.out.accept_bits, 0, (size_t)p->n_chains * p->bits_stride * 8, s));

    size_t init_lds = (size_t)MT_N * 4 + ((a.state_bytes + 3) & ~3);
    if (p->mode == MCQ_MODE_FULL3D && p->init == MCQ_INIT_RANDOM) init_lds += (size_t)p->N * p->N * p->N * 2;
    HIP_TRY(hipFuncSetAttribute((const void*)mcq_init_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)init_lds));
    if (ev) HIP_TRY(hipEventRecord(ev[0], s));
    hipLaunchKernelGGL(mcq_init_kernel, dim3((unsigned)p->n_chains), dim3(64), init_lds, s, a);
    HIP_TRY(hipGetLastError());
    if (ev) HIP_TRY(hipEventRecord(ev[1], s));

    const int G = p->lanes_per_chain ? p->lanes_per_chain : 16;
    if (p->mode == MCQ_MODE_BOARD) {
        if (G == 16) rc = launch_sweep<MCQ_MODE_BOARD, 16>(a, s);
        else if (G == 32) rc = launch_sweep<MCQ_MODE_BOARD, 32>(a, s);
        else rc = launch_sweep<MCQ_MODE_BOARD, 64>(a, s);
    } else {
        if (G == 16) rc = launch_sweep<MCQ_MODE_FULL3D, 16>(a, s);
        else if (G == 32) rc = launch_sweep<MCQ_MODE_FULL3D, 32>(a, s);
        else rc = launch_sweep<MCQ_MODE_FULL3D, 64>(a, s);
    }
    if (rc != MCQ_OK) return rc;
    if (ev) HIP_TRY(hipEventRecord(ev[2], s));
    return MCQ_OK;
}

int mcq_run_device(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, void* workspace,
                   size_t workspace_bytes, void* hip_stream) {
    return run_device_impl(p, seeds, out, workspace, workspace_bytes, hip_stream, nullptr);
}

int mcq_run_device_timed(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, void* workspace,
                         size_t workspace_bytes, void* hip_stream, float* init_ms, float* sweep_ms) {
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    int rc = run_device_impl(p, seeds, out, workspace, workspace_bytes, hip_stream, ev);
    float a = 0.f, b = 0.f;
    if (rc == MCQ_OK && p->n_chains > 0) {
        hipError_t e = hipEventSynchronize(ev[2]);
        if (e == hipSuccess) e = hipEventElapsedTime(&a, ev[0], ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&b, ev[1], ev[2]);
        if (e != hipSuccess) rc = fail(MCQ_EDEVICE, "event timing: %s", hipGetErrorString(e));
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (init_ms) *init_ms = a;
    if (sweep_ms) *sweep_ms = b;
    return rc;
}

int mcq_run_host(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, double* kernel_seconds) {
    int rc = validate(p);
    if (rc != MCQ_OK) return rc;
    if (!seeds || !out) return fail(MCQ_EINVAL, "null argument");
    if (p->device >= 0) HIP_TRY(hipSetDevice(p->device));
    if (kernel_seconds) *kernel_seconds = 0.0;
    if (p->n_chains == 0) return MCQ_OK;

    const size_t n = (size_t)p->n_chains, sb = mcq_state_bytes(p->N, p->mode);
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
        {(void**)&d.best_state, out->best_state, n * sb},
        {(void**)&d.final_state, out->final_state, n * sb},
    };
    uint32_t* d_seeds = nullptr;
    void* d_ws = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const size_t ws_bytes = mcq_workspace_bytes(p);
    rc = MCQ_OK;
    auto cleanup = [&]() {
        for (auto& b : bufs)
            if (*b.dev) (void)hipFree(*b.dev);
        if (d_seeds) (void)hipFree(d_seeds);
        if (d_ws) (void)hipFree(d_ws);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
#define HOST_TRY(expr)                                                                \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            cleanup();                                                                \
            return fail(e_ == hipErrorOutOfMemory ? MCQ_ENOMEM : MCQ_EDEVICE, #expr ": %s", hipGetErrorString(e_)); \
        }                                                                             \
    } while (0)
    for (auto& b : bufs)
        if (b.host) HOST_TRY(hipMalloc(b.dev, b.bytes));
    HOST_TRY(hipMalloc((void**)&d_seeds, n * 4));
    HOST_TRY(hipMalloc(&d_ws, ws_bytes));
    HOST_TRY(hipMemcpy(d_seeds, seeds, n * 4, hipMemcpyHostToDevice));
    HOST_TRY(hipEventCreate(&e0));
    HOST_TRY(hipEventCreate(&e1));
    HOST_TRY(hipEventRecord(e0, nullptr));
    rc = mcq_run_device(p, d_seeds, &d, d_ws, ws_bytes, nullptr);
    if (rc != MCQ_OK) {
        cleanup();
        return rc;
    }
    HOST_TRY(hipEventRecord(e1, nullptr));
    HOST_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HOST_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (kernel_seconds) *kernel_seconds = ms * 1e-3;
    for (auto& b : bufs)
        if (b.host) HOST_TRY(hipMemcpy(b.host, *b.dev, b.bytes, hipMemcpyDeviceToHost));
    cleanup();
    return MCQ_OK;
#undef HOST_TRY
}

}  // extern "C"
