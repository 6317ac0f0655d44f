/*
 * mcq_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the reference's per-chain Metropolis sweep
 * (galgantar/monte-carlo-collective: experiments.py, mcmc.py, mcmc_board.py).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * shipped path (libmcq_hip.so) never links or calls it.
 *
 * It follows the reference's OWN algorithm (O(N^2) scan of every queen with the
 * reference's six / seven attack predicates, O(Q^2) pairwise initial energy), not the
 * GPU kernels' neighbour-scan formulation, so that the two are independent derivations.
 *
 * Parity is PINNED: tests/golden/ holds vectors produced by importing the reference
 * itself (tools/gen_golden.py) and tests/test_oracle_golden.py checks this file against
 * all of them.  The random stream is NumPy's legacy global RandomState (MT19937), a
 * third-party dependency of the reference (requirements.txt:1, unpinned; NumPy 2.2.6 in
 * the build container); its published algorithm is restated below and checked against
 * NumPy itself in tests/test_oracle_rng.py.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mcq.h"

static __thread char g_err[256];

const char* mcq_oracle_last_error(void) { return g_err; }

static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

/* ------------------------------------------------------------------------------------------
 * NumPy legacy RandomState (MT19937).  Call sites in the reference: experiments.py:201,
 * 221, 227-229, 239, 288, 311-312, 317-319, 327; mcmc_board.py:28, 57; mcmc.py:82-84, 97.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t key[624];
    int pos;
} mt_t;

/* np.random.seed(int): Knuth's linear recurrence, pos = 624 so the first draw twists. */
static void mt_seed(mt_t* s, uint32_t seed) {
    for (int p = 0; p < 624; p++) {
        s->key[p] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)p + 1u;
    }
    s->pos = 624;
}

static void mt_twist(mt_t* s) {
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAT = 0x9908b0dfu;
    uint32_t* k = s->key;
    int i;
    for (i = 0; i < 624 - 397; i++) {
        uint32_t y = (k[i] & UP) | (k[i + 1] & LO);
        k[i] = k[i + 397] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
    }
    for (; i < 623; i++) {
        uint32_t y = (k[i] & UP) | (k[i + 1] & LO);
        k[i] = k[i + 397 - 624] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
    }
    uint32_t y = (k[623] & UP) | (k[0] & LO);
    k[623] = k[396] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
    s->pos = 0;
}

static uint32_t mt_u32(mt_t* s) {
    if (s->pos == 624) mt_twist(s);
    uint32_t y = s->key[s->pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* ------------------------------------------------------------------------------------------
 * Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
 * library's philox4x32_R(10, ...)): the counter-based generator of mcq_params.rng == MCQ_RNG_PHILOX4X32_10.  NOT a
 * stream of the reference (which only has NumPy's MT19937): it exists to run the same sweep without a generator state
 * in memory; parity in this mode is HIP == this file.  Known answers of the block function: tests/test_oracle_rng.py.
 * Stream convention: word w of a chain = philox(counter = (w / 4 low, w / 4 high, 0, 0), key = (seed, 0))[w % 4].
 * ---------------------------------------------------------------------------------------- */
static void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0, c1 = n1, c2 = n2, c3 = n3;
        k0 += 0x9E3779B9u, k1 += 0xBB67AE85u;
    }
    out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

int mcq_oracle_philox_block(const uint32_t* ctr, const uint32_t* key, uint32_t* out) {
    philox4x32_10(ctr, key, out);
    return 0;
}

/* One chain's random stream: NumPy's MT19937 (the reference's) or the Philox word stream. */
typedef struct {
    int kind;     /* MCQ_RNG_* */
    mt_t mt;
    uint32_t key; /* philox: the chain's seed */
    uint64_t w;   /* philox: index of the next word */
    uint32_t buf[4];
    uint64_t words; /* words drawn so far (mcq_outputs.stream_words) */
} rng_t;

static void rng_seed(rng_t* s, int kind, uint32_t seed) {
    s->kind = kind;
    s->words = 0;
    if (kind == MCQ_RNG_PHILOX4X32_10) s->key = seed, s->w = 0;
    else mt_seed(&s->mt, seed);
}

/* the stream chain r runs on: seeded (np.random.seed(seed), experiments.py:200-201, 287-288) or, with mcq_params.stream_states, the caller's
 * MT19937 state continued where it stands (seed=None: the reference skips the seeding and draws from NumPy's global RandomState) */
static void rng_start(rng_t* s, const mcq_params* p, const uint32_t* seeds, int64_t r) {
    rng_seed(s, p->rng, seeds[r]);
    if (p->stream_states) {
        const uint32_t* st = p->stream_states + (size_t)r * 625;
        memcpy(s->mt.key, st, 624 * sizeof(uint32_t));
        s->mt.pos = (int)st[624];
    }
}

static uint32_t rng_u32(rng_t* s) {
    s->words++;
    if (s->kind != MCQ_RNG_PHILOX4X32_10) return mt_u32(&s->mt);
    if ((s->w & 3) == 0) {
        const uint64_t blk = s->w >> 2;
        const uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u}, key[2] = {s->key, 0u};
        philox4x32_10(ctr, key, s->buf);
    }
    return s->buf[s->w++ & 3];
}

/* Masked rejection on 32-bit words, as RandomState.randint(0, m+1) and shuffle use for
 * ranges below 2^32: m == 0 consumes nothing. */
static uint32_t mt_bounded(rng_t* s, uint32_t m) {
    if (m == 0) return 0;
    uint32_t mask = m;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
        v = rng_u32(s) & mask;
    } while (v > m);
    return v;
}

/* np.random.random(): 53-bit double from two words, high part first. */
static double mt_double(rng_t* s) {
    uint32_t a = rng_u32(s) >> 5, b = rng_u32(s) >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

/* exported for tests/test_oracle_rng.py: kind 0 = raw u32, 1 = bounded(arg), 2 = double */
int mcq_oracle_rng_stream(uint32_t seed, int kind, uint32_t arg, int64_t n, uint32_t* out_u32,
                          double* out_f64) {
    rng_t s; /* kind + 16 selects the Philox word stream */
    rng_seed(&s, kind >= 16 ? MCQ_RNG_PHILOX4X32_10 : MCQ_RNG_MT19937_NUMPY, seed);
    kind &= 15;
    for (int64_t t = 0; t < n; t++) {
        if (kind == 0) out_u32[t] = rng_u32(&s);
        else if (kind == 1) out_u32[t] = mt_bounded(&s, arg);
        else out_f64[t] = mt_double(&s);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * beta schedules, experiments.py:13-77.  Each expression keeps the reference's evaluation
 * order; the file is compiled with -ffp-contract=off so no product-sum is fused.
 * ---------------------------------------------------------------------------------------- */
static double beta_at(const mcq_params* p, int64_t step) {
    if (p->beta_table) return p->beta_table[step]; /* the reference's own values (abi.beta_values: NumPy, like experiments.py:13-77) */
    const double bs = p->beta_start, be = p->beta_end;
    const int64_t n = p->n_steps;
    switch (p->sched) {
    case MCQ_SCHED_CONSTANT: /* experiments.py:13-16 */
        return p->beta_const;
    case MCQ_SCHED_LINEAR: { /* experiments.py:19-25 */
        if (n <= 1) return be;
        double frac = (double)step / (double)(n - 1);
        return bs + frac * (be - bs);
    }
    case MCQ_SCHED_EXPONENTIAL: { /* experiments.py:27-40 */
        if (n <= 1) return be;
        double log_ratio = log(be / bs);
        int64_t c = step < 0 ? 0 : (step > n - 1 ? n - 1 : step);
        double t = (double)c / (double)(n - 1);
        return bs * exp(log_ratio * t);
    }
    case MCQ_SCHED_LOGARITHMIC: { /* experiments.py:42-58 */
        if (n <= 1) return be;
        double log_norm = log((double)(1 + n));
        int64_t c = step < 0 ? 0 : (step > n ? n : step);
        return bs + (be - bs) * (log((double)(1 + c)) / log_norm);
    }
    default: { /* sinusoidal, experiments.py:60-77 */
        if (n <= 1) return be;
        int64_t c = step < 0 ? 0 : (step > n ? n : step);
        double x = 3.141592653589793 * (double)c / (double)n;
        return bs + (be - bs) * (1.0 - cos(x)) / 2.0;
    }
    }
}

int mcq_oracle_beta_table(const mcq_params* p, const int64_t* steps, int64_t n, double* out) {
    mcq_params q = *p;
    q.beta_table = NULL; /* always the oracle's own libm evaluation */
    p = &q;
    for (int64_t t = 0; t < n; t++) out[t] = beta_at(p, steps[t]);
    return 0;
}

static int gcd_int(int a, int b) {
    while (b) {
        int t = a % b;
        a = b;
        b = t;
    }
    return a;
}

/* largest m < N with gcd(m, 210) == 1 (mcmc_board.py:38-42, mcmc.py:46-50); 0 if none */
static int klarner_core(int N) {
    for (int m = N - 1; m > 0; m--)
        if (gcd_int(m, 210) == 1) return m;
    return 0;
}

static int iabs(int v) { return v < 0 ? -v : v; }

/* ------------------------------------------------------------------------------------------
 * Board chain: State3DQueensBoard (mcmc_board.py) + metropolis_mcmc_board (experiments.py:282-376)
 * ---------------------------------------------------------------------------------------- */
static int board_init(int N, int init, rng_t* rng, int* h) {
    if (init == MCQ_INIT_RANDOM) { /* mcmc_board.py:28: N*N bounded draws, row-major */
        for (int c = 0; c < N * N; c++) h[c] = (int)mt_bounded(rng, (uint32_t)(N - 1));
    } else if (init == MCQ_INIT_LATIN) { /* mcmc_board.py:30-31 */
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) h[i * N + j] = (i + j) % N;
    } else if (init == MCQ_INIT_KLARNER) { /* mcmc_board.py:33-57 */
        if (gcd_int(N, 210) == 1) {
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++) h[i * N + j] = (3 * i + 5 * j) % N;
        } else {
            int M = klarner_core(N);
            if (M == 0) return -1;
            memset(h, 0, sizeof(int) * (size_t)(N * N));
            for (int i = 0; i < M; i++)
                for (int j = 0; j < M; j++) h[i * N + j] = (3 * i + 5 * j) % M;
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++)
                    if (!(i < M && j < M)) h[i * N + j] = (int)mt_bounded(rng, (uint32_t)(N - 1));
        }
    } else {
        return -1;
    }
    return 0;
}

/* the six predicates of mcmc_board.py:103-119 / 177-191 */
static int board_attacks(int i, int j, int k, int i2, int j2, int k2) {
    int di = iabs(i2 - i), dj = iabs(j2 - j), dk = iabs(k2 - k);
    int same_ik = (i2 == i) && (k2 == k);
    int same_jk = (j2 == j) && (k2 == k);
    int plane_k = (k2 == k) && (di == dj);
    int plane_j = (j2 == j) && (di == dk);
    int plane_i = (i2 == i) && (dj == dk);
    int space = (di == dj) && (dj == dk);
    return same_ik | same_jk | plane_k | plane_j | plane_i | space;
}

/* mcmc_board.py:82-122 */
static int board_energy(int N, const int* h) {
    if (N < 2) return 0;
    int Q = N * N, e = 0;
    for (int a = 0; a < Q; a++)
        for (int b = a + 1; b < Q; b++)
            e += board_attacks(a / N, a % N, h[a], b / N, b % N, h[b]);
    return e;
}

/* mcmc_board.py:147-193: every column except (i, j) itself */
static int board_conflicts(int N, const int* h, int i, int j, int k) {
    int c = 0;
    for (int i2 = 0; i2 < N; i2++)
        for (int j2 = 0; j2 < N; j2++) {
            if (i2 == i && j2 == j) continue;
            c += board_attacks(i, j, k, i2, j2, h[i2 * N + j2]);
        }
    return c;
}

typedef struct {
    const mcq_params* p;
    const uint32_t* seeds;
    const mcq_outputs* out;
    int fast; /* 0: the reference's O(N^2) scan; 1: line counters (mcq_oracle_run_fast) */
} job_t;

static int64_t ulp_distance(double a, double b) {
    int64_t x, y;
    memcpy(&x, &a, 8);
    memcpy(&y, &b, 8);
    return x > y ? x - y : y - x; /* both are non-negative finite here */
}

/* experiments.py:238-239 / 326-327: the uniform is drawn on every step */
static int accept_move(rng_t* rng, double beta, int dE, int64_t* near_ties) {
    double e = exp(-beta * (double)dE);
    double prob = e < 1.0 ? e : 1.0; /* min(1.0, e); NaN cannot occur for finite beta */
    double u = mt_double(rng);
    if (near_ties && prob < 1.0 && ulp_distance(u, prob) <= 4) (*near_ties)++;
    return u < prob;
}

static int run_board_chain(const job_t* jb, int64_t r) {
    const mcq_params* p = jb->p;
    const mcq_outputs* o = jb->out;
    const int N = p->N, Q = N * N;
    int* h = (int*)malloc(sizeof(int) * (size_t)Q * 2);
    int* best_h = h + Q;
    if (!h) return MCQ_ENOMEM;

    rng_t rng;
    rng_start(&rng, p, jb->seeds, r); /* experiments.py:287-288 */
    if (board_init(N, p->init, &rng, h) != 0) {
        free(h);
        return MCQ_EINVAL;
    }
    int E = board_energy(N, h); /* experiments.py:291 */
    memcpy(best_h, h, sizeof(int) * (size_t)Q);
    int best = E;
    int64_t best_step = 0, accepted = 0, no_improve = 0, ties = 0, len = 1, executed = 0;

    int32_t* hist = o->energy_hist ? o->energy_hist + r * p->hist_stride : NULL;
    uint64_t* bits = o->accept_bits ? o->accept_bits + r * p->bits_stride : NULL;
    if (hist) hist[0] = E;
    if (bits)
        for (int64_t w = 0; w < p->bits_stride; w++) bits[w] = 0;
    if (o->initial_energy) o->initial_energy[r] = E;

    for (int64_t step = 0; step < p->n_steps; step++) { /* experiments.py:308-358 */
        double beta = beta_at(p, step);
        int i = (int)mt_bounded(&rng, (uint32_t)(N - 1));
        int j = (int)mt_bounded(&rng, (uint32_t)(N - 1));
        int old_k = h[i * N + j];
        int old_c = board_conflicts(N, h, i, j, old_k);
        int new_k = (int)mt_bounded(&rng, (uint32_t)(N - 1));
        while (new_k == old_k) new_k = (int)mt_bounded(&rng, (uint32_t)(N - 1));
        int new_c = board_conflicts(N, h, i, j, new_k);
        int dE = new_c - old_c;
        int acc = accept_move(&rng, beta, dE, &ties);
        executed++;
        int improved = 0;
        if (acc) {
            if (bits) bits[step >> 6] |= 1ull << (step & 63);
            h[i * N + j] = new_k;
            E += dE;
            accepted++;
            if (E < best) {
                best = E;
                memcpy(best_h, h, sizeof(int) * (size_t)Q);
                no_improve = 0;
                improved = 1;
            } else {
                no_improve++;
            }
        } else {
            no_improve++;
        }
        if (p->patience >= 0 && no_improve >= p->patience) break; /* before the append: experiments.py:349-353 */
        if (hist) hist[len] = E;
        if (improved) best_step = len;
        len++;
    }
    /* steps_to_best = first argmin of energy_history (experiments.py:364-365).  best only
     * decreases strictly, so the first index holding the minimum is the entry appended by
     * the last strict improvement (0 if none was appended). */
    if (o->hist_len) o->hist_len[r] = len;
    if (o->steps_executed) o->steps_executed[r] = executed;
    if (o->best_energy) o->best_energy[r] = best;
    if (o->final_energy) o->final_energy[r] = E;
    if (o->steps_to_best) o->steps_to_best[r] = best_step;
    if (o->n_accepted) o->n_accepted[r] = accepted;
    if (o->stream_words) o->stream_words[r] = p->rng == MCQ_RNG_MT19937_NUMPY ? (uint32_t)rng.words : 0u;
    if (o->near_ties) o->near_ties[r] = ties;
    if (o->best_state)
        for (int c = 0; c < Q; c++) o->best_state[r * Q + c] = (uint8_t)best_h[c];
    if (o->final_state)
        for (int c = 0; c < Q; c++) o->final_state[r * Q + c] = (uint8_t)h[c];
    free(h);
    return MCQ_OK;
}

/* ------------------------------------------------------------------------------------------
 * Full-3D chain: State3DQueens (mcmc.py) + metropolis_mcmc (experiments.py:199-279)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int i, j, k;
} cell_t;

/* the seven predicates of mcmc.py:148-166 / 205-224 */
static int full_attacks(cell_t a, cell_t b) {
    int di = iabs(b.i - a.i), dj = iabs(b.j - a.j), dk = iabs(b.k - a.k);
    int same_ij = (b.i == a.i) && (b.j == a.j);
    int same_ik = (b.i == a.i) && (b.k == a.k);
    int same_jk = (b.j == a.j) && (b.k == a.k);
    int plane_k = (b.k == a.k) && (di == dj);
    int plane_j = (b.j == a.j) && (di == dk);
    int plane_i = (b.i == a.i) && (dj == dk);
    int space = (di == dj) && (dj == dk);
    return same_ij | same_ik | same_jk | plane_k | plane_j | plane_i | space;
}

/* queens of a full_3d chain: State3DQueens(N, Q=...) (mcmc.py:6-18); Q = N^2 unless the caller names a count */
static int queens_of(const mcq_params* p) { return p->mode == MCQ_MODE_FULL3D && p->n_queens > 0 ? p->n_queens : p->N * p->N; }

static int full_init(int N, int Q, int init, rng_t* rng, cell_t* q, uint8_t* occ) {
    if (init != MCQ_INIT_RANDOM && Q != N * N) return -1; /* mcmc.py:21-25: latin / klarner assume Q = N^2 */
    memset(occ, 0, (size_t)N * N * N);
    if (init == MCQ_INIT_LATIN) { /* mcmc.py:28-34 */
        for (int i = 0; i < N; i++)
            for (int j = 0; j < N; j++) q[i * N + j] = (cell_t){i, j, (i + j) % N};
    } else if (init == MCQ_INIT_KLARNER) { /* mcmc.py:36-90 */
        if (gcd_int(N, 210) == 1) {
            for (int i = 0; i < N; i++)
                for (int j = 0; j < N; j++) q[i * N + j] = (cell_t){i, j, (3 * i + 5 * j) % N};
        } else {
            int M = klarner_core(N);
            if (M == 0) return -1;
            int n = 0;
            for (int i = 0; i < M; i++)
                for (int j = 0; j < M; j++) {
                    q[n] = (cell_t){i, j, (3 * i + 5 * j) % M};
                    occ[(q[n].i * N + q[n].j) * N + q[n].k] = 1; /* `used` set, mcmc.py:70 */
                    n++;
                }
            while (n < Q) { /* mcmc.py:81-88 */
                int i = (int)mt_bounded(rng, (uint32_t)(N - 1));
                int j = (int)mt_bounded(rng, (uint32_t)(N - 1));
                int k = (int)mt_bounded(rng, (uint32_t)(N - 1));
                if (!occ[(i * N + j) * N + k]) {
                    occ[(i * N + j) * N + k] = 1;
                    q[n++] = (cell_t){i, j, k};
                }
            }
            memset(occ, 0, (size_t)N * N * N);
        }
    } else if (init == MCQ_INIT_RANDOM) { /* mcmc.py:92-101 */
        /* np.random.choice(N^3, size=Q, replace=False) == permutation(N^3)[:Q]; the legacy
         * shuffle walks t = n-1 .. 1 and swaps with bounded(t). */
        int n = N * N * N;
        int* arr = (int*)malloc(sizeof(int) * (size_t)n);
        if (!arr) return -1;
        for (int t = 0; t < n; t++) arr[t] = t;
        for (int t = n - 1; t >= 1; t--) {
            int s = (int)mt_bounded(rng, (uint32_t)t);
            int tmp = arr[t];
            arr[t] = arr[s];
            arr[s] = tmp;
        }
        for (int c = 0; c < Q; c++) {
            int f = arr[c];
            q[c] = (cell_t){f / (N * N), (f / N) % N, f % N};
        }
        free(arr);
    } else {
        return -1;
    }
    for (int c = 0; c < Q; c++) { /* occ_set, mcmc.py:113-118 */
        size_t f = ((size_t)q[c].i * N + q[c].j) * N + q[c].k;
        if (occ[f]) return -2;
        occ[f] = 1;
    }
    return 0;
}

/* mcmc.py:134-169 */
static int full_energy(int Q, const cell_t* q) {
    int e = 0;
    if (Q < 2) return 0;
    for (int a = 0; a < Q; a++)
        for (int b = a + 1; b < Q; b++) e += full_attacks(q[a], q[b]);
    return e;
}

/* mcmc.py:185-226: queen q_idx removed by index, every other queen tested against `at` */
static int full_conflicts(int Q, const cell_t* q, int q_idx, cell_t at) {
    int c = 0;
    for (int b = 0; b < Q; b++)
        if (b != q_idx) c += full_attacks(at, q[b]);
    return c;
}

static int run_full_chain(const job_t* jb, int64_t r) {
    const mcq_params* p = jb->p;
    const mcq_outputs* o = jb->out;
    const int N = p->N, Q = queens_of(p);
    cell_t* q = (cell_t*)malloc(sizeof(cell_t) * (size_t)Q * 2);
    uint8_t* occ = (uint8_t*)malloc((size_t)N * N * N);
    if (!q || !occ) {
        free(q);
        free(occ);
        return MCQ_ENOMEM;
    }
    cell_t* best_q = q + Q;

    rng_t rng;
    rng_start(&rng, p, jb->seeds, r); /* experiments.py:200-201 */
    if (full_init(N, Q, p->init, &rng, q, occ) != 0) {
        free(q);
        free(occ);
        return MCQ_EINVAL;
    }
    int E = full_energy(Q, q); /* experiments.py:204 */
    memcpy(best_q, q, sizeof(cell_t) * (size_t)Q);
    int best = E;
    int64_t best_step = 0, accepted = 0, ties = 0, len = 1;

    int32_t* hist = o->energy_hist ? o->energy_hist + r * p->hist_stride : NULL;
    uint64_t* bits = o->accept_bits ? o->accept_bits + r * p->bits_stride : NULL;
    if (hist) hist[0] = E;
    if (bits)
        for (int64_t w = 0; w < p->bits_stride; w++) bits[w] = 0;
    if (o->initial_energy) o->initial_energy[r] = E;

    for (int64_t step = 0; step < p->n_steps; step++) { /* experiments.py:218-258; early stop is ignored here */
        double beta = beta_at(p, step);
        int qi = (int)mt_bounded(&rng, (uint32_t)(Q - 1));
        int old_c = full_conflicts(Q, q, qi, q[qi]);
        cell_t nw;
        for (;;) { /* experiments.py:226-231 */
            nw.i = (int)mt_bounded(&rng, (uint32_t)(N - 1));
            nw.j = (int)mt_bounded(&rng, (uint32_t)(N - 1));
            nw.k = (int)mt_bounded(&rng, (uint32_t)(N - 1));
            if (!occ[(nw.i * N + nw.j) * N + nw.k]) break;
        }
        int new_c = full_conflicts(Q, q, qi, nw);
        int dE = new_c - old_c;
        int acc = accept_move(&rng, beta, dE, &ties);
        if (acc) {
            if (bits) bits[step >> 6] |= 1ull << (step & 63);
            occ[(q[qi].i * N + q[qi].j) * N + q[qi].k] = 0; /* mcmc.py:178-181 */
            occ[(nw.i * N + nw.j) * N + nw.k] = 1;
            q[qi] = nw;
            E += dE;
            accepted++;
            if (E < best) {
                best = E;
                memcpy(best_q, q, sizeof(cell_t) * (size_t)Q);
                best_step = len;
            }
        }
        if (hist) hist[len] = E;
        len++;
    }
    if (o->hist_len) o->hist_len[r] = len;
    if (o->steps_executed) o->steps_executed[r] = p->n_steps;
    if (o->best_energy) o->best_energy[r] = best;
    if (o->final_energy) o->final_energy[r] = E;
    if (o->steps_to_best) o->steps_to_best[r] = best_step;
    if (o->n_accepted) o->n_accepted[r] = accepted;
    if (o->stream_words) o->stream_words[r] = p->rng == MCQ_RNG_MT19937_NUMPY ? (uint32_t)rng.words : 0u;
    if (o->near_ties) o->near_ties[r] = ties;
    for (int c = 0; c < Q; c++) {
        if (o->best_state) {
            uint8_t* d = o->best_state + ((size_t)r * Q + c) * 3;
            d[0] = (uint8_t)best_q[c].i, d[1] = (uint8_t)best_q[c].j, d[2] = (uint8_t)best_q[c].k;
        }
        if (o->final_state) {
            uint8_t* d = o->final_state + ((size_t)r * Q + c) * 3;
            d[0] = (uint8_t)q[c].i, d[1] = (uint8_t)q[c].j, d[2] = (uint8_t)q[c].k;
        }
    }
    free(q);
    free(occ);
    return MCQ_OK;
}

/* ------------------------------------------------------------------------------------------
 * "Best CPU" variant (SURVEY 8d's second baseline; bench.py's cpu_baseline.fast_port): the same chains with O(1) dE from
 * per-line occupancy counters instead of the reference's O(N^2) scan.  Two distinct cells attack iff they share one of the
 * 13 lines through a cell, and share at most one, so conflicts(cell) = sum over the 13 families of the queens on the cell's
 * line of that family, minus the moving queen where it is counted.  Families (D = 2N - 1):
 *   0 (j,k)  1 (i,k)  2 (i,j)                                                    axes          N^2 lines each
 *   3 (k,i-j) 4 (k,i+j)  5 (j,i-k) 6 (j,i+k)  7 (i,j-k) 8 (i,j+k)               plane diagonals  N D each
 *   9 (i-j,i-k) 10 (i-j,i+k) 11 (i+j,i-k) 12 (i+j,i+k)                          space diagonals  D^2 each
 * Results must equal mcq_oracle_run's bit for bit (tests/test_oracle_fast.py): same stream, same draws, same floats.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int N, D;
    int base[13];
    uint8_t* cnt;
} lines_t;

static size_t lines_total(int N) {
    const size_t D = 2 * (size_t)N - 1, Q = (size_t)N * N;
    return 3 * Q + 6 * N * D + 4 * D * D;
}

static void lines_setup(lines_t* L, int N, uint8_t* cnt) {
    const int D = 2 * N - 1, Q = N * N;
    L->N = N, L->D = D, L->cnt = cnt;
    int b = 0;
    for (int f = 0; f < 13; f++) {
        L->base[f] = b;
        b += f < 3 ? Q : (f < 9 ? N * D : D * D);
    }
    memset(cnt, 0, lines_total(N));
}

static inline void lines_of(const lines_t* L, int i, int j, int k, int idx[13]) {
    const int N = L->N, D = L->D, o = N - 1;
    idx[0] = L->base[0] + j * N + k;
    idx[1] = L->base[1] + i * N + k;
    idx[2] = L->base[2] + i * N + j;
    idx[3] = L->base[3] + k * D + (i - j + o);
    idx[4] = L->base[4] + k * D + (i + j);
    idx[5] = L->base[5] + j * D + (i - k + o);
    idx[6] = L->base[6] + j * D + (i + k);
    idx[7] = L->base[7] + i * D + (j - k + o);
    idx[8] = L->base[8] + i * D + (j + k);
    idx[9] = L->base[9] + (i - j + o) * D + (i - k + o);
    idx[10] = L->base[10] + (i - j + o) * D + (i + k);
    idx[11] = L->base[11] + (i + j) * D + (i - k + o);
    idx[12] = L->base[12] + (i + j) * D + (i + k);
}

static inline void lines_add(lines_t* L, int i, int j, int k, int d) {
    int idx[13];
    lines_of(L, i, j, k, idx);
    for (int f = 0; f < 13; f++) L->cnt[idx[f]] = (uint8_t)(L->cnt[idx[f]] + d);
}

static inline int lines_sum(const lines_t* L, int i, int j, int k) {
    int idx[13], c = 0;
    lines_of(L, i, j, k, idx);
    for (int f = 0; f < 13; f++) c += L->cnt[idx[f]];
    return c;
}

static int lines_energy(const lines_t* L) {
    int e = 0;
    const size_t n = lines_total(L->N);
    for (size_t t = 0; t < n; t++) e += L->cnt[t] * (L->cnt[t] - 1) / 2;
    return e;
}

static int run_board_chain_fast(const job_t* jb, int64_t r) {
    const mcq_params* p = jb->p;
    const mcq_outputs* o = jb->out;
    const int N = p->N, Q = N * N;
    int* h = (int*)malloc(sizeof(int) * (size_t)Q * 2);
    uint8_t* cnt = (uint8_t*)malloc(lines_total(N));
    if (!h || !cnt) {
        free(h), free(cnt);
        return MCQ_ENOMEM;
    }
    int* best_h = h + Q;
    rng_t rng;
    rng_start(&rng, p, jb->seeds, r);
    if (board_init(N, p->init, &rng, h) != 0) {
        free(h), free(cnt);
        return MCQ_EINVAL;
    }
    lines_t L;
    lines_setup(&L, N, cnt);
    for (int c = 0; c < Q; c++) lines_add(&L, c / N, c % N, h[c], 1);
    int E = lines_energy(&L);
    memcpy(best_h, h, sizeof(int) * (size_t)Q);
    int best = E;
    int64_t best_step = 0, accepted = 0, no_improve = 0, ties = 0, len = 1, executed = 0;
    int32_t* hist = o->energy_hist ? o->energy_hist + r * p->hist_stride : NULL;
    uint64_t* bits = o->accept_bits ? o->accept_bits + r * p->bits_stride : NULL;
    if (hist) hist[0] = E;
    if (bits)
        for (int64_t w = 0; w < p->bits_stride; w++) bits[w] = 0;
    if (o->initial_energy) o->initial_energy[r] = E;
    for (int64_t step = 0; step < p->n_steps; step++) {
        double beta = beta_at(p, step);
        int i = (int)mt_bounded(&rng, (uint32_t)(N - 1));
        int j = (int)mt_bounded(&rng, (uint32_t)(N - 1));
        int old_k = h[i * N + j];
        int new_k = (int)mt_bounded(&rng, (uint32_t)(N - 1));
        while (new_k == old_k) new_k = (int)mt_bounded(&rng, (uint32_t)(N - 1));
        /* the moving queen sits on every line through the old cell (-13) and, of the lines through the new cell, only on
         * the column (i, j), which holds nobody else: both cells count it once there, so that family cancels */
        int dE = lines_sum(&L, i, j, new_k) - lines_sum(&L, i, j, old_k) + 12;
        int acc = accept_move(&rng, beta, dE, &ties);
        executed++;
        int improved = 0;
        if (acc) {
            if (bits) bits[step >> 6] |= 1ull << (step & 63);
            lines_add(&L, i, j, old_k, -1);
            lines_add(&L, i, j, new_k, 1);
            h[i * N + j] = new_k;
            E += dE;
            accepted++;
            if (E < best) {
                best = E;
                memcpy(best_h, h, sizeof(int) * (size_t)Q);
                no_improve = 0;
                improved = 1;
            } else {
                no_improve++;
            }
        } else {
            no_improve++;
        }
        if (p->patience >= 0 && no_improve >= p->patience) break;
        if (hist) hist[len] = E;
        if (improved) best_step = len;
        len++;
    }
    if (o->hist_len) o->hist_len[r] = len;
    if (o->steps_executed) o->steps_executed[r] = executed;
    if (o->best_energy) o->best_energy[r] = best;
    if (o->final_energy) o->final_energy[r] = E;
    if (o->steps_to_best) o->steps_to_best[r] = best_step;
    if (o->n_accepted) o->n_accepted[r] = accepted;
    if (o->stream_words) o->stream_words[r] = p->rng == MCQ_RNG_MT19937_NUMPY ? (uint32_t)rng.words : 0u;
    if (o->near_ties) o->near_ties[r] = ties;
    if (o->best_state)
        for (int c = 0; c < Q; c++) o->best_state[r * Q + c] = (uint8_t)best_h[c];
    if (o->final_state)
        for (int c = 0; c < Q; c++) o->final_state[r * Q + c] = (uint8_t)h[c];
    free(h), free(cnt);
    return MCQ_OK;
}

static int run_full_chain_fast(const job_t* jb, int64_t r) {
    const mcq_params* p = jb->p;
    const mcq_outputs* o = jb->out;
    const int N = p->N, Q = queens_of(p);
    cell_t* q = (cell_t*)malloc(sizeof(cell_t) * (size_t)Q * 2);
    uint8_t* occ = (uint8_t*)malloc((size_t)N * N * N);
    uint8_t* cnt = (uint8_t*)malloc(lines_total(N));
    if (!q || !occ || !cnt) {
        free(q), free(occ), free(cnt);
        return MCQ_ENOMEM;
    }
    cell_t* best_q = q + Q;
    rng_t rng;
    rng_start(&rng, p, jb->seeds, r);
    if (full_init(N, Q, p->init, &rng, q, occ) != 0) {
        free(q), free(occ), free(cnt);
        return MCQ_EINVAL;
    }
    lines_t L;
    lines_setup(&L, N, cnt);
    for (int c = 0; c < Q; c++) lines_add(&L, q[c].i, q[c].j, q[c].k, 1);
    int E = lines_energy(&L);
    memcpy(best_q, q, sizeof(cell_t) * (size_t)Q);
    int best = E;
    int64_t best_step = 0, accepted = 0, ties = 0, len = 1;
    int32_t* hist = o->energy_hist ? o->energy_hist + r * p->hist_stride : NULL;
    uint64_t* bits = o->accept_bits ? o->accept_bits + r * p->bits_stride : NULL;
    if (hist) hist[0] = E;
    if (bits)
        for (int64_t w = 0; w < p->bits_stride; w++) bits[w] = 0;
    if (o->initial_energy) o->initial_energy[r] = E;
    for (int64_t step = 0; step < p->n_steps; step++) {
        double beta = beta_at(p, step);
        int qi = (int)mt_bounded(&rng, (uint32_t)(Q - 1));
        cell_t od = q[qi], nw;
        for (;;) {
            nw.i = (int)mt_bounded(&rng, (uint32_t)(N - 1));
            nw.j = (int)mt_bounded(&rng, (uint32_t)(N - 1));
            nw.k = (int)mt_bounded(&rng, (uint32_t)(N - 1));
            if (!occ[(nw.i * N + nw.j) * N + nw.k]) break;
        }
        /* old cell: the moving queen is on all 13 of its lines; new cell: on at most one (iff the two cells attack) */
        int dE = (lines_sum(&L, nw.i, nw.j, nw.k) - full_attacks(od, nw)) - (lines_sum(&L, od.i, od.j, od.k) - 13);
        int acc = accept_move(&rng, beta, dE, &ties);
        if (acc) {
            if (bits) bits[step >> 6] |= 1ull << (step & 63);
            lines_add(&L, od.i, od.j, od.k, -1);
            lines_add(&L, nw.i, nw.j, nw.k, 1);
            occ[(od.i * N + od.j) * N + od.k] = 0;
            occ[(nw.i * N + nw.j) * N + nw.k] = 1;
            q[qi] = nw;
            E += dE;
            accepted++;
            if (E < best) {
                best = E;
                memcpy(best_q, q, sizeof(cell_t) * (size_t)Q);
                best_step = len;
            }
        }
        if (hist) hist[len] = E;
        len++;
    }
    if (o->hist_len) o->hist_len[r] = len;
    if (o->steps_executed) o->steps_executed[r] = p->n_steps;
    if (o->best_energy) o->best_energy[r] = best;
    if (o->final_energy) o->final_energy[r] = E;
    if (o->steps_to_best) o->steps_to_best[r] = best_step;
    if (o->n_accepted) o->n_accepted[r] = accepted;
    if (o->stream_words) o->stream_words[r] = p->rng == MCQ_RNG_MT19937_NUMPY ? (uint32_t)rng.words : 0u;
    if (o->near_ties) o->near_ties[r] = ties;
    for (int c = 0; c < Q; c++) {
        if (o->best_state) {
            uint8_t* d = o->best_state + ((size_t)r * Q + c) * 3;
            d[0] = (uint8_t)best_q[c].i, d[1] = (uint8_t)best_q[c].j, d[2] = (uint8_t)best_q[c].k;
        }
        if (o->final_state) {
            uint8_t* d = o->final_state + ((size_t)r * Q + c) * 3;
            d[0] = (uint8_t)q[c].i, d[1] = (uint8_t)q[c].j, d[2] = (uint8_t)q[c].k;
        }
    }
    free(q), free(occ), free(cnt);
    return MCQ_OK;
}

/* ------------------------------------------------------------------------------------------
 * Replica exchange (mcq_params.exchange_every > 0; include/mcq.h).  NOT a mode of the reference: its report (section VI) names
 * better moves as future work; the swap uses the reference's accept rule (experiments.py:326-327) on a pair of chains.  Parity in
 * this mode is HIP == this file.  The chains of a ladder advance in lockstep, so a chain is an object with a step function here;
 * the step is the same restatement as run_board_chain / run_full_chain above (tests/test_exchange.py holds the two against each
 * other with a ladder of ones and an exchange period beyond the run).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const mcq_params* p; /* the chain's set */
    const mcq_outputs* o;
    int64_t r;
    int fast, N, Q;
    rng_t rng;
    int *h, *best_h;         /* board */
    cell_t *q, *best_q;      /* full_3d */
    uint8_t* occ;
    uint8_t* cnt;            /* fast: line counters */
    lines_t L;
    int E, best;
    int64_t best_step, accepted, ties, len;
    int32_t* hist;
    uint64_t* bits;
    int rung;
    int64_t n_exch;
} xchain_t;

static void xchain_free(xchain_t* c) {
    free(c->h), free(c->q), free(c->occ), free(c->cnt);
    c->h = NULL, c->q = NULL, c->occ = NULL, c->cnt = NULL;
}

static int xchain_init(xchain_t* c, const mcq_params* p, const mcq_outputs* o, const uint32_t* seeds, int64_t r, int fast, int rung) {
    memset(c, 0, sizeof *c);
    c->p = p, c->o = o, c->r = r, c->fast = fast, c->N = p->N, c->Q = queens_of(p), c->rung = rung;
    const int N = c->N, Q = c->Q;
    rng_start(&c->rng, p, seeds, r);
    if (fast) {
        c->cnt = (uint8_t*)malloc(lines_total(N));
        if (!c->cnt) return MCQ_ENOMEM;
        lines_setup(&c->L, N, c->cnt);
    }
    if (p->mode == MCQ_MODE_BOARD) {
        c->h = (int*)malloc(sizeof(int) * (size_t)Q * 2);
        if (!c->h) return MCQ_ENOMEM;
        c->best_h = c->h + Q;
        if (board_init(N, p->init, &c->rng, c->h) != 0) return MCQ_EINVAL;
        if (fast) {
            for (int t = 0; t < Q; t++) lines_add(&c->L, t / N, t % N, c->h[t], 1);
            c->E = lines_energy(&c->L);
        } else {
            c->E = board_energy(N, c->h);
        }
        memcpy(c->best_h, c->h, sizeof(int) * (size_t)Q);
    } else {
        c->q = (cell_t*)malloc(sizeof(cell_t) * (size_t)Q * 2);
        c->occ = (uint8_t*)malloc((size_t)N * N * N);
        if (!c->q || !c->occ) return MCQ_ENOMEM;
        c->best_q = c->q + Q;
        if (full_init(N, Q, p->init, &c->rng, c->q, c->occ) != 0) return MCQ_EINVAL;
        if (fast) {
            for (int t = 0; t < Q; t++) lines_add(&c->L, c->q[t].i, c->q[t].j, c->q[t].k, 1);
            c->E = lines_energy(&c->L);
        } else {
            c->E = full_energy(Q, c->q);
        }
        memcpy(c->best_q, c->q, sizeof(cell_t) * (size_t)Q);
    }
    c->best = c->E, c->len = 1;
    c->hist = o->energy_hist ? o->energy_hist + r * p->hist_stride : NULL;
    c->bits = o->accept_bits ? o->accept_bits + r * p->bits_stride : NULL;
    if (c->hist) c->hist[0] = c->E;
    if (c->bits)
        for (int64_t w = 0; w < p->bits_stride; w++) c->bits[w] = 0;
    if (o->initial_energy) o->initial_energy[r] = c->E;
    return MCQ_OK;
}

/* one Metropolis step at inverse temperature beta (experiments.py:308-358 / 218-258; no early stop in this mode) */
static void xchain_step(xchain_t* c, int64_t step, double beta) {
    const int N = c->N, Q = c->Q;
    rng_t* rng = &c->rng;
    int dE, acc;
    if (c->p->mode == MCQ_MODE_BOARD) {
        int* h = c->h;
        int i = (int)mt_bounded(rng, (uint32_t)(N - 1));
        int j = (int)mt_bounded(rng, (uint32_t)(N - 1));
        int old_k = h[i * N + j];
        int old_c = c->fast ? 0 : board_conflicts(N, h, i, j, old_k);
        int new_k = (int)mt_bounded(rng, (uint32_t)(N - 1));
        while (new_k == old_k) new_k = (int)mt_bounded(rng, (uint32_t)(N - 1));
        if (c->fast) dE = lines_sum(&c->L, i, j, new_k) - lines_sum(&c->L, i, j, old_k) + 12;
        else dE = board_conflicts(N, h, i, j, new_k) - old_c;
        acc = accept_move(rng, beta, dE, &c->ties);
        if (acc) {
            if (c->fast) lines_add(&c->L, i, j, old_k, -1), lines_add(&c->L, i, j, new_k, 1);
            h[i * N + j] = new_k;
        }
    } else {
        cell_t* q = c->q;
        int qi = (int)mt_bounded(rng, (uint32_t)(Q - 1));
        cell_t od = q[qi], nw;
        int old_c = c->fast ? 0 : full_conflicts(Q, q, qi, od);
        for (;;) {
            nw.i = (int)mt_bounded(rng, (uint32_t)(N - 1));
            nw.j = (int)mt_bounded(rng, (uint32_t)(N - 1));
            nw.k = (int)mt_bounded(rng, (uint32_t)(N - 1));
            if (!c->occ[(nw.i * N + nw.j) * N + nw.k]) break;
        }
        if (c->fast) dE = (lines_sum(&c->L, nw.i, nw.j, nw.k) - full_attacks(od, nw)) - (lines_sum(&c->L, od.i, od.j, od.k) - 13);
        else dE = full_conflicts(Q, q, qi, nw) - old_c;
        acc = accept_move(rng, beta, dE, &c->ties);
        if (acc) {
            if (c->fast) lines_add(&c->L, od.i, od.j, od.k, -1), lines_add(&c->L, nw.i, nw.j, nw.k, 1);
            c->occ[(od.i * N + od.j) * N + od.k] = 0;
            c->occ[(nw.i * N + nw.j) * N + nw.k] = 1;
            q[qi] = nw;
        }
    }
    if (acc) {
        if (c->bits) c->bits[step >> 6] |= 1ull << (step & 63);
        c->E += dE;
        c->accepted++;
        if (c->E < c->best) {
            c->best = c->E;
            if (c->h) memcpy(c->best_h, c->h, sizeof(int) * (size_t)Q);
            else memcpy(c->best_q, c->q, sizeof(cell_t) * (size_t)Q);
            c->best_step = c->len;
        }
    }
    if (c->hist) c->hist[c->len] = c->E;
    c->len++;
}

static void xchain_finish(xchain_t* c) {
    const mcq_outputs* o = c->o;
    const int64_t r = c->r;
    const int Q = c->Q;
    if (o->hist_len) o->hist_len[r] = c->len;
    if (o->steps_executed) o->steps_executed[r] = c->p->n_steps;
    if (o->best_energy) o->best_energy[r] = c->best;
    if (o->final_energy) o->final_energy[r] = c->E;
    if (o->steps_to_best) o->steps_to_best[r] = c->best_step;
    if (o->n_accepted) o->n_accepted[r] = c->accepted;
    if (o->stream_words) o->stream_words[r] = c->p->rng == MCQ_RNG_MT19937_NUMPY ? (uint32_t)c->rng.words : 0u;
    if (o->near_ties) o->near_ties[r] = c->ties;
    if (o->exchange_rung) o->exchange_rung[r] = c->rung;
    if (o->n_exchanges) o->n_exchanges[r] = c->n_exch;
    for (int t = 0; t < Q; t++) {
        if (c->h) {
            if (o->best_state) o->best_state[r * Q + t] = (uint8_t)c->best_h[t];
            if (o->final_state) o->final_state[r * Q + t] = (uint8_t)c->h[t];
        } else {
            if (o->best_state) {
                uint8_t* d = o->best_state + ((size_t)r * Q + t) * 3;
                d[0] = (uint8_t)c->best_q[t].i, d[1] = (uint8_t)c->best_q[t].j, d[2] = (uint8_t)c->best_q[t].k;
            }
            if (o->final_state) {
                uint8_t* d = o->final_state + ((size_t)r * Q + t) * 3;
                d[0] = (uint8_t)c->q[t].i, d[1] = (uint8_t)c->q[t].j, d[2] = (uint8_t)c->q[t].k;
            }
        }
    }
}

/* one ladder: chains [g * R, (g + 1) * R) in lockstep */
static int run_exchange_group(const job_t* jb, int64_t g) {
    const mcq_params* p0 = jb->p;
    const int R = p0->exchange_replicas;
    const int64_t K = p0->exchange_every, r0 = g * R;
    mcq_params ps = *p0; /* the ladder's set (chains_per_set is a multiple of R) */
    if (p0->n_sets > 1) {
        const int64_t t = r0 / p0->chains_per_set;
        const mcq_schedule* sc = &p0->sets[t];
        ps.sched = sc->sched, ps.beta_const = sc->beta_const, ps.beta_start = sc->beta_start, ps.beta_end = sc->beta_end;
        if (sc->init_plus1) ps.init = sc->init_plus1 - 1;
        if (ps.beta_table) ps.beta_table += t * ps.n_steps;
    }
    xchain_t ch[16];
    int on_rung[16]; /* rung -> chain of the ladder */
    int rc = MCQ_OK;
    memset(ch, 0, sizeof ch);
    for (int c = 0; c < R && rc == MCQ_OK; c++) {
        rc = xchain_init(&ch[c], &ps, jb->out, jb->seeds, r0 + c, jb->fast, c);
        on_rung[c] = c;
    }
    for (int64_t step = 0; step < ps.n_steps && rc == MCQ_OK; step++) {
        const double beta = beta_at(&ps, step);
        for (int c = 0; c < R; c++) xchain_step(&ch[c], step, beta * p0->exchange_ladder[ch[c].rung]);
        if ((step + 1) % K != 0) continue;
        const int64_t n = (step + 1) / K;
        for (int t = (int)(n & 1); t + 1 < R; t += 2) {
            xchain_t *a = &ch[on_rung[t]], *b = &ch[on_rung[t + 1]];
            const double ba = beta * p0->exchange_ladder[t], bb = beta * p0->exchange_ladder[t + 1];
            const double x = (ba - bb) * (double)(a->E - b->E);
            const double e = exp(x), prob = e < 1.0 ? e : 1.0;
            const double u = mt_double(&a->rng); /* always drawn, from the lower rung's stream */
            if (prob < 1.0 && ulp_distance(u, prob) <= 4) a->ties++;
            if (u < prob) {
                const int ca = on_rung[t];
                on_rung[t] = on_rung[t + 1], on_rung[t + 1] = ca;
                a->rung = t + 1, b->rung = t;
                a->n_exch++, b->n_exch++;
            }
        }
    }
    for (int c = 0; c < R; c++) {
        if (rc == MCQ_OK) xchain_finish(&ch[c]);
        xchain_free(&ch[c]);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * fan-out: one task per chain, chain r seeded with seeds[r] (experiments.py:507-517)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const job_t* jb;
    int64_t next;
    int rc;
    pthread_mutex_t mu;
} pool_t;

static int run_chain(const job_t* jb, int64_t r) {
    if (jb->fast) return jb->p->mode == MCQ_MODE_BOARD ? run_board_chain_fast(jb, r) : run_full_chain_fast(jb, r);
    return jb->p->mode == MCQ_MODE_BOARD ? run_board_chain(jb, r) : run_full_chain(jb, r);
}

static int run_one(const job_t* jb, int64_t r) {
    if (jb->p->exchange_every > 0) return run_exchange_group(jb, r); /* the task is a ladder */
    if (jb->p->n_sets > 1) { /* batched schedules: chain r follows sets[r / chains_per_set] */
        const mcq_schedule* sc = &jb->p->sets[r / jb->p->chains_per_set];
        mcq_params q = *jb->p;
        job_t one = *jb;
        q.sched = sc->sched, q.beta_const = sc->beta_const, q.beta_start = sc->beta_start, q.beta_end = sc->beta_end;
        if (sc->init_plus1) q.init = sc->init_plus1 - 1; /* a set may have its own init mode */
        if (q.beta_table) q.beta_table += (r / jb->p->chains_per_set) * q.n_steps; /* set-major table */
        one.p = &q;
        return run_chain(&one, r);
    }
    return run_chain(jb, r);
}

static int64_t n_tasks(const mcq_params* p) { return p->exchange_every > 0 ? p->n_chains / p->exchange_replicas : p->n_chains; }

static void* worker(void* arg) {
    pool_t* pl = (pool_t*)arg;
    for (;;) {
        pthread_mutex_lock(&pl->mu);
        int64_t r = pl->next++;
        pthread_mutex_unlock(&pl->mu);
        if (r >= n_tasks(pl->jb->p)) break;
        int rc = run_one(pl->jb, r);
        if (rc != MCQ_OK) {
            pthread_mutex_lock(&pl->mu);
            pl->rc = rc;
            pthread_mutex_unlock(&pl->mu);
        }
    }
    return NULL;
}

static int oracle_run_impl(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, int n_threads, int fast);

int mcq_oracle_run(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, int n_threads) {
    return oracle_run_impl(p, seeds, out, n_threads, 0);
}

int mcq_oracle_run_fast(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, int n_threads) {
    return oracle_run_impl(p, seeds, out, n_threads, 1);
}

static int oracle_run_impl(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, int n_threads, int fast) {
    if (!p || !seeds || !out) return fail(MCQ_EINVAL, "null argument");
    if (p->abi_version != MCQ_ABI_VERSION) return fail(MCQ_EINVAL, "abi_version mismatch");
    if (p->N < MCQ_MIN_N || p->N > (p->mode == MCQ_MODE_BOARD ? MCQ_MAX_N_BOARD : MCQ_MAX_N)) return fail(MCQ_EINVAL, "N out of range");
    if (p->mode != MCQ_MODE_BOARD && p->mode != MCQ_MODE_FULL3D) return fail(MCQ_EINVAL, "unknown mcmc_type");
    if (p->stream_states) {
        if (p->rng != MCQ_RNG_MT19937_NUMPY) return fail(MCQ_EINVAL, "stream_states continues an MT19937 stream (a Philox stream is a function of its seed)");
        for (int64_t r = 0; r < p->n_chains; r++)
            if (p->stream_states[r * 625 + 624] > 624u) return fail(MCQ_EINVAL, "stream_states: MT19937 position out of range");
    }
    if (p->init < MCQ_INIT_RANDOM || p->init > MCQ_INIT_KLARNER) return fail(MCQ_EINVAL, "Unknown init_mode");
    if (p->sched < MCQ_SCHED_CONSTANT || p->sched > MCQ_SCHED_SINUSOIDAL)
        return fail(MCQ_EINVAL, "Unknown betta_scheduling type");
    if (p->n_sets > 1) {
        if (!p->sets || p->chains_per_set <= 0 || p->n_chains != p->n_sets * p->chains_per_set)
            return fail(MCQ_EINVAL, "bad schedule sets");
        for (int64_t t = 0; t < p->n_sets; t++)
            if (p->sets[t].sched < MCQ_SCHED_CONSTANT || p->sets[t].sched > MCQ_SCHED_SINUSOIDAL)
                return fail(MCQ_EINVAL, "Unknown betta_scheduling type");
    }
    if (p->rng != MCQ_RNG_MT19937_NUMPY && p->rng != MCQ_RNG_PHILOX4X32_10) return fail(MCQ_EINVAL, "unknown rng");
    if (p->n_steps < 0 || p->n_chains < 0) return fail(MCQ_EINVAL, "negative n_steps / n_chains");
    if (out->energy_hist && p->hist_stride < p->n_steps + 1) return fail(MCQ_EINVAL, "hist_stride too small");
    if (out->accept_bits && p->bits_stride < (p->n_steps + 63) / 64) return fail(MCQ_EINVAL, "bits_stride too small");
    if (p->trace == MCQ_TRACE_I32 && (!out->energy_hist || !out->accept_bits))
        return fail(MCQ_EINVAL, "trace requested without buffers");
    if (p->n_queens < 0) return fail(MCQ_EINVAL, "negative n_queens");
    if (p->n_queens > 0 && p->n_queens != p->N * p->N) {
        if (p->mode != MCQ_MODE_FULL3D) return fail(MCQ_EINVAL, "n_queens applies to mcmc_type full_3d (a board has one queen per column)");
        int other_init = p->init != MCQ_INIT_RANDOM;
        for (int64_t t = 0; t < p->n_sets && p->n_sets > 1; t++)
            other_init |= p->sets[t].init_plus1 != 0 && p->sets[t].init_plus1 != MCQ_INIT_RANDOM + 1;
        if (other_init) return fail(MCQ_EINVAL, "latin / klarner initialization assumes Q = N^2");
        if (p->n_queens < 2) return fail(MCQ_EINVAL, "n_queens must be at least 2 in this build");
        if ((int64_t)p->n_queens >= (int64_t)p->N * p->N * p->N) return fail(MCQ_EINVAL, "n_queens must leave a free cell: Q < N^3");
    }
    if (p->exchange_every < 0) return fail(MCQ_EINVAL, "negative exchange_every");
    if (p->exchange_every > 0) {
        const int R = p->exchange_replicas;
        if (R != 2 && R != 4 && R != 8 && R != 16) return fail(MCQ_EINVAL, "exchange_replicas must be 2, 4, 8 or 16");
        if (!p->exchange_ladder) return fail(MCQ_EINVAL, "exchange_every > 0 without exchange_ladder");
        for (int t = 0; t < R; t++)  /* a rung runs at beta(step) * ladder[t]: a multiplier that is not a positive finite number has no meaning */
            if (!(p->exchange_ladder[t] > 0.0) || p->exchange_ladder[t] > 1.7976931348623157e308) return fail(MCQ_EINVAL, "exchange_ladder entries must be finite and positive");
        if (p->n_chains % R != 0 || (p->n_sets > 1 && p->chains_per_set % R != 0))
            return fail(MCQ_EINVAL, "n_chains (and chains_per_set) must be multiples of exchange_replicas");
        if (p->mode == MCQ_MODE_BOARD && p->patience >= 0 && p->patience <= p->n_steps)
            return fail(MCQ_EINVAL, "replica exchange needs early stopping off (early_stop_patience None)");
        if (p->trace == MCQ_TRACE_REDUCED) return fail(MCQ_EINVAL, "replica exchange runs with trace none or i32");
    }

    job_t jb = {p, seeds, out, fast};
    if (n_threads <= 1) {
        for (int64_t r = 0; r < n_tasks(p); r++) {
            int rc = run_one(&jb, r);
            if (rc != MCQ_OK) return fail(rc, "chain failed");
        }
        return MCQ_OK;
    }
    pool_t pl = {&jb, 0, MCQ_OK, PTHREAD_MUTEX_INITIALIZER};
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!th) return fail(MCQ_ENOMEM, "out of memory");
    int started = 0;
    for (; started < n_threads; started++)
        if (pthread_create(&th[started], NULL, worker, &pl) != 0) break;
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th);
    if (started == 0) return fail(MCQ_ENOMEM, "could not start threads");
    if (pl.rc != MCQ_OK) return fail(pl.rc, "chain failed");
    return MCQ_OK;
}
