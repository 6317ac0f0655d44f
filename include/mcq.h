/*
 * mcq.h -- C-ABI of the MI355X-native Metropolis sweep for the 3D N^2-queens problem.
 *
 * The reference (galgantar/monte-carlo-collective) has no FFI or plugin interface: its
 * seam is the Python function run_experiment() (experiments.py:475-573), which fans one
 * task per chain out to a process pool (experiments.py:507-517) and each task runs
 * metropolis_mcmc_board (experiments.py:282-376) or metropolis_mcmc
 * (experiments.py:199-279).  This header is the boundary a maintainer of the reference
 * would bind with ctypes to replace that fan-out + sweep: one call runs ALL chains.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; the library never returns owned memory.
 *   - every function returns 0 on success, a negative MCQ_E* code on error; the message
 *     for the last error of the calling thread is mcq_last_error().
 *   - no exceptions cross the boundary; the Python side maps MCQ_EINVAL to ValueError
 *     (the reference raises ValueError for unknown schedule / init modes:
 *     experiments.py:105, mcmc.py:104, mcmc_board.py:59) and the rest to RuntimeError.
 *
 * Two libraries export (subsets of) this interface:
 *   libmcq_hip.so     the product: hand-written HIP kernels for gfx950 (csrc/).
 *   libmcq_oracle.so  TEST INFRASTRUCTURE ONLY: a plain-C CPU restatement of the
 *                     reference algorithm (oracle/), exporting mcq_oracle_run().
 */
#ifndef MCQ_H
#define MCQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCQ_ABI_VERSION 5

/* error codes */
#define MCQ_OK 0
#define MCQ_EINVAL (-1)   /* bad parameter (-> ValueError) */
#define MCQ_EDEVICE (-2)  /* HIP runtime error, no device, launch failure */
#define MCQ_ENOMEM (-3)   /* workspace too small / allocation failure */

/* mcmc_type: experiments.py:497-502 ("board" -> board chain, anything else -> full_3d) */
#define MCQ_MODE_BOARD 0
#define MCQ_MODE_FULL3D 1

/* init_mode: mcmc_board.py:26-59, mcmc.py:20-104 */
#define MCQ_INIT_RANDOM 0
#define MCQ_INIT_LATIN 1
#define MCQ_INIT_KLARNER 2

/* betta_scheduling.type: experiments.py:79-105 */
#define MCQ_SCHED_CONSTANT 0
#define MCQ_SCHED_LINEAR 1
#define MCQ_SCHED_EXPONENTIAL 2
#define MCQ_SCHED_LOGARITHMIC 3
#define MCQ_SCHED_SINUSOIDAL 4

/* random stream */
#define MCQ_RNG_MT19937_NUMPY 0 /* NumPy legacy global RandomState: bit-parity with the reference */
#define MCQ_RNG_PHILOX4X32_10 1 /* counter-based fast mode, NOT a stream of the reference: word w of chain r is
                                   philox4x32-10(counter = (low 32 bits of w / 4, high bits of w / 4, 0, 0), key = (seeds[r], 0))[w % 4], consumed with the same
                                   rules as the NumPy stream (masked rejection, 53-bit doubles, Fisher-Yates); no generator state
                                   lives in memory.  Results equal the oracle's in the same mode, not the reference's. */

/* energy_history trace */
#define MCQ_TRACE_NONE 0 /* only per-chain summaries (measure_min_energy_vs_N discards histories: experiments.py:1061) */
#define MCQ_TRACE_I32 1  /* full int32 trace + accept bits (experiments.py:355, 329-332) */
#define MCQ_TRACE_REDUCED 2 /* no per-chain trace: per-entry sums over the chains (what the plots consume, experiments.py:593-595, 660-695) */

/* flags */
#define MCQ_FLAG_EXACT_EXP 1u        /* evaluate exp(-beta*dE) in float64 on every step (disable the float32 bracket) */
#define MCQ_FLAG_SEQUENTIAL_DRAWS 2u /* HIP: draw every proposal word by word (disable the batched selection); for testing */
#define MCQ_FLAG_LINE_COUNTERS 4u    /* HIP: boards up to N = 8 at 4 lanes per chain take dE from per-line occupancy counters in LDS (one byte per line of
                                        the 12 families, 2 N^2 + 6 N (2N-1) + 4 (2N-1)^2 bytes per chain) instead of bit-mask probes of the heights; ignored
                                        where it does not apply (larger N, other lane counts, Philox, reduced trace, exchange).  Never changes a result. */
#define MCQ_FLAG_SHARED_PACING 16u   /* HIP: the launch paces its wavefronts (s_setprio by progress) against ALL launches of this process on the device that
                                        set the flag -- one progress table per device -- and does so whatever its size; without the flag a launch paces itself against
                                        its own wavefronts only, and only when it puts two or more on a SIMD.  For callers that run several launches of equal length
                                        side by side (jobs.JobSet).  Never changes a result. */
/* HIP: bits 8..9 of flags = the hardware priority (s_setprio 0..3) the launch's wavefronts run at when the launch is too small to pace
 * itself (fewer than two wavefronts per SIMD).  For callers that run several launches side by side: the long ones get precedence, the
 * short ones fill the gaps.  Never changes a result.  MCQ_FLAG_PRIORITY(p) builds the bits. */
#define MCQ_FLAG_PRIORITY_SHIFT 8
#define MCQ_FLAG_PRIORITY(p) (((uint32_t)(p) & 3u) << MCQ_FLAG_PRIORITY_SHIFT)

/* Upper bounds of this build (N >= 2 is required by the reference loop at experiments.py:317-319; the reference itself is
 * unbounded).  full_3d: column occupancy is one word per column -- 16 bits up to N = 16, 32 up to N = 32, 64 up to N = 64 (that variant:
 * 16 lanes per chain, NumPy's stream, no replica exchange; the queen table and the N^3 cells np.random.choice permutes live in the
 * workspace: N^3 * 4 bytes for each of the chains one round of the init kernel takes -- as many as 1 GiB holds).  board: bit masks up to N = 32, a compare per probed
 * height beyond (slower, any size whose N*N heights fit a wavefront's share of the LDS and whose accept flags fit a byte). */
#define MCQ_MIN_N 2
#define MCQ_MAX_N 64        /* mcmc_type full_3d (beyond 32: 64-bit column words, 16 lanes per chain) */
#define MCQ_MAX_N_BOARD 128 /* mcmc_type board */

/* One set of a batched run: its beta schedule (run_beta_start_end_pairs loops over such pairs: experiments.py:741-846) and,
 * optionally, its own init mode. */
typedef struct mcq_schedule {
    int32_t sched;     /* MCQ_SCHED_* */
    int32_t init_plus1; /* 0: the set starts from mcq_params.init; otherwise MCQ_INIT_* + 1 -- the (init_mode, N) cells of
                           measure_min_energy_vs_N that share N (experiments.py:1050-1067) then run as one launch */
    double beta_const;
    double beta_start;
    double beta_end;
} mcq_schedule;

typedef struct mcq_params {
    int32_t abi_version;     /* MCQ_ABI_VERSION */
    int32_t N;               /* board edge; Q = N*N queens */
    int32_t mode;            /* MCQ_MODE_* */
    int32_t init;            /* MCQ_INIT_* */
    int32_t sched;           /* MCQ_SCHED_* */
    int32_t rng;             /* MCQ_RNG_* */
    int32_t trace;           /* MCQ_TRACE_* */
    uint32_t flags;          /* MCQ_FLAG_* */
    double beta_const;       /* constant schedule            (experiments.py:13-16)  */
    double beta_start;       /* annealing schedules          (experiments.py:19-77)  */
    double beta_end;
    int64_t n_steps;         /* steps per chain = schedule length                    */
    int64_t n_chains;        /* n_runs (< 2^31); chain r is seeded with seeds[r] (= base_seed + r, experiments.py:508) */
    int64_t patience;        /* early_stop_patience, board only (experiments.py:349-353); < 0 = None */
    int64_t hist_stride;     /* int32 elements per chain row of energy_hist, >= n_steps + 1, a multiple of 16 and < 2^24 (HIP: rows
                                are written in aligned 64-byte segments; energy_hist itself must be 64-byte aligned; runs of more
                                than 2^24 - 16 steps take trace = REDUCED or NONE) */
    int64_t bits_stride;     /* uint64 words per chain row of accept_bits, >= ceil(n_steps / 64) */
    int32_t lanes_per_chain; /* HIP only: 2 (boards), 4, 8 or 16 lanes of a wavefront per chain; 0 = library default */
    int32_t device;          /* HIP only, host-buffer entry point: device ordinal, < 0 = current device */
    /* Several schedules in ONE launch (everything else shared): n_sets <= 1 means the single schedule above.  Otherwise
     * chains [t * chains_per_set, (t + 1) * chains_per_set) follow sets[t]; n_chains == n_sets * chains_per_set,
     * chains_per_set is a multiple of 16, and with trace == REDUCED the step_* outputs are [n_sets][n_steps + 1]. */
    int64_t n_sets;
    int64_t chains_per_set;
    const mcq_schedule* sets; /* HOST pointer (also for mcq_run_device), n_sets entries */
    /* Optional beta(step) values, double[n_sets <= 1 ? 1 : n_sets][n_steps] (set-major): when given, the sweep uses exactly
     * these instead of evaluating the schedule on the device.  The reference evaluates its schedules with NumPy's exp / log /
     * cos (experiments.py:27-77), whose last bit differs between NumPy, glibc and the GPU's math library on ~0.1-5 % of the
     * arguments; a caller that wants beta bit-identical to the reference passes the reference's own values (the Python side
     * does: abi.host_beta_table).  mcq_run_device: DEVICE pointer; mcq_run_host: HOST pointer.  NULL: computed on the device
     * (exact for constant / linear; the other three within 2^-51 * max(|beta_start|, |beta_end|) of the reference's value). */
    const double* beta_table;
    /* Replica exchange (parallel tempering) between the chains of a launch -- NOT a mode of the reference (its report, section VI,
     * names better moves as future work), never a default, results equal the oracle's in the same mode.
     * exchange_every = K > 0 turns it on: chains [g * R, (g + 1) * R), R = exchange_replicas (2, 4, 8 or 16; n_chains and
     * chains_per_set are multiples of R), form one ladder.  Chain r starts on rung r % R; a chain on rung t runs every step s
     * at beta(s) * exchange_ladder[t] (one float64 multiply; beta(s) from the schedule / beta_table as without exchange).
     * After every K-th step (steps K-1, 2K-1, ... of the run, n = (s + 1) / K = 1, 2, ...) the rungs (t, t + 1) with
     * t = (n & 1), (n & 1) + 2, ... <= R - 2 are offered a swap: with a, b the chains on rungs t, t + 1,
     *     x = (beta_a - beta_b) * (double)(E_a - E_b),   beta_a = beta(s) * ladder[t],  beta_b = beta(s) * ladder[t + 1],
     * the chain on rung t draws u = random() from ITS stream (two words, right after its step-s words; always drawn) and the two
     * chains trade rungs iff u < min(1, exp(x)) -- the reference's accept rule (experiments.py:326-327) on the pair.  States,
     * energies and histories stay with their chains; only the rung (hence beta) moves.  Needs patience < 0 (no early stop) and
     * trace != REDUCED. */
    int64_t exchange_every;        /* 0 = off */
    int32_t exchange_replicas;     /* R */
    int32_t n_queens;              /* full_3d only: Q queens instead of N*N (State3DQueens(N, Q=...), mcmc.py:6-18; metropolis_mcmc(..., Q=...),
                                      experiments.py:199-203), 2 <= Q < N^3, random init only (latin / klarner assume Q = N^2: mcmc.py:21-25);
                                      0 = N*N.  state_bytes becomes 3 Q (mcq_state_bytes_for). */
    const double* exchange_ladder; /* HOST pointer (also for mcq_run_device), R finite positive multipliers.  Read during the call: validated, and copied to the
                                      device with a hipMemcpyAsync on the caller's stream from THIS (normally pageable) array, so it must stay valid until
                                      that copy has run -- mcq_run_host and the Python wrappers keep it alive and synchronise; a caller of mcq_run_device
                                      keeps it until the stream has passed the call (with exchange the call cannot be part of a stream capture) */
    /* Chains that CONTINUE a random stream instead of seeding one: metropolis_mcmc[_board](..., seed=None) skips np.random.seed and draws from
     * NumPy's global RandomState wherever it stands (experiments.py:200-201, 287-288).  Optional, uint32[n_chains][625]: per chain the 624 key
     * words and the position (0..624) of an MT19937 state exactly as np.random.get_state() returns them; seeds[r] is ignored for such a run.
     * HOST pointer (also for mcq_run_device: the library rewinds the unconsumed part of the generation into the layout its kernels stream from,
     * copies the states to the workspace and synchronises the stream once); MCQ_RNG_MT19937_NUMPY only.  mcq_outputs.stream_words tells how far
     * each chain went, so the caller can advance its own copy of the stream by as many words. */
    const uint32_t* stream_states;
} mcq_params;

/*
 * Per-chain outputs; every array is caller-allocated.  state_bytes = mcq_state_bytes(N, mode) (mcq_state_bytes_for(p) with n_queens):
 *   board:   N*N uint8 heights, row-major heights[i][j]        (mcmc_board.py:28)
 *   full_3d: Q*3 uint8 (i, j, k) per queen, in queen-index order (mcmc.py:101)
 * Pointers that may be NULL are marked optional.
 */
typedef struct mcq_outputs {
    int32_t* energy_hist;    /* optional unless trace == I32: [n_chains][hist_stride]; entry 0 = E0, entry s+1 = energy after step s */
    uint64_t* accept_bits;   /* optional unless trace == I32: [n_chains][bits_stride]; bit (s & 63) of word s >> 6 = step s accepted */
    int64_t* hist_len;       /* [n_chains] entries of energy_hist that are valid (n_steps + 1, fewer after an early stop) */
    int64_t* steps_executed; /* [n_chains] proposals made (hist_len - 1, +1 when the chain broke out early) */
    int32_t* initial_energy; /* [n_chains] E0 */
    int32_t* best_energy;    /* [n_chains] */
    int32_t* final_energy;   /* [n_chains] */
    int64_t* steps_to_best;  /* [n_chains] first index of min(energy_history) (experiments.py:364-365) */
    int64_t* n_accepted;     /* [n_chains] */
    int64_t* near_ties;      /* optional [n_chains]: steps whose uniform fell within 4 ulp of the acceptance probability */
    uint8_t* best_state;     /* optional [n_chains][state_bytes]; HIP: 16-byte aligned base (rows are copied 16 bytes at a time) */
    uint8_t* final_state;    /* optional [n_chains][state_bytes]; HIP: 16-byte aligned base */
    /* trace == REDUCED only, int64[n_steps + 1] each (per schedule set: [n_sets][n_steps + 1]), indexed by history entry e
     * (entry 0 = initial state): */
    int64_t* step_sum;       /* sum over chains of energy_history[e]                                   */
    int64_t* step_sumsq;     /* sum of squares                                                         */
    int64_t* step_accepted;  /* chains whose step e - 1 was accepted (entry 0: 0); includes the step at which a chain stopped
                                early, which is executed and listed in accepted_steps but appends no entry (experiments.py:329-353) */
    int64_t* step_count;     /* chains whose history has entry e (< n_chains only after early stops)    */
    /* exchange_every > 0 only, optional [n_chains] each: */
    int32_t* exchange_rung;  /* the rung the chain ends on */
    int64_t* n_exchanges;    /* accepted swaps the chain took part in */
    uint32_t* stream_words;  /* optional [n_chains]: 32-bit words the chain took from its random stream -- initial state, every step's draws (the
                                rejected words of randint included), exchange uniforms -- modulo 2^32; 0 with MCQ_RNG_PHILOX4X32_10 (a stream addressed
                                by position: nothing to hand back).  Equal between the oracle and the kernels like every other output: the streams
                                are consumed identically, not only the results */
} mcq_outputs;

/* ---- exported by libmcq_hip.so ------------------------------------------------------------ */

int mcq_abi_version(void);
const char* mcq_last_error(void);
int mcq_device_count(void);

/* lanes of a wavefront per chain used when mcq_params.lanes_per_chain == 0: board 4 up to N = 12 and 8 beyond, full_3d 8 (16 beyond
 * N = 32, the only width of that variant; 4 for N = 9..12 with NumPy's stream) (mcq_default_lanes: the value for small boards).  A board launch that leaves SIMDs empty runs at twice or four times the
 * lanes while every wavefront still has a SIMD to itself (N >= 20: at most 8; N <= 8: always 4); with replica exchange a ladder must fit one
 * wavefront.  mcq_effective_lanes tells.  The lane count never changes a result. */
/* mcq_params.stream_states, one chain: an MT19937 state as np.random.get_state() holds it (uint32[625]: key, position) in the layout the kernels stream
 * from (uint32[626]: words [0, out[625]) of the current generation, the rest rewound to the generation before; out[624] = position).  Pure host code:
 * what mcq_run_device does with every state before it copies them to the workspace; exported for the tests. */
void mcq_stream_layout(const uint32_t* numpy_state, uint32_t* out);
int32_t mcq_default_lanes(int32_t mode);
int32_t mcq_default_lanes_n(int32_t mode, int32_t N);
/* the lane count a launch with these parameters really runs with (lanes_per_chain, or the default above and its small-launch
 * rule, which looks at the current device); 0 on bad arguments */
int32_t mcq_effective_lanes(const mcq_params* p);
/* SIMDs of the current device (4 per compute unit; 1024 when no device answers): what the small-launch rule compares with */
int32_t mcq_device_simds(void);

/* bytes of one chain's state record in best_state / final_state; 0 on bad arguments.  mcq_state_bytes: Q = N*N queens;
 * mcq_state_bytes_for: the parameter block's own count (n_queens). */
size_t mcq_state_bytes(int32_t N, int32_t mode);
size_t mcq_state_bytes_for(const mcq_params* p);

/* bytes of device scratch mcq_run_device needs for these parameters; 0 on bad arguments */
size_t mcq_workspace_bytes(const mcq_params* p);

/*
 * Replaces run_experiment's fan-out + per-chain sweep (experiments.py:507-546) with
 * device-resident buffers.  `seeds` (uint32[n_chains]) and every non-NULL pointer of
 * `out` are DEVICE pointers; `workspace` is a 64-byte aligned device buffer of at least
 * mcq_workspace_bytes(p) bytes (hipMalloc results are).  Violations return MCQ_EINVAL.  Work is enqueued on `hip_stream` (a hipStream_t, NULL =
 * the default stream) and the call returns without synchronising.
 */
int mcq_run_device(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out,
                   void* workspace, size_t workspace_bytes, void* hip_stream);

/*
 * mcq_run_device plus HIP events recorded on `hip_stream` around the init kernel and around the
 * sweep kernel; waits for the last event and returns both device times in milliseconds.  Blocking.
 */
int mcq_run_device_timed(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, void* workspace,
                         size_t workspace_bytes, void* hip_stream, float* init_ms, float* sweep_ms);

/*
 * Statistics of a device-resident trace (what plot_energy_histories, experiments.py:593-595, and
 * plot_acceptance_rates_binned, experiments.py:660-695, consume), so that the trace never crosses PCIe.
 * `out` holds the DEVICE buffers a previous mcq_run_device filled.  All result arrays are device pointers:
 *   step_sum / step_sumsq / step_count  int64[n_steps + 1]: over the chains whose history reaches that entry
 *                                       (count < n_chains only after early stops); pass NULL to skip
 *   bin_lo  int64[n_bins + 1]: first step of every bin (ascending, bin_lo[n_bins] >= n_steps closes the last);
 *   bin_accepted / bin_proposed uint64[n_bins]: accepted / executed steps of all chains per bin; n_bins = 0 to skip
 * Enqueued on `hip_stream`; asynchronous.
 */
int mcq_trace_stats_device(const mcq_params* p, const mcq_outputs* out, int64_t* step_sum, int64_t* step_sumsq,
                           int64_t* step_count, int32_t n_bins, const int64_t* bin_lo, uint64_t* bin_accepted,
                           uint64_t* bin_proposed, void* hip_stream);

/*
 * The node-level summary of a finished launch, packed for the ONE all-reduce of a job list (the layout of the host side's
 * distributed.py): replaces the per-run gathering of run_experiment (experiments.py:519-546) and the statistics the drivers take from it
 * (experiments.py:1074-1096) with one or two small kernels per launch, where the host side needed ~10 tensor operations per job.
 * One mcq_pack_slot per schedule set of the launch (n_sets <= 1: one) says where that job's fields sit in `packed` (word offsets; -1 = absent).
 * The set's chains on this rank are [t * chains_per_set, t * chains_per_set + n_local).  `packed` must have been zeroed; the call WRITES the
 * counters and slots of its jobs (other ranks' shares arrive by the all-reduce) and adds to the stopped-chain histogram.
 * `out` holds the DEVICE buffers mcq_run_device filled (hist_len, steps_executed, best_energy, steps_to_best, n_accepted; with a stats offset
 * the four step_* arrays of trace == REDUCED).  Enqueued on `hip_stream`; asynchronous.
 */
typedef struct mcq_pack_slot {
    int64_t counters; /* 6 words: chains, accepted, proposed, sum of best_energy, sum of its squares, sum of steps_to_best */
    int64_t min_slot; /* this rank's slot of the per-rank minima: min(best_energy) + 1 (0 = the rank holds no chain of the job) */
    int64_t best;     /* first of the n_local per-chain slots of best_energy (already offset to this rank's first chain), or -1 */
    int64_t stb;      /* the same for steps_to_best, or -1 */
    int64_t stats;    /* first of 5 x (n_steps + 1) words: per-entry sum, sum of squares, accepted, count, chains that stopped early at the entry; or -1 */
} mcq_pack_slot;
int mcq_pack_summary_device(const mcq_params* p, const mcq_outputs* out, int64_t n_local, const mcq_pack_slot* slots /* HOST, one per set */,
                            int64_t* packed /* DEVICE */, void* hip_stream);

/*
 * The beta(step) table the sweep reads (experiments.py:13-77 evaluated on the device in float64, strict IEEE):
 * `beta_out` is a DEVICE buffer of double[n_sets][n_steps] (n_sets <= 1: [n_steps]); `c32_out` (optional, DEVICE,
 * float, same shape) receives (float)(-beta * log2(e)), the factor of the float32 accept bracket.  Inspection /
 * testing only: mcq_run_device computes its own tables.  Enqueued on `hip_stream`; asynchronous.
 */
int mcq_beta_table_device(const mcq_params* p, double* beta_out, float* c32_out, void* hip_stream);

/*
 * Same computation with HOST buffers: allocates device memory, uploads seeds, runs,
 * downloads every non-NULL output and frees.  Blocking.  `kernel_seconds` (optional)
 * receives the device time of init + sweep measured with HIP events.
 */
int mcq_run_host(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out,
                 double* kernel_seconds);

/* ---- exported by libmcq_oracle.so (tests / smoke / cpu_baseline only) --------------------- */

/* CPU restatement of the reference; host buffers; n_threads <= 1 runs chains in the calling thread. */
int mcq_oracle_run(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, int n_threads);
/* the same chains with O(1) dE from per-line occupancy counters: the "best CPU" baseline of bench.py; equals mcq_oracle_run */
int mcq_oracle_run_fast(const mcq_params* p, const uint32_t* seeds, const mcq_outputs* out, int n_threads);
/* one Philox-4x32-10 block: ctr[4], key[2] -> out[4] (known-answer tests) */
int mcq_oracle_philox_block(const uint32_t* ctr, const uint32_t* key, uint32_t* out);
const char* mcq_oracle_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* MCQ_H */
