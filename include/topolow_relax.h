/* include/topolow_relax.h -- C ABI of libtopolow_relax.so (MI355X / gfx950, HIP).
 *
 * Drop-in boundary for ONE path of omid-arhami/topolow v2.1.0: the native relaxation kernel
 * `optimize_layout_exact_cpp` (reference src/optimization.cpp:109-382) that
 * `euclidean_embedding()` reaches through
 *     .Call(`_topolow_optimize_layout_exact_cpp`, <16 args>)        (reference R/RcppExports.R:4-6,
 *                                                                    src/RcppExports.cpp:16-49)
 * plus the dense post-metric `as.matrix(dist(positions))` (reference R/core.R:474).
 *
 * Plain pointers and sizes only; no torch / R / C++ types.  Matrices are laid out as R lays
 * them out (column-major), so an R `.Call` shim or a ctypes/cgo/JNI stub can pass its buffers
 * straight through.  Every function returns 0 on success or a TOPOLOW_ERR_* code and writes a
 * message into errbuf (when given).  Inputs are never modified.
 */
#ifndef TOPOLOW_RELAX_H
#define TOPOLOW_RELAX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TOPOLOW_OK 0
#define TOPOLOW_ERR_TOO_FEW_POINTS 1 /* "Need at least 2 points for embedding" (reference :131) */
#define TOPOLOW_ERR_NONFINITE 2      /* "Numerical instability at iteration %d. ..." (:359-361) */
#define TOPOLOW_ERR_BAD_ARGUMENT 3
#define TOPOLOW_ERR_NO_DEVICE 4      /* no usable HIP device: there is NO CPU fallback */
#define TOPOLOW_ERR_HIP 5            /* a HIP runtime call failed; message has the call */
#define TOPOLOW_ERR_UNSUPPORTED 6
#define TOPOLOW_ERR_INTERRUPTED 7    /* the caller's interrupt callback asked to stop */

/* Parity statement.  The reference visits the pairs of an iteration in std::shuffle order seeded from
 * std::random_device (src/optimization.cpp:153-154,196): it cannot be reproduced run for run, its own test
 * accepts relative 1e-2 between two runs (tests/testthat/test-deprecated.R:65-67).  The CPU oracle the tests compare
 * with (oracle/, a restatement of src/optimization.cpp:109-382) is pinned by the results the reference itself ships
 * (tests/test_reference_results.py, data under tests/golden/ref_results/): the edge MAE of the reference's own H3N2
 * and HIV embeddings lies inside the oracle's 64-run distribution of the same call (0.59241 vs 0.58806 +- 0.0037;
 * 1.22454 vs 1.21109 +- 0.0121), those embeddings are rest points of the oracle's relaxation (error moves 0.05 %),
 * 3 x 48 likelihood_function() calls recorded in the reference's chains (H3N2, HIV, DENV) are reproduced to +1.3 % /
 * -0.2 % / +1.7 % in the mean (one call scatters by 1.5-2.4 %), its 20 per-fold CV errors to within 3 standard errors,
 * the mean, sd and quartiles of its pooled signed out-of-sample errors in sign and size.  What this library guarantees,
 * and tests (tests/test_gpu_contract.py against >= 20 oracle seeds per problem committed under tests/golden/;
 * tests/test_gpu_reference_results.py against the reference-held numbers directly):
 *   GS    the reference's arithmetic pair by pair in f64, in round-robin tournament order; the CPU oracle
 *         replaying that order agrees to <= 1e-12.  Final-MAE mean inside the oracle's
 *         mean +- max(3 sd, 1 %) on every pinned problem up to 1500 points and at config 3;
 *         stated band at N = 2048: 3 % (measured +1.8 % +- 0.6 %).
 *   SLAB  (AUTO above gs_max_n) the reference's per-pair update, applied row-owner style in Jacobi
 *         stages over random labels, fp32.  Final-MAE mean inside the oracle's
 *         mean +- max(3 sd, 1 %) on every pinned problem (N = 1500 ... 10 000, ndim 2 ... 5, thresholds,
 *         relative_epsilon 1e-4 ... 1e-10); run-to-run scatter: robust sd <= 2 sd_oracle + 0.5 %, at most 15 % of the runs
 *         further than max(4 sd_oracle, 3 %) from the oracle's mean (plain sd 0.9-1.8 x the oracle's, 3.3 x at
 *         N = 2048: DESIGN.md section 2b); the MAE it reports is the reference's edge MAE of the positions
 *         it returns to 2e-5; stop iteration within max(3 sd, 10 %) of the oracle's except on 2-D data
 *         (+55 %, same MAE).  Iterations that are ONE stage (k <= 3) of an fp32 problem with ndim 2..6 and
 *         >= 7168 points run as a symmetric sweep (csrc/relax_symm.h): the same update -- every
 *         point moved by the sum of its own halves of all its pairs at the positions the previous iteration
 *         left -- with each pair's distance and factor computed once; checked against a CPU model of that
 *         iteration in f64 (positions: mean 5e-5, max 5e-3 of the displacement scale; the fused check's MAE against
 *         the oracle's edge error to 2e-5: tests/test_gpu_symmetric.py) and against the row-owner sweep (fp32
 *         summation order only: 2e-6 per iteration, same stop iteration and final MAE to 1e-6 on whole runs);
 *         a row-sharded run shards it over its sessions (same band against one block); f64 sessions take an f64 form
 *         of it (csrc/relax_symm64.h; equal to the f64 CPU model to 1e-12 per iteration; its fused check is exact --
 *         the sweep also reads exact-minus-rounded target differences -- and equals the reference's edge MAE to 1e-11);
 *         TOPOLOW_SYMMETRIC=0 / TOPOLOW_SHARD_SYMMETRIC=0 switch it off.  Where the sweep applies, 2-, 4- and 8-stage
 *         iterations whose stages leave a resident wave >= 5 tiles (config 3: the two-stage ones, 3 < k <= 6) run as S
 *         symmetric sweeps that split the PAIRS -- stage st: the pairs between slabs a and b of the randomly labelled
 *         points with (a + b) mod S == st, stages in random order -- instead of "all points against one slab of the
 *         columns, S times": every point still meets one slab of partners per stage, both ends of a pair move together as
 *         in the reference; equal to a CPU model of that schedule (f64: 1e-12 per stage), final-MAE statistics unchanged
 *         on the pinned problems and seed for seed (profiles/r03_two_stage_study.txt); TOPOLOW_SYMMETRIC_TWO_STAGE=0
 *         keeps the row-owner form.
 * The deterministic pieces -- controller, cooling, error rule, guards, messages -- are exact. */

/* Schedules (topolow_options.schedule). */
#define TOPOLOW_SCHEDULE_AUTO 0   /* GS tournament for n <= gs_max_n, slab above */
#define TOPOLOW_SCHEDULE_SLAB 1   /* S-stage row-owner slabs, HBM-bound, multi-workgroup */
#define TOPOLOW_SCHEDULE_GS 2     /* exact Gauss-Seidel, round-robin tournament order, one workgroup */

/* Arithmetic of positions and pair updates (topolow_options.precision). */
#define TOPOLOW_PRECISION_AUTO 0  /* f64 for the GS kernel, f32 for the slab kernel */
#define TOPOLOW_PRECISION_F32 1
#define TOPOLOW_PRECISION_F64 2

typedef struct topolow_options {
  uint64_t seed;        /* seeds the pair-order / slab-order stream (the reference uses
                           std::random_device, src/optimization.cpp:153-154) */
  int32_t schedule;     /* TOPOLOW_SCHEDULE_* */
  int32_t precision;    /* TOPOLOW_PRECISION_* */
  int32_t slab_stages;  /* 0 = adaptive: topolow_slab_stages_at(iteration, k, ndim) */
  int32_t device;       /* HIP device ordinal; -1 = current device */
  int32_t gs_max_n;     /* AUTO switches to the slab schedule above this n; 0 = default */
  int32_t n_devices;    /* > 1 (or a non-NULL `devices`): ONE embedding row-block sharded over
                           devices[0..n_devices) -- see topolow_optimize_layout_exact_sharded */
  int32_t keep_labels;  /* 0 (default): the multi-workgroup schedules relabel the points at random (seeded
                           by `seed`), so that a slab / a tile is a random subset of the points rather
                           than a run of consecutive ones; non-zero: keep the caller's order */
  int32_t reserved[3];
  /* Polled every 50 iterations like Rcpp::checkUserInterrupt() in the reference
   * (src/optimization.cpp:364); a non-zero return abandons the run with TOPOLOW_ERR_INTERRUPTED
   * after device memory has been released.  NULL = never. */
  int32_t (*interrupt_cb)(void* user);
  void* interrupt_user;
  /* Sink of the `verbose` lines (the reference prints them with Rcpp::Rcout,
   * src/optimization.cpp:183-188,298-301,334-336,351-353): called with one NUL-terminated line
   * (newline included) at a time, on the calling thread.  NULL = stdout.  The R shim maps it to
   * Rprintf so the lines obey sink() and the console. */
  void (*print_cb)(const char* line, void* user);
  void* print_user;
  const int32_t* devices;   /* n_devices HIP ordinals; an ordinal may repeat (several row blocks on
                               one GPU).  NULL with n_devices > 1: ordinals 0..n_devices-1 */
} topolow_options;

/* Run statistics, filled by topolow_optimize_layout_exact when `stats` is non-NULL. */
typedef struct topolow_run_stats {
  int32_t schedule_used;
  int32_t precision_used;
  int32_t iterations_run;   /* iterations executed before stopping (>= `iterations` out) */
  int32_t n_checks;
  double  device_seconds;   /* relaxation loop only, device resident */
  double  total_seconds;    /* including upload/encode/download */
  int64_t stage_launches;
  double  setup_seconds;    /* session creation, verification of the inputs, upload, encode */
  int64_t reserved[3];
} topolow_run_stats;

void topolow_default_options(topolow_options* opt);

/* Replaces the payload of `_topolow_optimize_layout_exact_cpp` (reference
 * src/RcppExports.cpp:16-39 -> src/optimization.cpp:109-126), argument for argument:
 *   initial_positions     n x ndim  float64, column-major           (NumericMatrix)
 *   dissimilarity_matrix  n x n     float64, column-major, +Inf = unmeasured, symmetric
 *   threshold_matrix      n x n     int32,   column-major, 0 exact / 1 ">" / -1 "<"
 *   degrees               n         int32    (non-NA cells per row, R/core.R:341)
 *   edge_i, edge_j        n_edges   int32, 0-based, i<j   } upper-triangle measured pairs,
 *   edge_dist             n_edges   float64               } used by the convergence MAE only
 *   edge_thresh           n_edges   int32                 } (src/optimization.cpp:54-81)
 *   n_iter, k0, cooling_rate, c_repulsion, relative_epsilon, convergence_window,
 *   convergence_check_freq, verbose  -- as in the reference.
 * Outputs (the reference's returned list, src/optimization.cpp:375-381):
 *   positions_out n x ndim float64 column-major; converged (0/1); iterations (= best
 *   iteration); final_mae (= best MAE); final_k (= k at the best iteration).
 * opt may be NULL (defaults).  stats may be NULL.
 */
int topolow_optimize_layout_exact(
    const double* initial_positions, int32_t n, int32_t ndim,
    const double* dissimilarity_matrix, const int32_t* threshold_matrix,
    const int32_t* degrees,
    const int32_t* edge_i, const int32_t* edge_j, const double* edge_dist,
    const int32_t* edge_thresh, int64_t n_edges,
    int32_t n_iter, double k0, double cooling_rate, double c_repulsion,
    double relative_epsilon, int32_t convergence_window, int32_t convergence_check_freq,
    int32_t verbose, const topolow_options* opt,
    double* positions_out, int32_t* converged, int32_t* iterations, double* final_mae,
    double* final_k, topolow_run_stats* stats, char* errbuf, size_t errlen);

/* Batch form: `count` independent embeddings in one call -- one workgroup per embedding on the
 * exact Gauss-Seidel kernel (the reference's only parallel mode is one embedding per forked
 * process: R/adaptive_sampling.R:666,1301; its CV evaluator `likelihood_function`,
 * R/adaptive_sampling.R:2552-2726, runs `folds` such embeddings per parameter set).  Problems may
 * differ in every field, ndim included.  A problem that fails (e.g. non-finite positions) gets
 * its own error_code / message-free status; the call itself still returns TOPOLOW_OK. */
typedef struct topolow_problem {
  const double* initial_positions;     /* n x ndim, column-major */
  const double* dissimilarity_matrix;  /* n x n, +Inf = unmeasured; NULL together with           */
  const int32_t* threshold_matrix;     /* n x n   threshold_matrix: the edge list IS the matrix   */
  const int32_t* degrees;              /* n */
  const int32_t* edge_i;
  const int32_t* edge_j;
  const double* edge_dist;
  const int32_t* edge_thresh;
  int64_t n_edges;
  int32_t n, ndim, n_iter, convergence_window, convergence_check_freq, reserved0;
  double k0, cooling_rate, c_repulsion, relative_epsilon;
  uint64_t seed;
  /* Optional held-out pairs (0-based, any orientation) with their true dissimilarities: scored on
   * the returned positions, sum |truth - distance| -- the OutSampleError of the reference's
   * error_calculator_comparison (R/error_metrics.R:55-144) without materialising est_distances. */
  const int32_t* holdout_i;
  const int32_t* holdout_j;
  const double* holdout_truth;
  int64_t n_holdout;
} topolow_problem;

typedef struct topolow_result {
  double* positions_out;               /* n x ndim, column-major, caller-allocated */
  double final_mae, final_k;
  int32_t converged, iterations, iterations_run, n_checks;
  int32_t error_code;                  /* TOPOLOW_OK or TOPOLOW_ERR_NONFINITE */
  int32_t error_iteration;             /* iteration reported by the non-finite guard */
  double holdout_sum_abs;              /* sum |truth - distance| over the holdout pairs */
  int64_t holdout_count;
} topolow_result;

int topolow_optimize_layout_exact_batch(const topolow_problem* problems, topolow_result* results,
                                        int32_t count, int32_t precision, int32_t device,
                                        double* device_seconds, char* errbuf, size_t errlen);

/* Host-side helper of the batched CV evaluator: one fold's problem from the list of the full
 * matrix's non-NA cells, i.e. what the reference obtains per fold by masking the held-out cells
 * (R/adaptive_sampling.R:2600-2640) and re-running euclidean_embedding's pre-processing
 * (R/core.R:269-436) on the n x n matrix.  No device work.  Cells are listed in column-major order
 * (as R's which()); `pos_of` maps a linear column-major index to its cell or -1; `by_row`/`row_ptr`
 * list the same cells row by row (columns ascending).  `picks`: held-out linear indices (their
 * mirrors are held out too).  Outputs are caller-allocated: order n (order[0] = -1: input order
 * kept), degrees n, edges and holdout up to n_cells entries; `numeric_max` = largest value among the
 * remaining unprefixed cells (the scale of the random-walk start, R/core.R:407-415). */
typedef struct topolow_cell_list {
  int32_t n, reserved;
  int64_t n_cells;
  const int32_t* row;
  const int32_t* col;
  const double* value;      /* threshold prefix stripped */
  const int32_t* code;      /* 0 none, 1 ">", -1 "<" */
  const int64_t* pos_of;    /* n * n */
  const int64_t* by_row;    /* n_cells */
  const int64_t* row_ptr;   /* n + 1 */
} topolow_cell_list;

/* Fills the index arrays of a cell list from its (row, col) columns, listed in column-major order:
 * pos_of (n*n, -1 where no cell), by_row (n_cells), row_ptr (n+1).  No device work. */
int topolow_cell_list_index(int32_t n, int64_t n_cells, const int32_t* row, const int32_t* col,
                            int64_t* pos_of, int64_t* by_row, int64_t* row_ptr);

int topolow_cv_fold(const topolow_cell_list* cells, const int64_t* picks, int64_t n_picks,
                    int32_t preserve_order, int32_t named, int32_t* order, int32_t* degrees,
                    int32_t* edge_i, int32_t* edge_j, double* edge_dist, int32_t* edge_thresh,
                    int64_t* n_edges, int32_t* holdout_i, int32_t* holdout_j, double* holdout_truth,
                    int64_t* n_holdout, double* numeric_max);

/* A whole cross-validation sweep in one call (the consumer of the hot path: the reference's likelihood_function,
 * R/adaptive_sampling.R:2552-2726, folds x parameter sets times).  Fold f holds out the cells picks[picks_offset[f] ..
 * picks_offset[f + 1]) (linear column-major indices, as topolow_cv_fold) and runs with ndim[f], k0[f], ...; its start
 * positions are the reference's random walk (R/core.R:407-415) built from unit_draws[draws_offset[f] ..], the
 * ndim[f] x (n - 1) uniform(0, 1) numbers, row-major, that the caller drew at this point of its stream.  The folds'
 * problems are built on host threads and relaxed as ONE batch (topolow_optimize_layout_exact_batch); per fold only
 * the out-of-sample score comes back: sum |truth - distance| and count over its held-out numeric cells, the best
 * iteration and the converged flag; error_code[f] = TOPOLOW_ERR_BAD_ARGUMENT for a fold without valid measurements,
 * TOPOLOW_ERR_NONFINITE for a diverged one. */
int topolow_cv_sweep(const topolow_cell_list* cells, int32_t named, int32_t preserve_order, int32_t n_folds,
                     const int32_t* ndim, const double* k0, const double* cooling_rate, const double* c_repulsion,
                     const int64_t* picks, const int64_t* picks_offset, const double* unit_draws,
                     const int64_t* draws_offset, const uint64_t* seeds, int32_t n_iter, double relative_epsilon,
                     int32_t convergence_window, int32_t convergence_check_freq, int32_t precision, int32_t device,
                     double* holdout_sum_abs, int64_t* holdout_count, int32_t* iterations, int32_t* converged,
                     int32_t* error_code, double* device_seconds, char* errbuf, size_t errlen);


/* Replaces `as.matrix(stats::dist(positions))` (reference R/core.R:474):
 * positions n x ndim float64 column-major (host) -> est_distances n x n float64 (host). */
int topolow_est_distances(const double* positions, int32_t n, int32_t ndim,
                          double* est_distances, int32_t device, char* errbuf, size_t errlen);
/* Rows [row_begin, row_end) of the same matrix: out is (row_end - row_begin) x n, row-major (= rows
 * row_begin.. of the symmetric n x n result).  For problems whose n x n float64 result should not be
 * held at once (BASELINE config 4: 20 GB): the caller streams row blocks.  Either entry keeps device
 * memory bounded (tiles of <= 256 MB). */
int topolow_est_distances_rows(const double* positions, int32_t n, int32_t ndim, int32_t row_begin,
                               int32_t row_end, double* out, int32_t device, char* errbuf, size_t errlen);

/* ---------------------------------------------------------------------------------------
 * Device-resident session: the same relaxation with inputs kept in HBM, for callers that
 * run many iterations / many embeddings on data they already hold on the GPU (bench.py, the
 * row-sharded multi-GPU driver).  Pointers named d_* are DEVICE pointers.
 * ------------------------------------------------------------------------------------- */
typedef struct topolow_session topolow_session;

/* Creates a session for rows [row_begin, row_end) of an n-point problem (single GPU:
 * row_begin = 0, row_end = n).  The session owns an encoded fp32 target block of
 * (row_end-row_begin) x ld floats (see topolow_session_encoded_ld). */
int topolow_session_create(topolow_session** out, int32_t n, int32_t ndim, int32_t row_begin,
                           int32_t row_end, int32_t precision, int32_t device, char* errbuf,
                           size_t errlen);
void topolow_session_destroy(topolow_session* s);

/* Optional, before anything is loaded: store the points in a random order (a permutation drawn from
 * `seed`; 0 = the caller's order).  A slab -- a run of consecutive session labels -- is then a random
 * subset of the caller's points, not a run of points that lie next to each other on the reference's
 * random-walk start (R/core.R:407-415).  Every entry point that takes or returns HOST arrays keeps
 * speaking the caller's labels; device buffers handed to topolow_session_stage / _edge_error /
 * _check_partial are in session labels (topolow_session_labels gives the map).  Sessions that share
 * n and seed share the permutation (row-sharded runs). */
int topolow_session_set_relabel(topolow_session* s, uint64_t seed, char* errbuf, size_t errlen);
/* session label q holds the caller's point session_to_caller[q] (n entries). */
int topolow_session_labels(const topolow_session* s, int32_t* session_to_caller);

/* Encode the reference's dense inputs (host pointers, R layout) into the session's HBM
 * block.  Only rows [row_begin,row_end) are read. */
int topolow_session_load_dense(topolow_session* s, const double* dissimilarity_matrix,
                               const int32_t* threshold_matrix, const int32_t* degrees,
                               char* errbuf, size_t errlen);
/* COO entry for problems too large for dense host matrices (BASELINE config 4): edges are
 * the upper-triangle measured pairs (host pointers); degrees as above. */
int topolow_session_load_coo(topolow_session* s, const int32_t* edge_i, const int32_t* edge_j,
                             const double* edge_dist, const int32_t* edge_thresh,
                             int64_t n_edges, const int32_t* degrees, char* errbuf,
                             size_t errlen);
/* Device-side fill: the caller writes the session's encoded block itself (a device buffer of
 * (row_end-row_begin) x ld uint32 words, row-major: the word of (row r of the block, column c) sits at
 * topolow_encoded_index(r, c, ld) = r * ld + c; word = topolow_encode_target(); diagonal, padding
 * and unmeasured cells = topolow_encode_target(+Inf, 0)), then commits it with the degrees. */
void* topolow_session_encoded_ptr(topolow_session* s);
int32_t topolow_session_encoded_ld(const topolow_session* s);
int topolow_session_commit_encoded(topolow_session* s, const int32_t* degrees, char* errbuf,
                                   size_t errlen);
/* Edge list used by the convergence MAE (host pointers).  For a row-sharded session pass
 * only the edges this rank should reduce (e.g. those with edge_i in its row block). */
int topolow_session_set_edges(topolow_session* s, const int32_t* edge_i, const int32_t* edge_j,
                              const double* edge_dist, const int32_t* edge_thresh,
                              int64_t n_edges, char* errbuf, size_t errlen);
/* Positions: n x ndim float64 column-major host buffer. */
int topolow_session_set_positions(topolow_session* s, const double* positions, char* errbuf,
                                  size_t errlen);
int topolow_session_get_positions(topolow_session* s, double* positions, char* errbuf,
                                  size_t errlen);

/* Starts a run: resets the controller (best = DBL_MAX, k = k0, iteration 0). */
int topolow_session_begin(topolow_session* s, int32_t n_iter, double k0, double cooling_rate,
                          double c_repulsion, double relative_epsilon,
                          int32_t convergence_window, int32_t convergence_check_freq,
                          uint64_t seed, int32_t slab_stages, char* errbuf, size_t errlen);
/* Enqueues up to max_iters iterations (whole check intervals) on the session stream and
 * returns without waiting; *enqueued = iterations enqueued (0 when the run is over). */
int topolow_session_enqueue(topolow_session* s, int32_t max_iters, int32_t* enqueued,
                            char* errbuf, size_t errlen);
/* Waits for the launches enqueued so far on the session's streams and nothing else: a convergence check that is
 * waiting to ride on the next iteration's sweep (one-stage iterations reduce the pending check's MAE on the way) stays
 * pending, as it would in an uninterrupted run; no state is read back.  For callers that pace a run in slices
 * (bench.py) and go on enqueueing. */
int topolow_session_wait(topolow_session* s, char* errbuf, size_t errlen);
/* Waits for everything enqueued, a pending check included (it runs as a separate pass); reports progress. */
int topolow_session_sync(topolow_session* s, int32_t* iterations_run, int32_t* stopped,
                         double* last_mae, char* errbuf, size_t errlen);
/* Restores the best snapshot and returns the reference's result fields. */
int topolow_session_finish(topolow_session* s, double* positions_out, int32_t* converged,
                           int32_t* iterations, double* final_mae, double* final_k,
                           char* errbuf, size_t errlen);
/* The convergence checks of the current run so far: 3 doubles per check (iteration, MAE, k after
 * cooling) -- the values behind the reference's verbose lines (src/optimization.cpp:298-301).
 * Waits for the enqueued work.  *n_checks = checks recorded; at most max_checks are copied. */
int topolow_session_check_trace(topolow_session* s, double* out, int32_t max_checks, int32_t* n_checks);
/* Per-kernel timing for roofline accounting: while enabled, every slab-stage launch and
 * every convergence check is bracketed by HIP events on the session stream.
 * topolow_session_profile waits for the stream, returns the summed durations (ms) and launch
 * counts since profiling was enabled, and resets them. */
int topolow_session_set_profiling(topolow_session* s, int32_t enable);
/* Of the stage launches recorded so far, those that also reduced a convergence check's MAE (one-stage
 * iterations: the check is fused into the next iteration's sweep) -- summed duration (ms) and count.  Does
 * not reset anything: ask before topolow_session_profile, whose stage figures include these launches. */
int topolow_session_profile_fused(topolow_session* s, double* fused_ms, int64_t* fused_launches, char* errbuf,
                                  size_t errlen);
/* Iterations that ran as a symmetric sweep + apply (csrc/relax_symm.h) since profiling was enabled: their HIP-event
 * time, the plain ones and those whose sweep also reduced a check's MAE apart.  Not included in the two calls above. */
int topolow_session_profile_symmetric(topolow_session* s, double* plain_ms, int64_t* plain_iterations, double* fused_ms,
                                      int64_t* fused_iterations, char* errbuf, size_t errlen);
int topolow_session_profile(topolow_session* s, double* stage_ms, int64_t* stage_launches,
                            double* check_ms, int64_t* checks, char* errbuf, size_t errlen);
/* external != 0: the session launches on the caller's stream `hip_stream` (a hipStream_t; NULL
 * is the device's default stream, which is what torch uses unless told otherwise), so its
 * kernels are ordered with the caller's copies and collectives; the caller keeps ownership.
 * external == 0: back to the session's private non-blocking stream. */
int topolow_session_set_stream(topolow_session* s, void* hip_stream, int32_t external);
/* HIP stream (hipStream_t) the session launches on, for event timing by the caller. */
void* topolow_session_stream(topolow_session* s);
/* 1 when the convergence MAE is reduced from the encoded block (the edge list was verified to be
 * exactly its measured cells), 0 when it gathers the caller's edge list. */
int32_t topolow_session_uses_dense_mae(const topolow_session* s);
/* Number of slab-stage kernel launches so far, and the algorithmic bytes one iteration
 * moves (4*rows*n + 8*n*ndim + 4*n). */
int64_t topolow_session_stage_launches(const topolow_session* s);
int64_t topolow_session_bytes_per_iteration(const topolow_session* s);

/* ---------------------------------------------------------------------------------------
 * Row-sharded building blocks (one process per GPU; the all-gather between stages is the
 * caller's, e.g. torch.distributed/RCCL).  d_pos_* are device pointers to n x ndim
 * row-major positions in the session's precision.
 * ------------------------------------------------------------------------------------- */
/* Iteration body of the session: TOPOLOW_SCHEDULE_SLAB (default) or TOPOLOW_SCHEDULE_GS = exact
 * Gauss-Seidel across workgroups in tile-tournament order (whole-problem sessions only). */
int topolow_session_set_schedule(topolow_session* s, int32_t schedule);
/* Visiting order of the tile Gauss-Seidel schedule for iteration `iter` (as topolow_gs_pair_order). */
int64_t topolow_tilegs_pair_order(int32_t n, uint64_t seed, int32_t iter, int32_t* pairs_out);
/* Rows a caller-owned position buffer must hold: roundup4(n).  Rows [n, roundup4(n)) are the
 * phantom points of the padding columns and must be (1e18, 0, ..., 0) (1e150 in f64 sessions) in
 * BOTH ping-pong buffers; stages never write them. */
int32_t topolow_session_position_rows(const topolow_session* s);
/* Coordinates per point in such a buffer: ndim up to 10, 12 for ndim 11 and 12, 16 for ndim 13..16 (the
 * kernels are instantiated for 1..10, 12 and 16 coordinates; extra coordinates must be, and stay, zero). */
int32_t topolow_session_position_dim(const topolow_session* s);
/* Launches stage `stage` of iteration `iter` (0-based) for the session's row block: reads
 * all n positions from d_pos_in, writes rows [row_begin,row_end) of d_pos_out. */
int topolow_session_stage(topolow_session* s, const void* d_pos_in, void* d_pos_out,
                          int32_t iter, int32_t stage, int32_t n_stages, double k,
                          char* errbuf, size_t errlen);
/* The multi-process form of the convergence check (one process per GPU, the caller owns the
 * collectives): check_partial reduces this block's share of the measured pairs on d_pos into two
 * doubles (sum, count) at the DEVICE address d_out2; the caller all-reduces them (RCCL) and hands the
 * totals to controller_step, which runs the reference's controller (src/optimization.cpp:303-357) on
 * the device -- snapshot, counters, stop flag -- exactly as the session's own loop does.  Nothing
 * here waits for the device.  Valid between topolow_session_begin and topolow_session_finish. */
int topolow_session_check_partial(topolow_session* s, const void* d_pos, double* d_out2, char* errbuf,
                                  size_t errlen);
int topolow_session_controller_step(topolow_session* s, const double* d_total2, const void* d_pos,
                                    int32_t iter1, double k_after, char* errbuf, size_t errlen);
/* The fused form of a check for one-stage iterations: the ONE stage of iteration `iter` (0-based), which
 * also reduces this block's share of the convergence MAE of the positions it READS (d_pos_in: the previous
 * iteration's result, i.e. the previous iteration's check) into the two doubles at d_out2 -- no separate
 * pass over the block.  The caller all-reduces d_out2 and calls controller_step with d_pos_in and the
 * previous iteration's number.  topolow_session_can_fuse_checks: 1 when the session supports it (fp32,
 * MAE reduced from the block, even row count). */
int32_t topolow_session_can_fuse_checks(const topolow_session* s);
int topolow_session_stage_fused(topolow_session* s, const void* d_pos_in, void* d_pos_out, int32_t iter,
                                double k, double* d_out2, char* errbuf, size_t errlen);
/* ONE-stage iterations of a row-sharded run as the SYMMETRIC sweep sharded over the processes (one per GPU; fp32,
 * ndim 2..6, >= 7168 points: csrc/relax_symm.h).  Every unordered pair is visited once instead of twice (reference
 * src/optimization.cpp:198-283 visits each pair once and moves both ends), so a rank reads half the bytes of its
 * row-owner sweep.  Rank r of P owns SEGMENT r of the tile list of the upper triangle (equal tile counts), not a
 * row block, so the caller first brings the rows that hold the segment's tiles together:
 *   topolow_symm_segment_rows   host only: rows [*row_first, *row_end) of the matrix hold segment `segment` of
 *                               `n_segments`; returns 0 when a problem of n points has too few tiles to cut
 *   topolow_session_degree_terms  device float[n]: degree + 1 per point (reference :137-140); a row-block session
 *                               knows its own rows' -- the caller completes the array over the ranks (all-gather)
 *   topolow_session_symm_segment_build  d_rows: device words of those rows, n_rows x topolow_session_encoded_ld(s),
 *                               row-major as in the owners' blocks (topolow_session_encoded_ptr; the caller moved
 *                               them, e.g. RCCL send/recv); any_threshold: some rank's block holds threshold codes.
 *                               Builds the tile-major copy of the segment and the sweep's buffers; d_rows may be
 *                               freed on return.  TOPOLOW_ERR_UNSUPPORTED when the session cannot take the path.
 * and then, per one-stage iteration `iter` (0-based), between topolow_session_begin and _finish:
 *   topolow_session_symm_segment_sweep  reads all n positions of d_pos_in, sweeps the segment and leaves the
 *                               segment's share of every point's move in the session's moves buffer
 *                               (topolow_session_symm_moves: device float[n][ndim]); d_out2 != NULL: also the
 *                               segment's share of the MAE of d_pos_in, as topolow_session_stage_fused does
 *   (the caller sums the moves buffer over the ranks in place -- ONE all-reduce of n x ndim floats, 600 KB at
 *    BASELINE config 4 -- and d_out2 as for stage_fused)
 *   topolow_session_symm_segment_apply  d_pos_out[i] = d_pos_in[i] + moves[i] for ALL n points: every rank holds
 *                               the same sums, so every rank holds the same positions and no gather follows.
 * Nothing here waits for the device (build does, once). */
int32_t topolow_symm_segment_rows(int32_t n, int32_t segment, int32_t n_segments, int32_t* row_first,
                                  int32_t* row_end);
/* 1 when the session can take the path cut into n_segments (fp32 slab schedule, ndim 2..6, size gate, targets loaded). */
int32_t topolow_session_symm_segment_eligible(const topolow_session* s, int32_t n_segments);
float* topolow_session_degree_terms(topolow_session* s);
/* 1 when some target of the session's block carries a threshold code ('>' / '<'): the any_threshold of the ranks is
 * the OR of these. */
int32_t topolow_session_has_thresholds(const topolow_session* s);
int topolow_session_symm_segment_build(topolow_session* s, int32_t segment, int32_t n_segments, const void* d_rows,
                                       int32_t row_first, int32_t n_rows, int32_t any_threshold, char* errbuf,
                                       size_t errlen);
float* topolow_session_symm_moves(topolow_session* s);
int topolow_session_symm_segment_sweep(topolow_session* s, const void* d_pos_in, int32_t iter, double k,
                                       double* d_out2, char* errbuf, size_t errlen);
int topolow_session_symm_segment_apply(topolow_session* s, const void* d_pos_in, void* d_pos_out, int32_t iter,
                                       char* errbuf, size_t errlen);
/* First iteration (1-based) at which one of this block's rows became non-finite, 0 = none. Waits. */
int topolow_session_first_nonfinite(topolow_session* s, int32_t* iteration);
/* Partial edge error of this session's edge list on d_pos: (sum, count). Synchronous. */
int topolow_session_edge_error(topolow_session* s, const void* d_pos, double* sum,
                               int64_t* count, char* errbuf, size_t errlen);

/* ---------------------------------------------------------------------------------------
 * ONE embedding row-sharded over several GPUs from ONE process (BASELINE config 4: the R host is a
 * single process, reference src/RcppExports.cpp:16-39).  Block b owns a contiguous row block of the
 * encoded matrix and moves its own points; a stage kernel stores its updated rows straight into the
 * position buffers of all blocks (peer stores over xGMI), blocks meet at HIP-event barriers, and the
 * convergence controller is replicated from per-block (sum, count) partials -- no host round trip
 * per stage or per check (csrc/relax_sharded_engine.h).  Results equal the one-session run of the
 * same seed: positions bit for bit, the MAE to rounding (its partial sums are grouped by block) -- on
 * the multi-stage iterations and wherever the row-owner kernel runs.  ONE-stage iterations of fp32 runs with
 * ndim 2..6 and >= 7168 points run as the symmetric sweep sharded over the sessions (csrc/relax_symm.h): session b
 * sweeps segment b of the tile list of the upper triangle (equal tile counts; gathered once from all row blocks),
 * folds its partials per point and stores them into the inbox of the session that owns the point; behind the
 * barrier the owners move their points and store them into every session's positions (two barriers per
 * iteration).  Against one block: positions to the fp32 summation band (2e-5 of the coordinate scale per iteration),
 * every check's MAE to 2e-6, same verdicts (tests/test_gpu_sharded_native.py); TOPOLOW_SHARD_SYMMETRIC=0 keeps
 * the row-owner sweeps.
 * ------------------------------------------------------------------------------------- */
typedef struct topolow_shard_stats {
  int32_t blocks, iterations_run, n_checks;
  int32_t groups;                /* host threads / streams: blocks that share a GPU share one */
  double loop_seconds;           /* host wall time of the relaxation loop, all blocks */
  double total_seconds;          /* including session creation, upload, encode, download */
  double stage_kernel_seconds;   /* first GPU: summed durations of its blocks' stage kernels (HIP events) */
  double check_kernel_seconds;   /* first GPU: error passes + partial exchange + controllers */
  int64_t stage_launches;        /* block 0 */
  int64_t exchanges;             /* stage and check boundaries (an event barrier when groups > 1) */
  /* Measurement aid, IN and out (topolow_sessions_run_sharded only): when > 0 on entry, the GPUs are
   * drained once these iterations have run and `timed_seconds` covers the rest of the loop -- the
   * steady state without the 16-stage iterations of the unfolding phase. */
  int32_t warmup_iterations;
  int32_t symmetric_segments;    /* > 0: one-stage iterations ran as the symmetric sweep sharded over this many sessions */
  double timed_seconds;
  int64_t reserved[2];
} topolow_shard_stats;

/* Row block `block` of `blocks` over n rows: contiguous, whole 8-row workgroups; returns how many
 * blocks hold at least one row (<= blocks; trailing blocks of a small problem are empty and
 * dropped).  block < 0: only the count. */
int32_t topolow_shard_rows(int32_t n, int32_t blocks, int32_t block, int32_t* row_begin, int32_t* row_end);

/* The `.Call` payload row-sharded over opt->devices[0 .. n_devices) (an ordinal may repeat: several
 * row blocks on one GPU).  Arguments and outputs as topolow_optimize_layout_exact;
 * dissimilarity_matrix and threshold_matrix may both be NULL: the edge list then IS the matrix (large
 * problems never build the dense n x n host matrices).  Slab schedule only. */
int topolow_optimize_layout_exact_sharded(
    const double* initial_positions, int32_t n, int32_t ndim,
    const double* dissimilarity_matrix, const int32_t* threshold_matrix,
    const int32_t* degrees,
    const int32_t* edge_i, const int32_t* edge_j, const double* edge_dist,
    const int32_t* edge_thresh, int64_t n_edges,
    int32_t n_iter, double k0, double cooling_rate, double c_repulsion,
    double relative_epsilon, int32_t convergence_window, int32_t convergence_check_freq,
    int32_t verbose, const topolow_options* opt,
    double* positions_out, int32_t* converged, int32_t* iterations, double* final_mae,
    double* final_k, topolow_shard_stats* stats, char* errbuf, size_t errlen);

/* The same run over sessions the caller has created and loaded itself (targets, degrees and each
 * block's share of the MAE edges -- pair {lo, hi} belongs to the block that owns lo when lo + hi is
 * even and hi when it is odd): row blocks that tile [0, n) in order, slab schedule, one precision.
 * profile != 0: block 0's kernels are bracketed by timing events (fills the *_kernel_seconds). */
int topolow_sessions_run_sharded(topolow_session** sessions, int32_t count, const double* initial_positions,
                                 int32_t n_iter, double k0, double cooling_rate, double c_repulsion,
                                 double relative_epsilon, int32_t convergence_window,
                                 int32_t convergence_check_freq, uint64_t seed, int32_t slab_stages,
                                 int32_t (*interrupt_cb)(void* user), void* interrupt_user, int32_t profile,
                                 double* positions_out, int32_t* converged, int32_t* iterations,
                                 double* final_mae, double* final_k, topolow_shard_stats* stats,
                                 char* errbuf, size_t errlen);

/* ---------------------------------------------------------------------------------------
 * Host-side helpers exported for tests and integrators (no GPU needed).
 * ------------------------------------------------------------------------------------- */
/* Slab plan of iteration `iter`: writes n_stages x 4 int32 (r0_begin, r0_end, r1_begin,
 * r1_end; second range empty unless the slab wraps) in execution order; returns n_stages. */
int32_t topolow_slab_plan(int32_t n, int32_t slab_stages, uint64_t seed, int32_t iter,
                          int32_t* ranges_out, int32_t max_stages);
/* Stage count the adaptive policy picks for spring constant k in ndim dimensions: the smallest power of two
 * with k / stages <= min(3, ndim) (a stage is a Jacobi step, stable for k / stages < 2 ndim). */
int32_t topolow_slab_stages_for_k(double k, int32_t ndim);
/* 2-, 4- and 8-stage iterations of sessions that take the symmetric sweep (fp32 / f64 slab schedule, ndim 2..6, >= 7168
 * points, whole matrix) run as symmetric sweeps that split the PAIRS: the (randomly labelled) points are cut into S slabs
 * of labels [first_label[q], first_label[q + 1]) and stage st sweeps the pairs between slabs a and b with
 * (a + b) mod S == st, so every point meets one slab of partners per stage, as in the row-owner form, and both ends of a
 * pair move in the same stage, as in the reference.  topolow_symm_stage_bounds: the S + 1 slab boundaries (0 when the
 * problem is too small or S is not 2, 4, 8); topolow_symm_stage_order: the order of the stages in iteration `iter`.
 * TOPOLOW_SYMMETRIC_TWO_STAGE=0 keeps the row-owner stages. */
int32_t topolow_symm_stage_bounds(int32_t n, int32_t stages, int32_t* first_label);
int32_t topolow_symm_stage_order(uint64_t seed, int32_t iter, int32_t stages, int32_t* order);
/* Stage count of iteration `iter` (0-based) when slab_stages = 0: the policy above, and at least 16 stages
 * during the first 8 iterations, while the layout unfolds from its start. */
int32_t topolow_slab_stages_at(int32_t iter, double k, int32_t ndim);
/* Visiting order of the GS tournament schedule for iteration `iter`: n(n-1)/2 pairs
 * (a,b) as 2 int32 each, in an order equivalent to what the kernel executes. */
int64_t topolow_gs_pair_order(int32_t n, uint64_t seed, int32_t iter, int32_t* pairs_out);
/* fp32 target encoding used in HBM: value with the 2 low mantissa bits replaced by the
 * threshold code (0 exact, 1 ">", 2 "<", 3 skip); +Inf = unmeasured. */
uint32_t topolow_encode_target(double dissimilarity, int32_t threshold_code);
int64_t topolow_encoded_index(int32_t row_in_block, int32_t column, int32_t ld);
double topolow_decode_target(uint32_t bits, int32_t* threshold_code);
/* Convergence controller (reference src/optimization.cpp:303-357) on a scripted MAE
 * sequence; same code the device runs. */
int topolow_controller_script(const double* mae_seq, const int32_t* iter_seq,
                              const double* k_seq, int32_t n_obs, double k0, int32_t window,
                              double eps, int32_t* stopped_at_obs, int32_t* snapshot_flags,
                              double* best_mae, double* best_k, int32_t* best_iter);
const char* topolow_relax_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TOPOLOW_RELAX_H */
