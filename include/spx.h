/*
 * spx.h -- C ABI of libspx.so, the MI355X (gfx950) GP-EI engine behind the
 * Spearmint chooser plugin API.
 *
 * Boundary.  The reference has no FFI: its choosers
 * call numpy/scipy in-process.  This library replaces exactly the block
 *
 *     for mcmc_iter in range(mcmc_iters): overall_ei[:, i] = compute_ei(...)
 *     best_cand = argmax(mean(overall_ei, axis=1))
 *
 * of  spearmint/spearmint/chooser/GPEIChooser.py:143-153,
 *     GPEIOptChooser.py:331-341 (+ :269-271, :293-294),
 *     GPEIperSecChooser.py:284-302,
 * i.e. per hyper-parameter draw:  gp.Matern52/dist2 (gp.py:34-54,120-127),
 * chooser cov (GPEIChooser.py:117-122), spla.cholesky (:191), cho_solve (:194),
 * solve_triangular (:195), predictive mean/variance (:198-199), EI (:202-206).
 * The Python choosers in spearmint_amd/chooser/ bind it with ctypes
 * (see INTEGRATION.md for the stub a maintainer would add to the reference).
 *
 * Conventions: plain C, caller-owned buffers, row-major float64, int64 sizes.
 * No torch types.  Every function returns an int status: 0 = ok, <0 = error
 * (spx_last_error() gives the text; after a SUCCESSFUL call it may hold a text that
 * starts with "warning:" -- see spx_get_stat).  No exceptions or exit() cross the ABI.
 * One handle = one GPU = one HIP stream; calls on a handle are serialized by
 * the caller.  All calls are synchronous unless stated.
 *
 * Hyper-parameter row layout, everywhere: [mean, noise, amp2, ls[0..D)]
 * (the tuple order of GPEIOptChooser.py:628), H rows of (3 + D) doubles.
 */
#ifndef SPX_H
#define SPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spx_handle spx_handle;

/* status codes */
#define SPX_OK            0
#define SPX_ERR_ARG      -1   /* bad argument / call order                     */
#define SPX_ERR_HIP      -2   /* HIP runtime error (no device, OOM, ...)        */
#define SPX_ERR_NOT_PD   -3   /* covariance not positive definite == the
                                 numpy.linalg.LinAlgError spla.cholesky raises
                                 (GPEIChooser.py:191); see spx_not_pd_info()     */

/* flags for spx_ei_run / spx_ei_grid */
#define SPX_FLAG_PER_SEC     1   /* divide EI by exp(predicted log duration)
                                    (GPEIperSecChooser.py:437-491); needs
                                    spx_set_time_model()                         */
#define SPX_FLAG_KEEP_MOMENTS 2  /* keep func_m / func_v per (draw, candidate)
                                    for spx_get_moments() (test / debug)          */
#define SPX_FLAG_TIMING      4   /* bracket every kernel launch with HIP events
                                    on the handle's stream (spx_get_timings)      */
#define SPX_FLAG_TIME_ONLY   8   /* with PER_SEC | KEEP_MOMENTS: only the log-duration GP's predicted durations are
                                    computed (spx_get_time_mean); no EI, no winner.  The per-second chooser's pending
                                    branch needs exactly that from its first pass (GPEIperSecChooser.py:492-548: the two
                                    GPs have different observation sets there)                                      */

/* ---- lifetime ---------------------------------------------------------- */
/* Create an engine on HIP device `device_id` (lazy: the first call that needs
 * the GPU initialises it, so the handle can be created before a fork).       */
int  spx_create(int device_id, spx_handle** out);
/* One handle over n_dev GPUs of this node, for the single-process driver:
 * every call below works on it unchanged.  Candidate rows are sharded contiguously over the
 * devices, observations / hyper draws are replicated, each device is driven by its own host
 * thread, and spx_ei_run ends with the path's single collective -- one ncclAllGather (RCCL over
 * xGMI, communicator from ncclCommInitAll) of a 16-byte {best mean EI, global index} record per
 * device, followed by the same numpy-argmax reduction on every device.  Results are bit-identical
 * to a one-GPU handle (np.argmax(np.mean(overall_ei, axis=1)), GPEIChooser.py:153).
 * spx_gp_logprob shards its draws and spx_ei_grad_batch its points over the devices.
 * Repeated device ids (several engines on one GPU: test configuration) cannot form an RCCL
 * communicator; the records then go through host memory (SPX_TRANSPORT_HOST).  librccl is
 * loaded when this is first called, not when libspx is.                                        */
int  spx_create_multi(const int* device_ids, int32_t n_dev, spx_handle** out);
/* The same with the transport of the records chosen by the caller instead of by the device list: SPX_TRANSPORT_NONE =
 * as spx_create_multi decides (RCCL for distinct devices, host memory for repeated ids); SPX_TRANSPORT_HOST = host memory
 * even between distinct GPUs; SPX_TRANSPORT_RCCL = the RCCL code path even for repeated ids (a real librccl refuses such
 * a communicator -- for a stand-in library named by SPX_RCCL_LIB, tests/test_gpu_d2_fake_rccl.py).
 * Environment read by the library (deployment hooks, nothing else): SPX_RCCL_LIB = the librccl to dlopen;
 * SPX_RCCL_ANY_VERSION=1 waives the ncclGetVersion range check of the hand-declared binding.                        */
int  spx_create_multi_transport(const int* device_ids, int32_t n_dev, int32_t transport, spx_handle** out);
#define SPX_TRANSPORT_NONE 0   /* single-GPU handle: no collective                */
#define SPX_TRANSPORT_RCCL 1   /* ncclAllGather over the devices' streams          */
#define SPX_TRANSPORT_HOST 2   /* records staged through host memory               */
/* n_dev, transport and (up to cap) device ids of a handle; any pointer may be NULL */
int  spx_multi_query(spx_handle* h, int32_t* n_dev, int32_t* transport, int32_t* device_ids, int32_t cap);
/* One process per GPU (a launcher-based run, bench.py --gpus N): attach an RCCL communicator to a
 * single-GPU handle.  Rank 0 obtains an id (ncclGetUniqueId), the launcher's own channel carries its
 * SPX_COMM_ID_BYTES bytes to the other ranks, every rank attaches (ncclCommInitRank).  From then on
 * spx_ei_run ends with the path's collective -- one ncclAllGather of the ranks' 16-byte {best mean EI,
 * global index} records on the handle's stream and the same numpy-argmax reduction on every rank -- and
 * spx_get_best returns the global winner.  Candidates are sharded by the caller (spx_set_candidates'
 * index_base).                                                                                     */
#define SPX_COMM_ID_BYTES 128
int  spx_comm_unique_id(char* id_out /* SPX_COMM_ID_BYTES */);
/* librccl is bound at run time (dlopen; SPX_RCCL_LIB overrides the name) through hand-declared types, so the loaded
 * library's ncclGetVersion is checked before anything else of it is called: NCCL API 2.10 <= version < 3.0 (tested
 * with RCCL 2.27.7); anything else, or a library without ncclGetVersion, is refused with SPX_ERR_HIP and a message
 * naming the version (SPX_RCCL_ANY_VERSION=1 waives the range).  Returns the loaded library's version code
 * (22707 = 2.27.7) in *version_code; loads the library if it is not loaded yet.                                */
int  spx_rccl_version(int32_t* version_code);
int  spx_comm_attach(spx_handle* h, const char* id, int32_t nranks, int32_t rank);
/* Optional 2-D partition, "hypers x candidates" (SURVEY.md 8(e)): P = hyper_shards x P_c devices / ranks, device
 * r = rc * hyper_shards + rh evaluates the draws of hyper shard rh for the candidates of shard rc, and the path's
 * single collective becomes ONE ncclAllReduce(SUM) of the zero-padded M-vector of per-candidate EI sums on the
 * handle's stream (8 M bytes; the sums over a device's own draws are formed on the device in numpy's order), after
 * which every device divides by the number of draws and takes numpy's argmax over ALL candidates: no per-draw EI
 * leaves the device, no host-side reduction.  Each device then factors H / hyper_shards covariances instead of H.
 * The sum over draws is "local sums, then the reduction tree", so means agree with the candidates-only scheme to
 * ~1e-16 relative (near-ties may resolve differently): that scheme stays the default.
 *   multi-device handle: spx_set_partition(h, hyper_shards, 0, 0) -- hyper_shards must divide n_dev; the library
 *     shards draws and candidates itself on the following spx_set_hypers / spx_set_candidates (call it first).
 *     spx_set_fantasies and spx_ei_grad_batch are not available with hyper_shards > 1.
 *   one process per GPU (spx_comm_attach): the caller sets ITS draw shard (spx_set_hypers) and candidate shard
 *     (spx_set_candidates with index_base) and passes the totals; M_total > 0 switches the collective of spx_ei_run
 *     from the all-gather of records to the all-reduce of sums (hyper_shards = 1 is allowed: candidates only, but
 *     every rank ends up with the whole mean vector); spx_get_ei_mean then returns the global means of the handle's
 *     own candidates.  M_total = 0 switches back.                                                              */
int  spx_set_partition(spx_handle* h, int32_t hyper_shards, int64_t M_total, int32_t H_total);
void spx_destroy(spx_handle* h);
const char* spx_last_error(void);
int  spx_version(void);
/* number of visible HIP devices, or <0 */
int  spx_device_count(void);

/* ---- resident-data API (what bench.py times: inputs already in HBM) ----- */
/* comp = grid[complete,:] (N x D), vals = values[complete] (N)
 * -- GPEIChooser.py:135-138.                                                  */
int spx_set_observations(spx_handle* h, const double* comp, const double* vals,
                         int64_t N, int32_t D);
/* cand = grid[candidates,:] (M x D) -- GPEIChooser.py:136.  `index_base` is
 * added to local candidate indices in results (a rank that owns candidate
 * rows [base, base+M) of a sharded grid passes base).                         */
int spx_set_candidates(spx_handle* h, const double* cand, int64_t M, int32_t D,
                       int64_t index_base);
/* H hyper-parameter draws, rows [mean, noise, amp2, ls...] */
int spx_set_hypers(spx_handle* h, const double* hypers, int32_t H);
/* second GP on log durations (GPEIperSecChooser.py:176, :437-458):
 * log_durs (N), time_hypers H x (3+D).  Pass NULLs to clear.                  */
int spx_set_time_model(spx_handle* h, const double* log_durs,
                       const double* time_hypers);

/* Hot path, stage 1: for every draw build K(X,X)+noise (gp.py:34-54,120-127;
 * GPEIChooser.py:186,190), factor it (:191), and form what the solves need
 * (:194).  Returns SPX_ERR_NOT_PD like spla.cholesky raising LinAlgError.     */
int spx_factor(spx_handle* h);
/* Pending-experiment fantasies (GPEIChooser.py:209-266; "next" row 2 of SURVEY 8(f)).
 * Call after spx_set_observations was given comp_pend = [comp; pend] (n = N + P rows; the
 * vals argument is then only a placeholder) and spx_factor has run:
 *   fant  H x n x S, per draw row-major [i][s]: fant_vals of :245-246 (tile(vals) on top of
 *         pend_fant);   bests H x S: np.min(fant_vals, axis=0) (:249).   1 <= S <= 4096.
 * The next spx_ei_run scores every candidate against every fantasy (:253-263) and averages
 * over S in numpy's summation order (:265).  NULL / S = 0 clears; so does any call that
 * invalidates the factorisation.                                                      */
int spx_set_fantasies(spx_handle* h, const double* fant, const double* bests, int32_t S);

/* Hot path, stage 2: K(X*,X) (:187), triangular solve (:195), predictive
 * mean/variance (:198-199), EI (:202-206) for every (candidate, draw); then the
 * MCMC mean and the argmax (:153).  Results stay on the device.               */
int spx_ei_run(spx_handle* h, int32_t flags);
/* spx_factor followed by spx_ei_run as ONE call with ONE host synchronisation: the body of a chooser's next() for fixed
 * hyper-parameter draws (GPEIChooser.py:143-153: compute_ei per draw, argmax of the mean) and bench.py's timed step.
 * Same kernels and results as the two calls; a covariance that is not positive definite is reported as spx_factor
 * reports it (SPX_ERR_NOT_PD + spx_not_pd_info) once the step's single synchronisation has passed.  Fantasies set
 * before the call are dropped, as by spx_factor.  What it saves matters at small N (N = 128, 20 000 candidates, 10
 * draws: two synchronisations + three device-to-host copies were ~85 us of a 310 us step).                        */
int spx_ei_step(spx_handle* h, int32_t flags);

/* results of the last spx_ei_run / spx_ei_step */
/* best_idx = index_base + argmax_c mean_h EI[c,h]  with numpy's rule (first NaN
 * wins, else first maximum); best_val = that mean EI.                         */
int spx_get_best(spx_handle* h, int64_t* best_idx, double* best_val);
int spx_get_ei_mean(spx_handle* h, double* out /* M */);
/* overall_ei exactly as the reference lays it out: M x H, row-major           */
int spx_get_ei_draws(spx_handle* h, double* out /* M*H */);

/* ---- one-shot convenience: host buffers in, results out ------------------ */
/* == ei_over_hypers + argmax(mean) (GPEIOptChooser.py:331-341, :294).
 * ei_mean_out (M) and ei_draw_out (M x H) may be NULL.                         */
int spx_ei_grid(spx_handle* h,
                const double* comp, const double* vals, int64_t N, int32_t D,
                const double* cand, int64_t M,
                const double* hypers, int32_t H, int32_t flags,
                double* ei_mean_out, double* ei_draw_out,
                int64_t* best_idx, double* best_val);
/* == GPEIperSecChooser.ei_over_hypers with all draws evaluated (:284-302) */
int spx_ei_per_sec_grid(spx_handle* h,
                const double* comp, const double* vals, const double* log_durs,
                int64_t N, int32_t D, const double* cand, int64_t M,
                const double* hypers, const double* time_hypers, int32_t H,
                int32_t flags, double* ei_mean_out, double* ei_draw_out,
                int64_t* best_idx, double* best_val);

/* ---- hyper-parameter slice sampling (the caller of spx_gp_logprob) -----------------------
 * The reference's sample_hypers (spearmint/spearmint/chooser/GPEIChooser.py:268-346; GPEIOptChooser.py:621-706;
 * GPEIperSecChooser.py:558-700) on util.slice_sample (spearmint/spearmint/util.py:34-93): per iteration ONE joint
 * random-direction move over [mean, amp2, noise] and ONE component-wise sweep over the length scales, every
 * log-probability a covariance build + Cholesky + solve.  spx_sample_hypers runs `n_iter` such iterations for the
 * resident observations (spx_set_observations) inside the library: the control flow of the sampler, the priors, the
 * speculative batching of its evaluations (each batch = one spx_gp_logprob call) and numpy's legacy random stream
 * (MT19937: rand / randn / shuffle as numpy.random.RandomState produces them) are host C++ -- no interpreter between
 * two GPU calls.  It is the SAME Markov chain as the reference's: the same points are accepted and the generator is
 * left in the same state, draw for draw (tests/test_sampler_native.py pins it to the reference's golden trace).
 *
 *   cfg        which model is sampled and how deep the speculation goes (below)
 *   rng        numpy.random.get_state() in, the state to numpy.random.set_state() out
 *   hyper_io   [mean, noise, amp2, ls[0..D)] -- the chain's current point in, its last point out
 *   rows_out   n_iter rows [mean, noise, amp2, ls...]: the point after every iteration (NULL: not wanted)
 *   hist_io    12 doubles of bracket statistics the speculation learns from (zeros to start; keep between calls)
 *   stats_out  SPX_SAMPLER_NSTATS values: {spx_gp_logprob calls, hyper rows evaluated, slice moves, moves that needed no
 *              call, iterations completed, then calls by number of hyper rows: [5 + r] for r = 0 .. 32, [38] beyond, [39] nanoseconds
 *              inside the log-likelihood calls, [40] nanoseconds in all} (NULL ok)
 *
 * Errors (the iteration that failed is left as the reference leaves it: a finished joint move is applied, an
 * unfinished sweep is not; `rng` is where the reference's generator would be; rows_out holds the iterations done;
 * stats_out[4] = iterations completed):
 *   SPX_ERR_NOT_PD      a covariance the sampler really evaluated was not positive definite (spla.cholesky raises)
 *   SPX_ERR_SLICE_NAN   "Slice sampler got a NaN"          (util.py:59-61)
 *   SPX_ERR_SLICE_ZERO  "Slice sampler shrank to zero!"    (util.py:68-69)                                            */
#define SPX_SAMPLER_NSTATS 41
#define SPX_ERR_SLICE_NAN  -4
#define SPX_ERR_SLICE_ZERO -5
typedef struct spx_rng_state {      /* numpy.random.get_state(): ('MT19937', key, pos, has_gauss, cached_gaussian)     */
    uint32_t key[624];
    int32_t  pos;
    int32_t  has_gauss;
    double   gauss;
} spx_rng_state;
typedef struct spx_sampler_cfg {
    int32_t D;                    /* input dimensions = number of length scales                                        */
    int32_t n_iter;               /* iterations: (joint move, length-scale sweep) each                                   */
    int32_t noiseless;            /* 1: noise pinned to 1e-3 (GPEIChooser.py:270,326)                                    */
    int32_t check_mean;           /* 1: -inf for a mean outside [vals_min, vals_max] (GPEIChooser.py:289-290)            */
    int32_t amp2_prior_on_sqrt;   /* log-normal prior on sqrt(amp2) (GPEIOptChooser.py:668) instead of amp2 (:312)       */
    int32_t lookahead;            /* step-out points per side and shrink proposals evaluated per call (>= 1)             */
    int32_t follow_props;         /* cross-move speculation: proposals planned for the NEXT coordinate's move ...        */
    int32_t follow_hyps;          /* ... under the hypotheses "this move accepts its 1st .. follow_hyps-th proposal"     */
    int32_t max_rows;             /* hyper rows per spx_gp_logprob call the speculation may fill (<= 32 stays one launch) */
    double  noise_scale;          /* horseshoe prior on the noise   (GPEIChooser.py:309)                                 */
    double  amp2_scale;           /* log-normal prior on the amplitude                                                  */
    double  max_ls;               /* top-hat prior on the length scales (GPEIChooser.py:279)                             */
    double  vals_min, vals_max;   /* min / max of the observed values (the mean's support)                               */
} spx_sampler_cfg;
int spx_sample_hypers(spx_handle* h, const spx_sampler_cfg* cfg, spx_rng_state* rng, double* hyper_io,
                      double* rows_out, double* hist_io /* 12 */, int64_t* stats_out /* SPX_SAMPLER_NSTATS */);
/* The same sampler on a caller-supplied log-likelihood: fn(ctx, rows[n_rows][3 + D], n_rows, lp_out[n_rows]) returns
 * 0 and the data term -sum log diag L - 0.5 r'K^-1 r per row (-inf = not positive definite).  No handle, no GPU:
 * how the CPU tests hold the sampler to the reference's chain, and how a host evaluator can be plugged in.  The
 * choosers never use it -- their sampler is spx_sample_hypers on the GPU.                                              */
typedef int (*spx_logprob_fn)(void* ctx, const double* rows, int32_t n_rows, double* lp_out);
int spx_sample_hypers_with(spx_logprob_fn fn, void* ctx, const spx_sampler_cfg* cfg, spx_rng_state* rng,
                           double* hyper_io, double* rows_out, double* hist_io, int64_t* stats_out);
/* numpy's legacy generator, for tests: n_rand x rand(), then n_randn x randn(), then a shuffle of range(n_shuffle)
 * (outputs may be NULL when their count is 0).                                                                         */
int spx_rng_draw(spx_rng_state* rng, int32_t n_rand, double* rand_out, int32_t n_randn, double* randn_out,
                 int32_t n_shuffle, int32_t* shuffle_out);


/* ---- building blocks (per-stage parity tests, "next" rows) --------------- */
/* After spx_factor: K + noise I (N x N, full symmetric), its lower Cholesky
 * factor L (N x N, strict upper = 0) and alpha = K^-1 (vals - mean) (N) of
 * draw `draw` (time model: draw + H).  Any pointer may be NULL.               */
int spx_get_factor(spx_handle* h, int32_t draw, double* K, double* L, double* alpha);
/* Rows [row0, row0 + nrows) of L (nrows x N, row-major, zeros above the diagonal) and gamma = L^-1 (vals - mean) (N) of one
 * draw; either pointer may be NULL.  The pending branch (GPEIChooser.py:219-249) needs only the bottom P rows of the factor
 * of cov([comp; pend]) and gamma: pend_m = L21 gamma[:N] + mean, pend_K = L_S L_S^T - noise I (L21 = rows N.., columns < N;
 * L_S the trailing P x P block) -- a P x (N + P) block per draw instead of the whole factor.                          */
int spx_get_factor_rows(spx_handle* h, int32_t draw, int64_t row0, int64_t nrows, double* L_rows, double* gamma);
/* K(X*,X) of draw `draw` for candidates [c0, c0+nc): N x nc row-major.         */
int spx_get_cross_cov(spx_handle* h, int32_t draw, int64_t c0, int64_t nc, double* out);
/* func_m, func_v (M each) of draw `draw`; needs SPX_FLAG_KEEP_MOMENTS.          */
int spx_get_moments(spx_handle* h, int32_t draw, double* func_m, double* func_v);
/* exp(predicted log duration) of every candidate under time draw `draw`
 * (func_time_m, GPEIperSecChooser.py:452-458); needs SPX_FLAG_PER_SEC | SPX_FLAG_KEEP_MOMENTS. */
int spx_get_time_mean(spx_handle* h, int32_t draw, double* out /* M */);
/* GP marginal log-likelihood data term  -sum(log diag L) - 0.5 r' K^-1 r  for
 * each resident draw (GPEIChooser.py:281-285): out has H entries, -inf where
 * the covariance is not PD.  Needs spx_set_observations + spx_set_hypers.      */
int spx_gp_logprob(spx_handle* h, double* out);
/* Objective of the local refinement (GPEIOptChooser.py:360-525 grad_optimize_ei_over_hypers;
 * "next" row 3) at P points in ONE call -- the reference runs grid_subset (20) L-BFGS-B problems,
 * one objective evaluation at a time (:265-291): points (P x D) -> neg_ei[p] = the summed
 * negative EI over the resident draws, grad (P x D) its gradient, in the reference's scaling (its
 * grad_xp carries a factor one half).  A point's result does not depend on the other points of
 * the call.  Needs spx_factor or a previous spx_ei_grid.  With fantasies set (spx_set_fantasies;
 * the pending branch :441-525) EI and gradient are averaged over the S fantasies per draw; with
 * a factored time model (spx_set_time_model / spx_ei_per_sec_grid) the objective is EI per second
 * (GPEIperSecChooser.py:349-434; not combinable with fantasies, as in the reference).          */
int spx_ei_grad_batch(spx_handle* h, const double* points, int32_t P, double* neg_ei /* P */,
                      double* grad /* P x D */);
/* == spx_ei_grad_batch with P = 1 */
int spx_ei_grad(spx_handle* h, const double* point, double* neg_ei_sum, double* grad /* D */);
/* Sobol candidate grid on the device (ExperimentGrid.py:192-196 -> sobol_lib.py:125-157
 * i4_sobol_generate; "next" row 4): grid (n x dim, row-major) = transpose(i4_sobol_generate(dim,
 * n, skip)), bit-identical to the reference.  dirs = its scaled direction integers V[d][b]
 * (dim_max x 30 uint32, host; spearmint_amd/data/sobol_dirs_*.npy), dim <= dim_max,
 * skip + n - 2 < 2^30.  grid_out (host) may be NULL.  as_candidates != 0 also leaves the grid
 * resident as the candidate set (== spx_set_candidates(grid, n, dim, 0) without the host copy).
 * kernel_ms (may be NULL) receives the HIP-event duration of the generating kernel.          */
int spx_sobol_grid(spx_handle* h, const uint32_t* dirs, int32_t dim_max, int32_t dim, int64_t n,
                   int64_t skip, double* grid_out, int32_t as_candidates, double* kernel_ms);
/* which draw / pivot failed in the last SPX_ERR_NOT_PD                         */
int spx_not_pd_info(spx_handle* h, int32_t* draw, int32_t* pivot);

/* ---- measurement ---------------------------------------------------------- */
/* Per-kernel accumulated HIP-event time of the last spx_factor + spx_ei_run
 * executed with SPX_FLAG_TIMING.  Fills up to n entries of ms[] / launches[]
 * in the order of spx_timing_name(i); returns the number of stages.            */
int spx_get_timings(spx_handle* h, double* ms, int64_t* launches, int n);
/* Counters a caller can poll (single-GPU handles; "ranks_seen" also on a multi-device handle):
 *   "flow_fallbacks"   in-launch hand-off time-outs of the one-launch factorisation (k_lean_flow) so far.  The call
 *                      that saw one is repeated with one launch per block column (same bits) and returns SPX_OK with
 *                      a WARNING left in spx_last_error() (text starts with "warning:"); the handle then keeps that
 *                      form for "flow_rearm_after" clean factorisations (option, default 16; 0 = for good) and goes
 *                      back to the one-launch form by itself -- or at once on spx_set_option("lean_flow", 1).  Never
 *                      seen on a healthy device; bounded polls make it an error path instead of a hang;
 *   "flow_rearms"      times the handle went back to the one-launch form after a fallback;
 *   "flow_enabled"     1 while the one-launch factorisation is in use;   "n_cu"  compute units of the device;
 *   "last_step_fused"  1 if the last EI pass ran as a one-kernel form (no K* / beta in memory; no fantasies);
 *   "last_step_skipped_padding"  1 if the last EI pass left the padding of N (to the GEMM's 128-row tiles) uncomputed
 *                      (option "gemm_partial", default on: same bits, up to -31 % per pass just above a multiple of 128);
 *   "obs_dims"         D of the resident observations (0: none set);   "hip_runtime_version", "hip_driver_version", "clock_khz",
 *                      "mem_clock_khz", "wall_clock_khz", "l2_bytes", "mem_bus_bits": what the process runs on (bench.py: platform);
 *   "ranks_seen"       size of the communicator the last exchange ran on (one record per rank in its table): the ranks of the attached communicator
 *                      (spx_comm_attach), the device slots of a multi-device handle, 1 otherwise.                  */
int spx_get_stat(spx_handle* h, const char* name, int64_t* value);
const char* spx_timing_name(int i);
/* The correlation function of the GP -- the choosers' covar= argument, a function of
 * spearmint/spearmint/gp.py selected by name (GPEIChooser.py:52): option "covar" of a handle.
 * Every entry point (K, K*, log-likelihood, EI grid, EI gradient) follows it; changing it
 * invalidates the factorisation.                                                             */
#define SPX_COVAR_MATERN52 0   /* gp.Matern52 (gp.py:120-127), the default                     */
#define SPX_COVAR_MATERN32 1   /* gp.Matern32 (gp.py:107-113)                                   */
#define SPX_COVAR_ARDSE    2   /* gp.ARDSE    (gp.py:95-100)                                    */
#define SPX_COVAR_SE       3   /* gp.SE       (gp.py:87-93): ARDSE with the length scales ignored */
/* options: "covar" (SPX_COVAR_*); tuning knobs "kstar_budget_bytes" (K(X*,X) staging buffer;
 * 0 = default), "streams" (1|2), "timing" (0|1), "gemm_waves" (predict-GEMM variant of THIS
 * handle; values the build does not contain are rejected with SPX_ERR_ARG), and the forms of the
 * factorisation, all bit-identical, -1 = the default / chosen from the sizes:
 *   "lean_flow"     1 (default): the whole factorisation of spx_gp_logprob is ONE data-flow launch
 *                   (k_lean_flow: every dependency a hand-off inside the launch); 0: one launch per
 *                   block column, in the forms selected by "lean_ps" / "lean_lazy";
 *   "ei_flow"       the same choice for spx_factor (default 1);
 *   "lean_flow_cu"  k_lean_flow with one workgroup per CU (1) or two (0); -1: by size;
 *   "lean_flow_yield" with two per CU: a workgroup yields while its neighbour on the CU factors a
 *                   diagonal block (1, default) or does not (0);
 *   "lean_flow_cov" k_lean_flow builds the tiles of K(X,X) itself (1, default) or reads k_cov's (0);
 *   "lean_lazy"     trailing updates one (0) or two (1) block columns at a time;
 *   "lean_merge"    spx_gp_logprob: observation scaling and the right-hand-side rows in one launch (1, default) or two (0);
 *   "lean_ps"       1: the panel solve of a block column runs inside the update launch, handed the
 *                   inverse of the diagonal block behind its pivots; 0: a launch of its own.
 * If an in-launch hand-off ever times out (its polls are bounded; never observed), the call is
 * repeated with one launch per block column, a warning is left in spx_last_error(), and the handle
 * stays in that form for "flow_rearm_after" clean factorisations (see spx_get_stat).
 *   "gemm_partial"      N not a multiple of 128: the last 128-row block of the predict GEMM computes only the 16-row tiles
 *                       that hold observations (k_predict_gemm_tail) and K(X*,X) does not write the pad rows (1, default,
 *                       where it saves at least 12 % of the pass), or everything is computed on the padded size (0);
 *   "flow_rearm_after"  clean factorisations before the one-launch form is tried again (default 16, 0 = never);
 *   "flow_spin_limit"   polls a waiting workgroup makes before it gives up (0 = default, 2^20); tests set 1.
 * A kernel that is refused the dynamic LDS it asks for (hipFuncSetAttribute) makes the call fail with
 * SPX_ERR_HIP and a message naming the kernel and the size.                                       */
int spx_set_option(spx_handle* h, const char* name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* SPX_H */
