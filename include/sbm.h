/*
 * sbm.h -- C ABI of libsbm_hip.so, the MI355X (gfx950) implementation of the
 * parameter-fitting inner loop of FedericoV/SysBio_Modeling.
 *
 * Every entry point replaces one Python-level interface of the reference; the
 * reference has no FFI of its own (pure Python 2 + SciPy), so the "binding a
 * maintainer would add" is a ctypes stub -- see INTEGRATION.md.  Citations are
 * relative to the reference tree.
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs; no C++ or torch types.
 *   - every function returns 0 on success, <0 on error; sbm_last_error() gives
 *     the message (thread-local).  No exception crosses this boundary.
 *   - the caller owns every buffer.  "_dev" arguments are device pointers valid
 *     on the context's device; the library borrows them for the call only.
 *     Calls are enqueued on the context's stream and return without waiting
 *     (call sbm_ctx_synchronize); the *_host variants take host pointers, stage
 *     through library scratch and return when the results are in host memory.
 *   - all reals are float64, row-major; statuses / counters are int32.
 *   - one context per device per thread of use; contexts are independent.
 */
#ifndef SBM_H
#define SBM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBM_ABI_VERSION 4   /* 3: + sbm_lm_trust_step; 4: + SBM_IMPLICIT_EXTRAP */

typedef struct sbm_ctx sbm_ctx;
typedef struct sbm_model sbm_model;
typedef struct sbm_project sbm_project;

/* the implicit kernels keep a column of the sensitivity matrix per lane in registers */
#define SBM_IMPLICIT_MAX_NV 128

/* integrators.  The reference always uses LSODA at rtol = atol = 1e-10
 * (model/ode_model.py:122-123,167-168); these are the GPU replacements. */
enum {
  SBM_RK4_FIXED = 0,   /* classic RK4, fixed step h0 between output times                    */
  SBM_DOPRI45 = 1,     /* Dormand-Prince 5(4), error control on state AND sensitivities      */
  /* Stiff systems: implicit midpoint rule, fixed step h0, Newton iteration with the sparse LU
   * the model generator worked out for the model's Jacobian pattern; the sensitivities are
   * the exact derivative of the scheme (one linear solve per column with the matrix Newton
   * just factored).  Second order, symmetric: the error expands in h^2, so two runs with
   * step_mult 1 and 2 extrapolate to fourth order.  Needs n_vars <= SBM_IMPLICIT_MAX_NV
   * (state component i lives on lane i mod 64; any n_sens: one wavefront per 64 columns).    */
  SBM_IMPLICIT_MIDPOINT = 2,
  /* The same with a graded first step, cut into 13 midpoint substeps of sizes
   * h0 * 2^-12, 2^-12, 2^-11, ..., 1/2: for initial conditions off a fast manifold -- the
   * reference always starts from y = 0 (model/ode_model.py:151-152) -- whose initial layer no
   * fixed step resolves.  The pattern belongs to the first step of the step_mult = 1 grid;
   * with step_mult = m every substep is cut into m parts, so that the grids of runs with
   * different step_mult are nested (what the h^2 expansion needs).                           */
  SBM_IMPLICIT_MIDPOINT_GRADED = 3,
  /* Stiff systems in ONE call: implicit midpoint with error control inside the kernel.  Three solutions (steps H,
   * H/2, H/4 on nested grids) are carried side by side and never mixed; the symmetric rule's error expands in H^2,
   * so T22 = (4 y_{H/2} - y_H)/3 and T32 = (4 y_{H/4} - y_{H/2})/3 are fourth-order results at every output time:
   * T32 is written, |T32 - T22|/3 is the error estimate -- too large at any output time and the trajectory starts
   * over with more steps per unit time (at most 12 passes; one restart is the rule).  rtol / atol as for DOPRI45
   * (state AND sensitivities, column by column); h0 = coarse step of the first pass (<= 0: span / 64).  The first
   * step is graded as in SBM_IMPLICIT_MIDPOINT_GRADED.  What method='auto' of the Python classes switches to for
   * vectors DOPRI45 gives up on -- the role of LSODA's own switch to BDF (model/ode_model.py:122-123).  n_steps
   * counts the coarse steps of the accepted pass (each goes with 2 + 4 finer steps: seven midpoint solves),
   * n_reject those of the abandoned passes. */
  SBM_IMPLICIT_ADAPTIVE = 4,
  /* Dormand-Prince 8(5,3) (DOP853): twelve stages per step, error control as DOPRI45 (Hairer's combination of the
   * fifth- and third-order embedded estimates), for TIGHT tolerances -- at the default rtol 1e-9 a seventh of
   * DOPRI45's steps and a third of its right-hand-side evaluations on the 20-state model with sensitivities.  Runs on
   * the row-group sensitivity kernels and the state-rows / packed state kernels (models whose rows the generator could
   * split: sbm_model_info; others answer SBM_E_ARG).  opts.variant: AUTO or SMALL_BATCH. */
  SBM_DOP853 = 5,
  /* Stiff systems with LOCAL error control, one call, no step count and no restarts: extrapolated implicit Euler.  A
   * macro step of size H is integrated K times, with 1, 2, ..., K implicit-Euler steps (Newton with the model's
   * symbolic sparse LU, sensitivities as the exact derivative of every Euler step: one solve per column with Newton's
   * factors); polynomial extrapolation to step size zero gives a result of order K and an embedded one of order K - 1
   * whose difference drives a per-trajectory step-size controller inside the kernel, as LSODA controls its own steps
   * (model/ode_model.py:122-123,167-168).  L-stable: every sub-result damps stiff components, so does any combination.
   * rtol / atol as for DOPRI45 (state AND sensitivities, column by column); h0 = first step (<= 0: automatic);
   * step_mult = K (0: 4 / 6 / 8 by rtol; at most 10); max_steps counts macro steps.  n_steps = accepted macro steps
   * (K (K + 1) / 2 Euler steps each), n_reject = rejected ones.  Needs n_vars <= SBM_IMPLICIT_MAX_NV.  What
   * method='auto' of the Python classes switches to for vectors the explicit integrator gives up on (since ABI 4). */
  SBM_IMPLICIT_EXTRAP = 6
};

typedef struct sbm_integrator_opts {
  int32_t method;    /* SBM_RK4_FIXED | SBM_DOPRI45 | SBM_DOP853 | SBM_IMPLICIT_MIDPOINT[_GRADED] | SBM_IMPLICIT_ADAPTIVE | SBM_IMPLICIT_EXTRAP */
  int32_t max_steps; /* per trajectory, accepted + rejected; 0 -> 1000000.  DOPRI45, negative: a budget of
                      * (and DOP853) |max_steps| with an early exit (status SBM_MAX_STEPS at once) for a trajectory whose
                      * current step size would need more than 1.5 x what is left of the budget for the remaining time span --
                      * checked every 256 attempts from the 512th on, and only while the step size has stopped
                      * growing from one check to the next: the explicit method on a stiff system */
  double rtol;       /* DOPRI45, IMPLICIT_ADAPTIVE: relative tolerance; IMPLICIT_MIDPOINT: Newton tolerance */
  double atol;       /* DOPRI45, IMPLICIT_ADAPTIVE: absolute tolerance; IMPLICIT_MIDPOINT: Newton tolerance */
  double h0;         /* RK4, IMPLICIT_MIDPOINT: step size; DOPRI45: initial step (<=0 -> automatic) */
  double t0;         /* time of the initial condition; output times must be >= t0.
                      * odeint takes t_sim[0] for it (model/ode_model.py:122,167);
                      * Project always integrates from 0 (base_project.py:419)     */
  int32_t variant;   /* sensitivity kernel: SBM_VARIANT_AUTO | _PER_WAVE | _ROW_LANE | _ROW_GROUP | _SMALL_BATCH | _MFMA | _PACKED */
  int32_t step_mult; /* fixed-step methods: every output interval is cut into
                      * step_mult * ceil(dt / h0) equal steps (0 = 1).  Doubling it halves every
                      * step exactly, which is what Richardson extrapolation needs.          */
} sbm_integrator_opts;

/* Three implementations of the sensitivity integrator with identical results up to
 * rounding, all one trajectory per wavefront.  PER_WAVE: one sensitivity column per
 * lane, f / J_y / J_p evaluated on every lane from broadcast operands.  ROW_LANE:
 * they are evaluated once, lane i working on row i (rows of the same kinetic form
 * side by side), and handed to the columns through LDS.  ROW_GROUP: ROW_LANE with
 * the rows of a column split over several lanes, so that all 64 lanes carry
 * equations and the Runge-Kutta stages fit the register file; sensitivity columns
 * that do not fit one wavefront are cut into chunks, one wavefront each (the columns
 * of a trajectory are coupled only through the state, of which every chunk integrates
 * a copy under its own step-size control: status and step counts of a trajectory are
 * the worst / largest over its chunks), and a lane carries up to four state rows.
 * ROW_LANE needs n_vars <= 64 and n_sens <= 64, ROW_GROUP n_vars <= 256 (any n_sens);
 * AUTO picks ROW_GROUP when the model generator found a paying split, else ROW_LANE
 * when the model fits, else PER_WAVE.  A forced variant the model does not support
 * falls back in the same order; for large models that have the ROW_GROUP form (more
 * than 4096 sensitivity entries) the other kernels are not built and the field is
 * ignored.  For the state-only entry points PER_WAVE selects the one-trajectory-per-lane
 * kernel (models up to 64 state variables), AUTO the one-trajectory-per-wavefront kernel
 * up to 2047 trajectories and its packed form (two or four trajectories per wavefront,
 * bit-identical results) beyond, ROW_LANE / ROW_GROUP always the unpacked one. */
enum { SBM_VARIANT_AUTO = 0, SBM_VARIANT_PER_WAVE = 1, SBM_VARIANT_ROW_LANE = 2, SBM_VARIANT_ROW_GROUP = 3,
       /* AUTO, except that the ROW_GROUP kernel takes its small-batch split -- more, smaller column chunks, fewer
        * equations per lane: the work of ONE wavefront per step is what a call with a single parameter vector
        * waits for -- while n_traj x chunks <= 2048 (two resident wavefronts per SIMD).  The two splits take different step sequences:
        * results agree to the integration tolerance, not bit for bit, which is why AUTO never switches by itself
        * (a batch call's rows do not depend on the size of the batch). */
       SBM_VARIANT_SMALL_BATCH = 4,
       /* The product J_y S of the sensitivity right-hand side on the matrix cores (v_mfma_f64_16x16x4_f64 tiles, S kept
        * in the accumulator layout, J_y handed over as a dense tile image): cost independent of the sparsity of J_y.
        * Opt-in: on MI355X the f64 matrix rate equals the f64 vector rate, so it pays only for dense Jacobians
        * (DESIGN.md records the measured crossover).  n_vars <= 64, sensitivity entry points, DOPRI45 / RK4; falls
        * back to AUTO otherwise. */
       SBM_VARIANT_MFMA = 5,
       /* Small models (n_vars, n_sens <= 32 and at most 256 sensitivity entries): several trajectories per wavefront, each
        * in a segment of 4 / 8 / 16 / 32 lanes with its own step-size control.  AUTO takes it from 2048 trajectories on
        * (a trajectory's numbers do not depend on its wavefront mates, but the step sequence differs in rounding from
        * the one-trajectory-per-wavefront kernels'); this value forces it for any batch.  Other models: as AUTO. */
       SBM_VARIANT_PACKED = 6 };

/* per-trajectory status written next to the results (the reference does not
 * check LSODA failures, model/ode_model.py:122,167; non-zero statuses are what
 * the host maps to the reference's inf rows, squared_loss_function.py:28-32) */
enum {
  SBM_OK = 0,
  SBM_MAX_STEPS = 1,
  SBM_NON_FINITE = 2,
  SBM_STEP_UNDERFLOW = 3,
  SBM_NEWTON_FAIL = 4,  /* implicit midpoint: Newton did not converge in 12 iterations */
  SBM_TOL_NOT_REACHED = 5  /* SBM_IMPLICIT_ADAPTIVE after its last refinement pass, and the HOST control loop around the
                              fixed-step implicit integrator (sysbio_modeling_amd/_control.py): the finest result is
                              returned, its error estimate is above the tolerance */
};

/* ---- context ----------------------------------------------------------- */
/* stream: the hipStream_t every call of this context enqueues on (e.g. torch's
 * current stream); NULL = the device's default stream. */
int sbm_ctx_create(int device, void* stream, sbm_ctx** out);
int sbm_ctx_destroy(sbm_ctx* ctx);
int sbm_ctx_synchronize(sbm_ctx* ctx);
int sbm_ctx_device(const sbm_ctx* ctx);
const char* sbm_last_error(void);
int sbm_abi_version(void);

/* ---- model: replaces OdeModel.__init__ (model/ode_model.py:27-44) ------- */
/* plugin_path: shared object built from a generated model header
 * (sysbio_modeling_amd/symbolic/emit.py + csrc/sbm_plugin_main.hip); it holds
 * the integrator kernels specialised for that model's RHS / sensitivity RHS. */
int sbm_model_load(sbm_ctx* ctx, const char* plugin_path, sbm_model** out);
int sbm_model_unload(sbm_model* m);
/* n_vars = OdeModel.n_vars (model/abstract_model.py:18-21); n_params =
 * len(param_order); n_sens = number of non-fixed parameters. */
int sbm_model_info(const sbm_model* m, int32_t* n_vars, int32_t* n_params, int32_t* n_sens,
                   char* name_buf, int32_t name_buf_len);

/* ---- OdeModel.simulate (model/ode_model.py:128-169), batched ------------ */
/* P      [V][n_params]            one parameter vector per trajectory
 * t_out  [n_t]                    non-decreasing output times, t_out[0] >= opts->t0;
 *                                 integration starts at t = opts->t0
 * y0     [n_vars] or NULL         initial state (NULL -> zeros, as the
 *                                 reference's default, ode_model.py:151-152)
 * Y      [V][n_t][n_vars]         out
 * status [V], n_steps [V] (accepted steps), n_reject [V]: out, each nullable */
int sbm_simulate_batch(sbm_model* m, const double* P_dev, int32_t V, const double* t_out_dev,
                       int32_t n_t, const double* y0_dev, const sbm_integrator_opts* opts,
                       double* Y_dev, int32_t* status_dev, int32_t* n_steps_dev,
                       int32_t* n_reject_dev);
int sbm_simulate_batch_host(sbm_model* m, const double* P, int32_t V, const double* t_out,
                            int32_t n_t, const double* y0, const sbm_integrator_opts* opts,
                            double* Y, int32_t* status, int32_t* n_steps, int32_t* n_reject);

/* ---- OdeModel.calc_jacobian (model/ode_model.py:83-126), batched -------- */
/* Integrates the augmented system [y; S], S' = J_y S + J_p.
 * yS0 [n_vars + n_vars*n_sens] or NULL (-> zeros): the reference's
 *      init_conditions argument, state first then S, state-major/param-minor.
 * Y   [V][n_t][n_vars]           out, nullable (the reference discards it, :125)
 * S   [V][n_t][n_vars][n_sens]   out; S[..., i, j] = d y_i / d p_j, i.e. the
 *      reference's column i*k + j of calc_jacobian's return value */
int sbm_sens_batch(sbm_model* m, const double* P_dev, int32_t V, const double* t_out_dev,
                   int32_t n_t, const double* yS0_dev, const sbm_integrator_opts* opts,
                   double* Y_dev, double* S_dev, int32_t* status_dev, int32_t* n_steps_dev,
                   int32_t* n_reject_dev);
int sbm_sens_batch_host(sbm_model* m, const double* P, int32_t V, const double* t_out,
                        int32_t n_t, const double* yS0, const sbm_integrator_opts* opts,
                        double* Y, double* S, int32_t* status, int32_t* n_steps,
                        int32_t* n_reject);

/* ---- Project (project/base_project.py) ---------------------------------- */
/* Flattened, immutable description of a Project: what _set_local_param_idx
 * (:164-276), _measurements_as_dataframe (:296-341) and _make_mapping (:109-136)
 * build, as index arrays.  All arrays are HOST pointers, copied at load. */
typedef struct sbm_project_desc {
  int32_t n_experiments;     /* E, in the reference's sorted-by-name order (:624) */
  int32_t n_project_params;  /* q = Project.n_project_params                      */
  int32_t n_rows;            /* R = measurement rows (t = 0 rows already dropped) */
  int32_t n_sf_groups;       /* scale-factor groups; 0 = loss without SF          */
  int32_t n_prior_rows;      /* parameter-prior rows appended after the R rows    */
  int32_t n_sf_prior_rows;   /* scale-factor-prior rows appended after those      */

  /* theta -> p gather, get_experiment_parameters (:343-363):
   * p[e][m] = exp(theta[pmap[e][m]]) if pmap >= 0 else pfixed[e][m] */
  const int32_t* pmap;   /* [E][n_params] */
  const double* pfixed;  /* [E][n_params] */
  /* model param m -> sensitivity column (or -1 when fixed at code generation) */
  const int32_t* sens_col; /* [n_params] */

  /* per-experiment output grid: the entries of linspace(0, t_end, 1000) picked by
   * searchsorted (project/utils.py:18-21), unique and increasing */
  const int32_t* tgrid_off; /* [E+1] offsets into tgrid */
  const double* tgrid;      /* [tgrid_off[E]] */

  /* rows, in the reference's row order (experiment name, measurement name, time) */
  const int32_t* row_exp;     /* [R] experiment index                              */
  const int32_t* row_tidx;    /* [R] index into that experiment's tgrid            */
  const int32_t* row_var_off; /* [R+1] offsets into row_vars ('direct': 1 entry,   */
  const int32_t* row_vars;    /*      'sum': several; project/utils.py:10-89)      */
  const double* row_data;     /* [R] measurement mean                              */
  const double* row_sigma;    /* [R] measurement std (never 0,                     */
                              /*     measurement/abstract_measurement.py:15-16)    */
  const int32_t* row_sf;      /* [R] scale-factor group or -1                      */

  /* parameter log-priors (:316-325, :427-437): residual (theta[idx]-mean)/sigma */
  const int32_t* prior_idx;  /* [n_prior_rows] */
  const double* prior_mean;  /* [n_prior_rows] */
  const double* prior_sigma; /* [n_prior_rows] */

  /* scale-factor log-priors ("~~SF_Prior" rows, :327-335): residual
   * (log B_g - mean)/sigma, Jacobian row (dB_g/dtheta)/B_g
   * (linear_scale_factor.py:44-61, abstract_loss_function.py:93-120) */
  const int32_t* sf_prior_group; /* [n_sf_prior_rows] */
  const double* sf_prior_mean;   /* [n_sf_prior_rows] */
  const double* sf_prior_sigma;  /* [n_sf_prior_rows] */

  /* reference_compat != 0 reproduces two reference behaviours exactly:
   * the Jacobian is NOT divided by sigma (squared_loss_function.py:52-53,76)
   * and prior rows of J stay zero (base_project.py:519-530 is never called).
   * 0 gives d r / d theta (J/sigma, prior rows 1/sigma_prior). */
  int32_t reference_compat;
  /* SBM_LOSS_SQUARE:      r = (B s - d)/sigma                (squared_loss_function.py:27-42)
   * SBM_LOSS_LOG_SQUARE:  r = (log(B s) - log d)/sigma, geometric-mean scale factor with weights
   *                       (d/sigma)^2, J = J_m/s + (dB/dtheta)/B
   *                       (log_squared_loss_function.py:23-98, log_scale_factor.py:17-36); needs d > 0.
   * The reference's NormalizedSquareLossFunction (normalized_squared_loss_function.py:23-46) is
   * SBM_LOSS_SQUARE with row_sigma already multiplied by row_data on the host. */
  int32_t loss_type;

  /* 'custom' measurement mappings.  The reference takes two Python callbacks there -- (parameters, map function,
   * Jacobian map function), project/base_project.py:125-128, called per measurement at :380-383 and :461-464 --
   * which cannot run in a kernel.  Here the observable is an EXPRESSION g(y_a, y_b, ...; t) of model variables at
   * the sampled grid point, which the host compiles into postfix programs (SBM_OP_*):
   *   row_prog[r] = program of the row's measure, or -1 ('direct' / 'sum': plain sum of the row's variables).
   *   Program c lists n = prog_nvars[c] variables (the row's row_vars entries, in that order) and has 1 + n
   *   subprograms: the value g, then dg/dy_k for k = 0..n-1; the measure's Jacobian row is
   *   sum_k dg/dy_k * S[var_k, :] (what the reference's Jacobian map function returns, project/utils.py:29-45).
   * prog_sub_off holds the start of every subprogram in prog_code, program after program
   * (sum_c (1 + prog_nvars[c]) entries + 1).  All NULL / 0 without custom mappings. */
  int32_t n_programs;
  int32_t n_prog_code;
  int32_t n_prog_const;
  const int32_t* row_prog;     /* [R] */
  const int32_t* prog_nvars;   /* [n_programs] */
  const int32_t* prog_sub_off; /* [sum(1 + prog_nvars) + 1] */
  const int32_t* prog_code;    /* [n_prog_code] opcodes, operands inline */
  const double* prog_const;    /* [n_prog_const] */
  const double* row_time;      /* [R] sampled time of each row (what SBM_OP_TIME pushes) */
} sbm_project_desc;

enum { SBM_LOSS_SQUARE = 0, SBM_LOSS_LOG_SQUARE = 1 };

/* postfix programs of custom observables: a stack machine over doubles, at most SBM_PROG_MAX_STACK deep.
 * VAR k pushes the k-th variable of the row's list, CONST i pushes prog_const[i], POWI n raises the top to the
 * integer power n (|n| <= 64); the binary operators pop b then a and push a (op) b. */
enum { SBM_OP_END = 0, SBM_OP_VAR = 1, SBM_OP_CONST = 2, SBM_OP_ADD = 3, SBM_OP_SUB = 4, SBM_OP_MUL = 5, SBM_OP_DIV = 6,
       SBM_OP_NEG = 7, SBM_OP_POW = 8, SBM_OP_POWI = 9, SBM_OP_EXP = 10, SBM_OP_LOG = 11, SBM_OP_SQRT = 12,
       SBM_OP_TANH = 13, SBM_OP_SIN = 14, SBM_OP_COS = 15, SBM_OP_ABS = 16, SBM_OP_SIGN = 17, SBM_OP_TIME = 18,
       SBM_OP_COUNT = 19 };
#define SBM_PROG_MAX_STACK 16

int sbm_project_load(sbm_model* m, const sbm_project_desc* desc, sbm_project** out);
int sbm_project_unload(sbm_project* p);

/* Project.residuals (:708-729) for V parameter vectors at once.
 * Theta   [V][q]          log-space project vectors
 * sims    [V][R]          out, nullable: unscaled simulated values per row
 *                         (the 'mean' column of _simulations_df, :387-389)
 * R_out   [V][R+n_prior+n_sf_prior]  out: (B*sim - data)/sigma  (squared_loss_function.py:40)
 * sf      [V][n_sf]       out, nullable: scale factors B (linear_scale_factor.py:27-31)
 * norms   [V]             out, nullable: sum of squared residuals (2 * RSS)
 * status  [V]             out, nullable: max status over the vector's trajectories;
 *                         rows of a failed vector are +inf as in the reference */
int sbm_residuals_batch(sbm_project* p, const double* Theta_dev, int32_t V,
                        const sbm_integrator_opts* opts, double* sims_dev, double* R_dev,
                        double* sf_dev, double* norms_dev, int32_t* status_dev,
                        int32_t* n_steps_dev);

/* Project.calc_project_jacobian (:731-771): one augmented integration gives both.
 * J_out   [V][R+n_prior+n_sf_prior][q]   out: B*J + sim (x) dB/dtheta
 *                             (squared_loss_function.py:44-80, linear_scale_factor.py:33-42)
 * Jmodel  [V][R][q]           out, nullable: d sim / d theta before scale factors
 *                             (_model_jacobian_df, base_project.py:482-485)
 * sf_grad [V][n_sf][q]        out, nullable: dB/dtheta (linear_scale_factor.py:33-42)
 * grad    [V][q]              out, nullable: J^T r (calc_rss_gradient, :803-805) */
int sbm_jacobian_batch(sbm_project* p, const double* Theta_dev, int32_t V,
                       const sbm_integrator_opts* opts, double* sims_dev, double* R_dev,
                       double* J_dev, double* Jmodel_dev, double* sf_dev, double* sf_grad_dev,
                       double* norms_dev, double* grad_dev, int32_t* status_dev,
                       int32_t* n_steps_dev);

/* Richardson extrapolation for SBM_IMPLICIT_MIDPOINT evaluations of this project: levels = 1 integrates
 * every trajectory twice (step_mult and 2*step_mult) and combines the sampled states and sensitivities as
 * (4 fine - coarse)/3 BEFORE the residual / Jacobian assembly; levels = 2 adds a third run (h^4 term);
 * 0 (default) switches it off.  Other methods ignore it.  Host-level counterpart of
 * OdeModel's ``extrapolate`` option. */
int sbm_project_set_extrapolation(sbm_project* p, int32_t levels);

/* bytes of device scratch the project keeps for V vectors (allocated lazily,
 * grown on demand, freed at unload) */
int64_t sbm_project_scratch_bytes(const sbm_project* p, int32_t V, int32_t with_sens);

/* ---- loss functions on caller-supplied simulations -------------------- */
/* The reference's loss classes are also callable on hand-built frames
 * (SquareLossFunction.residuals / .jacobian / .evaluate,
 * LossFunctionWithScaleFactors.update_scale_factors / .scale_sim_values /
 * .update_sf_priors_residuals / .update_sf_priors_gradient,
 * squared_loss_function.py:23-108, abstract_loss_function.py:47-120,
 * log_squared_loss_function.py:28-98).  This entry runs the same assembly
 * kernel as sbm_residuals_batch on V stacked frames whose 'mean' columns the
 * caller supplies, so those methods keep working without an integrator. */
typedef struct sbm_loss_desc {
  int32_t n_rows;           /* R: rows of the simulations frame without "~~SF_Prior" rows */
  int32_t n_params;         /* q: columns of the simulations Jacobian (0 without one) */
  int32_t n_sf_groups;      /* G */
  int32_t n_sf_prior_rows;  /* appended after the R rows, as in sbm_project_desc */
  int32_t loss_type;        /* SBM_LOSS_* */
  int32_t reference_compat; /* 1: Jacobian rows not divided by sigma (see sbm_project_desc) */
  const double* row_data;   /* [R] measurement mean */
  const double* row_sigma;  /* [R] measurement std (non-zero) */
  const int32_t* row_sf;    /* [R] scale-factor group or -1 */
  const int32_t* row_plain; /* [R] nullable; 1 = "~Prior" row: (s - d)/sigma even under the
                               log loss (log_squared_loss_function.py:33-40 drops them
                               from the transform) */
  const int32_t* sf_prior_group; /* [n_sf_prior_rows] */
  const double* sf_prior_mean;
  const double* sf_prior_sigma;
} sbm_loss_desc;

/* All pointers are HOST memory (frames live on the host); staged through the
 * context's stream, synchronous on return.
 * sims   [V][R]               in
 * Jm     [V][R][q]            in, nullable: d sim / d theta (simulations_jacobian)
 * R      [V][R + n_sf_prior]  out residuals; all inf where a frame holds NaN (or a
 *                             non-positive simulation under the log loss)
 * J      [V][R + n_sf_prior][q] out, nullable (needs Jm)
 * sf     [V][G]               out, nullable
 * sf_grad[V][G][q]            out, nullable (needs Jm)
 * norms  [V]                  out, nullable: sum r^2
 * status [V]                  out, nullable: 0 or SBM_NON_FINITE */
int sbm_loss_eval_host(sbm_ctx* ctx, const sbm_loss_desc* desc, int32_t V, const double* sims,
                       const double* Jm, double* R, double* J, double* sf, double* sf_grad,
                       double* norms, int32_t* status);

/* ---- batched Levenberg-Marquardt step ---------------------------------- */
/* The caller after the path.  The reference fits one start at a time with
 * scipy.optimize.leastsq(project.residuals, x0, Dfun=project.calc_project_jacobian)
 * (tests/test_Project.py:202-213, :352-357); here every vector of an ensemble
 * takes its own damped Gauss-Newton step on the device:
 *     (J^T J + lambda_v diag(J^T J)) delta_v = -J^T r_v.
 * J      [V][M][q]  in  d r / d theta as sbm_jacobian_batch returns it with reference_compat = 0
 * r      [V][M]     in  residuals
 * lambda [V]        in  damping, >= 0
 * delta  [V][q]     out step
 * pred   [V]        out predicted decrease of 0.5 |r|^2 (Gauss-Newton model)
 * status [V]        out 0, or 1 where the system is not positive definite / the input not finite
 *                       (delta = 0: raise lambda)
 * Device pointers; q <= 128; enqueued on the context's stream. */
int sbm_lm_step(sbm_ctx* ctx, const double* J_dev, const double* r_dev, const double* lambda_dev,
                int32_t V, int32_t M, int32_t q, double* delta_dev, double* pred_dev,
                int32_t* status_dev);

/* The same step with the damping chosen as MINPACK's lmder chooses it (lmpar, More 1978): given a scaling D and a
 * trust-region radius Delta per vector, lambda >= 0 is found such that
 *     (J^T J + lambda D^2) delta = -J^T r   and   | ||D delta|| - Delta | <= 0.1 Delta
 * (lambda = 0 when the Gauss-Newton step is already inside) -- a safeguarded Newton iteration on the q x q system, a
 * few Cholesky factorisations inside ONE launch instead of trial integrations with lambda multiplied up and down.
 * dscale [V][q]  in / out  D: made max(D, column norm of J) here; zeros on the first call (MINPACK mode 1)
 * radius [V]     in        Delta > 0
 * lambda [V]     in / out  the previous parameter as a starting guess (0 on the first call) -> the one found
 * delta  [V][q]  out       step
 * pred   [V]     out       predicted decrease of 0.5 |r|^2 = 0.5 |J delta|^2 + lambda ||D delta||^2
 * dxnorm [V]     out       ||D delta||
 * status [V]     out       0, or 1: input not finite / no positive definite system found (delta = 0)
 * What the caller does with them is lmder's bookkeeping (project/fitting.py, algorithm='trust_region'): ratio of actual
 * to predicted reduction, radius update, acceptance.  Device pointers; q <= 128; enqueued on the context's stream.
 * The reference fits with scipy.optimize.leastsq = MINPACK lmder (tests/test_Project.py:202-213, 351-357). */
int sbm_lm_trust_step(sbm_ctx* ctx, const double* J_dev, const double* r_dev, double* dscale_dev,
                      const double* radius_dev, double* lambda_dev, int32_t V, int32_t M, int32_t q,
                      double* delta_dev, double* pred_dev, double* dxnorm_dev, int32_t* status_dev);

/* sbm_lm_trust_step with what a batched fitting loop needs around it folded in (project/fitting.py, since round 3):
 *   row_scale [M]  nullable  J is used as diag(row_scale) J -- a reference_compat Jacobian is not divided by sigma
 *                            (squared_loss_function.py:52-53,76), the normal equations want d r / d theta
 *   skip [V]       nullable  != 0: the vector is left alone (delta = 0, trial = theta, status 2): converged starts
 *   max_step                 > 0: every component of the step is clipped to +-max_step (exp(theta) stays finite); pred,
 *                            dxnorm and gtx are then those of the step TAKEN (pred = -g.x - x^T J^T J x / 2)
 *   theta, trial [V][q]      nullable, together: trial = theta + delta
 *   gtx [V]        nullable  out: g . delta, the directional derivative of 0.5 |r|^2 along the step (lmder's dirder x |r|^2)
 * A damped system that fails to factor after an earlier one succeeded leaves the LAST SOLVED step (round 2 returned the
 * unsolved right-hand side with status 0).  LDS: the q x (q + 1) matrix + a row tile of J that shrinks from 32 to 8 rows
 * to fit the device's limit (q = 128 fits with 16 rows); a q that does not fit is refused with SBM_E_ARG. */
int sbm_lm_trust_step_ex(sbm_ctx* ctx, const double* J_dev, const double* r_dev, double* dscale_dev,
                         const double* radius_dev, double* lambda_dev, int32_t V, int32_t M, int32_t q,
                         const double* row_scale_dev, const int32_t* skip_dev, double max_step,
                         const double* theta_dev, double* trial_dev, double* delta_dev, double* pred_dev,
                         double* dxnorm_dev, double* gtx_dev, int32_t* status_dev);

/* lmder's bookkeeping between two steps (MINPACK lmder.f, the loop around lmpar: what scipy.optimize.leastsq -- the
 * reference's optimiser, tests/test_Project.py:202-213, 351-357 -- does after every function evaluation), for V starts
 * in one launch: actual and predicted relative reduction, their ratio, the radius / lambda update (ratio <= 1/4: shrink
 * by 1/2 or by the parabola-fit factor down to 1/10; ratio >= 3/4 or lambda = 0: Delta = 2 ||D delta||), acceptance
 * (ratio >= 1e-4 and an integrable trial point) and the convergence tests (info 1: both reductions <= ftol with
 * ratio <= 2; info 2: Delta <= xtol ||D theta||).
 *   cost [V]            0.5 |r|^2 at the current points       norms_trial, status_trial [V]   |r|^2 and integration
 *   pred, dxnorm, gtx, step_status [V]   from sbm_lm_trust_step_ex     status of the trial points (sbm_*_batch)
 *   theta, dscale [V][q]                 current points, scaling
 *   iteration, first                     iteration index (recorded in n_iter on convergence); first != 0: lmder's
 *                                        first-iteration rule Delta = min(Delta, ||D delta||)
 *   radius, lambda, done, n_iter [V]     in / out            accept [V]  out: 1 = the trial point is taken
 *   counters [2]                         out: starts still running, trial points accepted
 *   ratio [V]                            nullable, out (traces)
 * sbm_lm_accept then copies theta, residuals, Jacobian and cost of the accepted trial points over the current ones. */
int sbm_lm_update(sbm_ctx* ctx, const double* cost_dev, const double* norms_trial_dev, const int32_t* status_trial_dev,
                  const double* pred_dev, const double* dxnorm_dev, const double* gtx_dev, const int32_t* step_status_dev,
                  const double* theta_dev, const double* dscale_dev, int32_t V, int32_t q, double ftol, double xtol,
                  int32_t iteration, int32_t first, double* radius_dev, double* lambda_dev, int32_t* done_dev,
                  int32_t* accept_dev, int32_t* n_iter_dev, int32_t* counters_dev, double* ratio_dev);
int sbm_lm_accept(sbm_ctx* ctx, const int32_t* accept_dev, int32_t V, int32_t M, int32_t q, const double* trial_dev,
                  const double* r_trial_dev, const double* J_trial_dev, const double* norms_trial_dev, double* theta_dev,
                  double* r_dev, double* J_dev, double* cost_dev);

/* Accepted steps of every TRAJECTORY of the project's last evaluation of V vectors, [V][E] (sbm_*_batch's n_steps is the
 * sum over a vector's E trajectories; an integrator budget applies per trajectory).  Device to device, on the stream. */
int sbm_project_trajectory_steps(sbm_project* project, int32_t V, int32_t* steps_dev);

/* ---- multi-GPU: the one exchange of the path ----------------------------- */
/* The path shards by parameter vector with no data-path collective; what every
 * rank may want afterwards is everybody's per-vector ||r||^2 (sbm_residuals_batch's
 * `norms`).  nccl_comm: an ncclComm_t of RCCL created by the caller (one rank per
 * GPU); send [count] and recv [n_ranks * count] are device buffers; enqueued on the
 * context's stream.  RCCL is looked up in the running process (no link dependency).
 * Python hosts use torch.distributed instead (sysbio_modeling_amd/distributed.py). */
int sbm_allgather_norms(sbm_ctx* ctx, void* nccl_comm, const double* send_dev, int32_t count,
                        double* recv_dev);

#ifdef __cplusplus
}
#endif
#endif /* SBM_H */
