/*
 * nempc.h -- C ABI of the MI355X-native NMPC callback engine (libnempc.so).
 *
 * The reference (Enderdead/pyNeuralEMPC) is pure Python and has no FFI of its own; what this
 * library replaces is the arithmetic behind the five solver callbacks of
 *     pyNeuralEMPC/optimizer/ipopt.py:30-96   (objective / gradient / constraints / jacobian / hessian)
 * i.e.   integrator/{discret,unity,rk4}.py  forward + jacobian (+ hessian),
 *        model/tensorflow.py:49-109         network forward / per-row jacobian / hessian,
 *        objective/jax.py:28-57             f, grad f, hess f   (quadratic + linear family),
 *        constraints.py:36-96               extra constraint rows (box rows on the states),
 * evaluated for a BATCH of B independent problems per call (the reference solves one).
 *
 * Conventions
 *  - every `const void*` / `void*` named Z, X0, f, grad, g, jac_*, lambda, sigma, hvals is a
 *    DEVICE pointer owned by the caller, holding elements of the handle's dtype
 *    (NEMPC_F64 -> double, NEMPC_F32 -> float), dense row-major, no padding;
 *  - weights / objective / bounds setters take HOST pointers to double;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls enqueue work and
 *    return; use nempc_sync or your own stream/event to wait;
 *  - every function returns 0 (NEMPC_OK) or a negative NEMPC_E* code and never throws;
 *    nempc_last_error() gives the message of the calling thread's last failure;
 *  - a handle is not re-entrant: one in-flight evaluation per handle (use one handle per stream).
 *
 * Layouts (identical to the reference, optimizer/ipopt.py:20-28):
 *    z  (n)      n = H*(nx+nu): z[0:H*nx] states (H,nx) row-major, z[H*nx:] controls (H,nu)
 *    g  (m)      m = H*nx defects [+ H*nx box rows = states.ravel() when enabled]
 *    jac_dense   (m,n) row-major -- what IpoptProblem.jacobian returns (ipopt.py:88-96)
 *    jac_tiles   (H,nx,nx+nu): row t = d Phi(x_{t-1},u_t) / d [x_{t-1} | u_t]  (compact contract)
 *
 * Rolling-window models (KerasTFModelRollingInput model/tensorflow.py:132-340, DiffDiscretJaxModelRollingWindow
 * model/jax.py:93-259): with rolling_window = w > 1 the network of step t reads the last w states and controls,
 *    xi_t = [ x_{t-w} .. x_{t-1} | u_{t-w+1} .. u_t | extras_t ]      (oldest first; newest first when
 *                                                                      rolling_reverse, tensorflow.py:120-127)
 * entries reaching before the horizon come from x0 and the history bound by nempc_bind_history.  The tile width
 * becomes w*(nx+nu) (window columns in the network's input order); the dense Jacobian / Hessian get the band the
 * reference builds with its projection matrices (jax.py:8-21, tensorflow.py:253-262, 312-340).
 */
#ifndef NEMPC_H
#define NEMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NEMPC_ABI_VERSION 7
#define NEMPC_MAX_LAYERS 8 /* dense layers incl. the linear output layer */

/* status codes */
#define NEMPC_OK 0
#define NEMPC_EINVAL (-1)   /* bad argument / dims */
#define NEMPC_ENOMEM (-2)   /* device allocation failed */
#define NEMPC_EHIP (-3)     /* a HIP runtime call failed */
#define NEMPC_ESTATE (-4)   /* call order: weights / objective not set yet */
#define NEMPC_EUNSUPPORTED (-5)

/* dtype */
#define NEMPC_F64 0
#define NEMPC_F32 1

/* integrator kind -- integrator/discret.py, unity.py, rk4.py */
#define NEMPC_DISCRET 0 /* Phi = x + f(x,u) */
#define NEMPC_UNITY 1   /* Phi = f(x,u) */
#define NEMPC_RK4 2     /* Phi = x + DT/6 (k1 + 2k2 + 2k3 + k4) */

/* activation of a dense layer -- what the Keras Dense layers of the model KerasTFModel wraps are built with
 * (model/tensorflow.py:8-29 takes ANY feed-forward Keras model; tensorflow.py:49-109 differentiates it by autodiff).
 * Derivatives are evaluated from the layer's output a = s(z): s' = 1 (linear), 1 - a^2 (tanh), [a > 0] (relu: the
 * gradient at 0 is 0, as TensorFlow's), a(1-a) (sigmoid), 1 - e^-a (softplus), 1 | a + alpha (elu), 1 | alpha (leaky_relu),
 * lambda | a + lambda alpha (selu, Keras' fixed constants) -- the MONOTONE activations, whose derivatives follow from the
 * output alone.  swish (= silu, z sigmoid(z)), gelu (z Phi(z), Keras' approximate=False), softsign, mish, exponential and relu6
 * are written from the pre-activation (the first two and mish are not monotone; for the others it is simply the form at
 * hand), which the layer-at-a-time matrix-core path and the generic kernel keep (NEMPC_KERNEL_LAYERED / AUTO / VALU; any
 * layer, the output layer included) -- asking for the register-resident kernels (NEMPC_KERNEL_MFMA*) with them is refused.
 * Per-layer MIXES of the monotone family under a linear output layer run on the register-resident kernels, too (hidden
 * widths <= 128, <= 3 hidden layers, 4 up to width 64): the layers' codes are launch arguments there.
 * elu and leaky_relu read their alpha from nempc_config.act_param (elu: > 0, leaky_relu: >= 0). */
#define NEMPC_ACT_LINEAR 0
#define NEMPC_ACT_TANH 1
#define NEMPC_ACT_RELU 2
#define NEMPC_ACT_SIGMOID 3
#define NEMPC_ACT_SOFTPLUS 4
#define NEMPC_ACT_ELU 5
#define NEMPC_ACT_LEAKY_RELU 6
#define NEMPC_ACT_SELU 7
#define NEMPC_ACT_SWISH 8
#define NEMPC_ACT_GELU 9
#define NEMPC_ACT_SOFTSIGN 10    /* z / (1 + |z|)                    -- these four, like swish / gelu, from the pre-activation: */
#define NEMPC_ACT_MISH 11        /* z tanh(softplus(z))                  the layered path and the generic kernel */
#define NEMPC_ACT_EXPONENTIAL 12 /* e^z */
#define NEMPC_ACT_RELU6 13       /* min(max(z, 0), 6): Keras' ReLU(max_value=6) */
#define NEMPC_ACT_COUNT 14
#define NEMPC_ACT_FIRST_ZBASED NEMPC_ACT_SWISH /* codes >= this one are written from the pre-activation */

/* row-kernel implementation */
#define NEMPC_KERNEL_AUTO 0
#define NEMPC_KERNEL_VALU 1 /* generic thread-per-row kernel, any dims */
#define NEMPC_KERNEL_MFMA 2 /* matrix-core kernels, padded hidden width in {32,64,128}, <=3 hidden layers, nx <= 16,
                               network inputs w*(nx+nu)+n_extra <= 32, ONE non-linear activation on every hidden layer
                               and a linear output layer: CU-cooperative kernel when the weight slices fit the
                               registers and the inputs <= 16, else wave-per-tile */
#define NEMPC_KERNEL_MFMA_TILE 3 /* force the wave-per-tile matrix-core kernel (A/B measurements) */
#define NEMPC_KERNEL_LAYERED 4 /* layer-at-a-time matrix-core path: one GEMM launch per dense layer over all B*H rows, any
                                  activation per layer (output layer included), hidden widths <= 1024, up to 8 layers,
                                  w*(nx+nu) <= 32, nx <= 16 -- what any feed-forward Keras model the reference wraps
                                  (model/tensorflow.py:8-29) that NEMPC_KERNEL_MFMA does not take runs on under AUTO.
                                  Rows and Lagrangian blocks (Discret / Unity / RK4), hence the batched solver too */

typedef struct nempc_handle_s* nempc_handle;

typedef struct nempc_config {
    int32_t abi_version;              /* = NEMPC_ABI_VERSION */
    int32_t device;                   /* HIP device ordinal */
    int32_t dtype;                    /* NEMPC_F64 | NEMPC_F32 */
    int32_t integrator;               /* NEMPC_DISCRET | NEMPC_UNITY | NEMPC_RK4 */
    int32_t H, nx, nu;                /* Integrator.H, Model.x_dim, Model.u_dim */
    int32_t n_layers;                 /* dense layers, >= 1; layer l applies activations[l] (below) */
    int32_t widths[NEMPC_MAX_LAYERS]; /* output width of each layer; widths[n_layers-1] == nx */
    int32_t max_batch;                /* capacity B_max of the workspaces */
    int32_t kernel;                   /* NEMPC_KERNEL_* */
    int32_t n_extra;                  /* extra per-step network inputs after [x | u]: tvp_dim + p_dim of the reference's
                                         Model (model/tensorflow.py:39-47); they feed the network but are not decision
                                         variables: no Jacobian / Hessian columns (tensorflow.py:65-66). 0 = none */
    int32_t rolling_window;           /* w >= 1: steps of history the network reads (1 = plain model).  w > 1 needs
                                         DISCRET or UNITY (the reference has no RK4 for rolling models) */
    int32_t rolling_reverse;          /* 0: window rows oldest -> newest (forward_rolling=True); 1: newest first */
    double DT;                        /* RK4 step (RK4Integrator.DT, rk4.py:49) */
    int32_t activations[NEMPC_MAX_LAYERS]; /* NEMPC_ACT_* of each layer, the output layer included (the reference's
                                         nn_model.h5: tanh, tanh, linear).  Any mix runs on the generic kernel; the
                                         matrix-core kernels take one non-linear activation on all hidden layers and
                                         a linear output layer (NEMPC_KERNEL_AUTO picks accordingly) */
    double act_param[NEMPC_MAX_LAYERS];    /* alpha of an elu / leaky_relu layer (Keras: ELU(alpha), LeakyReLU(negative_slope));
                                         ignored for the other activations.  The register-resident matrix-core kernels take
                                         elu with alpha = 1 only; leaky_relu, selu and elu(alpha != 1) run on the layered
                                         matrix-core path and the generic kernel */
} nempc_config;

/* lifetime ------------------------------------------------------------------------------- */
int nempc_create(const nempc_config* cfg, nempc_handle* out);
int nempc_destroy(nempc_handle h);

/* grow the workspaces to hold max_batch problems per call (no-op when they already do).  Everything else the handle
 * owns -- weights, objective, structure, bindings, communicator -- is kept.  Synchronises the device. */
int nempc_reserve(nempc_handle h, int32_t max_batch);

/* network weights: W[l] is (in_l, out_l) row-major -- the Keras `kernel` layout used by
 * KerasTFModel (model/tensorflow.py:8-51); b[l] is (out_l). Host doubles. */
int nempc_set_weights(nempc_handle h, const double* const* W, const double* const* b);

/* objective family standing in for JAXObjectifFunc (objective/jax.py:16-57):
 *   f = sum_t (x_t-xref_t)^T Q (x_t-xref_t) + (u_t-uref_t)^T R (u_t-uref_t) + cx_t.x_t + cu_t.u_t
 * Q (nx,nx), R (nu,nu), xref/cx (H,nx), uref/cu (H,nu); any pointer may be NULL (= zeros;
 * Q NULL = identity, R NULL = 0.1*identity, the SURVEY 8(d) defaults). Host doubles.
 * Ordering: the values overwrite the handle's parameter block in place with a copy on the NULL stream (no re-allocation, no
 * device synchronisation -- a tracking MPC moves xref / uref every step).  The NULL stream orders the copy against work on
 * blocking streams only: a caller that queues nempc_eval / nempc_hess / nempc_solve on a NON-BLOCKING stream must
 * nempc_sync that stream before changing the objective (same for nempc_set_terminal_weight), or the queued kernels may
 * read a half-updated block. */
int nempc_set_objective(nempc_handle h, const double* Q, const double* R, const double* xref,
                        const double* uref, const double* cx, const double* cu);

/* terminal cost: the last step's state term uses QT (nx,nx) instead of Q,
 *   ... + (x_{H-1}-xref_{H-1})^T QT (x_{H-1}-xref_{H-1});   NULL = back to Q.  Kept across nempc_set_objective calls.
 * (A member of the same family: JAXObjectifFunc takes any function of the whole trajectory, objective/jax.py:28-33.) */
int nempc_set_terminal_weight(nempc_handle h, const double* QT);

/* bind the extra network inputs for subsequent nempc_eval / nempc_hess / nempc_solve calls: E (B,H,n_extra) device,
 * dtype of the handle, row t of problem b = [tvp_t ; p] (time-varying parameters then constant parameters, the
 * concatenation order of KerasTFModel._gather_input).  The pointer is stored, not copied; required when
 * n_extra > 0.  B = problems the tensor covers: a later evaluation of more problems than that is refused
 * (NEMPC_EINVAL) instead of reading past the end.  */
int nempc_bind_extra(nempc_handle h, const void* E, int32_t B);

/* history of a rolling-window model for subsequent calls (set_prev_data, model/tensorflow.py:178-189):
 * hist_x (B, w-1, nx) = the w-1 states BEFORE x0, oldest first; hist_u (B, w-1, nu) = the w-1 controls before u_0.
 * Device pointers of the handle's dtype, stored not copied; required when rolling_window > 1.  B = problems the
 * tensors cover (evaluations of more problems are refused). */
int nempc_bind_history(nempc_handle h, const void* hist_x, const void* hist_u, int32_t B);

/* extra constraint rows g_box = states.ravel() (a Constraint in the sense of constraints.py:36-63
 * with constant selector Jacobian); lo/hi (nx) host doubles are only reported back through
 * nempc_constraint_bounds. enabled=0 removes the rows. */
int nempc_set_box_rows(nempc_handle h, int enabled, const double* lo, const double* hi);

/* sizes: n, m, structural nnz of jac, nnz of the lower-triangular Lagrangian Hessian */
int nempc_dims(nempc_handle h, int32_t* n, int32_t* m, int32_t* nnz_jac, int32_t* nnz_hess);

/* cl, cu (m) host doubles -- IpoptProblem.get_constraint_{lower,upper}_bounds, ipopt.py:104-108 */
int nempc_constraint_bounds(nempc_handle h, double* cl, double* cu);

/* sparsity patterns, host int32, row-major sorted (the order of np.nonzero):
 * jacobian pattern (nnz_jac) and lower-triangular Hessian pattern (nnz_hess) -- replaces the
 * random-sampling probe of integrator/base.py:89-115 + ipopt.py:55-62 by the exact band pattern */
int nempc_jac_structure(nempc_handle h, int32_t* rows, int32_t* cols);
int nempc_hess_structure(nempc_handle h, int32_t* rows, int32_t* cols);

/* one batched evaluation of the hessian-free callbacks for B <= max_batch problems.
 *   Z (B,n), X0 (B,nx) inputs.  Outputs, each may be NULL to skip:
 *   f (B)            IpoptProblem.objective    ipopt.py:30-35
 *   grad (B,n)       IpoptProblem.gradient     ipopt.py:37-42
 *   g (B,m)          IpoptProblem.constraints  ipopt.py:44-52
 *   jac_dense (B,m,n)IpoptProblem.jacobian     ipopt.py:88-96
 *   jac_tiles (B,H,nx,w*(nx+nu))  compact per-step tiles (rk4.py:85-92 shape; w = rolling_window)
 *   jac_sparse (B,nnz_jac)    values in nempc_jac_structure order */
int nempc_eval(nempc_handle h, int32_t B, const void* Z, const void* X0, void* f, void* grad,
               void* g, void* jac_dense, void* jac_tiles, void* jac_sparse, void* stream);

/* Lagrangian Hessian values, IpoptProblem.hessian ipopt.py:66-86:
 *   hvals (B,nnz_hess) = (sigma_b * d2f + sum_i lambda_{b,i} d2g_i)[rows, cols]
 *   lambda (B,m), sigma (B).  Optional outputs (may be NULL): hdense (B,n,n) full symmetric matrix;
 *   hblocks (B,H,w*(nx+nu),w*(nx+nu)) the per-step blocks sum_k lambda_{t,k} d2 Phi_k / d[x_{t-1}|u_t]^2
 *   (the lambda-contracted form of Model.hessian, model/tensorflow.py:77-109). */
int nempc_hess(nempc_handle h, int32_t B, const void* Z, const void* X0, const void* lambda,
               const void* sigma, void* hvals, void* hdense, void* hblocks, void* stream);

/* Gauss-Newton Hessian callback (BASELINE.json north_star; no reference counterpart -- the reference's Hessian,
 * ipopt.py:66-86, is the exact one above):
 *   hvals (B,nnz_hess) = (sigma_b * d2f + sum_t T_t^T diag(w_{b,t}) T_t)[rows, cols],   T_t = jac tile of step t
 * in the SAME pattern as nempc_hess (nempc_hess_structure): the per-step blocks take the place of the Lagrangian
 * blocks, built from the row kernel's first-order tiles alone (no second-order sweep: one row-kernel launch, one block
 * kernel, one assembly).  w (B, H*nx) positive weights per defect row, NULL = ones; sigma (B).  Positive semidefinite
 * for w >= 0 and a convex objective.  Optional hdense (B,n,n), hblocks (B,H,w*(nx+nu),w*(nx+nu)) as in nempc_hess. */
int nempc_hess_gn(nempc_handle h, int32_t B, const void* Z, const void* X0, const void* w, const void* sigma,
                  void* hvals, void* hdense, void* hblocks, void* stream);

/* Batched on-device solver (no reference counterpart: the reference hands ONE problem at a time to Ipopt /
 * SLSQP on the CPU, optimizer/ipopt.py:138-195, slsqp.py:143-197).  Solves the B problems
 *     min f(z)  s.t.  integrator defects = 0,  lb <= z <= ub
 * in lock step by SQP with the exact per-step Lagrangian blocks (Gauss-Newton on the first iterate): per iterate one
 * callback + Hessian-block evaluation, one regularised Riccati (block-tridiagonal KKT) solve per problem, l1-merit
 * backtracking on defect-only evaluations; finite variable bounds by a primal-dual interior point (barrier parameter
 * mu, multipliers of the bounds updated with their own step length).
 *   X0 (B,nx) device; Z (B,n) device: initial guess in, solution out; lb/ub (n) HOST doubles (NULL = unbounded,
 *   +-INFINITY allowed; the vectors DomainConstraint.get_lower/upper_bounds produce, constraints.py:26-30);
 *   status (B) device int32 out: 0 converged (Optimizer.SUCCESS), 1 not converged (Optimizer.FAIL);
 *   *iters (host, optional) outer iterations run.  Synchronises the stream internally (convergence polls).
 *   Box ROWS (nempc_set_box_rows) are state limits: they are intersected with lb / ub on the state variables (what the
 *   reference's glue would hand Ipopt as constraint rows, optimizer/ipopt.py:44-52, is a bound here).
 *   lb[i] == ub[i] (a fixed variable) is NEMPC_EINVAL: the log barrier needs an interior.
 *   rolling_window > 1: needs nempc_bind_history.  The window is made the state (s_t = the last w states and w-1
 *   controls, w*nx + (w-1)*nu entries) and the same solver runs on that stage-wise problem, the callbacks staying the
 *   window kernels on the caller's variables; one trial step per iteration and no compaction whatever the options say. */
typedef struct nempc_solver_opts {
    int32_t max_iter;        /* outer iterations, e.g. 200 */
    int32_t max_linesearch;  /* halvings of a step before the direction is given up and the next LQ solve damped, e.g. 6 */
    int32_t check_every;     /* period, in iterations, of the BLOCKING convergence poll used for matrix-core-bound stages
                              * (e.g. 4); small stages do not poll: the device publishes the counter every iteration */
    int32_t lq_kernel;       /* LQ solve: 0 auto (by stage size), 1 sweep, one thread per problem, 2 sweep, one wave per problem,
                              * 3 parallel-in-time scan, one lane per stage (2-state / 1-control stages, fp64, H <= 63:
                              * what auto picks there; NEMPC_EUNSUPPORTED elsewhere) */
    double tol_constraint;   /* max |defect| at convergence, e.g. 1e-8 */
    double tol_step;         /* max |dz| <= tol_step * (1 + max |z|), e.g. 1e-8 */
    double mu_init, mu_min, mu_factor; /* barrier schedule, e.g. 1e-1, 1e-9, 0.2 */
    double reg;              /* initial Levenberg term on the control Hessian, e.g. 1e-9 */
    int32_t compact;         /* 1: whenever a quarter of the still-active problems has converged (looked at whenever the
                                convergence counter is read) gather the unconverged ones to the front and launch only over them -- the
                                stragglers then stop costing batch-wide launches; 0: lock step over all B to the end.
                                Results are identical either way (per-problem arithmetic does not depend on the slot). */
    int32_t barrier;         /* bounds: 0 primal-dual interior point (multipliers of the bounds carried along; default),
                                1 primal log barrier (the round-1 method; kept for A/B) */
    int32_t* iters_out;      /* optional device int32 (B): iteration at which each problem converged (0 = did not) */
    int32_t linesearch;      /* 0 auto (deferred for small stages, nx*(nx+nu) < 12, inner loop otherwise), 2 deferred
                                backtracking: ONE trial evaluation per outer iteration for the whole
                                batch; a problem whose trial is rejected stays where it is and retries the same direction at
                                half the length in the next iteration.  In a lock-step batch an inner backtracking loop makes
                                every problem pay for the one that needs six halvings (measured: 5.7 trial evaluations per
                                iteration at B=1024, 70 % of the solve time; at configs[2] dims a trial is 5 % of an iteration and
                                the inner loop converges more problems per second).  1: inner backtracking loop (round 1). */
    int32_t lq_attempts;     /* Riccati sweeps a problem may try per iteration (each restart damps ten times harder) before it
                                keeps its damping and sits the iteration out; 0 = default (3) */
} nempc_solver_opts;

int nempc_solve(nempc_handle h, int32_t B, const void* X0, void* Z, const double* lb, const double* ub,
                const nempc_solver_opts* opts, int32_t* status, int32_t* iters, void* stream);

int nempc_sync(nempc_handle h, void* stream);

/* Multi-GPU (no reference counterpart -- the reference has no distributed code).  Ranks own disjoint
 * shards of the problem batch, one process per GPU; nothing on the callback path communicates.  The single exchange
 * is an all-gather of the solved first controls u0 = z[H*nx : H*nx+nu] of every problem, once per MPC step, issued on
 * RCCL (ncclAllGather over xGMI) by the library itself:
 *   nempc_comm_unique_id   rank 0 only: fills id (NEMPC_COMM_ID_BYTES host bytes = an ncclUniqueId); the host side
 *                          hands it to the other ranks (torch.distributed broadcast in pyneuralempc_amd/parallel.py)
 *   nempc_comm_init        every rank, collectively: builds the handle's communicator on the handle's device
 *   nempc_allgather_u0     every rank, collectively, asynchronous on `stream`: packs this rank's u0 -- taken from
 *                          Z (B,n) (what nempc_solve returns), or from u0 (B,nu) when Z is NULL -- into its slot of
 *                          gathered (nranks*rows_per_rank, nu) and all-gathers in place.  rows_per_rank >= B is the
 *                          common slot size (the largest shard when shards are ragged; the pad rows are zero).
 *   nempc_comm_size        nranks / rank of the handle's communicator (0 / -1 when there is none)
 * RCCL is bound at run time on the first of these calls (dlopen librccl.so.1; NEMPC_EUNSUPPORTED when absent). */
#define NEMPC_COMM_ID_BYTES 128
int nempc_comm_unique_id(void* id);
int nempc_comm_init(nempc_handle h, int32_t nranks, int32_t rank, const void* id);
int nempc_allgather_u0(nempc_handle h, int32_t B, int32_t rows_per_rank, const void* Z, const void* u0,
                       void* gathered, void* stream);
int nempc_comm_size(nempc_handle h, int32_t* nranks, int32_t* rank);
int nempc_comm_destroy(nempc_handle h);

/* Launch planning of the cooperative kernels (host arithmetic only, no device needed): ntiles 16-row tiles over
 * per_cu co-resident workgroups on each of num_cus compute units -> grid workgroups, workgroup i owning
 * tiles_per_wg + (i < tiles_rem) consecutive tiles.  The library sizes every chip-filling launch this way from
 * hipDeviceProp_t::multiProcessorCount of the handle's device (nempc_num_cus), never from a literal CU count. */
int nempc_plan_grid(int32_t ntiles, int32_t num_cus, int32_t per_cu, int32_t* grid, int32_t* tiles_per_wg,
                    int32_t* tiles_rem);
int nempc_num_cus(nempc_handle h);

/* which row kernel the handle resolved to (NEMPC_KERNEL_VALU | NEMPC_KERNEL_MFMA | NEMPC_KERNEL_MFMA_TILE |
 * NEMPC_KERNEL_LAYERED) */
int nempc_kernel_variant(nempc_handle h);

/* row kernel the handle's most recent evaluation actually launched: 1 generic (rows_valu_kernel),
 * 2 cooperative matrix-core (rows_coop_kernel), 3 wave-per-tile matrix-core (rows_mfma_kernel), 4 cooperative
 * matrix-core compiled for the problem's shape (rows_coopfx_kernel), 5 rows_coop_kernel writing the dense Jacobian
 * rows itself (no assembly launch), 6 / 7 rows_coopfx_kernel / rows_coop_kernel writing the band-pattern values of the
 * sparse contract itself (no tile round trip, no assembly launch), 8 the layer-at-a-time GEMM pipeline
 * (NEMPC_KERNEL_LAYERED); 0 = none yet */
int nempc_last_row_kernel(nempc_handle h);

/* network kernel the handle's most recent nempc_hess (or solver iteration) launched for the Lagrangian blocks: 1 generic
 * (rowhess_valu_kernel), 2 cooperative matrix-core, forward-over-reverse (rowhess_coop_kernel), 3 wave-per-tile
 * matrix-core (rowhess_mfma_kernel), 4 compiled for the problem's shape, layer-wise contraction (rowhess_coopfx_kernel);
 * 5 the layer-at-a-time GEMM sweeps of wide / deep networks (layered_gemm_kernel + layered_hcontract_kernel);
 * + 10 when it ran inside the RK4 pipeline (stage records, stage multipliers, that kernel in direct mode, congruence
 * sum: integrator/rk4.py:181-285); 0 = none yet */
int nempc_last_hess_kernel(nempc_handle h);

const char* nempc_last_error(void);
int nempc_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NEMPC_H */
