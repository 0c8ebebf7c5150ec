// Batched on-device NMPC solver ("a lock-step batched iteration on device over the
// banded KKT system turns evals/sec into solved-MPC/sec").  The reference solves one problem at a time with a
// CPU solver (Ipopt via cyipopt, optimizer/ipopt.py:138-195, or SciPy SLSQP, optimizer/slsqp.py:143-197);
// this is the batched counterpart that consumes the same callbacks for B problems at once.
//
// Method: SQP with the exact Lagrangian Hessian on the multiple-shooting NLP
//     min f(z)  s.t.  Phi(x_{t-1},u_t) - x_t = 0,  lb <= z <= ub      (z as in optimizer/ipopt.py:20-28)
// * Hessian = objective blocks + the per-step Lagrangian blocks sum_k lambda_{t,k} d2 Phi_k / d[x_{t-1}|u_t]^2 of
//   the Hessian callback (a pure Gauss-Newton model was tried first: with multipliers of order 10 and a tanh
//   network it needs 50-200 iterations and damped steps; the exact blocks give 5-20 full steps).  The blocks
//   couple only (x_{t-1}, u_t), so every QP is still an LQ problem in the linearised dynamics
//   dx_t = A_t dx_{t-1} + B_t du_t + g_t, solved exactly by one backward Riccati sweep and one forward sweep per
//   problem (block-tridiagonal KKT, O(H (nx+nu)^3)).  Indefinite control Hessians Quu are handled the DDP way:
//   the sweep is repeated with a larger Levenberg term (decade steps from 1e-3, at most lq_attempts levels per
//   iteration, tried side by side on lanes of one wave in the per-thread kernel, the first level that goes through is
//   used: a problem still indefinite then keeps its damping and sits the iteration out, so that the launch does
//   not wait for it).  Multipliers = Riccati costates of the previous step;
// * variable bounds (DomainConstraint, constraints.py:3-33): primal-dual interior point -- the multipliers of the
//   bounds are iterates with their own step length; their diagonal terms keep the LQ structure (the primal log
//   barrier of round 1 stays as an option); mu is decreased per problem once its sub-problem has converged;
// * globalisation: l1 merit f_mu + nu |g|_1 with nu tracking the Riccati costates, backtracking from the
//   fraction-to-the-boundary step -- one trial per iteration for small stages (a rejected problem stands still and
//   retries the same direction at half the length next iteration), an inner loop for matrix-core-bound ones.
// Every problem carries its own mu, nu, step length, damping and status; the batch advances in lock step, the
// unconverged problems are compacted to the front as it converges.  All arithmetic is in kernels here; the callbacks
// are the handle's own row/objective kernels.  Per iteration on the compiled small shape (2 states, 1 control, fp64), TWO
// launches: (1) the LQ kernel -- acceptance test of the previous iteration's trial point (a wave per problem, before the
// staging; the accepted point's evaluation and Lagrangian blocks are carried over as the iterate's), barrier terms while
// staging, the LQ solve parallel in time (lq_scan_problem: a lane per stage), then step norms, dual steps, convergence test,
// merit and the trial point with its multipliers by a wave per problem; (2) ONE callback launch at the trial point:
// Lagrangian blocks, defects and tiles (rowhess_coopfx_kernel with evaluation outputs).  The accept phase publishes the
// convergence counter to pinned host memory, the host stays at most two iterations ahead of the device.  Other shapes: the
// Riccati sweep (thread or wave per problem), separate evaluation / block / acceptance launches; inner-loop backtracking
// (matrix-core-bound stages) evaluates a second and later trial for the still-searching problems only.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "nempc_internal.h"
#include "activations.h"
#include "kernels_obj_impl.h"

#ifndef NEMPC_REG_FLOOR
#define NEMPC_REG_FLOOR 1e-3      // first damping level of a restarted Riccati sweep (see solver_lq_kernel)
#endif

namespace nempc {

namespace {

constexpr int INFO_LAM = 0, INFO_STEP = 1, INFO_AMAX = 2, INFO_G1 = 3, INFO_GINF = 4, INFO_D0 = 5, INFO_ZINF = 6,
              INFO_RESTARTS = 7,
              INFO_LSK = 8, INFO_LSA = 9,   // deferred backtracking: rejected trials in a row, step length to retry with,
              INFO_LSR = 10,                // Riccati restarts of the iteration that opened the search (sticky over retries)
              // non-monotone acceptance: the merit values of the last two accepted iterates, the penalty / barrier parameter
              // they were computed under, how many of them are valid
              INFO_PH1 = 11, INFO_PH2 = 12, INFO_PH3 = 13, INFO_PH4 = 14, INFO_PHNU = 15, INFO_PHMU = 16, INFO_PHN = 17,
              INFO_PHUP = 18,               // 1: the last accepted step went uphill in the merit (watchdog, see solver_merit_body)
              INFO_N = 19;

struct SolverArgs {
    int B, H, nx, nu, nin, n, m;
    const void* Z; const void* grad; const void* g; const void* tiles;   // at the current iterate
    const void* hblk;                                                    // (B,H,nin,nin) Lagrangian blocks
    void* lam; void* lamn;                                               // (B,m) multipliers: current, LQ costates
    const void* obj;  ObjOffsets oo;                                     // Qs, Rs live in the objective block
    const void* lb; const void* ub;                                      // (n) device, dtype T, +-inf allowed
    void* mu; void* pen; void* reg; void* alpha; void* phi0; void* dir;  // (B) per problem (pen = l1 penalty)
    int lq_attempts;      // Riccati sweeps a problem may try per iteration before it sits the iteration out
    int* hpub;            // pinned host memory the acceptance kernel publishes the convergence counter to: [0] iteration
                          // tag, [1] unconverged problems at that iteration, [2] first iteration at which none was left;
                          // inner-loop backtracking: [4] sequence number of the trial, [5] problems still searching after it
    int* n_done;          // acceptance launches of the inner loop: blocks finished (the last one publishes)
    int carry;            // the trial evaluation is a full one and becomes the next iterate's on acceptance (tiles_t, grad_t)
    const void *tiles_t, *grad_t;
    // ... with the Lagrangian blocks of the trial point too (one launch: blocks + evaluation, launch_rowhess_eval_mfma): the
    // step kernel writes the multipliers the trial point would have, lam + alpha (lamn - lam); hblk_t is carried over on
    // acceptance like the tiles
    void* lam_t; const void* hblk_t;
    // ... and the acceptance test of that trial point at the START of the next iteration's LQ kernel instead of in a launch
    // of its own (two launches per iteration): the LQ kernel's waves run solver_merit_body for their problems before they
    // stage.  The convergence counter alternates between two words by iteration parity -- the accept phase publishes and
    // clears the PREVIOUS iteration's word while this kernel's post-pass counts into its own
    int accept_first; int* n_active_prev; const void* gt_acc;
    int fuse_step;        // thread-per-problem Riccati kernel in LDS mode: it also does solver_step_kernel's work
    void* f_it; void* Zt_it;   // ... with the iterate's objective values (B) and the trial-point buffer (B,n)
    int* status; int* lsdone; int* n_active; int* n_pending;   // counters the host polls: unconverged problems / problems still backtracking
    int* iters_done; int cur_it;                                         // per problem: iteration at which it converged
    // bounds, primal-dual: zl / zu (B,n) multipliers of z >= lb / z <= ub, their steps, and the barrier diagonal the LQ
    // model adds to the Hessian (zl/(z-lb) + zu/(ub-z)); the barrier gradient -mu/(z-lb) + mu/(ub-z) is folded into grad
    void* zl; void* zu; void* dzl; void* dzu; void* alz; void* bh;
    int primal_dual;
    void* dz; void* info;                                                // (B,n), (B,INFO_N)
    void* Kst; void* kst; void* Pst; void* pst;                          // Riccati storage per problem
    void* tmp; size_t tmp_stride;                                        // global temporaries when LDS is too small
    int use_lds;                                                         // 1: per-problem working set staged in LDS
    int ppw;                                                             // problems per workgroup (LDS mode)
    int lds_stride;                                                      // elements per problem in LDS (odd)
    int spec, att_elems;   // thread-per-problem sweep: damping levels tried side by side (lanes per problem), elements of
                           // one attempt's region (gains, value function, temporaries) in the problem's LDS block
    double tol_g, tol_step, mu_min, mu_factor;
    double armijo_slack;                                                 // relative slack of the Armijo test (see merit kernel)
    int nonmono;                                                         // merit values of previous iterates the test may refer to (0: monotone)
    double reg_relax;                                                    // factor by which the Levenberg term is relaxed after a clean sweep
    double reg_raise;                                                    // ... and by which it is raised when a sweep meets a pivot that is not positive
    int max_ls;                                                          // halvings before the next LQ solve is damped
};

// ---- bounds by a PRIMAL-DUAL interior point (round 2; the first version was the primal log barrier, Hessian term
// mu/d^2).  With multipliers zl, zu of z >= lb, z <= ub the Newton system of the perturbed KKT conditions, reduced to
// the primal step, has  Hessian + diag(zl/dl + zu/du)  and gradient  grad f - mu/dl + mu/du  (dl = z-lb, du = ub-z);
// the dual steps follow from the primal one,  dzl = mu/dl - zl - (zl/dl) dz,  dzu = mu/du - zu + (zu/du) dz,  and are
// taken to their own fraction-to-the-boundary length.  The diagonal zl/dl stays moderate on a variable that sits at its
// bound with a small multiplier, where mu/dl^2 explodes -- the stragglers of the primal version were exactly the
// problems whose steps kept being cut by the boundary (NEMPC_SOLVER_TRACE).  Both terms keep the LQ structure.
// Runs before the LQ kernel: writes the diagonal bh and folds the barrier gradient into grad.
template <typename T>
__global__ __launch_bounds__(256) void solver_barrier_kernel(SolverArgs a) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0) *a.n_active = 0;      // counter of the convergence test that follows the LQ solve (a memset launch before)
    if (idx >= (size_t)a.B * a.n) return;
    const int b = (int)(idx / a.n), i = (int)(idx - (size_t)b * a.n);
    T ga = T(0), ha = T(0);
    const T mu = ((const T*)a.mu)[b];
    if (mu > T(0) && a.status[b] < 0) {
        const T z = ((const T*)a.Z)[idx], lo = ((const T*)a.lb)[i], hi = ((const T*)a.ub)[i];
        if (lo > -std::numeric_limits<T>::max()) {
            const T d = z - lo;
            ga -= mu / d;
            ha += a.primal_dual ? ((const T*)a.zl)[idx] / d : mu / (d * d);
        }
        if (hi < std::numeric_limits<T>::max()) {
            const T d = hi - z;
            ga += mu / d;
            ha += a.primal_dual ? ((const T*)a.zu)[idx] / d : mu / (d * d);
        }
    }
    ((T*)a.bh)[idx] = ha;
    ((T*)a.grad)[idx] += ga;
}

// after the LQ solve: dual steps and their fraction-to-the-boundary length (one wave per problem)
template <typename T>
__device__ __forceinline__ void solver_dual_body(const SolverArgs& a, int b, int lane, const T* __restrict__ zp,
                                                 const T* __restrict__ dzp, T mu, int status, const T* lbp = nullptr,
                                                 const T* ubp = nullptr, const T* zlp = nullptr, const T* zup = nullptr) {
    // zp / dzp: the problem's iterate and step (global memory, or the Riccati kernel's LDS copies); lbp / ubp / zlp / zup:
    // staged copies of the bounds and of this problem's bound multipliers, when the caller has them
    if (!a.primal_dual) return;
    const T tau = T(0.995);
    T amax = T(1);
    const T* lbv = lbp ? lbp : (const T*)a.lb;
    const T* ubv = ubp ? ubp : (const T*)a.ub;
    const T* zlv = zlp ? zlp : (const T*)a.zl + (size_t)b * a.n;
    const T* zuv = zup ? zup : (const T*)a.zu + (size_t)b * a.n;
    if (mu > T(0) && status < 0)
        for (int i = lane; i < a.n; i += 64) {
            const size_t idx = (size_t)b * a.n + i;
            const T z = zp[i], d = dzp[i], lo = lbv[i], hi = ubv[i];
            T sl = T(0), su = T(0);
            if (lo > -std::numeric_limits<T>::max()) {
                const T dl = z - lo, zl = zlv[i];
                sl = mu / dl - zl - (zl / dl) * d;
                if (sl < T(0)) amax = fmin(amax, -tau * zl / sl);
            }
            if (hi < std::numeric_limits<T>::max()) {
                const T du = hi - z, zu = zuv[i];
                su = mu / du - zu + (zu / du) * d;
                if (su < T(0)) amax = fmin(amax, -tau * zu / su);
            }
            ((T*)a.dzl)[idx] = sl;
            ((T*)a.dzu)[idx] = su;
        }
    amax = (T)wave_min_lane0((double)amax);
    if (lane == 0) ((T*)a.alz)[b] = amax;
}

// The barrier terms of variable i of problem b at the iterate -- what solver_barrier_kernel writes, for the LQ kernels
// that stage their working set in LDS and fold it in while staging (one launch less per iteration): ga joins the
// gradient, ha is the diagonal of the LQ model
template <typename T>
__device__ __forceinline__ void barrier_terms_of(bool primal_dual, T mu, int status, T z, T lo, T hi, T zl, T zu, T& ga, T& ha) {
    ga = T(0); ha = T(0);
    if (mu > T(0) && status < 0) {
        if (lo > -std::numeric_limits<T>::max()) {
            const T d = z - lo;
            ga -= mu / d;
            ha += primal_dual ? zl / d : mu / (d * d);
        }
        if (hi < std::numeric_limits<T>::max()) {
            const T d = hi - z;
            ga += mu / d;
            ha += primal_dual ? zu / d : mu / (d * d);
        }
    }
}
template <typename T>
__device__ __forceinline__ void solver_barrier_terms(const SolverArgs& a, int b, int i, T& ga, T& ha) {
    ga = T(0); ha = T(0);
    const T mu = ((const T*)a.mu)[b];
    const int status = a.status[b];
    if (mu > T(0) && status < 0) {
        const size_t idx = (size_t)b * a.n + i;
        const T z = ((const T*)a.Z)[idx], lo = ((const T*)a.lb)[i], hi = ((const T*)a.ub)[i];
        const bool lof = lo > -std::numeric_limits<T>::max(), hif = hi < std::numeric_limits<T>::max();
        barrier_terms_of<T>(a.primal_dual, mu, status, z, lo, hi, a.primal_dual && lof ? ((const T*)a.zl)[idx] : T(0),
                            a.primal_dual && hif ? ((const T*)a.zu)[idx] : T(0), ga, ha);
    }
}

template <typename T>
struct StepInfo {   // what the Riccati kernel leaves in the info row (read from there, or handed over in registers) ...
    T ginf, step, zinf, lam, d0, amax, lsk, lsa, restarts;
    // ... and the problem's scalars, loaded in one go ahead of their use (each was a dependent global round trip)
    T lsr, mu, nu, f;
    int status;
};
template <typename T>
__device__ __forceinline__ void solver_merit0_body(const SolverArgs& a, int b, int lane, const T* __restrict__ f,
                                                   const T* __restrict__ zp, const T* __restrict__ gp, const StepInfo<T>& si,
                                                   int& lsd, T& al, const T* lbp = nullptr, const T* ubp = nullptr);
template <typename T>
__device__ __forceinline__ void solver_merit_body(const SolverArgs& a, const T* __restrict__ Zt, const T* __restrict__ gt,
                                                  const T* __restrict__ ft, T* __restrict__ Zcur, int last_ls,
                                                  const int* __restrict__ list_in, int* __restrict__ list_out,
                                                  int publish, int slot, int lane, int* n_active_word, int tag, int soc = 0);

#ifdef NEMPC_LQ_STAMPS
__device__ long long nempc_lq_stamps[16];
#define LQ_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) nempc_lq_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define LQ_STAMP(i) do { } while (0)
#endif

// 1 / sqrt(d) for a pivot d > 1e-12: hardware estimate + two coupled Newton steps (double; ~1 ulp) or one (float).  The
// IEEE sqrt followed by the IEEE division is ~32 instructions of a chain that a lone lane issues one per ~10 cycles.
__device__ __forceinline__ double inv_sqrt_pos(double d) {
    const double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = fma(-g, h, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-g, h, 0.5);
    h = fma(h, r, h);
    return h + h;
}
__device__ __forceinline__ float inv_sqrt_pos(float d) {
    const float y = __builtin_amdgcn_rsqf(d);
    const float r = fmaf(-d * y, y, 1.0f);
    return fmaf(0.5f * y, r, y);
}

// ---- Parallel-in-time LQ solve for 2-state / 1-control stages (one wave per problem, one LANE per stage).
// The backward Riccati recursion is a chain of H dependent stages; a lone lane issues its ~140 instructions per stage one
// per 5 cycles (11 us of the iteration at H = 20).  The recursion is also a product of associative elements (Sarkka &
// Garcia-Fernandez, "Temporal parallelization of dynamic programming and linear quadratic control", IEEE TAC 2023): the
// conditional value function of a run of stages k..i-1,
//     V(x_k, x_i) = max_l  1/2 x_k' J x_k - eta' x_k - 1/2 l' C l - l' (x_i - A x_k - b),
// is closed under composition, (A, b, C, eta, J)_{k,j} (x) (A, b, C, eta, J)_{j,i} below, so a suffix scan over the stages
// (log2(H+1) combination steps, every lane working) gives every stage its value function, and the closed-loop rollout
// dx_{t+1} = (A + B K) dx_t + (c + B k) is a prefix scan of affine maps.
// What makes it usable here, where the exact Lagrangian blocks make stage Hessians indefinite:
//  * a stage element needs its OWN control Hessian U = Rs + W_uu + barrier + damping inverted, while the sweep only needs
//    Quu = U + B' P B > 0.  The cost 1/2 s_t |dx_{t+1}|^2 is added to stage t (through its dynamics) and taken back from the
//    state entering stage t+1 -- an exact rewriting -- with s_t = 1, raised to 2 |U| / |B|^2 where U + |B|^2 / 2 < 0: on
//    random indefinite problems the scan then goes through exactly when the sweep does (tools/lq_scan_proto.py: 0
//    disagreements in 5,400 solves, results equal to 1e-11);
//  * the sweep's own test decides: with every P_t known, Quu_t > 1e-12 is checked for all stages at once; a level that
//    fails is repeated with ten times the damping, the levels of a round side by side in lane segments of H + 1.
template <typename T>
struct LqEl { T a00, a01, a10, a11, b0, b1, c00, c01, c11, e0, e1, j00, j01, j11; };

__device__ __forceinline__ double lq_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(e, r, r);
    e = fma(-x, r, 1.0);
    return fma(e, r, r);
}

template <typename T>
__device__ __forceinline__ LqEl<T> lq_shfl_down(const LqEl<T>& v, int d) {
    LqEl<T> r;
    r.a00 = __shfl_down(v.a00, d, 64); r.a01 = __shfl_down(v.a01, d, 64); r.a10 = __shfl_down(v.a10, d, 64);
    r.a11 = __shfl_down(v.a11, d, 64); r.b0 = __shfl_down(v.b0, d, 64); r.b1 = __shfl_down(v.b1, d, 64);
    r.c00 = __shfl_down(v.c00, d, 64); r.c01 = __shfl_down(v.c01, d, 64); r.c11 = __shfl_down(v.c11, d, 64);
    r.e0 = __shfl_down(v.e0, d, 64); r.e1 = __shfl_down(v.e1, d, 64);
    r.j00 = __shfl_down(v.j00, d, 64); r.j01 = __shfl_down(v.j01, d, 64); r.j11 = __shfl_down(v.j11, d, 64);
    return r;
}

// (first: stages k..j-1) (x) (second: stages j..i-1)
template <typename T>
__device__ __forceinline__ LqEl<T> lq_combine(const LqEl<T>& i, const LqEl<T>& j) {
    // M = (I + C_i J_j)^-1
    const T n00 = fma(i.c00, j.j00, fma(i.c01, j.j01, T(1))), n01 = fma(i.c00, j.j01, i.c01 * j.j11);
    const T n10 = fma(i.c01, j.j00, i.c11 * j.j01), n11 = fma(i.c01, j.j01, fma(i.c11, j.j11, T(1)));
    const T id = lq_rcp(fma(n00, n11, -n01 * n10));
    const T m00 = n11 * id, m01 = -n01 * id, m10 = -n10 * id, m11 = n00 * id;
    // AM = A_j M
    const T am00 = fma(j.a00, m00, j.a01 * m10), am01 = fma(j.a00, m01, j.a01 * m11);
    const T am10 = fma(j.a10, m00, j.a11 * m10), am11 = fma(j.a10, m01, j.a11 * m11);
    LqEl<T> r;
    r.a00 = fma(am00, i.a00, am01 * i.a10); r.a01 = fma(am00, i.a01, am01 * i.a11);
    r.a10 = fma(am10, i.a00, am11 * i.a10); r.a11 = fma(am10, i.a01, am11 * i.a11);
    // b = AM (b_i + C_i eta_j) + b_j
    const T v0 = fma(i.c00, j.e0, fma(i.c01, j.e1, i.b0)), v1 = fma(i.c01, j.e0, fma(i.c11, j.e1, i.b1));
    r.b0 = fma(am00, v0, fma(am01, v1, j.b0)); r.b1 = fma(am10, v0, fma(am11, v1, j.b1));
    // C = AM C_i A_j' + C_j
    const T t00 = fma(am00, i.c00, am01 * i.c01), t01 = fma(am00, i.c01, am01 * i.c11);
    const T t10 = fma(am10, i.c00, am11 * i.c01), t11 = fma(am10, i.c01, am11 * i.c11);
    r.c00 = fma(t00, j.a00, fma(t01, j.a01, j.c00));
    r.c01 = fma(T(0.5), fma(t00, j.a10, t01 * j.a11) + fma(t10, j.a00, t11 * j.a01), j.c01);
    r.c11 = fma(t10, j.a10, fma(t11, j.a11, j.c11));
    // eta = A_i' M' (eta_j - J_j b_i) + eta_i
    const T w0 = j.e0 - fma(j.j00, i.b0, j.j01 * i.b1), w1 = j.e1 - fma(j.j01, i.b0, j.j11 * i.b1);
    const T x0 = fma(m00, w0, m10 * w1), x1 = fma(m01, w0, m11 * w1);
    r.e0 = fma(i.a00, x0, fma(i.a10, x1, i.e0)); r.e1 = fma(i.a01, x0, fma(i.a11, x1, i.e1));
    // J = A_i' (M' J_j) A_i + J_i
    const T y00 = fma(m00, j.j00, m10 * j.j01), y01 = fma(m00, j.j01, m10 * j.j11);
    const T y10 = fma(m01, j.j00, m11 * j.j01), y11 = fma(m01, j.j01, m11 * j.j11);
    const T z00 = fma(y00, i.a00, y01 * i.a10), z01 = fma(y00, i.a01, y01 * i.a11);
    const T z10 = fma(y10, i.a00, y11 * i.a10), z11 = fma(y10, i.a01, y11 * i.a11);
    r.j00 = fma(i.a00, z00, fma(i.a10, z10, i.j00));
    r.j01 = fma(T(0.5), fma(i.a00, z01, i.a10 * z11) + fma(i.a01, z00, i.a11 * z10), i.j01);
    r.j11 = fma(i.a01, z01, fma(i.a11, z11, i.j11));
    return r;
}

// One problem on one wave: its working set is staged in the LDS block `blk` (offsets as in solver_lq_kernel); the step and
// the costates are left in blk[Ldz..], blk[Llam..], the restart count (negative: sat out) in blk[Lbh].
template <typename T>
__device__ __forceinline__ void lq_scan_problem(const SolverArgs& a, int b, int ln, T* __restrict__ blk, int Lgr, int Lgc,
                                                int Ltl, int LW, int Llam, int Ldz, int Lbh, int status_b, T reg_b) {
    // (status_b, reg_b: the problem's status and damping, requested by the caller ahead of the staging)
    constexpr int nx = 2, nin = 3;
    const int H = a.H, n = a.n, uo = H * nx, seg = H + 1;
    const int specs = a.spec;
    T* info = (T*)a.info + (size_t)b * INFO_N;
    if (status_b >= 0) {          // finished problem: zero step, multipliers unchanged
        const T* lcur = (const T*)a.lam + (size_t)b * a.m;
        for (int i = ln; i < n; i += 64) blk[Ldz + i] = T(0);
        for (int i = ln; i < H * nx; i += 64) blk[Llam + i] = lcur[i];
        return;
    }
    int lv = 0;
    for (int k = 1; k < specs; ++k) lv += ln >= k * seg ? 1 : 0;
    const int e = ln - lv * seg;                    // stage of this lane (e == H: the terminal element)
    const bool active = ln < specs * seg, is_stage = active && e < H;
    const int t = is_stage ? e : H - 1;             // (the terminal lane reads the last state's entries)
    // inputs of this lane's stage: all read before anything of the block is written
    const T* At = blk + Ltl + t * nx * nin;
    const T* Wt = blk + LW + t * nin * nin;
    const T A00 = At[0], A01 = At[1], B0 = At[2], A10 = At[3], A11 = At[4], B1 = At[5];
    const T Wxx00 = Wt[0], Wxx01 = Wt[1], Wxx11 = Wt[4], Wux0 = Wt[6], Wux1 = Wt[7], Wuu = Wt[8];
    const T c0 = blk[Lgc + t * nx], c1 = blk[Lgc + t * nx + 1];
    const T gru = blk[Lgr + uo + t], bhu = blk[Lbh + uo + t];
    const int sx = is_stage ? (e > 0 ? e - 1 : 0) : H - 1;       // state entering the stage (none for stage 0)
    const T grx0 = blk[Lgr + sx * nx], grx1 = blk[Lgr + sx * nx + 1], bhx0 = blk[Lbh + sx * nx], bhx1 = blk[Lbh + sx * nx + 1];
    const T* Qs = (const T*)a.obj + a.oo.Qs;
    const T* QTs = (const T*)a.obj + a.oo.QTs;
    const T Rs = ((const T*)a.obj)[a.oo.Rs];
    const T q00 = Qs[0], q01 = T(0.5) * (Qs[1] + Qs[2]), q11 = Qs[3];
    const T qt00 = QTs[0], qt01 = T(0.5) * (QTs[1] + QTs[2]), qt11 = QTs[3];
    const T reg_in = reg_b;
    const T bb = fma(B0, B0, B1 * B1);
    int restarts = a.lq_attempts;
    bool any = false;
    T reg = reg_in;
    for (int base = 0; base < a.lq_attempts && !any; base += specs) {
        const int aj = base + lv;
        reg = reg_in;
        for (int k = 0; k < aj; ++k) reg = fmax(reg * (T)a.reg_raise, T(NEMPC_REG_FLOOR));
        bool ok = false;
        // shift of the state this stage produces (see above); the state entering it carries the previous stage's
        const T U = Rs + Wuu + bhu + reg;
        T sg = T(1);
        if (U < T(-0.5) * bb) sg = T(2) * (-U) * lq_rcp(fmax(bb, T(1e-300)));
        T P00, P01, P11, p0, p1, Quu, quf, Qux0, Qux1;
        // Second pass where the first one was ill-conditioned: a combination inverts I + C J, whose eigenvalue along B is
        // Quu / U' at the last level -- with U' = U + s |B|^2 far from Quu = U + B'PB (an objective without curvature:
        // P ~ 1e-6 against s = 1) the first pass loses log10(U' / Quu) digits.  Its P is good enough to choose s = B'PB / |B|^2,
        // i.e. U' = Quu, and the second pass is then as accurate as the sweep.  Well-scaled problems never take it.
        for (int pass = 0; pass < 2; ++pass) {
        ok = active && aj < a.lq_attempts;
        T sg_in = __shfl_up(sg, 1, 64);
        if (e == 0) sg_in = T(0);
        T Up = T(1);
        LqEl<T> E;
        if (is_stage) {
            Up = fma(sg, bb, U);
            ok = ok && Up > T(1e-12);
            const T iu = lq_rcp(Up);
            const T wu0 = fma(sg, fma(B0, A00, B1 * A10), Wux0), wu1 = fma(sg, fma(B0, A01, B1 * A11), Wux1);
            const T qu = fma(sg, fma(B0, c0, B1 * c1), gru);
            // state cost of the entering state: objective weight + Lagrangian block + barrier + s A'A - s_in I
            const T x00 = q00 + Wxx00 + bhx0 + sg * fma(A00, A00, A10 * A10) - sg_in;
            const T x01 = q01 + Wxx01 + sg * fma(A00, A01, A10 * A11);
            const T x11 = q11 + Wxx11 + bhx1 + sg * fma(A01, A01, A11 * A11) - sg_in;
            const T qx0 = fma(sg, fma(A00, c0, A10 * c1), grx0), qx1 = fma(sg, fma(A01, c0, A11 * c1), grx1);
            const T bi0 = B0 * iu, bi1 = B1 * iu;
            E.a00 = fma(-bi0, wu0, A00); E.a01 = fma(-bi0, wu1, A01); E.a10 = fma(-bi1, wu0, A10); E.a11 = fma(-bi1, wu1, A11);
            E.b0 = fma(-bi0, qu, c0); E.b1 = fma(-bi1, qu, c1);
            E.c00 = bi0 * B0; E.c01 = bi0 * B1; E.c11 = bi1 * B1;
            const T wi0 = wu0 * iu, wi1 = wu1 * iu;
            E.j00 = fma(-wi0, wu0, x00); E.j01 = fma(-wi0, wu1, x01); E.j11 = fma(-wi1, wu1, x11);
            E.e0 = fma(wi0, qu, -qx0); E.e1 = fma(wi1, qu, -qx1);
        } else {
            E.a00 = E.a01 = E.a10 = E.a11 = E.b0 = E.b1 = E.c00 = E.c01 = E.c11 = T(0);
            E.j00 = qt00 + bhx0 - sg_in; E.j01 = qt01; E.j11 = qt11 + bhx1 - sg_in;
            E.e0 = -grx0; E.e1 = -grx1;
        }
        // suffix scan: after the step of distance d lane e holds stages e .. e + 2d - 1 (cut at the terminal)
        for (int d = 1; d < seg; d <<= 1) {
            const LqEl<T> nb = lq_shfl_down(E, d);
            if (active && e + d <= H) E = lq_combine(E, nb);
        }
        // value function of the state this stage produces: the suffix from the next lane, shift undone
        P00 = __shfl_down(E.j00, 1, 64) + sg; P01 = __shfl_down(E.j01, 1, 64); P11 = __shfl_down(E.j11, 1, 64) + sg;
        p0 = -__shfl_down(E.e0, 1, 64); p1 = -__shfl_down(E.e1, 1, 64);
        // the sweep's stage: Quu = U + B'PB, qu = gu + B'(Pc + p), Qux = W_ux + B'PA
        const T PB0 = fma(P00, B0, P01 * B1), PB1 = fma(P01, B0, P11 * B1);
        const T Pc0 = fma(P00, c0, fma(P01, c1, p0)), Pc1 = fma(P01, c0, fma(P11, c1, p1));
        Quu = fma(B0, PB0, fma(B1, PB1, U));
        quf = fma(B0, Pc0, fma(B1, Pc1, gru));
        Qux0 = fma(PB0, A00, fma(PB1, A10, Wux0)); Qux1 = fma(PB0, A01, fma(PB1, A11, Wux1));
        ok = ok && (!is_stage || Quu > T(1e-12));
        if (pass == 0) {
            const bool far = is_stage && ok && (Quu < T(1.0 / 64) * Up || Quu > T(64) * Up);
            if (!__any(far)) break;
            if (is_stage && ok && bb > T(1e-300)) sg = (Quu - U) * lq_rcp(bb);
        }
        }
        const T iq = lq_rcp(Quu);
        const T kv = -quf * iq, K0 = e == 0 ? T(0) : -Qux0 * iq, K1 = e == 0 ? T(0) : -Qux1 * iq;
        // which level is used: the first whose stages all went through
        const unsigned long long okm = __ballot(ok || (active && !is_stage));
        int win = -1;
        for (int jj = specs - 1; jj >= 0; --jj) {
            const unsigned long long m = (seg >= 64 ? ~0ull : ((1ull << seg) - 1ull)) << (jj * seg);
            if (base + jj < a.lq_attempts && (okm & m) == m) win = jj;
        }
        if (win < 0) continue;
        any = true;
        restarts = base + win;
        // closed-loop rollout dx_{t+1} = F dx_t + f, dx_0 = 0: prefix scan of the affine maps
        T F00 = e == 0 ? T(0) : fma(B0, K0, A00), F01 = e == 0 ? T(0) : fma(B0, K1, A01);
        T F10 = e == 0 ? T(0) : fma(B1, K0, A10), F11 = e == 0 ? T(0) : fma(B1, K1, A11);
        T f0 = fma(B0, kv, c0), f1 = fma(B1, kv, c1);
        for (int d = 1; d < H; d <<= 1) {
            const T g00 = __shfl_up(F00, d, 64), g01 = __shfl_up(F01, d, 64), g10 = __shfl_up(F10, d, 64), g11 = __shfl_up(F11, d, 64);
            const T h0 = __shfl_up(f0, d, 64), h1 = __shfl_up(f1, d, 64);
            if (is_stage && e - d >= 0) {
                const T nf0 = fma(F00, h0, fma(F01, h1, f0)), nf1 = fma(F10, h0, fma(F11, h1, f1));
                const T n00 = fma(F00, g00, F01 * g10), n01 = fma(F00, g01, F01 * g11);
                const T n10 = fma(F10, g00, F11 * g10), n11 = fma(F10, g01, F11 * g11);
                F00 = n00; F01 = n01; F10 = n10; F11 = n11; f0 = nf0; f1 = nf1;
            }
        }
        T dx0 = __shfl_up(f0, 1, 64), dx1 = __shfl_up(f1, 1, 64);
        if (e == 0) { dx0 = T(0); dx1 = T(0); }
        const T du = fma(K0, dx0, fma(K1, dx1, kv));
        // (the new state from the stage's own linearisation, as the sweep writes it)
        const T xn0 = fma(A00, dx0, fma(A01, dx1, fma(B0, du, c0))), xn1 = fma(A10, dx0, fma(A11, dx1, fma(B1, du, c1)));
        if (is_stage && lv == win) {
            blk[Ldz + e * nx] = xn0; blk[Ldz + e * nx + 1] = xn1; blk[Ldz + uo + e] = du;
            blk[Llam + e * nx] = fma(P00, xn0, fma(P01, xn1, p0)); blk[Llam + e * nx + 1] = fma(P01, xn0, fma(P11, xn1, p1));
            if (e == 0) {
                ((T*)a.reg)[b] = reg;
                info[INFO_RESTARTS] = (T)restarts;
                blk[Lbh] = (T)restarts;
            }
        }
    }
    if (!any) {
        // out of attempts: no step this iteration, the damping keeps what it has climbed to (as in the sweep kernel)
        const T* lcur = (const T*)a.lam + (size_t)b * a.m;
        for (int i = ln; i < n; i += 64) blk[Ldz + i] = T(0);
        for (int i = ln; i < H * nx; i += 64) blk[Llam + i] = lcur[i];
        if (ln == 0) {
            reg = reg_in;
            for (int k = 0; k < a.lq_attempts; ++k) reg = fmax(reg * (T)a.reg_raise, T(NEMPC_REG_FLOOR));
            ((T*)a.reg)[b] = reg;
            info[INFO_STEP] = std::numeric_limits<T>::max();
            info[INFO_RESTARTS] = (T)(-a.lq_attempts);
            blk[Lbh] = (T)(-a.lq_attempts);
        }
    }
}

// One thread per problem.  The sweep is a long chain of tiny dependent matrix products: straight from global
// memory every operand costs a ~600-cycle round trip (measured 720 us per call at B=1024, 2/1, H=20).  So a
// workgroup first copies the whole working set of its `ppw` problems (iterate, gradient, defects, tiles,
// Lagrangian blocks) into LDS with coalesced loads, the first ppw lanes then run their sweeps out of LDS
// (per-problem stride odd -> conflict-free), and the step / costates are copied back cooperatively.
// Problems whose working set does not fit in LDS fall back to global temporaries (ppw = 64, use_lds = 0).
// NX, NU > 0: dimensions fixed at compile time -- every small loop unrolls and the temporaries live in registers
// (2/1 and 6/3, the BASELINE shapes); NX = NU = 0: runtime dimensions, temporaries in LDS / global memory.
// SCAN (2/1 stages, LDS mode): the sweeps are replaced by lq_scan_problem, a wave per problem; staging and the post-pass
// are the same code.
template <typename T, int NX, int NU, bool LDS, bool SCAN = false>
__global__ __launch_bounds__(256) void solver_lq_kernel(SolverArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);
    const int lane = threadIdx.x;
    constexpr bool FIX = NX > 0;
    const int H = a.H, nx = FIX ? NX : a.nx, nu = FIX ? NU : a.nu, nin = nx + nu, n = a.n;
    const int ppw = LDS ? a.ppw : 64;
    const int spec = LDS ? a.spec : 1;
    const int aj0 = lane < ppw * spec ? lane / ppw : 0;      // which of the side-by-side damping levels this lane sweeps
    const int pl = lane < ppw * spec ? lane - aj0 * ppw : lane;   // its problem's slot in the workgroup
    // element offsets of the per-problem LDS block: the arrays every attempt reads, the results, then one region per attempt
    const int Lz = 0, Lgr = Lz + n, Lgc = Lgr + n, Ltl = Lgc + H * nx, LW = Ltl + H * nx * nin, Llam = LW + H * nin * nin,
              Ldz = Llam + H * nx, Lbh = Ldz + n, Llc = Lbh + n, Lzl = Llc + H * nx, Lzu = Lzl + n,
              Latt = Lzu + n + aj0 * a.att_elems, LK = Latt, Lk = LK + H * nu * nx,
              LP = Lk + H * nu, Lp = LP + H * nx * nx, Ltmp = Lp + H * nx;
    const int b = blockIdx.x * ppw + pl;
    const bool mine = lane < ppw * spec && b < a.B;
    // cooperative staging, all four waves loading: a wave takes a problem (with fewer than three problems in the
    // workgroup, two or all four waves share one), its lanes the elements -- index arithmetic without divisions (a flat
    // index over problems x elements cost a ~40-instruction integer division per element, in a phase that a wave runs
    // once, at one instruction per ~10 cycles)
    const int b0 = blockIdx.x * ppw;
    const int np = a.B - b0 < ppw ? a.B - b0 : ppw;
    const int nthr = blockDim.x;
    const int wvu = __builtin_amdgcn_readfirstlane(lane >> 6);
    const int lgw = np >= 3 ? 0 : (np == 2 ? 1 : 2);                 // log2(waves per problem), 4 waves
    const int st_p0 = wvu >> lgw, st_pstep = 4 >> lgw, st_e0 = ((wvu & ((1 << lgw) - 1)) << 6) + (lane & 63), st_estep = 64 << lgw;
    // bounds (the same for every problem): one copy per workgroup behind the problem blocks; the post-pass reads them there
    T* const lds_lb = lds + (size_t)ppw * a.lds_stride;
    T* const lds_ub = lds_lb + n;
    auto stage_in = [&](const T* __restrict__ src, int per, int src_stride, int loff) {
        for (int pp = st_p0; pp < np; pp += st_pstep) {
            const T* sp = src + (size_t)(b0 + pp) * src_stride;
            T* dp = lds + (size_t)pp * a.lds_stride + loff;
            for (int e = st_e0; e < per; e += st_estep) dp[e] = sp[e];
        }
    };
    LQ_STAMP(0);
    if constexpr (LDS) {
        if (a.accept_first) {
            // acceptance test of the previous iteration's trial point for this workgroup's problems (a wave per problem), then
            // everything below reads the iterate it left: the staging waits behind the barrier
            for (int pp = wvu; pp < np; pp += 4)
                solver_merit_body<T>(a, (const T*)a.Zt_it, (const T*)a.gt_acc, (const T*)nullptr, (T*)a.Z, 2, (const int*)nullptr,
                                     (int*)nullptr, 1, b0 + pp, lane & 63, a.n_active_prev, a.cur_it);
            __syncthreads();
        }
    }
    // The scalars of the problem this wave solves (scan) and post-processes first -- status, damping, barrier parameter,
    // penalty, objective value, backtracking memory: requested with the staging loads, each was a ~2,000-cycle round trip of
    // its own where it is used (status before the solve, the others at the top of the post-pass)
    const int pf_b = b0 + wvu;
    const bool pf_ok = LDS && wvu < np;
    int pf_status = -1;
    T pf_reg = T(0), pf_mu = T(0), pf_pen = T(0), pf_f = T(0), pf_lsk = T(0), pf_lsa = T(0), pf_lsr = T(0);
    if (pf_ok) {
        pf_status = a.status[pf_b];
        pf_reg = ((const T*)a.reg)[pf_b];
        if (a.fuse_step) {
            const T* infp = (const T*)a.info + (size_t)pf_b * INFO_N;
            pf_mu = ((const T*)a.mu)[pf_b]; pf_pen = ((const T*)a.pen)[pf_b]; pf_f = ((const T*)a.f_it)[pf_b];
            pf_lsk = infp[INFO_LSK]; pf_lsa = infp[INFO_LSA]; pf_lsr = infp[INFO_LSR];
        }
    }
    if (LDS) {
        // counters of the convergence test (zeroed here unless the test runs inside this kernel: then the previous
        // iteration's acceptance kernel did it) and of the trial's acceptance test
        if (blockIdx.x == 0 && lane == 0) {
            if (a.fuse_step) *a.n_pending = 0; else *a.n_active = 0;
        }
        // Small stages (every array a few wavefuls: the BASELINE 2/1 shapes): ALL loads of a problem are issued before the
        // first LDS write, one global round trip instead of one per array and three inside the barrier terms
        const int Ntl = H * nx * nin, NW = H * nin * nin;
        const bool one_shot = n <= st_estep && Ntl <= 2 * st_estep && NW <= 3 * st_estep;
        if (one_shot) {
            for (int pp = st_p0; pp < np; pp += st_pstep) {
                const int bq = b0 + pp, e = st_e0, e1 = e + st_estep, e2 = e1 + st_estep;
                const bool in_n = e < n, pdl = a.primal_dual != 0;
                const T mu_q = ((const T*)a.mu)[bq];
                const int st_q = a.status[bq];
                const T vz = in_n ? ((const T*)a.Z)[(size_t)bq * n + e] : T(0);
                const T vgr = in_n ? ((const T*)a.grad)[(size_t)bq * n + e] : T(0);
                const T vlo = in_n ? ((const T*)a.lb)[e] : -std::numeric_limits<T>::max();
                const T vhi = in_n ? ((const T*)a.ub)[e] : std::numeric_limits<T>::max();
                const T vzl = in_n && pdl ? ((const T*)a.zl)[(size_t)bq * n + e] : T(0);
                const T vzu = in_n && pdl ? ((const T*)a.zu)[(size_t)bq * n + e] : T(0);
                const T vgc = e < H * nx ? ((const T*)a.g)[(size_t)bq * a.m + e] : T(0);
                const T vlc = e < H * nx && a.lam_t ? ((const T*)a.lam)[(size_t)bq * a.m + e] : T(0);
                const T* tq = (const T*)a.tiles + (size_t)bq * Ntl;
                const T* wq = (const T*)a.hblk + (size_t)bq * NW;
                const T vt0 = e < Ntl ? tq[e] : T(0), vt1 = e1 < Ntl ? tq[e1] : T(0);
                const T vw0 = e < NW ? wq[e] : T(0), vw1 = e1 < NW ? wq[e1] : T(0), vw2 = e2 < NW ? wq[e2] : T(0);
                T* blkp = lds + (size_t)pp * a.lds_stride;
                if (e < Ntl) blkp[Ltl + e] = vt0;
                if (e1 < Ntl) blkp[Ltl + e1] = vt1;
                if (e < NW) blkp[LW + e] = vw0;
                if (e1 < NW) blkp[LW + e1] = vw1;
                if (e2 < NW) blkp[LW + e2] = vw2;
                if (e < H * nx) { blkp[Lgc + e] = vgc; blkp[Llc + e] = vlc; }
                if (in_n) {
                    T ga, ha;
                    barrier_terms_of<T>(pdl, mu_q, st_q, vz, vlo, vhi, vzl, vzu, ga, ha);
                    blkp[Lz + e] = vz;
                    blkp[Lgr + e] = vgr + ga;
                    blkp[Lbh + e] = ha;
                    blkp[Lzl + e] = vzl; blkp[Lzu + e] = vzu;
                    lds_lb[e] = vlo; lds_ub[e] = vhi;          // (every problem's stagers write the same values)
                }
            }
            __syncthreads();
        } else {
        stage_in((const T*)a.Z, n, n, Lz);
        // gradient + barrier gradient, barrier diagonal: computed while staging (no solver_barrier_kernel launch)
        for (int pp = st_p0; pp < np; pp += st_pstep) {
            T* blkp = lds + (size_t)pp * a.lds_stride;
            for (int e = st_e0; e < n; e += st_estep) {
                T ga, ha;
                solver_barrier_terms<T>(a, b0 + pp, e, ga, ha);
                T gv = ((const T*)a.grad)[(size_t)(b0 + pp) * n + e];
                gv += ga;
                blkp[Lgr + e] = gv;
                blkp[Lbh + e] = ha;
            }
        }
        stage_in((const T*)a.g, H * nx, a.m, Lgc);
        if (a.primal_dual) { stage_in((const T*)a.zl, n, n, Lzl); stage_in((const T*)a.zu, n, n, Lzu); }
        for (int e = lane; e < n; e += nthr) { lds_lb[e] = ((const T*)a.lb)[e]; lds_ub[e] = ((const T*)a.ub)[e]; }
        if (a.lam_t) stage_in((const T*)a.lam, H * nx, a.m, Llc);
        stage_in((const T*)a.tiles, H * nx * nin, H * nx * nin, Ltl);
        stage_in((const T*)a.hblk, H * nin * nin, H * nin * nin, LW);
        __syncthreads();
        }
    }
    LQ_STAMP(1);
    if constexpr (SCAN) {
        static_assert(LDS && NX == 2 && NU == 1, "the scan is written for 2-state / 1-control stages staged in LDS");
        for (int pp = wvu; pp < np; pp += 4)
            lq_scan_problem<T>(a, b0 + pp, lane & 63, lds + (size_t)pp * a.lds_stride, Lgr, Lgc, Ltl, LW, Llam, Ldz, Lbh,
                               pp == wvu ? pf_status : a.status[b0 + pp], pp == wvu ? pf_reg : ((const T*)a.reg)[b0 + pp]);
    } else {
    T* dzg = mine ? (T*)a.dz + (size_t)b * n : nullptr;
    if (mine && a.status[b] >= 0) {
        // finished problem: zero step (written through the copy-out below in LDS mode)
        if (aj0 == 0) {
            T* dzp = LDS ? lds + (size_t)pl * a.lds_stride + Ldz : dzg;
            for (int i = 0; i < n; ++i) dzp[i] = T(0);
            if (LDS) {
                T* lp = lds + (size_t)pl * a.lds_stride + Llam;
                const T* lcur = (const T*)a.lam + (size_t)b * a.m;
                for (int i = 0; i < H * nx; ++i) lp[i] = lcur[i];
            }
        }
    } else if (mine) {
    T* info = (T*)a.info + (size_t)b * INFO_N;
    T* blk = lds + (size_t)pl * a.lds_stride;
    T* tb = LDS ? blk + Ltmp : (T*)a.tmp + b;
    const size_t ts = LDS ? 1 : a.tmp_stride;
    T tmpv[FIX ? (3 * NX * NX + 3 * NX * NU + NU * NU + 5 * NX + 3 * NU) : 1];
#define TMP(e) (*(FIX ? &tmpv[FIX ? (e) : 0] : &tb[(size_t)(e) * ts]))
    int off = 0;
    const int oP = off; off += nx * nx;
    const int oPA = off; off += nx * nx;
    const int oPn = off; off += nx * nx;
    const int oPB = off; off += nx * nu;
    const int oQux = off; off += nu * nx;
    const int oK = off; off += nu * nx;
    const int oQuu = off; off += nu * nu;
    const int oPc = off; off += nx;
    const int op = off; off += nx;
    const int opn = off; off += nx;
    const int odx = off; off += nx;
    const int odxn = off; off += nx;
    const int oqu = off; off += nu;
    const int okv = off; off += nu;
    const int odu = off; off += nu;

    const T* z = LDS ? blk + Lz : (const T*)a.Z + (size_t)b * n;
    const T* gr = LDS ? blk + Lgr : (const T*)a.grad + (size_t)b * n;
    const T* gc = LDS ? blk + Lgc : (const T*)a.g + (size_t)b * a.m;
    const T* tl = LDS ? blk + Ltl : (const T*)a.tiles + (size_t)b * H * nx * nin;
    const T* Wb = LDS ? blk + LW : (const T*)a.hblk + (size_t)b * H * nin * nin;
    const T* Qs = (const T*)a.obj + a.oo.Qs;
    const T* QTs = (const T*)a.obj + a.oo.QTs;
    const T* Rs = (const T*)a.obj + a.oo.Rs;
    // small stages: the objective's weights live in registers for the sweeps (read from global memory inside the stage
    // loop they were two ~500-cycle waits per stage: the stores of the sweep keep the compiler from hoisting them)
    constexpr bool REGTAB = FIX && NX * NX + NU * NU <= 16;
    T Qr[REGTAB ? NX * NX : 1], Rr[REGTAB ? NU * NU : 1];
    if (REGTAB) {
        #pragma unroll
        for (int e = 0; e < NX * NX; ++e) Qr[REGTAB ? e : 0] = Qs[e];
        #pragma unroll
        for (int e = 0; e < NU * NU; ++e) Rr[REGTAB ? e : 0] = Rs[e];
    }
#define QS(e) (REGTAB ? Qr[REGTAB ? (e) : 0] : Qs[e])
#define RS(e) (REGTAB ? Rr[REGTAB ? (e) : 0] : Rs[e])
    const T* lb = (const T*)a.lb;
    const T* ub = (const T*)a.ub;
    const T* bh = LDS ? blk + Lbh : (const T*)a.bh + (size_t)b * n;   // barrier diagonal (solver_barrier_kernel)
    T reg = ((const T*)a.reg)[b];
    T* Kst = LDS ? blk + LK : (T*)a.Kst + (size_t)b * H * nu * nx;
    T* kst = LDS ? blk + Lk : (T*)a.kst + (size_t)b * H * nu;
    T* Pst = LDS ? blk + LP : (T*)a.Pst + (size_t)b * H * nx * nx;
    T* pst = LDS ? blk + Lp : (T*)a.pst + (size_t)b * H * nx;
    T* dz = LDS ? blk + Ldz : dzg;
    const int uo = H * nx;

    // Damping levels side by side: a sweep that meets an indefinite pivot has to be repeated with more damping, and the
    // launch waited for the problem in hundreds that needs the third attempt (one sweep is a ~20 us latency chain of one
    // lane; the levels are known beforehand: decade steps).  `spec` lanes per problem now run the levels of a round at the
    // same time, each into its own region of the problem's block; the FIRST level that goes through is the one used, so
    // the result is what the sequential attempts gave.
    constexpr bool PREF = FIX && NX * (NX + NU) + (NX + NU) * (NX + NU) <= 16;
    struct StageIn {
        T At[PREF ? NX * (NX + NU) : 1], W[PREF ? (NX + NU) * (NX + NU) : 1], gc[PREF ? NX : 1], gru[PREF ? NU : 1],
            bhu[PREF ? NU : 1], grx[PREF ? NX : 1], bhx[PREF ? NX : 1];
    };
    auto fetch_stage = [&](const int t, StageIn& d) {
        #pragma unroll
        for (int e = 0; e < nx * nin; ++e) d.At[PREF ? e : 0] = tl[(size_t)t * nx * nin + e];
        #pragma unroll
        for (int e = 0; e < nin * nin; ++e) d.W[PREF ? e : 0] = Wb[(size_t)t * nin * nin + e];
        #pragma unroll
        for (int i = 0; i < nx; ++i) d.gc[PREF ? i : 0] = gc[t * nx + i];
        #pragma unroll
        for (int i = 0; i < nu; ++i) { d.gru[PREF ? i : 0] = gr[uo + t * nu + i]; d.bhu[PREF ? i : 0] = bh[uo + t * nu + i]; }
        if (t > 0) {
            #pragma unroll
            for (int i = 0; i < nx; ++i) { d.grx[PREF ? i : 0] = gr[(t - 1) * nx + i]; d.bhx[PREF ? i : 0] = bh[(t - 1) * nx + i]; }
        }
    };
#define AT(e) (PREF ? cur.At[PREF ? (e) : 0] : At[e])
#define WT(e) (PREF ? cur.W[PREF ? (e) : 0] : Wt[e])
#define GC(k) (PREF ? cur.gc[PREF ? (k) : 0] : gc[t * nx + (k)])
#define GRU(i) (PREF ? cur.gru[PREF ? (i) : 0] : gr[uo + t * nu + (i)])
#define BHU(i) (PREF ? cur.bhu[PREF ? (i) : 0] : bh[uo + t * nu + (i)])
#define GRX(i) (PREF ? cur.grx[PREF ? (i) : 0] : gr[(t - 1) * nx + (i)])
#define BHX(i) (PREF ? cur.bhx[PREF ? (i) : 0] : bh[(t - 1) * nx + (i)])
    StageIn sin_a, sin_b;
    int restarts = 0;
    bool solved = false, any = false;
    const T reg_in = reg;
    for (int base = 0; base < a.lq_attempts && !any; base += spec) {
    const int aj = base + aj0;
    reg = reg_in;
    for (int k = 0; k < aj; ++k) reg = fmax(reg * (T)a.reg_raise, T(NEMPC_REG_FLOOR));
    bool pd = aj < a.lq_attempts;
    // terminal value function: V_{H-1}(dx) = 1/2 dx' Hx dx + gx' dx
    #pragma unroll
    for (int i = 0; i < nx; ++i) {
        T ga = T(0), ha = T(0);
        ha = bh[(H - 1) * nx + i];
        #pragma unroll
        for (int j = 0; j < nx; ++j) TMP(oP + i * nx + j) = QTs[i * nx + j] + (i == j ? ha : T(0));   // terminal weight
        TMP(op + i) = gr[(H - 1) * nx + i];
    }
    if (PREF && pd) fetch_stage(H - 1, sin_a);
    // One stage of the backward recursion.  Small stages (2/1): the stage's inputs (tile, Lagrangian block, defects,
    // gradient and barrier entries) come from registers that were loaded a whole stage earlier (`cur`), and the next
    // stage's are requested at the top (`nxt`) -- the sweep no longer stops four or five times per stage for an LDS
    // round trip behind the stores of the recursion, which the compiler must assume to alias the loads.
    auto stage = [&](const int t, const StageIn& cur, StageIn& nxt) -> bool {
        if (PREF && t > 0) fetch_stage(t - 1, nxt);
        const T* At = tl + (size_t)t * nx * nin;   // [i][0:nx] = A, [i][nx:] = B
        const T* Wt = Wb + (size_t)t * nin * nin;  // Lagrangian block over (x_{t-1}, u_t)
        #pragma unroll
        for (int i = 0; i < nx; ++i) {
            #pragma unroll
            for (int j = 0; j < nx; ++j) Pst[(size_t)t * nx * nx + i * nx + j] = TMP(oP + i * nx + j);
            pst[(size_t)t * nx + i] = TMP(op + i);
        }
        // PA = P A, PB = P B, Pc = P c + p
        #pragma unroll
        for (int i = 0; i < nx; ++i) {
            if (t > 0 || FIX)     // (fixed shapes: unconditional -- a guard turns into a select per entry and stage)
                #pragma unroll
                for (int j = 0; j < nx; ++j) {
                    T v = T(0);
                    #pragma unroll
                    for (int k = 0; k < nx; ++k) v = fma(TMP(oP + i * nx + k), AT(k * nin + j), v);
                    TMP(oPA + i * nx + j) = v;
                }
            #pragma unroll
            for (int j = 0; j < nu; ++j) {
                T v = T(0);
                #pragma unroll
                for (int k = 0; k < nx; ++k) v = fma(TMP(oP + i * nx + k), AT(k * nin + nx + j), v);
                TMP(oPB + i * nu + j) = v;
            }
            T v = TMP(op + i);
            #pragma unroll
            for (int k = 0; k < nx; ++k) v = fma(TMP(oP + i * nx + k), GC(k), v);
            TMP(oPc + i) = v;
        }
        // Quu = Rs + barrier + reg + B' PB ; qu = gu + barrier + B' Pc ; Qux = B' PA
        #pragma unroll
        for (int i = 0; i < nu; ++i) {
            T ga = T(0), ha = T(0);
            ha = BHU(i);
            #pragma unroll
            for (int j = 0; j < nu; ++j) {
                T v = RS(i * nu + j) + WT((nx + i) * nin + nx + j) + (i == j ? ha + reg : T(0));
                #pragma unroll
                for (int k = 0; k < nx; ++k) v = fma(AT(k * nin + nx + i), TMP(oPB + k * nu + j), v);
                TMP(oQuu + i * nu + j) = v;
            }
            T v = GRU(i);
            #pragma unroll
            for (int k = 0; k < nx; ++k) v = fma(AT(k * nin + nx + i), TMP(oPc + k), v);
            TMP(oqu + i) = v;
            if (t > 0 || FIX)     // (fixed shapes: unconditional -- a guard turns into a select per entry and stage)
                #pragma unroll
                for (int j = 0; j < nx; ++j) {
                    T w = WT((nx + i) * nin + j);
                    #pragma unroll
                    for (int k = 0; k < nx; ++k) w = fma(AT(k * nin + nx + i), TMP(oPA + k * nx + j), w);
                    TMP(oQux + i * nx + j) = w;
                }
        }
        // Cholesky Quu = L L' (lower, in place)
        #pragma unroll
        for (int j = 0; j < nu; ++j) {
            T d = TMP(oQuu + j * nu + j);
            #pragma unroll
            for (int k = 0; k < j; ++k) d -= TMP(oQuu + j * nu + k) * TMP(oQuu + j * nu + k);
            if (!(d > T(1e-12))) { pd = false; break; }   // not positive definite: restart with more damping
            // the diagonal of L is only ever divided by: its INVERSE is kept in its place (one division per pivot instead
            // of one per use -- seven per stage at 2/1 -- in a sweep whose time is its instruction count)
            d = inv_sqrt_pos(d);
            TMP(oQuu + j * nu + j) = d;
            #pragma unroll
            for (int i = j + 1; i < nu; ++i) {
                T v = TMP(oQuu + i * nu + j);
                #pragma unroll
                for (int k = 0; k < j; ++k) v -= TMP(oQuu + i * nu + k) * TMP(oQuu + j * nu + k);
                TMP(oQuu + i * nu + j) = v * d;
            }
        }
        if (!pd) return false;
        // kv = -Quu^-1 qu ; K = -Quu^-1 Qux   (forward then backward substitution, column by column)
        const int ncolK = t > 0 || FIX ? nx : 0;
        for (int col = -1; col < ncolK; ++col) {
            #pragma unroll
            for (int i = 0; i < nu; ++i) {
                T v = (col < 0) ? TMP(oqu + i) : TMP(oQux + i * nx + col);
                #pragma unroll
                for (int k = 0; k < i; ++k) v -= TMP(oQuu + i * nu + k) * TMP(odu + k);
                TMP(odu + i) = v * TMP(oQuu + i * nu + i);
            }
            #pragma unroll
            for (int i = nu - 1; i >= 0; --i) {
                T v = TMP(odu + i);
                #pragma unroll
                for (int k = i + 1; k < nu; ++k) v -= TMP(oQuu + k * nu + i) * TMP(odu + k);
                v *= TMP(oQuu + i * nu + i);
                TMP(odu + i) = v;
            }
            #pragma unroll
            for (int i = 0; i < nu; ++i) {
                if (col < 0) { TMP(okv + i) = -TMP(odu + i); kst[(size_t)t * nu + i] = -TMP(odu + i); }
                else { TMP(oK + i * nx + col) = -TMP(odu + i); Kst[(size_t)t * nu * nx + i * nx + col] = -TMP(odu + i); }
            }
        }
        if (t > 0) {
            // P_{t-1} = Hx_{t-1} + A' PA + Qux' K ; p_{t-1} = gx_{t-1} + A' Pc + Qux' kv
            #pragma unroll
            for (int i = 0; i < nx; ++i) {
                T ga = T(0), ha = T(0);
                ha = BHX(i);
                #pragma unroll
                for (int j = 0; j < nx; ++j) {
                    T v = QS(i * nx + j) + WT(i * nin + j) + (i == j ? ha : T(0));
                    #pragma unroll
                    for (int k = 0; k < nx; ++k) v = fma(AT(k * nin + i), TMP(oPA + k * nx + j), v);
                    #pragma unroll
                    for (int k = 0; k < nu; ++k) v = fma(TMP(oQux + k * nx + i), TMP(oK + k * nx + j), v);
                    TMP(oPn + i * nx + j) = v;
                }
                T v = GRX(i);
                #pragma unroll
                for (int k = 0; k < nx; ++k) v = fma(AT(k * nin + i), TMP(oPc + k), v);
                #pragma unroll
                for (int k = 0; k < nu; ++k) v = fma(TMP(oQux + k * nx + i), TMP(okv + k), v);
                TMP(opn + i) = v;
            }
            #pragma unroll
            for (int i = 0; i < nx; ++i) {
                #pragma unroll
                for (int j = 0; j < nx; ++j)
                    TMP(oP + i * nx + j) = i == j ? TMP(oPn + i * nx + j) : T(0.5) * (TMP(oPn + i * nx + j) + TMP(oPn + j * nx + i));
                TMP(op + i) = TMP(opn + i);
            }
        }
            return true;
    };
    for (int t = H - 1; t >= 0 && pd; t -= 2) {
        pd = stage(t, sin_a, sin_b);
        if (pd && t > 0) pd = stage(t - 1, sin_b, sin_a);
    }
    {
        const unsigned long long ok = __ballot(pd);
        int win = -1;
        for (int jj = spec - 1; jj >= 0; --jj)
            if ((ok >> (jj * ppw + pl)) & 1ull) win = jj;
        if (win >= 0) { any = true; solved = win == aj0; restarts = base + win; }
    }
    // decade steps from a floor of 1e-3: whenever a sweep of this problem family fails, the damping that lets it through
    // is 0.1 .. 100 (NEMPC_SOLVER_STATS), and the whole launch waits for the one problem in hundreds that climbs there --
    // nine attempts from the relaxed value with a floor of 1e-6, six from 1e-3 (the kernel runs 37 us clean, 10 us more
    // per attempt).  Remembering per problem the level that worked last time did not help: the restarting problems are
    // mostly first-timers.
    }
    if (!any) {
        restarts = a.lq_attempts;
        reg = reg_in;
        for (int k = 0; k < a.lq_attempts; ++k) reg = fmax(reg * (T)a.reg_raise, T(NEMPC_REG_FLOOR));
    }
    if (solved || (!any && aj0 == 0)) ((T*)a.reg)[b] = reg;
    T* lamn = LDS ? blk + Llam : (T*)a.lamn + (size_t)b * a.m;
    if (!solved && (any || aj0 != 0)) {
        // another lane of this problem holds the level that is used (or reports that none went through)
    } else if (!solved) {
        // Out of attempts for this iteration.  The launch waits for its slowest problem, and the one problem in hundreds
        // that needs five or six decades of damping made every other one wait ~50 us for it: it now keeps the damping it
        // has climbed to, takes NO step this iteration (zero step, multipliers unchanged, marked by a negative restart
        // count) and goes on climbing in the next one.
        for (int i = 0; i < n; ++i) dz[i] = T(0);
        const T* lcur = (const T*)a.lam + (size_t)b * a.m;
        for (int i = 0; i < H * nx; ++i) lamn[i] = lcur[i];
        info[INFO_STEP] = std::numeric_limits<T>::max();
        info[INFO_RESTARTS] = (T)(-restarts);
        if (LDS) blk[Lbh] = (T)(-restarts);          // (the barrier diagonal is no longer needed: slot for the post-pass)
    } else {
    LQ_STAMP(9);
    // forward sweep
    // The norms, the directional derivative and the fraction-to-the-boundary length of the step are NOT part of the
    // recursion: in LDS mode a wave per problem computes them from the staged arrays after the sweeps (below), with its
    // 64 lanes over the elements -- in the sweep they were 60 % of the forward pass's instructions (three divisions per
    // stage among them), issued one per ~10 cycles by a lone lane (kernel 37 -> 22 us clean)
    const bool norms_here = !LDS;
    T lam_inf = T(0), step_inf = T(0), amax = T(1), D0 = T(0), g1 = T(0), ginf = T(0), zinf = T(0);
    const T tau = T(0.995);
    #pragma unroll
    for (int i = 0; i < nx; ++i) TMP(odx + i) = T(0);
    if (PREF && LDS) {
        // small stages: as in the backward sweep, a stage's inputs (gains, tile, defects, value function) are in registers,
        // requested one stage ahead -- the recursion carries dx only, and no longer stops twice per stage for LDS round
        // trips behind its own stores
        struct FwdIn {
            T k[PREF ? NU : 1], K[PREF ? NU * NX : 1], A[PREF ? NX * (NX + NU) : 1], gc[PREF ? NX : 1], P[PREF ? NX * NX : 1],
                p[PREF ? NX : 1];
        };
        auto fetch_f = [&](const int t, FwdIn& d) {
            #pragma unroll
            for (int e = 0; e < nu; ++e) d.k[PREF ? e : 0] = kst[(size_t)t * nu + e];
            #pragma unroll
            for (int e = 0; e < nu * nx; ++e) d.K[PREF ? e : 0] = Kst[(size_t)t * nu * nx + e];
            #pragma unroll
            for (int e = 0; e < nx * nin; ++e) d.A[PREF ? e : 0] = tl[(size_t)t * nx * nin + e];
            #pragma unroll
            for (int e = 0; e < nx; ++e) { d.gc[PREF ? e : 0] = gc[t * nx + e]; d.p[PREF ? e : 0] = pst[(size_t)t * nx + e]; }
            #pragma unroll
            for (int e = 0; e < nx * nx; ++e) d.P[PREF ? e : 0] = Pst[(size_t)t * nx * nx + e];
        };
        auto fstage = [&](const int t, const FwdIn& c, FwdIn& nxt) {
            if (t + 1 < H) fetch_f(t + 1, nxt);
            #pragma unroll
            for (int i = 0; i < nu; ++i) {
                T v = c.k[PREF ? i : 0];
                if (t > 0)
                    #pragma unroll
                    for (int k = 0; k < nx; ++k) v = fma(c.K[PREF ? i * nx + k : 0], TMP(odx + k), v);
                TMP(odu + i) = v;
            }
            #pragma unroll
            for (int i = 0; i < nx; ++i) {
                T v = c.gc[PREF ? i : 0];
                if (t > 0)
                    #pragma unroll
                    for (int k = 0; k < nx; ++k) v = fma(c.A[PREF ? i * nin + k : 0], TMP(odx + k), v);
                #pragma unroll
                for (int k = 0; k < nu; ++k) v = fma(c.A[PREF ? i * nin + nx + k : 0], TMP(odu + k), v);
                TMP(odxn + i) = v;
            }
            #pragma unroll
            for (int i = 0; i < nx; ++i) {
                T lam = c.p[PREF ? i : 0];
                #pragma unroll
                for (int k = 0; k < nx; ++k) lam = fma(c.P[PREF ? i * nx + k : 0], TMP(odxn + k), lam);
                lamn[t * nx + i] = lam;
                dz[t * nx + i] = TMP(odxn + i);
            }
            #pragma unroll
            for (int i = 0; i < nx; ++i) TMP(odx + i) = TMP(odxn + i);
            #pragma unroll
            for (int i = 0; i < nu; ++i) dz[uo + t * nu + i] = TMP(odu + i);
        };
        FwdIn fa, fb;
        fetch_f(0, fa);
        for (int t = 0; t < H; t += 2) {
            fstage(t, fa, fb);
            if (t + 1 < H) fstage(t + 1, fb, fa);
        }
    } else
    for (int t = 0; t < H; ++t) {
        const T* At = tl + (size_t)t * nx * nin;
        #pragma unroll
        for (int i = 0; i < nu; ++i) {
            T v = kst[(size_t)t * nu + i];
            if (t > 0)
                #pragma unroll
                for (int k = 0; k < nx; ++k) v = fma(Kst[(size_t)t * nu * nx + i * nx + k], TMP(odx + k), v);
            TMP(odu + i) = v;
        }
        #pragma unroll
        for (int i = 0; i < nx; ++i) {
            T v = gc[t * nx + i];
            if (t > 0)
                #pragma unroll
                for (int k = 0; k < nx; ++k) v = fma(At[i * nin + k], TMP(odx + k), v);
            #pragma unroll
            for (int k = 0; k < nu; ++k) v = fma(At[i * nin + nx + k], TMP(odu + k), v);
            TMP(odxn + i) = v;
        }
        #pragma unroll
        for (int i = 0; i < nx; ++i) {
            T lam = pst[(size_t)t * nx + i];
            #pragma unroll
            for (int k = 0; k < nx; ++k) lam = fma(Pst[(size_t)t * nx * nx + i * nx + k], TMP(odxn + k), lam);
            lamn[t * nx + i] = lam;
            const T d = TMP(odxn + i);
            dz[t * nx + i] = d;
            TMP(odx + i) = d;
            if (norms_here) {
                const T zz = z[t * nx + i], lo = lb[t * nx + i], hi = ub[t * nx + i];
                lam_inf = fmax(lam_inf, fabs(lam));
                step_inf = fmax(step_inf, fabs(d));
                zinf = fmax(zinf, fabs(zz));
                D0 = fma(gr[t * nx + i], d, D0);
                if (d < T(0) && lo > -std::numeric_limits<T>::max()) amax = fmin(amax, tau * (zz - lo) / (-d));
                if (d > T(0) && hi < std::numeric_limits<T>::max()) amax = fmin(amax, tau * (hi - zz) / d);
                const T gv = fabs(gc[t * nx + i]);
                g1 += gv;
                ginf = fmax(ginf, gv);
            }
        }
        #pragma unroll
        for (int i = 0; i < nu; ++i) {
            const T d = TMP(odu + i);
            dz[uo + t * nu + i] = d;
            if (norms_here) {
                const T zz = z[uo + t * nu + i], lo = lb[uo + t * nu + i], hi = ub[uo + t * nu + i];
                step_inf = fmax(step_inf, fabs(d));
                zinf = fmax(zinf, fabs(zz));
                D0 = fma(gr[uo + t * nu + i], d, D0);
                if (d < T(0) && lo > -std::numeric_limits<T>::max()) amax = fmin(amax, tau * (zz - lo) / (-d));
                if (d > T(0) && hi < std::numeric_limits<T>::max()) amax = fmin(amax, tau * (hi - zz) / d);
            }
        }
    }
    if (norms_here) {
        info[INFO_LAM] = lam_inf; info[INFO_STEP] = step_inf; info[INFO_AMAX] = amax; info[INFO_G1] = g1;
        info[INFO_GINF] = ginf; info[INFO_D0] = D0; info[INFO_ZINF] = zinf;
    }
    info[INFO_RESTARTS] = (T)restarts;
    if (LDS) blk[Lbh] = (T)restarts;
    }   // solved
#undef TMP
#undef QS
#undef RS
#undef AT
#undef WT
#undef GC
#undef GRU
#undef BHU
#undef GRX
#undef BHX
    }
    }   // !SCAN
    LQ_STAMP(2);
    if (LDS) {
        __syncthreads();
        LQ_STAMP(3);
        {
            // norms of the steps just computed: wave w takes problems w, w + #waves, ...
            const int wv = lane >> 6, ln = lane & 63, nwv = nthr >> 6;
            const T* lb = lds_lb;
            const T* ub = lds_ub;
            const T tau = T(0.995);
            for (int pp = wv; pp < np; pp += nwv) {
                const int bp = b0 + pp;
                const T* blk = lds + (size_t)pp * a.lds_stride;
                // the problem's scalars for the step part below: all requested now, used after the norms
                const T* infg = (const T*)a.info + (size_t)bp * INFO_N;
                StepInfo<T> si;
                if (pp == wvu && pf_ok) {       // (nothing in this kernel has written them since the prefetch)
                    si.status = pf_status;
                    si.mu = pf_mu; si.nu = pf_pen; si.f = pf_f; si.lsk = pf_lsk; si.lsa = pf_lsa; si.lsr = pf_lsr;
                } else {
                    si.status = a.status[bp];
                    if (a.fuse_step) {
                        si.mu = ((const T*)a.mu)[bp]; si.nu = ((const T*)a.pen)[bp]; si.f = ((const T*)a.f_it)[bp];
                        si.lsk = infg[INFO_LSK]; si.lsa = infg[INFO_LSA]; si.lsr = infg[INFO_LSR];
                    }
                }
                if (si.status >= 0) {
                    if (a.fuse_step) {      // finished problem: no step, its trial point is the iterate
                        if (ln == 0) a.lsdone[bp] = 1;
                        T* zt = (T*)a.Zt_it + (size_t)bp * n;
                        for (int i = ln; i < n; i += 64) zt[i] = blk[Lz + i];
                    }
                    continue;
                }
                T lam_inf = T(0), step_inf = T(0), amax = T(1), D0 = T(0), g1 = T(0), ginf = T(0), zinf = T(0);
                for (int i = ln; i < n; i += 64) {
                    const T d = blk[Ldz + i], zz = blk[Lz + i], lo = lb[i], hi = ub[i];
                    step_inf = fmax(step_inf, fabs(d));
                    zinf = fmax(zinf, fabs(zz));
                    D0 = fma(blk[Lgr + i], d, D0);
                    if (d < T(0) && lo > -std::numeric_limits<T>::max()) amax = fmin(amax, tau * (zz - lo) / (-d));
                    if (d > T(0) && hi < std::numeric_limits<T>::max()) amax = fmin(amax, tau * (hi - zz) / d);
                }
                for (int i = ln; i < H * nx; i += 64) {
                    lam_inf = fmax(lam_inf, fabs(blk[Llam + i]));
                    const T gv = fabs(blk[Lgc + i]);
                    g1 += gv;
                    ginf = fmax(ginf, gv);
                }
                // (on the vector unit alone: each shuffle tree was six dependent LDS round trips of a lone wave)
                lam_inf = (T)wave_max_lane0((double)lam_inf);
                step_inf = (T)wave_max_lane0((double)step_inf);
                amax = (T)wave_min_lane0((double)amax);
                D0 = (T)wave_sum_lane0((double)D0);
                g1 = (T)wave_sum_lane0((double)g1);
                ginf = (T)wave_max_lane0((double)ginf);
                zinf = (T)wave_max_lane0((double)zinf);
                LQ_STAMP(4);
                const T rst = blk[Lbh];                    // restart count left by the sweeping lane (negative: sat out)
                const bool sat_out = rst < T(0);           // no step this iteration: keep the sentinel
                if (ln == 0) {
                    T* info = (T*)a.info + (size_t)bp * INFO_N;
                    info[INFO_LAM] = lam_inf; info[INFO_STEP] = sat_out ? std::numeric_limits<T>::max() : step_inf;
                    info[INFO_AMAX] = amax; info[INFO_G1] = g1;
                    info[INFO_GINF] = ginf; info[INFO_D0] = D0; info[INFO_ZINF] = zinf;
                }
                if (a.fuse_step) {
                    // what solver_step_kernel does, from the LDS copies and the norms still in registers: dual steps of the
                    // bounds, convergence test / barrier update / merit at the iterate, first trial point
                    si.ginf = (T)wave_bcast_lane0((double)ginf); si.zinf = (T)wave_bcast_lane0((double)zinf);
                    si.lam = (T)wave_bcast_lane0((double)lam_inf);
                    si.d0 = (T)wave_bcast_lane0((double)D0); si.amax = (T)wave_bcast_lane0((double)amax);
                    si.step = sat_out ? std::numeric_limits<T>::max() : (T)wave_bcast_lane0((double)step_inf);
                    si.restarts = rst;
                    solver_dual_body<T>(a, bp, ln, blk + Lz, blk + Ldz, si.mu, si.status, lb, ub, blk + Lzl, blk + Lzu);
                    LQ_STAMP(5);
                    int lsd;
                    T al;
                    solver_merit0_body<T>(a, bp, ln, (const T*)a.f_it, blk + Lz, blk + Lgc, si, lsd, al, lb, ub);
                    LQ_STAMP(6);
                    T* zt = (T*)a.Zt_it + (size_t)bp * n;
                    for (int i = ln; i < n; i += 64) zt[i] = lsd ? blk[Lz + i] : fma(al, blk[Ldz + i], blk[Lz + i]);
                    if (a.lam_t) {       // multipliers of the trial point (what the acceptance kernel sets on acceptance)
                        T* lt = (T*)a.lam_t + (size_t)bp * a.m;
                        for (int i = ln; i < H * nx; i += 64) {
                            const T l0 = blk[Llc + i];          // (the current multipliers were staged with the working set)
                            lt[i] = lsd ? l0 : fma(al, blk[Llam + i] - l0, l0);
                        }
                    }
                }
            }
        }
        LQ_STAMP(7);
        auto stage_out = [&](T* __restrict__ dst, int per, int dst_stride, int loff) {
            for (int pp = st_p0; pp < np; pp += st_pstep) {
                T* dp = dst + (size_t)(b0 + pp) * dst_stride;
                const T* sp = lds + (size_t)pp * a.lds_stride + loff;
                for (int e = st_e0; e < per; e += st_estep) dp[e] = sp[e];
            }
        };
        stage_out((T*)a.dz, n, n, Ldz);
        stage_out((T*)a.lamn, H * nx, a.m, Llam);
    }
    LQ_STAMP(8);
}

// Wave-per-problem variant of the LQ solve (LDS mode only): the same recursion, formulas and summation order as
// solver_lq_kernel, but every small matrix operation of a stage is spread over the 64 lanes of the problem's wave
// (entry per lane) with wave-level synchronisation between dependent operations.  A thread-per-problem sweep is one
// serial chain of ~1000 FMAs per stage on 6/3-sized blocks with 4 of 64 lanes busy; here a stage is ~8 short phases.
// NX, NU > 0: stage dimensions fixed at compile time (the entry loops unroll and their LDS loads overlap).
template <typename T, int NX, int NU>
__global__ __launch_bounds__(256) void solver_lqw_kernel(SolverArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    T* lds = reinterpret_cast<T*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int H = a.H, nx = NX > 0 ? NX : a.nx, nu = NX > 0 ? NU : a.nu, nin = nx + nu, n = a.n;
    const int ppw = a.ppw;                       // == waves per workgroup
    const int b0 = blockIdx.x * ppw;
    const int np = a.B - b0 < ppw ? a.B - b0 : ppw;
    const int Lz = 0, Lgr = Lz + n, Lgc = Lgr + n, Ltl = Lgc + H * nx, LW = Ltl + H * nx * nin, Llam = LW + H * nin * nin,
              Ldz = Llam + H * nx, Lbh = Ldz + n, LK = Lbh + n, Lk = LK + H * nu * nx, LP = Lk + H * nu, Lp = LP + H * nx * nx,
              Ltmp = Lp + H * nx;
    auto stage_in = [&](const T* __restrict__ src, int per, int src_stride, int loff) {
        const int tot = np * per;
        const T* base = src + (size_t)b0 * src_stride;
        for (int i = tid; i < tot; i += 256) {
            const int pp = i / per, e = i - pp * per;
            lds[(size_t)pp * a.lds_stride + loff + e] = base[(size_t)pp * src_stride + e];
        }
    };
    if (blockIdx.x == 0 && tid == 0) *a.n_active = 0;       // counter of the convergence test that follows this kernel
    stage_in((const T*)a.Z, n, n, Lz);
    for (int i = tid; i < np * n; i += 256) {               // gradient + barrier gradient, barrier diagonal
        const int pp = i / n, e = i - pp * n;
        T ga, ha;
        solver_barrier_terms<T>(a, b0 + pp, e, ga, ha);
        T* blkp = lds + (size_t)pp * a.lds_stride;
        T gv = ((const T*)a.grad)[(size_t)(b0 + pp) * n + e];
        gv += ga;
        blkp[Lgr + e] = gv;
        blkp[Lbh + e] = ha;
    }
    stage_in((const T*)a.g, H * nx, a.m, Lgc);
    stage_in((const T*)a.tiles, H * nx * nin, H * nx * nin, Ltl);
    stage_in((const T*)a.hblk, H * nin * nin, H * nin * nin, LW);
    __syncthreads();

    const int b = b0 + wv;
    if (wv < np) {
        T* blk = lds + (size_t)wv * a.lds_stride;
        auto wsync = [] {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        if (a.status[b] >= 0) {
            // finished problem: zero step, multipliers unchanged
            for (int i = lane; i < n; i += 64) blk[Ldz + i] = T(0);
            const T* lcur = (const T*)a.lam + (size_t)b * a.m;
            for (int i = lane; i < H * nx; i += 64) blk[Llam + i] = lcur[i];
        } else {
            T* tb = blk + Ltmp;
            int off = 0;
            const int oP = off; off += nx * nx;
            const int oPA = off; off += nx * nx;
            const int oPn = off; off += nx * nx;
            const int oPB = off; off += nx * nu;
            const int oQux = off; off += nu * nx;
            const int oK = off; off += nu * nx;
            const int oQuu = off; off += nu * nu;
            const int oPc = off; off += nx;
            const int op = off; off += nx;
            const int opn = off; off += nx;
            const int odx = off; off += nx;
            const int odxn = off; off += nx;
            const int oqu = off; off += nu;
            const int okv = off; off += nu;
            const int odu = off; off += nu;
            const int oflag = off;                 // 1 slot: Cholesky verdict of the stage
            const T* z = blk + Lz;
            const T* gr = blk + Lgr;
            const T* gc = blk + Lgc;
            const T* tl = blk + Ltl;
            const T* Wb = blk + LW;
            const T* Qs = (const T*)a.obj + a.oo.Qs;
            const T* QTs = (const T*)a.obj + a.oo.QTs;
            const T* Rs = (const T*)a.obj + a.oo.Rs;
            const T* lb = (const T*)a.lb;
            const T* ub = (const T*)a.ub;
            const T* bh = blk + Lbh;                                             // barrier diagonal (solver_barrier_kernel)
            T reg = ((const T*)a.reg)[b];
            T* Kst = blk + LK;
            T* kst = blk + Lk;
            T* Pst = blk + LP;
            T* pst = blk + Lp;
            T* dz = blk + Ldz;
            T* lamn = blk + Llam;
            const int uo = H * nx;
            int restarts = 0;
            bool solved = false;
            for (int attempt = 0; attempt < a.lq_attempts; ++attempt) {
                bool pd = true;
                // terminal value function
                for (int e = lane; e < nx * nx + nx; e += 64) {
                    if (e < nx * nx) {
                        const int i = e / nx, j = e - i * nx;
                        T ga = T(0), ha = T(0);
                        if (i == j) ha = bh[(H - 1) * nx + i];
                        tb[oP + e] = QTs[e] + (i == j ? ha : T(0));   // terminal weight
                    } else {
                        const int i = e - nx * nx;
                        T ga = T(0), ha = T(0);
                        ha = bh[(H - 1) * nx + i];
                        tb[op + i] = gr[(H - 1) * nx + i] + ga;
                    }
                }
                wsync();
                for (int t = H - 1; t >= 0 && pd; --t) {
                    const T* At = tl + (size_t)t * nx * nin;
                    const T* Wt = Wb + (size_t)t * nin * nin;
                    // store P_t, p_t; PA = P A, PB = P B, Pc = P c + p
                    for (int e = lane; e < 2 * nx * nx + nx * nu + 2 * nx; e += 64) {
                        int r = e;
                        if (r < nx * nx) { Pst[(size_t)t * nx * nx + r] = tb[oP + r]; continue; }
                        r -= nx * nx;
                        if (r < nx) { pst[(size_t)t * nx + r] = tb[op + r]; continue; }
                        r -= nx;
                        if (r < nx * nx) {
                            if (t > 0) {
                                const int i = r / nx, j = r - i * nx;
                                T v = T(0);
                                for (int k = 0; k < nx; ++k) v = fma(tb[oP + i * nx + k], At[k * nin + j], v);
                                tb[oPA + r] = v;
                            }
                            continue;
                        }
                        r -= nx * nx;
                        if (r < nx * nu) {
                            const int i = r / nu, j = r - i * nu;
                            T v = T(0);
                            for (int k = 0; k < nx; ++k) v = fma(tb[oP + i * nx + k], At[k * nin + nx + j], v);
                            tb[oPB + r] = v;
                            continue;
                        }
                        r -= nx * nu;
                        T v = tb[op + r];
                        for (int k = 0; k < nx; ++k) v = fma(tb[oP + r * nx + k], gc[t * nx + k], v);
                        tb[oPc + r] = v;
                    }
                    wsync();
                    // Quu, qu, Qux
                    for (int e = lane; e < nu * nu + nu + nu * nx; e += 64) {
                        int r = e;
                        if (r < nu * nu) {
                            const int i = r / nu, j = r - i * nu;
                            T ga = T(0), ha = T(0);
                            if (i == j) ha = bh[uo + t * nu + i];
                            T v = Rs[i * nu + j] + Wt[(nx + i) * nin + nx + j] + (i == j ? ha + reg : T(0));
                            for (int k = 0; k < nx; ++k) v = fma(At[k * nin + nx + i], tb[oPB + k * nu + j], v);
                            tb[oQuu + r] = v;
                            continue;
                        }
                        r -= nu * nu;
                        if (r < nu) {
                            T ga = T(0), ha = T(0);
                            ha = bh[uo + t * nu + r];
                            T v = gr[uo + t * nu + r] + ga;
                            for (int k = 0; k < nx; ++k) v = fma(At[k * nin + nx + r], tb[oPc + k], v);
                            tb[oqu + r] = v;
                            continue;
                        }
                        r -= nu;
                        if (t > 0) {
                            const int i = r / nx, j = r - i * nx;
                            T w = Wt[(nx + i) * nin + j];
                            for (int k = 0; k < nx; ++k) w = fma(At[k * nin + nx + i], tb[oPA + k * nx + j], w);
                            tb[oQux + r] = w;
                        }
                    }
                    wsync();
                    // Cholesky Quu = L L' (lower, in place): small and serial, one lane
                    if (lane == 0) {
                        T ok = T(1);
                        for (int j = 0; j < nu && ok > T(0); ++j) {
                            T d = tb[oQuu + j * nu + j];
                            for (int k = 0; k < j; ++k) d -= tb[oQuu + j * nu + k] * tb[oQuu + j * nu + k];
                            if (!(d > T(1e-12))) { ok = T(0); break; }
                            d = inv_sqrt_pos(d);              // inverse diagonal, as in the thread-per-problem kernel
                            tb[oQuu + j * nu + j] = d;
                            for (int i = j + 1; i < nu; ++i) {
                                T v = tb[oQuu + i * nu + j];
                                for (int k = 0; k < j; ++k) v -= tb[oQuu + i * nu + k] * tb[oQuu + j * nu + k];
                                tb[oQuu + i * nu + j] = v * d;
                            }
                        }
                        tb[oflag] = ok;
                    }
                    wsync();
                    if (!(tb[oflag] > T(0))) { pd = false; break; }
                    // kv = -Quu^-1 qu ; K = -Quu^-1 Qux: one right-hand side per lane (lane 0: qu, lane 1+col: Qux[:,col]);
                    // the substitution runs in a per-lane strip of the du scratch (nu entries per column)
                    const int ncolK = t > 0 ? nx : 0;
                    for (int c1 = lane; c1 <= ncolK; c1 += 64) {
                        const int col = c1 - 1;
                        T* y = tb + oflag + 1 + c1 * nu;       // per-column work vector
                        for (int i = 0; i < nu; ++i) {
                            T v = (col < 0) ? tb[oqu + i] : tb[oQux + i * nx + col];
                            for (int k = 0; k < i; ++k) v -= tb[oQuu + i * nu + k] * y[k];
                            y[i] = v * tb[oQuu + i * nu + i];
                        }
                        for (int i = nu - 1; i >= 0; --i) {
                            T v = y[i];
                            for (int k = i + 1; k < nu; ++k) v -= tb[oQuu + k * nu + i] * y[k];
                            v *= tb[oQuu + i * nu + i];
                            y[i] = v;
                        }
                        for (int i = 0; i < nu; ++i) {
                            if (col < 0) { tb[okv + i] = -y[i]; kst[(size_t)t * nu + i] = -y[i]; }
                            else { tb[oK + i * nx + col] = -y[i]; Kst[(size_t)t * nu * nx + i * nx + col] = -y[i]; }
                        }
                    }
                    wsync();
                    if (t > 0) {
                        for (int e = lane; e < nx * nx + nx; e += 64) {
                            if (e < nx * nx) {
                                const int i = e / nx, j = e - i * nx;
                                T ga = T(0), ha = T(0);
                                if (i == j) ha = bh[(t - 1) * nx + i];
                                T v = Qs[i * nx + j] + Wt[i * nin + j] + (i == j ? ha : T(0));
                                for (int k = 0; k < nx; ++k) v = fma(At[k * nin + i], tb[oPA + k * nx + j], v);
                                for (int k = 0; k < nu; ++k) v = fma(tb[oQux + k * nx + i], tb[oK + k * nx + j], v);
                                tb[oPn + e] = v;
                            } else {
                                const int i = e - nx * nx;
                                T ga = T(0), ha = T(0);
                                ha = bh[(t - 1) * nx + i];
                                T v = gr[(t - 1) * nx + i] + ga;
                                for (int k = 0; k < nx; ++k) v = fma(At[k * nin + i], tb[oPc + k], v);
                                for (int k = 0; k < nu; ++k) v = fma(tb[oQux + k * nx + i], tb[okv + k], v);
                                tb[opn + i] = v;
                            }
                        }
                        wsync();
                        for (int e = lane; e < nx * nx + nx; e += 64) {
                            if (e < nx * nx) {
                                const int i = e / nx, j = e - i * nx;
                                tb[oP + e] = T(0.5) * (tb[oPn + i * nx + j] + tb[oPn + j * nx + i]);
                            } else {
                                tb[op + e - nx * nx] = tb[opn + e - nx * nx];
                            }
                        }
                        wsync();
                    }
                }
                if (pd) { solved = true; break; }
                reg = fmax(reg * (T)a.reg_raise, T(NEMPC_REG_FLOOR));     // as in the thread-per-problem kernel
                ++restarts;
            }
            if (lane == 0) ((T*)a.reg)[b] = reg;
            if (!solved) {
                // out of attempts: no step this iteration (see the thread-per-problem kernel)
                for (int i = lane; i < n; i += 64) dz[i] = T(0);
                const T* lcur = (const T*)a.lam + (size_t)b * a.m;
                for (int i = lane; i < H * nx; i += 64) lamn[i] = lcur[i];
                if (lane == 0) {
                    T* info = (T*)a.info + (size_t)b * INFO_N;
                    info[INFO_STEP] = std::numeric_limits<T>::max();
                    info[INFO_RESTARTS] = (T)(-restarts);
                }
            } else {
            // forward sweep; the norms are accumulated per lane and reduced at the end
            T lam_inf = T(0), step_inf = T(0), amax = T(1), D0 = T(0), g1 = T(0), ginf = T(0), zinf = T(0);
            const T tau = T(0.995);
            for (int i = lane; i < nx; i += 64) tb[odx + i] = T(0);
            wsync();
            for (int t = 0; t < H; ++t) {
                const T* At = tl + (size_t)t * nx * nin;
                for (int i = lane; i < nu; i += 64) {
                    T v = kst[(size_t)t * nu + i];
                    if (t > 0)
                        for (int k = 0; k < nx; ++k) v = fma(Kst[(size_t)t * nu * nx + i * nx + k], tb[odx + k], v);
                    tb[odu + i] = v;
                }
                wsync();
                for (int i = lane; i < nx; i += 64) {
                    T v = gc[t * nx + i];
                    if (t > 0)
                        for (int k = 0; k < nx; ++k) v = fma(At[i * nin + k], tb[odx + k], v);
                    for (int k = 0; k < nu; ++k) v = fma(At[i * nin + nx + k], tb[odu + k], v);
                    tb[odxn + i] = v;
                }
                wsync();
                for (int i = lane; i < nx + nu; i += 64) {
                    if (i < nx) {
                        T lam = pst[(size_t)t * nx + i];
                        for (int k = 0; k < nx; ++k) lam = fma(Pst[(size_t)t * nx * nx + i * nx + k], tb[odxn + k], lam);
                        lam_inf = fmax(lam_inf, fabs(lam));
                        lamn[t * nx + i] = lam;
                        const T d = tb[odxn + i], zz = z[t * nx + i], lo = lb[t * nx + i], hi = ub[t * nx + i];
                        dz[t * nx + i] = d;
                        step_inf = fmax(step_inf, fabs(d));
                        zinf = fmax(zinf, fabs(zz));
                        T ga = T(0), ha = T(0);
                                    D0 = fma(gr[t * nx + i] + ga, d, D0);
                        if (d < T(0) && lo > -std::numeric_limits<T>::max()) amax = fmin(amax, tau * (zz - lo) / (-d));
                        if (d > T(0) && hi < std::numeric_limits<T>::max()) amax = fmin(amax, tau * (hi - zz) / d);
                        const T gv = fabs(gc[t * nx + i]);
                        g1 += gv;
                        ginf = fmax(ginf, gv);
                    } else {
                        const int j = i - nx;
                        const T d = tb[odu + j], zz = z[uo + t * nu + j], lo = lb[uo + t * nu + j], hi = ub[uo + t * nu + j];
                        dz[uo + t * nu + j] = d;
                        step_inf = fmax(step_inf, fabs(d));
                        zinf = fmax(zinf, fabs(zz));
                        T ga = T(0), ha = T(0);
                                    D0 = fma(gr[uo + t * nu + j] + ga, d, D0);
                        if (d < T(0) && lo > -std::numeric_limits<T>::max()) amax = fmin(amax, tau * (zz - lo) / (-d));
                        if (d > T(0) && hi < std::numeric_limits<T>::max()) amax = fmin(amax, tau * (hi - zz) / d);
                    }
                }
                wsync();
                for (int i = lane; i < nx; i += 64) tb[odx + i] = tb[odxn + i];
                wsync();
            }
            for (int o = 32; o > 0; o >>= 1) {
                lam_inf = fmax(lam_inf, __shfl_down(lam_inf, o, 64));
                step_inf = fmax(step_inf, __shfl_down(step_inf, o, 64));
                amax = fmin(amax, __shfl_down(amax, o, 64));
                D0 += __shfl_down(D0, o, 64);
                g1 += __shfl_down(g1, o, 64);
                ginf = fmax(ginf, __shfl_down(ginf, o, 64));
                zinf = fmax(zinf, __shfl_down(zinf, o, 64));
            }
            if (lane == 0) {
                T* info = (T*)a.info + (size_t)b * INFO_N;
                info[INFO_LAM] = lam_inf; info[INFO_STEP] = step_inf; info[INFO_AMAX] = amax; info[INFO_G1] = g1;
                info[INFO_GINF] = ginf; info[INFO_D0] = D0; info[INFO_ZINF] = zinf; info[INFO_RESTARTS] = (T)restarts;
            }
            }   // solved
        }
    }
    __syncthreads();
    auto stage_out = [&](T* __restrict__ dst, int per, int dst_stride, int loff) {
        const int tot = np * per;
        T* base = dst + (size_t)b0 * dst_stride;
        for (int i = tid; i < tot; i += 256) {
            const int pp = i / per, e = i - pp * per;
            base[(size_t)pp * dst_stride + e] = lds[(size_t)pp * a.lds_stride + loff + e];
        }
    };
    stage_out((T*)a.dz, n, n, Ldz);
    stage_out((T*)a.lamn, H * nx, a.m, Llam);
}

// log-barrier value of one problem's variables, summed by a wave
template <typename T>
__device__ __forceinline__ double barrier_value(const T* z, const T* lb, const T* ub, int n, T mu, int lane) {
    double acc = 0.0;
    if (mu > T(0))
        for (int i = lane; i < n; i += 64) {
            if (lb[i] > -std::numeric_limits<T>::max()) acc -= (double)mu * log((double)(z[i] - lb[i]));
            if (ub[i] < std::numeric_limits<T>::max()) acc -= (double)mu * log((double)(ub[i] - z[i]));
        }
    return wave_bcast_lane0(wave_sum_lane0(acc));
}

template <typename T>
__device__ __forceinline__ double l1_norm(const T* g, int m, int lane) {
    double acc = 0.0;
    for (int i = lane; i < m; i += 64) acc += fabs((double)g[i]);
    return wave_bcast_lane0(wave_sum_lane0(acc));
}

// mode 0: after the LQ solve -- convergence / barrier update, penalty update, merit at the iterate, first step length
// mode 1: after evaluating the trial point -- Armijo test, accept (copy) or halve
// Convergence test, barrier update and merit at the iterate (after the LQ solve), one wave per problem.  Returns, the same
// in every lane, whether the problem takes no step this iteration (`lsd`) and the step length of its first trial (`al`).
template <typename T>
__device__ __forceinline__ StepInfo<T> step_info_from(const SolverArgs& a, int b, const T* __restrict__ f) {
    const T* info = (const T*)a.info + (size_t)b * INFO_N;
    return StepInfo<T>{info[INFO_GINF], info[INFO_STEP], info[INFO_ZINF], info[INFO_LAM], info[INFO_D0], info[INFO_AMAX],
                       info[INFO_LSK], info[INFO_LSA], info[INFO_RESTARTS], info[INFO_LSR],
                       ((const T*)a.mu)[b], ((const T*)a.pen)[b], f[b], a.status[b]};
}
// zp, gp: the problem's iterate and defects (global memory, or the Riccati kernel's LDS copies)
template <typename T>
__device__ __forceinline__ void solver_merit0_body(const SolverArgs& a, int b, int lane, const T* __restrict__ f,
                                                   const T* __restrict__ zp, const T* __restrict__ gp, const StepInfo<T>& si,
                                                   int& lsd, T& al, const T* lbp, const T* ubp) {
    T* mu = (T*)a.mu; T* nu = (T*)a.pen; T* alpha = (T*)a.alpha; T* phi0 = (T*)a.phi0; T* dir = (T*)a.dir;
    const T* lb = lbp ? lbp : (const T*)a.lb;
    const T* ub = ubp ? ubp : (const T*)a.ub;
    const int H = a.H, nx = a.nx;
    lsd = 1;
    al = T(0);
    if (si.status >= 0) { if (lane == 0) a.lsdone[b] = 1; return; }
    if (si.restarts < T(0)) {    // the Riccati sweep ran out of attempts: no step this iteration, still unconverged
        if (lane == 0) { a.lsdone[b] = 1; atomicAdd(a.n_active, 1); }
        return;
    }
    // converged for the current barrier parameter?  Sub-problems with mu above its floor are only solved to
    // an accuracy proportional to mu (kappa = 10); the last one to the requested tolerances.
    const T mub = si.mu;
    const bool last_mu = !(mub > (T)a.mu_min * T(1.0001));
    const T tg = last_mu ? (T)a.tol_g : fmax((T)a.tol_g, T(10) * mub);
    const T tsx = last_mu ? (T)a.tol_step * (T(1) + si.zinf) : fmax((T)a.tol_step, T(10) * mub) * (T(1) + si.zinf);
    const bool conv = si.ginf <= tg && si.step <= tsx;
    if (conv) {
        if (!last_mu) {
            // superlinear decrease: mu <- max(mu_min, min(mu_factor * mu, mu^1.5))
            if (lane == 0) mu[b] = fmax(fmin(mub * (T)a.mu_factor, mub * sqrt(mub)), (T)a.mu_min);
            // new sub-problem: skip this step (direction was computed for the old mu).  (Lowering mu one iteration
            // late instead, from the previous iteration's norms, so that no iteration is skipped, was measured:
            // same median iteration count, slightly fewer problems converged within 40 / 80 / 160 iterations.)
            if (lane == 0) { a.lsdone[b] = 1; atomicAdd(a.n_active, 1); }
        } else {
            if (lane == 0) { a.status[b] = 0; a.lsdone[b] = 1; a.iters_done[b] = a.cur_it + 1; }
        }
        return;
    }
    const double bar = barrier_value<T>(zp, lb, ub, a.n, mub, lane);
    const double g1 = l1_norm<T>(gp, H * nx, lane);
    // deferred backtracking: a problem whose last trial was rejected stands where it stood; this iteration
    // recomputed the same direction and tries it at half the rejected length
    lsd = 0;
    al = si.lsk > T(0) ? fmin(si.amax, si.lsa) : si.amax;
    if (lane == 0) {
        const T nun = fmax(si.nu, T(1.5) * si.lam + T(1e-3));
        nu[b] = nun;
        phi0[b] = (T)((double)si.f + bar + (double)nun * g1);
        dir[b] = si.d0 - nun * (T)g1;
        alpha[b] = al;
        // a retry iteration re-solves with the damping the opening iteration had to raise, so its own restart count
        // is zero: remember the opening one, or the accept below relaxes a term that was only just raised
        T* infw = (T*)a.info + (size_t)b * INFO_N;
        infw[INFO_LSR] = si.lsk > T(0) ? fmax(si.lsr, si.restarts) : si.restarts;
        a.lsdone[b] = 0;
        atomicAdd(a.n_active, 1);
    }
}

// After the LQ solve, ONE launch per iteration (three before: dual steps, merit at the iterate, first trial point): the
// dual steps of the bounds and their step length, the convergence test / barrier update / merit value, and the first
// trial point Zt = Z + alpha dz of the problems that take a step.  One wave per problem.
template <typename T>
__global__ __launch_bounds__(64) void solver_step_kernel(SolverArgs a, const T* __restrict__ f, const T* __restrict__ Zcur,
                                                         T* __restrict__ Zt) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b == 0 && lane == 0) *a.n_pending = 0;          // counted by the trial's acceptance test (solver_merit_kernel)
    if (b >= a.B) return;
    const T* z = Zcur + (size_t)b * a.n;
    const T* dz = (const T*)a.dz + (size_t)b * a.n;
    const StepInfo<T> si = step_info_from<T>(a, b, f);
    solver_dual_body<T>(a, b, lane, z, dz, si.mu, si.status);
    int lsd;
    T al;
    solver_merit0_body<T>(a, b, lane, f, z, (const T*)a.g + (size_t)b * a.m, si, lsd, al);
    for (int i = lane; i < a.n; i += 64) Zt[(size_t)b * a.n + i] = lsd ? z[i] : fma(al, dz[i], z[i]);
}

// Acceptance test of a trial point (Armijo on the l1 merit), one wave per problem
template <typename T>
__device__ __forceinline__ void solver_merit_body(const SolverArgs& a, const T* __restrict__ Zt, const T* __restrict__ gt,
                                                  const T* __restrict__ ft, T* __restrict__ Zcur, int last_ls,
                                                  const int* __restrict__ list_in, int* __restrict__ list_out,
                                                  int publish, int slot, int lane, int* n_active_word, int tag, int soc) {
    // list_in: the trial buffers (Zt, gt, ft) hold only the problems that were still searching after the previous trial,
    // densely, in the order of that list (inner-loop backtracking); null: one slot per problem.  list_out: the problems
    // this trial rejects are appended for the next one.  slot / lane: the problem slot this wave works on; n_active_word /
    // tag: the convergence counter that is published (and cleared) and the iteration number it is published under.
    const int b = list_in ? list_in[slot] : slot;
    // the convergence counter of the NEXT iteration, when its test runs inside the Riccati kernel (the host's copy of this
    // iteration's count was issued before this launch)
    if (slot == 0 && lane == 0 && publish) {
        // first acceptance launch of the iteration: the iteration's convergence counter is complete (the Riccati / step
        // kernel that counts ran earlier in the stream).  It goes to pinned host memory by a plain store -- the host reads
        // it there whenever it likes: no copy launch, no event, no drained stream
        if (a.hpub) {
            const int v = *n_active_word;
            a.hpub[1] = v;
            if (v == 0 && a.hpub[2] == 0) a.hpub[2] = tag;
            __threadfence_system();
            a.hpub[0] = tag;
            __threadfence_system();
        }
        if (a.fuse_step) *n_active_word = 0;
    }
    if (b >= a.B) return;
    T* mu = (T*)a.mu; T* nu = (T*)a.pen; T* reg = (T*)a.reg; T* alpha = (T*)a.alpha; T* phi0 = (T*)a.phi0; T* dir = (T*)a.dir;
    const T* info = (const T*)a.info + (size_t)b * INFO_N;
    const T* lb = (const T*)a.lb;
    const T* ub = (const T*)a.ub;
    const int H = a.H, nx = a.nx;
    // objective value of the trial point, when no launch has left it in `ft` (shapes without the fused evaluation, and the
    // compiled shape whose trial launch computes the Lagrangian blocks instead): the wave has the point in hand, and a
    // launch of its own for one number per problem costs more than the number.  First in the kernel, before the problem's
    // scalars are waited for: its loads share their round trip.  (carry: the trial point's objective gradient is written too,
    // it is the next iterate's)
    // The kernel is a chain of global round trips of a lone wave (~1.5 us each): everything that does not depend on the
    // decision is requested HERE, before anything is waited for -- the problem's scalars, this lane's first element of every
    // array the merit value and the acceptance read (the loops below take it from registers in their first trip), the
    // first trips of the carried-over tiles / blocks -- and the objective's own loads follow in the same flight.
    const int done = a.lsdone[b];
    // (soc: the second-order-correction trial of a step that was just rejected -- its length is the one that trial had;
    //  alpha[b] was halved on the rejection)
    const T mub = mu[b], nub = nu[b], al = soc ? alpha[b] * T(2) : alpha[b], ph0 = phi0[b], drb = dir[b];
    T ftb = ft ? ft[slot] : T(0);
    const T az = a.primal_dual ? ((const T*)a.alz)[b] : T(0);
    const T lsr = info[INFO_LSR], lsk = info[INFO_LSK], regb = reg[b];
    const T* zt = Zt + (size_t)slot * a.n;
    const T* gtb = gt + (size_t)slot * a.m;
    const bool in0 = lane < a.n, isl0 = lane < H * nx, pdl = a.primal_dual != 0;
    const T zi_0 = in0 ? zt[lane] : T(0), lo_0 = in0 ? lb[lane] : T(0), hi_0 = in0 ? ub[lane] : T(0);
    const T gi_0 = isl0 ? gtb[lane] : T(0);
    const T l0_0 = isl0 ? ((const T*)a.lam)[(size_t)b * a.m + lane] : T(0), l1_0 = isl0 ? ((const T*)a.lamn)[(size_t)b * a.m + lane] : T(0);
    const T zl_0 = in0 && pdl ? ((const T*)a.zl)[(size_t)b * a.n + lane] : T(0), dl_0 = in0 && pdl ? ((const T*)a.dzl)[(size_t)b * a.n + lane] : T(0);
    const T zu_0 = in0 && pdl ? ((const T*)a.zu)[(size_t)b * a.n + lane] : T(0), du_0 = in0 && pdl ? ((const T*)a.dzu)[(size_t)b * a.n + lane] : T(0);
    constexpr int PRE_T = 2, PRE_H = 3;       // trips of the tile / block copies held in registers (2/1: 120 / 180 elements)
    const int ntl_c = H * nx * (nx + a.nu), nhb_c = H * (nx + a.nu) * (nx + a.nu);
    T tpre[PRE_T], hpre[PRE_H];
    if (a.carry) {
        #pragma unroll
        for (int k = 0; k < PRE_T; ++k) tpre[k] = lane + 64 * k < ntl_c ? ((const T*)a.tiles_t)[(size_t)b * ntl_c + lane + 64 * k] : T(0);
        #pragma unroll
        for (int k = 0; k < PRE_H; ++k) hpre[k] = a.hblk_t && lane + 64 * k < nhb_c ? ((const T*)a.hblk_t)[(size_t)b * nhb_c + lane + 64 * k] : T(0);
    }
    if (!ft) ftb = (T)wave_bcast_lane0(objective_row_value<T>(b, lane, H, nx, a.nu, a.oo, (const T*)a.obj, zt,
                                                                 a.carry ? (T*)a.grad_t : (T*)nullptr));
    if (done) return;
    // log-barrier value and l1 norm of the defects at the trial point, one loop (the loads of both in flight together;
    // per-lane order and tree as barrier_value / l1_norm)
    double accb = 0.0, accg = 0.0;
    for (int i = lane; i < a.n; i += 64) {
        const bool first = i == lane;
        const T zi = first ? zi_0 : zt[i], lo = first ? lo_0 : lb[i], hi = first ? hi_0 : ub[i];
        const T gi = i < H * nx ? (first ? gi_0 : gtb[i]) : T(0);
        if (mub > T(0)) {
            if (lo > -std::numeric_limits<T>::max()) accb -= (double)mub * log((double)(zi - lo));
            if (hi < std::numeric_limits<T>::max()) accb -= (double)mub * log((double)(hi - zi));
        }
        if (i < H * nx) accg += fabs((double)gi);
    }
    const double bar = wave_bcast_lane0(wave_sum_lane0(accb));
    const double g1 = wave_bcast_lane0(wave_sum_lane0(accg));
    const T phit = (T)((double)ftb + bar + (double)nub * g1);
    // Armijo on the l1 merit; the directional derivative is negative for a descent direction
    // the slack absorbs the rounding of the merit value itself (f is a sum of ~n terms in T): without a dtype-sized one
    // an fp32 iterate close to its solution fails the test on noise, halves its step six times and gets damped
    const T slack = (T)a.armijo_slack;
    // Non-monotone reference (Grippo-Lampariello-Lucidi): the trial point has to improve on the LARGEST merit value of the
    // last iterates, not on the current one.  Near a feasible iterate a full SQP step moves along the constraint manifold
    // and raises the l1 merit through the second-order constraint violation times a penalty that tracks the multipliers
    // (order 250-700 at configs[2] dims): the monotone test cuts such steps to 2-5 % of their length for dozens of
    // iterations -- the Maratos effect (tools/c3_slow_trace.py).  History is only comparable under the same penalty and
    // barrier parameter; it starts over when either changes.
    T phref = ph0;
    const T phn = info[INFO_PHN];
    const bool hist_ok = a.nonmono > 0 && phn > T(0) && info[INFO_PHNU] == nub && info[INFO_PHMU] == mub;
    // Watchdog: after a step that was accepted UPHILL the next full step has to bring the merit below the value in front
    // of that step (a Maratos step is followed by one that restores feasibility and does; an iterate chattering across a
    // relu kink is not, and would otherwise be waved through forever), and a shortened one has to pass the monotone test.
    const bool up = info[INFO_PHUP] > T(0);
    if (hist_ok && !up) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (a.nonmono > k && phn > (T)k) phref = fmax(phref, info[INFO_PH1 + k]);
    } else if (hist_ok && up && al >= T(1)) {
        phref = info[INFO_PH1];
    }
    const bool ok = (phit == phit) && phit <= phref + T(1e-4) * al * fmin(drb, T(0)) + slack * fabs(phref);
    if (ok) {
        if (lane == 0 && a.nonmono > 0) {
            T* infw = (T*)a.info + (size_t)b * INFO_N;
            const T h1 = info[INFO_PH1], h2 = info[INFO_PH2], h3 = info[INFO_PH3];
            infw[INFO_PH4] = hist_ok ? h3 : ph0;
            infw[INFO_PH3] = hist_ok ? h2 : ph0;
            infw[INFO_PH2] = hist_ok ? h1 : ph0;
            infw[INFO_PH1] = ph0;
            infw[INFO_PHN] = hist_ok ? fmin(phn + T(1), T(4)) : T(1);
            infw[INFO_PHNU] = nub;
            infw[INFO_PHMU] = mub;
            infw[INFO_PHUP] = phit > ph0 ? T(1) : T(0);
        }
        T* lam = (T*)a.lam + (size_t)b * a.m;
        const T* lamn = (const T*)a.lamn + (size_t)b * a.m;
        // bound multipliers: their own step length, then kept within a factor kappa of mu / slack at the new point
        // (the safeguard of primal-dual interior-point codes: a multiplier far from the central path is pulled back)
        const bool duals = a.primal_dual && mub > T(0);
        const T kap = T(1e10);
        T* zl = (T*)a.zl + (size_t)b * a.n;
        T* zu = (T*)a.zu + (size_t)b * a.n;
        const T* dzl = (const T*)a.dzl + (size_t)b * a.n;
        const T* dzu = (const T*)a.dzu + (size_t)b * a.n;
        for (int i = lane; i < a.n; i += 64) {
            const bool first = i == lane;
            const T zi = first ? zi_0 : zt[i], lo = first ? lo_0 : lb[i], hi = first ? hi_0 : ub[i];
            const bool lof = duals && lo > -std::numeric_limits<T>::max(), hif = duals && hi < std::numeric_limits<T>::max();
            const bool isl = i < H * nx;
            const T l0 = isl ? (first ? l0_0 : lam[i]) : T(0), l1 = isl ? (first ? l1_0 : lamn[i]) : T(0);
            const T zl0 = lof ? (first ? zl_0 : zl[i]) : T(0), dl = lof ? (first ? dl_0 : dzl[i]) : T(0);
            const T zu0 = hif ? (first ? zu_0 : zu[i]) : T(0), du = hif ? (first ? du_0 : dzu[i]) : T(0);
            Zcur[(size_t)b * a.n + i] = zi;
            if (isl) lam[i] = fma(al, l1 - l0, l0);
            if (a.carry) {      // the accepted point's evaluation is the next iterate's: no launch for it
                ((T*)a.grad)[(size_t)b * a.n + i] = ((const T*)a.grad_t)[(size_t)b * a.n + i];
                if (i < a.m) ((T*)a.g)[(size_t)b * a.m + i] = gtb[i];
            }
            if (lof) {
                const T c = mub / (zi - lo);
                zl[i] = fmin(fmax(fma(az, dl, zl0), c / kap), c * kap);
            }
            if (hif) {
                const T c = mub / (hi - zi);
                zu[i] = fmin(fmax(fma(az, du, zu0), c / kap), c * kap);
            }
        }
        if (a.carry) {
            const int ntl = H * nx * (nx + a.nu);
            const T* ts = (const T*)a.tiles_t + (size_t)b * ntl;
            T* td = (T*)a.tiles + (size_t)b * ntl;
            #pragma unroll
            for (int k = 0; k < PRE_T; ++k)
                if (lane + 64 * k < ntl) td[lane + 64 * k] = tpre[k];
            for (int i = lane + 64 * PRE_T; i < ntl; i += 64) td[i] = ts[i];
            if (a.hblk_t) {
                const int nhb = nhb_c;
                const T* hs = (const T*)a.hblk_t + (size_t)b * nhb;
                T* hd = (T*)a.hblk + (size_t)b * nhb;
                #pragma unroll
                for (int k = 0; k < PRE_H; ++k)
                    if (lane + 64 * k < nhb) hd[lane + 64 * k] = hpre[k];
                for (int i = lane + 64 * PRE_H; i < nhb; i += 64) hd[i] = hs[i];
            }
            for (int i = a.n + lane; i < a.m; i += 64) ((T*)a.g)[(size_t)b * a.m + i] = gtb[i];      // (rows beyond n, if any)
            if (lane == 0) ((T*)a.f_it)[b] = ftb;
        }
        // relax the damping only after a sweep that went through at the first attempt: a term that had to be raised
        // this iteration would fail again right away and cost a full extra sweep
        if (lane == 0) {
            a.lsdone[b] = 1;
            // relax the damping only after a sweep that went through at the first attempt: a term that had to be raised
            // this iteration would fail again right away and cost a full extra sweep
            if (!(lsr > T(0))) reg[b] = fmax(regb * (T)a.reg_relax, T(1e-9));
            ((T*)a.info)[(size_t)b * INFO_N + INFO_LSK] = T(0);
        }
    } else if (lane == 0 && soc) {
        // the corrected point is no better: the backtracking goes on from the halved length the rejection left
        const int p = atomicAdd(a.n_pending, 1);
        if (list_out) list_out[p] = b;
    } else if (lane == 0) {
        alpha[b] = al * T(0.5);
        if (last_ls == 2) {
            // deferred backtracking (one trial per outer iteration): remember the halved length for the next iteration;
            // after max_ls rejections in a row the direction is given up and the next LQ solve is damped
            T* inf = (T*)a.info + (size_t)b * INFO_N;
            const T k = lsk + T(1);
            a.lsdone[b] = 1;
            if (k >= (T)a.max_ls) { inf[INFO_LSK] = T(0); reg[b] = fmin(fmax(regb * T(100), T(1e-6)), T(1e8)); }
            else { inf[INFO_LSK] = k; inf[INFO_LSA] = al * T(0.5); }
        } else {
            if (!last_ls) {     // still searching: the host polls the counter to stop the backtracking early
                const int p = atomicAdd(a.n_pending, 1);
                if (list_out) list_out[p] = b;
            }
            if (last_ls) {   // no progress: damp the next LQ solve
                a.lsdone[b] = 1;
                reg[b] = fmin(fmax(regb * T(100), T(1e-6)), T(1e8));
                ((T*)a.info)[(size_t)b * INFO_N + INFO_LSK] = T(0);
            }
        }
    }
}

// adaptive backtracking: the problems still searching after the first trial are a small minority -> they stand still this
// iteration and retry their (recomputed, identical) direction at the halved length in the next one
template <typename T>
__global__ __launch_bounds__(256) void solver_defer_kernel(SolverArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B || a.lsdone[b]) return;
    T* inf = (T*)a.info + (size_t)b * INFO_N;
    T* reg = (T*)a.reg;
    const T k = inf[INFO_LSK] + T(1);
    a.lsdone[b] = 1;
    if (k >= (T)a.max_ls) { inf[INFO_LSK] = T(0); reg[b] = fmin(fmax(reg[b] * T(100), T(1e-6)), T(1e8)); }
    else { inf[INFO_LSK] = k; inf[INFO_LSA] = ((const T*)a.alpha)[b]; }
}

// seq > 0 (inner-loop backtracking): the block that finishes last publishes the number of problems still searching to
// pinned host memory under that sequence number -- the host reads it there instead of copying it behind a drained stream
template <typename T>
__global__ __launch_bounds__(64) void solver_merit_kernel(SolverArgs a, const T* __restrict__ Zt, const T* __restrict__ gt,
                                                          const T* __restrict__ ft, T* __restrict__ Zcur, int last_ls,
                                                          const int* __restrict__ list_in, int* __restrict__ list_out,
                                                          int publish, int seq, int soc = 0) {
    solver_merit_body<T>(a, Zt, gt, ft, Zcur, last_ls, list_in, list_out, publish, (int)blockIdx.x, (int)threadIdx.x, a.n_active,
                         a.cur_it + 1, soc);
    if (seq > 0 && threadIdx.x == 0) {
        __threadfence();                                    // this block's list entry and count are out
        const int t = atomicAdd(a.n_done, 1);
        if (t == (int)gridDim.x - 1) {
            __threadfence();
            const int v = atomicAdd(a.n_pending, 0);
            *a.n_done = 0;
            a.hpub[5] = v;
            __threadfence_system();
            a.hpub[4] = seq;
            __threadfence_system();
        }
    }
}

// Next trial point of the problems still searching (inner-loop backtracking), gathered densely in the order of `list`
// together with their initial states and per-problem extras: the trial evaluation then runs over those problems only
// (a trial over the whole active batch cost the same whether one problem or all of them were still searching)
template <typename T>
__global__ __launch_bounds__(64) void solver_trial_list_kernel(int n, int nx, int ex_per, const int* __restrict__ list,
                                                               const T* __restrict__ Z, const T* __restrict__ dz,
                                                               const T* __restrict__ alpha, const T* __restrict__ X0,
                                                               const T* __restrict__ ex, T* __restrict__ Zt,
                                                               T* __restrict__ X0p, T* __restrict__ exp_,
                                                               int* __restrict__ n_pending) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot == 0 && lane == 0) *n_pending = 0;         // counter of the problems still searching after this trial
    const int b = list[slot];
    const T al = alpha[b];
    for (int i = lane; i < n; i += 64) Zt[(size_t)slot * n + i] = fma(al, dz[(size_t)b * n + i], Z[(size_t)b * n + i]);
    for (int i = lane; i < nx; i += 64) X0p[(size_t)slot * nx + i] = X0[(size_t)b * nx + i];
    for (int i = lane; i < ex_per; i += 64) exp_[(size_t)slot * ex_per + i] = ex[(size_t)b * ex_per + i];
}

// Second-order correction against the Maratos effect (inner-loop backtracking).  A full SQP step along a curved constraint
// manifold leaves defects c+ = c(z + alpha d) of second order in |d|; times the l1 merit's penalty (which tracks multipliers of
// order 250 - 700 at configs[2] dims) they outweigh the decrease of the objective and the step is cut to a few per cent for
// dozens of iterations (DESIGN "Batched solver", tools/c3_slow_trace.py).  The correction removes them to first order with the
// Jacobian of the ITERATE, which is already in hand: states only, dx_t = A_t dx_{t-1} + c+_t (dx_{-1} = 0: x0 is data) --
// a particular solution of  A delta = -c+  of the size of c+ itself -- one forward recursion per rejected problem, a wave each.
// The corrected points of the problems on `list` go to Zs densely (with their x0 / extras, like solver_trial_list_kernel),
// one defect-only row launch evaluates them, and the acceptance test takes them at the rejected trial's step length.  A
// corrected state that leaves its bounds makes the barrier term NaN and is rejected by the test itself.
template <typename T>
__global__ __launch_bounds__(64) void solver_soc_kernel(int n, int nx, int nu, int H, int m, int ex_per, const int* __restrict__ list,
                                                        const T* __restrict__ Zt, const T* __restrict__ gt,
                                                        const T* __restrict__ tiles, const T* __restrict__ X0,
                                                        const T* __restrict__ ex, T* __restrict__ Zs, T* __restrict__ X0p,
                                                        T* __restrict__ exp_, int* __restrict__ n_pending) {
    const int slot = blockIdx.x, lane = threadIdx.x;
    if (slot == 0 && lane == 0) *n_pending = 0;
    const int b = list[slot];
    const int nin = nx + nu;
    const T* zt = Zt + (size_t)b * n;                   // (the rejected trial ran over every active problem: slot == problem)
    const T* ct = gt + (size_t)b * m;
    T* zs = Zs + (size_t)slot * n;
    T dx = T(0);                                        // lane i < nx: dx_{t-1}[i]
    for (int t = 0; t < H; ++t) {
        const T* At = tiles + ((size_t)b * H + t) * nx * nin;        // row i: dPhi_i / d[x_{t-1} | u_t]
        T acc = lane < nx ? ct[t * nx + lane] : T(0);
        for (int e = 0; e < nx; ++e) {
            const T de = __shfl(dx, e);
            if (lane < nx) acc = fma(At[lane * nin + e], de, acc);
        }
        dx = acc;
        if (lane < nx) zs[t * nx + lane] = zt[t * nx + lane] + dx;
    }
    for (int i = H * nx + lane; i < n; i += 64) zs[i] = zt[i];
    for (int i = lane; i < nx; i += 64) X0p[(size_t)slot * nx + i] = X0[(size_t)b * nx + i];
    for (int i = lane; i < ex_per; i += 64) exp_[(size_t)slot * ex_per + i] = ex[(size_t)b * ex_per + i];
}

// strictly interior start + per-problem state
// the caller's initial guess, moved into the interior of its bounds.  Distance kept from a bound: 1e-2 for the default
// barrier parameter; a warm start (small mu_init) stays closer to the active bounds it was handed
template <typename T>
__device__ __forceinline__ T start_inside(T z, T lo, T hi, T mu0, int has_bounds) {
    const bool flo = lo > -std::numeric_limits<T>::max(), fhi = hi < std::numeric_limits<T>::max();
    T marg = has_bounds ? fmin(T(1e-2), fmax(T(10) * mu0, T(1e-6))) : T(1e-2);
    if (flo && fhi) marg = fmin(marg, T(0.25) * (hi - lo));
    if (flo) z = fmax(z, lo + marg);
    if (fhi) z = fmin(z, hi - marg);
    return z;
}

template <typename T>
__global__ __launch_bounds__(256) void solver_init_kernel(int B, int n, T* __restrict__ Z, const T* __restrict__ lb,
                                                          const T* __restrict__ ub, T* mu, T* nu, T* reg, int* status,
                                                          int* orig, int* iters_done, T* info, T* zl, T* zu, T mu0,
                                                          T reg0, int has_bounds, const T* __restrict__ Zin,
                                                          const T* __restrict__ X0in, T* __restrict__ X0c, int nx,
                                                          T* __restrict__ lam0, int m) {
    // (the working copies of the caller's Z and X0 and the zero multipliers come from this launch too: three copy / fill
    // launches less in a prologue that a single-problem closed loop pays every MPC step)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)B * nx) X0c[i] = X0in[i];
    if (i < (size_t)B * m) lam0[i] = T(0);
    if (i < (size_t)B) {
        mu[i] = has_bounds ? mu0 : T(0); nu[i] = T(1); reg[i] = reg0; status[i] = -1;
        orig[i] = (int)i; iters_done[i] = 0;
        for (int k = 0; k < INFO_N; ++k) info[i * INFO_N + k] = std::numeric_limits<T>::max();   // "no previous step"
        info[i * INFO_N + INFO_LSK] = T(0);
        info[i * INFO_N + INFO_LSR] = T(0);
        info[i * INFO_N + INFO_PHN] = T(0);
        info[i * INFO_N + INFO_PHUP] = T(0);
    }
    if (i >= (size_t)B * n) return;
    const int k = (int)(i % n);
    const T lo = lb[k], hi = ub[k];
    const bool flo = lo > -std::numeric_limits<T>::max(), fhi = hi < std::numeric_limits<T>::max();
    const T z = start_inside(Zin[i], lo, hi, mu0, has_bounds);
    Z[i] = z;
    if (zl) {   // bound multipliers on the central path of the first barrier parameter
        zl[i] = flo ? mu0 / (z - lo) : T(0);
        zu[i] = fhi ? mu0 / (hi - z) : T(0);
    }
}

// ---- compaction of the unconverged problems (the lock-step batch would otherwise launch every kernel B wide for a
// handful of stragglers).  One workgroup computes the stable partition of slots [0, Bact): perm[new slot] = old slot,
// unconverged first; the rows of every per-problem array that lives across iterations are then gathered into the
// second buffer set.
__global__ __launch_bounds__(1024) void solver_partition_kernel(int Bact, const int* __restrict__ status,
                                                                int* __restrict__ perm, int* __restrict__ count) {
    __shared__ int s_cnt[1024];
    __shared__ int s_tot;
    const int tid = threadIdx.x;
    const int per = (Bact + 1023) / 1024;
    const int lo = tid * per, hi = min(Bact, lo + per);
    int c = 0;
    for (int b = lo; b < hi; ++b) c += status[b] < 0 ? 1 : 0;
    s_cnt[tid] = c;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) { const int v = s_cnt[i]; s_cnt[i] = run; run += v; }
        s_tot = run;
        *count = run;
    }
    __syncthreads();
    int pa = s_cnt[tid], pi = s_tot + (lo - s_cnt[tid]);     // next active / inactive slot of this thread's range
    for (int b = lo; b < hi; ++b) {
        if (status[b] < 0) perm[pa++] = b; else perm[pi++] = b;
    }
}

struct CompactArrays {   // (src, dst, elements per problem, bytes per element); up to 14 arrays
    const void* src[14];
    void* dst[14];
    int per[14];
    int esz[14];
    int n;
};

__global__ __launch_bounds__(256) void solver_gather_kernel(int Bact, const int* __restrict__ perm, CompactArrays ca) {
    const int slot = blockIdx.x;
    if (slot >= Bact) return;
    const int from = perm[slot];
    for (int k = 0; k < ca.n; ++k) {
        const int words = ca.per[k] * ca.esz[k] / 4;          // every array is a multiple of 4 bytes per problem
        const unsigned* s = (const unsigned*)ca.src[k] + (size_t)from * words;
        unsigned* d = (unsigned*)ca.dst[k] + (size_t)slot * words;
        for (int i = threadIdx.x; i < words; i += 256) d[i] = s[i];
    }
}

// results back in the caller's order: Z_out[orig[slot]] = Z[slot]; unfinished problems are FAIL (1)
template <typename T>
__global__ __launch_bounds__(256) void solver_scatter_kernel(int B, int n, const T* __restrict__ Zc,
                                                             const int* __restrict__ status_c,
                                                             const int* __restrict__ orig,
                                                             const int* __restrict__ iters_c, T* __restrict__ Zout,
                                                             int* __restrict__ status_out, int* __restrict__ iters_out) {
    const int slot = blockIdx.x;
    if (slot >= B) return;
    const int o = orig[slot];
    for (int i = threadIdx.x; i < n; i += 256) Zout[(size_t)o * n + i] = Zc[(size_t)slot * n + i];
    if (threadIdx.x == 0) {
        status_out[o] = status_c[slot] == 0 ? 0 : 1;
        if (iters_out) iters_out[o] = iters_c[slot];
    }
}

// ---- rolling-window models (model/tensorflow.py:132-340: the network of step t reads the last w states and controls).
// The Riccati sweeps need a STAGE-wise problem: the window is made the state,
//     s_tau = [x_tau, x_{tau-1}, .., x_{tau-w+1} | u_{tau-1}, .., u_{tau-w+1}]            (ns = w nx + (w-1) nu entries)
//     s_{t+1} = [Phi(window of step t) ; shifted copies of s_t and u_t]
// (x_tau indexes [x0 ; states], slots before the horizon are x0 and the bound history and make up s_0), and the solver
// runs UNCHANGED on that plain problem with stage dimensions (ns, nu).  The callbacks stay the handle's own window kernels
// on the caller's variables: before an evaluation the primary copies [x_tau | u_t] are gathered out of the augmented
// iterate; after it the defects, tiles, Lagrangian blocks and the objective gradient are spread into augmented form --
// a tile / block column of the window IS a column of (s_t, u_t), and the shift rows are constant 0/1 rows.  The shift
// rows are linear and hold at the start (the copies are built from the primaries), so every Newton step keeps them and
// the network is always evaluated at the point the augmented iterate stands on.
struct RollTables {        // device, int32; built once per workspace (they depend on the handle's shape only)
    int* src = nullptr;    // (n_s)  augmented variable -> caller's variable v >= 0, or -(1 + k): entry k of the problem's data
                           //        [x0 (nx) | hist_x (w-1, nx) | hist_u (w-1, nu)]
    int* src0 = nullptr;   // (ns)   the same for s_0 (data only)
    int* prim = nullptr;   // (n_r)  caller's variable -> its primary copy in the augmented vector
    int* isprim = nullptr; // (n_s)  1 where the augmented variable is a primary copy
    int* dcol = nullptr;   // (ns + nu) column of (s_t, u_t) -> window column of the handle's tiles / blocks
    int* shift = nullptr;  // (ns)   row i >= nx of a stage: column of (s_t, u_t) it copies
};

template <typename T>
__device__ __forceinline__ T roll_data(int k, int b, int nx, int nu, int back, const T* __restrict__ X0,
                                       const T* __restrict__ hx, const T* __restrict__ hu) {
    if (k < nx) return X0[(size_t)b * nx + k];
    k -= nx;
    if (k < back * nx) return hx[(size_t)b * back * nx + k];
    return hu[(size_t)b * back * nu + (k - back * nx)];
}

// augmented start: primaries = the caller's guess moved inside its bounds (what solver_init_kernel would do to them:
// it then leaves them alone), copies = those same values, s_0 from the data
template <typename T>
__global__ __launch_bounds__(256) void roll_build_kernel(int B, int n_s, int ns, int n_r, int nx, int nu, int back,
                                                         RollTables rt, const T* __restrict__ Zin, const T* __restrict__ X0,
                                                         const T* __restrict__ hx, const T* __restrict__ hu,
                                                         const T* __restrict__ lb, const T* __restrict__ ub, T mu0,
                                                         int has_bounds, T* __restrict__ Zaug, T* __restrict__ S0) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)B * ns) {
        const int b = (int)(i / ns), k = (int)(i % ns);
        S0[i] = roll_data(-1 - rt.src0[k], b, nx, nu, back, X0, hx, hu);
    }
    if (i >= (size_t)B * n_s) return;
    const int b = (int)(i / n_s), k = (int)(i % n_s);
    const int v = rt.src[k];
    if (v < 0) { Zaug[i] = roll_data(-1 - v, b, nx, nu, back, X0, hx, hu); return; }
    const int pk = rt.prim[v];
    Zaug[i] = start_inside(Zin[(size_t)b * n_r + v], lb[pk], ub[pk], mu0, has_bounds);
}

// caller's variables (and multipliers of the network rows) out of the augmented ones
template <typename T>
__global__ __launch_bounds__(256) void roll_in_kernel(int B, int n_s, int n_r, int H, int nx, int ns, int m_r, RollTables rt,
                                                      const T* __restrict__ Zaug, T* __restrict__ Zr,
                                                      const T* __restrict__ lam_aug, T* __restrict__ lam_r) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (size_t)B * n_r) {
        const int b = (int)(i / n_r), v = (int)(i % n_r);
        Zr[i] = Zaug[(size_t)b * n_s + rt.prim[v]];
    }
    if (lam_r && i < (size_t)B * H * nx) {
        const int b = (int)(i / (H * nx)), r = (int)(i % (H * nx));
        lam_r[(size_t)b * m_r + r] = lam_aug[(size_t)b * H * ns + (r / nx) * ns + (r % nx)];
    }
}

// the window evaluation spread into augmented form; one workgroup per (problem, step).  Any of tiles / hblk / grad may be
// null (a trial point needs the defects only)
template <typename T>
__global__ __launch_bounds__(256) void roll_out_kernel(int H, int nx, int nu, int ns, int nin_w, int n_r, int m_r, RollTables rt,
                                                       const T* __restrict__ Zaug, const T* __restrict__ S0,
                                                       const T* __restrict__ g_r, const T* __restrict__ tiles_r,
                                                       const T* __restrict__ hblk_r, const T* __restrict__ grad_r,
                                                       T* __restrict__ g_aug, T* __restrict__ tiles_aug,
                                                       T* __restrict__ hblk_aug, T* __restrict__ grad_aug) {
    const int b = blockIdx.x / H, t = blockIdx.x % H, nin_s = ns + nu, n_s = H * nin_s;
    const T* za = Zaug + (size_t)b * n_s;
    // value of column c of (s_t, u_t)
    auto col_value = [&](int c) -> T {
        if (c >= ns) return za[H * ns + t * nu + (c - ns)];
        return t == 0 ? S0[(size_t)b * ns + c] : za[(t - 1) * ns + c];
    };
    for (int i = threadIdx.x; i < ns; i += 256)
        g_aug[(size_t)b * H * ns + t * ns + i] =
            i < nx ? g_r[(size_t)b * m_r + t * nx + i] : col_value(rt.shift[i]) - za[t * ns + i];
    if (tiles_aug) {
        T* ta = tiles_aug + ((size_t)b * H + t) * ns * nin_s;
        const T* tr = tiles_r + ((size_t)b * H + t) * nx * nin_w;
        for (int e = threadIdx.x; e < ns * nin_s; e += 256) {
            const int i = e / nin_s, c = e % nin_s;
            T v;
            if (i < nx) { const int d = rt.dcol[c]; v = d >= 0 ? tr[i * nin_w + d] : T(0); }
            else v = rt.shift[i] == c ? T(1) : T(0);
            ta[e] = v;
        }
    }
    if (hblk_aug) {
        T* ha = hblk_aug + ((size_t)b * H + t) * nin_s * nin_s;
        const T* hr = hblk_r + ((size_t)b * H + t) * nin_w * nin_w;
        for (int e = threadIdx.x; e < nin_s * nin_s; e += 256) {
            const int dp = rt.dcol[e / nin_s], dq = rt.dcol[e % nin_s];
            ha[e] = (dp >= 0 && dq >= 0) ? hr[dp * nin_w + dq] : T(0);
        }
    }
    if (grad_aug) {
        // this step's share of the gradient: the state block s_{t+1} and the control block u_t
        for (int i = threadIdx.x; i < nin_s; i += 256) {
            const int k = i < ns ? t * ns + i : H * ns + t * nu + (i - ns);
            grad_aug[(size_t)b * n_s + k] = rt.isprim[k] ? grad_r[(size_t)b * n_r + rt.src[k]] : T(0);
        }
    }
}

struct SolverWs {
    // rolling-window models: tables, the caller-side evaluation buffers and the augmented objective table
    RollTables rt;
    void *rZ = nullptr, *rg = nullptr, *rtiles = nullptr, *rhblk = nullptr, *rgrad = nullptr, *rlam = nullptr,
         *rZin = nullptr, *rS0 = nullptr, *rZout = nullptr, *robj = nullptr;
    std::vector<int> prim_host;               // (n_r) primary copy of every caller variable
    std::vector<unsigned char> robj_host;     // the augmented objective table as last uploaded
    void *Zt = nullptr, *f = nullptr, *ft = nullptr, *grad = nullptr, *g = nullptr, *gt = nullptr, *tiles = nullptr;
    void *tiles_t = nullptr, *grad_t = nullptr;     // trial point's tiles and objective gradient (carried over on acceptance)
    void *lam_t = nullptr, *hblk_t = nullptr;       // ... its multipliers and Lagrangian blocks (blocks + evaluation in one launch)
    void *X0p = nullptr, *exp_ = nullptr;           // initial states / extras of the problems still backtracking, dense
    void *Zsoc = nullptr, *gsoc = nullptr;          // second-order-correction trial points and their defects, dense
    int* pend[2] = {nullptr, nullptr};              // ... and their indices (two lists: one read, one appended to)
    void *lb = nullptr, *ub = nullptr, *mu = nullptr, *nu = nullptr, *reg = nullptr, *alpha = nullptr, *phi0 = nullptr,
         *dir = nullptr, *hblk = nullptr, *lam = nullptr, *lamn = nullptr, *sig = nullptr, *dz = nullptr, *Kst = nullptr, *kst = nullptr, *Pst = nullptr, *pst = nullptr,
         *tmp = nullptr;   // (info lives in infoc: it is read across iterations)
    int *lsdone = nullptr, *n_active = nullptr;
    // state that lives across iterations, in two buffer sets (compaction gathers from one into the other)
    void *Zc[2] = {nullptr, nullptr}, *X0c[2] = {nullptr, nullptr}, *lamc[2] = {nullptr, nullptr}, *muc[2] = {nullptr, nullptr},
         *nuc[2] = {nullptr, nullptr}, *regc[2] = {nullptr, nullptr}, *exc[2] = {nullptr, nullptr}, *infoc[2] = {nullptr, nullptr},
         *zlc[2] = {nullptr, nullptr}, *zuc[2] = {nullptr, nullptr};
    void *dzl = nullptr, *dzu = nullptr, *alz = nullptr, *bh = nullptr;
    int *stc[2] = {nullptr, nullptr}, *orig[2] = {nullptr, nullptr}, *itc[2] = {nullptr, nullptr};
    int *perm = nullptr, *count = nullptr;
    int* hpoll = nullptr;                      // pinned host: [0] convergence counter (blocking polls), [2] backtracking poll
    int *hpub = nullptr, *hpub_dev = nullptr;  // pinned host memory the device publishes the convergence counter to
    int cap = 0;
    int m = 0, n = 0;                         // row / variable counts the buffers were sized for (box rows change m)
    size_t ex_per = 0;
    std::vector<unsigned char> bounds_host;   // the bounds as last uploaded (lb | ub in the handle's dtype)
};

// polite busy-wait on a word the device writes (the waits are microseconds: no yield, no sleep)
inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

int lq_tmp_elems(int nx, int nu) { return 3 * nx * nx + 3 * nx * nu + nu * nu + 5 * nx + 3 * nu + 1 + (nx + 1) * nu; }

}  // namespace

void solver_free(Handle& h) {
    SolverWs* w = static_cast<SolverWs*>(h.solver_ws);
    if (!w) return;
    void** ptrs[] = {&w->Zt, &w->f, &w->ft, &w->grad, &w->g, &w->gt, &w->tiles, &w->lb, &w->ub, &w->mu, &w->nu, &w->reg,
                     &w->alpha, &w->phi0, &w->dir, &w->hblk, &w->lam, &w->lamn, &w->sig, &w->dz, &w->Kst, &w->kst, &w->Pst, &w->pst, &w->tmp,
                     &w->tiles_t, &w->grad_t, &w->X0p, &w->exp_, &w->lam_t, &w->hblk_t, &w->Zsoc, &w->gsoc};
    for (void** p : ptrs)
        if (*p) (void)hipFree(*p);
    if (w->lsdone) (void)hipFree(w->lsdone);
    for (int k = 0; k < 2; ++k)
        if (w->pend[k]) (void)hipFree(w->pend[k]);
    if (w->n_active) (void)hipFree(w->n_active);
    if (w->hpoll) (void)hipHostFree(w->hpoll);
    if (w->hpub) (void)hipHostFree(w->hpub);
    for (int k = 0; k < 2; ++k) {
        void* ps[] = {w->Zc[k], w->X0c[k], w->lamc[k], w->muc[k], w->nuc[k], w->regc[k], w->exc[k], w->infoc[k], w->stc[k],
                      w->orig[k], w->itc[k], w->zlc[k], w->zuc[k]};
        for (void* p : ps)
            if (p) (void)hipFree(p);
    }
    for (void* p : {w->dzl, w->dzu, w->alz, w->bh})
        if (p) (void)hipFree(p);
    if (w->perm) (void)hipFree(w->perm);
    if (w->count) (void)hipFree(w->count);
    for (void* p : {w->rZ, w->rg, w->rtiles, w->rhblk, w->rgrad, w->rlam, w->rZin, w->rS0, w->rZout, w->robj,
                    (void*)w->rt.src, (void*)w->rt.src0, (void*)w->rt.prim, (void*)w->rt.isprim, (void*)w->rt.dcol,
                    (void*)w->rt.shift})
        if (p) (void)hipFree(p);
    delete w;
    h.solver_ws = nullptr;
}

namespace {
// the handle's extras binding is pointed at the compacted copy while the solve runs; put back on every exit path
struct ExtraBindingGuard {
    Handle& h;
    const void* saved;
    int saved_B;
    explicit ExtraBindingGuard(Handle& hh) : h(hh), saved(hh.d_extra), saved_B(hh.extra_B) {}
    ~ExtraBindingGuard() { h.d_extra = saved; h.extra_B = saved_B; }
};
}  // namespace

template <typename T>
static int solve_impl(Handle& h, int B, const void* X0, void* Z, const double* lb, const double* ub,
                      const nempc_solver_opts& o, int32_t* status_dev, int32_t* iters_host, hipStream_t s) {
    // rolling-window models run as a plain problem in the augmented (window) state, see RollTables: nx, nin, n, m below are
    // the SOLVER's stage dimensions; *_r the handle's (the caller's variables, the callbacks' shapes)
    const bool rolling = h.w > 1;
    const int H = h.cfg.H, nx_r = h.cfg.nx, nu = h.cfg.nu, nin_r = h.nin, n_r = h.n, m_r = h.m, back = h.w - 1;
    const int nx = rolling ? h.w * nx_r + back * nu : nx_r, nin = rolling ? nx + nu : nin_r, n = rolling ? H * (nx + nu) : n_r,
              m = rolling ? H * nx : m_r;
    const size_t ex_per = (size_t)H * h.ne;
    if (!h.solver_ws) h.solver_ws = new SolverWs();
    {
        SolverWs& w = *static_cast<SolverWs*>(h.solver_ws);
        if (w.cap < B || w.ex_per != ex_per || w.m != m || w.n != n) {
            solver_free(h);
            h.solver_ws = new SolverWs();
            SolverWs& w2 = *static_cast<SolverWs*>(h.solver_ws);
            const size_t e = sizeof(T), Bn = (size_t)B;
            struct { void** p; size_t bytes; } al[] = {
                {&w2.Zt, Bn * n * e}, {&w2.f, Bn * e}, {&w2.ft, Bn * e}, {&w2.grad, Bn * n * e}, {&w2.g, Bn * m * e},
                {&w2.gt, Bn * m * e}, {&w2.tiles, Bn * H * nx * nin * e}, {&w2.lb, (size_t)n * e}, {&w2.ub, (size_t)n * e},
                {&w2.alpha, Bn * e}, {&w2.phi0, Bn * e},
                {&w2.dir, Bn * e}, {&w2.hblk, Bn * H * nin * nin * e}, {&w2.lamn, Bn * m * e},
                {&w2.sig, Bn * e}, {&w2.dz, Bn * n * e}, {&w2.Kst, Bn * H * nu * nx * e},
                {&w2.kst, Bn * H * nu * e}, {&w2.Pst, Bn * H * nx * nx * e}, {&w2.pst, Bn * H * nx * e},
                {&w2.tmp, Bn * lq_tmp_elems(nx, nu) * e}, {&w2.dzl, Bn * n * e}, {&w2.dzu, Bn * n * e}, {&w2.alz, Bn * e},
                {&w2.bh, Bn * n * e}, {&w2.tiles_t, Bn * H * nx * nin * e}, {&w2.grad_t, Bn * n * e},
                {&w2.X0p, Bn * nx * e}, {&w2.exp_, Bn * ex_per * e}, {&w2.lam_t, Bn * m * e}, {&w2.hblk_t, Bn * H * nin * nin * e},
                {&w2.Zsoc, Bn * n * e}, {&w2.gsoc, Bn * m * e}};
            for (auto& x : al) NEMPC_HIP(hipMalloc(x.p, x.bytes ? x.bytes : 16));
            for (int k = 0; k < 2; ++k) {
                struct { void** p; size_t bytes; } al2[] = {
                    {&w2.Zc[k], Bn * n * e}, {&w2.X0c[k], Bn * nx * e}, {&w2.lamc[k], Bn * m * e}, {&w2.muc[k], Bn * e},
                    {&w2.nuc[k], Bn * e}, {&w2.regc[k], Bn * e}, {&w2.exc[k], Bn * ex_per * e}, {&w2.infoc[k], Bn * INFO_N * e},
                    {&w2.zlc[k], Bn * n * e}, {&w2.zuc[k], Bn * n * e},
                    {(void**)&w2.stc[k], Bn * sizeof(int)}, {(void**)&w2.orig[k], Bn * sizeof(int)},
                    {(void**)&w2.itc[k], Bn * sizeof(int)}};
                for (auto& x : al2) NEMPC_HIP(hipMalloc(x.p, x.bytes ? x.bytes : 16));
            }
            NEMPC_HIP(hipMalloc((void**)&w2.lsdone, Bn * sizeof(int)));
            for (int k = 0; k < 2; ++k) NEMPC_HIP(hipMalloc((void**)&w2.pend[k], Bn * sizeof(int)));
            NEMPC_HIP(hipMalloc((void**)&w2.n_active, 8 * sizeof(int)));   // [unconverged, still backtracking, blocks done, -, unconverged (odd iterations)]
            NEMPC_HIP(hipMemset(w2.n_active, 0, 8 * sizeof(int)));
            NEMPC_HIP(hipHostMalloc((void**)&w2.hpoll, 4 * sizeof(int), hipHostMallocDefault));
            NEMPC_HIP(hipHostMalloc((void**)&w2.hpub, 8 * sizeof(int), hipHostMallocMapped));
            NEMPC_HIP(hipHostGetDevicePointer((void**)&w2.hpub_dev, w2.hpub, 0));
            NEMPC_HIP(hipMalloc((void**)&w2.perm, Bn * sizeof(int)));
            NEMPC_HIP(hipMalloc((void**)&w2.count, sizeof(int)));
            if (rolling) {
                struct { void** p; size_t bytes; } alr[] = {
                    {&w2.rZ, Bn * n_r * e}, {&w2.rg, Bn * m_r * e}, {&w2.rtiles, Bn * H * nx_r * nin_r * e},
                    {&w2.rhblk, Bn * H * nin_r * nin_r * e}, {&w2.rgrad, Bn * n_r * e}, {&w2.rlam, Bn * m_r * e},
                    {&w2.rZin, Bn * n * e}, {&w2.rS0, Bn * nx * e}, {&w2.rZout, Bn * n * e},
                    {&w2.robj, (size_t)obj_offsets(H, nx, nu).total * e}};
                for (auto& x : alr) NEMPC_HIP(hipMalloc(x.p, x.bytes));
                NEMPC_HIP(hipMemsetAsync(w2.rlam, 0, Bn * m_r * e, s));     // (box rows of the handle keep zero multipliers)
                // index tables.  x-slot k of s_tau is x_{tau-k} (index into [x0 ; states]), u-slot k is u_{tau-1-k}
                const int wx = h.w * nx_r;
                auto data_x = [&](int tau, int c) { return tau == 0 ? c : nx_r + (back + tau) * nx_r + c; };           // tau <= 0
                auto data_u = [&](int tau, int c) { return nx_r + back * nx_r + (back + tau) * nu + c; };               // tau < 0
                std::vector<int> src(n), src0(nx), prim(n_r, -1), isprim(n, 0), dcol(nin, -1), shift(nx, -1);
                auto slot_source = [&](int tau_s, int i) {     // entry i of s_{tau_s}: caller variable or -(1 + data index)
                    if (i < wx) {
                        const int tau = tau_s - i / nx_r, c = i % nx_r;
                        return tau >= 1 ? (tau - 1) * nx_r + c : -(1 + data_x(tau, c));
                    }
                    const int k = (i - wx) / nu, c = (i - wx) % nu, tau = tau_s - 1 - k;
                    return tau >= 0 ? H * nx_r + tau * nu + c : -(1 + data_u(tau, c));
                };
                for (int i = 0; i < nx; ++i) src0[i] = slot_source(0, i);
                for (int t = 0; t < H; ++t) {
                    for (int i = 0; i < nx; ++i) {
                        src[t * nx + i] = slot_source(t + 1, i);
                        if (i < nx_r) { isprim[t * nx + i] = 1; prim[t * nx_r + i] = t * nx + i; }
                    }
                    for (int c = 0; c < nu; ++c) {
                        const int k = H * nx + t * nu + c;
                        src[k] = H * nx_r + t * nu + c; isprim[k] = 1; prim[H * nx_r + t * nu + c] = k;
                    }
                }
                // window column d of step t (network input order, window_var in nempc_api.hip) -> column of (s_t, u_t)
                for (int d = 0; d < nin_r; ++d) {
                    int col;
                    if (d < wx) {
                        const int j = d / nx_r, c = d % nx_r;
                        col = (h.rev ? j : back - j) * nx_r + c;                 // x_{t - k}: x-slot k of s_t
                    } else {
                        const int j = (d - wx) / nu, c = (d - wx) % nu, k = h.rev ? j : back - j;   // u_{t - k}
                        col = k == 0 ? nx + c : wx + (k - 1) * nu + c;
                    }
                    dcol[col] = d;
                }
                // shift rows of s_{t+1}: x-slot k >= 1 copies x-slot k-1 of s_t; u-slot 0 copies u_t, u-slot k copies u-slot k-1
                for (int i = nx_r; i < nx; ++i) {
                    if (i < wx) shift[i] = i - nx_r;
                    else { const int k = (i - wx) / nu, c = (i - wx) % nu; shift[i] = k == 0 ? nx + c : wx + (k - 1) * nu + c; }
                }
                struct { int** p; const std::vector<int>* v; } tb[] = {{&w2.rt.src, &src}, {&w2.rt.src0, &src0}, {&w2.rt.prim, &prim},
                                                                       {&w2.rt.isprim, &isprim}, {&w2.rt.dcol, &dcol}, {&w2.rt.shift, &shift}};
                for (auto& x : tb) {
                    NEMPC_HIP(hipMalloc((void**)x.p, x.v->size() * sizeof(int)));
                    NEMPC_HIP(hipMemcpy(*x.p, x.v->data(), x.v->size() * sizeof(int), hipMemcpyHostToDevice));
                }
                w2.prim_host = prim;
            }
            w2.cap = B;
            w2.m = m; w2.n = n;
            w2.ex_per = ex_per;
        }
    }
    SolverWs& ws = *static_cast<SolverWs*>(h.solver_ws);
    // bounds -> device (dtype T); +-inf become +-max so that comparisons stay exact.  Box ROWS on the states
    // (nempc_set_box_rows; a Constraint's rows in the reference's glue, optimizer/ipopt.py:44-52) are bounds on the
    // state variables for this solver: intersected here, what controller.py:101-105 of this package does on the host.
    const T big = std::numeric_limits<T>::max();
    std::vector<T> hl(n, -big), hu(n, big);      // (rolling: the copies inside the window state carry no bounds of their own)
    bool has_bounds = false;
    for (int i = 0; i < n_r; ++i) {
        double lo = lb ? lb[i] : -INFINITY, hi = ub ? ub[i] : INFINITY;
        if (h.box && i < H * nx_r) {
            lo = std::max(lo, h.box_lo[i % nx_r]);
            hi = std::min(hi, h.box_hi[i % nx_r]);
        }
        if (lo > hi) { set_error("nempc_solve: lb > ub"); return NEMPC_EINVAL; }
        if (lo == hi) {
            set_error("nempc_solve: lb == ub (a fixed variable has no interior for the barrier); eliminate it from the problem");
            return NEMPC_EINVAL;
        }
        const int k = rolling ? ws.prim_host[i] : i;
        hl[k] = std::isfinite(lo) ? (T)lo : -big;
        hu[k] = std::isfinite(hi) ? (T)hi : big;
        has_bounds = has_bounds || std::isfinite(lo) || std::isfinite(hi);
    }
    // (an MPC loop solves with the same bounds every step: they are uploaded when they change)
    {
        const size_t nb = (size_t)n * sizeof(T);
        if (ws.bounds_host.size() != 2 * nb || memcmp(ws.bounds_host.data(), hl.data(), nb) != 0 ||
            memcmp(ws.bounds_host.data() + nb, hu.data(), nb) != 0) {
            NEMPC_HIP(hipMemcpyAsync(ws.lb, hl.data(), nb, hipMemcpyHostToDevice, s));
            NEMPC_HIP(hipMemcpyAsync(ws.ub, hu.data(), nb, hipMemcpyHostToDevice, s));
            NEMPC_HIP(hipStreamSynchronize(s));   // hl / hu are stack-owned
            ws.bounds_host.resize(2 * nb);
            memcpy(ws.bounds_host.data(), hl.data(), nb);
            memcpy(ws.bounds_host.data() + nb, hu.data(), nb);
        }
    }

    // ---- working copies in buffer set `cur`; the caller's Z / status are written once, at the end, in the caller's order
    int cur = 0;
    ExtraBindingGuard extra_guard(h);
    if (ex_per) {
        NEMPC_HIP(hipMemcpyAsync(ws.exc[0], h.d_extra, (size_t)B * ex_per * sizeof(T), hipMemcpyDeviceToDevice, s));
        h.d_extra = ws.exc[0];
        h.extra_B = B;
    }
    const unsigned gBn = (unsigned)(((size_t)B * std::max(n, std::max(m, nx)) + 255) / 256);
    const void* Zstart = Z;
    const void* X0start = X0;
    const void* obj_dev = h.d_obj;
    if (rolling) {
        // the window state's start (augmented guess, s_0) and the objective table over the augmented stage: the caller's
        // weights on the primary block of s, zeros on the copies
        hipLaunchKernelGGL(roll_build_kernel<T>, dim3(gBn), dim3(256), 0, s, B, n, nx, n_r, nx_r, nu, back, ws.rt, (const T*)Z,
                           (const T*)X0, (const T*)h.d_hist_x, (const T*)h.d_hist_u, (const T*)ws.lb, (const T*)ws.ub,
                           (T)o.mu_init, has_bounds ? 1 : 0, (T*)ws.rZin, (T*)ws.rS0);
        Zstart = ws.rZin; X0start = ws.rS0;
        const ObjOffsets ao = obj_offsets(H, nx, nu);
        const ObjHost& oh = h.obj_host;
        const std::vector<double>& QT = h.obj_QT.empty() ? oh.Q : h.obj_QT;
        std::vector<T> tab((size_t)ao.total, T(0));
        for (int i = 0; i < nx_r; ++i)
            for (int j = 0; j < nx_r; ++j) {
                tab[ao.Q + i * nx + j] = (T)oh.Q[i * nx_r + j];
                tab[ao.Qs + i * nx + j] = (T)(oh.Q[i * nx_r + j] + oh.Q[j * nx_r + i]);
                tab[ao.QT + i * nx + j] = (T)QT[i * nx_r + j];
                tab[ao.QTs + i * nx + j] = (T)(QT[i * nx_r + j] + QT[j * nx_r + i]);
            }
        for (int i = 0; i < nu; ++i)
            for (int j = 0; j < nu; ++j) {
                tab[ao.R + i * nu + j] = (T)oh.R[i * nu + j];
                tab[ao.Rs + i * nu + j] = (T)(oh.R[i * nu + j] + oh.R[j * nu + i]);
            }
        for (int t = 0; t < H; ++t) {
            for (int i = 0; i < nx_r; ++i) {
                tab[ao.xref + t * nx + i] = (T)oh.xref[t * nx_r + i];
                tab[ao.cx + t * nx + i] = (T)oh.cx[t * nx_r + i];
            }
            for (int i = 0; i < nu; ++i) {
                tab[ao.uref + t * nu + i] = (T)oh.uref[t * nu + i];
                tab[ao.cu + t * nu + i] = (T)oh.cu[t * nu + i];
            }
        }
        const size_t tb = tab.size() * sizeof(T);
        if (ws.robj_host.size() != tb || memcmp(ws.robj_host.data(), tab.data(), tb) != 0) {
            NEMPC_HIP(hipMemcpyAsync(ws.robj, tab.data(), tb, hipMemcpyHostToDevice, s));
            NEMPC_HIP(hipStreamSynchronize(s));   // tab is stack-owned
            ws.robj_host.assign((const unsigned char*)tab.data(), (const unsigned char*)tab.data() + tb);
        }
        obj_dev = ws.robj;
    }
    hipLaunchKernelGGL(solver_init_kernel<T>, dim3(gBn), dim3(256), 0, s, B, n, (T*)ws.Zc[0], (const T*)ws.lb,
                       (const T*)ws.ub, (T*)ws.muc[0], (T*)ws.nuc[0], (T*)ws.regc[0], ws.stc[0], ws.orig[0], ws.itc[0],
                       (T*)ws.infoc[0], (T*)ws.zlc[0], (T*)ws.zuc[0], (T)o.mu_init, (T)o.reg, has_bounds ? 1 : 0,
                       (const T*)Zstart, (const T*)X0start, (T*)ws.X0c[0], nx, (T*)ws.lamc[0], m);

    SolverArgs a{};
    a.H = H; a.nx = nx; a.nu = nu; a.nin = nin; a.n = n; a.m = m;
    a.grad = ws.grad; a.g = ws.g; a.tiles = ws.tiles;
    a.hblk = ws.hblk; a.lamn = ws.lamn;
    a.obj = obj_dev; a.oo = obj_offsets(H, nx, nu);
    a.lb = ws.lb; a.ub = ws.ub; a.alpha = ws.alpha; a.phi0 = ws.phi0;
    a.dir = ws.dir; a.lsdone = ws.lsdone; a.n_active = ws.n_active; a.n_pending = ws.n_active + 1; a.n_done = ws.n_active + 2;
    a.dz = ws.dz; a.Kst = ws.Kst;
    a.dzl = ws.dzl; a.dzu = ws.dzu; a.alz = ws.alz; a.bh = ws.bh;
    a.primal_dual = (has_bounds && o.barrier != 1) ? 1 : 0; a.kst = ws.kst; a.Pst = ws.Pst; a.pst = ws.pst;
    a.tmp = ws.tmp; a.tmp_stride = (size_t)B;
    auto point_at = [&](int k) {
        a.Z = ws.Zc[k]; a.lam = ws.lamc[k]; a.mu = ws.muc[k]; a.pen = ws.nuc[k]; a.reg = ws.regc[k];
        a.status = ws.stc[k]; a.iters_done = ws.itc[k]; a.info = ws.infoc[k]; a.zl = ws.zlc[k]; a.zu = ws.zuc[k];
        if (ex_per) h.d_extra = ws.exc[k];
    };
    point_at(0);
    static const int lq_attempts_env = [] { const char* e = getenv("NEMPC_LQ_ATTEMPTS"); return e ? atoi(e) : 0; }();   // A/B knob
    a.lq_attempts = o.lq_attempts > 0 ? o.lq_attempts : (lq_attempts_env > 0 ? lq_attempts_env : 3);
    const bool wave_wanted = o.lq_kernel != 1 && (o.lq_kernel == 2 || nx * (nx + nu) >= 12);
    // thread-per-problem sweep: up to three damping levels side by side (lanes and LDS regions per problem)
    const int lq_spec_env = [] { const char* e = getenv("NEMPC_LQ_SPEC"); return e ? atoi(e) : 0; }();   // A/B knob (tests)
    a.spec = wave_wanted ? 1 : std::max(1, std::min(std::min(a.lq_attempts, lq_spec_env > 0 ? lq_spec_env : 3), 8));
    a.att_elems = H * nu * nx + H * nu + H * nx * nx + H * nx + lq_tmp_elems(nx, nu);
    // parallel-in-time solve (lq_scan_problem): 2/1 stages in double, a stage per lane
    static const int lq_scan_env = [] { const char* e = getenv("NEMPC_LQ_SCAN"); return e ? atoi(e) : 1; }();   // A/B knob
    const bool scan_fits = sizeof(T) == 8 && nx == 2 && nu == 1 && H + 1 <= 64;
    if (o.lq_kernel == 3 && !scan_fits) {
        set_error("nempc_solve: lq_kernel = 3 (scan) is built for 2-state / 1-control stages in fp64 with H <= 63");
        return NEMPC_EUNSUPPORTED;
    }
    const bool lq_scan = scan_fits && (o.lq_kernel == 3 || (o.lq_kernel == 0 && lq_scan_env != 0));
    if (lq_scan) {
        a.spec = std::max(1, std::min(a.lq_attempts, 64 / (H + 1)));     // levels side by side: lane segments of H + 1
        a.att_elems = 0;                                                    // (no per-attempt region: the scan lives in registers)
    }
    int per_problem;
    // LDS of a workgroup: the problems' blocks + one copy of the bounds (2 n) behind them
    const size_t wg_extra = (size_t)2 * n * sizeof(T);
    const size_t lds_budget = (size_t)150 * 1024 > wg_extra ? (size_t)150 * 1024 - wg_extra : 0;
    for (;; --a.spec) {
        per_problem = 2 * n + 2 * H * nx + H * nx * nin + H * nin * nin + n + n + H * nx + 2 * n + a.spec * a.att_elems;
        per_problem |= 1;   // odd stride: the ppw lanes of a sweep hit different LDS banks
        if (a.spec == 1 || (size_t)per_problem * sizeof(T) <= lds_budget) break;   // (levels one after the other if not)
    }
    auto pick_ppw = [&](int Bact) {
        int ppw = (int)(lds_budget / ((size_t)per_problem * sizeof(T)));
        if (ppw > 16) ppw = 16;
        if (ppw * a.spec > 64) ppw = 64 / a.spec;       // the sweeping lanes are one wave
        // the sweep is one latency chain per lane whatever the number of active lanes: spread the batch over the CUs
        const int spread = (Bact + h.num_cus - 1) / h.num_cus;
        if (ppw > spread) ppw = spread < 1 ? 1 : spread;
        // the wave-per-problem kernel runs one problem per wave of a 256-thread workgroup
        if (wave_wanted && ppw > 4) ppw = 4;
        return ppw;
    };
    a.ppw = pick_ppw(B);
    a.use_lds = a.ppw >= 1;
    a.lds_stride = per_problem;
    a.tol_g = o.tol_constraint; a.tol_step = o.tol_step; a.mu_min = has_bounds ? o.mu_min : 0.0; a.mu_factor = o.mu_factor;
    {
        static const double slack_eps = [] { const char* e = getenv("NEMPC_SOLVER_SLACK_EPS"); return e ? atof(e) : 0.0; }();
        a.armijo_slack = std::max(1e-12, slack_eps * (double)std::numeric_limits<T>::epsilon());
    }
    a.max_ls = o.max_linesearch;
    {
        // NEMPC_SOLVER_NONMONO: 0 monotone Armijo test (rounds 1-3), 1 .. 4 merit values of previous iterates the test may
        // refer to (default 4).  configs[2] dims, B = 1024, converged after 40 / 60 / 80 / 160 iterations
        // (profiles/r04_solver_nonmonotone.txt): 664 / 825 / 918 / 1008 monotone, 722 / 854 / 932 / 1014 (1), 747 / 881 / 966 /
        // 1020 (2), 766 / 906 / 974 / 1019 (3), 775 / 924 / 985 / 1021 (4); C2 dims 965 / 979 / 997 / 1015 -> 968 / 994 / 1011 / 1019
        static const int nm_env = [] { const char* e = getenv("NEMPC_SOLVER_NONMONO"); return e ? atoi(e) : 4; }();
        a.nonmono = nm_env < 0 ? 0 : (nm_env > 4 ? 4 : nm_env);
        // Piecewise-linear networks (relu, leaky_relu, relu6) have no second-order constraint violation for the relaxed test
        // to forgive; what it does there is let an iterate chatter across a kink without its step ever shrinking.  A
        // network with such a layer keeps the monotone test.
        for (int l = 0; l < h.nl; ++l)
            if (act_is_piecewise_linear(h.act[l])) a.nonmono = 0;
    }
    {
        // The Levenberg term moves in half decades (round 5).  In whole decades -- x 10 when a sweep meets a pivot that is not
        // positive, x 0.1 after a clean one -- a problem whose reduced Hessian needs a term of, say, 2 alternates between 1
        // (fails, retried) and 10: it is damped five times harder than it has to be, and the slow problems of configs[2]'s dims
        // crawled through a dozen iterations of 4e-2 steps that way (tools/c3_slow_trace.py).  sqrt(10) each way: converged
        // after 40 / 60 / 80 iterations 994 / 1024 / 1024 of 1024 instead of 847 / 990 / 1021, the last problem through in
        // 81 ms instead of 123; C2 dims unchanged (968 / 991 / 1010 / 1021 against 968 / 993 / 1010 / 1019);
        // profiles/r05_solver_reg_steps.txt.  NEMPC_SOLVER_REG_RAISE / _RELAX are the A/B knobs.
        static const double relax_env = [] { const char* e = getenv("NEMPC_SOLVER_REG_RELAX"); return e ? atof(e) : 0.31622776601683794; }();
        a.reg_relax = relax_env > 0.0 && relax_env < 1.0 ? relax_env : 0.31622776601683794;
        static const double raise_env = [] { const char* e = getenv("NEMPC_SOLVER_REG_RAISE"); return e ? atof(e) : 3.1622776601683795; }();
        a.reg_raise = raise_env > 1.0 && raise_env <= 100.0 ? raise_env : 3.1622776601683795;
    }
    auto lqk = a.use_lds ? ((nx == 2 && nu == 1) ? solver_lq_kernel<T, 2, 1, true>
                                                  : ((nx == 6 && nu == 3) ? solver_lq_kernel<T, 6, 3, true> : solver_lq_kernel<T, 0, 0, true>))
                         : ((nx == 2 && nu == 1) ? solver_lq_kernel<T, 2, 1, false>
                                                  : ((nx == 6 && nu == 3) ? solver_lq_kernel<T, 6, 3, false> : solver_lq_kernel<T, 0, 0, false>));
    // wave-per-problem sweep when the working set is staged in LDS and a stage has enough entries to spread over a
    // wave: 6/3 stages 6.5 k vs 5.0 k MPC solves/s at C3; 2/1 stages are 4 entries wide and stay on the
    // thread-per-problem kernel (22.0 vs 23.1 ms per 40 iterations at C2).  nempc_solver_opts.lq_kernel forces one.
    const bool lq_wave = a.use_lds && wave_wanted;
    const int no_fuse_step = [] { const char* e = getenv("NEMPC_SOLVER_NO_FUSE_STEP"); return e ? atoi(e) : 0; }();   // (tests)
    a.fuse_step = a.use_lds && !lq_wave && !no_fuse_step;
    a.f_it = ws.f; a.Zt_it = ws.Zt;
    if (a.fuse_step) NEMPC_HIP(hipMemsetAsync(ws.n_active, 0, 8 * sizeof(int), s));
    if (lq_wave)
        lqk = (nx == 2 && nu == 1) ? solver_lqw_kernel<T, 2, 1>
                                   : ((nx == 6 && nu == 3) ? solver_lqw_kernel<T, 6, 3> : solver_lqw_kernel<T, 0, 0>);
    if constexpr (sizeof(T) == 8)
        if (lq_scan && a.use_lds && !lq_wave) lqk = solver_lq_kernel<T, 2, 1, true, true>;
    {
        const int ppw_max = (int)(lds_budget / ((size_t)per_problem * sizeof(T)));
        const size_t lds_max = a.use_lds ? (size_t)std::min(std::max(ppw_max, 1), 16) * per_problem * sizeof(T) + wg_extra : 0;
        NEMPC_HIP(ensure_dynamic_lds(reinterpret_cast<const void*>(lqk), lds_max));
    }

    const int no_carry = [] { const char* e = getenv("NEMPC_SOLVER_NO_CARRY"); return e ? atoi(e) : 0; }();   // A/B knob (tests)
    const int lsm_carry = o.linesearch == 0 ? (wave_wanted ? 1 : 2) : o.linesearch;
    bool carry = lsm_carry == 2 && !no_carry && a.use_lds && h.variant == NEMPC_KERNEL_MFMA && h.cfg.integrator != NEMPC_RK4 &&
                 !rolling;
    int pend_seq = 0;             // sequence number of the inner loop's acceptance launches (published with their count)
    bool have_eval = false;       // the evaluation buffers hold every active problem's current iterate
    bool have_blocks = false;     // ... and so does the block buffer
    // the trial point's Lagrangian blocks from the launch that evaluates it (compiled shape): an iteration is then the LQ
    // solve, ONE callback launch and the acceptance test
    const int hess_trial_env = [] { const char* e = getenv("NEMPC_SOLVER_HESS_TRIAL"); return e ? atoi(e) : 1; }();   // A/B knob (tests)
    bool hess_trial = carry && a.fuse_step && hess_trial_env != 0;
    a.carry = 0; a.tiles_t = ws.tiles_t; a.grad_t = ws.grad_t; a.lam_t = nullptr; a.hblk_t = nullptr;
    // the acceptance test inside the next iteration's LQ kernel (see SolverArgs::accept_first): two launches per iteration
    const int fuse_accept_env = [] { const char* e = getenv("NEMPC_SOLVER_FUSE_ACCEPT"); return e ? atoi(e) : 1; }();   // A/B knob (tests)
    const bool fuse_accept = fuse_accept_env != 0;
    static const bool stats_on = getenv("NEMPC_SOLVER_STATS") != nullptr;
    bool accept_pending = false;       // the last trial point has been evaluated, its acceptance test has not been launched
    a.accept_first = 0; a.n_active_prev = ws.n_active; a.gt_acc = ws.gt;
    a.hpub = ws.hpub_dev;
    for (int k = 0; k < 8; ++k) ws.hpub[k] = 0;       // (the previous solve on this handle ended with a synchronised stream)
    const bool published_polls = !wave_wanted;  // small stages (chains of latency-bound launches): no blocking polls, the host
                                                // reads the counter the device publishes
    int Bact = B;                 // slots [0, Bact) may still be unconverged; compaction keeps them in front
    int last_nact = B;            // unconverged problems at the last convergence poll
    int it = 0, rc;
    const int check = o.check_every > 0 ? o.check_every : 4;
    static const int trace_slot = [] { const char* e = getenv("NEMPC_SOLVER_TRACE"); return e ? atoi(e) : -1; }();
    const bool trace = trace_slot >= 0 && trace_slot < B;
    // rolling: the callbacks read the caller's x0 / history by problem index, so the batch keeps its order (a finished
    // problem is skipped by every solver kernel; its callback rows are evaluated and ignored)
    const bool compact = o.compact != 0 && !trace && !rolling;
    // ---- callbacks.  Plain models: the handle's launches on the solver's buffers.  Rolling: gather the caller's
    //      variables, launch on them, spread the results into the window-state form (RollTables)
    auto launch_rows = [&](int nb, const void* Zs, const void* X0s, void* g, void* tiles, void* tiles_scratch) -> int {
        return h.variant != NEMPC_KERNEL_VALU ? launch_rows_mfma(h, nb, Zs, X0s, g, tiles, s)
                                              : launch_rows_valu(h, nb, Zs, X0s, g, tiles ? tiles : tiles_scratch, s);
    };
    auto launch_blocks = [&](int nb, const void* Zs, const void* X0s, const void* lam, void* blocks) -> int {
        return h.variant != NEMPC_KERNEL_VALU ? launch_rowhess_mfma(h, nb, Zs, X0s, lam, blocks, s)
                                              : launch_rowhess_valu(h, nb, Zs, X0s, lam, blocks, s);
    };
    const unsigned gRn = (unsigned)(((size_t)B * n_r + 255) / 256);
    // everything at an iterate: defects, tiles, f, gradient, Lagrangian blocks
    auto roll_eval_iterate = [&](const void* Zaug, const void* lam_aug) -> int {
        hipLaunchKernelGGL(roll_in_kernel<T>, dim3(gRn), dim3(256), 0, s, B, n, n_r, H, nx_r, nx, m_r, ws.rt, (const T*)Zaug,
                           (T*)ws.rZ, (const T*)lam_aug, (T*)ws.rlam);
        int r = launch_rows(B, ws.rZ, X0, ws.rg, ws.rtiles, nullptr);
        if (r) return r;
        if ((r = launch_blocks(B, ws.rZ, X0, ws.rlam, ws.rhblk))) return r;
        if ((r = launch_objective(h, B, ws.rZ, ws.f, ws.rgrad, s))) return r;
        hipLaunchKernelGGL(roll_out_kernel<T>, dim3(B * H), dim3(256), 0, s, H, nx_r, nu, nx, nin_r, n_r, m_r, ws.rt,
                           (const T*)Zaug, (const T*)ws.rS0, (const T*)ws.rg, (const T*)ws.rtiles, (const T*)ws.rhblk,
                           (const T*)ws.rgrad, (T*)ws.g, (T*)ws.tiles, (T*)ws.hblk, (T*)ws.grad);
        return NEMPC_OK;
    };
    // a trial point: defects only (the acceptance kernel evaluates the objective from the augmented table itself)
    auto roll_eval_trial = [&](const void* Zaug) -> int {
        hipLaunchKernelGGL(roll_in_kernel<T>, dim3(gRn), dim3(256), 0, s, B, n, n_r, H, nx_r, nx, m_r, ws.rt, (const T*)Zaug,
                           (T*)ws.rZ, (const T*)nullptr, (T*)nullptr);
        const int r = launch_rows(B, ws.rZ, X0, ws.rg, nullptr, h.d_tiles_ws);
        if (r) return r;
        hipLaunchKernelGGL(roll_out_kernel<T>, dim3(B * H), dim3(256), 0, s, H, nx_r, nu, nx, nin_r, n_r, m_r, ws.rt,
                           (const T*)Zaug, (const T*)ws.rS0, (const T*)ws.rg, (const T*)nullptr, (const T*)nullptr,
                           (const T*)nullptr, (T*)ws.gt, (T*)nullptr, (T*)nullptr, (T*)nullptr);
        return NEMPC_OK;
    };
    // (A block-wise mirrored convexification of the stage Hessians, for the problems whose sweep needed damping, was built and
    // measured in round 4 -- profiles/r04_solver_convexify_c3.txt: +3 - 5 % problems converged at 1.5 ms per iteration; removed
    // in round 5: indefiniteness is not what holds the configs[2] solves back, DESIGN "Batched solver".)
    const bool soc_env = [] { const char* e = getenv("NEMPC_SOLVER_SOC"); return !(e && atoi(e) == 0); }();      // (read per solve)
    for (; it < o.max_iter; ++it) {
        a.B = Bact;
        a.cur_it = it;
        const size_t lds_need = a.use_lds ? (size_t)a.ppw * per_problem * sizeof(T) + wg_extra : 0;
        const unsigned gAn = (unsigned)(((size_t)Bact * n + 255) / 256);
        void* Zc = ws.Zc[cur];
        const void* X0c = ws.X0c[cur];
        // callbacks at the iterate: defects + tiles (row kernel), f + grad (objective kernel)
        // and the per-step Lagrangian blocks with the current multipliers (all zero on the first iterate:
        // Gauss-Newton step).  The RK4 matrix-core pipeline produces defects, tiles and blocks from one row launch.
        const bool rk4_pipeline = h.variant != NEMPC_KERNEL_VALU && h.cfg.integrator == NEMPC_RK4;
        bool fused_eval = false;
        if (rolling) {
            fused_eval = true;                  // (f and grad come with it)
            rc = roll_eval_iterate(Zc, ws.lamc[cur]);
        } else if (have_eval) {
            // deferred backtracking on a compiled shape: the trial evaluation of the last iteration was a full one and the
            // acceptance kernel kept, per problem, the evaluation of the point it stands on -- only the blocks are new, and
            // not even those when the trial launch computed them
            fused_eval = true;
            rc = have_blocks ? NEMPC_OK
                             : (h.variant != NEMPC_KERNEL_VALU ? launch_rowhess_mfma(h, Bact, Zc, X0c, ws.lamc[cur], ws.hblk, s)
                                                               : launch_rowhess_valu(h, Bact, Zc, X0c, ws.lamc[cur], ws.hblk, s));
        } else if (rk4_pipeline) {
            rc = launch_rowhess_rk4_mfma(h, Bact, Zc, X0c, ws.lamc[cur], ws.hblk, s, ws.g, ws.tiles);
            if (rc == NEMPC_EUNSUPPORTED) {
                // (a shape whose stage records would come from a wave-per-tile instantiation that is not used: rows through
                //  the matrix-core entry point -- it falls back by itself -- and the blocks from the generic kernel)
                if ((rc = launch_rows_mfma(h, Bact, Zc, X0c, ws.g, ws.tiles, s))) return rc;
                rc = launch_rowhess_valu(h, Bact, Zc, X0c, ws.lamc[cur], ws.hblk, s);
            }
        } else if (carry && hess_trial &&
                   (rc = launch_rowhess_eval_mfma(h, Bact, Zc, X0c, ws.lamc[cur], ws.hblk, ws.g, ws.tiles, s)) != NEMPC_EUNSUPPORTED) {
            // first iterate (and the one after a compaction) through the kernel the trial points go through -- the same
            // arithmetic whichever way an iterate's evaluation was obtained: blocks, defects, tiles; f and grad below
        } else {
            // compiled shapes: defects, tiles, f and grad from one launch
            fused_eval = h.variant == NEMPC_KERNEL_MFMA &&
                         (rc = launch_eval_fused(h, Bact, Zc, X0c, ws.g, ws.tiles, nullptr, ws.f, ws.grad, s)) != NEMPC_EUNSUPPORTED;
            if (!fused_eval)
                rc = h.variant != NEMPC_KERNEL_VALU ? launch_rows_mfma(h, Bact, Zc, X0c, ws.g, ws.tiles, s)
                                                    : launch_rows_valu(h, Bact, Zc, X0c, ws.g, ws.tiles, s);
            if (rc) return rc;
            rc = h.variant != NEMPC_KERNEL_VALU ? launch_rowhess_mfma(h, Bact, Zc, X0c, ws.lamc[cur], ws.hblk, s)
                                                : launch_rowhess_valu(h, Bact, Zc, X0c, ws.lamc[cur], ws.hblk, s);
        }
        if (rc) return rc;
        if (!fused_eval && (rc = launch_objective(h, Bact, Zc, ws.f, ws.grad, s))) return rc;
        // bounds: barrier diagonal for the LQ model, barrier gradient folded into grad
        if (!a.use_lds) hipLaunchKernelGGL(solver_barrier_kernel<T>, dim3(gAn), dim3(256), 0, s, a);
        a.lam_t = carry && hess_trial ? ws.lam_t : nullptr;
        // this iteration's convergence counter; the pending acceptance (if any) publishes and clears the other word
        a.n_active = fuse_accept && a.fuse_step ? ws.n_active + (it & 1) * 4 : ws.n_active;
        a.n_active_prev = ws.n_active + ((it & 1) ^ 1) * 4;
        a.accept_first = accept_pending ? 1 : 0;
        accept_pending = false;
        // LDS mode: four waves stage the working set, the first ppw lanes run the sweeps
        hipLaunchKernelGGL(lqk, dim3(a.use_lds ? (Bact + a.ppw - 1) / a.ppw : (Bact + 63) / 64), dim3(a.use_lds ? 256 : 64),
                           lds_need, s, a);
#ifdef NEMPC_LQ_STAMPS
        if (it == 3) {
            long long st[16];
            hipStreamSynchronize(s);
            hipMemcpyFromSymbol(st, HIP_SYMBOL(nempc_lq_stamps), sizeof(st));
            fprintf(stderr, "lq stamps (cycles since start):");
            for (int i = 1; i <= 9; ++i) fprintf(stderr, " %lld", st[i] - st[0]);
            fprintf(stderr, "\n");
        }
#endif
        if (!a.fuse_step)
            hipLaunchKernelGGL(solver_step_kernel<T>, dim3(Bact), dim3(64), 0, s, a, (const T*)ws.f, (const T*)Zc, (T*)ws.Zt);
        // Convergence poll.  Matrix-core-bound stages: a blocking copy every `check` iterations (an iteration is milliseconds,
        // a drained stream costs nothing next to iterations at a stale batch size).  Small stages: see after the
        // backtracking launches below.
        bool polled = false;
        int nact = Bact;
        if (!published_polls && ((it + 1) % check == 0 || it + 1 == o.max_iter)) {
            NEMPC_HIP(hipMemcpyAsync(ws.hpoll, ws.n_active, sizeof(int), hipMemcpyDeviceToHost, s));
            NEMPC_HIP(hipStreamSynchronize(s));
            nact = ws.hpoll[0];
            polled = true;
            last_nact = nact;
            // (report the iteration at which the last problem converged, as the published counter has it, not the poll's)
            if (nact == 0) { it = ws.hpub[2] > 0 ? ws.hpub[2] : it + 1; break; }
        }
        // Backtracking in a lock-step batch.  An inner loop makes every problem pay for the one that needs six halvings
        // (measured: 5.7 trial evaluations per iteration at B=1024, C2 dims, 70 % of the solve time).  DEFERRED (2): one
        // trial per iteration; a problem whose trial is rejected stands still and retries the same direction at half the
        // length next iteration.  ADAPTIVE (3): after the first trial, if fewer than a quarter of the active problems
        // are still searching they are deferred, otherwise the loop goes on.  auto: matrix-core-bound iterations (6/3
        // 3x128: a trial is 5 % of an iteration) keep the inner loop; small stages defer -- at equal wall time deferral
        // converges more problems at every budget measured (2/1 2x64, B=1024: 1014 converged in 24.8 ms against 1004 in
        // 24.9 ms), in the wide phase because a trial evaluation costs real time and among the stragglers because each
        // extra trial is a latency-bound launch chain plus a host poll.  It spends more ITERATIONS on a hard problem
        // (a retry is an iteration), so max_iter budgets are larger than with the inner loop.
        // (rolling: one trial per iteration whatever was asked for -- the dense trial lists of the inner loop would need the
        //  callbacks' x0 / history gathered per list)
        const int lsm = rolling ? 2 : (o.linesearch == 0 ? (wave_wanted ? 1 : 2) : o.linesearch);
        // second-order correction of rejected full steps: inner-loop backtracking only (the deferred form accepts inside the
        // next LQ kernel); NEMPC_SOLVER_SOC=0 switches it off (A/B, tests)
        const bool soc_on = lsm != 2 && !rolling && soc_env;
        int pending = 0;
        int pflip = 0;          // which of the two lists of searching problems is read next (the correction stage swaps them)
        const void* const extra_all = h.d_extra;
        // the acceptance kernel publishes the number of problems still searching under a sequence number: wait for it
        auto poll_pending = [&](int seq, int* out) -> int {
            volatile int* hp = ws.hpub;
            const auto t0 = std::chrono::steady_clock::now();
            long spins = 0;
            while (hp[4] != seq) {
                cpu_relax();
                if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                    NEMPC_HIP(hipStreamSynchronize(s));      // a device fault surfaces here instead of a hang
                    if (hp[4] != seq) {
                        set_error("nempc_solve: iteration " + std::to_string(it) + ": the backtracking counter of trial " +
                                  std::to_string(seq) + " was never published (stream drained, word still " +
                                  std::to_string(hp[4]) + ")");
                        return NEMPC_EHIP;
                    }
                }
            }
            *out = hp[5];
            return NEMPC_OK;
        };
        for (int ls = 0; ls < o.max_linesearch; ++ls) {
            // the first trial point comes from solver_step_kernel (or the Riccati kernel), for every active problem; later
            // ones (inner-loop backtracking) are built here for the problems still searching ONLY, densely: their
            // evaluation and acceptance test run over `pending` problems instead of the whole active batch
            const int nb = ls > 0 ? pending : Bact;
            const void* X0t = X0c;
            if (ls > 0) {
                hipLaunchKernelGGL(solver_trial_list_kernel<T>, dim3(nb), dim3(64), 0, s, n, nx, (int)ex_per,
                                   (const int*)ws.pend[(ls + pflip) & 1], (const T*)Zc, (const T*)ws.dz, (const T*)ws.alpha, (const T*)X0c,
                                   (const T*)extra_all, (T*)ws.Zt, (T*)ws.X0p, (T*)ws.exp_, a.n_pending);
                X0t = ws.X0p;
                if (ex_per) h.d_extra = ws.exp_;
            }
            // the merit function needs the defects only: the matrix-core kernel skips its reverse sweeps (tiles = null)
            // (compiled shapes: defects and f of the trial point from one forward-only launch)
            bool fused_trial = false, trial_done = false;
            a.carry = 0; a.hblk_t = nullptr;
            if (carry && hess_trial) {
                // blocks, defects and tiles of the trial point from one launch (the acceptance kernel evaluates the objective)
                rc = launch_rowhess_eval_mfma(h, nb, ws.Zt, X0t, ws.lam_t, ws.hblk_t, ws.gt, ws.tiles_t, s);
                if (rc == NEMPC_EUNSUPPORTED) hess_trial = false;
                else { trial_done = true; a.carry = 1; a.hblk_t = ws.hblk_t; }
            }
            if (carry && !trial_done) {
                // full evaluation of the trial point (tiles and gradient too): it is the next iterate's if accepted
                fused_trial = (rc = launch_eval_fused(h, nb, ws.Zt, X0t, ws.gt, ws.tiles_t, nullptr, ws.ft, ws.grad_t, s)) !=
                              NEMPC_EUNSUPPORTED;
                if (!fused_trial) carry = false;      // not a compiled shape
                else a.carry = 1;
            }
            if (trial_done) {
            } else if (rolling) {
                rc = roll_eval_trial(ws.Zt);
            } else {
                if (!fused_trial)
                    fused_trial = h.variant == NEMPC_KERNEL_MFMA &&
                        (rc = launch_eval_fused(h, nb, ws.Zt, X0t, ws.gt, nullptr, nullptr, ws.ft, nullptr, s)) != NEMPC_EUNSUPPORTED;
                if (!fused_trial) rc = launch_rows(nb, ws.Zt, X0t, ws.gt, nullptr, h.d_tiles_ws);
            }
            h.d_extra = extra_all;
            if (rc) return rc;
            // (no fused evaluation: the acceptance kernel computes the trial point's objective value itself)
            // Deferred backtracking with the trial point's blocks in hand: the test runs at the start of the next LQ kernel,
            // unless this is the last iteration of the budget or somebody reads the iterate before that kernel (diagnostics;
            // a compaction launches it below)
            accept_pending = fuse_accept && a.fuse_step && trial_done && lsm == 2 && published_polls && it + 1 < o.max_iter && !trace &&
                             !stats_on;
            if (!accept_pending)
                hipLaunchKernelGGL(solver_merit_kernel<T>, dim3(nb), dim3(64), 0, s, a, (const T*)ws.Zt,
                                   (const T*)ws.gt, fused_trial ? (const T*)ws.ft : (const T*)nullptr, (T*)Zc,
                                   lsm == 2 ? 2 : (ls + 1 == o.max_linesearch ? 1 : 0),
                                   ls > 0 ? (const int*)ws.pend[(ls + pflip) & 1] : (const int*)nullptr, ws.pend[(ls + 1 + pflip) & 1],
                                   ls == 0 ? 1 : 0, lsm == 2 ? 0 : ++pend_seq);
            have_eval = a.carry != 0;
            have_blocks = have_eval && a.hblk_t != nullptr;
            if (lsm == 2) break;          // one trial per outer iteration: nothing to poll
            // most iterations accept the first trial for every problem: one small poll saves the remaining
            // max_linesearch-1 callback evaluations
            if ((rc = poll_pending(pend_seq, &pending))) return rc;
            if (pending > 0 && ls == 0 && soc_on && !a.carry) {
                // second-order correction of the rejected full steps (solver_soc_kernel): one defect-only launch over the
                // rejected problems, accepted at the rejected trial's step length; what is still rejected backtracks as before
                const int* lin = ws.pend[(1 + pflip) & 1];
                int* lout = ws.pend[pflip & 1];
                hipLaunchKernelGGL(solver_soc_kernel<T>, dim3(pending), dim3(64), 0, s, n, nx, nu, H, m, (int)ex_per, lin, (const T*)ws.Zt,
                                   (const T*)ws.gt, (const T*)ws.tiles, (const T*)X0c, (const T*)extra_all, (T*)ws.Zsoc, (T*)ws.X0p,
                                   (T*)ws.exp_, a.n_pending);
                if (ex_per) h.d_extra = ws.exp_;
                rc = launch_rows(pending, ws.Zsoc, ws.X0p, ws.gsoc, nullptr, h.d_tiles_ws);
                h.d_extra = extra_all;
                if (rc) return rc;
                hipLaunchKernelGGL(solver_merit_kernel<T>, dim3(pending), dim3(64), 0, s, a, (const T*)ws.Zsoc, (const T*)ws.gsoc,
                                   (const T*)nullptr, (T*)Zc, 0, lin, lout, 0, ++pend_seq, 1);
                if ((rc = poll_pending(pend_seq, &pending))) return rc;
                pflip ^= 1;
            }
            if (pending == 0) break;
            if (lsm == 3 && ls == 0 && pending * 4 <= std::min(Bact, last_nact)) {
                hipLaunchKernelGGL(solver_defer_kernel<T>, dim3((Bact + 255) / 256), dim3(256), 0, s, a);
                break;
            }
        }
        if (published_polls) {
            // Small stages: an iteration is a chain of latency-bound launches, and a blocking poll drains the stream (10
            // polls of ~25 us in a 5.5 ms solve).  The acceptance kernel PUBLISHES the iteration's counter to pinned host
            // memory; the host only makes sure it never runs more than two iterations ahead of the device (it would
            // otherwise queue its whole budget before the first problem converges) and takes whatever the slot holds:
            // a count a couple of iterations old is an upper bound of the unconverged problems (they only ever leave),
            // which is all compaction needs, and "none left" is reported with the iteration at which it happened.
            volatile int* hp = ws.hpub;
            const int want = it + 1 - 2;
            if (want > 0) {
                const auto t0 = std::chrono::steady_clock::now();
                long spins = 0;
                while (hp[0] < want) {
                    cpu_relax();
                    if ((++spins & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                        NEMPC_HIP(hipStreamSynchronize(s));      // a device fault surfaces here instead of a hang
                        if (hp[0] < want) {
                            set_error("nempc_solve: iteration " + std::to_string(it) + ": the convergence counter of iteration " +
                                      std::to_string(want) + " was never published (stream drained, word still " +
                                      std::to_string(hp[0]) + ")");
                            return NEMPC_EHIP;
                        }
                    }
                }
            }
            if (hp[0] > 0) {
                if (hp[2] > 0) { it = hp[2]; break; }      // every problem had converged at that iteration
                nact = hp[1];
                polled = true;
                last_nact = nact;
            }
        }
        if (stats_on) {   // NEMPC_SOLVER_STATS=1: Riccati restarts per iteration over the active slots (diagnostic, synchronises)
            std::vector<T> inf((size_t)Bact * INFO_N);
            std::vector<T> rg((size_t)Bact);
            NEMPC_HIP(hipStreamSynchronize(s));
            NEMPC_HIP(hipMemcpy(inf.data(), a.info, inf.size() * sizeof(T), hipMemcpyDeviceToHost));
            NEMPC_HIP(hipMemcpy(rg.data(), a.reg, rg.size() * sizeof(T), hipMemcpyDeviceToHost));
            double sum = 0, mx = 0, rmx = 0; int nz = 0;
            for (int i = 0; i < Bact; ++i) {
                const double r = std::fabs((double)inf[(size_t)i * INFO_N + INFO_RESTARTS]);
                sum += r; if (r > mx) mx = r; if (r > 0) ++nz; if ((double)rg[i] > rmx) rmx = (double)rg[i];
            }
            fprintf(stderr, "[solver it %3d] active slots %4d  restarts: max %.0f mean %.2f  problems restarting %d  max damping %.1e\n",
                    it, Bact, mx, sum / Bact, nz, rmx);
        }
        if (trace) {   // NEMPC_SOLVER_TRACE=<slot>: one line per iteration for that slot (diagnostic, synchronises)
            T inf[INFO_N], muv, alv, regv, penv; int st, lsd;
            NEMPC_HIP(hipStreamSynchronize(s));
            NEMPC_HIP(hipMemcpy(inf, (const T*)a.info + (size_t)trace_slot * INFO_N, sizeof(inf), hipMemcpyDeviceToHost));
            NEMPC_HIP(hipMemcpy(&muv, (const T*)a.mu + trace_slot, sizeof(T), hipMemcpyDeviceToHost));
            NEMPC_HIP(hipMemcpy(&alv, (const T*)a.alpha + trace_slot, sizeof(T), hipMemcpyDeviceToHost));
            NEMPC_HIP(hipMemcpy(&regv, (const T*)a.reg + trace_slot, sizeof(T), hipMemcpyDeviceToHost));
            NEMPC_HIP(hipMemcpy(&penv, (const T*)a.pen + trace_slot, sizeof(T), hipMemcpyDeviceToHost));
            NEMPC_HIP(hipMemcpy(&st, a.status + trace_slot, sizeof(int), hipMemcpyDeviceToHost));
            NEMPC_HIP(hipMemcpy(&lsd, a.lsdone + trace_slot, sizeof(int), hipMemcpyDeviceToHost));
            fprintf(stderr, "[solver it %3d slot %d] mu %.2e step %.2e ginf %.2e lam %.2e amax %.2e alpha %.2e reg %.1e pen %.2e restarts %.0f status %d lsdone %d\n",
                    it, trace_slot, (double)muv, (double)inf[INFO_STEP], (double)inf[INFO_GINF], (double)inf[INFO_LAM],
                    (double)inf[INFO_AMAX], (double)alv, (double)regv, (double)penv, (double)inf[INFO_RESTARTS], st, lsd);
        }
        // ---- compaction: once a quarter of the active slots has finished, gather the unconverged problems to the
        //      front of the other buffer set and shrink every launch to them
        if (compact && polled && nact < Bact - Bact / 4 && Bact > 64) {
            if (accept_pending) {       // the gather below reads the iterate: the trial's acceptance test now, in a launch of its own
                hipLaunchKernelGGL(solver_merit_kernel<T>, dim3(Bact), dim3(64), 0, s, a, (const T*)ws.Zt, (const T*)ws.gt,
                                   (const T*)nullptr, (T*)Zc, 2, (const int*)nullptr, ws.pend[1], 1, 0);
                accept_pending = false;
            }
            const int nxt = cur ^ 1;
            hipLaunchKernelGGL(solver_partition_kernel, dim3(1), dim3(1024), 0, s, Bact, (const int*)ws.stc[cur], ws.perm,
                               ws.count);
            CompactArrays ca{};
            int k = 0;
            auto add = [&](const void* src, void* dst, int per, int esz) {
                ca.src[k] = src; ca.dst[k] = dst; ca.per[k] = per; ca.esz[k] = esz; ++k;
            };
            add(ws.Zc[cur], ws.Zc[nxt], n, sizeof(T));
            add(ws.X0c[cur], ws.X0c[nxt], nx, sizeof(T));
            add(ws.lamc[cur], ws.lamc[nxt], m, sizeof(T));
            add(ws.muc[cur], ws.muc[nxt], 1, sizeof(T));
            add(ws.nuc[cur], ws.nuc[nxt], 1, sizeof(T));
            add(ws.regc[cur], ws.regc[nxt], 1, sizeof(T));
            add(ws.stc[cur], ws.stc[nxt], 1, sizeof(int));
            add(ws.orig[cur], ws.orig[nxt], 1, sizeof(int));
            add(ws.itc[cur], ws.itc[nxt], 1, sizeof(int));
            add(ws.infoc[cur], ws.infoc[nxt], INFO_N, sizeof(T));
            add(ws.zlc[cur], ws.zlc[nxt], n, sizeof(T));
            add(ws.zuc[cur], ws.zuc[nxt], n, sizeof(T));
            if (ex_per) add(ws.exc[cur], ws.exc[nxt], (int)ex_per, sizeof(T));
            ca.n = k;
            hipLaunchKernelGGL(solver_gather_kernel, dim3(Bact), dim3(256), 0, s, Bact, (const int*)ws.perm, ca);
            // slots [Bact, B) hold problems that finished before earlier compactions: carry them over unchanged
            if (Bact < B) {
                const size_t rest = (size_t)(B - Bact);
                NEMPC_HIP(hipMemcpyAsync((T*)ws.Zc[nxt] + (size_t)Bact * n, (const T*)ws.Zc[cur] + (size_t)Bact * n,
                                         rest * n * sizeof(T), hipMemcpyDeviceToDevice, s));
                NEMPC_HIP(hipMemcpyAsync(ws.stc[nxt] + Bact, ws.stc[cur] + Bact, rest * sizeof(int), hipMemcpyDeviceToDevice, s));
                NEMPC_HIP(hipMemcpyAsync(ws.orig[nxt] + Bact, ws.orig[cur] + Bact, rest * sizeof(int), hipMemcpyDeviceToDevice, s));
                NEMPC_HIP(hipMemcpyAsync(ws.itc[nxt] + Bact, ws.itc[cur] + Bact, rest * sizeof(int), hipMemcpyDeviceToDevice, s));
            }
            // the unconverged problems now sit in front.  Their exact number is on the device; the host shrinks the
            // launches to the counter it has just read -- taken a period earlier, so an upper bound (problems only ever
            // leave the active set): a few finished problems ride along in the active prefix and are skipped by every
            // kernel, and no stream synchronisation is needed to learn the exact count
            cur = nxt;
            point_at(cur);
            have_eval = false;            // (the evaluation buffers are not gathered: one launch after a compaction)
            have_blocks = false;
            if (published_polls) {
                Bact = nact > 0 ? (nact < Bact ? nact : Bact) : 1;
            } else {
                NEMPC_HIP(hipMemcpyAsync(ws.hpoll, ws.count, sizeof(int), hipMemcpyDeviceToHost, s));
                NEMPC_HIP(hipStreamSynchronize(s));
                Bact = ws.hpoll[0] > 0 ? ws.hpoll[0] : 1;
            }
            a.ppw = pick_ppw(Bact);
        }
    }
    if (accept_pending) {           // (the loop was left with a trial point evaluated and not yet tested)
        hipLaunchKernelGGL(solver_merit_kernel<T>, dim3(a.B), dim3(64), 0, s, a, (const T*)ws.Zt, (const T*)ws.gt,
                           (const T*)nullptr, (T*)ws.Zc[cur], 2, (const int*)nullptr, ws.pend[1], 1, 0);
        accept_pending = false;
    }
    hipLaunchKernelGGL(solver_scatter_kernel<T>, dim3(B), dim3(256), 0, s, B, n, (const T*)ws.Zc[cur],
                       (const int*)ws.stc[cur], (const int*)ws.orig[cur], (const int*)ws.itc[cur],
                       rolling ? (T*)ws.rZout : (T*)Z, status_dev, (int*)o.iters_out);
    if (rolling)      // the caller's variables are the primary copies
        hipLaunchKernelGGL(roll_in_kernel<T>, dim3(gRn), dim3(256), 0, s, B, n, n_r, H, nx_r, nx, m_r, ws.rt, (const T*)ws.rZout,
                           (T*)Z, (const T*)nullptr, (T*)nullptr);
    NEMPC_HIP(hipGetLastError());
    NEMPC_HIP(hipStreamSynchronize(s));
    if (iters_host) *iters_host = it;
    return NEMPC_OK;
}

// Every exit of a failed solve leaves the handle as a finished one does: kernels that still write the published words
// (hpub) and the device counters may be queued when an error return is taken from inside the iteration loop, and the
// next solve zeroes those words on the assumption that nothing is in flight.  Drain the stream, reset the counters.
template <typename T>
static int solve_typed(Handle& h, int B, const void* X0, void* Z, const double* lb, const double* ub,
                       const nempc_solver_opts& o, int32_t* status_dev, int32_t* iters_host, hipStream_t s) {
    const int rc = solve_impl<T>(h, B, X0, Z, lb, ub, o, status_dev, iters_host, s);
    if (rc != NEMPC_OK) {
        (void)hipStreamSynchronize(s);
        if (SolverWs* w = static_cast<SolverWs*>(h.solver_ws)) {
            if (w->n_active) (void)hipMemset(w->n_active, 0, 8 * sizeof(int));
            if (w->hpub) for (int k = 0; k < 8; ++k) w->hpub[k] = 0;
        }
        (void)hipGetLastError();
    }
    return rc;
}

int solver_run(Handle& h, int B, const void* X0, void* Z, const double* lb, const double* ub,
               const nempc_solver_opts& o, int32_t* status_dev, int32_t* iters_host, hipStream_t s) {
    return h.cfg.dtype == NEMPC_F64 ? solve_typed<double>(h, B, X0, Z, lb, ub, o, status_dev, iters_host, s)
                                    : solve_typed<float>(h, B, X0, Z, lb, ub, o, status_dev, iters_host, s);
}

}  // namespace nempc
