// sbm_core.hip -- libsbm_hip.so: the C ABI of include/sbm.h.
//
// Contexts, plugin loading, batched simulate / sensitivity entry points, and the
// model-independent Project kernels:
//   k_gather_params   theta -> per-experiment parameter vectors
//                     (Project.get_experiment_parameters, project/base_project.py:343-363)
//   k_assemble        fused measurement sampling + 'direct'/'sum' mapping + linear scale
//                     factors + weighted residuals + log-parameter chain rule + scaled
//                     Jacobian (base_project.py:365-391,443-488; project/utils.py:10-89;
//                     loss_functions/squared_loss/squared_loss_function.py:27-80;
//                     linear_scale_factor.py:27-42)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "sbm_plugin.h"

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

static int sbm_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define SBM_HIP(expr)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess) return sbm_fail(-2, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

enum { SBM_E_ARG = -1, SBM_E_HIP = -2, SBM_E_PLUGIN = -3 };

// ---------------------------------------------------------------------------
// objects
// ---------------------------------------------------------------------------
struct sbm_ctx {
  int device;
  hipStream_t stream;
  bool own_stream;
};

struct sbm_model;

template <class T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int reserve(size_t want) {
    if (want <= n) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
    if (hipMalloc((void**)&p, want * sizeof(T)) != hipSuccess) return -1;
    n = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
};

struct sbm_model {
  sbm_ctx* ctx;
  void* dl;
  sbm_plugin_info_t info;
  sbm_plugin_launch_fn launch;
  // launch order of the sensitivity kernels (launch_sens): trajectories sorted by the cost of the
  // previous launch of the same size, most expensive first
  DevBuf<int32_t> order, cost_steps, cost_rej;
  int order_T = 0;   // number of trajectories `order` is a permutation of (0: none yet)
};

struct sbm_project {
  sbm_model* model;
  int E, q, R, G, NPR, NSP, compat, loss;
  int n_params, n_vars, n_sens;
  int n_t_max;  // longest per-experiment grid
  // static device data
  DevBuf<int32_t> pmap, sens_col, tgrid_off, grid_len, row_exp, row_tidx, row_var_off, row_vars, row_sf,
      prior_idx, inv_ptr, inv_m, sfp_group;
  DevBuf<double> pfixed, tgrid, row_data, row_sigma, prior_mean, prior_sigma, sfp_mean, sfp_sigma;
  // custom observables (postfix programs): see sbm_project_desc
  int n_programs = 0, n_custom_weights = 0;
  DevBuf<int32_t> row_prog, row_w0, prog_base, prog_sub_off, prog_code;
  DevBuf<double> prog_const, row_time;
  // per-call scratch, grown on demand
  DevBuf<double> P, Y, S, sims, sf;
  DevBuf<int32_t> traj_status, traj_steps, traj_rej, goff, glen;
  int scratch_V = 0;
  // Richardson extrapolation of the implicit-midpoint runs (sbm_project_set_extrapolation)
  int richardson = 0;
  DevBuf<double> Yx[2], Sx[2];
  DevBuf<int32_t> statx, stepsx;
};

extern "C" const char* sbm_last_error(void) { return g_err; }
extern "C" int sbm_abi_version(void) { return SBM_ABI_VERSION; }

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
extern "C" int sbm_ctx_create(int device, void* stream, sbm_ctx** out) {
  if (!out) return sbm_fail(SBM_E_ARG, "sbm_ctx_create: out is NULL");
  int n = 0;
  SBM_HIP(hipGetDeviceCount(&n));
  if (device < 0 || device >= n) return sbm_fail(SBM_E_ARG, "sbm_ctx_create: device %d of %d", device, n);
  SBM_HIP(hipSetDevice(device));
  sbm_ctx* c = new sbm_ctx();
  c->device = device;
  // NULL = the device's default (null) stream, which is also what torch enqueues on
  // unless told otherwise: copies made by the caller and our kernels stay ordered
  c->stream = (hipStream_t)stream;
  c->own_stream = false;
  *out = c;
  return 0;
}

extern "C" int sbm_ctx_destroy(sbm_ctx* c) {
  if (!c) return 0;
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

extern "C" int sbm_ctx_synchronize(sbm_ctx* c) {
  if (!c) return sbm_fail(SBM_E_ARG, "sbm_ctx_synchronize: ctx is NULL");
  SBM_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int sbm_ctx_device(const sbm_ctx* c) { return c ? c->device : -1; }

// ---------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------
extern "C" int sbm_model_load(sbm_ctx* ctx, const char* path, sbm_model** out) {
  if (!ctx || !path || !out) return sbm_fail(SBM_E_ARG, "sbm_model_load: NULL argument");
  SBM_HIP(hipSetDevice(ctx->device));
  void* dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!dl) return sbm_fail(SBM_E_PLUGIN, "dlopen(%s): %s", path, dlerror());
  auto info_fn = (sbm_plugin_info_fn)dlsym(dl, "sbm_plugin_info");
  auto launch_fn = (sbm_plugin_launch_fn)dlsym(dl, "sbm_plugin_launch");
  if (!info_fn || !launch_fn) {
    dlclose(dl);
    return sbm_fail(SBM_E_PLUGIN, "%s is not an sbm model plugin", path);
  }
  sbm_model* m = new sbm_model();
  m->ctx = ctx;
  m->dl = dl;
  m->launch = launch_fn;
  memset(&m->info, 0, sizeof(m->info));
  info_fn(&m->info);
  if (m->info.abi != SBM_PLUGIN_ABI) {
    int abi = m->info.abi;
    dlclose(dl);
    delete m;
    return sbm_fail(SBM_E_PLUGIN, "%s: plugin ABI %d, library expects %d", path, abi, SBM_PLUGIN_ABI);
  }
  *out = m;
  return 0;
}

extern "C" int sbm_model_unload(sbm_model* m) {
  if (!m) return 0;
  // the plugin's code object stays registered with the HIP runtime: do not dlclose
  (void)hipSetDevice(m->ctx->device);
  m->order.release(); m->cost_steps.release(); m->cost_rej.release();
  delete m;
  return 0;
}

extern "C" int sbm_model_info(const sbm_model* m, int32_t* n_vars, int32_t* n_params, int32_t* n_sens,
                              char* name_buf, int32_t name_buf_len) {
  if (!m) return sbm_fail(SBM_E_ARG, "sbm_model_info: model is NULL");
  if (n_vars) *n_vars = m->info.n_vars;
  if (n_params) *n_params = m->info.n_params;
  if (n_sens) *n_sens = m->info.n_sens;
  if (name_buf && name_buf_len > 0) {
    strncpy(name_buf, m->info.name, name_buf_len - 1);
    name_buf[name_buf_len - 1] = 0;
  }
  return 0;
}

// ---------------------------------------------------------------------------
// batched integration
// ---------------------------------------------------------------------------
static int check_opts(const sbm_integrator_opts* o, const char* who) {
  if (!o) return sbm_fail(SBM_E_ARG, "%s: opts is NULL", who);
  if (o->method == SBM_RK4_FIXED) {
    if (!(o->h0 > 0.0)) return sbm_fail(SBM_E_ARG, "%s: RK4 needs h0 > 0", who);
  } else if (o->method == SBM_DOPRI45 || o->method == SBM_DOP853 || o->method == SBM_IMPLICIT_ADAPTIVE ||
             o->method == SBM_IMPLICIT_EXTRAP) {
    if (!(o->rtol > 0.0) || !(o->atol > 0.0)) return sbm_fail(SBM_E_ARG, "%s: adaptive methods need rtol, atol > 0", who);
  } else if (o->method == SBM_IMPLICIT_MIDPOINT || o->method == SBM_IMPLICIT_MIDPOINT_GRADED) {
    if (!(o->h0 > 0.0)) return sbm_fail(SBM_E_ARG, "%s: implicit midpoint needs h0 > 0", who);
  } else {
    return sbm_fail(SBM_E_ARG, "%s: unknown method %d", who, o->method);
  }
  if (o->step_mult < 0 || o->step_mult > 65536) return sbm_fail(SBM_E_ARG, "%s: step_mult %d", who, o->step_mult);
  if (o->variant < SBM_VARIANT_AUTO || o->variant > SBM_VARIANT_PACKED)
    return sbm_fail(SBM_E_ARG, "%s: unknown kernel variant %d", who, o->variant);
  return 0;
}

static int check_model_fits(const sbm_model* m, const sbm_integrator_opts& o, const char* who) {
  if ((o.method == SBM_IMPLICIT_MIDPOINT || o.method == SBM_IMPLICIT_MIDPOINT_GRADED || o.method == SBM_IMPLICIT_ADAPTIVE ||
       o.method == SBM_IMPLICIT_EXTRAP) && m->info.n_vars > SBM_IMPLICIT_MAX_NV)
    return sbm_fail(SBM_E_ARG, "%s: the implicit midpoint kernels hold a column of the sensitivity matrix per lane in "
                    "registers: n_vars <= %d (model '%s' has %d)", who, SBM_IMPLICIT_MAX_NV, m->info.name, m->info.n_vars);
  return 0;
}

static int launch_failed(sbm_model* m, const sbm_integrator_opts& o, int e, const char* who) {
  if (e == (int)hipErrorInvalidConfiguration &&
      (o.method == SBM_IMPLICIT_MIDPOINT || o.method == SBM_IMPLICIT_MIDPOINT_GRADED || o.method == SBM_IMPLICIT_ADAPTIVE ||
       o.method == SBM_IMPLICIT_EXTRAP))
    return sbm_fail(SBM_E_ARG, "%s: model '%s' (%d state variables, %d sensitivity columns) does not fit the implicit kernel "
                    "asked for: its tables need more than the 160 KB of LDS of a compute unit (the error-controlled kernel "
                    "parks two copies of a 64-column block of S there); the fixed-step method may still fit", who,
                    m->info.name, m->info.n_vars, m->info.n_sens);
  if (e == (int)hipErrorInvalidConfiguration && o.method == SBM_DOP853)
    return sbm_fail(SBM_E_ARG, "%s: DOP853 runs on the row-group sensitivity kernels and the state-rows kernels; model '%s' "
                    "has no row split (or more than 256 state variables): use DOPRI45", who, m->info.name);
  return sbm_fail(SBM_E_HIP, "%s: kernel launch failed: %s", who, hipGetErrorString((hipError_t)e));
}

static int launch(sbm_model* m, int kind, const sbm_kernel_args& a, const char* who) {
  if (int rc = check_model_fits(m, a.opts, who)) return rc;
  SBM_HIP(hipSetDevice(m->ctx->device));
  int e = m->launch(kind, &a, (void*)m->ctx->stream);
  if (e != 0) return launch_failed(m, a.opts, e, who);
  return 0;
}

// ---- launch order of the sensitivity kernels ---------------------------------------------------
// An adaptive integrator takes a different number of steps for every parameter vector (configs[2]:
// 445 ... 574, mean 485).  One trajectory occupies one SIMD for its whole life, the hardware hands
// out workgroups in index order, and the kernel ends when the last trajectory does: in index order
// the chip idles for 5.8 % of the launch at 4 trajectories per SIMD (scripts/dev_balance.py).  A
// fitting loop integrates nearly the same ensemble again and again, so the step counts of the
// previous launch predict the next one: sort by them, longest first (LPT), 1.9 %.  The order only
// changes which workgroup integrates which trajectory, never a result.
__global__ void __launch_bounds__(1024) k_order(const int32_t* __restrict__ steps, const int32_t* __restrict__ rej, int T,
                                                int32_t* __restrict__ order) {
  constexpr int NB = 2048;
  __shared__ int hist[NB];
  __shared__ int smax;
  const int tid = threadIdx.x;
  for (int b = tid; b < NB; b += blockDim.x) hist[b] = 0;
  if (tid == 0) smax = 1;
  __syncthreads();
  int mx = 1;
  for (int i = tid; i < T; i += blockDim.x) mx = max(mx, steps[i] + rej[i]);
  atomicMax(&smax, mx);
  __syncthreads();
  const double scale = (double)(NB - 1) / (double)smax;
  // bin 0 = most expensive
  for (int i = tid; i < T; i += blockDim.x) {
    const int c = max(0, steps[i] + rej[i]);
    atomicAdd(&hist[NB - 1 - (int)(c * scale)], 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int b = 0; b < NB; ++b) { const int h = hist[b]; hist[b] = run; run += h; }
  }
  __syncthreads();
  for (int i = tid; i < T; i += blockDim.x) {
    const int c = max(0, steps[i] + rej[i]);
    order[atomicAdd(&hist[NB - 1 - (int)(c * scale)], 1)] = i;
  }
}

static int launch_sens(sbm_model* m, sbm_kernel_args a, const char* who) {
  if (int rc = check_model_fits(m, a.opts, who)) return rc;
  SBM_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->stream;
  const int T = a.n_traj;
  // the order is only worth its bookkeeping when trajectories queue behind each other
  const bool use = T >= 2048;
  if (use) {
    if (m->order.reserve(T) || m->cost_steps.reserve(T) || m->cost_rej.reserve(T))
      return sbm_fail(SBM_E_HIP, "%s: out of device memory", who);
    if (m->order_T == T) a.order = m->order.p;
    if (!a.n_steps) a.n_steps = m->cost_steps.p;
    if (!a.n_reject) a.n_reject = m->cost_rej.p;
  }
  int e = m->launch(SBM_KIND_SENS, &a, (void*)s);
  if (e != 0) return launch_failed(m, a.opts, e, who);
  if (use) {
    hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, s, a.n_steps, a.n_reject, T, m->order.p);
    SBM_HIP(hipGetLastError());
    m->order_T = T;
  }
  return 0;
}

extern "C" int sbm_simulate_batch(sbm_model* m, const double* P, int32_t V, const double* t_out, int32_t n_t,
                                  const double* y0, const sbm_integrator_opts* opts, double* Y, int32_t* status,
                                  int32_t* n_steps, int32_t* n_reject) {
  if (!m || !P || !t_out || !Y) return sbm_fail(SBM_E_ARG, "sbm_simulate_batch: NULL argument");
  if (V < 0 || n_t < 0) return sbm_fail(SBM_E_ARG, "sbm_simulate_batch: negative size");
  int rc = check_opts(opts, "sbm_simulate_batch");
  if (rc) return rc;
  if (V == 0 || n_t == 0) return 0;
  sbm_kernel_args a;
  memset(&a, 0, sizeof(a));
  a.P = P; a.t_out = t_out; a.y0 = y0; a.Y = Y;
  a.status = status; a.n_steps = n_steps; a.n_reject = n_reject;
  a.n_traj = V; a.n_t = n_t; a.opts = *opts;
  return launch(m, SBM_KIND_STATE, a, "sbm_simulate_batch");
}

extern "C" int sbm_sens_batch(sbm_model* m, const double* P, int32_t V, const double* t_out, int32_t n_t,
                              const double* yS0, const sbm_integrator_opts* opts, double* Y, double* S,
                              int32_t* status, int32_t* n_steps, int32_t* n_reject) {
  if (!m || !P || !t_out || !S) return sbm_fail(SBM_E_ARG, "sbm_sens_batch: NULL argument");
  if (V < 0 || n_t < 0) return sbm_fail(SBM_E_ARG, "sbm_sens_batch: negative size");
  int rc = check_opts(opts, "sbm_sens_batch");
  if (rc) return rc;
  if (V == 0 || n_t == 0) return 0;
  sbm_kernel_args a;
  memset(&a, 0, sizeof(a));
  a.P = P; a.t_out = t_out;
  a.y0 = yS0; a.s0 = yS0 ? yS0 + m->info.n_vars : nullptr;
  a.Y = Y; a.S = S;
  a.status = status; a.n_steps = n_steps; a.n_reject = n_reject;
  a.n_traj = V; a.n_t = n_t; a.opts = *opts;
  return launch_sens(m, a, "sbm_sens_batch");
}

// ---- host-pointer variants -------------------------------------------------
namespace {
struct HostStage {
  std::vector<void*> allocs;
  ~HostStage() {
    for (void* p : allocs) (void)hipFree(p);
  }
  template <class T>
  T* up(const T* h, size_t n, hipStream_t s) {
    if (!h || n == 0) return nullptr;
    T* d = nullptr;
    if (hipMalloc((void**)&d, n * sizeof(T)) != hipSuccess) return nullptr;
    allocs.push_back(d);
    if (hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, s) != hipSuccess) return nullptr;
    return d;
  }
  template <class T>
  T* dev(size_t n) {
    if (n == 0) return nullptr;
    T* d = nullptr;
    if (hipMalloc((void**)&d, n * sizeof(T)) != hipSuccess) return nullptr;
    allocs.push_back(d);
    return d;
  }
};
}  // namespace

#define SBM_NEED(ptr, what) \
  if (!(ptr)) return sbm_fail(SBM_E_HIP, "%s: device staging failed (out of memory?)", what)

extern "C" int sbm_simulate_batch_host(sbm_model* m, const double* P, int32_t V, const double* t_out, int32_t n_t,
                                       const double* y0, const sbm_integrator_opts* opts, double* Y,
                                       int32_t* status, int32_t* n_steps, int32_t* n_reject) {
  if (!m || !P || !t_out || !Y) return sbm_fail(SBM_E_ARG, "sbm_simulate_batch_host: NULL argument");
  if (V <= 0 || n_t <= 0) return V < 0 || n_t < 0 ? sbm_fail(SBM_E_ARG, "negative size") : 0;
  SBM_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->stream;
  const int nv = m->info.n_vars, np = m->info.n_params;
  HostStage st;
  double* dP = st.up(P, (size_t)V * np, s); SBM_NEED(dP, "simulate");
  double* dt = st.up(t_out, (size_t)n_t, s); SBM_NEED(dt, "simulate");
  double* dy0 = y0 ? st.up(y0, (size_t)nv, s) : nullptr;
  double* dY = st.dev<double>((size_t)V * n_t * nv); SBM_NEED(dY, "simulate");
  int32_t* dst = st.dev<int32_t>((size_t)V * 3); SBM_NEED(dst, "simulate");
  int rc = sbm_simulate_batch(m, dP, V, dt, n_t, dy0, opts, dY, dst, dst + V, dst + 2 * V);
  if (rc) return rc;
  SBM_HIP(hipMemcpyAsync(Y, dY, (size_t)V * n_t * nv * sizeof(double), hipMemcpyDeviceToHost, s));
  if (status) SBM_HIP(hipMemcpyAsync(status, dst, V * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (n_steps) SBM_HIP(hipMemcpyAsync(n_steps, dst + V, V * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (n_reject) SBM_HIP(hipMemcpyAsync(n_reject, dst + 2 * V, V * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  SBM_HIP(hipStreamSynchronize(s));
  return 0;
}

extern "C" int sbm_sens_batch_host(sbm_model* m, const double* P, int32_t V, const double* t_out, int32_t n_t,
                                   const double* yS0, const sbm_integrator_opts* opts, double* Y, double* S,
                                   int32_t* status, int32_t* n_steps, int32_t* n_reject) {
  if (!m || !P || !t_out || !S) return sbm_fail(SBM_E_ARG, "sbm_sens_batch_host: NULL argument");
  if (V <= 0 || n_t <= 0) return V < 0 || n_t < 0 ? sbm_fail(SBM_E_ARG, "negative size") : 0;
  SBM_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->stream;
  const int nv = m->info.n_vars, np = m->info.n_params, nk = m->info.n_sens;
  HostStage st;
  double* dP = st.up(P, (size_t)V * np, s); SBM_NEED(dP, "sens");
  double* dt = st.up(t_out, (size_t)n_t, s); SBM_NEED(dt, "sens");
  double* d0 = yS0 ? st.up(yS0, (size_t)nv * (1 + nk), s) : nullptr;
  double* dY = st.dev<double>((size_t)V * n_t * nv); SBM_NEED(dY, "sens");
  double* dS = st.dev<double>((size_t)V * n_t * nv * nk); SBM_NEED(dS, "sens");
  int32_t* dst = st.dev<int32_t>((size_t)V * 3); SBM_NEED(dst, "sens");
  int rc = sbm_sens_batch(m, dP, V, dt, n_t, d0, opts, dY, dS, dst, dst + V, dst + 2 * V);
  if (rc) return rc;
  if (Y) SBM_HIP(hipMemcpyAsync(Y, dY, (size_t)V * n_t * nv * sizeof(double), hipMemcpyDeviceToHost, s));
  SBM_HIP(hipMemcpyAsync(S, dS, (size_t)V * n_t * nv * nk * sizeof(double), hipMemcpyDeviceToHost, s));
  if (status) SBM_HIP(hipMemcpyAsync(status, dst, V * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (n_steps) SBM_HIP(hipMemcpyAsync(n_steps, dst + V, V * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (n_reject) SBM_HIP(hipMemcpyAsync(n_reject, dst + 2 * V, V * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  SBM_HIP(hipStreamSynchronize(s));
  return 0;
}

// ===========================================================================
// Project
// ===========================================================================
// trajectory id = v * E + e

// theta -> P.  grid: ceil(V*E*NP / 256)
__global__ void k_gather_params(const double* __restrict__ Theta, const int32_t* __restrict__ pmap,
                                const double* __restrict__ pfixed, int V, int E, int NP, int q,
                                double* __restrict__ P) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)V * E * NP;
  if (idx >= total) return;
  const int m = (int)(idx % NP);
  const size_t ve = idx / NP;
  const int e = (int)(ve % E);
  const size_t v = ve / E;
  const int g = pmap[e * NP + m];
  P[idx] = (g >= 0) ? exp(Theta[v * q + g]) : pfixed[e * NP + m];
}

// per-trajectory grid descriptors (same for every vector): grid = ceil(V*E/256)
__global__ void k_fill_grids(const int32_t* __restrict__ tgrid_off, int V, int E, int32_t* __restrict__ goff,
                             int32_t* __restrict__ glen) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= V * E) return;
  const int e = idx % E;
  goff[idx] = tgrid_off[e];
  glen[idx] = tgrid_off[e + 1] - tgrid_off[e];
}

__device__ __forceinline__ double block_sum(double v, double* red /*[4]*/) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  double s = 0.0;
  const int nw = blockDim.x >> 6;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

struct AssembleArgs {
  // static project data
  int E, q, R, G, NPR, NSP, NP, NV, NK, n_t, compat, loss;
  const int32_t *row_exp, *row_tidx, *row_var_off, *row_vars, *row_sf, *prior_idx, *sfp_group;
  const double *row_data, *row_sigma, *prior_mean, *prior_sigma, *sfp_mean, *sfp_sigma;
  const int32_t *inv_ptr, *inv_m;  // [E][q+1] CSR: model params mapped to project column c in experiment e
  const int32_t* sens_col;
  // per call
  const double* Theta;        // [V][q]
  const double* Y;            // [V*E][n_t][NV]
  const double* S;            // [V*E][n_t][NV][NK]   (nullable: residuals only)
  const int32_t* traj_status; // [V*E]
  const int32_t* traj_steps;  // [V*E]
  double* sims;               // [V][R]       (scratch or user)
  double* Rout;               // [V][R+NPR]
  double* sf;                 // [V][G]       nullable
  double* norms;              // [V]          nullable
  int32_t* status;            // [V]          nullable
  int32_t* n_steps;           // [V]          nullable: sum over the vector's trajectories
  double* J;                  // [V][R+NPR][q] nullable
  double* Jmodel;             // [V][R][q]     nullable
  double* grad;               // [V][q]        nullable
  double* sf_grad;            // [V][G][q]     nullable
  // direct mode (sbm_loss_eval): caller-supplied simulations / model Jacobian instead of Y / S
  const double* sims_in;      // [V][R]        nullable
  const double* Jm_in;        // [V][R][q]     nullable
  const int32_t* row_plain;   // [R] nullable: 1 = row keeps the plain (s - d)/sigma form under the log loss
  // custom observables: row_prog[r] = program or -1; row_w0[r] = first slot of the row's dg/dy weights in the LDS
  // table (or -1); prog_base[c] = first subprogram of program c in prog_sub_off
  int NW;                     // total weight slots (sum of the variable counts of the custom rows)
  const int32_t *row_prog, *row_w0, *prog_base, *prog_sub_off, *prog_code;
  const double *prog_const, *row_time;
};

// One subprogram of a custom observable: a postfix stack machine (include/sbm.h, SBM_OP_*).  The host has
// checked every program at load time (opcodes, operand ranges, stack depth <= SBM_PROG_MAX_STACK).
__device__ double sbm_prog_eval(const int32_t* __restrict__ code, const double* __restrict__ consts,
                                const double* __restrict__ Yb, const int32_t* __restrict__ vars, double t) {
  double st[SBM_PROG_MAX_STACK];
  int sp = 0;
  for (int pc = 0;; ++pc) {
    const int op = code[pc];
    if (op == SBM_OP_END) break;
    switch (op) {
      case SBM_OP_VAR: st[sp++] = Yb[vars[code[++pc]]]; break;
      case SBM_OP_CONST: st[sp++] = consts[code[++pc]]; break;
      case SBM_OP_TIME: st[sp++] = t; break;
      case SBM_OP_ADD: --sp; st[sp - 1] += st[sp]; break;
      case SBM_OP_SUB: --sp; st[sp - 1] -= st[sp]; break;
      case SBM_OP_MUL: --sp; st[sp - 1] *= st[sp]; break;
      case SBM_OP_DIV: --sp; st[sp - 1] /= st[sp]; break;
      case SBM_OP_POW: --sp; st[sp - 1] = pow(st[sp - 1], st[sp]); break;
      case SBM_OP_POWI: {
        int n = code[++pc];
        const bool inv = n < 0;
        n = inv ? -n : n;
        double b = st[sp - 1], r = 1.0;
        while (n) { if (n & 1) r *= b; b *= b; n >>= 1; }
        st[sp - 1] = inv ? 1.0 / r : r;
        break;
      }
      case SBM_OP_NEG: st[sp - 1] = -st[sp - 1]; break;
      case SBM_OP_EXP: st[sp - 1] = exp(st[sp - 1]); break;
      case SBM_OP_LOG: st[sp - 1] = log(st[sp - 1]); break;
      case SBM_OP_SQRT: st[sp - 1] = sqrt(st[sp - 1]); break;
      case SBM_OP_TANH: st[sp - 1] = tanh(st[sp - 1]); break;
      case SBM_OP_SIN: st[sp - 1] = sin(st[sp - 1]); break;
      case SBM_OP_COS: st[sp - 1] = cos(st[sp - 1]); break;
      case SBM_OP_ABS: st[sp - 1] = fabs(st[sp - 1]); break;
      case SBM_OP_SIGN: st[sp - 1] = st[sp - 1] > 0.0 ? 1.0 : (st[sp - 1] < 0.0 ? -1.0 : 0.0); break;
      default: return __builtin_nan("");
    }
  }
  return sp == 1 ? st[0] : __builtin_nan("");
}

// One block (256 threads) per parameter vector.
// LDS: sims[R], B[G], sde[G], sds[G], dB[G][q] (Jacobian mode), reduction scratch.
__global__ void __launch_bounds__(256) k_assemble(AssembleArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int v = blockIdx.x;
  const int tid = threadIdx.x;
  const int R = a.R, G = a.G, q = a.q, E = a.E;
  double* s_sim = smem;                 // [R]
  double* s_res = s_sim + R;            // [R]   weighted residuals
  double* s_B = s_res + R;              // [max(G,1)]
  double* s_sde = s_B + (G > 0 ? G : 1);
  double* s_sds = s_sde + (G > 0 ? G : 1);
  double* s_red = s_sds + (G > 0 ? G : 1);  // [4]
  // row tables staged once per block: every thread of pass A / B would otherwise chase them through
  // global memory for each of its elements (a chain of dependent loads per element: the kernel was bound
  // by that latency, not by HBM)
  double* s_d = s_red + 4;                  // [R] measurement
  double* s_sg = s_d + R;                   // [R] sigma
  int* s_roff = (int*)(s_sg + R);           // [R] (e * n_t + tidx) * NV: offset of the row's time point in the vector's Y block
  int* s_rexp = s_roff + R;                 // [R] experiment
  int* s_rv0 = s_rexp + R;                  // [R] first entry of the row's variable list
  int* s_rnv = s_rv0 + R;                   // [R] its length
  int* s_rvar = s_rnv + R;                  // [R] first variable (the only one of a 'direct' mapping)
  int* s_rsf = s_rvar + R;                  // [R] scale-factor group or -1
  int* s_rw0 = s_rsf + R;                   // [R + (R & 1)] first dg/dy weight slot of a custom row, -1 for plain rows
  double* s_w = (double*)(s_rw0 + R + (R & 1));   // [NW] dg/dy_k of the custom rows at this vector's sampled states
  double* s_dB = s_w + a.NW;                // 8-byte aligned.  [G][q], then the partial sums [2][G][rpp][q]
  __shared__ int s_bad;

  if (tid == 0) s_bad = 0;
  for (int r = tid; r < R; r += blockDim.x) {
    s_d[r] = a.row_data[r];
    s_sg[r] = a.row_sigma[r];
    s_rsf[r] = a.row_sf[r];
    s_rw0[r] = (a.row_w0 && !a.sims_in) ? a.row_w0[r] : -1;
    if (!a.sims_in) {
      const int e = a.row_exp[r], v0 = a.row_var_off[r];
      s_rexp[r] = e;
      s_roff[r] = (e * a.n_t + a.row_tidx[r]) * a.NV;
      s_rv0[r] = v0;
      s_rnv[r] = a.row_var_off[r + 1] - v0;
      s_rvar[r] = a.row_var_off[r + 1] > v0 ? a.row_vars[v0] : 0;
    } else {
      s_rexp[r] = 0; s_roff[r] = 0; s_rv0[r] = 0; s_rnv[r] = 0; s_rvar[r] = 0;
    }
  }
  __syncthreads();
  // status of the vector = worst status of its trajectories
  int st = 0, steps = 0;
  for (int e = tid; a.traj_status && e < E; e += blockDim.x) {
    st = max(st, a.traj_status[(size_t)v * E + e]);
    steps += a.traj_steps ? a.traj_steps[(size_t)v * E + e] : 0;
  }
  if (st != 0) atomicMax(&s_bad, st);
  if (a.n_steps) {
    const double tot = block_sum((double)steps, s_red);
    if (tid == 0) a.n_steps[v] = (int32_t)tot;
  }

  // ---- 1. sample + map: sims[r] = sum_{var in vars(r)} Y[traj][tidx][var] ----
  for (int r = tid; r < R; r += blockDim.x) {
    double s = 0.0;
    if (a.sims_in) {
      s = a.sims_in[(size_t)v * R + r];
    } else {
      const double* Yb = a.Y + (size_t)v * E * a.n_t * a.NV + s_roff[r];
      if (s_rw0[r] >= 0) {
        // custom observable: value g(y), and dg/dy_k parked in LDS for the Jacobian pass
        const int32_t* sub = a.prog_sub_off + a.prog_base[a.row_prog[r]];
        const int32_t* vars = a.row_vars + s_rv0[r];
        const double tr = a.row_time[r];
        s = sbm_prog_eval(a.prog_code + sub[0], a.prog_const, Yb, vars, tr);
        if (a.S)
          for (int k = 0; k < s_rnv[r]; ++k)
            s_w[s_rw0[r] + k] = sbm_prog_eval(a.prog_code + sub[1 + k], a.prog_const, Yb, vars, tr);
      } else if (s_rnv[r] == 1) s = Yb[s_rvar[r]];
      else for (int k = s_rv0[r]; k < s_rv0[r] + s_rnv[r]; ++k) s += Yb[a.row_vars[k]];
    }
    s_sim[r] = s;
    // NaN simulations -> inf rows; the log loss cannot take a non-positive simulation either
    const bool logrow = a.loss == SBM_LOSS_LOG_SQUARE && !(a.row_plain && a.row_plain[r]);
    if (!(s == s) || (logrow && !(s > 0.0))) atomicMax(&s_bad, (int)SBM_NON_FINITE);
  }
  __syncthreads();
  const int bad = s_bad;
  if (a.status && tid == 0) a.status[v] = bad;

  const int RT = R + a.NPR + a.NSP;
  if (bad) {
    // reference: NaN in simulations -> every residual / Jacobian entry is inf
    // (squared_loss_function.py:28-32,46-50)
    const double inf = __builtin_inf();
    for (int r = tid; r < RT; r += blockDim.x) a.Rout[(size_t)v * RT + r] = inf;
    if (a.sims) for (int r = tid; r < R; r += blockDim.x) a.sims[(size_t)v * R + r] = s_sim[r];
    if (a.sf) for (int g = tid; g < G; g += blockDim.x) a.sf[(size_t)v * G + g] = __builtin_nan("");
    if (a.norms && tid == 0) a.norms[v] = inf;
    if (a.J) for (size_t i = tid; i < (size_t)RT * q; i += blockDim.x) a.J[(size_t)v * RT * q + i] = inf;
    if (a.Jmodel) for (size_t i = tid; i < (size_t)R * q; i += blockDim.x) a.Jmodel[(size_t)v * R * q + i] = inf;
    if (a.grad) for (int c = tid; c < q; c += blockDim.x) a.grad[(size_t)v * q + c] = inf;
    if (a.sf_grad) for (int i = tid; i < G * q; i += blockDim.x) a.sf_grad[(size_t)v * G * q + i] = __builtin_nan("");
    return;
  }
  if (a.sims) for (int r = tid; r < R; r += blockDim.x) a.sims[(size_t)v * R + r] = s_sim[r];

  // ---- 2. scale factors B_g = sum(s d / sigma^2) / sum(s^2 / sigma^2) ----
  for (int g = 0; g < G; ++g) {
    double sde = 0.0, sds = 0.0;
    for (int r = tid; r < R; r += blockDim.x) {
      if (s_rsf[r] == g) {
        if (a.loss == SBM_LOSS_LOG_SQUARE) {
          // log_scale_factor.py:17-28: weights 1/(sigma/d)^2; log B = sum(w (log d - log s)) / sum(w)
          const double w = (s_d[r] * s_d[r]) / (s_sg[r] * s_sg[r]);
          sde += (log(s_d[r]) - log(s_sim[r])) * w;
          sds += w;
        } else {
          const double w = 1.0 / (s_sg[r] * s_sg[r]);
          sde += s_sim[r] * s_d[r] * w;
          sds += s_sim[r] * s_sim[r] * w;
        }
      }
    }
    sde = block_sum(sde, s_red);
    sds = block_sum(sds, s_red);
    if (tid == 0) {
      s_sde[g] = sde;
      s_sds[g] = sds;
      const double B = (a.loss == SBM_LOSS_LOG_SQUARE) ? exp(sde / sds) : sde / sds;
      s_B[g] = B;
      if (a.sf) a.sf[(size_t)v * G + g] = B;
    }
  }
  __syncthreads();

  // ---- 3. residuals ----
  double nrm = 0.0;
  for (int r = tid; r < RT; r += blockDim.x) {
    double res;
    if (r < R) {
      const int g = s_rsf[r];
      const double B = g >= 0 ? s_B[g] : 1.0;
      const bool logrow = a.loss == SBM_LOSS_LOG_SQUARE && !(a.row_plain && a.row_plain[r]);
      res = logrow ? (log(B * s_sim[r]) - log(s_d[r])) / s_sg[r]
                   : (B * s_sim[r] - s_d[r]) / s_sg[r];
      s_res[r] = res;
    } else if (r < R + a.NPR) {
      const int k = r - R;
      res = (a.Theta[(size_t)v * q + a.prior_idx[k]] - a.prior_mean[k]) / a.prior_sigma[k];
    } else {
      const int k = r - R - a.NPR;  // (log B - prior)/sigma, linear_scale_factor.py:55-61
      res = (log(s_B[a.sfp_group[k]]) - a.sfp_mean[k]) / a.sfp_sigma[k];
    }
    a.Rout[(size_t)v * RT + r] = res;
    nrm += res * res;
  }
  nrm = block_sum(nrm, s_red);
  if (a.norms && tid == 0) a.norms[v] = nrm;
  if (!a.J && !a.Jmodel && !a.grad && !a.sf_grad) return;

  // ---- 4. model Jacobian with the log-parameter chain rule (base_project.py:450-455,482-485):
  //      Jm[r][c] = exp(theta_c) * sum_{model params m of experiment e reading slot c} sum_{var in vars(r)} S[var][m]
  // pass A: write Jm, accumulate per SF group the two products J^T (d/sigma^2), J^T (s/sigma^2)
  double* Jv = a.J ? a.J + (size_t)v * RT * q : nullptr;
  double* Jm = a.Jmodel ? a.Jmodel + (size_t)v * R * q : nullptr;
  const double* th = a.Theta + (size_t)v * q;
  // partial sums owned by ONE thread each, [which][g][row lane][c]: no atomics, and the final
  // reduction runs in a fixed order, so results are bitwise reproducible run to run
  const int nthr = blockDim.x;
  const int cw = q < nthr ? q : nthr;  // columns handled side by side (consecutive threads -> consecutive columns)
  const int rpp = nthr / cw;           // row lanes
  const int Gn = G > 0 ? G : 1;
  double* s_part = s_dB + (size_t)Gn * q;  // [2][Gn][rpp][q]
  for (int i = tid; i < 2 * Gn * rpp * q; i += nthr) s_part[i] = 0.0;
  __syncthreads();

  const int cl = tid % cw, rl = tid / cw;
  for (int c0 = 0; c0 < q; c0 += cw) {
    const int c = c0 + cl;
    if (c >= q || rl >= rpp) continue;
    const double dth = a.Jm_in ? 1.0 : exp(th[c]);
    int gcur = -1;
    int e_cur = -1, k0 = 0, k1 = 0, sc1 = -1;
    double jde = 0.0, jds = 0.0;
    for (int r = rl; r < R; r += rpp) {
      double jm = 0.0;
      if (a.Jm_in) {
        jm = a.Jm_in[((size_t)v * R + r) * q + c];
      } else {
        const int e = s_rexp[r];
        if (e != e_cur) {   // rows are sorted by experiment: the column's model parameters change rarely
          e_cur = e;
          k0 = a.inv_ptr[e * (q + 1) + c];
          k1 = a.inv_ptr[e * (q + 1) + c + 1];
          sc1 = (k1 - k0 == 1) ? a.sens_col[a.inv_m[k0]] : -2;   // the common case: one model parameter per slot
        }
        const double* Sb = a.S + ((size_t)v * E * a.n_t * a.NV + s_roff[r]) * a.NK;
        const int w0 = s_rw0[r];
        if (sc1 >= 0 && s_rnv[r] == 1 && w0 < 0) {
          jm = Sb[(size_t)s_rvar[r] * a.NK + sc1];
        } else if (sc1 != -1) {
          for (int k = k0; k < k1; ++k) {
            const int sc = a.sens_col[a.inv_m[k]];
            if (sc < 0) continue;
            if (w0 < 0) {
              for (int kk = s_rv0[r]; kk < s_rv0[r] + s_rnv[r]; ++kk) jm += Sb[(size_t)a.row_vars[kk] * a.NK + sc];
            } else {   // custom observable: sum_k dg/dy_k * S[var_k][sc]
              for (int kk = 0; kk < s_rnv[r]; ++kk)
                jm = fma(s_w[w0 + kk], Sb[(size_t)a.row_vars[s_rv0[r] + kk] * a.NK + sc], jm);
            }
          }
        }
        jm *= dth;
      }
      if (Jm) Jm[(size_t)r * q + c] = jm;
      if (Jv) Jv[(size_t)r * q + c] = jm;
      const int g = s_rsf[r];
      if (g != gcur) {  // rows of one group are mostly consecutive: flush on change
        if (gcur >= 0) {
          s_part[((size_t)(0 * Gn + gcur) * rpp + rl) * q + c] += jde;
          s_part[((size_t)(1 * Gn + gcur) * rpp + rl) * q + c] += jds;
        }
        gcur = g; jde = 0.0; jds = 0.0;
      }
      if (g >= 0) {
        if (a.loss == SBM_LOSS_LOG_SQUARE) {  // log_scale_factor.py:30-36: sum J / (s (sigma/d)^2)
          jde += jm * (s_d[r] * s_d[r]) / (s_sg[r] * s_sg[r] * s_sim[r]);
        } else {
          const double w = 1.0 / (s_sg[r] * s_sg[r]);
          jde += jm * s_d[r] * w;
          jds += jm * s_sim[r] * w;
        }
      }
    }
    if (gcur >= 0) {
      s_part[((size_t)(0 * Gn + gcur) * rpp + rl) * q + c] += jde;
      s_part[((size_t)(1 * Gn + gcur) * rpp + rl) * q + c] += jds;
    }
  }
  __syncthreads();
  // dB_g/dtheta_c = jde/sds - 2 sde jds / sds^2   (linear_scale_factor.py:33-42)
  for (int i = tid; i < G * q; i += nthr) {
    const int g = i / q, c = i - g * q;
    double jde = 0.0, jds = 0.0;
    for (int l = 0; l < rpp; ++l) {
      jde += s_part[((size_t)(0 * Gn + g) * rpp + l) * q + c];
      jds += s_part[((size_t)(1 * Gn + g) * rpp + l) * q + c];
    }
    const double sds = s_sds[g], sde = s_sde[g];
    // square: dB = jde/sds - 2 sde jds/sds^2 ; log: dB = B * (-1/W) sum J/(s sigma'^2)
    const double db = (a.loss == SBM_LOSS_LOG_SQUARE) ? -s_B[g] * jde / sds
                                                      : jde / sds - 2.0 * sde * jds / (sds * sds);
    s_dB[i] = db;
    if (a.sf_grad) a.sf_grad[(size_t)v * G * q + i] = db;
  }
  __syncthreads();

  // pass B: J = B*Jm + sim (x) dB ; reference_compat: not divided by sigma, prior rows zero
  if (Jv) {
    for (size_t i = tid; i < (size_t)RT * q; i += blockDim.x) {
      const int r = (int)(i / q), c = (int)(i - (size_t)r * q);
      double val;
      if (r < R) {
        const int g = s_rsf[r];
        val = Jv[i];
        if (a.loss == SBM_LOSS_LOG_SQUARE && !(a.row_plain && a.row_plain[r])) {
          // log_squared_loss_function.py:66-98: J/s (+ (dB/dtheta)/B for rows with a scale factor)
          val = val / s_sim[r];
          if (g >= 0) val += s_dB[g * q + c] / s_B[g];
        } else if (g >= 0) {
          val = s_B[g] * val + s_sim[r] * s_dB[g * q + c];
        }
        if (!a.compat) val /= s_sg[r];
      } else if (r < R + a.NPR) {
        const int k = r - R;
        val = (!a.compat && a.prior_idx[k] == c) ? 1.0 / a.prior_sigma[k] : 0.0;
      } else {
        const int k = r - R - a.NPR;  // (dB/dtheta)/B, linear_scale_factor.py:44-53
        const int g = a.sfp_group[k];
        val = s_dB[g * q + c] / s_B[g];
        if (!a.compat) val /= a.sfp_sigma[k];
      }
      Jv[i] = val;
    }
  }
  // gradient of 0.5*sum r^2: (J^T r) with the Jacobian as returned (reference :803-805)
  if (a.grad) {
    __syncthreads();
    for (int c = tid; c < q; c += blockDim.x) {
      double gsum = 0.0;
      if (Jv) {
        for (int r = 0; r < RT; ++r) {
          const double res = a.Rout[(size_t)v * RT + r];
          gsum += Jv[(size_t)r * q + c] * res;
        }
      }
      a.grad[(size_t)v * q + c] = gsum;
    }
  }
}

template <class T>
static int upload(DevBuf<T>& b, const T* h, size_t n, hipStream_t s) {
  if (n == 0) n = 1;  // keep pointers non-NULL
  if (b.reserve(n)) return -1;
  if (h && hipMemcpyAsync(b.p, h, (n) * sizeof(T), hipMemcpyHostToDevice, s) != hipSuccess) return -1;
  return 0;
}

extern "C" int sbm_project_load(sbm_model* m, const sbm_project_desc* d, sbm_project** out) {
  if (!m || !d || !out) return sbm_fail(SBM_E_ARG, "sbm_project_load: NULL argument");
  const int E = d->n_experiments, q = d->n_project_params, R = d->n_rows, G = d->n_sf_groups,
            NPR = d->n_prior_rows, NSP = d->n_sf_prior_rows, NP = m->info.n_params;
  if (d->loss_type != SBM_LOSS_SQUARE && d->loss_type != SBM_LOSS_LOG_SQUARE)
    return sbm_fail(SBM_E_ARG, "sbm_project_load: unknown loss_type %d", d->loss_type);
  if (E <= 0 || q <= 0 || R < 0 || G < 0 || NPR < 0 || NSP < 0)
    return sbm_fail(SBM_E_ARG, "sbm_project_load: bad sizes E=%d q=%d R=%d G=%d", E, q, R, G);
  if (!d->pmap || !d->pfixed || !d->sens_col || !d->tgrid_off || !d->tgrid || (R && (!d->row_exp || !d->row_tidx ||
      !d->row_var_off || !d->row_vars || !d->row_data || !d->row_sigma || !d->row_sf)))
    return sbm_fail(SBM_E_ARG, "sbm_project_load: NULL array in descriptor");
  // validate on the host: the kernels index with these
  int n_t_max = 0;
  for (int e = 0; e < E; ++e) {
    const int len = d->tgrid_off[e + 1] - d->tgrid_off[e];
    if (len <= 0) return sbm_fail(SBM_E_ARG, "sbm_project_load: experiment %d has no output times", e);
    for (int k = d->tgrid_off[e] + 1; k < d->tgrid_off[e + 1]; ++k)
      if (!(d->tgrid[k] >= d->tgrid[k - 1])) return sbm_fail(SBM_E_ARG, "sbm_project_load: tgrid of experiment %d not sorted", e);
    if (d->tgrid[d->tgrid_off[e]] < 0.0) return sbm_fail(SBM_E_ARG, "sbm_project_load: negative output time");
    n_t_max = len > n_t_max ? len : n_t_max;
    for (int k = 0; k < NP; ++k) {
      const int g = d->pmap[e * NP + k];
      if (g >= q) return sbm_fail(SBM_E_ARG, "sbm_project_load: pmap[%d][%d]=%d >= q=%d", e, k, g, q);
    }
  }
  for (int k = 0; k < NP; ++k)
    if (d->sens_col[k] >= m->info.n_sens) return sbm_fail(SBM_E_ARG, "sbm_project_load: sens_col out of range");
  for (int r = 0; r < R; ++r) {
    const int e = d->row_exp[r];
    if (e < 0 || e >= E) return sbm_fail(SBM_E_ARG, "sbm_project_load: row %d experiment %d", r, e);
    const int len = d->tgrid_off[e + 1] - d->tgrid_off[e];
    if (d->row_tidx[r] < 0 || d->row_tidx[r] >= len) return sbm_fail(SBM_E_ARG, "sbm_project_load: row %d time index", r);
    if (d->row_var_off[r + 1] < d->row_var_off[r]) return sbm_fail(SBM_E_ARG, "sbm_project_load: row_var_off");
    for (int k = d->row_var_off[r]; k < d->row_var_off[r + 1]; ++k)
      if (d->row_vars[k] < 0 || d->row_vars[k] >= m->info.n_vars)
        return sbm_fail(SBM_E_ARG, "sbm_project_load: row %d maps to variable %d of %d", r, d->row_vars[k], m->info.n_vars);
    if (d->row_sf[r] >= G) return sbm_fail(SBM_E_ARG, "sbm_project_load: row %d sf group %d", r, d->row_sf[r]);
    if (d->row_sigma[r] == 0.0) return sbm_fail(SBM_E_ARG, "sbm_project_load: row %d has sigma 0", r);
    if (d->loss_type == SBM_LOSS_LOG_SQUARE && !(d->row_data[r] > 0.0))
      return sbm_fail(SBM_E_ARG, "sbm_project_load: LogSquare loss cannot handle measurements smaller or equal to zero (row %d)", r);
  }
  // custom observables: every program is checked here, once, so that the interpreter in the kernel can run unchecked
  const int NPG = d->n_programs;
  std::vector<int32_t> prog_base((size_t)(NPG > 0 ? NPG : 0) + 1, 0), row_w0((size_t)(R > 0 ? R : 1), -1);
  int n_weights = 0;
  if (NPG < 0) return sbm_fail(SBM_E_ARG, "sbm_project_load: n_programs < 0");
  if (NPG > 0) {
    if (!d->row_prog || !d->prog_nvars || !d->prog_sub_off || !d->prog_code || !d->row_time || d->n_prog_code <= 0 ||
        (d->n_prog_const > 0 && !d->prog_const))
      return sbm_fail(SBM_E_ARG, "sbm_project_load: NULL program table");
    for (int c = 0; c < NPG; ++c) {
      if (d->prog_nvars[c] < 0 || d->prog_nvars[c] > 64) return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d lists %d variables", c, d->prog_nvars[c]);
      prog_base[c + 1] = prog_base[c] + 1 + d->prog_nvars[c];
    }
    for (int c = 0; c < NPG; ++c) {
      for (int sidx = prog_base[c]; sidx < prog_base[c + 1]; ++sidx) {
        int pc = d->prog_sub_off[sidx], depth = 0;
        const int end = d->prog_sub_off[sidx + 1];
        if (pc < 0 || end > d->n_prog_code || pc >= end) return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d: bad subprogram bounds", c);
        bool closed = false;
        for (; pc < end && !closed; ++pc) {
          const int op = d->prog_code[pc];
          switch (op) {
            case SBM_OP_END: closed = true; break;
            case SBM_OP_VAR:
              if (++pc >= end || d->prog_code[pc] < 0 || d->prog_code[pc] >= d->prog_nvars[c])
                return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d: variable operand out of range", c);
              ++depth; break;
            case SBM_OP_CONST:
              if (++pc >= end || d->prog_code[pc] < 0 || d->prog_code[pc] >= d->n_prog_const)
                return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d: constant operand out of range", c);
              ++depth; break;
            case SBM_OP_TIME: ++depth; break;
            case SBM_OP_ADD: case SBM_OP_SUB: case SBM_OP_MUL: case SBM_OP_DIV: case SBM_OP_POW:
              if (depth < 2) return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d: stack underflow", c);
              --depth; break;
            case SBM_OP_POWI:
              if (++pc >= end || d->prog_code[pc] < -64 || d->prog_code[pc] > 64)
                return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d: integer power out of range", c);
              if (depth < 1) return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d: stack underflow", c);
              break;
            case SBM_OP_NEG: case SBM_OP_EXP: case SBM_OP_LOG: case SBM_OP_SQRT: case SBM_OP_TANH: case SBM_OP_SIN:
            case SBM_OP_COS: case SBM_OP_ABS: case SBM_OP_SIGN:
              if (depth < 1) return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d: stack underflow", c);
              break;
            default: return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d: unknown opcode %d", c, op);
          }
          if (depth > SBM_PROG_MAX_STACK) return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d needs a stack deeper than %d", c, SBM_PROG_MAX_STACK);
        }
        if (!closed || depth != 1) return sbm_fail(SBM_E_ARG, "sbm_project_load: program %d: subprogram does not leave exactly one value", c);
      }
    }
    for (int r = 0; r < R; ++r) {
      const int c = d->row_prog[r];
      if (c >= NPG) return sbm_fail(SBM_E_ARG, "sbm_project_load: row %d program %d of %d", r, c, NPG);
      if (c < 0) continue;
      if (d->row_var_off[r + 1] - d->row_var_off[r] != d->prog_nvars[c])
        return sbm_fail(SBM_E_ARG, "sbm_project_load: row %d lists %d variables, its program %d takes %d", r,
                        d->row_var_off[r + 1] - d->row_var_off[r], c, d->prog_nvars[c]);
      row_w0[r] = n_weights;
      n_weights += d->prog_nvars[c];
    }
  }
  for (int k = 0; k < NPR; ++k)
    if (d->prior_idx[k] < 0 || d->prior_idx[k] >= q) return sbm_fail(SBM_E_ARG, "sbm_project_load: prior index");
  for (int k = 0; k < NSP; ++k)
    if (d->sf_prior_group[k] < 0 || d->sf_prior_group[k] >= G) return sbm_fail(SBM_E_ARG, "sbm_project_load: sf prior group");

  SBM_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->stream;
  sbm_project* p = new sbm_project();
  p->model = m;
  p->E = E; p->q = q; p->R = R; p->G = G; p->NPR = NPR; p->NSP = NSP; p->compat = d->reference_compat;
  p->loss = d->loss_type;
  p->n_params = NP; p->n_vars = m->info.n_vars; p->n_sens = m->info.n_sens; p->n_t_max = n_t_max;

  // inverse map: for experiment e and project column c, the model params that read it
  std::vector<int32_t> inv_ptr((size_t)E * (q + 1), 0), inv_m;
  for (int e = 0; e < E; ++e) {
    for (int c = 0; c < q; ++c) {
      inv_ptr[(size_t)e * (q + 1) + c] = (int32_t)inv_m.size();
      for (int k = 0; k < NP; ++k)
        if (d->pmap[e * NP + k] == c) inv_m.push_back(k);
    }
    inv_ptr[(size_t)e * (q + 1) + q] = (int32_t)inv_m.size();
  }
  std::vector<int32_t> glen(E);
  for (int e = 0; e < E; ++e) glen[e] = d->tgrid_off[e + 1] - d->tgrid_off[e];
  const int nvars_total = R ? d->row_var_off[R] : 0;
  int bad = 0;
  bad |= upload(p->pmap, d->pmap, (size_t)E * NP, s);
  bad |= upload(p->pfixed, d->pfixed, (size_t)E * NP, s);
  bad |= upload(p->sens_col, d->sens_col, (size_t)NP, s);
  bad |= upload(p->tgrid_off, d->tgrid_off, (size_t)E + 1, s);
  bad |= upload(p->tgrid, d->tgrid, (size_t)d->tgrid_off[E], s);
  bad |= upload(p->grid_len, glen.data(), (size_t)E, s);
  bad |= upload(p->row_exp, d->row_exp, (size_t)R, s);
  bad |= upload(p->row_tidx, d->row_tidx, (size_t)R, s);
  bad |= upload(p->row_var_off, d->row_var_off, (size_t)R + 1, s);
  bad |= upload(p->row_vars, d->row_vars, (size_t)nvars_total, s);
  bad |= upload(p->row_data, d->row_data, (size_t)R, s);
  bad |= upload(p->row_sigma, d->row_sigma, (size_t)R, s);
  bad |= upload(p->row_sf, d->row_sf, (size_t)R, s);
  bad |= upload(p->prior_idx, d->prior_idx, (size_t)NPR, s);
  bad |= upload(p->prior_mean, d->prior_mean, (size_t)NPR, s);
  bad |= upload(p->prior_sigma, d->prior_sigma, (size_t)NPR, s);
  bad |= upload(p->sfp_group, d->sf_prior_group, (size_t)NSP, s);
  bad |= upload(p->sfp_mean, d->sf_prior_mean, (size_t)NSP, s);
  bad |= upload(p->sfp_sigma, d->sf_prior_sigma, (size_t)NSP, s);
  bad |= upload(p->inv_ptr, inv_ptr.data(), inv_ptr.size(), s);
  bad |= upload(p->inv_m, inv_m.data(), inv_m.size(), s);
  p->n_programs = NPG;
  p->n_custom_weights = n_weights;
  if (NPG > 0) {
    bad |= upload(p->row_prog, d->row_prog, (size_t)R, s);
    bad |= upload(p->row_w0, row_w0.data(), (size_t)R, s);
    bad |= upload(p->prog_base, prog_base.data(), prog_base.size(), s);
    bad |= upload(p->prog_sub_off, d->prog_sub_off, (size_t)prog_base[NPG] + 1, s);
    bad |= upload(p->prog_code, d->prog_code, (size_t)d->n_prog_code, s);
    bad |= upload(p->prog_const, d->prog_const, (size_t)d->n_prog_const, s);
    bad |= upload(p->row_time, d->row_time, (size_t)R, s);
  }
  hipError_t e = hipStreamSynchronize(s);  // host vectors go out of scope
  if (bad || e != hipSuccess) {
    sbm_project_unload(p);
    return sbm_fail(SBM_E_HIP, "sbm_project_load: device upload failed");
  }
  *out = p;
  return 0;
}

extern "C" int sbm_project_unload(sbm_project* p) {
  if (!p) return 0;
  (void)hipSetDevice(p->model->ctx->device);
  p->pmap.release(); p->sens_col.release(); p->tgrid_off.release(); p->grid_len.release(); p->row_exp.release();
  p->row_tidx.release(); p->row_var_off.release(); p->row_vars.release(); p->row_sf.release(); p->prior_idx.release();
  p->inv_ptr.release(); p->inv_m.release(); p->pfixed.release(); p->tgrid.release(); p->row_data.release();
  p->row_sigma.release(); p->prior_mean.release(); p->prior_sigma.release();
  p->sfp_group.release(); p->sfp_mean.release(); p->sfp_sigma.release();
  p->row_prog.release(); p->row_w0.release(); p->prog_base.release(); p->prog_sub_off.release(); p->prog_code.release();
  p->prog_const.release(); p->row_time.release();
  p->P.release(); p->Y.release(); p->S.release(); p->sims.release(); p->sf.release();
  p->traj_status.release(); p->traj_steps.release(); p->traj_rej.release(); p->goff.release(); p->glen.release();
  for (int l = 0; l < 2; ++l) { p->Yx[l].release(); p->Sx[l].release(); }
  p->statx.release(); p->stepsx.release();
  delete p;
  return 0;
}

extern "C" int64_t sbm_project_scratch_bytes(const sbm_project* p, int32_t V, int32_t with_sens) {
  if (!p || V < 0) return -1;
  const int64_t T = (int64_t)V * p->E;
  int64_t b = T * p->n_params * 8 + T * p->n_t_max * p->n_vars * 8 + T * 5 * 4 + (int64_t)V * (p->R + p->G) * 8;
  if (with_sens) b += T * p->n_t_max * p->n_vars * (int64_t)p->n_sens * 8;
  return b;
}

// Richardson combination of runs with step sizes h, h/2 (, h/4) of a symmetric one-step method (error
// expansion in h^2):  levels = 1: (4 b - a) / 3;  levels = 2: (16 (4 c - b)/3 - (4 b - a)/3) / 15.  In place in a.
__global__ void k_richardson(double* __restrict__ a, const double* __restrict__ b, const double* __restrict__ c,
                             size_t n, int levels) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double t1 = (4.0 * b[i] - a[i]) * (1.0 / 3.0);
  if (levels == 1) { a[i] = t1; return; }
  const double t2 = (4.0 * c[i] - b[i]) * (1.0 / 3.0);
  a[i] = (16.0 * t2 - t1) * (1.0 / 15.0);
}
__global__ void k_merge_status(int32_t* __restrict__ st, int32_t* __restrict__ steps, const int32_t* __restrict__ st2,
                               const int32_t* __restrict__ steps2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  st[i] = max(st[i], st2[i]);
  steps[i] += steps2[i];
}

extern "C" int sbm_project_set_extrapolation(sbm_project* p, int32_t levels) {
  if (!p) return sbm_fail(SBM_E_ARG, "sbm_project_set_extrapolation: project is NULL");
  if (levels < 0 || levels > 2) return sbm_fail(SBM_E_ARG, "sbm_project_set_extrapolation: levels %d (0, 1 or 2)", levels);
  p->richardson = levels;
  return 0;
}

// accepted steps of every trajectory of the project's last evaluation, [V][E] (vector-major), device to device
extern "C" int sbm_project_trajectory_steps(sbm_project* p, int32_t V, int32_t* steps_dev) {
  if (!p || !steps_dev) return sbm_fail(SBM_E_ARG, "sbm_project_trajectory_steps: NULL argument");
  if (V < 0) return sbm_fail(SBM_E_ARG, "sbm_project_trajectory_steps: V < 0");
  const size_t T = (size_t)V * p->E;
  if (T == 0) return 0;
  // the counts of the LAST evaluation, whatever the buffer could hold: a V other than that evaluation's would hand back
  // stale or uninitialised counts with a success code (round 3 checked the capacity only)
  if (!p->traj_steps.p || p->traj_steps.n < T || p->scratch_V != V)
    return sbm_fail(SBM_E_ARG, "sbm_project_trajectory_steps: V = %d, but the project's last evaluation had %d vectors", V,
                    p->traj_steps.p ? p->scratch_V : 0);
  SBM_HIP(hipSetDevice(p->model->ctx->device));
  SBM_HIP(hipMemcpyAsync(steps_dev, p->traj_steps.p, T * sizeof(int32_t), hipMemcpyDeviceToDevice, p->model->ctx->stream));
  return 0;
}

static int launch_assemble(const AssembleArgs& g, int V, hipStream_t s, const char* who) {
  const int Gn = g.G > 0 ? g.G : 1;
  const int cw_ = g.q < 256 ? g.q : 256, rpp_ = 256 / cw_;
  const size_t lds = sizeof(double) * ((size_t)2 * g.R + 3 * Gn + 4 + (size_t)Gn * g.q + (size_t)2 * Gn * rpp_ * g.q) +
                     sizeof(double) * 2 * (size_t)g.R + sizeof(int) * (7 * (size_t)g.R + 2) +   // + the staged row tables
                     sizeof(double) * (size_t)g.NW;                                               // + custom-row weights
  if (lds > 160 * 1024) return sbm_fail(SBM_E_ARG, "%s: project too large for the assembly kernel (%zu B of LDS)", who, lds);
  if (lds > 64 * 1024) SBM_HIP(hipFuncSetAttribute((const void*)k_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_assemble, dim3(V), dim3(256), lds, s, g);
  SBM_HIP(hipGetLastError());
  return 0;
}

static int project_run(sbm_project* p, const double* Theta, int V, const sbm_integrator_opts* opts, bool sens,
                       double* sims, double* Rout, double* J, double* Jmodel, double* sf, double* sf_grad, double* norms,
                       double* grad, int32_t* status, int32_t* n_steps, const char* who) {
  if (!p || !Theta || !Rout) return sbm_fail(SBM_E_ARG, "%s: NULL argument", who);
  if (V < 0) return sbm_fail(SBM_E_ARG, "%s: V < 0", who);
  int rc = check_opts(opts, who);
  if (rc) return rc;
  if (V == 0) return 0;
  sbm_model* m = p->model;
  SBM_HIP(hipSetDevice(m->ctx->device));
  hipStream_t s = m->ctx->stream;
  const size_t T = (size_t)V * p->E;
  const int NV = p->n_vars, NK = p->n_sens, NP = p->n_params, nt = p->n_t_max;
  // scratch reallocation must not race with kernels of an earlier call still using it
  const bool grow = T * NP > p->P.n || T * nt * NV > p->Y.n || (sens && T * nt * NV * NK > p->S.n) ||
                    T > p->traj_status.n || (size_t)V * p->R > p->sims.n;
  if (grow) SBM_HIP(hipStreamSynchronize(s));
  if (p->P.reserve(T * NP) || p->Y.reserve(T * nt * NV) || (sens && p->S.reserve(T * nt * NV * NK)) ||
      p->traj_status.reserve(T) || p->traj_steps.reserve(T) || p->traj_rej.reserve(T) || p->goff.reserve(T) ||
      p->glen.reserve(T) || p->sims.reserve((size_t)V * (p->R > 0 ? p->R : 1)))
    return sbm_fail(SBM_E_HIP, "%s: out of device memory for %d vectors (%lld bytes of scratch)", who, V,
                    (long long)sbm_project_scratch_bytes(p, V, sens));
  if (p->scratch_V != V) {
    hipLaunchKernelGGL(k_fill_grids, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, p->tgrid_off.p, V, p->E,
                       p->goff.p, p->glen.p);
    p->scratch_V = V;
  }
  {
    const size_t total = T * NP;
    hipLaunchKernelGGL(k_gather_params, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Theta, p->pmap.p,
                       p->pfixed.p, V, p->E, NP, p->q, p->P.p);
  }
  sbm_kernel_args a;
  memset(&a, 0, sizeof(a));
  a.P = p->P.p; a.t_out = p->tgrid.p; a.grid_off = p->goff.p; a.grid_len = p->glen.p;
  a.Y = p->Y.p; a.S = sens ? p->S.p : nullptr;
  a.status = p->traj_status.p; a.n_steps = p->traj_steps.p; a.n_reject = p->traj_rej.p;
  a.n_traj = (int32_t)T; a.n_t = nt; a.opts = *opts;
  rc = sens ? launch_sens(m, a, who) : launch(m, SBM_KIND_STATE, a, who);
  if (rc) return rc;
  const int levels = (opts->method == SBM_IMPLICIT_MIDPOINT || opts->method == SBM_IMPLICIT_MIDPOINT_GRADED) ? p->richardson : 0;
  if (levels > 0) {
    // the same ensemble again with every step halved (and halved again), then the combination in place
    const size_t nY = T * nt * NV, nS = T * nt * NV * NK;
    const bool grow2 = nY > p->Yx[0].n || (levels > 1 && nY > p->Yx[1].n) || (sens && (nS > p->Sx[0].n || (levels > 1 && nS > p->Sx[1].n))) ||
                       T > p->statx.n;
    if (grow2) SBM_HIP(hipStreamSynchronize(s));
    if (p->statx.reserve(T) || p->stepsx.reserve(T)) return sbm_fail(SBM_E_HIP, "%s: out of device memory", who);
    const int mult0 = opts->step_mult > 0 ? opts->step_mult : 1;
    for (int l = 0; l < levels; ++l) {
      if (p->Yx[l].reserve(nY) || (sens && p->Sx[l].reserve(nS))) return sbm_fail(SBM_E_HIP, "%s: out of device memory", who);
      sbm_kernel_args b = a;
      b.Y = p->Yx[l].p; b.S = sens ? p->Sx[l].p : nullptr;
      b.status = p->statx.p; b.n_steps = p->stepsx.p; b.n_reject = nullptr;
      b.opts.step_mult = mult0 << (l + 1);
      rc = sens ? launch_sens(m, b, who) : launch(m, SBM_KIND_STATE, b, who);
      if (rc) return rc;
      hipLaunchKernelGGL(k_merge_status, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, p->traj_status.p,
                         p->traj_steps.p, p->statx.p, p->stepsx.p, (int)T);
    }
    hipLaunchKernelGGL(k_richardson, dim3((unsigned)((nY + 255) / 256)), dim3(256), 0, s, p->Y.p, p->Yx[0].p,
                       levels > 1 ? p->Yx[1].p : nullptr, nY, levels);
    if (sens)
      hipLaunchKernelGGL(k_richardson, dim3((unsigned)((nS + 255) / 256)), dim3(256), 0, s, p->S.p, p->Sx[0].p,
                         levels > 1 ? p->Sx[1].p : nullptr, nS, levels);
    SBM_HIP(hipGetLastError());
  }

  AssembleArgs g;
  memset(&g, 0, sizeof(g));
  g.E = p->E; g.q = p->q; g.R = p->R; g.G = p->G; g.NPR = p->NPR; g.NSP = p->NSP; g.NP = NP; g.NV = NV; g.NK = NK; g.n_t = nt;
  g.compat = p->compat;
  g.loss = p->loss;
  g.row_exp = p->row_exp.p; g.row_tidx = p->row_tidx.p; g.row_var_off = p->row_var_off.p; g.row_vars = p->row_vars.p;
  g.row_sf = p->row_sf.p; g.prior_idx = p->prior_idx.p; g.row_data = p->row_data.p; g.row_sigma = p->row_sigma.p;
  g.prior_mean = p->prior_mean.p; g.prior_sigma = p->prior_sigma.p;
  g.sfp_group = p->sfp_group.p; g.sfp_mean = p->sfp_mean.p; g.sfp_sigma = p->sfp_sigma.p; g.inv_ptr = p->inv_ptr.p; g.inv_m = p->inv_m.p;
  g.sens_col = p->sens_col.p;
  if (p->n_programs > 0) {
    g.NW = p->n_custom_weights;
    g.row_prog = p->row_prog.p; g.row_w0 = p->row_w0.p; g.prog_base = p->prog_base.p; g.prog_sub_off = p->prog_sub_off.p;
    g.prog_code = p->prog_code.p; g.prog_const = p->prog_const.p; g.row_time = p->row_time.p;
  }
  g.Theta = Theta; g.Y = p->Y.p; g.S = sens ? p->S.p : nullptr;
  g.traj_status = p->traj_status.p; g.traj_steps = p->traj_steps.p;
  g.sims = sims ? sims : p->sims.p; g.Rout = Rout; g.sf = sf; g.norms = norms; g.status = status; g.n_steps = n_steps;
  g.J = sens ? J : nullptr; g.Jmodel = sens ? Jmodel : nullptr; g.grad = sens ? grad : nullptr;
  g.sf_grad = sens ? sf_grad : nullptr;
  return launch_assemble(g, V, s, who);
}

extern "C" int sbm_residuals_batch(sbm_project* p, const double* Theta, int32_t V, const sbm_integrator_opts* opts,
                                   double* sims, double* Rout, double* sf, double* norms, int32_t* status,
                                   int32_t* n_steps) {
  return project_run(p, Theta, V, opts, false, sims, Rout, nullptr, nullptr, sf, nullptr, norms, nullptr, status, n_steps,
                     "sbm_residuals_batch");
}

extern "C" int sbm_jacobian_batch(sbm_project* p, const double* Theta, int32_t V, const sbm_integrator_opts* opts,
                                  double* sims, double* Rout, double* J, double* Jmodel, double* sf, double* sf_grad,
                                  double* norms, double* grad, int32_t* status, int32_t* n_steps) {
  if (!J && !Jmodel) return sbm_fail(SBM_E_ARG, "sbm_jacobian_batch: J and Jmodel both NULL");
  if (grad && !J) return sbm_fail(SBM_E_ARG, "sbm_jacobian_batch: grad needs J");
  return project_run(p, Theta, V, opts, true, sims, Rout, J, Jmodel, sf, sf_grad, norms, grad, status, n_steps,
                     "sbm_jacobian_batch");
}

// ---------------------------------------------------------------------------------------------
// Loss functions on caller-supplied simulations (the reference's frame-level API, a13):
// the same assembly kernel, reading sims / the model Jacobian from the caller instead of Y / S.
// ---------------------------------------------------------------------------------------------
extern "C" int sbm_loss_eval_host(sbm_ctx* ctx, const sbm_loss_desc* d, int32_t V, const double* sims, const double* Jm,
                                  double* Rout, double* J, double* sf, double* sf_grad, double* norms, int32_t* status) {
  const char* who = "sbm_loss_eval_host";
  if (!ctx || !d || !sims || !Rout) return sbm_fail(SBM_E_ARG, "%s: NULL argument", who);
  const int R = d->n_rows, q = d->n_params, G = d->n_sf_groups, NSP = d->n_sf_prior_rows;
  if (V < 0 || R <= 0 || q < 0 || G < 0 || NSP < 0) return sbm_fail(SBM_E_ARG, "%s: bad sizes V=%d R=%d q=%d G=%d", who, V, R, q, G);
  if (d->loss_type != SBM_LOSS_SQUARE && d->loss_type != SBM_LOSS_LOG_SQUARE)
    return sbm_fail(SBM_E_ARG, "%s: unknown loss_type %d", who, d->loss_type);
  if (!d->row_data || !d->row_sigma || !d->row_sf) return sbm_fail(SBM_E_ARG, "%s: NULL array in descriptor", who);
  if ((J || sf_grad) && (!Jm || q <= 0)) return sbm_fail(SBM_E_ARG, "%s: J / sf_grad need the model Jacobian", who);
  if (NSP && (!d->sf_prior_group || !d->sf_prior_mean || !d->sf_prior_sigma)) return sbm_fail(SBM_E_ARG, "%s: NULL sf prior array", who);
  for (int r = 0; r < R; ++r) {
    if (d->row_sf[r] >= G) return sbm_fail(SBM_E_ARG, "%s: row %d sf group %d", who, r, d->row_sf[r]);
    const bool plain = d->row_plain && d->row_plain[r];
    if (plain && d->row_sf[r] >= 0) return sbm_fail(SBM_E_ARG, "%s: plain row %d cannot carry a scale factor", who, r);
    if (d->loss_type == SBM_LOSS_LOG_SQUARE && !plain && !(d->row_data[r] > 0.0))
      return sbm_fail(SBM_E_ARG, "%s: LogSquare loss cannot handle measurements smaller or equal to zero (row %d)", who, r);
  }
  for (int k = 0; k < NSP; ++k)
    if (d->sf_prior_group[k] < 0 || d->sf_prior_group[k] >= G) return sbm_fail(SBM_E_ARG, "%s: sf prior group", who);
  if (V == 0) return 0;
  SBM_HIP(hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  const int RT = R + NSP, qq = q > 0 ? q : 1;
  DevBuf<double> b_data, b_sigma, b_spm, b_sps, b_sims, b_Jm, b_R, b_J, b_sf, b_sfg, b_norms;
  DevBuf<int32_t> b_sfi, b_plain, b_spg, b_status;
  int bad = 0;
  bad |= upload(b_data, d->row_data, (size_t)R, s);
  bad |= upload(b_sigma, d->row_sigma, (size_t)R, s);
  bad |= upload(b_sfi, d->row_sf, (size_t)R, s);
  if (d->row_plain) bad |= upload(b_plain, d->row_plain, (size_t)R, s);
  bad |= upload(b_spg, d->sf_prior_group, (size_t)NSP, s);
  bad |= upload(b_spm, d->sf_prior_mean, (size_t)NSP, s);
  bad |= upload(b_sps, d->sf_prior_sigma, (size_t)NSP, s);
  bad |= upload(b_sims, sims, (size_t)V * R, s);
  if (Jm) bad |= upload(b_Jm, Jm, (size_t)V * R * q, s);
  bad |= b_R.reserve((size_t)V * RT) | b_sf.reserve((size_t)V * (G ? G : 1)) | b_norms.reserve(V) | b_status.reserve(V);
  if (J) bad |= b_J.reserve((size_t)V * RT * qq);
  if (Jm) bad |= b_sfg.reserve((size_t)V * (G ? G : 1) * qq);
  int rc = 0;
  if (bad) rc = sbm_fail(SBM_E_HIP, "%s: device allocation / upload failed", who);
  if (!rc) {
    AssembleArgs g;
    memset(&g, 0, sizeof(g));
    g.E = 1; g.q = qq; g.R = R; g.G = G; g.NPR = 0; g.NSP = NSP;
    g.compat = d->reference_compat; g.loss = d->loss_type;
    g.row_sf = b_sfi.p; g.row_data = b_data.p; g.row_sigma = b_sigma.p; g.row_plain = d->row_plain ? b_plain.p : nullptr;
    g.sfp_group = b_spg.p; g.sfp_mean = b_spm.p; g.sfp_sigma = b_sps.p;
    g.sims_in = b_sims.p; g.Jm_in = Jm ? b_Jm.p : nullptr;
    g.Rout = b_R.p; g.sf = b_sf.p; g.norms = b_norms.p; g.status = b_status.p;
    g.J = J ? b_J.p : nullptr; g.sf_grad = Jm ? b_sfg.p : nullptr;
    rc = launch_assemble(g, V, s, who);
  }
  hipError_t e = hipSuccess;
  if (!rc) {
    e = hipMemcpyAsync(Rout, b_R.p, (size_t)V * RT * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && J) e = hipMemcpyAsync(J, b_J.p, (size_t)V * RT * q * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && sf && G) e = hipMemcpyAsync(sf, b_sf.p, (size_t)V * G * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && sf_grad && G) e = hipMemcpyAsync(sf_grad, b_sfg.p, (size_t)V * G * q * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && norms) e = hipMemcpyAsync(norms, b_norms.p, (size_t)V * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && status) e = hipMemcpyAsync(status, b_status.p, (size_t)V * 4, hipMemcpyDeviceToHost, s);
  }
  hipError_t e2 = hipStreamSynchronize(s);
  b_data.release(); b_sigma.release(); b_spm.release(); b_sps.release(); b_sims.release(); b_Jm.release(); b_R.release();
  b_J.release(); b_sf.release(); b_sfg.release(); b_norms.release(); b_sfi.release(); b_plain.release(); b_spg.release();
  b_status.release();
  if (rc) return rc;
  if (e != hipSuccess || e2 != hipSuccess) return sbm_fail(SBM_E_HIP, "%s: %s", who, hipGetErrorString(e != hipSuccess ? e : e2));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Batched Levenberg-Marquardt step (the caller after the path: multi-start fitting).
// The reference fits with scipy.optimize.leastsq(project.residuals, x0, Dfun=project.calc_project_jacobian)
// (tests/test_Project.py:202-213, :352-357), one start at a time; here every parameter vector of an
// ensemble takes its own damped Gauss-Newton step:
//     (J^T J + lambda_v diag(J^T J)) delta_v = -J^T r_v          (Marquardt scaling)
// One 256-thread block per vector: J^T J and J^T r accumulated from row tiles staged in LDS,
// Cholesky and the two triangular solves in LDS.  q <= 128.
// ---------------------------------------------------------------------------------------------
struct LmArgs {
  const double* J;       // [V][M][q]
  const double* r;       // [V][M]
  const double* lambda;  // [V]
  double* delta;         // [V][q]
  double* pred;          // [V] predicted decrease of 0.5 |r|^2: -g.delta - 0.5 delta^T (J^T J) delta
  int32_t* status;       // [V] 0 ok, 1 not positive definite / non-finite input (delta = 0)
  int M, q, tile;
};

constexpr int LM_TILE = 32;   // rows of J staged per pass (fewer when the q x q matrix leaves less room: lm_lds_bytes)

// LDS of the normal-equation kernels: the q x (q + 1) matrix, `nvec` vectors of q, a row tile of J with the residuals
// beside it.  The tile shrinks (32, 16, 8 rows) until the total fits what the device gives a workgroup; 0 = no fit.
static int lm_lds_limit(sbm_ctx* ctx) {
  int lim = 0;
  if (hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx->device) != hipSuccess || lim <= 0) lim = 64 * 1024;
  return lim;
}
static size_t lm_lds_bytes(sbm_ctx* ctx, int q, int nvec, int* tile_out) {
  const size_t ld = (size_t)q + 1;
  const size_t limit = (size_t)lm_lds_limit(ctx);
  for (int tile = LM_TILE; tile >= 8; tile /= 2) {
    const size_t b = sizeof(double) * ((size_t)q * ld + (size_t)nvec * q + (size_t)tile * ld + tile);
    if (b <= limit) { *tile_out = tile; return b; }
  }
  return 0;
}

__global__ void __launch_bounds__(256) k_lm_step(LmArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lm_smem[];
  const int v = blockIdx.x, tid = threadIdx.x, q = a.q, M = a.M;
  const int ld = q + 1;                      // padded leading dimension of the q x q matrices
  double* A = lm_smem;                       // [q][ld]  J^T J, then its Cholesky factor (lower)
  double* dg = A + (size_t)q * ld;           // [q]      diag(J^T J)
  double* g = dg + q;                        // [q]      J^T r
  double* x = g + q;                         // [q]      right-hand side, then the solution
  const int TILE = a.tile;
  double* T = x + q;                         // [TILE][ld] row tile of J
  double* rt = T + (size_t)TILE * ld;        // [TILE]
  __shared__ int s_bad;
  if (tid == 0) s_bad = 0;
  const double* Jv = a.J + (size_t)v * M * q;
  const double* rv = a.r + (size_t)v * M;
  // each thread owns the entries e = tid, tid + 256, ... of the lower triangle (i >= j) and of g
  const int n_low = q * (q + 1) / 2;
  constexpr int MAXOWN = (128 * 129 / 2 + 255) / 256;
  double acc[MAXOWN];
  int oi[MAXOWN], oj[MAXOWN];
  int n_own = 0;
  for (int e = tid; e < n_low; e += 256) {
    // row i with i(i+1)/2 <= e
    int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= e) ++i;
    while (i * (i + 1) / 2 > e) --i;
    oi[n_own] = i; oj[n_own] = e - i * (i + 1) / 2; acc[n_own] = 0.0; ++n_own;
  }
  double gacc = 0.0;   // thread c < q owns g[c]
  for (int m0 = 0; m0 < M; m0 += TILE) {
    const int rows = min(TILE, M - m0);
    __syncthreads();
    for (int e = tid; e < rows * q; e += 256) {
      const int rr = e / q, c = e - rr * q;
      const double val = Jv[(size_t)(m0 + rr) * q + c];
      T[rr * ld + c] = val;
      if (!(fabs(val) < 1.0e300)) s_bad = 1;
    }
    for (int e = tid; e < rows; e += 256) {
      const double val = rv[m0 + e];
      rt[e] = val;
      if (!(fabs(val) < 1.0e300)) s_bad = 1;
    }
    __syncthreads();
    for (int k = 0; k < n_own; ++k) {
      double s = acc[k];
      for (int rr = 0; rr < rows; ++rr) s = fma(T[rr * ld + oi[k]], T[rr * ld + oj[k]], s);
      acc[k] = s;
    }
    if (tid < q) {
      double s = gacc;
      for (int rr = 0; rr < rows; ++rr) s = fma(T[rr * ld + tid], rt[rr], s);
      gacc = s;
    }
  }
  __syncthreads();
  for (int k = 0; k < n_own; ++k) A[oi[k] * ld + oj[k]] = acc[k];
  if (tid < q) g[tid] = gacc;
  __syncthreads();
  const double lam = a.lambda[v];
  if (tid < q) {
    const double d = A[tid * ld + tid];
    dg[tid] = d;
    // Marquardt scaling; a column J never touches (d = 0) gets a unit pivot: delta_c = 0
    A[tid * ld + tid] = d > 0.0 ? d * (1.0 + lam) : 1.0;
    x[tid] = -g[tid];
  }
  __syncthreads();
  bool bad = s_bad != 0 || !(lam >= 0.0);
  // right-looking Cholesky, lower triangle in place
  for (int k = 0; k < q && !bad; ++k) {
    const double piv = A[k * ld + k];
    if (!(piv > 0.0) || !(piv < 1.0e300)) { bad = true; break; }   // uniform: every thread reads the same value
    const double rp = 1.0 / sqrt(piv);
    __syncthreads();
    if (tid == 0) A[k * ld + k] = sqrt(piv);
    for (int i = k + 1 + tid; i < q; i += 256) A[i * ld + k] *= rp;
    __syncthreads();
    // trailing update: entries (i, j), k < j <= i
    const int nt = q - k - 1;
    for (int e = tid; e < nt * nt; e += 256) {
      const int i = k + 1 + e / nt, j = k + 1 + e % nt;
      if (j <= i) A[i * ld + j] = fma(-A[i * ld + k], A[j * ld + k], A[i * ld + j]);
    }
    __syncthreads();
  }
  if (bad) {
    for (int c = tid; c < q; c += 256) a.delta[(size_t)v * q + c] = 0.0;
    if (tid == 0) { a.pred[v] = 0.0; a.status[v] = 1; }
    return;
  }
  // L y = -g, then L^T x = y, column-oriented: one thread finishes x_i, all threads retire it from the
  // remaining right-hand sides (two barriers per column instead of a serial O(q^2) chain on one thread)
  for (int i = 0; i < q; ++i) {
    if (tid == 0) x[i] /= A[i * ld + i];
    __syncthreads();
    const double xi = x[i];
    for (int j = i + 1 + tid; j < q; j += 256) x[j] = fma(-A[j * ld + i], xi, x[j]);
    __syncthreads();
  }
  for (int i = q - 1; i >= 0; --i) {
    if (tid == 0) x[i] /= A[i * ld + i];
    __syncthreads();
    const double xi = x[i];
    for (int j = tid; j < i; j += 256) x[j] = fma(-A[i * ld + j], xi, x[j]);
    __syncthreads();
  }
  // predicted decrease of 0.5 |r|^2 under the Gauss-Newton model, H = J^T J:  -g.d - 0.5 d^T H d  with
  // (H + lambda D) d = -g  =>  d^T H d = -g.d - lambda sum D_i d_i^2   (D = diag(H), or 1 where it is 0)
  double part = 0.0;
  if (tid < q) {
    a.delta[(size_t)v * q + tid] = x[tid];
    const double Di = dg[tid] > 0.0 ? dg[tid] : 0.0;
    part = -0.5 * g[tid] * x[tid] + 0.5 * lam * Di * x[tid] * x[tid];
  }
  __shared__ double s_red[4];
  const double tot = block_sum(part, s_red);
  if (tid == 0) { a.pred[v] = tot; a.status[v] = 0; }
}

extern "C" int sbm_lm_step(sbm_ctx* ctx, const double* J, const double* r, const double* lambda, int32_t V, int32_t M,
                           int32_t q, double* delta, double* pred, int32_t* status) {
  if (!ctx || !J || !r || !lambda || !delta || !pred || !status) return sbm_fail(SBM_E_ARG, "sbm_lm_step: NULL argument");
  if (V < 0 || M <= 0 || q <= 0 || q > 128) return sbm_fail(SBM_E_ARG, "sbm_lm_step: bad sizes V=%d M=%d q=%d (q <= 128)", V, M, q);
  if (V == 0) return 0;
  SBM_HIP(hipSetDevice(ctx->device));
  const size_t ld = (size_t)q + 1;
  int tile = 0;
  const size_t lds = lm_lds_bytes(ctx, q, 3, &tile);
  if (!lds) return sbm_fail(SBM_E_ARG, "sbm_lm_step: q = %d does not fit the %d KB of LDS of a workgroup", q, lm_lds_limit(ctx) / 1024);
  LmArgs a{J, r, lambda, delta, pred, status, M, q, tile};
  (void)ld;
  if (lds > 64 * 1024) SBM_HIP(hipFuncSetAttribute((const void*)k_lm_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_lm_step, dim3(V), dim3(256), lds, ctx->stream, a);
  SBM_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// sbm_lm_trust_step: the Levenberg-Marquardt PARAMETER of a scaled trust region, per vector.
//
// What MINPACK's lmder does between two Jacobian evaluations (lmpar, More 1978), on the normal equations:
// given the scaling D (the largest column norm of J seen so far, kept by the caller from call to call) and a radius
// Delta, find lambda >= 0 with  (J^T J + lambda D^2) x = -J^T r  and  | ||D x|| - Delta | <= 0.1 Delta  (lambda = 0 if
// the Gauss-Newton step is already inside), by More's safeguarded Newton iteration on
// phi(lambda) = ||D x(lambda)|| - Delta:  lambda += (phi / Delta) / ||L^-1 D^2 x / ||D x||||^2  with L the Cholesky
// factor of the damped matrix, kept between the bounds the iteration itself produces.  At most 10 factorisations of a
// q x q matrix per call (two to three are the rule): microseconds, against the milliseconds of the integration that
// follows -- which is why the search for lambda happens here, in one launch, rather than as a sequence of trial
// INTEGRATIONS with lambda multiplied up and down (sbm_lm_step + project/fitting.py's 'marquardt' loop).
// One 256-thread block per vector; J^T J in registers (the lower triangle, spread over the threads), the matrix being
// factored in LDS.
// ---------------------------------------------------------------------------------------------
struct LmTrustArgs {
  const double* J;       // [V][M][q]
  const double* r;       // [V][M]
  double* dscale;        // [V][q]  in / out: D, made max(D, column norm of J) here (0 on the first call)
  const double* radius;  // [V]     Delta > 0
  double* lambda;        // [V]     in: the previous parameter (a starting guess), out: the one found
  double* delta;         // [V][q]  out: x
  double* pred;          // [V]     out: predicted decrease of 0.5 |r|^2 = -g.x - 0.5 x^T J^T J x
  double* dxnorm;        // [V]     out: ||D x||
  int32_t* status;       // [V]     out: 0, 1: non-finite input / no positive definite system found (x = 0), 2: skipped
  int M, q, tile;
  // extended entry point (sbm_lm_trust_step_ex); all nullable / 0
  const double* row_scale;   // [M]    J is used as diag(row_scale) J (reference_compat Jacobians: 1 / sigma)
  const int32_t* skip;       // [V]    != 0: leave the vector alone (x = 0, status 2)
  double max_step;           // > 0: every component of x is clipped to +-max_step; pred, dxnorm, gtx are those of the clipped step
  double* gtx;               // [V]    out: g . x (the directional derivative of 0.5 |r|^2 along the step)
  const double* theta;       // [V][q] with `trial`: trial = theta + x
  double* trial;             // [V][q]
};


// in-place Cholesky of the lower triangle of A (q x q, leading dimension ld) by the whole block; false if a pivot is
// not positive (the decision is uniform: every thread reads the same pivot)
__device__ __forceinline__ bool lm_cholesky(double* A, int q, int ld, int tid) {
  for (int k = 0; k < q; ++k) {
    const double piv = A[k * ld + k];
    if (!(piv > 0.0) || !(piv < 1.0e300)) return false;
    const double rp = 1.0 / sqrt(piv);
    __syncthreads();
    if (tid == 0) A[k * ld + k] = sqrt(piv);
    for (int i = k + 1 + tid; i < q; i += 256) A[i * ld + k] *= rp;
    __syncthreads();
    const int nt = q - k - 1;
    for (int e = tid; e < nt * nt; e += 256) {
      const int i = k + 1 + e / nt, j = k + 1 + e % nt;
      if (j <= i) A[i * ld + j] = fma(-A[i * ld + k], A[j * ld + k], A[i * ld + j]);
    }
    __syncthreads();
  }
  return true;
}
// x <- L^-1 x (column-oriented: one thread finishes x_i, all retire it from the rest)
__device__ __forceinline__ void lm_forward(const double* A, double* x, int q, int ld, int tid) {
  for (int i = 0; i < q; ++i) {
    if (tid == 0) x[i] /= A[i * ld + i];
    __syncthreads();
    const double xi = x[i];
    for (int j = i + 1 + tid; j < q; j += 256) x[j] = fma(-A[j * ld + i], xi, x[j]);
    __syncthreads();
  }
}
// x <- L^-T x
__device__ __forceinline__ void lm_backward(const double* A, double* x, int q, int ld, int tid) {
  for (int i = q - 1; i >= 0; --i) {
    if (tid == 0) x[i] /= A[i * ld + i];
    __syncthreads();
    const double xi = x[i];
    for (int j = tid; j < i; j += 256) x[j] = fma(-A[i * ld + j], xi, x[j]);
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) k_lm_trust(LmTrustArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lm_smem[];
  const int v = blockIdx.x, tid = threadIdx.x, q = a.q, M = a.M, TILE = a.tile;
  const int ld = q + 1;
  double* A = lm_smem;                       // [q][ld]  the damped matrix / its Cholesky factor
  double* D = A + (size_t)q * ld;            // [q]      scaling
  double* g = D + q;                         // [q]      J^T r
  double* x = g + q;                         // [q]      step
  double* w = x + q;                         // [q]      work vector of the Newton correction
  double* xg = w + q;                        // [q]      the last step that came out of a successful factorisation
  double* T = xg + q;                        // [TILE][ld] row tile of J
  double* rt = T + (size_t)TILE * ld;        // [TILE]
  __shared__ int s_bad;
  __shared__ double s_red[4];
  if (a.skip && a.skip[v]) {
    for (int c = tid; c < q; c += 256) {
      a.delta[(size_t)v * q + c] = 0.0;
      if (a.trial && a.theta) a.trial[(size_t)v * q + c] = a.theta[(size_t)v * q + c];
    }
    if (tid == 0) { a.pred[v] = 0.0; a.dxnorm[v] = 0.0; a.status[v] = 2; if (a.gtx) a.gtx[v] = 0.0; }
    return;
  }
  if (tid == 0) s_bad = 0;
  const double* Jv = a.J + (size_t)v * M * q;
  const double* rv = a.r + (size_t)v * M;
  const int n_low = q * (q + 1) / 2;
  constexpr int MAXOWN = (128 * 129 / 2 + 255) / 256;
  double acc[MAXOWN];
  int oi[MAXOWN], oj[MAXOWN];
  int n_own = 0;
  for (int e = tid; e < n_low; e += 256) {
    int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= e) ++i;
    while (i * (i + 1) / 2 > e) --i;
    oi[n_own] = i; oj[n_own] = e - i * (i + 1) / 2; acc[n_own] = 0.0; ++n_own;
  }
  double gacc = 0.0;
  for (int m0 = 0; m0 < M; m0 += TILE) {
    const int rows = min(TILE, M - m0);
    __syncthreads();
    for (int e = tid; e < rows * q; e += 256) {
      const int rr = e / q, c = e - rr * q;
      double val = Jv[(size_t)(m0 + rr) * q + c];
      if (a.row_scale) val *= a.row_scale[m0 + rr];
      T[rr * ld + c] = val;
      if (!(fabs(val) < 1.0e300)) s_bad = 1;
    }
    for (int e = tid; e < rows; e += 256) {
      const double val = rv[m0 + e];
      rt[e] = val;
      if (!(fabs(val) < 1.0e300)) s_bad = 1;
    }
    __syncthreads();
    for (int k = 0; k < n_own; ++k) {
      double sacc = acc[k];
      for (int rr = 0; rr < rows; ++rr) sacc = fma(T[rr * ld + oi[k]], T[rr * ld + oj[k]], sacc);
      acc[k] = sacc;
    }
    if (tid < q) {
      double sacc = gacc;
      for (int rr = 0; rr < rows; ++rr) sacc = fma(T[rr * ld + tid], rt[rr], sacc);
      gacc = sacc;
    }
  }
  __syncthreads();
  const double Delta = a.radius[v];
  double lam = a.lambda[v];
  // scaling: the largest column norm seen so far (MINPACK mode 1); a column J never touches gets 1
  for (int k = 0; k < n_own; ++k)
    if (oi[k] == oj[k]) {
      const double cn = sqrt(fmax(acc[k], 0.0));
      double d = fmax(a.dscale[(size_t)v * q + oi[k]], cn);
      if (!(d > 0.0)) d = 1.0;
      D[oi[k]] = d;
      a.dscale[(size_t)v * q + oi[k]] = d;
    }
  if (tid < q) g[tid] = gacc;
  __syncthreads();
  bool bad = s_bad != 0 || !(Delta > 0.0) || !(lam >= 0.0);
  // paru = || D^-1 g || / Delta: with that much damping the step is inside the region
  double part = 0.0;
  if (tid < q) { const double t = g[tid] / D[tid]; part = t * t; }
  const double gnorm = sqrt(block_sum(part, s_red));
  double paru = gnorm / Delta;
  if (!(paru > 0.0)) paru = 2.2e-308 / fmin(Delta, 0.1);
  double parl = 0.0, fp = 0.0, dxn = 0.0;
  bool have = false;
  double lam_good = 0.0, dxn_good = 0.0;      // ... of the last successful factorisation (its step is parked in xg)

  auto solve_with = [&](double par) -> bool {          // A <- chol(J^T J + par D^2); x <- -A^-1 g; dxn, fp
    for (int k = 0; k < n_own; ++k) {
      const int i = oi[k], j = oj[k];
      A[i * ld + j] = (i == j) ? fma(par * D[i], D[i], acc[k]) : acc[k];
    }
    if (tid < q) x[tid] = -g[tid];
    __syncthreads();
    if (!lm_cholesky(A, q, ld, tid)) return false;
    lm_forward(A, x, q, ld, tid);
    lm_backward(A, x, q, ld, tid);
    double p2 = 0.0;
    if (tid < q) { const double t = D[tid] * x[tid]; p2 = t * t; xg[tid] = x[tid]; }
    dxn = sqrt(block_sum(p2, s_red));
    fp = dxn - Delta;
    lam_good = par;
    dxn_good = dxn;
    return true;
  };
  auto newton_denominator = [&]() -> double {           // || L^-1 (D^2 x / dxn) ||^2 with the current factor
    if (tid < q) w[tid] = D[tid] * D[tid] * x[tid] / dxn;
    __syncthreads();
    lm_forward(A, w, q, ld, tid);
    double p2 = 0.0;
    if (tid < q) p2 = w[tid] * w[tid];
    return block_sum(p2, s_red);
  };

  if (!bad) {
    // the Gauss-Newton step, if J has full rank numerically
    if (solve_with(0.0)) {
      if (fp <= 0.1 * Delta) { lam = 0.0; have = true; }
      else { const double den = newton_denominator(); if (den > 0.0) parl = (fp / Delta) / den; }
    }
    if (!have) {
      bool any = false;                                // a damped system has been solved
      lam = fmin(fmax(lam, parl), paru);
      if (lam == 0.0) lam = (dxn > 0.0) ? gnorm / dxn : 1.0e-3 * paru;
      for (int it = 0; it < 10; ++it) {
        if (lam == 0.0) lam = fmax(2.2e-308, 1.0e-3 * paru);
        const double fp_old = fp;
        if (!solve_with(lam)) {                        // rounding: not positive definite at this damping yet
          // A is half factored and x holds -g: what counts from here on is the last step that WAS solved for (xg)
          parl = fmax(parl, lam);
          lam = fmax(10.0 * lam, 1.0e-3 * paru);
          if (lam > 1.0e3 * paru) break;               // (only non-finite data gets here)
          continue;
        }
        any = true;
        if (fabs(fp) <= 0.1 * Delta || (parl == 0.0 && fp <= fp_old && fp_old < 0.0) || it == 9) break;
        const double den = newton_denominator();
        const double parc = den > 0.0 ? (fp / Delta) / den : 0.0;
        if (fp > 0.0) parl = fmax(parl, lam);
        if (fp < 0.0) paru = fmin(paru, lam);
        lam = fmax(parl, lam + parc);
      }
      // the answer is the last DAMPED system that factored -- never the right-hand side a failed factorisation left
      // in x, nor the undamped step that was outside the region
      have = any && lam_good > 0.0;
      if (have) { lam = lam_good; dxn = dxn_good; }
      __syncthreads();
      if (have && tid < q) x[tid] = xg[tid];
      __syncthreads();
    }
  }
  if (bad || !have) {
    for (int c = tid; c < q; c += 256) {
      a.delta[(size_t)v * q + c] = 0.0;
      if (a.trial && a.theta) a.trial[(size_t)v * q + c] = a.theta[(size_t)v * q + c];
    }
    if (tid == 0) { a.pred[v] = 0.0; a.dxnorm[v] = 0.0; a.status[v] = 1; if (a.gtx) a.gtx[v] = 0.0; }
    return;
  }
  // a step bound per component (exp(theta) has to stay finite): clip, and report the quantities of the step TAKEN
  bool clipped = false;
  if (a.max_step > 0.0) {
    int cl = 0;
    if (tid < q && fabs(x[tid]) > a.max_step) { x[tid] = copysign(a.max_step, x[tid]); cl = 1; }
    clipped = __syncthreads_or(cl) != 0;
  }
  double gx = 0.0;
  if (tid < q) gx = g[tid] * x[tid];
  const double gtx = block_sum(gx, s_red);
  double tot;
  if (!clipped) {
    // predicted decrease of 0.5 |r|^2 under the Gauss-Newton model:  -g.x - 0.5 x^T H x  with  x^T H x = -g.x - lam dxn^2
    tot = -0.5 * gtx + 0.5 * lam * dxn * dxn;
  } else {
    double p2 = 0.0;
    if (tid < q) { const double t = D[tid] * x[tid]; p2 = t * t; }
    dxn = sqrt(block_sum(p2, s_red));
    double xhx = 0.0;                                  // x^T (J^T J) x from the lower triangle in registers
    for (int k = 0; k < n_own; ++k) xhx += (oi[k] == oj[k] ? 1.0 : 2.0) * acc[k] * x[oi[k]] * x[oj[k]];
    tot = -gtx - 0.5 * block_sum(xhx, s_red);
  }
  if (tid < q) {
    a.delta[(size_t)v * q + tid] = x[tid];
    if (a.trial && a.theta) a.trial[(size_t)v * q + tid] = a.theta[(size_t)v * q + tid] + x[tid];
  }
  if (tid == 0) { a.pred[v] = tot; a.dxnorm[v] = dxn; a.lambda[v] = lam; a.status[v] = 0; if (a.gtx) a.gtx[v] = gtx; }
}

extern "C" int sbm_lm_trust_step_ex(sbm_ctx* ctx, const double* J, const double* r, double* dscale, const double* radius,
                                    double* lambda, int32_t V, int32_t M, int32_t q, const double* row_scale,
                                    const int32_t* skip, double max_step, const double* theta, double* trial, double* delta,
                                    double* pred, double* dxnorm, double* gtx, int32_t* status) {
  if (!ctx || !J || !r || !dscale || !radius || !lambda || !delta || !pred || !dxnorm || !status)
    return sbm_fail(SBM_E_ARG, "sbm_lm_trust_step: NULL argument");
  if (V < 0 || M <= 0 || q <= 0 || q > 128) return sbm_fail(SBM_E_ARG, "sbm_lm_trust_step: bad sizes V=%d M=%d q=%d (q <= 128)", V, M, q);
  if ((theta == nullptr) != (trial == nullptr)) return sbm_fail(SBM_E_ARG, "sbm_lm_trust_step_ex: theta and trial go together");
  if (V == 0) return 0;
  SBM_HIP(hipSetDevice(ctx->device));
  int tile = 0;
  const size_t lds = lm_lds_bytes(ctx, q, 5, &tile);
  if (!lds) return sbm_fail(SBM_E_ARG, "sbm_lm_trust_step: q = %d does not fit the %d KB of LDS of a workgroup", q, lm_lds_limit(ctx) / 1024);
  LmTrustArgs a{J, r, dscale, radius, lambda, delta, pred, dxnorm, status, M, q, tile, row_scale, skip, max_step, gtx, theta, trial};
  if (lds > 64 * 1024) SBM_HIP(hipFuncSetAttribute((const void*)k_lm_trust, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_lm_trust, dim3(V), dim3(256), lds, ctx->stream, a);
  SBM_HIP(hipGetLastError());
  return 0;
}

extern "C" int sbm_lm_trust_step(sbm_ctx* ctx, const double* J, const double* r, double* dscale, const double* radius,
                                 double* lambda, int32_t V, int32_t M, int32_t q, double* delta, double* pred,
                                 double* dxnorm, int32_t* status) {
  return sbm_lm_trust_step_ex(ctx, J, r, dscale, radius, lambda, V, M, q, nullptr, nullptr, 0.0, nullptr, nullptr, delta, pred,
                              dxnorm, nullptr, status);
}

// ---------------------------------------------------------------------------------------------
// sbm_lm_update / sbm_lm_accept: lmder's bookkeeping between two trust-region steps, for V starts in two launches
// (round 2 spelled it as ~70 tensor selects per iteration: 68 000 micro-launches in a 100-iteration fit).
// ---------------------------------------------------------------------------------------------
struct LmUpdateArgs {
  const double* cost;       // [V] 0.5 |r|^2 at the current point (inf: a start that cannot be integrated)
  const double* norms_t;    // [V] |r|^2 at the trial point
  const int32_t* status_t;  // [V] integration status of the trial point (non-zero: failed)
  const double* pred;       // [V] from sbm_lm_trust_step_ex
  const double* dxnorm;     // [V]
  const double* gtx;        // [V]
  const int32_t* st;        // [V] status of the trust step (0 ok, 1 no system solved, 2 skipped)
  const double* theta;      // [V][q] current point (for ||D theta||)
  const double* dscale;     // [V][q]
  double* radius;           // [V] in / out
  double* lambda;           // [V] in / out
  int32_t* done;            // [V] in / out: 1 = converged (or never started)
  int32_t* accept;          // [V] out: 1 = take the trial point
  int32_t* n_iter;          // [V] in / out: iteration at which the start converged
  int32_t* counters;        // [2] out: starts still running, trial points accepted (zeroed by the caller's memset)
  double* ratio_out;        // [V] nullable: actual / predicted reduction (traces)
  double ftol, xtol;
  int V, q, iteration, first;
};

__global__ void __launch_bounds__(256) k_lm_update(LmUpdateArgs a) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= a.V) return;
  a.accept[v] = 0;
  if (a.done[v]) return;
  const double cost = a.cost[v];
  double cost_t = 0.5 * a.norms_t[v];
  if (!(cost_t < 1.0e300) || a.status_t[v] != 0) cost_t = __builtin_inf();
  const int st = a.st[v];
  double radius = a.radius[v], lam = a.lambda[v];
  const double dxn = a.dxnorm[v];
  if (a.first && st == 0) radius = fmin(radius, dxn);          // lmder: on the first iteration Delta = min(Delta, ||D p||)
  // lmder's quantities, relative to |r|^2 = 2 cost
  const double safe = cost > 0.0 ? cost : 1.0;
  const bool not_10x_worse = 0.1 * sqrt(cost_t) < sqrt(cost);   // (false for an infinite trial cost)
  const double actred = not_10x_worse ? 1.0 - cost_t / safe : -1.0;
  const double prered = a.pred[v] / safe;
  const double dirder = a.gtx[v] / (2.0 * safe);               // g . p / |r|^2 (= -(|J p|^2 + lam |D p|^2) / |r|^2 for an unclipped step)
  const double ratio = prered > 0.0 ? actred / prered : 0.0;
  if (a.ratio_out) a.ratio_out[v] = ratio;
  if (st != 0) {
    // no system could be solved: halve the radius, keep the point
    a.radius[v] = 0.5 * radius;
    atomicAdd(a.counters, 1);
    return;
  }
  if (ratio <= 0.25) {
    double temp = actred >= 0.0 ? 0.5 : 0.5 * dirder / ((dirder + 0.5 * actred) != 0.0 ? dirder + 0.5 * actred : -1.0);
    if (!not_10x_worse || temp < 0.1 || !(temp == temp) || !(fabs(temp) < 1.0e300)) temp = 0.1;
    radius = temp * fmin(radius, dxn / 0.1);
    lam = lam / temp;
  } else if (lam == 0.0 || ratio >= 0.75) {
    radius = dxn / 0.5;
    lam = 0.5 * lam;
  }
  const bool ok = ratio >= 1.0e-4 && cost_t < 1.0e300;
  a.accept[v] = ok ? 1 : 0;
  // lmder's convergence tests (info 1, 2); ||D theta|| at the point the iteration ends on
  double xn2 = 0.0;
  for (int c = 0; c < a.q; ++c) {
    const double t = a.dscale[(size_t)v * a.q + c] * a.theta[(size_t)v * a.q + c];
    xn2 = fma(t, t, xn2);
  }
  const bool conv_f = fabs(actred) <= a.ftol && prered <= a.ftol && 0.5 * ratio <= 1.0;
  const bool conv_x = radius <= a.xtol * sqrt(xn2);
  a.radius[v] = radius;
  a.lambda[v] = lam;
  if (conv_f || conv_x) {
    a.done[v] = 1;
    a.n_iter[v] = a.iteration + 1;
  } else {
    atomicAdd(a.counters, 1);
  }
  if (ok) atomicAdd(a.counters + 1, 1);
}

extern "C" int sbm_lm_update(sbm_ctx* ctx, const double* cost, const double* norms_trial, const int32_t* status_trial,
                             const double* pred, const double* dxnorm, const double* gtx, const int32_t* step_status,
                             const double* theta, const double* dscale, int32_t V, int32_t q, double ftol, double xtol,
                             int32_t iteration, int32_t first, double* radius, double* lambda, int32_t* done, int32_t* accept,
                             int32_t* n_iter, int32_t* counters, double* ratio_out) {
  if (!ctx || !cost || !norms_trial || !status_trial || !pred || !dxnorm || !gtx || !step_status || !theta || !dscale ||
      !radius || !lambda || !done || !accept || !n_iter || !counters)
    return sbm_fail(SBM_E_ARG, "sbm_lm_update: NULL argument");
  if (V < 0 || q <= 0) return sbm_fail(SBM_E_ARG, "sbm_lm_update: bad sizes V=%d q=%d", V, q);
  SBM_HIP(hipSetDevice(ctx->device));
  SBM_HIP(hipMemsetAsync(counters, 0, 2 * sizeof(int32_t), ctx->stream));
  if (V == 0) return 0;
  LmUpdateArgs a{cost, norms_trial, status_trial, pred, dxnorm, gtx, step_status, theta, dscale, radius, lambda, done, accept,
                 n_iter, counters, ratio_out, ftol, xtol, V, q, iteration, first};
  hipLaunchKernelGGL(k_lm_update, dim3((V + 255) / 256), dim3(256), 0, ctx->stream, a);
  SBM_HIP(hipGetLastError());
  return 0;
}

// accepted trial points become the current ones: theta, residuals, Jacobian (scaled by row_scale if given), cost
__global__ void __launch_bounds__(256) k_lm_accept(const int32_t* __restrict__ accept, int q, int M, const double* __restrict__ trial,
                                                   const double* __restrict__ r_t, const double* __restrict__ J_t,
                                                   const double* __restrict__ norms_t, double* __restrict__ theta,
                                                   double* __restrict__ r, double* __restrict__ J, double* __restrict__ cost) {
  const int v = blockIdx.x;
  if (!accept[v]) return;
  const size_t nJ = (size_t)M * q;
  const double2* src = reinterpret_cast<const double2*>(J_t + (size_t)v * nJ);
  double2* dst = reinterpret_cast<double2*>(J + (size_t)v * nJ);
  if ((nJ & 1) == 0 && ((((size_t)v * nJ) & 1) == 0)) {
    for (size_t e = threadIdx.x + (size_t)blockIdx.y * blockDim.x; e < nJ / 2; e += (size_t)blockDim.x * gridDim.y) dst[e] = src[e];
  } else {
    for (size_t e = threadIdx.x + (size_t)blockIdx.y * blockDim.x; e < nJ; e += (size_t)blockDim.x * gridDim.y)
      J[(size_t)v * nJ + e] = J_t[(size_t)v * nJ + e];
  }
  if (blockIdx.y == 0) {
    for (int e = threadIdx.x; e < M; e += blockDim.x) r[(size_t)v * M + e] = r_t[(size_t)v * M + e];
    for (int e = threadIdx.x; e < q; e += blockDim.x) theta[(size_t)v * q + e] = trial[(size_t)v * q + e];
    if (threadIdx.x == 0) cost[v] = 0.5 * norms_t[v];
  }
}

extern "C" int sbm_lm_accept(sbm_ctx* ctx, const int32_t* accept, int32_t V, int32_t M, int32_t q, const double* trial,
                             const double* r_trial, const double* J_trial, const double* norms_trial, double* theta,
                             double* r, double* J, double* cost) {
  if (!ctx || !accept || !trial || !r_trial || !J_trial || !norms_trial || !theta || !r || !J || !cost)
    return sbm_fail(SBM_E_ARG, "sbm_lm_accept: NULL argument");
  if (V < 0 || M <= 0 || q <= 0) return sbm_fail(SBM_E_ARG, "sbm_lm_accept: bad sizes V=%d M=%d q=%d", V, M, q);
  if (V == 0) return 0;
  SBM_HIP(hipSetDevice(ctx->device));
  const int ny = (int)(((size_t)M * q / 2 + 256 * 8 - 1) / (256 * 8));       // ~8 double2 per thread
  hipLaunchKernelGGL(k_lm_accept, dim3(V, ny < 1 ? 1 : (ny > 64 ? 64 : ny)), dim3(256), 0, ctx->stream, accept, q, M, trial, r_trial,
                     J_trial, norms_trial, theta, r, J, cost);
  SBM_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------
// The one collective of the path: all-gather of the per-vector residual norms.
// RCCL is resolved at run time from whatever instance the process already has loaded -- the one that
// created the caller's communicator -- so that libsbm_hip.so neither links RCCL nor brings a second copy
// into a process whose host framework ships its own.
// ---------------------------------------------------------------------------------------------
typedef int (*sbm_nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);

extern "C" int sbm_allgather_norms(sbm_ctx* ctx, void* nccl_comm, const double* send, int32_t count, double* recv) {
  if (!ctx || !nccl_comm || !send || !recv) return sbm_fail(SBM_E_ARG, "sbm_allgather_norms: NULL argument");
  if (count < 0) return sbm_fail(SBM_E_ARG, "sbm_allgather_norms: count < 0");
  if (count == 0) return 0;
  static sbm_nccl_allgather_fn fn = nullptr;
  if (!fn) {
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);      // already in the process?
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);    // no: load the system one
    if (!h) return sbm_fail(SBM_E_PLUGIN, "sbm_allgather_norms: RCCL not found: %s", dlerror());
    fn = (sbm_nccl_allgather_fn)dlsym(h, "ncclAllGather");
    if (!fn) return sbm_fail(SBM_E_PLUGIN, "sbm_allgather_norms: ncclAllGather not found in RCCL");
  }
  SBM_HIP(hipSetDevice(ctx->device));
  const int nccl_double = 8;   // ncclFloat64 (rccl.h)
  const int rc = fn(send, recv, (size_t)count, nccl_double, nccl_comm, ctx->stream);
  if (rc != 0) return sbm_fail(SBM_E_HIP, "sbm_allgather_norms: ncclAllGather returned %d", rc);
  return 0;
}
