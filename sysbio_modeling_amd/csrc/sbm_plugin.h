/* Internal ABI between libsbm_hip.so and a per-model plugin (sbm_model_<name>.so).
 * A plugin = csrc/sbm_plugin_main.hip compiled against one generated model
 * header; it owns the integrator kernels specialised for that model. */
#ifndef SBM_PLUGIN_H
#define SBM_PLUGIN_H

#include <stdint.h>
#include "../../include/sbm.h"

#define SBM_PLUGIN_ABI 2

typedef struct sbm_plugin_info_t {
  int32_t abi;
  int32_t n_vars;
  int32_t n_params;
  int32_t n_sens;
  char name[64];
} sbm_plugin_info_t;

/* one launch = n_traj independent trajectories */
typedef struct sbm_kernel_args {
  const double* P;          /* [n_traj][NP]                                             */
  const double* t_out;      /* output times; one shared grid or several concatenated     */
  const int32_t* grid_off;  /* [n_traj] offset of the trajectory's grid in t_out, or NULL */
  const int32_t* grid_len;  /* [n_traj] number of output times, or NULL (-> n_t)          */
  const double* y0;         /* [NV] or NULL                                               */
  const double* s0;         /* [NV][NK] or NULL                                           */
  double* Y;                /* [n_traj][n_t][NV]        nullable in sens mode             */
  double* S;                /* [n_traj][n_t][NV][NK]    sens mode only                    */
  int32_t* status;          /* [n_traj] nullable */
  int32_t* n_steps;         /* [n_traj] nullable */
  int32_t* n_reject;        /* [n_traj] nullable */
  const int32_t* order;     /* [n_traj] permutation or NULL: workgroup b integrates trajectory
                             * order[b] (sens kernels; longest first, see sbm_core.hip)    */
  int32_t n_traj;
  int32_t n_t;              /* rows allocated per trajectory in Y / S                     */
  sbm_integrator_opts opts;
} sbm_kernel_args;

enum { SBM_KIND_STATE = 0, SBM_KIND_SENS = 1 };

#ifdef __cplusplus
extern "C" {
#endif
void sbm_plugin_info(sbm_plugin_info_t* out);
/* returns a hipError_t as int */
int sbm_plugin_launch(int kind, const sbm_kernel_args* args, void* stream);
#ifdef __cplusplus
}
#endif

typedef void (*sbm_plugin_info_fn)(sbm_plugin_info_t*);
typedef int (*sbm_plugin_launch_fn)(int, const sbm_kernel_args*, void*);

#endif
