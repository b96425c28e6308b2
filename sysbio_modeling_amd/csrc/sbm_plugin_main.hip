// One translation unit per model: compile with
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared -DSBM_MODEL_HEADER='"<generated>.hpp"' \
//         sbm_plugin_main.hip -o sbm_model_<name>.so
// The generated header (symbolic/emit.py::emit_hip) defines `struct SbmModel`.
#include "sbm_integrators.hpp"

#ifndef SBM_MODEL_HEADER
#error "define SBM_MODEL_HEADER to the generated model header"
#endif
#include SBM_MODEL_HEADER

#include <string.h>

extern "C" void sbm_plugin_info(sbm_plugin_info_t* out) {
  out->abi = SBM_PLUGIN_ABI;
  out->n_vars = SbmModel::NV;
  out->n_params = SbmModel::NP;
  out->n_sens = SbmModel::NK;
  strncpy(out->name, SbmModel::NAME, sizeof(out->name) - 1);
  out->name[sizeof(out->name) - 1] = 0;
}

extern "C" int sbm_plugin_launch(int kind, const sbm_kernel_args* args, void* stream) {
  return sbm_launch_model<SbmModel>(kind, args, (hipStream_t)stream);
}
