"""The CPU legs of BASELINE configs[4] over the FULL time span, one vector, one core -- minutes of work, so not part of
a default bench.py run: measured once per round and committed (profiles/<round>/stiff50_cpu_full_span.json), from where
bench.py quotes it.

    python tests/tools/cpu_leg_stiff50.py profiles/r03/stiff50_cpu_full_span.json [vector]

  as_reference   the reference's default call: odeint(..., Dfun=None, rtol=atol=1e-10) on its 1000-point grid
                 (model/ode_model.py:122-123) -- LSODA differences and factors a dense 2550 x 2550 Jacobian;
  analytic_dfun  the reference's use_jac path (:114-120) with the generated analytic Jacobian as Dfun.
"""
import json
import os
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")      # one core, honestly: LAPACK would otherwise use them all
os.environ.setdefault("OMP_NUM_THREADS", "1")
import platform
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main(out_path, vector=3):
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import zoo_model
    from oracle import odeint_oracle as oo
    gm = zoo_model('stiff50')
    gm.c_library()
    p = models_zoo.stiff_ensemble(4096)[1][vector]
    grid = np.linspace(0.0, models_zoo.STIFF_T_END, 1000)
    res = {"vector": vector, "host": platform.processor() or platform.machine(), "cores_used": 1,
           "n_equations": gm.n_vars * (1 + gm.n_sens)}
    t0 = time.perf_counter()
    (_, _), info = oo.calc_jacobian(gm, p, grid, use_c=True, return_states=True, full_output=True)
    dt = time.perf_counter() - t0
    res["as_reference"] = {"seconds": dt, "lsoda_steps": int(info['nst'][-1]), "jacobian_evaluations": int(info['nje'][-1]),
                           "steps_per_s": int(info['nst'][-1]) / dt, "call": "odeint rtol=atol=1e-10, Dfun=None, compiled C RHS"}
    print(res["as_reference"], flush=True)
    jac = gm.sens_model_jac
    t0 = time.perf_counter()
    (_, _), info = oo.calc_jacobian(gm, p, grid, use_c=True, return_states=True, full_output=True, sens_model_jac=jac)
    dt = time.perf_counter() - t0
    res["analytic_dfun"] = {"seconds": dt, "lsoda_steps": int(info['nst'][-1]), "jacobian_evaluations": int(info['nje'][-1]),
                            "steps_per_s": int(info['nst'][-1]) / dt,
                            "call": "odeint rtol=atol=1e-10, Dfun=GeneratedModel.sens_model_jac (generated Python), compiled C RHS"}
    print(res["analytic_dfun"], flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    with open(out_path, 'w') as fh:
        json.dump(res, fh, indent=1)


if __name__ == '__main__':
    main(sys.argv[1], *[int(x) for x in sys.argv[2:3]])
