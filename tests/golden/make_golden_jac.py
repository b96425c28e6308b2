"""Generate tests/golden/jac_path_ref.npz: the REAL reference OdeModel on its ``use_jac`` path.

Run in the build container only (it reads /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_jac.py

The reference hands ``model_jac`` / ``sens_model_jac`` to LSODA as ``Dfun`` with ``col_deriv=True``
(model/ode_model.py:114-120,154-160).  Here the reference ``OdeModel`` integrates the build's Michaelis-Menten
and cascade20 models (non-stiff: LSODA stays on its Adams branch and never calls Dfun) and two stiff cascades
(it does) twice -- with the build's generated analytic Jacobian callbacks and without -- so that the
callback contract (which index is the row, which entries may be left untouched) is pinned by the reference
itself: a transposed or mis-filled Jacobian changes LSODA's step sequence and, on the BDF branch, its result.
Only numbers are stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, 'model'))

import ode_model as ref_ode_model  # noqa: E402  (reference)

from sysbio_modeling_amd.symbolic import zoo_model  # noqa: E402
from sysbio_modeling_amd import models_zoo  # noqa: E402


def main():
    out = {}
    cases = {'michaelis_menten': (np.array([[1e-3, 1e-3, 0.01, 0.01, 1e-3], [0.5, 0.3, 0.8, 0.2, 0.1]]),
                                  np.linspace(0, 100, 1000)),
             'cascade20': (models_zoo.cascade_ensemble(2)[1], np.linspace(0, 100, 1000))}
    from sysbio_modeling_amd.symbolic import GeneratedModel
    # LSODA evaluates Dfun on its BDF branch only: two stiff systems (rates spanning 1e6) make it do so
    cases['stiff50'] = (models_zoo.stiff_ensemble(2)[1], np.linspace(0, models_zoo.STIFF_T_END, 1000))
    cases['stiff12'] = (models_zoo.stiff_ensemble(2, n=12)[1], np.linspace(0, models_zoo.STIFF_T_END, 1000))
    for name, (P, grid) in cases.items():
        gm = GeneratedModel(models_zoo.stiff_spec(12, name='stiff12')) if name == 'stiff12' else zoo_model(name)
        sens = name != 'stiff50'           # 2550 x 2550 dense Jacobians in Python: state only there
        n, k = gm.n_vars, gm.n_sens
        with_jac = ref_ode_model.OdeModel(gm.model, gm.sens_model, n, list(gm.param_order), use_jit=False,
                                          model_jac=gm.model_jac, sens_model_jac=gm.sens_model_jac)
        assert with_jac.use_jac and with_jac.model_jac is not None
        without = ref_ode_model.OdeModel(gm.model, gm.sens_model, n, list(gm.param_order), use_jit=False)
        idx = np.arange(0, 1000, 111)
        Yj = np.stack([with_jac.simulate(p, grid)[idx] for p in P])
        Y0 = np.stack([without.simulate(p, grid)[idx] for p in P])
        if sens:
            Sj = np.stack([with_jac.calc_jacobian(p, grid, np.zeros(n + n * k))[idx] for p in P])
            S0 = np.stack([without.calc_jacobian(p, grid, np.zeros(n + n * k))[idx] for p in P])
        else:
            Sj = S0 = np.zeros((len(P), len(idx), 0))
        print(name, "with vs without Dfun: max |dY| %.2e  max |dS| %.2e" % (np.max(np.abs(Yj - Y0)), np.max(np.abs(Sj - S0), initial=0.0)))
        out.update({name + '_P': P, name + '_t': grid, name + '_idx': idx, name + '_Y_jac': Yj, name + '_S_jac': Sj,
                    name + '_Y_nojac': Y0, name + '_S_nojac': S0})
    np.savez_compressed(os.path.join(HERE, 'jac_path_ref.npz'), **out)


if __name__ == '__main__':
    main()
