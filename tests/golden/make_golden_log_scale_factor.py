"""tests/golden/log_scale_factor_ref.npz: the REAL reference ``LogScaleFactor`` on random inputs.

The reference's ``Project`` and ``SquareLossFunction`` cannot be imported here (Python-2 imports, numba), so the assembly half
of the oracle is pinned by the reference's own known answers only.  One class of that half CAN be
run as it stands: project/loss_functions/squared_loss/log_scale_factor.py needs numpy alone.  It is loaded from its file
(its parent packages are entered as empty packages so that project/__init__.py, which pulls in numba, does not run) and
called the way LogSquareLossFunction calls it -- update_sf, update_sf_gradient, the prior residual and its gradient, each
with FRESH copies of the arrays (the reference divides exp_std by exp_data in place, log_scale_factor.py:18,31).

Run in the build container only (reads /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_log_scale_factor.py
Only numbers are stored: the inputs of every case and what the class returned."""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.dont_write_bytecode = True


def reference_class():
    def package(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    package('refproject', os.path.join(REF, 'project'))
    package('refproject.loss_functions', os.path.join(REF, 'project', 'loss_functions'))
    package('refproject.loss_functions.squared_loss', os.path.join(REF, 'project', 'loss_functions', 'squared_loss'))
    load('refproject.loss_functions.abstract_scale_factor', os.path.join(REF, 'project', 'loss_functions', 'abstract_scale_factor.py'))
    return load('refproject.loss_functions.squared_loss.log_scale_factor',
                os.path.join(REF, 'project', 'loss_functions', 'squared_loss', 'log_scale_factor.py')).LogScaleFactor


def main():
    LogScaleFactor = reference_class()
    rng = np.random.default_rng(20261005)
    out = {}
    sizes = [3, 7, 16, 40, 101, 5]
    for c, n in enumerate(sizes):
        q = 1 + c % 4
        sim = np.exp(rng.uniform(-2.0, 3.0, n))
        data = sim * np.exp(rng.uniform(0.5, 1.5)) * np.exp(0.1 * rng.standard_normal(n))
        std = data * rng.uniform(0.02, 0.3, n)
        jac = rng.standard_normal((n, q)) * sim[:, None]
        prior = (None, None) if c % 2 == 0 else (float(rng.uniform(-1, 1)), float(rng.uniform(0.2, 2.0)))
        sf = LogScaleFactor(*prior)
        sf.update_sf(sim.copy(), data.copy(), std.copy())
        sf.update_sf_gradient(sim.copy(), data.copy(), std.copy(), jac.copy())
        out['sim_%d' % c], out['data_%d' % c], out['std_%d' % c], out['jac_%d' % c] = sim, data, std, jac
        out['prior_%d' % c] = np.array([np.nan, np.nan] if prior[0] is None else prior)
        out['sf_%d' % c] = np.array(sf.sf)
        out['sf_gradient_%d' % c] = sf.gradient
        pr, pg = sf.calc_sf_prior_residual(), sf.calc_sf_prior_gradient()
        out['prior_residual_%d' % c] = np.array(np.nan if pr is None else pr)
        out['prior_gradient_%d' % c] = np.full(q, np.nan) if pg is None else np.asarray(pg)
    out['n_cases'] = np.array(len(sizes))
    np.savez_compressed(os.path.join(HERE, 'log_scale_factor_ref.npz'), **out)
    print('log_scale_factor_ref.npz written:', len(sizes), 'cases')


if __name__ == '__main__':
    main()
