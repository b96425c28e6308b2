"""tests/golden/loss_grouping_ref.npz: the REAL reference ``LossFunctionWithScaleFactors`` (abstract_loss_function.py) with
the REAL ``LogScaleFactor`` on random (experiment, measure)-indexed frames.

Of the reference's loss package, SquareLossFunction and its subclasses import LinearScaleFactor and with it numba, which
this image does not have; the base class that owns the GROUPING logic -- which rows of which experiments enter which
scale factor, measures that share one, how the factors are applied, where the prior row goes -- needs numpy / pandas only
and runs here as it stands, and so does LogScaleFactor (make_golden_log_scale_factor.py).  Both are loaded from their files
(parent packages entered as empty packages so that project/__init__.py, which pulls in numba, does not run).  Not called:
update_sf_priors_gradient (uses DataFrame.ix, gone from pandas).

Recorded per case: the frames (experiment / measure labels, simulated value, measured value, error bar, time), the groups,
and what the reference answers -- the scale factor of every group after update_scale_factors, the simulations after
scale_sim_values, and the frame after update_sf_priors_residuals (the prior row's value = log B).

Run in the build container only (reads /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_loss_grouping.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.dont_write_bytecode = True


def reference_classes():
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
    P = os.path.join(REF, 'project')
    package('refproject', P)
    package('refproject.loss_functions', os.path.join(P, 'loss_functions'))
    package('refproject.loss_functions.squared_loss', os.path.join(P, 'loss_functions', 'squared_loss'))
    load('refproject.utils', os.path.join(P, 'utils.py'))
    load('refproject.loss_functions.abstract_scale_factor', os.path.join(P, 'loss_functions', 'abstract_scale_factor.py'))
    base = load('refproject.loss_functions.abstract_loss_function', os.path.join(P, 'loss_functions', 'abstract_loss_function.py'))
    lsf = load('refproject.loss_functions.squared_loss.log_scale_factor',
               os.path.join(P, 'loss_functions', 'squared_loss', 'log_scale_factor.py'))
    return base.LossFunctionWithScaleFactors, lsf.LogScaleFactor


CASES = [   # (measures per experiment, groups, prior on the first group? -- a plain name: for a shared factor the reference
            #  names the prior row after next(iter(frozenset)), abstract_loss_function.py:101, whichever that is)
    ([['A', 'B'], ['A', 'B', 'C'], ['B', 'C', 'D']], ['A', frozenset(['B', 'C'])], False),
    ([['A', 'B', 'C', 'D'], ['A', 'C']], ['A', frozenset(['C', 'D'])], True),
    ([['M'], ['M'], ['M', 'N']], ['M'], True),
    ([['A', 'B', 'C']], [frozenset(['A', 'B', 'C'])], False),
]


def main():
    Base, LogScaleFactor = reference_classes()
    rng = np.random.default_rng(20261007)
    out = {'n_cases': np.array(len(CASES))}
    for c, (layout, groups, with_prior) in enumerate(CASES):
        idx, sim_v, dat, std, tt = [], [], [], [], []
        for e, measures in enumerate(layout):
            for m in measures:
                n = int(rng.integers(2, 7))
                s = np.exp(rng.uniform(-1.0, 2.0, n))
                idx += [('Exp%d' % e, m)] * n
                sim_v += list(s)
                dat += list(s * np.exp(rng.uniform(0.3, 1.2)) * np.exp(0.05 * rng.standard_normal(n)))
                std += list(rng.uniform(0.05, 0.5, n))
                tt += list(np.arange(1, n + 1, dtype=float))
        first = groups[0] if isinstance(groups[0], str) else sorted(groups[0])[0]
        prior = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(0.3, 1.5)))
        if with_prior:
            idx.append(("~~SF_Prior", "~%s" % first))
            sim_v.append(0.0); dat.append(prior[0]); std.append(prior[1]); tt.append(np.nan)
        mi = pd.MultiIndex.from_tuples(idx)
        sim = pd.DataFrame({'mean': sim_v, 'timecourse': tt}, index=mi).sort_index()
        mea = pd.DataFrame({'mean': dat, 'std': std, 'timecourse': tt}, index=mi).sort_index()
        lf = Base(groups, LogScaleFactor)
        if with_prior:
            lf.set_scale_factor_priors(first, *prior)
        lf.update_scale_factors(sim, mea)
        scaled = lf.scale_sim_values(sim)
        after = sim.copy()
        lf.update_sf_priors_residuals(after)
        out['exp_%d' % c] = np.array([ix[0] for ix in sim.index])
        out['measure_%d' % c] = np.array([ix[1] for ix in sim.index])
        out['sim_%d' % c], out['data_%d' % c], out['std_%d' % c] = sim['mean'].values, mea['mean'].values, mea['std'].values
        out['time_%d' % c] = sim['timecourse'].values
        out['groups_%d' % c] = np.array(['|'.join([g] if isinstance(g, str) else sorted(g)) for g in groups])
        out['prior_%d' % c] = np.array(prior if with_prior else (np.nan, np.nan))
        out['sf_%d' % c] = np.array([lf.scale_factors[g].sf for g in groups], dtype=float)
        out['scaled_%d' % c] = scaled['mean'].values
        out['after_prior_update_%d' % c] = after['mean'].values
    np.savez_compressed(os.path.join(HERE, 'loss_grouping_ref.npz'), **out)
    print('loss_grouping_ref.npz written:', len(CASES), 'cases')


if __name__ == '__main__':
    main()
