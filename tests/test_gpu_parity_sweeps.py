"""Parity at ensemble scale and against TIGHT solutions, on the GPU box:

  * 1024 vectors of the headline ensemble (BASELINE configs[2]: cascade20, 820 coupled ODEs) against the oracle's
    LSODA call, every sampled state and sensitivity; where a vector exceeds |gpu - ref| <= 1e-8 |ref| + 5e-9 the
    disagreement must be LSODA's own error, shown against a DOP853 solution at rtol 1e-13 (conftest.check_parity);
  * DEFAULT integrator options against tight solutions under SURVEY.md section 8(d)'s criterion
    |gpu - ref| <= 1e-8 max(|ref|, 1e-6 column max) for cascades of 20, 40 and 70 states -- what the size-aware
    default tolerances (model/ode_model.py::default_tolerances) are for;
  * 128 vectors of the configs[3] project (8 experiments, 512 rows x 68 parameters) against the assembly oracle with
    tolerances propagated to first order from the trajectory tolerances (oracle/tolerances.py), and a few of them
    against the assembly oracle driven by the tight integrator.
"""
import multiprocessing as mp
import warnings

import numpy as np
import pytest

from tests.conftest import (parity_err, survey_err, tol_ratio, check_parity, lsoda_taus, tight_taus,
                            project_tolerances)

pytestmark = pytest.mark.gpu

_gm_cache = {}


def _cascade(n):
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.symbolic import GeneratedModel, zoo_model
    if n not in _gm_cache:
        _gm_cache[n] = zoo_model('cascade20') if n == 20 else GeneratedModel(models_zoo.cascade_spec(n, name='cascade%d' % n))
    return _gm_cache[n]


def _lsoda_worker(args):
    """(state rows, sensitivity rows) of the oracle's LSODA call at the sampled grid points, for a slice of vectors"""
    n, P, grid, idx = args
    from oracle import odeint_oracle as oo
    gm = _cascade(n)
    out = []
    for p in P:
        S, Y = oo.calc_jacobian(gm, p, grid, use_c=True, return_states=True)
        out.append((Y[idx], S[idx]))
    return out


def _tight_worker(args):
    n, P, t_out = args
    from oracle import odeint_oracle as oo
    gm = _cascade(n)
    return [oo.tight_solution(gm, p, t_out, use_c=True, atol=1e-30)[1:] for p in P]


def _pool_map(fn, jobs, workers=12):
    # spawned, not forked: this process holds a GPU context
    with mp.get_context('spawn').Pool(min(workers, len(jobs))) as pool:
        return [x for part in pool.map(fn, jobs) for x in part]


def test_headline_ensemble_1024_vectors_against_the_oracle(gpu_models):
    from sysbio_modeling_amd import models_zoo
    from oracle import odeint_oracle as oo
    gm = _cascade(20)
    gm.c_library()
    m = gpu_models('cascade20')
    _, P = models_zoo.cascade_ensemble(4096)
    pick = np.arange(0, 4096, 4)                       # 1024 vectors spread over the ensemble
    grid = np.linspace(0, 100.0, 1000)
    idx = np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)
    t_out = np.concatenate([[0.0], grid[idx]])
    S, Y = m.calc_jacobian_batch(P[pick], t_out, return_states=True)
    assert not m.last_info['status'].any()
    chunks = np.array_split(np.arange(len(pick)), 24)
    ref = _pool_map(_lsoda_worker, [(20, P[pick[c]], grid, idx) for c in chunks])
    ey = np.array([parity_err(Y[v, 1:], ref[v][0]) for v in range(len(pick))])
    es = np.array([parity_err(S[v, 1:], ref[v][1]) for v in range(len(pick))])
    over = np.flatnonzero((ey > 1.0) | (es > 1.0))
    # the exceptions (r01: 2 of 1024 at 1.3 units) must be LSODA's error, not the GPU's
    for v in over:
        tight = oo.tight_solution(gm, P[pick[v]], t_out, use_c=True, atol=1e-30)[1:]
        check_parity(np.concatenate([Y[v, 1:], S[v, 1:]], axis=1), np.concatenate(ref[v], axis=1), tight,
                     what='vector %d' % pick[v])
    assert len(over) <= 16, "more than 1.5 %% of the vectors disagree with LSODA: %s" % es[over]
    print("1024 vectors: state err median %.3f max %.3f, sens err median %.3f p99 %.3f max %.3f tolerance units; "
          "%d beyond 1 (all LSODA's own error)" % (np.median(ey), ey.max(), np.median(es), np.percentile(es, 99),
                                                   es.max(), len(over)))


@pytest.mark.parametrize('n,n_vec', [(20, 12), (40, 8), (70, 4)])
def test_default_options_meet_section_8d_against_a_tight_solution(n, n_vec):
    """DOP853 at rtol 1e-13 as the reference; OdeModel with its DEFAULT options."""
    from sysbio_modeling_amd import models_zoo
    from sysbio_modeling_amd.model import OdeModel
    gm = _cascade(n)
    gm.c_library()
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    if n == 20:
        _, P = models_zoo.cascade_ensemble(4096)
        P = P[np.linspace(0, 4095, n_vec).astype(int)]
        grid = np.linspace(0, 100.0, 1000)
        idx = np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)
    else:
        rng = np.random.default_rng(2026)
        P = models_zoo.cascade_nominal_params(n)[None, :] * np.exp(0.3 * rng.standard_normal((n_vec, 2 * n)))
        grid = np.linspace(0, 60.0, 1000)
        idx = np.array([100, 300, 600, 999])
    t_out = np.concatenate([[0.0], grid[idx]])
    S, Y = m.calc_jacobian_batch(P, t_out, return_states=True)        # default options
    assert not m.last_info['status'].any()
    tight = _pool_map(_tight_worker, [(n, P[c], t_out) for c in np.array_split(np.arange(n_vec), min(n_vec, 8))])
    ey = [survey_err(Y[v, 1:], tight[v][:, :n]) for v in range(n_vec)]
    es = [survey_err(S[v, 1:], tight[v][:, n:]) for v in range(n_vec)]
    print("cascade%d default %s: state %.2f sens %.2f section-8(d) units, %.0f steps per vector"
          % (n, {k: m.integrator_options[k] for k in ('rtol', 'atol')}, max(ey), max(es), m.last_info['n_steps'].mean()))
    assert max(ey) <= 1.0 and max(es) <= 1.0


def test_config3_project_rows_128_vectors(gpu_models):
    """Residual rows, scale factors and Jacobian rows of the configs[3] project for 128 vectors of its ensemble."""
    import torch
    from sysbio_modeling_amd import models_zoo
    from oracle.project_oracle import ProjectOracle
    gm = _cascade(20)
    model = gpu_models('cascade20')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        proj, th0 = models_zoo.cascade_config4_project(model)
    thetas = models_zoo.config4_ensemble(th0, 1024)[::8]
    out = proj.evaluate_batch(thetas, jacobian=True, want=('jacobian', 'model_jacobian', 'sf_gradient'))
    assert not out['status'].any()
    exps = [proj.get_experiment(i) for i in range(8)]
    po = ProjectOracle(gm, exps, proj._model_parameter_settings,
                       {k: (v['type'], v['variables'][0]) for k, v in proj._measurement_to_model_map.items()},
                       sf_groups=['s%d' % v for v in models_zoo.CASCADE_MEASURED_SPECIES])
    a = proj.descriptor_arrays()
    keys = ('residuals', 'jacobian', 'sf', 'model_jacobian')

    def errors(v, tight):
        po.tight = tight
        ref, sims, B = po.residuals(thetas[v], return_parts=True)
        Jm = po.model_jacobian(thetas[v])
        want = dict(residuals=ref, jacobian=po.calc_project_jacobian(thetas[v]), sf=B, model_jacobian=Jm)
        tau_s, tau_Jm = (tight_taus(a, sims, Jm) if tight else lsoda_taus(a, thetas[v], sims, Jm))
        t = project_tolerances(a, sims, B, tau_s, Jm, tau_Jm)
        return {k: tol_ratio(out[k][v], want[k], t[k]) for k in keys}, want, t
    worst = dict.fromkeys(keys, 0.0)
    over = []
    for v in range(len(thetas)):
        e, _, _ = errors(v, False)
        if max(e.values()) > 1.0:
            over.append(v)
        for k in keys:
            worst[k] = max(worst[k], e[k])
    print("configs[3], 128 vectors, worst error against the LSODA-driven oracle in propagated tolerance units:", worst,
          "; vectors beyond 1:", over)
    # Where a vector disagrees with the LSODA-driven oracle the disagreement must be LSODA's (as in the trajectory
    # sweep above): against the oracle driven by the TIGHT integrator, with section 8(d) propagated, the GPU passes
    # and the LSODA-driven oracle itself does not do better.  Three more vectors take the tight check regardless.
    assert len(over) <= 8
    for v in sorted(set(over) | {0, 64, 127}):
        e_tight, want_t, tol_t = errors(v, True)
        assert max(e_tight.values()) <= 1.0, (v, e_tight)
        if v in over:
            _, want_l, _ = errors(v, False)
            lsoda_vs_tight = max(tol_ratio(want_l[k], want_t[k], tol_t[k]) for k in keys)
            assert lsoda_vs_tight > max(e_tight.values()), (v, lsoda_vs_tight, e_tight)
    del torch
