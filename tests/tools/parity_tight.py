"""Evidence tool (not a test): the GPU integrator at several tolerances against a TIGHT solution (DOP853, rtol 1e-13,
compiled C right-hand side) under SURVEY.md section 8(d)'s criterion

    |gpu - ref| <= 1e-8 * max(|ref|, 1e-6 * column max-abs)

for cascade(n), n states / 2n parameters, and the same for the reference's integrator (odeint at rtol = atol = 1e-10).
This is what the size-aware default tolerance of OdeModel is tuned on.

    python tests/tools/parity_tight.py <n_states> <N vectors> [out.json]
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sysbio_modeling_amd import models_zoo
from sysbio_modeling_amd.symbolic import GeneratedModel, zoo_model

_gm = None


def survey_err(a, ref, block_floor=0.0):
    """max over entries of |a - ref| / (1e-8 * max(|ref|, 1e-6 * colmax, block_floor * blockmax)); a, ref: (T, ncols).
    Returns (err, description of the worst entry)."""
    colmax = np.max(np.abs(ref), axis=0, keepdims=True)
    blockmax = np.max(np.abs(ref))
    tol = 1e-8 * np.maximum(np.maximum(np.abs(ref), 1e-6 * colmax), block_floor * blockmax)
    d = np.abs(a - ref)
    with np.errstate(invalid='ignore', divide='ignore'):
        e = np.where(d == 0.0, 0.0, d / tol)
    w = np.unravel_index(np.argmax(e), e.shape)
    return float(np.max(e)), (float(abs(ref[w]) / blockmax), float(colmax[0, w[1]] / blockmax), float(d[w] / max(abs(ref[w]), 1e-300)))


def _model(n):
    return zoo_model('cascade20') if n == 20 else GeneratedModel(models_zoo.cascade_spec(n, name='cascade%d' % n))


def _refs(args):
    global _gm
    n, p, t_out, grid, idx = args
    from oracle import odeint_oracle as oo
    if _gm is None:
        _gm = _model(n)
    tight = oo.tight_solution(_gm, p, t_out, use_c=True, atol=1e-30)
    S, Y = oo.calc_jacobian(_gm, p, grid, use_c=True, return_states=True)
    return tight[1:], np.concatenate([Y[idx], S[idx]], axis=1)


def main():
    n = int(sys.argv[1])
    N = int(sys.argv[2])
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    if n == 20:
        _, P = models_zoo.cascade_ensemble(4096)
        P = P[np.linspace(0, 4095, N).astype(int)]
        grid = np.linspace(0, 100.0, 1000)
        idx = np.searchsorted(grid, models_zoo.CASCADE_MEASURE_TIMES)
    else:
        rng = np.random.default_rng(2026)
        P = models_zoo.cascade_nominal_params(n)[None, :] * np.exp(0.3 * rng.standard_normal((N, 2 * n)))
        grid = np.linspace(0, 60.0, 1000)
        idx = np.array([100, 300, 600, 999])
    t_out = np.concatenate([[0.0], grid[idx]])
    _model(n).c_library()
    t0 = time.time()
    with mp.get_context('spawn').Pool(min(N, 14)) as pool:
        refs = pool.map(_refs, [(n, p, t_out, grid, idx) for p in P])
    t_ref = time.time() - t0
    from sysbio_modeling_amd.model import OdeModel
    gm = _model(n)
    m = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=gm.spec.name)
    nv = gm.n_vars
    res = {'model': 'cascade%d' % n, 'vectors': N, 'criterion': '|a - tight| <= 1e-8 max(|tight|, 1e-6 colmax)',
           'default_options': dict(m.integrator_options), 'ref_seconds': t_ref, 'runs': {}}
    FLOORS = (0.0, 1e-14, 1e-12, 1e-10)

    def stats(get):
        out = {}
        for bf in FLOORS:
            ey = [survey_err(get(v)[:, :nv], refs[v][0][:, :nv], bf) for v in range(N)]
            es = [survey_err(get(v)[:, nv:], refs[v][0][:, nv:], bf) for v in range(N)]
            wy, ws = int(np.argmax([e[0] for e in ey])), int(np.argmax([e[0] for e in es]))
            out['block_floor_%g' % bf] = {
                'state_med_max': [float(np.median([e[0] for e in ey])), ey[wy][0]],
                'sens_med_max': [float(np.median([e[0] for e in es])), es[ws][0]],
                'worst_sens_entry(|ref|/blockmax, colmax/blockmax, rel_err)': es[ws][1]}
        return out
    res['lsoda_vs_tight'] = stats(lambda v: refs[v][1])
    print('lsoda', json.dumps(res['lsoda_vs_tight']), flush=True)
    runs = [('default', {})] + [('rtol%g_atol%g' % (r, a), dict(rtol=r, atol=a))
                                for r in (1e-9, 3e-10, 1e-10, 3e-11) for a in (1e-12, 1e-14, 1e-16, 1e-18, 1e-20)]
    for label, kw in runs:
        S, Y = m.calc_jacobian_batch(P, t_out, return_states=True, **kw)
        ok = not m.last_info['status'].any()
        YS = np.concatenate([Y[:, 1:], S[:, 1:]], axis=2)
        es5 = np.array([np.max(np.abs(S[v, 1:] - refs[v][1][:, nv:]) / (1e-8 * np.abs(refs[v][1][:, nv:]) + 5e-9)) for v in range(N)])
        res['runs'][label] = dict(stats(lambda v: YS[v]), ok=ok, steps_mean=float(m.last_info['n_steps'].mean()),
                                  sens_vs_lsoda_floor5e9_med_max=[float(np.median(es5)), float(es5.max())])
        print(label, json.dumps(res['runs'][label]), flush=True)
    print(json.dumps(res, indent=1))
    if out_path:
        with open(out_path, 'w') as fh:
            json.dump(res, fh, indent=1)


if __name__ == '__main__':
    main()
