import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')

# the parity tolerances are stated once, in oracle/tolerances.py (checker infrastructure):
#   parity_err  -- against the reference's LSODA: |gpu - ref| <= 1e-8 |ref| + 5e-9 (LSODA's own absolute noise)
#   survey_err  -- against a tight solution: SURVEY.md section 8(d), |gpu - ref| <= 1e-8 max(|ref|, 1e-6 colmax)
#   project_tolerances -- first-order propagation of either through the assembly formulas
from oracle.tolerances import (PARITY_RTOL, PARITY_ATOL, parity_err, survey_err, survey_tol, tol_ratio,  # noqa: E402,F401
                               lsoda_taus, tight_taus, project_tolerances)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def check_parity(gpu, lsoda, tight=None, what='', criterion='survey'):
    """The parity check of the integration half, in this order:
      1. against the reference's LSODA result, |gpu - ref| <= 1e-8 |ref| + 5e-9 (parity_err <= 1): done;
      2. where that fails -- LSODA at rtol = atol = 1e-10 is itself several 1e-9 off in absolute terms, more on deep
         models -- the disagreement must be LSODA's: the GPU result has to meet SURVEY section 8(d)'s criterion against
         a TIGHT solution (survey_err <= 1) and be closer to it than LSODA is.
    ``tight``: array or zero-argument callable (computed only when needed).  ``criterion='parity'`` judges the GPU
    against the tight solution with the same formula as against LSODA (1e-8 |ref| + 5e-9) instead of section 8(d)'s:
    for the implicit integrator, whose default tolerance is set to the parity level, not beyond.
    Returns (err vs LSODA, err vs tight or None)."""
    e = parity_err(gpu, lsoda)
    if e <= 1.0:
        return e, None
    assert tight is not None, "%s: %.2f tolerance units off the reference and no tight solution to arbitrate" % (what, e)
    t = tight() if callable(tight) else tight
    f = survey_err if criterion == 'survey' else parity_err
    eg, el = f(gpu, t), f(lsoda, t)
    unit = ("units of SURVEY 8(d) incl. this build's block floor (1e-12 of the block's largest entry, oracle/tolerances.py)"
            if criterion == 'survey' else "units of 1e-8 |ref| + 5e-9")
    assert eg <= 1.0 and el > eg, ("%s: %.2f tolerance units off the reference's LSODA; against a tight solution the GPU "
                                   "is %.2f and LSODA %.2f %s" % (what, e, eg, el, unit))
    return e, eg


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope='session')
def zoo():
    from sysbio_modeling_amd.symbolic import zoo_model
    return zoo_model


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope='session')
def gpu_models(zoo):
    """OdeModel per zoo model, loaded on cuda:0 (gpu tests only)."""
    from sysbio_modeling_amd.model import OdeModel
    cache = {}

    def get(name):
        if name not in cache:
            gm = zoo(name)
            cache[name] = OdeModel(gm.model, gm.sens_model, gm.n_vars, gm.param_order, model_name=name)
        return cache[name]
    return get


@pytest.fixture(autouse=True)
def poison_register_file(request):
    """Before every GPU test, run a kernel that leaves NaNs in the register file of every CU: a value
    read before it is written then shows up as NaN instead of as the plausible-looking leftovers of
    the previous launch (this is how an uninitialised read in the row-lane kernel was caught)."""
    if request.node.get_closest_marker('gpu') is not None and has_gpu():
        import torch
        x = torch.full((64 * 1024 * 1024,), float('nan'), dtype=torch.float64, device='cuda')
        y = x * 2.0 + x
        torch.cuda.synchronize()
        del x, y
    yield
