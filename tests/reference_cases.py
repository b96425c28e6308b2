"""Known-answer DATA of the reference's own tests, restated as inputs/expected outputs.

Numbers only (measurement arrays, parameter values, golden vectors); every block cites
the reference test that holds it.  Builders return this build's Experiment /
Measurement / settings objects so that the same case can be fed to the oracle and to
the GPU path.
"""
import numpy as np

from sysbio_modeling_amd.experiment import Experiment
from sysbio_modeling_amd.measurement import TimecourseMeasurement

# tests/test_OdeModel.py:21-26 -- 1-state model, p = (k_deg, k_synt) = (0.001, 0.01)
SIMPLE_P = np.array([0.001, 0.01])
SIMPLE_T10 = np.linspace(0, 100, 10)
SIMPLE_Y10_GOLDEN = np.array([0, 0.11049612, 0.21977129, 0.32783901, 0.43471262, 0.54040533,
                              0.64493016, 0.74830004, 0.85052773, 0.95162583])


def simple_closed_form(k_deg, k_synt, t):
    """y, dy/dk_deg, dy/dk_synt of y' = k_synt - k_deg y, y(0) = 0.
    (The d/dk_deg expression in tests/test_OdeModel.py:48-50 is mis-parenthesised and does
    not hold for the reference's own output -- SURVEY.md section 4; this is the correct form.)"""
    e = np.exp(-k_deg * t)
    y = k_synt * (1 - e) / k_deg
    dy_dksynt = (1 - e) / k_deg
    dy_dkdeg = k_synt * (t * e / k_deg - (1 - e) / k_deg ** 2)
    return y, dy_dkdeg, dy_dksynt


# tests/test_Project.py:29-64 -- two experiments sharing k_synt, k_deg depending on 'Deg_Rate'
LOW_DEG_T = np.array([0., 11.11111111, 22.22222222, 33.33333333, 44.44444444, 55.55555556,
                      66.66666667, 77.77777778, 88.88888889, 100.])
LOW_DEG_VALUES = np.array([0., 0.11049608, 0.21977125, 0.32783897, 0.43471258, 0.54995561,
                           0.65437492, 0.75764044, 0.85976492, 0.9516258]) * 3.75
HIGH_DEG_T = np.array([5.05050505, 9.09090909, 12.12121212, 16.16161616, 19.19191919,
                       23.23232323, 26.26262626, 30.3030303, 33.33333333, 37.37373737,
                       40.4040404, 44.44444444, 47.47474747, 50.50505051, 54.54545455,
                       57.57575758, 61.61616162, 64.64646465, 68.68686869, 71.71717172,
                       75.75757576, 78.78787879, 82.82828283, 85.85858586, 89.8989899,
                       92.92929293])
HIGH_DEG_VALUES = np.array([0.04925086, 0.08689927, 0.11415396, 0.14923229, 0.17462643, 0.20731014,
                            0.23097075, 0.26142329, 0.28346868, 0.31184237, 0.33238284, 0.3588196,
                            0.37795787, 0.39652489, 0.42042171, 0.43772125, 0.45998675, 0.47610533,
                            0.49685087, 0.51186911, 0.53119845, 0.54519147, 0.56320128, 0.57623905,
                            0.59301942, 0.60516718]) * 3.75


def simple_project_case():
    """Experiments, settings, mapping and sf_groups of tests/test_Project.py:27-72."""
    m1 = TimecourseMeasurement('Variable_1', LOW_DEG_VALUES.copy(), LOW_DEG_T.copy())
    low = Experiment('Low_Deg_Exp', m1, experiment_settings={'Deg_Rate': 'Low'})
    m2 = TimecourseMeasurement('Variable_1', HIGH_DEG_VALUES.copy(), HIGH_DEG_T.copy())
    high = Experiment('High_Deg_Exp', m2, experiment_settings={'Deg_Rate': 'High'})
    # tests/test_utils/simple_model_settings.py:3-5
    settings = {'Global': ['k_synt'], 'Shared': {'Group_1': {'k_deg': ('Deg_Rate',)}}}
    mapping = {'Variable_1': ('direct', 0)}
    sf_groups = [frozenset(['Variable_1'])]
    return [low, high], settings, mapping, sf_groups


def simple_project_theta(get_param_index):
    """log of (k_deg High 0.01, k_deg Low 0.001, k_synt 0.01), tests/test_Project.py:74-84."""
    theta = np.zeros(3)
    theta[get_param_index('Group_1', ('High',))] = 0.01
    theta[get_param_index('Group_1', ('Low',))] = 0.001
    theta[get_param_index('k_synt', 'Global')] = 0.01
    return np.log(theta)


def simple_model_analytical_jac(k_deg, k_synt, t):
    """tests/test_Project.py:19-23: d y / d log(k) rows (k_deg, k_synt)."""
    k_synt_jac = k_synt * (1 / k_deg - np.exp(-k_deg * t) / k_deg)
    k_deg_jac = k_deg * (k_synt * t * np.exp(-k_deg * t) / k_deg - k_synt / k_deg ** 2 +
                         k_synt * np.exp(-k_deg * t) / k_deg ** 2)
    return np.vstack((k_deg_jac, k_synt_jac))


# tests/test_Project.py:283-300 -- Michaelis-Menten, 'sum' mapping of both species
MM_PARAMS = np.array([1e-3, 0.001, 0.01, 0.01, 0.001])  # vmax, km, k_synt_s, k_deg_s, k_deg_p
MM_T = np.linspace(0, 100, 20)


def central_fd_jacobian(f, x, eps=None):
    """Central finite differences, the role statsmodels' approx_fprime(centered=True) plays in
    tests/test_Project.py:165,175,331,346 (step = eps^(1/3) * max(|x|, 0.1), its default)."""
    x = np.asarray(x, dtype=float)
    f0 = np.asarray(f(x))
    J = np.zeros((f0.size, x.size))
    for k in range(x.size):
        h = (np.finfo(float).eps ** (1.0 / 3)) * max(abs(x[k]), 0.1) if eps is None else eps
        xp, xm = x.copy(), x.copy()
        xp[k] += h
        xm[k] -= h
        J[:, k] = (np.asarray(f(xp)) - np.asarray(f(xm))).ravel() / (2 * h)
    return J
