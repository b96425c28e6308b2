"""Host-side boundary tests (CPU): the C-ABI library loads and exports every symbol of
include/sbm.h, the plugin data types keep the reference's behaviour
(tests/test_Experiment.py, test_Measurements.py, test_Project_Utils.py of the reference),
Project flattens settings into the index arrays the kernels consume, and nothing falls
back to a CPU path when the device is missing."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from sysbio_modeling_amd import _lib, build
from sysbio_modeling_amd.experiment import Experiment
from sysbio_modeling_amd.measurement import TimecourseMeasurement
from sysbio_modeling_amd.model import OdeModel
from sysbio_modeling_amd.project import Project
from sysbio_modeling_amd.project.utils import OrderedHashDict, sample_index, simulation_grid
from sysbio_modeling_amd.project.loss_functions import SquareLossFunction
from tests import reference_cases as rc
from tests.conftest import has_gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------
def _declared_functions():
    with open(os.path.join(REPO, 'include', 'sbm.h')) as fh:
        text = fh.read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sbm_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load_library()
    declared = _declared_functions()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), "libsbm_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "python binding has no signature for %s" % name
    assert sorted(_lib.SIGNATURES) == declared
    assert lib.sbm_abi_version() == _lib.ABI_VERSION == 4


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors vs the C compiler's view of include/sbm.h."""
    src = tmp_path / 'layout.c'
    src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "sbm.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sbm_integrator_opts), offsetof(sbm_integrator_opts, rtol),
         offsetof(sbm_integrator_opts, t0), sizeof(sbm_project_desc), offsetof(sbm_project_desc, pmap),
         offsetof(sbm_project_desc, reference_compat), sizeof(sbm_loss_desc), offsetof(sbm_loss_desc, row_data),
         offsetof(sbm_loss_desc, sf_prior_sigma), offsetof(sbm_project_desc, row_prog), offsetof(sbm_project_desc, row_time));
  return 0;
}''')
    exe = str(tmp_path / 'layout')
    import subprocess
    subprocess.check_call(['gcc', '-I', os.path.join(REPO, 'include'), str(src), '-o', exe])
    c = [int(x) for x in subprocess.check_output([exe]).split()]
    py = [ctypes.sizeof(_lib.IntegratorOpts), _lib.IntegratorOpts.rtol.offset, _lib.IntegratorOpts.t0.offset,
          ctypes.sizeof(_lib.ProjectDesc), _lib.ProjectDesc.pmap.offset, _lib.ProjectDesc.reference_compat.offset,
          ctypes.sizeof(_lib.LossDesc), _lib.LossDesc.row_data.offset, _lib.LossDesc.sf_prior_sigma.offset,
          _lib.ProjectDesc.row_prog.offset, _lib.ProjectDesc.row_time.offset]
    assert c == py


def test_plugins_export_plugin_abi():
    from sysbio_modeling_amd.symbolic import zoo_model
    for name in ('simple', 'michaelis_menten', 'cascade20'):
        path = zoo_model(name).plugin_path(build_if_missing=True)      # built by __graft_entry__.build(); here if missing
        assert path == build.plugin_path(name) and os.path.exists(path)
        out = subprocess.check_output(["nm", "-D", "--defined-only", path]).decode()
        assert 'sbm_plugin_info' in out and 'sbm_plugin_launch' in out


@pytest.mark.skipif(has_gpu(), reason="checks behaviour WITHOUT a device")
def test_no_cpu_fallback_without_device(zoo):
    """The product path fails loudly when it cannot reach the GPU."""
    gm = zoo('simple')
    m = OdeModel(gm.model, gm.sens_model, 1, gm.param_order, use_jit=False)
    with pytest.raises(_lib.SbmError):
        m.simulate(rc.SIMPLE_P, rc.SIMPLE_T10)


def test_odemodel_rejects_callables_it_cannot_compile():
    """Only code that unrolls to the reference's emitted straight-line form can be read from source (tests/test_ingest.py:
    static loops and branches do since round 4); control flow that depends on the data is refused -- there is no SciPy
    path to fall back on."""
    def f(y, t, yout, p):
        i = 0
        while y[i] > 0.0:
            yout[i] = -p[0] * y[i]
            i += 1
    with pytest.raises(TypeError, match="no CPU fallback"):
        OdeModel(f, None, 1, ['k'], use_jit=False)
    with pytest.raises(TypeError, match="no CPU fallback"):
        OdeModel(np.negative, None, 1, ['k'], use_jit=False)

    def g(y, t, yout, p):
        k = p[0]
        yout[0] = (-k * y[0])
    m = OdeModel(g, None, 1, ['k'], use_jit=False)       # the emitted form is compiled (nothing is built before use)
    assert m.n_vars == 1 and m.param_order == ['k'] and m.generated.spec.equations['y0'] is not None


def test_odemodel_signature_checks(zoo):
    gm = zoo('simple')
    with pytest.raises(ValueError):
        OdeModel(gm.model, gm.sens_model, 2, gm.param_order, use_jit=False)
    with pytest.raises(ValueError):
        OdeModel(gm.model, gm.sens_model, 1, ['k_synt', 'k_deg'], use_jit=False)
    with pytest.raises(ValueError):
        OdeModel(gm.model, gm.sens_model, 1, gm.param_order, use_jit=False, jit_type='llvm')
    m = OdeModel(gm.model, gm.sens_model, 1, gm.param_order, use_jit=False)
    assert m.n_vars == 1 and m.get_n_vars() == 1 and m.param_order == ['k_deg', 'k_synt']
    assert m.use_jac is True and m._jit_enabled is False and m.model_name == 'Model'
    # defaults: the explicit pair at the size-aware tolerance that meets SURVEY section 8(d) against a tight solution
    # (model/ode_model.py::default_tolerances), a step budget with early exit
    assert m.integrator_options == dict(method='dopri45', rtol=1e-9, atol=1e-18, max_steps=-50000)
    from sysbio_modeling_amd.model.ode_model import default_tolerances
    assert default_tolerances(40)['rtol'] == 1e-9 and default_tolerances(80)['rtol'] == pytest.approx(2.5e-10)
    o = _lib.make_opts(**m.integrator_options)
    assert (o.method, o.max_steps, o.variant) == (_lib.SBM_DOPRI45, -50000, 0)


def test_make_opts():
    o = _lib.make_opts('rk4', n_steps=100, t_end=50.0)
    assert o.method == _lib.SBM_RK4_FIXED and o.h0 == 0.5
    o = _lib.make_opts('dopri45', rtol=1e-7, atol=1e-9, t0=2.0)
    assert (o.method, o.rtol, o.atol, o.t0) == (_lib.SBM_DOPRI45, 1e-7, 1e-9, 2.0)
    with pytest.raises(ValueError):
        _lib.make_opts('lsoda')
    with pytest.raises(ValueError):
        _lib.make_opts('rk4')


# ---------------------------------------------------------------------------
# data types (reference tests/test_Measurements.py, tests/test_Experiment.py)
# ---------------------------------------------------------------------------
def test_measurement_validation():
    t = np.linspace(0, 10, 5)
    m = TimecourseMeasurement('A', np.arange(5.0), t)
    assert np.array_equal(m.std, np.ones(5))
    with pytest.raises(ValueError):
        TimecourseMeasurement('A', np.arange(5.0), t, np.array([1, 1, 0, 1, 1.0]))
    with pytest.raises(ValueError):
        TimecourseMeasurement('A', np.arange(5.0), t[:4])
    with pytest.raises(ValueError):
        TimecourseMeasurement('A', np.arange(5.0), t, np.ones(4))
    v, s, tp = m.get_nonzero_measurements()
    assert len(v) == 4 and tp[0] == 2.5 and len(m.values) == 5
    m.drop_timepoint_zero()
    assert len(m.values) == 4 and len(m.std) == 4 and m.timepoints[0] == 2.5


def test_experiment_behaviour():
    t = np.array([0.0, 1.0, 2.0])
    a = TimecourseMeasurement('b_var', np.ones(3), t)
    b = TimecourseMeasurement('a_var', np.ones(4), np.array([0.0, 0.5, 2.0, 3.0]))
    e = Experiment('exp1', [a, b], fixed_parameters={'k': 1.0}, experiment_settings={'s': 'x'})
    assert [m.variable_name for m in e.measurements] == ['a_var', 'b_var']      # kept sorted
    assert np.array_equal(e.get_unique_timepoints(), [0.5, 1.0, 2.0, 3.0])
    assert np.array_equal(e.get_unique_timepoints(include_zero=True), [0.0, 0.5, 1.0, 2.0, 3.0])
    assert e.get_variable_measurements('a_var') is b
    with pytest.raises(KeyError):
        e.get_variable_measurements('zzz')
    with pytest.raises(KeyError):
        e.add_measurement(TimecourseMeasurement('a_var', np.ones(3), t))
    with pytest.raises(ValueError):
        Experiment('_bad', a)
    single = Experiment('1ok', a)
    assert len(single.measurements) == 1 and single.settings == {} and single.param_global_vector_idx is None
    e.drop_timepoint_zero('a_var')
    assert len(b.timepoints) == 3 and len(a.timepoints) == 3
    e.drop_timepoint_zero()
    assert len(a.timepoints) == 2


def test_ordered_hash_dict_rules():
    """Key rules of tests/test_Project_Utils.py."""
    d = OrderedHashDict()
    d['a'] = 1
    d[frozenset(['b', 'c'])] = 2
    assert d['a'] == 1 and d['b'] == 2 and d['c'] == 2 and d[frozenset(['b', 'c'])] == 2
    assert 'b' in d and 'z' not in d
    with pytest.raises(KeyError):
        d['z']
    with pytest.raises(KeyError):
        d['b'] = 5                       # only through its group
    with pytest.raises(KeyError):
        d[frozenset(['a', 'q'])] = 3     # 'a' already present
    with pytest.raises(TypeError):
        d[5] = 1
    with pytest.raises(TypeError):
        d[frozenset([1, 2])] = 1
    d['a'] = 10
    d[frozenset(['b', 'c'])] = 20
    assert d['a'] == 10 and d['c'] == 20
    assert list(d.keys()) == ['a', frozenset(['b', 'c'])]


def test_loss_function_groups():
    lf = SquareLossFunction(['Variable_1'])
    assert 'Variable_1' in lf.scale_factors                                  # test_Loss_Functions.py:82-86
    lf = SquareLossFunction(frozenset(['Variable_1', 'Variable_2']))
    assert 'Variable_1' in lf.scale_factors and 'Variable_2' in lf.scale_factors   # :88-92
    assert lf.group_index('Variable_2') == 0 and lf.group_index('other') == -1
    lf = SquareLossFunction(sf_groups=['Lin', 'Square'])
    lf.set_scale_factor_priors('Lin', 1.0, 2.0)
    assert lf.scale_factors['Lin'].log_prior == 1.0 and lf.scale_factors['Lin'].log_prior_sigma == 2.0   # :210-211
    with pytest.raises(KeyError):
        lf.set_scale_factor_priors('Cube', 1.0, 2.0)
    assert hasattr(SquareLossFunction, 'scale_factors')                      # what Project tests (base_project.py:69)


def test_sampling_quirk():
    t = simulation_grid(100.0)
    assert len(t) == 1000 and t[0] == 0 and t[-1] == 100.0
    idx = sample_index(t, np.array([50.0, 100.0, 11.11111111]))
    assert t[idx[0]] == pytest.approx(50.05005005005005) and idx[1] == 999 and t[idx[2]] >= 11.11111111


# ---------------------------------------------------------------------------
# Project: settings -> index arrays (no device needed)
# ---------------------------------------------------------------------------
@pytest.fixture()
def simple_project(zoo):
    gm = zoo('simple')
    model = OdeModel(gm.model, gm.sens_model, 1, gm.param_order, use_jit=False)
    exps, settings, mapping, sf = rc.simple_project_case()
    return Project(model, exps, settings, mapping, sf_groups=sf)


def test_project_indexing_matches_reference_case(simple_project):
    proj = simple_project
    assert list(proj.get_param_index('k_synt').keys()) == ['Global']          # test_Project.py:88-90
    assert len(proj.get_param_index('Group_1')) == 2
    idx = proj.project_param_idx
    for exp in proj.experiments:
        assert idx['Group_1'][(exp.settings['Deg_Rate'],)] == exp.param_global_vector_idx['k_deg']
    assert proj.n_project_residuals == 35 and proj.n_project_params == 3       # :94
    # experiments sorted by name: High_Deg_Exp first -> its k_deg gets slot 1, Low 2, k_synt (Global) 0
    assert idx == {'k_synt': {'Global': 0}, 'Group_1': {('High',): 1, ('Low',): 2}}
    d = proj.descriptor_arrays()
    assert (d['E'], d['q'], d['R'], d['G']) == (2, 3, 35, 1)
    assert d['pmap'].tolist() == [[1, 0], [2, 0]]                              # param_order = [k_deg, k_synt]
    assert d['sens_col'].tolist() == [0, 1]
    assert d['row_exp'].tolist() == [0] * 26 + [1] * 9
    assert np.all(d['row_sf'] == 0) and np.all(d['row_vars'] == 0)
    assert d['tgrid_off'].tolist() == [0, 26, 35]
    # every row samples the first grid point at or after its measurement time
    for r in range(35):
        e = d['row_exp'][r]
        t_s = d['tgrid'][d['tgrid_off'][e] + d['row_tidx'][r]]
        t_m = proj._rows['t_meas'][r]
        grid = simulation_grid(proj._experiments[e].get_unique_timepoints()[-1])
        assert t_s == grid[np.searchsorted(grid, t_m)] and t_s >= t_m
    assert d['reference_compat'] == 1


def test_project_rows_follow_reference_order(zoo):
    gm = zoo('michaelis_menten')
    model = OdeModel(gm.model, gm.sens_model, 2, gm.param_order, use_jit=False)
    t = np.array([0.0, 10.0, 20.0])
    e_b = Experiment('B', [TimecourseMeasurement('Prod', np.array([1., 2., 3.]), t),
                           TimecourseMeasurement('Both', np.array([4., 5., 6.]), t)],
                     fixed_parameters={'km': 0.5})
    e_a = Experiment('A', [TimecourseMeasurement('Prod', np.array([7., 8.]), np.array([5.0, 20.0]))])
    with pytest.warns(UserWarning, match="global because no settings"):
        proj = Project(model, [e_b, e_a], {'Local': ['vmax']}, {'Prod': ('direct', 1), 'Both': ('sum', [0, 1])},
                       sf_groups=['Prod'])
    labels = proj.row_index()
    assert labels == [('A', 'Prod')] * 2 + [('B', 'Both')] * 2 + [('B', 'Prod')] * 2   # t = 0 rows dropped
    d = proj.descriptor_arrays()
    assert d['row_data'].tolist() == [7., 8., 5., 6., 2., 3.]
    assert d['row_var_off'].tolist() == [0, 1, 2, 4, 6, 7, 8] and d['row_vars'].tolist() == [1, 1, 0, 1, 0, 1, 1, 1]
    assert d['row_sf'].tolist() == [0, 0, -1, -1, 0, 0]
    # globals first (km, k_synt_s, k_deg_s, k_deg_p in model order), then locals per experiment
    assert proj.n_project_params == 6
    km = gm.param_order.index('km')
    assert d['pmap'][1, km] == -1 and d['pfixed'][1, km] == 0.5 and d['pmap'][0, km] == 0
    assert proj.get_param_index('vmax_A') == {'Local': 4} and proj.get_param_index('vmax_B') == {'Local': 5}


def test_project_validation_errors(zoo):
    gm = zoo('simple')
    model = OdeModel(gm.model, gm.sens_model, 1, gm.param_order, use_jit=False)
    exps, settings, mapping, sf = rc.simple_project_case()
    with pytest.raises(ValueError):
        Project(model, exps, settings, {'Variable_1': ('direct', 3)})
    with pytest.raises(ValueError):
        Project(model, exps, settings, {'Variable_1': ('weird', 0)})
    with pytest.raises(ValueError):
        Project(model, exps, {'Fixed': ['k_deg'], 'Global': ['k_synt']}, mapping)   # no value provided
    proj = Project(model, exps, settings, mapping, sf_groups=sf)
    with pytest.raises(KeyError):
        proj.add_experiment(exps[0])                                            # duplicate name
    with pytest.raises(KeyError):
        proj.remove_experiments_by_settings({'Absent:', 5})                     # test_Project.py:215-218
    with pytest.raises(KeyError):
        proj.remove_experiments_by_settings({'Deg_Rate': 'Nope'})
    with pytest.raises(KeyError):
        proj.set_parameter_log_prior('k_synt', 'Nope', 0.0, 1.0)
    with pytest.raises(ValueError):
        proj.get_simulations()


def test_project_add_remove_and_priors(simple_project):
    proj = simple_project
    t = np.linspace(0, 100, 10)
    extra = Experiment('Simple_Experiment', TimecourseMeasurement('Variable_1', np.ones(10), t),
                       experiment_settings={'Deg_Rate': 'Very High'})
    proj.add_experiment(extra)                                                  # test_Project.py:220-252
    assert any(e.name == 'Simple_Experiment' for e in proj.experiments)
    assert proj.n_project_params == 4 and proj.n_project_residuals == 44
    removed = proj.remove_experiments_by_settings({'Deg_Rate': 'Very High'})
    assert [e.name for e in removed] == ['Simple_Experiment'] and proj.n_project_params == 3
    proj.set_parameter_log_prior('k_synt', 'Global', np.log(0.01), 0.5)
    proj.set_scale_factor_log_prior('Variable_1', np.log(3.0), 0.1)
    d = proj.descriptor_arrays()
    assert d['prior_idx'].tolist() == [0] and d['prior_sigma'].tolist() == [0.5]
    assert d['sf_prior_group'].tolist() == [0] and proj.n_total_rows == 37
    assert proj.row_index(True)[-2:] == [('~Prior', 'k_synt Global'), ('~~SF_Prior', '~Variable_1')]
    mdf = proj.measurements_df
    assert list(mdf.columns) == ['mean', 'std', 'timepoints'] and len(mdf) == 37
    vec = proj.project_param_dict_to_vect({'k_synt': {'Global': 2.0}, 'Group_1': {('Low',): 3.0}}, default_value=-1)
    assert vec.tolist() == [2.0, -1.0, 3.0]
    assert proj.project_param_vect_to_dict(vec)['Group_1'][('High',)] == -1.0
    assert set(proj.group_experiments(['Deg_Rate']).keys()) == {('High',), ('Low',)}


def test_prebuilt_targets_are_used_on_a_box_without_hipcc(tmp_path, monkeypatch):
    """A box that only carries prebuilt .so files (no hipcc on PATH, HIPCC unset): an existing target is used as it
    is -- no rebuild attempt, no error -- and a missing one raises BuildError.  A hipcc that does not answer
    --version must not silently change the stamp either."""
    from sysbio_modeling_amd import build
    target = tmp_path / 'libdummy.so'
    target.write_bytes(b'prebuilt')
    (tmp_path / 'libdummy.so.stamp').write_text('stamp-of-the-box-that-built-it')
    src = tmp_path / 'dummy.hip'
    src.write_text('// nothing')
    calls = []

    def cmd_for(out):
        calls.append(out)
        return [build.hipcc_path(), '-shared', str(src), '-o', out]
    monkeypatch.setattr(build, '_hipcc_id', None)
    monkeypatch.setattr(build, 'hipcc_path', lambda: (_ for _ in ()).throw(build.BuildError("hipcc not found")))
    assert build._locked_build(str(target), cmd_for, [str(src)], 'dummy') == str(target)
    assert target.read_bytes() == b'prebuilt'
    with pytest.raises(build.BuildError):
        build._locked_build(str(tmp_path / 'missing.so'), cmd_for, [str(src)], 'dummy')
    with pytest.raises(build.BuildError):
        build._locked_build(str(target), cmd_for, [str(src)], 'dummy', force=True)
    # a compiler that exists but fails to report its version: same treatment, and nothing is cached
    fake = tmp_path / 'hipcc'
    fake.write_text('#!/bin/sh\nexit 3\n')
    fake.chmod(0o755)
    monkeypatch.setattr(build, 'hipcc_path', lambda: str(fake))
    monkeypatch.setattr(build, '_hipcc_id', None)
    assert build._locked_build(str(target), cmd_for, [str(src)], 'dummy') == str(target)
    assert build._hipcc_id is None and target.read_bytes() == b'prebuilt'


def test_explicit_pair_prediction(zoo):
    """_lib.predict_explicit_pair (what explicit_method='auto' asks): DOP853 at tight tolerances for every batch size --
    1.9x with the chip full, ~4x for a single vector, the measured figures of DESIGN.md section 5 -- DOPRI45 from rtol
    1e-5 up once the chip is full; the row-group splits' wavefront counts come from the generated header."""
    ch = zoo('cascade20').rowgroup_chunks()
    assert ch == {'RG0': 1, 'RG1': 5, 'RG2': 2}
    for n in (1, 256, 2048, 4096, 65536):
        pair, ratio = _lib.predict_explicit_pair(1e-9, n, ch['RG0'], ch['RG2'])
        assert pair == 'dop853' and (1.9 < ratio < 2.05 if n > 1024 else 3.8 < ratio < 4.0)
    assert _lib.predict_explicit_pair(1e-5, 8192, 1, 2)[0] == 'dopri45'
    assert _lib.predict_explicit_pair(1e-5, 1, 1, 2)[0] == 'dop853'
    assert zoo('michaelis_menten').rowgroup_chunks().get('RG0') == 1
