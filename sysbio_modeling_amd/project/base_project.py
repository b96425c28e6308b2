"""Project: experiments + model + loss -> residuals / Jacobian, evaluated on the GPU.

Drop-in for the hot path of the reference's ``Project`` (project/base_project.py):
same constructor, same public objective functions (``residuals`` :708,
``calc_project_jacobian`` :731, ``calc_rss_gradient`` :773,
``calc_sum_square_residuals`` :807, ``nlopt_fcn`` :829), same parameter-indexing
rules (``_set_local_param_idx`` :164-276) and residual row order (:296-341).

What the reference does per call with Python loops, pandas frames and one
``odeint`` per experiment is flattened ONCE here into index arrays
(``sbm_project_desc``, include/sbm.h) and evaluated for a whole ensemble of
parameter vectors by three kernels: theta->p gather, the integrator, and the
fused sample / residual / Jacobian assembly.  ``*_batch`` methods take
(V, n_project_params) arrays; the reference-named methods are the V = 1 case.

Out of scope (reference :974-1078): plotting and pretty-printing.
"""
from __future__ import annotations

import copy
import ctypes
import warnings
from collections import OrderedDict, defaultdict

import numpy as np

from . import utils
from .. import _lib
from .. import _control
from .loss_functions.squared_loss import SquareLossFunction


def _as_int32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Project(object):
    """Combine a model, experiments and a loss function into one objective.

    Parameters (reference project/base_project.py:32-58)
    ----------
    model : OdeModel
    experiments : list of Experiment
    model_parameter_settings : dict with optional keys 'Global' (list), 'Fixed' (list),
        'Local' (list), 'Shared' ({group: {param: (setting, ...) or None}})
    measurement_to_model_map : {measure_name: ('direct', var_index) | ('sum', [var_index, ...])}
    sf_groups : list of measure names / frozensets sharing one scale factor
    loss_function : SquareLossFunction (class)
    reference_compat : keep the reference's tested quirks -- Jacobian not divided by
        sigma, zero parameter-prior rows (see include/sbm.h)
    """

    def __init__(self, model, experiments, model_parameter_settings, measurement_to_model_map,
                 sf_groups=None, loss_function=SquareLossFunction, reference_compat=True):
        self.project_description = ""
        self._model = model
        self._model_parameter_settings = model_parameter_settings
        self.reference_compat = bool(reference_compat)

        if hasattr(loss_function, 'scale_factors'):
            self._loss_function = loss_function(sf_groups)
        elif sf_groups is not None:
            raise ValueError("Loss Function %s does not support scale factors" % type(loss_function))
        else:
            self._loss_function = loss_function

        self._parameter_priors = OrderedDict()
        self._scale_factor_priors = []

        self._measurement_to_model_map = {}
        self._measurement_to_model_map_raw = dict(measurement_to_model_map)   # as given (introspection)
        self._make_mapping(measurement_to_model_map)

        self._device_project = None
        self._experiments = []
        self._project_param_idx = None
        self._n_project_params = None
        self._residuals_per_param = None
        self._n_residuals = None
        self._rows = None
        self._last = {}
        # the project's OWN overrides: the model's integrator_options are read at call time (so that options set
        # on the model after the project was built are inherited), these go on top
        self.integrator_options = {}
        self.add_experiment(experiments)
        self._project_param_vector = np.zeros((self.n_project_params,))

    # ------------------------------------------------------------------
    # mapping / settings  (reference :109-162)
    # ------------------------------------------------------------------
    def _make_mapping(self, measurement_to_model_map):
        for measure_name, (mapping_type, mapping_args) in measurement_to_model_map.items():
            if mapping_type == 'direct':
                if not isinstance(mapping_args, (int, np.integer)):
                    raise TypeError("'direct' mapping of %s needs an int variable index" % measure_name)
                if mapping_args >= self._model.n_vars or mapping_args < 0:
                    raise ValueError('Index (%d) has to be smaller than %d' % (mapping_args, self._model.n_vars))
                variables = [int(mapping_args)]
            elif mapping_type == 'sum':
                if not isinstance(mapping_args, (list, tuple)):
                    raise TypeError("'sum' mapping of %s needs a list of variable indices" % measure_name)
                variables = [int(v) for v in mapping_args]
                for v in variables:
                    if v >= self._model.n_vars or v < 0:
                        raise ValueError('Index (%d) has to be smaller than %d' % (v, self._model.n_vars))
            elif mapping_type == 'custom':
                # the reference takes Python callbacks here (:125-128); the kernel takes a compiled expression
                from .observables import compile_observable
                gm = getattr(self._model, 'generated', None)
                names = list(gm.spec.variables) if gm is not None else ['y%d' % i for i in range(self._model.n_vars)]
                if isinstance(mapping_args, (tuple, list)) and len(mapping_args) == 3 and callable(mapping_args[1]):
                    # the reference's own form (parameters, map_fn, jacobian_map_fn): read off the callbacks once the
                    # experiments are known (they are run, traced, on an experiment's grid: observables.py)
                    self._measurement_to_model_map[measure_name] = {'type': 'custom', 'variables': [], 'program': None,
                                                                    'callbacks': tuple(mapping_args), 'names': names}
                    continue
                comp = compile_observable(mapping_args, names)
                self._measurement_to_model_map[measure_name] = {'type': 'custom', 'variables': list(comp['variables']),
                                                                'program': comp}
                continue
            else:
                raise ValueError('Invalid mapping type')
            self._measurement_to_model_map[measure_name] = {'type': mapping_type, 'variables': variables}

    def _resolve_callback_mappings(self):
        """'custom' mappings given as the reference's callbacks: traced and compiled on first sight of an experiment
        that carries the measure (observables.trace_callback_observable)."""
        from .observables import ObservableError, compile_observable, trace_callback_observable
        import sympy
        for measure_name, m in self._measurement_to_model_map.items():
            if m['type'] != 'custom' or not m.get('callbacks'):
                continue
            # The reference hands `experiment` and `measurement` to the callback on EVERY call (base_project.py:380-383): a
            # callback may branch on them.  The kernel evaluates ONE program per measure, so the callback is traced on every
            # experiment / measurement that carries the measure and the traces must agree; one that does not (a
            # per-experiment weight, say) is refused rather than evaluated with the first experiment's formula.
            first = None
            for experiment in self._experiments:
                for meas in [x for x in experiment.measurements if x.variable_name == measure_name]:
                    par, fn, jac = m['callbacks']
                    expr = trace_callback_observable(par, fn, jac, m['names'], experiment, meas,
                                                     len(experiment.param_global_vector_idx))
                    if first is None:
                        first = (expr, experiment.name)
                    elif sympy.simplify(sympy.sympify(expr) - sympy.sympify(first[0])) != 0:
                        raise ObservableError(
                            "'custom' mapping of %s: the callback computes %s on experiment %s and %s on experiment %s; one "
                            "compiled expression serves every experiment -- give the mapping as an expression, or one "
                            "measure name per formula" % (measure_name, first[0], first[1], expr, experiment.name))
            if first is not None:
                comp = compile_observable(first[0], m['names'])
                m['variables'], m['program'] = list(comp['variables']), comp

    def _update_project_settings(self):
        self._project_param_idx, self._n_project_params, self._residuals_per_param = self._set_local_param_idx()
        self._resolve_callback_mappings()
        self._n_residuals = self._update_n_residuals()
        self._rows = self._build_rows()
        self._drop_device_project()
        self._last = {}

    def _drop_device_project(self):
        if self._device_project is not None:
            _lib.load_library().sbm_project_unload(self._device_project)
            self._device_project = None

    def __del__(self):
        try:
            self._drop_device_project()
        except Exception:
            pass

    def _set_local_param_idx(self):
        """Model parameter -> slot of the project vector, per experiment.

        Global parameters take the first slots in list order; a Shared group gets one
        slot per distinct tuple of the settings it depends on, allocated in
        experiment-name order; Local parameters one slot per experiment; Fixed ones none.
        """
        settings = self._model_parameter_settings
        all_model_parameters = set(self._model.param_order)
        local_pars = list(settings.get('Local', []))
        project_fixed_pars = list(settings.get('Fixed', []))
        global_pars = list(settings.get('Global', []))
        shared_pars_groups = settings.get('Shared', {})
        shared_pars = set(p for grp in shared_pars_groups for p in shared_pars_groups[grp])

        no_settings_params = (all_model_parameters - set(local_pars) - set(project_fixed_pars) -
                              shared_pars - set(global_pars))
        if no_settings_params:
            # the reference prints a notice and appends them in set order (:193-197); model
            # order makes the layout reproducible
            ordered = [p for p in self._model.param_order if p in no_settings_params]
            warnings.warn("The following parameters are global because no settings were specified: %s"
                          % ", ".join(ordered))
            global_pars.extend(ordered)

        project_param_idx = OrderedDict()
        residuals_per_param = defaultdict(lambda: defaultdict(int))
        n_params = 0
        for p in global_pars:
            project_param_idx[p] = {'Global': n_params}
            n_params += 1

        for experiment in self._experiments:
            exp_param_idx = OrderedDict()
            exp_fixed_pars = list(experiment.fixed_parameters.keys()) if experiment.fixed_parameters else []
            all_fixed_pars = project_fixed_pars + exp_fixed_pars
            n_t = len(experiment.get_unique_timepoints())
            for p in project_fixed_pars:
                if p not in exp_fixed_pars:
                    raise ValueError('%s was declared as a fixed parameter, but in experiment %s no value provided'
                                     % (p, experiment.name))
            for p in global_pars:
                if p not in all_fixed_pars:
                    exp_param_idx[p] = project_param_idx[p]['Global']
                    residuals_per_param[p]['Global'] += n_t
            for p_group in shared_pars_groups:
                for p, dep in shared_pars_groups[p_group].items():
                    if p in exp_fixed_pars:
                        continue
                    if dep is None:
                        key = 'None'
                    else:
                        try:
                            key = tuple(experiment.settings[s] for s in tuple(dep))
                        except KeyError as err:
                            raise KeyError('%s is not a setting in experiment %s' % (err.args[0], experiment.name))
                    slots = project_param_idx.setdefault(p_group, {})
                    if key not in slots:
                        slots[key] = n_params
                        n_params += 1
                    exp_param_idx[p] = slots[key]
                    residuals_per_param[p_group][key] += n_t
            for p in local_pars:
                if p in exp_fixed_pars:
                    continue
                par_string = '%s_%s' % (p, experiment.name)
                project_param_idx[par_string] = {'Local': n_params}  # the reference raises KeyError here (:270)
                exp_param_idx[p] = n_params
                n_params += 1
                residuals_per_param[par_string]['Local'] += n_t
            for p in self._model.param_order:
                if p not in exp_param_idx and p not in exp_fixed_pars:
                    raise KeyError('%s not in %s fixed parameters.' % (p, experiment.name))
            experiment.param_global_vector_idx = exp_param_idx
        return project_param_idx, n_params, residuals_per_param

    def _update_n_residuals(self, include_zero=False):
        n_res = 0
        for experiment in self._experiments:
            for measurement in experiment.measurements:
                tp = measurement.timepoints
                n_res += len(tp) if include_zero else int(np.count_nonzero(tp != 0))
        return n_res

    # ------------------------------------------------------------------
    # row table = the reference's measurement frame (:296-341), as arrays
    # ------------------------------------------------------------------
    def _build_rows(self):
        exp_idx, names, data, sigma, t_meas = [], [], [], [], []
        for ei, experiment in enumerate(self._experiments):
            for measurement in experiment.measurements:
                if measurement.variable_name not in self._measurement_to_model_map:
                    raise KeyError("no measurement_to_model_map entry for %r" % measurement.variable_name)
                vals, std, tps = measurement.get_nonzero_measurements()
                exp_idx.extend([ei] * len(vals))
                names.extend([measurement.variable_name] * len(vals))
                data.extend(vals)
                sigma.extend(std)
                t_meas.extend(tps)
        rows = {'exp': _as_int32(exp_idx), 'measure': list(names), 'data': _as_f64(data),
                'sigma': _as_f64(sigma), 't_meas': _as_f64(t_meas)}
        # grids: per experiment, the points of linspace(0, t_end, 1000) the rows sample
        tidx = np.zeros(len(exp_idx), dtype=np.int32)
        grids, t_sampled = [], np.zeros(len(exp_idx))
        for ei, experiment in enumerate(self._experiments):
            t_sim = utils.simulation_grid(experiment.get_unique_timepoints()[-1])
            sel = np.flatnonzero(rows['exp'] == ei)
            gi = utils.sample_index(t_sim, rows['t_meas'][sel])
            if np.any(gi >= len(t_sim)):
                raise ValueError("measurement time beyond the simulation grid in experiment %s" % experiment.name)
            uniq, inv = np.unique(gi, return_inverse=True)
            grids.append(t_sim[uniq])
            tidx[sel] = inv
            t_sampled[sel] = t_sim[gi]
        rows['tidx'] = tidx
        rows['grids'] = grids
        rows['t_sim'] = t_sampled
        rows['sf'] = _as_int32([self._sf_group_of(nm) for nm in names])
        return rows

    def _sf_group_of(self, measure_name):
        lf = self._loss_function
        return lf.group_index(measure_name) if hasattr(lf, 'group_index') else -1

    def _prior_rows(self):
        idx, mean, sigma, labels = [], [], [], []
        for p_group in self._parameter_priors:
            for sett, (mu, sg) in self._parameter_priors[p_group].items():
                idx.append(self._project_param_idx[p_group][sett])
                mean.append(mu)
                sigma.append(sg)
                labels.append(("~Prior", p_group + ' ' + ''.join(str(s) for s in sett)))
        return _as_int32(idx), _as_f64(mean), _as_f64(sigma), labels

    def _sf_prior_rows(self):
        grp, mean, sigma, labels = [], [], [], []
        for measure in self._scale_factor_priors:
            sf = self._loss_function._scale_factors[measure]
            name = measure if isinstance(measure, str) else next(iter(measure))
            grp.append(self._loss_function.group_index(name))
            mean.append(sf.log_prior)
            sigma.append(sf.log_prior_sigma)
            labels.append(("~~SF_Prior", "~%s" % name))
        return _as_int32(grp), _as_f64(mean), _as_f64(sigma), labels

    # ------------------------------------------------------------------
    # flatten -> sbm_project_desc
    # ------------------------------------------------------------------
    def descriptor_arrays(self):
        """The arrays of ``sbm_project_desc`` (include/sbm.h) as a dict of numpy arrays."""
        m = self._model
        NP = len(m.param_order)
        E = len(self._experiments)
        pmap = -np.ones((E, NP), dtype=np.int32)
        pfixed = np.zeros((E, NP))
        for ei, experiment in enumerate(self._experiments):
            for k, name in enumerate(m.param_order):
                if name in experiment.param_global_vector_idx:
                    pmap[ei, k] = experiment.param_global_vector_idx[name]
                else:
                    pfixed[ei, k] = experiment.fixed_parameters[name]
        sens_params = list(getattr(m, 'sens_params', m.param_order))
        sens_col = _as_int32([sens_params.index(n) if n in sens_params else -1 for n in m.param_order])
        rows = self._rows
        tgrid_off = np.zeros(E + 1, dtype=np.int32)
        for ei, g in enumerate(rows['grids']):
            tgrid_off[ei + 1] = tgrid_off[ei] + len(g)
        tgrid = np.concatenate(rows['grids']) if rows['grids'] else np.zeros(0)
        var_off = [0]
        var_list = []
        for nm in rows['measure']:
            var_list.extend(self._measurement_to_model_map[nm]['variables'])
            var_off.append(len(var_list))
        p_idx, p_mean, p_sigma, _ = self._prior_rows()
        s_grp, s_mean, s_sigma, _ = self._sf_prior_rows()
        groups = self._loss_function.groups if hasattr(self._loss_function, 'groups') else []
        lf = self._loss_function
        sigma = rows['sigma']
        if getattr(lf, 'normalize_sigma_by_mean', False):
            sigma = _as_f64(sigma * rows['data'])   # normalized_squared_loss_function.py:40-46
        loss_type = int(getattr(lf, 'loss_type', 0))
        if loss_type == 1 and len(rows['data']) and np.min(rows['data']) <= 0:
            raise ValueError("LogSquare loss cannot handle measurements smaller or equal to zero")
        from .observables import program_tables
        progs = program_tables(rows['measure'], {nm: m['program'] for nm, m in self._measurement_to_model_map.items()
                                                 if m['type'] == 'custom' and nm in set(rows['measure'])})
        return dict(row_time=_as_f64(rows['t_sim']), **progs,
                    E=E, q=self._n_project_params, R=len(rows['exp']), G=len(groups), loss_type=loss_type,
                    pmap=pmap, pfixed=pfixed, sens_col=sens_col, tgrid_off=tgrid_off, tgrid=_as_f64(tgrid),
                    row_exp=rows['exp'], row_tidx=rows['tidx'], row_var_off=_as_int32(var_off),
                    row_vars=_as_int32(var_list), row_data=rows['data'], row_sigma=sigma,
                    row_sf=rows['sf'], prior_idx=p_idx, prior_mean=p_mean, prior_sigma=p_sigma,
                    sf_prior_group=s_grp, sf_prior_mean=s_mean, sf_prior_sigma=s_sigma,
                    reference_compat=int(self.reference_compat))

    def _device(self):
        """Load (once per settings change) the flattened project onto the model's device."""
        if self._device_project is not None:
            return self._device_project
        lib = _lib.load_library()
        a = self.descriptor_arrays()
        self._desc_keepalive = a

        def ip(x):
            return x.ctypes.data_as(_lib.c_int32_p) if x.size else ctypes.cast(None, _lib.c_int32_p)

        def dp(x):
            return x.ctypes.data_as(_lib.c_double_p) if x.size else ctypes.cast(None, _lib.c_double_p)

        desc = _lib.ProjectDesc(
            a['E'], a['q'], a['R'], a['G'], len(a['prior_idx']), len(a['sf_prior_group']),
            ip(a['pmap']), dp(a['pfixed']), ip(a['sens_col']), ip(a['tgrid_off']), dp(a['tgrid']),
            ip(a['row_exp']), ip(a['row_tidx']), ip(a['row_var_off']), ip(a['row_vars']),
            dp(a['row_data']), dp(a['row_sigma']), ip(a['row_sf']),
            ip(a['prior_idx']), dp(a['prior_mean']), dp(a['prior_sigma']),
            ip(a['sf_prior_group']), dp(a['sf_prior_mean']), dp(a['sf_prior_sigma']),
            a['reference_compat'], a['loss_type'],
            a['n_programs'], len(a['prog_code']), len(a['prog_const']),
            ip(a['row_prog']) if a['n_programs'] else ctypes.cast(None, _lib.c_int32_p), ip(a['prog_nvars']),
            ip(a['prog_sub_off']) if a['n_programs'] else ctypes.cast(None, _lib.c_int32_p), ip(a['prog_code']),
            dp(a['prog_const']), dp(a['row_time']) if a['n_programs'] else ctypes.cast(None, _lib.c_double_p))
        h = ctypes.c_void_p()
        _lib.check(lib.sbm_project_load(self._model.device_model.handle, ctypes.byref(desc), ctypes.byref(h)),
                   'sbm_project_load')
        self._device_project = h
        self._richardson_levels = 0      # a freshly loaded project starts without extrapolation
        return h

    # ------------------------------------------------------------------
    # getters (reference :536-597)
    # ------------------------------------------------------------------
    @property
    def n_project_params(self):
        return self._n_project_params

    @property
    def n_project_residuals(self):
        return self._n_residuals

    @property
    def n_total_rows(self):
        return self._n_residuals + len(self._prior_rows()[0]) + len(self._scale_factor_priors)

    @property
    def scale_factors(self):
        try:
            return self._loss_function.scale_factors
        except AttributeError:
            raise AttributeError("%s type doesn't support scale factors" % type(self._loss_function))

    @property
    def experiments(self):
        return iter(self._experiments)

    @property
    def project_param_vector(self):
        return np.copy(self._project_param_vector)

    @property
    def project_param_idx(self):
        return copy.deepcopy(self._project_param_idx)

    @property
    def parameter_priors(self):
        return copy.deepcopy(self._parameter_priors)

    def get_parameter_settings(self):
        """Every parameter group with its settings and their index in the project vector (a copy;
        reference :1064-1068)."""
        import copy
        return copy.deepcopy(self._project_param_idx)

    def print_param_settings(self):
        """Human-readable listing of the same (reference :1038-1062)."""
        for p_group, slots in self._project_param_idx.items():
            print("%s" % p_group)
            for settings, idx in slots.items():
                print("    %-30s theta[%d]" % (settings, idx))

    def get_ordered_project_params(self):
        names = [None] * self._n_project_params
        for p_group, slots in self._project_param_idx.items():
            for sett, gi in slots.items():
                names[gi] = (p_group, sett)
        return names

    def row_index(self, include_priors=False):
        """(experiment name, measure name) per residual row, in row order."""
        labels = [(self._experiments[e].name, nm) for e, nm in zip(self._rows['exp'], self._rows['measure'])]
        if include_priors:
            labels += self._prior_rows()[3] + self._sf_prior_rows()[3]
        return labels

    @property
    def measurements_df(self):
        import pandas as pd
        labels = self.row_index(include_priors=True)
        _, p_mean, p_sigma, _ = self._prior_rows()
        _, s_mean, s_sigma, _ = self._sf_prior_rows()
        mean = np.concatenate([self._rows['data'], p_mean, s_mean])
        std = np.concatenate([self.descriptor_arrays()['row_sigma'], p_sigma, s_sigma])
        tp = np.concatenate([self._rows['t_meas'], np.full(len(p_mean) + len(s_mean), np.nan)])
        return pd.DataFrame({'mean': mean, 'std': std, 'timepoints': tp},
                            index=pd.MultiIndex.from_tuples(labels))

    def get_simulations(self, scaled=False, include_priors=False):
        """Frame with columns 'mean' (simulated value per row) and 'timepoints' (the grid
        time actually sampled), as the reference's ``_simulations_df`` (:568-579)."""
        import pandas as pd
        if 'sims' not in self._last:
            raise ValueError("No simulations executed yet")
        sims = self._last['sims'].copy()
        if scaled and self._last.get('sf') is not None and len(self._last['sf']):
            grp = self._rows['sf']
            sims = sims * np.where(grp >= 0, self._last['sf'][np.clip(grp, 0, None)], 1.0)
        tp = self._rows['t_sim']
        labels = self.row_index(include_priors)
        if include_priors:
            theta = self._project_param_vector
            p_idx, _, _, _ = self._prior_rows()
            s_grp, _, _, _ = self._sf_prior_rows()
            sims = np.concatenate([sims, theta[p_idx], np.log(self._last['sf'][s_grp]) if len(s_grp) else []])
            tp = np.concatenate([tp, np.full(len(p_idx) + len(s_grp), np.nan)])
        return pd.DataFrame({'mean': sims, 'timepoints': tp}, index=pd.MultiIndex.from_tuples(labels))

    def get_model_jacobian_df(self, include_priors=False):
        import pandas as pd
        if 'Jmodel' not in self._last:
            raise ValueError("No jacobian calculations executed yet")
        J = self._last['Jmodel']
        labels = self.row_index(include_priors)
        if include_priors:
            J = np.vstack([J, np.zeros((len(labels) - J.shape[0], J.shape[1]))])
        return pd.DataFrame(J, index=pd.MultiIndex.from_tuples(labels),
                            columns=[str(c) for c in self.get_ordered_project_params()])

    # ------------------------------------------------------------------
    # experiments / priors  (reference :603-702)
    # ------------------------------------------------------------------
    def add_experiment(self, experiment):
        if not isinstance(experiment, list):
            experiment = [experiment]
        names = {e.name for e in self._experiments}
        for e in experiment:
            if e.name in names:
                raise KeyError("An Experiment with name %s is already present" % e.name)
            self._experiments.append(e)
            names.add(e.name)
        self._experiments.sort(key=lambda x: x.name)
        self._update_project_settings()

    def remove_experiments_by_settings(self, settings):
        if not isinstance(settings, dict):
            raise KeyError('settings must be a dict of {setting_name: value}')
        removed = []
        for exp_idx, experiment in enumerate(self._experiments):
            all_equal = True
            for setting in settings:
                if setting not in experiment.settings:
                    raise KeyError('%s is not a setting in experiment %s' % (setting, experiment.name))
                if experiment.settings[setting] != settings[setting]:
                    all_equal = False
            if all_equal:
                removed.append(exp_idx)
        if not removed:
            raise KeyError('None of the settings chosen were in the experiments in the project')
        deleted = [self._experiments.pop(i) for i in reversed(removed)]
        if len(self._experiments) == 0:
            warnings.warn('Project has no more experiments')
            self._drop_device_project()
            self._rows = None
            return deleted
        self._update_project_settings()
        return deleted

    def get_experiment(self, exp_idx):
        return copy.deepcopy(self._experiments[exp_idx])

    def get_experiment_index(self, exp_name):
        for exp_idx, experiment in enumerate(self._experiments):
            if exp_name == experiment.name:
                return exp_idx
        raise KeyError('%s not in experiments' % exp_name)

    def set_scale_factor_log_prior(self, measure_name, log_scale_factor_prior, log_sigma_scale_factor):
        self._loss_function.set_scale_factor_priors(measure_name, log_scale_factor_prior, log_sigma_scale_factor)
        if measure_name not in self._scale_factor_priors:
            self._scale_factor_priors.append(measure_name)
        self._update_project_settings()

    def set_parameter_log_prior(self, p_group, settings, log_scale_parameter_prior, log_sigma_parameter):
        try:
            self._project_param_idx[p_group][settings]
        except KeyError:
            raise KeyError('%s with settings %s not in the project parameters' % (p_group, settings))
        self._parameter_priors.setdefault(p_group, OrderedDict())[settings] = (log_scale_parameter_prior,
                                                                               log_sigma_parameter)
        self._update_project_settings()

    def get_param_index(self, parameter_group, settings='all'):
        if isinstance(settings, str) and settings == 'all':
            return self._project_param_idx[parameter_group]
        return self._project_param_idx[parameter_group][settings]

    def reset_calcs(self):
        self._last = {}
        self._project_param_vector = np.zeros((self.n_project_params,))

    def get_experiment_parameters(self, experiment):
        """Model parameter vector of one experiment at the current project vector
        (host restatement of the gather kernel, reference :343-363)."""
        p = np.zeros(len(self._model.param_order))
        for k, name in enumerate(self._model.param_order):
            if name in experiment.param_global_vector_idx:
                p[k] = np.exp(self._project_param_vector[experiment.param_global_vector_idx[name]])
            else:
                p[k] = experiment.fixed_parameters[name]
        return p

    # ------------------------------------------------------------------
    # device evaluation
    # ------------------------------------------------------------------
    def _options(self, **overrides):
        """Effective integrator options of a call: model defaults < project overrides < call overrides."""
        o = dict(getattr(self._model, 'integrator_options', None) or {})
        o.update(self.integrator_options)
        o.update(overrides)
        return o

    def _opts(self, **overrides):
        o = self._options(**overrides)
        levels = int(o.pop('extrapolate', 0) or 0)
        _lib.implicit_adaptive_defaults(o, set(overrides) | set(self.integrator_options))
        fixed = ('rk4', 'rk4_fixed') + _lib.FIXED_STEP_IMPLICIT
        if str(o.get('method', 'dopri45')).lower() in fixed and not o.get('h0', 0) > 0:
            o['t_end'] = max(g[-1] for g in self._rows['grids'])
            o.setdefault('n_steps', 4096)
        # Richardson levels of the implicit integrator are a property of the loaded project
        if levels != getattr(self, '_richardson_levels', 0):
            _lib.check(_lib.load_library().sbm_project_set_extrapolation(self._device(), levels),
                       'sbm_project_set_extrapolation')
            self._richardson_levels = levels
        return _lib.make_opts(**o)

    def _theta_dev(self, thetas):
        import torch
        if isinstance(thetas, torch.Tensor):
            th = thetas
            if th.dim() == 1:
                th = th.unsqueeze(0)
            if not th.is_cuda:
                th = th.cuda(self._model.device_model.ctx.device)
            th = th.to(torch.float64).contiguous()
            as_torch = True
        else:
            arr = np.atleast_2d(_as_f64(thetas))
            th = torch.from_numpy(arr).cuda(self._model.device_model.ctx.device)
            as_torch = False
        if th.shape[1] != self._n_project_params:
            raise ValueError("project vector has %d entries, project has %d parameters"
                             % (th.shape[1], self._n_project_params))
        return th, as_torch

    def evaluate_batch(self, thetas, jacobian=False, want=('residuals',), **integrator_overrides):
        """Evaluate V project vectors on the device.

        thetas : (V, q) numpy array or torch tensor (log-space).
        want : any of 'residuals', 'sims', 'sf', 'norms', 'status', 'n_steps' and, with
            ``jacobian=True``, 'jacobian', 'model_jacobian', 'gradient', 'sf_gradient'.
        Returns a dict of numpy arrays (torch CUDA tensors if ``thetas`` was one).

        ``method='implicit_controlled'`` integrates with the implicit midpoint rule under global error
        control and ``method='auto'`` with DOPRI45 first and that for the vectors DOPRI45 gives up on
        (stiff ones) -- the control loops of ``_control.py``; the result then also carries 'stiff' (V,)
        bool.  Every other method is one device call.
        """
        o = self._options(**integrator_overrides)
        method = str(o.get('method', 'dopri45')).lower()
        if method not in _control.IMPLICIT_CONTROLLED + _control.AUTO:
            return self._evaluate_once(thetas, jacobian, want, **integrator_overrides)
        import torch
        th, as_torch = self._theta_dev(thetas)
        V = th.shape[0]
        compare = ['sims'] + (['jacobian'] if jacobian else [])
        rtol, atol = float(o.get('rtol', 1e-9)), float(o.get('atol', 1e-12))
        keep = {k: v for k, v in o.items() if k in ('variant',)}

        def split(res):
            st, ns = res.pop('status'), res.pop('n_steps')
            return res, st, ns

        def controlled(idx):
            sub_th = th[torch.as_tensor(idx, device=th.device, dtype=torch.long)]

            def run(sub, mult):
                t = sub_th[torch.as_tensor(sub, device=th.device, dtype=torch.long)]
                return split(self._evaluate_once(t, jacobian, want, method='implicit_midpoint_graded',
                                                 n_steps=int(o.get('n_steps', 0) or 256), step_mult=mult,
                                                 extrapolate=0, rtol=max(1e-2 * rtol, 1e-13),
                                                 atol=max(1e-2 * atol, 1e-14), max_steps=0, **keep))   # (Newton tolerances)
            # Romberg on the ASSEMBLED outputs: residuals, scale factors, Jacobian ... are smooth functions of the
            # discrete solution, so they inherit its expansion in h^2
            return _control.controlled_romberg(run, len(idx), compare, rtol, atol,
                                               max_doublings=int(o.get('max_doublings', 9)))
        if method in _control.IMPLICIT_CONTROLLED:
            out, st, steps, _ = controlled(np.arange(V))
            stiff = np.ones(V, dtype=bool)
        else:
            budget = abs(int(o.get('max_steps') or 0)) or 50000
            def implicit(idx):
                # one device call: the implicit integrator with local error control in the kernel (SBM_IMPLICIT_EXTRAP)
                t = th[torch.as_tensor(idx, device=th.device, dtype=torch.long)]
                oi = _lib.implicit_adaptive_defaults(dict(method=o.get('stiff_method', _lib.STIFF_METHOD), rtol=rtol, atol=atol),
                                                     set(integrator_overrides) | set(self.integrator_options))
                oi['max_steps'] = 0          # the budget belongs to the explicit attempt; the kernel's own limit here
                return split(self._evaluate_once(t, jacobian, want, extrapolate=0, **oi, **keep)) + (None,)
            explicit_method = str(o.get('explicit_method', 'dopri45'))       # 'dop853': the eighth-order pair for the attempt
            if explicit_method == 'auto':       # the pair by predicted pass time (_lib.predict_explicit_pair)
                gmod = getattr(self._model, 'generated', None)
                ch = gmod.rowgroup_chunks() if (gmod is not None and jacobian) else {}
                explicit_method = _lib.predict_explicit_pair(rtol, V * max(1, len(self._experiments)), ch.get('RG0', 1),
                                                             ch.get('RG2', 1))[0]
            ex_tol = _lib.implicit_adaptive_defaults(dict(method=explicit_method, rtol=rtol, atol=atol),
                                                     set(integrator_overrides) | set(self.integrator_options))
            out, st, steps, stiff = _control.with_stiff_fallback(
                lambda: split(self._evaluate_once(th, jacobian, want, max_steps=-budget, extrapolate=0, **ex_tol, **keep)),
                implicit if self._model.n_vars <= _lib.IMPLICIT_MAX_NV else None, V)
        out['status'] = torch.as_tensor(st, dtype=torch.int32, device=th.device)
        out['n_steps'] = torch.as_tensor(np.minimum(steps, 2 ** 31 - 1), dtype=torch.int32, device=th.device)
        out['stiff'] = torch.as_tensor(stiff, device=th.device)
        if as_torch:
            return out
        return {k: v.cpu().numpy() for k, v in out.items()}

    def _evaluate_once(self, thetas, jacobian=False, want=('residuals',), **integrator_overrides):
        """One device call: all vectors with one integrator setting."""
        import torch
        proj = self._device()
        lib = _lib.load_library()
        th, as_torch = self._theta_dev(thetas)
        dev = th.device
        V, q = th.shape
        R = self._n_residuals
        RT = self.n_total_rows
        G = len(self._loss_function.groups) if hasattr(self._loss_function, 'groups') else 0
        f64, i32 = torch.float64, torch.int32
        out = {'residuals': torch.empty((V, RT), dtype=f64, device=dev),
               'sims': torch.empty((V, R), dtype=f64, device=dev),
               'sf': torch.empty((V, G), dtype=f64, device=dev),
               'norms': torch.empty((V,), dtype=f64, device=dev),
               'status': torch.empty((V,), dtype=i32, device=dev),
               'n_steps': torch.empty((V,), dtype=i32, device=dev)}
        opts = self._opts(**integrator_overrides)
        p = _lib.dev_ptr
        if not jacobian:
            _lib.check(lib.sbm_residuals_batch(proj, p(th), V, ctypes.byref(opts), p(out['sims']),
                                               p(out['residuals']), p(out['sf']) if G else None, p(out['norms']),
                                               p(out['status']), p(out['n_steps'])), 'sbm_residuals_batch')
        else:
            out['jacobian'] = torch.empty((V, RT, q), dtype=f64, device=dev)
            if 'model_jacobian' in want:
                out['model_jacobian'] = torch.empty((V, R, q), dtype=f64, device=dev)
            if 'gradient' in want:
                out['gradient'] = torch.empty((V, q), dtype=f64, device=dev)
            if 'sf_gradient' in want and G:
                out['sf_gradient'] = torch.empty((V, G, q), dtype=f64, device=dev)
            _lib.check(lib.sbm_jacobian_batch(proj, p(th), V, ctypes.byref(opts), p(out['sims']),
                                              p(out['residuals']), p(out['jacobian']),
                                              p(out.get('model_jacobian')), p(out['sf']) if G else None,
                                              p(out.get('sf_gradient')), p(out['norms']), p(out.get('gradient')),
                                              p(out['status']), p(out['n_steps'])), 'sbm_jacobian_batch')
        if as_torch:
            return out
        torch.cuda.synchronize(dev)
        return {k: v.cpu().numpy() for k, v in out.items()}

    def _remember(self, theta, res, with_jac):
        self._project_param_vector = np.array(theta, dtype=float).reshape(-1).copy()
        self._last['sims'] = res['sims'][0]
        self._last['sf'] = res['sf'][0]
        if with_jac:
            self._last['Jmodel'] = res['model_jacobian'][0]
        if hasattr(self._loss_function, '_store') and len(res['sf'][0]):
            self._loss_function._store(res['sf'][0], res['sf_gradient'][0] if with_jac else None)
        st = int(res['status'][0])
        if st != 0:
            warnings.warn("integration failed (%s): residuals are inf, as the reference returns for "
                          "NaN simulations" % _lib.STATUS_NAMES.get(st, '?'))

    # ------------------------------------------------------------------
    # public objective functions  (reference :708-892)
    # ------------------------------------------------------------------
    def _single(self):
        """Integrator overrides of the single-vector methods below -- what a serial optimiser calls (the reference's
        leastsq(project.residuals, x0, Dfun=project.calc_project_jacobian)): one parameter vector leaves the chip
        empty, so the sensitivity kernel takes its small-batch split unless the project's options name a variant.
        Results equal the corresponding row of a batch call to the integration tolerance, not bit for bit."""
        o = self._options()
        kw = {} if 'variant' in o else {'variant': 'small_batch'}
        # ... and a stiff trial point must not come back as inf rows where the reference's LSODA would integrate it:
        # method='auto' (OdeModel._single_method) unless the options name a method of their own
        if 'method' not in self.integrator_options and hasattr(self._model, '_single_method'):
            kw.update(self._model._single_method())
        return kw

    def residuals(self, project_param_vector):
        """(B*sim - data)/sigma for every measurement row, then prior rows; (m,) array."""
        self.reset_calcs()
        res = self.evaluate_batch(np.asarray(project_param_vector, dtype=float)[None, :], **self._single())
        self._remember(project_param_vector, res, False)
        return res['residuals'][0]

    def residuals_batch(self, thetas, **integrator_overrides):
        return self.evaluate_batch(thetas, **integrator_overrides)['residuals']

    def calc_project_jacobian(self, project_param_vector):
        """d(B*sim)/d theta, (m, n) array: B*J + sim (x) dB/dtheta (reference :731-771)."""
        self.reset_calcs()
        res = self.evaluate_batch(np.asarray(project_param_vector, dtype=float)[None, :], jacobian=True,
                                  want=('jacobian', 'model_jacobian', 'sf_gradient'), **self._single())
        self._remember(project_param_vector, res, True)
        return res['jacobian'][0]

    def calc_project_jacobian_batch(self, thetas, **integrator_overrides):
        return self.evaluate_batch(thetas, jacobian=True, want=('jacobian',), **integrator_overrides)['jacobian']

    def calc_rss_gradient(self, project_param_vector, *args):
        """(J^T r): gradient of 0.5*sum(r^2) (reference :773-805).  One augmented integration
        gives both r and J here; the reference integrates three times."""
        self.reset_calcs()
        res = self.evaluate_batch(np.asarray(project_param_vector, dtype=float)[None, :], jacobian=True,
                                  want=('jacobian', 'model_jacobian', 'gradient', 'sf_gradient'), **self._single())
        self._remember(project_param_vector, res, True)
        return res['gradient'][0]

    def calc_sum_square_residuals(self, project_param_vector):
        """0.5 * sum(r^2) (reference :807-827)."""
        r = self.residuals(project_param_vector)
        return 0.5 * np.sum(r ** 2)

    def calc_sum_square_residuals_batch(self, thetas, **integrator_overrides):
        return 0.5 * self.evaluate_batch(thetas, **integrator_overrides)['norms']

    def calc_scale_factors_entropy(self, temperature=1.0, sims=None):
        """Sum over the scale-factor groups of T * log-integral entropy terms (reference :854-872,
        abstract_loss_function.py:122-134, linear_scale_factor.py:63-81) for the simulations of the last
        evaluation (or ``sims``, one unscaled value per measurement row).  Every group needs a log prior,
        as in the reference.  Host-side quadrature on device-computed simulations."""
        from .loss_functions.squared_loss.linear_scale_factor import scale_factor_entropy
        if sims is None:
            if 'sims' not in self._last:
                raise ValueError("no simulation yet: call residuals() first")
            sims = self._last['sims']
        a = self.descriptor_arrays()
        grp, d, sg = a['row_sf'], a['row_data'], a['row_sigma']
        entropy = 0.0
        for gi, sf in enumerate(self._loss_function._scale_factors.values()):
            if sf.log_prior is None:
                raise ValueError("scale factor entropy needs a log prior on every scale factor "
                                 "(set_scale_factor_log_prior)")
            sel = grp == gi
            w = 1.0 / sg[sel] ** 2
            entropy += temperature * scale_factor_entropy(np.sum(sims[sel] ** 2 * w), np.sum(sims[sel] * d[sel] * w),
                                                          sf.log_prior, sf.log_prior_sigma, temperature)
        return entropy

    def free_energy(self, project_param_vector, temperature=1):
        """rss - scale-factor entropy (reference :874-892)."""
        rss = self.calc_sum_square_residuals(project_param_vector)
        return rss - self.calc_scale_factors_entropy(temperature)

    def free_energy_batch(self, thetas, temperature=1, **integrator_overrides):
        """free_energy for V parameter vectors: one batched device evaluation, then the quadratures."""
        res = self.evaluate_batch(thetas, **integrator_overrides)
        out = 0.5 * res['norms']
        for v in range(len(out)):
            out[v] = out[v] - self.calc_scale_factors_entropy(temperature, sims=res['sims'][v]) \
                if np.isfinite(out[v]) else np.inf
        return out

    def fit_batch(self, thetas0, **options):
        """Multi-start Levenberg-Marquardt from every row of ``thetas0`` at once (project/fitting.py);
        the batched counterpart of ``leastsq(self.residuals, x0, Dfun=self.calc_project_jacobian)``."""
        from .fitting import levenberg_marquardt_batch
        return levenberg_marquardt_batch(self, thetas0, **options)

    def nlopt_fcn(self, project_param_vector, grad):
        """nlopt-style objective: fills ``grad`` in place when it is non-empty (reference :829-852)."""
        if grad.size > 0:
            grad[:] = self.calc_rss_gradient(project_param_vector)
        return self.calc_sum_square_residuals(project_param_vector)

    # ------------------------------------------------------------------
    # dict <-> vector helpers (reference :898-951)
    # ------------------------------------------------------------------
    def project_param_dict_to_vect(self, param_dict, default_value=0.0):
        param_vector = np.ones((self.n_project_params,)) * default_value
        for p_group in self._project_param_idx:
            for exp_settings, global_idx in self._project_param_idx[p_group].items():
                try:
                    param_vector[global_idx] = param_dict[p_group][exp_settings]
                except KeyError:
                    pass
        return param_vector

    def project_param_vect_to_dict(self, param_vector):
        return {g: {s: param_vector[i] for s, i in slots.items()} for g, slots in self._project_param_idx.items()}

    def group_experiments(self, settings_groups):
        grouped = defaultdict(list)
        for experiment in self._experiments:
            key = []
            for setting in settings_groups:
                if setting not in experiment.settings:
                    raise KeyError('%s is not a setting in experiment %s' % (setting, experiment.name))
                key.append(experiment.settings[setting])
            grouped[tuple(key)].append(experiment)
        return grouped
