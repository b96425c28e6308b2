"""ORACLE (test infrastructure, not product code) -- assembly half of the hot path.

CPU restatement, in plain numpy loops, of what the reference's ``Project`` and
``SquareLossFunction`` compute for one log-space parameter vector:

  parameter indexing   project/base_project.py:164-276  (_set_local_param_idx)
  row table            project/base_project.py:296-341  (_measurements_as_dataframe)
  theta -> p           project/base_project.py:343-363  (get_experiment_parameters)
  simulate + sample    project/base_project.py:365-425  + project/utils.py:10-68
  sensitivities        project/base_project.py:443-517  + project/utils.py:29-89
  priors               project/base_project.py:427-437
  residuals / SF       loss_functions/squared_loss/squared_loss_function.py:27-42,
                       abstract_loss_function.py:46-120, linear_scale_factor.py:27-61
  Jacobian             squared_loss_function.py:44-106
  public API           project/base_project.py:708-852

The reference module itself cannot be imported here (Python-2 syntax at
base_project.py:195,1058; numba import at linear_scale_factor.py:5; removed
pandas APIs ``sortlevel`` :340 and ``.ix`` squared_loss_function.py:78), so this
restatement is PINNED by the reference's own known answers instead
(tests/test_oracle_golden.py): tests/test_Project.py:94 (35 residuals, 3
parameters), :114 (scale factor 3.75), :19-23 (analytic Jacobian with the
log-parameter chain rule), :165-176 (finite-difference identities),
:322-349 (Michaelis-Menten 'sum' mapping) and tests/test_Loss_Functions.py:
34-80,94-164,196-233 (residual / scale-factor / SF-gradient / SF-prior identities).

Where the reference is buggy outside anything its tests exercise, this oracle
implements the evident intent and says so:
  quirk 5  _map_model_jac_to_measures does not advance res_idx between the
           measurements of one experiment (:467,482,488)          -> advanced here
  quirk 6  Jacobian columns mis-indexed when an experiment fixes parameters
           (:469,482-483)                                         -> indexed by sens column
  quirk 7  update_scale_factors_gradient hstacks the per-measure Jacobians of a
           multi-measure group (squared_loss_function.py:102-104) -> rows stacked
``reference_compat=True`` (default) keeps the reference's tested behaviours:
Jacobian not divided by sigma (quirk 3), parameter-prior rows of J zero (quirk 4).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from . import odeint_oracle

N_GRID = 1000  # base_project.py:419,510


class ProjectOracle(object):
    def __init__(self, gm, experiments, model_parameter_settings, measurement_to_model_map,
                 sf_groups=None, reference_compat=True, use_c=True, loss='square'):
        # 'square' | 'log' (log_squared_loss_function.py, log_scale_factor.py) |
        # 'normalized' (normalized_squared_loss_function.py: sigma *= mean, then 'square')
        self.loss = loss
        self.gm = gm
        self.param_order = list(gm.param_order)
        self.n_vars = gm.n_vars
        self.sens_params = list(gm.sens_params)
        self.use_c = use_c
        self.compat = reference_compat
        self.experiments = sorted(experiments, key=lambda x: x.name)      # base_project.py:624
        self.settings = model_parameter_settings
        self.mmap = dict(measurement_to_model_map)
        # scale-factor groups: str or frozenset keys, in the given order (abstract_loss_function.py:31-33)
        self.sf_groups = []
        for g in (sf_groups or []):
            self.sf_groups.append([g] if isinstance(g, str) else sorted(g))
        self.parameter_priors = OrderedDict()
        self.sf_priors = OrderedDict()   # group index -> (log prior, sigma)
        self._index_parameters()
        self.scale_factors = [1.0] * len(self.sf_groups)

    # -- base_project.py:164-276 -------------------------------------------
    def _index_parameters(self):
        s = self.settings
        all_params = set(self.param_order)
        local_pars = list(s.get('Local', []))
        project_fixed = list(s.get('Fixed', []))
        global_pars = list(s.get('Global', []))
        shared_groups = s.get('Shared', {})
        shared_pars = set(p for g in shared_groups for p in shared_groups[g])
        no_settings = all_params - set(local_pars) - set(project_fixed) - shared_pars - set(global_pars)
        # unlisted parameters become global (:193-198); set order is arbitrary in the reference,
        # made deterministic here by model order
        global_pars.extend([p for p in self.param_order if p in no_settings])
        idx = OrderedDict()
        n = 0
        for p in global_pars:
            idx[p] = {'Global': n}
            n += 1
        self.exp_param_idx = []
        for exp in self.experiments:
            epi = OrderedDict()
            exp_fixed = list(exp.fixed_parameters.keys()) if exp.fixed_parameters else []
            all_fixed = project_fixed + exp_fixed
            for p in project_fixed:
                if p not in exp_fixed:
                    raise ValueError('%s was declared as a fixed parameter, but in experiment %s no value provided'
                                     % (p, exp.name))
            for p in global_pars:
                if p not in all_fixed:
                    epi[p] = idx[p]['Global']
            for g in shared_groups:
                for p, sett in shared_groups[g].items():
                    if p in exp_fixed:
                        continue
                    key = 'None' if sett is None else tuple(exp.settings[x] for x in tuple(sett))
                    idx.setdefault(g, {})
                    if key not in idx[g]:
                        idx[g][key] = n
                        n += 1
                    epi[p] = idx[g][key]
            for p in local_pars:
                if p in exp_fixed:
                    continue
                idx['%s_%s' % (p, exp.name)] = {'Local': n}
                epi[p] = n
                n += 1
            self.exp_param_idx.append(epi)
        self.project_param_idx = idx
        self.n_project_params = n

    def set_parameter_log_prior(self, p_group, settings, mean, sigma):
        self.parameter_priors.setdefault(p_group, OrderedDict())[settings] = (mean, sigma)

    def set_scale_factor_log_prior(self, measure_name, mean, sigma):
        for gi, g in enumerate(self.sf_groups):
            if measure_name in g or (not isinstance(measure_name, str) and sorted(measure_name) == g):
                self.sf_priors[gi] = (mean, sigma)
                return
        raise KeyError("%s not present as a scale factor" % measure_name)

    # -- rows: base_project.py:296-341 ---------------------------------------
    def rows(self):
        out = []
        for ei, exp in enumerate(self.experiments):
            for m in exp.measurements:
                vals, std, tps = m.get_nonzero_measurements()
                for v, s_, t in zip(vals, std, tps):
                    if getattr(self, 'loss', 'square') == 'normalized':
                        s_ = s_ * v                      # normalized_squared_loss_function.py:40-46
                    out.append((ei, m.variable_name, float(v), float(s_), float(t)))
        return out

    def _group_of(self, measure):
        for gi, g in enumerate(self.sf_groups):
            if measure in g:
                return gi
        return -1

    # -- base_project.py:343-363 ---------------------------------------------
    def experiment_parameters(self, ei, theta):
        exp = self.experiments[ei]
        p = np.zeros(len(self.param_order))
        for k, name in enumerate(self.param_order):
            if name in self.exp_param_idx[ei]:
                p[k] = np.exp(theta[self.exp_param_idx[ei][name]])
            else:
                p[k] = exp.fixed_parameters[name]
        return p

    def _t_sim(self, exp):
        t_end = exp.get_unique_timepoints()[-1]                      # :418
        return np.linspace(0, t_end, N_GRID)                         # :419

    # -- simulate + sample ----------------------------------------------------
    def simulate_rows(self, theta, with_jacobian=False):
        """sims [R], sim_times [R], model Jacobian [R, q] (or None)."""
        q = self.n_project_params
        sims, times, jac = [], [], []
        dtheta = np.exp(theta)                                       # :450
        n, k = self.n_vars, len(self.sens_params)
        for ei, exp in enumerate(self.experiments):
            p = self.experiment_parameters(ei, theta)
            t_sim = self._t_sim(exp)
            if getattr(self, 'tight', False):
                # checker mode: DOP853 at rtol 1e-13 instead of the reference's LSODA call, at the sampled grid points
                # only ("GPU more accurate than LSODA" must be distinguishable from "GPU wrong")
                need = sorted(set(int(i) for m in exp.measurements
                                  for i in np.searchsorted(t_sim, m.get_nonzero_measurements()[2])))
                sol = odeint_oracle.tight_solution(self.gm, p, np.concatenate([[0.0], t_sim[need]]),
                                                   sens=with_jacobian, use_c=self.use_c, atol=1e-30)[1:]
                Y = np.full((len(t_sim), n), np.nan)
                Y[need] = sol[:, :n]
                if with_jacobian:
                    S = np.full((len(t_sim), n * k), np.nan)
                    S[need] = sol[:, n:]
            # (full_output only to count LSODA's steps for the CPU baseline of bench.py: same call, same numbers)
            elif with_jacobian:
                (S, Y), info = odeint_oracle.calc_jacobian(self.gm, p, t_sim, use_c=self.use_c, return_states=True,
                                                           full_output=True)
            else:
                Y, info = odeint_oracle.simulate(self.gm, p, t_sim, use_c=self.use_c, full_output=True)
            if not getattr(self, 'tight', False):
                self.lsoda_steps = getattr(self, 'lsoda_steps', 0) + int(info['nst'][-1])
            for m in exp.measurements:
                mtype, margs = self.mmap[m.variable_name]
                _, _, tps = m.get_nonzero_measurements()
                t_idx = np.searchsorted(t_sim, tps)                  # project/utils.py:19,37,55,79
                if mtype == 'custom':
                    # base_project.py:125-128,380-383,461-464: (parameters, map function, Jacobian map function), called
                    # with the whole simulation.  (The reference hands the Jacobian callback only the sensitivities; a
                    # pointwise NONLINEAR observable also needs the states, passed here as a keyword.)
                    cpar, map_fn, jac_fn = margs
                    sim, _ = map_fn(Y, t_sim, exp, m, cpar, True)
                    sims.extend(sim)
                    times.extend(t_sim[t_idx])
                    var_list = None
                    if with_jacobian:
                        mj = jac_fn(S, t_sim, exp, m, cpar, True, model_sim=Y)
                else:
                    var_list = [margs] if mtype == 'direct' else list(margs)
                    sim = np.zeros(len(t_idx))
                    for v in var_list:                               # utils.py:20 / :61-66
                        sim += Y[t_idx, v]
                    sims.extend(sim)
                    times.extend(t_sim[t_idx])
                if with_jacobian:
                    if var_list is not None:
                        mj = np.zeros((len(t_idx), k))
                    for v in (var_list or []):                       # utils.py:38 / :86-88
                        mj += S[t_idx, v * k:(v + 1) * k]
                    rows = np.zeros((len(t_idx), q))
                    for name in self.param_order:                    # base_project.py:469-485
                        if name not in self.exp_param_idx[ei]:
                            continue                                 # fixed in this experiment (:476-480)
                        if name not in self.sens_params:
                            continue
                        pj = self.exp_param_idx[ei][name]
                        rows[:, pj] += mj[:, self.sens_params.index(name)] * dtheta[pj]
                    jac.append(rows)
        sims = np.asarray(sims)
        J = np.vstack(jac) if with_jacobian and jac else (np.zeros((0, q)) if with_jacobian else None)
        return sims, np.asarray(times), J

    def _prior_rows(self, theta):
        """(sim value, prior mean, prior sigma, project index) per parameter prior (:427-437)."""
        out = []
        for g in self.parameter_priors:
            for sett, (mean, sigma) in self.parameter_priors[g].items():
                pi = self.project_param_idx[g][sett]
                out.append((theta[pi], mean, sigma, pi))
        return out

    # -- scale factors: linear_scale_factor.py:27-42 ---------------------------
    def _sf(self, rows, sims, J=None):
        G = len(self.sf_groups)
        B = np.ones(G)
        dB = np.zeros((G, self.n_project_params))
        grp = np.array([self._group_of(r[1]) for r in rows], dtype=int)
        d = np.array([r[2] for r in rows])
        sg = np.array([r[3] for r in rows])
        for g in range(G):
            sel = grp == g
            s, dd, ss = sims[sel], d[sel], sg[sel]
            if getattr(self, 'loss', 'square') == 'log':
                es = ss / dd                                             # log_scale_factor.py:18
                inv_std_sum = np.sum(1.0 / es ** 2)                      # :20
                log_data = np.exp(np.sum(np.log(dd) / es ** 2) / inv_std_sum)    # :21
                log_sim = np.exp(-np.sum(np.log(s) / es ** 2) / inv_std_sum)     # :24
                B[g] = log_data * log_sim                                # :26
                if J is not None:
                    Jg = J[sel]
                    exp_grad = (-1.0 / inv_std_sum) * np.sum(Jg.T / (s * es ** 2), axis=1)   # :33-35
                    dB[g] = B[g] * exp_grad                              # :36
                continue
            sim_dot_exp = np.sum((s * dd) / ss ** 2)
            sim_dot_sim = np.sum((s * s) / ss ** 2)
            B[g] = sim_dot_exp / sim_dot_sim
            if J is not None:
                Jg = J[sel]
                jac_dot_exp = np.sum((Jg.T * dd) / ss ** 2, axis=1)
                jac_dot_sim = np.sum(Jg.T * s / ss ** 2, axis=1)
                dB[g] = jac_dot_exp / sim_dot_sim - 2 * sim_dot_exp * jac_dot_sim / sim_dot_sim ** 2
        return B, dB, grp, d, sg

    # -- public API: base_project.py:708-852 ------------------------------------
    def residuals(self, theta, return_parts=False):
        theta = np.asarray(theta, dtype=float)
        rows = self.rows()
        sims, times, _ = self.simulate_rows(theta)
        pri = self._prior_rows(theta)
        n_tot = len(rows) + len(pri) + len(self.sf_priors)
        if np.any(np.isnan(sims)):                                   # squared_loss_function.py:28-32
            res = np.full(n_tot, np.inf)
            return (res, sims, None) if return_parts else res
        B, _, grp, d, sg = self._sf(rows, sims)
        self.scale_factors = list(B)
        scaled = sims * np.where(grp >= 0, B[np.clip(grp, 0, None)], 1.0) if len(B) else sims
        if getattr(self, 'loss', 'square') == 'log':
            if np.min(d) <= 0:
                raise ValueError("LogSquare loss cannot handle measurements smaller or equal to zero")
            res = list((np.log(scaled) - np.log(d)) / sg)             # log_squared_loss_function.py:56-64
        else:
            res = list((scaled - d) / sg)                            # :40
        for val, mean, sigma, _ in pri:
            res.append((val - mean) / sigma)
        for gi, (mean, sigma) in self.sf_priors.items():             # linear_scale_factor.py:55-61
            res.append((np.log(B[gi]) - mean) / sigma)
        res = np.asarray(res)
        return (res, sims, B) if return_parts else res

    def model_jacobian(self, theta):
        theta = np.asarray(theta, dtype=float)
        _, _, J = self.simulate_rows(theta, with_jacobian=True)
        return J

    def calc_project_jacobian(self, theta):
        theta = np.asarray(theta, dtype=float)
        rows = self.rows()
        sims, _, J = self.simulate_rows(theta, with_jacobian=True)
        pri = self._prior_rows(theta)
        q = self.n_project_params
        n_tot = len(rows) + len(pri) + len(self.sf_priors)
        if np.any(np.isnan(sims)) or np.any(np.isnan(J)):             # squared_loss_function.py:46-50
            return np.full((n_tot, q), np.inf)
        B, dB, grp, d, sg = self._sf(rows, sims, J)
        out = J.copy()
        if getattr(self, 'loss', 'square') == 'log':
            out = J / sims[:, None]                                  # log_squared_loss_function.py:69,92
            for g in range(len(self.sf_groups)):
                sel = grp == g
                out[sel] = out[sel] + (dB[g] / B[g])[None, :]         # :92-93
        else:
            for g in range(len(self.sf_groups)):                     # :59-78: B*J + sim (x) dB/dtheta
                sel = grp == g
                out[sel] = J[sel] * B[g] + sims[sel][:, None] * dB[g][None, :]
        if not self.compat:
            out = out / sg[:, None]
        extra = []
        for val, mean, sigma, pi in pri:                             # base_project.py:519-530 never runs
            row = np.zeros(q)
            if not self.compat:
                row[pi] = 1.0 / sigma
            extra.append(row)
        for gi, (mean, sigma) in self.sf_priors.items():             # linear_scale_factor.py:44-53
            row = dB[gi] / B[gi]
            if not self.compat:
                row = row / sigma
            extra.append(row)
        if extra:
            out = np.vstack([out, np.asarray(extra)])
        return out

    def calc_sum_square_residuals(self, theta):
        r = self.residuals(theta)
        return 0.5 * np.sum(r ** 2)                                  # base_project.py:827

    def calc_rss_gradient(self, theta):
        r = self.residuals(theta)
        J = self.calc_project_jacobian(theta)
        return (J.T * r).sum(axis=1)                                 # base_project.py:803-805
