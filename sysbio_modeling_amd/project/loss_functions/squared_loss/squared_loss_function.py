"""Weighted squared loss with optional linear scale factors.

Configuration object for the loss the assembly kernel evaluates
(csrc/sbm_core.hip::k_assemble):  r = (B s - d) / sigma,  J = B dS + s (x) dB
(reference project/loss_functions/squared_loss/squared_loss_function.py:27-80,
abstract_loss_function.py:22-120).  It holds the scale-factor groups, their
priors and the last values computed on the device.

Inside a Project the loss is evaluated as part of ``sbm_residuals_batch`` /
``sbm_jacobian_batch``.  The reference's loss objects can also be called on
hand-built frames (its tests/test_Loss_Functions.py does); those methods
(``residuals``, ``jacobian``, ``evaluate``, ``update_scale_factors``,
``update_scale_factors_gradient``, ``scale_sim_values``) are kept here and run
the same device kernel through ``sbm_loss_eval_host`` (include/sbm.h).
"""
import ctypes
from copy import deepcopy

import numpy as np

from ...utils import OrderedHashDict
from .linear_scale_factor import LinearScaleFactor

PRIOR_LEVEL = "~Prior"          # reference base_project.py:250-262
SF_PRIOR_LEVEL = "~~SF_Prior"   # reference base_project.py:264-280


def _dptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class SquareLossFunction(object):
    loss_type = 0                    # SBM_LOSS_SQUARE (include/sbm.h)
    normalize_sigma_by_mean = False  # NormalizedSquareLossFunction sets this

    def __init__(self, sf_groups=None, sf_type=LinearScaleFactor):
        self._scale_factors = OrderedHashDict()
        if sf_groups is not None:
            if isinstance(sf_groups, (str, frozenset)):
                sf_groups = [sf_groups]
            for measure_group in sf_groups:
                self._scale_factors[measure_group] = sf_type()

    @property
    def scale_factors(self):  # noqa: F811 -- instance view shadows the class marker
        return deepcopy(self._scale_factors)

    @property
    def groups(self):
        """Scale-factor groups as lists of measure names, in definition order."""
        return [[k] if isinstance(k, str) else sorted(k) for k in self._scale_factors.keys()]

    def group_index(self, measure_name):
        for gi, g in enumerate(self._scale_factors.keys()):
            if g == measure_name or (not isinstance(g, str) and measure_name in g):
                return gi
        return -1

    def set_scale_factor_priors(self, measure_name, log_scale_factor_prior, log_sigma_scale_factor):
        try:
            sf = self._scale_factors[measure_name]
        except KeyError:
            raise KeyError("%s not present as a scale factor" % (measure_name,))
        sf.log_prior = log_scale_factor_prior
        sf.log_prior_sigma = log_sigma_scale_factor

    def _store(self, sf_values, sf_gradients=None):
        for gi, sf in enumerate(self._scale_factors.values()):
            sf._sf = float(sf_values[gi])
            if sf_gradients is not None:
                sf._sf_gradient = sf_gradients[gi].copy()

    # ------------------------------------------------------------------
    # frame-level API of the reference (device-evaluated)
    # ------------------------------------------------------------------
    @staticmethod
    def _labels(frame):
        """[(experiment, measure)] of a frame indexed like the reference's (MultiIndex level 0 / 1)."""
        return [tuple(ix[:2]) if isinstance(ix, tuple) else (None, ix) for ix in frame.index]

    def _frame_eval(self, simulations, experiment_measures, simulations_jacobian=None, want_jacobian=False,
                    reference_compat=True):
        """One launch of the assembly kernel on the frames' values.  Returns a dict with residuals (frame row
        order), jacobian (same), sf, sf_grad, status."""
        from .... import _lib
        labels = self._labels(simulations)
        if labels != self._labels(experiment_measures):
            raise ValueError("simulations and experiment_measures are not indexed alike")
        sim = np.asarray(simulations['mean'].values, dtype=np.float64)
        data = np.asarray(experiment_measures['mean'].values, dtype=np.float64)
        sigma = np.asarray(experiment_measures['std'].values, dtype=np.float64).copy()
        jm = None
        if simulations_jacobian is not None:
            jm = np.asarray(getattr(simulations_jacobian, 'values', simulations_jacobian), dtype=np.float64)
            if jm.shape[0] != len(labels):
                raise ValueError("simulations_jacobian has %d rows, simulations %d" % (jm.shape[0], len(labels)))
        groups = list(self._scale_factors.keys())
        sfs = list(self._scale_factors.values())
        body, plain, sfp = [], [], []   # frame row numbers
        row_sf, sfp_group = [], []
        for i, (exp_name, measure) in enumerate(labels):
            if exp_name == SF_PRIOR_LEVEL:
                name = measure[1:] if isinstance(measure, str) and measure.startswith('~') else measure
                g = self.group_index(name)
                if g >= 0 and sfs[g].log_prior is not None:
                    sfp.append(i)
                    sfp_group.append(g)
                    continue
                body.append(i); plain.append(1); row_sf.append(-1)   # no prior set: an ordinary row
            elif exp_name == PRIOR_LEVEL:
                body.append(i); plain.append(1); row_sf.append(-1)
            else:
                body.append(i); plain.append(0); row_sf.append(self.group_index(measure))
        for g, sf in enumerate(sfs):   # abstract_loss_function.py:94-107
            if sf.log_prior is not None and g not in sfp_group:
                key = groups[g]
                raise KeyError("No prior in simulations for %s" % (key if isinstance(key, str) else sorted(key)[0]))
        body = np.asarray(body, dtype=np.int64)
        sfp = np.asarray(sfp, dtype=np.int64)
        plain = np.asarray(plain, dtype=np.int32)
        if self.normalize_sigma_by_mean:   # normalized_squared_loss_function.py:40-46
            rel = plain == 0
            sigma[body[rel]] *= data[body[rel]]
        R, G, q = len(body), len(groups), (0 if jm is None else jm.shape[1])
        if R == 0:
            raise ValueError("no measurement rows in the frames")
        d_data = np.ascontiguousarray(data[body]); d_sigma = np.ascontiguousarray(sigma[body])
        d_sf = np.asarray(row_sf, dtype=np.int32); d_plain = np.ascontiguousarray(plain)
        d_spg = np.asarray(sfp_group, dtype=np.int32)
        # the prior a scale factor carries wins over the frame's (the reference writes log B into the frame's
        # simulation row and reads mean/std from the measures row, which Project fills from the same prior)
        d_spm = np.ascontiguousarray(data[sfp]); d_sps = np.ascontiguousarray(sigma[sfp])
        desc = _lib.LossDesc(R, q, G, len(sfp), int(self.loss_type), 1 if reference_compat else 0,
                             d_data.ctypes.data_as(_lib.c_double_p), d_sigma.ctypes.data_as(_lib.c_double_p),
                             d_sf.ctypes.data_as(_lib.c_int32_p), d_plain.ctypes.data_as(_lib.c_int32_p),
                             d_spg.ctypes.data_as(_lib.c_int32_p), d_spm.ctypes.data_as(_lib.c_double_p),
                             d_sps.ctypes.data_as(_lib.c_double_p))
        h_sim = np.ascontiguousarray(sim[body])
        h_jm = None if jm is None else np.ascontiguousarray(jm[body])
        RT = R + len(sfp)
        out_R = np.empty(RT); out_sf = np.empty(max(G, 1)); status = np.zeros(1, dtype=np.int32)
        out_J = np.empty((RT, q)) if (want_jacobian and jm is not None) else None
        out_sfg = np.empty((max(G, 1), q)) if jm is not None else None
        ctx = _lib.default_context()
        _lib.check(ctx.lib.sbm_loss_eval_host(ctx.handle, ctypes.byref(desc), 1, _dptr(h_sim), _dptr(h_jm), _dptr(out_R),
                                              _dptr(out_J), _dptr(out_sf), _dptr(out_sfg), None, _dptr(status)),
                   'sbm_loss_eval_host')
        order = np.concatenate([body, sfp])
        res = np.empty(len(labels)); res[order] = out_R
        jac = None
        if out_J is not None:
            jac = np.empty((len(labels), q)); jac[order] = out_J
        ok = status[0] == 0
        if ok and G:
            self._store(out_sf[:G], None if out_sfg is None else out_sfg[:G])
        return {'residuals': res, 'jacobian': jac, 'sf': out_sf[:G], 'sf_grad': None if out_sfg is None else out_sfg[:G],
                'ok': ok}

    def residuals(self, simulations, experiment_measures):
        """(B s - d)/sigma per frame row; all inf when the simulations hold NaN (reference :27-43)."""
        out = self._frame_eval(simulations, experiment_measures)['residuals']
        try:
            import pandas as pd
            return pd.Series(out, index=simulations.index)
        except ImportError:  # pragma: no cover
            return out

    def evaluate(self, simulations, experiment_measures):
        """0.5 sum r^2 (reference :23-25)."""
        return 0.5 * float(np.sum(np.asarray(self.residuals(simulations, experiment_measures)) ** 2))

    def jacobian(self, simulations, experiment_measures, simulations_jacobian):
        """B dS + s (x) dB for rows with a scale factor, dS unchanged otherwise, not divided by sigma
        (reference :45-80); all inf on NaN input."""
        if len(self._scale_factors) == 0 and self.loss_type == 0:
            jv = np.asarray(getattr(simulations_jacobian, 'values', simulations_jacobian))
            if not (np.isnan(np.asarray(simulations['mean'].values)).any() or np.isnan(jv).any()):
                return simulations_jacobian   # the reference hands the frame back untouched
        jv = np.asarray(getattr(simulations_jacobian, 'values', simulations_jacobian), dtype=np.float64)
        out = self._frame_eval(simulations, experiment_measures, simulations_jacobian, want_jacobian=True)
        jac = out['jacobian']
        if np.isnan(jv).any():
            jac = np.full_like(jv, np.inf)
        if hasattr(simulations_jacobian, 'copy') and hasattr(simulations_jacobian, 'index'):
            frame = simulations_jacobian.copy()
            frame.iloc[:, :] = jac
            return frame
        return jac

    def update_scale_factors(self, simulations, experiment_measures):
        """Stores B_g in ``scale_factors`` (reference abstract_loss_function.py:88-92)."""
        if len(self._scale_factors):
            self._frame_eval(simulations, experiment_measures)

    def update_scale_factors_gradient(self, simulations, experiment_measures, simulations_jacobian):
        """Stores dB_g/dtheta in ``scale_factors[...].gradient`` (reference :82-108)."""
        if len(self._scale_factors):
            self._frame_eval(simulations, experiment_measures, simulations_jacobian)

    def scale_sim_values(self, simulations):
        """Copy of the frame with every scale-factor group's rows multiplied by its current B
        (reference abstract_loss_function.py:47-60)."""
        scaled = simulations.copy()
        labels = self._labels(simulations)
        sfs = list(self._scale_factors.values())
        mult = np.ones(len(labels))
        for i, (exp_name, measure) in enumerate(labels):
            if exp_name in (PRIOR_LEVEL, SF_PRIOR_LEVEL):
                continue
            g = self.group_index(measure)
            if g >= 0:
                mult[i] = sfs[g].sf
        scaled['mean'] = np.asarray(simulations['mean'].values, dtype=np.float64) * mult
        return scaled
