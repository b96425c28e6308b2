"""tests/golden/containers_ref.npz: the REAL reference ``Experiment`` / ``TimecourseMeasurement`` on random measurement sets.

experiment/experiments.py and measurement/*.py are plain numpy and run here as they stand (loaded from their files;
measurement/ as a package for its relative import).  Recorded per case: the inputs (names, values, times, error bars --
NaN where none were given) and what the reference's objects answer -- the order the experiment keeps its measurements in,
the default error bars, get_unique_timepoints with and without zero, get_nonzero_measurements, and the arrays left after
drop_timepoint_zero (of one variable, then of all).

Run in the build container only (reads /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_containers.py"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
sys.dont_write_bytecode = True


def reference_classes():
    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    pkg = types.ModuleType('refmeasurement')
    pkg.__path__ = [os.path.join(REF, 'measurement')]
    sys.modules['refmeasurement'] = pkg
    load('refmeasurement.abstract_measurement', os.path.join(REF, 'measurement', 'abstract_measurement.py'))
    tm = load('refmeasurement.timecourse_measurement', os.path.join(REF, 'measurement', 'timecourse_measurement.py'))
    ex = load('refexperiments', os.path.join(REF, 'experiment', 'experiments.py'))
    return ex.Experiment, tm.TimecourseMeasurement


def cases():
    """(variable names, times, values, std or None) per measurement; deterministic"""
    rng = np.random.default_rng(20261006)
    out = []
    for c in range(5):
        names = ['zeta', 'Alpha', 'm%d' % c, 'beta'][:2 + c % 3]
        ms = []
        for j, nm in enumerate(names):
            n = int(rng.integers(3, 9))
            t = np.sort(rng.choice(np.arange(0.0, 12.0, 0.5), size=n, replace=False))
            if (c + j) % 2 == 0:
                t[0] = 0.0
            v = rng.uniform(0.0, 5.0, n)
            s = None if (c + j) % 3 == 0 else rng.uniform(0.1, 1.0, n)
            ms.append((nm, t, v, s))
        out.append(ms)
    return out


def main():
    Experiment, TimecourseMeasurement = reference_classes()
    out = {}
    all_cases = cases()
    for c, ms in enumerate(all_cases):
        def build():
            return Experiment('Exp%d' % c, [TimecourseMeasurement(nm, v.copy(), t.copy(), None if s is None else s.copy())
                                            for nm, t, v, s in ms])
        e = build()
        out['n_%d' % c] = np.array(len(ms))
        out['names_%d' % c] = np.array([nm for nm, _, _, _ in ms])
        for j, (nm, t, v, s) in enumerate(ms):
            out['t_%d_%d' % (c, j)], out['v_%d_%d' % (c, j)] = t, v
            out['s_%d_%d' % (c, j)] = np.full(len(t), np.nan) if s is None else s
        out['order_%d' % c] = np.array([m.variable_name for m in e.measurements])
        out['unique_%d' % c] = e.get_unique_timepoints()
        out['unique0_%d' % c] = e.get_unique_timepoints(include_zero=True)
        for j, m in enumerate(e.measurements):
            out['std_%d_%d' % (c, j)] = np.asarray(m.std, dtype=float)
            nz = m.get_nonzero_measurements()
            out['nz_v_%d_%d' % (c, j)], out['nz_s_%d_%d' % (c, j)], out['nz_t_%d_%d' % (c, j)] = (np.asarray(x, dtype=float) for x in nz)
        first = e.measurements[0].variable_name
        e.drop_timepoint_zero(first)
        for j, m in enumerate(e.measurements):
            out['d1_t_%d_%d' % (c, j)] = np.asarray(m.timepoints, dtype=float)
        e.drop_timepoint_zero()
        for j, m in enumerate(e.measurements):
            out['d2_t_%d_%d' % (c, j)], out['d2_v_%d_%d' % (c, j)] = np.asarray(m.timepoints, dtype=float), np.asarray(m.values, dtype=float)
        out['unique_after_%d' % c] = e.get_unique_timepoints(include_zero=True)
    out['n_cases'] = np.array(len(all_cases))
    np.savez_compressed(os.path.join(HERE, 'containers_ref.npz'), **out)
    print('containers_ref.npz written:', len(all_cases), 'cases')


if __name__ == '__main__':
    main()
