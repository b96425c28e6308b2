"""Symbolic front end: model text -> derived equations -> generated sources.

The reference's ``symbolic`` package exposes (through its tests and CLI) a
``make_jit_model`` facade that does not exist in its tree
(symbolic/__init__.py is empty; tests/test_sympy_tools.py:11).  This package
provides the working equivalent: ``make_ode_model`` returns a
:class:`GeneratedModel` whose ``model`` / ``sens_model`` callables follow the
reference callback contract and whose HIP plugin runs the same equations on
the GPU.
"""
from __future__ import annotations

import os
from collections import OrderedDict

from .sympy_tools import parse_model_file, process_model_dict, derive_sparse_jacobians
from .emit import ModelSpec, Derived, emit_python, emit_c, emit_hip

__all__ = ['parse_model_file', 'process_model_dict', 'make_ode_model', 'make_jit_model',
           'GeneratedModel', 'ModelSpec', 'generated_model_of']


class GeneratedModel(object):
    """All artefacts generated from one :class:`ModelSpec`.

    Attributes
    ----------
    model, sens_model : callables ``f(y, t, yout, p) -> None`` (CPU; they are what
        the oracle integrates with SciPy and what a reference ``OdeModel`` accepts)
    n_vars, param_order, sens_params : layout of y / p / sensitivity columns
    python_source, c_source, hip_source : generated text
    """

    def __init__(self, spec: ModelSpec, header_path=None):
        self.spec = spec
        self.name = spec.name
        self._derived = None
        cached = self._load_sources()
        if cached is None:
            self.python_source = emit_python(spec, self.derived)
            self.c_source = emit_c(spec, self.derived)
            self.hip_source = emit_hip(spec, self.derived)
            self._save_sources()
        else:
            self.python_source, self.c_source, self.hip_source = cached
        self.n_vars = spec.n_vars
        # (a spec read from a plain callable may have renamed parameters that are not identifiers: ingest.py)
        alias = getattr(spec, 'param_aliases', {})
        self.param_order = [alias.get(p, p) for p in spec.params]
        self.sens_params = [alias.get(p, p) for p in spec.sens_params]
        self.n_sens = spec.n_sens
        self._header_path = header_path
        ns = {}
        exec(compile(self.python_source, '<generated %s>' % spec.name, 'exec'), ns)
        self.model = ns['model']
        self.sens_model = ns['sens_model']
        # tag the callables so that OdeModel can find the GPU plugin behind them
        self.model._sbm_generated = self
        self.sens_model._sbm_generated = self
        self._plugin = None
        self._c_lib = None

    # -- derivation cache -------------------------------------------------------------------------------------
    # Deriving and printing a large model costs SymPy time (seconds to a minute for models of this size);
    # the three generated texts are kept under _build/gen/, keyed by the model's equations and the emitters' own
    # sources, so that a second process (a test, the benchmark, a rank of a multi-GPU job) starts in milliseconds.
    @property
    def derived(self):
        if self._derived is None:
            self._derived = Derived(self.spec)
        return self._derived

    def _cache_path(self):
        import hashlib
        from .. import build
        h = hashlib.sha1()
        sp = self.spec
        h.update(repr((sp.name, list(sp.variables), list(sp.params), list(sp.fixed),
                       [(k, str(v)) for k, v in sp.equations.items()],
                       sorted(getattr(sp, 'param_aliases', {}).items()))).encode())
        here = os.path.dirname(os.path.abspath(__file__))
        for fn in sorted(os.listdir(here)):
            if fn.endswith('.py'):
                with open(os.path.join(here, fn), 'rb') as fh:
                    h.update(fh.read())
        import re
        return os.path.join(build.GEN_DIR, 'src_%s_%s.json' % (re.sub(r'[^0-9A-Za-z_]', '_', sp.name), h.hexdigest()[:16]))

    def _load_sources(self):
        import json
        try:
            with open(self._cache_path()) as fh:
                d = json.load(fh)
            return d['python'], d['c'], d['hip']
        except (OSError, ValueError, KeyError):
            return None

    def _save_sources(self):
        import json
        from .. import build
        try:
            os.makedirs(build.GEN_DIR, exist_ok=True)
            path = self._cache_path()
            tmp = '%s.%d.tmp' % (path, os.getpid())
            with open(tmp, 'w') as fh:
                json.dump({'python': self.python_source, 'c': self.c_source, 'hip': self.hip_source}, fh)
            os.replace(tmp, path)
        except OSError:
            pass

    # -- analytic ODE Jacobians (the reference's model_jac / sens_model_jac, Dfun of LSODA) -----------------
    def _jacobian_callables(self):
        if getattr(self, '_jac_fns', None) is None:
            from .emit import emit_python_jacobians
            self.jacobian_python_source = emit_python_jacobians(self.spec, self.derived)
            ns = {}
            exec(compile(self.jacobian_python_source, '<generated %s jacobians>' % self.spec.name, 'exec'), ns)
            self._jac_fns = (ns['model_jac'], ns['sens_model_jac'])
        return self._jac_fns

    @property
    def model_jac(self):
        """``model_jac(y, t, jacout, p)``: jacout[b, a] = d f_a / d y_b, (n, n) (generated on first use)."""
        return self._jacobian_callables()[0]

    @property
    def sens_model_jac(self):
        """Jacobian of the augmented system, (n + n*k) square, same contract (generated on first use)."""
        return self._jacobian_callables()[1]

    # -- operation counts (bench.py: achieved fp64 rate) -------------------------------
    def flop_counts(self):
        """Floating-point operations of the generated code, from the emitter's own CSE'd expressions (every +, -, *, /
        and power counted as one; a reciprocal the GPU refines with two Newton steps still counts as one division):

          f        the state right-hand side alone (n equations)
          f_jac    f, the non-zeros of J_y and of J_p together (what one stage of a sensitivity kernel, or one Newton
                   evaluation of an implicit kernel, evaluates once per trajectory)
          nnz_jy, nnz_jp, nnz_lu   pattern sizes (nnz_lu: L + U of I - h J_y with fill-in, natural order)
        """
        if getattr(self, '_flops', None) is None:
            import sympy
            from . import emit_implicit
            d = self.derived

            def ops(repl, red):
                return int(sum(sympy.count_ops(e) for _, e in repl) + sum(sympy.count_ops(e) for e in red))
            pattern, _ = emit_implicit.symbolic_lu(self.spec.n_vars, [(r, c) for r, c, _ in d.jy])
            self._flops = dict(f=ops(d.repl_f, d.f_only), f_jac=ops(d.repl_all, list(d.f_red) + list(d.jy_red) + list(d.jp_red)),
                               nnz_jy=len(d.jy), nnz_jp=len(d.jp), nnz_lu=len(pattern))
        return dict(self._flops)

    def rowgroup_chunks(self):
        """{'RG0': n, 'RG1': n, 'RG2': n}: wavefronts per trajectory of the row-group splits the emitter planned (RG0: DOPRI45 /
        RK4, RG1: small batches, RG2: DOP853); {} when the model has no row split."""
        import re
        out = {}
        for name in ('RG0', 'RG1', 'RG2'):
            m = re.search(r'struct %s \{.*?RG_NCH = (\d+)' % name, self.hip_source, re.S)
            if m:
                out[name] = int(m.group(1))
        return out

    # -- GPU plugin ---------------------------------------------------------
    def header_path(self):
        from .. import build
        if self._header_path and os.path.exists(self._header_path):
            with open(self._header_path) as fh:
                if fh.read() == self.hip_source:
                    return self._header_path
        path, _ = build.write_generated_header(self.name, self.hip_source)
        return path

    def plugin_path(self, build_if_missing=True):
        """Path of the compiled plugin; built with hipcc on first use."""
        from .. import build
        if self._plugin and os.path.exists(self._plugin):
            return self._plugin
        header = self.header_path()
        tag = self.name if header.startswith(build.MODELS_DIR) else os.path.splitext(os.path.basename(header))[0]
        out = build.plugin_path(tag)
        if build_if_missing:
            out = build.build_plugin(tag, header)
        elif not os.path.exists(out):
            raise build.BuildError("model plugin %s has not been built" % out)
        self._plugin = out
        return out

    # -- compiled C RHS (oracle / CPU baseline only) -------------------------
    def c_library(self):
        import ctypes
        from .. import build
        if self._c_lib is None:
            so = build.build_c_rhs(self.name, self.c_source)
            lib = ctypes.CDLL(so)
            dp = ctypes.POINTER(ctypes.c_double)
            for fn in (lib.sbm_rhs, lib.sbm_sens_rhs):
                fn.argtypes = [dp, ctypes.c_double, dp, dp]
                fn.restype = None
            self._c_lib = lib
        return self._c_lib


def make_ode_model(model, name='Model', fixed_params=None, output_fh=None):
    """Model text / function / file / dict / ModelSpec -> :class:`GeneratedModel`.

    Counterpart of the reference's parse -> process -> make_ode_model chain
    (symbolic/sympy_tools.py:272,325,162).  ``output_fh`` receives the generated
    Python module text, as the reference's ``output_fh`` does (:202-204).
    """
    if isinstance(model, ModelSpec):
        spec = model
    elif isinstance(model, dict):
        spec = ModelSpec.from_model_dict(model, name=name)
    else:
        spec = ModelSpec.from_text(model, name=name, fixed_params=fixed_params)
    gm = GeneratedModel(spec)
    if output_fh is not None:
        output_fh.write(gm.python_source)
    return gm


def make_jit_model(model_fh, output_fh=None, calculate_sensitivities=True, name='Model', fixed_params=None):
    """Name the reference's tests and CLI import (tests/test_sympy_tools.py:11,
    tests/test_utils/michelis_menten_model.py:55-63)."""
    gm = make_ode_model(model_fh, name=name, fixed_params=fixed_params, output_fh=output_fh)
    return gm.sens_model if calculate_sensitivities else gm.model


def generated_model_of(fn):
    """The GeneratedModel behind a generated callable, or None."""
    return getattr(fn, '_sbm_generated', None)


def model_from_callables(model, sens_model, n_vars, param_order, name='Model'):
    """GeneratedModel for a PLAIN right-hand side ``model(y, t, yout, p)`` written in the reference's emitted form
    (``name = p[i]``, ``name = y[i]``, ``yout[i] = expression``: symbolic/sympy_tools.py:100-111,185-195 of the
    reference), read from its source -- see ingest.py.  Raises ingest.IngestError (a TypeError) for anything else."""
    import re
    from .ingest import spec_from_callables
    spec = spec_from_callables(model, sens_model, n_vars, param_order,
                               name=re.sub(r'[^0-9A-Za-z_]', '_', str(name)) or 'Model')
    gm = _INGESTED.get(id(spec))
    if gm is None:
        gm = GeneratedModel(spec)
        _INGESTED[id(spec)] = gm
    return gm


_INGESTED = {}


# ---------------------------------------------------------------------------
# zoo: models whose generated headers are committed under csrc/models/
# ---------------------------------------------------------------------------
_ZOO_CACHE = {}


def zoo_model(name):
    """'simple', 'michaelis_menten', 'cascade20', 'stiff50' -> GeneratedModel (cached)."""
    from .. import models_zoo, build
    if name in _ZOO_CACHE:
        return _ZOO_CACHE[name]
    makers = OrderedDict([
        ('simple', models_zoo.simple_spec),
        ('michaelis_menten', models_zoo.michaelis_menten_spec),
        ('cascade20', models_zoo.cascade_spec),
        ('stiff50', models_zoo.stiff_spec),
    ])
    if name not in makers:
        raise KeyError("unknown zoo model %r (have %s)" % (name, ", ".join(makers)))
    gm = GeneratedModel(makers[name](), header_path=os.path.join(build.MODELS_DIR, name + '.hpp'))
    _ZOO_CACHE[name] = gm
    return gm


ZOO_NAMES = ('simple', 'michaelis_menten', 'cascade20', 'stiff50')
