"""In-tree native builds (hipcc for gfx950, gcc for the C RHS used by the oracle).

Everything lands in ``sysbio_modeling_amd/_build/`` so that the shared objects
travel with the repository snapshot to the GPU box (they are git-ignored).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)
CSRC_DIR = os.path.join(PKG_DIR, 'csrc')
MODELS_DIR = os.path.join(CSRC_DIR, 'models')
BUILD_DIR = os.path.join(PKG_DIR, '_build')
GEN_DIR = os.path.join(BUILD_DIR, 'gen')
CORE_LIB = os.path.join(BUILD_DIR, 'libsbm_hip.so')
OFFLOAD_ARCH = 'gfx950'


class BuildError(RuntimeError):
    pass


def hipcc_path():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise BuildError("hipcc not found: the HIP kernels cannot be built (set HIPCC or install ROCm)")


def _run(cmd, what):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise BuildError("%s failed (%s):\n%s" % (what, " ".join(cmd), proc.stdout))
    return proc.stdout


_hipcc_id = None


def _toolchain_id():
    """What besides the sources decides the bytes of a build: compiler version and target.  Raises BuildError when
    hipcc is absent or does not answer -- a failed ``hipcc --version`` must not turn into a DIFFERENT stamp (that
    would force a silent full rebuild, or a build attempt on a box that only carries prebuilt files)."""
    global _hipcc_id
    if _hipcc_id is None:
        cc = hipcc_path()
        try:
            proc = subprocess.run([cc, '--version'], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        except OSError as e:
            raise BuildError("%s --version could not be run: %s" % (cc, e))
        if proc.returncode != 0 or not proc.stdout.strip():
            raise BuildError("%s --version failed (%d): %s" % (cc, proc.returncode, proc.stdout[-200:]))
        _hipcc_id = hashlib.sha1((proc.stdout + OFFLOAD_ARCH).encode()).hexdigest()[:12]
    return _hipcc_id


def _stamp_of(cmd, sources):
    """Fingerprint of one build: flags + toolchain + contents of every source."""
    h = hashlib.sha1()
    h.update(_toolchain_id().encode())
    # (paths relative to the repository: the snapshot on a GPU box lives under another root)
    h.update('\0'.join(c.replace(REPO_DIR, '<repo>') for c in cmd[1:]).encode())
    for s in sources:
        with open(s, 'rb') as fh:
            h.update(hashlib.sha1(fh.read()).digest())
    return h.hexdigest()


def _up_to_date(target, stamp):
    try:
        with open(target + '.stamp') as fh:
            return os.path.exists(target) and fh.read().strip() == stamp
    except OSError:
        return False


def _locked_build(target, cmd_for, sources, what, force=False):
    """Build ``target`` with ``cmd_for(tmp_out)`` unless its stamp (flags, toolchain, source contents) is current.
    Safe with one process per GPU starting cold at once: an fcntl lock serialises the builders of one target, the
    compiler writes to a private name and the result is moved into place atomically -- a concurrent dlopen
    sees the old file or the new one, never a half-written one.  When hipcc is absent (a box that only runs
    prebuilt files) an existing target is used as it is."""
    import fcntl
    try:
        ref_cmd = cmd_for(target)         # (resolves hipcc: raises BuildError on a box without it)
        stamp = _stamp_of(ref_cmd, sources)
    except BuildError:
        if os.path.exists(target) and not force:
            return target
        raise
    if not force and _up_to_date(target, stamp):
        return target
    os.makedirs(os.path.dirname(target), exist_ok=True)
    with open(target + '.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _up_to_date(target, stamp):     # another process built it while we waited
                return target
            tmp = '%s.%d.tmp' % (target, os.getpid())
            try:
                _run(cmd_for(tmp), what)
                os.replace(tmp, target)
                with open(target + '.stamp.tmp%d' % os.getpid(), 'w') as fh:
                    fh.write(stamp)
                os.replace(target + '.stamp.tmp%d' % os.getpid(), target + '.stamp')
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return target


def _core_sources():
    srcs = [os.path.join(CSRC_DIR, f) for f in ('sbm_core.hip', 'sbm_plugin.h')]
    srcs.append(os.path.join(REPO_DIR, 'include', 'sbm.h'))
    return srcs


def build_core(force=False, extra_flags=()):
    """libsbm_hip.so: C ABI, contexts, model loading, project assembly kernels."""
    def cmd_for(out):
        return [hipcc_path(), '--offload-arch=' + OFFLOAD_ARCH, '-O3', '-std=c++17', '-fPIC', '-shared',
                '-Wall', '-Wno-unused-function', *extra_flags,
                os.path.join(CSRC_DIR, 'sbm_core.hip'), '-o', out, '-ldl']
    return _locked_build(CORE_LIB, cmd_for, _core_sources(), 'build of libsbm_hip.so', force)


def plugin_path(name):
    return os.path.join(BUILD_DIR, 'sbm_model_%s.so' % name)


def build_plugin(name, header_path, force=False, extra_flags=()):
    """sbm_model_<name>.so from a generated model header."""
    os.makedirs(BUILD_DIR, exist_ok=True)
    # developer A/B builds: SBM_PLUGIN_FLAGS="-DFOO=1" compiles a separately named plugin
    env_flags = tuple(os.environ.get('SBM_PLUGIN_FLAGS', '').split())
    if env_flags:
        name = name + '_' + hashlib.sha1(' '.join(env_flags).encode()).hexdigest()[:8]
        extra_flags = tuple(extra_flags) + env_flags
    out = plugin_path(name)
    # every header of csrc/ goes into the stamp (round 4 added one and the explicit list missed it: stale plugins)
    srcs = [header_path, os.path.join(CSRC_DIR, 'sbm_plugin_main.hip')] + \
        sorted(os.path.join(CSRC_DIR, f) for f in os.listdir(CSRC_DIR) if f.endswith(('.hpp', '.h'))) + \
        [os.path.join(REPO_DIR, 'include', 'sbm.h')]
    def cmd_for(o):
        return [hipcc_path(), '--offload-arch=' + OFFLOAD_ARCH, '-O3', '-std=c++17', '-fPIC', '-shared',
                '-DSBM_MODEL_HEADER="%s"' % os.path.abspath(header_path), *extra_flags,
                os.path.join(CSRC_DIR, 'sbm_plugin_main.hip'), '-o', o]
    return _locked_build(out, cmd_for, srcs, 'build of model plugin %s' % name, force)


def build_c_rhs(name, c_source, force=False):
    """Compile a generated C right-hand side (oracle / CPU-baseline use only)."""
    os.makedirs(GEN_DIR, exist_ok=True)
    digest = hashlib.sha1(c_source.encode()).hexdigest()[:12]
    c_path = os.path.join(GEN_DIR, 'rhs_%s_%s.c' % (name, digest))
    so_path = os.path.join(GEN_DIR, 'rhs_%s_%s.so' % (name, digest))
    if os.path.exists(so_path) and not force:
        return so_path
    cc = shutil.which('gcc') or shutil.which('cc')
    if cc is None:
        raise BuildError("no C compiler for the oracle RHS")
    # several processes may want the same library at once (worker pools of the CPU baselines): each builds
    # under private names and moves the result into place atomically
    tag = '.%d.tmp' % os.getpid()
    with open(c_path + tag + '.c', 'w') as fh:
        fh.write(c_source)
    try:
        _run([cc, '-O2', '-fPIC', '-shared', '-o', so_path + tag, c_path + tag + '.c', '-lm'], 'build of C RHS %s' % name)
        os.replace(c_path + tag + '.c', c_path)
        os.replace(so_path + tag, so_path)
    finally:
        for leftover in (c_path + tag + '.c', so_path + tag):
            if os.path.exists(leftover):
                os.remove(leftover)
    return so_path


def write_generated_header(name, hip_source):
    """Header for a user model (zoo models are committed under csrc/models/)."""
    os.makedirs(GEN_DIR, exist_ok=True)
    digest = hashlib.sha1(hip_source.encode()).hexdigest()[:12]
    path = os.path.join(GEN_DIR, '%s_%s.hpp' % (name, digest))
    if not os.path.exists(path):
        tmp = '%s.%d.tmp' % (path, os.getpid())
        with open(tmp, 'w') as fh:
            fh.write(hip_source)
        os.replace(tmp, path)
    return path, digest
