"""Builds libechoscene_hip.so in-tree with hipcc for gfx950 (no JIT cache: the .so travels with
the repo snapshot to the GPU box).  ``python -m echoscene_amd.build`` or ``build()``."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# ES_BUILD_TAG=_stamp (with ES_BUILD_FLAGS=-DES_STAMP): an instrumented build next to the product library, loaded with ES_LIB_TAG=_stamp
TAG = os.environ.get('ES_BUILD_TAG', '')
LIB = os.path.join(HERE, 'libechoscene_hip%s.so' % TAG)
SOURCES = ['es_runtime.hip', 'es_rows.hip', 'es_vol.hip', 'es_vol32.hip', 'es_chamfer.hip', 'es_mc.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function'] + \
    os.environ.get('ES_BUILD_FLAGS', '').split()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(HERE, '..', 'include', 'echoscene_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace('.hip', TAG + '.o'))
        cmd = ['hipcc'] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError('hipcc failed: ' + ' '.join(cmd))
    cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
