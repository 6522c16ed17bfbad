"""Build recipe for libmultipoint_hip.so (gfx950 only).  hipcc cross-compiles without a GPU.

    python -m multipoint_amd.build [--force]

The shared library is written IN-TREE (multipoint_amd/libmultipoint_hip.so) so that it travels with
the repository snapshot to the GPU box; it is git-ignored (*.so).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ_DIR = os.path.join(CSRC, '_build')
LIB_PATH = os.path.join(HERE, 'libmultipoint_hip.so')
SOURCES = ['conv_mfma.hip', 'conv_wino.hip', 'conv_f16.hip', 'conv_first.hip', 'heads_post.hip', 'head_tail.hip', 'nms.hip', 'keypoints.hip',
           'sample_match.hip', 'match_extra.hip', 'pair_metrics.hip', 'detector_metrics.hip', 'homography.hip', 'homog_adapt.hip', 'api.hip']
HEADERS = [os.path.join(CSRC, 'mp_common.h'),
           os.path.join(HERE, '..', 'include', 'multipoint_hip.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function'] + \
    os.environ.get('MP_HIPCC_FLAGS', '').split()          # e.g. -DMP_TIMING (developer instrumentation)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src):
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + '.o')
    cmd = [HIPCC] + FLAGS + ['-c', os.path.join(CSRC, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, ' '.join(cmd), r.stderr))
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    todo = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + '.o')
        objs.append(obj)
        if force or _stale(obj, [os.path.join(CSRC, src)] + HEADERS):
            todo.append(src)
    if todo:
        if verbose:
            print('[multipoint_amd.build] hipcc --offload-arch=gfx950: %s' % ', '.join(todo), flush=True)
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as ex:
            list(ex.map(_compile, todo))
    if force or todo or _stale(LIB_PATH, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (' '.join(cmd), r.stderr))
        if verbose:
            print('[multipoint_amd.build] linked %s' % LIB_PATH, flush=True)
    return LIB_PATH


if __name__ == '__main__':
    build(force='--force' in sys.argv)
