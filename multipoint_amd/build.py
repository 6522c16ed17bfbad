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
SOURCES = ['conv_mfma.hip', 'conv_wino43.hip', 'conv_wino43b.hip', 'conv_split.hip', 'conv_f16.hip', 'conv_f16_res.hip', 'conv_first.hip', 'heads_post.hip', 'head_tail.hip', 'head_tail_f16.hip', 'nms.hip', 'keypoints.hip',
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


# sources whose inline-asm LDS-DMA statements read SGPR base pointers: hipcc pads no hazards inside an asm string, so the
# generated code is checked instead of padding every statement with s_nop (which costs 1 % of a launch)
DMA_SOURCES = ('conv_wino43.hip', 'conv_wino43b.hip', 'head_tail.hip', 'head_tail_f16.hip')


def check_dma_hazards(asm_path, wait_states=5):
    """Fail if a VALU write of an SGPR (v_readlane / v_readfirstlane: SGPR-spill reloads, uniformity copies) lands fewer than
    `wait_states` wait states in front of a global_load_lds that uses that SGPR as its base (every instruction in between is
    one wait state, `s_nop N` is N + 1): the hardware needs five there and nothing inserts them inside an asm statement."""
    import re
    lines = open(asm_path).read().split('\n')
    n = 0
    for i, l in enumerate(lines):
        m = re.search(r'global_load_lds_dword(?:x4)? v\d+, s\[(\d+):(\d+)\]', l)
        if not m:
            continue
        n += 1
        base = {int(m.group(1)), int(m.group(2))}
        ws, j = 0, i - 1
        while j >= 0 and ws < wait_states:
            t = lines[j].strip()
            j -= 1
            if t.endswith(':') and not t.startswith(('.Lfunc', ';')) and re.match(r'\.LBB\d+_\d+:', t):
                # a branch target: predecessors other than the fall-through are not visible to this linear walk, so the
                # wait states in front of the DMA must all lie INSIDE its own block
                raise RuntimeError('%s:%d: LDS-DMA only %d wait states behind the branch target %s (needs %d inside the '
                                   'block): open the asm statement with s_nop' % (asm_path, j + 2, ws, t, wait_states))
            if not t or t.startswith((';', '.')) or t.endswith(':'):
                continue
            w = re.match(r'(v_readlane_b32|v_readfirstlane_b32) s(\d+),', t)
            if w and int(w.group(2)) in base:
                raise RuntimeError('%s:%d: %s writes the base SGPR of an LDS-DMA only %d wait states later (needs %d): '
                                   'open the asm statement with s_nop' % (asm_path, j + 2, t, ws, wait_states))
            nop = re.match(r's_nop (\d+)', t)
            ws += int(nop.group(1)) + 1 if nop else 1
    if n == 0:
        raise RuntimeError('%s: no global_load_lds found -- the hazard check is looking at the wrong file' % asm_path)
    return n


# Convolution kernels must not use scratch memory: a spilled register's reload is a vector-memory load, and the wait in front of its
# first use (s_waitcnt vmcnt(0)) also waits for every LDS-DMA, staging load and epilogue store in flight (round-5 verdict).  The
# generated code of these sources is therefore checked per kernel (.private_segment_fixed_size of the code-object metadata) against
# a cap in bytes per lane; the caps above zero are the known exceptions, so that a REGRESSION fails the build:
#   conv_wino43b.hip   the any-frame-size F(4x4,3x3) kernel runs ONE wave per SIMD on all 512 registers (288 accumulators); its 13-46
#                      spilled registers sit in the item hand-over and the epilogue, 1-2 reloads per unit body (docs/HISTORY.md 3.3)
#   conv_f16.hip       the pooled instantiations for 16- and 8-pixel-wide M-blocks (frames narrower than 32 pixels) keep 1-2; the
#                      32-wide ones every benchmark shape runs have none since round 6
SCRATCH_CAPS = {'conv_mfma.hip': 0, 'conv_wino43.hip': 0, 'conv_wino43b.hip': 136, 'conv_split.hip': 0, 'conv_f16.hip': 12,
                'conv_f16_res.hip': 0, 'conv_first.hip': 0, 'head_tail.hip': 0, 'head_tail_f16.hip': 0}


def check_scratch(asm_path, cap):
    """Fail if a kernel of this source uses more than `cap` bytes of scratch per lane.  Returns {kernel: bytes} of the non-zero ones."""
    import re
    text = open(asm_path).read()
    found = re.findall(r'\.name:\s+(\S+)\n(?:(?!\.name:).*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)', text)
    if not found:
        raise RuntimeError('%s: no kernel metadata found -- the scratch check is looking at the wrong file' % asm_path)
    used = {k: int(v) for k, v in found if int(v) > 0}
    over = {k: v for k, v in used.items() if v > cap}
    if over:
        raise RuntimeError('%s: %d kernel(s) use more scratch than the %d bytes per lane this source is allowed (spilled registers: '
                           'their reloads wait for every load in flight): %s' % (asm_path, len(over), cap, ', '.join(
                               '%s %d B' % (k, v) for k, v in sorted(over.items())[:6])))
    return used


def _compile(src):
    """Compile one source.  The object only appears under its final name (which `_stale` looks at) AFTER the LDS-DMA hazard
    check has passed: a failed check must not leave a fresh-looking object behind for the next build to link."""
    stem = os.path.splitext(src)[0]
    obj = os.path.join(OBJ_DIR, stem + '.o')
    tmp_stem = stem + '.tmp%d' % os.getpid()
    tmp = os.path.join(OBJ_DIR, tmp_stem + '.o')
    extra = ['-save-temps=obj'] if (src in DMA_SOURCES or src in SCRATCH_CAPS) else []
    cmd = [HIPCC] + FLAGS + extra + ['-c', os.path.join(CSRC, src), '-o', tmp]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, ' '.join(cmd), r.stderr))
        if src in DMA_SOURCES:
            check_dma_hazards(os.path.join(OBJ_DIR, stem + '-hip-amdgcn-amd-amdhsa-gfx950.s'))   # named after the source
        if src in SCRATCH_CAPS:
            check_scratch(os.path.join(OBJ_DIR, stem + '-hip-amdgcn-amd-amdhsa-gfx950.s'), SCRATCH_CAPS[src])
        os.replace(tmp, obj)
    finally:
        for f in os.listdir(OBJ_DIR):                      # the temporary object and the -save-temps leftovers (19 MB)
            if f.startswith(tmp_stem) or f.startswith(stem + '-') or f.startswith(stem + '.hip-'):
                os.remove(os.path.join(OBJ_DIR, f))
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
