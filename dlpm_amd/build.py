"""Build libdlpm_amd.so (gfx950) in-tree with hipcc.

    python -m dlpm_amd.build            # incremental: only stale objects are recompiled
    python -m dlpm_amd.build --force

hipcc cross-compiles without a GPU; the resulting .so travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '_obj')
LIB_DIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIB_DIR, 'libdlpm_amd.so')
SOURCES = ['host.cpp', 'png.cpp', 'images.hip', 'noise.hip', 'conv_igemm.hip', 'conv_split.hip', 'conv_wino.hip', 'conv_wino4.hip', 'conv_splitk.hip', 'conv_direct.hip', 'head_fused.hip', 'groupnorm.hip', 'attention.hip', 'block_small.hip',
           'embed.hip', 'unet.hip', 'mlp.hip', 'sampler.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-Wall', '-Wno-unused-function']
# Instrumented / ablation builds (DLPM_BUILD_DEFS="DLPM_PHASE_TIMING F4_X=11 ...") never share objects or the library name
# with the product build: they go to _obj_<tag>/ and lib/libdlpm_amd_<tag>.so (load one with DLPM_LIB=<path>).
DEFS = os.environ.get('DLPM_BUILD_DEFS', '').split()
FLAGS += ['-D' + (d[2:] if d.startswith('-D') else d) for d in DEFS]
XFLAGS = os.environ.get('DLPM_BUILD_FLAGS', '').split()   # extra compiler flags of an experiment build (e.g. -fno-slp-vectorize)
FLAGS += XFLAGS
DEFS = DEFS + XFLAGS
if DEFS:
    import hashlib
    _tag = hashlib.sha256(' '.join(sorted(DEFS)).encode()).hexdigest()[:8]
    OBJ = os.path.join(HERE, '_obj_' + _tag)
    LIB = os.path.join(LIB_DIR, 'libdlpm_amd_%s.so' % _tag)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'dlpm_amd.h'))
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.rsplit('.', 1)[0] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + ['-x', 'hip', '-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print('[dlpm_amd.build]', os.path.basename(cmd[-3]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr))
        if r.stderr.strip() and verbose:
            print(r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-lz']
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s' % r.stderr)
        if verbose:
            print('[dlpm_amd.build] linked', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
