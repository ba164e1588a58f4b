"""Build libglx.so (the gfx950 HIP library behind the C-ABI of include/glx.h) in-tree.

hipcc cross-compiles for gfx950 without a GPU; the resulting .so sits next to this
file so that it travels with the source tree (it is git-ignored, not gpurun-ignored).
Every csrc/*.hip is compiled to its own object (in parallel), then linked; `force=True`
(what __graft_entry__.build() uses) recompiles everything, otherwise only sources newer
than their object are recompiled.
"""
import os
import subprocess
import glob
import hashlib
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libglx.so')
# RCCL (csrc/dist.hip) is bound at run time with dlopen, so the library loads without it
LINK_LIBS = ['-ldl', '-lpthread']


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def headers():
    return glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', 'include', h) for h in ('glx.h', 'glx_experimental.h')]


def _hipcc():
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    return hipcc if os.path.exists(hipcc) else 'hipcc'


# -amdgpu-mfma-vgpr-form: the kNN tile kernels read every accumulator right after the MFMA chain; with the accumulators in
# AGPRs that is 16 v_accvgpr_read per 32x32 tile and 120 registers (3 waves per SIMD) instead of 108 (4)
BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-Wno-unused-result',
              '-mllvm', '-amdgpu-mfma-vgpr-form']


def _flags():
    return BASE_FLAGS + os.environ.get('GLX_CXXFLAGS', '').split()


def source_hash():
    """Digest of every source the library is built from (bench.py stamps it on what it reports)."""
    h = hashlib.sha256()
    for p in sorted(sources() + headers()):
        with open(p, 'rb') as f:
            h.update(os.path.basename(p).encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def _flag_tag():
    return hashlib.sha256(' '.join(_flags()).encode()).hexdigest()[:8]


def needs_build():
    if not os.path.exists(LIB):
        return True
    stamp = os.path.join(HERE, 'libglx.hash')
    built_with = open(stamp).read().split() if os.path.exists(stamp) else []
    if len(built_with) < 2 or built_with[1] != _flag_tag():        # other compile flags (GLX_CXXFLAGS) than the library was built with
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + headers() if os.path.exists(d))


def build_lib(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 into graphlearning_amd/libglx.so."""
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in headers() if os.path.exists(h))
    flag_tag = _flag_tag()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.' + flag_tag + '.o')
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([_hipcc()] + _flags() + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + ' '.join(cmd) + '\n' + res.stdout + res.stderr)
    with ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    run([_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB + '.tmp'] + objs + LINK_LIBS)
    os.replace(LIB + '.tmp', LIB)
    with open(os.path.join(HERE, 'libglx.hash'), 'w') as f:
        f.write(source_hash() + ' ' + _flag_tag() + '\n')
    return LIB


if __name__ == '__main__':
    import sys
    print(build_lib(force='--force' in sys.argv, verbose=True))
