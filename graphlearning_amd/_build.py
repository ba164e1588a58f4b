"""Build libglx.so (the gfx950 HIP library behind the C-ABI of include/glx.h) in-tree.

hipcc cross-compiles for gfx950 without a GPU; the resulting .so sits next to this
file so that it travels with the source tree (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import glob

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libglx.so')


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + [os.path.join(HERE, '..', 'include', 'glx.h')]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 into graphlearning_amd/libglx.so."""
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    if not os.path.exists(hipcc):
        hipcc = 'hipcc'
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-value',
           '-Wno-unused-result'] + os.environ.get('GLX_CXXFLAGS', '').split() + ['-o', LIB + '.tmp'] + sources()
    if verbose:
        print(' '.join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    print(build_lib(force=True, verbose=True))
