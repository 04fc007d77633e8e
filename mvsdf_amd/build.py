"""Build recipe for libmvsdf_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SO = os.path.join(HERE, 'libmvsdf_hip.so')
SOURCES = ['capi_util.hip', 'basic.hip', 'trace.hip', 'diff_mlp.hip', 'loss_kernels.hip', 'optim_kernels.hip', 'step_kernels.hip', 'sample_kernels.hip', 'step_driver.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math',
         '-Wno-unused-result', '-Wno-pass-failed']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), tag=''):
    """tag != '': a side build (objects / .so get the suffix; used for ablation experiments only)."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    so_path = SO if not tag else SO.replace('.so', '_%s.so' % tag)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'mvsdf_hip.h'))
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', (tag and '_' + tag) + '.o'))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + list(extra_flags) + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % src)
        if verbose and out:
            print(out.decode())
    if force or procs or _stale(so_path, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so_path] + objs
        subprocess.check_call(cmd)
    return so_path


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
