"""Build recipe for libmvsdf_hip.so (hipcc, gfx950 only, in-tree).

Staleness is decided by CONTENT, not by file times: every object has a stamp holding the SHA-256 of its source, of every header under csrc/ + include/ and
of the compiler flags; the library has one over its objects' stamps.  A binary that travelled with a snapshot (built `.so` / `.o` files are git-ignored but do
ship to the GPU box) is reused only if it was built from exactly the sources beside it -- a touched file without a change rebuilds nothing, a changed byte
rebuilds what depends on it (tests/test_abi.py::test_build_is_keyed_by_content).

build(tag='dev') makes libmvsdf_hip_dev.so with -DMVSDF_DEV_SWITCHES: the same sources with the development A/B switches compiled in (csrc/capi_util.h::mv_dev_env);
tests/test_gpu_alt_paths.py and the sweep tools load it through MVSDF_LIB.  The product library reads none of them."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
SO = os.path.join(HERE, 'libmvsdf_hip.so')
SOURCES = ['capi_util.hip', 'basic.hip', 'trace.hip', 'diff_mlp.hip', 'loss_kernels.hip', 'optim_kernels.hip', 'step_kernels.hip', 'sample_kernels.hip', 'step_driver.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math',
         '-Wno-unused-result', '-Wno-pass-failed']
TAG_FLAGS = {'dev': ['-DMVSDF_DEV_SWITCHES']}


def _sha(paths, extra=''):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b'\0')
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_is(path, key):
    try:
        with open(path + '.stamp') as f:
            return f.read().strip() == key
    except OSError:
        return False


def so_path(tag=''):
    return SO if not tag else SO.replace('.so', '_%s.so' % tag)


def build(force=False, verbose=False, extra_flags=(), tag=''):
    """tag != '': a side build (objects / .so get the suffix): 'dev' = the development switches (see above); anything else with extra_flags = ablation experiments."""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    target = so_path(tag)
    flags = FLAGS + TAG_FLAGS.get(tag, []) + list(extra_flags)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))
    headers.append(os.path.join(HERE, '..', 'include', 'mvsdf_hip.h'))
    hkey = _sha(headers, ' '.join(flags))
    objs, keys, procs = [], [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace('.hip', (tag and '_' + tag) + '.o'))
        key = _sha([s], hkey)
        objs.append(o)
        keys.append(key)
        if force or not os.path.exists(o) or not _stamp_is(o, key):
            cmd = [hipcc] + flags + ['-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd))
            procs.append((src, o, key, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, o, key, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError('hipcc failed on %s' % src)
        with open(o + '.stamp', 'w') as f:
            f.write(key)
        if verbose and out:
            print(out.decode())
    lkey = hashlib.sha256(' '.join(keys).encode()).hexdigest()
    if force or procs or not os.path.exists(target) or not _stamp_is(target, lkey):
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', target] + objs)
        with open(target + '.stamp', 'w') as f:
            f.write(lkey)
    return target


if __name__ == '__main__':
    tag = ([a.split('=', 1)[1] for a in sys.argv if a.startswith('--tag=')] or [''])[0]
    print(build(force='--force' in sys.argv, verbose=True, tag=tag))
