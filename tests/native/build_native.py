"""Builds tests/native/libmfma_check.so (hipcc, gfx950): test infrastructure, kept out of the product library.  Called by __graft_entry__.build() in the
build container (the .so travels with the snapshot) and, if the file is missing or stale, by the test itself on the GPU box (same image, hipcc present)."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'mfma_check.hip')
SO = os.path.join(HERE, 'libmfma_check.so')
FLAGS = ['--offload-arch=gfx950', '-O2', '-std=c++17', '-fPIC', '-shared', '-Wno-unused-result']


def build(force=False):
    stamp = SO + '.stamp'
    key = hashlib.sha256(open(SRC, 'rb').read() + ' '.join(FLAGS).encode()).hexdigest()
    if not force and os.path.exists(SO) and os.path.exists(stamp) and open(stamp).read().strip() == key:
        return SO
    subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')] + FLAGS + ['-o', SO, SRC])
    open(stamp, 'w').write(key)
    return SO


if __name__ == '__main__':
    print(build(force=True))
