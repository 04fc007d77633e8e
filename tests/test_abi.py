"""The C-ABI library builds for gfx950, loads (no GPU needed) and exports every symbol include/mvsdf_hip.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def test_header_matches_exports_and_so():
    from mvsdf_amd import _lib, build
    so = build.build()
    hdr = open(os.path.join(ROOT, 'include', 'mvsdf_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = sorted(set(re.findall(r'\b(mvsdf_\w+)\s*\(', hdr)))
    assert declared == sorted(_lib.EXPORTS), (set(declared) ^ set(_lib.EXPORTS))
    L = ctypes.CDLL(so)
    for name in declared:
        assert getattr(L, name) is not None
    assert _lib.lib().mvsdf_version() >= 100
    assert _lib.lib().mvsdf_packed_floats(258, 256) == 272 * 256
    assert _lib.lib().mvsdf_packed_floats(256, 39) == 256 * 64          # K padded to 32


def test_product_does_not_import_oracle():
    """The product path must never route through the oracle (tier rule): no file under mvsdf_amd/ mentions it."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, 'mvsdf_amd')):
        for fn in fns:
            if fn.endswith(('.py', '.h', '.hip', '.cpp')):
                txt = open(os.path.join(dp, fn), errors='ignore').read()
                if re.search(r'^\s*(from|import)\s+oracle\b|#include\s+"[^"]*oracle', txt, flags=re.M):
                    bad.append(fn)
    assert not bad, bad


def test_ctypes_structs_have_the_sizes_the_library_was_compiled_with():
    """The ctypes mirrors of the header's structs (mvsdf_amd/_lib.py, native_step.py) against sizeof() inside the library: a field added on one
    side only shows up here instead of as silently shifted arguments."""
    from mvsdf_amd import _lib, native_step as ns
    out = (ctypes.c_size_t * 8)()
    _lib.lib().mvsdf_abi_struct_sizes.argtypes = [ctypes.POINTER(ctypes.c_size_t)]
    assert _lib.lib().mvsdf_abi_struct_sizes(out) == 8
    mirrors = [_lib.NetDesc, _lib.TraceParams, ns.StepDesc, ns.StepParams, ns.StepInputs, ns.StepLayout, ns.LossArgs, ns.LossLayout]
    for cls, size in zip(mirrors, list(out)):
        assert ctypes.sizeof(cls) == size, (cls.__name__, ctypes.sizeof(cls), size)



def test_build_is_keyed_by_content(tmp_path, monkeypatch):
    """mvsdf_amd/build.py decides by CONTENT (SHA-256 of source + every header + flags, kept in a stamp beside each object and the library), not by file times:
    a prebuilt binary that travelled with a snapshot is reused only if it was built from exactly the sources beside it.  Driven with a stand-in compiler (a script
    that records its calls) on a scratch copy of the layout: touching a file rebuilds nothing, one changed byte rebuilds that object and relinks, a header change
    rebuilds everything, another flag set (tag='dev') is its own build."""
    import importlib
    import stat
    from mvsdf_amd import build as B
    csrc = tmp_path / 'pkg' / 'csrc'
    csrc.mkdir(parents=True)
    (tmp_path / 'include').mkdir()
    (tmp_path / 'include' / 'mvsdf_hip.h').write_text('// abi\n')
    (csrc / 'x.h').write_text('// header\n')
    for n in ('a.hip', 'b.hip'):
        (csrc / n).write_text('// %s\n' % n)
    log = tmp_path / 'calls.log'
    cc = tmp_path / 'fakecc'
    cc.write_text('#!/bin/bash\nout=""; prev=""; for a in "$@"; do if [ "$prev" = "-o" ]; then out="$a"; fi; prev="$a"; done\necho "$@" >> %s\necho built > "$out"\n' % log)
    cc.chmod(cc.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv('HIPCC', str(cc))
    monkeypatch.setattr(B, 'HERE', str(tmp_path / 'pkg'))
    monkeypatch.setattr(B, 'CSRC', str(csrc))
    monkeypatch.setattr(B, 'SO', str(tmp_path / 'pkg' / 'libx.so'))
    monkeypatch.setattr(B, 'SOURCES', ['a.hip', 'b.hip'])

    def calls():
        n = log.read_text().count('\n') if log.exists() else 0
        log.write_text('')
        return n
    so = B.build()
    assert os.path.exists(so) and calls() == 3                     # two objects + the link
    B.build()
    assert calls() == 0
    os.utime(csrc / 'a.hip', (1, 1))                               # older ...
    os.utime(csrc / 'b.hip', None)                                 # ... and newer than the objects: file times decide nothing
    B.build()
    assert calls() == 0
    (csrc / 'b.hip').write_text('// b.hip!\n')                     # one byte more
    B.build()
    txt = log.read_text()
    assert calls() == 2 and 'b.hip' in txt and 'a.hip' not in txt   # that object + the link
    (csrc / 'x.h').write_text('// header 2\n')
    B.build()
    assert calls() == 3
    os.remove(so)                                                  # a lost library is relinked from the objects
    B.build()
    assert calls() == 1
    dev = B.build(tag='dev')                                       # another flag set: its own objects / library / stamps
    assert dev != so and os.path.exists(dev) and '-DMVSDF_DEV_SWITCHES' in log.read_text() and calls() == 3
    B.build()
    B.build(tag='dev')
    assert calls() == 0
