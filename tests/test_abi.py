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

