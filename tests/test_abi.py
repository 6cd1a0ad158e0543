"""The C-ABI library loads and exports every symbol include/sempyr.h declares; the ctypes table matches the header
(argument counts); host-only planning entry points behave (no GPU needed, no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'sempyr.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|const char\*)\s+(sempyr_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ('', 'void') else len([a for a in args.split(',') if a.strip()])
        out[m.group(1)] = n
    return out


def test_library_exports_every_declared_symbol():
    from semantic_pyramid_for_image_generation_amd import _lib
    decl = header_functions()
    assert len(decl) >= 30
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(raw, name), f'{name} declared in sempyr.h but not exported'
    for name, argtypes in _lib.SIGNATURES.items():
        assert name in decl, f'{name} bound in _lib.py but not declared in sempyr.h'
        assert len(argtypes) == decl[name], f'{name}: {len(argtypes)} ctypes args vs {decl[name]} in the header'
    assert set(decl) - set(_lib.SIGNATURES) == {'sempyr_last_error_string'}
    assert _lib.lib.sempyr_version() == 1


def test_conv_plan_and_tables_host_only():
    from semantic_pyramid_for_image_generation_amd import _lib
    L = _lib.lib
    d = _lib.ConvDesc(16, 64, 128, 128, 64, 4, 4, 2, 1, 0)     # D block 1: 64->64, 4x4 s2, 128^2 -> 64^2
    p = _lib.ConvPlanInfo()
    assert L.sempyr_conv2d_plan(ctypes.byref(d), _lib.CONV_FWD, ctypes.byref(p)) == 0
    assert (p.Ho, p.Wo, p.K[0], p.Mpad) == (64, 64, 64 * 16, 64)
    assert L.sempyr_conv2d_plan(ctypes.byref(d), _lib.CONV_DGRAD, ctypes.byref(p)) == 0
    assert p.nphases == 4 and list(p.K) == [256] * 4          # 2x2 taps per stride phase
    n = p.tab_entries
    kt = (ctypes.c_int32 * (2 * n))(); ws = (ctypes.c_int32 * n)()
    assert L.sempyr_conv2d_tables(ctypes.byref(d), _lib.CONV_DGRAD, ctypes.cast(kt, ctypes.c_void_p), ctypes.cast(ws, ctypes.c_void_p)) == 0
    # every weight element is used exactly once over the 4 phases
    used = sorted(w for w in ws if w >= 0)
    assert len(used) == 64 * 16 and len(set(used)) == len(used)
    # 1x1 stride-2 shortcut: three of the four phases have no taps
    d2 = _lib.ConvDesc(2, 64, 16, 16, 128, 1, 1, 2, 0, 0)
    assert L.sempyr_conv2d_plan(ctypes.byref(d2), _lib.CONV_DGRAD, ctypes.byref(p)) == 0
    assert list(p.K) == [128, 0, 0, 0]
    # invalid descriptors are refused with a message, never silently accepted
    bad = _lib.ConvDesc(1, 3, 8, 8, 4, 3, 3, 3, 1, 0)
    assert L.sempyr_conv2d_plan(ctypes.byref(bad), _lib.CONV_FWD, ctypes.byref(p)) < 0
    assert b'stride' in L.sempyr_last_error_string()
