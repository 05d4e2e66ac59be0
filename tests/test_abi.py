"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol
include/ehx.h declares, and fails loudly (no CPU fallback) when no gfx950 device is present."""
import ctypes as C
import os
import re

import pytest

import __graft_entry__ as graft
from embeddinghub_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        graft.build()
    return _lib.load()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "ehx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ehx_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    declared = _declared_symbols()
    assert declared, "no declarations parsed from include/ehx.h"
    assert sorted(_lib.SYMBOLS) == declared


def test_every_declared_symbol_is_exported(lib):
    raw = C.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(raw, name), "libehx.so does not export %s" % name


def test_abi_version(lib):
    assert lib.ehx_abi_version() == 5


def test_struct_layouts_match_header(tmp_path):
    """sizeof / offsetof of every struct field as a C compiler sees include/ehx.h == the ctypes mirror."""
    import subprocess
    fields = {"ehx_params": [f for f, _ in _lib.Params._fields_], "ehx_stats_t": [f for f, _ in _lib.Stats._fields_]}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "ehx.h"', 'int main(void) {']
    for st, fs in fields.items():
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (st, st))
        for f in fs:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (st, f, st, f))
    src += ['return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)]).decode().splitlines())
    for st, cls in (("ehx_params", _lib.Params), ("ehx_stats_t", _lib.Stats)):
        assert int(got[st]) == C.sizeof(cls), st
        for f, _ in cls._fields_:
            assert int(got["%s.%s" % (st, f)]) == getattr(cls, f).offset, (st, f)


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    rc = lib.ehx_init(None, 0)
    assert rc == _lib.ENODEVICE
    assert b"no CPU fallback" in lib.ehx_last_error()
    h = C.c_void_p()
    rc = lib.ehx_space_create(b"x", 1, 4, 0, 0, None, C.byref(h))
    assert rc == _lib.ENODEVICE and not h.value


def test_argument_validation_without_device(lib):
    h = C.c_void_p()
    assert lib.ehx_space_create(b"x", 1, 0, 0, 0, None, C.byref(h)) == _lib.EINVAL   # dims = 0
    assert lib.ehx_space_create(b"x", 1, 4, 9, 0, None, C.byref(h)) == _lib.EINVAL   # metric
    assert lib.ehx_space_create(b"x", 1, 4, 0, 7, None, C.byref(h)) == _lib.EUNSUPPORTED  # dtype
    assert lib.ehx_space_open(b"nope", 4, C.byref(h)) == _lib.ENOTFOUND
    assert lib.ehx_last_error() == b"Not found"  # server.cc:178 wording
    assert lib.ehx_set(None, b"k", 1, None) == _lib.EINVAL


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under embeddinghub_amd/ may reference it."""
    pkg = os.path.join(ROOT, "embeddinghub_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                for line in txt.splitlines():
                    s = line.strip()
                    if s.startswith(("import ", "from ", "#include")):
                        assert "oracle" not in s or "oracle/" in s and s.startswith("//"), (f, s)


def test_environment_knobs_live_in_one_table_and_are_documented():
    """every environment knob of the engine is read once, in csrc/ehx_env.h — nothing else in csrc/ calls
    getenv — and INTEGRATION.md §7 lists every one of them"""
    csrc = os.path.join(ROOT, "embeddinghub_amd", "csrc")
    env_h = open(os.path.join(csrc, "ehx_env.h")).read()
    knobs = sorted(set(re.findall(r'str\("(EHX_[A-Z0-9_]+)"\)|flag\("(EHX_[A-Z0-9_]+)"', env_h)))
    knobs = sorted({a or b for a, b in knobs})
    assert len(knobs) >= 25, knobs
    for name in sorted(os.listdir(csrc)):
        if name == "ehx_env.h" or not name.endswith((".cpp", ".hip", ".h")):
            continue
        src = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, name)).read())   # (comments may mention it)
        assert "getenv(" not in src, "%s calls getenv: knobs belong in ehx_env.h" % name
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [k for k in knobs if "`%s`" % k not in doc]
    assert not missing, "INTEGRATION.md §7 does not list %s" % missing
