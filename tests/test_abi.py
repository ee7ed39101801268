"""CPU: the C-ABI boundary.  Every function include/dmvs.h declares must be bound in
diffmvs_amd/_lib.py and exported by the built gfx950 library (no compute calls here: no GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from diffmvs_amd import _lib


def header_functions():
    src = open(os.path.join(ROOT, "include", "dmvs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(dmvs_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(_lib.SIGNATURES)


def test_library_builds_and_exports_every_symbol():
    from diffmvs_amd.build import build_hip
    path = build_hip()
    dll = ctypes.CDLL(path)
    for name in header_functions():
        assert hasattr(dll, name), name
    assert dll.dmvs_abi_version() == _lib.ABI_VERSION
    _lib.Lib(path)


def test_descriptor_structs_match_header_field_order():
    src = open(os.path.join(ROOT, "include", "dmvs.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, cls in (("dmvs_conv2d_desc", _lib.Conv2dDesc), ("dmvs_conv3d_desc", _lib.Conv3dDesc),
                       ("dmvs_getcost_desc", _lib.GetCostDesc)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, flags=re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            typ, names = decl.rsplit(" ", 1)[0], decl
            names = re.sub(r"^(const\s+)?(float|double|int32_t)\s*\*?", "", decl)
            for n in names.split(","):
                fields.append(n.replace("*", "").strip())
        got = [f[0].rstrip("_") for f in cls._fields_]
        assert got == fields, (cname, got, fields)


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_lib.DmvsError):
        _lib.Lib(str(tmp_path / "nope.so"))
