"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares."""
import os
import re

from rgbd_gan_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header=os.path.join("include", "rgbd_gan_hip.h")):
    text = open(os.path.join(ROOT, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rgbd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.PROTOTYPES) == names
    assert lib.rgbd_abi_version() == _lib.ABI_VERSION
    assert not [n for n in names if "debug" in n], "test hooks belong in csrc/rgbd_debug.h, not in the public header"
    hooks = declared_symbols(os.path.join("rgbd_gan_amd", "csrc", "rgbd_debug.h"))
    assert sorted(list(_lib.DEBUG_PROTOTYPES) + list(_lib.DEBUG_ONLY_PROTOTYPES)) == hooks
    for n in _lib.DEBUG_PROTOTYPES:
        assert hasattr(lib, n), n
    # the process-wide planner switches -- and the reference kernels behind them -- are NOT in the shipped library ...
    for n in _lib.DEBUG_ONLY_PROTOTYPES:
        assert not hasattr(lib, n), f"{n} is exported by the product library"
    # ... only in the debug build, which exports everything the product does
    dlib = _lib.load_debug()
    for n in names + hooks:
        assert hasattr(dlib, n), n


def test_integration_document_names_only_real_entry_points():
    """INTEGRATION.md is the reference-side binding document: it must exist, show a ctypes stub, and
    every rgbd_* symbol it names must be declared by the header (it was once emptied by an unrelated commit)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert len(text.splitlines()) > 50 and "ctypes.CDLL" in text and "argtypes" in text
    public = set(declared_symbols())
    hooks = set(declared_symbols(os.path.join("rgbd_gan_amd", "csrc", "rgbd_debug.h")))
    named = set(re.findall(r"`(rgbd_[a-z0-9_]+)`", text)) | set(re.findall(r"_lib\.(rgbd_[a-z0-9_]+)", text))
    named = {n for n in named if not n.endswith("_")}            # `rgbd_adain_*`-style family names
    assert len(named) > 40
    assert named - public - hooks == set(), sorted(named - public - hooks)
    m = re.search(r"ABI version (\d+)", text)
    assert m and int(m.group(1)) == _lib.ABI_VERSION
    # the stub's argument lists have the header's arity
    for name, n_args in re.findall(r"_lib\.(rgbd_[a-z0-9_]+)\.argtypes = \[([^\]]*)\]", text):
        assert len([a for a in n_args.split(",") if a.strip()]) == len(_lib.PROTOTYPES[name][0]), name


def test_bad_arguments_are_reported_not_crashed():
    lib = _lib.load()
    rc = lib.rgbd_conv2d_fprop_bf16(None, None, None, None, None, None, 1, 4, 4, 64, 64, 3, 3, 1, 0, 0, 0.2, None, 0, None)
    assert rc == -1
    assert b"null pointer" in lib.rgbd_last_error()
