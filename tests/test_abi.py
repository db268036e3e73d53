"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header declares."""
import os
import re

from rgbd_gan_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rgbd_gan_hip.h")).read()
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


def test_bad_arguments_are_reported_not_crashed():
    lib = _lib.load()
    rc = lib.rgbd_conv2d_fprop_bf16(None, None, None, None, None, None, 1, 4, 4, 64, 64, 3, 3, 1, 0, 0, 0.2, None, None)
    assert rc == -1
    assert b"null pointer" in lib.rgbd_last_error()
