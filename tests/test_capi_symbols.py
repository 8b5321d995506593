"""CPU: libvqafuse.so loads and exports every function include/vqaf.h declares, and the
ctypes signature table covers exactly that set (no compute calls here)."""
import ctypes
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "vqaf.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vqaf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = declared_functions()
    assert "vqaf_gather_mix" in names and "vqaf_topk" in names and "vqaf_score" in names
    assert len(names) >= 25


def test_library_exports_every_declared_symbol():
    from vqa_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the extension: make -C vqa_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/vqaf.h but not exported"


def test_ctypes_table_matches_header():
    from vqa_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_functions()
    lib = _lib.load()
    assert lib.vqaf_version() == 100
    assert lib.vqaf_status_string(-1).decode().startswith("unsupported")


def test_argument_validation_without_gpu():
    """Status codes for bad arguments are decided on the host before any launch."""
    from vqa_amd import _lib
    lib = _lib.load()
    assert lib.vqaf_topk(None, 1, 8, 4, None, None, None) == -5          # NULL
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    assert lib.vqaf_topk(p, 1, 8, 9, p, None, None) == -1                # K > N
    assert lib.vqaf_topk(p, 1, 5000, 9, p, None, None) == -1             # N > 4096
    assert lib.vqaf_gather_mix(p, p, None, 0, 0.0, None, 0.05, 1, 8, 4, 12, 0, 0, p, None) == -3  # D % 8
    assert lib.vqaf_gather_mix(p, p, None, 0, 0.0, None, 0.05, 1, 8, 4, 16, 7, 0, p, None) == -2  # dtype
    assert lib.vqaf_gather_mix(p, p, None, 1, 0.1, None, 0.05, 1, 8, 4, 16, 0, 0, p, None) == -5  # rng NULL
    assert lib.vqaf_score_workspace_bytes(4, 197, 512, 8) >= 4 * 197 * 8 * 4
