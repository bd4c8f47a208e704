"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
that include/rnnt_hip.h declares (no compute calls without a GPU)."""
import os
import re

import ctc_vr_amd.lib as rlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rnnt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rnnt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = rlib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"librnnt_hip.so does not export {s}"
    assert sorted(rlib.SIGNATURES) == syms, "ctypes SIGNATURES and include/rnnt_hip.h disagree"
    assert lib.rnnt_abi_version() == 3


def test_config_struct_layout():
    import ctypes
    assert ctypes.sizeof(rlib.RnntConfig) == 10 * 4
    assert [f[0] for f in rlib.RnntConfig._fields_] == [
        "max_streams", "max_chunk_frames", "max_cache_frames", "max_enc_frames", "max_tokens", "vocab_size", "blank_id",
        "n_steps", "device", "max_beam"]


def test_null_context_error_string():
    lib = rlib.load()
    assert lib.rnnt_last_error(None) == b"null context"
