"""CPU: the C-ABI library loads and exports exactly the symbols include/vocr.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "vocr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(vocr_[a-z0-9_]+)\s*\(", src))


def test_library_builds_and_exports_every_declared_symbol():
    from vistaocr_amd import _lib, build
    build.build()
    lib = _lib.load()
    declared = _header_symbols()
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "libvocr.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.vocr_abi_version() == _lib.ABI_VERSION == 3
    assert re.search(r"#define VOCR_ABI_VERSION\s+3\b", open(os.path.join(ROOT, "include", "vocr.h")).read())


def test_argument_validation_without_gpu():
    from vistaocr_amd import _lib
    lib = _lib.load()
    rc = lib.vocr_gemm(0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, None, 0, None)
    assert rc == -1 and b"vocr_gemm" in lib.vocr_last_error()
    rc = lib.vocr_lstm_fwd(None, None, None, None, None, None, None, None, 4, 4, 16, None, None)
    assert rc == -1
    assert lib.vocr_conv3x3_wgrad_workspace_bytes(32, 256, 7, 294, 256) > 0
    assert lib.vocr_ctc_workspace_bytes(294, 32, 96, 20) > 0


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vistaocr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "/root/reference" not in txt, f


def test_forward_fails_loudly_on_cpu():
    import torch
    import vistaocr_amd as va
    m = va.CnnOcrModel(input_line_height=30, rds_line_height=30, alphabet=va.english_alphabet(), lstm_input_dim=16,
                       num_lstm_layers=1, num_lstm_hidden_units=16, p_lstm_dropout=0.0, gpu=False, verbose=False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 30, 40), torch.tensor([40]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        va.CTCLoss()(torch.zeros(4, 1, 96), torch.tensor([1], dtype=torch.int32), torch.tensor([4], dtype=torch.int32),
                     torch.tensor([1], dtype=torch.int32))
