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
    assert lib.vocr_abi_version() == _lib.ABI_VERSION == 5
    assert re.search(r"#define VOCR_ABI_VERSION\s+5\b", open(os.path.join(ROOT, "include", "vocr.h")).read())


def test_stale_or_foreign_library_is_refused(monkeypatch):
    """A libvocr.so that answers another ABI version than the binding was written for must not be used (the version was
    never compared; a stale shipped binary was only caught if a symbol was missing)."""
    from vistaocr_amd import _lib
    _lib.load()
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.load()


def test_build_is_gated_by_a_hash_not_by_file_times(tmp_path, monkeypatch):
    """vistaocr_amd/build.py: an unchanged tree does not rebuild; a tree whose hash differs from the stamp does, whatever the file times
    say (a fresh checkout with a shipped .so, another compiler)."""
    import json
    import os as _os
    from vistaocr_amd import build
    build.build()
    cid = build._compiler_id(build._hipcc())
    assert build._stale(build._tree_hash(), cid) is None
    build.build()
    assert build.build_report()["built"] is False
    _os.utime(_os.path.join(build.CSRC, "misc.cpp"), None)                     # newer source file, same content: still up to date
    assert build._stale(build._tree_hash(), cid) is None
    assert build._stale("0" * 64, cid) == "sources or flags changed"           # another tree hash: stale
    assert build._stale(build._tree_hash(), "0" * 64) == "compiler changed"    # another hipcc: stale
    stamp = json.load(open(build.STAMP))
    monkeypatch.setattr(build, "STAMP", str(tmp_path / "stamp.json"))
    json.dump(dict(stamp, lib="0" * 64), open(build.STAMP, "w"))
    assert build._stale(stamp["tree"], cid) == "library is not the one the stamp describes"


def test_host_without_hipcc_accepts_the_shipped_library(monkeypatch):
    """with no compiler on the host the hash could never equal the build box's stamp (the compiler's version string was
    part of it) and build() raised 'hipcc not found' on an up-to-date tree.  Sources / flags and compiler are separate stamp fields now."""
    from vistaocr_amd import build
    build.build()
    monkeypatch.setattr(build, "_hipcc", lambda: None)
    assert build._stale(build._tree_hash(), None) is None
    assert build.build() == build.LIB
    assert build.build_report() == {"built": False, "reason": "no compiler here, shipped binary matches sources and flags"}
    monkeypatch.setattr(build, "_tree_hash", lambda: "0" * 64)                 # sources really differ: only then is the compiler missed
    with pytest.raises(RuntimeError, match="hipcc not found"):
        build.build()


def test_argument_validation_without_gpu():
    from vistaocr_amd import _lib
    lib = _lib.load()
    rc = lib.vocr_gemm(0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, 0, 0, None, 0, None)
    assert rc == -1 and b"vocr_gemm" in lib.vocr_last_error()
    rc = lib.vocr_lstm_fwd(None, None, None, None, None, None, None, None, 4, 4, 16, None, None)
    assert rc == -1
    # vocr_gemm_pair validates its leading dimensions before it builds buffer ranges from them
    one = ctypes.c_void_p(16)
    rc = lib.vocr_gemm_pair(0, 0, 0, 256, 256, 64, one, one, 8, one, one, 256, one, one, 256, None, None, 0, None, 0, None)
    assert rc == -1 and b"leading dimension" in lib.vocr_last_error()
    assert lib.vocr_conv3x3_wgrad_workspace_bytes(32, 256, 7, 294, 256) > 0
    assert lib.vocr_ctc_workspace_bytes(294, 32, 96, 20) > 0
    # round 5 entry points: shape answers that need no device, and argument validation in front of any launch
    assert lib.vocr_f16_padded_row(294) == 304 and lib.vocr_f16_padded_row(600) == 608 and lib.vocr_f16_padded_row(0) == 0
    assert lib.vocr_conv3x3_h16_supported(256, 256) == 1 and lib.vocr_conv3x3_h16_supported(1, 16) == 0 and lib.vocr_conv3x3_h16_supported(24, 64) == 0
    assert lib.vocr_conv3x3_wgrad_h16_supported(64, 64) == 1 and lib.vocr_conv3x3_wgrad_h16_supported(16, 64) == 0
    assert lib.vocr_conv3x3_wgrad_h16_supported(128, 192) == 0                       # Cout must be 64 or a multiple of 128
    assert lib.vocr_conv3x3_c1_wgrad_workspace_bytes(32, 60, 16) > 0
    assert lib.vocr_lstm_packed_supported(0, 512) == 0 and lib.vocr_lstm_packed_supported(33, 512) == 0 and lib.vocr_lstm_packed_supported(32, 128) == 0
    assert lib.vocr_seq_rowmap(None, 4, 4, 16, None, None, None) == -1 and b"vocr_seq_rowmap" in lib.vocr_last_error()
    assert lib.vocr_seq_rowmap(one, 4, 4, 15, one, None, None) == -1                 # rows must be a multiple of 4
    assert lib.vocr_gather_rows(one, one, None, 4, 8, None, None) == -1
    assert lib.vocr_lstm_fwd_packed(one, one, one, one, one, one, one, one, 4, 4, 512, 0, None, None) == -1
    assert lib.vocr_f32_to_f16_layouts(one, None, None, 1, 16, 4, 4, None) == -1      # at least one output
    assert lib.vocr_f32_to_f16_layouts(one, one, None, 1, 24, 4, 4, None) == -1       # the channel-blocked copy needs C % 16 == 0
    assert lib.vocr_conv3x3_h16_fwd(one, one, None, one, 1, 24, 4, 4, 64, None) == -1
    assert lib.vocr_conv3x3_c1_fwd(None, one, None, one, 1, 4, 4, 16, 0, None) == -1
    # round 6 entry points
    assert lib.vocr_lstm_xproj_pack_bytes(512) == 2 * 8 * 512 * 512 * 4 and lib.vocr_lstm_xproj_pack_bytes(0) == 0
    assert lib.vocr_lstm_xproj_pack(one, one, one, 256, None) == -1 and b"H must be 512" in lib.vocr_last_error()
    assert lib.vocr_lstm_follow_supported(16, 512) == 0 and lib.vocr_lstm_follow_supported(33, 512) == 0 and lib.vocr_lstm_follow_supported(32, 256) == 0
    # planes without a pack / a pack without planes / a follower on a shape that has none: refused in front of any launch
    lead = lambda *tail: lib.vocr_lstm_fwd_lead(one, None, one, one, one, one, one, one, one, 8, 32, 512, 0, *tail)
    assert lead(None, None, None, one, None, None) == -1 and b"next_planes without next_wpack" in lib.vocr_last_error()
    assert lead(one, None, None, None, None, None) == -1 and b"next_wpack without next_planes" in lib.vocr_last_error()
    assert lib.vocr_lstm_fwd_lead(one, None, one, one, one, one, one, one, one, 8, 8, 512, 0, one, None, None, one, None, None) == -1
    assert lib.vocr_dropout_mask(one, 16, 1.0, 1, None) == -1 and lib.vocr_dropout_mask(None, 16, 0.5, 1, None) == -1
    assert lib.vocr_gather_rows_fill_grad_workspace_bytes(96) == 64 * 96 * 4
    assert lib.vocr_gather_rows_fill_grad(one, None, 4, 8, one, one, None) == -1
    assert lib.vocr_conv3x3_h16_plan(32, 64, 30, 600, 64) == 1 and lib.vocr_conv3x3_h16_plan(32, 24, 30, 600, 64) == 0
    assert lib.vocr_conv3x3_h16_plan(32, 128, 15, 420, 128) in (2, 3, 4, 5)
    # bf16x6 GEMM: plane sizes (rows padded to 256, K to 32; three planes of bf16) and argument validation
    assert lib.vocr_gemm_x6_planes_bytes(9408, 1024) == 3 * (37 * 8) * (32 * 2) * 1024 and lib.vocr_gemm_x6_planes_bytes(0, 8) == 0
    assert lib.vocr_gemm_x6_workspace_bytes(4096, 1024, 9408) > 0
    assert lib.vocr_gemm_x6_split(one, None, 8, 0, None, 64, 32, 64, 1, one, None) == -1 and b"second piece" in lib.vocr_last_error()
    assert lib.vocr_gemm_x6_split(one, None, 0, 0, one, 64, 32, 64, 0, one, None) == -1          # a mask only with a K-contiguous source
    x6 = lambda *a: lib.vocr_gemm_x6(one, 256, 64, 0, 0, one, 128, 64, 0, 0, *a)
    assert x6(256, 128, 40, one, None, 0, 0, 128, None, None, 0, None, None) == -1 and b"k % 16" in lib.vocr_last_error()
    assert x6(256, 128, 64, one, None, 64, 0, 128, None, None, 0, None, None) == -1               # a cut without a second output
    assert lib.vocr_gemm_x6(one, 256, 64, 0, 2, one, 128, 64, 0, 0, 256, 128, 64, one, None, 0, 0, 128, None, None, 0, None, None) == -1   # view leaves its planes
    tv = lambda fn, *a: fn(one, 512, 64, 0, one, 128, 64, 512, 128, 64, *a)
    assert tv(lib.vocr_gemm_x6_two_views, 100, 0, 0, 0, 0, 0, 0, one, one, 128, None, None) == -1 and b"256-row tile" in lib.vocr_last_error()
    assert tv(lib.vocr_gemm_h3_two_views, 256, 0, 0, 0, 9, 0, 0, one, one, 128, None, None) == -1 and b"second view" in lib.vocr_last_error()
    # the opt-in fp16x3 split: two planes + the rows' maxima (fp32) behind them; the same validation in front of any launch
    assert lib.vocr_gemm_h3_planes_bytes(9408, 1024) == 2 * (37 * 8) * (32 * 2) * 1024 + 37 * 8 * 32 * 4 and lib.vocr_gemm_h3_planes_bytes(0, 8) == 0
    assert lib.vocr_gemm_h3_split(one, None, 8, 0, None, 64, 32, 64, 1, 0.0, one, None) == -1 and b"second piece" in lib.vocr_last_error()
    assert lib.vocr_gemm_h3_split(one, None, 0, 0, None, 64, 32, 64, 1, -1.0, one, None) == -1 and b"bound" in lib.vocr_last_error()
    assert lib.vocr_gemm_h3(one, 256, 64, 0, 0, one, 128, 64, 0, 0, 256, 128, 40, one, None, 0, 0, 128, None, None, 0, None, None) == -1 and b"vocr_gemm_h3" in lib.vocr_last_error()


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
