"""CPU: the plain-C restatement of the published CTC algorithm (oracle/ctc_ref.c, standing in for the absent
warp-ctc dependency) agrees with torch.nn.functional.ctc_loss — the PyTorch-CPU criterion of the parity target."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cref():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libctc_ref.so"))
    lib.ctc_ref.restype = ctypes.c_int
    return lib


@pytest.mark.parametrize("T,B,V,L", [(12, 3, 7, [3, 1, 0]), (50, 4, 96, [10, 7, 12, 1]), (30, 2, 166, [14, 2])])
def test_c_ctc_matches_torch(cref, T, B, V, L):
    g = torch.Generator().manual_seed(1)
    logits = (torch.rand(T, B, V, generator=g) * 6 - 3)
    act = [T - 2 * i for i in range(B)]
    tl = torch.tensor(L, dtype=torch.int32)
    tg = torch.randint(1, V, (max(1, int(tl.sum())),), generator=g).to(torch.int32)
    if tg.numel() > 2:
        tg[1] = tg[0]                           # repeated label
    lr = logits.clone().double().requires_grad_(True)
    ref = F.ctc_loss(F.log_softmax(lr, 2), tg[: int(tl.sum())].long(), torch.tensor(act), tl.long(), blank=0, reduction="none")
    ref.sum().backward()
    x = np.ascontiguousarray(logits.numpy())
    nll = np.zeros(B, dtype=np.float64)
    grad = np.zeros((T, B, V), dtype=np.float64)
    lab = np.ascontiguousarray(tg.numpy())
    rc = cref.ctc_ref(x.ctypes.data_as(ctypes.c_void_p), lab.ctypes.data_as(ctypes.c_void_p),
                      np.asarray(L, dtype=np.int32).ctypes.data_as(ctypes.c_void_p),
                      np.asarray(act, dtype=np.int32).ctypes.data_as(ctypes.c_void_p), T, B, V,
                      nll.ctypes.data_as(ctypes.c_void_p), grad.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    np.testing.assert_allclose(nll, ref.detach().numpy(), rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(grad, lr.grad.numpy(), rtol=1e-8, atol=1e-10)
