"""Diagnostic (GPU box): per-parameter gradient error of the HIP path vs the on-box oracle for golden cases."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests import golden_util as gu
from tests.test_model_golden_gpu import _build
import vistaocr_amd as va
from oracle import vista_oracle as vo

for name in sys.argv[1:] or ["c1", "varwidth_train"]:
    g, hp, al, model, x, w, tgt, tl = _build(name, "english")
    _, _, _, sd_np, _, _, _, _, (s1, s2) = gu.case_inputs(name)
    logits, lens = model(torch.from_numpy(x), torch.from_numpy(w))
    loss = va.CTCLoss()(logits, torch.from_numpy(tgt), lens, torch.from_numpy(tl))
    loss.backward()
    sd = vo.state_from_numpy(sd_np)
    taps = {}
    lo, ln = vo.forward(sd, hp, torch.from_numpy(x), w, (torch.from_numpy(s1), torch.from_numpy(s2)), training=True, lstm_training=False, taps=taps)
    lo.retain_grad()
    lloss = vo.ctc_criterion(lo, torch.from_numpy(tgt), ln, torch.from_numpy(tl))
    lloss.backward()
    print(name, "loss", float(loss), float(lloss), "max logit err", float((logits.cpu() - lo).abs().max()))
    for k, p in model.named_parameters():
        r = sd[k].grad
        d = (p.grad.cpu().double() - r.double())
        print("  %-28s |g| %.4e  rel_l2 %.3e  max_abs %.3e (max ref %.3e)" % (k, float(r.double().norm()), float(d.norm() / (r.double().norm() + 1e-30)), float(d.abs().max()), float(r.abs().max())))
    # float64 truth: is the HIP path as close to exact arithmetic as the fp32 PyTorch-CPU reference path is?
    sd64 = {k: (v.detach().double().requires_grad_(v.requires_grad)) for k, v in vo.state_from_numpy(sd_np).items()}
    l64, n64 = vo.forward(sd64, hp, torch.from_numpy(x).double(), w, (torch.from_numpy(s1).double(), torch.from_numpy(s2).double()), training=True, lstm_training=False)
    vo.ctc_criterion(l64, torch.from_numpy(tgt), n64, torch.from_numpy(tl)).backward()
    print("  vs float64 truth:   hip_err      ref32_err")
    for k, p in model.named_parameters():
        t = sd64[k].grad
        eh = float((p.grad.cpu().double() - t).norm() / (t.norm() + 1e-30))
        er = float((sd[k].grad.double() - t).norm() / (t.norm() + 1e-30))
        print("  %-28s %.3e   %.3e" % (k, eh, er))
