"""GPU box diagnostic: where do torch.optim.Adam + device clamp and the fused FlatClampAdam differ after two steps?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import closed_form as cf
from tests.test_round2_gpu import _small_setup
va, model_a, batch = _small_setup(seed=4)
_, model_b, _ = _small_setup(seed=4)
opt_a = torch.optim.Adam(model_a.parameters(), lr=1e-3)
opt_b = va.make_optimizer(model_b, lr=1e-3)
crit = va.CTCLoss()
for step in range(2):
    s1, s2 = cf.closed_form_pool_samples(batch[0].shape[0], seed=5 + step)
    for m in (model_a, model_b):
        m.pool_samples = [torch.from_numpy(s1), torch.from_numpy(s2)]
        m._dropout_calls = 10 * step
    la = va.train(batch, model_a, crit, opt_a)
    ga = {k: p.grad.detach().clone() for k, p in model_a.named_parameters()}
    lb = va.train(batch, model_b, crit, opt_b)
    gb = {k: p.grad.detach().clone() for k, p in model_b.named_parameters()}
    print("step", step, la, lb)
    for k in ga:
        d = (ga[k] - gb[k]).abs()
        if float(d.max()) > 0:
            print("   grad differs: %-28s max %.3e (scale %.3e) n=%d of %d" % (k, float(d.max()), float(gb[k].abs().max()), int((d > 0).sum()), d.numel()))
for (k, pa), (_, pb) in zip(model_a.named_parameters(), model_b.named_parameters()):
    d = (pa.detach() - pb.detach()).abs()
    nb = int((d > 2e-6).sum())
    if nb:
        print("%-28s n_bad %7d of %8d  max %.3e" % (k, nb, d.numel(), float(d.max())))
