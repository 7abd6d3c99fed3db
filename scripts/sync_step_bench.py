"""GPU box: throughput of the reference-style train() (returns the loss as a Python float: one device sync per step)
next to train_async() (what bench.py times)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vistaocr_amd as va
import bench
al = va.english_alphabet()
torch.manual_seed(0)
model = va.CnnOcrModel(alphabet=al, verbose=False, **bench.HP)
model.train()
opt = va.make_optimizer(model, lr=1e-3)
crit = va.CTCLoss()
x, tgt, widths, tl = bench.make_batch(0, len(al))
batch = (x.cuda(), tgt, widths, tl, {})
for name, fn in (("train_async (no per-step sync)", va.train_async), ("train (returns the loss float)", va.train)):
    for _ in range(5): fn(batch, model, crit, opt)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    per = []
    for _ in range(200):
        a = time.perf_counter(); fn(batch, model, crit, opt); per.append(time.perf_counter() - a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
    print("%-34s %.2f ms/step  %.0f line-images/s   host ms per call: min %.1f median %.1f max %.1f" % (name, dt * 1e3, 32 / dt, min(per) * 1e3, sorted(per)[100] * 1e3, max(per) * 1e3))
