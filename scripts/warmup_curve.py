"""GPU box: per-step wall time of the first steps of a fresh process (how long until train() reaches its steady state)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vistaocr_amd as va
import bench
al = va.english_alphabet()
torch.manual_seed(0)
model = va.CnnOcrModel(alphabet=al, verbose=False, **bench.HP)
crit = va.CTCLoss()
x, tgt, widths, tl = bench.make_batch(0, len(al))
x = x.pin_memory()
model.train()
opt = va.make_optimizer(model)
import gc
MODE = os.environ.get("MODE", "sync")
if MODE == "nogc":
    gc.disable()
ts = []
for i in range(160):
    if MODE == "sync":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    va.train((x, tgt, widths, tl, {}), model, crit, opt)
    if MODE == "sync":
        torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("per-step ms (MODE=%s: sync = synchronised after every step, else call-to-call as bench.py runs them):" % MODE)
for a in range(0, 160, 10):
    print("  steps %3d-%3d: %s" % (a, a + 9, " ".join("%.2f" % t for t in ts[a:a + 10])))
print("gc counts", gc.get_count(), "mean of steps 40..159: %.3f" % (sum(ts[40:]) / len(ts[40:])))
