"""GPU box: is the train step CPU(enqueue)-bound?  Compare host enqueue time per step with the GPU-synchronised time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import vistaocr_amd as va
al = va.english_alphabet()
torch.manual_seed(0)
model = va.CnnOcrModel(alphabet=al, verbose=False, **bench.HP); model.train()
opt = va.make_optimizer(model); crit = va.CTCLoss()
x, tgt, widths, tl = bench.make_batch(0, len(al)); batch = (x.cuda(), tgt, widths, tl, {})
for _ in range(3): va.train_async(batch, model, crit, opt)
torch.cuda.synchronize()
enq = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter(); va.train_async(batch, model, crit, opt); enq.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue per step: %.2f ms (min %.2f)   wall per step incl. final sync: %.2f ms   drain after last enqueue: %.2f ms" % (1e3 * sum(enq) / 10, 1e3 * min(enq), 1e3 * (t2 - t0) / 10, 1e3 * (t2 - t1)))
# phase split of the enqueue time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): va.train_async(batch, model, crit, opt)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
