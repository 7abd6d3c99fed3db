"""GPU box: inference throughput of the BASELINE model (forward + greedy decode to label sequences), batch 32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vistaocr_amd as va
import bench
al = va.english_alphabet()
torch.manual_seed(0)
model = va.CnnOcrModel(alphabet=al, verbose=False, **bench.HP)
model.eval()
x, tgt, widths, tl = bench.make_batch(0, len(al))
x = x.cuda()
dec = va.ArgmaxDecoder(al)
with torch.no_grad():
    for _ in range(3):
        logits, lens = model(x, widths); hyp = dec.decode(logits, lens, uxxxx=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        logits, lens = model(x, widths)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(20):
        logits, lens = model(x, widths); hyp = dec.decode(logits, lens, uxxxx=True)
    torch.cuda.synchronize(); t2 = time.perf_counter()
print("forward only: %.2f ms/batch, %.0f line-images/s;  forward + greedy decode to strings: %.2f ms/batch, %.0f line-images/s"
      % ((t1 - t0) / 20 * 1e3, 32 * 20 / (t1 - t0), (t2 - t1) / 20 * 1e3, 32 * 20 / (t2 - t1)))
