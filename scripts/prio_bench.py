"""GPU box: does it pay to run the step on a HIGH-priority stream (the side stream with the weight-gradient kernels is created with the
lowest priority, but torch's default stream has that priority too)?"""
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
print("priority range", torch.cuda.Stream.priority_range())
lo, hi = torch.cuda.Stream.priority_range()
def run(name, stream):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(10): va.train(batch, model, crit, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(150): va.train(batch, model, crit, opt)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 150
    print("%-40s %.2f ms/step  %.0f line-images/s" % (name, dt * 1e3, 32 / dt))
run("default stream", None)
run("high-priority stream (%d)" % hi, torch.cuda.Stream(priority=hi))
run("default stream", None)
run("high-priority stream (%d)" % hi, torch.cuda.Stream(priority=hi))
