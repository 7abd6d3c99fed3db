"""Round 6 probe (GPU box): the next layer's x-projection behind a forward sweep (vocr_lstm_fwd_lead + vocr_lstm_xproj_follow).
Checks the follower's planes against a torch fp64 product of the sweep's own y, and times: sweep alone, x-projection GEMM alone,
sweep then GEMM (what the step did), sweep with the follower beside it (two streams, done = both ended).
usage: python scripts/follow_ab.py [T] [B] [masked 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops, _lib
from vistaocr_amd._lib import call
T = int(sys.argv[1]) if len(sys.argv) > 1 else 294
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
MASKED = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
H = 512; G = 4 * H; R = T * B
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator().manual_seed(0)
rnd = lambda *s, a=1.0: ((torch.rand(*s, generator=g) - 0.5) * a).to(dev)
xproj = rnd(2, R, G, a=0.6)
wf, wr = rnd(G, H, a=0.2), rnd(G, H, a=0.2)
nwf, nwr = rnd(G, 2 * H, a=0.16), rnd(G, 2 * H, a=0.16)          # the next layer's W_ih
nb = rnd(2, G, a=0.1)
lens = torch.tensor(sorted([max(1, T - 3 * i) for i in range(B)], reverse=True), dtype=torch.int32, device=dev)
y = torch.empty(R, 2 * H, device=dev); gates = torch.empty(2, R, G, device=dev); cell = torch.empty(2, R, H, device=dev)
ws = torch.zeros(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
health = torch.zeros(4, dtype=torch.int32, device=dev)
mask = None
if MASKED:
    mask = torch.empty(R, 2 * H, device=dev)
    call("vocr_dropout_mask", mask.data_ptr(), mask.numel(), 0.5, 1234, torch.cuda.current_stream().cuda_stream)
wpack = ops.lstm_xproj_pack(nwf, nwr)
planes = torch.full((2, 2, R, G), float("nan"), device=dev)
main = torch.cuda.current_stream(); fs = ops.follow_stream(dev)
if os.environ.get("FOLLOW_FS_PRIO"):          # probe: the follower's stream at another queue priority
    lo_, hi_ = torch.cuda.Stream.priority_range()
    fs = torch.cuda.Stream(priority={"lo": lo_, "hi": hi_, "none": 0}[os.environ["FOLLOW_FS_PRIO"]])
import time
P = lambda t: t.data_ptr() if t is not None else None

def sweep(epoch=0):
    call("vocr_lstm_fwd_lead", P(xproj), None, P(wf), P(wr), P(lens), P(y), P(gates), P(cell), P(ws), T, B, H, 0, epoch, P(health), main.cuda_stream)

def follower(epoch, stream):
    call("vocr_lstm_xproj_follow", P(y), P(mask), P(wpack), P(nb), P(planes), P(lens), P(ws), T, B, H, 0, epoch, P(health), stream.cuda_stream)

def both():
    e = ops.next_epoch()
    fs.wait_stream(main)
    sweep(e)
    follower(e, fs)
    main.wait_stream(fs)

xo = torch.empty(2, R, G, device=dev)
ym = torch.empty_like(y)
def gemm():
    src = y
    if mask is not None:
        call("vocr_mul", P(y), P(mask), P(ym), y.numel(), main.cuda_stream); src = ym
    ops.gemm_pair(0, 0, 1, R, G, 2 * H, src, src, 2 * H, nwf, nwr, 2 * H, xo[0], xo[1], G, bias0=nb[0], bias1=nb[1])

# ---- numerics: follower beside the sweep vs fp64 on the sweep's y
both(); torch.cuda.synchronize()
print("health", health.tolist()[:2])
yy = (y * mask if mask is not None else y).double()
W = torch.cat([nwf, nwr], 0).double()                  # [8H][2H]
ref = [yy[:, s * H:(s + 1) * H] @ W[:, s * H:(s + 1) * H].T for s in (0, 1)]          # [R][8H] per source direction
ref[0] = ref[0] + nb.reshape(-1).double()
got = [torch.cat([planes[s, 0], planes[s, 1]], 1).double() for s in (0, 1)]
for s in (0, 1):
    d = (got[s] - ref[s]).abs().max().item()
    print("plane src=%d max|d| %.3e (ref max %.3f) nan %s" % (s, d, ref[s].abs().max().item(), bool(torch.isnan(got[s]).any())))
gemm(); torch.cuda.synchronize()
tot = torch.cat([xo[0], xo[1]], 1).double()
print("sum of planes vs the GEMM path: max|d| %.3e" % ((got[0] + got[1]) - tot).abs().max().item())
# the follower after the sweep has ended (a plain GEMM then): same planes bit for bit?
keep = planes.clone(); planes.fill_(float("nan"))
e = ops.next_epoch(); sweep(e); follower(e, main); torch.cuda.synchronize()
print("late follower bit-identical:", bool(torch.equal(keep, planes)))

def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def split():          # the two kernels' own durations when they run side by side
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda.synchronize()
    e = ops.next_epoch(); fs.wait_stream(main)
    order = os.environ.get("FOLLOW_ORDER", "same")
    if order == "early":          # the follower's waves are the OLDER ones (it sleeps at its gate until the sweep publishes progress)
        ev[2].record(fs); follower(e, fs); ev[3].record(fs); time.sleep(0.0002)
        ev[0].record(main); sweep(e); ev[1].record(main)
    else:
        ev[0].record(main); sweep(e); ev[1].record(main)
        if order == "late": time.sleep(0.0001)
        ev[2].record(fs); follower(e, fs); ev[3].record(fs)
    main.wait_stream(fs); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3, ev[2].elapsed_time(ev[3]) * 1e3, ev[0].elapsed_time(ev[3]) * 1e3
def sweep_clock():          # probe builds (-DVOCR_FOLLOW_PROBE): shader cycles and 100-MHz ticks of workgroup 0's sweep loop
    o = 6 * 2 * B * H * 4 + 4096 + (16 << 20) + 16 * 4 * H * 4 + 64 * 4 * H * 4
    o = (o + 255) // 256 * 256
    w = ws.view(torch.int32)[o // 4 + 900: o // 4 + 904].cpu().tolist()
    cyc, rt = (w[0] & 0xFFFFFFFF) | (w[1] << 32), (w[2] & 0xFFFFFFFF) | (w[3] << 32)
    return cyc, rt, (cyc / rt * 100 if rt > 0 else 0.0)
sweep(0); torch.cuda.synchronize()
print("sweep alone: loop %d cycles, %d ticks -> %.0f MHz" % sweep_clock())
for _ in range(3): sp = split()
if os.environ.get("FOLLOW_GATES"):
    o = 6 * 2 * B * H * 4 + 4096 + (16 << 20) + 16 * 4 * H * 4 + 64 * 4 * H * 4
    o = (o + 255) // 256 * 256
    w64 = ws.view(torch.int64)
    st = w64[(o // 4 + 900) // 2: (o // 4 + 900) // 2 + 4].cpu().tolist()
    print("sweep wg0: start tick %d, loop %d ticks (%.0f us), xcc %d" % (st[2], st[1], st[1] / 100.0, st[3]))
    for blk in (0, 1):
        g_ = w64[(o // 4 + 1100) // 2 + 32 * blk: (o // 4 + 1100) // 2 + 32 * blk + 31].cpu().tolist()
        print("follower wg%d xcc %d: gate of unit u opened (us after the sweep's start):" % (100 * blk, g_[30]), " ".join("%.0f" % ((v - st[2]) / 100.0) for v in g_[:19]))
print("sweep beside the follower: loop %d cycles, %d ticks -> %.0f MHz" % sweep_clock())
print("side by side: sweep %.0f us, follower %.0f us, both done after %.0f us   [VARIANT=%s DUTY=%s NOPRIO=%s NAP=%s PRIO=%s ORDER=%s]"
      % (sp + tuple(os.environ.get(k, "-") for k in ("VOCR_FOLLOW_VARIANT", "VOCR_FOLLOW_DUTY", "VOCR_FOLLOW_NOPRIO", "VOCR_FOLLOW_NAP_US", "FOLLOW_FS_PRIO", "FOLLOW_ORDER"))))
if os.environ.get("FOLLOW_QUICK"): sys.exit(0)
t_s = timed(lambda: sweep(0)); t_g = timed(gemm); t_sg = timed(lambda: (sweep(0), gemm())); t_b = timed(both)
def late():
    e = ops.next_epoch(); sweep(e); follower(e, main)
t_l = timed(late)
print("T=%d B=%d masked=%d: sweep %.0f us (%.2f/step) | gemm(+mul) %.0f | sweep then gemm %.0f | sweep + follower %.0f | sweep then follower (serial) %.0f | health %s"
      % (T, B, MASKED, t_s, t_s / T, t_g, t_sg, t_b, t_l, health.tolist()[:2]))
