"""Round 6 probe (GPU box): the next layer's x-projection inside a forward sweep (vocr_lstm_fwd_lead with next_wpack: four follower
waves per workgroup).  Checks the planes against a torch fp64 product of the sweep's own y and the sweep's outputs against the plain
sweep bit for bit, and times: sweep alone, x-projection GEMM alone, sweep then GEMM (what the step did), sweep with followers.
usage: python scripts/follow_ab.py [T] [B] [masked 0|1] [ragged 0|1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops, _lib
from vistaocr_amd._lib import call
T = int(sys.argv[1]) if len(sys.argv) > 1 else 294
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
MASKED = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
RAGGED = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
H = 512; G = 4 * H; R = T * B
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator().manual_seed(0)
rnd = lambda *s, a=1.0: ((torch.rand(*s, generator=g) - 0.5) * a).to(dev)
xproj = rnd(2, R, G, a=0.6)
wf, wr = rnd(G, H, a=0.2), rnd(G, H, a=0.2)
nwf, nwr = rnd(G, 2 * H, a=0.16), rnd(G, 2 * H, a=0.16)          # the next layer's W_ih
nb = rnd(2, G, a=0.1)
lens = torch.tensor(sorted([max(1, T - 3 * i) if RAGGED else T for i in range(B)], reverse=True), dtype=torch.int32, device=dev)
mk = lambda: (torch.empty(R, 2 * H, device=dev), torch.empty(2, R, G, device=dev), torch.empty(2, R, H, device=dev))
y, gates, cell = mk(); y0, gates0, cell0 = mk()
ws = torch.zeros(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
health = torch.zeros(4, dtype=torch.int32, device=dev)
mask = None
if MASKED:
    mask = torch.empty(R, 2 * H, device=dev)
    call("vocr_dropout_mask", mask.data_ptr(), mask.numel(), 0.5, 1234, torch.cuda.current_stream().cuda_stream)
wpack = ops.lstm_xproj_pack(nwf, nwr)
planes = torch.full((2, 2, R, G), float("nan"), device=dev)
main = torch.cuda.current_stream()
P = lambda t: t.data_ptr() if t is not None else None

def plain():
    call("vocr_lstm_fwd", P(xproj), P(wf), P(wr), P(lens), P(y0), P(gates0), P(cell0), P(ws), T, B, H, P(health), main.cuda_stream)

def lead(follow=True):
    call("vocr_lstm_fwd_lead", P(xproj), None, P(wf), P(wr), P(lens), P(y), P(gates), P(cell), P(ws), T, B, H, 0,
         P(wpack) if follow else None, P(nb) if follow else None, P(mask) if follow else None, P(planes) if follow else None, P(health), main.cuda_stream)

xo = torch.empty(2, R, G, device=dev)
ym = torch.empty_like(y)
def gemm():
    src = y
    if mask is not None:
        call("vocr_mul", P(y), P(mask), P(ym), y.numel(), main.cuda_stream); src = ym
    ops.gemm_pair(0, 0, 1, R, G, 2 * H, src, src, 2 * H, nwf, nwr, 2 * H, xo[0], xo[1], G, bias0=nb[0], bias1=nb[1])

plain(); lead(); torch.cuda.synchronize()
print("health", health.tolist()[:2], " sweep outputs identical to the plain sweep:", bool(torch.equal(y, y0) and torch.equal(gates, gates0) and torch.equal(cell, cell0)))
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
keep = planes.clone(); planes.fill_(float("nan")); lead(); torch.cuda.synchronize()
print("second run bit-identical:", bool(torch.equal(keep, planes)))

if os.environ.get("FOLLOW_STAMPS"):          # -DVOCR_FOLLOW_PROBE builds: 100-MHz stamps of chains 0 and 1 (member 0)
    w = ws.view(torch.int64)[:128].cpu().tolist()
    for c in (0, 1):
        t0 = w[c * 64 + 28]
        print("chain %d: loop %.0f us; follower done %.0f us after the loop's start; unit u began at (us) / with joined =" % (c, (w[c * 64 + 29] - t0) / 100.0, (w[c * 64 + 30] - t0) / 100.0),
              " ".join("%.0f/%d" % ((w[c * 64 + u] - t0) / 100.0, w[c * 64 + 32 + u]) for u in range((T + 15) // 16)))
if os.environ.get("FOLLOW_PHASES"):          # -DVOCR_LSTM_STAMPS -DVOCR_FOLLOW_PROBE builds: phase ticks of workgroup 0, waves 0 and 7
    names = ["poll", "mfma+lds", "barrier1", "reduce+act", "barrier2", "cell+stores", "-", "loop top"]
    for label, fn in (("without followers", lambda: lead(False)), ("WITH followers", lead)):
        fn(); torch.cuda.synchronize()
        w = ws.view(torch.int64)[256:272].cpu().tolist()
        for wv in (0, 1):
            print("%s, wave %d: ticks per step: " % (label, 7 * wv) + "  ".join("%s %.0f" % (names[k], w[wv * 8 + k] / T) for k in (7, 0, 1, 2, 3, 4, 5)) + "   sum %.0f" % (sum(w[wv * 8: wv * 8 + 8]) / T))
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t_p = timed(plain); t_s = timed(lambda: lead(False)); t_g = timed(gemm); t_sg = timed(lambda: (lead(False), gemm())); t_b = timed(lead)
print("T=%d B=%d masked=%d ragged=%d: plain sweep %.0f us (%.2f/step) | lead without followers %.0f | gemm(+mul) %.0f | sweep then gemm %.0f | sweep WITH followers %.0f | health %s"
      % (T, B, MASKED, RAGGED, t_p, t_p / T, t_s, t_g, t_sg, t_b, health.tolist()[:2]))
