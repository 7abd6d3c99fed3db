"""A/B of the LSTM sweep hand-off protocols (GPU box): per setting a fresh process (the switches are read once), µs per step, the
status word and a digest of y / gates / cell / dgates so that settings can be compared bit for bit.
usage: python scripts/lstm_ab.py ["ENV=V ENV2=V" ...]"""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BODY = r'''
import os, sys, time, hashlib
sys.path.insert(0, %r)
import torch
from vistaocr_amd import ops, _lib
from vistaocr_amd._lib import call
T, B, H = int(os.environ.get("T", "294")), int(os.environ.get("B", "32")), int(os.environ.get("H", "512"))
dev = torch.device("cuda:0"); lib = _lib.load()
g = torch.Generator().manual_seed(0)
xproj = ((torch.rand(2, T * B, 4 * H, generator=g) - 0.5) * 0.6).to(dev)
wf = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.2).to(dev); wr = ((torch.rand(4 * H, H, generator=g) - 0.5) * 0.2).to(dev)
lens = torch.tensor(sorted([max(1, T - 3 * i) for i in range(B)], reverse=True), dtype=torch.int32, device=dev)
y = torch.empty(T * B, 2 * H, device=dev); gates = torch.empty(2, T * B, 4 * H, device=dev); cell = torch.empty(2, T * B, H, device=dev)
ws = torch.zeros(lib.vocr_lstm_workspace_bytes(T, B, H) // 4 + 16, device=dev)
dy = ((torch.rand(T * B, 2 * H, generator=g) - 0.5) * 0.02).to(dev); dg = torch.empty(2, T * B, 4 * H, device=dev)
db = torch.empty(2, 4 * H, device=dev)
health = torch.zeros(4, dtype=torch.int32, device=dev)
wtf, wtr = ops.transpose2d(wf), ops.transpose2d(wr)
s = torch.cuda.current_stream().cuda_stream
def fwd(): call("vocr_lstm_fwd", xproj.data_ptr(), wf.data_ptr(), wr.data_ptr(), lens.data_ptr(), y.data_ptr(), gates.data_ptr(), cell.data_ptr(), ws.data_ptr(), T, B, H, health.data_ptr(), s)
def bwd(): call("vocr_lstm_bwd_bias", dy.data_ptr(), wtf.data_ptr(), wtr.data_ptr(), lens.data_ptr(), gates.data_ptr(), cell.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), T, B, H, health.data_ptr(), s)
out = []
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize(); dt = e0.elapsed_time(e1) / 10
    out.append("%%s %%.1f us (%%.2f/step)" %% (name, dt * 1e3, dt * 1e3 / T))
dig = lambda *ts: hashlib.sha1(b"".join(t.cpu().numpy().tobytes() for t in ts)).hexdigest()[:10]
print("  ".join(out), " health", health.tolist()[:2], " fwd", dig(y, gates, cell), " bwd", dig(dg, db), " nan", bool(torch.isnan(y).any() or torch.isnan(dg).any()))
''' % ROOT
settings = sys.argv[1:] or [""]
for st in settings:
    env = dict(os.environ)
    for kv in st.split():
        k, v = kv.split("=", 1); env[k] = v
    try:
        r = subprocess.run([sys.executable, "-c", BODY], env=env, capture_output=True, text=True, timeout=300)
        print("[%s] %s" % (st, (r.stdout.strip() or r.stderr.strip()[-400:])), flush=True)
    except subprocess.TimeoutExpired:
        print("[%s] TIMEOUT" % st, flush=True)
