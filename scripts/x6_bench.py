"""Probe (GPU box): the bf16x6 GEMM (vocr_gemm_x6: fp32 operands split exactly into three bf16 planes, six bf16 MFMAs per product) against the
f32-MFMA panel GEMM (vocr_gemm_pair) on the step's LSTM shapes: max error against fp64 of both, time of the split passes and of the product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vistaocr_amd import ops, _lib
from vistaocr_amd._lib import call
dev = torch.device("cuda:0"); lib = _lib.load()
SCH = os.environ.get("X6_SCHEME", "bf16x6")            # or fp16x3: the opt-in split (vocr_gemm_h3*)
NB, SPLIT, GEMM = ops._X6_ENTRY[SCH][:3]
print("split:", SCH)
s = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr() if t is not None else None
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def planes(x, rows, k, kc, ld=None, x2=None, seg=0, axis=0, mask=None, bound=0.0):
    buf = torch.empty(getattr(lib, NB)(rows, k) // 2, dtype=torch.int16, device=dev)
    if SCH == "fp16x3": fn = lambda: call(SPLIT, P(x), P(x2), seg, axis, P(mask), ld if ld else x.stride(0), rows, k, int(kc), float(bound), P(buf), s)
    else: fn = lambda: call(SPLIT, P(x), P(x2), seg, axis, P(mask), ld if ld else x.stride(0), rows, k, int(kc), P(buf), s)
    fn()
    return buf, fn
def x6gemm(a, a_rows, a_k, a_row0, a_kk0, b_, b_rows, b_k, b_row0, b_kk0, m, n, k, c0, c1=None, csplit=0, rsplit=0, ldc=None, bias0=None, bias1=None, ws_=None):
    call(GEMM, P(a), a_rows, a_k, a_row0, a_kk0, P(b_), b_rows, b_k, b_row0, b_kk0, m, n, k, P(c0), P(c1), csplit, rsplit, ldc, P(bias0), P(bias1), 0, P(ws_), s)
g = torch.Generator().manual_seed(0)
R, G, D = 9408, 2048, 1024
x = ((torch.rand(R, D, generator=g) - 0.5) * 2).to(dev)
wf = ((torch.rand(G, D, generator=g) - 0.5) * 0.16).to(dev); wr = ((torch.rand(G, D, generator=g) - 0.5) * 0.16).to(dev)
b = ((torch.rand(2, G, generator=g) - 0.5) * 0.1).to(dev)
# ---- x-projection: C = x [R][D] . [wf; wr]^T
w = torch.cat([wf, wr], 0).contiguous()                               # for the fp64 reference; the planes take the two matrices
xa, fx = planes(x, R, D, True, bound=1.0); wb, fw = planes(wf, 2 * G, D, True, x2=wr, seg=G, axis=1)
out = torch.empty(2, R, G, device=dev)
ws = torch.empty(lib.vocr_gemm_x6_workspace_bytes(2 * G, D, R) // 4 + 16, device=dev)
def x6(): x6gemm(xa, R, D, 0, 0, wb, 2 * G, D, 0, 0, R, 2 * G, D, out[0], out[1], csplit=G, ldc=G, bias0=b[0], bias1=b[1], ws_=ws)
ref = torch.empty(2, R, G, device=dev)
def f32(): ops.gemm_pair(0, 0, 1, R, G, D, x, x, D, wf, wr, D, ref[0], ref[1], G, bias0=b[0], bias1=b[1])
x6(); f32(); torch.cuda.synchronize()
ex = (x[:512].double() @ w.double().T + b.reshape(-1).double())
got = torch.cat([out[0, :512], out[1, :512]], 1).double(); rf = torch.cat([ref[0, :512], ref[1, :512]], 1).double()
print("x-projection 9408 x 4096 x 1024: max|err| vs fp64: split %.3e, f32 MFMA %.3e (|C| max %.2f); nan %s" % ((got - ex).abs().max(), (rf - ex).abs().max(), ex.abs().max(), bool(torch.isnan(out).any())))
tx, tw, tp, tf = timed(fx), timed(fw), timed(x6), timed(f32)
fl = 2.0 * R * 2 * G * D
print("   split x %.0f us, split W %.0f us, product %.0f us = %.0f TFLOP/s fp32-equivalent | f32 MFMA pair %.0f us = %.0f TFLOP/s" % (tx, tw, tp, fl / tp / 1e6, tf, fl / tf / 1e6))
# ---- data gradient: dx = dg_f . W_f + dg_r . W_r : [R][2G] . [2G][D] -> B operand = W^T, K-strided source [k = gate col][n = D]
dg = ((torch.rand(2, R, G, generator=g) - 0.5) * 0.02).to(dev)
dgc = torch.cat([dg[0], dg[1]], 1).contiguous()                      # [R][2G]: only for the fp64 reference; the planes take the two pieces
da, fda = planes(dg[0], R, 2 * G, True, ld=G, x2=dg[1], seg=G, axis=0)
wt, fwt = planes(wf, D, 2 * G, False, ld=D, x2=wr, seg=G, axis=0)    # rows = D (n), k = 2G: source [k][n], the two matrices along K
dx = torch.empty(R, D, device=dev); dxr = torch.empty(R, D, device=dev)
def x6d(): x6gemm(da, R, 2 * G, 0, 0, wt, D, 2 * G, 0, 0, R, D, 2 * G, dx, ldc=D, ws_=ws)
def f32d(): ops.gemm_pair(1, 0, 0, R, D, G, dg[0], dg[1], G, wf, wr, D, dxr, None, D)
x6d(); f32d(); torch.cuda.synchronize()
ex = dgc[:512].double() @ w.double()
print("data gradient 9408 x 1024 x 4096: max|err| vs fp64: split %.3e, f32 MFMA %.3e (|C| max %.3f)" % ((dx[:512].double() - ex).abs().max(), (dxr[:512].double() - ex).abs().max(), ex.abs().max()))
ta, tb_, tp, tf = timed(fda), timed(fwt), timed(x6d), timed(f32d)
fl = 2.0 * R * D * 2 * G
print("   split dg %.0f us, split W^T %.0f us, product %.0f us = %.0f TFLOP/s | f32 MFMA pair %.0f us = %.0f TFLOP/s" % (ta, tb_, tp, fl / tp / 1e6, tf, fl / tf / 1e6))
# ---- weight gradient: dW = dg^T . x : [G][R] . [R][D]: both operands K-strided sources (k = the row index R)
dgt, fdgt = planes(dg[0], 2 * G, R, False, ld=G, x2=dg[1], seg=G, axis=1)    # rows = 2G (m: gate columns of both directions), k = R
xt, fxt = planes(x, D, R, False, ld=D, bound=1.0)                               # rows = D (n), k = R
dw = torch.empty(2, G, D, device=dev); dwr = torch.empty(2, G, D, device=dev)
def x6w(): x6gemm(dgt, 2 * G, R, 0, 0, xt, D, R, 0, 0, 2 * G, D, R, dw[0], dw[1], rsplit=G, ldc=D, ws_=ws)
def f32w(): ops.gemm_pair(0, 1, 0, G, D, R, dg[0], dg[1], G, x, x, D, dwr[0], dwr[1], D)
x6w(); f32w(); torch.cuda.synchronize()
ex = dgc.double().T @ x.double()
print("weight gradient 4096 x 1024 x 9408: max|err| vs fp64: split %.3e, f32 MFMA %.3e (|C| max %.3f)" % ((dw.reshape(2 * G, D).double() - ex).abs().max(), (dwr.reshape(2 * G, D).double() - ex).abs().max(), ex.abs().max()))
ta, tb_, tp, tf = timed(fdgt), timed(fxt), timed(x6w), timed(f32w)
fl = 2.0 * R * D * 2 * G
print("   split dg^T %.0f us, split x^T %.0f us, product %.0f us = %.0f TFLOP/s | f32 MFMA pair %.0f us = %.0f TFLOP/s" % (ta, tb_, tp, fl / tp / 1e6, tf, fl / tf / 1e6))
# ---- recurrent weight gradient: dW_hh(dir) = dg_dir[shifted]^T . y[shifted, half]: k WINDOWS of the plane sets above (shift 32 rows = 2 k16 steps)
Hh = 512; sh = 32
yv = ((torch.rand(R, 2 * Hh, generator=g) - 0.5) * 1.5).to(dev)
yt, fyt = planes(yv, 2 * Hh, R, False, ld=2 * Hh, bound=1.0)                    # rows = 2H (n), k = R
dwh = torch.empty(2, G, Hh, device=dev); dwhr = torch.empty(2, G, Hh, device=dev)
def x6h():
    x6gemm(dgt, 2 * G, R, 0, sh // 16, yt, 2 * Hh, R, 0, 0, G, Hh, R - sh, dwh[0], ldc=Hh, ws_=ws)              # forward: dg_f[sh:]^T . y[:-sh, :H]
    x6gemm(dgt, 2 * G, R, G, 0, yt, 2 * Hh, R, Hh, sh // 16, G, Hh, R - sh, dwh[1], ldc=Hh, ws_=ws)            # reverse: dg_r[:-sh]^T . y[sh:, H:]
def f32h(): ops.gemm_pair(0, 1, 0, G, Hh, R - sh, dg[0][sh:], dg[1], G, yv, yv[sh:, Hh:], 2 * Hh, dwhr[0], dwhr[1], Hh)
x6h(); f32h(); torch.cuda.synchronize()
ex0 = dg[0][sh:].double().T @ yv[:R - sh, :Hh].double(); ex1 = dg[1][:R - sh].double().T @ yv[sh:, Hh:].double()
print("recurrent weight gradient 2 x (2048 x 512 x 9376): max|err| vs fp64: split %.3e / %.3e, f32 MFMA %.3e / %.3e (|C| max %.3f)"
      % ((dwh[0].double() - ex0).abs().max(), (dwh[1].double() - ex1).abs().max(), (dwhr[0].double() - ex0).abs().max(), (dwhr[1].double() - ex1).abs().max(), ex0.abs().max()))
ty, tp, tf = timed(fyt), timed(x6h), timed(f32h)
fl = 2.0 * 2 * G * Hh * (R - sh)
print("   split y^T %.0f us, two products %.0f us = %.0f TFLOP/s | f32 MFMA pair %.0f us = %.0f TFLOP/s" % (ty, tp, fl / tp / 1e6, tf, fl / tf / 1e6))
