"""Host-side operators of the CnnOcrModel path: torch.autograd.Function wrappers that own buffers (torch is
the allocator / stream provider) and call the hand-written HIP kernels through the C-ABI (include/vocr.h).

Every op raises if its tensors are not on a HIP device: there is no CPU fallback (oracle/ is test-only)."""
import math

import torch

from . import _lib
from ._lib import call, prof_range


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ranged(name):
    """Decorator: run a backward staticmethod inside a profiling range (VOCR_ROCTX=1; a no-op otherwise)."""
    def deco(fn):
        if not _lib.ROCTX:
            return fn

        def wrapped(*a, **k):
            with prof_range(name):
                return fn(*a, **k)
        wrapped.__name__ = fn.__name__
        wrapped.__doc__ = fn.__doc__
        return wrapped
    return deco


def _p(t):
    return t.data_ptr() if t is not None else None


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("vistaocr_amd: tensors must live on the MI355X (got a %s tensor); there is no CPU "
                               "fallback for the HIP path" % t.device)


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ---- side stream: weight-gradient GEMMs of an LSTM layer run concurrently with the next (latency-bound) sweep
import os as _os


def _exp(name, default):
    """An experiment switch (A/B of launch plans and kernel variants): its default unless the process runs with VOCR_EXPERIMENTS=1.
    A user's environment cannot change which kernels the product runs; tests and scripts/ opt in explicitly."""
    if _os.environ.get("VOCR_EXPERIMENTS", "0") != "1":
        return default
    return _os.environ.get(name, default)


# Measured on MI355X (bench.py): with one launch per LSTM step the overlap bought nothing (the GEMM delayed every step
# launch as much as it hid); with the persistent chain sweeps it is worth +4 % (1102 -> 1146 img/s).  A persistent sweep
# never waits on these GEMMs, so co-scheduling cannot deadlock it.
# One low-priority side stream per DEVICE (like the allocator's pools): it carries no model state, only "work was queued
# on it since the last join".
_SIDE_ENABLED = _exp("VOCR_SIDE_STREAM", "1") == "1"
_SIDE = {}


def _side(device=None):
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    st = _SIDE.get(idx)
    if st is None:
        lo, hi = torch.cuda.Stream.priority_range()        # (lowest priority value, highest priority value)
        with torch.cuda.device(idx):
            stream = torch.cuda.Stream(priority=lo) if _exp("VOCR_SIDE_LOWPRIO", "1") == "1" else torch.cuda.Stream()
        st = _SIDE[idx] = {"stream": stream, "pending": False}
    return st


def side_stream(device=None):
    return _side(device)["stream"]


_LSTM_FOLLOW = _exp("VOCR_LSTM_FOLLOW", "0") == "1"          # measured slower than the GEMM in front of the next sweep: an experiment switch


def lstm_follow_ok(b, h, din_next, h_next):
    """Layer l+1's x-projection can run behind layer l's forward sweep (include/vocr.h: vocr_lstm_follow_supported)."""
    return _LSTM_FOLLOW and h_next == h and din_next == 2 * h and bool(_lib.load().vocr_lstm_follow_supported(int(b), int(h)))


def mark_side_pending(device=None):
    _side(device)["pending"] = True


def join_side_stream(device=None):
    """Make the current stream wait for every weight-gradient kernel issued on this device's side stream."""
    if not torch.cuda.is_available():
        return
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    st = _SIDE.get(idx)
    if st is not None and st["pending"]:
        torch.cuda.current_stream(idx).wait_stream(st["stream"])
        st["pending"] = False


# ---- health words (include/vocr.h): one int32[2] per device, report-only.  [0]: a persistent LSTM sweep's hand-off timed
# out (that sweep's output is NaN-poisoned; the sweep tests a per-call word of its own, so later sweeps are unaffected),
# [1]: a NaN gradient reached the optimiser.  The kernels set them; train() reads them next to the loss (no extra sync),
# test_on_val / decode_dataset read them once per pass.  A report is cleared when it is raised, so a loop that handles the
# exception (reload the best snapshot, skip the batch) can go on.
_HEALTH = {}


def health(device):
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    h = _HEALTH.get(idx)
    if h is None:
        h = _HEALTH[idx] = torch.zeros(2, dtype=torch.int32, device=torch.device("cuda", idx))
    return h


def reset_health(device=None):
    """Clear the device's health words (stream-ordered memset)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    health(device).zero_()


def check_health(values, device=None):
    """Raise if the two health words (any int sequence, read from `device`) report a failure; the report is cleared first."""
    bad0, bad1 = int(values[0]) != 0, int(values[1]) != 0
    if (bad0 or bad1) and device is not None:
        reset_health(device)
    if bad0:
        raise RuntimeError("vistaocr_amd: a hand-off inside a persistent LSTM sweep timed out (its output is NaN-poisoned); "
                           "is another persistent sweep running on this GPU?  VOCR_LSTM_SWEEP=step selects per-step launches")
    if bad1:
        raise RuntimeError("vistaocr_amd: a NaN gradient reached the optimiser (the reference's clamp_ would propagate it too)")


def check_health_sync(device):
    """Synchronising form: copy the words to the host and check them (eval / decode passes call this once per pass)."""
    check_health(health(device).cpu().tolist(), device)


# ---- direct gradient sinks: an optimiser that owns a persistent gradient buffer per parameter (FlatClampAdam's flat
# buffer, zeroed once per step) marks ITS parameters (p._vocr_sink_owner = the optimiser).  The backward kernels then
# write weight gradients straight into p.grad instead of returning fresh tensors for autograd to add (saves ~56 small
# add kernels and allocations per step).  Valid because every parameter is used exactly once per forward; the owner
# refuses a second backward before zero_grad() (direct stores overwrite, they do not accumulate).  Parameters of other
# models / optimisers in the same process are untouched: there is no module-level switch.
def _sinks(params):
    """The .grad buffers of `params` if every one is owned by the same direct-gradient optimiser, else None."""
    owner = None
    out = []
    for p in params:
        o = getattr(p, "_vocr_sink_owner", None)
        g = p.grad
        if o is None or (owner is not None and o is not owner) or g is None or not g.is_cuda or not g.is_contiguous() \
                or g.dtype != torch.float32:
            return None
        owner = o
        out.append(g)
    if owner is None or not owner.owns_grads(params):
        return None
    owner.note_direct_write(params)
    return out


def _fire(hooks, name):
    """Backward milestones: callbacks registered on the MODEL (model._vocr_hooks), fired from inside its backward pass
    (used to start the data-parallel all-reduce of the sequence-side gradients while the CNN backward is running)."""
    if hooks:
        for fn in hooks.get(name, ()):
            fn()


def _ws(nbytes, device):
    return torch.empty((max(int(nbytes), 16) + 3) // 4, dtype=torch.float32, device=device)


# ---- per-forward preparation off the critical path: everything that depends only on the WEIGHTS (packed conv weights of
# layers 2.., b_ih + b_hh sums, W_hh transposes for the backward sweeps) is computed at the start of the forward pass on
# the side stream, which is idle then, instead of as ~25 small launches in front of the kernels that need them
# (~0.15 ms per step on the main stream).
class ForwardPrep(object):
    def __init__(self):
        self.packs = {}          # weight.data_ptr() -> (pack_fwd, pack_dgrad)
        self.lstm = {}           # w_hh_f.data_ptr() -> (bsum, wt_f, wt_r)
        self.xpack = {}          # w_ih_f.data_ptr() -> W_ih of both directions in the follower's fragment order (vocr_lstm_xproj_pack)
        self.x6w = {}            # w_ih_f.data_ptr() -> (planes of [W_f; W_r] rows = gate rows, k = input: the x-projection's B;
                                 #                       planes of its transpose rows = input, k = gate rows: the data gradient's B or None)
        self.event = None        # everything done (the LSTM items come last: they are needed ~4 ms into the step)
        self.pack_events = {}    # weight.data_ptr() -> event behind THAT layer's pack (the second conv layer needs its pack ~0.15 ms into
                                 # the step; one event behind all ~25 pack launches made it wait ~50 us for the deeper layers' packs)
        self._waited = False

    def wait(self):
        if not self._waited and self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
            self._waited = True

    def wait_pack(self, key):
        ev = self.pack_events.get(key)
        if ev is not None and not self._waited:
            torch.cuda.current_stream().wait_event(ev)


def lstm_bias_sum(b_ih_f, b_hh_f, b_ih_r, b_hh_r):
    """[2, 4H]: b_ih + b_hh of both directions (what the x-projection adds)."""
    H4 = b_ih_f.numel()
    bsum = torch.empty(2, H4, dtype=torch.float32, device=b_ih_f.device)
    call("vocr_add", _p(b_ih_f), _p(b_hh_f), _p(bsum[0]), H4, _stream())
    call("vocr_add", _p(b_ih_r), _p(b_hh_r), _p(bsum[1]), H4, _stream())
    return bsum


def lstm_xproj_pack(w_ih_f, w_ih_r):
    h = w_ih_f.shape[0] // 4
    wpack = torch.empty(_lib.load().vocr_lstm_xproj_pack_bytes(h) // 4, dtype=torch.float32, device=w_ih_f.device)
    call("vocr_lstm_xproj_pack", _p(w_ih_f), _p(w_ih_r), _p(wpack), h, _stream())
    return wpack


def x6_weight_planes(w_ih_f, w_ih_r, with_transpose):
    g4, din = w_ih_f.shape
    wk = x6_planes(w_ih_f, 2 * g4, din, True, din, x2=w_ih_r, seg=g4, axis=1)
    wt = x6_planes(w_ih_f, din, 2 * g4, False, din, x2=w_ih_r, seg=g4, axis=0) if with_transpose else None
    return wk, wt


def forward_prep(conv_weights, lstm_layers, with_transposes, f16=False, follow_layers=(), x6_layers=()):
    """conv_weights: 4-D fp32 weights to pack (f16: into the fp16-operand kernels' packs); lstm_layers: [(w_hh_f, b_ih_f, b_hh_f, w_hh_r,
    b_ih_r, b_hh_r), ...]; follow_layers: [(w_ih_f, w_ih_r), ...] of the layers whose x-projection runs behind the sweep below them;
    x6_layers: [(w_ih_f, w_ih_r), ...] of the layers whose GEMMs run as bf16x6 products (their weight planes are made here)."""
    if not _SIDE_ENABLED or _exp("VOCR_FWD_PREP", "1") != "1":
        return None
    prep = ForwardPrep()
    main = torch.cuda.current_stream()
    side = side_stream()
    side.wait_stream(main)                      # the weights were last written by the previous optimiser step on `main`
    made = []
    with torch.cuda.stream(side):
        for w in conv_weights:
            pk = conv3x3_pack_f16(w) if f16 else conv3x3_pack(w)
            prep.packs[w.data_ptr()] = pk
            made.extend(pk)
            ev = torch.cuda.Event()
            ev.record(side)
            prep.pack_events[w.data_ptr()] = ev
        for (w_hh_f, b_ih_f, b_hh_f, w_hh_r, b_ih_r, b_hh_r) in lstm_layers:
            H4 = b_ih_f.numel()
            bsum = torch.empty(2, H4, dtype=torch.float32, device=b_ih_f.device)
            call("vocr_add", _p(b_ih_f), _p(b_hh_f), _p(bsum[0]), H4, _stream())
            call("vocr_add", _p(b_ih_r), _p(b_hh_r), _p(bsum[1]), H4, _stream())
            wt = (transpose2d(w_hh_f), transpose2d(w_hh_r)) if with_transposes else (None, None)
            prep.lstm[w_hh_f.data_ptr()] = (bsum, wt[0], wt[1])
            made.extend(t for t in (bsum,) + wt if t is not None)
        for (w_ih_f, w_ih_r) in x6_layers:
            # the x-projection of a narrow input (layer 0) is a small product: it keeps the f32 kernels, its weights need no planes
            if w_ih_f.shape[1] >= 512:
                prep.x6w[w_ih_f.data_ptr()] = x6_weight_planes(w_ih_f, w_ih_r, with_transposes)
                made.extend(t for t in prep.x6w[w_ih_f.data_ptr()] if t is not None)
        for (w_ih_f, w_ih_r) in follow_layers:
            prep.xpack[w_ih_f.data_ptr()] = lstm_xproj_pack(w_ih_f, w_ih_r)
            made.append(prep.xpack[w_ih_f.data_ptr()])
        prep.event = torch.cuda.Event()
        prep.event.record(side)
    for t in made:
        t.record_stream(main)
    return prep


# ------------------------------------------------------------------------------------------------ conv + BN + ReLU
_BN_FUSED_FINAL = _exp("VOCR_BN_FUSED_FINAL", "1") == "1"
_WINO = _exp("VOCR_CONV_WINO", "1") == "1"
_WINO_WGRAD = _exp("VOCR_WGRAD_WINO", "1") == "1"


_WINO_MIN_CIN = int(_exp("VOCR_CONV_WINO_MIN_CIN", "1"))


def _wino_ok(cin, cout):
    """The F(2,3)-along-the-row kernel (conv_wino.hip) needs Cout % 4 == 0; input channels are padded to 4 inside the kernel."""
    return _WINO and cin >= _WINO_MIN_CIN and cout % 4 == 0


def conv3x3_pack(weight):
    """(pack_fwd, pack_dgrad) for conv3x3_forward.  Each pack is either the direct kernel's [(ci,kh,kw)][co] or the
    transformed-filter pack of the F(2,3) kernel (12 rows per input channel, followed by the direct pack for its tail pieces);
    conv3x3_forward tells them apart by shape."""
    cout, cin = weight.shape[0], weight.shape[1]
    dev = weight.device
    wf, wd = _wino_ok(cin, cout), _wino_ok(cout, cin)
    # rows per contraction channel: 21 (F(2,3): 12 transformed + 9 direct) or 27 (F(4,3): 18 + 9) - the library says which kernel a
    # convolution with that many output channels gets (forward: cout outputs; data gradient: cin outputs)
    lib = _lib.load()
    rows_f = lib.vocr_conv3x3_wino_pack_floats(cout, cin) // (cout * cin)
    rows_d = lib.vocr_conv3x3_wino_pack_floats(cin, cout) // (cout * cin)
    pf = torch.empty(cin * (rows_f if wf else 9), cout, dtype=torch.float32, device=dev)
    pd = torch.empty(cout * (rows_d if wd else 9), cin, dtype=torch.float32, device=dev)
    if wf or wd:
        call("vocr_conv3x3_wino_pack_weights", _p(weight), _p(pf) if wf else None, _p(pd) if wd else None, cout, cin, _stream())
    if not (wf and wd):
        call("vocr_conv3x3_pack_weights", _p(weight), None if wf else _p(pf), None if wd else _p(pd), cout, cin, _stream())
    return pf, pd


def conv3x3_pack_f16(weight):
    """fp16 weight packs (forward, data-gradient) for the fp16-operand MFMA conv."""
    cout, cin = weight.shape[0], weight.shape[1]
    lib = _lib.load()
    pf = torch.empty(lib.vocr_conv3x3_f16_pack_bytes(cout, cin, 0) // 2, dtype=torch.float16, device=weight.device)
    pd = torch.empty(lib.vocr_conv3x3_f16_pack_bytes(cout, cin, 1) // 2, dtype=torch.float16, device=weight.device)
    call("vocr_conv3x3_f16_pack_weights", _p(weight), _p(pf), _p(pd), cout, cin, _stream())
    return pf, pd


def f16_layouts(x, want_nhwc=True, want_nchwp=False):
    """fp32 [N,C,H,W] -> the fp16 operand copies of the fp16-operand conv kernels, one pass over x (vocr_f32_to_f16_layouts):
    nhwc  [N,C/16,H,W,16]  forward / data gradient (channel-blocked NHWC): the 8 channels of a lane half are 16 contiguous bytes and a
                           row segment of a 16-channel block is one contiguous run, so the kernel's DMA pieces use whole cache lines;
    nchwp [N,C,H,WP]  weight gradient: 8 consecutive pixels of a channel, rows zero-padded to WP = ceil8(W) + 8."""
    n, c, h, w = x.shape
    nhwc = torch.empty(n, c // 16, h, w, 16, dtype=torch.float16, device=x.device) if want_nhwc else None
    nchwp = torch.empty(n, c, h, _lib.load().vocr_f16_padded_row(w), dtype=torch.float16, device=x.device) if want_nchwp else None
    call("vocr_f32_to_f16_layouts", _p(x), _p(nhwc), _p(nchwp), n, c, h, w, _stream())
    return nhwc, nchwp


def h16_takes(n, cin, h, w, cout):
    """The all-DMA fp16 kernels (and the one-pass layout conversion in front of them) take this launch: channel counts
    (vocr_conv3x3_h16_supported) AND sizes - the conversion pass walks n*h rows in one grid dimension (<= 65535), the kernels address the
    fp16 copies with 32-bit byte offsets (< 2 GB).  Beyond that the register-staged fp16 kernels run, straight on the fp32 tensors."""
    return bool(_lib.load().vocr_conv3x3_h16_supported(cin, cout)) and n * h <= 65535 and n * h * w * max(cin, cout) * 2 < 2 ** 31


def conv3x3_forward_f16(x, wpack16, bias, cout, x_nhwc=None):
    """fp16 operands, fp32 accumulate, fp32 NCHW in and out (BASELINE configs[4]).  Cin % 16 == 0: the all-DMA kernel on an NHWC fp16
    copy of x (`x_nhwc`, made here unless the caller has it); otherwise (the rapid_ds stage's Cin = 1) the register-staged kernel."""
    n, cin, h, w = x.shape
    y = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    if h16_takes(n, cin, h, w, cout):
        if x_nhwc is None:
            x_nhwc = f16_layouts(x)[0]
        call("vocr_conv3x3_h16_fwd", _p(x_nhwc), _p(wpack16), _p(bias), _p(y), n, cin, h, w, cout, _stream())
    else:
        call("vocr_conv3x3_f16_fwd", _p(x), _p(wpack16), _p(bias), _p(y), n, cin, h, w, cout, _stream())
    return y


def wgrad_f16_layouts_ok(cin, cout):
    """The fp16 weight-gradient kernel on the channel-major padded copies takes this layer (Cin % 64 == 0, Cout 64 or a multiple of 128)."""
    return bool(_lib.load().vocr_conv3x3_wgrad_h16_supported(cin, cout))


def conv3x3_forward(x, wpack, bias, cout):
    n, cin, h, w = x.shape
    y = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    # the pack says which kernel it is for: transformed + 9 direct rows per input channel (conv_wino.hip: 21 for the F(2,3) kernel, 27 for
    # F(4,3) - the library decides by the output channel count, so the row count is checked against ITS answer: a pack built for another
    # format would be read out of bounds) or the 9 taps alone (conv.hip); the shape survives save_for_backward, an attribute would not
    lib = _lib.load()
    rows = wpack.shape[0] if wpack.dim() == 2 else -1
    wino_rows = lib.vocr_conv3x3_wino_pack_floats(cout, cin) // (cout * cin) * cin if lib.vocr_conv3x3_wino_supported(cin, cout) else -2
    if wpack.dim() != 2 or wpack.shape[1] != cout or rows not in (9 * cin, wino_rows):
        raise ValueError("conv3x3_forward: weight pack %s does not fit cin=%d cout=%d (expected %d or %d rows)"
                         % (tuple(wpack.shape), cin, cout, 9 * cin, wino_rows))
    if rows != 9 * cin and n * max(cin, cout) * h * w >= (1 << 29):
        # the minimal-filtering kernels address tensors with 32-bit byte offsets (< 2^29 elements); a larger one takes the direct kernel
        # on the direct-form rows that trail every transformed pack (the tail pieces' operand)
        wpack = wpack[rows - 9 * cin:]
        rows = 9 * cin
    fn = "vocr_conv3x3_wino_fwd" if rows != 9 * cin else "vocr_conv3x3_fwd"
    call(fn, _p(x), _p(wpack), _p(bias), _p(y), n, cin, h, w, cout, _stream())
    return y


def conv3x3_c1_forward(x, weight, bias, f16=False):
    """One input channel: the vector-arithmetic kernel straight on the layer's weight (no pack); f16 rounds the operands like the
    fp16-operand kernels do."""
    n, _, h, w = x.shape
    cout = weight.shape[0]
    y = torch.empty(n, cout, h, w, dtype=torch.float32, device=x.device)
    call("vocr_conv3x3_c1_fwd", _p(x), _p(weight), _p(bias), _p(y), n, h, w, cout, int(bool(f16)), _stream())
    return y


def conv3x3_c1_wgrad_takes(cin, cout):
    """One input channel and few output channels: the vector-arithmetic weight-gradient kernel (which can deliver the bias gradient too)."""
    return cin == 1 and cout < 32


def conv3x3_wgrad(x, dy, out=None, f16=False, x16p=None, dy16p=None, dbias_out=None):
    """dbias_out: where the layer's bias gradient (per-channel sum of dy) goes if the kernel that runs can produce it in the same pass
    over dy (ask conv3x3_c1_wgrad_takes); ignored by the other kernels."""
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    lib = _lib.load()
    dw = out if out is not None else torch.empty(cout, cin, 3, 3, dtype=torch.float32, device=x.device)
    if conv3x3_c1_wgrad_takes(cin, cout):
        # one input channel and few output channels (configs[4]'s rapid_ds stage, 1 -> 16): stream dy once, exact fp32 in either
        # configuration (116 us against 200 on the MFMA kernel below, whose 64-channel tile is three quarters empty there; with 64
        # output channels that kernel stays ahead, 52 against 69 us)
        ws = _ws(lib.vocr_conv3x3_c1_wgrad_workspace_bytes(n, h, cout), x.device)
        call("vocr_conv3x3_c1_wgrad", _p(x), _p(dy), _p(dw), _p(dbias_out), _p(ws), n, h, w, cout, _stream())
        return dw
    if f16 and x16p is not None and dy16p is not None:
        # fp16 operands from the channel-major padded copies (conv3x3_wgrad_h16_kernel): all-DMA staging, 9 tap accumulators per wave
        ws = _ws(lib.vocr_conv3x3_wgrad_h16_workspace_bytes(n, cin, h, w, cout), x.device)
        call("vocr_conv3x3_wgrad_h16", _p(x16p), _p(dy16p), _p(dw), _p(ws), n, cin, h, w, cout, _stream())
        return dw
    # Without those copies the fp16 configuration takes the fp32 row-pair kernel wherever it applies: it issues 4/9 of the direct form's
    # multiplications and measured FASTER than the register-staged fp16 weight-gradient kernel (2.8 vs 3.8 ms per step in the round-5
    # bench), on exact fp32 operands; `f16` then only decides the kernel of the layers that kernel does not take (Cin < 4)
    if _WINO_WGRAD and cin >= 4 and n * max(cin, cout) * h * w < (1 << 29):
        ws = _ws(lib.vocr_conv3x3_wgrad_wino_workspace_bytes(n, cin, h, w, cout), x.device)
        call("vocr_conv3x3_wgrad_wino", _p(x), _p(dy), _p(dw), _p(ws), n, cin, h, w, cout, _stream())
        return dw
    ws = _ws(lib.vocr_conv3x3_wgrad_workspace_bytes(n, cin, h, w, cout), x.device)
    call("vocr_conv3x3_wgrad_f16" if f16 else "vocr_conv3x3_wgrad", _p(x), _p(dy), _p(dw), _p(ws), n, cin, h, w, cout, _stream())
    return dw


def channel_sum(x, out=None):
    n, c, h, w = x.shape
    out = out if out is not None else torch.empty(c, dtype=torch.float32, device=x.device)
    nb = _lib.load().vocr_channel_sum_workspace_bytes(n, c, h * w)
    ws = _ws(nb, x.device) if nb else None
    call("vocr_channel_sum", _p(x), _p(out), n, c, h * w, _p(ws), _stream())
    return out


class ConvBnReluFn(torch.autograd.Function):
    """Conv2d(k3,p1) -> BatchNorm2d -> ReLU (reference ConvBNReLU, src/models/cnnlstm.py:263-266)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, training, eps, momentum, f16=False,
                pool_samples=None, pool_oh=0, pool_ow=0, num_batches_tracked=None, prep=None):
        """With `pool_samples` (N, C, 2) the FractionalMaxPool2d that follows this layer (cnnlstm.py:127,130) is applied in
        the same pass as BatchNorm + ReLU and the pooled tensor is returned: the unpooled activation is needed by nobody."""
        _need_gpu(x, weight, bias, gamma, beta, running_mean, running_var)
        x = _f32c(x)
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        lib = _lib.load()
        ctx.f16 = bool(f16)
        x16p = None
        if cin == 1 and not ctx.needs_input_grad[0]:
            # grey lines' first layer: bound by its output stream, not by arithmetic (no pack; nothing flows back into the image)
            pf = pd = None
            y = conv3x3_c1_forward(x, weight, bias, ctx.f16)
        elif ctx.f16:
            if prep is not None and weight.data_ptr() in prep.packs:
                prep.wait_pack(weight.data_ptr())
                pf, pd = prep.packs[weight.data_ptr()]
            else:
                pf, pd = conv3x3_pack_f16(weight)
            x_nhwc = None
            if h16_takes(n, cin, h, w, cout):
                # one pass over x: the forward operand and (training) the weight gradient's channel-major copy, kept for the backward
                x_nhwc, x16p = f16_layouts(x, True, bool(training) and wgrad_f16_layouts_ok(cin, cout))
            y = conv3x3_forward_f16(x, pf, bias, cout, x_nhwc)
        else:
            if prep is not None and weight.data_ptr() in prep.packs:
                prep.wait_pack(weight.data_ptr())
                pf, pd = prep.packs[weight.data_ptr()]
            else:
                pf, pd = conv3x3_pack(weight)
            y = conv3x3_forward(x, pf, bias, cout)
        mean = torch.empty(cout, dtype=torch.float32, device=x.device)
        invstd = torch.empty(cout, dtype=torch.float32, device=x.device)
        xhat_sum = None
        # training mode: statistics pass + apply pass in two launches (the apply pass adds each channel's partial sums itself; VOCR_BN_FUSED_FINAL=0:
        # round 3's three launches - bit-identical results)
        fused_final = training and _BN_FUSED_FINAL
        ws = None
        if training:
            ws = _ws(lib.vocr_bn_workspace_bytes(n, cout, h * w), x.device)
            xhat_sum = torch.empty(cout, dtype=torch.float32, device=x.device)
            if not fused_final:
                call("vocr_bn_train_stats", _p(y), n, cout, h * w, eps, momentum, _p(mean), _p(invstd), _p(running_mean),
                     _p(running_var), _p(num_batches_tracked), _p(xhat_sum), _p(ws), _stream())
        else:
            call("vocr_bn_eval_stats", _p(running_mean), _p(running_var), cout, eps, _p(mean), _p(invstd), _stream())
        ctx.pool = None
        idx = samples = None
        if pool_samples is not None:
            samples = _f32c(pool_samples)
            if tuple(samples.shape) != (n, cout, 2):
                raise RuntimeError("fractional pool samples must have shape (N, C, 2)")
            out = torch.empty(n, cout, pool_oh, pool_ow, dtype=torch.float32, device=x.device)
            idx = torch.empty(n, cout, pool_oh, pool_ow, dtype=torch.int32, device=x.device)
            if fused_final:
                call("vocr_bn_train_relu_fracpool2x2_fwd", _p(y), _p(gamma), _p(beta), _p(samples), _p(out), _p(idx), n, cout, h, w, pool_oh, pool_ow,
                     eps, momentum, _p(mean), _p(invstd), _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(xhat_sum), _p(ws), _stream())
            else:
                call("vocr_bn_relu_fracpool2x2_fwd", _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(samples), _p(out), _p(idx),
                     n, cout, h, w, pool_oh, pool_ow, _stream())
            ctx.pool = (pool_oh, pool_ow)
        else:
            out = torch.empty_like(y)
            if fused_final:
                call("vocr_bn_train_relu_apply", _p(y), _p(gamma), _p(beta), _p(out), n, cout, h * w, eps, momentum, _p(mean), _p(invstd),
                     _p(running_mean), _p(running_var), _p(num_batches_tracked), _p(xhat_sum), _p(ws), _stream())
            else:
                call("vocr_bn_relu_apply", _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(out), n, cout, h * w, _stream())
        ctx.training = training
        ctx.prefs = (weight, bias, gamma, beta)
        ctx.save_for_backward(x, y, mean, invstd, gamma, beta, pd, idx, samples, xhat_sum, x16p)
        return out

    @staticmethod
    @_ranged("bwd.conv_bn_relu")
    def backward(ctx, da):
        if not ctx.training:
            raise RuntimeError("vistaocr_amd: backward through eval-mode BatchNorm is not part of the reference path")
        x, y, mean, invstd, gamma, beta, pd, idx, samples, xhat_sum, x16p = ctx.saved_tensors
        da = _f32c(da)
        n, cin, h, w = x.shape
        cout = y.shape[1]
        lib = _lib.load()
        fused_pool_bwd = False
        if ctx.pool is not None:
            oh, ow = ctx.pool
            # Pooling gradient + ReLU + BatchNorm backward in one gather pass over the plane (VOCR_POOL_BWD_FUSED=0: the pooling
            # gradient as a pass of its own that materialises the full plane).  Alone on the chip it always beat the three passes
            # (139 vs 167 us per layer); in the step it was neutral in round 2 (starved beside the weight-gradient kernel) and is
            # worth -0.15 ms since the round-3 GEMMs (same-box A/B 18.96 -> 18.81 ms).  Bit-identical results either way.
            fused_pool_bwd = bool(lib.vocr_bn_relu_fracpool2x2_bwd_supported(h, w, oh, ow)) and _exp("VOCR_POOL_BWD_FUSED", "1") == "1"
            if not fused_pool_bwd:          # gradient of the fused pooling first: back to the full plane
                dfull = torch.empty(n, cout, h, w, dtype=torch.float32, device=da.device)
                call("vocr_fracpool2x2_bwd", _p(da), _p(idx), _p(dfull), n, cout, h, w, oh, ow, _stream())
                da = dfull
        dy = torch.empty_like(y)
        sinks = _sinks(ctx.prefs)
        if sinks is not None:
            dw, dbias, dgamma, dbeta = sinks
        else:
            dw = None
            dgamma = torch.empty_like(gamma)
            dbeta = torch.empty_like(beta)
            dbias = torch.empty(cout, dtype=torch.float32, device=x.device)
        ws = _ws(lib.vocr_bn_workspace_bytes(n, cout, h * w), x.device)
        if fused_pool_bwd:
            # pooled layer: pooling gradient (gather form) + ReLU + BatchNorm backward in one pass over the plane; the two
            # BatchNorm sums come from the pooled tensors alone
            call("vocr_bn_relu_fracpool2x2_bwd", _p(da), _p(idx), _p(samples), _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta),
                 _p(xhat_sum), _p(dy), _p(dgamma), _p(dbeta), _p(dbias), n, cout, h, w, oh, ow, _p(ws), _stream())
        else:
            call("vocr_bn_relu_bwd", _p(da), _p(y), _p(mean), _p(invstd), _p(gamma), _p(beta), _p(xhat_sum), _p(dy), _p(dgamma),
                 _p(dbeta), _p(dbias), n, cout, h * w, _p(ws), _stream())
        dy_nhwc = dy16p = None
        if ctx.f16:
            # one pass over dy: the data gradient's NHWC operand and the weight gradient's channel-major padded one
            want_nhwc = bool(ctx.needs_input_grad[0]) and h16_takes(dy.shape[0], cout, dy.shape[2], dy.shape[3], cin)
            if want_nhwc or x16p is not None:
                dy_nhwc, dy16p = f16_layouts(dy, want_nhwc, x16p is not None)
        if sinks is not None and _SIDE_ENABLED and ctx.needs_input_grad[0] and _exp("VOCR_CONV_OVERLAP", "1") == "1":
            # weight gradient (off the critical path, written straight into the optimiser's buffer) on the low-priority side
            # stream beside the data gradient: each kernel's last partial round of workgroups is filled by the other's
            side = side_stream()
            side.wait_stream(torch.cuda.current_stream())
            for t_ in (x, dy, x16p, dy16p):
                if t_ is not None:
                    t_.record_stream(side)
            with torch.cuda.stream(side):
                conv3x3_wgrad(x, dy, out=dw, f16=ctx.f16, x16p=x16p, dy16p=dy16p)
            mark_side_pending()
        else:
            dw = conv3x3_wgrad(x, dy, out=dw, f16=ctx.f16, x16p=x16p, dy16p=dy16p)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv3x3_forward_f16(dy, pd, None, cin, dy_nhwc) if ctx.f16 else conv3x3_forward(dy, pd, None, cin)
        else:
            # first layer = last backward op: whoever reads the gradients on this stream next finds the side stream's weight gradients
            # done.  BEHIND this layer's own kernels: in front of them (round 4) the main stream sat idle for the tail of the second
            # layer's weight gradient before it started this layer's BatchNorm backward and weight gradient (same-box A/B 15.43 / 15.42
            # against 15.38 / 15.40 ms)
            join_side_stream()
        if sinks is not None:
            return (dx,) + (None,) * 15
        return (dx, dw, dbias, dgamma, dbeta) + (None,) * 11


class ConvReluPoolFn(torch.autograd.Function):
    """rapid_ds stage: Conv2d(k3,p1) -> ReLU -> MaxPool2d(2,2) (src/models/cnnlstm.py:114-121)."""

    @staticmethod
    def forward(ctx, x, weight, bias, f16=False):
        _need_gpu(x, weight, bias)
        x = _f32c(x)
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        ctx.f16 = bool(f16)
        if cin == 1 and not ctx.needs_input_grad[0]:
            pf = pd = None
            y = conv3x3_c1_forward(x, weight, bias, ctx.f16)
        elif ctx.f16:
            pf, pd = conv3x3_pack_f16(weight)
            y = conv3x3_forward_f16(x, pf, bias, cout)
        else:
            pf, pd = conv3x3_pack(weight)
            y = conv3x3_forward(x, pf, bias, cout)
        oh, ow = h // 2, w // 2
        out = torch.empty(n, cout, oh, ow, dtype=torch.float32, device=x.device)
        idx = torch.empty(n, cout, oh, ow, dtype=torch.int32, device=x.device)
        call("vocr_relu_maxpool2_fwd", _p(y), _p(out), _p(idx), n, cout, h, w, _stream())
        ctx.save_for_backward(x, out, idx, pd)
        ctx.prefs = (weight, bias)
        ctx.hw = (h, w)
        return out

    @staticmethod
    @_ranged("bwd.rapid_ds")
    def backward(ctx, dout):
        x, out, idx, pd = ctx.saved_tensors
        dout = _f32c(dout)
        n, cin, _, _ = x.shape
        cout = out.shape[1]
        h, w = ctx.hw
        dy = torch.zeros(n, cout, h, w, dtype=torch.float32, device=x.device)
        call("vocr_relu_maxpool2_bwd", _p(dout), _p(out), _p(idx), _p(dy), n, cout, h, w, _stream())
        sinks = _sinks(ctx.prefs)
        if conv3x3_c1_wgrad_takes(cin, cout):
            # one pass over dy for both parameter gradients
            dbias = sinks[1] if sinks else torch.empty(cout, dtype=torch.float32, device=x.device)
            dw = conv3x3_wgrad(x, dy, out=sinks[0] if sinks else None, f16=ctx.f16, dbias_out=dbias)
        else:
            dbias = channel_sum(dy, out=sinks[1] if sinks else None)
            dw = conv3x3_wgrad(x, dy, out=sinks[0] if sinks else None, f16=ctx.f16)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv3x3_forward_f16(dy, pd, None, cin) if ctx.f16 else conv3x3_forward(dy, pd, None, cin)
        else:
            join_side_stream()          # last backward op of the model: behind its own kernels (see ConvBnReluFn.backward)
        if sinks is not None:
            return dx, None, None, None
        return dx, dw, dbias, None


class FracPoolFn(torch.autograd.Function):
    """nn.FractionalMaxPool2d(2, output_ratio=(0.5, 0.7)) with explicit samples (src/models/cnnlstm.py:127,130)."""

    @staticmethod
    def forward(ctx, x, samples, oh, ow):
        _need_gpu(x, samples)
        x = _f32c(x)
        samples = _f32c(samples)
        n, c, h, w = x.shape
        if tuple(samples.shape) != (n, c, 2):
            raise RuntimeError("fractional pool samples must have shape (N, C, 2)")
        out = torch.empty(n, c, oh, ow, dtype=torch.float32, device=x.device)
        idx = torch.empty(n, c, oh, ow, dtype=torch.int32, device=x.device)
        call("vocr_fracpool2x2_fwd", _p(x), _p(samples), _p(out), _p(idx), n, c, h, w, oh, ow, _stream())
        ctx.save_for_backward(idx)
        ctx.shape = (n, c, h, w, oh, ow)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        n, c, h, w, oh, ow = ctx.shape
        dout = _f32c(dout)
        dx = torch.empty(n, c, h, w, dtype=torch.float32, device=dout.device)       # fully written by the kernel
        call("vocr_fracpool2x2_bwd", _p(dout), _p(idx), _p(dx), n, c, h, w, oh, ow, _stream())
        return dx, None, None, None


# ------------------------------------------------------------------------------------------------ dense layers
def gemm(ta, tb, m, n, k, a, lda, b, ldb, c, ldc, bias=None, relu=False, accumulate=False):
    # split-K slabs (long-K weight gradients): summed in a fixed order by the library, so results are reproducible
    nb = _lib.load().vocr_gemm_workspace_bytes(m, n, k, int(bias is not None or relu))
    ws = _ws(nb, c.device) if nb else None
    call("vocr_gemm", int(ta), int(tb), m, n, k, _p(a), lda, _p(b), ldb, _p(c), ldc, _p(bias), int(relu), int(accumulate),
         _p(ws), nb, _stream())


def gemm_pair(mode, ta, tb, m, n, k, a0, a1, lda, b0, b1, ldb, c0, c1, ldc, bias0=None, bias1=None, relu=False):
    """Two products of one shape in one launch (include/vocr.h: vocr_gemm_pair): mode 0 = two independent products, mode 1 =
    c0 = a0 b0 + a1 b1.  The two directions of a BiLSTM layer."""
    nb = _lib.load().vocr_gemm_pair_workspace_bytes(m, n, k, int(mode))
    ws = _ws(nb, c0.device) if nb else None
    call("vocr_gemm_pair", int(mode), int(ta), int(tb), m, n, k, _p(a0), _p(a1), lda, _p(b0), _p(b1), ldb, _p(c0), _p(c1), ldc,
         _p(bias0), _p(bias1), int(relu), _p(ws), nb, _stream())


# ---- fp32 GEMM on the 16-bit matrix pipe by operand splitting (include/vocr.h: vocr_gemm_x6* / vocr_gemm_h3*): the LSTM projections and their gradients
_GEMM_X6 = _exp("VOCR_GEMM_X6", "1") == "1"
_X6_TWO_VIEWS = _exp("VOCR_X6_TWO_VIEWS", "1") == "1"          # the recurrent weight gradients of both directions as one launch
# which split the large LSTM products use: "bf16x6" (default: three bf16 planes that sum to the operand EXACTLY, six products), "fp16x3" (opt-in: two fp16
# planes with per-row power-of-two scales, three products - half the matrix instructions, a norm-wise instead of a per-product error bound), "f32" (the
# f32-MFMA kernels).  VOCR_LSTM_GEMM in the environment or set_lstm_gemm() before the first forward pass.
LSTM_GEMM_MODES = ("bf16x6", "fp16x3", "f32")
_LSTM_GEMM = _os.environ.get("VOCR_LSTM_GEMM", "bf16x6")
if _LSTM_GEMM not in LSTM_GEMM_MODES:
    raise ValueError("VOCR_LSTM_GEMM=%r: one of %s" % (_LSTM_GEMM, ", ".join(LSTM_GEMM_MODES)))
_X6_ENTRY = {"bf16x6": ("vocr_gemm_x6_planes_bytes", "vocr_gemm_x6_split", "vocr_gemm_x6", "vocr_gemm_x6_two_views"),
             "fp16x3": ("vocr_gemm_h3_planes_bytes", "vocr_gemm_h3_split", "vocr_gemm_h3", "vocr_gemm_h3_two_views")}


def set_lstm_gemm(mode):
    """Select the arithmetic of the BiLSTM stack's large GEMMs for this process (see LSTM_GEMM_MODES); returns the previous mode."""
    global _LSTM_GEMM
    if mode not in LSTM_GEMM_MODES:
        raise ValueError("lstm gemm mode %r: one of %s" % (mode, ", ".join(LSTM_GEMM_MODES)))
    prev, _LSTM_GEMM = _LSTM_GEMM, mode
    return prev


def lstm_gemm():
    return _LSTM_GEMM if _GEMM_X6 else "f32"


def x6_layer_ok(T, B, H, rows):
    """The BiLSTM layer's large GEMMs run as split products (any row layout; the recurrent weight gradient joins them when its time shift -
    B rows dense, 4 packed - is a whole number of k16 steps)."""
    return lstm_gemm() != "f32" and (4 * H) % 128 == 0 and (rows if rows else T * B) >= 256


def x6_planes(x, rows, k, k_contiguous, ld, x2=None, seg=0, axis=0, mask=None, scheme=None, bound=0.0):
    """The 16-bit planes of an fp32 operand in MFMA-fragment order (one pass over x; fp16x3: one more for the rows' maxima unless the caller
    knows a `bound` of every |element| after the mask - an LSTM output lies inside (-1, 1))."""
    scheme = scheme or lstm_gemm()
    nbytes, split = _X6_ENTRY[scheme][:2]
    lib = _lib.load()
    buf = torch.empty(getattr(lib, nbytes)(int(rows), int(k)) // 2, dtype=torch.int16, device=x.device)
    if scheme == "fp16x3":
        call(split, _p(x), _p(x2), int(seg), int(axis), _p(mask), int(ld), int(rows), int(k), int(bool(k_contiguous)), float(bound), _p(buf), _stream())
    else:
        call(split, _p(x), _p(x2), int(seg), int(axis), _p(mask), int(ld), int(rows), int(k), int(bool(k_contiguous)), _p(buf), _stream())
    buf.x6_scheme = scheme
    return buf


def gemm_x6(a, a_rows, a_k, b, b_rows, b_k, m, n, k, c0, ldc, c1=None, csplit=0, rsplit=0, bias0=None, bias1=None, a_row0=0, a_kk0=0, b_row0=0,
            b_kk0=0, relu=False):
    """c = A . B^T from split planes (views: row / k16 offsets into plane sets written for (a_rows, a_k) and (b_rows, b_k))."""
    scheme = getattr(a, "x6_scheme", None)
    if scheme is None or scheme != getattr(b, "x6_scheme", None):
        raise RuntimeError("gemm_x6: operands from different splits (%r, %r)" % (scheme, getattr(b, "x6_scheme", None)))
    ws = _ws(_lib.load().vocr_gemm_x6_workspace_bytes(m, n, k), c0.device)
    call(_X6_ENTRY[scheme][2], _p(a), int(a_rows), int(a_k), int(a_row0), int(a_kk0), _p(b), int(b_rows), int(b_k), int(b_row0), int(b_kk0), int(m), int(n),
         int(k), _p(c0), _p(c1), int(csplit), int(rsplit), int(ldc), _p(bias0), _p(bias1), int(relu), _p(ws), _stream())


def gemm_x6_two_views(a, a_rows, a_k, b, b_rows, b_k, m, n, k, rsplit, views0, views1, c0, c1, ldc, a_row0=0):
    """Two products of one shape in one launch (include/vocr.h: vocr_gemm_x6_two_views): rows < rsplit with views0 = (a_kk0, b_row0, b_kk0) into c0,
    rows >= rsplit with views1 into c1."""
    scheme = getattr(a, "x6_scheme", None)
    if scheme is None or scheme != getattr(b, "x6_scheme", None):
        raise RuntimeError("gemm_x6_two_views: operands from different splits (%r, %r)" % (scheme, getattr(b, "x6_scheme", None)))
    ws = _ws(_lib.load().vocr_gemm_x6_workspace_bytes(m, n, k), c0.device)
    call(_X6_ENTRY[scheme][3], _p(a), int(a_rows), int(a_k), int(a_row0), _p(b), int(b_rows), int(b_k), int(m), int(n), int(k), int(rsplit),
         int(views0[0]), int(views0[1]), int(views0[2]), int(views1[0]), int(views1[1]), int(views1[2]), _p(c0), _p(c1), int(ldc), _p(ws), _stream())


def colsum(x2d, out=None):
    m, n = x2d.shape
    out = out if out is not None else torch.empty(n, dtype=torch.float32, device=x2d.device)
    nb = _lib.load().vocr_colsum_workspace_bytes(m, n)
    ws = _ws(nb, x2d.device) if nb else None
    call("vocr_colsum", _p(x2d), _p(out), m, n, _p(ws), _stream())
    return out


def transpose2d(x):
    """[R][C] -> [C][R] (the bchw->wbch kernel with b=1 is a tiled matrix transpose)."""
    r, c = x.shape
    out = torch.empty(c, r, dtype=torch.float32, device=x.device)
    call("vocr_bchw_to_wbch", _p(x), _p(out), 1, r, 1, c, _stream())
    return out


class PermuteBchwToWbchFn(torch.autograd.Function):
    """cnn_output.permute(3, 0, 1, 2).contiguous().view(-1, c*h) (src/models/cnnlstm.py:275-278)."""

    @staticmethod
    def forward(ctx, x, hooks=None):
        _need_gpu(x)
        x = _f32c(x)
        b, c, h, w = x.shape
        out = torch.empty(w * b, c * h, dtype=torch.float32, device=x.device)
        call("vocr_bchw_to_wbch", _p(x), _p(out), b, c, h, w, _stream())
        ctx.shape = (b, c, h, w)
        ctx.hooks = hooks
        return out

    @staticmethod
    @_ranged("bwd.permute")
    def backward(ctx, dout):
        b, c, h, w = ctx.shape
        dout = _f32c(dout)
        # everything above the CNN (bridge, LSTM, prob) has produced its gradients by now
        _fire(ctx.hooks, "sequence_grads_ready")
        dx = torch.empty(b, c, h, w, dtype=torch.float32, device=dout.device)
        call("vocr_wbch_to_bchw", _p(dout), _p(dx), b, c, h, w, _stream())
        return dx, None


class LinearFn(torch.autograd.Function):
    """nn.Linear (+ optional ReLU): bridge_layer and prob_layer (src/models/cnnlstm.py:143-154)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        _need_gpu(x, weight, bias)
        x = _f32c(x)
        m, k = x.shape
        n = weight.shape[0]
        out = torch.empty(m, n, dtype=torch.float32, device=x.device)
        gemm(0, 1, m, n, k, x, k, weight, k, out, n, bias=bias, relu=relu)
        ctx.relu = relu
        ctx.prefs = (weight, bias)
        ctx.save_for_backward(x, weight, out if relu else None)
        return out

    @staticmethod
    @_ranged("bwd.linear")
    def backward(ctx, dout):
        x, weight, out = ctx.saved_tensors
        dout = _f32c(dout)
        m, k = x.shape
        n = weight.shape[0]
        if ctx.relu:
            dz = torch.empty_like(dout)
            call("vocr_relu_bwd", _p(dout), _p(out), _p(dz), dout.numel(), _stream())
        else:
            dz = dout
        sinks = _sinks(ctx.prefs)
        # critical path first: dx feeds the layer below (the bridge's dx is what the whole CNN backward waits for; with the weight
        # gradient, its split-K reduction and the bias column sums in front of it the chain was 0.27 ms longer)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            gemm(0, 0, m, k, n, dz, n, weight, k, dx, k)            # dx[m,k] = dz[m,n] W[n,k]
        # The bridge's parameter gradients go straight into the optimiser's buffers and nothing downstream of this node reads them, so they
        # run on the low-priority side stream (joined before the optimiser step like every other weight gradient).  The prob layer's
        # stay on the main stream: on the side stream its 0.05-ms product sat beside the top layer's persistent backward sweep for the
        # sweep's whole length and cost it 0.09 ms (same-box 15.35 / 15.38 -> 15.34 / 15.34 ms).  VOCR_LINEAR_DW_OVERLAP (experiments):
        # 1 both layers on the side stream, 0 neither.
        ov = _exp("VOCR_LINEAR_DW_OVERLAP", "2")
        if sinks is not None and _SIDE_ENABLED and ctx.needs_input_grad[0] and (ov == "1" or (ov == "2" and ctx.relu)):
            side = side_stream()
            side.wait_stream(torch.cuda.current_stream())
            for t_ in (dz, x):
                t_.record_stream(side)
            with torch.cuda.stream(side):
                gemm(1, 0, n, k, m, dz, n, x, k, sinks[0], k)       # dW[n,k] = dz^T[n,m] x[m,k]
                colsum(dz, out=sinks[1])
            mark_side_pending()
            return dx, None, None, None
        dw = sinks[0] if sinks else torch.empty_like(weight)
        gemm(1, 0, n, k, m, dz, n, x, k, dw, k)                     # dW[n,k] = dz^T[n,m] x[m,k]
        db = colsum(dz, out=sinks[1] if sinks else None)
        if sinks is not None:
            return dx, None, None, None
        return dx, dw, db, None


class MulMaskFn(torch.autograd.Function):
    """inter-layer LSTM dropout with an explicit pre-scaled mask (parity path)."""

    @staticmethod
    def forward(ctx, x, mask):
        _need_gpu(x, mask)
        x, mask = _f32c(x), _f32c(mask)
        out = torch.empty_like(x)
        call("vocr_mul", _p(x), _p(mask), _p(out), x.numel(), _stream())
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, dout):
        (mask,) = ctx.saved_tensors
        dout = _f32c(dout)
        dx = torch.empty_like(dout)
        call("vocr_mul", _p(dout), _p(mask), _p(dx), dout.numel(), _stream())
        return dx, None


def dropout_mask(nrows, ncols, p, seed, device):
    """[nrows, ncols] pre-scaled mask of vocr_dropout_fwd's counter-based draw (element index = row * ncols + col)."""
    mask = torch.empty(int(nrows), int(ncols), dtype=torch.float32, device=device)
    call("vocr_dropout_mask", _p(mask), mask.numel(), float(p), int(seed), _stream())
    return mask


class DropoutFn(torch.autograd.Function):
    """nn.LSTM(dropout=p) between layers in training: counter-based mask drawn on device."""

    @staticmethod
    def forward(ctx, x, p, seed):
        _need_gpu(x)
        x = _f32c(x)
        out = torch.empty_like(x)
        mask = torch.empty_like(x)
        call("vocr_dropout_fwd", _p(x), _p(out), _p(mask), x.numel(), float(p), int(seed), _stream())
        ctx.save_for_backward(mask)
        return out

    @staticmethod
    def backward(ctx, dout):
        (mask,) = ctx.saved_tensors
        dout = _f32c(dout)
        dx = torch.empty_like(dout)
        call("vocr_mul", _p(dout), _p(mask), _p(dx), dout.numel(), _stream())
        return dx, None, None


# ------------------------------------------------------------------------------------------------ packed sequence rows
def packed_row_count(lens, b):
    """Rows of the packed (chain-major) layout of include/vocr.h for a length-sorted batch: 4 * (sum_c L_c + chains + 1), L_c = lens[4c]."""
    nch = (b + 3) // 4
    return 4 * (sum(int(lens[4 * c]) for c in range(nch)) + nch + 1)


class SeqRowMaps(object):
    """Both index maps of a batch's packed layout (vocr_seq_rowmap), built once per forward on the device."""

    def __init__(self, lens_dev, lens_host, T, B):
        self.T, self.B = T, B
        self.rows = packed_row_count(lens_host, B)
        self.to_packed = torch.empty(T * B, dtype=torch.int32, device=lens_dev.device)
        self.to_dense = torch.empty(self.rows, dtype=torch.int32, device=lens_dev.device)
        call("vocr_seq_rowmap", _p(lens_dev), T, B, self.rows, _p(self.to_packed), _p(self.to_dense), _stream())


def gather_rows(src, row_map, nrows, fill=None):
    src = _f32c(src)
    n = src.shape[1]
    out = torch.empty(nrows, n, dtype=torch.float32, device=src.device)
    call("vocr_gather_rows", _p(src), _p(out), _p(row_map), nrows, n, _p(fill), _stream())
    return out


class GatherRowsFn(torch.autograd.Function):
    """Rows through an index map and back (packing: dense [T*B] -> packed rows; unpacking: the inverse, optionally with a fill row -
    the output layer's bias - where the packed layout has no frame).  The two maps are inverse to each other on the frames that
    exist, so the backward of one direction is the other direction with zeros for the rest."""

    @staticmethod
    def forward(ctx, x, fwd_map, bwd_map, nrows_out, fill=None):
        _need_gpu(x)
        ctx.bwd = (bwd_map, x.shape[0], fwd_map if fill is not None else None)
        return gather_rows(x, fwd_map, nrows_out, fill)

    @staticmethod
    def backward(ctx, dout):
        bwd_map, nrows_in, fill_map = ctx.bwd
        dfill = None
        if fill_map is not None and ctx.needs_input_grad[4]:
            # the fill row stood in every row without a source: its gradient is their column sum (what the dense path's bias gradient
            # collects from the padded frames; zero under the CTC criterion, not under any other loss on the returned logits)
            dout = _f32c(dout)
            n = dout.shape[1]
            dfill = torch.empty(n, dtype=torch.float32, device=dout.device)
            ws = _ws(_lib.load().vocr_gather_rows_fill_grad_workspace_bytes(n), dout.device)
            call("vocr_gather_rows_fill_grad", _p(dout), _p(fill_map), dout.shape[0], n, _p(dfill), _p(ws), _stream())
        return gather_rows(dout, bwd_map, nrows_in), None, None, None, dfill


# ------------------------------------------------------------------------------------------------ LSTM layer
class BiLstmLayerFn(torch.autograd.Function):
    """One bidirectional nn.LSTM layer on a packed batch (src/models/cnnlstm.py:148-149,288-290).
    x: [T*B, Din] time-major; returns y: [T*B, 2H] with zeros past each sequence's length.
    rows > 0: x and y are [rows, .] in the packed chain-major layout of include/vocr.h (no row for a padded frame of a whole chain)."""

    @staticmethod
    def forward(ctx, x, lens_dev, T, B, w_ih_f, w_hh_f, b_ih_f, b_hh_f, w_ih_r, w_hh_r, b_ih_r, b_hh_r, prep=None, direct_grads=True,
                drop_p=0.0, drop_seed=0, rows=0, pre=None, follow=None, drop_mask=None, x_bound=0.0):
        """drop_p > 0: nn.LSTM's inter-layer dropout on this layer's OUTPUT (counter-based mask of vocr_dropout_fwd, as DropoutFn) -
        here so that its backward can ride on the backward sweep's read of dy instead of being a pass of its own.  drop_mask: the same
        with a mask that already exists ([R, 2H], pre-scaled; explicit test masks, or vocr_dropout_mask's draw made ahead of the sweep).
        pre: (plane0, plane1), this layer's x-projection as the two source-direction planes a follower behind the layer below wrote
        (each [2, R, 4H]; x is then only kept for the backward).  follow: {"w_ih": (f, r) of the NEXT layer, "bias": its [2, 4H] bias
        sums or None, "wpack": its vocr_lstm_xproj_pack or None, "out": list} - the next layer's x-projection runs inside this layer's
        sweep (vocr_lstm_fwd_lead's follower waves) and its two planes are appended to follow["out"]."""
        _need_gpu(x, lens_dev, w_ih_f, w_hh_f, w_ih_r, w_hh_r)
        x = _f32c(x)
        lib = _lib.load()
        din = x.shape[1]
        H = w_hh_f.shape[1]
        dev = x.device
        R = int(rows) if rows else T * B              # rows of every sequence-side matrix of this layer
        assert x.shape[0] == R, (x.shape, R)
        xproj = torch.empty(2, R, 4 * H, dtype=torch.float32, device=dev) if pre is None else None
        ctx.wt = None
        if prep is not None and w_hh_f.data_ptr() in prep.lstm:
            prep.wait()
            bsum, wt_f, wt_r = prep.lstm[w_hh_f.data_ptr()]
            if wt_f is not None:
                ctx.wt = (wt_f, wt_r)
        else:
            bsum = torch.empty(2, 4 * H, dtype=torch.float32, device=dev)
            call("vocr_add", _p(b_ih_f), _p(b_hh_f), _p(bsum[0]), 4 * H, _stream())
            call("vocr_add", _p(b_ih_r), _p(b_hh_r), _p(bsum[1]), 4 * H, _stream())
        # packed rows: the sweeps leave the zero groups (and a last chain's rows >= B) alone - they must read as zeros downstream
        y = (torch.zeros if rows else torch.empty)(R, 2 * H, dtype=torch.float32, device=dev)
        gates = torch.empty(2, R, 4 * H, dtype=torch.float32, device=dev)
        cell = torch.empty(2, R, H, dtype=torch.float32, device=dev)
        ws = _ws(lib.vocr_lstm_workspace_bytes(T, B, H), dev)
        G = 4 * H

        ctx.x6 = x6_layer_ok(T, B, H, rows)
        ctx.x6_wt = None
        if pre is None and ctx.x6 and din >= 512:
            # bf16x6: the operands as three exact bf16 planes, six bf16 MFMAs per product (include/vocr.h: vocr_gemm_x6)
            if prep is not None and w_ih_f.data_ptr() in prep.x6w:
                wk, ctx.x6_wt = prep.x6w[w_ih_f.data_ptr()]
            else:
                wk, ctx.x6_wt = x6_weight_planes(w_ih_f, w_ih_r, torch.is_grad_enabled())
            xa = x6_planes(x, R, din, True, din, bound=x_bound)
            gemm_x6(xa, R, din, wk, 2 * G, din, R, 2 * G, din, xproj[0], G, c1=xproj[1], csplit=G, bias0=bsum[0], bias1=bsum[1])
        elif pre is None:
            # both directions' x-projections in one launch: 2 x (columns / 128) panels x row groups == the CU count
            gemm_pair(0, 0, 1, R, G, din, x, x, din, w_ih_f, w_ih_r, din, xproj[0], xproj[1], G, bias0=bsum[0], bias1=bsum[1])
        mask = drop_mask
        if follow is not None:
            # the NEXT layer's x-projection inside this layer's sweep (four more waves per workgroup: include/vocr.h, vocr_lstm_fwd_lead)
            nf, nr = follow["w_ih"]
            wpack = follow.get("wpack")
            if wpack is None:
                wpack = lstm_xproj_pack(nf, nr)
            nbias = follow.get("bias")
            if mask is None and drop_p and float(drop_p) > 0.0:
                mask = torch.empty_like(y)
                call("vocr_dropout_mask", _p(mask), mask.numel(), float(drop_p), int(drop_seed), _stream())
            planes = torch.empty(2, 2, R, G, dtype=torch.float32, device=dev)
            p0, p1 = (pre if pre is not None else (xproj, None))
            call("vocr_lstm_fwd_lead", _p(p0), _p(p1), _p(w_hh_f), _p(w_hh_r), _p(lens_dev), _p(y), _p(gates), _p(cell), _p(ws), T, B, H, R if rows else 0,
                 _p(wpack), _p(nbias), _p(mask), _p(planes), _p(health(dev)), _stream())
            follow["out"].append((planes[0], planes[1]))
        elif pre is not None:
            call("vocr_lstm_fwd_lead", _p(pre[0]), _p(pre[1]), _p(w_hh_f), _p(w_hh_r), _p(lens_dev), _p(y), _p(gates), _p(cell), _p(ws), T, B, H,
                 R if rows else 0, None, None, None, None, _p(health(dev)), _stream())
        elif rows:
            call("vocr_lstm_fwd_packed", _p(xproj), _p(w_hh_f), _p(w_hh_r), _p(lens_dev), _p(y), _p(gates), _p(cell), _p(ws), T, B, H, R,
                 _p(health(dev)), _stream())
        else:
            call("vocr_lstm_fwd", _p(xproj), _p(w_hh_f), _p(w_hh_r), _p(lens_dev), _p(y), _p(gates), _p(cell), _p(ws), T, B, H,
                 _p(health(dev)), _stream())
        ctx.dims = (T, B, H, din, int(rows))
        ctx.x_bound = float(x_bound)                # > 0: the caller's bound on |x| (the layer below's output, times its dropout scale): fp16x3 splits of x skip their maxima pass
        ctx.direct_grads = bool(direct_grads)       # False: the layer runs as several batch tiles, autograd adds their weight gradients
        ctx.b_refs = (b_ih_f, b_hh_f, b_ih_r, b_hh_r)
        out = y
        if mask is not None:
            # what the NEXT layer's weight gradient multiplies (the follower waves applied the mask on their own read of y)
            out = torch.empty_like(y)
            call("vocr_mul", _p(y), _p(mask), _p(out), y.numel(), _stream())
        elif drop_p and float(drop_p) > 0.0:
            out = torch.empty_like(y)
            mask = torch.empty_like(y)
            call("vocr_dropout_fwd", _p(y), _p(out), _p(mask), y.numel(), float(drop_p), int(drop_seed), _stream())

        ctx.has_mask = mask is not None
        ctx.save_for_backward(x, lens_dev, y, gates, cell, w_ih_f, w_hh_f, w_ih_r, w_hh_r, mask)
        return out

    @staticmethod
    @_ranged("bwd.bilstm")
    def backward(ctx, dy):
        x, lens_dev, y, gates, cell, w_ih_f, w_hh_f, w_ih_r, w_hh_r, mask = ctx.saved_tensors
        T, B, H, din, rows = ctx.dims
        R = rows if rows else T * B
        dy = _f32c(dy)
        lib = _lib.load()
        dev = x.device
        dg = (torch.zeros if rows else torch.empty)(2, R, 4 * H, dtype=torch.float32, device=dev)
        ws = _ws(lib.vocr_lstm_workspace_bytes(T, B, H), dev)
        if ctx.wt is not None:
            wt_f, wt_r = ctx.wt                                          # transposed at forward time on the side stream
        else:
            wt_f, wt_r = transpose2d(w_hh_f), transpose2d(w_hh_r)        # [H][4H]: contiguous B operand for the sweep
        dbias = torch.empty(2, 4 * H, dtype=torch.float32, device=dev)          # gradient of b_ih (= of b_hh), both directions
        # "parts" form (opt-in, VOCR_LSTM_BWD_PARTS=1; MEASURED SLOWER in the step, 17.70 vs 17.44 ms): the dropout mask rides on the
        # sweep's read of dy and the bias gradient's last reduction is left to the weight-gradient work on the side stream, so two
        # small kernels leave the critical path.  The sweeps then reach the chip ahead of the previous layer's weight-gradient GEMMs
        # (the LSTM-backward window ends 0.2 ms earlier), which pushes those GEMMs into the CNN backward, where a third MFMA kernel
        # beside the data- and weight-gradient convolutions costs 0.4 ms.  Bit-identical results either way (tests/test_round3_gpu.py).
        parts = not rows and bool(lib.vocr_lstm_bwd_parts_supported(T, B, H)) and _exp("VOCR_LSTM_BWD_PARTS", "0") == "1"
        if rows:
            # packed rows: the dropout mask rides on the sweep's read of dy (no pass of its own)
            call("vocr_lstm_bwd_packed", _p(dy), _p(mask), _p(wt_f), _p(wt_r), _p(lens_dev), _p(gates), _p(cell), _p(dg), _p(dbias), _p(ws),
                 T, B, H, R, _p(health(dev)), _stream())
        elif parts:
            call("vocr_lstm_bwd_parts", _p(dy), _p(mask), _p(wt_f), _p(wt_r), _p(lens_dev), _p(gates), _p(cell), _p(dg), _p(ws),
                 T, B, H, _p(health(dev)), _stream())
        else:
            if mask is not None:
                dym = torch.empty_like(dy)
                call("vocr_mul", _p(dy), _p(mask), _p(dym), dy.numel(), _stream())
                dy = dym
            call("vocr_lstm_bwd_bias", _p(dy), _p(wt_f), _p(wt_r), _p(lens_dev), _p(gates), _p(cell), _p(dg), _p(dbias), _p(ws),
                 T, B, H, _p(health(dev)), _stream())
        G = 4 * H
        # critical path first: dx feeds the layer below
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            # dx = dg_fwd W_ih_fwd + dg_rev W_ih_rev as ONE product whose K runs through both pairs (no accumulating second pass)
            if ctx.x6 and din >= 512:
                wt = ctx.x6_wt if ctx.x6_wt is not None else x6_weight_planes(w_ih_f, w_ih_r, True)[1]
                da = x6_planes(dg[0], R, 2 * G, True, G, x2=dg[1], seg=G, axis=0)
                gemm_x6(da, R, 2 * G, wt, din, 2 * G, R, din, 2 * G, dx, din)
            else:
                gemm_pair(1, 0, 0, R, din, G, dg[0], dg[1], G, w_ih_f, w_ih_r, din, dx, None, din)

        # weight gradients: if every parameter already owns a gradient buffer (FlatClampAdam aliases them into one
        # flat buffer), write them there from the side stream so they overlap the next layer's sweep; otherwise
        # return them to autograd on the current stream.
        params = (w_ih_f, w_hh_f, ctx.b_refs[0], ctx.b_refs[1], w_ih_r, w_hh_r, ctx.b_refs[2], ctx.b_refs[3])
        sinks = _sinks(params) if ctx.direct_grads else None
        direct = sinks is not None

        def weight_grads(outs, co=0):
            dwi_f, dwh_f, dbi_f, dbh_f, dwi_r, dwh_r, dbi_r, dbh_r = outs
            if parts:
                call("vocr_lstm_bias_from_parts", _p(dbias), _p(ws), T, B, H, _stream())
            sh = 4 if rows else B                        # rows between consecutive time steps of a sequence
            if ctx.x6:
                # bf16x6: the gate gradients of both directions as ONE transposed plane set (rows = gate columns, k = the frames; K is padded
                # with zeros to whole k16 steps) and the input's transpose
                k16 = (R + 15) // 16 * 16
                dgt = x6_planes(dg[0], 2 * G, R, False, G, x2=dg[1], seg=G, axis=1)
                xt = x6_planes(x, din, R, False, din, bound=ctx.x_bound)
                gemm_x6(dgt, 2 * G, R, xt, din, R, 2 * G, din, k16, dwi_f, din, c1=dwi_r, rsplit=G)
                if T > 1 and sh % 16 == 0 and (R - sh) % 16 == 0:
                    # the recurrent product's time shift (sh rows = sh / 16 k16 steps) is a k window of the same planes and of the output's transpose
                    yt = x6_planes(y, 2 * H, R, False, 2 * H, bound=1.0)       # |h| = |o tanh(c)| <= 1
                    if G % 256 == 0 and _X6_TWO_VIEWS:
                        # both directions in one launch: forward rows = dg_f[sh:]^T . y[:-sh, :H], reverse rows = dg_r[:-sh]^T . y[sh:, H:]
                        gemm_x6_two_views(dgt, 2 * G, R, yt, 2 * H, R, 2 * G, H, R - sh, G, (sh // 16, 0, 0), (0, H, sh // 16), dwh_f, dwh_r, H)
                    else:
                        gemm_x6(dgt, 2 * G, R, yt, 2 * H, R, G, H, R - sh, dwh_f, H, a_row0=0, a_kk0=sh // 16, b_row0=0, b_kk0=0)
                        gemm_x6(dgt, 2 * G, R, yt, 2 * H, R, G, H, R - sh, dwh_r, H, a_row0=G, a_kk0=0, b_row0=H, b_kk0=sh // 16)
                elif T > 1:
                    gemm_pair(co, 1, 0, G, H, R - sh, dg[0][sh:], dg[1], G, y, y[sh:, H:], 2 * H, dwh_f, dwh_r, H)
                else:
                    dwh_f.zero_()
                    dwh_r.zero_()
                dbi_f.copy_(dbias[0])
                dbi_r.copy_(dbias[1])
                dbh_f.copy_(dbias[0])
                dbh_r.copy_(dbias[1])
                return
            gemm_pair(co, 1, 0, G, din, R, dg[0], dg[1], G, x, x, din, dwi_f, dwi_r, din)
            if T > 1:
                # forward dir: h_{t-1} = y[t-1, :, :H];   reverse dir: h_{t+1} = y[t+1, :, H:]  (zero past the length).  Dense rows:
                # consecutive time steps are B rows apart; packed rows: 4 (the chain's groups), with an all-zero group on either side
                m = R - sh
                gemm_pair(co, 1, 0, G, H, m, dg[0][sh:], dg[1], G, y, y[sh:, H:], 2 * H, dwh_f, dwh_r, H)
            else:
                dwh_f.zero_()
                dwh_r.zero_()
            dbi_f.copy_(dbias[0])
            dbi_r.copy_(dbias[1])
            dbh_f.copy_(dbias[0])
            dbh_r.copy_(dbias[1])

        if direct and _SIDE_ENABLED and _exp("VOCR_LSTM_DW_OVERLAP", "1") == "1":
            side = side_stream()
            side.wait_stream(torch.cuda.current_stream())
            for t_ in (dg, x, y, dbias, ws):
                t_.record_stream(side)
            with torch.cuda.stream(side):
                weight_grads(sinks)                 # under the next layer's persistent sweep
            mark_side_pending()
            return (dx, None, None, None) + (None,) * 17
        if direct:
            weight_grads(sinks)
            return (dx, None, None, None) + (None,) * 17
        outs = [torch.empty_like(p) for p in params]
        weight_grads(outs)
        return (dx, None, None, None) + tuple(outs) + (None,) * 9


# ------------------------------------------------------------------------------------------------ CTC
class CtcFn(torch.autograd.Function):
    """Batch-summed CTC negative log-likelihood on pre-softmax activations, blank = 0."""

    @staticmethod
    def forward(ctx, logits, labels_dev, offsets_dev, label_lens_dev, act_lens_dev, max_label_len):
        _need_gpu(logits, labels_dev, offsets_dev, label_lens_dev, act_lens_dev)
        logits = _f32c(logits)
        T, B, V = logits.shape
        lib = _lib.load()
        dev = logits.device
        ws = _ws(lib.vocr_ctc_workspace_bytes(T, B, V, max_label_len), dev)
        nll = torch.empty(B, 1, dtype=torch.float32, device=dev)
        dlogits = torch.empty_like(logits)
        call("vocr_ctc_loss_grad", _p(logits), _p(labels_dev), _p(offsets_dev), _p(label_lens_dev), _p(act_lens_dev),
             _p(nll), _p(dlogits), _p(ws), T, B, V, int(max_label_len), _stream())
        loss = colsum(nll)                                   # shape (1,): sum over the batch, fixed order
        ctx.save_for_backward(dlogits)
        return loss

    @staticmethod
    @_ranged("bwd.ctc")
    def backward(ctx, dloss):
        (dlogits,) = ctx.saved_tensors
        dloss = _f32c(dloss)
        out = torch.empty_like(dlogits)
        call("vocr_scale_dev", _p(dlogits), _p(dloss), _p(out), dlogits.numel(), _stream())
        return out, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ decode / optimiser
def argmax_rows(logits):
    """(idx int32 [T,B], max float32 [T,B]) of logits [T,B,V]; first maximum wins."""
    _need_gpu(logits)
    logits = _f32c(logits)
    T, B, V = logits.shape
    idx = torch.empty(T, B, dtype=torch.int32, device=logits.device)
    mx = torch.empty(T, B, dtype=torch.float32, device=logits.device)
    call("vocr_argmax_rows", _p(logits), _p(idx), _p(mx), T * B, V, _stream())
    return idx, mx


def greedy_collapse(idx, mx, lens_dev, canon_dev, thresh):
    T, B = idx.shape
    labels = torch.zeros(B, T, dtype=torch.int32, device=idx.device)
    counts = torch.empty(B, dtype=torch.int32, device=idx.device)
    call("vocr_greedy_collapse", _p(idx), _p(mx), _p(lens_dev), _p(canon_dev), _p(labels), _p(counts), T, B, float(thresh),
         _stream())
    return labels, counts


def clamp_adam(p, g, m, v, lr, beta1, beta2, eps, weight_decay, clamp, grad_scale, step):
    _need_gpu(p, g, m, v)
    call("vocr_clamp_adam", _p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1), float(beta2), float(eps),
         float(weight_decay), float(clamp), float(grad_scale), int(step), _p(health(p.device)), _stream())


def clamp_(x, clamp):
    """In-place x.clamp_(-clamp, clamp) on the device, NaN preserved (train_cnn_lstm.py:143-145)."""
    _need_gpu(x)
    if not x.is_contiguous() or x.dtype != torch.float32:
        raise RuntimeError("vistaocr_amd.clamp_: need a contiguous fp32 tensor")
    call("vocr_clamp", _p(x), x.numel(), float(clamp), _p(health(x.device)), _stream())
    return x
