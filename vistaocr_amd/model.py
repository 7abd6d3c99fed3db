"""CnnOcrModel — drop-in for the reference's line recogniser (src/models/cnnlstm.py) whose arithmetic runs in
hand-written HIP kernels (libvocr.so).  Same constructor kwargs, same forward contract
`(x[B,C,H,W], widths[B]) -> (logits[T,B,V], lens int32[B] on CPU)`, same state_dict key names, same
FromSavedWeights / get_hyper_params / decode_without_lm / cnn_input_size_to_output_size surface.

Out of scope here: the WFST language-model decoders (init_lm / decode_with_lm*)."""
import logging
import math

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import prof_range
from .decoder import decode_greedy, greedy_label_sequences

logger = logging.getLogger("root")

# nn.Sequential slots of the reference's cnn (cnnlstm.py:124-134): conv at i, BatchNorm at i+1, ReLU at i+2
_CONV_SLOTS = ((0, None, 64), (3, 64, 64), "pool", (7, 64, 128), (10, 128, 128), "pool", (14, 128, 256),
               (17, 256, 256), (20, 256, 256))


class _ConvParams(nn.Module):
    """Parameter holder with nn.Conv2d's state_dict keys (weight, bias)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3))
        self.bias = nn.Parameter(torch.empty(cout))


class _BnParams(nn.Module):
    """Parameter/buffer holder with nn.BatchNorm2d's state_dict keys."""

    def __init__(self, c):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, 1e-5, 0.1
        self.weight = nn.Parameter(torch.empty(c))
        self.bias = nn.Parameter(torch.empty(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _LinearParams(nn.Module):
    def __init__(self, fin, fout):
        super().__init__()
        self.in_features, self.out_features = fin, fout
        self.weight = nn.Parameter(torch.empty(fout, fin))
        self.bias = nn.Parameter(torch.empty(fout))


class _LstmParams(nn.Module):
    """nn.LSTM's parameter names: weight_ih_l{k}[_reverse], weight_hh_l{k}[_reverse], bias_ih_.., bias_hh_.."""

    def __init__(self, input_size, hidden_size, num_layers, dropout):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers, self.dropout = input_size, hidden_size, num_layers, dropout
        self.bidirectional = True
        for l in range(num_layers):
            fin = input_size if l == 0 else 2 * hidden_size
            for sfx in ("", "_reverse"):
                self.register_parameter("weight_ih_l%d%s" % (l, sfx), nn.Parameter(torch.empty(4 * hidden_size, fin)))
                self.register_parameter("weight_hh_l%d%s" % (l, sfx), nn.Parameter(torch.empty(4 * hidden_size, hidden_size)))
                self.register_parameter("bias_ih_l%d%s" % (l, sfx), nn.Parameter(torch.empty(4 * hidden_size)))
                self.register_parameter("bias_hh_l%d%s" % (l, sfx), nn.Parameter(torch.empty(4 * hidden_size)))

    def layer(self, l, sfx):
        return (getattr(self, "weight_ih_l%d%s" % (l, sfx)), getattr(self, "weight_hh_l%d%s" % (l, sfx)),
                getattr(self, "bias_ih_l%d%s" % (l, sfx)), getattr(self, "bias_hh_l%d%s" % (l, sfx)))


class _Slots(nn.Module):
    """Container whose children are registered under explicit names (the reference's Sequential indices)."""

    def add(self, name, module):
        self.add_module(name, module)
        return module


class CnnOcrModel(nn.Module):
    def get_hyper_params(self):
        return self.hyper_params

    @classmethod
    def FromSavedWeights(cls, weight_file, verbose=True, gpu=None):
        """src/models/cnnlstm.py:40-71 — same checkpoint dict schema (written at train_cnn_lstm.py:427-438)."""
        from . import checkpoint
        weights = checkpoint.load(weight_file, map_location=lambda storage, loc: storage)
        if verbose:
            logger.info("Loading model from: %s" % weight_file)
            logger.info("\tFrom iteration: %d" % weights["iteration"])
            logger.info("\tWithout LM: Val CER: %.2f\tWER: %.2f" % (100 * weights["val_cer"], 100 * weights["val_wer"]))
            logger.info("\tModel Hyperparams = %s" % str(weights["model_hyper_params"]))
        hp = weights["model_hyper_params"]
        if gpu is not None:
            hp["gpu"] = gpu
        hp["verbose"] = verbose
        model = cls(**hp)
        model.rtl = weights["rtl"] if "rtl" in weights else True
        sd = weights["state_dict"]
        # checkpoints written with multigpu=True carry DataParallel's 'cnn.module.' prefix (utils/decode.py:58-71)
        sd = checkpoint.strip_dataparallel_prefix(sd)
        model.load_state_dict(sd, strict=True)
        return model

    def __init__(self, *args, **kwargs):
        super(CnnOcrModel, self).__init__()
        if len(args) > 0:
            raise Exception("Only keyword arguments allowed in CnnOcrModel")
        self.hyper_params = kwargs.copy()
        self.input_line_height = kwargs["input_line_height"]
        self.rds_line_height = kwargs["rds_line_height"]
        self.alphabet = kwargs["alphabet"]
        self.lstm_input_dim = kwargs["lstm_input_dim"]
        self.num_lstm_layers = kwargs["num_lstm_layers"]
        self.num_lstm_hidden_units = kwargs["num_lstm_hidden_units"]
        self.p_lstm_dropout = kwargs["p_lstm_dropout"]
        self.num_in_channels = kwargs.get("num_in_channels", 1)
        self.gpu = kwargs.get("gpu", True)
        self.multigpu = kwargs.get("multigpu", True)     # accepted for compatibility; DP is process-level (train.py)
        self.verbose = kwargs.get("verbose", True)
        # extension (BASELINE config 5): 'fp16' rounds conv operands to fp16 for the matrix cores, fp32 accumulate
        self.conv_dtype = kwargs.get("conv_dtype", "fp32")
        if self.conv_dtype not in ("fp32", "fp16"):
            raise Exception("conv_dtype must be 'fp32' or 'fp16'")
        self.lattice_decoder = None

        if self.rds_line_height > self.input_line_height:
            raise Exception("rapid-downsample line height must be less than or equal to input line height")
        if self.input_line_height % self.rds_line_height != 0:
            raise Exception("rapid-downsample line height must evenly divide input line height by a power of 2")
        num_rds = 0
        lh = self.input_line_height
        while lh > self.rds_line_height:
            num_rds += 1
            if lh % 2 != 0:
                raise Exception("rapid-downsample line height must eenly diide input line height by a power of 2")
            lh /= 2
        if lh != self.rds_line_height:
            raise Exception("rapid-downsample line height must eenly diide input line height by a power of 2")
        if self.num_lstm_hidden_units % 16 != 0:
            raise Exception("vistaocr_amd: num_lstm_hidden_units must be a multiple of 16 (MFMA 16x16x4 tiling)")
        self.num_rds_layers = num_rds

        self.rapid_ds = _Slots()
        last = self.num_in_channels
        for i in range(num_rds):
            self.rapid_ds.add("%02d-conv" % i, _ConvParams(last, 16))
            last = 16
        self.cnn = _Slots()
        self._plan = []
        for slot in _CONV_SLOTS:
            if slot == "pool":
                self._plan.append("pool")
                continue
            idx, cin, cout = slot
            cin = last if cin is None else cin
            conv = self.cnn.add(str(idx), _ConvParams(cin, cout))
            bn = self.cnn.add(str(idx + 1), _BnParams(cout))
            self._plan.append((conv, bn))

        cnn_out_h, _ = self.cnn_input_size_to_output_size((self.input_line_height, 20))
        cnn_feat_size = 256 * cnn_out_h
        self.bridge_layer = _Slots()
        self.bridge_layer.add("0", _LinearParams(cnn_feat_size, self.lstm_input_dim))
        self.lstm = _LstmParams(self.lstm_input_dim, self.num_lstm_hidden_units, self.num_lstm_layers, self.p_lstm_dropout)
        self.prob_layer = _Slots()
        self.prob_layer.add("0", _LinearParams(2 * self.num_lstm_hidden_units, len(self.alphabet)))

        for param in self.parameters():                      # cnnlstm.py:158-159
            torch.nn.init.uniform_(param, -0.08, 0.08)

        # explicit randomness hooks (parity tests): FractionalMaxPool samples and inter-layer dropout masks
        self.pool_samples = None          # [u1 (B,64,2), u2 (B,128,2)] or None -> torch.rand per forward
        self.dropout_masks = None         # list of [T,B,2H] pre-scaled masks or None -> drawn on device
        self._dropout_calls = 0
        self.dropout_seed = 0x5EED
        # packed sequence rows (forward(): "pack_padded_sequence's economy"): on by default for a batch whose packed row count is at
        # most `pack_threshold` of T*B (below that the two row gathers cost more than the GEMM rows they save)
        self.pack_sequences = True
        self.pack_threshold = 0.9
        # backward milestones of THIS model (train.make_optimizer registers the data-parallel bucket start here)
        self._vocr_hooks = {"sequence_grads_ready": []}

        if self.verbose:
            total = sum(p.numel() for p in self.parameters())
            logger.info("Total Model Params = %d" % total)
            logger.info("\tCNN Params = %d" % sum(p.numel() for p in self.cnn.parameters()))
            logger.info("\tLSTM Params = %d" % sum(p.numel() for p in self.lstm.parameters()))
        if torch.cuda.is_available() and self.gpu:
            self.cuda()
        else:
            logger.info("Warning: model built on CPU; forward needs the MI355X HIP path (no CPU fallback)")

    # ------------------------------------------------------------------ size bookkeeping (cnnlstm.py:211-260)
    def cnn_output_num_channels(self):
        return 256

    def cnn_input_size_to_output_size(self, in_size):
        out_h, out_w = in_size
        for _ in range(self.num_rds_layers):
            out_h = math.floor((out_h + 2.0 * 1 - 1 * (3 - 1) - 1) / 1 + 1)      # conv k3 p1
            out_w = math.floor((out_w + 2.0 * 1 - 1 * (3 - 1) - 1) / 1 + 1)
            out_h = math.floor((out_h + 2.0 * 0 - 1 * (2 - 1) - 1) / 2 + 1)      # MaxPool2d(2, 2)
            out_w = math.floor((out_w + 2.0 * 0 - 1 * (2 - 1) - 1) / 2 + 1)
        for step in self._plan:
            if step == "pool":
                out_h, out_w = math.floor(out_h * 0.5), math.floor(out_w * 0.7)
            else:
                out_h = math.floor((out_h + 2.0 * 1 - 1 * (3 - 1) - 1) / 1 + 1)
                out_w = math.floor((out_w + 2.0 * 1 - 1 * (3 - 1) - 1) / 1 + 1)
        return (out_h, out_w)

    def output_lengths(self, widths):
        """Frames per line the forward pass will report for these pixel widths (cnnlstm.py:281-283), on the host."""
        widths = widths.data if torch.is_tensor(widths) else widths
        return torch.tensor([self.cnn_input_size_to_output_size((self.input_line_height, int(wd)))[1] for wd in widths], dtype=torch.int32)

    # ------------------------------------------------------------------ forward (cnnlstm.py:268-296)
    def forward(self, x, actual_minibatch_widths):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("vistaocr_amd.CnnOcrModel.forward needs the model on an MI355X (model.cuda()); "
                               "there is no CPU fallback")
        x = x.to(dev, non_blocking=True)
        # weight-only preparation (conv weight packs - fp32 or fp16 form -, LSTM bias sums and W_hh transposes) on the idle side stream;
        # every layer waits for its own pack's event only
        f16 = self.conv_dtype == "fp16"
        convs = [st[0].weight for st in self._plan if st != "pool"]
        convs = [w_ for w_ in convs if w_.shape[1] > 1]             # a one-input-channel layer runs straight on its weight
        layers = [(self.lstm.layer(l, "")[1], self.lstm.layer(l, "")[2], self.lstm.layer(l, "")[3],
                   self.lstm.layer(l, "_reverse")[1], self.lstm.layer(l, "_reverse")[2], self.lstm.layer(l, "_reverse")[3])
                  for l in range(self.num_lstm_layers)]
        # layers whose x-projection runs BESIDE the forward sweep of the layer below (ops.BiLstmLayerFn's `follow`): decided per batch
        # below (the sweep kind depends on the batch size); their W_ih packs are weight-only work like the rest of the preparation
        H = self.num_lstm_hidden_units
        follow_w = [(self.lstm.layer(l, "")[0], self.lstm.layer(l, "_reverse")[0]) for l in range(1, self.num_lstm_layers)]
        may_follow = ops.lstm_follow_ok(x.shape[0], H, 2 * H, H)
        x6_w = [(self.lstm.layer(l, "")[0], self.lstm.layer(l, "_reverse")[0]) for l in range(self.num_lstm_layers)] \
            if ops.x6_layer_ok(1 << 10, x.shape[0], H, 0) else []
        prep = ops.forward_prep(convs, layers, torch.is_grad_enabled(), f16, follow_w if may_follow else (), x6_w)
        a = x
        for i in range(self.num_rds_layers):
            conv = getattr(self.rapid_ds, "%02d-conv" % i)
            a = ops.ConvReluPoolFn.apply(a, conv.weight, conv.bias, self.conv_dtype == "fp16")
        pool_i = 0
        fused_pool = False
        with prof_range("model.cnn"):
            for si, step in enumerate(self._plan):
                if step == "pool":
                    if fused_pool:                      # already applied inside the conv layer before it
                        fused_pool = False
                        continue
                    n, c, h, w = a.shape
                    oh, ow = math.floor(h * 0.5), math.floor(w * 0.7)
                    if self.pool_samples is not None:
                        u = self.pool_samples[pool_i].to(dev)
                    else:
                        u = torch.rand(n, c, 2, dtype=torch.float32, device=dev)
                    a = ops.FracPoolFn.apply(a, u, oh, ow)
                    pool_i += 1
                    continue
                conv, bn = step
                u, oh, ow = None, 0, 0
                if si + 1 < len(self._plan) and self._plan[si + 1] == "pool":
                    # the pooling layer that follows is fused into this layer's BatchNorm + ReLU pass
                    n, h, w = a.shape[0], a.shape[2], a.shape[3]
                    oh, ow = math.floor(h * 0.5), math.floor(w * 0.7)
                    if self.pool_samples is not None:
                        u = self.pool_samples[pool_i].to(dev)
                    else:
                        u = torch.rand(n, conv.weight.shape[0], 2, dtype=torch.float32, device=dev)
                    pool_i += 1
                    fused_pool = True
                a = ops.ConvBnReluFn.apply(a, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                           self.training, bn.eps, bn.momentum, self.conv_dtype == "fp16", u, oh, ow,
                                           bn.num_batches_tracked if self.training else None, prep)
        b, c, h, w = a.shape
        with prof_range("model.bridge"):
            feat = ops.PermuteBchwToWbchFn.apply(a, self._vocr_hooks)                                   # [w*b, c*h]
            br = getattr(self.bridge_layer, "0")
            lstm_in = ops.LinearFn.apply(feat, br.weight, br.bias, True)              # [w*b, D]

        widths = actual_minibatch_widths.data if torch.is_tensor(actual_minibatch_widths) else actual_minibatch_widths
        out_w = [self.cnn_input_size_to_output_size((self.input_line_height, int(wd)))[1] for wd in widths]
        for i in range(1, len(out_w)):
            if out_w[i] > out_w[i - 1]:     # pack_padded_sequence(enforce_sorted=True) precondition
                raise RuntimeError("`lengths` array must be sorted in decreasing order (widths must be sorted descending)")
        if len(out_w) != b or out_w[-1] <= 0:
            raise RuntimeError("bad widths for batch of %d: %s" % (b, out_w))
        T = out_w[0]
        if T > w:
            raise RuntimeError("width %d implies %d frames but the CNN produced %d" % (int(widths[0]), T, w))
        lens_cpu = torch.tensor(out_w, dtype=torch.int32)
        lens_dev = lens_cpu.to(dev, non_blocking=True)

        hseq = lstm_in[: T * b]
        # pack_padded_sequence's economy (cnnlstm.py:285-290): when enough of the padded [T, b] frames are padding, the whole LSTM stack
        # runs on the packed chain-major rows of include/vocr.h - every projection / gradient GEMM over the packed rows only, a chain's
        # workgroups leaving the sweeps after their own longest row - and the logits are unpacked at the end (a padded frame's logits are
        # the output layer's bias, exactly what the dense path computes from its all-zero LSTM output).  A uniform batch stays dense.
        maps = None
        if self.pack_sequences and b <= 32 and ops.packed_row_count(out_w, b) <= self.pack_threshold * T * b \
                and ops.packed_row_count(out_w, b) * 2 * self.num_lstm_hidden_units * 4 < 2 ** 31 \
                and _lib.load().vocr_lstm_packed_supported(b, self.num_lstm_hidden_units):
            maps = ops.SeqRowMaps(lens_dev, out_w, T, b)
            hseq = ops.GatherRowsFn.apply(hseq, maps.to_dense, maps.to_packed, maps.rows)
        rows = maps.rows if maps is not None else 0
        drop = self.lstm.training and self.p_lstm_dropout > 0      # nn.LSTM reads its own .training flag
        following = may_follow and 16 < b <= 32
        pre = None
        x_bound = 0.0          # a bound on |hseq| where one is known: an LSTM layer's output lies in [-1, 1], times the drawn dropout's scale (fp16x3 splits use it)
        for l in range(self.num_lstm_layers):
            f = self.lstm.layer(l, "")
            r = self.lstm.layer(l, "_reverse")
            # nn.LSTM's inter-layer dropout (every layer's output but the last's); drawn inside the layer op when it is one op
            draw = drop and l < self.num_lstm_layers - 1 and self.dropout_masks is None
            if draw:
                self._dropout_calls += 1
            last = l == self.num_lstm_layers - 1
            if b <= 64:
                # one op per layer.  The inter-layer dropout rides inside it: an explicit mask, or - on packed rows - the drawn mask of the
                # DENSE frame order moved through the row map, so that a frame's draw does not depend on whether the batch was packed
                fol, mk = None, None
                if not last and self.dropout_masks is not None:
                    mk = self.dropout_masks[l].to(dev).reshape(T * b, -1)
                    if maps is not None:
                        mk = ops.gather_rows(mk, maps.to_dense, maps.rows)
                elif draw and maps is not None:
                    mk = ops.gather_rows(ops.dropout_mask(T * b, 2 * H, self.p_lstm_dropout, self.dropout_seed + self._dropout_calls, dev),
                                         maps.to_dense, maps.rows)
                if following and not last:
                    # layer l + 1's x-projection inside this layer's sweep (ops.BiLstmLayerFn's `follow`)
                    nf, nr = self.lstm.layer(l + 1, ""), self.lstm.layer(l + 1, "_reverse")
                    key = nf[1].data_ptr()
                    if prep is not None:
                        prep.wait()
                    fol = {"w_ih": (nf[0], nr[0]), "out": [],
                           "bias": prep.lstm[key][0] if prep is not None and key in prep.lstm else ops.lstm_bias_sum(nf[2], nf[3], nr[2], nr[3]),
                           "wpack": prep.xpack.get(nf[0].data_ptr()) if prep is not None else None}
                with prof_range("model.lstm.l%d" % l):
                    dp, ds = (self.p_lstm_dropout, self.dropout_seed + self._dropout_calls) if draw and mk is None else (0.0, 0)
                    hseq = ops.BiLstmLayerFn.apply(hseq, lens_dev, T, b, f[0], f[1], f[2], f[3], r[0], r[1], r[2], r[3], prep, True, dp, ds, rows,
                                                   pre, fol, mk, x_bound)
                    # what the next layer reads: this layer's output, times 1 / (1 - p) where a mask we drew ourselves was applied (an explicit mask: unknown)
                    x_bound = 0.0 if (mk is not None and self.dropout_masks is not None) else \
                        (1.0 / (1.0 - self.p_lstm_dropout) if (dp > 0 or mk is not None) else 1.0)
                pre = fol["out"][0] if fol is not None else None
                continue
            with prof_range("model.lstm.l%d" % l):
                hseq = self._bilstm_layer(hseq, lens_dev, out_w, T, b, f, r, prep)
            if not last:
                if self.dropout_masks is not None:
                    hseq = ops.MulMaskFn.apply(hseq, self.dropout_masks[l].to(dev).reshape(T * b, -1))
                elif draw:
                    hseq = ops.DropoutFn.apply(hseq, self.p_lstm_dropout, self.dropout_seed + self._dropout_calls)
        pr = getattr(self.prob_layer, "0")
        with prof_range("model.prob"):
            prob_output = ops.LinearFn.apply(hseq, pr.weight, pr.bias, False)
            if maps is not None:
                prob_output = ops.GatherRowsFn.apply(prob_output, maps.to_packed, maps.to_dense, T * b, pr.bias)
            prob_output = prob_output.view(T, b, -1)
        return prob_output, lens_cpu

    def _bilstm_layer(self, hseq, lens_dev, out_w, T, b, f, r, prep):
        """One bidirectional layer of a batch of MORE than 64 rows.  The sweep kernels take up to 64 batch rows per call (include/vocr.h);
        the reference's --batch-size is free (src/train_cnn_lstm.py:155), so a larger batch runs as tiles of <= 64 rows: the recurrence
        never couples batch rows, the widths are sorted, so a tile is itself a valid packed batch and only sweeps its own longest
        sequence.  A tile's weight gradients go through autograd (which adds the tiles' contributions) instead of the direct sinks."""
        ntile = (b + 63) // 64
        rows = (b + ntile - 1) // ntile
        h3 = hseq.view(T, b, -1)
        outs = []
        for b0 in range(0, b, rows):
            b1 = min(b, b0 + rows)
            tt = int(out_w[b0])
            xt = h3[:tt, b0:b1].reshape(tt * (b1 - b0), h3.shape[2])
            yt = ops.BiLstmLayerFn.apply(xt, lens_dev[b0:b1], tt, b1 - b0, f[0], f[1], f[2], f[3], r[0], r[1], r[2], r[3], prep, False)
            yt = yt.view(tt, b1 - b0, -1)
            if tt < T:
                yt = torch.cat([yt, yt.new_zeros(T - tt, b1 - b0, yt.shape[2])], dim=0)
            outs.append(yt)
        return torch.cat(outs, dim=1).reshape(T * b, -1)

    # ------------------------------------------------------------------ decode (cnnlstm.py:479-541)
    def decode_without_lm(self, model_output, batch_actual_timesteps, uxxxx=False):
        return decode_greedy(model_output, batch_actual_timesteps, self.alphabet, uxxxx=uxxxx)

    def decode_labels(self, model_output, batch_actual_timesteps):
        """Integer label sequences of the greedy decode (the parity quantity of BASELINE.json)."""
        return greedy_label_sequences(model_output, batch_actual_timesteps, self.alphabet)[1]

    def init_lm(self, *a, **k):
        raise NotImplementedError("LM (eesen WFST) decoding is outside the hot path (SURVEY.md §2 row 14)")

    decode_with_lm = decode_with_lm_mt = init_lm
