"""Host-side shell around the hot path: the pieces of src/train_cnn_lstm.py,
src/lr_scheduler.py and src/datautils.py that drive train()/test_on_val() — restated so a user of the reference finds
the same control flow: width-grouped sampling, sort-by-width collation, plateau LR schedule with "reload best on LR
drop", validation with CER/WER, and the reference's checkpoint dictionary.  Pure host logic; the arithmetic stays in
the HIP path (model / criterion / decoder)."""
import itertools
import logging
import shutil
import warnings

import numpy as np
import torch

from .textutils import compute_cer_wer, form_target_transcription

logger = logging.getLogger("root")


class ReduceLROnPlateau:
    """The plateau schedule the reference trains with (behaviour of src/lr_scheduler.py:10-98, pinned by the traces in
    tests/golden/host_logic.npz): a metric that has not improved on the best one by more than `epsilon` for more than `patience`
    calls lowers every group's lr by `factor` (not below `min_lr`) and opens a `cooldown` window in which nothing is counted.
    step(metric) -> True when the lr was lowered on this call - or could not be lowered any further, which sets `.finished`.
    Works with any optimiser exposing .param_groups (FlatClampAdam, torch.optim.*)."""

    def __init__(self, optimizer, mode="min", factor=0.1, patience=10, epsilon=1e-4, cooldown=0, min_lr=0):
        if not hasattr(optimizer, "param_groups"):
            raise AssertionError("optimizer has no param_groups")
        if factor >= 1.0:
            raise ValueError("ReduceLROnPlateau does not support a factor>=1.0")
        if mode not in ("min", "max"):
            raise RuntimeError("ReduceLROnPlateau mode:%s not valid." % mode)
        self.optimizer, self.mode = optimizer, mode
        self.factor, self.patience, self.epsilon, self.cooldown, self.min_lr = factor, patience, epsilon, cooldown, min_lr
        self.finished = False
        self.reset()

    def reset(self):
        """Forget the best metric and every counter (the lr stays where it is)."""
        self.best = float("inf") if self.mode == "min" else -float("inf")
        self.wait = 0                   # calls without improvement since the last improvement / lr drop
        self.cooldown_counter = 0       # calls left in the window after an lr drop

    def in_cooldown(self):
        return self.cooldown_counter > 0

    def _improves(self, metric):
        # "max" compares the way the reference does (metric < best + epsilon): kept, its traces are the contract
        return metric < (self.best - self.epsilon if self.mode == "min" else self.best + self.epsilon)

    def _lower(self):
        floor = self.min_lr + self.min_lr * 1e-4
        for group in self.optimizer.param_groups:
            lr = float(group["lr"])
            if lr > floor:
                group["lr"] = max(lr * self.factor, self.min_lr)
                self.cooldown_counter, self.wait = self.cooldown, 0
            else:
                self.finished = True

    def step(self, metrics):
        if metrics is None:
            warnings.warn("Learning Rate Plateau Reducing requires metrics.", RuntimeWarning)
            return False
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.wait = 0
        if self._improves(metrics):
            self.best, self.wait = metrics, 0
            return False
        if self.cooldown_counter > 0:
            return False
        if self.wait < self.patience:
            self.wait += 1
            return False
        self._lower()
        return True


class GroupedSampler(torch.utils.data.Sampler):
    """Width-grouped sampling (behaviour of src/datautils.py:4-51): the groups of `data_source.size_group_keys` one after the other,
    the members of a group in a fresh random order (`rand`), in one random order drawn once per group (`fixed_rand`) or as stored;
    at most `max_items` indices when that is positive.  A group's permutation is drawn when the iteration reaches the group, so the
    torch RNG is consumed exactly as the reference consumes it."""

    def __init__(self, data_source, rand=True, max_items=-1, fixed_rand=False):
        self.size_group_keys = data_source.size_group_keys
        self.size_groups = data_source.size_groups
        self.num_samples = len(data_source)
        self.rand, self.fixed_rand, self.max_items = rand, fixed_rand, max_items
        self.rand_perm = {}

    def _visit_order(self, key):
        n = len(self.size_groups[key])
        if self.fixed_rand:
            if key not in self.rand_perm:
                self.rand_perm[key] = torch.randperm(n).long()
            return self.rand_perm[key].tolist()
        return torch.randperm(n).long().tolist() if self.rand else range(n)

    def __iter__(self):
        members = (self.size_groups[key][pos] for key in self.size_group_keys if len(self.size_groups[key]) for pos in self._visit_order(key))
        return itertools.islice(members, self.max_items) if self.max_items > 0 else members

    def __len__(self):
        return self.num_samples


def SortByWidthCollater(batch):
    """src/datautils.py:54-176 (non-seq2seq, non-bylang path): sort by metadata['width'] descending, zero-pad the
    images to the widest, unpadded widths as IntTensor, flat IntTensor targets, target lengths, metadata lists."""
    batch = sorted(batch, key=lambda d: d[-1]["width"], reverse=True)
    biggest = batch[0][0].size()
    x = torch.zeros(len(batch), biggest[0], biggest[1], biggest[2])
    widths = torch.IntTensor(len(batch))
    target_lens = torch.IntTensor(len(batch))
    writer_ids = torch.LongTensor(len(batch))
    ids, have_writers = [], False
    for i, (tensor, transcript, md) in enumerate(batch):
        x[i, :, :, :tensor.size(2)] = tensor
        widths[i] = md["width"]
        target_lens[i] = len(transcript)
        if "writer-id" in md:
            writer_ids[i] = md["writer-id"]
            have_writers = True
        if "utt-id" in md:
            ids.append(md["utt-id"])
    targets = torch.IntTensor(int(target_lens.sum().item()))
    o = 0
    for _, transcript, _ in batch:
        for ch in transcript:
            targets[o] = ch
            o += 1
    meta = {}
    if have_writers:
        meta["writer-ids"] = writer_ids
    if ids:
        meta["utt-ids"] = ids
    return x, targets, widths, target_lens, meta


def _check_device_health(model):
    from . import ops
    dev = next(model.parameters()).device
    if dev.type == "cuda":
        ops.check_health_sync(dev)


def test_on_val(val_dataloader, model, criterion):
    """src/train_cnn_lstm.py:32-100 — running means of loss/batch, CER and WER over the validation loader."""
    cer_avg = wer_avg = loss_avg = 0.0
    n = 0
    model.eval()
    with torch.no_grad():
        for x, target, widths, target_lens, _ in val_dataloader:
            out, lens = model(x.cuda(non_blocking=True), widths)
            loss = criterion(out, target, lens, target_lens)
            hyps = model.decode_without_lm(out, lens, uxxxx=True)
            bsz = x.size(0)
            n += 1
            loss_avg += (float(loss) / bsz - loss_avg) / n
            o = 0
            bc = bw = 0.0
            tnp = target.numpy()
            for i, hyp in enumerate(hyps):
                L = int(target_lens[i])
                ref = form_target_transcription(tnp[o:o + L], model.alphabet)
                o += L
                c, w = compute_cer_wer(hyp, ref)
                bc += c
                bw += w
            cer_avg += (bc / bsz - cer_avg) / n
            wer_avg += (bw / bsz - wer_avg) / n
    model.train()
    _check_device_health(model)          # a timed-out LSTM sweep poisons its logits with NaN: never report them as a score
    return loss_avg, cer_avg, wer_avg


def save_snapshot(path, iteration, model, optimizer, rtl, cur_lr, val_loss, val_cer, val_wer, line_height):
    """The checkpoint dictionary of src/train_cnn_lstm.py:427-438, readable by this build's and by the reference's
    FromSavedWeights: the alphabet is pickled under the reference's class path (checkpoint.save) and, for a model built
    with the reference's default multigpu=True, the CNN keys carry nn.DataParallel's 'cnn.module.' prefix the reference's
    strict load expects (this build strips it again on load).  `optimizer` state is whatever the optimiser reports."""
    from . import checkpoint
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    hp = dict(model.get_hyper_params())
    if hp.get("multigpu", True) and hp.get("gpu", True):
        sd = checkpoint.add_dataparallel_prefix(sd)
    checkpoint.save({"iteration": iteration, "state_dict": sd, "optimizer": optimizer.state_dict(), "model_hyper_params": hp,
                     "rtl": rtl, "cur_lr": cur_lr, "val_loss": val_loss, "val_cer": val_cer, "val_wer": val_wer,
                     "line_height": line_height}, path)


def fit(model, criterion, optimizer, train_dataloader, validation_dataloader, train_fn, snapshot_prefix, batch_size,
        n_epochs=1, snapshot_every_n_iterations=1000, patience=10, min_lr=1e-7, rtl=False, line_height=30,
        validate_fn=test_on_val):
    """The loop of src/train_cnn_lstm.py:375-470: train, validate every N iterations, plateau schedule on val WER,
    snapshot, copy to best, reload best after an LR drop, stop when the schedule is exhausted.  Returns a history dict."""
    snapshot_path = snapshot_prefix + "-cur_snapshot.pth"
    best_model_path = snapshot_prefix + "-best_model.pth"
    scheduler = ReduceLROnPlateau(optimizer, mode="min", patience=patience, min_lr=min_lr)
    lr_alpha = float(optimizer.param_groups[0]["lr"])
    hist = dict(loss=[], val=[], lr_drops=[], stopped_early=False)
    # The model, optimiser and loaders exist by now: take them out of the cyclic garbage collector's sight.  Its first full collection
    # would otherwise walk every object of the process some 20 steps into training and stall the loop for 55-75 ms (three to four
    # steps of the 18.5 ms BASELINE step); garbage created by the steps themselves is still collected.
    import gc
    gc.collect()
    gc.freeze()
    try:
        return _fit_loop(model, criterion, optimizer, train_dataloader, validation_dataloader, train_fn, snapshot_path, best_model_path,
                         batch_size, n_epochs, snapshot_every_n_iterations, rtl, line_height, validate_fn, scheduler, lr_alpha, hist)
    finally:
        gc.unfreeze()       # the freeze is this call's, not the process's: a host that calls fit() again, or lives on, collects as before


def _fit_loop(model, criterion, optimizer, train_dataloader, validation_dataloader, train_fn, snapshot_path, best_model_path, batch_size,
              n_epochs, snapshot_every_n_iterations, rtl, line_height, validate_fn, scheduler, lr_alpha, hist):
    iteration = 0
    best_val_wer = float("inf")
    for _epoch in range(1, n_epochs + 1):
        for batch in train_dataloader:
            iteration += 1
            hist["loss"].append(train_fn(batch, model, criterion, optimizer) / batch_size)
            if iteration % snapshot_every_n_iterations != 0:
                continue
            val_loss, val_cer, val_wer = validate_fn(validation_dataloader, model, criterion)
            early_exit = lowered = False
            if scheduler.step(val_wer):
                lowered = True
                hist["lr_drops"].append(iteration)
                early_exit = scheduler.finished
                lr_alpha = max(lr_alpha * scheduler.factor, scheduler.min_lr)
            hist["val"].append((iteration, val_loss, val_cer, val_wer))
            save_snapshot(snapshot_path, iteration, model, optimizer, rtl, lr_alpha, val_loss, val_cer, val_wer, line_height)
            if val_wer < best_val_wer:
                best_val_wer = val_wer
                shutil.copyfile(snapshot_path, best_model_path)
            if early_exit:
                hist["stopped_early"] = True
                return hist
            if lowered:
                from . import checkpoint
                weights = checkpoint.load(best_model_path, map_location="cpu")
                model.load_state_dict(checkpoint.strip_dataparallel_prefix(weights["state_dict"]))
    return hist


def decode_dataset(model, dataloader, outdir, visual_to_logical=None, seed=7):
    """The inference driver of src/decode_testset.py:42-206 without the LM branch: forward every batch, greedy-decode,
    and write `hyp-chars.txt` ("<uxxxx ...> (<utt-id>)") and `hyp-chars.txt.utf8` ("<utf8> (<utt-id minus last _part>)").
    The reference runs the decode in a background process because its per-frame numpy argmax is slow; here the argmax
    is a GPU kernel and one small D2H copy, so decode runs inline.  `visual_to_logical` stands in for the ICU bidi step
    (`utf8_visual_to_logical`, src/textutils.py:186-212; identity for left-to-right scripts).  Seeds like the reference
    (FractionalMaxPool draws samples in eval too, decode_testset.py:70-71).  Returns the number of lines written."""
    import os
    from .textutils import utf8_to_uxxxx
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    os.makedirs(outdir, exist_ok=True)
    model.eval()
    n = 0
    with torch.no_grad(), open(os.path.join(outdir, "hyp-chars.txt"), "w") as fh, \
            open(os.path.join(outdir, "hyp-chars.txt.utf8"), "w") as fh8:
        for x, _target, widths, _target_lens, meta in dataloader:
            out, lens = model(x.cuda(non_blocking=True), widths)
            hyps = model.decode_without_lm(out, lens, uxxxx=False)
            for i, hyp in enumerate(hyps):
                hyp_utf8 = visual_to_logical(hyp) if visual_to_logical is not None else hyp
                uttid = meta["utt-ids"][i]
                fh.write("%s (%s)\n" % (utf8_to_uxxxx(hyp_utf8), uttid))
                fh8.write("%s (%s)\n" % (hyp_utf8, uttid[:uttid.rfind("_")]))
                n += 1
    _check_device_health(model)
    return n
