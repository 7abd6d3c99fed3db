"""CTCLoss — stands in for `warpctc_pytorch.CTCLoss` at the reference's call sites
(src/train_cnn_lstm.py:12,358,52,138): callable
    criterion(logits[T,B,V] (pre-softmax, on the GPU), targets IntTensor[sum L] (CPU), act_lens IntTensor[B] (CPU),
              target_lens IntTensor[B] (CPU)) -> Tensor of shape (1,) = batch-SUMMED negative log-likelihood
supporting .backward() (gradient w.r.t. the logits) and .cuda().  blank = 0."""
import torch
import torch.nn as nn

from . import ops


class CTCLoss(nn.Module):
    def __init__(self, size_average=False, length_average=False):
        super().__init__()
        if size_average or length_average:
            raise NotImplementedError("the reference uses CTCLoss() with the default batch-summed cost")

    def forward(self, acts, labels, act_lens, label_lens):
        if not acts.is_cuda:
            raise RuntimeError("vistaocr_amd.CTCLoss needs activations on the MI355X; there is no CPU fallback")
        dev = acts.device
        labels = torch.as_tensor(labels).to(torch.int32).reshape(-1)
        label_lens_c = torch.as_tensor(label_lens).to(torch.int32).reshape(-1).cpu()
        act_lens_c = torch.as_tensor(act_lens).to(torch.int32).reshape(-1).cpu()
        B = acts.shape[1]
        if label_lens_c.numel() != B or act_lens_c.numel() != B:
            raise RuntimeError("CTCLoss: act_lens / label_lens must have one entry per batch element")
        if int(label_lens_c.sum()) != labels.numel():
            raise RuntimeError("CTCLoss: sum(label_lens) != len(labels)")
        if int(act_lens_c.max()) > acts.shape[0]:
            raise RuntimeError("CTCLoss: act_lens exceeds the time dimension")
        offsets = torch.zeros(B, dtype=torch.int32)
        if B > 1:
            offsets[1:] = torch.cumsum(label_lens_c, 0)[:-1].to(torch.int32)
        max_l = int(label_lens_c.max()) if B > 0 else 0
        if labels.numel() == 0:
            labels = torch.zeros(1, dtype=torch.int32)
        # one host->device copy for the four small integer arrays (they were four ~5 us copies on the critical path)
        nl = labels.numel()
        packed = torch.cat([labels.cpu(), offsets, label_lens_c, act_lens_c]).pin_memory().to(dev, non_blocking=True)
        return ops.CtcFn.apply(acts, packed[:nl], packed[nl:nl + B], packed[nl + B:nl + 2 * B], packed[nl + 2 * B:nl + 3 * B], max_l)
