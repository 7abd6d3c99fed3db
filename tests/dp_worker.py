"""One data-parallel rank of the GPU tests (started as a fresh process per rank by tests/test_round3_gpu.py; never imported by the
product).  Builds the replica with the identical init, gives the rank its own RNG (va.seed_rank), checks that the same input batch
gives DIFFERENT logits on different ranks (their FractionalMaxPool samples differ) and that after two real train steps on per-rank
batches (gradients summed over the ranks, two buckets) every rank holds bit-identical weights.  Rank 0 writes a JSON report."""
import argparse
import hashlib
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--share-gpu", action="store_true")
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev_index = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(dev_index)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group(args.backend)
    import vistaocr_amd as va
    al = va.english_alphabet()
    torch.manual_seed(0)
    model = va.CnnOcrModel(alphabet=al, verbose=False, input_line_height=30, rds_line_height=30, lstm_input_dim=32, num_lstm_layers=2,
                           num_lstm_hidden_units=64, p_lstm_dropout=0.5, num_in_channels=1)
    init_hash = hashlib.sha256(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy().tobytes()).hexdigest()
    va.seed_rank(model, rank, base=77)
    opt = va.make_optimizer(model, lr=1e-3)
    crit = va.CTCLoss()
    model.train()
    B, W, L = 4, 200, 5
    g = torch.Generator().manual_seed(5)                      # the SAME batch on every rank ...
    x_same = torch.rand(B, 1, 30, W, generator=g)
    widths = torch.full((B,), W, dtype=torch.int32)
    with torch.no_grad():
        lg, _ = model(x_same, widths)                         # ... gives different logits: the ranks' pool samples differ
    logit_sum = float(lg.double().sum())
    losses = []
    for step in range(2):
        gb = torch.Generator().manual_seed(1000 * step + rank)    # per-rank data
        x = torch.rand(B, 1, 30, W, generator=gb)
        tgt = torch.randint(1, len(al), (B * L,), generator=gb).to(torch.int32)
        tl = torch.full((B,), L, dtype=torch.int32)
        losses.append(va.train((x, tgt, widths, tl, {}), model, crit, opt))
    torch.cuda.synchronize()
    w_hash = hashlib.sha256(opt.flat_p.detach().cpu().numpy().tobytes()).hexdigest()
    rec = dict(rank=rank, init_hash=init_hash, weights_hash=w_hash, logit_sum=logit_sum, losses=losses, dropout_seed=int(model.dropout_seed),
               device=dev_index, backend=args.backend, world=dist.get_world_size())
    allrec = [None] * world
    dist.all_gather_object(allrec, rec)
    if rank == 0:
        with open(args.out, "w") as fh:
            json.dump(allrec, fh)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
