#!/usr/bin/env python
"""bench.py — training line-images/sec of the CnnOcrModel hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (modelled on the reference's src/speed_test.py): per GPU a batch of 32
synthetic 1x30x600 grey lines x~U[0,1), 20 labels/line, English alphabet (V=96), 3x BiLSTM-512, lstm_input_dim 128,
dropout 0.5, fp32.  One step = forward + CTC + backward + (RCCL all-reduce of the flat gradient) + clamp(+-5) +
Adam, exactly src/train_cnn_lstm.py:131-150.  The image batch is resident in HBM when the timed region starts
(targets/lengths stay on the host, as in the reference's contract).  Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

F32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak (= vector peak)
HBM_PEAK_GBS = 8000.0

HP = dict(num_in_channels=1, input_line_height=30, rds_line_height=30, lstm_input_dim=128, num_lstm_layers=3,
          num_lstm_hidden_units=512, p_lstm_dropout=0.5)
B, HIMG, WIMG, LABELS = 32, 30, 600, 20


def make_batch(rank, vocab):
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(B, 1, HIMG, WIMG, generator=g)
    widths = torch.full((B,), WIMG, dtype=torch.int32)
    tgt = torch.randint(1, vocab, (B * LABELS,), generator=g).to(torch.int32)
    tl = torch.full((B,), LABELS, dtype=torch.int32)
    return x, tgt, widths, tl


def conv_flops(args):
    n, cin, h, w, cout = args[4:9]
    return 2.0 * n * h * w * cin * cout * 9


CPU_BASELINE_THREADS = 32     # oneDNN/ATen stop scaling (and collapse on the per-time-step LSTM ops) far below 256 threads


def cpu_baseline_worker(vocab):
    """The oracle (CPU restatement of the same step on PyTorch-CPU) on ONE batch-32 step of the same workload."""
    from oracle import vista_oracle as vo
    torch.manual_seed(0)
    n_thr = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(n_thr)
    sd = vo.init_uniform_state(HP, vocab, seed=0)
    opt = torch.optim.Adam([p for _, p in vo.trainable(sd)], lr=1e-3)
    x, tgt, widths, tl = make_batch(0, vocab)
    u = (torch.rand(B, 64, 2), torch.rand(B, 128, 2))
    # tiny warm-up (thread pools, oneDNN primitive caches) on 2 short lines, not timed
    vo.train_step(sd, HP, opt, x[:2, :, :, :120].contiguous(), [120, 120], tgt[:2 * LABELS][:8], torch.tensor([4, 4], dtype=torch.int32),
                  (u[0][:2], u[1][:2]))
    t0 = time.time()
    vo.train_step(sd, HP, opt, x, widths.tolist(), tgt, tl, u)
    dt = time.time() - t0
    return dict(value=round(B / dt, 3), unit="line-images/sec", cores=n_thr, kind="port",
                sample="1 timed train step (fwd+CTC+bwd+clamp+Adam) of the same batch-32 30x600 workload on the oracle "
                       "(PyTorch-CPU restatement), %.1f s" % dt)


def cpu_baseline(vocab, limit_s=240):
    """Run the CPU leg in a child process with a hard time limit so a slow host can never stall the bench."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker"], capture_output=True, text=True,
                           timeout=limit_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return dict(value=None, unit="line-images/sec", cores=0, kind="port", sample="worker failed: " + r.stderr[-200:])
    except subprocess.TimeoutExpired:
        return dict(value=round(B / limit_s, 3), unit="line-images/sec", cores=min(os.cpu_count() or 1, CPU_BASELINE_THREADS), kind="port",
                    sample="upper bound: one batch-32 step did not finish within %d s" % limit_s)


def main():
    if "--cpu-baseline-worker" in sys.argv:
        import vistaocr_amd as va
        print(json.dumps(cpu_baseline_worker(len(va.english_alphabet()))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--conv-dtype", default="fp32", choices=["fp32", "fp16"],
                    help="fp16 = BASELINE config 5's fp16-operand conv MFMA (fp32 accumulate); the headline metric is fp32")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("VOCR_FORCE_DIST") == "1"      # FORCE: exercise RCCL init/all-reduce on one GPU
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import vistaocr_amd as va
    from vistaocr_amd import _lib
    hp = dict(HP, num_lstm_hidden_units=args.hidden)
    if args.conv_dtype != "fp32":
        hp["conv_dtype"] = args.conv_dtype
    al = va.english_alphabet()
    torch.manual_seed(0)                                  # same init on every rank (replicas)
    model = va.CnnOcrModel(alphabet=al, verbose=False, **hp)
    model.train()
    opt = va.make_optimizer(model, lr=1e-3)          # flat Adam; all-reduce in two buckets, the big one under the CNN backward
    crit = va.CTCLoss()
    x, tgt, widths, tl = make_batch(rank, len(al))
    x = x.cuda()
    batch = (x, tgt, widths, tl, {})

    def step():
        return va.train_async(batch, model, crit, opt)

    for _ in range(args.warmup):
        step()
    timed = ["vocr_conv3x3_fwd", "vocr_conv3x3_wgrad", "vocr_lstm_fwd", "vocr_lstm_fwd_range", "vocr_lstm_bwd_bias", "vocr_gemm"]
    _lib.enable_timing(timed)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    recs = _lib.timing_records()
    _lib.enable_timing(None)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    final_loss = float(loss)

    if rank == 0:
        ms = 1000.0 * dt / args.steps
        value = B * world * args.steps / dt
        # roofline of the dominant kernel: conv3x3 implicit-GEMM, f32 MFMA-bound.  Measured on the launches of the
        # forward pass (the first n_conv of every step's launches of this kernel): the data-gradient launches of the same
        # kernel run beside the weight-gradient kernel on the side stream, so their wall durations measure the pair.
        n_conv = sum(1 for k, v in model.state_dict().items() if k.endswith(".weight") and v.dim() == 4)
        conv_recs = recs.get("vocr_conv3x3_fwd", [])
        per_step = max(1, len(conv_recs) // max(1, args.steps))
        fwd_recs = [r for i, r in enumerate(conv_recs) if i % per_step < n_conv]
        cf_flops = sum(conv_flops(a) for a, _, _ in fwd_recs)
        cf_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in fwd_recs)
        n_launch = max(1, len(fwd_recs))
        achieved = cf_flops / (cf_ms * 1e-3) / 1e12 if cf_ms > 0 else 0.0
        all_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in conv_recs)
        all_tf = sum(conv_flops(a) for a, _, _ in conv_recs) / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0
        traffic = None
        try:        # HBM-side bytes per launch of the same kernel from the committed PMC passes (cannot be collected live)
            tj = json.load(open(os.path.join(ROOT, "profiles", "conv_traffic.json")))
            traffic = int((tj["fetch_KB_per_launch"] + tj["write_KB_per_launch"]) * 1024)
        except Exception:
            pass
        breakdown = {}
        for name, lst in recs.items():
            breakdown[name] = round(sum(e0.elapsed_time(e1) for _, e0, e1 in lst) / args.steps, 3)
        out = {
            "metric": "line-images/sec (train, batch 32, 30x600 grey)", "value": round(value, 2), "unit": "line-images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.conv_dtype == "fp32" else "f32 with fp16 conv operands (fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "configs[1]: 32 synthetic 1x30x600 grey lines per GPU, 20 labels/line, V=96, "
                                   "3xBiLSTM-%d, fwd+CTC+bwd+allreduce+clamp+Adam" % args.hidden,
                       "global_batch": B * world, "parallelism": "dp%d" % world, "final_loss": round(final_loss, 3)},
            "roofline": {"bound": "mfma", "kernel": "conv3x3_kernel (implicit GEMM, f32 MFMA 32x32x2), forward-pass launches",
                         "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "avg_launch_ms": round(cf_ms / n_launch, 4), "launches_per_step": n_launch // max(1, args.steps),
                         "all_launches_incl_dgrad_beside_wgrad": {"achieved": round(all_tf, 2), "launches_per_step": per_step}},
            "ms_per_step_by_entry_point": breakdown,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(len(al))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the JSON line is the LAST thing on stdout: flush whatever native libraries (RCCL's version banner) still
        # hold in C stdio buffers first
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
