#!/usr/bin/env python
"""bench.py — training line-images/sec of the CnnOcrModel hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: started plainly (no WORLD_SIZE in the environment) this script spawns its own N rank processes, one per GPU, and
relays rank 0's JSON line; started under `python -m torch.distributed.run --nproc-per-node N ...` it is one of the ranks.
The parent of a self-launch never touches the GPU (no exec of a GPU-initialised process, children are fresh interpreters).

Workload (modelled on the reference's src/speed_test.py): per GPU a batch of 32
synthetic 1x30x600 grey lines x~U[0,1), 20 labels/line, English alphabet (V=96), 3x BiLSTM-512, lstm_input_dim 128,
dropout 0.5, fp32.  One step = train() of src/train_cnn_lstm.py:131-150: forward + CTC + backward + (RCCL all-reduce of
the flat gradient) + clamp(+-5) + Adam, returning the loss as a Python float.  `value` is measured with the image batch
resident in HBM; the same K steps fed from a pinned host batch (H2D inside the step) are reported as `h2d_inclusive`.
Prints ONE JSON line on rank 0."""
import argparse
import json
import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # before the first HIP call: see vistaocr_amd/__init__.py
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak (= vector peak)
HBM_PEAK_GBS = 8000.0

HP = dict(num_in_channels=1, input_line_height=30, rds_line_height=30, lstm_input_dim=128, num_lstm_layers=3,
          num_lstm_hidden_units=512, p_lstm_dropout=0.5)
B, HIMG, WIMG, LABELS = 32, 30, 600, 20
CPU_BASELINE_THREADS = 32     # oneDNN/ATen stop scaling (and collapse on the per-time-step LSTM ops) far below 256 threads
CPU_TIMED_STEPS = 2
PROFILE_STEPS = 5             # un-timed pass that records HIP events around every entry point (breakdown only)


def make_batch(rank, vocab):
    import torch
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.rand(B, 1, HIMG, WIMG, generator=g)
    widths = torch.full((B,), WIMG, dtype=torch.int32)
    tgt = torch.randint(1, vocab, (B * LABELS,), generator=g).to(torch.int32)
    tl = torch.full((B,), LABELS, dtype=torch.int32)
    return x, tgt, widths, tl


def parity_samples():
    import torch
    g = torch.Generator().manual_seed(777)
    return torch.rand(B, 64, 2, generator=g), torch.rand(B, 128, 2, generator=g)


def conv_flops(args):
    n, cin, h, w, cout = args[4:9]
    return 2.0 * n * h * w * cin * cout * 9


# ------------------------------------------------------------------------------------------------ CPU leg (child process)
def cpu_baseline_worker(parity_file):
    """The oracle (CPU restatement of the same step on PyTorch-CPU).  (1) parity: ONE forward of the same batch with the
    GPU model's own initial weights and the same pool samples (dropout off on both sides) -> loss / greedy labels vs the
    HIP path's; (2) baseline: CPU_TIMED_STEPS timed train steps (fwd+CTC+bwd+clamp+Adam) of the same workload."""
    import torch
    import vistaocr_amd as va
    from oracle import vista_oracle as vo
    from vistaocr_amd.textutils import compute_cer_wer
    al = va.english_alphabet()
    vocab = len(al)
    n_cores = os.cpu_count() or 1
    n_thr = min(n_cores, CPU_BASELINE_THREADS)
    torch.set_num_threads(n_thr)
    x, tgt, widths, tl = make_batch(0, vocab)
    out = {}
    hp = dict(HP)
    if parity_file and os.path.exists(parity_file):
        blob = torch.load(parity_file, map_location="cpu", weights_only=True)
        hp = dict(HP, num_lstm_hidden_units=int(blob["hidden"]))
        sd = {}
        for k, v in blob["state"].items():
            if k.endswith("num_batches_tracked"):
                continue
            t = v.clone().float()
            if not (k.endswith("running_mean") or k.endswith("running_var")):
                t.requires_grad_(True)
            sd[k] = t
        u = parity_samples()
        t0 = time.time()
        with torch.no_grad():
            lo, ln = vo.forward(sd, hp, x, widths.tolist(), u, training=True, lstm_training=False)
            loss_o = float(vo.ctc_criterion(lo, tgt, ln, tl))
        strs_o, labels_o = vo.greedy_decode(lo, ln, al.idx_to_char, uxxxx=True)
        top2 = torch.sort(lo, dim=2, descending=True)[0]
        margin = float((top2[:, :, 0] - top2[:, :, 1]).min())
        loss_h = float(blob["loss"])
        labels_h = [[int(v) for v in row] for row in blob["labels"]]
        strs_h = list(blob["strings"])
        cer = 0.0
        for hyp, ref in zip(strs_h, strs_o):
            c, _ = compute_cer_wer(hyp, ref) if ref.strip() != "" else ((0.0 if hyp.strip() == "" else 1.0), 0.0)
            cer += c / len(strs_o)
        out["parity"] = dict(loss_rel_err=abs(loss_h - loss_o) / abs(loss_o), label_mismatches=sum(int(a != b) for a, b in zip(labels_h, labels_o)),
                             lines=len(labels_o), labels_emitted=sum(len(l) for l in labels_o), cer=cer, hip_loss=loss_h, oracle_loss=loss_o,
                             lens_equal=bool(ln.tolist() == list(blob["lens"])), min_top2_margin=margin,
                             oracle_forward_s=round(time.time() - t0, 1),
                             what="one forward of the bench batch, the HIP model's initial weights copied to the CPU oracle, same "
                                  "pool samples, dropout off on both sides")
        state = sd
    else:
        state = vo.init_uniform_state(hp, vocab, seed=0)
    opt = torch.optim.Adam([p for _, p in vo.trainable(state)], lr=1e-3)
    u = (torch.rand(B, 64, 2), torch.rand(B, 128, 2))
    # tiny warm-up (thread pools, oneDNN primitive caches) on 2 short lines, not timed
    vo.train_step(state, hp, opt, x[:2, :, :, :120].contiguous(), [120, 120], tgt[:8], torch.tensor([4, 4], dtype=torch.int32),
                  (u[0][:2], u[1][:2]))
    times = []
    for _ in range(CPU_TIMED_STEPS):
        t0 = time.time()
        vo.train_step(state, hp, opt, x, widths.tolist(), tgt, tl, u)
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    out["cpu_baseline"] = dict(value=round(B / dt, 3), unit="line-images/sec", cores=n_thr, host_cores=n_cores, kind="port",
                               sample="%d timed train steps (fwd+CTC+bwd+clamp+Adam) of the same batch-32 30x600 workload on the "
                                      "oracle (PyTorch-CPU restatement), %s s, torch.set_num_threads(%d) of os.cpu_count()=%d"
                                      % (len(times), "/".join("%.1f" % t for t in times), n_thr, n_cores))
    return out


def cpu_leg(parity_file, limit_s=240):
    """Run the CPU leg in a child process with a hard time limit so a slow host can never stall the bench."""
    n_thr = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", parity_file or ""], capture_output=True,
                           text=True, timeout=limit_s, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return dict(cpu_baseline=dict(value=None, unit="line-images/sec", cores=0, kind="port", sample="worker failed: " + r.stderr[-300:]))
    except subprocess.TimeoutExpired:
        return dict(cpu_baseline=dict(value=round(B / (limit_s / (CPU_TIMED_STEPS + 1.0)), 3), unit="line-images/sec", cores=n_thr,
                                      host_cores=os.cpu_count(), kind="port",
                                      sample="upper bound: parity forward + %d steps did not finish within %d s" % (CPU_TIMED_STEPS, limit_s)))


# ------------------------------------------------------------------------------------------------ self-launch of N ranks
def launch_ranks(args, argv):
    import socket
    import torch
    n = args.gpus
    have = torch.cuda.device_count()          # does not initialise the GPU in this process
    if have < n and not args.share_gpu:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (n, have))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        if args.share_gpu:      # two persistent LSTM sweeps from two processes must not compete for one GPU's CUs
            env.setdefault("VOCR_LSTM_PERSISTENT", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for cand in reversed((out0 or "").strip().splitlines()):
        if cand.startswith("{"):
            line = cand
            break
    if any(rcs) or line is None:
        sys.stderr.write((out0 or "")[-2000:])
        raise SystemExit("bench.py: rank exit codes %s, no result line" % rcs)
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    device_index = 0 if args.share_gpu else local_rank
    torch.cuda.set_device(device_index)
    use_dist = world > 1 or os.environ.get("VOCR_FORCE_DIST") == "1"      # FORCE: exercise RCCL init/all-reduce on one GPU
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=args.backend)

    import vistaocr_amd as va
    from vistaocr_amd import _lib
    hp = dict(HP, num_lstm_hidden_units=args.hidden)
    if args.conv_dtype != "fp32":
        hp["conv_dtype"] = args.conv_dtype
    al = va.english_alphabet()
    torch.manual_seed(0)                                  # same init on every rank (replicas)
    model = va.CnnOcrModel(alphabet=al, verbose=False, **hp)
    crit = va.CTCLoss()
    x_host, tgt, widths, tl = make_batch(rank, len(al))
    x_host = x_host.pin_memory()

    # ---- parity leg, GPU side (rank 0, N=1): one forward with fixed pool samples and dropout off; the CPU oracle repeats it
    # with these very weights (the child process reads them from a temporary file)
    parity_file = None
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if want_cpu and args.conv_dtype == "fp32":
        init_state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        model.train()
        model.lstm.eval()
        model.pool_samples = list(parity_samples())
        with torch.no_grad():
            lg, lens = model(x_host, widths)
            ploss = float(crit(lg, tgt, lens, tl))
            pstr, plabels = va.decoder.greedy_label_sequences(lg, lens, al)
        model.pool_samples = None
        model.load_state_dict(init_state)               # undo the BatchNorm running-stat update of that forward
        fd, parity_file = tempfile.mkstemp(suffix=".pt", prefix="vocr_parity_")
        os.close(fd)
        torch.save(dict(state=init_state, hidden=args.hidden, loss=ploss, labels=plabels, strings=pstr, lens=lens.tolist()), parity_file)
        del init_state, lg
    model.train()
    opt = va.make_optimizer(model, lr=1e-3)          # flat Adam; all-reduce in two buckets, the big one under the CNN backward
    x_dev = x_host.cuda()
    batch_dev = (x_dev, tgt, widths, tl, {})
    batch_host = (x_host, tgt, widths, tl, {})

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(batch, steps, event_every=None):
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            if event_every is not None:
                _lib.enable_timing(CONV_FWD_ONLY if (event_every and i % event_every == 0) else None, keep=True)
            loss = va.train(batch, model, crit, opt)       # the function the reference calls; returns the loss float
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, loss

    for _ in range(args.warmup):
        va.train(batch_dev, model, crit, opt)
    # timed region: HIP events only around the forward-pass launches of the dominant kernel (conv3x3 with a bias: 7 per step),
    # on the stream they are launched on, in every 10th timed step (an event pair costs the queue a few microseconds of
    # overlap; 26 pairs per step slowed the step by 3-5 %); the full per-entry-point breakdown comes from a separate
    # un-timed pass below
    CONV_FWD_ONLY = {"vocr_conv3x3_fwd": lambda a: a[2] is not None, "vocr_conv3x3_wino_fwd": lambda a: a[2] is not None}
    _lib.enable_timing(CONV_FWD_ONLY)
    dt, final_loss = timed(batch_dev, args.steps, event_every=args.event_every)
    conv_recs = _lib.timing_records().get("vocr_conv3x3_fwd", []) + _lib.timing_records().get("vocr_conv3x3_wino_fwd", [])
    _lib.enable_timing(None)
    dt_h2d, _ = timed(batch_host, args.steps)
    names = ["vocr_conv3x3_fwd", "vocr_conv3x3_wino_fwd", "vocr_conv3x3_wgrad", "vocr_conv3x3_wgrad_wino", "vocr_lstm_fwd", "vocr_lstm_fwd_range", "vocr_lstm_bwd_bias", "vocr_gemm",
             "vocr_bn_train_stats", "vocr_bn_relu_apply", "vocr_bn_relu_fracpool2x2_fwd", "vocr_bn_relu_bwd", "vocr_fracpool2x2_bwd",
             "vocr_ctc_loss_grad", "vocr_clamp_adam"]
    _lib.enable_timing(names)
    for _ in range(PROFILE_STEPS):
        va.train(batch_dev, model, crit, opt)
    torch.cuda.synchronize()
    prof = _lib.timing_records()
    _lib.enable_timing(None)
    opt.check_health()
    ranks_seen = dist.get_world_size() if use_dist else 1

    out = None
    if rank == 0:
        ms = 1000.0 * dt / args.steps
        value = B * world * args.steps / dt
        # roofline of the dominant kernel: conv3x3 implicit-GEMM, f32 MFMA-bound.  Measured on the launches of the
        # forward pass (the first n_conv of every step's launches of this kernel): the data-gradient launches of the same
        # kernel run beside the weight-gradient kernel on the side stream, so their wall durations measure the pair.
        n_conv = sum(1 for k, v in model.state_dict().items() if k.endswith(".weight") and v.dim() == 4)
        fwd_recs = conv_recs
        cf_flops = sum(conv_flops(a) for a, _, _ in fwd_recs)
        cf_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in fwd_recs)
        n_launch = max(1, len(fwd_recs))
        achieved = cf_flops / (cf_ms * 1e-3) / 1e12 if cf_ms > 0 else 0.0
        prof_conv = prof.get("vocr_conv3x3_fwd", []) + prof.get("vocr_conv3x3_wino_fwd", [])
        all_ms = sum(e0.elapsed_time(e1) for _, e0, e1 in prof_conv)
        all_tf = sum(conv_flops(a) for a, _, _ in prof_conv) / (all_ms * 1e-3) / 1e12 if all_ms > 0 else 0.0
        per_step = len(prof_conv) // PROFILE_STEPS
        traffic = None
        try:        # HBM-side bytes per launch of the same kernel from the committed PMC passes (cannot be collected live)
            tj = json.load(open(os.path.join(ROOT, "profiles", "conv_traffic.json")))
            traffic = int((tj["fetch_KB_per_launch"] + tj["write_KB_per_launch"]) * 1024)
        except Exception:
            pass
        breakdown = {}
        for name, lst in prof.items():
            if lst:
                breakdown[name] = round(sum(e0.elapsed_time(e1) for _, e0, e1 in lst) / PROFILE_STEPS, 3)
        out = {
            "metric": "line-images/sec (train, batch 32, 30x600 grey)", "value": round(value, 2), "unit": "line-images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.conv_dtype == "fp32" else "f32 with fp16 conv operands (fp32 accumulate)", "data": "synthetic",
            "config": {"workload": "configs[1]: 32 synthetic 1x30x600 grey lines per GPU, 20 labels/line, V=96, "
                                   "3xBiLSTM-%d, train() = fwd+CTC+bwd+allreduce+clamp+Adam, loss returned as a float" % args.hidden,
                       "global_batch": B * world, "parallelism": "dp%d" % world, "ranks_seen": ranks_seen,
                       "backend": (args.backend + ("/RCCL" if args.backend == "nccl" else "")) if use_dist else "none",
                       "final_loss": round(float(final_loss), 3)},
            "h2d_inclusive": {"value": round(B * world * args.steps / dt_h2d, 2), "ms_per_step": round(1000.0 * dt_h2d / args.steps, 3),
                              "what": "same K steps with the image batch in pinned host memory (H2D inside train())"},
            "roofline": {"bound": "mfma", "kernel": "conv3x3_wino_kernel (F(2,3) along the row: 2/3 of the multiplications) behind vocr_conv3x3_wino_fwd (vocr_conv3x3_fwd with VOCR_CONV_WINO=0), implicit GEMM on f32 MFMA 32x32x2, forward-pass launches; achieved = ALGORITHMIC direct-convolution FLOPs / time",
                         "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                         "avg_launch_ms": round(cf_ms / n_launch, 4), "launches_per_step": n_conv, "launches_timed": n_launch,
                         "all_launches_incl_dgrad_beside_wgrad": {"achieved": round(all_tf, 2), "launches_per_step": per_step},
                         "whole_step": {"flop": 1.856e12 if args.hidden == 512 else None,
                                        "achieved": round(1.856e12 / (ms * 1e-3) / 1e12, 2) if args.hidden == 512 else None,
                                        "frac": round(1.856e12 / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4) if args.hidden == 512 else None}},
            "ms_per_step_by_entry_point": breakdown,
        }
        if want_cpu:
            out.update(cpu_leg(parity_file))
    if parity_file and os.path.exists(parity_file):
        os.unlink(parity_file)
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


def main():
    if "--cpu-baseline-worker" in sys.argv:
        i = sys.argv.index("--cpu-baseline-worker")
        pf = sys.argv[i + 1] if i + 1 < len(sys.argv) else ""
        print(json.dumps(cpu_baseline_worker(pf)))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-every", type=int, default=10, help="HIP events around the dominant kernel in every n-th timed step (0: none)")
    ap.add_argument("--hidden", type=int, default=512)
    ap.add_argument("--conv-dtype", default="fp32", choices=["fp32", "fp16"],
                    help="fp16 = BASELINE config 5's fp16-operand conv MFMA (fp32 accumulate); the headline metric is fp32")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: plumbing tests on boxes without N GPUs")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks on device 0 (plumbing test of the launcher; needs --backend gloo)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, sys.argv[1:])
        return
    run_rank(args)


if __name__ == "__main__":
    main()
