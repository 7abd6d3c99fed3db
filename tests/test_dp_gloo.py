"""CPU, world_size 2 (gloo): the data-parallel exchange of the train step — one all-reduce (SUM) over the flat
gradient buffer that aliases every parameter's .grad — is correct by construction.  (On the GPUs the same call
runs over RCCL/xGMI; the fused clamp+Adam kernel itself is GPU-only and tested in test_ops_gpu.py.)"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vistaocr_amd as va
    torch.manual_seed(0)                     # replicas: identical init on every rank
    m = va.CnnOcrModel(alphabet=va.english_alphabet(), gpu=False, verbose=False, input_line_height=30, rds_line_height=30,
                       lstm_input_dim=16, num_lstm_layers=1, num_lstm_hidden_units=16, p_lstm_dropout=0.0)
    opt = va.FlatClampAdam(m.parameters(), lr=1e-3)
    n = opt.flat_g.numel()
    assert n == sum(p.numel() for p in m.parameters())
    # parameters and grads alias the flat buffers
    off = 0
    for p in m.parameters():
        assert p.data.data_ptr() == opt.flat_p.data_ptr() + 4 * off
        assert p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * off
        off += p.numel()
    opt.zero_grad()
    # autograd accumulates IN PLACE into the flat gradient
    loss = sum((p * (rank + 1)).sum() for p in m.parameters())
    loss.backward()
    assert torch.all(opt.flat_g == float(rank + 1))
    opt.all_reduce_grads()
    want = float(sum(r + 1 for r in range(world)))
    ok = bool(torch.all(opt.flat_g == want))
    # same replica weights everywhere
    w = opt.flat_p.clone()
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    ok = ok and bool(torch.equal(w, opt.flat_p))
    opt.zero_grad()
    ok = ok and float(opt.flat_g.abs().sum()) == 0.0
    q.put((rank, ok))
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _worker_buckets(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vistaocr_amd as va
    from vistaocr_amd import ops
    torch.manual_seed(0)
    m = va.CnnOcrModel(alphabet=va.english_alphabet(), gpu=False, verbose=False, input_line_height=30, rds_line_height=30,
                       lstm_input_dim=16, num_lstm_layers=1, num_lstm_hidden_units=16, p_lstm_dropout=0.0)
    opt = va.make_optimizer(m)
    n_cnn = sum(p.numel() for p in m.cnn.parameters())
    ok = opt._split == n_cnn and 0 < n_cnn < opt.flat_g.numel()
    opt.zero_grad()
    opt.flat_g[opt._split:] = float(rank + 1)          # sequence-side gradients are final ...
    for fn in m._vocr_hooks["sequence_grads_ready"]:
        fn()                                            # ... the backward fires the hook: tail all-reduce starts
    ok = ok and opt._tail_work is not None
    opt.flat_g[:opt._split] = float(10 * (rank + 1))   # CNN gradients arrive later
    opt.all_reduce_grads()
    want_tail = float(sum(r + 1 for r in range(world)))
    ok = ok and bool(torch.all(opt.flat_g[opt._split:] == want_tail)) and bool(torch.all(opt.flat_g[:opt._split] == 10 * want_tail))
    ok = ok and opt._tail_work is None
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_bucket_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_buckets, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def _worker_rng(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import vistaocr_amd as va
    torch.manual_seed(0)                     # identical init ...
    m = va.CnnOcrModel(alphabet=va.english_alphabet(), gpu=False, verbose=False, input_line_height=30, rds_line_height=30,
                       lstm_input_dim=16, num_lstm_layers=2, num_lstm_hidden_units=16, p_lstm_dropout=0.5)
    va.seed_rank(m, rank, base=1234)         # ... then per-rank randomness
    w = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    lo, hi = w.clone(), w.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same_weights = bool(torch.equal(lo, hi))
    u = torch.rand(4, 64, 2)                 # what CnnOcrModel.forward draws for the first FractionalMaxPool2d
    gathered = [torch.empty_like(u) for _ in range(world)]
    dist.all_gather(gathered, u)
    seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(seeds, torch.tensor([m.dropout_seed], dtype=torch.int64))
    differ = all(not torch.equal(gathered[i], gathered[j]) for i in range(world) for j in range(i))
    seeds_differ = len({int(s) for s in seeds}) == world
    # the exchange still gives every rank the same reduced gradient -> same weights after the (GPU-only) update
    opt = va.make_optimizer(m)
    opt.zero_grad()
    opt.flat_g.copy_(torch.rand(opt.flat_g.numel()))        # rank-specific gradients (the generators differ)
    opt.all_reduce_grads()
    g_lo, g_hi = opt.flat_g.clone(), opt.flat_g.clone()
    dist.all_reduce(g_lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(g_hi, op=dist.ReduceOp.MAX)
    q.put((rank, same_weights and differ and seeds_differ and bool(torch.equal(g_lo, g_hi))))
    dist.destroy_process_group()


def test_rank_offset_rng_world2():
    """Two ranks: identical weights after the identical init, different pool samples / dropout streams after seed_rank, identical
    reduced gradients after the exchange."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rng, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]
