"""World-size-2 gloo tests of the multi-GPU plumbing (runs on CPU): arena broadcast + prompt sharding."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kandinsky2_amd.parallel import broadcast_arena, gather_outputs, launch_ranks, shard_range


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
    try:
        n = 3 * 1024 * 1024 + 17
        g = torch.Generator().manual_seed(7)
        full = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g)
        arena = broadcast_arena(full.clone() if rank == 0 else None, n, "cpu", src=0, chunk_bytes=1 << 20)
        ok = torch.equal(arena, full)
        lo, hi = shard_range(7, rank, world)
        local = torch.arange(lo, hi, dtype=torch.float32)[:, None].repeat(1, 3)
        got = gather_outputs(local, dst=0)
        if rank == 0:
            cat = torch.cat(got, 0)
            ok = ok and torch.equal(cat[:, 0], torch.arange(7, dtype=torch.float32))
        q.put((rank, ok, (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (1, 7, 8, 32, 33):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_arena_broadcast_and_gather_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert sorted(span for _, _, span in res) == [(0, 4), (4, 7)]


def _launched_rank(rank, world, outdir):
    """what a bench rank does with the environment the self-launcher exports: init from env, one collective, observed world size"""
    assert int(os.environ["RANK"]) == rank and int(os.environ["WORLD_SIZE"]) == world and os.environ["MASTER_ADDR"] == "127.0.0.1"
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    try:
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        with open(os.path.join(outdir, f"rank{rank}.txt"), "w") as f:
            f.write(f"{dist.get_world_size()} {t.item()}")
    finally:
        dist.destroy_process_group()


def _failing_rank(rank, world):
    if rank == 1:
        raise SystemExit(3)


def test_self_launcher_exports_the_torchrun_environment_gloo_world2(tmp_path):
    """`bench.py --gpus N` without torch.distributed.run spawns its ranks itself (kandinsky2_amd.parallel.launch_ranks)."""
    launch_ranks(_launched_rank, 2, args=(str(tmp_path),))
    got = sorted(open(tmp_path / f"rank{r}.txt").read() for r in range(2))
    assert got == ["2 2.0", "2 2.0"]            # both ranks saw world size 2 and the max over ranks


def test_self_launcher_fails_loudly():
    import pytest
    with pytest.raises(RuntimeError, match="only 1 device"):
        launch_ranks(_launched_rank, 4, args=("/tmp",), devices_visible=1)       # never a quiet single-rank run
    with pytest.raises(RuntimeError, match="ranks failed"):
        launch_ranks(_failing_rank, 2)


def _plan_worker(rank, world, port, q):
    """a rank that RECEIVES the arena builds the same engine as rank 0 - same bytes, same plan, same tile configuration
    (= same fp32 summation orders = same bits) - because the configuration is a function of the problem shapes and the shipped tile table
    only.  k22_unet_create / k22_unet_plan / k22_unet_tuning_report are host code: they run here without a GPU."""
    import ctypes as C
    import kandinsky2_amd as k22
    from kandinsky2_amd import _lib
    from kandinsky2_amd.pack import pack_arena
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        arch = k22.make_arch(k22.tiny_model_config())
        m = k22.Text2ImUNetHIP(arch, backend_dtype=torch.bfloat16, meta_params=rank != 0)
        table = m.arena_table()
        mine = None
        if rank == 0:
            mine, t0 = pack_arena(arch, k22.init_unet_state_dict(arch, seed=0), torch.bfloat16, "cpu")
            assert list(t0.items()) == list(table.items())          # the layout derived from shapes only is the layout rank 0 packed
        arena = broadcast_arena(mine, m.arena_bytes() if rank else mine.numel(), "cpu", src=0, chunk_bytes=1 << 22)
        L = _lib.lib()
        cfg = m._engine_config()
        arr = (_lib.K22Weight * len(table))()
        names = [n.encode() for n in table]
        for i, (name, (off, _n)) in enumerate(table.items()):
            arr[i].name = names[i]
            arr[i].ptr = arena.data_ptr() + off
        h = C.c_void_p()
        _lib.check(L.k22_unet_create(C.byref(cfg), arr, len(table), C.byref(h)))
        nbytes = C.c_size_t()
        _lib.check(L.k22_unet_plan(h, 2, 16, 16, C.byref(nbytes)))
        buf = C.create_string_buffer(1 << 16)
        _lib.check(L.k22_unet_tuning_report(h, buf, len(buf)))
        L.k22_unet_destroy(h)
        import hashlib
        digest = hashlib.sha256(arena.numpy().tobytes()).hexdigest()
        got = [None] * world
        dist.all_gather_object(got, (digest, nbytes.value, buf.value.decode()))
        q.put((rank, got))
    finally:
        dist.destroy_process_group()


def test_a_rank_that_receives_the_arena_plans_the_same_engine_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_plan_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    for _rank, got in res:
        assert got[0] == got[1], "arena bytes, workspace size and tile configurations must be equal on every rank"
        assert "taps" in got[0][2] and got[0][1] > 0
