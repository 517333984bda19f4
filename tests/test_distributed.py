"""N>1 path on CPU: world_size-2 gloo processes shard the env range and
all-gather the trajectory slab (the only collective on the path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from robopianist_amd import distributed as rpd


def test_shard_envs_partition():
    for total in (0, 1, 7, 4096, 8191):
        for world in (1, 2, 3, 8):
            blocks = [rpd.shard_envs(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for a, b in zip(blocks, blocks[1:]):
                assert a[1] == b[0]
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        rpd.shard_envs(8, 2, 2)
    assert rpd.rank_seed(12345, 3) == 15345


def test_trajectory_record_round_trip():
    torch.manual_seed(0)
    E, nv = 5, 140
    qpos = torch.randn(E, nv)
    act = torch.rand(E, 88) > 0.7
    rec = rpd.pack_trajectory_record(qpos, torch.arange(E), torch.ones(E), torch.full((E,), 2), act)
    assert rec.shape == (E, nv + 6) and rec.dtype == torch.float32
    assert torch.equal(rec[:, :nv], qpos)
    assert torch.equal(rpd.unpack_key_activation(rec, nv), act)
    # fp64 state keeps its precision through the record (the default follows qpos)
    q64 = torch.randn(E, nv, dtype=torch.float64)
    rec64 = rpd.pack_trajectory_record(q64, torch.arange(E), torch.ones(E), torch.full((E,), 2), act)
    assert rec64.shape == (E, nv + 5) and rec64.dtype == torch.float64
    assert torch.equal(rec64[:, :nv], q64) and torch.equal(rec64[:, nv + 2], torch.full((E,), 2.0, dtype=torch.float64))
    assert torch.equal(rpd.unpack_key_activation(rec64, nv), act)
    rec32 = rpd.pack_trajectory_record(q64, torch.arange(E), torch.ones(E), torch.full((E,), 2), act, dtype=torch.float32)
    assert rec32.dtype == torch.float32 and torch.equal(rpd.unpack_key_activation(rec32, nv), act)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_envs, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = rpd.init_from_env(backend="gloo")
    start, stop = rpd.shard_envs(total_envs, r, w)
    # each rank "simulates" its own block: record = global env id in column 0
    g = torch.Generator().manual_seed(rpd.rank_seed(12345, r))
    local = torch.zeros((stop - start, 4))
    local[:, 0] = torch.arange(start, stop)
    local[:, 1] = torch.rand(stop - start, generator=g)
    full = rpd.gather_trajectories(local)
    # the enqueue-only form used by bench.py (overlaps the next step)
    full2, work = rpd.gather_trajectories(local, async_op=True)
    work.wait()
    assert torch.equal(full, full2)
    dist.barrier()
    out_q.put((r, full[:, 0].tolist(), float(full[:, 1].sum())))
    dist.destroy_process_group()


def test_two_rank_gloo_gather():
    world, total = 2, 16
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    # every rank sees every env exactly once, in global order, and identical payloads
    for _, ids, _ in res:
        assert ids == list(map(float, range(total)))
    assert res[0][2] == pytest.approx(res[1][2])


def _nccl_worker(rank, world, port, total_envs, out_q):
    """One rank per GPU over RCCL (backend "nccl" IS RCCL on ROCm): the same gather as the gloo test, device tensors."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r, w, local_rank = rpd.init_from_env(backend="nccl")
    dev = torch.device("cuda", local_rank)
    start, stop = rpd.shard_envs(total_envs, r, w)
    local = torch.zeros((stop - start, 145), dtype=torch.float64, device=dev)   # (the fp64 record's width: 140 + 5)
    local[:, 0] = torch.arange(start, stop, device=dev)
    local[:, 1] = float(rpd.rank_seed(12345, r))
    out = torch.empty((total_envs, 145), dtype=torch.float64, device=dev)
    full, work = rpd.gather_trajectories(local, async_op=True, out=out)   # (the enqueue-only form bench.py uses)
    work.wait()
    torch.cuda.synchronize(dev)
    full2 = rpd.gather_trajectories(local)
    assert torch.equal(full, full2) and full.data_ptr() == out.data_ptr()
    dist.barrier()
    out_q.put((r, full[:, 0].cpu().tolist(), full[:, 1].cpu().tolist()))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available() or torch.cuda.device_count() < 2,
                    reason="the RCCL gather needs two GPUs (one rank per device); the gloo twin above runs everywhere")
def test_two_rank_rccl_gather():
    """Round 5: the first multi-GPU driver run must not also be RCCL's first run -- wherever two devices are
    visible, two ranks all-gather the trajectory record over the nccl (= RCCL) backend, blocks in global env order."""
    world, total = 2, 4096
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_nccl_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for _, ids, seeds in res:
        assert ids == list(map(float, range(total)))
        assert seeds[:total // 2] == [12345.0] * (total // 2) and seeds[total // 2:] == [13345.0] * (total // 2)


def _nccl_single_rank_worker(port, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
    local = torch.arange(4096 * 145, dtype=torch.float64, device=dev).reshape(4096, 145)
    out = torch.empty_like(local)
    full, work = rpd.gather_trajectories(local, async_op=True, out=out, force_collective=True)
    work.wait()
    torch.cuda.synchronize(dev)
    ok = bool(torch.equal(full, local)) and full.data_ptr() == out.data_ptr()
    dist.barrier()
    out_q.put(ok)
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_single_rank_rccl_all_gather_executes_on_the_one_gpu_there_is():
    """Round 6: RCCL refuses several ranks per device, so on a one-GPU box the two-rank test
    above skips -- but a ONE-rank nccl (= RCCL) group does initialise the library, build a communicator and execute
    all_gather_into_tensor on the device, through the very call bench.py's gather makes.  The first multi-GPU driver run
    is then at least not RCCL's first run in this image."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_single_rank_worker, args=(_free_port(), q))
    p.start()
    assert q.get(timeout=300) is True
    p.join(timeout=120)
    assert p.exitcode == 0
