"""N>1 host logic on CPU: world_size 2 over gloo (what the 8-GPU run does over RCCL)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from emavfi import dist as vdist


def test_shard_range_partitions_every_count():
    for n in (0, 1, 7, 8, 63, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [vdist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert vdist.shard_range(64, 3, 8) == (24, 32)  # BASELINE configs[3]: 64 pairs, 8 per GPU
    with pytest.raises(ValueError):
        vdist.shard_range(8, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w = vdist.init("gloo")
    assert (r, w) == (rank, world)
    # the path's one collective: rank 0's packed blob reaches every rank bit for bit
    g = torch.Generator().manual_seed(1234)
    blob = torch.randint(0, 256, (3262464,), dtype=torch.uint8, generator=g) if rank == 0 else torch.zeros(3262464, dtype=torch.uint8)
    vdist.broadcast_packed(blob, 0)
    expect = torch.randint(0, 256, (3262464,), dtype=torch.uint8, generator=torch.Generator().manual_seed(1234))
    assert torch.equal(blob, expect)
    # benchmark timing protocol: barrier, then MAX over ranks
    vdist.barrier()
    t = vdist.max_over_ranks(1.0 + rank)
    assert t == float(world)
    lo, hi = vdist.shard_range(64, rank, world)
    torch.save({"rank": rank, "span": (lo, hi), "max": t}, os.path.join(out_dir, f"r{rank}.pt"))
    vdist.barrier()
    torch.distributed.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(2)]
    assert [r["span"] for r in res] == [(0, 32), (32, 64)]
    assert all(r["max"] == 2.0 for r in res)
