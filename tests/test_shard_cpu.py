"""World-size-2/3 gloo tests (CPU) of the multi-GPU host logic in cudaimageprocessing_amd/shard.py:
frame/strip partitioning and the neighbour halo exchange.  The per-strip arithmetic is injected
from the CPU oracle here (tests may use the oracle; the product path uses the HIP kernels and has
no CPU fallback), so what is under test is that the exchanged halos are the right rows, in the
right place, with the right depth -- stitched strips must equal the oracle on the whole image."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cudaimageprocessing_amd import _lib, shard, synth as S

W, H = 40, 37


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def oracle_full(golden, img, filter_id, k, sigma, y0=None, y1=None):
    if filter_id == _lib.FILTER_GRAYSCALE:
        return golden.grayscale(img)
    if filter_id == _lib.FILTER_SOBEL:
        return golden.sobel(img)
    if filter_id == _lib.FILTER_MEAN:
        return golden.meanblur(img, k)
    if filter_id == _lib.FILTER_MEDIAN:
        return golden.medianblur(img, k)
    return golden.gaussian(img, k, sigma)


def oracle_compute_factory(h_total, w):
    """compute(rows_in, in_row0, out, y_begin, y_end) backed by the oracle: the rows present are
    pasted into an otherwise zero image, so a halo that is too shallow shows up as wrong pixels."""
    from oracle import golden

    def factory(filter_id, k, sigma):
        def compute(rows_in, in_row0, out, y_begin, y_end):
            full = np.zeros((h_total, w, 4), np.uint8)
            r = rows_in.numpy()
            full[in_row0:in_row0 + r.shape[0]] = r
            res = oracle_full(golden, full, filter_id, k, sigma)
            out.copy_(torch.from_numpy(res[y_begin:y_end].copy()))
        return compute
    return factory


def worker(rank, world, port, stages, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        y0, y1 = shard.strip_range(H, world, rank)
        # every rank generates only its own rows of the synthetic image (counter-based generator)
        strip = torch.from_numpy(S.synth_image(4, W, y1 - y0, first_idx=y0 * W).copy())
        out = shard.strip_chain(strip, stages, W, H, rank, world, compute_factory=oracle_compute_factory(H, W))
        gathered = [None] * world
        dist.all_gather_object(gathered, out.numpy())
        if rank == 0:
            q.put(np.concatenate(gathered))
    finally:
        dist.destroy_process_group()


def run_world(world, stages):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, stages, q)) for r in range(world)]
    for p in procs:
        p.start()
    result = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return result


CASES = [
    [(_lib.FILTER_SOBEL, 3, 0.0)],
    [(_lib.FILTER_MEDIAN, 5, 0.0)],
    [(_lib.FILTER_GAUSSIAN, 15, 2.5)],
    [(_lib.FILTER_GRAYSCALE, 0, 0.0), (_lib.FILTER_GAUSSIAN, 5, 1.0), (_lib.FILTER_MEDIAN, 3, 0.0), (_lib.FILTER_SOBEL, 3, 0.0)],
]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("stages", CASES, ids=["sobel", "median5", "gauss15", "chain"])
def test_strips_with_halo_exchange_equal_whole_image(golden, world, stages):
    img = S.synth_image(4, W, H)
    expect = img
    for filter_id, k, sigma in stages:
        expect = oracle_full(golden, expect, filter_id, k, sigma)
    got = run_world(world, stages)
    assert np.array_equal(got, expect)


def test_thin_strips_take_the_single_piece_path(golden):
    # 37 rows over 3 ranks = 12-13 rows per strip, Gaussian 15x15 needs 8 halo rows: 2*8 > 12
    stages = [(_lib.FILTER_GAUSSIAN, 15, 2.5)]
    got = run_world(3, stages)
    assert np.array_equal(got, golden.gaussian(S.synth_image(4, W, H), 15, 2.5))


def test_partitions():
    assert [shard.batch_range(4096, 8, r) for r in (0, 1, 7)] == [(0, 512), (512, 1024), (3584, 4096)]
    assert [shard.batch_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard.batch_range(2, 4, 3) == (2, 2)
    rows = [shard.strip_range(32768, 8, r) for r in range(8)]
    assert rows[0] == (0, 4096) and rows[7] == (28672, 32768)
    assert all(rows[i][1] == rows[i + 1][0] for i in range(7))
    cover = [shard.strip_range(37, 3, r) for r in range(3)]
    assert cover[0][0] == 0 and cover[-1][1] == 37 and all(cover[i][1] == cover[i + 1][0] for i in range(2))
    for f, k in ((_lib.FILTER_GRAYSCALE, 0), (_lib.FILTER_SOBEL, 3), (_lib.FILTER_MEAN, 3), (_lib.FILTER_MEDIAN, 5),
                 (_lib.FILTER_GAUSSIAN, 15), (_lib.FILTER_GAUSSIAN, 3)):
        assert shard._halo_rows(f, k) == _lib.lib().mi_filter_halo_rows(f, k)
    # an image narrower than the radius: the flat-index taps reach ceil(r / w) rows further, not one
    assert shard._halo_rows(_lib.FILTER_MEAN, 17, 4) == 10 and shard._halo_rows(_lib.FILTER_MEAN, 17, 8) == 9
    assert shard._halo_rows(_lib.FILTER_MEAN, 31, 3) == 20 and shard._halo_rows(_lib.FILTER_GAUSSIAN, 15, 3) == 10
    assert shard._halo_rows(_lib.FILTER_MEDIAN, 5, 3) == 3 and shard._halo_rows(_lib.FILTER_SOBEL, 3, 3) == 1
