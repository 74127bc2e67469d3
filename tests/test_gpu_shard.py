"""Row-strip sharding on the device path: fresh child processes (one per rank, "gloo" rendezvous, all on
this box's one GPU) run shard.StripPlan with its default compute -- the HIP kernels through
mi_filter_strip_u8x4_dev -- including the halo exchange into the padded strip buffers.  The stitched result
must equal the single launch on the whole image and the oracle.  (On an 8-GPU node the same code runs one
rank per GPU over RCCL; gloo only changes how the halo rows travel.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

from cudaimageprocessing_amd import _lib, ops, shard, synth as S  # noqa: E402

W, H = 1024, 203
G15 = (_lib.FILTER_GAUSSIAN, 15, 2.5)
CASES = {
    "sobel": [(_lib.FILTER_SOBEL, 3, 0.0)],
    "median5": [(_lib.FILTER_MEDIAN, 5, 0.0)],
    "gauss15": [G15],
    "c2_chain": [(_lib.FILTER_GRAYSCALE, 0, 0.0), (_lib.FILTER_GAUSSIAN, 5, 1.0), (_lib.FILTER_MEDIAN, 3, 0.0),
                 (_lib.FILTER_SOBEL, 3, 0.0)],
    "c4_pair": [G15, (_lib.FILTER_MEDIAN, 5, 0.0)],
}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda:0")
        y0, y1 = shard.strip_range(H, world, rank)
        results = {}
        for name, stages in CASES.items():
            pad = max(shard._halo_rows(f, k) for f, k, _ in stages)
            plan = shard.StripPlan(W, H, rank, world, pad, dev)
            # every rank generates only its own rows (counter-based generator), straight into the plan
            plan.strip().copy_(ops.synth(4, W, y1 - y0, first_idx=y0 * W, device=dev))
            # poison the pad rows: a stage that reads a halo row it did not receive shows up as wrong pixels
            plan.bufs[0][:pad].fill_(0xA5)
            plan.bufs[0][pad + plan.n:].fill_(0xA5)
            plan.bufs[1].fill_(0x5A)
            out = plan.run_chain(stages)
            torch.cuda.synchronize()
            results[name] = out.cpu().numpy()
        gathered = [None] * world
        dist.all_gather_object(gathered, results)
        if rank == 0:
            q.put({name: np.concatenate([g[name] for g in gathered]) for name in CASES})
    finally:
        dist.destroy_process_group()


def oracle_chain(golden, img, stages):
    for f, k, s in stages:
        if f == _lib.FILTER_GRAYSCALE:
            img = golden.grayscale(img)
        elif f == _lib.FILTER_SOBEL:
            img = golden.sobel(img)
        elif f == _lib.FILTER_MEDIAN:
            img = golden.medianblur(img, k)
        elif f == _lib.FILTER_GAUSSIAN:
            img = golden.gaussian(img, k, s)
    return img


@pytest.mark.parametrize("world", [2, 3])
def test_strip_plan_device_path_in_child_processes(world, golden):
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    # take the result BEFORE joining (rank 0 blocks in put() until it is read), and give up as soon as
    # a rank has died without delivering one
    import queue
    stitched = None
    for _ in range(30):
        try:
            stitched = q.get(timeout=5)
            break
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(30)
        if p.is_alive():
            p.kill()
    assert stitched is not None and all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    img = S.synth_image(4, W, H)
    d = torch.from_numpy(img).cuda()
    single = {
        "sobel": ops.sobel(d),
        "median5": ops.median_blur(d, 5),
        "gauss15": ops.gaussian_blur(d, 15, 2.5),
        "c2_chain": ops.fused_chain(d, 1.0),
        "c4_pair": ops.median_blur(ops.gaussian_blur(d, 15, 2.5), 5),
    }
    for name, stages in CASES.items():
        assert np.array_equal(stitched[name], single[name].cpu().numpy()), name    # strips == one launch, bit for bit
        want = oracle_chain(golden, img, stages)
        if any(f == _lib.FILTER_GAUSSIAN for f, _, _ in stages):
            if name == "gauss15":      # separable Gaussian: within 1 LSB of the reference's sum
                assert np.abs(stitched[name].astype(np.int16) - want.astype(np.int16)).max() <= 1
        else:
            assert np.array_equal(stitched[name], want), name
