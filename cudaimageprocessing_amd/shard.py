"""Multi-GPU decomposition: one process per GPU, torch.distributed (RCCL over xGMI
when the backend is "nccl"; "gloo" on CPU for tests).

* Batch sharding: frames are independent, so every rank takes a contiguous range of frames and
  nothing is exchanged (BASELINE.json config 3).
* Row-strip sharding of one large image (config 4): rank g owns rows [H*g/G, H*(g+1)/G).  A
  stencil of radius r needs `halo_rows(filter)` rows from each neighbour (r, plus one for the
  reference's flat-index wrap when r > 1).  One neighbour exchange (point-to-point send/recv,
  no collective) runs while the interior rows -- those that need no halo -- are already being
  filtered; the two edge bands follow.  The per-pixel arithmetic is the single-GPU kernel's, so
  the stitched result is bit-identical to a single-GPU run.

The reference has no multi-GPU code at all; this module is defined by BASELINE.json only.
"""
from . import _lib


def batch_range(n_frames, world, rank):
    """Contiguous frame range [start, stop) of `rank`: ceil(n/world) frames per rank."""
    per = -(-n_frames // world)
    start = min(n_frames, rank * per)
    return start, min(n_frames, start + per)


def strip_range(h, world, rank):
    """Row range [y0, y1) of `rank`'s strip."""
    return h * rank // world, h * (rank + 1) // world


def _dist():
    import torch.distributed as dist
    return dist


def exchange_halo(strip, halo, rank, world, group=None):
    """Send this strip's first/last `halo` rows to the upper/lower neighbour and receive theirs.
    Returns (top, bottom, works): top = the rows above the strip (None on rank 0), bottom = the
    rows below (None on the last rank); wait on every work in `works` before reading them."""
    import torch
    dist = _dist()
    ops, top, bottom = [], None, None
    n = strip.shape[0]
    if halo > n and world > 1:
        raise ValueError(f"strip of {n} rows is thinner than the halo ({halo} rows)")
    if halo == 0 or world == 1:
        return None, None, []
    if rank > 0:
        top = torch.empty((halo,) + tuple(strip.shape[1:]), dtype=strip.dtype, device=strip.device)
        ops.append(dist.P2POp(dist.isend, strip[:halo].contiguous(), rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, top, rank - 1, group))
    if rank < world - 1:
        bottom = torch.empty((halo,) + tuple(strip.shape[1:]), dtype=strip.dtype, device=strip.device)
        ops.append(dist.P2POp(dist.isend, strip[n - halo:].contiguous(), rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, bottom, rank + 1, group))
    works = dist.batch_isend_irecv(ops) if ops else []
    return top, bottom, works


def _device_compute(filter_id, ksize, sigma, w, h_total):
    from . import ops

    def compute(rows_in, in_row0, out, y_begin, y_end):
        ops.filter_strip(filter_id, rows_in, out, w, h_total, in_row0, y_begin, y_end, ksize, sigma)
    return compute


def strip_filter(strip, filter_id, w, h_total, rank, world, ksize=3, sigma=0.0, group=None, compute=None):
    """Filter this rank's strip (rows strip_range(h_total, world, rank) of the image, shape
    (rows, w, 4) uint8) and return the same rows of the filtered image.

    `compute(rows_in, in_row0, out, y_begin, y_end)` produces output rows [y_begin, y_end) into
    `out` from input rows starting at global row `in_row0`; the default is the HIP kernel
    (ops.filter_strip).  Tests on CPU pass an oracle-backed function; there is no CPU fallback
    in the product path."""
    import torch
    if compute is None:
        if not strip.is_cuda:
            raise RuntimeError("strip_filter needs CUDA/HIP tensors (no CPU fallback)")
        compute = _device_compute(filter_id, ksize, sigma, w, h_total)
    halo = _halo_rows(filter_id, ksize)
    y0, y1 = strip_range(h_total, world, rank)
    n = y1 - y0
    assert strip.shape[0] == n and strip.shape[1] == w, "strip does not match strip_range()"
    out = torch.empty_like(strip)
    top, bottom, works = exchange_halo(strip, halo, rank, world, group)

    a = halo if top is not None else 0
    b = halo if bottom is not None else 0
    if n >= 2 * halo and (a or b):
        # interior rows need only local data: filter them while the halos travel
        if n - a - b > 0:
            compute(strip, y0, out[a:n - b], y0 + a, y1 - b)
        for wk in works:
            wk.wait()
        if a:     # rows [y0, y0+halo) read input rows [y0-halo, y0+2*halo)
            compute(torch.cat([top, strip[:2 * halo]]), y0 - halo, out[:a], y0, y0 + a)
        if b:     # rows [y1-halo, y1) read input rows [y1-2*halo, y1+halo)
            compute(torch.cat([strip[n - 2 * halo:], bottom]), y1 - 2 * halo, out[n - b:], y1 - b, y1)
    else:
        # no neighbours, no halo, or a strip thinner than two halos: one piece
        for wk in works:
            wk.wait()
        parts = [t for t in (top, strip, bottom) if t is not None]
        rows_in = torch.cat(parts) if len(parts) > 1 else strip
        compute(rows_in, y0 - a, out, y0, y1)
    return out


def _halo_rows(filter_id, ksize):
    """Rows of halo per side (mirrors mi_filter_halo_rows; kept in Python so that host logic
    can be tested without the library)."""
    if filter_id == _lib.FILTER_GRAYSCALE:
        return 0
    if filter_id == _lib.FILTER_SOBEL:
        return 1
    r = ksize // 2
    return 1 if r <= 1 else r + 1


def strip_chain(strip, stages, w, h_total, rank, world, group=None, compute_factory=None):
    """A chain of filters on strips, one halo exchange per stage (each stage consumes the halo
    of the previous stage's output).  stages: list of (filter_id, ksize, sigma)."""
    cur = strip
    for filter_id, ksize, sigma in stages:
        compute = compute_factory(filter_id, ksize, sigma) if compute_factory else None
        cur = strip_filter(cur, filter_id, w, h_total, rank, world, ksize, sigma, group, compute)
    return cur
