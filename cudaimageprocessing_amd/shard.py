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


def _p2p(sends, recvs, group=None):
    """Neighbour send/recv (no collective).  sends/recvs: lists of (tensor, peer).  Returns a function
    that waits for the transfers (and finishes the receives) when called.  With the "nccl" backend
    (RCCL over xGMI) device tensors travel as they are.  "gloo" moves host memory only, so device
    tensors are staged through host copies -- that is the rehearsal mode for boxes with fewer GPUs than
    ranks (tests, MI_BENCH_BACKEND=gloo), not the production transport."""
    dist = _dist()
    if not sends and not recvs:
        return lambda: None
    staged = dist.get_backend(group) == "gloo" and any(t.is_cuda for t, _ in sends + recvs)
    if staged:
        h_send = [(t.cpu(), peer) for t, peer in sends]
        h_recv = [(t.new_empty(t.shape, device="cpu"), peer) for t, peer in recvs]
    else:
        h_send, h_recv = sends, recvs
    ops = [dist.P2POp(dist.isend, t, peer, group) for t, peer in h_send]
    ops += [dist.P2POp(dist.irecv, t, peer, group) for t, peer in h_recv]
    works = dist.batch_isend_irecv(ops)

    def wait():
        for wk in works:
            wk.wait()
        if staged:
            for (dst, _), (src, _) in zip(recvs, h_recv):
                dst.copy_(src)
    return wait


def _device_compute(filter_id, ksize, sigma, w, h_total):
    from . import ops

    def compute(rows_in, in_row0, out, y_begin, y_end):
        ops.filter_strip(filter_id, rows_in, out, w, h_total, in_row0, y_begin, y_end, ksize, sigma)
    return compute


class StripPlan:
    """This rank's strip of a w x h_total image with everything a chain of stencil stages needs,
    allocated once: two buffers of (pad + rows + pad) image rows used in turn as a stage's input and
    output.  The strip sits in the middle of a buffer; a stage's halo rows are received straight into
    the `pad` rows around it, so the kernels read strip and halo as one contiguous run of rows -- no
    per-stage allocation, concatenation or copy.  pad = the deepest halo any stage will ask for."""

    def __init__(self, w, h_total, rank, world, pad, device, group=None):
        import torch
        self.w, self.h_total, self.rank, self.world, self.group = w, h_total, rank, world, group
        self.y0, self.y1 = strip_range(h_total, world, rank)
        self.n = self.y1 - self.y0
        self.pad = int(pad)
        self.bufs = [torch.empty((self.n + 2 * self.pad, w, 4), dtype=torch.uint8, device=device) for _ in range(2)]
        self.cur = 0

    def strip(self, i=None):
        """The strip's rows inside buffer i (default: the current one) -- fill this with the input."""
        b = self.bufs[self.cur if i is None else i]
        return b[self.pad:self.pad + self.n]

    def run_stage(self, filter_id, ksize=3, sigma=0.0, compute=None):
        """One filter over the strip: neighbour exchange of `halo` rows, interior rows filtered while
        the halos travel, then the two edge bands.  Returns the output strip (a view of the other
        buffer, which becomes the current one)."""
        halo = _halo_rows(filter_id, ksize, self.w)
        if halo > self.pad:
            raise ValueError(f"stage needs {halo} halo rows, the plan was made for {self.pad}")
        src, dst = self.bufs[self.cur], self.bufs[self.cur ^ 1]
        if compute is None:
            if not src.is_cuda:
                raise RuntimeError("strip filtering needs CUDA/HIP tensors (no CPU fallback)")
            compute = _device_compute(filter_id, ksize, sigma, self.w, self.h_total)
        n, pad, y0, y1, rank, world = self.n, self.pad, self.y0, self.y1, self.rank, self.world
        if halo > n and world > 1:
            raise ValueError(f"strip of {n} rows is thinner than the halo ({halo} rows)")
        up, down = (rank > 0 and halo > 0), (rank < world - 1 and halo > 0)
        sends, recvs = [], []
        if up:
            sends.append((src[pad:pad + halo], rank - 1))
            recvs.append((src[pad - halo:pad], rank - 1))
        if down:
            sends.append((src[pad + n - halo:pad + n], rank + 1))
            recvs.append((src[pad + n:pad + n + halo], rank + 1))
        wait = _p2p(sends, recvs, self.group)
        a, b = (halo if up else 0), (halo if down else 0)
        out = dst[pad:pad + n]
        if n >= 2 * halo and (a or b):
            if n - a - b > 0:   # interior rows need only local data: filter them while the halos travel
                compute(src[pad:pad + n], y0, out[a:n - b], y0 + a, y1 - b)
            wait()
            if a:     # rows [y0, y0+halo) read input rows [y0-halo, y0+2*halo)
                compute(src[pad - halo:pad + 2 * halo], y0 - halo, out[:a], y0, y0 + a)
            if b:     # rows [y1-halo, y1) read input rows [y1-2*halo, y1+halo)
                compute(src[pad + n - 2 * halo:pad + n + halo], y1 - 2 * halo, out[n - b:], y1 - b, y1)
        else:         # no neighbours, no halo, or a strip thinner than two halos: one piece
            wait()
            compute(src[pad - a:pad + n + b], y0 - a, out, y0, y1)
        self.cur ^= 1
        return out

    def run_chain(self, stages, compute_factory=None):
        """One halo exchange per stage (each stage consumes the halo of the previous stage's output).
        stages: list of (filter_id, ksize, sigma)."""
        out = self.strip()
        for filter_id, ksize, sigma in stages:
            compute = compute_factory(filter_id, ksize, sigma) if compute_factory else None
            out = self.run_stage(filter_id, ksize, sigma, compute)
        return out


def strip_filter(strip, filter_id, w, h_total, rank, world, ksize=3, sigma=0.0, group=None, compute=None):
    """Filter this rank's strip (rows strip_range(h_total, world, rank) of the image, shape
    (rows, w, 4) uint8) and return the same rows of the filtered image (convenience form: makes a
    one-stage StripPlan and copies the strip in; pipelines keep a plan and fill plan.strip()).

    `compute(rows_in, in_row0, out, y_begin, y_end)` produces output rows [y_begin, y_end) into
    `out` from input rows starting at global row `in_row0`; the default is the HIP kernel
    (ops.filter_strip).  Tests on CPU pass an oracle-backed function; there is no CPU fallback
    in the product path."""
    return strip_chain(strip, [(filter_id, ksize, sigma)], w, h_total, rank, world, group,
                       (lambda f, k, s: compute) if compute else None)


def _halo_rows(filter_id, ksize, w=None):
    """Rows of halo per side (mirrors mi_filter_halo_rows; kept in Python so that host logic
    can be tested without the library).  `w`: image width, which matters only below the radius (the taps are
    flat indices, so r pixels to the side reach ceil(r / w) rows further)."""
    if filter_id == _lib.FILTER_GRAYSCALE:
        return 0
    if filter_id == _lib.FILTER_SOBEL:
        return 1
    r = ksize // 2
    if r <= 1:
        return 1
    return r + (1 if w is None or w >= r else -(-r // w))


def strip_chain(strip, stages, w, h_total, rank, world, group=None, compute_factory=None):
    """A chain of filters on strips, one halo exchange per stage.  stages: list of
    (filter_id, ksize, sigma).  Convenience form of StripPlan.run_chain."""
    if compute_factory is None and not strip.is_cuda:
        raise RuntimeError("strip filtering needs CUDA/HIP tensors (no CPU fallback)")
    y0, y1 = strip_range(h_total, world, rank)
    assert strip.shape[0] == y1 - y0 and strip.shape[1] == w, "strip does not match strip_range()"
    pad = max([_halo_rows(f, k, w) for f, k, _ in stages] + [0])
    plan = StripPlan(w, h_total, rank, world, pad, strip.device, group)
    plan.strip().copy_(strip)
    return plan.run_chain(stages, compute_factory)
