"""Row-slab partitioning of the image across ranks and the neighbour halo exchange.

The filter window of a pixel reaches b = (box-1)/2 rows up and down (rpf.cpp:561-571) and nothing else is
shared between pixels, so the image shards into contiguous row slabs, one per GPU / process, and the only
data-path communication is a send/recv of b boundary rows with the rank above and the rank below
(`torch.distributed` P2P: RCCL over xGMI on the GPUs, gloo in the CPU tests).  Feature planes never change
between passes, so their halo is exchanged once; colours change every pass and are re-exchanged before
each pass.  Planes are laid out [plane][row][x][s], so a halo is one contiguous span per plane.
"""
from collections import namedtuple

Slab = namedtuple("Slab", "row0 row1 halo_top halo_bottom")  # owned image rows [row0,row1); halo rows held


def partition_rows(H, world, rank):
    """contiguous, near-equal split of H rows: rank r owns [r*H//world, (r+1)*H//world)"""
    return (rank * H) // world, ((rank + 1) * H) // world


def slab_for(H, world, rank, halo):
    """rank's slab; every rank must own at least `halo` rows, because a neighbour's halo is filled from this rank's
    OWNED boundary rows only (a thinner slab would have to forward rows it does not own)"""
    a, b = partition_rows(H, world, rank)
    if world > 1 and b - a < halo:
        raise ValueError("row slab of rank %d has %d rows, fewer than the %d halo rows its neighbours need: "
                         "use fewer ranks (H = %d, world = %d)" % (rank, b - a, halo, H, world))
    return Slab(a, b, min(halo, a), min(halo, H - b))


def buffer_rows(slab):
    """rows present in the rank's buffers, and the [row_begin,row_end) range the rank filters"""
    n = slab.row1 - slab.row0
    return slab.halo_top + n + slab.halo_bottom, slab.halo_top, slab.halo_top + n


def exchange_halo(t, slab, rank, world, group=None):
    """Refresh the halo rows of ``t`` ([planes, H_buf, W, S], any dtype) from the neighbouring ranks' owned
    boundary rows.  Collective over (rank-1, rank, rank+1); a no-op for world == 1."""
    import torch.distributed as dist
    if world == 1:
        return
    n_own = slab.row1 - slab.row0
    top0 = slab.halo_top  # first owned buffer row
    assert n_own >= max(slab.halo_top, slab.halo_bottom), "slab thinner than its halo (see slab_for)"
    ops, recvs = [], []
    # gloo moves host memory only: device tensors are staged through the host (CPU tests, single-GPU rehearsals of the
    # multi-rank path); nccl (= RCCL) sends device memory directly over xGMI
    stage = t.is_cuda and dist.get_backend(group) == "gloo"
    wire = (lambda x: x.cpu()) if stage else (lambda x: x)
    if rank > 0 and slab.halo_top > 0:
        h = slab.halo_top
        send_up = wire(t[:, top0:top0 + h].contiguous())    # my first h owned rows -> bottom halo of rank-1
        recv_up = wire(t.new_empty((t.shape[0], h) + tuple(t.shape[2:])))
        ops += [dist.P2POp(dist.isend, send_up, rank - 1, group), dist.P2POp(dist.irecv, recv_up, rank - 1, group)]
        recvs.append((recv_up, 0, h))
    if rank < world - 1 and slab.halo_bottom > 0:
        h = slab.halo_bottom
        send_dn = wire(t[:, top0 + n_own - h:top0 + n_own].contiguous())  # my last h owned rows -> top halo of rank+1
        recv_dn = wire(t.new_empty((t.shape[0], h) + tuple(t.shape[2:])))
        ops += [dist.P2POp(dist.isend, send_dn, rank + 1, group), dist.P2POp(dist.irecv, recv_dn, rank + 1, group)]
        recvs.append((recv_dn, top0 + n_own, h))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    for buf, r0, h in recvs:
        t[:, r0:r0 + h].copy_(buf)  # (host -> device when staged)


class HaloPlan:
    """exchange_halo() with its send / receive staging buffers allocated once (nothing is allocated inside a timed
    loop): ``plan = HaloPlan(t, slab, rank, world)``, then ``plan.exchange(t)`` before every pass."""

    def __init__(self, t, slab, rank, world, group=None):
        import torch.distributed as dist
        self.slab, self.rank, self.world, self.group = slab, rank, world, group
        self.stage = world > 1 and t.is_cuda and dist.get_backend(group) == "gloo"
        n_own = slab.row1 - slab.row0
        assert world == 1 or n_own >= max(slab.halo_top, slab.halo_bottom), "slab thinner than its halo (see slab_for)"

        def buf(h):
            shape = (t.shape[0], h) + tuple(t.shape[2:])
            return t.new_empty(shape, device="cpu") if self.stage else t.new_empty(shape)

        self.up = (buf(slab.halo_top), buf(slab.halo_top)) if world > 1 and rank > 0 and slab.halo_top > 0 else None
        self.dn = (buf(slab.halo_bottom), buf(slab.halo_bottom)) if world > 1 and rank < world - 1 and slab.halo_bottom > 0 else None

    def exchange(self, t):
        if self.world == 1:
            return
        import torch.distributed as dist
        slab, rank, group = self.slab, self.rank, self.group
        n_own, top0 = slab.row1 - slab.row0, slab.halo_top
        ops = []
        if self.up is not None:
            h = slab.halo_top
            self.up[0].copy_(t[:, top0:top0 + h])                       # my first h owned rows -> bottom halo of rank-1
            ops += [dist.P2POp(dist.isend, self.up[0], rank - 1, group), dist.P2POp(dist.irecv, self.up[1], rank - 1, group)]
        if self.dn is not None:
            h = slab.halo_bottom
            self.dn[0].copy_(t[:, top0 + n_own - h:top0 + n_own])       # my last h owned rows -> top halo of rank+1
            ops += [dist.P2POp(dist.isend, self.dn[0], rank + 1, group), dist.P2POp(dist.irecv, self.dn[1], rank + 1, group)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.up is not None:
            t[:, 0:slab.halo_top].copy_(self.up[1])
        if self.dn is not None:
            t[:, top0 + n_own:top0 + n_own + slab.halo_bottom].copy_(self.dn[1])
