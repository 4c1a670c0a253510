"""Frame-window sharding of one clip over the GPUs of a node (new component; the reference has no
multi-GPU inference -- /root/reference/Experimental_root/models/validation_seq_infer.py:24 notes it
"cannot work on multiple gpu").  SURVEY.md §8e.

Rank r owns a contiguous window of frames and runs the clip schedule on it.  The only cross-rank
dependency is the temporal shift: each of the 16 temporal-fusion convs needs ``fold`` channels of the
frame before its window (from rank r-1's last frame) and ``fold`` channels of the frame after it (from
rank r+1's first frame) of THAT layer's input.  ``HaloExchanger`` is the ``halo_fn`` of
schedule.bsvd_clip: it packs the two boundary slices with ``bsvd_halo_pack`` and swaps them with the
temporal neighbours using point-to-point send/recv (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests).  Neighbour-only traffic: no all-reduce, one xGMI link per direction.
The outermost ranks see zeros, exactly the stream start/end of the reference (bsvd_arch.py:94,104).
"""
import torch
import torch.distributed as dist

from .schedule import Halo


def shard_range(num_frames, world, rank):
    """Contiguous, balanced frame window [start, end) of ``rank``."""
    base, rem = divmod(num_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class _Pending:
    def __init__(self, reqs, hp, hn, staged=(), keep=()):
        # `keep` pins the packed send slices until finish(): the transfer runs on the backend's own stream
        self.reqs, self.hp, self.hn, self.staged, self.keep = reqs, hp, hn, staged, keep

    def finish(self):
        for req in self.reqs:
            req.wait()          # NCCL/RCCL: orders the current stream after the transfer, no host sync
        for host, dev in self.staged:          # host-staged backends (gloo has no device point-to-point)
            dev.copy_(host, non_blocking=False)
        self.keep = ()
        return self.hp, self.hn


class HaloExchanger:
    def __init__(self, ex, rank=None, world=None, group=None):
        self.ex = ex
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.bytes_sent = 0
        self.exchanges = 0
        # gloo implements send/recv for CPU tensors only: stage device slices through the host there (tests, or a box
        # without RCCL point-to-point).  RCCL ("nccl") moves device buffers directly over xGMI.
        self.host_staging = dist.get_backend(group) == "gloo"

    def __call__(self, spec, v):
        """v: [T,H,W,C] input of temporal-fusion layer ``spec`` on this rank -> (Halo|None, Halo|None)."""
        return self.start(spec, v).finish()

    def start(self, spec, v):
        """Packs and posts the sends/receives; returns a handle whose finish() yields the halos.  Between start()
        and finish() the caller convolves the interior frames, hiding the transfer (8.3 / 4.1 MB per message)."""
        fold = spec.fold
        has_left, has_right = self.rank > 0, self.rank + 1 < self.world
        if fold == 0 or not (has_left or has_right):
            return _Pending([], None, None)
        mk = getattr(self.ex, "empty_halo", None) or (lambda t, n: torch.empty(tuple(t.shape[1:3]) + (n,), dtype=t.dtype, device=t.device))
        raw = lambda t: getattr(t, "t", t) if not isinstance(t, torch.Tensor) else t      # the torch tensor behind a transformed slice
        ops, recv_prev, recv_next, staged, keep = [], None, None, [], []
        stage = self.host_staging and v.is_cuda

        def wire(t, receiving):
            """tensor handed to the backend: the device buffer itself, or a host mirror when staging"""
            if not stage:
                return raw(t)
            t = raw(t)
            h = torch.empty(t.shape, dtype=t.dtype, device="cpu") if receiving else t.cpu()
            if receiving:
                staged.append((h, t))
            return h

        if has_right:
            send_last = self.ex.halo_pack(v[-1], fold, fold)          # my last frame's [fold:2fold] -> right's halo_prev
            recv_next = mk(v, fold)
            ops += [dist.P2POp(dist.isend, wire(send_last, False), self._peer(self.rank + 1), self.group),
                    dist.P2POp(dist.irecv, wire(recv_next, True), self._peer(self.rank + 1), self.group)]
            self.bytes_sent += raw(send_last).numel() * 4
            keep.append(send_last)
        if has_left:
            send_first = self.ex.halo_pack(v[0], 0, fold)             # my first frame's [0:fold] -> left's halo_next
            recv_prev = mk(v, fold)
            ops += [dist.P2POp(dist.isend, wire(send_first, False), self._peer(self.rank - 1), self.group),
                    dist.P2POp(dist.irecv, wire(recv_prev, True), self._peer(self.rank - 1), self.group)]
            self.bytes_sent += raw(send_first).numel() * 4
            keep.append(send_first)
        reqs = dist.batch_isend_irecv(ops)
        self.exchanges += 1
        hp = None if recv_prev is None else Halo(recv_prev, fold, 0)
        hn = None if recv_next is None else Halo(recv_next, fold, 0)
        return _Pending(reqs, hp, hn, staged, keep)

    def _peer(self, group_rank):
        return group_rank if self.group is None else dist.get_global_rank(self.group, group_rank)
