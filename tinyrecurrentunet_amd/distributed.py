"""MI355X-native ``distributed``: data-parallel gradient all-reduce with the reference's entry points
(``/root/reference/distributed.py``: ``reduce_tensor`` :42-46, ``init_distributed`` :48-58,
``apply_gradient_allreduce`` :95-147), one process per GPU over ``torch.distributed`` ("nccl" = RCCL on ROCm,
xGMI on one node).

Differences in mechanism, not in results: the reference concatenates ~100 gradient tensors, all-reduces the
copy and copies 100 slices back every step (:127-134), after 177 separate broadcasts at start (:105-108).
Here the TRU-Net backward already leaves every gradient as a view of ONE flat fp32 tensor (engine._wg_finish), so the
SUM all-reduce runs on that tensor in place (1.19 MB: latency-bound on xGMI, so exactly one collective per step) and
nothing is packed or scattered; gradients of any other module are packed by one multi-tensor op into a persistent
flat buffer first.  Parameters/buffers are broadcast as one flat tensor per dtype.  As in the reference,
parameters whose ``grad is None`` (the never-executed TGRU, R4) are skipped, BatchNorm is not synchronised
and the loss all-reduce is only for logging."""
import torch
import torch.distributed as dist
from torch.autograd import Variable

from . import _lib


def reduce_tensor(tensor, num_gpus):
    """distributed.py:42-46."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= num_gpus
    return rt


def init_distributed(rank, num_gpus, group_name, dist_backend, dist_url):
    """distributed.py:48-58 (``group_name`` is accepted and, as in current torch, unused)."""
    assert torch.cuda.is_available(), "Distributed mode requires a GPU."
    torch.cuda.set_device(rank % torch.cuda.device_count())
    dist.init_process_group(dist_backend, init_method=dist_url, world_size=num_gpus, rank=rank)


def broadcast_state(module, src=0):
    """One broadcast per dtype instead of one per tensor (distributed.py:105-108)."""
    by_dtype = {}
    for t in module.state_dict().values():
        if torch.is_tensor(t):
            by_dtype.setdefault(t.dtype, []).append(t)
    for tensors in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src)
        off = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


class FlatGradAllReduce:
    """Persistent flat gradient bucket; ``reduce()`` = pack, one all-reduce(SUM), /world, unpack."""

    def __init__(self, module):
        self.module = module
        self.flat = None
        self.params = None
        self.in_place = None      # True when the last reduce ran on the engine's flat gradient tensor itself
        self.timing = None        # a list here makes reduce() append (start, end) events around the exchange step

    @staticmethod
    def _mean_all_reduce(flat):
        """SUM over ranks and / world (distributed.py:127-134).  RCCL divides inside the collective (ncclAvg: the
        operands are pre-multiplied by 1/world, exact for the power-of-two rank counts of one node), so the exchange
        step is ONE launch; gloo has no AVG, there the division is a second pass."""
        if dist.get_backend() == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat /= dist.get_world_size()

    @staticmethod
    def _engine_flat(params):
        info = _lib.flat_grad_of(params[0].grad)
        if info is None:
            return None
        flat, layout, _ = info
        if len(layout) != len(params):
            return None
        base = flat.data_ptr()
        for p in params:
            o = layout.get(id(p))
            if o is None or p.grad.data_ptr() != base + 4 * o:
                return None
        return flat

    def reduce(self):
        params = [p for p in self.module.parameters() if p.requires_grad and p.grad is not None]
        if not params:
            return
        if self.timing is not None and params[0].is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            self._reduce(params)
            ev[1].record()
            self.timing.append(ev)
        else:
            self._reduce(params)

    def _reduce(self, params):
        flat = self._engine_flat(params)
        if flat is not None:               # every p.grad is a view of this tensor: reduce it where it lies
            self._mean_all_reduce(flat)
            self.in_place = True
            return
        self.in_place = False
        key = tuple(id(p) for p in params)
        n = sum(p.numel() for p in params)
        if self.flat is None or self.params != key or self.flat.numel() != n:
            self.flat = torch.empty(n, device=params[0].device, dtype=params[0].grad.dtype)
            self.params = key
            self.views = []
            off = 0
            for p in params:
                self.views.append(self.flat[off:off + p.numel()].view_as(p))
                off += p.numel()
        grads = [p.grad for p in params]
        torch._foreach_copy_(self.views, grads)
        self._mean_all_reduce(self.flat)
        torch._foreach_copy_(grads, self.views)


def apply_gradient_allreduce(module):
    """distributed.py:95-147: broadcast rank 0's state, then all-reduce (mean) the gradients once per
    backward, from an autograd-engine callback armed by a forward hook."""
    broadcast_state(module, 0)
    bucket = FlatGradAllReduce(module)
    module._grad_bucket = bucket
    module.needs_reduction = False

    def allreduce_params():
        if module.needs_reduction:
            module.needs_reduction = False
            bucket.reduce()

    def allreduce_hook(*unused):
        Variable._execution_engine.queue_callback(allreduce_params)

    for param in list(module.parameters()):
        if param.requires_grad:
            param.register_hook(allreduce_hook)

    def set_needs_reduction(self, input, output):
        self.needs_reduction = True

    module.register_forward_hook(set_needs_reduction)
    return module
