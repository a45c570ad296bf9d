"""Cross-rank scalar reductions with the reference's autograd contract
(ref:distributed/torchshard.py:18-41): forward = in-place all-reduce SUM over all ranks
(RCCL on MI355X: the "nccl" backend of torch.distributed), backward = identity, so each rank
back-propagates local_loss_sum / global_token_count."""
import torch

from .comm import current_comm


def _world_size() -> int:
    return current_comm().world


def _reduce(tensor: torch.Tensor) -> torch.Tensor:
    comm = current_comm()          # torch.distributed (RCCL) unless the calling thread installed a test group
    if comm.world == 1:
        return tensor
    return comm.all_reduce(tensor)


class Reduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input: torch.Tensor) -> torch.Tensor:
        ctx.mark_dirty(input)
        return _reduce(input)

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor) -> torch.Tensor:
        return grad_output


def reduce_sum(input: torch.Tensor) -> torch.Tensor:
    if torch.is_grad_enabled() and input.requires_grad:
        # all_reduce works in place; give autograd a fresh tensor to mark dirty
        return Reduce.apply(input.clone())
    return _reduce(input)


def reduce_sum_pair(loss_sum: torch.Tensor, num_tokens: torch.Tensor):
    """Both scalars of ref:engine:309-310 in ONE 2-element all-reduce (fp64 keeps the token
    count exact).  Same values and the same identity backward as two reduce_sum calls."""
    if _world_size() == 1:
        return loss_sum, num_tokens
    buf = torch.stack([loss_sum.detach().double(), num_tokens.double()])
    current_comm().all_reduce(buf)
    total = buf[0].to(loss_sum.dtype)
    # value = global sum, gradient = identity w.r.t. the local sum
    loss_out = loss_sum + (total - loss_sum.detach())
    return loss_out, buf[1].round().to(num_tokens.dtype)
