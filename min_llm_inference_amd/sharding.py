"""Row sharding of the continuous batch across the GPUs of one node.

Every batch row is independent in all kernels of the path, so rank r owns rows [r*B/N, (r+1)*B/N) with its own
lengths, page table and page pool; weights are replicated.  The only exchange per decode step is the all-gather
of the generated token ids (int32, B/N per rank) through torch.distributed -- backend "nccl" (= RCCL over xGMI)
on GPUs, "gloo" in the CPU tests.  No KV page ever crosses a GPU boundary.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, rank, world):
    """Contiguous, balanced split of n_rows batch rows; the first n_rows % world ranks get one extra row."""
    base, extra = divmod(n_rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_items(items, rank, world):
    """Items are dealt round-robin so every rank's queue has the same length mix."""
    return items[rank::world]


class TokenGather:
    """Per-step all-gather of the ranks' decoder outputs into one [world * rows_per_rank] int32 tensor."""

    def __init__(self, rows_per_rank, world, device):
        self.world = world
        self.out = torch.empty(world * rows_per_rank, dtype=torch.int32, device=device) if world > 1 else None

    def __call__(self, local_tokens):
        if self.world == 1:
            return local_tokens
        dist.all_gather_into_tensor(self.out, local_tokens.contiguous().view(-1))
        return self.out
