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
    """Per-step all-gather of the ranks' decoder outputs into one [world * rows_per_rank] int32 tensor.

    The gather is asynchronous: the next decode step does not depend on the other ranks' tokens (only the host
    scheduler consumes them), so the collective runs on the communicator's stream beside the next step's
    kernels.  Callers alternate between `depth` input buffers (see buffer()) so a step never overwrites tokens
    that are still being gathered; wait() drains everything."""

    def __init__(self, rows_per_rank, world, device, depth=2):
        self.world = world
        self.depth = depth
        self.step = 0
        self.inputs = [torch.full((rows_per_rank,), -1, dtype=torch.int32, device=device) for _ in range(depth)]
        self.outputs = [torch.empty(world * rows_per_rank, dtype=torch.int32, device=device) for _ in range(depth)] \
            if world > 1 else None
        self.pending = [None] * depth

    def buffer(self):
        """The decoder-result buffer to fill in the current step (its previous gather has completed)."""
        slot = self.step % self.depth
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None
        return self.inputs[slot]

    def __call__(self):
        """Start gathering the buffer handed out by buffer() for this step."""
        slot = self.step % self.depth
        self.step += 1
        if self.world == 1:
            return self.inputs[slot]
        self.pending[slot] = dist.all_gather_into_tensor(self.outputs[slot], self.inputs[slot], async_op=True)
        return self.outputs[slot]

    def wait(self):
        for i, w in enumerate(self.pending):
            if w is not None:
                w.wait()
                self.pending[i] = None

    def latest(self):
        """Gathered tokens of the most recent step (after wait())."""
        slot = (self.step - 1) % self.depth
        return self.inputs[slot] if self.world == 1 else self.outputs[slot]
