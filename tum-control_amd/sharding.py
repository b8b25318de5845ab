"""
sharding.py -- multi-GPU layout of the batch dimension.

OCP instances are independent, so the batch is cut into contiguous blocks, one per rank (one
process per GPU); there is no collective inside a solve. The only exchange is the rooted gather
of the per-instance result slabs to rank 0 (RCCL over xGMI on GPUs, gloo in the CPU tests):
payload (u0[2], cost, status, qp_iter) as 5 doubles = 40 B per instance, ONE collective per step.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous block [lo, hi) of `total` instances owned by `rank`; the first total % world ranks
    get one extra instance."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_groups(n_groups, group_size, world, rank):
    """Group-aligned block of a batch made of `n_groups` scenario groups of `group_size` instances each (configs 3/4: one
    pose with its sigma points / Monte-Carlo draws): ranks own whole groups, so the PCE moment reduction of a group never
    needs another GPU. Returns (g_lo, g_hi, b_lo, b_hi): group range and instance range of `rank`."""
    g_lo, g_hi = shard_range(n_groups, world, rank)
    return g_lo, g_hi, g_lo * int(group_size), g_hi * int(group_size)


def shard_sizes(total, world):
    return [shard_range(total, world, r)[1] - shard_range(total, world, r)[0] for r in range(world)]


class ResultGatherer:
    """Rooted gather of equally sized result slabs; buffers allocated once (no per-step allocation)."""

    def __init__(self, world, rank, per_rank, device, nf=3, ni=2, root=0):
        self.world, self.rank, self.root = world, rank, root
        self.is_root = rank == root
        if self.is_root:
            self.all_f = torch.zeros((world, per_rank, nf), dtype=torch.float64, device=device)
            self.all_i = torch.zeros((world, per_rank, ni), dtype=torch.int32, device=device)
            self._lf = list(self.all_f.unbind(0))
            self._li = list(self.all_i.unbind(0))
        else:
            self.all_f = self.all_i = None
            self._lf = self._li = None

    def gather(self, res_f, res_i=None):
        dist.gather(res_f, self._lf, dst=self.root)
        if res_i is not None:
            dist.gather(res_i, self._li, dst=self.root)
        return self.all_f, self.all_i
