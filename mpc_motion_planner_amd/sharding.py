"""Batch sharding over ranks: the problems of a batch are independent, so rank r of G owns the contiguous
slice [r*total/G, (r+1)*total/G) of the global seeded batch, solves it with no data-path collective, and ONE gather of
fixed-size result records [xs | us | T] to rank 0 reassembles the global result in order.

bench.py (N GPUs, RCCL) and tests/test_sharding_gloo.py (2 ranks, gloo, CPU tensors, the oracle as the stand-in solver)
run exactly this code.  `torch.distributed` is only touched when world > 1.
"""
import torch


def shard_bounds(rank, world, total):
    """contiguous slice [lo, hi) of rank `rank`; sizes differ by at most one when world does not divide total"""
    return rank * total // world, (rank + 1) * total // world


def global_total(scaling, batch, world):
    """weak scaling: `batch` problems PER rank; strong scaling: `batch` problems in the whole job"""
    if scaling not in ("weak", "strong"):
        raise ValueError("scaling must be 'weak' or 'strong'")
    return batch * world if scaling == "weak" else batch


class ShardedBatch:
    """Result buffers + the gather of one rank's shard.  record = 21*N + 1 doubles per problem."""

    def __init__(self, total, rank, world, N, device, dist=None, width=None):
        """width: doubles per record (default 21*N + 1 = one solution; the receding-horizon workload gathers final-state records instead)"""
        self.total, self.rank, self.world, self.N, self.dist = int(total), int(rank), int(world), int(N), dist
        self.lo, self.hi = shard_bounds(rank, world, total)
        self.count = self.hi - self.lo
        self.cap = max(shard_bounds(r, world, total)[1] - shard_bounds(r, world, total)[0] for r in range(world))
        self.width = int(width) if width is not None else 21 * N + 1
        self.sol = torch.zeros(self.cap, self.width, dtype=torch.float64, device=device)      # padded to the largest shard
        self.gathered = [torch.zeros_like(self.sol) for _ in range(world)] if (world > 1 and rank == 0) else None

    def gather_rows(self, rows):
        """rows [count][width] (a tensor on the buffers' device) -> record buffer, then the gather to rank 0"""
        self.sol[:self.count].copy_(rows[:self.count])
        if self.world > 1:
            self.dist.gather(self.sol, self.gathered, dst=0)

    def pack_and_gather(self, sol_x, sol_u, sol_T):
        """sol_x [count][N][14], sol_u [count][N][7], sol_T [count] (tensors on the buffers' device) -> record buffer, then the
        gather to rank 0 (the only collective of the path)."""
        c = self.count
        torch.cat([sol_x[:c].reshape(c, -1), sol_u[:c].reshape(c, -1), sol_T[:c, None]], dim=1, out=self.sol[:c])
        if self.world > 1:
            self.dist.gather(self.sol, self.gathered, dst=0)

    def assemble(self):
        """rank 0: the global result [total][width] in problem order"""
        if self.world == 1:
            return self.sol[:self.count]
        parts = []
        for r in range(self.world):
            lo, hi = shard_bounds(r, self.world, self.total)
            parts.append(self.gathered[r][:hi - lo])
        return torch.cat(parts)
