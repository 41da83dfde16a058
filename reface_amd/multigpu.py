"""Multi-GPU plumbing of the sharded test-bench path (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

The path shards over (source, target) pairs (inference_test_bench.py:410: the batch loop has no cross-sample dependency), so
the only exchange is at start-up: rank 0 holds the weights and broadcasts them ONCE, as one flat buffer per dtype (a handful
of large RCCL broadcasts instead of ~2000 small ones); there is no collective in the step loop.  Used by bench.py and
scripts/inference_test_bench.py; exercised on CPU tensors with the gloo backend by tests/test_host_cpu.py.
"""
import torch


def broadcast_tensors(tensors, src=0, chunk_bytes=1 << 30):
    """In-place broadcast of a list of tensors from `src`: tensors are grouped by (dtype, device), packed into flat buffers of
    at most `chunk_bytes`, broadcast, and unpacked on the receivers.  Returns the number of collectives issued."""
    import torch.distributed as dist
    groups = {}
    for t in tensors:
        groups.setdefault((t.dtype, t.device), []).append(t)
    rank = dist.get_rank()
    calls = 0
    for (dt, dev), ts in groups.items():
        es = torch.empty((), dtype=dt).element_size()
        i = 0
        while i < len(ts):
            j, n = i, 0
            while j < len(ts) and (j == i or (n + ts[j].numel()) * es <= chunk_bytes):
                n += ts[j].numel()
                j += 1
            flat = torch.empty(n, dtype=dt, device=dev)
            if rank == src:
                off = 0
                for t in ts[i:j]:
                    flat[off:off + t.numel()].copy_(t.reshape(-1))
                    off += t.numel()
            dist.broadcast(flat, src)
            calls += 1
            if rank != src:
                off = 0
                for t in ts[i:j]:
                    t.copy_(flat[off:off + t.numel()].view_as(t))
                    off += t.numel()
            del flat
            i = j
    return calls


def broadcast_module(module, src=0):
    """Parameters and buffers of `module` from rank `src` to every rank."""
    return broadcast_tensors([p.data for p in module.parameters()] + [b.data for b in module.buffers()], src)


def max_over_ranks(seconds, device="cpu"):
    """The job's elapsed time is the slowest rank's: one scalar all-reduce (MAX) after the timed region."""
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_indices(n, rank, world):
    """Pairs of rank r: r, r + world, r + 2*world, ... (no data-path collective)."""
    return list(range(rank, n, world))
