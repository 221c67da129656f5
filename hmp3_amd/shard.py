"""Stream sharding for multi-GPU runs: independent streams, contiguous blocks per rank,
no exchange step.  One process per GPU."""


def shard_range(nstreams, world, rank):
    """[first, last) of the streams rank owns; sizes differ by at most one"""
    base, rem = divmod(int(nstreams), int(world))
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def owner_of(stream, nstreams, world):
    for r in range(world):
        a, b = shard_range(nstreams, world, r)
        if a <= stream < b:
            return r
    raise IndexError(stream)
