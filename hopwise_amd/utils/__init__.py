from .enum_type import InputType, KGDataLoaderState, ModelType  # noqa: F401


def seed_for_rank(seed):
    """Philox key of this process: the configured seed with the rank folded in, so that the ranks of a data-parallel job
    draw different negatives / dropout masks at the same batch position (the reference seeds every rank with
    ``seed + local_rank``, quick_start.py run_hopwises; RecBole's DistributedSampler gives each rank different data).
    Rank 0 -- and any single-process run -- keeps ``seed`` itself."""
    import os

    rank = 0
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
        else:
            rank = int(os.environ.get("RANK", "0"))
    except Exception:
        rank = int(os.environ.get("RANK", "0"))
    return (int(seed) + rank * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
