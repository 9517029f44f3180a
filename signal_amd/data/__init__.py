from .pipeline import DevicePrefetcher, PKSampler, SyntheticTriplets, epoch_seed, shard_for_rank  # noqa: F401
