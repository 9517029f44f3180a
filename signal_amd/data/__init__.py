from .pipeline import DevicePrefetcher, PKSampler, SyntheticTriplets, shard_for_rank  # noqa: F401
