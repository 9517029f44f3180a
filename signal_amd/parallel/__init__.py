from .reducer import GradReducer, plan_buckets  # noqa: F401
