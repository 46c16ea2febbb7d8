"""Scheduler registry (scheduler/__init__.py:10-19 of the reference)."""
from .base import QueueTransport, Scheduler, ZmqTransport, encode_request  # noqa: F401

SCHEDULER_REGISTRY = {"base": Scheduler}


def load_scheduler(scheduler_type: str = "base", **kwargs):
    if scheduler_type not in SCHEDULER_REGISTRY:
        raise ValueError(f"Unknown scheduler type: {scheduler_type}. Available: {sorted(SCHEDULER_REGISTRY)}")
    return SCHEDULER_REGISTRY[scheduler_type](**kwargs)
