"""Scheduler registry (drop-in for /root/reference/vox_serve/scheduler/__init__.py:10-64)."""
from typing import Dict, Type

from .base import QueueTransport, Scheduler, ZmqTransport, encode_request  # noqa: F401
from .disaggregation import DisaggregationScheduler
from .input_streaming import InputStreamingScheduler
from .offline import OfflineScheduler
from .online import OnlineScheduler

SCHEDULER_REGISTRY: Dict[str, Type[Scheduler]] = {
    "base": Scheduler, "online": OnlineScheduler, "offline": OfflineScheduler,
    "disaggregation": DisaggregationScheduler, "input_streaming": InputStreamingScheduler,
}


def load_scheduler(scheduler_type: str = "base", **kwargs) -> Scheduler:
    key = scheduler_type.lower()
    if key not in SCHEDULER_REGISTRY:
        raise ValueError(f"Unsupported scheduler type '{scheduler_type}'. Available types: {list(SCHEDULER_REGISTRY)}")
    return SCHEDULER_REGISTRY[key](**kwargs)


def register_scheduler(scheduler_type: str, scheduler_class: Type[Scheduler]) -> None:
    SCHEDULER_REGISTRY[scheduler_type.lower()] = scheduler_class


def list_supported_schedulers() -> Dict[str, Type[Scheduler]]:
    return SCHEDULER_REGISTRY.copy()
