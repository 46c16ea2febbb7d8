from .base import CudaGraphWorker, HipGraphWorker, ModelWorker  # noqa: F401
