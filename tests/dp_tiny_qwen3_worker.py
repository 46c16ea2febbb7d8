"""Worker factory for the GPU serving-pool test (scheduler_entry --worker-factory tests.dp_tiny_qwen3_worker:make): the tiny
synthetic Qwen3-TTS + codec of tests/test_gpu_worker.py on the daemon's GPU (native engine + native codec)."""
import torch


def make(device="cuda:0", max_batch_size=4, max_num_pages=None, page_size=16, dp_rank=0, dp_size=1, **kw):
    from tests.test_gpu_worker import build
    from vox_serve_amd.worker import ModelWorker
    m, _ = build(torch.device(device), max_tokens=30)
    return ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=16, device=m.device, dp_rank=dp_rank, dp_size=dp_size)
