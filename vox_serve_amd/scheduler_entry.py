"""Scheduler daemon of the data-parallel serving pool: one process per GPU
(drop-in for /root/reference/vox_serve/scheduler_entry.py:1-105; started by launch.py's ServingPool).

The parent pins the GPU through the environment of the child (HIP_VISIBLE_DEVICES) BEFORE the interpreter starts; this
module therefore must not import torch (or anything that does) at module level — the HIP runtime reads the mask when it
is first loaded.  The daemon builds its ModelWorker + Scheduler on "cuda:0" of what it can see, announces itself on the
shared result transport (`__rank<r>__|READY|{...}`: the server waits for it instead of sleeping a fixed time) and then
runs the scheduler loop until it is terminated.
"""
import argparse
import importlib
import json
import os
import sys


def _build_worker(args, device):
    """`--worker-factory module:function` (tests, tiny configurations) or the registry's model by name."""
    common = dict(max_batch_size=args.max_batch_size, max_num_pages=args.max_num_pages, page_size=args.page_size,
                  top_p=args.top_p, top_k=args.top_k, min_p=args.min_p, temperature=args.temperature,
                  max_tokens=args.max_tokens, repetition_penalty=args.repetition_penalty,
                  repetition_window=args.repetition_window, cfg_scale=args.cfg_scale, greedy=args.greedy,
                  enable_nvtx=args.enable_nvtx, dp_rank=args.dp_rank, dp_size=args.dp_size,
                  detokenize_interval=args.detokenize_interval)
    if args.worker_factory:
        mod, _, fn = args.worker_factory.partition(":")
        return getattr(importlib.import_module(mod), fn)(device=device, **common)
    from vox_serve_amd.model import load_model
    from vox_serve_amd.worker import ModelWorker
    mk = {k: v for k, v in common.items() if k in ("top_p", "top_k", "min_p", "temperature", "max_tokens",
                                                    "repetition_penalty", "repetition_window", "cfg_scale", "greedy",
                                                    "detokenize_interval")}
    if common["max_num_pages"] is None:
        common["max_num_pages"] = 2048
    model = load_model(args.model_name, device=device, synthetic=args.synthetic, checkpoint_dir=args.checkpoint_dir,
                       max_batch_size=args.max_batch_size, max_num_pages=common["max_num_pages"],
                       page_size=args.page_size, **mk)
    return ModelWorker(model_name=args.model_name, model=model, device=device, **common)


def _run_scheduler_daemon(args) -> None:
    visible = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "not set"))
    print(f"[DP ENTRY] rank {args.dp_rank}/{args.dp_size}: torch already imported: {'torch' in sys.modules}, "
          f"visible devices: {visible}", file=sys.stderr, flush=True)      # (stderr: a parent's stdout may be a protocol, e.g. bench.py's one JSON line)
    import torch                    # first import in this process: sees the mask the parent set

    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    device = "cuda:0" if n_dev else "cpu"
    if device == "cpu" and not args.worker_factory:
        raise RuntimeError("scheduler daemon: no GPU visible (the HIP library is the only compute path)")
    import logging
    logging.basicConfig(level=getattr(logging, str(args.log_level).upper(), logging.INFO))
    from vox_serve_amd.scheduler import load_scheduler
    from vox_serve_amd.ipc import make_transport

    worker = _build_worker(args, device)
    transport = make_transport(args.request_socket_path, args.result_socket_path)
    kind = "disaggregation" if args.enable_disaggregation else args.scheduler_type
    scheduler = load_scheduler(kind, model_worker=worker, max_batch_size=args.max_batch_size, transport=transport,
                               async_scheduling=args.async_scheduling)
    scheduler.idle_sleep_s = 0.0005
    ready = {"dp_rank": args.dp_rank, "dp_size": args.dp_size, "pid": os.getpid(), "device": device,
             "visible_devices": visible, "torch_devices": n_dev}
    transport.send_result(f"__rank{args.dp_rank}__".encode() + b"|READY|" + json.dumps(ready).encode())
    print(f"[DP ENTRY] rank {args.dp_rank}: scheduler '{kind}' serving {args.model_name} on {device}", file=sys.stderr, flush=True)
    scheduler.run_forever()


def main(argv=None):
    p = argparse.ArgumentParser(description="vox-hip scheduler daemon (one per GPU)")
    p.add_argument("--dp-rank", type=int, required=True)
    p.add_argument("--dp-size", type=int, required=True)
    p.add_argument("--model-name", type=str, required=True)
    p.add_argument("--scheduler-type", type=str, default="base")
    p.add_argument("--max-batch-size", type=int, default=8)
    p.add_argument("--max-num-pages", type=int, default=None)
    p.add_argument("--page-size", type=int, default=128)
    p.add_argument("--request-socket-path", type=str, required=True)
    p.add_argument("--result-socket-path", type=str, required=True)
    p.add_argument("--log-level", type=str, default="INFO")
    p.add_argument("--top-p", type=float, default=None)
    p.add_argument("--top-k", type=int, default=None)
    p.add_argument("--min-p", type=float, default=None)
    p.add_argument("--temperature", type=float, default=None)
    p.add_argument("--max-tokens", type=int, default=None)
    p.add_argument("--repetition-penalty", type=float, default=None)
    p.add_argument("--repetition-window", type=int, default=None)
    p.add_argument("--cfg-scale", type=float, default=None)
    p.add_argument("--greedy", action="store_true")
    p.add_argument("--enable-cuda-graph", action="store_true")        # accepted for CLI parity: frames are always hipGraphs
    p.add_argument("--enable-disaggregation", action="store_true")
    p.add_argument("--enable-nvtx", action="store_true")
    p.add_argument("--enable-torch-compile", action="store_true")     # accepted for CLI parity: no tracing compiler here
    p.add_argument("--async-scheduling", action="store_true")
    p.add_argument("--detokenize-interval", type=int, default=None)
    # not in the reference: there are no downloads on an offline box
    p.add_argument("--synthetic", action="store_true", help="random-init weights of the named architecture")
    p.add_argument("--checkpoint-dir", type=str, default=None)
    p.add_argument("--worker-factory", type=str, default=None, help="module:function building the ModelWorker")
    _run_scheduler_daemon(p.parse_args(argv))


if __name__ == "__main__":
    main()
