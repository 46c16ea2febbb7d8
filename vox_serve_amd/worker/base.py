"""ModelWorker — the per-GPU executor the schedulers call (drop-in surface of
/root/reference/vox_serve/worker/base.py:14-790 and worker/cuda_graph_worker.py:12-1280).

Host bookkeeping follows the reference line by line where it is observable from outside:
  prepare_lm_inputs   worker/base.py:210-360   FIFO page allocation, page growth, position ids incl. quirk Q1
  free_kv_cache       worker/base.py:757-771
  run_detokenize      worker/base.py:616-693 / cuda_graph_worker.py:1162-1280   last-chunk padding by repeating the final
                      token, trim int(len*(n-0.5)/interval), PCM16 truncation, done_all rule
  run_lm_prefill / run_lm_decode   cuda_graph_worker.py:806-1160 — but one native call per step: the whole frame (talker,
                      sampling, 15 depth steps, feedback) is a single hipGraph replay; the python per-request loops of
                      Qwen3TTSModel.sampling / depth_sampling (qwen3_tts.py:1931-1962, 1995-2002) run once per frame on
                      one D2H copy of the sampled ids instead of >= 16*B `.item()` synchronisations.
"""
import contextlib
import logging
import os
import queue
import time
from typing import List, Optional

import numpy as np
import torch

from ..requests import LMInputs, Request
from ..sampling import native_config
from ..tokenizer.base import DecoderCache


def engine_has_status_row(engine) -> bool:
    """True for engines whose per-frame token snapshot carries a status row in front of out_ids (Qwen3Engine: persistent kernels).
    LMEngine / CSMEngine subclass Qwen3Engine for its graph / plan machinery WITHOUT running its __init__: they have the
    `snapshot_src` / `read_ids` methods but no `_out_block`, so the presence of a method says nothing — the buffer does."""
    return getattr(engine, "status_row", None) is not None and getattr(engine, "_out_block", None) is not None


class OutOfPages(queue.Empty):
    """The step needs more KV pages than the free list holds.  Raised by prepare_lm_inputs BEFORE any request state is
    touched, so the scheduler can defer the new prompt and retry (the reference lets `queue.Empty` escape from the
    middle of the allocation, worker/base.py:283-287, and the server dies)."""


class ModelWorker:
    def __init__(self, model_name: str = None, max_batch_size: int = 8, max_num_pages: int = 2048, page_size: int = 128,
                 top_p: float = None, top_k: int = None, min_p: float = None, temperature: float = None,
                 max_tokens: int = None, repetition_penalty: float = None, repetition_window: int = None,
                 cfg_scale: float = None, greedy: bool = False, enable_nvtx: bool = False,
                 enable_torch_compile: bool = False, detokenizer_device: Optional[str] = None, dp_rank: int = 0,
                 dp_size: int = 1, detokenize_interval: int = None, model=None, device: str = "cuda:0", seed: int = 0,
                 exact_rows: Optional[int] = None):
        if exact_rows is not None:
            # rows up to which the wave64 VALU kernels are used (include/voxhip.h: vox_ctx_set_exact_rows): must precede the engines' creation
            from .. import _native as _N
            _N.set_exact_rows(exact_rows)
        if model is None:
            from ..model import load_model
            model = load_model(model_name, device=device, top_p=top_p, top_k=top_k, min_p=min_p, temperature=temperature,
                               max_tokens=max_tokens, repetition_penalty=repetition_penalty,
                               repetition_window=repetition_window, cfg_scale=cfg_scale, greedy=greedy,
                               audio_decoder_device=detokenizer_device, detokenize_interval=detokenize_interval,
                               max_batch_size=max_batch_size, max_num_pages=max_num_pages, page_size=page_size)
        # One process per GPU whose host side only shuffles a few KB per step: intra-op CPU parallelism buys nothing, and on a
        # 256-core host every multi-threaded torch CPU op (a 50 KB torch.stack!) wakes the whole OpenMP pool, whose spinning
        # threads then starve the HIP runtime's own threads — measured: 35 ms per scheduler step instead of 4.
        if os.environ.get("VOX_KEEP_TORCH_THREADS") != "1" and torch.get_num_threads() > 1:
            torch.set_num_threads(1)
        self.model = model
        self.device = device
        self.detokenizer_device = detokenizer_device or getattr(model, "audio_decoder_device", None) or device
        self.max_batch_size, self.max_num_pages, self.page_size = max_batch_size, max_num_pages, page_size
        self.dp_rank, self.dp_size = dp_rank, dp_size
        base = logging.getLogger(__name__)
        self.logger = logging.LoggerAdapter(base, {}) if dp_size > 1 else base
        if dp_size > 1:
            self.logger.process = lambda msg, kw: (f"[DP {dp_rank}/{dp_size}] {msg}", kw)
        self.nvtx_enabled = enable_nvtx
        self.seed = seed
        self.empty_pages = queue.Queue()
        for i in range(max_num_pages):
            self.empty_pages.put(i)
        self.needs_watermarking = getattr(model, "needs_watermarking", False)
        self.has_depth_transformer = getattr(model, "has_depth_transformer", False)
        self.async_scheduling = False   # set by a scheduler running the reference's async loop (scheduler/base.py:166-221)
        self._pending = None         # the deferred request-state update of the last launched step (async scheduling)
        self._detok_fence_needed = True
        self._detok_stream = None    # launch_detokenize: the codec chunk + its D2H copy run here, beside the LM frame
        self._snap, self._snap_i = None, 0
        self._code_pins, self._code_pin_i = [], 0
        # host time of the two per-step bookkeeping halves (plan building; request-state update after the tokens arrived, the
        # wait for them excluded): bench.py reports them per step next to the serving-path throughput
        self.host_stats = {"prepare_s": 0.0, "after_s": 0.0, "steps": 0}
        self._resident = None        # request ids whose next inputs already sit in the engine's rows (feedback path)
        self._resident_reqs = []     # ... and the requests themselves (their repetition-cache rows live in the engine)
        self._next_feats = None
        # long prompts: the reference has one 1024-token prefill bucket and never schedules anything longer
        # (scheduler/base.py:283-286, cuda_graph_worker.py:61); here a single request longer than the engine's row
        # capacity is prefilled in context chunks (run_lm_prefill), so the scheduler may pick prompts up to the
        # engine's sequence capacity.
        e = getattr(model, "engine", None)
        cap = int(getattr(e, "max_seq_len", 0)) - 2
        if cap > 1024:
            self.cuda_graph_seq_len_buckets = [1024, cap]
        # the reference's prefill graphs hold 8 requests (cuda_graph_worker.py:62); a smaller engine holds fewer
        self.prefill_graph_batch_size = min(8, max_batch_size, int(getattr(e, "max_batch", max_batch_size)))

    # ---- properties read by schedulers (scheduler/base.py:127, 243-245) ----
    detokenize_interval = property(lambda self: self.model.detokenize_interval)
    detokenize_overlap = property(lambda self: self.model.detokenize_overlap)
    supports_audio_input = property(lambda self: self.model.supports_audio_input)

    @property
    def available_batch_sizes(self) -> Optional[List[int]]:
        return None

    defers_on_page_exhaustion = True                  # prepare_lm_inputs raises OutOfPages before mutating anything
    materialize_repetition_cache = False              # lm_inputs["repetition_cache"] as the reference builds it (tests only)
    prefill_graph_batch_size = 8                      # cuda_graph_worker.py:62
    cuda_graph_seq_len_buckets = [1024]               # cuda_graph_worker.py:61

    # -------------------------------------------------------------------------------------------------
    def prepare_lm_inputs(self, lm_requests: List[Request], detokenize_requests: List[Request]) -> Optional[LMInputs]:
        t0 = time.perf_counter()
        try:
            return self._prepare_lm_inputs(lm_requests, detokenize_requests)
        finally:
            self.host_stats["prepare_s"] += time.perf_counter() - t0
            self.host_stats["steps"] += 1

    def _prepare_lm_inputs(self, lm_requests: List[Request], detokenize_requests: List[Request]) -> Optional[LMInputs]:
        for req in detokenize_requests:
            req.audio_decode_idx = req.next_audio_decode_idx.copy()
        if len(lm_requests) == 0:
            return None
        qo_indptr, paged_kv_indptr, paged_kv_indices, paged_kv_last_page_len = [0], [0], [], []
        input_ids_list, position_ids_list, feats, masks, reps = [], [], [], [], []
        is_prefill = any(not req.done_lm_prefill for req in lm_requests)
        prefill_flags = [not req.done_lm_prefill for req in lm_requests]
        # An input-streaming decode row gets its text column rewritten on the host below (which invalidates residency), so the
        # step restages from req.input_tokens / input_features: the deferred update of the previous step must land first.
        if self._pending is not None and (is_prefill or [r.request_id for r in lm_requests] != self._resident
                                          or any(getattr(r, "is_input_streaming", False) for r in lm_requests)):
            self.drain()             # this step reads per-request host state (restaging): finish the deferred update first
        ps = self.page_size
        # admission: preprocess the new prompts (once) and count the pages this step takes before touching any KV state
        need = 0
        for req in lm_requests:
            if not req.done_lm_prefill:
                self._preprocess_request(req)
                need += (len(req.input_tokens) + ps - 1) // ps
            elif req.kv_last_page_len + 1 > ps:
                need += 1
        if need > self.empty_pages.qsize():
            raise OutOfPages(f"step needs {need} KV pages, {self.empty_pages.qsize()} free")
        for req in lm_requests:
            if not req.done_lm_prefill:
                n = len(req.input_tokens)
                input_ids_list.append(req.input_tokens)
                position_ids_list.extend(range(n))
                feats.append(req.input_features)
                masks.append(req.input_masks)
                reps.append(req.repetition_cache)
                req.kv_token_len = n
                req.kv_pages = [self.empty_pages.get_nowait() for _ in range((n + ps - 1) // ps)]
                req.kv_last_page_len = n % ps or ps
                qo_indptr.append(qo_indptr[-1] + n)
                req.next_position_id = n + 1          # quirk Q1 (worker/base.py:299)
                req.done_lm_prefill = True
            else:
                if req.is_input_streaming:
                    self._inject_streaming_text_token(req)
                input_ids_list.append(req.input_tokens)
                feats.append(req.input_features)
                masks.append(req.input_masks)
                reps.append(req.repetition_cache)
                req.kv_token_len += 1
                req.kv_last_page_len += 1
                if req.kv_last_page_len > ps:
                    req.kv_pages.append(self.empty_pages.get_nowait())
                    req.kv_last_page_len = 1
                qo_indptr.append(qo_indptr[-1] + 1)
                position_ids_list.append(req.next_position_id)
                req.next_position_id += 1
            paged_kv_indptr.append(paged_kv_indptr[-1] + len(req.kv_pages))
            paged_kv_indices.extend(req.kv_pages)
            paged_kv_last_page_len.append(req.kv_last_page_len)
        input_ids = torch.cat([t.to("cpu") for t in input_ids_list], dim=0)
        position_ids = torch.tensor(position_ids_list, dtype=torch.int32)
        input_masks = (torch.cat([m.to("cpu") for m in masks if m is not None], dim=0)
                       if self.model.needs_input_masks and masks else None)
        input_features = None
        # a decode step over the batch that is already resident in the engine's rows (the feedback path) restages nothing:
        # the per-step concatenation of the requests' feature rows (a device kernel + 2 B tiny ops) is skipped
        resident = (not is_prefill and self._resident is not None and [r.request_id for r in lm_requests] == self._resident
                    and not self.materialize_repetition_cache)
        if self.model.needs_input_features and feats and not resident:
            fl = [f for f in feats if f is not None]
            dev = next((f.device for f in fl if f.is_cuda), torch.device("cpu"))   # decode rows live on the GPU
            input_features = torch.cat([f.to(dev) for f in fl], dim=0)
        # The reference re-stacks every request's repetition cache into one [B,W,C,V] tensor each step (worker/base.py:344-347).
        # Here the caches live in the engine's rows while a batch is resident (_stage_repetition reads req.repetition_cache
        # directly), so the per-step host copy is only materialised on request (host-trace tests).
        repetition_cache = None
        if self.materialize_repetition_cache and self.model.use_repetition_penalty and reps and all(c is not None for c in reps):
            repetition_cache = torch.stack(reps, dim=0)
        return {"qo_indptr": qo_indptr, "paged_kv_indptr": paged_kv_indptr, "paged_kv_indices": paged_kv_indices,
                "paged_kv_last_page_len": paged_kv_last_page_len, "input_ids": input_ids, "position_ids": position_ids,
                "input_features": input_features, "input_masks": input_masks, "repetition_cache": repetition_cache,
                "is_prefill": is_prefill, "prefill_flags": prefill_flags}

    def _preprocess_request(self, req: Request) -> None:
        """model.preprocess -> request fields (worker/base.py:246-270), once per request."""
        if getattr(req, "_preprocessed", False):
            return
        if req.is_input_streaming and not self.model.supports_input_streaming:
            raise ValueError(f"Input streaming is not supported by model {self.model.model_name}. "
                             f"Only Qwen3-TTS models support input streaming mode.")
        kw = req.model_kwargs.copy()
        if req.is_input_streaming:
            kw["is_input_streaming"] = True
        pre = self.model.preprocess(prompt=req.prompt, audio_path=req.audio_path, **kw)
        req.input_tokens = pre.input_tokens
        if req.input_tokens is not None:
            req.input_length = req.input_tokens.shape[0]
        if pre.input_features is not None:
            req.input_features = pre.input_features
        if pre.input_masks is not None:
            req.input_masks = pre.input_masks
        if pre.repetition_cache is not None:
            req.repetition_cache = pre.repetition_cache
        if getattr(pre, "decoder_cache", None) is not None:
            req.decoder_cache = pre.decoder_cache
            self._detok_fence_needed = True      # its slots were reset on this (the LM) stream: the next chunk orders itself behind
        req._preprocessed = True

    def _inject_streaming_text_token(self, req: Request) -> None:
        """worker/base.py:362-394: next queued text token, then tts_eos once, then tts_pad."""
        tk = self.model.tokens
        req._last_injected = None            # what this step took from the request's text state (undo_decode_advance puts it back)
        try:
            tok = req.pending_text_tokens.get_nowait()
            req.input_tokens[0, -1] = tok
            req.text_token_cursor += 1
            req._last_injected = ("token", tok)
        except queue.Empty:
            if req.text_complete and not req.eos_injected:
                req.input_tokens[0, -1] = tk.tts_eos
                req.eos_injected = True
                req._last_injected = ("eos", None)
            else:
                req.input_tokens[0, -1] = tk.tts_pad
        self._resident = None        # the text column changed on the host: restage

    # -------------------------------------------------------------------------------------------------
    def _sampling(self):
        cfg = native_config(self.model.default_sampling_config)
        if getattr(self.model.engine, "rep_cache", None) is not None:       # persisted caches: the penalty is live
            cfg.repetition_penalty = float(self.model.default_sampling_config.repetition_penalty)
        return cfg

    def _token_plan(self, lm_inputs: LMInputs):
        """Per-row (request, visible kv length, page, slot) exactly as FlashInferPrefillWrapper.plan derives them
        (flashinfer_utils.py:86-124): rows are right-aligned to the end of their request's KV."""
        ps = self.page_size
        qo, ip, idx, last = (lm_inputs[k] for k in ("qo_indptr", "paged_kv_indptr", "paged_kv_indices", "paged_kv_last_page_len"))
        q_req, kvlen, page, slot = [], [], [], []
        for r in range(len(last)):
            m = qo[r + 1] - qo[r]
            n = (ip[r + 1] - ip[r] - 1) * ps + last[r]
            for j in range(m):
                t = n - m + j
                q_req.append(r); kvlen.append(t + 1); page.append(idx[ip[r] + t // ps]); slot.append(t % ps)
        return q_req, kvlen, page, slot

    def _stage_features(self, feats, dst):
        dst[: feats.shape[0]].copy_(feats.to(dst.device, dst.dtype))

    def run_lm_prefill(self, requests: List[Request], lm_inputs: LMInputs):
        if len(requests) == 0:
            return None
        e = self.model.engine
        n_rows, n_req = int(lm_inputs["input_ids"].shape[0]), len(requests)
        if n_rows > e.max_rows or n_req > e.max_batch:
            if n_req != 1:
                # a prompt (plus the decode rows piggy-backed onto its step) that does not fit the engine's row capacity:
                # the reference never schedules one (one 1024-token bucket, scheduler/base.py:283-286) and raises for any
                # overflow; here the step is split per request — each prefill alone (chunked when it is itself too long),
                # then the decode rows as a normal decode step.  Per-request results are those of the unsplit step.
                return self._run_split_prefill(requests, lm_inputs)
            return self._run_long_prefill(requests, lm_inputs)
        self.nvtx_range_push(f"lm_prefill_bs{n_req}")            # range names of cuda_graph_worker.py:813-1228
        q_req, kvlen, page, slot = self._token_plan(lm_inputs)
        e.row_ids[:n_rows].copy_(lm_inputs["input_ids"].to(torch.int32))
        if lm_inputs["input_masks"] is not None:     # Qwen3 consumes the text column's mask, CSM every column's
            mk = lm_inputs["input_masks"] if e.row_masks.dim() == 2 else lm_inputs["input_masks"][:, -1]
            e.row_masks[:n_rows].copy_(mk.to(torch.uint8))
        if lm_inputs["input_features"] is not None:
            self._stage_features(lm_inputs["input_features"], e.row_feats)
        self._stage_repetition(requests, e)
        qo = lm_inputs["qo_indptr"]
        e.upload_plan(pos=lm_inputs["position_ids"].numpy(), kvlen=kvlen, page=page, slot=slot, q_req=q_req,
                      last_rows=[q - 1 for q in qo[1:]], indptr=lm_inputs["paged_kv_indptr"],
                      indices=lm_inputs["paged_kv_indices"])
        self.nvtx_range_push("cuda_graph_replay")                # (here: the eager prefill launches)
        e.prefill(n_rows, n_req, max(kvlen), self._sampling(), seed=self.seed, feedback=True)
        self.nvtx_range_pop()
        self.nvtx_range_push("sampling")                         # request-state half of sampling (+ depth loop results)
        task = self._finish_launch(requests)
        self.nvtx_range_pop()
        self.nvtx_range_pop()
        return task

    @staticmethod
    def _slice_inputs(lm_inputs: LMInputs, lo: int, hi: int) -> LMInputs:
        """lm_inputs restricted to requests [lo, hi) of the step (rows qo_indptr[lo]..qo_indptr[hi])."""
        qo, ip = lm_inputs["qo_indptr"], lm_inputs["paged_kv_indptr"]
        r0, r1 = qo[lo], qo[hi]

        def rows(t):
            return None if t is None else t[r0:r1]
        rep = lm_inputs.get("repetition_cache")
        return {"qo_indptr": [q - r0 for q in qo[lo:hi + 1]], "paged_kv_indptr": [i - ip[lo] for i in ip[lo:hi + 1]],
                "paged_kv_indices": lm_inputs["paged_kv_indices"][ip[lo]:ip[hi]],
                "paged_kv_last_page_len": lm_inputs["paged_kv_last_page_len"][lo:hi],
                "input_ids": rows(lm_inputs["input_ids"]), "position_ids": rows(lm_inputs["position_ids"]),
                "input_features": rows(lm_inputs["input_features"]), "input_masks": rows(lm_inputs["input_masks"]),
                "repetition_cache": None if rep is None else rep[lo:hi], "is_prefill": True}

    def _run_split_prefill(self, requests: List[Request], lm_inputs: LMInputs):
        qo = lm_inputs["qo_indptr"]
        fresh = lm_inputs.get("prefill_flags") or [qo[i + 1] - qo[i] > 1 for i in range(len(requests))]
        decode_idx = [i for i, f in enumerate(fresh) if not f]
        for i, f in enumerate(fresh):
            if f:
                self.run_lm_prefill(requests[i:i + 1], self._slice_inputs(lm_inputs, i, i + 1))
        # the decode rows are contiguous behind the prefill in every scheduler's selection; fall back to one call per row
        if decode_idx:
            lo, hi = decode_idx[0], decode_idx[-1] + 1
            if decode_idx == list(range(lo, hi)):
                sub = self._slice_inputs(lm_inputs, lo, hi)
                sub["is_prefill"] = False
                self.run_lm_decode(requests[lo:hi], sub)
            else:
                for i in decode_idx:
                    sub = self._slice_inputs(lm_inputs, i, i + 1)
                    sub["is_prefill"] = False
                    self.run_lm_decode(requests[i:i + 1], sub)
        return None

    def _run_long_prefill(self, requests: List[Request], lm_inputs: LMInputs):
        """One request whose prompt exceeds the engine's row capacity: equal context chunks append K/V (no head, no
        sampling), the last chunk carries the prompt's final row and runs the normal prefill tail.  Per-row arithmetic is
        that of the unchunked prefill (rows only ever see K/V at earlier positions)."""
        e = self.model.engine
        n_rows = int(lm_inputs["input_ids"].shape[0])
        q_req, kvlen, page, slot = self._token_plan(lm_inputs)
        n_chunks = -(-n_rows // e.max_rows)
        size = -(-n_rows // n_chunks)
        pos = lm_inputs["position_ids"].numpy()
        ids = lm_inputs["input_ids"].to(torch.int32)
        self._stage_repetition(requests, e)
        for c in range(n_chunks):
            a, b = c * size, min(n_rows, (c + 1) * size)
            m, last = b - a, c == n_chunks - 1
            e.row_ids[:m].copy_(ids[a:b])
            if lm_inputs["input_masks"] is not None:
                mk = lm_inputs["input_masks"] if e.row_masks.dim() == 2 else lm_inputs["input_masks"][:, -1]
                e.row_masks[:m].copy_(mk[a:b].to(torch.uint8))
            if lm_inputs["input_features"] is not None:
                self._stage_features(lm_inputs["input_features"][a:b], e.row_feats)
            e.upload_plan(pos=pos[a:b], kvlen=kvlen[a:b], page=page[a:b], slot=slot[a:b], q_req=[0] * m, last_rows=[m - 1],
                          indptr=lm_inputs["paged_kv_indptr"], indices=lm_inputs["paged_kv_indices"])
            e.prefill(m, 1 if last else 0, max(kvlen[a:b]), self._sampling(), seed=self.seed, feedback=True)
        return self._finish_launch(requests)

    def run_lm_decode(self, requests: List[Request], lm_inputs: LMInputs):
        if len(requests) == 0:
            return None
        e = self.model.engine
        B = len(requests)
        self.nvtx_range_push(f"lm_decode_bs{B}")
        ids = [r.request_id for r in requests]
        if self._resident != ids:        # batch composition changed: restage the per-request inputs
            if hasattr(e, "note_restage"):
                e.note_restage()
            e.input_ids[:B].copy_(lm_inputs["input_ids"].to(torch.int32))
            if lm_inputs["input_masks"] is not None:
                mk = lm_inputs["input_masks"] if e.input_masks.dim() == 2 else lm_inputs["input_masks"][:, -1]
                e.input_masks[:B].copy_(mk.to(torch.uint8))
            if lm_inputs["input_features"] is not None:
                self._stage_features(lm_inputs["input_features"], e.input_features)
            self._stage_repetition(requests, e)
        e.upload_plan(pos=lm_inputs["position_ids"].numpy(), kvlen=[r.kv_token_len for r in requests],
                      page=[r.kv_pages[-1] for r in requests], slot=[r.kv_last_page_len - 1 for r in requests],
                      indptr=lm_inputs["paged_kv_indptr"], indices=lm_inputs["paged_kv_indices"])
        self.nvtx_range_push("cuda_graph_replay")                # one hipGraph: talker + sampling + the whole depth loop
        e.frame(B, max(r.kv_token_len for r in requests), self._sampling(), seed=self.seed, feedback=True)
        self.nvtx_range_pop()
        self.nvtx_range_push("sampling")
        task = self._finish_launch(requests)
        self.nvtx_range_pop()
        self.nvtx_range_pop()
        return task

    # ---- async scheduling (scheduler/base.py:163-221 of the reference: `sampling` hands back a coroutine that updates the
    # request objects; the scheduler awaits it while the next step already runs) ----
    def _finish_launch(self, requests: List[Request]):
        """After a frame / prefill was enqueued: update the requests now (synchronous scheduling, returns None) or hand back
        a coroutine that does it later.  The deferred form snapshots the sampled ids (pinned host buffer) and the next-step
        features in STREAM ORDER right behind the frame, so the following frame may overwrite the engine's buffers while the
        host is still reading this one's."""
        if not self.async_scheduling:
            self._after_frame(requests)
            return None
        e, B = self.model.engine, len(requests)
        self.drain()                                   # at most one step in flight behind the one being launched
        # engines with a status row (Qwen3-TTS: persistent kernels) ship it with the ids: row 0 of the snapshot is the frame's status
        # (gate on the DATA: LMEngine / CSMEngine inherit the snapshot_src method from Qwen3Engine but own no status row)
        with_status = engine_has_status_row(e)
        if self._snap is None:
            mk = lambda: {"ids": torch.zeros(e.out_ids.shape[0] + 1, *e.out_ids.shape[1:], dtype=e.out_ids.dtype).pin_memory(),
                          "feats": torch.zeros_like(e.next_features) if getattr(e, "next_features", None) is not None else None,
                          "event": torch.cuda.Event()}
            self._snap = [mk(), mk()]
        snap = self._snap[self._snap_i]
        self._snap_i ^= 1

        def take_snapshot():
            with e._OnStream(e):
                if with_status:
                    snap["ids"][:B + 1].copy_(e.snapshot_src(B), non_blocking=True)
                else:
                    snap["ids"][0].zero_()
                    snap["ids"][1:B + 1].copy_(e.out_ids[:B], non_blocking=True)
                if snap["feats"] is not None:
                    snap["feats"][:B].copy_(e.next_features[:B])
                snap["event"].record()
        take_snapshot()
        snap_seq = getattr(e, "launch_seq", 0)
        self._resident = [r.request_id for r in requests]
        self._resident_reqs = list(requests)
        positions = [r.next_position_id for r in requests]      # as of this step (the next step's prepare advances them)
        state = {"done": False}

        def finish():
            if state["done"]:
                return
            state["done"] = True
            snap["event"].synchronize()
            if with_status and int(snap["ids"][0].view(-1)[0]) != 0:
                # a hand-off of the persistent kernels timed out in THIS step: replay it on the launch chain (and the step already
                # enqueued behind it, if any), re-reading this step's outputs in between — the stream the clients get is unchanged
                e.recover(back=e.launch_seq - snap_seq + 1, code=int(snap["ids"][0].view(-1)[0]), on_first_done=take_snapshot)
                snap["event"].synchronize()
                if int(snap["ids"][0].view(-1)[0]) != 0:
                    raise RuntimeError("persistent kernels: status still set after recovery")
            self._after_frame(requests, out=snap["ids"][1:B + 1].to(torch.long), feats=snap["feats"], positions=positions)
            if self._pending is finish:
                self._pending = None
        self._pending = finish

        async def task():
            finish()
        return task()

    def drain(self):
        """Run the deferred request-state update of the last launched step now (idempotent)."""
        if self._pending is not None:
            self._pending()

    def _stage_repetition(self, requests: List[Request], e):
        """Per-request repetition caches live in the engine's rows while a batch is resident; when the batch changes the
        old rows go back to their requests and the new ones are copied in (the reference re-stacks them every step:
        worker/base.py:344-347, and stores the rows back after sampling: glm_voice.py:585-588)."""
        rc = getattr(e, "rep_cache", None)
        if rc is None:
            return
        for i, req in enumerate(self._resident_reqs):
            if req.repetition_cache is not None and not req.done_all:
                req.repetition_cache = rc[i].to(torch.bool)
        for i, req in enumerate(requests):
            if req.repetition_cache is not None:
                rc[i].copy_(req.repetition_cache.to(rc.device, torch.uint8))
        self._resident_reqs = []

    def _after_frame(self, requests: List[Request], out=None, feats=None, positions=None):
        """The request-state half of the plugin's `sampling` (+ `depth_sampling`), once per frame on one D2H copy.
        out / feats: stream-ordered snapshots of the step (async scheduling); default: read the engine's buffers now."""
        e, m = self.model.engine, self.model
        B = len(requests)
        if out is None:
            # the one synchronisation of the step (engines with a status row: it travels with the ids, and a hand-off timeout of the
            # persistent kernels is recovered from inside read_ids before anything is handed to the requests)
            out = e.read_ids(B) if hasattr(e, "read_ids") else e.out_ids[:B].cpu().to(torch.long)
            self._resident = [r.request_id for r in requests]
            self._resident_reqs = list(requests)
        t0 = time.perf_counter()
        try:
            self._update_requests(requests, out, feats, positions)
        finally:
            self.host_stats["after_s"] += time.perf_counter() - t0

    def _update_requests(self, requests, out, feats, positions):
        e, m = self.model.engine, self.model
        B = len(requests)
        # async scheduling launches step N+1 before step N's tokens are seen: a request that finished at N has one surplus
        # row in N+1, whose output is dropped here (the reference's async loop has the same one-step lag)
        if positions is None:
            positions = [r.next_position_id for r in requests]
        live = [i for i, r in enumerate(requests) if not r.done_lm_generation]
        if len(live) != B:
            requests, positions = [requests[i] for i in live], [positions[i] for i in live]
            out = out[live]
            if feats is not None:
                feats = feats[live]
            B = len(requests)
            if B == 0:
                return
        if hasattr(m, "update_requests"):                              # single-stack families
            saved = [r.next_position_id for r in requests]             # (the plugin's max_tokens rule reads next_position_id)
            for r, p_ in zip(requests, positions):
                r.next_position_id = p_
            try:
                m.update_requests(requests, out.view(B, -1))
            finally:
                for r, p_ in zip(requests, saved):
                    r.next_position_id = p_
            return
        # Qwen3-TTS: qwen3_tts.py:1931-1962, 1995-2002.  The per-request rows are views of three tensors built once per step (the
        # reference builds them request by request: at 32 requests that is ~250 tiny torch calls, 0.3 ms of host time per step)
        feats = (e.next_features if feats is None else feats)[:B].clone()
        C = m.n_codebooks
        pad = m.config.tts_pad_id
        rows = out[:B].clone()
        nxt = torch.zeros(B, C, dtype=torch.long)
        nxt[:, 0] = rows[:, 0]
        nxt[:, -1] = pad
        masks = torch.ones(B, C, dtype=torch.bool)
        c0 = rows[:, 0].tolist()
        for i, req in enumerate(requests):
            row = rows[i:i + 1]
            req.input_tokens = nxt[i:i + 1]
            if getattr(req, "is_input_streaming", False):
                req.input_tokens[0, -1] = 0            # the worker injects the next text token (worker/base.py:362-394)
            req.input_masks = masks[i:i + 1]
            req.input_features = feats[i:i + 1]
            req.lm_output_tokens.append(row)
            if not m.is_stop_id(c0[i]):
                req.lm_output_audio_tokens.append(row)
            else:
                req.done_lm_generation = True
                req.finish_reason = "stop_id_encountered"
        for req, p_ in zip(requests, positions):
            if p_ > m.max_tokens:
                req.done_lm_generation = True
                req.finish_reason = "max_tokens_reached"

    # -------------------------------------------------------------------------------------------------
    def run_detokenize(self, requests: List[Request]):
        """worker/base.py:616-690 of the reference: decode the selected windows, PCM16 into each request's output queue."""
        self.finish_detokenize(self.launch_detokenize(requests))

    def launch_detokenize(self, requests: List[Request]):
        """First half of `run_detokenize`: stage the windows and ENQUEUE the codec chunk(s) plus the D2H copy of their audio on
        the worker's detokenize stream, without waiting.  The scheduler launches the LM frame next, so the chunk and the frame
        share the GPU (the engine and the codec run on their own streams), and collects the audio with `finish_detokenize`.
        Same calls on the same data as the one-shot form; only the point where the host waits moves."""
        if len(requests) == 0:
            return None
        interval = self.detokenize_interval
        token_ids, mapping = [], []
        self.nvtx_range_push(f"detokenize_bs{len(requests)}")
        for ri, req in enumerate(requests):
            for ci in range(len(req.audio_decode_idx)):
                d = req.audio_decode_idx[ci]
                new = list(req.lm_output_audio_tokens[d: d + interval])
                if not new:
                    continue
                if len(new) < interval:
                    new.extend([new[-1]] * (interval - len(new)))      # pad by repeating the final token
                token_ids.append(torch.cat(new, dim=0))
                mapping.append((ri, ci))
        parts, event, n_last = [None] * len(mapping), None, []
        if token_ids:
            caches = [requests[ri].decoder_cache for ri, _ in mapping]
            stateful = all(c is not None for c in caches)
            # A request's streaming codec state lives in ONE native slot that decode_chunk advances in place, so a slot may
            # appear at most once per call: several windows of one request (offline / online policies hand them out
            # together) are decoded in consecutive rounds, window k of every request in round k — each window continues
            # from the state its predecessor left (seamless audio).  The reference decodes all of a request's windows from
            # the same starting state and keeps the last one's (worker/base.py:641-656).
            n_rounds = max(ci for _, ci in mapping) + 1 if stateful else 1
            # the chunk runs on a stream of the detokenizer's device (the LM's GPU, or a second one: worker/base.py:55-78, 641-644 of the
            # reference); entering the stream context makes that device current, so the plugin's native calls use its own context
            on_gpu = torch.cuda.is_available() and str(self.detokenizer_device).startswith("cuda")
            if on_gpu and self._detok_stream is None:
                self._detok_stream = torch.cuda.Stream(device=self.detokenizer_device)
            # The chunk's inputs come from the host; the only device state it shares with the LM stream is a decoder cache
            # that `preprocess` (re)initialised there.  Fence only then: a `wait_stream` leaves a barrier packet pending on the
            # detokenize queue until the LM frame in flight ends, and a pending barrier on a second hardware queue slows
            # every dispatch of that frame (N.graph_capture).
            if on_gpu and (self._detok_fence_needed or any(getattr(t, "is_cuda", False) for t in token_ids)):
                # the slot reset / cache initialisation ran on the CURRENT stream of the detokenizer's device (tokenizer/base.py
                # device_bound) — with the detokenizer on a second GPU that is not the LM's current stream; device token windows
                # were produced on their own device's current stream
                ddev = torch.device(self.detokenizer_device)
                waits = {torch.cuda.current_stream(ddev)}
                for t_ in token_ids:
                    if getattr(t_, "is_cuda", False):
                        waits.add(torch.cuda.current_stream(t_.device))
                for st_ in waits:
                    self._detok_stream.wait_stream(st_)
                self._detok_fence_needed = False
            ctx = torch.cuda.stream(self._detok_stream) if on_gpu else contextlib.nullcontext()
            with ctx:
                for rnd in range(n_rounds):
                    sel = [i for i, (_, ci) in enumerate(mapping) if not stateful or ci == rnd]
                    if not sel:
                        continue
                    batch = torch.stack([token_ids[i] for i in sel], dim=0)
                    if on_gpu and not batch.is_cuda:
                        batch = self._pinned_upload(batch)
                    cache = DecoderCache.cat([caches[i] for i in sel]) if stateful else None
                    self.nvtx_range_push("detokenize_replay")
                    audio = self.model.postprocess(batch, decoder_cache=cache)
                    self.nvtx_range_pop()
                    if cache is not None:        # tensor-valued state (CosyVoice2's speech tail) goes back to each request's own
                        for j, i in enumerate(sel):                      # cache (cuda_graph_worker.py:1241)
                            caches[i].copy_from(cache[j:j + 1])
                    if self.needs_watermarking:
                        audio = self.run_watermark(audio)
                    audio = audio.detach().float()
                    if audio.is_cuda:
                        host = torch.empty(audio.shape, dtype=torch.float32, pin_memory=True)
                        host.copy_(audio, non_blocking=True)       # stream-ordered right behind the chunk
                    else:
                        host = audio
                    for j, i in enumerate(sel):
                        parts[i] = (host, j)
                if on_gpu:
                    event = torch.cuda.Event()
                    event.record(self._detok_stream)
            for ri, ci in mapping:
                req = requests[ri]
                d = req.audio_decode_idx[ci]
                n_last.append(len(req.lm_output_audio_tokens[d: d + interval]))
        self.nvtx_range_pop()
        # the done_all rule (worker/base.py:674-678) reads done_lm_generation / the token count as they are when run_detokenize is
        # CALLED, i.e. before the LM step of the same scheduler iteration; with the overlap the second half runs after that
        # step, so both are snapshotted here (an EOS sampled by this step must not end the request before its tail window)
        done_snap = [(bool(r.done_lm_generation), len(r.lm_output_audio_tokens)) for r in requests]
        return {"requests": requests, "mapping": mapping, "parts": parts, "event": event, "n_last": n_last, "done_snap": done_snap}

    def _pinned_upload(self, batch: torch.Tensor) -> torch.Tensor:
        """Token windows -> the detokenizer's device through a pinned staging block, non-blocking on the current (detokenize)
        stream.  A pageable H2D copy makes the host wait for the LM frame in flight (measured with async scheduling at 32
        requests: 2.6 ms per chunk); the block is reused only after the copy that read it has run."""
        ring = self._code_pins
        if not ring:
            ring.extend({"buf": None, "event": None} for _ in range(4))
        ent = ring[self._code_pin_i]
        self._code_pin_i = (self._code_pin_i + 1) % len(ring)
        n = batch.numel()
        if ent["buf"] is None or ent["buf"].numel() < n or ent["buf"].dtype != batch.dtype:
            ent["buf"] = torch.empty(max(n, 4096), dtype=batch.dtype).pin_memory()
        elif ent["event"] is not None:
            ent["event"].synchronize()
        host = ent["buf"][:n].view(batch.shape)
        host.copy_(batch)
        dev = host.to(self.detokenizer_device, non_blocking=True)
        ent["event"] = ent["event"] or torch.cuda.Event()
        ent["event"].record()
        return dev

    def finish_detokenize(self, pending):
        """Second half of `run_detokenize`: wait for the audio of `launch_detokenize`, PCM16 -> the requests' output queues."""
        if pending is None:
            return
        requests, interval = pending["requests"], self.detokenize_interval
        if pending["event"] is not None:
            pending["event"].synchronize()
        for i, (ri, ci) in enumerate(pending["mapping"]):
            req = requests[ri]
            host, j = pending["parts"][i]
            a16 = (host[j].numpy() * 32767).astype(np.int16)
            n_last = pending["n_last"][i]
            if n_last < interval:
                a16 = a16[:, : int(a16.shape[1] * (n_last - 0.5) / interval)]
            req.output_audio.put(a16.tobytes())
        for req, (was_done, n_tokens) in zip(requests, pending["done_snap"]):
            if was_done and req.audio_decode_idx and req.audio_decode_idx[-1] + interval >= n_tokens:
                req.done_all = True

    def run_watermark(self, audio):
        return audio          # hook kept (worker/base.py:104-121); Qwen3 needs none

    def nvtx_range_push(self, name: str):
        """worker/base.py:736-747 — torch.cuda.nvtx is roctx on ROCm builds of torch (rocprofv3 --marker-trace shows the ranges)."""
        if self.nvtx_enabled:
            torch.cuda.synchronize()
            torch.cuda.nvtx.range_push(name)

    def nvtx_range_pop(self):
        if self.nvtx_enabled:
            torch.cuda.synchronize()
            torch.cuda.nvtx.range_pop()

    def undo_decode_advance(self, requests: List[Request]):
        """A launch failed after prepare_lm_inputs had advanced these decode rows (no K/V written, no token produced): put
        their KV length / position / page bookkeeping back, so that the next step does not attend to an unwritten slot."""
        ps = self.page_size
        # a step still in flight behind the failed one (async scheduling) produced real tokens: its request-state update must
        # not be lost with the failed launch
        try:
            self.drain()
        except Exception as ex:              # never mask the launch failure being handled
            self.logger.error(f"deferred request update failed during roll-back: {ex!r}")
            self._pending = None
        for req in requests:
            if not req.done_lm_prefill or req.done_all or not getattr(req, "kv_pages", None):
                continue
            req.kv_token_len -= 1
            req.next_position_id -= 1
            req.kv_last_page_len -= 1
            if req.kv_last_page_len == 0:
                self.empty_pages.put(req.kv_pages.pop())
                req.kv_last_page_len = ps
            inj = getattr(req, "_last_injected", None)      # input streaming: the text token / EOS this step consumed goes back
            if inj is not None:
                if inj[0] == "token":
                    with req.pending_text_tokens.mutex:
                        req.pending_text_tokens.queue.appendleft(inj[1])
                    req.text_token_cursor -= 1
                else:
                    req.eos_injected = False
                req._last_injected = None
        self._resident = None                # the engine's rows are in an unknown state: restage next step

    def free_kv_cache(self, request: Request):
        if getattr(request, "kv_pages", None):
            for p in request.kv_pages:
                self.empty_pages.put(p)
            request.kv_pages = []
            request.kv_token_len = 0
            request.kv_last_page_len = 0
        dc = getattr(request, "decoder_cache", None)
        if dc is not None and hasattr(self.model, "audio_decoder") and hasattr(self.model.audio_decoder, "release_cache"):
            self.model.audio_decoder.release_cache(dc)
            request.decoder_cache = None


# The reference's default worker is the graph-capturing subclass; here graph capture lives in the engine.
CudaGraphWorker = ModelWorker
HipGraphWorker = ModelWorker
