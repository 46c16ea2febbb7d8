"""OnlineScheduler — deadline-aware streaming (drop-in for /root/reference/vox_serve/scheduler/online.py:9-295).

A streaming request is *pressing* when the client is about to run out of audio: no chunk sent yet, or the playback of
everything sent so far ends within one second (`_update_pressing_status`, :266-295).  LM batches take the new prefill
first, then pressing decodes, then fill up with the rest (:16-96).  The detokenizer only runs when a pressing request
has a window ready; pressing requests share the batch in proportion to their backlog, leftovers go to the others
(:98-243)."""
import time
from typing import List

from ..requests import Request
from ._chunks import ChunkCursor
from .base import Scheduler


class OnlineScheduler(Scheduler):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.detokenize_max_batch_size = self.max_batch_size
        self.pressing_margin_s = 1.0

    # ---- LM batch: prefill, then pressing decodes, then the others ----
    def _order_decodes(self, decodes: List[Request]) -> List[Request]:
        return [r for r in decodes if r.is_pressing] + [r for r in decodes if not r.is_pressing]

    def _lm_batch_cap(self, prefill_cycle: bool, max_prefill_batch_size: int) -> int:
        # online.py:76-95: a prefill cycle is capped by the prefill batch size alone, not by max_batch_size
        return max_prefill_batch_size if prefill_cycle else self.max_batch_size

    # ---- detokenizer batch ----
    def _select_detokenize_requests(self) -> List[Request]:
        w = self.model_worker
        interval, overlap = w.detokenize_interval, w.detokenize_overlap
        cands = [r for r in self.active_requests if ChunkCursor(r, interval, overlap).ready()]
        if not cands:
            return []
        pressing = [r for r in cands if r.is_pressing]
        relaxed = [r for r in cands if not r.is_pressing]
        if not pressing:                                   # nothing urgent: only flush completions
            return [r for r in cands if r.done_all]
        cap = self.detokenize_max_batch_size
        backlog = [ChunkCursor(r, interval, overlap).remaining_windows() for r in pressing]
        total = sum(backlog)
        if total <= cap:
            share = backlog
        else:                                              # proportional split, at least one window each
            share = [max(1, n * cap // total) for n in backlog]
            while sum(share) > cap:
                for i in range(len(share)):
                    if share[i] > 1:
                        share[i] -= 1
                        if sum(share) <= cap:
                            break
        picked, used = [], 0
        for req, quota in zip(pressing, share):
            if quota <= 0:
                continue
            starts = ChunkCursor(req, interval, overlap).take(quota)
            if starts:
                req.next_audio_decode_idx = starts
                used += len(starts)
                picked.append(req)
            elif req.done_all:
                picked.append(req)
        left = cap - used
        for req in relaxed:
            if left <= 0:
                break
            starts = ChunkCursor(req, interval, overlap).take(left)
            if starts:
                req.next_audio_decode_idx = starts
                left -= len(starts)
                picked.append(req)
            elif req.done_all:
                picked.append(req)
        return picked

    # ---- pressing status ----
    def _prepare_requests(self):
        super()._prepare_requests()
        self._update_pressing_status()

    def _update_pressing_status(self, now: float = None):
        now = time.time() if now is None else now
        for req in self.active_requests:
            if not req.is_streaming:
                req.is_pressing = False
            elif not req.chunk_send_timestamps:
                req.is_pressing = True
            else:   # the client started playing when the first chunk arrived; the newest chunk starts after the others
                newest_starts = req.chunk_send_timestamps[0] + sum(req.chunk_durations) - req.chunk_durations[-1]
                req.is_pressing = now >= newest_starts - self.pressing_margin_s
