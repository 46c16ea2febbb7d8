"""OfflineScheduler — throughput mode (drop-in for /root/reference/vox_serve/scheduler/offline.py:4-136): LM steps run as
long as ANY request is still generating; the detokenizer only runs once every active request has finished its LM part,
and then takes as many windows per call as the batch allows (all windows of a request before moving to the next)."""
from ._chunks import ChunkCursor
from .base import Scheduler


class OfflineScheduler(Scheduler):
    prefill_len_default = 200      # offline.py:44: unknown prompt lengths are budgeted at 200 tokens

    def _select_detokenize_requests(self):
        if any(not r.done_lm_generation for r in self.active_requests):
            return []
        w = self.model_worker
        picked, used = [], 0
        for req in self.active_requests:
            if used >= self.max_batch_size:
                break
            cur = ChunkCursor(req, w.detokenize_interval, w.detokenize_overlap)
            starts = cur.take(self.max_batch_size - used)
            if starts:
                req.next_audio_decode_idx = starts
                used += len(starts)
                picked.append(req)
            elif req.done_lm_generation:          # drained after the last call: only the completion message is left
                req.done_all = True
                picked.append(req)
        return picked
