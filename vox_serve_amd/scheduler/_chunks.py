"""Chunk arithmetic shared by the scheduling policies: which audio-token windows of a request can be detokenized now.

A request's audio tokens are consumed in windows of `interval` tokens that advance by `step = interval - overlap`;
`next_audio_decode_idx` holds the window starts handed to the last detokenizer call.  Once generation is over the
final, shorter window is decodable too.  (scheduler/base.py:302-333 and the same rule inlined in online.py:104-120,
offline.py:95-124, disaggregation.py:178-200.)"""
from typing import List


class ChunkCursor:
    def __init__(self, req, interval: int, overlap: int):
        self.req, self.interval, self.step = req, interval, interval - overlap
        self.next = req.next_audio_decode_idx[-1] + self.step if req.next_audio_decode_idx else 0
        self.n_tokens = len(req.lm_output_audio_tokens)

    @property
    def has_full_window(self) -> bool:
        return self.next + self.interval <= self.n_tokens

    @property
    def has_tail(self) -> bool:
        """generation finished and some tokens are still undecoded"""
        return self.req.done_lm_generation and self.next < self.n_tokens

    @property
    def exhausted(self) -> bool:
        return self.req.done_lm_generation and self.next >= self.n_tokens

    def ready(self) -> bool:
        """the one-window rule of the base / online candidate scan; marks fully drained requests done_all"""
        if self.req.done_lm_generation:
            if self.next >= self.n_tokens:
                self.req.done_all = True
            return True
        return self.has_full_window

    def take(self, budget: int) -> List[int]:
        """window starts for up to `budget` windows: every full window, then the tail if generation is over"""
        out, nxt = [], self.next
        while budget > 0 and nxt + self.interval <= self.n_tokens:
            out.append(nxt)
            nxt += self.step
            budget -= 1
        if self.req.done_lm_generation and budget > 0 and nxt < self.n_tokens:
            out.append(nxt)
        return out

    def remaining_windows(self) -> int:
        """online.py:146-152: floor(remaining / step) (+1 for the tail of a finished request)"""
        rem = self.n_tokens - self.next
        n = max(0, rem // self.step)
        if self.req.done_lm_generation and rem > 0:
            n += 1
        return n
