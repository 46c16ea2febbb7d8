"""InputStreamingScheduler — text arrives while audio is already being generated (drop-in for
/root/reference/vox_serve/scheduler/input_streaming.py:26-322).

Wire messages (beside the regular `{json}|audio` request): `id|TEXT_STREAM_START|{json}`, `id|TEXT_UPDATE|text`,
`id|TEXT_COMPLETE|`.  Text is buffered until MIN_INITIAL_TEXT_CHARS characters are there; the prefill then runs on the
FIRST text token only and the rest is queued token by token for the worker to inject into the text column of each
decode step (worker/base.py:362-394); a request whose queue ran dry before TEXT_COMPLETE sits out LM steps."""
import json
from typing import Optional

from ..requests import Request
from .base import Scheduler

MIN_INITIAL_TEXT_CHARS = 20


class InputStreamingScheduler(Scheduler):
    def _tokenizer(self):
        return self.model_worker.model.text_tokenizer

    def _find(self, request_id: str) -> Optional[Request]:
        return next((r for r in self.active_requests if r.request_id == request_id), None)

    def _prepare_prefill_with_minimal_text(self, req: Request) -> None:
        tok = self._tokenizer()
        ids = tok.encode(req.input_text_buffer, add_special_tokens=False)
        if not ids:                                   # nothing tokenizable: hand the raw text to preprocess
            req.prompt = req.input_text_buffer
        else:
            req.prompt = tok.decode([ids[0]], skip_special_tokens=True)
            for t in ids[1:]:
                req.pending_text_tokens.put(t)
            req.total_text_tokens = len(ids) - 1
        req.prefill_ready = True

    def _handle_request_payload(self, message_payload: bytes) -> Optional[Request]:
        parts = message_payload.split(b"|", 2)
        if len(parts) >= 2:
            try:
                kind = parts[1].decode("utf-8")
            except UnicodeDecodeError:
                kind = None
            body = parts[2] if len(parts) > 2 else b""
            if kind == "TEXT_STREAM_START":
                return self._handle_text_stream_start(parts[0], body or b"{}")
            if kind == "TEXT_UPDATE":
                self._handle_text_update(parts[0], body)
                return None
            if kind == "TEXT_COMPLETE":
                self._handle_text_complete(parts[0])
                return None
        return super()._handle_request_payload(message_payload)

    def _handle_text_stream_start(self, request_id_bytes: bytes, config_json: bytes) -> Request:
        try:
            cfg = json.loads(config_json.decode("utf-8"))
        except json.JSONDecodeError:
            cfg = {}
        streaming = cfg.get("is_streaming", True)
        return Request(request_id=request_id_bytes.decode("utf-8"), prompt="",
                       audio_path=cfg.get("audio_path") if self.model_worker.supports_audio_input else None,
                       is_streaming=streaming, is_pressing=streaming, is_input_streaming=True, input_text_buffer="",
                       text_complete=False, prefill_ready=False, waiting_for_text=False,
                       model_kwargs=cfg.get("model_kwargs", {}))

    def _handle_text_update(self, request_id_bytes: bytes, text_chunk: bytes) -> None:
        rid, text = request_id_bytes.decode("utf-8"), text_chunk.decode("utf-8")
        if not text:
            return
        req = self._find(rid)
        if req is None:
            self.logger.warning(f"TEXT_UPDATE for unknown request: {rid}")
            return
        if req.text_complete:
            self.logger.warning(f"TEXT_UPDATE after TEXT_COMPLETE for {rid}, ignoring")
            return
        if not req.done_lm_prefill:
            req.input_text_buffer += text
            if not req.prefill_ready and len(req.input_text_buffer) >= MIN_INITIAL_TEXT_CHARS:
                self._prepare_prefill_with_minimal_text(req)
        else:
            ids = self._tokenizer().encode(text, add_special_tokens=False)
            for t in ids:
                req.pending_text_tokens.put(t)
            req.total_text_tokens += len(ids)
            req.waiting_for_text = False

    def _handle_text_complete(self, request_id_bytes: bytes) -> None:
        rid = request_id_bytes.decode("utf-8")
        req = self._find(rid)
        if req is None:
            self.logger.warning(f"TEXT_COMPLETE for unknown request: {rid}")
            return
        req.text_complete = True
        if req.prefill_ready:
            req.waiting_for_text = False              # from now on the worker feeds tts_eos once, then tts_pad
        elif req.input_text_buffer:
            self._prepare_prefill_with_minimal_text(req)
        else:
            req.done_lm_generation = req.done_all = True
            req.finish_reason = "no_text_provided"

    def _lm_candidates(self):
        prefill, decode = [], []
        for req in self.active_requests:
            if req.done_lm_generation:
                continue
            if not req.done_lm_prefill:
                if req.is_input_streaming and not req.prefill_ready:
                    continue                          # still buffering the initial text
                prefill.append(req)
            else:
                if req.is_input_streaming and req.pending_text_tokens.empty() and not req.text_complete:
                    req.waiting_for_text = True       # pause: no text to inject yet
                    continue
                decode.append(req)
        return prefill, decode
