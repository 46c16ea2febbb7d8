"""Worker factory for the serving-pool tests (scheduler_entry --worker-factory tests.dp_fake_worker:make): the real
ModelWorker host logic over a fake plugin and a fake LM whose tokens depend on the prompt alone, on CPU — a request's PCM
is then a pure function of the request, whichever daemon serves it."""
import torch

from vox_serve_amd.model.base import PreprocessOutput
from vox_serve_amd.worker import ModelWorker


class FakePlugin:
    model_name = "fake"
    supports_input_streaming = False
    needs_input_masks = True
    needs_input_features = True
    use_repetition_penalty = False
    supports_audio_input = False
    needs_watermarking = False
    has_depth_transformer = False
    detokenize_interval = 4
    detokenize_overlap = 0
    n_codebooks = 3

    def preprocess(self, prompt=None, audio_path=None, **kw):
        n = 2 + len(prompt) % 5
        return PreprocessOutput(input_tokens=torch.arange(n * 3, dtype=torch.long).view(n, 3),
                                input_masks=torch.ones(n, 3, dtype=torch.bool), input_features=torch.zeros(n, 8))

    def postprocess(self, token_ids, **kw):
        base = (token_ids[:, :, 0].float() % 97) / 100.0 - 0.4
        return base.repeat_interleave(5, dim=1)[:, None, :]


class FakeLMWorker(ModelWorker):
    n_frames = 9

    def run_lm_prefill(self, reqs, li):
        self._fake(reqs)

    def run_lm_decode(self, reqs, li):
        self._fake(reqs)

    def _fake(self, reqs):
        for r in reqs:
            k = len(r.lm_output_tokens)
            h = sum(r.prompt.encode()) * 31 + 7 * k
            row = torch.tensor([[h % 1009, k, 7]], dtype=torch.long)
            r.input_tokens, r.input_masks, r.input_features = row, torch.ones(1, 3, dtype=torch.bool), torch.zeros(1, 8)
            r.lm_output_tokens.append(row)
            if k + 1 >= self.n_frames + len(r.prompt) % 4:
                r.done_lm_generation, r.finish_reason = True, "stop_id_encountered"
            else:
                r.lm_output_audio_tokens.append(row)


def make(device="cpu", max_batch_size=8, max_num_pages=None, page_size=4, dp_rank=0, dp_size=1, **kw):
    return FakeLMWorker(model=FakePlugin(), max_batch_size=max_batch_size, max_num_pages=max_num_pages or 64,
                        page_size=page_size, device="cpu", dp_rank=dp_rank, dp_size=dp_size)
