"""CPU tests of the host side: worker bookkeeping + scheduler policy against traces captured from the reference's own
ModelWorker / Scheduler (tests/golden/g6_host_traces.json), the wire format, the C-ABI export table, the DP pool over
gloo (world_size 2), DecoderCache semantics and the Qwen3 prompt layout."""
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
        from vox_serve_amd.model.base import PreprocessOutput
        n = int(prompt)
        return PreprocessOutput(input_tokens=torch.arange(n * 3, dtype=torch.long).view(n, 3),
                                input_masks=torch.ones(n, 3, dtype=torch.bool), input_features=torch.zeros(n, 8))

    def postprocess(self, token_ids, **kw):
        base = (token_ids[:, :, 0].float() % 97) / 100.0 - 0.4
        return base.repeat_interleave(5, dim=1)[:, None, :]


def test_worker_and_scheduler_replay_reference_trace():
    from vox_serve_amd.requests import Request
    from vox_serve_amd.scheduler import Scheduler
    from vox_serve_amd.worker import ModelWorker
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "g6_host_traces.json")))
    w = ModelWorker(model=FakePlugin(), max_batch_size=8, max_num_pages=12, page_size=4, device="cpu")
    s = Scheduler(w, max_batch_size=8)
    arrivals = {0: [("A", 6)], 1: [("B", 3)], 4: [("C", 9)]}
    finish_after = {"A": 9, "B": 5, "C": 6}
    for rec in g["trace"]:
        step = rec["step"]
        for rid, n in arrivals.get(step, []):
            s.active_requests.append(Request(request_id=rid, prompt=str(n)))
        s.active_requests = [r for r in s.active_requests if not r.done_all]
        detok = s._select_detokenize_requests()
        lm = s._select_lm_requests()
        li = w.prepare_lm_inputs(lm, detok)
        w.run_detokenize(detok)
        assert [[r.request_id, list(r.audio_decode_idx)] for r in detok] == rec["detok"], step
        assert [r.request_id for r in lm] == rec["lm"], step
        pcm = []
        for r in detok:
            while not r.output_audio.empty():
                b = r.output_audio.get()
                pcm.append([r.request_id, len(b), hashlib.sha256(b).hexdigest()[:16]])
            if r.done_all:
                w.free_kv_cache(r)
        assert pcm == rec["pcm"], step                     # padding rule, trim length and PCM16 truncation
        if li is not None:
            for k, gk in (("qo_indptr", "qo"), ("paged_kv_indptr", "indptr"), ("paged_kv_indices", "indices"),
                          ("paged_kv_last_page_len", "last")):
                assert li[k] == rec[gk], (step, k)
            assert li["position_ids"].tolist() == rec["pos"] and li["is_prefill"] == rec["is_prefill"]
            assert int(li["input_ids"].shape[0]) == rec["n_rows"]
        else:
            assert "qo" not in rec
        for r in lm:
            k = len(r.lm_output_tokens)
            row = torch.tensor([[100 * (ord(r.request_id) - 64) + k, k, 7]], dtype=torch.long)
            r.input_tokens, r.input_masks, r.input_features = row, torch.ones(1, 3, dtype=torch.bool), torch.zeros(1, 8)
            r.lm_output_tokens.append(row)
            if k + 1 >= finish_after[r.request_id]:
                r.done_lm_generation, r.finish_reason = True, "stop_id_encountered"
            else:
                r.lm_output_audio_tokens.append(row)
        assert {r.request_id: list(r.kv_pages or []) for r in s.active_requests} == rec["pages"], step
        assert [r.request_id for r in s.active_requests if r.done_all] == rec["done"], step
    free = []
    while not w.empty_pages.empty():
        free.append(w.empty_pages.get())
    assert free == g["free_pages_after"]                   # FIFO free list order after all frees


def test_wire_format_and_full_loop_with_fake_lm():
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.worker import ModelWorker

    class W(ModelWorker):
        def run_lm_prefill(self, reqs, li):
            self._fake(reqs)

        def run_lm_decode(self, reqs, li):
            self._fake(reqs)

        def _fake(self, reqs):
            for r in reqs:
                k = len(r.lm_output_tokens)
                row = torch.tensor([[k + 1, k, 7]], dtype=torch.long)
                r.input_tokens, r.input_masks, r.input_features = row, torch.ones(1, 3, dtype=torch.bool), torch.zeros(1, 8)
                r.lm_output_tokens.append(row)
                if k + 1 >= 6:
                    r.done_lm_generation, r.finish_reason = True, "stop_id_encountered"
                else:
                    r.lm_output_audio_tokens.append(row)
    t = QueueTransport()
    s = Scheduler(W(model=FakePlugin(), max_num_pages=16, page_size=4, device="cpu"), transport=t)
    t.requests.put(encode_request("r1", "5"))
    t.requests.put(b"garbage-without-delimiter")
    t.requests.put(encode_request("r2", "2", is_streaming=False))
    s.run_until_idle(200)
    msgs = []
    while not t.results.empty():
        msgs.append(t.results.get())
    for rid in (b"r1", b"r2"):
        mine = [m for m in msgs if m.startswith(rid + b"|")]
        assert mine[-1].startswith(rid + b"|COMPLETION|")
        assert json.loads(mine[-1].split(b"|", 2)[2]) == {"status": "completed", "reason": "stop_id_encountered"}
        audio = b"".join(m.split(b"|", 2)[2] for m in mine[:-1])
        # 5 audio frames: one full chunk of 4 (20 samples) + a padded chunk of 1 trimmed to int(20*(1-0.5)/4)=2 samples
        assert len(audio) == 2 * (20 + 2)
    assert s.model_worker.empty_pages.qsize() == 16      # every page returned


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "voxhip.h")).read()
    declared = set(re.findall(r"\b(vox_[a-z0-9_]+)\s*\(", hdr))
    so = os.path.join(ROOT, "vox_serve_amd", "libvoxhip.so")
    if not os.path.exists(so):
        from vox_serve_amd import build
        build.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    exported = set(re.findall(r"\bT (vox_[a-z0-9_]+)", out))
    assert declared and declared <= exported, sorted(declared - exported)
    import ctypes
    L = ctypes.CDLL(so)                 # loads without a GPU; no compute entry point is called here
    assert L.vox_abi_version() == 1
    from vox_serve_amd import _native
    assert set(_native._SIGS) <= exported


def test_product_never_imports_the_oracle():
    for dp, _, files in os.walk(os.path.join(ROOT, "vox_serve_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b|libvoxref|voxref\.h|import_module\(.oracle", txt, re.M), f


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vox_serve_amd import _native as N
    with pytest.raises(N.VoxError):
        N.ctx()


def test_decoder_cache_ops():
    from vox_serve_amd.tokenizer.qwen3_codec import Qwen3TTSDecoderCache as C
    a, b = C(slot=torch.tensor([3], dtype=torch.int32)), C(slot=torch.tensor([5], dtype=torch.int32))
    cat = C.cat([a, b])
    assert cat.slot.tolist() == [3, 5] and cat[1:].slot.tolist() == [5]
    a.copy_from(b)
    assert a.slot.tolist() == [5]
    with pytest.raises(ValueError):
        C.cat([])


def test_qwen3_prompt_layout_matches_reference_rules():
    """custom-voice layout: 3 role + codec prefix (3, or 4 with a language id) + speaker + tts_bos + P text + tts_eos +
    (tts_pad, codec_bos)  (qwen3_tts.py:1617-1626, 1640-1778): 64 text tokens with a language id -> 75 rows."""
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel, Qwen3TTSTokens
    m = Qwen3TTSModel.__new__(Qwen3TTSModel)
    m.tokens, m.tts_model_type = Qwen3TTSTokens(spk_id={"vivian": 3066}), "custom_voice"
    from vox_serve_amd.engine import Qwen3Cfg
    m.config = Qwen3Cfg()
    P = 64
    ids = [11, 12, 13] + list(range(1000, 1000 + P)) + [21, 22, 23, 24, 25]
    toks, masks = m.layout(ids, language="english", speaker="Vivian")
    t = m.tokens
    assert toks.shape == (P + 11, 17)
    assert toks[:3, -1].tolist() == [11, 12, 13] and not masks[:3, -1].any()
    assert toks[3:7, 0].tolist() == [t.codec_think, t.codec_think_bos, 2050, t.codec_think_eos]
    assert toks[3:8, -1].tolist() == [t.tts_pad] * 5 and toks[7, 0] == 3066
    assert (toks[8, -1], toks[8, 0]) == (t.tts_bos, t.codec_pad)
    assert toks[9:9 + P, -1].tolist() == list(range(1000, 1000 + P)) and (toks[9:9 + P, 0] == t.codec_pad).all()
    assert (toks[-2, -1], toks[-2, 0]) == (t.tts_eos, t.codec_pad) and (toks[-1, -1], toks[-1, 0]) == (t.tts_pad, t.codec_bos)
    assert masks[3:, -1].all()
    toks2, _ = m.layout(ids, language="auto", speaker="vivian")
    assert toks2.shape[0] == P + 10 and toks2[3:6, 0].tolist() == [t.codec_nothink, t.codec_think_bos, t.codec_think_eos]


def test_qwen3_preprocess_layout_equals_reference(golden):
    """Every mode of the reference's Qwen3TTSModel.preprocess (g14: custom voice with language id / dialect speaker /
    instruct, voice design, x-vector-only and ICL cloning, input streaming): tokens and masks equal, row for row."""
    import json
    from vox_serve_amd.engine import Qwen3Cfg
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel, Qwen3TTSTokens
    g = golden("g14_qwen3_preprocess")
    ids = json.loads(str(g["special_ids"]))
    spk_id, dialect, lang = ids.pop("spk_id"), ids.pop("spk_is_dialect"), ids.pop("codec_language_id")
    for tag, kind, kw in json.loads(str(g["cases"])):
        m = Qwen3TTSModel.__new__(Qwen3TTSModel)
        m.tokens = Qwen3TTSTokens(**ids, codec_eos=300, codec_language_id=lang, spk_id=spk_id, spk_is_dialect=dialect)
        m.tts_model_type, m.config = kind, Qwen3Cfg(n_groups=4)
        icl = kind == "base" and not kw.get("x_vector_only_mode")
        toks, masks, info = m.layout(
            g[f"{tag}_prompt_ids"].tolist(), kw.get("language", "english"), kw.get("speaker", "ryan"),
            g[f"{tag}_instruct_ids"].tolist() if f"{tag}_instruct_ids" in g else None, bool(kw.get("is_input_streaming")),
            ref_text_ids=g[f"{tag}_ref_text_ids"].tolist() if icl else None,
            ref_codes0=g["ref_codes"][:, 0].tolist() if icl else None, return_info=True)
        assert torch.equal(toks, torch.from_numpy(g[f"{tag}_tokens"])), tag
        assert torch.equal(masks, torch.from_numpy(g[f"{tag}_masks"])), tag
        feat_rows = np.nonzero(g[f"{tag}_features"].any(axis=1))[0].tolist()
        want = ([info["speaker_row"]] if info["speaker_row"] is not None else []) + \
            (list(range(info["icl_row"], toks.shape[0])) if info["icl_row"] is not None else [])
        assert feat_rows == want, tag
    m.tts_model_type = "base"
    with pytest.raises(ValueError):
        m.layout([1, 2, 3, 4], "auto", None, None, True, ref_text_ids=[1, 2, 3, 4, 5, 6], ref_codes0=[1, 2])


def test_reference_clip_loading_and_resampling(tmp_path):
    """Qwen3TTSModel._load_audio_to_np reads (waveform, sr) pairs, .npy and 16-bit wav (stereo averaged); clips at other rates are
    brought to 24 kHz before the encoders (the reference resamples with librosa on the host, qwen3_tts.py:1513-1518)."""
    import wave
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel
    m = Qwen3TTSModel.__new__(Qwen3TTSModel)
    t = np.arange(16000) / 16000.0
    x = (0.5 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    a, sr = m._load_audio_to_np((x, 16000))
    assert sr == 16000 and np.array_equal(a, x)
    np.save(tmp_path / "c.npy", x)
    assert np.array_equal(m._load_audio_to_np(str(tmp_path / "c.npy"))[0], x)
    pcm = (x * 32767).astype("<i2")
    with wave.open(str(tmp_path / "s.wav"), "wb") as f:
        f.setnchannels(2), f.setsampwidth(2), f.setframerate(16000), f.writeframes(np.stack([pcm, pcm], 1).tobytes())
    a, sr = m._load_audio_to_np(str(tmp_path / "s.wav"))
    assert sr == 16000 and a.shape == x.shape and np.abs(a - pcm / 32768.0).max() < 1e-7
    y = Qwen3TTSModel._to_24k(a, sr)
    assert y.shape == (24000,) and y.dtype == np.float32
    spec = np.abs(np.fft.rfft(y * np.hanning(len(y))))
    assert abs(int(np.argmax(spec)) - 440) <= 1 and abs(float(np.abs(y).max()) - 0.5) < 0.02
    assert Qwen3TTSModel._to_24k(x, 24000) is not None and np.array_equal(Qwen3TTSModel._to_24k(x, 24000), x)
    with pytest.raises(ValueError):
        m._load_audio_to_np("clip.mp3")


def test_strided_conv_as_two_taps_over_frame_rows():
    """The speech-tokenizer encoder runs a causal stride-r conv (kernel 2r) as a two-tap GEMM over the input viewed as rows of r frames
    (tokenizer/qwen3_codec_encoder.py::strided_taps); here the packing is checked against F.conv1d on the CPU, ragged tail included
    (MimiConv1d pads the tail with zeros up to a whole stride)."""
    import torch.nn.functional as F
    from vox_serve_amd.tokenizer.qwen3_codec_encoder import strided_taps
    g = torch.Generator().manual_seed(0)
    for r, cin, cout, L in ((4, 8, 16, 40), (5, 6, 4, 43), (2, 16, 16, 7)):
        w, b = torch.randn(cout, cin, 2 * r, generator=g), torch.randn(cout, generator=g)
        x = torch.randn(L, cin, generator=g)                               # time-major [t][C]
        Lo = -(-L // r)
        want = F.conv1d(F.pad(x.t()[None], (r, Lo * r - L)), w, b, stride=r)[0].t()          # causal: left pad 2r - r, zero tail
        rows = F.pad(x, (0, 0, 0, Lo * r - L)).reshape(Lo, r * cin)
        taps = strided_taps(w, r)
        prev = torch.cat([torch.zeros(1, r * cin), rows[:-1]])
        got = prev @ taps[0].t() + rows @ taps[1].t() + b
        assert got.shape == want.shape == (Lo, cout) and torch.allclose(got, want, atol=1e-4)
    with pytest.raises(ValueError):
        strided_taps(torch.zeros(2, 2, 7), 4)


def test_mel_filterbank_table_equals_the_oracle_restatement():
    """The speaker encoder's constant mel table (built on the host by the plugin) equals the oracle's restatement of librosa.filters.mel —
    the table the reference-generated fixture g15 was computed with — for both checkpoint variants (128 and 80 mels); rows are
    area-normalised triangles (Slaney): non-negative, each with one peak."""
    from oracle import spk_ref as SR
    from vox_serve_amd.model.qwen3_tts_speaker import slaney_mel_filterbank
    for n_mels in (128, 80):
        a, b = slaney_mel_filterbank(24000, 1024, n_mels, 0.0, 12000.0), SR.mel_filterbank(24000, 1024, n_mels, 0.0, 12000.0)
        assert a.shape == (n_mels, 513) and a.dtype == np.float32 and np.array_equal(a, b)
        assert (a >= 0).all() and (a.sum(axis=1) > 0).all()
        peaks = a.argmax(axis=1)
        assert (np.diff(peaks) > 0).all()


def test_base_checkpoint_loader_wires_the_clone_encoders(monkeypatch):
    """load_model's Qwen3 loader: a `...-Base` name selects the voice-clone model type and hands the plugin the speaker encoder
    (speaker_encoder.* of the LM checkpoint) and the speech tokenizer's encoder half (encoder.* of the codec checkpoint); the other
    names do not."""
    import inspect
    import vox_serve_amd.model as M
    from vox_serve_amd.model import qwen3_tts as Q
    accepted = set(inspect.signature(Q.Qwen3TTSModel.__init__).parameters)
    seen = {}

    class Capture:
        def __init__(self, model_name, weights, codec_weights, **kw):
            seen.clear()
            seen.update(kw)

    monkeypatch.setattr(Q, "Qwen3TTSModel", Capture)
    monkeypatch.setattr(M, "_load_safetensors_dir", lambda path, device: {"encoder.layers.0.conv.weight": 1, "decoder.x": 2})
    lm = {"speaker_encoder.fc.weight": 3, "talker.model.norm.weight": 4}
    M.MODEL_REGISTRY["qwen/qwen3-tts-12hz-1.7b-base"]("Qwen/Qwen3-TTS-12Hz-1.7B-Base", device="cpu", weights=lm, codec_weights={},
                                                      checkpoint_dir="/ckpt")
    assert seen["tts_model_type"] == "base" and set(seen) <= accepted
    assert seen["speaker_encoder_weights"] == {"fc.weight": 3} and seen["audio_encoder_weights"] == {"layers.0.conv.weight": 1}
    M.MODEL_REGISTRY["qwen3-tts"]("qwen3-tts", device="cpu", weights=lm, codec_weights={}, checkpoint_dir="/ckpt")
    assert seen["tts_model_type"] == "custom_voice" and "speaker_encoder_weights" not in seen
    M.MODEL_REGISTRY["qwen3-tts-voice-design"]("qwen3-tts-voice-design", device="cpu", weights=lm, codec_weights={}, checkpoint_dir="/ckpt")
    assert seen["tts_model_type"] == "voice_design"


def test_registry_errors():
    from vox_serve_amd.model import load_model
    with pytest.raises(ValueError):
        load_model("no-such-model", device="cpu")


DP_SCRIPT = r"""
import os, sys, hashlib, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from vox_serve_amd.worker.dp_pool import broadcast_weights, run_sharded, route
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
g = torch.Generator().manual_seed(0)
ref = {"a": torch.randn(1000, generator=g), "b": torch.randn(37, 5, generator=g).to(torch.bfloat16), "c": torch.randn(3, generator=g)}
w = {k: (v.clone() if rank == 0 else torch.zeros_like(v)) for k, v in ref.items()}
broadcast_weights(w, src=0, bucket_bytes=1024)
assert all(torch.equal(w[k], ref[k]) for k in ref), "weight broadcast"
reqs = [f"req{i}" for i in range(7)]
def serve(mine):           # stand-in for scheduler+worker: output depends only on the request, as in real DP
    return {r: hashlib.sha256((r + "|" + "".join(f"{float(x):.6f}" for x in w["c"])).encode()).digest() for r in mine}
out = run_sharded(reqs, serve)
if rank == 0:
    single = serve(reqs)
    assert out == single, "DP-N output for request i must equal the single-process output"
    assert [route(i, world) for i in range(4)] == [0, 1, 0, 1]
    print("DP_OK")
dist.destroy_process_group()
"""


def test_dp_pool_two_ranks_gloo(tmp_path):
    script = tmp_path / "dp.py"
    script.write_text(DP_SCRIPT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0 and "DP_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


# ---------------------------------------------------------------- g8: scheduler policies ------------------------------
def _g8_requests(spec):
    from vox_serve_amd.requests import Request
    out = []
    for s_ in spec:
        r = Request(request_id=s_["id"], prompt="x")
        r.done_lm_prefill, r.done_lm_generation = s_["done_lm_prefill"], s_["done_lm_generation"]
        r.input_length = s_["input_length"] or None
        r.lm_output_audio_tokens = [None] * s_["n_tokens"]
        r.next_audio_decode_idx = list(s_["next_idx"])
        r.is_streaming, r.is_pressing = s_["is_streaming"], s_["is_pressing"]
        r.is_input_streaming, r.prefill_ready, r.text_complete = s_["is_input_streaming"], s_["prefill_ready"], s_["text_complete"]
        for t in range(s_["n_pending_text"]):
            r.pending_text_tokens.put(100 + t)
        r.chunk_send_timestamps = [1000.0 + t for t in s_["chunk_times"]]
        r.chunk_durations = list(s_["chunk_durs"])
        out.append(r)
    return out


def test_scheduler_policies_match_reference():
    """Online / Offline / InputStreaming selection on 60 scripted request states == what the reference classes chose."""
    import json
    import types
    from vox_serve_amd.scheduler import load_scheduler
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g8_scheduler_policies.json")))
    for ci, case in enumerate(g["cases"]):
        for name, want in case["out"].items():
            w = types.SimpleNamespace(detokenize_interval=case["interval"], detokenize_overlap=case["overlap"],
                                      available_batch_sizes=None, supports_audio_input=False)
            if case["graph_worker"]:      # limits of the graph-capturing worker (cuda_graph_worker.py:61-62)
                w.prefill_graph_batch_size, w.cuda_graph_seq_len_buckets = case["prefill_graph_batch_size"], case["seq_len_buckets"]
            else:                          # eager worker: the reference falls back to (max_batch_size, 1024)
                w.prefill_graph_batch_size, w.cuda_graph_seq_len_buckets = case["max_batch_size"], [1024]
            sch = load_scheduler(name, model_worker=w, max_batch_size=case["max_batch_size"])
            sch.active_requests = _g8_requests(case["requests"])
            if name == "online":
                sch._update_pressing_status(now=case["now"])
                assert [r.is_pressing for r in sch.active_requests] == want["pressing"], (ci, name)
            assert [r.request_id for r in sch._select_lm_requests()] == want["lm"], (ci, name, "lm")
            assert [bool(r.waiting_for_text) for r in sch.active_requests] == want["waiting_for_text"], (ci, name)
            if name != "input_streaming":
                sel = sch._select_detokenize_requests()
                got = [[r.request_id, list(r.next_audio_decode_idx), bool(r.done_all)] for r in sel]
                assert got == want["detok"], (ci, name, got, want["detok"])
                assert [bool(r.done_all) for r in sch.active_requests] == want["done_all"], (ci, name)


def test_detokenize_batching_window():
    """Opt-in `detokenize_min_batch`: a ready window waits for company unless it is a first chunk, a tail, or a second window."""
    import types
    from vox_serve_amd.requests import Request
    from vox_serve_amd.scheduler import Scheduler

    def mk(i, n_tokens, next_idx, done=False):
        r = Request(request_id=f"r{i}", prompt="x")
        r.done_lm_prefill, r.done_lm_generation = True, done
        r.lm_output_audio_tokens = [None] * n_tokens
        r.next_audio_decode_idx = list(next_idx)
        return r
    w = types.SimpleNamespace(detokenize_interval=10, detokenize_overlap=0, available_batch_sizes=None, supports_audio_input=False)
    sel = lambda s: [r.request_id for r in s._select_detokenize_requests()]
    ref = Scheduler(w, max_batch_size=8)
    bat = Scheduler(w, max_batch_size=8, detokenize_min_batch=4)
    # two of six generating requests have their second window ready: the reference policy takes them, the window waits
    states = lambda: [mk(0, 20, [0]), mk(1, 21, [0]), mk(2, 15, [0]), mk(3, 13, [0]), mk(4, 12, [0]), mk(5, 11, [0])]
    ref.active_requests, bat.active_requests = states(), states()
    assert sel(ref) == ["r0", "r1"] and sel(bat) == []
    assert [r.next_audio_decode_idx for r in bat.active_requests] == [[0]] * 6          # nothing was advanced
    # ... until four are ready
    st = states(); st[2], st[3] = mk(2, 20, [0]), mk(3, 22, [0])
    bat.active_requests = st
    assert sel(bat) == ["r0", "r1", "r2", "r3"]
    # a first chunk, a tail, or a piled-up second window go at once (with everything else that is ready)
    bat.active_requests = [mk(0, 10, []), mk(1, 20, [0])] + states()[2:]
    assert sel(bat) == ["r0", "r1"]
    bat.active_requests = [mk(0, 14, [0], done=True)] + states()[1:]
    assert sel(bat) == ["r0", "r1"]
    bat.active_requests = [mk(0, 30, [0])] + states()[2:]
    assert sel(bat) == ["r0"]
    # fewer generating requests than the window: all of them ready is enough
    bat.active_requests = [mk(0, 20, [0]), mk(1, 20, [0])]
    assert sel(bat) == ["r0", "r1"]


def test_input_streaming_messages_match_reference():
    import json
    import types
    from vox_serve_amd.scheduler import load_scheduler

    class Tok:   # the deterministic tokenizer the golden generator used
        def encode(self, text, add_special_tokens=False):
            return [sum(map(ord, text[i:i + 3])) % 5000 + 10 for i in range(0, len(text), 3)]

        def decode(self, ids, skip_special_tokens=True):
            return "".join(f"<{i}>" for i in ids)

    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g8_scheduler_policies.json")))
    w = types.SimpleNamespace(detokenize_interval=10, detokenize_overlap=0, available_batch_sizes=None,
                              supports_audio_input=False, model=types.SimpleNamespace(text_tokenizer=Tok()))
    sch = load_scheduler("input_streaming", model_worker=w, max_batch_size=4)
    for i, (msg, want) in enumerate(zip(g["stream_script"], g["stream_log"])):
        if i == 5:
            sch.active_requests[0].done_lm_prefill = True
        req = sch._handle_request_payload(msg.encode())
        if req is not None:
            sch.active_requests.append(req)
        got = [{"id": r.request_id, "prompt": r.prompt, "buffer": r.input_text_buffer, "prefill_ready": r.prefill_ready,
                "pending": list(r.pending_text_tokens.queue), "total_text_tokens": r.total_text_tokens,
                "text_complete": r.text_complete, "done_all": r.done_all, "finish_reason": r.finish_reason,
                "is_streaming": r.is_streaming, "model_kwargs": r.model_kwargs} for r in sch.active_requests]
        assert got == want, (i, msg)


def test_disaggregation_scheduler_runs_two_pipelines():
    """LM thread + detokenizer thread over a fake worker: every request gets all of its audio and one completion."""
    import types
    import torch
    from vox_serve_amd.scheduler import QueueTransport, encode_request, load_scheduler

    class FakeWorker:
        detokenize_interval, detokenize_overlap, supports_audio_input, available_batch_sizes = 4, 0, False, None
        prefill_graph_batch_size, cuda_graph_seq_len_buckets = 8, [1024]

        def __init__(self):
            self.freed = []

        def prepare_lm_inputs(self, lm, det):
            for r in lm:
                if not r.done_lm_prefill:
                    r.done_lm_prefill, r.next_position_id = True, 1
            return {"is_prefill": False}

        def run_lm_prefill(self, reqs, li):
            self.run_lm_decode(reqs, li)

        def run_lm_decode(self, reqs, li):
            for r in reqs:
                r.lm_output_audio_tokens.append(torch.zeros(1, 1, dtype=torch.long))
                if len(r.lm_output_audio_tokens) >= 10:
                    r.done_lm_generation, r.finish_reason = True, "max_tokens_reached"

        def run_detokenize(self, reqs):
            for r in reqs:
                for d in r.audio_decode_idx:
                    n = len(r.lm_output_audio_tokens[d:d + 4])
                    r.output_audio.put(bytes(2 * n))
                if r.done_lm_generation and r.audio_decode_idx and r.audio_decode_idx[-1] + 4 >= len(r.lm_output_audio_tokens):
                    r.done_all = True

        def free_kv_cache(self, r):
            self.freed.append(r.request_id)

    tr, w = QueueTransport(), FakeWorker()
    sch = load_scheduler("disaggregation", model_worker=w, max_batch_size=4, transport=tr)
    for i in range(5):
        tr.requests.put(encode_request(f"q{i}", "hello"))
    sch.run_until_idle(timeout_s=30)
    msgs = []
    while not tr.results.empty():
        msgs.append(tr.results.get())
    for i in range(5):
        audio = [m for m in msgs if m.startswith(f"q{i}|AUDIO|".encode())]
        done = [m for m in msgs if m.startswith(f"q{i}|COMPLETION|".encode())]
        assert sum(len(m.split(b"|", 2)[2]) for m in audio) == 20 and len(done) == 1, (i, len(audio), len(done))
    assert sorted(w.freed) == [f"q{i}" for i in range(5)] and not sch.active_requests


def test_overlapped_detokenize_keeps_the_tail_window_when_eos_lands_on_a_window_boundary():
    """`finish_detokenize` runs after the LM step of the same iteration (the codec chunk overlaps the frame); the done_all rule
    (worker/base.py:674-678 of the reference) must still see the request as it was BEFORE that step: for a model with
    detokenize_overlap > 0 the window that starts `interval - overlap` later is real audio.  The overlapped loop must produce
    the byte stream of the reference order (detokenize, send, then the LM step) for every EOS position."""
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.worker import ModelWorker

    class Plug(FakePlugin):
        detokenize_overlap = 1

    def serve(n_frames, overlap):
        class W(ModelWorker):
            def run_lm_prefill(self, reqs, li):
                self._fake(reqs)
            run_lm_decode = run_lm_prefill

            def _fake(self, reqs):
                for r in reqs:
                    k = len(r.lm_output_tokens)
                    row = torch.tensor([[3 * k + 1, k, 7]], dtype=torch.long)
                    r.input_tokens, r.input_masks, r.input_features = row, torch.ones(1, 3, dtype=torch.bool), torch.zeros(1, 8)
                    r.lm_output_tokens.append(row)
                    if k + 1 > n_frames:
                        r.done_lm_generation, r.finish_reason = True, "stop_id_encountered"
                    else:
                        r.lm_output_audio_tokens.append(row)
        t = QueueTransport()
        s = Scheduler(W(model=Plug(), max_num_pages=32, page_size=4, device="cpu"), transport=t)
        s.overlap_detokenize = overlap
        t.requests.put(encode_request("r", "3"))
        s.run_until_idle(200)
        msgs = []
        while not t.results.empty():
            msgs.append(t.results.get())
        assert msgs[-1].startswith(b"r|COMPLETION|") and sum(m.startswith(b"r|COMPLETION|") for m in msgs) == 1
        return [m.split(b"|", 2)[2] for m in msgs[:-1]]

    for n in range(3, 15):
        assert serve(n, True) == serve(n, False), n


def test_failed_prefill_launch_answers_once_and_rolls_the_piggy_backed_rows_back():
    """A prefill launch that raises: the new prompt gets ONE error COMPLETION (no AUDIO / second COMPLETION from the
    detokenize half of the same iteration), and the decode rows that shared the step get their KV bookkeeping back."""
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.worker import ModelWorker

    class W(ModelWorker):
        seen = []

        def run_lm_prefill(self, reqs, li):
            if any(r.request_id == "bad" for r in reqs):
                raise RuntimeError("boom")
            self._fake(reqs)

        def run_lm_decode(self, reqs, li):
            W.seen.append([(r.request_id, r.kv_token_len, r.next_position_id) for r in reqs])
            self._fake(reqs)

        def _fake(self, reqs):
            for r in reqs:
                k = len(r.lm_output_tokens)
                row = torch.tensor([[k + 1, k, 7]], dtype=torch.long)
                r.input_tokens, r.input_masks, r.input_features = row, torch.ones(1, 3, dtype=torch.bool), torch.zeros(1, 8)
                r.lm_output_tokens.append(row)
                if k + 1 >= 9:
                    r.done_lm_generation, r.finish_reason = True, "stop_id_encountered"
                else:
                    r.lm_output_audio_tokens.append(row)
    t = QueueTransport()
    w = W(model=FakePlugin(), max_num_pages=16, page_size=4, device="cpu")
    s = Scheduler(w, transport=t)
    t.requests.put(encode_request("ok", "4"))
    for _ in range(3):
        s._step()
    t.requests.put(encode_request("bad", "3"))
    s.run_until_idle(200)
    msgs = []
    while not t.results.empty():
        msgs.append(t.results.get())
    bad = [m for m in msgs if m.startswith(b"bad|")]
    assert len(bad) == 1 and json.loads(bad[0].split(b"|", 2)[2])["status"] == "error"
    ok = [m for m in msgs if m.startswith(b"ok|")]
    assert sum(m.startswith(b"ok|COMPLETION|") for m in ok) == 1 and json.loads(ok[-1].split(b"|", 2)[2])["status"] == "completed"
    # the surviving row's KV length advances by exactly one per executed decode step (the failed step does not count)
    lens = [row[0][1] for row in W.seen if row and row[0][0] == "ok"]
    assert lens == list(range(lens[0], lens[0] + len(lens)))
    assert w.empty_pages.qsize() == 16


def test_undo_decode_advance_puts_the_streaming_text_state_back_and_keeps_the_step_in_flight():
    """Roll-back of decode rows after a failed launch: the text token (or EOS) an input-streaming row consumed in that step
    is handed back in order, and a deferred request-state update still pending from the step before is run, not dropped."""
    from types import SimpleNamespace
    from vox_serve_amd.requests import Request
    from vox_serve_amd.worker import ModelWorker
    w = ModelWorker(model=FakePlugin(), max_num_pages=16, page_size=4, device="cpu")
    w.model.tokens = SimpleNamespace(tts_eos=901, tts_pad=902)

    def row(rid, queued, complete=False):
        r = Request(request_id=rid, prompt="x")
        r.done_lm_prefill, r.is_input_streaming, r.text_complete = True, True, complete
        r.input_tokens = torch.zeros(1, 3, dtype=torch.long)
        r.kv_pages, r.kv_token_len, r.kv_last_page_len, r.next_position_id = [w.empty_pages.get()], 3, 3, 4
        for t_ in queued:
            r.pending_text_tokens.put(t_)
        return r
    a, b, c = row("a", [11, 12]), row("b", [], complete=True), row("c", [])
    for r in (a, b, c):
        w._inject_streaming_text_token(r)
        r.kv_token_len += 1; r.kv_last_page_len += 1; r.next_position_id += 1       # what prepare_lm_inputs advances
    assert [int(r.input_tokens[0, -1]) for r in (a, b, c)] == [11, 901, 902] and b.eos_injected and a.text_token_cursor == 1
    ran = []
    w._pending = lambda: (ran.append("finish"), setattr(w, "_pending", None))
    w.undo_decode_advance([a, b, c])
    assert ran == ["finish"] and w._pending is None and w._resident is None
    assert list(a.pending_text_tokens.queue) == [11, 12] and a.text_token_cursor == 0      # same token again on the retry
    assert not b.eos_injected and c.pending_text_tokens.empty()
    assert all((r.kv_token_len, r.kv_last_page_len, r.next_position_id) == (3, 3, 4) for r in (a, b, c))
    w._inject_streaming_text_token(a); w._inject_streaming_text_token(b)
    assert int(a.input_tokens[0, -1]) == 11 and int(b.input_tokens[0, -1]) == 901


def test_device_bound_tokenizers_and_per_device_contexts():
    """`audio_decoder_device` plumbing that can be checked without a GPU: the guard is a no-op for CPU / index-less devices, the
    decorator wraps the constructor and the public methods only (private helpers run inside the caller's guard), and every
    native tokenizer class carries it."""
    import contextlib
    from vox_serve_amd import _native as N
    from vox_serve_amd.tokenizer.base import device_bound
    assert isinstance(N.device_guard("cpu"), contextlib.nullcontext) and isinstance(N.device_guard("cuda"), contextlib.nullcontext)
    calls = []

    @device_bound
    class T:
        def __init__(self, x, device="cpu"):
            self.device, self.x = device, x

        def decode(self, y):
            calls.append("decode")
            return self._helper(y)

        def _helper(self, y):
            return self.x + y

        def __call__(self, y):
            return self.decode(y)
    t = T(2, device="cpu")
    assert t(3) == 5 and t.decode(1) == 3 and calls == ["decode", "decode"]
    assert T.decode.__wrapped__ is not None and not hasattr(T._helper, "__wrapped__")
    import vox_serve_amd.tokenizer.cosyvoice2 as c2, vox_serve_amd.tokenizer.glm as gl, vox_serve_amd.tokenizer.mimi as mi
    import vox_serve_amd.tokenizer.qwen3_codec as qc, vox_serve_amd.tokenizer.snac as sn
    for cls in (c2.CosyVoice2Decoder, gl.GLMAudioDecoder, mi.MimiDecoder, qc.Qwen3TTSDecoder, sn.SNACDecoder):
        assert hasattr(cls.__init__, "__wrapped__"), cls


def test_status_row_gate_is_on_the_buffer_not_on_the_inherited_method():
    """Round-5 advice (high): LMEngine / CSMEngine subclass Qwen3Engine without running its __init__, so they HAVE snapshot_src /
    read_ids but no `_out_block`; the worker's async snapshot gated on the method and crashed for every non-Qwen3 model."""
    import types
    from vox_serve_amd import _native as N
    from vox_serve_amd.engine import CSMEngine, LMEngine, Qwen3Engine
    from vox_serve_amd.worker.base import engine_has_status_row
    for cls in (LMEngine, CSMEngine):
        e = object.__new__(cls)
        assert hasattr(e, "snapshot_src") and not engine_has_status_row(e)
        with pytest.raises(N.VoxError, match="no status row"):
            e.snapshot_src(2)
    q = object.__new__(Qwen3Engine)
    assert not engine_has_status_row(q)                       # not constructed: no buffer yet
    q._out_block = torch.zeros(3, 17, dtype=torch.int32)
    q.status_row = q._out_block[0]
    assert engine_has_status_row(q) and q.snapshot_src(1).shape == (2, 17)
    assert not engine_has_status_row(types.SimpleNamespace(out_ids=torch.zeros(2)))
