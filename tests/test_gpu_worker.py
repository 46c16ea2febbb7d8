"""GPU: the whole serving loop — Scheduler -> ModelWorker -> Qwen3TTSModel (native engine + native codec) — on a tiny
synthetic model: wire format, streaming chunks, trim rule, resource release, determinism, and agreement of the
worker-driven token stream with a hand-driven engine (the path the oracle parity tests pin)."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def build(dev, max_tokens=None, max_prefill_tokens=64, audio_decoder_device=None):
    from oracle import qwen3_ref as QR, voxref as vr           # weights recipe only (test side)
    from tests.test_gpu_codec import small_cfg
    from oracle import qwen3_codec_ref as CR
    from tests.test_gpu_qwen3 import to_engine_cfg
    from vox_serve_amd.model.qwen3_tts import Qwen3TTSModel, Qwen3TTSTokens
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.tokenizer.qwen3_codec import Qwen3CodecConfig
    cfg = QR.tiny_cfg()
    cfg.n_groups = 4
    W = {k: vr.to_torch(v).to(dev) for k, v in QR.random_weights(cfg, 7, 0.08).items()}
    cc = small_cfg()
    pc = Qwen3CodecConfig(**{k: getattr(cc, k) for k in Qwen3CodecConfig.__dataclass_fields__})
    toks = Qwen3TTSTokens(tts_bos=5, tts_eos=6, tts_pad=cfg.tts_pad_id, codec_bos=10, codec_eos=cfg.eos_id, codec_pad=11,
                          codec_think=12, codec_nothink=13, codec_think_bos=14, codec_think_eos=15,
                          codec_language_id={"english": 16}, spk_id={"a": 17})
    m = Qwen3TTSModel("tiny", W, CR.random_codec_weights(cc, 3), config=to_engine_cfg(cfg), codec_config=pc, tokens=toks,
                      device=str(dev), detokenize_interval=4, max_batch_size=4, page_size=16, max_num_pages=64,
                      max_seq_len=512, max_prefill_tokens=max_prefill_tokens, audio_decoder_device=audio_decoder_device)
    m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=max_tokens, repetition_penalty=1.05, repetition_window=-1)
    return m, cfg


def serve(m, prompts, async_scheduling=False, overlap_detokenize=True, detokenize_min_batch=0):
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.worker import ModelWorker
    t = QueueTransport()
    w = ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=16, device=m.device)
    s = Scheduler(w, max_batch_size=4, transport=t, async_scheduling=async_scheduling, detokenize_min_batch=detokenize_min_batch)
    s.overlap_detokenize = overlap_detokenize
    for rid, ids in prompts.items():
        t.requests.put(encode_request(rid, "", model_kwargs={"prompt_token_ids": ids, "speaker": "a"}))
    s.run_until_idle(2000)
    out = {rid: {"pcm": b"", "done": None} for rid in prompts}
    while not t.results.empty():
        msg = t.results.get()
        rid, kind, body = msg.split(b"|", 2)
        if kind == b"AUDIO":
            out[rid.decode()]["pcm"] += body
        else:
            out[rid.decode()]["done"] = json.loads(body)
    return out, w


def test_scheduler_worker_engine_codec_end_to_end():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    m, cfg = build(dev, max_tokens=30)
    prompt = [1, 2, 3, 40, 41, 42, 43, 7, 8, 9, 10, 11]           # 3 role + 4 text + 5 template tail
    out, w = serve(m, {"r1": prompt, "r2": prompt[:3] + [50, 51] + prompt[-5:]})
    for rid, o in out.items():
        assert o["done"]["status"] == "completed" and o["done"]["reason"] in ("max_tokens_reached", "stop_id_encountered")
        assert len(o["pcm"]) % 2 == 0 and len(o["pcm"]) > 0
        pcm = np.frombuffer(o["pcm"], dtype=np.int16)
        assert np.abs(pcm).max() > 50                                # audible signal, int16 range
    assert w.empty_pages.qsize() == 64                               # KV pages returned
    assert len(m.audio_decoder._free_slots) == m.audio_decoder.max_slots   # codec slots returned
    # determinism: the same arrivals give byte-identical audio
    out2, _ = serve(m, {"r1": prompt, "r2": prompt[:3] + [50, 51] + prompt[-5:]})
    assert all(out[r]["pcm"] == out2[r]["pcm"] for r in out)
    # a request served alone: frames generated == max_tokens rule, samples = frames*hop with the reference's trim
    solo, w2 = serve(m, {"r1": prompt})
    hop = m.audio_decoder.hop
    n_prompt = 3 + 3 + 1 + 1 + 4 + 2
    frames_audio = len(solo["r1"]["pcm"]) // 2 / hop
    assert 1 <= frames_audio <= 30 - n_prompt + 1
    m.engine.close(); m.audio_decoder.close()


def _drive_worker(m, prompts, steps, page=16, async_scheduling=False):
    """ModelWorker host loop (prepare_lm_inputs / run_lm_prefill / run_lm_decode), one prefill per step like the scheduler.
    async_scheduling: the order of Scheduler._step_async — run_lm_* hand back the request-state update as a coroutine, a prefill's is
    run at once, a decode step's while the NEXT step is already enqueued (so a finished request is seen one step late)."""
    from vox_serve_amd.requests import Request
    from vox_serve_amd.worker import ModelWorker
    w = ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=page, device=m.device)
    w.async_scheduling = async_scheduling

    def run(task):
        if task is not None:
            try:
                task.send(None)
            except StopIteration:
                pass
    reqs = []
    for i, kw in enumerate(prompts):
        r = Request(request_id=f"w{i}", prompt="", model_kwargs=kw)
        run(w.run_lm_prefill([r] + [], w.prepare_lm_inputs([r], [])))
        reqs.append(r)
    prev = None
    for _ in range(steps):
        live = [r for r in reqs if not r.done_lm_generation]
        if not live:
            break
        task = w.run_lm_decode(live, w.prepare_lm_inputs(live, []))
        assert (task is not None) == async_scheduling
        run(prev)
        prev = task
    run(prev)
    w.drain()
    w.async_scheduling = False
    return reqs, w


@pytest.mark.parametrize("async_scheduling", [False, True])
def test_worker_drives_glm_cosyvoice2_and_csm_plugins(async_scheduling):
    """The single-stack and CSM plugins through ModelWorker: token streams equal the oracle's (short prompts: bit-exact),
    stop / audio-token bookkeeping follows the reference plugins' `sampling`.  async_scheduling = True: the deferred request-state
    update (pinned snapshot of out_ids) of engines WITHOUT a status row — LMEngine and CSMEngine inherit Qwen3Engine's snapshot_src
    method but not its buffer (round-5 advice: the worker gated on the method and raised AttributeError for every non-Qwen3 model)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    from oracle import csm_ref as CR, lm_ref as LR, qwen3_ref as QR, voxref as vr
    from vox_serve_amd.sampling import SamplingConfig
    greedy = SamplingConfig(greedy=True)
    kw = dict(device=str(dev), max_batch_size=4, page_size=16, max_num_pages=64, max_seq_len=512, max_prefill_tokens=64)

    # ---- GLM-4-Voice (tiny) ----
    from vox_serve_amd.model.glm_voice import GLMVoiceConfig, GLMVoiceModel
    cfg = LR.tiny_glm_cfg()
    c = cfg.stack
    S = LR.random_glm_state_dict(cfg, seed=3, std=0.08)
    pc = GLMVoiceConfig(ffn_hidden_size=c.ffn, hidden_size=c.hidden, multi_query_group_num=c.kv_heads, num_attention_heads=c.heads,
                        num_layers=c.layers, padded_vocab_size=cfg.vocab_out, vocab_size=cfg.vocab_out,
                        eos_token_id=[cfg.vocab_out - 3, cfg.vocab_out - 2, cfg.vocab_out - 1], audio_offset=cfg.vocab_out // 2)
    m = GLMVoiceModel("tiny-glm", {k: vr.to_torch(v).to(dev) for k, v in S.items()}, config=pc, sampling=greedy, max_pos=512, **kw)
    prompts = [[5, 17, 99, 300, 7], [1200, 4, 8]]
    reqs, w = _drive_worker(m, [{"prompt_token_ids": p} for p in prompts], 12, async_scheduling=async_scheduling)
    ref = LR.LMRef(cfg, LR.from_glm_state_dict(cfg, S), page_size=16, max_pages=64)
    rr = []
    for p in prompts:
        q = LR.LMRequest()
        ref.sample(ref.prefill(q, np.array(p, np.int32)), [q])
        rr.append(q)
    for _ in range(12):
        ref.sample(ref.decode(rr), rr)
    for q, r in zip(rr, reqs):
        got = [int(t[0, 0]) for t in r.lm_output_tokens]
        assert got == q.tokens[: len(got)], (got, q.tokens)
        assert [int(t[0, 0]) for t in r.lm_output_audio_tokens] == [t for t in got if t >= pc.audio_offset and t not in pc.eos_token_id]
    m.engine.close()

    # ---- CosyVoice2 (tiny): prompt rows are embeddings, decode rows speech ids ----
    from vox_serve_amd.model.cosyvoice2 import CosyVoice2Config, CosyVoice2Model
    cfg = LR.tiny_cosyvoice2_cfg()
    c = cfg.stack
    S = LR.random_cosyvoice2_state_dict(cfg, seed=4, std=0.08)
    St = {k: vr.to_torch(v).to(dev) for k, v in S.items()}
    pc = CosyVoice2Config(llm_input_size=c.hidden, llm_output_size=c.hidden, speech_token_size=cfg.vocab_out - 3, hidden_size=c.hidden,
                          intermediate_size=c.ffn, num_attention_heads=c.heads, num_key_value_heads=c.kv_heads, num_hidden_layers=c.layers)
    ref_ids, ref_speech = torch.tensor([3, 9]), torch.tensor([7, 100])
    m = CosyVoice2Model("tiny-cosy", St, config=pc, sampling=greedy, max_pos=512,
                        speaker_ref={"ref_text_ids": ref_ids, "prompt_speech_token": ref_speech}, **kw)
    reqs, w = _drive_worker(m, [{"prompt_token_ids": [11, 12]}], 10, async_scheduling=async_scheduling)
    text = torch.cat([ref_ids, torch.tensor([11, 12])])
    feats = torch.cat([St["llm_embedding.weight"][0][None], St["llm.model.model.embed_tokens.weight"][text.to(dev)],
                       St["llm_embedding.weight"][1][None], St["speech_embedding.weight"][ref_speech.to(dev)]], 0)
    ref = LR.LMRef(cfg, LR.from_cosyvoice2_state_dict(cfg, S), page_size=16, max_pages=64)
    q = LR.LMRequest()
    n = feats.shape[0]
    ref.sample(ref.prefill(q, np.zeros(n, np.int32), np.ones(n, np.uint8), vr.from_torch(feats)), [q])
    for _ in range(10):
        ref.sample(ref.decode([q]), [q])
    got = [int(t[0, 0]) for t in reqs[0].lm_output_tokens]
    assert got == q.tokens[: len(got)], (got, q.tokens)
    m.engine.close()

    # ---- CSM (tiny) ----
    from tests.test_gpu_csm import to_engine_cfg
    from vox_serve_amd.model.csm import CSMModel
    cfg = CR.tiny_csm_cfg()
    W = CR.random_csm_state_dict(cfg, 7, 0.08)
    m = CSMModel("tiny-csm", {k: vr.to_torch(v).to(dev) for k, v in W.items()}, config=to_engine_cfg(cfg), sampling=greedy, **kw)
    reqs, w = _drive_worker(m, [{"prompt_token_ids": [4, 200, 31]}, {"prompt_token_ids": [9, 8, 7, 6, 5]}], 8, async_scheduling=async_scheduling)
    ref = CR.CSMRef(cfg, W, page_size=16, max_pages=64, max_batch=4)
    rr, frames = [], []
    for p in ([4, 200, 31], [9, 8, 7, 6, 5]):
        ids = np.zeros((len(p), cfg.n_codebooks + 1), np.int32)
        masks = np.zeros_like(ids, dtype=np.uint8)
        ids[:, -1], masks[:, -1] = p, 1
        q = QR.RefRequest()
        lg, hid = ref.prefill(q, ids, masks)
        ref.frame([q], lg, hid)
        rr.append(q)
    for _ in range(8):
        live = [q for q in rr if not (q.frames and q.frames[-1][0] == 0)]
        if live:
            ref.frame(live)
    for q, r in zip(rr, reqs):
        got = [t[0].tolist() for t in r.lm_output_tokens]
        want = [f.tolist() for f in q.frames][: len(got)]
        assert got == want, (got[:2], want[:2])
        assert len(r.lm_output_audio_tokens) == sum(1 for f in got if f[0] != 0)
    assert w.empty_pages.qsize() == 64 - sum(len(r.kv_pages) for r in reqs)
    m.engine.close()


def test_csm_served_end_to_end_with_mimi():
    """Scheduler -> ModelWorker -> CSMModel (native frame engine + native Mimi): audio chunks + completion on the wire."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    from oracle import csm_ref as CR, mimi_ref as MR, voxref as vr
    from tests.test_gpu_csm import to_engine_cfg
    from vox_serve_amd.model.csm import CSMModel
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.tokenizer.mimi import MimiConfig
    from vox_serve_amd.worker import ModelWorker
    cfg = CR.tiny_csm_cfg()
    mc = MR.tiny_mimi_cfg()             # 6 codebooks, like the tiny CSM
    pc = MimiConfig(**{k: getattr(mc, k) for k in MimiConfig.__dataclass_fields__ if hasattr(mc, k)})
    m = CSMModel("tiny-csm", {k: vr.to_torch(v).to(dev) for k, v in CR.random_csm_state_dict(cfg, 7, 0.08).items()},
                 config=to_engine_cfg(cfg), sampling=SamplingConfig(greedy=True), codec_weights=MR.random_mimi_weights(mc, 1),
                 codec_config=pc, device=str(dev), max_batch_size=4, page_size=16, max_num_pages=64, max_seq_len=512,
                 max_prefill_tokens=64)
    t = QueueTransport()
    w = ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=16, device=m.device)
    s = Scheduler(w, max_batch_size=4, transport=t)
    for rid, ids in {"a": [4, 200, 31], "b": [9, 8, 7, 6, 5]}.items():
        t.requests.put(encode_request(rid, "", model_kwargs={"prompt_token_ids": ids}))
    for _ in range(400):                # greedy tiny models rarely emit the stop code: bound the run, then finish by hand
        s._step()
        if all(len(r.lm_output_audio_tokens) >= 25 for r in s.active_requests) or not s.active_requests:
            break
    for r in s.active_requests:
        r.done_lm_generation, r.finish_reason = True, "max_tokens_reached"
    s.run_until_idle(200)
    got = {"a": [0, None], "b": [0, None]}
    while not t.results.empty():
        rid, kind, body = t.results.get().split(b"|", 2)
        if kind == b"AUDIO":
            got[rid.decode()][0] += len(body)
            assert np.abs(np.frombuffer(body, dtype=np.int16)).max() > 50
        else:
            got[rid.decode()][1] = json.loads(body)
    hop = m.audio_decoder.hop
    for rid, (nbytes, done) in got.items():
        assert done and done["status"] == "completed"
        assert nbytes >= 2 * hop * 20 and nbytes % 2 == 0, (rid, nbytes)
    assert w.empty_pages.qsize() == 64
    m.engine.close(); m.audio_decoder.close()


def test_long_prompt_is_prefilled_in_context_chunks():
    """A prompt longer than the engine's row capacity (the reference never schedules it: one 1024-token bucket) is
    prefilled in equal context chunks + a final chunk: same first frame, same K/V, same continuation as the unchunked
    prefill of an engine with room for it (every chunk stays on the 33+ row GEMM path, so per-row arithmetic is identical)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    prompt = [1, 2, 3] + rng.integers(20, 500, 120).tolist() + [7, 8, 9, 10, 11]      # 128 template tokens -> 134 prefill rows
    outs = []
    for cap in (48, 256):
        m, cfg = build(dev, max_tokens=200, max_prefill_tokens=cap)
        reqs, w = _drive_worker(m, [{"prompt_token_ids": prompt, "speaker": "a"}], steps=6)
        r = reqs[0]
        toks = torch.cat([t.reshape(1, -1).cpu() for t in r.lm_output_tokens])
        n = r.kv_token_len
        kv = m.engine.kv[:, r.kv_pages].float().cpu().clone()
        outs.append((toks, n, kv))
        if cap == 48:
            assert m.engine.max_rows == 48 and w.cuda_graph_seq_len_buckets[-1] > 134
        m.engine.close(); m.audio_decoder.close()
    (t0, n0, kv0), (t1, n1, kv1) = outs
    assert n0 == n1 and t0.shape == t1.shape and t0.shape[0] == 7
    assert torch.equal(t0, t1)
    assert torch.equal(kv0, kv1)


def test_csm_stateful_mimi_option_streams_without_seams():
    """CSMModel(stateful_codec=True): every request keeps its Mimi state in a slot (allocated by preprocess, released with the
    request), so the served PCM equals ONE Mimi decode of the request's whole code sequence (the reference's stateless
    chunks differ from that at every chunk border)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    from oracle import csm_ref as CR, mimi_ref as MR, voxref as vr
    from tests.test_gpu_csm import to_engine_cfg
    from vox_serve_amd.model.csm import CSMModel
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.tokenizer.mimi import MimiConfig
    from vox_serve_amd.worker import ModelWorker
    cfg, mc = CR.tiny_csm_cfg(), MR.tiny_mimi_cfg()
    pc = MimiConfig(**{k: getattr(mc, k) for k in MimiConfig.__dataclass_fields__ if hasattr(mc, k)})
    Wm = MR.random_mimi_weights(mc, 1)
    pcm = {}
    for stateful in (True, False):
        m = CSMModel("tiny-csm", {k: vr.to_torch(v).to(dev) for k, v in CR.random_csm_state_dict(cfg, 7, 0.08).items()},
                     config=to_engine_cfg(cfg), sampling=SamplingConfig(greedy=True), codec_weights=Wm, codec_config=pc, device=str(dev),
                     max_batch_size=4, page_size=16, max_num_pages=64, max_seq_len=512, max_prefill_tokens=64, stateful_codec=stateful)
        t = QueueTransport()
        w = ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=16, device=m.device)
        s = Scheduler(w, max_batch_size=4, transport=t)
        for rid, ids in {"a": [4, 200, 31], "b": [9, 8, 7, 6, 5]}.items():
            t.requests.put(encode_request(rid, "", model_kwargs={"prompt_token_ids": ids}))
        reqs = {}
        for _ in range(400):
            s._step()
            reqs.update({r.request_id: r for r in s.active_requests})
            if all(len(r.lm_output_audio_tokens) >= 30 for r in s.active_requests) or not s.active_requests:
                break
        for r in s.active_requests:
            r.done_lm_generation, r.finish_reason = True, "max_tokens_reached"
        s.run_until_idle(200)
        out = {"a": b"", "b": b""}
        while not t.results.empty():
            rid, kind, body = t.results.get().split(b"|", 2)
            if kind == b"AUDIO":
                out[rid.decode()] += body
        pcm[stateful] = out
        if stateful:
            hop, interval = m.audio_decoder.hop, m.detokenize_interval
            for rid, r in reqs.items():
                codes = torch.cat(r.lm_output_audio_tokens, 0)[:, :mc.n_q].T[None].long()           # [1, Q, T]
                n_full = (codes.shape[2] // interval) * interval                                    # whole chunks: no padding / trim
                whole = MR.MimiRef(mc, Wm).decode(codes[:, :, :n_full]).numpy()[0, 0]
                got = np.frombuffer(out[rid], dtype=np.int16)[: n_full * hop].astype(np.int32)
                want = (whole * 32767).astype(np.int16).astype(np.int32)
                assert got.shape == want.shape and n_full >= 2 * interval
                assert np.abs(got - want).max() <= 2, rid                                           # int16 truncation of 1e-5-close floats
            assert len(m.audio_decoder._free_slots) == m.audio_decoder.max_slots                     # slots returned
        m.engine.close(); m.audio_decoder.close()
    # same tokens either way (the codec does not feed back), different audio after the first chunk border
    a_s, a_l = (np.frombuffer(pcm[k]["a"], dtype=np.int16) for k in (True, False))
    n = min(len(a_s), len(a_l))
    assert np.array_equal(a_s[:100], a_l[:100]) and not np.array_equal(a_s[:n], a_l[:n])


def test_orpheus_served_end_to_end_with_snac_windows():
    """Orpheus plugin (single-stack LM engine + SNAC decoder) through Scheduler -> ModelWorker: 28-token detokenizer windows
    advancing by 7 (detokenize_overlap 21), 2048 samples per window, EOS dropped from the audio tokens, resources released,
    byte-identical audio on a repeat (seeded device noise)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    from oracle import snac_ref as SR
    from vox_serve_amd.model.orpheus import OrpheusConfig, OrpheusModel
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.synth import synth_orpheus_weights
    from vox_serve_amd.tokenizer.snac import SNACConfig
    from vox_serve_amd.worker import ModelWorker
    oc = OrpheusConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                       head_dim=64, vocab_size=10 + 7 * 64, audio_token_base=10, stop_token_id=2)
    sc = SR.tiny_snac_cfg()
    pc = SNACConfig(latent_dim=sc.latent_dim, decoder_dim=sc.decoder_dim, decoder_rates=list(sc.rates), codebook_size=sc.codebook_size,
                    codebook_dim=sc.codebook_dim, vq_strides=list(sc.vq_strides))

    def serve():
        m = OrpheusModel("tiny-orpheus", synth_orpheus_weights(oc, dev, seed=3, std=0.08), SR.random_snac_weights(sc, 1), config=oc,
                         codec_config=pc, device=str(dev), max_batch_size=4, page_size=16, max_num_pages=64, max_seq_len=512,
                         max_prefill_tokens=64, max_pos=512, noise_seed=5)
        m.default_sampling_config = SamplingConfig(greedy=True, max_tokens=70, repetition_penalty=1.3, repetition_window=-1)
        t = QueueTransport()
        w = ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=16, device=str(dev))
        s = Scheduler(w, max_batch_size=4, transport=t)
        for rid, ids in (("a", [5, 9, 200, 31, 7]), ("b", [6, 8, 300, 12])):
            t.requests.put(encode_request(rid, "", model_kwargs={"prompt_token_ids": ids, "voice": "tara"}))
        s.run_until_idle(3000)
        out = {"a": b"", "b": b""}
        done = {}
        while not t.results.empty():
            rid, kind, body = t.results.get().split(b"|", 2)
            if kind == b"AUDIO":
                out[rid.decode()] += body
            else:
                done[rid.decode()] = json.loads(body)
        free = w.empty_pages.qsize()
        m.engine.close(); m.audio_decoder.close()
        return out, done, free

    out, done, free = serve()
    assert free == 64 and set(done) == {"a", "b"} and all(d["status"] == "completed" for d in done.values())
    for rid, pcm in out.items():
        assert len(pcm) % 2 == 0 and len(pcm) >= 2 * 128            # at least one window (tiny SNAC: hop 32 -> 128 samples per window)
        assert np.abs(np.frombuffer(pcm, np.int16)).max() > 50
    out2, _, _ = serve()
    assert out == out2


def test_async_scheduling_gives_the_same_audio():
    """Scheduler(async_scheduling=True): the worker's run_lm_* return the request-state update as a coroutine and the next
    step is launched before the previous one's tokens reach the host (scheduler/base.py:166-221 of the reference).  EOS /
    max_tokens are seen one step late (a surplus row, dropped): the audio is byte-identical to the synchronous loop's."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    from vox_serve_amd import _native as N
    # The one-step lag changes which rows share a launch around a request's last frame, and a row's bits depend on the row
    # count of its launch once calls of 3+ rows run on the matrix cores (default).  With exact_rows 8 every call of this
    # test (<= 4 rows) uses the row-count-invariant wave64 kernels, so the two schedules must agree byte for byte.
    N.set_exact_rows(8)
    try:
        m, cfg = build(dev, max_tokens=34)
        prompt = [1, 2, 3, 40, 41, 42, 43, 7, 8, 9, 10, 11]
        prompts = {"r1": prompt, "r2": prompt[:3] + [50, 51] + prompt[-5:], "r3": prompt[:3] + [60] + prompt[-5:]}
        sync, w1 = serve(m, prompts)
        asy, w2 = serve(m, prompts, async_scheduling=True)
        assert w2.async_scheduling is False and w2._pending is None            # loop left the worker drained and synchronous
        for rid in prompts:
            assert asy[rid]["done"] == sync[rid]["done"], rid
            assert asy[rid]["pcm"] == sync[rid]["pcm"] and len(asy[rid]["pcm"]) > 0, rid
        assert w2.empty_pages.qsize() == 64
        m.engine.close(); m.audio_decoder.close()
    finally:
        N.set_exact_rows(2)


def test_detokenize_beside_the_lm_frame_gives_the_same_audio():
    """The scheduler enqueues the codec chunk on the worker's detokenize stream, launches the LM step behind it and collects the
    audio afterwards (first chunks are still sent at once); the reference's order (decode, send, then the LM step) and the
    opt-in batching window (`detokenize_min_batch`) must give every request the same bytes — the streaming codec state and
    the token windows do not depend on when the host waits or on which requests share a codec call."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    m, cfg = build(dev, max_tokens=40)
    prompt = [1, 2, 3, 40, 41, 42, 43, 7, 8, 9, 10, 11]
    prompts = {"r1": prompt, "r2": prompt[:3] + [50, 51] + prompt[-5:], "r3": prompt[:3] + [60] + prompt[-5:],
               "r4": prompt[:3] + [70, 71, 72] + prompt[-5:]}
    ref, w0 = serve(m, prompts, overlap_detokenize=False)
    for kw in ({}, {"detokenize_min_batch": 3}, {"detokenize_min_batch": 3, "async_scheduling": False, "overlap_detokenize": False}):
        got, w = serve(m, prompts, **kw)
        for rid in prompts:
            assert got[rid]["done"] == ref[rid]["done"], (kw, rid)
            assert got[rid]["pcm"] == ref[rid]["pcm"] and len(got[rid]["pcm"]) > 0, (kw, rid)
        assert w.empty_pages.qsize() == 64
    m.engine.close(); m.audio_decoder.close()


@pytest.mark.parametrize("evolving", [False, True])
def test_cosyvoice2_served_end_to_end_with_flow_and_hift(evolving):
    """Scheduler -> ModelWorker -> CosyVoice2Model (native LM engine) -> CosyVoice2Decoder (native flow + HiFT): 28-token windows
    overlapping by 3 become 24000-sample AUDIO messages; greedy decoding and the seeded noise streams make two runs byte-identical.
    evolving=False: the plugin's default shared prompt cache; True: use_detokenizer_cache=True — every request owns a detokenizer
    cache slot that its chunks advance (two requests of different lengths: their caches are in different states in the same call),
    the slots return to the pool on completion, and the audio differs from the shared-prompt mode's after the first window."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    from oracle import flow_ref as FR, hift_ref as HR, lm_ref as LR, voxref as vr
    from tests.test_gpu_flow import flow_plugin_cfg
    from tests.test_gpu_hift import to_plugin_cfg
    from vox_serve_amd.model.cosyvoice2 import CosyVoice2Config, CosyVoice2Model
    from vox_serve_amd.sampling import SamplingConfig
    from vox_serve_amd.scheduler import QueueTransport, Scheduler, encode_request
    from vox_serve_amd.worker import ModelWorker
    cfg = LR.tiny_cosyvoice2_cfg()
    c = cfg.stack
    St = {k: vr.to_torch(v).to(dev) for k, v in LR.random_cosyvoice2_state_dict(cfg, seed=4, std=0.08).items()}
    pc = CosyVoice2Config(llm_input_size=c.hidden, llm_output_size=c.hidden, speech_token_size=cfg.vocab_out - 3, hidden_size=c.hidden,
                          intermediate_size=c.ffn, num_attention_heads=c.heads, num_key_value_heads=c.kv_heads, num_hidden_layers=c.layers)
    fc, hc = FR.tiny_flow_cfg(), HR.HiftCfg(base_channels=256, f0_channels=64)
    fc.vocab = cfg.vocab_out - 3
    g = torch.Generator().manual_seed(2)
    Np = 8
    ref = {"ref_text_ids": torch.tensor([3, 9]), "prompt_speech_token": torch.randint(0, fc.vocab, (1, Np), generator=g),
           "prompt_feat": (0.7 * torch.randn(1, 2 * Np, fc.mel, generator=g)), "embedding": torch.randn(1, fc.spk_dim, generator=g)}

    def serve():
        m = CosyVoice2Model("tiny-cosy", St, config=pc, sampling=SamplingConfig(greedy=True, max_tokens=75), max_pos=512, speaker_ref=ref,
                            codec_weights={"flow": FR.random_flow_weights(fc, seed=3), "hift": HR.random_hift_weights(hc, seed=2)},
                            codec_config={"flow": flow_plugin_cfg(fc), "hift": to_plugin_cfg(hc)}, codec_seed=9, device=str(dev),
                            max_batch_size=4, page_size=16, max_num_pages=64, max_seq_len=512, max_prefill_tokens=64,
                            use_detokenizer_cache=evolving)
        t = QueueTransport()
        w = ModelWorker(model=m, max_batch_size=4, max_num_pages=64, page_size=16, device=str(dev))
        s = Scheduler(w, max_batch_size=4, transport=t)
        for rid, ids in (("a", [11, 12]), ("b", [13, 14, 15])):
            t.requests.put(encode_request(rid, "", model_kwargs={"prompt_token_ids": ids}))
        toks = {}
        for _ in range(400):
            s._step()
            for r in s.active_requests:
                toks[r.request_id] = [int(x[0, 0]) for x in r.lm_output_audio_tokens]
            if not s.active_requests and t.requests.empty():
                break
        out, done = {"a": [], "b": []}, {}
        while not t.results.empty():
            rid, kind, body = t.results.get().split(b"|", 2)
            if kind == b"AUDIO":
                out[rid.decode()].append(body)
            else:
                done[rid.decode()] = json.loads(body)
        free = w.empty_pages.qsize()
        return m, out, done, free, toks

    m, out, done, free, toks = serve()
    assert free == 64 and set(done) == {"a", "b"} and all(d["status"] == "completed" for d in done.values())
    for rid, chunks in out.items():
        assert len(chunks) >= 2 and len(chunks[0]) == 2 * 24000           # full windows are 24000 samples of PCM16
        assert np.abs(np.frombuffer(chunks[0], np.int16)).max() > 500
    # the second window of request a == a direct decode of its tokens 25..52 under the noise streams that call used is not reproducible from
    # here (the streams advance per call), so check the deterministic part instead: a second service run gives the same bytes
    if evolving:
        assert sorted(m.audio_decoder._free_slots) == list(range(m.audio_decoder.max_slots))      # every request's slot came back
    m.engine.close(); m.audio_decoder.close()
    m2, out2, done2, _, toks2 = serve()
    assert toks2 == toks and out2 == out
    m2.engine.close(); m2.audio_decoder.close()
    _COSY_SERVED[evolving] = out
    if len(_COSY_SERVED) == 2:       # same tokens (the LM does not depend on the detokenizer mode); the first window starts from the
        a, b = _COSY_SERVED[False]["a"], _COSY_SERVED[True]["a"]      # same prompt caches, later windows see the request's own history
        assert len(a) == len(b) and a[1] != b[1]


_COSY_SERVED = {}


def test_detokenizer_on_its_own_device_gives_the_same_audio():
    """`audio_decoder_device` (worker/base.py:55-78, 641-644 of the reference: the LM on one GPU, the detokenizer on another): the
    codec is created, reset and run under its own device's libvoxhip context and stream; the served PCM equals the default
    placement's byte for byte.  With two GPUs visible the codec really sits on cuda:1; on a one-GPU box the same code path runs with
    an explicit cuda:0 (the per-device context table, the device guard of every tokenizer call, the worker's stream placement)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from vox_serve_amd import _native as N
    dev = torch.device("cuda:0")
    dec_dev = "cuda:1" if torch.cuda.device_count() >= 2 else "cuda:0"
    prompt = [1, 2, 3, 40, 41, 42, 43, 7, 8, 9, 10, 11]
    m, _ = build(dev, max_tokens=30)
    want, _ = serve(m, {"r1": prompt, "r2": prompt[:3] + [50, 51] + prompt[-5:]})
    m.engine.close(); m.audio_decoder.close()
    m2, _ = build(dev, max_tokens=30, audio_decoder_device=dec_dev)
    assert str(m2.audio_decoder.device) == dec_dev and torch.device(dec_dev).index in N._ctxs
    got, w = serve(m2, {"r1": prompt, "r2": prompt[:3] + [50, 51] + prompt[-5:]})
    assert str(w.detokenizer_device) == dec_dev and w._detok_stream.device == torch.device(dec_dev)
    assert all(got[r]["pcm"] == want[r]["pcm"] and got[r]["done"] == want[r]["done"] for r in want)
    assert torch.cuda.current_device() == 0                     # the guards restore the caller's device
    m2.engine.close(); m2.audio_decoder.close()
