"""Model registry (drop-in for /root/reference/vox_serve/model/__init__.py:17-179): MODEL_REGISTRY, register_model,
load_model.  Checkpoints come from a local directory of safetensors (no network on the serving box); sampling
overrides replace the model defaults field by field like the reference (model/__init__.py:132-156)."""
import glob
import os
from typing import Callable, Dict

from ..sampling import SamplingConfig
from .base import BaseLM, BaseLMWithDepth, PreprocessOutput  # noqa: F401

MODEL_REGISTRY: Dict[str, Callable] = {}


def register_model(pattern, model_class=None, *more):
    """`register_model(pattern, model_class)` — the reference's call (model/__init__.py:161-169; the pattern is stored
    lower-cased).  Also usable as a decorator over several patterns (`@register_model("a", "b")`), which is how the
    built-in families below register their loader functions."""
    if model_class is not None and not isinstance(model_class, str):
        MODEL_REGISTRY[pattern.lower()] = model_class
        return None
    names = (pattern,) + ((model_class,) if model_class is not None else ()) + more

    def deco(fn):
        for n in names:
            MODEL_REGISTRY[n.lower()] = fn
        return fn
    return deco


def get_model_class(model_name: str):
    """Exact (case-insensitive) match first, then substring match of a registered pattern (model/__init__.py:44-70)."""
    low = model_name.lower()
    if low in MODEL_REGISTRY:
        return MODEL_REGISTRY[low]
    for pattern, cls in MODEL_REGISTRY.items():
        if pattern in low:
            return cls
    raise ValueError(f"No model class found for '{model_name}'. Available model patterns: {list(MODEL_REGISTRY.keys())}")


def list_supported_models() -> Dict[str, Callable]:
    return MODEL_REGISTRY.copy()


def _load_safetensors_dir(path, device):
    from safetensors.torch import load_file
    sd = {}
    for f in sorted(glob.glob(os.path.join(path, "*.safetensors"))):
        sd.update(load_file(f, device=str(device)))
    return sd


@register_model("qwen3-tts", "Qwen/Qwen3-TTS-12Hz-1.7B-CustomVoice", "qwen3-tts-voice-design",
                "Qwen/Qwen3-TTS-12Hz-1.7B-VoiceDesign", "qwen3-tts-base", "Qwen/Qwen3-TTS-12Hz-1.7B-Base")
def _qwen3(model_name, device="cuda:0", weights=None, codec_weights=None, checkpoint_dir=None, codec_checkpoint_dir=None,
           synthetic=False, **kw):
    from .qwen3_tts import Qwen3TTSModel
    if weights is None:
        if synthetic or checkpoint_dir is None:
            if not synthetic:
                raise FileNotFoundError("no checkpoint_dir given (offline box): pass checkpoint_dir=... or synthetic=True")
            from ..engine import Qwen3Cfg
            from ..synth import synth_qwen3_codec_weights, synth_qwen3_weights
            weights, codec_weights = synth_qwen3_weights(Qwen3Cfg(), device), synth_qwen3_codec_weights()
        else:
            weights = _load_safetensors_dir(checkpoint_dir, device)
            codec_weights = {k.removeprefix("decoder."): v for k, v in
                             _load_safetensors_dir(codec_checkpoint_dir or checkpoint_dir, "cpu").items()}
    name = model_name.lower()
    mtype = "voice_design" if "design" in name else ("base" if name.endswith("base") else "custom_voice")
    if mtype == "base" and "speaker_encoder_weights" not in kw:
        # voice cloning: the speaker encoder ships inside the LM checkpoint (speaker_encoder.*), the codec encoder inside the speech
        # tokenizer's (encoder.*)
        if synthetic or checkpoint_dir is None:
            from ..synth import synth_qwen3_codec_encoder_weights, synth_qwen3_speaker_encoder_weights
            kw["speaker_encoder_weights"] = synth_qwen3_speaker_encoder_weights()
            kw["audio_encoder_weights"] = synth_qwen3_codec_encoder_weights()
        else:
            spk = {k.removeprefix("speaker_encoder."): v for k, v in weights.items() if k.startswith("speaker_encoder.")}
            enc = {k.removeprefix("encoder."): v for k, v in
                   _load_safetensors_dir(codec_checkpoint_dir or checkpoint_dir, "cpu").items() if k.startswith("encoder.")}
            kw["speaker_encoder_weights"], kw["audio_encoder_weights"] = spk or None, enc or None
    return Qwen3TTSModel(model_name, weights, codec_weights, device=device, tts_model_type=mtype, **kw)


def _single_stack(model_name, cls_path, cfg_path, synth_name, device, weights, checkpoint_dir, synthetic, kw):
    import importlib
    mod = importlib.import_module(cls_path[0], __package__)
    cls, cfg_cls = getattr(mod, cls_path[1]), getattr(mod, cfg_path)
    config = kw.pop("config", None) or cfg_cls()
    if weights is None:
        if checkpoint_dir is not None:
            weights = _load_safetensors_dir(checkpoint_dir, device)
        elif synthetic:
            from .. import synth
            weights = getattr(synth, synth_name)(config, device)
        else:
            raise FileNotFoundError("no checkpoint_dir given (offline box): pass checkpoint_dir=... or synthetic=True")
    kw.pop("detokenize_interval", None)
    return cls(model_name, weights, config=config, device=device, **kw)


@register_model("glm", "zai-org/glm-4-voice-9b")
def _glm(model_name, device="cuda:0", weights=None, checkpoint_dir=None, synthetic=False, **kw):
    if synthetic and kw.get("codec_weights") is None:
        from .. import synth
        kw["codec_weights"] = synth.synth_glm_codec_weights()        # random-init detokenizer (flow + HiFT), like the LM weights
    return _single_stack(model_name, (".glm_voice", "GLMVoiceModel"), "GLMVoiceConfig", "synth_glm_weights", device,
                         weights, checkpoint_dir, synthetic, kw)


@register_model("cosyvoice2", "FunAudioLLM/CosyVoice2-0.5B")
def _cosyvoice2(model_name, device="cuda:0", weights=None, checkpoint_dir=None, synthetic=False, **kw):
    if synthetic and kw.get("codec_weights") is None and "speaker_ref" not in kw:
        # random-init detokenizer (flow + HiFT) with a synthetic speaker prompt, like the LM weights
        from .. import synth
        cw, prompt = synth.synth_cosyvoice2_codec_weights()
        kw["codec_weights"] = cw
        kw["speaker_ref"] = dict(prompt, ref_text_ids=__import__("torch").zeros(0, dtype=__import__("torch").long))
    return _single_stack(model_name, (".cosyvoice2", "CosyVoice2Model"), "CosyVoice2Config", "synth_cosyvoice2_weights",
                         device, weights, checkpoint_dir, synthetic, kw)


@register_model("csm", "sesame/csm-1b")
def _csm(model_name, device="cuda:0", weights=None, checkpoint_dir=None, codec_checkpoint_dir=None, synthetic=False, **kw):
    from ..engine import CSMCfg
    from .csm import CSMModel
    config = kw.pop("config", None) or CSMCfg()
    if weights is None:
        if checkpoint_dir is not None:
            weights = _load_safetensors_dir(checkpoint_dir, device)
        elif synthetic:
            from ..synth import synth_csm_weights
            weights = synth_csm_weights(config, device)
        else:
            raise FileNotFoundError("no checkpoint_dir given (offline box): pass checkpoint_dir=... or synthetic=True")
    kw.pop("detokenize_interval", None)
    if codec_checkpoint_dir is not None and "codec_weights" not in kw:      # kyutai/moshiko-pytorch-bf16 tokenizer checkpoint
        kw["codec_weights"] = _load_safetensors_dir(codec_checkpoint_dir, "cpu")
    return CSMModel(model_name, weights, config=config, device=device, **kw)


@register_model("orpheus", "canopylabs/orpheus-3b-0.1-ft")
def _orpheus(model_name, device="cuda:0", weights=None, codec_weights=None, checkpoint_dir=None, codec_checkpoint_dir=None,
             synthetic=False, **kw):
    from .orpheus import OrpheusConfig, OrpheusModel
    config = kw.pop("config", None) or OrpheusConfig()
    kw.pop("detokenize_interval", None)
    if weights is None:
        if checkpoint_dir is not None:
            weights = _load_safetensors_dir(checkpoint_dir, device)
            if codec_weights is None:                                      # hubertsiuzdak/snac_24khz checkpoint directory
                codec_weights = _load_safetensors_dir(codec_checkpoint_dir or checkpoint_dir, "cpu")
        elif synthetic:
            from ..synth import synth_orpheus_weights, synth_snac_weights
            weights, codec_weights = synth_orpheus_weights(config, device), synth_snac_weights()
        else:
            raise FileNotFoundError("no checkpoint_dir given (offline box): pass checkpoint_dir=... or synthetic=True")
    return OrpheusModel(model_name, weights, codec_weights, config=config, device=device, **kw)


def load_model(model_name: str, device: str = "cuda", top_p=None, top_k=None, min_p=None, temperature=None,
               max_tokens=None, repetition_penalty=None, repetition_window=None, cfg_scale=None, greedy=False,
               enable_torch_compile=False, audio_decoder_device=None, detokenize_interval=None, **kw):
    loader = get_model_class(model_name)
    if detokenize_interval is not None and loader is not _qwen3:      # model/__init__.py:124-126
        raise ValueError(f"Detokenize interval is only supported for Qwen3TTS models, got {model_name}")
    overrides = dict(top_p=top_p, top_k=top_k, min_p=min_p, temperature=temperature, max_tokens=max_tokens,
                     repetition_penalty=repetition_penalty, repetition_window=repetition_window, cfg_scale=cfg_scale)
    has_overrides = greedy or any(v is not None for v in overrides.values())

    def merged(cur):      # per-field override (model/__init__.py:132-156)
        return SamplingConfig(greedy=greedy, **{k: (v if v is not None else getattr(cur, k)) for k, v in overrides.items()})
    if has_overrides and loader in (_glm, _cosyvoice2, _orpheus):
        # these plugins size the engine's persisted repetition cache from the sampling config: hand them the merged config
        # BEFORE the engine exists (the reference swaps the config after construction; its cache is per-request)
        kw["sampling_overrides"] = merged
    m = loader(model_name, device=device, audio_decoder_device=audio_decoder_device, detokenize_interval=detokenize_interval, **kw)
    if has_overrides:
        m.default_sampling_config = merged(m.default_sampling_config)
    return m
