"""Synthetic (random-init) weights of the named architectures, generated on the device.  There is no network
for checkpoints on the bench box: SURVEY §8d recipe — N(0, 0.02^2) bf16, norm weights 1, seed 0."""
import torch


def synth_qwen3_weights(cfg, device, seed=0, std=0.02):
    g = torch.Generator(device=device).manual_seed(seed)

    def w(*shape):
        return (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(torch.bfloat16)

    def ones(n):
        return torch.ones(n, device=device, dtype=torch.bfloat16)

    W = {}

    def stack(prefix, c):
        for i in range(c.layers):
            p = f"{prefix}.layers.{i}."
            W[p + "self_attn.q_proj.weight"] = w(c.heads * c.head_dim, c.hidden)
            W[p + "self_attn.k_proj.weight"] = w(c.kv_heads * c.head_dim, c.hidden)
            W[p + "self_attn.v_proj.weight"] = w(c.kv_heads * c.head_dim, c.hidden)
            W[p + "self_attn.o_proj.weight"] = w(c.hidden, c.heads * c.head_dim)
            W[p + "self_attn.q_norm.weight"] = ones(c.head_dim)
            W[p + "self_attn.k_norm.weight"] = ones(c.head_dim)
            W[p + "mlp.gate_proj.weight"] = w(c.ffn, c.hidden)
            W[p + "mlp.up_proj.weight"] = w(c.ffn, c.hidden)
            W[p + "mlp.down_proj.weight"] = w(c.hidden, c.ffn)
            W[p + "input_layernorm.weight"] = ones(c.hidden)
            W[p + "post_attention_layernorm.weight"] = ones(c.hidden)
        W[prefix + ".norm.weight"] = ones(c.hidden)

    H = cfg.talker.hidden
    stack("talker.model", cfg.talker)
    W["talker.model.codec_embedding.weight"] = w(cfg.vocab, H)
    W["talker.model.text_embedding.weight"] = w(cfg.text_vocab, cfg.text_hidden)
    W["talker.text_projection.linear_fc1.weight"] = w(cfg.text_hidden, cfg.text_hidden)
    W["talker.text_projection.linear_fc1.bias"] = w(cfg.text_hidden)
    W["talker.text_projection.linear_fc2.weight"] = w(H, cfg.text_hidden)
    W["talker.text_projection.linear_fc2.bias"] = w(H)
    W["talker.codec_head.weight"] = w(cfg.vocab, H)
    stack("talker.code_predictor.model", cfg.depth)
    for j in range(cfg.n_groups - 1):
        W[f"talker.code_predictor.model.codec_embedding.{j}.weight"] = w(cfg.depth_vocab, H)
        W[f"talker.code_predictor.lm_head.{j}.weight"] = w(cfg.depth_vocab, cfg.depth.hidden)
    W["talker.code_predictor.small_to_mtp_projection.weight"] = w(cfg.depth.hidden, H)
    W["talker.code_predictor.small_to_mtp_projection.bias"] = w(cfg.depth.hidden)
    return W


def synth_qwen3_codec_weights(cfg=None, seed=0):
    """Random-init Qwen3 12 Hz codec decoder weights (CPU fp32 tensors holding bf16-representable values), scaled so
    that the waveform stays O(0.1) through the 12 stacked residual units."""
    import math
    from .tokenizer.qwen3_codec import Qwen3CodecConfig, param_shapes
    cfg = cfg or Qwen3CodecConfig()
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, s in param_shapes(cfg).items():
        if k.endswith("cluster_usage"):
            t = 1.0 + torch.rand(s, generator=g)
        elif k.endswith("embedding_sum"):
            t = torch.randn(s, generator=g)
        elif k.endswith(("layernorm.weight", "norm.weight")):
            t = 1.0 + 0.1 * torch.randn(s, generator=g)
        elif k.endswith((".alpha", ".beta")):
            t = 0.3 * torch.randn(s, generator=g)
        elif k.endswith("layer_scale.scale"):
            t = torch.full(s, 0.05)
        elif k.endswith(".gamma"):
            t = torch.full(s, 0.1)
        elif k.endswith(".bias"):
            t = 0.02 * torch.randn(s, generator=g)
        else:
            fan_in = s[1] * (s[2] if len(s) == 3 else 1)
            if "block.1.conv.weight" in k:
                fan_in = s[0] * 2
            elif "upsample" in k and k.endswith("0.conv.weight"):
                fan_in = s[0]
            elif k.endswith("dwconv.conv.weight"):
                fan_in = 7
            t = torch.randn(s, generator=g) / math.sqrt(fan_in)
            if k.endswith("conv2.conv.weight"):
                t = t * 0.25
            if s[0] == 1:
                t = t * 0.1
        W[k] = t.to(torch.bfloat16).float()
    return W


def _gen(device, seed, std):
    g = torch.Generator(device=device).manual_seed(seed)
    w = lambda *shape: (torch.randn(*shape, generator=g, device=device, dtype=torch.float32) * std).to(torch.bfloat16)
    ones = lambda n: torch.ones(n, device=device, dtype=torch.bfloat16)
    return w, ones


def synth_glm_weights(c, device, seed=0, std=0.02):
    """Random-init GLM-4-Voice state_dict (reference names, glm_voice.py:85-305); c: model.glm_voice.GLMVoiceConfig."""
    w, ones = _gen(device, seed, std)
    d = c.hidden_size // c.num_attention_heads
    qkv = c.hidden_size + 2 * d * c.multi_query_group_num
    W = {"transformer.embedding.word_embeddings.weight": w(c.padded_vocab_size, c.hidden_size),
         "transformer.encoder.final_layernorm.weight": ones(c.hidden_size),
         "transformer.output_layer.weight": w(c.padded_vocab_size, c.hidden_size)}
    for i in range(c.num_layers):
        p = f"transformer.encoder.layers.{i}."
        W[p + "self_attention.query_key_value.weight"] = w(qkv, c.hidden_size)
        W[p + "self_attention.query_key_value.bias"] = w(qkv)
        W[p + "self_attention.dense.weight"] = w(c.hidden_size, c.hidden_size)
        W[p + "mlp.dense_h_to_4h.weight"] = w(2 * c.ffn_hidden_size, c.hidden_size)
        W[p + "mlp.dense_4h_to_h.weight"] = w(c.hidden_size, c.ffn_hidden_size)
        W[p + "input_layernorm.weight"] = ones(c.hidden_size)
        W[p + "post_attention_layernorm.weight"] = ones(c.hidden_size)
    return W


def synth_cosyvoice2_weights(c, device, seed=0, std=0.02):
    """Random-init CosyVoice2 LLM state_dict (reference names, cosyvoice2.py:106-316); c: CosyVoice2Config."""
    w, ones = _gen(device, seed, std)
    H, d = c.hidden_size, c.hidden_size // c.num_attention_heads
    V = c.speech_token_size + 3
    W = {"llm.model.model.embed_tokens.weight": w(c.vocab_size, H), "llm.model.model.norm.weight": ones(H),
         "llm_embedding.weight": w(2, H), "llm_decoder.weight": w(V, H), "llm_decoder.bias": w(V),
         "speech_embedding.weight": w(V, H)}
    for i in range(c.num_hidden_layers):
        p = f"llm.model.model.layers.{i}."
        for n, rows in (("q", c.num_attention_heads), ("k", c.num_key_value_heads), ("v", c.num_key_value_heads)):
            W[p + f"self_attn.{n}_proj.weight"] = w(rows * d, H)
            W[p + f"self_attn.{n}_proj.bias"] = w(rows * d)
        W[p + "self_attn.o_proj.weight"] = w(H, c.num_attention_heads * d)
        W[p + "mlp.gate_proj.weight"] = w(c.intermediate_size, H)
        W[p + "mlp.up_proj.weight"] = w(c.intermediate_size, H)
        W[p + "mlp.down_proj.weight"] = w(H, c.intermediate_size)
        W[p + "input_layernorm.weight"] = ones(H)
        W[p + "post_attention_layernorm.weight"] = ones(H)
    return W


def synth_csm_weights(c, device, seed=0, std=0.02):
    """Random-init CSM state_dict (reference names, csm.py:55-313); c: engine.CSMCfg."""
    w, ones = _gen(device, seed, std)
    W = {}

    def stack(prefix, s):
        for i in range(s.layers):
            p = f"{prefix}.layers.{i}."
            W[p + "self_attn.q_proj.weight"] = w(s.heads * s.head_dim, s.hidden)
            W[p + "self_attn.k_proj.weight"] = w(s.kv_heads * s.head_dim, s.hidden)
            W[p + "self_attn.v_proj.weight"] = w(s.kv_heads * s.head_dim, s.hidden)
            W[p + "self_attn.o_proj.weight"] = w(s.hidden, s.heads * s.head_dim)
            W[p + "mlp.gate_proj.weight"] = w(s.ffn, s.hidden)
            W[p + "mlp.up_proj.weight"] = w(s.ffn, s.hidden)
            W[p + "mlp.down_proj.weight"] = w(s.hidden, s.ffn)
            W[p + "input_layernorm.weight"] = ones(s.hidden)
            W[p + "post_attention_layernorm.weight"] = ones(s.hidden)
        W[prefix + ".norm.weight"] = ones(s.hidden)
    b, d = c.backbone, c.depth
    stack("backbone_model", b)
    W["backbone_model.embed_tokens.embed_audio_tokens.weight"] = w(c.n_codebooks * c.vocab, b.hidden)
    W["embed_text_tokens.weight"] = w(c.text_vocab, b.hidden)
    W["lm_head.weight"] = w(c.vocab, b.hidden)
    stack("depth_decoder.model", d)
    W["depth_decoder.model.inputs_embeds_projector.weight"] = w(d.hidden, b.hidden)
    W["depth_decoder.codebooks_head.weight"] = w(c.n_codebooks - 1, d.hidden, c.vocab)
    return W


def synth_mimi_weights(cfg=None, seed=0):
    """Random-init Mimi decoder weights (CPU fp32 tensors holding bf16-representable values): fan-in scaled so that the
    activations stay O(1) through the SEANet stack; the last conv is scaled down so the waveform is O(0.1)."""
    import math
    from .tokenizer.mimi import MimiConfig, param_shapes
    cfg = cfg or MimiConfig()
    g = torch.Generator().manual_seed(seed)
    W = {}
    shapes = param_shapes(cfg)
    last = max((k for k in shapes if k.startswith("decoder.model.") and k.endswith(".conv.conv.weight")), key=lambda k: int(k.split(".")[2]))
    for k, s in shapes.items():
        if k.endswith("cluster_usage"):
            t = torch.rand(s, generator=g) * 3 + 0.5
        elif k.endswith("embedding_sum"):
            t = torch.randn(s, generator=g) * 2.0
        elif k.endswith(".scale"):
            t = torch.full(s, 0.1)
        elif "norm" in k and k.endswith("weight"):
            t = 1 + 0.1 * torch.randn(s, generator=g)
        elif k.endswith("bias"):
            t = 0.02 * torch.randn(s, generator=g)
        elif "convtr" in k and k.startswith("decoder"):
            t = torch.randn(s, generator=g) * math.sqrt(2.0 / (s[0] * 2))
        elif k.startswith("upsample"):
            t = 0.7 + 0.2 * torch.randn(s, generator=g)
        else:
            t = torch.randn(s, generator=g) * math.sqrt(1.5 / math.prod(s[1:]))
        if k in (last, last.replace("weight", "bias")):
            t = t * 0.01
        W[k] = t.to(torch.bfloat16).to(torch.float32)
    return W


def synth_orpheus_weights(c, device, seed=0, std=0.02):
    """Random-init Orpheus / Llama-3.2 state_dict (reference names, orpheus.py:36-200; tied lm_head omitted); c: OrpheusConfig."""
    w, ones = _gen(device, seed, std)
    H, d = c.hidden_size, c.head_dim
    W = {"model.embed_tokens.weight": w(c.vocab_size, H), "model.norm.weight": ones(H)}
    for i in range(c.num_hidden_layers):
        p = f"model.layers.{i}."
        for n, rows in (("q", c.num_attention_heads), ("k", c.num_key_value_heads), ("v", c.num_key_value_heads)):
            W[p + f"self_attn.{n}_proj.weight"] = w(rows * d, H)
        W[p + "self_attn.o_proj.weight"] = w(H, c.num_attention_heads * d)
        W[p + "mlp.gate_proj.weight"] = w(c.intermediate_size, H)
        W[p + "mlp.up_proj.weight"] = w(c.intermediate_size, H)
        W[p + "mlp.down_proj.weight"] = w(H, c.intermediate_size)
        W[p + "input_layernorm.weight"] = ones(H)
        W[p + "post_attention_layernorm.weight"] = ones(H)
    return W


def synth_snac_weights(cfg=None, seed=0):
    """Random-init SNAC decoder-side state_dict (reference names with weight-norm parametrizations; CPU fp32 tensors), gains
    chosen so that the waveform stays O(0.1) through the 12 residual units."""
    import math
    from .tokenizer.snac import SNACConfig
    c = cfg or SNACConfig()
    g = torch.Generator().manual_seed(seed)
    W = {}

    def wn(name, shape, gain=1.0):
        W[name + ".parametrizations.weight.original0"] = gain * (0.7 + 0.6 * torch.rand(shape[0], 1, 1, generator=g))
        W[name + ".parametrizations.weight.original1"] = torch.randn(shape, generator=g)

    for i in range(len(c.vq_strides)):
        q = f"quantizer.quantizers.{i}."
        W[q + "codebook.weight"] = torch.randn(c.codebook_size, c.codebook_dim, generator=g)
        wn(q + "out_proj", (c.latent_dim, c.codebook_dim, 1))
        W[q + "out_proj.bias"] = 0.02 * torch.randn(c.latent_dim, generator=g)
    d = "decoder.model."
    wn(d + "0", (c.latent_dim, 1, 7))
    W[d + "0.bias"] = 0.02 * torch.randn(c.latent_dim, generator=g)
    wn(d + "1", (c.decoder_dim, c.latent_dim, 1))
    W[d + "1.bias"] = 0.02 * torch.randn(c.decoder_dim, generator=g)
    ch = c.decoder_dim
    for bi, r in enumerate(c.decoder_rates):
        b = f"{d}{2 + bi}.block."
        cin, cout = ch, ch // 2
        W[b + "0.alpha"] = 0.5 + torch.rand(1, cin, 1, generator=g)
        wn(b + "1", (cin, cout, 2 * r))
        W[b + "1.bias"] = 0.02 * torch.randn(cout, generator=g)
        j = 2
        if c.noise:
            wn(b + "2.linear", (cout, cout, 1), 0.3)
            j = 3
        for u in range(3):
            ru = f"{b}{j + u}.block."
            W[ru + "0.alpha"] = 0.5 + torch.rand(1, cout, 1, generator=g)
            wn(ru + "1", (cout, 1, 7))
            W[ru + "1.bias"] = 0.02 * torch.randn(cout, generator=g)
            W[ru + "2.alpha"] = 0.5 + torch.rand(1, cout, 1, generator=g)
            wn(ru + "3", (cout, cout, 1), 0.35)
            W[ru + "3.bias"] = 0.02 * torch.randn(cout, generator=g)
        ch = cout
    n = 2 + len(c.decoder_rates)
    W[f"{d}{n}.alpha"] = 0.5 + torch.rand(1, ch, 1, generator=g)
    wn(f"{d}{n + 1}", (1, ch, 7), 0.12)
    W[f"{d}{n + 1}.bias"] = 0.02 * torch.randn(1, generator=g)
    return W


def synth_cosyvoice2_codec_weights(flow_cfg=None, hift_cfg=None, seed=0):
    """Random-init CosyVoice2 detokenizer weights (flow.pt / hift.pt names; CPU fp32 tensors holding bf16-representable values, fan-in
    scaled) + a synthetic speaker prompt {prompt_speech_token [1, Np], prompt_feat [1, 2 Np, 80], embedding [1, 192]}."""
    import math
    from .tokenizer.cosyvoice_flow import FlowConfig
    from .tokenizer.hifigan import HiFTConfig
    fc, hc = flow_cfg or FlowConfig(), hift_cfg or HiFTConfig()
    g = torch.Generator().manual_seed(seed)
    F, H = {}, {}

    def w(d, name, shape, gain=1.0):
        fan = 1
        for v in shape[1:]:
            fan *= v
        d[name] = (gain * torch.randn(shape, generator=g) / math.sqrt(max(fan, 1))).to(torch.bfloat16).float()

    def lin(d, n, o, i, bias=True, gain=1.0):
        w(d, n + ".weight", (o, i), gain)
        if bias:
            d[n + ".bias"] = (0.05 * torch.randn(o, generator=g)).to(torch.bfloat16).float()

    def ln(d, n, c):
        d[n + ".weight"] = (1.0 + 0.1 * torch.randn(c, generator=g)).to(torch.bfloat16).float()
        d[n + ".bias"] = (0.05 * torch.randn(c, generator=g)).to(torch.bfloat16).float()

    def conv(d, n, o, i, k, gain=1.0):
        w(d, n + ".weight", (o, i, k), gain)
        d[n + ".bias"] = (0.05 * torch.randn(o, generator=g)).to(torch.bfloat16).float()

    D = fc.dim
    F["input_embedding.weight"] = torch.randn(fc.vocab_size, D, generator=g).to(torch.bfloat16).float()
    lin(F, "spk_embed_affine_layer", fc.mel, fc.spk_embed_dim)
    lin(F, "encoder_proj", fc.mel, D)
    for e in ("encoder.embed", "encoder.up_embed"):
        lin(F, e + ".out.0", D, D)
        ln(F, e + ".out.1", D)
    ln(F, "encoder.after_norm", D)
    conv(F, "encoder.pre_lookahead_layer.conv1", D, D, fc.pre_lookahead_len + 1)
    conv(F, "encoder.pre_lookahead_layer.conv2", D, D, 3)
    conv(F, "encoder.up_layer.conv", D, D, 5)
    for grp, nl in (("encoder.encoders", fc.enc_layers), ("encoder.up_encoders", fc.up_layers)):
        for i in range(nl):
            p = f"{grp}.{i}."
            for q in ("linear_q", "linear_k", "linear_v"):
                lin(F, p + "self_attn." + q, D, D)
            lin(F, p + "self_attn.linear_out", D, D, gain=0.5)
            lin(F, p + "self_attn.linear_pos", D, D, bias=False)
            for b in ("pos_bias_u", "pos_bias_v"):
                F[p + "self_attn." + b] = (0.3 * torch.randn(fc.enc_heads, D // fc.enc_heads, generator=g)).to(torch.bfloat16).float()
            lin(F, p + "feed_forward.w_1", fc.enc_ffn, D)
            lin(F, p + "feed_forward.w_2", D, fc.enc_ffn, gain=0.5)
            ln(F, p + "norm_ff", D)
            ln(F, p + "norm_mha", D)
    es, C = "decoder.estimator.", fc.est_channels
    TE, inner = 4 * C, fc.est_heads * fc.est_head_dim
    lin(F, es + "time_mlp.linear_1", TE, 4 * fc.mel)
    lin(F, es + "time_mlp.linear_2", TE, TE)
    groups = [(es + "down_blocks.0.", 4 * fc.mel)] + [(f"{es}mid_blocks.{i}.", C) for i in range(fc.est_mid_blocks)] + [(es + "up_blocks.0.", 2 * C)]
    for gp, cin in groups:
        p = gp + "0."
        lin(F, p + "mlp.1", C, TE)
        conv(F, p + "block1.block.0", C, cin, 3)
        ln(F, p + "block1.block.2", C)
        conv(F, p + "block2.block.0", C, C, 3)
        ln(F, p + "block2.block.2", C)
        conv(F, p + "res_conv", C, cin, 1)
        for j in range(fc.est_blocks):
            p = f"{gp}1.{j}."
            ln(F, p + "norm1", C)
            for q in ("to_q", "to_k", "to_v"):
                lin(F, p + "attn1." + q, inner, C, bias=False)
            lin(F, p + "attn1.to_out.0", C, inner, gain=0.5)
            ln(F, p + "norm3", C)
            lin(F, p + "ff.net.0.proj", 4 * C, C)
            lin(F, p + "ff.net.2", C, 4 * C, gain=0.5)
    for n in ("down_blocks.0.2", "up_blocks.0.2", "final_block.block.0"):
        conv(F, es + n, C, C, 3)
    ln(F, es + "final_block.block.2", C)
    conv(F, es + "final_proj", fc.mel, C, 1)

    # HiFT (weight-norm parametrizations like the checkpoint)
    def wn(name, shape, gain=1.0):
        H[name + ".parametrizations.weight.original0"] = (gain * (0.7 + 0.6 * torch.rand(shape[0], 1, 1, generator=g))).to(torch.bfloat16).float()
        H[name + ".parametrizations.weight.original1"] = torch.randn(shape, generator=g).to(torch.bfloat16).float()

    def wnb(name, shape, gain=1.0, nb=None):
        wn(name, shape, gain)
        H[name + ".bias"] = (0.02 * torch.randn(shape[0] if nb is None else nb, generator=g)).to(torch.bfloat16).float()

    H1, nst, nk = hc.nb_harmonics + 1, len(hc.upsample_rates), len(hc.resblock_kernel_sizes)
    H["m_source.l_linear.weight"] = (0.6 * torch.randn(1, H1, generator=g)).to(torch.bfloat16).float()
    H["m_source.l_linear.bias"] = torch.zeros(1)
    wnb("conv_pre", (hc.base_channels, hc.in_channels, 7))

    def resblock(p, ch, k):
        for j in range(3):
            wnb(f"{p}.convs1.{j}", (ch, ch, k))
            wnb(f"{p}.convs2.{j}", (ch, ch, k), gain=0.25)
            for a in ("activations1", "activations2"):
                H[f"{p}.{a}.{j}.alpha"] = (0.5 + torch.rand(ch, generator=g)).to(torch.bfloat16).float()

    down = [1] + list(hc.upsample_rates[::-1][:-1])
    cum = [int(v) for v in torch.tensor(down).cumprod(0).tolist()][::-1]
    for i, (u, k) in enumerate(zip(hc.upsample_rates, hc.upsample_kernel_sizes)):
        cin, cout = hc.base_channels // 2 ** i, hc.base_channels // 2 ** (i + 1)
        wnb(f"ups.{i}", (cin, cout, k), nb=cout)
        sk = 1 if cum[i] == 1 else 2 * cum[i]
        H[f"source_downs.{i}.weight"] = (torch.randn(cout, hc.istft_n_fft + 2, sk, generator=g) / math.sqrt((hc.istft_n_fft + 2) * sk)).to(torch.bfloat16).float()
        H[f"source_downs.{i}.bias"] = (0.02 * torch.randn(cout, generator=g)).to(torch.bfloat16).float()
        resblock(f"source_resblocks.{i}", cout, hc.source_resblock_kernel_sizes[i])
        for j, k2 in enumerate(hc.resblock_kernel_sizes):
            resblock(f"resblocks.{i * nk + j}", cout, k2)
    wnb("conv_post", (hc.istft_n_fft + 2, hc.base_channels // 2 ** nst, 7), gain=0.35)
    H["conv_post.bias"][: hc.istft_n_fft // 2 + 1] += 1.5
    cin = hc.in_channels
    for li in range(5):
        wnb(f"f0_predictor.condnet.{2 * li}", (hc.f0_channels, cin, 3), gain=1.2)
        cin = hc.f0_channels
    H["f0_predictor.classifier.weight"] = (torch.randn(1, hc.f0_channels, generator=g) * (120.0 / math.sqrt(hc.f0_channels))).to(torch.bfloat16).float()
    H["f0_predictor.classifier.bias"] = torch.full((1,), 60.0)
    Np = 40
    prompt = {"prompt_speech_token": torch.randint(0, fc.vocab_size, (1, Np), generator=g),
              "prompt_feat": (0.7 * torch.randn(1, 2 * Np, fc.mel, generator=g)).to(torch.bfloat16).float(),
              "embedding": torch.randn(1, fc.spk_embed_dim, generator=g).to(torch.bfloat16).float()}
    return {"flow": F, "hift": H}, prompt


def synth_glm_codec_weights(flow_cfg=None, hift_cfg=None, seed=0):
    """Random-init GLM-4-Voice detokenizer weights (flow / hift checkpoint names; CPU fp32 tensors holding bf16-representable values)."""
    import math
    from .tokenizer.glm import GLMFlowConfig, glm_hift_config
    fc, hc = flow_cfg or GLMFlowConfig(), hift_cfg or glm_hift_config()
    g = torch.Generator().manual_seed(seed)
    F = {}

    def w(name, shape, gain=1.0):
        fan = 1
        for v in shape[1:]:
            fan *= v
        F[name] = (gain * torch.randn(shape, generator=g) / math.sqrt(max(fan, 1))).to(torch.bfloat16).float()

    def vec(name, n, mean=0.0, std=0.05):
        F[name] = (mean + std * torch.randn(n, generator=g)).to(torch.bfloat16).float()

    def lin(n, o, i, bias=True, gain=1.0):
        w(n + ".weight", (o, i), gain)
        if bias:
            vec(n + ".bias", o)

    def nrm(n, c):
        vec(n + ".weight", c, 1.0, 0.1)
        vec(n + ".bias", c)

    def conv(n, o, i, k, gain=1.0):
        w(n + ".weight", (o, i, k), gain)
        vec(n + ".bias", o)

    D, C = fc.dim, fc.est_channels
    F["input_embedding.weight"] = torch.randn(fc.vocab_size, D, generator=g).to(torch.bfloat16).float()
    lin("spk_embed_affine_layer", fc.mel, fc.spk_embed_dim)
    lin("encoder_proj", fc.mel, D)
    lin("encoder.embed.out.0", D, D)
    nrm("encoder.embed.out.1", D)
    nrm("encoder.after_norm", D)
    for i in range(fc.enc_layers):
        p = f"encoder.encoders.{i}."
        for q in ("linear_q", "linear_k", "linear_v"):
            lin(p + "self_attn." + q, D, D)
        lin(p + "self_attn.linear_out", D, D, gain=0.5)
        lin(p + "self_attn.linear_pos", D, D, bias=False)
        for b in ("pos_bias_u", "pos_bias_v"):
            F[p + "self_attn." + b] = (0.3 * torch.randn(fc.enc_heads, D // fc.enc_heads, generator=g)).to(torch.bfloat16).float()
        lin(p + "feed_forward.w_1", fc.enc_ffn, D)
        lin(p + "feed_forward.w_2", D, fc.enc_ffn, gain=0.5)
        nrm(p + "norm_ff", D)
        nrm(p + "norm_mha", D)
    for i in range(fc.reg_layers):
        conv(f"length_regulator.model.{3 * i}", fc.mel, fc.mel, 3)
        nrm(f"length_regulator.model.{3 * i + 1}", fc.mel)
    conv(f"length_regulator.model.{3 * fc.reg_layers}", fc.mel, fc.mel, 1)
    es, TE, inner = "decoder.estimator.", 4 * C, fc.est_heads * fc.est_head_dim
    lin(es + "time_mlp.linear_1", TE, 4 * fc.mel)
    lin(es + "time_mlp.linear_2", TE, TE)
    groups = [(es + "down_blocks.0.", 4 * fc.mel), (es + "down_blocks.1.", C)] + [(f"{es}mid_blocks.{i}.", C) for i in range(fc.est_mid_blocks)] + \
             [(es + "up_blocks.0.", 2 * C), (es + "up_blocks.1.", 2 * C)]
    for gp, cin in groups:
        p = gp + "0."
        lin(p + "mlp.1", C, TE)
        conv(p + "block1.block.0", C, cin, 3)
        nrm(p + "block1.block.1", C)
        conv(p + "block2.block.0", C, C, 3)
        nrm(p + "block2.block.1", C)
        conv(p + "res_conv", C, cin, 1)
        for j in range(fc.est_blocks):
            p = f"{gp}1.{j}."
            nrm(p + "norm1", C)
            for q in ("to_q", "to_k", "to_v"):
                lin(p + "attn1." + q, inner, C, bias=False)
            lin(p + "attn1.to_out.0", C, inner, gain=0.5)
            nrm(p + "norm3", C)
            lin(p + "ff.net.0.proj", 4 * C, C)
            lin(p + "ff.net.2", C, 4 * C, gain=0.5)
    conv(es + "down_blocks.0.2.conv", C, C, 3)
    conv(es + "down_blocks.1.2", C, C, 3)
    conv(es + "up_blocks.0.2.conv", C, C, 4)                    # ConvTranspose1d weight [Cin, Cout, 4]
    conv(es + "up_blocks.1.2", C, C, 3)
    conv(es + "final_block.block.0", C, C, 3)
    nrm(es + "final_block.block.1", C)
    conv(es + "final_proj", fc.mel, C, 1)
    # the vocoder: the CosyVoice2 recipe with GLM's stage layout, weight_g / weight_v names
    from .tokenizer.hifigan import HiFTConfig
    cw, _ = synth_cosyvoice2_codec_weights(hift_cfg=hc, seed=seed + 1)
    H = {k.replace(".parametrizations.weight.original0", ".weight_g").replace(".parametrizations.weight.original1", ".weight_v"): v
         for k, v in cw["hift"].items()}
    return {"flow": F, "hift": H}


def _fan_in_weights(shapes, seed, gain):
    """Random-init conv / linear stacks: bf16-representable fp32 values, N(0, gain^2 / fan_in); small biases; unit norms."""
    import math
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, s in shapes.items():
        if k.endswith("cluster_usage"):
            t = 1.0 + torch.rand(s, generator=g)
        elif k.endswith("embed_sum"):
            t = torch.randn(s, generator=g)
        elif k.endswith("layernorm.weight"):
            t = 1.0 + 0.1 * torch.randn(s, generator=g)
        elif k.endswith(".scale"):
            t = torch.full(s, 0.3)
        elif k.endswith(".bias"):
            t = 0.05 * torch.randn(s, generator=g)
        else:
            fan = s[1] * (s[2] if len(s) == 3 else 1)
            t = torch.randn(s, generator=g) * (gain / math.sqrt(fan))
        W[k] = t.to(torch.bfloat16).float() if t.dim() > 1 and not k.endswith("embed_sum") else t.float()
    return W


def synth_qwen3_speaker_encoder_weights(cfg=None, seed=0):
    """Random-init ECAPA speaker encoder weights under the reference state_dict names (model/qwen3_tts_speaker.py)."""
    from .model.qwen3_tts_speaker import Qwen3TTSSpeakerEncoderConfig, param_shapes
    return _fan_in_weights(param_shapes(cfg or Qwen3TTSSpeakerEncoderConfig()), seed, 1.2)


def synth_qwen3_codec_encoder_weights(cfg=None, seed=0):
    """Random-init speech-tokenizer encoder weights under the MimiModel state_dict names (tokenizer/qwen3_codec_encoder.py)."""
    from .tokenizer.qwen3_codec_encoder import Qwen3TTSTokenizerV2EncoderConfig, param_shapes
    return _fan_in_weights(param_shapes(cfg or Qwen3TTSTokenizerV2EncoderConfig()), seed, 1.4)
