/*
 * voxhip.h — C ABI of libvoxhip.so, the MI355X (gfx950) native hot path of the vox-serve speech-LM
 * serving loop.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * The reference (vox-serve, 100 % Python) has no FFI of its own: the seam it crosses into native code
 * is the `flashinfer` wheel, reached through vox_serve/flashinfer_utils.py and vox_serve/sampling.py.
 * Each entry point below names the reference interface it replaces (file:line under /root/reference).
 * INTEGRATION.md shows the ctypes stubs that bind them from the reference's modules.
 *
 * Conventions
 *   - every `const void*` / `void*` tensor argument is a DEVICE pointer unless the name ends in _host;
 *   - bf16 tensors are contiguous row-major; "stream" is a hipStream_t passed as void*;
 *   - all calls only ENQUEUE work on the given stream (safe inside hipGraph capture), allocate nothing
 *     after the context / engine was created, and return VOX_OK or an error code (vox_last_error()).
 *   - numerics follow the fixed-order contract written in DESIGN.md §"Numeric contract" (the same one
 *     oracle/voxref.c restates on the CPU): results do not depend on grid shape or batch size.
 */
#ifndef VOXHIP_H
#define VOXHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VOX_ABI_VERSION 1
#define VOX_ATTN_CHUNK 32 /* KV tokens per attention partial (part of the numeric contract) */

typedef enum {
    VOX_OK = 0,
    VOX_ERR_INVALID = 1, /* bad argument / unsupported shape */
    VOX_ERR_HIP = 2,     /* a HIP runtime call failed */
    VOX_ERR_NOMEM = 3,
    VOX_ERR_STATE = 4
} vox_status;

typedef struct vox_ctx vox_ctx;
typedef struct vox_stack vox_stack;
typedef struct vox_qwen3 vox_qwen3;
typedef struct vox_graph vox_graph;

int vox_abi_version(void);
const char* vox_last_error(void);

/* One context per process/GPU (the reference: one scheduler process per GPU, launch.py:183-279). */
int vox_ctx_create(int device, vox_ctx** out);
void vox_ctx_destroy(vox_ctx* ctx);
/* device properties the host side sizes things with: [0]=CU count, [1]=LDS bytes/CU, [2]=HBM bytes */
int vox_ctx_props(vox_ctx* ctx, int64_t out[3]);
/* Summation order of a linear.  Calls with at most `rows` rows (1..8, default 2) run the wave64 VALU kernels: canonical order
 * (K/8 chunks of 8, chunk c on lane c % 64, sequential fmaf, xor-butterfly).  Calls with more rows run on the matrix cores:
 * the order is v_mfma_f32_16x16x32_bf16's own arithmetic (4 fused steps of 8 k per instruction: products truncated to
 * 2^(Emax-24), accumulator floored, one RNE rounding per step — measured, profiles/round2_mfma_arith.md) composed with the
 * kernel's K split: 9..128 rows and K % 256 == 0 (K/256 in {4,8,12,16,24,32} as the fusion allows): 8 contiguous K ranges added in
 * order; other shapes: 1024-wide segments, 32-wide steps interleaved over 4 accumulators; 129+ rows with a workspace: 256-wide
 * slabs added in order.  oracle/voxref.c restates all of them and oracle/policy.py the routing, so EVERY setting is bit-exact
 * against the CPU oracle.  Results still depend on the row count of the call (as the reference's do: cuBLAS picks kernels by
 * shape).  3..8 rows are ~25 % faster per frame on the matrix cores (Qwen3-TTS B=8: 6.1 -> 4.6 ms), hence the default.
 * Set before the engines are created (they size their fragment-major weight copies by it) and before the first graph capture. */
int vox_ctx_set_exact_rows(vox_ctx* ctx, int rows);

/* ---- hipGraph capture (replaces torch.cuda.graph in worker/cuda_graph_worker.py:189-805) ---------- */
int vox_graph_begin(vox_ctx* ctx, void* stream);
int vox_graph_end(vox_ctx* ctx, void* stream, vox_graph** out);
int vox_graph_launch(vox_graph* g, void* stream);
void vox_graph_destroy(vox_graph* g);

/* ---- ops: drop-in for vox_serve/flashinfer_utils.py ------------------------------------------------ */

/* rms_norm (flashinfer_utils.py:251-267): y = bf16(x * rsqrt(mean(x^2)+eps) * w); x,y [rows,cols] bf16 */
int vox_rmsnorm(vox_ctx* ctx, void* stream, const void* x, const void* w, void* y, int rows, int cols, float eps);

/* cos/sin table for apply_rope_pos_ids / apply_llama31_rope_pos_ids (flashinfer_utils.py:270-324).
 * Writes max_pos*(rot/2)*2 floats to a HOST buffer; upload it once and pass the device copy below. */
int vox_rope_table_host(float* cs_host, int max_pos, int rot, double theta, double scale, int llama31,
                        double low_freq_factor, double high_freq_factor, int old_context_len);

/* apply_rope_pos_ids: out-of-place rotary on q [N,Hq,D] and k [N,Hkv,D]; pos int32 [N] */
int vox_rope(vox_ctx* ctx, void* stream, const void* q, const void* k, void* q_out, void* k_out, const int32_t* pos,
             int N, int Hq, int Hkv, int D, int rot, int interleave, const float* cs_table, int table_max_pos);

/* Flashinfer{Prefill,Decode}Wrapper.set_kv_cache (flashinfer_utils.py:134-145, 232-244):
 * kv_layer [P,2,page_size,Hkv,D]; k,v [N,Hkv,D]; page/slot int32 [N] (page<0: skip = graph padding) */
int vox_kv_append(vox_ctx* ctx, void* stream, void* kv_layer, const void* k, const void* v, const int32_t* page,
                  const int32_t* slot, int N, int page_size, int Hkv, int D);

/* Batch{Decode,Prefill}WithPagedKVCacheWrapper.run (flashinfer_utils.py:127-132, 228-230).
 * Query row i belongs to request q_req[i] and attends to the first q_kvlen[i] tokens of its KV
 * (decode: whole length; causal prefill row j of m with kv length n: n-m+j+1).  max_kvlen bounds the
 * launch grid (rows may be shorter).  workspace: vox_attn_workspace_bytes() bytes. */
int64_t vox_attn_workspace_bytes(int Nq, int Hq, int D, int max_kvlen);
int vox_paged_attention(vox_ctx* ctx, void* stream, const void* q, const void* kv_layer, const int32_t* q_req,
                        const int32_t* q_kvlen, const int32_t* kv_indptr, const int32_t* kv_indices, void* out,
                        void* workspace, int Nq, int Hq, int Hkv, int D, int page_size, int max_kvlen, float scale);

/* nn.Linear on bf16 (qwen3_tts.py:562-601): y[b,n] = bf16(dot(W[n,:],x[b,:]) + bias[n]);
 * residual != NULL: y = bf16(residual + y).  act: 0 none, 1 SiLU applied to the rounded output. */
int vox_linear(vox_ctx* ctx, void* stream, const void* W, const void* bias, const void* x, const void* residual,
               void* y, int B, int N, int K, int act);
/* h = act(gate_proj(x)) * up_proj(x)  (Qwen3TTSMLP.forward, qwen3_tts.py:573-575) */
int vox_linear_silu_mul(vox_ctx* ctx, void* stream, const void* Wg, const void* Wu, const void* x, void* h, int B,
                        int N, int K);

/* ---- sampler: drop-in for vox_serve/sampling.py ----------------------------------------------------- */
typedef struct {
    int32_t greedy;      /* SamplingConfig.greedy or temperature == 0 (sampling.py:99-104) */
    int32_t top_k;       /* 0 = unset */
    float top_p;         /* 1.0 = unset */
    float min_p;         /* 0.0 = unset */
    float temperature;   /* 1.0 default */
    float repetition_penalty; /* 1.0 = off */
} vox_sampling_config;

/* logits[b, ids[j]] = finfo(bf16).min (qwen3_tts.py:1894-1895) */
int vox_suppress(vox_ctx* ctx, void* stream, void* logits, int B, int V, const int32_t* ids, int n);
/* Sampler.apply_repetition_penalty (sampling.py:122-146); cache uint8/bool [B,W,C,V], codebook row 0 */
int vox_rep_penalty(vox_ctx* ctx, void* stream, void* logits, const uint8_t* cache, int B, int W, int C, int V,
                    float penalty);
/* Sampler.update_repetition_penalty_cache, codebook-0 form incl. the cross-request leak (sampling.py:150-178) */
int vox_rep_update(vox_ctx* ctx, void* stream, uint8_t* cache, const int32_t* ids, int B, int W, int C, int V,
                   int window);
/* The multi-codebook forms of the two (sampling.py:122-178 with logits [B, Cl, V] / output_ids [B, Cl], Cl == C > 1): the
 * penalty of logits row (b, c) looks at cache codebook c; the update marks, like the reference's advanced-indexing assignment
 * `repetition_cache[:, w, :, output_ids] = True`, EVERY id of the step in EVERY batch row and EVERY codebook plane (all window
 * slots for the global window, the newest slot after the shift otherwise). */
int vox_rep_penalty_mc(vox_ctx* ctx, void* stream, void* logits, const uint8_t* cache, int B, int Cl, int W, int C, int V,
                       float penalty);
int vox_rep_update_mc(vox_ctx* ctx, void* stream, uint8_t* cache, const int32_t* ids, int B, int Cl, int W, int C, int V,
                      int window);
/* Sampler.run_sampling (sampling.py:85-118): out_ids int32 [B].  Stochastic modes draw from a Philox4x32-10
 * stream keyed by (seed, offset, row) — the contract oracle/voxref.c::vr_sample restates. */
int vox_sample(vox_ctx* ctx, void* stream, const void* logits, int B, int V, const vox_sampling_config* cfg,
               uint64_t seed, uint64_t offset, int32_t* out_ids);

/* ---- decoder stack engine: replaces vox_serve/model/<family>.py decoder layers --------------------- */
typedef struct {
    int32_t hidden, layers, heads, kv_heads, head_dim, ffn;
    float eps;
    int32_t qk_norm;        /* per-head RMSNorm on q,k (qwen3_tts.py:620-625) */
    int32_t qkv_bias;       /* GLM / CosyVoice2 */
    int32_t rope_dim;       /* rotary dims (<= head_dim) */
    int32_t rope_interleave;
    int32_t page_size;
    int32_t max_rows;       /* max query rows per forward (batch for decode, tokens for prefill) */
    int32_t max_kvlen;      /* bounds attention partial workspace */
} vox_stack_config;

typedef struct { /* per-layer device pointers, bf16; wqkv = rows [q;k;v] concatenated */
    const void *wqkv, *bqkv, *wo, *wgate, *wup, *wdown, *ln1, *ln2, *qnorm, *knorm;
} vox_layer_weights;

int vox_stack_create(vox_ctx* ctx, const vox_stack_config* cfg, const vox_layer_weights* layers,
                     const void* final_norm, const float* rope_table, int rope_table_max_pos, vox_stack** out);
void vox_stack_destroy(vox_stack* s);

typedef struct { /* device int32 arrays describing one ragged forward (FlashInfer*Wrapper.plan outputs) */
    const int32_t *pos, *q_req, *q_kvlen, *page, *slot, *kv_indptr, *kv_indices;
    int32_t n_rows;
    int32_t max_kvlen; /* grid bound for this call (<= cfg.max_kvlen) */
    /* optional hints that shorten the dependent-load chain of decode rows (all 0 / NULL = unused): */
    const int32_t* page_table; /* [n_rows][pt_stride]: page of KV block j of row i (same content as indices/indptr) */
    int32_t pt_stride;
    int32_t fixed_kvlen;       /* > 0: every row attends to exactly this many tokens */
    int32_t fixed_pos;         /* >= 0 with fixed_kvlen: every row's position id */
    int32_t identity_pages;    /* request r owns the single page r and row i belongs to request i */
} vox_rows;

/* x [n_rows,hidden] bf16 is updated in place layer by layer; y (may alias x) = final RMSNorm(x).
 * kv: [layers][P,2,page,Hkv,D], layer stride kv_layer_stride elements. */
int vox_stack_forward(vox_stack* s, void* stream, void* x, void* y, void* kv, int64_t kv_layer_stride,
                      const vox_rows* rows);

/* ---- Qwen3-TTS frame engine: talker decode + on-device sampling + 15-step depth loop ----------------
 * replaces CudaGraphWorker.run_lm_decode/run_lm_depth (worker/cuda_graph_worker.py:946-1160) and
 * Qwen3TTSModel.forward/sampling/depth_forward/depth_sampling (model/qwen3_tts.py:1805-2004).      */
typedef struct {
    vox_stack_config talker, depth;
    int32_t vocab, text_vocab, text_hidden, depth_vocab, n_groups, eos_id, tts_pad_id;
    int32_t max_batch;
} vox_qwen3_config;

typedef struct {
    const vox_layer_weights *talker_layers, *depth_layers;
    const void *talker_norm, *depth_norm;
    const void *codec_embedding, *text_embedding;         /* [vocab,H], [text_vocab,text_hidden] */
    const void *tp_fc1_w, *tp_fc1_b, *tp_fc2_w, *tp_fc2_b; /* text_projection */
    const void* codec_head;                                /* [vocab,H] */
    const void* const* depth_codec_embedding;              /* n_groups-1 x [depth_vocab,H] */
    const void* depth_lm_head;                             /* [n_groups-1, depth_vocab, depth_hidden] */
    const void *mtp_w, *mtp_b;                             /* small_to_mtp_projection */
    const float *talker_rope, *depth_rope;
    int32_t talker_rope_max_pos, depth_rope_max_pos;
} vox_qwen3_weights;

typedef struct { /* device buffers owned by the caller (graph-stable addresses) */
    int32_t* input_ids;      /* [max_batch, n_groups+1]; col 0 codec id, col -1 text id */
    uint8_t* input_masks;    /* [max_batch] (mask of the last column, qwen3_tts.py:1848) */
    void* input_features;    /* [max_batch, H] bf16 */
    int32_t *pos, *kvlen, *page, *slot, *kv_indptr, *kv_indices; /* plan() outputs */
    int32_t* page_table;     /* optional [max_batch][pt_stride] per-row page table for decode frames (may be NULL) */
    int64_t pt_stride;
    void* kv;                /* talker KV [layers][P,2,page,Hkv,D] */
    int64_t kv_layer_stride;
    int32_t* out_ids;        /* [max_batch, n_groups+1] sampled frame */
    void* out_logits;        /* [max_batch, vocab] bf16 codebook-0 logits after suppress/penalty */
    void* out_hidden;        /* [max_batch, H] bf16 backbone hidden (post-norm) */
    void* out_depth_logits;  /* optional [n_groups-1, max_batch, depth_vocab] bf16, may be NULL */
    void* next_features;     /* [max_batch, H] bf16: sum of depth code embeddings (next input_features) */
    uint64_t* rng_offset;    /* device counter, advanced once per frame */
} vox_qwen3_io;

int vox_qwen3_create(vox_ctx* ctx, const vox_qwen3_config* cfg, const vox_qwen3_weights* w, vox_qwen3** out);
void vox_qwen3_destroy(vox_qwen3* m);
/* Persistent depth step (round 4).  With one request in the frame, depth steps 2 .. n_groups-1 can each run as ONE launch of 256
 * resident blocks (5 layers x 4 stages + the codebook head; stage outputs handed between the blocks as tagged 8-byte granules)
 * instead of 21 dependent launches — bit-identical results.  Chosen at vox_qwen3_create: on by default for the Qwen3-TTS depth shape on a
 * part with >= 256 CUs, VOX_DEPTH_PERSIST=0 in the environment keeps the launch chain.  All 256 blocks must be resident together: run ONE
 * such frame at a time per GPU (two engines replaying one-request frames concurrently on two streams can starve each other's blocks).
 * The same mechanism runs the MLP half of every talker layer of a one-request frame (o_proj + residual, gate/up, down + residual in one
 * launch; VOX_TALKER_PERSIST=0 keeps the three launches).  `*enabled`: bit 0 = depth steps, bit 1 = talker MLP halves.  `*enabled`: whether this engine runs them; `*error_code`: 0, or the code of the first hand-off that timed
 * out (bounded spins: a stuck launch gives up with garbage instead of hanging; everything since is invalid).  Synchronises the device. */
int vox_qwen3_depth_persist_status(vox_qwen3* m, int32_t* enabled, uint32_t* error_code);
/* Fail loud within one frame (round 5).  The reference's graph replay cannot return garbage silently
 * (worker/cuda_graph_worker.py:946-1056); a persistent kernel whose hand-off times out can, so its error word travels with every frame:
 *   vox_qwen3_set_status     the LAST kernel of every frame / prefill writes the error code (0 = every hand-off arrived) to `status_dev`
 *                            (device int32, e.g. the word in front of the out_ids rows the host copies anyway); decode frames also save
 *                            their inputs (ids / masks / features of their rows, the frame counter) to a two-slot shadow.  Call before
 *                            the first frame graph is captured.
 *   vox_qwen3_frame_restore  puts the inputs of the frame `back` frames ago (1 = the last, 2 = the one before: one frame may be in
 *                            flight behind a failed one; 0 = the frame that had been started from the counter's CURRENT value, i.e.
 *                            during a replay the launch behind the failed one, when the caller had staged its inputs by hand) back
 *                            in place, rows 0..n_rows-1 and the counter (n_rows = 0: the counter only — a prefill's rows are staged
 *                            by the caller); the status word reads 0 afterwards, 0x7fffffff if no such frame is in the shadow (and
 *                            nothing was written: callers validate with it BEFORE they reset anything).
 *   vox_qwen3_persist_reset  waits for the device, clears the error words, re-zeroes the hand-off granules; disable != 0 turns the
 *                            persistent kernels off: later frames take the bit-identical launch chains (drop graphs captured before).
 *   vox_qwen3_persist_set_spins   bound of every poll loop in passes (>= ~0.5 us each; default 40000 = >= 20 ms, or VOX_PERSIST_SPINS).
 *   vox_qwen3_persist_inject      TEST HOOK: in the next `count` launches of persistent kernel `which` (0 depth step, 1 talker MLP half)
 *                            block 1 withholds its first publish — the effect of a block that never got a CU.
 * Recovery = restore + reset(disable) + re-run the frame: bit-identical to a healthy run (tests/test_gpu_qwen3.py). */
int vox_qwen3_set_status(vox_qwen3* m, int32_t* status_dev);
int vox_qwen3_frame_restore(vox_qwen3* m, void* stream, const vox_qwen3_io* io, int back, int n_rows);
int vox_qwen3_persist_reset(vox_qwen3* m, int disable);
int vox_qwen3_persist_set_spins(vox_qwen3* m, uint32_t spins);
int vox_qwen3_persist_inject(vox_qwen3* m, int which, uint32_t count);
/* Enqueue one whole frame for `batch` rows.  With feedback != 0 the engine also writes the next step's
 * input_ids / input_masks / input_features in place (qwen3_tts.py:1931-1944), so frames can be replayed
 * back-to-back from one captured graph while the host only advances the plan arrays. */
int vox_qwen3_frame(vox_qwen3* m, void* stream, const vox_qwen3_io* io, int batch, int max_kvlen,
                    const vox_sampling_config* sampling, uint64_t seed, int feedback);
/* Ragged prefill of n_rows tokens (one or more requests); logits/hidden of row `last_rows[r]` go to
 * out_logits/out_hidden row r, then sampling + depth loop run for n_req rows as in vox_qwen3_frame.
 * n_req == 0 (here and in vox_lm_prefill / vox_csm_prefill): a context chunk of a prompt longer than the row capacity —
 * embedding + decoder stack only (K/V appended at the rows' page/slot), nothing sampled, no state advanced; the chunk
 * that holds the prompt's last row is a normal call (the reference raises for > 1024 tokens, cuda_graph_worker.py:61). */
int vox_qwen3_prefill(vox_qwen3* m, void* stream, const vox_qwen3_io* io, const int32_t* row_ids /*[n_rows,n_groups+1]*/,
                      const uint8_t* row_masks, const void* row_features, const int32_t* q_req, int n_rows,
                      const int32_t* last_rows, int n_req, int max_kvlen, const vox_sampling_config* sampling,
                      uint64_t seed, int feedback);

/* Voice-clone prompt features of Qwen3TTSModel.preprocess (model/qwen3_tts.py:1657-1672 speaker position,
 * :1733-1744 ICL rows): spk_out[H] = bf16(speaker_embedding - codec_embedding[codec_pad_id]) when speaker_embedding != NULL;
 * icl_out[t][H] = bf16 running sum over codebooks 1..n_groups-1 of code_predictor.codec_embedding[cb-1][ref_codes[t][cb]]
 * (ref_codes int32 [n_frames][n_groups]; each add rounded to bf16 in codebook order, as the reference's bf16 `+=`). */
int vox_qwen3_prompt_features(vox_qwen3* m, void* stream, const int32_t* ref_codes, int n_frames, const void* speaker_embedding,
                              int codec_pad_id, void* spk_out, void* icl_out);

/* ---- CSM-1B frame engine: backbone decode + on-device sampling + 31-step depth loop --------------------
 * replaces CSMModel.forward / sampling / depth_forward / depth_sampling (model/csm.py:637-770) and the depth loop of
 * ModelWorker.run_lm_depth (worker/base.py:546-614).  Backbone input = sum over the 33 masked per-column embeddings
 * (csm.py:647-653); the depth transformer consumes [backbone hidden, embedding of codebook 0] through
 * inputs_embeds_projector and predicts codebook i with codebooks_head.weight[i-1] (csm.py:235-255).               */
typedef struct vox_csm vox_csm;
typedef struct {
    vox_stack_config backbone, depth;
    int32_t vocab;       /* audio vocabulary per codebook (2051) */
    int32_t text_vocab;  /* rows of the text embedding */
    int32_t n_codebooks; /* audio codebooks (32); input rows have n_codebooks + 1 columns, the last one is text */
    int32_t max_batch;
} vox_csm_config;
typedef struct {
    const vox_layer_weights* backbone_layers;
    const vox_layer_weights* depth_layers;
    const void *backbone_norm, *depth_norm;
    const void* audio_embedding; /* [n_codebooks*vocab, H]: backbone_model.embed_tokens.embed_audio_tokens */
    const void* text_embedding;  /* [text_vocab, H] */
    const void* lm_head;         /* [vocab, H] */
    const void* depth_proj;      /* [depth_hidden, H]: inputs_embeds_projector */
    const void* depth_heads;     /* [n_codebooks-1, vocab, depth_hidden]: codebooks_head.weight transposed per codebook */
    const float *backbone_rope, *depth_rope;
    int32_t backbone_rope_max_pos, depth_rope_max_pos;
} vox_csm_weights;
typedef struct { /* device buffers owned by the caller (graph-stable addresses) */
    int32_t* input_ids;      /* [max_batch, n_codebooks+1] */
    uint8_t* input_masks;    /* [max_batch, n_codebooks+1] */
    int32_t *pos, *kvlen, *page, *slot, *kv_indptr, *kv_indices;
    int32_t* page_table;     /* optional per-row page table for decode frames */
    int64_t pt_stride;
    void* kv;
    int64_t kv_layer_stride;
    int32_t* out_ids;        /* [max_batch, n_codebooks+1] sampled frame (text column = codebook 0, csm.py:697) */
    void* out_logits;        /* [max_batch, vocab] bf16 */
    void* out_hidden;        /* optional [max_batch, H] bf16 backbone hidden (post-norm) */
    void* out_depth_logits;  /* optional [n_codebooks-1, max_batch, vocab] bf16 */
    uint64_t* rng_offset;
} vox_csm_io;
int vox_csm_create(vox_ctx* ctx, const vox_csm_config* cfg, const vox_csm_weights* w, vox_csm** out);
void vox_csm_destroy(vox_csm* m);
int vox_csm_frame(vox_csm* m, void* stream, const vox_csm_io* io, int batch, int max_kvlen,
                  const vox_sampling_config* sampling, uint64_t seed, int feedback);
int vox_csm_prefill(vox_csm* m, void* stream, const vox_csm_io* io, const int32_t* row_ids /*[n_rows,n_codebooks+1]*/,
                    const uint8_t* row_masks /*[n_rows,n_codebooks+1]*/, const int32_t* q_req, int n_rows,
                    const int32_t* last_rows, int n_req, int max_kvlen, const vox_sampling_config* sampling, uint64_t seed,
                    int feedback);

/* ---- single-stack speech LM engine (GLM-4-Voice, CosyVoice2 LLM, Orpheus-style families) ------------------
 * replaces <Family>Model.forward + sampling (model/glm_voice.py:517-590, model/cosyvoice2.py:1008-1090) and
 * the decode/prefill graph replays of CudaGraphWorker (worker/cuda_graph_worker.py:806-1056).                */
typedef struct vox_lm vox_lm;
typedef struct {
    vox_stack_config stack;
    int32_t vocab_in, vocab_out;
    int32_t ids_stride;   /* columns of input_ids (n_codebooks); column 0 is embedded */
    int32_t input_mode;   /* 0: x = embedding[id];  1: x = mask ? input_features : embedding[clamp(id)] (cosyvoice2.py:1020-1024) */
    int32_t max_batch;
} vox_lm_config;
typedef struct {
    const vox_layer_weights* layers;
    const void *final_norm, *embedding /*[vocab_in,H]*/, *head_w /*[vocab_out,H]*/, *head_b /*[vocab_out] or NULL*/;
    const float* rope;
    int32_t rope_max_pos;
} vox_lm_weights;
typedef struct {
    int32_t* input_ids;      /* [max_batch, ids_stride] */
    uint8_t* input_masks;    /* [max_batch] (mode 1) */
    void* input_features;    /* [max_batch, H] bf16 (mode 1) */
    int32_t *pos, *kvlen, *page, *slot, *kv_indptr, *kv_indices, *page_table;
    int64_t pt_stride;
    void* kv;
    int64_t kv_layer_stride;
    int32_t* out_ids;        /* [max_batch] */
    void* out_logits;        /* [max_batch, vocab_out] bf16 (after the repetition penalty) */
    uint8_t* rep_cache;      /* optional [max_batch, rep_w, 1, vocab_out]; penalty applied and cache updated in place */
    int32_t rep_w, rep_window;
    uint64_t* rng_offset;
} vox_lm_io;
int vox_lm_create(vox_ctx* ctx, const vox_lm_config* cfg, const vox_lm_weights* w, vox_lm** out);
void vox_lm_destroy(vox_lm* m);
int vox_lm_frame(vox_lm* m, void* stream, const vox_lm_io* io, int batch, int max_kvlen,
                 const vox_sampling_config* sampling, uint64_t seed, int feedback);
int vox_lm_prefill(vox_lm* m, void* stream, const vox_lm_io* io, const int32_t* row_ids, const uint8_t* row_masks,
                   const void* row_features, const int32_t* q_req, int n_rows, const int32_t* last_rows, int n_req,
                   int max_kvlen, const vox_sampling_config* sampling, uint64_t seed, int feedback);

/* ---- Qwen3-TTS 12 Hz codec decoder (token -> waveform), streaming ------------------------------------
 * replaces Qwen3TTSTokenizerV2Decoder.forward_chunk (tokenizer/qwen3_codec.py:1541-1666), the DecoderCache
 * plumbing (tokenizer/base.py:7-173) and CudaGraphWorker.run_detokenize's cache cat / copy-in / copy-out
 * (worker/cuda_graph_worker.py:1217-1241) with in-place per-slot state.                                */
typedef struct vox_codec vox_codec;

typedef struct { /* one conv / transposed conv / linear as an implicit GEMM; w: bf16 [n_taps][n][cin] */
    const void* w;
    const float* bias;     /* [bias_mod] or NULL */
    int32_t n_taps, n, cin, bias_mod;
} vox_conv_w;
typedef struct { const float *alpha, *inv_beta; } vox_snake_w; /* exp(alpha), 1/(exp(beta)+1e-9) */
typedef struct {
    const float *ln1, *scale1, *ln2, *scale2;
    vox_conv_w qkv, o, gate_up, down;
} vox_codec_layer_w;
typedef struct {
    vox_conv_w tconv, pw1, pw2;
    const float *dw_w, *dw_b, *ln_w, *ln_b, *gamma;
} vox_codec_up_w;
typedef struct { vox_snake_w act1, act2; vox_conv_w conv1, conv2; } vox_codec_res_w;
typedef struct { vox_snake_w snake0; vox_conv_w tconv; vox_codec_res_w res[3]; } vox_codec_block_w;
typedef struct {
    const float* emb;      /* [num_quantizers][codebook_size][vq_dim] = embedding_sum / clamp(cluster_usage) */
    vox_conv_w rvq_first_out, rvq_rest_out, pre_conv, in_proj, out_proj, dec0;
    vox_codec_layer_w layers[16];
    const float* final_norm;
    const float* inv_freq; /* [head_dim/2] */
    vox_codec_up_w up[2];
    vox_codec_block_w blocks[4];
    vox_snake_w final_snake;
    const float* final_w;  /* [C][7] */
    float final_b;
} vox_codec_weights;
typedef struct {
    int32_t codebook_size, codebook_dim, vq_dim, latent_dim, decoder_dim, hidden, intermediate, head_dim, num_heads,
        num_layers, num_quantizers, window, rates[4], n_blocks, n_upsample;
    float rms_eps, rope_theta;
} vox_codec_config;

/* Operand precision of the decoder's conv / linear GEMMs = the number of bf16 terms an fp32 activation enters the matrix cores as
 * (fp32 accumulation).  2 (default): the two leading terms, 16 significand bits — waveform within 1e-4 RMS of the reference module
 * evaluated in fp32 (measured 1.7e-5 on the full-size fixture); the activated tensors between the decoder convs are then STORED as
 * those two terms (split once by the producing kernel), as are their history rows in the streaming state.  3: every product exact.
 * 1: activations rounded to bf16 — the precision the reference itself serves at (it runs the decoder in bf16, qwen3_tts.py:1061-1064).
 * Set it before the object's first decode call: once a chunk has run the slots hold history in that format, and a CHANGE of the
 * value is refused with VOX_ERR_INVALID (same value again: no-op). */
int vox_codec_set_operand_planes(vox_codec* m, int planes);
int vox_codec_create(vox_ctx* ctx, const vox_codec_config* cfg, const vox_codec_weights* w, int max_batch, int max_slots,
                     int frames_per_chunk, vox_codec** out);
void vox_codec_destroy(vox_codec* m);
int64_t vox_codec_state_bytes(vox_codec* m); /* streaming state per slot */
/* a new request takes over `slot`: zero its state (audio_decoder_initial_cache, model/qwen3_tts.py:1243-1260) */
int vox_codec_reset_slot(vox_codec* m, void* stream, int slot);
/* codes int32 [n, T, code_stride] (first num_quantizers columns), slots int32 [n], out fp32 [n, T*hop] */
int vox_codec_decode_chunk(vox_codec* m, void* stream, const int32_t* codes, int code_stride, const int32_t* slots, int n,
                           int T, float* out);

/* ---- Mimi codec decoder (token -> waveform) for CSM, stateless per chunk -------------------------------------
 * replaces MimiDecoder.decode / MimiModel.decode (tokenizer/mimi.py:2993-3022, 3085-3089) as CSMModel.postprocess calls
 * it (model/csm.py:772-787): split RVQ decode, channel-wise x2 transposed-conv upsample, 8-layer causal transformer
 * (LayerNorm, RoPE on interleaved pairs, LayerScale, GELU MLP), SEANet decoder (ratios 8,6,5,4).  Every chunk starts
 * from zero history, like the reference.                                                                          */
typedef struct vox_mimi vox_mimi;
typedef struct {
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b, *scale1, *scale2;
    vox_conv_w qkv, o, fc1, fc2;
} vox_mimi_layer_w;
typedef struct { vox_conv_w tconv, conv1, conv2; } vox_mimi_block_w; /* ELU, transposed conv, [ELU conv k3, ELU conv k1] + skip */
typedef struct {
    const float* emb;           /* [n_q][bins][vq_dim] = embedding_sum / clamp(cluster_usage) */
    vox_conv_w rvq_first_out, rvq_rest_out;
    const float* up_w;          /* [dim][4]: channel-wise transposed conv, kernel 4, stride 2 */
    vox_mimi_layer_w layers[16];
    vox_conv_w dec0;
    vox_mimi_block_w blocks[4];
    const float* final_w;       /* [n_filters][last_kernel] */
    float final_b;
} vox_mimi_weights;
typedef struct {
    int32_t bins, vq_dim, dim, num_heads, num_layers, ffn, n_q, n_filters, ratios[4], kernel_size, last_kernel_size, context;
    float max_period, ln_eps;
} vox_mimi_config;
int vox_mimi_create(vox_ctx* ctx, const vox_mimi_config* cfg, const vox_mimi_weights* w, int max_batch, int max_frames,
                    vox_mimi** out);
void vox_mimi_destroy(vox_mimi* m);
/* codes int32 [n, T, code_stride] (first n_q columns, clamped to [0, bins-1]); out fp32 [n, T * 2 * prod(ratios)] */
int vox_mimi_decode(vox_mimi* m, void* stream, const int32_t* codes, int code_stride, int n, int T, float* out);
/* Streaming option (the reference decodes every 10-frame chunk from a fresh state, mimi.py:3085-3089, although
 * the module carries streaming state, mimi.py:2042-2215,1213-1305): per-slot look-back rows of every causal / transposed
 * conv input and a K/V ring (context + chunk rows, post-RoPE keys) per transformer layer, RoPE at absolute positions.  Chunked
 * output == one decode of the whole sequence (tests/test_gpu_codec.py), at the same cost per chunk as the stateless call.
 * vox_mimi_stream_enable allocates the state for max_slots requests (once); slots: device int32 [n], distinct. */
int vox_mimi_stream_enable(vox_mimi* m, int max_slots);
int vox_mimi_reset_slot(vox_mimi* m, void* stream, int slot);
int vox_mimi_decode_chunk(vox_mimi* m, void* stream, const int32_t* codes, int code_stride, const int32_t* slots, int n, int T,
                          float* out);

/* ---- SNAC decoder (token -> waveform of the Orpheus family) ---------------------------------------------------------
 * Replaces SNAC.decode (/root/reference/vox_serve/tokenizer/snac.py:438-441: ResidualVectorQuantize.from_codes :350-357 +
 * Decoder :119-158 with DecoderBlock :215-241, ResidualUnit :160-176, NoiseBlock :201-212, Snake1d :253-267) as
 * OrpheusModel.postprocess calls it (model/orpheus.py:483-507): the depthwise / no-local-attention variant (snac_24khz, what the
 * reference's Orpheus plugin loads) and the dense-conv / LocalMHA variants of the module (conv0 / res[].dense / attn_*).
 * Stateless per window, fp32 activations, convolutions as implicit GEMMs on the matrix cores with exact products (fp32
 * activations split into three bf16 terms, fp32 weights carried as two bf16 planes = 16 significand bits).
 * Weights: weight-norm already folded (w = g * v / ||v||).  tab[i]: out_proj_i(codebook_i) + bias, tabulated [codebook_size][latent].
 * Conv weights as vox_conv_w with 2 * taps planes (taps of the high plane, then the same taps of the residual plane).
 * NoiseBlock: x + noise[b, t] * conv1x1(x); noise is either given (fp32, per stage i the block [n][T_i], stages concatenated) or
 * generated on the device: Philox4x32-10 keyed by `seed`, counter (t, stream, 0, 0), stream = stream_base[b] + stage
 * (stream_base NULL: b * n_stages), Box-Muller of words 0 and 1 — the stream oracle/snac_ref.py::philox_noise restates. */
typedef struct {
    vox_snake_w act1, act2;
    const float *dw_w, *dw_b;      /* depthwise conv: [C][7], [C] */
    vox_conv_w pw;                 /* 1x1 conv + bias */
    vox_conv_w dense;              /* non-depthwise variant: the k7 conv as 7 taps x 2 planes [C][C] + bias (n_taps == 0: depthwise, dw_w) */
} vox_snac_res_w;
typedef struct {
    vox_snake_w snake0;
    vox_conv_w tconv;              /* ConvTranspose1d(k = 2r, stride r, padding r/2): 2 taps x 2 planes, N = r * Cout, bias_mod = Cout */
    vox_conv_w noise;              /* NoiseBlock 1x1 conv (no bias); n_taps == 0: no noise block */
    vox_snac_res_w res[3];         /* dilations 1, 3, 9 */
} vox_snac_block_w;
typedef struct {
    const float* tab[4];           /* per VQ level: [codebook_size][latent_dim] */
    const float *dw0_w, *dw0_b;    /* depthwise k7 on the latent: [latent][7], [latent] */
    vox_conv_w pw0;                /* 1x1 latent -> decoder_dim */
    vox_snac_block_w blocks[4];
    vox_snake_w final_snake;
    const float* final_w;          /* [C_last][7] */
    float final_b;
    /* variants of the 32 / 44 kHz checkpoints (snac.py:20-90, 119-176) */
    vox_conv_w conv0;              /* non-depthwise: ONE k7 conv latent -> decoder_dim, 7 taps x 2 planes + bias (n_taps == 0: dw0 + pw0) */
    const float *attn_ln_w, *attn_ln_b;   /* LocalMHA after the input conv(s): LayerNorm (eps 1e-5) */
    vox_conv_w attn_qkv, attn_out;        /* to_qkv [3C][C], to_out [C][C], no bias, 1 tap x 2 planes; heads of 64, rotary per window position */
} vox_snac_weights;
typedef struct {
    int32_t latent_dim, decoder_dim, codebook_size, n_levels, vq_strides[4], rates[4], noise;
    int32_t attn_window;           /* 0: no LocalMHA; else T must be a multiple of it (<= 32) */
} vox_snac_config;
typedef struct vox_snac vox_snac;
int vox_snac_create(vox_ctx* ctx, const vox_snac_config* cfg, const vox_snac_weights* w, int max_batch, int max_T, vox_snac** out);
void vox_snac_destroy(vox_snac* m);
/* codes: device int32 [n][sum_i T / vq_strides[i]], level-major per request (all of level 0, then level 1, ...), clamped to the
 * codebook; T % vq_strides[0] == 0, T <= max_T.  out: fp32 [n][out_len] = samples [out_off, out_off + out_len) of the T * hop decoded. */
int vox_snac_decode(vox_snac* m, void* stream, const int32_t* codes, int n, int T, const float* noise, uint64_t seed,
                    const uint32_t* stream_base, float* out, int out_off, int out_len);

/* ---- HiFT vocoder (mel -> waveform of the CosyVoice2 / GLM-4-Voice detokenizers) ----------------------------------------
 * Replaces HiFTGenerator.forward_chunk (/root/reference/vox_serve/tokenizer/hifigan.py:641-665: ConvRNNF0Predictor :394-426,
 * SourceModuleHnNSF2 / SineGen2 :251-391, _stft :542-552, decode :596-628 with ResBlock :98-141 and Snake :45-95,
 * _istft_graph_safe :566-594) as CosyVoice2Decoder.decode_chunk calls it (tokenizer/cosyvoice2.py:1043-1046: no source cache).
 * Stateless per chunk, fp32 activations time-major [request * t][C]; every conv / transposed conv is an implicit GEMM on the
 * matrix cores with exact products (fp32 activations = three bf16 terms, fp32 weights = two bf16 planes).
 * Weights: weight norm already folded.  A conv of kernel k (dilation d, "same" padding) is a vox_conv_w of 2k taps: the k taps of
 * the high plane [j][Cout][Cin] (tap j reads row t + (j - (k-1)/2) d), then the k taps of the residual plane.  Cin is padded to a
 * multiple of 32 with zero columns (conv_pre / f0 conv 0: mel channels).  ConvTranspose1d(k, stride u, padding (k-u)/2) is the
 * taps d = dmin..dmax (dmin = ceil((-(k-u)/2 - (u-1)) / u), dmax = floor((k-1-(k-u)/2) / u)) of an N = u * Cout GEMM:
 * W_d[phi * Cout + co][ci] = w[ci][co][d u + phi + (k-u)/2] (0 outside the kernel), tap d reads row t - d; bias_mod = Cout.
 * source_downs are plain fp32 [Cout][n_fft + 2][k] (strided convs over the 18-row STFT of the source; direct kernel).
 * Harmonic source: SineGen2's torch.rand initial phases never reach the output (the 1/scale linear resampling reads samples
 * scale*j + scale/2 - 1 and scale*j + scale/2 only, the phases are added to sample 0), so only its additive noise is part of
 * the contract: given (fp32 [n][T * scale][H + 1]) or generated on the device: Philox4x32-10 keyed by `seed`, counter
 * (l * (H + 1) + h, stream, 0, 0), stream = stream_base[b] + 1 (stream_base NULL: 2 b + 1), Box-Muller of words 0 and 1 — the
 * stream oracle/hift_ref.py::make_noise restates. */
typedef struct { vox_conv_w c1[3], c2[3]; vox_snake_w a1[3], a2[3]; } vox_hift_resblock_w;
typedef struct {
    vox_conv_w f0_conv[5];             /* k3 convs of the f0 predictor (ELU after each); THREE weight planes (9 taps): the harmonic
                                          source multiplies an f0 error by 2 pi scale T */
    const float* f0_cls_w;             /* [f0_channels] */
    float f0_cls_b;
    const float* src_lin_w;            /* [H + 1] */
    float src_lin_b;
    vox_conv_w conv_pre;               /* k7 */
    vox_conv_w ups[4];
    const float *sd_w[4], *sd_b[4];    /* source_downs[i]: fp32 [Cout_i][n_fft + 2][k_i], [Cout_i] */
    vox_hift_resblock_w src_rb[4];     /* source_resblocks[i] */
    vox_hift_resblock_w rb[12];        /* resblocks[i * n_kernels + j] */
    vox_conv_w conv_post;              /* k7, N = n_fft + 2 */
} vox_hift_weights;
typedef struct {
    int32_t in_channels, in_channels_padded, base_channels, nb_harmonics, sampling_rate, n_stages, upsample_rates[4],
        upsample_kernels[4], n_fft, hop_len, n_kernels, resblock_kernels[4], dilations[3], source_resblock_kernels[4], f0_channels;
    float nsf_alpha, nsf_sigma, voiced_threshold, lrelu_slope, audio_limit;
    int32_t sine_gen_v1;   /* 1: GLM-4-Voice's SineGen (tokenizer/glm.py:2254-2331): phase accumulated per SAMPLE, theta = 2 pi (cumsum(f0 h / sr) mod 1)
                              + a random initial phase per harmonic (uniform stream stream_base[b], element h; or rand_ini), which does matter */
} vox_hift_config;
typedef struct vox_hift vox_hift;
int vox_hift_create(vox_ctx* ctx, const vox_hift_config* cfg, const vox_hift_weights* w, int max_batch, int max_T, vox_hift** out);
void vox_hift_destroy(vox_hift* m);
/* mel: device fp32 [n][in_channels][T] (the reference's layout); wav: fp32 [n][T * scale], scale = prod(upsample_rates) * hop_len;
 * source (optional): fp32 [n][T * scale], the merged harmonic source (the reference returns it for its cache). */
/* rand_ini (sine_gen_v1 only, optional): fp32 [n][H + 1] uniforms in [0, 1) for the initial phases (NULL: the seeded stream) */
int vox_hift_decode(vox_hift* m, void* stream, const float* mel, int n, int T, const float* noise, uint64_t seed,
                    const uint32_t* stream_base, float* wav, float* source, const float* rand_ini);

/* ---- CosyVoice2 flow: speech tokens -> mel (conformer encoder + 10-step conditional flow matching) -----------------------
 * Replaces CausalMaskedDiffWithXvec.forward_chunk (/root/reference/vox_serve/tokenizer/cosyvoice_flow.py:2909-2980:
 * UpsampleConformerEncoder.forward_chunk :1185-1358 with RelPositionMultiHeadedAttention :742-860, CausalConditionalCFM
 * .solve_euler_with_cache :2700-2793, CausalConditionalDecoder.forward_chunk :2440-2586) and the flow half of
 * CosyVoice2Decoder.init_cache / decode_chunk (tokenizer/cosyvoice2.py:862-1046) in the plugin's default shared-prompt mode
 * (model/cosyvoice2.py:325,1093-1103: every chunk of every request is decoded against the static caches of the prompt).
 * fp32 activations time-major [request * t][C]; linears / convs are implicit GEMMs with exact products against bf16 weights
 * (one plane: the reference casts this module to bf16, cosyvoice2.py:837); attention, LayerNorm, Mish / SiLU / GELU in fp32.
 * vox_flow_set_prompt runs init_cache natively (prompt tokens + the first 3 again, prompt mel as the condition, empty caches) and
 * keeps the truncated caches (16-entry prefix + suffix, 128 entries; 64 for the first encoder stack) on the device.
 * The CFM start noise [mel][frames] (one draw per call, shared by the batch, as the reference): given, or the seeded Philox stream
 * (counter (c * frames + t, stream, 0, 0), Box-Muller) that oracle/flow_ref.py::cfm_noise restates. */
typedef struct {
    vox_conv_w qkv, out, pos;            /* linear_q | linear_k | linear_v fused (N = 3 D), linear_out, linear_pos (no bias) */
    const float *bias_u, *bias_v;        /* [heads][d_k] */
    vox_conv_w w1, w2;                   /* feed_forward (SiLU between) */
    const float *ln_mha_w, *ln_mha_b, *ln_ff_w, *ln_ff_b;
} vox_flow_conformer_w;
typedef struct {
    vox_conv_w conv1, conv2, res;        /* CausalBlock1D convs k3 (taps t-2, t-1, t), res_conv k1 */
    const float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    vox_conv_w mlp;                      /* Mish -> Linear(time_embed -> C) */
} vox_flow_resnet_w;
typedef struct {
    const float *ln1_w, *ln1_b, *ln3_w, *ln3_b;
    vox_conv_w qkv, out, ff1, ff2;       /* to_q | to_k | to_v fused (no bias), to_out.0, GELU proj, ff out */
} vox_flow_tblock_w;
typedef struct {
    const float* embedding;              /* input_embedding [vocab][D] fp32 */
    vox_conv_w spk;                      /* spk_embed_affine_layer */
    vox_conv_w embed_lin, up_embed_lin;  /* LinearNoSubsampling.out.0 */
    const float *embed_ln_w, *embed_ln_b, *up_embed_ln_w, *up_embed_ln_b, *after_w, *after_b;
    vox_conv_w pre1, pre2, up_conv;      /* PreLookaheadLayer conv1 (taps t .. t+3), conv2 (t-2 .. t), Upsample1D conv (t-4 .. t) */
    const vox_flow_conformer_w* enc;     /* [enc_layers] then [up_layers] */
    vox_conv_w enc_proj;
    vox_conv_w time1, time2;             /* TimestepEmbedding */
    const vox_flow_resnet_w* resnets;    /* [1 + mid + 1]: down, mid..., up */
    const vox_flow_tblock_w* tblocks;    /* [(1 + mid + 1) * n_blocks] in the same order */
    vox_conv_w down_conv, up_conv2, final_conv, final_proj;
    const float *final_ln_w, *final_ln_b;
} vox_flow_weights;
typedef struct {
    int32_t vocab, dim, mel, spk_dim, enc_layers, up_layers, enc_heads, enc_ffn, pre_lookahead, est_ch, est_heads, est_head_dim,
        est_blocks, est_mid, n_steps, max_cache, prefix;
    float cfg_rate;
} vox_flow_config;
typedef struct vox_flow vox_flow;
/* time_emb: host fp32 [n_steps][4 mel], the sinusoidal embedding of every Euler step's t (SinusoidalPosEmb, scale 1000); dt: host fp32
 * [n_steps], the step sizes of the cosine schedule — both computed by the caller exactly as the reference computes them */
int vox_flow_create(vox_ctx* ctx, const vox_flow_config* cfg, const vox_flow_weights* w, int max_batch, int max_T, int max_prompt_T,
                    const float* time_emb, const float* dt, vox_flow** out);
void vox_flow_destroy(vox_flow* m);
/* prompt_tokens: device int32 [n_prompt]; prompt_feat: device fp32 [2 n_prompt][mel]; embedding: device fp32 [spk_dim];
 * noise: device fp32 [mel][2 (n_prompt + 3)] or NULL (seeded stream `stream`); prompt_mel (optional): fp32 [mel][2 (n_prompt + 3)] */
int vox_flow_set_prompt(vox_flow* m, void* stream, const int32_t* prompt_tokens, int n_prompt, const float* prompt_feat,
                        const float* embedding, const float* noise, uint64_t seed, uint32_t noise_stream, float* prompt_mel);
/* tokens: device int32 [n][T]; mel: fp32 [n][mel][2T] (the reference's layout); mu (optional, debugging): the encoder output [n][2T][mel] */
int vox_flow_decode_chunk(vox_flow* m, void* stream, const int32_t* tokens, int n, int T, const float* noise, uint64_t seed,
                          uint32_t noise_stream, float* mel, float* mu);

/* Per-request evolving caches — CosyVoice2Decoder.decode_chunk with shared_prompt_cache_mode=False, i.e. the plugin's
 * use_detokenizer_cache=True (tokenizer/cosyvoice2.py:1010-1083, model/cosyvoice2.py:514-560, 1104-1117): every request owns a
 * copy of the five flow caches.  They start as the prompt's (vox_flow_slot_reset = the reference's expanded initial cache), every
 * chunk is decoded against them, and they then take the chunk's rows under the reference's sliding window (first PREFIX_LEN rows +
 * the most recent ones, :1016-1046) — held as a ring, so no row is moved.  `slots`: HOST int32 [n]; the requests of one call must
 * have equal cache states (the reference concatenates their cache tensors, so it needs equal lengths; here the ring offsets must agree
 * as well).  vox_flow_slot_state -> {encoder, up-encoder, estimator} cache lengths, then their ring offsets. */
int vox_flow_enable_slots(vox_flow* m, int n_slots);
int vox_flow_slot_reset(vox_flow* m, void* stream, int slot);
int vox_flow_slot_state(vox_flow* m, int slot, int32_t out[6]);
int vox_flow_decode_chunk_slots(vox_flow* m, void* stream, const int32_t* tokens, int n, int T, const int32_t* slots, const float* noise,
                                uint64_t seed, uint32_t noise_stream, float* mel, float* mu);

/* fade_in_out (tokenizer/cosyvoice2.py:46-54): wav [n][L] (in place): the first `fade` samples of every row become
 * wav * window[:fade] + prev_tail * window[fade:] (prev_tail [n][fade] or NULL = silence; window: device double [2 fade]) */
int vox_fade_in_out(void* stream, float* wav, int n, int L, const float* prev_tail, const double* window, int fade);
/* z [mel][frames] = the seeded CFM start noise vox_flow_* draw when `noise` is NULL (for callers that replay captured graphs) */
int vox_flow_fill_noise(void* stream, uint64_t seed, uint32_t noise_stream, int mel, int frames, float* z);

/* ---- GLM-4-Voice flow: speech tokens -> mel ---------------------------------------------------------------------------
 * Replaces GLMFlowModel.inference (/root/reference/vox_serve/tokenizer/glm.py:2065-2112: BlockConformerEncoder :1005-1112 with
 * BlockRelPositionMultiHeadedAttention :434-599, InterpolateRegulator :1114-1148, ConditionalCFM.solve_euler :1951-1990 with
 * ConditionalDecoder.forward :1812-1895) as GLMAudioDecoder.forward calls it (:2640-2651).  Stateless per call.  Same arithmetic
 * conventions as vox_flow_* (fp32 activations, bf16 weights as the GEMM B operand, one plane).  The two estimator calls per Euler
 * step (conditional / unconditional) run as one doubled batch.  GroupNorm weights ride in the ln*_w / ln*_b fields of
 * vox_flow_resnet_w.  mel rows of the regulator are kept mel_padded wide (a multiple of 32; pad columns zero).
 * Start noise: per request [mel][frames] (the reference draws randn_like(mu)): given, or Philox stream first_stream + b. */
typedef struct {
    const float* embedding;              /* input_embedding [vocab][D] */
    vox_conv_w spk, embed_lin;
    const float *embed_ln_w, *embed_ln_b, *after_w, *after_b;
    const vox_flow_conformer_w* enc;     /* [enc_layers] */
    vox_conv_w enc_proj;                 /* N = mel_padded (rows >= mel zero) */
    vox_conv_w reg_conv[4];              /* k3 "same" convs mel_padded -> mel_padded */
    const float *reg_gn_w[4], *reg_gn_b[4];
    vox_conv_w reg_out;                  /* k1, mel_padded -> mel */
    vox_conv_w time1, time2;
    const vox_flow_resnet_w* resnets;    /* [2 + mid + 2]: down 0, down 1, mid..., up 0, up 1; convs k3 "same" (taps t-1, t, t+1) */
    const vox_flow_tblock_w* tblocks;    /* [(4 + mid) * n_blocks] */
    vox_conv_w down_s2;                  /* Downsample1D conv (k3, stride 2, pad 1) as ONE tap over [x[2t-1] | x[2t] | x[2t+1]] (Cin = 3 C) */
    vox_conv_w down_conv1;               /* k3 "same" */
    vox_conv_w up_tconv;                 /* ConvTranspose1d(4, 2, 1): taps d = -1, 0, 1, N = 2 C, bias_mod = C */
    vox_conv_w up_conv1, final_conv, final_proj;
    const float *final_gn_w, *final_gn_b;
} vox_glmflow_weights;
typedef struct {
    int32_t vocab, dim, mel, mel_padded, spk_dim, enc_layers, enc_heads, enc_ffn, block_size, est_ch, est_heads, est_head_dim, est_blocks,
        est_mid, n_steps, reg_layers, groups;
    float cfg_rate;
} vox_glmflow_config;
typedef struct vox_glmflow vox_glmflow;
int vox_glmflow_create(vox_ctx* ctx, const vox_glmflow_config* cfg, const vox_glmflow_weights* w, int max_batch, int max_T, int max_mel,
                       const float* time_emb, const float* dt, vox_glmflow** out);
void vox_glmflow_destroy(vox_glmflow* m);
/* tokens: device int32 [n][T]; Tm: mel frames per request ((T / 12.5 * 22050 / 256).int() in the reference); embedding: device fp32
 * [n][spk_dim] or NULL (zeros, as GLMAudioDecoder passes); noise: device fp32 [n][mel][Tm] or NULL; mel: fp32 [n][mel][Tm] */
int vox_glmflow_decode(vox_glmflow* m, void* stream, const int32_t* tokens, int n, int T, int Tm, const float* embedding, const float* noise,
                       uint64_t seed, uint32_t first_stream, float* mel);

/* ---- Qwen3-TTS speaker encoder (voice cloning, prompt side) -------------------------------------------------
 * replaces mel_spectrogram + Qwen3TTSSpeakerEncoder.forward as called by Qwen3TTSModel._extract_speaker_embedding
 * (model/qwen3_tts.py:21-88, 835-891, 1288-1328): 24 kHz clip -> reflect-padded Hann STFT (DFT accumulated in fp64) -> log-mel
 * -> ECAPA-TDNN (TDNN, SE-Res2Net blocks with reflect "same" padding, multi-layer aggregation, attentive statistics pooling, fc)
 * -> x-vector.  fp32 activations over the checkpoint's bf16 weights (one weight plane per conv, exact products).           */
typedef struct vox_spkenc vox_spkenc;
typedef struct {
    int32_t n_mels, n_mels_padded, n_fft, hop;   /* n_fft a power of two <= 2048 */
    int32_t n_blocks;                            /* SE-Res2Net blocks (3) */
    int32_t channels, scale, se_channels;        /* 512, 8, 128 */
    int32_t mfa_channels, att_channels, enc_dim; /* 1536 (= n_blocks * channels), 128, 1024 | 2048 */
    int32_t kernel0, dilation0;                  /* first TDNN (5, 1) */
    int32_t kernels[4], dilations[4];            /* Res2Net convs of each block (3; 2, 3, 4) */
} vox_spkenc_config;
typedef struct { vox_conv_w tdnn1, res2[7], tdnn2, se1, se2; } vox_spkenc_block_w;
typedef struct {
    const float* mel_basis; /* [n_mels][n_fft/2 + 1]  (librosa.filters.mel, Slaney) */
    const float* window;    /* [n_fft] periodic Hann */
    vox_conv_w conv0;
    vox_spkenc_block_w blocks[4];
    vox_conv_w mfa, asp_tdnn, asp_conv, fc;
} vox_spkenc_weights;
int vox_spkenc_create(vox_ctx* ctx, const vox_spkenc_config* cfg, const vox_spkenc_weights* w, int max_samples, vox_spkenc** out);
void vox_spkenc_destroy(vox_spkenc* m);
/* audio: device fp32 [n_samples]; mel_out: device fp32 [T][n_mels] or NULL (T = n_samples / hop frames, returned through n_frames);
 * emb_out: device fp32 [enc_dim] */
int vox_spkenc_embed(vox_spkenc* m, void* stream, const float* audio, int n_samples, float* mel_out, int32_t* n_frames, float* emb_out);

/* ---- Qwen3-TTS speech-tokenizer encoder (ICL voice cloning, prompt side) ---------------------------------------
 * replaces Qwen3TTSTokenizerV2Model.encode (tokenizer/qwen3_codec.py:1743-1773: MimiModel.encode of `transformers` behind
 * Qwen3TTSTokenizerV2Encoder, :1669-1679) as called by Qwen3TTSModel._encode_audio_to_codes (model/qwen3_tts.py:1330-1371):
 * 24 kHz clip -> SEANet encoder (causal convs, strided convs as two-tap GEMMs over r-frame rows) -> 8-layer sliding-window
 * transformer -> stride-2 downsample (replicate padding) -> split RVQ encode (nearest centroid per layer) -> codes
 * [ceil(n / 1920)][16].  fp32 activations, one bf16 weight plane (the reference serves the tokenizer in bf16).             */
typedef struct vox_codecenc vox_codecenc;
typedef struct {
    int32_t num_filters, ratios[4], kernel_size, residual_kernel_size, last_kernel_size, compress;
    int32_t hidden, num_heads, head_dim, num_layers, ffn, window;
    int32_t codebook_size, codebook_dim, n_semantic, n_acoustic; /* RVQ layers used: 1 + 15 */
    float ln_eps;
} vox_codecenc_config;
typedef struct { vox_conv_w conv1, conv2, down; } vox_codecenc_stage_w; /* ResnetBlock convs (k3, k1), strided conv as 2 taps over [L/r][r C] */
typedef struct {
    const float *in_w, *in_b;      /* first conv: [num_filters][kernel_size], [num_filters] */
    vox_codecenc_stage_w stage[4];
    vox_conv_w last;
    vox_mimi_layer_w layers[16];   /* qkv = [q; k; v] rows, o, fc1, fc2; LayerNorm + LayerScale vectors */
    const float* inv_freq;         /* [head_dim / 2] */
    vox_conv_w downsample;         /* 2 taps over frame pairs [T/2][2 hidden] */
    vox_conv_w sem_proj, ac_proj;  /* 1x1 input projections hidden -> codebook_dim */
    const float *sem_emb, *ac_emb; /* [layers][codebook_size][codebook_dim] = embed_sum / clamp(cluster_usage, 1e-5) */
} vox_codecenc_weights;
int vox_codecenc_create(vox_ctx* ctx, const vox_codecenc_config* cfg, const vox_codecenc_weights* w, int max_samples, vox_codecenc** out);
void vox_codecenc_destroy(vox_codecenc* m);
/* audio: device fp32 [n_samples]; codes: device int32 [T][n_semantic + n_acoustic], T = ceil(n_samples / hop) (returned through
 * n_frames); latents_out: device fp32 [T][hidden] (the frames that were quantised) or NULL */
int vox_codecenc_encode(vox_codecenc* m, void* stream, const float* audio, int n_samples, int32_t* codes, int32_t* n_frames, float* latents_out);

#ifdef __cplusplus
}
#endif
#endif /* VOXHIP_H */
