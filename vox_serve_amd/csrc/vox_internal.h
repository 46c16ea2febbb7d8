// Internal (non-ABI) declarations shared by the libvoxhip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/voxhip.h"
#include "vox_device.h"

int vox_fail(int code, const char* fmt, ...);

#define VOX_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return vox_fail(VOX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                            __FILE__, __LINE__);                                               \
    } while (0)
#define VOX_TRY(expr)               \
    do {                            \
        int _s = (expr);            \
        if (_s != VOX_OK) return _s; \
    } while (0)

struct vox_ctx {
    int device;
    int n_cu;
    int64_t lds_bytes, hbm_bytes;
    void* samp_ws;   // sampler scratch: SAMP_WS_ROWS x (65536 u32 histogram, all-zero between launches) + keys
    int exact_rows = 2;   // linears with at most this many rows use the wave64 VALU kernels, more rows the matrix cores (vox_ctx_set_exact_rows)
};
// sampler scratch geometry (sampler.hip): one LM stream per context uses it at a time
#define SAMP_WS_ROWS 64
#define SAMP_WS_VMAX 262144
#define SAMP_WS_BYTES ((size_t)SAMP_WS_ROWS * (65536 * 4 + SAMP_WS_VMAX * 2))

struct vox_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};

// ---- kernel launchers (kernels_lm.hip) ----------------------------------------------------------
struct LinearCall {
    const void *W = nullptr, *W2 = nullptr, *bias = nullptr, *x = nullptr, *residual = nullptr, *norm_w = nullptr;
    void *y = nullptr, *x_out = nullptr;
    const float *part_o = nullptr, *part_ml = nullptr;
    const int* kvlen = nullptr;
    const int* x_rows = nullptr;
    long x_stride = 0, x_out_stride = 0;  // elements; 0 = K
    float eps = 1e-6f;
    int B = 0, N = 0, K = 0, Hq = 0, D = 0, max_chunks = 0;
    void* splitk_ws = nullptr;      // fp32 workspace of the > 32-row split-K GEMM: >= ceil(K/512) * 128 * Ntot * 4 bytes
    size_t splitk_ws_bytes = 0;
    // 17+ rows path only: the epilogue also writes RMSNorm(y) * post_norm_w (eps = this call's eps) to post_norm_out, and a
    // later call passes that buffer as x_prenormed instead of re-normalising
    const void* post_norm_w = nullptr;
    void* post_norm_out = nullptr;
    const void* x_prenormed = nullptr;
    void* norm_scratch = nullptr;   // [B,K] bf16: lets > 32-row calls with a norm prologue normalise once up front
    int keep_weights = 0;   // weights are re-read within the frame (depth loop): do not stream them past the caches
    // 9..32 rows path (full-K MFMA GEMM) only — "fragment-major" operands: tile (t, k-step ks) of a [rows or cols][K] matrix
    // is the 1 KiB block [64 lanes][8 bf16] an MFMA 16x16x32 operand register holds (lane = (row % 16) + 16 * ((k % 32) / 8)),
    // at element offset ((t * (K / 32) + ks) * 512): every operand load is one contiguous 1 KiB wave request.
    const void *W_frag = nullptr, *W2_frag = nullptr;   // pre-swizzled copies of W / W2 (vox_launch_swizzle_frag)
    const void* x_frag = nullptr;                        // the input in fragment-major form (replaces x as the A operand)
    void* y_frag = nullptr;                              // also write the output fragment-major (the next linear's x_frag)
    int y_rowmajor = 1;                                  // 0: skip the row-major y (only y_frag is consumed)
    int pro = 0, epi = 0;  // PRO_* / EPI_*
    int fixed_order = 0;   // 1: keep the fixed-order VALU kernel even above exact_rows rows (depth step 1 of a small-batch frame)
    int exact_rows = 2;    // rows up to which the wave64 VALU kernels are used (the context's setting; vox_launch_linear fills it)
};
enum { VOX_PRO_COPY = 0, VOX_PRO_RMSNORM = 1, VOX_PRO_ATTN = 2 };
enum { VOX_EPI_STORE = 0, VOX_EPI_SILU = 1, VOX_EPI_SILU_MUL = 2 };
bool vox_linear_is_rows_gemm(const LinearCall& c);
bool vox_linear_is_fullk(const LinearCall& c);      // true: the call takes the 9..32 rows path that honours *_frag
bool vox_fullk_weight_ok(int N, int K);
bool vox_stream_weight_ok(int N, int K);             // a weight of this shape can be used fragment-major
int vox_launch_swizzle_frag(hipStream_t st, const void* src, void* dst, int rows, int K);   // rows % 16 == 0, K % 32 == 0
int vox_launch_linear(vox_ctx* ctx, hipStream_t st, const LinearCall& c);
int vox_launch_rmsnorm(hipStream_t st, const void* x, const void* w, void* y, int rows, int H, float eps);

struct HeadCall {
    const void *q_src = nullptr, *k_src = nullptr, *v_src = nullptr;
    long q_stride = 0, k_stride = 0, v_stride = 0;
    void *q_out = nullptr, *k_out = nullptr, *kv = nullptr;
    const void *qn = nullptr, *kn = nullptr;
    const float* cs = nullptr;
    const int *pos = nullptr, *page = nullptr, *slot = nullptr;
    float eps = 1e-6f;
    int N = 0, Hq = 0, Hkv = 0, D = 0, rot = 0, interleave = 0, page_size = 0, table_max_pos = 0;
};
int vox_launch_head_prepare(hipStream_t st, const HeadCall& c);

struct AttnCall {
    const void *q = nullptr, *kv = nullptr;
    const int *q_req = nullptr, *q_kvlen = nullptr, *indptr = nullptr, *indices = nullptr;
    float *part_o = nullptr, *part_ml = nullptr;
    float scale = 1.0f;
    int Nq = 0, Hq = 0, Hkv = 0, D = 0, page_size = 0, max_chunks = 0, max_kvlen = 0;
    // fused decode mode (qkv != NULL): per-head norm + RoPE + KV append of each row's own token inside the kernel
    const void *qkv = nullptr, *qn = nullptr, *kn = nullptr;
    const float* cs = nullptr;
    const int *pos = nullptr, *page = nullptr, *slot = nullptr;
    float eps = 1e-6f;
    int rot = 0, interleave = 0, table_max_pos = 0;
    void* out = nullptr;   // bf16 [Nq,Hq,D]: written directly when the launch covers a single chunk
    void* out_frag = nullptr;   // optional: the same output ALSO in fragment-major form (A operand of a 9..32 rows o_proj)
    const int* ptab = nullptr;
    int pt_stride = 0, fixed_kvlen = 0, fixed_pos = -1, identity_pages = 0;
};
// Persistent depth step (one request): 5 layers + the codebook head of one depth-loop step in one launch (kernels_lm.hip).
struct DepthStepCall {
    const void* layers_dev = nullptr;      // device array of n_layers x 9 pointers {wqkv, wo, wgate, wup, wdown, ln1, ln2, qnorm, knorm}
    int n_layers = 0;
    const void *final_norm = nullptr, *head_w = nullptr, *x_in = nullptr;
    void* logits = nullptr;
    void* gran = nullptr;                  // 4096 granules of 8 bytes (x 512, qkv 2048, h 1536), zeroed at creation
    unsigned *epoch = nullptr, *err = nullptr;
    void* kv = nullptr;                    // depth KV cache of layer 0; row 0 owns page 0
    long kv_layer_stride = 0;              // elements
    const float* cs = nullptr;
    float eps = 1e-6f, scale = 1.0f;
    int hidden = 0, heads = 0, kv_heads = 0, head_dim = 0, ffn = 0, vocab = 0, qk_norm = 0, qkv_bias = 0, rope_dim = 0,
        rope_interleave = 0, page_size = 0, table_max_pos = 0;
    int n_tokens = 0;                      // visible tokens of the step (position n_tokens - 1)
    // Greedy pick of the PREVIOUS step's codebook inside this launch (pick_logits != NULL; instead of a sampler launch in between):
    // every block takes the first maximum of pick_logits[pick_vocab] and reads its input row pick_tab[id][hidden]; block 0 also
    // writes the id to *pick_out and adds (pick_init: stores) pick_emb[id][pick_H] to pick_feat — what the sampler launch did.  x_in is unused.
    const void *pick_logits = nullptr, *pick_tab = nullptr, *pick_emb = nullptr;
    int* pick_out = nullptr;
    void* pick_feat = nullptr;
    int pick_vocab = 0, pick_H = 0, pick_init = 0;
    // ... or the SAMPLED pick (pick_top_k > 0; vocabulary of exactly 2048 entries): top-k / top-p / min-p at a temperature with the draw
    // philox(seed, pick_offset + *pick_offset_dev * pick_offset_mul, row 0) — k_sample_topk's contract, every block for itself
    int pick_top_k = 0;
    float pick_top_p = 1.0f, pick_min_p = 0.0f, pick_temperature = 1.0f;
    uint64_t pick_seed = 0, pick_offset = 0, pick_offset_mul = 0;
    const uint64_t* pick_offset_dev = nullptr;
};
// Persistent MLP half of a talker layer at one row (o_proj + residual, gate/up, down + residual in one launch; kernels_lm.hip)
struct TalkerMlpCall {
    const void *wo = nullptr, *wgate = nullptr, *wup = nullptr, *wdown = nullptr, *ln2 = nullptr, *attn = nullptr;
    void* x = nullptr;                     // the residual row, updated in place
    void* gran = nullptr;                  // 4096 granules of 8 bytes (x' 1024, h 3072), zeroed at creation
    unsigned *epoch = nullptr, *err = nullptr;
    float eps = 1e-6f;
    int hidden = 0, nq = 0, ffn = 0;
    // optional fourth stage: the NEXT layer's input_layernorm + q/k/v projection (nqkv = 4096 outputs, plain row for the attention launch)
    const void *wqkv_next = nullptr, *ln1_next = nullptr;
    void* qkv_out = nullptr;
    int nqkv = 0;
    // optional stage in front: the layer's decode attention inside the launch (blocks 0..15; gran then holds 5120 granules, the last
    // 1024 for the attention row) — `attn` is unused
    const struct AttnCall* attn_call = nullptr;
    // optional (with attn_call): EVERY layer of the stack in this one launch.  layer_tab: device array of n_layers rows of nine pointers
    // {wo, wgate, wup, wdown, ln2, wqkv, ln1, qnorm, knorm}; the layers' K / V caches lie kv_layer_stride elements apart from
    // attn_call->kv; attn_call->qkv holds layer 0's projection; gran then holds 7168 granules.  The per-layer weight fields above are unused.
    const void* layer_tab = nullptr;
    int n_layers = 0;
    long kv_layer_stride = 0;
};
bool vox_talker_mlp_supported(const TalkerMlpCall& c);
bool vox_talker_attn_supported(const struct AttnCall& c);
int vox_launch_talker_mlp(hipStream_t st, const TalkerMlpCall& c);
bool vox_depth_step_supported(const DepthStepCall& c);
int vox_launch_depth_step(hipStream_t st, const DepthStepCall& c);
int vox_launch_attn_partial(hipStream_t st, const AttnCall& c);
bool vox_attn_decode8_supported(const AttnCall& c);      // decode rows, 2..8 chunks: every chunk + the merge in one launch
int vox_launch_attn_decode8(hipStream_t st, const AttnCall& c);
bool vox_attn_short_supported(const AttnCall& c);
int vox_launch_attn_short(hipStream_t st, const AttnCall& c);
bool vox_attn1_linear_supported(const AttnCall& c, const struct LinearCall& l);
int vox_launch_attn1_linear(hipStream_t st, const AttnCall& c, const struct LinearCall& l);
int vox_launch_attn_merge(hipStream_t st, const float* part_o, const float* part_ml, const int* kvlen, void* out,
                          int Nq, int Hq, int D, int max_chunks, void* out_frag = nullptr);
int vox_launch_gather(hipStream_t st, const void* table, const int* ids, int id_stride, int id_off, void* dst,
                      long dst_stride, int B, int H, int vocab);
// shadow of a decode frame's inputs, two slots (engine.hip: vox_qwen3_set_status / vox_qwen3_frame_restore)
struct Qwen3Shadow {
    uint64_t* rng = nullptr;     // [2]
    int32_t* ids = nullptr;      // [2][max_batch][G1]
    uint8_t* mask = nullptr;     // [2][max_batch]
    bf16_t* feat = nullptr;      // [2][max_batch][H]
    int G1 = 0, H = 0, max_batch = 0;
};
int vox_launch_qwen3_mix(hipStream_t st, const void* text, const void* codec_table, const int* ids, int id_stride,
                         const uint8_t* mask, const void* feat, void* y, int B, int H, int vocab,
                         const Qwen3Shadow* shadow = nullptr, const uint64_t* rng = nullptr);
int vox_launch_kv_append(hipStream_t st, void* kv, const void* k, const void* v, const int* page, const int* slot,
                         int N, int page_size, int Hkv, int D);

// ---- sampler (sampler.hip) ------------------------------------------------------------------------
struct SampleCall {
    void* logits = nullptr;            // bf16 [B,V] (row stride V), modified in place by suppress/penalty
    int B = 0, V = 0;
    const int* suppress_ids = nullptr; // device
    int n_suppress = 0;
    const uint8_t* rep_cache = nullptr; // [B,W,C,V]
    int W = 0, C = 0;
    vox_sampling_config cfg{};
    uint64_t seed = 0, offset = 0;
    const uint64_t* offset_dev = nullptr;  // optional device counter added to offset
    uint64_t offset_mul = 1;               // effective offset = offset + (*offset_dev) * offset_mul
    int* out_ids = nullptr;            // out_ids[b*out_stride + out_col]
    int out_stride = 1, out_col = 0;
    // fused tail: embedding of the sampled id
    const void* emb_table = nullptr;   // [emb_vocab, H] bf16
    int emb_vocab = 0, H = 0;
    void* emb_dst = nullptr;           // row b at emb_dst + b*emb_dst_stride (elements)
    long emb_dst_stride = 0;
    // second gather of the same id from a table of pre-projected rows (the depth loop's next input: projection(embedding[id])
    // is a pure function of id, tabulated once at engine creation with the same fixed-order kernel)
    const void* emb2_table = nullptr;  // [emb_vocab, H2] bf16
    int H2 = 0;
    void* emb2_dst = nullptr;          // row b at emb2_dst + b*emb2_dst_stride (elements)
    long emb2_dst_stride = 0;
    void* feat_acc = nullptr;          // [B,H] bf16: feat = bf16(feat + emb) (qwen3_tts.py:2002)
    int feat_init = 0;                 // 1: feat = emb' where emb' = bf16(0 + emb)
    void* ws = nullptr;                // ctx->samp_ws (needed by top-p/min-p-only modes and V > 32768)
};
int vox_launch_sample(hipStream_t st, const SampleCall& c);
int vox_launch_suppress(hipStream_t st, void* logits, int B, int V, const int* ids, int n);
int vox_launch_rep_penalty(hipStream_t st, void* logits, const uint8_t* cache, int B, int W, int C, int V, float p);
int vox_launch_rep_update(hipStream_t st, uint8_t* cache, const int* ids, int B, int W, int C, int V, int window);
int vox_launch_rep_penalty_mc(hipStream_t st, void* logits, const uint8_t* cache, int B, int Cl, int W, int C, int V, float p);
int vox_launch_rep_update_mc(hipStream_t st, uint8_t* cache, const int* ids, int B, int Cl, int W, int C, int V, int window);
