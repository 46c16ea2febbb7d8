// Host-side engines of libvoxhip: context / hipGraph capture, the generic decoder stack
// (vox_stack_*) and the Qwen3-TTS frame engine (vox_qwen3_*): talker step + on-device sampling +
// the whole 15-step depth loop enqueued on one stream with no host round trip, so one hipGraph holds
// an entire audio frame (the reference needs 16 graph replays, 16 plan() syncs and >=16*B .item() syncs
// per frame: worker/cuda_graph_worker.py:946-1160, model/qwen3_tts.py:1935-1997).
#include <stdarg.h>
#include <string.h>

#include <vector>

#include "vox_internal.h"

static thread_local char g_err[512] = "";

int vox_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" {

int vox_abi_version(void) { return VOX_ABI_VERSION; }
const char* vox_last_error(void) { return g_err; }

int vox_ctx_create(int device, vox_ctx** out) {
    if (!out) return vox_fail(VOX_ERR_INVALID, "ctx_create: out is NULL");
    int n = 0;
    VOX_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return vox_fail(VOX_ERR_INVALID, "ctx_create: device %d of %d", device, n);
    VOX_HIP(hipSetDevice(device));
    hipDeviceProp_t p;
    VOX_HIP(hipGetDeviceProperties(&p, device));
    vox_ctx* c = new vox_ctx();
    c->device = device;
    c->n_cu = p.multiProcessorCount;
    c->lds_bytes = (int64_t)p.maxSharedMemoryPerMultiProcessor;
    c->hbm_bytes = (int64_t)p.totalGlobalMem;
    c->samp_ws = nullptr;
    if (hipMalloc(&c->samp_ws, SAMP_WS_BYTES) != hipSuccess || hipMemset(c->samp_ws, 0, SAMP_WS_BYTES) != hipSuccess) {
        delete c;
        return vox_fail(VOX_ERR_NOMEM, "ctx_create: sampler scratch");
    }
    *out = c;
    return VOX_OK;
}
void vox_ctx_destroy(vox_ctx* ctx) {
    if (!ctx) return;
    (void)hipFree(ctx->samp_ws);
    delete ctx;
}
int vox_ctx_set_exact_rows(vox_ctx* ctx, int rows) {
    if (!ctx || rows < 1 || rows > 8) return vox_fail(VOX_ERR_INVALID, "ctx_set_exact_rows: rows must be 1..8");
    ctx->exact_rows = rows;
    return VOX_OK;
}
int vox_ctx_props(vox_ctx* ctx, int64_t out[3]) {
    if (!ctx || !out) return vox_fail(VOX_ERR_INVALID, "ctx_props: NULL");
    out[0] = ctx->n_cu; out[1] = ctx->lds_bytes; out[2] = ctx->hbm_bytes;
    return VOX_OK;
}

int vox_graph_begin(vox_ctx* ctx, void* stream) {
    (void)ctx;
    VOX_HIP(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    return VOX_OK;
}
int vox_graph_end(vox_ctx* ctx, void* stream, vox_graph** out) {
    (void)ctx;
    hipGraph_t g = nullptr;
    VOX_HIP(hipStreamEndCapture((hipStream_t)stream, &g));
#ifdef VOX_DEV_KNOBS
    if (getenv("VOX_GRAPH_INFO")) {     // (dev knob: node / edge counts and node types of the captured graph)
        size_t nn = 0, ne = 0;
        (void)hipGraphGetNodes(g, nullptr, &nn); (void)hipGraphGetEdges(g, nullptr, nullptr, &ne);
        std::vector<hipGraphNode_t> nodes(nn);
        (void)hipGraphGetNodes(g, nodes.data(), &nn);
        int types[16] = {0};
        for (auto n : nodes) { hipGraphNodeType t; if (hipGraphNodeGetType(n, &t) == hipSuccess && (int)t < 16) types[(int)t]++; }
        fprintf(stderr, "[vox graph] %zu nodes, %zu edges; types:", nn, ne);
        for (int t = 0; t < 16; ++t) if (types[t]) fprintf(stderr, " %d:%d", t, types[t]);
        fprintf(stderr, "\n");
        if (const char* dot = getenv("VOX_GRAPH_DOT")) (void)hipGraphDebugDotPrint(g, dot, 0);
    }
#endif
    hipGraphExec_t e = nullptr;
    hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
    if (err != hipSuccess) {
        (void)hipGraphDestroy(g);
        return vox_fail(VOX_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(err));
    }
    vox_graph* vg = new vox_graph();
    vg->graph = g;
    vg->exec = e;
    *out = vg;
    return VOX_OK;
}
int vox_graph_launch(vox_graph* g, void* stream) {
    if (!g) return vox_fail(VOX_ERR_INVALID, "graph_launch: NULL");
    VOX_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return VOX_OK;
}
void vox_graph_destroy(vox_graph* g) {
    if (!g) return;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
}

// ---------------------------------------------------------------------------------------------------
// per-op ABI
// ---------------------------------------------------------------------------------------------------
int vox_rmsnorm(vox_ctx* ctx, void* stream, const void* x, const void* w, void* y, int rows, int cols, float eps) {
    (void)ctx;
    return vox_launch_rmsnorm((hipStream_t)stream, x, w, y, rows, cols, eps);
}

int vox_rope_table_host(float* cs, int max_pos, int rot, double theta, double scale, int llama31, double lo,
                        double hi, int old_ctx) {
    if (!cs || rot <= 0 || rot % 2) return vox_fail(VOX_ERR_INVALID, "rope_table: bad args");
    const int half = rot / 2;
    for (int i = 0; i < half; ++i) {
        double f = 1.0 / pow(theta, (double)(2 * i) / (double)rot);
        if (llama31) {
            double smooth = (f * old_ctx / (2.0 * M_PI) - lo) / (hi - lo);
            smooth = smooth < 0 ? 0 : (smooth > 1 ? 1 : smooth);
            f = (1.0 - smooth) * (f / scale) + smooth * f;
        } else {
            f = f / scale;
        }
        const float ff = (float)f;
        for (int p = 0; p < max_pos; ++p) {
            const float ang = (float)p * ff;
            cs[((size_t)p * half + i) * 2 + 0] = (float)cos((double)ang);
            cs[((size_t)p * half + i) * 2 + 1] = (float)sin((double)ang);
        }
    }
    return VOX_OK;
}

int vox_rope(vox_ctx* ctx, void* stream, const void* q, const void* k, void* q_out, void* k_out, const int32_t* pos,
             int N, int Hq, int Hkv, int D, int rot, int interleave, const float* cs, int table_max_pos) {
    (void)ctx;
    HeadCall c;
    c.q_src = q; c.k_src = k; c.q_stride = (long)Hq * D; c.k_stride = (long)Hkv * D;
    c.q_out = q_out; c.k_out = k_out; c.cs = cs; c.pos = pos; c.N = N; c.Hq = Hq; c.Hkv = Hkv; c.D = D;
    c.rot = rot; c.interleave = interleave; c.table_max_pos = table_max_pos; c.page_size = 1;
    return vox_launch_head_prepare((hipStream_t)stream, c);
}

int vox_kv_append(vox_ctx* ctx, void* stream, void* kv, const void* k, const void* v, const int32_t* page,
                  const int32_t* slot, int N, int page_size, int Hkv, int D) {
    (void)ctx;
    return vox_launch_kv_append((hipStream_t)stream, kv, k, v, page, slot, N, page_size, Hkv, D);
}

static inline int n_chunks(int kvlen) { return kvlen <= 0 ? 1 : (kvlen + VOX_ATTN_CHUNK - 1) / VOX_ATTN_CHUNK; }

int64_t vox_attn_workspace_bytes(int Nq, int Hq, int D, int max_kvlen) {
    return (int64_t)Nq * Hq * n_chunks(max_kvlen) * (D + 2) * 4;
}

int vox_paged_attention(vox_ctx* ctx, void* stream, const void* q, const void* kv, const int32_t* q_req,
                        const int32_t* q_kvlen, const int32_t* indptr, const int32_t* indices, void* out, void* ws,
                        int Nq, int Hq, int Hkv, int D, int page_size, int max_kvlen, float scale) {
    (void)ctx;
    const int mc = n_chunks(max_kvlen);
    AttnCall c;
    c.q = q; c.kv = kv; c.q_req = q_req; c.q_kvlen = q_kvlen; c.indptr = indptr; c.indices = indices;
    c.part_o = (float*)ws; c.part_ml = (float*)ws + (size_t)Nq * Hq * mc * D; c.scale = scale; c.Nq = Nq;
    c.Hq = Hq; c.Hkv = Hkv; c.D = D; c.page_size = page_size; c.max_chunks = mc; c.max_kvlen = max_kvlen;
    VOX_TRY(vox_launch_attn_partial((hipStream_t)stream, c));
    return vox_launch_attn_merge((hipStream_t)stream, c.part_o, c.part_ml, q_kvlen, out, Nq, Hq, D, mc);
}

int vox_linear(vox_ctx* ctx, void* stream, const void* W, const void* bias, const void* x, const void* residual,
               void* y, int B, int N, int K, int act) {
    LinearCall c;
    c.W = W; c.bias = bias; c.x = x; c.residual = residual; c.y = y; c.B = B; c.N = N; c.K = K;
    c.pro = VOX_PRO_COPY; c.epi = act ? VOX_EPI_SILU : VOX_EPI_STORE;
    return vox_launch_linear(ctx, (hipStream_t)stream, c);
}
int vox_linear_silu_mul(vox_ctx* ctx, void* stream, const void* Wg, const void* Wu, const void* x, void* h, int B,
                        int N, int K) {
    LinearCall c;
    c.W = Wg; c.W2 = Wu; c.x = x; c.y = h; c.B = B; c.N = N; c.K = K;
    c.pro = VOX_PRO_COPY; c.epi = VOX_EPI_SILU_MUL;
    return vox_launch_linear(ctx, (hipStream_t)stream, c);
}

int vox_suppress(vox_ctx* ctx, void* stream, void* logits, int B, int V, const int32_t* ids, int n) {
    (void)ctx;
    return vox_launch_suppress((hipStream_t)stream, logits, B, V, ids, n);
}
int vox_rep_penalty(vox_ctx* ctx, void* stream, void* logits, const uint8_t* cache, int B, int W, int C, int V,
                    float penalty) {
    (void)ctx;
    return vox_launch_rep_penalty((hipStream_t)stream, logits, cache, B, W, C, V, penalty);
}
int vox_rep_update(vox_ctx* ctx, void* stream, uint8_t* cache, const int32_t* ids, int B, int W, int C, int V,
                   int window) {
    (void)ctx;
    return vox_launch_rep_update((hipStream_t)stream, cache, ids, B, W, C, V, window);
}
int vox_rep_penalty_mc(vox_ctx* ctx, void* stream, void* logits, const uint8_t* cache, int B, int Cl, int W, int C, int V,
                       float penalty) {
    (void)ctx;
    return vox_launch_rep_penalty_mc((hipStream_t)stream, logits, cache, B, Cl, W, C, V, penalty);
}
int vox_rep_update_mc(vox_ctx* ctx, void* stream, uint8_t* cache, const int32_t* ids, int B, int Cl, int W, int C, int V,
                      int window) {
    (void)ctx;
    return vox_launch_rep_update_mc((hipStream_t)stream, cache, ids, B, Cl, W, C, V, window);
}
int vox_sample(vox_ctx* ctx, void* stream, const void* logits, int B, int V, const vox_sampling_config* cfg,
               uint64_t seed, uint64_t offset, int32_t* out_ids) {
    if (!cfg) return vox_fail(VOX_ERR_INVALID, "sample: cfg NULL");
    SampleCall c;
    c.ws = ctx ? ctx->samp_ws : nullptr;
    c.logits = const_cast<void*>(logits); c.B = B; c.V = V; c.cfg = *cfg; c.cfg.repetition_penalty = 1.0f;
    c.seed = seed; c.offset = offset; c.out_ids = out_ids;
    return vox_launch_sample((hipStream_t)stream, c);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// decoder stack
// ---------------------------------------------------------------------------------------------------
struct vox_stack {
    vox_ctx* ctx;
    vox_stack_config cfg;
    std::vector<vox_layer_weights> layers;
    const void* final_norm;
    const float* rope;
    int rope_max_pos;
    // workspace
    void *qkv, *q, *h, *attn_out, *xn, *skws;
    size_t skws_bytes;
    // 9..32 rows path: fragment-major copies of the layer weights (NULL where the shape is not eligible) and activations
    struct FragW { void *qkv = nullptr, *o = nullptr, *gate = nullptr, *up = nullptr, *down = nullptr; };
    std::vector<FragW> fw;
    void *xfrag = nullptr, *hfrag = nullptr, *afrag = nullptr;
    float* attn_ws;
    size_t attn_ws_floats;
    int keep_weights = 0;   // the stack runs many times per frame (depth loop): keep its weights cache-resident
    // persistent MLP half of a one-row decode layer (k_talker_mlp): hand-off granules, epoch / error words
    void* mlp_gran = nullptr;
    unsigned* mlp_words = nullptr;
    int mlp_persist = 0;
    int mlp_attn = 0;       // ... and the layer's decode attention inside that launch (VOX_TALKER_ATTN=0: its own launch in front)
    void* mlp_tab = nullptr;      // ... and every layer in ONE launch: the device table of the layers' weight pointers (VOX_TALKER_MULTI=0: a launch per layer)
};

// decode_rows: every row is the newest token of a distinct request (its K/V are not read by any other row), so the
// head norm / RoPE / KV append fuse into the attention kernel; otherwise (prefill) a separate pass appends first.
// Timing ablation for development builds only (-DVOX_DEV_KNOBS: VOX_ABLATE bitmask skips parts of the frame, results are
// wrong when set): 1 attention, 2 depth loop, 4 talker layers, 8 samplers, 16 qkv, 32 o_proj, 64 gate/up, 128 down,
// 256 / 512 older attention variants.  The shipped library compiles this to a constant 0.
#ifdef VOX_DEV_KNOBS
static int ablate() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("VOX_ABLATE"); v = e ? atoi(e) : 0; }
    return v;
}
#else
static constexpr int ablate() { return 0; }
#endif

// bound of the persistent kernels' poll loops (passes; a pass is a round trip to the memory side, >= ~0.5 us: >= 20 ms by default);
// VOX_PERSIST_SPINS in the environment overrides it at engine creation, vox_qwen3_persist_set_spins() later
static unsigned vox_persist_spins_default() {
    static const unsigned v = [] { const char* e = getenv("VOX_PERSIST_SPINS"); const long x = e ? atol(e) : 0; return x > 0 ? (unsigned)x : 40000u; }();
    return v;
}

static int stack_layers(vox_stack* s, hipStream_t st, void* x, void* kv, int64_t kv_stride, const vox_rows* r,
                        bool decode_rows = false, int fixed_order = 0) {
    const vox_stack_config& c = s->cfg;
    const int n = r->n_rows;
    if (n <= 0) return VOX_OK;
    if (n > c.max_rows) return vox_fail(VOX_ERR_INVALID, "stack_forward: %d rows > max_rows %d", n, c.max_rows);
    const int nq = c.heads * c.head_dim, nkv = c.kv_heads * c.head_dim, nqkv = nq + 2 * nkv;
    const int mc = n_chunks(r->max_kvlen);
    const size_t need = (size_t)n * c.heads * mc * (c.head_dim + 2);
    if (need > s->attn_ws_floats)
        return vox_fail(VOX_ERR_INVALID, "stack_forward: attention workspace too small (%zu > %zu floats)", need,
                        s->attn_ws_floats);
    float* part_o = s->attn_ws;
    float* part_ml = s->attn_ws + (size_t)n * c.heads * mc * c.head_dim;
    const float scale = 1.0f / sqrtf((float)c.head_dim);
    bool xn_ready = false;   // s->xn holds norm(x) for the next norm-prologue linear (written by the producing GEMM's reduce)
    bool xf_ready = false;   // s->xfrag holds x in fragment-major form (written by the producing full-K GEMM's epilogue)
    bool qkv_done = false;   // s->qkv already holds this layer's projection (fourth stage of the previous layer's k_talker_mlp)
    for (int l = 0; l < c.layers; ++l) {
        const vox_layer_weights& w = s->layers[l];
        void* kvl = (char*)kv + (size_t)l * kv_stride * 2;
        LinearCall a;  // input_layernorm + fused q/k/v projection
        a.W = w.wqkv; a.bias = w.bqkv; a.x = x; a.norm_w = w.ln1; a.eps = c.eps; a.y = s->qkv;
        a.B = n; a.N = nqkv; a.K = c.hidden; a.pro = VOX_PRO_RMSNORM; a.epi = VOX_EPI_STORE;
        a.fixed_order = fixed_order; a.exact_rows = s->ctx->exact_rows; a.keep_weights = s->keep_weights; a.splitk_ws = s->skws; a.splitk_ws_bytes = s->skws_bytes; a.norm_scratch = s->xn;
        if (xn_ready) a.x_prenormed = s->xn;
        const vox_stack::FragW fwl = s->fw.empty() ? vox_stack::FragW{} : s->fw[l];
        a.W_frag = fwl.qkv;
        if (xf_ready && vox_linear_is_fullk(a)) a.x_frag = s->xfrag;
        if (qkv_done) qkv_done = false;      // the previous layer's persistent launch already produced this layer's q | k | v
        else if (!(ablate() & 16)) VOX_TRY(vox_launch_linear(s->ctx, st, a));
        HeadCall hc;  // per-head norm + RoPE + paged append
        hc.q_src = s->qkv; hc.k_src = (bf16_t*)s->qkv + nq; hc.v_src = (bf16_t*)s->qkv + nq + nkv;
        hc.q_stride = hc.k_stride = hc.v_stride = nqkv;
        hc.q_out = s->q; hc.kv = kvl; hc.qn = c.qk_norm ? w.qnorm : nullptr; hc.kn = c.qk_norm ? w.knorm : nullptr;
        hc.cs = s->rope; hc.pos = r->pos; hc.page = r->page; hc.slot = r->slot; hc.eps = c.eps; hc.N = n;
        hc.Hq = c.heads; hc.Hkv = c.kv_heads; hc.D = c.head_dim; hc.rot = c.rope_dim; hc.interleave = c.rope_interleave;
        hc.page_size = c.page_size; hc.table_max_pos = s->rope_max_pos;
        if (!decode_rows) VOX_TRY(vox_launch_head_prepare(st, hc));
        AttnCall ac;
        if (decode_rows) {
            ac.qkv = s->qkv; ac.qn = hc.qn; ac.kn = hc.kn; ac.cs = s->rope; ac.pos = r->pos; ac.page = r->page;
            ac.slot = r->slot; ac.eps = c.eps; ac.rot = c.rope_dim; ac.interleave = c.rope_interleave;
            ac.table_max_pos = s->rope_max_pos;
            ac.ptab = r->page_table; ac.pt_stride = r->pt_stride; ac.identity_pages = r->identity_pages;
            if (r->fixed_kvlen > 0) { ac.fixed_kvlen = r->fixed_kvlen; ac.fixed_pos = r->fixed_pos; }
        }
        ac.q = s->q; ac.kv = kvl; ac.q_req = r->q_req; ac.q_kvlen = r->q_kvlen; ac.indptr = r->kv_indptr;
        ac.indices = r->kv_indices; ac.part_o = part_o; ac.part_ml = part_ml; ac.scale = scale; ac.Nq = n;
        ac.Hq = c.heads; ac.Hkv = c.kv_heads; ac.D = c.head_dim; ac.page_size = c.page_size; ac.max_chunks = mc;
        ac.max_kvlen = r->max_kvlen;
        ac.out = s->attn_out;
        LinearCall o;  // o_proj + residual
        o.W = w.wo; o.x = s->attn_out; o.residual = x; o.y = x; o.B = n; o.N = c.hidden; o.K = nq;
        o.pro = VOX_PRO_COPY; o.epi = VOX_EPI_STORE; o.eps = c.eps;      // (eps of the fused post-norm the split-K reduce may run)
        o.fixed_order = fixed_order; o.exact_rows = s->ctx->exact_rows; o.keep_weights = s->keep_weights; o.splitk_ws = s->skws; o.splitk_ws_bytes = s->skws_bytes;
        xn_ready = vox_linear_is_rows_gemm(o);      // its reduce also writes post_attention_layernorm(x) for gate/up
        if (xn_ready) { o.post_norm_w = w.ln2; o.post_norm_out = s->xn; }
        LinearCall g;  // post_attention_layernorm + gate/up + SiLU*up
        g.W = w.wgate; g.W2 = w.wup; g.x = x; g.norm_w = w.ln2; g.eps = c.eps; g.y = s->h;
        g.B = n; g.N = c.ffn; g.K = c.hidden; g.pro = VOX_PRO_RMSNORM; g.epi = VOX_EPI_SILU_MUL;
        g.fixed_order = fixed_order; g.exact_rows = s->ctx->exact_rows; g.keep_weights = s->keep_weights; g.splitk_ws = s->skws; g.splitk_ws_bytes = s->skws_bytes; g.norm_scratch = s->xn;
        LinearCall d;  // down + residual
        d.W = w.wdown; d.x = s->h; d.residual = x; d.y = x; d.B = n; d.N = c.hidden; d.K = c.ffn;
        d.pro = VOX_PRO_COPY; d.epi = VOX_EPI_STORE; d.eps = c.eps;
        d.fixed_order = fixed_order; d.exact_rows = s->ctx->exact_rows; d.keep_weights = s->keep_weights; d.splitk_ws = s->skws; d.splitk_ws_bytes = s->skws_bytes;
        // fragment-major hand-offs between consecutive full-K GEMMs (o -> gate/up -> down -> next layer's qkv)
        const bool o_fk = s->xfrag && vox_linear_is_fullk(o), g_fk = s->xfrag && vox_linear_is_fullk(g), d_fk = s->xfrag && vox_linear_is_fullk(d);
        o.W_frag = fwl.o; g.W_frag = fwl.gate; g.W2_frag = fwl.up; d.W_frag = fwl.down;
        if (!fwl.gate || !fwl.up) g.W_frag = g.W2_frag = nullptr;
        if (o_fk && !(decode_rows && c.max_kvlen <= 16 && vox_attn1_linear_supported(ac, o))) { ac.out_frag = s->afrag; o.x_frag = s->afrag; }
        xf_ready = o_fk && g_fk;
        if (xf_ready) { o.y_frag = s->xfrag; g.x_frag = s->xfrag; }
        if (g_fk && d_fk) { g.y_frag = s->hfrag; g.y_rowmajor = 0; d.x_frag = s->hfrag; }
        if (decode_rows && c.max_kvlen <= 16 && !(ablate() & 256) && vox_attn1_linear_supported(ac, o)) {
            VOX_TRY(vox_launch_attn1_linear(st, ac, o));     // short context: attention recomputed inside o_proj
        } else {
            // one request, <= 256 visible tokens: the attention runs INSIDE the persistent launch of the layer's MLP half (blocks 0..15)
            const bool attn_in = decode_rows && n == 1 && s->mlp_persist && s->mlp_attn && !ablate() && vox_talker_attn_supported(ac);
            if ((ablate() & 1) || attn_in) {
            } else if (decode_rows && c.max_kvlen <= 32 && !(ablate() & 512) && vox_attn_short_supported(ac)) {
                VOX_TRY(vox_launch_attn_short(st, ac));      // one wave per (row, kv head), registers only
            } else if (decode_rows && !(ablate() & 1024) && vox_attn_decode8_supported(ac)) {
                VOX_TRY(vox_launch_attn_decode8(st, ac));    // <= 256 visible tokens: every chunk and the merge in one launch
            } else {
                VOX_TRY(vox_launch_attn_partial(st, ac));
                if (mc > 1) VOX_TRY(vox_launch_attn_merge(st, part_o, part_ml, r->q_kvlen, s->attn_out, n, c.heads, c.head_dim, mc, ac.out_frag));
            }
            if (attn_in && s->mlp_tab && l == 0) {
                // ... and all layers in that one launch: the next layer's q | k | v reaches its attention blocks as hand-off granules
                TalkerMlpCall tm;
                tm.x = x; tm.gran = s->mlp_gran; tm.epoch = s->mlp_words; tm.err = s->mlp_words + 1; tm.eps = c.eps;
                tm.hidden = c.hidden; tm.nq = nq; tm.ffn = c.ffn; tm.nqkv = nqkv;
                tm.attn_call = &ac; tm.layer_tab = s->mlp_tab; tm.n_layers = c.layers; tm.kv_layer_stride = (long)kv_stride;      // (elements: the layers lie kv_stride * 2 BYTES apart)
                VOX_TRY(vox_launch_talker_mlp(st, tm));
                return VOX_OK;
            }
            if (decode_rows && n == 1 && s->mlp_persist && !ablate()) {
                // one request: o_proj + residual, gate/up, down + residual as ONE persistent launch (bit-identical to the three below)
                TalkerMlpCall tm;
                tm.wo = w.wo; tm.wgate = w.wgate; tm.wup = w.wup; tm.wdown = w.wdown; tm.ln2 = w.ln2; tm.attn = s->attn_out; tm.x = x;
                tm.gran = s->mlp_gran; tm.epoch = s->mlp_words; tm.err = s->mlp_words + 1; tm.eps = c.eps;
                tm.hidden = c.hidden; tm.nq = nq; tm.ffn = c.ffn;
                if (attn_in) tm.attn_call = &ac;
                if (l + 1 < c.layers && nqkv == 4096 && !s->layers[l + 1].bqkv) {      // ... and the next layer's q | k | v projection
                    tm.wqkv_next = s->layers[l + 1].wqkv; tm.ln1_next = s->layers[l + 1].ln1; tm.qkv_out = s->qkv; tm.nqkv = nqkv;
                    qkv_done = true;
                }
                VOX_TRY(vox_launch_talker_mlp(st, tm));
                xn_ready = xf_ready = false;
                continue;
            }
            if (!(ablate() & 32)) VOX_TRY(vox_launch_linear(s->ctx, st, o));
        }
        if (xn_ready) g.x_prenormed = s->xn;
        if (!(ablate() & 64)) VOX_TRY(vox_launch_linear(s->ctx, st, g));
        xn_ready = l + 1 < c.layers && vox_linear_is_rows_gemm(d);    // ... and the next layer's input_layernorm(x)
        if (xn_ready) { d.post_norm_w = s->layers[l + 1].ln1; d.post_norm_out = s->xn; }
        xf_ready = d_fk && l + 1 < c.layers;                          // (the next qkv decides whether it can use it)
        if (xf_ready) d.y_frag = s->xfrag;
        if (!(ablate() & 128)) VOX_TRY(vox_launch_linear(s->ctx, st, d));
    }
    return VOX_OK;
}

extern "C" {

// Persistent MLP half for one-row decode layers (k_talker_mlp): the Qwen3-TTS talker shape on a part with >= 256 CUs;
// VOX_TALKER_PERSIST=0 keeps the three launches.  Only an owner that ships the launch's error word to the host with every frame
// (vox_qwen3: the status row, engine.hip k_rng_bump) may turn it on: a bare vox_stack, vox_lm and vox_csm have no status row,
// so a hand-off timeout there would return garbage silently — they keep the launch chain (bit-identical).
static int stack_enable_mlp_persist(vox_stack* s) {
    TalkerMlpCall probe;
    probe.hidden = s->cfg.hidden; probe.nq = s->cfg.heads * s->cfg.head_dim; probe.ffn = s->cfg.ffn;
    const char* e = getenv("VOX_TALKER_PERSIST");
    const bool want = e ? e[0] == '1' : true;
    int n_cu = 0, dev_id = 0;
    (void)hipGetDevice(&dev_id);
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_id);
    if (want && n_cu >= 256 && vox_talker_mlp_supported(probe) && !s->cfg.qkv_bias) {
        if (hipMalloc(&s->mlp_gran, 7168 * 8) != hipSuccess || hipMalloc((void**)&s->mlp_words, 16) != hipSuccess)
            return vox_fail(VOX_ERR_NOMEM, "stack_create: hipMalloc");
        VOX_HIP(hipMemset(s->mlp_gran, 0, 7168 * 8));
        const unsigned words[4] = {1u, 0u, vox_persist_spins_default(), 0u};      // epoch, error, poll bound, test hook
        VOX_HIP(hipMemcpy(s->mlp_words, words, 16, hipMemcpyHostToDevice));
        s->mlp_persist = 1;
        // the decode attention INSIDE that launch (blocks 0..15, <= 256 visible tokens): bit-identical.  Round 5 measured it equal to the
        // two-launch form (what the overlapped weight stream saved, the 512-thread attention under that stream lost: its 16 K/V requests
        // per thread took 8.9 us to issue).  Round 6: with one address computation per tile instead of one per request (a 32-token chunk
        // lies in one page: VOX_KV_ASM_FETCH) and the tuned polls the frame is 2.2 % shorter with it (2.157 -> 2.110 ms) — default on,
        // VOX_TALKER_ATTN=0 keeps the attention launch
        const char* ea = getenv("VOX_TALKER_ATTN");
        s->mlp_attn = !(ea && ea[0] == '0') && s->cfg.heads == 16 && s->cfg.kv_heads == 8 && s->cfg.head_dim == 128;
        // every layer in one launch (round 6): needs the attention in the launch, head norms, a RoPE table and bias-free projections
        const char* em = getenv("VOX_TALKER_MULTI");
        bool multi = s->mlp_attn && !(em && em[0] == '0') && s->cfg.qk_norm && s->rope && s->cfg.layers >= 1;
        for (const vox_layer_weights& w : s->layers) multi = multi && !w.bqkv && w.qnorm && w.knorm;
        if (multi) {
            std::vector<const void*> tab;
            for (const vox_layer_weights& w : s->layers)
                for (const void* q : {w.wo, w.wgate, w.wup, w.wdown, w.ln2, w.wqkv, w.ln1, w.qnorm, w.knorm}) tab.push_back(q);
            if (hipMalloc(&s->mlp_tab, tab.size() * sizeof(void*)) != hipSuccess) return vox_fail(VOX_ERR_NOMEM, "stack_create: hipMalloc");
            VOX_HIP(hipMemcpy(s->mlp_tab, tab.data(), tab.size() * sizeof(void*), hipMemcpyHostToDevice));
        }
    }
    return VOX_OK;
}
int vox_stack_create(vox_ctx* ctx, const vox_stack_config* cfg, const vox_layer_weights* layers, const void* final_norm,
                     const float* rope, int rope_max_pos, vox_stack** out) {
    if (!ctx || !cfg || !layers || !out) return vox_fail(VOX_ERR_INVALID, "stack_create: NULL argument");
    if (cfg->hidden % 8 || cfg->head_dim % 8 || cfg->ffn % 8 || cfg->heads % cfg->kv_heads)
        return vox_fail(VOX_ERR_INVALID, "stack_create: dims must be multiples of 8 and heads %% kv_heads == 0");
    vox_stack* s = new vox_stack();
    s->ctx = ctx;
    s->cfg = *cfg;
    s->layers.assign(layers, layers + cfg->layers);
    s->final_norm = final_norm;
    s->rope = rope;
    s->rope_max_pos = rope_max_pos;
    const size_t nq = (size_t)cfg->heads * cfg->head_dim, nkv = (size_t)cfg->kv_heads * cfg->head_dim;
    const size_t R = cfg->max_rows;
    s->attn_ws_floats = (size_t)R * cfg->heads * n_chunks(cfg->max_kvlen) * (cfg->head_dim + 2);
    {   // split-K GEMM partials: the widest of (qkv, o, gate+up, down) at 128 rows per pass
        const size_t sh = (cfg->hidden + 255) / 256, sf = (cfg->ffn + 255) / 256, sq = (nq + 255) / 256;
        size_t m = sh * (nq + 2 * nkv);
        m = m > sq * cfg->hidden ? m : sq * cfg->hidden;
        m = m > sh * 2 * cfg->ffn ? m : sh * 2 * cfg->ffn;
        m = m > sf * cfg->hidden ? m : sf * cfg->hidden;
        s->skws_bytes = R > 8 ? m * 128 * 4 : 256;
    }
    if (hipMalloc(&s->qkv, R * (nq + 2 * nkv) * 2) != hipSuccess || hipMalloc(&s->q, R * nq * 2) != hipSuccess ||
        hipMalloc(&s->h, R * cfg->ffn * 2) != hipSuccess || hipMalloc(&s->attn_out, R * nq * 2) != hipSuccess ||
        hipMalloc(&s->xn, R * cfg->hidden * 2) != hipSuccess || hipMalloc(&s->skws, s->skws_bytes) != hipSuccess ||
        hipMalloc((void**)&s->attn_ws, s->attn_ws_floats * 4) != hipSuccess) {
        delete s;
        return vox_fail(VOX_ERR_NOMEM, "stack_create: hipMalloc failed");
    }
    if (R > (size_t)ctx->exact_rows) {
        // fragment-major weight copies for the 9..32 rows GEMM (rows * 2 bytes each: doubles the stack's weight footprint;
        // the 1..8 rows GEMV and the 33+ rows split-K GEMM keep reading the caller's row-major tensors)
        auto mk = [&](const void* src, int rows, int K, void** dst) -> bool {
            *dst = nullptr;
            if (!src || !(vox_fullk_weight_ok(rows, K) || vox_stream_weight_ok(rows, K))) return true;
            if (hipMalloc(dst, (size_t)rows * K * 2) != hipSuccess) return false;
            return vox_launch_swizzle_frag(nullptr, src, *dst, rows, K) == VOX_OK;
        };
        s->fw.resize(cfg->layers);
        bool ok = hipMalloc(&s->xfrag, (size_t)128 * cfg->hidden * 2) == hipSuccess && hipMalloc(&s->hfrag, (size_t)128 * cfg->ffn * 2) == hipSuccess &&
                  hipMalloc(&s->afrag, (size_t)128 * nq * 2) == hipSuccess;
        if (ok) { (void)hipMemset(s->afrag, 0, (size_t)128 * nq * 2); (void)hipMemset(s->xfrag, 0, (size_t)128 * cfg->hidden * 2); (void)hipMemset(s->hfrag, 0, (size_t)128 * cfg->ffn * 2); }
        for (int l = 0; ok && l < cfg->layers; ++l) {
            const vox_layer_weights& w = s->layers[l];
            ok = mk(w.wqkv, (int)(nq + 2 * nkv), cfg->hidden, &s->fw[l].qkv) && mk(w.wo, cfg->hidden, (int)nq, &s->fw[l].o) &&
                 mk(w.wgate, cfg->ffn, cfg->hidden, &s->fw[l].gate) && mk(w.wup, cfg->ffn, cfg->hidden, &s->fw[l].up) &&
                 mk(w.wdown, cfg->hidden, cfg->ffn, &s->fw[l].down);
        }
        if (!ok || hipDeviceSynchronize() != hipSuccess) {
            vox_stack_destroy(s);
            return vox_fail(VOX_ERR_NOMEM, "stack_create: fragment-major weight copies failed");
        }
    }
#ifdef VOX_DEV_KNOBS
    if (const char* e = getenv("VOX_STACK_KEEP")) s->keep_weights = atoi(e);
#endif
    *out = s;
    return VOX_OK;
}
void vox_stack_destroy(vox_stack* s) {
    if (!s) return;
    (void)hipFree(s->mlp_gran); (void)hipFree(s->mlp_words); (void)hipFree(s->mlp_tab);
    for (auto& f : s->fw) { (void)hipFree(f.qkv); (void)hipFree(f.o); (void)hipFree(f.gate); (void)hipFree(f.up); (void)hipFree(f.down); }
    (void)hipFree(s->xfrag); (void)hipFree(s->hfrag); (void)hipFree(s->afrag);
    (void)hipFree(s->qkv); (void)hipFree(s->q); (void)hipFree(s->h); (void)hipFree(s->attn_out); (void)hipFree(s->xn); (void)hipFree(s->skws); (void)hipFree(s->attn_ws);
    delete s;
}

int vox_stack_forward(vox_stack* s, void* stream, void* x, void* y, void* kv, int64_t kv_layer_stride,
                      const vox_rows* rows) {
    if (!s || !rows) return vox_fail(VOX_ERR_INVALID, "stack_forward: NULL");
    hipStream_t st = (hipStream_t)stream;
    // the decode-row hints promise one new token per request: take the fused (norm + RoPE + append in-kernel) path
    const bool decode_rows = rows->fixed_kvlen > 0 || rows->page_table != nullptr || rows->identity_pages;
    VOX_TRY(stack_layers(s, st, x, kv, kv_layer_stride, rows, decode_rows, rows->n_rows <= s->ctx->exact_rows));
    if (y && s->final_norm)
        VOX_TRY(vox_launch_rmsnorm(st, x, s->final_norm, y, rows->n_rows, s->cfg.hidden, s->cfg.eps));
    return VOX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// Qwen3-TTS frame engine
// ---------------------------------------------------------------------------------------------------
struct vox_qwen3 {
    vox_ctx* ctx;
    vox_qwen3_config cfg;
    vox_qwen3_weights w;
    std::vector<const void*> depth_emb;
    void* proj_tab = nullptr;   // [n_groups-2][depth_vocab][depth hidden]: small_to_mtp_projection(depth_emb[i][v]), tabulated once
    vox_stack *talker, *depth;
    // device buffers
    void *te, *t1, *text, *x, *depth_x, *dx, *dlogits, *dkv;
    int32_t* iota;           // 0..max(2*max_batch, ...)
    int32_t* dmeta;          // depth plan arrays
    int32_t* suppress;
    int n_suppress;
    int64_t dkv_stride;
    // depth meta offsets (in int32 elements) for i==1 and per i>=2
    int32_t *d1_pos, *d1_req, *d1_kvlen, *d1_slot, *d_indptr, *odd_rows;
    std::vector<int32_t*> di_pos, di_kvlen;
    // persistent depth step (one request): device copy of the depth layers' weight pointers, the hand-off granules, the epoch
    // and error words (kernels_lm.hip: k_depth_step)
    void *dstep_layers = nullptr, *dstep_gran = nullptr;
    unsigned* dstep_words = nullptr;      // [0] epoch, [1] error
    int dstep = 0;                         // 1: depth steps 2.. of a one-request frame run as ONE launch each
    // Fail-loud support (vox_qwen3_set_status): the last kernel of every frame / prefill writes the persistent kernels' error word to
    // the caller's device word, which travels with the token snapshot the host reads anyway; every decode frame saves its inputs
    // (ids, masks, features of the rows it runs, the frame counter) to one of two shadow slots — the counter's parity picks the slot,
    // so a frame still in flight behind a failed one does not overwrite it — for vox_qwen3_frame_restore.
    int32_t* status = nullptr;
    Qwen3Shadow shadow{};
};

__global__ void k_frame_init(int* out_ids, int stride, int col, int val, int B, const uint64_t* rng, uint64_t* sh_rng) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out_ids[(size_t)b * stride + col] = val;
    if (b == 0 && rng && sh_rng) sh_rng[*rng & 1] = *rng;       // shadow of the frame counter (frames and prefills; it is bumped by the frame's last kernel)
}
// status word of the frame: the first error code of the persistent kernels (0 = every hand-off arrived)
__device__ __forceinline__ void frame_status(int32_t* status, const unsigned* e0, const unsigned* e1) {
    if (!status) return;
    unsigned v = e0 ? e0[1] : 0u;
    if (!v && e1) v = e1[1];
    status[0] = (int32_t)v;
}
__global__ void k_qwen3_feedback(const int* out_ids, int* input_ids, uint8_t* masks, const bf16_t* next_feat,
                                 bf16_t* feat, uint64_t* rng, int G1, int H, int pad_id, int B, int32_t* status, const unsigned* e0,
                                 const unsigned* e1) {
    const int b = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int j = 0; j < G1; ++j) input_ids[(size_t)b * G1 + j] = 0;
        input_ids[(size_t)b * G1] = out_ids[(size_t)b * G1];
        input_ids[(size_t)b * G1 + G1 - 1] = pad_id;
        masks[b] = 1;
        if (b == 0 && rng) *rng += 1;
        if (b == 0) frame_status(status, e0, e1);
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H; i += gridDim.x * 256)
        feat[(size_t)b * H + i] = next_feat[(size_t)b * H + i];
}
__global__ void k_rng_bump(uint64_t* rng, int32_t* status, const unsigned* e0, const unsigned* e1) {
    if (rng) *rng += 1;
    frame_status(status, e0, e1);
}
// Inputs of the frame `back` frames ago (1 = the last one) back into place: slot = parity of the counter value that frame started
// from; a slot that does not hold that counter value (no such frame was run) leaves everything alone and reports 0x7fffffff.
__global__ __launch_bounds__(256) void k_qwen3_restore(Qwen3Shadow sh, int* ids, uint8_t* masks, bf16_t* feat, uint64_t* rng, int back, int n_rows,
                                                       int32_t* status) {
    const uint64_t want = *rng - (uint64_t)back;
    const int slot = (int)(want & 1);
    __shared__ int ok;
    if (threadIdx.x == 0) ok = sh.rng[slot] == want;
    __syncthreads();
    if (!ok) { if (threadIdx.x == 0 && status) status[0] = 0x7fffffff; return; }
    for (int b = 0; b < n_rows; ++b) {
        for (int j = threadIdx.x; j < sh.G1; j += 256) ids[(size_t)b * sh.G1 + j] = sh.ids[((size_t)slot * sh.max_batch + b) * sh.G1 + j];
        if (threadIdx.x == 0) masks[b] = sh.mask[(size_t)slot * sh.max_batch + b];
        for (int i = threadIdx.x; i < sh.H; i += 256) feat[(size_t)b * sh.H + i] = sh.feat[((size_t)slot * sh.max_batch + b) * sh.H + i];
    }
    __syncthreads();
    if (threadIdx.x == 0) { *rng = want; if (status) status[0] = 0; }
}
__global__ void k_copy_rows(const bf16_t* src, long src_stride, bf16_t* dst, long dst_stride, int H) {
    const int b = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H; i += gridDim.x * 256)
        dst[(size_t)b * dst_stride + i] = src[(size_t)b * src_stride + i];
}

static int qwen3_tail(vox_qwen3* m, hipStream_t st, const vox_qwen3_io* io, int B, const vox_sampling_config* sc,
                      uint64_t seed, int feedback) {
    // Runs after out_logits [B,vocab] and depth_x rows 2b (hidden) are in place:
    // codebook-0 sampling, the depth loop, and the feedback of the next step's inputs.
    const vox_qwen3_config& c = m->cfg;
    const int H = c.talker.hidden, Hd = c.depth.hidden, G = c.n_groups, G1 = G + 1;
    hipLaunchKernelGGL(k_frame_init, dim3((B + 63) / 64), dim3(64), 0, st, io->out_ids, G1, G, c.tts_pad_id, B, (const uint64_t*)io->rng_offset,
                       m->status ? m->shadow.rng : nullptr);
    {
        SampleCall s;
        s.logits = io->out_logits; s.B = B; s.V = c.vocab; s.suppress_ids = m->suppress; s.n_suppress = m->n_suppress;
        s.cfg = *sc; s.cfg.repetition_penalty = 1.0f;  // Qwen3: cache never persisted => identity
        s.seed = seed; s.offset = 0; s.offset_dev = io->rng_offset; s.offset_mul = (uint64_t)G;
        s.out_ids = io->out_ids; s.out_stride = G1; s.out_col = 0;
        s.emb_table = m->w.codec_embedding; s.emb_vocab = c.vocab; s.H = H;
        s.emb_dst = (bf16_t*)m->depth_x + H; s.emb_dst_stride = 2L * H; s.ws = m->ctx->samp_ws;
        VOX_TRY(vox_launch_sample(st, s));
    }
    // One request, greedy: the pick of codebook i happens at the start of the persistent launch of step i + 1 (every block takes the
    // argmax itself) instead of in a sampler launch between the two — 14 launches less per frame; VOX_DEPTH_PICK=0 keeps them.
    static const bool pick_env = [] { const char* e = getenv("VOX_DEPTH_PICK"); return !(e && e[0] == '0'); }();
    // ... and round 6: the SAMPLED pick too (top-k <= 256 over the 2048-entry depth vocabulary: k_sample_topk's contract inside the launch);
    // VOX_DEPTH_PICK_SAMPLED=0 keeps the sampler launches for sampled frames
    static const bool pick_sampled_env = [] { const char* e = getenv("VOX_DEPTH_PICK_SAMPLED"); return !(e && e[0] == '0'); }();
    const bool greedy_cfg = sc->greedy || sc->temperature == 0.0f;
    const bool sampled_ok = pick_sampled_env && !greedy_cfg && sc->top_k > 0 && sc->top_k <= 256 && c.depth_vocab == 2048 && sc->temperature > 0.0f &&
                            sc->min_p <= 1.0f;
    const bool fuse_pick = pick_env && m->dstep && B == 1 && !ablate() && (greedy_cfg || sampled_ok) && m->proj_tab && c.depth_vocab % 4 == 0 &&
                           c.depth_vocab <= 65536 && H % 8 == 0;
    DepthStepCall pend;         // the deferred pick (pick_* fields only)
    bool have_pend = false;
#ifdef VOX_DEV_KNOBS
    static int depth_rep = -1;      // (dev knob VOX_DEPTH_REPEAT: the depth loop n times over, timing only)
    if (depth_rep < 0) { const char* e = getenv("VOX_DEPTH_REPEAT"); depth_rep = e ? atoi(e) : 1; }
    for (int rep = 0; rep < depth_rep; ++rep)
#endif
    for (int i = (ablate() & 4096) ? 2 : 1; i < G && !(ablate() & 2); ++i) {      // (dev knob 4096: leave out depth step 1)
        const int rows = i == 1 ? 2 * B : B;
        if (i == 1 || !m->proj_tab) {     // (steps >= 2: the previous step's sampler gathered the tabulated projection into dx)
            LinearCall p;  // small_to_mtp_projection
            p.W = m->w.mtp_w; p.bias = m->w.mtp_b; p.x = m->depth_x; p.y = m->dx; p.B = rows; p.N = Hd; p.K = H;
            p.pro = VOX_PRO_COPY; p.epi = VOX_EPI_STORE; p.fixed_order = B <= m->ctx->exact_rows; p.exact_rows = m->ctx->exact_rows; p.keep_weights = m->depth->keep_weights;
            p.splitk_ws = m->depth->skws; p.splitk_ws_bytes = m->depth->skws_bytes;
            VOX_TRY(vox_launch_linear(m->ctx, st, p));
        }
        vox_rows r{};
        r.pos = i == 1 ? m->d1_pos : m->di_pos[i];
        r.q_req = i == 1 ? m->d1_req : m->iota;
        r.q_kvlen = i == 1 ? m->d1_kvlen : m->di_kvlen[i];
        r.page = r.q_req;
        r.slot = i == 1 ? m->d1_slot : m->di_pos[i];
        r.kv_indptr = m->d_indptr;
        r.kv_indices = m->iota;
        r.n_rows = rows;
        r.max_kvlen = i + 1;
        if (i > 1) { r.fixed_kvlen = i + 1; r.fixed_pos = i; r.identity_pages = 1; }
        void* dl = io->out_depth_logits ? (void*)((bf16_t*)io->out_depth_logits + (size_t)(i - 1) * c.max_batch * c.depth_vocab)
                                        : m->dlogits;
        if (m->dstep && B == 1 && i > 1 && !ablate()) {
            // one request: the step's 5 layers + head as ONE persistent launch (bit-identical to the chain below)
            const vox_stack_config& dcfg = m->depth->cfg;
            DepthStepCall ds;
            ds.layers_dev = m->dstep_layers; ds.n_layers = dcfg.layers; ds.final_norm = m->w.depth_norm;
            ds.head_w = (const bf16_t*)m->w.depth_lm_head + (size_t)(i - 1) * c.depth_vocab * Hd;
            ds.x_in = m->dx; ds.logits = dl; ds.gran = m->dstep_gran; ds.epoch = m->dstep_words; ds.err = m->dstep_words + 1;
            ds.kv = m->dkv; ds.kv_layer_stride = (long)m->dkv_stride; ds.cs = m->depth->rope; ds.eps = dcfg.eps;
            ds.scale = 1.0f / sqrtf((float)dcfg.head_dim);
            ds.hidden = dcfg.hidden; ds.heads = dcfg.heads; ds.kv_heads = dcfg.kv_heads; ds.head_dim = dcfg.head_dim; ds.ffn = dcfg.ffn;
            ds.vocab = c.depth_vocab; ds.qk_norm = dcfg.qk_norm; ds.qkv_bias = dcfg.qkv_bias; ds.rope_dim = dcfg.rope_dim;
            ds.rope_interleave = dcfg.rope_interleave; ds.page_size = dcfg.page_size; ds.table_max_pos = m->depth->rope_max_pos;
            ds.n_tokens = i + 1;
            if (have_pend) {
                ds.pick_logits = pend.pick_logits; ds.pick_tab = pend.pick_tab; ds.pick_emb = pend.pick_emb; ds.pick_out = pend.pick_out;
                ds.pick_feat = pend.pick_feat; ds.pick_vocab = pend.pick_vocab; ds.pick_H = pend.pick_H; ds.pick_init = pend.pick_init;
                ds.pick_top_k = pend.pick_top_k; ds.pick_top_p = pend.pick_top_p; ds.pick_min_p = pend.pick_min_p; ds.pick_temperature = pend.pick_temperature;
                ds.pick_seed = pend.pick_seed; ds.pick_offset = pend.pick_offset; ds.pick_offset_mul = pend.pick_offset_mul; ds.pick_offset_dev = pend.pick_offset_dev;
                have_pend = false;
            }
            VOX_TRY(vox_launch_depth_step(st, ds));
        } else {
        VOX_TRY(stack_layers(m->depth, st, m->dx, m->dkv, m->dkv_stride, &r, i > 1, B <= m->ctx->exact_rows));
        LinearCall h;  // depth final norm + lm_head[i-1]
        h.W = (const bf16_t*)m->w.depth_lm_head + (size_t)(i - 1) * c.depth_vocab * Hd;
        h.x = m->dx; h.x_rows = i == 1 ? m->odd_rows : nullptr; h.norm_w = m->w.depth_norm; h.eps = c.depth.eps;
        h.y = dl; h.B = B; h.N = c.depth_vocab; h.K = Hd; h.pro = VOX_PRO_RMSNORM; h.epi = VOX_EPI_STORE;
        h.keep_weights = m->depth->keep_weights;
#ifdef VOX_DEV_KNOBS
        { static int hk = -2; if (hk == -2) { const char* e = getenv("VOX_HEAD_KEEP"); hk = e ? atoi(e) : -1; } if (hk >= 0) h.keep_weights = hk; }
#endif
        if (!(ablate() & 2048)) VOX_TRY(vox_launch_linear(m->ctx, st, h));      // (dev knob 2048: leave out the depth heads)
        }
        SampleCall s;
        s.logits = dl; s.B = B; s.V = c.depth_vocab; s.cfg = *sc; s.cfg.repetition_penalty = 1.0f;
        s.seed = seed; s.offset = (uint64_t)i; s.offset_dev = io->rng_offset; s.offset_mul = (uint64_t)G;
        s.out_ids = io->out_ids; s.out_stride = G1; s.out_col = i;
        s.emb_table = m->depth_emb[i - 1]; s.emb_vocab = c.depth_vocab; s.H = H;
        s.emb_dst = m->depth_x; s.emb_dst_stride = H;
        if (m->proj_tab && i + 1 < G) {
            s.emb_dst = nullptr;                 // the raw embedding only feeds the feature accumulation
            s.emb2_table = (const bf16_t*)m->proj_tab + (size_t)(i - 1) * c.depth_vocab * Hd; s.H2 = Hd;
            s.emb2_dst = m->dx; s.emb2_dst_stride = Hd;
        }
        s.feat_acc = io->next_features; s.feat_init = i == 1; s.ws = m->ctx->samp_ws;
        if (fuse_pick && i + 1 < G && s.emb2_table && s.feat_acc) {       // (step i + 1 >= 2 of one request is a persistent launch)
            pend.pick_logits = dl; pend.pick_vocab = c.depth_vocab; pend.pick_tab = s.emb2_table; pend.pick_emb = s.emb_table;
            pend.pick_out = io->out_ids + i; pend.pick_feat = io->next_features; pend.pick_H = H; pend.pick_init = s.feat_init;
            pend.pick_top_k = 0;
            if (!greedy_cfg) {      // the sampler call's parameters, as vox_launch_sample would have taken them
                pend.pick_top_k = s.cfg.top_k; pend.pick_top_p = s.cfg.top_p; pend.pick_min_p = s.cfg.min_p; pend.pick_temperature = s.cfg.temperature;
                pend.pick_seed = s.seed; pend.pick_offset = s.offset; pend.pick_offset_mul = s.offset_mul; pend.pick_offset_dev = (const uint64_t*)s.offset_dev;
            }
            have_pend = true;
            continue;
        }
        if (!(ablate() & 8)) VOX_TRY(vox_launch_sample(st, s));
    }
    const unsigned* e0 = m->dstep_words;                     // (NULL when the engine never had the persistent kernels)
    const unsigned* e1 = m->talker->mlp_words;
    if (feedback) {
        hipLaunchKernelGGL(k_qwen3_feedback, dim3((H + 255) / 256, B), dim3(256), 0, st, io->out_ids, io->input_ids,
                           io->input_masks, (const bf16_t*)io->next_features, (bf16_t*)io->input_features,
                           io->rng_offset, G1, H, c.tts_pad_id, B, m->status, e0, e1);
    } else if (io->rng_offset || m->status) {
        hipLaunchKernelGGL(k_rng_bump, dim3(1), dim3(1), 0, st, io->rng_offset, m->status, e0, e1);
    }
    return VOX_OK;
}

static int qwen3_embed(vox_qwen3* m, hipStream_t st, const int32_t* ids, const uint8_t* masks, const void* feats,
                       int n, const uint64_t* shadow_rng = nullptr) {
    const vox_qwen3_config& c = m->cfg;
    const int H = c.talker.hidden, G1 = c.n_groups + 1;
    VOX_TRY(vox_launch_gather(st, m->w.text_embedding, ids, G1, G1 - 1, m->te, c.text_hidden, n, c.text_hidden,
                              c.text_vocab));
    LinearCall f1;
    f1.W = m->w.tp_fc1_w; f1.bias = m->w.tp_fc1_b; f1.x = m->te; f1.y = m->t1; f1.B = n; f1.N = c.text_hidden;
    f1.K = c.text_hidden; f1.pro = VOX_PRO_COPY; f1.epi = VOX_EPI_SILU;
    VOX_TRY(vox_launch_linear(m->ctx, st, f1));
    LinearCall f2;
    f2.W = m->w.tp_fc2_w; f2.bias = m->w.tp_fc2_b; f2.x = m->t1; f2.y = m->text; f2.B = n; f2.N = H;
    f2.K = c.text_hidden; f2.pro = VOX_PRO_COPY; f2.epi = VOX_EPI_STORE;
    VOX_TRY(vox_launch_linear(m->ctx, st, f2));
    // decode frames with a status word: the mix kernel (which reads ids / masks / features anyway) also saves them to the shadow slot
    return vox_launch_qwen3_mix(st, m->text, m->w.codec_embedding, ids, G1, masks, feats, m->x, n, H, c.vocab,
                                shadow_rng && m->status ? &m->shadow : nullptr, shadow_rng);
}


// Voice-clone prompt features (qwen3_tts.py:1657-1672, 1733-1744): row 0..T-1 = the bf16 running sum over codebooks
// 1..G-1 of code_predictor.codec_embedding[cb-1][ref_codes[t][cb]] (each add rounded to bf16, in codebook order, as the
// reference's in-place `+=` on a bf16 tensor does); spk_out = bf16(speaker_embedding - codec_embedding[codec_pad]).
struct IclTabs { const bf16_t* emb[31]; };
__global__ __launch_bounds__(256) void k_qwen3_prompt_features(IclTabs tabs, const int32_t* codes, int T, int G, int H, int dvocab,
                                                                const bf16_t* spk, const bf16_t* codec_emb, int pad_id,
                                                                bf16_t* spk_out, bf16_t* icl_out) {
    int t = blockIdx.x;
    if (t == T) {
        if (spk)
            for (int i = threadIdx.x; i < H; i += 256) spk_out[i] = f2bf(bf2f(spk[i]) - bf2f(codec_emb[(size_t)pad_id * H + i]));
        return;
    }
    for (int i = threadIdx.x; i < H; i += 256) {
        float acc = 0.f;
        for (int cb = 1; cb < G; ++cb) {
            int c = codes[(size_t)t * G + cb];
            c = c < 0 ? 0 : (c >= dvocab ? dvocab - 1 : c);
            acc = bfround(acc + bf2f(tabs.emb[cb - 1][(size_t)c * H + i]));
        }
        icl_out[(size_t)t * H + i] = f2bf(acc);
    }
}

extern "C" {

int vox_qwen3_create(vox_ctx* ctx, const vox_qwen3_config* cfg, const vox_qwen3_weights* w, vox_qwen3** out) {
    if (!ctx || !cfg || !w || !out) return vox_fail(VOX_ERR_INVALID, "qwen3_create: NULL argument");
    if (cfg->n_groups < 2 || cfg->max_batch < 1) return vox_fail(VOX_ERR_INVALID, "qwen3_create: bad config");
    vox_qwen3* m = new vox_qwen3();
    m->ctx = ctx;
    m->cfg = *cfg;
    m->w = *w;
    m->depth_emb.assign(w->depth_codec_embedding, w->depth_codec_embedding + cfg->n_groups - 1);
    const int B = cfg->max_batch, G = cfg->n_groups, H = cfg->talker.hidden, Hd = cfg->depth.hidden;
    vox_stack_config tc = cfg->talker, dc = cfg->depth;
    dc.page_size = G;
    dc.max_rows = 2 * B;
    dc.max_kvlen = G;
    int s = vox_stack_create(ctx, &tc, w->talker_layers, w->talker_norm, w->talker_rope, w->talker_rope_max_pos, &m->talker);
    if (s != VOX_OK) { delete m; return s; }
    s = stack_enable_mlp_persist(m->talker);      // this owner ships the launch's error word with every frame (status row)
    if (s != VOX_OK) { vox_stack_destroy(m->talker); delete m; return s; }
    s = vox_stack_create(ctx, &dc, w->depth_layers, w->depth_norm, w->depth_rope, w->depth_rope_max_pos, &m->depth);
    if (s != VOX_OK) { vox_stack_destroy(m->talker); delete m; return s; }
    m->depth->keep_weights = 1;   // 0.16 GB re-read 15 times per frame: Infinity-Cache resident
#ifdef VOX_DEV_KNOBS
    if (const char* e = getenv("VOX_DEPTH_KEEP")) m->depth->keep_weights = atoi(e);
#endif
    const size_t R = tc.max_rows;
    m->dkv_stride = (int64_t)B * 2 * G * dc.kv_heads * dc.head_dim;
    bool ok = hipMalloc(&m->te, R * cfg->text_hidden * 2) == hipSuccess &&
              hipMalloc(&m->t1, R * cfg->text_hidden * 2) == hipSuccess &&
              hipMalloc(&m->text, R * H * 2) == hipSuccess && hipMalloc(&m->x, R * H * 2) == hipSuccess &&
              hipMalloc(&m->depth_x, (size_t)2 * B * H * 2) == hipSuccess &&
              hipMalloc(&m->dx, (size_t)2 * B * Hd * 2) == hipSuccess &&
              hipMalloc(&m->dlogits, (size_t)B * cfg->depth_vocab * 2) == hipSuccess &&
              hipMalloc(&m->dkv, (size_t)dc.layers * m->dkv_stride * 2) == hipSuccess;
    if (!ok) return vox_fail(VOX_ERR_NOMEM, "qwen3_create: hipMalloc failed");
    VOX_HIP(hipMemset(m->dkv, 0, (size_t)dc.layers * m->dkv_stride * 2));
    // static plan arrays of the depth loop (worker/base.py:559-612)
    const int niota = (int)(R > (size_t)2 * B + 1 ? R : 2 * B + 1);
    std::vector<int32_t> host;
    auto push = [&](const std::vector<int32_t>& v) { size_t o = host.size(); host.insert(host.end(), v.begin(), v.end()); return o; };
    std::vector<int32_t> iota(niota), d1p(2 * B), d1r(2 * B), d1k(2 * B), odd(B), indptr(B + 1);
    for (int i = 0; i < niota; ++i) iota[i] = i;
    for (int b = 0; b < B; ++b) {
        d1p[2 * b] = 0; d1p[2 * b + 1] = 1; d1r[2 * b] = d1r[2 * b + 1] = b;
        d1k[2 * b] = 1; d1k[2 * b + 1] = 2; odd[b] = 2 * b + 1;
    }
    for (int b = 0; b <= B; ++b) indptr[b] = b;
    const size_t o_iota = push(iota), o_d1p = push(d1p), o_d1r = push(d1r), o_d1k = push(d1k), o_odd = push(odd),
                 o_ind = push(indptr);
    std::vector<size_t> o_pos(G, 0), o_kvl(G, 0);
    for (int i = 2; i < G; ++i) {
        o_pos[i] = push(std::vector<int32_t>(B, i));
        o_kvl[i] = push(std::vector<int32_t>(B, i + 1));
    }
    std::vector<int32_t> sup;
    for (int i = cfg->vocab - 1024; i < cfg->vocab; ++i)
        if (i >= 0 && i != cfg->eos_id) sup.push_back(i);   // qwen3_tts.py:1082-1086
    const size_t o_sup = push(sup);
    if (hipMalloc((void**)&m->dmeta, host.size() * 4) != hipSuccess) return vox_fail(VOX_ERR_NOMEM, "qwen3_create: hipMalloc");
    VOX_HIP(hipMemcpy(m->dmeta, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    m->iota = m->dmeta + o_iota; m->d1_pos = m->dmeta + o_d1p; m->d1_req = m->dmeta + o_d1r;
    m->d1_kvlen = m->dmeta + o_d1k; m->d1_slot = m->d1_pos; m->odd_rows = m->dmeta + o_odd;
    m->d_indptr = m->dmeta + o_ind;
    m->di_pos.assign(G, nullptr); m->di_kvlen.assign(G, nullptr);
    for (int i = 2; i < G; ++i) { m->di_pos[i] = m->dmeta + o_pos[i]; m->di_kvlen[i] = m->dmeta + o_kvl[i]; }
    m->suppress = m->dmeta + o_sup;
    m->n_suppress = (int)sup.size();
    if (G > 2) {
        // projection(embedding) tables for depth steps 2..G-1 (fixed-order kernel: bit-identical to the per-step launch
        // they replace; G-2 launches fewer per frame)
        const size_t tab = (size_t)cfg->depth_vocab * Hd;
        if (hipMalloc(&m->proj_tab, (size_t)(G - 2) * tab * 2) != hipSuccess) return vox_fail(VOX_ERR_NOMEM, "qwen3_create: hipMalloc");
        for (int i = 0; i + 2 < G; ++i) {
            LinearCall p;
            p.W = w->mtp_w; p.bias = w->mtp_b; p.x = m->depth_emb[i]; p.y = (bf16_t*)m->proj_tab + (size_t)i * tab;
            p.B = cfg->depth_vocab; p.N = Hd; p.K = H; p.pro = VOX_PRO_COPY; p.epi = VOX_EPI_STORE; p.fixed_order = 1;
            const int rc = vox_launch_linear(ctx, nullptr, p);
            if (rc != VOX_OK) return rc;
        }
        VOX_HIP(hipDeviceSynchronize());
    }
    {
        // persistent depth step (default for one-request frames): available when the depth transformer has the shape the kernel is
        // written for (Qwen3-TTS: hidden 1024, 16 / 8 heads of 128, FFN 3072, 2048-entry codebook heads) on a part with >= 256 CUs;
        // VOX_DEPTH_PERSIST=0 keeps the launch chain
        DepthStepCall probe;
        const vox_stack_config& dq = m->depth->cfg;       // (as the stack normalised it)
        probe.hidden = dq.hidden; probe.heads = dq.heads; probe.kv_heads = dq.kv_heads; probe.head_dim = dq.head_dim; probe.ffn = dq.ffn;
        probe.vocab = cfg->depth_vocab; probe.qk_norm = dq.qk_norm; probe.qkv_bias = dq.qkv_bias; probe.rope_dim = dq.rope_dim;
        probe.rope_interleave = dq.rope_interleave; probe.page_size = dq.page_size; probe.n_layers = dq.layers;
        probe.n_tokens = G;      // the LARGEST step (step i sees i + 1 <= G tokens): a config whose later steps the kernel cannot take keeps the chain
        const char* e = getenv("VOX_DEPTH_PERSIST");
        const bool want = e ? e[0] == '1' : true;
        int n_cu = 0, dev_id = 0;
        (void)hipGetDevice(&dev_id);
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_id);
        // 256 blocks of 512 threads must all be resident at once (one per CU): a partitioned or smaller part keeps the launch chain
        if (want && n_cu >= 256 && G >= 2 && vox_depth_step_supported(probe) && w->depth_rope) {
            std::vector<const void*> ptrs;
            for (int l = 0; l < dc.layers; ++l) {
                const vox_layer_weights& lw = w->depth_layers[l];
                for (const void* q : {lw.wqkv, lw.wo, lw.wgate, lw.wup, lw.wdown, lw.ln1, lw.ln2, lw.qnorm, lw.knorm}) ptrs.push_back(q);
            }
            if (hipMalloc(&m->dstep_layers, ptrs.size() * sizeof(void*)) != hipSuccess || hipMalloc(&m->dstep_gran, 4096 * 8) != hipSuccess ||
                hipMalloc((void**)&m->dstep_words, 16) != hipSuccess)
                return vox_fail(VOX_ERR_NOMEM, "qwen3_create: hipMalloc");
            VOX_HIP(hipMemcpy(m->dstep_layers, ptrs.data(), ptrs.size() * sizeof(void*), hipMemcpyHostToDevice));
            VOX_HIP(hipMemset(m->dstep_gran, 0, 4096 * 8));
            const unsigned words[4] = {1u, 0u, vox_persist_spins_default(), 0u};      // epoch, error, poll bound, test hook
            VOX_HIP(hipMemcpy(m->dstep_words, words, 16, hipMemcpyHostToDevice));
            m->dstep = 1;
        }
    }
    *out = m;
    return VOX_OK;
}

// 0 = every hand-off of the persistent depth steps so far arrived; else the code of the first spin that gave up (results since then
// are garbage).  *enabled: whether this engine runs them.  Synchronises the device.
int vox_qwen3_depth_persist_status(vox_qwen3* m, int32_t* enabled, uint32_t* error_code) {
    if (!m) return vox_fail(VOX_ERR_INVALID, "qwen3_depth_persist_status: NULL");
    if (enabled) *enabled = (m->dstep ? 1 : 0) | (m->talker->mlp_persist ? 2 : 0);     // bit 0: depth steps, bit 1: talker MLP halves
    if (error_code) {
        *error_code = 0;
        unsigned words[2];
        if (m->dstep) {
            VOX_HIP(hipMemcpy(words, m->dstep_words, 8, hipMemcpyDeviceToHost));
            *error_code = words[1];
        }
        if (!*error_code && m->talker->mlp_persist) {
            VOX_HIP(hipMemcpy(words, m->talker->mlp_words, 8, hipMemcpyDeviceToHost));
            *error_code = words[1];
        }
    }
    return VOX_OK;
}

// Per-frame status word (device, int32): written by the LAST kernel of every frame / prefill with the persistent kernels' error code
// (0 = fine).  The host reads it with the token snapshot of the frame: a hand-off that timed out is seen in the same frame.  Also
// enables the input shadow that vox_qwen3_frame_restore reads.  Call before any frame graph is captured.
int vox_qwen3_set_status(vox_qwen3* m, int32_t* status_dev) {
    if (!m) return vox_fail(VOX_ERR_INVALID, "qwen3_set_status: NULL");
    if (status_dev && !m->shadow.rng) {
        const vox_qwen3_config& c = m->cfg;
        Qwen3Shadow& sh = m->shadow;
        sh.G1 = c.n_groups + 1; sh.H = c.talker.hidden; sh.max_batch = c.max_batch;
        const size_t nb = (size_t)2 * c.max_batch;
        if (hipMalloc((void**)&sh.rng, 16) != hipSuccess || hipMalloc((void**)&sh.ids, nb * sh.G1 * 4) != hipSuccess ||
            hipMalloc((void**)&sh.mask, nb) != hipSuccess || hipMalloc((void**)&sh.feat, nb * sh.H * 2) != hipSuccess)
            return vox_fail(VOX_ERR_NOMEM, "qwen3_set_status: hipMalloc");
        VOX_HIP(hipMemset(sh.rng, 0xff, 16));
    }
    m->status = status_dev;
    return VOX_OK;
}
// Put the inputs of the frame `back` frames ago (1: the last frame, 2: the one before; 0: the frame that was started from the CURRENT
// counter value — during a replay: the launch that ran behind the failed one with inputs the caller had staged by hand) back: input_ids / input_masks / input_features of
// rows 0..n_rows-1 and the frame counter (io->rng_offset), from the shadow slot of that frame.  n_rows = 0 restores the counter only (a
// prefill's inputs are staged by the caller).  The status word reads 0 afterwards, 0x7fffffff when no such frame is in the shadow.
int vox_qwen3_frame_restore(vox_qwen3* m, void* stream, const vox_qwen3_io* io, int back, int n_rows) {
    if (!m || !io || !m->shadow.rng || !io->rng_offset) return vox_fail(VOX_ERR_INVALID, "qwen3_frame_restore: no shadow (vox_qwen3_set_status) or no frame counter");
    if (back < 0 || back > 2 || n_rows < 0 || n_rows > m->cfg.max_batch) return vox_fail(VOX_ERR_INVALID, "qwen3_frame_restore: bad arguments");
    hipLaunchKernelGGL(k_qwen3_restore, dim3(1), dim3(256), 0, (hipStream_t)stream, m->shadow, io->input_ids, io->input_masks,
                       (bf16_t*)io->input_features, io->rng_offset, back, n_rows, m->status);
    VOX_HIP(hipGetLastError());
    return VOX_OK;
}
// After a hand-off timeout (or as a precaution): wait for the device, clear the error words, re-zero the granules, move the epochs on;
// disable != 0 also turns the persistent kernels off for this engine — frames enqueued from now on take the bit-identical launch
// chains (graphs captured earlier still hold persistent launches: the caller drops them).
int vox_qwen3_persist_reset(vox_qwen3* m, int disable) {
    if (!m) return vox_fail(VOX_ERR_INVALID, "qwen3_persist_reset: NULL");
    VOX_HIP(hipDeviceSynchronize());
    auto reset = [](unsigned* words, void* gran, size_t n_gran) -> hipError_t {
        if (!words) return hipSuccess;
        unsigned w[4];
        hipError_t e = hipMemcpy(w, words, 16, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        w[0] += 1u; w[1] = 0u; w[3] = 0u;
        if ((e = hipMemcpy(words, w, 16, hipMemcpyHostToDevice)) != hipSuccess) return e;
        return hipMemset(gran, 0, n_gran * 8);
    };
    VOX_HIP(reset(m->dstep_words, m->dstep_gran, 4096));
    VOX_HIP(reset(m->talker->mlp_words, m->talker->mlp_gran, 7168));
    if (disable) { m->dstep = 0; m->talker->mlp_persist = 0; }
    return VOX_OK;
}
// bound of the persistent kernels' poll loops, in passes (>= ~0.5 us each; default 40000 or VOX_PERSIST_SPINS); tests lower it
int vox_qwen3_persist_set_spins(vox_qwen3* m, uint32_t spins) {
    if (!m || spins < 1) return vox_fail(VOX_ERR_INVALID, "qwen3_persist_set_spins: bad arguments");
    VOX_HIP(hipDeviceSynchronize());
    if (m->dstep_words) VOX_HIP(hipMemcpy(m->dstep_words + 2, &spins, 4, hipMemcpyHostToDevice));
    if (m->talker->mlp_words) VOX_HIP(hipMemcpy(m->talker->mlp_words + 2, &spins, 4, hipMemcpyHostToDevice));
    return VOX_OK;
}
// Test hook: in each of the next `count` launches of the chosen persistent kernel (0: depth step, 1: talker MLP half) block 1 withholds
// its first publish, so the blocks waiting for it run into the poll bound — the failure a stalled block would cause.
int vox_qwen3_persist_inject(vox_qwen3* m, int which, uint32_t count) {
    if (!m || which < 0 || which > 1) return vox_fail(VOX_ERR_INVALID, "qwen3_persist_inject: bad arguments");
    unsigned* words = which == 0 ? m->dstep_words : m->talker->mlp_words;
    if (!words) return vox_fail(VOX_ERR_INVALID, "qwen3_persist_inject: that persistent kernel is not enabled");
    VOX_HIP(hipDeviceSynchronize());
    VOX_HIP(hipMemcpy(words + 3, &count, 4, hipMemcpyHostToDevice));
    return VOX_OK;
}

void vox_qwen3_destroy(vox_qwen3* m) {
    if (!m) return;
    vox_stack_destroy(m->talker); vox_stack_destroy(m->depth);
    (void)hipFree(m->proj_tab); (void)hipFree(m->dstep_layers); (void)hipFree(m->dstep_gran); (void)hipFree(m->dstep_words);
    (void)hipFree(m->shadow.rng); (void)hipFree(m->shadow.ids); (void)hipFree(m->shadow.mask); (void)hipFree(m->shadow.feat);
    for (void* p : {m->te, m->t1, m->text, m->x, m->depth_x, m->dx, m->dlogits, m->dkv, (void*)m->dmeta}) (void)hipFree(p);
    delete m;
}

static int qwen3_head(vox_qwen3* m, hipStream_t st, const vox_qwen3_io* io, int B, const int32_t* x_rows) {
    const vox_qwen3_config& c = m->cfg;
    const int H = c.talker.hidden;
    LinearCall h;  // talker final norm (-> hidden for the depth model) + codec_head
    h.W = m->w.codec_head; h.x = m->x; h.x_rows = x_rows; h.norm_w = m->w.talker_norm; h.eps = c.talker.eps;
    h.x_out = m->depth_x; h.x_out_stride = 2L * H; h.y = io->out_logits; h.B = B; h.N = c.vocab; h.K = H;
    h.pro = VOX_PRO_RMSNORM; h.epi = VOX_EPI_STORE;
    VOX_TRY(vox_launch_linear(m->ctx, st, h));
    if (io->out_hidden)
        hipLaunchKernelGGL(k_copy_rows, dim3((H + 255) / 256, B), dim3(256), 0, st, (const bf16_t*)m->depth_x, 2L * H,
                           (bf16_t*)io->out_hidden, (long)H, H);
    return VOX_OK;
}

int vox_qwen3_frame(vox_qwen3* m, void* stream, const vox_qwen3_io* io, int B, int max_kvlen,
                    const vox_sampling_config* sc, uint64_t seed, int feedback) {
    if (!m || !io || !sc) return vox_fail(VOX_ERR_INVALID, "qwen3_frame: NULL");
    if (B < 1 || B > m->cfg.max_batch) return vox_fail(VOX_ERR_INVALID, "qwen3_frame: batch %d > max_batch", B);
    hipStream_t st = (hipStream_t)stream;
    VOX_TRY(qwen3_embed(m, st, io->input_ids, io->input_masks, io->input_features, B, (const uint64_t*)io->rng_offset));
    vox_rows r{};
    r.pos = io->pos; r.q_req = m->iota; r.q_kvlen = io->kvlen; r.page = io->page; r.slot = io->slot;
    r.kv_indptr = io->kv_indptr; r.kv_indices = io->kv_indices; r.n_rows = B; r.max_kvlen = max_kvlen;
    r.page_table = io->page_table; r.pt_stride = (int32_t)io->pt_stride;
    if (!(ablate() & 4)) VOX_TRY(stack_layers(m->talker, st, m->x, io->kv, io->kv_layer_stride, &r, true));
    VOX_TRY(qwen3_head(m, st, io, B, nullptr));
    return qwen3_tail(m, st, io, B, sc, seed, feedback);
}

int vox_qwen3_prefill(vox_qwen3* m, void* stream, const vox_qwen3_io* io, const int32_t* row_ids,
                      const uint8_t* row_masks, const void* row_features, const int32_t* q_req, int n_rows,
                      const int32_t* last_rows, int n_req, int max_kvlen, const vox_sampling_config* sc, uint64_t seed,
                      int feedback) {
    if (!m || !io || !sc) return vox_fail(VOX_ERR_INVALID, "qwen3_prefill: NULL");
    if (n_req < 0 || n_req > m->cfg.max_batch || n_rows > m->cfg.talker.max_rows)
        return vox_fail(VOX_ERR_INVALID, "qwen3_prefill: n_req %d / n_rows %d out of range", n_req, n_rows);
    hipStream_t st = (hipStream_t)stream;
    VOX_TRY(qwen3_embed(m, st, row_ids, row_masks, row_features, n_rows));
    vox_rows r{};
    r.pos = io->pos; r.q_req = q_req; r.q_kvlen = io->kvlen; r.page = io->page; r.slot = io->slot;
    r.kv_indptr = io->kv_indptr; r.kv_indices = io->kv_indices; r.n_rows = n_rows; r.max_kvlen = max_kvlen;
    VOX_TRY(stack_layers(m->talker, st, m->x, io->kv, io->kv_layer_stride, &r));
    if (n_req == 0) return VOX_OK;      // context chunk of a long prompt: K/V appended, nothing sampled
    VOX_TRY(qwen3_head(m, st, io, n_req, last_rows));
    return qwen3_tail(m, st, io, n_req, sc, seed, feedback);
}

int vox_qwen3_prompt_features(vox_qwen3* m, void* stream, const int32_t* ref_codes, int n_frames, const void* speaker_embedding,
                              int codec_pad_id, void* spk_out, void* icl_out) {
    if (!m) return vox_fail(VOX_ERR_INVALID, "qwen3_prompt_features: NULL");
    const vox_qwen3_config& c = m->cfg;
    if (n_frames < 0 || (n_frames && (!ref_codes || !icl_out)) || (speaker_embedding && !spk_out) || c.n_groups > 32 ||
        codec_pad_id < 0 || codec_pad_id >= c.vocab)
        return vox_fail(VOX_ERR_INVALID, "qwen3_prompt_features: bad arguments");
    IclTabs tabs{};
    for (int i = 0; i < c.n_groups - 1; ++i) tabs.emb[i] = (const bf16_t*)m->depth_emb[i];
    hipLaunchKernelGGL(k_qwen3_prompt_features, dim3(n_frames + 1), dim3(256), 0, (hipStream_t)stream, tabs, ref_codes, n_frames,
                       c.n_groups, c.talker.hidden, c.depth_vocab, (const bf16_t*)speaker_embedding,
                       (const bf16_t*)m->w.codec_embedding, codec_pad_id, (bf16_t*)spk_out, (bf16_t*)icl_out);
    VOX_HIP(hipGetLastError());
    return VOX_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// CSM-1B frame engine
// ---------------------------------------------------------------------------------------------------
struct vox_csm {
    vox_ctx* ctx;
    vox_csm_config cfg;
    vox_csm_weights w;
    vox_stack *backbone, *depth;
    void *x, *depth_x, *dx, *dlogits, *dkv;
    void* proj_tab = nullptr;   // [C-2][V][depth hidden]: inputs_embeds_projector(audio_embedding[i*V + v]), i = 1..C-2, tabulated once
    int32_t* dmeta;
    int32_t *iota, *d1_pos, *d1_req, *d1_kvlen, *d_indptr, *odd_rows;
    std::vector<int32_t*> di_pos, di_kvlen;
    int64_t dkv_stride;
};

// x[r] = bf16( sum_k mask[r,k] * emb_k[ids[r,k]] ), fp32, k ascending: audio codebooks 0..C-1 (table row k*V + id), then
// the text column (csm.py:647-653: `(embeds * masks).sum(dim=1)` — one rounding at the end)
__global__ __launch_bounds__(256) void k_csm_embed(const int* ids, const uint8_t* masks, const bf16_t* audio_emb,
                                                   const bf16_t* text_emb, bf16_t* x, int C, int V, int text_vocab, int H) {
    const int r = blockIdx.y, C1 = C + 1;
    const int* id = ids + (size_t)r * C1;
    const uint8_t* mk = masks + (size_t)r * C1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H; i += gridDim.x * 256) {
        float acc = 0.0f;
        for (int k = 0; k < C; ++k) {
            if (!mk[k]) continue;
            int t = id[k];
            t = t < 0 ? 0 : (t >= V ? V - 1 : t);
            acc = acc + bf2f(audio_emb[((size_t)k * V + t) * H + i]);
        }
        if (mk[C]) {
            int t = id[C];
            t = t < 0 ? 0 : (t >= text_vocab ? text_vocab - 1 : t);
            acc = acc + bf2f(text_emb[(size_t)t * H + i]);
        }
        x[(size_t)r * H + i] = f2bf(acc);
    }
}
// output row: every column starts as codebook 0 (`output_ids.repeat(1, n_codebooks)`, csm.py:697)
__global__ void k_csm_fill_row(int* out_ids, int C1, int B) {
    const int b = blockIdx.x, j = threadIdx.x;
    if (b < B && j > 0 && j < C1) out_ids[(size_t)b * C1 + j] = out_ids[(size_t)b * C1];
}
// next frame inputs: the 32 sampled codes, text column 0 and masked out (csm.py:705-709, 764-765)
__global__ void k_csm_feedback(const int* out_ids, int* input_ids, uint8_t* masks, uint64_t* rng, int C1, int B) {
    const int b = blockIdx.x, j = threadIdx.x;
    if (b < B && j < C1) {
        input_ids[(size_t)b * C1 + j] = j < C1 - 1 ? out_ids[(size_t)b * C1 + j] : 0;
        masks[(size_t)b * C1 + j] = j < C1 - 1 ? 1 : 0;
    }
    if (b == 0 && j == 0 && rng) *rng += 1;
}

static int csm_head(vox_csm* m, hipStream_t st, const vox_csm_io* io, int B, const int32_t* x_rows) {
    const vox_csm_config& c = m->cfg;
    const int H = c.backbone.hidden;
    LinearCall h;  // backbone final norm (-> hidden row of the depth input) + lm_head
    h.W = m->w.lm_head; h.x = m->x; h.x_rows = x_rows; h.norm_w = m->w.backbone_norm; h.eps = c.backbone.eps;
    h.x_out = m->depth_x; h.x_out_stride = 2L * H; h.y = io->out_logits; h.B = B; h.N = c.vocab; h.K = H;
    h.pro = VOX_PRO_RMSNORM; h.epi = VOX_EPI_STORE;
    VOX_TRY(vox_launch_linear(m->ctx, st, h));
    if (io->out_hidden)
        hipLaunchKernelGGL(k_copy_rows, dim3((H + 255) / 256, B), dim3(256), 0, st, (const bf16_t*)m->depth_x, 2L * H,
                           (bf16_t*)io->out_hidden, (long)H, H);
    return VOX_OK;
}

static int csm_tail(vox_csm* m, hipStream_t st, const vox_csm_io* io, int B, const vox_sampling_config* sc, uint64_t seed,
                    int feedback) {
    const vox_csm_config& c = m->cfg;
    const int H = c.backbone.hidden, Hd = c.depth.hidden, C = c.n_codebooks, C1 = C + 1, V = c.vocab;
    {
        SampleCall s;   // codebook 0 from the backbone logits; its embedding is the second row of the depth input
        s.logits = io->out_logits; s.B = B; s.V = V; s.cfg = *sc; s.cfg.repetition_penalty = 1.0f;   // never persisted (Q3)
        s.seed = seed; s.offset = 0; s.offset_dev = io->rng_offset; s.offset_mul = (uint64_t)C;
        s.out_ids = io->out_ids; s.out_stride = C1; s.out_col = 0;
        s.emb_table = m->w.audio_embedding; s.emb_vocab = V; s.H = H;
        s.emb_dst = (bf16_t*)m->depth_x + H; s.emb_dst_stride = 2L * H; s.ws = m->ctx->samp_ws;
        VOX_TRY(vox_launch_sample(st, s));
    }
    hipLaunchKernelGGL(k_csm_fill_row, dim3(B), dim3(64), 0, st, io->out_ids, C1, B);
    for (int i = 1; i < C; ++i) {
        const int rows = i == 1 ? 2 * B : B;
        if (i == 1 || !m->proj_tab) {     // (steps >= 2: the previous step's sampler gathered the tabulated projection into dx)
            LinearCall p;  // inputs_embeds_projector (no bias)
            p.W = m->w.depth_proj; p.x = m->depth_x; p.y = m->dx; p.B = rows; p.N = Hd; p.K = H;
            p.pro = VOX_PRO_COPY; p.epi = VOX_EPI_STORE; p.fixed_order = B <= m->ctx->exact_rows; p.exact_rows = m->ctx->exact_rows; p.keep_weights = m->depth->keep_weights;
            p.splitk_ws = m->depth->skws; p.splitk_ws_bytes = m->depth->skws_bytes;
            VOX_TRY(vox_launch_linear(m->ctx, st, p));
        }
        vox_rows r{};
        r.pos = i == 1 ? m->d1_pos : m->di_pos[i];
        r.q_req = i == 1 ? m->d1_req : m->iota;
        r.q_kvlen = i == 1 ? m->d1_kvlen : m->di_kvlen[i];
        r.page = r.q_req;
        r.slot = i == 1 ? m->d1_pos : m->di_pos[i];
        r.kv_indptr = m->d_indptr;
        r.kv_indices = m->iota;
        r.n_rows = rows;
        r.max_kvlen = i + 1;
        if (i > 1) { r.fixed_kvlen = i + 1; r.fixed_pos = i; r.identity_pages = 1; }
        VOX_TRY(stack_layers(m->depth, st, m->dx, m->dkv, m->dkv_stride, &r, i > 1, B <= m->ctx->exact_rows));
        void* dl = io->out_depth_logits ? (void*)((bf16_t*)io->out_depth_logits + (size_t)(i - 1) * c.max_batch * V) : m->dlogits;
        LinearCall h;  // depth final norm + codebooks_head[i-1]
        h.W = (const bf16_t*)m->w.depth_heads + (size_t)(i - 1) * V * Hd;
        h.x = m->dx; h.x_rows = i == 1 ? m->odd_rows : nullptr; h.norm_w = m->w.depth_norm; h.eps = c.depth.eps;
        h.y = dl; h.B = B; h.N = V; h.K = Hd; h.pro = VOX_PRO_RMSNORM; h.epi = VOX_EPI_STORE;
        h.keep_weights = m->depth->keep_weights;
        VOX_TRY(vox_launch_linear(m->ctx, st, h));
        SampleCall s;
        s.logits = dl; s.B = B; s.V = V; s.cfg = *sc; s.cfg.repetition_penalty = 1.0f;
        s.seed = seed; s.offset = (uint64_t)i; s.offset_dev = io->rng_offset; s.offset_mul = (uint64_t)C;
        s.out_ids = io->out_ids; s.out_stride = C1; s.out_col = i;
        s.emb_table = (const bf16_t*)m->w.audio_embedding + (size_t)i * V * H; s.emb_vocab = V; s.H = H;   // embed_audio_tokens_single(ids, i)
        s.emb_dst = m->depth_x; s.emb_dst_stride = H; s.ws = m->ctx->samp_ws;
        if (m->proj_tab) {
            s.emb_table = nullptr; s.emb_dst = nullptr;        // the raw embedding has no other consumer
            if (i + 1 < C) {
                s.emb2_table = (const bf16_t*)m->proj_tab + (size_t)(i - 1) * V * Hd; s.H2 = Hd;
                s.emb2_dst = m->dx; s.emb2_dst_stride = Hd;
            }
        }
        VOX_TRY(vox_launch_sample(st, s));
    }
    if (feedback)
        hipLaunchKernelGGL(k_csm_feedback, dim3(B), dim3(64), 0, st, io->out_ids, io->input_ids, io->input_masks, io->rng_offset, C1, B);
    else if (io->rng_offset)
        hipLaunchKernelGGL(k_rng_bump, dim3(1), dim3(1), 0, st, io->rng_offset, (int32_t*)nullptr, (const unsigned*)nullptr, (const unsigned*)nullptr);
    return VOX_OK;
}

static void csm_embed(vox_csm* m, hipStream_t st, const int32_t* ids, const uint8_t* masks, int n) {
    const vox_csm_config& c = m->cfg;
    const int H = c.backbone.hidden;
    hipLaunchKernelGGL(k_csm_embed, dim3((H + 255) / 256, n), dim3(256), 0, st, ids, masks, (const bf16_t*)m->w.audio_embedding,
                       (const bf16_t*)m->w.text_embedding, (bf16_t*)m->x, c.n_codebooks, c.vocab, c.text_vocab, H);
}

extern "C" {

int vox_csm_create(vox_ctx* ctx, const vox_csm_config* cfg, const vox_csm_weights* w, vox_csm** out) {
    if (!ctx || !cfg || !w || !out) return vox_fail(VOX_ERR_INVALID, "csm_create: NULL argument");
    if (cfg->n_codebooks < 2 || cfg->n_codebooks > 63 || cfg->max_batch < 1) return vox_fail(VOX_ERR_INVALID, "csm_create: bad config");
    vox_csm* m = new vox_csm();
    m->ctx = ctx; m->cfg = *cfg; m->w = *w;
    const int B = cfg->max_batch, C = cfg->n_codebooks, H = cfg->backbone.hidden, Hd = cfg->depth.hidden;
    vox_stack_config bc = cfg->backbone, dc = cfg->depth;
    dc.page_size = C; dc.max_rows = 2 * B; dc.max_kvlen = C;
    int s = vox_stack_create(ctx, &bc, w->backbone_layers, w->backbone_norm, w->backbone_rope, w->backbone_rope_max_pos, &m->backbone);
    if (s != VOX_OK) { delete m; return s; }
    s = vox_stack_create(ctx, &dc, w->depth_layers, w->depth_norm, w->depth_rope, w->depth_rope_max_pos, &m->depth);
    if (s != VOX_OK) { vox_stack_destroy(m->backbone); delete m; return s; }
    m->depth->keep_weights = 1;   // re-read by each of the 31 dependent steps: keep them in the Infinity Cache
    const size_t R = bc.max_rows;
    m->dkv_stride = (int64_t)B * 2 * C * dc.kv_heads * dc.head_dim;
    bool ok = hipMalloc(&m->x, R * H * 2) == hipSuccess && hipMalloc(&m->depth_x, (size_t)2 * B * H * 2) == hipSuccess &&
              hipMalloc(&m->dx, (size_t)2 * B * Hd * 2) == hipSuccess &&
              hipMalloc(&m->dlogits, (size_t)B * cfg->vocab * 2) == hipSuccess &&
              hipMalloc(&m->dkv, (size_t)dc.layers * m->dkv_stride * 2) == hipSuccess;
    if (!ok) return vox_fail(VOX_ERR_NOMEM, "csm_create: hipMalloc failed");
    VOX_HIP(hipMemset(m->dkv, 0, (size_t)dc.layers * m->dkv_stride * 2));
    const int niota = (int)(R > (size_t)2 * B + 1 ? R : 2 * B + 1);
    std::vector<int32_t> host;
    auto push = [&](const std::vector<int32_t>& v) { size_t o = host.size(); host.insert(host.end(), v.begin(), v.end()); return o; };
    std::vector<int32_t> iota(niota), d1p(2 * B), d1r(2 * B), d1k(2 * B), odd(B), indptr(B + 1);
    for (int i = 0; i < niota; ++i) iota[i] = i;
    for (int b = 0; b < B; ++b) {
        d1p[2 * b] = 0; d1p[2 * b + 1] = 1; d1r[2 * b] = d1r[2 * b + 1] = b;
        d1k[2 * b] = 1; d1k[2 * b + 1] = 2; odd[b] = 2 * b + 1;
    }
    for (int b = 0; b <= B; ++b) indptr[b] = b;
    const size_t o_iota = push(iota), o_d1p = push(d1p), o_d1r = push(d1r), o_d1k = push(d1k), o_odd = push(odd), o_ind = push(indptr);
    std::vector<size_t> o_pos(C, 0), o_kvl(C, 0);
    for (int i = 2; i < C; ++i) {
        o_pos[i] = push(std::vector<int32_t>(B, i));
        o_kvl[i] = push(std::vector<int32_t>(B, i + 1));
    }
    if (hipMalloc((void**)&m->dmeta, host.size() * 4) != hipSuccess) return vox_fail(VOX_ERR_NOMEM, "csm_create: hipMalloc");
    VOX_HIP(hipMemcpy(m->dmeta, host.data(), host.size() * 4, hipMemcpyHostToDevice));
    m->iota = m->dmeta + o_iota; m->d1_pos = m->dmeta + o_d1p; m->d1_req = m->dmeta + o_d1r; m->d1_kvlen = m->dmeta + o_d1k;
    m->odd_rows = m->dmeta + o_odd; m->d_indptr = m->dmeta + o_ind;
    m->di_pos.assign(C, nullptr); m->di_kvlen.assign(C, nullptr);
    for (int i = 2; i < C; ++i) { m->di_pos[i] = m->dmeta + o_pos[i]; m->di_kvlen[i] = m->dmeta + o_kvl[i]; }
    if (C > 2) {
        // projector(embedding) tables for depth steps 2..C-1 (fixed-order kernel: bit-identical to the per-step launch they
        // replace; C-2 launches fewer per frame)
        const size_t tab = (size_t)cfg->vocab * Hd;
        if (hipMalloc(&m->proj_tab, (size_t)(C - 2) * tab * 2) != hipSuccess) return vox_fail(VOX_ERR_NOMEM, "csm_create: hipMalloc");
        for (int i = 1; i + 1 < C; ++i) {
            LinearCall p;
            p.W = w->depth_proj; p.x = (const bf16_t*)w->audio_embedding + (size_t)i * cfg->vocab * H;
            p.y = (bf16_t*)m->proj_tab + (size_t)(i - 1) * tab;
            p.B = cfg->vocab; p.N = Hd; p.K = H; p.pro = VOX_PRO_COPY; p.epi = VOX_EPI_STORE; p.fixed_order = 1;
            const int rc = vox_launch_linear(ctx, nullptr, p);
            if (rc != VOX_OK) return rc;
        }
        VOX_HIP(hipDeviceSynchronize());
    }
    *out = m;
    return VOX_OK;
}
void vox_csm_destroy(vox_csm* m) {
    if (!m) return;
    vox_stack_destroy(m->backbone); vox_stack_destroy(m->depth);
    (void)hipFree(m->proj_tab);
    for (void* p : {m->x, m->depth_x, m->dx, m->dlogits, m->dkv, (void*)m->dmeta}) (void)hipFree(p);
    delete m;
}
int vox_csm_frame(vox_csm* m, void* stream, const vox_csm_io* io, int B, int max_kvlen, const vox_sampling_config* sc,
                  uint64_t seed, int feedback) {
    if (!m || !io || !sc) return vox_fail(VOX_ERR_INVALID, "csm_frame: NULL");
    if (B < 1 || B > m->cfg.max_batch) return vox_fail(VOX_ERR_INVALID, "csm_frame: batch %d > max_batch", B);
    hipStream_t st = (hipStream_t)stream;
    csm_embed(m, st, io->input_ids, io->input_masks, B);
    vox_rows r{};
    r.pos = io->pos; r.q_req = m->iota; r.q_kvlen = io->kvlen; r.page = io->page; r.slot = io->slot;
    r.kv_indptr = io->kv_indptr; r.kv_indices = io->kv_indices; r.n_rows = B; r.max_kvlen = max_kvlen;
    r.page_table = io->page_table; r.pt_stride = (int32_t)io->pt_stride;
    VOX_TRY(stack_layers(m->backbone, st, m->x, io->kv, io->kv_layer_stride, &r, true, B <= m->ctx->exact_rows));
    VOX_TRY(csm_head(m, st, io, B, nullptr));
    return csm_tail(m, st, io, B, sc, seed, feedback);
}
int vox_csm_prefill(vox_csm* m, void* stream, const vox_csm_io* io, const int32_t* row_ids, const uint8_t* row_masks,
                    const int32_t* q_req, int n_rows, const int32_t* last_rows, int n_req, int max_kvlen,
                    const vox_sampling_config* sc, uint64_t seed, int feedback) {
    if (!m || !io || !sc) return vox_fail(VOX_ERR_INVALID, "csm_prefill: NULL");
    if (n_req < 0 || n_req > m->cfg.max_batch || n_rows > m->cfg.backbone.max_rows)
        return vox_fail(VOX_ERR_INVALID, "csm_prefill: n_req %d / n_rows %d out of range", n_req, n_rows);
    hipStream_t st = (hipStream_t)stream;
    csm_embed(m, st, row_ids, row_masks, n_rows);
    vox_rows r{};
    r.pos = io->pos; r.q_req = q_req; r.q_kvlen = io->kvlen; r.page = io->page; r.slot = io->slot;
    r.kv_indptr = io->kv_indptr; r.kv_indices = io->kv_indices; r.n_rows = n_rows; r.max_kvlen = max_kvlen;
    VOX_TRY(stack_layers(m->backbone, st, m->x, io->kv, io->kv_layer_stride, &r, false, n_rows <= m->ctx->exact_rows));
    if (n_req == 0) return VOX_OK;      // context chunk of a long prompt
    VOX_TRY(csm_head(m, st, io, n_req, last_rows));
    return csm_tail(m, st, io, n_req, sc, seed, feedback);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------
// single-stack LM engine (GLM-4-Voice, CosyVoice2, ...)
// ---------------------------------------------------------------------------------------------------
struct vox_lm {
    vox_ctx* ctx;
    vox_lm_config cfg;
    vox_lm_weights w;
    vox_stack* stack;
    void *x, *hidden;
    int32_t* iota;
};

// x[b] = mask[b] ? feat[b] : x[b]     (cosyvoice2.py:1024)
__global__ __launch_bounds__(256) void k_where_rows(bf16_t* x, const uint8_t* mask, const bf16_t* feat, int H) {
    const int b = blockIdx.y;
    if (!mask[b]) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < H; i += gridDim.x * 256) x[(size_t)b * H + i] = feat[(size_t)b * H + i];
}
// next step inputs: ids[b,0] = sampled id; (mode 1) mask = 0 (features no longer used: cosyvoice2.py:1062-1064)
__global__ void k_lm_feedback(const int* out_ids, int* input_ids, int stride, uint8_t* masks, uint64_t* rng, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        input_ids[(size_t)b * stride] = out_ids[b];
        if (masks) masks[b] = 0;
    }
    if (b == 0 && rng) *rng += 1;
}

static int lm_run(vox_lm* m, hipStream_t st, const vox_lm_io* io, const int32_t* ids, const uint8_t* masks,
                  const void* feats, const vox_rows& r, bool decode_rows, const int32_t* last_rows, int n_req,
                  const vox_sampling_config* sc, uint64_t seed, int feedback) {
    const vox_lm_config& c = m->cfg;
    const int H = c.stack.hidden, n = r.n_rows;
    VOX_TRY(vox_launch_gather(st, m->w.embedding, ids, c.ids_stride, 0, m->x, H, n, H, c.vocab_in));
    if (c.input_mode == 1 && masks && feats)
        hipLaunchKernelGGL(k_where_rows, dim3((H + 255) / 256, n), dim3(256), 0, st, (bf16_t*)m->x, masks, (const bf16_t*)feats, H);
    VOX_TRY(stack_layers(m->stack, st, m->x, io->kv, io->kv_layer_stride, &r, decode_rows, n <= m->ctx->exact_rows));
    if (n_req == 0) return VOX_OK;      // context chunk of a long prompt
    LinearCall h;   // final norm + output head
    h.W = m->w.head_w; h.bias = m->w.head_b; h.x = m->x; h.x_rows = last_rows; h.norm_w = m->w.final_norm; h.eps = c.stack.eps;
    h.y = io->out_logits; h.B = n_req; h.N = c.vocab_out; h.K = H; h.pro = VOX_PRO_RMSNORM; h.epi = VOX_EPI_STORE;
    h.fixed_order = n_req <= m->ctx->exact_rows; h.exact_rows = m->ctx->exact_rows;
    h.norm_scratch = m->stack->xn;       // (wide models: the head's norm prologue may be normalised once into the stack's scratch)
    VOX_TRY(vox_launch_linear(m->ctx, st, h));
    SampleCall s;
    s.logits = io->out_logits; s.B = n_req; s.V = c.vocab_out; s.cfg = *sc; s.seed = seed; s.offset = 0;
    s.offset_dev = io->rng_offset; s.offset_mul = 1; s.out_ids = io->out_ids; s.out_stride = 1; s.out_col = 0;
    s.ws = m->ctx->samp_ws;
    if (io->rep_cache && sc->repetition_penalty != 1.0f) { s.rep_cache = io->rep_cache; s.W = io->rep_w; s.C = 1; }
    else s.cfg.repetition_penalty = 1.0f;
    VOX_TRY(vox_launch_sample(st, s));
    if (io->rep_cache && sc->repetition_penalty != 1.0f)
        VOX_TRY(vox_launch_rep_update(st, io->rep_cache, io->out_ids, n_req, io->rep_w, 1, c.vocab_out, io->rep_window));
    if (feedback)
        hipLaunchKernelGGL(k_lm_feedback, dim3((n_req + 63) / 64), dim3(64), 0, st, io->out_ids, io->input_ids, c.ids_stride,
                           c.input_mode == 1 ? io->input_masks : nullptr, io->rng_offset, n_req);
    else if (io->rng_offset)
        hipLaunchKernelGGL(k_rng_bump, dim3(1), dim3(1), 0, st, io->rng_offset, (int32_t*)nullptr, (const unsigned*)nullptr, (const unsigned*)nullptr);
    return VOX_OK;
}

extern "C" {

int vox_lm_create(vox_ctx* ctx, const vox_lm_config* cfg, const vox_lm_weights* w, vox_lm** out) {
    if (!ctx || !cfg || !w || !out) return vox_fail(VOX_ERR_INVALID, "lm_create: NULL argument");
    vox_lm* m = new vox_lm();
    m->ctx = ctx; m->cfg = *cfg; m->w = *w;
    int s = vox_stack_create(ctx, &cfg->stack, w->layers, w->final_norm, w->rope, w->rope_max_pos, &m->stack);
    if (s != VOX_OK) { delete m; return s; }
    const size_t R = cfg->stack.max_rows;
    std::vector<int32_t> iota(R);
    for (size_t i = 0; i < R; ++i) iota[i] = (int32_t)i;
    if (hipMalloc(&m->x, R * cfg->stack.hidden * 2) != hipSuccess || hipMalloc((void**)&m->iota, R * 4) != hipSuccess)
        return vox_fail(VOX_ERR_NOMEM, "lm_create: hipMalloc failed");
    VOX_HIP(hipMemcpy(m->iota, iota.data(), R * 4, hipMemcpyHostToDevice));
    *out = m;
    return VOX_OK;
}
void vox_lm_destroy(vox_lm* m) {
    if (!m) return;
    vox_stack_destroy(m->stack);
    (void)hipFree(m->x); (void)hipFree(m->iota);
    delete m;
}
int vox_lm_frame(vox_lm* m, void* stream, const vox_lm_io* io, int B, int max_kvlen, const vox_sampling_config* sc,
                 uint64_t seed, int feedback) {
    if (!m || !io || !sc) return vox_fail(VOX_ERR_INVALID, "lm_frame: NULL");
    if (B < 1 || B > m->cfg.max_batch) return vox_fail(VOX_ERR_INVALID, "lm_frame: batch %d > max_batch", B);
    vox_rows r{};
    r.pos = io->pos; r.q_req = m->iota; r.q_kvlen = io->kvlen; r.page = io->page; r.slot = io->slot;
    r.kv_indptr = io->kv_indptr; r.kv_indices = io->kv_indices; r.n_rows = B; r.max_kvlen = max_kvlen;
    r.page_table = io->page_table; r.pt_stride = (int32_t)io->pt_stride;
    return lm_run(m, (hipStream_t)stream, io, io->input_ids, io->input_masks, io->input_features, r, true, nullptr, B, sc,
                  seed, feedback);
}
int vox_lm_prefill(vox_lm* m, void* stream, const vox_lm_io* io, const int32_t* row_ids, const uint8_t* row_masks,
                   const void* row_features, const int32_t* q_req, int n_rows, const int32_t* last_rows, int n_req,
                   int max_kvlen, const vox_sampling_config* sc, uint64_t seed, int feedback) {
    if (!m || !io || !sc) return vox_fail(VOX_ERR_INVALID, "lm_prefill: NULL");
    if (n_req < 0 || n_req > m->cfg.max_batch || n_rows > m->cfg.stack.max_rows)
        return vox_fail(VOX_ERR_INVALID, "lm_prefill: n_req %d / n_rows %d out of range", n_req, n_rows);
    vox_rows r{};
    r.pos = io->pos; r.q_req = q_req; r.q_kvlen = io->kvlen; r.page = io->page; r.slot = io->slot;
    r.kv_indptr = io->kv_indptr; r.kv_indices = io->kv_indices; r.n_rows = n_rows; r.max_kvlen = max_kvlen;
    return lm_run(m, (hipStream_t)stream, io, row_ids, row_masks, row_features, r, false, last_rows, n_req, sc, seed, feedback);
}

}  // extern "C"
