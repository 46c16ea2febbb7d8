"""ctypes binding of libvoxhip.so (the C ABI declared in include/voxhip.h).

The HIP library is the product path: there is NO CPU fallback.  Loading fails loudly when the shared
object is missing, and every compute entry point raises if no MI355X context can be created.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VOX_LIB", os.path.join(HERE, "libvoxhip.so"))   # VOX_LIB: development override


class VoxError(RuntimeError):
    pass


class SamplingCfg(Structure):
    _fields_ = [("greedy", c_int32), ("top_k", c_int32), ("top_p", c_float), ("min_p", c_float),
                ("temperature", c_float), ("repetition_penalty", c_float)]


class StackConfig(Structure):
    _fields_ = [(n, c_int32) for n in ("hidden", "layers", "heads", "kv_heads", "head_dim", "ffn")] + \
               [("eps", c_float)] + \
               [(n, c_int32) for n in ("qk_norm", "qkv_bias", "rope_dim", "rope_interleave", "page_size",
                                       "max_rows", "max_kvlen")]


class LayerWeights(Structure):
    _fields_ = [(n, c_void_p) for n in ("wqkv", "bqkv", "wo", "wgate", "wup", "wdown", "ln1", "ln2", "qnorm", "knorm")]


class Rows(Structure):
    _fields_ = [(n, c_void_p) for n in ("pos", "q_req", "q_kvlen", "page", "slot", "kv_indptr", "kv_indices")] + \
               [("n_rows", c_int32), ("max_kvlen", c_int32), ("page_table", c_void_p), ("pt_stride", c_int32),
                ("fixed_kvlen", c_int32), ("fixed_pos", c_int32), ("identity_pages", c_int32)]


class Qwen3Config(Structure):
    _fields_ = [("talker", StackConfig), ("depth", StackConfig)] + \
               [(n, c_int32) for n in ("vocab", "text_vocab", "text_hidden", "depth_vocab", "n_groups", "eos_id",
                                       "tts_pad_id", "max_batch")]


class Qwen3Weights(Structure):
    _fields_ = [("talker_layers", POINTER(LayerWeights)), ("depth_layers", POINTER(LayerWeights)),
                ("talker_norm", c_void_p), ("depth_norm", c_void_p), ("codec_embedding", c_void_p),
                ("text_embedding", c_void_p), ("tp_fc1_w", c_void_p), ("tp_fc1_b", c_void_p), ("tp_fc2_w", c_void_p),
                ("tp_fc2_b", c_void_p), ("codec_head", c_void_p), ("depth_codec_embedding", POINTER(c_void_p)),
                ("depth_lm_head", c_void_p), ("mtp_w", c_void_p), ("mtp_b", c_void_p), ("talker_rope", c_void_p),
                ("depth_rope", c_void_p), ("talker_rope_max_pos", c_int32), ("depth_rope_max_pos", c_int32)]


class Qwen3IO(Structure):
    _fields_ = [("input_ids", c_void_p), ("input_masks", c_void_p), ("input_features", c_void_p), ("pos", c_void_p),
                ("kvlen", c_void_p), ("page", c_void_p), ("slot", c_void_p), ("kv_indptr", c_void_p),
                ("kv_indices", c_void_p), ("page_table", c_void_p), ("pt_stride", c_int64), ("kv", c_void_p),
                ("kv_layer_stride", c_int64), ("out_ids", c_void_p),
                ("out_logits", c_void_p), ("out_hidden", c_void_p), ("out_depth_logits", c_void_p),
                ("next_features", c_void_p), ("rng_offset", c_void_p)]


_SIGS = {
    "vox_abi_version": (c_int, []),
    "vox_last_error": (c_char_p, []),
    "vox_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
    "vox_ctx_destroy": (None, [c_void_p]),
    "vox_ctx_props": (c_int, [c_void_p, POINTER(c_int64)]),
    "vox_graph_begin": (c_int, [c_void_p, c_void_p]),
    "vox_graph_end": (c_int, [c_void_p, c_void_p, POINTER(c_void_p)]),
    "vox_graph_launch": (c_int, [c_void_p, c_void_p]),
    "vox_graph_destroy": (None, [c_void_p]),
    "vox_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float]),
    "vox_rope_table_host": (c_int, [c_void_p, c_int, c_int, c_double, c_double, c_int, c_double, c_double, c_int]),
    "vox_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                         c_int, c_int, c_int, c_void_p, c_int]),
    "vox_kv_append": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                              c_int, c_int]),
    "vox_attn_workspace_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "vox_paged_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float]),
    "vox_linear": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                           c_int]),
    "vox_linear_silu_mul": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "vox_suppress": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int]),
    "vox_rep_penalty": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float]),
    "vox_rep_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "vox_rep_penalty_mc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float]),
    "vox_rep_update_mc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int]),
    "vox_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, POINTER(SamplingCfg), c_uint64, c_uint64,
                           c_void_p]),
    "vox_stack_create": (c_int, [c_void_p, POINTER(StackConfig), POINTER(LayerWeights), c_void_p, c_void_p, c_int,
                                 POINTER(c_void_p)]),
    "vox_stack_destroy": (None, [c_void_p]),
    "vox_stack_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, POINTER(Rows)]),
    "vox_qwen3_create": (c_int, [c_void_p, POINTER(Qwen3Config), POINTER(Qwen3Weights), POINTER(c_void_p)]),
    "vox_qwen3_destroy": (None, [c_void_p]),
    "vox_qwen3_frame": (c_int, [c_void_p, c_void_p, POINTER(Qwen3IO), c_int, c_int, POINTER(SamplingCfg), c_uint64,
                                c_int]),
    "vox_qwen3_prefill": (c_int, [c_void_p, c_void_p, POINTER(Qwen3IO), c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p, c_int, c_int, POINTER(SamplingCfg), c_uint64, c_int]),
    "vox_qwen3_prompt_features": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "vox_qwen3_depth_persist_status": (c_int, [c_void_p, POINTER(c_int32), POINTER(ctypes.c_uint32)]),
    "vox_qwen3_set_status": (c_int, [c_void_p, c_void_p]),
    "vox_qwen3_frame_restore": (c_int, [c_void_p, c_void_p, POINTER(Qwen3IO), c_int, c_int]),
    "vox_qwen3_persist_reset": (c_int, [c_void_p, c_int]),
    "vox_qwen3_persist_set_spins": (c_int, [c_void_p, ctypes.c_uint32]),
    "vox_qwen3_persist_inject": (c_int, [c_void_p, c_int, ctypes.c_uint32]),
}

# symbols added by later translation units (codec); bound if present, listed so the export test sees them
_OPTIONAL_SIGS = {}

_lib = None
_capturing = set()   # devices with a stream capture in progress (graph_capture)
_ctxs = {}          # one context per HIP device of this process (the LM's GPU; a detokenizer on a second GPU gets its own)


def lib():
    """Load libvoxhip.so.  Raises (never falls back) if the shared object is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VoxError(f"{LIB_PATH} not found: build it with `python -m vox_serve_amd.build` "
                           "(the HIP library is the only compute path; there is no CPU fallback)")
        # torch first: its wheel carries its own HIP runtime, and the process must have ONE — with libvoxhip (linked against
        # /opt/rocm's libamdhip64) loaded before torch, the two runtimes coexist and hipGetDeviceCount in ours reports no device
        # (seen with `python __graft_entry__.py smoke`, whose build() step loaded the library before anything imported torch)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in {**_SIGS, **_OPTIONAL_SIGS}.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.vox_abi_version() != 1:
            raise VoxError("libvoxhip ABI version mismatch")
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise VoxError(f"libvoxhip error {status}: {lib().vox_last_error().decode()}")


def ctx():
    """The context of the CURRENT torch device (created on first use).  One process per GPU is the norm; a plugin whose
    audio_decoder_device differs from its LM device runs its detokenizer calls under `device_guard`, and they land here
    with that device current — its own context, workspace and streams."""
    import torch
    if not torch.cuda.is_available():
        raise VoxError("no HIP device visible: libvoxhip needs an MI355X (there is no CPU fallback)")
    d = torch.cuda.current_device()
    h = _ctxs.get(d)
    if h is None:
        h = c_void_p()
        check(lib().vox_ctx_create(d, ctypes.byref(h)))
        _ctxs[d] = h
    return h


def device_guard(device):
    """Context manager making `device` the current HIP device (no-op for CPU / index-less devices)."""
    import contextlib
    import torch
    d = torch.device(device)
    if d.type != "cuda" or d.index is None or not torch.cuda.is_available():
        return contextlib.nullcontext()
    return torch.cuda.device(d)


def set_exact_rows(rows: int):
    """Linears with at most `rows` rows (1..8, default 2) run the wave64 VALU kernels, more rows the matrix cores; every setting
    is bit-exact against the oracle under the same policy (see include/voxhip.h).  Call before the engines are created."""
    L = lib()
    L.vox_ctx_set_exact_rows.restype = ctypes.c_int
    L.vox_ctx_set_exact_rows.argtypes = [c_void_p, ctypes.c_int]
    check(L.vox_ctx_set_exact_rows(ctx(), int(rows)))


def stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


_capture_streams = {}


class graph_capture:
    """`with graph_capture() as cap: <enqueue on N.stream()>` -> `cap.graph` (a vox_graph handle).

    Stream capture needs a non-default stream, so the body runs on a private per-device stream, fenced against the caller's
    current stream on both sides — a one-time cost per captured shape.  REPLAYS (`vox_graph_launch(g, N.stream())`) and eager
    calls go on the caller's own current stream: engines and detokenizers own no stream of their own.  That is deliberate and
    measured: while a graph runs, a barrier packet pending on a SECOND hardware queue (what `other.wait_stream(engine_stream)`
    leaves behind until the graph finishes) makes the command processor alternate between the queues, and every dependent
    dispatch of the running graph costs about 1.3 us more on an MI355X — 0.5 ms of a 3.3 ms Qwen3-TTS frame of ~510 dependent
    kernels (profiles/round3_queue_fence.txt).  Whoever wants two graphs side by side (LM frame + codec chunk) puts them on
    two streams itself and orders them with events it waits for on the HOST."""

    def __init__(self):
        import torch
        self._torch = torch
        d = torch.cuda.current_device()
        if d not in _capture_streams:
            _capture_streams[d] = torch.cuda.Stream(device=d)
        self.cs = _capture_streams[d]
        self.graph = None

    def __enter__(self):
        torch = self._torch
        d = torch.cuda.current_device()
        if d in _capturing:      # the capture stream is one per device and not re-entrant
            raise VoxError("graph_capture: a capture is already active on this device (nested or concurrent captures are not supported)")
        _capturing.add(d)
        self._dev = d
        self.cur = torch.cuda.current_stream()
        self.cs.wait_stream(self.cur)
        self._ctx = torch.cuda.stream(self.cs)
        self._ctx.__enter__()
        try:
            check(lib().vox_graph_begin(ctx(), stream()))
        except Exception:
            self._ctx.__exit__(None, None, None)
            _capturing.discard(self._dev)
            raise
        return self

    def __exit__(self, et, ev, tb):
        gh = c_void_p()
        try:
            status = lib().vox_graph_end(ctx(), stream(), ctypes.byref(gh))    # capture must end even when the body raised
        finally:
            self._ctx.__exit__(None, None, None)
            self.cur.wait_stream(self.cs)
            _capturing.discard(self._dev)
        if et is None:
            check(status)
            self.graph = gh
        else:
            # the body raised: its exception propagates; a graph that was instantiated all the same is released, and a failed
            # end-of-capture is at least reported
            if gh:
                lib().vox_graph_destroy(gh)
            if status != 0:
                import logging
                logging.getLogger(__name__).error("graph_capture: vox_graph_end failed (%s) while unwinding %s", status, et.__name__)
        return False


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL).  Tensors must be contiguous."""
    if t is None:
        return None
    assert t.is_contiguous(), "libvoxhip takes contiguous tensors"
    return c_void_p(t.data_ptr())
