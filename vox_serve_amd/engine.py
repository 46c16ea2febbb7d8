"""Host side of the Qwen3-TTS frame engine (libvoxhip `vox_qwen3_*`).

Mirrors what CudaGraphWorker does for the LM hot loop in the reference
(/root/reference/vox_serve/worker/cuda_graph_worker.py:57-183 buffers, :353-486 decode graphs,
:946-1160 run_lm_decode + run_lm_depth) with one difference in kind: a whole audio frame — talker
step, codebook-0 sampling, 15 depth steps with their sampling and embedding feedback — is ONE hipGraph
replay with no host synchronisation inside; the host only uploads the few plan() integers per frame.
"""
import ctypes
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

from . import _native as N


@dataclass
class StackCfg:
    hidden: int
    layers: int
    heads: int
    kv_heads: int
    head_dim: int
    ffn: int
    eps: float = 1e-6
    rope_theta: float = 1e6
    rope_scale: float = 1.0
    rope_dim: Optional[int] = None
    rope_interleave: bool = False
    rope_llama31: Optional[tuple] = None
    qk_norm: bool = True
    qkv_bias: bool = False


@dataclass
class Qwen3Cfg:
    """Defaults = Qwen3-TTS-12Hz-1.7B (reference dataclass defaults, model/qwen3_tts.py:112-253)."""
    talker: StackCfg = field(default_factory=lambda: StackCfg(2048, 28, 16, 8, 128, 6144))
    depth: StackCfg = field(default_factory=lambda: StackCfg(1024, 5, 16, 8, 128, 3072))
    vocab: int = 3072
    text_vocab: int = 151936
    text_hidden: int = 2048
    depth_vocab: int = 2048
    n_groups: int = 16
    eos_id: int = 2150
    tts_pad_id: int = 151671
    max_pos: int = 4096


def rope_table(max_pos, c: StackCfg, device):
    rot = c.rope_dim or c.head_dim
    host = np.empty((max_pos, rot // 2, 2), np.float32)
    lo, hi, ctx = c.rope_llama31 if c.rope_llama31 else (1.0, 4.0, 8192)
    N.check(N.lib().vox_rope_table_host(host.ctypes.data_as(ctypes.c_void_p), max_pos, rot, float(c.rope_theta),
                                        float(c.rope_scale), 1 if c.rope_llama31 else 0, float(lo), float(hi), int(ctx)))
    return torch.from_numpy(host).to(device)


def _stack_config(c: StackCfg, page_size, max_rows, max_kvlen) -> N.StackConfig:
    return N.StackConfig(c.hidden, c.layers, c.heads, c.kv_heads, c.head_dim, c.ffn, c.eps, int(c.qk_norm),
                         int(c.qkv_bias), c.rope_dim or c.head_dim, int(c.rope_interleave), page_size, max_rows,
                         max_kvlen)


def pack_stack_weights(W: Dict[str, torch.Tensor], prefix: str, c: StackCfg, keep: list):
    """Reference state_dict names -> per-layer pointer structs; q/k/v rows concatenated (layout only)."""
    arr = (N.LayerWeights * c.layers)()
    for i in range(c.layers):
        p = f"{prefix}.layers.{i}."
        wqkv = torch.cat([W[p + "self_attn.q_proj.weight"], W[p + "self_attn.k_proj.weight"],
                          W[p + "self_attn.v_proj.weight"]], 0).contiguous()
        bqkv = None
        if c.qkv_bias:
            bqkv = torch.cat([W[p + "self_attn.q_proj.bias"], W[p + "self_attn.k_proj.bias"],
                              W[p + "self_attn.v_proj.bias"]], 0).contiguous()
        ts = dict(wqkv=wqkv, bqkv=bqkv, wo=W[p + "self_attn.o_proj.weight"], wgate=W[p + "mlp.gate_proj.weight"],
                  wup=W[p + "mlp.up_proj.weight"], wdown=W[p + "mlp.down_proj.weight"],
                  ln1=W[p + "input_layernorm.weight"], ln2=W[p + "post_attention_layernorm.weight"],
                  qnorm=W.get(p + "self_attn.q_norm.weight"), knorm=W.get(p + "self_attn.k_norm.weight"))
        for k, t in ts.items():
            if t is not None:
                t = t.contiguous()
                keep.append(t)
                setattr(arr[i], k, t.data_ptr())
    return arr


class Qwen3Engine:
    def __init__(self, cfg: Qwen3Cfg, weights: Dict[str, torch.Tensor], max_batch=8, page_size=128, max_pages=256,
                 max_seq_len=2304, max_prefill_rows=1024, keep_depth_logits=False, device="cuda"):
        self.cfg, self.device = cfg, torch.device(device)
        self.max_batch, self.page_size, self.max_pages, self.max_seq_len = max_batch, page_size, max_pages, max_seq_len
        self.L = N.lib()
        self.ctx = N.ctx()
        dev = self.device
        W = {k: (v if v.is_cuda else v.to(dev)) for k, v in weights.items()}
        self._keep = []
        t, d = cfg.talker, cfg.depth
        G, G1, H = cfg.n_groups, cfg.n_groups + 1, t.hidden
        self.max_rows = max(max_prefill_rows, max_batch)
        self.t_rope = rope_table(cfg.max_pos, t, dev)
        self.d_rope = rope_table(64, d, dev)
        self.tl = pack_stack_weights(W, "talker.model", t, self._keep)
        self.dl = pack_stack_weights(W, "talker.code_predictor.model", d, self._keep)
        qc = N.Qwen3Config(_stack_config(t, page_size, self.max_rows, max_seq_len),
                           _stack_config(d, G, 2 * max_batch, G), cfg.vocab, cfg.text_vocab, cfg.text_hidden,
                           cfg.depth_vocab, G, cfg.eos_id, cfg.tts_pad_id, max_batch)
        demb = [W[f"talker.code_predictor.model.codec_embedding.{j}.weight"].contiguous() for j in range(G - 1)]
        self._demb_arr = (ctypes.c_void_p * (G - 1))(*[e.data_ptr() for e in demb])
        lm_head = torch.stack([W[f"talker.code_predictor.lm_head.{j}.weight"] for j in range(G - 1)], 0).contiguous()

        def P(name):
            x = W[name].contiguous()
            self._keep.append(x)
            return x.data_ptr()
        self._keep += demb + [lm_head]
        qw = N.Qwen3Weights(
            ctypes.cast(self.tl, ctypes.POINTER(N.LayerWeights)), ctypes.cast(self.dl, ctypes.POINTER(N.LayerWeights)),
            P("talker.model.norm.weight"), P("talker.code_predictor.model.norm.weight"),
            P("talker.model.codec_embedding.weight"), P("talker.model.text_embedding.weight"),
            P("talker.text_projection.linear_fc1.weight"), P("talker.text_projection.linear_fc1.bias"),
            P("talker.text_projection.linear_fc2.weight"), P("talker.text_projection.linear_fc2.bias"),
            P("talker.codec_head.weight"), ctypes.cast(self._demb_arr, ctypes.POINTER(ctypes.c_void_p)),
            lm_head.data_ptr(), P("talker.code_predictor.small_to_mtp_projection.weight"),
            P("talker.code_predictor.small_to_mtp_projection.bias"), self.t_rope.data_ptr(), self.d_rope.data_ptr(),
            cfg.max_pos, 64)
        h = ctypes.c_void_p()
        N.check(self.L.vox_qwen3_create(self.ctx, ctypes.byref(qc), ctypes.byref(qw), ctypes.byref(h)))
        self.h = h

        # graph-stable device buffers (cuda_graph_worker.py:383-390)
        i32 = dict(dtype=torch.int32, device=dev)
        R = self.max_rows
        self.input_ids = torch.zeros(max_batch, G1, **i32)
        self.input_masks = torch.ones(max_batch, dtype=torch.uint8, device=dev)
        self.input_features = torch.zeros(max_batch, H, dtype=torch.bfloat16, device=dev)
        # one int32 plan block, uploaded with a single async copy per frame:
        # [pos R | kvlen R | page R | slot R | q_req R | last_rows B | indptr B+1 | indices max_pages]
        self._plan_layout = {}
        off = 0
        self.pt_stride = (max_seq_len + page_size - 1) // page_size + 1
        for name, n in (("pos", R), ("kvlen", R), ("page", R), ("slot", R), ("q_req", R), ("last_rows", max_batch),
                        ("indptr", max_batch + 1), ("indices", max_pages), ("ptab", max_batch * self.pt_stride)):
            self._plan_layout[name] = (off, n)
            off += n
        self.plan_dev = torch.zeros(off, **i32)
        self._init_plan_host(off)
        self.kv = torch.zeros(t.layers, max_pages, 2, page_size, t.kv_heads, t.head_dim, dtype=torch.bfloat16, device=dev)
        # row 0 of this block is the frame's STATUS row (word 0: error code of the persistent kernels, written by the frame's last
        # kernel), rows 1.. are out_ids: the host's one D2H copy per frame (`read_ids` / `snapshot_src`) brings both, so a hand-off
        # timeout is seen in the same frame (include/voxhip.h: vox_qwen3_set_status)
        self._out_block = torch.zeros(max_batch + 1, G1, **i32)
        self.out_ids = self._out_block[1:]
        self.status_row = self._out_block[0]
        N.check(self.L.vox_qwen3_set_status(self.h, self.status_row.data_ptr()))
        self._launch_log = []             # the last two launches (frame / prefill + arguments + plan block): what `recover` replays
        self.launch_seq = 0
        self.persist_failures = []        # (launch_seq, error code) of every hand-off timeout seen (and recovered from)
        self.out_logits = torch.zeros(max_batch, cfg.vocab, dtype=torch.bfloat16, device=dev)
        self.out_hidden = torch.zeros(max_batch, H, dtype=torch.bfloat16, device=dev)
        self.out_depth_logits = (torch.zeros(G - 1, max_batch, cfg.depth_vocab, dtype=torch.bfloat16, device=dev)
                                 if keep_depth_logits else None)
        self.next_features = torch.zeros(max_batch, H, dtype=torch.bfloat16, device=dev)
        self.rng_offset = torch.zeros(1, dtype=torch.int64, device=dev)
        self.row_ids = torch.zeros(R, G1, **i32)
        self.row_masks = torch.zeros(R, dtype=torch.uint8, device=dev)
        self.row_feats = torch.zeros(R, H, dtype=torch.bfloat16, device=dev)
        self._graphs = {}
        self.keep_hidden = True
        # The engine owns no stream: everything is enqueued on the CALLER's current stream, so results are ordered with
        # whatever the caller does next and no stream ever waits for another while a frame graph runs (N.graph_capture
        # explains why that matters: ~1.3 us per dependent dispatch).  Capture happens once per shape on a private stream.

    @property
    def stream(self):
        """The stream the engine's work is on = the caller's current stream (kept for callers that time or synchronise it)."""
        return torch.cuda.current_stream()

    class _OnStream:
        """Kept for callers that bracket engine-ordered work (`with eng._OnStream(eng): event.record()`): a no-op now that
        the engine runs on the current stream."""
        def __init__(self, eng):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    # ---- plan upload -------------------------------------------------------------------------------
    def _pd(self, name):
        off, n = self._plan_layout[name]
        return self.plan_dev[off:off + n]

    PLAN_RING = 3

    def _init_plan_host(self, n):
        """The host side of the plan: one ordinary array that is edited in place (`_plan_np`) and a small ring of pinned
        blocks it is copied through.  A pinned block is reused only after the H2D copy that read it has run (its event), so
        an upload never waits for the frame in flight: with async scheduling the host stages frame N + 1 while N runs."""
        self._plan_np = np.zeros(n, dtype=np.int32)
        self._plan_pin = [torch.zeros(n, dtype=torch.int32).pin_memory() for _ in range(self.PLAN_RING)]
        self._plan_pin_np = [t.numpy() for t in self._plan_pin]
        self._plan_ev = [None] * self.PLAN_RING
        self._plan_k = 0
        self._ptab_last = {}

    def upload_plan(self, **arrays):
        """Host int lists/arrays -> a pinned block -> one async H2D copy in stream order (no synchronisation with the frame
        in flight).  Host cost matters: plain numpy writes, and the per-row page table is rewritten only for rows whose page
        list changed since the last upload."""
        hv = self._plan_np
        for name, a in arrays.items():
            a = np.asarray(a, dtype=np.int32)
            off, _ = self._plan_layout[name]
            hv[off: off + a.size] = a.ravel()
        if "indptr" in arrays and "indices" in arrays and len(arrays["indptr"]) - 1 <= self.max_batch:
            ip, ix = np.asarray(arrays["indptr"], dtype=np.int64), np.asarray(arrays["indices"], dtype=np.int32)
            off, _ = self._plan_layout["ptab"]
            pt = hv[off: off + self.max_batch * self.pt_stride].reshape(self.max_batch, self.pt_stride)
            last = self._ptab_last
            for b in range(len(ip) - 1):
                row = ix[ip[b]: ip[b + 1]][: self.pt_stride]
                prev = last.get(b)
                if prev is None or prev.size != row.size or not np.array_equal(prev, row):
                    pt[b, : row.size] = row
                    last[b] = row.copy()
        k = self._plan_k
        self._plan_k = (k + 1) % self.PLAN_RING
        self._plan_gen = getattr(self, "_plan_gen", 0) + 1
        self._plan_slot_gen = getattr(self, "_plan_slot_gen", [0] * self.PLAN_RING)
        self._plan_slot_gen[k] = self._plan_gen
        self._plan_last = (k, self._plan_gen)
        if self._plan_ev[k] is not None:
            self._plan_ev[k].synchronize()          # PLAN_RING uploads ago: long done
        self._plan_pin_np[k][:] = hv
        with self._OnStream(self):
            self.plan_dev.copy_(self._plan_pin[k], non_blocking=True)
            if self._plan_ev[k] is None:
                self._plan_ev[k] = torch.cuda.Event()
            self._plan_ev[k].record()

    def _io(self):
        return N.Qwen3IO(self.input_ids.data_ptr(), self.input_masks.data_ptr(), self.input_features.data_ptr(),
                         self._pd("pos").data_ptr(), self._pd("kvlen").data_ptr(), self._pd("page").data_ptr(),
                         self._pd("slot").data_ptr(), self._pd("indptr").data_ptr(), self._pd("indices").data_ptr(),
                         self._pd("ptab").data_ptr(), self.pt_stride, self.kv.data_ptr(), self.kv[0].numel(), self.out_ids.data_ptr(), self.out_logits.data_ptr(),
                         self.out_hidden.data_ptr() if self.keep_hidden else None,
                         self.out_depth_logits.data_ptr() if self.out_depth_logits is not None else None,
                         self.next_features.data_ptr(), self.rng_offset.data_ptr())

    @staticmethod
    def sampling_cfg(greedy=True, top_k=0, top_p=1.0, min_p=0.0, temperature=1.0):
        return N.SamplingCfg(int(greedy), int(top_k or 0), float(1.0 if top_p is None else top_p),
                             float(min_p or 0.0), float(temperature), 1.0)

    # ---- one frame ---------------------------------------------------------------------------------
    launch_seq = 0            # launches enqueued so far (frames + prefills); engines without a status row only count
    max_graphs = 512          # frame graphs kept (batch x kv bucket x sampling); the oldest goes first.  Prefill graphs: own LRU (_prefill_graph)

    def frame(self, batch, max_kvlen, sampling=None, seed=0, feedback=True, use_graph=True):
        """Enqueue one frame on the current stream.  plan arrays / inputs must already be on the device."""
        sampling = sampling or self.sampling_cfg()
        bucket = max(256, 1 << (int(max_kvlen) - 1).bit_length())       # kv-length bucket bounds the attention grid
        bucket = min(bucket, self.max_seq_len)
        if max_kvlen > bucket:
            raise N.VoxError(f"kv length {max_kvlen} exceeds max_seq_len {self.max_seq_len}")
        self._log_launch("frame", (batch, max_kvlen, sampling, seed, feedback, use_graph), batch)
        with self._OnStream(self):
            self._frame_on_stream(batch, bucket, sampling, seed, feedback, use_graph)

    # ---- fail loud within one frame, recover bit-identically -------------------------------------------
    def _log_launch(self, kind, args, rows):
        if getattr(self, "_replaying", False):
            return
        self.launch_seq += 1
        if not hasattr(self, "status_row"):
            return
        self._launch_log.append({"seq": self.launch_seq, "kind": kind, "args": args, "rows": rows, "plan": getattr(self, "_plan_last", None),
                                 "restaged": self.__dict__.pop("_restaged", False)})
        del self._launch_log[:-2]

    def note_restage(self):
        """The caller rewrote input_ids / input_masks / input_features by hand (batch composition changed): the NEXT launch does
        not start from the previous launch's feedback, so it cannot be replayed behind it."""
        self._restaged = True

    def snapshot_src(self, batch):
        """What the host copies per frame: the status row + out_ids[:batch] — one contiguous D2H.  Engines without a status row
        (LMEngine / CSMEngine inherit this method, not the buffer) say so instead of failing on a missing attribute."""
        blk = self.__dict__.get("_out_block")
        if blk is None:
            raise N.VoxError(f"{type(self).__name__} has no status row: copy out_ids[:batch] (worker/base.py: engine_has_status_row)")
        return blk[: batch + 1]

    def read_ids(self, batch):
        """out_ids[:batch] on the host (int64) through ONE blocking D2H that also brings the frame's status word; a hand-off timeout
        of the persistent kernels is recovered from here (the frame re-run on the launch chain, bit-identical) before the ids are returned."""
        if not hasattr(self, "status_row"):
            return self.out_ids[:batch].cpu().to(torch.long)
        blk = self.snapshot_src(batch).cpu()
        if int(blk[0, 0]) != 0:
            self.recover(1, code=int(blk[0, 0]))
            blk = self.snapshot_src(batch).cpu()
            if int(blk[0, 0]) != 0:
                raise N.VoxError(f"persistent kernels: status {int(blk[0, 0]):#x} after recovery")
        return blk[1:].to(torch.long)

    def _drop_graphs(self):
        torch.cuda.synchronize()
        for g in list(self._graphs.values()) + [g for g in self.__dict__.get("_pf_graphs", {}).values() if g]:
            self.L.vox_graph_destroy(g)
        self._graphs.clear()
        self.__dict__.get("_pf_graphs", {}).clear()

    def recover(self, back=1, code=0, on_first_done=None):
        """A hand-off of a persistent kernel timed out in the launch `back` launches ago (1 = the last one; 2 = one more launch was
        already enqueued behind it — async scheduling): everything since is garbage.  Order of work:
          1. wait for the device and CHECK that the replay is possible — a record of the launch(es), their plan blocks still in the
             pinned ring, the failed launch's inputs in the shadow (vox_qwen3_frame_restore leaves everything alone when they are not)
             — before any device state is changed: a launch that cannot be replayed raises with the engine exactly as the failure left it
             (error words still set, so every later frame keeps reporting it);
          2. turn the persistent kernels off for this engine (the launch chains are bit-identical) and drop the graphs that hold them;
          3. run the failed launch again, call `on_first_done()` (the caller re-reads its outputs), then (back = 2) the launch behind it.
             That one normally starts from the first one's feedback; if the caller had RESTAGED its inputs by hand (`note_restage`: the
             batch composition changed), its own staged inputs are put back from ITS shadow slot first (every decode frame saves its
             inputs on entry, so what the caller had written is still there) — a restaged prefill needs nothing: its row buffers are
             only ever written by the caller."""
        import logging
        torch.cuda.synchronize()
        log = self._launch_log[-back:]
        if len(log) < back or back < 1:
            raise N.VoxError(f"persistent kernels: hand-off timeout (code {code:#x}) and no record of the launch to replay")
        for ent in log:                                   # (1) host-side checks: nothing touched yet
            if ent["plan"] is not None:
                k, gen = ent["plan"]
                if self._plan_slot_gen[k] != gen:
                    raise N.VoxError(f"persistent kernels: hand-off timeout (code {code:#x}); the plan block of launch {ent['seq']} was "
                                     "overwritten and it cannot be replayed")
        io = self._io()
        first = log[0]
        # (1, device side) the failed launch's inputs + frame counter back from the shadow: a no-op that reports 0x7fffffff when absent
        N.check(self.L.vox_qwen3_frame_restore(self.h, N.stream(), ctypes.byref(io), back, first["rows"] if first["kind"] == "frame" else 0))
        torch.cuda.synchronize()
        if int(self.status_row[0].item()) != 0:
            raise N.VoxError(f"persistent kernels: hand-off timeout (code {code:#x}) and the failed launch's inputs are not in the shadow")
        logging.getLogger(__name__).error("persistent kernels: a hand-off timed out (code %#x) in launch %d; switching this engine to the "
                                          "launch chains and replaying %d launch(es)", code, first["seq"], back)
        self.persist_failures.append((first["seq"], code))
        N.check(self.L.vox_qwen3_persist_reset(self.h, 1))          # (2)
        self._drop_graphs()
        self._replaying = True
        try:
            for i, ent in enumerate(log):                 # (3)
                if ent["plan"] is not None:
                    self.plan_dev.copy_(self._plan_pin[ent["plan"][0]], non_blocking=True)
                if i > 0 and ent["restaged"] and ent["kind"] == "frame":
                    # back = 0: the frame that started from the counter value the replay has just reached
                    N.check(self.L.vox_qwen3_frame_restore(self.h, N.stream(), ctypes.byref(io), 0, ent["rows"]))
                    torch.cuda.synchronize()
                    if int(self.status_row[0].item()) != 0:
                        raise N.VoxError("persistent kernels: the restaged launch behind the failed one is not in the shadow")
                getattr(self, ent["kind"])(*ent["args"])
                torch.cuda.synchronize()
                if i == 0 and on_first_done is not None:
                    on_first_done()
        finally:
            self._replaying = False

    def _frame_on_stream(self, batch, bucket, sampling, seed, feedback, use_graph):
        if not use_graph:
            io = self._io()
            self._native_frame(io, N.stream(), batch, bucket, sampling, seed, feedback)
            return
        key = (batch, bucket, bytes(sampling), seed, bool(feedback), self.keep_hidden)
        g = self._graphs.get(key)
        # (persistent kernels: their bounded spins turn a stuck hand-off into wrong data + an error word, never a hang; the word
        # travels with every frame's token snapshot — read_ids / snapshot_src — and `recover` replays the frame on the launch chain)
        if g is None:
            io = self._io()
            st = N.stream()
            # warm-up outside capture (sets kernel attributes), on a scratch copy of the mutable state
            state = self._mutable_state()
            saved = [x.clone() for x in state]
            self._native_frame(io, st, batch, bucket, sampling, seed, feedback)
            torch.cuda.current_stream().synchronize()
            for dst, src in zip(state, saved):
                dst.copy_(src)
            torch.cuda.current_stream().synchronize()
            with N.graph_capture() as cap:
                self._native_frame(io, N.stream(), batch, bucket, sampling, seed, feedback)
            gh = cap.graph
            while len(self._graphs) >= self.max_graphs:      # bounded: the oldest captured shape goes (insertion order)
                old_key = next(iter(self._graphs))
                torch.cuda.current_stream().synchronize()     # (it may still be running)
                self.L.vox_graph_destroy(self._graphs.pop(old_key))
            g = self._graphs[key] = gh
        N.check(self.L.vox_graph_launch(g, N.stream()))

    def prefill(self, n_rows, n_req, max_kvlen, sampling=None, seed=0, feedback=True, use_graph=True):
        """Ragged prefill of rows staged in row_ids/row_masks/row_feats + plan arrays.  A shape (rows, requests, kv bound) seen
        for the second time is captured into a hipGraph and replayed from then on (the reference captures padded prefill
        buckets up front, cuda_graph_worker.py:206-352; here every buffer of the call sits at a fixed address, so the exact
        shape can be captured without padding and without changing which kernels — hence which bits — a prompt gets)."""
        sampling = sampling or self.sampling_cfg()
        self._log_launch("prefill", (n_rows, n_req, max_kvlen, sampling, seed, feedback, use_graph), n_req)
        with self._OnStream(self):
            if not use_graph:
                return self._prefill_on_stream(n_rows, n_req, max_kvlen, sampling, seed, feedback)
            key = ("prefill", n_rows, n_req, min(max(32, int(max_kvlen)), self.max_seq_len), bytes(sampling), seed, bool(feedback),
                   self.keep_hidden)
            ent = self._prefill_graph(key)
            if ent is None:                                   # not (yet) worth a graph: eager (also sets kernel attributes)
                return self._prefill_on_stream(n_rows, n_req, max_kvlen, sampling, seed, feedback)
            if ent is False:                                  # capture (capture does not execute), then fall through to replay
                with N.graph_capture() as cap:
                    self._prefill_on_stream(n_rows, n_req, max_kvlen, sampling, seed, feedback)
                gh = cap.graph
                ent = self._pf_graphs[key] = gh
            N.check(self.L.vox_graph_launch(ent, N.stream()))

    # Prefill graphs live in their own bounded LRU, apart from the frame graphs: a shape is captured on its
    # `prefill_capture_after`-th sighting (capture + instantiate cost more than one eager prefill, so a shape seen twice does
    # not pay), at most `max_prefill_graphs` are kept and the least recently used one is destroyed when a new one arrives;
    # sighting counters are bounded too (oldest forgotten).  The frame graphs (batch x kv bucket) are never evicted.
    max_prefill_graphs = 64
    prefill_capture_after = 3

    def _prefill_graph(self, key):
        """-> graph handle (replay), False (capture now) or None (run eagerly)."""
        pg = self.__dict__.setdefault("_pf_graphs", OrderedDict())
        seen = self.__dict__.setdefault("_pf_seen", OrderedDict())
        g = pg.get(key)
        if g is not None:
            pg.move_to_end(key)
            return g
        n = seen.pop(key, 0) + 1
        if n < self.prefill_capture_after:
            seen[key] = n
            while len(seen) > 1024:
                seen.popitem(last=False)
            return None
        while len(pg) >= self.max_prefill_graphs:
            _, old = pg.popitem(last=False)
            self.L.vox_graph_destroy(old)
        return False

    def _mutable_state(self):
        return [self.input_ids, self.input_masks, self.input_features, self.rng_offset]

    def _native_frame(self, io, st, batch, bucket, sampling, seed, feedback):
        N.check(self.L.vox_qwen3_frame(self.h, st, ctypes.byref(io), batch, bucket, ctypes.byref(sampling), seed,
                                       int(feedback)))

    def _native_prefill(self, *args):
        N.check(self.L.vox_qwen3_prefill(*args))

    def _native_destroy(self):
        self.L.vox_qwen3_destroy(self.h)

    def _prefill_on_stream(self, n_rows, n_req, max_kvlen, sampling, seed, feedback):
        io = self._io()
        self._native_prefill(self.h, N.stream(), ctypes.byref(io), self.row_ids.data_ptr(),
                                         self.row_masks.data_ptr(), self.row_feats.data_ptr(),
                                         self._pd("q_req").data_ptr(), n_rows, self._pd("last_rows").data_ptr(), n_req,
                                         min(max(32, max_kvlen), self.max_seq_len), ctypes.byref(sampling), seed,
                                         int(feedback))

    def depth_persist_status(self):
        """(enabled, error_code) of the persistent depth steps (include/voxhip.h: vox_qwen3_depth_persist_status); synchronises."""
        en, err = ctypes.c_int32(0), ctypes.c_uint32(0)
        N.check(self.L.vox_qwen3_depth_persist_status(self.h, ctypes.byref(en), ctypes.byref(err)))
        return int(en.value), int(err.value)      # enabled: bit 0 = depth steps, bit 1 = talker MLP halves

    def close(self):
        for g in list(self._graphs.values()) + list(self.__dict__.get("_pf_graphs", {}).values()):
            if g:
                self.L.vox_graph_destroy(g)
        self._graphs.clear()
        self.__dict__.get("_pf_graphs", {}).clear()
        if self.h:
            self._native_destroy()
            self.h = None


# ---------------------------------------------------------------------------------------------------
@dataclass
class LMCfg:
    """Single-stack speech LM (GLM-4-Voice, CosyVoice2 LLM, ...)."""
    stack: StackCfg
    vocab_in: int
    vocab_out: int
    n_codebooks: int = 1
    input_mode: int = 0          # 1: x = mask ? input_features : embedding[id]   (cosyvoice2.py:1020-1024)
    max_pos: int = 4096


class LMIO(ctypes.Structure):
    _fields_ = [("input_ids", ctypes.c_void_p), ("input_masks", ctypes.c_void_p), ("input_features", ctypes.c_void_p),
                ("pos", ctypes.c_void_p), ("kvlen", ctypes.c_void_p), ("page", ctypes.c_void_p), ("slot", ctypes.c_void_p),
                ("kv_indptr", ctypes.c_void_p), ("kv_indices", ctypes.c_void_p), ("page_table", ctypes.c_void_p),
                ("pt_stride", ctypes.c_int64), ("kv", ctypes.c_void_p), ("kv_layer_stride", ctypes.c_int64),
                ("out_ids", ctypes.c_void_p), ("out_logits", ctypes.c_void_p), ("rep_cache", ctypes.c_void_p),
                ("rep_w", ctypes.c_int32), ("rep_window", ctypes.c_int32), ("rng_offset", ctypes.c_void_p)]


class LMConfigC(ctypes.Structure):
    _fields_ = [("stack", N.StackConfig), ("vocab_in", ctypes.c_int32), ("vocab_out", ctypes.c_int32),
                ("ids_stride", ctypes.c_int32), ("input_mode", ctypes.c_int32), ("max_batch", ctypes.c_int32)]


class LMWeightsC(ctypes.Structure):
    _fields_ = [("layers", ctypes.POINTER(N.LayerWeights)), ("final_norm", ctypes.c_void_p), ("embedding", ctypes.c_void_p),
                ("head_w", ctypes.c_void_p), ("head_b", ctypes.c_void_p), ("rope", ctypes.c_void_p),
                ("rope_max_pos", ctypes.c_int32)]


class LMEngine(Qwen3Engine):
    """Host side of `vox_lm_*`: same plan / stream / hipGraph machinery as Qwen3Engine, one decoder stack.
    `layers`: list of dicts with keys wqkv,bqkv,wo,wgate,wup,wdown,ln1,ln2 (+qnorm,knorm) -> bf16 tensors."""

    def __init__(self, cfg: LMCfg, layers, final_norm, embedding, head_w, head_b=None, max_batch=8, page_size=128,
                 max_pages=256, max_seq_len=2304, max_prefill_rows=1024, rep_window=None, device="cuda"):
        self.cfg, self.device = cfg, torch.device(device)
        self.max_batch, self.page_size, self.max_pages, self.max_seq_len = max_batch, page_size, max_pages, max_seq_len
        self.L, self.ctx = N.lib(), N.ctx()
        L = self.L
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.vox_lm_create.restype, L.vox_lm_create.argtypes = ci, [vp, ctypes.POINTER(LMConfigC), ctypes.POINTER(LMWeightsC), ctypes.POINTER(vp)]
        L.vox_lm_destroy.restype, L.vox_lm_destroy.argtypes = None, [vp]
        L.vox_lm_frame.restype, L.vox_lm_frame.argtypes = ci, [vp, vp, ctypes.POINTER(LMIO), ci, ci, ctypes.POINTER(N.SamplingCfg), ctypes.c_uint64, ci]
        L.vox_lm_prefill.restype, L.vox_lm_prefill.argtypes = ci, [vp, vp, ctypes.POINTER(LMIO), vp, vp, vp, vp, ci, vp, ci, ci, ctypes.POINTER(N.SamplingCfg), ctypes.c_uint64, ci]
        dev, c = self.device, cfg.stack
        self._keep = []
        arr = (N.LayerWeights * c.layers)()
        for i, lw in enumerate(layers):
            for k, t in lw.items():
                if t is not None:
                    t = t.to(dev).contiguous()
                    self._keep.append(t)
                    setattr(arr[i], k, t.data_ptr())
        self.rope = rope_table(cfg.max_pos, c, dev)
        self.max_rows = max(max_prefill_rows, max_batch)
        keep = lambda t: (self._keep.append(t.to(dev).contiguous()) or self._keep[-1].data_ptr()) if t is not None else None
        cc = LMConfigC(_stack_config(c, page_size, self.max_rows, max_seq_len), cfg.vocab_in, cfg.vocab_out, cfg.n_codebooks,
                       cfg.input_mode, max_batch)
        cw = LMWeightsC(ctypes.cast(arr, ctypes.POINTER(N.LayerWeights)), keep(final_norm), keep(embedding), keep(head_w),
                        keep(head_b), self.rope.data_ptr(), cfg.max_pos)
        self._arr = arr
        h = ctypes.c_void_p()
        N.check(L.vox_lm_create(self.ctx, ctypes.byref(cc), ctypes.byref(cw), ctypes.byref(h)))
        self.h = h
        i32 = dict(dtype=torch.int32, device=dev)
        R, H, C = self.max_rows, c.hidden, cfg.n_codebooks
        self.input_ids = torch.zeros(max_batch, C, **i32)
        self.input_masks = torch.zeros(max_batch, dtype=torch.uint8, device=dev)
        self.input_features = torch.zeros(max_batch, H, dtype=torch.bfloat16, device=dev)
        self.pt_stride = (max_seq_len + page_size - 1) // page_size + 1
        self._plan_layout, off = {}, 0
        for name, n in (("pos", R), ("kvlen", R), ("page", R), ("slot", R), ("q_req", R), ("last_rows", max_batch),
                        ("indptr", max_batch + 1), ("indices", max_pages), ("ptab", max_batch * self.pt_stride)):
            self._plan_layout[name] = (off, n)
            off += n
        self.plan_dev = torch.zeros(off, **i32)
        self._init_plan_host(off)
        self.kv = torch.zeros(c.layers, max_pages, 2, page_size, c.kv_heads, c.head_dim, dtype=torch.bfloat16, device=dev)
        self.out_ids = torch.zeros(max_batch, **i32)
        self.out_logits = torch.zeros(max_batch, cfg.vocab_out, dtype=torch.bfloat16, device=dev)
        self.rep_window = rep_window
        self.rep_cache = None
        if rep_window is not None:
            self.rep_w = rep_window if rep_window > 0 else 1
            self.rep_cache = torch.zeros(max_batch, self.rep_w, 1, cfg.vocab_out, dtype=torch.uint8, device=dev)
        self.rng_offset = torch.zeros(1, dtype=torch.int64, device=dev)
        self.row_ids = torch.zeros(R, C, **i32)
        self.row_masks = torch.zeros(R, dtype=torch.uint8, device=dev)
        self.row_feats = torch.zeros(R, H, dtype=torch.bfloat16, device=dev)
        self._graphs, self.keep_hidden = {}, False

    def _io(self):
        return LMIO(self.input_ids.data_ptr(), self.input_masks.data_ptr(), self.input_features.data_ptr(),
                    self._pd("pos").data_ptr(), self._pd("kvlen").data_ptr(), self._pd("page").data_ptr(),
                    self._pd("slot").data_ptr(), self._pd("indptr").data_ptr(), self._pd("indices").data_ptr(),
                    self._pd("ptab").data_ptr(), self.pt_stride, self.kv.data_ptr(), self.kv[0].numel(),
                    self.out_ids.data_ptr(), self.out_logits.data_ptr(),
                    self.rep_cache.data_ptr() if self.rep_cache is not None else None,
                    self.rep_w if self.rep_cache is not None else 0, self.rep_window or 0, self.rng_offset.data_ptr())

    def _mutable_state(self):
        st = [self.input_ids, self.input_masks, self.rng_offset]
        return st + ([self.rep_cache] if self.rep_cache is not None else [])

    def _native_frame(self, io, st, batch, bucket, sampling, seed, feedback):
        N.check(self.L.vox_lm_frame(self.h, st, ctypes.byref(io), batch, bucket, ctypes.byref(sampling), seed, int(feedback)))

    def _native_prefill(self, *args):
        N.check(self.L.vox_lm_prefill(*args))

    def _native_destroy(self):
        self.L.vox_lm_destroy(self.h)

    @staticmethod
    def sampling_cfg(greedy=True, top_k=0, top_p=1.0, min_p=0.0, temperature=1.0, repetition_penalty=1.0):
        return N.SamplingCfg(int(greedy), int(top_k or 0), float(1.0 if top_p is None else top_p), float(min_p or 0.0),
                             float(temperature), float(repetition_penalty or 1.0))


# ---------------------------------------------------------------------------------------------------
@dataclass
class CSMCfg:
    """CSM-1B (model/csm.py; transformers CsmConfig defaults): llama-3.2-1B-style backbone + 4-layer depth decoder."""
    backbone: StackCfg = field(default_factory=lambda: StackCfg(2048, 16, 32, 8, 64, 8192, eps=1e-5, rope_theta=5e5,
                                                                rope_scale=32.0, rope_llama31=(1.0, 4.0, 8192), qk_norm=False))
    depth: StackCfg = field(default_factory=lambda: StackCfg(1024, 4, 8, 2, 128, 8192, eps=1e-5, rope_theta=5e5,
                                                             rope_scale=32.0, rope_llama31=(1.0, 4.0, 8192), qk_norm=False))
    vocab: int = 2051
    text_vocab: int = 128256
    n_codebooks: int = 32
    max_pos: int = 2048


class CsmConfigC(ctypes.Structure):
    _fields_ = [("backbone", N.StackConfig), ("depth", N.StackConfig), ("vocab", ctypes.c_int32), ("text_vocab", ctypes.c_int32),
                ("n_codebooks", ctypes.c_int32), ("max_batch", ctypes.c_int32)]


class CsmWeightsC(ctypes.Structure):
    _fields_ = [("backbone_layers", ctypes.POINTER(N.LayerWeights)), ("depth_layers", ctypes.POINTER(N.LayerWeights))] + \
               [(n, ctypes.c_void_p) for n in ("backbone_norm", "depth_norm", "audio_embedding", "text_embedding", "lm_head",
                                               "depth_proj", "depth_heads", "backbone_rope", "depth_rope")] + \
               [("backbone_rope_max_pos", ctypes.c_int32), ("depth_rope_max_pos", ctypes.c_int32)]


class CsmIO(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("input_ids", "input_masks", "pos", "kvlen", "page", "slot", "kv_indptr",
                                               "kv_indices", "page_table")] + \
               [("pt_stride", ctypes.c_int64), ("kv", ctypes.c_void_p), ("kv_layer_stride", ctypes.c_int64)] + \
               [(n, ctypes.c_void_p) for n in ("out_ids", "out_logits", "out_hidden", "out_depth_logits", "rng_offset")]


class CSMEngine(Qwen3Engine):
    """Host side of `vox_csm_*`.  `weights`: the reference's state_dict names (CsmForConditionalGeneration):
    backbone_model.layers.*, backbone_model.norm, backbone_model.embed_tokens.embed_audio_tokens, embed_text_tokens,
    lm_head, depth_decoder.model.{layers,norm,inputs_embeds_projector}, depth_decoder.codebooks_head.weight [31,Hd,V]."""

    def __init__(self, cfg: CSMCfg, weights: Dict[str, torch.Tensor], max_batch=8, page_size=128, max_pages=256,
                 max_seq_len=2304, max_prefill_rows=1024, keep_depth_logits=False, device="cuda"):
        self.cfg, self.device = cfg, torch.device(device)
        self.max_batch, self.page_size, self.max_pages, self.max_seq_len = max_batch, page_size, max_pages, max_seq_len
        self.L, self.ctx = N.lib(), N.ctx()
        L, vp, ci = self.L, ctypes.c_void_p, ctypes.c_int
        L.vox_csm_create.restype, L.vox_csm_create.argtypes = ci, [vp, ctypes.POINTER(CsmConfigC), ctypes.POINTER(CsmWeightsC), ctypes.POINTER(vp)]
        L.vox_csm_destroy.restype, L.vox_csm_destroy.argtypes = None, [vp]
        L.vox_csm_frame.restype, L.vox_csm_frame.argtypes = ci, [vp, vp, ctypes.POINTER(CsmIO), ci, ci, ctypes.POINTER(N.SamplingCfg), ctypes.c_uint64, ci]
        L.vox_csm_prefill.restype, L.vox_csm_prefill.argtypes = ci, [vp, vp, ctypes.POINTER(CsmIO), vp, vp, vp, ci, vp, ci, ci, ctypes.POINTER(N.SamplingCfg), ctypes.c_uint64, ci]
        dev = self.device
        W = {k: (v if v.is_cuda else v.to(dev)) for k, v in weights.items()}
        self._keep = []
        b, d = cfg.backbone, cfg.depth
        C, C1, H, V = cfg.n_codebooks, cfg.n_codebooks + 1, b.hidden, cfg.vocab
        self.max_rows = max(max_prefill_rows, max_batch)
        self.b_rope, self.d_rope = rope_table(cfg.max_pos, b, dev), rope_table(64, d, dev)
        self.bl = pack_stack_weights(W, "backbone_model", b, self._keep)
        self.dl = pack_stack_weights(W, "depth_decoder.model", d, self._keep)
        # codebooks_head.weight is [C-1, Hd, V] (x @ W): the GEMV streams rows of [V, Hd] -> transpose once (layout only)
        heads = W["depth_decoder.codebooks_head.weight"].transpose(1, 2).contiguous()

        def P(name):
            x = W[name].contiguous()
            self._keep.append(x)
            return x.data_ptr()
        self._keep.append(heads)
        cc = CsmConfigC(_stack_config(b, page_size, self.max_rows, max_seq_len), _stack_config(d, C, 2 * max_batch, C), V,
                        cfg.text_vocab, C, max_batch)
        cw = CsmWeightsC(ctypes.cast(self.bl, ctypes.POINTER(N.LayerWeights)), ctypes.cast(self.dl, ctypes.POINTER(N.LayerWeights)),
                         P("backbone_model.norm.weight"), P("depth_decoder.model.norm.weight"),
                         P("backbone_model.embed_tokens.embed_audio_tokens.weight"), P("embed_text_tokens.weight"),
                         P("lm_head.weight"), P("depth_decoder.model.inputs_embeds_projector.weight"), heads.data_ptr(),
                         self.b_rope.data_ptr(), self.d_rope.data_ptr(), cfg.max_pos, 64)
        h = ctypes.c_void_p()
        N.check(L.vox_csm_create(self.ctx, ctypes.byref(cc), ctypes.byref(cw), ctypes.byref(h)))
        self.h = h
        i32 = dict(dtype=torch.int32, device=dev)
        R = self.max_rows
        self.input_ids = torch.zeros(max_batch, C1, **i32)
        self.input_masks = torch.zeros(max_batch, C1, dtype=torch.uint8, device=dev)
        self.pt_stride = (max_seq_len + page_size - 1) // page_size + 1
        self._plan_layout, off = {}, 0
        for name, n in (("pos", R), ("kvlen", R), ("page", R), ("slot", R), ("q_req", R), ("last_rows", max_batch),
                        ("indptr", max_batch + 1), ("indices", max_pages), ("ptab", max_batch * self.pt_stride)):
            self._plan_layout[name] = (off, n)
            off += n
        self.plan_dev = torch.zeros(off, **i32)
        self._init_plan_host(off)
        self.kv = torch.zeros(b.layers, max_pages, 2, page_size, b.kv_heads, b.head_dim, dtype=torch.bfloat16, device=dev)
        self.out_ids = torch.zeros(max_batch, C1, **i32)
        self.out_logits = torch.zeros(max_batch, V, dtype=torch.bfloat16, device=dev)
        self.out_hidden = torch.zeros(max_batch, H, dtype=torch.bfloat16, device=dev)
        self.out_depth_logits = (torch.zeros(C - 1, max_batch, V, dtype=torch.bfloat16, device=dev) if keep_depth_logits else None)
        self.rng_offset = torch.zeros(1, dtype=torch.int64, device=dev)
        self.row_ids = torch.zeros(R, C1, **i32)
        self.row_masks = torch.zeros(R, C1, dtype=torch.uint8, device=dev)
        self._graphs, self.keep_hidden = {}, True

    def _io(self):
        return CsmIO(self.input_ids.data_ptr(), self.input_masks.data_ptr(), self._pd("pos").data_ptr(),
                     self._pd("kvlen").data_ptr(), self._pd("page").data_ptr(), self._pd("slot").data_ptr(),
                     self._pd("indptr").data_ptr(), self._pd("indices").data_ptr(), self._pd("ptab").data_ptr(),
                     self.pt_stride, self.kv.data_ptr(), self.kv[0].numel(), self.out_ids.data_ptr(), self.out_logits.data_ptr(),
                     self.out_hidden.data_ptr() if self.keep_hidden else None,
                     self.out_depth_logits.data_ptr() if self.out_depth_logits is not None else None, self.rng_offset.data_ptr())

    def _mutable_state(self):
        return [self.input_ids, self.input_masks, self.rng_offset]

    def _native_frame(self, io, st, batch, bucket, sampling, seed, feedback):
        N.check(self.L.vox_csm_frame(self.h, st, ctypes.byref(io), batch, bucket, ctypes.byref(sampling), seed, int(feedback)))

    def _native_destroy(self):
        self.L.vox_csm_destroy(self.h)

    def _prefill_on_stream(self, n_rows, n_req, max_kvlen, sampling, seed, feedback):
        io = self._io()
        N.check(self.L.vox_csm_prefill(self.h, N.stream(), ctypes.byref(io), self.row_ids.data_ptr(), self.row_masks.data_ptr(),
                                       self._pd("q_req").data_ptr(), n_rows, self._pd("last_rows").data_ptr(), n_req,
                                       min(max(32, max_kvlen), self.max_seq_len), ctypes.byref(sampling), seed, int(feedback)))
