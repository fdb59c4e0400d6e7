"""Device engine: PCM batches -> log-mel features -> conformer encoder states on one MI355X.

Fills the slot the reference gives to its remote model call
(ref:src/omnilingual_asr/models/inference/gemini_pipeline.py:474-539).  All device arithmetic is in
liboasr_hip.so; torch only owns the memory and the stream.  One engine belongs to one device and is
not re-entrant: callers serialise `encode_batch` (the pipeline holds a lock).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _hip, tokenizer
from .weights import CTC_LOGIT_PITCH, Geometry, init_random_weights, pack_for_device, resolve_geometry

SAMPLE_RATE = 16000


def pad_frames(t: int) -> int:
    """Stacked-frame pitch per utterance: multiple of 32 (attention / convolution tiling)."""
    return max(32, (int(t) + 31) // 32 * 32)


class HipEngine:
    def __init__(self, model="G300", *, weights: dict | None = None, seed: int = 0,
                 device: int | str | torch.device | None = None, use_graphs: bool = True, fp8: bool = False):
        _hip.lib()
        _hip.require_device()
        self.geometry: Geometry = resolve_geometry(model)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None \
            else torch.device(device)
        if self.device.type != "cuda":
            raise _hip.OasrError("HipEngine needs a HIP device; there is no CPU fallback")
        g = self.geometry
        if weights is None:
            weights = init_random_weights(g, seed)
        self.fp8 = bool(fp8)    # e4m3 weights + per-row quantised activations in the eight large projections
        self._packed = pack_for_device(weights, g, self.device, fp8=self.fp8)   # keeps device buffers alive
        self._pack_gemm_weights()
        self._tables = torch.from_numpy(_hip.logmel_table_blob().view(np.int32)).to(self.device)
        self._handle = self._create_handle()
        self._buffers: dict = {}
        self._graphs: dict = {}
        self._resample_tables: dict = {}
        self._pinned: dict = {}
        self._tickets: dict = {}
        self._copy_stream = None
        self.use_graphs = use_graphs
        self._pcm_stage = None
        self._pcm_stage_events = [None, None]
        self._copy_pool = None
        self.max_buffer_sets = 6        # distinct (batch, padded length) shapes kept resident

    # ---- set-up ---------------------------------------------------------------------------------------
    GEMM_WEIGHTS = ("ffn1_w1", "ffn1_w2", "wqkv", "wo", "pw1", "pw2", "ffn2_w1", "ffn2_w2")

    def _pack_gemm_weights(self):
        """Every bf16 projection matrix goes into the GEMM kernel's packed layout (1 KiB pieces of 16 rows x 32 k
        in the order the kernel copies them: `oasr_gemm_pack_w_bf16`); the row-major copies are dropped."""
        L = _hip.lib()
        with torch.cuda.device(self.device):
            st = _hip.current_stream_ptr()

            def packed(w):
                out = torch.empty_like(w)
                _hip.check(L.oasr_gemm_pack_w_bf16(w.data_ptr(), w.shape[0], w.shape[1], out.data_ptr(), st),
                           "oasr_gemm_pack_w_bf16")
                return out

            def packed8(w):
                out = torch.empty_like(w)
                _hip.check(L.oasr_gemm_pack_w_fp8(w.data_ptr(), w.shape[0], w.shape[1], out.data_ptr(), st),
                           "oasr_gemm_pack_w_fp8")
                return out

            self._packed["top"]["fp_w"] = packed(self._packed["top"]["fp_w"])
            for layer in self._packed["layers"]:
                for name in self.GEMM_WEIGHTS:
                    layer[name] = packed(layer[name])
                    if self.fp8:
                        layer[name + "_q"] = packed8(layer[name + "_q"])
            if "ctc_w" in self._packed:
                self._packed["ctc_w"] = packed(self._packed["ctc_w"])
            torch.cuda.synchronize()
        self._packed["gemm_weights_packed"] = True

    def _create_handle(self):
        g = self.geometry
        cfg = _hip.EncoderConfig(g.feat_dim, g.d_model, g.ffn_dim, g.n_heads, g.n_layers,
                                 g.conv_kernel, g.left_max, g.right_max, g.ln_eps)
        top = _hip.TopWeights(**{k: self._packed["top"][k].data_ptr() for k in _hip.TOP_FIELDS})
        arr = (_hip.LayerWeights * max(1, g.n_layers))()
        for i, L in enumerate(self._packed["layers"]):
            for k in _hip.LAYER_FIELDS:
                setattr(arr[i], k, L[k].data_ptr())
        handle = C.c_void_p()
        _hip.check(_hip.lib().oasr_encoder_create(C.byref(cfg), C.byref(top), arr, C.byref(handle)),
                   "oasr_encoder_create")
        _hip.check(_hip.lib().oasr_encoder_set_packed_weights(handle, int(self._packed["gemm_weights_packed"])),
                   "oasr_encoder_set_packed_weights")
        if self.fp8:
            q = (_hip.LayerFp8Weights * max(1, g.n_layers))()
            for i, L in enumerate(self._packed["layers"]):
                for k in _hip.FP8_FIELDS:
                    setattr(q[i], k, L[k].data_ptr())
            _hip.check(_hip.lib().oasr_encoder_set_fp8(handle, q), "oasr_encoder_set_fp8")
        return handle

    def close(self):
        if getattr(self, "_handle", None):
            _hip.lib().oasr_encoder_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- buffers --------------------------------------------------------------------------------------
    def _get_buffers(self, batch: int, n_samples_max: int):
        frames = _hip.num_frames(n_samples_max)
        t_pad = pad_frames(frames // 2)
        # even, and covers every frame of the widest row (2 * t_pad + 1 when the frame count is odd): the
        # per-utterance statistics run over all frames whatever the row's batch neighbours are
        frame_cap = 2 * t_pad + 2
        key = (batch, t_pad)
        buf = self._buffers.get(key)
        if buf is not None:
            self._buffers[key] = self._buffers.pop(key)          # most recently used last
        if buf is None:
            self._evict_buffers(self.max_buffer_sets - 1)
            dev, g = self.device, self.geometry
            rows = int(_hip.lib().oasr_encoder_rows(batch, t_pad))
            slots = int(_hip.lib().oasr_logmel_stats_slots(frame_cap))
            ws_bytes = int(_hip.lib().oasr_encoder_workspace_bytes(self._handle, batch, t_pad))
            buf = {
                "t_pad": t_pad, "frame_cap": frame_cap, "rows": rows,
                "logmel": torch.zeros((batch, frame_cap, 80), dtype=torch.float32, device=dev),
                "stats": torch.zeros((batch, slots, 80, 2), dtype=torch.float32, device=dev),
                "feats": torch.zeros((batch, t_pad, 160), dtype=torch.float32, device=dev),
                "lengths": torch.zeros((batch,), dtype=torch.int32, device=dev),
                "out": torch.zeros((rows, g.d_model), dtype=torch.float32, device=dev),
                "workspace": torch.zeros((ws_bytes,), dtype=torch.uint8, device=dev),
                "ws_bytes": ws_bytes,
                # CTC stage
                "hid_bf16": torch.zeros((rows, g.d_model), dtype=torch.bfloat16, device=dev),
                "logits": torch.zeros((rows, CTC_LOGIT_PITCH), dtype=torch.float32, device=dev),
                "tokens": torch.zeros((batch, t_pad), dtype=torch.int32, device=dev),
                "tok_frames": torch.zeros((batch, t_pad), dtype=torch.int32, device=dev),
                "counts": torch.zeros((batch,), dtype=torch.int32, device=dev),
                "ctc_ids": torch.zeros((int(_hip.lib().oasr_ctc_workspace_bytes(batch, t_pad)) // 4,),
                                       dtype=torch.int32, device=dev),
            }
            self._buffers[key] = buf
        return buf

    def _evict_buffers(self, keep: int) -> None:
        """Drop the least recently used buffer sets beyond `keep`, together with everything that holds
        raw pointers into them (captured graphs) or was sized for them (slots): a service that sees many
        distinct (batch, length) shapes would otherwise keep ~1 GB per shape for ever."""
        while len(self._buffers) > max(keep, 0):
            torch.cuda.synchronize(self.device)                  # nothing in flight may still use them
            (b, t_pad), _ = next(iter(self._buffers.items()))
            del self._buffers[(b, t_pad)]
            same = lambda bb, n: bb == b and pad_frames(_hip.num_frames(n) // 2) == t_pad
            self._graphs = {k: v for k, v in self._graphs.items() if not same(k[2][0], k[2][1])}
            self._tickets = {k: v for k, v in self._tickets.items() if not same(k[0], k[1])}
            self._pinned = {k: v for k, v in self._pinned.items() if not same(k[0], k[1])}

    # ---- stages ---------------------------------------------------------------------------------------
    def features(self, pcm: torch.Tensor, n_valid: torch.Tensor, buf=None):
        """pcm float32 [B, N] and n_valid int32 [B] on the device -> (feats [B, t_pad, 160], lengths)."""
        batch, stride = pcm.shape
        if buf is None:
            buf = self._get_buffers(batch, stride)
        st = _hip.current_stream_ptr()
        L = _hip.lib()
        _hip.check(L.oasr_logmel_f32(_hip._ptr(pcm, torch.float32), stride, _hip._ptr(n_valid, torch.int32),
                                     batch, buf["frame_cap"], self._tables.data_ptr(),
                                     buf["logmel"].data_ptr(), buf["stats"].data_ptr(), st), "oasr_logmel_f32")
        _hip.check(L.oasr_feature_norm_stack(buf["logmel"].data_ptr(), buf["stats"].data_ptr(),
                                             n_valid.data_ptr(), batch, buf["frame_cap"], buf["t_pad"],
                                             buf["feats"].data_ptr(), buf["lengths"].data_ptr(), st),
                   "oasr_feature_norm_stack")
        return buf["feats"], buf["lengths"]

    def encode_features(self, feats: torch.Tensor, lengths: torch.Tensor, n_layers: int = -1, buf=None):
        """feats float32 [B, t_pad, 160], lengths int32 [B] -> hidden float32 [B, t_pad, d] (a view)."""
        batch, t_pad, _ = feats.shape
        if buf is None:
            buf = self._get_buffers(batch, 160 * (2 * t_pad - 1) + 400)
        if buf["t_pad"] != t_pad:
            raise _hip.OasrError("feature pitch does not match the buffer set")
        _hip.check(_hip.lib().oasr_encoder_forward(
            self._handle, _hip._ptr(feats, torch.float32), _hip._ptr(lengths, torch.int32), batch, t_pad,
            buf["out"].data_ptr(), buf["workspace"].data_ptr(), buf["ws_bytes"], n_layers,
            _hip.current_stream_ptr()), "oasr_encoder_forward")
        g = self.geometry
        return buf["out"][: batch * t_pad].view(batch, t_pad, g.d_model)

    def resample_16k(self, x, rate_in: int) -> torch.Tensor:
        """Mono float32 signal(s) [N] or [B, N] at `rate_in` (host array or device tensor) -> float32
        device tensor at 16 kHz, by the polyphase kernel with the host-designed low-pass."""
        from . import audio as _audio
        xt = torch.as_tensor(x, dtype=torch.float32)
        squeeze = xt.dim() == 1
        if squeeze:
            xt = xt[None, :]
        xd = xt.to(self.device).contiguous()
        if int(rate_in) == SAMPLE_RATE:
            return xd[0] if squeeze else xd
        key = int(rate_in)
        tab = self._resample_tables.get(key)
        if tab is None:
            up, down, half, hp = _audio.polyphase_branches(key, SAMPLE_RATE)
            tab = (up, down, half, torch.from_numpy(hp).to(self.device))
            self._resample_tables[key] = tab
        up, down, half, hp = tab
        batch, n_in = xd.shape
        L = _hip.lib()
        n_out = int(L.oasr_resample_length(n_in, up, down))
        y = torch.empty((batch, n_out), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _hip.check(L.oasr_resample_poly(xd.data_ptr(), n_in, n_in, batch, hp.data_ptr(), up, down, hp.shape[1],
                                            half, y.data_ptr(), n_out, n_out, _hip.current_stream_ptr()),
                       "oasr_resample_poly")
        return y[0] if squeeze else y

    def pcm_to_16k(self, samples: np.ndarray, channels: int, rate_in: int, sample_bytes: int = 2, bits: int = 16,
                   on_device: bool = False):
        """Interleaved integer samples as a WAV file holds them (e.g. a memory-mapped file) or as the FLAC decoder
        writes them -> mono float32 at 16 kHz on the host.  `samples`: any contiguous array over the sample bytes;
        `sample_bytes` 1 (unsigned 8-bit), 2, 3 (packed 24-bit) or 4, value = integer / 2^(bits - 1).  The bytes go
        to the device through two page-locked staging blocks, the conversion, down-mix and rate conversion run
        there, and only the 16 kHz signal comes back - the host never holds a float copy of the file at its own
        rate.  Equal, bit for bit, to decoding on the host (float32 samples, channels averaged in float64)
        followed by `resample_16k`.  With `on_device` the 16 kHz signal stays where it is (a float32 device
        tensor): `submit(..., source=...)` assembles batches from it there."""
        raw = np.ascontiguousarray(samples).reshape(-1).view(np.uint8)
        frame_bytes = int(channels) * int(sample_bytes)
        if channels < 1 or sample_bytes not in (1, 2, 3, 4) or raw.shape[0] % frame_bytes:
            raise ValueError("pcm_to_16k wants whole frames of channels * sample_bytes bytes")
        total, frames = int(raw.shape[0]), int(raw.shape[0]) // frame_bytes
        if frames == 0:
            return np.zeros((0,), dtype=np.float32)
        block = 12 << 22                                                 # 48 MiB: a whole number of samples of any width
        with torch.cuda.device(self.device):
            dev = torch.empty((total,), dtype=torch.uint8, device=self.device)
            if self._pcm_stage is None:
                self._pcm_stage = [torch.empty((block,), dtype=torch.uint8).pin_memory() for _ in range(2)]
                self._pcm_stage_events = [None, None]
            # A staging block may be overwritten only once the copy that last read it has finished - also a copy
            # queued by an EARLIER call: with `on_device` this method returns without waiting for the device, so
            # the events live on the engine, not in the call.
            events = self._pcm_stage_events
            for i, off in enumerate(range(0, total, block)):
                n = min(block, total - off)
                stage = self._pcm_stage[i & 1]
                if events[i & 1] is not None:
                    events[i & 1].synchronize()                          # the block's previous copy has left it
                self._parallel_copy(stage.numpy(), raw, off, n)
                dev[off: off + n].copy_(stage[:n], non_blocking=True)
                events[i & 1] = torch.cuda.Event()
                events[i & 1].record()
            mono = torch.empty((frames,), dtype=torch.float32, device=self.device)
            _hip.check(_hip.lib().oasr_pcm_to_mono_f32(dev.data_ptr(), frames, int(channels), int(sample_bytes), int(bits),
                                                       mono.data_ptr(), _hip.current_stream_ptr()), "oasr_pcm_to_mono_f32")
            del dev
            y = self.resample_16k(mono, rate_in)
            return y if on_device else y.cpu().numpy()

    def _parallel_copy(self, dst: np.ndarray, src: np.ndarray, off: int, n: int, parts: int = 4) -> None:
        """dst[:n] = src[off : off + n] by a few threads (numpy releases the GIL while it copies): the source is a
        memory-mapped file, and one thread's page faults and copy leave most of the host's bandwidth unused."""
        if n < (8 << 20):
            dst[:n] = src[off: off + n]
            return
        if self._copy_pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._copy_pool = ThreadPoolExecutor(max_workers=parts, thread_name_prefix="oasr-stage")
        step = -(-n // parts)

        def piece(i):
            lo, hi = i * step, min(n, (i + 1) * step)
            if hi > lo:
                dst[lo:hi] = src[off + lo: off + hi]
        list(self._copy_pool.map(piece, range(parts)))

    def decode(self, hidden: torch.Tensor, lengths: torch.Tensor, buf=None):
        """Encoder states float32 [B, t_pad, d] (as returned by `encode_features`) and lengths int32 [B]
        -> (tokens int32 [B, t_pad], frames int32 [B, t_pad], counts int32 [B]) on the device: vocabulary
        projection (bf16 GEMM, float32 logits) and greedy CTC decoding."""
        if "ctc_w" not in self._packed:
            raise _hip.OasrError("this weight set has no CTC head")
        batch, t_pad, d = hidden.shape
        if buf is None:
            buf = self._get_buffers(batch, 160 * (2 * t_pad - 1) + 400)
        rows = buf["rows"]
        L, st = _hip.lib(), _hip.current_stream_ptr()
        if hidden.data_ptr() != buf["out"].data_ptr():
            buf["out"][: batch * t_pad].copy_(hidden.reshape(batch * t_pad, d))
        _hip.check(L.oasr_cast_bf16(buf["out"].data_ptr(), buf["hid_bf16"].data_ptr(), rows * d, st), "oasr_cast_bf16")
        gemm = L.oasr_gemm_bf16_pw if self._packed["gemm_weights_packed"] else L.oasr_gemm_bf16
        _hip.check(gemm(buf["hid_bf16"].data_ptr(), self._packed["ctc_w"].data_ptr(),
                                    self._packed["ctc_b"].data_ptr(), None, buf["logits"].data_ptr(), rows,
                                    CTC_LOGIT_PITCH, d, 1.0, _hip.EPI_BIAS_F32, st), "oasr_gemm_bf16 (ctc)")
        _hip.check(L.oasr_ctc_greedy(buf["logits"].data_ptr(), CTC_LOGIT_PITCH, self.geometry.vocab_size,
                                     _hip._ptr(lengths, torch.int32), batch, t_pad, tokenizer.BLANK,
                                     buf["tokens"].data_ptr(), buf["tok_frames"].data_ptr(),
                                     buf["counts"].data_ptr(), buf["ctc_ids"].data_ptr(), st), "oasr_ctc_greedy")
        return buf["tokens"], buf["tok_frames"], buf["counts"]

    def step(self, pcm: torch.Tensor, n_valid: torch.Tensor):
        """One pass of the whole device path over one resident batch (eager launches)."""
        buf = self._get_buffers(pcm.shape[0], pcm.shape[1])
        feats, lengths = self.features(pcm, n_valid, buf)
        return self.encode_features(feats, lengths, -1, buf), lengths

    def step_graph(self, pcm: torch.Tensor, n_valid: torch.Tensor):
        """Same as `step`, replayed from a HIP graph captured on first use for this (pcm buffer,
        shape): the ~200 launches of a forward become one graph launch."""
        key = (pcm.data_ptr(), n_valid.data_ptr(), tuple(pcm.shape))
        entry = self._graphs.get(key)
        if entry is None:
            self.step(pcm, n_valid)                  # warm-up: allocates buffers outside capture
            torch.cuda.synchronize(self.device)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                hidden, lengths = self.step(pcm, n_valid)
            entry = (graph, hidden, lengths)
            self._graphs[key] = entry
        graph, hidden, lengths = entry
        graph.replay()
        return hidden, lengths

    # ---- host-facing convenience ----------------------------------------------------------------------
    def _upload(self, pcm, n_valid):
        """Host batch -> resident device tensors.  Host arrays go through a page-locked staging buffer
        kept per shape (a pageable 61 MB batch costs ~10 ms to copy, a pinned one ~1 ms); the graph of
        `step_graph` is keyed on the device buffer, so equal shapes replay one captured graph."""
        if isinstance(pcm, torch.Tensor) and pcm.is_cuda:
            pcm_d = pcm.to(torch.float32).contiguous()
            if pcm_d.shape[1] % 4:
                pcm_d = torch.nn.functional.pad(pcm_d, (0, 4 - pcm_d.shape[1] % 4))
        else:
            host = np.asarray(pcm, dtype=np.float32) if not isinstance(pcm, torch.Tensor) else pcm.to(torch.float32).numpy()
            if host.ndim != 2:
                raise ValueError("pcm must be [batch, samples]")
            b, n = host.shape
            n4 = (n + 3) // 4 * 4
            stage, pcm_d, _ = self._slot(b, n4)
            if host.ctypes.data != stage.data_ptr():     # not already assembled in place (see host_batch)
                stage.numpy()[:, :n] = host
                if n4 != n:
                    stage.numpy()[:, n:] = 0.0
            pcm_d.copy_(stage, non_blocking=True)
        if pcm_d.dim() != 2:
            raise ValueError("pcm must be [batch, samples]")
        nv_host = torch.as_tensor(self._checked_lengths(n_valid, pcm_d.shape[0], pcm_d.shape[1]))
        slot = self._pinned.get(tuple(pcm_d.shape))
        if slot is not None and slot[1].data_ptr() == pcm_d.data_ptr():
            slot[2].copy_(nv_host)
            nv = slot[2]
        else:
            nv = nv_host.to(self.device)
        return pcm_d, nv

    def _slot(self, b: int, n4: int):
        slot = self._pinned.get((b, n4))
        if slot is None:
            slot = (torch.zeros((b, n4), dtype=torch.float32).pin_memory(),
                    torch.zeros((b, n4), dtype=torch.float32, device=self.device),
                    torch.zeros((b,), dtype=torch.int32, device=self.device))
            if len(self._pinned) >= 4:          # a few shapes at most: full batches and one tail
                self._pinned.pop(next(iter(self._pinned)))
            self._pinned[(b, n4)] = slot
        return slot

    @staticmethod
    def _checked_lengths(n_valid, batch: int, width: int) -> np.ndarray:
        """int32 [batch] valid-sample counts; the kernels trust them, so they are checked here."""
        nv = np.asarray(n_valid, dtype=np.int64).reshape(-1)
        if nv.shape[0] != batch:
            raise ValueError("n_valid has %d entries for a batch of %d" % (nv.shape[0], batch))
        if nv.size and (nv.min() < 0 or nv.max() > width):
            raise ValueError("n_valid must lie in [0, %d] (the batch width)" % width)
        return nv.astype(np.int32)

    @staticmethod
    def bucket_samples(n: int) -> int:
        """A batch width >= n (a multiple of 4) shared by every sample count that pads to the same number
        of encoder frames: batches assembled at this width share buffers, slots and captured graphs across
        nearby lengths (at most one shape per 32 encoder frames instead of one per distinct sample count)."""
        n = max(int(n), 400)
        t_pad = pad_frames(_hip.num_frames(n) // 2)
        # a row pads to t_pad up to 2 * t_pad + 1 frames, i.e. up to 320 * t_pad + 559 samples; the last
        # three of those do not fit a multiple of 4 and take the next bucket
        width = 320 * t_pad + 556
        return width if n <= width else 320 * (t_pad + 32) + 556

    def host_batch(self, b: int, n: int) -> np.ndarray:
        """A page-locked [b, round_up(n, 4)] float32 array to assemble a batch in; handing it (whole) to
        `encode_batch` / `transcribe_batch` skips the staging copy.  Contents persist between calls."""
        return self._slot(b, (n + 3) // 4 * 4)[0].numpy()

    def _run(self, pcm_d, nv):
        return self.step_graph(pcm_d, nv) if self.use_graphs else self.step(pcm_d, nv)

    def encode_batch(self, pcm, n_valid):
        """pcm: float32 [B, N] numpy array or tensor (host or device); n_valid: valid samples per row.
        Returns (hidden [B, t_pad, d] float32 device tensor, lengths numpy int32 [B])."""
        with torch.cuda.device(self.device):
            pcm_d, nv = self._upload(pcm, n_valid)
            hidden, lengths = self._run(pcm_d, nv)
            return hidden, lengths.cpu().numpy()

    def transcribe_batch(self, pcm, n_valid):
        """`encode_batch` plus the CTC stage.  Returns (hidden, lengths, tokens) where tokens is a list
        of (ids int32 array, frame-index int32 array) per utterance, on the host."""
        with torch.cuda.device(self.device):
            pcm_d, nv = self._upload(pcm, n_valid)
            hidden, lengths = self._run(pcm_d, nv)
            tok, frm, cnt = self.decode(hidden, lengths)
            tok_h, frm_h, cnt_h = tok.cpu().numpy(), frm.cpu().numpy(), cnt.cpu().numpy()
            out = [(tok_h[i, : cnt_h[i]].copy(), frm_h[i, : cnt_h[i]].copy()) for i in range(tok_h.shape[0])]
            return hidden, lengths.cpu().numpy(), out

    # ---- two-deep asynchronous path -------------------------------------------------------------------
    # submit() queues one batch (upload on a copy stream, device path, CTC stage, results to page-locked
    # memory) and returns at once; collect() waits for that batch only.  With two slots
    # the host assembles and uploads batch k+1 while the device works on batch k:
    #     a = host_batch_slot(b, n, 0); fill a; ta = submit(a-slot) ...
    # Intermediate device buffers are shared by both slots: batches run back to back on one stream.
    def _ticket(self, b: int, n4: int, which: int):
        key = (b, n4, which & 1)
        t = self._tickets.get(key)
        if t is None:
            buf = self._get_buffers(b, n4)
            t_pad = buf["t_pad"]
            pin = lambda shape, dt: torch.zeros(shape, dtype=dt).pin_memory()
            t = {
                "key": key, "stage": pin((b, n4), torch.float32), "nv_pin": pin((b,), torch.int32),
                "pcm_d": torch.zeros((b, n4), dtype=torch.float32, device=self.device),
                "nv_d": torch.zeros((b,), dtype=torch.int32, device=self.device),
                "tokens": pin((b, t_pad), torch.int32), "frames": pin((b, t_pad), torch.int32),
                "counts": pin((b,), torch.int32), "lengths": pin((b,), torch.int32),
                "copied": torch.cuda.Event(), "done": torch.cuda.Event(), "busy": False,
            }
            # the zero fills above are queued on the current stream, possibly behind a whole batch: the copy
            # stream waits for `done` before it writes the slot, so mark the fills as the slot's first use
            t["done"].record()
            if len(self._tickets) >= 6:         # full batches and one tail shape, two slots each
                old = self._tickets.pop(next(iter(self._tickets)))
                old["done"].synchronize()       # nothing may still be reading buffers about to be freed
            self._tickets[key] = t
        return t

    def host_batch_slot(self, b: int, n: int, which: int):
        """(ticket, page-locked [b, round_up(n, 4)] float32 array) of slot `which` (0 or 1).  Waits for
        the slot's previous batch if it is still in flight, so the array is free to overwrite."""
        with torch.cuda.device(self.device):
            t = self._ticket(b, (n + 3) // 4 * 4, which)
            if t["busy"]:
                t["done"].synchronize()
                t["busy"] = False
            return t, t["stage"].numpy()

    def submit(self, ticket, n_valid, source=None) -> None:
        """Queue the batch assembled in the ticket's array; returns without waiting for the device.
        `source` = (signal, starts): the batch is instead gathered on the device from a float32 device tensor
        (row i = signal[starts[i] : starts[i] + n_valid[i]], zeros after), and nothing is uploaded."""
        t = ticket
        if source is not None:
            return self._submit_from_device(t, n_valid, *source)
        with torch.cuda.device(self.device):
            if self._copy_stream is None:
                self._copy_stream = torch.cuda.Stream(self.device)
            main = torch.cuda.current_stream()
            t["nv_pin"].numpy()[:] = self._checked_lengths(n_valid, *t["stage"].shape)
            self._copy_stream.wait_event(t["done"])      # the slot's device copy may still feed its previous batch
            with torch.cuda.stream(self._copy_stream):
                t["pcm_d"].copy_(t["stage"], non_blocking=True)
                t["nv_d"].copy_(t["nv_pin"], non_blocking=True)
                t["copied"].record()
            main.wait_event(t["copied"])
            hidden, lengths = self._run(t["pcm_d"], t["nv_d"])
            tok, frm, cnt = self.decode(hidden, lengths)
            for name, src in (("tokens", tok), ("frames", frm), ("counts", cnt), ("lengths", lengths)):
                t[name].copy_(src, non_blocking=True)
            t["done"].record()
            t["busy"] = True

    def _submit_from_device(self, t, n_valid, signal: torch.Tensor, starts) -> None:
        b, width = t["stage"].shape
        nv = self._checked_lengths(n_valid, b, width)
        st = np.asarray(starts, dtype=np.int64).reshape(-1)
        if signal.dtype != torch.float32 or signal.dim() != 1 or not signal.is_cuda or st.shape[0] != b:
            raise ValueError("source must be (float32 1-D device tensor, one start per row)")
        if st.size and (st.min() < 0 or int((st + nv).max()) > signal.shape[0]):
            raise ValueError("chunk bounds lie outside the signal")
        with torch.cuda.device(self.device):
            # the slot's small arrays are reused: the previous batch of this slot (same stream) has long read them
            t["nv_pin"].numpy()[:] = nv
            if "starts_pin" not in t:
                t["starts_pin"] = torch.zeros((b,), dtype=torch.int64).pin_memory()
                t["starts_d"] = torch.zeros((b,), dtype=torch.int64, device=self.device)
            t["starts_pin"].numpy()[:] = st
            t["nv_d"].copy_(t["nv_pin"], non_blocking=True)
            t["starts_d"].copy_(t["starts_pin"], non_blocking=True)
            _hip.check(_hip.lib().oasr_gather_chunks(signal.data_ptr(), int(signal.shape[0]), t["starts_d"].data_ptr(),
                                                     t["nv_d"].data_ptr(), b, width, t["pcm_d"].data_ptr(),
                                                     _hip.current_stream_ptr()), "oasr_gather_chunks")
            hidden, lengths = self._run(t["pcm_d"], t["nv_d"])
            tok, frm, cnt = self.decode(hidden, lengths)
            for name, src in (("tokens", tok), ("frames", frm), ("counts", cnt), ("lengths", lengths)):
                t[name].copy_(src, non_blocking=True)
            t["done"].record()
            t["busy"] = True

    def collect(self, ticket):
        """Wait for the ticket's batch: (lengths int32 [B], [(token ids, frame indices)] per row).  The
        arrays are copies; the ticket can be reused at once."""
        t = ticket
        t["done"].synchronize()
        t["busy"] = False
        cnt = t["counts"].numpy()
        tok, frm = t["tokens"].numpy(), t["frames"].numpy()
        toks = [(tok[i, : cnt[i]].copy(), frm[i, : cnt[i]].copy()) for i in range(tok.shape[0])]
        return t["lengths"].numpy().copy(), toks
