"""ctypes binding of liboasr_hip.so (the C ABI declared in include/oasr.h).

There is no CPU fallback: if the library cannot be loaded, `lib()` raises, and every compute entry
point needs a HIP device.  Tensors are torch-allocated; only raw device pointers, sizes and the
current HIP stream cross the boundary.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path

_LIB_NAME = "liboasr_hip.so"
_lock = threading.Lock()
_lib = None

c_i32, c_i64, c_f32, c_f64, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_void_p

EPI_BIAS_BF16, EPI_BIAS_SWISH_BF16, EPI_GLU_BF16, EPI_RESIDUAL_F32, EPI_BIAS_F32 = range(5)
GEMM_W_PACKED, GEMM_A_PACKED, GEMM_C_PACKED = 1, 2, 4    # layout flags of oasr_gemm_bf16_ex


class EncoderConfig(C.Structure):
    _fields_ = [("feat_dim", c_i32), ("d_model", c_i32), ("ffn_dim", c_i32), ("n_heads", c_i32),
                ("n_layers", c_i32), ("conv_kernel", c_i32), ("left_max", c_i32),
                ("right_max", c_i32), ("ln_eps", c_f32)]


LAYER_FIELDS = (
    "ffn1_ln_w", "ffn1_ln_b", "ffn1_w1", "ffn1_b1", "ffn1_w2", "ffn1_b2",
    "attn_ln_w", "attn_ln_b", "wqkv", "bqkv", "wo", "bo", "dist_emb",
    "conv_ln_w", "conv_ln_b", "pw1", "dw", "dw_ln_w", "dw_ln_b", "pw2",
    "ffn2_ln_w", "ffn2_ln_b", "ffn2_w1", "ffn2_b1", "ffn2_w2", "ffn2_b2",
    "final_ln_w", "final_ln_b",
)
TOP_FIELDS = ("fp_ln_w", "fp_ln_b", "fp_w", "fp_b")


class LayerWeights(C.Structure):
    _fields_ = [(n, c_vp) for n in LAYER_FIELDS]


FP8_FIELDS = tuple(n + sfx for n in ("ffn1_w1", "ffn1_w2", "wqkv", "wo", "pw1", "pw2", "ffn2_w1", "ffn2_w2")
                   for sfx in ("_q", "_s"))


class LayerFp8Weights(C.Structure):
    _fields_ = [(n, c_vp) for n in FP8_FIELDS]


class TopWeights(C.Structure):
    _fields_ = [(n, c_vp) for n in TOP_FIELDS]


class OasrError(RuntimeError):
    pass


def library_path() -> Path:
    env = os.environ.get("OASR_HIP_LIBRARY")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent.parent / "lib" / _LIB_NAME


_SIGNATURES = {
    "oasr_version": (c_i32, []),
    "oasr_last_error": (C.c_char_p, []),
    "oasr_plan_chunks": (c_i32, [c_i64, c_i32, c_f64, c_f64, C.POINTER(c_i64), c_i64, C.POINTER(c_i64)]),
    "oasr_shard_range": (c_i32, [c_i64, c_i32, c_i32, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "oasr_num_frames": (c_i64, [c_i64]),
    "oasr_num_stacked": (c_i64, [c_i64]),
    "oasr_logmel_tables": (c_i32, [C.POINTER(c_f32), C.POINTER(c_f32)]),
    "oasr_logmel_table_words": (c_i64, []),
    "oasr_logmel_table_blob": (c_i32, [C.POINTER(C.c_uint32), c_i64]),
    "oasr_logmel_stats_slots": (c_i64, [c_i64]),
    "oasr_logmel_f32": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "oasr_feature_norm_stack": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "oasr_encoder_create": (c_i32, [C.POINTER(EncoderConfig), C.POINTER(TopWeights),
                                    C.POINTER(LayerWeights), C.POINTER(c_vp)]),
    "oasr_encoder_destroy": (None, [c_vp]),
    "oasr_encoder_rows": (c_i64, [c_i32, c_i64]),
    "oasr_encoder_workspace_bytes": (c_i64, [c_vp, c_i32, c_i64]),
    "oasr_encoder_forward": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "oasr_layernorm": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_f32, c_i32, c_i32, c_vp]),
    "oasr_layernorm2": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_vp]),
    "oasr_gemm_bf16": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "oasr_gemm_bf16_pw": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "oasr_gemm_pack_w_bf16": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "oasr_gemm_bf16_ex": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_i32, c_i32, c_vp]),
    "oasr_encoder_set_packed_weights": (c_i32, [c_vp, c_i32]),
    "oasr_encoder_set_fp8": (c_i32, [c_vp, c_vp]),
    "oasr_layernorm_fp8": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_i32, c_vp]),
    "oasr_layernorm2_fp8": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_vp]),
    "oasr_quant_fp8_rows": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "oasr_gemm_fp8": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "oasr_gemm_fp8_pw": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i64, c_i64, c_f32, c_i32, c_vp]),
    "oasr_gemm_pack_w_fp8": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "oasr_attention_relkey": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "oasr_dwconv_causal": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp]),
    "oasr_dwconv_ln_swish": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_f32, c_vp]),
    "oasr_flac_info": (c_i32, [c_vp, c_i64, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i64), c_vp]),
    "oasr_flac_decode": (c_i32, [c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i64)]),
    "oasr_flac_decode16": (c_i32, [c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i64)]),
    "oasr_flac_decode_range": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_i32]),
    "oasr_adpcm_ima_decode": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, C.POINTER(c_i64)]),
    "oasr_mp3_info": (c_i32, [c_vp, c_i64, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i64)]),
    "oasr_mp3_decode": (c_i32, [c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i64), c_vp]),
    "oasr_mp3_decode_mono": (c_i32, [c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i64), c_vp]),
    "oasr_mp3_open": (c_i32, [c_vp, c_i64, C.POINTER(c_vp), C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i64)]),
    "oasr_mp3_decode_mono_range": (c_i32, [c_vp, c_i64, c_i64, c_vp]),
    "oasr_mp3_close": (None, [c_vp]),
    "oasr_vorbis_info": (c_i32, [c_vp, c_i64, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i64)]),
    "oasr_vorbis_decode": (c_i32, [c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i64), c_i32]),
    "oasr_vorbis_decode_mono": (c_i32, [c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i64), c_i32]),
    "oasr_vorbis_open": (c_i32, [c_vp, c_i64, C.POINTER(c_vp), C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i64)]),
    "oasr_vorbis_decode_mono_range": (c_i32, [c_vp, c_i64, c_i64, c_vp]),
    "oasr_vorbis_close": (None, [c_vp]),
    "oasr_vorbis_packet_tails": (c_i32, [c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i64)]),
    "oasr_ogg_flac_unwrap": (c_i32, [c_vp, c_i64, c_vp, c_i64, C.POINTER(c_i64)]),
    "oasr_gather_chunks": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp]),
    "oasr_pcm_to_mono_f32": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "oasr_resample_length": (c_i64, [c_i64, c_i32, c_i32]),
    "oasr_resample_poly": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp]),
    "oasr_cast_bf16": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "oasr_ctc_workspace_bytes": (c_i64, [c_i32, c_i64]),
    "oasr_ctc_static_max_frames": (c_i64, []),
    "oasr_ctc_greedy_static": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "oasr_ctc_greedy": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib() -> C.CDLL:
    """Load liboasr_hip.so once.  Raises OasrError when it is missing — there is no fallback."""
    global _lib
    with _lock:
        if _lib is None:
            path = library_path()
            if not path.exists():
                raise OasrError(
                    f"{path} not found: build it with `make -C omnilingual-asr_amd/csrc` "
                    "(or __graft_entry__.build()); this package has no CPU fallback")
            # torch first: it ships its own HIP runtime, and the process must end up with one.  If this library
            # pulled in the system's copy before torch loaded its own (a host-only call such as the FLAC decoder
            # made before any engine exists), the engine created later would find no device.
            import torch  # noqa: F401
            handle = C.CDLL(str(path))
            for name, (res, args) in _SIGNATURES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            _lib = handle
        return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().oasr_last_error().decode("utf-8", "replace")
        raise OasrError(f"{what or 'liboasr_hip'} failed ({rc}): {msg}")


# ---- host helpers -----------------------------------------------------------------------------------

def plan_chunks(total_samples: int, sample_rate: int, chunk_seconds: float,
                min_chunking_seconds: float):
    """[(start_sample, end_sample)] — see oasr_plan_chunks in include/oasr.h."""
    import numpy as np

    n = c_i64(0)
    cap = 64
    while True:
        buf = (c_i64 * (2 * cap))()
        rc = lib().oasr_plan_chunks(int(total_samples), int(sample_rate), float(chunk_seconds),
                                    float(min_chunking_seconds), buf, cap, C.byref(n))
        if rc == -2:  # OASR_ECAPACITY
            cap = int(n.value)
            continue
        check(rc, "oasr_plan_chunks")
        return np.ctypeslib.as_array(buf)[: 2 * n.value].reshape(-1, 2).copy()


def shard_range(n_items: int, world: int, rank: int) -> tuple[int, int]:
    lo, hi = c_i64(0), c_i64(0)
    check(lib().oasr_shard_range(int(n_items), int(world), int(rank), C.byref(lo), C.byref(hi)),
          "oasr_shard_range")
    return int(lo.value), int(hi.value)


def num_frames(n_samples: int) -> int:
    return int(lib().oasr_num_frames(int(n_samples)))


def num_stacked(n_samples: int) -> int:
    return int(lib().oasr_num_stacked(int(n_samples)))


def logmel_tables():
    """(window[400], mel_weights[257, 80]) float32 numpy arrays."""
    import numpy as np

    win = np.zeros(400, dtype=np.float32)
    mel = np.zeros((257, 80), dtype=np.float32)
    check(lib().oasr_logmel_tables(win.ctypes.data_as(C.POINTER(c_f32)),
                                   mel.ctypes.data_as(C.POINTER(c_f32))), "oasr_logmel_tables")
    return win, mel


def logmel_table_blob():
    import numpy as np

    words = int(lib().oasr_logmel_table_words())
    blob = np.zeros(words, dtype=np.uint32)
    check(lib().oasr_logmel_table_blob(blob.ctypes.data_as(C.POINTER(C.c_uint32)), words),
          "oasr_logmel_table_blob")
    return blob


# ---- device helpers (torch tensors in, raw pointers out) --------------------------------------------

def _ptr(t, dtype=None):
    import torch

    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise OasrError("expected a tensor on a HIP device")
    if dtype is not None and t.dtype != dtype:
        raise OasrError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise OasrError("expected a contiguous tensor")
    return t.data_ptr()


def current_stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream


def require_device() -> None:
    import torch

    if not torch.cuda.is_available():
        raise OasrError("no HIP device visible: the omnilingual_asr compute path needs an AMD GPU "
                        "(gfx950); there is no CPU fallback")
