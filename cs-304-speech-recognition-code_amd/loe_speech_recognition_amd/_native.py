"""ctypes binding of liblsr_hip.so (the C ABI declared in include/lsr_hip.h) and the
per-process device engine the drop-in classes call.

The HIP library is the only compute path: loading fails loudly when the shared object has not
been built, and creating the engine fails loudly without a GPU.  PyTorch is used for device
buffers, the stream handle and (in the distributed helpers) torch.distributed only.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_uint8, c_void_p
from typing import List, Optional, Sequence, Tuple

import numpy as np

DIM = 39
MAX_STATES = 127        # per decoding graph / alignment chain (the reference's int8 back-pointers)
NARROW_STATES = 64      # one wavefront per utterance up to here; two above
LL_STRIDE = 64          # emission score row stride for models of at most 64 states (128 above)
HOP = 160
N_FFT = 320
MIN_FRAMES = 9          # the width-9 delta filter needs 9 frames (librosa raises below that)
KERNEL_IDS = {"frame_index": 0, "mel_power": 1, "mfcc_finish": 2, "emission": 3, "viterbi": 4, "stats": 5,
              "separation": 6}

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_PKG_DIR, "..", "lib", "liblsr_hip.so"))

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


class NativeError(RuntimeError):
    pass


def _declare(lib) -> None:
    P = c_void_p
    sig = {
        "lsr_ctx_create": (c_int, [c_int, POINTER(c_void_p)]),
        "lsr_ctx_destroy": (c_int, [P]),
        "lsr_last_error": (c_char_p, [P]),
        "lsr_set_stream": (c_int, [P, P]),
        "lsr_synchronize": (c_int, [P]),
        "lsr_profile_enable": (c_int, [P, c_int]),
        "lsr_profile_reset": (c_int, [P]),
        "lsr_profile_get": (c_int, [P, c_int, POINTER(c_double), POINTER(c_int64)]),
        "lsr_host_hann_window": (None, [POINTER(c_double)]),
        "lsr_host_mel_filterbank": (None, [POINTER(c_float)]),
        "lsr_host_dct_matrix": (None, [POINTER(c_double)]),
        "lsr_model_upload": (c_int, [P, c_int, POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
        "lsr_model_set_mixtures": (c_int, [P, c_int, POINTER(c_double)]),
        "lsr_model_slot": (c_int, [P, c_int]),
        "lsr_loop_graph_upload": (c_int, [P, c_int, c_int, POINTER(c_float), POINTER(c_int32), POINTER(c_int32), c_int]),
        "lsr_loop_set_between": (c_int, [P, c_double, c_int]),
        "lsr_mfcc_batch": (c_int, [P, P, P, P, c_int, c_int64, P]),
        "lsr_mfcc_batch_i16": (c_int, [P, P, P, P, c_int, c_int64, P]),
        "lsr_recognize_batch_i16": (c_int, [P, P, P, P, c_int, c_int64, c_int, P, P, P, P, c_int]),
        "lsr_recognize_batch_noalloc": (c_int, [P, P, P, P, c_int, c_int64, c_int, P, P, P, P, c_int]),
        "lsr_recognize_host": (c_int, [P, P, P, c_int, c_int, c_int64, c_int, P, P, P, c_int]),
        "lsr_recognize_host_timing": (c_int, [P, POINTER(c_double), POINTER(c_int)]),
        "lsr_upload_host": (c_int, [P, P, P, c_int64, c_int, c_int64, c_int, P]),
        "lsr_download_host": (c_int, [P, P, c_int64, P, c_int64, c_int]),
        "lsr_host_check_finite": (c_int, [P, c_int]),
        "lsr_shard_bounds": (None, [c_int64, c_int, c_int, POINTER(c_int64), POINTER(c_int64)]),
        "lsr_comm_get_unique_id": (c_int, [P, P]),
        "lsr_comm_init": (c_int, [P, P, c_int, c_int]),
        "lsr_comm_destroy": (c_int, [P]),
        "lsr_allgather_labels": (c_int, [P, P, P, c_int64, c_int, c_int64, P, P]),
        "lsr_allreduce_stats": (c_int, [P, P, c_int64]),
        "lsr_label_record_bytes": (c_int64, [c_int]),
        "lsr_host_pack_label_records": (c_int, [P, P, c_int64, c_int, c_int64, P]),
        "lsr_host_unpack_label_records": (c_int, [P, c_int, c_int64, c_int64, c_int, P, P]),
        "lsr_pack_label_records": (c_int, [P, P, P, c_int64, c_int, c_int64, P]),
        "lsr_unpack_label_records": (c_int, [P, P, c_int, c_int64, c_int, P, P]),
        "lsr_labels_from_paths": (c_int, [P, P, P, c_int, P, P, c_int]),
        "lsr_ll_stride": (c_int, [P]),
        "lsr_decode_chain_batch_wide": (c_int, [P, P, P, c_int, c_int64, c_int, P, c_int64, P, P, P, P, P]),
        "lsr_chain_emission_mode": (c_int, [P, c_int]),
        "lsr_emission_batch": (c_int, [P, P, c_int64, P]),
        "lsr_decode_loop_batch": (c_int, [P, P, P, c_int, c_int64, c_int, P, P, P, P, c_int]),
        "lsr_decode_loop_sweep": (c_int, [P, P, P, c_int, c_int64, c_int, POINTER(c_double), POINTER(c_int32), c_int,
                                          P, P, P, c_int]),
        "lsr_recognize_batch": (c_int, [P, P, P, P, c_int, c_int64, c_int, P, P, P, P, c_int]),
        "lsr_decode_chain_batch": (c_int, [P, P, P, c_int, c_int64, c_int, P, c_int64, P, P, P, P, P, P]),
        "lsr_accumulate_stats": (c_int, [P, P, P, c_int64, c_int, P, P, P]),
        "lsr_remux_slots": (c_int, [P, P, P, P, P, P, c_int, c_int64, c_int, c_int, P, P]),
        "lsr_separation_scan": (c_int, [P, P, c_int, P, P, c_int, c_int64, c_int, c_double, c_double, c_int, c_int,
                                        P, P, P, P, P, P]),
        "lsr_gather_segments": (c_int, [P, P, c_int, P, P, c_int, c_int64, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)          # AttributeError here = the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args


EXPORTED_SYMBOLS = [
    "lsr_ctx_create", "lsr_ctx_destroy", "lsr_last_error", "lsr_set_stream", "lsr_synchronize",
    "lsr_profile_enable", "lsr_profile_reset", "lsr_profile_get", "lsr_host_hann_window",
    "lsr_host_mel_filterbank", "lsr_host_dct_matrix", "lsr_model_upload", "lsr_loop_graph_upload",
    "lsr_loop_set_between", "lsr_mfcc_batch", "lsr_emission_batch", "lsr_decode_loop_batch",
    "lsr_recognize_batch", "lsr_decode_chain_batch", "lsr_accumulate_stats", "lsr_remux_slots",
    "lsr_mfcc_batch_i16", "lsr_recognize_batch_i16", "lsr_model_set_mixtures", "lsr_separation_scan",
    "lsr_gather_segments", "lsr_decode_loop_sweep", "lsr_recognize_batch_noalloc",
    "lsr_recognize_host", "lsr_recognize_host_timing", "lsr_upload_host", "lsr_download_host", "lsr_host_check_finite", "lsr_shard_bounds", "lsr_comm_get_unique_id", "lsr_comm_init",
    "lsr_comm_destroy", "lsr_allgather_labels", "lsr_allreduce_stats", "lsr_label_record_bytes",
    "lsr_host_pack_label_records", "lsr_host_unpack_label_records", "lsr_pack_label_records", "lsr_unpack_label_records",
    "lsr_labels_from_paths", "lsr_ll_stride", "lsr_decode_chain_batch_wide", "lsr_chain_emission_mode",
    "lsr_model_slot",
]


def load_library():
    """Load liblsr_hip.so once.  No GPU is needed to load it (only to create a context)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `make -C {os.path.dirname(os.path.dirname(LIB_PATH))}` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        # PyTorch's ROCm build ships its own copy of the HIP runtime.  It has to be in the process before this
        # library is: loaded the other way round, the library binds to the system's runtime, and that second
        # runtime finds no device once PyTorch's has initialised the GPU.
        import torch  # noqa: F401

        lib = ctypes.CDLL(LIB_PATH)
        _declare(lib)
        _lib = lib
    return _lib


def host_tables() -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(hann window f64[320], mel filterbank f32[40,161], DCT matrix f64[13,40]) as the kernels use them."""
    lib = load_library()
    win = np.empty(320, np.float64)
    mel = np.empty((40, 161), np.float32)
    dct = np.empty((13, 40), np.float64)
    lib.lsr_host_hann_window(win.ctypes.data_as(POINTER(c_double)))
    lib.lsr_host_mel_filterbank(mel.ctypes.data_as(POINTER(c_float)))
    lib.lsr_host_dct_matrix(dct.ctypes.data_as(POINTER(c_double)))
    return win, mel, dct


def frames_of(n_samples: int) -> int:
    return 1 + int(n_samples) // HOP


class FeatureBatch:
    """Ragged batch of (T_b, 39) feature matrices resident on the device."""

    def __init__(self, feats, frame_off, lengths: np.ndarray):
        self.feats = feats                    # torch float32 [total_frames, 39]
        self.frame_off = frame_off            # torch int64 [B+1]
        self.lengths = np.asarray(lengths, dtype=np.int64)
        self.n = int(self.lengths.shape[0])
        self.total_frames = int(self.lengths.sum())
        self.max_frames = int(self.lengths.max())
        self.host_off = np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)


class AudioBatch:
    """Ragged batch of 16 kHz clips resident on the device."""

    def __init__(self, audio, sample_off, frame_off, n_samples: np.ndarray):
        self.audio = audio
        self.sample_off = sample_off
        self.frame_off = frame_off
        self.n_samples = np.asarray(n_samples, dtype=np.int64)
        self.lengths = 1 + self.n_samples // HOP
        self.n = int(self.n_samples.shape[0])
        self.total_frames = int(self.lengths.sum())
        self.max_frames = int(self.lengths.max())
        self.host_off = np.concatenate([[0], np.cumsum(self.lengths)]).astype(np.int64)


class Engine:
    """One liblsr_hip context on one GPU plus the buffer plumbing around it."""

    def __init__(self, device: Optional[int] = None):
        import torch

        global _hip_pid
        self.torch = torch
        self.lib = load_library()
        if _hip_pid is not None and _hip_pid != os.getpid():
            # the HIP runtime does not survive fork(): its state in the child is a copy of the parent's
            raise NativeError(
                f"this process (pid {os.getpid()}) was forked from pid {_hip_pid} after that process had initialised the "
                "GPU; HIP cannot be used in such a child.  Start workers with the 'spawn' method "
                "(multiprocessing.get_context('spawn')) or fork them before the first GPU call.")
        if not torch.cuda.is_available():
            raise NativeError("no GPU visible: liblsr_hip needs an AMD GPU (there is no CPU fallback)")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
        self.device_index = device
        _hip_pid = os.getpid()
        torch.cuda.set_device(device)
        self.device = torch.device("cuda", device)
        ctx = c_void_p()
        rc = self.lib.lsr_ctx_create(device, ctypes.byref(ctx))
        if rc != 0 or not ctx.value:
            raise NativeError(f"lsr_ctx_create({device}) failed with code {rc}")
        self.ctx = ctx
        self._check(self.lib.lsr_set_stream(self.ctx, c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        self._model_token = None
        self._graph_token = None
        self._rotated = {}               # token -> triangularised whitening matrices (at most 8 kept)
        # models resident on the device: token -> (slot, states); slot 0 takes the uploads without a token
        self._resident = {}
        self._slot = 0
        self._between = None

    # -- helpers ----------------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            msg = self.lib.lsr_last_error(self.ctx)
            raise NativeError(f"liblsr_hip error {rc}: {msg.decode() if msg else ''}")

    def _check_audio(self, rc: int) -> None:
        """_check for the calls that read host audio: a non-finite sample is the caller's ValueError (librosa's
        valid_audio raises "Audio buffer is not finite everywhere" for the reference, mfcc.py:31)."""
        if rc != 0:
            msg = self.lib.lsr_last_error(self.ctx)
            text = msg.decode() if msg else ""
            if "non-finite" in text:
                raise ValueError(f"Audio buffer is not finite everywhere ({text})")
        self._check(rc)

    def close(self) -> None:
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.lib.lsr_ctx_destroy(self.ctx)
            self.ctx = c_void_p()

    def synchronize(self) -> None:
        self._check(self.lib.lsr_synchronize(self.ctx))

    def use_stream(self, stream=None) -> None:
        """Launch the library's kernels on ``stream`` (a torch.cuda.Stream; None: torch's current stream) from now
        on.  Ordering against work already queued on the previous stream is the caller's business (events)."""
        if stream is None:
            stream = self.torch.cuda.current_stream(self.device)
        self._check(self.lib.lsr_set_stream(self.ctx, c_void_p(stream.cuda_stream)))

    @staticmethod
    def _ptr(t) -> c_void_p:
        return c_void_p(0) if t is None else c_void_p(t.data_ptr())

    def to_device(self, arr: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)

    def empty(self, shape, dtype):
        return self.torch.empty(shape, dtype=dtype, device=self.device)

    # -- profiling --------------------------------------------------------------------------------
    def profile_enable(self, on: bool, only: Optional[Sequence[str]] = None) -> None:
        """Event pairs around every kernel launch, or around the launches of the kernels named in ``only``."""
        mode = 0
        if on:
            mode = 1 if only is None else sum(2 << KERNEL_IDS[k] for k in set(only))
        self._check(self.lib.lsr_profile_enable(self.ctx, mode))

    def profile_reset(self) -> None:
        self._check(self.lib.lsr_profile_reset(self.ctx))

    def profile_get(self, kernel: str) -> Tuple[float, int]:
        ms, n = c_double(0), c_int64(0)
        self._check(self.lib.lsr_profile_get(self.ctx, KERNEL_IDS[kernel], ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    # -- batches ----------------------------------------------------------------------------------
    UPLOAD_PIPELINE_BYTES = 16 << 20      # lists of host arrays of at least this size go through lsr_upload_host

    def _upload_list(self, arrays, ndim: int, inner: int, audio: bool = False):
        """Device tensor holding the arrays of a list back to back, through the pinned / chunked upload pipeline,
        or None when the list does not qualify (small, mixed dtypes, non-contiguous): the caller then concatenates
        on the host and copies once.  ``audio``: float32 arrays are checked for non-finite samples on the way
        (ValueError, as librosa refuses such audio)."""
        if not isinstance(arrays, list) or len(arrays) == 0:
            return None
        first = arrays[0]
        if not isinstance(first, np.ndarray) or first.size * first.itemsize * len(arrays) < self.UPLOAD_PIPELINE_BYTES // 4:
            return None                                    # cheap estimate before touching every element
        from . import _lsr_pyhelper as H          # built by the Makefile next to this file

        ptrs = np.empty(len(arrays), np.uint64)
        sizes = np.empty(len(arrays), np.int64)
        kind = H.array_table(arrays, ptrs, sizes, ndim, inner)
        if kind < 0:
            return None
        dtype, item = (self.torch.int16, 2) if kind == H.INT16 else (self.torch.float32, 4)
        total = int(sizes.sum())
        if total * item < self.UPLOAD_PIPELINE_BYTES:
            return None
        out = self.empty((total,), dtype)
        self._check(self.lib.lsr_host_check_finite(self.ctx, 3 if audio else 1))
        try:
            self._check_audio(self.lib.lsr_upload_host(self.ctx, c_void_p(ptrs.ctypes.data), c_void_p(sizes.ctypes.data),
                                                       len(arrays), item, 0, 0, self._ptr(out)))
        finally:
            self.lib.lsr_host_check_finite(self.ctx, 1)
        return out

    def to_host(self, tensor) -> np.ndarray:
        """NumPy copy of a device tensor; large contiguous ones come back through lsr_download_host (pinned chunks
        unpacked by threads) instead of one synchronous copy into pageable memory."""
        nbytes = tensor.numel() * tensor.element_size()
        if nbytes < self.UPLOAD_PIPELINE_BYTES or not tensor.is_contiguous() or not tensor.is_cuda:
            return tensor.cpu().numpy()
        out = np.empty(tuple(tensor.shape), dtype=self.torch.empty(0, dtype=tensor.dtype).numpy().dtype)
        self._check(self.lib.lsr_download_host(self.ctx, self._ptr(tensor), nbytes, c_void_p(out.ctypes.data), 0, 0))
        return out

    def to_host_pinned(self, tensor) -> np.ndarray:
        """NumPy copy of a small device tensor through a pinned staging buffer that is kept between calls
        (no pageable bounce buffer, no allocation per call); the result is a fresh array."""
        if not tensor.is_cuda:
            return tensor.numpy().copy()
        t = self.torch
        n = tensor.numel()
        stage = self.__dict__.get("_pinned_stage")
        if stage is None or stage.dtype != tensor.dtype or stage.numel() < n:
            stage = t.empty((max(n, 1),), dtype=tensor.dtype, pin_memory=True)
            self.__dict__["_pinned_stage"] = stage
        stage[:n].copy_(tensor.reshape(-1), non_blocking=True)
        t.cuda.current_stream(self.device).synchronize()
        return stage[:n].numpy().reshape(tuple(tensor.shape)).copy()

    def audio_batch(self, signals: Sequence[np.ndarray]) -> AudioBatch:
        if len(signals) == 0:
            raise ValueError("empty batch")
        n = np.array([s.shape[0] for s in signals], dtype=np.int64)
        frames = 1 + n // HOP
        if frames.min() < MIN_FRAMES:
            raise ValueError(f"a clip yields {int(frames.min())} frames; the width-9 delta filter needs at least {MIN_FRAMES}")
        soff = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
        foff = np.concatenate([[0], np.cumsum(frames)]).astype(np.int64)
        audio = self._upload_list(signals, 1, 0, audio=True)
        if audio is None:
            # int16 PCM stays int16 (converted on the device); everything else is float32 like the reference's loader
            dtype = np.int16 if all(s.dtype == np.int16 for s in signals) else np.float32
            flat = np.concatenate([np.asarray(s, dtype=dtype) for s in signals])
            if dtype == np.float32 and not np.isfinite(flat).all():
                bad = next(i for i, s in enumerate(signals) if not np.isfinite(np.asarray(s, dtype=np.float32)).all())
                raise ValueError(f"Audio buffer is not finite everywhere (clip {bad})")
            audio = self.to_device(flat)
        return AudioBatch(audio, self.to_device(soff), self.to_device(foff), n)

    def feature_batch(self, feats: Sequence[np.ndarray]) -> FeatureBatch:
        if len(feats) == 0:
            raise ValueError("empty batch")
        lengths = np.array([f.shape[0] for f in feats], dtype=np.int64)
        if lengths.min() < 1:
            raise IndexError("index 0 is out of bounds for axis 0 with size 0")
        for f in feats:
            if f.ndim != 2 or f.shape[1] != DIM:          # a real check: the kernels index rows of DIM floats
                raise ValueError(f"feature arrays must have shape (T, {DIM}), got {f.shape}")
        foff = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        flat = self._upload_list(feats, 2, DIM)
        if flat is not None and flat.dtype == self.torch.float32:
            flat = flat.view(-1, DIM)
        else:
            flat = self.to_device(np.concatenate([np.asarray(f, dtype=np.float32) for f in feats], axis=0))
        return FeatureBatch(flat, self.to_device(foff), lengths)

    # -- front-end --------------------------------------------------------------------------------
    def mfcc(self, batch: AudioBatch):
        """float32 device tensor [total_frames, 39] (time-major)."""
        out = self.empty((batch.total_frames, DIM), self.torch.float32)
        fn = self.lib.lsr_mfcc_batch_i16 if batch.audio.dtype == self.torch.int16 else self.lib.lsr_mfcc_batch
        self._check(fn(self.ctx, self._ptr(batch.audio), self._ptr(batch.sample_off),
                                            self._ptr(batch.frame_off), batch.n, batch.total_frames, self._ptr(out)))
        return FeatureBatch(out, batch.frame_off, batch.lengths)

    # -- model / graph ----------------------------------------------------------------------------
    def set_model(self, token, mean: np.ndarray, whiten: np.ndarray, log_norm: np.ndarray,
                  log_weights: Optional[np.ndarray] = None, n_mix: int = 1, rotated: Optional[np.ndarray] = None,
                  triangular: Optional[bool] = None) -> None:
        """Upload stacked Gaussians.  By default the whitening matrices are first rotated to lower
        triangular form: U^T = Q R gives L = R^T with L L^T = U U^T, so |(x - mu) L|^2 is the same
        Mahalanobis distance (an orthogonal change of basis of the whitened vector) while the
        device skips the 4 x 4 blocks that are structurally zero (55 instead of 100 per state).
        ``rotated`` hands in those L when the caller already has them (the batch factorisation of a
        training step makes them in the same pass).  ``triangular=False`` (or LSR_DENSE_WHITEN=1)
        uploads SciPy's U unchanged."""
        if token is not None and token == self._model_token:
            return
        if triangular is None:
            triangular = os.environ.get("LSR_DENSE_WHITEN", "0") != "1"
        # a model that is still resident only needs its slot selected: callers that alternate between a few models
        # (a loop decoder and an isolated-word collection, say) switch tokens on every call
        key = (token, triangular, n_mix)
        hit = self._resident.get(key) if token is not None else None
        if hit is not None:
            self._resident[key] = self._resident.pop(key)               # most recently used last
            self._select_slot(hit[0])
            self._model_token = token
            self.n_model_states = hit[1]
            return
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        whiten = np.ascontiguousarray(whiten, dtype=np.float64)
        if triangular:
            cached = self._rotated.get(token) if token is not None else None
            if cached is None and rotated is not None:
                cached = np.ascontiguousarray(rotated, dtype=np.float64)
                assert cached.shape == whiten.shape
            elif cached is None:
                cached = np.tril(np.swapaxes(np.linalg.qr(np.swapaxes(whiten, 1, 2), mode="r"), 1, 2))      # batched
            if token is not None and rotated is None:
                if len(self._rotated) >= 8:
                    self._rotated.pop(next(iter(self._rotated)))
                self._rotated[token] = cached
            whiten = cached
        log_norm = np.ascontiguousarray(log_norm, dtype=np.float64)
        S = mean.shape[0]
        if mean.shape != (S, DIM) or whiten.shape != (S, DIM, DIM) or log_norm.shape != (S,):
            raise ValueError(f"model arrays must have shapes ({S}, {DIM}), ({S}, {DIM}, {DIM}), ({S},): got "
                             f"{mean.shape}, {whiten.shape}, {log_norm.shape}")
        if token is None:
            slot = 0
        else:
            used = {v[0] for v in self._resident.values()}
            free = [k for k in range(1, self.MODEL_SLOTS) if k not in used]
            if free:
                slot = free[0]
            else:                                                       # evict the least recently used
                oldest = next(iter(self._resident))
                slot = self._resident.pop(oldest)[0]
        self._select_slot(slot)
        dp = POINTER(c_double)
        self._model_token = None
        self._check(self.lib.lsr_model_upload(self.ctx, S, mean.ctypes.data_as(dp), whiten.ctypes.data_as(dp),
                                              log_norm.ctypes.data_as(dp)))
        if n_mix > 1:
            lw = np.ascontiguousarray(log_weights, dtype=np.float64)
            if lw.shape != (S,) or S % n_mix != 0:
                raise ValueError(f"mixture weights must have shape ({S},) with {S} a multiple of n_mix = {n_mix}")
            self._check(self.lib.lsr_model_set_mixtures(self.ctx, n_mix, lw.ctypes.data_as(dp)))
        self._model_token = token
        self.n_model_states = S // n_mix
        if token is not None:
            self._resident[key] = (slot, self.n_model_states)

    MODEL_SLOTS = 8                      # LSR_MODEL_SLOTS

    def _select_slot(self, slot: int) -> None:
        if slot != self._slot:
            self._check(self.lib.lsr_model_slot(self.ctx, slot))
            self._slot = slot

    def set_loop_graph(self, token, band: np.ndarray, lower: Sequence[int], upper: Sequence[int], silence_word: int) -> None:
        if token is not None and token == self._graph_token:
            return
        band = np.ascontiguousarray(band, dtype=np.float32)
        lo = np.ascontiguousarray(lower, dtype=np.int32)
        up = np.ascontiguousarray(upper, dtype=np.int32)
        S = band.shape[1]
        self._check(self.lib.lsr_loop_graph_upload(self.ctx, S, lo.shape[0], band.ctypes.data_as(POINTER(c_float)),
                                                   lo.ctypes.data_as(POINTER(c_int32)), up.ctypes.data_as(POINTER(c_int32)),
                                                   int(silence_word)))
        self._graph_token = token

    def set_between(self, between) -> None:
        """Python ints/floats and float32 scalars add in float32; np.float64 adds in float64 (NumPy promotion)."""
        wide = isinstance(between, np.float64)
        key = (float(between), wide)
        if key != self._between:
            self._check(self.lib.lsr_loop_set_between(self.ctx, float(between), 1 if wide else 0))
            self._between = key

    # -- emission / decoding ----------------------------------------------------------------------
    def emission(self, fb: FeatureBatch):
        rows = (fb.total_frames + 63) // 64 * 64
        out = self.empty((rows, int(self.lib.lsr_ll_stride(self.ctx))), self.torch.float32)
        self._check(self.lib.lsr_emission_batch(self.ctx, self._ptr(fb.feats), fb.total_frames, self._ptr(out)))
        return out

    def decode_loop(self, fb: FeatureBatch, want_paths: bool = False, label_cap: int = 32):
        t = self.torch
        scores = self.empty((fb.n,), t.float32)
        paths = self.empty((fb.total_frames,), t.int8) if want_paths else None
        labels = self.empty((fb.n, label_cap), t.uint8)
        n_labels = self.empty((fb.n,), t.int32)
        self._check(self.lib.lsr_decode_loop_batch(self.ctx, self._ptr(fb.feats), self._ptr(fb.frame_off), fb.n,
                                                   fb.total_frames, fb.max_frames, self._ptr(scores), self._ptr(paths),
                                                   self._ptr(labels), self._ptr(n_labels), label_cap))
        return scores, paths, labels, n_labels

    def decode_loop_sweep(self, fb: FeatureBatch, betweens: Sequence, label_cap: int = 32):
        """One emission pass, one loop decode per penalty: (scores [n,B], labels [n,B,cap], n_labels [n,B])."""
        t = self.torch
        n = len(betweens)
        if n == 0:
            raise ValueError("no penalties given")
        vals = np.array([float(b) for b in betweens], dtype=np.float64)
        wide = np.array([1 if isinstance(b, np.float64) else 0 for b in betweens], dtype=np.int32)
        scores = self.empty((n, fb.n), t.float32)
        labels = self.empty((n, fb.n, label_cap), t.uint8)
        n_labels = self.empty((n, fb.n), t.int32)
        self._check(self.lib.lsr_decode_loop_sweep(self.ctx, self._ptr(fb.feats), self._ptr(fb.frame_off), fb.n,
                                                   fb.total_frames, fb.max_frames, vals.ctypes.data_as(POINTER(c_double)),
                                                   wide.ctypes.data_as(POINTER(c_int32)), n, self._ptr(scores),
                                                   self._ptr(labels), self._ptr(n_labels), label_cap))
        return scores, labels, n_labels

    def recognize(self, ab: AudioBatch, want_paths: bool = False, label_cap: int = 32, out=None):
        t = self.torch
        if out is None:
            out = (self.empty((ab.n,), t.float32), self.empty((ab.total_frames,), t.int8) if want_paths else None,
                   self.empty((ab.n, label_cap), t.uint8), self.empty((ab.n,), t.int32))
        scores, paths, labels, n_labels = out
        fn = self.lib.lsr_recognize_batch_i16 if ab.audio.dtype == self.torch.int16 else self.lib.lsr_recognize_batch
        self._check(fn(self.ctx, self._ptr(ab.audio), self._ptr(ab.sample_off),
                                                 self._ptr(ab.frame_off), ab.n, ab.total_frames, ab.max_frames,
                                                 self._ptr(scores), self._ptr(paths), self._ptr(labels),
                                                 self._ptr(n_labels), label_cap))
        return out

    def labels_from_paths(self, paths: Sequence[np.ndarray], label_cap: int = 32):
        """Device label pass over given state paths (int8 arrays): (labels u8[B, cap], n_labels i32[B]) tensors."""
        lengths = np.array([len(p) for p in paths], dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        flat = np.concatenate([np.asarray(p, dtype=np.int8) for p in paths])
        d_paths, d_off = self.to_device(flat), self.to_device(off)
        labels = self.empty((len(paths), label_cap), self.torch.uint8)
        n_labels = self.empty((len(paths),), self.torch.int32)
        self._check(self.lib.lsr_labels_from_paths(self.ctx, self._ptr(d_paths), self._ptr(d_off), len(paths),
                                                   self._ptr(labels), self._ptr(n_labels), label_cap))
        self.synchronize()
        return labels, n_labels

    def recognize_host(self, signals: Sequence[np.ndarray], label_cap: int = 32, chunk_samples: int = 0,
                       threads: int = 0):
        """Host clips in, host results out through the chunked upload pipeline (lsr_recognize_host):
        (scores f32[B], labels u8[B, cap], n_labels i32[B]) as NumPy arrays.  ``signals`` are 1-D
        int16 arrays (all of them) or anything convertible to float32; no concatenation happens here,
        the library packs straight from the arrays into pinned staging memory."""
        if len(signals) == 0:
            raise ValueError("empty batch")
        from . import _lsr_pyhelper as H          # built by the Makefile next to this file

        clips = signals if isinstance(signals, list) else list(signals)
        ptrs = np.empty(len(clips), np.uint64)
        n = np.empty(len(clips), np.int64)
        kind = H.clip_table(clips, ptrs, n)
        if kind < 0:                              # mixed types, views, other dtypes: convert like audio_batch does
            is_i16 = all(isinstance(s, np.ndarray) and s.dtype == np.int16 for s in clips)
            clips = [np.ascontiguousarray(s, dtype=np.int16 if is_i16 else np.float32) for s in clips]
            kind = H.clip_table(clips, ptrs, n)
            if kind < 0:
                raise ValueError("clips must be 1-D arrays")
        is_i16 = kind == H.INT16
        if (1 + n // HOP).min() < MIN_FRAMES:
            raise ValueError(f"a clip yields {int((1 + n // HOP).min())} frames; the width-9 delta filter needs at least {MIN_FRAMES}")
        B = len(clips)
        scores = np.empty(B, np.float32)
        labels = np.empty((B, label_cap), np.uint8)
        n_labels = np.empty(B, np.int32)
        self._check_audio(self.lib.lsr_recognize_host(self.ctx, c_void_p(ptrs.ctypes.data), c_void_p(n.ctypes.data), B,
                                                      1 if is_i16 else 0, int(chunk_samples), int(threads),
                                                      c_void_p(scores.ctypes.data), c_void_p(labels.ctypes.data),
                                                      c_void_p(n_labels.ctypes.data), label_cap))
        return scores, labels, n_labels

    def recognize_host_timing(self) -> dict:
        """Host seconds of the last recognize_host call by phase (diagnostic)."""
        t = (c_double * 4)()
        n = c_int(0)
        self._check(self.lib.lsr_recognize_host_timing(self.ctx, t, ctypes.byref(n)))
        return {"wait_slot_s": t[0], "pack_s": t[1], "submit_s": t[2], "drain_s": t[3], "chunks": n.value}

    def recognize_noalloc(self, ab: AudioBatch, out):
        """lsr_recognize_batch_noalloc into caller-provided outputs (scores, paths or None, labels, n_labels):
        no device allocation happens inside the call (float32 audio, bounded batch size)."""
        scores, paths, labels, n_labels = out
        self._check(self.lib.lsr_recognize_batch_noalloc(self.ctx, self._ptr(ab.audio), self._ptr(ab.sample_off),
                                                         self._ptr(ab.frame_off), ab.n, ab.total_frames, ab.max_frames,
                                                         self._ptr(scores), self._ptr(paths), self._ptr(labels),
                                                         self._ptr(n_labels), labels.shape[1]))
        return out

    CHAIN_EMISSION = {"auto": 0, "all": 1, "reachable": 2}

    def chain_emission_mode(self, mode: str) -> None:
        """Emission scores computed ahead of a chain decode: every state ("all"), only the states the chains
        of a 64-frame tile can visit ("reachable"), or the library's choice ("auto", the default)."""
        self._check(self.lib.lsr_chain_emission_mode(self.ctx, self.CHAIN_EMISSION[mode]))

    def decode_chain(self, fb: FeatureBatch, chain_off: np.ndarray, chain_state: np.ndarray, chain_band: np.ndarray,
                     chain_start: np.ndarray, want_paths: bool = True, want_final: bool = False):
        t = self.torch
        chain_off = np.ascontiguousarray(chain_off, dtype=np.int64)
        total_chain = int(chain_off[-1])
        lens = np.diff(chain_off)
        if chain_off.shape[0] != fb.n + 1 or lens.min() < 1 or lens.max() > MAX_STATES:
            raise ValueError(f"every chain needs 1..{MAX_STATES} states")
        wide = int(lens.max()) > NARROW_STATES
        if wide and want_final:
            raise ValueError(f"last-frame scores of every chain position are only available up to {NARROW_STATES} states")
        chain_state = np.ascontiguousarray(chain_state, dtype=np.int32)
        if chain_state.shape != (total_chain,) or chain_state.min() < 0 or chain_state.max() >= self.n_model_states:
            raise ValueError("chain_state out of range of the uploaded model")
        chain_band = np.ascontiguousarray(chain_band, dtype=np.float32)
        if chain_band.shape != (3, total_chain):
            raise ValueError(f"chain_band must have shape (3, {total_chain}), got {chain_band.shape}")
        if np.isnan(chain_band).any():
            raise ValueError("NaN transition log-probability")
        chain_start = np.ascontiguousarray(chain_start, dtype=np.uint8)
        if chain_start.shape != (total_chain,):
            raise ValueError(f"chain_start must have shape ({total_chain},), got {chain_start.shape}")
        d_off, d_state = self.to_device(chain_off), self.to_device(chain_state)
        d_band, d_start = self.to_device(chain_band), self.to_device(chain_start)
        self.last_chain = (d_off, d_state)           # device copies, reused by remux_slots
        scores = self.empty((fb.n,), t.float32)
        paths = self.empty((fb.total_frames,), t.int8) if want_paths else None
        final = self.empty((fb.n, 64), t.float32) if want_final else None
        if wide:
            self._check(self.lib.lsr_decode_chain_batch_wide(self.ctx, self._ptr(fb.feats), self._ptr(fb.frame_off), fb.n,
                                                             fb.total_frames, fb.max_frames, self._ptr(d_off), total_chain,
                                                             self._ptr(d_state), self._ptr(d_band), self._ptr(d_start),
                                                             self._ptr(scores), self._ptr(paths)))
        else:
            self._check(self.lib.lsr_decode_chain_batch(self.ctx, self._ptr(fb.feats), self._ptr(fb.frame_off), fb.n,
                                                        fb.total_frames, fb.max_frames, self._ptr(d_off), total_chain,
                                                        self._ptr(d_state), self._ptr(d_band), self._ptr(d_start),
                                                        self._ptr(scores), self._ptr(paths), self._ptr(final)))
        # The chain arrays must outlive the kernels.  They were allocated on the stream the kernels run on, so the
        # caching allocator can only hand their memory to work that is ordered after those kernels; holding the
        # references until the next call keeps that true even if the caller switches torch streams in between.
        self._chain_keepalive = (d_off, d_state, d_band, d_start)
        return scores, paths, final

    def decode_chain_device(self, fb: FeatureBatch, d_off, total_chain: int, d_state, d_band, d_start, wide: bool = False,
                            want_paths: bool = True, want_final: bool = False):
        """decode_chain for a chain description that already lives on the device (int64 offsets [B + 1], int32
        model state, float32 band [3, total_chain], uint8 start flags per chain position), as a trainer keeps it
        between iterations: no host validation, no upload.  Returns (scores, paths), or (scores, paths, final)
        with ``want_final`` (last-frame scores of every chain position, chains of at most 64 states)."""
        t = self.torch
        if not (d_off.dtype == t.int64 and d_state.dtype == t.int32 and d_band.dtype == t.float32 and d_start.dtype == t.uint8):
            raise TypeError("chain description: offsets int64, states int32, band float32, start flags uint8")
        if not (d_off.shape == (fb.n + 1,) and d_state.shape == (total_chain,) and d_band.shape == (3, total_chain)
                and d_start.shape == (total_chain,) and d_band.is_contiguous()):
            raise ValueError(f"chain description does not match {fb.n} utterances / {total_chain} chain positions")
        if wide and want_final:
            raise ValueError("the last-frame score table is only available for chains of at most 64 states")
        scores = self.empty((fb.n,), t.float32)
        paths = self.empty((fb.total_frames,), t.int8) if want_paths else None
        final = self.empty((fb.n, 64), t.float32) if want_final else None
        fn = self.lib.lsr_decode_chain_batch_wide if wide else self.lib.lsr_decode_chain_batch
        args = [self.ctx, self._ptr(fb.feats), self._ptr(fb.frame_off), fb.n, fb.total_frames, fb.max_frames, self._ptr(d_off),
                total_chain, self._ptr(d_state), self._ptr(d_band), self._ptr(d_start), self._ptr(scores), self._ptr(paths)]
        if not wide:
            args.append(self._ptr(final))
        self._check(fn(*args))
        self.last_chain = (d_off, d_state)
        self._chain_keepalive = (d_off, d_state, d_band, d_start)
        return (scores, paths, final) if want_final else (scores, paths)

    def uniform_chains(self, key, n_utts: int, ids: np.ndarray, band: np.ndarray, start: np.ndarray):
        """Device description of ``n_utts`` copies of ONE chain (every utterance is decoded against the same
        states: an isolated word model, or a collection of them side by side): (offsets, total, states, band,
        start flags) for decode_chain_device.  Kept per (key, n_utts) - the key names the model the arrays were
        made from - so that a caller decoding batch after batch builds and uploads them once."""
        cache = self.__dict__.setdefault("_uniform_chains", {})
        hit = cache.get((key, n_utts))
        if hit is not None:
            cache[(key, n_utts)] = cache.pop((key, n_utts))
            return hit
        n = int(ids.shape[0])
        if not 1 <= n <= MAX_STATES:
            raise ValueError(f"every chain needs 1..{MAX_STATES} states")
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        band = np.ascontiguousarray(band, dtype=np.float32)
        if ids.min() < 0 or ids.max() >= self.n_model_states:
            raise ValueError("chain_state out of range of the uploaded model")
        if band.shape != (3, n) or np.isnan(band).any():
            raise ValueError("NaN transition log-probability" if band.shape == (3, n) else "band must be [3, states]")
        t = self.torch
        d_off = t.arange(n_utts + 1, dtype=t.int64, device=self.device) * n
        out = (d_off, n * n_utts, self.to_device(ids).repeat(n_utts), self.to_device(band).repeat(1, n_utts).contiguous(),
               self.to_device(np.ascontiguousarray(start, dtype=np.uint8)).repeat(n_utts))
        if len(cache) >= 8:
            cache.pop(next(iter(cache)))
        cache[(key, n_utts)] = out
        return out

    def remux_slots(self, paths, fb: "FeatureBatch", chain_off, chain_state, chain_word, n_states: int,
                    drop_last_run: bool):
        """(slots int32[total_frames], transition counts int64[n_states, n_states]) device tensors.
        chain_* are device tensors (int64 offsets, int32 state and word ordinal per chain position)."""
        t = self.torch
        slots = self.empty((fb.total_frames,), t.int32)
        trans = self.empty((n_states, n_states), t.int64)
        self._check(self.lib.lsr_remux_slots(self.ctx, self._ptr(paths), self._ptr(fb.frame_off), self._ptr(chain_off),
                                             self._ptr(chain_state), self._ptr(chain_word), fb.n, fb.total_frames,
                                             n_states, 1 if drop_last_run else 0, self._ptr(slots), self._ptr(trans)))
        return slots, trans

    def accumulate_stats(self, fb_feats, slots, total_frames: int, n_slots: int):
        """(count f64[n], sum_x f64[n,39], sum_xx f64[n,39,39]) device tensors."""
        t = self.torch
        count = self.empty((n_slots,), t.float64)
        sum_x = self.empty((n_slots, DIM), t.float64)
        sum_xx = self.empty((n_slots, DIM, DIM), t.float64)
        self._check(self.lib.lsr_accumulate_stats(self.ctx, self._ptr(fb_feats), self._ptr(slots), total_frames, n_slots,
                                                  self._ptr(count), self._ptr(sum_x), self._ptr(sum_xx)))
        return count, sum_x, sum_xx

    # -- silence stripping ------------------------------------------------------------------------
    def separation_scan(self, signals: Sequence[np.ndarray], frame_size: int, high: float, low: float,
                        max_silence_frames: int, min_kept_frames: int = 9, want_energy: bool = False) -> dict:
        """Two-threshold scan of every clip.  Returns host arrays first/end/status int32[B],
        noise uint8[total scan frames], scan_off int64[B+1], sample_off int64[B+1] (+ energy
        float32 when asked) and the device audio buffer for gather_segments."""
        if len(signals) == 0:
            raise ValueError("empty batch")
        t = self.torch
        n = np.array([s.shape[0] for s in signals], dtype=np.int64)
        if n.min() < 1:
            raise ValueError("zero-size array to reduction operation maximum which has no identity")
        dtype = np.int16 if all(s.dtype == np.int16 for s in signals) else np.float32
        flat = np.concatenate([np.asarray(s, dtype=dtype) for s in signals])
        soff = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
        scan_off = np.concatenate([[0], np.cumsum(n // frame_size + 1)]).astype(np.int64)
        total = int(scan_off[-1])
        B = n.shape[0]
        d_audio, d_soff, d_scan = self.to_device(flat), self.to_device(soff), self.to_device(scan_off)
        first, end, status = (self.empty((B,), t.int32) for _ in range(3))
        noise = self.empty((total,), t.uint8)
        energy = self.empty((total,), t.float32) if want_energy else None
        volume = self.empty((B,), t.float32)
        self._check(self.lib.lsr_separation_scan(self.ctx, self._ptr(d_audio), 1 if dtype == np.int16 else 0,
                                                 self._ptr(d_soff), self._ptr(d_scan), B, total, int(frame_size),
                                                 float(high), float(low), int(max_silence_frames), int(min_kept_frames),
                                                 self._ptr(first), self._ptr(end), self._ptr(status), self._ptr(noise),
                                                 self._ptr(energy), self._ptr(volume)))
        out = {"first": first.cpu().numpy(), "end": end.cpu().numpy(), "status": status.cpu().numpy(),
               "noise": noise.cpu().numpy(), "max_volume": volume.cpu().numpy(), "scan_off": scan_off,
               "sample_off": soff, "audio": d_audio}
        if want_energy:
            out["energy"] = energy.cpu().numpy()
        return out

    def gather_segments(self, src, src_start: np.ndarray, lengths: np.ndarray):
        """Device concatenation of slices src[src_start[i] : src_start[i] + lengths[i]]."""
        lengths = np.asarray(lengths, dtype=np.int64)
        src_start = np.ascontiguousarray(src_start, dtype=np.int64)
        if lengths.shape[0] == 0 or lengths.min() < 0 or src_start.min() < 0 or \
                int((src_start + lengths).max()) > src.shape[0]:
            raise ValueError("segments outside the source buffer")
        dst_off = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        dst = self.empty((int(dst_off[-1]),), src.dtype)
        d_start, d_off = self.to_device(src_start), self.to_device(dst_off)
        self._check(self.lib.lsr_gather_segments(self.ctx, self._ptr(src), src.element_size(), self._ptr(d_start),
                                                 self._ptr(d_off), lengths.shape[0], int(lengths.max()), self._ptr(dst)))
        self._gather_keepalive = (d_start, d_off, src)      # as in decode_chain: no host synchronisation needed
        return dst, dst_off


_engine: Optional[Engine] = None
_engine_pid: Optional[int] = None
_hip_pid: Optional[int] = None          # pid of the process in which an Engine first touched the GPU


def get_engine() -> Engine:
    """Process-wide engine, created on first use.  A spawned worker, or one forked before any GPU call, creates
    its own; a worker forked AFTER the parent used the GPU gets a NativeError (HIP does not survive fork)."""
    global _engine, _engine_pid
    if _engine is None or _engine_pid != os.getpid():
        _engine = Engine()                # raises in a child forked from a process that initialised the GPU
        _engine_pid = os.getpid()
    return _engine


_token_counter = 0


def new_token() -> Tuple[int, int]:
    """Identity of a set of uploaded arrays; unique within and across processes."""
    global _token_counter
    _token_counter += 1
    return (os.getpid(), _token_counter)


def records_to_strings(labels: np.ndarray, n_labels: np.ndarray, names: Sequence[str]) -> List[str]:
    """Label strings of fixed-width label records: row i holds min(n_labels[i], cap) word indices into ``names``
    (a negative count reads as no label).  One-character names - the digit labels - take a vectorised path: a
    10 000-utterance batch otherwise spends longer in the per-utterance join than on the device."""
    labels = np.asarray(labels)
    cap = labels.shape[1]
    cnt = np.clip(np.asarray(n_labels, dtype=np.int64), 0, cap)
    if labels.shape[0] and len(names) <= 256 and all(len(n) == 1 and 0 < ord(n) < 128 for n in names):
        lut = np.zeros(256, np.uint8)
        lut[:len(names)] = [ord(n) for n in names]
        chars = lut[labels]
        chars[np.arange(cap)[None, :] >= cnt[:, None]] = 0          # fixed-width byte strings drop trailing NULs
        return np.ascontiguousarray(chars).view(f"S{cap}").ravel().astype(str).tolist()
    return ["".join(names[w] for w in labels[i, :cnt[i]]) for i in range(labels.shape[0])]


def split_ragged(flat: np.ndarray, host_off: np.ndarray) -> List[np.ndarray]:
    return [flat[host_off[i]:host_off[i + 1]] for i in range(host_off.shape[0] - 1)]
