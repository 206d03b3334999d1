"""Host-side Python mirror of the C ABI in include/xvec_hip.h (ctypes; no torch types cross it).

The reference's interface for this path is a command line (Kaldi's `nnet3-xvector-compute`, call sites
egs/sre/v2/sid/nnet3/xvector/extract_xvectors_new.sh:86-93); the product is the C++/HIP library
`libxvec_hip.so` plus the drop-in executables under bin/.  This module only exists so that tests/ and
bench.py can drive the same C ABI from Python; it contains no compute and no fallback: if the shared
library is missing or no gfx950 device is present, calls fail loudly.
"""
import ctypes
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# XVEC_LIB: an alternative build of the same library (kernel experiments: tools/ab_libs.sh); never set in production
LIB_PATH = os.environ.get("XVEC_LIB") or os.path.join(_HERE, "libxvec_hip.so")
BIN_DIR = os.path.join(_HERE, "bin")

XV_OK = 0
XV_ERR_IO, XV_ERR_MODEL, XV_ERR_DEVICE, XV_ERR_ARG, XV_ERR_INTERNAL = 1, 2, 3, 4, 5
PREC_BF16X3, PREC_BF16, PREC_FP16, PREC_FP16X3, PREC_FP16X2, PREC_AUTO, PREC_FP16MX, PREC_FP16MX2, PREC_FP16X3E = range(9)
PREC_DEFAULT = -1   # XV_PREC_DEFAULT: the one policy of every entry point (fp16mx2 where the model allows, else fp16x3)
PRECISIONS = {"default": PREC_DEFAULT, "bf16x3": PREC_BF16X3, "bf16": PREC_BF16, "fp16": PREC_FP16, "fp16x3": PREC_FP16X3,
              "fp16x2": PREC_FP16X2, "auto": PREC_AUTO, "fp16mx": PREC_FP16MX, "fp16mx2": PREC_FP16MX2}
PRECISION_NAMES = {v: k for k, v in PRECISIONS.items()}
# MFMA issue time per algorithmic product in units of one fp16 16x16x32 pass (fp16mx: + one 4-bit 16x16x128 per four)
MFMA_PASSES = {"bf16x3": 3, "bf16": 1, "fp16": 1, "fp16x3": 3, "fp16x2": 2, "fp16mx": 1.25}
EPI_ACT, EPI_F32, EPI_STATS = 0, 1, 2


class XvError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("xvec_hip status %d: %s" % (status, msg))
        self.status = status


class ModelInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("input_dim", "output_dim", "left_context", "right_context",
                                              "min_frames", "num_layers", "output_is_segment", "reserved")]


class SegDesc(ctypes.Structure):
    _fields_ = [("hi", ctypes.c_void_p), ("lo", ctypes.c_void_p), ("ld", ctypes.c_int32),
                ("row_shift", ctypes.c_int32), ("k_len", ctypes.c_int32), ("gmax", ctypes.c_void_p),
                ("lo4", ctypes.c_void_p), ("lo4_scale", ctypes.c_void_p)]


class GemmDesc(ctypes.Structure):
    _fields_ = [("precision", ctypes.c_int32), ("epilogue", ctypes.c_int32), ("nseg", ctypes.c_int32),
                ("seg", SegDesc * 8),
                ("w_hi", ctypes.c_void_p), ("w_lo", ctypes.c_void_p), ("ldw", ctypes.c_int32),
                ("rows", ctypes.c_int32), ("n_pad", ctypes.c_int32),
                ("bias", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("offset", ctypes.c_void_p),
                ("relu", ctypes.c_int32), ("bn", ctypes.c_int32),
                ("out_hi", ctypes.c_void_p), ("out_lo", ctypes.c_void_p), ("ldo", ctypes.c_int32),
                ("out_f32", ctypes.c_void_p), ("ldf", ctypes.c_int32), ("m_valid", ctypes.c_int32),
                ("partial", ctypes.c_void_p), ("ldp", ctypes.c_int32), ("grp_range", ctypes.c_void_p),
                ("hip_stream", ctypes.c_void_p),
                ("w4", ctypes.c_void_p), ("ldw4", ctypes.c_int32), ("w4_scale", ctypes.c_void_p),
                ("gmax_out", ctypes.c_void_p),
                ("w4b", ctypes.c_void_p), ("ldw4b", ctypes.c_int32), ("w4b_scale", ctypes.c_void_p),
                ("out_lo4", ctypes.c_void_p), ("out_lo4_scale", ctypes.c_void_p),
                ("p8", ctypes.c_int32)]


class Calibration(ctypes.Structure):
    """xv_calibration: what xv_ctx_calibrate measured and chose."""
    _fields_ = [("chosen", ctypes.c_int32), ("checked", ctypes.c_int32), ("err_mx", ctypes.c_float), ("err_mx2", ctypes.c_float),
                ("checked_mx", ctypes.c_int32), ("err_lite", ctypes.c_float), ("lite_mask", ctypes.c_uint64),
                ("err_holdout", ctypes.c_float), ("checked_holdout", ctypes.c_int32), ("lite_dropped", ctypes.c_int32),
                ("tail", ctypes.c_float)]

    def as_dict(self):
        d = {"chosen": PRECISION_NAMES.get(self.chosen, str(self.chosen)), "checked": self.checked, "checked_mx": self.checked_mx,
             "err_mx": self.err_mx, "err_mx2": self.err_mx2, "tail": self.tail}
        if self.lite_mask:
            d["lite_mask"] = int(self.lite_mask)
            d["err_lite"] = self.err_lite
            d["err_holdout"] = self.err_holdout
            d["checked_holdout"] = self.checked_holdout
        if self.lite_dropped:
            d["lite_dropped"] = self.lite_dropped
        return d


# every symbol include/xvec_hip.h declares (tests check the library exports exactly these)
ABI_SYMBOLS = [
    "xv_last_error", "xv_version", "xv_model_load", "xv_model_load_rxfilename", "xv_model_free", "xv_model_info",
    "xv_model_macs", "xv_model_describe", "xv_model_pack", "xv_ctx_create", "xv_ctx_create_from_blob",
    "xv_ctx_create_from_device_blob", "xv_ctx_free",
    "xv_ctx_info", "xv_forward_batch", "xv_forward_batch_device", "xv_ctx_synchronize", "xv_ctx_set_profiling",
    "xv_ctx_profile_report", "xv_extract_utterances", "xv_ctx_calibrate", "xv_ctx_set_fast_mode", "xv_ctx_fast_mode",
    "xv_calibrate_table", "xv_ctx_set_calibration", "xv_ctx_model_fingerprint", "xv_ctx_share_calibration",
    "xv_ctx_set_calibration_file", "xv_calibration_file_read", "xv_calibration_file_publish", "xv_recognize_feature_pipeline", "xv_ctx_set_lite_layers", "xv_ctx_lite_layers",
    "xv_extract_table", "xv_frontend_cmvn_select", "xv_plan_chunks", "xv_ctx_create_broadcast", "xv_kernel_tdnn_gemm",
    "xv_backend_apply", "xv_segment_mean", "xv_pack_mx_residual", "xv_pack_mx_residual64", "xv_tile_mx_scales", "xv_pack_mx_weights", "xv_pack_mx_weights64",
]

_lib = None


def build(verbose=False):
    """Compile every HIP/C++ source for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libxvec_hip.so failed")
    return LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no Python/CPU fallback for the HIP path)" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (+ HSA runtime).  If torch is
    # going to be used in this process (tests, bench: device tensors / streams / torch.distributed) it must load
    # its runtime BEFORE libxvec_hip.so resolves the same soname, otherwise torch ends up on a mixed runtime and
    # reports "No HIP GPUs are available".  The command-line tools never load torch and use /opt/rocm's runtime.
    if os.environ.get("XVEC_NO_TORCH_PRELOAD", "") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(LIB_PATH)
    L.xv_last_error.restype = ctypes.c_char_p
    L.xv_version.restype = ctypes.c_char_p
    L.xv_model_macs.restype = ctypes.c_double
    L.xv_model_macs.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    L.xv_model_describe.restype = ctypes.c_size_t
    L.xv_model_describe.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.xv_model_load.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p,
                                ctypes.POINTER(ctypes.c_void_p)]
    L.xv_model_load_rxfilename.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                           ctypes.POINTER(ctypes.c_void_p)]
    L.xv_model_free.argtypes = [ctypes.c_void_p]
    L.xv_model_free.restype = None
    L.xv_model_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ModelInfo)]
    L.xv_model_pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    L.xv_ctx_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.xv_ctx_create_from_blob.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_void_p)]
    L.xv_ctx_create_from_device_blob.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                                 ctypes.POINTER(ctypes.c_void_p)]
    L.xv_pack_mx_residual.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.xv_pack_mx_residual64.argtypes = L.xv_pack_mx_residual.argtypes
    L.xv_pack_mx_weights.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.xv_pack_mx_weights64.argtypes = L.xv_pack_mx_weights.argtypes
    L.xv_tile_mx_scales.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    L.xv_ctx_free.argtypes = [ctypes.c_void_p]
    L.xv_ctx_free.restype = None
    L.xv_ctx_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ModelInfo), ctypes.POINTER(ctypes.c_int32),
                              ctypes.POINTER(ctypes.c_int32)]
    L.xv_forward_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    L.xv_forward_batch_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    L.xv_ctx_synchronize.argtypes = [ctypes.c_void_p]
    L.xv_ctx_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    L.xv_ctx_profile_report.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.xv_ctx_profile_report.restype = ctypes.c_size_t
    L.xv_extract_utterances.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                        ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.xv_ctx_create_broadcast.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int,
                                          ctypes.POINTER(ctypes.c_void_p)]
    L.xv_kernel_tdnn_gemm.argtypes = [ctypes.POINTER(GemmDesc)]
    L.xv_ctx_calibrate.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_float,
                                   ctypes.POINTER(Calibration)]
    L.xv_ctx_set_fast_mode.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    L.xv_ctx_fast_mode.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32)]
    L.xv_ctx_set_lite_layers.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    L.xv_ctx_lite_layers.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    L.xv_calibrate_table.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                     ctypes.c_int32, ctypes.c_float, ctypes.POINTER(Calibration)]
    L.xv_ctx_set_calibration.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_float]
    L.xv_ctx_model_fingerprint.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
    L.xv_ctx_share_calibration.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_float, ctypes.c_char_p,
                                           ctypes.POINTER(ctypes.c_int32)]
    L.xv_ctx_set_calibration_file.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    L.xv_calibration_file_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64),
                                           ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64)]
    L.xv_calibration_file_publish.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint64, ctypes.c_float,
                                              ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64),
                                              ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_uint64)]
    _lib = L
    return L


def _check(status):
    if status != XV_OK:
        raise XvError(status, lib().xv_last_error().decode(errors="replace"))


class Model:
    """Parsed nnet3 model lowered to a TDNN program (host only; works without a GPU)."""

    def __init__(self, raw=None, rxfilename=None, nnet_config=None, output_node=None):
        L = lib()
        self._h = ctypes.c_void_p()
        cfg = nnet_config.encode() if nnet_config else None
        node = output_node.encode() if output_node else None
        if raw is not None:
            buf = ctypes.create_string_buffer(bytes(raw), len(raw))
            _check(L.xv_model_load(buf, len(raw), cfg, node, ctypes.byref(self._h)))
        else:
            _check(L.xv_model_load_rxfilename(rxfilename.encode(), cfg, node, ctypes.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value and lib is not None:   # lib is None while the interpreter shuts down
            lib().xv_model_free(self._h)
            self._h = ctypes.c_void_p()

    @property
    def info(self):
        mi = ModelInfo()
        _check(lib().xv_model_info(self._h, ctypes.byref(mi)))
        return mi

    def macs(self, frames):
        return lib().xv_model_macs(self._h, int(frames))

    def describe(self):
        n = lib().xv_model_describe(self._h, None, 0)
        buf = ctypes.create_string_buffer(n)
        lib().xv_model_describe(self._h, buf, n)
        return buf.value.decode()

    def pack(self, precision=PREC_DEFAULT):
        n = ctypes.c_size_t(0)
        _check(lib().xv_model_pack(self._h, precision, None, ctypes.byref(n)))
        buf = ctypes.create_string_buffer(n.value)
        _check(lib().xv_model_pack(self._h, precision, buf, ctypes.byref(n)))
        return buf.raw[:n.value]


class Context:
    """Weights resident on one MI355X + workspaces.  Raises XvError(XV_ERR_DEVICE) without a gfx950 GPU."""

    def __init__(self, model=None, blob=None, device=0, precision=PREC_DEFAULT, device_blob=None):
        """device_blob = (device pointer as int, nbytes): the packed image already in this GPU's memory (e.g. the
        buffer a RCCL broadcast filled); the weights then never visit the host."""
        L = lib()
        self._h = ctypes.c_void_p()
        if device_blob is not None:
            _check(L.xv_ctx_create_from_device_blob(ctypes.c_void_p(int(device_blob[0])), int(device_blob[1]), device,
                                                    ctypes.byref(self._h)))
        elif blob is not None:
            self._blob = ctypes.create_string_buffer(bytes(blob), len(blob))
            _check(L.xv_ctx_create_from_blob(self._blob, len(blob), device, ctypes.byref(self._h)))
        else:
            _check(L.xv_ctx_create(model._h, device, precision, ctypes.byref(self._h)))
        mi = ModelInfo()
        p, d = ctypes.c_int32(), ctypes.c_int32()
        _check(L.xv_ctx_info(self._h, ctypes.byref(mi), ctypes.byref(p), ctypes.byref(d)))
        self.info, self.precision, self.device = mi, p.value, d.value

    def close(self):
        """Frees the context's device and pinned memory now (otherwise when the object is collected)."""
        if getattr(self, "_h", None) and self._h.value and lib is not None:
            lib().xv_ctx_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        self.close()

    def forward_batch(self, feats, row_offsets):
        """feats: float32 numpy [rows, input_dim]; row_offsets: int32 [B+1] -> numpy [B, output_dim]."""
        import numpy as np
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        offs = np.ascontiguousarray(row_offsets, dtype=np.int32)
        B = len(offs) - 1
        # segment-level models: one row per chunk; frame-level models: one row per input frame
        n_out = B if self.info.output_is_segment else int(offs[-1] - offs[0])
        out = np.empty((n_out, self.info.output_dim), dtype=np.float32)
        _check(lib().xv_forward_batch(self._h, feats.ctypes.data, offs.ctypes.data, B, out.ctypes.data))
        return out

    def forward_batch_device(self, feats_ptr, row_offsets, out_ptr, out_ld, stream=None):
        """Device pointers (ints); asynchronous on `stream` (a hipStream_t as int, None = context stream)."""
        import numpy as np
        offs = np.ascontiguousarray(row_offsets, dtype=np.int32)
        _check(lib().xv_forward_batch_device(self._h, feats_ptr, offs.ctypes.data, len(offs) - 1, out_ptr, out_ld,
                                             stream))

    def synchronize(self):
        _check(lib().xv_ctx_synchronize(self._h))

    def set_profiling(self, on):
        _check(lib().xv_ctx_set_profiling(self._h, 1 if on else 0))

    def profile_report(self):
        """[(label, launches, total_ms)] of everything recorded since the last call."""
        buf = ctypes.create_string_buffer(1 << 18)   # one call: reading the report resets it
        lib().xv_ctx_profile_report(self._h, buf, len(buf))
        rows = []
        for line in buf.value.decode().splitlines():
            label, calls, ms = line.split("\t")
            rows.append((label, int(calls), float(ms)))
        return rows

    def frontend(self, raw, raw_offsets, vad=None, cmn_window=300, center=True):
        """apply-cmvn-sliding (norm-vars=false) + select-voiced-frames on the device.  Returns (feats, offsets)."""
        import numpy as np
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        offs = np.ascontiguousarray(raw_offsets, dtype=np.int32)
        n = len(offs) - 1
        out = np.empty_like(raw)
        out_off = np.zeros(n + 1, dtype=np.int32)
        v = None if vad is None else np.ascontiguousarray(vad, dtype=np.float32)
        L = lib()
        L.xv_frontend_cmvn_select.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                              ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        _check(L.xv_frontend_cmvn_select(self._h, raw.ctypes.data, offs.ctypes.data, n, None if v is None else v.ctypes.data,
                                         cmn_window, 1 if center else 0, out.ctypes.data, out_off.ctypes.data))
        return out[:out_off[-1]], out_off

    def extract_table(self, feature_rspecifier, vector_wspecifier, chunk_size=-1, min_chunk_size=100, pad_input=True,
                      batch_frames=0):
        """What one nnet3-xvector-compute process does, on this context.  Returns (done, failed)."""
        L = lib()
        L.xv_extract_table.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int32, ctypes.c_int32,
                                       ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64),
                                       ctypes.POINTER(ctypes.c_int64)]
        done, failed = ctypes.c_int64(0), ctypes.c_int64(0)
        _check(L.xv_extract_table(self._h, feature_rspecifier.encode(), vector_wspecifier.encode(), chunk_size,
                                  min_chunk_size, 1 if pad_input else 0, batch_frames, ctypes.byref(done),
                                  ctypes.byref(failed)))
        return done.value, failed.value

    def calibrate(self, feats, row_offsets, tol=7.5e-5):
        """xv_ctx_calibrate on host chunks: picks the fastest arithmetic whose embeddings stay within tol of the three-pass
        ones ON THESE CHUNKS and switches the context to it.  Returns {"chosen": name, "checked", "err_mx", "err_mx2"}."""
        import numpy as np
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        offs = np.ascontiguousarray(row_offsets, dtype=np.int32)
        c = Calibration()
        _check(lib().xv_ctx_calibrate(self._h, feats.ctypes.data, offs.ctypes.data, len(offs) - 1, tol, ctypes.byref(c)))
        return c.as_dict()

    def calibrate_table(self, feature_rspecifier, chunk_size=-1, min_chunk_size=100, pad_input=True, max_utts=64, tol=7.5e-5):
        """The same on the first chunk of the first max_utts utterances of a feature table."""
        c = Calibration()
        _check(lib().xv_calibrate_table(self._h, feature_rspecifier.encode(), chunk_size, min_chunk_size, 1 if pad_input else 0,
                                        max_utts, tol, ctypes.byref(c)))
        return c.as_dict()

    @property
    def fast_mode(self):
        """Name of the arithmetic the fast chunks run in right now (xv_ctx_fast_mode)."""
        p = ctypes.c_int32()
        _check(lib().xv_ctx_fast_mode(self._h, ctypes.byref(p)))
        return PRECISION_NAMES.get(p.value, str(p.value))

    def set_fast_mode(self, name):
        _check(lib().xv_ctx_set_fast_mode(self._h, PRECISIONS[name]))

    @property
    def lite_mask(self):
        """Layers (bit = index) that run the 1.25-pass arithmetic inside the fp16mx2 context (xv_ctx_lite_layers)."""
        m = ctypes.c_uint64()
        _check(lib().xv_ctx_lite_layers(self._h, ctypes.byref(m)))
        return int(m.value)

    def set_lite_mask(self, mask):
        _check(lib().xv_ctx_set_lite_layers(self._h, ctypes.c_uint64(int(mask))))

    def set_calibration(self, enable=True, tol=7.5e-5):
        """extract_table then calibrates on the head of its own table before the first batch."""
        _check(lib().xv_ctx_set_calibration(self._h, 1 if enable else 0, tol))

    @property
    def model_fingerprint(self):
        """Fingerprint of the packed model image this context runs (what a calibration file names)."""
        v = ctypes.c_uint64()
        _check(lib().xv_ctx_model_fingerprint(self._h, ctypes.byref(v)))
        return int(v.value)

    def share_calibration(self, path, tol=7.5e-5, note=None):
        """The shared choice of a recipe (xv_ctx_share_calibration): applies the file's choice when it exists, else publishes this
        context's current choice atomically and adopts what the file then holds.  Returns "read" / "published" / "adopted"."""
        out = ctypes.c_int32(-1)
        _check(lib().xv_ctx_share_calibration(self._h, os.fsencode(path), ctypes.c_float(tol), note.encode() if note else None,
                                              ctypes.byref(out)))
        return ("read", "published", "adopted")[out.value]

    def set_calibration_file(self, path):
        """extract_table applies / creates the shared calibration file before its first batch (None: off)."""
        _check(lib().xv_ctx_set_calibration_file(self._h, os.fsencode(path) if path else None))

    def extract_utterances(self, feats, row_offsets, chunk_size=-1, min_chunk_size=100, pad_input=True):
        import numpy as np
        feats = np.ascontiguousarray(feats, dtype=np.float32)
        offs = np.ascontiguousarray(row_offsets, dtype=np.int32)
        n = len(offs) - 1
        out = np.zeros((n, self.info.output_dim), dtype=np.float32)
        ok = np.zeros(n, dtype=np.int32)
        _check(lib().xv_extract_utterances(self._h, feats.ctypes.data, offs.ctypes.data, n, chunk_size, min_chunk_size,
                                           1 if pad_input else 0, out.ctypes.data, ok.ctypes.data))
        return out, ok.astype(bool)


class Watchdog:
    """Bounded wait for the start-up collective of a multi-rank job (the ONE broadcast of the packed weights, dist_extract.py /
    bench.py): a rank that never joins - died while loading, a link that is down - would otherwise leave the others inside
    the collective without a word until someone kills the job.  After `seconds` (XVEC_BCAST_TIMEOUT, default 120: the other
    ranks enter the broadcast while rank 0 still reads, lowers and packs the model - on a cold box that alone can take many seconds) the process
    prints what it was waiting for and EXITS with status 3; the launcher then tears the other ranks down.  A plain exit of a
    fresh-started process - nothing is re-executed (a process that has touched the GPU must never exec)."""

    def __init__(self, what, seconds=None):
        import threading
        if seconds is None:
            try:
                seconds = float(os.environ.get("XVEC_BCAST_TIMEOUT", "120"))
            except ValueError:
                seconds = 120.0
        self.what, self.seconds = what, seconds
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self._t.start()

    def _fire(self):
        sys.stderr.write("ERROR (xvec_hip watchdog) rank %s: %s did not complete within %.0f s; exiting\n"
                         % (os.environ.get("RANK", "0"), self.what, self.seconds))
        sys.stderr.flush()
        os._exit(3)

    def cancel(self):
        self._t.cancel()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.cancel()
        return False


def create_broadcast(model, devices, precision=PREC_DEFAULT):
    """xv_ctx_create_broadcast: one process, one context per listed GPU; the packed weights are uploaded to the first
    device and broadcast with ONE ncclBroadcast (RCCL), each context is built from the image its device received."""
    L = lib()
    n = len(devices)
    devs = (ctypes.c_int * n)(*devices)
    handles = (ctypes.c_void_p * n)()
    _check(L.xv_ctx_create_broadcast(model._h, devs, n, precision, handles))
    out = []
    for i in range(n):
        c = Context.__new__(Context)
        c._h = ctypes.c_void_p(handles[i])
        mi = ModelInfo()
        p, d = ctypes.c_int32(), ctypes.c_int32()
        _check(L.xv_ctx_info(c._h, ctypes.byref(mi), ctypes.byref(p), ctypes.byref(d)))
        c.info, c.precision, c.device = mi, p.value, d.value
        out.append(c)
    return out


def recognize_feature_pipeline(rspecifier):
    """The reference's feature pipeline as text (xv_recognize_feature_pipeline): {"feats", "vad", "cmn_window", "min_cmn_window",
    "center"} when the string is exactly `ark:apply-cmvn-sliding ... | select-voiced-frames ... |` with options the device
    front-end implements, else None."""
    L = lib()
    L.xv_recognize_feature_pipeline.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_char_p, ctypes.c_size_t,
                                                ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int32),
                                                ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    found, w, mw, c = ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0), ctypes.c_int32(0)
    fb, vb = ctypes.create_string_buffer(4096), ctypes.create_string_buffer(4096)
    _check(L.xv_recognize_feature_pipeline(rspecifier.encode(), ctypes.byref(found), fb, 4096, vb, 4096, ctypes.byref(w), ctypes.byref(mw),
                                           ctypes.byref(c)))
    if not found.value:
        return None
    return {"feats": fb.value.decode(), "vad": vb.value.decode(), "cmn_window": w.value, "min_cmn_window": mw.value, "center": bool(c.value)}


def calibration_file_read(path):
    """The shared choice a calibration file holds: {"model": fingerprint, "precision": name, "lite_mask": int}, or None when the
    file does not exist (xv_calibration_file_read; XvError when it cannot be parsed)."""
    found, prec = ctypes.c_int32(0), ctypes.c_int32(-1)
    model, lite = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _check(lib().xv_calibration_file_read(os.fsencode(path), ctypes.byref(found), ctypes.byref(model), ctypes.byref(prec),
                                          ctypes.byref(lite)))
    if not found.value:
        return None
    return {"model": int(model.value), "precision": PRECISION_NAMES.get(prec.value, str(prec.value)), "lite_mask": int(lite.value)}


def calibration_file_publish(path, model, precision, lite_mask=0, tol=7.5e-5, note=None):
    """Publishes a choice unless the file exists (atomic: the first of several concurrent publishers wins); returns
    (published, what the file holds afterwards)."""
    won, prec = ctypes.c_int32(0), ctypes.c_int32(-1)
    m, lite = ctypes.c_uint64(0), ctypes.c_uint64(0)
    _check(lib().xv_calibration_file_publish(os.fsencode(path), ctypes.c_uint64(model), PRECISIONS[precision], ctypes.c_uint64(lite_mask),
                                             ctypes.c_float(tol), note.encode() if note else None, ctypes.byref(won), ctypes.byref(m),
                                             ctypes.byref(prec), ctypes.byref(lite)))
    return bool(won.value), {"model": int(m.value), "precision": PRECISION_NAMES.get(prec.value, str(prec.value)),
                             "lite_mask": int(lite.value)}


def plan_chunks(num_rows, chunk_size, min_chunk_size, pad_input, min_net_frames, cap=4096):
    """[(start, len, left_pad, right_pad)] or None when the utterance counts as failed (host logic, no GPU)."""
    arr = [(ctypes.c_int32 * cap)() for _ in range(4)]
    n = ctypes.c_int32(0)
    L = lib()
    L.xv_plan_chunks.argtypes = [ctypes.c_int32] * 6 + [ctypes.c_void_p] * 4 + [ctypes.POINTER(ctypes.c_int32)]
    st = L.xv_plan_chunks(num_rows, chunk_size, min_chunk_size, 1 if pad_input else 0, min_net_frames, cap,
                          arr[0], arr[1], arr[2], arr[3], ctypes.byref(n))
    if st != XV_OK:
        return None
    return [(arr[0][i], arr[1][i], arr[2][i], arr[3][i]) for i in range(n.value)]


def backend_apply(x, mean=None, transform=None, normalize=False, scaleup=True, device=0, return_ratio=False):
    """Speaker-level back-end on the device: ivector-subtract-global-mean -> transform-vec -> ivector-normalize-length
    (egs/sre/v2/run_sre10.sh:238-241), each stage optional.  x: [n, dim] float32.  Returns [n, out_dim] (and the length
    ratios when asked)."""
    import numpy as np
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, dim = x.shape
    mean_p = tr_p = None
    t_rows = t_cols = 0
    if mean is not None:
        mean = np.ascontiguousarray(mean, dtype=np.float32)
        if mean.shape != (dim,):
            raise XvError(XV_ERR_ARG, "mean has shape %s, vectors have dimension %d" % (mean.shape, dim))
        mean_p = mean.ctypes.data
    if transform is not None:
        transform = np.ascontiguousarray(transform, dtype=np.float32)
        t_rows, t_cols = transform.shape
        tr_p = transform.ctypes.data
    out = np.empty((n, t_rows if transform is not None else dim), dtype=np.float32)
    ratio = np.empty(n, dtype=np.float32)
    L = lib()
    L.xv_backend_apply.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    _check(L.xv_backend_apply(device, x.ctypes.data, n, dim, mean_p, tr_p, t_rows, t_cols, 1 if normalize else 0,
                              1 if scaleup else 0, out.ctypes.data, ratio.ctypes.data))
    return (out, ratio) if return_ratio else out


def segment_mean(x, segments, acc64=False, device=0):
    """ivector-mean on the device: row s of the result is the mean of x[segments[s]] (rows added in list order;
    fp32 accumulator like the per-speaker loop, fp64 with acc64=True like the global mean)."""
    import numpy as np
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, dim = x.shape
    off = np.zeros(len(segments) + 1, dtype=np.int32)
    off[1:] = np.cumsum([len(s) for s in segments])
    idx = np.ascontiguousarray(np.concatenate([np.asarray(s, dtype=np.int32) for s in segments]) if len(segments) else
                               np.zeros(0, np.int32), dtype=np.int32)
    out = np.empty((len(segments), dim), dtype=np.float32)
    L = lib()
    L.xv_segment_mean.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                  ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    _check(L.xv_segment_mean(device, x.ctypes.data, n, dim, off.ctypes.data, idx.ctypes.data if idx.size else None,
                             len(segments), 1 if acc64 else 0, out.ctypes.data))
    return out


def kernel_tdnn_gemm(desc):
    _check(lib().xv_kernel_tdnn_gemm(ctypes.byref(desc)))


def pack_mx_residual(w, w_hi_f16, segs, walk64=False):
    """e2m1 residual plane + E8M0 scales [n_pad, K / 32] (one per block of four K steps and lane group) of XV_PREC_FP16MX
    for one weight matrix (host; no GPU).
    w: float32 [n_pad, K]; w_hi_f16: uint16 [n_pad, K] (fp16 bit patterns); segs: [(source id, row shift, k_len)]."""
    import numpy as np
    w = np.ascontiguousarray(w, dtype=np.float32)
    hi = np.ascontiguousarray(w_hi_f16, dtype=np.uint16)
    n_pad, K = w.shape
    src = np.array([s[0] for s in segs], dtype=np.int32)
    shift = np.array([s[1] for s in segs], dtype=np.int32)
    klen = np.array([s[2] for s in segs], dtype=np.int32)
    assert int(klen.sum()) == K and K % 128 == 0
    w4 = np.zeros((n_pad, K // 128 * 64), dtype=np.uint8)
    sc = np.zeros((n_pad, K // 32), dtype=np.uint8)
    fn = lib().xv_pack_mx_residual64 if walk64 else lib().xv_pack_mx_residual   # walk64: the K order of tdnn_gemm_kernel_p8
    _check(fn(w.ctypes.data, hi.ctypes.data, n_pad, len(segs), src.ctypes.data, shift.ctypes.data,
              klen.ctypes.data, w4.ctypes.data, sc.ctypes.data))
    return w4, sc


def pack_mx_weights(w, segs, walk64=False):
    """4-bit image of a weight matrix for the second K walk of XV_PREC_FP16MX2 + its scales [n_pad, K / 32] (natural order).
    walk64: in the order of tdnn_gemm_kernel_p8's second walk, rows of 2 K bytes (GemmDesc.ldw4b = 2 K then)."""
    import numpy as np
    w = np.ascontiguousarray(w, dtype=np.float32)
    n_pad, K = w.shape
    src = np.array([s[0] for s in segs], dtype=np.int32)
    shift = np.array([s[1] for s in segs], dtype=np.int32)
    klen = np.array([s[2] for s in segs], dtype=np.int32)
    assert int(klen.sum()) == K and np.all(klen % (256 if walk64 else 128) == 0)
    w4b = np.zeros((n_pad, 2 * K if walk64 else K // 2), dtype=np.uint8)
    sc = np.zeros((n_pad, K // 32), dtype=np.uint8)
    fn = lib().xv_pack_mx_weights64 if walk64 else lib().xv_pack_mx_weights
    _check(fn(w.ctypes.data, n_pad, len(segs), src.ctypes.data, shift.ctypes.data, klen.ctypes.data, w4b.ctypes.data, sc.ctypes.data))
    return w4b, sc


def tile_mx_scales(scales, epilogue):
    """natural [n_pad, K / 32] scales of pack_mx_residual -> the staging order GemmDesc.w4_scale wants for `epilogue`."""
    import numpy as np
    sc = np.ascontiguousarray(scales, dtype=np.uint8)
    out = np.zeros(sc.size, dtype=np.uint8)
    _check(lib().xv_tile_mx_scales(sc.ctypes.data, sc.shape[0], sc.shape[1] * 32, int(epilogue), out.ctypes.data))
    return out
