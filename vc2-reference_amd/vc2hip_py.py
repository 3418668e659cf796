"""ctypes binding of libvc2hip.so (include/vc2hip.h) for the tests, bench.py and smoke().

Plumbing only: every method is one C-ABI call; there is no Python or CPU implementation of
the codec here, and construction fails loudly when the library or a GPU is missing."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VC2HIP_LIB") or os.path.join(HERE, "libvc2hip.so")   # (VC2HIP_LIB: A/B of two builds, tools only)

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")

# vc2hip_create_with_flags (include/vc2hip.h: each flag selects the slower / more general of two correct paths)
FLAGS = {"STORE32": 0x001, "NO_STREAM": 0x002, "NO_PAIR": 0x004, "NO_BANDPLANES": 0x008, "NO_HEADS": 0x010, "NO_CBR_INDEX": 0x020,
         "GENERIC_DWT": 0x040, "SINGLE_PASS_VBR": 0x080, "CBR_GENERAL": 0x100, "LD_DIAGONALS": 0x200,
         "PLANES8_ALWAYS": 0x400, "PLANES8_NEVER": 0x800, "TWO_PASS_VBR": 0x1000}

KERNELS = {"DD97": 0, "LeGall": 1, "DD137": 2, "Haar0": 3, "Haar1": 4, "Fidelity": 5, "Daub97": 6}
CF = {"444": 0, "422": 1, "420": 2}
MODES = {"HQ_ConstQ": 0, "HQ_CBR": 1, "LD": 2}


class Geom(C.Structure):
    _fields_ = [("luma_h", C.c_int), ("luma_w", C.c_int), ("chroma_h", C.c_int),
                ("chroma_w", C.c_int), ("depth", C.c_int), ("y_slices", C.c_int),
                ("x_slices", C.c_int)]


class PictureFormat(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("chroma_format", C.c_int),
                ("bit_depth", C.c_int), ("word_bytes", C.c_int), ("chroma_bit_depth", C.c_int)]


class CodingParams(C.Structure):
    _fields_ = [("kernel", C.c_int), ("depth", C.c_int), ("y_slices", C.c_int),
                ("x_slices", C.c_int), ("mode", C.c_int), ("q_index", C.c_int),
                ("compressed_bytes", C.c_int), ("prefix", C.c_int), ("scalar", C.c_int)]


class Vc2HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


EXPORTS = [
    "vc2hip_create", "vc2hip_create_with_flags", "vc2hip_create_on_stream", "vc2hip_destroy", "vc2hip_last_error",
    "vc2hip_error_string", "vc2hip_sync", "vc2hip_padded_size", "vc2hip_slice_size_is_valid",
    "vc2hip_quant_matrix", "vc2hip_slice_bytes", "vc2hip_dwt_forward", "vc2hip_dwt_inverse",
    "vc2hip_quantise_np", "vc2hip_dequantise_np", "vc2hip_dequantise_ld", "vc2hip_hq_pack",
    "vc2hip_hq_unpack", "vc2hip_ld_unpack", "vc2hip_cbr_qindices", "vc2hip_quantise_ld", "vc2hip_ld_pack",
    "vc2hip_ld_qindices", "vc2hip_encode_picture_ld", "vc2hip_set_streams", "vc2hip_raw_picture_bytes",
    "vc2hip_max_payload_bytes", "vc2hip_encode_picture_hq", "vc2hip_decode_picture_hq",
    "vc2hip_decode_picture_ld", "vc2hip_encode_batch_dev", "vc2hip_decode_batch_dev",
    "vc2hip_profile_enable", "vc2hip_profile_only", "vc2hip_profile_count", "vc2hip_profile_get", "vc2hip_profile_reset",
    "vc2hip_host_alloc", "vc2hip_host_free", "vc2hip_encode_picture_begin", "vc2hip_encode_picture_end",
    "vc2hip_decode_picture_begin", "vc2hip_decode_picture_end", "vc2hip_band_plane_bits",
]


def load_library():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; whichever copy is loaded first serves the whole
    # process, and torch cannot initialise on top of the system runtime.  Let torch win when present.
    if os.environ.get("VC2HIP_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.vc2hip_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.vc2hip_create_with_flags.argtypes = [C.c_int, C.c_uint, C.POINTER(vp)]
    lib.vc2hip_create_on_stream.argtypes = [C.c_int, vp, C.POINTER(vp)]
    lib.vc2hip_destroy.argtypes = [vp]
    lib.vc2hip_destroy.restype = None
    lib.vc2hip_last_error.argtypes = [vp]
    lib.vc2hip_last_error.restype = C.c_char_p
    lib.vc2hip_error_string.argtypes = [C.c_int]
    lib.vc2hip_error_string.restype = C.c_char_p
    lib.vc2hip_sync.argtypes = [vp]
    lib.vc2hip_set_streams.argtypes = [vp, C.c_int]
    lib.vc2hip_quant_matrix.argtypes = [C.c_int, C.c_int, i32p]
    lib.vc2hip_slice_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i32p]
    lib.vc2hip_dwt_forward.argtypes = [vp, i32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p]
    lib.vc2hip_dwt_inverse.argtypes = [vp, i32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int]
    for f in (lib.vc2hip_quantise_np, lib.vc2hip_dequantise_np, lib.vc2hip_dequantise_ld, lib.vc2hip_quantise_ld):
        f.argtypes = [vp, i32p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, i32p, i32p]
    lib.vc2hip_hq_pack.argtypes = [vp, i32p, i32p, i32p, C.POINTER(Geom), i32p, C.c_int, C.c_int,
                                   vp, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.vc2hip_hq_unpack.argtypes = [vp, u8p, C.c_size_t, C.POINTER(Geom), C.c_int, C.c_int, i32p,
                                     i32p, i32p, i32p, C.POINTER(C.c_size_t)]
    lib.vc2hip_ld_unpack.argtypes = [vp, u8p, C.c_size_t, C.POINTER(Geom), i32p, i32p, i32p, i32p,
                                     i32p, C.POINTER(C.c_size_t)]
    lib.vc2hip_cbr_qindices.argtypes = [vp, i32p, i32p, i32p, C.POINTER(Geom), i32p, i32p, C.c_int, i32p]
    lib.vc2hip_ld_qindices.argtypes = [vp, i32p, i32p, i32p, C.POINTER(Geom), i32p, i32p, i32p]
    lib.vc2hip_ld_pack.argtypes = [vp, i32p, i32p, i32p, C.POINTER(Geom), i32p, i32p, u8p, C.c_size_t,
                                   C.POINTER(C.c_size_t)]
    lib.vc2hip_raw_picture_bytes.argtypes = [C.POINTER(PictureFormat)]
    lib.vc2hip_raw_picture_bytes.restype = C.c_size_t
    lib.vc2hip_max_payload_bytes.argtypes = [C.POINTER(PictureFormat), C.POINTER(CodingParams)]
    lib.vc2hip_max_payload_bytes.restype = C.c_size_t
    for f in (lib.vc2hip_encode_picture_hq, lib.vc2hip_encode_picture_ld):
        f.argtypes = [vp, u8p, C.POINTER(PictureFormat), C.POINTER(CodingParams), u8p, C.c_size_t,
                      C.POINTER(C.c_size_t), vp]
    for f in (lib.vc2hip_decode_picture_hq, lib.vc2hip_decode_picture_ld):
        f.argtypes = [vp, u8p, C.c_size_t, C.POINTER(PictureFormat), C.POINTER(CodingParams), u8p]
    lib.vc2hip_encode_batch_dev.argtypes = [vp, vp, C.c_int, C.POINTER(PictureFormat),
                                            C.POINTER(CodingParams), vp, C.c_size_t, vp]
    lib.vc2hip_decode_batch_dev.argtypes = [vp, vp, C.c_size_t, vp, C.c_int, C.POINTER(PictureFormat),
                                            C.POINTER(CodingParams), vp]
    lib.vc2hip_profile_enable.argtypes = [vp, C.c_int]
    lib.vc2hip_profile_only.argtypes = [vp, C.c_char_p]
    lib.vc2hip_profile_count.argtypes = [vp]
    lib.vc2hip_profile_get.argtypes = [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int),
                                       C.POINTER(C.c_double)]
    lib.vc2hip_profile_reset.argtypes = [vp]
    lib.vc2hip_band_plane_bits.argtypes = [vp]
    lib.vc2hip_host_alloc.argtypes = [C.c_size_t]
    lib.vc2hip_host_alloc.restype = vp
    lib.vc2hip_host_free.argtypes = [vp]
    lib.vc2hip_host_free.restype = None
    lib.vc2hip_encode_picture_begin.argtypes = [vp, vp, C.POINTER(PictureFormat), C.POINTER(CodingParams), vp, C.c_size_t, vp,
                                                C.POINTER(C.c_int)]
    lib.vc2hip_encode_picture_end.argtypes = [vp, C.c_int, C.POINTER(C.c_size_t)]
    lib.vc2hip_decode_picture_begin.argtypes = [vp, vp, C.c_size_t, C.POINTER(PictureFormat), C.POINTER(CodingParams), vp,
                                                C.POINTER(C.c_int)]
    lib.vc2hip_decode_picture_end.argtypes = [vp, C.c_int]
    return lib


def picture_format(width, height, cf, bits, word_bytes=2, chroma_bits=0):
    return PictureFormat(width, height, CF[cf], bits, word_bytes, chroma_bits)


def coding_params(lib, fmt, kernel, depth, u, a, mode="HQ_ConstQ", q=0, s=0, prefix=0, scalar=1):
    """u / a are the reference's -u / -a slice sizes in units of 2**depth (EncodeParams.cpp:90-91)."""
    ch = fmt.height // 2 if fmt.chroma_format == 2 else fmt.height
    cw = fmt.width if fmt.chroma_format == 0 else fmt.width // 2
    ys = lib.vc2hip_slice_size_is_valid(depth, fmt.height, ch, u)
    xs = lib.vc2hip_slice_size_is_valid(depth, fmt.width, cw, a)
    if not ys or not xs:
        raise ValueError("The given waveletDepth, hSlice, and vSlice parameters cannot encode this input.")
    return CodingParams(KERNELS[kernel], depth, ys, xs, MODES[mode], q, s, prefix, scalar)


class Vc2Hip:
    """One context (= one GPU, one stream)."""

    def __init__(self, device=0, stream=None, flags=0):
        self.lib = load_library()
        h = C.c_void_p()
        if stream is not None and flags:
            raise ValueError("vc2hip_create_on_stream takes no flags: give either a stream or flags")
        if stream is None and flags:
            rc = self.lib.vc2hip_create_with_flags(device, C.c_uint(flags), C.byref(h))
        elif stream is None:
            rc = self.lib.vc2hip_create(device, C.byref(h))
        else:
            rc = self.lib.vc2hip_create_on_stream(device, C.c_void_p(stream), C.byref(h))
        if rc != 0:
            raise Vc2HipError(rc, "vc2hip_create failed: " + self.lib.vc2hip_error_string(rc).decode()
                              + " (a MI355X is required; there is no CPU fallback)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.vc2hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise Vc2HipError(rc, self.lib.vc2hip_last_error(self.h).decode())

    # ---- host helpers
    def padded_size(self, n, depth):
        return self.lib.vc2hip_padded_size(n, depth)

    def quant_matrix(self, kernel, depth):
        out = np.zeros(3 * depth + 1, np.int32)
        rc = self.lib.vc2hip_quant_matrix(kernel, depth, out)
        if rc:
            raise Vc2HipError(rc, "invalid wavelet kernel")
        return out

    def slice_bytes(self, ys, xs, total, scalar):
        out = np.empty((ys, xs), np.int32)
        self.lib.vc2hip_slice_bytes(ys, xs, total, scalar, out)
        return out

    # ---- fine-grained
    def dwt_forward(self, plane, kernel, depth):
        plane = np.ascontiguousarray(plane, np.int32)
        h, w = plane.shape
        out = np.empty((self.padded_size(h, depth), self.padded_size(w, depth)), np.int32)
        self._chk(self.lib.vc2hip_dwt_forward(self.h, plane, h, w, kernel, depth, out))
        return out

    def dwt_inverse(self, coef, kernel, depth, shape=None):
        coef = np.ascontiguousarray(coef, np.int32)
        ph, pw = coef.shape
        h, w = shape if shape is not None else (ph, pw)
        out = np.empty((h, w), np.int32)
        self._chk(self.lib.vc2hip_dwt_inverse(self.h, coef, ph, pw, kernel, depth, out, h, w))
        return out

    def _q(self, fn, plane, depth, qidx, qm):
        plane = np.ascontiguousarray(plane, np.int32)
        qidx = np.ascontiguousarray(qidx, np.int32)
        out = np.empty_like(plane)
        self._chk(fn(self.h, plane, plane.shape[0], plane.shape[1], depth, qidx, qidx.shape[0],
                     qidx.shape[1], np.ascontiguousarray(qm, np.int32), out))
        return out

    def quantise_np(self, plane, depth, qidx, qm):
        return self._q(self.lib.vc2hip_quantise_np, plane, depth, qidx, qm)

    def dequantise_np(self, plane, depth, qidx, qm):
        return self._q(self.lib.vc2hip_dequantise_np, plane, depth, qidx, qm)

    def dequantise_ld(self, plane, depth, qidx, qm):
        return self._q(self.lib.vc2hip_dequantise_ld, plane, depth, qidx, qm)

    def quantise_ld(self, plane, depth, qidx, qm):
        return self._q(self.lib.vc2hip_quantise_ld, plane, depth, qidx, qm)

    @staticmethod
    def geom(y, u, depth, ys, xs):
        return Geom(y.shape[0], y.shape[1], u.shape[0], u.shape[1], depth, ys, xs)

    def hq_pack(self, y, u, v, depth, qidx, prefix=0, scalar=1, cbr=None):
        y, u, v = (np.ascontiguousarray(a, np.int32) for a in (y, u, v))
        qidx = np.ascontiguousarray(qidx, np.int32)
        g = self.geom(y, u, depth, qidx.shape[0], qidx.shape[1])
        cap = qidx.size * (prefix + 4 + 3 * 255 * scalar) + 64
        cbr_p = None
        if cbr is not None:
            cbr = np.ascontiguousarray(cbr, np.int32)
            cap = int(cbr.sum()) + qidx.size * prefix + 64
            cbr_p = cbr.ctypes.data_as(C.c_void_p)
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        self._chk(self.lib.vc2hip_hq_pack(self.h, y, u, v, C.byref(g), qidx, prefix, scalar, cbr_p, out,
                                          cap, C.byref(n)))
        return out[:n.value].copy()

    def hq_unpack(self, data, lshape, cshape, depth, ys, xs, prefix=0, scalar=1):
        y = np.zeros(lshape, np.int32)
        u = np.zeros(cshape, np.int32)
        v = np.zeros(cshape, np.int32)
        q = np.zeros((ys, xs), np.int32)
        g = self.geom(y, u, depth, ys, xs)
        used = C.c_size_t()
        data = np.ascontiguousarray(data, np.uint8)
        self._chk(self.lib.vc2hip_hq_unpack(self.h, data, data.size, C.byref(g), prefix, scalar, y, u, v, q,
                                            C.byref(used)))
        return y, u, v, q, used.value

    def ld_unpack(self, data, lshape, cshape, depth, slice_bytes):
        ys, xs = slice_bytes.shape
        y = np.zeros(lshape, np.int32)
        u = np.zeros(cshape, np.int32)
        v = np.zeros(cshape, np.int32)
        q = np.zeros((ys, xs), np.int32)
        g = self.geom(y, u, depth, ys, xs)
        used = C.c_size_t()
        data = np.ascontiguousarray(data, np.uint8)
        self._chk(self.lib.vc2hip_ld_unpack(self.h, data, data.size, C.byref(g),
                                            np.ascontiguousarray(slice_bytes, np.int32), y, u, v, q,
                                            C.byref(used)))
        return y, u, v, q, used.value

    def cbr_qindices(self, y, u, v, depth, qm, slice_bytes, scalar):
        y, u, v = (np.ascontiguousarray(a, np.int32) for a in (y, u, v))
        ys, xs = slice_bytes.shape
        g = self.geom(y, u, depth, ys, xs)
        q = np.zeros((ys, xs), np.int32)
        self._chk(self.lib.vc2hip_cbr_qindices(self.h, y, u, v, C.byref(g), np.ascontiguousarray(qm, np.int32),
                                               np.ascontiguousarray(slice_bytes, np.int32), scalar, q))
        return q

    def ld_qindices(self, y, u, v, depth, qm, slice_bytes):
        y, u, v = (np.ascontiguousarray(a, np.int32) for a in (y, u, v))
        ys, xs = slice_bytes.shape
        g = self.geom(y, u, depth, ys, xs)
        q = np.zeros((ys, xs), np.int32)
        self._chk(self.lib.vc2hip_ld_qindices(self.h, y, u, v, C.byref(g), np.ascontiguousarray(qm, np.int32),
                                              np.ascontiguousarray(slice_bytes, np.int32), q))
        return q

    def ld_pack(self, y, u, v, depth, qidx, slice_bytes):
        y, u, v = (np.ascontiguousarray(a, np.int32) for a in (y, u, v))
        qidx = np.ascontiguousarray(qidx, np.int32)
        g = self.geom(y, u, depth, qidx.shape[0], qidx.shape[1])
        cap = int(slice_bytes.sum()) + 64
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        self._chk(self.lib.vc2hip_ld_pack(self.h, y, u, v, C.byref(g), qidx,
                                          np.ascontiguousarray(slice_bytes, np.int32), out, cap, C.byref(n)))
        return out[:n.value].copy()

    # ---- fused pictures (host buffers)
    def raw_picture_bytes(self, fmt):
        return self.lib.vc2hip_raw_picture_bytes(C.byref(fmt))

    def max_payload_bytes(self, fmt, cp):
        return self.lib.vc2hip_max_payload_bytes(C.byref(fmt), C.byref(cp))

    def encode_picture_hq(self, raw, fmt, cp):
        raw = np.frombuffer(raw, np.uint8)
        cap = self.max_payload_bytes(fmt, cp) + 64
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        qidx = np.zeros((cp.y_slices, cp.x_slices), np.int32)
        fn = self.lib.vc2hip_encode_picture_ld if cp.mode == MODES["LD"] else self.lib.vc2hip_encode_picture_hq
        self._chk(fn(self.h, raw, C.byref(fmt), C.byref(cp), out, cap, C.byref(n), qidx.ctypes.data_as(C.c_void_p)))
        return out[:n.value].tobytes(), qidx

    def decode_picture(self, payload, fmt, cp):
        payload = np.frombuffer(payload, np.uint8)
        out = np.empty(self.raw_picture_bytes(fmt), np.uint8)
        fn = self.lib.vc2hip_decode_picture_ld if cp.mode == MODES["LD"] else self.lib.vc2hip_decode_picture_hq
        self._chk(fn(self.h, payload, payload.size, C.byref(fmt), C.byref(cp), out))
        return out.tobytes()

    # ---- device-resident batches (pointers are raw device addresses, e.g. tensor.data_ptr())
    def encode_batch_dev(self, d_raw, n, fmt, cp, d_payload, stride, d_lens):
        self._chk(self.lib.vc2hip_encode_batch_dev(self.h, d_raw, n, C.byref(fmt), C.byref(cp), d_payload,
                                                   stride, d_lens))

    def decode_batch_dev(self, d_payload, stride, d_lens, n, fmt, cp, d_raw_out):
        self._chk(self.lib.vc2hip_decode_batch_dev(self.h, d_payload, stride, d_lens, n, C.byref(fmt),
                                                   C.byref(cp), d_raw_out))

    def encode_pictures_pipelined(self, raws, fmt, cp):
        """pictures in host memory through the pipelined calls (pinned staging, two in flight); returns the payloads"""
        rb = self.raw_picture_bytes(fmt)
        cap = self.max_payload_bytes(fmt, cp) + 64
        n_slots = 2
        ins = [self.lib.vc2hip_host_alloc(rb + 64) for _ in range(n_slots)]
        outs = [self.lib.vc2hip_host_alloc(cap) for _ in range(n_slots)]
        res, open_ = [], []
        try:
            def finish():
                slot, ticket = open_.pop(0)
                n = C.c_size_t()
                self._chk(self.lib.vc2hip_encode_picture_end(self.h, ticket, C.byref(n)))
                res.append(C.string_at(outs[slot], n.value))
            for k, raw in enumerate(raws):
                if len(open_) == n_slots:
                    finish()
                slot = k % n_slots
                C.memmove(ins[slot], raw, rb)
                t = C.c_int()
                self._chk(self.lib.vc2hip_encode_picture_begin(self.h, ins[slot], C.byref(fmt), C.byref(cp), outs[slot], cap, None,
                                                               C.byref(t)))
                open_.append((slot, t.value))
            while open_:
                finish()
        finally:
            for p in ins + outs:
                self.lib.vc2hip_host_free(p)
        return res

    def decode_pictures_pipelined(self, payloads, fmt, cp):
        rb = self.raw_picture_bytes(fmt)
        cap = max(len(p) for p in payloads) + 64
        n_slots = 2
        ins = [self.lib.vc2hip_host_alloc(cap) for _ in range(n_slots)]
        outs = [self.lib.vc2hip_host_alloc(rb + 64) for _ in range(n_slots)]
        res, open_ = [], []
        try:
            def finish():
                slot, ticket = open_.pop(0)
                self._chk(self.lib.vc2hip_decode_picture_end(self.h, ticket))
                res.append(C.string_at(outs[slot], rb))
            for k, pay in enumerate(payloads):
                if len(open_) == n_slots:
                    finish()
                slot = k % n_slots
                C.memmove(ins[slot], pay, len(pay))
                t = C.c_int()
                self._chk(self.lib.vc2hip_decode_picture_begin(self.h, ins[slot], len(pay), C.byref(fmt), C.byref(cp), outs[slot],
                                                               C.byref(t)))
                open_.append((slot, t.value))
            while open_:
                finish()
        finally:
            for p in ins + outs:
                self.lib.vc2hip_host_free(p)
        return res

    def set_streams(self, k):
        self._chk(self.lib.vc2hip_set_streams(self.h, k))

    def sync(self):
        self._chk(self.lib.vc2hip_sync(self.h))

    # ---- profiling
    def profile_enable(self, on=True):
        self.lib.vc2hip_profile_enable(self.h, 1 if on else 0)

    def profile_only(self, name=None):
        self.lib.vc2hip_profile_only(self.h, name.encode() if name else None)

    def profile_reset(self):
        self.lib.vc2hip_profile_reset(self.h)

    def band_plane_bits(self):
        """0 / 16 / 8: the band planes of this context's most recent HQ decode call (vc2hip_band_plane_bits)"""
        return int(self.lib.vc2hip_band_plane_bits(self.h))

    def profile(self):
        out = {}
        for i in range(self.lib.vc2hip_profile_count(self.h)):
            name = C.c_char_p()
            n = C.c_int()
            ms = C.c_double()
            self.lib.vc2hip_profile_get(self.h, i, C.byref(name), C.byref(n), C.byref(ms))
            out[name.value.decode()] = (n.value, ms.value)
        return out
