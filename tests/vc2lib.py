"""ctypes bindings used by the tests: the CPU oracle (oracle/libvc2oracle.so, test
infrastructure) and the product library (libvc2hip.so, through its C-ABI only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libvc2oracle.so")
REF_VLC_SO = os.path.join(ORACLE_DIR, "_ref", "libvc2ref_vlc.so")

KERNELS = {"DD97": 0, "LeGall": 1, "DD137": 2, "Haar0": 3, "Haar1": 4, "Fidelity": 5, "Daub97": 6}
CF = {"444": 0, "422": 1, "420": 2}
MODES = {"HQ_ConstQ": 0, "HQ_CBR": 1, "LD": 2}

i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class Geom(C.Structure):
    _fields_ = [("luma_h", C.c_int), ("luma_w", C.c_int), ("chroma_h", C.c_int),
                ("chroma_w", C.c_int), ("depth", C.c_int), ("y_slices", C.c_int),
                ("x_slices", C.c_int)]


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("cf", C.c_int),
                ("bit_depth", C.c_int), ("word_bytes", C.c_int), ("kernel", C.c_int),
                ("depth", C.c_int), ("y_size", C.c_int), ("x_size", C.c_int),
                ("mode", C.c_int), ("q_index", C.c_int), ("compressed_bytes", C.c_int),
                ("scalar", C.c_int), ("prefix", C.c_int), ("frame_rate", C.c_int),
                ("interlaced", C.c_int), ("bottom_field_first", C.c_int), ("fragment_length", C.c_int)]


def make_params(width, height, cf, bits, kernel, depth, u, a, mode="HQ_ConstQ", q=0, s=0,
                scalar=1, prefix=0, word_bytes=2, frame_rate=3, interlaced=False, bottom_field_first=False,
                fragment_length=0):
    return Params(width, height, CF[cf], bits, word_bytes, KERNELS[kernel], depth, u, a,
                  MODES[mode], q, s, scalar, prefix, frame_rate, int(interlaced), int(bottom_field_first),
                  fragment_length)


class OracleError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.vc2o_last_error.restype = C.c_char_p
        lib.vc2o_padded_size.restype = C.c_int
        lib.vc2o_slice_size_is_valid.restype = C.c_int
        lib.vc2o_ingest.argtypes = [u8p, C.c_int, C.c_int, C.c_size_t, i32p]
        lib.vc2o_ingest.restype = None
        lib.vc2o_clip_emit.argtypes = [i32p, C.c_size_t, C.c_int, C.c_int, u8p]
        lib.vc2o_clip_emit.restype = None
        lib.vc2o_pad.argtypes = [i32p, C.c_int, C.c_int, i32p, C.c_int, C.c_int]
        lib.vc2o_pad.restype = None
        lib.vc2o_dwt_forward.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.vc2o_dwt_inverse.argtypes = [i32p, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.vc2o_quant_matrix.argtypes = [C.c_int, C.c_int, i32p]
        for f in (lib.vc2o_quant, lib.vc2o_scale):
            f.argtypes = [C.c_int32, C.c_int, C.POINTER(C.c_int32)]
        lib.vc2o_quant_factor.argtypes = [C.c_int, C.POINTER(C.c_int32)]
        for f in (lib.vc2o_quantise_np, lib.vc2o_dequantise_np, lib.vc2o_quantise_ld,
                  lib.vc2o_dequantise_ld):
            f.argtypes = [i32p, C.c_int, C.c_int, C.c_int, i32p, C.c_int, C.c_int, i32p, i32p]
        lib.vc2o_slice_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, i32p]
        lib.vc2o_hq_pack.argtypes = [i32p, i32p, i32p, C.POINTER(Geom), i32p, C.c_int, C.c_int,
                                     C.c_void_p, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        lib.vc2o_hq_unpack.argtypes = [u8p, C.c_size_t, C.POINTER(Geom), C.c_int, C.c_int, i32p,
                                       i32p, i32p, i32p, C.POINTER(C.c_size_t)]
        lib.vc2o_cbr_qindices.argtypes = [i32p, i32p, i32p, C.POINTER(Geom), i32p, i32p, C.c_int,
                                          i32p]
        lib.vc2o_ld_pack.argtypes = [i32p, i32p, i32p, C.POINTER(Geom), i32p, i32p, u8p,
                                     C.c_size_t, C.POINTER(C.c_size_t)]
        lib.vc2o_ld_unpack.argtypes = [u8p, C.c_size_t, C.POINTER(Geom), i32p, i32p, i32p, i32p,
                                       i32p, C.POINTER(C.c_size_t)]
        lib.vc2o_ld_qindices.argtypes = [i32p, i32p, i32p, C.POINTER(Geom), i32p, i32p, i32p]
        lib.vc2o_encode_stream.argtypes = [C.POINTER(Params), u8p, C.c_int, u8p, C.c_size_t,
                                           C.POINTER(C.c_size_t)]
        lib.vc2o_decode_stream.argtypes = [C.POINTER(Params), u8p, C.c_size_t, u8p, C.c_size_t,
                                           C.POINTER(C.c_int)]

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(rc, self.lib.vc2o_last_error().decode())

    # --- scalars
    def quant(self, v, aq):
        out = C.c_int32()
        self._chk(self.lib.vc2o_quant(v, aq, C.byref(out)))
        return out.value

    def scale(self, v, aq):
        out = C.c_int32()
        self._chk(self.lib.vc2o_scale(v, aq, C.byref(out)))
        return out.value

    def quant_factor(self, q):
        out = C.c_int32()
        self._chk(self.lib.vc2o_quant_factor(q, C.byref(out)))
        return out.value

    def padded_size(self, n, depth):
        return self.lib.vc2o_padded_size(n, depth)

    def quant_matrix(self, kernel, depth):
        out = np.zeros(3 * depth + 1, np.int32)
        self._chk(self.lib.vc2o_quant_matrix(kernel, depth, out))
        return out

    # --- planes
    def ingest(self, raw, word_bytes, bits, shape):
        out = np.empty(shape, np.int32)
        self.lib.vc2o_ingest(np.frombuffer(raw, np.uint8), word_bytes, bits, out.size, out)
        return out

    def clip_emit(self, plane, word_bytes, bits):
        plane = np.ascontiguousarray(plane, np.int32)
        out = np.empty(plane.size * word_bytes, np.uint8)
        self.lib.vc2o_clip_emit(plane, plane.size, word_bytes, bits, out)
        return out

    def pad(self, plane, depth):
        h, w = plane.shape
        ph, pw = self.padded_size(h, depth), self.padded_size(w, depth)
        out = np.empty((ph, pw), np.int32)
        self.lib.vc2o_pad(np.ascontiguousarray(plane, np.int32), h, w, out, ph, pw)
        return out

    def dwt_forward(self, plane, kernel, depth):
        p = self.pad(plane, depth)
        self._chk(self.lib.vc2o_dwt_forward(p, p.shape[0], p.shape[1], kernel, depth))
        return p

    def dwt_inverse(self, coef, kernel, depth, shape=None):
        p = np.array(coef, np.int32, order="C")
        self._chk(self.lib.vc2o_dwt_inverse(p, p.shape[0], p.shape[1], kernel, depth))
        if shape is not None:
            p = np.ascontiguousarray(p[:shape[0], :shape[1]])
        return p

    def _q(self, fn, plane, depth, qidx, qm):
        plane = np.ascontiguousarray(plane, np.int32)
        qidx = np.ascontiguousarray(qidx, np.int32)
        out = np.empty_like(plane)
        self._chk(fn(plane, plane.shape[0], plane.shape[1], depth, qidx, qidx.shape[0],
                     qidx.shape[1], np.ascontiguousarray(qm, np.int32), out))
        return out

    def quantise_np(self, plane, depth, qidx, qm):
        return self._q(self.lib.vc2o_quantise_np, plane, depth, qidx, qm)

    def dequantise_np(self, plane, depth, qidx, qm):
        return self._q(self.lib.vc2o_dequantise_np, plane, depth, qidx, qm)

    def quantise_ld(self, plane, depth, qidx, qm):
        return self._q(self.lib.vc2o_quantise_ld, plane, depth, qidx, qm)

    def dequantise_ld(self, plane, depth, qidx, qm):
        return self._q(self.lib.vc2o_dequantise_ld, plane, depth, qidx, qm)

    def slice_bytes(self, ys, xs, total, scalar):
        out = np.empty((ys, xs), np.int32)
        self._chk(self.lib.vc2o_slice_bytes(ys, xs, total, scalar, out))
        return out

    @staticmethod
    def geom(y, u, depth, ys, xs):
        return Geom(y.shape[0], y.shape[1], u.shape[0], u.shape[1], depth, ys, xs)

    def hq_pack(self, y, u, v, depth, qidx, prefix=0, scalar=1, cbr=None):
        g = self.geom(y, u, depth, qidx.shape[0], qidx.shape[1])
        cap = (y.size + u.size + v.size) * 4 + qidx.size * (4 + prefix) + 1024
        if cbr is not None:
            cap = max(cap, int(cbr.sum()) + qidx.size * prefix + 1024)
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        cbr_p = None if cbr is None else np.ascontiguousarray(cbr, np.int32).ctypes.data_as(C.c_void_p)
        self._cbr_keep = cbr
        self._chk(self.lib.vc2o_hq_pack(y, u, v, C.byref(g), np.ascontiguousarray(qidx, np.int32),
                                        prefix, scalar, cbr_p, out, cap, C.byref(n)))
        return out[:n.value].copy()

    def hq_unpack(self, data, lshape, cshape, depth, ys, xs, prefix=0, scalar=1):
        y = np.zeros(lshape, np.int32)
        u = np.zeros(cshape, np.int32)
        v = np.zeros(cshape, np.int32)
        q = np.zeros((ys, xs), np.int32)
        g = self.geom(y, u, depth, ys, xs)
        used = C.c_size_t()
        data = np.ascontiguousarray(data, np.uint8)
        self._chk(self.lib.vc2o_hq_unpack(data, data.size, C.byref(g), prefix, scalar, y, u, v, q,
                                          C.byref(used)))
        return y, u, v, q, used.value

    def cbr_qindices(self, y, u, v, depth, qm, slice_bytes, scalar):
        ys, xs = slice_bytes.shape
        g = self.geom(y, u, depth, ys, xs)
        q = np.zeros((ys, xs), np.int32)
        self._chk(self.lib.vc2o_cbr_qindices(y, u, v, C.byref(g), np.ascontiguousarray(qm, np.int32),
                                             np.ascontiguousarray(slice_bytes, np.int32), scalar, q))
        return q

    def ld_qindices(self, y, u, v, depth, qm, slice_bytes):
        ys, xs = slice_bytes.shape
        g = self.geom(y, u, depth, ys, xs)
        q = np.zeros((ys, xs), np.int32)
        self._chk(self.lib.vc2o_ld_qindices(y, u, v, C.byref(g), np.ascontiguousarray(qm, np.int32),
                                            np.ascontiguousarray(slice_bytes, np.int32), q))
        return q

    def ld_pack(self, y, u, v, depth, qidx, slice_bytes):
        g = self.geom(y, u, depth, qidx.shape[0], qidx.shape[1])
        cap = int(slice_bytes.sum()) + 1024
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        self._chk(self.lib.vc2o_ld_pack(y, u, v, C.byref(g), np.ascontiguousarray(qidx, np.int32),
                                        np.ascontiguousarray(slice_bytes, np.int32), out, cap,
                                        C.byref(n)))
        return out[:n.value].copy()

    def ld_unpack(self, data, lshape, cshape, depth, slice_bytes):
        ys, xs = slice_bytes.shape
        y = np.zeros(lshape, np.int32)
        u = np.zeros(cshape, np.int32)
        v = np.zeros(cshape, np.int32)
        q = np.zeros((ys, xs), np.int32)
        g = self.geom(y, u, depth, ys, xs)
        used = C.c_size_t()
        data = np.ascontiguousarray(data, np.uint8)
        self._chk(self.lib.vc2o_ld_unpack(data, data.size, C.byref(g),
                                          np.ascontiguousarray(slice_bytes, np.int32), y, u, v, q,
                                          C.byref(used)))
        return y, u, v, q, used.value

    # --- whole files
    def encode_stream(self, params, raw, n_frames):
        raw = np.frombuffer(raw, np.uint8)
        cap = raw.size * 3 + 4096
        out = np.empty(cap, np.uint8)
        n = C.c_size_t()
        self._chk(self.lib.vc2o_encode_stream(C.byref(params), raw, n_frames, out, cap, C.byref(n)))
        return out[:n.value].tobytes()

    def decode_stream(self, params, stream, max_frames):
        ch = params.height // 2 if params.cf == 2 else params.height
        cw = params.width if params.cf == 0 else params.width // 2
        fb = (params.width * params.height + 2 * ch * cw) * params.word_bytes
        out = np.empty(fb * max_frames, np.uint8)
        n = C.c_int()
        s = np.frombuffer(stream, np.uint8)
        self._chk(self.lib.vc2o_decode_stream(C.byref(params), s, s.size, out, out.size, C.byref(n)))
        return out[:fb * n.value].tobytes(), n.value


def load_oracle():
    if not os.path.exists(ORACLE_SO) or (
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "vc2_oracle.c"))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libvc2oracle.so"], stdout=subprocess.DEVNULL)
    return Oracle(C.CDLL(ORACLE_SO))


def load_ref_vlc():
    """The reference's own VLC.cpp (oracle/_ref), or None if it was never built."""
    if not os.path.exists(REF_VLC_SO):
        if os.path.exists("/root/reference/src/Library/src/VLC.cpp"):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
        else:
            return None
    lib = C.CDLL(REF_VLC_SO)
    lib.ref_svlc_write_bounded.argtypes = [i32p, C.c_int, C.c_long, u8p, C.c_long]
    lib.ref_svlc_write_bounded.restype = C.c_long
    lib.ref_svlc_read_bounded.argtypes = [u8p, C.c_long, C.c_long, C.c_int, i32p]
    lib.ref_svlc_read_bounded.restype = C.c_long
    lib.ref_svlc_numbits.argtypes = [C.c_int32]
    lib.ref_svlc_code.argtypes = [C.c_int32]
    lib.ref_svlc_code.restype = C.c_uint
    return lib


def load_hip():
    from vc2hip_py import Vc2Hip  # host-side mirror; raises if the library is missing
    return Vc2Hip()
