/*
 * vc2hip.h -- C-ABI of libvc2hip.so: the MI355X (gfx950) VC-2 HQ/LD hot path.
 *
 * bbc/vc2-reference has no FFI; its boundary for this path is the set of
 * `Library` free functions the EncodeStream / DecodeStream mains call.  Every
 * entry point below names the reference interface it replaces (file:line under
 * /root/reference).  Conventions: plain pointers and sizes, caller owns every
 * buffer, no exceptions cross the ABI -- each call returns VC2HIP_OK or a
 * negative VC2HIP_E* code whose text (identical to the reference's exception
 * what() string where one exists) is available from vc2hip_last_error().
 *
 * int32 planes are row-major with stride == width, exactly the element order of
 * the reference's Array2D (src/Library/Arrays.h:17-50).
 */
#ifndef VC2HIP_H
#define VC2HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* libvc2hip.so is built with hidden visibility: the declarations below are its whole export list */
#endif

typedef struct vc2hip_ctx vc2hip_ctx; /* one per GPU: stream, device scratch, error state */

/* WaveletKernel, src/Library/WaveletTransform.h:26 (== wavelet_index in the stream) */
enum { VC2HIP_DD97 = 0, VC2HIP_LEGALL = 1, VC2HIP_DD137 = 2, VC2HIP_HAAR0 = 3,
       VC2HIP_HAAR1 = 4, VC2HIP_FIDELITY = 5, VC2HIP_DAUB97 = 6 };
/* ColourFormat, src/Library/Picture.h:17 */
enum { VC2HIP_CF444 = 0, VC2HIP_CF422 = 1, VC2HIP_CF420 = 2 };
/* Mode, src/EncodeStream/EncodeParams.h */
enum { VC2HIP_HQ_CONSTQ = 0, VC2HIP_HQ_CBR = 1, VC2HIP_LD = 2 };

enum {
  VC2HIP_OK = 0,
  VC2HIP_EINVAL = -1,       /* bad argument / "invalid wavelet kernel" (WaveletTransform.cpp:258) */
  VC2HIP_EQINDEX = -2,      /* "quantization index exceeds maximum implemented value." (Quantisation.cpp:61) */
  VC2HIP_ESCALAR = -3,      /* "Slice scalar is too small, consider using a larger slice scalar." (Slices.cpp:116) */
  VC2HIP_ECBR_TOOBIG = -4,  /* "SliceIO, HQ CBR mode: Too many bytes for the slice" (Slices.cpp:357) */
  VC2HIP_ECBR_LEN = -5,     /* "Slice component length exceeds 1 byte when divided by slice size scalar. ..." (Slices.cpp:365) */
  VC2HIP_ECBR_WRONG = -6,   /* "SliceIO, HQ CBR mode: Wrong number of bytes for a slice" (Slices.cpp:446) */
  VC2HIP_EBOUNDED = -7,     /* "Attempt to write beyond end of bounded write" (VLC.cpp:154) */
  VC2HIP_ELD_TOOBIG = -8,   /* "SliceIO, LD mode: Too many bytes for the U and V slices" (Slices.cpp:210) */
  VC2HIP_ECAP = -9,         /* caller's output buffer too small */
  VC2HIP_ESTREAM = -10,     /* truncated / malformed slice data */
  VC2HIP_ECODE32 = -11,     /* |quantised coefficient| > 65534: outside the reference's 32-bit VLC domain (VLC.h:27) */
  VC2HIP_EHIP = -100        /* HIP runtime failure (no device, out of memory, ...) */
};

/* ---------------------------------------------------------------------------------------------
 * context
 * ------------------------------------------------------------------------------------------- */
int vc2hip_create(int device, vc2hip_ctx **out);
/* The same with switches that select, for tests and A/B measurements, between two correct paths of the library (each
 * flag takes the slower / more general one; 0 = vc2hip_create).  The release library reads no environment variable. */
#define VC2HIP_FLAG_STORE32         0x001u /* int32 coefficient elements instead of 16-bit + escape */
#define VC2HIP_FLAG_NO_STREAM       0x002u /* LDS tile kernels instead of the streaming transform kernels */
#define VC2HIP_FLAG_NO_PAIR         0x004u /* one launch per transform level (no two-level kernels) */
#define VC2HIP_FLAG_NO_BANDPLANES   0x008u /* the decoder keeps every band in the slice records */
#define VC2HIP_FLAG_NO_HEADS        0x010u /* no side-by-side record heads for the deep levels */
#define VC2HIP_FLAG_NO_CBR_INDEX    0x020u /* HQ_CBR decode always through the general slice index */
#define VC2HIP_FLAG_GENERIC_DWT     0x040u /* generic transform kernels only */
#define VC2HIP_FLAG_SINGLE_PASS_VBR 0x080u /* VBR packing with decoupled look-back instead of slots + scan + compaction, whatever the batch (default: from 112 pictures per call on, where the round-4 slice coder applies) */
#define VC2HIP_FLAG_CBR_GENERAL     0x100u /* HQ_CBR quantiser search without the register kernels */
#define VC2HIP_FLAG_LD_DIAGONALS    0x200u /* LD index search: one launch per slice anti-diagonal instead of one launch */
#define VC2HIP_FLAG_PLANES8_ALWAYS  0x400u /* decoder: one byte per band-plane coefficient from the first picture (default: once a batch has shown small coefficients) */
#define VC2HIP_FLAG_PLANES8_NEVER   0x800u /* decoder: 16-bit band planes only */
#define VC2HIP_FLAG_TWO_PASS_VBR    0x1000u /* VBR packing through slots + scan + compaction also where the one-pass slice coder is the default */
int vc2hip_create_with_flags(int device, unsigned flags, vc2hip_ctx **out);
/* same, but launch on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) */
int vc2hip_create_on_stream(int device, void *hip_stream, vc2hip_ctx **out);
void vc2hip_destroy(vc2hip_ctx *ctx);
const char *vc2hip_last_error(const vc2hip_ctx *ctx);
const char *vc2hip_error_string(int code);
int vc2hip_sync(vc2hip_ctx *ctx); /* wait for the stream, then surface device-side error flags */

/* ---------------------------------------------------------------------------------------------
 * host-side helpers (pure host arithmetic, kept in the library so every binding agrees)
 * ------------------------------------------------------------------------------------------- */
/* paddedSize, WaveletTransform.cpp:74-77 */
int vc2hip_padded_size(int size, int depth);
/* sliceSizeIsValid, WaveletTransform.cpp:116-136: number of slices, or 0 */
int vc2hip_slice_size_is_valid(int depth, int len_luma, int len_chroma, int n_size);
/* quantMatrix, WaveletTransform.cpp:345-423; out has 3*depth+1 entries */
int vc2hip_quant_matrix(int kernel, int depth, int32_t *out);
/* slice_bytes(ySlices,xSlices,totalBytes,scalar), Slices.cpp:28-49; out is ySlices*xSlices */
int vc2hip_slice_bytes(int y_slices, int x_slices, int total_bytes, int scalar, int32_t *out);

/* ---------------------------------------------------------------------------------------------
 * fine-grained entry points, host int32 planes in / out: 1:1 with Library functions
 * ------------------------------------------------------------------------------------------- */
/* waveletTransform(const Array2D&, kernel, depth), WaveletTransform.cpp:262-281 (includes waveletPad
 * :79-94).  in: h x w.  out: paddedSize(h) x paddedSize(w), in-place interleaved subband order. */
int vc2hip_dwt_forward(vc2hip_ctx *ctx, const int32_t *in, int h, int w, int kernel, int depth,
                       int32_t *out);
/* inverseWaveletTransform(const Array2D&, kernel, depth, shape), WaveletTransform.cpp:321-342.
 * in: ph x pw (padded).  out: h x w (cropped top-left). */
int vc2hip_dwt_inverse(vc2hip_ctx *ctx, const int32_t *in, int ph, int pw, int kernel, int depth,
                       int32_t *out, int h, int w);
/* quantise_transform_np(const Array2D&, const Array2D& qIndices, const Array1D& qMatrix),
 * Quantisation.cpp:479-489; qidx is ys x xs */
int vc2hip_quantise_np(vc2hip_ctx *ctx, const int32_t *coef, int ph, int pw, int depth,
                       const int32_t *qidx, int ys, int xs, const int32_t *qmatrix, int32_t *out);
/* inverse_quantise_transform_np, Quantisation.cpp:534-544 */
int vc2hip_dequantise_np(vc2hip_ctx *ctx, const int32_t *q, int ph, int pw, int depth,
                         const int32_t *qidx, int ys, int xs, const int32_t *qmatrix, int32_t *out);
/* inverse_quantise_transform (LD, DC-predicted LL band), Quantisation.cpp:369-379, :287-306 */
int vc2hip_dequantise_ld(vc2hip_ctx *ctx, const int32_t *q, int ph, int pw, int depth,
                         const int32_t *qidx, int ys, int xs, const int32_t *qmatrix, int32_t *out);

/* quantise_transform (LD, DC-predicted LL band), Quantisation.cpp:358-367 over :213-234 */
int vc2hip_quantise_ld(vc2hip_ctx *ctx, const int32_t *coef, int ph, int pw, int depth,
                       const int32_t *qidx, int ys, int xs, const int32_t *qmatrix, int32_t *out);

/* geometry of the three quantised planes handed to the slice coders */
typedef struct {
  int luma_h, luma_w;     /* padded */
  int chroma_h, chroma_w; /* padded */
  int depth;
  int y_slices, x_slices;
} vc2hip_geom;

/* operator<<(ostream&, const Slices&) under sliceio::highQualityVBR(prefix,scalar) /
 * highQualityCBR(bytes,prefix,scalar), Slices.cpp:645-660 over :469-533 / :305-382.
 * y,u,v: QUANTISED planes.  cbr_slice_bytes == NULL selects VBR. */
int vc2hip_hq_pack(vc2hip_ctx *ctx, const int32_t *y, const int32_t *u, const int32_t *v,
                   const vc2hip_geom *g, const int32_t *qidx, int prefix, int scalar,
                   const int32_t *cbr_slice_bytes, uint8_t *out, size_t cap, size_t *out_len);
/* operator>>(istream&, Slices&) under highQualityVBR, Slices.cpp:662-694 over :535-612 */
int vc2hip_hq_unpack(vc2hip_ctx *ctx, const uint8_t *in, size_t len, const vc2hip_geom *g,
                     int prefix, int scalar, int32_t *y, int32_t *u, int32_t *v, int32_t *qidx,
                     size_t *consumed);
/* operator>>(istream&, Slices&) under sliceio::lowDelay(bytes), Slices.cpp:246-303.  Every slice is read at its own
 * offset (the running sum of slice_bytes): a corrupt slice whose luma length field exceeds the slice does not move the
 * slices behind it as the reference's stream reader would -- vc2hip_decode_picture_ld / vc2hip_decode_batch_dev follow
 * the reference there. */
int vc2hip_ld_unpack(vc2hip_ctx *ctx, const uint8_t *in, size_t len, const vc2hip_geom *g,
                     const int32_t *slice_bytes, int32_t *y, int32_t *u, int32_t *v,
                     int32_t *qidx, size_t *consumed);
/* operator<<(ostream&, const Slices&) under sliceio::lowDelay(bytes), Slices.cpp:645-660 over
 * :195-244.  y,u,v: QUANTISED planes, LL band as DC-prediction residuals (vc2hip_quantise_ld). */
int vc2hip_ld_pack(vc2hip_ctx *ctx, const int32_t *y, const int32_t *u, const int32_t *v,
                   const vc2hip_geom *g, const int32_t *qidx, const int32_t *slice_bytes,
                   uint8_t *out, size_t cap, size_t *out_len);
/* quantIndicesLD(coefficients, qMatrix, sliceBytes), EncodeStream.cpp:141-245 (per-slice search with
 * the DC-prediction state machine of SliceQuantiserRef).  y,u,v: TRANSFORM planes. */
int vc2hip_ld_qindices(vc2hip_ctx *ctx, const int32_t *y, const int32_t *u, const int32_t *v,
                       const vc2hip_geom *g, const int32_t *qmatrix, const int32_t *slice_bytes,
                       int32_t *qidx);
/* quantIndicesCBR(coefficients, qMatrix, sliceBytes, scalar), EncodeStream.cpp:73-125.
 * y,u,v: TRANSFORM (unquantised) planes. */
int vc2hip_cbr_qindices(vc2hip_ctx *ctx, const int32_t *y, const int32_t *u, const int32_t *v,
                        const vc2hip_geom *g, const int32_t *qmatrix, const int32_t *slice_bytes,
                        int scalar, int32_t *qidx);

/* ---------------------------------------------------------------------------------------------
 * fused picture path: the per-picture body of EncodeStream.cpp:482-647 and
 * DecodeStream.cpp:451-613 / :289-450 (sample words in, slice payload out, and back)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int width, height; /* luma picture size (unpadded)                       */
  int chroma_format; /* VC2HIP_CF*                                          */
  int bit_depth;     /* luma depth (EncodeStream -l); the decoder's one depth (sequence header, DecodeStream.cpp:268) */
  int word_bytes;    /* bytes per sample in the raw planar file (-n), 1..4  */
  int chroma_bit_depth; /* encoder input only: depth of the chroma words (EncodeStream -c, pictureio::bitDepth(luma, chroma),
                           EncodeStream.cpp:322); 0 = bit_depth.  Ignored by the decode calls, as the reference's decoder does. */
} vc2hip_picture_format;

typedef struct {
  int kernel, depth;      /* -k -d                                              */
  int y_slices, x_slices; /* from vc2hip_slice_size_is_valid                    */
  int mode;               /* VC2HIP_HQ_CONSTQ / HQ_CBR / LD                     */
  int q_index;            /* ConstQ                                             */
  int compressed_bytes;   /* CBR / LD picture byte budget (-s)                  */
  int prefix, scalar;     /* -P -S                                              */
} vc2hip_coding_params;

/* bytes of one raw planar picture (Y then U then V, big-endian MSB-justified words,
 * Arrays.cpp:333-426) */
size_t vc2hip_raw_picture_bytes(const vc2hip_picture_format *fmt);
/* upper bound of one picture's slice payload */
size_t vc2hip_max_payload_bytes(const vc2hip_picture_format *fmt, const vc2hip_coding_params *cp);

/* host buffers: H2D, kernels, D2H, synchronous.  payload = the slice bytes that follow the
 * transform parameters inside an HQ picture data unit.  qidx_out (ys*xs) may be NULL. */
int vc2hip_encode_picture_hq(vc2hip_ctx *ctx, const void *raw, const vc2hip_picture_format *fmt,
                             const vc2hip_coding_params *cp, uint8_t *payload, size_t cap,
                             size_t *len, int32_t *qidx_out);
/* LD: payload = the slice bytes of an LD picture data unit (cp->mode == VC2HIP_LD,
 * cp->compressed_bytes = -s), EncodeStream.cpp:482-647 in LD mode */
int vc2hip_encode_picture_ld(vc2hip_ctx *ctx, const void *raw, const vc2hip_picture_format *fmt,
                             const vc2hip_coding_params *cp, uint8_t *payload, size_t cap,
                             size_t *len, int32_t *qidx_out);
int vc2hip_decode_picture_hq(vc2hip_ctx *ctx, const uint8_t *payload, size_t len,
                             const vc2hip_picture_format *fmt, const vc2hip_coding_params *cp,
                             void *raw_out);
int vc2hip_decode_picture_ld(vc2hip_ctx *ctx, const uint8_t *payload, size_t len,
                             const vc2hip_picture_format *fmt, const vc2hip_coding_params *cp,
                             void *raw_out);

/* Pipelined picture calls -- the per-GPU host path of SURVEY.md 8(e): one host thread, pinned staging buffers, two
 * pictures in flight per context, so that the H2D copy of one picture, the kernels of another and the D2H copy of a
 * third overlap (each in-flight picture has its own HIP stream and workspace inside the context).  This is what the
 * tools' per-GPU workers drive in place of the reference's one-picture-at-a-time loop (EncodeStream.cpp:452-770,
 * DecodeStream.cpp:289-613); results are those of the synchronous calls.
 *   vc2hip_host_alloc / _free   page-locked host memory for raw pictures and payloads (hipHostMalloc)
 *   *_begin                     enqueue copy-in + kernels (+ copy-out of the raw picture when decoding); returns a ticket
 *   *_end                       wait for that ticket; encode: *len and the payload bytes are in `payload` afterwards
 * The buffers handed to _begin must come from vc2hip_host_alloc and stay untouched until _end returned.  At most
 * VC2HIP_MAX_INFLIGHT tickets may be open per context (VC2HIP_EINVAL beyond); tickets end in the order they began. */
#define VC2HIP_MAX_INFLIGHT 2
void *vc2hip_host_alloc(size_t bytes);
void vc2hip_host_free(void *p);
int vc2hip_encode_picture_begin(vc2hip_ctx *ctx, const void *raw, const vc2hip_picture_format *fmt,
                                const vc2hip_coding_params *cp, uint8_t *payload, size_t cap, int32_t *qidx_out, int *ticket);
int vc2hip_encode_picture_end(vc2hip_ctx *ctx, int ticket, size_t *len);
int vc2hip_decode_picture_begin(vc2hip_ctx *ctx, const uint8_t *payload, size_t len, const vc2hip_picture_format *fmt,
                                const vc2hip_coding_params *cp, void *raw_out, int *ticket);
int vc2hip_decode_picture_end(vc2hip_ctx *ctx, int ticket);

/* device-resident batches: n independent pictures per call, asynchronous on the ctx stream.
 *   d_raw       n * vc2hip_raw_picture_bytes() bytes of raw planar pictures (device memory)
 *   d_payload   n slots of payload_stride bytes each (device memory)
 *   d_lens      n uint64 payload lengths (device memory; written by encode, read by decode)
 * d_raw, d_payload and payload_stride must be multiples of 16 bytes (VC2HIP_EINVAL otherwise); when a
 * picture's raw size is not a multiple of 16 the pictures of a batch are still packed back to back.
 * Nothing is allocated or synchronised inside these calls once the ctx has seen the geometry
 * (first call sizes the workspace).  The kernels run on the ctx stream (vc2hip_create makes its own,
 * vc2hip_create_on_stream takes the caller's): buffers that another stream has written -- a framework's
 * fill or gather kernel, a copy -- must be complete before the call, and the results are complete after
 * vc2hip_sync (or an event the caller records on the ctx stream).
 * One exception, HQ decode on a context made by vc2hip_create / _with_flags without a PLANES8 flag: the form of the
 * decoder's band planes (16-bit or byte elements: same results, different speed) follows the batch before; that batch's
 * payload lengths and escape count come back through pinned memory behind an event.  The context's SECOND decode call
 * waits for that event once (hipEventSynchronize: the first batch must have run); every later call only queries it.
 * On a caller's stream (vc2hip_create_on_stream) the call never waits, and while that stream is being captured into a
 * graph it records nothing.  Callers that need the same kernels launched whatever the host's timing (graph capture,
 * running far ahead of the GPU) create the context with VC2HIP_FLAG_PLANES8_ALWAYS or _NEVER. */
/* Cut every device-resident batch into k contiguous sub-batches, each on its own HIP stream and workspace,
 * forked from and joined to the context's stream (k = 1: off, the default).  The launches of the sub-batches
 * overlap on the GPU; results are identical.  Extension, no counterpart in the reference. */
int vc2hip_set_streams(vc2hip_ctx *ctx, int k);
int vc2hip_encode_batch_dev(vc2hip_ctx *ctx, const void *d_raw, int n,
                            const vc2hip_picture_format *fmt, const vc2hip_coding_params *cp,
                            void *d_payload, size_t payload_stride, uint64_t *d_lens);
int vc2hip_decode_batch_dev(vc2hip_ctx *ctx, const void *d_payload, size_t payload_stride,
                            const uint64_t *d_lens, int n, const vc2hip_picture_format *fmt,
                            const vc2hip_coding_params *cp, void *d_raw_out);

/* Which form the band planes of the context's most recent HQ decode call had: 0 = none (the slice records only), 16 = 16-bit
 * elements, 8 = byte elements (see above: the adaptive choice, or the PLANES8 flags).  Introspection for tests and
 * measurements -- the results never depend on it.  Extension, no counterpart in the reference. */
int vc2hip_band_plane_bits(const vc2hip_ctx *ctx);

/* ---------------------------------------------------------------------------------------------
 * measurement: per-kernel HIP-event timing on the ctx stream (bench.py's roofline leg)
 * ------------------------------------------------------------------------------------------- */
int vc2hip_profile_enable(vc2hip_ctx *ctx, int on); /* on: every launch carries a start / stop event pair */
/* name != NULL: only the launches of that profile entry carry events (a timed region that should pay for the events of one
 * kernel, not of thirty launches per step); NULL: all of them again */
int vc2hip_profile_only(vc2hip_ctx *ctx, const char *name);
/* after vc2hip_sync(): number of distinct kernel names seen since enable */
int vc2hip_profile_count(vc2hip_ctx *ctx);
/* i-th entry: name, launches, total milliseconds */
int vc2hip_profile_get(vc2hip_ctx *ctx, int i, const char **name, int *launches, double *total_ms);
int vc2hip_profile_reset(vc2hip_ctx *ctx);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
