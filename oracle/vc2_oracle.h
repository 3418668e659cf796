/*
 * vc2_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, scalar, obviously-correct restatement of the bbc/vc2-reference hot
 * path (WaveletTransform -> Quantisation -> Slices/VLC) plus the minimum of
 * stream syntax needed to reproduce whole reference streams.  It is the CPU
 * checker the HIP path is compared with; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library
 * (libvc2hip.so) never links or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - quant() known answers of the reference's own unit test
 *     (tests/Quantisation.cpp:30-36) -> tests/test_oracle.py
 *   - exp-Golomb coding cross-checked against the reference's own VLC.cpp
 *     compiled unmodified into oracle/_ref/ (it is the one Library TU that
 *     builds without Boost)
 *   - whole-stream / decoded-file SHA-256 digests of reference output recorded
 *     in SURVEY.md Appendix B (cfg 1..4)  -> tests/test_oracle_digests.py
 *   - LD profile: no reference vectors exist in this container: parity unpinned.
 *
 * All arithmetic is 32-bit two's complement, like the reference's `int`.
 * Every function returns 0 on success or a negative VC2O_E* code; the text of
 * the reference exception that would have been thrown is in vc2o_last_error().
 */
#ifndef VC2_ORACLE_H
#define VC2_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* WaveletTransform.h:26 enum order == wavelet_index written to the stream */
enum { VC2O_DD97 = 0, VC2O_LEGALL = 1, VC2O_DD137 = 2, VC2O_HAAR0 = 3,
       VC2O_HAAR1 = 4, VC2O_FIDELITY = 5, VC2O_DAUB97 = 6 };
/* Picture.h:17 */
enum { VC2O_CF444 = 0, VC2O_CF422 = 1, VC2O_CF420 = 2 };

enum { VC2O_OK = 0, VC2O_EINVAL = -1, VC2O_EQINDEX = -2, VC2O_ESCALAR = -3,
       VC2O_ECBR_TOOBIG = -4, VC2O_ECBR_LEN = -5, VC2O_ECBR_WRONG = -6,
       VC2O_EBOUNDED = -7, VC2O_ELD_TOOBIG = -8, VC2O_ECAP = -9, VC2O_ESTREAM = -10 };

/* Geometry of one coded picture: padded plane sizes + slice grid. */
typedef struct {
  int luma_h, luma_w;     /* padded luma plane   */
  int chroma_h, chroma_w; /* padded chroma plane */
  int depth;              /* wavelet depth       */
  int y_slices, x_slices; /* slice grid          */
} vc2o_geom;

const char *vc2o_last_error(void);

/* ---- geometry: WaveletTransform.cpp:74-77, :116-136; Picture.cpp:49-73 ---- */
int vc2o_padded_size(int size, int depth);
int vc2o_slice_size_is_valid(int depth, int len_luma, int len_chroma, int n_size);
void vc2o_chroma_dims(int h, int w, int cf, int *ch, int *cw);

/* ---- sample I/O: Arrays.cpp:333-426, Picture.cpp:284-292 ---- */
void vc2o_ingest(const uint8_t *raw, int word_bytes, int bit_depth, size_t n, int32_t *out);
void vc2o_clip_emit(const int32_t *in, size_t n, int word_bytes, int bit_depth, uint8_t *out);

/* ---- transform: WaveletTransform.cpp:79-94, :262-342, :478-1265 ---- */
void vc2o_pad(const int32_t *in, int h, int w, int32_t *out, int ph, int pw);
int vc2o_dwt_forward(int32_t *plane, int ph, int pw, int kernel, int depth);
int vc2o_dwt_inverse(int32_t *plane, int ph, int pw, int kernel, int depth);
int vc2o_quant_matrix(int kernel, int depth, int32_t *out /* 3*depth+1 */);

/* ---- quantiser: Quantisation.cpp:40-95 ---- */
int vc2o_quant_factor(int q, int32_t *out);
int vc2o_quant(int32_t v, int aq, int32_t *out);
int vc2o_scale(int32_t v, int aq, int32_t *out);
/* Quantisation.cpp:479-489 / :534-544 (one plane, per-slice indices) */
int vc2o_quantise_np(const int32_t *coef, int ph, int pw, int depth, const int32_t *qidx,
                     int ys, int xs, const int32_t *qmatrix, int32_t *out);
int vc2o_dequantise_np(const int32_t *q, int ph, int pw, int depth, const int32_t *qidx,
                       int ys, int xs, const int32_t *qmatrix, int32_t *out);
/* Quantisation.cpp:357-379 (LD: LL band DC-predicted, :191-234, :287-306) */
int vc2o_quantise_ld(const int32_t *coef, int ph, int pw, int depth, const int32_t *qidx,
                     int ys, int xs, const int32_t *qmatrix, int32_t *out);
int vc2o_dequantise_ld(const int32_t *q, int ph, int pw, int depth, const int32_t *qidx,
                       int ys, int xs, const int32_t *qmatrix, int32_t *out);

/* ---- slices: Slices.cpp:18-119, :195-612 ---- */
int vc2o_slice_bytes(int ys, int xs, int total_bytes, int scalar, int32_t *out);
/* component_slice_bytes of slice (v,h) of one quantised plane */
int vc2o_component_slice_bytes(const int32_t *plane, int ph, int pw, int depth, int ys, int xs,
                               int v, int h, int scalar, int32_t *bytes_out);
/* HQ picture payload (all slices, raster order).  cbr_slice_bytes NULL => VBR. */
int vc2o_hq_pack(const int32_t *y, const int32_t *u, const int32_t *v, const vc2o_geom *g,
                 const int32_t *qidx, int prefix, int scalar, const int32_t *cbr_slice_bytes,
                 uint8_t *out, size_t cap, size_t *out_len);
int vc2o_hq_unpack(const uint8_t *in, size_t len, const vc2o_geom *g, int prefix, int scalar,
                   int32_t *y, int32_t *u, int32_t *v, int32_t *qidx, size_t *consumed);
/* EncodeStream.cpp:73-125 (works on the TRANSFORM, not quantised, planes) */
int vc2o_cbr_qindices(const int32_t *y, const int32_t *u, const int32_t *v, const vc2o_geom *g,
                      const int32_t *qmatrix, const int32_t *slice_bytes, int scalar,
                      int32_t *qidx);
/* LD slices: Slices.cpp:195-303; EncodeStream.cpp:141-245 */
int vc2o_ld_pack(const int32_t *y, const int32_t *u, const int32_t *v, const vc2o_geom *g,
                 const int32_t *qidx, const int32_t *slice_bytes, uint8_t *out, size_t cap,
                 size_t *out_len);
int vc2o_ld_unpack(const uint8_t *in, size_t len, const vc2o_geom *g, const int32_t *slice_bytes,
                   int32_t *y, int32_t *u, int32_t *v, int32_t *qidx, size_t *consumed);
int vc2o_ld_qindices(const int32_t *y, const int32_t *u, const int32_t *v, const vc2o_geom *g,
                     const int32_t *qmatrix, const int32_t *slice_bytes, int32_t *qidx);

/* ---- whole-file drivers mirroring EncodeStream.cpp:247-788 (-o Stream) and
 *      DecodeStream.cpp:103-992 (-o Decoded), progressive only ---- */
typedef struct {
  int width, height, cf;     /* -x -y -f */
  int bit_depth, word_bytes; /* -l -n    */
  int kernel, depth;         /* -k -d    */
  int y_size, x_size;        /* -u -a (slice size in units of 2^depth) */
  int mode;                  /* 0 HQ_ConstQ, 1 HQ_CBR, 2 LD */
  int q_index;               /* -q */
  int compressed_bytes;      /* -s */
  int scalar, prefix;        /* -S -P */
  int frame_rate;            /* -r (default 3) */
  int interlaced;            /* -i: every frame is coded as two field pictures (EncodeStream.cpp:431, :468-475) */
  int bottom_field_first;    /* -b (default top field first, EncodeParams.cpp:124) */
  int fragment_length;       /* -F: >0 writes HQ_CBR / LD pictures as fragments (EncodeStream.cpp:444-445) */
} vc2o_params;

/* returns bytes written to `out` via *out_len; n_frames pictures are read from raw */
int vc2o_encode_stream(const vc2o_params *p, const uint8_t *raw, int n_frames, uint8_t *out,
                       size_t cap, size_t *out_len);
/* decodes every picture of `stream` into raw planar frames; *n_frames_out counts them.
 * optional dumps (may be NULL): last picture's quantised coefficient planes / indices */
int vc2o_decode_stream(const vc2o_params *p, const uint8_t *stream, size_t len, uint8_t *raw_out,
                       size_t cap, int *n_frames_out);

/* stream-syntax pieces (DataUnit.cpp:80-123, :236-266, :563-881, :1062-1078) */
size_t vc2o_write_parse_info(uint8_t *out, int parse_code, uint32_t next, uint32_t prev);
int vc2o_write_sequence_header_payload(const vc2o_params *p, uint8_t *out, size_t cap,
                                       size_t *len, int *major_version);
int vc2o_write_hq_picture_header(uint32_t picture_number, int kernel, int depth, int slices_x,
                                 int slices_y, int prefix, int scalar, int major_version,
                                 uint8_t *out, size_t cap, size_t *len);

#ifdef __cplusplus
}
#endif
#endif
