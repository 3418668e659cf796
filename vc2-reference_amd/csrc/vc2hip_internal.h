// Internal declarations shared by the libvc2hip translation units (not installed).
//
// Device data layout (DESIGN.md "Data layout in HBM"):
//   * sample planes        : raw planar words exactly as in the file (big-endian, MSB justified)
//   * LL planes            : one compact row-major int32 plane per wavelet level and component
//   * coefficient store    : [picture][slice][component][band][row][col] int32 -- every slice's
//                            coefficients already in the order the slice coder emits them, so the
//                            pack / unpack kernels stream 4 KiB-contiguous records and the DWT
//                            kernels write / gather whole band blocks of a slice contiguously.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/vc2hip.h"

#define VC2_MAX_DEPTH 7
#define VC2_MAX_BANDS (3 * VC2_MAX_DEPTH + 1)

// device-side error flags (OR-ed into ctx->d_err by kernels)
enum : unsigned {
  VC2_DEVERR_QINDEX = 1u << 0,     // adjusted quantiser index > 119
  VC2_DEVERR_SCALAR = 1u << 1,     // slice component needs > 255*scalar bytes
  VC2_DEVERR_CBR_TOOBIG = 1u << 2, // CBR: V component does not fit the slice
  VC2_DEVERR_CBR_LEN = 1u << 3,    // CBR: V length byte > 255
  VC2_DEVERR_CODE32 = 1u << 4,     // |coef| > 65534
  VC2_DEVERR_STREAM = 1u << 5,     // slice data runs past the payload
  VC2_DEVERR_LD_TOOBIG = 1u << 6,
  VC2_DEVERR_HANDOFF = 1u << 7,    // LD index search: a row waited in vain for the row above (single-launch form)
};

struct CompGeom {
  int h, w;     // picture plane (unpadded)
  int ph, pw;   // padded plane
  int sh, sw;   // slice size in samples (ph / ys, pw / xs)
  int coef_off; // first coefficient of this component inside a slice record
  int n0;       // coefficients of one slice in the LL band: (sh >> depth) * (sw >> depth)
};

struct Geom {
  CompGeom c[3];
  int depth, ys, xs;
  int slice_coefs; // coefficients per slice record (all three components)
};

// offset of band `band` (0 = LL, then HL,LH,HH per level, coarsest first) inside a component
// record whose LL block has n0 coefficients: level L bands hold n0 * 4^(L-1) each.
__host__ __device__ inline int band_offset(int n0, int band) {
  if (band == 0) return 0;
  const int L = (band - 1) / 3 + 1, kind = (band - 1) % 3;
  // n0 * (1 + 3 * (4^(L-1) - 1) / 3) = n0 * 4^(L-1)
  return n0 * ((1 << (2 * (L - 1))) * (1 + kind));
}
// band index of coefficient j (coding order) of a component record
__host__ __device__ inline int band_of_index(int j, int n0) {
  const int m = j / n0;
  if (m == 0) return 0;
#if defined(__HIP_DEVICE_COMPILE__)
  const int lg = 31 - __clz(m);
#else
  const int lg = 31 - __builtin_clz((unsigned)m);
#endif
  const int L = lg / 2 + 1;
  return 3 * (L - 1) + (m >> (2 * (L - 1)));
}

struct QuantTables {
  int32_t qf[120];
  int32_t off[120];
  // exact unsigned division by qf (Granlund-Montgomery): n / qf = (t + ((n - t) >> 1)) >> (l - 1),
  // t = mulhi(magic, n); valid for every 32-bit n when qf > 0 (indices 0..115)
  uint32_t magic[120];
  int32_t shift[120];
  float inv4[120]; // the smallest float >= 4 / qf (k_cbr_search_reg)
};

// ------------------------------------------------------------------------------------------
// kernel parameter blocks
// ------------------------------------------------------------------------------------------
struct LevelParams {
  // input of a forward level / output of an inverse level
  void *plane[3];             // FIRST/FINAL: raw sample words, else LL_l plane (ST elements)
  long long plane_stride[3];  // per picture, in bytes (raw) or elements (int32)
  void *ll[3];                // LL_{l+1} plane, ST elements (forward: output unless LAST; inverse: input)
  long long ll_stride[3];     // per picture, elements
  void *store;                // coefficient store (int32_t or int16_t elements: the kernels' ST parameter)
  long long store_stride;     // per picture, elements
  int32_t *store_wide;        // int16 store: values outside 16 bits, same element index (vc2hip_store.h)
  int32_t *plane_wide[3];     // the same for the level planes (not FIRST / FINAL) ...
  int32_t *ll_wide[3];        // ... and the LL_{l+1} planes
  const int32_t *qidx;        // inverse: per-slice quantiser indices, n_pictures * ys * xs
  int in_h[3], in_w[3];       // plane size at this level (padded >> level)
  int pic_h[3], pic_w[3];     // FIRST/FINAL: unpadded picture size
  int fh[3], fw[3];           // slice footprint at this level, samples
  int tsy[3], tsx[3];         // slices per tile
  int tiles_y[3], tiles_x[3];
  int coef_off[3];
  int band_off[3];            // offset of this level's HL band inside a component record
  int band_n[3];              // coefficients per band block (bsh * bsw)
  int band;                   // band index of HL at this level (for the quant matrix)
  int ys, xs, slice_coefs;
  int rec_stride[3];          // tile kernels (vc2hip_dwt_fast.hip): elements from one slice's coefficients of a component to the next
                              // slice's -- slice_coefs in the slice records, HeadSplit::n[c] in the record heads
  int word_bytes, sample_shift, sample_offset; // raw sample format (luma; the decoder's one format)
  int sample_shift_c, sample_offset_c;         // FIRST: the chroma words' (EncodeStream -c: its own depth)
  int clip_lo, clip_hi;
  int ll_from_store;          // inverse, coarsest level: LL comes from store band 0 (dequantised)
  int ll_to_store;            // forward, last level: LL goes to store band 0
  int dequant;                // inverse: apply scale() to store values
  unsigned *err;              // device error flags
  int debug_skip;             // -DVC2HIP_ABLATE builds only (tools/ablate_*.py): 1 no loads, 2 no lifting, 4 no stores
  int qmatrix[VC2_MAX_BANDS];
  // streaming kernels (vc2hip_dwt_stream.hip), set by vc2_stream_level_applicable
  int st_strips[3], st_segs[3]; // wavefronts across / down a plane (segments: whole rows of slices)
  int st_out[3];                // chunks (8 samples) a strip owns
  int st_llps[3];               // log2 chunks per slice
  int st_tail;                  // some plane's pair count is not a multiple of four: the TAIL instantiation
  int st_prio;                  // > 0: wavefronts take turns at the highest issue priority, a new turn every 2^st_prio row pairs
  int st_segmax, st_npic;       // most segments of any component; pictures of the launch (the kernels' work-item numbering)
  int st_lds;                   // dynamic LDS bytes of ONE wavefront (a workgroup holds VC2_STREAM_WG_WAVES of them)
  // inverse, streaming kernels: element offset (from the picture's store) of this level's HL band plane, LH and HH behind
  // it, when the decoder keeps the level's bands as planes (BandPlanes below); -1: in the slice records
  long long bp_base[3];
  int bp8;                      // those planes hold one byte per coefficient (BandPlanes::bytes8)
};

// Two consecutive levels in one launch (vc2hip_dwt_pair.hip): `a` is the finer level exactly as the one-level kernels see
// it (its st_* fields describe the wavefronts: a lane holds 8 samples of level a's input plane), `b` the level below it
// (fill_level of level + 1: the same lanes hold 4 samples of ITS input, which never leave the registers).
struct PairParams {
  LevelParams a, b;
  int spl[3];        // slices per lane: 1, or 2 when a slice footprint at level a is 4 samples wide (then st_llps = 0)
  int img_b;         // LDS images of level b: byte offset behind level a's (forward)
  int grp_a;         // forward: slice rows per burst of stores = images of level a (a power of two); level b has grp_a + 1
  int sz_a, sz_b;    // forward: elements of one image of level a / b
  int ring_ll, n_ll; // forward: byte offset and rows (a power of two) of the ring of level b's LL rows
  int ss_a[3], ss_b[3]; // forward: elements from one slice's run to the next in the LDS images (padded against bank conflicts)
  int piece_a[3], piece_b[3]; // forward: bytes per flush piece (16, 8 or 4)
};

// Tuning knobs (lanes per slice, wavefronts per workgroup, segments per strip ...) are compile-time decisions of the
// release library; the environment can override them only in the -DVC2HIP_ABLATE build (A/B measurements on one box).
#include <stdlib.h>
#ifdef VC2HIP_ABLATE
static inline int vc2_tune_int(const char *name, int def) { const char *e = getenv(name); return e ? atoi(e) : def; }
#else
static inline int vc2_tune_int(const char *, int def) { return def; }
#endif

// Work-skipping switches for the timing experiments of tools/ablate_*.py exist only in a library built with
// -DVC2HIP_ABLATE (the tools build their own); the release kernels carry no such test.
#ifdef VC2HIP_ABLATE
#define VC2_SKIP(p, bit) (((p).debug_skip & (bit)) != 0)
#else
#define VC2_SKIP(p, bit) false
#endif

struct PackParams {
  const void *store;          // int32_t or int16_t elements (store16)
  const int32_t *store_wide;
  int store16;
  long long store_stride;
  const int32_t *qidx;        // n_pictures * n_slices (ConstQ: filled by host)
  int n_slices, slice_coefs;
  int comp_n[3], comp_off[3], comp_n0[3];
  int depth;
  int prefix, scalar;
  int qmatrix[VC2_MAX_BANDS];
  // VBR: fixed-stride slots + sizes; CBR: direct
  uint8_t *slots;             // n_pictures * n_slices * slot_bytes
  int slot_bytes;
  uint32_t *sizes;            // n_pictures * n_slices (tile_slices: n_pictures * workgroups per picture)
  int tile_slices;            // > 0: the slices of a pack workgroup (that many) share a slot of tile_slices * slot_bytes, back to back
  const int32_t *cbr_bytes;   // per slice (one picture's worth, shared), NULL => VBR
  const uint32_t *cbr_offsets;
  uint8_t *payload;
  long long payload_stride;
  unsigned *err;
  int quantise;               // 0: store already holds quantised values (fine-grained API)
  int big_lut;                // set by the launcher: LDS holds subband tables for components of up to 2048 coefficients
  float inv_scalar;           // set by the launcher: the smallest float >= 1 / scalar
  union {
    unsigned char band_lut[768]; // set by the launcher: subband of a coefficient index, 512 luma + 256 chroma entries
    unsigned lane16[200];        // k_hq_pack16 / k_hq_pack16w (vc2hip_pack16.h): per lane, the matrix entries of its coefficients; head / body sizes
  };
  int debug_skip;             // timing experiments only (VC2HIP_DEBUG_PACK): 1 no code writes, 2 no copy-out
  // single-pass VBR: slice offsets by decoupled look-back over workgroup tiles (4 slices each).
  // lookback: per picture [0] = tile ticket, [1 + t] = status of tile t: flag (2 bits) << 62 | bytes
  unsigned long long *lookback;
  long long lookback_stride;  // u64 words per picture
  unsigned long long *lens;   // per picture payload length (written by the last tile)
};

// Decode side only: the bands of the finest levels -- those the streaming inverse kernels read -- are kept as whole
// planes behind the slice records instead of inside them.  The inverse transform then reads every band row with fully
// coalesced loads (lanes = neighbouring columns) where the slice records give it 8 or 16 bytes per slice at slice-record
// distance; the slice decoder, whose lanes are neighbouring slices, writes a band row piece per lane, also contiguous.
// (The encoder keeps slice records throughout: its transform is not bound by the store writes.)
constexpr int VC2_BP_MAX = 3;
struct BandPlanes {
  int levels;                       // finest levels kept as planes (0: none)
  int bytes8;                       // round 5: a plane element is ONE BYTE (quantised coefficients are small; -128 = escape: the value is in
                                    // the wide array at the element's index, like the 16-bit sentinel's).  A plane that starts at element
                                    // `base` of the picture's store keeps its start; element e of it lives at byte 2 * base + (e - base).
  int from[3];                      // first coefficient of a component record that lives in a plane
  long long base[3][VC2_BP_MAX];    // element offset, from the picture's store, of band 1 (HL) of level l; LH, HH follow
  int ow[3][VC2_BP_MAX], np[3][VC2_BP_MAX];     // width / height of a band plane
  int lbsh[3][VC2_BP_MAX], lbsw[3][VC2_BP_MAX]; // log2 of a slice's block in it
};

// Record heads: the coefficients of the DEEP levels (those below the streaming kernels: LL and the bands of levels >= Ls) are
// the first head_n[c] coefficients of every component record -- 8 to 64 bytes per component at slice-record distance, and
// the tile kernels of those levels fetched (wrote) a whole 128-byte line for each (3.6 - 7 x their bytes, rocprofv3
// FETCH_SIZE).  With a head split they live in one dense array per component instead: [slice][coefficient]; inside a
// slice the order is that of the record, so the deep levels see a coefficient store with rec_stride[c] = n[c] and
// coef_off[c] = base[c].
struct HeadSplit {
  int n[3];            // coefficients of a component record that live in the head (0: no split); multiples of 8
  long long base[3];   // element offset, from the picture's store, of the component's head array: [slice][n[c]]
};

struct UnpackParams {
  const uint8_t *payload;
  long long payload_stride;
  const unsigned long long *lens; // per picture
  const uint32_t *offsets;    // n_pictures * n_slices slice start offsets
  void *store;                // int32_t or int16_t elements (store16)
  int32_t *store_wide;
  int store16;
  long long store_stride;
  int32_t *qidx;
  int n_slices, slice_coefs;
  int comp_n[3], comp_off[3];
  int prefix, scalar;
  unsigned *err;
  BandPlanes bp;
  HeadSplit hs;
  int xs;                     // slices across
  unsigned long long *stats;  // [0]: pieces that carried an escape from a byte plane (k_hq_unpack16<true>; feedback for the next batch's layout)
  int debug_skip;             // timing experiments only (VC2HIP_DEBUG_UNPACK, -DVC2HIP_ABLATE): 1 stream reads from a hot 16 KiB window, 2 no stores
};

struct CbrParams {
  const void *store;          // int32_t or int16_t elements (store16)
  const int32_t *store_wide;
  int store16;
  long long store_stride;
  int32_t *qidx;
  const int32_t *slice_bytes;
  int n_slices, slice_coefs;
  int comp_n[3], comp_off[3], comp_n0[3];
  int scalar;
  int n_bands;
  int qmatrix[VC2_MAX_BANDS];
  unsigned *err;
  int only_marked;            // set by the launcher: search only the slices whose index is VC2_CBR_MARK
  int general_only;           // VC2HIP_FLAG_CBR_GENERAL: no register kernel (tests, A/B)
  float inv_scalar;           // set by the launcher
  int qm_min;                 // set by the launcher: the smallest matrix entry
  union {
    unsigned char band_lut[768]; // set by the launcher: subband of a coefficient index, 512 luma + 256 chroma entries
    unsigned lane8[72];          // k_cbr_search16 (vc2hip_cbr16.h): per lane the matrix entries of its runs and head; [64..67] head / run counts
  };
};
#define VC2_CBR_MARK 0x7FFFFFFF

// ------------------------------------------------------------------------------------------
// launchers (implemented in the kernel TUs)
// ------------------------------------------------------------------------------------------
struct Launcher; // profiling hook, defined in vc2hip_api.hip
// Every kernel launch goes through VC2_LAUNCH between vc2_prof_begin / vc2_prof_end.  With profiling on it carries
// its own start / stop events (hipExtLaunchKernelGGL: the timestamps of the dispatch packet itself, no extra packets
// on the stream between kernels); off, the events are null and it is a plain launch.
void vc2_prof_pair(Launcher &L, hipEvent_t *a, hipEvent_t *b);
#define VC2_LAUNCH(L, kernel, grid, block, lds, s, ...)                                        \
  do {                                                                                         \
    hipEvent_t vc2_ev_a_, vc2_ev_b_;                                                           \
    vc2_prof_pair(L, &vc2_ev_a_, &vc2_ev_b_);                                                  \
    hipExtLaunchKernelGGL(kernel, grid, block, lds, s, vc2_ev_a_, vc2_ev_b_, 0, __VA_ARGS__);  \
  } while (0)

// Raise a kernel's dynamic-LDS limit to `bytes`, once per (kernel, device): the attribute belongs to the
// function on the CURRENT device, and one process may drive several GPUs (the tools' --gpus N).
void vc2_allow_lds(const void *kernel, size_t bytes);

void vc2_upload_tables(const QuantTables &t, hipStream_t s);
int vc2_launch_forward_level(Launcher &L, int kernel, bool first, const LevelParams &p, int n_pictures,
                             hipStream_t s); // generic kernels: int32 store and planes only
int vc2_launch_inverse_level(Launcher &L, int kernel, bool final_level, const LevelParams &p,
                             int n_pictures, hipStream_t s);
size_t vc2_level_lds_bytes(int kernel, const LevelParams &p);

void vc2_launch_pack(Launcher &L, const PackParams &p, int n_pictures, hipStream_t s);
int vc2_pack_slices_per_tile(const PackParams &p);
bool vc2_pack_one_pass_default(const PackParams &p); // VBR slices coded by k_hq_pack16: look-back, no slots
void vc2_launch_scan_sizes(Launcher &L, const uint32_t *sizes, uint32_t *offsets,
                           unsigned long long *totals, int n_slices, int n_pictures, hipStream_t s);
void vc2_launch_compact(Launcher &L, const uint8_t *slots, int slot_bytes, const uint32_t *sizes,
                        const uint32_t *offsets, uint8_t *payload, long long payload_stride,
                        int n_slices, int n_pictures, hipStream_t s);
void vc2_launch_cbr(Launcher &L, const CbrParams &p, int n_pictures, hipStream_t s);
void vc2_launch_unpack(Launcher &L, const UnpackParams &p, int n_pictures, hipStream_t s);
void vc2_launch_cbr_index(Launcher &L, const uint8_t *payload, long long stride, const unsigned long long *lens, const int32_t *budget,
                          const uint32_t *cbr_offs, unsigned long long total, uint32_t *offsets, int n_slices, int prefix, int scalar,
                          int n_pictures, unsigned *bad, hipStream_t s);
void vc2_launch_slice_index(Launcher &L, const uint8_t *payload, long long payload_stride,
                            const unsigned long long *lens, uint32_t *offsets, int n_slices,
                            int prefix, int scalar, int n_pictures, unsigned *err, hipStream_t s,
                            void *workspace, size_t workspace_bytes, const unsigned *skip = nullptr);
size_t vc2_slice_index_workspace(int n_pictures, size_t max_payload, int prefix, int scalar);

// layout conversion for the fine-grained API (interleaved in-place plane <-> coefficient store)
void vc2_launch_plane_to_store(Launcher &L, const int32_t *plane, int ph, int pw, int depth, int ys,
                               int xs, int32_t *store, int slice_coefs, int coef_off, hipStream_t s);
void vc2_launch_store_to_plane(Launcher &L, const int32_t *store, int slice_coefs, int coef_off,
                               int32_t *plane, int ph, int pw, int depth, int ys, int xs,
                               const int32_t *qidx, const int *qmatrix, int mode /*0 copy,1 scale*/,
                               unsigned *err, hipStream_t s);
void vc2_launch_quantise_store(Launcher &L, int32_t *store, int n_slices, int slice_coefs, int comp_n,
                               int comp_off, int n0, const int32_t *qidx, const int *qmatrix,
                               unsigned *err, hipStream_t s);
// LD: DC-predicted reconstruction of the LL band of one component (Quantisation.cpp:287-306)
struct LdLl3Params {
  const int32_t *store;
  long long store_stride;
  int slice_coefs, coef_off[3], llh[3], llw[3], ys, xs;
  const int32_t *qidx;
  int qm0;
  int32_t *ll_plane[3];
  long long ll_stride[3];
  unsigned *err;
};
bool vc2_launch_ld_ll3(Launcher &L, const LdLl3Params &p, int n_pictures, hipStream_t s); // false: a plane does not fit in LDS (use vc2_launch_ld_ll)
void vc2_launch_ld_ll(Launcher &L, const int32_t *store, long long store_stride, int slice_coefs,
                      int coef_off, int n0, int llh, int llw, int ys, int xs, const int32_t *qidx,
                      int qm0, int32_t *ll_plane, long long ll_stride, int n_pictures, unsigned *err,
                      hipStream_t s);
struct LdUnpackParams {
  const uint8_t *payload;
  long long payload_stride;
  const int32_t *slice_bytes; // per slice
  const uint32_t *offsets;    // per slice (prefix sums of slice_bytes)
  int32_t *store;
  long long store_stride;
  int32_t *qidx;
  int n_slices, slice_coefs;
  int comp_n[3], comp_off[3];
  unsigned *err;
  // A slice whose luma length field exceeds the slice (corrupt data; also what the reference's decoder makes of its own
  // interlaced LD streams, DecodeStream.cpp:331): the reference reads that many bits from the stream (Slices.cpp:246-303),
  // so every later slice of the picture starts late.  The first pass decodes at the nominal offsets and raises the
  // picture's flag when it meets such a slice; a serial walk then finds the true starts of a flagged picture and a second
  // pass decodes it again from those (both return at once for pictures without the flag).
  const unsigned long long *lens; // per picture: payload bytes the decoder may look at (null: payload_stride)
  unsigned *shifted;              // per picture: flag (null: no such handling, slices stay at their nominal offsets)
  uint32_t *starts;               // per picture x slice: true start offsets of a flagged picture
  int redo;                       // the second pass
};
void vc2_launch_ld_unpack(Launcher &L, const LdUnpackParams &p, int n_pictures, hipStream_t s);
void vc2_launch_ld_walk(Launcher &L, const LdUnpackParams &p, int n_pictures, hipStream_t s);

// LD encode: quantiser search / DC-predicted quantisation (one launch per slice anti-diagonal) and
// the slice writer
struct LdEncParams {
  int32_t *store;             // in: transform coefficients; out: quantised (LL as prediction residuals)
  int32_t *scratch;           // slices too large for LDS: the trials' quantised values, same shape as the store (null: such slices are refused)
  long long store_stride;
  int32_t *qidx;              // n_pictures * n_slices; written when search != 0
  const int32_t *slice_bytes; // per slice
  const uint32_t *offsets;    // per slice (prefix sums of slice_bytes)
  int32_t *restored[3];       // reconstructed LL planes (scratch), ll_h x ll_w per picture
  long long restored_stride[3];
  int ll_w[3];
  int bh[3], bw[3];           // LL block of one slice
  int ys, xs, n_slices, slice_coefs;
  int comp_n[3], comp_off[3], comp_n0[3];
  int depth;
  int rs_ints;                // LDS ints per wavefront for the LL blocks with their halo: sum of (bh + 1) * (bw + 1)
  int qmatrix[VC2_MAX_BANDS];
  int search;
  int diagonals;              // VC2HIP_FLAG_LD_DIAGONALS: one launch per slice anti-diagonal instead of the single-launch search
  int *tab;                   // device scratch for the search tables (LD_TAB_INTS ints)
  int img_words;              // LDS words of one slice image (pack)
  uint8_t *payload;
  long long payload_stride;
  unsigned *err;
  int debug_dead_row;         // -DVC2HIP_ABLATE builds only: the workgroup of this slice row of picture 0 exits at once (hand-over failure test)
};
void vc2_launch_ld_quantise(Launcher &L, const LdEncParams &p, int n_pictures, hipStream_t s);
void vc2_launch_ld_pack(Launcher &L, const LdEncParams &p, int n_pictures, hipStream_t s);
