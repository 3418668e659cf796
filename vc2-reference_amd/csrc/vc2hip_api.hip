// libvc2hip C-ABI (include/vc2hip.h): context, workspace, launch sequencing, error mapping.
// No torch types, no CPU fallback: every entry point either runs the HIP kernels or fails.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <map>
#include <mutex>
#include <utility>
#include <vector>

#include "vc2hip_internal.h"

void vc2_upload_tables_slices(const QuantTables &t, hipStream_t s);
int vc2_halo_x(int kernel);
int vc2_halo_y(int kernel);
void vc2_upload_vlc_lut(hipStream_t s);
void vc2_upload_unpack_lut(hipStream_t s);
bool vc2_slice_index_supported(int prefix, int scalar);
size_t vc2_pack_lds_bytes(int prefix, int scalar);
int vc2_pack_image_mode(int prefix, int scalar);
void vc2_upload_tables_fast(const QuantTables &t, hipStream_t s);
bool vc2_fast_level_applicable(LevelParams &p);
int vc2_launch_forward_fast(Launcher &L, int kernel, bool first, const LevelParams &p, int n, bool store16, hipStream_t s);
int vc2_launch_inverse_fast(Launcher &L, int kernel, bool final_level, const LevelParams &p, int n, bool store16, hipStream_t s);
void vc2_upload_tables_stream(const QuantTables &t, hipStream_t s);
size_t vc2_stream_level_applicable(LevelParams &p, int kernel, bool edge, bool inverse, bool store16, int n_pictures);
int vc2_launch_forward_stream(Launcher &L, int kernel, bool first, const LevelParams &p, int n, bool store16, size_t lds, hipStream_t s);
int vc2_launch_inverse_stream(Launcher &L, int kernel, bool final_level, const LevelParams &p, int n, bool store16, size_t lds, hipStream_t s);
size_t vc2_pair_applicable(PairParams &pp, int kernel, bool edge, bool inverse, bool store16, int n_pictures);
int vc2_launch_forward_pair(Launcher &L, int kernel, bool first, const PairParams &pp, int n, bool store16, size_t lds, hipStream_t s);
int vc2_launch_inverse_pair(Launcher &L, int kernel, bool final_level, const PairParams &pp, int n, bool store16, size_t lds, hipStream_t s);
void vc2_upload_tables_pair(const QuantTables &t, hipStream_t s);
int vc2_launch_plane_transform(Launcher &L, int kernel, int32_t *plane, long long plane_stride, int ph, int pw, int depth, bool inverse,
                               int n, hipStream_t s);
void vc2_launch_plane_ingest(Launcher &L, const void *raw, long long raw_stride, int pic_h, int pic_w, int word_bytes, int bit_depth,
                             int32_t *plane, long long plane_stride, int ph, int pw, int n, hipStream_t s);
void vc2_launch_plane_emit(Launcher &L, const int32_t *plane, long long plane_stride, int pw, void *raw, long long raw_stride, int pic_h,
                           int pic_w, int word_bytes, int bit_depth, int n, hipStream_t s);
void vc2_launch_ll_into_plane(Launcher &L, const int32_t *ll, long long ll_stride, int llh, int llw, int32_t *plane, long long plane_stride,
                              int pw, int depth, int n, hipStream_t s);
void vc2_launch_fill_i32(Launcher &L, int32_t *p, int32_t v, size_t n, hipStream_t s);
void vc2_launch_fill_u64(Launcher &L, unsigned long long *p, unsigned long long v, size_t n, hipStream_t s);

// ------------------------------------------------------------------------------------------
// profiling hook: optional hipEvent pair around every launch
// ------------------------------------------------------------------------------------------
struct ProfEntry {
  std::string name;
  int launches = 0;
  double ms = 0;
};
struct Launcher {
  bool on = false;
  std::string only;               // non-empty: event pairs for the launches of these names only (vc2hip_profile_only: "a" or "a,b,c")
  const char *last_name = nullptr;
  std::string launch_error;
  std::vector<ProfEntry> entries;
  // A start / stop event pair per kernel launch, attached to the launch itself (see VC2_LAUNCH)
  struct Pending { int entry; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> pool;
  int cur_entry = -1;
  hipEvent_t get() {
    hipEvent_t e;
    if (!pool.empty()) { e = pool.back(); pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
  }
  void collect() {
    for (auto &p : pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { entries[p.entry].ms += ms; entries[p.entry].launches++; }
      pool.push_back(p.a);
      pool.push_back(p.b);
    }
    pending.clear();
  }
};
void vc2_prof_begin(Launcher &L, const char *name, hipStream_t s) {
  (void)s;
  L.last_name = name;
  if (!L.on) return;
  if (!L.only.empty() && ("," + L.only + ",").find(std::string(",") + name + ",") == std::string::npos) return; // (a comma-separated list of names)
  int idx = -1;
  for (size_t i = 0; i < L.entries.size(); ++i) if (L.entries[i].name == name) { idx = (int)i; break; }
  if (idx < 0) { L.entries.push_back(ProfEntry{name, 0, 0}); idx = (int)L.entries.size() - 1; }
  L.cur_entry = idx;
}
void vc2_prof_pair(Launcher &L, hipEvent_t *a, hipEvent_t *b) {
  *a = *b = nullptr;
  if (!L.on || L.cur_entry < 0) return;
  *a = L.get();
  *b = L.get();
  L.pending.push_back(Launcher::Pending{L.cur_entry, *a, *b});
}
void vc2_prof_end(Launcher &L, hipStream_t s) {
  (void)s;
  // every launcher calls this right after its launches: catch launch failures (bad grid / LDS size)
  const hipError_t le = hipGetLastError();
  if (le != hipSuccess && L.launch_error.empty())
    L.launch_error = std::string("kernel launch failed (") + (L.last_name ? L.last_name : "?") + "): " + hipGetErrorString(le);
  L.cur_entry = -1;
#ifdef VC2HIP_ABLATE // VC2HIP_DEBUG_SYNC=1 (tools/probe/fault_bisect.py): wait for every stage and name it -- the last name printed before a fault is the stage at fault
  {
    static const bool dbg_sync = getenv("VC2HIP_DEBUG_SYNC") != nullptr;
    if (dbg_sync) {
      fprintf(stderr, "vc2hip stage %s ...", L.last_name ? L.last_name : "?");
      fflush(stderr);
      const hipError_t e = hipStreamSynchronize(s);
      fprintf(stderr, " %s\n", e == hipSuccess ? "done" : hipGetErrorString(e));
      fflush(stderr);
    }
  }
#endif
}
void vc2_prof_break(Launcher &L) { (void)L; }

// ------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------
struct Buf {
  void *p = nullptr;
  size_t cap = 0;
};
enum { B_RAW, B_STORE, B_STOREW, B_LL0, B_LLW, B_QIDX, B_SLOTS, B_SIZES, B_OFFS, B_LENS, B_PAYLOAD,
       B_INDEX, B_PLANE, B_PLANE2, B_CBRB, B_CBRO, B_QM, B_COUNT };

struct vc2hip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  unsigned *d_err = nullptr;
  unsigned *h_err = nullptr; // pinned
  Buf buf[B_COUNT];
  Launcher L;
  std::string err;
  // cached CBR / LD slice-size tables (re-uploaded only when the parameters change)
  int cbr_key[5] = {-1, -1, -1, -1, -1};
  uint64_t cbr_total = 0;
  int debug_skip = 0;
  // VBR payload assembly.  Default: fixed-stride slots + scan + compaction (three launches).
  // VC2HIP_SINGLE_PASS_VBR=1: decoupled look-back inside the pack kernel -- measured 2x SLOWER on
  // MI355X when it was measured (1.28 vs 0.62 + 0.20 ms per 16 UHD pictures: the look-back sits on every workgroup's
  // critical path and the kernel is latency-bound), kept as a tested alternative.
  // Round 6: where k_hq_pack16 codes the slices AND a launch holds enough pictures the look-back IS the default (its grid
  // puts the same tile of all pictures side by side: vc2hip_pack16.h); VC2HIP_FLAG_TWO_PASS_VBR keeps the slots there too.
  bool two_pass_vbr = true;
  bool force_two_pass_vbr = false;
#ifndef VC2_ONE_PASS_MIN_PICTURES
#define VC2_ONE_PASS_MIN_PICTURES 112
#endif
  bool force_generic = false; // VC2HIP_GENERIC_DWT=1: always use the generic level kernels (tests)
  bool allow_store16 = true;  // VC2HIP_STORE32=1: keep the int32 coefficient store on the batch path too (tests, A/B)
  bool allow_planes = true; // decode: band planes for the streaming levels (A/B and test switch VC2HIP_NO_BANDPLANES)
  // Byte band planes (BandPlanes::bytes8): 0 = by what the previous batch looked like (its payload bits per sample, its
  // escapes), 1 = always, 2 = never (VC2HIP_FLAG_PLANES8_ALWAYS / _NEVER).  The choice never changes a result: a value that
  // does not fit a byte escapes to the wide array; it decides between half the band-plane bytes and many escapes.
  int planes8_mode = 0;
  bool planes8_on = false;             // the adaptive state: off until a batch has shown small coefficients
  unsigned long long *d_stat = nullptr; // device: [0] pieces with an escape from a byte plane (k_hq_unpack16<true>)
  unsigned long long *h_stat = nullptr; // pinned: [0] that count, [1..] the batch's payload lengths (first 60 pictures)
  hipEvent_t stat_ev = nullptr;
  bool stat_pending = false, stat_was8 = false, stat_seen = false;
  int stat_n = 0;                      // lengths copied
  int stat_n_total = 0;                // pictures of that batch (the escape count is over all of them)
  int last_plane_bits = 0;             // vc2hip_band_plane_bits: the most recent HQ decode call's band planes (0 none, 16, 8)
  double stat_samples = 0;             // samples per picture of that batch
  int ld_batch = 1;         // pictures of the LD batch being encoded (fill_ld_enc sizes the scratch array with it)
  bool allow_heads = true;  // record heads for the levels below them (A/B and test switch VC2HIP_NO_HEADS)
  bool allow_cbr_index = true; // decode of HQ_CBR pictures: offsets from the budgets, verified (VC2HIP_NO_CBR_INDEX=1: always the general index)
  bool allow_stream = true;   // VC2HIP_NO_STREAM=1: tile kernels instead of the streaming level kernels (tests, A/B)
  bool allow_pair = true;     // VC2HIP_FLAG_NO_PAIR: one launch per transform level (vc2hip_dwt_pair.hip off; tests, A/B)
  bool cbr_general = false;   // VC2HIP_FLAG_CBR_GENERAL: the HQ_CBR search without the register kernels
  bool ld_diagonals = false;  // VC2HIP_FLAG_LD_DIAGONALS: the LD index search with one launch per slice anti-diagonal
  unsigned flags = 0;
  // vc2hip_set_streams(k > 1): device-resident batches are cut into k contiguous sub-batches, each on its own
  // stream and workspace (a child context), forked from / joined to `stream` with events.  The kernels of the
  // sub-batches overlap: the tail of one launch is filled by the next stream's work.
  std::vector<vc2hip_ctx *> lanes;    // lanes[0] is the context itself (its own stream), the others are children
  bool in_split = false;              // the context is running its own sub-batch as lane 0
  hipEvent_t fork_ev = nullptr;
  std::vector<hipEvent_t> join_ev;    // end of lane i's latest sub-batch
  bool lanes_pending = false;         // `stream` has not yet been made to wait for the lanes' latest sub-batches
  struct Range { const uint8_t *lo, *hi; };
  struct LaneUse { Range r[2], w[2]; bool valid = false; };
  std::vector<LaneUse> lane_use;      // what lane i's latest sub-batch read and wrote (caller buffers)
  std::vector<ProfEntry> merged; // profile of this context and its lanes, rebuilt by vc2hip_profile_count
  // pipelined picture calls: VC2HIP_MAX_INFLIGHT slots, each a child context (own stream and workspace)
  struct Flight {
    vc2hip_ctx *lane = nullptr;
    bool open = false, encode = false;
    uint8_t *payload = nullptr;      // caller's pinned buffer (encode)
    size_t cap = 0;
    unsigned long long *h_len = nullptr; // pinned
  };
  Flight flight[VC2HIP_MAX_INFLIGHT];
  int flight_next = 0;               // slot of the next _begin
};

static const char *code_text(int code) {
  switch (code) {
    case VC2HIP_OK: return "ok";
    case VC2HIP_EINVAL: return "invalid argument";
    case VC2HIP_EQINDEX: return "quantization index exceeds maximum implemented value.";
    case VC2HIP_ESCALAR: return "Slice scalar is too small, consider using a larger slice scalar.";
    case VC2HIP_ECBR_TOOBIG: return "SliceIO, HQ CBR mode: Too many bytes for the slice";
    case VC2HIP_ECBR_LEN: return "Slice component length exceeds 1 byte when divided by slice size scalar. See above for suggestions to prevent this.";
    case VC2HIP_ECBR_WRONG: return "SliceIO, HQ CBR mode: Wrong number of bytes for a slice";
    case VC2HIP_EBOUNDED: return "Attempt to write beyond end of bounded write";
    case VC2HIP_ELD_TOOBIG: return "SliceIO, LD mode: Too many bytes for the U and V slices";
    case VC2HIP_ECAP: return "output buffer too small";
    case VC2HIP_ESTREAM: return "truncated or malformed slice data";
    case VC2HIP_ECODE32: return "quantised coefficient magnitude exceeds 65534 (outside the 32-bit exp-Golomb code domain)";
    case VC2HIP_EHIP: return "HIP runtime error";
  }
  return "unknown error";
}
extern "C" const char *vc2hip_error_string(int code) { return code_text(code); }

static int set_err(vc2hip_ctx *c, int code, const char *msg = nullptr) {
  if (c) c->err = msg ? msg : code_text(code);
  return code;
}
static int hip_fail(vc2hip_ctx *c, hipError_t e, const char *what) {
  char b[256];
  snprintf(b, sizeof b, "HIP error in %s: %s", what, hipGetErrorString(e));
  return set_err(c, VC2HIP_EHIP, b);
}
#define HIPCHK(ctx, call)                                       \
  do {                                                          \
    hipError_t e_ = (call);                                     \
    if (e_ != hipSuccess) return hip_fail(ctx, e_, #call);      \
  } while (0)

extern "C" const char *vc2hip_last_error(const vc2hip_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

// ---- host-side arithmetic shared with the bindings -----------------------------------------
extern "C" int vc2hip_padded_size(int size, int depth) {
  const int cell = 1 << depth;
  return cell * ((size + cell - 1) / cell);
}
extern "C" int vc2hip_slice_size_is_valid(int depth, int len_luma, int len_chroma, int n_size) {
  if (depth <= 0 || depth > 31) return 0;
  const int unit = 1 << depth;
  const int max_slices = (len_luma < len_chroma ? len_luma : len_chroma) / unit;
  if (n_size <= 0 || n_size > max_slices) return 0;
  const int ts = n_size * unit;
  const int pl = vc2hip_padded_size(len_luma, depth), pc = vc2hip_padded_size(len_chroma, depth);
  const int n = (pl + ts - 1) / ts;
  if (pl % n == 0 && (pl / n) % unit == 0 && pc % n == 0 && (pc / n) % unit == 0) return n;
  return 0;
}
// WaveletTransform.cpp:345-423: float variables, double pow, float log/floor (math.h overloads)
extern "C" int vc2hip_quant_matrix(int kernel, int depth, int32_t *out) {
  static const float A[7] = {1.280868846f, 1.224744871f, 1.280868846f, 1.414213562f, 1.414213562f, 0.682408629f, 1.139917028f};
  static const float B[7] = {0.820572875f, 0.847791248f, 0.809253958f, 0.707106871f, 0.707106871f, 1.367856979f, 0.887168005f};
  static const int S[7] = {1, 1, 1, 0, 1, 0, 1};
  if (kernel < 0 || kernel > 6 || depth < 0 || depth > 30) return VC2HIP_EINVAL;
  if (depth == 0) { out[0] = 0; return 0; }
  const float a2 = A[kernel] * A[kernel], ab = A[kernel] * B[kernel], b2 = B[kernel] * B[kernel];
  std::vector<float> gl(depth + 1), gh(depth + 1), gd(depth + 1);
  float mn = 3.402823466e+38f;
  for (int lv = depth; lv > 0; --lv) {
    const float sc = (float)(pow((double)a2, depth - lv) / pow(2.0, S[kernel] * (depth - lv + 1)));
    gl[lv] = sc * a2; gh[lv] = sc * ab; gd[lv] = sc * b2;
    mn = fminf(fminf(fminf(gl[lv], gh[lv]), gd[lv]), mn);
  }
  auto qv = [&](float g) { return (int)floorf(4.0f * logf(g / mn) / logf(2.0f) + 0.5f); };
  int i = 0;
  out[i++] = qv(gl[1]);
  for (int lv = 1; lv <= depth; ++lv) { out[i++] = qv(gh[lv]); out[i++] = qv(gh[lv]); out[i++] = qv(gd[lv]); }
  return 0;
}
void vc2_allow_lds(const void *kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void *, int>, size_t> done;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  size_t &have = done[std::make_pair(kernel, dev)];
  if (bytes > have) {
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    have = bytes;
  }
}

static int gcd_i(int a, int b) { a = abs(a); b = abs(b); while (b) { int t = a % b; a = b; b = t; } return a; }
extern "C" int vc2hip_slice_bytes(int ys, int xs, int total_bytes, int scalar, int32_t *out) {
  if (ys < 1 || xs < 1 || scalar < 1) return VC2HIP_EINVAL;
  const int n = ys * xs;
  int num = total_bytes / scalar - 4 * n, den = n;
  const int g = gcd_i(num, den);
  if (g) { num /= g; den /= g; }
  const int ratio = num / den, rem = num - ratio * den;
  int residue = 0;
  for (int i = 0; i < n; ++i) {
    residue += rem;
    if (residue < den) out[i] = ratio * scalar + 4;
    else { out[i] = (ratio + 1) * scalar + 4; residue -= den; }
  }
  return 0;
}
// SMPTE 2042-1 quant_factor / quant_offset (the reference holds them as a table, Quantisation.cpp:40-83)
static void make_tables(QuantTables &t) {
  for (int q = 0; q < 120; ++q) {
    const uint64_t base = 1ull << (q / 4);
    uint64_t f;
    switch (q % 4) {
      case 0: f = 4 * base; break;
      case 1: f = (503829 * base + 52958) / 105917; break;
      case 2: f = (665857 * base + 58854) / 117708; break;
      default: f = (440253 * base + 32722) / 65444; break;
    }
    t.qf[q] = (int32_t)(uint32_t)f;
    t.off[q] = q == 0 ? 1 : (q == 1 ? 2 : (int32_t)(((uint32_t)t.qf[q] + 1u)) / 2);
    t.magic[q] = 0;
    t.shift[q] = 0;
    if (t.qf[q] > 1) {
      const uint64_t d = (uint64_t)t.qf[q];
      int l = 0;
      while ((1ull << l) < d) ++l;
      t.magic[q] = (uint32_t)((((1ull << l) - d) << 32) / d + 1);
      t.shift[q] = l - 1;
    }
    const double r = 4.0 / (double)(uint32_t)t.qf[q];
    float up = (float)r;
    if ((double)up < r) up = nextafterf(up, INFINITY);
    t.inv4[q] = up;
  }
}

// ---- lifetime ---------------------------------------------------------------------------------
static int create_common(int device, hipStream_t stream, bool own, vc2hip_ctx **out, unsigned flags = 0) {
  if (!out) return VC2HIP_EINVAL;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return VC2HIP_EHIP;
  vc2hip_ctx *c = new vc2hip_ctx;
  c->device = device;
  // the switches between two correct paths (vc2hip_create_with_flags); only the ablation build (tools/, never the product
  // path) also takes them from the environment, so that its A/B runs need no code
#ifdef VC2HIP_ABLATE
  {
    static const struct { const char *name; unsigned flag; } env[] = {
      {"VC2HIP_STORE32", VC2HIP_FLAG_STORE32}, {"VC2HIP_NO_STREAM", VC2HIP_FLAG_NO_STREAM}, {"VC2HIP_NO_PAIR", VC2HIP_FLAG_NO_PAIR},
      {"VC2HIP_NO_BANDPLANES", VC2HIP_FLAG_NO_BANDPLANES}, {"VC2HIP_NO_HEADS", VC2HIP_FLAG_NO_HEADS},
      {"VC2HIP_NO_CBR_INDEX", VC2HIP_FLAG_NO_CBR_INDEX}, {"VC2HIP_GENERIC_DWT", VC2HIP_FLAG_GENERIC_DWT},
      {"VC2HIP_SINGLE_PASS_VBR", VC2HIP_FLAG_SINGLE_PASS_VBR}, {"VC2HIP_TWO_PASS_VBR", VC2HIP_FLAG_TWO_PASS_VBR}, {"VC2HIP_CBR_GENERAL", VC2HIP_FLAG_CBR_GENERAL},
      {"VC2HIP_PLANES8_ALWAYS", VC2HIP_FLAG_PLANES8_ALWAYS}, {"VC2HIP_PLANES8_NEVER", VC2HIP_FLAG_PLANES8_NEVER}};
    for (const auto &e : env) { const char *v = getenv(e.name); if (v && v[0] == '1') flags |= e.flag; }
    { const char *v = getenv("VC2HIP_LD_ROWS"); if (v && v[0] == '0') flags |= VC2HIP_FLAG_LD_DIAGONALS; }
    { const char *e = getenv("VC2HIP_DEBUG_SKIP"); c->debug_skip = e ? atoi(e) : 0; }
  }
#endif
  c->force_generic = (flags & VC2HIP_FLAG_GENERIC_DWT) != 0;
  c->allow_store16 = !(flags & VC2HIP_FLAG_STORE32);
  c->allow_stream = !(flags & VC2HIP_FLAG_NO_STREAM);
  c->allow_pair = !(flags & VC2HIP_FLAG_NO_PAIR);
  c->allow_planes = !(flags & VC2HIP_FLAG_NO_BANDPLANES);
  c->planes8_mode = (flags & VC2HIP_FLAG_PLANES8_NEVER) ? 2 : (flags & VC2HIP_FLAG_PLANES8_ALWAYS) ? 1 : 0;
  c->allow_heads = !(flags & VC2HIP_FLAG_NO_HEADS);
  c->allow_cbr_index = !(flags & VC2HIP_FLAG_NO_CBR_INDEX);
  c->two_pass_vbr = !(flags & VC2HIP_FLAG_SINGLE_PASS_VBR);
  c->force_two_pass_vbr = (flags & VC2HIP_FLAG_TWO_PASS_VBR) != 0;
  c->cbr_general = (flags & VC2HIP_FLAG_CBR_GENERAL) != 0;
  c->ld_diagonals = (flags & VC2HIP_FLAG_LD_DIAGONALS) != 0;
  c->flags = flags;
  if (hipSetDevice(device) != hipSuccess) { delete c; return VC2HIP_EHIP; }
  if (own) { if (hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) != hipSuccess) { delete c; return VC2HIP_EHIP; } }
  c->stream = stream;
  c->own_stream = own;
  if (hipMalloc((void **)&c->d_err, 256 + 4096) != hipSuccess || // error word, then the LD search tables
     
      hipHostMalloc((void **)&c->h_err, sizeof(unsigned)) != hipSuccess ||
      hipMalloc((void **)&c->d_stat, 64) != hipSuccess || hipHostMalloc((void **)&c->h_stat, 64 * 8) != hipSuccess ||
      hipEventCreateWithFlags(&c->stat_ev, hipEventDisableTiming) != hipSuccess) { delete c; return VC2HIP_EHIP; }
  (void)hipMemsetAsync(c->d_stat, 0, 64, c->stream);
  (void)hipMemsetAsync(c->d_err, 0, sizeof(unsigned), c->stream);
  QuantTables t;
  make_tables(t);
  vc2_upload_tables(t, c->stream);
  vc2_upload_tables_slices(t, c->stream);
  vc2_upload_tables_fast(t, c->stream);
  vc2_upload_tables_pair(t, c->stream);
  vc2_upload_tables_stream(t, c->stream);
  vc2_upload_vlc_lut(c->stream);
  vc2_upload_unpack_lut(c->stream);
  if (hipStreamSynchronize(c->stream) != hipSuccess) { delete c; return VC2HIP_EHIP; }
  *out = c;
  return VC2HIP_OK;
}
extern "C" int vc2hip_create(int device, vc2hip_ctx **out) { return create_common(device, nullptr, true, out); }
extern "C" int vc2hip_create_with_flags(int device, unsigned flags, vc2hip_ctx **out) { return create_common(device, nullptr, true, out, flags); }
extern "C" int vc2hip_create_on_stream(int device, void *s, vc2hip_ctx **out) { return create_common(device, (hipStream_t)s, false, out); }
extern "C" void vc2hip_destroy(vc2hip_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (vc2hip_ctx *l : c->lanes) if (l != c) vc2hip_destroy(l);
  for (auto &f : c->flight) { if (f.lane) vc2hip_destroy(f.lane); if (f.h_len) (void)hipHostFree(f.h_len); }
  for (hipEvent_t e : c->join_ev) (void)hipEventDestroy(e);
  if (c->fork_ev) (void)hipEventDestroy(c->fork_ev);
  c->L.collect();
  for (auto e : c->L.pool) (void)hipEventDestroy(e);
  for (auto &b : c->buf) if (b.p) (void)hipFree(b.p);
  if (c->d_err) (void)hipFree(c->d_err);
  if (c->h_err) (void)hipHostFree(c->h_err);
  if (c->h_stat) (void)hipHostFree(c->h_stat);
  if (c->d_stat) (void)hipFree(c->d_stat);
  if (c->stat_ev) (void)hipEventDestroy(c->stat_ev);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

void vc2_ld_disable_rows();
static int err_from_flags(vc2hip_ctx *c, unsigned f) {
  if (!f) return VC2HIP_OK;
  if (f & VC2_DEVERR_HANDOFF) { // not a property of the input: the batch has to be submitted again
    vc2_ld_disable_rows();
    return set_err(c, VC2HIP_EHIP, "LD index search: a hand-over between workgroups timed out; nothing was written for the batch. "
                                   "The library now searches with one launch per slice diagonal: submit the batch again.");
  }
  if (f & VC2_DEVERR_QINDEX) return set_err(c, VC2HIP_EQINDEX);
  if (f & VC2_DEVERR_SCALAR) return set_err(c, VC2HIP_ESCALAR);
  if (f & VC2_DEVERR_CBR_TOOBIG) return set_err(c, VC2HIP_ECBR_TOOBIG);
  if (f & VC2_DEVERR_CBR_LEN) return set_err(c, VC2HIP_ECBR_LEN);
  if (f & VC2_DEVERR_CODE32) return set_err(c, VC2HIP_ECODE32);
  if (f & VC2_DEVERR_LD_TOOBIG) return set_err(c, VC2HIP_ELD_TOOBIG);
  return set_err(c, VC2HIP_ESTREAM);
}
// make the context's stream wait for whatever its lanes still have in flight
static int join_lanes(vc2hip_ctx *c) {
  if (!c->lanes_pending || c->in_split) return VC2HIP_OK;
  for (size_t i = 1; i < c->lanes.size(); ++i)
    if (c->lane_use[i].valid) HIPCHK(c, hipStreamWaitEvent(c->stream, c->join_ev[i], 0));
  c->lanes_pending = false;
  return VC2HIP_OK;
}
// every entry point that enqueues on the context's stream starts here
static int enter(vc2hip_ctx *c) {
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->in_split) vc2_prof_break(c->L); // whatever the caller enqueued since is not part of the next kernel
  return join_lanes(c);
}
#define ENTER(ctx) do { int rc_ = enter(ctx); if (rc_) return rc_; } while (0)

extern "C" int vc2hip_set_streams(vc2hip_ctx *c, int k) {
  if (!c || k < 1 || k > 16) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (vc2hip_ctx *l : c->lanes) if (l != c) vc2hip_destroy(l);
  c->lanes.clear();
  c->lane_use.clear();
  for (hipEvent_t e : c->join_ev) (void)hipEventDestroy(e);
  c->join_ev.clear();
  if (k == 1) return VC2HIP_OK;
  if (!c->fork_ev) HIPCHK(c, hipEventCreateWithFlags(&c->fork_ev, hipEventDisableTiming));
  for (int i = 0; i < k; ++i) {
    vc2hip_ctx *l = c;
    if (i > 0) {
      const int rc = vc2hip_create_with_flags(c->device, c->flags, &l);
      if (rc) return set_err(c, rc);
      l->L.on = c->L.on;
    }
    c->lanes.push_back(l);
    hipEvent_t e;
    HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->join_ev.push_back(e);
  }
  c->lane_use.assign((size_t)k, vc2hip_ctx::LaneUse());
  return VC2HIP_OK;
}

// Run fn(lane, first picture, picture count) for contiguous sub-batches on the lanes' streams.  use(first, count,
// LaneUse&) names the caller buffers a sub-batch reads and writes.  A lane waits for the context's stream (fork)
// and for every other lane whose previous sub-batch touched what it is about to touch; consecutive calls with the
// same partition (encode -> decode of the same batch) therefore chain lane by lane without a barrier.  The
// context's stream joins the lanes lazily: at the next entry point that uses it, or at once when the stream
// belongs to the caller (vc2hip_create_on_stream).
template <class U, class F> static int split_batch(vc2hip_ctx *c, int n, U use, F fn) {
  const int k = std::min<int>((int)c->lanes.size(), n);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventRecord(c->fork_ev, c->stream));
  std::vector<vc2hip_ctx::LaneUse> cur((size_t)c->lanes.size());
  std::vector<int> first((size_t)k), count((size_t)k);
  auto overlap = [](const vc2hip_ctx::Range &a, const vc2hip_ctx::Range &b) { return a.lo && b.lo && a.lo < b.hi && b.lo < a.hi; };
  for (int i = 0, f0 = 0; i < k; ++i) {
    count[i] = n / k + (i < n % k ? 1 : 0);
    first[i] = f0;
    f0 += count[i];
    use(first[i], count[i], cur[i]);
    cur[i].valid = true;
  }
  for (int i = 0; i < k; ++i) { // all waits before any join event is recorded again
    vc2hip_ctx *l = c->lanes[i];
    if (i > 0) HIPCHK(c, hipStreamWaitEvent(l->stream, c->fork_ev, 0));
    for (size_t j = 0; j < c->lanes.size(); ++j) {
      const vc2hip_ctx::LaneUse &p = c->lane_use[j];
      if ((int)j == i || !p.valid) continue;
      bool dep = false;
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          dep |= overlap(cur[i].r[a], p.w[b]) || overlap(cur[i].w[a], p.w[b]) || overlap(cur[i].w[a], p.r[b]);
      if (dep) HIPCHK(c, hipStreamWaitEvent(l->stream, c->join_ev[j], 0));
    }
  }
  int rc = VC2HIP_OK;
  for (int i = 0; i < k; ++i) {
    vc2hip_ctx *l = c->lanes[i];
    c->in_split = (i == 0);
    const int r = fn(l, first[i], count[i]);
    c->in_split = false;
    if (r && !rc) rc = i ? set_err(c, r, l->err.c_str()) : r;
    HIPCHK(c, hipEventRecord(c->join_ev[i], l->stream));
    c->lane_use[i] = cur[i];
  }
  c->lanes_pending = true;
  if (!c->own_stream) return join_lanes(c) ? VC2HIP_EHIP : rc;
  return rc;
}

extern "C" int vc2hip_sync(vc2hip_ctx *c) {
  if (!c) return VC2HIP_EINVAL;
  ENTER(c);
  HIPCHK(c, hipMemcpyAsync(c->h_err, c->d_err, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemsetAsync(c->d_err, 0, sizeof(unsigned), c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->L.collect();
  int lane_rc = VC2HIP_OK;
  for (vc2hip_ctx *l : c->lanes) { // their work was joined into c->stream; surface their error flags too
    if (l == c) continue;
    const int r = vc2hip_sync(l);
    if (r && !lane_rc) lane_rc = set_err(c, r, l->err.c_str());
  }
  if (!c->L.launch_error.empty()) {
    const std::string m = c->L.launch_error;
    c->L.launch_error.clear();
    return set_err(c, VC2HIP_EHIP, m.c_str());
  }
  const int rc = err_from_flags(c, *c->h_err);
  return rc ? rc : lane_rc;
}

static int need(vc2hip_ctx *c, int which, size_t bytes, void **out) {
  Buf &b = c->buf[which];
  if (b.cap < bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    // Large workspaces are sized in whole 2 MiB units (the driver's large page-table fragments; no measurable effect on
    // the kernels: the 10 % spread of the level-0 transforms between processes that prompted it follows the box and the
    // moment, not the allocation -- see VC2_STREAM_WG_WAVES in vc2hip_dwt_stream.hip).
    size_t cap = bytes >= (1u << 21) ? (bytes + (1u << 21) - 1) & ~(size_t)((1u << 21) - 1) : (bytes + 4095) & ~(size_t)4095;
    { static const size_t unit = (size_t)vc2_tune_int("VC2HIP_ALLOC_ROUND_MB", 0) << 20; // (ablation build: another unit)
      if (unit && bytes >= (1u << 21)) cap = (bytes + unit - 1) / unit * unit; }
    HIPCHK(c, hipMalloc(&b.p, cap));
    b.cap = cap;
#ifdef VC2HIP_ABLATE
    if (getenv("VC2HIP_DEBUG_ALLOC")) fprintf(stderr, "vc2hip alloc: buffer %d at %p, %zu bytes\n", which, b.p, cap);
#endif
  }
  *out = b.p;
  return VC2HIP_OK;
}
#define NEED(ctx, which, bytes, ptr)                                   \
  do {                                                                 \
    void *p_;                                                          \
    int rc_ = need(ctx, which, bytes, &p_);                            \
    if (rc_) return rc_;                                               \
    ptr = (decltype(ptr))p_;                                           \
  } while (0)

// ---- profiling ---------------------------------------------------------------------------------
extern "C" int vc2hip_profile_enable(vc2hip_ctx *c, int on) {
  if (!c) return VC2HIP_EINVAL;
  c->L.on = on != 0;
  for (vc2hip_ctx *l : c->lanes) l->L.on = on != 0;
  return 0;
}
extern "C" int vc2hip_profile_only(vc2hip_ctx *c, const char *name) {
  if (!c) return VC2HIP_EINVAL;
  c->L.only = name ? name : "";
  for (vc2hip_ctx *l : c->lanes) l->L.only = c->L.only;
  return 0;
}
// entries of the context and of its lanes, merged by kernel name
extern "C" int vc2hip_profile_count(vc2hip_ctx *c) {
  if (!c) return 0;
  c->merged = c->L.entries;
  for (vc2hip_ctx *l : c->lanes)
    if (l != c) for (const ProfEntry &e : l->L.entries) {
      bool found = false;
      for (ProfEntry &m : c->merged) if (m.name == e.name) { m.launches += e.launches; m.ms += e.ms; found = true; break; }
      if (!found) c->merged.push_back(e);
    }
  return (int)c->merged.size();
}
extern "C" int vc2hip_profile_get(vc2hip_ctx *c, int i, const char **name, int *launches, double *total_ms) {
  if (!c || i < 0 || i >= (int)c->merged.size()) return VC2HIP_EINVAL;
  if (name) *name = c->merged[i].name.c_str();
  if (launches) *launches = c->merged[i].launches;
  if (total_ms) *total_ms = c->merged[i].ms;
  return 0;
}
extern "C" int vc2hip_profile_reset(vc2hip_ctx *c) {
  if (!c) return VC2HIP_EINVAL;
  c->L.collect(); c->L.entries.clear(); c->merged.clear();
  for (vc2hip_ctx *l : c->lanes) if (l != c) { l->L.collect(); l->L.entries.clear(); }
  return 0;
}

// ------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------
static int make_geom(Geom &g, const int ph[3], const int pw[3], const int h[3], const int w[3], int depth,
                     int ys, int xs) {
  if (depth < 1 || depth > VC2_MAX_DEPTH || ys < 1 || xs < 1) return VC2HIP_EINVAL;
  g.depth = depth; g.ys = ys; g.xs = xs;
  int off = 0;
  for (int c = 0; c < 3; ++c) {
    CompGeom &cg = g.c[c];
    cg.h = h[c]; cg.w = w[c]; cg.ph = ph[c]; cg.pw = pw[c];
    if (ph[c] == 0 || pw[c] == 0) { cg.sh = cg.sw = cg.n0 = 0; cg.coef_off = off; continue; }
    if (ph[c] % ys || pw[c] % xs) return VC2HIP_EINVAL;
    cg.sh = ph[c] / ys; cg.sw = pw[c] / xs;
    if (cg.sh % (1 << depth) || cg.sw % (1 << depth)) return VC2HIP_EINVAL;
    cg.n0 = (cg.sh >> depth) * (cg.sw >> depth);
    cg.coef_off = off;
    off += cg.sh * cg.sw;
  }
  g.slice_coefs = off;
  return VC2HIP_OK;
}

static void chroma_dims(int h, int w, int cf, int *ch, int *cw) {
  *ch = cf == VC2HIP_CF420 ? h / 2 : h;
  *cw = cf == VC2HIP_CF444 ? w : w / 2;
}

// encoder-side picture geometry (WaveletTransform.cpp:1267-1273) or decoder-side
// (DecodeStream.cpp:483-498: chroma derived from the padded luma size)
static int picture_geom(Geom &g, const vc2hip_picture_format *f, const vc2hip_coding_params *cp, bool decoder) {
  if (!f || !cp || f->width < 1 || f->height < 1 || f->word_bytes < 1 || f->word_bytes > 4 ||
      f->bit_depth < 1 || f->bit_depth > 8 * f->word_bytes || f->chroma_format < 0 || f->chroma_format > 2 ||
      f->chroma_bit_depth < 0 || f->chroma_bit_depth > 8 * f->word_bytes)
    return VC2HIP_EINVAL;
  int ch, cw;
  chroma_dims(f->height, f->width, f->chroma_format, &ch, &cw);
  int h[3] = {f->height, ch, ch}, w[3] = {f->width, cw, cw}, ph[3], pw[3];
  ph[0] = vc2hip_padded_size(f->height, cp->depth);
  pw[0] = vc2hip_padded_size(f->width, cp->depth);
  if (decoder) chroma_dims(ph[0], pw[0], f->chroma_format, &ph[1], &pw[1]);
  else { ph[1] = vc2hip_padded_size(ch, cp->depth); pw[1] = vc2hip_padded_size(cw, cp->depth); }
  ph[2] = ph[1]; pw[2] = pw[1];
  return make_geom(g, ph, pw, h, w, cp->depth, cp->y_slices, cp->x_slices);
}

extern "C" size_t vc2hip_raw_picture_bytes(const vc2hip_picture_format *f) {
  int ch, cw;
  chroma_dims(f->height, f->width, f->chroma_format, &ch, &cw);
  return ((size_t)f->height * f->width + 2 * (size_t)ch * cw) * f->word_bytes;
}
static size_t max_slice_bytes(int prefix, int scalar) { return (size_t)prefix + 4 + 3 * 255 * (size_t)scalar; }
extern "C" size_t vc2hip_max_payload_bytes(const vc2hip_picture_format *f, const vc2hip_coding_params *cp) {
  (void)f;
  const size_t n = (size_t)cp->y_slices * cp->x_slices;
  if (cp->mode == VC2HIP_HQ_CONSTQ) return n * max_slice_bytes(cp->prefix, cp->scalar);
  if (cp->mode == VC2HIP_HQ_CBR) return (size_t)cp->compressed_bytes + n * (cp->prefix + (size_t)cp->scalar + 4);
  return (size_t)cp->compressed_bytes + n;
}

// ------------------------------------------------------------------------------------------
// level sequencing
// ------------------------------------------------------------------------------------------
struct LLPlanes { // per level l >= 1: compact LL_l planes for the three components
  void *p[VC2_MAX_DEPTH + 1][3];    // int32_t elements, or int16_t with ...
  int32_t *w[VC2_MAX_DEPTH + 1][3]; // ... their wide planes (vc2hip_store.h); null for int32 planes
  long long stride[VC2_MAX_DEPTH + 1][3];
};

// planes start on 16-byte boundaries whatever the element size: sizes are rounded up to 8 elements
static size_t ll_elems(const Geom &g, int n) {
  size_t e = 0;
  for (int l = 1; l <= g.depth; ++l)
    for (int c = 0; c < 3; ++c) e += (((size_t)(g.c[c].ph >> l) * (g.c[c].pw >> l) * n + 7) & ~(size_t)7);
  return e;
}
static size_t ll_bytes(const Geom &g, int n) { return ll_elems(g, n) * sizeof(int32_t); }
static void ll_layout(const Geom &g, int n, void *base, int elem_bytes, int32_t *wide, LLPlanes &ll) {
  size_t off = 0;
  memset(&ll, 0, sizeof ll);
  for (int l = 1; l <= g.depth; ++l)
    for (int c = 0; c < 3; ++c) {
      const size_t e = (size_t)(g.c[c].ph >> l) * (g.c[c].pw >> l);
      ll.p[l][c] = (char *)base + off * elem_bytes;
      ll.w[l][c] = wide ? wide + off : nullptr;
      ll.stride[l][c] = (long long)e;
      off += (e * n + 7) & ~(size_t)7;
    }
}
static void ll_layout(const Geom &g, int n, int32_t *base, LLPlanes &ll) { ll_layout(g, n, base, 4, nullptr, ll); }

static void fill_level(LevelParams &p, const Geom &g, int level, int kernel, const int32_t *qm) {
  const int D = g.depth, Lv = D - level;
  p.band = 3 * (Lv - 1) + 1;
  p.ys = g.ys; p.xs = g.xs; p.slice_coefs = g.slice_coefs;
  const int hx = vc2_halo_x(kernel), hy = vc2_halo_y(kernel);
  for (int c = 0; c < 3; ++c) {
    const CompGeom &cg = g.c[c];
    p.in_h[c] = cg.ph >> level; p.in_w[c] = cg.pw >> level;
    p.pic_h[c] = cg.h; p.pic_w[c] = cg.w;
    p.coef_off[c] = cg.coef_off;
    p.rec_stride[c] = g.slice_coefs;
    if (cg.ph == 0) { p.tiles_x[c] = p.tiles_y[c] = 0; p.fh[c] = p.fw[c] = 2; p.tsy[c] = p.tsx[c] = 1; continue; }
    p.fh[c] = cg.sh >> level; p.fw[c] = cg.sw >> level;
    // tile 32 x 128 samples, whole slices, power-of-two slice counts, LDS <= 64 KiB
    int tsy = 1, tsx = 1;
    while (tsy * 2 * p.fh[c] <= 64 && tsy * 2 <= g.ys) tsy *= 2;
    while (tsx * 2 * p.fw[c] <= 128 && tsx * 2 <= g.xs) tsx *= 2;
    while ((size_t)(tsy * p.fh[c] + 2 * hy) * (tsx * p.fw[c] + 2 * hx) * 4 > 150 * 1024) {
      if (tsx > 1) tsx /= 2; else if (tsy > 1) tsy /= 2; else break;
    }
    p.tsy[c] = tsy; p.tsx[c] = tsx;
    p.tiles_y[c] = (g.ys + tsy - 1) / tsy;
    p.tiles_x[c] = (g.xs + tsx - 1) / tsx;
    p.band_n[c] = (p.fh[c] / 2) * (p.fw[c] / 2);
    p.band_off[c] = p.band_n[c]; // == n0 * 4^(Lv-1): HL at 1x, LH at 2x, HH at 3x
  }
  for (int b = 0; b < 3 * D + 1; ++b) p.qmatrix[b] = qm ? qm[b] : 0;
}

// forward transform of n pictures: raw words (first level fused) or int32 LL_0 planes -> store
// s16: 16-bit store and level planes with their wide planes (only with the fast kernels: use_store16)
static int run_forward(vc2hip_ctx *c, const Geom &g, int kernel, int n, const void *const src[3],
                       const long long src_stride[3], bool src_raw, const vc2hip_picture_format *f,
                       void *store, const LLPlanes &ll, bool s16 = false, int32_t *store_wide = nullptr) {
  auto level_params = [&](LevelParams &p, int level) {
    memset(&p, 0, sizeof p);
    fill_level(p, g, level, kernel, nullptr);
    p.store = store; p.store_stride = (long long)g.ys * g.xs * g.slice_coefs;
    p.store_wide = store_wide;
    p.err = c->d_err;
    p.ll_to_store = (level == g.depth - 1);
    const bool first = (level == 0) && src_raw;
    for (int k = 0; k < 3; ++k) {
      if (level == 0) { p.plane[k] = (void *)src[k]; p.plane_stride[k] = src_stride[k]; }
      else { p.plane[k] = ll.p[level][k]; p.plane_wide[k] = ll.w[level][k]; p.plane_stride[k] = ll.stride[level][k]; }
      p.ll[k] = ll.p[level + 1][k]; p.ll_wide[k] = ll.w[level + 1][k]; p.ll_stride[k] = ll.stride[level + 1][k];
    }
    if (first) {
      p.word_bytes = f->word_bytes;
      p.sample_shift = 8 * f->word_bytes - f->bit_depth;
      p.sample_offset = 1 << (f->bit_depth - 1);
      const int cd = f->chroma_bit_depth ? f->chroma_bit_depth : f->bit_depth;
      p.sample_shift_c = 8 * f->word_bytes - cd;
      p.sample_offset_c = 1 << (cd - 1);
    }
  };
  for (int level = 0; level < g.depth; ++level) {
    LevelParams p;
    level_params(p, level);
    const bool first = (level == 0) && src_raw;
    // two levels in one launch (vc2hip_dwt_pair.hip): level + 1's input plane is never written
    if (!c->force_generic && c->allow_stream && c->allow_pair && level + 1 < g.depth) {
      PairParams pp;
      memset(&pp, 0, sizeof pp);
      pp.a = p;
      pp.a.debug_skip = c->debug_skip;
      level_params(pp.b, level + 1);
      const size_t lds = vc2_pair_applicable(pp, kernel, first, false, s16, n);
      if (lds) {
        int rc = vc2_launch_forward_pair(c->L, kernel, first, pp, n, s16, lds, c->stream);
        if (rc) return set_err(c, rc, "invalid wavelet kernel");
        ++level;
        continue;
      }
    }
    LevelParams pf = p;
    pf.debug_skip = c->debug_skip;
    if (!c->force_generic && c->allow_stream) {
      LevelParams ps = p;
      const size_t lds = vc2_stream_level_applicable(ps, kernel, first, false, s16, n);
      if (lds) {
        int rc = vc2_launch_forward_stream(c->L, kernel, first, ps, n, s16, lds, c->stream);
        if (rc) return set_err(c, rc, "invalid wavelet kernel");
        continue;
      }
    }
    if (!c->force_generic && vc2_fast_level_applicable(pf)) {
      int rc = vc2_launch_forward_fast(c->L, kernel, first, pf, n, s16, c->stream);
      if (rc) return set_err(c, rc, "invalid wavelet kernel");
      continue;
    }
    if (s16) return set_err(c, VC2HIP_EINVAL, "internal: 16-bit store without the fast level kernels");
    if (vc2_level_lds_bytes(kernel, p) > 160 * 1024) return set_err(c, VC2HIP_EINVAL, "slice too large for one LDS tile");
    int rc = vc2_launch_forward_level(c->L, kernel, first, p, n, c->stream);
    if (rc) return set_err(c, rc, "invalid wavelet kernel");
  }
  return VC2HIP_OK;
}

// inverse transform of n pictures: store (optionally dequantised on load) -> int32 planes or raw words
// bp: the decoder's band planes (levels < bp->levels read their bands from them; the store stride then includes them).
// stream_mask != nullptr: a dry run -- nothing is launched, bit l of *stream_mask says whether level l would go through
// the streaming kernel (the planning step of the band planes asks this before the slices are decoded).
static int run_inverse(vc2hip_ctx *c, const Geom &g, int kernel, int n, void *store, const int32_t *qidx,
                       const int32_t *qm, bool dequant, bool ll_ready, const LLPlanes &ll, void *const dst[3],
                       const long long dst_stride[3], bool dst_raw, const vc2hip_picture_format *f,
                       bool s16 = false, int32_t *store_wide = nullptr, const BandPlanes *bp = nullptr,
                       long long store_stride = 0, unsigned *stream_mask = nullptr, const HeadSplit *hs = nullptr, int head_level = 1 << 30,
                       unsigned *fast_mask = nullptr, unsigned *tail_mask = nullptr) {
  if (tail_mask) *tail_mask = 0;
  if (stream_mask) *stream_mask = 0;
  if (fast_mask) *fast_mask = 0;
  auto level_params = [&](LevelParams &p, int level) {
    memset(&p, 0, sizeof p);
    fill_level(p, g, level, kernel, qm);
    if (hs && level >= head_level) // the deep levels' coefficients live in the record heads (HeadSplit)
      for (int k = 0; k < 3; ++k) if (g.c[k].ph) { p.rec_stride[k] = hs->n[k]; p.coef_off[k] = (int)hs->base[k]; }
    p.store = store; p.store_stride = store_stride ? store_stride : (long long)g.ys * g.xs * g.slice_coefs;
    for (int k = 0; k < 3; ++k) p.bp_base[k] = (bp && level < bp->levels && g.c[k].ph) ? bp->base[k][level] : -1;
    p.bp8 = bp && level < bp->levels && bp->bytes8;
    p.store_wide = store_wide;
    p.qidx = qidx; p.err = c->d_err; p.dequant = dequant;
    p.ll_from_store = (level == g.depth - 1) && !ll_ready;
    const bool fin = (level == 0) && dst_raw;
    for (int k = 0; k < 3; ++k) {
      if (level == 0) { p.plane[k] = dst ? dst[k] : nullptr; p.plane_stride[k] = dst_stride ? dst_stride[k] : 0; }
      else { p.plane[k] = ll.p[level][k]; p.plane_wide[k] = ll.w[level][k]; p.plane_stride[k] = ll.stride[level][k]; }
      p.ll[k] = ll.p[level + 1][k]; p.ll_wide[k] = ll.w[level + 1][k]; p.ll_stride[k] = ll.stride[level + 1][k];
    }
    if (fin) {
      p.word_bytes = f->word_bytes;
      p.sample_shift = 8 * f->word_bytes - f->bit_depth;
      p.sample_offset = 1 << (f->bit_depth - 1);
      p.clip_lo = -(1 << (f->bit_depth - 1));
      p.clip_hi = (1 << (f->bit_depth - 1)) - 1;
    }
  };
  for (int level = g.depth - 1; level >= 0; --level) {
    LevelParams p;
    level_params(p, level);
    const bool fin = (level == 0) && dst_raw;
    // levels `level` and `level - 1` in one launch (vc2hip_dwt_pair.hip): the plane between them is never written
    if (!stream_mask && !c->force_generic && c->allow_stream && c->allow_pair && level >= 1) {
      PairParams pp;
      memset(&pp, 0, sizeof pp);
      level_params(pp.a, level - 1);
      pp.b = p;
      pp.a.debug_skip = c->debug_skip;
      const bool fin_a = (level - 1 == 0) && dst_raw;
      // The pair that ends with the picture's samples keeps its two launches: measured on 32 UHD pictures, k_inv_pair over
      // levels 1 + 0 takes 0.68 ms where the two one-level kernels take 0.47 + 0.16 -- level a alone runs at four
      // wavefronts per SIMD and at the memory system's pace (0.455 ms in this kernel's frame), with level b's engine in its
      // registers it runs at two and at the pace its instruction stream issues.  The deeper pairs gain.
      static const int inv_final = vc2_tune_int("VC2HIP_PAIR_INV_FINAL", 0);
      const size_t lds = (fin_a && !inv_final) ? 0 : vc2_pair_applicable(pp, kernel, fin_a, true, s16, n);
      if (lds) {
        int rc = vc2_launch_inverse_pair(c->L, kernel, fin_a, pp, n, s16, lds, c->stream);
        if (rc) return set_err(c, rc, "invalid wavelet kernel");
        --level;
        continue;
      }
    }
    LevelParams pf = p;
    pf.debug_skip = c->debug_skip;
    if (!c->force_generic && c->allow_stream) {
      LevelParams ps = p;
      const size_t lds = vc2_stream_level_applicable(ps, kernel, fin, true, s16, n);
      if (lds) {
        if (stream_mask) { *stream_mask |= 1u << level; if (tail_mask && ps.st_tail) *tail_mask |= 1u << level; continue; }
        int rc = vc2_launch_inverse_stream(c->L, kernel, fin, ps, n, s16, lds, c->stream);
        if (rc) return set_err(c, rc, "invalid wavelet kernel");
        continue;
      }
    }
    if (stream_mask) { if (fast_mask && !c->force_generic && vc2_fast_level_applicable(pf)) *fast_mask |= 1u << level; continue; }
    if (bp && level < bp->levels) return set_err(c, VC2HIP_EINVAL, "internal: band planes without the streaming kernel");
    if (!c->force_generic && vc2_fast_level_applicable(pf)) {
      int rc = vc2_launch_inverse_fast(c->L, kernel, fin, pf, n, s16, c->stream);
      if (rc) return set_err(c, rc, "invalid wavelet kernel");
      continue;
    }
    if (s16) return set_err(c, VC2HIP_EINVAL, "internal: 16-bit store without the fast level kernels");
    if (vc2_level_lds_bytes(kernel, p) > 160 * 1024) return set_err(c, VC2HIP_EINVAL, "slice too large for one LDS tile");
    int rc = vc2_launch_inverse_level(c->L, kernel, fin, p, n, c->stream);
    if (rc) return set_err(c, rc, "invalid wavelet kernel");
  }
  return VC2HIP_OK;
}

// The HQ batch path keeps the store and the level planes as 16-bit elements + wide planes (vc2hip_store.h) when every
// level runs through the fast kernels and every component record can be moved eight coefficients at a time.
static bool use_store16(const vc2hip_ctx *c, const Geom &g, int kernel) {
  if (!c->allow_store16 || c->force_generic) return false;
  for (int k = 0; k < 3; ++k) {
    if (!g.c[k].ph) continue;
    if ((g.c[k].sh * g.c[k].sw) % 8 || g.c[k].coef_off % 8) return false;
  }
  if (g.slice_coefs % 8) return false;
  for (int level = 0; level < g.depth; ++level) {
    LevelParams p;
    memset(&p, 0, sizeof p);
    fill_level(p, g, level, kernel, nullptr);
    if (!vc2_fast_level_applicable(p)) return false;
  }
  return true;
}

// A slice that does not fit the LDS tile of any level kernel (the reference admits slices up to the whole picture):
// the transform then runs on whole planes in HBM (vc2hip_dwt_plane.hip), the store is filled / read by the layout
// conversion kernels.
static bool needs_plane_path(const vc2hip_ctx *c, const Geom &g, int kernel) {
  for (int level = 0; level < g.depth; ++level) {
    LevelParams p;
    memset(&p, 0, sizeof p);
    fill_level(p, g, level, kernel, nullptr);
    LevelParams pf = p;
    if (!c->force_generic && vc2_fast_level_applicable(pf)) continue;
    if (vc2_level_lds_bytes(kernel, p) > 160 * 1024) return true;
  }
  return false;
}
static size_t plane_elems(const Geom &g) {
  size_t e = 0;
  for (int k = 0; k < 3; ++k) e += (size_t)g.c[k].ph * g.c[k].pw;
  return e;
}
// raw pictures -> store (forward), general geometry
static int plane_forward(vc2hip_ctx *c, const Geom &g, int kernel, int n, const void *const src[3], const long long ss[3],
                         const vc2hip_picture_format *f, int32_t *d_store) {
  int32_t *d_plane;
  NEED(c, B_PLANE, plane_elems(g) * n * 4, d_plane);
  size_t off = 0;
  for (int k = 0; k < 3; ++k) {
    const CompGeom &cg = g.c[k];
    if (!cg.ph) continue;
    const long long ps = (long long)cg.ph * cg.pw;
    int32_t *pl = d_plane + off;
    off += (size_t)ps * n;
    vc2_launch_plane_ingest(c->L, src[k], ss[k], cg.h, cg.w, f->word_bytes, k && f->chroma_bit_depth ? f->chroma_bit_depth : f->bit_depth, pl, ps,
                            cg.ph, cg.pw, n, c->stream);
    const int rc = vc2_launch_plane_transform(c->L, kernel, pl, ps, cg.ph, cg.pw, g.depth, false, n, c->stream);
    if (rc) return set_err(c, rc, "invalid wavelet kernel");
    for (int p = 0; p < n; ++p)
      vc2_launch_plane_to_store(c->L, pl + (size_t)p * ps, cg.ph, cg.pw, g.depth, g.ys, g.xs,
                                d_store + (size_t)p * g.ys * g.xs * g.slice_coefs, g.slice_coefs, cg.coef_off, c->stream);
  }
  return VC2HIP_OK;
}
// store (quantised) -> raw pictures (inverse), general geometry
// ll (LD pictures): the DC-predicted LL reconstruction of every component, which replaces the plane's LL band
static int plane_inverse(vc2hip_ctx *c, const Geom &g, int kernel, int n, const int32_t *d_store, const int32_t *d_q, const int32_t *qm,
                         void *const dst[3], const long long ds[3], const vc2hip_picture_format *f, const LLPlanes *ll = nullptr) {
  int32_t *d_plane;
  int *d_qm;
  NEED(c, B_PLANE, plane_elems(g) * n * 4, d_plane);
  NEED(c, B_QM, 256, d_qm);
  HIPCHK(c, hipMemcpyAsync(d_qm, qm, (size_t)(3 * g.depth + 1) * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream)); // qm lives on the caller's stack
  const int ns = g.ys * g.xs;
  size_t off = 0;
  for (int k = 0; k < 3; ++k) {
    const CompGeom &cg = g.c[k];
    if (!cg.ph) continue;
    const long long ps = (long long)cg.ph * cg.pw;
    int32_t *pl = d_plane + off;
    off += (size_t)ps * n;
    for (int p = 0; p < n; ++p)
      vc2_launch_store_to_plane(c->L, d_store + (size_t)p * ns * g.slice_coefs, g.slice_coefs, cg.coef_off, pl + (size_t)p * ps, cg.ph,
                                cg.pw, g.depth, g.ys, g.xs, d_q + (size_t)p * ns, d_qm, 1, c->d_err, c->stream);
    if (ll) vc2_launch_ll_into_plane(c->L, (const int32_t *)ll->p[g.depth][k], ll->stride[g.depth][k], cg.ph >> g.depth, cg.pw >> g.depth, pl, ps,
                                     cg.pw, g.depth, n, c->stream);
    const int rc = vc2_launch_plane_transform(c->L, kernel, pl, ps, cg.ph, cg.pw, g.depth, true, n, c->stream);
    if (rc) return set_err(c, rc, "invalid wavelet kernel");
    vc2_launch_plane_emit(c->L, pl, ps, cg.pw, dst[k], ds[k], cg.h, cg.w, f->word_bytes, f->bit_depth, n, c->stream);
  }
  return VC2HIP_OK;
}

static void fill_comp_arrays(const Geom &g, int n[3], int off[3], int n0[3]) {
  for (int c = 0; c < 3; ++c) { n[c] = g.c[c].sh * g.c[c].sw; off[c] = g.c[c].coef_off; n0[c] = g.c[c].n0 ? g.c[c].n0 : 1; }
}

// pack n pictures from the store into d_payload; VBR goes through slots + scan + compaction
static int run_pack(vc2hip_ctx *c, const Geom &g, int n, const void *store, const int32_t *d_qidx,
                    const int32_t *qm, bool quantise, int prefix, int scalar, const int32_t *d_cbr_bytes,
                    const uint32_t *d_cbr_offs, uint64_t cbr_total, uint8_t *d_payload, long long stride,
                    unsigned long long *d_lens, bool s16 = false, const int32_t *store_wide = nullptr) {
  const int ns = g.ys * g.xs;
  PackParams p;
  memset(&p, 0, sizeof p);
  p.store = store; p.store_stride = (long long)ns * g.slice_coefs;
  p.store16 = s16; p.store_wide = store_wide;
  p.qidx = d_qidx; p.n_slices = ns; p.slice_coefs = g.slice_coefs;
  fill_comp_arrays(g, p.comp_n, p.comp_off, p.comp_n0);
  p.depth = g.depth; p.prefix = prefix; p.scalar = scalar;
  for (int b = 0; b < 3 * g.depth + 1; ++b) p.qmatrix[b] = qm ? qm[b] : 0;
  p.err = c->d_err; p.quantise = quantise;
  p.payload = d_payload; p.payload_stride = stride;
  const bool gimg = vc2_pack_image_mode(prefix, scalar) < 0; // slice images in the slots (global memory): CBR goes through slots too
  if (d_cbr_bytes && !gimg) {
    p.cbr_bytes = d_cbr_bytes; p.cbr_offsets = d_cbr_offs;
    vc2_launch_pack(c->L, p, n, c->stream);
    vc2_launch_fill_u64(c->L, d_lens, cbr_total, (size_t)n, c->stream);
    return VC2HIP_OK;
  }
  // single pass: slice offsets by decoupled look-back inside the pack kernel -- the default where k_hq_pack16 codes the
  // slices (round 6), everywhere with VC2HIP_FLAG_SINGLE_PASS_VBR, nowhere with VC2HIP_FLAG_TWO_PASS_VBR
  // (the look-back pays from ~100 pictures per launch on: the grid puts the same tile of ALL pictures side by side, and a
  // picture's tile t - 1 is only that many workgroups ahead of tile t.  128 UHD pictures: 1.69 ms against 1.47 + 0.51 for
  // slots and compaction; 96: 1.46 = 1.46; 64: 1.15 against 0.75 + 0.25)
  const bool one_pass = !gimg && !d_cbr_bytes &&
                        (!c->two_pass_vbr || (!c->force_two_pass_vbr && n >= VC2_ONE_PASS_MIN_PICTURES && vc2_pack_one_pass_default(p)));
  if (one_pass) {
    const long long lb_stride = (long long)((ns + 3) / 4) + 8;
    unsigned long long *lb;
    NEED(c, B_SIZES, (size_t)n * lb_stride * 8, lb);
    HIPCHK(c, hipMemsetAsync(lb, 0, (size_t)n * lb_stride * 8, c->stream));
    vc2_prof_break(c->L);
    p.lookback = lb; p.lookback_stride = lb_stride; p.lens = d_lens;
    vc2_launch_pack(c->L, p, n, c->stream);
    return VC2HIP_OK;
  }
  // (images kept in the slots themselves: room for the two guard words of an image)
  const int slot = (int)((max_slice_bytes(prefix, scalar) + (gimg ? 8 : 0) + 15) & ~(size_t)15);
  uint8_t *slots; uint32_t *sizes, *offs;
  // Shared slots (below) hold WHOLE tiles: a picture's last tile has room for all its spt slices even when fewer exist
  // (1080p: 16200 slices in 1013 tiles of 16 = room for 16208).  Rounds 3 - 5 sized the buffer by the slice count; the
  // 2 MiB rounding of large workspaces hid the n * 8 slots that were missing until a batch of 136 HD pictures ran the
  // compaction off the end of it (round 6: tools/probe/fault_bisect.py, tests/test_gpu_parity.py::test_shared_slots_hold_whole_tiles)
  const int spt_room = (gimg || d_cbr_bytes) ? 0 : vc2_pack_slices_per_tile(p);
  const size_t slot_count = spt_room ? (size_t)((ns + spt_room - 1) / spt_room) * spt_room : (size_t)ns;
  NEED(c, B_SLOTS, (size_t)n * slot_count * slot + 32, slots); // (+32: the compaction reads whole 16-byte pieces and the dword behind them)
  NEED(c, B_SIZES, (size_t)n * ns * 4, sizes);
  NEED(c, B_OFFS, (size_t)n * ns * 4, offs);
  p.slots = slots; p.slot_bytes = slot; p.sizes = sizes;
  if (d_cbr_bytes) { p.cbr_bytes = d_cbr_bytes; p.cbr_offsets = d_cbr_offs; }
  // VBR with several slices per wavefront of the packer (slices of up to 256 + 2 x 128 coefficients: HD pictures): the
  // slices of a pack workgroup share a slot, back to back -- sizes, offsets and the compaction are per workgroup (a
  // sixteenth or an eighth of the entries and copies, each as many times as long).  32 HD pictures: pack 0.221 -> 0.227 ms
  // (byte-aligned copy-out), compaction 0.075 -> 0.045.  With a wavefront per slice (UHD: four slices of ~290 bytes per
  // workgroup) the two sides cancel (0.375 + 0.087 against 0.387 + 0.065 ms per 16 pictures), with large slices (UHD-2,
  // scalar 8) it loses: one slot per slice there.
  const int spt = (gimg || d_cbr_bytes) ? 0 : vc2_pack_slices_per_tile(p);
  p.tile_slices = spt;
  vc2_launch_pack(c->L, p, n, c->stream);
  if (spt) {
    const int nt = (ns + spt - 1) / spt;
    vc2_launch_scan_sizes(c->L, sizes, offs, d_lens, nt, n, c->stream);
    vc2_launch_compact(c->L, slots, spt * slot, sizes, offs, d_payload, stride, nt, n, c->stream);
    return VC2HIP_OK;
  }
  vc2_launch_scan_sizes(c->L, sizes, offs, d_lens, ns, n, c->stream); // (CBR: the same offsets as d_cbr_offs; sizes = budgets)
  vc2_launch_compact(c->L, slots, slot, sizes, offs, d_payload, stride, ns, n, c->stream);
  return VC2HIP_OK;
}

// ------------------------------------------------------------------------------------------
// fine-grained entry points (host planes; test / drop-in granularity)
// ------------------------------------------------------------------------------------------
static int one_plane_geom(Geom &g, int ph, int pw, int depth, int ys, int xs) {
  int phs[3] = {ph, 0, 0}, pws[3] = {pw, 0, 0};
  return make_geom(g, phs, pws, phs, pws, depth, ys, xs);
}

extern "C" int vc2hip_dwt_forward(vc2hip_ctx *c, const int32_t *in, int h, int w, int kernel, int depth,
                                  int32_t *out) {
  if (!c || !in || !out || h < 1 || w < 1 || depth < 1 || depth > VC2_MAX_DEPTH) return set_err(c, VC2HIP_EINVAL);
  if (kernel < 0 || kernel > 6) return set_err(c, VC2HIP_EINVAL, "invalid wavelet kernel");
  ENTER(c);
  const int ph = vc2hip_padded_size(h, depth), pw = vc2hip_padded_size(w, depth);
  Geom g;
  int rc = one_plane_geom(g, ph, pw, depth, ph >> depth, pw >> depth);
  if (rc) return set_err(c, rc);
  // waveletPad (WaveletTransform.cpp:79-94) on the host for this test-granularity entry point
  std::vector<int32_t> padded((size_t)ph * pw);
  for (int y = 0; y < ph; ++y)
    for (int x = 0; x < pw; ++x) padded[(size_t)y * pw + x] = in[(size_t)(y < h ? y : h - 1) * w + (x < w ? x : w - 1)];
  int32_t *d_plane, *d_store, *d_ll;
  const size_t pb = (size_t)ph * pw * 4;
  NEED(c, B_PLANE, pb, d_plane);
  NEED(c, B_STORE, pb, d_store);
  NEED(c, B_LL0, ll_bytes(g, 1) + 16, d_ll);
  LLPlanes ll;
  ll_layout(g, 1, d_ll, ll);
  HIPCHK(c, hipMemcpyAsync(d_plane, padded.data(), pb, hipMemcpyHostToDevice, c->stream));
  const void *src[3] = {d_plane, nullptr, nullptr};
  const long long ss[3] = {(long long)ph * pw, 0, 0};
  rc = run_forward(c, g, kernel, 1, src, ss, false, nullptr, d_store, ll);
  if (rc) return rc;
  vc2_launch_store_to_plane(c->L, d_store, g.slice_coefs, 0, d_plane, ph, pw, depth, g.ys, g.xs, nullptr, nullptr, 0,
                            c->d_err, c->stream);
  HIPCHK(c, hipMemcpyAsync(out, d_plane, pb, hipMemcpyDeviceToHost, c->stream));
  return vc2hip_sync(c);
}

extern "C" int vc2hip_dwt_inverse(vc2hip_ctx *c, const int32_t *in, int ph, int pw, int kernel, int depth,
                                  int32_t *out, int h, int w) {
  if (!c || !in || !out || depth < 1 || depth > VC2_MAX_DEPTH || h < 1 || w < 1 || h > ph || w > pw) return set_err(c, VC2HIP_EINVAL);
  if (kernel < 0 || kernel > 6) return set_err(c, VC2HIP_EINVAL, "invalid wavelet kernel");
  if (ph % (1 << depth) || pw % (1 << depth)) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = one_plane_geom(g, ph, pw, depth, ph >> depth, pw >> depth);
  if (rc) return set_err(c, rc);
  int32_t *d_plane, *d_store, *d_ll;
  const size_t pb = (size_t)ph * pw * 4;
  NEED(c, B_PLANE, pb, d_plane);
  NEED(c, B_STORE, pb, d_store);
  NEED(c, B_LL0, ll_bytes(g, 1) + 16, d_ll);
  LLPlanes ll;
  ll_layout(g, 1, d_ll, ll);
  HIPCHK(c, hipMemcpyAsync(d_plane, in, pb, hipMemcpyHostToDevice, c->stream));
  vc2_launch_plane_to_store(c->L, d_plane, ph, pw, depth, g.ys, g.xs, d_store, g.slice_coefs, 0, c->stream);
  void *dst[3] = {d_plane, nullptr, nullptr};
  const long long ds[3] = {(long long)ph * pw, 0, 0};
  rc = run_inverse(c, g, kernel, 1, d_store, nullptr, nullptr, false, false, ll, dst, ds, false, nullptr);
  if (rc) return rc;
  std::vector<int32_t> full((size_t)ph * pw);
  HIPCHK(c, hipMemcpyAsync(full.data(), d_plane, pb, hipMemcpyDeviceToHost, c->stream));
  rc = vc2hip_sync(c);
  if (rc) return rc;
  for (int y = 0; y < h; ++y) memcpy(out + (size_t)y * w, full.data() + (size_t)y * pw, (size_t)w * 4); // resize(shape), :340
  return VC2HIP_OK;
}

// common body of quantise_np / dequantise_np / dequantise_ld
static int plane_quant_op(vc2hip_ctx *c, const int32_t *in, int ph, int pw, int depth, const int32_t *qidx, int ys,
                          int xs, const int32_t *qm, int32_t *out, int op /*0 quant,1 scale,2 scale+LD*/) {
  if (!c || !in || !out || !qidx || !qm) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = one_plane_geom(g, ph, pw, depth, ys, xs);
  if (rc) return set_err(c, rc);
  const size_t pb = (size_t)ph * pw * 4;
  const int ns = ys * xs, nb = 3 * depth + 1;
  int32_t *d_plane, *d_store, *d_q, *d_ll = nullptr;
  int *d_qm;
  NEED(c, B_PLANE, pb, d_plane);
  NEED(c, B_STORE, pb, d_store);
  NEED(c, B_QIDX, (size_t)ns * 4, d_q);
  NEED(c, B_QM, 256, d_qm);
  HIPCHK(c, hipMemcpyAsync(d_plane, in, pb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_q, qidx, (size_t)ns * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_qm, qm, (size_t)nb * 4, hipMemcpyHostToDevice, c->stream));
  vc2_launch_plane_to_store(c->L, d_plane, ph, pw, depth, ys, xs, d_store, g.slice_coefs, 0, c->stream);
  if (op == 0) {
    vc2_launch_quantise_store(c->L, d_store, ns, g.slice_coefs, g.c[0].sh * g.c[0].sw, 0, g.c[0].n0, d_q, d_qm, c->d_err, c->stream);
    vc2_launch_store_to_plane(c->L, d_store, g.slice_coefs, 0, d_plane, ph, pw, depth, ys, xs, d_q, d_qm, 0, c->d_err, c->stream);
  } else {
    vc2_launch_store_to_plane(c->L, d_store, g.slice_coefs, 0, d_plane, ph, pw, depth, ys, xs, d_q, d_qm, 1, c->d_err, c->stream);
  }
  const int llh = ph >> depth, llw = pw >> depth;
  if (op == 2) {
    NEED(c, B_LL0, (size_t)llh * llw * 4, d_ll);
    vc2_launch_ld_ll(c->L, d_store, 0, g.slice_coefs, 0, g.c[0].n0, llh, llw, ys, xs, d_q, qm[0], d_ll, 0, 1, c->d_err, c->stream);
  }
  HIPCHK(c, hipMemcpyAsync(out, d_plane, pb, hipMemcpyDeviceToHost, c->stream));
  std::vector<int32_t> llh_host;
  if (op == 2) {
    llh_host.resize((size_t)llh * llw);
    HIPCHK(c, hipMemcpyAsync(llh_host.data(), d_ll, llh_host.size() * 4, hipMemcpyDeviceToHost, c->stream));
  }
  rc = vc2hip_sync(c);
  if (rc) return rc;
  if (op == 2) {
    const int s = 1 << depth;
    for (int y = 0; y < llh; ++y)
      for (int x = 0; x < llw; ++x) out[(size_t)y * s * pw + (size_t)x * s] = llh_host[(size_t)y * llw + x];
  }
  return VC2HIP_OK;
}
extern "C" int vc2hip_quantise_np(vc2hip_ctx *c, const int32_t *coef, int ph, int pw, int depth, const int32_t *qidx,
                                  int ys, int xs, const int32_t *qm, int32_t *out) {
  return plane_quant_op(c, coef, ph, pw, depth, qidx, ys, xs, qm, out, 0);
}
extern "C" int vc2hip_dequantise_np(vc2hip_ctx *c, const int32_t *q, int ph, int pw, int depth, const int32_t *qidx,
                                    int ys, int xs, const int32_t *qm, int32_t *out) {
  return plane_quant_op(c, q, ph, pw, depth, qidx, ys, xs, qm, out, 1);
}
extern "C" int vc2hip_dequantise_ld(vc2hip_ctx *c, const int32_t *q, int ph, int pw, int depth, const int32_t *qidx,
                                    int ys, int xs, const int32_t *qm, int32_t *out) {
  return plane_quant_op(c, q, ph, pw, depth, qidx, ys, xs, qm, out, 2);
}

static int geom_from_abi(Geom &g, const vc2hip_geom *a) {
  int ph[3] = {a->luma_h, a->chroma_h, a->chroma_h}, pw[3] = {a->luma_w, a->chroma_w, a->chroma_w};
  return make_geom(g, ph, pw, ph, pw, a->depth, a->y_slices, a->x_slices);
}

// upload three planes and scatter them into the store
static int planes_to_store(vc2hip_ctx *c, const Geom &g, const int32_t *const pl[3], int32_t *d_store) {
  size_t mx = 0;
  for (int k = 0; k < 3; ++k) mx = std::max(mx, (size_t)g.c[k].ph * g.c[k].pw * 4);
  int32_t *d_plane;
  NEED(c, B_PLANE, mx * 3, d_plane);
  for (int k = 0; k < 3; ++k) {
    int32_t *dp = d_plane + (mx / 4) * k;
    HIPCHK(c, hipMemcpyAsync(dp, pl[k], (size_t)g.c[k].ph * g.c[k].pw * 4, hipMemcpyHostToDevice, c->stream));
    vc2_launch_plane_to_store(c->L, dp, g.c[k].ph, g.c[k].pw, g.depth, g.ys, g.xs, d_store, g.slice_coefs, g.c[k].coef_off, c->stream);
  }
  return VC2HIP_OK;
}
static int store_to_planes(vc2hip_ctx *c, const Geom &g, const int32_t *d_store, int32_t *const pl[3]) {
  size_t mx = 0;
  for (int k = 0; k < 3; ++k) mx = std::max(mx, (size_t)g.c[k].ph * g.c[k].pw * 4);
  int32_t *d_plane;
  NEED(c, B_PLANE, mx * 3, d_plane);
  for (int k = 0; k < 3; ++k) {
    int32_t *dp = d_plane + (mx / 4) * k;
    vc2_launch_store_to_plane(c->L, d_store, g.slice_coefs, g.c[k].coef_off, dp, g.c[k].ph, g.c[k].pw, g.depth, g.ys, g.xs,
                              nullptr, nullptr, 0, c->d_err, c->stream);
    HIPCHK(c, hipMemcpyAsync(pl[k], dp, (size_t)g.c[k].ph * g.c[k].pw * 4, hipMemcpyDeviceToHost, c->stream));
  }
  return VC2HIP_OK;
}

static int cbr_offsets_upload(vc2hip_ctx *c, const int32_t *bytes, int ns, int prefix, int32_t **d_b, uint32_t **d_o, uint64_t *total) {
  c->cbr_key[0] = -1; // whatever was cached is overwritten
  std::vector<uint32_t> offs(ns);
  uint64_t run = 0;
  for (int i = 0; i < ns; ++i) { offs[i] = (uint32_t)run; run += (uint64_t)bytes[i] + prefix; }
  *total = run;
  int32_t *db; uint32_t *dof;
  NEED(c, B_CBRB, (size_t)ns * 4, db);
  NEED(c, B_CBRO, (size_t)ns * 4, dof);
  HIPCHK(c, hipMemcpyAsync(db, bytes, (size_t)ns * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(dof, offs.data(), (size_t)ns * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *d_b = db; *d_o = dof;
  return VC2HIP_OK;
}

extern "C" int vc2hip_hq_pack(vc2hip_ctx *c, const int32_t *y, const int32_t *u, const int32_t *v, const vc2hip_geom *ga,
                              const int32_t *qidx, int prefix, int scalar, const int32_t *cbr, uint8_t *out, size_t cap,
                              size_t *out_len) {
  if (!c || !y || !u || !v || !ga || !qidx || !out || !out_len || prefix < 0 || scalar < 1) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = geom_from_abi(g, ga);
  if (rc) return set_err(c, rc);
  const int ns = g.ys * g.xs;
  int32_t *d_store, *d_q; uint8_t *d_pay; unsigned long long *d_len;
  NEED(c, B_STORE, (size_t)ns * g.slice_coefs * 4, d_store);
  NEED(c, B_QIDX, (size_t)ns * 4, d_q);
  NEED(c, B_LENS, 64, d_len);
  const int32_t *pl[3] = {y, u, v};
  if ((rc = planes_to_store(c, g, pl, d_store))) return rc;
  HIPCHK(c, hipMemcpyAsync(d_q, qidx, (size_t)ns * 4, hipMemcpyHostToDevice, c->stream));
  size_t paycap = (size_t)ns * max_slice_bytes(prefix, scalar);
  int32_t *d_cb = nullptr; uint32_t *d_co = nullptr; uint64_t total = 0;
  if (cbr) {
    if ((rc = cbr_offsets_upload(c, cbr, ns, prefix, &d_cb, &d_co, &total))) return rc;
    paycap = total;
  }
  NEED(c, B_PAYLOAD, paycap + 64, d_pay);
  rc = run_pack(c, g, 1, d_store, d_q, nullptr, false, prefix, scalar, d_cb, d_co, total, d_pay, (long long)paycap, d_len);
  if (rc) return rc;
  unsigned long long len = 0;
  HIPCHK(c, hipMemcpyAsync(&len, d_len, 8, hipMemcpyDeviceToHost, c->stream));
  if ((rc = vc2hip_sync(c))) return rc;
  if (len > cap) return set_err(c, VC2HIP_ECAP);
  HIPCHK(c, hipMemcpy(out, d_pay, len, hipMemcpyDeviceToHost));
  *out_len = (size_t)len;
  return VC2HIP_OK;
}

// slice offsets of one VBR payload already on the device
// cbr_budget / cbr_offs / cbr_total (device tables of the HQ_CBR slice budgets, or null): the offsets are first claimed
// from the budgets and checked against the stream (vc2_launch_cbr_index); the general index only works if that fails
static int build_index(vc2hip_ctx *c, const uint8_t *d_pay, long long stride, const unsigned long long *d_lens, int n, int ns,
                       int prefix, int scalar, uint32_t **d_offs_out, const int32_t *cbr_budget = nullptr,
                       const uint32_t *cbr_offs = nullptr, uint64_t cbr_total = 0) {
  uint32_t *d_offs; void *ws;
  NEED(c, B_OFFS, (size_t)n * ns * 4, d_offs);
  const size_t wsb = vc2_slice_index_workspace(n, (size_t)stride, prefix, scalar);
  NEED(c, B_INDEX, wsb, ws);
  HIPCHK(c, hipMemsetAsync(d_offs, 0xFF, (size_t)n * ns * 4, c->stream)); // unreachable slices read past the payload
  unsigned *bad = nullptr;
  if (cbr_budget && c->allow_cbr_index) {
    bad = (unsigned *)((char *)c->d_err + 64); // (the error word's block: 256 bytes, the LD tables behind them)
    HIPCHK(c, hipMemsetAsync(bad, 0, sizeof(unsigned), c->stream));
  }
  vc2_prof_break(c->L);
  if (bad) vc2_launch_cbr_index(c->L, d_pay, stride, d_lens, cbr_budget, cbr_offs, cbr_total, d_offs, ns, prefix, scalar, n, bad, c->stream);
  vc2_launch_slice_index(c->L, d_pay, stride, d_lens, d_offs, ns, prefix, scalar, n, c->d_err, c->stream, ws, wsb, bad);
  *d_offs_out = d_offs;
  return VC2HIP_OK;
}

extern "C" int vc2hip_hq_unpack(vc2hip_ctx *c, const uint8_t *in, size_t len, const vc2hip_geom *ga, int prefix, int scalar,
                                int32_t *y, int32_t *u, int32_t *v, int32_t *qidx, size_t *consumed) {
  if (!c || !in || !ga || !y || !u || !v || !qidx || prefix < 0 || scalar < 1) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = geom_from_abi(g, ga);
  if (rc) return set_err(c, rc);
  const int ns = g.ys * g.xs;
  int32_t *d_store, *d_q; uint8_t *d_pay; unsigned long long *d_len; uint32_t *d_offs;
  NEED(c, B_STORE, (size_t)ns * g.slice_coefs * 4, d_store);
  NEED(c, B_QIDX, (size_t)ns * 4, d_q);
  NEED(c, B_LENS, 64, d_len);
  const size_t stride = (len + 63) & ~(size_t)63;
  NEED(c, B_PAYLOAD, stride + 64, d_pay);
  unsigned long long l64 = len;
  HIPCHK(c, hipMemcpyAsync(d_pay, in, len, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_len, &l64, 8, hipMemcpyHostToDevice, c->stream));
  if ((rc = build_index(c, d_pay, (long long)stride, d_len, 1, ns, prefix, scalar, &d_offs))) return rc;
  UnpackParams p;
  memset(&p, 0, sizeof p);
  p.payload = d_pay; p.payload_stride = (long long)stride; p.lens = d_len; p.offsets = d_offs;
  p.store = d_store; p.store_stride = (long long)ns * g.slice_coefs; p.qidx = d_q;
  p.n_slices = ns; p.slice_coefs = g.slice_coefs;
  int n0[3];
  fill_comp_arrays(g, p.comp_n, p.comp_off, n0);
  p.prefix = prefix; p.scalar = scalar; p.err = c->d_err;
  vc2_launch_unpack(c->L, p, 1, c->stream);
  int32_t *pl[3] = {y, u, v};
  if ((rc = store_to_planes(c, g, d_store, pl))) return rc;
  HIPCHK(c, hipMemcpyAsync(qidx, d_q, (size_t)ns * 4, hipMemcpyDeviceToHost, c->stream));
  uint32_t last_off = 0;
  HIPCHK(c, hipMemcpyAsync(&last_off, d_offs + ns - 1, 4, hipMemcpyDeviceToHost, c->stream));
  if ((rc = vc2hip_sync(c))) return rc;
  if (consumed) { // end of the last slice
    size_t pos = (size_t)last_off + prefix + 1;
    for (int k = 0; k < 3 && pos < len; ++k) pos += 1 + (size_t)in[pos] * scalar;
    *consumed = pos;
  }
  return VC2HIP_OK;
}

extern "C" int vc2hip_cbr_qindices(vc2hip_ctx *c, const int32_t *y, const int32_t *u, const int32_t *v, const vc2hip_geom *ga,
                                   const int32_t *qm, const int32_t *slice_bytes, int scalar, int32_t *qidx) {
  if (!c || !y || !u || !v || !ga || !qm || !slice_bytes || !qidx || scalar < 1) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = geom_from_abi(g, ga);
  if (rc) return set_err(c, rc);
  const int ns = g.ys * g.xs;
  int32_t *d_store, *d_q, *d_sb;
  NEED(c, B_STORE, (size_t)ns * g.slice_coefs * 4, d_store);
  NEED(c, B_QIDX, (size_t)ns * 4, d_q);
  NEED(c, B_CBRB, (size_t)ns * 4, d_sb);
  c->cbr_key[0] = -1;
  const int32_t *pl[3] = {y, u, v};
  if ((rc = planes_to_store(c, g, pl, d_store))) return rc;
  HIPCHK(c, hipMemcpyAsync(d_sb, slice_bytes, (size_t)ns * 4, hipMemcpyHostToDevice, c->stream));
  CbrParams p;
  memset(&p, 0, sizeof p);
  p.store = d_store; p.store_stride = (long long)ns * g.slice_coefs; p.qidx = d_q; p.slice_bytes = d_sb;
  p.n_slices = ns; p.slice_coefs = g.slice_coefs;
  fill_comp_arrays(g, p.comp_n, p.comp_off, p.comp_n0);
  p.scalar = scalar; p.err = c->d_err;
  p.general_only = c->cbr_general;
  for (int b = 0; b < 3 * g.depth + 1; ++b) p.qmatrix[b] = qm[b];
  p.n_bands = 3 * g.depth + 1;
  vc2_launch_cbr(c->L, p, 1, c->stream);
  HIPCHK(c, hipMemcpyAsync(qidx, d_q, (size_t)ns * 4, hipMemcpyDeviceToHost, c->stream));
  return vc2hip_sync(c);
}

static int ld_offsets_upload(vc2hip_ctx *c, const int32_t *bytes, int ns, int32_t **d_b, uint32_t **d_o, uint64_t *total) {
  return cbr_offsets_upload(c, bytes, ns, 0, d_b, d_o, total);
}

static void fill_ld_unpack(LdUnpackParams &p, const Geom &g, const uint8_t *d_pay, long long stride, const int32_t *d_sb,
                           const uint32_t *d_so, int32_t *d_store, int32_t *d_q, unsigned *err) {
  memset(&p, 0, sizeof p);
  const int ns = g.ys * g.xs;
  p.payload = d_pay; p.payload_stride = stride; p.slice_bytes = d_sb; p.offsets = d_so;
  p.store = d_store; p.store_stride = (long long)ns * g.slice_coefs; p.qidx = d_q;
  p.n_slices = ns; p.slice_coefs = g.slice_coefs;
  int n0[3];
  fill_comp_arrays(g, p.comp_n, p.comp_off, n0);
  p.err = err;
}

extern "C" int vc2hip_ld_unpack(vc2hip_ctx *c, const uint8_t *in, size_t len, const vc2hip_geom *ga, const int32_t *slice_bytes,
                                int32_t *y, int32_t *u, int32_t *v, int32_t *qidx, size_t *consumed) {
  if (!c || !in || !ga || !slice_bytes || !y || !u || !v || !qidx) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = geom_from_abi(g, ga);
  if (rc) return set_err(c, rc);
  const int ns = g.ys * g.xs;
  int32_t *d_store, *d_q, *d_sb; uint32_t *d_so; uint8_t *d_pay; uint64_t total;
  NEED(c, B_STORE, (size_t)ns * g.slice_coefs * 4, d_store);
  NEED(c, B_QIDX, (size_t)ns * 4, d_q);
  if ((rc = ld_offsets_upload(c, slice_bytes, ns, &d_sb, &d_so, &total))) return rc;
  if (total > len) return set_err(c, VC2HIP_ESTREAM);
  NEED(c, B_PAYLOAD, len + 64, d_pay);
  HIPCHK(c, hipMemcpyAsync(d_pay, in, len, hipMemcpyHostToDevice, c->stream));
  LdUnpackParams p;
  fill_ld_unpack(p, g, d_pay, (long long)len, d_sb, d_so, d_store, d_q, c->d_err);
  vc2_launch_ld_unpack(c->L, p, 1, c->stream);
  int32_t *pl[3] = {y, u, v};
  if ((rc = store_to_planes(c, g, d_store, pl))) return rc;
  HIPCHK(c, hipMemcpyAsync(qidx, d_q, (size_t)ns * 4, hipMemcpyDeviceToHost, c->stream));
  if (consumed) *consumed = (size_t)total;
  return vc2hip_sync(c);
}

// ------------------------------------------------------------------------------------------
// LD encode
// ------------------------------------------------------------------------------------------
static int fill_ld_enc(vc2hip_ctx *c, LdEncParams &p, const Geom &g, int32_t *d_store, int32_t *d_q, const int32_t *qm,
                       const LLPlanes &ll, const int32_t *d_sb, const uint32_t *d_so, int max_slice, uint8_t *d_pay,
                       long long stride) {
  memset(&p, 0, sizeof p);
  const int ns = g.ys * g.xs;
  p.store = d_store; p.store_stride = (long long)ns * g.slice_coefs; p.qidx = d_q;
  p.slice_bytes = d_sb; p.offsets = d_so;
  for (int k = 0; k < 3; ++k) {
    p.restored[k] = (int32_t *)ll.p[g.depth][k]; p.restored_stride[k] = ll.stride[g.depth][k];
    p.ll_w[k] = g.c[k].pw >> g.depth;
    p.bh[k] = (g.c[k].ph >> g.depth) / g.ys; p.bw[k] = (g.c[k].pw >> g.depth) / g.xs;
  }
  p.ys = g.ys; p.xs = g.xs; p.n_slices = ns; p.slice_coefs = g.slice_coefs;
  p.diagonals = c->ld_diagonals;
  p.depth = g.depth;
  p.rs_ints = 0;
  for (int k = 0; k < 3; ++k) if (g.c[k].ph) p.rs_ints += (p.bh[k] + 1) * (p.bw[k] + 1);
  fill_comp_arrays(g, p.comp_n, p.comp_off, p.comp_n0);
  for (int b = 0; b < 3 * g.depth + 1; ++b) p.qmatrix[b] = qm ? qm[b] : 0;
  p.img_words = (max_slice + 3) / 4 + 2;
  p.payload = d_pay; p.payload_stride = stride; p.err = c->d_err;
  p.tab = (int *)((char *)c->d_err + 256);
  // what has to fit in LDS whatever the slice's size: the LL blocks with their halo (search) and one slice's bytes (writer)
  if (512 + (size_t)p.rs_ints * 4 > 160 * 1024 || (size_t)p.img_words * 4 > 160 * 1024)
    return set_err(c, VC2HIP_EINVAL, "slice too large for the LD encode kernels");
  if ((size_t)g.slice_coefs * 8 + 512 + (size_t)p.rs_ints * 4 > 160 * 1024) { // the coefficients do not: the search reads the store, its trials write a scratch array
    NEED(c, B_PLANE2, (size_t)((p.store_stride ? p.store_stride : 1)) * 4 * (size_t)std::max(1, c->ld_batch), p.scratch);
  }
  c->ld_batch = 1;
  return VC2HIP_OK;
}

/* quantise_transform (LD, DC-predicted LL band), Quantisation.cpp:358-367 over :213-234 */
extern "C" int vc2hip_quantise_ld(vc2hip_ctx *c, const int32_t *coef, int ph, int pw, int depth, const int32_t *qidx,
                                  int ys, int xs, const int32_t *qm, int32_t *out) {
  if (!c || !coef || !out || !qidx || !qm) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = one_plane_geom(g, ph, pw, depth, ys, xs);
  if (rc) return set_err(c, rc);
  const size_t pb = (size_t)ph * pw * 4;
  const int ns = ys * xs;
  int32_t *d_plane, *d_store, *d_q, *d_ll;
  NEED(c, B_PLANE, pb, d_plane);
  NEED(c, B_STORE, pb, d_store);
  NEED(c, B_QIDX, (size_t)ns * 4, d_q);
  NEED(c, B_LL0, ll_bytes(g, 1) + 16, d_ll);
  LLPlanes ll;
  ll_layout(g, 1, d_ll, ll);
  HIPCHK(c, hipMemcpyAsync(d_plane, coef, pb, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_q, qidx, (size_t)ns * 4, hipMemcpyHostToDevice, c->stream));
  vc2_launch_plane_to_store(c->L, d_plane, ph, pw, depth, ys, xs, d_store, g.slice_coefs, 0, c->stream);
  LdEncParams p;
  if ((rc = fill_ld_enc(c, p, g, d_store, d_q, qm, ll, nullptr, nullptr, 0, nullptr, 0))) return rc;
  p.search = 0;
  vc2_launch_ld_quantise(c->L, p, 1, c->stream);
  vc2_launch_store_to_plane(c->L, d_store, g.slice_coefs, 0, d_plane, ph, pw, depth, ys, xs, nullptr, nullptr, 0, c->d_err, c->stream);
  HIPCHK(c, hipMemcpyAsync(out, d_plane, pb, hipMemcpyDeviceToHost, c->stream));
  return vc2hip_sync(c);
}

/* quantIndicesLD(coefficients, qMatrix, sliceBytes), EncodeStream.cpp:141-245.  y,u,v: TRANSFORM planes */
extern "C" int vc2hip_ld_qindices(vc2hip_ctx *c, const int32_t *y, const int32_t *u, const int32_t *v, const vc2hip_geom *ga,
                                  const int32_t *qm, const int32_t *slice_bytes, int32_t *qidx) {
  if (!c || !y || !u || !v || !ga || !qm || !slice_bytes || !qidx) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = geom_from_abi(g, ga);
  if (rc) return set_err(c, rc);
  const int ns = g.ys * g.xs;
  int32_t *d_store, *d_q, *d_sb, *d_ll; uint32_t *d_so; uint64_t total;
  NEED(c, B_STORE, (size_t)ns * g.slice_coefs * 4, d_store);
  NEED(c, B_QIDX, (size_t)ns * 4, d_q);
  NEED(c, B_LL0, ll_bytes(g, 1) + 16, d_ll);
  LLPlanes ll;
  ll_layout(g, 1, d_ll, ll);
  const int32_t *pl[3] = {y, u, v};
  if ((rc = planes_to_store(c, g, pl, d_store))) return rc;
  if ((rc = ld_offsets_upload(c, slice_bytes, ns, &d_sb, &d_so, &total))) return rc;
  LdEncParams p;
  if ((rc = fill_ld_enc(c, p, g, d_store, d_q, qm, ll, d_sb, d_so, 0, nullptr, 0))) return rc;
  p.search = 1;
  vc2_launch_ld_quantise(c->L, p, 1, c->stream);
  HIPCHK(c, hipMemcpyAsync(qidx, d_q, (size_t)ns * 4, hipMemcpyDeviceToHost, c->stream));
  return vc2hip_sync(c);
}

/* operator<<(ostream&, const Slices&) under sliceio::lowDelay(bytes), Slices.cpp:645-660 over :195-244.
 * y,u,v: QUANTISED planes (LL band as prediction residuals). */
extern "C" int vc2hip_ld_pack(vc2hip_ctx *c, const int32_t *y, const int32_t *u, const int32_t *v, const vc2hip_geom *ga,
                              const int32_t *qidx, const int32_t *slice_bytes, uint8_t *out, size_t cap, size_t *out_len) {
  if (!c || !y || !u || !v || !ga || !qidx || !slice_bytes || !out || !out_len) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = geom_from_abi(g, ga);
  if (rc) return set_err(c, rc);
  const int ns = g.ys * g.xs;
  int mx = 1;
  for (int i = 0; i < ns; ++i) { if (slice_bytes[i] < 1) return set_err(c, VC2HIP_EINVAL); mx = std::max(mx, slice_bytes[i]); }
  int32_t *d_store, *d_q, *d_sb; uint32_t *d_so; uint8_t *d_pay; uint64_t total;
  NEED(c, B_STORE, (size_t)ns * g.slice_coefs * 4, d_store);
  NEED(c, B_QIDX, (size_t)ns * 4, d_q);
  const int32_t *pl[3] = {y, u, v};
  if ((rc = planes_to_store(c, g, pl, d_store))) return rc;
  HIPCHK(c, hipMemcpyAsync(d_q, qidx, (size_t)ns * 4, hipMemcpyHostToDevice, c->stream));
  if ((rc = ld_offsets_upload(c, slice_bytes, ns, &d_sb, &d_so, &total))) return rc;
  if (total > cap) return set_err(c, VC2HIP_ECAP);
  NEED(c, B_PAYLOAD, total + 64, d_pay);
  LLPlanes ll;
  memset(&ll, 0, sizeof ll);
  LdEncParams p;
  if ((rc = fill_ld_enc(c, p, g, d_store, d_q, nullptr, ll, d_sb, d_so, mx, d_pay, (long long)total))) return rc;
  vc2_launch_ld_pack(c->L, p, 1, c->stream);
  if ((rc = vc2hip_sync(c))) return rc;
  HIPCHK(c, hipMemcpy(out, d_pay, total, hipMemcpyDeviceToHost));
  *out_len = (size_t)total;
  return VC2HIP_OK;
}

// ------------------------------------------------------------------------------------------
// fused picture path
// ------------------------------------------------------------------------------------------
static void raw_planes(const vc2hip_picture_format *f, const void *base, const void *pl[3], long long stride[3]) {
  int ch, cw;
  chroma_dims(f->height, f->width, f->chroma_format, &ch, &cw);
  const size_t ln = (size_t)f->height * f->width * f->word_bytes, cn = (size_t)ch * cw * f->word_bytes;
  pl[0] = base;
  pl[1] = (const uint8_t *)base + ln;
  pl[2] = (const uint8_t *)base + ln + cn;
  stride[0] = stride[1] = stride[2] = (long long)(ln + 2 * cn);
}

extern "C" int vc2hip_encode_batch_dev(vc2hip_ctx *c, const void *d_raw, int n, const vc2hip_picture_format *f,
                                       const vc2hip_coding_params *cp, void *d_payload, size_t payload_stride,
                                       uint64_t *d_lens) {
  if (!c || !d_raw || n < 1 || !f || !cp || !d_payload || !d_lens) return set_err(c, VC2HIP_EINVAL);
  if (((size_t)d_raw | (size_t)d_payload | payload_stride) & 15 || ((size_t)d_lens & 7))
    return set_err(c, VC2HIP_EINVAL, "device buffers and the payload stride must be 16-byte aligned");
  if (c->lanes.size() > 1 && n > 1 && !c->in_split) {
    const size_t rb = vc2hip_raw_picture_bytes(f);
    const uint8_t *raw8 = (const uint8_t *)d_raw, *pay8 = (const uint8_t *)d_payload, *len8 = (const uint8_t *)d_lens;
    return split_batch(c, n,
      [&](int first, int count, vc2hip_ctx::LaneUse &u) {
        u.r[0] = {raw8 + (size_t)first * rb, raw8 + (size_t)(first + count) * rb}; u.r[1] = {nullptr, nullptr};
        u.w[0] = {pay8 + (size_t)first * payload_stride, pay8 + (size_t)(first + count) * payload_stride};
        u.w[1] = {len8 + (size_t)first * 8, len8 + (size_t)(first + count) * 8};
      },
      [&](vc2hip_ctx *l, int first, int count) {
        return vc2hip_encode_batch_dev(l, raw8 + (size_t)first * rb, count, f, cp,
                                       (uint8_t *)d_payload + (size_t)first * payload_stride, payload_stride, d_lens + first);
      });
  }
  if (cp->mode != VC2HIP_HQ_CONSTQ && cp->mode != VC2HIP_HQ_CBR && cp->mode != VC2HIP_LD) return set_err(c, VC2HIP_EINVAL);
  if (cp->kernel < 0 || cp->kernel > 6) return set_err(c, VC2HIP_EINVAL, "invalid wavelet kernel");
  if (cp->mode != VC2HIP_LD && (cp->scalar < 1 || cp->prefix < 0)) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  Geom g;
  int rc = picture_geom(g, f, cp, false);
  if (rc) return set_err(c, rc);
  const int ns = g.ys * g.xs;
  int32_t qm[VC2_MAX_BANDS];
  if ((rc = vc2hip_quant_matrix(cp->kernel, cp->depth, qm))) return set_err(c, rc);
  const bool s16 = cp->mode != VC2HIP_LD && use_store16(c, g, cp->kernel);
  int32_t *d_store, *d_ll, *d_q, *d_storew = nullptr, *d_llw = nullptr;
  NEED(c, B_STORE, (size_t)n * ns * g.slice_coefs * 4, d_store); // 16-bit elements use the first half
  NEED(c, B_LL0, ll_bytes(g, n) + 16, d_ll);
  if (s16) { // touched only by values outside 16 bits
    NEED(c, B_STOREW, (size_t)n * ns * g.slice_coefs * 4, d_storew);
    NEED(c, B_LLW, ll_bytes(g, n) + 16, d_llw);
  }
  NEED(c, B_QIDX, (size_t)n * ns * 4, d_q);
  LLPlanes ll;
  ll_layout(g, n, d_ll, s16 ? 2 : 4, d_llw, ll);
  const void *src[3]; long long ss[3];
  raw_planes(f, d_raw, src, ss);
  if (needs_plane_path(c, g, cp->kernel)) { // (a slice beyond any LDS tile: HQ and LD alike -- the store is int32 then)
    if ((rc = plane_forward(c, g, cp->kernel, n, src, ss, f, d_store))) return rc;
  } else if ((rc = run_forward(c, g, cp->kernel, n, src, ss, true, f, d_store, ll, s16, d_storew))) return rc;
  int32_t *d_cb = nullptr; uint32_t *d_co = nullptr; uint64_t total = 0;
  if (cp->mode == VC2HIP_LD) {
    // EncodeStream.cpp:509-512 (slice_bytes with scalar 1), :141-245, :195-244
    const int key[5] = {g.ys, g.xs, cp->compressed_bytes, 1, -7};
    if (memcmp(key, c->cbr_key, sizeof key) || !c->buf[B_CBRB].p) {
      std::vector<int32_t> sb(ns);
      vc2hip_slice_bytes(g.ys, g.xs, cp->compressed_bytes, 1, sb.data());
      if ((rc = ld_offsets_upload(c, sb.data(), ns, &d_cb, &d_co, &total))) return rc;
      memcpy(c->cbr_key, key, sizeof key);
      c->cbr_total = total;
    }
    d_cb = (int32_t *)c->buf[B_CBRB].p; d_co = (uint32_t *)c->buf[B_CBRO].p; total = c->cbr_total;
    if (total > payload_stride) return set_err(c, VC2HIP_ECAP);
    LdEncParams p;
    c->ld_batch = n;
    if ((rc = fill_ld_enc(c, p, g, d_store, d_q, qm, ll, d_cb, d_co, cp->compressed_bytes / ns + 5, (uint8_t *)d_payload,
                          (long long)payload_stride))) return rc;
    p.search = 1;
    vc2_launch_ld_quantise(c->L, p, n, c->stream);
    vc2_launch_ld_pack(c->L, p, n, c->stream);
    vc2_launch_fill_u64(c->L, (unsigned long long *)d_lens, total, (size_t)n, c->stream);
    return VC2HIP_OK;
  }
  if (cp->mode == VC2HIP_HQ_CBR) {
      const int key[5] = {g.ys, g.xs, cp->compressed_bytes, cp->scalar, cp->prefix};
    if (memcmp(key, c->cbr_key, sizeof key) || !c->buf[B_CBRB].p) {
      std::vector<int32_t> sb(ns);
      vc2hip_slice_bytes(g.ys, g.xs, cp->compressed_bytes, cp->scalar, sb.data());
      if ((rc = cbr_offsets_upload(c, sb.data(), ns, cp->prefix, &d_cb, &d_co, &total))) return rc;
      memcpy(c->cbr_key, key, sizeof key);
      c->cbr_total = total;
    }
    d_cb = (int32_t *)c->buf[B_CBRB].p; d_co = (uint32_t *)c->buf[B_CBRO].p; total = c->cbr_total;
    if (total > payload_stride) return set_err(c, VC2HIP_ECAP);
    CbrParams p;
    memset(&p, 0, sizeof p);
    p.store = d_store; p.store_stride = (long long)ns * g.slice_coefs; p.qidx = d_q; p.slice_bytes = d_cb;
    p.store16 = s16; p.store_wide = d_storew;
    p.n_slices = ns; p.slice_coefs = g.slice_coefs;
    fill_comp_arrays(g, p.comp_n, p.comp_off, p.comp_n0);
    p.scalar = cp->scalar; p.err = c->d_err;
    p.general_only = c->cbr_general;
    for (int b = 0; b < 3 * g.depth + 1; ++b) p.qmatrix[b] = qm[b];
    p.n_bands = 3 * g.depth + 1;
    vc2_launch_cbr(c->L, p, n, c->stream);
  } else {
    if (payload_stride < vc2hip_max_payload_bytes(f, cp)) return set_err(c, VC2HIP_ECAP);
    // quantIndicesConstQ, EncodeStream.cpp:128-138
    vc2_launch_fill_i32(c->L, d_q, cp->q_index, (size_t)n * ns, c->stream);
  }
  return run_pack(c, g, n, d_store, d_q, qm, true, cp->prefix, cp->scalar, d_cb, d_co, total, (uint8_t *)d_payload,
                  (long long)payload_stride, (unsigned long long *)d_lens, s16, d_storew);
}

static int decode_batch_common(vc2hip_ctx *c, const void *d_payload, size_t payload_stride, const uint64_t *d_lens, int n,
                               const vc2hip_picture_format *f, const vc2hip_coding_params *cp, void *d_raw_out, bool ld) {
  if (!c || !d_payload || n < 1 || !f || !cp || !d_raw_out) return set_err(c, VC2HIP_EINVAL);
  if (((size_t)d_raw_out | (size_t)d_payload | payload_stride) & 15 || ((size_t)d_lens & 7))
    return set_err(c, VC2HIP_EINVAL, "device buffers and the payload stride must be 16-byte aligned");
  if (cp->kernel < 0 || cp->kernel > 6) return set_err(c, VC2HIP_EINVAL, "invalid wavelet kernel");
  ENTER(c);
  Geom g;
  int rc = picture_geom(g, f, cp, true);
  if (rc) return set_err(c, rc);
  const int ns = g.ys * g.xs;
  int32_t qm[VC2_MAX_BANDS];
  if ((rc = vc2hip_quant_matrix(cp->kernel, cp->depth, qm))) return set_err(c, rc);
  const bool s16 = !ld && use_store16(c, g, cp->kernel);
  const void *dstc[3]; long long ds[3];
  raw_planes(f, d_raw_out, dstc, ds);
  void *dst[3] = {(void *)dstc[0], (void *)dstc[1], (void *)dstc[2]};
  const bool plane_path = needs_plane_path(c, g, cp->kernel);
  // Band planes (vc2hip_internal.h): the finest levels, as long as they go through the streaming inverse kernel, a
  // slice's block row in them is at least 4 coefficients (8: rows that keep 16-byte pieces aligned) and they are not
  // the level whose LL comes from the store.  16-bit store only (the int32 store is the fallback decoder's).
  BandPlanes bp;
  memset(&bp, 0, sizeof bp);
  long long sstride = (long long)ns * g.slice_coefs; // elements per picture: slice records, then band planes
  if (s16 && !plane_path && c->allow_planes) {
    unsigned mask = 0;
    LLPlanes none;
    memset(&none, 0, sizeof none);
    unsigned tails = 0;
    (void)run_inverse(c, g, cp->kernel, n, nullptr, nullptr, qm, true, ld, none, dst, ds, true, f, s16, nullptr, nullptr, 0, &mask, nullptr, 1 << 30,
                      nullptr, &tails);
    for (int k = 0; k < 3; ++k) bp.from[k] = g.c[k].ph ? g.c[k].sh * g.c[k].sw : 0;
    for (int l = 0; l < VC2_BP_MAX && l < g.depth - 1 && (mask >> l & 1); ++l) {
      bool ok = true;
      for (int k = 0; k < 3 && ok; ++k) {
        const CompGeom &cg = g.c[k];
        if (!cg.ph) continue;
        const int bsh = (cg.sh >> l) / 2, bsw = (cg.sw >> l) / 2, ow = (cg.pw >> l) / 2;
        ok = bsh >= 1 && bsw >= 4 && (bsh & (bsh - 1)) == 0 && (bsw & (bsw - 1)) == 0 && ow % (bsw >= 8 ? 8 : 4) == 0 &&
             (bsh * bsw) % 8 == 0;
      }
      if (!ok) break;
      for (int k = 0; k < 3; ++k) {
        const CompGeom &cg = g.c[k];
        if (!cg.ph) continue;
        const int bsh = (cg.sh >> l) / 2, bsw = (cg.sw >> l) / 2;
        bp.np[k][l] = (cg.ph >> l) / 2; bp.ow[k][l] = (cg.pw >> l) / 2;
        bp.lbsh[k][l] = 31 - __builtin_clz(bsh); bp.lbsw[k][l] = 31 - __builtin_clz(bsw);
        bp.base[k][l] = sstride;
        sstride += ((long long)3 * bp.np[k][l] * bp.ow[k][l] + 7) & ~7ll;
        bp.from[k] -= 3 * bsh * bsw;
      }
      bp.levels = l + 1;
    }
    if (sstride >= (1ll << 31)) { memset(&bp, 0, sizeof bp); sstride = (long long)ns * g.slice_coefs; } // 32-bit element offsets
    // One byte per plane coefficient?  What the previous batch of this context looked like decides (its lengths and its
    // escape count arrive through pinned memory behind an event: no wait here): small coefficients <=> few payload bits per
    // sample.  The planes keep their places and their wide elements: only the bytes of a plane's narrow elements halve.
    if (bp.levels) {
      // (the context's FIRST look is waited for -- once, at its second batch, while the first is all the GPU has to do
      // anyway: a caller that never synchronises between batches would otherwise keep the 16-bit planes for as long as it
      // runs ahead of the GPU; every later look is only taken when it has arrived.  Never on a caller's stream
      // (vc2hip_create_on_stream): that caller may be capturing a graph or deliberately running ahead of the GPU -- there
      // the look is only ever queried, and VC2HIP_FLAG_PLANES8_ALWAYS / _NEVER make the choice deterministic: include/vc2hip.h)
      if (c->stat_pending && !c->stat_seen && c->own_stream) (void)hipEventSynchronize(c->stat_ev);
      if (c->stat_pending && hipEventQuery(c->stat_ev) == hipSuccess) {
        c->stat_pending = false;
        c->stat_seen = true;
        double bytes = 0;
        for (int k = 0; k < c->stat_n; ++k) bytes += (double)c->h_stat[1 + k];
        const double bits = c->stat_n ? 8.0 * bytes / (c->stat_n * c->stat_samples) : 99.0;
        // (pieces of eight coefficients; counted over ALL pictures of the batch, not only the stat_n whose lengths were copied)
        const double esc = (double)c->h_stat[0] * 8.0 / (std::max(1, c->stat_n_total) * c->stat_samples);
#ifdef VC2HIP_ABLATE
        if (getenv("VC2HIP_PLANES8_DEBUG")) fprintf(stderr, "planes8: previous batch %d pictures, %.2f payload bits per sample, %s planes, escape pieces %.4f%%\n", c->stat_n, bits, c->stat_was8 ? "byte" : "16-bit", 100 * esc);
#endif
        if (c->stat_was8 && esc > 0.005) c->planes8_on = false;       // more than 0.5 % of the pieces carried an escape
        else if (!c->stat_was8 || esc < 0.002) c->planes8_on = bits < (c->planes8_on ? 6.5 : 6.0);
      }
      bool b8 = c->planes8_mode == 1 || (c->planes8_mode == 0 && c->planes8_on);
      for (int l = 0; l < bp.levels; ++l) if (tails >> l & 1) b8 = false; // (the TAIL instantiations read 16-bit planes only)
      bp.bytes8 = b8;
    }
  }
  // Record heads (HeadSplit, vc2hip_internal.h): the levels below the streaming ones, when all of them run on the tile
  // kernels, read their coefficients from dense per-component arrays behind the records (and band planes)
  HeadSplit hs;
  memset(&hs, 0, sizeof hs);
  int head_level = 1 << 30;
  if (s16 && !plane_path && c->allow_heads) {
    unsigned smask = 0, fmask = 0;
    LLPlanes none;
    memset(&none, 0, sizeof none);
    (void)run_inverse(c, g, cp->kernel, n, nullptr, nullptr, qm, true, ld, none, dst, ds, true, f, s16, nullptr, nullptr, 0, &smask, nullptr, 1 << 30, &fmask);
    int ls = g.depth; // first level of the run of tile-kernel levels that reaches the deepest one
    while (ls > 0 && !(smask >> (ls - 1) & 1) && (fmask >> (ls - 1) & 1)) --ls;
    bool ok = ls >= 1 && ls < g.depth && ls >= bp.levels; // (level 0 keeps its bands where the final kernels expect them)
    for (int k = 0; k < 3 && ok; ++k) {
      if (!g.c[k].ph) continue;
      const int hn = (g.c[k].sh >> ls) * (g.c[k].sw >> ls);
      ok = hn >= 8 && hn % 8 == 0 && hn <= bp.from[k];
    }
    if (ok) {
      long long at = sstride;
      for (int k = 0; k < 3; ++k) {
        if (!g.c[k].ph) continue;
        hs.n[k] = (g.c[k].sh >> ls) * (g.c[k].sw >> ls);
        hs.base[k] = at;
        at += ((long long)ns * hs.n[k] + 7) & ~7ll;
      }
      if (at < (1ll << 31)) { sstride = at; head_level = ls; } else memset(&hs, 0, sizeof hs);
    }
  }
  int32_t *d_store, *d_ll, *d_q, *d_storew = nullptr, *d_llw = nullptr;
  {
    static const int pad = vc2_tune_int("VC2HIP_DEC_STORE_PAD", 0); // (experiment: elements between the pictures' stores; a multiple of 8)
    if (s16 && pad > 0 && sstride + pad < (1ll << 31)) sstride += pad & ~7;
  }
  NEED(c, B_STORE, std::max((size_t)n * ns * g.slice_coefs * 4, (size_t)n * (size_t)sstride * 2), d_store); // (16-bit elements: records, band planes and record heads)
  NEED(c, B_LL0, ll_bytes(g, n) + 16, d_ll);
  if (s16) {
    NEED(c, B_STOREW, (size_t)n * sstride * 4, d_storew);
    NEED(c, B_LLW, ll_bytes(g, n) + 16, d_llw);
  }
  NEED(c, B_QIDX, (size_t)n * ns * 4, d_q);
  LLPlanes ll;
  ll_layout(g, n, d_ll, s16 ? 2 : 4, d_llw, ll);
  if (!ld) {
    if (!d_lens || cp->scalar < 1 || cp->prefix < 0) return set_err(c, VC2HIP_EINVAL);
    uint32_t *d_offs;
    const int32_t *d_cb = nullptr; const uint32_t *d_co = nullptr; uint64_t cbr_total = 0;
    if (cp->mode == VC2HIP_HQ_CBR && cp->compressed_bytes > 0 && c->allow_cbr_index) { // the caller says CBR: the budgets predict the offsets
      const int key[5] = {g.ys, g.xs, cp->compressed_bytes, cp->scalar, cp->prefix};
      if (memcmp(key, c->cbr_key, sizeof key) || !c->buf[B_CBRB].p) {
        std::vector<int32_t> sb(ns);
        int32_t *db; uint32_t *dof; uint64_t total;
        if (vc2hip_slice_bytes(g.ys, g.xs, cp->compressed_bytes, cp->scalar, sb.data()) == VC2HIP_OK &&
            cbr_offsets_upload(c, sb.data(), ns, cp->prefix, &db, &dof, &total) == VC2HIP_OK) {
          memcpy(c->cbr_key, key, sizeof key);
          c->cbr_total = total;
        }
      }
      if (!memcmp(key, c->cbr_key, sizeof key)) { d_cb = (const int32_t *)c->buf[B_CBRB].p; d_co = (const uint32_t *)c->buf[B_CBRO].p; cbr_total = c->cbr_total; }
    }
    if ((rc = build_index(c, (const uint8_t *)d_payload, (long long)payload_stride, (const unsigned long long *)d_lens, n, ns,
                          cp->prefix, cp->scalar, &d_offs, d_cb, d_co, cbr_total))) return rc;
    UnpackParams p;
    memset(&p, 0, sizeof p);
    p.payload = (const uint8_t *)d_payload; p.payload_stride = (long long)payload_stride;
    p.lens = (const unsigned long long *)d_lens; p.offsets = d_offs;
    p.store = d_store; p.store_stride = sstride; p.qidx = d_q;
    p.store16 = s16; p.store_wide = d_storew;
    p.n_slices = ns; p.slice_coefs = g.slice_coefs;
    int n0[3];
    fill_comp_arrays(g, p.comp_n, p.comp_off, n0);
    p.prefix = cp->prefix; p.scalar = cp->scalar; p.err = c->d_err;
    p.bp = bp; p.hs = hs; p.xs = g.xs;
    c->last_plane_bits = bp.levels ? (bp.bytes8 ? 8 : 16) : 0;
    p.stats = c->d_stat;
    // (no look while the caller's stream is being captured into a graph: the copies and the event would become graph nodes
    // and the event could never be queried)
    bool capturing = false;
    if (!c->own_stream) {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      capturing = hipStreamIsCapturing(c->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
    }
    const bool track = bp.levels && c->planes8_mode == 0 && !c->stat_pending && !capturing;
    if (track) HIPCHK(c, hipMemsetAsync(c->d_stat, 0, 8, c->stream));
    vc2_launch_unpack(c->L, p, n, c->stream);
    if (track) { // this batch's escape count and payload lengths for the next call's choice
      c->stat_n = std::min(n, 60);
      c->stat_n_total = n;
      c->stat_samples = 0;
      for (int k = 0; k < 3; ++k) c->stat_samples += (double)g.c[k].h * g.c[k].w;
      c->stat_was8 = bp.bytes8 != 0;
      HIPCHK(c, hipMemcpyAsync(c->h_stat, c->d_stat, 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(c->h_stat + 1, d_lens, (size_t)c->stat_n * 8, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipEventRecord(c->stat_ev, c->stream));
      c->stat_pending = true;
    }
  } else {
    // DecodeStream.cpp:312, :331-333: per-slice sizes from the picture byte budget
    int32_t *d_sb; uint32_t *d_so; uint64_t total;
    const int key[5] = {g.ys, g.xs, cp->compressed_bytes, 1, -7};
    if (memcmp(key, c->cbr_key, sizeof key) || !c->buf[B_CBRB].p) {
      std::vector<int32_t> sb(ns);
      vc2hip_slice_bytes(g.ys, g.xs, cp->compressed_bytes, 1, sb.data());
      if ((rc = ld_offsets_upload(c, sb.data(), ns, &d_sb, &d_so, &total))) return rc;
      memcpy(c->cbr_key, key, sizeof key);
      c->cbr_total = total;
    }
    d_sb = (int32_t *)c->buf[B_CBRB].p; d_so = (uint32_t *)c->buf[B_CBRO].p; total = c->cbr_total;
    if (total > payload_stride) return set_err(c, VC2HIP_ESTREAM);
    LdUnpackParams p;
    fill_ld_unpack(p, g, (const uint8_t *)d_payload, (long long)payload_stride, d_sb, d_so, d_store, d_q, c->d_err);
    // slices whose luma length exceeds the slice (corrupt pictures): flag, serial walk, second pass (LdUnpackParams)
    unsigned *d_shifted; uint32_t *d_starts;
    NEED(c, B_SIZES, (size_t)n * 4, d_shifted);
    NEED(c, B_OFFS, (size_t)n * ns * 4, d_starts);
    int flag_fill = 0;
#ifdef VC2HIP_ABLATE
    if (getenv("VC2HIP_DEBUG_LD_REDO")) flag_fill = 1; // every picture takes the walk and the second pass (tests of that path on valid streams)
#endif
    HIPCHK(c, hipMemsetAsync(d_shifted, flag_fill, (size_t)n * 4, c->stream));
    vc2_prof_break(c->L);
    p.lens = (const unsigned long long *)d_lens; p.shifted = d_shifted; p.starts = d_starts;
    vc2_launch_ld_unpack(c->L, p, n, c->stream);
    vc2_launch_ld_walk(c->L, p, n, c->stream);
    p.redo = 1;
    vc2_launch_ld_unpack(c->L, p, n, c->stream);
    LdLl3Params lp;
    lp.store = d_store; lp.store_stride = (long long)ns * g.slice_coefs; lp.slice_coefs = g.slice_coefs;
    lp.ys = g.ys; lp.xs = g.xs; lp.qidx = d_q; lp.qm0 = qm[0]; lp.err = c->d_err;
    for (int k = 0; k < 3; ++k) {
      lp.coef_off[k] = g.c[k].coef_off; lp.llh[k] = g.c[k].ph >> g.depth; lp.llw[k] = g.c[k].pw >> g.depth;
      lp.ll_plane[k] = (int32_t *)ll.p[g.depth][k]; lp.ll_stride[k] = ll.stride[g.depth][k];
    }
    if (!vc2_launch_ld_ll3(c->L, lp, n, c->stream))
    for (int k = 0; k < 3; ++k)
      vc2_launch_ld_ll(c->L, d_store, (long long)ns * g.slice_coefs, g.slice_coefs, g.c[k].coef_off, g.c[k].n0,
                       g.c[k].ph >> g.depth, g.c[k].pw >> g.depth, g.ys, g.xs, d_q, qm[0], (int32_t *)ll.p[g.depth][k],
                       ll.stride[g.depth][k], n, c->d_err, c->stream);
  }
  if (plane_path) return plane_inverse(c, g, cp->kernel, n, d_store, d_q, qm, dst, ds, f, ld ? &ll : nullptr);
  return run_inverse(c, g, cp->kernel, n, d_store, d_q, qm, true, ld, ll, dst, ds, true, f, s16, d_storew, &bp, sstride, nullptr,
                     hs.n[0] ? &hs : nullptr, head_level);
}

extern "C" int vc2hip_band_plane_bits(const vc2hip_ctx *c) { return c ? c->last_plane_bits : 0; }

extern "C" int vc2hip_decode_batch_dev(vc2hip_ctx *c, const void *d_payload, size_t payload_stride, const uint64_t *d_lens,
                                       int n, const vc2hip_picture_format *f, const vc2hip_coding_params *cp, void *d_raw_out) {
  if (c && c->lanes.size() > 1 && n > 1 && !c->in_split && d_payload && f && cp && d_raw_out) {
    const size_t rb = vc2hip_raw_picture_bytes(f);
    const uint8_t *out8 = (const uint8_t *)d_raw_out, *pay8 = (const uint8_t *)d_payload, *len8 = (const uint8_t *)d_lens;
    return split_batch(c, n,
      [&](int first, int count, vc2hip_ctx::LaneUse &u) {
        u.r[0] = {pay8 + (size_t)first * payload_stride, pay8 + (size_t)(first + count) * payload_stride};
        u.r[1] = {len8 ? len8 + (size_t)first * 8 : nullptr, len8 ? len8 + (size_t)(first + count) * 8 : nullptr};
        u.w[0] = {out8 + (size_t)first * rb, out8 + (size_t)(first + count) * rb}; u.w[1] = {nullptr, nullptr};
      },
      [&](vc2hip_ctx *l, int first, int count) {
        return vc2hip_decode_batch_dev(l, pay8 + (size_t)first * payload_stride, payload_stride,
                                       d_lens ? d_lens + first : nullptr, count, f, cp, (uint8_t *)d_raw_out + (size_t)first * rb);
      });
  }
  return decode_batch_common(c, d_payload, payload_stride, d_lens, n, f, cp, d_raw_out, cp && cp->mode == VC2HIP_LD);
}

extern "C" int vc2hip_encode_picture_hq(vc2hip_ctx *c, const void *raw, const vc2hip_picture_format *f,
                                        const vc2hip_coding_params *cp, uint8_t *payload, size_t cap, size_t *len,
                                        int32_t *qidx_out) {
  if (!c || !raw || !f || !cp || !payload || !len) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  const size_t rb = vc2hip_raw_picture_bytes(f), pcap = (vc2hip_max_payload_bytes(f, cp) + 15) & ~(size_t)15;
  uint8_t *d_raw, *d_pay; unsigned long long *d_len;
  NEED(c, B_RAW, rb + 64, d_raw);
  NEED(c, B_PAYLOAD, pcap + 64, d_pay);
  NEED(c, B_LENS, 64, d_len);
  HIPCHK(c, hipMemcpyAsync(d_raw, raw, rb, hipMemcpyHostToDevice, c->stream));
  int rc = vc2hip_encode_batch_dev(c, d_raw, 1, f, cp, d_pay, pcap, (uint64_t *)d_len);
  if (rc) return rc;
  unsigned long long l = 0;
  HIPCHK(c, hipMemcpyAsync(&l, d_len, 8, hipMemcpyDeviceToHost, c->stream));
  if (qidx_out)
    HIPCHK(c, hipMemcpyAsync(qidx_out, c->buf[B_QIDX].p, (size_t)cp->y_slices * cp->x_slices * 4, hipMemcpyDeviceToHost, c->stream));
  if ((rc = vc2hip_sync(c))) return rc;
  if (l > cap) return set_err(c, VC2HIP_ECAP);
  HIPCHK(c, hipMemcpy(payload, d_pay, l, hipMemcpyDeviceToHost));
  *len = (size_t)l;
  return VC2HIP_OK;
}

extern "C" int vc2hip_encode_picture_ld(vc2hip_ctx *c, const void *raw, const vc2hip_picture_format *f,
                                        const vc2hip_coding_params *cp, uint8_t *payload, size_t cap, size_t *len,
                                        int32_t *qidx_out) {
  if (!cp || cp->mode != VC2HIP_LD) return set_err(c, VC2HIP_EINVAL);
  return vc2hip_encode_picture_hq(c, raw, f, cp, payload, cap, len, qidx_out);
}

static int decode_picture_host(vc2hip_ctx *c, const uint8_t *payload, size_t len, const vc2hip_picture_format *f,
                               const vc2hip_coding_params *cp, void *raw_out, bool ld) {
  if (!c || !payload || !f || !cp || !raw_out) return set_err(c, VC2HIP_EINVAL);
  ENTER(c);
  const size_t rb = vc2hip_raw_picture_bytes(f);
  const size_t stride = (len + 63) & ~(size_t)63;
  uint8_t *d_raw, *d_pay; unsigned long long *d_len;
  NEED(c, B_RAW, rb + 64, d_raw);
  NEED(c, B_PAYLOAD, stride + 64, d_pay);
  NEED(c, B_LENS, 64, d_len);
  unsigned long long l64 = len;
  HIPCHK(c, hipMemcpyAsync(d_pay, payload, len, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_len, &l64, 8, hipMemcpyHostToDevice, c->stream));
  int rc = decode_batch_common(c, d_pay, stride, (const uint64_t *)d_len, 1, f, cp, d_raw, ld);
  if (rc) return rc;
  HIPCHK(c, hipMemcpyAsync(raw_out, d_raw, rb, hipMemcpyDeviceToHost, c->stream));
  return vc2hip_sync(c);
}
extern "C" int vc2hip_decode_picture_hq(vc2hip_ctx *c, const uint8_t *payload, size_t len, const vc2hip_picture_format *f,
                                        const vc2hip_coding_params *cp, void *raw_out) {
  return decode_picture_host(c, payload, len, f, cp, raw_out, false);
}
extern "C" int vc2hip_decode_picture_ld(vc2hip_ctx *c, const uint8_t *payload, size_t len, const vc2hip_picture_format *f,
                                        const vc2hip_coding_params *cp, void *raw_out) {
  return decode_picture_host(c, payload, len, f, cp, raw_out, true);
}

// ------------------------------------------------------------------------------------------
// pipelined picture calls (include/vc2hip.h): two pictures in flight per context
// ------------------------------------------------------------------------------------------
extern "C" void *vc2hip_host_alloc(size_t bytes) {
  void *p = nullptr;
  return hipHostMalloc(&p, bytes ? bytes : 1) == hipSuccess ? p : nullptr;
}
extern "C" void vc2hip_host_free(void *p) { if (p) (void)hipHostFree(p); }

static int flight_begin(vc2hip_ctx *c, bool encode, vc2hip_ctx::Flight **out, int *ticket) {
  if (!ticket) return set_err(c, VC2HIP_EINVAL);
  HIPCHK(c, hipSetDevice(c->device));
  vc2hip_ctx::Flight &f = c->flight[c->flight_next];
  if (f.open) return set_err(c, VC2HIP_EINVAL, "too many pictures in flight (VC2HIP_MAX_INFLIGHT): end the oldest ticket first");
  if (!f.lane) {
    const int rc = vc2hip_create(c->device, &f.lane);
    if (rc) return set_err(c, rc, "cannot create a stream for a picture in flight");
  }
  if (!f.h_len) HIPCHK(c, hipHostMalloc((void **)&f.h_len, 64)); // (its own test: a failure here must not leave a lane without it)
  f.open = true;
  f.encode = encode;
  *ticket = c->flight_next;
  c->flight_next = (c->flight_next + 1) % VC2HIP_MAX_INFLIGHT;
  *out = &f;
  return VC2HIP_OK;
}

extern "C" int vc2hip_encode_picture_begin(vc2hip_ctx *c, const void *raw, const vc2hip_picture_format *f,
                                           const vc2hip_coding_params *cp, uint8_t *payload, size_t cap, int32_t *qidx_out,
                                           int *ticket) {
  if (!c || !raw || !f || !cp || !payload) return set_err(c, VC2HIP_EINVAL);
  vc2hip_ctx::Flight *fl;
  int rc = flight_begin(c, true, &fl, ticket);
  if (rc) return rc;
  vc2hip_ctx *l = fl->lane;
  fl->payload = payload; fl->cap = cap;
  const size_t rb = vc2hip_raw_picture_bytes(f), pcap = (vc2hip_max_payload_bytes(f, cp) + 15) & ~(size_t)15;
  uint8_t *d_raw, *d_pay; unsigned long long *d_len;
  auto fail = [&](int code) { fl->open = false; return set_err(c, code, l->err.c_str()); };
  if ((rc = need(l, B_RAW, rb + 64, (void **)&d_raw)) || (rc = need(l, B_PAYLOAD, pcap + 64, (void **)&d_pay)) ||
      (rc = need(l, B_LENS, 64, (void **)&d_len))) return fail(rc);
  if (hipMemcpyAsync(d_raw, raw, rb, hipMemcpyHostToDevice, l->stream) != hipSuccess) return fail(VC2HIP_EHIP);
  if ((rc = vc2hip_encode_batch_dev(l, d_raw, 1, f, cp, d_pay, pcap, (uint64_t *)d_len))) return fail(rc);
  if (hipMemcpyAsync(fl->h_len, d_len, 8, hipMemcpyDeviceToHost, l->stream) != hipSuccess) return fail(VC2HIP_EHIP);
  if (qidx_out && // (from vc2hip_host_alloc like every buffer of a _begin call: a pageable destination makes the call block)
      hipMemcpyAsync(qidx_out, l->buf[B_QIDX].p, (size_t)cp->y_slices * cp->x_slices * 4, hipMemcpyDeviceToHost, l->stream) != hipSuccess)
    return fail(VC2HIP_EHIP);
  return VC2HIP_OK;
}

extern "C" int vc2hip_encode_picture_end(vc2hip_ctx *c, int ticket, size_t *len) {
  if (!c || ticket < 0 || ticket >= VC2HIP_MAX_INFLIGHT || !len || !c->flight[ticket].open || !c->flight[ticket].encode)
    return set_err(c, VC2HIP_EINVAL);
  vc2hip_ctx::Flight &fl = c->flight[ticket];
  vc2hip_ctx *l = fl.lane;
  fl.open = false;
  int rc = vc2hip_sync(l); // the picture's stream: lengths are on the host, error flags read
  if (rc) return set_err(c, rc, l->err.c_str());
  const unsigned long long n = *fl.h_len;
  if (n > fl.cap) return set_err(c, VC2HIP_ECAP);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(fl.payload, l->buf[B_PAYLOAD].p, n, hipMemcpyDeviceToHost, l->stream)); // exactly the coded bytes
  HIPCHK(c, hipStreamSynchronize(l->stream));
  *len = (size_t)n;
  return VC2HIP_OK;
}

extern "C" int vc2hip_decode_picture_begin(vc2hip_ctx *c, const uint8_t *payload, size_t len, const vc2hip_picture_format *f,
                                           const vc2hip_coding_params *cp, void *raw_out, int *ticket) {
  if (!c || !payload || !f || !cp || !raw_out) return set_err(c, VC2HIP_EINVAL);
  vc2hip_ctx::Flight *fl;
  int rc = flight_begin(c, false, &fl, ticket);
  if (rc) return rc;
  vc2hip_ctx *l = fl->lane;
  const size_t rb = vc2hip_raw_picture_bytes(f), stride = (len + 63) & ~(size_t)63;
  uint8_t *d_raw, *d_pay; unsigned long long *d_len;
  auto fail = [&](int code) { fl->open = false; return set_err(c, code, l->err.c_str()); };
  if ((rc = need(l, B_RAW, rb + 64, (void **)&d_raw)) || (rc = need(l, B_PAYLOAD, stride + 64, (void **)&d_pay)) ||
      (rc = need(l, B_LENS, 64, (void **)&d_len))) return fail(rc);
  *fl->h_len = len;
  if (hipMemcpyAsync(d_pay, payload, len, hipMemcpyHostToDevice, l->stream) != hipSuccess ||
      hipMemcpyAsync(d_len, fl->h_len, 8, hipMemcpyHostToDevice, l->stream) != hipSuccess) return fail(VC2HIP_EHIP);
  if ((rc = decode_batch_common(l, d_pay, stride, (const uint64_t *)d_len, 1, f, cp, d_raw, cp->mode == VC2HIP_LD))) return fail(rc);
  if (hipMemcpyAsync(raw_out, d_raw, rb, hipMemcpyDeviceToHost, l->stream) != hipSuccess) return fail(VC2HIP_EHIP);
  return VC2HIP_OK;
}

extern "C" int vc2hip_decode_picture_end(vc2hip_ctx *c, int ticket) {
  if (!c || ticket < 0 || ticket >= VC2HIP_MAX_INFLIGHT || !c->flight[ticket].open || c->flight[ticket].encode)
    return set_err(c, VC2HIP_EINVAL);
  vc2hip_ctx::Flight &fl = c->flight[ticket];
  fl.open = false;
  const int rc = vc2hip_sync(fl.lane);
  return rc ? set_err(c, rc, fl.lane->err.c_str()) : VC2HIP_OK;
}
