// hosttest -- CPU-only unit tests of the host layer, written after the reference's own gtest files
// (/root/reference/tests/Arrays.cpp, DataUnit.cpp, Utils.cpp): same known answers, same exception strings.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>

#include <algorithm>

#include "Arrays.h"
#include "Frame.h"
#include "Picture.h"
#include "Slices.h"
#include "WaveletTransform.h"
#include "DataUnit.h"
#include "Utils.h"
#include "VLC.h"

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

static std::string thrown(void (*f)()) { try { f(); } catch (const std::logic_error &e) { return e.what(); } return "<no exception>"; }

int main() {
  { // views, subbands, blocks (Arrays.h:17-50, WaveletTransform.cpp:428-476, Picture.cpp:231-271)
    Array2D a(8, 16);
    for (Index y = 0; y < 8; ++y) for (Index x = 0; x < 16; ++x) a[y][x] = (int)(100 * y + x);
    const View2D v(a, Range(1, 8, 2), Range(4, 16, 4));
    EXPECT(v.shape()[0] == 4 && v.shape()[1] == 3 && v.at(0, 0) == 104 && v.at(3, 2) == 712);
    const BlockVector bands = split_into_subbands(a, 2);
    EXPECT(bands.size() == 7 && bands[0].shape()[0] == 2 && bands[0].shape()[1] == 4);
    EXPECT(bands[0][1][1] == 404);                      // LL: rows / columns 0 mod 4
    EXPECT(bands[1][0][0] == 2 && bands[2][0][0] == 200 && bands[3][0][0] == 202);   // level 1: HL, LH, HH at stride 4, phase 2
    EXPECT(bands[4][0][0] == 1 && bands[5][0][0] == 100 && bands[6][3][7] == 715);   // level 2: stride 2, phase 1
    const Array2D back = merge_subbands(bands);
    bool same = back.shape()[0] == 8 && back.shape()[1] == 16;
    for (Index y = 0; same && y < 8; ++y) for (Index x = 0; x < 16; ++x) same = same && back[y][x] == a[y][x];
    EXPECT(same);
    Picture pic(PictureFormat(8, 16, CF422));
    pic.y(a);
    const PictureArray blocks = split_into_blocks(pic, 2, 4);
    EXPECT(blocks.shape()[0] == 2 && blocks.shape()[1] == 4 && blocks[1][3].y().shape()[1] == 4 && blocks[1][3].c1().shape()[1] == 2);
    EXPECT(blocks[1][3].y()[0][0] == 412);
    const Picture merged = merge_blocks(blocks);
    EXPECT(merged.y()[7][15] == 715 && merged.format().chromaWidth() == 8);
    // component_slice_bytes, Slices.cpp:97-119: bits through the last non-zero coefficient in subband order
    Array2D sl(4, 4);
    EXPECT(component_slice_bytes(sl, 1, 1) == 0);
    sl[0][0] = 1;                                        // LL first: one 4-bit code
    EXPECT(component_slice_bytes(sl, 1, 1) == 1 && component_slice_bytes(sl, 1, 3) == 3);
    sl[3][3] = -2;                                       // HH last: 4 LL + 4 HL + 4 LH + 3 HH zeros ... then 4 bits
    EXPECT(luma_slice_bits(sl, 1) == 4 + 3 + 4 + 4 + 3 + 4);
    EXPECT(component_slice_bytes(sl, 1, 2) == 4);
    EXPECT(thrown([] { Array2D big(16, 32); for (std::size_t i = 0; i < big.num_elements(); ++i) big.data()[i] = 30000; component_slice_bytes(big, 2, 1); }) ==
           "Slice scalar is too small, consider using a larger slice scalar.");
    // slice_bytes(v, h, ...), Slices.cpp:18-26: 1000 bytes over 3 x 2 slices
    int sum = 0;
    for (int vv = 0; vv < 3; ++vv) for (int h = 0; h < 2; ++h) sum += slice_bytes(vv, h, 3, 2, 1000, 6);
    EXPECT(sum == 1000 && slice_bytes(0, 0, 3, 2, 1000, 6) == 166 && slice_bytes(2, 1, 3, 2, 1000, 6) == 167);
  }
  { // the multi_array idioms of the reference's sources (Arrays.h:17-50): extents, indices, views as l- and r-values, ranges
    Array2D a(extents[6][8]);
    EXPECT(a.shape()[0] == 6 && a.shape()[1] == 8 && shape(a)[1] == 8);
    for (Index y = 0; y < 6; ++y) for (Index x = 0; x < 8; ++x) a[y][x] = (int)(10 * y + x);
    const Array2D odd = a[indices[Range(1, 6, 2)][Range()]];           // r-value: a dense copy of rows 1, 3, 5
    EXPECT(odd.shape()[0] == 3 && odd.shape()[1] == 8 && odd[0][0] == 10 && odd[2][7] == 57);
    Array2D b(a.ranges());                                               // same extents, zero filled
    EXPECT(b.shape()[0] == 6 && b.shape()[1] == 8 && b[5][7] == 0);
    b[indices[Range(0, 6, 2)][Range()]] = odd;                           // l-value: rows 0, 2, 4 take them
    EXPECT(b[0][3] == 13 && b[2][0] == 30 && b[4][7] == 57 && b[1][3] == 0);
    View2D win = b[indices[Range(2, 6, 2)][Range(1, 8, 3)]];
    EXPECT(win.shape()[0] == 2 && win.shape()[1] == 3 && win[0][0] == 31 && win[1][2] == 57);
    win[1][1] = -7;
    EXPECT(b[4][4] == -7);
    b.resize(extents[2][3]);
    EXPECT(b.num_elements() == 6 && b[0][2] == 12);
    Array1D m(extents[7]);
    EXPECT(m.size() == 7 && Array1D(m.ranges()).size() == 7);
    BlockArray blocks(extents[2][3]);
    blocks[1][2] = odd;
    EXPECT(shape(blocks)[0] == 2 && shape(blocks)[1] == 3 && blocks[1][2][2][7] == 57);
    BlockVector bands(extents[4]);
    EXPECT(bands.size() == 4);
    Array2D q(extents[3][5]);                                            // the fill idiom of EncodeStream.cpp:128-138
    std::fill(q.data(), q.data() + q.num_elements(), 21);
    EXPECT(q.shape()[0] == 3 && q.shape()[1] == 5 && q[2][4] == 21 && q[0][0] == 21);
    // Frame (Frame.h:18-40): fields are alternate rows; writing both fields back restores the frame
    Frame fr(PictureFormat(6, 8, CF420), true, false);
    Array2D y(extents[6][8]), u(extents[3][4]);
    for (Index r = 0; r < 6; ++r) for (Index c = 0; c < 8; ++c) y[r][c] = (int)(100 * r + c);
    for (Index r = 0; r < 3; ++r) for (Index c = 0; c < 4; ++c) u[r][c] = (int)(7 * r - c);
    Frame even(PictureFormat(6, 8, CF422), true, true);
    Array2D u2(extents[6][4]);
    for (Index r = 0; r < 6; ++r) for (Index c = 0; c < 4; ++c) u2[r][c] = (int)(9 * r + c);
    even.y(y); even.c1(u2); even.c2(u2);
    const Picture top = even.topField(), bot = even.bottomField();
    EXPECT(top.y().shape()[0] == 3 && top.y()[1][2] == 202 && bot.y()[1][2] == 302 && bot.c1()[2][3] == 48);
    EXPECT(even.firstField().y()[0][0] == 0 && even.secondField().y()[0][0] == 100);
    Frame rebuilt(PictureFormat(6, 8, CF422), true, true);
    rebuilt.firstField(top); rebuilt.secondField(bot);
    bool same = true;
    for (Index r = 0; r < 6; ++r) for (Index c = 0; c < 8; ++c) same = same && rebuilt.y()[r][c] == y[r][c];
    for (Index r = 0; r < 6; ++r) for (Index c = 0; c < 4; ++c) same = same && rebuilt.c2()[r][c] == u2[r][c];
    EXPECT(same);
    (void)fr; (void)u;
  }
  { // tests/Arrays.cpp:6-16
    Array2D a(3, 7);
    EXPECT(a.shape()[0] == 3 && a.shape()[1] == 7 && a.num_elements() == 21);
    a[2][6] = 5;
    Shape2D s = {{2, 9}};
    a.resize(s);
    EXPECT(a.shape()[0] == 2 && a.shape()[1] == 9 && a[1][8] == 0);
  }
  { // tests/DataUnit.cpp:18-53: parse-info known answers
    struct { unsigned type, next, prev; DataUnitType want; } cases[] = {
        {0x00, 0, 0, SEQUENCE_HEADER}, {0x10, 20, 20, END_OF_SEQUENCE}, {0x20, 128, 0, AUXILIARY_DATA},
        {0x30, 0, 3245, PADDING_DATA}, {0xC8, 23, 0, LD_PICTURE}, {0xE8, 13, 13, HQ_PICTURE},
        {0xCC, 13, 13, LD_FRAGMENT}, {0xEC, 13, 0, HQ_FRAGMENT}};
    for (auto &c : cases) {
      unsigned char b[13] = {'B', 'B', 'C', 'D', (unsigned char)c.type};
      for (int i = 0; i < 4; ++i) { b[5 + i] = (unsigned char)(c.next >> (24 - 8 * i)); b[9 + i] = (unsigned char)(c.prev >> (24 - 8 * i)); }
      const DataUnit du = readParseInfo(b);
      EXPECT(du.type == c.want && du.next_parse_offset == c.next && du.prev_parse_offset == c.prev);
      EXPECT(du.length() == (int)c.next - 13);
    }
    EXPECT(thrown([] { const unsigned char b[13] = {'A', 'B', 'C', 'D'}; readParseInfo(b); }) ==
           "Read bytes do not match expected parse_info_header.");
    EXPECT(thrown([] { const unsigned char b[13] = {'B', 'B', 'C', 'D', 0xFF}; readParseInfo(b); }) ==
           "Stream Error: Unknown data unit type.");
    // writer is the inverse
    std::vector<unsigned char> w;
    writeParseInfo(w, HQ_PICTURE, 0x01020304, 17);
    const DataUnit du = readParseInfo(w.data());
    EXPECT(w.size() == 13 && du.type == HQ_PICTURE && du.next_parse_offset == 0x01020304 && du.prev_parse_offset == 17);
  }
  { // tests/DataUnit.cpp:93-102
    const SequenceHeader t = getDefaultSourceParameters(4);
    EXPECT(t.width == 352 && t.height == 288 && t.chromaFormat == CF420 && t.frameRate == FR25_2 && t.bitdepth == 8);
  }
  { // tests/Utils.cpp:11-56
    EXPECT(utils::getPictureNumber(0, 0, 1) == 0 && utils::getPictureNumber(1, 0, 1) == 1 && utils::getPictureNumber(2, 0, 2) == 2);
    EXPECT(utils::getPictureNumber(1, 1, 1) == 2 && utils::getPictureNumber(2, 1, 2) == 4 && utils::getPictureNumber(1, 2, 2) == 5);
    EXPECT(utils::getPictureNumber(0, (1ULL << 32) - 1, 1) == (1ULL << 32) - 1 && utils::getPictureNumber(0, 1ULL << 32, 1) == 0);
    EXPECT(thrown([] { utils::getPictureNumber(-5, 0, 1); }) == "field number should be positive");
    EXPECT(thrown([] { utils::getPictureNumber(2, 0, 1); }) == "field number exceeds number of fields per frame");
    EXPECT(thrown([] { utils::getPictureNumber(0, 0, 3); }) == "number of fields per frame should be 1 (progressive) or 2 (interlaced)");
    EXPECT(utils::intlog2(1) == 0 && utils::intlog2(2) == 1 && utils::intlog2(3) == 2 && utils::intlog2(993) == 10);
    EXPECT(utils::rationalise(8294400, 16200).numerator == 512 && utils::rationalise(8294400, 16200).denominator == 1);
  }
  { // sequence header write -> parse round trip over formats that hit every branch of the matcher
    struct { int h, w; ColourFormat cf; FrameRate fr; int bd; } fm[] = {
        {1080, 1920, CF422, FR25, 10}, {2160, 3840, CF422, FR25, 10}, {2160, 3840, CF422, FR50, 10},
        {4320, 7680, CF444, FR25, 12}, {288, 352, CF420, FR25_2, 8}, {1080, 2048, CF444, FR48, 12},
        {100, 174, CF420, FR30, 8}, {1080, 1920, CF422, FR120, 10}, {720, 1280, CF422, FR50, 16}};
    for (auto &f : fm) {
      const SequenceHeader in(PROFILE_HQ, f.h, f.w, f.cf, false, f.fr, true, f.bd);
      int major = 0;
      const std::vector<unsigned char> b = writeSequenceHeader(in, false, &major);
      std::size_t used = 0;
      const SequenceHeader out = readSequenceHeader(b.data(), b.size(), &used);
      EXPECT(used == b.size());
      EXPECT(out.height == f.h && out.width == f.w && out.chromaFormat == f.cf && out.bitdepth == f.bd);
      EXPECT(out.frameRate == f.fr && !out.interlace && out.profile == PROFILE_HQ && out.major_version == major);
    }
    // cfg 1 of the benchmark: 4 coded bytes, base format 12 + custom scan format
    int major = 0;
    const std::vector<unsigned char> b = writeSequenceHeader(SequenceHeader(PROFILE_HQ, 1080, 1920, CF422, false, FR25, true, 10), false, &major);
    EXPECT(b.size() == 4 && major == 2);
  }
  { // picture header round trip, v2 and v3
    for (int major = 2; major <= 3; ++major) {
      const std::vector<unsigned char> b = writePictureHeaderHQ(0xDEADBEEF, DD97, 4, 120, 135, 2, 3, major);
      unsigned long pn; PicturePreamble pre;
      const std::size_t used = readPictureHeader(b.data(), b.size(), false, major, &pn, &pre);
      EXPECT(used == b.size() && pn == 0xDEADBEEF && pre.wavelet_kernel == DD97 && pre.depth == 4);
      EXPECT(pre.slices_x == 120 && pre.slices_y == 135 && pre.slice_prefix == 2 && pre.slice_size_scalar == 3);
    }
  }
  { // exp-Golomb header fields
    BitWriter w;
    for (unsigned v : {0u, 1u, 2u, 7u, 1920u, 65535u}) w.putUnsignedVLC(v);
    w.align();
    BitReader r(w.bytes().data(), w.bytes().size());
    for (unsigned v : {0u, 1u, 2u, 7u, 1920u, 65535u}) EXPECT(r.getUnsignedVLC() == v);
  }
  if (failures) { std::printf("%d failure(s)\n", failures); return 1; }
  std::printf("all host tests passed\n");
  return 0;
}
