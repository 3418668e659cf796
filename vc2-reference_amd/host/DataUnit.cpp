#include "DataUnit.h"

#include <ostream>
#include <sstream>
#include <stdexcept>

#include "VLC.h"

std::ostream &operator<<(std::ostream &os, DataUnitType t) {
  static const char *names[] = {"Unknown Data Unit", "Sequence Header", "End of Sequence", "Auxiliary Data",
                                "Padding Data", "HQ Picture", "LD Picture", "HQ Fragment", "LD Fragment"};
  return os << names[(int)t];
}

unsigned char parseCode(DataUnitType t) {
  switch (t) {
    case SEQUENCE_HEADER: return 0x00;
    case END_OF_SEQUENCE: return 0x10;
    case LD_PICTURE: return 0xC8;
    case HQ_PICTURE: return 0xE8;
    case HQ_FRAGMENT: return 0xEC;
    case LD_FRAGMENT: return 0xCC;
    default: return 0x20;
  }
}

void writeParseInfo(std::vector<unsigned char> &out, DataUnitType t, unsigned long next, unsigned long prev) {
  const unsigned char head[5] = {0x42, 0x42, 0x43, 0x44, parseCode(t)};
  out.insert(out.end(), head, head + 5);
  for (int i = 3; i >= 0; --i) out.push_back((unsigned char)(next >> (8 * i)));
  for (int i = 3; i >= 0; --i) out.push_back((unsigned char)(prev >> (8 * i)));
}

DataUnit readParseInfo(const unsigned char *p) {
  if (p[0] != 0x42 || p[1] != 0x42 || p[2] != 0x43 || p[3] != 0x44)
    throw std::logic_error("Read bytes do not match expected parse_info_header.");
  DataUnit d;
  switch (p[4]) {
    case 0x00: d.type = SEQUENCE_HEADER; break;
    case 0x10: d.type = END_OF_SEQUENCE; break;
    case 0x20: d.type = AUXILIARY_DATA; break;
    case 0x30: d.type = PADDING_DATA; break;
    case 0xC8: d.type = LD_PICTURE; break;
    case 0xE8: d.type = HQ_PICTURE; break;
    case 0xCC: d.type = LD_FRAGMENT; break;
    case 0xEC: d.type = HQ_FRAGMENT; break;
    default: throw std::logic_error("Stream Error: Unknown data unit type.");
  }
  d.next_parse_offset = ((unsigned long)p[5] << 24) | ((unsigned long)p[6] << 16) | ((unsigned long)p[7] << 8) | p[8];
  d.prev_parse_offset = ((unsigned long)p[9] << 24) | ((unsigned long)p[10] << 16) | ((unsigned long)p[11] << 8) | p[12];
  return d;
}

SequenceHeader::SequenceHeader()
    : major_version(1), minor_version(0), profile(PROFILE_UNKNOWN), width(0), height(0), chromaFormat(CF444),
      interlace(false), frameRate(FR0), topFieldFirst(false), frameRateNumer(0), frameRateDenom(0), bitdepth(0),
      lumaExcursion(0), lumaOffset(0), colorDiffExcursion(0), colorDiffOffset(0), pixelAspectRatio(AR_UNSET),
      pixelAspectRatioNumer(0), pixelAspectRatioDenom(0), cleanWidth(-1), cleanHeight(-1), leftOffset(-1),
      topOffset(-1), colorSpec(CS_UNSET), colorPrimaries(0), colorMatrix(0), transferFunction(0) {}

SequenceHeader::SequenceHeader(Profile p, int h, int w, ColourFormat cf, bool il, FrameRate fr, bool tff, int bd,
                               PixelAspectRatio par, int cw, int chh, int lo, int to, ColorSpec cs, int cp, int cm,
                               int tf, bool use_v3)
    : major_version(1), minor_version(0), profile(p), width(w), height(h), chromaFormat(cf), interlace(il),
      frameRate(fr), topFieldFirst(tff), frameRateNumer(0), frameRateDenom(0), bitdepth(bd), lumaExcursion(0),
      lumaOffset(0), colorDiffExcursion(0), colorDiffOffset(0), pixelAspectRatio(par), pixelAspectRatioNumer(0),
      pixelAspectRatioDenom(0), cleanWidth(cw), cleanHeight(chh), leftOffset(lo), topOffset(to), colorSpec(cs),
      colorPrimaries(cp), colorMatrix(cm), transferFunction(tf) {
  if (profile == PROFILE_HQ) major_version = 2;
  if (use_v3 || frameRate > MAX_V2_FRAMERATE || bitdepth > 12) major_version = 3;
}

// the table of base video formats, as data: {height,width,cf,interlace,frame rate,tff,depth,aspect,
// clean w,clean h,left,top,colour spec}
static const int BASE[23][13] = {
    {480, 640, CF420, 0, FR24000_1001, 0, 8, AR1_1, 640, 480, 0, 0, CS_CUSTOM},
    {120, 176, CF420, 0, FR15000_1001, 0, 8, AR10_11, 176, 120, 0, 0, CS_SDTV_525},
    {144, 176, CF420, 0, FR25_2, 1, 8, AR12_11, 176, 144, 0, 0, CS_SDTV_625},
    {240, 352, CF420, 0, FR15000_1001, 0, 8, AR10_11, 352, 240, 0, 0, CS_SDTV_525},
    {288, 352, CF420, 0, FR25_2, 1, 8, AR12_11, 352, 288, 0, 0, CS_SDTV_625},
    {480, 704, CF420, 0, FR15000_1001, 0, 8, AR10_11, 704, 480, 0, 0, CS_SDTV_525},
    {576, 704, CF420, 0, FR25_2, 1, 8, AR12_11, 704, 576, 0, 0, CS_SDTV_625},
    {480, 720, CF422, 1, FR30000_1001, 0, 10, AR10_11, 704, 480, 8, 0, CS_SDTV_525},
    {576, 720, CF422, 1, FR25, 1, 10, AR12_11, 704, 576, 8, 0, CS_SDTV_625},
    {720, 1280, CF422, 0, FR60000_1001, 1, 10, AR1_1, 1280, 720, 0, 0, CS_HDTV},
    {720, 1280, CF422, 0, FR50, 1, 10, AR1_1, 1280, 720, 0, 0, CS_HDTV},
    {1080, 1920, CF422, 1, FR30000_1001, 1, 10, AR1_1, 1920, 1080, 0, 0, CS_HDTV},
    {1080, 1920, CF422, 1, FR25, 1, 10, AR1_1, 1920, 1080, 0, 0, CS_HDTV},
    {1080, 1920, CF422, 0, FR60000_1001, 1, 10, AR1_1, 1920, 1080, 0, 0, CS_HDTV},
    {1080, 1920, CF422, 0, FR50, 1, 10, AR1_1, 1920, 1080, 0, 0, CS_HDTV},
    {1080, 2048, CF444, 0, FR24, 1, 12, AR1_1, 2048, 1080, 0, 0, CS_D_CINEMA},
    {2160, 4096, CF444, 0, FR24, 1, 12, AR1_1, 4096, 2160, 0, 0, CS_D_CINEMA},
    {2160, 3840, CF422, 0, FR60000_1001, 1, 10, AR1_1, 3840, 2160, 0, 0, CS_UHDTV},
    {2160, 3840, CF422, 0, FR50, 1, 10, AR1_1, 3840, 2160, 0, 0, CS_UHDTV},
    {4320, 7680, CF422, 0, FR60000_1001, 1, 10, AR1_1, 7680, 4320, 0, 0, CS_UHDTV},
    {4320, 7680, CF422, 0, FR50, 1, 10, AR1_1, 7680, 4320, 0, 0, CS_UHDTV},
    {1080, 1920, CF422, 0, FR24000_1001, 1, 10, AR1_1, 1920, 1080, 0, 0, CS_HDTV},
    {486, 720, CF422, 1, FR30000_1001, 0, 10, AR10_11, 720, 486, 0, 0, CS_HDTV}};

SequenceHeader getDefaultSourceParameters(int i) {
  if (i < 0 || i > 22) throw std::logic_error("DataUnitIO: unknown base video format");
  const int *b = BASE[i];
  return SequenceHeader(PROFILE_UNKNOWN, b[0], b[1], (ColourFormat)b[2], b[3] != 0, (FrameRate)b[4], b[5] != 0, b[6],
                        (PixelAspectRatio)b[7], b[8], b[9], b[10], b[11], (ColorSpec)b[12]);
}

namespace {
struct VideoFormat { // the coded fields of a sequence header
  int major_version = 0, minor_version = 0, profile = 0, level = 0, base_video_format = 0;
  bool custom_dimensions_flag = false; int frame_width = 0, frame_height = 0;
  bool custom_color_diff_format_flag = false; int color_diff_format = 0;
  bool custom_scan_format_flag = false; int source_sampling = 0;
  bool custom_frame_rate_flag = false; int frame_rate = 0; unsigned frame_rate_numer = 0, frame_rate_denom = 0;
  bool custom_pixel_aspect_ratio_flag = false; int pixel_aspect_ratio = 0; unsigned par_numer = 0, par_denom = 0;
  bool custom_clean_area_flag = false; int clean_width = 0, clean_height = 0, left_offset = 0, top_offset = 0;
  bool custom_signal_range_flag = false; int bitdepth = 0, luma_offset = 0, luma_excursion = 0, cd_offset = 0, cd_excursion = 0;
  bool custom_color_spec_flag = false; int color_spec = 0;
  bool custom_color_primaries_flag = false; int color_primaries = 0;
  bool custom_color_matrix_flag = false; int color_matrix = 0;
  bool custom_transfer_function_flag = false; int transfer_function = 0;
};

bool matchesAll(const SequenceHeader &f, int idx) { // DataUnit.cpp:486-503
  const SequenceHeader b = getDefaultSourceParameters(idx);
  return f.width == b.width && f.height == b.height && f.chromaFormat == b.chromaFormat && f.frameRate == b.frameRate &&
         f.bitdepth == b.bitdepth && f.interlace == b.interlace && f.topFieldFirst == b.topFieldFirst &&
         (f.pixelAspectRatio == -1 || f.pixelAspectRatio == b.pixelAspectRatio) &&
         (f.cleanWidth == -1 || f.cleanWidth == b.cleanWidth) && (f.cleanHeight == -1 || f.cleanHeight == b.cleanHeight) &&
         (f.leftOffset == -1 || f.leftOffset == b.leftOffset) && (f.topOffset == -1 || f.topOffset == b.topOffset) &&
         (f.colorSpec == -1 || f.colorSpec == b.colorSpec);
}
bool matches(const SequenceHeader &f, int w, int h, ColourFormat cf, FrameRate r, int bd, bool tff) { // :470-484
  return f.width == w && f.height == h && f.chromaFormat == cf && f.frameRate == r && f.bitdepth == bd &&
         f.topFieldFirst == tff;
}
int checkMatch(const SequenceHeader &f, int idx) { // DataUnit.cpp:505-530
  const SequenceHeader b = getDefaultSourceParameters(idx);
  const int n = (f.width != b.width) + (f.height != b.height) + (f.chromaFormat != b.chromaFormat) +
                (f.frameRate != b.frameRate) + (f.bitdepth != b.bitdepth) + (f.interlace != b.interlace) +
                (f.pixelAspectRatio != -1 && f.pixelAspectRatio != b.pixelAspectRatio) +
                (f.cleanWidth != -1 && f.cleanWidth != b.cleanWidth) + (f.cleanHeight != -1 && f.cleanHeight != b.cleanHeight) +
                (f.leftOffset != -1 && f.leftOffset != b.leftOffset) + (f.topOffset != -1 && f.topOffset != b.topOffset) +
                (f.colorSpec != -1 && f.colorSpec != b.colorSpec);
  return f.topFieldFirst == b.topFieldFirst ? n : -1;
}

VideoFormat toVideoFormat(const SequenceHeader &f) { // DataUnit.cpp:563-784
  VideoFormat v;
  v.major_version = f.major_version;
  v.minor_version = f.minor_version;
  v.profile = f.profile == PROFILE_HQ ? 3 : 0;
  auto set = [&](int base, int level) { v.base_video_format = base; v.level = level; };
  auto scan = [&](int base, int level) { set(base, level); v.custom_scan_format_flag = true; v.source_sampling = 0; };
  if (f.interlace) {
    if (matchesAll(f, 7)) set(7, 2);
    else if (matchesAll(f, 8)) set(8, 2);
    else if (matchesAll(f, 22)) set(22, 2);
    else if (f.chromaFormat == CF422 && f.width == 720 && f.height >= 480 && f.height <= 486 &&
             f.frameRate == FR30000_1001 && f.bitdepth == 10) {
      set(7, 2);
      v.custom_dimensions_flag = true; v.frame_width = f.width; v.frame_height = f.height;
    }
    else if (matchesAll(f, 11)) set(11, 3);
    else if (matchesAll(f, 12)) set(12, 3);
  } else {
    if (matchesAll(f, 1)) set(1, 1);
    else if (matchesAll(f, 2)) set(2, 1);
    else if (matchesAll(f, 3)) set(3, 1);
    else if (matchesAll(f, 4)) set(4, 1);
    else if (matchesAll(f, 5)) set(5, 1);
    else if (matchesAll(f, 6)) set(6, 1);
    else if (matches(f, 720, 480, CF422, FR30000_1001, 10, false)) scan(7, 2);
    else if (matches(f, 720, 576, CF422, FR25, 10, true)) scan(8, 2);
    else if (matches(f, 720, 486, CF422, FR30000_1001, 10, false)) scan(22, 2);
    else if (matchesAll(f, 9)) set(9, 3);
    else if (matchesAll(f, 10)) set(10, 3);
    else if (matches(f, 1920, 1080, CF422, FR30000_1001, 10, true)) scan(11, 3);
    else if (matches(f, 1920, 1080, CF422, FR25, 10, true)) scan(12, 3);
    else if (matchesAll(f, 13)) set(13, 3);
    else if (matchesAll(f, 14)) set(14, 3);
    else if (matchesAll(f, 21)) set(21, 3);
    else if (matchesAll(f, 15)) set(15, 4);
    else if (matches(f, 2048, 1080, CF444, FR48, 12, true)) { set(15, 4); v.custom_frame_rate_flag = true; v.frame_rate = FR48; }
    else if (matchesAll(f, 16)) set(16, 5);
    else if (matchesAll(f, 17)) set(17, 6);
    else if (matchesAll(f, 18)) set(18, 6);
    else if (matchesAll(f, 19)) set(19, 7);
    else if (matchesAll(f, 20)) set(20, 7);
  }
  if (v.base_video_format == 0) { // closest base format, differences as custom flags
    v.level = 0;
    int best = 999;
    for (int i = 1; i <= 22; ++i) {
      const int n = checkMatch(f, i);
      if (n == -1) continue;
      if (n < best) { v.base_video_format = i; best = n; }
    }
    const SequenceHeader b = getDefaultSourceParameters(v.base_video_format);
    if (f.interlace != b.interlace) { v.custom_scan_format_flag = true; v.source_sampling = f.interlace; }
    if (f.width != b.width || f.height != b.height) { v.custom_dimensions_flag = true; v.frame_width = f.width; v.frame_height = f.height; }
    if (f.chromaFormat != b.chromaFormat) { v.custom_color_diff_format_flag = true; v.color_diff_format = f.chromaFormat; }
    if (f.frameRate != b.frameRate) {
      v.custom_frame_rate_flag = true; v.frame_rate = f.frameRate;
      if (f.frameRate == FR0) { v.frame_rate_numer = f.frameRateNumer; v.frame_rate_denom = f.frameRateDenom; }
    }
    if (f.bitdepth != b.bitdepth) {
      v.custom_signal_range_flag = true;
      switch (f.bitdepth) {
        case 0: v.bitdepth = 0; v.luma_excursion = f.lumaExcursion; v.luma_offset = f.lumaOffset;
                v.cd_excursion = f.colorDiffExcursion; v.cd_offset = f.colorDiffOffset; break;
        case 8: v.bitdepth = 1; break;
        case 10: v.bitdepth = 3; break;
        case 12: v.bitdepth = 4; break;
        case 16: v.bitdepth = 7; break;
        default: throw std::logic_error("DataUnitIO: invalid bit depth");
      }
    }
    if (f.pixelAspectRatio != AR_UNSET && f.pixelAspectRatio != b.pixelAspectRatio) {
      v.custom_pixel_aspect_ratio_flag = true; v.pixel_aspect_ratio = f.pixelAspectRatio;
      if (v.pixel_aspect_ratio == 0) { v.par_numer = f.pixelAspectRatioNumer; v.par_denom = f.pixelAspectRatioDenom; }
    }
    const bool clean_set = f.cleanHeight != -1 || f.cleanWidth != -1 || f.leftOffset != -1 || f.topOffset != -1;
    if (clean_set && (f.cleanHeight != b.cleanHeight || f.cleanWidth != b.cleanWidth || f.leftOffset != b.leftOffset ||
                      f.topOffset != b.topOffset)) {
      v.custom_clean_area_flag = true; v.clean_height = f.cleanHeight; v.clean_width = f.cleanWidth;
      v.left_offset = f.leftOffset; v.top_offset = f.topOffset;
    } else if (v.custom_dimensions_flag && !clean_set) {
      v.custom_clean_area_flag = true; v.clean_height = v.frame_height; v.clean_width = v.frame_width;
      v.left_offset = 0; v.top_offset = 0;
    }
    if (f.colorSpec != CS_UNSET && f.colorSpec != b.colorSpec) { v.custom_color_spec_flag = true; v.color_spec = f.colorSpec; }
    if (f.colorSpec == CS_CUSTOM) {
      if (f.colorPrimaries != b.colorPrimaries) { v.custom_color_primaries_flag = true; v.color_primaries = f.colorPrimaries; }
      if (f.colorMatrix != b.colorMatrix) { v.custom_color_matrix_flag = true; v.color_matrix = f.colorMatrix; }
      if (f.transferFunction != b.transferFunction) { v.custom_transfer_function_flag = true; v.transfer_function = f.transferFunction; }
    }
  }
  return v;
}
} // namespace

std::vector<unsigned char> writeSequenceHeader(const SequenceHeader &hdr, bool fragmented, int *major_version) {
  VideoFormat f = toVideoFormat(hdr);
  if (fragmented && hdr.major_version < 3) f.major_version = 3; // DataUnit.cpp:1065-1067
  BitWriter w; // DataUnit.cpp:786-881
  w.putUnsignedVLC(f.major_version); w.putUnsignedVLC(f.minor_version); w.putUnsignedVLC(f.profile); w.putUnsignedVLC(f.level);
  w.putUnsignedVLC(f.base_video_format);
  w.putBoolean(f.custom_dimensions_flag);
  if (f.custom_dimensions_flag) { w.putUnsignedVLC(f.frame_width); w.putUnsignedVLC(f.frame_height); }
  w.putBoolean(f.custom_color_diff_format_flag);
  if (f.custom_color_diff_format_flag) w.putUnsignedVLC(f.color_diff_format);
  w.putBoolean(f.custom_scan_format_flag);
  if (f.custom_scan_format_flag) w.putUnsignedVLC(f.source_sampling);
  w.putBoolean(f.custom_frame_rate_flag);
  if (f.custom_frame_rate_flag) {
    w.putUnsignedVLC(f.frame_rate);
    if (f.frame_rate == FR0) { w.putUnsignedVLC(f.frame_rate_numer); w.putUnsignedVLC(f.frame_rate_denom); }
  }
  w.putBoolean(f.custom_pixel_aspect_ratio_flag);
  if (f.custom_pixel_aspect_ratio_flag) {
    w.putUnsignedVLC(f.pixel_aspect_ratio);
    if (f.pixel_aspect_ratio == AR0) { w.putUnsignedVLC(f.par_numer); w.putUnsignedVLC(f.par_denom); }
  }
  w.putBoolean(f.custom_clean_area_flag);
  if (f.custom_clean_area_flag) {
    w.putUnsignedVLC(f.clean_width); w.putUnsignedVLC(f.clean_height); w.putUnsignedVLC(f.left_offset); w.putUnsignedVLC(f.top_offset);
  }
  w.putBoolean(f.custom_signal_range_flag);
  if (f.custom_signal_range_flag) {
    w.putUnsignedVLC(f.bitdepth);
    if (f.bitdepth == 0) { w.putUnsignedVLC(f.luma_offset); w.putUnsignedVLC(f.luma_excursion); w.putUnsignedVLC(f.cd_offset); w.putUnsignedVLC(f.cd_excursion); }
  }
  w.putBoolean(f.custom_color_spec_flag);
  if (f.custom_color_spec_flag) {
    w.putUnsignedVLC(f.color_spec);
    if (f.color_spec == CS_CUSTOM) {
      w.putBoolean(f.custom_color_primaries_flag); if (f.custom_color_primaries_flag) w.putUnsignedVLC(f.color_primaries);
      w.putBoolean(f.custom_color_matrix_flag); if (f.custom_color_matrix_flag) w.putUnsignedVLC(f.color_matrix);
      w.putBoolean(f.custom_transfer_function_flag); if (f.custom_transfer_function_flag) w.putUnsignedVLC(f.transfer_function);
    }
  }
  w.putUnsignedVLC(f.source_sampling); // picture coding mode
  w.align();
  if (major_version) *major_version = f.major_version;
  return w.bytes();
}

SequenceHeader readSequenceHeader(const unsigned char *p, std::size_t n, std::size_t *consumed) {
  BitReader r(p, n);
  VideoFormat f;
  f.major_version = r.getUnsignedVLC(); f.minor_version = r.getUnsignedVLC(); f.profile = r.getUnsignedVLC(); f.level = r.getUnsignedVLC();
  f.base_video_format = r.getUnsignedVLC();
  if ((f.custom_dimensions_flag = r.getBoolean())) { f.frame_width = r.getUnsignedVLC(); f.frame_height = r.getUnsignedVLC(); }
  if ((f.custom_color_diff_format_flag = r.getBoolean())) f.color_diff_format = r.getUnsignedVLC();
  if ((f.custom_scan_format_flag = r.getBoolean())) f.source_sampling = r.getUnsignedVLC();
  if ((f.custom_frame_rate_flag = r.getBoolean())) {
    f.frame_rate = r.getUnsignedVLC();
    if (f.frame_rate == FR0) { f.frame_rate_numer = r.getUnsignedVLC(); f.frame_rate_denom = r.getUnsignedVLC(); }
  }
  if ((f.custom_pixel_aspect_ratio_flag = r.getBoolean())) {
    f.pixel_aspect_ratio = r.getUnsignedVLC();
    if (f.pixel_aspect_ratio == AR0) { f.par_numer = r.getUnsignedVLC(); f.par_denom = r.getUnsignedVLC(); }
  }
  if ((f.custom_clean_area_flag = r.getBoolean())) {
    f.clean_width = r.getUnsignedVLC(); f.clean_height = r.getUnsignedVLC(); f.left_offset = r.getUnsignedVLC(); f.top_offset = r.getUnsignedVLC();
  }
  if ((f.custom_signal_range_flag = r.getBoolean())) {
    f.bitdepth = r.getUnsignedVLC();
    if (f.bitdepth == 0) { f.luma_offset = r.getUnsignedVLC(); f.luma_excursion = r.getUnsignedVLC(); f.cd_offset = r.getUnsignedVLC(); f.cd_excursion = r.getUnsignedVLC(); }
  }
  if ((f.custom_color_spec_flag = r.getBoolean())) {
    f.color_spec = r.getUnsignedVLC();
    if (f.color_spec == CS_CUSTOM) {
      if ((f.custom_color_primaries_flag = r.getBoolean())) f.color_primaries = r.getUnsignedVLC();
      if ((f.custom_color_matrix_flag = r.getBoolean())) f.color_matrix = r.getUnsignedVLC();
      if ((f.custom_transfer_function_flag = r.getBoolean())) f.transfer_function = r.getUnsignedVLC();
    }
  }
  f.source_sampling = r.getUnsignedVLC();
  r.align();
  if (consumed) *consumed = r.bytePos();

  // copy_video_fmt_to_hdr, DataUnit.cpp:1203-1312
  const SequenceHeader b = getDefaultSourceParameters(f.base_video_format);
  SequenceHeader h = b;
  h.major_version = f.major_version;
  h.minor_version = f.minor_version;
  if (f.profile == 0) h.profile = PROFILE_LD; else if (f.profile == 3) h.profile = PROFILE_HQ;
  if (f.custom_dimensions_flag) { h.width = f.frame_width; h.height = f.frame_height; }
  if (f.custom_color_diff_format_flag) h.chromaFormat = (ColourFormat)f.color_diff_format;
  if (f.custom_scan_format_flag) h.interlace = f.source_sampling != 0;
  if (f.custom_frame_rate_flag) {
    h.frameRate = (FrameRate)f.frame_rate;
    if (f.frame_rate == FR0) { h.frameRateNumer = f.frame_rate_numer; h.frameRateDenom = f.frame_rate_denom; }
    if (f.frame_rate > MAX_V2_FRAMERATE && h.major_version < 3) h.major_version = 3;
  }
  if (f.custom_pixel_aspect_ratio_flag) {
    h.pixelAspectRatio = (PixelAspectRatio)f.pixel_aspect_ratio;
    if (f.pixel_aspect_ratio == AR0) { h.pixelAspectRatioNumer = f.par_numer; h.pixelAspectRatioDenom = f.par_denom; }
  }
  if (f.custom_clean_area_flag) { h.cleanWidth = f.clean_width; h.cleanHeight = f.clean_height; h.leftOffset = f.left_offset; h.topOffset = f.top_offset; }
  if (f.custom_signal_range_flag) {
    static const int depth_of[9] = {0, 8, 8, 10, 12, 10, 12, 16, 16};
    if (f.bitdepth >= 0 && f.bitdepth <= 8) h.bitdepth = depth_of[f.bitdepth];
    if (f.bitdepth == 0) { h.lumaOffset = f.luma_offset; h.lumaExcursion = f.luma_excursion; h.colorDiffOffset = f.cd_offset; h.colorDiffExcursion = f.cd_excursion; }
    if (f.bitdepth > 4 && h.major_version < 3) h.major_version = 3;
  }
  if (f.custom_color_spec_flag) {
    h.colorSpec = (ColorSpec)f.color_spec;
    if (f.color_spec == CS_CUSTOM) {
      if (f.custom_color_primaries_flag) h.colorPrimaries = f.color_primaries;
      if (f.custom_color_matrix_flag) h.colorMatrix = f.color_matrix;
      if (f.custom_transfer_function_flag) h.transferFunction = f.transfer_function;
    }
  }
  return h;
}

static void putTransformParams(BitWriter &w, WaveletKernel kernel, int depth, bool v3_flags, int slices_x, int slices_y,
                               unsigned a, unsigned b) {
  w.putUnsignedVLC((unsigned)kernel);
  w.putUnsignedVLC((unsigned)depth);
  if (v3_flags) { w.putBoolean(false); w.putBoolean(false); } // asym_transform_index_flag, asym_transform_flag
  w.putUnsignedVLC((unsigned)slices_x); w.putUnsignedVLC((unsigned)slices_y);
  w.putUnsignedVLC(a); w.putUnsignedVLC(b);
  w.putBoolean(false); // custom quantisation matrix
  w.align();
}

std::vector<unsigned char> writePictureHeaderHQ(unsigned long picture_number, WaveletKernel kernel, int depth,
                                                int slices_x, int slices_y, int prefix, int scalar, int major_version) {
  BitWriter w;
  w.putBytes(4, picture_number);
  putTransformParams(w, kernel, depth, major_version >= 3, slices_x, slices_y, (unsigned)prefix, (unsigned)scalar);
  return w.bytes();
}

std::vector<unsigned char> writePictureHeaderLD(unsigned long picture_number, WaveletKernel kernel, int depth,
                                                int slices_x, int slices_y, const utils::Rational &slice_bytes,
                                                int major_version) {
  BitWriter w;
  w.putBytes(4, picture_number);
  putTransformParams(w, kernel, depth, major_version >= 3, slices_x, slices_y, (unsigned)slice_bytes.numerator,
                     (unsigned)slice_bytes.denominator);
  return w.bytes();
}

std::vector<unsigned char> writeTransformParams(WaveletKernel kernel, int depth, bool v3_flags, int slices_x,
                                                int slices_y, unsigned a, unsigned b) {
  BitWriter w;
  putTransformParams(w, kernel, depth, v3_flags, slices_x, slices_y, a, b);
  return w.bytes();
}

static void readParams(BitReader &r, bool low_delay, int major_version, PicturePreamble *pre) {
  const unsigned wavelet_index = r.getUnsignedVLC();
  pre->depth = (int)r.getUnsignedVLC();
  pre->wavelet_kernel = wavelet_index <= 6 ? (WaveletKernel)wavelet_index : NullKernel;
  if (major_version >= 3) {
    if (r.getBoolean()) r.getUnsignedVLC(); // asym_transform_index_flag -> wavelet_index_ho
    if (r.getBoolean()) r.getUnsignedVLC(); // asym_transform_flag -> dwt_depth_ho
  }
  pre->slices_x = (int)r.getUnsignedVLC();
  pre->slices_y = (int)r.getUnsignedVLC();
  const int a = (int)r.getUnsignedVLC(), b = (int)r.getUnsignedVLC();
  if (low_delay) { pre->slice_prefix = 0; pre->slice_size_scalar = 0; pre->slice_bytes = utils::rationalise(a, b); }
  else { pre->slice_prefix = a; pre->slice_size_scalar = b; pre->slice_bytes = utils::rationalise(0, 1); }
  if (r.getBoolean()) throw std::logic_error("DataUnitIO: Custom Quantisation Matrix flag not supported");
  r.align();
}

std::size_t readPictureHeader(const unsigned char *p, std::size_t n, bool low_delay, int major_version,
                              unsigned long *picture_number, PicturePreamble *pre) {
  BitReader r(p, n);
  *picture_number = r.getBytes(4);
  readParams(r, low_delay, major_version, pre);
  return r.bytePos();
}

std::size_t readTransformParams(const unsigned char *p, std::size_t n, bool low_delay, int major_version,
                                PicturePreamble *pre) {
  BitReader r(p, n);
  readParams(r, low_delay, major_version, pre);
  return r.bytePos();
}

std::size_t readFragmentHeader(const unsigned char *p, std::size_t n, Fragment *frag) {
  BitReader r(p, n);
  frag->picture_number = r.getBytes(4);
  frag->length = (unsigned)r.getBytes(2);
  frag->n_slices = (int)r.getBytes(2);
  frag->slice_offset_x = frag->slice_offset_y = 0;
  if (frag->n_slices != 0) {
    frag->slice_offset_x = (int)r.getBytes(2);
    frag->slice_offset_y = (int)r.getBytes(2);
  }
  return r.bytePos();
}

std::vector<std::size_t> sliceSizesHQ(const unsigned char *payload, std::size_t len, int n_slices, int prefix, int scalar) {
  std::vector<std::size_t> sizes((std::size_t)n_slices);
  std::size_t pos = 0;
  for (int i = 0; i < n_slices; ++i) {
    std::size_t q = pos + (std::size_t)prefix + 1;
    for (int c = 0; c < 3; ++c) {
      if (q >= len) throw std::logic_error("slice data runs past the end of the picture");
      q += 1 + (std::size_t)payload[q] * (std::size_t)scalar;
    }
    if (q > len) throw std::logic_error("slice data runs past the end of the picture");
    sizes[(std::size_t)i] = q - pos;
    pos = q;
  }
  return sizes;
}

void writeFragmentedPicture(std::vector<unsigned char> &out, bool low_delay, unsigned long picture_number,
                            const std::vector<unsigned char> &transform_params, const unsigned char *payload,
                            const std::vector<std::size_t> &slice_sizes, int slices_x, int fragment_length,
                            unsigned long *prev_parse_offset) {
  const DataUnitType type = low_delay ? LD_FRAGMENT : HQ_FRAGMENT;
  auto bytes = [&](int n, unsigned long v) { for (int i = n - 1; i >= 0; --i) out.push_back((unsigned char)(v >> (8 * i))); };
  auto unit = [&](std::size_t data_size) {
    writeParseInfo(out, type, (unsigned long)data_size + 13, *prev_parse_offset);
    *prev_parse_offset = (unsigned long)data_size + 13;
  };
  unit(transform_params.size() + 8);
  bytes(4, picture_number); bytes(2, (unsigned long)transform_params.size()); bytes(2, 0);
  out.insert(out.end(), transform_params.begin(), transform_params.end());

  auto fragment = [&](std::size_t first_byte, std::size_t n_bytes, int nslices, int ox, int oy) {
    unit(n_bytes + 12);
    bytes(4, picture_number); bytes(2, (unsigned long)n_bytes); bytes(2, (unsigned long)nslices);
    bytes(2, (unsigned long)ox); bytes(2, (unsigned long)oy);
    out.insert(out.end(), payload + first_byte, payload + first_byte + n_bytes);
  };
  std::size_t frag_start = 0, frag_bytes = 0, pos = 0;
  int nslices = 0, ox = 0, oy = 0;
  for (std::size_t i = 0; i < slice_sizes.size(); ++i) {
    if (nslices > 0 && (long)(frag_bytes + slice_sizes[i]) > (long)fragment_length) {
      fragment(frag_start, frag_bytes, nslices, ox, oy);
      ox = (int)(i % (std::size_t)slices_x); oy = (int)(i / (std::size_t)slices_x);
      nslices = 0; frag_start = pos; frag_bytes = 0;
    }
    frag_bytes += slice_sizes[i];
    pos += slice_sizes[i];
    ++nslices;
  }
  fragment(frag_start, frag_bytes, nslices, ox, oy);
}
