// DataUnit.h -- VC-2 stream syntax kept on the host: parse info, sequence header (base video format
// matching + custom flags), picture header and transform parameters
// (/root/reference/src/Library/DataUnit.h, src/DataUnit.cpp:80-123, :236-266, :395-560, :563-1060,
// :1086-1144, :1203-1410).  Slice data inside a picture is produced / consumed by libvc2hip.
#ifndef VC2HOST_DATAUNIT_H
#define VC2HOST_DATAUNIT_H

#include <cstddef>
#include <iosfwd>
#include <vector>

#include "Picture.h"
#include "Utils.h"
#include "WaveletTransform.h"

enum DataUnitType { UNKNOWN_DATA_UNIT, SEQUENCE_HEADER, END_OF_SEQUENCE, AUXILIARY_DATA, PADDING_DATA,
                    HQ_PICTURE, LD_PICTURE, HQ_FRAGMENT, LD_FRAGMENT };
std::ostream &operator<<(std::ostream &os, DataUnitType t);

enum FrameRate { FR_UNSET = -1, FR0, FR24000_1001, FR24, FR25, FR30000_1001, FR30, FR50, FR60000_1001, FR60,
                 FR15000_1001, FR25_2, FR48, FR48_1001, FR96, FR100, FR120_1001, FR120 };
enum PixelAspectRatio { AR_UNSET = -1, AR0, AR1_1, AR10_11, AR12_11, AR40_33, AR16_11, AR4_3 };
enum ColorSpec { CS_UNSET = -1, CS_CUSTOM, CS_SDTV_525, CS_SDTV_625, CS_HDTV, CS_D_CINEMA, CS_UHDTV, CS_HDRTV_PQ,
                 CS_HDRTV_HLG };
const FrameRate MAX_V2_FRAMERATE = FR48;
enum Profile { PROFILE_UNKNOWN, PROFILE_LD, PROFILE_HQ };

// 13-byte parse info (DataUnit.cpp:80-123 out, :1111-1144 in)
struct DataUnit {
  DataUnitType type;
  unsigned long next_parse_offset, prev_parse_offset;
  DataUnit() : type(UNKNOWN_DATA_UNIT), next_parse_offset(0), prev_parse_offset(0) {}
  int length() const { return (int)next_parse_offset - 13; }
};
unsigned char parseCode(DataUnitType t);
void writeParseInfo(std::vector<unsigned char> &out, DataUnitType t, unsigned long next, unsigned long prev);
// parses 13 bytes; throws the reference's logic_error strings on a bad prefix / unknown type
DataUnit readParseInfo(const unsigned char *p);

struct SequenceHeader {
  SequenceHeader();
  SequenceHeader(Profile profile, int height, int width, ColourFormat chromaFormat, bool interlace,
                 FrameRate frameRate, bool topFieldFirst, int bitdepth, PixelAspectRatio pixelAspectRatio = AR_UNSET,
                 int cleanWidth = -1, int cleanHeight = -1, int leftOffset = -1, int topOffset = -1,
                 ColorSpec colorSpec = CS_UNSET, int colorPrimaries = 0, int colorMatrix = 0,
                 int transferFunction = 0, bool use_v3 = false);
  int major_version, minor_version;
  Profile profile;
  int width, height;
  ColourFormat chromaFormat;
  bool interlace;
  FrameRate frameRate;
  bool topFieldFirst;
  unsigned frameRateNumer, frameRateDenom;
  int bitdepth;
  int lumaExcursion, lumaOffset, colorDiffExcursion, colorDiffOffset;
  PixelAspectRatio pixelAspectRatio;
  unsigned pixelAspectRatioNumer, pixelAspectRatioDenom;
  int cleanWidth, cleanHeight, leftOffset, topOffset;
  ColorSpec colorSpec;
  int colorPrimaries, colorMatrix, transferFunction;
};
SequenceHeader getDefaultSourceParameters(int base_video_format_index); // DataUnit.cpp:428-460

// payload of a sequence-header data unit (DataUnit.cpp:563-881); returns the major version written
std::vector<unsigned char> writeSequenceHeader(const SequenceHeader &hdr, bool fragmented, int *major_version);
// DataUnit.cpp:883-1060 + :1203-1318
SequenceHeader readSequenceHeader(const unsigned char *p, std::size_t n, std::size_t *consumed);

struct PicturePreamble {
  WaveletKernel wavelet_kernel;
  int depth, slices_x, slices_y, slice_prefix, slice_size_scalar;
  utils::Rational slice_bytes;
};
// picture header + transform parameters of an HQ / LD picture (DataUnit.cpp:241-259 / :130-148)
std::vector<unsigned char> writePictureHeaderHQ(unsigned long picture_number, WaveletKernel kernel, int depth,
                                                int slices_x, int slices_y, int prefix, int scalar,
                                                int major_version);
std::vector<unsigned char> writePictureHeaderLD(unsigned long picture_number, WaveletKernel kernel, int depth,
                                                int slices_x, int slices_y, const utils::Rational &slice_bytes,
                                                int major_version);
// transform parameters alone, as the first fragment of a picture carries them: the two asymmetric-transform
// flags are always present there (DataUnit.cpp:159-176 / :270-287).  a,b = prefix,scalar (HQ) or the
// slice-bytes numerator,denominator (LD).
std::vector<unsigned char> writeTransformParams(WaveletKernel kernel, int depth, bool v3_flags, int slices_x,
                                                int slices_y, unsigned a, unsigned b);
// DataUnit.cpp:1340-1410 without the picture number; returns bytes consumed
std::size_t readTransformParams(const unsigned char *p, std::size_t n, bool low_delay, int major_version,
                                PicturePreamble *pre);

// fragment header that follows the parse info of an HQ/LD fragment (DataUnit.cpp:1146-1164 after the
// 4-byte picture number): data length, slice count and, when the count is non-zero, the slice offset
struct Fragment {
  unsigned long picture_number;
  unsigned length;
  int n_slices, slice_offset_x, slice_offset_y;
};
std::size_t readFragmentHeader(const unsigned char *p, std::size_t n, Fragment *frag);
// bytes of each slice of an HQ payload (prefix, index byte, three length-prefixed components,
// Slices.cpp:535-612) or of an LD payload (slice_bytes table); throws if the payload is short
std::vector<std::size_t> sliceSizesHQ(const unsigned char *payload, std::size_t len, int n_slices, int prefix, int scalar);
// all data units of one fragmented picture (DataUnit.cpp:156-232 / :267-342): the parameter fragment, then
// fragments of whole slices filled up to fragment_length bytes.  Chains prev_parse_offset.
void writeFragmentedPicture(std::vector<unsigned char> &out, bool low_delay, unsigned long picture_number,
                            const std::vector<unsigned char> &transform_params, const unsigned char *payload,
                            const std::vector<std::size_t> &slice_sizes, int slices_x, int fragment_length,
                            unsigned long *prev_parse_offset);
// DataUnit.cpp:1332-1410; low_delay selects the LD field layout; returns bytes consumed
std::size_t readPictureHeader(const unsigned char *p, std::size_t n, bool low_delay, int major_version,
                              unsigned long *picture_number, PicturePreamble *pre);
#endif
