// DecodeStream -- command-line compatible with /root/reference/src/DecodeStream (DecodeParams.cpp:39-92,
// DecodeStream.cpp:103-992: synchronise, data-unit dispatch, HQ and LD pictures, four output modes), with
// the per-picture body on MI355X through libvc2hip.  Extension: --gpus N decodes picture k on GPU k mod N.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <thread>

#include "Args.h"
#include "DataUnit.h"
#include "Hip.h"
#include "Picture.h"
#include "Quantisation.h"
#include "Slices.h"
#include "Utils.h"
#include "WaveletTransform.h"

using std::cerr; using std::clog; using std::cout; using std::endl; using std::string;

enum Output { TRANSFORM, QUANTISED, INDICES, DECODED };
static Output parseOutput(const string &t) {
  if (t == "Transform") return TRANSFORM;
  if (t == "Quantised") return QUANTISED;
  if (t == "Indices") return INDICES;
  if (t == "Decoded") return DECODED;
  throw std::invalid_argument("invalid output");
}
static void writeSigned4(std::ostream &os, const Array2D &a) {
  std::vector<unsigned char> b(a.num_elements() * 4);
  for (std::size_t i = 0; i < a.num_elements(); ++i) {
    const unsigned v = (unsigned)a.data()[i];
    b[4 * i] = (unsigned char)(v >> 24); b[4 * i + 1] = (unsigned char)(v >> 16);
    b[4 * i + 2] = (unsigned char)(v >> 8); b[4 * i + 3] = (unsigned char)v;
  }
  os.write((const char *)b.data(), (std::streamsize)b.size());
}
static void writePicture4(std::ostream &os, const Picture &p) { writeSigned4(os, p.y()); writeSigned4(os, p.c1()); writeSigned4(os, p.c2()); }

static const std::vector<ArgSpec> SPECS = {{'v', "verbose", false, ""}, {'o', "output", true, ""}, {'G', "gpus", true, ""}, {'h', "help", false, ""}};
static const char *USAGE = "DecodeStream (MI355X / libvc2hip)\nUsage: DecodeStream [-v] [-o Transform|Quantised|Indices|Decoded] [--gpus N] inFile outFile\n";

struct Job { // one picture waiting for its GPU
  bool ld;
  PicturePreamble pre;
  const unsigned char *data;
  std::size_t len;
  std::vector<unsigned char> raw;
  string error;
};

int main(int argc, char *argv[]) {
  try {
    if (argc < 2) { clog << USAGE; return EXIT_SUCCESS; }
    string inFileName, outFileName; bool verbose; Output output; int gpus;
    try {
      Args a(SPECS, argc, argv);
      if (a.isSet("help")) { cout << USAGE; return EXIT_SUCCESS; }
      if (a.positional.size() != 2) throw std::invalid_argument("Required arguments missing: inFile, outFile");
      inFileName = a.positional[0]; outFileName = a.positional[1];
      verbose = a.isSet("verbose");
      output = a.isSet("output") ? parseOutput(a.get("output")) : DECODED;
      gpus = a.getInt("gpus", 1);
      if (gpus < 1) throw std::invalid_argument("gpus must be >= 1");
    } catch (const std::exception &e) { cerr << "Command line error: " << e.what() << endl; return EXIT_FAILURE; }

    std::ifstream inFile; std::ofstream outFile;
    std::istream *in = &std::cin; std::ostream *out = &cout;
    if (inFileName != "-") { inFile.open(inFileName.c_str(), std::ios::binary); if (!inFile) { perror((string("Failed to open input file \"") + inFileName + "\"").c_str()); return EXIT_FAILURE; } in = &inFile; }
    if (outFileName != "-") { outFile.open(outFileName.c_str(), std::ios::binary); if (!outFile) { perror((string("Failed to open output file \"") + outFileName + "\"").c_str()); return EXIT_FAILURE; } out = &outFile; }
    const std::vector<unsigned char> s((std::istreambuf_iterator<char>(*in)), std::istreambuf_iterator<char>());

    // dataunitio::synchronise, DataUnit.cpp:1086-1109
    std::size_t pos = 0;
    while (pos + 4 <= s.size() && !(s[pos] == 0x42 && s[pos + 1] == 0x42 && s[pos + 2] == 0x43 && s[pos + 3] == 0x44)) ++pos;

    bool have_seq_hdr = false;
    int height = 0, width = 0, bytes = 0, depthBits = 0, major_version = 2;
    ColourFormat chromaFormat = CF_UNSET;
    int frame = 0;
    std::vector<Job> jobs;

    auto flush = [&]() { // decode the queued pictures, picture k on GPU k, and write them in order
      if (jobs.empty()) return;
      vc2hip_picture_format pf = {width, height, (int)chromaFormat, depthBits, bytes};
      std::vector<std::thread> th;
      for (std::size_t g = 0; g < jobs.size(); ++g)
        th.emplace_back([&, g]() {
          Job &j = jobs[g];
          try {
            vc2hip_ctx *c = hipContext((int)g);
            vc2hip_coding_params cp = {(int)j.pre.wavelet_kernel, j.pre.depth, j.pre.slices_y, j.pre.slices_x,
                                       j.ld ? VC2HIP_LD : VC2HIP_HQ_CONSTQ, 0,
                                       j.ld ? (j.pre.slice_bytes.numerator * j.pre.slices_y * j.pre.slices_x) / j.pre.slice_bytes.denominator : 0,
                                       j.pre.slice_prefix, j.pre.slice_size_scalar};
            j.raw.resize(vc2hip_raw_picture_bytes(&pf));
            hipCheck(c, (j.ld ? vc2hip_decode_picture_ld : vc2hip_decode_picture_hq)(c, j.data, j.len, &pf, &cp, j.raw.data()));
          } catch (const std::exception &e) { j.error = e.what(); }
        });
      for (auto &t : th) t.join();
      for (Job &j : jobs) {
        if (!j.error.empty()) throw std::logic_error(j.error);
        if (verbose) clog << "Writing decoded output file" << endl;
        out->write((const char *)j.raw.data(), (std::streamsize)j.raw.size());
        ++frame;
      }
      jobs.clear();
    };

    while (true) {
      if (pos >= s.size()) { flush(); clog << "End of data stream reached successfully, exiting." << endl; break; }
      if (pos + 13 > s.size()) { flush(); clog << "An error has occured in the data stream, exiting." << endl; return EXIT_FAILURE; }
      const DataUnit du = readParseInfo(&s[pos]);
      if (verbose) clog << endl << "Have read data unit of type: " << du.type << endl;
      const unsigned char *body = &s[0] + pos + 13;
      const std::size_t avail = s.size() - pos - 13;
      std::size_t used = 0;
      switch (du.type) {
        case SEQUENCE_HEADER: {
          flush();
          if (verbose) clog << "Parsing Sequence Header" << endl << endl;
          const SequenceHeader h = readSequenceHeader(body, avail, &used);
          if (verbose) {
            clog << "height        = " << h.height << endl << "width         = " << h.width << endl;
            clog << "chroma format = " << h.chromaFormat << endl << "interlaced    = " << std::boolalpha << h.interlace << endl;
          }
          if (h.interlace) throw std::logic_error("interlaced streams are not supported by the MI355X tools yet");
          height = h.height; width = h.width; chromaFormat = h.chromaFormat; depthBits = h.bitdepth;
          bytes = h.bitdepth == 8 ? 1 : 2; // DecodeStream.cpp:268-271
          major_version = h.major_version;
          have_seq_hdr = true;
          break;
        }
        case END_OF_SEQUENCE:
          if (verbose) clog << "End of Sequence after " << frame + (int)jobs.size() << " frames" << endl;
          break;
        case AUXILIARY_DATA:
          if (du.length() < 0) throw std::logic_error("Auxilliary data length is less than zero.");
          used = (std::size_t)du.length();
          break;
        case PADDING_DATA:
          if (du.length() < 0) throw std::logic_error("Padding data length is less than zero.");
          used = (std::size_t)du.length();
          break;
        case HQ_PICTURE:
        case LD_PICTURE: {
          const bool ld = du.type == LD_PICTURE;
          if (verbose) clog << "Parsing Picture Header" << endl;
          unsigned long picnum;
          PicturePreamble pre;
          const std::size_t hdr = readPictureHeader(body, avail, ld, major_version, &picnum, &pre);
          if (verbose) {
            clog << "Picture number      : " << picnum << endl << "Wavelet Kernel      : " << pre.wavelet_kernel << endl;
            clog << "Transform Depth     : " << pre.depth << endl << "Slices Horizontally : " << pre.slices_x << endl;
            clog << "Slices Verically    : " << pre.slices_y << endl;
          }
          // extent of the slice data: next_parse_offset when present, else the rest of the input
          const std::size_t unit = du.next_parse_offset ? (std::size_t)du.next_parse_offset - 13 : avail;
          const std::size_t dlen = (unit > hdr ? unit : hdr) - hdr;
          if (!have_seq_hdr) { clog << "Cannot decode frame, no previous sequence header!" << endl; used = unit; break; }
          if (output == DECODED) {
            Job j; j.ld = ld; j.pre = pre; j.data = body + hdr; j.len = dlen < avail - hdr ? dlen : avail - hdr;
            jobs.push_back(j);
            if ((int)jobs.size() == gpus) flush();
          } else { // diagnostic outputs through the fine-grained functions
            const int ph = paddedSize(height, pre.depth), pw = paddedSize(width, pre.depth);
            const PictureFormat tf(ph, pw, chromaFormat); // DecodeStream.cpp:483-498
            Picture q(tf);
            Array2D qIndices(pre.slices_y, pre.slices_x);
            if (ld) {
              const int compressed = (pre.slice_bytes.numerator * pre.slices_y * pre.slices_x) / pre.slice_bytes.denominator;
              unpackSlicesLD(body + hdr, dlen, q, pre.depth, qIndices, slice_bytes(pre.slices_y, pre.slices_x, compressed, 1), nullptr);
            } else unpackSlicesHQ(body + hdr, dlen, q, pre.depth, qIndices, pre.slice_prefix, pre.slice_size_scalar, nullptr);
            if (output == INDICES) { for (std::size_t i = 0; i < qIndices.num_elements(); ++i) out->put((char)qIndices.data()[i]); }
            else if (output == QUANTISED) writePicture4(*out, q);
            else {
              const Array1D qm = quantMatrix(pre.wavelet_kernel, pre.depth);
              writePicture4(*out, ld ? inverse_quantise_transform(q, qIndices, qm) : inverse_quantise_transform_np(q, qIndices, qm));
            }
            ++frame;
          }
          used = unit;
          break;
        }
        case HQ_FRAGMENT:
        case LD_FRAGMENT:
          throw std::logic_error("picture fragments are not supported by the MI355X tools yet");
        default:
          break;
      }
      if (du.type == END_OF_SEQUENCE && du.next_parse_offset == 0) { pos += 13; continue; }
      pos += du.next_parse_offset ? (std::size_t)du.next_parse_offset : 13 + used;
    }
    out->flush();
  } catch (const std::exception &ex) {
    cout << "Error: " << ex.what() << endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
