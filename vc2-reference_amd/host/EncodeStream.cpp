// EncodeStream -- command-line compatible with /root/reference/src/EncodeStream (EncodeParams.cpp:55-249
// flags and validation, EncodeStream.cpp:247-788 flow and output modes), with the per-picture body
// running on MI355X through libvc2hip.  Extension: --gpus N encodes frame k on GPU k mod N.
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <mutex>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <thread>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "Args.h"
#include "DataUnit.h"
#include "Frame.h"
#include "Hip.h"
#include "Pipeline.h"
#include "Picture.h"
#include "Quantisation.h"
#include "Slices.h"
#include "Utils.h"
#include "WaveletTransform.h"

using std::cerr; using std::clog; using std::cout; using std::endl; using std::string;

enum Mode { HQ_ConstQ, HQ_CBR, LD };
enum Output { TRANSFORM, QUANTISED, INDICES, PACKAGED, STREAM, DECODED, PSNR };

static Mode parseMode(const string &t) {
  if (t == "HQ_ConstQ") return HQ_ConstQ;
  if (t == "HQ_CBR") return HQ_CBR;
  if (t == "LD") return LD;
  throw std::invalid_argument("invalid mode");
}
static Output parseOutput(const string &t) {
  if (t == "Transform") return TRANSFORM;
  if (t == "Quantised") return QUANTISED;
  if (t == "Indices") return INDICES;
  if (t == "Packaged") return PACKAGED;
  if (t == "Stream") return STREAM;
  if (t == "Decoded") return DECODED;
  if (t == "PSNR") return PSNR;
  throw std::invalid_argument("invalid output");
}
static const char *modeName(Mode m) { return m == HQ_ConstQ ? "HQ_ConstQ" : (m == HQ_CBR ? "HQ_CBR" : "LD"); }
static const char *outputName(Output o) {
  static const char *n[] = {"Transform", "Quantised", "Indices", "Packaged", "Stream", "Decoded", "PSNR"};
  return n[o];
}

static void writeSigned4(std::ostream &os, const Array2D &a) { // pictureio::wordWidth(4) + signed_binary
  std::vector<unsigned char> b(a.num_elements() * 4);
  for (std::size_t i = 0; i < a.num_elements(); ++i) {
    const unsigned v = (unsigned)a.data()[i];
    b[4 * i] = (unsigned char)(v >> 24); b[4 * i + 1] = (unsigned char)(v >> 16);
    b[4 * i + 2] = (unsigned char)(v >> 8); b[4 * i + 3] = (unsigned char)v;
  }
  os.write((const char *)b.data(), (std::streamsize)b.size());
}
static void writePicture4(std::ostream &os, const Picture &p) { writeSigned4(os, p.y()); writeSigned4(os, p.c1()); writeSigned4(os, p.c2()); }

static const std::vector<ArgSpec> SPECS = {
    {'v', "verbose", false, ""}, {'m', "mode", true, ""}, {'o', "output", true, ""}, {'a', "hSlice", true, ""},
    {'u', "vSlice", true, ""}, {'d', "waveletDepth", true, ""}, {'k', "kernel", true, ""},
    {'b', "bottomFieldFirst", false, ""}, {'t', "topFieldFirst", false, ""}, {'i', "interlace", false, ""},
    {'p', "progressive", false, ""}, {'c', "chromaDepth", true, ""}, {'l', "lumaDepth", true, ""},
    {'z', "bitDepth", true, ""}, {'n', "bytes", true, ""}, {'f', "format", true, ""}, {'x', "width", true, ""},
    {'y', "height", true, ""}, {'r', "framerate", true, ""}, {'S', "scalar", true, ""}, {'P', "prefix", true, ""},
    {'F', "fragmentLength", true, ""}, {'s', "compressedBytes", true, ""}, {'q', "quantIndex", true, ""},
    {'G', "gpus", true, ""}, {'D', "devices", true, ""}, {'h', "help", false, ""}};

static const char *USAGE =
    "EncodeStream (MI355X / libvc2hip)\n"
    "Usage: EncodeStream -m <HQ_ConstQ|HQ_CBR|LD> -k <kernel> -d <depth> -u <vSlice> -a <hSlice> -f <4:4:4|4:2:2|4:2:0>\n"
    "       -x <width> -y <height> [-l lumaDepth] [-c chromaDepth] [-z bitDepth] [-n bytes] [-r framerate]\n"
    "       [-i [-t|-b]] [-F fragmentLength] [-q quantIndex] [-s compressedBytes] [-S scalar] [-P prefix] [-o Transform|Quantised|Indices|Packaged|\n"
    "       Stream|Decoded|PSNR] [-v] [--gpus N | --devices a,b,..] inFile outFile      (\"-\" = standard input / output)\n";

int main(int argc, char *argv[]) {
  try {
    if (argc < 2) { clog << USAGE; return EXIT_SUCCESS; }
    // ---- parameters: EncodeParams.cpp:80-204 ----
    string inFileName, outFileName; bool verbose; int height, width, bytes, lumaDepth, chromaDepth; ColourFormat chromaFormat;
    WaveletKernel kernel; int waveletDepth, ySize, xSize; Output output; Mode mode; int frameRate, sliceScalar, slicePrefix;
    int fragmentLength, compressedBytes, qIndex, gpus; bool interlaced, topFieldFirst;
    std::vector<int> devices;
    try {
      Args a(SPECS, argc, argv);
      if (a.isSet("help")) { cout << USAGE; return EXIT_SUCCESS; }
      if (a.positional.size() != 2) throw std::invalid_argument("Required arguments missing: inFile, outFile");
      for (const char *r : {"mode", "hSlice", "vSlice", "waveletDepth", "kernel", "format", "width", "height"}) a.require(r);
      inFileName = a.positional[0]; outFileName = a.positional[1];
      verbose = a.isSet("verbose");
      height = a.getInt("height", 0); width = a.getInt("width", 0);
      chromaFormat = parseColourFormat(a.get("format"));
      bytes = a.getInt("bytes", 2);
      int bitDepth = a.getInt("bitDepth", 0);
      lumaDepth = a.getInt("lumaDepth", 0); chromaDepth = a.getInt("chromaDepth", 0);
      kernel = parseWaveletKernel(a.get("kernel"));
      waveletDepth = a.getInt("waveletDepth", 0); ySize = a.getInt("vSlice", 0); xSize = a.getInt("hSlice", 0);
      output = a.isSet("output") ? parseOutput(a.get("output")) : STREAM;
      mode = parseMode(a.get("mode"));
      frameRate = a.getInt("framerate", 3);
      sliceScalar = a.getInt("scalar", 1); slicePrefix = a.getInt("prefix", 0);
      fragmentLength = a.getInt("fragmentLength", 0); compressedBytes = a.getInt("compressedBytes", 0);
      qIndex = a.getInt("quantIndex", 0); gpus = a.getInt("gpus", 1);
      interlaced = a.isSet("interlace"); topFieldFirst = !a.isSet("bottomFieldFirst"); // EncodeParams.cpp:123-124
      if (a.isSet("bitDepth") && (a.isSet("lumaDepth") || a.isSet("chromaDepth")))
        throw std::invalid_argument("bitDepth is incompatible with luma depth (and/or chroma depth): use one or the other");
      if (a.isSet("progressive") && a.isSet("interlace"))
        throw std::invalid_argument("image can't be both interlaced and progressive: specify one or the other");
      if (a.isSet("progressive") && (a.isSet("topFieldFirst") || a.isSet("bottomFieldFirst")))
        throw std::invalid_argument("field parity is incompatible with progressive image");
      if (a.isSet("topFieldFirst") && a.isSet("bottomFieldFirst"))
        throw std::invalid_argument("image can't be both top field first and bottom field first: specify one or the other");
      if (!a.isSet("bitDepth")) bitDepth = 8 * bytes;
      if (!a.isSet("lumaDepth")) lumaDepth = bitDepth;
      if (!a.isSet("chromaDepth")) chromaDepth = lumaDepth;
      if (height < 1) throw std::invalid_argument("picture height must be > 0");
      if (width < 1) throw std::invalid_argument("picture width must be > 0");
      if (bytes < 1 || bytes > 4) throw std::invalid_argument("bytes must be in range 1 to 4");
      if (a.isSet("bitDepth")) { if (bitDepth < 1 || bitDepth > 8 * bytes) throw std::invalid_argument("bit depth must be in range 1 to 8*(bytes per sample)"); }
      else {
        if (lumaDepth < 1 || lumaDepth > 8 * bytes) throw std::invalid_argument("luma bit depth must be in range 1 to 8*(bytes per sample)");
        if (chromaDepth < 1 || chromaDepth > 8 * bytes) throw std::invalid_argument("chroma bit depth must be in range 1 to 8*(bytes per sample)");
      }
      if (kernel == NullKernel) throw std::invalid_argument("invalid wavelet kernel");
      if (waveletDepth < 1) throw std::invalid_argument("wavelet depth must be 1 or more");
      const bool hq = mode == HQ_CBR || mode == HQ_ConstQ;
      if (!hq && a.isSet("scalar")) throw std::invalid_argument("Slice Scalar is only used in HQ_CBR and HQ_ConstQ modes");
      if (!hq && a.isSet("prefix")) throw std::invalid_argument("Slice Prefix is only used in HQ_CBR and HQ_ConstQ modes");
      if (mode == HQ_ConstQ && a.isSet("fragmentLength")) throw std::invalid_argument("Fragment length is only used in HQ_CBR and LD modes");
      if (mode == HQ_ConstQ && a.isSet("compressedBytes")) throw std::invalid_argument("Compressed bytes is only used in HQ_CBR and LD modes");
      if (mode != HQ_ConstQ && a.isSet("quantIndex")) throw std::invalid_argument("Quantisation index is only used in HQ_ConstQ mode");
      if (mode != HQ_ConstQ && !a.isSet("compressedBytes")) throw std::invalid_argument("Compressed bytes must be set in HQ_CBR and LD modes");
      if (mode == HQ_ConstQ && !a.isSet("quantIndex")) throw std::invalid_argument("Quantisation index must be set in HQ_ConstQ mode");
      if (hq && sliceScalar < 1) throw std::invalid_argument("slice scalar must be >=1");
      if (hq && slicePrefix < 0) throw std::invalid_argument("slice prefix must be >=0");
      if (mode != HQ_ConstQ && compressedBytes < 1) throw std::invalid_argument("number of compressed bytes must be >0");
      if (mode == HQ_ConstQ && (qIndex < 0 || qIndex > 119)) throw std::invalid_argument("quantisation index must be in the range 0 to 119");
      if (gpus < 1) throw std::invalid_argument("gpus must be >= 1");
      if (a.isSet("devices")) { // explicit HIP devices of the workers (a device may repeat): overrides --gpus
        const string list = a.get("devices");
        for (std::size_t p0 = 0; p0 <= list.size();) {
          const std::size_t p1 = list.find(',', p0) == string::npos ? list.size() : list.find(',', p0);
          try { devices.push_back(std::stoi(list.substr(p0, p1 - p0))); }
          catch (...) { throw std::invalid_argument("Couldn't read argument value from string '" + list + "' for arg --devices"); }
          p0 = p1 + 1;
        }
      } else for (int g = 0; g < gpus; ++g) devices.push_back(g);
    } catch (const std::exception &e) {
      cerr << "Command line error: " << e.what() << endl;
      return EXIT_FAILURE;
    }

    // ---- streams ----
    std::ifstream inFile; std::ofstream outFile;
    std::istream *in = &std::cin; std::ostream *out = &cout;
    if (inFileName != "-") { inFile.open(inFileName.c_str(), std::ios::binary); if (!inFile) { perror((string("Failed to open input file \"") + inFileName + "\"").c_str()); return EXIT_FAILURE; } in = &inFile; }
    if (outFileName != "-") { outFile.open(outFileName.c_str(), std::ios::binary); if (!outFile) { perror((string("Failed to open output file \"") + outFileName + "\"").c_str()); return EXIT_FAILURE; } out = &outFile; }

    const PictureFormat format(height, width, chromaFormat);
    if (verbose) {
      clog << "mode= " << modeName(mode) << endl << "bytes per sample= " << bytes << endl;
      clog << "luma depth (bits) = " << lumaDepth << endl << "chroma depth (bits) = " << chromaDepth << endl;
      clog << "height = " << height << endl << "width = " << width << endl << "chroma format = " << chromaFormat << endl;
      clog << "interlaced = " << std::boolalpha << interlaced << endl;
      if (interlaced) clog << "top field first = " << std::boolalpha << topFieldFirst << endl;
      clog << "wavelet kernel = " << kernel << endl << "wavelet depth = " << waveletDepth << endl;
      clog << "vertical slice size (in units of 2**(wavelet depth)) = " << ySize << endl;
      clog << "horizontal slice size (in units of 2**(wavelet depth)) = " << xSize << endl;
      clog << "compressed bytes = " << compressedBytes << endl << "output = " << outputName(output) << endl;
    }
    // a picture is a frame or one field of it (EncodeStream.cpp:367-377)
    const int framePics = interlaced ? 2 : 1;
    const PictureFormat picFormat = interlaced ? fieldFormat(format) : format;
    const int ySlices = sliceSizeIsValid(waveletDepth, picFormat.lumaHeight(), picFormat.chromaHeight(), ySize);
    const int xSlices = sliceSizeIsValid(waveletDepth, picFormat.lumaWidth(), picFormat.chromaWidth(), xSize);
    if (ySlices == 0 || xSlices == 0)
      throw std::logic_error("The given waveletDepth, hSlice, and vSlice parameters cannot encode this input. See above for suggested parameters.");
    const int pictureBytes = interlaced ? compressedBytes / 2 : compressedBytes;
    if (verbose) {
      clog << "Vertical slices per picture          = " << ySlices << endl;
      clog << "Horizontal slices per picture        = " << xSlices << endl;
      if (mode == HQ_CBR) {
        const utils::Rational r = utils::rationalise(pictureBytes, ySlices * xSlices);
        clog << "Slice bytes numerator                = " << r.numerator << endl;
        clog << "Slice bytes denominator              = " << r.denominator << endl;
      }
    }
    const Array1D qMatrix = quantMatrix(kernel, waveletDepth);
    if (verbose) {
      clog << "Quantisation matrix = " << qMatrix[0];
      for (std::size_t i = 1; i < qMatrix.size(); ++i) clog << ", " << qMatrix[i];
      clog << endl;
    }

    vc2hip_picture_format pf = {width, picFormat.lumaHeight(), (int)chromaFormat, lumaDepth, bytes, chromaDepth};
    vc2hip_coding_params cp = {(int)kernel, waveletDepth, ySlices, xSlices,
                               mode == HQ_CBR ? VC2HIP_HQ_CBR : (mode == LD ? VC2HIP_LD : VC2HIP_HQ_CONSTQ),
                               qIndex, pictureBytes, slicePrefix, sliceScalar};
    const std::size_t frameBytes = (std::size_t)format.samples() * bytes;
    const std::size_t picBytes = (std::size_t)picFormat.samples() * bytes;
    const bool fragmented = (mode == HQ_CBR || mode == LD) && fragmentLength > 0; // EncodeStream.cpp:444-445

    std::vector<unsigned char> du; // output staging of the ordered writer
    unsigned long prev_parse_offset = 0;
    int major_version = 2;
    if (output == STREAM) {
      if (verbose) clog << endl << "Writing Sequence Header" << endl << endl;
      const SequenceHeader sh(mode == LD ? PROFILE_LD : PROFILE_HQ, height, width, chromaFormat, interlaced, (FrameRate)frameRate,
                              topFieldFirst, lumaDepth);
      const std::vector<unsigned char> body = writeSequenceHeader(sh, fragmented, &major_version);
      writeParseInfo(du, SEQUENCE_HEADER, body.size() + 13, prev_parse_offset);
      prev_parse_offset = body.size() + 13;
      du.insert(du.end(), body.begin(), body.end());
      out->write((const char *)du.data(), (std::streamsize)du.size());
    }
    const Array2D ldSliceBytes = mode == LD ? slice_bytes(ySlices, xSlices, pictureBytes, 1) : Array2D();

    unsigned long long frame = 0;
    bool done = false;

    if (output == STREAM) {
      // The fused picture path (EncodeStream.cpp:482-647 on the device), pipelined (Pipeline.h): picture k is coded by
      // worker k mod N, which keeps two pictures in flight on its GPU.  I/O runs at the speed of the copies: when the
      // input is a regular file every worker preads its own frames straight into pinned memory, and every worker
      // writes its own data unit straight from pinned memory -- the output range is reserved in picture order (parse
      // offsets and picture numbers chain: DataUnit.cpp:112-123, Utils.cpp:52-63), the pwrite happens outside that
      // critical section.  Pipes and interlaced frames (a frame is two pictures, first field first: Frame.cpp:90-104)
      // are read by this thread; pipes are written in order.
      out->flush();
      struct stat st;
      const int ifd = inFileName == "-" ? 0 : ::open(inFileName.c_str(), O_RDONLY);
      const bool inRegular = ifd >= 0 && fstat(ifd, &st) == 0 && S_ISREG(st.st_mode) && inFileName != "-";
      const long long inSize = inRegular ? (long long)st.st_size : 0;
      int ofd = 1;
      // (the raw descriptor is opened BEFORE the stream that wrote the sequence header is closed: a named pipe never
      // sees a moment without a writer, which its reader would take for the end of the stream)
      if (outFileName != "-") { ofd = ::open(outFileName.c_str(), O_WRONLY | O_APPEND); outFile.close(); }
      if (ifd < 0 || ofd < 0) throw std::runtime_error("cannot reopen the input / output file for the pipelined path");
      const bool outSeekable = outFileName != "-" && fstat(ofd, &st) == 0 && S_ISREG(st.st_mode);
      if (outSeekable) { ::close(ofd); ofd = ::open(outFileName.c_str(), O_WRONLY); } // (pwrite ignores the offset of O_APPEND descriptors)
      auto writeAll = [&](const unsigned char *p, std::size_t n, long long at) {
        while (n) {
          const ssize_t r = outSeekable ? pwrite(ofd, p, n, (off_t)at) : ::write(ofd, p, n);
          if (r < 0 && errno == EINTR) continue;
          if (r <= 0) throw std::runtime_error(string("Failed to write output file \"") + outFileName + "\"");
          p += r; n -= (std::size_t)r; at += r;
        }
      };
      const std::chrono::steady_clock::time_point tCtx0 = std::chrono::steady_clock::now();
      GpuWorkers workers(devices, picBytes, vc2hip_max_payload_bytes(&pf, &cp) + 64);
      const double ctxSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - tCtx0).count();
      const utils::Rational ldRatio = utils::rationalise(pictureBytes, ySlices * xSlices);
      // ---- the ordered part of the writer
      std::mutex om;
      std::condition_variable ocv;
      unsigned long long nextOut = 0;
      long long outPos = (long long)du.size(); // behind the sequence header
      string firstError;
      // VC2_TOOL_STATS=1: the steady-state rate on stderr -- pictures per second between the completion of picture 4 * workers
      // (contexts created, workspaces allocated, pages touched) and the last one (tools/cli_throughput.sh)
      const bool stats = getenv("VC2_TOOL_STATS") != nullptr;
      const unsigned long long warmPics = 4ull * devices.size();
      std::chrono::steady_clock::time_point tWarm, tStart = std::chrono::steady_clock::now();
      workers.setSink([&](unsigned long long seq, const unsigned char *data, std::size_t len, const string &err) {
        const unsigned long picnum = utils::getPictureNumber((int)(seq % framePics), seq / framePics, framePics);
        std::vector<unsigned char> head; // parse info + picture header (or, fragmented, the whole picture)
        std::unique_lock<std::mutex> lock(om);
        ocv.wait(lock, [&]() { return nextOut == seq; });
        if (!err.empty() || !firstError.empty()) {
          if (firstError.empty()) firstError = err;
          ++nextOut; ocv.notify_all();
          return;
        }
        if (verbose) clog << "Forward transform" << endl << "Quantise transform coefficients" << endl << "Writing compressed output to file" << endl;
        bool payloadFollows = true;
        if (fragmented) {
          // DataUnit.cpp:156-232 / :267-342: parameters fragment + fragments of whole slices
          const std::vector<unsigned char> params =
              mode == LD ? writeTransformParams(kernel, waveletDepth, true, xSlices, ySlices, (unsigned)ldRatio.numerator, (unsigned)ldRatio.denominator)
                         : writeTransformParams(kernel, waveletDepth, true, xSlices, ySlices, (unsigned)slicePrefix, (unsigned)sliceScalar);
          std::vector<std::size_t> sizes;
          if (mode == LD) for (std::size_t i = 0; i < ldSliceBytes.num_elements(); ++i) sizes.push_back((std::size_t)ldSliceBytes.data()[i]);
          else sizes = sliceSizesHQ(data, len, ySlices * xSlices, slicePrefix, sliceScalar);
          writeFragmentedPicture(head, mode == LD, picnum, params, data, sizes, xSlices, fragmentLength, &prev_parse_offset);
          payloadFollows = false;
        } else {
          const std::vector<unsigned char> hdr =
              mode == LD ? writePictureHeaderLD(picnum, kernel, waveletDepth, xSlices, ySlices, ldRatio, major_version)
                         : writePictureHeaderHQ(picnum, kernel, waveletDepth, xSlices, ySlices, slicePrefix, sliceScalar, major_version);
          const unsigned long next = (unsigned long)(hdr.size() + len + 13);
          writeParseInfo(head, mode == LD ? LD_PICTURE : HQ_PICTURE, next, prev_parse_offset);
          prev_parse_offset = next;
          head.insert(head.end(), hdr.begin(), hdr.end());
        }
        const long long at = outPos;
        outPos += (long long)head.size() + (payloadFollows ? (long long)len : 0);
        if (seq == warmPics) tWarm = std::chrono::steady_clock::now();
        try {
          if (!outSeekable) { // a pipe: in order, inside the critical section
            writeAll(head.data(), head.size(), 0);
            if (payloadFollows) writeAll(data, len, 0);
          }
        } catch (const std::exception &ex) { firstError = ex.what(); }
        ++nextOut;
        lock.unlock();
        ocv.notify_all();
        if (outSeekable) { // the range is ours: write it while the next picture reserves its own
          try { writeAll(head.data(), head.size(), at); if (payloadFollows) writeAll(data, len, at + (long long)head.size()); }
          catch (const std::exception &ex) { std::lock_guard<std::mutex> l2(om); if (firstError.empty()) firstError = ex.what(); }
        }
      });
      std::vector<unsigned char> frameBuf; // interlaced / piped input: the frame is read here and goes to the workers from here
      unsigned long long seq = 0;
      PictureResult res;
      auto drain = [&](bool all) { // completed pictures (their bytes went through the sink): errors surface in order
        while (all ? workers.wait(res) : workers.poll(res)) if (!res.error.empty()) throw std::logic_error(res.error);
        // a failed write (a full disk) ends the run at once instead of coding the rest of the input for nothing
        std::lock_guard<std::mutex> l2(om);
        if (!firstError.empty()) throw std::logic_error(firstError);
      };
      const long long wholeFrames = inRegular ? inSize / (long long)frameBytes : -1;
      for (;; ++frame) {
        if (verbose) clog << "Reading input frame number " << frame;
        bool got;
        if (inRegular && !interlaced) { // the worker reads it
          got = (long long)frame < wholeFrames;
          if (got) { (void)workers.inputBuffer(seq); workers.submitEncodeFile(seq, ifd, (long long)frame * (long long)frameBytes, frameBytes, pf, cp, mode == LD); ++seq; }
        } else {
          unsigned char *dst;
          if (interlaced) { frameBuf.resize(frameBytes); dst = frameBuf.data(); }
          else dst = workers.inputBuffer(seq);
          std::size_t n = 0;
          while (n < frameBytes) {
            const ssize_t r = ::read(ifd, dst + n, frameBytes - n);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            n += (std::size_t)r;
          }
          got = n == frameBytes;
          if (got) {
            if (interlaced) {
              for (int pic = 0; pic < framePics; ++pic) {
                extractFieldRaw(frameBuf.data(), format, bytes, (pic == 0) == topFieldFirst, workers.inputBuffer(seq));
                workers.submitEncode(seq++, pf, cp, mode == LD);
              }
            } else workers.submitEncode(seq++, pf, cp, mode == LD);
          }
        }
        if (!got) {
          if (frame == 0) { cerr << "\rFailed to read input frame number 0" << endl; return EXIT_FAILURE; }
          if (verbose) clog << "\rEnd of input reached after " << frame << " frames" << endl;
          break; // (a buffer taken for the missing frame is simply never submitted)
        } else if (verbose) clog << endl;
        drain(false);
      }
      workers.close();
      drain(true);
      if (!firstError.empty()) throw std::logic_error(firstError);
      du.clear();
      writeParseInfo(du, END_OF_SEQUENCE, 0, prev_parse_offset);
      writeAll(du.data(), du.size(), outPos);
      if (stats) {
        const std::chrono::steady_clock::time_point tEnd = std::chrono::steady_clock::now();
        const double all = std::chrono::duration<double>(tEnd - tStart).count();
        cerr << "EncodeStream stats: " << seq << " pictures in " << all << " s (" << ctxSeconds << " s of it creating the workers' contexts)";
        if (seq > warmPics + 1) cerr << "; steady state " << (double)(seq - warmPics) / std::chrono::duration<double>(tEnd - tWarm).count() << " pictures/s";
        cerr << endl;
      }
      if (outFileName != "-") ::close(ofd);
      if (inFileName != "-") ::close(ifd);
      return EXIT_SUCCESS;
    }

    std::vector<std::vector<unsigned char> > raws(1);
    while (!done) {
      // diagnostic outputs work on one frame at a time
      int got = 0;
      for (; got < 1; ++got) {
        raws[got].resize(frameBytes);
        if (verbose) clog << "Reading input frame number " << frame + got;
        in->read((char *)raws[got].data(), (std::streamsize)frameBytes);
        if ((std::size_t)in->gcount() < frameBytes) {
          if (frame + got == 0) { cerr << "\rFailed to read input frame number 0" << endl; return EXIT_FAILURE; }
          if (verbose) clog << "\rEnd of input reached after " << frame + got << " frames" << endl;
          done = true;
          break;
        } else if (verbose) clog << endl;
      }
      if (got == 0) break;

      // diagnostic outputs: the fine-grained Library functions, one frame at a time
      for (int g = 0; g < got; ++g, ++frame) {
        Picture inFrame(format);
        { Array2D y(format.lumaShape()), u(format.chromaShape()), v(format.chromaShape());
          const unsigned char *p = raws[g].data();
          unpackSamples(p, bytes, lumaDepth, true, true, y); p += y.num_elements() * bytes;
          unpackSamples(p, bytes, chromaDepth, true, true, u); p += u.num_elements() * bytes;
          unpackSamples(p, bytes, chromaDepth, true, true, v);
          inFrame.y(y); inFrame.c1(u); inFrame.c2(v); }
        Picture outFrame(format);
        int stats[128] = {0};
        bool finished = true; // false when the output mode stops before the decoded picture
        for (int pic = 0; pic < framePics; ++pic) {
          const bool top = (pic == 0) == topFieldFirst;
          const Picture picture = interlaced ? fieldOf(inFrame, top) : inFrame;
          if (verbose) clog << "Forward transform" << endl;
          const Picture transform = waveletTransform(picture, kernel, waveletDepth);
          if (output == TRANSFORM) { clog << "Writing transform coefficients to output file" << endl; writePicture4(*out, transform); finished = false; continue; }
          Array2D qIndices(ySlices, xSlices), sliceBytes;
          if (mode == HQ_CBR) {
            if (verbose) clog << "Determine quantisation indices" << endl;
            sliceBytes = slice_bytes(ySlices, xSlices, pictureBytes, sliceScalar);
            qIndices = quantIndicesCBR(transform, qMatrix, sliceBytes, sliceScalar);
          } else if (mode == LD) {
            if (verbose) clog << "Determine quantisation indices" << endl;
            sliceBytes = ldSliceBytes;
            qIndices = quantIndicesLD(transform, qMatrix, sliceBytes);
          } else for (std::size_t i = 0; i < qIndices.num_elements(); ++i) qIndices.data()[i] = qIndex;
          for (std::size_t i = 0; i < qIndices.num_elements(); ++i) ++stats[qIndices.data()[i] & 127];
          if (output == INDICES) {
            clog << "Writing quantisation indices to output file" << endl;
            for (std::size_t i = 0; i < qIndices.num_elements(); ++i) out->put((char)qIndices.data()[i]);
            finished = false;
            continue;
          }
          if (verbose) clog << "Quantise transform coefficients" << endl;
          // LL (DC) subband prediction in LD mode only (EncodeStream.cpp:541-548)
          const Picture quantised = mode == LD ? quantise_transform(transform, qIndices, qMatrix) : quantise_transform_np(transform, qIndices, qMatrix);
          if (output == QUANTISED) { clog << "Writing quantised transform coefficients to output file" << endl; writePicture4(*out, quantised); finished = false; continue; }
          if (output == PACKAGED) {
            if (verbose) clog << "Split quantised coefficients into slices" << endl << "Writing compressed output to file" << endl;
            const std::vector<unsigned char> b = mode == LD ? packSlicesLD(quantised, waveletDepth, qIndices, sliceBytes)
                                                            : packSlicesHQ(quantised, waveletDepth, qIndices, slicePrefix, sliceScalar, mode == HQ_CBR ? &sliceBytes : nullptr);
            out->write((const char *)b.data(), (std::streamsize)b.size());
            finished = false;
            continue;
          }
          // the reference dequantises without DC prediction here even in LD mode (EncodeStream.cpp:651)
          if (verbose) clog << "Inverse quantise" << endl;
          const Picture restored = inverse_quantise_transform_np(quantised, qIndices, qMatrix);
          if (verbose) clog << "Inverse transform" << endl;
          Picture decoded = inverseWaveletTransform(restored, kernel, waveletDepth, picture.format());
          if (verbose) clog << "Clip decoded picture" << endl;
          decoded = clip(decoded, -utils::pow(2, lumaDepth - 1), utils::pow(2, lumaDepth - 1) - 1, -utils::pow(2, chromaDepth - 1), utils::pow(2, chromaDepth - 1) - 1);
          if (interlaced) setField(outFrame, decoded, top);
          else outFrame = decoded;
        }
        if (!finished) continue;
        // quantiser statistics + PSNR over the frame, EncodeStream.cpp:676-753
        float mean = 0, meanSquare = 0;
        const int totalSlices = framePics * ySlices * xSlices;
        for (int z = 0; z < 128; ++z) { mean += z * stats[z]; meanSquare += z * z * stats[z]; }
        mean /= totalSlices; meanSquare /= totalSlices;
        const float stdDev = std::sqrt(meanSquare - mean * mean);
        auto psnr = [](const Array2D &a, const Array2D &b, int depth) {
          long long ss = 0;
          for (std::size_t i = 0; i < a.num_elements(); ++i) { const int d = a.data()[i] - b.data()[i]; ss += (long long)(d * d); }
          const float rms = std::sqrt(float(ss) / float(a.num_elements())) / utils::pow(2, depth);
          return -20 * std::log10(rms);
        };
        const float yp = psnr(inFrame.y(), outFrame.y(), lumaDepth), up = psnr(inFrame.c1(), outFrame.c1(), chromaDepth), vp = psnr(inFrame.c2(), outFrame.c2(), chromaDepth);
        if (verbose) {
          clog << endl << std::fixed << std::setprecision(2) << "Mean, Standard Deviation of quantiser index = " << mean << ", " << stdDev << endl;
          clog << std::fixed << std::setprecision(4) << "PSNR for Y/R, U/G, V/B = " << yp << ", " << up << ", " << vp << endl;
        }
        if (output == DECODED) { // wordWidth(bytes) + offset_binary, right justified (EncodeStream.cpp:756-760)
          if (verbose) clog << "Writing decoded output frame " << frame << endl;
          std::vector<unsigned char> b(frameBytes);
          unsigned char *p = b.data();
          packSamples(outFrame.y(), bytes, lumaDepth, false, true, p); p += outFrame.y().num_elements() * bytes;
          packSamples(outFrame.c1(), bytes, chromaDepth, false, true, p); p += outFrame.c1().num_elements() * bytes;
          packSamples(outFrame.c2(), bytes, chromaDepth, false, true, p);
          out->write((const char *)b.data(), (std::streamsize)b.size());
        } else { // PSNR
          *out << "Frame " << frame << endl << std::fixed << std::setprecision(2) << mean << " " << stdDev << endl;
          *out << std::fixed << std::setprecision(4) << yp << " " << up << " " << vp << endl;
        }
      }
    }
    if (output == STREAM) { du.clear(); writeParseInfo(du, END_OF_SEQUENCE, 0, prev_parse_offset); out->write((const char *)du.data(), (std::streamsize)du.size()); }
    out->flush();
  } catch (const std::exception &ex) {
    cout << "Error: " << ex.what() << endl; // the reference reports on standard output (EncodeStream.cpp:782-785)
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
