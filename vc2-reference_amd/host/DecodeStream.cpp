// DecodeStream -- command-line compatible with /root/reference/src/DecodeStream (DecodeParams.cpp:39-92,
// DecodeStream.cpp:103-992: synchronise, data-unit dispatch, HQ and LD pictures, four output modes), with
// the per-picture body on MI355X through libvc2hip.  Extension: --gpus N decodes picture k on GPU k mod N.
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <algorithm>
#include <atomic>
#include <functional>
#include <fstream>
#include <iostream>
#include <iterator>
#include <map>
#include <mutex>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
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

static const std::vector<ArgSpec> SPECS = {{'v', "verbose", false, ""}, {'o', "output", true, ""}, {'G', "gpus", true, ""}, {'D', "devices", true, ""}, {'h', "help", false, ""}};
static const char *USAGE = "DecodeStream (MI355X / libvc2hip)\nUsage: DecodeStream [-v] [-o Transform|Quantised|Indices|Decoded] [--gpus N | --devices a,b,..] inFile outFile\n";


struct Reassembly { // slices of a fragmented picture collected so far (DecodeStream.cpp:62-101)
  bool ld;
  PicturePreamble pre;
  int compressedBytes;
  Array2D sliceBytes; // LD
  std::vector<std::vector<unsigned char> > slices;
  int decoded;
};

int main(int argc, char *argv[]) {
  try {
    if (argc < 2) { clog << USAGE; return EXIT_SUCCESS; }
    string inFileName, outFileName; bool verbose; Output output; int gpus;
    std::vector<int> devices;
    try {
      Args a(SPECS, argc, argv);
      if (a.isSet("help")) { cout << USAGE; return EXIT_SUCCESS; }
      if (a.positional.size() != 2) throw std::invalid_argument("Required arguments missing: inFile, outFile");
      inFileName = a.positional[0]; outFileName = a.positional[1];
      verbose = a.isSet("verbose");
      output = a.isSet("output") ? parseOutput(a.get("output")) : DECODED;
      gpus = a.getInt("gpus", 1);
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
    } catch (const std::exception &e) { cerr << "Command line error: " << e.what() << endl; return EXIT_FAILURE; }

    std::ifstream inFile; std::ofstream outFile;
    std::ostream *out = &cout;
    // (the input is opened once, below, as a descriptor: opening and closing it here first would drop the only reader of a
    // named pipe whose writer is already producing)
    // (decoded pictures go out through a raw descriptor, below; the stream serves the diagnostic outputs)
    if (outFileName != "-" && output != DECODED) { outFile.open(outFileName.c_str(), std::ios::binary); if (!outFile) { perror((string("Failed to open output file \"") + outFileName + "\"").c_str()); return EXIT_FAILURE; } out = &outFile; }
    // the whole stream: a regular file is mapped (no copy at all until a picture's bytes go into a pinned buffer), a pipe
    // is read in 8 MiB pieces (the round-2 tool read either through istreambuf_iterator, a byte at a time)
    struct Bytes {
      const unsigned char *p = nullptr; std::size_t n = 0;
      std::size_t size() const { return n; }
      const unsigned char &operator[](std::size_t i) const { return p[i]; }
    } s;
    std::vector<unsigned char> piped;
    {
      const int ifd = inFileName == "-" ? 0 : ::open(inFileName.c_str(), O_RDONLY);
      struct stat st;
      if (ifd < 0) { perror((string("Failed to open input file \"") + inFileName + "\"").c_str()); return EXIT_FAILURE; }
      void *map = MAP_FAILED;
      if (inFileName != "-" && fstat(ifd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0)
        map = mmap(nullptr, (std::size_t)st.st_size, PROT_READ, MAP_PRIVATE, ifd, 0);
      if (map != MAP_FAILED) { s.p = (const unsigned char *)map; s.n = (std::size_t)st.st_size; }
      else {
        std::vector<unsigned char> chunk(8u << 20);
        for (;;) {
          const ssize_t r = ::read(ifd, chunk.data(), chunk.size());
          if (r < 0 && errno == EINTR) continue;
          if (r <= 0) break;
          piped.insert(piped.end(), chunk.begin(), chunk.begin() + r);
        }
        s.p = piped.data(); s.n = piped.size();
      }
      if (inFileName != "-") ::close(ifd);
    }
    // Decoded output: every picture has its place in the file (frames are of one size per sequence), so the worker that
    // decoded it writes it there, straight from its pinned buffer (pwrite; a pipe is written in order by this thread)
    int ofd = -1;
    bool outSeekable = false;
    long long outBase = 0; // bytes of the sequences before the current one
    if (output == DECODED) {
      out->flush();
      struct stat st;
      // A regular file is NOT truncated up front: every picture has its place, and the size is set once at the end (also on
      // an error: exactly the frames written before it remain, as with the reference's tool).  Decoding over an existing
      // file of the same size then reuses its pages -- on the GPU boxes a NEW file in /dev/shm takes 4 - 6 GB/s however
      // many threads fill it (tools/probe/pagetouch.c: the kernel's page allocation for the file), a populated one 10+.
      ofd = outFileName == "-" ? 1 : ::open(outFileName.c_str(), O_WRONLY | O_CREAT, 0666);
      if (ofd < 0) { perror((string("Failed to open output file \"") + outFileName + "\"").c_str()); return EXIT_FAILURE; }
      outSeekable = outFileName != "-" && fstat(ofd, &st) == 0 && S_ISREG(st.st_mode);
      // ... but whatever ends this process early (a signal, std::terminate from a worker, a watchdog) would then leave a longer
      // old file's frames behind the new ones, and they look valid.  So an existing file IS emptied up front
      // unless the caller asks for the reuse (VC2_DECODESTREAM_REUSE=1: tools/cli_throughput.sh's "over an existing file" rows).
      if (outSeekable && st.st_size > 0 && getenv("VC2_DECODESTREAM_REUSE") == nullptr && ftruncate(ofd, 0) != 0) {
        perror((string("Failed to truncate output file \"") + outFileName + "\"").c_str());
        return EXIT_FAILURE;
      }
    }
    auto writeAt = [&](const unsigned char *p, std::size_t n, long long at) {
      while (n) {
        const ssize_t r = outSeekable ? pwrite(ofd, p, n, (off_t)at) : ::write(ofd, p, n);
        if (r < 0 && errno == EINTR) continue;
        if (r <= 0) throw std::runtime_error(string("Failed to write output file \"") + outFileName + "\"");
        p += r; n -= (std::size_t)r; at += r;
      }
    };

    // dataunitio::synchronise, DataUnit.cpp:1086-1109
    std::size_t pos = 0;
    while (pos + 4 <= s.size() && !(s[pos] == 0x42 && s[pos + 1] == 0x42 && s[pos + 2] == 0x43 && s[pos + 3] == 0x44)) ++pos;

    bool have_seq_hdr = false;
    int height = 0, width = 0, bytes = 0, depthBits = 0, major_version = 2;
    bool interlaced = false, topFieldFirst = true;
    ColourFormat chromaFormat = CF_UNSET;
    int frame = 0, pic = 0;
    std::vector<unsigned char> outFrame; // interlaced: the frame being filled field by field
    std::map<unsigned long, Reassembly> reassembling;

    // Decoded output: pictures go to the per-GPU workers (picture k to worker k mod N, two in flight per worker:
    // Pipeline.h) and come back in order.  The largest data unit of the input bounds a picture's payload.
    std::size_t maxUnit = 0;
    for (std::size_t p0 = pos; p0 + 13 <= s.size();) {
      const std::size_t nxt = ((std::size_t)s[p0 + 5] << 24) | ((std::size_t)s[p0 + 6] << 16) | ((std::size_t)s[p0 + 7] << 8) | s[p0 + 8];
      if (nxt == 0) { maxUnit = std::max(maxUnit, s.size() - p0); break; }
      maxUnit = std::max(maxUnit, nxt);
      p0 += nxt;
    }
    std::unique_ptr<GpuWorkers> workers;
    unsigned long long seq = 0;
    std::size_t seqPictureBytes = 0; // raw bytes of one picture of the current worker pool
    double ctxSeconds = 0;
    std::mutex writeMutex; // pwrites of several threads to one file queue behind its inode lock and take longer than one after the other (pagetouch.c)
    long long doneBytes = 0; // end of the last frame written in order
    // the size of a regular output file: the frames written in order (on every way out of this block)
    bool fileFinished = false; // (once: the guard below runs again after the explicit call and the ::close behind it)
    auto finishFile = [&]() {
      workers.reset(); // (joins the workers: none of them is still writing)
      if (fileFinished) return;
      fileFinished = true;
      if (outSeekable && ofd > 1 && ftruncate(ofd, (off_t)doneBytes) != 0)
        perror((string("Failed to cut output file \"") + outFileName + "\" to the frames written").c_str());
    };
    struct FileGuard { std::function<void()> f; ~FileGuard() { f(); } } fileGuard{finishFile};
    const bool stats = getenv("VC2_TOOL_STATS") != nullptr; // steady-state rate on stderr, as in EncodeStream
    const int warmPics = 4 * (int)devices.size();
    std::chrono::steady_clock::time_point tWarm, tStart = std::chrono::steady_clock::now();
    long long outPos = 0;          // pipe / interlaced: where the next frame goes
    bool sinkWrites = false;       // the workers write progressive frames of a regular file themselves
    auto writeDecoded = [&](const PictureResult &r) {
      if (!r.error.empty()) throw std::logic_error(r.error);
      if (verbose) clog << "Copy picture to output frame" << endl;
      if (interlaced) { // DecodeStream.cpp:417-428: first field, then second field, then the frame is written
        const PictureFormat frameFormat(height, width, chromaFormat);
        if (pic == 0) outFrame.assign((std::size_t)frameFormat.samples() * bytes, 0);
        insertFieldRaw(outFrame.data(), frameFormat, bytes, (pic == 0) == topFieldFirst, r.bytes.data());
        if (pic == 0) { pic = 1; return; }
        pic = 0;
        if (verbose) clog << "Clipping output" << endl << "Writing decoded output file" << endl;
        writeAt(outFrame.data(), outFrame.size(), outPos);
        outPos += (long long)outFrame.size();
        doneBytes += (long long)outFrame.size();
      } else {
        if (verbose) clog << "Clipping output" << endl << "Writing decoded output file" << endl;
        if (!sinkWrites) { writeAt(r.bytes.data(), r.bytes.size(), outPos); outPos += (long long)r.bytes.size(); }
        doneBytes += (long long)seqPictureBytes;
      }
      ++frame;
      if (frame == warmPics) tWarm = std::chrono::steady_clock::now();
    };
    auto flush = [&]() { // everything submitted so far is decoded and written (a new sequence header may change the format)
      if (!workers) return;
      PictureResult r;
      workers->close();
      while (workers->wait(r)) writeDecoded(r);
      workers.reset();
      if (sinkWrites) outPos = outBase = outBase + (long long)seq * (long long)seqPictureBytes;
      seq = 0;
    };

    // one complete picture's slice bytes: queue it for a GPU, or run the diagnostic outputs
    auto handlePicture = [&](bool ld, const PicturePreamble &pre, int compressedBytes, const unsigned char *data, std::size_t dlen,
                             std::vector<unsigned char> *owned) {
      const int pictureHeight = interlaced ? height / 2 : height;
      if (output == DECODED) {
        (void)owned;
        vc2hip_picture_format pf = {width, pictureHeight, (int)chromaFormat, depthBits, bytes, 0};
        vc2hip_coding_params cp = {(int)pre.wavelet_kernel, pre.depth, pre.slices_y, pre.slices_x, ld ? VC2HIP_LD : VC2HIP_HQ_CONSTQ, 0,
                                   compressedBytes, pre.slice_prefix, pre.slice_size_scalar};
        if (!workers) {
          const std::chrono::steady_clock::time_point tCtx0 = std::chrono::steady_clock::now();
          workers.reset(new GpuWorkers(devices, std::max(maxUnit, dlen), vc2hip_raw_picture_bytes(&pf)));
          ctxSeconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - tCtx0).count();
          seqPictureBytes = vc2hip_raw_picture_bytes(&pf);
          sinkWrites = outSeekable && !interlaced;
          outBase = outPos;
          if (sinkWrites) {
            const long long base = outBase;
            const std::size_t pb = seqPictureBytes;
            // every worker writes its picture into its place, straight from its pinned buffer, one pwrite at a time (round 3
            // mapped the file and let the workers' copies fault its pages in: 104 - 129 frames/s; see the note at the open)
            workers->setSink([&writeAt, &writeMutex, base, pb](unsigned long long sq, const unsigned char *data, std::size_t len, const string &err) {
              if (!err.empty() || !len) return;
              std::lock_guard<std::mutex> lock(writeMutex);
              writeAt(data, len, base + (long long)sq * (long long)pb);
            });
          }
        }
        std::memcpy(workers->inputBuffer(seq, dlen), data, dlen); // (grows the slot's buffer when this picture is larger than the bound: fragmented VBR streams)
        workers->submitDecode(seq++, dlen, pf, cp, ld);
        PictureResult r;
        while (workers->poll(r)) writeDecoded(r);
        return;
      }
      // diagnostic outputs through the fine-grained functions
      const int ph = paddedSize(pictureHeight, pre.depth), pw = paddedSize(width, pre.depth);
      const PictureFormat tf(ph, pw, chromaFormat); // DecodeStream.cpp:483-498
      Picture q(tf);
      Array2D qIndices(pre.slices_y, pre.slices_x);
      if (ld) unpackSlicesLD(data, dlen, q, pre.depth, qIndices, slice_bytes(pre.slices_y, pre.slices_x, compressedBytes, 1), nullptr);
      else unpackSlicesHQ(data, dlen, q, pre.depth, qIndices, pre.slice_prefix, pre.slice_size_scalar, nullptr);
      if (verbose) clog << "Merge slices into full picture" << endl;
      if (output == INDICES) {
        clog << "Writing quantisation indices to output file" << endl;
        for (std::size_t i = 0; i < qIndices.num_elements(); ++i) out->put((char)qIndices.data()[i]);
      } else if (output == QUANTISED) {
        clog << "Writing quantised transform coefficients to output file" << endl;
        writePicture4(*out, q);
      } else {
        if (verbose) clog << "Inverse quantise" << endl;
        const Array1D qm = quantMatrix(pre.wavelet_kernel, pre.depth);
        clog << "Writing transform coefficients to output file" << endl;
        writePicture4(*out, ld ? inverse_quantise_transform(q, qIndices, qm) : inverse_quantise_transform_np(q, qIndices, qm));
      }
    };
    // LD byte budget of one picture as the reference derives it (DecodeStream.cpp:312, :331): the budget in the
    // transform parameters is already per picture, and the reference halves it once more for interlaced streams
    auto ldPictureBytes = [&](const PicturePreamble &pre) {
      const int compressed = (pre.slice_bytes.numerator * pre.slices_y * pre.slices_x) / pre.slice_bytes.denominator;
      return interlaced ? compressed / 2 : compressed;
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
          interlaced = h.interlace; topFieldFirst = h.topFieldFirst;
          if (interlaced && ((h.height & 1) || (PictureFormat(h.height, h.width, h.chromaFormat).chromaHeight() & 1)))
            throw std::logic_error("interlaced coding needs even luma and chroma heights");
          height = h.height; width = h.width; chromaFormat = h.chromaFormat; depthBits = h.bitdepth;
          bytes = h.bitdepth == 8 ? 1 : 2; // DecodeStream.cpp:268-271
          major_version = h.major_version;
          have_seq_hdr = true;
          break;
        }
        case END_OF_SEQUENCE:
          flush();
          if (verbose) clog << "End of Sequence after " << frame << " frames" << endl;
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
          if (verbose) {
            const Array1D qm = quantMatrix(pre.wavelet_kernel, pre.depth);
            clog << "Quantisation matrix = " << qm[0];
            for (std::size_t i = 1; i < qm.size(); ++i) clog << ", " << qm[i];
            clog << endl;
            if (interlaced) clog << "Reading compressed input field " << pic << " of frame " << frame << endl;
            else clog << "Reading compressed input frame number " << frame << endl;
          }
          // The reference parses the slices straight from the input stream (DecodeStream.cpp:513): in a corrupt stream
          // whose length bytes run the last slices past their data unit it reads on into the bytes that follow (and then
          // looks for the next parse info).  The decoder therefore sees up to one more slice's worth of the input behind
          // the unit; a valid picture ends where its unit ends and never looks at them.
          std::size_t visible = dlen;
          if (!ld) visible += (std::size_t)pre.slice_prefix + 4 + 3 * 255 * (std::size_t)pre.slice_size_scalar;
          else { // LD: a luma length field beyond its slice makes the reference's reader run on (Slices.cpp:246-303); a
                 // slice of n bytes can claim fewer than 2 * (8 n - 7) bits, so twice the budget bounds what it looks at
            const std::size_t reach = 2 * (std::size_t)ldPictureBytes(pre) + 2 * (std::size_t)pre.slices_y * pre.slices_x + 16;
            if (reach > visible) visible = reach;
          }
          handlePicture(ld, pre, ld ? ldPictureBytes(pre) : 0, body + hdr, visible < avail - hdr ? visible : avail - hdr, nullptr);
          if (output != DECODED) ++frame;
          used = unit;
          break;
        }
        case HQ_FRAGMENT:
        case LD_FRAGMENT: {
          // DecodeStream.cpp:614-797 (LD) / :799-977 (HQ): a parameters fragment opens the picture, slice
          // fragments fill it; the picture is decoded once all of its slices have arrived
          const bool ld = du.type == LD_FRAGMENT;
          if (verbose) clog << "Parsing " << (ld ? "LD" : "HQ") << " Fragment" << endl;
          Fragment frag;
          const std::size_t fh = readFragmentHeader(body, avail, &frag);
          const std::size_t unit = du.next_parse_offset ? (std::size_t)du.next_parse_offset - 13 : avail;
          used = unit;
          if (frag.n_slices == 0) {
            if (verbose) clog << "Parsing Picture Header" << endl;
            PicturePreamble pre;
            readTransformParams(body + fh, avail - fh, ld, major_version, &pre);
            if (verbose) {
              clog << "Picture number      : " << frag.picture_number << endl << "Wavelet Kernel      : " << pre.wavelet_kernel << endl;
              clog << "Transform Depth     : " << pre.depth << endl << "Slices Horizontally : " << pre.slices_x << endl;
              clog << "Slices Verically    : " << pre.slices_y << endl;
            }
            if (!have_seq_hdr) { clog << "Cannot decode frame, no previous sequence header!" << endl; break; }
            Reassembly r;
            r.ld = ld; r.pre = pre; r.decoded = 0;
            r.compressedBytes = ld ? ldPictureBytes(pre) : 0;
            if (ld) r.sliceBytes = slice_bytes(pre.slices_y, pre.slices_x, r.compressedBytes, 1);
            r.slices.assign((std::size_t)pre.slices_y * pre.slices_x, std::vector<unsigned char>());
            reassembling[frag.picture_number] = r;
            break;
          }
          std::map<unsigned long, Reassembly>::iterator it = reassembling.find(frag.picture_number);
          if (it == reassembling.end()) {
            clog << "Cannot decode slices as no picture header yet read for picture number " << frag.picture_number << endl;
            break;
          }
          Reassembly &r = it->second;
          if (verbose)
            clog << "Picture " << frag.picture_number << ": Reading " << frag.n_slices << " slices, starting from ("
                 << frag.slice_offset_x << ", " << frag.slice_offset_y << ")" << endl;
          const int ns = r.pre.slices_x * r.pre.slices_y;
          int si = frag.slice_offset_y * r.pre.slices_x + frag.slice_offset_x;
          const unsigned char *p = body + fh;
          std::size_t left = (unit > fh ? unit : fh) - fh;
          if (left > avail - fh) left = avail - fh;
          for (int k = 0; k < frag.n_slices && si < ns; ++k, ++si) { // Slices.cpp:662-694 with ExpectedSlicesForFragment
            std::size_t size;
            if (ld) size = (std::size_t)r.sliceBytes.data()[si];
            else size = sliceSizesHQ(p, left, 1, r.pre.slice_prefix, r.pre.slice_size_scalar)[0];
            if (size > left) throw std::logic_error("slice data runs past the end of the fragment");
            r.slices[(std::size_t)si].assign(p, p + size);
            p += size; left -= size;
          }
          r.decoded += frag.n_slices;
          if (r.decoded >= ns) {
            std::vector<unsigned char> payload;
            for (const std::vector<unsigned char> &sl : r.slices) payload.insert(payload.end(), sl.begin(), sl.end());
            const Reassembly done = r;
            reassembling.erase(it);
            const std::size_t plen = payload.size();
            handlePicture(done.ld, done.pre, done.compressedBytes, payload.data(), plen, &payload);
            if (output != DECODED) ++frame;
          }
          break;
        }
        default:
          break;
      }
      if (du.type == END_OF_SEQUENCE && du.next_parse_offset == 0) { pos += 13; continue; }
      pos += du.next_parse_offset ? (std::size_t)du.next_parse_offset : 13 + used;
    }
    out->flush();
    if (stats && output == DECODED) {
      const std::chrono::steady_clock::time_point tEnd = std::chrono::steady_clock::now();
      cerr << "DecodeStream stats: " << frame << " frames in " << std::chrono::duration<double>(tEnd - tStart).count() << " s (" << ctxSeconds
           << " s of it creating the workers' contexts)";
      if (frame > warmPics + 1) cerr << "; steady state " << (double)(frame - warmPics) / std::chrono::duration<double>(tEnd - tWarm).count() << " frames/s";
      cerr << endl;
    }
    finishFile();
    if (ofd > 1) ::close(ofd);
  } catch (const std::exception &ex) {
    cout << "Error: " << ex.what() << endl;
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
