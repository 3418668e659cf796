// VLC.h -- interleaved exp-Golomb and fixed-width fields on memory buffers, for the few dozen header
// bytes per data unit that stay on the host (/root/reference/src/Library/VLC.h, src/VLC.cpp:21-66,
// :151-257, :326-348).  Slice data is coded on the GPU, not here.
#ifndef VC2HOST_VLC_H
#define VC2HOST_VLC_H

#include <cstddef>
#include <vector>

class BitWriter {
 public:
  BitWriter() : cache(0), cached(0) {}
  void putBit(bool bit);
  void putBits(unsigned n, unsigned value);
  void putUnsignedVLC(unsigned value);       // VLC.cpp:21-52
  void putBoolean(bool b) { putBit(b); }
  void align();                               // VLC.cpp:246-250
  void putBytes(int n, unsigned long value);  // VLC.cpp:326-335 (aligns first)
  const std::vector<unsigned char> &bytes() const { return out; }

 private:
  std::vector<unsigned char> out;
  unsigned cache;
  int cached;
};

class BitReader {
 public:
  BitReader(const unsigned char *p, std::size_t n) : buf(p), len(n), pos(0), cache(0), cached(0), eof_(false) {}
  bool getBit();
  unsigned getBits(unsigned n);
  unsigned getUnsignedVLC();                  // VLC.cpp:283-295 + :54-66
  bool getBoolean() { return getBit(); }
  void align();
  unsigned long getBytes(int n);
  std::size_t bytePos() const { return pos; }
  bool eof() const { return eof_; }

 private:
  const unsigned char *buf;
  std::size_t len, pos;
  unsigned cache;
  int cached;
  bool eof_;
};
#endif
