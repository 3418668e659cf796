#include "VLC.h"

void BitWriter::putBit(bool bit) {
  cache = ((cache << 1) | (bit ? 1u : 0u)) & 0xFFu;
  if (++cached == 8) { out.push_back((unsigned char)cache); cached = 0; }
}
void BitWriter::putBits(unsigned n, unsigned value) { while (n > 0) { --n; putBit((value >> n) & 1u); } }
void BitWriter::putUnsignedVLC(unsigned value) {
  if (value == 0) { putBit(true); return; }
  value += 1;
  int top = 31;
  while (!((value >> top) & 1u)) --top;
  for (int b = top - 1; b >= 0; --b) { putBit(false); putBit((value >> b) & 1u); }
  putBit(true);
}
void BitWriter::align() { while (cached) putBit(false); }
void BitWriter::putBytes(int n, unsigned long value) {
  align();
  while (n > 0) { --n; out.push_back((unsigned char)(value >> (8 * n))); }
}

bool BitReader::getBit() {
  if (cached == 0) {
    if (pos < len) cache = buf[pos]; else { cache = 0xFF; eof_ = true; }
    ++pos;
    cached = 8;
  }
  --cached;
  return (cache >> cached) & 1u;
}
unsigned BitReader::getBits(unsigned n) { unsigned v = 0; while (n--) v = (v << 1) | (getBit() ? 1u : 0u); return v; }
unsigned BitReader::getUnsignedVLC() {
  unsigned value = 1;
  while (!getBit()) value = (value << 1) | (getBit() ? 1u : 0u);
  return value - 1;
}
void BitReader::align() { cached = 0; }
unsigned long BitReader::getBytes(int n) {
  align();
  unsigned long v = 0;
  while (n-- > 0) {
    unsigned b = 0xFF;
    if (pos < len) b = buf[pos]; else eof_ = true;
    ++pos;
    v = (v << 8) | b;
  }
  return v;
}
