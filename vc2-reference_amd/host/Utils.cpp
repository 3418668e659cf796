#include "Utils.h"

#include <cstdlib>
#include <stdexcept>

namespace utils {
int pow(int base, int exp) {
  int value = 1;
  for (; exp > 0; --exp) value *= base;
  return value;
}
int intlog2(int value) {
  int log = 0;
  for (--value; value > 0; value >>= 1) ++log;
  return log;
}
unsigned long getPictureNumber(int fieldNumber, unsigned long long frameNumber, int fieldsPerFrame) {
  if (fieldNumber < 0) throw std::logic_error("field number should be positive");
  if (fieldNumber > fieldsPerFrame) throw std::logic_error("field number exceeds number of fields per frame");
  if (fieldsPerFrame != 1 && fieldsPerFrame != 2)
    throw std::logic_error("number of fields per frame should be 1 (progressive) or 2 (interlaced)");
  const unsigned long long big = (unsigned long long)fieldNumber + frameNumber * (unsigned long long)fieldsPerFrame;
  return (unsigned long)(big % (1ULL << 32));
}
static int gcd(int a, int b) {
  a = std::abs(a);
  b = std::abs(b);
  while (b) { const int t = a % b; a = b; b = t; }
  return a;
}
Rational rationalise(int numerator, int denominator) {
  const int g = gcd(numerator, denominator);
  Rational r;
  r.numerator = g ? numerator / g : numerator;
  r.denominator = g ? denominator / g : denominator;
  return r;
}
}  // namespace utils
