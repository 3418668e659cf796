// Utils.h -- /root/reference/src/Library/Utils.h, src/Utils.cpp:33-72
#ifndef VC2HOST_UTILS_H
#define VC2HOST_UTILS_H

namespace utils {
struct Rational {
  int numerator, denominator;
};
int pow(int base, int exp);                                                       // Utils.cpp:33-37
int intlog2(int value);                                                           // Utils.cpp:40-48
unsigned long getPictureNumber(int fieldNumber, unsigned long long frameNumber, int fieldsPerFrame); // :52-63
Rational rationalise(int numerator, int denominator);                             // :65-72
}  // namespace utils
#endif
