// frametest -- prints what class Frame (host/Frame.h) does to a few frames, one line per fact.  tests/test_host.py builds
// it twice: with host/Frame.cpp, and -- when /root/reference exists -- with the reference's own src/Library/src/Frame.cpp
// compiled from where it lies against host/*.h (no reference text is kept in this repository); the two outputs must be
// equal.  That is both the proof that host/Arrays.h carries the container idioms the reference's sources are written in
// and the parity check of our field accessors.
#include <cstdio>

#include "Frame.h"

static unsigned long long digest(const Array2D &a) {
  unsigned long long h = 1469598103934665603ull;
  h = (h ^ (unsigned long long)a.shape()[0]) * 1099511628211ull;
  h = (h ^ (unsigned long long)a.shape()[1]) * 1099511628211ull;
  for (std::size_t i = 0; i < a.num_elements(); ++i) h = (h ^ (unsigned long long)(unsigned)a.data()[i]) * 1099511628211ull;
  return h;
}
static void show(const char *what, const Picture &p) {
  std::printf("%s %dx%d/%dx%d %016llx %016llx %016llx\n", what, p.format().lumaHeight(), p.format().lumaWidth(),
              p.format().chromaHeight(), p.format().chromaWidth(), digest(p.y()), digest(p.c1()), digest(p.c2()));
}
static Array2D ramp(int h, int w, int seed) {
  Array2D a(extents[h][w]);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) a[y][x] = seed * 7919 + y * 131 - x * 17 + ((x * y) & 5);
  return a;
}

int main() {
  const struct { int h, w; ColourFormat cf; } cases[] = {{8, 12, CF444}, {8, 12, CF422}, {8, 12, CF420}, {4, 2, CF420}, {16, 32, CF422}};
  for (const auto &c : cases) {
    for (int tffFirst = 0; tffFirst < 2; ++tffFirst) {
      Frame f(c.h, c.w, c.cf, true, tffFirst != 0);
      const PictureFormat fmt = f.format();
      f.y(ramp(fmt.lumaHeight(), fmt.lumaWidth(), 1));
      f.c1(ramp(fmt.chromaHeight(), fmt.chromaWidth(), 2));
      f.c2(ramp(fmt.chromaHeight(), fmt.chromaWidth(), 3));
      std::printf("case %dx%d cf%d tff%d interlaced%d\n", c.h, c.w, (int)c.cf, (int)f.topFieldFirst(), (int)f.interlaced());
      show("frame ", f.frame());
      show("top   ", f.topField());
      show("bottom", f.bottomField());
      show("first ", f.firstField());
      show("second", f.secondField());
      // a second frame assembled from the fields in the other order of calls
      Frame g(fmt, false, true);
      g.interlaced(true); g.topFieldFirst(tffFirst == 0);
      g.secondField(f.firstField()); g.firstField(f.secondField());
      show("swap  ", g.frame());
      g.bottomField(f.bottomField()); g.topField(f.topField());
      show("back  ", g.frame());
      Frame k(fmt);
      k.frame(f);
      show("copy  ", k);
    }
  }
  return 0;
}
