// How fast can a process fill a NEW file in /dev/shm?  (DecodeStream's output path.)  gcc -O2 -pthread pagetouch.c
//   ./a.out <MiB> <threads>   ->  GB/s of: memcpy into a fresh MAP_SHARED mapping (a fault per page), the same after
//   MADV_POPULATE_WRITE, pwrite, and memcpy into a mapping of an already populated file.
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
static size_t total, piece = 32u << 20;
static int nthreads, mode, fd;
static unsigned char *map, *src;
static void *work(void *arg) {
  const long t = (long)arg;
  for (size_t off = t * piece; off < total; off += (size_t)nthreads * piece) {
    if (mode == 1) { if (madvise(map + off, piece, MADV_POPULATE_WRITE)) perror("madvise"); }
    if (mode == 2) { size_t n = 0; while (n < piece) { ssize_t r = pwrite(fd, src + n, piece - n, off + n); if (r <= 0) { perror("pwrite"); break; } n += r; } }
    else memcpy(map + off, src, piece);
  }
  return 0;
}
static double run(int m) {
  pthread_t th[64];
  mode = m;
  const double t0 = now();
  for (long t = 0; t < nthreads; ++t) pthread_create(&th[t], 0, work, (void *)t);
  for (long t = 0; t < nthreads; ++t) pthread_join(th[t], 0);
  return total / (now() - t0) / 1e9;
}
int main(int argc, char **argv) {
  total = (size_t)atol(argv[1]) << 20; nthreads = atoi(argv[2]);
  src = malloc(piece); memset(src, 7, piece);
  const char *names[4] = {"mapping, fault per page", "mapping, MADV_POPULATE_WRITE first", "pwrite", "mapping of a populated file"};
  for (int m = 0; m < 4; ++m) {
    if (m < 3) { unlink("/dev/shm/pagetouch.bin"); fd = open("/dev/shm/pagetouch.bin", O_RDWR | O_CREAT, 0600); if (ftruncate(fd, total)) return 1; }
    map = mmap(0, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    printf("%-36s %d threads: %.2f GB/s\n", names[m], nthreads, run(m));
    munmap(map, total);
    if (m < 2) close(fd);
  }
  unlink("/dev/shm/pagetouch.bin");
  return 0;
}
