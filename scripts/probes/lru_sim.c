// LRU miss count of a stream of line ids through a fully associative cache of `cap` lines (scripts/l2_model.py): what a 4 MB L2 that
// sees an XCD's gathers in slice order could at best hold on to.  gcc -O2 -shared -fPIC -o /tmp/lru_sim.so scripts/probes/lru_sim.c
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
int64_t lru_misses(const int32_t* stream, int64_t m, int64_t nlines, int64_t cap) {
  // doubly linked list over line ids + a "resident" flag: O(1) per access
  int32_t* prev = (int32_t*)malloc((size_t)nlines * 4);
  int32_t* next = (int32_t*)malloc((size_t)nlines * 4);
  unsigned char* in = (unsigned char*)calloc((size_t)nlines, 1);
  int32_t head = -1, tail = -1;
  int64_t count = 0, misses = 0;
  for (int64_t i = 0; i < m; ++i) {
    const int32_t x = stream[i];
    if (in[x]) {
      if (head == x) continue;
      // unlink
      if (prev[x] >= 0) next[prev[x]] = next[x];
      if (next[x] >= 0) prev[next[x]] = prev[x];
      if (tail == x) tail = prev[x];
    } else {
      ++misses;
      in[x] = 1;
      if (count == cap) {          // evict the least recently used line
        const int32_t t = tail;
        tail = prev[t];
        if (tail >= 0) next[tail] = -1;
        in[t] = 0;
      } else {
        ++count;
      }
    }
    prev[x] = -1;
    next[x] = head;
    if (head >= 0) prev[head] = x;
    head = x;
    if (tail < 0) tail = x;
  }
  free(prev); free(next); free(in);
  return misses;
}
