/*
 * TEST / BENCH INFRASTRUCTURE ONLY -- multi-threaded CPU timing loops for the
 * block codec.  The block routine under test is passed in as a function
 * pointer, so the same driver times either the compiled reference
 * (oracle/_ref/libsimdcomp_ref.so, kind "reference") or the scalar restatement
 * (oracle/liboracle.so, kind "port").  Blocks are split into contiguous ranges,
 * one per thread: the reference's block routines are stateless and re-entrant
 * (no mutable globals on this path), which is what makes that legal.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

typedef void (*unpack_seeded_fn)(uint32_t, const void *, uint32_t *, uint32_t);
typedef void (*unpack_plain_fn)(const void *, uint32_t *, uint32_t);
typedef void (*pack_seeded_fn)(uint32_t, const uint32_t *, void *, uint32_t);
typedef void (*pack_plain_fn)(const uint32_t *, void *, uint32_t);

typedef struct {
  int kind; /* low 2 bits: 0 unpack seeded, 1 unpack plain, 2 pack seeded, 3 pack plain; kind >> 2: block widening */
  void *fn;
  const uint32_t *seeds;
  const uint8_t *payload_c;
  uint8_t *payload;
  const uint32_t *ints_c;
  uint32_t *ints;
  uint32_t bit;
  size_t first, last;
  size_t ring; /* decode/encode the int side into a ring of this many blocks (0 = full array) */
} job_t;

static void *run_job(void *arg) {
  job_t *j = (job_t *)arg;
  size_t k;
  /* kind >> 2 widens the block: 0 = 128 integers (4 lanes), 1 = 256 (the 8-lane AVX2 format), 2 = 512 (AVX-512) */
  const size_t ints = (size_t)128 << (j->kind >> 2);
  const size_t stride = (ints / 8) * j->bit;
  for (k = j->first; k < j->last; ++k) {
    size_t slot = j->ring ? (k % j->ring) : k;
    switch (j->kind & 3) {
    case 0:
      ((unpack_seeded_fn)j->fn)(j->seeds[k], j->payload_c + k * stride, j->ints + slot * ints, j->bit);
      break;
    case 1:
      ((unpack_plain_fn)j->fn)(j->payload_c + k * stride, j->ints + slot * ints, j->bit);
      break;
    case 2:
      ((pack_seeded_fn)j->fn)(j->seeds[k], j->ints_c + slot * ints, j->payload + k * stride, j->bit);
      break;
    default:
      ((pack_plain_fn)j->fn)(j->ints_c + slot * ints, j->payload + k * stride, j->bit);
      break;
    }
  }
  return NULL;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* Runs `reps` timed passes over blocks [0, nblocks) with `nthreads` threads and
 * returns the best wall time of one pass in seconds (one untimed warm-up pass
 * first).  kind as in job_t.  For unpack kinds `ints` is the output, for pack
 * kinds the input. */
double cpu_driver_time(int kind, void *fn, const uint32_t *seeds, uint8_t *payload, uint32_t *ints,
                       uint32_t bit, size_t nblocks, int nthreads, int reps) {
  pthread_t *th;
  job_t *jobs;
  double best = 1e30;
  int r, t;
  if (nthreads < 1) nthreads = 1;
  th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
  jobs = (job_t *)malloc(sizeof(job_t) * (size_t)nthreads);
  for (t = 0; t < nthreads; ++t) {
    jobs[t].kind = kind;
    jobs[t].fn = fn;
    jobs[t].seeds = seeds;
    jobs[t].payload_c = payload;
    jobs[t].payload = payload;
    jobs[t].ints_c = ints;
    jobs[t].ints = ints;
    jobs[t].bit = bit;
    jobs[t].first = nblocks * (size_t)t / (size_t)nthreads;
    jobs[t].last = nblocks * (size_t)(t + 1) / (size_t)nthreads;
    jobs[t].ring = 0;
  }
  for (r = -1; r < reps; ++r) {
    double t0 = now_s(), dt;
    if (nthreads == 1) {
      run_job(&jobs[0]);
    } else {
      for (t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, run_job, &jobs[t]);
      for (t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    dt = now_s() - t0;
    if (r >= 0 && dt < best) best = dt;
  }
  free(th);
  free(jobs);
  return best;
}

/* ---- the caller loops of example.c, with the block routines passed in --------
 * so that the very same loop runs over the compiled reference or over the
 * scalar restatement. */
typedef uint32_t (*maxbitsd1_fn)(uint32_t, const uint32_t *);
typedef uint32_t (*maxbits_fn)(const uint32_t *);

/* example.c:56-75: per block the width byte, then the payload; d1 offset = last
 * value of the previous block.  Returns the number of bytes written. */
size_t stream_compress_d1(void *maxbitsd1, void *packd1, const uint32_t *in, size_t nblocks, uint32_t initvalue,
                          uint8_t *buffer) {
  uint8_t *start = buffer;
  uint32_t offset = initvalue;
  size_t k;
  for (k = 0; k < nblocks; ++k) {
    uint32_t b = ((maxbitsd1_fn)maxbitsd1)(offset, in + k * 128);
    *buffer++ = (uint8_t)b;
    ((pack_seeded_fn)packd1)(offset, in + k * 128, buffer, b);
    offset = in[k * 128 + 127];
    buffer += b * 16;
  }
  return (size_t)(buffer - start);
}

/* example.c:135-150: the same stream without differential coding. */
size_t stream_compress_plain(void *maxbits, void *pack, const uint32_t *in, size_t nblocks, uint8_t *buffer) {
  uint8_t *start = buffer;
  size_t k;
  for (k = 0; k < nblocks; ++k) {
    uint32_t b = ((maxbits_fn)maxbits)(in + k * 128);
    *buffer++ = (uint8_t)b;
    ((pack_plain_fn)pack)(in + k * 128, buffer, b);
    buffer += b * 16;
  }
  return (size_t)(buffer - start);
}

/* example.c:101-110: decode with no stored offsets, offset = last decoded value. */
void stream_decode_d1(void *unpackd1, const uint8_t *buffer, size_t nblocks, uint32_t initvalue, uint32_t *out) {
  uint32_t offset = initvalue;
  size_t k;
  for (k = 0; k < nblocks; ++k) {
    uint8_t b = *buffer++;
    ((unpack_seeded_fn)unpackd1)(offset, buffer, out + k * 128, b);
    buffer += (size_t)b * 16;
    offset = out[k * 128 + 127];
  }
}

/* example.c:170-175 (the plain decode loop). */
void stream_decode_plain(void *unpack, const uint8_t *buffer, size_t nblocks, uint32_t *out) {
  size_t k;
  for (k = 0; k < nblocks; ++k) {
    uint8_t b = *buffer++;
    ((unpack_plain_fn)unpack)(buffer, out + k * 128, b);
    buffer += (size_t)b * 16;
  }
}

/* The record walk those decode loops perform, on its own: widths and arena
 * offsets (stream position minus record number) of the records in
 * buffer[0..nbytes).  Returns the number of records; *status = 1 when a width
 * byte exceeds 32 or the last record is cut short. */
size_t stream_walk(const uint8_t *buffer, size_t nbytes, size_t max_blocks, uint8_t *bits, uint64_t *off, int *status) {
  size_t p = 0, k = 0;
  *status = 0;
  while (p < nbytes) {
    uint8_t b = buffer[p];
    if (b > 32) {
      *status = 1;
      return k;
    }
    if (k < max_blocks) {
      bits[k] = b;
      off[k] = (uint64_t)(p - k);
    }
    p += 1 + 16 * (size_t)b;
    ++k;
  }
  if (p != nbytes) *status = 1;
  return k;
}

/* ---- batched search on the CPU: the reference's simdsearchwithlengthd1 (or the
 * restatement's) called once per query, for timing next to the device batch.
 * Returns the best wall time of `reps` passes; results go to pos[] / val[]. */
typedef int (*searchlen_fn)(uint32_t, const void *, uint32_t, int, uint32_t, uint32_t *);

double query_search_time(void *fn, const uint32_t *seeds, const uint8_t *payload, const uint64_t *off, const uint8_t *bits,
                         const uint64_t *query_blocks, const uint32_t *keys, size_t nq, int32_t *pos, uint32_t *val,
                         int reps) {
  double best = 1e30;
  int r;
  size_t q;
  for (r = -1; r < reps; ++r) {
    double t0 = now_s(), dt;
    for (q = 0; q < nq; ++q) {
      size_t k = (size_t)query_blocks[q];
      pos[q] = ((searchlen_fn)fn)(seeds[k], payload + off[k], bits[k], 128, keys[q], &val[q]);
    }
    dt = now_s() - t0;
    if (r >= 0 && dt < best) best = dt;
  }
  return best;
}
