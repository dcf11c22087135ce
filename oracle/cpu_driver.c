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
  int kind; /* 0 unpack seeded, 1 unpack plain, 2 pack seeded, 3 pack plain */
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
  const size_t stride = 16u * j->bit;
  for (k = j->first; k < j->last; ++k) {
    size_t slot = j->ring ? (k % j->ring) : k;
    switch (j->kind) {
    case 0:
      ((unpack_seeded_fn)j->fn)(j->seeds[k], j->payload_c + k * stride, j->ints + slot * 128, j->bit);
      break;
    case 1:
      ((unpack_plain_fn)j->fn)(j->payload_c + k * stride, j->ints + slot * 128, j->bit);
      break;
    case 2:
      ((pack_seeded_fn)j->fn)(j->seeds[k], j->ints_c + slot * 128, j->payload + k * stride, j->bit);
      break;
    default:
      ((pack_plain_fn)j->fn)(j->ints_c + slot * 128, j->payload + k * stride, j->bit);
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
