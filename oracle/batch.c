/*
 * oracle/batch.c -- run one CPU codec over a batch of chunks with a static
 * partition over `threads` pthreads and report the best wall time. This is the
 * "CPU path timed beside" the GPU path: same chunk
 * arrays, same order. TEST INFRASTRUCTURE ONLY.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdlib.h>
#include <time.h>

#include "batch.h"
#include "oracle.h"

typedef struct
{
  batch_codec_fn fn;
  size_t begin, end;
  const uint8_t* const* in_ptrs;
  const size_t* in_sizes;
  uint8_t* const* out_ptrs;
  const size_t* out_caps;
  size_t* out_sizes;
  int errors;
} job_t;

static void* worker(void* arg)
{
  job_t* j = (job_t*)arg;
  j->errors = 0;
  for (size_t i = j->begin; i < j->end; ++i) {
    size_t out = 0;
    const int rc = j->fn(j->in_ptrs[i], j->in_sizes[i], j->out_ptrs[i], j->out_caps[i], &out);
    j->out_sizes[i] = out;
    if (rc != 0) {
      j->errors++;
    }
  }
  return NULL;
}

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double batch_run_generic(
    batch_codec_fn fn, int threads, int repeats, size_t n_chunks,
    const uint8_t* const* in_ptrs, const size_t* in_sizes,
    uint8_t* const* out_ptrs, const size_t* out_caps, size_t* out_sizes, int* errors)
{
  if (threads < 1) {
    threads = 1;
  }
  if (repeats < 1) {
    repeats = 1;
  }
  job_t* jobs = (job_t*)calloc((size_t)threads, sizeof(job_t));
  pthread_t* tids = (pthread_t*)calloc((size_t)threads, sizeof(pthread_t));
  double best = 1e300;
  int errs = 0;
  for (int r = 0; r < repeats; ++r) {
    const double t0 = now_s();
    for (int t = 0; t < threads; ++t) {
      jobs[t].fn = fn;
      jobs[t].begin = n_chunks * (size_t)t / (size_t)threads;
      jobs[t].end = n_chunks * (size_t)(t + 1) / (size_t)threads;
      jobs[t].in_ptrs = in_ptrs;
      jobs[t].in_sizes = in_sizes;
      jobs[t].out_ptrs = out_ptrs;
      jobs[t].out_caps = out_caps;
      jobs[t].out_sizes = out_sizes;
      if (threads == 1) {
        worker(&jobs[t]);
      } else {
        pthread_create(&tids[t], NULL, worker, &jobs[t]);
      }
    }
    errs = 0;
    for (int t = 0; t < threads; ++t) {
      if (threads > 1) {
        pthread_join(tids[t], NULL);
      }
      errs += jobs[t].errors;
    }
    const double dt = now_s() - t0;
    if (dt < best) {
      best = dt;
    }
  }
  free(jobs);
  free(tids);
  if (errors) {
    *errors = errs;
  }
  return best;
}

#ifndef BATCH_NO_PORT
/* ---- codec table of the port (not part of the liblz4/snappy shim build) -- */

static int c_lz4_dec(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  return oracle_lz4_decompress(s, n, d, cap, out);
}
static int c_snappy_dec(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  return oracle_snappy_decompress(s, n, d, cap, out);
}
static int c_lz4_enc(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  *out = oracle_lz4_compress(s, n, d, cap);
  return (*out == 0 && n != 0) ? 1 : 0;
}
static int c_snappy_enc(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  *out = oracle_snappy_compress(s, n, d, cap);
  return *out == 0 ? 1 : 0;
}

static int c_cascaded_dec(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  return oracle_cascaded_decompress(s, n, d, cap, out);
}
static int c_bitcomp_dec(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  return oracle_bitcomp_decompress(s, n, d, cap, out);
}
static int c_ans_dec(const uint8_t* s, size_t n, uint8_t* d, size_t cap, size_t* out)
{
  return oracle_ans_decompress(s, n, d, cap, out);
}

double oracle_batch_run(
    int codec, int threads, int repeats, size_t n_chunks,
    const uint8_t* const* in_ptrs, const size_t* in_sizes,
    uint8_t* const* out_ptrs, const size_t* out_caps, size_t* out_sizes, int* errors)
{
  static const batch_codec_fn table[7] = {c_lz4_dec, c_snappy_dec, c_lz4_enc, c_snappy_enc,
                                          c_cascaded_dec, c_bitcomp_dec, c_ans_dec};
  if (codec < 0 || codec > 6) {
    return -1.0;
  }
  return batch_run_generic(
      table[codec], threads, repeats, n_chunks, in_ptrs, in_sizes, out_ptrs, out_caps, out_sizes, errors);
}
#endif /* BATCH_NO_PORT */
