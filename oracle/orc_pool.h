/*
 * oracle/orc_pool.h -- TEST INFRASTRUCTURE.  Minimal persistent worker pool standing in for
 * util/IndexThreadReduce.h:32-194 (NUM_THREADS workers, mutex + condvar, index ranges handed out in
 * chunks of `step`; step == 0 => one static chunk per worker).  Only used to time the CPU baseline
 * with the reference's threading; the sequential path (nthreads <= 1) is the parity oracle.
 */
#ifndef ORC_POOL_H
#define ORC_POOL_H

#include <pthread.h>
#include <stdlib.h>

#define ORC_MAX_THREADS 16

typedef void (*orc_job_fn)(void *ctx, int first, int last, int tid);

typedef struct orc_pool {
  pthread_t th[ORC_MAX_THREADS];
  int nthreads;
  pthread_mutex_t mu;
  pthread_cond_t todo, done;
  orc_job_fn fn;
  void *ctx;
  int next, end, step;
  int generation;
  int running;
  int quit;
  int started;
} orc_pool;

static orc_pool g_orc_pool;

typedef struct orc_worker_arg {
  int tid;
} orc_worker_arg;
static orc_worker_arg g_orc_wargs[ORC_MAX_THREADS];

static void *orc_worker(void *p) {
  int tid = ((orc_worker_arg *)p)->tid;
  orc_pool *P = &g_orc_pool;
  int seen = 0;
  pthread_mutex_lock(&P->mu);
  for (;;) {
    while (!P->quit && (seen == P->generation || tid >= P->nthreads)) pthread_cond_wait(&P->todo, &P->mu);
    if (P->quit) break;
    /* grab chunks until the range is exhausted */
    for (;;) {
      if (P->next >= P->end) break;
      int first = P->next;
      int last = first + P->step;
      if (last > P->end) last = P->end;
      P->next = last;
      pthread_mutex_unlock(&P->mu);
      P->fn(P->ctx, first, last, tid);
      pthread_mutex_lock(&P->mu);
    }
    seen = P->generation;
    P->running--;
    if (P->running == 0) pthread_cond_signal(&P->done);
  }
  pthread_mutex_unlock(&P->mu);
  return 0;
}

/* Runs fn over [first,end) with `nthreads` workers (tid 0..nthreads-1). nthreads <= 1: inline. */
static void orc_parallel_for(int nthreads, orc_job_fn fn, void *ctx, int first, int end, int step) {
  if (nthreads > ORC_MAX_THREADS) nthreads = ORC_MAX_THREADS;
  if (nthreads <= 1) {
    if (end > first) fn(ctx, first, end, 0);
    return;
  }
  orc_pool *P = &g_orc_pool;
  if (!P->started) {
    pthread_mutex_init(&P->mu, 0);
    pthread_cond_init(&P->todo, 0);
    pthread_cond_init(&P->done, 0);
    P->generation = 0;
    P->quit = 0;
    P->nthreads = 0;
    P->started = 1;
    for (int i = 0; i < ORC_MAX_THREADS; i++) {
      g_orc_wargs[i].tid = i;
      pthread_create(&P->th[i], 0, orc_worker, &g_orc_wargs[i]);
    }
  }
  if (end <= first) return;
  if (step <= 0) step = ((end - first) + nthreads - 1) / nthreads;
  pthread_mutex_lock(&P->mu);
  P->fn = fn;
  P->ctx = ctx;
  P->next = first;
  P->end = end;
  P->step = step;
  P->nthreads = nthreads;
  P->running = nthreads;
  P->generation++;
  pthread_cond_broadcast(&P->todo);
  while (P->running > 0) pthread_cond_wait(&P->done, &P->mu);
  P->nthreads = 0;
  pthread_mutex_unlock(&P->mu);
}

#endif
