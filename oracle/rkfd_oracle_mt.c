/* rkfd_oracle_mt.c - bench.py's CPU baselines, timed inside C (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).
 *
 * The reference (mi-lib/roki-fd) is single-threaded; SURVEY 8d asks for the generous baseline too: every host core working on
 * disjoint instances.  Round 2 ran that leg as Python threads calling the oracle through ctypes once per rollout: 256 threads
 * contending for the interpreter lock measured 14.5 x one core.  Here the threads and the clock live in C:
 * every thread owns one oracle (the oracle keeps no global state) and does the bench's workload - rollouts of `horizon` steps
 * from the standing states: set state, forget contact / pivot state, rkFDUpdateInit, horizon x rkFDUpdate - until the deadline.
 */
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "rkfd_oracle.h"

typedef struct {
  const rkfdModel *m;
  const double *dis, *vel;   /* [ninst][ndof] */
  int ninst, horizon, first, stride;
  double seconds;
  long steps;
  double elapsed;
} Work;

static double now(void)
{
  struct timespec t;
  clock_gettime( CLOCK_MONOTONIC, &t );
  return (double)t.tv_sec + 1e-9*(double)t.tv_nsec;
}

static void *worker(void *arg)
{
  Work *w = (Work *)arg;
  const rkfdModel *m = w->m;
  rkfdOracle *o = rkfdOracleCreate( m );
  int *zi = (int *)calloc( (size_t)( m->ncand > m->nlink ? m->ncand : m->nlink ) + 1, sizeof(int) );
  double *zd = (double *)calloc( (size_t)3*( m->ncand > m->nlink ? m->ncand : m->nlink ) + 3, sizeof(double) );
  const double t0 = now();
  int inst = w->first;
  w->steps = 0;
  if( !o || !zi || !zd ){ free( zi ); free( zd ); if( o ) rkfdOracleDestroy( o ); return NULL; }
  while( now() - t0 < w->seconds ){
    const int k = inst % w->ninst;
    rkfdOracleSetState( o, w->dis + (size_t)k*m->ndof, w->vel + (size_t)k*m->ndof );
    if( m->ncand > 0 ) rkfdOracleSetContact( o, zi, zi, zd );
    rkfdOracleSetPivot( o, zi, zd );
    rkfdOracleSetBroken( o, zi );
    rkfdOracleUpdateInit( o );
    if( w->horizon > 0 ){
      rkfdOracleUpdateN( o, w->horizon );
      w->steps += w->horizon;
    } else {
      /* one continuous trajectory of up to 1000 steps, in blocks of 100 */
      int n = 0;
      while( n < 1000 && now() - t0 < w->seconds ){ rkfdOracleUpdateN( o, 100 ); n += 100; w->steps += 100; }
    }
    inst += w->stride;
  }
  w->elapsed = now() - t0;
  rkfdOracleDestroy( o );
  free( zi ); free( zd );
  return NULL;
}

/* nthreads OS threads for `seconds`; returns the steps done by all of them, *elapsed = the longest thread's wall time.
 * nthreads = 1 runs in the calling thread. */
long rkfdOracleRolloutsMT(const rkfdModel *m, int nthreads, int ninst, const double *dis, const double *vel, int horizon, double seconds, double *elapsed)
{
  Work *w;
  pthread_t *th;
  long total = 0;
  double el = 0;
  int i;
  if( nthreads < 1 ) nthreads = 1;
  w = (Work *)calloc( (size_t)nthreads, sizeof(Work) );
  th = (pthread_t *)calloc( (size_t)nthreads, sizeof(pthread_t) );
  if( !w || !th ){ free( w ); free( th ); return -1; }
  for( i=0; i<nthreads; i++ ){
    w[i].m = m; w[i].dis = dis; w[i].vel = vel; w[i].ninst = ninst; w[i].horizon = horizon;
    w[i].first = i; w[i].stride = nthreads; w[i].seconds = seconds;
  }
  if( nthreads == 1 ) worker( &w[0] );
  else {
    for( i=0; i<nthreads; i++ ) if( pthread_create( &th[i], NULL, worker, &w[i] ) != 0 ){ nthreads = i; break; }
    for( i=0; i<nthreads; i++ ) pthread_join( th[i], NULL );
  }
  for( i=0; i<nthreads; i++ ){ total += w[i].steps; if( w[i].elapsed > el ) el = w[i].elapsed; }
  if( elapsed ) *elapsed = el;
  free( w ); free( th );
  return total;
}
