/* rkfd_capi.hip - C ABI (include/rkfd_hip.h) over the gfx950 kernel in rkfd_device.h. */
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dlfcn.h>
#include <link.h>
#include <pthread.h>
#include <unistd.h>
#include <sys/stat.h>
#include <string>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rkfd_hip.h"
#include "rkfd_device.h"
#include "rkfd_devmodel_host.h"
#include "rkfd_device_src.inc"       /* the device headers as string literals (tools/embed_sources.py), for hipRTC */

static thread_local char g_err[512] = "";
#define SETERR(...) snprintf( g_err, sizeof(g_err), __VA_ARGS__ )
#define HIPCHK(call, ret) do{ hipError_t e_ = (call); if( e_ != hipSuccess ){ \
  SETERR( "%s failed: %s", #call, hipGetErrorString( e_ ) ); return ret; } }while(0)

/* one workgroup = one wavefront = one world instance; state lives in LDS for the whole launch.
 * Register budget: the kernels need ~150-160 VGPRs when built with -mllvm -disable-machine-licm (see Makefile):
 * three waves per SIMD, i.e. up to twelve instances per CU where the LDS allows (10 for the humanoid worlds).
 * With machine LICM on, the backend hoists ~35 registers of literals (sincos polynomial coefficients, fp64
 * constants) and addresses out of the step loop and the kernels need 184-193. */
#define RKFD_KERNEL(name, prof, vqp, pk, waves) \
extern "C" __global__ void __launch_bounds__(RKFD_WAVE, waves) \
name(rkfdDevModel m, rkfdDevState st, int first, int mode, int nsteps, int *errflag) \
{ \
  extern __shared__ __attribute__((aligned(16))) char lds[]; \
  const int b = first + (int)blockIdx.x; \
  if( b >= st.batch ) return; \
  rkfd_instance<prof, vqp, pk>( m, st, b, lds, mode, nsteps, errflag ); \
}
RKFD_KERNEL( rkfd_step_kernel, false, 0, false, 3 )
/* the contact matrix as a packed lower triangle (worlds where that lets one more instance share a CU) */
RKFD_KERNEL( rkfd_step_kernel_pk, false, 0, true, 3 )
/* the variant that also carries the Vert plugin's QP (worlds with rigid pairs under the Vert plugin);
 * kept apart so that its code and registers do not weigh on the MLCP / penalty kernel */
/* (two waves per SIMD: up to 24 unknowns the QP keeps the factor of its Q in registers - 96 of them - and the QP's LDS allows at
 *  most eight instances per CU anyway) */
RKFD_KERNEL( rkfd_step_kernel_vqp, false, 1, false, 2 )
/* diagnostic instantiations with in-kernel phase stamps (rkfdBatchProfile) */
RKFD_KERNEL( rkfd_step_kernel_prof, true, 0, false, 3 )
RKFD_KERNEL( rkfd_step_kernel_prof_pk, true, 0, true, 3 )
RKFD_KERNEL( rkfd_step_kernel_prof_vqp, true, 1, false, 2 )
/* the variant that carries the Volume plugin (worlds with rigid pairs under it).  Built for two waves per SIMD: the phase is
 * latency-bound (a dozen lanes at work, dependent LDS / readlane chains), so a second wave per SIMD is worth more than the
 * ~90 vector registers it spills (measured, box on the floor / humanoid on two soles, M steps/s: one wave 2.33 / 1.49, two
 * 4.55 / 1.68, three 4.47 / 1.41 - the humanoid's 32 KB of LDS allow five instances per CU either way) */
extern "C" __global__ void __launch_bounds__(RKFD_WAVE, 2)
rkfd_step_kernel_vol(rkfdDevModel m, rkfdDevState st, int first, int mode, int nsteps, int *errflag)
{
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int b = first + (int)blockIdx.x;
  if( b >= st.batch ) return;
  rkfd_instance<false, 2, false>( m, st, b, lds, mode, nsteps, errflag );
}
extern "C" __global__ void __launch_bounds__(RKFD_WAVE, 2)
rkfd_step_kernel_prof_vol(rkfdDevModel m, rkfdDevState st, int first, int mode, int nsteps, int *errflag)
{
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int b = first + (int)blockIdx.x;
  if( b >= st.batch ) return;
  rkfd_instance<true, 2, false>( m, st, b, lds, mode, nsteps, errflag );
}

/* rkfdBatchRestore: one workgroup copies one instance's state rows back from the snapshot */
extern "C" __global__ void __launch_bounds__(RKFD_WAVE)
rkfd_restore_kernel(rkfdDevState st, rkfdDevState sn, int first, int ND, int NLM, int NC)
{
  const size_t b = (size_t)first + blockIdx.x;
  if( b >= (size_t)st.batch ) return;
  for( int j=threadIdx.x; j<ND; j+=RKFD_WAVE ){
    st.dis[b*ND+j] = sn.dis[b*ND+j]; st.vel[b*ND+j] = sn.vel[b*ND+j]; st.acc[b*ND+j] = sn.acc[b*ND+j];
  }
  for( int j=threadIdx.x; j<NLM; j+=RKFD_WAVE ){
    st.piv_type[b*NLM+j] = sn.piv_type[b*NLM+j]; st.piv_prev[b*NLM+j] = sn.piv_prev[b*NLM+j];
    st.brk[b*NLM+j] = sn.brk[b*NLM+j];
  }
  for( int j=threadIdx.x; j<NC; j+=RKFD_WAVE ){
    st.cv_active[b*NC+j] = sn.cv_active[b*NC+j]; st.cv_type[b*NC+j] = sn.cv_type[b*NC+j];
  }
  for( int j=threadIdx.x; j<3*NC; j+=RKFD_WAVE ){
    st.cv_ref[b*3*NC+j] = sn.cv_ref[b*3*NC+j]; st.cv_f[b*3*NC+j] = sn.cv_f[b*3*NC+j];
  }
}

typedef void (*rkfdKernel)(rkfdDevModel, rkfdDevState, int, int, int, int *);
#define RKFD_MAX_SPLIT 8

struct rkfdBatch {
  int device, batch, nlink, ndof, ncand;
  rkfdDevModelHost host;
  rkfdDevModel dm;        /* device-pointer version */
  void *dblob;
  rkfdDevState st;
  rkfdDevState snap;      /* rkfdBatchSnapshot: device-resident copy of the state (allocated on first use) */
  int has_snap;
  int *d_err;
  size_t lds_bytes;
  rkfdKernel kern, kern_prof;
  /* rkfdBatchSpecialize: the step kernel compiled for this world (hipRTC); NULL = the generic kernels above */
  hipModule_t spec_mod;
  hipFunction_t spec_fn;
  /* rkfdBatchSetInstancesPerWave( b, 2 ): a second device model (sweep schedule with four links per iteration) for the
   * world-specific kernel built with RKFD_W = 2 - two instances per wavefront, 32 lanes each */
  int steps_per_launch;      /* under split launches: steps one launch carries (rkfdBatchSetStepsPerLaunch) */
  int ipw;                           /* instances per wavefront of the specialised kernel: 1 or 2 */
  const rkfdModel *model_for_w2;     /* the caller's model (must outlive the batch, as for rkfdBatchCreate's own use) */
  rkfdDevModelHost host2;
  rkfdDevModel dm2;
  void *dblob2;
  /* split launches (rkfdBatchSetSplit): the batch goes out as nsplit kernels on internal streams, so that the
   * tail of one step of one part overlaps the next step of another (the instances are independent) */
  int nsplit;
  hipStream_t sub[RKFD_MAX_SPLIT];
  hipEvent_t fork, done[RKFD_MAX_SPLIT];
  int pending;                       /* work on the internal streams that no stream has been joined with yet */
  /* optional per-launch timing (rkfdBatchTimeLaunches) */
  int timing;
  std::vector<hipEvent_t> *tev;      /* pool of pre-created events; start / stop pairs occupy [0, tused) */
  size_t tused;
};

extern "C" const char *rkfdHipLastError(void){ return g_err; }

extern "C" int rkfdHipDeviceCount(void)
{
  int n = 0;
  if( hipGetDeviceCount( &n ) != hipSuccess ) return 0;
  return n;
}

template<class T> static int dalloc(T **p, size_t n)
{
  hipError_t e = hipMalloc( (void **)p, sizeof(T)*( n ? n : 1 ) );
  if( e != hipSuccess ){ SETERR( "hipMalloc(%zu bytes) failed: %s", sizeof(T)*n, hipGetErrorString( e ) ); return -1; }
  e = hipMemset( *p, 0, sizeof(T)*( n ? n : 1 ) );
  if( e != hipSuccess ){ SETERR( "hipMemset failed: %s", hipGetErrorString( e ) ); return -1; }
  return 0;
}

extern "C" rkfdBatch *rkfdBatchCreate(const rkfdModel *m, int batch, int device, int max_rigid)
{
  if( !m || batch < 1 ){ SETERR( "rkfdBatchCreate: bad arguments" ); return NULL; }
  int ndev = 0;
  if( hipGetDeviceCount( &ndev ) != hipSuccess || ndev == 0 ){
    SETERR( "rkfdBatchCreate: no HIP device available (the MI355X path has no CPU fallback)" );
    return NULL;
  }
  if( device < 0 || device >= ndev ){ SETERR( "rkfdBatchCreate: device %d out of range (%d visible)", device, ndev ); return NULL; }
  HIPCHK( hipSetDevice( device ), NULL );

  rkfdBatch *b = (rkfdBatch *)calloc( 1, sizeof(rkfdBatch) );
  if( b ){ b->nsplit = 1; b->steps_per_launch = 5; b->tev = new std::vector<hipEvent_t>(); }
  if( !b ){ SETERR( "out of memory" ); return NULL; }
  char err[256];
  if( rkfd_devmodel_build( m, max_rigid, &b->host, err, sizeof(err) ) < 0 ){
    SETERR( "rkfdBatchCreate: %s", err );
    rkfdBatchDestroy( b );
    return NULL;
  }
  b->device = device; b->batch = batch; b->nlink = m->nlink; b->ndof = m->ndof; b->ncand = m->ncand;
  b->lds_bytes = b->host.lds_bytes;
  /* diagnostic (tools/sweep_residency.sh): ask for more LDS than needed, to measure throughput against residency */
  if( const char *pad = getenv( "RKFD_LDS_PAD_BYTES" ) ) b->lds_bytes += (size_t)( atoi( pad ) > 0 ? atoi( pad ) : 0 );
  if( b->lds_bytes > 160*1024 ){
    SETERR( "rkfdBatchCreate: one instance needs %zu bytes of LDS (> 160 KiB)", b->lds_bytes );
    rkfdBatchDestroy( b );
    return NULL;
  }
  if( hipMalloc( &b->dblob, b->host.bytes ) != hipSuccess ||
      hipMemcpy( b->dblob, b->host.blob, b->host.bytes, hipMemcpyHostToDevice ) != hipSuccess ){
    SETERR( "rkfdBatchCreate: cannot copy the model to the device" );
    rkfdBatchDestroy( b );
    return NULL;
  }
  b->dm = b->host.dm;
  rkfd_devmodel_rebase( &b->dm, b->host.blob, b->dblob );
  b->ipw = 1; b->model_for_w2 = m;

  const size_t B = batch, ND = m->ndof, NL = m->nlink, NC = m->ncand;
  int bad = 0;
  bad |= dalloc( &b->st.dis, B*ND ); bad |= dalloc( &b->st.vel, B*ND ); bad |= dalloc( &b->st.acc, B*ND );
  bad |= dalloc( &b->st.motor_in, B*NL ); bad |= dalloc( &b->st.piv_type, B*NL ); bad |= dalloc( &b->st.piv_prev, B*NL );
  bad |= dalloc( &b->st.brk, B*NL );
  bad |= dalloc( &b->st.cv_active, B*NC ); bad |= dalloc( &b->st.cv_type, B*NC );
  bad |= dalloc( &b->st.cv_ref, B*NC*3 ); bad |= dalloc( &b->st.cv_f, B*NC*3 );
  bad |= dalloc( &b->st.stat, B*4 );
  bad |= dalloc( &b->d_err, 1 );
  b->st.dbg = NULL; b->st.dbg_stride = 0; b->st.batch = batch; b->st.prof = NULL;
  if( bad ){ rkfdBatchDestroy( b ); return NULL; }
  b->kern = b->dm.vert_rigid ? rkfd_step_kernel_vqp : ( b->dm.ma_packed ? rkfd_step_kernel_pk : rkfd_step_kernel );
  b->kern_prof = b->dm.vert_rigid ? rkfd_step_kernel_prof_vqp : ( b->dm.ma_packed ? rkfd_step_kernel_prof_pk : rkfd_step_kernel_prof );
  if( b->dm.vol_np > 0 ){ b->kern = rkfd_step_kernel_vol; b->kern_prof = rkfd_step_kernel_prof_vol; }
  if( b->lds_bytes > 64*1024 ){
    hipError_t e = hipFuncSetAttribute( (const void *)b->kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes );
    if( e == hipSuccess ) e = hipFuncSetAttribute( (const void *)b->kern_prof, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_bytes );
    if( e != hipSuccess ){ SETERR( "hipFuncSetAttribute(LDS=%zu) failed: %s", b->lds_bytes, hipGetErrorString( e ) ); rkfdBatchDestroy( b ); return NULL; }
  }
  return b;
}

extern "C" void rkfdBatchDestroy(rkfdBatch *b)
{
  if( !b ) return;
  (void)hipFree( b->st.dis ); (void)hipFree( b->st.vel ); (void)hipFree( b->st.acc );
  (void)hipFree( b->st.motor_in ); (void)hipFree( b->st.piv_type ); (void)hipFree( b->st.piv_prev ); (void)hipFree( b->st.brk );
  (void)hipFree( b->st.cv_active ); (void)hipFree( b->st.cv_type ); (void)hipFree( b->st.cv_ref ); (void)hipFree( b->st.cv_f );
  (void)hipFree( b->st.stat );
  (void)hipFree( b->snap.dis ); (void)hipFree( b->snap.vel ); (void)hipFree( b->snap.acc );
  (void)hipFree( b->snap.piv_type ); (void)hipFree( b->snap.piv_prev ); (void)hipFree( b->snap.brk );
  (void)hipFree( b->snap.cv_active ); (void)hipFree( b->snap.cv_type ); (void)hipFree( b->snap.cv_ref ); (void)hipFree( b->snap.cv_f );
  if( b->fork ){
    (void)hipEventDestroy( b->fork );
    for( int k=0; k<RKFD_MAX_SPLIT; k++ ){ (void)hipStreamSynchronize( b->sub[k] ); (void)hipStreamDestroy( b->sub[k] ); (void)hipEventDestroy( b->done[k] ); }
  }
  if( b->tev ){ for( size_t i=0; i<b->tev->size(); i++ ) (void)hipEventDestroy( (*b->tev)[i] ); delete b->tev; }
  if( b->spec_mod ) (void)hipModuleUnload( b->spec_mod );
  (void)hipFree( b->d_err ); (void)hipFree( b->dblob ); (void)hipFree( b->dblob2 );
  rkfd_devmodel_free( &b->host ); rkfd_devmodel_free( &b->host2 );
  free( b );
}

/* host-only: LDS bytes one instance of world m would occupy (no GPU needed) */
extern "C" int rkfdLdsBytesFor(const rkfdModel *m, int max_rigid)
{
  rkfdDevModelHost h;
  char err[256];
  if( !m || rkfd_devmodel_build( m, max_rigid, &h, err, sizeof(err) ) < 0 ){ SETERR( "rkfdLdsBytesFor: %s", m ? err : "null model" ); return -1; }
  const int n = (int)h.lds_bytes;
  rkfd_devmodel_free( &h );
  return n;
}

extern "C" int rkfdBatchSize(const rkfdBatch *b){ return b ? b->batch : -1; }
extern "C" int rkfdBatchDof(const rkfdBatch *b){ return b ? b->ndof : -1; }
extern "C" int rkfdBatchLdsBytes(const rkfdBatch *b){ return b ? (int)b->lds_bytes : -1; }
extern "C" int rkfdBatchResidency(const rkfdBatch *b)
{
  int n = 0;
  if( !b ) return -1;
  if( hipOccupancyMaxActiveBlocksPerMultiprocessor( &n, (const void *)b->kern, RKFD_WAVE, b->lds_bytes ) != hipSuccess ) return -1;
  /* the runtime's answer is optimistic about LDS: the hardware hands it out in 1280-byte pieces, 128 per CU
   * (measured with tools/ubench/residency.hip, profiles/r01_lds_residency.txt), e.g. 16160 bytes -> 9, not 10 */
  const int pieces = (int)( ( b->lds_bytes + 1279 )/1280 );
  if( pieces > 0 && 128/pieces < n ) n = 128/pieces;
  if( b->ipw == 2 && b->spec_fn ){
    /* two instances per wavefront: workgroups of 2 x the LDS, at most two waves per SIMD (the kernel is built for that) */
    const int p2 = (int)( ( 2*b->host2.lds_bytes + (size_t)b->host2.dm.lds_shared + 1279 )/1280 );
    int w = p2 > 0 ? 128/p2 : 0;
    if( w > 8 ) w = 8;
    n = 2*w;
  }
  return n;
}
extern "C" double *rkfdBatchDevDis(rkfdBatch *b){ return b ? b->st.dis : NULL; }
extern "C" double *rkfdBatchDevVel(rkfdBatch *b){ return b ? b->st.vel : NULL; }
extern "C" double *rkfdBatchDevAcc(rkfdBatch *b){ return b ? b->st.acc : NULL; }

#define H2D(dst, src, bytes) do{ if( src ) HIPCHK( hipMemcpy( dst, src, bytes, hipMemcpyHostToDevice ), -1 ); }while(0)
#define D2H(dst, src, bytes) do{ if( dst ) HIPCHK( hipMemcpy( dst, src, bytes, hipMemcpyDeviceToHost ), -1 ); }while(0)

static int sync_streams(rkfdBatch *b);      /* the synchronous accessors wait for split launches first */

extern "C" int rkfdBatchSetState(rkfdBatch *b, const double *dis, const double *vel)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  const size_t n = sizeof(double)*(size_t)b->batch*b->ndof;
  H2D( b->st.dis, dis, n ); H2D( b->st.vel, vel, n );
  return 0;
}
extern "C" int rkfdBatchGetState(rkfdBatch *b, double *dis, double *vel, double *acc)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  const size_t n = sizeof(double)*(size_t)b->batch*b->ndof;
  D2H( dis, b->st.dis, n ); D2H( vel, b->st.vel, n ); D2H( acc, b->st.acc, n );
  return 0;
}
extern "C" int rkfdBatchSetMotorInput(rkfdBatch *b, const double *input)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  H2D( b->st.motor_in, input, sizeof(double)*(size_t)b->batch*b->nlink );
  return 0;
}
extern "C" int rkfdBatchGetContact(rkfdBatch *b, int *active, int *type, double *ref, double *f)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  const size_t n = (size_t)b->batch*b->ncand;
  std::vector<int> act( n );
  HIPCHK( hipMemcpy( act.data(), b->st.cv_active, sizeof(int)*n, hipMemcpyDeviceToHost ), -1 );
  D2H( type, b->st.cv_type, sizeof(int)*n );
  D2H( ref, b->st.cv_ref, sizeof(double)*3*n ); D2H( f, b->st.cv_f, sizeof(double)*3*n );
  /* a candidate out of contact has no state: the device keeps whatever it last held there
   * (it is rewritten at the next first contact), the boundary reports zeros */
  if( ref && n ) rkfd_ref_to_model( &b->host, ref, n );      /* device link frame -> model link frame */
  for( size_t i=0; i<n; i++ ){
    if( active ) active[i] = act[i];
    if( act[i] ) continue;
    if( type ) type[i] = 0;
    if( ref ) ref[3*i] = ref[3*i+1] = ref[3*i+2] = 0;
    if( f ) f[3*i] = f[3*i+1] = f[3*i+2] = 0;
  }
  return 0;
}
extern "C" int rkfdBatchSetContact(rkfdBatch *b, const int *active, const int *type, const double *ref)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  const size_t n = (size_t)b->batch*b->ncand;
  H2D( b->st.cv_active, active, sizeof(int)*n ); H2D( b->st.cv_type, type, sizeof(int)*n );
  if( ref && n ){
    std::vector<double> rd( ref, ref + 3*n );
    rkfd_ref_to_device( &b->host, rd.data(), n );             /* model link frame -> device link frame */
    HIPCHK( hipMemcpy( b->st.cv_ref, rd.data(), sizeof(double)*3*n, hipMemcpyHostToDevice ), -1 );
  }
  return 0;
}
extern "C" int rkfdBatchGetPivot(rkfdBatch *b, int *type, double *prev_trq)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  const size_t n = (size_t)b->batch*b->nlink;
  D2H( type, b->st.piv_type, sizeof(int)*n ); D2H( prev_trq, b->st.piv_prev, sizeof(double)*n );
  return 0;
}
/* breakable float joints: 1 per link whose joint has broken, [batch][nlink] */
extern "C" int rkfdBatchGetBroken(rkfdBatch *b, int *broken)
{
  if( !b || !broken ){ SETERR( "null argument" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  D2H( broken, b->st.brk, sizeof(int)*(size_t)b->batch*b->nlink );
  return 0;
}
extern "C" int rkfdBatchSetBroken(rkfdBatch *b, const int *broken)
{
  if( !b || !broken ){ SETERR( "null argument" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  H2D( b->st.brk, broken, sizeof(int)*(size_t)b->batch*b->nlink );
  return 0;
}
extern "C" int rkfdBatchSetPivot(rkfdBatch *b, const int *type, const double *prev_trq)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  const size_t n = (size_t)b->batch*b->nlink;
  H2D( b->st.piv_type, type, sizeof(int)*n ); H2D( b->st.piv_prev, prev_trq, sizeof(double)*n );
  return 0;
}

/* next start / stop pair of the timing pool (grown in blocks: creating events inside the launch path is expensive);
 * pairs are handed out whole, so a failed creation never leaves the pool misaligned */
static int timing_pair(rkfdBatch *b, hipEvent_t *e0, hipEvent_t *e1)
{
  *e0 = *e1 = NULL;
  if( b->tused + 2 > b->tev->size() ){
    for( int i=0; i<1024; i++ ){
      hipEvent_t a = NULL, c = NULL;
      if( hipEventCreate( &a ) != hipSuccess ) break;
      if( hipEventCreate( &c ) != hipSuccess ){ (void)hipEventDestroy( a ); break; }
      b->tev->push_back( a ); b->tev->push_back( c );
    }
    if( b->tused + 2 > b->tev->size() ) return 0;
  }
  *e0 = (*b->tev)[b->tused]; *e1 = (*b->tev)[b->tused+1];
  b->tused += 2;
  return 1;
}
/* make `stream` wait for everything the internal streams hold (no host synchronisation) */
static int join_streams(rkfdBatch *b, hipStream_t stream)
{
  if( b->nsplit > 1 && b->pending ){
    for( int k=0; k<b->nsplit; k++ ) HIPCHK( hipStreamWaitEvent( stream, b->done[k], 0 ), -1 );
    b->pending = 0;
  }
  return 0;
}
/* host waits for the internal streams (used by the synchronous accessors) */
static int sync_streams(rkfdBatch *b)
{
  if( b->nsplit > 1 ) for( int k=0; k<b->nsplit; k++ ) HIPCHK( hipStreamSynchronize( b->sub[k] ), -1 );
  b->pending = 0;
  return 0;
}
/* one kernel launch over `count` instances starting at `first`: the kernel compiled for this world when there is one */
static int launch_one(rkfdBatch *b, rkfdKernel kern, int count, int first, int mode, int nsteps, hipStream_t stream)
{
  if( b->spec_fn && !b->st.prof ){
    if( b->ipw == 2 ){
      /* two instances per wavefront: half as many workgroups, each with the LDS of two instances; the kernel learns where the
       * part ends through st.batch (a half beyond it is a stand-in that stores nothing) */
      rkfdDevState st2 = b->st;
      st2.batch = first + count;
      void *args[] = { &b->dm2, &st2, &first, &mode, &nsteps, &b->d_err };
      HIPCHK( hipModuleLaunchKernel( b->spec_fn, ( count+1 )/2, 1, 1, RKFD_WAVE, 1, 1, (unsigned)( 2*b->host2.lds_bytes + (size_t)b->host2.dm.lds_shared ), stream, args, NULL ), -1 );
      return 0;
    }
    void *args[] = { &b->dm, &b->st, &first, &mode, &nsteps, &b->d_err };
    HIPCHK( hipModuleLaunchKernel( b->spec_fn, count, 1, 1, RKFD_WAVE, 1, 1, (unsigned)b->lds_bytes, stream, args, NULL ), -1 );
    return 0;
  }
  hipLaunchKernelGGL( kern, dim3( count ), dim3( RKFD_WAVE ), b->lds_bytes, stream, b->dm, b->st, first, mode, nsteps, b->d_err );
  HIPCHK( hipGetLastError(), -1 );
  return 0;
}

static int launch(rkfdBatch *b, int mode, int nsteps, void *stream)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  rkfdKernel kern = b->st.prof ? b->kern_prof : b->kern;
  if( b->nsplit <= 1 || b->st.prof ){
    if( b->st.prof && sync_streams( b ) < 0 ) return -1;
    hipEvent_t e0 = NULL, e1 = NULL;
    if( b->timing && !b->st.prof && timing_pair( b, &e0, &e1 ) ) HIPCHK( hipEventRecord( e0, (hipStream_t)stream ), -1 );
    if( launch_one( b, kern, b->batch, 0, mode, nsteps, (hipStream_t)stream ) < 0 ) return -1;
    if( e0 && e1 ) HIPCHK( hipEventRecord( e1, (hipStream_t)stream ), -1 );
    return 0;
  }
  /* fork: the parts start after what the caller's stream holds now; they do NOT wait for each other, and the
   * caller's stream does not wait for them until rkfdBatchJoin / rkfdBatchStatus */
  HIPCHK( hipEventRecord( b->fork, (hipStream_t)stream ), -1 );
  for( int k=0; k<b->nsplit; k++ ) HIPCHK( hipStreamWaitEvent( b->sub[k], b->fork, 0 ), -1 );
  /* several steps go out as that many rounds of one-step launches, not as fused kernels: a fused launch holds its
   * slots for all its steps, so with more instances than slots the rest of the batch waits that long and the tail
   * of the last round runs thin (4096 instances, 200 steps: 9.2 M steps/s fused, 12.7 M in rounds; tools/fused_vs_stepwise.py) */
  /* (not for worlds under the Vert plugin: the QP makes the step times of the instances vary widely, and there the
   * per-step barrier of a round costs more than the slots a fused launch holds: 2.1 M against 3.6 M steps/s) */
  /* (round 3: not one step per launch either - rounds of b->steps_per_launch steps, 5 by default: rollouts of 25 steps on config 4,
   *  steps per launch 1 / 2 / 3 / 5 / 9 / 13 / 25: 11.51 / 11.63 / 11.65 / 11.82 / 11.79 / 11.82 / 11.76 M steps/s - a launch loads
   *  an instance's state and the world's tables once, against slots held a little longer; profiles/r03_steps_per_launch.txt) */
  int rounds = 1, per = nsteps;
  if( mode == 0 && nsteps > 1 && !b->dm.vert_rigid ){ per = b->steps_per_launch < nsteps ? b->steps_per_launch : nsteps; rounds = ( nsteps + per - 1 )/per; }
  for( int r=0; r<rounds; r++ )
    for( int k=0; k<b->nsplit; k++ ){
      const int lo = (int)( (long long)b->batch*k/b->nsplit ), hi = (int)( (long long)b->batch*( k+1 )/b->nsplit );
      if( hi <= lo ) continue;
      hipEvent_t e0 = NULL, e1 = NULL;
      if( b->timing && timing_pair( b, &e0, &e1 ) ) HIPCHK( hipEventRecord( e0, b->sub[k] ), -1 );
      if( launch_one( b, kern, hi-lo, lo, mode, ( r+1 )*per <= nsteps ? per : nsteps - r*per, b->sub[k] ) < 0 ) return -1;
      if( e0 && e1 ) HIPCHK( hipEventRecord( e1, b->sub[k] ), -1 );
    }
  for( int k=0; k<b->nsplit; k++ ) HIPCHK( hipEventRecord( b->done[k], b->sub[k] ), -1 );
  b->pending = 1;
  return 0;
}
/* ---- the step kernel compiled for one world (hipRTC) ---------------------------------------------------- */
/* source of the specialised kernel: the dimensions of the world as literals in front of the same device code */
static std::string spec_source(const rkfdDevModel &d, int ipw = 1)
{
  char buf[4096];
  snprintf( buf, sizeof(buf),
    "#define RKFD_SPEC 1\n#define RKFD_W %d\n"
    "#define RKFD_SPEC_NLINK %d\n#define RKFD_SPEC_NDOF %d\n#define RKFD_SPEC_NCAND %d\n#define RKFD_SPEC_NLINK_MODEL %d\n"
    "#define RKFD_SPEC_NLEVEL %d\n#define RKFD_SPEC_NROUND %d\n#define RKFD_SPEC_NSCHED %d\n#define RKFD_SPEC_MAXRG %d\n"
    "#define RKFD_SPEC_NPOOL %d\n#define RKFD_SPEC_NFLOAT %d\n#define RKFD_SPEC_MAXACT %d\n#define RKFD_SPEC_NSIDE %d\n"
    "#define RKFD_SPEC_NPUROW %d\n#define RKFD_SPEC_PU_D0 %d\n#define RKFD_SPEC_PU_ALIAS %d\n#define RKFD_SPEC_VERT_RIGID %d\n#define RKFD_SPEC_QSCR_ALIAS %d\n"
    "#define RKFD_SPEC_HAS_SLIDE %d\n#define RKFD_SPEC_MA_SIZE %d\n#define RKFD_SPEC_MA_PACKED %d\n"
    "#define RKFD_SPEC_MAX_ITER %d\n#define RKFD_SPEC_SOLVER %d\n#define RKFD_SPEC_PYRAMID %d\n#define RKFD_SPEC_ANCHOR %d\n#define RKFD_SPEC_MLCP_MFMA %d\n"
    "#define RKFD_SPEC_HAS_BRF %d\n#define RKFD_SPEC_LDS_INSTANCE %d\n#define RKFD_SPEC_LDS_SHARED %d\n"
    "#define RKFD_SPEC_VOL_NPAIR %d\n#define RKFD_SPEC_VOL_NP %d\n#define RKFD_SPEC_VOL_NCP %d\n#define RKFD_SPEC_VOL_PV %d\n#define RKFD_SPEC_VOL_NF %d\n"
    "#include \"rkfd_device.h\"\n"
    "extern \"C\" __global__ void __launch_bounds__(64, %d)\n"
    "rkfd_step_kernel_spec(rkfdDevModel m, rkfdDevState st, int first, int mode, int nsteps, int *errflag)\n"
    "{\n"
    "  extern __shared__ __attribute__((aligned(16))) char lds[];\n"
    "  int b = first + (int)blockIdx.x*RKFD_W + HALF();\n"
    "  if( RKFD_W == 1 && b >= st.batch ) return;\n"
    "  const bool live = b < st.batch;\n"
    "  if( !live ) b -= 1;\n"
    "  rkfd_instance<false, %s, %s>( m, st, b, lds + HALF()*RKFD_SPEC_LDS_INSTANCE, mode, nsteps, errflag, live, lds + RKFD_W*RKFD_SPEC_LDS_INSTANCE );\n"
    "}\n",
    ipw, d.nlink, d.ndof, d.ncand, d.nlink_model, d.nlevel, d.nround, d.nsched, d.maxrg, d.npool, d.nfloat, d.maxact, d.nside,
    d.npurow, d.pu_d0, d.pu_alias, d.vert_rigid, d.qscr_alias, d.has_slide, d.ma_size, d.ma_packed, d.max_iter, d.solver, d.pyramid, d.anchor, d.mlcp_mfma,
    d.has_brf, d.lds_instance, ipw == 2 ? d.lds_shared : 0, d.vol_npair, d.vol_np, d.vol_ncp, d.vol_pv, d.vol_nf,
    ( d.vol_np > 0 || ipw == 2 || d.vert_rigid == 2 ) ? 2 : 3, d.vol_np > 0 ? "2" : ( d.vert_rigid ? "1" : "0" ), d.ma_packed ? "true" : "false" );
  std::string src;
  if( const char *pre = getenv( "RKFD_SPEC_DEFINE" ) ){      /* diagnostic: NAME[,NAME...] defined as 1 in front of the source */
    std::string names( pre ); size_t p0 = 0;
    while( p0 < names.size() ){ size_t p1 = names.find( ',', p0 ); if( p1 == std::string::npos ) p1 = names.size(); src += "#define " + names.substr( p0, p1-p0 ) + " 1\n"; p0 = p1+1; }
  }
  return src + std::string( buf );
}
/* hipRTC, bound at run time in a PRIVATE link namespace.  hipRTC finds its compiler (libamd_comgr) by soname, and
 * a process serves every request for a soname with the first library it loaded under it: a host program that has
 * loaded a framework bundling an older comgr (PyTorch does) would silently get that compiler - with it this kernel
 * spills (437 VGPR spills, 3.9 M instead of 14 M steps/s on the humanoid).  dlmopen( LM_ID_NEWLM ) gives hipRTC and
 * the comgr next to it (RUNPATH $ORIGIN) a namespace of their own, whatever the host program loaded before or
 * after - no preload needed from a C, Python or any other caller.  The ROCm library directory is the one this
 * library was built against (RKFD_ROCM_LIBDIR, from the Makefile) unless RKFD_ROCM_LIBDIR / ROCM_PATH say otherwise;
 * RKFD_RTC_NAMESPACE=shared skips the private namespace (diagnostic). */
#ifndef RKFD_ROCM_LIBDIR
#define RKFD_ROCM_LIBDIR "/opt/rocm/lib"
#endif
struct rkfdRtc {
  void *h;
  int priv;     /* 1: loaded in a private namespace */
  hiprtcResult (*create)(hiprtcProgram *, const char *, const char *, int, const char **, const char **);
  hiprtcResult (*compile)(hiprtcProgram, int, const char **);
  hiprtcResult (*logsize)(hiprtcProgram, size_t *);
  hiprtcResult (*log)(hiprtcProgram, char *);
  hiprtcResult (*codesize)(hiprtcProgram, size_t *);
  hiprtcResult (*code)(hiprtcProgram, char *);
  hiprtcResult (*destroy)(hiprtcProgram *);
  const char *(*errstr)(hiprtcResult);
};
static const rkfdRtc *rtc_api(void)
{
  static rkfdRtc api;
  static int state = 0;          /* 0 not tried, 1 ready, -1 failed */
  static pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
  pthread_mutex_lock( &mu );
  if( state == 0 ){
    std::string dir = RKFD_ROCM_LIBDIR;
    if( const char *e = getenv( "RKFD_ROCM_LIBDIR" ) ) dir = e;
    else if( const char *r = getenv( "ROCM_PATH" ) ){ std::string d = std::string( r ) + "/lib"; if( access( ( d + "/libhiprtc.so" ).c_str(), R_OK ) == 0 ) dir = d; }
    const char *mode = getenv( "RKFD_RTC_NAMESPACE" );
    const char *names[] = { "/libhiprtc.so.7", "/libhiprtc.so" };
    void *h = NULL; int priv = 0;
    if( !mode || strcmp( mode, "shared" ) != 0 )
      for( int k=0; k<2 && !h; k++ ) h = dlmopen( LM_ID_NEWLM, ( dir + names[k] ).c_str(), RTLD_NOW | RTLD_LOCAL );
    if( h ) priv = 1;
    for( int k=0; k<2 && !h; k++ ) h = dlopen( ( dir + names[k] ).c_str(), RTLD_NOW | RTLD_LOCAL );
    if( !h ) h = dlopen( "libhiprtc.so", RTLD_NOW | RTLD_LOCAL );
    if( !h ){ SETERR( "hipRTC: cannot load libhiprtc from %s: %s", dir.c_str(), dlerror() ); state = -1; }
    else {
      api.h = h; api.priv = priv;
#define RTCSYM(field, name) *(void **)&api.field = dlsym( h, name )
      RTCSYM( create, "hiprtcCreateProgram" ); RTCSYM( compile, "hiprtcCompileProgram" ); RTCSYM( logsize, "hiprtcGetProgramLogSize" );
      RTCSYM( log, "hiprtcGetProgramLog" ); RTCSYM( codesize, "hiprtcGetCodeSize" ); RTCSYM( code, "hiprtcGetCode" );
      RTCSYM( destroy, "hiprtcDestroyProgram" ); RTCSYM( errstr, "hiprtcGetErrorString" );
#undef RTCSYM
      if( !api.create || !api.compile || !api.logsize || !api.log || !api.codesize || !api.code || !api.destroy || !api.errstr ){
        SETERR( "hipRTC: %s lacks an entry point", dir.c_str() ); state = -1;
      } else state = 1;
    }
  }
  const int s = state;
  pthread_mutex_unlock( &mu );
  return s == 1 ? &api : NULL;
}
/* ---- ahead-of-time specialised kernels ---------------------------------------------------------------------
 * The world-specific kernel is a pure function of (the generated preamble, the device sources the library carries, the compiler
 * options): its code object can be made when the library is built and merely LOADED at run time.  `make spec` does that for the
 * worlds of BASELINE.json's configurations (tools/make_spec.py -> roki-fd_amd/spec/rkfd_spec_<key>.co, beside the library), so the
 * headline needs no run-time compiler; any other world is compiled through hipRTC on first use and - when the directory is
 * writable - kept there too.  The key is a 64-bit FNV-1a hash over preamble, sources and options: a stale file cannot be picked
 * up after the device code changed.  RKFD_SPEC_DIR names another directory; RKFD_SPEC_STORE=0 switches the store off. */
static unsigned long long spec_key(const std::string &src, const char *const *opts, int nopts)
{
  unsigned long long h = 1469598103934665603ull;
  auto eat = [&h](const char *p){ for( ; *p; p++ ){ h ^= (unsigned char)*p; h *= 1099511628211ull; } h ^= 0xffu; h *= 1099511628211ull; };
  eat( src.c_str() );
  for( int i=0; i<rkfd_src_count; i++ ){ eat( rkfd_src_name[i] ); eat( rkfd_src_text[i] ); }
  for( int i=0; i<nopts; i++ ) eat( opts[i] );
  return h;
}
static std::string spec_dir(void)
{
  if( const char *e = getenv( "RKFD_SPEC_DIR" ) ) return e;
  Dl_info di;
  if( dladdr( (const void *)&spec_key, &di ) && di.dli_fname ){
    std::string p( di.dli_fname );
    const size_t k = p.rfind( '/' );
    return ( k == std::string::npos ? std::string( "." ) : p.substr( 0, k ) ) + "/spec";
  }
  return "spec";
}
static bool spec_store_on(void){ const char *e = getenv( "RKFD_SPEC_STORE" ); return !( e && atoi( e ) == 0 ); }
static std::string spec_path(unsigned long long key)
{
  char nm[64];
  snprintf( nm, sizeof(nm), "/rkfd_spec_%016llx.co", key );
  return spec_dir() + nm;
}
static int spec_from_store(unsigned long long key, std::vector<char> &code)
{
  if( !spec_store_on() ) return -1;
  FILE *f = fopen( spec_path( key ).c_str(), "rb" );
  if( !f ) return -1;
  fseek( f, 0, SEEK_END ); const long n = ftell( f ); fseek( f, 0, SEEK_SET );
  int r = -1;
  if( n > 0 ){ code.resize( (size_t)n ); if( fread( code.data(), 1, (size_t)n, f ) == (size_t)n ) r = 0; }
  fclose( f );
  if( r == 0 && getenv( "RKFD_SPEC_DEBUG" ) ) fprintf( stderr, "rkfd: specialised kernel loaded from %s (no run-time compile)\n", spec_path( key ).c_str() );
  return r;
}
static void spec_to_store(unsigned long long key, const std::vector<char> &code)
{
  if( !spec_store_on() ) return;
  char uniq[48];
  snprintf( uniq, sizeof(uniq), ".tmp.%ld.%p", (long)getpid(), (void *)&code );      /* (several ranks of a node may compile the same world at once) */
  const std::string dir = spec_dir(), path = spec_path( key ), tmp = path + uniq;
  (void)mkdir( dir.c_str(), 0777 );
  FILE *f = fopen( tmp.c_str(), "wb" );
  if( !f ) return;                                  /* (a read-only installation: compile every time) */
  const bool ok = fwrite( code.data(), 1, code.size(), f ) == code.size();
  fclose( f );
  if( ok ) (void)rename( tmp.c_str(), path.c_str() ); else (void)remove( tmp.c_str() );
}
static int g_spec_last_from_store = 0;      /* diagnostic: did the last spec_compile of this thread's process hit the store */
extern "C" int rkfdSpecializeLastFromStore(void){ return g_spec_last_from_store; }

/* compile for gfx950 from the sources the library carries (rkfd_device_src.inc): nothing is read from disk but the store above */
static int spec_compile(const rkfdDevModel &d, std::vector<char> &code, int ipw = 1)
{
  const std::string src = spec_source( d, ipw );
  const char *opts[] = { "--offload-arch=gfx950", "-O3", "-Wno-unused-value", "-mllvm", "-disable-machine-licm" };
  const int nopts = (int)( sizeof(opts)/sizeof(opts[0]) );
  const unsigned long long key = spec_key( src, opts, nopts );
  g_spec_last_from_store = 0;
  if( !getenv( "RKFD_SPEC_DUMP" ) && spec_from_store( key, code ) == 0 ){
    g_spec_last_from_store = 1;
    if( const char *dump = getenv( "RKFD_SPEC_DUMP_CODE" ) ){ FILE *f = fopen( dump, "wb" ); if( f ){ fwrite( code.data(), 1, code.size(), f ); fclose( f ); } }
    return 0;
  }
  const rkfdRtc *rtc = rtc_api();
  if( !rtc ) return -1;
  /* one compile at a time: the environment snapshot below is per namespace, not per call */
  static pthread_mutex_t cmu = PTHREAD_MUTEX_INITIALIZER;
  struct Unlock { pthread_mutex_t *m; ~Unlock(){ pthread_mutex_unlock( m ); } } unlock = { &cmu };
  pthread_mutex_lock( &cmu );
  if( rtc->priv ){
    /* The private namespace carries its own copy of the C library, and that copy's `environ` still points at the array the
     * process had when the namespace was made.  The host's setenv (Python's os.environ, say) moves the real array and frees
     * the old one: the compiler's getenv would walk freed memory (seen: a segmentation fault in the 88th test of a pytest
     * process that had set new variables between two specialisations).  Round 2 pointed the copy at the host's live array
     * before every compile - which the host can move again DURING the compile (another thread's setenv), and which helper
     * threads or exit handlers of the namespace may still read after it.  Now the namespace gets an environment
     * of its OWN: a deep copy (array and strings) taken under this mutex before every compile, which nothing but this
     * function ever replaces.  Earlier snapshots are kept alive as long as the library is loaded (a thread of the namespace
     * may still hold a pointer into one): a few kilobytes per compile, one compile per world. */
    char ***penv = (char ***)dlsym( rtc->h, "environ" );
    if( penv && penv != &environ ){
      size_t n = 0, bytes = 0;
      for( char **e = environ; e && *e; e++ ){ n++; bytes += strlen( *e ) + 1; }
      char **arr = (char **)malloc( sizeof(char *)*( n+1 ) + bytes );
      if( arr ){
        char *p = (char *)( arr + n + 1 );
        size_t k = 0;
        for( char **e = environ; e && *e && k < n; e++, k++ ){ const size_t l = strlen( *e ) + 1; memcpy( p, *e, l ); arr[k] = p; p += l; }
        arr[k] = NULL;
        if( getenv( "RKFD_SPEC_DEBUG" ) ) fprintf( stderr, "rkfd: hipRTC's namespace gets its own snapshot of the environment (%zu variables)\n", k );
        *penv = arr;      /* (the previous snapshot is deliberately not freed, see above) */
      }
    }
  }
  if( const char *dump = getenv( "RKFD_SPEC_DUMP" ) ){ FILE *f = fopen( dump, "w" ); if( f ){ fputs( src.c_str(), f ); fclose( f ); } }   /* diagnostic */
  hiprtcProgram prog;
  if( rtc->create( &prog, src.c_str(), "rkfd_step_kernel_spec.hip", rkfd_src_count, (const char **)rkfd_src_text, (const char **)rkfd_src_name ) != HIPRTC_SUCCESS ){ SETERR( "hiprtcCreateProgram failed" ); return -1; }
  const hiprtcResult r = rtc->compile( prog, nopts, opts );
  if( r != HIPRTC_SUCCESS ){
    size_t n = 0; rtc->logsize( prog, &n );
    std::string log( n ? n : 1, ' ' ); if( n ) rtc->log( prog, &log[0] );
    SETERR( "hipRTC: %s: %.400s", rtc->errstr( r ), log.c_str() );
    rtc->destroy( &prog );
    return -1;
  }
  size_t n = 0;
  rtc->codesize( prog, &n ); code.resize( n ); rtc->code( prog, code.data() );
  if( const char *dump = getenv( "RKFD_SPEC_DUMP_CODE" ) ){ FILE *f = fopen( dump, "wb" ); if( f ){ fwrite( code.data(), 1, n, f ); fclose( f ); } }   /* diagnostic */
  rtc->destroy( &prog );
  if( rtc->priv ) spec_to_store( key, code );      /* (only what the compiler this library was built with produced: the private namespace) */
  return 0;
}
extern "C" int rkfdSpecializeCompileW(const rkfdModel *m, int max_rigid, int ipw)
{
  rkfdDevModelHost h;
  char err[256];
  if( ipw != 1 && ipw != 2 ){ SETERR( "rkfdSpecializeCompileW: 1 or 2 instances per wavefront" ); return -1; }
  if( !m || rkfd_devmodel_build_w( m, max_rigid, ipw == 2 ? 4 : 8, &h, err, sizeof(err) ) < 0 ){ SETERR( "rkfdSpecializeCompile: %s", m ? err : "null model" ); return -1; }
  std::vector<char> code;
  const int r = spec_compile( h.dm, code, ipw );
  rkfd_devmodel_free( &h );
  return r < 0 ? -1 : (int)code.size();
}
extern "C" int rkfdSpecializeCompile(const rkfdModel *m, int max_rigid){ return rkfdSpecializeCompileW( m, max_rigid, 1 ); }
extern "C" int rkfdBatchSpecialize(rkfdBatch *b)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  if( b->spec_fn ) return 0;
  if( b->lds_bytes > 64*1024 ){ SETERR( "rkfdBatchSpecialize: worlds above 64 KiB of LDS per instance keep the generic kernel" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  std::vector<char> code;
  if( spec_compile( b->ipw == 2 ? b->host2.dm : b->dm, code, b->ipw == 2 ? 2 : 1 ) < 0 ) return -1;
  HIPCHK( hipModuleLoadData( &b->spec_mod, code.data() ), -1 );

  HIPCHK( hipModuleGetFunction( &b->spec_fn, b->spec_mod, "rkfd_step_kernel_spec" ), -1 );
  {
    /* the compiler behind hipRTC is whichever libamd_comgr the process loaded first; a framework that bundles an older
     * one (PyTorch does) gives a kernel that spills (437 VGPR spills, 3.9 M instead of 14 M steps/s on config 4): refuse it */
    int regs = -1, scratch = -1;
    hipFuncGetAttribute( &regs, HIP_FUNC_ATTRIBUTE_NUM_REGS, b->spec_fn );
    hipFuncGetAttribute( &scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, b->spec_fn );
    if( getenv( "RKFD_SPEC_DEBUG" ) ) fprintf( stderr, "rkfdBatchSpecialize: %d instance(s) per wavefront, %d VGPRs, %d bytes of scratch per lane, %zu bytes of LDS per instance\n", b->ipw == 2 ? 2 : 1, regs, scratch, b->ipw == 2 ? b->host2.lds_bytes : b->lds_bytes );
    if( scratch > ( b->dm.vol_np > 0 ? 512 : 160 ) ){      /* (the Volume variant is built for two waves per SIMD and spills a few registers on purpose; worlds with two moving
                                                             * contact sides sit at the 168-register limit and spill a handful - tools/spec_resources.py; the wrong compiler: 912 B) */
      (void)hipModuleUnload( b->spec_mod ); b->spec_mod = NULL; b->spec_fn = NULL;
      SETERR( "rkfdBatchSpecialize: the compiler hipRTC resolved to in this process produced a spilling kernel (%d VGPRs, %d bytes of scratch per lane); "
              "point RKFD_ROCM_LIBDIR at the ROCm libraries this library was built with; the generic kernel stays in use", regs, scratch );
      return -1;
    }
  }
  return 0;
}

/* Two instances per wavefront (RKFD_W = 2, rkfd_devmodel.h): takes effect in the world-specific kernel - call before
 * rkfdBatchSpecialize.  Builds the second device model (sweep schedule with four links per iteration); fails with a message
 * when the world does not fit 32 lanes per instance. */
extern "C" int rkfdBatchSetInstancesPerWave(rkfdBatch *b, int ipw)
{
  if( !b || ( ipw != 1 && ipw != 2 ) ){ SETERR( "rkfdBatchSetInstancesPerWave: 1 or 2" ); return -1; }
  if( b->spec_fn ){ SETERR( "rkfdBatchSetInstancesPerWave: call it before rkfdBatchSpecialize" ); return -1; }
  if( ipw == 1 ){ b->ipw = 1; return 0; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( !b->dblob2 ){
    char err[256];
    if( rkfd_devmodel_build_w( b->model_for_w2, b->host.dm.maxrg, 4, &b->host2, err, sizeof(err) ) < 0 ){ SETERR( "rkfdBatchSetInstancesPerWave: %s", err ); return -1; }
    if( b->host2.lds_bytes*2 + (size_t)b->host2.dm.lds_shared > 160*1024 ){ SETERR( "rkfdBatchSetInstancesPerWave: two instances need %zu bytes of LDS (> 160 KiB)", 2*b->host2.lds_bytes ); rkfd_devmodel_free( &b->host2 ); return -1; }
    HIPCHK( hipMalloc( &b->dblob2, b->host2.bytes ), -1 );
    HIPCHK( hipMemcpy( b->dblob2, b->host2.blob, b->host2.bytes, hipMemcpyHostToDevice ), -1 );
    b->dm2 = b->host2.dm;
    rkfd_devmodel_rebase( &b->dm2, b->host2.blob, b->dblob2 );
  }
  b->ipw = 2;
  return 0;
}
extern "C" int rkfdBatchInstancesPerWave(const rkfdBatch *b){ return b ? ( b->ipw == 2 && b->spec_fn ? 2 : 1 ) : -1; }


/* rkfdBatchTuneInstancesPerWave: MEASURE which of the two mappings is faster for this world on this device at this batch
 * size, and keep it.  The two give the same results to the last bit (tests/test_gpu_edge.py: two_instances_per_wavefront), so
 * the choice changes the time of a step and nothing else.  The batch must hold a state to step from (rkfdBatchSetState +
 * rkfdBatchUpdateInit); the state is put back as it was, an earlier rkfdBatchSnapshot is left alone.  Compiles both kernels
 * (or takes them from the ahead-of-time store).  Returns the chosen count (1 or 2), -1 on error; ms[2] (may be NULL) gets
 * the milliseconds of nsteps steps under either (ms[1] < 0: the world is not eligible for two - rkfdLastError says why).
 * Which wins is a property of the world: two instances per wavefront halve the instructions issued per step, but not the
 * LDS an instance holds, so they win where the step is bound by instruction issue (config 1b, 2, 3: +20 %, +16 %, +5 %) and
 * lose where it is bound by how many instances the LDS lets a CU hold (config 4: -4 %); profiles/r03_ipw_ab.txt. */
static int tune_time(rkfdBatch *b, int nsteps, double *ms)
{
  hipEvent_t e0, e1;
  HIPCHK( hipEventCreate( &e0 ), -1 ); HIPCHK( hipEventCreate( &e1 ), -1 );
  int r = 0;
  float best = -1;
  for( int rep=0; rep<3 && r == 0; rep++ ){      /* first pass: warm-up (code object upload, clocks); then the faster of two */
    if( rkfdBatchRestore( b, NULL ) < 0 || rkfdBatchJoin( b, NULL ) < 0 ){ r = -1; break; }
    if( hipEventRecord( e0, NULL ) != hipSuccess || rkfdBatchUpdate( b, nsteps, NULL ) < 0 || rkfdBatchJoin( b, NULL ) < 0
        || hipEventRecord( e1, NULL ) != hipSuccess || hipEventSynchronize( e1 ) != hipSuccess ){ r = -1; break; }
    float t = 0;
    (void)hipEventElapsedTime( &t, e0, e1 );
    if( rep > 0 && ( best < 0 || t < best ) ) best = t;
  }
  (void)hipEventDestroy( e0 ); (void)hipEventDestroy( e1 );
  *ms = best;
  return r;
}
extern "C" int rkfdBatchTuneInstancesPerWave(rkfdBatch *b, int nsteps, double *ms)
{
  if( !b || nsteps < 1 ){ SETERR( "rkfdBatchTuneInstancesPerWave: bad arguments" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  double t[2] = { -1, -1 };
  if( ms ){ ms[0] = ms[1] = -1; }
  /* the caller's snapshot steps aside for the one the measurement restores from */
  const rkfdDevState keep = b->snap;
  const int had = b->has_snap;
  b->has_snap = 0; memset( &b->snap, 0, sizeof(b->snap) );
  int r = rkfdBatchSnapshot( b );
  int chosen = -1;
  if( r == 0 ){
    /* (a) what the batch has now, brought to one instance per wavefront */
    if( b->spec_fn && b->ipw == 2 ){ (void)hipModuleUnload( b->spec_mod ); b->spec_mod = NULL; b->spec_fn = NULL; }
    b->ipw = 1;
    r = rkfdBatchSpecialize( b );
    if( r == 0 ) r = tune_time( b, nsteps, &t[0] );
  }
  if( r == 0 ){
    /* (b) two: set the first kernel aside, build the second */
    hipModule_t mod1 = b->spec_mod; hipFunction_t fn1 = b->spec_fn;
    b->spec_mod = NULL; b->spec_fn = NULL;
    int two = rkfdBatchSetInstancesPerWave( b, 2 );
    if( two == 0 ) two = rkfdBatchSpecialize( b );
    if( two == 0 ) r = tune_time( b, nsteps, &t[1] );
    if( two == 0 && r == 0 && t[1] < t[0] ){ (void)hipModuleUnload( mod1 ); chosen = 2; }
    else{
      if( b->spec_mod ) (void)hipModuleUnload( b->spec_mod );
      b->spec_mod = mod1; b->spec_fn = fn1; b->ipw = 1;
      if( r == 0 ) chosen = 1;
    }
  }
  /* the state as it was, and the caller's snapshot back in place */
  if( b->has_snap ){
    if( rkfdBatchRestore( b, NULL ) < 0 || rkfdBatchJoin( b, NULL ) < 0 || hipDeviceSynchronize() != hipSuccess ) chosen = -1;
    (void)hipFree( b->snap.dis ); (void)hipFree( b->snap.vel ); (void)hipFree( b->snap.acc );
    (void)hipFree( b->snap.piv_type ); (void)hipFree( b->snap.piv_prev ); (void)hipFree( b->snap.brk );
    (void)hipFree( b->snap.cv_active ); (void)hipFree( b->snap.cv_type ); (void)hipFree( b->snap.cv_ref ); (void)hipFree( b->snap.cv_f );
  }
  b->snap = keep; b->has_snap = had;
  if( ms ){ ms[0] = t[0]; ms[1] = t[1]; }
  return chosen;
}

extern "C" int rkfdBatchSetStepsPerLaunch(rkfdBatch *b, int n)
{
  if( !b || n < 1 ){ SETERR( "rkfdBatchSetStepsPerLaunch: n >= 1" ); return -1; }
  b->steps_per_launch = n;
  return 0;
}
extern "C" int rkfdBatchSetSplit(rkfdBatch *b, int nsplit)
{
  if( !b || nsplit < 1 || nsplit > RKFD_MAX_SPLIT ){ SETERR( "rkfdBatchSetSplit: 1 <= nsplit <= %d", RKFD_MAX_SPLIT ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  if( nsplit > 1 && !b->fork ){
    HIPCHK( hipEventCreateWithFlags( &b->fork, hipEventDisableTiming ), -1 );
    for( int k=0; k<RKFD_MAX_SPLIT; k++ ){
      HIPCHK( hipStreamCreateWithFlags( &b->sub[k], hipStreamNonBlocking ), -1 );
      HIPCHK( hipEventCreateWithFlags( &b->done[k], hipEventDisableTiming ), -1 );
    }
  }
  b->nsplit = nsplit;
  return 0;
}
extern "C" int rkfdBatchJoin(rkfdBatch *b, void *stream)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  return join_streams( b, (hipStream_t)stream );
}
extern "C" int rkfdBatchTimeLaunches(rkfdBatch *b, int on)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  b->tused = 0;
  b->timing = on ? 1 : 0;
  if( on && b->tev->empty() ){ hipEvent_t e0, e1; (void)timing_pair( b, &e0, &e1 ); b->tused = 0; }     /* pre-create the first block */
  return 0;
}
extern "C" int rkfdBatchLaunchTiming(rkfdBatch *b, int *launches, double *total_ms)
{
  if( !b || !launches || !total_ms ){ SETERR( "rkfdBatchLaunchTiming: bad arguments" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  HIPCHK( hipDeviceSynchronize(), -1 );
  *launches = (int)( b->tused/2 ); *total_ms = 0;
  for( size_t i=0; i+1<b->tused; i+=2 ){
    float ms = 0;
    HIPCHK( hipEventElapsedTime( &ms, (*b->tev)[i], (*b->tev)[i+1] ), -1 );
    *total_ms += ms;
  }
  return 0;
}
extern "C" int rkfdBatchUpdateInit(rkfdBatch *b, void *stream){ return launch( b, 1, 0, stream ); }
extern "C" int rkfdBatchUpdate(rkfdBatch *b, int nsteps, void *stream)
{
  if( nsteps < 1 ){ SETERR( "rkfdBatchUpdate: nsteps must be >= 1" ); return -1; }
  return launch( b, 0, nsteps, stream );
}
extern "C" int rkfdBatchEval(rkfdBatch *b, int doUpRef, void *stream){ return launch( b, doUpRef ? 1 : 2, 0, stream ); }

/* diagnostic: nsteps x rkFDUpdate with in-kernel phase stamps; out[batch][8] shader-clock cycles of
 * {kinematics, collision+penalty, sweep 2, sweep 3, MLCP, tail, unused, whole launch}.  Synchronous. */
extern "C" int rkfdBatchProfile(rkfdBatch *b, int nsteps, unsigned long long *out)
{
  if( !b || !out ){ SETERR( "rkfdBatchProfile: bad arguments" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  unsigned long long *d = NULL;
  const size_t n = sizeof(unsigned long long)*(size_t)b->batch*RKFD_NPROF;
  HIPCHK( hipMalloc( (void **)&d, n ), -1 );
  HIPCHK( hipMemset( d, 0, n ), -1 );
  b->st.prof = d;
  int r = launch( b, 0, nsteps, NULL );
  b->st.prof = NULL;
  if( r == 0 ){
    hipError_t e = hipMemcpy( out, d, n, hipMemcpyDeviceToHost );
    if( e != hipSuccess ){ SETERR( "hipMemcpy failed: %s", hipGetErrorString( e ) ); r = -1; }
  }
  (void)hipFree( d );
  return r;
}

extern "C" int rkfdBatchStatus(rkfdBatch *b, void *stream)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( join_streams( b, (hipStream_t)stream ) < 0 ) return -1;
  HIPCHK( hipStreamSynchronize( (hipStream_t)stream ), -1 );
  int e = 0;
  HIPCHK( hipMemcpy( &e, b->d_err, sizeof(int), hipMemcpyDeviceToHost ), -1 );
  if( e == 1 ){
    static const char *const sname[] = { "Vert", "MLCP", "Volume" };
    SETERR( "a rigid contact occurred but no rigid solver is set up on the device for this world: plugin %s, max_rigid = %d "
            "(a batch created with max_rigid = 0 carries no rigid path; so does a plugin / world combination the device does not solve)",
            ( b->dm.solver >= 0 && b->dm.solver <= 2 ) ? sname[b->dm.solver] : "?", b->dm.maxrg );
  }
  if( e == 2 && b->dm.vol_np > 0 )
    SETERR( "contact capacity exceeded in at least one instance: more rigid pairs in volumetric contact than %d, more than %d contact-plane "
            "conditions in one pair, or more elastic contact vertices than the %d active-contact slots; what was beyond the capacity was dropped",
            b->dm.vol_np, b->dm.vol_ncp, b->dm.maxact );
  else if( e == 2 )
    SETERR( "contact capacity exceeded in at least one instance: more rigid contact vertices than max_rigid (%d), or more "
            "rigid + elastic contact vertices than the %d active-contact slots; contacts beyond the capacity were dropped", b->dm.maxrg, b->dm.maxact );
  if( e == 3 ) SETERR( "the %s plugin's QP ran out of iterations (256) or of basis history (64) in at least one instance", b->dm.vol_np > 0 ? "Volume" : "Vert" );
  if( e == 4 ) SETERR( "Volume plugin: a rigid pair the device cannot clip (a shape that is not convex, or more than 64 faces together) came into "
                       "contact in at least one instance; it was left without a contact force" );
  if( e != 0 ){
    /* the condition is reported once: the flag is cleared, so a later status describes what happened after this call */
    const int zero = 0;
    HIPCHK( hipMemcpy( b->d_err, &zero, sizeof(int), hipMemcpyHostToDevice ), -1 );
  }
  return e;
}

/* ---- contact statistics, snapshot / restore ---------------------------------------------------------------- */
extern "C" int rkfdBatchContactStats(rkfdBatch *b, int reset, double *mean_rigid, double *mean_elastic, long long *instance_steps)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  HIPCHK( hipDeviceSynchronize(), -1 );
  std::vector<unsigned int> h( (size_t)b->batch*4 );
  HIPCHK( hipMemcpy( h.data(), b->st.stat, sizeof(unsigned int)*h.size(), hipMemcpyDeviceToHost ), -1 );
  unsigned long long rg = 0, el = 0, n = 0;
  for( int i=0; i<b->batch; i++ ){ rg += h[4*(size_t)i]; el += h[4*(size_t)i+1]; n += h[4*(size_t)i+2]; }
  if( mean_rigid ) *mean_rigid = n ? (double)rg/(double)n : 0.0;
  if( mean_elastic ) *mean_elastic = n ? (double)el/(double)n : 0.0;
  if( instance_steps ) *instance_steps = (long long)n;
  if( reset ) HIPCHK( hipMemset( b->st.stat, 0, sizeof(unsigned int)*h.size() ), -1 );
  return 0;
}

extern "C" int rkfdBatchSnapshot(rkfdBatch *b)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( sync_streams( b ) < 0 ) return -1;
  HIPCHK( hipDeviceSynchronize(), -1 );
  const size_t B = b->batch, ND = b->ndof, NL = b->nlink, NC = b->ncand;
  if( !b->has_snap ){
    int bad = 0;
    bad |= dalloc( &b->snap.dis, B*ND ); bad |= dalloc( &b->snap.vel, B*ND ); bad |= dalloc( &b->snap.acc, B*ND );
    bad |= dalloc( &b->snap.piv_type, B*NL ); bad |= dalloc( &b->snap.piv_prev, B*NL ); bad |= dalloc( &b->snap.brk, B*NL );
    bad |= dalloc( &b->snap.cv_active, B*NC ); bad |= dalloc( &b->snap.cv_type, B*NC );
    bad |= dalloc( &b->snap.cv_ref, B*NC*3 ); bad |= dalloc( &b->snap.cv_f, B*NC*3 );
    if( bad ) return -1;
    b->snap.batch = b->batch;
    b->has_snap = 1;
  }
#define SNAPCP(f, n) HIPCHK( hipMemcpy( b->snap.f, b->st.f, (n), hipMemcpyDeviceToDevice ), -1 )
  SNAPCP( dis, sizeof(double)*B*ND ); SNAPCP( vel, sizeof(double)*B*ND ); SNAPCP( acc, sizeof(double)*B*ND );
  SNAPCP( piv_type, sizeof(int)*B*NL ); SNAPCP( piv_prev, sizeof(double)*B*NL ); SNAPCP( brk, sizeof(int)*B*NL );
  SNAPCP( cv_active, sizeof(int)*B*NC ); SNAPCP( cv_type, sizeof(int)*B*NC );
  SNAPCP( cv_ref, sizeof(double)*B*NC*3 ); SNAPCP( cv_f, sizeof(double)*B*NC*3 );
#undef SNAPCP
  return 0;
}

extern "C" int rkfdBatchRestore(rkfdBatch *b, void *stream)
{
  if( !b ){ SETERR( "null batch" ); return -1; }
  if( !b->has_snap ){ SETERR( "rkfdBatchRestore: no snapshot has been taken (rkfdBatchSnapshot)" ); return -1; }
  HIPCHK( hipSetDevice( b->device ), -1 );
  if( b->nsplit <= 1 ){
    hipLaunchKernelGGL( rkfd_restore_kernel, dim3( b->batch ), dim3( RKFD_WAVE ), 0, (hipStream_t)stream, b->st, b->snap, 0, b->ndof, b->nlink, b->ncand );
    HIPCHK( hipGetLastError(), -1 );
    return 0;
  }
  /* split launches: every part restores itself on its own stream, in order with its steps */
  HIPCHK( hipEventRecord( b->fork, (hipStream_t)stream ), -1 );
  for( int k=0; k<b->nsplit; k++ ){
    HIPCHK( hipStreamWaitEvent( b->sub[k], b->fork, 0 ), -1 );
    const int lo = (int)( (long long)b->batch*k/b->nsplit ), hi = (int)( (long long)b->batch*( k+1 )/b->nsplit );
    if( hi > lo ){
      hipLaunchKernelGGL( rkfd_restore_kernel, dim3( hi-lo ), dim3( RKFD_WAVE ), 0, b->sub[k], b->st, b->snap, lo, b->ndof, b->nlink, b->ncand );
      HIPCHK( hipGetLastError(), -1 );
    }
    HIPCHK( hipEventRecord( b->done[k], b->sub[k] ), -1 );
  }
  b->pending = 1;
  return 0;
}

#include "rkfd_capi_node.inc"      /* the node level: one host thread + stream per device, one RCCL all-gather of final states */
