// micro-benchmark of the projected Gauss-Seidel variants of csrc/device/rkfd_dev_mlcp.h in isolation (diagnostic; not part
// of the product): one wavefront per workgroup solves a synthetic nc-contact problem `reps` times; prints shader-clock
// cycles per solve (10 sweeps) for 1 workgroup per CU and for `res` workgroups per CU (the step kernel's residency).
// build: hipcc --offload-arch=gfx950 -O3 -Iinclude -Iroki-fd_amd/csrc -o tools/ubench/pgs tools/ubench/pgs.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "rkfd_device.h"

template<int C> __device__ __forceinline__ void mov_lane(double &dst, double src)
{
  unsigned long long keep;
  asm volatile( "s_mov_b64 %1, exec\n\ts_mov_b64 exec, %3\n\tv_mov_b64 %0, %2\n\ts_mov_b64 exec, %1" : "+v"(dst), "=&s"(keep) : "v"(src), "s"( 1ull << C ) );
}
// experiment: the DPP scheme for a literal contact count, no per-update guards.  SL: 0 = wave-uniform branch on the sliding test,
// 1 = branch-free (reciprocal always evaluated, selected)
template<int NC, int SL> __device__ __forceinline__ void pgs_static(const double *MA, int r0, int ld, int lane, double mu, double in_, double i1, double i2,
                                                                   double &rn, double &r1, double &r2, double &fn, double &f1, double &f2)
{
  for( int it=0; it<10; it++ ){
    {
      double a0[NC], a1[NC], a2[NC];
#pragma unroll
      for( int c=0; c<NC; c++ ){ a0[c] = MA[r0*ld+3*c]; a1[c] = MA[( r0+1 )*ld+3*c]; a2[c] = MA[( r0+2 )*ld+3*c]; }
#define NUPD(c) { double ff = fn - rn*in_; if( ff < RKFD_DEV_TOL ) ff = 0.0; const double dl = ff - fn; if( SL >= 2 ) mov_lane<c>( fn, ff ); else if( lane == c ) fn = ff; \
        ROWBC_FMAC( c, rn, dl, a0[c] ); ROWBC_FMAC( c, r1, dl, a1[c] ); ROWBC_FMAC( c, r2, dl, a2[c] ); }
      NUPD(0) NUPD(1) NUPD(2) NUPD(3) if( NC > 4 ){ NUPD(4) NUPD(5) NUPD(6) NUPD(7) }
#undef NUPD
    }
    double fs = mu*fn; fs = fs*fs;
#define TUPD(c) { const double a0 = MA[r0*ld+3*c+1], a1 = MA[( r0+1 )*ld+3*c+1], a2 = MA[( r0+2 )*ld+3*c+1], b0 = MA[r0*ld+3*c+2], b1 = MA[( r0+1 )*ld+3*c+2], b2 = MA[( r0+2 )*ld+3*c+2]; \
      const double ff0 = f1 - r1*i1, ff1 = f2 - r2*i2; const double fnorm = ff0*ff0 + ff1*ff1; \
      const bool zero = fnorm < RKFD_DEV_TOL || fs < RKFD_DEV_TOL; double n1 = ff0, n2 = ff1; \
      if( SL == 3 ){ if( ( BALLOT( zero ) >> c ) & 1ull ){ n1 = 0.0; n2 = 0.0; } } else { n1 = zero ? 0.0 : ff0; n2 = zero ? 0.0 : ff1; } \
      if( SL == 0 || SL >= 2 ){ if( ( BALLOT( !zero && fnorm > fs ) >> c ) & 1ull ){ const double sc = fs*RKFD_RCP( fnorm ); n1 = ff0*sc; n2 = ff1*sc; } } \
      else { const double sc = fs*RKFD_RCP( fnorm ); const bool sl = !zero && fnorm > fs; n1 = sl ? ff0*sc : n1; n2 = sl ? ff1*sc : n2; } \
      const double d1 = n1 - f1, d2 = n2 - f2; if( SL >= 2 ){ mov_lane<c>( f1, n1 ); mov_lane<c>( f2, n2 ); } else if( lane == c ){ f1 = n1; f2 = n2; } \
      ROWBC_FMAC( c, rn, d2, b0 ); ROWBC_FMAC( c, r1, d2, b1 ); ROWBC_FMAC( c, r2, d2, b2 ); ROWBC_FMAC( c, rn, d1, a0 ); ROWBC_FMAC( c, r1, d1, a1 ); ROWBC_FMAC( c, r2, d1, a2 ); }
    TUPD(0) TUPD(1) TUPD(2) TUPD(3) if( NC > 4 ){ TUPD(4) TUPD(5) TUPD(6) TUPD(7) }
#undef TUPD
  }
}

template<int V> __global__ void __launch_bounds__(64, 3) k_pgs(const double *A, const double *B, double *out, long long *cyc, int nc, int maxrg, int reps, int ldspad)
{
  extern __shared__ __attribute__((aligned(16))) char lds[];
  double *MA = (double *)lds;
  const int lane = LANE();
  const int M = 3*nc, ld = M+1;
  for( int i=lane; i<M*M; i+=64 ) MA[( i/M )*ld + i%M] = A[i];
  SYNC();
  const bool on = lane < nc;
  const int r0 = on ? 3*lane : 0;
  double acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  for( int r=0; r<reps; r++ ){
    double rn = 0, r1 = 0, r2 = 0, fn = 0, f1 = 0, f2 = 0, in_ = 0, i1 = 0, i2 = 0, mu = 0;
    if( on ){
      rn = B[r0]; r1 = B[r0+1]; r2 = B[r0+2];
      in_ = 1.0/MA[r0*ld+r0]; i1 = 1.0/MA[( r0+1 )*ld+r0+1]; i2 = 1.0/MA[( r0+2 )*ld+r0+2]; mu = 0.5;
    }
    if( V == 0 ) rkfd_pgs_general<false>( MA, r0, ld, nc, 10, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    if( V == 1 ) rkfd_pgs_dpp<false>( MA, r0, ld, nc, maxrg, 10, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    if( V == 2 ) rkfd_pgs_registers<false>( MA, r0, ld, nc, 10, on, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    if( V == 3 ) pgs_static<8, 0>( MA, r0, ld, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    if( V == 4 ) pgs_static<8, 1>( MA, r0, ld, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    if( V == 5 ) pgs_static<8, 2>( MA, r0, ld, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    if( V == 6 ) pgs_static<8, 3>( MA, r0, ld, lane, mu, in_, i1, i2, rn, r1, r2, fn, f1, f2 );
    acc += fn + f1 + f2;
    asm volatile( "" : "+v"(acc) );
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x*64+lane] = acc;
  if( lane == 0 ) cyc[blockIdx.x] = ( t1 - t0 )/reps;
}

int main(int argc, char **argv)
{
  const int res = argc > 1 ? atoi( argv[1] ) : 11;
  hipDeviceProp_t p; hipGetDeviceProperties( &p, 0 );
  const int ncu = p.multiProcessorCount;
  for( int nc : { 3, 4, 8, 12, 16, 24 } ){
    const int M = 3*nc;
    // A = G G' + 1e-4 I with G M x 6 (a rigid body's contact matrix has rank 6), b < 0 so that forces are non-zero
    std::vector<double> G( M*6 ), A( M*M ), B( M );
    srand( 7 );
    for( auto &g : G ) g = rand()/(double)RAND_MAX - 0.5;
    for( int i=0; i<M; i++ ) for( int j=0; j<M; j++ ){ double s = i == j ? 1e-4 : 0; for( int k=0; k<6; k++ ) s += G[i*6+k]*G[j*6+k]; A[i*M+j] = s; }
    for( int i=0; i<M; i++ ) B[i] = i%3 == 0 ? -0.01 : 0.002*( rand()/(double)RAND_MAX - 0.5 );
    double *dA, *dB, *dout; long long *dcyc;
    const int nblk = ncu*res;
    hipMalloc( &dA, sizeof(double)*M*M ); hipMalloc( &dB, sizeof(double)*M ); hipMalloc( &dout, sizeof(double)*64*nblk ); hipMalloc( &dcyc, sizeof(long long)*nblk );
    hipMemcpy( dA, A.data(), sizeof(double)*M*M, hipMemcpyHostToDevice ); hipMemcpy( dB, B.data(), sizeof(double)*M, hipMemcpyHostToDevice );
    const size_t ldsb = 14064;         // the humanoid's footprint: 11 workgroups per CU
    std::vector<double> ref;
    for( int v=0; v<7; v++ ){
      if( v == 2 && nc > 4 ) continue;
      if( v >= 3 && nc != 8 ) continue;
      if( v == 1 && nc > 16 ) continue;
      for( int full=0; full<2; full++ ){
        const int nb = full ? nblk : ncu;
        void (*k)(const double *, const double *, double *, long long *, int, int, int, int) = v == 0 ? k_pgs<0> : ( v == 1 ? k_pgs<1> : ( v == 2 ? k_pgs<2> : ( v == 3 ? k_pgs<3> : ( v == 4 ? k_pgs<4> : ( v == 5 ? k_pgs<5> : k_pgs<6> ) ) ) ) );
        hipLaunchKernelGGL( k, dim3( nb ), dim3( 64 ), ldsb, 0, dA, dB, dout, dcyc, nc, 16, 20, 0 );
        hipDeviceSynchronize();
        std::vector<long long> c( nb ); std::vector<double> o( 64 );
        hipMemcpy( c.data(), dcyc, sizeof(long long)*nb, hipMemcpyDeviceToHost ); hipMemcpy( o.data(), dout, sizeof(double)*64, hipMemcpyDeviceToHost );
        double mean = 0; for( auto x : c ) mean += x; mean /= nb;
        if( v == 0 && !full ) ref = o;
        double dev = 0; for( int i=0; i<nc; i++ ) dev = fmax( dev, fabs( o[i] - ref[i] ) );
        printf( "nc %2d  %-9s %s per CU: %8.0f cycles per solve = %6.1f per update   (result deviates from the general loop by %.1e)\n", nc,
                v == 0 ? "general" : ( v == 1 ? "dpp" : ( v == 2 ? "registers" : ( v == 3 ? "static8" : ( v == 4 ? "static8-nb" : ( v == 5 ? "static8-mov" : "static8-mov-z" ) ) ) ) ), full ? "all " : "one ", mean, mean/( 20.0*nc ), dev );
      }
    }
    hipFree( dA ); hipFree( dB ); hipFree( dout ); hipFree( dcyc );
  }
  return 0;
}
