// micro-benchmark of the projected Gauss-Seidel variants of csrc/device/rkfd_dev_mlcp.h in isolation (diagnostic; not part
// of the product): one wavefront per workgroup solves a synthetic nc-contact problem `reps` times; prints shader-clock
// cycles per solve (10 sweeps) for 1 workgroup per CU and for `res` workgroups per CU (the step kernel's residency).
// build: hipcc --offload-arch=gfx950 -O3 -Iinclude -Iroki-fd_amd/csrc -o tools/ubench/pgs tools/ubench/pgs.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "rkfd_device.h"

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
  for( int nc : { 3, 4, 8, 12, 16 } ){
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
    for( int v=0; v<3; v++ ){
      if( v == 2 && nc > 4 ) continue;
      for( int full=0; full<2; full++ ){
        const int nb = full ? nblk : ncu;
        void (*k)(const double *, const double *, double *, long long *, int, int, int, int) = v == 0 ? k_pgs<0> : ( v == 1 ? k_pgs<1> : k_pgs<2> );
        hipLaunchKernelGGL( k, dim3( nb ), dim3( 64 ), ldsb, 0, dA, dB, dout, dcyc, nc, 16, 20, 0 );
        hipDeviceSynchronize();
        std::vector<long long> c( nb ); std::vector<double> o( 64 );
        hipMemcpy( c.data(), dcyc, sizeof(long long)*nb, hipMemcpyDeviceToHost ); hipMemcpy( o.data(), dout, sizeof(double)*64, hipMemcpyDeviceToHost );
        double mean = 0; for( auto x : c ) mean += x; mean /= nb;
        if( v == 0 && !full ) ref = o;
        double dev = 0; for( int i=0; i<nc; i++ ) dev = fmax( dev, fabs( o[i] - ref[i] ) );
        printf( "nc %2d  %-9s %s per CU: %8.0f cycles per solve = %6.1f per update   (result deviates from the general loop by %.1e)\n", nc,
                v == 0 ? "general" : ( v == 1 ? "dpp" : "registers" ), full ? "all " : "one ", mean, mean/( 20.0*nc ), dev );
      }
    }
    hipFree( dA ); hipFree( dB ); hipFree( dout ); hipFree( dcyc );
  }
  return 0;
}
