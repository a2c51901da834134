// does a VALU instruction cost fewer cycles when only the first 16 (or 32) lanes of the wavefront are enabled?
// (diagnostic; not part of the product)  12 waves per CU, 8 independent fp64 FMA chains / 32-bit ops per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 2048
template<int ACTIVE, int KIND> __global__ void __launch_bounds__(64) k(double *out, long long *cyc, double a, double b)
{
  double x[8];
  int y[8];
  for( int i=0; i<8; i++ ){ x[i] = out[threadIdx.x] + i; y[i] = threadIdx.x + i; }
  long long t0 = __builtin_amdgcn_s_memtime();
  if( (int)threadIdx.x < ACTIVE ){
    for( int i=0; i<N; i++ ){
#pragma unroll
      for( int j=0; j<8; j++ ){
        if( KIND == 0 ) x[j] = fma( x[j], a, b );
        else y[j] = ( y[j]*3 ) ^ (int)threadIdx.x;
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0; for( int i=0; i<8; i++ ) s += x[i] + y[i];
  out[blockIdx.x*64+threadIdx.x] = s;
  if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}
template<int ACTIVE, int KIND> void run(const char *name, int nblk, double *out, long long *cyc)
{
  hipLaunchKernelGGL( ( k<ACTIVE, KIND> ), dim3( nblk ), dim3( 64 ), 0, 0, out, cyc, 0.999, 0.001 );
  hipDeviceSynchronize();
  long long h[4096]; hipMemcpy( h, cyc, sizeof(long long)*nblk, hipMemcpyDeviceToHost );
  double m = 0; for( int i=0; i<nblk; i++ ) m += h[i]; m /= nblk;
  printf( "%-28s %d blocks: %8.0f cycles per wave = %.2f per instruction per wave\n", name, nblk, m, m/( 8.0*N ) );
}
int main()
{
  hipDeviceProp_t p; hipGetDeviceProperties( &p, 0 );
  const int ncu = p.multiProcessorCount;
  double *out; long long *cyc; hipMalloc( &out, sizeof(double)*64*4096 ); hipMalloc( &cyc, sizeof(long long)*4096 );
  for( int per : { 4, 12 } ){
    const int nb = ncu*per;
    printf( "%d waves per CU\n", per );
    run<64, 0>( "fma f64, 64 lanes", nb, out, cyc ); run<32, 0>( "fma f64, lanes 0-31", nb, out, cyc ); run<16, 0>( "fma f64, lanes 0-15", nb, out, cyc ); run<8, 0>( "fma f64, lanes 0-7", nb, out, cyc );
    run<64, 1>( "int mul+xor, 64 lanes", nb, out, cyc ); run<16, 1>( "int mul+xor, lanes 0-15", nb, out, cyc );
  }
  return 0;
}
