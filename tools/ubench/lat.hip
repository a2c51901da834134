// micro-benchmarks of dependent-chain latencies on one wave (diagnostic; not part of the product)
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
__global__ void k_fma(double *out, long long *cyc, double a, double b){
  double x = out[threadIdx.x];
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 16
  for( int i=0; i<N; i++ ) x = fma( x, a, b );
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x; if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_fma_indep(double *out, long long *cyc, double a, double b){
  double x0 = out[threadIdx.x], x1 = x0+1, x2 = x0+2, x3 = x0+3;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for( int i=0; i<N/4; i++ ){ x0 = fma( x0, a, b ); x1 = fma( x1, a, b ); x2 = fma( x2, a, b ); x3 = fma( x3, a, b ); }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x0+x1+x2+x3; if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_readlane(double *out, long long *cyc, double a, int lane){
  double x = out[threadIdx.x];
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for( int i=0; i<N; i++ ){
    int lo = __builtin_amdgcn_readlane( __double2loint( x ), lane ), hi = __builtin_amdgcn_readlane( __double2hiint( x ), lane );
    x = fma( __hiloint2double( hi, lo ), a, x );
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x; if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_lds(double *out, long long *cyc){
  __shared__ int idx[1024];
  for( int i=threadIdx.x; i<1024; i+=64 ) idx[i] = ( i*17 + 5 ) & 1023;
  __syncthreads();
  int p = threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for( int i=0; i<N; i++ ) p = idx[p];
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = p; if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_swz(double *out, long long *cyc, double a){
  double x = out[threadIdx.x];
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for( int i=0; i<N; i++ ){
    int lo = __builtin_amdgcn_ds_swizzle( __double2loint( x ), ( 3 << 5 ) | 0x18 ), hi = __builtin_amdgcn_ds_swizzle( __double2hiint( x ), ( 3 << 5 ) | 0x18 );
    x = fma( __hiloint2double( hi, lo ), a, x );
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x; if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_dpp(double *out, long long *cyc, double a){
  double x = out[threadIdx.x];
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for( int i=0; i<N; i++ ){
    int lo = __builtin_amdgcn_update_dpp( 0, __double2loint( x ), 0xB1, 0xF, 0xF, true ), hi = __builtin_amdgcn_update_dpp( 0, __double2hiint( x ), 0xB1, 0xF, 0xF, true );
    x = __hiloint2double( hi, lo ) + x;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x; if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_rcp(double *out, long long *cyc){
  double x = out[threadIdx.x] + 1.5;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 8
  for( int i=0; i<N; i++ ) x = __builtin_amdgcn_rcp( x ) + 1.0;
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x; if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_gload(double *out, long long *cyc, const int *tab){
  int p = threadIdx.x;
  long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 4
  for( int i=0; i<512; i++ ) p = tab[p];
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = p; if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}

// frequency of the s_memtime counter: spin for a fixed number of ticks, time the launch with HIP events
__global__ void k_spin(long long *cyc, long long ticks){
  long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime(), t1;
  do { t1 = __builtin_amdgcn_s_memtime(); } while( t1 - t0 < ticks );
  long long r1 = __builtin_amdgcn_s_memrealtime();
  if( threadIdx.x == 0 ){ cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
// dependent fp64 FMA chain, long enough to time with HIP events: wall time per instruction
__global__ void k_fma_long(double *out, long long *cyc, double a, double b, int n){
  double x = out[threadIdx.x];
  long long t0 = __builtin_amdgcn_s_memtime();
  for( int j=0; j<n; j++ ){
#pragma unroll 16
    for( int i=0; i<N; i++ ) x = fma( x, a, b );
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = x; if( threadIdx.x == 0 ) cyc[blockIdx.x] = t1 - t0;
}
int main(){
  double *out; long long *cyc; int *tab;
  hipMalloc( &out, 64*8 ); hipMemset( out, 0, 64*8 ); hipMalloc( &cyc, 8*4096 ); hipMalloc( &tab, 4096*4 );
  int h[4096]; for( int i=0; i<4096; i++ ) h[i] = ( i*33 + 7 ) & 4095;
  hipMemcpy( tab, h, sizeof(h), hipMemcpyHostToDevice );
  long long c;
#define RUN(name, per, ...) for( int r=0; r<2; r++ ){ hipLaunchKernelGGL( name, dim3(1), dim3(64), 0, 0, __VA_ARGS__ ); hipDeviceSynchronize(); } \
  hipMemcpy( &c, cyc, 8, hipMemcpyDeviceToHost ); printf( "%-14s %7.1f ticks per op\n", #name, (double)c/(per) );
  RUN( k_fma, N, out, cyc, 0.999, 0.001 )
  RUN( k_fma_indep, N, out, cyc, 0.999, 0.001 )
  RUN( k_readlane, N, out, cyc, 0.5, 3 )
  RUN( k_lds, N, out, cyc )
  RUN( k_swz, N, out, cyc, 0.5 )
  RUN( k_dpp, N, out, cyc, 0.5 )
  RUN( k_rcp, N, out, cyc )
  RUN( k_gload, 512, out, cyc, tab )
  // clock rate of s_memtime: time a long kernel with events
  hipEvent_t e0, e1; hipEventCreate( &e0 ); hipEventCreate( &e1 );
  hipEventRecord( e0 ); for( int r=0; r<50; r++ ) hipLaunchKernelGGL( k_fma, dim3(1), dim3(64), 0, 0, out, cyc, 0.999, 0.001 ); hipEventRecord( e1 ); hipEventSynchronize( e1 );
  float ms; hipEventElapsedTime( &ms, e0, e1 ); hipMemcpy( &c, cyc, 8, hipMemcpyDeviceToHost );
  printf( "k_fma: %lld ticks per launch, %.3f ms per launch (incl. launch gaps)\n", c, ms/50 );
  {
    long long tk[2]; float msx;
    hipEventRecord( e0 ); hipLaunchKernelGGL( k_spin, dim3(1), dim3(64), 0, 0, cyc, 1ll<<28 ); hipEventRecord( e1 ); hipEventSynchronize( e1 );
    hipEventElapsedTime( &msx, e0, e1 ); hipMemcpy( tk, cyc, 16, hipMemcpyDeviceToHost );
    printf( "k_spin: %lld s_memtime ticks, %lld s_memrealtime ticks in %.3f ms -> s_memtime %.1f MHz, s_memrealtime %.1f MHz\n", tk[0], tk[1], msx, tk[0]/(msx*1e3), tk[1]/(msx*1e3) );
    hipEventRecord( e0 ); hipLaunchKernelGGL( k_fma_long, dim3(1), dim3(64), 0, 0, out, cyc, 0.999, 0.001, 2048 ); hipEventRecord( e1 ); hipEventSynchronize( e1 );
    hipEventElapsedTime( &msx, e0, e1 ); hipMemcpy( tk, cyc, 8, hipMemcpyDeviceToHost );
    printf( "k_fma_long (1 wave): %.3f ns and %.2f ticks per dependent v_fma_f64\n", msx*1e6/( 2048.0*N ), (double)tk[0]/( 2048.0*N ) );
    /* the same with every SIMD busy: 4 waves per CU on all CUs */
    hipEventRecord( e0 ); hipLaunchKernelGGL( k_fma_long, dim3(1024), dim3(64), 0, 0, out, cyc, 0.999, 0.001, 2048 ); hipEventRecord( e1 ); hipEventSynchronize( e1 );
    hipEventElapsedTime( &msx, e0, e1 ); hipMemcpy( tk, cyc, 8, hipMemcpyDeviceToHost );
    printf( "k_fma_long (1024 waves): %.3f ns and %.2f ticks per dependent v_fma_f64\n", msx*1e6/( 2048.0*N ), (double)tk[0]/( 2048.0*N ) );
    hipEventRecord( e0 ); hipLaunchKernelGGL( k_fma_long, dim3(2048), dim3(64), 0, 0, out, cyc, 0.999, 0.001, 2048 ); hipEventRecord( e1 ); hipEventSynchronize( e1 );
    hipEventElapsedTime( &msx, e0, e1 ); hipMemcpy( tk, cyc, 8, hipMemcpyDeviceToHost );
    printf( "k_fma_long (2048 waves, 2 per SIMD): %.3f ns per v_fma_f64 per wave -> %.3f ns per SIMD instruction\n", msx*1e6/( 2048.0*N ), msx*1e6/( 2048.0*N )/2 );
  }
  return 0;
}
