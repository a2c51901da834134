// how many one-wave workgroups with a given dynamic LDS size are really resident on a CU at once
// (diagnostic; the HIP occupancy query and the hardware's LDS allocation granularity do not always agree)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(64, 3) k_res(int *cnt, int *mx, long long ticks){
  extern __shared__ char lds[];
  const unsigned hw = __builtin_amdgcn_s_getreg( ( 31 << 11 ) | 4 );     // HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg( ( 31 << 11 ) | 20 );   // XCC_ID
  const unsigned cu = ( hw >> 8 ) & 0xF, sh = ( hw >> 12 ) & 1, se = ( hw >> 13 ) & 7;
  const int key = (int)( ( ( xcc & 0xF ) << 8 ) | ( se << 5 ) | ( sh << 4 ) | cu );
  if( threadIdx.x == 0 ){
    lds[0] = 1;
    const int now = atomicAdd( &cnt[key], 1 ) + 1;
    atomicMax( &mx[key], now );
    long long t0 = __builtin_amdgcn_s_memtime();
    while( __builtin_amdgcn_s_memtime() - t0 < ticks ){}
    atomicSub( &cnt[key], 1 );
  }
}
int main(int argc, char **argv){
  int *cnt, *mx; hipMalloc( &cnt, 4096*4 ); hipMalloc( &mx, 4096*4 );
  static int h[4096];
  for( int a=1; a<argc; a++ ){
    const int bytes = atoi( argv[a] );
    hipMemset( cnt, 0, 4096*4 ); hipMemset( mx, 0, 4096*4 );
    hipFuncSetAttribute( (const void *)k_res, hipFuncAttributeMaxDynamicSharedMemorySize, bytes );
    hipLaunchKernelGGL( k_res, dim3(16384), dim3(64), bytes, 0, cnt, mx, 200000ll );
    hipDeviceSynchronize();
    hipMemcpy( h, mx, 4096*4, hipMemcpyDeviceToHost );
    int ncu = 0, lo = 1<<30, hi = 0; long sum = 0;
    for( int i=0; i<4096; i++ ) if( h[i] ){ ncu++; sum += h[i]; if( h[i] < lo ) lo = h[i]; if( h[i] > hi ) hi = h[i]; }
    int q = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor( &q, (const void *)k_res, 64, bytes );
    printf( "LDS %6d B: %d CUs seen, resident workgroups per CU min %d max %d mean %.2f (occupancy query says %d)\n", bytes, ncu, lo, hi, (double)sum/ncu, q );
  }
  return 0;
}
