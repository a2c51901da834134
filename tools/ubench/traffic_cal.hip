// calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS kernel's access pattern (MI355X_MICROARCH.md: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern"): one wavefront per row reads ND
// doubles (8 B per lane, lanes < ND) of its row and writes ND doubles of another array, rows laid out instance-major like
// the step kernel's state.  Known bytes: rows x ND x 8 each way.  Run under  rocprofv3 --pmc FETCH_SIZE  /  --pmc WRITE_SIZE.
// (diagnostic; not part of the product)   usage: traffic_cal [rows] [nd]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void __launch_bounds__(64) rkfd_traffic_cal(const double *in, double *out, int nd)
{
  const size_t b = blockIdx.x;
  if( (int)threadIdx.x < nd ) out[b*nd + threadIdx.x] = in[b*nd + threadIdx.x] + 1.0;
}
int main(int argc, char **argv)
{
  const size_t rows = argc > 1 ? atol( argv[1] ) : ( 1u << 21 );
  const int nd = argc > 2 ? atoi( argv[2] ) : 30;
  double *in, *out;
  hipMalloc( &in, sizeof(double)*rows*nd ); hipMalloc( &out, sizeof(double)*rows*nd );
  hipMemset( in, 0, sizeof(double)*rows*nd );
  for( int r=0; r<3; r++ ) hipLaunchKernelGGL( rkfd_traffic_cal, dim3( rows ), dim3( 64 ), 0, 0, in, out, nd );
  hipDeviceSynchronize();
  printf( "rows %zu nd %d: %zu bytes read and %zu bytes written per launch (3 launches)\n", rows, nd, sizeof(double)*rows*nd, sizeof(double)*rows*nd );
  return 0;
}
