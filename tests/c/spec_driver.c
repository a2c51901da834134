/* rkfdBatchSpecialize from a plain C program (gcc, no Python, no preloaded compiler library): the world-specific step
 * kernel must compile, load and give the results of the generic kernel bit for bit.  GPU test. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "roki_fd_amd.h"

#define B 64
int main(int argc, char *argv[])
{
  rkFD fd;
  char path[1024];
  const rkfdModel *m;
  rkfdBatch *b[2];
  double *dis, *vel, *out[2];
  int i, k, n, rc = 0;

  if( argc < 2 ) return 1;
  rkFDCreate( &fd );
  snprintf( path, sizeof(path), "%s/contact_rigid.ztk", argv[1] ); rkFDContactInfoScanFile( &fd, path );
  snprintf( path, sizeof(path), "%s/humanoid30.ztk", argv[1] ); if( !rkFDChainRegFile( &fd, path ) ) return 2;
  snprintf( path, sizeof(path), "%s/floor.ztk", argv[1] ); if( !rkFDChainRegFile( &fd, path ) ) return 2;
  rkFDSetSolver( &fd, MLCP );
  if( !( m = rkFDBuildModel( &fd ) ) ) return 3;
  n = m->ndof;
  dis = (double *)calloc( (size_t)B*n, sizeof(double) ); vel = (double *)calloc( (size_t)B*n, sizeof(double) );
  out[0] = (double *)calloc( (size_t)B*n*3, sizeof(double) ); out[1] = (double *)calloc( (size_t)B*n*3, sizeof(double) );
  for( i=0; i<B; i++ ){
    dis[i*n+2] = 0.3667 - 0.0006 - 1.0e-5*i;                       /* base height: the soles just in the floor */
    for( k=6; k<n; k++ ) dis[i*n+k] = 0.001*( ( i*31 + k*7 )%13 - 6 );
  }
  for( k=0; k<2; k++ ){
    if( !( b[k] = rkfdBatchCreate( m, B, 0, 8 ) ) ){ fprintf( stderr, "%s\n", rkfdHipLastError() ); return 4; }
    if( k == 1 && rkfdBatchSpecialize( b[k] ) < 0 ){ fprintf( stderr, "specialize: %s\n", rkfdHipLastError() ); return 5; }
    if( rkfdBatchSetState( b[k], dis, vel ) < 0 || rkfdBatchUpdateInit( b[k], NULL ) < 0 || rkfdBatchUpdate( b[k], 12, NULL ) < 0 ) return 6;
    if( rkfdBatchStatus( b[k], NULL ) != 0 ){ fprintf( stderr, "status: %s\n", rkfdHipLastError() ); return 7; }
    if( rkfdBatchGetState( b[k], out[k], out[k]+(size_t)B*n, out[k]+(size_t)2*B*n ) < 0 ) return 8;
  }
  if( memcmp( out[0], out[1], sizeof(double)*(size_t)B*n*3 ) != 0 ){ printf( "specialised kernel differs from the generic one\n" ); rc = 9; }
  else printf( "specialised == generic over %d instances x 12 steps\n", B );
  {
    /* which compiler library served hipRTC: the one beside this build's ROCm, in its own link namespace */
    FILE *fp = fopen( "/proc/self/maps", "r" );
    char line[2048], seen[1024] = "";
    while( fp && fgets( line, sizeof(line), fp ) ){
      char *p = strstr( line, "libamd_comgr" );
      if( p ){ char *q = strrchr( line, ' ' ); if( q && !strstr( seen, q+1 ) ){ strncat( seen, q+1, sizeof(seen)-strlen(seen)-1 ); printf( "comgr: %s", q+1 ); } }
    }
    if( fp ) fclose( fp );
  }
  rkfdBatchDestroy( b[0] ); rkfdBatchDestroy( b[1] );
  rkFDDestroy( &fd );
  free( dis ); free( vel ); free( out[0] ); free( out[1] );
  return rc;
}
