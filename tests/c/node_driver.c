/* node_driver.c - the node level of the C ABI from plain C (gcc, no HIP headers): config 4's world on every visible GPU,
 * `total` instances sharded over them, `steps` x rkFDUpdate without any per-step communication, then the one collective:
 * rkfdNodeGather (ncclAllGather of the final {dis, vel}).  Prints a checksum of the gathered states and of the plain copies.
 * build: gcc -O1 -Iinclude tests/c/node_driver.c -Lroki-fd_amd -lrkfd_amd -Wl,-rpath,$PWD/roki-fd_amd -o node_driver
 * usage: node_driver <model dir> <total> <steps> [ndev] */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "roki_fd_amd.h"
#include "rkfd_hip.h"

int main(int argc, char *argv[])
{
  const char *dir = argc > 1 ? argv[1] : "models";
  const int total = argc > 2 ? atoi( argv[2] ) : 64, steps = argc > 3 ? atoi( argv[3] ) : 10, ndev = argc > 4 ? atoi( argv[4] ) : 0;
  char name[1024];
  rkfdWorldHandle *w = rkfdWorldCreate();
  const rkfdModel *m;
  rkfdNode *node;
  double *dis, *vel, *gd, *gv, *init;
  int h, i, k, st;
  double s1 = 0, s2 = 0;

  snprintf( name, sizeof(name), "%s/contact_rigid.ztk", dir );
  if( rkfdWorldSetContactInfo( w, name ) != 0 ) return 1;
  snprintf( name, sizeof(name), "%s/humanoid30.ztk", dir );
  if( ( h = rkfdWorldRegFile( w, name ) ) < 0 ) return 1;
  snprintf( name, sizeof(name), "%s/floor.ztk", dir );
  if( rkfdWorldRegFile( w, name ) < 0 ) return 1;
  rkfdWorldPairChainUnreg( w, h );
  rkfdWorldSetPrp( w, 0.001, 100.0, 10, RKFD_SOLVER_MLCP );
  if( !( m = rkfdWorldModel( w ) ) ) return 1;

  node = rkfdNodeCreate( m, total, 8, ndev, NULL );
  if( !node ){ fprintf( stderr, "rkfdNodeCreate: %s\n", rkfdHipLastError() ); return 2; }
  printf( "devices %d\n", rkfdNodeDevices( node ) );
  for( k=0; k<rkfdNodeDevices( node ); k++ ){
    int dev, lo, hi;
    rkfdNodeShard( node, k, &dev, &lo, &hi );
    printf( "shard %d: device %d, instances [%d, %d)\n", k, dev, lo, hi );
  }
  dis = (double *)calloc( (size_t)total*m->ndof, sizeof(double) ); vel = (double *)calloc( (size_t)total*m->ndof, sizeof(double) );
  gd = (double *)calloc( (size_t)total*m->ndof, sizeof(double) ); gv = (double *)calloc( (size_t)total*m->ndof, sizeof(double) );
  init = (double *)calloc( (size_t)m->ndof, sizeof(double) );
  rkfdWorldChainInitDis( w, h, init );
  for( i=0; i<total; i++ ){
    memcpy( dis + (size_t)i*m->ndof, init, sizeof(double)*m->ndof );
    dis[(size_t)i*m->ndof+2] += 0.002*( i % 7 );          /* every instance drops from its own height */
    dis[(size_t)i*m->ndof+8] += 0.01*( i % 5 );
  }
  if( rkfdNodeSetState( node, dis, vel ) < 0 || rkfdNodeUpdateInit( node ) < 0 || rkfdNodeUpdate( node, steps ) < 0 ){
    fprintf( stderr, "%s\n", rkfdHipLastError() ); return 2;
  }
  if( ( st = rkfdNodeStatus( node ) ) != 0 ){ fprintf( stderr, "status %d: %s\n", st, rkfdHipLastError() ); return 2; }
  if( rkfdNodeGather( node, gd, gv ) < 0 ){ fprintf( stderr, "rkfdNodeGather: %s\n", rkfdHipLastError() ); return 3; }
  if( rkfdNodeGetState( node, dis, vel, NULL ) < 0 ) return 2;
  for( i=0; i<total*m->ndof; i++ ){ s1 += gd[i]*( 1 + i % 13 ) + gv[i]; s2 += dis[i]*( 1 + i % 13 ) + vel[i]; }
  printf( "gathered %.17g\ncopied   %.17g\n", s1, s2 );
  printf( "%s\n", memcmp( gd, dis, sizeof(double)*(size_t)total*m->ndof ) == 0 && memcmp( gv, vel, sizeof(double)*(size_t)total*m->ndof ) == 0 ? "identical" : "DIFFERENT" );
  rkfdNodeDestroy( node );
  rkfdWorldFree( w );
  free( dis ); free( vel ); free( gd ); free( gv ); free( init );
  return 0;
}
