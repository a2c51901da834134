/* boxdrop_hardsoft.c - the scenario of the reference's example/chain/boxdrop_hardsoft_test.c
 * written against include/roki_fd_amd.h: boxes dropped on a floor that is rigid on one half
 * and elastic on the other.  The random initial attitudes of the reference driver are replaced
 * by fixed ones so that the run is reproducible.  Every rkFDUpdate runs on the GPU.
 *
 * build: gcc -O2 -Iinclude examples/boxdrop_hardsoft.c -Lroki-fd_amd -lrkfd_amd -Wl,-rpath,$PWD/roki-fd_amd -o boxdrop
 * usage: ./boxdrop [nbox] [steps] [model dir] [mlcp|vert|volume]
 */
#include <stdlib.h>
#include <string.h>
#include "roki_fd_amd.h"

#define DT    0.001
#define NMAX  9

int main(int argc, char *argv[])
{
  rkFD fd;
  int i, k, n, steps;
  rkFDCell *cell[NMAX];
  zVec dis[NMAX];
  char name[BUFSIZ];
  const char *dir = argc > 3 ? argv[3] : "models";

  rkFDCreate( &fd );
  snprintf( name, sizeof(name), "%s/contactinfo.ztk", dir );
  if( !rkFDContactInfoScanFile( &fd, name ) ) return 1;

  n = argc > 1 ? atoi( argv[1] ) : 2;
  steps = argc > 2 ? atoi( argv[2] ) : 500;
  if( n > NMAX ) n = NMAX;
  for( i=0; i<n; i++ ){
    snprintf( name, sizeof(name), "%s/box.ztk", dir );
    if( !( cell[i] = rkFDChainRegFile( &fd, name ) ) ) return 1;
    dis[i] = zVecAlloc( rkChainJointSize( rkFDCellChain(cell[i]) ) );
    zVecElemNC(dis[i],0) = 0.3*i;
    zVecElemNC(dis[i],1) = ( i % 2 ) ? 1.0 : -1.0;      /* odd boxes over the rigid half, even over the soft half */
    zVecElemNC(dis[i],2) = 0.1 + i*0.05;
    zVecElemNC(dis[i],3) = zDeg2Rad( 10.0*(i+1) );
    zVecElemNC(dis[i],4) = zDeg2Rad( -7.0*(i+1) );
    zVecElemNC(dis[i],5) = zDeg2Rad( 5.0*(i+1) );
    rkFDChainSetDis( cell[i], dis[i] );
    rkCDPairChainUnreg( rkFDCDBase(&fd.cd), rkFDCellChain(cell[i]) );
  }
  snprintf( name, sizeof(name), "%s/floor_hardsoft.ztk", dir );
  if( !rkFDChainRegFile( &fd, name ) ) return 1;

  rkFDODE2Assign( &fd, Regular );
  rkFDODE2AssignRegular( &fd, RKG );
  rkFDPrpSetDT( &fd, DT );
  /* rkFDCreate selects the Vert plugin (reference src/rkfd_sim.c:52); the reference's drivers switch
   * with rkFDSetSolver( &fd, MLCP ) / ( &fd, Vert ) - the fourth argument picks one here */
  if( argc > 4 && strcmp( argv[4], "vert" ) == 0 ) rkFDSetSolver( &fd, Vert );
  else if( argc > 4 && strcmp( argv[4], "volume" ) == 0 ) rkFDSetSolver( &fd, Volume );      /* what the reference's own driver selects */
  else rkFDSetSolver( &fd, MLCP );

  rkFDUpdateInit( &fd );
  if( rkFDStatus( &fd ) != 0 ) return 2;
  for( k=0; k<steps; k++ ){
    rkFDUpdate( &fd );
    if( rkFDStatus( &fd ) != 0 ) return 2;
  }
  rkFDUpdateDestroy( &fd );
  printf( "t %.6f\n", rkFDTime(&fd) );
  for( i=0; i<n; i++ ){
    rkChainGetJointDisAll( rkFDCellChain(cell[i]), dis[i] );
    printf( "box %d", i );
    for( k=0; k<6; k++ ) printf( " %.12e", zVecElemNC(dis[i],k) );
    printf( "\n" );
    zVecFree( dis[i] );
  }
  rkFDDestroy( &fd );
  return 0;
}
